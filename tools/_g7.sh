# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
python tools/soak_poseinit.py 160 > $O/r04_soak_poseinit.json 2>$O/g66.err; cat $O/r04_soak_poseinit.json; tail -2 $O/g66.err
