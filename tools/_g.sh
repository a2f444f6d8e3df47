# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g.sh')
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
tools/valu_ceiling > $O/r06_valu_ceiling.json 2> $O/valu_ceiling.err; tail -c 600 $O/r06_valu_ceiling.json
timeout 600 python -m pytest tests/test_poseinit.py tests/test_ortho.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
python bench.py 2>$O/bench0.err | tail -1 | cut -c1-1500
bash tools/ledger.sh r06
cat $O/r06_ledger.json
