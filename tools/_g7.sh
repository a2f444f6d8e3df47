# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh'): the whole GPU suite + smoke + the default bench line
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q --durations=12 > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -2; grep " call " $O/gpu_tests.log | head -8 | cut -c1-120
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py 2>$O/bench.err ) 2>&1 | tail -5 | cut -c1-2100
