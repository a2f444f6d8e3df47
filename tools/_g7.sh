# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh'): the whole GPU suite, the default bench line, smoke
R=$(pwd); O=$R/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > $O/g60_tests.log 2>&1; tail -4 $O/g60_tests.log
( time python bench.py > $O/g60_bench_default.json 2>$O/g60_bench.err ) 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
