# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh'): the whole GPU suite + smoke
R=$(pwd); O=$R/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
