# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh'): the whole GPU suite with durations
R=$(pwd); O=$R/gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q --durations=25 > $O/g65_tests.log 2>&1; grep -A30 "slowest" $O/g65_tests.log | cut -c1-150; tail -2 $O/g65_tests.log
