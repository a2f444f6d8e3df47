R=$(pwd); O=$R/gpurun_out
timeout 2700 python -m pytest tests -m gpu -q > $O/g36_tests.log 2>&1; tail -6 $O/g36_tests.log
