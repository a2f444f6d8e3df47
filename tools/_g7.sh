R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_handchain_gpu.py -q --durations=12 > $O/g50.log 2>&1; tail -22 $O/g50.log | cut -c1-200
