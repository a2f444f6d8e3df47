R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_handchain_gpu.py -q -k two_hands > $O/g48.log 2>&1; tail -30 $O/g48.log | cut -c1-700
