R=$(pwd); O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_handchain_gpu.py -q > $O/g32_hand.log 2>&1; tail -40 $O/g32_hand.log
