R=$(pwd); O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_handchain_gpu.py -q -k free_object_scale > $O/g39_hand.log 2>&1; tail -30 $O/g39_hand.log | cut -c1-500
