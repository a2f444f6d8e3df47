R=$(pwd); O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_handchain_gpu.py -q > $O/g37_hand.log 2>&1; tail -40 $O/g37_hand.log | cut -c1-400
