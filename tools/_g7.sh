R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_handchain_gpu.py -q -k fixed_hand_mesh > $O/g51.log 2>&1; tail -25 $O/g51.log | cut -c1-500
