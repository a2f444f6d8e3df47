R=$(pwd); O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_handchain_gpu.py -q -k tied_object_scale > $O/g46.log 2>&1; tail -30 $O/g46.log | cut -c1-500
