R=$(pwd); O=$R/gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "mano_lbs or collision" > $O/g45.log 2>&1; tail -30 $O/g45.log | cut -c1-400
