R=$(pwd); O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_handchain_gpu.py -q -k ordinal_depth > $O/g42.log 2>&1; tail -30 $O/g42.log | cut -c1-600
