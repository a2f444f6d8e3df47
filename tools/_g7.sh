R=$(pwd); O=$R/gpurun_out
timeout 900 python -m pytest tests/test_poseinit.py -q -m gpu -k resident > $O/g41.log 2>&1; tail -30 $O/g41.log | cut -c1-400
