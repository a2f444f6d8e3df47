R=$(pwd); O=$R/gpurun_out
timeout 900 python -m pytest tests/test_poseinit.py -q -m gpu -k "written_out_oracle or resident" > $O/g44.log 2>&1; tail -30 $O/g44.log | cut -c1-500
