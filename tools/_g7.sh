R=$(pwd); O=$R/gpurun_out
python -m pytest tests -x -q -m gpu > $O/g13_t.log 2>&1
