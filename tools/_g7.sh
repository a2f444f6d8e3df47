R=$(pwd); O=$R/gpurun_out
python tools/poseinit_tune.py > $O/g47_tune.json 2>$O/g47.err; cat $O/g47_tune.json; tail -2 $O/g47.err
