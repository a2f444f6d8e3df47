R=$(pwd); O=$R/gpurun_out
python tools/chain_parity.py cfg2depth 400 > $O/g43_cfg2depth_400.json 2>$O/g43.err; tail -c 300 $O/g43_cfg2depth_400.json; tail -3 $O/g43.err
