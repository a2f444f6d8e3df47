R=$(pwd); O=$R/gpurun_out; mkdir -p $O; N=r05
cd $R
timeout 500 bash tools/pmc_loop.sh > $O/${N}_pmc_loop.json 2>/dev/null; echo "pmc_loop rc $?"
timeout 500 bash tools/pmc_loop.sh --step2 > $O/${N}_pmc_loop_cfg3.json 2>/dev/null; echo "pmc_loop cfg3 rc $?"
timeout 500 bash tools/pmc_poseinit.sh > $O/${N}_pmc_poseinit.json 2>/dev/null; echo "pmc_poseinit rc $?"
timeout 600 python tools/chain_parity.py cfg2 400 > $O/${N}_freerun_cfg2_400.json 2>/dev/null; echo "freerun rc $?"
ls -la $O | grep -E "pmc|freerun"
