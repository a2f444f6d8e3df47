# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g.sh')
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python tools/chain_only.py cfg2 2>&1 | tail -12
