R=$(pwd); O=$R/gpurun_out; mkdir -p $O
for rep in 1 2; do for v in base spec basepre specpre; do echo "== $v"; HOMAN_AMD_LIB=variants/lib_$v.so CHAIN_SKIP=1 python tools/chain_only.py cfg2 2>&1 | grep "shipped:\|main_only:"; done; done
