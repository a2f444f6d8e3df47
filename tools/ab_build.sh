#!/bin/bash
# builds a variant of libhoman_amd.so with extra -D flags into variants/ for A/B runs: HOMAN_AMD_LIB=variants/lib_<name>.so
# usage: tools/ab_build.sh <name> [-DFOO=1 ...]
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; shift
mkdir -p $R/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -Wno-unused-function "$@" -I $R/homan_amd/csrc -o $R/variants/lib_$name.so $R/homan_amd/csrc/*.hip && echo $R/variants/lib_$name.so
