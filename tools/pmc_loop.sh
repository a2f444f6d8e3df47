#!/bin/bash
# HBM traffic and execution-unit counters of the heavy silhouette kernels IN THE STEADY-STATE OPTIMISATION LOOP (persistent
# outputs on, hand-side stream running next to them): separate rocprofv3 --pmc passes (kernel-trace only) over
# tools/bench_clips.py, one clip of cfg2 [or "--step2" / "--clips N" passed through], averaged over the last LAST launches
# of every kernel.  Usage (GPU box): bash tools/pmc_loop.sh [bench_clips args...] > profiles/rNN_pmc_loop.json
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_loop; rm -rf $O; mkdir -p $O
STEPS=30; LAST=20
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O -o p$i -- python $R/tools/bench_clips.py --clips 1 --steps $STEPS --warmup 5 "$@" > $O/run$i.log 2>&1
done
cd $R
python tools/pmc_loop_summary.py $O $LAST
rm -rf $O
