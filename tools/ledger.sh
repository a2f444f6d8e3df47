#!/bin/bash
# Per-kernel ledger of the fused iteration's two chains: every configuration of tools/cfg1_floor.py under
# rocprofv3 --kernel-trace, per-kernel averages + one iteration's timeline.  cfg1 and b1 (ONE frame) are the kernels at their
# latency floors - almost no work, so what is left is launch + dependent round trips -, cfg2 is the headline workload.
# usage (GPU box): bash tools/ledger.sh r06        -> gpurun_out/r06_ledger_{cfg1,b1,cfg2}_{kernel_stats,timeline}.txt + r06_ledger.json
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-r06}; O=$R/gpurun_out; mkdir -p $O
cd $R
python tools/cfg1_floor.py cfg1 b1 cfg2 > $O/${N}_ledger.json 2>/dev/null      # un-profiled rates of the three
cd /tmp && export TMPDIR=/tmp
for c in cfg1 b1 cfg2; do
  rocprofv3 --kernel-trace --stats -d $O/lg_$c -o lg -- python $R/tools/cfg1_floor.py $c > /dev/null 2>&1
  python $R/tools/prof_summary.py $O/lg_$c/lg_results.db "python tools/cfg1_floor.py $c" | grep -v "_ZN2at\|_ZN12_GLOBAL" > $O/${N}_ledger_${c}_kernel_stats.txt
  python $R/tools/prof_timeline.py $O/lg_$c/lg_results.db > $O/${N}_ledger_${c}_timeline.txt
  rm -rf $O/lg_$c
done
