#!/bin/bash
# Short refresh of the round's evidence at the final tree (what changed after tools/profile_round.sh ran): bench lines of the default
# run / the driver's flags / cfg3 / cfg2 + depth, rocprofv3 kernel stats + mid-run timelines of the five profiled loops.
# usage (GPU box): bash tools/refresh_final_evidence.sh r06   (~8 min; copy gpurun_out/r06_* into profiles/)
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-r06}; O=$R/gpurun_out; mkdir -p $O; cd $R
b() { n=$1; shift; HOMAN_BENCH_DETAIL=$O/${N}_bench_$n.json python bench.py "$@" > $O/${N}_bench_${n}_line.json 2> $O/${N}_bench_$n.err; }
b cfg2_driver_flags --gpus 1 --steps 20 --warmup 5
b cfg2 --parity
b cfg3 --step2
b cfg2_depth --depth --multi-clip 4
python tools/bench_clips.py --clips 8 --steps 200 --mixed > $O/${N}_bench_mixed_shard.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2_profiled.json rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --no-cpu-baseline --legs '' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/pb -o pb -- python $R/tools/bench_clips.py --clips 8 --steps 100 > $O/${N}_bench_batch8_profiled.json 2>/dev/null
HOMAN_POSEINIT_LOOPS=fused rocprofv3 --kernel-trace --stats -d $O/pp -o pp -- python $R/bench.py --pose-init 500 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/pd -o pd -- python $R/bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/p3 -o p3 -- python $R/bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
cd $R
s() { python tools/prof_summary.py $O/$1/$1_results.db "$3" > $O/${N}_p_$2_kernel_stats.txt; python tools/prof_timeline.py $O/$1/$1_results.db > $O/${N}_p_$2_timeline.txt; }
s pd cfg2_depth "python bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 --legs ''"
s p3 cfg3 "python bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0 --legs ''"
s ph cfg2_headline "python bench.py --multi-clip 0 --no-cpu-baseline --legs ''"
s pb cfg4_batch8 "python tools/bench_clips.py --clips 8 --steps 100"
python tools/prof_summary.py $O/pp/pp_results.db "HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline" > $O/${N}_p_poseinit_kernel_stats.txt; python tools/prof_timeline.py $O/pp/pp_results.db k_pose_keep_best > $O/${N}_p_poseinit_timeline.txt      # (the window between two best-ever launches of ANY candidate group)
rm -rf $O/ph $O/pb $O/pp $O/pd $O/p3
