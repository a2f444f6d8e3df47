#!/bin/bash
# Regenerates the round's evidence under gpurun_out/ on the GPU box (copy what is to be judged into profiles/):
#   rNN_pmc_loop.json / _cfg3     PMC passes over the steady-state loop (tools/pmc_loop.sh [--step2]); bench.py reads them from profiles/
#   rNN_pmc_poseinit.json         the same for the fused loop of the pose initialisation (tools/pmc_poseinit.sh)
#   rNN_bench_cfg{2,3,5}.json     bench lines (cfg2 = the driver's default command; _driver_flags = --steps 20 --warmup 5)
#   rNN_p_cfg2_headline_*         rocprofv3 --kernel-trace --stats of the headline loop alone: its per-kernel averages are
#                                 the ones bench.py's roofline object must agree with
#   rNN_p_cfg4_batch8_*           the same for an 8-clip batch;  rNN_p_poseinit_*  for the pose initialisation's fused loop
#   rNN_freerun_cfg2_400.json     400 free-running steps, HIP loop vs the oracle's reproducible loop (object parameters bit-equal)
#   rNN_valu_ceiling.json / rNN_valu_mix.json   cycles per wave64 VALU instruction by opcode class; the heavy kernels' static mix priced with it
#   rNN_ledger_*                  per-kernel ledger of the iteration at one frame / cfg1 / cfg2 (tools/ledger.sh)
#   rNN_chain_only.txt            the shipped launch graph vs the silhouette chain alone vs both chains without edges
#   rNN_raster_trace*.txt         per-workgroup phase stamps of the rasteriser (debug build -DRASTER_TRACE)
# usage (GPU box): bash tools/profile_round.sh r06   (~40 min; copy gpurun_out/r06_* into profiles/)
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-r06}; O=$R/gpurun_out; mkdir -p $O
cd $R
# (PROFILE_SKIP_PMC=1 / PROFILE_SKIP_FREERUN=1: a shorter refresh that keeps the committed PMC passes / free run)
if [ -z "$PROFILE_SKIP_PMC" ]; then
bash tools/pmc_loop.sh > $O/${N}_pmc_loop.json 2>/dev/null
cp $O/${N}_pmc_loop.json profiles/${N}_pmc_loop.json
bash tools/pmc_loop.sh --step2 > $O/${N}_pmc_loop_cfg3.json 2>/dev/null
cp $O/${N}_pmc_loop_cfg3.json profiles/${N}_pmc_loop_cfg3.json
bash tools/pmc_poseinit.sh > $O/${N}_pmc_poseinit.json 2>/dev/null
cp $O/${N}_pmc_poseinit.json profiles/${N}_pmc_poseinit.json
bash tools/pmc_loop.sh --depth > $O/${N}_pmc_loop_depth.json 2>/dev/null       # cfg2 as BASELINE words it: three rasters per iteration, averaged apart
cp $O/${N}_pmc_loop_depth.json profiles/${N}_pmc_loop_depth.json
fi
if [ -z "$PROFILE_SKIP_MICRO" ]; then
[ -x tools/valu_ceiling ] && tools/valu_ceiling > $O/${N}_valu_ceiling.json 2>/dev/null && python tools/valu_mix.py $O/${N}_valu_ceiling.json > $O/${N}_valu_mix.json 2>/dev/null
bash tools/ledger.sh $N
if [ -x tools/fetch_calib ]; then      # FETCH_SIZE / WRITE_SIZE on narrow and scattered accesses of known size
  FC=$O/fetch_calib; rm -rf $FC; mkdir -p $FC
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $FC -o f -- $R/tools/fetch_calib > $FC/requested.json 2>/dev/null
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $FC -o w -- $R/tools/fetch_calib > /dev/null 2>&1 )
  python tools/fetch_calib_summary.py $FC > $O/${N}_fetch_calib.json; rm -rf $FC
fi
python tools/chain_only.py cfg2 2>&1 | grep -v amdgpu.ids > $O/${N}_chain_only.txt
if [ -f variants/lib_trace.so ]; then
  for f in 1 30; do HOMAN_AMD_LIB=variants/lib_trace.so python tools/raster_trace.py --frames $f 2>/dev/null | tail -6; done > $O/${N}_raster_trace.txt
  for d in sil obj hand; do HOMAN_AMD_LIB=variants/lib_trace.so python tools/raster_trace.py --depth $d 2>/dev/null | tail -6; done > $O/${N}_raster_trace_depth.txt
fi
fi
# bench lines: the COMPACT line bench.py prints goes to *_line.json, the full record (bench.py's detail file) to *.json
b() { # name, bench flags...
  n=$1; shift
  HOMAN_BENCH_DETAIL=$O/${N}_bench_$n.json python bench.py "$@" > $O/${N}_bench_${n}_line.json 2> $O/${N}_bench_$n.err
}
b cfg2_driver_flags --gpus 1 --steps 20 --warmup 5          # the driver's command: iterations 5-25 of a fresh fit
b cfg2 --parity                                             # the default workload + the opt-in parity / end-to-end legs
b cfg3 --step2
b cfg5_n1 --shared-scale --steps 200
b cfg2_depth --depth --multi-clip 4                         # cfg2 as BASELINE.json words it (sil/kp/depth/smooth)
b poseinit --pose-init 500                                  # object-pose initialisation
[ -z "$PROFILE_SKIP_FREERUN" ] && python tools/chain_parity.py cfg2 400 > $O/${N}_freerun_cfg2_400.json 2>/dev/null
[ -z "$PROFILE_SKIP_FREERUN" ] && python tools/chain_parity.py cfg3 400 > $O/${N}_freerun_cfg3_400.json 2>/dev/null
[ -z "$PROFILE_SKIP_FREERUN" ] && python tools/chain_parity.py cfg2depth 400 > $O/${N}_freerun_cfg2_depth_400.json 2>/dev/null
python tools/bench_clips.py --clips 8 --steps 200 --mixed > $O/${N}_bench_mixed_shard.json 2>/dev/null
# N > 1 ranks on the one GPU of this box (gloo moves the collectives' 4 bytes through the host): bench.py starts its ranks itself
HOMAN_BENCH_BACKEND=gloo b cfg2_gpus2_gloo --gpus 2 --steps 200 --warmup 20 --multi-clip 2 --steady 0
HOMAN_BENCH_BACKEND=gloo b cfg5_gpus2_gloo --gpus 2 --shared-scale --multi-clip 4 --steps 100 --warmup 10
cd /tmp && export TMPDIR=/tmp
HEAD="python bench.py --multi-clip 0 --no-cpu-baseline --legs ''"
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2_profiled.json rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --no-cpu-baseline --legs '' > /dev/null 2>&1
BATCH="python tools/bench_clips.py --clips 8 --steps 100"
rocprofv3 --kernel-trace --stats -d $O/pb -o pb -- python $R/tools/bench_clips.py --clips 8 --steps 100 > $O/${N}_bench_batch8_profiled.json 2>/dev/null
POSE="HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline"
HOMAN_POSEINIT_LOOPS=fused rocprofv3 --kernel-trace --stats -d $O/pp -o pp -- python $R/bench.py --pose-init 500 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/pd -o pd -- python $R/bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/p3 -o p3 -- python $R/bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pd/pd_results.db "python bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0" > $O/${N}_p_cfg2_depth_kernel_stats.txt
python tools/prof_timeline.py $O/pd/pd_results.db > $O/${N}_p_cfg2_depth_timeline.txt
python tools/prof_summary.py $O/p3/p3_results.db "python bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0" > $O/${N}_p_cfg3_kernel_stats.txt
python tools/prof_timeline.py $O/p3/p3_results.db > $O/${N}_p_cfg3_timeline.txt
python tools/prof_summary.py $O/ph/ph_results.db "$HEAD" > $O/${N}_p_cfg2_headline_kernel_stats.txt
python tools/prof_timeline.py $O/ph/ph_results.db > $O/${N}_p_cfg2_headline_timeline.txt
python tools/prof_summary.py $O/pb/pb_results.db "$BATCH" > $O/${N}_p_cfg4_batch8_kernel_stats.txt
python tools/prof_timeline.py $O/pb/pb_results.db > $O/${N}_p_cfg4_batch8_timeline.txt
python tools/prof_summary.py $O/pp/pp_results.db "$POSE" > $O/${N}_p_poseinit_kernel_stats.txt
python tools/prof_timeline.py $O/pp/pp_results.db > $O/${N}_p_poseinit_timeline.txt
rm -rf $O/ph $O/pb $O/pp $O/pd $O/p3
