#!/bin/bash
# Regenerates the round's evidence under gpurun_out/ on the GPU box (copy what is to be judged into profiles/):
#   rNN_pmc_loop.json            PMC passes over the steady-state loop (tools/pmc_loop.sh); bench.py reads profiles/rNN_pmc_loop.json
#   rNN_bench_cfg{2,3,5}.json    bench lines (cfg2 = the driver's default command; _driver_flags = --steps 20 --warmup 5)
#   rNN_p_cfg2_headline_*        rocprofv3 --kernel-trace --stats of the headline loop alone: its per-kernel averages are
#                                the ones bench.py's roofline object must agree with
#   rNN_p_cfg4_batch8_*          the same for an 8-clip batch
# usage (GPU box): bash tools/profile_round.sh r03
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-r03}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash tools/pmc_loop.sh > $O/${N}_pmc_loop.json 2>/dev/null
cp $O/${N}_pmc_loop.json profiles/${N}_pmc_loop.json
python bench.py > $O/${N}_bench_cfg2.json 2> $O/${N}_bench_cfg2.err
python bench.py --steps 20 --warmup 5 > $O/${N}_bench_cfg2_driver_flags.json 2>/dev/null      # the flags the driver passed in round 1: iterations 5-25 of a fresh fit
python bench.py --step2 --parity-seeds 0 > $O/${N}_bench_cfg3.json 2>/dev/null
python bench.py --shared-scale --steps 200 > $O/${N}_bench_cfg5_n1.json 2>/dev/null
python bench.py --depth --parity-seeds 0 --multi-clip 0 > $O/${N}_bench_cfg2_depth.json 2>/dev/null          # cfg2 as BASELINE.json words it (sil/kp/depth/smooth)
python bench.py --pose-init 500 > $O/${N}_bench_poseinit.json 2>/dev/null                                     # object-pose initialisation
# N > 1 ranks on the one GPU of this box (gloo moves the collectives' 4 bytes through the host): the launch line of the driver
HOMAN_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 200 --warmup 20 --multi-clip 2 --steady 0 > $O/${N}_bench_cfg2_gpus2_gloo.json 2>/dev/null
HOMAN_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 \
    bench.py --gpus 2 --shared-scale --multi-clip 4 --steps 100 --warmup 10 > $O/${N}_bench_cfg5_gpus2_gloo.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
HEAD="python bench.py --multi-clip 0 --parity-seeds 0 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --parity-seeds 0 --no-cpu-baseline > $O/${N}_bench_cfg2_profiled.json 2>/dev/null
BATCH="python tools/bench_clips.py --clips 8 --steps 100"
rocprofv3 --kernel-trace --stats -d $O/pb -o pb -- python $R/tools/bench_clips.py --clips 8 --steps 100 > $O/${N}_bench_batch8_profiled.json 2>/dev/null
cd $R
python tools/prof_summary.py $O/ph/ph_results.db "$HEAD" > $O/${N}_p_cfg2_headline_kernel_stats.txt
python tools/prof_timeline.py $O/ph/ph_results.db > $O/${N}_p_cfg2_headline_timeline.txt
python tools/prof_summary.py $O/pb/pb_results.db "$BATCH" > $O/${N}_p_cfg4_batch8_kernel_stats.txt
python tools/prof_timeline.py $O/pb/pb_results.db > $O/${N}_p_cfg4_batch8_timeline.txt
rm -rf $O/ph $O/pb
