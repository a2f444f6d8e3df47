set -x
R=$(pwd); O=$R/gpurun_out
python tools/chain_parity.py cfg1 20 > $O/cp_cfg1.json 2> $O/cp_cfg1.err
python tools/chain_parity.py cfg2small 20 > $O/cp_cfg2small.json 2> $O/cp_cfg2small.err
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000"
for i in 1 2; do
 (cd _ab_base && python bench.py $F > $O/ab_base_$i.json 2>/dev/null)
 python bench.py $F > $O/ab_new_$i.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --parity-seeds 0 --lockstep 0 --no-cpu-baseline > /dev/null 2>&1
cd $R/_ab_base && cd /tmp && rocprofv3 --kernel-trace --stats -d $O/phb -o phb -- python $R/_ab_base/bench.py --multi-clip 0 --parity-seeds 0 --lockstep 0 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/ph/ph_results.db new > $O/ab_new_kernel_stats.txt
python tools/prof_summary.py $O/phb/phb_results.db base > $O/ab_base_kernel_stats.txt
rm -rf $O/ph $O/phb
