R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/pc -o pc -- python $R/tools/cfg1_floor.py > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pc/pc_results.db "python tools/cfg1_floor.py" | head -16 | cut -c1-150
python tools/prof_timeline.py $O/pc/pc_results.db k_adam 2405
rm -rf $O/pc
