R=$(pwd); O=$R/gpurun_out
python -m pytest tests/test_raster_gpu.py tests/test_parity_gpu.py tests/test_clip_batch_gpu.py tests/test_poseinit.py -x -q -m gpu > $O/g21_t.log 2>&1
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0"
for i in 1 2; do
 (cd _ab_base && python bench.py --steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 > $O/g21_base_$i.json 2>/dev/null)
 python bench.py $F > $O/g21_new_$i.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 0 --freerun 0 --e2e-clips 0 --multi-clip 0 > $O/g21_new_drv.json 2>/dev/null
python bench.py --step2 $F --multi-clip 0 > $O/g21_new_cfg3.json 2>/dev/null
python bench.py --pose-init 500 --no-cpu-baseline > $O/g21_pi.json 2>/dev/null
