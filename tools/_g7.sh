R=$(pwd); O=$R/gpurun_out
python -m pytest tests/test_parity_gpu.py tests/test_clip_batch_gpu.py tests/test_model_gpu.py tests/test_poseinit.py -x -q -m gpu > $O/g7_t.log 2>&1
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0"
for i in 1 2; do
 (cd _ab_base && python bench.py --steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 > $O/g7_base_$i.json 2>/dev/null)
 python bench.py $F > $O/g7_new_$i.json 2>/dev/null
 HOMAN_RIGID_CHUNKED=1 python bench.py $F > $O/g7_chunk_$i.json 2>/dev/null
done
