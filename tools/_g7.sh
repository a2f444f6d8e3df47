R=$(pwd); O=$R/gpurun_out
HOMAN_AMD_LIB=$R/variants/lib_u9.so python -m pytest tests/test_raster_gpu.py tests/test_parity_gpu.py -x -q -m gpu > $O/g22_t.log 2>&1
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0"
python bench.py $F > $O/g22_u8.json 2>/dev/null
HOMAN_AMD_LIB=$R/variants/lib_u9.so python bench.py $F > $O/g22_u9.json 2>/dev/null
HOMAN_AMD_LIB=$R/variants/lib_u9.so HOMAN_SWEEP_BLOCKS=1024 python bench.py $F > $O/g22_u9_b1024.json 2>/dev/null
HOMAN_AMD_LIB=$R/variants/lib_u10.so HOMAN_SWEEP_BLOCKS=1024 python bench.py $F > $O/g22_u10_b1024.json 2>/dev/null
HOMAN_AMD_LIB=$R/variants/lib_u10.so HOMAN_SWEEP_BLOCKS=768 python bench.py $F > $O/g22_u10_b768.json 2>/dev/null
HOMAN_AMD_LIB=$R/variants/lib_u9.so HOMAN_SWEEP_BLOCKS=1024 python bench.py --steps 20 --warmup 5 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 0 --freerun 0 --e2e-clips 0 --multi-clip 0 > $O/g22_u9_drv.json 2>/dev/null
