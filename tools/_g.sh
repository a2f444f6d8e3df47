R=$(pwd); O=$R/gpurun_out; mkdir -p $O
HOMAN_BENCH_DETAIL=$O/r06_bench_cfg3.json python bench.py --step2 > $O/r06_bench_cfg3_line.json 2> $O/r06_bench_cfg3.err
tail -1 $O/r06_bench_cfg3_line.json | cut -c1-300
python bench.py 2>/dev/null | tail -1 > $O/r06_bench_default_line.json; python -c "
import json; d=json.load(open('$O/r06_bench_default_line.json')); print(d['value'], d['steady_state'], d['cfg2_depth'], d['cfg3'], d['multi_clip'])"
