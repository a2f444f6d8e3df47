#!/bin/bash
# same-box A/B of two builds of libhoman_amd.so (alternated): headline / steady state / 8-clip batch of the default workload with
# the in-graph kernel times, then the pose initialisation.   usage: tools/ab2.sh variants/lib_a.so homan_amd/lib/libhoman_amd.so [reps]
A=$1; B=$2; REPS=${3:-2}
for rep in $(seq $REPS); do
  for L in $A $B; do
    HOMAN_AMD_LIB=$L HOMAN_BENCH_DETAIL=/tmp/ab2_detail.json python bench.py --no-cpu-baseline --legs '' --multi-clip 8 > /dev/null 2>&1
    python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab2_detail.json"))
k = (d.get("steady_state") or {}).get("roofline", {}).get("kernels", {})
print(sys.argv[1], "it/s %.0f" % d["value"], "steady %.0f" % (d.get("steady_state") or {}).get("value", 0),
      "batch8 %.0f" % (d.get("multi_clip") or {}).get("value", 0),
      " ".join("%s %.1f" % (n[2:], v["avg_launch_us"]) for n, v in k.items()))
PY
    HOMAN_AMD_LIB=$L python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   pose-init %.0f pose-steps/s' % d['value'])"
  done
done
