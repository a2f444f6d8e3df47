#!/bin/bash
# same-box A/B of two builds of libhoman_amd.so: alternates them, prints it/s of the headline (400 steps), of the driver's short
# flags (20 + 5) and of the 8-clip batch.   usage: tools/ab.sh scratch/lib_base.so homan_amd/lib/libhoman_amd.so [extra bench flags]
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    HOMAN_AMD_LIB=$L python bench.py --parity-seeds 0 --no-cpu-baseline --lockstep 0 --steady 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['kernels']
print('$L', 'it/s %.0f' % d['value'], 'multi %.0f' % (d['multi_clip'] or {}).get('value',0), ' '.join('%s %.1f' % (k[2:], v['avg_launch_us']) for k,v in r.items()))"
    HOMAN_AMD_LIB=$L python bench.py --steps 20 --warmup 5 --multi-clip 0 --parity-seeds 0 --no-cpu-baseline --lockstep 0 --steady 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']['kernels']
print('$L', 'driver-flags it/s %.0f' % d['value'], ' '.join('%s %.1f' % (k[2:], v['avg_launch_us']) for k,v in r.items()))"
  done
done
