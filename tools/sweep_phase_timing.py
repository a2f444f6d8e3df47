#!/usr/bin/env python
"""Where a k_bwd_sweep wave spends its cycles (head / item setup / line records / pairs / outputs), from the cycle-counter
marks compiled in with -DSWEEP_TIMING (csrc/raster.hip, hm_debug_sweep_timing).  Build an instrumented library next to the
product one and point this script at it:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -I homan_amd/csrc -DSWEEP_TIMING \\
          -o /tmp/libhoman_amd_swt.so homan_amd/csrc/*.hip
    python tools/sweep_phase_timing.py /tmp/libhoman_amd_swt.so

Shares are of wave-cycles (lane 0 of every wave, summed over the launches of tools/bench_raster.py); the instrumentation
itself costs ~10 %."""
import ctypes
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homan_amd import build as hbuild  # noqa: E402

hbuild.LIB_PATH = os.path.abspath(sys.argv[1])
from homan_amd import lib as hlib  # noqa: E402

L = hlib.lib()
L.hm_debug_sweep_timing.argtypes = [ctypes.c_void_p]
L.hm_debug_sweep_timing.restype = ctypes.c_int
sys.argv = ["bench_raster.py", "20", "both"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_raster.py"), run_name="__main__")
out = (ctypes.c_ulonglong * 8)()
L.hm_debug_sweep_timing(out)
t = [float(x) for x in out]
names = ["head (first face, offsets, records -> LDS)", "item setup + owner loads", "line records + counts + scan", "pairs",
         "outputs"]
tot = sum(t[:5])
for n, v in zip(names, t):
    print(f"{n:44s} {100 * v / tot:5.1f} %")
print(f"wave launches: {int(t[5])}")
