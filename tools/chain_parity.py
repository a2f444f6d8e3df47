"""GPU box: free-running HIP loop vs the CPU oracle's reproducible loop (bench.free_run_parity), a few configurations.
usage: python tools/chain_parity.py [cfg1|cfg2small|cfg2] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
import bench  # noqa: E402
from homan_amd import synth  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mano = synthetic_mano(0)
if which == "cfg1":
    out = bench.free_run_parity(mano, steps=steps, frames=10, size=128, obj="cube", lw=dict(synth.CFG1_LOSS_WEIGHTS))
elif which == "cfg2small":
    out = bench.free_run_parity(mano, steps=steps, frames=6, size=64, obj="bottle")
else:
    out = bench.free_run_parity(mano, steps=steps, frames=30, size=256, obj="bottle")
print(json.dumps(out))
