"""GPU box: many pose-initialisation fits through the resident fitter (alternating masks and meshes): device memory and the number
of resident fitters stay flat.  usage: python tools/soak_poseinit.py [fits]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from homan_amd import pose_optimization as po  # noqa: E402
from homan_amd import synth  # noqa: E402

fits = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n, size = 200, 128
meshes = [synth.bottle_mesh(), synth.box_mesh()]
K = np.array([[480.0, 0, 175.0], [0, 480.0, 175.0], [0, 0, 1.0]], np.float32)
sq = np.array([75.0, 60.0, 200.0, 200.0], np.float32)
yy, xx = np.mgrid[0:size, 0:size]
masks = [((xx - 64 - 10 * k) ** 2 / (20 + 4 * k) ** 2 + (yy - 64) ** 2 / 40 ** 2 < 1).astype(np.float32) for k in range(3)]
bbox = np.array([110.0, 90.0, 80.0, 120.0], np.float32)
torch.manual_seed(0)
rots = po.compute_random_rotations(n)
rows = []
t0 = time.perf_counter()
for i in range(fits):
    ov, of = meshes[i % 2]
    po.find_optimal_pose(torch.from_numpy(ov), torch.from_numpy(of).long(), masks[i % 3], bbox, sq, (350, 350), K=K, num_iterations=10,
                         num_initializations=n, rotations_init=rots, rend_size=size)
    if i % 20 == 19:
        torch.cuda.synchronize()
        rows.append(dict(fit=i + 1, allocated_MB=round(torch.cuda.memory_allocated() / 2 ** 20, 1),
                         reserved_MB=round(torch.cuda.memory_reserved() / 2 ** 20, 1), fitters=len(po._FITTERS)))
print(json.dumps(dict(fits=fits, seconds=round(time.perf_counter() - t0, 2), rows=rows)))
