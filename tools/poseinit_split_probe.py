#!/usr/bin/env python
"""Probe: the pose initialisation's fused step over 500 candidates as ONE loop against TWO loops of 250 candidates replayed side by
side on two streams (the candidates are independent: would the small launches of one half hide under the heavy kernels of the other?).
usage: python tools/poseinit_split_probe.py [n] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from homan_amd import ops, synth  # noqa: E402
from homan_amd import pose_optimization as po  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
size = 256
ov, of = synth.bottle_mesh()
verts, faces = torch.from_numpy(ov), torch.from_numpy(of).long()
K = np.array([[480.0, 0, 175.0], [0, 480.0, 175.0], [0, 0, 1.0]], np.float32)
sq = np.array([75.0, 60.0, 200.0, 200.0], np.float32)
Rgt = torch.tensor(synth._rot_x(1.3) @ synth._rot_y(0.4), dtype=torch.float32)
tgt_pose = (verts @ Rgt + torch.tensor([0.0, -0.02, 0.6]))[None]
roi = po.get_K_crop_resize(torch.as_tensor(K)[None], torch.tensor([[sq[0], sq[1], sq[0] + sq[2], sq[1] + sq[2]]]), [size])
roi[:, :2] /= size
tgt_model = po.PoseOptimizer(ref_image=np.zeros((size, size), np.float32), vertices=verts, faces=faces,
                             rotation_init=po.matrix_to_rot6d(torch.eye(3)[None]), translation_init=torch.zeros(1, 1, 3), K=roi)
with torch.no_grad():
    mask = ops.silhouette_render_noaa(tgt_pose.cuda(), tgt_model._K_all, tgt_model._sil_ctx).cpu().numpy()[0]
ys, xs = np.nonzero(mask > 0)
bbox = np.array([sq[0] + xs.min() * sq[2] / size, sq[1] + ys.min() * sq[2] / size, (xs.max() - xs.min()) * sq[2] / size,
                 (ys.max() - ys.min()) * sq[2] / size], np.float32)
torch.manual_seed(0)
rots = po.compute_random_rotations(n)
trans0 = po.TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(verts.unsqueeze(0), rots), torch.as_tensor(K)[None]).unsqueeze(1)


def loop_for(sel):
    m = po.PoseOptimizer(ref_image=mask, vertices=verts, faces=faces, rotation_init=po.matrix_to_rot6d(rots[sel]),
                         translation_init=trans0[sel], num_initializations=len(sel), K=roi)
    lp = po._FusedPoseLoop(m, 1e-2)
    lp.run(3)              # two eager steps + the capture's first replay
    return lp


allc = list(range(n))
one = loop_for(allc)
halves = [loop_for(allc[:n // 2]), loop_for(allc[n // 2:])]
torch.cuda.synchronize()
out = {}
t0 = time.perf_counter()
for _ in range(steps):
    one.graph.replay()
torch.cuda.synchronize()
out["one_loop_pose_steps_per_s"] = n * steps / (time.perf_counter() - t0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t0 = time.perf_counter()
for _ in range(steps):
    with torch.cuda.stream(s1):
        halves[0].graph.replay()
    with torch.cuda.stream(s2):
        halves[1].graph.replay()
torch.cuda.synchronize()
out["two_half_loops_pose_steps_per_s"] = n * steps / (time.perf_counter() - t0)
t0 = time.perf_counter()
for _ in range(steps):
    halves[0].graph.replay()
    halves[1].graph.replay()
torch.cuda.synchronize()
out["two_half_loops_one_stream_pose_steps_per_s"] = n * steps / (time.perf_counter() - t0)
print(json.dumps(out))
