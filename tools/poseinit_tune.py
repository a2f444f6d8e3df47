"""GPU box: the pose initialisation's per-step time against the sweep's workgroup count (hm_tune_sweep_blocks; results do not
depend on it).  usage: python tools/poseinit_tune.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from homan_amd import lib as hlib  # noqa: E402
from homan_amd import ops, synth  # noqa: E402
from homan_amd import pose_optimization as po  # noqa: E402

n, size = 500, 256
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
os.environ["HOMAN_POSE_FITTER"] = "0"          # a fresh capture per setting
fit = lambda k: po.find_optimal_pose(verts, faces, mask, bbox, sq, (350, 350), K=K, num_iterations=k, num_initializations=n,
                                     rotations_init=rots, rend_size=size)
res = {}
for blocks in [int(a) for a in (sys.argv[1:] or ["1280", "768", "1024", "1536", "2048", "2560", "4096", "1280"])]:
    hlib.lib().hm_tune_sweep_blocks(blocks)
    fit(3)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit(50)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    res.setdefault(str(blocks), []).append(round(best * 1e3, 2))
print(json.dumps(res))
