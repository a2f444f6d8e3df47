"""GPU box: where the fixed cost of one find_optimal_pose fit goes (constructor, loop setup + capture, replays, ranking).
usage: python tools/poseinit_phases.py [poses] [size]"""
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
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


fit = lambda k: po.find_optimal_pose(verts, faces, mask, bbox, sq, (350, 350), K=K, num_iterations=k, num_initializations=n,
                                     rotations_init=rots, rend_size=size)
fit(3)
res = {}
for k in (3, 50, 100):
    res[f"fit_{k}_ms"] = min(timed(lambda: fit(k))[1] for _ in range(3))
dev = torch.device("cuda")
v, f = verts.float().to(dev), faces.to(dev)
r = rots.float().to(dev)
_, res["tco_init_ms"] = timed(lambda: po.TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(v.unsqueeze(0), r),
                                                                            torch.as_tensor(K)[None].to(dev)).unsqueeze(1))
t0 = po.TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(v.unsqueeze(0), r), torch.as_tensor(K)[None].to(dev)).unsqueeze(1)
mk = lambda: po.PoseOptimizer(ref_image=mask, vertices=v, faces=f, rotation_init=po.matrix_to_rot6d(r), translation_init=t0,
                              num_initializations=n, K=roi.to(dev))
res["ctor_ms"] = min(timed(mk)[1] for _ in range(3))
model = mk()
_, res["sil_ctx_ms"] = timed(lambda: ops.SilhouetteContext(model.faces, model.vertices.shape[1], n, size // 2, dev))
for k in (2, 3, 50):
    res[f"loop_{k}_ms"] = min(timed(lambda: po._fused_loop(mk(), 1e-2, k))[1] for _ in range(2)) - res["ctor_ms"]
out = po._fused_loop(model, 1e-2, 3)
_, res["rank_ms"] = timed(lambda: po._install_ranked_poses(model, out[0], out[1], out[2], True))
res["per_step_ms"] = (res["fit_100_ms"] - res["fit_50_ms"]) / 50
res["fixed_ms"] = res["fit_50_ms"] - 50 * res["per_step_ms"]
print(json.dumps(res))
