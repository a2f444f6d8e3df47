"""Work counters of k_bwd_sweep in the fused loop of the pose initialisation (debug build: tools/ab_build.sh stats -DSWEEP_STATS
-DRASTER_PHASES; HOMAN_AMD_LIB=variants/lib_stats.so python tools/poseinit_stats.py).  GPU box."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from homan_amd import lib as _lib, ops, pose_optimization as po, synth  # noqa: E402

n, size = 500, 256
ov, of = synth.bottle_mesh()
verts, faces = torch.from_numpy(ov), torch.from_numpy(of).long()
K = np.array([[480.0, 0, 175.0], [0, 480.0, 175.0], [0, 0, 1.0]], np.float32)
sq = np.array([75.0, 60.0, 200.0, 200.0], np.float32)
Rgt = torch.tensor(synth._rot_x(1.3) @ synth._rot_y(0.4), dtype=torch.float32)
tgt_pose = (verts @ Rgt + torch.tensor([0.0, -0.02, 0.6]))[None]
roi = po.get_K_crop_resize(torch.as_tensor(K)[None], torch.tensor([[sq[0], sq[1], sq[0] + sq[2], sq[1] + sq[2]]]), [size])
roi[:, :2] /= size
tm = po.PoseOptimizer(ref_image=np.zeros((size, size), np.float32), vertices=verts, faces=faces,
                      rotation_init=po.matrix_to_rot6d(torch.eye(3)[None]), translation_init=torch.zeros(1, 1, 3), K=roi)
with torch.no_grad():
    mask = ops.silhouette_render_noaa(tgt_pose.cuda(), tm._K_all, tm._sil_ctx).cpu().numpy()[0]
ys, xs = np.nonzero(mask > 0)
bbox = np.array([sq[0] + xs.min() * sq[2] / size, sq[1] + ys.min() * sq[2] / size, (xs.max() - xs.min()) * sq[2] / size,
                 (ys.max() - ys.min()) * sq[2] / size], np.float32)
torch.manual_seed(0)
rots = po.compute_random_rotations(n)
trans0 = po.TCO_init_from_boxes_zup_autodepth(bbox, torch.matmul(verts.unsqueeze(0), rots), torch.as_tensor(K)[None]).unsqueeze(1)
L = _lib.lib()
L.hm_debug_sweep_stats.argtypes = [ctypes.c_void_p]
out = (ctypes.c_ulonglong * 16)()
names = ["s2_items", "s2_geo", "act0", "act1", "on0", "on1", "pairs", "s2_trips", "s1_items", "s1_geo", "s1_reach", "pair_rounds",
         "s1_own", "s1_a0", "s1_out_empty", "s1_a1"]
L.hm_debug_raster_phases.argtypes = [ctypes.c_void_p]
rout = (ctypes.c_ulonglong * 24)()
rnames = ["scan", "records", "near_units", "hz", "far_units", "tail"]
for steps in (3, 10, 25, 50):
    m = po.PoseOptimizer(ref_image=mask, vertices=verts, faces=faces, rotation_init=po.matrix_to_rot6d(rots), translation_init=trans0,
                         num_initializations=n, K=roi)
    po._fused_loop(m, 1e-2, steps - 1)
    L.hm_debug_sweep_stats(out)
    L.hm_debug_raster_phases(rout)
    po._fused_loop(m, 1e-2, 2)        # (two un-captured steps: counted below)
    L.hm_debug_sweep_stats(out)
    L.hm_debug_raster_phases(rout)
    act = max(1, int(rout[6]))
    print("   raster wave-0 cycles per active workgroup: " + "  ".join(f"{k}={int(v) / act:.0f}" for k, v in zip(rnames, rout)) +
          f"; active {int(rout[6]) // 2} idle {int(rout[7]) // 2} per launch; candidates near {int(rout[12]) // 2} far {int(rout[13]) // 2}; "
          f"units near {int(rout[8]) // 2} far {int(rout[9]) // 2}, far surviving {int(rout[21]) // 2}; covered pairs {int(rout[11]) // 2}")
    print(f"around step {steps}: per launch " + "  ".join(f"{k}={int(v) // 2}" for k, v in zip(names, out)))
