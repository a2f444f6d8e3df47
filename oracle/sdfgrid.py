"""CPU restatement of the `sdf.SDF` callable (voxel signed-distance grid).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED (third-party
package hassony2/multiperson `sdf` @ HEAD, not in /root/reference).
Reference call sites: homan/interactions/scenesdf.py:9,32,119.
Core: oracle/csrc/sdf.c (conventions stated there).
"""
import numpy as np
import torch

from . import clib


class SDF:
    def __init__(self, clamp_outside=False):
        # clamp_outside=True skips the distance evaluation of outside voxels (value 0);
        # identical after the reference's phi.clamp(0) (scenesdf.py:121).
        self.clamp_outside = clamp_outside

    def __call__(self, faces, vertices, grid_size=32):
        f = np.ascontiguousarray(faces.detach().cpu().numpy().astype(np.int32))
        v = np.ascontiguousarray(vertices.detach().cpu().numpy(), dtype=np.float32)
        B, V = v.shape[:2]
        phi = np.empty((B, grid_size, grid_size, grid_size), np.float32)
        clib.lib().orc_sdf_grid(clib.iptr(f), clib.fptr(v), B, V, f.shape[0], grid_size,
                                int(self.clamp_outside), clib.fptr(phi))
        return torch.from_numpy(phi)
