"""Ordinal-depth row: the oracle's depth image and its backward, checked on the CPU.

The depth backward restates the NMR kernel from its published formula; here it is checked against torch autograd
(float64) of the forward depth formula on the samples the hard rasteriser assigned to each face.
"""
import numpy as np
import torch

from oracle import model as o_model
from oracle import nmr


def _two_triangles():
    # NDC faces (B=1, NF=2, 3, 3): a large slanted triangle and a smaller one in front of part of it
    f = torch.tensor([[[[-0.8, -0.7, 0.9], [0.7, -0.6, 0.5], [-0.1, 0.8, 0.7]],
                       [[-0.3, -0.2, 0.45], [0.5, -0.3, 0.40], [0.1, 0.5, 0.42]]]], dtype=torch.float32)
    return f


def _depth_f64(faces, idx, size):
    """depth image recomputed in float64 from the (differentiable) faces, ownership fixed to `idx`."""
    B, NF = faces.shape[:2]
    ys, xs = torch.meshgrid(torch.arange(size, dtype=torch.float64), torch.arange(size, dtype=torch.float64),
                            indexing="ij")
    out = torch.full((B, size, size), 100.0, dtype=torch.float64)
    for b in range(B):
        for fn in range(NF):
            sel = idx[b] == fn
            if not sel.any():
                continue
            p = 0.5 * (faces[b, fn, :, :2] * size + size - 1)
            M = torch.cat([p, torch.ones(3, 1, dtype=torch.float64)], 1).T            # columns = vertices
            w = torch.linalg.solve(M, torch.stack([xs[sel], ys[sel], torch.ones_like(xs[sel])]))   # (3, n)
            zp = 1.0 / (w / faces[b, fn, :, 2:3]).sum(0)
            out[b][sel] = zp
    return out


def test_depth_backward_matches_autograd_of_forward_formula():
    size = 32
    faces = _two_triangles().requires_grad_(True)
    alpha, depth, idx = nmr._RasterizeAlphaDepth.apply(faces, size, 0.1, 100.0, 1e-3)
    g = torch.from_numpy(np.random.default_rng(0).normal(size=tuple(depth.shape)).astype(np.float32))
    (depth * g).sum().backward()
    got = faces.grad.clone()

    f64 = faces.detach().double().requires_grad_(True)
    d64 = _depth_f64(f64, idx, size)
    assert torch.allclose(d64.float(), depth.detach(), rtol=1e-5, atol=1e-6)
    (d64 * g.double()).sum().backward()
    want = f64.grad.float()
    # interior samples only enter (both triangles lie inside the image, clamps inactive away from edges): the
    # analytic backward equals the true derivative up to the edge samples whose barycentrics were clamped
    assert torch.allclose(got, want, rtol=2e-2, atol=2e-2 * want.abs().max()), (got, want)


def test_ordinal_depth_loss_counts_pairs_like_the_reference():
    B, S = 3, 8
    d0 = torch.full((B, S, S), 100.0)
    d1 = torch.full((B, S, S), 100.0)
    s0 = torch.zeros(B, S, S, dtype=torch.bool)
    s1 = torch.zeros(B, S, S, dtype=torch.bool)
    s0[:2, 2:6, 2:6] = True
    s1[1:, 4:8, 4:8] = True
    d0[s0] = 0.6
    d1[s1] = 0.5                                   # layer 1 in front where both render
    masks = torch.zeros(B, 2, S, S, dtype=torch.bool)
    masks[:, 0, 2:6, 2:6] = True                   # annotation: layer 0 owns the overlap
    out = o_model.compute_ordinal_depth_loss(masks, [s0, s1], [d0, d1])["loss_depth"]
    # pairs: (0,0): 2 frames, (1,1): 2 frames, (0,1) and (1,0): frame 1 only -> 6
    want = np.log1p(np.exp(0.1)) / 6.0
    assert abs(float(out) - want) < 1e-6
