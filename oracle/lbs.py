"""CPU restatement (pure torch, fp32) of the MANO layer the reference calls.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  PARITY UNPINNED: the layer is
the third-party package `mano` (hassony2/MANO @ HEAD, reference
requirements.txt:18), absent from /root/reference, and the licensed
MANO_RIGHT.pkl is absent too.  Reference call sites:
  mano.model.load(model_path=, num_pca_comps=, use_pca=, is_right=, model_type=,
                  batch_size=, flat_hand_mean=)            homan/manomodel.py:19-80
  layer(betas=, global_orient=, hand_pose=, transl=) -> 6-tuple, [0]=verts,
                  [1]=joints, [5]=full pose               homan/manomodel.py:119-123
  layer.hand_mean (45,), layer.hand_components (n_pca,45)  homan/manomodel.py:50-51,110-118
Published algorithm: SMPL/MANO linear blend skinning as in smplx.lbs
(Loper et al. 2015; Romero et al. 2017): shape blend shapes, joint regression,
Rodrigues (angle = |r + 1e-8|), pose blend shapes on (R[1:] - I), kinematic
chain over MANO parents, rest-pose removal, weighted skinning.
"""
import torch


def batch_rodrigues(rot_vecs):
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def lbs(betas, full_pose, model):
    """betas (B,10), full_pose (B,48) axis-angle -> verts (B,778,3), joints (B,16,3)."""
    B = full_pose.shape[0]
    v_template, shapedirs, posedirs = model["v_template"], model["shapedirs"], model["posedirs"]
    J_regressor, weights, parents = model["J_regressor"], model["lbs_weights"], model["parents"]
    v_shaped = v_template[None] + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    rot_mats = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
    nj = rot_mats.shape[1]
    pose_feature = (rot_mats[:, 1:] - torch.eye(3)[None, None]).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, posedirs).view(B, -1, 3)
    # kinematic chain
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:].long()]
    tm = torch.cat([torch.cat([rot_mats, rel[..., None]], -1),
                    torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(B, nj, 1, 4)], -2)   # (B,16,4,4)
    chain = [tm[:, 0]]
    for i in range(1, nj):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    G = torch.stack(chain, 1)
    posed_joints = G[:, :, :3, 3]
    Jh = torch.cat([J, torch.zeros(B, nj, 1)], -1)[..., None]                     # (B,16,4,1)
    corr = torch.matmul(G, Jh)                                                     # (B,16,4,1)
    A = G - torch.cat([torch.zeros(B, nj, 4, 3), corr], -1)
    T = torch.matmul(weights[None].expand(B, -1, -1), A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1)], -1)[..., None]
    verts = torch.matmul(T, vh)[:, :, :3, 0]
    return verts, posed_joints


class _WrittenOutVerts(torch.autograd.Function):
    """Hand vertices (model space, before the translation) from (pca, rot, betas) in the written-out evaluation order of
    oracle/csrc/lbs_exact.c - bit-equal with csrc/mano.hip - with the gradients of the torch restatement above at the same
    inputs (same mathematics: the two forwards agree within fp32 rounding, tests/test_objchain.py)."""

    @staticmethod
    def forward(ctx, pca, rot, betas, layout, faithful_fn):
        import numpy as np
        from . import clib
        B = pca.shape[0]
        p = np.ascontiguousarray(pca.detach().numpy(), np.float32)
        r = np.ascontiguousarray(rot.detach().numpy(), np.float32)
        be = np.ascontiguousarray(betas.detach().numpy(), np.float32)
        out = np.empty((B, 778, 3), np.float32)
        vt, M, Jt, Js, w, comps, mean, parents = layout
        clib.lib().orc_mano_forward(clib.fptr(vt), clib.fptr(M), clib.fptr(Jt), clib.fptr(Js), clib.fptr(w), clib.fptr(comps),
                                    clib.fptr(mean), clib.iptr(parents), clib.fptr(p), p.shape[1], clib.fptr(r), clib.fptr(be),
                                    B, clib.fptr(out))
        ctx.save_for_backward(pca, rot, betas)
        ctx.faithful_fn = faithful_fn
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        pca, rot, betas = ctx.saved_tensors
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) for t in (pca, rot, betas)]
            v = ctx.faithful_fn(*ins)
            grads = torch.autograd.grad(v, ins, g, allow_unused=True)
        return grads[0], grads[1], grads[2], None, None


def written_out_verts(pca, rot, betas, layout, faithful_fn):
    return _WrittenOutVerts.apply(pca, rot, betas, layout, faithful_fn)


class ManoLayer(torch.nn.Module):
    """What `mano.model.load(...)` returns, as far as the reference uses it."""

    def __init__(self, model_np, num_pca_comps=16, flat_hand_mean=True, use_pca=False):
        super().__init__()
        self.model = {k: torch.as_tensor(v) for k, v in model_np.items()
                      if hasattr(v, "shape") and k not in ("faces", "closed_faces")}
        self.model["parents"] = torch.as_tensor(model_np["parents"]).long()
        self.use_pca = use_pca
        self.hand_components = torch.as_tensor(model_np["hand_components"][:num_pca_comps])
        hm = torch.as_tensor(model_np["hand_mean"])
        self.hand_mean = torch.zeros_like(hm) if flat_hand_mean else hm
        self.faces = torch.as_tensor(model_np["faces"].astype("int64"))

    def forward(self, betas=None, global_orient=None, hand_pose=None, transl=None):
        if self.use_pca:
            hand_pose = torch.einsum("bi,ij->bj", hand_pose, self.hand_components)
        full_pose = torch.cat([global_orient, hand_pose + self.hand_mean[None]], dim=1)
        verts, joints = lbs(betas, full_pose, self.model)
        if transl is not None:
            verts = verts + transl[:, None]
            joints = joints + transl[:, None]
        return verts, joints, None, None, global_orient, full_pose
