"""Seeded synthetic clips in the reference's input-dict schema.

There is no network, no dataset and no MANO licence in the build or bench
environment, so every test / benchmark input is generated here:
Core50-style pinhole camera (reference homan/datasets/core50.py:253-260),
procedural watertight object normalised like core50.py:34-38, MANO-shaped hand,
smooth ground-truth motion, targets rendered from the ground truth with
hand-occluded pixels set to -1 (reference homan/lib2d/maskutils.py:33-36), and an
initial state = perturbed ground truth.  The output dicts are what
``optimize_hand_object`` concatenates (reference homan/jointopt.py:55-91).
"""
import math

import numpy as np
import torch

from .mano_assets import synthetic_mano

BBOX_EXPANSION_FACTOR = 0.3  # reference homan/constants.py:33


def box_mesh(nx=5, ny=5, nz=10, scale=0.08):
    """Watertight subdivided box: 4(nx*ny+ny*nz+nx*nz) triangles (500 for 5,5,10)."""
    grid = {}
    verts, faces = [], []

    def vid(i, j, k):
        key = (i, j, k)
        if key not in grid:
            grid[key] = len(verts)
            verts.append([i / nx - 0.5, j / ny - 0.5, (k / nz - 0.5) * 2.0])
        return grid[key]

    def quad(a, b, c, d, flip):
        if flip:
            faces.extend([[a, c, b], [a, d, c]])
        else:
            faces.extend([[a, b, c], [a, c, d]])

    for k, flip in ((0, True), (nz, False)):
        for i in range(nx):
            for j in range(ny):
                quad(vid(i, j, k), vid(i + 1, j, k), vid(i + 1, j + 1, k), vid(i, j + 1, k), flip)
    for j, flip in ((0, False), (ny, True)):
        for i in range(nx):
            for k in range(nz):
                quad(vid(i, j, k), vid(i + 1, j, k), vid(i + 1, j, k + 1), vid(i, j, k + 1), flip)
    for i, flip in ((0, True), (nx, False)):
        for j in range(ny):
            for k in range(nz):
                quad(vid(i, j, k), vid(i, j + 1, k), vid(i, j + 1, k + 1), vid(i, j, k + 1), flip)
    v = np.asarray(verts, np.float64)
    return _normalise(v, scale), np.asarray(faces, np.int32)


def bottle_mesh(segments=50, rings=30, scale=0.2):
    """Watertight lathe 'bottle': 2*segments*rings triangles (3000 for 50x30)."""
    ts = np.linspace(0.0, 1.0, rings)
    # radius profile: body, shoulder, neck
    prof = np.where(ts < 0.55, 0.30, np.where(ts < 0.75, 0.30 - (ts - 0.55) / 0.20 * 0.19, 0.11))
    prof = prof * (1.0 + 0.04 * np.sin(ts * 9.0))
    ang = np.arange(segments) * (2 * math.pi / segments)
    verts = [[0.0, 0.0, -0.5]]
    for r, t in zip(prof, ts):
        for a in ang:
            verts.append([r * math.cos(a), r * math.sin(a), t - 0.5])
    verts.append([0.0, 0.0, 0.5])
    top = len(verts) - 1
    faces = []
    for s in range(segments):
        s1 = (s + 1) % segments
        faces.append([0, 1 + s1, 1 + s])
        for rr in range(rings - 1):
            a, b = 1 + rr * segments + s, 1 + rr * segments + s1
            c, d = a + segments, b + segments
            faces.extend([[a, b, d], [a, d, c]])
        base = 1 + (rings - 1) * segments
        faces.append([top, base + s, base + s1])
    return _normalise(np.asarray(verts, np.float64), scale), np.asarray(faces, np.int32)


def _normalise(v, scale):
    v = v - v.mean(0)
    v = v / np.linalg.norm(v, 2, 1).max() * scale / 2
    return v.astype(np.float32)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], np.float64)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float64)


def _proj_px(verts, K_px):
    """(B,V,3) @ pixel intrinsics -> (B,V,2) pixels."""
    hom = torch.einsum("bij,bvj->bvi", K_px, verts)
    return hom[..., :2] / hom[..., 2:]


def _square_roi(pts2d, expansion):
    """Per-frame square box (x0, y0, side) around 2-D points, expanded."""
    mn, mx = pts2d.min(1)[0], pts2d.max(1)[0]
    side = (mx - mn).max(1)[0] * (1.0 + expansion)
    ctr = (mn + mx) / 2
    return ctr[:, 0] - side / 2, ctr[:, 1] - side / 2, side


def _k_roi(K_px, x0, y0, side):
    """Crop-resize intrinsics, already divided by the ROI raster size (fx/b, (cx-x0)/b)."""
    K = K_px.clone()
    K[:, 0, 2] = K[:, 0, 2] - x0
    K[:, 1, 2] = K[:, 1, 2] - y0
    K[:, :2] = K[:, :2] / side[:, None, None]
    return K


def make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=None,
              hand_verts_fn=None, mano=None, pca_dim=45, hands=("right",)):
    """Build one synthetic clip.

    silhouette_fn(verts (B,V,3), faces (B,F,3) int32, K (B,3,3), size) -> (B,size,size)
        renderer used for the target masks (HIP product path on the GPU, the
        oracle in CPU tests; both give the same bits).
    hand_verts_fn(pca (B,P), rot (B,3), betas (B,10)[, side]) -> (B,778,3)
        MANO forward used for the ground truth (same remark).
    hands: sides of the hands in the clip, hand 0 first ("right" alone: every BASELINE config; ("right", "left"): the
        second hand holds the object from the other side; per-frame dicts then carry h entries per field, hands
        interleaved frame-major after collation like the reference expects, homan/homan.py:62-63).
    Returns dict(person_parameters, object_parameters, objvertices, objfaces,
                 camintr, image_size, gt=...).
    """
    assert silhouette_fn is not None and hand_verts_fn is not None
    mano = synthetic_mano(0) if mano is None else mano
    rng = np.random.default_rng(seed)
    torch_gen = torch.Generator().manual_seed(seed)
    B = frames
    if isinstance(obj, (tuple, list)):        # (vertices (V,3), faces (F,3)) of a mesh of the caller's
        ov, of = np.asarray(obj[0], np.float32), np.asarray(obj[1], np.int32)
    elif obj == "cube":
        ov, of = box_mesh()
    elif obj == "bottle":
        ov, of = bottle_mesh()
    else:
        raise ValueError(obj)
    ov_t = torch.from_numpy(ov)[None].repeat(B, 1, 1)
    of_t = torch.from_numpy(of)[None].repeat(B, 1, 1)

    f = 480.0 * image_size / 350.0
    c = 175.0 * image_size / 350.0
    K_px = torch.tensor([[f, 0, c], [0, f, c], [0, 0, 1.0]]).repeat(B, 1, 1)
    camintr_nc = K_px.clone()
    camintr_nc[:, :2] /= image_size

    t = np.arange(B)
    t_o = np.stack([0.02 * np.sin(2 * np.pi * t / B), 0.01 * np.cos(2 * np.pi * t / B),
                    np.full(B, 0.6)], 1)
    # object long axis (local z) roughly upright in the image, slowly turning
    R_o = np.stack([_rot_x(1.3) @ _rot_y(0.03 * ti) for ti in t])
    radius = float(np.abs(ov[:, :2]).max())
    t_h = t_o + np.array([-(radius + 0.075), 0.0, -0.02])
    R_h = np.stack([_rot_y(0.4 + 0.01 * ti) @ _rot_x(-0.3) for ti in t])
    pca_gt = np.zeros((B, pca_dim), np.float32)
    pca_gt[:, :16] = (rng.normal(size=16) * 0.3)[None] + rng.normal(size=(B, 16)) * 0.02
    rot_gt = (np.array([0.1, -0.2, 0.15]) + rng.normal(size=(B, 3)) * 0.01).astype(np.float32)
    mano_trans_gt = np.zeros((B, 3), np.float32)

    R_o_t = torch.from_numpy(R_o).float()
    R_h_t = torch.from_numpy(R_h).float()
    t_o_t = torch.from_numpy(t_o).float()[:, None]
    t_h_t = torch.from_numpy(t_h).float()[:, None]
    verts_obj_gt = torch.matmul(ov_t, R_o_t) + t_o_t
    side0 = () if hands[0] == "right" else (hands[0],)          # (renderers written before `side` existed keep working)
    hand_local_gt = hand_verts_fn(torch.from_numpy(pca_gt), torch.from_numpy(rot_gt),
                                  torch.zeros(B, 10), *side0).detach().cpu().float()
    verts_hand_gt = torch.matmul(hand_local_gt + torch.from_numpy(mano_trans_gt)[:, None], R_h_t) + t_h_t
    from .mano_assets import hand_models
    hand_faces = torch.from_numpy(hand_models(mano)[hands[0]]["faces"].astype(np.int32))
    hf_t = hand_faces[None].repeat(B, 1, 1)

    # ROIs and targets
    p2d_o = _proj_px(verts_obj_gt, K_px)
    p2d_h = _proj_px(verts_hand_gt, K_px)
    xo, yo, so = _square_roi(p2d_o, BBOX_EXPANSION_FACTOR)
    xh, yh, sh = _square_roi(p2d_h, BBOX_EXPANSION_FACTOR)
    K_roi_o = _k_roi(K_px, xo, yo, so)
    K_roi_h = _k_roi(K_px, xh, yh, sh)
    sil_o = silhouette_fn(verts_obj_gt, of_t, K_roi_o, rend_size).detach().cpu()
    sil_h_in_o = silhouette_fn(verts_hand_gt, hf_t, K_roi_o, rend_size).detach().cpu()
    sil_h = silhouette_fn(verts_hand_gt, hf_t, K_roi_h, rend_size).detach().cpu()
    sil_o_in_h = silhouette_fn(verts_obj_gt, of_t, K_roi_h, rend_size).detach().cpu()
    tm_o = (sil_o > 0.5).float()
    tm_o[sil_h_in_o > 0.5] = -1.0
    tm_h = (sil_h > 0.5).float()
    tm_h[(sil_o_in_h > 0.5) & (sil_h <= 0.5)] = -1.0

    # full-image instance masks (PointRend-like modal masks; only the ordinal depth term reads them): the hand sits
    # in front of the object in this scene, so it owns the overlap
    full_h = silhouette_fn(verts_hand_gt, hf_t, camintr_nc, image_size).detach().cpu() > 0.5
    full_o = (silhouette_fn(verts_obj_gt, of_t, camintr_nc, image_size).detach().cpu() > 0.5) & ~full_h

    verts2d = p2d_h + torch.randn(p2d_h.shape, generator=torch_gen)

    # perturbed initial state
    def noisy_rot(R):
        Rn = R.clone()
        Rn[:, :, :2] += torch.randn(R.shape[0], 3, 2, generator=torch_gen) * 0.05
        return Rn

    R_o_init, R_h_init = noisy_rot(R_o_t), noisy_rot(R_h_t)
    t_o_init = t_o_t + torch.randn(B, 1, 3, generator=torch_gen) * 0.005
    t_h_init = t_h_t + torch.randn(B, 1, 3, generator=torch_gen) * 0.005
    hand_local_init = hand_verts_fn(torch.zeros(B, pca_dim), torch.from_numpy(rot_gt),
                                    torch.zeros(B, 10), *side0).detach().cpu().float()
    verts_hand_init = torch.matmul(hand_local_init, R_h_init) + t_h_init

    # per-hand tracks: hand 0 above (its random draws are interleaved with the object's: clips with one hand are unchanged),
    # further hands from generators of their own
    tracks = [dict(side=hands[0], faces=hand_faces, t_init=t_h_init, R_init=R_h_init, mano_rot=torch.from_numpy(rot_gt),
                   tm=tm_h, full=full_h, verts_init=verts_hand_init, verts2d=verts2d, K_roi=K_roi_h, verts_gt=verts_hand_gt)]
    for hi, side in enumerate(hands[1:], start=1):
        rng_h = np.random.default_rng(seed + 7919 * hi)
        gen_h = torch.Generator().manual_seed(seed + 7919 * hi)
        sgn = -1.0 if hi % 2 else 1.0                     # the other side of the object
        t_hh = t_o + np.array([-sgn * (radius + 0.075), 0.0, -0.02 - 0.01 * hi])
        R_hh = np.stack([_rot_y(sgn * (0.4 + 0.01 * ti)) @ _rot_x(-0.3) for ti in t])
        pca_h = np.zeros((B, pca_dim), np.float32)
        pca_h[:, :16] = (rng_h.normal(size=16) * 0.3)[None] + rng_h.normal(size=(B, 16)) * 0.02
        rot_h = (np.array([0.1, 0.2 * sgn, 0.15]) + rng_h.normal(size=(B, 3)) * 0.01).astype(np.float32)
        R_hh_t, t_hh_t = torch.from_numpy(R_hh).float(), torch.from_numpy(t_hh).float()[:, None]
        local_gt = hand_verts_fn(torch.from_numpy(pca_h), torch.from_numpy(rot_h), torch.zeros(B, 10), side).detach().cpu().float()
        v_gt = torch.matmul(local_gt, R_hh_t) + t_hh_t
        faces_h = torch.from_numpy(hand_models(mano)[side]["faces"].astype(np.int32))
        hf_h = faces_h[None].repeat(B, 1, 1)
        p2d = _proj_px(v_gt, K_px)
        xh2, yh2, sh2 = _square_roi(p2d, BBOX_EXPANSION_FACTOR)
        K_roi2 = _k_roi(K_px, xh2, yh2, sh2)
        sil = silhouette_fn(v_gt, hf_h, K_roi2, rend_size).detach().cpu()
        sil_o_in = silhouette_fn(verts_obj_gt, of_t, K_roi2, rend_size).detach().cpu()
        tm = (sil > 0.5).float()
        tm[(sil_o_in > 0.5) & (sil <= 0.5)] = -1.0
        tm_o[silhouette_fn(v_gt, hf_h, K_roi_o, rend_size).detach().cpu() > 0.5] = -1.0       # it occludes the object too
        full = silhouette_fn(v_gt, hf_h, camintr_nc, image_size).detach().cpu() > 0.5
        full_o = full_o & ~full
        R_init = R_hh_t.clone()
        R_init[:, :, :2] += torch.randn(B, 3, 2, generator=gen_h) * 0.05
        t_init = t_hh_t + torch.randn(B, 1, 3, generator=gen_h) * 0.005
        local_init = hand_verts_fn(torch.zeros(B, pca_dim), torch.from_numpy(rot_h), torch.zeros(B, 10), side).detach().cpu().float()
        tracks.append(dict(side=side, faces=faces_h, t_init=t_init, R_init=R_init, mano_rot=torch.from_numpy(rot_h), tm=tm,
                           full=full, verts_init=torch.matmul(local_init, R_init) + t_init,
                           verts2d=p2d + torch.randn(p2d.shape, generator=gen_h), K_roi=K_roi2, verts_gt=v_gt))

    person_parameters, object_parameters = [], []
    for b in range(B):
        cat = lambda key: torch.cat([tr[key][b:b + 1] for tr in tracks])           # (h, ...)
        person_parameters.append(dict(
            translations=cat("t_init").clone(),                     # (h,1,3)
            rotations=cat("R_init").clone(),                        # (h,3,3)
            hand_side=[tr["side"] for tr in tracks],
            faces=torch.stack([tr["faces"] for tr in tracks]).clone(),      # (h,1538,3)
            mano_trans=torch.zeros(len(tracks), 3),
            mano_rot=cat("mano_rot").clone(),
            mano_betas=torch.zeros(len(tracks), 10),
            mano_pca_pose=torch.zeros(len(tracks), pca_dim),
            target_masks=cat("tm").clone(),                         # (h,S,S)
            masks=cat("full").float(),                              # (h,H,W) instance masks
            verts=cat("verts_init").clone(),                        # (h,778,3)
            verts2d=cat("verts2d").clone(),                         # (h,778,2) px
            K_roi=cat("K_roi").clone(),                             # (h,3,3)
            cams=torch.tensor([[1.0, 0.0, 0.0]]).repeat(len(tracks), 1),
        ))
        object_parameters.append(dict(
            translations=t_o_init[b:b + 1].clone(),                 # (1,1,3)
            rotations=R_o_init[b:b + 1].clone(),                    # (1,3,3)
            target_masks=tm_o[b:b + 1].clone(),                     # (1,S,S)
            K_roi=K_roi_o[b:b + 1, None].clone(),                   # (1,1,3,3)
            full_mask=full_o[b].float(),                            # (H,W) instance mask
        ))
    gt = dict(verts_object=verts_obj_gt, verts_hand=verts_hand_gt, pca=torch.from_numpy(pca_gt),
              rotations_object=R_o_t, translations_object=t_o_t, rotations_hand=R_h_t,
              translations_hand=t_h_t)
    return dict(person_parameters=person_parameters, object_parameters=object_parameters,
                objvertices=ov_t, objfaces=of_t, camintr=camintr_nc.numpy(), image_size=image_size,
                rend_size=rend_size, gt=gt)


# reference defaults, fit_vid_dataset.py:91-158,164-165 (step 1) and README.md:217,238 (step 2)
STEP1_LOSS_WEIGHTS = dict(lw_smooth_obj=2000.0, lw_smooth_hand=2000.0, lw_v2d_hand=50.0, lw_inter=1.0,
                          lw_contact=0.0, lw_depth=0.0, lw_pca=0.004, lw_sil_obj=1.0, lw_sil_hand=0.0,
                          lw_collision=0.0, lw_scale_obj=0.001, lw_scale_hand=0.001)
STEP2_LOSS_WEIGHTS = dict(STEP1_LOSS_WEIGHTS, lw_collision=0.001, lw_contact=1.0)
CFG1_LOSS_WEIGHTS = dict({k: 0.0 for k in STEP1_LOSS_WEIGHTS}, lw_sil_obj=1.0, lw_v2d_hand=50.0)


def hip_clip_fns(mano_model=None, device="cuda"):
    """(silhouette_fn, hand_verts_fn) for make_clip backed by the HIP product kernels."""
    from . import ops
    from .manomodel import ManoModel
    mano_model = synthetic_mano(0) if mano_model is None else mano_model
    mm = ManoModel(mano_model=mano_model, device=device)

    def hand_fn(pca, rot, betas, side="right"):
        with torch.no_grad():
            return mm.forward_pca(pca.to(device), rot=rot.to(device), betas=betas.to(device), side=side)["verts"].cpu()

    def sil_fn(verts, faces, K, size):
        with torch.no_grad():
            sctx = ops.SilhouetteContext(faces.to(device), verts.shape[1], verts.shape[0], size, device)
            return ops.silhouette_render(verts.to(device), K.to(device), sctx).cpu()

    return sil_fn, hand_fn


def weakcams_from_translations(translations, camintr, image_size=640):
    """Scaled-orthographic hand cameras [s, tx, ty] (the `cams` of person_parameters) under which `--hand_proj_mode ortho`
    places the hand at `translations` (B,1,3 or B,3): the inverse of homan_amd.homan.weakcam_persp_trans (reference
    homan/utils/camera.py:85-97).  Synthetic-input helper."""
    t = torch.as_tensor(translations, dtype=torch.float32).reshape(-1, 3)
    K = torch.as_tensor(camintr, dtype=torch.float32).reshape(-1, 3, 3)
    fx, fy = K[:, 0, 0] * image_size, K[:, 1, 1] * image_size
    cx, cy = K[:, 0, 2] * image_size, K[:, 1, 2] * image_size
    pscale = fx / t[:, 2]
    px, py = t[:, 0] * fx / t[:, 2] + cx, t[:, 1] * fy / t[:, 2] + cy
    s = 2.0 * pscale / image_size
    return torch.stack([s, px / pscale - 1.0 / s, py / pscale - 1.0 / s], 1)
