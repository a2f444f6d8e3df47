"""HIP silhouette rasteriser vs the CPU oracle (GPU box)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(B=3, S=64, obj="bottle", seed=0):
    from homan_amd import synth
    g = torch.Generator().manual_seed(seed)
    ov, of = synth.bottle_mesh() if obj == "bottle" else synth.box_mesh()
    V = ov.shape[0]
    verts = torch.from_numpy(ov)[None].repeat(B, 1, 1)
    ang = torch.rand(B, generator=g) * 6.28
    R = torch.stack([torch.tensor(synth._rot_x(1.1 + 0.1 * i) @ synth._rot_y(float(a)), dtype=torch.float32)
                     for i, a in enumerate(ang)])
    t = torch.tensor([[0.0, 0.0, 0.6]]) + torch.randn(B, 3, generator=g) * 0.01
    verts = verts @ R + t[:, None]
    # ROI intrinsics so the object fills ~60% of the raster
    K = torch.tensor([[2.6, 0, 0.5], [0, 2.6, 0.5], [0, 0, 1.0]]).repeat(B, 1, 1)
    faces = torch.from_numpy(of)[None].repeat(B, 1, 1)
    return verts, faces, K, V


def _oracle_idx(faces9, S):
    """faces9 (B,F,9) NDC -> fill_back doubling -> oracle face-index map (B,2S,2S)."""
    from oracle import clib
    B, F = faces9.shape[:2]
    f = faces9.reshape(B, F, 3, 3)
    both = np.ascontiguousarray(np.concatenate([f, f[:, :, ::-1]], 1).reshape(B, 2 * F, 9), np.float32)
    idx = np.empty((B, 2 * S, 2 * S), np.int32)
    dep = np.empty((B, 2 * S, 2 * S), np.float32)
    clib.lib().orc_nmr_face_index_map(clib.fptr(both), B, 2 * F, 2 * S, 0.1, 100.0, clib.iptr(idx), clib.fptr(dep))
    return both, idx


@pytest.mark.parametrize("S,obj", [(64, "bottle"), (32, "cube"), (128, "bottle"), (512, "cube")])
def test_face_index_map_bit_exact(S, obj):
    from homan_amd import ops
    verts, faces, K, V = _scene(B=3, S=S, obj=obj)
    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), V, 3, S, dev)
    img = ops.silhouette_render(verts.to(dev), K.to(dev), sctx)
    torch.cuda.synchronize()
    faces9 = sctx.faces9().cpu().numpy()
    _, idx_ref = _oracle_idx(faces9, S)
    idx = sctx.idx_map().cpu().numpy()
    assert (idx >= 0).sum() > 100
    np.testing.assert_array_equal(idx, idx_ref)
    # flip + 2x2 pool of the oracle map == HIP silhouettes, exactly (quarter steps)
    alpha = torch.from_numpy((idx_ref >= 0).astype(np.float32)).flip(1)
    pooled = torch.nn.functional.avg_pool2d(alpha[:, None], 2)[:, 0]
    np.testing.assert_array_equal(img.cpu().numpy(), pooled.numpy())


def test_projection_matches_oracle():
    from homan_amd import ops
    from oracle import nmr
    verts, faces, K, V = _scene(B=2, S=64)
    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), V, 2, 64, dev)
    ops.silhouette_render(verts.to(dev), K.to(dev), sctx)
    f9 = sctx.faces9().cpu()
    ndc = nmr.projection(verts, K, torch.eye(3)[None], torch.zeros(1, 3), torch.zeros(1, 5), 1)
    ref = nmr.vertices_to_faces(ndc, faces).reshape(2, -1, 9)
    # same operations in the same order on both sides (oracle/nmr.py projection, csrc/raster_setup.hip project_vertex): bit-equal
    np.testing.assert_array_equal(f9.numpy(), ref.numpy())


@pytest.mark.parametrize("S,obj", [(64, "bottle"), (32, "cube"), (256, "bottle"), (512, "cube")])     # 512: the largest size
def test_pseudo_gradient_matches_oracle(S, obj):
    """Same NDC faces + same upstream image gradient -> same per-vertex NDC gradient (sum order differs)."""
    from homan_amd import ops
    from oracle import clib
    B = 3
    verts, faces, K, V = _scene(B=B, S=S, obj=obj, seed=1)
    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), V, B, S, dev)
    sctx.grad_ndc = torch.zeros(B, V, 3, device=dev)
    v = verts.to(dev).requires_grad_(True)
    img = ops.silhouette_render(v, K.to(dev), sctx)
    g = torch.Generator().manual_seed(5)
    gimg = torch.randn(B, S, S, generator=g)
    img.backward(gimg.to(dev))
    torch.cuda.synchronize()
    faces9 = sctx.faces9().cpu().numpy()
    both, idx_ref = _oracle_idx(faces9, S)
    F = faces9.shape[1]
    ga = (gimg / 4).repeat_interleave(2, 1).repeat_interleave(2, 2).flip(1).contiguous().numpy()
    gf = np.zeros((B, 2 * F, 9), np.float32)
    clib.lib().orc_nmr_grad_faces_alpha(clib.fptr(both), clib.iptr(idx_ref), clib.fptr(ga), B, 2 * F, 2 * S, 1e-3,
                                        clib.fptr(gf))
    gf = torch.from_numpy(gf).view(B, 2 * F, 3, 3)
    f2 = torch.cat([faces, faces.flip(2)], 1).long()
    ref = torch.zeros(B, V, 3)
    for b in range(B):
        ref[b].index_add_(0, f2[b].reshape(-1), gf[b].reshape(-1, 3))
    got = sctx.grad_ndc.cpu()
    assert ref.abs().max() > 0
    scale = ref.abs().max()
    np.testing.assert_allclose((got / scale).numpy(), (ref / scale).numpy(), atol=2e-5)
    assert torch.isfinite(v.grad).all()


@pytest.mark.parametrize("weight", [3.0, -0.7])
def test_fused_loss_matches_oracle_end_to_end(weight):
    """verts -> loss_sil, IoU and d loss / d verts against the oracle renderer + torch autograd.  The NMR pseudo-gradient
    is not linear in the upstream gradient (it selects samples by its sign): a negative weight takes the path that
    rebuilds the sweep planes in the backward instead of reusing the forward's."""
    from homan_amd import ops
    from oracle import nmr, yana
    B, S = 4, 64
    verts, faces, K, V = _scene(B=B, S=S, seed=3)
    # targets: silhouettes of a slightly shifted object, with an ignore band
    r = nmr.Renderer(image_size=S, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    target = r(verts + torch.tensor([0.006, -0.004, 0.0]), faces, mode="silhouettes")
    tm = (target > 0.5).float()
    tm[:, :, :10] = -1
    ref_mask, keep = (tm > 0).float(), (tm >= 0).float()
    vo = verts.clone().requires_grad_(True)
    rend = r(vo, faces, K=K, mode="silhouettes")
    image = keep * rend
    loss_o = (torch.sum((image - ref_mask) ** 2) / keep.sum()) / B
    iou_o = yana.batch_mask_iou(image, ref_mask).mean()
    (loss_o * weight).backward()

    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), V, B, S, dev)
    vh = verts.to(dev).requires_grad_(True)
    loss_h, iou_h, img_h = ops.silhouette_loss(vh, K.to(dev), keep.to(dev), ref_mask.to(dev),
                                               keep.sum().reshape(1).to(dev), sctx)
    (loss_h * weight).sum().backward()
    mism = (img_h.cpu() != rend.detach()).float().mean().item()
    assert mism < 1e-4, mism            # projection rounding may flip a sample or two
    np.testing.assert_allclose(loss_h.item(), loss_o.item(), rtol=2e-4)
    np.testing.assert_allclose(iou_h.item(), iou_o.item(), rtol=2e-4)
    scale = vo.grad.abs().max()
    err = ((vh.grad.cpu() - vo.grad) / scale).abs()
    assert err.max() < 2e-2 and err.mean() < 1e-4, (err.max(), err.mean())


def test_persistent_outputs_skip_is_invisible():
    """hm_sil_fwd(persistent_outputs=1) leaves background regions untouched when they were background in the previous
    call: a sequence in which the object wanders across the raster (regions turn empty, covered and empty again) must give
    the same silhouettes, loss, index map and fused-loss image as fresh full-write calls."""
    from homan_amd import lib as hlib
    from homan_amd import ops
    dev = torch.device("cuda")
    B, S = 3, 64
    verts, faces, K, V = _scene(B=B, S=S, obj="cube", seed=5)
    K = K.clone()
    K[:, 0, 0] = K[:, 1, 1] = 1.2                          # small object: most regions are background
    keep = torch.ones(B, S, S)
    ref = torch.zeros(B, S, S)
    ref[:, 20:44, 20:44] = 1.0
    keep_sum = keep.sum().reshape(1)
    L, P = hlib.lib(), hlib.ptr
    F = faces.shape[1]

    def call(sctx, v, pooled, out, persistent):
        hlib.check(L.hm_sil_fwd(P(v), P(sctx.faces), 0, P(Kd), B, V, F, S, 1.0, 0.1, 100.0, P(keepd), P(refd), P(ksd),
                                P(pooled), P(out), P(sctx.work_order), None, None, 0, None, None, None, 0, persistent,
                                P(sctx.workspace), hlib.stream()), "hm_sil_fwd")

    Kd, keepd, refd, ksd = K.to(dev), keep.to(dev), ref.to(dev), keep_sum.to(dev)
    sp = ops.SilhouetteContext(faces.to(dev), V, B, S, dev)             # persistent sequence
    pooled_p, out_p = torch.full((B, S, S), 7.0, device=dev), torch.zeros(2, device=dev)
    shifts = [(0.0, 0.0), (0.12, 0.0), (0.12, 0.1), (-0.1, 0.1), (0.0, 0.0), (0.0, 0.0), (0.3, 0.3), (0.0, -0.12)]
    for step, (dx, dy) in enumerate(shifts):
        v = (verts + torch.tensor([dx, dy, 0.0])).to(dev).contiguous()
        call(sp, v, pooled_p, out_p, 1)
        sf = ops.SilhouetteContext(faces.to(dev), V, B, S, dev)         # fresh, full-write reference
        pooled_f, out_f = torch.empty(B, S, S, device=dev), torch.zeros(2, device=dev)
        call(sf, v, pooled_f, out_f, 0)
        assert torch.equal(pooled_p, pooled_f), step
        assert torch.equal(out_p, out_f), step
        assert torch.equal(sp.idx_map(), sf.idx_map()), step
        # the backward consumes dimg and the sweep planes from the workspace: same gradients
        gp, gf = torch.empty(B, V, 3, device=dev), torch.empty(B, V, 3, device=dev)
        one = torch.ones(1, device=dev)
        for sc, g in ((sp, gp), (sf, gf)):
            hlib.check(L.hm_sil_bwd(P(v), P(Kd), B, V, F, S, 1.0, 1e-3, 1, P(one), None, P(ksd), P(sc.adj_off),
                                    P(sc.adj_items), None, P(g), None, P(sc.workspace), 0, hlib.stream()), "hm_sil_bwd")
        assert torch.equal(gp, gf), step


@pytest.mark.parametrize("obj", ["cube", "bottle"])
def test_no_antialiasing_render_and_gradient_match_oracle(obj):
    """nr.Renderer(anti_aliasing=False) (reference homan/pose_optimization.py:89-96): coverage image bit-exact vs the
    oracle renderer, pseudo-gradient of a generic image loss within the summation-order tolerance."""
    from homan_amd import ops
    from oracle import nmr
    B, S = 3, 64
    verts, faces, K, V = _scene(B=B, S=S, obj=obj, seed=9)
    r = nmr.Renderer(image_size=S, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1, anti_aliasing=False)
    vo = verts.clone().requires_grad_(True)
    img_o = r(vo, faces, mode="silhouettes")
    assert img_o.shape == (B, S, S)
    gen = torch.Generator().manual_seed(1)
    target = (torch.rand(B, S, S, generator=gen) > 0.5).float()
    w = torch.rand(B, S, S, generator=gen)
    ((w * (img_o - target) ** 2).sum()).backward()

    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), V, B, S // 2, dev)
    vh = verts.to(dev).requires_grad_(True)
    img_h = ops.silhouette_render_noaa(vh, K.to(dev), sctx)
    mism = (img_h.detach().cpu() != img_o.detach()).float().mean().item()
    assert mism < 1e-4, mism            # projection rounding may flip a sample or two
    ((w.to(dev) * (img_h - target.to(dev)) ** 2).sum()).backward()
    scale = vo.grad.abs().max()
    err = ((vh.grad.cpu() - vo.grad) / scale).abs()
    assert err.max() < 2e-2 and err.mean() < 1e-4, (err.max(), err.mean())


@pytest.mark.parametrize("obj", ["bottle", "cube"])
def test_sweep_scheduling_and_capacity_paths_do_not_change_the_gradient(obj):
    """The edge sweeps' results must not depend on scheduling: the number of persistent workgroups (hm_tune_sweep_blocks)
    leaves the gradient bit-identical, and with the capacity tables of the work list shrunk to a handful of entries
    (hm_debug_sweep_caps: binary search for a unit's first face, faces accumulated with float atomics) it still matches
    to summation order."""
    from homan_amd import lib as hlib
    from homan_amd import ops
    L = hlib.lib()
    B, S = 3, 64
    verts, faces, K, V = _scene(B=B, S=S, obj=obj, seed=2)
    dev = torch.device("cuda")
    gimg = torch.randn(B, S, S, generator=torch.Generator().manual_seed(7)).to(dev)

    def grad(blocks=0, cap=0):
        prev_b, prev_c = L.hm_tune_sweep_blocks(blocks), L.hm_debug_sweep_caps(cap)
        try:
            sctx = ops.SilhouetteContext(faces.to(dev), V, B, S, dev)
            v = verts.to(dev).requires_grad_(True)
            ops.silhouette_render(v, K.to(dev), sctx).backward(gimg)
            torch.cuda.synchronize()
            return v.grad.clone()
        finally:
            L.hm_tune_sweep_blocks(prev_b)
            L.hm_debug_sweep_caps(prev_c)

    ref = grad()
    assert ref.abs().max() > 0
    for blocks in (8, 64, 768):
        assert torch.equal(grad(blocks=blocks), ref), blocks
    scale = ref.abs().max()
    for cap in (1, 3, 40):
        np.testing.assert_allclose((grad(cap=cap) / scale).cpu().numpy(), (ref / scale).cpu().numpy(), atol=2e-6,
                                   err_msg=f"cap {cap}")


def test_large_batches_take_the_four_faces_per_thread_work_list():
    """B * F >= 400k switches the sweep work list to four faces per compaction thread (500 candidate poses in the pose
    initialisation).  804 frames = 268 copies of three poses: every copy must get the gradient of the 3-frame batch (to
    summation order - the units are composed differently)."""
    from homan_amd import ops
    S = 32
    verts, faces, K, V = _scene(B=3, S=S, obj="cube", seed=4)
    assert 804 * faces.shape[1] >= 400000
    dev = torch.device("cuda")
    gimg = torch.randn(3, S, S, generator=torch.Generator().manual_seed(9))

    def grad(rep):
        B = 3 * rep
        sctx = ops.SilhouetteContext(faces.repeat(rep, 1, 1).to(dev), V, B, S, dev)
        v = verts.repeat(rep, 1, 1).to(dev).requires_grad_(True)
        ops.silhouette_render(v, K.repeat(rep, 1, 1).to(dev), sctx).backward(gimg.repeat(rep, 1, 1).to(dev))
        torch.cuda.synchronize()
        return v.grad.cpu()

    small, large = grad(1), grad(268)
    scale = small.abs().max()
    assert scale > 0
    np.testing.assert_allclose((large.reshape(268, 3, V, 3) / scale).numpy(),
                               (small[None].expand(268, -1, -1, -1) / scale).numpy(), atol=2e-6)


def test_nothing_visible_gives_empty_images_and_zero_gradients():
    """Edge case: the whole mesh off-screen (empty work list, zero units) or behind the near plane."""
    from homan_amd import ops
    S = 32
    verts, faces, K, V = _scene(B=2, S=S, obj="cube", seed=6)
    dev = torch.device("cuda")
    for shift in (torch.tensor([5.0, 0.0, 0.0]), torch.tensor([0.0, 0.0, -2.0])):
        sctx = ops.SilhouetteContext(faces.to(dev), V, 2, S, dev)
        v = (verts + shift).to(dev).requires_grad_(True)
        img = ops.silhouette_render(v, K.to(dev), sctx)
        assert float(img.detach().abs().sum()) == 0.0
        img.backward(torch.ones_like(img))
        torch.cuda.synchronize()
        assert torch.isfinite(v.grad).all() and float(v.grad.abs().sum()) == 0.0


def test_64_bit_offset_instantiations_give_the_same_results():
    """The line expansion and the sweeps exist twice: with 32-bit byte offsets off scalar bases (taken while the source arrays
    stay below 4 GB: every test and benchmark size) and with 64-bit element indices (beyond).  HOMAN_FORCE_W64=1 - read when the
    library first launches them, hence a fresh process - takes the 64-bit instantiations: the bit-exactness tests of the index
    map's consumers (pseudo-gradient against the oracle, fused loss, the no-anti-aliasing modes) must pass through them too."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, HOMAN_FORCE_W64="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_raster_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-k", "pseudo_gradient or fused_loss or no_antialiasing or scheduling"], env=env, capture_output=True,
                       text=True, cwd=root, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def _multi_scene(B, S, seed=3):
    """three renders of one scene: object at ROI cameras with keep / ref masks and the rigid transform in the face setup, the same
    object at a full-image camera with a depth output, a second mesh (cube: other V, F) with a depth output"""
    from homan_amd import synth
    g = torch.Generator().manual_seed(seed)
    dev = torch.device("cuda")
    ov, of = synth.bottle_mesh()
    cv, cf = synth.box_mesh()
    mesh_o = torch.from_numpy(ov)[None].repeat(B, 1, 1).float().contiguous()
    rot = torch.randn(B, 3, 2, generator=g)
    trans = torch.tensor([[0.0, 0.0, 0.6]]) + torch.randn(B, 3, generator=g) * 0.01
    scale = torch.tensor([1.1])
    K_roi = torch.tensor([[2.6, 0, 0.5], [0, 2.6, 0.5], [0, 0, 1.0]]).repeat(B, 1, 1)
    K_full = torch.tensor([[1.3, 0, 0.45], [0, 1.3, 0.55], [0, 0, 1.0]]).repeat(B, 1, 1)
    keep = (torch.rand(B, S, S, generator=g) > 0.2).float()
    ref = (torch.rand(B, S, S, generator=g) > 0.5).float() * keep
    verts_c = torch.from_numpy(cv)[None].repeat(B, 1, 1).float() + torch.tensor([[0.02, -0.01, 0.55]]) \
        + torch.randn(B, 1, 3, generator=g) * 0.01
    d = dict(mesh_o=mesh_o, rot=rot, trans=trans, scale=scale, K_roi=K_roi, K_full=K_full, keep=keep, ref=ref, verts_c=verts_c,
             faces_o=torch.from_numpy(of).int(), faces_c=torch.from_numpy(cf).int())
    return {k: v.to(dev).contiguous() for k, v in d.items()}


@pytest.mark.parametrize("B,S", [(3, 64), (5, 128)])
def test_multi_render_launch_equals_separate_calls(B, S):
    """hm_sil_fwd_multi: three renders (two meshes, three cameras, keep / ref on one, depth outputs on two) as ONE setup + ONE raster
    launch = three hm_sil_fwd_clips calls, bit for bit - outputs, index maps, packed faces, camera-space vertices - and each
    workspace then feeds its own backward (silhouette sweeps / depth-map backward) to the same gradients."""
    from homan_amd import lib as hlib
    from homan_amd import ops
    sc = _multi_scene(B, S)
    dev = sc["mesh_o"].device
    L, P, ck = hlib.lib(), hlib.ptr, hlib.check
    Vo, Fo, Vc, Fc = sc["mesh_o"].shape[1], sc["faces_o"].shape[0], sc["verts_c"].shape[1], sc["faces_c"].shape[0]
    stream = hlib.stream()

    def contexts():
        return (ops.SilhouetteContext(sc["faces_o"][None].expand(B, -1, -1), Vo, B, S, dev),
                ops.SilhouetteContext(sc["faces_o"][None].expand(B, -1, -1), Vo, B, S, dev),
                ops.SilhouetteContext(sc["faces_c"][None].expand(B, -1, -1), Vc, B, S, dev))

    def outputs():
        return [torch.full((B, S, S), -7.0, device=dev) for _ in range(5)] + [torch.full((B, Vo, 3), -7.0, device=dev)]

    def renders(ctxs, out):
        p_sil, p_do, d_do, p_dc, d_dc, vo = out
        rigid = dict(rigid_rot6d=sc["rot"], rigid_trans=sc["trans"], rigid_scale=sc["scale"], rigid_abs=1)
        common = dict(S=S, B=B, orig_size=1.0, znear=ops.NMR_NEAR, zfar=ops.NMR_FAR, clip_len=B)
        return [dict(verts=sc["mesh_o"], faces=sc["faces_o"], K=sc["K_roi"], keep=sc["keep"], ref=sc["ref"], pooled=p_sil,
                     work_order=ctxs[0].work_order, cam_verts_out=vo, workspace=ctxs[0].workspace, V=Vo, F=Fo, **rigid, **common),
                dict(verts=sc["mesh_o"], faces=sc["faces_o"], K=sc["K_full"], pooled=p_do, pooled_depth=d_do,
                     work_order=ctxs[1].work_order, workspace=ctxs[1].workspace, V=Vo, F=Fo, **rigid, **common),
                dict(verts=sc["verts_c"], faces=sc["faces_c"], K=sc["K_full"], pooled=p_dc, pooled_depth=d_dc,
                     work_order=ctxs[2].work_order, workspace=ctxs[2].workspace, V=Vc, F=Fc, **common)]

    # ---- separate calls
    c_a, o_a = contexts(), outputs()
    for r in renders(c_a, o_a):
        ck(L.hm_sil_fwd_clips(P(r["verts"]), P(r["faces"]), 0, P(r["K"]), B, r["V"], r["F"], S, 1.0, ops.NMR_NEAR, ops.NMR_FAR,
                              P(r.get("keep")), P(r.get("ref")), None, P(r["pooled"]), None, P(r["work_order"]),
                              P(r.get("pooled_depth")), None, 0, P(r.get("rigid_rot6d")), P(r.get("rigid_trans")),
                              P(r.get("rigid_scale")), r.get("rigid_abs", 0), 0, P(r["workspace"]), B, 0,
                              P(r.get("cam_verts_out")), stream), "hm_sil_fwd_clips")
    # ---- one launch pair
    c_b, o_b = contexts(), outputs()
    arr = hlib.sil_renders(renders(c_b, o_b))
    ck(L.hm_sil_fwd_multi(arr, 3, 3, stream), "hm_sil_fwd_multi")
    torch.cuda.synchronize()
    for a, b in zip(o_a, o_b):
        assert torch.equal(a, b)
    assert float(o_b[0].sum()) > 10 and float((o_b[2] < ops.NMR_FAR).sum()) > 10 and float((o_b[4] < ops.NMR_FAR).sum()) > 10
    for ca, cb in zip(c_a, c_b):
        assert torch.equal(ca.idx_map(), cb.idx_map())
        assert torch.equal(ca.faces9(), cb.faces9())
    # ---- the backward passes run on the workspaces the multi launch filled
    up = torch.ones(1, device=dev)
    keep_sum = sc["keep"].sum().reshape(1)
    g_depth = torch.randn(B, S, S, generator=torch.Generator().manual_seed(5)).to(dev)
    grads = []
    for ctxs, out in ((c_a, o_a), (c_b, o_b)):
        gv = torch.empty(B, Vo, 3, device=dev)
        ck(L.hm_sil_bwd_clips(P(out[5]), P(sc["K_roi"]), B, Vo, Fo, S, 1.0, ops.NMR_EPS, 2, P(up), None, P(keep_sum),
                              P(ctxs[0].adj_off), P(ctxs[0].adj_items), None, P(gv), None, P(ctxs[0].workspace), B, None, 0, 0,
                              stream), "hm_sil_bwd_clips")
        gd_o, gd_c = torch.empty(B, Vo, 3, device=dev), torch.empty(B, Vc, 3, device=dev)
        ck(L.hm_depth_bwd(P(out[5]), P(sc["K_full"]), B, Vo, Fo, S, 1.0, P(g_depth), P(ctxs[1].adj_off), P(ctxs[1].adj_items),
                          P(gd_o), P(ctxs[1].workspace), stream), "hm_depth_bwd")
        ck(L.hm_depth_bwd(P(sc["verts_c"]), P(sc["K_full"]), B, Vc, Fc, S, 1.0, P(g_depth), P(ctxs[2].adj_off),
                          P(ctxs[2].adj_items), P(gd_c), P(ctxs[2].workspace), stream), "hm_depth_bwd")
        grads.append((gv, gd_o, gd_c))
    torch.cuda.synchronize()
    for a, b in zip(*grads):
        assert torch.equal(a, b)
        assert float(a.abs().sum()) > 0
    # a second multi launch on the same workspaces (persistent outputs, bins re-armed by the first) gives the same images
    for r in arr:
        r.persistent_outputs = 1
    ck(L.hm_sil_fwd_multi(arr, 3, 3, stream), "hm_sil_fwd_multi")
    ck(L.hm_sil_fwd_multi(arr, 3, 1, stream), "hm_sil_fwd_multi setup")
    ck(L.hm_sil_fwd_multi(arr, 3, 2, stream), "hm_sil_fwd_multi raster")
    torch.cuda.synchronize()
    for a, b in zip(o_a, o_b):
        assert torch.equal(a, b)


def test_multi_render_rejects_bad_arguments():
    from homan_amd import lib as hlib
    L = hlib.lib()
    arr = (hlib.SilRender * 1)()
    assert L.hm_sil_fwd_multi(arr, 1, 3, hlib.stream()) == -1          # NULL everything
    assert L.hm_sil_fwd_multi(arr, 0, 3, hlib.stream()) == -1
    assert L.hm_sil_fwd_multi(arr, 5, 3, hlib.stream()) == -1


@pytest.mark.parametrize("n", [1, 4])
def test_multi_render_launch_one_and_four_renders(n):
    """hm_sil_fwd_multi at its limits: ONE render (= hm_sil_fwd_clips) and FOUR (the table's capacity), different sizes S per
    render included - every render's outputs equal its own single call."""
    from homan_amd import lib as hlib
    from homan_amd import ops
    B = 2
    sc = _multi_scene(B, 64, seed=9)
    dev = sc["mesh_o"].device
    L, P, ck = hlib.lib(), hlib.ptr, hlib.check
    stream = hlib.stream()
    specs = [("verts_c", "faces_c", "K_full", 64), ("mesh_o", "faces_o", "K_roi", 32), ("verts_c", "faces_c", "K_roi", 96),
             ("mesh_o", "faces_o", "K_full", 64)][:n]
    # (mesh_o is mesh-space: rendered as it lies, 0.6 m in front of the camera through the translation below)
    verts = {"verts_c": sc["verts_c"], "mesh_o": (sc["mesh_o"] + torch.tensor([0.0, 0.0, 0.6], device=dev)).contiguous()}
    single, multi, rend = [], [], []
    for vk, fk, kk, S in specs:
        V, F = verts[vk].shape[1], sc[fk].shape[0]
        for store in (single, multi):
            ctx = ops.SilhouetteContext(sc[fk][None].expand(B, -1, -1), V, B, S, dev)
            store.append((ctx, torch.full((B, S, S), -3.0, device=dev), torch.full((B, S, S), -3.0, device=dev)))
        ctx, p, d = single[-1]
        ck(L.hm_sil_fwd_clips(P(verts[vk]), P(sc[fk]), 0, P(sc[kk]), B, V, F, S, 1.0, ops.NMR_NEAR, ops.NMR_FAR, None, None, None, P(p),
                              None, P(ctx.work_order), P(d), None, 0, None, None, None, 0, 0, P(ctx.workspace), B, 0, None, stream),
           "hm_sil_fwd_clips")
        ctx, p, d = multi[-1]
        rend.append(dict(verts=verts[vk], faces=sc[fk], K=sc[kk], pooled=p, pooled_depth=d, work_order=ctx.work_order,
                         workspace=ctx.workspace, B=B, V=V, F=F, S=S, orig_size=1.0, znear=ops.NMR_NEAR, zfar=ops.NMR_FAR, clip_len=B))
    ck(L.hm_sil_fwd_multi(hlib.sil_renders(rend), n, 3, stream), "hm_sil_fwd_multi")
    torch.cuda.synchronize()
    for (ca, pa, da), (cb, pb, db) in zip(single, multi):
        assert torch.equal(pa, pb) and torch.equal(da, db) and torch.equal(ca.idx_map(), cb.idx_map())
        assert float(pa.sum()) > 1.0
