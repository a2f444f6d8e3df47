"""Each HIP leaf op vs the CPU oracle (values and gradients).  GPU box."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _close(got, ref, rtol=1e-4, atol_frac=1e-5, msg=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale + atol_frac * scale, f"{msg}: err {err:.3e} scale {scale:.3e}"


def _hand_obj(B=4, seed=0):
    from homan_amd import synth
    from homan_amd.mano_assets import synthetic_mano
    g = torch.Generator().manual_seed(seed)
    m = synthetic_mano(0)
    vh = torch.from_numpy(m["v_template"])[None].repeat(B, 1, 1) + torch.randn(B, 1, 3, generator=g) * 0.01
    vh = vh + torch.tensor([0.0, 0.0, 0.55])
    ov, of = synth.bottle_mesh()
    vo = torch.from_numpy(ov)[None].repeat(B, 1, 1) + torch.tensor([0.02, 0.0, 0.56]) + torch.randn(B, 1, 3, generator=g) * 0.01
    return m, vh, vo, torch.from_numpy(of)


@pytest.mark.parametrize("V", [300, 1502, 9000])     # 256 threads | one 1024-thread workgroup per frame | chunks + ticket
def test_rigid_transform_and_grads(V):
    from homan_amd import ops
    from oracle import model as om
    g = torch.Generator().manual_seed(0)
    N = 5
    mesh = torch.randn(N, V, 3, generator=g) * 0.1
    rot6d = torch.randn(N, 3, 2, generator=g)
    trans = torch.randn(N, 1, 3, generator=g)
    scale = torch.tensor([-1.3])
    w1, w2 = torch.randn(N, V, 3, generator=g), torch.randn(N, V, 3, generator=g)
    ins_o = [t.clone().requires_grad_(True) for t in (mesh, rot6d, trans, scale)]
    v, vd = om.transform_persp(ins_o[0], ins_o[2], om.rot6d_to_matrix(ins_o[1]), ins_o[3].abs())
    ((v * w1).sum() + (vd * w2).sum()).backward()
    ins_h = [t.clone().to(DEV).requires_grad_(True) for t in (mesh, rot6d, trans, scale)]
    vh, vdh = ops.rigid_transform(ins_h[0], ins_h[1], ins_h[2], ins_h[3], abs_scale=True)
    ((vh * w1.to(DEV)).sum() + (vdh * w2.to(DEV)).sum()).backward()
    # the oracle writes rot6d -> R and (s v) R + t out operation by operation and the kernel follows that order with
    # -ffp-contract=off: the vertices the rasteriser sees are the oracle's BIT FOR BIT (coverage of the hard rasteriser
    # depends on their last bit)
    np.testing.assert_array_equal(vh.detach().cpu().numpy(), v.detach().numpy())
    np.testing.assert_array_equal(vdh.detach().cpu().numpy(), vd.detach().numpy())
    for a, b, n in zip(ins_h, ins_o, ("mesh", "rot6d", "trans", "scale")):
        _close(a.grad, b.grad, rtol=2e-4, msg="grad " + n)


def test_mano_lbs_and_grads(mano_model):
    from homan_amd import ops
    from oracle import lbs
    g = torch.Generator().manual_seed(1)
    B = 6
    pca = torch.randn(B, 45, generator=g) * 0.4
    rot = torch.randn(B, 3, generator=g) * 0.5
    rot[0] = 0.0          # exercises the |r + 1e-8| branch at zero rotation
    betas = torch.randn(B, 10, generator=g) * 0.5
    trans = torch.randn(B, 3, generator=g) * 0.05
    w = torch.randn(B, 778, 3, generator=g)
    layer = lbs.ManoLayer(mano_model, num_pca_comps=16, flat_hand_mean=True)
    ins_o = [t.clone().requires_grad_(True) for t in (pca, rot, betas, trans)]
    hp = ins_o[0][:, :16] @ torch.as_tensor(mano_model["hand_components"][:16]) + torch.as_tensor(mano_model["hand_mean"])
    vo = layer(betas=ins_o[2], global_orient=ins_o[1], hand_pose=hp, transl=torch.zeros(B, 3))[0] + ins_o[3][:, None]
    (vo * w).sum().backward()
    mctx = ops.ManoContext(mano_model, DEV)
    ins_h = [t.clone().to(DEV).requires_grad_(True) for t in (pca, rot, betas, trans)]
    vh = ops.mano_lbs(ins_h[0], ins_h[1], ins_h[2], ins_h[3], mctx)
    (vh * w.to(DEV)).sum().backward()
    assert (vh.cpu() - vo).abs().max() < 2e-6
    # against the layer written out in the kernels' evaluation order (oracle/csrc/lbs_exact.c): the same bits
    from homan_amd.mano_assets import kernel_layout
    from oracle import clib
    lay = kernel_layout(mano_model, flat_hand_mean=False)
    exact = np.empty((B, 778, 3), np.float32)
    p_, r_, b_ = (np.ascontiguousarray(t.numpy(), np.float32) for t in (pca, rot, betas))
    clib.lib().orc_mano_forward(*[clib.fptr(a) for a in lay[:7]], clib.iptr(lay[7]), clib.fptr(p_), 45, clib.fptr(r_), clib.fptr(b_),
                                B, clib.fptr(exact))
    assert np.array_equal(vh.detach().cpu().numpy(), exact + trans.numpy()[:, None])
    for a, b, n in zip(ins_h, ins_o, ("pca", "rot", "betas", "trans")):
        _close(a.grad, b.grad, rtol=3e-4, atol_frac=1e-4, msg="grad " + n)
    _, joints = ops.mano_joints(ins_h[0], ins_h[1], ins_h[2], ins_h[3], mctx)
    jo = layer(betas=betas, global_orient=rot, hand_pose=hp.detach(), transl=torch.zeros(B, 3))[1] + trans[:, None]
    assert (joints.cpu() - jo).abs().max() < 2e-6


def test_v2d_smooth_priors():
    from homan_amd import ops
    from oracle import model as om
    g = torch.Generator().manual_seed(2)
    B, V = 5, 778
    verts = torch.randn(B, V, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.6])
    camintr = torch.tensor([[1.37, 0, 0.5], [0, 1.37, 0.5], [0, 0, 1.0]]).repeat(B, 1, 1)
    ref2d = torch.rand(B, V, 2, generator=g) * 256
    rws = ops.ReduceWorkspace(DEV)
    losses = om.OracleLosses(camintr, None, None, ref2d, None, 1, "centroid", 32)
    vo = verts.clone().requires_grad_(True)
    lo, mo = losses.compute_verts2d_loss_hand(vo, 256)
    (lo["loss_v2d_hand"] * 7).backward()
    vh = verts.clone().to(DEV).requires_grad_(True)
    lh, mh = ops.v2d_loss(vh, camintr.to(DEV), ref2d.to(DEV), 256, 1, rws)
    (lh * 7).backward()
    _close(lh, lo["loss_v2d_hand"], msg="v2d loss")
    np.testing.assert_allclose(mh.item(), mo["v2d_hand"], rtol=1e-5)
    _close(vh.grad, vo.grad, msg="v2d grad")
    # smooth
    vob = torch.randn(B, 1500, 3, generator=g)
    a, b = verts.clone().requires_grad_(True), vob.clone().requires_grad_(True)
    sm = om.compute_smooth_loss(a, b)
    (sm["loss_smooth_hand"] * 3 + sm["loss_smooth_obj"] * 5).backward()
    ah, bh = verts.clone().to(DEV).requires_grad_(True), vob.clone().to(DEV).requires_grad_(True)
    sh, so = ops.smooth_loss(ah, 1, rws), ops.smooth_loss(bh, 1, rws)
    (sh * 3 + so * 5).backward()
    _close(sh, sm["loss_smooth_hand"], msg="smooth hand")
    _close(so, sm["loss_smooth_obj"], msg="smooth obj")
    _close(ah.grad, a.grad, msg="smooth hand grad")
    _close(bh.grad, b.grad, msg="smooth obj grad")
    # priors
    pca = torch.randn(B, 45, generator=g)
    s_o, s_h = torch.tensor([1.2]), torch.tensor([0.9])
    po, so_, sh_ = pca.clone().requires_grad_(True), s_o.clone().requires_grad_(True), s_h.clone().requires_grad_(True)
    ref = om.compute_pca_loss(po)["loss_pca"] * 2 + om.compute_intrinsic_scale_prior(so_, torch.ones(1)) * 3 + \
        om.compute_intrinsic_scale_prior(sh_, torch.ones(1)) * 4
    ref.backward()
    ph, soh, shh = [t.clone().to(DEV).requires_grad_(True) for t in (pca, s_o, s_h)]
    one = torch.ones(1, device=DEV)
    l0, l1, l2 = ops.priors(ph, soh, one, shh, one)
    (l0 * 2 + l1 * 3 + l2 * 4).backward()
    _close(l0 * 2 + l1 * 3 + l2 * 4, ref, msg="priors")
    _close(ph.grad, po.grad, msg="pca grad")
    _close(soh.grad, so_.grad, msg="sobj grad")
    _close(shh.grad, sh_.grad, msg="shand grad")


def test_inter_and_contact(mano_model):
    from homan_amd import ops
    from oracle import model as om
    m, vh, vo, of = _hand_obj(B=5)
    B = 5
    camintr = torch.tensor([[1.37, 0, 0.5], [0, 1.37, 0.5], [0, 0, 1.0]]).repeat(B, 1, 1)
    vh[4] += torch.tensor([0.0, 0.0, 5.0])       # frame 4: z gap > 3 -> not interacting
    vh[3] += torch.tensor([2.0, 0.0, 0.0])       # frame 3: boxes do not overlap
    rws = ops.ReduceWorkspace(DEV)
    losses = om.OracleLosses(camintr, None, None, None, None, 1, "centroid", 32)
    a, b = vh.clone().requires_grad_(True), vo.clone().requires_grad_(True)
    lo, mo = losses.compute_interaction_loss(a.view(-1, 1, 778, 3), b.unsqueeze(1))
    (lo["loss_inter"] * 2).sum().backward()
    ah, bh = vh.clone().to(DEV).requires_grad_(True), vo.clone().to(DEV).requires_grad_(True)
    lh = ops.inter_loss(ah, bh, camintr.to(DEV), rws)
    (lh * 2).sum().backward()
    assert lh.shape == (1,)
    _close(lh, lo["loss_inter"], msg="inter")
    _close(ah.grad, a.grad, msg="inter grad hand")
    _close(bh.grad, b.grad, msg="inter grad obj")
    idx, d2, metric = ops.nearest_vertices(ah, bh, rws)
    np.testing.assert_allclose(metric.item(), mo["handobj_maxdist"], rtol=1e-4)
    # metric-only search (object-vertex groups pruned by bounding spheres): the very same float
    assert ops.nearest_vertices(ah, bh, rws, metric_only=True)[2].item() == metric.item()
    assert float(d2.min(1).values.max().sqrt()) == metric.item()
    # contact
    a, b = vh.clone().requires_grad_(True), vo.clone().requires_grad_(True)
    closed = torch.as_tensor(m["closed_faces"].astype(np.int64))
    co = om.compute_contact_loss(a, b, of[None].repeat(B, 1, 1), closed)["loss_contact"]
    (co * 3).sum().backward()
    ah.grad = None
    bh.grad = None
    # nearest-neighbour search: exact (the reference's |x|^2+|y|^2-2xy algebra loses ~1e-7 absolute on d^2 and may
    # pick another vertex among near-ties), so check it against a float64 brute force ...
    d_exact = ((vh.double()[:, :, None] - vo.double()[:, None]) ** 2).sum(-1)
    best = d_exact.min(2)
    picked = torch.gather(d_exact, 2, idx.cpu().long()[..., None])[..., 0]
    assert ((picked - best.values) <= 1e-6 * best.values + 1e-12).all()
    # ... and the loss + gradients with the oracle's own neighbour choice
    from oracle import yana
    idx_o = yana.batch_pairwise_dist(vh, vo).min(2)[1]
    agree = (idx_o == idx.cpu().long()).float().mean().item()
    assert agree > 0.97, agree
    ch_own = ops.contact_loss(ah.detach(), bh.detach(), idx, rws)
    _close(ch_own, co, rtol=2e-4, msg="contact (own neighbours)")
    ch = ops.contact_loss(ah, bh, idx_o.int().to(DEV), rws)
    (ch * 3).sum().backward()
    assert ch.shape == (1,)
    _close(ch, co, msg="contact")
    _close(ah.grad, a.grad, rtol=1e-3, atol_frac=1e-3, msg="contact grad hand")
    _close(bh.grad, b.grad, rtol=1e-3, atol_frac=1e-3, msg="contact grad obj")


@pytest.mark.parametrize("obj", ["bottle", "cube"])
def test_collision_vs_oracle(obj, mano_model):
    """SDF interpenetration loss vs the oracle.  The 500-triangle cube (not a multiple of the 64-lane wavefront) with the
    hand pushed deep inside is the case that once read a reduction result from lanes that had left the loop."""
    from homan_amd import ops, synth
    from oracle import model as om
    m, vh, vo, of = _hand_obj(B=3, seed=4)
    B = 3
    # push the hand into the object so that vertices of each mesh lie inside the other
    vh = vh + torch.tensor([0.03, 0.0, 0.0])
    if obj == "cube":
        ov, of = synth.box_mesh()
        of = torch.from_numpy(of)
        vo = torch.from_numpy(ov)[None].repeat(B, 1, 1) * 1.5 + vh.mean(1, keepdim=True) + torch.tensor([0.01, 0.0, 0.0])
    closed = torch.as_tensor(m["closed_faces"].astype(np.int64))
    a, b = vh.clone().requires_grad_(True), vo.clone().requires_grad_(True)
    lo, meta = om.sdf_scene_loss([closed, of], [a, b])
    assert lo.item() > 0
    (lo * 2).backward()
    cctx = ops.CollisionContext(m["closed_faces"], of, B, 778, vo.shape[1], DEV)
    ah, bh = vh.clone().to(DEV).requires_grad_(True), vo.clone().to(DEV).requires_grad_(True)
    lh = ops.collision_loss(ah, bh, cctx)
    (lh * 2).backward()
    for which in (0, 1):
        ref = meta["sdfs"][which]
        got = cctx.grid(which).cpu()
        assert ((got > 0) == (ref > 0)).all(), f"inside masks differ for mesh {which}"
        _close(got, ref, rtol=1e-5, msg=f"phi {which}")
        # ... bit for bit, in fact: a min over the triangles of ONE point-triangle routine (csrc/sdf.hip <-> oracle/csrc/sdf.c);
        # the written-out collision term of the free-running parity runs relies on it (oracle/handchain.py)
        assert torch.equal(got, ref.float().clamp(min=0)), f"phi {which}: {(got - ref).abs().max().item()}"
    _close(lh, lo, rtol=1e-4, msg="collision loss")
    _close(ah.grad, a.grad, rtol=1e-3, atol_frac=1e-3, msg="collision grad hand")
    _close(bh.grad, b.grad, rtol=1e-3, atol_frac=1e-3, msg="collision grad obj")
    # per-vertex penetration depths (reference scenesdf.py:141-146) and the evaluation metric built on them
    # (reference eval/pointmetrics.py:102-124)
    from homan_amd import pointmetrics
    dv = ops.collision_dist_values(vh.to(DEV), vo.to(DEV), cctx)
    for key in ((1, 0), (0, 1)):
        assert meta["dist_values"][key].max() > 0
        _close(dv[key], meta["dist_values"][key].detach(), rtol=1e-4, atol_frac=1e-5, msg=f"dist_values {key}")
    want = om.get_inter_metrics(vh, vo, closed[None], of[None])
    got = pointmetrics.get_inter_metrics(vh.to(DEV), vo.to(DEV), closed[None], of[None].to(DEV))
    assert got["has_contact"] == want["has_contact"] and all(want["has_contact"])
    np.testing.assert_allclose(got["pen_depths"], want["pen_depths"], rtol=1e-4)


def test_pair_terms_launch_equals_its_four_entry_points():
    """hm_pair_terms_fwd_clips (search | interaction | object smoothness | hand terms as block ranges of ONE launch) returns,
    bit for bit, what hm_nn_fwd_clips / hm_inter_fwd_clips / hm_smooth_fwd_clips / hm_hand_terms_fwd_clips return - two
    clips of four frames, per-clip outputs `stride` floats apart."""
    from homan_amd import constants as c
    from homan_amd import lib as hl
    from homan_amd.clipbatch import ClipReduceWorkspace
    L, P = hl.lib(), hl.ptr
    g = torch.Generator().manual_seed(5)
    C, CL, Vh, Vo, stride = 2, 4, 778, 1500, 14
    B = C * CL
    vh = (torch.randn(B, Vh, 3, generator=g) * 0.04 + torch.tensor([0.02, 0.0, 0.6])).to(DEV)
    vo = (torch.randn(B, Vo, 3, generator=g) * 0.05 + torch.tensor([0.0, 0.0, 0.62])).to(DEV)
    camintr = torch.tensor([[1.37, 0, 0.5], [0, 1.37, 0.5], [0, 0, 1.0]]).repeat(B, 1, 1).to(DEV)
    ref2d = (torch.rand(B, Vh, 2, generator=g) * 256).to(DEV)
    pca = torch.randn(B, 45, generator=g).to(DEV)
    s_o, s_h, one = torch.tensor([1.2, 0.8], device=DEV), torch.tensor([1.0, 1.0], device=DEV), torch.ones(C, device=DEV)
    order = torch.randperm(Vo, generator=g).to(torch.int32).to(DEV)
    stream = hl.stream()

    def run(fused):
        z = lambda *s: torch.zeros(*s, device=DEV)
        vals, rec = z(C, stride), z(B, 8)
        u_smo, u_v2d, u_smh, u_pca, u_so, u_sh = z(B, Vo, 3), z(B, Vh, 3), z(B, Vh, 3), z(B, 45), z(C), z(C)
        ws = [ClipReduceWorkspace(DEV, C) for _ in range(4)]
        slot = lambda i: vals.data_ptr() + 4 * i
        ht = (P(ref2d), 256.0, P(u_v2d), slot(0), P(u_smh), slot(2), P(pca), CL * 45, P(s_o), P(one), P(s_h), P(one), P(u_pca),
              P(u_so), P(u_sh), slot(3))
        if fused:
            hl.check(L.hm_pair_terms_fwd_clips(P(vh), P(vo), P(camintr), B, Vh, Vo, slot(6), P(order), P(ws[0].buf),
                                               c.INTERACTION_BBOX_EXPANSION, float(c.INTERACTION_Z_THRESH), P(rec), slot(7),
                                               P(ws[1].buf), P(u_smo), slot(8), P(ws[2].buf), *ht, P(ws[3].buf), None, None, None, None,
                                               None, None, None, None, CL, stride, stream), "pair terms")
        else:
            hl.check(L.hm_nn_fwd_clips(P(vh), P(vo), B, Vh, Vo, None, None, slot(6), P(ws[0].buf), CL, stride, P(order), stream),
                     "nn")
            hl.check(L.hm_inter_fwd_clips(P(vh), P(vo), P(camintr), B, Vh, Vo, c.INTERACTION_BBOX_EXPANSION,
                                          float(c.INTERACTION_Z_THRESH), P(rec), slot(7), P(ws[1].buf), CL, stride, stream), "inter")
            hl.check(L.hm_smooth_fwd_clips(P(vo), B, Vo, 1, P(u_smo), slot(8), P(ws[2].buf), CL, stride, stream), "smooth")
            hl.check(L.hm_hand_terms_fwd_clips(P(vh), P(camintr), 1, ht[0], ht[1], B, Vh, *ht[2:], P(ws[3].buf), CL, stride,
                                               stream), "hand terms")
        torch.cuda.synchronize()
        return [t.clone() for t in (vals, rec, u_smo, u_v2d, u_smh, u_pca, u_so, u_sh)]

    a, b = run(True), run(False)
    assert a[0].abs().sum() > 0 and (a[0][0] != a[0][1]).any()
    for x, y, name in zip(a, b, ["vals", "rec", "u_smo", "u_v2d", "u_smh", "u_pca", "u_so", "u_sh"]):
        assert torch.equal(x, y), name


def test_adam_step_with_log_row_equals_two_launches():
    """hm_adam_step_log = hm_log_total_clips + hm_adam_step: same parameters, same moments, same log rows, and the row lands
    in the slot of the step being taken (the step counter moves after every workgroup has read it)."""
    from homan_amd.jointopt import HmAdam
    from homan_amd import lib as hl
    g = torch.Generator().manual_seed(9)
    C, n, steps = 3, 13, 5
    weights = torch.rand(n, generator=g).to(DEV)
    weights[2] = 0.0

    def run(fused):
        gg = torch.Generator().manual_seed(11)
        params = [torch.nn.Parameter(torch.randn(7, 45, generator=gg).to(DEV)), torch.nn.Parameter(torch.randn(3000, generator=gg).to(DEV))]
        for p in params:
            p.grad = torch.zeros_like(p)
        opt = HmAdam([{"params": params[:1], "lr": 1e-2}, {"params": params[1:], "lr": 1e-1}])
        vals, log = torch.zeros(C, n + 1, device=DEV), torch.zeros(steps, C, n + 1, device=DEV)
        for t in range(steps):
            for p in params:
                p.grad.copy_(torch.randn(p.shape, generator=gg).to(DEV))
            vals[:, :n] = torch.rand(C, n, generator=gg).to(DEV)
            if fused:
                opt.step(zero_grad=False, log=(vals, weights, n, steps, log, C))
            else:
                hl.check(hl.lib().hm_log_total_clips(hl.ptr(vals), hl.ptr(weights), n, hl.ptr(opt.step_t), steps, hl.ptr(log),
                                                     C, hl.stream()), "log")
                opt.step(zero_grad=False)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in params] + [log.clone(), opt.step_t.clone()] + [m for ms in opt.state for m in ms]

    a, b = run(True), run(False)
    assert int(a[3][0]) == steps and a[2].abs().sum() > 0
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_metric_search_with_rigid_group_spheres_is_exact():
    """hm_nn_fwd_rigid_clips: bounding spheres of the object's vertex groups carried from MESH space into every frame by the
    rigid transform only decide which groups are scanned - the metric is the same float as without them, and the true
    max-over-frames of the min hand-object vertex distance (float64 brute force)."""
    from homan_amd import lib as hl
    from homan_amd import ops
    from homan_amd.clipbatch import ClipReduceWorkspace
    from homan_amd.jointopt import _morton_order
    L, P = hl.lib(), hl.ptr
    g = torch.Generator().manual_seed(3)
    C, CL, Vh, Vo = 2, 3, 778, 1502
    B = C * CL
    mesh = (torch.randn(1, Vo, 3, generator=g) * torch.tensor([0.03, 0.03, 0.08])).repeat(B, 1, 1).to(DEV)
    rot6d = torch.randn(B, 3, 2, generator=g).to(DEV)
    trans = (torch.randn(B, 1, 3, generator=g) * 0.02 + torch.tensor([0.0, 0.0, 0.6])).to(DEV)
    scale = torch.tensor([1.3, -0.8], device=DEV)                      # one per clip, used as |s|
    vo = torch.cat([ops.rigid_transform(mesh[c * CL:(c + 1) * CL], rot6d[c * CL:(c + 1) * CL], trans[c * CL:(c + 1) * CL],
                                        scale[c:c + 1], abs_scale=True)[0] for c in range(C)]).contiguous()
    vh = (torch.randn(B, Vh, 3, generator=g) * 0.04 + torch.tensor([0.09, 0.0, 0.6])).to(DEV)
    order = _morton_order(mesh[0]).to(DEV)
    ng = (Vo + 63) // 64
    vs = mesh[:, order.long()]
    vs = torch.cat([vs, vs[:, -1:].expand(-1, ng * 64 - Vo, -1)], 1).reshape(B, ng, 64, 3)
    ctr = torch.stack([vs[:, k, :(64 if k < ng - 1 else Vo - 64 * (ng - 1))].mean(1) for k in range(ng)], 1)
    rad = ((vs - ctr[:, :, None]) ** 2).sum(-1).sqrt().amax(2)
    spheres = torch.cat([ctr, rad[..., None]], -1).contiguous()
    outs = []
    # (and with the hand's vertices dealt to the workgroups in another order - a spatial sort in the loop, any permutation here)
    perm = torch.randperm(Vh, generator=torch.Generator().manual_seed(3)).to(device=DEV, dtype=torch.int32)
    for sph, ho in ((None, None), (spheres, None), (spheres, perm), (None, perm)):
        out = torch.zeros(C, 5, device=DEV)
        ws = ClipReduceWorkspace(DEV, C)
        hl.check(L.hm_nn_fwd_rigid_clips(P(vh), P(vo), B, Vh, Vo, None, None, P(out), P(ws.buf), CL, 5, P(order),
                                         P(sph) if sph is not None else None, P(rot6d), P(trans.reshape(B, 3).contiguous()),
                                         P(scale), P(ho) if ho is not None else None, None, hl.stream()), "nn")
        torch.cuda.synchronize()
        outs.append(out[:, 0].clone())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)
    # ... and with the SEED of the previous call (nn_seed: the pair that held each frame's minimum last time bounds this call's
    # minimum before anything is scanned): zero-filled at first, then carried over three calls while the hand moves - every call
    # returns the unseeded search's value, and after a call the seed names, per frame, a pair AT the minimum
    seed = torch.zeros((2 + (Vh + 127) // 128) * B, dtype=torch.int32, device=DEV)
    vh_t = vh.clone()
    for it in range(4):
        res = []
        for sd in (None, seed):
            out = torch.zeros(C, 5, device=DEV)
            ws = ClipReduceWorkspace(DEV, C)
            hl.check(L.hm_nn_fwd_rigid_clips(P(vh_t), P(vo), B, Vh, Vo, None, None, P(out), P(ws.buf), CL, 5, P(order), P(spheres),
                                             P(rot6d), P(trans.reshape(B, 3).contiguous()), P(scale), P(perm),
                                             P(sd) if sd is not None else None, hl.stream()), "nn")
            torch.cuda.synchronize()
            res.append(out[:, 0].clone())
        assert torch.equal(res[0], res[1]), it
        pairs = seed[:2 * B].reshape(B, 2).long()
        dpair = (vh_t[torch.arange(B), pairs[:, 0]] - vo[torch.arange(B), pairs[:, 1]]).double().norm(dim=1)
        dmin = torch.cdist(vh_t.double(), vo.double()).amin((1, 2))
        np.testing.assert_allclose(dpair.cpu().numpy(), dmin.cpu().numpy(), rtol=1e-5)
        vh_t = vh_t + 0.004 * (it + 1) * torch.tensor([1.0, -0.5, 0.25], device=DEV)        # the hand moves on
    d = torch.cdist(vh.double(), vo.double()).amin((1, 2)).reshape(C, CL).amax(1)
    np.testing.assert_allclose(outs[1].cpu().numpy(), d.cpu().numpy(), rtol=1e-5)
