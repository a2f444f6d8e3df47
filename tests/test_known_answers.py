"""Known-answer tests: analytic truths that hold for ANY correct implementation of the published
algorithms, independent of the oracle's restatement - the leaves (NMR rasteriser, voxel SDF, MANO LBS) are third-party
and un-pinned upstream, so these anchor the oracle itself, and the HIP kernels directly.

Every case runs on the CPU oracle (always) and on the HIP kernels (-m gpu)."""
import numpy as np
import pytest
import torch

IMPLS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]
K_UNIT = torch.tensor([[[1.0, 0, 0.0], [0, 1.0, 0.0], [0, 0, 1.0]]])      # image point = (x/z, y/z) on [0,1]^2


def _silhouettes(impl, verts, faces, S, K=K_UNIT):
    """(B,V,3) camera-space vertices -> (B,S,S) anti-aliased silhouettes (fill_back, 2x SSAA, orig_size 1)."""
    B = verts.shape[0]
    if impl == "oracle":
        from oracle import nmr
        r = nmr.Renderer(image_size=S, K=K.repeat(B, 1, 1), R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
        return r(verts, faces, mode="silhouettes")
    from homan_amd import ops
    dev = torch.device("cuda")
    sctx = ops.SilhouetteContext(faces.to(dev), verts.shape[1], B, S, dev)
    return ops.silhouette_render(verts.to(dev), K.repeat(B, 1, 1).to(dev), sctx).cpu()


@pytest.mark.parametrize("impl", IMPLS)
def test_full_screen_quad_covers_every_sample(impl):
    v = torch.tensor([[[-1.0, -1.0, 2.0], [3.0, -1.0, 2.0], [3.0, 3.0, 2.0], [-1.0, 3.0, 2.0]]])      # image [-.5, 1.5]^2
    f = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
    assert torch.equal(_silhouettes(impl, v, f, 32), torch.ones(1, 32, 32))


@pytest.mark.parametrize("impl", IMPLS)
def test_half_plane_edge_gives_quarter_steps_and_the_right_columns(impl):
    """A vertical edge at image x = x_e: samples sit at (i + 0.5) / (2S); output pixel c averages the sample columns 2c and
    2c + 1.  Columns left of the edge are 1, right of it 0, the column the edge cuts is 0.5 when it runs between the two
    sample columns of that pixel (never anything but a multiple of 1/4: two of four samples share each column)."""
    S = 32
    for x_e, cut, value in ((20.5 / 32, 20, 0.5), (20.0 / 32, None, None), (20.25 / 32 + 1e-4, 20, 0.5)):
        v = torch.tensor([[[-1.0, -1.0, 1.0], [x_e, -1.0, 1.0], [x_e, 3.0, 1.0], [-1.0, 3.0, 1.0]]]) * 2.0   # z = 2, same image
        f = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
        img = _silhouettes(impl, v, f, S)[0]
        assert bool((img == img[:1]).all())                                  # no dependence on the row
        row = img[0]
        assert set(np.unique(row.numpy()).tolist()) <= {0.0, 0.25, 0.5, 0.75, 1.0}
        edge_px = int(np.floor(x_e * S))
        assert bool((row[:edge_px] == 1).all()) and bool((row[edge_px + 1:] == 0).all())
        if cut is not None:
            assert row[cut].item() == value


@pytest.mark.parametrize("impl", IMPLS)
def test_back_facing_triangle_is_rendered_through_fill_back(impl):
    tri = torch.tensor([[[0.2, 0.2, 1.0], [0.8, 0.2, 1.0], [0.5, 0.8, 1.0]]])
    front = _silhouettes(impl, tri, torch.tensor([[[0, 1, 2]]]), 32)
    back = _silhouettes(impl, tri, torch.tensor([[[0, 2, 1]]]), 32)
    assert front.sum() > 50 and torch.equal(front, back)


@pytest.mark.parametrize("impl", IMPLS)
def test_cube_sdf_matches_the_analytic_distance(impl):
    """Inside an axis-aligned cube the distance to the surface is h - max|x_i|; outside the reference keeps 0."""
    h = 0.6
    c = np.array([[-1, -1, -1], [1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, 1], [1, -1, 1], [1, 1, 1], [-1, 1, 1]], np.float32) * h
    f = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6], [1, 2, 6], [1, 6, 5],
                  [0, 4, 7], [0, 7, 3]], np.int32)
    N = 32
    ctr = (np.arange(N, dtype=np.float32) + 0.5) * (2.0 / N) - 1.0            # voxel centres of align_corners=False
    zz, yy, xx = np.meshgrid(ctr, ctr, ctr, indexing="ij")
    inside = np.maximum(np.maximum(np.abs(xx), np.abs(yy)), np.abs(zz)) < h
    want = np.where(inside, h - np.maximum(np.maximum(np.abs(xx), np.abs(yy)), np.abs(zz)), 0.0).astype(np.float32)
    if impl == "oracle":
        from oracle import sdfgrid
        phi = sdfgrid.SDF(clamp_outside=True)(torch.from_numpy(f), torch.from_numpy(c)[None]).clamp(0)[0].numpy()
    else:
        from homan_amd import ops
        dev = torch.device("cuda")
        # the collision op normalises each mesh by its own box (scale_factor 0): mesh 1 = the cube scaled to half-size h,
        # i.e. vertices / box_scale land exactly on +-1 -> use a cube of half-size 1 and compare in its own frame
        cube = torch.from_numpy(c / h)[None].to(dev)
        other = (torch.from_numpy(c / h)[None] * 0.5).to(dev)
        cctx = ops.CollisionContext(f, torch.from_numpy(f), 1, 8, 8, dev)
        ops.collision_loss(other, cube, cctx, scale_factor=0.0)
        phi = cctx.grid(1)[0].cpu().numpy()
        inside = np.maximum(np.maximum(np.abs(xx), np.abs(yy)), np.abs(zz)) < 1.0
        want = np.where(inside, 1.0 - np.maximum(np.maximum(np.abs(xx), np.abs(yy)), np.abs(zz)), 0.0).astype(np.float32)
    assert ((phi > 0) == inside).all()
    np.testing.assert_allclose(phi, want, atol=1e-6)


@pytest.mark.parametrize("impl", IMPLS)
def test_lbs_at_rest_is_the_template_and_rot6d_identity(impl, mano_model):
    tmpl = torch.from_numpy(np.asarray(mano_model["v_template"], np.float32))
    B = 2
    pca, rot, betas = torch.zeros(B, 16), torch.zeros(B, 3), torch.zeros(B, 10)
    trans = torch.tensor([[0.1, -0.2, 0.3], [0.0, 0.0, 0.0]])
    eye6 = torch.eye(3)[:, :2].reshape(1, 3, 2).repeat(B, 1, 1)
    if impl == "oracle":
        from oracle import lbs
        from oracle.model import rot6d_to_matrix
        layer = lbs.ManoLayer(mano_model, num_pca_comps=16, flat_hand_mean=True, use_pca=True)
        verts = layer(betas=betas, global_orient=rot, hand_pose=pca, transl=trans)[0]
        R = rot6d_to_matrix(eye6)
        moved = verts @ R + 0.0
    else:
        from homan_amd import ops
        from homan_amd.manomodel import ManoModel
        dev = torch.device("cuda")
        mm = ManoModel("extra_data/mano", pca_comps=16, mano_model=mano_model, device=dev)
        verts = mm.forward_pca(pca.to(dev), rot=rot.to(dev), betas=betas.to(dev), side="right", flat_hand_mean=True,
                               trans=trans.to(dev))["verts"]          # (with the mean pose added, "rest" is not the template)
        moved = ops.rigid_transform(verts.contiguous(), eye6.to(dev), torch.zeros(B, 1, 3, device=dev),
                                    torch.ones(1, device=dev), False)[0].cpu()
        verts = verts.cpu()
    np.testing.assert_allclose(verts.numpy(), (tmpl[None] + trans[:, None]).numpy(), atol=1e-6)
    np.testing.assert_allclose(moved.numpy(), verts.numpy(), atol=1e-6)       # rot6d of the identity rotates nothing


@pytest.mark.parametrize("impl", IMPLS)
def test_contact_loss_of_coincident_point_sets_is_zero(impl, mano_model):
    tmpl = torch.from_numpy(np.asarray(mano_model["v_template"], np.float32))
    pts = tmpl[None].repeat(2, 1, 1) + torch.tensor([[[0.0, 0.0, 0.5]], [[0.01, 0.0, 0.6]]])
    if impl == "oracle":
        from oracle import model as om
        closed = torch.as_tensor(np.asarray(mano_model["closed_faces"]).astype(np.int64))
        loss = om.compute_contact_loss(pts, pts.clone(), closed[None], closed)["loss_contact"]
    else:
        from homan_amd import ops
        dev = torch.device("cuda")
        rws = ops.ReduceWorkspace(dev)
        nn = ops.nearest_vertices(pts.to(dev), pts.clone().to(dev), rws)
        assert torch.equal(nn[0].cpu(), torch.arange(778)[None].repeat(2, 1).to(nn[0].dtype))
        loss = ops.contact_loss(pts.to(dev), pts.clone().to(dev), nn[0], rws, 0.02)
    assert abs(float(loss.reshape(-1)[0])) < 1e-9


@pytest.mark.parametrize("impl", IMPLS)
def test_pseudo_gradient_pulls_the_silhouette_towards_the_target(impl):
    """The NMR backward is a heuristic, but its purpose is checkable: for L = sum (silhouette - target)^2 with the target a
    copy of the shape shifted by +3 px in x and -2 px in y (image rows grow downwards = camera y grows), gradient descent
    must move every vertex towards +x and -y, i.e. dL/dx < 0 and dL/dy > 0 at all four corners."""
    S = 64
    def quad(x0, y0, x1, y1, z=2.0):
        return torch.tensor([[[x0, y0, 1.0], [x1, y0, 1.0], [x1, y1, 1.0], [x0, y1, 1.0]]]) * z
    f = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
    px = 1.0 / S
    v = quad(0.3, 0.35, 0.6, 0.7).requires_grad_(True)
    target = _silhouettes(impl, quad(0.3 + 3 * px, 0.35 - 2 * px, 0.6 + 3 * px, 0.7 - 2 * px), f, S)
    if impl == "oracle":
        from oracle import nmr
        r = nmr.Renderer(image_size=S, K=K_UNIT, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
        img = r(v, f, mode="silhouettes")
        ((img - target) ** 2).sum().backward()
        g = v.grad[0]
    else:
        from homan_amd import ops
        dev = torch.device("cuda")
        sctx = ops.SilhouetteContext(f.to(dev), 4, 1, S, dev)
        vd = v.detach().to(dev).requires_grad_(True)
        img = ops.silhouette_render(vd, K_UNIT.to(dev), sctx)
        ((img - target.to(dev)) ** 2).sum().backward()
        g = vd.grad[0].cpu()
    assert bool((g[:, 0] < 0).all()), g           # move right
    assert bool((g[:, 1] > 0).all()), g           # move up in the image = smaller camera y


@pytest.mark.parametrize("impl", IMPLS)
def test_pseudo_gradient_magnitude_of_a_single_edge(impl):
    """MAGNITUDE of the NMR pseudo-gradient (Kato et al. 2018, backward of the silhouette) on a case small enough to write
    down: one triangle whose right edge B-C is vertical at sample position x_e, and an upstream gradient dL/dsilhouette
    that is -1 on ONE pixel column right of the edge (rows of the middle of the edge), 0 elsewhere.  Only the row sweeps of
    that edge meet a non-zero gradient: for every sample row d0 crossing the edge, every uncovered sample s right of it with
    g_s = dL/dpixel / 4 < 0 adds  g_s / (dist + eps)  to an end point e of the edge, dist = (x_s - x_e) * (2 / is) / w_e(d0),
    w_e = the end point's interpolation weight on that row, eps = 1e-3.  The sum is evaluated here in float64 and pushed
    through d(NDC x)/d(camera x) = 2 / z."""
    S, z, eps = 64, 2.0, 1e-3
    is_ = 2 * S
    x_e = 0.6 + 0.3 / is_                       # sample position 76.6: the last covered sample of a row is 76
    A, Bv, C = (0.15, 0.5), (x_e, 0.2), (x_e, 0.8)
    v = (torch.tensor([[[A[0], A[1], 1.0], [Bv[0], Bv[1], 1.0], [C[0], C[1], 1.0]]]) * z)
    f = torch.tensor([[[0, 1, 2]]])
    col, rows = 40, range(24, 40)
    W = torch.zeros(1, S, S)
    W[0, rows.start:rows.stop, col] = -1.0
    if impl == "oracle":
        from oracle import nmr
        vv = v.clone().requires_grad_(True)
        r = nmr.Renderer(image_size=S, K=K_UNIT, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
        img = r(vv, f, mode="silhouettes")
        (img * W).sum().backward()
        g = vv.grad[0].double().numpy()
    else:
        from homan_amd import ops
        dev = torch.device("cuda")
        sctx = ops.SilhouetteContext(f.to(dev), 3, 1, S, dev)
        vv = v.to(dev).requires_grad_(True)
        img = ops.silhouette_render(vv, K_UNIT.to(dev), sctx)
        (img * W.to(dev)).sum().backward()
        g = vv.grad[0].double().cpu().numpy()
        img = img.cpu()
    assert img[0, 30, 30] == 1 and img[0, 30, col] == 0                     # inside left of the edge, empty at the column
    px_e = x_e * is_ - 0.5                                                  # sample-space position of the edge
    py = {1: (1 - Bv[1]) * is_ - 0.5, 2: (1 - C[1]) * is_ - 0.5}            # end points along the sample rows (y is flipped)
    expected = {}
    for e, other in ((1, 2), (2, 1)):
        acc = 0.0
        for d0 in range(int(np.ceil(min(py.values()))), int(np.floor(max(py.values()))) + 1):
            row = (is_ - 1 - d0) >> 1
            if row not in rows:
                continue
            w_e = (py[other] - d0) / (py[other] - py[e])
            for xs in (2 * col, 2 * col + 1):
                acc += (0.25 * -1.0) / ((xs - px_e) * (2.0 / is_) / w_e + eps)
        expected[e] = acc * 2.0 / z
    for e in (1, 2):
        np.testing.assert_allclose(g[e, 0], expected[e], rtol=2e-4, err_msg=f"vertex {e}")
        assert expected[e] < 0                                              # gradient descent moves the edge right, into the column
    assert abs(g[0, 0]) < 1e-7 * abs(expected[1])                           # the apex's edges sweep away from the column


@pytest.mark.parametrize("impl", IMPLS)
def test_a_face_with_a_vertex_on_the_image_plane_is_culled_in_every_pass(impl):
    """A vertex at camera depth ~1e-21 projects beyond 1e15 NDC units: the edge functions of its faces overflow to inf - inf.
    Such a face is culled - in the forward (no coverage from it) AND in the pseudo-gradient (the oracle's backward
    used to cull back faces only): a quad next to it renders and gets the finite gradient it gets alone."""
    S = 32
    z = 2.0
    quad = torch.tensor([[0.3, 0.3, 1.0], [0.7, 0.3, 1.0], [0.7, 0.7, 1.0], [0.3, 0.7, 1.0]]) * z
    wild = torch.tensor([[0.4 * z, 0.4 * z, z], [0.5 * z, 0.6 * z, z], [1e-3, 1e-3, 1e-21]])        # third vertex ON the image plane
    f_quad = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
    f_both = torch.tensor([[[0, 1, 2], [0, 2, 3], [4, 5, 6]]])
    v_alone, v_both = quad[None].clone(), torch.cat([quad, wild])[None].clone()
    target = torch.zeros(1, S, S)
    target[:, 8:20, 12:26] = 1.0

    def run(v, f):
        v = v.clone().requires_grad_(True)
        if impl == "oracle":
            from oracle import nmr
            r = nmr.Renderer(image_size=S, K=K_UNIT, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
            img = r(v, f, mode="silhouettes")
            ((img - target) ** 2).sum().backward()
            return img.detach(), v.grad[0]
        from homan_amd import ops
        dev = torch.device("cuda")
        sctx = ops.SilhouetteContext(f.to(dev), v.shape[1], 1, S, dev)
        vd = v.detach().to(dev).requires_grad_(True)
        img = ops.silhouette_render(vd, K_UNIT.to(dev), sctx)
        ((img - target.to(dev)) ** 2).sum().backward()
        return img.detach().cpu(), vd.grad[0].cpu()

    img_a, g_a = run(v_alone, f_quad)
    img_b, g_b = run(v_both, f_both)
    assert torch.equal(img_a, img_b)                       # the wild face covers nothing
    assert bool(torch.isfinite(g_b).all())
    assert torch.equal(g_b[:4], g_a) and not g_b[4:].any()         # ... and pulls on nothing
