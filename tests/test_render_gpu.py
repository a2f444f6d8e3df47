"""Visualisation renders: rgb / depth / alpha of `renderer.render` and the model-level
render / visualize_hand_object surface (reference homan/homan.py:510-613, homan/visualize.py:44-128) against the oracle's
restatement of the NMR `render` output (lighting + per-face colours; PARITY UNPINNED leaf, see oracle/nmr.py)."""
import copy

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(B=3, S=64):
    from homan_amd import synth
    ov, of = synth.bottle_mesh()
    g = torch.Generator().manual_seed(3)
    verts = torch.from_numpy(ov)[None].repeat(B, 1, 1)
    ang = torch.rand(B, generator=g) * 1.5
    R = torch.stack([torch.tensor(synth._rot_x(float(a)) @ synth._rot_y(0.3), dtype=torch.float32) for a in ang])
    verts = verts @ R + torch.tensor([0.0, 0.0, 0.55]) + torch.randn(B, 1, 3, generator=g) * 0.01
    faces = torch.from_numpy(of).long()[None].repeat(B, 1, 1)
    K = torch.tensor([[[1.3, 0, 0.5], [0, 1.3, 0.5], [0, 0, 1]]]).repeat(B, 1, 1)
    tex = torch.rand(B, faces.shape[1], 1, 1, 1, 3, generator=g)
    return verts, faces, K, tex


@pytest.mark.parametrize("S", [64, 640])      # 640: the full-image size of the reference's `model.renderer` (EPIC frames)
def test_render_rgb_depth_alpha_vs_oracle(S):
    from homan_amd import nmr as hnmr
    from oracle import nmr as onmr
    verts, faces, K, tex = _scene(B=3 if S == 64 else 2, S=S)
    ro = onmr.Renderer(image_size=S, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    rh = hnmr.Renderer(image_size=S, K=K.to(DEV), orig_size=1)
    for r in (ro, rh):      # the light of reference homan/homan.py:173-176
        r.light_direction = [1, 0.5, 1]
        r.light_intensity_direction = 0.3
        r.light_intensity_ambient = 0.5
        r.background_color = [1.0, 1.0, 1.0]
    rgb_o, dep_o, al_o = ro.render(verts, faces, tex)
    rgb_h, dep_h, al_h = rh.render(verts.to(DEV), faces.to(DEV), tex.to(DEV))
    assert torch.equal(al_h.cpu(), al_o)                                   # coverage is integer work: bit-exact
    np.testing.assert_allclose(dep_h.cpu().numpy(), dep_o.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb_h.cpu().numpy(), rgb_o.numpy(), rtol=0, atol=2e-6)      # float shading: 2e-6 absolute
    assert 0.03 < float(al_o.mean()) < 0.9 and float(rgb_o.std()) > 0.05   # a real image, lit and partly background
    sil = rh(verts.to(DEV), faces.to(DEV), mode="silhouettes")
    assert torch.equal(sil, al_h)
    # default light / black background (upstream constructor defaults)
    ro2 = onmr.Renderer(image_size=S, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    rh2 = hnmr.Renderer(image_size=S, K=K.to(DEV), orig_size=1)
    np.testing.assert_allclose(rh2.render(verts.to(DEV), faces.to(DEV), tex.to(DEV))[0].cpu().numpy(),
                               ro2.render(verts, faces, tex)[0].numpy(), rtol=0, atol=2e-6)


def test_model_render_and_visualize(mano_model, tmp_path):
    """model.render / render_gt / render_with_gt / visualize_hand_object: shapes, dtypes and the reference's
    composition - the combined [object, hand] mesh in gold / grey over a white background equals a direct oracle render
    of the same combined mesh."""
    from homan_amd import synth, visualize
    from homan_amd.jointopt import build_model
    from homan_amd.meshutils import COLORS
    from oracle import nmr as onmr
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    S = 64
    clip = synth.make_clip(seed=5, frames=3, rend_size=S, image_size=S, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                        optimize_mano=True, image_size=S, mano_model=mano_model, rend_size=S)
    B = 3
    Fo, Fh = model.faces_object.shape[1], model.faces_hand.shape[1]
    assert tuple(model.faces.shape) == (B, Fo + Fh, 3) and tuple(model.textures.shape) == (B, Fo + Fh, 1, 1, 1, 3)
    np.testing.assert_allclose(model.textures[0, 0].reshape(3).cpu().numpy(), COLORS["gold"], rtol=1e-6)
    np.testing.assert_allclose(model.textures[0, -1].reshape(3).cpu().numpy(), COLORS["grey"], rtol=1e-6)
    imgs, masks = model.render(model.renderer, viz_len=7, max_in_batch=2)
    assert imgs.shape == (B, S, S, 3) and masks.shape == (B, S, S) and masks.dtype == bool and imgs.dtype == np.float32
    with torch.no_grad():
        vo, vh = model.get_verts_object()[0].cpu(), model.get_verts_hand()[0].cpu()
    comb = torch.cat([vo, vh], 1)
    ro = onmr.Renderer(image_size=S, K=model.camintr.cpu(), R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    ro.light_direction, ro.light_intensity_direction, ro.background_color = [1, 0.5, 1], 0.3, [1.0, 1.0, 1.0]
    rgb_o, _, al_o = ro.render(comb, model.faces.cpu(), model.textures.cpu())
    np.testing.assert_array_equal(masks, al_o.numpy().astype(bool))
    np.testing.assert_allclose(imgs, np.clip(rgb_o.numpy().transpose(0, 2, 3, 1), 0, 1), rtol=0, atol=2e-6)
    # ground-truth colours and the pred + gt overlay
    g_imgs, _ = model.render_gt(model.renderer, verts_hand_gt=vh.to(DEV), verts_object_gt=vo.to(DEV), viz_len=2)
    assert g_imgs.shape == (2, S, S, 3) and not np.allclose(g_imgs, imgs[:2])
    w_imgs, w_masks = model.render_with_gt(model.renderer, verts_hand_gt=[vh.to(DEV) + 0.02], verts_object_gt=vo.to(DEV) + 0.02,
                                           viz_len=7, max_in_batch=2)
    assert w_imgs.shape == (B, S, S, 3) and (w_masks.sum() >= masks.sum())
    images = (np.random.RandomState(0).rand(B, 48, S, 3) * 255).astype(np.uint8)      # h < w: padded to a square
    front, top = visualize.visualize_hand_object(model, images, viz_len=7, max_in_batch=2)
    assert front.shape == (B, 48, S, 3) and front.dtype == np.uint8 and top.shape == (B, S, S, 3) and top.dtype == np.uint8
    m = masks[0][:48]
    np.testing.assert_array_equal(front[0][m], (imgs[0][:48][m] * 255).astype(np.uint8))
    np.testing.assert_array_equal(front[0][~m], images[0][~m])
    assert not np.array_equal(top, (imgs * 255).astype(np.uint8))       # the rotated view differs from the frontal one
    model.save_obj(tmp_path / "scene.obj")
    lines = open(tmp_path / "scene.obj").read().splitlines()
    assert sum(l.startswith("v ") for l in lines) == comb.shape[1] and sum(l.startswith("f ") for l in lines) == Fo + Fh


@pytest.mark.parametrize("mode", ["eager", "fused"])
def test_optimize_hand_object_saves_frames(mode, mano_model, tmp_path):
    """reference jointopt.py:158-176: a frame every viz_step iterations when the input images are given."""
    from homan_amd import synth
    from homan_amd.jointopt import optimize_hand_object
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    S = 64
    clip = synth.make_clip(seed=5, frames=3, rend_size=S, image_size=S, obj="bottle", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    images = (np.random.RandomState(0).rand(3, S, S, 3) * 255).astype(np.uint8)
    model, evo, imgs = optimize_hand_object(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                                            objvertices=clip["objvertices"], objfaces=clip["objfaces"],
                                            camintr=clip["camintr"], loss_weights=dict(synth.STEP2_LOSS_WEIGHTS),
                                            num_iterations=5, images=images, viz_step=2, viz_folder=str(tmp_path / mode),
                                            optimize_mano=True, image_size=S, rend_size=S, mano_model=mano_model, mode=mode)
    assert list(imgs) == [0, 2, 4] and len(evo["loss"]) == 5
    from PIL import Image
    im = np.asarray(Image.open(imgs[4]))
    assert im.shape == (S, 3 * S // 2, 3)       # (frontal over top-down) x 3 frames, halved


def test_render_at_350_the_core50_image_size():
    """reference homan/getdataset.py:35 sets image_size=350 for Core50 and homan/homan.py:168-176 builds the renderer at
    image_size: not a multiple of 32, rendered on a 352 grid with rescaled intrinsics and cropped (ops.SilhouetteContext).
    Same rays, different rounding of the sample positions: coverage / depth / colour agree with the oracle's native 350
    render except on a sliver of boundary samples."""
    from homan_amd import nmr as hnmr
    from homan_amd import ops
    from oracle import nmr as onmr
    S = 350
    verts, faces, K, tex = _scene(B=2, S=S)
    ro = onmr.Renderer(image_size=S, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
    rh = hnmr.Renderer(image_size=S, K=K.to(DEV), orig_size=1)
    rgb_o, dep_o, al_o = ro.render(verts, faces, tex)
    rgb_h, dep_h, al_h = rh.render(verts.to(DEV), faces.to(DEV), tex.to(DEV))
    assert tuple(al_h.shape) == (2, S, S) and tuple(rgb_h.shape) == (2, 3, S, S) and tuple(dep_h.shape) == (2, S, S)
    diff = (al_h.cpu() != al_o)
    assert float(diff.float().mean()) < 2e-4, float(diff.float().mean())          # boundary samples only
    assert float((al_h.cpu() - al_o).abs().max()) <= 0.25                          # ... and one sample of four at most
    same = ~diff
    np.testing.assert_allclose(dep_h.cpu()[same].numpy(), dep_o[same].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rgb_h.cpu().permute(0, 2, 3, 1)[same].numpy(), rgb_o.permute(0, 2, 3, 1)[same].numpy(), atol=2e-3)
    assert 0.03 < float(al_o.mean()) < 0.9
    # the differentiable silhouette at 350: value and NMR pseudo-gradient against the oracle's native 350 render
    ctx = ops.SilhouetteContext(faces.to(DEV), verts.shape[1], 2, S, DEV)
    vh = verts.to(DEV).clone().requires_grad_(True)
    sil_h = ops.silhouette_render(vh, K.to(DEV), ctx)
    g = torch.Generator().manual_seed(0)
    w = torch.rand(2, S, S, generator=g) - 0.5
    (sil_h * w.to(DEV)).sum().backward()
    vo_ = verts.clone().requires_grad_(True)
    sil_o = ro(vo_, faces, mode="silhouettes")
    (sil_o * w).sum().backward()
    assert tuple(sil_h.shape) == (2, S, S)
    assert float((sil_h.detach().cpu() != sil_o.detach()).float().mean()) < 2e-4
    scale = vo_.grad.abs().max().item()
    assert scale > 0
    err = (vh.grad.cpu() - vo_.grad).abs().max().item() / scale
    assert err < 0.05, err      # boundary samples flip between the two grids: a few per cent of the largest entry


def test_core50_sized_model_depth_term_and_viz(mano_model):
    """image_size=350 end to end: HOMan builds, `render` / `visualize_hand_object` work, and the ordinal depth term
    (reference homan/homan.py:384-419) evaluates and back-propagates at the full-image size."""
    from homan_amd import synth, visualize
    from homan_amd.jointopt import build_model
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=2, frames=2, rend_size=64, image_size=350, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                        objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                        optimize_mano=True, image_size=350, mano_model=mano_model, rend_size=64, ordinal_depth=True)
    imgs, masks = model.render(model.renderer, viz_len=2)
    assert imgs.shape == (2, 350, 350, 3) and masks.shape == (2, 350, 350) and masks.any()
    images = (np.random.RandomState(0).rand(2, 350, 350, 3) * 255).astype(np.uint8)
    front, top = visualize.visualize_hand_object(model, images, viz_len=2)
    assert front.shape == (2, 350, 350, 3)
    lw = dict(synth.STEP1_LOSS_WEIGHTS, lw_depth=1.0)
    loss_dict, _ = model(loss_weights=lw)
    assert "loss_depth" in loss_dict and torch.isfinite(loss_dict["loss_depth"]).all()
    sum(loss_dict[k] * lw[k.replace("loss", "lw")] for k in loss_dict).sum().backward()
    assert torch.isfinite(model.translations_object.grad).all()
