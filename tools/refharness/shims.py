"""Make the reference's own composition code importable on CPU (build container only).

Runs the reference's Python IN PLACE from /root/reference (never copied, no
bytecode written) over the repo's CPU oracle leaves, so that golden vectors for
everything above the leaves come from the reference's own code (SURVEY 8c,
appendix C).  Nothing here ships to the GPU box as a dependency of tests: the
tests read the .npz vectors this harness writes into tests/golden/.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"
REPO_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
_installed = False


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], child, m)
    return m


def install(mano_model=None):
    """Inject shims; returns the imported reference modules."""
    global _installed
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if REPO_ROOT not in sys.path:
        sys.path.insert(0, REPO_ROOT)
    from oracle import lbs as o_lbs, nmr as o_nmr, sdfgrid as o_sdf, yana as o_yana
    from homan_amd.mano_assets import synthetic_mano

    mano_model = synthetic_mano(0) if mano_model is None else mano_model
    if not _installed:
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
        torch.cuda.FloatTensor = torch.FloatTensor
        torch.cuda.LongTensor = torch.LongTensor
        for name in ("trimesh", "cv2", "detectron2", "detectron2.structures", "detectron2.structures.boxes"):
            sys.modules[name] = mock.MagicMock()
        sys.modules["cv2"].resize = lambda img, size: np.zeros((size[1], size[0], 3), np.uint8)

        _module("neural_renderer", projection=o_nmr.projection, Renderer=o_nmr.Renderer)
        _module("neural_renderer.renderer", Renderer=o_nmr.Renderer)
        _module("sdf", SDF=o_sdf.SDF)
        _module("mano")

        def mano_load(model_path=None, num_pca_comps=16, use_pca=False, is_right=True, model_type="mano",
                      batch_size=1, flat_hand_mean=True, **_):
            # homan/manomodel.py:19-80 names the side through the file it asks for (is_right is True in every call there)
            from homan_amd.mano_assets import hand_models
            side = "left" if model_path is not None and "LEFT" in os.path.basename(model_path) else "right"
            return o_lbs.ManoLayer(hand_models(mano_model)[side], num_pca_comps=num_pca_comps,
                                   flat_hand_mean=flat_hand_mean, use_pca=use_pca)

        _module("mano.model", load=mano_load)
        _module("libyana")
        _module("libyana.verify")
        _module("libyana.verify.checkshape", check_shape=lambda *a, **k: None)
        _module("libyana.conversions")
        _module("libyana.conversions.npt", tensorify=o_yana.tensorify, numpify=o_yana.numpify)
        _module("libyana.camutils")
        _module("libyana.camutils.project", batch_proj2d=o_yana.batch_proj2d)
        _module("libyana.camutils.camconvs")
        _module("libyana.metrics")
        _module("libyana.metrics.iou", batch_mask_iou=o_yana.batch_mask_iou)
        _module("libyana.distutils", batch_pairwise_dist=o_yana.batch_pairwise_dist)
        _module("libyana.lib3d")
        _module("libyana.lib3d.trans3d")
        _module("libyana.lib3d.kcrop", get_K_crop_resize=o_yana.get_K_crop_resize)
        _module("libyana.visutils")
        _module("libyana.visutils.imagify")
        _module("libyana.vidutils")
        _module("libyana.vidutils.np2vid", make_video=lambda *a, **k: None)
        _module("libyana.meshutils")
        _installed = True

    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)  # homan/lossutils.py:15 loads local_data/closed_fmano.npy relative to cwd
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import homan.homan as ref_homan
        import homan.jointopt as ref_jointopt
        import homan.lossutils as ref_lossutils
    finally:
        os.chdir(cwd)
        sys.path.remove(REFERENCE_ROOT)
    # data substitution: closed-hand topology of the synthetic MANO instead of the real one
    if mano_model.get("synthetic", False):
        ref_lossutils.MANO_CLOSED_FACES = mano_model["closed_faces"].astype(np.int64)
    # the loop's visualisation needs NMR textured renders + cv2: return dummy frames
    ref_jointopt.visualize_hand_object = lambda model, images, dist=1, viz_len=7: (
        np.zeros((1, 8, 8, 3), np.uint8), np.zeros((1, 8, 8, 3), np.uint8))
    return ref_homan, ref_jointopt, ref_lossutils


def set_rend_size(size):
    """The reference hard-codes REND_SIZE=256 (homan/constants.py:32); small goldens override the
    constant that homan.losses imported (value substitution only)."""
    import homan.losses as ref_losses
    ref_losses.REND_SIZE = size


def import_pose_optimization(rend_size=None):
    """The reference's object-pose initialisation (homan/pose_optimization.py), imported in place over the same leaves."""
    install()
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import homan.pose_optimization as ref_po
    finally:
        os.chdir(cwd)
        sys.path.remove(REFERENCE_ROOT)
    if rend_size is not None:
        ref_po.REND_SIZE = rend_size        # value substitution (homan/constants.py:32 fixes 256)
    return ref_po
