"""Shared helpers for the test-suite (golden loading, oracle-backed clip generation)."""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """Joint-optimisation goldens (the pose-initialisation golden has its own schema, tests/test_poseinit.py)."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "ref_*.npz")))
    return [n for n in names if not n.startswith("ref_poseinit")]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    rec = {k: z[k] for k in z.files}
    inputs = {k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("in_") and k != "in_camintr"}
    weights = {"lw_" + k[3:]: float(v) for k, v in rec.items() if k.startswith("lw_")}
    meta = dict(image_size=int(rec["meta_image_size"]), steps=int(rec["meta_steps"]),
                optimize_object_scale=bool(rec["meta_optimize_object_scale"]),
                optimize_mano=bool(rec["meta_optimize_mano"]), lr=float(rec["meta_lr"]),
                hand_sides=[str(x) for x in rec["meta_hand_sides"]] if "meta_hand_sides" in rec else ["right"],
                inter_type=str(rec["meta_inter_type"]) if "meta_inter_type" in rec else "centroid",
                has_trajectory="evo_loss" in rec)
    return rec, inputs, rec["in_camintr"], weights, meta


def model_kwargs(inputs, camintr, meta):
    kw = dict(inputs)
    kw.update(hand_sides=list(meta["hand_sides"]), camintr=camintr, class_name="default", int_scale_init=1,
              hand_proj_mode="persp", optimize_mano=meta["optimize_mano"], optimize_mano_beta=True,
              optimize_object_scale=meta["optimize_object_scale"], image_size=meta["image_size"],
              inter_type=meta["inter_type"])
    return kw


def oracle_clip_fns(mano_model):
    """(silhouette_fn, hand_verts_fn) backed by the CPU oracle, for homan_amd.synth.make_clip."""
    from homan_amd.mano_assets import hand_models
    from oracle import lbs, nmr
    layers = {side: lbs.ManoLayer(m, num_pca_comps=16, flat_hand_mean=False) for side, m in hand_models(mano_model).items()}

    def hand_fn(pca, rot, betas, side="right"):
        layer = layers[side]
        hp = pca[:, :16] @ layer.hand_components
        if side == "left":          # homan/manomodel.py:131-132, applied before the mean pose is added
            hp = hp.clone()
            hp[:, 1::3] *= -1
            hp[:, 2::3] *= -1
        return layer(betas=betas, global_orient=rot, hand_pose=hp, transl=torch.zeros(len(rot), 3))[0]

    def sil_fn(verts, faces, K, size):
        r = nmr.Renderer(image_size=size, K=K, R=torch.eye(3)[None], t=torch.zeros(1, 3), orig_size=1)
        return r(verts, faces, mode="silhouettes")

    return sil_fn, hand_fn
