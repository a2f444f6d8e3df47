"""MANO model data for the hand leaf: a chumpy-free loader for a user-supplied
``MANO_RIGHT.pkl`` and a deterministic synthetic MANO-shaped model.

The reference loads the licensed MANO pickle through the third-party ``mano``
package (reference homan/manomodel.py:19-80); neither the file nor the package
is redistributable, so tests and benchmarks run on a synthetic model with the
same tensor surface: 778 vertices, 16 joints on the MANO kinematic tree,
10 shape and 135 pose blend shapes, 45x45 PCA pose basis, 1538 open faces and
1552 closed faces (the reference's ``local_data/closed_fmano.npy`` topology:
open faces followed by a 14-triangle wrist cap).
"""
import io
import os
import pickle

import numpy as np

MANO_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int32)
NUM_VERTS = 778
NUM_JOINTS = 16
NUM_BETAS = 10
NUM_POSE_BASIS = 135

_CACHE = {}


def _fibonacci_sphere(n):
    i = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1 - 2 * i / n)
    theta = np.pi * (1 + 5 ** 0.5) * i
    return np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], 1)


def _wrist_cap(faces, verts, n_cap=14):
    """Pick `n_cap` hull faces around the -x pole forming a disc without interior vertices."""
    edge2faces = {}
    for fi, f in enumerate(faces):
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            edge2faces.setdefault((min(a, b), max(a, b)), []).append(fi)
    cent = verts[faces].mean(1)
    start = int(np.argmin(cent[:, 0]))
    chosen = [start]
    vset = set(int(v) for v in faces[start])
    while len(chosen) < n_cap:
        best, best_x = None, None
        for fi in chosen:
            f = faces[fi]
            for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
                for fj in edge2faces[(min(a, b), max(a, b))]:
                    if fj in chosen:
                        continue
                    new = [int(v) for v in faces[fj] if int(v) not in vset]
                    if len(new) != 1:
                        continue
                    if best is None or cent[fj, 0] < best_x or (cent[fj, 0] == best_x and fj < best):
                        best, best_x = fj, cent[fj, 0]
        assert best is not None, "could not grow the wrist cap"
        chosen.append(best)
        vset.update(int(v) for v in faces[best])
    return chosen


def synthetic_mano(seed=0):
    """Deterministic MANO-shaped hand model (dict of numpy arrays, fp32 / int32)."""
    key = ("synth", seed)
    if key in _CACHE:
        return _CACHE[key]
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(seed)
    semi = np.array([0.09, 0.045, 0.016])
    centre = np.array([0.02, 0.0, 0.0])
    unit = _fibonacci_sphere(NUM_VERTS)
    # mild, smooth, strictly convex deformation (superellipsoid-ish) keeps every point on the hull
    v = unit * semi + centre
    hull = ConvexHull(v)
    assert len(hull.vertices) == NUM_VERTS
    faces = hull.simplices.astype(np.int64)
    # outward orientation
    n = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    flip = (n * (v[faces].mean(1) - centre)).sum(1) < 0
    faces[flip] = faces[flip][:, ::-1]
    order = np.lexsort((faces[:, 2], faces[:, 1], faces[:, 0]))
    faces = faces[order]
    assert faces.shape[0] == 2 * NUM_VERTS - 4
    cap = _wrist_cap(faces, v)
    keep = np.ones(len(faces), bool)
    keep[cap] = False
    open_faces = faces[keep]
    closed_faces = np.concatenate([open_faces, faces[cap]], 0)

    # joints: wrist + 5 chains of 3 (MANO order index, middle, pinky, ring, thumb)
    tgt = np.zeros((NUM_JOINTS, 3))
    tgt[0] = centre + [-0.075, 0.0, 0.0]
    lateral = {0: 0.018, 1: 0.0, 2: -0.032, 3: -0.017, 4: 0.033}
    for fgr in range(5):
        for k, x in enumerate((-0.012, 0.026, 0.056)):
            tgt[1 + 3 * fgr + k] = centre + [x, lateral[fgr], 0.0]
    d2 = ((v[None] - tgt[:, None]) ** 2).sum(-1)              # (16,778)
    jreg = np.exp(-d2 / (2 * 0.014 ** 2))
    jreg /= jreg.sum(1, keepdims=True)

    # skinning weights: distance to the bone starting at each joint, top-4, renormalised
    child_dir = np.zeros((NUM_JOINTS, 3))
    for j in range(NUM_JOINTS):
        kids = [c for c in range(NUM_JOINTS) if MANO_PARENTS[c] == j]
        child_dir[j] = (tgt[kids[0]] - tgt[j]) if (len(kids) == 1) else np.array([0.03, 0, 0])
    t = np.clip(((v[None] - tgt[:, None]) * child_dir[:, None]).sum(-1) /
                (child_dir ** 2).sum(-1)[:, None], 0, 1)
    closest = tgt[:, None] + t[..., None] * child_dir[:, None]
    db = ((v[None] - closest) ** 2).sum(-1).T                  # (778,16)
    w = np.exp(-db / (2 * 0.012 ** 2)) + 1e-12
    kth = np.sort(w, 1)[:, -4][:, None]
    w = np.where(w >= kth, w, 0.0)
    w /= w.sum(1, keepdims=True)

    q = (v - centre) / semi                                     # (778,3) in [-1,1]
    basis = np.stack([np.ones(NUM_VERTS), q[:, 0], q[:, 1], q[:, 2], q[:, 0] * q[:, 1],
                      q[:, 0] ** 2, np.sin(2 * q[:, 0]), np.cos(3 * q[:, 1])], 1)  # (778,8)
    shapedirs = np.einsum("lcm,vm->vcl", rng.normal(size=(NUM_BETAS, 3, 8)), basis) * 1.5e-3
    posedirs = np.einsum("kcm,vm->kvc", rng.normal(size=(NUM_POSE_BASIS, 3, 8)), basis) * 6e-4
    posedirs = posedirs.reshape(NUM_POSE_BASIS, NUM_VERTS * 3)
    comps, _ = np.linalg.qr(rng.normal(size=(45, 45)))
    hand_mean = rng.normal(size=45) * 0.1

    out = dict(
        v_template=v.astype(np.float32),
        shapedirs=shapedirs.astype(np.float32),
        posedirs=posedirs.astype(np.float32),
        J_regressor=jreg.astype(np.float32),
        lbs_weights=w.astype(np.float32),
        parents=MANO_PARENTS.copy(),
        hand_components=comps.astype(np.float32),
        hand_mean=hand_mean.astype(np.float32),
        faces=open_faces.astype(np.int32),
        closed_faces=closed_faces.astype(np.int32),
        synthetic=True,
    )
    _CACHE[key] = out
    return out


def kernel_layout(model_np, flat_hand_mean=False):
    """The model arrays in the layout csrc/mano.hip reads (model DATA, shared by the kernels' host side and the CPU oracle's
    written-out forward): v_template (778,3), M (145,2334) = [posedirs ; shapedirs^T], the joint regressor folded into
    J_template (16,3) + J_shapedirs (16,3,10) - formed in DOUBLE and rounded once, so that the floats do not depend on the
    host's BLAS -, lbs weights (778,16), the first 16 PCA components (16,45), the mean pose (45), parents (16) int32."""
    vt = np.asarray(model_np["v_template"], np.float32)
    assert vt.shape == (778, 3)
    sd = np.asarray(model_np["shapedirs"], np.float32)              # (778,3,10)
    pd = np.asarray(model_np["posedirs"], np.float32)               # (135, 2334)
    M = np.concatenate([pd, sd.reshape(778 * 3, 10).T], 0)          # (145, 2334)
    jr = np.asarray(model_np["J_regressor"], np.float64)            # (16,778)
    J_t = (jr @ vt.astype(np.float64)).astype(np.float32)           # (16,3)
    J_s = np.einsum("jv,vcl->jcl", jr, sd.astype(np.float64)).astype(np.float32)      # (16,3,10)
    hm = np.asarray(model_np["hand_mean"], np.float32)
    hand_mean = np.zeros_like(hm) if flat_hand_mean else hm
    host = [vt, M, J_t, J_s, np.asarray(model_np["lbs_weights"], np.float32),
            np.asarray(model_np["hand_components"][:16], np.float32), hand_mean, np.asarray(model_np["parents"], np.int32)]
    return [np.ascontiguousarray(a) for a in host]


def mirrored(model):
    """The other hand of a MANO-shaped model: every geometric quantity reflected in y (the lateral axis of the synthetic
    hand), face windings reversed so that normals stay outward; pose basis and mean pose kept.  Stands in for
    MANO_LEFT.pkl next to a synthetic right hand (the real left model is a separate licensed file, read by `get_mano`)."""
    key = ("mirror", id(model))
    if key in _CACHE:
        return _CACHE[key][1]
    flip = np.array([1.0, -1.0, 1.0], np.float32)
    out = dict(model)
    out["v_template"] = (model["v_template"] * flip).astype(np.float32)
    out["shapedirs"] = (model["shapedirs"] * flip[None, :, None]).astype(np.float32)
    out["posedirs"] = (model["posedirs"].reshape(NUM_POSE_BASIS, NUM_VERTS, 3) * flip).reshape(NUM_POSE_BASIS, -1).astype(np.float32)
    out["faces"] = np.ascontiguousarray(model["faces"][:, ::-1])
    if model.get("closed_faces") is not None:
        out["closed_faces"] = np.ascontiguousarray(model["closed_faces"][:, ::-1])
    out["side"] = "left" if model.get("side", "right") == "right" else "right"
    _CACHE[key] = (model, out)          # (keeps `model` alive: the key is its id)
    return out


def hand_models(mano_model=None, mano_root="extra_data/mano"):
    """{"right": model, "left": model} from whatever the caller has: a {"right", "left"} dictionary, one (right-hand) model
    - its mirror image then serves as the left hand -, or nothing (the files under `mano_root`, else synthetic)."""
    if isinstance(mano_model, dict) and "right" in mano_model and "v_template" not in mano_model:
        right = mano_model["right"]
        return {"right": right, "left": mano_model.get("left") or mirrored(right)}
    if mano_model is not None:
        return {"right": mano_model, "left": mirrored(mano_model)}
    right = get_mano(mano_root, "right")
    left_file = os.path.join(mano_root, "MANO_LEFT.pkl")
    return {"right": right, "left": get_mano(mano_root, "left") if os.path.exists(left_file) else mirrored(right)}


class _Stub:
    """Stand-in for chumpy objects inside the official MANO pickles."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


class _ManoUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] in ("chumpy",):
            return _Stub
        return super().find_class(module, name)


def _arr(x):
    if isinstance(x, _Stub):
        for k in ("x", "r", "a"):
            if k in x.__dict__:
                return _arr(x.__dict__[k])
        raise ValueError("unrecognised chumpy object in MANO pickle")
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    return np.asarray(x)


def load_mano_pkl(path):
    """Read an official ``MANO_{RIGHT,LEFT}.pkl`` without chumpy.

    Field mapping follows the smplx-style loader the reference's ``mano``
    dependency uses (reference homan/manomodel.py:19-80 call sites):
    posedirs (778,3,135) -> (135, 2334); J_regressor sparse -> dense;
    kintree_table[0] -> parents; hands_components / hands_mean.
    """
    with open(path, "rb") as fh:
        data = _ManoUnpickler(io.BytesIO(fh.read()), encoding="latin1").load()
    posedirs = _arr(data["posedirs"]).astype(np.float32)
    parents = _arr(data["kintree_table"])[0].astype(np.int64).copy()
    parents[0] = -1
    faces = _arr(data["f"]).astype(np.int32)
    out = dict(
        v_template=_arr(data["v_template"]).astype(np.float32),
        shapedirs=_arr(data["shapedirs"]).astype(np.float32)[:, :, :NUM_BETAS],
        posedirs=posedirs.reshape(-1, posedirs.shape[-1]).T.copy(),
        J_regressor=_arr(data["J_regressor"]).astype(np.float32),
        lbs_weights=_arr(data["weights"]).astype(np.float32),
        parents=parents.astype(np.int32),
        hand_components=_arr(data["hands_components"]).astype(np.float32),
        hand_mean=_arr(data["hands_mean"]).astype(np.float32),
        faces=faces,
        closed_faces=None,
        synthetic=False,
    )
    return out


def get_mano(mano_root="extra_data/mano", side="right", closed_faces_path="local_data/closed_fmano.npy"):
    """Real MANO if ``<mano_root>/MANO_RIGHT.pkl`` exists (reference layout), else synthetic."""
    fname = os.path.join(mano_root, "MANO_RIGHT.pkl" if side == "right" else "MANO_LEFT.pkl")
    key = ("real", os.path.abspath(fname))
    if os.path.exists(fname):
        if key not in _CACHE:
            m = load_mano_pkl(fname)
            if os.path.exists(closed_faces_path):
                m["closed_faces"] = np.load(closed_faces_path).astype(np.int32)
            else:
                raise FileNotFoundError(
                    f"{closed_faces_path} (closed MANO faces, reference homan/lossutils.py:15) not found")
            _CACHE[key] = m
        return _CACHE[key]
    return synthetic_mano(0)
