"""CPU-only checks: the C-ABI library loads and exports every symbol include/homan_amd.h declares, the ctypes table
matches the header, and the host-side logic (collation, Adam groups, assets, generator, adjacency, sharding)."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "homan_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from homan_amd import build, lib
    assert os.path.exists(build.LIB_PATH), "run `python -m homan_amd.build` (or __graft_entry__.build())"
    handle = lib.lib()          # loads after torch so that the HIP runtime in the process is reused
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in include/homan_amd.h but not exported"
    assert sorted(lib.exported_symbols()) == declared, set(lib.exported_symbols()) ^ set(declared)


def test_ctypes_arity_matches_header():
    from homan_amd import lib
    text = open(os.path.join(ROOT, "include", "homan_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, (res, args) in lib._SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(args), (name, n, len(args))


def test_sil_render_struct_matches_header_and_library():
    """HmSilRender (hm_sil_fwd_multi): the ctypes Structure has the header's fields in the header's order and the size the
    library was built with (hm_sil_render_bytes: no GPU needed)."""
    from homan_amd import lib
    text = open(os.path.join(ROOT, "include", "homan_amd.h")).read()
    body = re.search(r"typedef struct HmSilRender \{(.*?)\} HmSilRender;", text, flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        kind = "p" if "*" in decl else ("f" if decl.startswith("float") else "i")
        names = re.sub(r"^(const\s+)?(float|int|void)\s*\**", "", decl)
        fields += [(n.strip().lstrip("*").strip(), kind) for n in names.split(",")]
    want = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float}
    assert [(n, want[k]) for n, k in fields] == list(lib.SilRender._fields_)
    assert lib.lib().hm_sil_render_bytes() == ctypes.sizeof(lib.SilRender)


def test_no_cpu_fallback_in_product():
    """homan_amd must not import the oracle, and HOMan refuses to build without a GPU."""
    import homan_amd
    for root, _, files in os.walk(os.path.dirname(homan_amd.__file__)):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
    if not torch.cuda.is_available():
        from tests import util
        rec, inputs, camintr, weights, meta = util.load_golden(util.golden_names()[0])
        with pytest.raises(RuntimeError):
            homan_amd.HOMan(**util.model_kwargs(inputs, camintr, meta))


def test_synthetic_mano_topology(mano_model):
    f, fc = mano_model["faces"], mano_model["closed_faces"]
    assert f.shape == (1538, 3) and fc.shape == (1552, 3) and mano_model["v_template"].shape == (778, 3)
    np.testing.assert_array_equal(fc[:1538], f)        # closed = open + 14-triangle wrist cap
    edges = {}
    for a, b, c in fc:
        for e in ((a, b), (b, c), (c, a)):
            k = (min(e), max(e))
            edges[k] = edges.get(k, 0) + 1
    assert set(edges.values()) == {2}                  # watertight
    assert len(edges) == 3 * 1552 // 2
    np.testing.assert_allclose(mano_model["lbs_weights"].sum(1), 1.0, atol=1e-5)
    np.testing.assert_allclose(mano_model["J_regressor"].sum(1), 1.0, atol=1e-5)
    assert list(mano_model["parents"]) == [-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14]


def test_mano_pkl_loader_without_chumpy(tmp_path, mano_model):
    """An official-style pickle (chumpy objects, sparse regressor) loads through the stub unpickler."""
    import sys
    import types
    import scipy.sparse as sp
    from homan_amd import mano_assets
    chumpy = types.ModuleType("chumpy")
    ch = types.ModuleType("chumpy.ch")

    # what chumpy.ch.Ch pickles to: an object whose state carries the array in 'x'
    Ch = type("Ch", (), {"__init__": lambda self, x: setattr(self, "x", x), "__module__": "chumpy.ch",
                         "__qualname__": "Ch"})
    ch.Ch = Ch
    chumpy.ch = ch
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = chumpy, ch
    try:
        m = mano_model
        data = dict(v_template=m["v_template"], shapedirs=Ch(m["shapedirs"]),
                    posedirs=m["posedirs"].T.reshape(778, 3, 135), J_regressor=sp.csc_matrix(m["J_regressor"]),
                    weights=m["lbs_weights"], kintree_table=np.stack([np.where(m["parents"] < 0, 2 ** 32 - 1, m["parents"]),
                                                                       np.arange(16)]).astype(np.uint32),
                    hands_components=m["hand_components"], hands_mean=m["hand_mean"], f=m["faces"].astype(np.uint32))
        path = tmp_path / "MANO_RIGHT.pkl"
        with open(path, "wb") as fh:
            pickle.dump(data, fh, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    out = mano_assets.load_mano_pkl(str(path))
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "hand_components", "hand_mean"):
        np.testing.assert_allclose(out[k], m[k], atol=0, err_msg=k)
    assert list(out["parents"]) == list(m["parents"])


def test_adjacency_csr():
    from homan_amd.ops import build_adjacency
    faces = np.array([[0, 1, 2], [2, 1, 3], [3, 1, 0]])
    off, items = build_adjacency(faces, 5)
    off, items = off.numpy(), items.numpy()
    assert off.tolist() == [0, 2, 5, 7, 9, 9]
    for v in range(5):
        got = sorted(items[off[v]:off[v + 1]].tolist())
        want = sorted(f * 3 + c for f in range(3) for c in range(3) if faces[f, c] == v)
        assert got == want


def test_parameter_groups_follow_reference_name_rules():
    """reference homan/jointopt.py:128-151: rigid = no 'mano' and no 'rotation'; [pca, betas]; 'rotation' w/o 'mano'."""
    from homan_amd.jointopt import parameter_groups

    class Dummy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            for n in ("translations_object", "rotations_object", "translations_hand", "rotations_hand", "cams_hand",
                      "mano_pca_pose", "mano_rot", "mano_trans", "mano_betas", "int_scales_object"):
                setattr(self, n, torch.nn.Parameter(torch.zeros(2)))
    m = Dummy()
    groups = parameter_groups(m, 1e-2)
    names = {id(p): n for n, p in m.named_parameters()}
    got = [sorted(names[id(p)] for p in g["params"]) for g in groups]
    assert got[0] == ["cams_hand", "int_scales_object", "translations_hand", "translations_object"]
    assert got[1] == ["mano_betas", "mano_pca_pose"]
    assert got[2] == ["rotations_hand", "rotations_object"]
    assert [g["lr"] for g in groups] == [1e-2, 1e-1, 1e-1]
    stepped = {n for g in got for n in g}
    assert "mano_rot" not in stepped and "mano_trans" not in stepped       # reference quirk: never stepped


def test_collate_inputs_matches_oracle_collation(mano_model):
    from homan_amd import synth
    from homan_amd.jointopt import collate_inputs
    from oracle.jointopt import collate_inputs as oracle_collate
    from tests import util
    sil_fn, hand_fn = util.oracle_clip_fns(mano_model)
    clip = synth.make_clip(seed=5, frames=3, rend_size=32, image_size=32, obj="cube", silhouette_fn=sil_fn,
                           hand_verts_fn=hand_fn)
    a = collate_inputs(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    b = oracle_collate(clip["person_parameters"], clip["object_parameters"], clip["objvertices"], clip["objfaces"])
    assert sorted(a) == sorted(b)
    for k in a:
        if isinstance(a[k], torch.Tensor):
            assert torch.equal(a[k], b[k]), k
    assert a["translations_object"].shape == (3, 1, 3) and a["rotations_hand"].shape == (3, 3, 3)
    assert a["camintr_rois_object"].shape == (3, 3, 3) and a["faces_hand"].shape == (1, 1538, 3)
    tm = a["target_masks_object"]
    assert set(np.unique(tm.numpy()).tolist()) <= {-1.0, 0.0, 1.0}
    # determinism of the generator
    clip2 = synth.make_clip(seed=5, frames=3, rend_size=32, image_size=32, obj="cube", silhouette_fn=sil_fn,
                            hand_verts_fn=hand_fn)
    assert torch.equal(clip["object_parameters"][1]["target_masks"], clip2["object_parameters"][1]["target_masks"])


def test_meshes_are_watertight_with_requested_sizes():
    from homan_amd import synth
    for verts, faces, nf in (synth.box_mesh() + (500,), synth.bottle_mesh() + (3000,)):
        assert faces.shape == (nf, 3)
        cnt = {}
        for a, b, c in faces:
            for e in ((a, b), (b, c), (c, a)):
                assert e not in cnt                      # consistent orientation: each directed edge once
                cnt[e] = 1
        for (a, b) in cnt:
            assert (b, a) in cnt                         # ... and its twin exists: closed surface
        np.testing.assert_allclose(np.linalg.norm(verts, axis=1).max() * 2, 0.08 if nf == 500 else 0.2, rtol=1e-5)


def test_shard_clips():
    from homan_amd.dist import shard_clips
    shards = [shard_clips(64, r, 8) for r in range(8)]
    assert all(len(s) == 8 for s in shards) and sorted(sum(shards, [])) == list(range(64))
    shards = [shard_clips(10, r, 4) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(10)) and max(len(s) for s in shards) == 3


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    """bench.py --gpus N under a launcher that started another number of ranks must not report N (the flag was
    parsed and ignored).  Checked before anything touches the GPU."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_line_is_compact_and_keeps_the_contract():
    """The driver parses the LAST stdout line of bench.py; round 4's 23 KB line (per-step parity traces inside) came back as
    `parsed: null`.  Build the line from a canned full record - the round-4 one, traces and all - and from a synthetic worst
    case, and hold the bound and the contract's keys."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_cfg2_driver_flags.json")))
    assert len(json.dumps(full)) > 20000                       # (the record that did not parse)
    worst = dict(full, per_rank_its=[6249.123456789] * 8, ranks=8, backend="nccl", parity_ok=True,
                 parity_vs="oracle's reproducible (written-out) loop, end state",
                 config=dict(full["config"], workload="x" * 5000, nested={"a": [1] * 1000}),
                 cfg2_depth=dict(value=3716.123456, unit="it/s", ms_per_step=0.269123456, dominant_kernel_us=54.123456, note="y" * 500),
                 cfg3=dict(value=5457.123456, unit="it/s", ms_per_step=0.183123456, dominant_kernel_us=46.123456, kernels_us={"a": 1.0}))
    legs = bench.compact_line(dict(full, cfg2_depth=worst["cfg2_depth"], cfg3=worst["cfg3"]))
    assert set(legs["cfg2_depth"]) == {"value", "ms_per_step", "dominant_kernel_us"} and "cfg3" in legs      # one number each
    for rec in (full, worst):
        line = bench.compact_line(rec)
        s = json.dumps(line)
        assert len(s) <= bench.MAX_LINE_BYTES < 4096, len(s)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in line, k
        assert isinstance(line["config"]["workload"], str)
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
            assert k in line["roofline"], k
        assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-4
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in line["cpu_baseline"], k
        assert "final_loss_parity" not in line and "end_to_end" not in line and "kernels" not in line["roofline"]
        assert json.loads(s) == line


def test_launch_hints_are_per_thread_and_profiling_globals_are_not_in_the_release_build():
    """include/homan_amd.h: the only state outside the caller's buffers are the hm_tune_* launch hints, and those are
    thread-local - a thread that sets one does not change what another thread's launches read (they were
    process-wide statics).  And the device-side profiling arrays of the instrumented builds (-DRASTER_PHASES, -DSWEEP_STATS,
    -DSWEEP_UNIT_PROFILE, -DHM_CHAIN_STAMPS) are not part of the shipped library."""
    import threading
    from homan_amd import build, lib
    L = lib.lib()
    default = L.hm_tune_sweep_blocks(0)                     # (<= 0: query)
    seen = {}
    try:
        assert L.hm_tune_sweep_blocks(768) == default and L.hm_tune_sweep_blocks(0) == 768

        def other():
            seen["start"] = L.hm_tune_sweep_blocks(0)       # a fresh thread reads the default, not this thread's 768
            L.hm_tune_sweep_blocks(512)
            seen["own"] = L.hm_tune_sweep_blocks(0)
            seen["pad"] = L.hm_tune_raster_lds_pad(4096)
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert seen == {"start": default, "own": 512, "pad": 0}
        assert L.hm_tune_sweep_blocks(0) == 768 and L.hm_tune_raster_lds_pad(-1) == 0
    finally:
        L.hm_tune_sweep_blocks(default)
    blob = open(build.LIB_PATH, "rb").read()
    for name in (b"g_raster_ph", b"g_sweep_n", b"g_unit_prof", b"g_chain_ts"):
        assert name not in blob, name
    # timing-only kernel variants ("results are wrong": ceiling builds of the sweeps) live in csrc/raster_experiments.h, which
    # refuses to compile without -DHM_EXPERIMENT; the release build never defines it and no kernel source spells a variant out
    assert not any("HM_EXPERIMENT" in f or "SWEEP_EXP" in f for f in build.FLAGS)
    exp = open(os.path.join(build.CSRC, "raster_experiments.h")).read()
    assert "#ifndef HM_EXPERIMENT\n#error" in exp
    for path in build.sources() + [os.path.join(build.CSRC, h) for h in os.listdir(build.CSRC) if h.endswith(".h")]:
        if os.path.basename(path) in ("raster_experiments.h", "raster_hooks.h"):
            continue
        src = open(path).read()
        for word in ("SWEEP_EXP", "SWEEP_WGDYN", "results are wrong", "raster_experiments.h"):
            assert word not in src, (os.path.basename(path), word)
