#!/usr/bin/env python
"""Probe (round 6): what would ONE raster launch over the three renders of cfg2 + ordinal depth cost?
The three renders of an iteration (object silhouette at the ROI camera, object depth at the full-image camera, hand depth at the
full-image camera) are today three setup + raster launch pairs on their own workspaces.  This tool takes the steady state of a
cfg2 + depth fit and times, alone on the GPU, with HIP events on the launch stream:
  separate   the three hm_sil_fwd calls one after the other (as in the fused loop);
  merged60   the two OBJECT renders as one call over 60 frames (same mesh, per-frame K);
  merged90   all three as one call over 90 frames through the EXISTING entry point: the hand's frames carry their own faces
             (faces_bstride = 3F) padded with degenerate triangles, vertices padded to the object's count.
Only a timing probe: the merged calls also compute the fused loss terms for the depth frames (zero masks).
Usage: python tools/merged_raster_probe.py [reps]"""
import copy
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from homan_amd import lib as hlib  # noqa: E402
from homan_amd import ops, synth  # noqa: E402
from homan_amd.fused import FusedStepper  # noqa: E402
from homan_amd.jointopt import build_model  # noqa: E402
from homan_amd.mano_assets import synthetic_mano  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, S = 30, 256
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=B, rend_size=S, image_size=S, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]),
                    objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"],
                    optimize_mano=True, image_size=S, mano_model=mano, rend_size=S, sync_metrics=False, ordinal_depth=True)
lw = dict(synth.STEP1_LOSS_WEIGHTS)
lw["lw_depth"] = 1.0
st = FusedStepper(model, lw, 1e-2, 500)
st.run(400)
torch.cuda.synchronize()
L, P, ck = hlib.lib(), hlib.ptr, hlib.check
dev = st.vo.device
m = st.model.models[0] if hasattr(st.model, "models") else st.model
vo, vh = st.vo.clone(), st.vh.clone()
Vo, Vh = vo.shape[1], vh.shape[1]
K_sil, K_full = st.sil_K.clone(), m.camintr.clone().float().contiguous()
keep, ref = st.sil_keep.clone(), st.sil_ref.clone()
faces_o = m.faces_object[0].to(torch.int32).contiguous()
faces_h = m.faces_hand[0].to(torch.int32).contiguous()
Fo, Fh = faces_o.shape[0], faces_h.shape[0]
print(f"Vo {Vo} Fo {Fo} Vh {Vh} Fh {Fh}", file=sys.stderr)
stream = torch.cuda.current_stream().cuda_stream


def ctx_for(faces, V, nb):
    return ops.SilhouetteContext(faces[None].expand(nb, -1, -1), V, nb, S, dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def fwd(verts, faces, bstride, K, nb, V, F, keep_, ref_, pooled, depth, ctx):
    ck(L.hm_sil_fwd(P(verts), P(faces), bstride, P(K), nb, V, F, S, 1.0, ops.NMR_NEAR, ops.NMR_FAR,
                    P(keep_) if keep_ is not None else None, P(ref_) if ref_ is not None else None, None, P(pooled), None,
                    P(ctx.work_order), P(depth) if depth is not None else None, None, 0, None, None, None, 0, 1,
                    P(ctx.workspace), stream), "hm_sil_fwd")


out = {}
# ---- separate
c_sil, c_do, c_dh = ctx_for(faces_o, Vo, B), ctx_for(faces_o, Vo, B), ctx_for(faces_h, Vh, B)
p_sil, p_do, d_do, p_dh, d_dh = (torch.empty(B, S, S, device=dev) for _ in range(5))
f_sil = lambda: fwd(vo, faces_o, 0, K_sil, B, Vo, Fo, keep, ref, p_sil, None, c_sil)          # noqa: E731
f_do = lambda: fwd(vo, faces_o, 0, K_full, B, Vo, Fo, None, None, p_do, d_do, c_do)           # noqa: E731
f_dh = lambda: fwd(vh, faces_h, 0, K_full, B, Vh, Fh, None, None, p_dh, d_dh, c_dh)           # noqa: E731
out["sil_alone_us"] = timed(f_sil)
out["obj_depth_alone_us"] = timed(f_do)
out["hand_depth_alone_us"] = timed(f_dh)
out["separate_us"] = timed(lambda: (f_sil(), f_do(), f_dh()))
# ---- merged60: both object renders
c60 = ctx_for(faces_o, Vo, 2 * B)
v60 = torch.cat([vo, vo]).contiguous()
K60 = torch.cat([K_sil, K_full]).contiguous()
z = torch.zeros_like(keep)
keep60, ref60 = torch.cat([keep, z]).contiguous(), torch.cat([ref, z]).contiguous()
p60, d60 = torch.empty(2 * B, S, S, device=dev), torch.empty(2 * B, S, S, device=dev)
out["merged60_us"] = timed(lambda: fwd(v60, faces_o, 0, K60, 2 * B, Vo, Fo, keep60, ref60, p60, d60, c60))
out["merged60_plus_hand_us"] = timed(lambda: (fwd(v60, faces_o, 0, K60, 2 * B, Vo, Fo, keep60, ref60, p60, d60, c60), f_dh()))
# ---- merged90: per-frame faces, hand padded
c90 = ctx_for(faces_o, Vo, 3 * B)
vh_pad = torch.zeros(B, Vo, 3, device=dev)
vh_pad[:, :Vh] = vh
vh_pad[:, Vh:] = vh[:, :1]
v90 = torch.cat([vo, vo, vh_pad]).contiguous()
fh_pad = torch.zeros(Fo, 3, dtype=torch.int32, device=dev)
fh_pad[:Fh] = faces_h
f90 = torch.cat([faces_o[None].expand(2 * B, -1, -1), fh_pad[None].expand(B, -1, -1)]).contiguous()
K90 = torch.cat([K_sil, K_full, K_full]).contiguous()
keep90, ref90 = torch.cat([keep, z, z]).contiguous(), torch.cat([ref, z, z]).contiguous()
p90, d90 = torch.empty(3 * B, S, S, device=dev), torch.empty(3 * B, S, S, device=dev)
out["merged90_us"] = timed(lambda: fwd(v90, f90, 3 * Fo, K90, 3 * B, Vo, Fo, keep90, ref90, p90, d90, c90))
# sanity: the merged renders show what the separate ones show
torch.cuda.synchronize()
out["pooled_equal"] = bool(torch.equal(p90[:B], p_sil) and torch.equal(p90[B:2 * B], p_do) and torch.equal(p90[2 * B:], p_dh))
out["depth_equal"] = bool(torch.equal(d90[B:2 * B], d_do) and torch.equal(d90[2 * B:], d_dh))
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "merged_raster_probe.json"), "w") as fh:
    json.dump(out, fh, indent=1)
