import copy, os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from homan_amd import lib as hlib, synth
from homan_amd.jointopt import build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
B,S,F,V = 30,256,3000,1502
sctx = model.losses.sil_ctx
verts = model.get_verts_object()[0].detach().contiguous()
pooled = torch.empty(B,S,S,device="cuda"); out2 = torch.empty(2,device="cuda")
for libname in sys.argv[1:]:
    L = ctypes.CDLL(libname)
    fn = L.hm_bench_raster_fwd; fn.restype = ctypes.c_int
    fn.argtypes = hlib._SIGNATURES["hm_bench_raster_fwd"][1]
    ms = torch.zeros(1)
    rc = fn(hlib.ptr(verts), hlib.ptr(sctx.faces), hlib.ptr(model.camintr_rois_object), B,V,F,S, hlib.ptr(model.keep_mask_object), hlib.ptr(model.ref_mask_object), hlib.ptr(model.losses.keep_sum), hlib.ptr(pooled), hlib.ptr(out2), hlib.ptr(sctx.region_order), hlib.ptr(sctx.workspace), 20, ms.data_ptr(), hlib.stream())
    print(os.path.basename(libname), rc, "avg us", ms.item()*1e3, "alpha frac", (pooled>0).float().mean().item())
a = ctypes.c_int(0); b2 = ctypes.c_int(0)
print("occupancy rc", L.hm_debug_occupancy(ctypes.byref(a), ctypes.byref(b2)), "raster blocks/CU", a.value, "sweep blocks/CU", b2.value)
p = torch.cuda.get_device_properties(0); print(p.name, "shared per block", p.shared_memory_per_block, "per SM", getattr(p, "shared_memory_per_multiprocessor", None))
import numpy as np
part = torch.empty(B, 1024, 4, device="cuda")
L.hm_debug_read_partials.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
print(L.hm_debug_read_partials(hlib.ptr(sctx.workspace), B, V, F, S, hlib.ptr(part), hlib.stream()))
torch.cuda.synchronize()
d = part[..., 3].cpu().numpy().ravel(); st = part[..., 2].cpu().numpy().ravel()
print("wave cycles: mean", d.mean(), "median", np.median(d), "p90", np.percentile(d, 90), "p99", np.percentile(d, 99), "max", d.max(), "sum", d.sum())
st = st - st.min(); 
print("start span (cycles)", st.max(), "end span", (st + d).max())
order = np.argsort(st)
# concurrency over time
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([st + d, -np.ones_like(st)], 1)])
ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1]); T = ev[:, 0]
for frac in (0.02, 0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    i = np.searchsorted(T, frac * T.max()); print("t=%.2f (%.0f kcyc) concurrency %d" % (frac, frac*T.max()/1e3, conc[min(i, len(conc) - 1)]))
fr = np.repeat(np.arange(B), 1024)
for bb in (0, 10, 20, 29):
    m = fr == bb; print("frame", bb, "start min/max kcyc", st[m].min()/1e3, st[m].max()/1e3, "end max", (st[m]+d[m]).max()/1e3)
dm = part[0, :, 3].cpu().numpy().reshape(32, 32)
np.set_printoptions(linewidth=250, threshold=100000)
print((dm / 1000).astype(int))
