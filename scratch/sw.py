import copy, os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from homan_amd import lib as hlib, synth
from homan_amd.jointopt import build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
B,F = 30,3000
dbg = torch.zeros(B*F, 3, device="cuda")
L = hlib.lib()
L.hm_debug_set_sweep_buffer.argtypes=[ctypes.c_void_p]; L.hm_debug_set_sweep_buffer(dbg.data_ptr())
for _ in range(2):
    vo,_ = model.get_verts_object(); l,_m = model.losses.compute_sil_loss_object(vo); l["loss_sil_obj"].sum().backward()
torch.cuda.synchronize()
d = dbg.cpu().numpy(); st, du, it = d[:,0], d[:,1], d[:,2]
st = st - st.min()
print("waves", len(du), "dur ticks(10ns): mean", du.mean(), "median", np.median(du), "p99", np.percentile(du,99), "max", du.max(), "sum", du.sum())
print("items: mean", it.mean(), "max", it.max(), "p99", np.percentile(it, 99))
print("span", (st+du).max())
ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([st + du, -np.ones_like(st)], 1)]); ev = ev[np.argsort(ev[:, 0])]
conc = np.cumsum(ev[:, 1]); T = ev[:, 0]
for frac in (0.05, 0.2, 0.4, 0.6, 0.8, 0.9, 0.95):
    i = np.searchsorted(T, frac * T.max()); print("t=%.2f (%.0f) concurrency %d" % (frac, frac*T.max(), conc[min(i, len(conc) - 1)]))
slow = np.argsort(du)[-8:]; print("slowest faces (b,f,dur,heavy,bits):", [(int(i//F), int(i%F), int(du[i]), int(it[i]), int(d[i,0])) for i in slow]); print("total bits", d[:,0].sum(), "total heavy items", it.sum(), "faces with heavy", (it>0).sum())
sil = model.losses.last_silhouettes; keep = model.keep_mask_object; ref = model.ref_mask_object
neg = ((keep == 1) & (ref == 1) & (sil < 1)).float() * (1 - sil) * 4
pos = ((keep == 1) & (ref == 0) & (sil > 0)).float() * sil * 4
print("neg samples per frame", neg.sum((1, 2)).cpu().numpy()[:6], "pos", pos.sum((1, 2)).cpu().numpy()[:6], "covered", (sil * 4).sum((1, 2)).cpu().numpy()[:3])
print("iou", _m)
