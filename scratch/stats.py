import copy, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from homan_amd import synth
from homan_amd.jointopt import build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
clip = synth.make_clip(seed=0, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
model = build_model(copy.deepcopy(clip["person_parameters"]), copy.deepcopy(clip["object_parameters"]), objvertices=clip["objvertices"], objfaces=clip["objfaces"], camintr=clip["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False)
vo,_ = model.get_verts_object(); model.losses.compute_sil_loss_object(vo)
f9 = model.losses.sil_ctx.faces9().cpu().numpy()   # (B,F,9)
is_=512
px = 0.5*(f9[...,0::3]*is_+is_-1); py = 0.5*(f9[...,1::3]*is_+is_-1)
x0=np.clip(np.floor(px.min(-1))-1,0,is_-1); x1=np.clip(np.ceil(px.max(-1))+1,0,is_-1)
y0=np.clip(np.floor(py.min(-1))-1,0,is_-1); y1=np.clip(np.ceil(py.max(-1))+1,0,is_-1)
print("bbox w mean/max", (x1-x0).mean(), (x1-x0).max(), "h", (y1-y0).mean(), (y1-y0).max())
b=0
cnt=np.zeros((32,32),int)
for f in range(f9.shape[1]):
    cnt[int(y0[b,f])//16:int(y1[b,f])//16+1, int(x0[b,f])//16:int(x1[b,f])//16+1]+=1
print("per-tile candidates frame0: mean over nonzero", cnt[cnt>0].mean(), "max", cnt.max(), "nonzero tiles", (cnt>0).sum(), "sum", cnt.sum())
print(np.sort(cnt.ravel())[-20:])
