import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, time
from homan_amd import ops, synth
from homan_amd.mano_assets import synthetic_mano
m = synthetic_mano(0); B=30
g = torch.Generator().manual_seed(0)
vh = (torch.from_numpy(m["v_template"])[None].repeat(B,1,1) + torch.tensor([0.,0.,0.55])).cuda()
ov, of = synth.bottle_mesh()
vo = (torch.from_numpy(ov)[None].repeat(B,1,1) + torch.tensor([0.02,0.,0.56])).cuda()
rws = ops.ReduceWorkspace("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
idx, d2, met = ops.nearest_vertices(vh, vo, rws)
print("nn us", timeit(lambda: ops.nearest_vertices(vh, vo, rws)))
print("contact us", timeit(lambda: ops.contact_loss(vh, vo, idx, rws)))
print("smooth us", timeit(lambda: ops.smooth_loss(vo, 1, rws)))
