"""Load balance of the three heavy silhouette kernels INSIDE the replayed graph: raw in-graph stamps (hm_sil_timestamps) of
every sweep wave and every raster / lines workgroup at chosen iterations of a cfg2 fit -> span, busy-time percentiles,
concurrency over time.  GPU box.  usage: python tools/sil_wave_balance.py [--at 5 12 200] [--clips 1]"""
import argparse, copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser(); ap.add_argument("--at", type=int, nargs="+", default=[5, 12, 25, 200]); ap.add_argument("--clips", type=int, default=1)
args = ap.parse_args()
import torch, numpy as np
from homan_amd import lib as hlib, synth
from homan_amd.jointopt import FusedStepper, build_model
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
sil_fn, hand_fn = synth.hip_clip_fns(mano)
models = []
for i in range(args.clips):
    c = synth.make_clip(seed=i, frames=30, rend_size=256, image_size=256, obj="bottle", silhouette_fn=sil_fn, hand_verts_fn=hand_fn)
    models.append(build_model(copy.deepcopy(c["person_parameters"]), copy.deepcopy(c["object_parameters"]), objvertices=c["objvertices"],
                  objfaces=c["objfaces"], camintr=c["camintr"], optimize_mano=True, image_size=256, mano_model=mano, rend_size=256, sync_metrics=False))
st = FusedStepper(models if args.clips > 1 else models[0], dict(synth.STEP1_LOSS_WEIGHTS), 1e-2, max(args.at) + 10)
L = hlib.lib(); sctx = st.model.sil_ctx
ws, dims = hlib.ptr(sctx.workspace), (sctx.B, sctx.V, sctx.F, sctx.S)
n64 = L.hm_sil_timestamps_bytes(*dims) // 8
saved = torch.zeros(n64, dtype=torch.int64, device="cuda")
done = 0
for at in sorted(args.at):
    st.run(at - done); done = at
    hlib.check(L.hm_sil_timestamps(ws, *dims, 1, hlib.stream()), "ts")
    st.run(1); done += 1
    hlib.check(L.hm_sil_timestamps_save(ws, *dims, saved.data_ptr(), hlib.stream()), "save")
    hlib.check(L.hm_sil_timestamps(ws, *dims, 0, hlib.stream()), "ts")
    torch.cuda.synchronize()
    raw = saved.cpu().numpy().astype(np.uint64)
    sw = raw[-2 * 4 * 4096:].reshape(-1, 2)
    sw = sw[(sw[:, 0] > 0) & (sw[:, 1] > 0)]
    t0 = sw[:, 0].min(); 
    s = (sw[:, 0] - t0).astype(np.float64) / 100.0; e = (sw[:, 1] - t0).astype(np.float64) / 100.0   # 100 MHz
    busy = e - s
    print(f"iter {at}: waves {len(sw)} span {e.max():.1f} us; start p50 {np.median(s):.1f} p90 {np.percentile(s,90):.1f} p99 {np.percentile(s,99):.1f} max {s.max():.1f} late(>5us) {(s > 5).sum()}; busy mean {busy.mean():.1f} p50 {np.median(busy):.1f} p90 {np.percentile(busy,90):.1f} p99 {np.percentile(busy,99):.1f} max {busy.max():.1f}; end p50 {np.median(e):.1f} p90 {np.percentile(e,90):.1f}")
    # raster + lines workgroups
    nr = 30 * args.clips * 32 * 32 // 4
    for name, blk in (("raster", raw[:2 * nr].reshape(-1, 2)), ("lines", raw[2 * nr:-2 * 4 * 4096].reshape(-1, 2))):
        blk = blk[(blk[:, 0] > 0) & (blk[:, 1] > 0)]
        if not len(blk): continue
        t0 = blk[:, 0].min()
        s = (blk[:, 0] - t0) / 100.0; e = (blk[:, 1] - t0) / 100.0; busy = e - s
        # concurrency over time
        ts = np.linspace(0, e.max(), 12)[1:-1]
        conc = [int(((s <= t) & (e > t)).sum()) for t in ts]
        print(f"   {name}: wgs {len(blk)} span {e.max():.1f}; busy mean {busy.mean():.1f} p50 {np.median(busy):.1f} p90 {np.percentile(busy,90):.1f} max {busy.max():.1f}; sum busy/span {busy.sum()/e.max():.0f} wg; concurrency {conc}")
