"""Runs bench.lockstep_parity (teacher-forced HIP-vs-oracle parity along the fused loop's trajectory) -> one JSON line.
usage: python tools/lockstep.py [--step2] [--steps 50] [--frames 30] [--size 256] [--obj bottle] [--no-free]"""
import argparse
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step2", action="store_true")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--obj", default="bottle")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-free", action="store_true")
    args = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", str(min(os.cpu_count() or 1, 64)))
    import torch
    torch.set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
    import bench_parity as bench
    from homan_amd.mano_assets import synthetic_mano
    out = bench.lockstep_parity(synthetic_mano(0), step2=args.step2, steps=args.steps, frames=args.frames, size=args.size,
                                obj=args.obj, seed=args.seed, free_run=not args.no_free)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
