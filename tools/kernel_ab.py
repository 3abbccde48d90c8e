#!/usr/bin/env python3
"""A/B timing of kernel forms on the bench workload (one process, one context per variant).

    python tools/kernel_ab.py --size 8192 --variants 'base:cheb_prefetch=1,nt_stores=0;pf2:cheb_prefetch=2,nt_stores=0;...'

Per variant: ms per outer step over --steps timed steps and, from one extra step each, the HIP-event probe of the hot
kernels (mean duration, algorithmic bytes per cell of the timed launches, fraction of 8 TB/s)."""
import argparse
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

PROBES = {8: "chebk<2,4>", 9: "chebk<1,4>", 0: "cg_pq", 7: "mg_up", 1: "mg_down", 10: "mg_down_rf", 2: "flux_march<2>"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=8192)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--turbulent", action="store_true")
    ap.add_argument("--variants", type=str, required=True)
    ap.add_argument("--probes", type=str, default="8,9,0,7,1,10,2")
    args = ap.parse_args()
    probes = [int(p) for p in args.probes.split(",") if p]
    for spec in args.variants.split(";"):
        name, _, opts = spec.partition(":")
        tuning = {k: float(v) for k, v in (kv.split("=") for kv in opts.split(",") if kv)}
        dev, ctl, cfg = bench.setup_device(args.size, args.turbulent, 2, 1e-8, 0, tuning=tuning)
        step = 0
        for _ in range(args.warmup):
            advance_one_step(dev, ctl, step)
            step += 1
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            advance_one_step(dev, ctl, step)
            step += 1
        dev.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        st = dev.stats()
        row = {"variant": name, "tuning": tuning, "ms_per_step": round(ms, 3), "p_iters": st["p_iters"], "mom_iters": st["mom_iters"]}
        cells = args.size * args.size
        for pid in probes:
            dev.probe(pid)
            advance_one_step(dev, ctl, step)
            step += 1
            dev.synchronize()
            n, us, bpc = dev.probe_read(with_bytes=True)
            dev.probe(-1)
            if n:
                row[PROBES[pid]] = {"n": n, "us": round(us, 1), "B/cell": round(bpc, 1),
                                    "frac": round(bpc * cells / (us * 1e-6) / 8e12, 4)}
        print(json.dumps(row), flush=True)
        dev.close()


if __name__ == "__main__":
    main()
