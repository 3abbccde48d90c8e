#!/usr/bin/env python3
"""Step time of the bench workload on small meshes (launch / host-sync bound regime): python tools/small_mesh_timing.py"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

# optional arguments: sizes "32,48t,64" (t: k-omega), then name=value tuning options of cfd_set_option
cases = ((32, False), (64, False), (128, False), (256, False), (512, False), (1024, False), (64, True), (256, True))
if len(sys.argv) > 1:
    cases = tuple((int(v.rstrip("t")), v.endswith("t")) for v in sys.argv[1].split(","))
tuning = {k: float(v) for k, v in (kv.split("=") for kv in sys.argv[2:])} or None
for n, turb in cases:
    dev, ctl, cfg = bench.setup_device(n, turb, 2, 1e-8, 0, tuning=tuning)
    step = 0
    for _ in range(5):
        advance_one_step(dev, ctl, step)
        step += 1
    dev.synchronize()
    dev.host_counters(reset=True)
    dev.host_syncs_by_phase(reset=True)
    k = 20
    per_step = []
    for _ in range(k):
        t0 = time.perf_counter()
        advance_one_step(dev, ctl, step)
        dev.synchronize()
        per_step.append(1e3 * (time.perf_counter() - t0))
        step += 1
    # median: one step in a few hundred stalls for tens of milliseconds on the box (host side, no kernel is that long)
    ms = sorted(per_step)[k // 2]
    launches, syncs = dev.host_counters()
    st = dev.stats()
    print(json.dumps({"n": n, "turbulent": turb, "ms_per_step": round(ms, 3), "ms_mean": round(sum(per_step) / k, 3), "cells_per_s": round(n * n / ms * 1e3),
                      "launches_per_step": round(launches / k, 1), "syncs_per_step": round((syncs - k) / k, 1),
                      "syncs_by_phase": {k: round(v / k_, 1) for k, v in dev.host_syncs_by_phase().items() if v} if (k_ := k) else {}, "mom_iters": st["mom_iters"], "p_iters": st["p_iters"], "k_iters": st["k_iters"],
                      "speculative": {q: v for q, v in dev.kernel_counts().items() if q.startswith("speculative")},
                      "pcg_tail": dev.kernel_counts()["pcg_tail"]}), flush=True)
    dev.close()
