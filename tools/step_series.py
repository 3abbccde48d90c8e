#!/usr/bin/env python3
"""Per-step wall time and iteration counts of the bench workload: python tools/step_series.py N steps name=value,..."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

n, steps = int(sys.argv[1]), int(sys.argv[2])
tuning = {k: float(v) for k, v in (kv.split("=") for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else []))}
dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning=tuning)
rows = []
for step in range(steps):
    dev.synchronize()
    t0 = time.perf_counter()
    info = advance_one_step(dev, ctl, step)
    dev.synchronize()
    st = dev.stats()
    rows.append((round(1e3 * (time.perf_counter() - t0), 2), st["p_iters"], st["mom_iters"], float("%.3g" % st["p_relres"])))
print(json.dumps({"tuning": tuning, "rows": rows}))
dev.close()
