#!/usr/bin/env python3
"""Step time of the wall-graded bench channel on small meshes, semi-coarsened line multigrid against the round-2 block line
smoother (launch-bound regime: the reference's own cases are this size): python tools/small_graded_timing.py"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

for n in (32, 64, 128, 256, 512, 1024):
    for semi in (1, 0):
        dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning={"mg_semi": semi}, graded=True)
        step = 0
        for _ in range(5):
            advance_one_step(dev, ctl, step)
            step += 1
        dev.synchronize()
        dev.host_counters(reset=True)
        k, per_step = 20, []
        for _ in range(k):
            t0 = time.perf_counter()
            advance_one_step(dev, ctl, step)
            dev.synchronize()
            per_step.append(1e3 * (time.perf_counter() - t0))
            step += 1
        launches, syncs = dev.host_counters()
        st = dev.stats()
        print(json.dumps({"n": n, "mg_semi": semi, "ms_per_step": round(sorted(per_step)[k // 2], 3), "launches_per_step": round(launches / k, 1),
                          "syncs_per_step": round((syncs - k) / k, 1), "p_iters": st["p_iters"], "p_solver": st["p_solver"],
                          "semi_levels": dev.kernel_counts()["mg_semi_levels"]}), flush=True)
        dev.close()
