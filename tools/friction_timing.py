#!/usr/bin/env python3
"""Cost of the in-loop friction diagnostic (cfd_wall_friction + cfd_bulk_velocity every 10 steps) on the 4096^2
bench workload: python tools/friction_timing.py [N]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd import post_processing as pp  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


class Wall:      # j_min wall of the uniform unit channel: tangent +x, half a cell to the wall, faces of length dx
    side, lo, hi = 2, 0, n - 1
    t_hat = np.ascontiguousarray(np.tile([1.0, 0.0], (n, 1)))
    dn = np.full(n, 0.5 / n)
    seg_len = np.full(n, 1.0 / n)
    length = 1.0
    s_face = (np.arange(n) + 0.5) / n


res = {}
for with_friction in (False, True, False, True):
    dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0)
    for step in range(5):
        advance_one_step(dev, ctl, step)
    dev.synchronize()
    t0 = time.perf_counter()
    for step in range(5, 45):
        advance_one_step(dev, ctl, step)
        if with_friction and step % 10 == 0:
            out = pp.friction_coefficient_device(dev, Wall, nu=cfg["physics"]["nu"], hydraulic_diameter=2.0,
                                                 pressure_gradient=cfg["physics"]["pressure_gradient"])
    dev.synchronize()
    res.setdefault(with_friction, []).append(1e3 * (time.perf_counter() - t0) / 40)
    dev.close()
print(json.dumps({"n": n, "ms_per_step_without": res[False], "ms_per_step_with_friction_every_10": res[True],
                  "ratio": min(res[True]) / min(res[False]), "last": {k: v for k, v in out.items() if k in ("cf", "tau_avg", "bulk_velocity")}}))
