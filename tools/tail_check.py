#!/usr/bin/env python3
"""V-cycle tail on/off: same Krylov iteration counts and solutions (to rounding), time per step."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd import _lib as L  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

for n in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("96", "256", "1024"))]:
    res = {}
    for tail in ((1, 0) if len(sys.argv) > 2 else (0, 1)):
        dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning={"mg_tail": tail})
        for step in range(5):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        dev.host_counters(reset=True)
        t0 = time.perf_counter()
        for step in range(5, 25):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        launches, syncs = dev.host_counters()
        p = dev.download(L.F_P)
        res[tail] = (p, dev.stats()["p_iters"], ms, launches / 20, dev.kernel_counts()["mg_tail_levels"], dev.level_smoothers())
        dev.close()
    d = np.linalg.norm(res[0][0] - res[1][0]) / np.linalg.norm(res[0][0])
    print(json.dumps({"n": n, "rel_diff_p": d, "p_iters": [res[0][1], res[1][1]], "ms": [round(res[0][2], 3), round(res[1][2], 3)],
                      "launches": [res[0][3], res[1][3]], "tail_levels": res[1][4], "levels": len(res[1][5])}), flush=True)
