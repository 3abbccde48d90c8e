#!/usr/bin/env python3
"""Host <-> device cost of one field at the bench size (what a caller pays who hands fields over host buffers every step,
which run_case does not: fields stay resident, states go to the host at save_interval only).  python tools/pcie_rate.py"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0)
a = np.random.default_rng(0).standard_normal((n + 2, n + 2))
for name, fn in (("upload", lambda: dev.upload(L.F_U, a)), ("download", lambda: dev.download(L.F_U, out=a))):
    fn()
    dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    dev.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{name}: {1e3 * dt:.1f} ms per {n}x{n} field ({a.nbytes / dt / 1e9:.1f} GB/s incl. the transpose and the host copy)", flush=True)
dev.close()
