#!/usr/bin/env python3
"""A few outer steps of the bench workload at one size, for rocprofv3 --kernel-trace --stats:
python3 tools/step_profile.py N [steps] [turbulent]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
turb = len(sys.argv) > 3 and sys.argv[3] not in ("0", "false")
dev, ctl, cfg = bench.setup_device(n, turb, 2, 1e-8, 0)
for step in range(steps):
    advance_one_step(dev, ctl, step)
dev.synchronize()
print(dev.host_counters(), dev.stats())
dev.close()
