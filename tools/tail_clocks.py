#!/usr/bin/env python3
"""Where one k_mg_tail launch spends its time: 100 MHz timestamps of its phases (thread 0 of the workgroup), read from
an instrumented build of the library (-DTAIL_CLOCKS, tools/_clk/libcfd_hip_clk.so; see the recipe below).

    cd cfd_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DTAIL_CLOCKS -c linalg.hip \
        -o ../../tools/_clk/linalg.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_clk/libcfd_hip_clk.so \
        stencil.o ../../tools/_clk/linalg.o comm.o api.o noalloc.o
    python tools/tail_clocks.py 128        # 128^2 mesh: the tail starts at 64^2, as it does at 8192^2
"""
import ctypes
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
os.environ["CFD_HIP_LIB"] = str(ROOT / "tools" / "_clk" / "libcfd_hip_clk.so")
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from cfd_amd import _lib as L  # noqa: E402
from cfd_amd.solver import advance_one_step  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tuning = {}
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    tuning[k] = float(v) if "." in v else int(v)
dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning=tuning)
for step in range(6):
    advance_one_step(dev, ctl, step)
dev.synchronize()
lib = ctypes.CDLL(os.environ["CFD_HIP_LIB"])
buf = (ctypes.c_ulonglong * 256)()
rc = lib.cfd_debug_tail_clocks(buf)
assert rc == 0, rc
nm = int(buf[255])
t = [buf[i] for i in range(nm)]
print(f"mesh {n}^2  levels {dev.level_info() if hasattr(dev, 'level_info') else ''}")
print(f"marks {nm}  total {(t[-1] - t[0]) * 10} ns")
print("phase durations (ns):", [(t[i + 1] - t[i]) * 10 for i in range(nm - 1)])
dev.close()
