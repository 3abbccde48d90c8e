#!/usr/bin/env python3
"""Wall time per outer step of the reference's own shipped case (Runs/Coarse_mesh: 5 x 20 k-omega channel, SIMPLE, as restated
in oracle/cases.py) through run_case on the GPU, against the CPU oracle on this host: python tools/ref_case_timing.py [steps]"""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from oracle import cases                      # noqa: E402
from oracle import cfd_oracle as orc          # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
TUNING = {k: float(v) for k, v in (kv.split("=") for kv in sys.argv[2].split(","))} if len(sys.argv) > 2 else None
from cfd_amd import mesh as hip_mesh          # noqa: E402
from cfd_amd import solver as hip_solver      # noqa: E402

def gpu_run(model, n_steps):
    cfg, mp, kind = cases.channel_komega_5x20(steps=n_steps, model=model)
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        cfg["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp, "path": tmp / "mesh" / "mesh.npz"}
        cfg["paths"] = {"case_dir": tmp, "mesh_dir": tmp / "mesh", "states_dir": tmp / "states", "post_dir": tmp / "postproc"}
        cases.ramp_initial_u(cfg, hip_mesh.make_mesh(cfg["mesh"]))
        if TUNING:
            cfg["solver"]["tuning"] = dict(TUNING)
        t0 = time.perf_counter()
        fields, step, t = hip_solver.run_case(cfg)
        return time.perf_counter() - t0, step, mp


for model in ("k_omega", "sst"):
    gpu_run(model, 20)                         # library load, first use of every kernel
    import ctypes as C
    from cfd_amd import _lib
    lib = _lib.load()
    cnt = (C.c_longlong * 2)()
    w1, s1, mp = gpu_run(model, 100)
    lib.cfd_host_counters(cnt, 1)
    w2, s2, mp = gpu_run(model, 100 + steps)
    lib.cfd_host_counters(cnt, 0)
    print(f"     launches per step {cnt[0] / s2:.0f}, stream synchronisations per step {cnt[1] / s2:.1f}")
    kc = hip_solver.LAST_RUN["kernel_counts"]
    print(f"GPU  {model:8s} {mp['count_i']}x{mp['count_j']}  {1e3 * (w2 - w1) / (s2 - s1):.3f} ms/step (difference of a {s2}- and a {s1}-step run_case)  "
          f"semi_levels {kc['mg_semi_levels']} line_levels {kc['mg_line_levels_max']} spec {kc['speculative_iterations']}/{kc['speculative_redone']} "
          f"stats {hip_solver.LAST_RUN['stats']['mom_solver']} {hip_solver.LAST_RUN['stats']['mom_iters']} {hip_solver.LAST_RUN['stats']['p_iters']} {hip_solver.LAST_RUN['stats']['k_iters']}", flush=True)
    cfg2, mp2, _ = cases.channel_komega_5x20(steps=min(steps, 40), model=model)
    mesh = orc.rect_mesh(**mp2)
    cases.ramp_initial_u(cfg2, mesh)
    t0 = time.perf_counter()
    ref, rstep, rtime, hist = orc.run_case(cfg2, mesh)
    wall = time.perf_counter() - t0
    print(f"CPU  {model:8s} oracle {rstep} steps  {1e3 * wall / rstep:.3f} ms/step", flush=True)
