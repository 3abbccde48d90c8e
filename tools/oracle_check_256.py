#!/usr/bin/env python3
"""Ad-hoc end-to-end check of the default GPU path against the CPU oracle at 256 x 256 (laminar and k-omega forced channel,
three outer steps, tolerances 1e-13): python tools/oracle_check_256.py [n] [name=value,...]"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
sys.path.insert(0, str(ROOT / "tests"))
import cases  # noqa: E402
import cfd_oracle as orc  # noqa: E402
from conftest import mask_corners  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tuning = {k: float(v) for k, v in (kv.split("=") for kv in (sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] else []))}
for turbulent in (False, True):
    cfg, mp, kind = cases.channel_forced(n=n, steps=3, turbulent=turbulent, ratio_j=1.01)
    mesh = orc.rect_mesh(**mp)
    cases.perturbed_parabola(cfg, mesh)
    ref, rstep, rtime, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True)
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        cfg2, mp2, _ = cases.channel_forced(n=n, steps=3, turbulent=turbulent, ratio_j=1.01)
        cfg2["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp2, "path": tmp / "mesh" / "mesh.npz"}
        cfg2["paths"] = {"case_dir": tmp, "mesh_dir": tmp / "mesh", "states_dir": tmp / "states", "post_dir": tmp / "postproc"}
        mesh2 = hip_mesh.make_mesh(cfg2["mesh"])
        cases.perturbed_parabola(cfg2, mesh2)
        cfg2["solver"]["turbulence_solver"] = {"tol": 1e-13, "maxiter": 2000}
        cfg2["solver"]["momentum"]["tol"] = 1e-13
        if tuning:
            cfg2["solver"]["tuning"] = dict(tuning)
        fields, step, time = hip_solver.run_case(cfg2)
    out = {}
    for key in ("u", "v", "p", "k", "omega", "nu_t"):
        if key in ref:
            den = np.linalg.norm(mask_corners(ref[key]).ravel())
            err = float(np.linalg.norm((mask_corners(fields[key]) - mask_corners(ref[key])).ravel()))
            unorm = float(np.linalg.norm(ref["u"].ravel()))
            out[key] = (err / max(den, 1e-300), "ok" if err <= 1e-10 * den + 1e-12 * unorm else "ABOVE THE BAR", float(den), unorm)
    print("stats", hip_solver.LAST_RUN["stats"], flush=True)
    print("turbulent" if turbulent else "laminar", n, step, out, {k: v for k, v in hip_solver.LAST_RUN["kernel_counts"].items() if v}, flush=True)
