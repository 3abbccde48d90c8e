#!/usr/bin/env python3
"""MG-Krylov iterations and true residuals of the pressure solve on stretched meshes: alternating block line smoother
(mg_semi = 0) against the semi-coarsened whole-line multigrid (mg_semi = 1).  GPU; run through gpurun."""
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
from oracle import cases                      # noqa: E402
from oracle import cfd_oracle as orc          # noqa: E402
from test_gpu_parity import _bc_variant, _device_for      # noqa: E402


def run(mesh, bc, per_i, per_j, semi, tol=1e-8):
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    dev = _device_for(mesh, bc, 1e-3, turbulent=False)
    dev.set_option("mg_semi", semi)
    dev.upload_fields({k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p", "nu_t")})
    dev.momentum_diagonal(3e-3, 1.0)
    ci, cj, diag, _ = dev.pressure_system()
    a = orc.five_point_matrix(diag, ci[:-1], ci[1:], cj[:, :-1], cj[:, 1:], per_i, per_j)
    x_true = np.random.default_rng(1).standard_normal((ni, nj))
    b = a.dot(x_true.ravel()).reshape(ni, nj)
    sol, iters, relres = dev.pressure_krylov(b, tol=tol, maxiter=400)
    true_res = np.linalg.norm(a.dot(sol.ravel()) - b.ravel()) / np.linalg.norm(b.ravel())
    kc = dev.kernel_counts()
    solver = dev.stats()["p_solver"]
    dev.close()
    return iters, relres, true_res, kc["mg_semi_levels"], solver


CASES = {
    "channel_128_r1.04": lambda: (orc.rect_mesh(1.0, 1.0, 128, 128, 1.0, 1.04), _bc_variant("per_i", 128, 128), True, False),
    "channel_256_Lx4": lambda: (orc.rect_mesh(4.0, 1.0, 256, 256, 1.0, 1.0118), _bc_variant("per_i", 256, 256), True, False),
    "annulus_132x24": lambda: (orc.annulus_mesh(0.5, 6.0, 132, 24, 0.0, float(np.pi), 1.02, 1.0),
                               cases._half_cylinder_bc(132, 24, False), False, False),
    "annulus_512x256": lambda: (orc.annulus_mesh(0.5, 20.0, 512, 256, 0.0, float(np.pi), 1.012, 1.0),
                                cases._half_cylinder_bc(512, 256, False), False, False),
    "inout_257x129": lambda: (orc.rect_mesh(2.0, 1.0, 257, 129, 1.0, 1.02), _bc_variant("inout", 257, 129), False, False),
    "inout_300x128": lambda: (orc.rect_mesh(2.0, 1.0, 300, 128, 1.0, 1.02), _bc_variant("inout", 300, 128), False, False),
    "channel_1100x96_r1.05": lambda: (orc.rect_mesh(3.0, 1.0, 1100, 96, 1.0, 1.05), _bc_variant("per_i", 1100, 96), True, False),
    "channel_2200x64_r1.08": lambda: (orc.rect_mesh(3.0, 1.0, 2200, 64, 1.0, 1.08), _bc_variant("per_i", 2200, 64), True, False),
}

if __name__ == "__main__":
    for tag, make in CASES.items():
        if len(sys.argv) > 1 and sys.argv[1] not in tag:
            continue
        mesh, bc, pi, pj = make()
        for semi in (0, 1):
            print(tag, "mg_semi", semi, "iters %d relres %.2e true %.2e semi_levels %d %s" % run(mesh, bc, pi, pj, semi), flush=True)
