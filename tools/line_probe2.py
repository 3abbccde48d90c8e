import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from oracle import cases, cfd_oracle as orc
from test_gpu_parity import _device_for, _bc_variant

def run(tag, mesh, bc, per_i, per_j, opts, coarse=40):
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    rng = np.random.default_rng(1)
    x_true = rng.standard_normal((ni, nj))
    dev = _device_for(mesh, bc, 1e-3, turbulent=False)
    dev.set_multigrid(3, 3, coarse_sweeps=coarse, fused=True, omegas=(1.0, 0.55, 0.9))
    for k, v in opts.items():
        dev.set_option(k, v)
    f = {k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p", "nu_t")}
    dev.upload_fields(f)
    dev.momentum_diagonal(3e-3, 1.0)
    ci, cj, diag, _ = dev.pressure_system()
    a = orc.five_point_matrix(diag, ci[:-1], ci[1:], cj[:, :-1], cj[:, 1:], per_i, per_j)
    b = a.dot(x_true.ravel()).reshape(ni, nj)
    sol, iters, relres = dev.pressure_krylov(b, tol=1e-8, maxiter=400)
    print(f"{tag:28s} {str(opts):60s} coarse {coarse:4d} iters {iters:4d}", flush=True)
    dev.close()

m1 = orc.rect_mesh(1.0, 1.0, 128, 128, 1.0, 1.04)
m2 = orc.rect_mesh(4.0, 1.0, 256, 256, 1.0, 1.0118)
m3 = orc.rect_mesh(1.0, 1.0, 256, 256, 1.0, 1.0)
for opts, coarse in (({"mg_line": 1}, 40), ({"mg_line": 1}, 400), ({"mg_line": 1, "mg_line_omega": 0.7}, 40), ({"mg_line": 1, "mg_line_omega": 0.9}, 40),
                     ({"mg_line": 1, "mg_line_omega": 1.0}, 40)):
    run("chan128 r1.04", m1, _bc_variant("per_i", 128, 128), True, False, opts, coarse)
    run("chan256 Lx4", m2, _bc_variant("per_i", 256, 256), True, False, opts, coarse)
    run("uniform256", m3, _bc_variant("per_i", 256, 256), True, False, opts, coarse)
run("uniform256", m3, _bc_variant("per_i", 256, 256), True, False, {"mg_line": 0}, 40)
