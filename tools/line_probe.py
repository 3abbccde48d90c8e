"""GPU check of the alternating line smoother: MG-Krylov iterations (rtol 1e-8) on graded channels and a graded annulus
with point smoothing, with the line smoother forced on and with the automatic choice; solution checked against scipy."""
import sys
from pathlib import Path
import numpy as np
import scipy.sparse.linalg as spla
sys.path.insert(0, str(Path(__file__).resolve().parents[1])); sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
from oracle import cases, cfd_oracle as orc
from test_gpu_parity import _device_for, _bc_variant

def run(tag, mesh, bc, per_i, per_j):
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    rng = np.random.default_rng(1)
    x_true = rng.standard_normal((ni, nj))
    for mode in (0, 1, -1):
        dev = _device_for(mesh, bc, 1e-3, turbulent=False)
        dev.set_option("mg_line", mode)
        f = {k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p", "nu_t")}
        dev.upload_fields(f)
        dev.momentum_diagonal(3e-3, 1.0)
        ci, cj, diag, _ = dev.pressure_system()
        a = orc.five_point_matrix(diag, ci[:-1], ci[1:], cj[:, :-1], cj[:, 1:], per_i, per_j)
        b = a.dot(x_true.ravel()).reshape(ni, nj)
        sol, iters, relres = dev.pressure_krylov(b, tol=1e-8, maxiter=400)
        res = np.linalg.norm(a.dot(sol.ravel()) - b.ravel()) / np.linalg.norm(b.ravel())
        lv = dev.level_smoothers()
        print(f"{tag:34s} mg_line={mode:2d} iters {iters:4d} relres {relres:.1e} true {res:.1e} solver {dev.stats()['p_solver']}"
              f" aniso {lv[0][2]:.1f} line levels {sum(1 for l in lv if l[3])}/{len(lv)}", flush=True)
        dev.close()

for n, r in ((128, 1.04), (256, 1.0118), (512, 1.0059)):
    mesh = orc.rect_mesh(1.0, 1.0, n, n, 1.0, r)      # one-sided geometric grading
    dy = np.diff(mesh["node"][0, :, 1]); print("max/min dy", dy.max() / dy.min())
    run(f"channel {n}^2 ratio {r}", mesh, _bc_variant("per_i", n, n), True, False)
mesh = orc.rect_mesh(4.0, 1.0, 256, 256, 1.0, 1.0118)
run("channel 256^2 Lx=4H", mesh, _bc_variant("per_i", 256, 256), True, False)
mesh = orc.rect_mesh(2.0, 1.0, 257, 129, 1.0, 1.02)
run("inout 257x129", mesh, _bc_variant("inout", 257, 129), False, False)
mesh = orc.annulus_mesh(0.5, 6.0, 132, 24, 0.0, float(np.pi), 1.02, 1.0)
run("annulus 132x24", mesh, cases._half_cylinder_bc(132, 24, False), False, False)
mesh = orc.annulus_mesh(0.5, 20.0, 512, 256, 0.0, float(np.pi), 1.012, 1.0)
run("annulus 512x256", mesh, cases._half_cylinder_bc(512, 256, False), False, False)
