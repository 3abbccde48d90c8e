"""CPU prototype (not collected by pytest): would the momentum Chebyshev solves need fewer iterations when they start
from x_old + (the increment the same solve produced one outer step earlier) instead of x_old?  Uses the oracle's
assembled systems of the bench workload at a small size.   python tests/proto_warm_start.py [n] [steps]"""
import sys
from pathlib import Path

import numpy as np
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from cfd_amd.workloads import channel_forced_config  # noqa: E402
from oracle import cfd_oracle as orc  # noqa: E402


def cheb_iters(a, b, x0, tol=1e-8, maxit=200):
    d = a.diagonal()
    off = abs(a).sum(axis=1).A1 - abs(d)
    rho = float((off / d).max())
    x = x0.copy()
    r = b - a @ x
    nb = np.linalg.norm(b)
    if np.linalg.norm(r) <= tol * nb:
        return 0
    sigma1 = 1.0 / rho
    rk = 1.0 / sigma1
    dvec = r / d
    for it in range(1, maxit + 1):
        x += dvec
        r = b - a @ x
        if np.linalg.norm(r) <= tol * nb:
            return it
        rn = 1.0 / (2.0 * sigma1 - rk)
        dvec = rn * rk * dvec + (2.0 * rn / rho) * (r / d)
        rk = rn
    return maxit


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    cfg, mp = channel_forced_config(n, False, 2, 1e-8)
    mesh = orc.rect_mesh(**mp)
    y = mesh["cell_center"][:, :, 1]
    rng = np.random.default_rng(1234)
    cfg["initial"]["u"] = (1.0 - (2.0 * y - 1.0) ** 2) + 1.0e-3 * rng.standard_normal(y.shape)
    cfg["time"]["max_steps"] = steps
    x0_queue, incr, log, call = [], {}, [], [0]
    real_ts = orc.transport_system

    def ts(phi, *a, **k):
        x0_queue.append(phi[1:-1, 1:-1].ravel().copy())
        return real_ts(phi, *a, **k)

    def momentum(self, a, b):
        x0 = x0_queue.pop(0)
        idx = call[0] % 4                      # u, v of inner iteration 1, u, v of inner iteration 2
        call[0] += 1
        plain = cheb_iters(a, b, x0)
        warm = cheb_iters(a, b, x0 + incr[idx]) if idx in incr else None
        x = spla.splu(a.tocsc()).solve(b)
        incr[idx] = x - x0
        log.append((idx, plain, warm))
        return x

    orc.transport_system = ts
    orc.LinearSolvers.momentum = momentum
    orc.run_case(cfg, mesh, solver_mode="reference", freeze_pressure=True)
    names = ("u inner 1", "v inner 1", "u inner 2", "v inner 2")
    for s in range(0, len(log), 4):
        print("step", s // 4, "  ".join(f"{names[i]}: {p} -> {w}" for i, p, w in log[s:s + 4]), flush=True)


if __name__ == "__main__":
    main()
