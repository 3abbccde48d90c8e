"""Launch counts of the Chebyshev momentum solves with and without the six-iteration launches (option cheb_k6)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from cfd_amd.solver import advance_one_step

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-12
for spec in (0, 1):
    for k6 in (1, 0):
        dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning={"cheb_k6": k6, "speculate": spec})
        dev.set_solver(tol, 200, tol, 200, tol, 200, True)
        rows = []
        for step in range(5):
            k_before = dev.kernel_counts()
            advance_one_step(dev, ctl, step)
            k_after = dev.kernel_counts()
            st = dev.stats()
            rows.append((k_after["cheb4_2rhs"] - k_before["cheb4_2rhs"], k_after["cheb4_1rhs"] - k_before["cheb4_1rhs"],
                         st["mom_iters"], k_after["speculative_redone"]))
        print(f"spec {spec} k6 {k6}: per step (2rhs, 1rhs, mom_iters, redone) {rows}", flush=True)
        dev.close()
