"""Would a single-precision V-cycle (fp64 Krylov recurrences around it) cost PCG iterations?  CPU prototype on the
model problem of tools/mg_proto.py; decides whether a mixed-precision preconditioner is worth writing as kernels.
    python tools/mg_proto_fp32.py [n]"""
import sys

import numpy as np

from mg_proto import MG, Op, model, pcg


def cast(op, dt):
    o = Op.__new__(Op)
    o.ci, o.cj, o.extra = op.ci.astype(dt), op.cj.astype(dt), op.extra.astype(dt)
    o.nj, o.ni = op.nj, op.ni
    o.D = (o.ci + np.roll(o.ci, -1, 1) + o.cj[:-1] + o.cj[1:] + o.extra).astype(dt)
    return o


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rng = np.random.default_rng(0)
    for graded in (False, True):
        op = model(n)
        if graded:      # smoothly varying face conductances over two decades, like a wall-refined mesh
            w = np.exp(2.3 * np.cos(np.linspace(0.0, np.pi, n)))[:, None] * np.ones((1, n))
            wj = np.vstack([w[:1], 0.5 * (w[1:] + w[:-1]), w[-1:]])
            op = Op(op.ci * w, op.cj * wj, op.extra * w)
        b = rng.standard_normal((n, n)); b[0, 0] = 0.0
        kw = dict(om=(1.0, 0.55, 0.9), nu=3)
        mg64 = MG(op, **kw)
        mg32 = MG(cast(op, np.float32), **kw)
        for tol in (1e-8, 1e-13):
            i64 = pcg(op, lambda r: mg64.cyc(0, r), b, tol=tol)
            i32 = pcg(op, lambda r: mg32.cyc(0, r.astype(np.float32)).astype(np.float64), b, tol=tol)
            print(f"n={n} graded={graded} tol={tol:g}: fp64 V-cycle {i64} its, fp32 V-cycle {i32} its", flush=True)
