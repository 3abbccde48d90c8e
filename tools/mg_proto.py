"""CPU prototype of the structured multigrid-PCG of linalg.hip on a model problem (uniform
periodic-i / Neumann-j pressure operator with one detached reference cell): iteration counts of
smoother / transfer variants before they are written as kernels.  Not part of the product."""
import sys
import numpy as np


class Op:
    def __init__(s, ci, cj, extra):
        s.ci, s.cj, s.extra = ci, cj, extra           # ci (nj, ni) west faces (periodic), cj (nj+1, ni) south faces
        s.nj, s.ni = extra.shape
        s.D = ci + np.roll(ci, -1, 1) + cj[:-1] + cj[1:] + extra

    def off(s, x):
        y = s.ci * np.roll(x, 1, 1) + np.roll(s.ci, -1, 1) * np.roll(x, -1, 1)
        y[1:] += s.cj[1:-1] * x[:-1]
        y[:-1] += s.cj[1:-1] * x[1:]
        return y

    def A(s, x):
        return s.D * x - s.off(x)


def coarsen(op):
    ci = 0.5 * (op.ci[0::2, 0::2] + op.ci[1::2, 0::2])
    cj = 0.5 * (op.cj[0::2, 0::2] + op.cj[0::2, 1::2])
    e = op.extra
    ex = (e[0::2, 0::2] + e[1::2, 0::2] + e[0::2, 1::2] + e[1::2, 1::2]) / 2.0
    return Op(ci, cj, ex)


def prolong_const(e):
    return np.repeat(np.repeat(e, 2, 0), 2, 1)


def prolong_bilin(e):
    # cell-centred bilinear: weights 3/4, 1/4 per direction; periodic in i, constant extrapolation in j
    nj, ni = e.shape
    em = np.roll(e, 1, 1); ep = np.roll(e, -1, 1)
    xi = np.empty((nj, 2 * ni))
    xi[:, 0::2] = 0.75 * e + 0.25 * em
    xi[:, 1::2] = 0.75 * e + 0.25 * ep
    dn = np.vstack([xi[:1], xi[:-1]]); up = np.vstack([xi[1:], xi[-1:]])
    out = np.empty((2 * nj, 2 * ni))
    out[0::2] = 0.75 * xi + 0.25 * dn
    out[1::2] = 0.75 * xi + 0.25 * up
    return out


def restrict_sum(r):
    return r[0::2, 0::2] + r[1::2, 0::2] + r[0::2, 1::2] + r[1::2, 1::2]


def restrict_bilinT(r):
    # transpose of prolong_bilin
    nj2, ni2 = r.shape
    a = 0.75 * r[0::2] + 0.75 * r[1::2]
    lo = 0.25 * r[0::2]; hi = 0.25 * r[1::2]
    a[:-1] += lo[1:]; a[0] += lo[0]
    a[1:] += hi[:-1]; a[-1] += hi[-1]
    b = 0.75 * a[:, 0::2] + 0.75 * a[:, 1::2]
    b += np.roll(0.25 * a[:, 0::2], -1, 1) + np.roll(0.25 * a[:, 1::2], 1, 1)
    return b


class MG:
    def __init__(s, op, smoother="jac", om=(0.8, 0.8), P="const", R="sum", gamma=1.0, nmin=4, nu=2):
        s.ops = [op]
        while s.ops[-1].ni > nmin and s.ops[-1].nj > nmin and s.ops[-1].ni % 2 == 0 and s.ops[-1].nj % 2 == 0:
            s.ops.append(coarsen(s.ops[-1]))
        s.smoother, s.om, s.gamma, s.nu = smoother, om, gamma, nu
        s.P = prolong_const if P == "const" else prolong_bilin
        s.R = restrict_sum if R == "sum" else restrict_bilinT
        s.masks = []
        for o in s.ops:
            jj, ii = np.mgrid[0:o.nj, 0:o.ni]
            s.masks.append(((ii + jj) & 1) == 0)

    def smooth(s, l, x, b, post):
        o = s.ops[l]
        if s.smoother == "jac":
            oms = s.om[::-1] if post else s.om
            for k in range(s.nu):
                w = oms[k % len(oms)]
                x = x + w * (b - o.A(x)) / o.D
            return x
        red = s.masks[l]
        order = [~red, red] if post else [red, ~red]
        for m in order:
            xn = (b + o.off(x)) / o.D
            x = np.where(m, x + s.om[0] * (xn - x), x)
        return x

    def cyc(s, l, b):
        o = s.ops[l]
        if l == len(s.ops) - 1:
            x = np.zeros_like(b)
            for _ in range(40):
                x = x + 0.8 * (b - o.A(x)) / o.D
            return x
        x = s.smooth(l, np.zeros_like(b), b, False)
        r = b - o.A(x)
        e = s.cyc(l + 1, s.R(r))
        x = x + s.gamma * s.P(e)
        return s.smooth(l, x, b, True)


def pcg(op, M, b, tol=1e-8, maxit=100):
    x = np.zeros_like(b); r = b.copy(); nb = np.linalg.norm(b)
    z = M(r); p = z.copy(); rz = (r * z).sum()
    for it in range(1, maxit + 1):
        q = op.A(p); a = rz / (p * q).sum()
        x += a * p; r -= a * q
        if np.linalg.norm(r) <= tol * nb:
            return it
        z = M(r); rz2 = (r * z).sum(); p = z + (rz2 / rz) * p; rz = rz2
    return maxit


def model(n, aniso=1.0):
    ci = np.ones((n, n)) * aniso
    cj = np.ones((n + 1, n)); cj[0] = 0; cj[-1] = 0
    extra = np.zeros((n, n))
    # detached reference cell (0,0): couplings zeroed, neighbours keep them on the diagonal
    extra[0, 1] += ci[0, 1]; extra[0, n - 1] += ci[0, 0]; extra[1, 0] += cj[1, 0]
    extra[0, 0] = ci[0, 0] + ci[0, 1] + cj[1, 0]
    ci[0, 0] = 0; ci[0, 1] = 0; cj[1, 0] = 0
    return Op(ci, cj, extra)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    op = model(n)
    rng = np.random.default_rng(0)
    b = rng.standard_normal((n, n)); b[0, 0] = 0.0
    variants = {
        "jac .8/.8 const/sum (current)": dict(),
        "jac gamma 1.3": dict(gamma=1.3),
        "jac gamma 1.6": dict(gamma=1.6),
        "jac cheb .69/1.23": dict(om=(0.6906, 1.2317)),
        "jac .9/.9": dict(om=(0.9, 0.9)),
        "jac 1.0/0.6": dict(om=(1.0, 0.6)),
        "rbgs const/sum": dict(smoother="rb", om=(1.0,)),
        "rbgs 1.15 const/sum": dict(smoother="rb", om=(1.15,)),
        "jac bilin/bilinT": dict(P="bilin", R="bilinT"),
        "jac bilin/sum (nonsym)": dict(P="bilin"),
        "rbgs bilin/bilinT": dict(smoother="rb", om=(1.0,), P="bilin", R="bilinT"),
        "rbgs bilin/sum (nonsym)": dict(smoother="rb", om=(1.0,), P="bilin"),
        "jac cheb bilin/bilinT": dict(om=(0.6906, 1.2317), P="bilin", R="bilinT"),
    }
    for name, kw in variants.items():
        mg = MG(op, **kw)
        print(f"{name:34s} its(1e-8) = {pcg(op, lambda r: mg.cyc(0, r), b)}", flush=True)
