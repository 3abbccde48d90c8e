"""CPU prototype: V-cycle-preconditioned CG on anisotropic structured operators (wall-graded channel, annulus-like
grading along i), point-Jacobi against block line-Jacobi smoothers (lines along j and / or i, cut into blocks of L
cells) with the coarsening and transfer of linalg.hip.  Iteration counts at rtol 1e-8.  Not part of the product."""
import sys
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def graded(n, ratio_maxmin, two_sided=True):
    """n spacings, geometric, max/min = ratio_maxmin; two-sided: fine at both ends."""
    if ratio_maxmin == 1.0:
        return np.full(n, 1.0 / n)
    if two_sided:
        h = n // 2
        r = ratio_maxmin ** (1.0 / (h - 1))
        a = r ** np.arange(h)
        s = np.concatenate([a, a[::-1]])
    else:
        r = ratio_maxmin ** (1.0 / (n - 1))
        s = r ** np.arange(n)
    return s / s.sum()


class Op:
    """5-point operator, [j][i]; periodic in i if per_i; Neumann elsewhere except one Dirichlet side to make it regular."""
    def __init__(s, ci, cj, extra, per_i):
        s.ci, s.cj, s.extra, s.per_i = ci, cj, extra, per_i     # ci (nj, ni+1) faces, cj (nj+1, ni)
        s.nj, s.ni = extra.shape
        s.D = ci[:, :-1] + ci[:, 1:] + cj[:-1] + cj[1:] + extra

    def off(s, x):
        y = np.zeros_like(x)
        y[:, 1:] += s.ci[:, 1:-1] * x[:, :-1]
        y[:, :-1] += s.ci[:, 1:-1] * x[:, 1:]
        if s.per_i:
            y[:, 0] += s.ci[:, 0] * x[:, -1]
            y[:, -1] += s.ci[:, -1] * x[:, 0]
        y[1:] += s.cj[1:-1] * x[:-1]
        y[:-1] += s.cj[1:-1] * x[1:]
        return y

    def A(s, x):
        return s.D * x - s.off(x)

    def matrix(s):
        n = s.ni * s.nj
        idx = np.arange(n).reshape(s.nj, s.ni)
        rows, cols, vals = [idx.ravel()], [idx.ravel()], [s.D.ravel()]
        def add(a, b, v):
            rows.append(a.ravel()); cols.append(b.ravel()); vals.append(-v.ravel())
        add(idx[:, 1:], idx[:, :-1], s.ci[:, 1:-1]); add(idx[:, :-1], idx[:, 1:], s.ci[:, 1:-1])
        if s.per_i:
            add(idx[:, 0], idx[:, -1], s.ci[:, 0]); add(idx[:, -1], idx[:, 0], s.ci[:, -1])
        add(idx[1:], idx[:-1], s.cj[1:-1]); add(idx[:-1], idx[1:], s.cj[1:-1])
        return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))


def make_op(hx, hy, per_i=True, dirichlet_right=False):
    ni, nj = len(hx), len(hy)
    # face distances
    dxf = np.empty(ni + 1); dxf[1:-1] = 0.5 * (hx[:-1] + hx[1:]); dxf[0] = dxf[-1] = 0.5 * (hx[0] + hx[-1])
    dyf = np.empty(nj + 1); dyf[1:-1] = 0.5 * (hy[:-1] + hy[1:]); dyf[0] = hy[0]; dyf[-1] = hy[-1]
    ci = hy[:, None] / dxf[None, :] * np.ones((nj, ni + 1))
    cj = hx[None, :] / dyf[:, None] * np.ones((nj + 1, ni))
    extra = np.zeros((nj, ni))
    cj[0] = 0.0; cj[-1] = 0.0
    if not per_i:
        if dirichlet_right:
            extra[:, -1] += 2.0 * ci[:, -1]
        ci[:, 0] = 0.0; ci[:, -1] = 0.0
    if per_i or not dirichlet_right:
        extra[0, 0] += 1e-3 * (ci[0, 0] + cj[1, 0])       # regularise (stand-in for the pinned cell)
    return Op(ci, cj, extra, per_i)


def coarsen(op):
    fi = 2 if op.ni % 2 == 0 and op.ni > 4 else 1
    fj = 2 if op.nj % 2 == 0 and op.nj > 4 else 1
    return coarsen_f(op, fi, fj)


def coarsen_f(op, fi, fj):
    ci, cj, e = op.ci, op.cj, op.extra
    if fj == 2:
        ci = ci[0::2] + ci[1::2]
        cj = cj[0::2]
        e = e[0::2] + e[1::2]
    if fi == 2:
        ci = ci[:, 0::2]
        cj = cj[:, 0::2] + cj[:, 1::2]
        e = e[:, 0::2] + e[:, 1::2]
    # rediscretisation: parallel sum across the face (done above), series (1/f) along the normal
    ci = ci / fi
    cj = cj / fj
    e = e / max(fi, fj)
    return Op(ci, cj, e, op.per_i), fi, fj


def block_solver(op, direction, L):
    """M = the couplings along `direction` inside blocks of L cells (+ full diagonal): LU of the block-tridiagonal part."""
    n = op.ni * op.nj
    idx = np.arange(n).reshape(op.nj, op.ni)
    rows, cols, vals = [idx.ravel()], [idx.ravel()], [op.D.ravel()]
    if direction == "j":
        keep = (np.arange(1, op.nj) % L) != 0                   # face between rows g-1 and g is inside a block
        a, b, v = idx[1:][keep], idx[:-1][keep], op.cj[1:-1][keep]
    else:
        keep = (np.arange(1, op.ni) % L) != 0
        a, b, v = idx[:, 1:][:, keep], idx[:, :-1][:, keep], op.ci[:, 1:-1][:, keep]
    rows += [a.ravel(), b.ravel()]; cols += [b.ravel(), a.ravel()]; vals += [-v.ravel(), -v.ravel()]
    M = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    lu = spla.splu(M)
    return lambda r: lu.solve(r.ravel()).reshape(r.shape)


class MG:
    def __init__(s, op, smoother="point", om=(1.0, 0.55, 0.9), nu=3, L=64, om_line=0.8, semi=None):
        s.levels = [op]; s.f = []
        while max(s.levels[-1].ni, s.levels[-1].nj) > 8 and len(s.levels) < 12:
            o = s.levels[-1]
            if semi is not None:
                fi, fj = semi(o)
                c, fi, fj = coarsen_f(o, fi, fj)
            else:
                c, fi, fj = coarsen(o)
            if fi == 1 and fj == 1:
                break
            s.levels.append(c); s.f.append((fi, fj))
        s.smoother, s.om, s.nu, s.om_line = smoother, om, nu, om_line
        s.solv = []
        for o in s.levels:
            d = {}
            if "j" in smoother: d["j"] = block_solver(o, "j", L)
            if "i" in smoother: d["i"] = block_solver(o, "i", L)
            s.solv.append(d)
        s.coarse = spla.splu(s.levels[-1].matrix().tocsc())

    def smooth(s, l, x, b, post):
        o = s.levels[l]
        if s.smoother == "point":
            oms = s.om[::-1] if post else s.om
            for k in range(s.nu):
                x = x + oms[k % len(oms)] * (b - o.A(x)) / o.D
            return x
        seq = [c for c in s.smoother if c in "ijp"]
        if post: seq = seq[::-1]
        for c in seq:
            r = b - o.A(x)
            if c == "p": x = x + 0.8 * r / o.D
            else: x = x + s.om_line * s.solv[l][c](r)
        return x

    def cyc(s, l, b):
        o = s.levels[l]
        if l == len(s.levels) - 1:
            return s.coarse.solve(b.ravel()).reshape(b.shape)
        x = s.smooth(l, np.zeros_like(b), b, False)
        r = b - o.A(x)
        fi, fj = s.f[l]
        rc = r
        if fj == 2: rc = rc[0::2] + rc[1::2]
        if fi == 2: rc = rc[:, 0::2] + rc[:, 1::2]
        e = s.cyc(l + 1, rc)
        if fj == 2: e = np.repeat(e, 2, 0)
        if fi == 2: e = np.repeat(e, 2, 1)
        x = x + e
        return s.smooth(l, x, b, True)


def pcg(op, M, b, tol=1e-8, maxit=300):
    x = np.zeros_like(b); r = b.copy(); nb = np.linalg.norm(b)
    z = M(r); p = z.copy(); rz = (r * z).sum()
    for it in range(1, maxit + 1):
        q = op.A(p); a = rz / (p * q).sum()
        x += a * p; r -= a * q
        if np.linalg.norm(r) <= tol * nb:
            return it
        z = M(r); rz2 = (r * z).sum(); p = z + (rz2 / rz) * p; rz = rz2
    return maxit


def aniso_semi(thresh):
    def f(o):
        # coarsen only along the strongly coupled direction while the (geometric) mean coupling ratio is beyond thresh
        ri = np.exp(np.mean(np.log(o.cj[1:-1].mean() / o.ci[:, 1:-1].mean())))
        fi = 2 if o.ni % 2 == 0 and o.ni > 4 else 1
        fj = 2 if o.nj % 2 == 0 and o.nj > 4 else 1
        if ri > thresh and fj == 2: return 1, 2
        if ri < 1.0 / thresh and fi == 2: return 2, 1
        return fi, fj
    return f


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rng = np.random.default_rng(0)
    cases = {
        "uniform": make_op(graded(n, 1.0), graded(n, 1.0)),
        "channel dy max/min 5": make_op(graded(n, 1.0), graded(n, 5.0)),
        "channel dy max/min 20": make_op(graded(n, 1.0), graded(n, 20.0)),
        "channel dy 20, Lx = 4 H": make_op(4.0 * graded(n, 1.0), graded(n, 20.0)),
        "annulus-like (i graded 1-sided 100, dirichlet right)": make_op(graded(n, 100.0, False), 0.3 * graded(n // 2, 1.0), per_i=False, dirichlet_right=True),
    }
    variants = {
        "point V(3,3)": dict(),
        "j-line L64 + p": dict(smoother="jp"),
        "j-line L64 x1": dict(smoother="j"),
        "j-line L16 x1": dict(smoother="j", L=16),
        "i-line L64 x1": dict(smoother="i"),
        "j+i lines L64": dict(smoother="ji"),
        "j+i lines L16": dict(smoother="ji", L=16),
        "point + semi(4)": dict(semi=aniso_semi(4.0)),
    }
    for cname, op in cases.items():
        b = rng.standard_normal((op.nj, op.ni))
        print(f"== {cname}  ({op.ni} x {op.nj})", flush=True)
        for vname, kw in variants.items():
            mg = MG(op, **kw)
            print(f"   {vname:22s} its(1e-8) = {pcg(op, lambda r: mg.cyc(0, r), b):4d}   levels {len(mg.levels)}", flush=True)
