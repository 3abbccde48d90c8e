"""CPU prototype (round 3): which V-cycle brings the stretched-mesh pressure solves down?  Same coarsening rule and
piecewise-constant transfers as linalg.hip; smoothers are sequences of sweeps:
    P   damped point Jacobi            J / I   block line Jacobi along j / i (blocks of L cells, damping om_line)
    Zi  zebra line Gauss-Seidel over i-lines (even rows, then odd rows; reversed in the post-smoother)
    Zj  the same over j-lines (even columns, then odd columns)
and the coarsening per level is full (2 x 2), semi-j, semi-i or chosen from the level's anisotropy.
Workloads: the bench's wall-graded channel (max / min dy = 20) and its half annulus (r 0.5 .. 20), scaled to n cells
per side with the line blocks scaled alike (62 cells at 8192 -> 62 n / 8192).  PCG iterations at rtol 1e-8.
Not part of the product."""
import math
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


class Op:
    def __init__(s, ci, cj, extra, per_i):
        s.ci, s.cj, s.extra, s.per_i = ci, cj, extra, per_i     # ci (nj, ni+1) faces, cj (nj+1, ni)
        s.nj, s.ni = extra.shape
        s.D = ci[:, :-1] + ci[:, 1:] + cj[:-1] + cj[1:] + extra
        if per_i:
            pass

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

    def aniso(s):
        wi = s.ci[:, :-1] + s.ci[:, 1:]
        wj = s.cj[:-1] + s.cj[1:]
        return float((wj / np.maximum(wi, 1e-300)).max()), float((wi / np.maximum(wj, 1e-300)).max())


def graded2(n, ratio):
    h = n // 2
    r = ratio ** (1.0 / (h - 1))
    a = r ** np.arange(h)
    s = np.concatenate([a, a[::-1]])
    return s / s.sum()


def channel(n, ratio=20.0, lx=1.0):
    hx = np.full(n, lx / n)
    hy = graded2(n, ratio) if ratio != 1.0 else np.full(n, 1.0 / n)
    ni, nj = n, n
    dxf = np.full(ni + 1, lx / n)
    dyf = np.empty(nj + 1); dyf[1:-1] = 0.5 * (hy[:-1] + hy[1:]); dyf[0] = hy[0]; dyf[-1] = hy[-1]
    ci = hy[:, None] / dxf[None, :] * np.ones((nj, ni + 1))
    cj = hx[None, :] / dyf[:, None] * np.ones((nj + 1, ni))
    cj[0] = 0.0; cj[-1] = 0.0
    extra = np.zeros((nj, ni))
    extra[0, 0] += 1e-3 * (ci[0, 0] + cj[1, 0])
    return Op(ci, cj, extra, True)


def annulus(nr, nt, r_in=0.5, r_out=20.0):
    """i = radial (graded), j = azimuthal; wall at i = 0 (Neumann), outer boundary half Dirichlet (outlet), symmetry
    j sides (Neumann).  Same growth rule as cfd_amd.workloads.half_annulus_config."""
    dth = math.pi / nt
    ratio = min(1.0 + dth, 1.0 + 8.2 / nr)
    g = ratio ** np.arange(nr)
    dr = g / g.sum() * (r_out - r_in)
    rn = r_in + np.concatenate([[0.0], np.cumsum(dr)])     # node radii
    rc = 0.5 * (rn[:-1] + rn[1:])
    drf = np.empty(nr + 1); drf[1:-1] = rc[1:] - rc[:-1]; drf[0] = dr[0]; drf[-1] = dr[-1]
    ci = (rn * dth / drf)[None, :] * np.ones((nt, nr + 1))          # radial faces: length r dtheta over distance dr
    cj = (dr / (rc * dth))[None, :] * np.ones((nt + 1, nr))        # azimuthal faces
    cj[0] = 0.0; cj[-1] = 0.0
    extra = np.zeros((nt, nr))
    extra[: nt // 2, -1] += 2.0 * ci[: nt // 2, -1]                 # outlet half: Dirichlet p
    ci[:, 0] = 0.0; ci[:, -1] = 0.0
    return Op(ci, cj, extra, False)


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
    return Op(ci / fi, cj / fj, e / max(fi, fj), op.per_i), fi, fj


def line_solver(op, direction, L, rows=None):
    """LU of D + the couplings along `direction` inside blocks of L cells (L <= 0: whole lines)."""
    n = op.ni * op.nj
    idx = np.arange(n).reshape(op.nj, op.ni)
    r_, c_, v_ = [idx.ravel()], [idx.ravel()], [op.D.ravel()]
    if direction == "j":
        keep = np.ones(op.nj - 1, bool) if L <= 0 else (np.arange(1, op.nj) % L) != 0
        a, b, v = idx[1:][keep], idx[:-1][keep], op.cj[1:-1][keep]
    else:
        keep = np.ones(op.ni - 1, bool) if L <= 0 else (np.arange(1, op.ni) % L) != 0
        a, b, v = idx[:, 1:][:, keep], idx[:, :-1][:, keep], op.ci[:, 1:-1][:, keep]
        if op.per_i and L <= 0:
            r_ += [idx[:, 0].ravel(), idx[:, -1].ravel()]; c_ += [idx[:, -1].ravel(), idx[:, 0].ravel()]
            v_ += [-op.ci[:, 0].ravel(), -op.ci[:, -1].ravel()]
    r_ += [a.ravel(), b.ravel()]; c_ += [b.ravel(), a.ravel()]; v_ += [-v.ravel(), -v.ravel()]
    M = sp.csc_matrix((np.concatenate(v_), (np.concatenate(r_), np.concatenate(c_))), shape=(n, n))
    lu = spla.splu(M)
    return lambda r: lu.solve(r.ravel()).reshape(r.shape)


class MG:
    def __init__(s, op, seq=("J", "I", "P"), L=64, om_line=0.8, om_p=(0.8,), coarsening="full", thresh=4.0, min_size=8,
                 Lj=None, zeb_om=1.0):
        s.levels = [op]; s.f = []
        while len(s.levels) < 24:
            o = s.levels[-1]
            ci_ok = o.ni % 2 == 0 and o.ni > min_size
            cj_ok = o.nj % 2 == 0 and o.nj > min_size
            fi, fj = (2 if ci_ok else 1), (2 if cj_ok else 1)
            if coarsening == "semi_j" and cj_ok:
                fi = 1
            elif coarsening == "semi_i" and ci_ok:
                fj = 1
            elif coarsening == "auto":
                aj, ai = o.aniso()             # max strong-j ratio, max strong-i ratio
                if aj > thresh and ai <= thresh and cj_ok: fi = 1
                elif ai > thresh and aj <= thresh and ci_ok: fj = 1
            if fi == 1 and fj == 1:
                break
            c, fi, fj = coarsen_f(o, fi, fj)
            s.levels.append(c); s.f.append((fi, fj))
        s.seq, s.om_line, s.om_p, s.zeb_om = seq, om_line, om_p, zeb_om
        Lj = L if Lj is None else Lj
        s.solv = []
        for o in s.levels[:-1]:
            d = {}
            if any(c in ("J", "Zj") for c in seq): d["j"] = line_solver(o, "j", Lj)
            if any(c in ("I", "Zi") for c in seq): d["i"] = line_solver(o, "i", L)
            s.solv.append(d)
        s.coarse = spla.splu(s.levels[-1].matrix().tocsc())
        s.work = sum(o.ni * o.nj for o in s.levels) / (op.ni * op.nj)

    def smooth(s, l, x, b, post):
        o = s.levels[l]
        seq = s.seq[::-1] if post else s.seq
        kp = 0
        for c in seq:
            if c == "P":
                oms = s.om_p[::-1] if post else s.om_p
                x = x + oms[kp % len(oms)] * (b - o.A(x)) / o.D
                kp += 1
            elif c in ("J", "I"):
                x = x + s.om_line * s.solv[l][c.lower()](b - o.A(x))
            elif c == "Zi":
                for par in ((1, 0) if post else (0, 1)):
                    r = b - o.A(x)
                    r[1 - par::2] = 0.0
                    x = x + s.zeb_om * s.solv[l]["i"](r)
            elif c == "Zj":
                for par in ((1, 0) if post else (0, 1)):
                    r = b - o.A(x)
                    r[:, 1 - par::2] = 0.0
                    x = x + s.zeb_om * s.solv[l]["j"](r)
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
        return s.smooth(l, x + e, b, True)


def pcg(op, M, b, tol=1e-8, maxit=200):
    x = np.zeros_like(b); r = b.copy(); nb = np.linalg.norm(b)
    z = M(r); p = z.copy(); rz = (r * z).sum()
    for it in range(1, maxit + 1):
        q = op.A(p); a = rz / (p * q).sum()
        x += a * p; r -= a * q
        if np.linalg.norm(r) <= tol * nb:
            return it
        z = M(r); rz2 = (r * z).sum(); p = z + (rz2 / rz) * p; rz = rz2
    return maxit


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    which = sys.argv[2] if len(sys.argv) > 2 else "all"
    Lb = max(4, round(62 * n / 8192))          # line blocks scaled with the mesh
    rng = np.random.default_rng(0)
    cases = {}
    if which in ("all", "channel"):
        cases["graded channel"] = channel(n, 20.0)
    if which in ("all", "annulus"):
        cases["half annulus"] = annulus(n, n // 2)
    if which in ("all", "uniform"):
        cases["uniform"] = channel(n, 1.0)
    variants = {
        "shipped: J I P, blocks": dict(seq=("J", "I", "P"), L=Lb),
        "J I P, 64-cell blocks": dict(seq=("J", "I", "P"), L=64),
        "J I P, whole lines": dict(seq=("J", "I", "P"), L=0),
        "Zi + semi_j, blocks": dict(seq=("Zi",), L=Lb, coarsening="semi_j"),
        "Zi + semi_j, whole": dict(seq=("Zi",), L=0, coarsening="semi_j"),
        "Zj + semi_i, blocks": dict(seq=("Zj",), L=Lb, coarsening="semi_i"),
        "Zj + semi_i, whole": dict(seq=("Zj",), L=0, coarsening="semi_i"),
        "Zi Zj full, blocks": dict(seq=("Zi", "Zj"), L=Lb),
        "Zi Zj full, whole": dict(seq=("Zi", "Zj"), L=0),
        "Zi P full, blocks": dict(seq=("Zi", "P"), L=Lb),
        "J Zi full, blocks": dict(seq=("J", "Zi"), L=Lb),
        "Zi + auto(4), blocks": dict(seq=("Zi",), L=Lb, coarsening="auto"),
        "Zi Zj + auto(4), blocks": dict(seq=("Zi", "Zj"), L=Lb, coarsening="auto"),
    }
    for cname, op in cases.items():
        b = rng.standard_normal((op.nj, op.ni))
        print(f"== {cname}  ({op.ni} x {op.nj}), line blocks {Lb}, aniso (j, i) = {op.aniso()}", flush=True)
        for vname, kw in variants.items():
            mg = MG(op, **kw)
            its = pcg(op, lambda r: mg.cyc(0, r), b)
            print(f"   {vname:28s} its = {its:4d}   levels {len(mg.levels):2d}  cells/finest {mg.work:.2f}", flush=True)
