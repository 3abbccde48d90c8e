// Stencil kernels of the SIMPLE/PISO outer step for gfx950 (MI355X).
//
// Every kernel is HBM-bound fp64 5-point work (no MFMA): threads sweep rows of the
// [j][i] planes with i on the 64 lanes of a wavefront so that each wave-instruction
// touches one contiguous 512-byte run; the vertical neighbours come from the same
// column of the adjacent rows and are served by L1/L2.  Reductions use wavefront
// shuffles + one LDS stage per workgroup and a fixed-order second pass (deterministic).
//
// Arithmetic order follows the reference expression by expression (file:line cited at
// each kernel) and the build uses -ffp-contract=off, so interior results agree with the
// numpy oracle to the last bit except where a reduction order differs.
#include "internal.h"

namespace cfd {

// ----------------------------------------------------------------------------------
// small device helpers
// ----------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
    return v;
}

// block = 256 threads (64 x 4); returns the block result in thread 0
__device__ __forceinline__ double block_sum_256(double v) {
    __shared__ double sm[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ double block_min_256(double v) {
    __shared__ double sm[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    v = wave_min(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    return fmin(fmin(sm[0], sm[1]), fmin(sm[2], sm[3]));
}

struct FluxIn {
    const double *u, *v, *p, *k, *ap;   // k != nullptr: p_eff = p + (2/3) k  (solver.py:427-436)
    int rc;
};

__device__ __forceinline__ double peff(const FluxIn &f, size_t c) {
    return f.k ? f.p[c] + (2.0 / 3.0) * f.k[c] : f.p[c];
}

// Face flux through the i-face whose right-hand cell has plane index cR
// (solver.py:455-513 with Rhie-Chow, 516-544 without).
__device__ __forceinline__ double flux_i(const FluxIn &f, const MeshDev &m, size_t cR, bool lo, bool hi) {
    const double wr = m.wri[cR], wl = 1.0 - wr;
    const double ui = wl * f.u[cR - 1] + wr * f.u[cR];
    const double vi = wl * f.v[cR - 1] + wr * f.v[cR];
    double fl = ui * m.nix[cR] + vi * m.niy[cR];
    if (f.rc) {
        double af = lo ? f.ap[cR] : (hi ? f.ap[cR - 1] : wl * f.ap[cR - 1] + wr * f.ap[cR]);
        af = af > TINY ? af : TINY;
        const double s = m.si[cR];
        fl = fl - (s * s) / af * (peff(f, cR) - peff(f, cR - 1));
    }
    return fl;
}
__device__ __forceinline__ double flux_j(const FluxIn &f, const MeshDev &m, size_t cN, int pitch, bool lo, bool hi) {
    const size_t cS = cN - pitch;
    const double wn = m.wnj[cN], ws = 1.0 - wn;
    const double uj = ws * f.u[cS] + wn * f.u[cN];
    const double vj = ws * f.v[cS] + wn * f.v[cN];
    double fl = uj * m.njx[cN] + vj * m.njy[cN];
    if (f.rc) {
        double af = lo ? f.ap[cN] : (hi ? f.ap[cS] : ws * f.ap[cS] + wn * f.ap[cN]);
        af = af > TINY ? af : TINY;
        const double s = m.sj[cN];
        fl = fl - (s * s) / af * (peff(f, cN) - peff(f, cS));
    }
    return fl;
}

// Green-Gauss gradient of a ghosted plane at cell c (solver.py:322-358)
template <typename F>
__device__ __forceinline__ void green_gauss(F phi, const MeshDev &m, size_t c, int pitch, double &gx, double &gy) {
    const double pc = phi(c);
    double wr = m.wri[c];
    const double pw = (1.0 - wr) * phi(c - 1) + wr * pc;
    wr = m.wri[c + 1];
    const double pe = (1.0 - wr) * pc + wr * phi(c + 1);
    double wn = m.wnj[c];
    const double ps = (1.0 - wn) * phi(c - pitch) + wn * pc;
    wn = m.wnj[c + pitch];
    const double pn = (1.0 - wn) * pc + wn * phi(c + pitch);
    const double vol = m.vol[c];
    gx = (m.nix[c + 1] * pe - m.nix[c] * pw + m.njx[c + pitch] * pn - m.njx[c] * ps) / vol;
    gy = (m.niy[c + 1] * pe - m.niy[c] * pw + m.njy[c + pitch] * pn - m.njy[c] * ps) / vol;
}

// first-order upwind divergence (solver.py:649-669)
__device__ __forceinline__ double upwind(const double *phi, size_t c, int pitch, double fw, double fe, double fs,
                                         double fn, double vol) {
    const double pc = phi[c];
    const double pw = fw >= 0.0 ? phi[c - 1] : pc;
    const double pe = fe >= 0.0 ? pc : phi[c + 1];
    const double ps = fs >= 0.0 ? phi[c - pitch] : pc;
    const double pn = fn >= 0.0 ? pc : phi[c + pitch];
    const double div = fe * pe - fw * pw + fn * pn - fs * ps;
    return -div / vol;
}

// ----------------------------------------------------------------------------------
// eddy viscosity (turbulence.py:388-394): whole ghosted plane, corners included
// ----------------------------------------------------------------------------------
constexpr double SST_A1 = 0.31;
// sst: 0 = k-omega; 1 = SST before _S exists (a1 k / omega, turbulence.py:382-384); 2 = SST limiter
__global__ void k_eddy_viscosity(Geo g, const double *k, const double *om, double *nut, int turbulent, int sst,
                                 const double *S, const double *F2) {
    const int ig = blockIdx.x * 64 + threadIdx.x, jg = blockIdx.y * 4 + threadIdx.y;
    if (ig > g.ni + 1 || jg > g.nj + 1) return;
    const size_t c = g.at(ig, jg);
    if (!turbulent) {
        nut[c] = 0.0;
        return;
    }
    const double ks = fmax(k[c], K_FLOOR), os = fmax(om[c], OMEGA_FLOOR);
    if (sst == 0) {
        nut[c] = 1.0 * ks / os;
    } else {
        const double denom = sst == 2 ? fmax(SST_A1 * os, S[c] * F2[c]) : os;
        nut[c] = fmax(SST_A1 * ks / denom, 0.0);
    }
}

int launch_eddy_viscosity(Ctx *c) {
    const int sst = c->turb_model == CFD_TURB_SST ? (c->sst_cached ? 2 : 1) : 0;
    hipLaunchKernelGGL(k_eddy_viscosity, grid2d(c->g.ni + 2, c->g.nj + 2), block2d(), 0, c->stream, c->g, c->k,
                       c->omega, c->nut, c->turbulent, sst, c->sst_S, c->sst_F2);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// turbulence._sources_komega (turbulence.py:427-444) for velocity gradients the CALLER computed (the function-level
// API; the outer step evaluates the same expressions inside k_flux_march<4>): interior cells only.
__global__ void k_komega_sources(Geo g, const double *k, const double *om, const double *nut, const double *ux, const double *uy,
                                 const double *vx, const double *vy, double *sk, double *sw) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    const double s2 = 2.0 * (ux[c] * ux[c]) + 2.0 * (vy[c] * vy[c]) + (uy[c] + vx[c]) * (uy[c] + vx[c]);
    const double ks = fmax(k[c], K_FLOOR), os = fmax(om[c], OMEGA_FLOOR);
    sk[c] = nut[c] * s2 - 0.09 * ks * os;
    sw[c] = (5.0 / 9.0) * s2 - (3.0 / 40.0) * (os * os);
}

int launch_komega_sources(Ctx *c, const double *ux, const double *uy, const double *vx, const double *vy, double *sk, double *sw) {
    hipLaunchKernelGGL(k_komega_sources, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->k, c->omega, c->nut, ux, uy,
                       vx, vy, sk, sw);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// turbulence.effective_diffusivity, constant-sigma branch (turbulence.py:568-577): nu + sigma nu_t on the whole
// ghosted plane, corners included
__global__ void k_effective_diffusivity(Geo g, const double *nut, double nu, double sigma, double *out) {
    const int ig = blockIdx.x * 64 + threadIdx.x, jg = blockIdx.y * 4 + threadIdx.y;
    if (ig > g.ni + 1 || jg > g.nj + 1) return;
    const size_t c = g.at(ig, jg);
    out[c] = nu + sigma * nut[c];
}

int launch_effective_diffusivity(Ctx *c, double nu, double sigma, double *out) {
    hipLaunchKernelGGL(k_effective_diffusivity, grid2d(c->g.ni + 2, c->g.nj + 2), block2d(), 0, c->stream, c->g, c->nut, nu, sigma, out);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// face diffusion conductances + momentum diagonal (solver.py:147-230, 439-452)
// nu_cell = nu + coef * nu_t.  Writes a_i (west faces + the i_max face), a_j, a_P.
// ----------------------------------------------------------------------------------
// nut may be any cell plane: nu = 0, coef = 1 gives nu_cell = the plane itself, bit for bit (SST
// blended diffusivities _D_k, _D_omega)
__global__ void k_diffusion_ap(Geo g, MeshDev m, const double *nut, double nu, double coef, double inv_dt, double *ai,
                               double *aj, double *ap) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    const int P = g.pitch;
    // nut == nullptr: one viscosity for every face, the reference's scalar branch (solver.py:205-207)
    if (!nut) {
        const double aw = nu * m.si[c] / m.di[c], ae = nu * m.si[c + 1] / m.di[c + 1];
        const double as = nu * m.sj[c] / m.dj[c], an = nu * m.sj[c + P] / m.dj[c + P];
        ai[c] = aw;
        if (i == g.ni - 1) ai[c + 1] = ae;
        aj[c] = as;
        if (j == g.nj - 1) aj[c + P] = an;
        ap[c] = m.vol[c] * inv_dt + aw + ae + as + an;
        return;
    }
    const double nuP = nu + coef * nut[c];
    // west
    double nf = nuP;
    if (i > 0) {
        const double wr = m.wri[c];
        nf = (1.0 - wr) * (nu + coef * nut[c - 1]) + wr * nuP;
    }
    const double aw = nf * m.si[c] / m.di[c];
    ai[c] = aw;
    // east
    nf = nuP;
    if (i < g.ni - 1) {
        const double wr = m.wri[c + 1];
        nf = (1.0 - wr) * nuP + wr * (nu + coef * nut[c + 1]);
    }
    const double ae = nf * m.si[c + 1] / m.di[c + 1];
    if (i == g.ni - 1) ai[c + 1] = ae;
    // south
    nf = nuP;
    if (!g.blo(j)) {
        const double wn = m.wnj[c];
        nf = (1.0 - wn) * (nu + coef * nut[c - P]) + wn * nuP;
    }
    const double as = nf * m.sj[c] / m.dj[c];
    aj[c] = as;
    // north
    nf = nuP;
    if (!g.bhi(j)) {
        const double wn = m.wnj[c + P];
        nf = (1.0 - wn) * nuP + wn * (nu + coef * nut[c + P]);
    }
    const double an = nf * m.sj[c + P] / m.dj[c + P];
    if (j == g.nj - 1) aj[c + P] = an;
    ap[c] = m.vol[c] * inv_dt + aw + ae + as + an;
}

int launch_diffusion_ap(Ctx *c, double dt, double nut_coef, double *ai, double *aj, double *ap, const double *nu_plane) {
    hipLaunchKernelGGL(k_diffusion_ap, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(),
                       nu_plane ? nu_plane : (c->scalar_nu ? (const double *)nullptr : c->nut), nu_plane ? 0.0 : c->nu,
                       nu_plane ? 1.0 : nut_coef, 1.0 / dt, ai, aj, ap);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// k-omega: the eddy-viscosity update fused into the conductance pass.  nu_t = k / omega is evaluated on the fly for
// the cell and its four neighbours (same expression as k_eddy_viscosity, so the planes are bit-identical), the
// nu_t plane is written as a by-product -- ghost ring and corners included, by the threads of the adjacent boundary
// cells -- and both conductance sets come out of one pass: momentum (nu + nu_t -> a_i, a_j, a_P) and, with
// BOTH, the k / omega transport (nu + 0.5 nu_t -> t_i, t_j, t_P; solver.py:1098-1107).  136 instead of
// 24 + 88 + 88 B/cell.
template <bool BOTH>
__global__ void k_komega_coeffs(Geo g, MeshDev m, const double *k, const double *om, double nu, double inv_dt, double *nut,
                                double *ai, double *aj, double *ap, double *ti, double *tj, double *tp) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    const int P = g.pitch;
    auto nt = [&](size_t q) { return 1.0 * fmax(k[q], K_FLOOR) / fmax(om[q], OMEGA_FLOOR); };
    const double nP = nt(c), nW = nt(c - 1), nE = nt(c + 1), nS = nt(c - P), nN = nt(c + P);
    nut[c] = nP;
    const bool w = i == 0, e = i == g.ni - 1, so = j == 0, no = j == g.nj - 1;
    if (w) nut[c - 1] = nW;
    if (e) nut[c + 1] = nE;
    if (so) nut[c - P] = nS;
    if (no) nut[c + P] = nN;
    if (w && so) nut[c - P - 1] = nt(c - P - 1);
    if (e && so) nut[c - P + 1] = nt(c - P + 1);
    if (w && no) nut[c + P - 1] = nt(c + P - 1);
    if (e && no) nut[c + P + 1] = nt(c + P + 1);
    const double vdt = m.vol[c] * inv_dt;
    const double wrw = m.wri[c], wre = m.wri[c + 1], wns = m.wnj[c], wnn = m.wnj[c + P];
#pragma unroll
    for (int set = 0; set < (BOTH ? 2 : 1); ++set) {
        const double coef = set == 0 ? 1.0 : 0.5;
        const double nuP = nu + coef * nP;
        double nf = nuP;
        if (i > 0) nf = (1.0 - wrw) * (nu + coef * nW) + wrw * nuP;
        const double aw = nf * m.si[c] / m.di[c];
        nf = nuP;
        if (i < g.ni - 1) nf = (1.0 - wre) * nuP + wre * (nu + coef * nE);
        const double ae = nf * m.si[c + 1] / m.di[c + 1];
        nf = nuP;
        if (!g.blo(j)) nf = (1.0 - wns) * (nu + coef * nS) + wns * nuP;
        const double as = nf * m.sj[c] / m.dj[c];
        nf = nuP;
        if (!g.bhi(j)) nf = (1.0 - wnn) * nuP + wnn * (nu + coef * nN);
        const double an = nf * m.sj[c + P] / m.dj[c + P];
        double *oi = set == 0 ? ai : ti, *oj = set == 0 ? aj : tj, *op = set == 0 ? ap : tp;
        oi[c] = aw;
        if (e) oi[c + 1] = ae;
        oj[c] = as;
        if (no) oj[c + P] = an;
        op[c] = vdt + aw + ae + as + an;
    }
}

// both = true: the turbulence step (momentum and k / omega transport conductances); false: the predictor
int launch_komega_coeffs(Ctx *c, double dt, bool both) {
    if (both)
        hipLaunchKernelGGL((k_komega_coeffs<true>), grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), c->k, c->omega,
                           c->nu, 1.0 / dt, c->nut, c->ai, c->aj, c->ap, c->ti, c->tj, c->tp);
    else
        hipLaunchKernelGGL((k_komega_coeffs<false>), grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), c->k, c->omega,
                           c->nu, 1.0 / dt, c->nut, c->ai, c->aj, c->ap, c->ti, c->tj, c->tp);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// flow boundary conditions (bc.py:259-347).  blockIdx.y = side.  Ghosts depend only on
// interior values, so one launch covers all four sides; the host resolved overlapping
// segments (last one wins) into per-index tables.
// ----------------------------------------------------------------------------------
struct WrapList {
    double *p[12];
    int n;
};

__global__ void k_flow_bc(Geo g, FlowBCDev bc, double *u, double *v, double *p, WrapList wl, int wrap_p_i,
                          int wrap_p_j) {
    const int side = blockIdx.y;
    const int n = (side < 2) ? g.nj : g.ni;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if ((side == CFD_SIDE_JMIN && g.cutlo) || (side == CFD_SIDE_JMAX && g.cuthi)) return;   // slab cut: halo exchange
    const FlowBCSide &s = bc.side[side];
    if (s.n != n) return;   // table not set
    size_t cin, cgh, cop;    // interior cell, its ghost, the interior cell on the opposite side
    if (side == CFD_SIDE_IMIN) {
        cin = g.at(1, t + 1); cgh = g.at(0, t + 1); cop = g.at(g.ni, t + 1);
    } else if (side == CFD_SIDE_IMAX) {
        cin = g.at(g.ni, t + 1); cgh = g.at(g.ni + 1, t + 1); cop = g.at(1, t + 1);
    } else if (side == CFD_SIDE_JMIN) {
        cin = g.at(t + 1, 1); cgh = g.at(t + 1, 0); cop = g.at(t + 1, g.nj);
    } else {
        cin = g.at(t + 1, g.nj); cgh = g.at(t + 1, g.nj + 1); cop = g.at(t + 1, 1);
    }
    const int kuv = s.kind_uv[t], kp = s.kind_p[t];
    if (kuv == CFD_BC_PERIODIC) {
        u[cgh] = u[cop];
        v[cgh] = v[cop];
        for (int q = 0; q < wl.n; ++q) wl.p[q][cgh] = wl.p[q][cop];
    } else if (kuv == CFD_BC_AFFINE) {
        u[cgh] = s.ua[t] * u[cin] + s.ub[t];
        v[cgh] = s.va[t] * v[cin] + s.vb[t];
    } else if (kuv == CFD_BC_SYMMETRY) {
        const double up = u[cin], vp = v[cin], nx = s.nx[t], ny = s.ny[t];
        const double un = up * nx + vp * ny;
        u[cgh] = up - 2.0 * un * nx;
        v[cgh] = vp - 2.0 * un * ny;
    }
    if (kp == CFD_BC_PERIODIC) {
        const bool wrap = (side < 2) ? wrap_p_i : wrap_p_j;
        if (wrap) p[cgh] = p[cop];
    } else if (kp == CFD_BC_AFFINE) {
        p[cgh] = s.pa[t] * p[cin] + s.pb[t];
    }
}

int launch_flow_bc(Ctx *c, double *u, double *v, double *p) {
    WrapList wl{};
    wl.n = 0;
    if (c->turbulent) {
        wl.p[wl.n++] = c->k;
        wl.p[wl.n++] = c->omega;
    }
    wl.p[wl.n++] = c->nut;
    if (c->turb_model == CFD_TURB_SST && c->sst_cached) {
        // the reference keeps its SST caches inside `fields`, so the periodic copy wraps them too (bc.py:94-120)
        for (double *q : {c->sst_S, c->sst_F1, c->sst_F2, c->sst_Dk, c->sst_Dw, c->sst_dw}) wl.p[wl.n++] = q;
    }
    const int n = c->g.ni > c->g.nj ? c->g.ni : c->g.nj;
    hipLaunchKernelGGL(k_flow_bc, dim3((n + 255) / 256, 4, 1), dim3(256, 1, 1), 0, c->stream, c->g, c->fbc, u, v, p, wl,
                       c->wrap_p_i, c->wrap_p_j);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// k / omega boundary program (turbulence.py:266-341): ops run in list order because a
// wall op rewrites INTERIOR cells that later ops read at the corners.
// ----------------------------------------------------------------------------------
__global__ void k_turb_op(Geo g, int op, int side, int lo, int hi, double a, const double *vals, double *f,
                          double *f2) {
    const int t = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t > hi) return;
    if (op == CFD_TOP_WRAP_I) {   // t runs over j
        f[g.at(0, t + 1)] = f[g.at(g.ni, t + 1)];
        f[g.at(g.ni + 1, t + 1)] = f[g.at(1, t + 1)];
        f2[g.at(0, t + 1)] = f2[g.at(g.ni, t + 1)];
        f2[g.at(g.ni + 1, t + 1)] = f2[g.at(1, t + 1)];
        return;
    }
    if (op == CFD_TOP_WRAP_J) {   // t runs over i
        f[g.at(t + 1, 0)] = f[g.at(t + 1, g.nj)];
        f[g.at(t + 1, g.nj + 1)] = f[g.at(t + 1, 1)];
        f2[g.at(t + 1, 0)] = f2[g.at(t + 1, g.nj)];
        f2[g.at(t + 1, g.nj + 1)] = f2[g.at(t + 1, 1)];
        return;
    }
    size_t cin, cgh;
    if (side == CFD_SIDE_IMIN) {
        cin = g.at(1, t + 1); cgh = g.at(0, t + 1);
    } else if (side == CFD_SIDE_IMAX) {
        cin = g.at(g.ni, t + 1); cgh = g.at(g.ni + 1, t + 1);
    } else if (side == CFD_SIDE_JMIN) {
        cin = g.at(t + 1, 1); cgh = g.at(t + 1, 0);
    } else {
        cin = g.at(t + 1, g.nj); cgh = g.at(t + 1, g.nj + 1);
    }
    const double val = vals[t - lo];
    if (op == CFD_TOP_SET_INTERIOR)
        f[cin] = val;
    else
        f[cgh] = a * f[cin] + val;
}

int launch_turb_bc(Ctx *c) {
    for (const cfd_turb_op &o : c->turb_ops) {
        double *f = (o.field == CFD_F_K) ? c->k : c->omega;
        int lo = o.lo, hi = o.hi;
        if (o.op == CFD_TOP_WRAP_I) { lo = 0; hi = c->g.nj - 1; }
        if (o.op == CFD_TOP_WRAP_J) { lo = 0; hi = c->g.ni - 1; }
        const int n = hi - lo + 1;
        if (n <= 0) continue;
        hipLaunchKernelGGL(k_turb_op, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->g, o.op, o.side, lo, hi, o.a,
                           c->turb_vals + o.value_off, f, c->omega);
        CFD_HIP(hipGetLastError());
    }
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// adaptive time step (solver.py:547-582): two global minima in one pass
// ----------------------------------------------------------------------------------
__global__ void k_dt(Geo g, MeshDev m, const double *u, const double *v, const double *nut, double nu, int use_cfl,
                     double cfl, int use_diff, double cfl_diff, double *partials, int tiles_x, int tiles) {
    double madv = INFINITY, mdif = INFINITY;
    FluxIn f{u, v, nullptr, nullptr, nullptr, 0};
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        const size_t c = g.cell(i, j);
        if (use_cfl) {
            const double fw = flux_i(f, m, c, false, false), fe = flux_i(f, m, c + 1, false, false);
            const double fs = flux_j(f, m, c, g.pitch, false, false), fn = flux_j(f, m, c + g.pitch, g.pitch, false, false);
            const double tot = fabs(fw) + fabs(fe) + fabs(fs) + fabs(fn);
            const double safe = tot > TINY ? tot : TINY;
            madv = fmin(madv, cfl * m.vol[c] / safe);
        }
        if (use_diff) {
            const double ne = nu + nut[c];
            const double ns = ne > 0.0 ? ne : TINY;
            const double l = m.lmin[c];
            mdif = fmin(mdif, cfl_diff * (l * l) / ns);
        }
    }
    madv = block_min_256(madv);
    mdif = block_min_256(mdif);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        partials[blockIdx.x] = madv;
        partials[MAX_PARTIALS + blockIdx.x] = mdif;
    }
}

// has_fixed1: out[1] is not a minimum of partials but the value the caller knows already (the diffusive time-step minimum
// of a laminar run, a constant of mesh, nu and cfl_diff: launch_correct_flux)
__global__ void k_min_finish(const double *partials, int n, int nq, double *out, double fixed1 = 0.0, int has_fixed1 = 0) {      // nq <= 8; shape of k_sum_finish
    __shared__ double sm[8][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = 0; q < nq; ++q) {
        double v = INFINITY;
        for (int t = tid; t < n; t += 256) v = fmin(v, partials[(size_t)q * MAX_PARTIALS + t]);
        v = wave_min(v);
        if (lane == 0) sm[q][wave] = v;
    }
    __syncthreads();
    if (tid < nq) out[tid] = (has_fixed1 && tid == 1) ? fixed1 : fmin(fmin(sm[tid][0], sm[tid][1]), fmin(sm[tid][2], sm[tid][3]));
}

// The diffusive minimum min(cfl_diff l^2 / nu) of a laminar run does not change while mesh, nu and cfl_diff stay: the
// first look-ahead pass computes it (and finish_continuity / launch_dt keep what the host then reads), the later ones
// leave the l plane alone (8 B per cell) and k_min_finish puts the kept value in its slot -- the same double.
static inline bool dt_diff_known(const Ctx *c) {
    return c->lmin_const && c->dtdiff_valid && !c->turbulent && c->dtdiff_cfl == c->dtp_cfl_diff && c->dtdiff_nu == c->nu;
}
static inline void dt_diff_keep(Ctx *c, double value) {
    if (!c->dt_pass_diff_fresh) return;
    c->dt_pass_diff_fresh = false;
    c->dtdiff_valid = true; c->dtdiff_cfl = c->dtp_cfl_diff; c->dtdiff_nu = c->nu; c->dtdiff_val = value;
}

static inline int tiles_of(const Geo &g, int &tx) {
    tx = (g.ni + 63) / 64;
    return tx * ((g.nj + 3) / 4);
}
static inline int red_blocks(int tiles) { return tiles < 2048 ? tiles : 2048; }

// ----------------------------------------------------------------------------------
// Marching form of the Rhie-Chow flux kernels (momentum RHS, pressure RHS, continuity).
// One wavefront owns a strip of 62 columns plus one halo lane on each side and walks down the
// rows of a segment: every plane row is loaded once per wave as one coalesced run, horizontal
// neighbours come from DPP wave shifts, vertical neighbours from registers of the previous
// rows.  Each face flux is evaluated ONCE (the west and the south face of every cell; the east
// face is the right lane's west face, the north face the next row's south face) with the very
// expressions of flux_i / flux_j above, so all results keep the bits of the tile kernels.
// The tile kernels read 6 rows for 4 and every face twice (1.5-1.9x the algorithmic bytes at
// 8192^2, profiles/r1e_pmc8192_bytes_per_cell.csv); this form reads (Ls + 2) / Ls rows.
//   MODE 0: continuity sum only            (13 planes read,            104 B/cell)
//   MODE 1: b = -div F (pressure RHS)      (13 read, 1 written,        112 B/cell)
//   MODE 2: momentum RHS b_u, b_v + the continuity sum of the same fluxes (13 (+k) read, 2 written)
//   MODE 4: k / omega right-hand sides (solver.py:1109-1136, turbulence.py:427-444): Rhie-Chow flux with
//           p_eff = p + 2/3 k, upwind advection of k and omega, Green-Gauss grad u, grad v for the production
//           term, boundary terms with the transport conductances (ma.ai / ma.aj = t_i / t_j; ma.bu / ma.bv = b_k / b_omega)
//   MODE 2 with DT: additionally the two time-step minima of MODE 3 for the NEXT outer step, from the interpolated
//           part of the same face fluxes (the fields of a finished iteration are the fields compute_adaptive_dt
//           sees at the start of the next step); +lmin read, laminar runs only (nu_t changes in between otherwise)
//   MODE 3: the two global minima of the adaptive time step (solver.py:547-582) from the plain interpolated
//           fluxes (no Rhie-Chow term): u, v, 7 metric planes, lmin, nu_t read, 88 B/cell
// ----------------------------------------------------------------------------------
constexpr int FXW = 62;

struct FluxMarch {
    int nstrips, nsegs, Ls, blocks;
    int wide = 4;        // 8: lockstep workgroups (blocks counts those)
};
constexpr int WIDE_FLUX = 128;       // Ctx::wide_mask bit of the flux kernels
// Axis-aligned mesh?  Counts the faces whose area vector is not (S, 0) (i-faces) / (0, S) (j-faces) bit for bit: west and
// south face of every cell, face ni of the last column, face nj of the last row of the mesh (not of a slab cut: that row
// is the neighbour's).  Zero on every rank -> the marching flux kernels take n_ix = S_i, n_jy = S_j, n_iy = n_jx = 0 from
// the area planes (template argument RN) and leave the four normal planes alone.
__global__ void k_rect_normals(Geo g, MeshDev m, double *partials, int tiles_x, int tiles) {
    double a = 0.0, a2 = 0.0;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        const size_t c = g.cell(i, j);
        int bad = (m.nix[c] != m.si[c]) | (m.niy[c] != 0.0) | (m.njx[c] != 0.0) | (m.njy[c] != m.sj[c]);
        if (i == g.ni - 1) bad |= (m.nix[c + 1] != m.si[c + 1]) | (m.niy[c + 1] != 0.0);
        if (j == g.nj - 1 && !g.cuthi) bad |= (m.njx[c + g.pitch] != 0.0) | (m.njy[c + g.pitch] != m.sj[c + g.pitch]);
        a += bad ? 1.0 : 0.0;
        // separable areas: S_i of a row equals the row's first entry (face ni included), S_j of a column the column's entry
        // in row 0 (row nj included: the top face row or, at a cut, the neighbour's row 0 -- which chains the ranks' rows 0)
        const size_t r0 = g.cell(0, j), c0 = g.cell(i, 0);
        int sep = (m.si[c] != m.si[r0]) | (m.sj[c] != m.sj[c0]);
        if (i == g.ni - 1) sep |= (m.si[c + 1] != m.si[r0]);
        if (j == g.nj - 1) sep |= (m.sj[c + g.pitch] != m.sj[c0]);
        a2 += sep ? 1.0 : 0.0;
    }
    a = block_sum_256(a);
    __syncthreads();
    a2 = block_sum_256(a2);
    if (threadIdx.x == 0 && threadIdx.y == 0) { partials[blockIdx.x] = a; partials[MAX_PARTIALS + blockIdx.x] = a2; }
}
// evaluated once per mesh, at the first marching flux pass (every rank gets there at the same point of the run: the sum
// is all-reduced over the slabs); option rect_normals = 0 keeps the general kernels
// 0: general kernels, 1: normals from the area planes, 2: ... and separable areas (option rect_normals = 0 / 1 / 2 caps it)
int rect_normals(Ctx *c) {
    if (!c->rn_opt) return 0;
    if (c->rn_state < 0) {
        int tx;
        const int tiles = tiles_of(c->g, tx), nb = red_blocks(tiles);
        hipLaunchKernelGGL(k_rect_normals, dim3(nb), block2d(), 0, c->stream, c->g, c->mesh(), c->partials, tx, tiles);
        double bad[2] = {1.0, 1.0};
        if (hipGetLastError() != hipSuccess || sum_finish(c, c->partials, nb, 2, c->scal + 54) != CFD_OK ||
            read_scalars(c, 54, 2, bad) != CFD_OK)
            bad[0] = 1.0;
        c->rn_state = bad[0] == 0.0 ? (bad[1] == 0.0 ? 2 : 1) : 0;
    }
    const int form = c->rn_state < c->rn_opt ? c->rn_state : c->rn_opt;
    c->kc[KC_MESH_FORM] = form;           // (a state, not a count: what the flux kernels of this mesh run as)
    return form;
}

// lockstep form for the momentum / pressure right-hand-side passes on meshes with enough strips
static inline void flux_march_widen(const Ctx *c, FluxMarch &m) {
    if (!(c->wide_mask & WIDE_FLUX) || m.nstrips < 16) return;
    m.wide = 8;
    const long nb = (long)((m.nstrips + 7) / 8) * m.nsegs;
    m.blocks = (int)(nb < MAX_PARTIALS ? nb : MAX_PARTIALS);
}
#define LAUNCH_FLUX(fm, K4, K8, ...)                                                                        \
    do {                                                                                                    \
        if ((fm).wide == 8) hipLaunchKernelGGL(K8, dim3((fm).blocks), dim3(512), 0, c->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(K4, dim3((fm).blocks), dim3(256), 0, c->stream, __VA_ARGS__);               \
    } while (0)
static inline FluxMarch flux_march_geo(const Geo &g) {
    FluxMarch m;
    m.nstrips = (g.ni + FXW - 1) / FXW;
    int Ls = 32;
    while (Ls > 4 && (long)m.nstrips * ((g.nj + Ls - 1) / Ls) < 8192) Ls /= 2;
    m.Ls = Ls;
    m.nsegs = (g.nj + Ls - 1) / Ls;
    const long nb = ((long)m.nstrips * m.nsegs + 3) / 4;
    m.blocks = (int)(nb < MAX_PARTIALS ? nb : MAX_PARTIALS);
    return m;
}

struct FluxRow {
    double u, v, p, ap;            // fields of the row
    double wri, nix, niy, si;      // west i-face of the cell
    double wnj, njx, njy, sj;      // south j-face of the cell
    double vol, k;
    double lmin, nut;              // MODE 3 (MODE 4: nut)
    double w;                      // MODE 4: omega
};

struct MomArgs {
    const double *kgrad, *ai, *aj;
    double inv_dt, force_u, force_v;
    int per_i, per_j;
    double *bu, *bv;
    // MODE 3 (time step), MODE 4 (k / omega right-hand sides: nut, omega)
    const double *nut, *omega;
    double nu, cfl, cfl_diff;
    int use_cfl, use_diff;
    int nts;                       // nontemporal stores of the right-hand side planes (store_stream)
};

// RN (axis-aligned meshes, Ctx::rect_normals): every face's area vector is (S, 0) on the i-faces and (0, S) on the
// j-faces, bit for bit (checked on the device once per mesh, rect_normals()), so the four normal planes are not read:
// n_ix = S_i and n_jy = S_j come from the area planes the pass reads anyway, n_iy = n_jx = 0 (-32 B per cell; x * n_ix +
// y * (+-0) is x * S_i to the bit wherever it is not zero itself).
#define FX_NIX(R) (RN ? (R).si : (R).nix)
#define FX_NIY(R) (RN ? 0.0 : (R).niy)
#define FX_NJX(R) (RN ? 0.0 : (R).njx)
#define FX_NJY(R) (RN ? (R).sj : (R).njy)
// WPB = 8: lockstep workgroups of eight neighbouring strips (MarchItems, internal.h)
// RNA = 2: the areas are separable as well -- S_i the same along a row, S_j the same along a column, bit for bit (tensor-
// product node grids; checked with the normals) -- so S_i is one value per row (every lane loads the row's first entry)
// and S_j one register per lane and work item: another 16 B per cell less.
template <int MODE, bool KG, bool DT = false, int WPB = 4, int RNA = 0>
__global__ __launch_bounds__(64 * WPB) void k_flux_march(Geo g, MeshDev m, FluxIn f, FluxMarch fm, MomArgs ma, double *bout,
                                                         double *partials) {
    constexpr bool RN = RNA >= 1 && MODE != 3;         // (the time-step pass reads no area planes)
    constexpr bool SEP = RNA == 2 && MODE != 3;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const size_t P = g.pitch;
    const int rc = MODE == 3 ? 0 : f.rc;
    double acc = MODE == 3 ? INFINITY : 0.0, acc2 = INFINITY, acc3 = INFINITY;
    double accb = 0.0;       // MODE 1: sum b^2, the (b, b) the pressure solve starts from
    const MarchItems<WPB> items(fm.nstrips, fm.nsegs, 0, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, fm.nstrips, wave, strip, seg, active);
        const int iraw = strip * FXW - 1 + lane;          // -1 .. ni : ghost columns included
        const int ic = iraw < g.ni ? iraw : g.ni;
        const size_t colo = (size_t)(ic + CO);
        const bool out = lane >= 1 && lane <= FXW && iraw < g.ni && active;
        const bool ilo = iraw == 0, ihi = iraw == g.ni;
        const int j0 = seg * fm.Ls, j1 = min(j0 + fm.Ls, g.nj);
        const double sj_col = SEP ? m.sj[(size_t)NG * P + colo] : 0.0;
        auto fetch = [&](int j, FluxRow &R) {
            const size_t c = (size_t)(j + NG) * P + colo;
            const size_t c_si = SEP ? (size_t)(j + NG) * P + CO : c;
            R.u = f.u[c]; R.v = f.v[c];
            R.wri = m.wri[c]; R.wnj = m.wnj[c];
            if (RN) { R.nix = 0.0; R.niy = 0.0; R.njx = 0.0; R.njy = 0.0; }
            else { R.nix = m.nix[c]; R.niy = m.niy[c]; R.njx = m.njx[c]; R.njy = m.njy[c]; }
            R.vol = m.vol[c];
            if (MODE == 3) {
                R.p = 0.0; R.ap = 1.0; R.si = 0.0; R.sj = 0.0; R.k = 0.0;
                R.lmin = ma.use_diff ? m.lmin[c] : 1.0;
                R.nut = ma.use_diff ? ma.nut[c] : 0.0;
            } else if (MODE == 4) {
                R.k = ma.kgrad[c];
                R.w = ma.omega[c];
                R.nut = ma.nut[c];
                R.p = f.p[c] + (2.0 / 3.0) * R.k;        // p_eff (peff())
                R.ap = rc ? f.ap[c] : 1.0;
                R.si = m.si[c_si]; R.sj = SEP ? sj_col : m.sj[c];
            } else {
                R.p = f.p[c];
                R.ap = rc ? f.ap[c] : 1.0;
                R.si = m.si[c_si]; R.sj = SEP ? sj_col : m.sj[c];
                R.k = KG ? (2.0 / 3.0) * ma.kgrad[c] : 0.0;
                if (DT) R.lmin = ma.use_diff ? m.lmin[c] : 1.0;
            }
        };
        FluxRow nxt, cur{}, prv{};
        double u2 = 0.0, v2 = 0.0, p2 = 0.0, k2 = 0.0, w2 = 0.0;      // row j-2 (MODE 2, 4)
        double fw_p = 0.0, fs_p = 0.0;                       // west / south flux of row j-1
        double lw_p = 0.0, ls_p = 0.0;                       // their interpolated parts (DT)
        fetch(j0 - 1, nxt);
        for (int j = j0 - 1; j <= j1; ++j) {
            cur = nxt;
            if (j < j1) fetch(j + 1, nxt);
            // west-face flux of row j (flux_i)
            const double uL = lane_left(cur.u), vL = lane_left(cur.v), pL = lane_left(cur.p);
            double fw_c, lw_c, ls_c;
            {
                const double wr = cur.wri, wl = 1.0 - wr;
                const double ui = wl * uL + wr * cur.u;
                const double vi = wl * vL + wr * cur.v;
                fw_c = ui * FX_NIX(cur) + vi * FX_NIY(cur);
                lw_c = fw_c;
                if (rc) {
                    const double apL = lane_left(cur.ap);
                    double af = ilo ? cur.ap : (ihi ? apL : wl * apL + wr * cur.ap);
                    af = af > TINY ? af : TINY;
                    const double s = cur.si;
                    fw_c = fw_c - (s * s) / af * (cur.p - pL);
                }
            }
            // south-face flux of row j (flux_j); face j = nj is the north face of the last row
            double fs_c;
            {
                const bool lo = g.blo(j), hi = (j == g.nj) && !g.cuthi;
                const double wn = cur.wnj, ws = 1.0 - wn;
                const double uj = ws * prv.u + wn * cur.u;
                const double vj = ws * prv.v + wn * cur.v;
                fs_c = uj * FX_NJX(cur) + vj * FX_NJY(cur);
                ls_c = fs_c;
                if (rc) {
                    double af = lo ? cur.ap : (hi ? prv.ap : ws * prv.ap + wn * cur.ap);
                    af = af > TINY ? af : TINY;
                    const double s = cur.sj;
                    fs_c = fs_c - (s * s) / af * (cur.p - prv.p);
                }
            }
            // output row jo = j - 1: cell data in prv, north neighbour in cur, south neighbour in *2
            const int jo = j - 1;
            const double fe_p = lane_right(fw_p);
            const double le_p = DT ? lane_right(lw_p) : 0.0;
            if (MODE == 2) {
                // DPP shifts must run in all lanes: hoisted out of the output branch
                const double uW = lane_left(prv.u), uE = lane_right(prv.u);
                const double vW = lane_left(prv.v), vE = lane_right(prv.v);
                const double pW = lane_left(prv.p), pE = lane_right(prv.p);
                const double wriE = lane_right(prv.wri), nixE = lane_right(FX_NIX(prv)), niyE = RN ? 0.0 : lane_right(prv.niy);
                const double kW = KG ? lane_left(prv.k) : 0.0, kE = KG ? lane_right(prv.k) : 0.0;
                if (jo >= j0 && out) {
                    const double fw = fw_p, fe = fe_p, fs = fs_p, fn = fs_c;
                    const double vol = prv.vol;
                    const size_t c = (size_t)(jo + NG) * P + colo;
                    // upwind (solver.py:649-669)
                    double advu, advv;
                    {
                        const double pc = prv.u;
                        const double pw = fw >= 0.0 ? uW : pc, pe = fe >= 0.0 ? pc : uE;
                        const double ps = fs >= 0.0 ? u2 : pc, pn = fn >= 0.0 ? pc : cur.u;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advu = -div / vol;
                    }
                    {
                        const double pc = prv.v;
                        const double pw = fw >= 0.0 ? vW : pc, pe = fe >= 0.0 ? pc : vE;
                        const double ps = fs >= 0.0 ? v2 : pc, pn = fn >= 0.0 ? pc : cur.v;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advv = -div / vol;
                    }
                    // Green-Gauss gradient of p (and of 2/3 k) (solver.py:322-358)
                    double gx, gy;
                    {
                        const double pc = prv.p;
                        const double pw = (1.0 - prv.wri) * pW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * pE;
                        const double ps = (1.0 - prv.wnj) * p2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.p;
                        gx = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        gy = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                    }
                    if (KG) {
                        const double pc = prv.k;
                        const double pw = (1.0 - prv.wri) * kW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * kE;
                        const double ps = (1.0 - prv.wnj) * k2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.k;
                        const double kx = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        const double ky = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                        gx = gx + kx;
                        gy = gy + ky;
                    }
                    const double su = advu - gx + ma.force_u, sv = advv - gy + ma.force_v;
                    double b_u = vol * ma.inv_dt * prv.u + vol * su;
                    double b_v = vol * ma.inv_dt * prv.v + vol * sv;
                    if (!ma.per_i) {
                        if (iraw == 0) { const double a = ma.ai[c]; b_u += a * uW; b_v += a * vW; }
                        if (iraw == g.ni - 1) { const double a = ma.ai[c + 1]; b_u += a * uE; b_v += a * vE; }
                    }
                    if (!ma.per_j) {
                        if (g.blo(jo)) { const double a = ma.aj[c]; b_u += a * u2; b_v += a * v2; }
                        if (g.bhi(jo)) { const double a = ma.aj[c + P]; b_u += a * cur.u; b_v += a * cur.v; }
                    }
                    store_stream(ma.bu + c, b_u, ma.nts);
                    store_stream(ma.bv + c, b_v, ma.nts);
                    const double div = (fe - fw + fn - fs) / vol;
                    acc += div * div;
                    if (DT) {
                        if (ma.use_cfl) {
                            const double tot = fabs(lw_p) + fabs(le_p) + fabs(ls_p) + fabs(ls_c);
                            const double safe = tot > TINY ? tot : TINY;
                            acc2 = fmin(acc2, ma.cfl * vol / safe);
                        }
                        if (ma.use_diff) {
                            const double ne = ma.nu + 0.0;      // laminar: the nu_t plane is zero
                            const double ns = ne > 0.0 ? ne : TINY;
                            const double l = prv.lmin;
                            acc3 = fmin(acc3, ma.cfl_diff * (l * l) / ns);
                        }
                    }
                }
                u2 = prv.u; v2 = prv.v; p2 = prv.p; k2 = prv.k;
            } else if (MODE == 4) {
                const double uW = lane_left(prv.u), uE = lane_right(prv.u);
                const double vW = lane_left(prv.v), vE = lane_right(prv.v);
                const double kW = lane_left(prv.k), kE = lane_right(prv.k);
                const double wW = lane_left(prv.w), wE = lane_right(prv.w);
                const double wriE = lane_right(prv.wri), nixE = lane_right(FX_NIX(prv)), niyE = RN ? 0.0 : lane_right(prv.niy);
                if (jo >= j0 && out) {
                    const double fw = fw_p, fe = fe_p, fs = fs_p, fn = fs_c;
                    const double vol = prv.vol;
                    const size_t c = (size_t)(jo + NG) * P + colo;
                    double advk, advw;
                    {
                        const double pc = prv.k;
                        const double pw = fw >= 0.0 ? kW : pc, pe = fe >= 0.0 ? pc : kE;
                        const double ps = fs >= 0.0 ? k2 : pc, pn = fn >= 0.0 ? pc : cur.k;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advk = -div / vol;
                    }
                    {
                        const double pc = prv.w;
                        const double pw = fw >= 0.0 ? wW : pc, pe = fe >= 0.0 ? pc : wE;
                        const double ps = fs >= 0.0 ? w2 : pc, pn = fn >= 0.0 ? pc : cur.w;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advw = -div / vol;
                    }
                    double ux, uy, vx, vy;
                    {
                        const double pc = prv.u;
                        const double pw = (1.0 - prv.wri) * uW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * uE;
                        const double ps = (1.0 - prv.wnj) * u2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.u;
                        ux = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        uy = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                    }
                    {
                        const double pc = prv.v;
                        const double pw = (1.0 - prv.wri) * vW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * vE;
                        const double ps = (1.0 - prv.wnj) * v2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.v;
                        vx = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        vy = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                    }
                    const double s2 = 2.0 * (ux * ux) + 2.0 * (vy * vy) + (uy + vx) * (uy + vx);
                    const double ks = fmax(prv.k, K_FLOOR), os = fmax(prv.w, OMEGA_FLOOR);
                    const double srck = prv.nut * s2 - 0.09 * ks * os;
                    const double srcw = (5.0 / 9.0) * s2 - (3.0 / 40.0) * (os * os);
                    double b_k = vol * ma.inv_dt * prv.k + vol * (advk + srck);
                    double b_w = vol * ma.inv_dt * prv.w + vol * (advw + srcw);
                    if (!ma.per_i) {
                        if (iraw == 0) { const double a = ma.ai[c]; b_k += a * kW; b_w += a * wW; }
                        if (iraw == g.ni - 1) { const double a = ma.ai[c + 1]; b_k += a * kE; b_w += a * wE; }
                    }
                    if (!ma.per_j) {
                        if (g.blo(jo)) { const double a = ma.aj[c]; b_k += a * k2; b_w += a * w2; }
                        if (g.bhi(jo)) { const double a = ma.aj[c + P]; b_k += a * cur.k; b_w += a * cur.w; }
                    }
                    store_stream(ma.bu + c, b_k, ma.nts);
                    store_stream(ma.bv + c, b_w, ma.nts);
                }
                u2 = prv.u; v2 = prv.v; k2 = prv.k; w2 = prv.w;
            } else if (MODE == 3) {
                if (jo >= j0 && out) {
                    if (ma.use_cfl) {
                        const double tot = fabs(fw_p) + fabs(fe_p) + fabs(fs_p) + fabs(fs_c);
                        const double safe = tot > TINY ? tot : TINY;
                        acc = fmin(acc, ma.cfl * prv.vol / safe);
                    }
                    if (ma.use_diff) {
                        const double ne = ma.nu + prv.nut;
                        const double ns = ne > 0.0 ? ne : TINY;
                        const double l = prv.lmin;
                        acc2 = fmin(acc2, ma.cfl_diff * (l * l) / ns);
                    }
                }
            } else if (jo >= j0 && out) {
                const double vol = prv.vol;
                const double div = (fe_p - fw_p + fs_c - fs_p) / vol;
                if (MODE == 0) acc += div * div;
                else { const double bv = vol * (-div); store_stream(bout + (size_t)(jo + NG) * P + colo, bv, ma.nts); acc += bv; accb += bv * bv; }   // sum b (nullspace form), sum b^2
            }
            fw_p = fw_c; fs_p = fs_c;
            lw_p = lw_c; ls_p = ls_c;
            prv = cur;
            if (WPB > 4) __syncthreads();       // lockstep: j0, j1 depend on the segment only
        }
    }
    auto min_waves = [](const double *a) {
        double r = a[0];
#pragma unroll
        for (int q = 1; q < WPB; ++q) r = fmin(r, a[q]);
        return r;
    };
    if (MODE == 3) {
        __shared__ double sa[WPB], sb[WPB];
        const double wa = wave_min(acc), wb = wave_min(acc2);
        if (lane == 0) { sa[wave] = wa; sb[wave] = wb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[blockIdx.x] = min_waves(sa);
            partials[MAX_PARTIALS + blockIdx.x] = min_waves(sb);
        }
    } else {
        __shared__ double sm[WPB], sa[WPB], sb[WPB];
        const double w = wave_sum(acc);
        const double wa = DT ? wave_min(acc2) : (MODE == 1 ? wave_sum(accb) : 0.0), wb = DT ? wave_min(acc3) : 0.0;
        if (lane == 0) { sm[wave] = w; if (DT || MODE == 1) { sa[wave] = wa; sb[wave] = wb; } }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[blockIdx.x] = sum_waves<WPB>(sm);
            if (MODE == 1) partials[MAX_PARTIALS + blockIdx.x] = sum_waves<WPB>(sa);
            if (DT) {
                partials[MAX_PARTIALS + blockIdx.x] = min_waves(sa);
                partials[2 * MAX_PARTIALS + blockIdx.x] = min_waves(sb);
            }
        }
    }
}

// ----------------------------------------------------------------------------------
// Correction + boundary fill + closing flux pass of a SIMPLE iteration in ONE marching pass: k_correct (136 B/cell)
// and k_flux_march<2, ., DT> (128 B/cell) share the metric planes, V and a_P and hand u, v, p to each other through
// HBM; here a wave corrects a row as it enters its window (u = u0 + ru ((u* - V/a_P grad p') - u0), ..., the
// expressions of k_correct / green_gauss), fills the ghost cells it needs from the BC tables in registers (the
// expressions of k_flow_bc: rows -1 and nj from the corrected rows 0 and nj - 1, ghost columns from the neighbouring
// lane; a periodic i-direction needs nothing, the halo lanes correct the wrapped cells themselves), stores the
// corrected interior and runs the MODE 2 body of k_flux_march on the registers: 17 planes read, 5 written = 176 B/cell
// instead of 264.  Two halo lanes / rows per side (the gradient of p' needs one more than the fluxes).  The memory
// ghosts of u, v, p are filled by k_flow_bc afterwards, as before.  u*, v* are read by neighbouring waves, so the
// corrected u, v go to two other planes (the caller rebinds the plane pointers).
// Conditions (launch_correct_flux checks them): no periodicity along j, every entry of the j-side tables
// AFFINE / SYMMETRY (u, v) and AFFINE (p), the i-sides either periodic (with the p wrap) or like the j-sides.
// j-slabs: a cut is to this kernel what a segment boundary is -- the wave corrects the neighbour's row beyond the cut
// itself (row -1 / nj, bitwise what the neighbour computes: same expressions on the same operands) instead of
// building a ghost row.  That takes p' two rows deep, u*, v*, u0, v0, p0, a_P (and k) one row deep and the metric
// planes in the ghost rows (once per mesh: launch_correct_flux).
// ----------------------------------------------------------------------------------
constexpr int CFW = 60;
struct CorrIn {
    const double *us, *vs, *u0, *v0, *p0, *pc, *ap;
    double *uo, *vo, *po;
    double ru, rv, rp;
    int rc, per_i, nts;
};
// a raw row in two parts: the correction of row r needs E of the rows r, r + 1 but L of row r only, so E is
// requested one row earlier than L and fewer registers wait for memory
struct CorrE {
    double pc, wnj, njx, njy;
};
struct CorrL {
    double us, vs, u0, v0, p0, ap, vol;
    double wri, nix, niy, si, sj;      // west / south face of the cell (lane ihi: west face = face ni)
    double k, lmin;
};
__device__ __forceinline__ void bc_ghost(const FlowBCSide &s, int t, double ui, double vi, double pi, double &u, double &v, double &p) {
    const int kuv = s.kind_uv[t];
    if (kuv == CFD_BC_AFFINE) {
        u = s.ua[t] * ui + s.ub[t];
        v = s.va[t] * vi + s.vb[t];
    } else {        // CFD_BC_SYMMETRY
        const double nx = s.nx[t], ny = s.ny[t];
        const double un = ui * nx + vi * ny;
        u = ui - 2.0 * un * nx;
        v = vi - 2.0 * un * ny;
    }
    p = s.pa[t] * pi + s.pb[t];
}

// RN: see k_flux_march.  The area S_j of a row's south face then travels with the early part of the row (CorrE, in place
// of n_jy), so that the correction of row r still waits for no load of row r + 1's late part.
template <bool KG, bool DT, int WPB, bool LOCK = (WPB > 4), int RNA = 0>
__global__ __launch_bounds__(64 * WPB) void k_correct_flux(Geo g, MeshDev m, CorrIn ci, FlowBCDev bc, FluxMarch fm, MomArgs ma,
                                                           double *partials) {
    constexpr bool RN = RNA >= 1, SEP = RNA == 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t P = g.pitch;
    const int rc = ci.rc;
    double acc = 0.0, acc2 = INFINITY, acc3 = INFINITY;
    const MarchItems<WPB, LOCK> items(fm.nstrips, fm.nsegs, 0, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, fm.nstrips, wave, strip, seg, active);
        const int iraw = strip * CFW - 2 + lane;          // -2 .. ni + 1
        // cell whose fields this lane corrects: the wrapped cell (periodic i) or the nearest interior cell (the value
        // of a ghost lane is then replaced from the BC tables)
        int cw = iraw;
        if (ci.per_i) { if (cw < 0) cw += g.ni; if (cw >= g.ni) cw -= g.ni; }
        else cw = cw < 0 ? 0 : (cw >= g.ni ? g.ni - 1 : cw);
        if (cw < 0) cw = 0;
        if (cw >= g.ni) cw = g.ni - 1;
        const size_t colw = (size_t)cw + CO;
        // p' keeps its own ghost columns (k_pcorr_bc): a non-periodic boundary cell reads the ghost value
        const int cp = ci.per_i ? cw : (iraw < -1 ? -1 : (iraw > g.ni ? g.ni : iraw));
        const size_t colp = (size_t)(cp + CO);
        const bool out = lane >= 2 && lane < 62 && iraw < g.ni && active;
        const bool ilo = iraw == 0, ihi = iraw == g.ni;
        const bool glo = !ci.per_i && iraw == -1, ghi = !ci.per_i && iraw == g.ni;      // ghost lanes of a non-periodic i-direction
        const int j0 = seg * fm.Ls, j1 = min(j0 + fm.Ls, g.nj);
        const int rlo = g.cutlo ? -2 : -1, rhi = g.cuthi ? g.nj + 1 : g.nj;      // rows the planes hold
        const double sj_col = SEP ? m.sj[(size_t)NG * P + colw] : 0.0;
        auto fetchE = [&](int r, CorrE &R) {
            const int rr = r < rlo ? rlo : (r > rhi ? rhi : r);
            const size_t c = (size_t)(rr + NG) * P + colw;
            R.pc = ci.pc[(size_t)(rr + NG) * P + colp];
            R.wnj = m.wnj[c];
            if (RN) { R.njx = 0.0; R.njy = SEP ? sj_col : m.sj[c]; }          // (S_j in the slot of n_jy)
            else { R.njx = m.njx[c]; R.njy = m.njy[c]; }
        };
        auto kload = [&](int rr) -> double {
            return KG ? (2.0 / 3.0) * ma.kgrad[(size_t)(rr + NG) * P + (size_t)((iraw < g.ni ? (iraw < -1 ? -1 : iraw) : g.ni) + CO)] : 0.0;
        };
        auto fetchL = [&](int r, CorrL &R) {
            const int rr = r < rlo ? rlo : (r > rhi ? rhi : r);
            const size_t c = (size_t)(rr + NG) * P + colw;
            R.us = ci.us[c]; R.vs = ci.vs[c]; R.u0 = ci.u0[c]; R.v0 = ci.v0[c]; R.p0 = ci.p0[c];
            R.ap = ci.ap[c]; R.vol = m.vol[c];
            R.wri = m.wri[c]; R.si = m.si[SEP ? (size_t)(rr + NG) * P + CO : c];
            if (RN) { R.nix = 0.0; R.niy = 0.0; R.sj = 0.0; }
            else { R.nix = m.nix[c]; R.niy = m.niy[c]; R.sj = m.sj[c]; }
            R.k = kload(rr);
            R.lmin = (DT && ma.use_diff) ? m.lmin[c] : 1.0;
            if (ihi) {       // the west face of this lane is face ni, whose entries are separate from face 0's
                const size_t ce = (size_t)(rr + NG) * P + (size_t)(g.ni + CO);
                R.wri = m.wri[ce];
                if (!SEP) R.si = m.si[ce];
                if (!RN) { R.nix = m.nix[ce]; R.niy = m.niy[ce]; }
            }
        };
        // raw window around row r: pcA = p' of row r - 1; Eb, Ec, Ed = E of rows r, r + 1, r + 2 (Ed in flight);
        // Lb, Lc = L of rows r, r + 1 (Lc in flight)
        double pcA;
        CorrE Eb, Ec, Ed;
        CorrL Lb, Lc;
        auto correct_row = [&](int r, FluxRow &R) {       // r in [0, nj) (-1 / nj beyond a cut), in the form the flux stage reads
            const double pcc = Eb.pc;
            const double pW = lane_left(pcc), pE = lane_right(pcc);
            // east face of the cell = west face of the right lane (lane ihi holds face ni, the east face of cell ni - 1)
            const double wriE = lane_right(Lb.wri), nixE = lane_right(RN ? Lb.si : Lb.nix), niyE = RN ? 0.0 : lane_right(Lb.niy);
            // lane ihi itself is a ghost (non-periodic, value replaced below) or the wrapped cell 0, whose own west face
            // is face 0
            double wri = Lb.wri, nix = RN ? Lb.si : Lb.nix, niy = RN ? 0.0 : Lb.niy;
            if (ihi && ci.per_i) {
                const size_t c0 = (size_t)(r + NG) * P + colw;
                wri = m.wri[c0];
                if (SEP) nix = Lb.si;
                else if (RN) nix = m.si[c0];
                else { nix = m.nix[c0]; niy = m.niy[c0]; }
            }
            const double pw = (1.0 - wri) * pW + wri * pcc;
            const double pe = (1.0 - wriE) * pcc + wriE * pE;
            const double ps = (1.0 - Eb.wnj) * pcA + Eb.wnj * pcc;
            const double pn = (1.0 - Ec.wnj) * pcc + Ec.wnj * Ec.pc;
            const double vol = Lb.vol;
            // (RN: E.njx = 0 and E.njy = S_j as loaded)
            const double gx = (nixE * pe - nix * pw + Ec.njx * pn - Eb.njx * ps) / vol;
            const double gy = (niyE * pe - niy * pw + Ec.njy * pn - Eb.njy * ps) / vol;
            const double d = vol / Lb.ap;
            const double uc = Lb.us - d * gx, vc = Lb.vs - d * gy;
            double un = Lb.u0 + ci.ru * (uc - Lb.u0);
            double vn = Lb.v0 + ci.rv * (vc - Lb.v0);
            double pn2 = Lb.p0 + ci.rp * pcc;
            if (out && r >= j0 && r < j1) {
                const size_t c = (size_t)(r + NG) * P + colw;
                store_stream(ci.uo + c, un, ci.nts);
                store_stream(ci.vo + c, vn, ci.nts);
                store_stream(ci.po + c, pn2, ci.nts);
            }
            // ghost columns of a non-periodic i-direction from the neighbouring interior lane (k_flow_bc)
            const double uR = lane_right(un), vR = lane_right(vn), pR = lane_right(pn2);
            const double uL = lane_left(un), vL = lane_left(vn), pL = lane_left(pn2);
            // (rows beyond a cut have no table entry of this rank: their ghost columns are read by no cell that is stored)
            const int tr = r < 0 ? 0 : (r >= g.nj ? g.nj - 1 : r);
            if (glo) bc_ghost(bc.side[CFD_SIDE_IMIN], tr, uR, vR, pR, un, vn, pn2);
            if (ghi) bc_ghost(bc.side[CFD_SIDE_IMAX], tr, uL, vL, pL, un, vn, pn2);
            R.u = un; R.v = vn; R.p = pn2; R.ap = rc ? Lb.ap : 1.0;
            R.wri = Lb.wri; R.nix = Lb.nix; R.niy = Lb.niy; R.si = Lb.si;
            R.wnj = Eb.wnj; R.njx = Eb.njx; R.njy = Eb.njy; R.sj = RN ? Eb.njy : Lb.sj;
            R.vol = Lb.vol; R.k = Lb.k; R.lmin = Lb.lmin;
        };
        // first row this segment needs corrected
        const bool below = j0 > 0 || g.cutlo;        // there is a row below the segment (another segment's or the neighbour slab's)
        const int r0 = below ? j0 - 1 : 0;
        FluxRow cur{}, prv{};
        {
            CorrE A;
            fetchE(r0 - 1, A);
            pcA = A.pc;
            fetchE(r0, Eb);
            fetchE(r0 + 1, Ec);
            fetchE(r0 + 2, Ed);
            fetchL(r0, Lb);
            fetchL(r0 + 1, Lc);
        }
        int wrow = r0;                        // the row Eb / Lb hold
        bool have_cur = false;
        if (!below) {
            // the ghost row -1 stands in prv: the flux stage reads only its u, v, p (and k), from the corrected row 0
            correct_row(0, cur);
            prv = cur;
            prv.k = kload(-1);
            bc_ghost(bc.side[CFD_SIDE_JMIN], cw, cur.u, cur.v, cur.p, prv.u, prv.v, prv.p);
            have_cur = true;
        }
        double u2 = 0.0, v2 = 0.0, p2 = 0.0, k2 = 0.0;
        double fw_p = 0.0, fs_p = 0.0, lw_p = 0.0, ls_p = 0.0;
        for (int j = r0; j <= j1; ++j) {
            if (j < g.nj || g.cuthi) {
                if (!have_cur) {
                    if (wrow != j) {          // window -> row j
                        pcA = Eb.pc; Eb = Ec; Ec = Ed; Lb = Lc;
                        fetchE(j + 2, Ed);
                        fetchL(j + 1, Lc);
                        wrow = j;
                    }
                    correct_row(j, cur);
                }
                have_cur = false;
            } else {
                // ghost row nj from the corrected row nj - 1 (= prv); its south-face entries and k are row nj's
                bc_ghost(bc.side[CFD_SIDE_JMAX], cw, prv.u, prv.v, prv.p, cur.u, cur.v, cur.p);
                cur.wnj = Ec.wnj; cur.njx = Ec.njx; cur.njy = Ec.njy; cur.sj = RN ? Ec.njy : Lc.sj; cur.k = Lc.k;
            }
            // ---- from here on: the MODE 2 body of k_flux_march, same expressions
            const double uL = lane_left(cur.u), vL = lane_left(cur.v), pL = lane_left(cur.p);
            double fw_c, lw_c, ls_c;
            {
                const double wr = cur.wri, wl = 1.0 - wr;
                const double ui = wl * uL + wr * cur.u;
                const double vi = wl * vL + wr * cur.v;
                fw_c = ui * FX_NIX(cur) + vi * FX_NIY(cur);
                lw_c = fw_c;
                if (rc) {
                    const double apL = lane_left(cur.ap);
                    double af = ilo ? cur.ap : (ihi ? apL : wl * apL + wr * cur.ap);
                    af = af > TINY ? af : TINY;
                    const double s = cur.si;
                    fw_c = fw_c - (s * s) / af * (cur.p - pL);
                }
            }
            double fs_c;
            {
                const bool lo = g.blo(j), hi = (j == g.nj) && !g.cuthi;
                const double wn = cur.wnj, ws = 1.0 - wn;
                const double uj = ws * prv.u + wn * cur.u;
                const double vj = ws * prv.v + wn * cur.v;
                fs_c = uj * FX_NJX(cur) + vj * FX_NJY(cur);
                ls_c = fs_c;
                if (rc) {
                    double af = lo ? cur.ap : (hi ? prv.ap : ws * prv.ap + wn * cur.ap);
                    af = af > TINY ? af : TINY;
                    const double s = cur.sj;
                    fs_c = fs_c - (s * s) / af * (cur.p - prv.p);
                }
            }
            const int jo = j - 1;
            const double fe_p = lane_right(fw_p);
            const double le_p = DT ? lane_right(lw_p) : 0.0;
            {
                const double uW = lane_left(prv.u), uE = lane_right(prv.u);
                const double vW = lane_left(prv.v), vE = lane_right(prv.v);
                const double pW = lane_left(prv.p), pE = lane_right(prv.p);
                const double wriE = lane_right(prv.wri), nixE = lane_right(FX_NIX(prv)), niyE = RN ? 0.0 : lane_right(prv.niy);
                const double kW = KG ? lane_left(prv.k) : 0.0, kE = KG ? lane_right(prv.k) : 0.0;
                if (jo >= j0 && out) {
                    const double fw = fw_p, fe = fe_p, fs = fs_p, fn = fs_c;
                    const double vol = prv.vol;
                    const size_t c = (size_t)(jo + NG) * P + colw;
                    double advu, advv;
                    {
                        const double pc = prv.u;
                        const double pw = fw >= 0.0 ? uW : pc, pe = fe >= 0.0 ? pc : uE;
                        const double ps = fs >= 0.0 ? u2 : pc, pn = fn >= 0.0 ? pc : cur.u;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advu = -div / vol;
                    }
                    {
                        const double pc = prv.v;
                        const double pw = fw >= 0.0 ? vW : pc, pe = fe >= 0.0 ? pc : vE;
                        const double ps = fs >= 0.0 ? v2 : pc, pn = fn >= 0.0 ? pc : cur.v;
                        const double div = fe * pe - fw * pw + fn * pn - fs * ps;
                        advv = -div / vol;
                    }
                    double gx, gy;
                    {
                        const double pc = prv.p;
                        const double pw = (1.0 - prv.wri) * pW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * pE;
                        const double ps = (1.0 - prv.wnj) * p2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.p;
                        gx = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        gy = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                    }
                    if (KG) {
                        const double pc = prv.k;
                        const double pw = (1.0 - prv.wri) * kW + prv.wri * pc;
                        const double pe = (1.0 - wriE) * pc + wriE * kE;
                        const double ps = (1.0 - prv.wnj) * k2 + prv.wnj * pc;
                        const double pn = (1.0 - cur.wnj) * pc + cur.wnj * cur.k;
                        const double kx = (nixE * pe - FX_NIX(prv) * pw + FX_NJX(cur) * pn - FX_NJX(prv) * ps) / vol;
                        const double ky = (niyE * pe - FX_NIY(prv) * pw + FX_NJY(cur) * pn - FX_NJY(prv) * ps) / vol;
                        gx = gx + kx;
                        gy = gy + ky;
                    }
                    const double su = advu - gx + ma.force_u, sv = advv - gy + ma.force_v;
                    double b_u = vol * ma.inv_dt * prv.u + vol * su;
                    double b_v = vol * ma.inv_dt * prv.v + vol * sv;
                    if (!ma.per_i) {
                        if (iraw == 0) { const double a = ma.ai[c]; b_u += a * uW; b_v += a * vW; }
                        if (iraw == g.ni - 1) { const double a = ma.ai[c + 1]; b_u += a * uE; b_v += a * vE; }
                    }
                    if (!ma.per_j) {
                        if (g.blo(jo)) { const double a = ma.aj[c]; b_u += a * u2; b_v += a * v2; }
                        if (g.bhi(jo)) { const double a = ma.aj[c + P]; b_u += a * cur.u; b_v += a * cur.v; }
                    }
                    store_stream(ma.bu + c, b_u, ma.nts);
                    store_stream(ma.bv + c, b_v, ma.nts);
                    const double div = (fe - fw + fn - fs) / vol;
                    acc += div * div;
                    if (DT) {
                        if (ma.use_cfl) {
                            const double tot = fabs(lw_p) + fabs(le_p) + fabs(ls_p) + fabs(ls_c);
                            const double safe = tot > TINY ? tot : TINY;
                            acc2 = fmin(acc2, ma.cfl * vol / safe);
                        }
                        if (ma.use_diff) {
                            const double ne = ma.nu + 0.0;
                            const double ns = ne > 0.0 ? ne : TINY;
                            const double l = prv.lmin;
                            acc3 = fmin(acc3, ma.cfl_diff * (l * l) / ns);
                        }
                    }
                }
                u2 = prv.u; v2 = prv.v; p2 = prv.p; k2 = prv.k;
            }
            fw_p = fw_c; fs_p = fs_c;
            lw_p = lw_c; ls_p = ls_c;
            prv = cur;
            if (LOCK) __syncthreads();
        }
    }
    auto min_waves = [](const double *a) {
        double r = a[0];
#pragma unroll
        for (int q = 1; q < WPB; ++q) r = fmin(r, a[q]);
        return r;
    };
    __shared__ double sm[WPB], sa[WPB], sb[WPB];
    const double w = wave_sum(acc);
    const double wa = DT ? wave_min(acc2) : 0.0, wb = DT ? wave_min(acc3) : 0.0;
    if (lane == 0) { sm[wave] = w; if (DT) { sa[wave] = wa; sb[wave] = wb; } }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = sum_waves<WPB>(sm);
        if (DT) {
            partials[MAX_PARTIALS + blockIdx.x] = min_waves(sa);
            partials[2 * MAX_PARTIALS + blockIdx.x] = min_waves(sb);
        }
    }
}

constexpr int DT_AHEAD_OFF = 96;      // Ctx::scal slots of the time-step minima computed ahead

int launch_dt(Ctx *c, int use_cfl, double cfl, int use_cfl_diff, double cfl_diff, double out[2]) {
    const bool same = c->dtp_set && c->dtp_use_cfl == use_cfl && c->dtp_use_diff == use_cfl_diff && c->dtp_cfl == cfl &&
                      c->dtp_cfl_diff == cfl_diff;
    c->dtp_set = true; c->dtp_use_cfl = use_cfl; c->dtp_use_diff = use_cfl_diff; c->dtp_cfl = cfl; c->dtp_cfl_diff = cfl_diff;
    if (c->dt_ahead && same && c->dt_ahead_cur == c->cur) {
        // the last finished iteration evaluated the minima of these very fields (k_flux_march<2, ., DT>); in k-omega /
        // SST runs nu_t has changed since, so the diffusive one is taken now (nu_t and lmin only: 16 B/cell)
        c->dt_ahead = false;
        if (use_cfl_diff && !c->dt_ahead_diff) {
            int tx;
            const int tiles = tiles_of(c->g, tx), nb = red_blocks(tiles);
            hipLaunchKernelGGL(k_dt, dim3(nb), block2d(), 0, c->stream, c->g, c->mesh(), c->U(), c->V(), c->nut, c->nu, 0, cfl, 1,
                               cfl_diff, c->partials, tx, tiles);
            hipLaunchKernelGGL(k_min_finish, dim3(1), dim3(256), 0, c->stream, c->partials, nb, 2, c->scal);
            CFD_HIP(hipGetLastError());
            CFD_HIP(hipMemcpyAsync(c->scal + DT_AHEAD_OFF + 1, c->scal + 1, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        }
        if (c->dt_host_valid && !(use_cfl_diff && !c->dt_ahead_diff)) {
            // already on the host (read together with the continuity residual of the pass that computed them)
            c->dt_host_valid = false;
            out[0] = c->h_dt_ahead[0]; out[1] = c->h_dt_ahead[1];
            if (use_cfl_diff) dt_diff_keep(c, out[1]);
            return CFD_OK;
        }
        if (c->dt_host_valid) {
            // the advective minimum is on the host and reduced already; the diffusive one was just recomputed
            c->dt_host_valid = false;
            CFD_TRY(all_reduce(c, c->scal + DT_AHEAD_OFF + 1, 1, 1));
            CFD_TRY(read_scalars(c, DT_AHEAD_OFF + 1, 1, out + 1));
            out[0] = c->h_dt_ahead[0];
            return CFD_OK;
        }
        CFD_TRY(all_reduce(c, c->scal + DT_AHEAD_OFF, 2, 1));
        CFD_TRY(read_scalars(c, DT_AHEAD_OFF, 2, out));
        if (use_cfl_diff && c->dt_ahead_diff) dt_diff_keep(c, out[1]);
        return CFD_OK;
    }
    c->dt_pass_diff_fresh = false;
    int nb;
    if (c->march_flux) {
        // marching form: every row of u, v and of the metric planes is read once (88 instead of ~117 B/cell)
        const FluxMarch fm = flux_march_geo(c->g);
        nb = fm.blocks;
        FluxIn f{c->U(), c->V(), nullptr, nullptr, nullptr, 0};
        MomArgs ma{};
        ma.nut = c->nut; ma.nu = c->nu; ma.cfl = cfl; ma.cfl_diff = cfl_diff; ma.use_cfl = use_cfl; ma.use_diff = use_cfl_diff;
        hipLaunchKernelGGL((k_flux_march<3, false>), dim3(nb), dim3(256), 0, c->stream, c->g, c->mesh(), f, fm, ma, (double *)nullptr,
                           c->partials);
    } else {
        int tx;
        const int tiles = tiles_of(c->g, tx);
        nb = red_blocks(tiles);
        hipLaunchKernelGGL(k_dt, dim3(nb), block2d(), 0, c->stream, c->g, c->mesh(), c->U(), c->V(), c->nut, c->nu, use_cfl,
                           cfl, use_cfl_diff, cfl_diff, c->partials, tx, tiles);
    }
    CFD_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_min_finish, dim3(1), dim3(256), 0, c->stream, c->partials, nb, 2, c->scal);
    CFD_TRY(all_reduce(c, c->scal, 2, 1));    // global minimum over the slabs
    CFD_HIP(hipGetLastError());
    return read_scalars(c, 0, 2, out);
}

// with_dt: the pass also left the time-step minima of these fields in scal[DT_AHEAD_OFF ..): the same
// synchronisation brings them to the host, so the next cfd_compute_dt call costs none
static int finish_continuity(Ctx *c, int nb, double *out, bool with_dt = false) {
    c->n_red_blocks = nb;
    CFD_TRY(reduce_finish(c, 1, 0));
    double s;
    c->dt_host_valid = false;
    if (c->spec_on) {
        // speculative iteration: this is its one host read -- everything the solves left in Ctx::scal comes along
        if (with_dt) CFD_TRY(all_reduce(c, c->scal + DT_AHEAD_OFF, 2, 1));
        CFD_TRY(read_scalars_all(c));
        s = c->h_spec[0];
        if (with_dt) {
            c->h_dt_ahead[0] = c->h_spec[DT_AHEAD_OFF]; c->h_dt_ahead[1] = c->h_spec[DT_AHEAD_OFF + 1];
            c->dt_host_valid = true;
        }
    } else if (with_dt && c->host_lag) {
        CFD_TRY(all_reduce(c, c->scal + DT_AHEAD_OFF, 2, 1));       // global minimum over the slabs
        CFD_TRY(read_scalars2(c, 0, 1, &s, DT_AHEAD_OFF, 2, c->h_dt_ahead));
        c->dt_host_valid = true;
    } else {
        CFD_TRY(read_scalars(c, 0, 1, &s));
    }
    *out = sqrt(s / global_cells(c));
    return CFD_OK;
}

// Host side of k_correct_flux: can this context take the fused pass, and the launch.
bool correct_flux_ok(const Ctx *c) {
    if (!c->fuse_correct || !c->march_flux || !c->rhs_lookahead) return false;
    if (c->per_j || c->pp_per_j || c->g.ni < c->fused_min_cf || c->g.nj < 4) return false;
    // every rank of a slab run must decide alike (the pass has its own exchanges): fuse_correct_slab is set by
    // cfd_set_flow_bc_tables' callers only through options, the side classes of the cut sides do not matter
    if (c->comm.nranks > 1 && !c->fuse_correct_slab) return false;
    if (!c->g.cutlo && c->fbc_class[CFD_SIDE_JMIN] != 2) return false;
    if (!c->g.cuthi && c->fbc_class[CFD_SIDE_JMAX] != 2) return false;
    if (c->per_i) return c->fbc_class[CFD_SIDE_IMIN] == 1 && c->fbc_class[CFD_SIDE_IMAX] == 1 && c->wrap_p_i && c->pp_per_i;
    return c->fbc_class[CFD_SIDE_IMIN] == 2 && c->fbc_class[CFD_SIDE_IMAX] == 2 && !c->pp_per_i;
}

// u*, v* in U1 / V1, snapshot in U / V / P (solver.py:1596-1598), p' (with its ghosts) in pcorr: leaves the corrected
// fields in the planes that U1 / V1 / P1 name afterwards (u, v by rebinding two work planes), b_u, b_v of the next
// iteration, the continuity sum and (laminar runs) the next step's time-step minima in the partial sums.
// The caller flips c->cur, fills the memory ghosts (k_flow_bc) and reads the sums (finish_correct_flux).
int launch_correct_flux(Ctx *c, double dt, double ru, double rv, double rp, bool *dt_too_out, int *nb_out) {
    if (c->comm.nranks > 1) {
        const Geo &g = c->g;
        if (!c->mesh_halo2) {
            // once per mesh: the neighbour's metric rows beyond the cuts (face-j planes two rows deep: row nj + 1 holds the
            // south face of the neighbour's second row)
            const HaloSpec cellp[7] = {{&g, 1, c->m_vol}, {&g, 1, c->m_wri}, {&g, 1, c->m_nix}, {&g, 1, c->m_niy},
                                       {&g, 1, c->m_si}, {&g, 1, c->m_lmin}, {&g, 2, c->m_sj}};
            const HaloSpec facep[3] = {{&g, 2, c->m_wnj}, {&g, 2, c->m_njx}, {&g, 2, c->m_njy}};
            CFD_TRY(halo_exchange_multi(c, 4, cellp));
            CFD_TRY(halo_exchange_multi(c, 3, cellp + 4));
            CFD_TRY(halo_exchange_multi(c, 3, facep));
            c->mesh_halo2 = true;
        }
        // u*, v* (flow_bc after the predictor), u0, v0, p0 (flow_bc at the start of the iteration), a_P (do_predictor) and
        // k (turbulence update) hold the neighbour's rows already
        CFD_TRY(halo_exchange(c, g, 2, c->pcorr));
    }
    FluxMarch fm = flux_march_geo(c->g);
    fm.nstrips = (c->g.ni + CFW - 1) / CFW;
    {
        const long nb = ((long)fm.nstrips * fm.nsegs + 3) / 4;
        fm.blocks = (int)(nb < MAX_PARTIALS ? nb : MAX_PARTIALS);
    }
    flux_march_widen(c, fm);
    // lockstep workgroups of four strips (two workgroups per CU at this kernel's register count) measured against eight
    const bool lock4 = fm.wide == 8 && c->cf_form == 2;
    if (fm.wide == 8 && c->cf_form == 0) fm.wide = 4;
    if (lock4 || fm.wide == 4) {
        const long nb = lock4 ? (long)((fm.nstrips + 3) / 4) * fm.nsegs : ((long)fm.nstrips * fm.nsegs + 3) / 4;
        fm.blocks = (int)(nb < MAX_PARTIALS ? nb : MAX_PARTIALS);
        fm.wide = 4;
    }
    double *uo = c->kry[2], *vo = c->kry[3];
    CorrIn ci{c->U1(), c->V1(), c->U(), c->V(), c->P(), c->pcorr, c->ap, uo, vo, c->P1(), ru, rv, rp, c->rhie_chow, c->per_i, c->nts()};
    MomArgs ma{c->turbulent ? c->k : nullptr, c->ai, c->aj, 1.0 / dt, -c->dpdx, -c->dpdy, c->per_i, c->per_j, c->bu, c->bv};
    ma.nts = c->nts();
    c->kc[KC_FLUX_MARCH] += 1;
    if (fm.nstrips > 1) c->kc[KC_FLUX_MARCH_MULTISTRIP] += 1;
    c->kc[KC_CORRECT_FLUX] += 1;
    const bool dt_too = c->dt_lookahead && c->dtp_set && (c->dtp_use_cfl || !c->turbulent);
    // u*, v*, u0, v0, p0, p', a_P, V, w x 2, n x 4, S x 2 read; u, v, p, b_u, b_v written (+ k, + lmin)
    // (axis-aligned mesh: the four normal planes are not read)
    const int rn = rect_normals(c);
    const bool diff_known = dt_too && c->dtp_use_diff && dt_diff_known(c);
    c->last_rhs_bytes = 176.0 + (c->turbulent ? 8.0 : 0.0) + ((dt_too && !c->turbulent && c->dtp_use_diff && !diff_known) ? 8.0 : 0.0) - (rn ? 32.0 : 0.0) - (rn == 2 ? 16.0 : 0.0);
    if (dt_too) {
        ma.nu = c->nu; ma.cfl = c->dtp_cfl; ma.cfl_diff = c->dtp_cfl_diff; ma.use_cfl = c->dtp_use_cfl;
        ma.use_diff = (c->turbulent || diff_known) ? 0 : c->dtp_use_diff;
        c->dt_pass_diff_fresh = ma.use_diff != 0;
#define LAUNCH_CF(KGv, DTv)                                                                                                   \
    do {                                                                                                                     \
        if (rn == 2) {                                                                                                       \
            if (lock4) hipLaunchKernelGGL((k_correct_flux<KGv, DTv, 4, true, 2>), dim3(fm.blocks), dim3(256), 0, c->stream, c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials); \
            else LAUNCH_FLUX(fm, (k_correct_flux<KGv, DTv, 4, false, 2>), (k_correct_flux<KGv, DTv, 8, true, 2>), c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials); \
        } else if (rn) {                                                                                                     \
            if (lock4) hipLaunchKernelGGL((k_correct_flux<KGv, DTv, 4, true, 1>), dim3(fm.blocks), dim3(256), 0, c->stream, c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials); \
            else LAUNCH_FLUX(fm, (k_correct_flux<KGv, DTv, 4, false, 1>), (k_correct_flux<KGv, DTv, 8, true, 1>), c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials); \
        } else if (lock4) hipLaunchKernelGGL((k_correct_flux<KGv, DTv, 4, true>), dim3(fm.blocks), dim3(256), 0, c->stream, c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials); \
        else LAUNCH_FLUX(fm, (k_correct_flux<KGv, DTv, 4>), (k_correct_flux<KGv, DTv, 8>), c->g, c->mesh(), ci, c->fbc, fm, ma, c->partials);                       \
    } while (0)
        if (c->turbulent) LAUNCH_CF(true, true);
        else LAUNCH_CF(false, true);
        hipLaunchKernelGGL(k_min_finish, dim3(1), dim3(256), 0, c->stream, c->partials + MAX_PARTIALS, fm.blocks, 2,
                           c->scal + DT_AHEAD_OFF, diff_known ? c->dtdiff_val : 0.0, diff_known ? 1 : 0);
    } else if (c->turbulent) {
        LAUNCH_CF(true, false);
    } else {
        LAUNCH_CF(false, false);
    }
    CFD_HIP(hipGetLastError());
    // the corrected u, v become U1 / V1 (the planes that held u*, v* become work planes)
    std::swap(c->u[c->cur ^ 1], c->kry[2]);
    std::swap(c->v[c->cur ^ 1], c->kry[3]);
    *dt_too_out = dt_too;
    *nb_out = fm.blocks;
    return CFD_OK;
}

// after c->cur has been flipped: the bookkeeping of launch_momentum_rhs's look-ahead and the read of the sums
int finish_correct_flux(Ctx *c, bool dt_too, int nb, double *cont) {
    if (dt_too) {
        c->dt_ahead = true;
        c->dt_ahead_diff = !c->turbulent;
        c->dt_ahead_cur = c->cur;
    }
    return finish_continuity(c, nb, cont, dt_too);
}

// ----------------------------------------------------------------------------------
// momentum right-hand sides (solver.py:672-691 advective RHS, 868-876 sources,
// 723-730 transient + boundary terms), fused: nothing but b_u, b_v is written.
// kgrad != nullptr adds grad(2/3 k) to grad p (solver.py:871-872, 1621-1626).
// ----------------------------------------------------------------------------------
__global__ void k_momentum_rhs(Geo g, MeshDev m, FluxIn f, const double *kgrad, const double *ai, const double *aj,
                               double inv_dt, double force_u, double force_v, int per_i, int per_j, double *bu,
                               double *bv) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double fw = flux_i(f, m, c, i == 0, false), fe = flux_i(f, m, c + 1, false, i == g.ni - 1);
    const double fs = flux_j(f, m, c, P, g.blo(j), false), fn = flux_j(f, m, c + P, P, false, g.bhi(j));
    const double vol = m.vol[c];
    const double advu = upwind(f.u, c, P, fw, fe, fs, fn, vol);
    const double advv = upwind(f.v, c, P, fw, fe, fs, fn, vol);
    double gx, gy;
    const double *pp = f.p;
    green_gauss([pp](size_t q) { return pp[q]; }, m, c, P, gx, gy);
    if (kgrad) {
        double kx, ky;
        green_gauss([kgrad](size_t q) { return (2.0 / 3.0) * kgrad[q]; }, m, c, P, kx, ky);
        gx = gx + kx;
        gy = gy + ky;
    }
    const double su = advu - gx + force_u, sv = advv - gy + force_v;
    double b_u = vol * inv_dt * f.u[c] + vol * su;
    double b_v = vol * inv_dt * f.v[c] + vol * sv;
    if (!per_i) {
        if (i == 0) { b_u += ai[c] * f.u[c - 1]; b_v += ai[c] * f.v[c - 1]; }
        if (i == g.ni - 1) { b_u += ai[c + 1] * f.u[c + 1]; b_v += ai[c + 1] * f.v[c + 1]; }
    }
    if (!per_j) {
        if (g.blo(j)) { b_u += aj[c] * f.u[c - P]; b_v += aj[c] * f.v[c - P]; }
        if (g.bhi(j)) { b_u += aj[c + P] * f.u[c + P]; b_v += aj[c + P] * f.v[c + P]; }
    }
    bu[c] = b_u;
    bv[c] = b_v;
}

int launch_momentum_rhs(Ctx *c, double dt, const double *u, const double *v, const double *p, double *cont) {
    // the predictor's Rhie-Chow flux uses p_field = fields["p"] (solver.py:1602-1607);
    // the 2/3 k term enters through the gradient only.
    FluxIn f{u, v, p, nullptr, c->ap, c->rhie_chow};
    if (c->march_flux) {
        // marching kernel: the continuity sum of the same fluxes comes for free (cont != nullptr reads it)
        FluxMarch fm = flux_march_geo(c->g);
        flux_march_widen(c, fm);
        MomArgs ma{c->turbulent ? c->k : nullptr, c->ai, c->aj, 1.0 / dt, -c->dpdx, -c->dpdy, c->per_i, c->per_j, c->bu, c->bv};
        ma.nts = c->nts();
        c->kc[KC_FLUX_MARCH] += 1;
        if (fm.nstrips > 1) c->kc[KC_FLUX_MARCH_MULTISTRIP] += 1;
        // cont != nullptr: the pass that closes an inner iteration -- in laminar runs it also evaluates the time-step
        // minima the next outer step will ask for (same fields; parameters of the last cfd_compute_dt call)
        const bool dt_too = cont && c->dt_lookahead && c->dtp_set && (c->dtp_use_cfl || !c->turbulent);
        // 13 planes read (u, v, p, a_P, V, w x 2, n x 4, S x 2), b_u, b_v written; + k in turbulent runs; + lmin for the
        // diffusive minimum taken ahead
        const int rn = rect_normals(c);
        const bool diff_known = dt_too && c->dtp_use_diff && dt_diff_known(c);
        c->last_rhs_bytes = 120.0 + (c->turbulent ? 8.0 : 0.0) + ((dt_too && !c->turbulent && c->dtp_use_diff && !diff_known) ? 8.0 : 0.0) - (rn ? 32.0 : 0.0) - (rn == 2 ? 16.0 : 0.0);
#define LAUNCH_FM2(KGv, DTv)                                                                                                  \
    do {                                                                                                                     \
        if (rn == 2) LAUNCH_FLUX(fm, (k_flux_march<2, KGv, DTv, 4, 2>), (k_flux_march<2, KGv, DTv, 8, 2>), c->g, c->mesh(), f, fm, ma, (double *)nullptr, c->partials); \
        else if (rn) LAUNCH_FLUX(fm, (k_flux_march<2, KGv, DTv, 4, 1>), (k_flux_march<2, KGv, DTv, 8, 1>), c->g, c->mesh(), f, fm, ma, (double *)nullptr, c->partials); \
        else LAUNCH_FLUX(fm, (k_flux_march<2, KGv, DTv>), (k_flux_march<2, KGv, DTv, 8>), c->g, c->mesh(), f, fm, ma, (double *)nullptr, c->partials); \
    } while (0)
        if (dt_too) {
            // turbulent runs: nu_t changes before the next step, only the advective minimum can be taken ahead
            ma.nu = c->nu; ma.cfl = c->dtp_cfl; ma.cfl_diff = c->dtp_cfl_diff; ma.use_cfl = c->dtp_use_cfl;
            ma.use_diff = (c->turbulent || diff_known) ? 0 : c->dtp_use_diff;
            c->dt_pass_diff_fresh = ma.use_diff != 0;
            if (c->turbulent) LAUNCH_FM2(true, true);
            else LAUNCH_FM2(false, true);
            hipLaunchKernelGGL(k_min_finish, dim3(1), dim3(256), 0, c->stream, c->partials + MAX_PARTIALS, fm.blocks, 2,
                               c->scal + DT_AHEAD_OFF, diff_known ? c->dtdiff_val : 0.0, diff_known ? 1 : 0);
            c->dt_ahead = true;
            c->dt_ahead_diff = !c->turbulent;
            c->dt_ahead_cur = c->cur;
        } else if (c->turbulent) {
            LAUNCH_FM2(true, false);
        } else {
            LAUNCH_FM2(false, false);
        }
#undef LAUNCH_FM2
        CFD_HIP(hipGetLastError());
        if (cont) return finish_continuity(c, fm.blocks, cont, dt_too);
        return CFD_OK;
    }
    c->kc[KC_FLUX_TILE] += 1;
    hipLaunchKernelGGL(k_momentum_rhs, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), f,
                       c->turbulent ? c->k : nullptr, c->ai, c->aj, 1.0 / dt, -c->dpdx, -c->dpdy, c->per_i, c->per_j,
                       c->bu, c->bv);
    CFD_HIP(hipGetLastError());
    if (cont) return launch_continuity(c, u, v, p, cont);
    return CFD_OK;
}

// solve_momentum with the caller's advective right-hand sides (solver.py:848-883): adv_u, adv_v are cell planes the
// caller computed (compute_advective_rhs, solver.py:672-691), extra is an optional second pressure-like ghosted plane
// (extra_pressure_field, solver.py:871-872).  Same expression order as k_momentum_rhs.
__global__ void k_momentum_rhs_given(Geo g, MeshDev m, const double *u, const double *v, const double *p, const double *extra,
                                     const double *advu, const double *advv, const double *ai, const double *aj, double inv_dt,
                                     double force_u, double force_v, int per_i, int per_j, double *bu, double *bv) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double vol = m.vol[c];
    double gx, gy;
    green_gauss([p](size_t q) { return p[q]; }, m, c, P, gx, gy);
    if (extra) {
        double ex, ey;
        green_gauss([extra](size_t q) { return extra[q]; }, m, c, P, ex, ey);
        gx = gx + ex;
        gy = gy + ey;
    }
    const double su = advu[c] - gx + force_u, sv = advv[c] - gy + force_v;
    double b_u = vol * inv_dt * u[c] + vol * su;
    double b_v = vol * inv_dt * v[c] + vol * sv;
    if (!per_i) {
        if (i == 0) { b_u += ai[c] * u[c - 1]; b_v += ai[c] * v[c - 1]; }
        if (i == g.ni - 1) { b_u += ai[c + 1] * u[c + 1]; b_v += ai[c + 1] * v[c + 1]; }
    }
    if (!per_j) {
        if (g.blo(j)) { b_u += aj[c] * u[c - P]; b_v += aj[c] * v[c - P]; }
        if (g.bhi(j)) { b_u += aj[c + P] * u[c + P]; b_v += aj[c + P] * v[c + P]; }
    }
    bu[c] = b_u;
    bv[c] = b_v;
}

int launch_momentum_rhs_given(Ctx *c, double dt, const double *advu, const double *advv, const double *extra, double force_u,
                              double force_v) {
    hipLaunchKernelGGL(k_momentum_rhs_given, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), c->U(), c->V(),
                       c->P(), extra, advu, advv, c->ai, c->aj, 1.0 / dt, force_u, force_v, c->per_i, c->per_j, c->bu, c->bv);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// k / omega right-hand sides (solver.py:1109-1136, turbulence.py:427-444).  The flux uses
// p_eff = p + 2/3 k and the momentum a_P; boundary terms use the transport conductances.
// ----------------------------------------------------------------------------------
__global__ void k_turb_rhs(Geo g, MeshDev m, FluxIn f, const double *om, const double *nut, const double *ti,
                           const double *tj, double inv_dt, int per_i, int per_j, double *bk, double *bw) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double fw = flux_i(f, m, c, i == 0, false), fe = flux_i(f, m, c + 1, false, i == g.ni - 1);
    const double fs = flux_j(f, m, c, P, g.blo(j), false), fn = flux_j(f, m, c + P, P, false, g.bhi(j));
    const double vol = m.vol[c];
    const double *k = f.k;
    const double advk = upwind(k, c, P, fw, fe, fs, fn, vol);
    const double advw = upwind(om, c, P, fw, fe, fs, fn, vol);
    double ux, uy, vx, vy;
    const double *uu = f.u, *vv = f.v;
    green_gauss([uu](size_t q) { return uu[q]; }, m, c, P, ux, uy);
    green_gauss([vv](size_t q) { return vv[q]; }, m, c, P, vx, vy);
    const double s2 = 2.0 * (ux * ux) + 2.0 * (vy * vy) + (uy + vx) * (uy + vx);
    const double ks = fmax(k[c], K_FLOOR), os = fmax(om[c], OMEGA_FLOOR);
    const double srck = nut[c] * s2 - 0.09 * ks * os;
    const double srcw = (5.0 / 9.0) * s2 - (3.0 / 40.0) * (os * os);
    double b_k = vol * inv_dt * k[c] + vol * (advk + srck);
    double b_w = vol * inv_dt * om[c] + vol * (advw + srcw);
    if (!per_i) {
        if (i == 0) { b_k += ti[c] * k[c - 1]; b_w += ti[c] * om[c - 1]; }
        if (i == g.ni - 1) { b_k += ti[c + 1] * k[c + 1]; b_w += ti[c + 1] * om[c + 1]; }
    }
    if (!per_j) {
        if (g.blo(j)) { b_k += tj[c] * k[c - P]; b_w += tj[c] * om[c - P]; }
        if (g.bhi(j)) { b_k += tj[c + P] * k[c + P]; b_w += tj[c + P] * om[c + P]; }
    }
    bk[c] = b_k;
    bw[c] = b_w;
}

int launch_turb_rhs(Ctx *c, double dt) {
    FluxIn f{c->U(), c->V(), c->P(), c->k, c->ap, c->rhie_chow};
    if (c->march_flux) {
        FluxMarch fm = flux_march_geo(c->g);
        flux_march_widen(c, fm);
        MomArgs ma{c->k, c->ti, c->tj, 1.0 / dt, 0.0, 0.0, c->per_i, c->per_j, c->bu, c->bv};
        ma.nts = c->nts();
        ma.nut = c->nut; ma.omega = c->omega;
        const int rn = rect_normals(c);
        if (rn == 2)
            LAUNCH_FLUX(fm, (k_flux_march<4, false, false, 4, 2>), (k_flux_march<4, false, false, 8, 2>), c->g, c->mesh(), f, fm, ma,
                        (double *)nullptr, c->partials);
        else if (rn)
            LAUNCH_FLUX(fm, (k_flux_march<4, false, false, 4, 1>), (k_flux_march<4, false, false, 8, 1>), c->g, c->mesh(), f, fm, ma,
                        (double *)nullptr, c->partials);
        else
            LAUNCH_FLUX(fm, (k_flux_march<4, false>), (k_flux_march<4, false, false, 8>), c->g, c->mesh(), f, fm, ma,
                        (double *)nullptr, c->partials);
        CFD_HIP(hipGetLastError());
        return CFD_OK;
    }
    hipLaunchKernelGGL(k_turb_rhs, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), f, c->omega,
                       c->nut, c->ti, c->tj, 1.0 / dt, c->per_i, c->per_j, c->bu, c->bv);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// pressure-correction right-hand side (solver.py:926-933, 985): b = V * (-(div F)/V)
// ----------------------------------------------------------------------------------
__global__ void k_pressure_rhs(Geo g, MeshDev m, FluxIn f, double *b) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double fw = flux_i(f, m, c, i == 0, false), fe = flux_i(f, m, c + 1, false, i == g.ni - 1);
    const double fs = flux_j(f, m, c, P, g.blo(j), false), fn = flux_j(f, m, c + P, P, false, g.bhi(j));
    const double vol = m.vol[c];
    const double div = (fe - fw + fn - fs) / vol;
    b[c] = vol * (-div);
}

// symmetric elimination of the reference cell (solver.py:1033-1044):
// its row becomes D x = D value, its column moves to the right-hand side.
// pin_coef = couplings of the W/E/S/N neighbours to the reference cell.
__global__ void k_pin_rhs(Geo g, int pi, int pj, double value, const double *pin_coef, const double *D, int per_i,
                          int per_j, double *b) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const size_t c = g.cell(pi, pj);
    b[c] = value * D[c];
    const size_t P = g.pitch;
    if (pi > 0) b[c - 1] += pin_coef[0] * value;
    else if (per_i && g.ni > 1) b[c + (g.ni - 1)] += pin_coef[0] * value;
    if (pi < g.ni - 1) b[c + 1] += pin_coef[1] * value;
    else if (per_i && g.ni > 1) b[c - (g.ni - 1)] += pin_coef[1] * value;
    if (pj > 0) b[c - P] += pin_coef[2] * value;
    else if (per_j && g.nj > 1) b[c + (size_t)(g.nj - 1) * P] += pin_coef[2] * value;
    if (pj < g.nj - 1) b[c + P] += pin_coef[3] * value;
    else if (per_j && g.nj > 1) b[c - (size_t)(g.nj - 1) * P] += pin_coef[3] * value;
}

__global__ void k_plane_sum(Geo g, const double *f, double *partials, int tiles_x, int tiles) {
    double a = 0.0;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        a += f[g.cell(i, j)];
    }
    a = block_sum_256(a);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x] = a;
}
// bb (may be null): (b, b) summed before this row changed; kept right by the difference of the two squares
__global__ void k_ns_rhs(Geo g, int pi, int pj, const double *total, double *b, double *bb) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const size_t c = g.cell(pi, pj);
        const double was = b[c], now = was - total[0];
        b[c] = now;
        if (bb) bb[0] += now * now - was * was;
    }
}

int launch_pressure_rhs(Ctx *c, const double *u, const double *v, const double *p, double *b) {
    FluxIn f{u, v, p, nullptr, c->ap, c->rhie_chow};
    c->bb_ready = false;
    if (c->march_flux) {
        FluxMarch fm = flux_march_geo(c->g);
        flux_march_widen(c, fm);
        MomArgs ma{};
        ma.nts = c->nts();
        c->kc[KC_FLUX_MARCH] += 1;
        if (fm.nstrips > 1) c->kc[KC_FLUX_MARCH_MULTISTRIP] += 1;
        const int rn = rect_normals(c);
        c->probe_begin(13);
        if (rn == 2) LAUNCH_FLUX(fm, (k_flux_march<1, false, false, 4, 2>), (k_flux_march<1, false, false, 8, 2>), c->g, c->mesh(), f, fm, ma, b, c->partials);
        else if (rn) LAUNCH_FLUX(fm, (k_flux_march<1, false, false, 4, 1>), (k_flux_march<1, false, false, 8, 1>), c->g, c->mesh(), f, fm, ma, b, c->partials);
        else LAUNCH_FLUX(fm, (k_flux_march<1, false>), (k_flux_march<1, false, false, 8>), c->g, c->mesh(), f, fm, ma, b, c->partials);
        c->probe_end(13, rn == 2 ? 64.0 : (rn ? 80.0 : 112.0));       // u, v, p, a_P, V, w x 2, n x 4 (not on axis-aligned meshes), S x 2 read, b written
    } else {
        hipLaunchKernelGGL(k_pressure_rhs, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), f, b);
    }
    CFD_HIP(hipGetLastError());
    if (!c->has_dirichlet && c->p_ns_active) {
        // singular (nullspace) form: every row is kept; the reference cell's row -- the one the reference
        // replaces -- absorbs the global imbalance, b_ref -= sum(b), so that the system is compatible
        if (c->march_flux) {
            FluxMarch fm = flux_march_geo(c->g);
            flux_march_widen(c, fm);
            c->n_red_blocks = fm.blocks;
            if (c->comm.nranks == 1 && c->pin_i >= 0) {
                // one rank: the marching pass also summed b^2, which saves the solve its first dot product (scal[83];
                // on slabs the correction of the reference cell's row would have to travel, not worth a collective)
                CFD_TRY(reduce_finish(c, 2, 82));
                hipLaunchKernelGGL(k_ns_rhs, dim3(1), dim3(64), 0, c->stream, c->g, c->pin_i, c->pin_j, c->scal + 82, b, c->scal + 83);
                CFD_HIP(hipGetLastError());
                c->bb_ready = true;
                return CFD_OK;
            }
        } else {
            int tx;
            const int tiles = tiles_of(c->g, tx), nb = red_blocks(tiles);
            hipLaunchKernelGGL(k_plane_sum, dim3(nb), block2d(), 0, c->stream, c->g, b, c->partials, tx, tiles);
            c->n_red_blocks = nb;
        }
        CFD_TRY(reduce_finish(c, 1, 80));               // scal[80] = sum over all slabs
        if (c->pin_i >= 0) {
            hipLaunchKernelGGL(k_ns_rhs, dim3(1), dim3(64), 0, c->stream, c->g, c->pin_i, c->pin_j, c->scal + 80, b, (double *)nullptr);
            CFD_HIP(hipGetLastError());
        }
        return CFD_OK;
    }
    if (!c->has_dirichlet && c->pin_i >= 0) {      // pin_i < 0: the reference cell lives on another slab
        const Level &L = c->levels[0];
        hipLaunchKernelGGL(k_pin_rhs, dim3(1), dim3(64), 0, c->stream, c->g, c->pin_i, c->pin_j, c->pin_value,
                           c->scal + 64, L.D, c->pp_per_i, c->pp_per_j, b);
        CFD_HIP(hipGetLastError());
    }
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// pressure operator planes (solver.py:935-1031): masked face couplings d_f = S^2 / a_f
// and the diagonal with Neumann (0) / Dirichlet (2 d_f) / periodic (d_f) boundary faces.
// ----------------------------------------------------------------------------------
__device__ __forceinline__ double dcoef_i(const MeshDev &m, const double *ap, size_t cR, bool lo, bool hi) {
    const double wr = m.wri[cR], wl = 1.0 - wr;
    double af = lo ? ap[cR] : (hi ? ap[cR - 1] : wl * ap[cR - 1] + wr * ap[cR]);
    af = af > TINY ? af : TINY;
    const double s = m.si[cR];
    return (s * s) / af;
}
__device__ __forceinline__ double dcoef_j(const MeshDev &m, const double *ap, size_t cN, int P, bool lo, bool hi) {
    const double wn = m.wnj[cN], ws = 1.0 - wn;
    double af = lo ? ap[cN] : (hi ? ap[cN - P] : ws * ap[cN - P] + wn * ap[cN]);
    af = af > TINY ? af : TINY;
    const double s = m.sj[cN];
    return (s * s) / af;
}

struct PMasks {
    const int32_t *m[4];
};

__global__ void k_pressure_coeffs(Geo g, MeshDev m, const double *ap, PMasks pm, int per_i, int per_j, double *ci,
                                  double *cj, double *D) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double dw = dcoef_i(m, ap, c, i == 0, false), de = dcoef_i(m, ap, c + 1, false, i == g.ni - 1);
    const double ds = dcoef_j(m, ap, c, P, g.blo(j), false), dn = dcoef_j(m, ap, c + P, P, false, g.bhi(j));
    double aw = dw, ae = de, as = ds, an = dn;
    if (!per_i) {
        if (i == 0) aw = (pm.m[0] && pm.m[0][j]) ? 2.0 * dw : 0.0;
        if (i == g.ni - 1) ae = (pm.m[1] && pm.m[1][j]) ? 2.0 * de : 0.0;
    }
    if (!per_j) {
        if (g.blo(j)) as = (pm.m[2] && pm.m[2][i]) ? 2.0 * ds : 0.0;
        if (g.bhi(j)) an = (pm.m[3] && pm.m[3][i]) ? 2.0 * dn : 0.0;
    }
    D[c] = aw + ae + as + an;
    // face planes keep the boundary-face value too (2 d_f Dirichlet / 0 Neumann): the
    // operator treats x outside the domain as 0 unless the axis is periodic.
    ci[c] = aw;
    if (i == g.ni - 1) ci[c + 1] = ae;
    cj[c] = as;
    if (j == g.nj - 1) cj[c + P] = an;
}

// detach the reference cell: remember the neighbours' couplings to it, then zero them
__global__ void k_pin_operator(Geo g, int pi, int pj, int per_i, int per_j, double *ci, double *cj, double *pin_coef) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const size_t c = g.cell(pi, pj);
    const size_t P = g.pitch;
    const size_t fw = c, fe = c + 1, fs = c, fn = c + P;
    // the periodic seam stores one physical face twice (face 0 and face N)
    const size_t fw2 = c + g.ni, fe2 = c - (g.ni - 1), fs2 = c + (size_t)g.nj * P, fn2 = c - (size_t)(g.nj - 1) * P;
    pin_coef[0] = (pi > 0) ? ci[fw] : (per_i ? ci[fw2] : 0.0);
    pin_coef[1] = (pi < g.ni - 1) ? ci[fe] : (per_i ? ci[fe2] : 0.0);
    pin_coef[2] = (pj > 0) ? cj[fs] : (per_j ? cj[fs2] : 0.0);
    pin_coef[3] = (pj < g.nj - 1) ? cj[fn] : (per_j ? cj[fn2] : 0.0);
    // the cell's own four face values (on a periodic seam they can differ from the neighbour's copy)
    pin_coef[4] = ci[fw]; pin_coef[5] = ci[fe]; pin_coef[6] = cj[fs]; pin_coef[7] = cj[fn];
    ci[fw] = 0.0; ci[fe] = 0.0; cj[fs] = 0.0; cj[fn] = 0.0;
    if (per_i && pi == 0) ci[fw2] = 0.0;
    if (per_i && pi == g.ni - 1) ci[fe2] = 0.0;
    if (per_j && pj == 0) cj[fs2] = 0.0;
    if (per_j && pj == g.nj - 1) cj[fn2] = 0.0;
}

int launch_pressure_coeffs(Ctx *c, Level &L) {
    PMasks pm{{c->pmask[0], c->pmask[1], c->pmask[2], c->pmask[3]}};
    hipLaunchKernelGGL(k_pressure_coeffs, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), c->ap, pm,
                       c->pp_per_i, c->pp_per_j, L.ci, L.cj, L.D);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

int launch_pin_operator(Ctx *c, Level &L) {
    if (!c->has_dirichlet && c->pin_i >= 0) {
        hipLaunchKernelGGL(k_pin_operator, dim3(1), dim3(64), 0, c->stream, c->g, c->pin_i, c->pin_j, c->pp_per_i,
                           c->pp_per_j, L.ci, L.cj, c->scal + 64);
        CFD_HIP(hipGetLastError());
    }
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// p' ghost cells (solver.py:279-319)
// ----------------------------------------------------------------------------------
__global__ void k_pcorr_bc(Geo g, PMasks pm, int per_i, int per_j, double *pc) {
    const int side = blockIdx.y;
    if ((side == CFD_SIDE_JMIN && g.cutlo) || (side == CFD_SIDE_JMAX && g.cuthi)) return;
    const int n = (side < 2) ? g.nj : g.ni;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    size_t cin, cgh, cop;
    if (side == CFD_SIDE_IMIN) {
        cin = g.at(1, t + 1); cgh = g.at(0, t + 1); cop = g.at(g.ni, t + 1);
    } else if (side == CFD_SIDE_IMAX) {
        cin = g.at(g.ni, t + 1); cgh = g.at(g.ni + 1, t + 1); cop = g.at(1, t + 1);
    } else if (side == CFD_SIDE_JMIN) {
        cin = g.at(t + 1, 1); cgh = g.at(t + 1, 0); cop = g.at(t + 1, g.nj);
    } else {
        cin = g.at(t + 1, g.nj); cgh = g.at(t + 1, g.nj + 1); cop = g.at(t + 1, 1);
    }
    const bool per = (side < 2) ? per_i : per_j;
    if (per) {
        pc[cgh] = pc[cop];
    } else {
        const bool dir = pm.m[side] && pm.m[side][t];
        pc[cgh] = dir ? -pc[cin] : pc[cin];
    }
}

int launch_pcorr_bc(Ctx *c) {
    PMasks pm{{c->pmask[0], c->pmask[1], c->pmask[2], c->pmask[3]}};
    const int n = c->g.ni > c->g.nj ? c->g.ni : c->g.nj;
    hipLaunchKernelGGL(k_pcorr_bc, dim3((n + 255) / 256, 4, 1), dim3(256), 0, c->stream, c->g, pm, c->pp_per_i,
                       c->pp_per_j, c->pcorr);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// velocity / pressure correction with under-relaxation (solver.py:1056-1087)
// ----------------------------------------------------------------------------------
__global__ void k_correct(Geo g, MeshDev m, const double *pc, const double *ap, const double *us, const double *vs,
                          const double *u0, const double *v0, const double *p0, double *uo, double *vo, double *po,
                          double ru, double rv, double rp, int nts) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    double gx, gy;
    green_gauss([pc](size_t q) { return pc[q]; }, m, c, g.pitch, gx, gy);
    const double d = m.vol[c] / ap[c];
    const double uc = us[c] - d * gx, vc = vs[c] - d * gy;
    const double a = u0[c], b = v0[c];
    store_stream(uo + c, a + ru * (uc - a), nts);
    store_stream(vo + c, b + rv * (vc - b), nts);
    store_stream(po + c, p0[c] + rp * pc[c], nts);
}

int launch_correct(Ctx *c, const double *us, const double *vs, const double *u0, const double *v0, const double *p0,
                   double *uo, double *vo, double *po, double ru, double rv, double rp) {
    hipLaunchKernelGGL(k_correct, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), c->pcorr, c->ap,
                       us, vs, u0, v0, p0, uo, vo, po, ru, rv, rp, c->nts());
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// continuity residual (solver.py:1205-1215): sqrt(mean(((div F)/V)^2))
// ----------------------------------------------------------------------------------
__global__ void k_continuity(Geo g, MeshDev m, FluxIn f, double *partials, int tiles_x, int tiles) {
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        const int P = g.pitch;
        const size_t c = g.cell(i, j);
        const double fw = flux_i(f, m, c, i == 0, false), fe = flux_i(f, m, c + 1, false, i == g.ni - 1);
        const double fs = flux_j(f, m, c, P, g.blo(j), false), fn = flux_j(f, m, c + P, P, false, g.bhi(j));
        const double div = (fe - fw + fn - fs) / m.vol[c];
        acc += div * div;
    }
    acc = block_sum_256(acc);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[blockIdx.x] = acc;
}

int launch_continuity(Ctx *c, const double *u, const double *v, const double *p, double *out) {
    int tx;
    const int tiles = tiles_of(c->g, tx), nb = red_blocks(tiles);
    FluxIn f{u, v, p, nullptr, c->ap, c->rhie_chow};
    if (c->march_flux) {
        const FluxMarch fm = flux_march_geo(c->g);
        hipLaunchKernelGGL((k_flux_march<0, false>), dim3(fm.blocks), dim3(256), 0, c->stream, c->g, c->mesh(), f, fm, MomArgs{},
                           (double *)nullptr, c->partials);
        CFD_HIP(hipGetLastError());
        return finish_continuity(c, fm.blocks, out);
    }
    hipLaunchKernelGGL(k_continuity, dim3(nb), block2d(), 0, c->stream, c->g, c->mesh(), f, c->partials, tx, tiles);
    CFD_HIP(hipGetLastError());
    c->n_red_blocks = nb;
    CFD_TRY(reduce_finish(c, 1, 0));
    double s;
    CFD_TRY(read_scalars(c, 0, 1, &s));
    *out = sqrt(s / global_cells(c));
    return CFD_OK;
}

// volume-weighted sum: out = {sum f V, sum V}  (post_processing.py:602-610)
__global__ void k_weighted_sum(Geo g, const double *f, const double *w, double *partials, int tiles_x, int tiles) {
    double a = 0.0, b = 0.0;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        const size_t c = g.cell(i, j);
        a += f[c] * w[c];
        b += w[c];
    }
    a = block_sum_256(a);
    b = block_sum_256(b);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        partials[blockIdx.x] = a;
        partials[MAX_PARTIALS + blockIdx.x] = b;
    }
}

int launch_weighted_sum(Ctx *c, const double *f, const double *w, double out[2]) {
    int tx;
    const int tiles = tiles_of(c->g, tx), nb = red_blocks(tiles);
    hipLaunchKernelGGL(k_weighted_sum, dim3(nb), block2d(), 0, c->stream, c->g, f, w, c->partials, tx, tiles);
    CFD_HIP(hipGetLastError());
    c->n_red_blocks = nb;
    CFD_TRY(reduce_finish(c, 2, 0));
    return read_scalars(c, 0, 2, out);
}

// ----------------------------------------------------------------------------------
// wall shear stress along one boundary segment (post_processing.py:646-664): an edge kernel over the faces lo..hi of
// one side; tab holds, per face, t_hat_x, t_hat_y, dn_safe and the face length (O(n) geometry the host derives from
// the mesh once).  tau[q] = rho * nu_eff * (u_t(cell) - u_t(ghost)) / (2 dn_safe); partials: sum tau * len, sum len.
// Slabs: a rank handles the faces it owns (rows [j0, j0 + nj) of an i side; a j side only on the rank that holds it).
// ----------------------------------------------------------------------------------
__global__ void k_wall_friction(Geo g, int side, int lo, int n, int j0, const double *u, const double *v, const double *nut,
                                const double *tab, double nu, double rho, int use_nut, double *tau, double *partials) {
    const int P = g.pitch;
    double a = 0.0, b = 0.0;
    for (int q = blockIdx.x * 256 + threadIdx.y * 64 + threadIdx.x; q < n; q += gridDim.x * 256) {
        const int idx = lo + q;
        int i, j;
        ptrdiff_t dg;
        bool mine = true;
        if (side == CFD_SIDE_IMIN || side == CFD_SIDE_IMAX) {
            j = idx - j0;
            i = side == CFD_SIDE_IMIN ? 0 : g.ni - 1;
            dg = side == CFD_SIDE_IMIN ? -1 : 1;
            mine = j >= 0 && j < g.nj;
        } else {
            i = idx;
            j = side == CFD_SIDE_JMIN ? 0 : g.nj - 1;
            dg = side == CFD_SIDE_JMIN ? -(ptrdiff_t)P : (ptrdiff_t)P;
            mine = side == CFD_SIDE_JMIN ? !g.cutlo : !g.cuthi;
        }
        double t = 0.0;
        if (mine) {
            const size_t c = g.cell(i, j);
            const double tx = tab[q], ty = tab[n + q], dn = tab[2 * n + q], len = tab[3 * n + q];
            const double ut_cell = u[c] * tx + v[c] * ty;
            const double ut_ghost = u[c + dg] * tx + v[c + dg] * ty;
            const double dudn = (ut_cell - ut_ghost) / (2.0 * dn);
            const double nu_eff = use_nut ? nu + nut[c] : nu;
            t = rho * nu_eff * dudn;
            a += t * len;
            b += len;
        }
        tau[q] = t;
    }
    a = block_sum_256(a);
    b = block_sum_256(b);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        partials[blockIdx.x] = a;
        partials[MAX_PARTIALS + blockIdx.x] = b;
    }
}

// tab_dev: 4 n doubles (device), tau_dev: n doubles (device); out = {sum tau len, sum len} over all slabs
int launch_wall_friction(Ctx *c, int side, int lo, int n, const double *tab_dev, double nu, double rho, int use_nut,
                         double *tau_dev, double out[2]) {
    int nb = (n + 255) / 256;
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(k_wall_friction, dim3(nb), block2d(), 0, c->stream, c->g, side, lo, n, c->comm.nranks > 1 ? c->comm.j_offset : 0,
                       c->U(), c->V(), c->nut, tab_dev, nu, rho, use_nut, tau_dev, c->partials);
    CFD_HIP(hipGetLastError());
    c->n_red_blocks = nb;
    CFD_TRY(reduce_finish(c, 2, 0));
    return read_scalars(c, 0, 2, out);
}

// ----------------------------------------------------------------------------------
// Menter SST (turbulence.py:41-208, 343-355, 447-534)
// ----------------------------------------------------------------------------------
constexpr double SST_SIGMA_K1 = 0.85, SST_SIGMA_K2 = 1.0, SST_SIGMA_W1 = 0.5, SST_SIGMA_W2 = 0.856;
constexpr double SST_BETA1 = 0.075, SST_BETA2 = 0.0828, SST_GAMMA1 = 5.0 / 9.0, SST_GAMMA2 = 0.44, SST_BETA_STAR = 0.09;

__device__ __forceinline__ double sst_f2(double k, double om, double d, double nu) {     // turbulence.py:143-155
    const double sk = sqrt(fmax(k, 0.0));
    const double t1 = 2.0 * sk / (SST_BETA_STAR * om * d);
    const double t2 = 500.0 * nu / (d * d * om);
    const double a = fmax(t1, t2);
    return tanh(a * a);
}

// wall distance: min over all wall FACE CENTRES (brute force like turbulence.py:96-136), 1.0 without walls
__global__ void k_wall_distance(Geo g, const double *xc, const double *yc, const double *wall_xy, int n_wall, double *dw) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    __shared__ double sx[256], sy[256];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const bool in = i < g.ni && j < g.nj;
    const size_t c = in ? g.cell(i, j) : 0;
    const double x = in ? xc[c] : 0.0, y = in ? yc[c] : 0.0;
    double d = INFINITY;
    for (int base = 0; base < n_wall; base += 256) {
        __syncthreads();
        if (base + tid < n_wall) { sx[tid] = wall_xy[2 * (base + tid)]; sy[tid] = wall_xy[2 * (base + tid) + 1]; }
        __syncthreads();
        const int n = min(256, n_wall - base);
        for (int q = 0; q < n; ++q) d = fmin(d, hypot(x - sx[q], y - sy[q]));
    }
    if (!in) return;
    if (isinf(d)) d = 1.0;
    dw[c] = fmax(d, 1.0e-12);
}

// ghosts of up to 6 cell planes = nearest interior cell (edges: adjacent cell, corners: the interior
// corner cell; turbulence.py:127-135, 469-473, 500-505, 515-521)
struct Planes6 {
    double *p[6];
    int n;
};
__global__ void k_ghost_extend(Geo g, Planes6 pl) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 2 * (g.ni + 2) + 2 * g.nj;
    if (t >= per) return;
    int ig, jg;
    if (t < g.ni + 2) { ig = t; jg = 0; }
    else if (t < 2 * (g.ni + 2)) { ig = t - (g.ni + 2); jg = g.nj + 1; }
    else if (t < 2 * (g.ni + 2) + g.nj) { ig = 0; jg = t - 2 * (g.ni + 2) + 1; }
    else { ig = g.ni + 1; jg = t - 2 * (g.ni + 2) - g.nj + 1; }
    if ((jg == 0 && g.cutlo) || (jg == g.nj + 1 && g.cuthi)) return;     // a cut: the neighbour's row (halo exchange)
    const int is = min(max(ig, 1), g.ni), js = min(max(jg, 1), g.nj);
    for (int q = 0; q < pl.n; ++q) pl.p[q][g.at(ig, jg)] = pl.p[q][g.at(is, js)];
}
static int launch_ghost_extend(Ctx *c, std::initializer_list<double *> planes) {
    Planes6 pl{};
    for (double *q : planes) pl.p[pl.n++] = q;
    const int per = 2 * (c->g.ni + 2) + 2 * c->g.nj;
    hipLaunchKernelGGL(k_ghost_extend, dim3((per + 255) / 256), dim3(256), 0, c->stream, c->g, pl);
    CFD_HIP(hipGetLastError());
    // j-slabs: the i-edge ghosts of my rows are set, now the cut rows (ghost columns included) from the neighbours
    for (int q = 0; q < pl.n; q += 4)
        CFD_TRY(halo_exchange(c, c->g, 1, pl.p[q], q + 1 < pl.n ? pl.p[q + 1] : nullptr, q + 2 < pl.n ? pl.p[q + 2] : nullptr,
                              q + 3 < pl.n ? pl.p[q + 3] : nullptr));
    return CFD_OK;
}

int launch_wall_distance(Ctx *c) {
    hipLaunchKernelGGL(k_wall_distance, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->m_xc, c->m_yc, c->wall_xy,
                       c->n_wall, c->sst_dw);
    CFD_HIP(hipGetLastError());
    return launch_ghost_extend(c, {c->sst_dw});
}

// _F2 on the whole ghosted planes, corners included (tail of turbulence.apply_bcs, 343-355)
__global__ void k_sst_f2_full(Geo g, const double *k, const double *om, const double *dw, double nu, double *F2) {
    const int ig = blockIdx.x * 64 + threadIdx.x, jg = blockIdx.y * 4 + threadIdx.y;
    if (ig > g.ni + 1 || jg > g.nj + 1) return;
    const size_t c = g.at(ig, jg);
    F2[c] = sst_f2(fmax(k[c], K_FLOOR), fmax(om[c], OMEGA_FLOOR), fmax(dw[c], 1.0e-12), nu);
}
int launch_sst_f2_full(Ctx *c) {
    hipLaunchKernelGGL(k_sst_f2_full, grid2d(c->g.ni + 2, c->g.nj + 2), block2d(), 0, c->stream, c->g, c->k, c->omega, c->sst_dw,
                       c->nu, c->sst_F2);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// Sources of the SST k / omega equations (turbulence.py:447-534) fused with the upwind advection of k and
// omega (solver.py:1109-1121): writes the interior of _S, _F1, _F2, _D_k, _D_omega and
// sk = adv(k) + source_k, sw = adv(omega) + source_omega.
struct SstOut {
    double *S, *F1, *F2, *Dk, *Dw, *sk, *sw;
};
__global__ void k_sst_sources(Geo g, MeshDev m, FluxIn f, const double *om, const double *nut, const double *dwall, const double *xc,
                              const double *yc, double nu, SstOut o) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double fw = flux_i(f, m, c, i == 0, false), fe = flux_i(f, m, c + 1, false, i == g.ni - 1);
    const double fs = flux_j(f, m, c, P, g.blo(j), false), fn = flux_j(f, m, c + P, P, false, g.bhi(j));
    const double vol = m.vol[c];
    const double *kk = f.k;
    const double advk = upwind(kk, c, P, fw, fe, fs, fn, vol);
    const double advw = upwind(om, c, P, fw, fe, fs, fn, vol);
    double ux, uy, vx, vy;
    const double *uu = f.u, *vv = f.v;
    green_gauss([uu](size_t q) { return uu[q]; }, m, c, P, ux, uy);
    green_gauss([vv](size_t q) { return vv[q]; }, m, c, P, vx, vy);
    const double s2 = 2.0 * (ux * ux) + 2.0 * (vy * vy) + (uy + vx) * (uy + vx);
    const double smag = sqrt(fmax(s2, 0.0));
    const double k = fmax(kk[c], K_FLOOR), w = fmax(om[c], OMEGA_FLOOR), nt = nut[c];
    const double d = fmax(dwall[c], 1.0e-12);
    // central differences over the cell-centre spacing (turbulence.py:179-208)
    double dx, dy;
    if (i > 0 && i < g.ni - 1) dx = xc[c + 1] - xc[c - 1];
    else if (i == 0) dx = 2.0 * (xc[c + 1] - xc[c]);
    else dx = 2.0 * (xc[c] - xc[c - 1]);
    if (fabs(dx) < 1.0e-15) dx = 1.0;
    if (!g.blo(j) && !g.bhi(j)) dy = yc[c + P] - yc[c - P];      // at a cut: the neighbour's cell-centre row
    else if (g.blo(j)) dy = 2.0 * (yc[c + P] - yc[c]);
    else dy = 2.0 * (yc[c] - yc[c - P]);
    if (fabs(dy) < 1.0e-15) dy = 1.0;
    const double kx = (kk[c + 1] - kk[c - 1]) / dx, ky = (kk[c + P] - kk[c - P]) / dy;
    const double wx = (om[c + 1] - om[c - 1]) / dx, wy = (om[c + P] - om[c - P]) / dy;
    const double cross = kx * wx + ky * wy;
    const double cd = fmax(2.0 * SST_SIGMA_W2 / w * cross, 1.0e-20);
    // blending functions (turbulence.py:143-172)
    const double sk = sqrt(fmax(k, 0.0));
    const double t1 = sk / (SST_BETA_STAR * w * d);
    const double t2 = 500.0 * nu / (d * d * w);
    const double t3 = 4.0 * SST_SIGMA_W2 * k / (cd * d * d);
    const double a1 = fmin(fmax(t1, t2), t3);
    const double F1 = tanh(pow(a1, 4.0));
    const double F2 = sst_f2(k, w, d, nu);
    const double sig_k = F1 * SST_SIGMA_K1 + (1.0 - F1) * SST_SIGMA_K2;
    const double sig_w = F1 * SST_SIGMA_W1 + (1.0 - F1) * SST_SIGMA_W2;
    const double beta = F1 * SST_BETA1 + (1.0 - F1) * SST_BETA2;
    const double gamma = F1 * SST_GAMMA1 + (1.0 - F1) * SST_GAMMA2;
    o.S[c] = smag; o.F1[c] = F1; o.F2[c] = F2;
    o.Dk[c] = nu + sig_k * nt;
    o.Dw[c] = nu + sig_w * nt;
    const double pk = fmin(nt * s2, 20.0 * SST_BETA_STAR * k * w);
    const double srck = pk - SST_BETA_STAR * k * w;
    const double srcw = gamma * s2 - beta * (w * w) + 2.0 * (1.0 - F1) * SST_SIGMA_W2 / w * cross;
    o.sk[c] = advk + srck;
    o.sw[c] = advw + srcw;
}
int launch_sst_sources(Ctx *c) {
    FluxIn f{c->U(), c->V(), c->P(), c->k, c->ap, c->rhie_chow};
    SstOut o{c->sst_S, c->sst_F1, c->sst_F2, c->sst_Dk, c->sst_Dw, c->kry[13], c->kry[14]};
    hipLaunchKernelGGL(k_sst_sources, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), f, c->omega, c->nut,
                       c->sst_dw, c->m_xc, c->m_yc, c->nu, o);
    CFD_HIP(hipGetLastError());
    return launch_ghost_extend(c, {c->sst_S, c->sst_F1, c->sst_F2, c->sst_Dk, c->sst_Dw});
}

// b = V/dt phi + V src + boundary terms with the scalar's own face conductances (solver.py:723-730)
__global__ void k_sst_rhs(Geo g, const double *vol, const double *phi, const double *src, const double *ci, const double *cj,
                          double inv_dt, int per_i, int per_j, double *b) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    const double v = vol[c];
    double r = v * inv_dt * phi[c] + v * src[c];
    if (!per_i) {
        if (i == 0) r += ci[c] * phi[c - 1];
        if (i == g.ni - 1) r += ci[c + 1] * phi[c + 1];
    }
    if (!per_j) {
        if (g.blo(j)) r += cj[c] * phi[c - P];
        if (g.bhi(j)) r += cj[c + P] * phi[c + P];
    }
    b[c] = r;
}
int launch_sst_rhs(Ctx *c, double dt, const double *phi, const double *src, const double *ci, const double *cj, double *b) {
    hipLaunchKernelGGL(k_sst_rhs, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->m_vol, phi, src, ci, cj, 1.0 / dt,
                       c->per_i, c->per_j, b);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// materialising variants for the fine-grained API (face fluxes, gradient, advection)
// ----------------------------------------------------------------------------------
__global__ void k_face_flux(Geo g, MeshDev m, FluxIn f, double *fi, double *fj) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const int P = g.pitch;
    const size_t c = g.cell(i, j);
    fi[c] = flux_i(f, m, c, i == 0, false);
    if (i == g.ni - 1) fi[c + 1] = flux_i(f, m, c + 1, false, true);
    fj[c] = flux_j(f, m, c, P, g.blo(j), false);
    if (j == g.nj - 1) fj[c + P] = flux_j(f, m, c + P, P, false, !g.cuthi);
}

int launch_face_flux(Ctx *c, const double *u, const double *v, const double *p, int rc, int use_peff, double *fi,
                     double *fj) {
    FluxIn f{u, v, p, use_peff ? c->k : nullptr, c->ap, rc};
    hipLaunchKernelGGL(k_face_flux, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), f, fi, fj);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

__global__ void k_grad(Geo g, MeshDev m, const double *phi, double *gxo, double *gyo) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    double gx, gy;
    green_gauss([phi](size_t q) { return phi[q]; }, m, c, g.pitch, gx, gy);
    gxo[c] = gx;
    gyo[c] = gy;
}

int launch_grad(Ctx *c, const double *phi, double *gx, double *gy) {
    hipLaunchKernelGGL(k_grad, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->mesh(), phi, gx, gy);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

__global__ void k_advect(Geo g, const double *vol, const double *phi, const double *fi, const double *fj, double *out) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    const size_t c = g.cell(i, j);
    out[c] = upwind(phi, c, g.pitch, fi[c], fi[c + 1], fj[c], fj[c + g.pitch], vol[c]);
}

int launch_advect(Ctx *c, const double *phi, const double *fi, const double *fj, double *out) {
    hipLaunchKernelGGL(k_advect, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, c->m_vol, phi, fi, fj, out);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// whole ghosted plane: f = max(f, lo)  (solver.py:1150-1151)
__global__ void k_clip(Geo g, double *f, double lo) {
    const int ig = blockIdx.x * 64 + threadIdx.x, jg = blockIdx.y * 4 + threadIdx.y;
    if (ig > g.ni + 1 || jg > g.nj + 1) return;
    const size_t c = g.at(ig, jg);
    f[c] = fmax(f[c], lo);
}

int launch_clip(Ctx *c, double *f, double lo) {
    hipLaunchKernelGGL(k_clip, grid2d(c->g.ni + 2, c->g.nj + 2), block2d(), 0, c->stream, c->g, f, lo);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

int launch_copy_plane(Ctx *c, const double *src, double *dst) {
    CFD_HIP(hipMemcpyAsync(dst, src, c->g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return CFD_OK;
}

// ----------------------------------------------------------------------------------
// metric planes from node coordinates on the device (mesh.py:186-399).  One thread per
// cell writes its own cell quantities and its west / south face quantities; the last
// column / row also writes the i_max / j_max boundary faces.
// ----------------------------------------------------------------------------------
struct Pt {
    double x, y;
};
struct MeshOut {
    double *vol, *wri, *wnj, *nix, *niy, *njx, *njy, *si, *sj, *di, *dj, *lmin;
};

template <typename NodeFn>
__device__ __forceinline__ void cell_centroid(NodeFn nd, int i, int j, double &cx, double &cy, double &vol) {
    const Pt a = nd(i, j), b = nd(i + 1, j), cc = nd(i + 1, j + 1), d = nd(i, j + 1);
    const double c0 = a.x * b.y - b.x * a.y;
    const double c1 = b.x * cc.y - cc.x * b.y;
    const double c2 = cc.x * d.y - d.x * cc.y;
    const double c3 = d.x * a.y - a.x * d.y;
    const double area = 0.5 * (c0 + c1 + c2 + c3);
    vol = fabs(area);
    cx = ((a.x + b.x) * c0 + (b.x + cc.x) * c1 + (cc.x + d.x) * c2 + (d.x + a.x) * c3) / (6.0 * area);
    cy = ((a.y + b.y) * c0 + (b.y + cc.y) * c1 + (cc.y + d.y) * c2 + (d.y + a.y) * c3) / (6.0 * area);
}

__device__ __forceinline__ double interp_t(double lx, double ly, double rx, double ry, double fx, double fy) {
    const double dx = rx - lx, dy = ry - ly;
    double den = dx * dx + dy * dy;
    den = den > 1.0e-14 ? den : 1.0;
    double t = ((fx - lx) * dx + (fy - ly) * dy) / den;
    t = fmin(fmax(t, 0.0), 1.0);
    return t;
}

template <typename NodeFn>
__device__ void metrics_cell(Geo g, NodeFn nd, MeshOut o, int i, int j) {
    const size_t c = g.cell(i, j);
    const int P = g.pitch;
    double cx, cy, vol;
    cell_centroid(nd, i, j, cx, cy, vol);
    o.vol[c] = vol;
    const Pt a = nd(i, j), b = nd(i + 1, j), cc = nd(i + 1, j + 1), d = nd(i, j + 1);
    const double elx = b.x - a.x, ely = b.y - a.y;       // i-edge, south
    const double ehx = cc.x - d.x, ehy = cc.y - d.y;     // i-edge, north
    const double ewx = d.x - a.x, ewy = d.y - a.y;       // j-edge, west
    const double eex = cc.x - b.x, eey = cc.y - b.y;     // j-edge, east
    const double spi = 0.5 * (hypot(elx, ely) + hypot(ehx, ehy));
    const double spj = 0.5 * (hypot(ewx, ewy) + hypot(eex, eey));
    o.lmin[c] = fmin(spi, spj);
    // ---- west face
    {
        double nx = ewy, ny = -ewx;
        const double ax = 0.5 * (elx + ehx), ay = 0.5 * (ely + ehy);
        if (nx * ax + ny * ay < 0.0) { nx = -nx; ny = -ny; }
        o.nix[c] = nx; o.niy[c] = ny;
        const double s = hypot(nx, ny);
        o.si[c] = s;
        const double sn = s > 1.0e-14 ? s : 1.0;
        const double ux = nx / sn, uy = ny / sn;
        const double fx = 0.5 * (a.x + d.x), fy = 0.5 * (a.y + d.y);
        double dd, wr = 0.5;
        if (i == 0) {
            dd = 2.0 * fabs((cx - fx) * ux + (cy - fy) * uy);
        } else {
            double lx, ly, lv;
            cell_centroid(nd, i - 1, j, lx, ly, lv);
            wr = interp_t(lx, ly, cx, cy, fx, fy);
            dd = fabs((cx - lx) * ux + (cy - ly) * uy);
        }
        o.wri[c] = wr;
        o.di[c] = dd > 1.0e-14 ? dd : 1.0e-14;
    }
    // ---- i_max boundary face
    if (i == g.ni - 1) {
        double nx = eey, ny = -eex;
        const double fx = 0.5 * (b.x + cc.x), fy = 0.5 * (b.y + cc.y);
        if (nx * (fx - cx) + ny * (fy - cy) < 0.0) { nx = -nx; ny = -ny; }
        o.nix[c + 1] = nx; o.niy[c + 1] = ny;
        const double s = hypot(nx, ny);
        o.si[c + 1] = s;
        const double sn = s > 1.0e-14 ? s : 1.0;
        const double dd = 2.0 * fabs((cx - fx) * (nx / sn) + (cy - fy) * (ny / sn));
        o.wri[c + 1] = 0.5;
        o.di[c + 1] = dd > 1.0e-14 ? dd : 1.0e-14;
    }
    // ---- south face
    {
        double nx = -ely, ny = elx;
        const double ax = 0.5 * (ewx + eex), ay = 0.5 * (ewy + eey);
        if (nx * ax + ny * ay < 0.0) { nx = -nx; ny = -ny; }
        o.njx[c] = nx; o.njy[c] = ny;
        const double s = hypot(nx, ny);
        o.sj[c] = s;
        const double sn = s > 1.0e-14 ? s : 1.0;
        const double ux = nx / sn, uy = ny / sn;
        const double fx = 0.5 * (a.x + b.x), fy = 0.5 * (a.y + b.y);
        double dd, wn = 0.5;
        if (g.blo(j)) {
            dd = 2.0 * fabs((cx - fx) * ux + (cy - fy) * uy);
        } else {
            double lx, ly, lv;
            cell_centroid(nd, i, j - 1, lx, ly, lv);     // at a slab cut the node row -1 is the halo row
            wn = interp_t(lx, ly, cx, cy, fx, fy);
            dd = fabs((cx - lx) * ux + (cy - ly) * uy);
        }
        o.wnj[c] = wn;
        o.dj[c] = dd > 1.0e-14 ? dd : 1.0e-14;
    }
    // ---- j_max boundary face
    if (j == g.nj - 1) {
        double nx = -ehy, ny = ehx;
        const double fx = 0.5 * (d.x + cc.x), fy = 0.5 * (d.y + cc.y);
        if (nx * (fx - cx) + ny * (fy - cy) < 0.0) { nx = -nx; ny = -ny; }
        o.njx[c + P] = nx; o.njy[c + P] = ny;
        const double s = hypot(nx, ny);
        o.sj[c + P] = s;
        const double sn = s > 1.0e-14 ? s : 1.0;
        const double dd = 2.0 * fabs((cx - fx) * (nx / sn) + (cy - fy) * (ny / sn));
        o.wnj[c + P] = 0.5;
        o.dj[c + P] = dd > 1.0e-14 ? dd : 1.0e-14;
    }
}

// nodes as two planes [jn][in] with pitch np
__global__ void k_metrics_nodes(Geo g, const double *xn, const double *yn, int np, MeshOut o) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    auto nd = [xn, yn, np](int a, int b) { return Pt{xn[(size_t)b * np + a], yn[(size_t)b * np + a]}; };
    metrics_cell(g, nd, o, i, j);
}
// rectangular mesh: nodes are the tensor product of two 1-D arrays (mesh.py:82-87)
__global__ void k_metrics_rect(Geo g, const double *x1, const double *y1, MeshOut o) {
    const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
    if (i >= g.ni || j >= g.nj) return;
    auto nd = [x1, y1](int a, int b) { return Pt{x1[a], y1[b]}; };
    metrics_cell(g, nd, o, i, j);
}

static MeshOut mesh_out(Ctx *c) {
    return MeshOut{c->m_vol, c->m_wri, c->m_wnj, c->m_nix, c->m_niy, c->m_njx, c->m_njy,
                   c->m_si,  c->m_sj,  c->m_di,  c->m_dj,  c->m_lmin};
}

int launch_metrics_from_nodes(Ctx *c, const double *xn, const double *yn, int node_pitch) {
    hipLaunchKernelGGL(k_metrics_nodes, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, xn, yn, node_pitch,
                       mesh_out(c));
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}
int launch_metrics_from_rect(Ctx *c, const double *x1, const double *y1) {
    hipLaunchKernelGGL(k_metrics_rect, grid2d(c->g.ni, c->g.nj), block2d(), 0, c->stream, c->g, x1, y1, mesh_out(c));
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

}  // namespace cfd
