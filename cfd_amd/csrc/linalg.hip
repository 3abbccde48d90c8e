// Krylov solvers and geometric multigrid on the structured index space (gfx950).
//
// Operator semantics (OpDev): (A x)_P = D x_P - ci[W] x_W - ci[E] x_E - cj[S] x_S - cj[N] x_N
// with x = 0 outside the domain unless the axis is periodic (index wrap).  The west and
// east couplings of a cell are separate plane entries, so the reference's periodic-seam
// asymmetry is represented exactly.
//
//  * momentum / k-omega transport (replaces spilu+BiCGStab / fresh pyamg,
//    solver.py:785-818): right-Jacobi-preconditioned BiCGStab, two right-hand sides per
//    launch (u and v, k and omega share one matrix), Jacobi folded into the operator:
//    (A M^-1 p)_P = p_P - sum c_n p_n / D_n.
//  * pressure correction (replaces the cached pyamg hierarchy, solver.py:821-845):
//    V-cycle-preconditioned CG; BiCGStab with the same V-cycle when the frozen
//    operator is not symmetric.  Cell-centred 2x aggregation, piecewise-constant
//    transfer, rediscretised coarse operators (face conductances in parallel / series),
//    damped-Jacobi smoothing.
//
// All vector kernels are grid-stride over 64x4 tiles with at most 2048 workgroups, dot
// products are wave-shuffle + LDS reductions into per-workgroup partials followed by a
// fixed-order single-workgroup pass (deterministic, no atomics).  Krylov scalars stay on
// the device; the host reads one residual norm per iteration.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "internal.h"

namespace cfd {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int NQ>
__device__ __forceinline__ void block_sums(double (&v)[NQ], double *partials) {
    __shared__ double sm[NQ][4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double w = wsum(v[q]);
        if ((tid & 63) == 0) sm[q][tid >> 6] = w;
    }
    __syncthreads();
    if (tid < NQ) partials[(size_t)tid * MAX_PARTIALS + blockIdx.x] = (sm[tid][0] + sm[tid][1]) + (sm[tid][2] + sm[tid][3]);
}

// Second stage of every sum: one workgroup of 256 threads, nq <= 8 sums of n partials each.  Per thread a strided run,
// then wave shuffles, then the four wave sums -- one barrier in all (the first form ran an eight-step LDS tree with a
// barrier per step for every q: 11 us per call at 8192^2, 34 calls per step, all of them on the critical path).
// Fixed order: the result does not depend on anything but the partials.
__global__ void k_sum_finish(const double *partials, int n, int nq, double *out) {
    __shared__ double sm[8][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = 0; q < nq; ++q) {
        double v = 0.0;
        for (int t = tid; t < n; t += 256) v += partials[(size_t)q * MAX_PARTIALS + t];
        v = wsum(v);
        if (lane == 0) sm[q][wave] = v;
    }
    __syncthreads();
    if (tid < nq) out[tid] = (sm[tid][0] + sm[tid][1]) + (sm[tid][2] + sm[tid][3]);
}

// The same sum inside the kernel that consumes it (option fold_sums, one rank): the first 256 threads of the workgroup
// add the n partials in k_sum_finish's order -- strided run, wave shuffles, the four wave sums: the same bits -- and
// every thread gets the result.  One barrier; every thread of the workgroup must call it.  Saves the dependent
// single-workgroup launch between producer and consumer (two per PCG iteration).
__device__ __forceinline__ double fold_sum(const double *partials, int n, int tid, double *sm4) {
    if (tid < 256) {
        double v = 0.0;
        for (int t = tid; t < n; t += 256) v += partials[t];
        v = wsum(v);
        if ((tid & 63) == 0) sm4[tid >> 6] = v;
    }
    __syncthreads();
    return (sm4[0] + sm4[1]) + (sm4[2] + sm4[3]);
}

// partials -> out[0..nq), then summed over the slabs of a multi-rank run
int sum_finish(Ctx *c, const double *partials, int nblocks, int nq, double *out) {
    if (nq < 1 || nq > 8) { set_error("internal: sum_finish of %d sums", nq); return CFD_E_ARG; }
    hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(256), 0, c->stream, partials, nblocks, nq, out);
    CFD_HIP(hipGetLastError());
    return all_reduce(c, out, nq, 0);
}

int reduce_finish(Ctx *c, int nq, int scal_off) {
    return sum_finish(c, c->partials, c->n_red_blocks, nq, c->scal + scal_off);
}

int read_scalars(Ctx *c, int off, int n, double *host) {
    CFD_HIP(hipMemcpyAsync(c->h_scal, c->scal + off, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CFD_HIP(hipStreamSynchronize(c->stream));
    for (int q = 0; q < n; ++q) host[q] = c->h_scal[q];
    return CFD_OK;
}

int read_scalars_begin(Ctx *c, int off, int n) {
    CFD_HIP(hipMemcpyAsync(c->h_scal + off, c->scal + off, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CFD_HIP(hipEventRecord(c->ev_read, c->stream));
    return CFD_OK;
}
int read_scalars_end(Ctx *c, int off, int n, double *host) {
    count_sync();     // counted like a stream synchronisation (the host blocks), though the GPU need not drain
    CFD_HIP(hipEventSynchronize(c->ev_read));
    for (int q = 0; q < n; ++q) host[q] = c->h_scal[off + q];
    return CFD_OK;
}
int read_scalars_all(Ctx *c) {
    CFD_HIP(hipMemcpyAsync(c->h_spec, c->scal, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CFD_HIP(hipStreamSynchronize(c->stream));
    return CFD_OK;
}
int read_scalars2(Ctx *c, int off1, int n1, double *host1, int off2, int n2, double *host2) {
    CFD_HIP(hipMemcpyAsync(c->h_scal + off1, c->scal + off1, n1 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (n2 > 0) CFD_HIP(hipMemcpyAsync(c->h_scal + off2, c->scal + off2, n2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CFD_HIP(hipStreamSynchronize(c->stream));
    for (int q = 0; q < n1; ++q) host1[q] = c->h_scal[off1 + q];
    for (int q = 0; q < n2; ++q) host2[q] = c->h_scal[off2 + q];
    return CFD_OK;
}

struct Tiling {
    int tx, tiles, blocks;
};
static inline Tiling tiling(const Geo &g) {
    Tiling t;
    t.tx = (g.ni + 63) / 64;
    t.tiles = t.tx * ((g.nj + 3) / 4);
    t.blocks = t.tiles < 2048 ? t.tiles : 2048;
    return t;
}

#define TILE_LOOP(g, T)                                                               \
    for (int tile_ = blockIdx.x; tile_ < (T).tiles; tile_ += gridDim.x)               \
        for (int i = (tile_ % (T).tx) * 64 + threadIdx.x, j = (tile_ / (T).tx) * 4 + threadIdx.y, once_ = 1; \
             once_ && i < (g).ni && j < (g).nj; once_ = 0)

// neighbour sum  sum_n c_n * f(n)  of the 5-point operator at cell (i, j)
template <typename F>
__device__ __forceinline__ double offdiag(const OpDev &op, F xf, size_t c, int i, int j) {
    const Geo &g = op.g;
    const size_t P = g.pitch;
    double s = 0.0;
    if (i > 0) s += op.ci[c] * xf(c - 1);
    else if (op.per_i) s += op.ci[c] * xf(c + (g.ni - 1));
    if (i < g.ni - 1) s += op.ci[c + 1] * xf(c + 1);
    else if (op.per_i) s += op.ci[c + 1] * xf(c - (g.ni - 1));
    // at a slab cut the ghost row holds the neighbour rank's row (halo exchange)
    if (j > 0 || g.cutlo) s += op.cj[c] * xf(c - P);
    else if (op.per_j) s += op.cj[c] * xf(c + (size_t)(g.nj - 1) * P);
    if (j < g.nj - 1 || g.cuthi) s += op.cj[c + P] * xf(c + P);
    else if (op.per_j) s += op.cj[c + P] * xf(c - (size_t)(g.nj - 1) * P);
    return s;
}

// Offset of the row above for the north-face load, zero on the last memory row of a plane.  Opaque to the optimiser:
// written as a plain conditional the compiler turns `cj[c + (last ? 0 : P)]` into `last ? cs : cj[c + P]`, a move of
// the value the c_s load has just been issued for, and the wave then waits for the row it is prefetching.
__device__ __forceinline__ size_t north_offset(bool not_last, size_t P) {
    size_t off = not_last ? P : 0;
    asm volatile("" : "+s"(off));
    return off;
}

template <int NB>
struct Vecs {
    double *p[NB];
};
template <int NB>
struct CVecs {
    const double *p[NB];
};
struct Act {
    int a[2];
};

// ------------------------------------------------------------------ misc vector kernels
__global__ void k_inv_diag(Geo g, Tiling T, const double *D, double *inv) {
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        inv[c] = 1.0 / D[c];
    }
}

// y = A x (plain)
__global__ void k_apply(OpDev op, Tiling T, const double *x, double *y) {
    const Geo g = op.g;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        y[c] = op.D[c] * x[c] - offdiag(op, [x](size_t q) { return x[q]; }, c, i, j);
    }
}

int launch_spmv(Ctx *c, const OpDev &op, const double *x, double *y) {
    const Tiling T = tiling(op.g);
    hipLaunchKernelGGL(k_apply, dim3(T.blocks), block2d(), 0, c->stream, op, T, x, y);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}

// partial[r] = sum (A x_r - b_r)^2
template <int NB>
__global__ void k_resid_norm(OpDev op, Tiling T, CVecs<NB> x, CVecs<NB> b, double *partials) {
    const Geo g = op.g;
    double acc[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) acc[r] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            const double *xr = x.p[r];
            const double res = op.D[c] * xr[c] - offdiag(op, [xr](size_t q) { return xr[q]; }, c, i, j) - b.p[r][c];
            acc[r] += res * res;
        }
    }
    block_sums<NB>(acc, partials);
}

// partial[r] = sum a_r * b_r
template <int NB>
__global__ void k_dot(Geo g, Tiling T, CVecs<NB> a, CVecs<NB> b, double *partials) {
    double acc[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) acc[r] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int r = 0; r < NB; ++r) acc[r] += a.p[r][c] * b.p[r][c];
    }
    block_sums<NB>(acc, partials);
}

// ------------------------------------------------------------------ BiCGStab (device scalars)
// scalar block (doubles): rho[2][2] (generation, rhs), sigma[2], ts[2], tt[2], rr[2]
struct BiScal {
    double rho[2][2], sigma[2], ts[2], tt[2], rr[2];
};
constexpr int BI_OFF = 16;   // offset of BiScal inside Ctx::scal (in doubles)
constexpr int RR_HIST_OFF = 112;     // Ctx::scal slots: |r|^2 of the last PCG iterations (index it & 7)
constexpr int CHEB_SAVE_OFF = 104;   // Ctx::scal slots: the sums of the first Chebyshev launch of a solve (read one launch later)
// Ctx::scal slots of a speculative iteration (internal.h: Ctx::speculate): Chebyshev [0, 6) sums of the first launch
// (|r_0|^2, |r_{K-1}|^2, |b|^2 per right-hand side), then four slots per later launch (|r_k0|^2, |r_{k0+K-1}|^2 of the
// right-hand sides still in it); MG-PCG: per solve of the iteration 8 history slots of |r|^2 and |b|^2
constexpr int SPEC_CHEB_OFF = 128, SPEC_CHEB_SLOTS = 8 + 4 * (Ctx::SPEC_MAX_LAUNCHES - 1);      // 128 .. 179
constexpr int SPEC_PCG_OFF = 192, SPEC_PCG_STRIDE = 10;                                        // 192 .. 231
constexpr double SPEC_MARGIN = 0.5;      // a schedule moves to an earlier iterate only if that one met the test with this factor to spare

__device__ __forceinline__ double safe_div(double a, double b) { return b != 0.0 ? a / b : 0.0; }

// p = r + beta (p - omega v)       (first: p = r)
template <int NB>
__global__ void k_bi_p(Geo g, Tiling T, const BiScal *s, int gen, int first, Act act, CVecs<NB> r, CVecs<NB> v, Vecs<NB> p) {
    double beta[NB], omega[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const double alpha = safe_div(s->rho[gen ^ 1][q], s->sigma[q]);
        omega[q] = safe_div(s->ts[q], s->tt[q]);
        beta[q] = first ? 0.0 : safe_div(s->rho[gen][q], s->rho[gen ^ 1][q]) * safe_div(alpha, omega[q]);
    }
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (!act.a[q]) continue;
            p.p[q][c] = first ? r.p[q][c] : r.p[q][c] + beta[q] * (p.p[q][c] - omega[q] * v.p[q][c]);
        }
    }
}

// y = A (M^-1 x) with M = diag(D) folded in (JAC) or y = A xhat (explicit preconditioned vector);
// MODE 0: partial[q] = sum w_q y_q          (sigma = (rhat, v))
// MODE 1: partial[q] = sum y_q x_q, partial[NB+q] = sum y_q y_q     ((t, s), (t, t))
template <int NB, bool JAC, int MODE>
__global__ void k_bi_apply(OpDev op, Tiling T, const double *inv, Act act, CVecs<NB> x, CVecs<NB> xh, CVecs<NB> w,
                           Vecs<NB> y, double *partials) {
    const Geo g = op.g;
    double acc[2 * NB];
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) acc[q] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (!act.a[q]) continue;
            double yv;
            if (JAC) {
                const double *xr = x.p[q];
                yv = xr[c] - offdiag(op, [xr, inv](size_t n) { return xr[n] * inv[n]; }, c, i, j);
            } else {
                const double *xr = xh.p[q];
                yv = op.D[c] * xr[c] - offdiag(op, [xr](size_t n) { return xr[n]; }, c, i, j);
            }
            y.p[q][c] = yv;
            if (MODE == 0) {
                acc[q] += w.p[q][c] * yv;
            } else {
                acc[q] += yv * x.p[q][c];
                acc[NB + q] += yv * yv;
            }
        }
    }
    block_sums<2 * NB>(acc, partials);
}

// s = r - alpha v   (in place in r)
template <int NB>
__global__ void k_bi_s(Geo g, Tiling T, const BiScal *s, int gen, Act act, CVecs<NB> v, Vecs<NB> r) {
    double alpha[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) alpha[q] = safe_div(s->rho[gen][q], s->sigma[q]);
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (!act.a[q]) continue;
            r.p[q][c] = r.p[q][c] - alpha[q] * v.p[q][c];
        }
    }
}

// x += alpha phat + omega shat ; r = s - omega t ; partial[q] = (rhat, r), partial[NB+q] = (r, r)
template <int NB, bool JAC>
__global__ void k_bi_x(Geo g, Tiling T, const BiScal *s, int gen, const double *inv, Act act, CVecs<NB> p, CVecs<NB> ph,
                       CVecs<NB> sh, CVecs<NB> t, CVecs<NB> rhat, Vecs<NB> r, Vecs<NB> x, double *partials) {
    double alpha[NB], omega[NB], acc[2 * NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        alpha[q] = safe_div(s->rho[gen][q], s->sigma[q]);
        omega[q] = safe_div(s->ts[q], s->tt[q]);
        acc[q] = acc[NB + q] = 0.0;
    }
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            if (!act.a[q]) continue;
            const double sv = r.p[q][c];
            double phat, shat;
            if (JAC) {
                const double iv = inv[c];
                phat = p.p[q][c] * iv;
                shat = sv * iv;
            } else {
                phat = ph.p[q][c];
                shat = sh.p[q][c];
            }
            x.p[q][c] = x.p[q][c] + alpha[q] * phat + omega[q] * shat;
            const double rn = sv - omega[q] * t.p[q][c];
            r.p[q][c] = rn;
            acc[q] += rhat.p[q][c] * rn;
            acc[NB + q] += rn * rn;
        }
    }
    block_sums<2 * NB>(acc, partials);
}

// r = b, rhat = b, x = 0 ; partial[q] = (b, b)
template <int NB>
__global__ void k_bi_init(Geo g, Tiling T, CVecs<NB> b, Vecs<NB> r, Vecs<NB> rhat, Vecs<NB> x, double *partials) {
    double acc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[q] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double bv = b.p[q][c];
            r.p[q][c] = bv;
            rhat.p[q][c] = bv;
            x.p[q][c] = 0.0;
            acc[q] += bv * bv;
        }
    }
    block_sums<NB>(acc, partials);
}

// warm start: x = x0, r = rhat = b - A x0 ; partial[q] = (b, b), partial[NB+q] = (r, r)
template <int NB>
__global__ void k_bi_init_ws(OpDev op, Tiling T, CVecs<NB> b, CVecs<NB> x0, Vecs<NB> r, Vecs<NB> rhat, Vecs<NB> x,
                             double *partials) {
    const Geo g = op.g;
    double acc[2 * NB];
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) acc[q] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double *xs = x0.p[q];
            const double bv = b.p[q][c], xc = xs[c];
            const double rv = bv - (op.D[c] * xc - offdiag(op, [xs](size_t n) { return xs[n]; }, c, i, j));
            r.p[q][c] = rv;
            rhat.p[q][c] = rv;
            if (x.p[q] != xs) x.p[q][c] = xc;
            acc[q] += bv * bv;
            acc[NB + q] += rv * rv;
        }
    }
    block_sums<2 * NB>(acc, partials);
}

static int finish_to(Ctx *c, int blocks, int nq, size_t scal_off_doubles) {
    c->n_red_blocks = blocks;
    return reduce_finish(c, nq, (int)scal_off_doubles);
}

#define OFF_OF(member) ((int)(BI_OFF + offsetof(BiScal, member) / sizeof(double)))

// Jacobi-preconditioned BiCGStab for NB right-hand sides sharing one operator.
template <int NB>
static int bicgstab_jacobi(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2],
                           double tol, int maxiter, int *iters, double relres[2], double rms[2]) {
    const Geo g = op.g;
    const Tiling T = tiling(g);
    const dim3 grid(T.blocks), blk = block2d();
    hipStream_t st = c->stream;
    BiScal *S = reinterpret_cast<BiScal *>(c->scal + BI_OFF);
    double *inv = c->kry[0];
    Vecs<NB> r, rh, p, v, t, xv;
    CVecs<NB> bc;
    for (int q = 0; q < NB; ++q) {
        r.p[q] = c->kry[1 + q];
        rh.p[q] = c->kry[3 + q];
        p.p[q] = c->kry[5 + q];
        v.p[q] = c->kry[7 + q];
        t.p[q] = c->kry[9 + q];
        xv.p[q] = x[q];
        bc.p[q] = b[q];
    }
    auto cv = [](const Vecs<NB> &a) {
        CVecs<NB> o;
        for (int q = 0; q < NB; ++q) o.p[q] = a.p[q];
        return o;
    };
    hipLaunchKernelGGL(k_inv_diag, grid, blk, 0, st, g, T, op.D, inv);
    CFD_TRY(halo_exchange(c, g, 1, inv));
    CFD_HIP(hipMemsetAsync(S, 0, sizeof(BiScal), st));
    double bb[2] = {0, 0}, rr[2] = {0, 0};
    if (x0) {
        // the reference starts from zero (scipy default); starting from the previous field only
        // changes the iteration path, the stopping test |r| <= tol |b| is the reference's
        CVecs<NB> x0c;
        for (int q = 0; q < NB; ++q) x0c.p[q] = x0[q];
        hipLaunchKernelGGL((k_bi_init_ws<NB>), grid, blk, 0, st, op, T, bc, x0c, r, rh, xv, c->partials);
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, 2 * NB, 0));
        double both[4];
        CFD_TRY(read_scalars(c, 0, 2 * NB, both));
        for (int q = 0; q < NB; ++q) { bb[q] = both[q]; rr[q] = both[NB + q]; }
        // rho[0][q] = (rhat, r) = (r, r)
        CFD_HIP(hipMemcpyAsync(c->scal + OFF_OF(rho[0][0]), c->scal + NB, NB * sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
        hipLaunchKernelGGL((k_bi_init<NB>), grid, blk, 0, st, g, T, bc, r, rh, xv, c->partials);
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, NB, OFF_OF(rho[0][0])));
        CFD_TRY(read_scalars(c, OFF_OF(rho[0][0]), NB, bb));
        for (int q = 0; q < NB; ++q) rr[q] = bb[q];
    }
    Act act{{0, 0}};
    int n_active = 0;
    for (int q = 0; q < NB; ++q) {
        if (!(bb[q] == bb[q]) || !(rr[q] == rr[q])) { set_error("momentum-like solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
        act.a[q] = rr[q] > 0.0 && sqrt(rr[q]) > tol * sqrt(bb[q]);
        n_active += act.a[q];
        relres[q] = 0.0;
    }
    int it = 0;
    while (n_active > 0 && it < maxiter) {
        const int gen = it & 1;
        hipLaunchKernelGGL((k_bi_p<NB>), grid, blk, 0, st, g, T, S, gen, it == 0, act, cv(r), cv(v), p);
        CFD_TRY(halo_exchange(c, g, 1, p.p[0], NB == 2 ? p.p[NB - 1] : nullptr));
        if (NB == 2) c->probe_begin(3);
        hipLaunchKernelGGL((k_bi_apply<NB, true, 0>), grid, blk, 0, st, op, T, inv, act, cv(p), cv(p), cv(rh), v,
                           c->partials);
        if (NB == 2) c->probe_end(3);
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, NB, OFF_OF(sigma[0])));
        hipLaunchKernelGGL((k_bi_s<NB>), grid, blk, 0, st, g, T, S, gen, act, cv(v), r);
        CFD_TRY(halo_exchange(c, g, 1, r.p[0], NB == 2 ? r.p[NB - 1] : nullptr));
        hipLaunchKernelGGL((k_bi_apply<NB, true, 1>), grid, blk, 0, st, op, T, inv, act, cv(r), cv(r), cv(r), t,
                           c->partials);
        CFD_HIP(hipGetLastError());
        // partials: [0..NB) = (t,s) -> ts, [NB..2NB) = (t,t) -> tt ; ts and tt are adjacent 2-arrays
        c->n_red_blocks = T.blocks;
        if (NB == 2) {
            CFD_TRY(reduce_finish(c, 4, OFF_OF(ts[0])));
        } else {
            CFD_TRY(reduce_finish(c, 1, OFF_OF(ts[0])));
            CFD_TRY(sum_finish(c, c->partials + MAX_PARTIALS, T.blocks, 1, c->scal + OFF_OF(tt[0])));
        }
        hipLaunchKernelGGL((k_bi_x<NB, true>), grid, blk, 0, st, g, T, S, gen, inv, act, cv(p), cv(p), cv(p), cv(t),
                           cv(rh), r, xv, c->partials);
        CFD_HIP(hipGetLastError());
        // (rhat, r) -> rho[gen^1][q] ; (r, r) -> rr[q]
        c->n_red_blocks = T.blocks;
        CFD_TRY(reduce_finish(c, NB, OFF_OF(rho[gen ^ 1][0])));
        CFD_TRY(sum_finish(c, c->partials + (size_t)NB * MAX_PARTIALS, T.blocks, NB, c->scal + OFF_OF(rr[0])));
        ++it;
        c->kc[KC_BICGSTAB_IT] += 1;
        double got[2];
        CFD_TRY(read_scalars(c, OFF_OF(rr[0]), NB, got));
        for (int q = 0; q < NB; ++q) {
            if (!act.a[q]) continue;
            rr[q] = got[q];
            if (!(rr[q] == rr[q])) { set_error("BiCGStab: non-finite residual at iteration %d", it); return CFD_E_NUMERIC; }
            if (sqrt(rr[q]) <= tol * sqrt(bb[q])) {
                act.a[q] = 0;
                --n_active;
            }
        }
    }
    *iters = it;
    for (int q = 0; q < NB; ++q) relres[q] = bb[q] > 0.0 ? sqrt(rr[q] / bb[q]) : 0.0;
    if (rms) {
        CVecs<NB> xc;
        for (int q = 0; q < NB; ++q) xc.p[q] = x[q];
        CFD_TRY(halo_exchange(c, g, 1, x[0], NB == 2 ? x[NB - 1] : nullptr));
        hipLaunchKernelGGL((k_resid_norm<NB>), grid, blk, 0, st, op, T, xc, bc, c->partials);
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, NB, 0));
        double s2[2];
        CFD_TRY(read_scalars(c, 0, NB, s2));
        for (int q = 0; q < NB; ++q) rms[q] = sqrt(s2[q] / global_cells(c));
    }
    return CFD_OK;
}


// ------------------------------------------------------------------ multigrid kernels
// x = omega b / D   (first Jacobi sweep from a zero guess)
__global__ void k_mg_first(Geo g, Tiling T, const double *D, const double *b, double omega, double *x) {
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        x[c] = omega * b[c] / D[c];
    }
}
// xn = x + omega (b - A x) / D
__global__ void k_mg_jacobi(OpDev op, Tiling T, const double *b, const double *x, double omega, double *xn) {
    const Geo g = op.g;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double d = op.D[c];
        const double ax = d * x[c] - offdiag(op, [x](size_t q) { return x[q]; }, c, i, j);
        xn[c] = x[c] + omega * (b[c] - ax) / d;
    }
}
// coarse b = sum over the aggregate of (b - A x); one thread per coarse cell
__global__ void k_mg_restrict(OpDev op, const double *b, const double *x, Geo gc, Tiling Tc, int fi, int fj, double *bc) {
    const Geo g = op.g;
    TILE_LOOP(gc, Tc) {
        double s = 0.0;
        for (int dj = 0; dj < fj; ++dj) {
            const int jf = j * fj + dj;
            if (jf >= g.nj) break;
            for (int di = 0; di < fi; ++di) {
                const int ifn = i * fi + di;
                if (ifn >= g.ni) break;
                const size_t c = g.cell(ifn, jf);
                s += b[c] - (op.D[c] * x[c] - offdiag(op, [x](size_t q) { return x[q]; }, c, ifn, jf));
            }
        }
        bc[gc.cell(i, j)] = s;
    }
}
// x += P e (piecewise constant)
__global__ void k_mg_prolong(Geo g, Tiling T, Geo gc, int fi, int fj, const double *e, double gamma, double *x) {
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        x[c] += gamma * e[gc.cell(i / fi, j / fj)];
    }
}
// coarse face conductances: parallel sum across the face, series (1/f) along its normal
// A fine level with an odd count ends in an aggregate of ONE cell: the faces between it and a two-cell aggregate join
// centres 1.5 fine spacings apart (the periodic seam included), its boundary face keeps the fine half spacing.
__global__ void k_mg_coarsen(Geo g, const double *ci, const double *cj, const double *D, Geo gc, Tiling Tc, int fi, int fj,
                             int per_i, int per_j, int odd_fix, double *cic, double *cjc, double *ec) {
    TILE_LOOP(gc, Tc) {
        const size_t cc = gc.cell(i, j);
        const size_t P = g.pitch;
        // west face of the aggregate = fine face index i*fi ; i_max face handled by the last column
        double w = 0.0, e_hi = 0.0, s = 0.0, n_hi = 0.0, extra = 0.0;
        const int i0 = i * fi, j0 = j * fj;
        const int i1 = min(i0 + fi, g.ni), j1 = min(j0 + fj, g.nj);
        for (int jf = j0; jf < j1; ++jf) {
            w += ci[g.cell(i0, jf)];
            if (i == gc.ni - 1) e_hi += ci[g.cell(g.ni - 1, jf) + 1];
        }
        for (int ifn = i0; ifn < i1; ++ifn) {
            s += cj[g.cell(ifn, j0)];
            if (j == gc.nj - 1) n_hi += cj[g.cell(ifn, g.nj - 1) + P];
        }
        for (int jf = j0; jf < j1; ++jf)
            for (int ifn = i0; ifn < i1; ++ifn) {
                const size_t c = g.cell(ifn, jf);
                extra += D[c] - (ci[c] + ci[c + 1] + cj[c] + cj[c + P]);
            }
        double dw = fi, de = fi, ds = fj, dn = fj;
        if (odd_fix && fi == 2 && (g.ni & 1) && gc.ni > 1) {
            if (i == gc.ni - 1) dw = 1.5;
            if (i == 0 && per_i) dw = 1.5;
            de = per_i ? 1.5 : 1.0;
        }
        if (odd_fix && fj == 2 && (g.nj & 1) && gc.nj > 1) {
            if (j == gc.nj - 1) ds = 1.5;
            if (j == 0 && per_j) ds = 1.5;
            dn = per_j ? 1.5 : 1.0;
        }
        cic[cc] = w / dw;
        if (i == gc.ni - 1) cic[cc + 1] = e_hi / de;
        cjc[cc] = s / ds;
        if (j == gc.nj - 1) cjc[cc + gc.pitch] = n_hi / dn;
        const int fm = fi > fj ? fi : fj;
        ec[cc] = extra > 0.0 ? extra / fm : 0.0;
    }
}
__global__ void k_mg_coarse_diag(Geo gc, Tiling Tc, const double *cic, const double *cjc, const double *ec, double *Dc) {
    TILE_LOOP(gc, Tc) {
        const size_t cc = gc.cell(i, j);
        Dc[cc] = cic[cc] + cic[cc + 1] + cjc[cc] + cjc[cc + gc.pitch] + ec[cc];
    }
}
// coarsest level: `sweeps` damped-Jacobi sweeps from zero in ONE workgroup (<= 1024 cells)
__global__ void k_mg_coarsest(OpDev op, const double *b, double omega, int sweeps, double *xa, double *xb) {
    const Geo g = op.g;
    const int n = g.ni * g.nj;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const size_t c = g.cell(t % g.ni, t / g.ni);
        const double d = op.D[c];
        xa[c] = d > 0.0 ? omega * b[c] / d : 0.0;
    }
    __syncthreads();
    double *src = xa, *dst = xb;
    for (int s = 1; s < sweeps; ++s) {
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
            const int i = t % g.ni, j = t / g.ni;
            const size_t c = g.cell(i, j);
            const double d = op.D[c];
            const double *xs = src;
            const double ax = d * xs[c] - offdiag(op, [xs](size_t q) { return xs[q]; }, c, i, j);
            dst[c] = d > 0.0 ? xs[c] + omega * (b[c] - ax) / d : 0.0;
        }
        __syncthreads();
        double *tmp = src; src = dst; dst = tmp;
    }
    // result must end in xa
    if (src != xa) {
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
            const size_t c = g.cell(t % g.ni, t / g.ni);
            xa[c] = src[c];
        }
    }
}


// ------------------------------------------------------------------ V-cycle tail in one workgroup
// Levels lt .. nl-1 of the hierarchy (at most TAIL_MAX_CELLS cells on the largest, TAIL_NT on each of the others) are
// cycled by a single workgroup: x and b of every tail level stay in LDS, the phases are separated by workgroup barriers.
// Replaces, per V-cycle, two launches per fused level, eight per unfused level and the coarsest-level kernel -- all of
// them far below the size at which a launch is worth its latency.
//   Where the time of that one workgroup goes decided the layout (tools/tail_clocks.py: 41.7 us per cycle from 64^2
//   before, of which 18 on the 4096-cell level -- arithmetic of ONE compute unit, a double-precision division per cell
//   and sweep -- 7.6 in the 39 barriers of the coarsest level, 5 waiting for coefficient loads level by level):
//   * the couplings of the first two tail levels are loaded once and stay in registers (TailSet<4>, TailSet<1>), those of
//     the smaller levels are fetched once into LDS; nothing is read from global memory after the prologue;
//   * a sweep is x' = (1 - w) x + (w / D)(b + sum c x_nb) with 1 / D formed at load time (the same damped Jacobi step
//     as k_mg_jacobi, in a form without a division; the V-cycle only has to be one fixed symmetric operator, it does not
//     have to round like the multi-launch cycle);
//   * neighbour slots follow from four boundary flags per cell, parent / child slots of the transfers are formed at load
//     time: no integer division inside the cycle;
//   * levels of at most 64 cells are cycled by wave 0 alone, which needs no workgroup barrier (LDS operations of one
//     wave execute in order); the other waves wait at the one barrier that ends that section.
#ifdef TAIL_CLOCKS
// (instrumented build of tools/tail_clocks.py only: 100 MHz timestamps of the phases of one launch)
__device__ unsigned long long g_tail_clk[256];
struct TailMark {
    int n = 0;
    __device__ __forceinline__ void operator()() { if (threadIdx.x == 0 && n < 255) g_tail_clk[n++] = wall_clock64(); }
};
#else
struct TailMark {
    __device__ __forceinline__ void operator()() const {}
};
#endif

template <int S>
struct TailSet {
    // (the four-slot set of the first level keeps no diagonal: only the residual needs it, and takes it from the table)
    double cw[S], ce[S], cs[S], cn[S], dg[S == 1 ? 1 : 0 + 1], inv[S];
    int fl[S];           // 1: i == 0, 2: i == ni - 1, 4: j == 0, 8: j == nj - 1, 16 / 32: the aggregate has a second column / row;
                         // from bit 6: slot of the parent cell on the next level
    int ch[S == 1 ? 1 : 0 + 1];   // slot of the first child on the previous level (never asked of the first level)
    int n, ni, dwi, dsj; // cells, row length, what a neighbour slot moves by across the i / j boundary
    static constexpr int slots = S;
};
struct TailLds {
    double *X, *B, *T, *CF;
    int *CFI;
    int ncache;
};
__device__ __forceinline__ TailLds tail_lds(const TailDev &a, double *lds) {
    TailLds m;
    const int n0 = a.L[0].ni * a.L[0].nj;
    m.X = lds; m.B = lds + a.total; m.T = lds + 2 * a.total;
    m.CF = m.T + n0;
    m.ncache = a.ncache;
    m.CFI = reinterpret_cast<int *>(m.CF + 6 * a.ncache);
    return m;
}
template <bool WAVE>
__device__ __forceinline__ void tail_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
// couplings of cell idx of tail level t as the cycle uses them (boundary couplings without a neighbour: zero), from the planes
struct TailCell {
    double cw, ce, cs, cn, dg, inv;
    int fl, par, ch;
};
__device__ __forceinline__ TailCell tail_cell(const TailArgs &a, int t, int idx) {
    const TailLevel &L = a.L[t];
    const int ni = L.g.ni, nj = L.g.nj;
    const int i = idx % ni, j = idx / ni;
    const size_t c = L.g.cell(i, j), P = L.g.pitch;
    // (every address is inside the plane: faces ni and nj exist, whatever bounds them)
    const double d = L.D[c], w = L.ci[c], e = L.ci[c + 1], s = L.cj[c], nn = L.cj[c + P];
    TailCell r;
    r.fl = (i == 0 ? 1 : 0) | (i == ni - 1 ? 2 : 0) | (j == 0 ? 4 : 0) | (j == nj - 1 ? 8 : 0);
    r.cw = (!(r.fl & 1) || a.per_i) ? w : 0.0;
    r.ce = (!(r.fl & 2) || a.per_i) ? e : 0.0;
    r.cs = (!(r.fl & 4) || a.per_j) ? s : 0.0;
    r.cn = (!(r.fl & 8) || a.per_j) ? nn : 0.0;
    r.dg = d;
    r.inv = d > 0.0 ? 1.0 / d : 0.0;
    r.par = 0;
    r.ch = 0;
    if (t + 1 < a.n) r.par = (j >> (L.fj - 1)) * a.L[t + 1].g.ni + (i >> (L.fi - 1));
    if (t > 0) {
        const TailLevel &F = a.L[t - 1];
        const int i0 = i << (F.fi - 1), j0 = j << (F.fj - 1);
        r.ch = j0 * F.g.ni + i0;
        if (F.fi == 2 && i0 + 1 < F.g.ni) r.fl |= 16;
        if (F.fj == 2 && j0 + 1 < F.g.nj) r.fl |= 32;
    }
    return r;
}
// The couplings of every tail cell as the cycle uses them, packed once per hierarchy (build_pressure_hierarchy) into a
// table of six double and two int arrays over the concatenated levels: the cycle's prologue is then plain coalesced
// loads, without index arithmetic, branches or divisions.
__global__ void k_tail_pack(TailArgs a, double *tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.total) return;
    int t = 0, idx = k;
    while (t < a.n - 1 && idx >= a.L[t].g.ni * a.L[t].g.nj) { idx -= a.L[t].g.ni * a.L[t].g.nj; ++t; }
    const TailCell r = tail_cell(a, t, idx);
    const int nt = a.total;
    tab[k] = r.cw; tab[nt + k] = r.ce; tab[2 * nt + k] = r.cs; tab[3 * nt + k] = r.cn; tab[4 * nt + k] = r.dg; tab[5 * nt + k] = r.inv;
    int *ti = reinterpret_cast<int *>(tab + 6 * (size_t)nt);
    ti[k] = r.fl | (r.par << 6);
    ti[nt + k] = r.ch;
}
template <int S>
__device__ __forceinline__ void tail_set_geo(const TailDev &a, int t, TailSet<S> &s) {
    s.ni = a.L[t].ni;
    s.n = a.L[t].ni * a.L[t].nj;
    s.dwi = a.per_i ? s.ni : 1;
    s.dsj = a.per_j ? s.n : s.ni;
}
// (uniform base + 32-bit byte offset: one address register per slot instead of a 64-bit pointer per load)
template <typename T>
__device__ __forceinline__ T tail_ld(const void *base, unsigned byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
// (slots beyond the level's cells hold the couplings of some cell: every use is guarded by idx < n)
template <int S>
__device__ __forceinline__ void tail_load_table(const TailDev &a, int t, TailSet<S> &s, int tid) {
    tail_set_geo(a, t, s);
    const int nt = a.total;
    const double *t1 = a.tab + nt, *t2 = a.tab + 2 * nt, *t3 = a.tab + 3 * nt, *t4 = a.tab + 4 * nt, *t5 = a.tab + 5 * nt;
    const int *ti = reinterpret_cast<const int *>(a.tab + 6 * (size_t)nt), *tj = ti + nt;
    if (S > 1) s.dg[0] = 0.0, s.ch[0] = 0;
#pragma unroll
    for (int q = 0; q < S; ++q) {
        const int idx = tid + q * TAIL_NT;
        const unsigned k = (unsigned)(a.L[t].off + (idx < s.n ? idx : 0));
        s.cw[q] = tail_ld<double>(a.tab, k * 8u); s.ce[q] = tail_ld<double>(t1, k * 8u); s.cs[q] = tail_ld<double>(t2, k * 8u);
        s.cn[q] = tail_ld<double>(t3, k * 8u); s.inv[q] = tail_ld<double>(t5, k * 8u);
        if (S == 1) s.dg[q] = tail_ld<double>(t4, k * 8u);
        s.fl[q] = tail_ld<int>(ti, k * 4u);
        if (S == 1) s.ch[q] = tail_ld<int>(tj, k * 4u);
    }
}
// levels 2 .. of the tail: their part of the table is copied into LDS, from where tail_load_cache takes a level's
// couplings whenever the cycle comes to it
__device__ __forceinline__ void tail_fill_cache(const TailDev &a, const TailLds &m, int tid) {
    if (tid >= m.ncache) return;
    const int nt = a.total, nc = m.ncache;
    const unsigned k = (unsigned)(a.L[2].off + tid);
    const int *ti = reinterpret_cast<const int *>(a.tab + 6 * (size_t)nt);
#pragma unroll
    for (int c = 0; c < 6; ++c) m.CF[c * nc + tid] = tail_ld<double>(a.tab + c * nt, k * 8u);
    m.CFI[tid] = tail_ld<int>(ti, k * 4u);
    m.CFI[nc + tid] = tail_ld<int>(ti + nt, k * 4u);
}
__device__ __forceinline__ void tail_load_cache(const TailDev &a, const TailLds &m, int t, TailSet<1> &s, int tid) {
    tail_set_geo(a, t, s);
    const int nc = m.ncache, k = a.L[t].coff + (tid < s.n ? tid : 0);
    s.cw[0] = m.CF[k]; s.ce[0] = m.CF[nc + k]; s.cs[0] = m.CF[2 * nc + k]; s.cn[0] = m.CF[3 * nc + k];
    s.dg[0] = m.CF[4 * nc + k]; s.inv[0] = m.CF[5 * nc + k];
    s.fl[0] = m.CFI[k];
    s.ch[0] = m.CFI[nc + k];
}
// (the cycle's arithmetic may contract into fused multiply-adds: nothing here has to round like the reference)
#pragma clang fp contract(fast)
// sum of c x over the four neighbours of slot q
template <typename SET>
__device__ __forceinline__ double tail_nbr(const SET &s, int q, int idx, const double *x) {
    int f = s.fl[q];
    if (SET::slots > 1) asm volatile("" : "+v"(f));      // (as tail_slot: the four boundary offsets of a slot are formed where they are used, not kept)
    const int iw = idx - 1 + ((f & 1) ? s.dwi : 0), ie = idx + 1 - ((f & 2) ? s.dwi : 0);
    const int is = idx - s.ni + ((f & 4) ? s.dsj : 0), in = idx + s.ni - ((f & 8) ? s.dsj : 0);
    double acc = s.cw[q] * x[iw];
    acc += s.ce[q] * x[ie];
    acc += s.cs[q] * x[is];
    acc += s.cn[q] * x[in];
    return acc;
}
// a four-slot set leaves few registers for the sweep itself: one slot's loads and arithmetic at a time (the other three
// waves of the SIMD cover the latency), instead of all twenty loads hoisted to the top and their values spilled
// (the slot's cell index is formed anew in every sweep: hoisted out of the sweep loops, the LDS addresses of the twenty
// neighbours would be registers the coupling sets need)
template <typename SET>
__device__ __forceinline__ int tail_slot(int tid, int q) {
    int idx = tid + q * TAIL_NT;
    if (SET::slots > 1) asm volatile("" : "+v"(idx));      // (a one-slot set has the registers: its addresses may stay)
    return idx;
}
#define TAIL_SLOT_FENCE(SET) do { if (SET::slots > 1) __builtin_amdgcn_sched_barrier(0); } while (0)
template <bool WAVE, typename SET>
__device__ __forceinline__ void tail_first(const SET &s, const double *b, double om, double *x, int tid) {
#pragma unroll
    for (int q = 0; q < SET::slots; ++q) {
        const int idx = tail_slot<SET>(tid, q);
        if (idx < s.n) x[idx] = (om * s.inv[q]) * b[idx];
        TAIL_SLOT_FENCE(SET);
    }
    tail_sync<WAVE>();
}
template <bool WAVE, typename SET>
__device__ __forceinline__ void tail_jacobi(const SET &s, const double *b, const double *x, double om, double *xn, int tid) {
    const double keep = 1.0 - om;
#pragma unroll
    for (int q = 0; q < SET::slots; ++q) {
        const int idx = tail_slot<SET>(tid, q);
        if (idx < s.n) xn[idx] = keep * x[idx] + (om * s.inv[q]) * (b[idx] + tail_nbr(s, q, idx, x));
        TAIL_SLOT_FENCE(SET);
    }
    tail_sync<WAVE>();
}
// where the result of level t ends (x of the level or the scratch plane): the same for every wave, whether or not it took part
__device__ __forceinline__ double *tail_result(const TailDev &a, const TailLds &m, int t) {
    const int swaps = (t == a.n - 1) ? a.coarse_sweeps - 1 : a.post;
    return (swaps & 1) ? m.T : m.X + a.L[t].off;
}
// pre-smoothing from a zero guess, residual into the scratch plane
template <bool WAVE, typename SET>
__device__ __forceinline__ void tail_down(const TailDev &a, const TailLds &m, int t, const SET &s, int tid) {
    double *xt = m.X + a.L[t].off, *bt = m.B + a.L[t].off;
    // the diagonal of a four-slot set: asked for now, there when the residual is formed
    double dgq[SET::slots];
#pragma unroll
    for (int q = 0; q < SET::slots; ++q) {
        if (SET::slots > 1) {
            const int idx = tid + q * TAIL_NT;
            dgq[q] = tail_ld<double>(a.tab + 4 * a.total, (unsigned)(a.L[t].off + (idx < s.n ? idx : 0)) * 8u);
        } else {
            dgq[q] = s.dg[q];
        }
    }
    // the pre-smoothed x must end in xt: start in xt for an odd number of sweeps, in T otherwise
    double *cur = (a.pre & 1) ? xt : m.T, *oth = (a.pre & 1) ? m.T : xt;
    tail_first<WAVE>(s, bt, a.om[0], cur, tid);
    for (int k = 1; k < a.pre; ++k) {
        tail_jacobi<WAVE>(s, bt, cur, a.om[k % a.nom], oth, tid);
        double *tmp = cur; cur = oth; oth = tmp;
    }
#pragma unroll
    for (int q = 0; q < SET::slots; ++q) {
        const int idx = tail_slot<SET>(tid, q);
        if (idx < s.n) m.T[idx] = (bt[idx] - dgq[q] * xt[idx]) + tail_nbr(s, q, idx, xt);
        TAIL_SLOT_FENCE(SET);
    }
    tail_sync<WAVE>();
}
// right-hand side of level t >= 1: the residual of level t - 1 (in the scratch plane) summed over the aggregate
template <bool WAVE>
__device__ __forceinline__ void tail_restrict(const TailDev &a, const TailLds &m, int t, const TailSet<1> &s, int tid) {
    if (tid < s.n) {
        const int nif = a.L[t - 1].ni, c0 = s.ch[0], f = s.fl[0];
        double v = m.T[c0];
        if (f & 16) v += m.T[c0 + 1];
        if (f & 32) {
            v += m.T[c0 + nif];
            if (f & 16) v += m.T[c0 + nif + 1];
        }
        m.B[a.L[t].off + tid] = v;
    }
    // the level's first sweep writes x of the level, or (even number of sweeps) the scratch plane that was just read
    if (!(a.pre & 1) && t < a.n - 1) tail_sync<WAVE>();
}
template <bool WAVE, typename SET>
__device__ __forceinline__ void tail_coarsest(const TailDev &a, const TailLds &m, int t, const SET &s, int tid) {
    double *cur = m.X + a.L[t].off, *oth = m.T, *bt = m.B + a.L[t].off;
    tail_first<WAVE>(s, bt, a.om_coarse, cur, tid);
    for (int k = 1; k < a.coarse_sweeps; ++k) {
        tail_jacobi<WAVE>(s, bt, cur, a.om_coarse, oth, tid);
        double *tmp = cur; cur = oth; oth = tmp;
    }
}
// coarse correction of level t + 1 added, post-smoothing
template <bool WAVE, typename SET>
__device__ __forceinline__ void tail_up(const TailDev &a, const TailLds &m, int t, const SET &s, int tid) {
    const double *e = tail_result(a, m, t + 1);
    double *xt = m.X + a.L[t].off, *bt = m.B + a.L[t].off;
#pragma unroll
    for (int q = 0; q < SET::slots; ++q) {
        const int idx = tail_slot<SET>(tid, q);
        if (idx < s.n) xt[idx] += a.gamma * e[s.fl[q] >> 6];
        TAIL_SLOT_FENCE(SET);
    }
    tail_sync<WAVE>();
    double *cur = xt, *oth = m.T;
    for (int k = 0; k < a.post; ++k) {
        tail_jacobi<WAVE>(s, bt, cur, a.om[k % a.nom], oth, tid);
        double *tmp = cur; cur = oth; oth = tmp;
    }
}
#pragma clang fp contract(off)
// One V-cycle on the right-hand side in B[0 .. n0); A / Bs hold levels 0 / 1 (tail_load_table), the cache the others.
// Returns where the result of level 0 is.  Every thread of the workgroup calls it; the last barrier is behind it.
template <typename SETA, typename MARK>
__device__ __forceinline__ const double *tail_cycle(const TailDev &a, const TailLds &m, const SETA &A, const TailSet<1> &Bs,
                                                    int tid, MARK &mark) {
    const int nt = a.n;
    const bool wave0 = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;      // (uniform: what wave 0 sets up alone stays in scalar registers)
    TailSet<1> C;
    int ts = nt;                 // levels ts .. nt-1 have at most 64 cells: wave 0 cycles them without workgroup barriers
    for (int t = nt - 1; t >= 0 && a.L[t].ni * a.L[t].nj <= 64; --t) ts = t;
    // ---- down, workgroup-wide levels
    const int nfull = ts < nt - 1 ? ts : nt - 1;       // levels 0 .. nfull-1 are smoothed by the whole workgroup on the way down
    for (int t = 0; t < nfull; ++t) {
        if (t == 0) {
            tail_down<false>(a, m, t, A, tid);
        } else if (t == 1) {
            tail_restrict<false>(a, m, t, Bs, tid);
            tail_down<false>(a, m, t, Bs, tid);
        } else {
            tail_load_cache(a, m, t, C, tid);
            tail_restrict<false>(a, m, t, C, tid);
            tail_down<false>(a, m, t, C, tid);
        }
        mark();
    }
    int next_up;
    if (ts < nt) {
        if (wave0) {
            for (int t = ts; t < nt - 1; ++t) {
                if (t == 0) {
                    tail_down<true>(a, m, t, A, tid);
                } else if (t == 1) {
                    tail_restrict<true>(a, m, t, Bs, tid);
                    tail_down<true>(a, m, t, Bs, tid);
                } else {
                    tail_load_cache(a, m, t, C, tid);
                    tail_restrict<true>(a, m, t, C, tid);
                    tail_down<true>(a, m, t, C, tid);
                }
            }
            {
                const int t = nt - 1;
                if (t == 0) {
                    tail_coarsest<true>(a, m, t, A, tid);
                } else if (t == 1) {
                    tail_restrict<true>(a, m, t, Bs, tid);
                    tail_coarsest<true>(a, m, t, Bs, tid);
                } else {
                    tail_load_cache(a, m, t, C, tid);
                    tail_restrict<true>(a, m, t, C, tid);
                    tail_coarsest<true>(a, m, t, C, tid);
                }
            }
            mark();
            for (int t = nt - 2; t >= ts; --t) {
                if (t == 0) {
                    tail_up<true>(a, m, t, A, tid);
                } else if (t == 1) {
                    tail_up<true>(a, m, t, Bs, tid);
                } else {
                    tail_load_cache(a, m, t, C, tid);
                    tail_up<true>(a, m, t, C, tid);
                }
            }
            mark();
        }
        __syncthreads();
        next_up = ts - 1;
    } else {
        // (a coarsest level of more than 64 cells)
        const int t = nt - 1;
        if (t == 0) {
            tail_coarsest<false>(a, m, t, A, tid);
        } else if (t == 1) {
            tail_restrict<false>(a, m, t, Bs, tid);
            tail_coarsest<false>(a, m, t, Bs, tid);
        } else {
            tail_load_cache(a, m, t, C, tid);
            tail_restrict<false>(a, m, t, C, tid);
            tail_coarsest<false>(a, m, t, C, tid);
        }
        next_up = nt - 2;
        mark();
    }
    // ---- up, workgroup-wide levels
    for (int t = next_up; t >= 0; --t) {
        if (t == 0) {
            tail_up<false>(a, m, t, A, tid);
        } else if (t == 1) {
            tail_up<false>(a, m, t, Bs, tid);
        } else {
            tail_load_cache(a, m, t, C, tid);
            tail_up<false>(a, m, t, C, tid);
        }
        mark();
    }
    return tail_result(a, m, 0);
}

__global__ __launch_bounds__(TAIL_NT) void k_mg_tail(TailDev a, const double *b0, double *x0) {
    extern __shared__ double lds[];              // 2 * total + cells of the largest level + the coefficient cache, sized by the launch
    const TailLds m = tail_lds(a, lds);
    const int tid = threadIdx.x;
    TailMark mark;
    mark();
    TailSet<TAIL_CPT> A;
    TailSet<1> Bs{};
    const int n0 = a.L[0].ni * a.L[0].nj;
    // everything the cycle reads from global memory, in flight at once: couplings of every level (plain table loads
    // first, they need no address arithmetic), then the right-hand side
    tail_load_table(a, 0, A, tid);
    if (a.n > 1) tail_load_table(a, 1, Bs, tid);
    double bv[TAIL_CPT];
#pragma unroll
    for (int q = 0; q < TAIL_CPT; ++q) {
        const int idx = tid + q * TAIL_NT, ic = idx < n0 ? idx : 0;
        bv[q] = b0[a.g0.cell(ic % a.L[0].ni, ic / a.L[0].ni)];
    }
    tail_fill_cache(a, m, tid);
#pragma unroll
    for (int q = 0; q < TAIL_CPT; ++q) {
        const int idx = tid + q * TAIL_NT;
        if (idx < n0) m.B[idx] = bv[q];
    }
    __syncthreads();
    mark();
    const double *e = tail_cycle(a, m, A, Bs, tid, mark);
    // ---- result of the first tail level back to its plane
    for (int idx = tid; idx < n0; idx += TAIL_NT) x0[a.g0.cell(idx % a.L[0].ni, idx / a.L[0].ni)] = e[idx];
    mark();
#ifdef TAIL_CLOCKS
    if (tid == 0) g_tail_clk[255] = mark.n;
#endif
}
#ifdef TAIL_CLOCKS
extern "C" int cfd_debug_tail_clocks(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_clk), sizeof(g_tail_clk));
}
#endif

// ------------------------------------------------------------------ a whole MG-PCG solve in one workgroup
// Meshes whose pressure hierarchy fits the LDS of one workgroup together with the Krylov vectors (the reference's own
// cases are 5 x 20 ... 15 x 40 cells; up to about 50 x 50): the solve that pcg_mg() issues as seven launches per iteration --
// the V-cycle tail, three dot products with their finishing kernels, the direction and the residual update -- is ONE launch:
// r lives in the right-hand side slots of tail level 0, p and the diagonal of level 0 in two more LDS planes, x and q = A p
// in registers, the dot products are workgroup sums, and the stopping test |r| <= tol |b| is evaluated on the device after
// every iteration, so the solve takes exactly the iterations the host-checked loop would take without the host looking.
// out[0] = iterations, out[1] = |r|^2, out[2] = |b|^2 (out[8] on the slot layout of the speculative path).
// Same algorithm as pcg_mg (k_cg_p / k_cg_q / k_cg_xr expressions, V-cycle of k_mg_tail); x = 0 initially; the nullspace
// shift of the pinned cell is applied before the store.
struct PcgTailArgs {
    double tol, pin_value;
    int maxiter, ns_active, pin_idx;      // pin_idx: slot of the reference cell on level 0 (-1: none)
    int bb_slot;                          // where |b|^2 goes in out[]
};
__device__ __forceinline__ double tail_block_sum(double v, double *red, int &buf, int tid) {
    const double w = wsum(v);
    double *r = red + 16 * buf;
    if ((tid & 63) == 0) r[tid >> 6] = w;
    __syncthreads();
    double tot = 0.0;
    const int nw = (int)blockDim.x >> 6;          // (a small mesh is launched with as many waves as it has cells for)
    for (int k = 0; k < nw; ++k) tot += r[k];
    buf ^= 1;        // the next sum writes the other half: no barrier between these reads and those writes
    return tot;
}
// CPT: cells per thread on level 0 (1 for meshes of at most TAIL_NT cells: the one-slot set leaves the registers to the cycle)
template <int CPT>
__global__ __launch_bounds__(TAIL_NT) void k_pcg_tail(TailDev a, PcgTailArgs pa, const double *b0, double *x0, double *out) {
    extern __shared__ double lds[];      // the tail's planes, then p, the diagonal of level 0, x and the reduction slots
    const TailLds m = tail_lds(a, lds);
    const int tid = threadIdx.x;
    const int n0 = a.L[0].ni * a.L[0].nj;
    double *Pv = m.CF + 7 * (size_t)m.ncache, *DG = Pv + n0, *XS = DG + n0, *red = XS + n0;
    TailMark mark;
    TailSet<CPT> A;
    TailSet<1> Bs{};
    double bv[CPT], qv[CPT];
    tail_load_table(a, 0, A, tid);
    if (a.n > 1) tail_load_table(a, 1, Bs, tid);
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = tid + q * TAIL_NT, ic = idx < n0 ? idx : 0;
        bv[q] = b0[a.g0.cell(ic % a.L[0].ni, ic / a.L[0].ni)];
    }
    tail_fill_cache(a, m, tid);
    double bsq = 0.0;
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = tid + q * TAIL_NT;
        if (idx < n0) {
            m.B[idx] = bv[q];
            XS[idx] = 0.0;
            DG[idx] = tail_ld<double>(a.tab + 4 * a.total, (unsigned)idx * 8u);
            bsq += bv[q] * bv[q];
        }
    }
    int buf = 0;
    const double bb = tail_block_sum(bsq, red, buf, tid);      // (its barrier also publishes B, DG and the cache)
    double rr = bb, rz_old = 0.0;
    int it = 0;
    if (bb == bb && bb != 0.0) {
        while (it < pa.maxiter) {
            const double *z = tail_cycle(a, m, A, Bs, tid, mark);
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int idx = tid + q * TAIL_NT;
                if (idx < n0) acc += m.B[idx] * z[idx];
            }
            const double rz = tail_block_sum(acc, red, buf, tid);
            const double beta = it == 0 ? 0.0 : safe_div(rz, rz_old);
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int idx = tid + q * TAIL_NT;
                if (idx < n0) Pv[idx] = it == 0 ? z[idx] : z[idx] + beta * Pv[idx];
            }
            __syncthreads();
            acc = 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int idx = tid + q * TAIL_NT;
                qv[q] = 0.0;
                if (idx < n0) {
                    qv[q] = DG[idx] * Pv[idx] - tail_nbr(A, q, idx, Pv);
                    acc += qv[q] * Pv[idx];
                }
            }
            const double pq = tail_block_sum(acc, red, buf, tid);
            const double alpha = safe_div(rz, pq);
            acc = 0.0;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int idx = tid + q * TAIL_NT;
                if (idx < n0) {
                    XS[idx] = XS[idx] + alpha * Pv[idx];
                    const double rn = m.B[idx] - alpha * qv[q];
                    m.B[idx] = rn;
                    acc += rn * rn;
                }
            }
            rr = tail_block_sum(acc, red, buf, tid);            // (its barrier publishes r to the next V-cycle)
            rz_old = rz;
            ++it;
            if (!(rr == rr) || sqrt(rr) <= pa.tol * sqrt(bb)) break;
        }
    }
    // nullspace form: the constant that moves the solution to the reference value at the reference cell
    double shift = 0.0;
    if (pa.ns_active) {
        double *sh = red + 32;
#pragma unroll
        for (int q = 0; q < CPT; ++q)
            if (tid + q * TAIL_NT == pa.pin_idx) sh[0] = pa.pin_value - XS[pa.pin_idx];
        __syncthreads();
        if (pa.pin_idx >= 0) shift = sh[0];
    }
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int idx = tid + q * TAIL_NT;
        if (idx < n0) x0[a.g0.cell(idx % a.L[0].ni, idx / a.L[0].ni)] = XS[idx] + shift;
    }
    if (tid == 0) {
        out[0] = (double)it;
        out[1] = rr;
        out[pa.bb_slot] = bb;
    }
}

// ------------------------------------------------------------------ a whole Chebyshev transport solve in one workgroup
// The momentum / k / omega solve of a mesh of at most CHEB_TAIL_MAX_CELLS cells as ONE launch: what chebyshev_jacobi()
// issues as (row-sum bound + finish, first pass + finish, host read, kpred one-iteration passes, finish, host read) runs
// in one 1024-thread workgroup with both iterates of every right-hand side in LDS and the operator rows in registers.
// Same algorithm and the same expressions as the one-iteration form (k_rowsum_bound, k_cheb, the host's prediction of the
// iteration count from the Chebyshev bound, the stopping test on the iterate the last pass started from): the iterates are
// those of that form bit for bit; only the residual sums are taken in another order.
// out[0] = status (1: solved; 0: not a system for this solver, x untouched -- the caller goes to BiCGStab; 2: bound not met
// within the limit, x = best iterate -- BiCGStab continues from it; 3: non-finite right-hand side), out[1] = iterations,
// out[2 + q] = |r|^2, out[4 + q] = |b|^2, out[6] = rho.
struct ChebTailArgs {
    const double *ci, *cj, *D;
    Geo g;
    int per_i, per_j;
    const double *b[2], *x0[2];
    double *x[2];
    double tol, rho_max;
    int maxiter, cheb_max_iters;
    double *out;
};
template <int NQ>
__device__ __forceinline__ void cheb_tail_sums(double (&v)[NQ], double *red, int &buf, int tid) {
    double *r = red + NQ * 16 * buf;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const double w = wsum(v[q]);
        if ((tid & 63) == 0) r[q * 16 + (tid >> 6)] = w;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double tot = 0.0;
        const int nw = (int)blockDim.x >> 6;
        for (int k = 0; k < nw; ++k) tot += r[q * 16 + k];
        v[q] = tot;
    }
    buf ^= 1;
}
template <int NB, int CPT>
__global__ __launch_bounds__(TAIL_NT) void k_cheb_tail(ChebTailArgs a) {
    extern __shared__ double lds[];          // per right-hand side two iterate planes of n doubles, then the reduction slots
    const int tid = threadIdx.x;
    const Geo g = a.g;
    const int ni = g.ni, nj = g.nj, n = ni * nj;
    // plane w (0: A, 1: B) of right-hand side q (selectors, not pointer arrays: those would live in scratch memory)
    auto plane = [&](int q, int w) { return lds + (size_t)(2 * q + w) * n; };
    double *red = lds + (size_t)2 * NB * n;
    // operator rows and right-hand sides of this thread's cells; couplings without a neighbour are zero (offdiag skips them)
    double d[CPT], cw[CPT], ce[CPT], cs[CPT], cn[CPT], bq[CPT][NB];
    int fl[CPT];
    const int dwi = a.per_i ? ni : 1, dsj = a.per_j ? n : ni;
    double rho_loc = 0.0;
#pragma unroll
    for (int s = 0; s < CPT; ++s) {
        const int idx = tid + s * TAIL_NT, ic = idx < n ? idx : 0;
        const int i = ic % ni, j = ic / ni;
        const size_t c = g.cell(i, j), P = g.pitch;
        fl[s] = (i == 0 ? 1 : 0) | (i == ni - 1 ? 2 : 0) | (j == 0 ? 4 : 0) | (j == nj - 1 ? 8 : 0);
        const double w = a.ci[c], e = a.ci[c + 1], so = a.cj[c], no = a.cj[c + P];
        d[s] = a.D[c];
        cw[s] = (!(fl[s] & 1) || a.per_i) ? w : 0.0;
        ce[s] = (!(fl[s] & 2) || a.per_i) ? e : 0.0;
        cs[s] = (!(fl[s] & 4) || a.per_j) ? so : 0.0;
        cn[s] = (!(fl[s] & 8) || a.per_j) ? no : 0.0;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            bq[s][q] = a.b[q][c];
            if (idx < n) plane(q, 1)[idx] = a.x0[q][c];
        }
        if (idx < n) {
            // k_rowsum_bound: sum of the couplings that have a neighbour, in offdiag's order
            double sum = 0.0;
            if (!(fl[s] & 1) || a.per_i) sum += w * 1.0;
            if (!(fl[s] & 2) || a.per_i) sum += e * 1.0;
            if (!(fl[s] & 4) || a.per_j) sum += so * 1.0;
            if (!(fl[s] & 8) || a.per_j) sum += no * 1.0;
            rho_loc = fmax(rho_loc, sum / d[s]);
        }
    }
    int buf = 0;
    // rho = max over the cells (a maximum: any order gives the same bits)
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rho_loc = fmax(rho_loc, __shfl_down(rho_loc, off, 64));
        double *r = red + 4 * 16 * 2;
        if ((tid & 63) == 0) r[tid >> 6] = rho_loc;
        __syncthreads();              // (also publishes x_0 in the B planes)
        double m = r[0];
        const int nw = (int)blockDim.x >> 6;
        for (int k = 1; k < nw; ++k) m = fmax(m, r[k]);
        rho_loc = m;
    }
    const double rho = rho_loc;
    auto nbr = [&](const double *x, int s, int idx) {
        const int f = fl[s];
        const int iw = idx - 1 + ((f & 1) ? dwi : 0), ie = idx + 1 - ((f & 2) ? dwi : 0);
        const int is = idx - ni + ((f & 4) ? dsj : 0), in = idx + ni - ((f & 8) ? dsj : 0);
        double sum = 0.0;
        sum += cw[s] * x[iw];
        sum += ce[s] * x[ie];
        sum += cs[s] * x[is];
        sum += cn[s] * x[in];
        return sum;
    };
    // one pass of k_cheb: x_k in xk, x_{k-1} in xm, x_{k+1} to xo (xo may be xm: own cell only); acc[q] += |r_k|^2
    auto pass = [&](int wk, int wm, int wo, double c1, double c2, bool first, double (&acc)[2 * NB]) {
#pragma unroll
        for (int s = 0; s < CPT; ++s) {
            const int idx = tid + s * TAIL_NT;
            if (idx < n) {
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const double *xk = plane(q, wk);
                    const double xc = xk[idx];
                    const double r = bq[s][q] - (d[s] * xc - nbr(xk, s, idx));
                    acc[q] += r * r;
                    double step = c2 * (r / d[s]);
                    if (first) acc[NB + q] += bq[s][q] * bq[s][q];
                    else step += c1 * (xc - plane(q, wm)[idx]);
                    plane(q, wo)[idx] = xc + step;
                }
            }
        }
    };
    double acc[2 * NB];
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) acc[q] = 0.0;
    // first pass: r_0, |b|^2, x_1 = x_0 + r_0 / D
    pass(1, 1, 0, 0.0, 1.0, true, acc);
    cheb_tail_sums<2 * NB>(acc, red, buf, tid);          // (its barrier publishes x_1)
    double rr[NB], bb[NB];
    int status = -1, kdone = 0;
    int result = 1;                                      // plane B: x_0
#pragma unroll
    for (int q = 0; q < NB; ++q) { rr[q] = acc[q]; bb[q] = acc[NB + q]; }
    for (int q = 0; q < NB; ++q) {
        if (!(rr[q] == rr[q]) || !(bb[q] == bb[q])) { if (status < 0) status = 3; }
        else if (bb[q] == 0.0 && rr[q] > 0.0) { if (status < 0) status = 0; }
    }
    if (status < 0 && (!(rho >= 0.0) || rho > a.rho_max)) status = 0;
    auto done = [&](const double *r2) {
        for (int q = 0; q < NB; ++q)
            if (sqrt(r2[q]) > a.tol * sqrt(bb[q])) return false;
        return true;
    };
    if (status < 0 && done(rr)) status = 1;              // x_0 already meets the test
    if (status < 0) {
        // iterations the Chebyshev bound asks for: 2 s^k / (1 + s^2k) <= tol |b| / |r_0|
        const double sigma1 = rho > 1e-300 ? 1.0 / rho : 1e300;
        const double sg = rho > 1e-300 ? sigma1 - sqrt(sigma1 * sigma1 - 1.0) : 0.0;
        double need = 1.0;
        for (int q = 0; q < NB; ++q)
            if (rr[q] > 0.0) need = fmin(need, a.tol * sqrt(bb[q]) / sqrt(rr[q]));
        int kpred = 1;
        if (sg > 0.0) {
            const double kp = log(need / 2.0) / log(sg);
            kpred = kp > 1.0 ? (kp < 1.0e6 ? (int)ceil(kp) : 1000000) : 1;
        }
        if (kpred > a.cheb_max_iters || kpred > a.maxiter) status = 0;
        if (status < 0) {
            double rk = rho;
            const int klimit = min(a.maxiter, kpred + 8 + kpred / 2);
            for (int k = 1;; ++k) {
                const int cur = (k & 1) ? 0 : 1, oth = cur ^ 1;       // x_k sits in A for odd k, in B for even k
                const double rnew = 1.0 / (2.0 * sigma1 - rk);
                double c1 = rnew * rk, c2 = 2.0 * rnew / rho;
                if (rho < 1.0e-14) { c1 = 0.0; c2 = 1.0; }
                rk = rnew;
#pragma unroll
                for (int q = 0; q < 2 * NB; ++q) acc[q] = 0.0;
                pass(cur, oth, oth, c1, c2, false, acc);
                if (k >= kpred || k >= klimit) {
                    cheb_tail_sums<2 * NB>(acc, red, buf, tid);       // |r_k|^2 of the iterate the pass started from
                    bool finite = true;
                    for (int q = 0; q < NB; ++q) { rr[q] = acc[q]; finite = finite && rr[q] == rr[q]; }
                    kdone = k;
                    result = cur;
                    if (!finite) { status = 3; break; }
                    if (done(rr)) { status = 1; break; }
                    if (k >= klimit) { status = 2; break; }
                } else {
                    __syncthreads();
                }
            }
        }
    }
    if (status == 1 || status == 2) {
#pragma unroll
        for (int s = 0; s < CPT; ++s) {
            const int idx = tid + s * TAIL_NT;
            if (idx < n) {
#pragma unroll
                for (int q = 0; q < NB; ++q) a.x[q][g.cell(idx % ni, idx / ni)] = plane(q, result)[idx];
            }
        }
    }
    if (tid == 0) {
        a.out[0] = (double)status;
        a.out[1] = (double)kdone;
        for (int q = 0; q < NB; ++q) { a.out[2 + q] = rr[q]; a.out[4 + q] = bb[q]; }
        a.out[6] = rho;
    }
}

// ------------------------------------------------------------------ fused marching multigrid kernels
// One wavefront owns a column strip: lane = column, it marches down the rows of a segment.
// Vertical neighbours stay in registers (sliding windows), horizontal neighbours come from
// lane shuffles, every plane row is loaded exactly once per wave as one coalesced 512-byte
// run.  Two Jacobi sweeps are fused per pass (temporal blocking): strips overlap by HALO = 2
// columns on each side (60 of 64 lanes produce output) and segments by 2 rows.
//   k_mg_down: x = two damped-Jacobi sweeps from a zero guess, then the residual b - A x is
//              summed over 2x2 aggregates straight into the coarse right-hand side.
//   k_mg_up  : x' = two sweeps on (x + P e); at the finest level also (b, x') for PCG.
// Cells outside the domain are neutral (x = b = 0, D = 1, couplings 0); periodic axes wrap.
constexpr int MH = 2;             // horizontal halo = number of fused sweeps
constexpr int MW = 64 - 2 * MH;   // output columns per wave

struct RowC {
    double b, d, inv, cw, ce, cs, cn;
};

__device__ __forceinline__ int wrap_index(int v, int n, int per) {
    if (v >= 0 && v < n) return v;
    if (!per) return -1;
    v %= n;
    return v < 0 ? v + n : v;
}
// plane row (= j + NG) that holds logical row j, or -1 when the row lies outside the domain.
// Rows leave the slab by at most two cells: across a slab cut they are the halo rows, across a
// periodic seam one conditional add / subtract wraps them.
__device__ __forceinline__ int mem_row(int j, const Geo &g, int per) {
    if (j >= 0 && j < g.nj) return j + NG;
    if (j < 0) return g.cutlo ? j + NG : (per ? j + g.nj + NG : -1);
    return g.cuthi ? j + NG : (per ? j - g.nj + NG : -1);
}

// 1/d without the IEEE division sequence: v_rcp_f64 + two Newton steps (the V-cycle is a
// preconditioner; the last bit of the smoother does not matter, its instruction count does)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Unconditional loads from a clamped (always valid) cell: lanes / rows outside the domain
// read finite garbage that only ever multiplies x = 0.
// `up` = distance to the row above (0 for the outermost halo row, whose north face is never used).
__device__ __forceinline__ RowC load_rowc(const OpDev &op, const double *b, size_t c, size_t up) {
    RowC r;
    r.b = b[c]; r.d = op.D[c]; r.cw = op.ci[c]; r.ce = op.ci[c + 1]; r.cs = op.cj[c]; r.cn = op.cj[c + up];
    r.inv = fast_rcp(r.d);
    return r;
}

// Diagonal of the finest pressure level from its face planes: k_pressure_coeffs forms
// D = ((c_W + c_E) + c_S) + c_N from the very values it stores (boundary faces included), so the D
// plane need not be read (8 of the 42 / 50 / 72 B per cell of the finest-level kernels).  The one
// exception is the detached reference cell: its four couplings were zeroed in the planes but still
// count in D of the cell and of its four neighbours; DiagFix carries them (same summation order, so
// the recomputed diagonal has the bits of the stored one).
__device__ __forceinline__ double diag_from_faces(double cw, double ce, double cs, double cn, int col, int row,
                                                  const DiagFix &f, const OpDev &op) {
    // the sum first, the rows next to a detached reference cell again with their fix-ups: written the other way round
    // (fix the faces up, then sum) every row pays four register copies where the two paths join
    double d = ((cw + ce) + cs) + cn;
    if (f.pi >= 0) {
        int di = col - f.pi, dj = row - f.pj;
        if (op.per_i) { if (di == op.g.ni - 1) di = -1; else if (di == 1 - op.g.ni) di = 1; }
        if (op.per_j) { if (dj == op.g.nj - 1) dj = -1; else if (dj == 1 - op.g.nj) dj = 1; }
        if (dj >= -1 && dj <= 1) {           // wave-uniform
            if (dj == 0) {
                if (di == 0) { cw += f.oW; ce += f.oE; cs += f.oS; cn += f.oN; }
                else if (di == -1) ce += f.cW;
                else if (di == 1) cw += f.cE;
            } else if (di == 0) {
                if (dj == -1) cn += f.cS;
                else if (dj == 1) cs += f.cN;
            }
            d = ((cw + ce) + cs) + cn;
        }
    }
    return d;
}

__device__ __forceinline__ double jac(double xc, double xs, double xn, const RowC &r, double omega) {
    const double xl = lane_left(xc), xr = lane_right(xc);
    const double ax = r.d * xc - (r.cw * xl + r.ce * xr + r.cs * xs + r.cn * xn);
    return xc + omega * (r.b - ax) * r.inv;
}
__device__ __forceinline__ double resid(double xc, double xs, double xn, const RowC &r) {
    const double xl = lane_left(xc), xr = lane_right(xc);
    return r.b - (r.d * xc - (r.cw * xl + r.ce * xr + r.cs * xs + r.cn * xn));
}

struct MarchGeo {
    int nstrips, nsegs, Ls;
    int xcd = 0;         // 1: xcd_block() map (the grid is then a multiple of 8)
};
static inline MarchGeo march_geo(const Geo &g) {
    MarchGeo m;
    m.nstrips = (g.ni + MW - 1) / MW;
    int Ls = 32;      // 16 / 32 / 64 measured within 1 % of each other at 8192^2
    while (Ls > 4 && (long)m.nstrips * ((g.nj + Ls - 1) / Ls) < 8192) Ls /= 2;
    m.Ls = Ls;
    m.nsegs = (g.nj + Ls - 1) / Ls;
    return m;
}

template <bool FINE>
__global__ __launch_bounds__(256) void k_mg_down(OpDev op, MarchGeo mg, DiagFix df, const double *b, double omega, double omega2,
                                                 double *x, Geo gc, double *bc) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const int nw = gridDim.x * 4;
    const size_t P = g.pitch;
    for (int item = blockIdx.x * 4 + wave; item < mg.nstrips * mg.nsegs; item += nw) {
        const int strip = item % mg.nstrips, seg = item / mg.nstrips;
        const int iraw = strip * MW - MH + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= MH && lane < 64 - MH && iraw < g.ni;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // software pipeline: the next pair of rows is loaded while the current pair is processed
        struct Raw {
            double b, d, cw, ce, cs, cn;
            int row;
            bool valid;
        };
        auto fetch = [&](int j, Raw &R) {
            const int mr = mem_row(j, g, op.per_j);
            R.valid = cvalid && mr >= 0;
            R.row = (mr >= 0 ? mr : NG) - NG;
            const size_t cc = (size_t)(mr >= 0 ? mr : NG) * P + colo;
            R.b = b[cc]; R.cw = op.ci[cc]; R.ce = op.ci[cc + 1]; R.cs = op.cj[cc];
            R.cn = op.cj[cc + north_offset(mr < g.nj + 2 * NG - 1, P)];
            // the outermost halo row of a high slab cut has no north face in memory: its D is read
            R.d = (df.on && mr != g.nj + 2 * NG - 1) ? 0.0 : op.D[cc];
            if (!df.on || mr == g.nj + 2 * NG - 1) R.row = -1000000;
        };
        auto mk = [&](const Raw &R) -> RowC {
            const double d = R.row > -1000000 ? diag_from_faces(R.cw, R.ce, R.cs, R.cn, (int)colo - CO, R.row, df, op) : R.d;
            return RowC{R.b, d, fast_rcp(d), R.cw, R.ce, R.cs, R.cn};
        };
        bool v2, v1;
        Raw pre;
        fetch(j0 - 2, pre);
        RowC rc2 = mk(pre);
        v2 = pre.valid;
        fetch(j0 - 1, pre);
        RowC rc1 = mk(pre);
        v1 = pre.valid;
        double x1a = v2 ? omega * rc2.b * rc2.inv : 0.0, x1b = v1 ? omega * rc1.b * rc1.inv : 0.0, x1c;
        double x2a = 0.0, x2b = 0.0, x2c = 0.0, rsum = 0.0;
        Raw nx[2];
        fetch(j0, nx[0]);
        fetch(j0 + 1, nx[1]);
        for (int jr = j0; jr < j1 + 2; jr += 2) {
            RowC rn[2];
            bool vn[2];
            const Raw cu[2] = {nx[0], nx[1]};
            if (jr + 2 < j1 + 2) {
                fetch(jr + 2, nx[0]);
                fetch(jr + 3, nx[1]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rn[q] = mk(cu[q]);
                vn[q] = cu[q].valid;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = jr + q;                 // newest row
                const RowC rc0 = rn[q];
                x1c = vn[q] ? omega * rc0.b * rc0.inv : 0.0;
                // second sweep on row j-1
                x2c = jac(x1b, x1a, x1c, rc1, omega2);
                if (!v1) x2c = 0.0;                   // outside the domain: x = 0
                if (j - 1 >= j0 && j - 1 < j1 && out) x[(size_t)(j - 1 + NG) * P + colo] = x2c;
                // residual on row j-2
                const int jo = j - 2;
                const double res = resid(x2b, x2a, x2c, rc2);
                if (jo >= j0 && jo < j1) {
                    rsum += res;
                    if (((jo - j0) & 1) || jo == j1 - 1) {
                        const double tot = rsum + lane_right(rsum);
                        if (out && !((lane - MH) & 1)) bc[gc.cell(iraw >> 1, jo >> 1)] = (iraw + 1 < g.ni) ? tot : rsum;
                        rsum = 0.0;
                    }
                }
                x1a = x1b; x1b = x1c;
                x2a = x2b; x2b = x2c;
                rc2 = rc1; rc1 = rc0;
                v2 = v1; v1 = vn[q];
            }
        }
    }
}

template <bool FINE>
__global__ __launch_bounds__(256) void k_mg_up(OpDev op, MarchGeo mg, DiagFix df, const double *b, const double *x, Geo gc,
                                               const double *e, double gamma, double omega, double omega2, double *xn,
                                               double *partials) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const int nw = gridDim.x * 4;
    const size_t P = g.pitch;
    double dot = 0.0;
    for (int item = blockIdx.x * 4 + wave; item < mg.nstrips * mg.nsegs; item += nw) {
        const int strip = item % mg.nstrips, seg = item / mg.nstrips;
        const int iraw = strip * MW - MH + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const size_t coloc = (size_t)((cvalid ? col : 0) >> 1) + CO;
        const bool out = lane >= MH && lane < 64 - MH && iraw < g.ni;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // software pipeline: the next pair of rows is loaded while the current pair is processed
        struct Raw {
            double b, d, cw, ce, cs, cn, x, e;
            int row;
            bool valid;
        };
        auto fetch = [&](int j, Raw &R) {
            const int mr = mem_row(j, g, op.per_j);
            R.valid = cvalid && mr >= 0;
            const int js = (mr >= 0 ? mr : NG) - NG;     // -2, -1 / nj, nj+1 are halo rows at a slab cut
            const size_t cc = (size_t)(js + NG) * P + colo;
            R.x = x[cc];
            R.e = e[(size_t)((js >> 1) + NG) * gc.pitch + coloc];
            R.b = b[cc]; R.cw = op.ci[cc]; R.ce = op.ci[cc + 1]; R.cs = op.cj[cc];
            R.cn = op.cj[cc + north_offset(mr < g.nj + 2 * NG - 1, P)];
            R.d = (df.on && mr != g.nj + 2 * NG - 1) ? 0.0 : op.D[cc];
            R.row = (!df.on || mr == g.nj + 2 * NG - 1) ? -1000000 : js;
        };
        auto mk = [&](const Raw &R) -> RowC {
            const double d = R.row > -1000000 ? diag_from_faces(R.cw, R.ce, R.cs, R.cn, (int)colo - CO, R.row, df, op) : R.d;
            return RowC{R.b, d, fast_rcp(d), R.cw, R.ce, R.cs, R.cn};
        };
        bool v2, v1;
        double x0a, x0b, x0c;
        Raw pre;
        fetch(j0 - 2, pre);
        RowC rc2 = mk(pre);
        v2 = pre.valid;
        (void)v2;
        x0a = pre.valid ? pre.x + gamma * pre.e : 0.0;
        fetch(j0 - 1, pre);
        RowC rc1 = mk(pre);
        v1 = pre.valid;
        x0b = pre.valid ? pre.x + gamma * pre.e : 0.0;
        double x1a = 0.0, x1b = 0.0, x1c = 0.0;
        Raw nx[2];
        fetch(j0, nx[0]);
        fetch(j0 + 1, nx[1]);
        for (int jr = j0; jr < j1 + 2; jr += 2) {
            RowC rn[2];
            double xl[2];
            bool vn[2];
            const Raw cu[2] = {nx[0], nx[1]};
            if (jr + 2 < j1 + 2) {
                fetch(jr + 2, nx[0]);
                fetch(jr + 3, nx[1]);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rn[q] = mk(cu[q]);
                vn[q] = cu[q].valid;
                xl[q] = cu[q].valid ? cu[q].x + gamma * cu[q].e : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = jr + q;
                const RowC rc0 = rn[q];
                x0c = xl[q];
                // first sweep on row j-1
                x1c = jac(x0b, x0a, x0c, rc1, omega);
                if (!v1) x1c = 0.0;
                // second sweep on row j-2
                const int jo = j - 2;
                const double x2 = jac(x1b, x1a, x1c, rc2, omega2);
                if (jo >= j0 && jo < j1 && out) {
                    xn[(size_t)(jo + NG) * P + colo] = x2;
                    if (FINE) dot += rc2.b * x2;
                }
                x0a = x0b; x0b = x0c;
                x1a = x1b; x1b = x1c;
                rc2 = rc1; rc1 = rc0;
                v2 = v1; v1 = vn[q];
            }
        }
    }
    if (FINE) {
        __shared__ double sm[4];
        const double w = wsum(dot);
        if (lane == 0) sm[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

// ---- K fused sweeps per pass (K = 3 on single-rank contexts: V(3,3) needs 5 PCG iterations at 8192^2
// where V(2,2) needs 7, for the bytes of the two-sweep kernels).  Same structure as above with K halo
// lanes / rows per side; stage s of row j runs while row j + s is being loaded.
template <int K>
struct SweepOm {
    double w[K];
};

// what a row fetch brings in, as loaded: conversions and arithmetic wait until the row is consumed, one trip later
// (anything computed from a value in the fetch itself makes the wave wait for the loads it has just issued)
struct MgRaw {
    double b, d, cw, ce, cs, cn, x, e, q;
    float fw, fe, fs, fn, fx;          // single-precision faces / iterate (F32 / XF forms)
    float fb, fc;                      // single-precision right-hand side / coarse correction (VT forms)
    int row;
    bool valid;
};

// The PCG residual update r = r_old - alpha q folded into the finest-level pre-smoother (k_mg_down_k<true, K, true>):
// the kernel reads r_old and q where it would read its right-hand side, forms r in registers and stores the rows it
// owns to `rnew` (a different plane than r_old: the halo rows of a segment are the neighbouring segments' own rows).
struct ResidFuse {
    const double *q;
    const double *rz, *pq;       // alpha = rz / pq (CgScal of the running solve)
    double *rnew;
    double *rr_partials = nullptr;     // non-null: |r|^2 of the rows a block owns, one partial sum per block
    int rr_slot = 0;                   // ... and the Ctx::scal slot their sum goes to (host side)
    const double *pq_part = nullptr;   // fold_sums: (p, q) still waits as pq_n partial sums; the kernel adds them up itself
    int pq_n = 0;                      // (fold_sum) and its first workgroup stores the sum to pq_out
    double *pq_out = nullptr;
};

// VT (vector types of the first coarse levels, option mg_c32), two bits: 1 the NEXT level keeps its vectors in single
// precision (the coarse right-hand side goes out, the coarse correction comes in as float); 2 THIS level does (b and x
// float).  0 everywhere: every vector plane double; finest level over a float level: 1; a float level over a double one:
// 2; over another float level: 3.  Like the faces
// and the finest level's x and z this is storage of the preconditioner only: 4096^2 cells of an 8192^2 mesh move 38 B
// per cell and V-cycle instead of 74.  Float planes use the first half of the double plane (same element index).
template <bool UP, bool RF = false, bool XO = false, bool F32 = false, bool XF = false, int VT = 0>
__device__ __forceinline__ void mg_fetch(const OpDev &op, const DiagFix &df, const Geo &g, const Geo &gc, const double *b,
                                         const double *x, const double *e, int j, bool cvalid, size_t colo, size_t coloc,
                                         MgRaw &R, const double *q = nullptr) {
    const size_t P = g.pitch;
    const int mr = mem_row(j, g, op.per_j);
    R.valid = cvalid && mr >= 0;
    const int js = (mr >= 0 ? mr : NG) - NG;
    const size_t cc = (size_t)(js + NG) * P + colo;
    if (UP) {
        if (XF) R.fx = reinterpret_cast<const float *>(x)[cc];
        else R.x = x[cc];
        if (VT & 1) R.fc = reinterpret_cast<const float *>(e)[(size_t)((js >> 1) + NG) * gc.pitch + coloc];
        else R.e = e[(size_t)((js >> 1) + NG) * gc.pitch + coloc];
    }
    if (XO) R.x = x[cc];
    if (VT & 2) R.fb = reinterpret_cast<const float *>(b)[cc];
    else R.b = b[cc];
    if (RF) R.q = q[cc];
    const size_t cn_at = cc + north_offset(mr < g.nj + 2 * NG - 1, P);
    if (F32) {
        R.fw = op.cif[cc]; R.fe = op.cif[cc + 1]; R.fs = op.cjf[cc]; R.fn = op.cjf[cn_at];
    } else {
        R.cw = op.ci[cc]; R.ce = op.ci[cc + 1]; R.cs = op.cj[cc]; R.cn = op.cj[cn_at];
    }
    R.d = (df.on && mr != g.nj + 2 * NG - 1) ? 0.0 : op.D[cc];
    R.row = (!df.on || mr == g.nj + 2 * NG - 1) ? -1000000 : js;
}
template <bool F32 = false, bool BF = false>
__device__ __forceinline__ RowC mg_row(const MgRaw &R, const DiagFix &df, const OpDev &op, size_t colo) {
    const double cw = F32 ? (double)R.fw : R.cw, ce = F32 ? (double)R.fe : R.ce, cs = F32 ? (double)R.fs : R.cs, cn = F32 ? (double)R.fn : R.cn;
    const double d = R.row > -1000000 ? diag_from_faces(cw, ce, cs, cn, (int)colo - CO, R.row, df, op) : R.d;
    return RowC{BF ? (double)R.fb : R.b, d, fast_rcp(d), cw, ce, cs, cn};
}

template <bool FINE, int K, bool RF = false, int WPB = 4, bool F32 = false, int VT = 0>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mg_down_k(OpDev op, MarchGeo mg, DiagFix df, const double *b, SweepOm<K> om, double *x,
                                                        Geo gc, double *bc, ResidFuse rf = ResidFuse{}, int nts = 0) {
    const Geo g = op.g;
    double alpha = 0.0;
    if (RF) {
        double pq;
        if (rf.pq_n > 0) {
            __shared__ double fsm[4];
            pq = fold_sum(rf.pq_part, rf.pq_n, threadIdx.x, fsm);
            if (blockIdx.x == 0 && threadIdx.x == 0) rf.pq_out[0] = pq;
        } else {
            pq = rf.pq[0];
        }
        alpha = safe_div(rf.rz[0], pq);
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const size_t P = g.pitch;
    constexpr int W = 64 - 2 * K;
    double rr2 = 0.0;
    const MarchItems<WPB> items(mg.nstrips, mg.nsegs, mg.xcd, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, mg.nstrips, wave, strip, seg, active);
        const int iraw = strip * W - K + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= K && lane < 64 - K && iraw < g.ni && active;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // RC[s], V[s]: operator row / validity of row j - s;  X[s][0..2]: sweep s on rows (r - 2, r - 1, r), r = j - (s - 1)
        RowC RC[K + 1];
        bool V[K + 1];
        double X[K + 1][3];
#pragma unroll
        for (int s2 = 0; s2 <= K; ++s2) {
            RC[s2] = RowC{0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0};
            V[s2] = false;
            X[s2][0] = 0.0; X[s2][1] = 0.0; X[s2][2] = 0.0;
        }
        double rsum = 0.0;
        // two rows of loads in flight per wave (see k_mg_up_k)
        const int jend = j1 + K;
        auto fetch = [&](int j, MgRaw &R) {
            mg_fetch<false, RF, false, F32, false, VT>(op, df, g, gc, b, nullptr, nullptr, j < jend ? j : jend - 1, cvalid, colo, 0, R, rf.q);
        };
        auto trip = [&](int j, MgRaw cur) {
            if (RF) {
                cur.b = cur.b - alpha * cur.q;                               // k_cg_r's expression
                // own rows, and at a slab cut the ghost rows next to the first / last segment (formed from the neighbour's
                // r_old and q rows, they are the neighbour's r): the way up reads them
                const bool own = (j >= j0 && j < j1) || (g.cutlo && j < 0 && j0 == 0) || (g.cuthi && j >= g.nj && j1 == g.nj);
                if (own && out) store_stream(rf.rnew + (size_t)(j + NG) * P + colo, cur.b, nts & 1);
                if (j >= j0 && j < j1 && out) rr2 += cur.b * cur.b;            // (r, r) of this slab's own cells
            }
            RC[0] = mg_row<F32, (VT & 2) != 0>(cur, df, op, colo);
            V[0] = cur.valid;
            X[1][2] = V[0] ? om.w[0] * RC[0].b * RC[0].inv : 0.0;          // first sweep from a zero guess
#pragma unroll
            for (int s2 = 2; s2 <= K; ++s2) {                               // sweep s2 on row j - (s2 - 1)
                double xn = jac(X[s2 - 1][1], X[s2 - 1][0], X[s2 - 1][2], RC[s2 - 1], om.w[s2 - 1]);
                if (!V[s2 - 1]) xn = 0.0;
                X[s2][2] = xn;
                if (s2 == K) {
                    const int row = j - (K - 1);
                    if (row >= j0 && row < j1 && out) {
                        // finest level with single-precision faces: the pre-smoothed iterate, which only the post-smoothing
                        // pass reads, is stored in single precision too (same pitch in elements, first half of the plane)
                        if ((F32 && FINE) || (VT & 2)) store_stream_f32(reinterpret_cast<float *>(x) + (size_t)(row + NG) * P + colo, (float)xn, nts & 1);
                        else store_stream(x + (size_t)(row + NG) * P + colo, (nts & 2) ? (double)(float)xn : xn, nts & 1);
                    }
                }
            }
            // residual on row j - K, summed over 2 x 2 aggregates into the coarse right-hand side
            const int jo = j - K;
            const double res = resid(X[K][1], X[K][0], X[K][2], RC[K]);
            if (jo >= j0 && jo < j1) {
                rsum += res;
                if (((jo - j0) & 1) || jo == j1 - 1) {
                    const double tot = rsum + lane_right(rsum);
                    if (out && !((lane - K) & 1)) {
                        const double cv = (iraw + 1 < g.ni) ? tot : rsum;
                        if (VT & 1) reinterpret_cast<float *>(bc)[gc.cell(iraw >> 1, jo >> 1)] = (float)cv;
                        else bc[gc.cell(iraw >> 1, jo >> 1)] = (nts & 2) ? (double)(float)cv : cv;
                    }
                    rsum = 0.0;
                }
            }
#pragma unroll
            for (int s2 = K; s2 >= 1; --s2) { RC[s2] = RC[s2 - 1]; V[s2] = V[s2 - 1]; }
#pragma unroll
            for (int s2 = 1; s2 <= K; ++s2) { X[s2][0] = X[s2][1]; X[s2][1] = X[s2][2]; }
        };
        MgRaw r0, r1;
        fetch(j0 - K, r0);
        fetch(j0 - K + 1, r1);
        for (int j = j0 - K; j < jend; j += 2) {
            {
                const MgRaw cur = r0;
                fetch(j + 2, r0);
                trip(j, cur);
            }
            if (j + 1 < jend) {
                const MgRaw cur = r1;
                fetch(j + 3, r1);
                trip(j + 1, cur);
            }
            if (WPB > 4) __syncthreads();       // lockstep: same segment, same trip count for every wave of the workgroup
        }
    }
    if (RF) {
        if (rf.rr_partials) {            // kernel argument: every wave of the grid takes the same way
            __shared__ double sm[WPB];
            const double w = wsum(rr2);
            if (lane == 0) sm[wave] = w;
            __syncthreads();
            if (threadIdx.x == 0) rf.rr_partials[blockIdx.x] = sum_waves<WPB>(sm);
        }
    }
}

// ZF (finest level, PCG): the result z goes out in single precision -- k_cg_pq reads it as such -- and (r, z) is taken
// with the rounded value, the z the direction step will see.
template <bool FINE, int K, int WPB = 4, bool F32 = false, bool ZF = false, int VT = 0>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_mg_up_k(OpDev op, MarchGeo mg, DiagFix df, const double *b, const double *x, Geo gc,
                                                      const double *e, double gamma, SweepOm<K> om, double *xn, double *partials, int nts = 0) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const size_t P = g.pitch;
    constexpr int W = 64 - 2 * K;
    double dot = 0.0;
    const MarchItems<WPB> items(mg.nstrips, mg.nsegs, mg.xcd, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, mg.nstrips, wave, strip, seg, active);
        const int iraw = strip * W - K + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const size_t coloc = (size_t)((cvalid ? col : 0) >> 1) + CO;
        const bool out = lane >= K && lane < 64 - K && iraw < g.ni && active;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // RC[s], V[s]: row j - s;  X[s][0..2]: iterate after s sweeps on rows (r - 2, r - 1, r), r = j - s
        RowC RC[K + 1];
        bool V[K + 1];
        double X[K + 1][3];
#pragma unroll
        for (int s2 = 0; s2 <= K; ++s2) {
            RC[s2] = RowC{0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0};
            V[s2] = false;
            X[s2][0] = 0.0; X[s2][1] = 0.0; X[s2][2] = 0.0;
        }
        // two rows of loads in flight per wave (as in k_chebk): rows j + 2 and j + 3 are asked for while rows j and j + 1 are
        // worked on; a fetch beyond the last row repeats it and is never consumed (no branch around the loads)
        const int jend = j1 + K;
        auto fetch = [&](int j, MgRaw &R) {
            mg_fetch<true, false, false, F32, (F32 && FINE) || (VT & 2), VT>(op, df, g, gc, b, x, e, j < jend ? j : jend - 1, cvalid, colo, coloc, R);
        };
        auto trip = [&](int j, const MgRaw &cur) {
            RC[0] = mg_row<F32, (VT & 2) != 0>(cur, df, op, colo);
            V[0] = cur.valid;
            X[0][2] = cur.valid ? (((F32 && FINE) || (VT & 2)) ? (double)cur.fx : cur.x) + gamma * ((VT & 1) ? (double)cur.fc : cur.e) : 0.0;   // prolongated iterate on row j
#pragma unroll
            for (int s2 = 1; s2 <= K; ++s2) {                               // sweep s2 on row j - s2
                double v = jac(X[s2 - 1][1], X[s2 - 1][0], X[s2 - 1][2], RC[s2], om.w[s2 - 1]);
                if (!V[s2]) v = 0.0;
                X[s2][2] = v;
                if (s2 == K) {
                    const int row = j - K;
                    if (row >= j0 && row < j1 && out) {
                        if (ZF) {
                            const float vf = (float)v;
                            store_stream_f32(reinterpret_cast<float *>(xn) + (size_t)(row + NG) * P + colo, vf, nts & 1);
                            dot += RC[K].b * (double)vf;
                        } else if (VT & 2) {
                            store_stream_f32(reinterpret_cast<float *>(xn) + (size_t)(row + NG) * P + colo, (float)v, nts & 1);
                        } else {
                            store_stream(xn + (size_t)(row + NG) * P + colo, (nts & 2) ? (double)(float)v : v, nts & 1);
                            if (FINE) dot += RC[K].b * v;
                        }
                    }
                }
            }
#pragma unroll
            for (int s2 = K; s2 >= 1; --s2) { RC[s2] = RC[s2 - 1]; V[s2] = V[s2 - 1]; }
#pragma unroll
            for (int s2 = 0; s2 <= K; ++s2) { X[s2][0] = X[s2][1]; X[s2][1] = X[s2][2]; }
        };
        MgRaw r0, r1;
        fetch(j0 - K, r0);
        fetch(j0 - K + 1, r1);
        for (int j = j0 - K; j < jend; j += 2) {
            {
                const MgRaw cur = r0;
                fetch(j + 2, r0);
                trip(j, cur);
            }
            if (j + 1 < jend) {
                const MgRaw cur = r1;
                fetch(j + 3, r1);
                trip(j + 1, cur);
            }
            if (WPB > 4) __syncthreads();       // lockstep: one barrier per two rows
        }
    }
    if (FINE) {
        __shared__ double sm[WPB];
        const double w = wsum(dot);
        if (lane == 0) sm[wave] = w;
        __syncthreads();
        if (threadIdx.x == 0) partials[blockIdx.x] = sum_waves<WPB>(sm);
    }
}

// Tail of a line-smoothed pre-smoothing pass in one marching pass: one damped point-Jacobi sweep on the iterate the
// line sweeps left, the residual of the result and its restriction (k_mg_jacobi + k_mg_restrict: 48 + 42 B per cell
// and 2 launches).  Two halo lanes / rows per side: the residual of a cell needs its neighbours after the sweep.
template <int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_mg_down_x(OpDev op, MarchGeo mg, DiagFix df, const double *b, const double *xin, double omega,
                                                        double *xout, Geo gc, double *bc) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t P = g.pitch;
    constexpr int K = 2, W = 64 - 2 * K;
    const MarchItems<WPB> items(mg.nstrips, mg.nsegs, mg.xcd, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, mg.nstrips, wave, strip, seg, active);
        const int iraw = strip * W - K + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= K && lane < 64 - K && iraw < g.ni && active;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // RC[s], V[s]: row j - s;  X0: the incoming iterate on rows (j - 2, j - 1, j);  X1: after the sweep on rows (j - 3, j - 2, j - 1)
        RowC RC[3];
        bool V[3];
        double X0[3] = {0.0, 0.0, 0.0}, X1[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) { RC[s2] = RowC{0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0}; V[s2] = false; }
        double rsum = 0.0;
        MgRaw nxt;
        mg_fetch<false, false, true>(op, df, g, gc, b, xin, nullptr, j0 - K, cvalid, colo, 0, nxt);
        for (int j = j0 - K; j < j1 + K; ++j) {
            const MgRaw cur = nxt;
            if (j + 1 < j1 + K) mg_fetch<false, false, true>(op, df, g, gc, b, xin, nullptr, j + 1, cvalid, colo, 0, nxt);
            RC[0] = mg_row(cur, df, op, colo);
            V[0] = cur.valid;
            X0[2] = cur.valid ? cur.x : 0.0;
            double v = jac(X0[1], X0[0], X0[2], RC[1], omega);              // the sweep on row j - 1
            if (!V[1]) v = 0.0;
            X1[2] = v;
            const int row = j - 1;
            if (row >= j0 && row < j1 && out) xout[(size_t)(row + NG) * P + colo] = v;
            // residual on row j - 2, summed over 2 x 2 aggregates into the coarse right-hand side
            const int jo = j - 2;
            const double res = resid(X1[1], X1[0], X1[2], RC[2]);
            if (jo >= j0 && jo < j1) {
                rsum += res;
                if (((jo - j0) & 1) || jo == j1 - 1) {
                    const double tot = rsum + lane_right(rsum);
                    if (out && !((lane - K) & 1)) bc[gc.cell(iraw >> 1, jo >> 1)] = (iraw + 1 < g.ni) ? tot : rsum;
                    rsum = 0.0;
                }
            }
            RC[2] = RC[1]; RC[1] = RC[0]; V[2] = V[1]; V[1] = V[0];
            X0[0] = X0[1]; X0[1] = X0[2];
            X1[0] = X1[1]; X1[1] = X1[2];
            if (WPB > 4) __syncthreads();
        }
    }
}

// ------------------------------------------------------------------ alternating line smoother
// Point Jacobi smooths only along the strongly coupled direction of an anisotropic operator: on a wall-graded channel
// (c_j / c_i up to (dx / dy_min)^2) or an annulus refined towards the cylinder (c_i >> c_j there) the V-cycle above
// needs 50 .. 300 PCG iterations.  On levels whose strong / weak coupling ratio exceeds Ctx::mg_line_threshold the
// smoother is block line Jacobi in BOTH directions plus one point sweep (pre: j-lines, i-lines, point; post: the
// reverse, so the V-cycle stays symmetric):  x <- x + omega M^-1 (b - A x),  M = the diagonal plus the couplings
// along the line direction inside a block.
//   j-lines (k_line_j): lane = column, one wavefront marches down a segment of <= 32 rows = one block; Thomas
//       elimination with factors precomputed per level (k_line_factor_j: multiplier m_j = c_s,j / p_{j-1}, inverse
//       pivot 1 / p_j); the forward-eliminated right-hand sides wait in LDS for the backward pass.
//   i-lines (k_line_i): lane = cell of the line, blocks of <= 62 cells between two halo lanes; parallel cyclic
//       reduction over the 64 lanes (6 steps of wave shuffles), couplings across the block ends explicit.
// Blocks are balanced (no short block at a wall, where the coupling is strongest).  CPU prototype and iteration
// counts: tools/mg_proto_aniso.py.
constexpr int LJ_MAX = 64;        // rows per j-block
constexpr int LI_W = 62;          // cells per i-block
constexpr int LJ_CHUNK = 8;       // rows whose loads k_line_j issues together

struct LineGeo {
    int nstrips, wstrip, nsegs, Ls;
    int nsegs_i, Ls_i;       // row segments of the i-line kernel (its rows are independent: a pure work split)
};
static inline LineGeo line_geo(const Geo &g) {
    LineGeo m;
    m.nstrips = (g.ni + LI_W - 1) / LI_W;
    m.wstrip = (g.ni + m.nstrips - 1) / m.nstrips;
    m.nsegs = (g.nj + LJ_MAX - 1) / LJ_MAX;
    m.Ls = (g.nj + m.nsegs - 1) / m.nsegs;
    int Li = 32;
    while (Li > 4 && (long)m.nstrips * ((g.nj + Li - 1) / Li) < 8192) Li /= 2;
    m.Ls_i = Li;
    m.nsegs_i = (g.nj + Li - 1) / Li;
    return m;
}
static inline int line_blocks(const LineGeo &m, bool iline = false) {
    const long nb = ((long)m.nstrips * (iline ? m.nsegs_i : m.nsegs) + 3) / 4;
    return (int)(nb < 2 * MAX_PARTIALS ? nb : 2 * MAX_PARTIALS);
}

// Thomas factors of the j-blocks: p_j0 = D_j0, p_j = D_j - c_s,j^2 / p_{j-1}; lm = c_s,j / p_{j-1} (0 on the first row
// of a block), linv = 1 / p_j.  One thread per column and block.
__global__ void k_line_factor_j(OpDev op, LineGeo lg, double *lm, double *linv) {
    const Geo g = op.g;
    const long n = (long)g.ni * lg.nsegs;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int i = (int)(t % g.ni), seg = (int)(t / g.ni);
        const int j0 = seg * lg.Ls, j1 = min(j0 + lg.Ls, g.nj);
        double pprev = 1.0;
        for (int j = j0; j < j1; ++j) {
            const size_t c = g.cell(i, j);
            const double cs = j > j0 ? op.cj[c] : 0.0;
            const double m = j > j0 ? cs / pprev : 0.0;
            const double pj = op.D[c] - m * cs;
            lm[c] = m;
            linv[c] = 1.0 / pj;
            pprev = pj;
        }
    }
}
// the same for the i-blocks (strips of wstrip cells): one thread per row and block
__global__ void k_line_factor_i(OpDev op, LineGeo lg, double *lm, double *linv) {
    const Geo g = op.g;
    const long n = (long)g.nj * lg.nstrips;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const int j = (int)(t % g.nj), strip = (int)(t / g.nj);
        const int i0 = strip * lg.wstrip, i1 = min(i0 + lg.wstrip, g.ni);
        double pprev = 1.0;
        for (int i = i0; i < i1; ++i) {
            const size_t c = g.cell(i, j);
            const double cw = i > i0 ? op.ci[c] : 0.0;
            const double m = i > i0 ? cw / pprev : 0.0;
            const double pi = op.D[c] - m * cw;
            lm[c] = m;
            linv[c] = 1.0 / pi;
            pprev = pi;
        }
    }
}

// max over the cells of (c_S + c_N) / (c_W + c_E) and of its inverse (cells without couplings in a direction skipped)
__global__ void k_level_aniso(OpDev op, Tiling T, double *partials) {
    const Geo g = op.g;
    double mj = 0.0, mi = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double a = op.ci[c] + op.ci[c + 1], b = op.cj[c] + op.cj[c + g.pitch];
        if (a > 0.0 && b > 0.0) { mj = fmax(mj, b / a); mi = fmax(mi, a / b); }
    }
    __shared__ double sa[4], sb[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mj = fmax(mj, __shfl_down(mj, off, 64)); mi = fmax(mi, __shfl_down(mi, off, 64)); }
    if ((tid & 63) == 0) { sa[tid >> 6] = mj; sb[tid >> 6] = mi; }
    __syncthreads();
    if (tid == 0) {      // negated: finished by a min (k_min_finish1) and a MIN all-reduce
        partials[blockIdx.x] = -fmax(fmax(sa[0], sa[1]), fmax(sa[2], sa[3]));
        partials[MAX_PARTIALS + blockIdx.x] = -fmax(fmax(sb[0], sb[1]), fmax(sb[2], sb[3]));
    }
}

// One block j-line Jacobi sweep.  ZERO: the iterate is zero (first pre-smoothing sweep), xin is not read.
// Lane = column; a wavefront marches down its block (forward elimination) and back up (substitution).  The
// forward-eliminated right-hand sides wait in the OUTPUT plane itself: each lane re-reads, last row first, what it
// stored a few microseconds earlier (L2 hits), so the kernel needs no LDS and no long register arrays and runs at
// full occupancy (it held 32 KiB of LDS per wavefront before: five wavefronts per CU, 2.1 ms per sweep at 8192^2).
template <bool ZERO>
__global__ __launch_bounds__(256) void k_line_j(OpDev op, LineGeo lg, const double *__restrict__ lm, const double *__restrict__ linv,
                                                const double *__restrict__ b, const double *__restrict__ xin, double omega,
                                                double *__restrict__ xout) {
    const Geo g = op.g;
    const double *__restrict__ ci = op.ci;
    const double *__restrict__ cj = op.cj;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const size_t P = g.pitch;
    for (int item = blockIdx.x * 4 + wave; item < lg.nstrips * lg.nsegs; item += nw) {
        const int strip = item % lg.nstrips, seg = item / lg.nstrips;
        const int iraw = strip * lg.wstrip - 1 + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= 1 && lane <= lg.wstrip && iraw < g.ni;
        const int j0 = seg * lg.Ls, j1 = min(j0 + lg.Ls, g.nj);
        // old iterate on the rows just outside the block (explicit couplings); zero outside the domain
        auto xrow = [&](int j) -> double {
            if (ZERO) return 0.0;
            const int mr = mem_row(j, g, op.per_j);
            return (cvalid && mr >= 0) ? xin[(size_t)mr * P + colo] : 0.0;
        };
        const double xlo = xrow(j0 - 1), xhi = xrow(j1);
        // both passes take LJ_CHUNK rows at a time: all loads of a chunk are issued before its dependent chain runs
        // (a coarse level has too few wavefronts to hide one memory latency per row)
        double gprev = 0.0;
        for (int jb = j0; jb < j1; jb += LJ_CHUNK) {
            double vb[LJ_CHUNK], vm[LJ_CHUNK], vx[LJ_CHUNK], vw[LJ_CHUNK], ve[LJ_CHUNK];
#pragma unroll
            for (int k = 0; k < LJ_CHUNK; ++k) {
                const size_t c = (size_t)(min(jb + k, j1 - 1) + NG) * P + colo;
                vb[k] = b[c];
                vm[k] = lm[c];
                if (!ZERO) { vx[k] = cvalid ? xin[c] : 0.0; vw[k] = ci[c]; ve[k] = ci[c + 1]; }
            }
#pragma unroll
            for (int k = 0; k < LJ_CHUNK; ++k) {
                const int j = jb + k;
                if (j < j1) {
                    const size_t c = (size_t)(j + NG) * P + colo;
                    double rhs = vb[k];
                    if (!ZERO) {
                        const double xl = lane_left(vx[k]), xr = lane_right(vx[k]);
                        rhs += vw[k] * xl + ve[k] * xr;
                    }
                    if (j == j0) rhs += cj[c] * xlo;
                    if (j == j1 - 1) rhs += cj[c + P] * xhi;
                    const double gj = rhs + vm[k] * gprev;
                    if (out) xout[c] = gj;
                    gprev = gj;
                }
            }
        }
        double xnext = 0.0;
        for (int jb = j1 - 1; jb >= j0; jb -= LJ_CHUNK) {
            double vg[LJ_CHUNK], vn[LJ_CHUNK], vi[LJ_CHUNK], vo[LJ_CHUNK];
#pragma unroll
            for (int k = 0; k < LJ_CHUNK; ++k) {
                const size_t c = (size_t)(max(jb - k, j0) + NG) * P + colo;
                vg[k] = xout[c];
                vn[k] = cj[c + P];
                vi[k] = linv[c];
                vo[k] = ZERO ? 0.0 : xin[c];
            }
#pragma unroll
            for (int k = 0; k < LJ_CHUNK; ++k) {
                const int j = jb - k;
                if (j >= j0) {
                    const size_t c = (size_t)(j + NG) * P + colo;
                    const double cn = j < j1 - 1 ? vn[k] : 0.0;
                    const double xt = (vg[k] + cn * xnext) * vi[k];
                    xnext = xt;
                    if (out) xout[c] = vo[k] + omega * (xt - vo[k]);
                }
            }
        }
    }
}

// One block i-line Jacobi sweep.  Lane = cell of the line: the two first-order recurrences of the Thomas algorithm
// (elimination g_l = r_l + m_l g_{l-1}, substitution x_l = h_l + e_l x_{l+1}) are inclusive scans over the wavefront,
// six DPP steps each (row_shr 1, 2, 4, 8, row_bcast 15, row_bcast 31; no LDS); the multipliers m and the inverse
// pivots come from k_line_factor_i.  The substitution runs in mirrored lane order (lane l holds column 63 - l: its
// operands are loaded mirrored, only the eliminated right-hand side crosses the lanes, by one ds_bpermute pair), so
// that it is a forward scan too.  (The parallel cyclic reduction it replaces needed 96 ds_bpermute and seven
// reciprocals per row: 3.8 ms per sweep at 8192^2.)
__device__ __forceinline__ double shfl_from(double v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_take(double keep, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), CTRL, ROWMASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), CTRL, ROWMASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void scan_step(double &m, double &v) {
    const double M = dpp_take<CTRL, ROWMASK>(1.0, m);       // lanes without a source keep the identity (1, 0)
    const double V = dpp_take<CTRL, ROWMASK>(0.0, v);
    v = v + m * V;
    m = m * M;
}
// y_l = v_l + m_l y_{l-1} over the 64 lanes (y_{-1} = 0): returns y_l
__device__ __forceinline__ double wave_recurrence(double m, double v) {
    scan_step<0x111, 0xf>(m, v);        // row_shr:1
    scan_step<0x112, 0xf>(m, v);        // row_shr:2
    scan_step<0x114, 0xf>(m, v);        // row_shr:4
    scan_step<0x118, 0xf>(m, v);        // row_shr:8
    scan_step<0x142, 0xa>(m, v);        // row_bcast:15 into rows 1 and 3
    scan_step<0x143, 0xc>(m, v);        // row_bcast:31 into rows 2 and 3
    return v;
}
template <bool ZERO>
__global__ __launch_bounds__(256) void k_line_i(OpDev op, LineGeo lg, const double *__restrict__ lm, const double *__restrict__ linv,
                                                const double *__restrict__ b, const double *__restrict__ xin, double omega,
                                                double *__restrict__ xout) {
    const Geo g = op.g;
    const double *__restrict__ ci = op.ci;
    const double *__restrict__ cj = op.cj;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const size_t P = g.pitch;
    for (int item = blockIdx.x * 4 + wave; item < lg.nstrips * lg.nsegs_i; item += nw) {
        const int strip = item % lg.nstrips, seg = item / lg.nstrips;
        const int nu = min(lg.wstrip, g.ni - strip * lg.wstrip);            // unknowns of this block: lanes 1 .. nu
        const int iraw = strip * lg.wstrip - 1 + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool act = lane >= 1 && lane <= nu;
        // mirrored lane order of the substitution
        const int lane_m = 63 - lane;
        const int col_m = wrap_index(strip * lg.wstrip - 1 + lane_m, g.ni, op.per_i);
        const size_t colo_m = (size_t)(col_m >= 0 ? col_m : 0) + CO;
        const bool act_m = lane_m >= 1 && lane_m <= nu;
        const int j0 = seg * lg.Ls_i, j1 = min(j0 + lg.Ls_i, g.nj);
        auto xrow = [&](int j) -> double {
            if (ZERO) return 0.0;
            const int mr = mem_row(j, g, op.per_j);
            return (cvalid && mr >= 0) ? xin[(size_t)mr * P + colo] : 0.0;
        };
        double xs = xrow(j0 - 1), xc = xrow(j0);
#pragma unroll 2
        for (int j = j0; j < j1; ++j) {
            const size_t c = (size_t)(j + NG) * P + colo, cm = (size_t)(j + NG) * P + colo_m;
            const double xn = xrow(j + 1);
            double m = 0.0, r = 0.0;
            if (act) {
                m = lm[c];
                r = b[c];
                if (!ZERO) r += cj[c] * xs + cj[c + P] * xn;
            }
            if (!ZERO) {        // old values of the two cells next to the block ends (explicit couplings)
                const double xl = lane_left(xc), xr = lane_right(xc);
                if (lane == 1) r += ci[c] * xl;
                if (lane == nu) r += ci[c + 1] * xr;
            }
            const double gl = wave_recurrence(m, r);
            const double gm = shfl_from(gl, lane_m);
            double e = 0.0, h = 0.0, xo = 0.0;
            if (act_m) {
                const double iv = linv[cm];
                h = gm * iv;
                if (lane_m < nu) e = ci[cm + 1] * iv;
                if (!ZERO) xo = xin[cm];
            }
            const double xt = wave_recurrence(e, h);
            if (act_m) xout[cm] = xo + omega * (xt - xo);
            xs = xc; xc = xn;
        }
    }
}

// fused marching kernels: two or three sweeps per pass (a slab cut carries NG = 4 halo rows)
static inline bool level_fusable(const Ctx *c, int lvl, const Geo &F, const Geo &C) {
    const bool k_ok = c->mg_pre == c->mg_post && (c->mg_pre == 2 || c->mg_pre == 3);
    if (c->levels[lvl].line || c->semi_on) return false;       // line-smoothed levels run their own kernels
    return c->mg_fused && k_ok && F.ni >= 16 && F.nj >= 16 && C.ni == (F.ni + 1) / 2 && C.nj == (F.nj + 1) / 2;
}
static inline MarchGeo march_geo_k(const Geo &g, int K, bool march_ls_model);
constexpr long MARCH_WAVE_SLOTS = 4096;      // 256 CUs x 4 SIMDs x 4 waves (amdgpu_waves_per_eu(4, 4) of the three-sweep kernels)
static inline int march_blocks(const MarchGeo &m) {
    const long items = (long)m.nstrips * m.nsegs;
    long nb = (items + 3) / 4;
    if (m.xcd) nb = (nb + 7) / 8 * 8;
    return (int)(nb < MAX_PARTIALS ? nb : MAX_PARTIALS);
}
// lockstep workgroups (MarchItems<8>): on meshes with enough strips to fill them
// which: WIDE_* bit of the kernel asking (Ctx::wide_mask holds the kernels that measured faster in lockstep)
enum { WIDE_CG_PQ = 1, WIDE_CHEB1 = 2, WIDE_CHEB2 = 4, WIDE_MG_UP = 8, WIDE_MG_DOWN_RF = 16, WIDE_MG_DOWN = 32, WIDE_MG_COARSE = 64 };
static inline int march_wide(const Ctx *c, const MarchGeo &m, int which) {
    return ((c->wide_mask & which) && m.nstrips >= 16) ? 8 : 4;
}
static inline int march_grid(const MarchGeo &m, int wide, bool lock4 = false) {
    if (wide <= 4 && !lock4) return march_blocks(m);
    const long items = (long)((m.nstrips + wide - 1) / wide) * m.nsegs;
    return (int)(items < MAX_PARTIALS ? items : MAX_PARTIALS);
}
// K halo lanes / rows per side: narrower strips, longer segments to amortise the 2K extra rows
static inline MarchGeo march_geo_k(const Geo &g, int K, bool march_ls_model) {
    if (K <= 2) return march_geo(g);
    MarchGeo m;
    m.nstrips = (g.ni + (64 - 2 * K) - 1) / (64 - 2 * K);
    int Ls = 64;
    auto items_at = [&](int len) { return (long)m.nstrips * ((g.nj + len - 1) / len); };
    if (march_ls_model && items_at(64) < 3 * MARCH_WAVE_SLOTS) {
        // A level that fills the GPU's wave slots (four waves per SIMD) less than three times over: its launch lasts
        // rounds x (rows per item) row trips, a round being one item per slot -- 2.2 rounds cost three, and halving the
        // segments to get more items buys 2K redundant rows per item (4096^2 at Ls = 32: 9088 items of 38 rows, three
        // rounds; 2048^2 at Ls = 8: 14 rows per item for 8 of output).  Take the segment length that minimises
        // rounds x (Ls + 2K + start-up); levels below one round end at the shortest chain of dependent row trips.
        double best = 1.0e30;
        for (int len = 2; len <= 64; len += 2) {       // (even: a segment holds whole 2 x 2 aggregates)
            const long rounds = (items_at(len) + MARCH_WAVE_SLOTS - 1) / MARCH_WAVE_SLOTS;
            const double cost = (double)rounds * (len + 2 * K + 3);
            if (cost < best) { best = cost; Ls = len; }
        }
        m.Ls = Ls;
        m.nsegs = (g.nj + Ls - 1) / Ls;
        return m;
    }
    while (Ls > 8 && items_at(Ls) < 8192) Ls /= 2;
    // levels too small to fill the GPU even then are chains of dependent row trips (Ls + 2K of them per launch):
    // shorter segments shorten the chain, the redundant halo rows cost nothing there
    // (8192^2: 43.35 -> 43.15 ms per step from the 1024^2 ... 128^2 levels; 1024^2: see DESIGN section 5)
    while (Ls > 2 && items_at(Ls) < 2048) Ls /= 2;
    m.Ls = Ls;
    m.nsegs = (g.nj + Ls - 1) / Ls;
    return m;
}

// ------------------------------------------------------------------ Chebyshev (time-step dominated systems)
// The transport matrices are  A = V/dt + sum a_f  on the diagonal and -a_f off it.  With
// Jacobi scaling the spectrum of D^-1 A lies in [1 - rho, 1 + rho], rho = max_P sum_n c_n / D_P
// (row-sum bound, rigorous).  When the step is CFL-limited rho is well below one (2/3 at
// cfl_diff = 0.5 on a uniform mesh) and the Chebyshev semi-iteration on that interval converges
// like CG *without any inner product*: one fused pass per iteration
//     r_k = b - A x_k ; d_k = c1 (x_k - x_{k-1}) + c2 r_k / D ; x_{k+1} = x_k + d_k
// (88 B/cell for two right-hand sides), |r_k|^2 falls out as a by-product and the host looks at
// it only around the iteration the Chebyshev bound predicts.  The stopping test is the
// reference's, |b - A x| <= tol |b| on the returned iterate.  Systems with rho close to one
// (large dt) go to BiCGStab below.
__global__ void k_rowsum_bound(OpDev op, Tiling T, double *partials) {
    const Geo g = op.g;
    double m = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        m = fmax(m, offdiag(op, [](size_t) { return 1.0; }, c, i, j) / op.D[c]);
    }
    __shared__ double sm[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, 64));
    if ((tid & 63) == 0) sm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = -fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));   // negated: finished by a min
}
__global__ void k_min_finish1(const double *partials, int n, double *out) {
    __shared__ double sm[256];
    double v = INFINITY;
    for (int t = threadIdx.x; t < n; t += 256) v = fmin(v, partials[t]);
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = fmin(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// partials: [0, NB) = |r_k|^2, and on the first pass [NB, 2NB) = |b|^2
template <int NB>
__global__ void k_cheb(OpDev op, Tiling T, double c1, double c2, int first, CVecs<NB> b, CVecs<NB> xk, CVecs<NB> xm,
                       Vecs<NB> xo, double *partials) {
    const Geo g = op.g;
    double acc[2 * NB];
#pragma unroll
    for (int q = 0; q < 2 * NB; ++q) acc[q] = 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double d = op.D[c];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const double *xr = xk.p[q];
            const double xc = xr[c], bq = b.p[q][c];
            const double r = bq - (d * xc - offdiag(op, [xr](size_t n) { return xr[n]; }, c, i, j));
            acc[q] += r * r;
            double step = c2 * (r / d);
            if (first) acc[NB + q] += bq * bq;
            else step += c1 * (xc - xm.p[q][c]);
            xo.p[q][c] = xc + step;
        }
    }
    block_sums<2 * NB>(acc, partials);
}


// Two Chebyshev iterations per pass, marching like the multigrid kernels above: one wavefront owns
// a 60-column strip and walks down a row segment with x_k, x_{k+1} and the operator rows in
// registers.  Per cell and launch it reads x_k, x_{k-1}, b (per right-hand side) and D, c_i, c_j once
// and writes x_{k+1}, x_{k+2}: 104 B for two iterations of two right-hand sides (the one-iteration
// kernel above moves 88 B per iteration).  |r_k|^2, |r_{k+1}|^2 (and |b|^2 on the first launch) come
// out as partial sums [q], [NB + q], [2 NB + q].
struct RowA {
    double d, inv, cw, ce, cs, cn;
};
template <int NB>
__global__ __launch_bounds__(256) void k_cheb2(OpDev op, MarchGeo mg, double c1a, double c2a, double c1b, double c2b,
                                               int first, CVecs<NB> b, CVecs<NB> xk, CVecs<NB> xm, Vecs<NB> x1o,
                                               Vecs<NB> x2o, double *partials) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const int nw = gridDim.x * 4;
    const size_t P = g.pitch;
    double acc[3 * NB];
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) acc[q] = 0.0;
    for (int item = blockIdx.x * 4 + wave; item < mg.nstrips * mg.nsegs; item += nw) {
        const int strip = item % mg.nstrips, seg = item / mg.nstrips;
        const int iraw = strip * MW - MH + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= MH && lane < 64 - MH && iraw < g.ni;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // row j: operator row + x_k ; returns the cell index (clamped to a valid cell when outside)
        auto load = [&](int j, bool &valid, RowA &a, double (&xv)[NB]) -> size_t {
            const int mr = mem_row(j, g, op.per_j);
            valid = cvalid && mr >= 0;
            const size_t c = (size_t)(mr >= 0 ? mr : NG) * P + colo;
            a.d = op.D[c]; a.cw = op.ci[c]; a.ce = op.ci[c + 1]; a.cs = op.cj[c]; a.cn = op.cj[c + north_offset(mr < g.nj + 2 * NG - 1, P)];
            a.inv = fast_rcp(a.d);
#pragma unroll
            for (int q = 0; q < NB; ++q) xv[q] = valid ? xk.p[q][c] : 0.0;
            return c;
        };
        RowA ra0, ra1, ra2;
        double xa[NB], xb[NB], xc[NB], ya[NB], yb[NB], yc[NB], b2[NB];
        bool v0, v1, v2;
        size_t cell1, cell2 = 0;
        load(j0 - 2, v2, ra2, xa);
        cell1 = load(j0 - 1, v1, ra1, xb);
#pragma unroll
        for (int q = 0; q < NB; ++q) { ya[q] = 0.0; yb[q] = 0.0; yc[q] = 0.0; b2[q] = 0.0; }
        for (int j = j0; j < j1 + 2; ++j) {
            const size_t cell0 = load(j, v0, ra0, xc);
            const bool in1 = j - 1 >= j0 && j - 1 < j1 && out, in2 = j - 2 >= j0 && j - 2 < j1 && out;
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                // first iteration on row j-1
                const double bq = b.p[q][cell1];
                const double xl = lane_left(xb[q]), xr = lane_right(xb[q]);
                const double r = bq - (ra1.d * xb[q] - (ra1.cw * xl + ra1.ce * xr + ra1.cs * xa[q] + ra1.cn * xc[q]));
                double step = c2a * (r * ra1.inv);
                if (!first) step += c1a * (xb[q] - xm.p[q][cell1]);
                yc[q] = v1 ? xb[q] + step : 0.0;
                if (in1) {
                    x1o.p[q][cell1] = yc[q];
                    acc[q] += r * r;
                    if (first) acc[2 * NB + q] += bq * bq;
                }
                // second iteration on row j-2
                const double yl = lane_left(yb[q]), yr = lane_right(yb[q]);
                const double r2 = b2[q] - (ra2.d * yb[q] - (ra2.cw * yl + ra2.ce * yr + ra2.cs * ya[q] + ra2.cn * yc[q]));
                if (in2) {
                    x2o.p[q][cell2] = yb[q] + c1b * (yb[q] - xa[q]) + c2b * (r2 * ra2.inv);
                    acc[NB + q] += r2 * r2;
                }
                xa[q] = xb[q]; xb[q] = xc[q];
                ya[q] = yb[q]; yb[q] = yc[q];
                b2[q] = bq;
            }
            ra2 = ra1; ra1 = ra0;
            v2 = v1; v1 = v0;
            cell2 = cell1; cell1 = cell0;
        }
    }
    __shared__ double sm[3 * NB][4];
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) {
        const double w = wsum(acc[q]);
        if (lane == 0) sm[q][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < 3 * NB)
        partials[(size_t)threadIdx.x * MAX_PARTIALS + blockIdx.x] =
            (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

// K Chebyshev iterations per pass (same marching scheme, K halo lanes / rows on each side): x_k,
// x_{k-1}, b, D, c_i, c_j are still read once and x_{k+K-1}, x_{k+K} written, so the 104 B buy K
// iterations.  Stage s (1..K) produces iterate k+s on row j-s while row j is being loaded; |r_k|^2 and
// |r_{k+K-1}|^2 (the residuals of the two iterates that exist in memory afterwards) are the partial
// sums [q] and [NB + q], |b|^2 is [2 NB + q] on the first launch.  At a slab cut the K halo rows come
// from the neighbour (K <= NG).
//
// The kernel runs two (NB = 2) or three waves per SIMD, so what a wave keeps in flight decides how close it gets
// to the memory system's rate: PF = 2 keeps TWO rows of loads outstanding (the row being consumed was requested
// two trips earlier).  The registers for that come from not holding what can be derived: the north coupling of a
// row is the south coupling of the row above it (one register per segment covers the periodic-j seam, whose two
// copies are separate plane entries), and the store address is recomputed from the row index.
template <int K>
struct ChebStages {
    double c1[K], c2[K];
};
struct RowB {
    double d, inv, cw, ce, cs;
};
template <int NB, int K, int PF, int WPB = 4, bool LOCK = (WPB > 4)>
__global__ __launch_bounds__(64 * WPB) void k_chebk(OpDev op, MarchGeo mg, ChebStages<K> cf, int first, int nts, CVecs<NB> b,
                                                    CVecs<NB> xk, CVecs<NB> xm, Vecs<NB> xlo, Vecs<NB> xhi, double *partials) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const size_t P = g.pitch;
    constexpr int W = 64 - 2 * K;
    double acc[3 * NB];
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) acc[q] = 0.0;
    const MarchItems<WPB, LOCK> items(mg.nstrips, mg.nsegs, mg.xcd, wave);
    for (int item = items.first; item < items.count; item += items.stride) {
        int strip, seg;
        bool active;
        items.decode(item, mg.nstrips, wave, strip, seg, active);
        const int iraw = strip * W - K + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= K && lane < 64 - K && iraw < g.ni && active;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        // north coupling of the last row of a periodic-j mesh (and of its wrapped copy, row -1): the seam's own entry
        const double cn_top = op.per_j ? op.cj[(size_t)(g.nj + NG) * P + colo] : 0.0;
        // X[s][0..2]: iterate k+s on rows (j-s-2, j-s-1, j-s); RA[s], B[s], V[s]: row j-s
        double X[K + 1][3][NB], B[K + 1][NB];
        RowB RA[K + 1];
        bool V[K + 1];
#pragma unroll
        for (int s2 = 0; s2 <= K; ++s2) {
            V[s2] = false;
            RA[s2] = RowB{1.0, 1.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < NB; ++q) { X[s2][0][q] = 0.0; X[s2][1][q] = 0.0; X[s2][2][q] = 0.0; B[s2][q] = 0.0; }
        }
        struct RowIn {
            double d, cw, ce, cs, x[NB], b[NB], xm[NB];
            bool valid;
        };
        // unconditional loads from a clamped (always valid) cell: rows past the end of the segment are fetched and
        // never consumed, which keeps the load / wait pattern of the pipeline free of branches
        const int ja = j0 - K, jb = j1 + K;
        auto fetch = [&](int j, RowIn &R) {
            const int mr = mem_row(j < jb ? j : jb - 1, g, op.per_j);
            R.valid = cvalid && mr >= 0;
            const size_t c = (size_t)(mr >= 0 ? mr : NG) * P + colo;
            R.d = op.D[c]; R.cw = op.ci[c]; R.ce = op.ci[c + 1]; R.cs = op.cj[c];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                R.x[q] = xk.p[q][c];
                R.b[q] = b.p[q][c];
                R.xm[q] = xm.p[q][c];      // first launch of a solve: xm = xk (same plane, the load hits), c1[0] = 0
            }
        };
        double XM[2][NB];          // x_{k-1} on rows j, j-1
#pragma unroll
        for (int q = 0; q < NB; ++q) { XM[0][q] = 0.0; XM[1][q] = 0.0; }
        // one trip of the pipeline: row j (in `cur`) enters, stage s advances row j - s
        auto trip = [&](int j, const RowIn &cur) {
            V[0] = cur.valid;
            RA[0].d = cur.d; RA[0].cw = cur.cw; RA[0].ce = cur.ce; RA[0].cs = cur.cs;
            RA[0].inv = fast_rcp(cur.d);
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                X[0][2][q] = cur.valid ? cur.x[q] : 0.0;
                B[0][q] = cur.b[q];
                XM[0][q] = cur.xm[q];
            }
#pragma unroll
            for (int s2 = 1; s2 <= K; ++s2) {
                const int row = j - s2;
                const bool inr = row >= j0 && row < j1 && out;
                const RowB ra = RA[s2];
                // north coupling of row j - s2 = south coupling of row j - s2 + 1, except across the periodic-j seam
                const double cn = (op.per_j && (row == g.nj - 1 || row == -1)) ? cn_top : RA[s2 - 1].cs;
                const size_t cell = (size_t)(row + NG) * P + colo;       // used by in-range rows only
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const double xc = X[s2 - 1][1][q];
                    const double xl = lane_left(xc), xr = lane_right(xc);
                    const double r = B[s2][q] - (ra.d * xc - (ra.cw * xl + ra.ce * xr + ra.cs * X[s2 - 1][0][q] + cn * X[s2 - 1][2][q]));
                    double step = cf.c2[s2 - 1] * (r * ra.inv);
                    if (s2 == 1) {
                        step += cf.c1[0] * (xc - XM[1][q]);        // c1[0] = 0 on the first launch of a solve
                    } else {
                        step += cf.c1[s2 - 1] * (xc - X[s2 - 2][0][q]);
                    }
                    const double xn = V[s2] ? xc + step : 0.0;
                    X[s2][2][q] = xn;
                    if (inr) {
                        if (s2 == 1) {
                            acc[q] += r * r;
                            if (first) acc[2 * NB + q] += B[1][q] * B[1][q];
                        }
                        if (s2 == K) acc[NB + q] += r * r;
                        if (s2 == K - 1) store_stream(xlo.p[q] + cell, xn, nts);
                        if (s2 == K) store_stream(xhi.p[q] + cell, xn, nts);
                    }
                }
            }
            // slide every window one row down
#pragma unroll
            for (int s2 = K; s2 >= 1; --s2) {
                RA[s2] = RA[s2 - 1]; V[s2] = V[s2 - 1];
#pragma unroll
                for (int q = 0; q < NB; ++q) B[s2][q] = B[s2 - 1][q];
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) XM[1][q] = XM[0][q];
#pragma unroll
            for (int s2 = 0; s2 <= K; ++s2)
#pragma unroll
                for (int q = 0; q < NB; ++q) { X[s2][0][q] = X[s2][1][q]; X[s2][1][q] = X[s2][2][q]; }
        };
        if (PF == 2) {
            RowIn r0, r1;
            fetch(ja, r0);
            fetch(ja + 1, r1);
            for (int j = ja; j < jb; j += 2) {
                {
                    const RowIn cur = r0;
                    fetch(j + 2, r0);
                    trip(j, cur);
                }
                if (j + 1 < jb) {
                    const RowIn cur = r1;
                    fetch(j + 3, r1);
                    trip(j + 1, cur);
                }
                if (LOCK) __syncthreads();       // lockstep: ja, jb depend on the segment only
            }
        } else {
            RowIn nxt;
            fetch(ja, nxt);
            for (int j = ja; j < jb; ++j) {
                const RowIn cur = nxt;
                fetch(j + 1, nxt);
                trip(j, cur);
                if (LOCK) __syncthreads();
            }
        }
    }
    __shared__ double sm[3 * NB][WPB];
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) {
        const double w = wsum(acc[q]);
        if (lane == 0) sm[q][wave] = w;
    }
    __syncthreads();
    if (threadIdx.x < 3 * NB) partials[(size_t)threadIdx.x * MAX_PARTIALS + blockIdx.x] = sum_waves<WPB>(sm[threadIdx.x]);
}

// Chebyshev coefficient stream of Saad's Algorithm 12.1 on [1 - rho, 1 + rho]
struct ChebCoef {
    double rho, sigma1, rk;
    int k = 0;
    explicit ChebCoef(double rho_) : rho(rho_), sigma1(rho_ > 1e-300 ? 1.0 / rho_ : 1e300), rk(rho_) {}
    // coefficients of iteration k -> k+1:  x_{k+1} = x_k + c1 (x_k - x_{k-1}) + c2 r_k / D
    void next(double &c1, double &c2) {
        if (k == 0) { c1 = 0.0; c2 = 1.0; }
        else {
            const double rnew = 1.0 / (2.0 * sigma1 - rk);
            c1 = rnew * rk; c2 = 2.0 * rnew / rho;
            if (rho < 1.0e-14) { c1 = 0.0; c2 = 1.0; }     // diagonal matrix: x_1 was already exact
            rk = rnew;
        }
        ++k;
    }
    // iterations the bound 2 s^k / (1 + s^2k) <= need asks for
    int predict(double need) const {
        const double s = rho > 1e-300 ? sigma1 - sqrt(sigma1 * sigma1 - 1.0) : 0.0;
        if (!(s > 0.0) || !(need < 1.0)) return 1;
        const double kp = log(need / 2.0) / log(s);
        return kp > 1.0 ? (kp < 1.0e6 ? (int)ceil(kp) : 1000000) : 1;
    }
};

// The result of a solve sits in a Krylov work plane `where`, the caller expects it in the context plane `named`
// (u*, v*, k, omega): instead of copying the plane, the two exchange roles -- the context's pointer is rebound to
// `where` and the work-plane slot takes the old plane.  Same bits everywhere in the named plane as after a whole-plane
// copy.  false: one of the two is not a plane the context can rebind (the caller copies).
static bool adopt_plane(Ctx *c, double *named, double *where) {
    double **slot = nullptr, **work = nullptr;
    double **cand[6] = {&c->u[0], &c->u[1], &c->v[0], &c->v[1], &c->k, &c->omega};
    for (double **p : cand)
        if (*p == named) slot = p;
    for (int q = 0; q < 16; ++q)
        if (c->kry[q] == where) work = &c->kry[q];
    if (!slot || !work) return false;
    *work = named;
    *slot = where;
    return true;
}

template <int NB, int K>
static int chebyshev_fused(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2],
                           double tol, int maxiter, int *iters, double relres[2], double rms[2], int *used) {
    const Geo g = op.g;
    hipStream_t st = c->stream;
    MarchGeo mgeo = march_geo(g);
    if (K > 2) {
        // K halo lanes / rows per side: wider strips overlap, longer segments amortise the 2K extra rows
        mgeo.nstrips = (g.ni + (64 - 2 * K) - 1) / (64 - 2 * K);
        int Ls = c->march_ls > 0 ? c->march_ls : 32;
        while (Ls > 8 && (long)mgeo.nstrips * ((g.nj + Ls - 1) / Ls) < 8192) Ls /= 2;
        mgeo.Ls = Ls;
        mgeo.nsegs = (g.nj + Ls - 1) / Ls;
        mgeo.xcd = c->xcd_map;
    }
    // Six fused iterations for the launches that carry ONE right-hand side (option cheb_k6, default on; single rank: a cut
    // holds four halo rows): the one-right-hand-side form of K = 4 runs one workgroup of eight waves per CU at 154 registers,
    // and so does K = 6 below 256 -- the same 64 B per cell buy six iterations for 12 instead of 8 halo lanes
    // (v of the bench channel: 8 + 12 iterations in 2 + 2 launches instead of 2 + 3).  The stages are the same expressions
    // on the same operands: every iterate has the bits of the K = 4 sequence.
    constexpr int K6 = 6;
    MarchGeo mgeo6 = mgeo;
    mgeo6.nstrips = (g.ni + (64 - 2 * K6) - 1) / (64 - 2 * K6);
    // lockstep workgroups: per launch (one or two right-hand sides), see the loop
    const bool wide_ok = K == 4 && c->cheb_prefetch == 2;
    const int wide2 = wide_ok ? march_wide(c, mgeo, WIDE_CHEB2) : 4, wide1 = wide_ok ? march_wide(c, mgeo, WIDE_CHEB1) : 4;
    // iterate buffers: launch m writes x_{Km-1}, x_{Km} into set m & 1 and reads set (m - 1) & 1
    double *set[2][2][NB];
    CVecs<NB> bc;
    for (int q = 0; q < NB; ++q) {
        bc.p[q] = b[q];
        set[1][0][q] = (x[q] == x0[q]) ? c->kry[11 + q] : x[q];     // x may alias x0 (k / omega)
        set[1][1][q] = c->kry[5 + q];
        set[0][0][q] = c->kry[7 + q];
        set[0][1][q] = c->kry[9 + q];
    }
    if (c->comm.nranks > 1) {
        // the K fused iterations read K rows across a cut
        // (operator planes, right-hand sides, starting vectors: one exchange)
        if (g.cutlo || g.cuthi) {
            HaloSpec sp[7] = {{&g, K, const_cast<double *>(op.ci)}, {&g, K, const_cast<double *>(op.cj)}, {&g, K, const_cast<double *>(op.D)}};
            int ns = 3;
            for (int q = 0; q < NB; ++q) sp[ns++] = HaloSpec{&g, K, const_cast<double *>(x0[q])};
            for (int q = 0; q < NB; ++q) sp[ns++] = HaloSpec{&g, K - 1, b[q]};
            CFD_TRY(halo_exchange_multi(c, ns, sp));
        }
    }
    const double rho = -c->h_scal[12];
    ChebCoef cc(rho);
    // The right-hand sides converge at different rates (v of a channel starts from r0 / |b| ~ 0.1, u from
    // ~1e-5): each one leaves the iteration as soon as its own test |b - A x| <= tol |b| holds, and the
    // launches that remain carry the other one alone (64 instead of 104 B/cell).
    double bb[2] = {0, 0};
    bool act[2] = {true, NB == 2};
    int kq[2] = {1, 1}, kdone[2] = {0, 0};
    int klimit = maxiter;
    auto conv = [&](int q, double r2) { return sqrt(r2) <= tol * sqrt(bb[q]); };
    // where the result of a retired right-hand side waits (its planes are not touched by the launches that follow);
    // it reaches x[q] when the solve ends: by rebinding the plane pointers (adopt_plane) when every right-hand side has
    // converged here, by a copy when the solve goes on in BiCGStab, which works on the caller's x[q]
    double *final_where[2] = {nullptr, nullptr};
    auto deliver = [&](bool may_swap) -> int {
        for (int q = 0; q < NB; ++q) {
            if (!final_where[q] || final_where[q] == x[q]) continue;
            if (!(may_swap && c->plane_swap && adopt_plane(c, x[q], final_where[q])))
                CFD_HIP(hipMemcpyAsync(x[q], final_where[q], g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
            final_where[q] = nullptr;
        }
        return CFD_OK;
    };
    auto retire = [&](int q, int k, double r2, double *where) -> int {
        final_where[q] = where;
        relres[q] = bb[q] > 0.0 ? sqrt(r2 / bb[q]) : 0.0;
        if (rms) rms[q] = sqrt(r2 / global_cells(c));
        act[q] = false;
        kdone[q] = k;
        return CFD_OK;
    };
    const bool lag = c->host_lag && c->cheb_lag_ok;
    bool first_pending = false;
    // speculative replay (internal.h: Ctx::speculate): this solve takes each right-hand side from the iterate the previous
    // solve with this operator took it from; the sums its stopping test needs stay on the device until the iteration ends
    const Ctx::ChebPlan plan = c->cheb_plan;
    const int spec_kmax = std::min(maxiter, K * Ctx::SPEC_MAX_LAUNCHES);
    const bool spec = c->spec_on && !c->spec_cheb.used && plan.valid && plan.D == op.D && plan.nb == NB && plan.K == K &&
                      plan.tol == tol && plan.retire_k[0] < spec_kmax && plan.retire_k[NB - 1] < spec_kmax;
    if (!spec && op.D == c->ap) c->cheb_plan.valid = false;      // renewed below if this solve ends inside the Chebyshev iteration
    if (spec) {
        Ctx::SpecCheb &sc = c->spec_cheb;
        sc = Ctx::SpecCheb{};
        sc.used = true; sc.nb = NB; sc.K = K; sc.tol = tol;
    }
    int k_next = 0;                    // iterate the next launch starts from
    unsigned char kseq[Ctx::SPEC_MAX_LAUNCHES] = {};      // iterations fused by launch m of this solve
    for (int m = 1;; ++m) {
        int qs[2], na = 0;
        for (int q = 0; q < NB; ++q)
            if (act[q]) qs[na++] = q;
        double *cur_k[2] = {nullptr, nullptr}, *cur_m[2] = {nullptr, nullptr};
        for (int q = 0; q < NB; ++q) {
            cur_k[q] = m == 1 ? const_cast<double *>(x0[q]) : set[(m - 1) & 1][1][q];
            cur_m[q] = m == 1 ? const_cast<double *>(x0[q]) : set[(m - 1) & 1][0][q];
        }
        if (m > 1 && c->comm.nranks > 1) {
            // x_k needs K rows, x_{k-1} K - 1: a single message set of depth K for all of them
            if (na == 2) CFD_TRY(halo_exchange(c, g, K, cur_k[0], cur_k[1], cur_m[0], cur_m[1]));
            else CFD_TRY(halo_exchange(c, g, K, cur_k[qs[0]], cur_m[qs[0]]));
        }
        const int probe_id = (na == 2) ? 8 : 9;      // 8: both right-hand sides in the launch, 9: one
        const int wide_in = na == 2 ? wide2 : wide1;
        // (the first launch of a solve keeps K: its sums are read one launch late as those of iterates 0 and K - 1)
        const bool six_ok = K == 4 && c->cheb_k6 && m > 1 && na == 1 && NB == 2 && c->comm.nranks == 1 && wide_ok && wide_in == 8 &&
                            !(mgeo.nstrips >= 16 && (c->wide_mask & 512) && (c->wide_mask & WIDE_CHEB1)) && mgeo6.nstrips >= 16;
        // ... where that saves launches: a six-iteration launch costs about 1.2 four-iteration ones, so it pays when the
        // iterate the bound predicts (kq) is reached in fewer launches of six than of four, decided launch by launch (12
        // iterations to go: 6 + 6; 10: 6 + 4; 8: 4 + 4).  A replay repeats the sequence of the solve it replays, so that
        // it sums the residuals of the same iterates (Ctx::ChebPlan::kseq).
        bool six = false;
        if (six_ok && spec) {
            six = m <= Ctx::SPEC_MAX_LAUNCHES && plan.kseq[m - 1] == K6;
        } else if (six_ok) {
            const int rem = kq[qs[0]] - k_next + 1;
            const int n4 = (rem + K - 1) / K, n6 = (rem + K6 - 1) / K6;
            six = rem > 0 && 6 * n6 < 5 * n4;
        }
        const int Km = six ? K6 : K;
        if (m <= Ctx::SPEC_MAX_LAUNCHES) kseq[m - 1] = (unsigned char)Km;
        const int k0m = k_next;
        k_next += Km;
        ChebStages<K> cf;
        ChebStages<K6> cf6;
        if (six) for (int s2 = 0; s2 < K6; ++s2) cc.next(cf6.c1[s2], cf6.c2[s2]);
        else for (int s2 = 0; s2 < K; ++s2) cc.next(cf.c1[s2], cf.c2[s2]);
        // lockstep workgroups of four strips: the two-right-hand-side form (two waves per SIMD) and, bit 512, the
        // one-right-hand-side form (154 registers = three waves per SIMD: one workgroup of eight leaves a third of the
        // CU's wave slots empty, three workgroups of four fill them)
        const bool lock4 = wide_ok && mgeo.nstrips >= 16 &&
                           ((wide_in == 4 && na == 2 && NB == 2 && (c->wide_mask & 256)) || (na == 1 && (c->wide_mask & 512) && (c->wide_mask & WIDE_CHEB1)));
        const int wide = lock4 ? 4 : wide_in;
        const dim3 grid(six ? march_grid(mgeo6, 8, false) : march_grid(mgeo, wide, lock4));
        c->probe_begin(probe_id);
        if (na == NB) {
            CVecs<NB> xk, xm;
            Vecs<NB> o1, o2;
            for (int q = 0; q < NB; ++q) {
                xk.p[q] = cur_k[q]; xm.p[q] = cur_m[q];
                o1.p[q] = set[m & 1][0][q]; o2.p[q] = set[m & 1][1][q];
            }
            if (K == 2)
                hipLaunchKernelGGL((k_cheb2<NB>), grid, dim3(256), 0, st, op, mgeo, cf.c1[0], cf.c2[0], cf.c1[1], cf.c2[1],
                                   m == 1 ? 1 : 0, bc, xk, xm, o1, o2, c->partials);
            else if (wide == 8)
                hipLaunchKernelGGL((k_chebk<NB, K, 2, 8>), grid, dim3(512), 0, st, op, mgeo, cf, m == 1 ? 1 : 0, c->nts(), bc, xk, xm, o1, o2,
                                   c->partials);
            else if (lock4)
                hipLaunchKernelGGL((k_chebk<NB, K, 2, 4, true>), grid, dim3(256), 0, st, op, mgeo, cf, m == 1 ? 1 : 0, c->nts(), bc, xk, xm, o1, o2,
                                   c->partials);
            else if (c->cheb_prefetch == 2)
                hipLaunchKernelGGL((k_chebk<NB, K, 2>), grid, dim3(256), 0, st, op, mgeo, cf, m == 1 ? 1 : 0, c->nts(), bc, xk, xm, o1, o2,
                                   c->partials);
            else
                hipLaunchKernelGGL((k_chebk<NB, K, 1>), grid, dim3(256), 0, st, op, mgeo, cf, m == 1 ? 1 : 0, c->nts(), bc, xk, xm, o1, o2,
                                   c->partials);
        } else {
            const int q = qs[0];
            CVecs<1> b1{{b[q]}}, xk{{cur_k[q]}}, xm{{cur_m[q]}};
            Vecs<1> o1{{set[m & 1][0][q]}}, o2{{set[m & 1][1][q]}};
            if (six)
                hipLaunchKernelGGL((k_chebk<1, K6, 2, 8>), grid, dim3(512), 0, st, op, mgeo6, cf6, 0, c->nts(), b1, xk, xm, o1, o2, c->partials);
            else if (K == 2)
                hipLaunchKernelGGL((k_cheb2<1>), grid, dim3(256), 0, st, op, mgeo, cf.c1[0], cf.c2[0], cf.c1[1], cf.c2[1], 0, b1, xk,
                                   xm, o1, o2, c->partials);
            else if (wide == 8)
                hipLaunchKernelGGL((k_chebk<1, K, 2, 8>), grid, dim3(512), 0, st, op, mgeo, cf, 0, c->nts(), b1, xk, xm, o1, o2, c->partials);
            else if (lock4)
                hipLaunchKernelGGL((k_chebk<1, K, 2, 4, true>), grid, dim3(256), 0, st, op, mgeo, cf, 0, c->nts(), b1, xk, xm, o1, o2, c->partials);
            else if (c->cheb_prefetch == 2)
                hipLaunchKernelGGL((k_chebk<1, K, 2>), grid, dim3(256), 0, st, op, mgeo, cf, 0, c->nts(), b1, xk, xm, o1, o2, c->partials);
            else
                hipLaunchKernelGGL((k_chebk<1, K, 1>), grid, dim3(256), 0, st, op, mgeo, cf, 0, c->nts(), b1, xk, xm, o1, o2, c->partials);
        }
        // algorithmic bytes per cell: x_k, x_{k-1}, b read and two iterates written per right-hand side, D, c_i, c_j once
        // (the first launch of a solve has no x_{k-1})
        c->probe_end(probe_id, 8.0 * (3 + (5 - (m == 1 ? 1 : 0)) * na));
        c->kc[K == 4 ? (na == 2 ? KC_CHEB4_2RHS : KC_CHEB4_1RHS) : KC_CHEB2] += 1;
        if (six) c->kc[KC_CHEB6_1RHS] += 1;             // (counted among the one-right-hand-side launches as well)
        CFD_HIP(hipGetLastError());
        int k0 = k0m;                              // this launch measured |r_k0| and |r_{k0+Km-1}|
        if (spec) {
            // the sums of every launch go to slots of their own (this rank's part; one all-reduce over all of them when the
            // solve ends): the closing read sees the whole residual history and can move the schedule either way
            Ctx::SpecCheb &sc = c->spec_cheb;
            const int off = m == 1 ? SPEC_CHEB_OFF : SPEC_CHEB_OFF + 8 + 4 * (m - 2);
            // m = 1: [t] |r_0|^2, [na + t] |r_{K-1}|^2, [2 na + t] |b|^2 (na = NB); later: [t] |r_k0|^2, [na + t] |r_{k0+K-1}|^2
            hipLaunchKernelGGL(k_sum_finish, dim3(1), dim3(256), 0, st, c->partials, (int)grid.x, m == 1 ? 3 * na : 2 * na, c->scal + off);
            CFD_HIP(hipGetLastError());
            sc.launches = m;
            sc.na[m - 1] = na;
            sc.k0[m - 1] = k0m;
            sc.kl[m - 1] = k0m + Km - 1;
            sc.t[m - 1][0] = sc.t[m - 1][1] = -1;
            for (int t = 0; t < na; ++t) sc.t[m - 1][qs[t]] = t;
            for (int t = 0; t < na; ++t) {
                const int q = qs[t], kr = plan.retire_k[q];
                if (kr != k0 && kr != k0 + Km - 1) continue;
                sc.k[q] = kr;
                final_where[q] = kr == k0 ? cur_k[q] : set[m & 1][0][q];
                act[q] = false;
                kdone[q] = kr;
                relres[q] = 0.0;                   // filled in when the iteration's closing read has arrived (spec_verify)
                if (rms) rms[q] = 0.0;
            }
            if (!act[0] && !act[NB - 1]) {
                *iters = std::max(kdone[0], kdone[NB - 1]);
                *used = 1;
                CFD_TRY(all_reduce(c, c->scal + SPEC_CHEB_OFF, 8 + 4 * (m - 1), 0));
                return deliver(true);
            }
            bool overdue = false;                  // (a schedule whose points this launch sequence never visits: not replayable)
            for (int q = 0; q < NB; ++q) overdue = overdue || (act[q] && plan.retire_k[q] < k0 + Km);
            if (overdue) { set_error("Chebyshev replay: schedule point missed"); return CFD_E_STATE; }
            continue;
        }
        // The sums of the first launch (|r_0|, |b|: how many iterations to expect) are looked at one launch later
        // when the previous solve needed a second launch too: the copy is enqueued behind launch 1, the host waits
        // for it only after launch 2 is in the queue, so the GPU does not idle while the host decides.
        if (m == 1 && lag) {
            CFD_TRY(sum_finish(c, c->partials, (int)grid.x, 3 * na, c->scal + CHEB_SAVE_OFF));
            CFD_TRY(read_scalars_begin(c, CHEB_SAVE_OFF, 3 * na));
            first_pending = true;
            continue;
        }
        int kmin = 1 << 30;
        for (int t = 0; t < na; ++t) kmin = std::min(kmin, kq[qs[t]]);
        const int n_pass = first_pending ? 2 : 1;
        for (int pass = 0; pass < n_pass; ++pass) {
          const bool late_first = first_pending;         // pass 0 of two: the data of launch 1, read now
          const int mm = late_first ? 1 : m;
          const int Kp = late_first ? K : Km;            // iterations of the launch whose sums this pass looks at
          if (late_first) k0 = 0; else k0 = k0m;
          if (!late_first) {
              kmin = 1 << 30;
              for (int t = 0; t < na; ++t) if (act[qs[t]]) kmin = std::min(kmin, kq[qs[t]]);
          }
          if (late_first || mm == 1 || k0 + Kp - 1 >= kmin || k0 + Kp - 1 >= klimit) {
            // partial sums of a launch with na right-hand sides: [t] = |r_k0|^2, [na + t] = |r_{k0+K-1}|^2, [2 na + t] = |b|^2
            const int nq = mm == 1 ? 3 * na : 2 * na;
            double got[6];
            if (late_first) {
                CFD_TRY(read_scalars_end(c, CHEB_SAVE_OFF, nq, got));
                first_pending = false;
            } else {
                CFD_TRY(sum_finish(c, c->partials, (int)grid.x, nq, c->scal));
                CFD_TRY(read_scalars(c, 0, nq, got));
            }
            for (int q = 0; q < nq; ++q)
                if (!(got[q] == got[q])) { set_error("Chebyshev: non-finite residual near iteration %d", k0); return CFD_E_NUMERIC; }
            // iterates a retiring right-hand side is taken from: those of the launch whose sums these are
            double *it_k0[2], *it_last[2];
            for (int q = 0; q < NB; ++q) {
                it_k0[q] = late_first ? const_cast<double *>(x0[q]) : cur_k[q];
                it_last[q] = late_first ? set[1][0][q] : set[m & 1][0][q];
            }
            if (mm == 1) {
                bool all_done = true;
                int kmax = 1;
                for (int q = 0; q < NB; ++q) {
                    bb[q] = got[2 * NB + q];
                    if (bb[q] == 0.0 && got[q] > 0.0) return CFD_OK;     // degenerate: leave it to BiCGStab
                    const double need = got[q] > 0.0 ? fmin(1.0, tol * sqrt(bb[q]) / sqrt(got[q])) : 1.0;
                    kq[q] = cc.predict(need);
                    kmax = std::max(kmax, kq[q]);
                    all_done = all_done && conv(q, got[q]);
                }
                if (!all_done && (kmax > c->cheb_max_iters || kmax > maxiter)) return CFD_OK;
                klimit = std::min(maxiter, kmax + 8 + kmax / 2);
            }
            for (int t = 0; t < na; ++t) {
                const int q = qs[t];
                if (!act[q]) continue;               // retired by the late look at launch 1
                if (conv(q, got[t])) CFD_TRY(retire(q, k0, got[t], it_k0[q]));
                else if (conv(q, got[na + t])) CFD_TRY(retire(q, k0 + Kp - 1, got[na + t], it_last[q]));
            }
            if (!act[0] && !act[NB - 1]) {
                *iters = std::max(kdone[0], kdone[NB - 1]);
                *used = 1;
                c->cheb_lag_ok = m >= 2 && !late_first;
                if (op.D == c->ap) {               // the momentum solve's schedule, which a speculative iteration replays
                    c->cheb_plan.valid = true;
                    c->cheb_plan.D = op.D; c->cheb_plan.nb = NB; c->cheb_plan.K = K; c->cheb_plan.tol = tol;
                    c->cheb_plan.retire_k[0] = kdone[0]; c->cheb_plan.retire_k[1] = kdone[NB - 1];
                    for (int l = 0; l < Ctx::SPEC_MAX_LAUNCHES; ++l) c->cheb_plan.kseq[l] = kseq[l];
                }
                return deliver(true);
            }
            if (late_first) continue;                // now the sums of launch 2 itself, if it is time for them
            if (k0 + Kp - 1 >= klimit) {
                // bound not met (spectrum outside the interval): hand the newest iterate to BiCGStab
                CFD_TRY(deliver(false));
                for (int q = 0; q < NB; ++q)
                    if (act[q] && set[m & 1][1][q] != x[q])
                        CFD_HIP(hipMemcpyAsync(x[q], set[m & 1][1][q], g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
                *used = 2;
                *iters = k0 + Kp;
                return CFD_OK;
            }
          }
        }
    }
}

static inline int tail_threads(long cells) {
    const long t = (cells + 63) / 64 * 64;
    return (int)(t < 64 ? 64 : (t > TAIL_NT ? TAIL_NT : t));
}
static int raise_tail_lds_limit(const void *kernel) {
    CFD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS_DOUBLES * (int)sizeof(double)));
    return CFD_OK;
}
// The whole solve as one launch of k_cheb_tail (one rank, at most Ctx::cheb_tail_cells cells).  One host read afterwards
// instead of two or three -- or none in a speculative iteration: status and residual sums wait in a device slot for the
// iteration's closing read (spec_verify); anything but "solved" there repeats the iteration on the host-checked path, which
// takes the fall-backs (BiCGStab) as before.
constexpr int SPEC_CHEBT_OFF = 232;          // Ctx::scal: 8 doubles per solve, up to SPEC_CHEBT_MAX solves per iteration
constexpr int CHEBT_OUT_OFF = 16;            // ... of a host-checked solve
template <int NB>
static int chebyshev_tail(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2],
                          double tol, int maxiter, int *iters, double relres[2], double rms[2], int *used) {
    const Geo g = op.g;
    const long n = (long)g.ni * g.nj;
    ChebTailArgs a{};
    a.ci = op.ci; a.cj = op.cj; a.D = op.D; a.g = g; a.per_i = op.per_i; a.per_j = op.per_j;
    for (int q = 0; q < NB; ++q) { a.b[q] = b[q]; a.x0[q] = x0[q]; a.x[q] = x[q]; }
    a.tol = tol; a.rho_max = c->cheb_rho_max; a.maxiter = maxiter; a.cheb_max_iters = c->cheb_max_iters;
    // (speculative only after a host-checked solve of this kind of system ended "solved": a system that keeps going to
    // BiCGStab would otherwise repeat every iteration)
    const int kind = op.D == c->ap ? 0 : 1;
    const bool spec = c->spec_on && c->spec_chebt.n < Ctx::SPEC_CHEBT_MAX && c->chebt_good[kind];
    a.out = c->scal + (spec ? SPEC_CHEBT_OFF + 8 * c->spec_chebt.n : CHEBT_OUT_OFF);
    const size_t lds_bytes = ((size_t)2 * NB * n + 144) * sizeof(double);
    static bool lds_limit_raised = false;
    if (!lds_limit_raised) {
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_cheb_tail<1, 1>)));
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_cheb_tail<2, 1>)));
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_cheb_tail<1, TAIL_CPT>)));
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_cheb_tail<2, TAIL_CPT>)));
        lds_limit_raised = true;
    }
    if (n <= TAIL_NT) hipLaunchKernelGGL((k_cheb_tail<NB, 1>), dim3(1), dim3(tail_threads(n)), lds_bytes, c->stream, a);
    else hipLaunchKernelGGL((k_cheb_tail<NB, TAIL_CPT>), dim3(1), dim3(TAIL_NT), lds_bytes, c->stream, a);
    CFD_HIP(hipGetLastError());
    c->kc[KC_CHEB_TAIL] += 1;
    if (spec) {
        Ctx::SpecChebT &sp = c->spec_chebt;
        sp.nb[sp.n] = NB;
        sp.momentum[sp.n] = op.D == c->ap;
        sp.n += 1;
        for (int q = 0; q < NB; ++q) { relres[q] = 0.0; if (rms) rms[q] = 0.0; }      // known at the closing read
        *iters = c->stats.mom_iters;
        *used = 1;
        return CFD_OK;
    }
    double h[7];
    CFD_TRY(read_scalars(c, CHEBT_OUT_OFF, 7, h));
    const int status = (int)h[0];
    c->chebt_good[kind] = status == 1;
    if (status == 3) { set_error("momentum-like solve: non-finite right-hand side or residual"); return CFD_E_NUMERIC; }
    *iters = (int)h[1];
    *used = status;              // 0: not used, 1: solved, 2: best iterate in x, BiCGStab continues
    if (status == 1) {
        for (int q = 0; q < NB; ++q) {
            relres[q] = h[4 + q] > 0.0 ? sqrt(h[2 + q] / h[4 + q]) : 0.0;
            if (rms) rms[q] = sqrt(h[2 + q] / global_cells(c));
        }
    }
    return CFD_OK;
}

// returns CFD_OK with *used = 0 when the system is not time-step dominated (caller falls back)
template <int NB>
static int chebyshev_jacobi(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2],
                            double tol, int maxiter, int *iters, double relres[2], double rms[2], int *used) {
    *used = 0;
    if (!x0 || !c->use_chebyshev) return CFD_OK;
    const Geo g = op.g;
    if (c->cheb_tail && c->comm.nranks == 1 && !g.cutlo && !g.cuthi && (long)g.ni * g.nj <= c->cheb_tail_cells)
        return chebyshev_tail<NB>(c, op, b, x, x0, tol, maxiter, iters, relres, rms, used);
    const Tiling T = tiling(g);
    const dim3 grid(T.blocks), blk = block2d();
    hipStream_t st = c->stream;
    const bool fused_path = c->mg_fused && g.ni >= c->fused_min_cheb && g.nj >= c->fused_min_cheb;
    // the bound of the cached laminar momentum operator is remembered with it
    const bool known = fused_path && op.D == c->ap && c->ap_cached && c->rho_cached >= 0.0;
    if (!known) {
        hipLaunchKernelGGL(k_rowsum_bound, grid, blk, 0, st, op, T, c->partials);
        hipLaunchKernelGGL(k_min_finish1, dim3(1), dim3(256), 0, st, c->partials, T.blocks, c->scal + 12);
        CFD_HIP(hipGetLastError());
        CFD_TRY(all_reduce(c, c->scal + 12, 1, 1));
    }
    if (fused_path) {
        if (known) c->h_scal[12] = -c->rho_cached;
        else CFD_TRY(read_scalars(c, 12, 1, c->h_scal + 12));      // h_scal[12] = -rho
        const double rho_f = -c->h_scal[12];
        if (op.D == c->ap && c->ap_cached) c->rho_cached = rho_f;
        if (!(rho_f >= 0.0) || rho_f > c->cheb_rho_max) return CFD_OK;
        // four iterations per launch where no slab cut limits the halo depth to two rows
        if (c->cheb_stages >= 4 && g.ni >= 64 && g.nj >= 64)
            return chebyshev_fused<NB, 4>(c, op, b, x, x0, tol, maxiter, iters, relres, rms, used);
        return chebyshev_fused<NB, 2>(c, op, b, x, x0, tol, maxiter, iters, relres, rms, used);
    }
    // first pass: r_0, |b|^2, x_1 = x_0 + r_0 / D  (theta = 1)
    CVecs<NB> bc, xk, xm;
    Vecs<NB> xo;
    double *bufA[NB], *bufB[NB];
    for (int q = 0; q < NB; ++q) {
        bc.p[q] = b[q];
        // x may alias x0 (k / omega are solved in place): the passes read x_k at the neighbours, so the
        // first iterate then goes to a work plane and the result is copied back at the end
        bufA[q] = (x[q] == x0[q]) ? c->kry[7 + q] : x[q];
        bufB[q] = c->kry[5 + q];
        xk.p[q] = x0[q];
        xm.p[q] = x0[q];
        xo.p[q] = bufA[q];
    }
    c->probe_begin(8);
    hipLaunchKernelGGL((k_cheb<NB>), grid, blk, 0, st, op, T, 0.0, 1.0, 1, bc, xk, xm, xo, c->partials);
    c->probe_end(8, 24.0 + 24.0 * NB);
    c->kc[KC_CHEB1] += 1;
    CFD_HIP(hipGetLastError());
    CFD_TRY(sum_finish(c, c->partials, T.blocks, 2 * NB, c->scal));
    double h[5];
    CFD_HIP(hipMemcpyAsync(c->h_scal, c->scal, 13 * sizeof(double), hipMemcpyDeviceToHost, st));
    CFD_HIP(hipStreamSynchronize(st));
    for (int q = 0; q < 2 * NB; ++q) h[q] = c->h_scal[q];
    const double rho = -c->h_scal[12];
    double rr[2] = {0, 0}, bb[2] = {0, 0};
    for (int q = 0; q < NB; ++q) {
        rr[q] = h[q];
        bb[q] = h[NB + q];
        if (!(rr[q] == rr[q]) || !(bb[q] == bb[q])) { set_error("momentum-like solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
        if (bb[q] == 0.0 && rr[q] > 0.0) return CFD_OK;      // degenerate: leave it to BiCGStab
    }
    if (!(rho >= 0.0) || rho > c->cheb_rho_max) return CFD_OK;
    auto done = [&](const double *r2) {
        for (int q = 0; q < NB; ++q)
            if (sqrt(r2[q]) > tol * sqrt(bb[q])) return false;
        return true;
    };
    auto finish = [&](int k, double *const where[]) -> int {
        for (int q = 0; q < NB; ++q) {
            if (where[q] != x[q])
                CFD_HIP(hipMemcpyAsync(x[q], where[q], g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
            relres[q] = bb[q] > 0.0 ? sqrt(rr[q] / bb[q]) : 0.0;
            if (rms) rms[q] = sqrt(rr[q] / global_cells(c));
        }
        *iters = k;
        *used = 1;
        return CFD_OK;
    };
    if (done(rr)) {
        double *w[2] = {const_cast<double *>(x0[0]), const_cast<double *>(x0[NB - 1])};
        return finish(0, w);
    }
    // iterations the Chebyshev bound asks for: 2 s^k / (1 + s^2k) <= tol |b| / |r_0|
    const double sigma1 = rho > 1e-300 ? 1.0 / rho : 1e300;
    const double s = rho > 1e-300 ? sigma1 - sqrt(sigma1 * sigma1 - 1.0) : 0.0;
    double need = 1.0;
    for (int q = 0; q < NB; ++q)
        if (rr[q] > 0.0) need = fmin(need, tol * sqrt(bb[q]) / sqrt(rr[q]));
    int kpred = 1;
    if (s > 0.0) {
        const double kp = log(need / 2.0) / log(s);
        kpred = kp > 1.0 ? (kp < 1.0e6 ? (int)ceil(kp) : 1000000) : 1;
    }
    if (kpred > c->cheb_max_iters || kpred > maxiter) return CFD_OK;
    // passes k = 1, 2, ...: x_k sits in A for odd k, in B for even k
    double rk = rho;           // rho_{k-1} of Saad's Algorithm 12.1 (rho_0 = delta / theta = rho)
    int k = 1;
    const int klimit = std::min(maxiter, kpred + 8 + kpred / 2);
    for (;; ++k) {
        double **cur = (k & 1) ? bufA : bufB, **oth = (k & 1) ? bufB : bufA;
        const double rnew = 1.0 / (2.0 * sigma1 - rk);
        double c1 = rnew * rk, c2 = 2.0 * rnew / rho;
        if (rho < 1.0e-14) { c1 = 0.0; c2 = 1.0; }          // diagonal matrix: x_1 was already exact
        rk = rnew;
        for (int q = 0; q < NB; ++q) {
            xk.p[q] = cur[q];
            xm.p[q] = (k == 1) ? x0[q] : oth[q];
            xo.p[q] = oth[q];
        }
        CFD_TRY(halo_exchange(c, g, 1, cur[0], NB == 2 ? cur[NB - 1] : nullptr));
        c->probe_begin(8);
        hipLaunchKernelGGL((k_cheb<NB>), grid, blk, 0, st, op, T, c1, c2, 0, bc, xk, xm, xo, c->partials);
        c->probe_end(8, 24.0 + 32.0 * NB);
        c->kc[KC_CHEB1] += 1;
        CFD_HIP(hipGetLastError());
        if (k >= kpred || k >= klimit) {
            // |r_k|^2 of the iterate the pass started from
            CFD_TRY(sum_finish(c, c->partials, T.blocks, NB, c->scal));
            double got[2];
            CFD_TRY(read_scalars(c, 0, NB, got));
            for (int q = 0; q < NB; ++q) {
                rr[q] = got[q];
                if (!(rr[q] == rr[q])) { set_error("Chebyshev: non-finite residual at iteration %d", k); return CFD_E_NUMERIC; }
            }
            if (done(rr)) return finish(k, cur);
            if (k >= klimit) {
                // bound not met (spectrum outside the interval): hand the best iterate to BiCGStab
                for (int q = 0; q < NB; ++q)
                    if (cur[q] != x[q])
                        CFD_HIP(hipMemcpyAsync(x[q], cur[q], g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
                *used = 2;
                *iters = k;
                return CFD_OK;
            }
        }
    }
}

// Relative residuals below this are the rounding level of the fp64 recurrences: a solve that stagnates there
// although a tighter tolerance was asked for (the 1e-13 of the parity runs) counts as converged.
constexpr double ROUNDING_LEVEL_RELRES = 1.0e-12;
constexpr double STALL_ACCEPT_RELRES = 1.0e-6;      // (the reference ends in a direct solve there, whatever its accuracy: solver.py:809-818)

int momentum_like_solve(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2],
                        int nb, double tol, int maxiter, int *iters, double relres[2], double rms[2]) {
    int used = 0, k_cheb_done = 0;
    if (nb == 2) CFD_TRY(chebyshev_jacobi<2>(c, op, b, x, x0, tol, maxiter, &k_cheb_done, relres, rms, &used));
    else CFD_TRY(chebyshev_jacobi<1>(c, op, b, x, x0, tol, maxiter, &k_cheb_done, relres, rms, &used));
    c->stats_last_solver = used == 1 ? 1 : 0;
    if (used == 1) { *iters = k_cheb_done; return CFD_OK; }
    // used == 2: Chebyshev stalled, its last iterate (in x) is the BiCGStab starting point
    const double *xs[2] = {x[0], x[1]};
    const double *const *start = used == 2 ? xs : x0;
    int it = 0;
    int rc = nb == 2 ? bicgstab_jacobi<2>(c, op, b, x, start, tol, maxiter, &it, relres, rms)
                     : bicgstab_jacobi<1>(c, op, b, x, start, tol, maxiter, &it, relres, rms);
    *iters = it + (used == 2 ? k_cheb_done : 0);
    if (rc != CFD_OK) return rc;
    // The reference never returns an unconverged transport solution: bicgstab -> cg -> direct solve
    // (solver.py:809-818).  Here: one restart from the iterate reached with four times the budget, then an error.
    const double accept = tol > ROUNDING_LEVEL_RELRES ? tol : ROUNDING_LEVEL_RELRES;
    auto unconverged = [&]() {
        for (int q = 0; q < nb; ++q)
            if (!(relres[q] <= accept)) return true;
        return false;
    };
    if (it >= maxiter && unconverged()) {
        int it2 = 0;
        rc = nb == 2 ? bicgstab_jacobi<2>(c, op, b, x, xs, tol, 4 * maxiter, &it2, relres, rms)
                     : bicgstab_jacobi<1>(c, op, b, x, xs, tol, 4 * maxiter, &it2, relres, rms);
        *iters += it2;
        if (rc != CFD_OK) return rc;
        // A tolerance below the rounding level asks for "as far as fp64 goes": a solve that two restarts leave between
        // that level and the parity bar of the solved fields has gone there.  Counted, never silent.
        bool stalled = tol < ROUNDING_LEVEL_RELRES;
        for (int q = 0; q < nb; ++q)
            if (!(relres[q] <= STALL_ACCEPT_RELRES)) stalled = false;
        if (stalled && unconverged()) {
            c->stats.transport_stalled += 1;
            return CFD_OK;
        }
        if (unconverged()) {
            set_error("transport solve did not converge: |r|/|b| = %.3e, %.3e after %d iterations (tol %.1e)", relres[0],
                      nb == 2 ? relres[1] : 0.0, *iters, tol);
            return CFD_E_NOCONV;
        }
    }
    return CFD_OK;
}


static inline OpDev level_op(const Ctx *c, const Level &L) {
    return OpDev{L.ci, L.cj, L.D, L.g, c->pp_per_i, c->pp_per_j};
}
// operator of a fused V-cycle level: with the single-precision face planes where the level has them
static inline OpDev level_op_mg(const Ctx *c, const Level &L) {
    OpDev op = level_op(c, L);
    if (c->mg_f32 && L.cif) { op.cif = L.cif; op.cjf = L.cjf; }
    return op;
}
// face planes of a coarse level rounded in place to single-precision values: whichever kernel cycles the level (fused
// marching kernels on the copies below, the LDS tail, the one-sweep kernels, the line smoother) then sees one operator,
// and the V-cycle does not depend on where the tail starts or on the slab decomposition
__global__ void k_round_faces_f32(double *ci, double *cj, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        ci[q] = (double)(float)ci[q];
        cj[q] = (double)(float)cj[q];
    }
}
__global__ void k_faces_to_f32(const double *ci, const double *cj, float *cif, float *cjf, size_t n) {
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (size_t)gridDim.x * blockDim.x) {
        cif[q] = (float)ci[q];
        cjf[q] = (float)cj[q];
    }
}
static inline void factors(const Geo &f, const Geo &co, int &fi, int &fj) {
    fi = (co.ni == f.ni) ? 1 : 2;
    fj = (co.nj == f.nj) ? 1 : 2;
}
// geometry the level above sees of level C: this rank's slab of it when C is the first replicated level
static inline Geo coarse_view(const Ctx *c, int lc) {
    return (c->l_agg > 0 && lc == c->l_agg) ? c->levels[lc].gs : c->levels[lc].g;
}

// asymmetry probe: partial[0] = max |ci[0,j] - ci[Ni,j]| etc. (sum of abs differences is enough)
__global__ void k_seam_asym(Geo g, const double *ci, const double *cj, int per_i, int per_j, double *out) {
    __shared__ double sa[256], sb[256];
    double a = 0.0, b = 0.0;
    if (per_i)
        for (int j = threadIdx.x; j < g.nj; j += 256) {
            const size_t c = g.cell(0, j);
            a += fabs(ci[c] - ci[c + g.ni]);
            b += fabs(ci[c]);
        }
    if (per_j)
        for (int i = threadIdx.x; i < g.ni; i += 256) {
            const size_t c = g.cell(i, 0);
            a += fabs(cj[c] - cj[c + (size_t)g.nj * g.pitch]);
            b += fabs(cj[c]);
        }
    sa[threadIdx.x] = a; sb[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sa[threadIdx.x] += sa[threadIdx.x + s]; sb[threadIdx.x] += sb[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sa[0]; out[1] = sb[0]; }
}

#include "semi.inc"

int build_pressure_hierarchy(Ctx *c) {
    c->p_iters_hint = 0;
    Level &L0 = c->levels[0];
    CFD_TRY(launch_pressure_coeffs(c, L0));
    hipStream_t st = c->stream;
    // symmetric unless a periodic seam carries different coefficients on its two copies.  Three regimes:
    // copies equal to 1e-13: PCG, and the singular (nullspace) form is allowed; copies that differ at the rounding
    // level of the metrics (a mesh whose spacing is not exactly representable: 1/160 gives 1e-12 .. 1e-11 in a_P):
    // still PCG -- its residual recurrence uses the operator as stored, so it converges to the same solution -- but
    // with the reference cell eliminated, because the left null vector of such an operator is not exactly constant
    // and the singular system would be inconsistent at that level; anything larger: BiCGStab.
    c->p_symmetric = true;
    bool seam_exact = true;
    if (c->pp_per_i || c->pp_per_j) {
        hipLaunchKernelGGL(k_seam_asym, dim3(1), dim3(256), 0, st, c->g, L0.ci, L0.cj, c->pp_per_i, c->pp_per_j,
                           c->scal + 8);
        CFD_HIP(hipGetLastError());
        double a[2];
        CFD_TRY(all_reduce(c, c->scal + 8, 2, 0));
        CFD_TRY(read_scalars(c, 8, 2, a));
        c->p_symmetric = !(a[0] > c->p_sym_tol * a[1]);
        c->p_seam_asym = a[1] > 0.0 ? a[0] / a[1] : 0.0;
        seam_exact = !(a[0] > 1.0e-13 * a[1]);
    }
    c->p_ns_active = !c->has_dirichlet && c->p_symmetric && seam_exact && c->p_nullspace;
    if (!c->p_ns_active) CFD_TRY(launch_pin_operator(c, L0));
    // finest-level diagonal from the face planes (see diag_from_faces): every rank needs the four
    // couplings of the detached reference cell, which only its owner knows
    c->p_diag = DiagFix{};
    if (c->recompute_diag) {
        DiagFix f{};
        f.on = 1; f.pi = -1; f.pj = -1;
        if (!c->has_dirichlet && !c->p_ns_active) {
            if (c->pin_i < 0) CFD_HIP(hipMemsetAsync(c->scal + 64, 0, 8 * sizeof(double), st));
            CFD_TRY(all_reduce(c, c->scal + 64, 8, 0));
            double pc[8];
            CFD_TRY(read_scalars(c, 64, 8, pc));
            f.pi = c->pin_i_any; f.pj = c->pin_j_rel;
            f.cW = pc[0]; f.cE = pc[1]; f.cS = pc[2]; f.cN = pc[3];
            f.oW = pc[4]; f.oE = pc[5]; f.oS = pc[6]; f.oN = pc[7];
        }
        c->p_diag = f;
    }
    const int la = c->l_agg;
    // the fused kernels read K halo rows of the operator, and the diagonal recomputed from the faces needs the
    // north face of the outermost of them: all NG rows
    CFD_TRY(halo_exchange(c, L0.g, NG, L0.ci, L0.cj, L0.D));
    for (int l = 0; l + 1 < (int)c->levels.size(); ++l) {
        Level &F = c->levels[l], &C = c->levels[l + 1];
        const bool to_agg = la > 0 && l + 1 == la;
        const Geo cg = coarse_view(c, l + 1);
        double *ci_t = to_agg ? C.s0 : C.ci, *cj_t = to_agg ? C.s1 : C.cj, *D_t = to_agg ? C.s2 : C.D;
        int fi, fj;
        factors(F.g, cg, fi, fj);
        const Tiling Tc = tiling(cg);
        // C.t is used as the scratch for the diagonal extra
        hipLaunchKernelGGL(k_mg_coarsen, dim3(Tc.blocks), block2d(), 0, st, F.g, F.ci, F.cj, F.D, cg, Tc, fi, fj,
                           c->pp_per_i, c->pp_per_j, c->mg_odd_fix, ci_t, cj_t, C.t);
        if (c->mg_f32) {
            // before the diagonal is summed from them: D stays the exact sum of the faces every kernel reads
            const size_t n = cg.plane_elems();
            hipLaunchKernelGGL(k_round_faces_f32, dim3((int)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, ci_t, cj_t, n);
        }
        hipLaunchKernelGGL(k_mg_coarse_diag, dim3(Tc.blocks), block2d(), 0, st, cg, Tc, ci_t, cj_t, C.t, D_t);
        CFD_HIP(hipGetLastError());
        if (to_agg) {
            // replicate the level: every rank gathers all slabs, the top boundary face row of c_j
            // comes from the last rank (gathered through the scratch plane)
            const size_t P = cg.pitch, rows = (size_t)cg.nj * P;
            CFD_TRY(all_gather_rows(c, ci_t + NG * P, C.ci + NG * P, rows));
            CFD_TRY(all_gather_rows(c, cj_t + NG * P, C.cj + NG * P, rows));
            CFD_TRY(all_gather_rows(c, D_t + NG * P, C.D + NG * P, rows));
            CFD_TRY(all_gather_rows(c, cj_t + (size_t)(NG + cg.nj) * P, C.t + NG * P, P));
            CFD_HIP(hipMemcpyAsync(C.cj + (size_t)(NG + C.g.nj) * P, C.t + (size_t)(NG + c->comm.nranks - 1) * P,
                                   P * sizeof(double), hipMemcpyDeviceToDevice, st));
        } else if (l + 1 < la) {
            CFD_TRY(halo_exchange(c, C.g, NG, C.ci, C.cj, C.D));
        }
    }
    // ---- which levels get the alternating line smoother?  (strong / weak coupling ratio of the level's operator)
    {
        const int nl = (int)c->levels.size();
        size_t need = 0;
        for (int l = 0; l < nl; ++l) {
            Level &L = c->levels[l];
            L.line = false;
            L.aniso = 1.0;
            if (l == nl - 1 && nl > 1) continue;                 // the coarsest level is solved, not smoothed
            const OpDev op = level_op(c, L);
            const Tiling T = tiling(L.g);
            hipLaunchKernelGGL(k_level_aniso, dim3(T.blocks), block2d(), 0, st, op, T, c->partials);
            hipLaunchKernelGGL(k_min_finish1, dim3(1), dim3(256), 0, st, c->partials, T.blocks, c->scal + 72);
            hipLaunchKernelGGL(k_min_finish1, dim3(1), dim3(256), 0, st, c->partials + MAX_PARTIALS, T.blocks, c->scal + 73);
            CFD_HIP(hipGetLastError());
            if (l < la) CFD_TRY(all_reduce(c, c->scal + 72, 2, 1));
            double a[2];
            CFD_TRY(read_scalars(c, 72, 2, a));
            L.aniso = fmax(-a[0], -a[1]);
            const bool big_enough = L.g.ni >= 4 && L.g.nj >= 4;
            // directions: mg_line_dirs = 3: both wherever a level is line-smoothed; 0: each direction by its own ratio
            // (a wall-graded channel is strongly coupled along j everywhere and never along i)
            L.line_dirs = 0;
            if (big_enough && c->mg_line > 0) L.line_dirs = 3;
            else if (big_enough && c->mg_line < 0 && L.aniso > c->mg_line_threshold) {
                L.line_dirs = c->mg_line_dirs ? 3 : ((-a[0] > c->mg_line_threshold ? 1 : 0) | (-a[1] > c->mg_line_threshold ? 2 : 0));
            }
            L.line = L.line_dirs != 0;
            if (L.line_dirs & 1) need += 2 * L.g.plane_elems() * sizeof(double);
            if (L.line_dirs & 2) need += 2 * L.g.plane_elems() * sizeof(double);
        }
        // stretched mesh: the semi-coarsened line multigrid takes the whole cycle where it can (one rank, frozen operator,
        // lines of at most SEMI_MAX_NI cells); the levels keep their flags, their block factors are then not needed
        CFD_TRY(semi_build(c, c->levels[0].line));
        if (c->semi_on) need = 0;
        if (need > c->line_mem_bytes) {
            if (c->line_mem) CFD_HIP(hipFree(c->line_mem));
            c->line_mem = nullptr;
            c->line_mem_bytes = 0;
            CFD_HIP(hipMalloc((void **)&c->line_mem, need));
            c->line_mem_bytes = need;
        }
        size_t off = 0;
        int n_line = 0;
        for (int l = 0; l < nl; ++l) {
            Level &L = c->levels[l];
            if (!L.line) continue;
            ++n_line;
            if (c->semi_on) continue;
            const size_t pb = L.g.plane_elems() * sizeof(double);
            const LineGeo lg = line_geo(L.g);
            if (L.line_dirs & 1) {
                L.lm = (double *)(c->line_mem + off);
                L.linv = (double *)(c->line_mem + off + pb);
                off += 2 * pb;
                const long n = (long)L.g.ni * lg.nsegs;
                const int nb = (int)std::min<long>((n + 255) / 256, 2048);
                hipLaunchKernelGGL(k_line_factor_j, dim3(nb), dim3(256), 0, st, level_op(c, L), lg, L.lm, L.linv);
            }
            if (L.line_dirs & 2) {
                L.lmi = (double *)(c->line_mem + off);
                L.linvi = (double *)(c->line_mem + off + pb);
                off += 2 * pb;
                const long n = (long)L.g.nj * lg.nstrips;
                const int nb = (int)std::min<long>((n + 255) / 256, 2048);
                hipLaunchKernelGGL(k_line_factor_i, dim3(nb), dim3(256), 0, st, level_op(c, L), lg, L.lmi, L.linvi);
            }
            CFD_HIP(hipGetLastError());
        }
        if (n_line > c->kc[KC_MG_LINE_LEVELS_MAX]) c->kc[KC_MG_LINE_LEVELS_MAX] = n_line;
    }
    // ---- V-cycle tail: the first level from which everything below fits the LDS of one workgroup (replicated
    // levels only on j-slabs; point-smoothed levels only)
    c->tail_lt = -1;
    {
        const int nl = (int)c->levels.size();
        int last_line = -1;
        for (int l = 0; l < nl; ++l)
            if (c->levels[l].line) last_line = l;
        if (c->mg_tail && nl > 1) {
            // From level 0 on a mesh too small for the fused marching kernels (a side below 16 cells: the reference's own
            // cases are 5 x 20 ... 15 x 40): the whole V-cycle is then ONE launch instead of eight kernels of the
            // one-sweep-per-launch form around the tail.  tail_from = 1 keeps the finest level out.
            const int lt0 = (c->tail_from == 0 && !level_fusable(c, 0, c->levels[0].g, coarse_view(c, 1))) ? 0 : 1;
            for (int lt = std::max(lt0, std::max(la, last_line + 1)); lt < nl; ++lt) {
                // the first tail level has up to TAIL_CPT cells per thread, the second one, the others share one cached cell per thread
                long total = 0, ncache = 0, second = 0;
                for (int l = lt; l < nl; ++l) {
                    const long cells = (long)c->levels[l].g.ni * c->levels[l].g.nj;
                    total += cells;
                    if (l == lt + 1) second = cells;
                    if (l >= lt + 2) ncache += cells;
                }
                const long top = (long)c->levels[lt].g.ni * c->levels[lt].g.nj;
                if (top > TAIL_MAX_CELLS || second > TAIL_NT || ncache > TAIL_NT || nl - lt > TAIL_MAX_LEVELS ||
                    tail_lds_doubles(total, top, ncache) > TAIL_LDS_DOUBLES)
                    continue;
                TailArgs &a = c->tail;
                a = TailArgs{};
                a.n = nl - lt;
                a.total = (int)total;
                a.ncache = (int)ncache;
                int off = 0, coff = 0;
                for (int l = lt; l < nl; ++l) {
                    const Level &L = c->levels[l];
                    TailLevel &t = a.L[l - lt];
                    t.ci = L.ci; t.cj = L.cj; t.D = L.D; t.g = L.g; t.off = off;
                    t.fi = t.fj = 1;
                    if (l + 1 < nl) factors(L.g, c->levels[l + 1].g, t.fi, t.fj);
                    off += L.g.ni * L.g.nj;
                    t.coff = coff;
                    if (l >= lt + 2) coff += L.g.ni * L.g.nj;
                }
                a.per_i = c->pp_per_i; a.per_j = c->pp_per_j;
                if ((size_t)a.total * 64 > TAIL_TABLE_BYTES) { set_error("internal: V-cycle tail table exceeds its share of the workspace"); return CFD_E_STATE; }
                a.tab = c->tail_tab;
                hipLaunchKernelGGL(k_tail_pack, dim3((a.total + 255) / 256), dim3(256), 0, st, a, c->tail_tab);
                CFD_HIP(hipGetLastError());
                c->tail_lt = lt;
                break;
            }
        }
    }
    // ---- the whole solve in one workgroup (k_pcg_tail): the hierarchy from level 0 with the Krylov planes beside it
    c->ptail_ok = false;
    {
        const int nl = (int)c->levels.size();
        bool plain = c->pcg_tail && c->comm.nranks == 1 && !c->semi_on && nl <= TAIL_MAX_LEVELS;
        for (int l = 0; l < nl; ++l) plain = plain && !c->levels[l].line;
        long total = 0, ncache = 0, second = 0;
        for (int l = 0; l < nl; ++l) {
            const long cells = (long)c->levels[l].g.ni * c->levels[l].g.nj;
            total += cells;
            if (l == 1) second = cells;
            if (l >= 2) ncache += cells;
        }
        const long top = (long)c->levels[0].g.ni * c->levels[0].g.nj;
        if (plain && top <= c->pcg_tail_cells && top <= TAIL_MAX_CELLS && second <= TAIL_NT && ncache <= TAIL_NT &&
            pcg_tail_lds_doubles(total, top, ncache) <= TAIL_LDS_DOUBLES && (size_t)total * 64 <= TAIL_TABLE_BYTES) {
            TailArgs &a = c->ptail;
            a = TailArgs{};
            a.n = nl;
            a.total = (int)total;
            a.ncache = (int)ncache;
            int off = 0, coff = 0;
            for (int l = 0; l < nl; ++l) {
                const Level &L = c->levels[l];
                TailLevel &t = a.L[l];
                t.ci = L.ci; t.cj = L.cj; t.D = L.D; t.g = L.g; t.off = off;
                t.fi = t.fj = 1;
                if (l + 1 < nl) factors(L.g, c->levels[l + 1].g, t.fi, t.fj);
                off += L.g.ni * L.g.nj;
                t.coff = coff;
                if (l >= 2) coff += L.g.ni * L.g.nj;
            }
            a.per_i = c->pp_per_i; a.per_j = c->pp_per_j;
            double *tab = c->tail_tab + TAIL_TABLE_BYTES / sizeof(double);       // (second half: the V-cycle tail has the first)
            a.tab = tab;
            hipLaunchKernelGGL(k_tail_pack, dim3((a.total + 255) / 256), dim3(256), 0, st, a, tab);
            CFD_HIP(hipGetLastError());
            c->ptail_ok = true;
        }
    }
    // ---- single-precision face planes of the levels the fused marching kernels cycle.  The V-cycle built on them is
    // still one fixed symmetric operator (faces are stored once per pair of cells, the diagonal of the nullspace form is
    // the sum of the same rounded faces), which is all PCG asks of its preconditioner: the solve converges to the
    // solution of the double-precision operator as before, and each fused pass reads 8 B per cell less.
    {
        const int nl = (int)c->levels.size();
        const int n_top = c->tail_lt >= 0 ? c->tail_lt : nl - 1;
        size_t need = 0;
        std::vector<char> want(nl, 0);
        for (int l = 0; l < nl; ++l) {
            c->levels[l].cif = c->levels[l].cjf = nullptr;
            // (every point-smoothed level above the tail: whether a level is cycled by the fused kernels can change with
            // cfd_set_multigrid, which does not rebuild the hierarchy)
            if (!c->mg_f32 || l >= n_top || c->levels[l].line) continue;
            want[l] = 1;
            need += 2 * ((c->levels[l].g.plane_elems() * sizeof(float) + 255) / 256 * 256);
        }
        // (carved from the context's workspace by cfd_create for every level, whether or not it ends up fused)
        if (need > c->f32_mem_bytes) { set_error("internal: single-precision face planes exceed their share of the workspace"); return CFD_E_STATE; }
        size_t off = 0;
        for (int l = 0; l < nl; ++l) {
            if (!want[l]) continue;
            Level &L = c->levels[l];
            const size_t n = L.g.plane_elems(), pb = (n * sizeof(float) + 255) / 256 * 256;
            L.cif = (float *)(c->f32_mem + off);
            L.cjf = (float *)(c->f32_mem + off + pb);
            off += 2 * pb;
            const int nb = (int)std::min<size_t>((n + 255) / 256, 4096);
            hipLaunchKernelGGL(k_faces_to_f32, dim3(nb), dim3(256), 0, st, L.ci, L.cj, L.cif, L.cjf, n);
            CFD_HIP(hipGetLastError());
        }
    }
    c->p_built = true;
    return CFD_OK;
}

// One pass of the alternating line smoother on level L: x_in (nullptr: zero) -> result; the planes ping-pong
// between cur and oth, the result ends in *cur.  pre: j-lines, i-lines, point; post: point, i-lines, j-lines.
static int line_smooth(Ctx *c, const Level &L, const OpDev &op, const double *rhs, double **cur, double **oth, bool post,
                       bool zero_guess, bool distributed, bool skip_point = false) {
    hipStream_t st = c->stream;
    const LineGeo lg = line_geo(L.g);
    const dim3 grid(line_blocks(lg)), gridi(line_blocks(lg, true));
    const Tiling T = tiling(L.g);
    const double om = c->mg_line_omega;
    bool zero = zero_guess;
    auto sweep = [&](int kind) -> int {         // 0: j-lines, 1: i-lines, 2: point
        if (kind == 2 && skip_point) return CFD_OK;      // the caller's fused transfer kernel does the point sweep
        if (kind == 0 && !(L.line_dirs & 1)) return CFD_OK;
        if (kind == 1 && !(L.line_dirs & 2)) return CFD_OK;
        if (!zero && distributed) CFD_TRY(halo_exchange(c, L.g, 1, *cur));
        if (kind == 0) {
            if (zero) hipLaunchKernelGGL((k_line_j<true>), grid, dim3(256), 0, st, op, lg, L.lm, L.linv, rhs, (const double *)nullptr, om, *oth);
            else hipLaunchKernelGGL((k_line_j<false>), grid, dim3(256), 0, st, op, lg, L.lm, L.linv, rhs, (const double *)*cur, om, *oth);
        } else if (kind == 1) {
            if (zero) hipLaunchKernelGGL((k_line_i<true>), gridi, dim3(256), 0, st, op, lg, L.lmi, L.linvi, rhs, (const double *)nullptr, om, *oth);
            else hipLaunchKernelGGL((k_line_i<false>), gridi, dim3(256), 0, st, op, lg, L.lmi, L.linvi, rhs, (const double *)*cur, om, *oth);
        } else {
            if (zero) hipLaunchKernelGGL(k_mg_first, dim3(T.blocks), block2d(), 0, st, L.g, T, L.D, rhs, om, *oth);
            else hipLaunchKernelGGL(k_mg_jacobi, dim3(T.blocks), block2d(), 0, st, op, T, rhs, (const double *)*cur, om, *oth);
        }
        CFD_HIP(hipGetLastError());
        std::swap(*cur, *oth);
        zero = false;
        return CFD_OK;
    };
    if (!post) { CFD_TRY(sweep(0)); CFD_TRY(sweep(1)); CFD_TRY(sweep(2)); }
    else { CFD_TRY(sweep(2)); CFD_TRY(sweep(1)); CFD_TRY(sweep(0)); }
    return CFD_OK;
}

// independent-wave (256 threads) or lockstep (512 threads, MarchItems<8>) form of a marching kernel
#define LAUNCH_WIDE(wide, K4, K8, grid, ...)                                             \
    do {                                                                                 \
        if ((wide) == 8) hipLaunchKernelGGL(K8, grid, dim3(512), 0, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL(K4, grid, dim3(256), 0, st, __VA_ARGS__);               \
    } while (0)

// same, with the single-precision face planes where the level has them (K4F / K8F)
#define LAUNCH_WIDE_F(f32, wide, K4, K8, K4F, K8F, grid, ...)                            \
    do {                                                                                 \
        if (f32) LAUNCH_WIDE(wide, K4F, K8F, grid, __VA_ARGS__);                         \
        else LAUNCH_WIDE(wide, K4, K8, grid, __VA_ARGS__);                               \
    } while (0)

// line-smoothed level whose point sweep is fused with the grid transfers (k_mg_down_x, k_mg_up_k<false, 1>)
static inline bool line_transfer_fusable(const Ctx *c, int lvl, const Geo &F, const Geo &C) {
    return c->mg_fused && c->levels[lvl].line && lvl >= c->l_agg && !F.cutlo && !F.cuthi && F.ni >= 16 && F.nj >= 16 &&
           C.ni == (F.ni + 1) / 2 && C.nj == (F.nj + 1) / 2;
}
static inline MarchGeo march_geo_w(const Geo &g, int W) {
    MarchGeo m = march_geo(g);
    m.nstrips = (g.ni + W - 1) / W;
    return m;
}

// One V-cycle: z ~= A^-1 r starting from zero.  The result lands in *zout, which is
// either z0 or z1 (Jacobi ping-pong).
// a tail as built with the hierarchy + what cfd_set_multigrid may have changed since
static TailDev tail_launch_args(const Ctx *c, const TailArgs &built) {
    TailDev a{};
    a.n = built.n; a.total = built.total; a.ncache = built.ncache; a.tab = built.tab;
    a.g0 = built.L[0].g;
    for (int t = 0; t < built.n; ++t) a.L[t] = TailDevLevel{built.L[t].g.ni, built.L[t].g.nj, built.L[t].off, built.L[t].coff};
    a.per_i = c->pp_per_i; a.per_j = c->pp_per_j;
    a.pre = c->mg_pre < 1 ? 1 : c->mg_pre; a.post = c->mg_post; a.coarse_sweeps = c->mg_coarse_sweeps < 1 ? 1 : c->mg_coarse_sweeps;
    a.nom = c->mg_nom < 4 ? c->mg_nom : 4;
    for (int q = 0; q < a.nom; ++q) a.om[q] = c->om_at(q);
    a.om_coarse = c->om_mean(); a.gamma = c->mg_gamma;
    return a;
}

static int vcycle_impl(Ctx *c, const double *r, double *z0, double *z1, double **zout, int *dot_blocks,
                       const ResidFuse *rf = nullptr, bool *z_f32 = nullptr) {
    if (z_f32) *z_f32 = false;
    if (c->semi_on) {
        if (rf && rf->rr_partials) { set_error("internal: (r, r) by-product asked of a semi-coarsened hierarchy"); return CFD_E_STATE; }
        const bool z32 = z_f32 && c->mg_f32 && c->semi_z32;      // the caller takes z as a float plane (the PCG direction step does)
        CFD_TRY(semi_vcycle(c, r, z0, z1 != z0 ? z1 : nullptr, z32, dot_blocks, rf));
        if (z32) *z_f32 = true;
        *zout = z0;
        return CFD_OK;
    }
    hipStream_t st = c->stream;
    const int nl = (int)c->levels.size();
    // sweep s of a smoothing pass uses omega (even s) or omega2 (odd s): with omega2 != omega the two
    // sweeps form the degree-2 Chebyshev polynomial smoother on the upper part of the Jacobi spectrum
    const double om = c->om_at(0), om2 = c->om_at(1);
    const int pre = c->mg_pre < 1 ? 1 : c->mg_pre, post = c->mg_post;
    const int KS = pre;                          // sweeps fused per pass where level_fusable()
    SweepOm<3> om3{{c->om_at(0), c->om_at(1), c->om_at(2)}};
    std::vector<double *> cur(nl), oth(nl);
    std::vector<const double *> rhs(nl);
    for (int l = 0; l < nl; ++l) {
        cur[l] = (l == 0) ? z0 : c->levels[l].x;
        oth[l] = (l == 0) ? z1 : c->levels[l].t;
        rhs[l] = (l == 0) ? r : c->levels[l].b;
    }
    const int la = c->l_agg;   // levels [0, la) are j-slabs with halo exchange, the rest is replicated
    // Without any Dirichlet pressure and in the nullspace form, every level's diagonal is exactly the sum of
    // its four face planes (the coarsening adds D - sum(faces) = 0), so the coarse levels need not read D either
    DiagFix coarse_df{};
    if (c->p_ns_active && c->recompute_diag) { coarse_df.on = 1; coarse_df.pi = -1; coarse_df.pj = -1; }
    // j-slabs: the right-hand side of the next level and the pre-smoothed x of this one (which the post-smoothing
    // pass reads) cross the cuts in ONE exchange right after the kernel that produced both
    std::vector<char> rhs_sent(nl, 0), x_sent(nl, 0);
    c->kc[KC_VCYCLE] += 1;
    int n_fused_levels = 0;
    const int lt = c->tail_lt;                   // levels [lt, nl) are cycled by k_mg_tail in one launch
    const int n_top = lt >= 0 ? lt : nl - 1;     // levels [0, n_top) run their own kernels
    // the first coarse level keeps b and x in single precision (mg_fetch: VT) where the two finest levels both run the
    // three-sweep kernels with single-precision faces
    // (j-slabs: both levels slab-distributed, so that no gather touches a float plane; at a cut the float planes travel as
    // planes of half the pitch, like the finest level's x)
    // Only levels with MORE than mg_c32_min_cells cells of the GLOBAL level (default 2^18 = the size up to which a slab run
    // replicates a level, far above what a tail takes): the choice is then the same on one rank and on slabs, whose tails
    // start at different levels on small meshes, and the slab V-cycle stays the single-rank V-cycle.
    // nf32 = number of coarse levels (1 .. nf32) with float vectors: at most two (the second one is 1 / 16 of the finest).
    int nf32 = 0;
    if (c->mg_c32 && c->mg_f32 && !c->mg_v32_emul && KS == 3 && level_fusable(c, 0, c->levels[0].g, coarse_view(c, 1)) &&
        level_op_mg(c, c->levels[0]).cif != nullptr) {
        for (int l = 1; l <= c->mg_c32 && l < n_top && l + 1 < nl; ++l) {
            const long cells = (long)c->levels[l].g.ni * c->levels[l].g.nj * (l < la ? c->comm.nranks : 1);
            if (cells <= c->mg_c32_min_cells || !(c->comm.nranks == 1 || l < la) || !level_fusable(c, l, c->levels[l].g, coarse_view(c, l + 1)) ||
                level_op_mg(c, c->levels[l]).cif == nullptr)
                break;
            nf32 = l;
        }
    }
    const bool c32 = nf32 >= 1;
    if (nf32 > c->kc[KC_MG_FLOAT_LEVELS_MAX]) c->kc[KC_MG_FLOAT_LEVELS_MAX] = nf32;
    auto own32 = [&](int l) { return l >= 1 && l <= nf32; };        // level l keeps b and x as float
    for (int l = 0; l < n_top; ++l) {
        const Level &L = c->levels[l];
        const Level &C = c->levels[l + 1];
        const OpDev op = level_op(c, L), opm = level_op_mg(c, L);
        const bool to_agg = la > 0 && l + 1 == la;
        const Geo cg = coarse_view(c, l + 1);
        double *cb = to_agg ? C.s0 : C.b;
        if (level_fusable(c, l, L.g, cg)) {
            MarchGeo mgeo = march_geo_k(L.g, KS, c->march_ls_model != 0);
            mgeo.xcd = KS == 3 ? c->xcd_map : 0;
            const int wide = KS == 3 ? march_wide(c, mgeo, l > 0 ? WIDE_MG_COARSE : (rf ? WIDE_MG_DOWN_RF : WIDE_MG_DOWN)) : 4;
            const dim3 grid(march_grid(mgeo, wide));
            if (++n_fused_levels > c->kc[KC_MG_FUSED_LEVELS_MAX]) c->kc[KC_MG_FUSED_LEVELS_MAX] = n_fused_levels;
            if (l == 0 && rf) CFD_TRY(halo_exchange(c, L.g, KS, const_cast<double *>(rf->q)));      // r_old's rows are there already
            else if (!rhs_sent[l]) {
                Geo gb = L.g;
                if (own32(l)) gb.pitch /= 2;                   // float right-hand side of a coarse level
                CFD_TRY(halo_exchange(c, gb, KS, const_cast<double *>(rhs[l])));
            }
            if (l == 0 && rf) {
                // rhs[0] is r_old here; the kernel leaves r = r_old - alpha q in rf->rnew, which the way up reads
                if (KS != 3) { set_error("internal: fused residual update needs three sweeps per pass"); return CFD_E_STATE; }
                c->probe_begin(10);
                if (c32) LAUNCH_WIDE(wide, (k_mg_down_k<true, 3, true, 4, true, 1>), (k_mg_down_k<true, 3, true, 8, true, 1>), grid, opm, mgeo, c->p_diag, rhs[l], om3, cur[l], cg, cb, *rf, c->nts(L.g));
                else LAUNCH_WIDE_F(opm.cif != nullptr, wide, (k_mg_down_k<true, 3, true>), (k_mg_down_k<true, 3, true, 8>), (k_mg_down_k<true, 3, true, 4, true>), (k_mg_down_k<true, 3, true, 8, true>), grid, opm, mgeo, c->p_diag, rhs[l], om3, cur[l], cg, cb, *rf, c->nts(L.g) | (c->mg_v32_emul ? 2 : 0));
                c->probe_end(10, (opm.cif ? 38.0 : 50.0) - (c32 ? 1.0 : 0.0));      // single precision: faces 8 B instead of 16, x 4 B instead of 8 (coarse right-hand side: 1 instead of 2)
                c->kc[KC_MG_DEFERRED_RESID] += 1;
                rhs[0] = rf->rnew;
                if (rf->rr_partials) {
                    CFD_HIP(hipGetLastError());
                    CFD_TRY(sum_finish(c, rf->rr_partials, (int)grid.x, 1, c->scal + rf->rr_slot));
                }
            } else if (l == 0) {
                c->probe_begin(1);
                if (c32)
                    LAUNCH_WIDE(wide, (k_mg_down_k<true, 3, false, 4, true, 1>), (k_mg_down_k<true, 3, false, 8, true, 1>), grid, opm, mgeo, c->p_diag, rhs[l], om3, cur[l], cg, cb, ResidFuse{}, c->nts(L.g));
                else if (KS == 3)
                    LAUNCH_WIDE_F(opm.cif != nullptr, wide, (k_mg_down_k<true, 3>), (k_mg_down_k<true, 3, false, 8>), (k_mg_down_k<true, 3, false, 4, true>), (k_mg_down_k<true, 3, false, 8, true>), grid, opm, mgeo, c->p_diag, rhs[l], om3, cur[l], cg, cb, ResidFuse{}, c->nts(L.g) | (c->mg_v32_emul ? 2 : 0));
                else
                    hipLaunchKernelGGL((k_mg_down<true>), grid, dim3(256), 0, st, op, mgeo, c->p_diag, rhs[l], om, om2, cur[l], cg, cb);
                c->probe_end(1, (c->p_diag.on ? 34.0 : 42.0) - (opm.cif ? 12.0 : 0.0) - (c32 ? 1.0 : 0.0));
            } else if (own32(l) && own32(l + 1)) {
                LAUNCH_WIDE(wide, (k_mg_down_k<false, 3, false, 4, true, 3>), (k_mg_down_k<false, 3, false, 8, true, 3>), grid, opm, mgeo, coarse_df, rhs[l], om3, cur[l], cg, cb, ResidFuse{}, c->nts(L.g));
            } else if (own32(l)) {
                LAUNCH_WIDE(wide, (k_mg_down_k<false, 3, false, 4, true, 2>), (k_mg_down_k<false, 3, false, 8, true, 2>), grid, opm, mgeo, coarse_df, rhs[l], om3, cur[l], cg, cb, ResidFuse{}, c->nts(L.g));
            } else if (KS == 3) {
                LAUNCH_WIDE_F(opm.cif != nullptr, wide, (k_mg_down_k<false, 3>), (k_mg_down_k<false, 3, false, 8>), (k_mg_down_k<false, 3, false, 4, true>), (k_mg_down_k<false, 3, false, 8, true>), grid, opm, mgeo, coarse_df, rhs[l], om3, cur[l], cg, cb, ResidFuse{}, c->nts(L.g) | (c->mg_v32_emul ? 2 : 0));
            } else {
                hipLaunchKernelGGL((k_mg_down<false>), grid, dim3(256), 0, st, op, mgeo, coarse_df, rhs[l], om, om2, cur[l], cg, cb);
            }
            CFD_HIP(hipGetLastError());
            if (l < la && (L.g.cutlo || L.g.cuthi)) {
                // (a single-precision x: rows of half the bytes, which is a plane of half the pitch to the exchange)
                Geo gx = L.g, gcb = C.g;
                if ((l == 0 && opm.cif && KS == 3) || own32(l)) gx.pitch /= 2;
                if (own32(l + 1)) gcb.pitch /= 2;              // float right-hand side of the next level
                HaloSpec sp[2] = {{&gx, KS, cur[l]}, {nullptr, 0, nullptr}};
                int ns = 1;
                x_sent[l] = 1;
                if (l + 1 < la && l + 2 < nl && level_fusable(c, l + 1, C.g, coarse_view(c, l + 2))) {
                    sp[ns++] = HaloSpec{&gcb, KS, cb};
                    rhs_sent[l + 1] = 1;
                }
                CFD_TRY(halo_exchange_multi(c, ns, sp));
            }
        } else if (L.line) {
            // alternating line smoother from a zero guess (its first sweep writes the other plane), then residual + restriction
            std::swap(cur[l], oth[l]);
            if (line_transfer_fusable(c, l, L.g, cg)) {
                CFD_TRY(line_smooth(c, L, op, rhs[l], &cur[l], &oth[l], false, true, false, true));
                const MarchGeo mgeo = march_geo_w(L.g, 60);
                hipLaunchKernelGGL((k_mg_down_x<4>), dim3(march_blocks(mgeo)), dim3(256), 0, st, op, mgeo, DiagFix{}, rhs[l], (const double *)cur[l],
                                   c->mg_line_omega, oth[l], cg, cb);
                CFD_HIP(hipGetLastError());
                std::swap(cur[l], oth[l]);
                continue;
            }
            CFD_TRY(line_smooth(c, L, op, rhs[l], &cur[l], &oth[l], false, true, l < la));
            CFD_TRY(halo_exchange(c, L.g, 1, cur[l]));
            int fi, fj;
            factors(L.g, cg, fi, fj);
            const Tiling Tc = tiling(cg);
            hipLaunchKernelGGL(k_mg_restrict, dim3(Tc.blocks), block2d(), 0, st, op, rhs[l], cur[l], cg, Tc, fi, fj, cb);
            CFD_HIP(hipGetLastError());
        } else {
            const Tiling T = tiling(L.g);
            hipLaunchKernelGGL(k_mg_first, dim3(T.blocks), block2d(), 0, st, L.g, T, L.D, rhs[l], om, cur[l]);
            for (int s = 1; s < pre; ++s) {
                CFD_TRY(halo_exchange(c, L.g, 1, cur[l]));
                hipLaunchKernelGGL(k_mg_jacobi, dim3(T.blocks), block2d(), 0, st, op, T, rhs[l], cur[l], c->om_at(s), oth[l]);
                std::swap(cur[l], oth[l]);
            }
            CFD_TRY(halo_exchange(c, L.g, 1, cur[l]));
            int fi, fj;
            factors(L.g, cg, fi, fj);
            const Tiling Tc = tiling(cg);
            hipLaunchKernelGGL(k_mg_restrict, dim3(Tc.blocks), block2d(), 0, st, op, rhs[l], cur[l], cg, Tc, fi, fj, cb);
            CFD_HIP(hipGetLastError());
        }
        if (to_agg) CFD_TRY(all_gather_rows(c, cb + NG * (size_t)cg.pitch, C.b + NG * (size_t)cg.pitch, (size_t)cg.nj * cg.pitch));
    }
    if (lt >= 0) {
        const TailDev a = tail_launch_args(c, c->tail);
        const size_t lds_bytes = (size_t)tail_lds_doubles(a.total, (long)a.L[0].ni * a.L[0].nj, a.ncache) * sizeof(double);
        static bool lds_limit_raised = false;
        if (!lds_limit_raised) {
            CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_mg_tail)));
            lds_limit_raised = true;
        }
        hipLaunchKernelGGL(k_mg_tail, dim3(1), dim3(TAIL_NT), lds_bytes, st, a, rhs[lt], cur[lt]);
        CFD_HIP(hipGetLastError());
        if (nl - lt > c->kc[KC_MG_TAIL_LEVELS]) c->kc[KC_MG_TAIL_LEVELS] = nl - lt;
    } else {
        const Level &L = c->levels[nl - 1];
        const OpDev op = level_op(c, L);
        if (nl == 1) {
            // single-level "hierarchy": plain Jacobi sweeps
            const Tiling T = tiling(L.g);
            const double omc = c->om_mean();
            hipLaunchKernelGGL(k_mg_first, dim3(T.blocks), block2d(), 0, st, L.g, T, L.D, rhs[0], omc, cur[0]);
            for (int s = 1; s < c->mg_coarse_sweeps; ++s) {
                hipLaunchKernelGGL(k_mg_jacobi, dim3(T.blocks), block2d(), 0, st, op, T, rhs[0], cur[0], omc, oth[0]);
                std::swap(cur[0], oth[0]);
            }
        } else {
            // the many sweeps of the coarsest solve use the mean damping (an undamped omega = 1 would leave
            // the checkerboard mode untouched)
            hipLaunchKernelGGL(k_mg_coarsest, dim3(1), dim3(256), 0, st, op, rhs[nl - 1], c->om_mean(), c->mg_coarse_sweeps,
                               cur[nl - 1], oth[nl - 1]);
        }
        CFD_HIP(hipGetLastError());
    }
    *dot_blocks = 0;
    for (int l = n_top - 1; l >= 0; --l) {
        const Level &L = c->levels[l];
        const Level &C = c->levels[l + 1];
        const OpDev op = level_op(c, L), opm = level_op_mg(c, L);
        const Geo cg = coarse_view(c, l + 1);
        // coarse correction as this level sees it: its own slab rows of the replicated plane
        // (whose neighbouring rows are the halo), or the exchanged slab plane
        const double *e = cur[l + 1];
        if (la > 0 && l + 1 == la) e += (size_t)C.joff * C.g.pitch;
        else if (l + 1 < la) {
            Geo ge = C.g;
            if (own32(l + 1)) ge.pitch /= 2;                   // float iterate of the next level
            CFD_TRY(halo_exchange(c, ge, level_fusable(c, l, L.g, cg) && KS > 2 ? 2 : 1, cur[l + 1]));   // rows (j0 - K) >> 1 ...
        }
        if (level_fusable(c, l, L.g, cg)) {
            MarchGeo mgeo = march_geo_k(L.g, KS, c->march_ls_model != 0);
            mgeo.xcd = KS == 3 ? c->xcd_map : 0;
            const int wide = KS == 3 ? march_wide(c, mgeo, l > 0 ? WIDE_MG_COARSE : WIDE_MG_UP) : 4;
            const dim3 grid(march_grid(mgeo, wide));
            if (!x_sent[l]) {
                Geo gx = L.g;
                if ((l == 0 && opm.cif && KS == 3) || own32(l)) gx.pitch /= 2;
                CFD_TRY(halo_exchange(c, gx, KS, cur[l]));
            }
            if (l == 0) {
                c->probe_begin(7);
                if (KS == 3 && opm.cif && z_f32) {
                    // the caller (fused PCG) takes z in single precision
                    if (c32) LAUNCH_WIDE(wide, (k_mg_up_k<true, 3, 4, true, true, 1>), (k_mg_up_k<true, 3, 8, true, true, 1>), grid, opm, mgeo, c->p_diag, rhs[l], cur[l], cg, e,
                                         c->mg_gamma, om3, oth[l], c->partials, c->nts(L.g));
                    else LAUNCH_WIDE(wide, (k_mg_up_k<true, 3, 4, true, true>), (k_mg_up_k<true, 3, 8, true, true>), grid, opm, mgeo, c->p_diag, rhs[l], cur[l], cg, e,
                                     c->mg_gamma, om3, oth[l], c->partials, c->nts(L.g));
                    *z_f32 = true;
                } else if (c32)
                    LAUNCH_WIDE(wide, (k_mg_up_k<true, 3, 4, true, false, 1>), (k_mg_up_k<true, 3, 8, true, false, 1>), grid, opm, mgeo, c->p_diag, rhs[l], cur[l], cg, e, c->mg_gamma,
                                om3, oth[l], c->partials, c->nts(L.g));
                else if (KS == 3)
                    LAUNCH_WIDE_F(opm.cif != nullptr, wide, (k_mg_up_k<true, 3>), (k_mg_up_k<true, 3, 8>), (k_mg_up_k<true, 3, 4, true>), (k_mg_up_k<true, 3, 8, true>), grid, opm, mgeo, c->p_diag, rhs[l], cur[l], cg, e, c->mg_gamma,
                                om3, oth[l], c->partials, c->nts(L.g) | (c->mg_v32_emul ? 2 : 0));
                else
                    hipLaunchKernelGGL((k_mg_up<true>), grid, dim3(256), 0, st, op, mgeo, c->p_diag, rhs[l], cur[l], cg, e, c->mg_gamma, om, om2,
                                       oth[l], c->partials);
                c->probe_end(7, (c->p_diag.on ? 42.0 : 50.0) - (opm.cif ? 12.0 : 0.0) - ((z_f32 && *z_f32) ? 4.0 : 0.0) - (c32 ? 1.0 : 0.0));
                *dot_blocks = (int)grid.x;
            } else if (own32(l) && own32(l + 1)) {
                LAUNCH_WIDE(wide, (k_mg_up_k<false, 3, 4, true, false, 3>), (k_mg_up_k<false, 3, 8, true, false, 3>), grid, opm, mgeo, coarse_df, rhs[l], cur[l], cg, e, c->mg_gamma, om3,
                            oth[l], c->partials, c->nts(L.g));
            } else if (own32(l)) {
                LAUNCH_WIDE(wide, (k_mg_up_k<false, 3, 4, true, false, 2>), (k_mg_up_k<false, 3, 8, true, false, 2>), grid, opm, mgeo, coarse_df, rhs[l], cur[l], cg, e, c->mg_gamma, om3,
                            oth[l], c->partials, c->nts(L.g));
            } else if (KS == 3) {
                LAUNCH_WIDE_F(opm.cif != nullptr, wide, (k_mg_up_k<false, 3>), (k_mg_up_k<false, 3, 8>), (k_mg_up_k<false, 3, 4, true>), (k_mg_up_k<false, 3, 8, true>), grid, opm, mgeo, coarse_df, rhs[l], cur[l], cg, e, c->mg_gamma, om3,
                            oth[l], c->partials, c->nts(L.g) | (c->mg_v32_emul ? 2 : 0));
            } else {
                hipLaunchKernelGGL((k_mg_up<false>), grid, dim3(256), 0, st, op, mgeo, coarse_df, rhs[l], cur[l], cg, e, c->mg_gamma, om, om2,
                                   oth[l], c->partials);
            }
            std::swap(cur[l], oth[l]);
            CFD_HIP(hipGetLastError());
            continue;
        }
        if (line_transfer_fusable(c, l, L.g, cg)) {
            const MarchGeo mgeo = march_geo_w(L.g, 62);
            hipLaunchKernelGGL((k_mg_up_k<false, 1>), dim3(march_blocks(mgeo)), dim3(256), 0, st, op, mgeo, DiagFix{}, rhs[l], (const double *)cur[l],
                               cg, e, c->mg_gamma, SweepOm<1>{{c->mg_line_omega}}, oth[l], c->partials, 0);
            CFD_HIP(hipGetLastError());
            std::swap(cur[l], oth[l]);
            CFD_TRY(line_smooth(c, L, op, rhs[l], &cur[l], &oth[l], true, false, false, true));
            continue;
        }
        const Tiling T = tiling(L.g);
        int fi, fj;
        factors(L.g, cg, fi, fj);
        hipLaunchKernelGGL(k_mg_prolong, dim3(T.blocks), block2d(), 0, st, L.g, T, cg, fi, fj, e, c->mg_gamma, cur[l]);
        if (L.line) {
            CFD_HIP(hipGetLastError());
            CFD_TRY(line_smooth(c, L, op, rhs[l], &cur[l], &oth[l], true, false, l < la));
            continue;
        }
        for (int s = 0; s < post; ++s) {
            CFD_TRY(halo_exchange(c, L.g, 1, cur[l]));
            hipLaunchKernelGGL(k_mg_jacobi, dim3(T.blocks), block2d(), 0, st, op, T, rhs[l], cur[l], c->om_at(s), oth[l]);
            std::swap(cur[l], oth[l]);
        }
        CFD_HIP(hipGetLastError());
    }
    // keep coarse-level pointers consistent for the next cycle
    for (int l = 1; l < nl; ++l) {
        c->levels[l].x = cur[l];
        c->levels[l].t = oth[l];
    }
    *zout = cur[0];
    return CFD_OK;
}

int vcycle(Ctx *c, const double *r, double *z) {
    double *out = nullptr;
    double *z1 = c->kry[15];
    int db = 0;
    CFD_TRY(vcycle_impl(c, r, z, z1, &out, &db));
    if (out != z) CFD_HIP(hipMemcpyAsync(z, out, c->g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return CFD_OK;
}

// ------------------------------------------------------------------ PCG (pressure)
struct CgScal {
    double rz[2], pq, rr, bb;
};
constexpr int CG_OFF = 40;
#define CG_OFF_OF(member) ((int)(CG_OFF + offsetof(CgScal, member) / sizeof(double)))

// q = A p ; partial = (p, q)
__global__ void k_cg_q(OpDev op, Tiling T, const double *p, double *q, double *partials) {
    const Geo g = op.g;
    double acc[1] = {0.0};
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double v = op.D[c] * p[c] - offdiag(op, [p](size_t n) { return p[n]; }, c, i, j);
        q[c] = v;
        acc[0] += v * p[c];
    }
    block_sums<1>(acc, partials);
}
// x += alpha p ; r -= alpha q ; partial = (r, r)
__global__ void k_cg_xr(Geo g, Tiling T, const CgScal *s, int gen, const double *p, const double *q, double *x, double *r,
                        double *partials) {
    const double alpha = safe_div(s->rz[gen], s->pq);
    double acc[1] = {0.0};
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        x[c] = x[c] + alpha * p[c];
        const double rn = r[c] - alpha * q[c];
        r[c] = rn;
        acc[0] += rn * rn;
    }
    block_sums<1>(acc, partials);
}
// p = z + beta p  (first: p = z)
__global__ void k_cg_p(Geo g, Tiling T, const CgScal *s, int gen, int first, const double *z, double *p) {
    const double beta = first ? 0.0 : safe_div(s->rz[gen], s->rz[gen ^ 1]);
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        p[c] = first ? z[c] : z[c] + beta * p[c];
    }
}
// r = b ; x = 0 ; partial = (b, b)
__global__ void k_cg_init(Geo g, Tiling T, const double *b, double *r, double *x, double *partials) {
    double acc[1] = {0.0};
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double v = b[c];
        r[c] = v;
        x[c] = 0.0;
        acc[0] += v * v;
    }
    block_sums<1>(acc, partials);
}

// Fused PCG direction step (marching, one halo lane / row):  p' = z + beta p ;  q = A p' ;  (p', q) ;
// and the solution update of the PREVIOUS iteration, x += alpha_prev p, which reads the same p.
// 64 B/cell on the finest level (z, p, x, c_i, c_j read; p', q, x written; D is recomputed from the face planes;
// 40 / 56 B on the first / second step of a solve, which skip p / x), 72 B on levels that read a D plane, instead of
// 24 + 40 for k_cg_p + k_cg_q and 16 of the 48 of k_cg_xr, whose remainder is k_cg_r below.  p' goes to a second plane: neighbouring
// strips recompute p' on their halo from the old p.
// WPB = 4: every wave takes work items (strip, segment) on its own.  WPB = 8 / 16 (lockstep form): a workgroup takes
// WPB neighbouring strips of ONE segment and its waves meet at a barrier after every row, so that the workgroup reads
// WPB x 512 contiguous bytes of each plane row at a time -- the strips' 64-lane windows are not aligned to the
// 128-byte lines (62 output columns per strip), and waves drifting apart made every one of them pay for its two
// partial lines (tools/stream_probe3.hip: 4.1 TB/s for misaligned independent waves, 4.8 - 4.9 in lockstep).
// FIRST / XZERO (first / second step of a solve: no p_old / no x to read) are template arguments and the diagonal is
// formed where a row is consumed: a wave-uniform branch around a load, or arithmetic on a row's coefficients right
// after its loads were issued, makes the wave wait for the row it has just asked for (the prefetch then hides nothing).
// NOX (option pcg_xsum, default): the step leaves x alone.  Every direction p_k of the solve stays in a plane of its own
// (a ring of CG_RING planes) and alpha_k in a device slot, and x = sum_k alpha_k p_k is summed once, in the order the
// updates would have been applied, when the solve ends or the ring is full (k_cg_xsum): a five-iteration solve reads
// five planes and writes one instead of reading and writing x in every step (-16 B per cell and step from the third).
// `alpha_out` (NOX, not the first step) receives alpha_prev.
template <int WPB, bool FIRST, bool XZERO, bool ZF = false, bool NOX = false>
__global__ __launch_bounds__(64 * WPB) void k_cg_pq(OpDev op, MarchGeo mg, DiagFix df, CgScal *s, int gen,
                                                    const double *z, const double *pold, double *pnew, double *q, double *x, double *partials,
                                                    int nts, double *alpha_out, const double *rz_part, int rz_n) {
    const Geo g = op.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row loops and addresses stay scalar
    const size_t P = g.pitch;
    constexpr bool first = FIRST, xzero = XZERO || NOX;
    // fold_sums: (r, z) of this iteration still waits as rz_n partial sums of the V-cycle's last pass (fold_sum); the first
    // workgroup stores it where the later kernels of the iteration read it
    double rz_cur;
    if (rz_n > 0) {
        __shared__ double fsm[4];
        rz_cur = fold_sum(rz_part, rz_n, threadIdx.x, fsm);
        if (blockIdx.x == 0 && threadIdx.x == 0) s->rz[gen] = rz_cur;
    } else {
        rz_cur = s->rz[gen];
    }
    const double beta = first ? 0.0 : safe_div(rz_cur, s->rz[gen ^ 1]);
    const double alpha_prev = first ? 0.0 : safe_div(s->rz[gen ^ 1], s->pq);
    if (NOX && !first && blockIdx.x == 0 && threadIdx.x == 0) alpha_out[0] = alpha_prev;
    double dot = 0.0;
    constexpr bool LOCK = WPB > 4;
    const int ngroups = (mg.nstrips + WPB - 1) / WPB;
    const int n_items = LOCK ? ngroups * mg.nsegs : mg.nstrips * mg.nsegs;
    const int first_item = LOCK ? (int)blockIdx.x : xcd_block(blockIdx.x, gridDim.x, mg.xcd) * 4 + wave;
    const int stride = LOCK ? (int)gridDim.x : (int)gridDim.x * 4;
    for (int item = first_item; item < n_items; item += stride) {
        // lockstep: (group of WPB strips, segment) per workgroup; a wave beyond the last strip repeats it without output
        const int strip_raw = LOCK ? (item % ngroups) * WPB + wave : item % mg.nstrips;
        const int strip = strip_raw < mg.nstrips ? strip_raw : mg.nstrips - 1;
        const int seg = LOCK ? item / ngroups : item / mg.nstrips;
        const int iraw = strip * 62 - 1 + lane;
        const int col = wrap_index(iraw, g.ni, op.per_i);
        const bool cvalid = col >= 0;
        const size_t colo = (size_t)(cvalid ? col : 0) + CO;
        const bool out = lane >= 1 && lane < 63 && iraw < g.ni && strip_raw < mg.nstrips;
        const int j0 = seg * mg.Ls, j1 = min(j0 + mg.Ls, g.nj);
        double pa = 0.0, pb = 0.0, pc, po_b = 0.0, po_c;
        double d_b = 1.0, cw_b = 0.0, ce_b = 0.0, cs_b = 0.0, cn_b = 0.0;
        size_t cell_b = NG * P + colo;
        // software pipeline: row j + 1 is loaded while row j is processed
        struct RowIn {
            double z, po, d, cw, ce, cs, cn, xv;
            float zf;                  // ZF: z as the finest-level post-smoother stored it (converted where consumed)
            bool valid;
            int row;
            size_t cell;
        };
        auto fetch = [&](int j, RowIn &R) {
            const int mr = mem_row(j, g, op.per_j);
            R.valid = cvalid && mr >= 0;
            const size_t cc = (size_t)(mr >= 0 ? mr : NG) * P + colo;
            R.cell = cc;
            R.po = first ? 0.0 : pold[cc];
            if (ZF) R.zf = reinterpret_cast<const float *>(z)[cc];
            else R.z = z[cc];
            R.xv = (first || xzero) ? 0.0 : x[cc];      // xzero: x = 0 was never stored (second iteration)
            R.cw = op.ci[cc]; R.ce = op.ci[cc + 1]; R.cs = op.cj[cc];
            R.cn = op.cj[cc + north_offset(mr < g.nj + 2 * NG - 1, P)];
            // rows j0-1 .. j1 only: the outermost halo row is never touched here
            R.row = (mr >= 0 ? mr : NG) - NG;
            R.d = df.on ? 0.0 : op.D[cc];
        };
        double xv_b = 0.0, xv_c;
        auto trip = [&](int j, const RowIn &cur) {
            const bool valid = cur.valid;
            const size_t cc = cur.cell;
            po_c = cur.po;
            xv_c = cur.xv;
            pc = valid ? (ZF ? (double)cur.zf : cur.z) + beta * po_c : 0.0;
            const double cw_c = cur.cw, ce_c = cur.ce, cs_c = cur.cs, cn_c = cur.cn;
            const double d_c = df.on ? diag_from_faces(cw_c, ce_c, cs_c, cn_c, (int)colo - CO, cur.row, df, op) : cur.d;
            // row j-1
            const double pl = lane_left(pb), pr = lane_right(pb);
            if (j - 1 >= j0 && out) {
                const double qv = d_b * pb - (cw_b * pl + ce_b * pr + cs_b * pa + cn_b * pc);
                store_stream(pnew + cell_b, pb, nts);
                store_stream(q + cell_b, qv, nts);
                if (!first && !NOX) store_stream(x + cell_b, xv_b + alpha_prev * po_b, nts);
                dot += pb * qv;
            }
            pa = pb; pb = pc; po_b = po_c; xv_b = xv_c;
            d_b = d_c; cw_b = cw_c; ce_b = ce_c; cs_b = cs_c; cn_b = cn_c;
            cell_b = cc;
        };
        // two rows of loads in flight per wave; a fetch beyond the last row repeats it and is never consumed
        RowIn r0, r1;
        fetch(j0 - 1, r0);
        fetch(min(j0, j1), r1);
        for (int j = j0 - 1; j <= j1; j += 2) {
            {
                const RowIn cur = r0;
                fetch(min(j + 2, j1), r0);
                trip(j, cur);
            }
            if (j + 1 <= j1) {
                const RowIn cur = r1;
                fetch(min(j + 3, j1), r1);
                trip(j + 1, cur);
            }
            if (LOCK) __syncthreads();       // same j0, j1 for every wave of the workgroup: uniform trip count
        }
    }
    __shared__ double sm[WPB];
    const double w = wsum(dot);
    if (lane == 0) sm[wave] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int q4 = 0; q4 < WPB; q4 += 4) tot += (sm[q4] + sm[q4 + 1]) + (sm[q4 + 2] + sm[q4 + 3]);
        partials[blockIdx.x] = tot;
    }
}
// r -= alpha q ; partial = (r, r)
__global__ void k_cg_r(Geo g, Tiling T, CgScal *s, int gen, const double *q, double *r, double *partials, int nts,
                       const double *pq_part = nullptr, int pq_n = 0) {
    double pq;
    if (pq_n > 0) {          // fold_sums: (p, q) summed here (see k_cg_pq)
        __shared__ double fsm[4];
        pq = fold_sum(pq_part, pq_n, (int)(threadIdx.y * blockDim.x + threadIdx.x), fsm);
        if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) s->pq = pq;
    } else {
        pq = s->pq;
    }
    const double alpha = safe_div(s->rz[gen], pq);
    double acc[1] = {0.0};
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        const double rn = r[c] - alpha * q[c];
        store_stream(r + c, rn, nts);
        acc[0] += rn * rn;
    }
    block_sums<1>(acc, partials);
}
// x += alpha p  (the update the fused direction step still owes after the last iteration)
__global__ void k_cg_xlast(Geo g, Tiling T, const CgScal *s, int gen, int xzero, const double *p, const double *shift, double *x) {
    const double alpha = safe_div(s->rz[gen], s->pq);
    const double sh = shift ? shift[0] : 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        x[c] = ((xzero ? 0.0 : x[c]) + alpha * p[c]) + sh;
    }
}
// x = (x +) sum_k alpha_k p_k over the n <= CG_RING direction planes a solve has left (option pcg_xsum, see k_cg_pq<.., NOX>),
// added in iteration order with the expressions of the stepwise updates (0 + a_0 p_0, + a_1 p_1, ..., + shift: the same bits).
// alpha_k of all but the last direction wait in ahist[k]; the last one is still rz / pq in the solver's scalars.
constexpr int CG_RING = 6;
constexpr int CG_AHIST_OFF = 48;          // Ctx::scal slots [48, 48 + CG_RING)
struct XsumArgs {
    const double *p[CG_RING];
    int n, from_x;
};
__device__ __forceinline__ void xsum_alphas(const CgScal *s, int gen_last, const XsumArgs &a, const double *ahist, double (&al)[CG_RING]) {
    const double last = safe_div(s->rz[gen_last], s->pq);
#pragma unroll
    for (int k = 0; k < CG_RING; ++k) al[k] = k < a.n - 1 ? ahist[k] : last;
}
__global__ void k_cg_xsum(Geo g, Tiling T, const CgScal *s, int gen_last, XsumArgs a, const double *ahist, const double *shift, double *x) {
    double al[CG_RING];
    xsum_alphas(s, gen_last, a, ahist, al);
    const double sh = shift ? shift[0] : 0.0;
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        double acc = a.from_x ? x[c] : 0.0;
#pragma unroll
        for (int k = 0; k < CG_RING; ++k)
            if (k < a.n) acc = acc + al[k] * a.p[k][c];
        x[c] = acc + sh;
    }
}
// nullspace form with pcg_xsum: the shift from the solution at the reference cell, summed like k_cg_xsum does
__global__ void k_ns_shift_sum(Geo g, int pi, int pj, double value, const CgScal *s, int gen_last, XsumArgs a, const double *ahist,
                               const double *x, double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double al[CG_RING];
    xsum_alphas(s, gen_last, a, ahist, al);
    const size_t c = g.cell(pi, pj);
    double acc = a.from_x ? x[c] : 0.0;
#pragma unroll
    for (int k = 0; k < CG_RING; ++k)
        if (k < a.n) acc = acc + al[k] * a.p[k][c];
    out[0] = value - acc;
}
// nullspace form: the constant that moves the solution to the reference value at the reference cell;
// with_p: the last x += alpha p is still owed (fused path)
__global__ void k_ns_shift(Geo g, int pi, int pj, double value, const CgScal *s, int gen, int with_p, int xzero, const double *p,
                           const double *x, double *out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const size_t c = g.cell(pi, pj);
    double xr = xzero ? 0.0 : x[c];
    if (with_p) xr = xr + safe_div(s->rz[gen], s->pq) * p[c];
    out[0] = value - xr;
}
__global__ void k_add_scalar(Geo g, Tiling T, const double *sh, double *x) {
    const double v = sh[0];
    TILE_LOOP(g, T) {
        const size_t c = g.cell(i, j);
        x[c] = x[c] + v;
    }
}
// scal[81] = shift (owner rank computes it, the other slabs contribute zero to the sum)
static int ns_shift_scalar(Ctx *c, const CgScal *S, int gen, int with_p, int xzero, const double *p, const double *x) {
    if (c->pin_i >= 0)
        hipLaunchKernelGGL(k_ns_shift, dim3(1), dim3(64), 0, c->stream, c->g, c->pin_i, c->pin_j, c->pin_value, S, gen, with_p, xzero,
                           p, x, c->scal + 81);
    else
        CFD_HIP(hipMemsetAsync(c->scal + 81, 0, sizeof(double), c->stream));
    CFD_HIP(hipGetLastError());
    return all_reduce(c, c->scal + 81, 1, 0);
}

// The whole solve as one launch of k_pcg_tail (meshes whose hierarchy fits one workgroup's LDS, Ctx::ptail_ok).  The host
// reads three scalars afterwards -- or, in a speculative iteration, leaves them to the iteration's closing read: the
// device has already applied the stopping test, so there is no schedule to replay and nothing to repeat unless the
// solve ran into maxiter.
static int pcg_tail_solve(Ctx *c, const double *b, double *x, double tol, int maxiter, int *iters, double *relres) {
    const TailDev a = tail_launch_args(c, c->ptail);
    const Geo &g = a.g0;
    const long top = (long)g.ni * g.nj;
    const bool spec = c->spec_on && c->spec_pcg.n < 4 && c->ptail_good;     // (after a host-checked solve that converged)
    PcgTailArgs pa{};
    pa.tol = tol; pa.pin_value = c->pin_value; pa.maxiter = maxiter; pa.ns_active = c->p_ns_active ? 1 : 0;
    pa.pin_idx = c->pin_i >= 0 ? c->pin_j * g.ni + c->pin_i : -1;
    pa.bb_slot = spec ? 8 : 2;
    double *out = spec ? c->scal + SPEC_PCG_OFF + SPEC_PCG_STRIDE * c->spec_pcg.n : c->scal + CG_OFF;
    static bool lds_limit_raised = false;
    if (!lds_limit_raised) {
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_pcg_tail<1>)));
        CFD_TRY(raise_tail_lds_limit(reinterpret_cast<const void *>(&k_pcg_tail<TAIL_CPT>)));
        lds_limit_raised = true;
    }
    const size_t lds_bytes = (size_t)pcg_tail_lds_doubles(a.total, top, a.ncache) * sizeof(double);
    // one cell per thread: as many waves as the finest level has cells for (a 5 x 20 mesh: two waves meet at the barriers, not sixteen)
    if (top <= TAIL_NT) hipLaunchKernelGGL(k_pcg_tail<1>, dim3(1), dim3(tail_threads(top)), lds_bytes, c->stream, a, pa, b, x, out);
    else hipLaunchKernelGGL(k_pcg_tail<TAIL_CPT>, dim3(1), dim3(TAIL_NT), lds_bytes, c->stream, a, pa, b, x, out);
    CFD_HIP(hipGetLastError());
    c->kc[KC_PCG_TAIL] += 1;
    c->bb_ready = false;
    if (spec) {
        Ctx::SpecPcg &sp = c->spec_pcg;
        sp.hint[sp.n] = -1;              // stopped by the device: slot[0] = iterations, slot[1] = |r|^2, slot[8] = |b|^2
        sp.tol[sp.n] = tol;
        sp.n += 1;
        *iters = c->p_iters_hint;
        *relres = 0.0;                   // known when the iteration's closing read has arrived (spec_verify)
        return CFD_OK;
    }
    double h[3];
    CFD_TRY(read_scalars(c, CG_OFF, 3, h));
    if (!(h[2] == h[2])) { set_error("pressure solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
    if (!(h[1] == h[1])) { set_error("PCG: non-finite residual at iteration %d", (int)h[0]); return CFD_E_NUMERIC; }
    *iters = (int)h[0];
    *relres = h[2] > 0.0 ? sqrt(h[1] / h[2]) : 0.0;
    c->p_iters_hint = *iters;
    c->ptail_good = sqrt(h[1]) <= tol * sqrt(h[2]);
    return CFD_OK;
}

static int pcg_mg(Ctx *c, const double *b, double *x, double tol, int maxiter, int *iters, double *relres) {
    if (c->ptail_ok && c->pcg_tail) return pcg_tail_solve(c, b, x, tol, maxiter, iters, relres);
    const Level &L0 = c->levels[0];
    const OpDev op = level_op(c, L0);
    const Geo g = op.g;
    const Tiling T = tiling(g);
    const dim3 grid(T.blocks), blk = block2d();
    hipStream_t st = c->stream;
    CgScal *S = reinterpret_cast<CgScal *>(c->scal + CG_OFF);
    double *r = c->kry[1], *p = c->kry[2], *q = c->kry[3], *za = c->kry[4], *zb = c->kry[5];
    // fused direction step: marching kernel, p ping-pongs between two planes
    const bool fused = c->mg_fused && g.ni >= c->fused_min_pcg && g.nj >= c->fused_min_pcg;
    double *pbuf[2] = {c->kry[2], c->kry[6]};
    // pcg_xsum: the directions of the solve stay in a ring of planes and x is summed from them at the end (k_cg_pq<.., NOX>);
    // beside kry[2] and kry[6] the ring takes Chebyshev / transport work planes, which idle during a pressure solve (the
    // test entry points hand in kry[11] / kry[12] as b and x: whatever the caller uses is left out; kry[10] keeps the saved
    // right-hand side of the nearly-symmetric regime)
    const bool xsum = fused && c->pcg_xsum;
    // fold_sums (one rank): (r, z) and (p, q) are added up by the kernel that consumes them (fold_sum) instead of by a
    // single-workgroup launch in between; (p, q) partial sums then go to a buffer of their own, because the direction step
    // writes them while its other workgroups still read the (r, z) ones
    // (auto: meshes of up to 4M cells, where a dependent launch costs a visible share of the step -- 2 - 3 % of it at
    // 64^2 .. 1024^2; at 8192^2 the sum at the head of a thousand workgroups costs as much as the launch it replaces)
    const bool fold = fused && c->comm.nranks == 1 && (c->fold_sums > 0 || (c->fold_sums < 0 && (long)g.ni * g.nj <= (1L << 22)));
    double *pq_part = fold ? c->partials + (size_t)5 * MAX_PARTIALS : c->partials;
    int pq_pending = 0;                // (p, q) of the last direction step still waits as this many partial sums
    double *pring[CG_RING];
    {
        const int cand[9] = {2, 6, 7, 8, 9, 11, 12, 13, 14};
        int n = 0;
        for (int q = 0; q < 9 && n < CG_RING; ++q)
            if (c->kry[cand[q]] != b && c->kry[cand[q]] != x) pring[n++] = c->kry[cand[q]];
    }
    double *ahist = c->scal + CG_AHIST_OFF;
    bool x_started = false;            // x holds the sum of the directions of earlier ring fills
    // planes 0 .. n-1 of the ring = the directions p_{it-n} .. p_{it-1}, whose last alpha is still rz[(it-1) & 1] / pq
    auto ring_args = [&](int it_done) {
        XsumArgs a{};
        a.n = (it_done - 1) % CG_RING + 1;
        a.from_x = x_started ? 1 : 0;
        for (int k = 0; k < CG_RING; ++k) a.p[k] = pring[k];
        return a;
    };
    MarchGeo mgeo = march_geo(g);
    mgeo.nstrips = (g.ni + 61) / 62;
    mgeo.xcd = c->xcd_map;
    // lockstep workgroups of 8 / 16 neighbouring strips on meshes wide enough to fill them
    const int wide = march_wide(c, mgeo, WIDE_CG_PQ);
    const dim3 mgrid(march_grid(mgeo, wide));
    CFD_HIP(hipMemsetAsync(S, 0, sizeof(CgScal), st));
    if (fused) {
        // the right-hand side plane is a work plane of the caller: it becomes the residual in place, and
        // x = 0 is never stored (the first two direction steps and k_cg_xlast know it)
        r = const_cast<double *>(b);
    }
    if (fused && c->bb_ready) {
        // the pass that built this right-hand side summed b^2 as well
        CFD_HIP(hipMemcpyAsync(c->scal + CG_OFF_OF(bb), c->scal + 83, sizeof(double), hipMemcpyDeviceToDevice, st));
    } else {
        if (fused) {
            CVecs<1> a1{{b}};
            hipLaunchKernelGGL((k_dot<1>), grid, blk, 0, st, g, T, a1, a1, c->partials);
        } else {
            hipLaunchKernelGGL(k_cg_init, grid, blk, 0, st, g, T, b, r, x, c->partials);
        }
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, 1, CG_OFF_OF(bb)));
    }
    c->bb_ready = false;
    double bb = -1.0, rr = 0.0;
    *iters = 0;
    *relres = 0.0;
    // |b|^2 is looked at when the first residual is: with an iteration count to expect (a previous solve with this
    // operator) the host does not stop the GPU for it.  A zero right-hand side then runs through the iterations
    // with alpha = beta = 0 (safe_div) and leaves x = 0.
    const bool lazy_bb = c->host_lag && c->p_iters_hint > 0;
    if (!lazy_bb) {
        CFD_TRY(read_scalars(c, CG_OFF_OF(bb), 1, &bb));
        if (!(bb == bb)) { set_error("pressure solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
        if (bb == 0.0) {
            if (fused) CFD_HIP(hipMemsetAsync(x, 0, g.plane_elems() * sizeof(double), st));
            if (c->p_ns_active && c->pin_value != 0.0) {
                CFD_TRY(ns_shift_scalar(c, S, 0, 0, 1, x, x));
                hipLaunchKernelGGL(k_add_scalar, grid, blk, 0, st, g, T, c->scal + 81, x);
                CFD_HIP(hipGetLastError());
            }
            return CFD_OK;
        }
        rr = bb;
    }
    int it = 0;
    const int hint0 = c->p_iters_hint;
    int converged_at = -1;
    // speculative iteration (internal.h: Ctx::speculate): as many iterations as the previous solve needed, |r|^2 of the last
    // two and |b|^2 kept in device slots for the iteration's closing read instead of being read here
    const bool spec = c->spec_on && lazy_bb && fused && hint0 >= 1 && hint0 < maxiter && c->spec_pcg.n < 4;
    bool spec_pending = false;
    // Iterations whose |r| the host does not look at leave r = r - alpha q to the pre-smoother of the next V-cycle
    // (k_mg_down_k<true, 3, true>): r ping-pongs between the right-hand side plane and kry[1].  Needs the three-sweep
    // marching kernels on the finest level; on j-slabs q crosses the cuts in place of r.
    // (semi-coarsened hierarchies: the two kernels of the finest level's way down form r, option semi_resid; no (r, r)
    // by-product there)
    const bool semi_defer = c->semi_on && c->semi_resid && c->fuse_resid && fused;
    const bool can_defer = semi_defer || (c->fuse_resid && fused && c->levels.size() > 1 && c->mg_pre == 3 && c->mg_post == 3 &&
                                          level_fusable(c, 0, L0.g, coarse_view(c, 1)));
    double *r_other = c->kry[1];
    bool deferred = false;
    int deferred_gen = 0, deferred_rr_slot = 0;
    while (it < maxiter) {
        const int gen = it & 1;
        double *z = nullptr;
        int db = 0;
        bool zf = false;           // z came out of the V-cycle in single precision (finest level with single-precision faces)
        if (deferred) {
            ResidFuse rf{q, &S->rz[deferred_gen], &S->pq, r_other};
            if (deferred_rr_slot) { rf.rr_partials = c->partials + (size_t)7 * MAX_PARTIALS; rf.rr_slot = deferred_rr_slot; }
            if (pq_pending > 0) { rf.pq_part = pq_part; rf.pq_n = pq_pending; rf.pq_out = &S->pq; pq_pending = 0; }
            CFD_TRY(vcycle_impl(c, r, za, zb, &z, &db, &rf, fused ? &zf : nullptr));
            std::swap(r, r_other);
            deferred = false;
        } else {
            CFD_TRY(vcycle_impl(c, r, za, zb, &z, &db, nullptr, fused ? &zf : nullptr));
        }
        int rz_pending = 0;
        if (db > 0 && fold) {
            rz_pending = db;                                      // the direction step adds them up itself
        } else if (db > 0) {
            CFD_TRY(finish_to(c, db, 1, CG_OFF_OF(rz[gen])));     // (r, z) came out of the fused up-sweep
        } else {
            CVecs<1> a1{{r}}, a2{{z}};
            hipLaunchKernelGGL((k_dot<1>), grid, blk, 0, st, g, T, a1, a2, c->partials);
            CFD_HIP(hipGetLastError());
            CFD_TRY(finish_to(c, T.blocks, 1, CG_OFF_OF(rz[gen])));
        }
        if (fused) {
            double *pold = xsum ? pring[(it + CG_RING - 1) % CG_RING] : pbuf[gen ^ 1];
            p = xsum ? pring[it % CG_RING] : pbuf[gen];
            if (xsum && it > 0 && it % CG_RING == 0) {
                // the ring is full: its directions go into x before the first plane takes the next one
                c->probe_begin(12);
                hipLaunchKernelGGL(k_cg_xsum, grid, blk, 0, st, g, T, S, (it - 1) & 1, ring_args(it), ahist, (const double *)nullptr, x);
                c->probe_end(12, 8.0 * (CG_RING + 1 + (x_started ? 1 : 0)));
                CFD_HIP(hipGetLastError());
                x_started = true;
            }
            if (zf && (g.cutlo || g.cuthi) && c->comm.nranks > 1) {
                Geo gz = g;
                gz.pitch /= 2;                     // rows of half the bytes
                HaloSpec sp[2] = {{&gz, 1, z}, {&g, 1, pold}};
                CFD_TRY(halo_exchange_multi(c, it == 0 ? 1 : 2, sp));
            } else {
                CFD_TRY(halo_exchange(c, g, 1, z, it == 0 ? nullptr : pold));
            }
            c->probe_begin(0);
#define LAUNCH_CG_PQ(FIRST, XZERO, ZF, NOX)                                                                                   \
    do {                                                                                                                      \
        double *aout = ahist + (it + CG_RING - 1) % CG_RING;                                                                  \
        if (wide == 8)                                                                                                        \
            hipLaunchKernelGGL((k_cg_pq<8, FIRST, XZERO, ZF, NOX>), mgrid, dim3(512), 0, st, op, mgeo, c->p_diag, S, gen, z, pold, p, q, x, \
                               pq_part, c->nts(g), aout, (const double *)c->partials, rz_pending);                            \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_cg_pq<4, FIRST, XZERO, ZF, NOX>), mgrid, dim3(256), 0, st, op, mgeo, c->p_diag, S, gen, z, pold, p, q, x, \
                               pq_part, c->nts(g), aout, (const double *)c->partials, rz_pending);                            \
    } while (0)
            if (xsum) {
                if (zf) {
                    if (it == 0) LAUNCH_CG_PQ(true, false, true, true);
                    else LAUNCH_CG_PQ(false, false, true, true);
                } else {
                    if (it == 0) LAUNCH_CG_PQ(true, false, false, true);
                    else LAUNCH_CG_PQ(false, false, false, true);
                }
            } else if (zf) {
                if (it == 0) LAUNCH_CG_PQ(true, false, true, false);
                else if (it == 1) LAUNCH_CG_PQ(false, true, true, false);
                else LAUNCH_CG_PQ(false, false, true, false);
            } else {
                if (it == 0) LAUNCH_CG_PQ(true, false, false, false);
                else if (it == 1) LAUNCH_CG_PQ(false, true, false, false);
                else LAUNCH_CG_PQ(false, false, false, false);
            }
#undef LAUNCH_CG_PQ
            // algorithmic bytes: z, c_i, c_j read, p', q written (40); + p_old from the second step (48 -> with the x store 56);
            // + the x read from the third (64); pcg_xsum: 40 / 48, x is not touched
            c->probe_end(0, (xsum ? (it == 0 ? 40.0 : 48.0) : (it == 0 ? 40.0 : (it == 1 ? 56.0 : 64.0))) - (zf ? 4.0 : 0.0));
            c->kc[KC_CG_PQ] += 1;
            CFD_HIP(hipGetLastError());
            if (fold) pq_pending = (int)mgrid.x;
            else CFD_TRY(sum_finish(c, pq_part, (int)mgrid.x, 1, c->scal + CG_OFF_OF(pq)));
            // (lagged host reads look at |r| of the expected last iteration and the one before it only)
            if (can_defer && it + (c->host_lag ? 2 : 3) < c->p_iters_hint && it + 1 < maxiter) {
                deferred = true;               // the next V-cycle forms r itself; nobody reads |r| of this iteration
                deferred_gen = gen;
                deferred_rr_slot = 0;
                ++it;
                continue;
            }
            // the iteration before the expected last one: its |r|^2 is only read together with the last one's, so the
            // next V-cycle's pre-smoother forms r AND sums (r, r) into the history slot (no k_cg_r pass: 24 B per cell)
            if (can_defer && !semi_defer && c->host_lag && c->rr_byproduct && it + 2 == c->p_iters_hint && it + 1 < maxiter) {
                deferred = true;
                deferred_gen = gen;
                deferred_rr_slot = RR_HIST_OFF + ((it + 1) & 7);
                ++it;
                continue;
            }
            c->probe_begin(11);
            hipLaunchKernelGGL(k_cg_r, grid, blk, 0, st, g, T, S, gen, q, r, c->partials, c->nts(g), (const double *)pq_part, pq_pending);
            pq_pending = 0;
            c->probe_end(11, 24.0);
        } else {
            hipLaunchKernelGGL(k_cg_p, grid, blk, 0, st, g, T, S, gen, it == 0, z, p);
            CFD_TRY(halo_exchange(c, g, 1, p));
            c->probe_begin(0);
            hipLaunchKernelGGL(k_cg_q, grid, blk, 0, st, op, T, p, q, c->partials);
            c->probe_end(0);
            CFD_HIP(hipGetLastError());
            CFD_TRY(finish_to(c, T.blocks, 1, CG_OFF_OF(pq)));
            hipLaunchKernelGGL(k_cg_xr, grid, blk, 0, st, g, T, S, gen, p, q, x, r, c->partials);
        }
        CFD_HIP(hipGetLastError());
        ++it;
        // The host looks at |r| (one stream synchronisation, during which the GPU idles) first at the count the
        // previous solve with this frozen operator needed.  The iteration before it leaves its |r|^2 in a
        // device slot that the same read brings along: had it met the test already, the next solve expects that
        // count (this one has then iterated once more than it had to; the stopping test itself is unchanged).  Earlier iterations leave their partial sums of (r, r) unreduced (one small
        // kernel and, on j-slabs, one all-reduce less).
        if (c->host_lag) {
            if (it + 1 < hint0 && it < maxiter) continue;
            CFD_TRY(finish_to(c, T.blocks, 1, RR_HIST_OFF + (it & 7)));
            if (it < hint0 && it < maxiter) continue;
            if (spec && it == hint0) {
                converged_at = it;
                spec_pending = true;
                break;
            }
            double hist[8];
            CFD_TRY(read_scalars2(c, RR_HIST_OFF, 8, hist, CG_OFF_OF(bb), bb < 0.0 ? 1 : 0, &bb));
            if (!(bb == bb)) { set_error("pressure solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
            rr = hist[it & 7];
            if (!(rr == rr)) { set_error("PCG: non-finite residual at iteration %d", it); return CFD_E_NUMERIC; }
            if (sqrt(rr) <= tol * sqrt(bb)) {
                converged_at = it;
                if (it == hint0)
                    for (int k = it - 1; k >= 1 && k + 1 >= hint0; --k)
                        if (sqrt(hist[k & 7]) <= (c->spec_wanted ? SPEC_MARGIN : 1.0) * tol * sqrt(bb)) converged_at = k;      // (see spec_verify)
                break;
            }
            continue;
        }
        if (it + 2 < c->p_iters_hint && it < maxiter) continue;
        CFD_TRY(finish_to(c, T.blocks, 1, CG_OFF_OF(rr)));
        CFD_TRY(read_scalars(c, CG_OFF_OF(rr), 1, &rr));
        if (!(rr == rr)) { set_error("PCG: non-finite residual at iteration %d", it); return CFD_E_NUMERIC; }
        if (sqrt(rr) <= tol * sqrt(bb)) break;
    }
    c->p_iters_hint = converged_at > 0 ? converged_at : it;
    if (pq_pending > 0) CFD_TRY(sum_finish(c, pq_part, pq_pending, 1, c->scal + CG_OFF_OF(pq)));      // (every exit of the loop has consumed it)
    if (xsum && it > 0) {
        const XsumArgs xa = ring_args(it);
        c->kc[KC_PCG_XSUM] += 1;
        if (c->p_ns_active) {
            if (c->pin_i >= 0)
                hipLaunchKernelGGL(k_ns_shift_sum, dim3(1), dim3(64), 0, st, g, c->pin_i, c->pin_j, c->pin_value, S, (it - 1) & 1, xa, ahist,
                                   x, c->scal + 81);
            else
                CFD_HIP(hipMemsetAsync(c->scal + 81, 0, sizeof(double), st));
            CFD_HIP(hipGetLastError());
            CFD_TRY(all_reduce(c, c->scal + 81, 1, 0));
        }
        c->probe_begin(12);
        hipLaunchKernelGGL(k_cg_xsum, grid, blk, 0, st, g, T, S, (it - 1) & 1, xa, ahist,
                           c->p_ns_active ? c->scal + 81 : (const double *)nullptr, x);
        c->probe_end(12, 8.0 * (xa.n + 1 + xa.from_x));
        CFD_HIP(hipGetLastError());
    } else if (fused && it > 0) {
        if (c->p_ns_active) CFD_TRY(ns_shift_scalar(c, S, (it - 1) & 1, 1, it == 1, p, x));
        c->probe_begin(12);
        hipLaunchKernelGGL(k_cg_xlast, grid, blk, 0, st, g, T, S, (it - 1) & 1, it == 1, p,
                           c->p_ns_active ? c->scal + 81 : (const double *)nullptr, x);
        c->probe_end(12, it == 1 ? 16.0 : 24.0);
        CFD_HIP(hipGetLastError());
    } else {
        if (fused) CFD_HIP(hipMemsetAsync(x, 0, g.plane_elems() * sizeof(double), st));
        if (c->p_ns_active) {
            CFD_TRY(ns_shift_scalar(c, S, 0, 0, 0, x, x));
            hipLaunchKernelGGL(k_add_scalar, grid, blk, 0, st, g, T, c->scal + 81, x);
            CFD_HIP(hipGetLastError());
        }
    }
    *iters = it;
    *relres = bb > 0.0 ? sqrt(rr / bb) : 0.0;
    if (spec_pending) {
        Ctx::SpecPcg &sp = c->spec_pcg;
        double *slot = c->scal + SPEC_PCG_OFF + SPEC_PCG_STRIDE * sp.n;
        CFD_HIP(hipMemcpyAsync(slot, c->scal + RR_HIST_OFF, 8 * sizeof(double), hipMemcpyDeviceToDevice, st));
        CFD_HIP(hipMemcpyAsync(slot + 8, c->scal + CG_OFF_OF(bb), sizeof(double), hipMemcpyDeviceToDevice, st));
        sp.hint[sp.n] = hint0;
        sp.tol[sp.n] = tol;
        sp.n += 1;
        *relres = 0.0;           // known when the iteration's closing read has arrived (spec_verify)
    }
    return CFD_OK;
}

// The stopping tests of a speculative iteration, evaluated on the closing read (Ctx::h_spec = the whole of Ctx::scal).
// *ok = false: a solve had not converged where its schedule ended -- the caller undoes the iteration and repeats it on the
// host-checked path.  Fills in what the solves could not know: residual norms of the statistics, res[1], res[2].
int spec_verify(Ctx *c, double *res, bool *ok) {
    *ok = true;
    const double *h = c->h_spec;
    const Ctx::SpecCheb &sc = c->spec_cheb;
    if (sc.used) {
        for (int q = 0; q < sc.nb; ++q) {
            const double bb = h[SPEC_CHEB_OFF + 2 * sc.nb + q];
            auto conv = [&](double r2, double margin) { return r2 == r2 && bb == bb && sqrt(r2) <= margin * sc.tol * sqrt(bb); };
            // the residual history of this right-hand side in iteration order: launch m measured k0 = K (m - 1) and k0 + K - 1
            double taken = NAN;
            int earliest = -1;
            for (int m = 1; m <= sc.launches; ++m) {
                const int t = sc.t[m - 1][q];
                if (t < 0) continue;
                const int off = m == 1 ? SPEC_CHEB_OFF : SPEC_CHEB_OFF + 8 + 4 * (m - 2), na = sc.na[m - 1];
                const double at[2] = {h[off + t], h[off + na + t]};
                const int kk[2] = {sc.k0[m - 1], sc.kl[m - 1]};
                for (int e = 0; e < 2; ++e) {
                    if (kk[e] == sc.k[q]) taken = at[e];
                    if (earliest < 0 && kk[e] < sc.k[q] && conv(at[e], SPEC_MARGIN)) earliest = kk[e];
                }
            }
            if (!conv(taken, 1.0)) { *ok = false; continue; }
            c->stats.mom_relres[q] = bb > 0.0 ? sqrt(taken / bb) : 0.0;
            if (res) res[1 + q] = sqrt(taken / global_cells(c));
            if (earliest >= 0) c->cheb_plan.retire_k[q] = earliest;      // the next replay leaves at the first iterate that was good enough
        }
    }
    const Ctx::SpecChebT &st = c->spec_chebt;
    for (int s = 0; s < st.n; ++s) {
        const double *slot = h + SPEC_CHEBT_OFF + 8 * s;
        if ((int)slot[0] != 1) {          // a fall-back or an error: the host-checked path takes it
            *ok = false;
            c->chebt_good[st.momentum[s] ? 0 : 1] = false;
            continue;
        }
        if (!st.momentum[s]) continue;
        for (int q = 0; q < st.nb[s]; ++q) {
            c->stats.mom_relres[q] = slot[4 + q] > 0.0 ? sqrt(slot[2 + q] / slot[4 + q]) : 0.0;
            if (res) res[1 + q] = sqrt(slot[2 + q] / global_cells(c));
        }
        c->stats.mom_iters = (int)slot[1];
    }
    const Ctx::SpecPcg &sp = c->spec_pcg;
    for (int s = 0; s < sp.n; ++s) {
        const double *slot = h + SPEC_PCG_OFF + SPEC_PCG_STRIDE * s;
        if (sp.hint[s] < 0) {
            // k_pcg_tail: the device applied the stopping test; not met means the solve ran into maxiter
            const double bb = slot[8], rr = slot[1];
            if (!(bb == bb) || !(rr == rr) || !(sqrt(rr) <= sp.tol[s] * sqrt(bb))) { *ok = false; c->ptail_good = false; continue; }
            c->stats.p_relres = bb > 0.0 ? sqrt(rr / bb) : 0.0;
            c->stats.p_iters = (int)slot[0];
            c->p_iters_hint = (int)slot[0];
            continue;
        }
        const double bb = slot[8], rr = slot[sp.hint[s] & 7];
        if (!(bb == bb) || !(rr == rr) || !(sqrt(rr) <= sp.tol[s] * sqrt(bb))) { *ok = false; continue; }
        c->stats.p_relres = bb > 0.0 ? sqrt(rr / bb) : 0.0;
        if (s == sp.n - 1 && sp.hint[s] >= 2) {
            const double r1 = slot[(sp.hint[s] - 1) & 7];
            // (with a factor to spare, like the Chebyshev schedules: an iterate that meets the test narrowly in one step
            // misses it in the next, and every miss repeats an inner iteration -- a 2048^2 graded channel alternated
            // between 11.7 and 18.5 ms per step for twenty steps at a time; 13.7 -> 12.45 ms per step over 300 steps.
            // Measured and not adopted: the largest count of the last eight solves as the expectation (no repeats, one to
            // two iterations more on every solve: 13.5 ms), one expectation per inner iteration of a step (12.48 ms))
            if (r1 == r1 && sqrt(r1) <= SPEC_MARGIN * sp.tol[s] * sqrt(bb)) c->p_iters_hint = sp.hint[s] - 1;
        }
    }
    return CFD_OK;
}

// BiCGStab with the V-cycle as right preconditioner (non-symmetric frozen operator)
static int bicgstab_mg(Ctx *c, const double *b, double *x, double tol, int maxiter, int *iters, double *relres) {
    const Level &L0 = c->levels[0];
    const OpDev op = level_op(c, L0);
    const Geo g = op.g;
    const Tiling T = tiling(g);
    const dim3 grid(T.blocks), blk = block2d();
    hipStream_t st = c->stream;
    BiScal *S = reinterpret_cast<BiScal *>(c->scal + BI_OFF);
    Vecs<1> r{{c->kry[1]}}, rh{{c->kry[2]}}, p{{c->kry[3]}}, v{{c->kry[4]}}, t{{c->kry[5]}}, xv{{x}};
    double *ph = c->kry[6], *sh = c->kry[7], *w0 = c->kry[8], *w1 = c->kry[9];
    CVecs<1> bc{{b}};
    auto cv = [](const Vecs<1> &a) { return CVecs<1>{{a.p[0]}}; };
    CFD_HIP(hipMemsetAsync(S, 0, sizeof(BiScal), st));
    hipLaunchKernelGGL((k_bi_init<1>), grid, blk, 0, st, g, T, bc, r, rh, xv, c->partials);
    CFD_HIP(hipGetLastError());
    CFD_TRY(finish_to(c, T.blocks, 1, OFF_OF(rho[0][0])));
    double bb, rr;
    CFD_TRY(read_scalars(c, OFF_OF(rho[0][0]), 1, &bb));
    if (!(bb == bb)) { set_error("pressure solve: non-finite right-hand side"); return CFD_E_NUMERIC; }
    *iters = 0;
    *relres = 0.0;
    if (bb == 0.0) return CFD_OK;
    rr = bb;
    Act act{{1, 0}};
    int it = 0;
    while (it < maxiter) {
        const int gen = it & 1;
        hipLaunchKernelGGL((k_bi_p<1>), grid, blk, 0, st, g, T, S, gen, it == 0, act, cv(r), cv(v), p);
        double *out = nullptr;
        int db = 0;
        CFD_TRY(vcycle_impl(c, p.p[0], w0, w1, &out, &db));
        CFD_HIP(hipMemcpyAsync(ph, out, g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
        CFD_TRY(halo_exchange(c, g, 1, ph));
        hipLaunchKernelGGL((k_bi_apply<1, false, 0>), grid, blk, 0, st, op, T, (const double *)nullptr, act, cv(p),
                           CVecs<1>{{ph}}, cv(rh), v, c->partials);
        CFD_HIP(hipGetLastError());
        CFD_TRY(finish_to(c, T.blocks, 1, OFF_OF(sigma[0])));
        hipLaunchKernelGGL((k_bi_s<1>), grid, blk, 0, st, g, T, S, gen, act, cv(v), r);
        CFD_TRY(vcycle_impl(c, r.p[0], w0, w1, &out, &db));
        CFD_HIP(hipMemcpyAsync(sh, out, g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, st));
        CFD_TRY(halo_exchange(c, g, 1, sh));
        hipLaunchKernelGGL((k_bi_apply<1, false, 1>), grid, blk, 0, st, op, T, (const double *)nullptr, act, cv(r),
                           CVecs<1>{{sh}}, cv(r), t, c->partials);
        CFD_HIP(hipGetLastError());
        c->n_red_blocks = T.blocks;
        CFD_TRY(reduce_finish(c, 1, OFF_OF(ts[0])));
        CFD_TRY(sum_finish(c, c->partials + MAX_PARTIALS, T.blocks, 1, c->scal + OFF_OF(tt[0])));
        hipLaunchKernelGGL((k_bi_x<1, false>), grid, blk, 0, st, g, T, S, gen, (const double *)nullptr, act, cv(p),
                           CVecs<1>{{ph}}, CVecs<1>{{sh}}, cv(t), cv(rh), r, xv, c->partials);
        CFD_HIP(hipGetLastError());
        c->n_red_blocks = T.blocks;
        CFD_TRY(reduce_finish(c, 1, OFF_OF(rho[gen ^ 1][0])));
        CFD_TRY(sum_finish(c, c->partials + MAX_PARTIALS, T.blocks, 1, c->scal + OFF_OF(rr[0])));
        ++it;
        CFD_TRY(read_scalars(c, OFF_OF(rr[0]), 1, &rr));
        if (!(rr == rr)) { set_error("MG-BiCGStab: non-finite residual at iteration %d", it); return CFD_E_NUMERIC; }
        if (sqrt(rr) <= tol * sqrt(bb)) break;
    }
    *iters = it;
    *relres = sqrt(rr / bb);
    return CFD_OK;
}

int pressure_solve(Ctx *c, const double *b, double *x, double tol, int maxiter, int *iters, double *relres) {
    if (!c->p_built) { set_error("pressure operator not built"); return CFD_E_STATE; }
    c->stats.p_solver = c->p_symmetric ? 0 : 1;
    const double accept0 = tol > ROUNDING_LEVEL_RELRES ? tol : ROUNDING_LEVEL_RELRES;
    // Seam copies that differ by more than rounding but by less than p_sym_tol (the reference's cell-centroid formula
    // loses digits on thin cells: 1e-6 on a wall-graded 8192^2 mesh): PCG first -- its residual recurrence uses the
    // operator as stored, so what it converges to is the solution of the stored system, in one V-cycle per iteration
    // where BiCGStab needs two -- and BiCGStab from the saved right-hand side if PCG does not get there (PCG works in
    // the right-hand side plane).
    const bool nearly = c->p_symmetric && c->p_seam_asym > 1.0e-9;
    if (nearly) {
        CFD_HIP(hipMemcpyAsync(c->kry[10], b, c->g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        const int cap = maxiter < 120 ? maxiter : 120;
        // a breakdown of PCG on the slightly non-symmetric stored operator (non-finite (r, r)) is a reason to fall back
        // as well, not an error of the run: before this regime existed these operators went straight to MG-BiCGStab
        const int rc_pcg = pcg_mg(c, b, x, tol, cap, iters, relres);
        if (rc_pcg != CFD_OK && rc_pcg != CFD_E_NUMERIC) return rc_pcg;
        if (rc_pcg == CFD_E_NUMERIC || (*iters >= cap && !(*relres <= accept0))) {
            CFD_HIP(hipMemcpyAsync(const_cast<double *>(b), c->kry[10], c->g.plane_elems() * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            c->stats.p_solver = 1;
            c->p_iters_hint = 0;
            CFD_TRY(bicgstab_mg(c, b, x, tol, maxiter, iters, relres));
        } else if (*iters >= cap && cap < maxiter) {
            *iters = cap;        // converged at the cap: not a failure
        }
    } else {
        CFD_TRY(c->p_symmetric ? pcg_mg(c, b, x, tol, maxiter, iters, relres) : bicgstab_mg(c, b, x, tol, maxiter, iters, relres));
    }
    // The reference raises when its pressure solver gives up (solver.py:842-844); its cached-pyamg branch returns
    // whatever maxiter V-cycles reached (solver.py:829-833).  An unconverged p' is never silent here: an error when
    // the operator is rebuilt every solve, the p_unconverged count of cfd_get_stats (and a warning of the host
    // driver) under the frozen operator.
    const double accept = tol > ROUNDING_LEVEL_RELRES ? tol : ROUNDING_LEVEL_RELRES;
    if (*iters >= maxiter && !(*relres <= accept)) {
        c->stats.p_unconverged += 1;
        if (!c->freeze_pressure) {
            set_error("Pressure solver did not converge: |r|/|b| = %.3e after %d iterations (tol %.1e)", *relres, *iters, tol);
            return CFD_E_NOCONV;
        }
    }
    return CFD_OK;
}


// ------------------------------------------------------------------ kernel timing (bench.py)
int time_kernel(Ctx *c, int kernel, int reps, double *avg_us) {
    if (reps < 1) reps = 1;
    hipStream_t st = c->stream;
    hipEvent_t e0, e1;
    CFD_HIP(hipEventCreate(&e0));
    CFD_HIP(hipEventCreate(&e1));
    const Geo g = c->g;
    const Tiling T = tiling(g);
    const dim3 grid(T.blocks), blk = block2d();
    auto body = [&](int n) -> int {
        for (int it = 0; it < n; ++it) {
            switch (kernel) {
                case 0: {
                    if (!c->p_built) { set_error("pressure operator not built"); return CFD_E_STATE; }
                    const OpDev op = level_op(c, c->levels[0]);
                    hipLaunchKernelGGL(k_cg_q, grid, blk, 0, st, op, T, c->kry[2], c->kry[3], c->partials);
                    break;
                }
                case 1: {
                    if (!c->p_built) { set_error("pressure operator not built"); return CFD_E_STATE; }
                    const OpDev op = level_op(c, c->levels[0]);
                    if (c->levels.size() > 1 && level_fusable(c, 0, c->levels[0].g, coarse_view(c, 1))) {
                        const MarchGeo mgeo = march_geo(g);
                        hipLaunchKernelGGL((k_mg_up<true>), dim3(march_blocks(mgeo)), dim3(256), 0, st, op, mgeo, c->p_diag, c->kry[1],
                                           c->kry[4], c->levels[1].g, c->levels[1].x, c->mg_gamma, c->om_at(0), c->om_at(1), c->kry[5], c->partials);
                    } else {
                        hipLaunchKernelGGL(k_mg_jacobi, grid, blk, 0, st, op, T, c->kry[1], c->kry[4], c->om_at(0), c->kry[5]);
                    }
                    break;
                }
                case 2:
                    CFD_TRY(launch_momentum_rhs(c, c->dt_ap > 0 ? c->dt_ap : 1.0, c->U(), c->V(), c->P()));
                    break;
                case 3: {
                    const OpDev op{c->ai, c->aj, c->ap, g, c->per_i, c->per_j};
                    Act act{{1, 1}};
                    CVecs<2> x{{c->kry[5], c->kry[6]}}, w{{c->kry[3], c->kry[4]}};
                    Vecs<2> y{{c->kry[7], c->kry[8]}};
                    hipLaunchKernelGGL((k_bi_apply<2, true, 0>), grid, blk, 0, st, op, T, c->kry[0], act, x, x, w, y,
                                       c->partials);
                    break;
                }
                case 4: {
                    if (!c->p_built || c->levels.size() < 2) { set_error("no coarse level"); return CFD_E_STATE; }
                    const Level &L = c->levels[0], &C = c->levels[1];
                    int fi, fj;
                    factors(L.g, C.g, fi, fj);
                    const Tiling Tc = tiling(C.g);
                    hipLaunchKernelGGL(k_mg_restrict, dim3(Tc.blocks), blk, 0, st, level_op(c, L), c->kry[1], c->kry[4],
                                       C.g, Tc, fi, fj, C.b);
                    break;
                }
                case 5: {
                    double dummy;
                    CFD_TRY(launch_continuity(c, c->U(), c->V(), c->P(), &dummy));
                    break;
                }
                default:
                    set_error("unknown kernel id %d", kernel);
                    return CFD_E_ARG;
            }
        }
        return CFD_OK;
    };
    CFD_TRY(body(2));
    CFD_HIP(hipStreamSynchronize(st));
    CFD_HIP(hipEventRecord(e0, st));
    CFD_TRY(body(reps));
    CFD_HIP(hipEventRecord(e1, st));
    CFD_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    CFD_HIP(hipEventElapsedTime(&ms, e0, e1));
    CFD_HIP(hipGetLastError());
    *avg_us = 1000.0 * ms / reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return CFD_OK;
}

}  // namespace cfd
