// Internal declarations shared by the HIP translation units of libcfd_hip.so.
// gfx950 only.  Device layout: planes [row][col], col contiguous, NG = CFD_GHOST_ROWS ghost rows below and
// above the interior rows and CO = CFD_COL_OFFSET spare columns on the left: interior cell (i, j) lives at
// (j + NG) * pitch + i + CO (Geo::cell), so that cell i = 0 starts a 128-byte line; element (ig, jg) of the
// reference's ghosted (Ni+2) x (Nj+2) array is Geo::at(ig, jg) = cell(ig - 1, jg - 1).  Face planes share
// the geometry: i-face f (west face of cell f) sits at the index of cell f, j-face g (south face) at cell g.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cfd_hip.h"

namespace cfd {

constexpr double TINY = 1.0e-14;
constexpr double K_FLOOR = 1.0e-15;      // turbulence.py:28
constexpr double OMEGA_FLOOR = 1.0e-6;   // turbulence.py:29
constexpr int MAX_PARTIALS = 4096;
constexpr int MAX_LEVELS = 24;

// ghost rows below and above the interior of every plane: the inner one is the reference's ghost layer, the
// others are halo rows of the fused kernels (three smoothing sweeps, four Chebyshev iterations per pass) and of
// the j-slab exchange.  Plane row r holds j = r - NG.
constexpr int NG = CFD_GHOST_ROWS;
// Interior cell i = 0 sits CO doubles into its plane row: one full 128-byte line, so that a 64-lane row of a tile
// kernel (columns 64 t .. 64 t + 63) covers exactly four lines instead of straddling five (the fabric-side
// FETCH_SIZE of the pointwise kernels was 1.19 - 1.25 times their algorithmic reads with the former offset of 2).
#ifndef CFD_CO
#define CFD_CO CFD_COL_OFFSET
#endif
constexpr int CO = CFD_CO;

struct Geo {
    int ni, nj, pitch;
    // j-slab decomposition: cutlo / cuthi = 1 when the low / high j edge of this slab is a cut
    // to a neighbour rank (its ghost rows then hold the neighbour's interior rows, filled by
    // the halo exchange) instead of a physical boundary.
    int cutlo, cuthi;
    __host__ __device__ inline bool blo(int j) const { return j == 0 && !cutlo; }          // physical low-j boundary row
    __host__ __device__ inline bool bhi(int j) const { return j == nj - 1 && !cuthi; }     // physical high-j boundary row
    // NG ghost rows below and above the interior: the innermost one is the reference's ghost layer, the
    // others are halo rows of the fused kernels / of the j-slab exchange.  ghosted index jg = j + 1.
    __host__ __device__ inline size_t at(int ig, int jg) const { return (size_t)(jg + NG - 1) * pitch + ig + CO - 1; }
    __host__ __device__ inline size_t cell(int i, int j) const { return (size_t)(j + NG) * pitch + i + CO; }
    __host__ inline size_t plane_elems() const { return (size_t)(nj + 2 * NG) * pitch; }
};

inline int pitch_for(int ni) { return ((ni + CO + 2) + 31) / 32 * 32; }

struct MeshDev {
    const double *vol, *wri, *wnj, *nix, *niy, *njx, *njy, *si, *sj, *di, *dj, *lmin;
};

// 5-point operator on masked or unmasked face planes.
// (A x)_P = D x_P - ci[W] x_W - ci[E] x_E - cj[S] x_S - cj[N] x_N ; boundary couplings are
// dropped unless the axis is periodic (then the x index wraps).
struct OpDev {
    const double *ci, *cj, *D;
    Geo g;
    int per_i, per_j;
    // single-precision copies of the face planes: read by the fused V-cycle kernels only (the preconditioner of the
    // pressure Krylov solve stays a fixed symmetric operator; the Krylov operator itself is ci / cj); null: none
    const float *cif, *cjf;
};

// finest pressure level: diagonal recomputed from the face planes (linalg.hip, diag_from_faces);
// (pi, pj) = reference cell in this slab's logical coordinates (may lie outside it), c* = its couplings
struct DiagFix {
    int on = 0, pi = -1, pj = -1;
    double cW = 0.0, cE = 0.0, cS = 0.0, cN = 0.0;     // the neighbours' (zeroed) couplings to the reference cell
    double oW = 0.0, oE = 0.0, oS = 0.0, oN = 0.0;     // the reference cell's own (zeroed) face values
};

struct FlowBCSide {
    const int32_t *kind_uv, *kind_p;
    const double *ua, *ub, *va, *vb, *pa, *pb, *nx, *ny;
    int n;
};

struct FlowBCDev {
    FlowBCSide side[4];
};

void set_error(const char *fmt, ...);

// host-side cost counters (cfd_host_counters): kernel launches and blocking stream synchronisations issued by the
// library in this process -- the two quantities that bound the step time of small meshes and of thin j-slabs
// (process wide and shared by every context and host thread, hence atomic; the phase a thread is in is its own)
extern std::atomic<long long> g_launches, g_syncs, g_syncs_phase[9];      // [phase id of api.hip's Phase, 8 = outside any phase]
extern thread_local int g_phase;
inline void count_sync() {
    g_syncs.fetch_add(1, std::memory_order_relaxed);
    g_syncs_phase[g_phase].fetch_add(1, std::memory_order_relaxed);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                    \
    do {                                                                               \
        cfd::g_launches.fetch_add(1, std::memory_order_relaxed);                       \
        kernel<<<(grid), (block), (shmem), (stream)>>>(__VA_ARGS__);                   \
    } while (0)
#define hipStreamSynchronize(stream) (cfd::count_sync(), (hipStreamSynchronize)(stream))

#define CFD_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            cfd::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return CFD_E_HIP;                                                                 \
        }                                                                                     \
    } while (0)

#define CFD_TRY(expr)                 \
    do {                              \
        int rc__ = (expr);            \
        if (rc__ != CFD_OK) return rc__; \
    } while (0)

// lane i <- lane i-1 / lane i+1 through DPP wave shifts (VALU, no LDS round trip)
__device__ __forceinline__ double lane_left(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_right(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Workgroups are dealt to the 8 XCDs round-robin by their index (block b runs on XCD b % 8), and each XCD has its own
// L2.  Neighbouring column strips share halo lanes, neighbouring row segments share halo rows: with this map the
// blocks of one XCD own a contiguous run of work items, so the lines two neighbours share are fetched into one L2
// instead of two.  nb (the grid) must be a multiple of 8.  (Measured slower than the plain map, see Ctx::xcd_map.)
__device__ __forceinline__ int xcd_block(int b, int nb, int on) {
    return on ? (b & 7) * (nb >> 3) + (b >> 3) : b;
}

// Work items of a marching kernel.  WPB = 4: every wave takes (strip, segment) items on its own.  WPB = 8 / 16, the
// lockstep form: a workgroup takes WPB neighbouring strips of ONE segment and its waves meet at a barrier after every
// row step, so that the workgroup reads WPB x 512 contiguous bytes of each plane row at a time (the strips' 64-lane
// windows are not aligned to the 128-byte lines; waves drifting apart made each of them pay for its two partial lines:
// tools/stream_probe3.hip, 4.1 TB/s for misaligned independent waves, 4.8 - 4.9 in lockstep).  Every wave of a lockstep
// workgroup has the same segment, hence the same trip counts; a wave beyond the last strip repeats the last strip
// without output, so that it reaches the same barriers.
template <int WPB, bool LOCK = (WPB > 4)>
struct MarchItems {
    int first, stride, count, ngroups;
    __device__ __forceinline__ MarchItems(int nstrips, int nsegs, int xcd, int wave) {
        ngroups = (nstrips + WPB - 1) / WPB;
        if (LOCK) { first = blockIdx.x; stride = gridDim.x; count = ngroups * nsegs; }
        else { first = xcd_block(blockIdx.x, gridDim.x, xcd) * 4 + wave; stride = gridDim.x * 4; count = nstrips * nsegs; }
    }
    __device__ __forceinline__ void decode(int item, int nstrips, int wave, int &strip, int &seg, bool &active) const {
        if (LOCK) {
            const int raw = (item % ngroups) * WPB + wave;
            active = raw < nstrips;
            strip = active ? raw : nstrips - 1;
            seg = item / ngroups;
        } else {
            active = true;
            strip = item % nstrips;
            seg = item / nstrips;
        }
    }
};
// sum of the per-wave values of a workgroup of WPB waves (WPB a multiple of 4), fixed order
template <int WPB>
__device__ __forceinline__ double sum_waves(const double *sm) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < WPB; q += 4) tot += (sm[q] + sm[q + 1]) + (sm[q + 2] + sm[q + 3]);
    return tot;
}

// Streaming store: planes larger than the 256 MiB Infinity Cache are written once and read back only after
// everything else has passed through the cache, so their stores carry the nontemporal hint (tools/stream_probe2.hip:
// +5-8 % on the marching access pattern with plain loads; nontemporal LOADS lose, they bypass the L1 reuse of
// neighbouring lanes / rows).  `nt` is wave-uniform (a kernel argument).
__device__ __forceinline__ void store_stream(double *p, double v, int nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ void store_stream_f32(float *p, float v, int nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// kernel families counted by Ctx::kc (cfd_kernel_counts): which production kernels a run actually took
enum {
    KC_CHEB4_2RHS = 0, KC_CHEB4_1RHS, KC_CHEB2, KC_CHEB1, KC_BICGSTAB_IT, KC_FLUX_MARCH, KC_FLUX_MARCH_MULTISTRIP,
    KC_FLUX_TILE, KC_MG_FUSED_LEVELS_MAX, KC_MG_DEFERRED_RESID, KC_CG_PQ, KC_VCYCLE, KC_MG_LINE_LEVELS_MAX, KC_MG_TAIL_LEVELS, KC_CORRECT_FLUX, KC_MG_SEMI_LEVELS, KC_SPEC_ITERS, KC_SPEC_REDONE, KC_PCG_TAIL, KC_CHEB_TAIL,
    KC_CHEB6_1RHS, KC_MG_FLOAT_LEVELS_MAX, KC_MESH_FORM, KC_PCG_XSUM, KC_COUNT = 24
};

// ------------------------------------------------------------------ multigrid level
struct Level {
    Geo g;
    double *ci, *cj, *D;     // operator planes (masked)
    double *x, *b, *t;       // solution, right-hand side, scratch (Jacobi ping-pong)
    // j-slab runs only, first replicated ("agglomerated") level: this rank's slab of the level
    // (geometry gs, rows [joff, joff + gs.nj) of the global level) and three slab staging planes
    // that the distributed level above writes before the all-gather into the global planes.
    Geo gs{};
    int joff = 0;
    double *s0 = nullptr, *s1 = nullptr, *s2 = nullptr;
    // alternating line smoother (levels whose face couplings are strongly anisotropic): factors of the tridiagonal
    // j-blocks (Thomas multipliers and inverse pivots), computed when the hierarchy is built
    bool line = false;
    int line_dirs = 0;            // bit 0: j-lines, bit 1: i-lines
    float *cif = nullptr, *cjf = nullptr;         // see OpDev
    double *lm = nullptr, *linv = nullptr;        // j-blocks
    double *lmi = nullptr, *linvi = nullptr;      // i-blocks
    double aniso = 1.0;           // max over the cells of strong / weak coupling (diagnostic, cfd_level_aniso)
};

// Semi-coarsened line multigrid (stretched meshes, semi.inc): level l holds ni x (nj >> l) cells; whole i-lines are
// solved exactly (zebra line Gauss-Seidel over the rows), j is the only direction that is coarsened.
struct SemiLevel {
    Geo g;
    double *ci, *cj, *D;          // operator (level 0: the planes of the finest pressure level)
    double *x, *b, *x2;           // levels >= 1 (x2: the even rows of the way down when the way up is the fused pass)
    float *lmf, *linvf, *cjf;     // single precision: Thomas factors of the whole i-lines (multiplier, inverse pivot), c_j
    float *zf;                    // periodic lines (Sherman-Morrison): T'^-1 u per cell (null: lines are not periodic)
    double *rs;                   // ... and per row {beta / gamma, 1 / (1 + v . z)}
};

// V-cycle tail: the levels small enough to live in the LDS of one workgroup are cycled by ONE launch (k_mg_tail)
constexpr int TAIL_NT = 1024;                       // threads of that workgroup
constexpr int TAIL_CPT = 4;                         // cells per thread on the largest tail level
constexpr int TAIL_MAX_CELLS = TAIL_NT * TAIL_CPT;
constexpr int TAIL_MAX_LEVELS = 12;
constexpr int TAIL_LDS_DOUBLES = 20480;             // 160 KiB, what one gfx950 workgroup may declare
// LDS doubles of a tail: x and b of every level, a scratch plane of the largest, six coupling planes and two int planes
// of the cached levels
constexpr size_t TAIL_TABLE_BYTES = (size_t)(TAIL_MAX_CELLS + 2 * TAIL_NT) * 64;      // k_tail_pack's table: 60 bytes per tail cell (the workspace holds two)
inline long tail_lds_doubles(long total, long top, long ncache) { return 2 * total + top + 7 * ncache; }
// ... of a whole solve (k_pcg_tail): p, the diagonal and x of the finest level, reduction slots
inline long pcg_tail_lds_doubles(long total, long top, long ncache) { return tail_lds_doubles(total, top, ncache) + 3 * top + 40; }
struct TailLevel {
    const double *ci, *cj, *D;
    Geo g;
    int fi, fj;          // coarsening factors towards the next level
    int off;             // offset of this level's x and b inside the LDS arrays
    int coff;            // levels 2 ..: offset of this level's couplings inside the LDS cache
};
struct TailArgs {
    TailLevel L[TAIL_MAX_LEVELS];
    int n, total;        // levels, sum of their cells
    int ncache;          // cells of levels 2 .. (their couplings are cached in LDS)
    const double *tab;   // the couplings of every tail cell, packed by k_tail_pack
    int per_i, per_j, pre, post, coarse_sweeps, nom;
    double om[4], om_coarse, gamma;
};
// what the cycle itself needs of a tail, small enough to arrive with the first lines of the kernel arguments
// (the cycle reads no plane: the couplings come packed from TailArgs::tab)
struct TailDevLevel {
    int ni, nj, off, coff;
};
struct TailDev {
    int n, total, ncache;
    int per_i, per_j, pre, post, coarse_sweeps, nom;
    double om[4], om_coarse, gamma;
    const double *tab;
    Geo g0;              // plane geometry of the first tail level (right-hand side in, result out)
    TailDevLevel L[TAIL_MAX_LEVELS];
};

// j-slab communication: callbacks supplied by the host (torch.distributed over RCCL in
// production), all of them ordered on Ctx::stream.
struct Comm {
    int rank = 0, nranks = 1;
    int nj_global = 0, j_offset = 0;
    cfd_halo_fn halo = nullptr;
    cfd_allreduce_fn allreduce = nullptr;
    cfd_allgather_fn allgather = nullptr;
    void *user = nullptr;
    void *rccl = nullptr;         // ncclComm_t of the native transport (comm.hip); takes precedence over the callbacks
    long long n_halo = 0, n_reduce = 0, n_gather = 0;
    int phase = 8;                // phase of the outer step the collectives are booked under (8: outside any phase)
    long long by_phase[9][3] = {};
};

// ------------------------------------------------------------------ context
struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Geo g{};
    char *ws = nullptr;          // workspace base
    size_t ws_bytes = 0, ws_used = 0;
    bool own_ws = false;

    // mesh planes
    double *m_vol, *m_wri, *m_wnj, *m_nix, *m_niy, *m_njx, *m_njy, *m_si, *m_sj, *m_di, *m_dj, *m_lmin;
    bool mesh_set = false;

    // field planes.  cur = 0/1 selects which of the two (u, v, p) sets is "current";
    // the other one is the scratch that receives the momentum solution.
    double *u[2], *v[2], *p[2];
    int cur = 0;
    double *su, *sv, *sp;        // explicit snapshot planes (PISO, fine-grained API)
    double *k, *omega, *nut, *pcorr;
    double *ap, *ai, *aj;        // momentum diagonal and face diffusion conductances
    double *tp, *ti, *tj;        // same for the k/omega transport (nu + 0.5 nu_t)
    double *bu, *bv;             // right-hand sides (momentum / k, omega)
    double *stage_a, *stage_b;   // staging planes for host <-> device transposes
    double *kry[16];             // Krylov work planes
    int n_kry = 16;

    // pressure operator (finest level = levels[0])
    std::vector<Level> levels;
    int l_agg = 0;               // j-slab runs: first replicated level (levels below it are slabs); 0 = single rank
    Comm comm;
    bool p_built = false;
    int p_iters_hint = 0;         // PCG iterations of the previous solve with the current operator (0: unknown)
    bool p_symmetric = true;
    // Pure-Neumann / periodic pressure systems (no Dirichlet p anywhere): the reference pins one cell
    // (solver.py:1033-1044).  The detached cell costs the V-cycle-preconditioned CG two of its ten iterations
    // at 8192^2, so the symmetric case solves the singular system instead (all rows kept, right-hand side
    // made compatible at the reference cell's own row, which is the row the reference drops) and shifts
    // the solution to the pinned value afterwards: the same p' up to rounding.  p_ns_active is decided when
    // the operator is built; non-symmetric operators (BiCGStab) keep the eliminated cell.
    int p_nullspace = 1;
    bool p_ns_active = false;
    DiagFix p_diag;
    int recompute_diag = 1;
    int pin_i_any = -1, pin_j_rel = -1;   // reference cell as given (slab runs: row relative to this slab)

    // reductions
    double *partials = nullptr;  // [q][MAX_PARTIALS]
    double *scal = nullptr;      // device scalars
    double *h_scal = nullptr;    // pinned host mirror
    int n_red_blocks = 0;

    // BC tables
    char *bc_buf = nullptr;      // device buffer for flow tables
    FlowBCDev fbc{};
    bool fbc_set[4] = {false, false, false, false};
    int fbc_class[4] = {0, 0, 0, 0};   // 1: every entry periodic (u, v and p); 2: every entry AFFINE / SYMMETRY (u, v) and AFFINE (p); 0: anything else
    int cf_form = 2;                   // k_correct_flux on meshes of >= 16 strips: 0 independent waves, 1 lockstep workgroups of 8 strips, 2 of 4
    double p_sym_tol = 1.0e-5;         // relative seam asymmetry of the frozen pressure operator up to which it is solved by PCG
    double p_seam_asym = 0.0;          // the measured one (diagnostic)
    int fuse_correct_slab = 1;         // ... also on j-slab contexts (every rank must hold the same value)
    // Speculative inner iteration (option "speculate"; needs host_lag and the marching flux kernels): the
    // Chebyshev momentum solve and the MG-PCG pressure solve replay the schedule of the previous iteration (which
    // right-hand side leaves after which iteration; how many PCG iterations) WITHOUT host reads, leaving the residual
    // sums their stopping tests would have looked at in device slots; the one read that ends the iteration (continuity
    // sum, time-step minima) brings those slots along and the stopping tests are evaluated then.  A test that fails
    // means a solve needed more iterations than last time: the iteration is undone (its inputs are still in the other
    // half of the ping-pong planes) and repeated on the host-checked path, which also renews the schedule.  Every
    // SPEC_RENEW-th iteration runs host-checked anyway so that the schedule can shrink.
    int speculate = -1;                // -1: on for contexts of <= SPEC_AUTO_CELLS cells (j-slabs: SPEC_AUTO_CELLS_SLAB), 0 / 1: off / on
    static constexpr long SPEC_AUTO_CELLS = 1L << 22, SPEC_AUTO_CELLS_SLAB = 1L << 24;     // cells of this context (a slab: its own rows)
    bool spec_on = false;              // inside a speculative iteration
    bool spec_wanted = false;          // this context runs its inner iterations speculatively where it can (run_iteration)
    int spec_age = 0;                  // speculative iterations since the last host-checked one
    static constexpr int SPEC_RENEW = 48;
    double *h_spec = nullptr;          // pinned host copy of Ctx::scal (256 doubles), filled by the closing read
    struct ChebPlan {
        bool valid = false;
        const double *D = nullptr;     // the operator the schedule belongs to
        int nb = 0, K = 0;
        double tol = 0.0;
        int retire_k[2] = {0, 0};      // iterate each right-hand side was taken from
        unsigned char kseq[12] = {};   // iterations fused by each launch of that solve (4, or 6: chebyshev_fused)
    } cheb_plan;
    static constexpr int SPEC_MAX_LAUNCHES = 12;
    struct SpecCheb {
        bool used = false;
        int nb = 0, K = 0, launches = 0;
        double tol = 0.0;
        int na[SPEC_MAX_LAUNCHES] = {};            // right-hand sides in launch m (0-based)
        int k0[SPEC_MAX_LAUNCHES] = {}, kl[SPEC_MAX_LAUNCHES] = {};     // first and last iterate whose residual launch m summed
        int t[SPEC_MAX_LAUNCHES][2] = {};          // position of right-hand side q among them, -1: not in the launch
        int k[2] = {0, 0};                         // iterate each right-hand side was taken from
    } spec_cheb;
    struct SpecPcg {
        int n = 0;
        int hint[4] = {0, 0, 0, 0};
        double tol[4] = {0, 0, 0, 0};
    } spec_pcg;
    static constexpr int SPEC_CHEBT_MAX = 3;
    struct SpecChebT {                 // one-launch Chebyshev solves (k_cheb_tail) of a speculative iteration
        int n = 0;
        int nb[SPEC_CHEBT_MAX] = {};
        bool momentum[SPEC_CHEBT_MAX] = {};
    } spec_chebt;
    bool chebt_good[2] = {false, false};   // the last host-checked one-launch solve (momentum / other) ended "solved"
    bool ptail_good = false;               // the last host-checked one-launch pressure solve converged
    int cheb_tail = 1;                 // 1: transport solves of meshes of at most cheb_tail_cells cells are one launch (k_cheb_tail)
    int cheb_tail_cells = 2800;        // (the size up to which the pressure solve is one launch as well)
    int cf_agreed = -1;                // slab ranks' common answer to correct_flux_ok (-1: not asked yet)
    bool bc_fresh = false;             // the ghosts of u, v, p [bc_cur] are those of k_flow_bc + exchange on the current values
    int bc_cur = -1, skip_fresh_bc = 1;
    bool mesh_halo2 = false;           // metric planes hold the neighbour slab's rows beyond the cuts (launch_correct_flux)
    long mg_c32_min_cells = 1L << 18;  // ... from this many cells of the global level on
    int mg_c32 = 2;                    // fused V-cycle: this many coarse levels (0 .. 2, from the first) keep b and x in single precision (vcycle_impl: nf32)
    // diffusive time-step minimum of a laminar run, kept once the host has read it (stencil.hip: dt_diff_known)
    int lmin_const = 1;
    bool dtdiff_valid = false, dt_pass_diff_fresh = false;
    double dtdiff_cfl = 0.0, dtdiff_nu = 0.0, dtdiff_val = 0.0;
    int fold_sums = -1;                // fused MG-PCG on one rank: (r, z) and (p, q) summed inside the kernels that consume them (-1: meshes of <= 4M cells)
    int cheb_k6 = 1;                   // Chebyshev launches with one right-hand side left fuse six iterations (single rank, lockstep form)
    int rn_opt = 2, rn_state = -1;     // axis-aligned mesh (stencil.hip: rect_normals): option (cap); -1 not looked at yet / 0 / 1 / 2
    int pcg_xsum = 1;                  // fused MG-PCG: x summed from the kept directions when the solve ends (k_cg_xsum) instead of updated every step
    int fuse_correct = 1;              // correction + boundary fill + closing flux pass in one marching kernel where correct_flux_ok()
    int per_i = 0, per_j = 0, wrap_p_i = 0, wrap_p_j = 0;
    int32_t *pmask[4] = {nullptr, nullptr, nullptr, nullptr};
    int pp_per_i = 0, pp_per_j = 0, has_dirichlet = 0, pin_i = 0, pin_j = 0;
    double pin_value = 0.0;
    bool pbc_set = false;
    std::vector<cfd_turb_op> turb_ops;
    double *turb_vals = nullptr;
    int n_turb_vals = 0;

    // parameters
    double nu = 1e-3, dpdx = 0, dpdy = 0;
    int rhie_chow = 1, turbulent = 0;
    double mom_tol = 1e-8, p_tol = 1e-8, turb_tol = 1e-8;
    int mom_maxiter = 200, p_maxiter = 200, turb_maxiter = 200, freeze_pressure = 1;
    int mg_pre = 3, mg_post = 3, mg_coarse_sweeps = 40, mg_min = 4;
    double mg_gamma = 1.0;       // scaling of the prolongated coarse correction
    // Jacobi damping of the smoothing sweeps, used cyclically (sweep s takes mg_om[s % mg_nom]); the sweeps of
    // a pass commute, so the preconditioner is symmetric for any list.  Single-rank default: V(3,3) with
    // (1.25, 0.5, 1.0) -- 5 PCG iterations per solve at 8192^2, also once the perturbations of the bench workload have
    // grown (from outer step 17 on, where (1.0, 0.55, 0.9) stops 20 % above the tolerance after five and takes a sixth);
    // (1.0, 0.6) with V(2,2) needs 7, (0.8, 0.8) 7.5.  j-slab contexts run the same kernels (a cut carries NG = 4 halo rows).
    double mg_om[4] = {1.25, 0.5, 1.0, 0.0};
    int mg_nom = 3;
    inline double om_at(int s) const { return mg_om[s % mg_nom]; }
    inline double om_mean() const { double a = 0.0; for (int q = 0; q < mg_nom; ++q) a += mg_om[q]; return a / mg_nom; }
    int warm_start = 1;          // Krylov initial guess = previous field (reference: zero)
    int use_chebyshev = 1;       // Chebyshev semi-iteration for time-step dominated transport systems
    double cheb_rho_max = 0.95;  // ... i.e. Jacobi row-sum bound below this
    int cheb_stages = 4;         // Chebyshev iterations fused per launch (2 or 4)
    int cheb_prefetch = 2;       // rows of loads a k_chebk wave keeps in flight (1 or 2)
    int nt_stores = -1;          // nontemporal stores in the streaming kernels: -1 = when a plane exceeds the Infinity Cache
    int march_ls = 0;            // rows per segment of the K >= 3 marching kernels (0 = default)
    int march_ls_model = 1;      // 1: multigrid levels of few wave rounds take the segment length of march_geo_k's cost model
    long long kc[KC_COUNT] = {};  // launches per kernel family (cfd_kernel_counts)
    inline int nts(const Geo &geo) const {
        return nt_stores >= 0 ? nt_stores : (geo.plane_elems() * sizeof(double) > ((size_t)256 << 20) ? 1 : 0);
    }
    inline int nts() const { return nts(g); }
    int cheb_max_iters = 80;     // ... and predicted iteration count below this; else BiCGStab
    int stats_last_solver = 0;   // 1 = the last transport solve was Chebyshev
    int mg_fused = 1;            // fused marching V-cycle kernels where applicable
    int mg_line = -1;            // j-line (tridiagonal) Jacobi smoother: -1 = on levels whose c_j / c_i anisotropy asks for it
    // time-step minima of the next outer step evaluated by the pass that closes an inner iteration (laminar runs)
    int dt_lookahead = 1;
    bool dt_ahead = false, dt_ahead_diff = false, dtp_set = false;
    int dt_ahead_cur = 0, dtp_use_cfl = 0, dtp_use_diff = 0;
    double dtp_cfl = 0.0, dtp_cfl_diff = 0.0;
    int semi_resid = 1;          // ... and on semi-coarsened hierarchies by the two kernels of the finest level's way down
    int fuse_resid = 1;          // PCG: r = r - alpha q formed by the next V-cycle's pre-smoother on unchecked iterations
    int fuse_nut = 1;            // k-omega: eddy viscosity evaluated inside the conductance pass (k_komega_coeffs)
    int march_flux = 1;          // marching form of the Rhie-Chow flux kernels (momentum / pressure RHS, continuity)
    // momentum right-hand sides computed ahead of time by the fused continuity pass of the previous
    // inner iteration (valid while the fields, a_P and dt are the ones it saw)
    int rhs_lookahead = 1;
    bool rhs_ahead = false;
    double rhs_dt = -1.0;
    int rhs_cur = -1;
    double k_clip = 1e-10, omega_clip = 1e-10;
    // Menter SST (turbulence.py:41-208, 343-355, 447-534); planes live in one allocation of their own
    // (sst_mem, made by cfd_set_turbulence_model) so that k-omega and laminar runs do not pay for them
    int turb_model = 0;           // CFD_TURB_KOMEGA / CFD_TURB_SST
    char *sst_mem = nullptr;
    char *line_mem = nullptr;      // Thomas factor planes of the line-smoothed multigrid levels (own allocation, made when a
                                   // hierarchy with such levels is built: stretched meshes only; cfd_extra_device_bytes)
    size_t line_mem_bytes = 0;
    char *f32_mem = nullptr;       // single-precision face planes of the fused multigrid levels (part of the workspace)
    size_t f32_mem_bytes = 0;
    int rr_byproduct = 1;          // (r, r) of the iteration before the expected last one from the next pre-smoother
    int mg_v32_emul = 0;           // measurement aid: the fused V-cycle kernels round what they store (x, coarse right-hand side, correction) to single precision
    int mg_f32 = 1;                // fused V-cycle levels read their face coefficients in single precision
    int scalar_nu = 0;           // 1: conductances from the scalar nu alone (function-level callers, solver.py:205-207)
    // stretched meshes: zebra whole-line relaxation along i + semi-coarsening along j (semi.inc) in place of the
    // alternating block line smoother; chosen when the finest level's coupling ratio exceeds mg_line_threshold
    // ... per solver.  The fused Chebyshev form stays at 16: it lets a right-hand side leave at the first iterate that meets
    // the stopping test, the one-iteration-per-launch form runs to the count its bound predicts (and one more); at the
    // reference's own rtol of 1e-8 the latter reproduces every printed digit of the committed 5 x 20 friction logs, the
    // former lands 2e-6 (k-omega) to 5e-6 (SST) away -- for 0.2 ms of a 2.6 ms step.
    int fused_min_cheb = 16, fused_min_pcg = 4, fused_min_cf = 4;
    int fused_min = 4;                 // smallest side from which the marching forms of the solvers run (PCG direction step, fused
                                       // Chebyshev, fused correction): periodic wraps need ni >= the kernel's halo lanes (2)
    int tail_from = 0;                 // first multigrid level k_mg_tail may start from (0: a small mesh's whole V-cycle in one launch)
    int semi_z32 = 1;                  // semi-coarsened cycle: z handed to the PCG direction step in single precision (with mg_f32), as the
                                       // uniform path does.  Off until the end of round 3 (it saved 4 % per iteration and cost 0.4 - 0.6
                                       // iterations per solve then: 105.9 against 104.7 ms); with the final kernels the 8192-wide graded
                                       // channel runs 75.9 against 79.7 ms per step, the annulus 49.3 against 51.3, same iteration counts
    int semi_fuse_up = 1;              // semi-coarsened cycle: both colours of the way up in one marching pass on the large levels
    int mg_semi = 1;
    bool semi_on = false;
    std::vector<SemiLevel> semi;
    char *semi_mem = nullptr;
    size_t semi_mem_bytes = 0;
    int mg_odd_fix = 1;            // coarse faces next to a single-cell last aggregate (odd fine count) scaled by the real centre distance (0: by the coarsening factor, as round 1 did; 5000^2: 10.2 -> 8 PCG iterations)
    int mg_line_dirs = 1;          // 1: a line-smoothed level sweeps both directions; 0: only those whose own coupling ratio exceeds the threshold (wall-graded 8192^2 channel: 79 iterations instead of 12, the wide centre cells need the i-lines)
    double mg_line_threshold = 10.0;   // a level is line-smoothed when its strong / weak coupling ratio exceeds this
    double mg_line_omega = 0.8;
    // Contiguous-per-XCD block -> work item map of the marching kernels (A-B switch "xcd_map").  Measured at 8192^2: the
    // map LOSES 18 % (k_chebk) to 40 % (k_mg_up_k) against the plain one, where the eight XCDs stream neighbouring strips
    // of the same rows at the same time (DRAM pages stay open across the channels); the halo lines two XCDs then both
    // fetch cost less than that.  Off.
    int xcd_map = 0;
    // marching kernels that run as lockstep workgroups of 8 neighbouring strips (MarchItems<8>); bits: 1 k_cg_pq,
    // 2 k_chebk with one right-hand side, 4 with two, 8 k_mg_up_k (finest level), 16 k_mg_down_k with the PCG residual
    // update, 32 k_mg_down_k (finest level), 64 the coarse levels of both, 128 k_flux_march (momentum and pressure
    // right-hand sides), 256 k_chebk with two right-hand sides as lockstep workgroups of FOUR strips (its 222 registers
    // allow two waves per SIMD either way).  Default: what measured faster at 8192^2.
    int wide_mask = 1 | 2 | 8 | 16 | 128 | 256;
    bool bb_ready = false;       // scal[83] holds (b, b) of the pressure right-hand side just built (launch_pressure_rhs)
    int plane_swap = 1;          // the Chebyshev result plane becomes u* / v* / k / omega by rebinding pointers, not by a copy
    int mg_tail = 1;             // 1: levels that fit one workgroup's LDS are cycled by one launch
    double *tail_tab = nullptr;  // 2 x TAIL_TABLE_BYTES of the workspace: the V-cycle tail's table, the whole-solve tail's
    TailArgs ptail{};            // the hierarchy from level 0 for k_pcg_tail (the whole MG-PCG solve in one launch)
    bool ptail_ok = false;
    int pcg_tail = 1;            // 1: meshes of at most pcg_tail_cells cells whose hierarchy fits one workgroup's LDS solve p' in one launch
    int pcg_tail_cells = TAIL_MAX_CELLS;   // (what fits is decided by the LDS: about 2800 cells)
    int tail_lt = -1;            // first level of the tail (-1: none), set when the hierarchy is built
    TailArgs tail{};
    double *sst_S = nullptr, *sst_F1 = nullptr, *sst_F2 = nullptr, *sst_Dk = nullptr, *sst_Dw = nullptr, *sst_dw = nullptr;
    double *m_xc = nullptr, *m_yc = nullptr;     // cell centres
    double *wp = nullptr, *wi = nullptr, *wj = nullptr;   // omega transport operator (k keeps tp, ti, tj)
    double *wall_xy = nullptr;    // wall face centres (x, y) pairs
    int n_wall = 0;
    bool sst_cached = false;      // _S, _F1, _F2, _D_k, _D_omega, _d_wall exist (after the first update / a restart)
    bool sst_dw_set = false, cc_set = false;
    double dt_ap = -1.0;          // dt the a_P plane was built with
    bool ap_cached = false;       // laminar: a_P / a_i / a_j are current for dt_ap
    double rho_cached = -1.0;     // Jacobi row-sum bound of the cached momentum operator (< 0: unknown)

    cfd_solve_stats stats{};

    // phase timing
    bool timing = false;
    double phase_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // lagged host reads: the copy of a few device scalars is enqueued, the event recorded behind it, more kernels follow;
    // the host waits for the event only, so the GPU keeps working while the host looks at the numbers
    hipEvent_t ev_read = nullptr;
    int host_lag = 1;            // 0: every read is a full stream synchronisation (A-B switch "host_lag")
    bool cheb_lag_ok = false;    // the last Chebyshev solve needed at least two launches
    bool dt_host_valid = false;  // h_dt_ahead holds the (globally reduced) minima the last cont pass computed ahead
    double h_dt_ahead[2] = {0.0, 0.0};

    // live kernel probe (bench.py): HIP-event pairs around every launch of one kernel id, or (CFD_PROBE_ALL) of every id
    int probe_kernel = -1;
    int probe_capacity = 2048;    // launches a probe run can time (option "probe_capacity")
    std::vector<hipEvent_t> probe_e0, probe_e1;
    std::vector<int> probe_id;
    std::vector<double> probe_b;  // algorithmic bytes per cell of each timed launch (they differ per launch)
    int probe_n = 0;
    double probe_bytes = 0.0;
    double last_rhs_bytes = 0.0;  // algorithmic bytes per cell of the last momentum right-hand side launch
    inline bool probing(int id) const { return (id == probe_kernel || probe_kernel == CFD_PROBE_ALL) && probe_n < (int)probe_e0.size(); }
    inline void probe_begin(int id) {
        if (probing(id)) (void)hipEventRecord(probe_e0[probe_n], stream);
    }
    inline void probe_end(int id, double bytes_per_cell = 0.0) {
        if (probing(id)) {
            (void)hipEventRecord(probe_e1[probe_n], stream);
            probe_id[probe_n] = id;
            probe_b[probe_n] = bytes_per_cell;
            ++probe_n;
            probe_bytes += bytes_per_cell;
        }
    }

    MeshDev mesh() const {
        return MeshDev{m_vol, m_wri, m_wnj, m_nix, m_niy, m_njx, m_njy, m_si, m_sj, m_di, m_dj, m_lmin};
    }
    double *U() const { return u[cur]; }
    double *V() const { return v[cur]; }
    double *P() const { return p[cur]; }
    double *U1() const { return u[1 - cur]; }
    double *V1() const { return v[1 - cur]; }
    double *P1() const { return p[1 - cur]; }
};

// ------------------------------------------------------------------ launch helpers
inline dim3 block2d() { return dim3(64, 4, 1); }
inline dim3 grid2d(int nx, int ny) { return dim3((nx + 63) / 64, (ny + 3) / 4, 1); }

// stencil.hip
int launch_eddy_viscosity(Ctx *c);
int launch_komega_sources(Ctx *c, const double *ux, const double *uy, const double *vx, const double *vy, double *sk, double *sw);
int launch_effective_diffusivity(Ctx *c, double nu, double sigma, double *out);
int launch_komega_coeffs(Ctx *c, double dt, bool both);   // k-omega: nu_t + conductances in one pass
int launch_diffusion_ap(Ctx *c, double dt, double nut_coef, double *ai, double *aj, double *ap,
                        const double *nu_plane = nullptr);
int launch_wall_distance(Ctx *c);
int launch_sst_f2_full(Ctx *c);
int launch_sst_sources(Ctx *c);
int launch_sst_rhs(Ctx *c, double dt, const double *phi, const double *src, const double *ci, const double *cj, double *b);
int launch_flow_bc(Ctx *c, double *u, double *v, double *p);
int launch_turb_bc(Ctx *c);
int launch_dt(Ctx *c, int use_cfl, double cfl, int use_cfl_diff, double cfl_diff, double out[2]);
int launch_momentum_rhs(Ctx *c, double dt, const double *u, const double *v, const double *p, double *cont = nullptr);
int launch_turb_rhs(Ctx *c, double dt);
int launch_pressure_rhs(Ctx *c, const double *u, const double *v, const double *p, double *b);
int launch_pressure_coeffs(Ctx *c, Level &L);     // unpinned face planes and diagonal
int launch_pin_operator(Ctx *c, Level &L);         // detach the reference cell (solver.py:1033-1044)
int launch_pcorr_bc(Ctx *c);
int launch_correct(Ctx *c, const double *us, const double *vs, const double *u0, const double *v0,
                   const double *p0, double *uo, double *vo, double *po, double ru, double rv, double rp);
int launch_continuity(Ctx *c, const double *u, const double *v, const double *p, double *out);
int launch_face_flux(Ctx *c, const double *u, const double *v, const double *p, int rc, int use_peff,
                     double *fi, double *fj);
int launch_grad(Ctx *c, const double *phi, double *gx, double *gy);
int launch_advect(Ctx *c, const double *phi, const double *fi, const double *fj, double *out);
int launch_clip(Ctx *c, double *f, double lo);
int launch_copy_plane(Ctx *c, const double *src, double *dst);
int launch_weighted_sum(Ctx *c, const double *f, const double *w, double out[2]);
int launch_metrics_from_nodes(Ctx *c, const double *xn, const double *yn, int node_pitch);
int launch_metrics_from_rect(Ctx *c, const double *x1, const double *y1);

// comm.hip -- no-ops on a single rank
int halo_exchange(Ctx *c, const Geo &g, int depth, double *a, double *b = nullptr, double *d = nullptr,
                  double *e = nullptr);
// several planes of different levels / depths in one exchange (one RCCL group on the native transport)
struct HaloSpec {
    const Geo *g;
    int depth;
    double *plane;
};
int halo_exchange_multi(Ctx *c, int n, const HaloSpec *specs);
int all_reduce(Ctx *c, double *dev, int n, int op);    // op: 0 sum, 1 min
int all_gather_rows(Ctx *c, const double *src, double *dst, size_t elems_each);
int rccl_unique_id(void *id, size_t bytes);
int rccl_attach(Ctx *c, const void *id, size_t bytes);
void rccl_release(Ctx *c);
int rccl_ranks(Ctx *c, int *ranks);
inline double global_cells(const Ctx *c) {
    return (double)c->g.ni * (double)(c->comm.nranks > 1 ? c->comm.nj_global : c->g.nj);
}

// linalg.hip
int rect_normals(Ctx *c);
int sum_finish(Ctx *c, const double *partials, int nblocks, int nq, double *out);   // + all-reduce over slabs
int reduce_finish(Ctx *c, int nq, int scal_off);       // partials -> scal[scal_off .. +nq)
int read_scalars(Ctx *c, int off, int n, double *host);
int read_scalars_begin(Ctx *c, int off, int n);             // enqueue the copy into h_scal[off ..) and record ev_read
int read_scalars_end(Ctx *c, int off, int n, double *host); // wait for ev_read only
int read_scalars2(Ctx *c, int off1, int n1, double *host1, int off2, int n2, double *host2);   // two ranges, one synchronisation
int read_scalars_all(Ctx *c);                           // the whole of Ctx::scal into Ctx::h_spec, one synchronisation
int spec_verify(Ctx *c, double *res, bool *ok);         // stopping tests of a speculative iteration on that copy
int momentum_like_solve(Ctx *c, const OpDev &op, double *const b[2], double *const x[2], const double *const x0[2], int nb,
                        double tol, int maxiter, int *iters, double relres[2], double rms[2]);
int build_pressure_hierarchy(Ctx *c);
int pressure_solve(Ctx *c, const double *b, double *x, double tol, int maxiter, int *iters, double *relres);   // b is a work plane: PCG turns it into the residual in place
int vcycle(Ctx *c, const double *r, double *z);
int launch_spmv(Ctx *c, const OpDev &op, const double *x, double *y);
int time_kernel(Ctx *c, int kernel, int reps, double *avg_us);

// stencil.hip: entry points of the per-phase host wrappers
int launch_momentum_rhs_given(Ctx *c, double dt, const double *advu, const double *advv, const double *extra, double force_u,
                              double force_v);
bool correct_flux_ok(const Ctx *c);
int launch_correct_flux(Ctx *c, double dt, double ru, double rv, double rp, bool *dt_too, int *nb);
int finish_correct_flux(Ctx *c, bool dt_too, int nb, double *cont);
int launch_wall_friction(Ctx *c, int side, int lo, int n, const double *tab_dev, double nu, double rho, int use_nut,
                         double *tau_dev, double out[2]);

// noalloc.hip: looks up this GPU's copy of the static scratch array (the current device's), once
int noalloc_prepare_device();

// api.hip
int to_device(Ctx *c, const double *host, double *plane, int kind);
int from_device(Ctx *c, const double *plane, double *host, int kind);

}  // namespace cfd
