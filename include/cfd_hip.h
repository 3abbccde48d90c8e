/*
 * cfd_hip.h -- C ABI of the MI355X (gfx950) SIMPLE/PISO hot path.
 *
 * The reference (AltoAuto/CFD) has no FFI: its boundary is the Python module API of
 * "Project 1/solver.py".  The entry points below are what a ctypes binding of that
 * module binds for the hot path; each one cites the reference routine it replaces
 * (file:line relative to /root/reference).  Plain pointers and sizes only.
 *
 * Conventions
 *   - every function returns 0 on success, a negative CFD_E_* code on failure; the
 *     message is available from cfd_last_error() (thread local).
 *   - host arrays are the reference's numpy layout: C-order float64, shape
 *     (Ni+2, Nj+2) for ghosted cell fields [i][j] (j contiguous), (Ni+1, Nj) for
 *     i-face arrays, (Ni, Nj+1) for j-face arrays, (Ni, Nj) for interior arrays.
 *   - device layout: transposed planes [j][i], i contiguous, CFD_GHOST_ROWS
 *     ghost rows below and above the interior (the innermost is the reference's ghost
 *     layer), row pitch padded to 256 B; j-slabs are contiguous row blocks.
 *   - one context per GPU / per process; calls on one context must not overlap.
 */
#ifndef CFD_HIP_H
#define CFD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cfd_ctx cfd_ctx;

enum {
    CFD_OK = 0,
    CFD_E_ARG = -1,      /* bad argument / shape                                  */
    CFD_E_HIP = -2,      /* HIP runtime error                                     */
    CFD_E_STATE = -3,    /* call order violated (mesh / BC not set, ...)          */
    CFD_E_NOCONV = -4,   /* a linear solve did not reach its tolerance            */
    CFD_E_NUMERIC = -5   /* non-finite value met                                  */
};

/* field ids for upload / download (fields dict of solver.py:370-387, turbulence.py:215-259) */
enum { CFD_F_U = 0, CFD_F_V = 1, CFD_F_P = 2, CFD_F_K = 3, CFD_F_OMEGA = 4, CFD_F_NUT = 5,
       CFD_F_PCORR = 6, CFD_F_AP = 7, CFD_F_AI = 8, CFD_F_AJ = 9, CFD_F_BU = 10, CFD_F_BV = 11,
       /* Menter SST caches the reference keeps inside `fields` (turbulence.py:447-534, 343-355): _S, _F1, _F2,
        * _D_k, _D_omega, _d_wall; valid ids only after cfd_set_turbulence_model(ctx, CFD_TURB_SST)          */
       CFD_F_SST_S = 12, CFD_F_SST_F1 = 13, CFD_F_SST_F2 = 14, CFD_F_SST_DK = 15, CFD_F_SST_DW = 16, CFD_F_SST_DWALL = 17,
       /* the snapshot planes of cfd_snapshot (u_prev, v_prev, p_prev of solver.correct_velocity, solver.py:1056-1087) */
       CFD_F_U_PREV = 18, CFD_F_V_PREV = 19, CFD_F_P_PREV = 20,
       CFD_F_COUNT = 21 };

/* turbulence closures (turbulence.py:19-62) */
enum { CFD_TURB_KOMEGA = 0, CFD_TURB_SST = 1 };

/* Ghost rows below and above the interior rows of every device plane (plane row r holds j = r - CFD_GHOST_ROWS):
 * the halo callback of a j-slab host addresses rows with it (cfd_set_comm); depth <= CFD_GHOST_ROWS.        */
#define CFD_GHOST_ROWS 4
int cfd_ghost_rows(void);
/* Columns to the left of interior column i = 0 in every plane row (one 128-byte line; the last of them is the
 * reference's ghost column): interior cell (i, j) of a plane sits at (j + CFD_GHOST_ROWS) * pitch + i + CFD_COL_OFFSET. */
#define CFD_COL_OFFSET 16
int cfd_col_offset(void);

/* array kinds (host shapes) */
enum { CFD_KIND_CELL_GHOSTED = 0, CFD_KIND_CELL = 1, CFD_KIND_FACE_I = 2, CFD_KIND_FACE_J = 3 };

/* boundary sides */
enum { CFD_SIDE_IMIN = 0, CFD_SIDE_IMAX = 1, CFD_SIDE_JMIN = 2, CFD_SIDE_JMAX = 3 };

/* per-index ghost-cell operation codes of the flow BC table (bc.py:196-347) */
enum { CFD_BC_NONE = 0, CFD_BC_AFFINE = 1, CFD_BC_SYMMETRY = 2, CFD_BC_PERIODIC = 3 };

/* turbulence BC op codes (turbulence.py:266-341), executed in list order */
enum { CFD_TOP_WRAP_I = 0, CFD_TOP_WRAP_J = 1, CFD_TOP_SET_INTERIOR = 2, CFD_TOP_AFFINE_GHOST = 3 };

typedef struct cfd_turb_op {
    int32_t op;          /* CFD_TOP_*                                              */
    int32_t field;       /* CFD_F_K or CFD_F_OMEGA                                 */
    int32_t side;        /* CFD_SIDE_*                                             */
    int32_t lo, hi;      /* inclusive boundary-index range (bc.py:16-35)           */
    int32_t value_off;   /* offset into the values array (hi-lo+1 doubles)         */
    double a;            /* ghost = a * phi_P + values[idx]  (AFFINE_GHOST)        */
} cfd_turb_op;

typedef struct cfd_solve_stats {
    int32_t mom_iters;       /* iterations of the (u, v) predictor solve            */
    int32_t p_iters;         /* Krylov iterations of the last pressure solve        */
    int32_t k_iters;         /* iterations of the (k, omega) transport solve        */
    int32_t p_solver;        /* 0 = MG-PCG, 1 = MG-BiCGStab (asymmetric seam)       */
    int32_t mom_solver;      /* 0 = Jacobi-BiCGStab, 1 = Jacobi-Chebyshev           */
    int32_t k_solver;        /* same for (k, omega)                                 */
    double mom_relres[2];    /* final |r|/|b| of u and v                            */
    double p_relres;         /* final |r|/|b| of the pressure correction            */
    double k_relres[2];
    int32_t p_unconverged;   /* pressure solves since cfd_create that stopped at maxiter above the tolerance
                              * (frozen operator: counted, the reference's cached-pyamg branch does not raise either,
                              * solver.py:829-833; fresh operator: CFD_E_NOCONV like solver.py:842-844).  Transport
                              * solves never end unconverged: CFD_E_NOCONV after a restart with four times maxiter */
    int32_t transport_stalled; /* transport solves since cfd_create that were asked for a tolerance below the rounding
                              * level of the fp64 recurrences (< 1e-12) and stalled between 1e-12 and 1e-6 after both
                              * restarts: accepted as "as far as fp64 goes" and counted here (the host warns); with a
                              * tolerance >= 1e-12 the same stall is CFD_E_NOCONV                                      */
} cfd_solve_stats;

const char *cfd_last_error(void);
int cfd_version(void);

/* ---- context ------------------------------------------------------------------ */
/* Replaces the in-process state of solver.run_case (solver.py:1367-1451): all planes
 * live in HBM.  workspace == NULL: the library hipMallocs cfd_workspace_bytes() itself;
 * otherwise the caller (e.g. a torch tensor) owns the memory, which lets the host
 * build zero-copy views for RCCL halo exchange.  device: HIP device ordinal.       */
/* The workspace holds every plane a context needs on uniform and mildly stretched meshes, the single-precision face
 * planes of the multigrid levels included.  Allocated by the library itself, outside the workspace, only when the
 * case asks for them (cfd_extra_device_bytes reports the total): the Menter-SST planes (cfd_set_turbulence_model: 11
 * planes), the wall-face list, and -- when the first pressure hierarchy of a STRETCHED mesh is built (coupling ratio
 * above "mg_line_threshold") -- the line-multigrid data: semi-coarsened levels and their factors (about 13 planes'
 * worth of bytes: 7.5 GB at 8192^2), or the block factors of the alternating line smoother (4 planes per level).
 * A failed allocation is reported by the call that builds the hierarchy (CFD_E_HIP), not silently.                   */
size_t cfd_workspace_bytes(int ni, int nj);
int cfd_create(cfd_ctx **out, int ni, int nj, int device, void *workspace, size_t workspace_bytes);
int cfd_destroy(cfd_ctx *ctx);

/* ---- j-slab decomposition (no counterpart in the single-process
 * reference, whose run_case owns the whole mesh, solver.py:1367-1451) ----------------
 * One context per rank holds rows [j_offset, j_offset + nj_local) of a global (Ni, nj_global)
 * mesh; nj_local is the same on every rank and even.  Ghost rows at a cut carry the
 * neighbour's interior rows; physical j boundaries exist only on the first / last rank.
 * The multigrid hierarchy is slab-distributed on its fine levels and replicated
 * ("agglomerated") below.  Periodicity along j is not supported across slabs.
 * All arrays passed to the other entry points are the slab-local slices; side tables of
 * i_min / i_max have nj_local entries.  cfd_mesh_from_rect takes y_nodes with nj_local + 3
 * entries starting one node below the slab; cfd_mesh_from_nodes is single-rank only.
 * Communication goes through three callbacks that the host implements on the context's
 * stream (torch.distributed over RCCL in cfd_amd/slab.py); each returns 0 on success.   */
/* exchange `depth` rows at each cut of `nplanes` planes of pitch `pitch` (doubles) and
 * nj_local interior rows: plane row r holds j = r - CFD_GHOST_ROWS.  My rows [nj_local-depth, nj_local) go to
 * rows [-depth, 0) of rank+1, my rows [0, depth) to rows [nj_local, nj_local+depth) of rank-1.  */
typedef int (*cfd_halo_fn)(void *user, int nplanes, double *const *planes, int pitch, int nj_local, int depth);
/* in-place all-reduce of n device doubles; op 0 = sum, 1 = min                                  */
typedef int (*cfd_allreduce_fn)(void *user, double *dev, int n, int op);
/* dst[rank * elems .. (rank+1) * elems) = src of that rank, for all ranks (device pointers)     */
typedef int (*cfd_allgather_fn)(void *user, const double *src, double *dst, size_t elems);
/* min_slab_rows (>= 16): a multigrid level stays slab-distributed while each slab keeps at least this
 * many rows; the next level is replicated.  Must be the same on every rank.                        */
size_t cfd_workspace_bytes_slab(int ni, int nj_local, int nranks, int min_slab_rows);
int cfd_create_slab(cfd_ctx **out, int ni, int nj_local, int device, void *workspace, size_t workspace_bytes,
                    int rank, int nranks, int min_slab_rows);
int cfd_set_comm(cfd_ctx *ctx, cfd_halo_fn halo, cfd_allreduce_fn allreduce, cfd_allgather_fn allgather, void *user);
/* Native transport: the library itself enqueues the three collectives on the context's stream with RCCL
 * (grouped ncclSend / ncclRecv of the halo rows to the two neighbours, ncclAllReduce, ncclAllGather), no host
 * code between the kernels.  No counterpart in the reference (run_case, solver.py:1367-1451, is one process).
 * Rank 0 calls cfd_rccl_unique_id and hands the CFD_RCCL_ID_BYTES bytes to every rank by any host channel;
 * then all ranks call cfd_set_comm_rccl (collective, blocking).  The transport set last is the one in use.
 * RCCL is loaded at run time (librccl.so.1), preferring a copy the process already holds.                    */
#define CFD_RCCL_ID_BYTES 128
int cfd_rccl_unique_id(void *id, size_t bytes);
int cfd_set_comm_rccl(cfd_ctx *ctx, const void *id, size_t bytes);
/* ranks of the library's own RCCL communicator, read back from RCCL (ncclCommCount); 0 when the context has none
 * (single rank, or the callback transport).  No counterpart in the reference.                                  */
int cfd_rccl_ranks(cfd_ctx *ctx, int *ranks);
/* collectives issued so far by this context: halo exchanges, all-reduces, all-gathers; native = 1 with RCCL */
int cfd_comm_counts(cfd_ctx *ctx, long long counts[3], int *native);
/* the same three counts per phase of the outer step, counts[3 * phase + kind]; phases in the order of
 * cfd_phase_times (bc, adv, mom, press, corr, cont, turb, dt), phase 8 = outside any phase              */
int cfd_comm_counts_by_phase(cfd_ctx *ctx, long long counts[27]);
/* refresh the cut ghost rows of a field plane (after cfd_upload_field on every rank)            */
int cfd_exchange_field(cfd_ctx *ctx, int field_id);
/* device bytes this context currently holds outside its workspace (see cfd_workspace_bytes)       */
int cfd_extra_device_bytes(cfd_ctx *ctx, size_t *bytes);
/* number of multigrid levels and index of the first replicated one (0 on a single rank)         */
int cfd_level_info(cfd_ctx *ctx, int *n_levels, int *first_replicated);
/* multigrid level `level` of the built pressure hierarchy: its size, the largest strong / weak coupling ratio of
 * its operator (max over the cells of (c_S + c_N) / (c_W + c_E) and of the inverse) and whether it is smoothed by
 * alternating block lines (ratio above "mg_line_threshold", default 10) instead of point Jacobi.  The line smoother
 * replaces pyamg's strength-of-connection coarsening (solver.py:821-833) as the answer to stretched meshes.   */
int cfd_level_smoother(cfd_ctx *ctx, int level, int *ni, int *nj, double *aniso, int *line);

int cfd_set_stream(cfd_ctx *ctx, void *hip_stream);
int cfd_synchronize(cfd_ctx *ctx);
int cfd_pitch(cfd_ctx *ctx);                       /* row pitch in doubles          */
/* byte offset of a field plane inside the workspace (for zero-copy host views).  Ask again after every step: the
 * library rebinds plane pointers instead of copying planes (the u, v ping-pong of a SIMPLE iteration, Chebyshev and
 * correction results that become u*, v*, k, omega), so the plane that holds a field changes over a run.            */
int64_t cfd_field_offset(cfd_ctx *ctx, int field_id);

/* ---- mesh (data contract of mesh.precompute_from_nodes, mesh.py:186-399) -------- */
/* Host arrays in the reference's .npz layout.  w_left is not passed: the contract has
 * w_left = 1 - w_right exactly (mesh.py:268-276) and the host wrapper checks it.     */
int cfd_upload_mesh(cfd_ctx *ctx,
                    const double *cell_volume,      /* (Ni, Nj)                     */
                    const double *w_right_i,        /* (Ni+1, Nj)                   */
                    const double *w_north_j,        /* (Ni, Nj+1)                   */
                    const double *face_normal_i,    /* (Ni+1, Nj, 2)                */
                    const double *face_normal_j,    /* (Ni, Nj+1, 2)                */
                    const double *diff_d_i,         /* (Ni+1, Nj)                   */
                    const double *diff_d_j,         /* (Ni, Nj+1)                   */
                    const double *cell_spacing_i,   /* (Ni, Nj)                     */
                    const double *cell_spacing_j);  /* (Ni, Nj)                     */
/* Same planes computed on the device from the node coordinates (Ni+1, Nj+1, 2); avoids
 * the 27-doubles-per-cell host mesh at 8192^2 (mesh.py:186-399).   */
int cfd_mesh_from_nodes(cfd_ctx *ctx, const double *node);
/* Same, for a rectangular mesh given by its two 1-D node coordinate arrays
 * (mesh.generate_rect_nodes, mesh.py:82-87) without materialising the node grid.    */
int cfd_mesh_from_rect(cfd_ctx *ctx, const double *x_nodes, const double *y_nodes);
/* download one device mesh plane (for parity tests): id 0..11 = vol, wr_i, wn_j, nix,
 * niy, njx, njy, s_i, s_j, d_i, d_j, lmin; kind as CFD_KIND_*                        */
int cfd_download_mesh_plane(cfd_ctx *ctx, int plane_id, double *host);

/* ---- fields --------------------------------------------------------------------- */
/* solver.initialize_fields / load_state / save_fields (solver.py:370-419, 1296-1310)  */
int cfd_upload_field(cfd_ctx *ctx, int field_id, const double *host, int kind);
int cfd_download_field(cfd_ctx *ctx, int field_id, double *host, int kind);

/* ---- parameters ----------------------------------------------------------------- */
/* physics.nu, physics.pressure_gradient, solver.rhie_chow, turbulence on/off
 * (solver.py:1393-1397, 1529, turbulence.py:215-224)                                 */
int cfd_set_physics(cfd_ctx *ctx, double nu, double dpdx, double dpdy, int rhie_chow, int turbulent);
/* solver.momentum / solver.pressure {tol, maxiter} (solver.py:863-864, 919-920);
 * k/omega transport uses the reference's hard-coded 1e-8 / 200 unless overridden
 * (solver.py:788-789, 1138-1139).  freeze_pressure = 1 keeps the first pressure
 * operator of the run (cached pyamg hierarchy, solver.py:821-833).       */
int cfd_set_solver(cfd_ctx *ctx, double mom_tol, int mom_maxiter, double p_tol, int p_maxiter,
                   double turb_tol, int turb_maxiter, int freeze_pressure);
/* multigrid shape: pre/post Jacobi sweeps, damping, coarsest-level sweeps; fused = 1 uses
 * the temporally blocked marching kernels (all sweeps of a pass in one kernel) when
 * nu_pre = nu_post = 2 or 3                                                                 */
int cfd_set_multigrid(cfd_ctx *ctx, int nu_pre, int nu_post, double omega, int coarse_sweeps, int fused);
/* Two-factor form of cfd_set_smoother_omegas (kept for callers of round 1): damping of the odd smoothing sweeps, the
 * even ones use `omega` of cfd_set_multigrid; the sweeps of one pass commute, so the preconditioner stays symmetric.
 * omega2 <= 0: `omega` on every sweep.  The single-block coarsest solve uses the mean of the factors.  The DEFAULT
 * dampings are the three of cfd_set_smoother_omegas below, not a pair.  (Replaces pyamg's smoother choice,
 * solver.py:821-833.)                                                                                   */
int cfd_set_smoother_omega2(cfd_ctx *ctx, double omega2);
/* General form: n = 1..4 damping factors in (0, 2), mean <= 1, used cyclically by the sweeps of a pass.
 * Default: V(3,3) with (1.25, 0.5, 1.0) (three sweeps fused per pass: 5 MG-PCG iterations per solve at
 * 8192^2); j-slab contexts run the same kernels (a cut carries CFD_GHOST_ROWS halo rows).              */
int cfd_set_smoother_omegas(cfd_ctx *ctx, int n, const double *omegas);
/* Systems without any Dirichlet pressure (the reference pins one cell, solver.py:1033-1044):
 *   on = 1 (default): when the operator is symmetric the singular system is solved with every row kept
 *   (right-hand side made compatible at the reference cell's row, the row the reference drops) and the
 *   solution is shifted to the reference value: the same p' up to rounding, two V-cycle-PCG iterations
 *   fewer at 8192^2;  on = 0: the reference cell is eliminated symmetrically.  Non-symmetric operators
 *   (periodic seam with varying coefficients, MG-BiCGStab) always eliminate the cell.                 */
int cfd_set_pressure_nullspace(cfd_ctx *ctx, int on);
/* Form of the Rhie-Chow flux kernels (solver.py:455-513, 672-691, 926-933, 1205-1215).
 *   marching  : 1 (default) = one wavefront per 62-column strip walking down the rows, every face flux
 *               evaluated once; 0 = the 64x4-tile kernels (same bits, 1.5-1.9x the HBM traffic);
 *   lookahead : 1 (default) = the continuity pass at the end of an inner iteration also writes the
 *               momentum right-hand sides of the next one (same fields, a_P and dt); they are used only
 *               if nothing they depend on changed in between.  Results are identical either way.   */
int cfd_set_flux_kernels(cfd_ctx *ctx, int marching, int lookahead);
/* Transport (momentum, k-omega) solver policy; replaces the reference's spilu + scipy bicgstab
 * with x0 = 0 (solver.py:785-818), same stopping test |b - A x| <= tol |b|.
 *   warm_start : start from the previous field instead of zero;
 *   chebyshev  : Jacobi-Chebyshev semi-iteration (no inner products) whenever the Jacobi row-sum
 *                bound rho = max sum_n c_n / D is <= rho_max and the Chebyshev bound predicts
 *                <= max_iters iterations (time-step dominated systems); Jacobi-BiCGStab otherwise
 *                and whenever warm_start is off.                                              */
int cfd_set_transport_solver(cfd_ctx *ctx, int warm_start, int chebyshev, double rho_max, int max_iters);
/* turbulence clip after the k/omega solve (solver.py:1147-1151)                       */
int cfd_set_turbulence_clip(cfd_ctx *ctx, double k_min, double omega_min);

/* ---- boundary conditions --------------------------------------------------------- */
/* Flow ghost-cell table for one side, resolved per boundary index by the host from
 * the CONFIG["boundary"] segments (bc.apply_boundary_conditions, bc.py:259-347):
 *   kind_uv[n], kind_p[n]: CFD_BC_*;  u_g = ua*u_P + ub, v_g = va*v_P + vb,
 *   p_g = pa*p_P + pb;  symmetry uses the unit normal (nx, ny) (bc.py:241-256).       */
int cfd_set_flow_bc(cfd_ctx *ctx, int side, int n, const int32_t *kind_uv, const int32_t *kind_p,
                    const double *ua, const double *ub, const double *va, const double *vb,
                    const double *pa, const double *pb, const double *nx, const double *ny);
/* periodicity of the transport systems (bc.get_periodic_pairs, bc.py:72-91) and
 * whether the periodic wrap also copies p (bc.py:266-286)                             */
int cfd_set_periodicity(cfd_ctx *ctx, int per_i, int per_j, int wrap_p_i, int wrap_p_j);
/* pressure-correction BCs: Dirichlet-p masks per side (solver.py:245-276), the
 * periodicity seen by p' (solver.py:959-966) and the reference cell
 * (solver.py:1033-1044).  masks may be NULL (all false).                              */
int cfd_set_pressure_bc(cfd_ctx *ctx, const int32_t *mask_imin, const int32_t *mask_imax,
                        const int32_t *mask_jmin, const int32_t *mask_jmax,
                        int per_i, int per_j, int has_dirichlet, int pin_i, int pin_j, double pin_value);
/* k / omega boundary program (turbulence.apply_bcs, turbulence.py:291-341)             */
int cfd_set_turb_bc(cfd_ctx *ctx, int n_ops, const cfd_turb_op *ops, int n_values, const double *values);

/* ---- the outer step ---------------------------------------------------------------- */
/* bc.apply_boundary_conditions on the current u, v, p (+ periodic wrap of k, omega, nu_t) */
int cfd_apply_flow_bc(cfd_ctx *ctx);
/* solver.compute_adaptive_dt (solver.py:547-582) with nu_eff = nu + stored (lagged) nu_t
 * (solver.py:1473-1475).  use_cfl / use_cfl_diff = 0 means "None".  out = {dt, dt_adv,
 * dt_diff}.  Global min reduction on the device.                                        */
int cfd_compute_dt(cfd_ctx *ctx, int use_cfl, double cfl, int use_cfl_diff, double cfl_diff,
                   double dt_min, int use_dt_max, double dt_max, double out[3]);
/* One SIMPLE inner iteration (solver.py:1591-1681): BCs, snapshot, nu_t, a_P, advective
 * RHS, momentum predictor, BCs, pressure correction, velocity/pressure correction with
 * under-relaxation, BCs, continuity residual.  res = {continuity_l2, mom_u_l2, mom_v_l2}. */
int cfd_simple_iter(cfd_ctx *ctx, double dt, double relax_u, double relax_v, double relax_p,
                    double res[3]);
/* One PISO step (solver.py:1494-1589) with n_correctors correctors against the
 * start-of-step snapshot.                                                */
int cfd_piso_step(cfd_ctx *ctx, double dt, int n_correctors, double relax_u, double relax_v,
                  double relax_p, double res[3]);
/* solver.update_turbulence (solver.py:1090-1153): Wilcox k-omega, or Menter SST after
 * cfd_set_turbulence_model(ctx, CFD_TURB_SST)                                            */
int cfd_turbulence_step(cfd_ctx *ctx, double dt);
/* Turbulence closure: CFD_TURB_KOMEGA (default; turbulence.py:19-34, 388-394, 427-444) or CFD_TURB_SST
 * (Menter SST, turbulence.py:41-208, 343-386, 447-566).  SST allocates its eleven extra planes (six
 * caches, cell centres, the omega transport operator) once, outside the caller's workspace (a host that
 * implements the cfd_set_comm callbacks sees those pointers in its halo callback on j-slab runs).  It needs
 * cfd_upload_cell_centers and cfd_set_wall_faces before the first step.                                */
int cfd_set_turbulence_model(cfd_ctx *ctx, int model);
/* mesh["cell_center"] (Ni, Nj, 2) (mesh.py:375-394): read by the SST wall distance and the
 * finite-difference gradients of k and omega (turbulence.py:96-136, 179-208).  j-slabs: the slab's rows;
 * collective (the rows next to a cut are exchanged); cfd_set_wall_faces takes the wall faces of the WHOLE mesh */
int cfd_upload_cell_centers(cfd_ctx *ctx, const double *cell_center);
/* Face centres (n, 2) of every boundary face of type "wall", all segments concatenated
 * (turbulence._wall_face_centers, turbulence.py:69-93); n = 0: no wall (distance 1.0)                 */
int cfd_set_wall_faces(cfd_ctx *ctx, int n, const double *xy);
/* Whether the SST caches exist: the reference creates them in the first update_turbulence call and
 * uses a1 k / omega without limiter until then (turbulence.py:376-386).  set: a restart uploaded the
 * CFD_F_SST_* planes of a saved state and marks them valid                                            */
int cfd_sst_get_cached(cfd_ctx *ctx, int *cached);
int cfd_sst_set_cached(cfd_ctx *ctx, int cached);
int cfd_get_stats(cfd_ctx *ctx, cfd_solve_stats *out);
/* volume-weighted mean of u or v over the interior (post_processing.py:602-610)          */
int cfd_bulk_velocity(cfd_ctx *ctx, int field_id, double *out);

/* The same mean for callers that cannot allocate or synchronise: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  NO COUNTERPART IN THE REFERENCE (its bulk velocity is one numpy line,
 * post_processing.py:602-610).  Allocates nothing (no hipMalloc / hipFree, no allocating library routine), takes
 * no context: field_dev and volume_dev are device planes in the library's layout -- cell (i, j) at
 * (j + CFD_GHOST_ROWS) * pitch + i + CFD_COL_OFFSET, pitch >= ni + CFD_COL_OFFSET + 2 doubles -- e.g. workspace +
 * cfd_field_offset(ctx, id);
 * out_dev receives three doubles ON THE DEVICE: {sum f V / sum V, sum f V, sum V}; both kernels are enqueued on
 * hip_stream (NULL = the default stream) of the CURRENT device and the call returns without synchronising.  The
 * partial sums live in a fixed 64 KiB device array of the library (one copy per GPU, held until the process ends):
 * at most 32768 x 32768 cells, larger sizes are rejected (CFD_E_ARG); calls that overlap on one GPU race on that
 * array, calls that follow each other (any sizes) do not.  Same kernels, tiles and summation order as
 * cfd_bulk_velocity: the mean has the same bits (pitch may exceed the library's own, any stream of the device).
 * FIRST USE ON A DEVICE: the address of that device's copy of the array is looked up once (hipGetSymbolAddress
 * under a mutex, which may load the code object) -- by cfd_create on that device, else by the first call of this
 * entry point there.  A caller that is under HIP graph capture, or must not block, makes one plain call (or a
 * cfd_create) on the device beforehand; every later call allocates nothing, locks nothing and looks nothing up.  */
int cfd_bulk_velocity_noalloc(const double *field_dev, const double *volume_dev, int ni, int nj, int pitch,
                              void *hip_stream, double *out_dev);

/* ---- fine-grained entry points (parity tests; README.md:196-205 function list) ------- */
/* turbulence.eddy_viscosity, k-omega branch (turbulence.py:388-394): nu_t plane from k, omega */
int cfd_eddy_viscosity(cfd_ctx *ctx);
/* turbulence._sources_komega (turbulence.py:427-444) from the k, omega, nu_t planes and the CALLER'S velocity
 * gradients grad_u, grad_v (host (Ni, Nj, 2), as solver.grad_scalar returns them): source_k, source_omega host (Ni, Nj) */
int cfd_komega_sources(cfd_ctx *ctx, const double *grad_u, const double *grad_v, double *source_k, double *source_omega);
/* turbulence.effective_diffusivity, constant-sigma branch (turbulence.py:568-577): nu + sigma * nu_t of the nu_t plane,
 * host (Ni+2, Nj+2), corners included                                                                            */
int cfd_effective_diffusivity(cfd_ctx *ctx, double nu, double sigma, double *out);
/* solver._diffusion_coefficients + _momentum_diagonal (solver.py:147-230, 439-452):
 * writes a_i, a_j, a_P planes for nu_eff = nu + nut_coef * nu_t                           */
int cfd_momentum_diagonal(cfd_ctx *ctx, double dt, double nut_coef);
/* solver._face_mass_flux / _face_volumetric_flux (solver.py:455-544) from the current
 * u, v, p (+2/3 k when use_peff) and the a_P plane; host outputs (Ni+1,Nj), (Ni,Nj+1)     */
int cfd_face_flux(cfd_ctx *ctx, int rhie_chow, int use_peff, double *flux_i, double *flux_j);
/* solver.grad_scalar / numba_kernels.grad_scalar_numba (solver.py:322-358,
 * numba_kernels.py:39-83): gradient of a field plane -> host (Ni, Nj, 2)                  */
int cfd_grad(cfd_ctx *ctx, int field_id, double *grad);
/* solver.advect_scalar / numba_kernels.advect_scalar_numba (solver.py:649-669,
 * numba_kernels.py:5-36) with host-supplied face fluxes -> host (Ni, Nj)                   */
int cfd_advect(cfd_ctx *ctx, int field_id, const double *flux_i, const double *flux_j, double *out);
/* advective + pressure-gradient momentum RHS planes b_u, b_v (solver.py:672-691, 868-883,
 * 723-730) for the current fields and the a_P / a_i / a_j planes                           */
int cfd_momentum_rhs(cfd_ctx *ctx, double dt);
/* solver.solve_momentum's right-hand sides from the CALLER'S advective terms (solver.py:848-883): adv_u, adv_v are
 * host (Ni, Nj) arrays (adv_rhs["u"], adv_rhs["v"] interiors), extra an optional host (Ni+2, Nj+2) array whose
 * gradient is added to grad p (extra_pressure_field, solver.py:871-872), (dpdx, dpdy) = pressure_gradient       */
int cfd_momentum_rhs_from(cfd_ctx *ctx, double dt, const double *adv_u, const double *adv_v, const double *extra,
                          double dpdx, double dpdy);
/* post_processing.friction_coefficient_array's wall shear stress (post_processing.py:646-664) over the faces
 * lo..hi (inclusive) of one side, from the u, v (and nu_t when use_nu_t) planes in HBM: nothing but O(hi - lo)
 * numbers crosses PCIe.  t_hat (n, 2), dn (n: |(fc - cc) . n_hat| floored at 1e-12), seg_len (n) are the host's
 * geometry of those faces (post_processing.py:103-170, 646, 654-655).  tau: optional host array of n shear stresses
 * (j-slabs: summed over the ranks, every rank gets the whole array -- collective, so either every rank passes tau
 * or none does); out = {sum tau * seg_len, sum seg_len} over all ranks.  Leaves the step's look-ahead state alone. */
int cfd_wall_friction(cfd_ctx *ctx, int side, int lo, int hi, const double *t_hat, const double *dn, const double *seg_len,
                      double nu, double rho, int use_nu_t, double *tau, double out[2]);
/* solve the two momentum systems for the stored RHS planes (solver.py:785-818, 885-894)     */
int cfd_momentum_solve(cfd_ctx *ctx, double res[2]);
/* pressure-correction system (solver.py:904-1053): RHS from the current fields and solve     */
int cfd_pressure_solve(cfd_ctx *ctx);
/* download the assembled pressure operator planes (tests): c_i (Ni+1,Nj), c_j (Ni,Nj+1), diag,
 * rhs (Ni,Nj)                                                                               */
int cfd_download_pressure_system(cfd_ctx *ctx, double *c_i, double *c_j, double *diag, double *rhs);
/* solver.correct_velocity (solver.py:1056-1087) against the snapshot planes                  */
int cfd_correct(cfd_ctx *ctx, double relax_u, double relax_v, double relax_p);
int cfd_snapshot(cfd_ctx *ctx);   /* u0, v0, p0 := current (solver.py:1596-1598)              */
/* solver.continuity_residual (solver.py:1205-1215)                                           */
int cfd_continuity(cfd_ctx *ctx, double *out);
/* turbulence.apply_bcs program                                                               */
int cfd_apply_turb_bc(cfd_ctx *ctx);

/* ---- standalone linear-algebra entry points (tests, roofline micro-benchmarks) ------------ */
/* y = A x for the momentum operator (which = 0) or the pressure operator (which = 1);
 * x, y host (Ni, Nj)                                                                         */
int cfd_spmv(cfd_ctx *ctx, int which, const double *x, double *y);
/* one multigrid V-cycle z = M r on the pressure operator; host (Ni, Nj)                      */
int cfd_vcycle(cfd_ctx *ctx, const double *r, double *z);
/* solve A x = b on the pressure operator with the configured Krylov method; returns iters     */
int cfd_pressure_krylov(cfd_ctx *ctx, const double *b, double *x, double tol, int maxiter, int *iters,
                        double *relres);
/* timing helpers for bench.py: run `reps` launches of one hot kernel on resident data and
 * return the average kernel time in microseconds measured with HIP events on the context
 * stream.  kernel: 0 = pressure SpMV, 1 = Jacobi smoother sweep (finest level),
 * 2 = momentum RHS, 3 = batched momentum SpMV, 4 = residual+restrict, 5 = continuity        */
int cfd_time_kernel(cfd_ctx *ctx, int kernel, int reps, double *avg_us);
/* live probe: record a HIP-event pair on the context stream around EVERY launch of one hot
 * kernel while the solver runs, up to "probe_capacity" launches (cfd_set_option, default 2048, at most 16384; -1 = off): 0 = PCG direction step, 1 = finest-level
 * multigrid pre-smoother pass, 10 = the same with the PCG residual update folded in, 7 = post-smoother pass,
 * 2 = momentum right-hand sides (+ continuity, + time-step minima), 8 / 9 = Chebyshev launches with two / one
 * right-hand sides, 11 = PCG residual update, 12 = last PCG solution update, 13 = pressure right-hand side; CFD_PROBE_ALL =
 * all of them at once.  cfd_probe_read returns how many
 * launches were timed and their average duration, cfd_probe_read_id the same for one kernel of a CFD_PROBE_ALL run.  */
#define CFD_PROBE_ALL 1000
int cfd_probe(cfd_ctx *ctx, int kernel);
/* avg_bytes_per_cell (may be NULL): mean ALGORITHMIC bytes per cell of the timed launches -- they differ from launch to
 * launch for some kernels (the first two direction steps of a PCG solve skip the x / p_old planes: 40 / 56 / 64 B)  */
int cfd_probe_read(cfd_ctx *ctx, int *count, double *avg_us, double *avg_bytes_per_cell);
int cfd_probe_read_id(cfd_ctx *ctx, int kernel, int *count, double *avg_us, double *avg_bytes_per_cell);
/* Tuning / A-B switches of the kernel forms, for tests and measurements (results are the same bits either way unless
 * noted; the library reads no environment variable): "cheb_stages" (1 / 2 / 4 Chebyshev iterations per launch),
 * "cheb_prefetch" (1 / 2 rows of loads in flight per wave), "fuse_nut", "dt_lookahead", "fuse_resid" (0 / 1: the
 * fusions of DESIGN.md section 3), "nt_stores" (-1 auto / 0 / 1: nontemporal stores of the streaming kernels),
 * "march_ls" (rows per segment of the marching kernels, 0 = default), "mg_line" (-1 auto / 0 / 1: alternating line smoother
 * on the anisotropic / no / all levels; with "mg_line_threshold" and "mg_line_omega" it changes the V-cycle, hence
 * iteration counts, not the converged solution); "scalar_nu" (0 / 1: cfd_momentum_diagonal forms the face conductances
 * from the scalar nu alone, the scalar branch of solver._diffusion_coefficients, solver.py:205-207, for callers of the
 * per-phase functions that pass a scalar viscosity); "mg_line_dirs" (1 / 0: both line directions on a line-smoothed level /
 * only those whose own coupling ratio exceeds the threshold), "mg_tail" (1 / 0: V-cycle tail in one launch), "wide_mask"
 * (bit mask of the marching kernels that run as lockstep workgroups), "host_lag" (1 / 0: lagged host reads of the
 * residual norms), "plane_swap" (1 / 0: Chebyshev results adopted by pointer exchange / copied), "xcd_map", "mg_gamma",
 * "mg_om0" .. "mg_om2" (smoother dampings), "mg_f32" (1 / 0: the multigrid levels of the pressure preconditioner carry
 * face coefficients rounded to single precision -- the Krylov operator stays double, the solution is unchanged),
 * "mg_semi" (1 / 0: stretched meshes are cycled by zebra whole-line relaxation along i with semi-coarsening along j /
 * by the alternating block line smoother with full coarsening; one rank, frozen operator, lines of at most 8192 cells),
 * "cheb_tail" / "cheb_tail_cells" (1 / 0, cell limit: Chebyshev transport solves of small meshes in one launch),
 * "pcg_tail" (1 / 0: the pressure solve of a mesh whose hierarchy fits one workgroup's LDS is one launch with the
 * stopping test on the device / the launch-per-phase loop), "pcg_tail_cells" (cell limit of that path), "march_ls_model",
 * "speculate" (-1 / 0 / 1: inner iterations with one host read at their end, see cfd_kernel_counts; auto = contexts of
 * at most 4M cells, j-slab contexts of at most 16M cells), "fuse_correct_slab", "skip_fresh_bc", "probe_capacity" (launches a probe run can time), "mg_v32_emul" (measurement aid), "mg_c32" (0 .. 2: coarse multigrid levels, from the first, with single-precision vectors; levels of more than "mg_c32_min_cells" cells of the global level, default 2^18), "rect_normals" (1 / 0: axis-aligned meshes do not read the normal planes), "cheb_k6" (1 / 0: six fused Chebyshev iterations for a single right-hand side where that saves a launch), "fold_sums" (-1 / 0 / 1: PCG inner products summed inside the consuming kernels, auto = one-rank contexts of at most 4M cells), "semi_resid" (1 / 0: on semi-coarsened hierarchies the PCG residual update of unchecked iterations is formed by the finest level's way down / by k_cg_r; same bits), "lmin_const" (1 / 0: in laminar runs the diffusive time-step minimum, a constant of mesh, nu and cfl_diff, is computed by the first look-ahead pass and kept / computed by every pass; same bits), "pcg_xsum" (1 / 0: the pressure solution is summed from the kept PCG directions when the solve ends / updated in every direction step; same bits), "rr_byproduct" (1 / 0: (r, r) of the second-to-last expected PCG iteration summed by the
 * next pre-smoothing pass / by its own pass).  INTEGRATION.md section 4 has the table.  No counterpart in the reference.     */
int cfd_set_option(cfd_ctx *ctx, const char *name, double value);
/* Launch counts per kernel family since the last reset, so that a parity test can assert WHICH production kernels
 * its run went through: [0] 4-iteration Chebyshev launches with two right-hand sides, [1] with one, [2] 2-iteration
 * launches, [3] 1-iteration launches, [4] BiCGStab iterations, [5] marching flux launches, [6] of those with more
 * than one column strip, [7] tile flux launches, [8] max number of fused multigrid levels in a V-cycle, [9] PCG
 * iterations whose residual update ran inside the pre-smoother, [10] fused PCG direction steps, [11] V-cycles,
 * [12] max number of line-smoothed levels, [13] levels cycled by the single-launch V-cycle tail, [14] launches of the fused
 * correction / flux kernel, [15] levels of the semi-coarsened line multigrid, [16] inner iterations that ran speculatively
 * (option "speculate": solves replay the previous iteration's schedule, one host read at the iteration's end, stopping
 * tests evaluated then), [17] of those, iterations whose test failed and that were repeated host-checked, [18] pressure
 * solves that were ONE launch (k_pcg_tail: the whole MG-PCG solve of a mesh whose hierarchy fits one workgroup's LDS),
 * [19] transport solves (momentum, k / omega) that were ONE launch (k_cheb_tail), [20] of the one-right-hand-side Chebyshev
 * launches, those that fused six iterations (k_chebk<1,6>), [21] max number of coarse multigrid levels that kept their vectors
 * in single precision, [22] form of the marching flux kernels on this mesh (0 general, 1 axis-aligned: normals from the area
 * planes, 2 separable areas as well; a state, not a count), [23] pressure solves whose solution was summed from the kept
 * directions (k_cg_xsum).  n <= 24 entries are written.  No counterpart in the reference.  */
int cfd_kernel_counts(cfd_ctx *ctx, long long *counts, int n, int reset);
/* Host-side cost of the calls so far, process wide: out[0] = kernel launches, out[1] = blocking stream
 * synchronisations the library issued (each one idles the GPU while the host reads a scalar).  They bound the step
 * time of small meshes and thin j-slabs (tools/small_mesh_timing.py).  No counterpart in the reference.            */
int cfd_host_counters(long long out[2], int reset);
/* the blocking synchronisations of cfd_host_counters by the phase of the step that issued them: [0] boundary conditions,
 * [1] coefficients / right-hand sides, [2] momentum solves, [3] pressure solve, [4] correction, [5] continuity / closing
 * pass, [6] turbulence update, [7] time step, [8] outside any phase.  No counterpart in the reference.                  */
int cfd_host_syncs_by_phase(long long out[9], int reset);
/* accumulated device time per phase since the last reset (HIP events), microseconds:
 * {bc, adv(rhs), mom, press, corr, cont, turb, dt}                                          */
int cfd_phase_times(cfd_ctx *ctx, double out[8], int reset);
int cfd_enable_phase_timing(cfd_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* CFD_HIP_H */
