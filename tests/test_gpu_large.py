"""GPU parity at sizes where the production kernels of the 8192^2 run are the ones that execute (-m gpu).

The small reference cases (<= 24 x 32 cells) run through the small-mesh forms of several kernels: the 4-iteration
Chebyshev kernel needs both sides >= 64, the marching flux kernels have a strip seam only beyond 62 columns, a
multigrid level is fused only from 16 x 16 cells up, and a PCG solve defers its residual update to the next
pre-smoother only when it needs five or more iterations.  The cases here (oracle/cases.py LARGE_END_TO_END, golden
vectors written by `python oracle/gen_golden.py large` from the reference itself) are 64 x 64 and 160 x 96 cells;
each test asserts through cfd_kernel_counts WHICH kernel families its run went through.

Tolerances as in test_gpu_parity.py: stencil kernels bit-exact against the reference's vectors, solved fields
relative L2 <= 1e-10, continuity history 2e-6 (7 printed digits).  The golden runs solve momentum with ILU-BiCGStab
to 1e-13 (the ILU factors make the last iteration overshoot: the pinned oracle with DIRECT solves reproduces them to
2e-11 in p) and pressure with a sparse LU.  The pressure correction amplifies an error of the predictor by
~1/dt x cond: with the Jacobi-type transport solvers of the GPU path stopped at 1e-13 p lands 7e-9 from the
reference on the 160 x 96 k-omega case, with 1e-15 it lands 5e-11 -- so the GPU side of these comparisons runs its
transport solves to 1e-15 ("both sides solve much tighter than the comparison tolerance").
"""
import numpy as np
import pytest

from conftest import golden_mesh, load_golden, mask_corners, rel_l2
from oracle import cases
from oracle import cfd_oracle as orc

pytestmark = pytest.mark.gpu

FIELD_TOL = 1.0e-10
TRANSPORT_TOL = 1.0e-15      # see the module docstring


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")


def _run(name, tmp_path, tuning=None):
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    builder, fixup = cases.LARGE_END_TO_END[name]
    cfg, mp, kind = builder()
    cfg["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp, "path": tmp_path / "mesh" / "mesh.npz"}
    cfg["paths"] = {"case_dir": tmp_path, "mesh_dir": tmp_path / "mesh", "states_dir": tmp_path / "states",
                    "post_dir": tmp_path / "postproc"}
    mesh = hip_mesh.make_mesh(cfg["mesh"])
    if fixup is not None:
        fixup(cfg, mesh)
    cfg["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
    cfg["solver"]["momentum"]["tol"] = TRANSPORT_TOL
    # the 160 x 96 meshes are graded (max / min dy = 6.6, coupling ratio up to 18): left alone, their multigrid levels
    # would be line-smoothed; these tests are about the kernels of the uniform 8192^2 run, so point smoothing is forced
    # (the line-smoothed path has its own tests below)
    cfg["solver"]["tuning"] = dict({"mg_line": 0}, **(tuning or {}))
    fields, step, time = hip_solver.run_case(cfg)
    log = np.loadtxt(tmp_path / "postproc" / f"{cfg['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
    return fields, step, time, log, dict(hip_solver.LAST_RUN), cfg, mesh


def _compare(g, fields, step, time, log):
    assert step == int(g["step"])
    assert abs(time - float(g["time"])) <= 1e-11 * max(1.0, abs(float(g["time"])))
    unorm = max(1.0, np.linalg.norm(g["field_u"].ravel()))
    for key in [k for k in g if k.startswith("field_")]:
        ref, mine = g[key], fields[key[6:]]
        err = np.linalg.norm((mask_corners(mine) - mask_corners(ref)).ravel())
        den = np.linalg.norm(mask_corners(ref).ravel())
        assert err <= FIELD_TOL * den + 1e-12 * unorm, (key, err, den)
    h = g["history"]
    assert log.shape[0] == h.shape[0]
    assert np.array_equal(log[:, 0], h[:, 0]) and np.array_equal(log[:, 2], h[:, 2])
    assert np.all(np.abs(log[:, 3] - h[:, 3]) <= 2e-6 * h[:, 3] + 1e-12)
    assert np.all(np.abs(log[:, 1] - h[:, 1]) <= 2e-6 * np.maximum(h[:, 1], 1e-30) + 1e-300)


@pytest.mark.parametrize("tail", [1, 0])
@pytest.mark.parametrize("name", list(cases.LARGE_END_TO_END))
def test_production_kernels_against_reference(name, tail, tmp_path):
    """run_case on the GPU vs the reference (golden vectors) after 3 outer steps, and proof of the kernel path.
    tail = 1 (default): the levels below 4096 cells are cycled by k_mg_tail in one launch; tail = 0: every level
    runs its own kernels, as the levels above the tail do at 8192^2."""
    g = load_golden(f"e2e_{name}_frozen.npz")
    fields, step, time, log, last, cfg, mesh = _run(name, tmp_path, tuning={"mg_tail": tail})
    _compare(g, fields, step, time, log)
    kc = last["kernel_counts"]
    big = "160x96" in name
    # marching flux kernels; more than one 62-column strip on every mesh of this file
    assert kc["flux_march"] > 0 and kc["flux_march_multistrip"] == kc["flux_march"] and kc["flux_tile"] == 0, kc
    if tail:
        # 64x64 | 32x32 -> 16x16 -> 8x8 -> 4x4;  160x96 | 80x48 -> 40x24 -> 20x12 -> 10x6 -> 5x3
        assert kc["mg_fused_levels_max"] >= 1 and kc["mg_tail_levels"] >= 4, kc
    else:
        # fused V-cycle levels: 64x64 -> 32x32 -> 16x16 (-> 8x8 unfused): 3; 160x96 -> 80x48 -> 40x24 (-> 20x12): 3
        assert kc["mg_fused_levels_max"] >= 3 and kc["mg_tail_levels"] == 0, kc
    assert kc["cg_pq"] > 0 and kc["mg_deferred_resid"] > 0, kc       # PCG with residual updates inside the pre-smoother
    if "couette" not in name:
        # CFL-limited steps: the momentum (and k / omega) solves are 4-iteration Chebyshev launches, with both
        # right-hand sides and, once one of them has converged, with one
        assert kc["cheb4_2rhs"] > 0, kc
        if "komega" not in name:      # (the eddy viscosity of the k-omega run pushes the later steps to BiCGStab)
            assert last["stats"]["mom_solver"] == "chebyshev", last["stats"]
    if big and "couette" not in name and "komega" not in name:
        assert kc["cheb4_1rhs"] > 0, kc
    assert last["stats"]["p_unconverged"] == 0


@pytest.mark.parametrize("name", ["channel_forced_160x96", "channel_forced_komega_160x96"])
def test_production_kernels_against_oracle(name, tmp_path):
    """The same CONFIG through the pinned CPU oracle (direct solves, frozen pressure operator) and through the GPU:
    five outer steps, i.e. beyond what the golden vectors hold."""
    builder, fixup = cases.LARGE_END_TO_END[name]
    cfg, mp, kind = builder()
    cfg["time"]["max_steps"] = 5
    mesh = orc.rect_mesh(**mp)
    fixup(cfg, mesh)
    ref, rstep, rtime, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True)

    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    cfg2, mp2, _ = builder()
    cfg2["time"]["max_steps"] = 5
    cfg2["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp2, "path": tmp_path / "mesh" / "mesh.npz"}
    cfg2["paths"] = {"case_dir": tmp_path, "mesh_dir": tmp_path / "mesh", "states_dir": tmp_path / "states",
                     "post_dir": tmp_path / "postproc"}
    mesh2 = hip_mesh.make_mesh(cfg2["mesh"])
    fixup(cfg2, mesh2)
    cfg2["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
    cfg2["solver"]["momentum"]["tol"] = TRANSPORT_TOL
    cfg2["solver"]["tuning"] = {"mg_line": 0, "mg_tail": 0}
    fields, step, time = hip_solver.run_case(cfg2)
    assert step == rstep == 5 and abs(time - rtime) <= 1e-11 * rtime
    unorm = np.linalg.norm(ref["u"].ravel())
    for key in ("u", "v", "p", "k", "omega", "nu_t"):
        if key not in ref:
            continue
        err = np.linalg.norm((mask_corners(fields[key]) - mask_corners(ref[key])).ravel())
        den = np.linalg.norm(mask_corners(ref[key]).ravel())
        assert err <= FIELD_TOL * den + 1e-12 * unorm, (key, err, den)
    log = np.loadtxt(tmp_path / "postproc" / f"{cfg2['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
    conts = np.array([r["cont"] for r in hist])
    assert np.all(np.abs(log[:, 3] - conts) <= 2e-6 * conts + 1e-12)
    kc = hip_solver.LAST_RUN["kernel_counts"]
    assert kc["cheb4_2rhs"] > 0 and kc["flux_march_multistrip"] > 0 and kc["mg_fused_levels_max"] >= 3 and kc["mg_deferred_resid"] > 0, kc


@pytest.mark.parametrize("prefetch,nt", [(1, 0), (2, 0), (2, 1)])
def test_chebyshev_kernel_forms_keep_the_bits(prefetch, nt, tmp_path):
    """One or two rows of loads in flight per wave and nontemporal stores are scheduling / cache-policy choices:
    the 160 x 96 k-omega run gives the same bits in every field."""
    name = "channel_forced_komega_160x96"
    base = _run(name, tmp_path / "base", tuning={"cheb_prefetch": 1, "nt_stores": 0})
    other = _run(name, tmp_path / "other", tuning={"cheb_prefetch": prefetch, "nt_stores": nt})
    for key in base[0]:
        assert np.array_equal(base[0][key], other[0][key]), key
    assert np.array_equal(base[3], other[3])
    assert other[4]["kernel_counts"]["cheb4_2rhs"] > 0


FUNC_LARGE = {
    "annulus_132x24": cases._half_cylinder_bc(132, 24, True),
    "rect_136x20": {
        "i_min": {"type": "periodic", "range": (0, 19)},
        "i_max": {"type": "periodic", "range": (0, 19)},
        "j_min": {"type": "wall", "range": (0, 135)},
        "j_max": {"type": "symmetry", "range": (0, 135)},
        "pressure_reference": {"cell": (70, 3), "value": 0.25},
    },
}


@pytest.mark.parametrize("tag", list(FUNC_LARGE))
@pytest.mark.parametrize("marching", [True, False])
def test_multistrip_kernels_against_reference_vectors(tag, marching):
    """The reference's own functions (bc, eddy viscosity, conductances, Rhie-Chow flux, upwind advection, Green-Gauss
    gradient, time step, pressure system, correction, continuity: solver.py:322-358, 455-513, 649-669, ...) on meshes
    of 132 / 136 columns, i.e. three column strips of the marching kernels: bit-exact."""
    from cfd_amd import _lib as L
    from test_gpu_parity import _device_for
    g = load_golden(f"func_{tag}.npz")
    bc_cfg = FUNC_LARGE[tag]
    mesh = golden_mesh(g)
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    nu, dt = (float(x) for x in g["params"])
    dev = _device_for(mesh, bc_cfg, nu)
    dev.set_flux_kernels(marching, False)
    dev.upload_fields({k: g[f"in_{k}"].copy() for k in ("u", "v", "p", "k", "omega", "nu_t")})
    dev.apply_flow_bc()
    for name in ("u", "v", "p", "k", "omega", "nu_t"):
        assert np.array_equal(dev.download(L.FIELD_IDS[name]), g[f"bc_{name}"]), f"flow BC {name}"
    dev.apply_turb_bc()
    for name in ("k", "omega"):
        assert np.array_equal(dev.download(L.FIELD_IDS[name]), g[f"tbc_{name}"]), f"turbulence BC {name}"
    dev.eddy_viscosity()
    assert np.array_equal(dev.download(L.F_NUT), g["nu_t"])
    dev.momentum_diagonal(dt, 1.0)
    assert np.array_equal(dev.download(L.F_AI, L.KIND_FACE_I), g["a_i"])
    assert np.array_equal(dev.download(L.F_AJ, L.KIND_FACE_J), g["a_j"])
    assert np.array_equal(dev.download(L.F_AP, L.KIND_CELL), g["a_p"])
    fi, fj = dev.face_flux(rhie_chow=True)
    assert np.array_equal(fi, g["flux_i"]) and np.array_equal(fj, g["flux_j"])
    for name in ("u", "v", "k", "omega"):
        assert np.array_equal(dev.advect(L.FIELD_IDS[name], g["flux_i"], g["flux_j"]), g[f"adv_{name}"]), name
    for name in ("u", "v", "p", "k"):
        assert np.array_equal(dev.grad(L.FIELD_IDS[name]), g[f"grad_{name}"]), name
    d, da, dd = dev.compute_dt(0.5, 0.4, 0.0, 10.0)
    assert d == float(g["dt"]) and (da, dd) == tuple(g["dt_limits"])

    # momentum right-hand sides (marching kernel with three strips) vs the pinned oracle on the same state
    state = {k: g[f"bc_{k}"].copy() for k in ("u", "v", "p")}
    state["k"], state["omega"], state["nu_t"] = g["tbc_k"].copy(), g["tbc_omega"].copy(), g["nu_t"].copy()
    nu_eff = nu + state["nu_t"][1:-1, 1:-1]
    per_i, per_j = orc.periodic_axes(bc_cfg, ni, nj)
    ap = orc.momentum_diag(mesh, nu_eff, dt)
    ofi, ofj = orc.face_flux(state["u"], state["v"], state["p"], mesh, ap)
    gp = orc.green_gauss(state["p"], mesh) + orc.green_gauss((2.0 / 3.0) * state["k"], mesh)
    dev.momentum_rhs(dt)
    for q, (name, fid) in enumerate((("u", L.F_BU), ("v", L.F_BV))):
        src = orc.upwind_advection(state[name], mesh, ofi, ofj) - gp[..., q] + 0.0
        _, b, _ = orc.transport_system(state[name], mesh, nu_eff, dt, src, per_i, per_j)
        assert np.array_equal(dev.download(fid, L.KIND_CELL), b.reshape(ni, nj)), name

    dev.pressure_solve()      # pressure right-hand side through the marching kernel, then MG-Krylov
    pc = dev.download(L.F_PCORR)
    assert rel_l2(pc[1:-1, 1:-1], g["p_corr"][1:-1, 1:-1]) < FIELD_TOL

    dev.upload(L.F_PCORR, g["p_corr"])
    snap = {k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p")}
    for name, key in (("u", "in_u0"), ("v", "in_v0"), ("p", "in_p0")):
        snap[name][1:-1, 1:-1] = g[key]
    cur = {k: dev.download(L.FIELD_IDS[k]) for k in ("u", "v", "p")}
    dev.upload_fields(snap)
    dev.snapshot()
    dev.upload_fields(cur)
    dev.correct(0.7, 0.6, 0.3)
    for name in ("u", "v", "p"):
        assert np.array_equal(dev.download(L.FIELD_IDS[name])[1:-1, 1:-1], g[f"corr_{name}"][1:-1, 1:-1]), name
    dev.upload_fields(state)
    dev.momentum_diagonal(dt, 1.0)
    cont = dev.continuity()
    assert abs(cont - float(g["continuity"])) <= 1e-12 * float(g["continuity"])
    kc = dev.kernel_counts()
    if marching:
        assert kc["flux_march_multistrip"] > 0 and kc["flux_tile"] == 0, kc
    dev.close()


# ---------------------------------------------------------------------------------------------------------------------
# stretched meshes: alternating line smoother (the reference's real cases are wall-graded, Runs/*/config.py ratio_j 1.03)
# ---------------------------------------------------------------------------------------------------------------------
SEMI_CASES = {
    # tag: (mesh, boundary, periodic i, most MG-PCG iterations at rtol 1e-8 with the semi-coarsened line multigrid)
    "channel_128_r1.04": lambda: (orc.rect_mesh(1.0, 1.0, 128, 128, 1.0, 1.04), "per_i", True, 14),
    "channel_256_Lx4": lambda: (orc.rect_mesh(4.0, 1.0, 256, 256, 1.0, 1.0118), "per_i", True, 13),
    "annulus_132x24": lambda: (orc.annulus_mesh(0.5, 6.0, 132, 24, 0.0, float(np.pi), 1.02, 1.0), "cyl", False, 8),
    "annulus_512x256": lambda: (orc.annulus_mesh(0.5, 20.0, 512, 256, 0.0, float(np.pi), 1.012, 1.0), "cyl", False, 9),
    "inout_300x128": lambda: (orc.rect_mesh(2.0, 1.0, 300, 128, 1.0, 1.02), "inout", False, 11),
    "channel_1100x96_r1.05": lambda: (orc.rect_mesh(3.0, 1.0, 1100, 96, 1.0, 1.05), "per_i", True, 12),
    "channel_2200x64_r1.08": lambda: (orc.rect_mesh(3.0, 1.0, 2200, 64, 1.0, 1.08), "per_i", True, 10),
    "channel_4200x32_r1.1": lambda: (orc.rect_mesh(6.0, 1.0, 4200, 32, 1.0, 1.1), "per_i", True, 10),
}


@pytest.mark.parametrize("tag", sorted(SEMI_CASES))
def test_semi_coarsened_line_multigrid(tag):
    """Stretched meshes are cycled by zebra whole-line relaxation along i + semi-coarsening along j (csrc/semi.inc):
    lines of 128 .. 4200 cells (one to eight cells per thread, 4 and 16 waves per line, periodic and not), the solution
    of the double-precision system (true residual from the scipy matrix), a symmetric preconditioner (PCG, and
    (r1, B r2) = (B r1, r2)), no more iterations than the alternating block line smoother it replaces."""
    from test_gpu_parity import _bc_variant, _device_for
    mesh, kind, per_i, bound = SEMI_CASES[tag]()
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    bc = cases._half_cylinder_bc(ni, nj, False) if kind == "cyl" else _bc_variant(kind, ni, nj)
    res = {}
    for semi in (1, 0):
        dev = _device_for(mesh, bc, 1e-3, turbulent=False)
        dev.set_option("mg_semi", semi)
        dev.upload_fields({k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p", "nu_t")})
        dev.momentum_diagonal(3e-3, 1.0)
        ci, cj, diag, _ = dev.pressure_system()
        a = orc.five_point_matrix(diag, ci[:-1], ci[1:], cj[:, :-1], cj[:, 1:], per_i, False)
        rng = np.random.default_rng(1)
        x_true = rng.standard_normal((ni, nj))
        b = a.dot(x_true.ravel()).reshape(ni, nj)
        sol, iters, relres = dev.pressure_krylov(b, tol=1e-8, maxiter=400)
        true_res = np.linalg.norm(a.dot(sol.ravel()) - b.ravel()) / np.linalg.norm(b.ravel())
        if semi:
            r1, r2 = rng.standard_normal((ni, nj)), rng.standard_normal((ni, nj))
            z1, z2 = dev.vcycle(r1), dev.vcycle(r2)
            sym = abs(np.vdot(r1, z2) - np.vdot(z1, r2)) / abs(np.vdot(r1, z2))
            assert np.vdot(r1, z1) > 0.0 and np.vdot(r2, z2) > 0.0 and sym < 1e-5, sym
        res[semi] = (iters, relres, true_res, dev.kernel_counts()["mg_semi_levels"], dev.stats()["p_solver"])
        dev.close()
    it, rr, tr, levels, solver = res[1]
    assert levels >= 3 and res[0][3] == 0 and solver == "mg-pcg", res
    assert rr <= 1e-8 and tr <= 2e-8, res
    assert it <= bound and it <= res[0][0], res


def _krylov_case(mesh, bc, per_i, per_j, mode, tol=1e-8, options=None):
    from test_gpu_parity import _device_for
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    dev = _device_for(mesh, bc, 1e-3, turbulent=False)
    dev.set_option("mg_line", mode)
    for name, value in (options or {}).items():
        dev.set_option(name, value)
    dev.upload_fields({k: np.zeros((ni + 2, nj + 2)) for k in ("u", "v", "p", "nu_t")})
    dev.momentum_diagonal(3e-3, 1.0)
    ci, cj, diag, _ = dev.pressure_system()
    a = orc.five_point_matrix(diag, ci[:-1], ci[1:], cj[:, :-1], cj[:, 1:], per_i, per_j)
    x_true = np.random.default_rng(1).standard_normal((ni, nj))
    b = a.dot(x_true.ravel()).reshape(ni, nj)
    sol, iters, relres = dev.pressure_krylov(b, tol=tol, maxiter=400)
    true_res = np.linalg.norm(a.dot(sol.ravel()) - b.ravel()) / np.linalg.norm(b.ravel())
    levels = dev.level_smoothers()
    solver = dev.stats()["p_solver"]
    dev.close()
    return iters, relres, true_res, levels, solver


@pytest.mark.parametrize("tag", ["channel_128_r1.04", "channel_256_Lx4", "annulus_132x24", "annulus_512x256", "inout_257x129"])
def test_line_smoother_on_stretched_meshes(tag):
    """MG-Krylov iterations at rtol 1e-8 with point Jacobi and with the alternating line smoother (chosen by itself from
    the level's coupling ratio): the same solution (true residual from the scipy matrix), 3-5 x fewer iterations, at most
    15 on the wall-graded 128 x 128 channel with growth ratio 1.04 (max / min dy = 146) that needed ~75."""
    from test_gpu_parity import _bc_variant
    if tag == "channel_128_r1.04":
        mesh, bc, per, bound = orc.rect_mesh(1.0, 1.0, 128, 128, 1.0, 1.04), _bc_variant("per_i", 128, 128), (True, False), 15
    elif tag == "channel_256_Lx4":
        mesh, bc, per, bound = orc.rect_mesh(4.0, 1.0, 256, 256, 1.0, 1.0118), _bc_variant("per_i", 256, 256), (True, False), 24
    elif tag == "annulus_132x24":
        mesh, bc, per, bound = (orc.annulus_mesh(0.5, 6.0, 132, 24, 0.0, float(np.pi), 1.02, 1.0),
                                cases._half_cylinder_bc(132, 24, False), (False, False), 16)
    elif tag == "annulus_512x256":
        mesh, bc, per, bound = (orc.annulus_mesh(0.5, 20.0, 512, 256, 0.0, float(np.pi), 1.012, 1.0),
                                cases._half_cylinder_bc(512, 256, False), (False, False), 15)
    else:
        mesh, bc, per, bound = orc.rect_mesh(2.0, 1.0, 257, 129, 1.0, 1.02), _bc_variant("inout", 257, 129), (False, False), 15
    it_point, rr_point, res_point, lv_point, _ = _krylov_case(mesh, bc, per[0], per[1], 0)
    it_line, rr_line, res_line, lv_line, solver = _krylov_case(mesh, bc, per[0], per[1], -1)
    assert lv_line[0][2] > 10.0 and lv_line[0][3] and not any(l[3] for l in lv_point), (lv_line, lv_point)
    assert res_point <= 2e-8 and res_line <= 2e-8 and rr_line <= 1e-8
    assert solver == "mg-pcg"                                  # the line-smoothed V-cycle is a symmetric preconditioner
    assert it_line <= bound and 2 * it_line <= it_point, (it_line, it_point)


def test_line_smoother_leaves_uniform_meshes_alone():
    from test_gpu_parity import _bc_variant
    mesh = orc.rect_mesh(1.0, 1.0, 256, 256, 1.0, 1.0)
    it, rr, res, levels, _ = _krylov_case(mesh, _bc_variant("per_i", 256, 256), True, False, -1)
    # (ratio 2 = the wall rows, which have one j-face only)
    assert not any(l[3] for l in levels) and max(l[2] for l in levels) < 2.5 and it <= 7


def test_graded_komega_channel_against_oracle(tmp_path):
    """128 x 128 k-omega channel, wall-graded with growth ratio 1.04 (the stretching of the reference's own runs, at a
    size where the line-smoothed multigrid levels, the multi-strip flux kernels and BiCGStab / Chebyshev all run):
    three outer steps against the pinned CPU oracle with direct solves."""
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver

    def build():
        return cases.channel_forced(n=128, steps=3, turbulent=True, ratio_j=1.04, name="channel_graded_komega_128")

    cfg, mp, kind = build()
    mesh = orc.rect_mesh(**mp)
    cases.perturbed_parabola(cfg, mesh)
    ref, rstep, rtime, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True)
    cfg2, mp2, _ = build()
    cfg2["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp2, "path": tmp_path / "mesh" / "mesh.npz"}
    cfg2["paths"] = {"case_dir": tmp_path, "mesh_dir": tmp_path / "mesh", "states_dir": tmp_path / "states",
                     "post_dir": tmp_path / "postproc"}
    mesh2 = hip_mesh.make_mesh(cfg2["mesh"])
    cases.perturbed_parabola(cfg2, mesh2)
    cfg2["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 4000}
    cfg2["solver"]["momentum"]["tol"] = TRANSPORT_TOL
    cfg2["solver"]["momentum"]["maxiter"] = 4000
    fields, step, time = hip_solver.run_case(cfg2)
    assert step == rstep == 3 and abs(time - rtime) <= 1e-11 * rtime
    unorm = np.linalg.norm(ref["u"].ravel())
    for key in ("u", "v", "p", "k", "omega", "nu_t"):
        err = np.linalg.norm((mask_corners(fields[key]) - mask_corners(ref[key])).ravel())
        den = np.linalg.norm(mask_corners(ref[key]).ravel())
        assert err <= FIELD_TOL * den + 1e-12 * unorm, (key, err, den)
    log = np.loadtxt(tmp_path / "postproc" / f"{cfg2['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
    conts = np.array([r["cont"] for r in hist])
    assert np.all(np.abs(log[:, 3] - conts) <= 2e-6 * conts + 1e-12)
    last = hip_solver.LAST_RUN
    assert last["kernel_counts"]["mg_line_levels_max"] >= 3 and last["stats"]["p_unconverged"] == 0, last
    assert last["stats"]["p_iters"] <= 40, last["stats"]           # rtol 1e-13; point Jacobi needs > 150 here


@pytest.mark.parametrize("turbulent", [False, True])
def test_lockstep_workgroups_match_independent_waves(turbulent, tmp_path):
    """Marching kernels as lockstep workgroups of 8 (or 4) neighbouring strips (tuning wide_mask, the default on meshes
    of >= 16 strips) against the independent-wave forms on a 1024 x 64 channel: same iteration counts, fields equal to
    rounding (the workgroup sums of the inner products are taken in another order)."""
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    runs = {}
    for mask in (0, 411, 511):
        cfg, mp, kind = cases.channel_forced(n=64, steps=3, turbulent=turbulent, ratio_j=1.0)
        mp.update({"count_i": 1024, "length_i": 16.0 * mp["length_i"]})
        for side, n in (("i_min", 64), ("i_max", 64), ("j_min", 1024), ("j_max", 1024)):
            cfg["boundary"][side]["range"] = (0, n - 1)
        work = tmp_path / f"m{mask}"
        cfg["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp, "path": work / "mesh" / "mesh.npz"}
        cfg["paths"] = {"case_dir": work, "mesh_dir": work / "mesh", "states_dir": work / "states", "post_dir": work / "postproc"}
        mesh = hip_mesh.make_mesh(cfg["mesh"])
        cases.perturbed_parabola(cfg, mesh)
        cfg["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
        cfg["solver"]["momentum"]["tol"] = TRANSPORT_TOL
        cfg["solver"]["tuning"] = {"mg_line": 0, "wide_mask": mask}
        fields, step, time = hip_solver.run_case(cfg)
        log = np.loadtxt(work / "postproc" / f"{cfg['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
        runs[mask] = (fields, log, dict(hip_solver.LAST_RUN))
    base = runs[0]
    assert base[2]["kernel_counts"]["cheb4_2rhs"] > 0 and base[2]["kernel_counts"]["cg_pq"] > 0
    for mask in (411, 511):
        fields, log, last = runs[mask]
        for key in base[0]:
            if key == "p_corr":
                continue
            assert rel_l2(mask_corners(fields[key]), mask_corners(base[0][key])) < FIELD_TOL, (mask, key)
        assert np.array_equal(log[:, 2], base[1][:, 2])
        assert np.all(np.abs(log[:, 3] - base[1][:, 3]) <= 1e-9 * base[1][:, 3] + 1e-14)
        assert last["stats"]["p_iters"] == base[2]["stats"]["p_iters"]


def _run_pair(builder, fixup, tmp_path, extra=None):
    """The same case with the correction fused into the closing flux pass (default) and with the separate kernels."""
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    runs = []
    for fused in (1, 0):
        cfg, mp, kind = builder()
        work = tmp_path / f"f{fused}"
        gen = hip_mesh.generate_rect_mesh if kind == "rect" else hip_mesh.generate_annulus_mesh
        cfg["mesh"] = {"generator": gen, "params": mp, "path": work / "mesh" / "mesh.npz"}
        cfg["paths"] = {"case_dir": work, "mesh_dir": work / "mesh", "states_dir": work / "states", "post_dir": work / "postproc"}
        mesh = hip_mesh.make_mesh(cfg["mesh"])
        if fixup is not None:
            fixup(cfg, mesh)
        cfg["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
        cfg["solver"]["momentum"]["tol"] = TRANSPORT_TOL
        cfg["solver"]["tuning"] = dict({"mg_line": 0, "fuse_correct": fused}, **(extra or {}))
        fields, step, time = hip_solver.run_case(cfg)
        log = np.loadtxt(work / "postproc" / f"{cfg['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
        runs.append((fields, log, dict(hip_solver.LAST_RUN)))
    return runs


def _wide_channel(turbulent):
    cfg, mp, kind = cases.channel_forced(n=64, steps=3, turbulent=turbulent, ratio_j=1.0)
    mp.update({"count_i": 1024, "length_i": 16.0 * mp["length_i"]})
    for side, n in (("i_min", 64), ("i_max", 64), ("j_min", 1024), ("j_max", 1024)):
        cfg["boundary"][side]["range"] = (0, n - 1)
    return cfg, mp, kind


def _square_channel(turbulent):
    return cases.channel_forced(n=256, steps=3, turbulent=turbulent, ratio_j=1.01)


DEFAULT_PATH_CASES = {
    # 17 strips of 60 columns: lockstep workgroups of 8 (wide_mask default), the fused closing pass, point-smoothed multigrid
    "channel_1024x64": (lambda: _wide_channel(False), {"mg_line": 0}),
    "channel_komega_1024x64": (lambda: _wide_channel(True), {"mg_line": 0}),
    # wall-graded (dy max / min = 12.6): every tuning at its default, the pressure solve through the line-smoothed multigrid
    "channel_256x256": (lambda: _square_channel(False), {}),
    "channel_komega_256x256": (lambda: _square_channel(True), {}),
}
# p of the 256^2 k-omega run: the pressure correction turns an error du of the predictor into a_P h du, and with
# nu_t up to 3.5e3 on cells of 8.5e-4 (a_P h ~ nu_t / dy <= 4e6, |u| / |p| = 280 in rms) even du = eps |u| may show up
# as 2e-7 |p|: 1e-10 is not attainable for this field on this case by any solver that is not bitwise the reference's.
# Measured: 5.2e-10 with the transport solves at 1e-15 (2.4e-8 at 1e-13).  Asserted: 2e-9; every other field 1e-10.
P_BAR = {"channel_komega_256x256": 2.0e-9}


@pytest.mark.parametrize("name", list(DEFAULT_PATH_CASES))
def test_default_kernel_forms_against_oracle(name, tmp_path):
    """The kernel forms the 8192^2 run takes by default (lockstep workgroups, correction fused into the closing flux
    pass, single-precision multigrid storage) against the pinned CPU oracle with direct solves: three outer steps,
    every solved field to 1e-10 -- the lockstep and fused forms are otherwise only compared with other GPU forms."""
    builder, tuning = DEFAULT_PATH_CASES[name]
    cfg, mp, kind = builder()
    mesh = orc.rect_mesh(**mp)
    cases.perturbed_parabola(cfg, mesh)
    ref, rstep, rtime, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True)

    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    cfg2, mp2, _ = builder()
    cfg2["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp2, "path": tmp_path / "mesh" / "mesh.npz"}
    cfg2["paths"] = {"case_dir": tmp_path, "mesh_dir": tmp_path / "mesh", "states_dir": tmp_path / "states",
                     "post_dir": tmp_path / "postproc"}
    mesh2 = hip_mesh.make_mesh(cfg2["mesh"])
    cases.perturbed_parabola(cfg2, mesh2)
    cfg2["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
    cfg2["solver"]["momentum"]["tol"] = TRANSPORT_TOL
    if tuning:
        cfg2["solver"]["tuning"] = dict(tuning)
    fields, step, time = hip_solver.run_case(cfg2)
    last = dict(hip_solver.LAST_RUN)
    assert step == rstep == 3 and abs(time - rtime) <= 1e-11 * rtime
    unorm = np.linalg.norm(ref["u"].ravel())
    for key in ("u", "v", "p", "k", "omega", "nu_t"):
        if key not in ref:
            continue
        err = np.linalg.norm((mask_corners(fields[key]) - mask_corners(ref[key])).ravel())
        den = np.linalg.norm(mask_corners(ref[key]).ravel())
        bar = P_BAR.get(name, FIELD_TOL) if key == "p" else FIELD_TOL
        assert err <= bar * den + 1e-12 * unorm, (key, err, den)
    log = np.loadtxt(tmp_path / "postproc" / f"{cfg2['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
    conts = np.array([r["cont"] for r in hist])
    assert np.all(np.abs(log[:, 3] - conts) <= 2e-6 * conts + 1e-12)
    kc = last["kernel_counts"]
    assert kc["correct_flux"] > 0 and kc["cheb4_2rhs"] > 0 and last["stats"]["p_unconverged"] == 0, (kc, last["stats"])
    if "1024x64" in name:
        assert kc["flux_march_multistrip"] > 0 and kc["mg_fused_levels_max"] >= 2, kc
    else:
        assert kc["mg_semi_levels"] >= 2 or kc["mg_line_levels_max"] >= 2, kc


FUSED_CORRECTION_CASES = {
    "channel_forced_160x96": cases.LARGE_END_TO_END["channel_forced_160x96"],                  # periodic i, walls, three strips
    "channel_forced_komega_160x96": cases.LARGE_END_TO_END["channel_forced_komega_160x96"],    # + grad(2/3 k)
    "inlet_outlet_160x96": cases.LARGE_END_TO_END["inlet_outlet_160x96"],                      # ghost columns from the tables
    "half_cylinder_132x24": (lambda: cases.half_cylinder(nr=132, nt=24, steps=3), None),       # symmetry rows, curvilinear
    "half_cylinder_komega_132x24": (lambda: cases.half_cylinder(nr=132, nt=24, steps=3, turbulent=True), None),
    "channel_1024x64": (lambda: _wide_channel(False), cases.perturbed_parabola),               # lockstep workgroups
    "channel_komega_1024x64": (lambda: _wide_channel(True), cases.perturbed_parabola),
}


@pytest.mark.parametrize("name", list(FUSED_CORRECTION_CASES))
def test_correction_fused_into_the_closing_flux_pass(name, tmp_path):
    """k_correct_flux (velocity / pressure correction, ghost cells from the BC tables in registers, momentum right-hand
    sides, continuity sum and time-step minima in one marching pass) against k_correct + k_flow_bc + k_flux_march<2>:
    the same bits in every field; the continuity sum is taken over other strip widths (rounding)."""
    builder, fixup = FUSED_CORRECTION_CASES[name]
    # The turbulent half cylinder blows up in its first step, in the reference as here (nu_t ~ 1e9, dt ~ 1e-15, continuity
    # ~ 1e6): a transport solve of its, asked for 1e-15, stalls at 7e-12 .. 2e-9 after 10000 Jacobi-BiCGStab iterations with the default
    # single-precision multigrid faces (at 1e-13 with double faces: other last bits of p).  Such a stall below the rounding
    # level is accepted and counted (transport_stalled), so the default options run this case too.
    # Whether one does depends on the last bits of p (it did with the V-cycle tail's first arithmetic, it does not with the
    # division-free one): both runs of the pair must agree on it, and a warning says so when it happens.
    if name == "half_cylinder_komega_132x24":
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            (f1, log1, last1), (f0, log0, last0) = _run_pair(builder, fixup, tmp_path, None)
        stalled = last1["stats"]["transport_stalled"]
        assert last0["stats"]["transport_stalled"] == stalled
        assert (stalled > 0) == any(issubclass(w.category, RuntimeWarning) and "stalled" in str(w.message) for w in caught)
    else:
        (f1, log1, last1), (f0, log0, last0) = _run_pair(builder, fixup, tmp_path, None)
        assert last1["stats"]["transport_stalled"] == 0
    assert last1["kernel_counts"]["correct_flux"] > 0 and last0["kernel_counts"]["correct_flux"] == 0
    for key in f0:
        assert np.array_equal(mask_corners(f1[key]), mask_corners(f0[key])), key
    assert np.array_equal(log1[:, [0, 2]], log0[:, [0, 2]])
    assert np.all(np.abs(log1[:, 3] - log0[:, 3]) <= 1e-11 * log0[:, 3] + 1e-300)
    assert last1["stats"]["p_iters"] == last0["stats"]["p_iters"]


@pytest.mark.parametrize("shape", [(8, 4), (59, 7), (60, 8), (61, 9), (120, 16), (121, 33), (180, 5)])
@pytest.mark.parametrize("kind", ["periodic", "inlet_outlet"])
def test_fused_correction_at_strip_and_segment_edges(kind, shape, tmp_path):
    """Mesh sizes around the 60-column strips and the row segments of k_correct_flux (one strip exactly full, one cell
    more, the smallest mesh the fused pass takes, fewer rows than a segment): the same bits as the separate kernels."""
    ni, nj = shape
    builder = (lambda: cases.lam_channel(ni, nj, steps=3)) if kind == "periodic" else (lambda: cases.inlet_outlet_channel(ni, nj, steps=3))
    (f1, log1, last1), (f0, log0, last0) = _run_pair(builder, None, tmp_path)
    assert last1["kernel_counts"]["correct_flux"] > 0 and last0["kernel_counts"]["correct_flux"] == 0
    for key in f0:
        assert np.array_equal(mask_corners(f1[key]), mask_corners(f0[key])), key
    assert np.array_equal(log1[:, [0, 2]], log0[:, [0, 2]])
    assert np.all(np.abs(log1[:, 3] - log0[:, 3]) <= 1e-11 * log0[:, 3] + 1e-300)


@pytest.mark.parametrize("ratio_i,expect", [(1.0 + 1.0e-8, "pcg"), (1.02, "fallback")])
def test_nearly_symmetric_seam_goes_through_pcg_with_a_fallback(ratio_i, expect):
    """A periodic seam whose two coefficient copies differ by more than rounding (the reference's centroid formula loses
    digits on thin cells: 1e-6 on the wall-graded 8192^2 mesh; here a mesh graded by 1e-8 per cell along the periodic
    axis) is solved by MG-PCG, one V-cycle per iteration, as long as the relative difference stays below p_sym_tol;
    a strongly graded periodic axis pushed through the same door (p_sym_tol = 1e-2) either converges or is handed to
    MG-BiCGStab from the saved right-hand side -- the true residual meets the tolerance either way."""
    from test_gpu_parity import _bc_variant
    mesh = orc.rect_mesh(1.0, 1.0, 192, 96, ratio_i, 1.02)
    bc = _bc_variant("per_i", 192, 96)
    opts = {} if expect == "pcg" else {"p_sym_tol": 1.0e-2}
    it, rr, res, levels, solver = _krylov_case(mesh, bc, True, False, 0, options=opts)
    it_b, rr_b, res_b, _, solver_b = _krylov_case(mesh, bc, True, False, 0, options={"p_sym_tol": 0.0})
    assert solver_b == "mg-bicgstab" and res_b <= 2e-8
    assert res <= 2e-8 and rr <= 1e-8, (res, rr, it, solver)
    if expect == "pcg":
        assert solver == "mg-pcg" and it <= 2 * it_b + 2, (it, it_b)


def test_fused_correction_at_the_benchmark_size():
    """BASELINE.json's 8192^2 workload, three outer steps: the fused correction / flux kernel (lockstep workgroups of four
    strips, 137 strips, 256 segments) leaves the same bits in u, v, p as the three kernels it replaces, and the run takes
    the same iteration counts."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for fused in (1, 0):
        dev, ctl, cfg = bench.setup_device(8192, False, 2, 1e-8, 0, tuning={"fuse_correct": fused})
        for step in range(3):
            advance_one_step(dev, ctl, step)
        kc = dev.kernel_counts()
        out[fused] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], dev.stats(), kc)
        dev.close()
    # (an inner iteration whose speculative replay had to be repeated host-checked launches its kernels twice)
    assert out[1][2]["correct_flux"] == 6 + out[1][2]["speculative_redone"] and out[0][2]["correct_flux"] == 0
    assert out[1][2]["speculative_redone"] == out[0][2]["speculative_redone"] <= 1
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(mask_corners(a), mask_corners(b))
    assert out[1][1]["p_iters"] == out[0][1]["p_iters"] and out[1][1]["mom_iters"] == out[0][1]["mom_iters"]


@pytest.mark.parametrize("n", [750, 1001])
def test_odd_levels_scale_coarse_faces_by_the_centre_distance(n):
    """Hierarchies with odd levels (750 -> 375 -> 188 ..., 1001 -> 501 -> 251 ...): the coarse faces next to the
    one-cell last aggregate are scaled by the distance between the aggregate centres (mg_odd_fix = 1, default).  The
    preconditioner changes, the solution does not: same fields to the solver tolerance, and not more MG-PCG iterations
    than with the plain factor-two scaling."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for fix in (1, 0):
        dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-10, 0, tuning={"mg_odd_fix": fix})
        its = 0
        for step in range(3):
            advance_one_step(dev, ctl, step)
            its += dev.stats()["p_iters"]
        out[fix] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], its)
        dev.close()
    assert out[1][1] <= out[0][1], (out[1][1], out[0][1])
    for a, b in zip(out[1][0], out[0][0]):
        a, b = mask_corners(a), mask_corners(b)
        scale = max(np.abs(a).max(), 1e-30)
        # (two preconditioners, solves stopped at 1e-10 relative to |b|: the fields agree to what that leaves in the
        # smallest field, v; which iterate a solve is taken from also depends on the replayed schedule)
        assert np.abs(a - b).max() <= 3e-7 * scale, np.abs(a - b).max() / scale


def test_speculative_inner_iterations():
    """Inner iterations that replay the previous iteration's solver schedule with ONE host read at their end (option
    `speculate`, default on) against the host-checked path: same fields to the solver tolerance, same iteration counts in
    steady state, <= 3 stream synchronisations per outer step instead of 8 - 11; and a schedule that no longer fits
    (pressure tolerance tightened between two steps: the replayed count is too short) makes the iteration repeat itself
    host-checked -- the stopping tests are never skipped."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for spec in (1, 0):
        dev, ctl, cfg = bench.setup_device(256, False, 2, 1e-8, 0, tuning={"speculate": spec})
        for step in range(6):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        kc0 = dev.kernel_counts()
        dev.host_counters(reset=True)
        for step in range(6, 12):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        launches, syncs = dev.host_counters()
        st = dev.stats()
        kc1 = dev.kernel_counts()
        # tighten the pressure tolerance: five iterations are not enough any more
        dev.set_solver(1e-8, 200, 1e-11, 200, 1e-8, 200, True)
        advance_one_step(dev, ctl, 12)
        st2 = dev.stats()
        kc2 = dev.kernel_counts()
        out[spec] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], st, st2, kc0, kc1, kc2, (syncs - 1) / 6.0)
        dev.close()
    f1, st1, st1b, k0, k1, k2, syncs1 = out[1]
    f0, st0, st0b, _, q1, q2, syncs0 = out[0]
    assert q2["speculative_iterations"] == 0 and q2["speculative_redone"] == 0
    assert k1["speculative_iterations"] - k0["speculative_iterations"] == 12 and k1["speculative_redone"] == k0["speculative_redone"]
    assert syncs1 <= 3.0 < syncs0, (syncs1, syncs0)
    assert st1["p_iters"] == st0["p_iters"] and st1["mom_iters"] == st0["mom_iters"]
    assert 0.0 < st1["p_relres"] <= 1e-8 and all(0.0 < r <= 1e-8 for r in st1["mom_relres"])        # filled in by the closing read
    # the step after the tolerance change: the first replay fails its test and is repeated; both paths end converged
    assert k2["speculative_redone"] == k1["speculative_redone"] + 1, (k1, k2)
    assert st1b["p_iters"] == st0b["p_iters"] > st1["p_iters"] and st1b["p_relres"] <= 1e-11
    for a, b in zip(f1, f0):
        assert rel_l2(mask_corners(a), mask_corners(b)) <= 1e-7


@pytest.mark.parametrize("n,p_tol,spec", [(256, 1e-8, 1), (256, 1e-13, 0), (192, 1e-13, 1)])
def test_pcg_solution_summed_from_the_kept_directions(n, p_tol, spec):
    """Option `pcg_xsum` (default on): the fused MG-PCG direction step leaves x alone, the directions of the solve stay in
    a ring of six planes and x = sum alpha_k p_k is summed once at the end in the order of the stepwise updates -- the
    same bits as x += alpha p in every step.  The tight tolerance takes more than six iterations per solve (the ring is
    emptied into x on the way), 192^2 has an odd multigrid level and a nullspace shift folded into the last sum."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for xs in (1, 0):
        dev, ctl, cfg = bench.setup_device(n, False, 2, 1e-8, 0, tuning={"pcg_xsum": xs, "speculate": spec})
        dev.set_solver(1e-8, 200, p_tol, 200, 1e-8, 200, True)
        for step in range(4):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        out[xs] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P, L.F_PCORR)], dev.stats(), dev.kernel_counts())
        dev.close()
    (f1, st1, k1), (f0, st0, k0) = out[1], out[0]
    assert k1["cg_pq"] == k0["cg_pq"] > 0 and st1["p_iters"] == st0["p_iters"]
    assert k1["pcg_xsum"] > 0 and k0["pcg_xsum"] == 0
    assert (st1["p_iters"] > 6) == (p_tol < 1e-10), st1["p_iters"]
    for a, b in zip(f1, f0):
        assert np.array_equal(mask_corners(a), mask_corners(b)), np.abs(a - b).max()


@pytest.mark.parametrize("kind", ["uniform", "graded", "komega", "annulus"])
def test_axis_aligned_meshes_leave_the_normal_planes_alone(kind):
    """On a mesh whose face area vectors are (S, 0) / (0, S) bit for bit (every rectangular mesh, graded or not; checked
    on the device once per mesh) the marching flux kernels take the normals from the area planes they read anyway and do
    not read the four normal planes (option `rect_normals` >= 1): the pressure right-hand side pass reports 80 algorithmic
    bytes per cell instead of 112.  Where the areas are separable too -- S_i one value per row, S_j one per column, as on
    every tensor-product node grid -- they come from one entry per row / one register per lane (`rect_normals` = 2, the
    default): 64 bytes.  Every field is the one the general kernels give, bit for bit.  A curvilinear mesh (annulus) keeps
    the general kernels."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for rn in (2, 1, 0):
        if kind == "annulus":
            dev, ctl, cfg = bench.setup_annulus(192, 96, 2, 1e-8, 0, tuning={"rect_normals": rn})
        else:
            dev, ctl, cfg = bench.setup_device(192, kind == "komega", 2, 1e-8, 0, tuning={"rect_normals": rn}, graded=kind == "graded")
        dev.probe(13)
        for step in range(3):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        n, _, bpc = dev.probe_read(with_bytes=True)
        dev.probe(-1)
        names = (L.F_U, L.F_V, L.F_P) + ((L.F_K, L.F_OMEGA, L.F_NUT) if kind == "komega" else ())
        out[rn] = ([dev.download(f) for f in names], n, bpc, dev.kernel_counts())
        dev.close()
    f0, n0, b0, k0 = out[0]
    assert n0 > 0 and b0 == 112.0 and k0["flux_march"] > 0
    for rn, expect in ((1, 80.0), (2, 64.0)):
        f1, n1, b1, k1 = out[rn]
        assert n1 == n0 and b1 == (112.0 if kind == "annulus" else expect), (rn, n1, n0, b1)
        assert k1["flux_march"] == k0["flux_march"] and k1["correct_flux"] == k0["correct_flux"]
        assert k1["mesh_form"] == (0 if kind == "annulus" else rn) and k0["mesh_form"] == 0
        if kind != "annulus":
            assert k1["correct_flux"] > 0
        for a, b in zip(f1, f0):
            assert np.array_equal(mask_corners(a), mask_corners(b)), (rn, np.abs(a - b).max())


@pytest.mark.parametrize("kind", ["graded", "annulus"])
def test_residual_update_inside_the_semi_coarsened_way_down(kind):
    """Stretched meshes (semi-coarsened line multigrid): on the PCG iterations whose |r| nobody reads, r = r - alpha q is
    formed by the two kernels of the finest level's way down (k_semi_even_down / k_semi_odd_down<.., true>: they read
    r_old and q where they would read r and store their rows of r) instead of by a k_cg_r pass of its own (option
    `semi_resid`, default on).  Same expression, same operands: fields, iteration counts and residuals are the ones the
    separate pass gives, bit for bit."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for sr in (1, 0):
        if kind == "annulus":
            dev, ctl, cfg = bench.setup_annulus(256, 128, 2, 1e-8, 0, tuning={"semi_resid": sr})
        else:
            dev, ctl, cfg = bench.setup_device(256, False, 2, 1e-8, 0, tuning={"semi_resid": sr}, graded=True)
        hist = []
        for step in range(4):
            r = advance_one_step(dev, ctl, step)
            hist.append((r["dt"], r["cont"], r["mom_u"], r["mom_v"], dev.stats()["p_iters"]))
        dev.synchronize()
        out[sr] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], hist, dev.kernel_counts())
        dev.close()
    f1, h1, k1 = out[1]
    f0, h0, k0 = out[0]
    assert k1["mg_semi_levels"] > 1 and k0["mg_semi_levels"] == k1["mg_semi_levels"]
    assert k1["mg_deferred_resid"] > 0 and k0["mg_deferred_resid"] == 0, (k1, k0)
    assert k1["vcycles"] == k0["vcycles"]
    assert h1 == h0, (h1, h0)
    for a, b in zip(f1, f0):
        assert np.array_equal(mask_corners(a), mask_corners(b)), np.abs(a - b).max()


def test_laminar_diffusive_time_step_minimum_is_kept():
    """In a laminar run min(cfl_diff l^2 / nu) is a constant of mesh, nu and cfl_diff: the first look-ahead pass of
    k_correct_flux computes it, the host reads it with the continuity residual, and the later passes leave the l plane
    alone (8 B per cell; option `lmin_const`, default on) while k_min_finish puts the kept double in its slot.  Time
    steps and fields are the ones the option off gives, bit for bit, also across a change of cfl_diff (new key: one
    fresh pass)."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for lc in (1, 0):
        dev, ctl, cfg = bench.setup_device(192, False, 2, 1e-8, 0, tuning={"lmin_const": lc}, graded=True)
        dev.probe(2)
        dts = []
        for step in range(6):
            if step == 3:
                ctl.cfl_diff = 0.5 * ctl.cfl_diff
            r = advance_one_step(dev, ctl, step)
            dts.append((r["dt"], r["dt_adv"], r["dt_diff"]))
        dev.synchronize()
        n, _, bpc = dev.probe_read(with_bytes=True)
        dev.probe(-1)
        out[lc] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], dts, n, bpc)
        dev.close()
    f1, d1, n1, b1 = out[1]
    f0, d0, n0, b0 = out[0]
    assert d1 == d0, (d1, d0)
    assert d0[3][2] == 0.5 * d0[2][2]
    assert n1 == n0 and b0 - 8.0 < b1 < b0, (n1, n0, b1, b0)     # mean bytes per cell over the probed launches
    for a, b in zip(f1, f0):
        assert np.array_equal(mask_corners(a), mask_corners(b)), np.abs(a - b).max()


@pytest.mark.parametrize("turbulent,spec", [(False, 1), (False, 0), (True, 1)])
def test_six_fused_chebyshev_iterations_for_a_single_right_hand_side(turbulent, spec):
    """Option `cheb_k6` (default on, single rank, lockstep form: meshes of >= 832 columns): once one of the two right-hand
    sides of a Chebyshev solve has converged, the launches that carry the other one alone fuse six iterations instead of
    four where that saves a launch (k_chebk<1, 6>: 12 halo lanes instead of 8 for the same 64 B per cell; 10 - 12
    iterations to go: 6 + 4 / 6 + 6 instead of 4 + 4 + 4).  Same expressions on the same operands, so
    every iterate has the bits of the four-iteration sequence; what may differ is the iterate a right-hand side is taken
    from (residuals are summed at the first and last iterate of a launch): with the solves driven to 1e-12 the fields
    agree to 1e-9, host-checked and replayed schedules alike, in fewer launches."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for k6 in (1, 0):
        dev, ctl, cfg = bench.setup_device(2048, turbulent, 2, 1e-8, 0, tuning={"cheb_k6": k6, "speculate": spec})
        dev.set_solver(1e-12, 200, 1e-12, 200, 1e-12, 200, True)
        for step in range(5):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        names = (L.F_U, L.F_V, L.F_P) + ((L.F_K, L.F_OMEGA) if turbulent else ())
        out[k6] = ([dev.download(f) for f in names], dev.stats(), dev.kernel_counts())
        dev.close()
    (f1, st1, k1), (f0, st0, k0) = out[1], out[0]
    # (the two-right-hand-side launches may differ by a few: a right-hand side can leave one measured iterate earlier or later)
    # (the k-omega run's solves leave few iterations for a single right-hand side: never more launches, fewer in the laminar runs)
    assert k1["cheb4_2rhs"] > 0 and 0 < k1["cheb4_1rhs"] <= k0["cheb4_1rhs"], (k1, k0)
    assert k1["cheb4_2rhs"] + k1["cheb4_1rhs"] <= k0["cheb4_2rhs"] + k0["cheb4_1rhs"], (k1, k0)
    if not turbulent:
        assert k1["cheb4_1rhs"] < k0["cheb4_1rhs"], (k1, k0)
        assert k1["cheb6_1rhs"] > 0
    assert k0["cheb6_1rhs"] == 0
    assert abs(st1["mom_iters"] - st0["mom_iters"]) <= 2
    assert k1["bicgstab_iters"] == k0["bicgstab_iters"] == 0 and st1["p_iters"] == st0["p_iters"]
    if spec:
        assert k1["speculative_iterations"] > 0 and k1["speculative_redone"] <= 1
    for a, b in zip(f1, f0):
        assert rel_l2(mask_corners(a), mask_corners(b)) <= 1e-9


def test_first_coarse_level_in_single_precision():
    """Option `mg_c32` (default on, one rank): where the two finest multigrid levels both run the fused three-sweep kernels
    with single-precision faces, the first coarse level keeps its right-hand side and its iterate in single precision as
    well (38 instead of 74 B per coarse cell and V-cycle).  Storage of the preconditioner only: the V-cycle stays a
    symmetric positive operator to single-precision rounding, PCG takes the same number of iterations and converges to the
    solution of the double-precision system."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    rng = np.random.default_rng(5)
    r1, r2 = rng.standard_normal((512, 512)), rng.standard_normal((512, 512))
    r1 -= r1.mean()
    r2 -= r2.mean()
    for c32 in (2, 1, 0):
        dev, ctl, cfg = bench.setup_device(512, False, 2, 1e-8, 0, tuning={"mg_c32": c32, "mg_c32_min_cells": 1024})
        dev.set_solver(1e-8, 200, 1e-11, 200, 1e-8, 200, True)
        for step in range(4):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        st = dev.stats()
        z1, z2 = dev.vcycle(r1), dev.vcycle(r2)
        out[c32] = ([dev.download(f) for f in (L.F_U, L.F_V, L.F_P)], st, z1, z2, dev.kernel_counts())
        dev.close()
    f0, st0, zb1, zb2, k0 = out[0]
    for c32 in (1, 2):          # float vectors on the first / the first two coarse levels (512^2: levels 256^2 and 128^2)
        f1, st1, za1, za2, k1 = out[c32]
        assert k1["mg_fused_levels_max"] >= 3 and abs(st1["p_iters"] - st0["p_iters"]) <= 1 and st1["p_relres"] <= 1e-11
        assert k1["mg_float_levels"] == c32 and k0["mg_float_levels"] == 0
        sym = abs(np.vdot(r1, za2) - np.vdot(za1, r2)) / abs(np.vdot(r1, za2))
        assert np.vdot(r1, za1) > 0.0 and np.vdot(r2, za2) > 0.0 and sym < 1e-5, sym
        # the preconditioners differ at the single-precision level, and only there
        dz = np.abs(za1 - zb1).max() / np.abs(zb1).max()
        assert 0.0 < dz < 1e-5, (c32, dz)
        for a, b in zip(f1, f0):
            assert rel_l2(mask_corners(a), mask_corners(b)) <= 1e-8
    assert np.abs(out[2][2] - out[1][2]).max() > 0.0       # the second level's rounding is there as well


@pytest.mark.parametrize("n,turbulent,graded,spec", [(64, False, False, 1), (256, False, False, 0), (192, True, False, 1), (256, False, True, 1)])
def test_inner_products_summed_by_the_consuming_kernels(n, turbulent, graded, spec):
    """Option `fold_sums` (default on, one rank): (r, z) and (p, q) of the fused MG-PCG are added up inside the kernel that
    consumes them (direction step; residual update) in the order of the single-workgroup finish kernel they replace -- the
    same bits, two dependent launches per PCG iteration fewer.  Uniform, k-omega and stretched (semi-coarsened) meshes,
    host-checked and replayed schedules."""
    import bench
    from cfd_amd import _lib as L
    from cfd_amd.solver import advance_one_step
    out = {}
    for fold in (1, 0):
        dev, ctl, cfg = bench.setup_device(n, turbulent, 2, 1e-8, 0, tuning={"fold_sums": fold, "speculate": spec}, graded=graded)
        for step in range(2):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        launches0, _ = dev.host_counters()
        for step in range(2, 5):
            advance_one_step(dev, ctl, step)
        dev.synchronize()
        launches1, _ = dev.host_counters()
        names = (L.F_U, L.F_V, L.F_P, L.F_PCORR) + ((L.F_K, L.F_OMEGA) if turbulent else ())
        out[fold] = ([dev.download(f) for f in names], dev.stats(), dev.kernel_counts(), launches1 - launches0)
        dev.close()
    (f1, st1, k1, l1), (f0, st0, k0, l0) = out[1], out[0]
    assert k1["cg_pq"] == k0["cg_pq"] > 0 and st1["p_iters"] == st0["p_iters"]
    assert l1 < l0 and l0 - l1 >= k1["cg_pq"] // 2, (l1, l0, k1["cg_pq"])
    for a, b in zip(f1, f0):
        assert np.array_equal(mask_corners(a), mask_corners(b)), np.abs(a - b).max()


def _pcg_tail_pair(builder, fixup, tmp_path, extra=None):
    """The same case with the pressure solve as one launch (k_pcg_tail, default on small meshes) and as the launch-per-phase loop."""
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    runs = []
    for one in (1, 0):
        cfg, mp, kind = builder()
        work = tmp_path / f"t{one}"
        gen = hip_mesh.generate_rect_mesh if kind == "rect" else hip_mesh.generate_annulus_mesh
        cfg["mesh"] = {"generator": gen, "params": mp, "path": work / "mesh" / "mesh.npz"}
        cfg["paths"] = {"case_dir": work, "mesh_dir": work / "mesh", "states_dir": work / "states", "post_dir": work / "postproc"}
        mesh = hip_mesh.make_mesh(cfg["mesh"])
        if fixup is not None:
            fixup(cfg, mesh)
        cfg["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
        cfg["solver"]["momentum"]["tol"] = TRANSPORT_TOL
        cfg["solver"]["tuning"] = dict({"mg_line": 0, "pcg_tail": one}, **(extra or {}))
        fields, step, time = hip_solver.run_case(cfg)
        log = np.loadtxt(work / "postproc" / f"{cfg['case_name']}_residual_log.txt", skiprows=1, ndmin=2)
        runs.append((fields, log, dict(hip_solver.LAST_RUN)))
    return runs


PCG_TAIL_CASES = {
    "periodic_16x12": (lambda: cases.lam_channel(16, 12, steps=4), None),                 # nullspace form, reference cell
    "periodic_31x17": (lambda: cases.lam_channel(31, 17, steps=4), None),                 # odd sizes: aggregates of one column / row
    "periodic_32x32": (lambda: cases.channel_forced(n=32, steps=3), cases.perturbed_parabola),
    "komega_32x32": (lambda: cases.channel_forced(n=32, steps=3, turbulent=True), cases.perturbed_parabola),
    "inlet_outlet_20x10": (lambda: cases.inlet_outlet_channel(20, 10, steps=4), None),    # Dirichlet pressure: no nullspace
    "inlet_outlet_40x24": (lambda: cases.inlet_outlet_channel(40, 24, steps=3), None),
    "periodic_48x40": (lambda: cases.lam_channel(48, 40, steps=3), None),                 # more than 1024 cells: four cells per thread
    "inlet_outlet_50x52": (lambda: cases.inlet_outlet_channel(50, 52, steps=3), None),    # 2600 cells, close to what the LDS holds
    "couette_piso_8x10": (lambda: cases.couette_piso(8, 10, steps=4), cases.perturbed_rest),   # several solves per step
    "half_cylinder_12x16": (lambda: cases.half_cylinder(nr=12, nt=16, steps=3), None),    # curvilinear, symmetry rows
    "channel_komega_5x20": (lambda: cases.channel_komega_5x20(steps=6, tol=cases.TIGHT), cases.ramp_initial_u),  # the reference's own case
}


@pytest.mark.parametrize("speculate", [0, 1])
@pytest.mark.parametrize("name", list(PCG_TAIL_CASES))
def test_whole_pressure_solve_in_one_launch(name, speculate, tmp_path):
    """k_pcg_tail (the MG-PCG solve of a mesh whose hierarchy fits one workgroup's LDS: V-cycles, dot products, updates and
    the stopping test in ONE launch) against pcg_mg's launch-per-phase loop: the same iteration counts (or fewer), fields equal
    to the solver tolerance, with the host-checked and the speculative inner iterations."""
    builder, fixup = PCG_TAIL_CASES[name]
    (f1, log1, last1), (f0, log0, last0) = _pcg_tail_pair(builder, fixup, tmp_path, {"speculate": speculate})
    k1, k0 = last1["kernel_counts"], last0["kernel_counts"]
    assert k1["pcg_tail"] > 0 and k1["cg_pq"] == 0 and k1["vcycles"] == 0, k1
    assert k0["pcg_tail"] == 0 and k0["vcycles"] > 0, k0
    assert last1["stats"]["p_unconverged"] == 0 and last0["stats"]["p_unconverged"] == 0
    # (near the rounding level -- these cases ask for 1e-13 -- the loop's single-precision multigrid faces on the marching
    # levels cost it up to four iterations more than the cycle with double faces in LDS: 27 against 31 at 40 x 24)
    assert last0["stats"]["p_iters"] - 5 <= last1["stats"]["p_iters"] <= last0["stats"]["p_iters"] + 1, (last1["stats"], last0["stats"])
    unorm = max(np.linalg.norm(f0["u"].ravel()), 1e-300)
    for key in f0:
        err = np.linalg.norm((mask_corners(f1[key]) - mask_corners(f0[key])).ravel())
        den = np.linalg.norm(mask_corners(f0[key]).ravel())
        assert err <= FIELD_TOL * den + 1e-12 * unorm, (key, err, den)
    assert np.array_equal(log1[:, [0, 2]], log0[:, [0, 2]])


def test_whole_pressure_solve_reports_a_solve_that_runs_out_of_iterations():
    """maxiter too small for the tolerance: the one-launch solve stops at maxiter and the context counts the unconverged
    solve exactly as the launch-per-phase loop does (frozen operator: counted, not raised), speculative or not."""
    import bench
    from cfd_amd.solver import advance_one_step
    for spec in (0, 1):
        counts = {}
        for one in (1, 0):
            dev, ctl, cfg = bench.setup_device(32, False, 2, 1e-8, 0, tuning={"pcg_tail": one, "speculate": spec})
            for step in range(3):
                advance_one_step(dev, ctl, step)
            assert dev.stats()["p_unconverged"] == 0
            dev.set_solver(1e-8, 200, 1e-13, 2, 1e-8, 200, True)      # two iterations cannot reach 1e-13
            advance_one_step(dev, ctl, 3)
            st = dev.stats()
            counts[one] = (st["p_unconverged"], st["p_iters"])
            assert dev.kernel_counts()["pcg_tail"] > 0 if one else dev.kernel_counts()["pcg_tail"] == 0
            dev.close()
        assert counts[1] == counts[0] and counts[1][0] > 0 and counts[1][1] == 2, counts


CHEB_TAIL_CASES = {
    "periodic_16x12": (lambda: cases.lam_channel(16, 12, steps=4), None),
    "periodic_31x17": (lambda: cases.lam_channel(31, 17, steps=4), None),
    "periodic_32x32": (lambda: cases.channel_forced(n=32, steps=3), cases.perturbed_parabola),
    "komega_32x32": (lambda: cases.channel_forced(n=32, steps=3, turbulent=True), cases.perturbed_parabola),   # k / omega solves too
    "inlet_outlet_40x24": (lambda: cases.inlet_outlet_channel(40, 24, steps=3), None),
    "inlet_outlet_50x52": (lambda: cases.inlet_outlet_channel(50, 52, steps=3), None),    # four cells per thread
    "couette_piso_8x10": (lambda: cases.couette_piso(8, 10, steps=4), cases.perturbed_rest),
    "half_cylinder_12x16": (lambda: cases.half_cylinder(nr=12, nt=16, steps=3), None),
    "channel_komega_5x20": (lambda: cases.channel_komega_5x20(steps=6, tol=cases.TIGHT), cases.ramp_initial_u),
    "channel_sst_5x20": (lambda: cases.channel_komega_5x20(steps=6, tol=cases.TIGHT, model="sst"), cases.ramp_initial_u),
}


@pytest.mark.parametrize("speculate", [0, 1])
@pytest.mark.parametrize("name", list(CHEB_TAIL_CASES))
def test_whole_transport_solve_in_one_launch(name, speculate, tmp_path):
    """k_cheb_tail (row-sum bound, first pass, the iteration count the Chebyshev bound predicts, the passes and the
    stopping test in ONE launch, iterates in LDS) against the one-iteration-per-launch form it restates: the iterates are
    the same bits, only the residual sums are taken in another order, so the fields agree to rounding; fall-backs to
    BiCGStab (systems that are not time-step dominated) are taken alike."""
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    builder, fixup = CHEB_TAIL_CASES[name]
    runs = []
    for one in (1, 0):
        cfg, mp, kind = builder()
        work = tmp_path / f"c{one}"
        gen = hip_mesh.generate_rect_mesh if kind == "rect" else hip_mesh.generate_annulus_mesh
        cfg["mesh"] = {"generator": gen, "params": mp, "path": work / "mesh" / "mesh.npz"}
        cfg["paths"] = {"case_dir": work, "mesh_dir": work / "mesh", "states_dir": work / "states", "post_dir": work / "postproc"}
        mesh = hip_mesh.make_mesh(cfg["mesh"])
        if fixup is not None:
            fixup(cfg, mesh)
        cfg["solver"]["turbulence_solver"] = {"tol": TRANSPORT_TOL, "maxiter": 2000}
        cfg["solver"]["momentum"]["tol"] = TRANSPORT_TOL
        # the launch-per-iteration form of the comparison: the fused form (from 16 cells per side) retires a right-hand side
        # at the first iterate that meets the test, which is another -- equally valid -- iterate
        cfg["solver"]["tuning"] = {"mg_line": 0, "cheb_tail": one, "speculate": speculate, "fused_min_cheb": 4096}
        fields, step, time = hip_solver.run_case(cfg)
        runs.append((fields, dict(hip_solver.LAST_RUN)))
    (f1, last1), (f0, last0) = runs
    k1, k0 = last1["kernel_counts"], last0["kernel_counts"]
    assert k0["cheb_tail"] == 0
    assert last1["stats"]["mom_solver"] == last0["stats"]["mom_solver"], (last1["stats"], last0["stats"])
    if last0["stats"]["mom_solver"] == "chebyshev":
        assert k1["cheb_tail"] > 0 and k1["cheb1"] == 0 and k0["cheb1"] > 0, (k1, k0)
        assert last1["stats"]["mom_iters"] == last0["stats"]["mom_iters"]
    unorm = max(np.linalg.norm(f0["u"].ravel()), 1e-300)
    for key in f0:
        err = np.linalg.norm((mask_corners(f1[key]) - mask_corners(f0[key])).ravel())
        den = np.linalg.norm(mask_corners(f0[key]).ravel())
        assert err <= 1e-12 * den + 1e-14 * unorm, (key, err, den)
