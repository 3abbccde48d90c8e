"""GPU tests of the per-phase host wrappers (cfd_amd.solver / cfd_amd.turbulence / cfd_amd.numba_kernels /
cfd_amd.post_processing): the reference's function names and call shapes (README.md:196-205), each checked
against vectors the reference itself produced (tests/golden/func_<tag>.npz, phase_<tag>.npz; generator:
oracle/gen_golden.py).  Stencil phases: bit-exact.  Phases that end in a linear solve: relative L2 <= 1e-10
(both sides solve to 1e-13).  Reductions: 1e-12 relative (summation order differs from numpy's pairwise sum).
"""
import numpy as np
import pytest

from conftest import golden_mesh, load_golden, mask_corners, rel_l2
from oracle import cases

pytestmark = pytest.mark.gpu

FIELD_TOL = 1.0e-10
TIGHT = {"tol": 1.0e-13, "maxiter": 2000}
TAGS = list(cases.FUNCTION_VECTOR_BCS)


@pytest.fixture(scope="module", autouse=True)
def need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    yield
    from cfd_amd import phases
    phases.release()


def _load(tag):
    g = load_golden(f"func_{tag}.npz")
    ph = load_golden(f"phase_{tag}.npz")
    mesh = golden_mesh(g)
    nu, dt = (float(x) for x in g["params"])
    return g, ph, mesh, cases.FUNCTION_VECTOR_BCS[tag], nu, dt


def _state(g):
    """The boundary-conditioned state the reference evaluated its phases on (gen_golden.function_vectors)."""
    s = {k: g[f"bc_{k}"].copy() for k in ("u", "v", "p")}
    s["k"], s["omega"], s["nu_t"] = g["tbc_k"].copy(), g["tbc_omega"].copy(), g["nu_t"].copy()
    return s


@pytest.mark.parametrize("tag", TAGS)
def test_apply_boundary_conditions(tag):
    from cfd_amd import solver
    g, _, mesh, bc_cfg, _, _ = _load(tag)
    fields = {k: g[f"in_{k}"].copy() for k in ("u", "v", "p", "k", "omega", "nu_t")}
    keep = {k: v for k, v in fields.items()}
    solver.apply_boundary_conditions(fields, mesh, bc_cfg)
    for name in fields:
        assert fields[name] is keep[name]                                   # in place, like the reference
        assert np.array_equal(fields[name], g[f"bc_{name}"]), name
    # u, v, p alone (a laminar caller's dict)
    lam = {k: g[f"in_{k}"].copy() for k in ("u", "v", "p")}
    solver.apply_boundary_conditions(lam, mesh, bc_cfg)
    for name in lam:
        assert np.array_equal(lam[name], g[f"bc_{name}"]), name


@pytest.mark.parametrize("tag", TAGS)
def test_turbulence_apply_bcs_and_eddy_viscosity(tag):
    from cfd_amd import turbulence
    g, ph, mesh, bc_cfg, nu, _ = _load(tag)
    cfg = {"enabled": True, "model": "k_omega"}
    fields = {k: g[f"bc_{k}"].copy() for k in ("u", "v", "p", "k", "omega", "nu_t")}
    assert turbulence.initialize(fields, mesh, cfg, bc_cfg) is fields
    turbulence.apply_bcs(fields, mesh, bc_cfg, cfg, nu)
    for name in ("k", "omega"):
        assert np.array_equal(fields[name], g[f"tbc_{name}"]), name
    nut = turbulence.eddy_viscosity(fields, cfg)
    assert nut is fields["nu_t"] and np.array_equal(nut, g["nu_t"])
    # disabled closure: nu_t zeroed / created (turbulence.py:217-224, 364-366)
    off = {"enabled": False}
    fields["nu_t"][:, :] = 3.0
    assert np.count_nonzero(turbulence.eddy_viscosity(fields, off)) == 0
    bare = {k: fields[k] for k in ("u", "v", "p")}
    turbulence.initialize(bare, mesh, off)
    assert bare["nu_t"].shape == fields["u"].shape and not bare["nu_t"].any()
    # validation of the reference (turbulence.py:231-251)
    with pytest.raises(ValueError, match="Initial k is missing"):
        turbulence.initialize({k: fields[k] for k in ("u", "v", "p")}, mesh, cfg)
    with pytest.raises(NotImplementedError):
        turbulence.initialize(fields, mesh, {"enabled": True, "model": "spalart"})


@pytest.mark.parametrize("tag", TAGS)
def test_turbulence_pointwise_closures_without_a_mesh(tag):
    """turbulence.sources / effective_diffusivity / eddy_viscosity have no mesh argument (turbulence.py:362, 401, 541):
    they run on a GPU context of the field shape, created on first use -- no earlier call that saw the mesh is
    needed.  Sources bit-exact against the reference's vectors (src_k, src_omega of gen_golden.function_vectors)."""
    from cfd_amd import phases, turbulence
    g, ph, mesh, bc_cfg, nu, _ = _load(tag)
    phases.release()                                            # no context of any mesh is alive
    cfg = {"enabled": True, "model": "k_omega"}
    s = _state(g)
    turbulence.initialize(s, mesh, cfg)                         # host bookkeeping only: records the closure
    sk, sw = turbulence.sources(s, g["grad_u"], g["grad_v"])
    assert sk.shape == sw.shape == g["src_k"].shape
    assert np.array_equal(sk, g["src_k"]) and np.array_equal(sw, g["src_omega"])
    for field in ("k", "omega"):
        d = turbulence.effective_diffusivity(nu, s["nu_t"], field)
        assert d.shape == s["nu_t"].shape and np.array_equal(d, nu + 0.5 * s["nu_t"])
    with pytest.raises(ValueError, match="Unsupported field"):
        turbulence.effective_diffusivity(nu, s["nu_t"], "epsilon")
    phases.release()
    fresh = {k: s[k].copy() for k in ("k", "omega")}
    fresh["nu_t"] = np.zeros_like(s["k"])
    nut = turbulence.eddy_viscosity(fresh, cfg)                 # first GPU call of this shape
    assert nut is fresh["nu_t"] and np.array_equal(nut, g["nu_t"])
    # SST: the blended planes a transport step cached in `fields` are handed back; the stand-alone sources are refused
    sst = {"enabled": True, "model": "sst"}
    turbulence.initialize(s, mesh, sst)
    cached = dict(s, _D_k=np.full_like(s["k"], 2.0), _D_omega=np.full_like(s["k"], 3.0))
    assert turbulence.effective_diffusivity(nu, s["nu_t"], "k", cached) is cached["_D_k"]
    assert turbulence.effective_diffusivity(nu, s["nu_t"], "omega", cached) is cached["_D_omega"]
    assert np.array_equal(turbulence.effective_diffusivity(nu, s["nu_t"], "omega", s), nu + 0.5 * s["nu_t"])
    with pytest.raises(NotImplementedError):
        turbulence.sources(s, g["grad_u"], g["grad_v"], mesh)
    turbulence.initialize(s, mesh, cfg)                         # back to k-omega for the tests that follow


@pytest.mark.parametrize("tag", TAGS)
def test_compute_adaptive_dt(tag):
    from cfd_amd import solver
    g, ph, mesh, _, nu, _ = _load(tag)
    s = _state(g)
    nu_eff = nu + s["nu_t"][1:-1, 1:-1]
    assert solver.compute_adaptive_dt(s, mesh, 0.5, 10.0, 0.0, cfl_diff=0.4, nu_eff=nu_eff) == float(g["dt"])
    assert solver.compute_adaptive_dt(s, mesh, 0.5, 10.0, 0.0, cfl_diff=0.4, nu_eff=nu) == float(ph["dt_scalar"])
    assert solver.compute_adaptive_dt(s, mesh, 0.5, 10.0) == float(ph["dt_adv_only"])
    assert solver.compute_adaptive_dt(s, mesh, 0.5, 1.0e-7, 0.0, cfl_diff=0.4, nu_eff=nu_eff) == float(ph["dt_clamped"])


@pytest.mark.parametrize("tag", TAGS)
def test_compute_advective_rhs(tag):
    from cfd_amd import solver
    g, ph, mesh, _, nu, dt = _load(tag)
    s = _state(g)
    nu_eff = nu + s["nu_t"][1:-1, 1:-1]
    rhs = solver.compute_advective_rhs(s, mesh, nu_eff, dt, use_rhie_chow=True, p_field=s["p"])
    assert np.array_equal(rhs["flux_i"], g["flux_i"]) and np.array_equal(rhs["flux_j"], g["flux_j"])
    assert np.array_equal(rhs["u"], ph["adv_rhs_u"]) and np.array_equal(rhs["v"], ph["adv_rhs_v"])
    assert np.array_equal(rhs["u"][1:-1, 1:-1], g["adv_u"])
    # scalar viscosity, p taken from the dict (solver.py:205-207, 459)
    rhs0 = solver.compute_advective_rhs(s, mesh, nu, dt)
    for key in ("u", "v", "flux_i", "flux_j"):
        assert np.array_equal(rhs0[key], ph[f"adv0_{key}"]), key
    # the stand-alone stencils of the same module
    for name in ("u", "v", "k", "omega"):
        assert np.array_equal(solver.advect_scalar(s[name], mesh, g["flux_i"], g["flux_j"]), g[f"adv_{name}"]), name
    for name in ("u", "v", "p", "k"):
        assert np.array_equal(solver.grad_scalar(s[name], mesh), g[f"grad_{name}"]), name


@pytest.mark.parametrize("tag", TAGS)
def test_solve_momentum(tag):
    from cfd_amd import solver
    g, ph, mesh, bc_cfg, nu, dt = _load(tag)
    s = _state(g)
    nu_eff = nu + s["nu_t"][1:-1, 1:-1]
    adv = {"u": ph["adv_rhs_u"], "v": ph["adv_rhs_v"]}
    ghosts = {k: mask_corners(s[k]) for k in ("u", "v")}
    info = solver.solve_momentum(s, mesh, dt, nu_eff, adv, bc_cfg, TIGHT, p_field=s["p"], pressure_gradient=(-0.3, 0.1),
                                 extra_pressure_field=(2.0 / 3.0) * s["k"])
    assert np.array_equal(info["a_p_u"], ph["mom_a_p"]) and np.array_equal(info["a_p_v"], ph["mom_a_p"])
    for name in ("u", "v"):
        assert rel_l2(s[name][1:-1, 1:-1], ph[f"mom_{name}"][1:-1, 1:-1]) < FIELD_TOL, name
        edge = mask_corners(s[name])
        edge[1:-1, 1:-1] = ghosts[name][1:-1, 1:-1] = 0.0
        assert np.array_equal(edge, ghosts[name]), f"{name}: ghosts must stay untouched"
    assert info["mom_u_l2"] < 1e-10 and info["mom_v_l2"] < 1e-10       # RMS of A x - b after a 1e-13 solve
    # scalar viscosity, no forcing, no extra field, p from the dict
    s0 = _state(g)
    solver.solve_momentum(s0, mesh, dt, nu, {"u": ph["adv0_u"], "v": ph["adv0_v"]}, bc_cfg, TIGHT)
    for name in ("u", "v"):
        assert rel_l2(s0[name][1:-1, 1:-1], ph[f"mom0_{name}"][1:-1, 1:-1]) < FIELD_TOL, name


@pytest.mark.parametrize("tag", TAGS)
def test_solve_pressure(tag):
    from cfd_amd import solver
    g, ph, mesh, bc_cfg, nu, dt = _load(tag)
    s = _state(g)
    nu_eff = nu + s["nu_t"][1:-1, 1:-1]
    pc = solver.solve_pressure(s, mesh, dt, nu_eff, bc_cfg, TIGHT, a_p_override=g["a_p"], p_field=s["p"])
    assert pc is s["p_corr"]
    assert rel_l2(pc[1:-1, 1:-1], g["p_corr"][1:-1, 1:-1]) < FIELD_TOL
    # assembled diagonal from a scalar viscosity; p_corr already in the dict is reused in place
    before = s["p_corr"]
    pc0 = solver.solve_pressure(s, mesh, dt, nu, bc_cfg, TIGHT)
    assert pc0 is before
    assert rel_l2(pc0[1:-1, 1:-1], ph["p_corr_scalar"][1:-1, 1:-1]) < FIELD_TOL
    # amg_cache keeps the first operator like the reference's cached pyamg hierarchy (solver.py:827-833): a second
    # call with another diagonal still solves with the first one
    cache = {}
    first = solver.solve_pressure(_state(g), mesh, dt, nu_eff, bc_cfg, TIGHT, a_p_override=g["a_p"], amg_cache=cache).copy()
    assert cache["shape"] == (g["a_p"].size, g["a_p"].size)
    s2 = _state(g)
    again = solver.solve_pressure(s2, mesh, dt, nu_eff, bc_cfg, TIGHT, a_p_override=g["a_p"], amg_cache=cache)
    assert rel_l2(again[1:-1, 1:-1], first[1:-1, 1:-1]) < FIELD_TOL


@pytest.mark.parametrize("tag", TAGS)
def test_correct_velocity(tag):
    from cfd_amd import solver
    g, _, mesh, bc_cfg, _, _ = _load(tag)
    s = _state(g)
    s["p_corr"] = g["p_corr"].copy()
    solver.correct_velocity(s, mesh, bc_cfg, g["a_p"], g["a_p"], g["in_u0"], g["in_v0"], g["in_p0"], 0.7, 0.6, 0.3)
    for name in ("u", "v", "p"):
        assert np.array_equal(s[name][1:-1, 1:-1], g[f"corr_{name}"][1:-1, 1:-1]), name
    assert np.array_equal(mask_corners(s["p_corr"]), mask_corners(g["corr_p_corr"]))
    with pytest.raises(RuntimeError, match="Pressure correction field is missing"):
        solver.correct_velocity(_state(g), mesh, bc_cfg, g["a_p"], g["a_p"], g["in_u0"], g["in_v0"], g["in_p0"], 0.7, 0.6, 0.3)


@pytest.mark.parametrize("tag", TAGS)
def test_continuity_residual(tag):
    from cfd_amd import solver
    g, ph, mesh, _, nu, dt = _load(tag)
    s = _state(g)
    nu_eff = nu + s["nu_t"][1:-1, 1:-1]
    for got, want in ((solver.continuity_residual(s, mesh, nu_eff, dt, p_field=s["p"]), float(g["continuity"])),
                      (solver.continuity_residual(s, mesh, nu, dt), float(ph["continuity_scalar"])),
                      (solver.continuity_residual(s, mesh, nu, dt, use_rhie_chow=False), float(ph["continuity_norc"]))):
        assert abs(got - want) <= 1e-12 * want


@pytest.mark.parametrize("model", ["k_omega", "sst"])
@pytest.mark.parametrize("tag", TAGS)
def test_update_turbulence(tag, model):
    from cfd_amd import solver
    g, ph, mesh, bc_cfg, nu, dt = _load(tag)
    s = _state(g)
    cfg = {"enabled": True, "model": model, "solver": TIGHT}
    solver.update_turbulence(s, mesh, cfg, bc_cfg, dt, nu)
    from cfd_amd import turbulence
    turbulence.eddy_viscosity(s, cfg)
    for name in ("k", "omega", "nu_t"):
        assert rel_l2(mask_corners(s[name]), mask_corners(ph[f"turb_{model}_{name}"])) < FIELD_TOL, name
    if model == "sst":
        for name in ("_S", "_F1", "_F2", "_D_k", "_D_omega", "_d_wall"):
            # derived from the solved k, omega through tanh(arg^4) / tanh(arg^2) blends: the 1e-10 of k, omega is
            # amplified by the blend's slope on this random state (1.3e-10 seen for _F2 on annulus_132x24)
            assert rel_l2(s[name][1:-1, 1:-1], ph[f"turb_sst_{name}"][1:-1, 1:-1]) < 10.0 * FIELD_TOL, name
    # update_interval gate (solver.py:1096-1101): the first of every two calls returns without touching anything
    s2 = _state(g)
    gate = {"enabled": True, "model": model, "update_interval": 2, "solver": TIGHT}
    solver.update_turbulence(s2, mesh, gate, bc_cfg, dt, nu)
    assert gate["_step_counter"] == 1 and np.array_equal(s2["k"], g["tbc_k"])
    # disabled: nu_t zeroed, nothing else (solver.py:1092-1094)
    s2["nu_t"][:, :] = 1.0
    solver.update_turbulence(s2, mesh, {"enabled": False}, bc_cfg, dt, nu)
    assert not s2["nu_t"].any() and np.array_equal(s2["k"], g["tbc_k"])


@pytest.mark.parametrize("tag", TAGS)
def test_numba_plug_point(tag):
    """numba_kernels.advect_scalar_numba / grad_scalar_numba (numba_kernels.py:5-83) with the reference's argument
    lists, against the reference's advect_scalar / grad_scalar results (the numba kernels restate those)."""
    from cfd_amd import numba_kernels as nk
    g, _, mesh, _, _, _ = _load(tag)
    s = _state(g)
    for name in ("u", "k"):
        got = nk.advect_scalar_numba(s[name], g["flux_i"], g["flux_j"], mesh["cell_volume"])
        assert np.array_equal(got, g[f"adv_{name}"]), name
    for name in ("u", "p"):
        got = nk.grad_scalar_numba(s[name], mesh["w_left_i"], mesh["w_right_i"], mesh["w_south_j"], mesh["w_north_j"],
                                   mesh["face_normal_i"], mesh["face_normal_j"], mesh["cell_volume"])
        assert np.array_equal(got, g[f"grad_{name}"]), name


@pytest.mark.parametrize("side", ["j_min", "i_min"])
@pytest.mark.parametrize("tag", TAGS)
def test_wall_friction_edge_kernel(tag, side):
    """cfd_wall_friction (edge kernel + device sums) against post_processing.friction_coefficient_array of the
    reference (post_processing.py:588-687) on the same fields."""
    from cfd_amd import _lib as L
    from cfd_amd import post_processing as pp
    from cfd_amd import phases
    g, _, mesh, bc_cfg, nu, _ = _load(tag)
    s = _state(g)
    eng = phases.engine(mesh)
    eng.load(s, ("u", "v", "p"))
    eng.dev.upload(L.F_NUT, s["nu_t"])
    eng.nut_zero = False
    wall = pp.WallFriction(mesh, side, None)
    got = pp.friction_coefficient_device(eng.dev, wall, nu=nu, hydraulic_diameter=0.7, pressure_gradient=(-0.3, 0.0),
                                         want_tau=True)
    want = g[f"friction_{side}"]          # cf, tau_avg, bulk_velocity, re, cf_dpdx
    for key, ref in zip(("cf", "tau_avg", "bulk_velocity", "re", "cf_dpdx"), want):
        assert abs(got[key] - ref) <= 1e-12 * abs(ref), key
    host = pp.friction_coefficient_array(s, mesh, side, None, nu=nu, hydraulic_diameter=0.7, pressure_gradient=(-0.3, 0.0))
    assert np.array_equal(got["tau"], host["tau"]) and np.array_equal(got["s"], host["s"])
    # a sub-range and nu_t in the viscosity
    n = wall.hi + 1
    lo, hi = n // 4, n - 2
    sub = pp.WallFriction(mesh, side, (lo, hi))
    got = pp.friction_coefficient_device(eng.dev, sub, nu=nu, use_nu_t=True, want_tau=True)
    host = pp.friction_coefficient_array(s, mesh, side, (lo, hi), nu=nu, use_nu_t=True)
    assert np.array_equal(got["tau"], host["tau"])
    assert abs(got["tau_avg"] - host["tau_avg"]) <= 1e-12 * abs(host["tau_avg"])
    with pytest.raises(Exception, match="outside side"):
        eng.dev.wall_friction(wall.side, 0, n, np.zeros((n + 1, 2)), np.ones(n + 1), np.ones(n + 1), nu)


def test_run_case_friction_log_without_field_download(tmp_path, monkeypatch):
    """run_case's friction log comes from the edge kernel: the numbers of the reference's own committed log
    (Runs/Coarse_mesh, real pyamg, rtol 1e-8), and no field leaves the GPU before the end of the run."""
    from conftest import GOLDEN
    from cfd_amd import device
    from cfd_amd import mesh as hip_mesh
    from cfd_amd import solver as hip_solver
    calls = []
    orig = device.Device.download_fields
    monkeypatch.setattr(device.Device, "download_fields", lambda self, f: (calls.append(1), orig(self, f))[1])
    cfg, mp, _ = cases.channel_komega_5x20(steps=21, tol=1e-8)
    cfg["mesh"] = {"generator": hip_mesh.generate_rect_mesh, "params": mp, "path": tmp_path / "mesh" / "Coarse_mesh.npz"}
    cfg["paths"] = {"case_dir": tmp_path, "mesh_dir": tmp_path / "mesh", "states_dir": tmp_path / "states",
                    "post_dir": tmp_path / "postproc"}
    mesh = hip_mesh.make_mesh(cfg["mesh"])
    cases.ramp_initial_u(cfg, mesh)
    cfg["post"]["save_interval"] = 0
    hip_solver.run_case(cfg)
    assert len(calls) == 1                       # the one that fills the returned fields
    fric = np.loadtxt(tmp_path / "postproc" / "channel_k_omega_friction_log.txt", skiprows=1, ndmin=2)
    rfric = np.loadtxt(GOLDEN / "ref_runs" / "channel_k_omega_friction_log.txt", skiprows=1)[:fric.shape[0]]
    assert fric.shape[0] == 3
    for col in (2, 3, 4, 5, 6):
        assert np.all(np.abs(fric[:, col] - rfric[:, col]) <= 2e-6 * np.abs(rfric[:, col])), col
