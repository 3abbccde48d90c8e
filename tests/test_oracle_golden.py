"""Pin the CPU oracle (oracle/cfd_oracle.py) against the reference.

Golden vectors come from oracle/gen_golden.py (the reference itself, run in the build
container) and from the reference's committed run artefacts (tests/golden/ref_runs).
Runs on CPU only; no GPU, no /root/reference access.
"""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from conftest import GOLDEN, golden_mesh, load_golden, mask_corners, rel_l2
from oracle import cases
from oracle import cfd_oracle as orc

FUNC_CASES = {
    "annulus_12x16": ("annulus", cases._half_cylinder_bc(12, 16, True)),
    "rect_8x10": ("rect", {
        "i_min": {"type": "periodic", "range": (0, 9)},
        "i_max": {"type": "periodic", "range": (0, 9)},
        "j_min": {"type": "wall", "range": (0, 7)},
        "j_max": {"type": "symmetry", "range": (0, 7)},
        "pressure_reference": {"cell": (1, 2), "value": 0.25},
    }),
    # wider than two column strips of the marching kernels (oracle/gen_golden.py large; matrices stored as CSR)
    "annulus_132x24": ("annulus", cases._half_cylinder_bc(132, 24, True)),
    "rect_136x20": ("rect", {
        "i_min": {"type": "periodic", "range": (0, 19)},
        "i_max": {"type": "periodic", "range": (0, 19)},
        "j_min": {"type": "wall", "range": (0, 135)},
        "j_max": {"type": "symmetry", "range": (0, 135)},
        "pressure_reference": {"cell": (70, 3), "value": 0.25},
    }),
}
NODE_ARGS = {"annulus_12x16": (0.5, 6.0, 12, 16, 0.0, float(np.pi), 1.15, 1.0), "rect_8x10": (1.0, 0.5, 8, 10, 1.1, 1.2),
             "annulus_132x24": (0.5, 6.0, 132, 24, 0.0, float(np.pi), 1.02, 1.0), "rect_136x20": (3.0, 0.5, 136, 20, 1.01, 1.1)}


def _same_matrix(a, g, key):
    """The reference's matrix: dense for the small vector sets, CSR triplets for the large ones."""
    if key in g:
        return np.array_equal(a.toarray(), g[key])
    import scipy.sparse as sp
    ref = sp.csr_matrix((g[key + "_data"], g[key + "_indices"], g[key + "_indptr"]), shape=a.shape)
    d = (a.tocsr() - ref).tocoo()
    return d.nnz == 0 or not np.any(d.data != 0.0)


def test_mesh_matches_committed_reference_mesh_bitwise():
    """Tier A: Runs/Coarse_mesh/mesh/Coarse_mesh.npz (generate_rect_mesh(0.02,0.05,5,20,1.0,1.03))."""
    ref = load_golden("ref_runs/Coarse_mesh.npz")
    mine = orc.rect_mesh(0.02, 0.05, 5, 20, 1.0, 1.03)
    assert set(mine) == set(ref)
    for k in ref:
        assert np.array_equal(np.asarray(mine[k]), ref[k]), k


@pytest.mark.parametrize("tag", list(FUNC_CASES))
def test_mesh_matches_reference_generators_bitwise(tag):
    g = load_golden(f"func_{tag}.npz")
    ref = golden_mesh(g)
    node = ref["node"]
    mine = orc.metrics_from_nodes(node)
    for k in ref:
        assert np.array_equal(np.asarray(mine[k]), ref[k]), k
    n2 = orc.annulus_nodes(*NODE_ARGS[tag]) if tag.startswith("annulus") else orc.rect_nodes(*NODE_ARGS[tag])
    assert np.array_equal(n2, node)


@pytest.mark.parametrize("tag", list(FUNC_CASES))
def test_per_function_vectors(tag):
    g = load_golden(f"func_{tag}.npz")
    _, bc_cfg = FUNC_CASES[tag]
    mesh = golden_mesh(g)
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    nu, dt = g["params"]
    f = {k: g[f"in_{k}"].copy() for k in ("u", "v", "p", "k", "omega", "nu_t")}

    orc.apply_flow_bcs(f, mesh, bc_cfg)
    for k in f:
        assert np.array_equal(f[k], g[f"bc_{k}"]), f"flow bc {k}"
    orc.apply_turbulence_bcs(f, mesh, bc_cfg, nu)
    for k in ("k", "omega"):
        assert np.array_equal(f[k], g[f"tbc_{k}"]), f"turb bc {k}"
    nut = orc.eddy_viscosity(f, True)
    assert np.array_equal(nut, g["nu_t"])
    nu_eff = nu + nut[1:-1, 1:-1]

    a_i, a_j = orc.face_diffusion(mesh, nu_eff)
    assert np.array_equal(a_i, g["a_i"]) and np.array_equal(a_j, g["a_j"])
    a_i, a_j = orc.face_diffusion(mesh, float(nu))
    assert np.array_equal(a_i, g["a_i_const"]) and np.array_equal(a_j, g["a_j_const"])
    ap = orc.momentum_diag(mesh, nu_eff, dt)
    assert np.array_equal(ap, g["a_p"])
    fi, fj = orc.face_flux(f["u"], f["v"], f["p"], mesh, ap)
    assert np.array_equal(fi, g["flux_i"]) and np.array_equal(fj, g["flux_j"])
    vi, vj = orc.face_flux(f["u"], f["v"], None, mesh)
    assert np.array_equal(vi, g["vflux_i"]) and np.array_equal(vj, g["vflux_j"])
    d, da, dd = orc.adaptive_dt(f, mesh, 0.5, 10.0, 0.0, 0.4, nu_eff)
    assert d == float(g["dt"])
    assert (da, dd) == tuple(g["dt_limits"])
    for name in ("u", "v", "k", "omega"):
        assert np.array_equal(orc.upwind_advection(f[name], mesh, fi, fj), g[f"adv_{name}"]), name
    for name in ("u", "v", "p", "k"):
        assert np.array_equal(orc.green_gauss(f[name], mesh), g[f"grad_{name}"]), name
    sk, sw = orc.komega_sources(f, g["grad_u"], g["grad_v"])
    assert np.array_equal(sk, g["src_k"]) and np.array_equal(sw, g["src_omega"])

    per_i, per_j = orc.periodic_axes(bc_cfg, ni, nj)
    a, b, ap2 = orc.transport_system(f["u"], mesh, nu_eff, dt, g["in_source"], per_i, per_j)
    assert _same_matrix(a, g, "sys_A")
    assert np.array_equal(b, g["sys_b"]) and np.array_equal(ap2, g["sys_ap"])

    a, b = orc.pressure_system(f["u"], f["v"], f["p"], mesh, ap, bc_cfg)
    assert _same_matrix(a, g, "psys_A")
    assert np.array_equal(b, g["psys_b"])

    h = {k: v.copy() for k, v in f.items()}
    h["p_corr"] = np.zeros_like(h["p"])
    h["p_corr"][1:-1, 1:-1] = spla.spsolve(a.tocsc(), b).reshape(ni, nj)
    assert rel_l2(h["p_corr"][1:-1, 1:-1], g["p_corr"][1:-1, 1:-1]) < 1e-12
    h["p_corr"][1:-1, 1:-1] = g["p_corr"][1:-1, 1:-1]
    orc.apply_correction(h, mesh, bc_cfg, ap, g["in_u0"], g["in_v0"], g["in_p0"], 0.7, 0.6, 0.3)
    for k in ("u", "v", "p"):
        assert np.array_equal(h[k], g[f"corr_{k}"]), k
    assert np.array_equal(mask_corners(h["p_corr"]), mask_corners(g["corr_p_corr"]))
    assert orc.continuity_l2(f, mesh, nu_eff, dt) == float(g["continuity"])
    for side in ("j_min", "i_min"):
        fr = orc.wall_friction(f, mesh, side, None, nu, hydraulic_diameter=0.7,
                               pressure_gradient=(-0.3, 0.0))
        got = np.array([fr["cf"], fr["tau_avg"], fr["bulk_velocity"], fr["re"], fr["cf_dpdx"]])
        np.testing.assert_allclose(got, g[f"friction_{side}"], rtol=1e-14)


def _run_oracle(name, freeze):
    builder, fixup = {**cases.ALL_END_TO_END, **cases.SST_END_TO_END, **cases.LARGE_END_TO_END}[name]
    cfg, mp, kind = builder()
    mesh = orc.rect_mesh(**mp) if kind == "rect" else orc.annulus_mesh(**mp)
    if fixup is not None:
        fixup(cfg, mesh)
    return orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=freeze)


E2E = [(n, "fresh") for n in ("lam_channel", "inlet_outlet_channel", "couette_piso", "half_cylinder",
                              "channel_forced_24")]
E2E += [(n, "frozen") for n in cases.ALL_END_TO_END]
E2E += [(n, "frozen") for n in cases.SST_END_TO_END]      # SST rows: fields include _S, _F1, _F2, _D_k, _D_omega, _d_wall
# 64 x 64 and 160 x 96: the sizes at which the GPU run goes through its production kernels (tests/test_gpu_large.py)
E2E += [(n, "frozen") for n in cases.LARGE_END_TO_END]


@pytest.mark.parametrize("name,mode", E2E)
def test_end_to_end_against_reference(name, mode):
    """Fields after N outer steps and the residual history vs the reference itself."""
    g = load_golden(f"e2e_{name}_{mode}.npz")
    fields, step, time, hist = _run_oracle(name, freeze=(mode == "frozen"))
    assert step == int(g["step"])
    assert abs(time - float(g["time"])) <= 1e-12 * max(1.0, abs(float(g["time"])))
    for key in [k for k in g if k.startswith("field_")]:
        ref = g[key]
        mine = fields[key[6:]]
        err = np.linalg.norm((mask_corners(mine) - mask_corners(ref)).ravel())
        den = np.linalg.norm(mask_corners(ref).ravel())
        floor = 1e-12 * max(1.0, np.linalg.norm(g["field_u"].ravel()))   # v, p may be ~0
        assert err <= 1e-9 * den + floor, (key, err, den)
    h = g["history"]
    logged = {int(r[0]): r for r in h}
    for rec in hist:
        if rec["step"] in logged:
            r = logged[rec["step"]]
            assert rec["inner"] == int(r[2])
            assert abs(rec["cont"] - r[3]) <= 1e-6 * r[3] + 1e-12
            assert abs(rec["time"] - r[1]) <= 1e-12 * max(1.0, r[1])
    if "friction" in g:
        fr = {int(r[0]): r for r in g["friction"]}
        for rec in hist:
            if rec["step"] in fr and "friction" in rec:
                r = fr[rec["step"]]
                np.testing.assert_allclose(rec["friction"]["cf"], r[2], rtol=1e-8)
                np.testing.assert_allclose(rec["friction"]["bulk_velocity"], r[4], rtol=1e-10)


def test_committed_run_logs_and_state():
    """Tier A: the reference's own committed 5x20 k-omega run (real pyamg, rtol 1e-8):
    residual / friction logs lines 2-4 and states/k_omega/fields_000000.npz."""
    cfg, mp, _ = cases.channel_komega_5x20(steps=21, tol=1e-8)
    mesh = orc.rect_mesh(**mp)
    cases.ramp_initial_u(cfg, mesh)
    snap = {}

    def grab(fields, rec):
        if rec["step"] == 0:
            snap.update({k: v.copy() for k, v in fields.items()})

    _, _, _, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True, on_step=grab)
    res = np.loadtxt(GOLDEN / "ref_runs" / "channel_k_omega_residual_log.txt", skiprows=1)[:3]
    fric = np.loadtxt(GOLDEN / "ref_runs" / "channel_k_omega_friction_log.txt", skiprows=1)[:3]
    by_step = {r["step"]: r for r in hist}
    for row, frow in zip(res, fric):
        r = by_step[int(row[0])]
        assert r["inner"] == int(row[2])
        assert abs(r["time"] - row[1]) <= 2e-6 * max(row[1], 1e-30)
        assert abs(r["cont"] - row[3]) <= 2e-6 * row[3]
        assert abs(4.0 * r["friction"]["cf"] - frow[2]) <= 2e-6 * abs(frow[2])
        assert abs(r["friction"]["bulk_velocity"] - frow[5]) <= 2e-6 * abs(frow[5])
    st = load_golden("ref_runs/k_omega_fields_000000.npz")
    for k in ("u", "k", "omega", "nu_t"):
        assert rel_l2(mask_corners(snap[k]), mask_corners(st[k])) < 1e-7, k
    for k in ("v", "p"):
        assert np.max(np.abs(mask_corners(snap[k]) - mask_corners(st[k]))) < 1e-9, k


def test_committed_sst_run_logs_and_state():
    """Tier A for the SST model: the reference's committed 5x20 SST run (real pyamg, rtol 1e-8):
    residual / friction logs lines 2-4 and states/sst/fields_000000.npz incl. the cached SST planes."""
    cfg, mp, _ = cases.channel_komega_5x20(steps=21, tol=1e-8, model="sst")
    mesh = orc.rect_mesh(**mp)
    cases.ramp_initial_u(cfg, mesh)
    snap = {}

    def grab(fields, rec):
        if rec["step"] == 0:
            snap.update({k: v.copy() for k, v in fields.items()})

    _, _, _, hist = orc.run_case(cfg, mesh, solver_mode="direct", freeze_pressure=True, on_step=grab)
    res = np.loadtxt(GOLDEN / "ref_runs" / "channel_sst_residual_log.txt", skiprows=1)[:3]
    fric = np.loadtxt(GOLDEN / "ref_runs" / "channel_sst_friction_log.txt", skiprows=1)[:3]
    by_step = {r["step"]: r for r in hist}
    for row, frow in zip(res, fric):
        r = by_step[int(row[0])]
        assert r["inner"] == int(row[2])
        assert abs(r["time"] - row[1]) <= 2e-6 * max(row[1], 1e-30)
        assert abs(r["cont"] - row[3]) <= 2e-6 * row[3]
        assert abs(4.0 * r["friction"]["cf"] - frow[2]) <= 2e-6 * abs(frow[2])
    st = load_golden("ref_runs/sst_fields_000000.npz")
    for k in ("u", "k", "omega", "nu_t", "_S", "_F1", "_F2", "_D_k", "_D_omega", "_d_wall"):
        assert rel_l2(mask_corners(snap[k]), mask_corners(st[k])) < 1e-7, k
    for k in ("v", "p"):
        assert np.max(np.abs(mask_corners(snap[k]) - mask_corners(st[k]))) < 1e-9, k
