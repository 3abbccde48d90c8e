"""Golden-vector generator (TEST INFRASTRUCTURE; runs only where /root/reference exists).

Imports the reference solver from /root/reference/"Project 1" and writes small
input/output fixtures to tests/golden/.  The reference never travels to the GPU box;
the .npz / .txt files written here do.

Three tiers, recorded in every file's `tier` entry:
  A  data the reference's own runs hold (Runs/Coarse_mesh mesh, states, logs) copied
     verbatim into tests/golden/ref_runs/ (data files, loaded with allow_pickle=False);
  B  the UNMODIFIED reference run here: per-function vectors (pure numpy functions)
     and laminar end-to-end cases through the reference's own no-pyamg branch
     (ILU-CG, fresh pressure matrix every solve, solver.py:834-845) with the CONFIG
     tolerances tightened to 1e-13;
  C  the reference with a harness-side module named `pyamg` (pyamg is an un-vendored,
     unpinned dependency that is absent here): `ruge_stuben_solver(A)`
     -> object whose .solve(b, tol, maxiter) is a sparse LU of A factored ONCE, i.e.
     the reference's cached-hierarchy (frozen pressure operator) semantics with an
     exact solve.  Tier C is cross-checked here against tier A (the committed logs of
     the real-pyamg run) before anything is written.

Usage:  python oracle/gen_golden.py            (regenerates everything but the SST and the large rows)
        python oracle/gen_golden.py sst        (SST rows: ref_runs/sst_*, e2e_*sst*)
        python oracle/gen_golden.py large      (func_rect_136x20, func_annulus_132x24, e2e_*_64_*, e2e_*_160x96_*)
        python oracle/gen_golden.py phase      (phase_<tag>.npz: per-phase function outputs on the func_<tag> states)
"""
from __future__ import annotations

import copy
import io
import os
import shutil
import sys
import tempfile
import types
from contextlib import redirect_stdout
from pathlib import Path

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np
import scipy.sparse.linalg as spla

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
OUT = REPO / "tests" / "golden"
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REF / "Project 1"))

from oracle import cases  # noqa: E402


# ---------------------------------------------------------------- pyamg stand-in
class _FactoredOnce:
    def __init__(self, a):
        self._lu = spla.splu(a.tocsc())

    def solve(self, b, tol=1e-8, maxiter=200):
        return self._lu.solve(np.asarray(b, dtype=float))


def install_pyamg_standin():
    mod = types.ModuleType("pyamg")
    mod.ruge_stuben_solver = lambda a: _FactoredOnce(a)
    mod.__harness_standin__ = True
    sys.modules["pyamg"] = mod


def remove_pyamg_standin():
    sys.modules.pop("pyamg", None)


# ---------------------------------------------------------------- reference import
def load_reference():
    import solver as ref_solver          # noqa: F401  (reference)
    import mesh as ref_mesh              # noqa: F401
    import bc as ref_bc                  # noqa: F401
    import turbulence as ref_turb        # noqa: F401
    import post_processing as ref_post   # noqa: F401
    return ref_solver, ref_mesh, ref_bc, ref_turb, ref_post


def ref_mesh_dict(ref_mesh, kind, params):
    gen = ref_mesh.generate_rect_mesh if kind == "rect" else ref_mesh.generate_annulus_mesh
    return gen(**params)


# ---------------------------------------------------------------- end-to-end runs
def run_reference_case(ref_solver, ref_mesh, builder, fixup, tier):
    """Run one authored case through reference.run_case; capture full-precision
    residual history by wrapping the reference's log writers (harness side)."""
    cfg, mp, kind = builder()
    gen = ref_mesh.generate_rect_mesh if kind == "rect" else ref_mesh.generate_annulus_mesh
    tmp = Path(tempfile.mkdtemp(prefix="golden_"))
    try:
        mesh_path = tmp / "mesh" / "mesh.npz"
        cfg["mesh"] = {"generator": gen, "params": mp, "path": mesh_path}
        cfg["paths"] = {"case_dir": tmp, "mesh_dir": tmp / "mesh",
                        "states_dir": tmp / "states", "post_dir": tmp / "postproc"}
        mesh_path.parent.mkdir(parents=True)
        mesh = ref_mesh.make_mesh(cfg["mesh"])
        if fixup is not None:
            fixup(cfg, mesh)
        hist, fric = [], []
        orig_res, orig_fr = ref_solver.log_residual, ref_solver.log_friction

        def cap_res(post_dir, case_name, step, time, inner, res, mu, mv):
            hist.append((step, time, inner, res, mu, mv))
            return orig_res(post_dir, case_name, step, time, inner, res, mu, mv)

        def cap_fr(post_dir, case_name, step, time, cf, tau, bulk_velocity=None, re=None, cf_dpdx=None):
            fric.append((step, time, cf, tau, bulk_velocity if bulk_velocity is not None else np.nan,
                         re if re is not None else np.nan, cf_dpdx if cf_dpdx is not None else np.nan))
            return orig_fr(post_dir, case_name, step, time, cf, tau, bulk_velocity=bulk_velocity,
                           re=re, cf_dpdx=cf_dpdx)

        ref_solver.log_residual, ref_solver.log_friction = cap_res, cap_fr
        run_cfg = copy.deepcopy({k: v for k, v in cfg.items() if k != "mesh"})
        run_cfg["mesh"] = cfg["mesh"]
        try:
            with redirect_stdout(io.StringIO()):
                fields, step, time = ref_solver.run_case(run_cfg)
        finally:
            ref_solver.log_residual, ref_solver.log_friction = orig_res, orig_fr
        res_txt = (tmp / "postproc" / f"{cfg['case_name']}_residual_log.txt")
        out = {f"field_{k}": v for k, v in fields.items() if k != "p_corr"}
        out["step"] = np.array(step)
        out["time"] = np.array(time)
        out["history"] = np.array(hist, dtype=float).reshape(-1, 6)
        if fric:
            out["friction"] = np.array(fric, dtype=float)
        out["residual_log_text"] = np.array(res_txt.read_text() if res_txt.exists() else "")
        out["tier"] = np.array(tier)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------- per-function vectors
def function_vectors(ref_solver, ref_mesh, ref_bc, ref_turb, ref_post, kind, params, seed, bc_cfg, nu, tag, dense=True):
    mesh = ref_mesh_dict(ref_mesh, kind, params)
    ni, nj = int(mesh["count_i"]), int(mesh["count_j"])
    rng = np.random.default_rng(seed)
    shp = (ni + 2, nj + 2)
    f = {"u": rng.standard_normal(shp), "v": rng.standard_normal(shp),
         "p": rng.standard_normal(shp)}
    f["k"] = np.abs(rng.standard_normal(shp)) * 1e-2 + 1e-4
    f["omega"] = np.abs(rng.standard_normal(shp)) * 5.0 + 1.0
    f["nu_t"] = np.zeros(shp)
    out = {f"mesh_{k}": np.asarray(v) for k, v in mesh.items()}
    out.update({f"in_{k}": v.copy() for k, v in f.items()})
    turb_cfg = {"enabled": True, "model": "k_omega"}
    dt = 0.013

    # BCs (flow, then turbulence)
    g = {k: v.copy() for k, v in f.items()}
    ref_bc.apply_boundary_conditions(g, mesh, bc_cfg)
    out.update({f"bc_{k}": v.copy() for k, v in g.items()})
    ref_turb.initialize(g, mesh, turb_cfg, bc_cfg)
    ref_turb.apply_bcs(g, mesh, bc_cfg, turb_cfg, nu)
    out.update({f"tbc_{k}": g[k].copy() for k in ("k", "omega")})
    nut = ref_turb.eddy_viscosity(g, turb_cfg).copy()
    out["nu_t"] = nut
    nu_eff = nu + nut[1:-1, 1:-1]

    a_i, a_j = ref_solver._diffusion_coefficients(mesh, nu_eff)
    out["a_i"], out["a_j"] = a_i, a_j
    a_i0, a_j0 = ref_solver._diffusion_coefficients(mesh, nu)
    out["a_i_const"], out["a_j_const"] = a_i0, a_j0
    ap = ref_solver._momentum_diagonal(mesh, nu_eff, dt)
    out["a_p"] = ap
    fi, fj = ref_solver._face_mass_flux(g, mesh, ap, use_rhie_chow=True, p_field=g["p"])
    out["flux_i"], out["flux_j"] = fi, fj
    vi, vj = ref_solver._face_volumetric_flux(g, mesh)
    out["vflux_i"], out["vflux_j"] = vi, vj
    out["dt"] = np.array(ref_solver.compute_adaptive_dt(g, mesh, 0.5, 10.0, 0.0, cfl_diff=0.4, nu_eff=nu_eff))
    out["dt_limits"] = np.array(ref_solver._dt_limit_estimate(g, mesh, 0.5, 0.4, nu_eff))
    for name in ("u", "v", "k", "omega"):
        out[f"adv_{name}"] = ref_solver.advect_scalar(g[name], mesh, fi, fj)
    for name in ("u", "v", "p", "k"):
        out[f"grad_{name}"] = ref_solver.grad_scalar(g[name], mesh)
    sk, sw = ref_turb._sources_komega(g, out["grad_u"], out["grad_v"])
    out["src_k"], out["src_omega"] = sk, sw

    src = rng.standard_normal((ni, nj))
    out["in_source"] = src
    a_mat, b, ap2 = ref_solver._assemble_implicit_system(g["u"], mesh, nu_eff, dt, src, bc_cfg=bc_cfg)
    if dense:
        out["sys_A"] = a_mat.toarray()
    else:       # large meshes: CSR triplets instead of the dense matrix
        csr = a_mat.tocsr()
        out["sys_A_data"], out["sys_A_indices"], out["sys_A_indptr"] = csr.data, csr.indices, csr.indptr
    out["sys_b"], out["sys_ap"] = b, ap2

    cap = {}
    orig = ref_solver._solve_pressure_system

    def grab(a, bb, tol=1e-8, maxiter=200, amg_cache=None):
        cap["A"], cap["b"] = (a.toarray() if dense else a.tocsr()), np.array(bb)
        return spla.spsolve(a.tocsc(), bb)

    ref_solver._solve_pressure_system = grab
    try:
        h = {k: v.copy() for k, v in g.items()}
        ref_solver.solve_pressure(h, mesh, dt, nu_eff, bc_cfg, a_p_override=ap, p_field=h["p"])
    finally:
        ref_solver._solve_pressure_system = orig
    if dense:
        out["psys_A"] = cap["A"]
    else:
        out["psys_A_data"], out["psys_A_indices"], out["psys_A_indptr"] = cap["A"].data, cap["A"].indices, cap["A"].indptr
    out["psys_b"] = cap["b"]
    out["p_corr"] = h["p_corr"].copy()

    u0 = rng.standard_normal((ni, nj))
    v0 = rng.standard_normal((ni, nj))
    p0 = rng.standard_normal((ni, nj))
    out["in_u0"], out["in_v0"], out["in_p0"] = u0, v0, p0
    ref_solver.correct_velocity(h, mesh, bc_cfg, ap, ap, u0, v0, p0, 0.7, 0.6, 0.3)
    out["corr_u"], out["corr_v"], out["corr_p"] = h["u"].copy(), h["v"].copy(), h["p"].copy()
    out["corr_p_corr"] = h["p_corr"].copy()
    out["continuity"] = np.array(ref_solver.continuity_residual(g, mesh, nu_eff, dt, p_field=g["p"]))
    for side in ("j_min", "i_min"):
        fr = ref_post.friction_coefficient_array(g, mesh, side, None, nu=nu, hydraulic_diameter=0.7,
                                                 pressure_gradient=(-0.3, 0.0))
        out[f"friction_{side}"] = np.array([fr["cf"], fr["tau_avg"], fr["bulk_velocity"], fr["re"], fr["cf_dpdx"]])
    out["params"] = np.array([nu, dt])
    out["tier"] = np.array("B")
    np.savez_compressed(OUT / f"func_{tag}.npz", **out)
    print(f"wrote func_{tag}.npz")


# ---------------------------------------------------------------- tier A copies
def copy_reference_artifacts():
    dst = OUT / "ref_runs"
    dst.mkdir(parents=True, exist_ok=True)
    run = REF / "Runs" / "Coarse_mesh"
    shutil.copyfile(run / "mesh" / "Coarse_mesh.npz", dst / "Coarse_mesh.npz")
    for s in ("000000", "000300", "049800"):
        shutil.copyfile(run / "states" / "k_omega" / f"fields_{s}.npz", dst / f"k_omega_fields_{s}.npz")
    for name in ("channel_k_omega_residual_log.txt", "channel_k_omega_friction_log.txt"):
        lines = (run / "postproc" / "k_omega" / name).read_text().splitlines()[:201]
        (dst / name).write_text("\n".join(lines) + "\n")
    for p in dst.iterdir():
        os.chmod(p, 0o644)
    print("copied tier-A artefacts")


def selfcheck_against_committed(out):
    """Tier C vs tier A: the 5x20 k-omega run with the reference's own tolerances must
    reproduce the committed logs."""
    txt = (REF / "Runs/Coarse_mesh/postproc/k_omega/channel_k_omega_residual_log.txt").read_text().splitlines()
    committed = np.array([[float(x) for x in ln.split()] for ln in txt[1:4]])
    hist = out["history"]
    mine = hist[np.isin(hist[:, 0], committed[:, 0])]
    rel = np.abs(mine[:, 3] - committed[:, 3]) / committed[:, 3]
    print("selfcheck continuity_l2 rel diff vs committed log:", rel)
    assert np.all(rel < 5e-6), rel
    trel = np.abs(mine[:, 1] - committed[:, 1]) / np.maximum(committed[:, 1], 1e-30)
    assert np.all(trel < 5e-6), trel


def main_sst():
    """SST rows only (python oracle/gen_golden.py sst): tier-A artefacts of the committed SST run,
    tier-C runs of the authored SST cases, cross-checked against the committed SST logs."""
    OUT.mkdir(parents=True, exist_ok=True)
    dst = OUT / "ref_runs"
    dst.mkdir(parents=True, exist_ok=True)
    run = REF / "Runs" / "Coarse_mesh"
    shutil.copyfile(run / "states" / "sst" / "fields_000000.npz", dst / "sst_fields_000000.npz")
    for name in ("channel_sst_residual_log.txt", "channel_sst_friction_log.txt"):
        lines = (run / "postproc" / "sst" / name).read_text().splitlines()[:201]
        (dst / name).write_text("\n".join(lines) + "\n")
    for q in dst.iterdir():
        os.chmod(q, 0o644)
    ref_solver, ref_mesh, ref_bc, ref_turb, ref_post = load_reference()
    install_pyamg_standin()
    try:
        chk = run_reference_case(ref_solver, ref_mesh,
                                 lambda: cases.channel_komega_5x20(steps=21, tol=1e-8, model="sst"),
                                 cases.ramp_initial_u, "C")
        txt = (run / "postproc/sst/channel_sst_residual_log.txt").read_text().splitlines()
        committed = np.array([[float(x) for x in ln.split()] for ln in txt[1:4]])
        hist = chk["history"]
        mine = hist[np.isin(hist[:, 0], committed[:, 0])]
        rel = np.abs(mine[:, 3] - committed[:, 3]) / committed[:, 3]
        print("SST selfcheck continuity_l2 rel diff vs committed log:", rel)
        assert np.all(rel < 5e-6), rel
        np.savez_compressed(OUT / "e2e_channel_sst_5x20_reftol.npz", **chk)
        for name, (builder, fixup) in cases.SST_END_TO_END.items():
            out = run_reference_case(ref_solver, ref_mesh, builder, fixup, "C")
            np.savez_compressed(OUT / f"e2e_{name}_frozen.npz", **out)
            print(f"wrote e2e_{name}_frozen.npz  steps={int(out['step'])}")
    finally:
        remove_pyamg_standin()


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    copy_reference_artifacts()
    remove_pyamg_standin()
    ref_solver, ref_mesh, ref_bc, ref_turb, ref_post = load_reference()

    # ---- tier B: per-function vectors
    nr, nt = 12, 16
    ann_bc = cases._half_cylinder_bc(nr, nt, True)
    function_vectors(ref_solver, ref_mesh, ref_bc, ref_turb, ref_post, "annulus",
                     dict(radius_inner=0.5, radius_outer=6.0, count_r=nr, count_t=nt,
                          theta_min=0.0, theta_max=float(np.pi), ratio_r=1.15, ratio_t=1.0),
                     11, ann_bc, 2.0e-2, "annulus_12x16")
    ni, nj = 8, 10
    rect_bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "symmetry", "range": (0, ni - 1)},
        "pressure_reference": {"cell": (1, 2), "value": 0.25},
    }
    function_vectors(ref_solver, ref_mesh, ref_bc, ref_turb, ref_post, "rect",
                     dict(length_i=1.0, length_j=0.5, count_i=ni, count_j=nj, ratio_i=1.1, ratio_j=1.2),
                     12, rect_bc, 1.0e-3, "rect_8x10")

    # ---- tier B: laminar end-to-end, unmodified reference (fresh pressure matrix)
    for name in ("lam_channel", "inlet_outlet_channel", "couette_piso", "half_cylinder", "channel_forced_24"):
        builder, fixup = cases.ALL_END_TO_END[name]
        out = run_reference_case(ref_solver, ref_mesh, builder, fixup, "B")
        np.savez_compressed(OUT / f"e2e_{name}_fresh.npz", **out)
        print(f"wrote e2e_{name}_fresh.npz  steps={int(out['step'])}")

    # ---- tier C: frozen operator / k-omega via the harness pyamg stand-in
    install_pyamg_standin()
    try:
        chk = run_reference_case(ref_solver, ref_mesh,
                                 lambda: cases.channel_komega_5x20(steps=21, tol=1e-8),
                                 cases.ramp_initial_u, "C")
        selfcheck_against_committed(chk)
        np.savez_compressed(OUT / "e2e_channel_komega_5x20_reftol.npz", **chk)
        for name, (builder, fixup) in cases.ALL_END_TO_END.items():
            out = run_reference_case(ref_solver, ref_mesh, builder, fixup, "C")
            np.savez_compressed(OUT / f"e2e_{name}_frozen.npz", **out)
            print(f"wrote e2e_{name}_frozen.npz  steps={int(out['step'])}")
    finally:
        remove_pyamg_standin()


def main_large():
    """python oracle/gen_golden.py large: per-function vectors on meshes wider than two column strips of the
    marching kernels (>= 130 columns) and the end-to-end cases of cases.LARGE_END_TO_END (tier C: frozen
    operator through the harness pyamg stand-in, CONFIG tolerances 1e-13)."""
    OUT.mkdir(parents=True, exist_ok=True)
    remove_pyamg_standin()
    ref_solver, ref_mesh, ref_bc, ref_turb, ref_post = load_reference()
    ni, nj = 136, 20
    rect_bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "symmetry", "range": (0, ni - 1)},
        "pressure_reference": {"cell": (70, 3), "value": 0.25},
    }
    function_vectors(ref_solver, ref_mesh, ref_bc, ref_turb, ref_post, "rect",
                     dict(length_i=3.0, length_j=0.5, count_i=ni, count_j=nj, ratio_i=1.01, ratio_j=1.1),
                     21, rect_bc, 1.0e-3, "rect_136x20", dense=False)
    nr, nt = 132, 24
    function_vectors(ref_solver, ref_mesh, ref_bc, ref_turb, ref_post, "annulus",
                     dict(radius_inner=0.5, radius_outer=6.0, count_r=nr, count_t=nt,
                          theta_min=0.0, theta_max=float(np.pi), ratio_r=1.02, ratio_t=1.0),
                     22, cases._half_cylinder_bc(nr, nt, True), 2.0e-2, "annulus_132x24", dense=False)
    install_pyamg_standin()
    try:
        for name, (builder, fixup) in cases.LARGE_END_TO_END.items():
            out = run_reference_case(ref_solver, ref_mesh, builder, fixup, "C")
            out.pop("residual_log_text", None)
            np.savez_compressed(OUT / f"e2e_{name}_frozen.npz", **out)
            print(f"wrote e2e_{name}_frozen.npz  steps={int(out['step'])}", flush=True)
    finally:
        remove_pyamg_standin()


def main_phase():
    """python oracle/gen_golden.py phase: outputs of the reference's per-phase functions whose results the func_*
    vectors do not hold yet (solve_momentum, update_turbulence for both closures, the scalar-viscosity branches),
    on the boundary-conditioned state of func_<tag>.npz.  Written to phase_<tag>.npz (tier C for the k / omega
    solves, which go through `pyamg.ruge_stuben_solver`, solver.py:1138-1139: harness stand-in, exact solve)."""
    ref_solver, ref_mesh, ref_bc, ref_turb, ref_post = load_reference()
    tight = {"tol": 1.0e-13, "maxiter": 2000}
    for tag, bc_cfg in cases.FUNCTION_VECTOR_BCS.items():
        with np.load(OUT / f"func_{tag}.npz", allow_pickle=False) as z:
            g = {k: z[k] for k in z.files}
        mesh = {k[5:]: v for k, v in g.items() if k.startswith("mesh_")}
        nu, dt = (float(x) for x in g["params"])
        state = {k: g[f"bc_{k}"].copy() for k in ("u", "v", "p")}
        state["k"], state["omega"], state["nu_t"] = g["tbc_k"].copy(), g["tbc_omega"].copy(), g["nu_t"].copy()
        nu_eff = nu + state["nu_t"][1:-1, 1:-1]
        out = {}

        # predictor from the caller's advective terms, forcing and the 2/3 k pressure-like term
        h = {k: v.copy() for k, v in state.items()}
        adv = ref_solver.compute_advective_rhs(h, mesh, nu_eff, dt, use_rhie_chow=True, p_field=h["p"])
        out["adv_rhs_u"], out["adv_rhs_v"] = adv["u"], adv["v"]
        info = ref_solver.solve_momentum(h, mesh, dt, nu_eff, adv, bc_cfg, tight, p_field=h["p"],
                                         pressure_gradient=(-0.3, 0.1), extra_pressure_field=(2.0 / 3.0) * h["k"])
        out["mom_u"], out["mom_v"], out["mom_a_p"] = h["u"].copy(), h["v"].copy(), info["a_p_u"]
        out["mom_l2"] = np.array([info["mom_u_l2"], info["mom_v_l2"]])

        # the scalar-viscosity branches (solver.py:205-207)
        h = {k: v.copy() for k, v in state.items()}
        out["a_p_scalar"] = ref_solver._momentum_diagonal(mesh, nu, dt)
        adv0 = ref_solver.compute_advective_rhs(h, mesh, nu, dt, use_rhie_chow=True)
        out["adv0_u"], out["adv0_v"], out["adv0_flux_i"], out["adv0_flux_j"] = adv0["u"], adv0["v"], adv0["flux_i"], adv0["flux_j"]
        out["continuity_scalar"] = np.array(ref_solver.continuity_residual(h, mesh, nu, dt))
        out["continuity_norc"] = np.array(ref_solver.continuity_residual(h, mesh, nu, dt, use_rhie_chow=False))
        out["dt_scalar"] = np.array(ref_solver.compute_adaptive_dt(h, mesh, 0.5, 10.0, 0.0, cfl_diff=0.4, nu_eff=nu))
        out["dt_adv_only"] = np.array(ref_solver.compute_adaptive_dt(h, mesh, 0.5, 10.0))
        out["dt_clamped"] = np.array(ref_solver.compute_adaptive_dt(h, mesh, 0.5, 1.0e-7, 0.0, cfl_diff=0.4, nu_eff=nu_eff))
        info0 = ref_solver.solve_momentum(h, mesh, dt, nu, adv0, bc_cfg, tight)
        out["mom0_u"], out["mom0_v"] = h["u"].copy(), h["v"].copy()

        # pressure correction with the assembled diagonal (no override) and the scalar viscosity
        # (through the stand-in's exact solve: the reference's ILU-CG does not reach 1e-13 on the seam-asymmetric
        # operator of rect_136x20)
        install_pyamg_standin()
        try:
            h = {k: v.copy() for k, v in state.items()}
            pc = ref_solver.solve_pressure(h, mesh, dt, nu, bc_cfg, tight)
            out["p_corr_scalar"] = pc.copy()

            # one transport step of either closure
            for model in ("k_omega", "sst"):
                h = {k: v.copy() for k, v in state.items()}
                cfg = {"enabled": True, "model": model}
                ref_solver.update_turbulence(h, mesh, cfg, bc_cfg, dt, nu)
                ref_turb.eddy_viscosity(h, cfg)
                for key, val in h.items():
                    if key in ("k", "omega", "nu_t") or key.startswith("_"):
                        out[f"turb_{model}_{key}"] = np.asarray(val).copy()
        finally:
            remove_pyamg_standin()
        out["tier"] = np.array("C")
        np.savez_compressed(OUT / f"phase_{tag}.npz", **out)
        print(f"wrote phase_{tag}.npz", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "phase":
        main_phase()
    elif len(sys.argv) > 1 and sys.argv[1] == "sst":
        main_sst()
    elif len(sys.argv) > 1 and sys.argv[1] == "large":
        main_large()
    else:
        main()
