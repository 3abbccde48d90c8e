"""Authored case configurations (TEST INFRASTRUCTURE, see oracle/cfd_oracle.py header).

The four README example cases BASELINE.json names (test/, test2/, half_cylinder/,
half_cylinder_turbulent/) are absent from the reference checkout, so
equivalent CONFIG dicts are authored here in the reference's CONFIG schema
(solver.py:1367-1431, bc.py:288-347).  The same dict drives the reference
(oracle/gen_golden.py), the oracle and the HIP build, so the three are compared on
identical inputs.  `channel_komega_5x20` restates Runs/Coarse_mesh/run_test.py:25-137.

Each builder returns (config, mesh_params, mesh_kind); the caller supplies the mesh
generator of the implementation under test and all paths.
"""
from __future__ import annotations

import numpy as np

TIGHT = 1.0e-13


def _base(case_name, nu, time, solver, boundary, initial, turbulence=None, forcing=(0.0, 0.0), post=None):
    return {
        "case_name": case_name,
        "physics": {"nu": nu, "pressure_gradient": forcing},
        "time": time,
        "solver": solver,
        "restart": {"enabled": False},
        "boundary": boundary,
        "turbulence": turbulence or {"enabled": False, "model": "none"},
        "initial": initial,
        "post": post or {"plot_interval": 0, "save_interval": 0, "print_interval": 1000000,
                         "residual_log_interval": 1, "friction_log_interval": 1},
    }


def _solver(scheme="simple", max_iter=3, ctol=0.0, relax=(0.7, 0.7, 0.3), n_corr=3, tol=TIGHT):
    return {
        "scheme": scheme, "rhie_chow": True,
        "momentum": {"tol": tol, "maxiter": 2000},
        "pressure": {"tol": tol, "maxiter": 2000},
        "simple": {"max_iter": max_iter, "continuity_tol": ctol,
                   "relax_u": relax[0], "relax_v": relax[1], "relax_p": relax[2]},
        "piso": {"n_correctors": n_corr},
    }


def lam_channel(ni=16, nj=12, steps=5):
    """'test/': laminar periodic channel, SIMPLE, pressure-gradient forcing; Poiseuille
    known answer u = -dpdx/(2 nu) y (H - y) (post_processing.py:385-408)."""
    mp = dict(length_i=2.0, length_j=1.0, count_i=ni, count_j=nj, ratio_i=1.0, ratio_j=1.0)
    bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "wall", "range": (0, ni - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    cfg = _base("lam_channel", 1.0e-2,
                {"cfl": 0.5, "cfl_diff": 0.5, "dt_max": 0.05, "dt_min": 0.0, "max_steps": steps},
                _solver("simple", 3, 0.0), bc, {"u": 0.1, "v": 0.0, "p": 0.0}, forcing=(-0.1, 0.0))
    return cfg, mp, "rect"


def inlet_outlet_channel(ni=20, nj=10, steps=5):
    """Developing channel: inlet i_min, outlet i_max (Dirichlet p, no pin), walls."""
    mp = dict(length_i=3.0, length_j=1.0, count_i=ni, count_j=nj, ratio_i=1.05, ratio_j=1.0)
    bc = {
        "i_min": {"type": "inlet", "range": (0, nj - 1), "velocity": (1.0, 0.0)},
        "i_max": {"type": "outlet", "range": (0, nj - 1), "pressure": 0.0},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "wall", "range": (0, ni - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    cfg = _base("inlet_outlet_channel", 2.0e-2,
                {"cfl": 0.4, "dt_max": 0.05, "dt_min": 0.0, "max_steps": steps,
                 "cfl_schedule": [(0.2, 0), (0.4, 4)]},
                _solver("simple", 4, 1.0e-9), bc, {"u": 1.0, "v": 0.0, "p": 0.0})
    return cfg, mp, "rect"


def couette_piso(ni=8, nj=10, steps=5, dt=0.02):
    """'test2/': Couette flow, periodic i, moving lid, PISO x3, fixed dt expressed as
    cfl = cfl_diff = 1e30 with dt_max = dt (the reference's plain
    fixed-dt mode raises UnboundLocalError at solver.py:1737-1743)."""
    mp = dict(length_i=1.0, length_j=1.0, count_i=ni, count_j=nj, ratio_i=1.0, ratio_j=1.08)
    bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "wall", "range": (0, ni - 1), "velocity": (1.0, 0.0)},
        "pressure_reference": {"cell": (2, 3), "value": 0.0},
    }
    cfg = _base("couette_piso", 5.0e-2,
                {"cfl": 1.0e30, "cfl_diff": 1.0e30, "dt_max": dt, "dt_min": 0.0, "max_steps": steps},
                _solver("piso", 3, 0.0, relax=(1.0, 1.0, 1.0), n_corr=3), bc,
                {"u": None, "v": None, "p": 0.0})
    cfg["_seed"] = 77
    return cfg, mp, "rect"


def perturbed_rest(cfg, mesh, amp=0.05):
    """Non-homogeneous start so that the PISO correctors have work to do."""
    rng = np.random.default_rng(cfg.pop("_seed"))
    shape = mesh["cell_volume"].shape
    cfg["initial"]["u"] = amp * rng.standard_normal(shape)
    cfg["initial"]["v"] = amp * rng.standard_normal(shape)


def _half_cylinder_bc(nr, nt, turbulent):
    half = nt // 2
    outlet = {"type": "outlet", "range": (0, half - 1), "pressure": 0.0}
    inlet = {"type": "inlet", "range": (half, nt - 1), "velocity": (1.0, 0.0)}
    if turbulent:
        inlet.update({"k": 1.0e-3, "omega": 5.0})
    return {
        "i_min": {"type": "wall", "range": (0, nt - 1)},
        "i_max": [outlet, inlet],
        "j_min": {"type": "symmetry", "range": (0, nr - 1)},
        "j_max": {"type": "symmetry", "range": (0, nr - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }


def half_cylinder(nr=12, nt=16, steps=5, turbulent=False, model="k_omega"):
    """'half_cylinder/' and 'half_cylinder_turbulent/': half annulus, i = radial,
    j = azimuthal (theta 0..pi); wall on the cylinder, outer boundary split into
    outlet (downstream half) and inlet (upstream half), symmetry on the x axis."""
    mp = dict(radius_inner=0.5, radius_outer=6.0, count_r=nr, count_t=nt,
              theta_min=0.0, theta_max=float(np.pi), ratio_r=1.15, ratio_t=1.0)
    bc = _half_cylinder_bc(nr, nt, turbulent)
    turb = None
    init = {"u": 1.0, "v": 0.0, "p": 0.0}
    if turbulent:
        turb = {"enabled": True, "model": model, "update_interval": 1}
        init["scalars"] = {"k": 1.0e-3, "omega": 5.0}
    name = ("half_cylinder_sst" if model == "sst" else "half_cylinder_komega") if turbulent else "half_cylinder"
    cfg = _base(name, 2.0e-2,
                {"cfl": 0.5, "dt_max": 0.05, "dt_min": 0.0, "max_steps": steps},
                _solver("simple", 3, 0.0, relax=(0.6, 0.6, 0.2)), bc, init, turbulence=turb)
    return cfg, mp, "annulus"


def channel_komega_5x20(steps=21, tol=1.0e-8, model="k_omega"):
    """Runs/Coarse_mesh/run_test.py:25-137 (the reference's own committed run)."""
    length, height, dpdx, nu = 0.02, 0.05, -0.057, 1.5e-5
    dh = 4.0 * height
    u_bulk = 1.0e4 * nu / dh
    k_val = 1.5 * (u_bulk * 0.05) ** 2
    omega_val = np.sqrt(k_val) / (0.09 ** 0.25 * (0.07 * dh))
    ni, nj = 5, 20
    mp = dict(length_i=length, length_j=height, count_i=ni, count_j=nj, ratio_i=1.0, ratio_j=1.03)
    bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "symmetry", "range": (0, ni - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    solver = _solver("simple", 5, 1.0e-4, relax=(0.5, 0.5, 0.13), tol=tol)
    solver["momentum"]["maxiter"] = 200 if tol >= 1e-8 else 2000
    solver["pressure"]["maxiter"] = 200 if tol >= 1e-8 else 2000
    post = {"plot_interval": 0, "save_interval": 0, "print_interval": 1000000,
            "residual_log_interval": 10, "friction_log_interval": 10,
            "friction": {"enabled": True, "boundary": "j_min", "range": None, "u_ref": None,
                         "rho": 1.0, "use_nu_t": False, "flow_axis": "i",
                         "hydraulic_diameter": dh},
            "line_plot": {"enabled": False}}
    cfg = _base(f"channel_{model}", nu,
                {"cfl": 0.5, "cfl_diff": 0.5, "dt_max": 0.1, "dt_min": 0.0, "max_steps": steps},
                solver, bc, {"u": None, "v": 0.0, "p": 0.0,
                             "scalars": {"k": k_val, "omega": omega_val}},
                turbulence={"enabled": True, "model": model, "update_interval": 1},
                forcing=(dpdx, 0.0), post=post)
    cfg["_u_ramp"] = u_bulk     # initial u = u_bulk * ramp(y), run_test.py:54-58
    return cfg, mp, "rect"


def ramp_initial_u(cfg, mesh):
    """run_test.py:54-58."""
    y = mesh["cell_center"][:, :, 1]
    ramp = np.clip((y - y.min()) / (y.max() - y.min()), 0.0, 1.0)
    cfg["initial"]["u"] = cfg.pop("_u_ramp") * ramp


def channel_forced(n=64, steps=3, turbulent=False, ratio_j=1.0, max_iter=2, seed=1234, tol=TIGHT, model="k_omega",
                   nj=None, name="channel_forced"):
    """Miniature of the 8192^2 benchmark workload: periodic-i channel,
    walls, pressure-gradient forcing, perturbed parabolic initial u (n x nj cells, square by default)."""
    nj = n if nj is None else nj
    mp = dict(length_i=1.0, length_j=1.0, count_i=n, count_j=nj, ratio_i=1.0, ratio_j=ratio_j)
    bc = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, n - 1)},
        "j_max": {"type": "wall", "range": (0, n - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    turb = None
    init = {"u": None, "v": 0.0, "p": 0.0}
    if turbulent:
        turb = {"enabled": True, "model": model, "update_interval": 1}
        init["scalars"] = {"k": 1.0e-4, "omega": 10.0}
    cfg = _base(name, 1.0e-3,
                {"cfl": 0.5, "cfl_diff": 0.5, "dt_max": 1.0e-3, "dt_min": 0.0, "max_steps": steps},
                _solver("simple", max_iter, 0.0, tol=tol), bc, init, turbulence=turb,
                forcing=(-0.1, 0.0))
    cfg["_seed"] = seed
    return cfg, mp, "rect"


def perturbed_parabola(cfg, mesh, u_max=1.0):
    """IC of SURVEY.md 8(d): u = U (1 - (2y/H - 1)^2) + 1e-3 N(0,1), rng 1234."""
    y = mesh["cell_center"][:, :, 1]
    h = float(mesh["node"][0, -1, 1] - mesh["node"][0, 0, 1])
    rng = np.random.default_rng(cfg.pop("_seed"))
    cfg["initial"]["u"] = u_max * (1.0 - (2.0 * y / h - 1.0) ** 2) + 1.0e-3 * rng.standard_normal(y.shape)


ALL_END_TO_END = {
    "lam_channel": (lam_channel, None),
    "inlet_outlet_channel": (inlet_outlet_channel, None),
    "couette_piso": (couette_piso, perturbed_rest),
    "half_cylinder": (lambda: half_cylinder(turbulent=False), None),
    "half_cylinder_komega": (lambda: half_cylinder(turbulent=True), None),
    "channel_komega_5x20": (lambda: channel_komega_5x20(steps=21, tol=TIGHT), ramp_initial_u),
    "channel_forced_24": (lambda: channel_forced(n=24, steps=3), perturbed_parabola),
    "channel_forced_komega_24": (lambda: channel_forced(n=24, steps=3, turbulent=True), perturbed_parabola),
}

# Menter SST variants; channel_sst_5x20 restates the second half of
# Runs/Coarse_mesh/run_test.py (model_name = "sst", same CONFIG otherwise)
SST_END_TO_END = {
    "channel_sst_5x20": (lambda: channel_komega_5x20(steps=21, tol=TIGHT, model="sst"), ramp_initial_u),
    "half_cylinder_sst": (lambda: half_cylinder(turbulent=True, model="sst"), None),
    "channel_forced_sst_24": (lambda: channel_forced(n=24, steps=3, turbulent=True, ratio_j=1.04, model="sst"),
                              perturbed_parabola),
}


# Cases at sizes where the production kernels of the 8192^2 run are the ones that execute:
# >= 3 column strips of the marching flux kernels (160 > 2 x 62), k_chebk<.,4> (both sides >= 64), three fused
# multigrid levels (160x96 -> 80x48 -> 40x24 -> 20x12), PCG solves long enough for deferred residual updates.
# channel_forced_64 / _komega_64 are the 64 x 64 miniatures SURVEY.md 8(c) lists.
def _couette_large():
    cfg, mp, kind = couette_piso(ni=160, nj=96, steps=3, dt=2.0e-3)
    mp["ratio_j"] = 1.02
    cfg["case_name"] = "couette_piso_160x96"
    return cfg, mp, kind


def _inout_large():
    cfg, mp, kind = inlet_outlet_channel(ni=160, nj=96, steps=3)
    mp["ratio_i"] = 1.005
    mp["ratio_j"] = 1.02
    cfg["case_name"] = "inlet_outlet_160x96"
    return cfg, mp, kind


LARGE_END_TO_END = {
    "channel_forced_64": (lambda: channel_forced(n=64, steps=3), perturbed_parabola),
    "channel_forced_komega_64": (lambda: channel_forced(n=64, steps=3, turbulent=True), perturbed_parabola),
    "channel_forced_160x96": (lambda: channel_forced(n=160, nj=96, steps=3, ratio_j=1.02, name="channel_forced_160x96"),
                              perturbed_parabola),
    "channel_forced_komega_160x96": (lambda: channel_forced(n=160, nj=96, steps=3, ratio_j=1.02, turbulent=True,
                                                            name="channel_forced_komega_160x96"), perturbed_parabola),
    "inlet_outlet_160x96": (_inout_large, None),
    "couette_piso_160x96": (_couette_large, perturbed_rest),
}


def _periodic_rect_bc(ni, nj, cell):
    return {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, ni - 1)},
        "j_max": {"type": "symmetry", "range": (0, ni - 1)},
        "pressure_reference": {"cell": cell, "value": 0.25},
    }


# boundary configurations of the per-function vectors tests/golden/func_<tag>.npz (and phase_<tag>.npz)
FUNCTION_VECTOR_BCS = {
    "annulus_12x16": _half_cylinder_bc(12, 16, True),
    "rect_8x10": _periodic_rect_bc(8, 10, (1, 2)),
    "annulus_132x24": _half_cylinder_bc(132, 24, True),
    "rect_136x20": _periodic_rect_bc(136, 20, (70, 3)),
}
