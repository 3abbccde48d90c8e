"""Synthetic benchmark workload of SURVEY.md 8(d): periodic-i channel with walls and
pressure-gradient forcing on an N x N rectangular mesh (CONFIG in the reference's schema)."""
from __future__ import annotations

import numpy as np


def wall_graded_ratio(nj, max_over_min=20.0):
    """ratio_j of a channel refined towards BOTH walls (two geometric segments, the reference's segmented spacing,
    mesh.py:4-79) whose largest and smallest dy differ by `max_over_min` (~20)."""
    r = float(max_over_min) ** (1.0 / (nj // 2 - 1))
    return [(0.5, r), (0.5, 1.0 / r)]


def channel_forced_config(n, turbulent=False, max_iter=2, tol=1.0e-8, ratio_j=1.0, steps=1, nj=None, length_j=1.0):
    """nj / length_j: taller channel of the weak-scaling runs (n x nj cells, same cell size when
    length_j = nj / n)."""
    nj = n if nj is None else int(nj)
    mesh_params = dict(length_i=1.0, length_j=float(length_j), count_i=n, count_j=nj, ratio_i=1.0, ratio_j=ratio_j)
    boundary = {
        "i_min": {"type": "periodic", "range": (0, nj - 1)},
        "i_max": {"type": "periodic", "range": (0, nj - 1)},
        "j_min": {"type": "wall", "range": (0, n - 1)},
        "j_max": {"type": "wall", "range": (0, n - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    initial = {"u": None, "v": 0.0, "p": 0.0}
    turbulence = {"enabled": False, "model": "none"}
    nu, dt_max = 1.0e-3, 1.0e-3
    if turbulent:
        turbulence = {"enabled": True, "model": "k_omega", "update_interval": 1}
        initial["scalars"] = {"k": 1.0e-4, "omega": 10.0}
        # The reference integrates the omega sink -beta omega^2 explicitly (turbulence.py:440-442): the step
        # is stable only while dt * beta * omega < 1, and omega reaches 85 nu / y^2 in the wall cells
        # (turbulence.py:281).  On fine meshes that bound lies far below the CFL limits (8192^2: 5.8e-7 against
        # 7.5e-6), omega overshoots below zero, is clipped, and nu_t = k / omega explodes -- in the reference
        # as much as here.  The k-omega variant of the workload therefore caps dt at 0.4 of that bound.
        y_wall = 0.5 * float(length_j) / nj
        if isinstance(ratio_j, (list, tuple)):       # wall-graded: first cell of the first segment
            frac, r = ratio_j[0]
            m = nj // 2
            y_wall = 0.5 * float(length_j) * frac * (r - 1.0) / (r ** m - 1.0)
        dt_max = min(dt_max, 0.4 / (0.075 * 85.0 * nu / (y_wall * y_wall)))
    config = {
        "case_name": "channel_forced",
        "physics": {"nu": nu, "pressure_gradient": (-0.1, 0.0)},
        "time": {"cfl": 0.5, "cfl_diff": 0.5, "dt_max": dt_max, "dt_min": 0.0, "max_steps": steps},
        "solver": {
            "scheme": "simple", "rhie_chow": True,
            "momentum": {"tol": tol, "maxiter": 500},
            "pressure": {"tol": tol, "maxiter": 500},
            "turbulence_solver": {"tol": tol, "maxiter": 500},
            "simple": {"max_iter": max_iter, "continuity_tol": 0.0, "relax_u": 0.7, "relax_v": 0.7, "relax_p": 0.3},
        },
        "restart": {"enabled": False},
        "boundary": boundary,
        "turbulence": turbulence,
        "initial": initial,
        "post": {"plot_interval": 0, "save_interval": 0, "print_interval": 0},
    }
    return config, mesh_params


def perturbed_parabola_rows(n, y_nodes, seed=1234, u_max=1.0, j_offset=0, nj_local=None):
    """u = U (1 - (2y/H - 1)^2) + 1e-3 N(0,1) on the (Ni, Nj) interior (rect mesh: cell-centre
    y is the mid-point of the node rows).  j_offset / nj_local: only the rows of one j-slab, with
    the noise stream seeded per slab (seed + j_offset) so that no rank draws the whole field."""
    y_nodes = np.asarray(y_nodes)
    yc = 0.5 * (y_nodes[1:] + y_nodes[:-1])
    h = float(y_nodes[-1] - y_nodes[0])
    nj_local = len(yc) - j_offset if nj_local is None else int(nj_local)
    yc = yc[j_offset:j_offset + nj_local]
    rng = np.random.default_rng(seed + j_offset)
    return (u_max * (1.0 - (2.0 * yc / h - 1.0) ** 2))[None, :] + 1.0e-3 * rng.standard_normal((n, nj_local))


def half_annulus_config(nr, nt, max_iter=2, tol=1.0e-8, steps=1, r_in=0.5, r_out=20.0, ratio_r=None):
    """Curvilinear variant of the workload (4096 x 2048 half annulus, r_in 0.5, r_out 20, ratio_r ~
    1.002): flow past a half cylinder, i = radial, j = azimuthal; wall on the cylinder, outer boundary split into
    outlet (downstream half) and inlet (upstream half), symmetry on the axis.  ratio_r defaults to the growth that
    makes the outermost cells square (dr = r dtheta at r_out)."""
    import math
    if ratio_r is None:
        dth = math.pi / nt
        ratio_r = 1.0 + dth * (1.0 - 0.0)      # dr_k = dr_0 ratio^k with dr / r -> dtheta for large k
        ratio_r = min(ratio_r, 1.0 + 8.2 / nr)  # keep dr_0 representable: total growth <= e^8.2 ~ 3600 (1.002 at nr = 4096)
    half = nt // 2
    boundary = {
        "i_min": {"type": "wall", "range": (0, nt - 1)},
        "i_max": [{"type": "outlet", "range": (0, half - 1), "pressure": 0.0},
                  {"type": "inlet", "range": (half, nt - 1), "velocity": (1.0, 0.0)}],
        "j_min": {"type": "symmetry", "range": (0, nr - 1)},
        "j_max": {"type": "symmetry", "range": (0, nr - 1)},
        "pressure_reference": {"cell": (0, 0), "value": 0.0},
    }
    mesh_params = dict(radius_inner=r_in, radius_outer=r_out, count_r=nr, count_t=nt, theta_min=0.0,
                       theta_max=math.pi, ratio_r=ratio_r, ratio_t=1.0)
    config = {
        "case_name": "half_annulus",
        "physics": {"nu": 2.0e-2, "pressure_gradient": (0.0, 0.0)},
        "time": {"cfl": 0.5, "cfl_diff": 0.5, "dt_max": 0.05, "dt_min": 0.0, "max_steps": steps},
        "solver": {
            "scheme": "simple", "rhie_chow": True,
            "momentum": {"tol": tol, "maxiter": 500},
            "pressure": {"tol": tol, "maxiter": 500},
            "turbulence_solver": {"tol": tol, "maxiter": 500},
            "simple": {"max_iter": max_iter, "continuity_tol": 0.0, "relax_u": 0.6, "relax_v": 0.6, "relax_p": 0.2},
        },
        "restart": {"enabled": False},
        "boundary": boundary,
        "turbulence": {"enabled": False, "model": "none"},
        "initial": {"u": 1.0, "v": 0.0, "p": 0.0},
        "post": {"plot_interval": 0, "save_interval": 0, "print_interval": 0},
    }
    return config, mesh_params
