"""Drop-in for the reference's `solver` module on MI355X.

`run_case(CONFIG) -> (fields, step, time)` keeps the reference's call shape, CONFIG
schema, mesh / state `.npz` layouts and log formats (solver.py:1367-1800), but every
field and metric plane lives in HBM and every numerical phase is a HIP kernel behind
libcfd_hip.so.  The host only sequences the outer loop, evaluates schedules, writes the
logs and moves arrays at upload / save / return.  There is no CPU compute path: a
missing extension or GPU raises.

Extensions over the reference (all optional, defaults reproduce the reference):
  solver.pressure.freeze_operator (default True)  keep the first pressure matrix of the
      run for all later solves -- what the reference does when pyamg is installed
      (cached hierarchy keyed on shape only, solver.py:821-833).  False =
      the reference's no-pyamg branch (fresh matrix every solve, solver.py:834-845).
  solver.turbulence_solver.{tol, maxiter}  (reference: hard-coded 1e-8 / 200).
  solver.multigrid.{nu_pre, nu_post, omega, coarse_sweeps, fused}.
  solver.transport.{warm_start, chebyshev, rho_max, max_iters}  momentum / k-omega solver policy
      (Jacobi-Chebyshev for time-step dominated systems, Jacobi-BiCGStab otherwise; the
      reference's stopping test |r| <= tol |b| either way).
  torch.distributed initialised with world > 1: every rank calls run_case with the same CONFIG
      and owns one j-slab (cfd_amd/slab.py); rank 0 writes logs and states.
  solver.device  HIP device ordinal (default 0 or LOCAL_RANK).
  time.dt without time.cfl: true fixed-dt mode (the reference crashes there).
"""
from __future__ import annotations

import os
import time as time_module
from pathlib import Path

import numpy as np

from . import post_processing
from .bc import BoundaryGeometry, BoundaryTables, _normalize_segments
from .device import Device
from .mesh import ensure_single_mesh_file, make_mesh

USE_NUMBA = False            # kept for API compatibility (solver.py:21); the HIP kernels replace numba

# the reference's per-phase functions (README.md:196-205), same call shapes, each executed on the GPU
# (cfd_amd/phases.py); run_case below does not go through them -- its loop keeps every plane in HBM
from .phases import (advect_scalar, apply_boundary_conditions, compute_adaptive_dt, compute_advective_rhs,  # noqa: E402,F401
                     continuity_residual, correct_velocity, grad_scalar, solve_momentum, solve_pressure,
                     update_turbulence)


# ------------------------------------------------------------------ mesh / state IO
def load_mesh(mesh_cfg):
    """solver.py:361-367."""
    mesh_path = Path(mesh_cfg["path"])
    ensure_single_mesh_file(mesh_path)
    if not mesh_path.exists():
        make_mesh(mesh_cfg)
    with np.load(mesh_path, allow_pickle=False) as data:
        return {key: data[key] for key in data.files}


def initialize_fields(mesh, initial_cfg):
    """solver.py:370-387."""
    count_i = int(mesh["count_i"])
    count_j = int(mesh["count_j"])
    shape = (count_i + 2, count_j + 2)
    fields = {name: np.zeros(shape, dtype=float) for name in ("u", "v", "p")}
    for name in ("u", "v", "p"):
        fields[name][1:-1, 1:-1] = initial_cfg[name]
    for name, value in initial_cfg.get("scalars", {}).items():
        fields[name] = np.zeros(shape, dtype=float)
        fields[name][1:-1, 1:-1] = value
    return fields


def _load_field_array(target, arr):
    if arr.shape == target.shape:
        target[:, :] = arr
    elif arr.shape == target[1:-1, 1:-1].shape:
        target[1:-1, 1:-1] = arr
    else:
        raise ValueError(f"Field shape {arr.shape} does not match target {target.shape}.")


def load_state(state_path, mesh, initial_cfg):
    """solver.py:401-419."""
    fields = initialize_fields(mesh, initial_cfg)
    with np.load(state_path, allow_pickle=False) as data:
        for key in data.files:
            if key in ("step", "time", "cell_center"):
                continue
            if key not in fields:
                fields[key] = np.zeros_like(fields["p"])
            _load_field_array(fields[key], data[key])
        step = int(data["step"]) if "step" in data.files else None
        time = float(data["time"]) if "time" in data.files else None
    return fields, step, time


def save_fields(mesh, fields, output_dir, step, time):
    """solver.py:1296-1310."""
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    save_path = output_dir / f"fields_{step:06d}.npz"
    payload = {"step": np.array(step, dtype=int), "time": np.array(time, dtype=float),
               "cell_center": mesh["cell_center"]}
    for key, value in fields.items():
        if key != "p_corr":
            payload[key] = value
    np.savez(save_path, **payload)
    return save_path


# ------------------------------------------------------------------ logs (formats: solver.py:1218-1293)
def log_residual(post_dir, case_name, step, time, inner_iter, residual, mom_u_l2, mom_v_l2):
    post_dir.mkdir(parents=True, exist_ok=True)
    log_path = post_dir / f"{case_name}_residual_log.txt"
    if not log_path.exists():
        log_path.write_text("step time inner_iter continuity_l2 mom_u_l2 mom_v_l2\n")
    with log_path.open("a", encoding="utf-8") as handle:
        handle.write(f"{step} {time:.6e} {inner_iter} {residual:.6e} {mom_u_l2:.6e} {mom_v_l2:.6e}\n")
    return log_path


def log_timing(post_dir, case_name, step, time, inner_iter, timing):
    post_dir.mkdir(parents=True, exist_ok=True)
    log_path = post_dir / f"{case_name}_timing_log.txt"
    if not log_path.exists():
        log_path.write_text("step time inner_iter total bc adv mom press corr cont turb post save\n")
    keys = ("total", "bc", "adv", "mom", "press", "corr", "cont", "turb", "post", "save")
    with log_path.open("a", encoding="utf-8") as handle:
        handle.write(f"{step} {time:.6e} {inner_iter} " + " ".join(f"{timing[k]:.6e}" for k in keys) + "\n")
    return log_path


def log_friction(post_dir, case_name, step, time, cf, tau_avg, bulk_velocity=None, re=None, cf_dpdx=None):
    post_dir.mkdir(parents=True, exist_ok=True)
    log_path = post_dir / f"{case_name}_friction_log.txt"
    if not log_path.exists():
        log_path.write_text("step time f_darcy f_darcy_dpdx tau_avg u_bulk reynolds\n")
    nan = float("nan")
    bulk_velocity = nan if bulk_velocity is None else bulk_velocity
    re = nan if re is None else re
    cf_dpdx = nan if cf_dpdx is None else cf_dpdx
    with log_path.open("a", encoding="utf-8") as handle:
        handle.write(f"{step} {time:.6e} {4.0 * float(cf):.6e} {4.0 * float(cf_dpdx):.6e} "
                     f"{tau_avg:.6e} {bulk_velocity:.6e} {re:.6e}\n")
    return log_path


def _ensure_single_log_file(post_dir, log_path):
    if not post_dir.exists():
        return
    others = [p for p in sorted(post_dir.glob("*_residual_log.txt")) if p.resolve() != log_path.resolve()]
    if others:
        names = ", ".join(p.name for p in others)
        raise ValueError(f"Multiple residual logs found in {post_dir}: {names}. Remove extras.")


def _truncate_log(post_dir, case_name, restart_step):
    """solver.py:1330-1354."""
    log_path = post_dir / f"{case_name}_residual_log.txt"
    if not log_path.exists():
        return
    lines = log_path.read_text().splitlines()
    if not lines:
        return
    kept = [lines[0]]
    for line in lines[1:]:
        parts = line.split()
        if not parts:
            continue
        try:
            step = int(float(parts[0]))
        except ValueError:
            continue
        if step < restart_step:
            kept.append(line)
        else:
            break
    log_path.write_text("\n".join(kept) + "\n")


def _resolve_restart_state(config, states_dir):
    restart_cfg = config.get("restart", {})
    if not restart_cfg.get("enabled", False):
        return None
    if restart_cfg.get("path"):
        return Path(restart_cfg["path"])
    if restart_cfg.get("step") is None:
        raise ValueError("restart enabled but no path or step provided")
    return Path(states_dir) / f"fields_{int(restart_cfg['step']):06d}.npz"


def _resolve_case_paths(config, mesh_path):
    paths = config.get("paths", {})
    case_dir = Path(paths.get("case_dir", mesh_path.parent.parent))
    return (case_dir, Path(paths.get("mesh_dir", case_dir / "mesh")),
            Path(paths.get("states_dir", case_dir / "states")), Path(paths.get("post_dir", case_dir / "postproc")))


# ------------------------------------------------------------------ schedules (solver.py:613-646)
def _parse_value_schedule(schedule, name):
    if schedule is None:
        return None
    items = []
    for entry in schedule:
        if len(entry) != 2:
            raise ValueError(f"{name} entries must be (value, step).")
        items.append((float(entry[0]), int(entry[1])))
    items.sort(key=lambda x: x[1])
    steps = [s for _, s in items]
    if any(s < 0 for s in steps):
        raise ValueError(f"{name} steps must be non-negative.")
    if any(b <= a for a, b in zip(steps, steps[1:])):
        raise ValueError(f"{name} steps must be strictly increasing.")
    return items


def _scheduled_value(step, base_value, schedule):
    if not schedule:
        return base_value
    if step <= schedule[0][1]:
        return schedule[0][0]
    for (val_a, step_a), (val_b, step_b) in zip(schedule[:-1], schedule[1:]):
        if step <= step_b:
            if step_b == step_a:
                return val_b
            t = (step - step_a) / float(step_b - step_a)
            return (1.0 - t) * val_a + t * val_b
    return schedule[-1][0]


# ------------------------------------------------------------------ turbulence helpers
def _turbulence_on(turb_cfg):
    return bool(turb_cfg.get("enabled", False)) and turb_cfg.get("model") not in (None, "none")


def _validate_turbulence(fields, turb_cfg, bc_cfg):
    """turbulence.initialize (turbulence.py:215-259)."""
    if not _turbulence_on(turb_cfg):
        return
    model = turb_cfg.get("model")
    if model not in ("k_omega", "sst"):
        raise NotImplementedError(f"Unknown turbulence model: {model}")
    for boundary, segments in bc_cfg.items():
        if boundary == "pressure_reference" or segments is None:
            continue
        for seg in _normalize_segments(segments):
            if seg.get("type") == "inlet" and (("k" not in seg) or ("omega" not in seg)):
                raise ValueError("Turbulence inlet requires 'k' and 'omega' values.")
    if "k" not in fields:
        raise ValueError("Initial k is missing; set it in initial scalars.")
    if "omega" not in fields:
        raise ValueError("Initial omega is missing; set it in initial scalars.")


# ------------------------------------------------------------------ engine set-up (shared with bench.py)
def default_device_ordinal(solver_cfg):
    if "device" in solver_cfg:
        return int(solver_cfg["device"])
    return int(os.environ.get("LOCAL_RANK", "0"))


def configure_device(dev, config, geometry, nu, turb_on):
    """Push physics, solver and boundary parameters of CONFIG into a Device."""
    solver_cfg = config.get("solver", {})
    momentum_cfg = solver_cfg.get("momentum", {})
    pressure_cfg = solver_cfg.get("pressure", {})
    turb_solver_cfg = solver_cfg.get("turbulence_solver", {})
    mg = solver_cfg.get("multigrid", {})
    dev.set_physics(nu, config["physics"].get("pressure_gradient", (0.0, 0.0)),
                    solver_cfg.get("rhie_chow", True), turb_on)
    dev.set_solver(momentum_cfg.get("tol", 1.0e-8), momentum_cfg.get("maxiter", 200),
                   pressure_cfg.get("tol", 1.0e-8), pressure_cfg.get("maxiter", 200),
                   turb_solver_cfg.get("tol", 1.0e-8), turb_solver_cfg.get("maxiter", 200),
                   pressure_cfg.get("freeze_operator", True))
    if mg:       # otherwise the library defaults: V(3,3) with dampings (1.25, 0.5, 1.0)
        nu_default = 3
        dev.set_multigrid(mg.get("nu_pre", nu_default), mg.get("nu_post", nu_default), mg.get("omega"),
                          mg.get("coarse_sweeps", 40), mg.get("fused", True), mg.get("omega2"), mg.get("omegas"))
    tr = solver_cfg.get("transport", {})
    dev.set_transport_solver(tr.get("warm_start", True), tr.get("chebyshev", True), tr.get("rho_max", 0.95),
                             tr.get("max_iters", 80))
    dev.set_pressure_nullspace(pressure_cfg.get("nullspace", True))
    fk = solver_cfg.get("flux_kernels", {})
    dev.set_flux_kernels(fk.get("marching", True), fk.get("lookahead", True))
    for name, value in solver_cfg.get("tuning", {}).items():       # A-B switches of the kernel forms (tests, measurements)
        dev.set_option(name, value)
    params = config.get("turbulence", {}).get("params", {})
    dev.set_turbulence_clip(params.get("k_min", 1.0e-10), params.get("omega_min", 1.0e-10))
    tables = BoundaryTables(config["boundary"], dev.ni, dev.nj_global, geometry, nu, turb_on)
    if dev.nranks > 1:
        tables = tables.slab(dev.j_offset, dev.nj)      # geometry and tables are built for the whole mesh
    dev.set_boundary(tables)
    return tables


def _distributed_world():
    """World size of an initialised torch.distributed job (1 otherwise; torch is not imported for it)."""
    import sys
    if "torch" not in sys.modules:
        return 1
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class StepControl:
    """Per-run constants of the outer loop parsed from CONFIG (solver.py:1369-1424)."""

    def __init__(self, config):
        solver_cfg = config.get("solver", {})
        time_cfg = config["time"]
        self.dt = time_cfg.get("dt")
        self.max_steps = time_cfg["max_steps"]
        self.cfl = time_cfg.get("cfl")
        self.cfl_schedule = _parse_value_schedule(time_cfg.get("cfl_schedule"), "cfl_schedule")
        self.cfl_diff = time_cfg.get("cfl_diff")
        self.cfl_diff_schedule = _parse_value_schedule(time_cfg.get("cfl_diff_schedule"), "cfl_diff_schedule")
        self.dt_max = time_cfg.get("dt_max", self.dt)
        self.dt_max_schedule = _parse_value_schedule(time_cfg.get("dt_max_schedule"), "dt_max_schedule")
        self.dt_min = time_cfg.get("dt_min", 0.0)
        if self.cfl is None and self.cfl_schedule is None and self.dt is None:
            raise ValueError("time.dt must be provided when time.cfl is not set.")
        if (self.cfl is not None or self.cfl_schedule is not None) and self.dt_max is None \
                and self.dt_max_schedule is None:
            raise ValueError("time.dt_max must be provided when time.cfl is set.")
        self.scheme = solver_cfg.get("scheme", "simple").lower()
        simple_cfg = solver_cfg.get("simple", {})
        self.simple_max_iter = simple_cfg.get("max_iter", 10)
        self.continuity_tol = simple_cfg.get("continuity_tol", 1.0e-4)
        self.relax_u = simple_cfg.get("relax_u", 0.7)
        self.relax_v = simple_cfg.get("relax_v", 0.7)
        self.relax_p = simple_cfg.get("relax_p", 0.3)
        piso_cfg = solver_cfg.get("piso", {})
        self.piso_correctors = int(piso_cfg.get("n_correctors", 3))
        self.piso_schedule = _parse_value_schedule(piso_cfg.get("corrector_schedule"), "piso.corrector_schedule")
        self.piso_relax_u = piso_cfg.get("relax_u", self.relax_u)
        self.piso_relax_v = piso_cfg.get("relax_v", self.relax_v)
        self.piso_relax_p = piso_cfg.get("relax_p", self.relax_p)
        turb_cfg = config.get("turbulence", {})
        self.turb_cfg = turb_cfg
        self.turb_on = _turbulence_on(turb_cfg)


def advance_one_step(dev, ctl, step):
    """One outer step on the device (solver.py:1452-1686 without logging / IO).
    Returns dict(dt, dt_adv, dt_diff, inner, cont, mom_u, mom_v, step_cfl, step_cfl_diff, step_dt_max)
    or None when the adaptive dt collapsed to <= 0 (solver.py:1487-1488)."""
    dev.apply_flow_bc()
    step_cfl = _scheduled_value(step, ctl.cfl, ctl.cfl_schedule)
    step_cfl_diff = _scheduled_value(step, ctl.cfl_diff, ctl.cfl_diff_schedule)
    step_dt_max = _scheduled_value(step, ctl.dt_max, ctl.dt_max_schedule)
    dt_adv = dt_diff = None
    if step_cfl is not None:
        if step_cfl_diff is None:
            step_cfl_diff = step_cfl
        dt, dt_adv, dt_diff = dev.compute_dt(step_cfl, step_cfl_diff, ctl.dt_min, step_dt_max)
        if dt <= 0.0:
            return None
    else:
        dt = float(ctl.dt)
    cont = None
    inner = 0
    mom_u = mom_v = float("nan")
    if ctl.scheme == "piso":
        n_corr = ctl.piso_correctors
        if ctl.piso_schedule is not None:
            n_corr = int(_scheduled_value(step, ctl.piso_correctors, ctl.piso_schedule))
        n_corr = max(1, n_corr)
        cont, mom_u, mom_v = dev.piso_step(dt, n_corr, ctl.piso_relax_u, ctl.piso_relax_v, ctl.piso_relax_p)
        inner = n_corr
    else:
        for inner in range(1, ctl.simple_max_iter + 1):
            cont, mom_u, mom_v = dev.simple_iter(dt, ctl.relax_u, ctl.relax_v, ctl.relax_p)
            if cont < ctl.continuity_tol:
                break
    # update_turbulence gate (solver.py:1090-1101)
    if ctl.turb_on:
        interval = int(ctl.turb_cfg.get("update_interval", 1))
        run_it = True
        if interval > 1:
            counter = ctl.turb_cfg.get("_step_counter", 0) + 1
            ctl.turb_cfg["_step_counter"] = counter
            run_it = counter % interval == 0
        if run_it:
            dev.turbulence_step(dt)
    return {"dt": dt, "dt_adv": dt_adv, "dt_diff": dt_diff, "inner": inner, "cont": cont,
            "mom_u": mom_u, "mom_v": mom_v, "step_cfl": step_cfl, "step_cfl_diff": step_cfl_diff,
            "step_dt_max": step_dt_max}


# what the last run_case call went through (tests assert on it): launches per kernel family, multigrid levels,
# solver statistics of the last step
LAST_RUN = {}


# ------------------------------------------------------------------ run_case
def run_case(config):
    """Run a case configuration through the SIMPLE / PISO loop on the GPU
    (solver.run_case, solver.py:1367-1800).  Returns (fields, step, time)."""
    solver_cfg = config.get("solver", {})
    mesh_cfg = config["mesh"]
    mesh_path = Path(mesh_cfg["path"])
    mesh = load_mesh(mesh_cfg)
    ctl = StepControl(config)
    if ctl.scheme not in ("simple", "piso"):
        ctl.scheme = "simple"        # the reference treats anything but "piso" as SIMPLE (solver.py:1494, 1590)
    nu = config["physics"]["nu"]
    post_cfg = config.get("post", {})
    save_interval = post_cfg.get("save_interval", 0)
    print_interval = post_cfg.get("print_interval", 1)
    residual_log_interval = int(post_cfg.get("residual_log_interval", 1))
    friction_log_interval = int(post_cfg.get("friction_log_interval", 1))
    timing_cfg = post_cfg.get("timing", {})
    timing_enabled = bool(timing_cfg.get("enabled", False))
    friction_cfg = post_cfg.get("friction", {})
    friction_enabled = bool(friction_cfg.get("enabled", False))

    case_name = config.get("case_name", mesh_path.stem)
    _, _, states_dir, post_dir = _resolve_case_paths(config, mesh_path)
    states_dir.mkdir(parents=True, exist_ok=True)
    post_dir.mkdir(parents=True, exist_ok=True)
    log_path = post_dir / f"{case_name}_residual_log.txt"
    writer = True                      # only rank 0 of a j-slab job touches the log files
    if _distributed_world() > 1:
        import torch.distributed as dist
        writer = dist.get_rank() == 0
    if writer:
        _ensure_single_log_file(post_dir, log_path)

    restart_path = _resolve_restart_state(config, states_dir)
    if restart_path:
        fields, step, time = load_state(restart_path, mesh, config["initial"])
        if step is None:
            step = int(config.get("restart", {}).get("step", 0) or 0)
        if time is None:
            dt_ref = ctl.dt_max if ctl.dt_max is not None else (ctl.dt or 0.0)
            time = step * dt_ref
        if writer:
            _truncate_log(post_dir, case_name, step)
    else:
        fields = initialize_fields(mesh, config["initial"])
        time = 0.0
        step = 0
        if writer and log_path.exists():
            log_path.unlink()

    turb_cfg = config.get("turbulence", {})
    turb_on = ctl.turb_on
    _validate_turbulence(fields, turb_cfg, config["boundary"])
    count_i, count_j = int(mesh["count_i"]), int(mesh["count_j"])
    if "nu_t" not in fields:
        fields["nu_t"] = np.zeros((count_i + 2, count_j + 2), dtype=float)
    elif not turb_on:
        fields["nu_t"][:, :] = 0.0

    # under an initialised torch.distributed job every rank calls run_case with the same CONFIG
    # and owns one j-slab; rank 0 writes the logs and states, every rank returns the whole fields
    if _distributed_world() > 1:
        from .slab import slab_device
        dev = slab_device(count_i, count_j, default_device_ordinal(solver_cfg),
                          min_slab_rows=solver_cfg.get("slab", {}).get("min_rows"),
                          transport=solver_cfg.get("slab", {}).get("transport"))
    else:
        dev = Device(count_i, count_j, default_device_ordinal(solver_cfg))
    root = dev.rank == 0
    try:
        dev.upload_mesh(mesh)
        configure_device(dev, config, BoundaryGeometry.from_nodes(mesh["node"]), nu, turb_on)
        if turb_on:
            dev.set_turbulence_model(turb_cfg.get("model"), mesh, config["boundary"])
        dev.upload_fields(fields)
        if timing_enabled:
            dev.enable_phase_timing(True)
        fields["p_corr"] = np.zeros((count_i + 2, count_j + 2), dtype=float)

        step_end = step + ctl.max_steps
        start_wall_time = time_module.perf_counter()
        host_fresh = False
        wall = None
        while step < step_end:
            step_start = time_module.perf_counter()
            timing = dict.fromkeys(("total", "bc", "adv", "mom", "press", "corr", "cont", "turb", "post", "save"), 0.0)
            info = advance_one_step(dev, ctl, step)
            if info is None:
                break
            host_fresh = False
            dt, inner_iter, cont_res = info["dt"], info["inner"], info["cont"]
            mom_u_l2, mom_v_l2 = info["mom_u"], info["mom_v"]
            if timing_enabled:
                ph = dev.phase_times(reset=True)
                for key in ("bc", "adv", "mom", "press", "corr", "cont", "turb"):
                    timing[key] = ph[key]

            if root and cont_res is not None and (residual_log_interval <= 1 or step % residual_log_interval == 0):
                log_residual(post_dir, case_name, step, time, inner_iter, cont_res, mom_u_l2, mom_v_l2)
            if friction_enabled and (friction_log_interval <= 1 or step % friction_log_interval == 0):
                # edge kernel + device reductions: nothing but a few scalars crosses PCIe (cfd_wall_friction)
                if wall is None:
                    wall = post_processing.WallFriction(mesh, friction_cfg.get("boundary", "j_min"),
                                                        friction_cfg.get("range"))
                friction = post_processing.friction_coefficient_device(
                    dev, wall, nu=config["physics"]["nu"], rho=friction_cfg.get("rho", 1.0),
                    u_ref=friction_cfg.get("u_ref"), use_nu_t=bool(friction_cfg.get("use_nu_t", False)),
                    flow_axis=friction_cfg.get("flow_axis", "i"),
                    hydraulic_diameter=friction_cfg.get("hydraulic_diameter"),
                    pressure_gradient=config["physics"].get("pressure_gradient"))
                if root:
                    log_friction(post_dir, case_name, step, time, friction["cf"], friction["tau_avg"],
                                 bulk_velocity=friction.get("bulk_velocity"), re=friction.get("re"),
                                 cf_dpdx=friction.get("cf_dpdx"))

            elapsed = time_module.perf_counter() - start_wall_time
            if print_interval is None:
                print_interval = 1
            if root and (print_interval == 0 or step % int(print_interval) == 0):
                adv_min, diff_min = info["dt_adv"], info["dt_diff"]
                dt_limit = "fixed"
                if adv_min is not None and diff_min is not None:
                    dt_limit = "adv" if adv_min <= diff_min else "diff"
                    if info["step_dt_max"] is not None and np.isclose(dt, float(info["step_dt_max"])):
                        dt_limit = "max"
                    if ctl.dt_min is not None and np.isclose(dt, float(ctl.dt_min)):
                        dt_limit = "min"
                resid_value = cont_res if cont_res is not None else float("nan")
                print(
                    "step {step} time {time:.6e} dt_min {dt_min:.3e} dt_max {dt_max:.3e} "
                    "dt_lim {dt_lim} wall {elapsed:.2f}s "
                    "scheme {scheme} inner {inner} cont {resid:.6e} mom_u {mom_u:.6e} mom_v {mom_v:.6e}".format(
                        step=step, time=time, dt_min=dt, dt_max=dt, dt_lim=dt_limit, elapsed=elapsed,
                        scheme=ctl.scheme, inner=inner_iter, resid=resid_value, mom_u=mom_u_l2, mom_v=mom_v_l2),
                    flush=True)
            if save_interval and step % save_interval == 0:
                t0 = time_module.perf_counter()
                if not host_fresh:
                    dev.download_fields(fields)
                    host_fresh = True
                if root:
                    save_path = save_fields(mesh, fields, states_dir, step, time)
                    print(f"saved {save_path}", flush=True)
                timing["save"] += time_module.perf_counter() - t0

            timing["total"] = time_module.perf_counter() - step_start
            if timing_enabled and root:
                log_timing(post_dir, case_name, step, time, inner_iter, timing)
            time += dt
            step += 1

        dev.download_fields(fields)
        stats = dev.stats()
        LAST_RUN.clear()
        LAST_RUN.update({"kernel_counts": dev.kernel_counts(), "stats": stats})
        if stats["p_unconverged"]:
            import warnings
            warnings.warn(f"{stats['p_unconverged']} pressure solve(s) stopped at maxiter above solver.pressure.tol "
                          "(frozen operator: continued like the reference's cached-pyamg branch, solver.py:829-833)",
                          RuntimeWarning, stacklevel=2)
    finally:
        dev.close()
    return fields, step, time
