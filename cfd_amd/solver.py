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
# File names, array keys, log headers and the %.6e number format below are the contract with the reference's
# files (solver.py:361-419 states, 1218-1293 logs, 1296-1364 save / restart); how they are produced is this
# module's own: one ghosted-array helper for every state array, one row writer for the three logs, schedules
# as sorted knot arrays.
_STATE_META = ("step", "time", "cell_center")


def _ghosted(mesh, interior=0.0):
    """A (Ni+2, Nj+2) float array, zero ghosts, interior set from a scalar or an (Ni, Nj) array."""
    arr = np.zeros((int(mesh["count_i"]) + 2, int(mesh["count_j"]) + 2), dtype=float)
    arr[1:-1, 1:-1] = interior
    return arr


def _fill_ghosted(target, values):
    """Stored arrays come ghosted or interior-only (load_state accepts both, solver.py:390-398)."""
    values = np.asarray(values)
    for view in (target, target[1:-1, 1:-1]):
        if view.shape == values.shape:
            view[...] = values
            return
    raise ValueError(f"Field shape {values.shape} does not match target {target.shape}.")


def load_mesh(mesh_cfg):
    """The mesh dict of mesh.precompute_from_nodes from its .npz, generated first when the file is absent."""
    path = Path(mesh_cfg["path"])
    ensure_single_mesh_file(path)
    if not path.is_file():
        make_mesh(mesh_cfg)
    with np.load(path, allow_pickle=False) as npz:
        return dict(npz.items())


def initialize_fields(mesh, initial_cfg):
    """u, v, p and the named scalars of CONFIG["initial"] as ghosted arrays."""
    wanted = {name: initial_cfg[name] for name in ("u", "v", "p")}
    wanted.update(initial_cfg.get("scalars", {}))
    return {name: _ghosted(mesh, value) for name, value in wanted.items()}


def load_state(state_path, mesh, initial_cfg):
    """(fields, step, time) of a saved state; arrays the initial config does not name are created on the fly."""
    fields = initialize_fields(mesh, initial_cfg)
    with np.load(state_path, allow_pickle=False) as npz:
        stamp = {key: npz[key].item() for key in ("step", "time") if key in npz.files}
        for name in npz.files:
            if name not in _STATE_META:
                _fill_ghosted(fields.setdefault(name, _ghosted(mesh)), npz[name])
    return (fields, int(stamp["step"]) if "step" in stamp else None,
            float(stamp["time"]) if "time" in stamp else None)


def save_fields(mesh, fields, output_dir, step, time):
    """states/fields_{step:06d}.npz: every entry of `fields` but p_corr, plus step, time, cell_center."""
    folder = Path(output_dir)
    folder.mkdir(parents=True, exist_ok=True)
    target = folder / f"fields_{step:06d}.npz"
    arrays = {name: arr for name, arr in fields.items() if name != "p_corr"}
    np.savez(target, step=np.array(step, dtype=int), time=np.array(time, dtype=float),
             cell_center=mesh["cell_center"], **arrays)
    return target


# ------------------------------------------------------------------ logs
_LOG_COLUMNS = {
    "residual": "step time inner_iter continuity_l2 mom_u_l2 mom_v_l2",
    "timing": "step time inner_iter total bc adv mom press corr cont turb post save",
    "friction": "step time f_darcy f_darcy_dpdx tau_avg u_bulk reynolds",
}


def _log_file(post_dir, case_name, kind):
    return Path(post_dir) / f"{case_name}_{kind}_log.txt"


def _append_row(post_dir, case_name, kind, step, time, counters, numbers):
    """One line of a log: step, time, integer columns as they are, float columns in the reference's %.6e;
    the header goes in front of the first line of a new file."""
    path = _log_file(post_dir, case_name, kind)
    path.parent.mkdir(parents=True, exist_ok=True)
    cells = [str(step), f"{time:.6e}", *(str(v) for v in counters), *(f"{float(v):.6e}" for v in numbers)]
    text = " ".join(cells) + "\n"
    if not path.exists():
        text = _LOG_COLUMNS[kind] + "\n" + text
    with path.open("a", encoding="utf-8") as handle:
        handle.write(text)
    return path


def log_residual(post_dir, case_name, step, time, inner_iter, residual, mom_u_l2, mom_v_l2):
    return _append_row(post_dir, case_name, "residual", step, time, (inner_iter,), (residual, mom_u_l2, mom_v_l2))


def log_timing(post_dir, case_name, step, time, inner_iter, timing):
    phases = _LOG_COLUMNS["timing"].split()[3:]
    return _append_row(post_dir, case_name, "timing", step, time, (inner_iter,), [timing[name] for name in phases])


def log_friction(post_dir, case_name, step, time, cf, tau_avg, bulk_velocity=None, re=None, cf_dpdx=None):
    """Darcy friction factors are four times the Fanning coefficients the diagnostic returns."""
    absent = float("nan")
    optional = [absent if v is None else v for v in (bulk_velocity, re)]
    darcy_dpdx = absent if cf_dpdx is None else 4.0 * float(cf_dpdx)
    return _append_row(post_dir, case_name, "friction", step, time, (), (4.0 * float(cf), darcy_dpdx, tau_avg, *optional))


def _only_residual_log(post_dir, log_path):
    """A postproc directory holds the residual log of ONE case (the plotting code globs for it)."""
    post_dir = Path(post_dir)
    if not post_dir.is_dir():
        return
    mine = Path(log_path).resolve()
    strays = sorted(p.name for p in post_dir.glob("*_residual_log.txt") if p.resolve() != mine)
    if strays:
        raise ValueError(f"Multiple residual logs found in {post_dir}: {', '.join(strays)}. Remove extras.")


def _rows_before(rows, first_dropped_step):
    """The leading rows whose step number lies below the restart step; rows without a number are skipped."""
    for row in rows:
        try:
            number = int(float(row.split(None, 1)[0]))
        except (IndexError, ValueError):
            continue
        if number >= first_dropped_step:
            return
        yield row


def _cut_log_at(post_dir, case_name, restart_step):
    """A restart continues the residual log from `restart_step`: what was logged from there on goes."""
    path = _log_file(post_dir, case_name, "residual")
    rows = path.read_text().splitlines() if path.exists() else []
    if rows:
        path.write_text("\n".join([rows[0], *_rows_before(rows[1:], restart_step)]) + "\n")


def _restart_source(config, states_dir):
    """Path of the state a restart starts from: restart.path, or the fields file of restart.step; None: fresh run."""
    restart = config.get("restart") or {}
    if not restart.get("enabled", False):
        return None
    explicit, from_step = restart.get("path"), restart.get("step")
    if explicit:
        return Path(explicit)
    if from_step is None:
        raise ValueError("restart enabled but no path or step provided")
    return Path(states_dir) / f"fields_{int(from_step):06d}.npz"


def _case_directories(config, mesh_path):
    """(case, mesh, states, postproc) directories; the case directory defaults to two levels above the mesh file."""
    given = config.get("paths", {})
    case_dir = Path(given.get("case_dir", Path(mesh_path).parent.parent))
    sub = {"mesh_dir": "mesh", "states_dir": "states", "post_dir": "postproc"}
    return (case_dir, *(Path(given.get(key, case_dir / name)) for key, name in sub.items()))


# ------------------------------------------------------------------ schedules
class Schedule:
    """(value, step) knots of a CONFIG schedule (cfl_schedule, dt_max_schedule, piso.corrector_schedule, ...):
    piecewise linear in the step number, held at the end values outside the knots.  Compares equal to the sorted
    list of (value, step) pairs."""

    def __init__(self, entries, name):
        pairs = []
        for entry in entries:
            if len(entry) != 2:
                raise ValueError(f"{name} entries must be (value, step).")
            pairs.append((int(entry[1]), float(entry[0])))
        pairs.sort()
        self.steps = np.array([p[0] for p in pairs], dtype=np.int64)
        self.values = np.array([p[1] for p in pairs], dtype=float)
        if (self.steps < 0).any():
            raise ValueError(f"{name} steps must be non-negative.")
        if (np.diff(self.steps) <= 0).any():
            raise ValueError(f"{name} steps must be strictly increasing.")

    def __call__(self, step):
        hi = int(np.searchsorted(self.steps, step, side="left"))       # first knot at or after `step`
        if hi == 0 or hi == self.steps.size:
            return float(self.values[min(hi, self.steps.size - 1)])
        lo = hi - 1
        t = (step - int(self.steps[lo])) / float(int(self.steps[hi]) - int(self.steps[lo]))
        # the reference's blend, term by term, so that a scheduled cfl has the reference's bits (solver.py:632-646)
        return (1.0 - t) * float(self.values[lo]) + t * float(self.values[hi])

    def pairs(self):
        return [(float(v), int(s)) for s, v in zip(self.steps, self.values)]

    def __eq__(self, other):
        return self.pairs() == (other.pairs() if isinstance(other, Schedule) else other)

    def __len__(self):
        return int(self.steps.size)


def _schedule_of(schedule, name):
    return None if schedule is None else Schedule(schedule, name)


def _value_at(step, base_value, schedule):
    return schedule(step) if schedule is not None and len(schedule) else base_value


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
        self.cfl_schedule = _schedule_of(time_cfg.get("cfl_schedule"), "cfl_schedule")
        self.cfl_diff = time_cfg.get("cfl_diff")
        self.cfl_diff_schedule = _schedule_of(time_cfg.get("cfl_diff_schedule"), "cfl_diff_schedule")
        self.dt_max = time_cfg.get("dt_max", self.dt)
        self.dt_max_schedule = _schedule_of(time_cfg.get("dt_max_schedule"), "dt_max_schedule")
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
        self.piso_schedule = _schedule_of(piso_cfg.get("corrector_schedule"), "piso.corrector_schedule")
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
    step_cfl = _value_at(step, ctl.cfl, ctl.cfl_schedule)
    step_cfl_diff = _value_at(step, ctl.cfl_diff, ctl.cfl_diff_schedule)
    step_dt_max = _value_at(step, ctl.dt_max, ctl.dt_max_schedule)
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
            n_corr = int(_value_at(step, ctl.piso_correctors, ctl.piso_schedule))
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
_CONSOLE_LINE = ("step {step} time {time:.6e} dt_min {dt:.3e} dt_max {dt:.3e} dt_lim {limit} wall {wall:.2f}s "
                 "scheme {scheme} inner {inner} cont {cont:.6e} mom_u {mom_u:.6e} mom_v {mom_v:.6e}")      # solver.py:1752-1770
_TIMED_PHASES = ("bc", "adv", "mom", "press", "corr", "cont", "turb")


class OutputControl:
    """What a run writes and how often: CONFIG["post"], CONFIG["paths"], CONFIG["case_name"] (solver.py:1416-1431)."""

    def __init__(self, config, mesh_path):
        post = config.get("post", {})
        self.save_every = post.get("save_interval", 0)
        shown = post.get("print_interval", 1)
        self.print_every = 1 if shown is None else int(shown)
        self.residual_every = int(post.get("residual_log_interval", 1))
        self.friction_every = int(post.get("friction_log_interval", 1))
        self.timing = bool(post.get("timing", {}).get("enabled", False))
        self.friction = post.get("friction", {})
        self.friction_on = bool(self.friction.get("enabled", False))
        self.case = config.get("case_name", Path(mesh_path).stem)
        _, _, self.states_dir, self.post_dir = _case_directories(config, mesh_path)
        self.residual_log = _log_file(self.post_dir, self.case, "residual")

    @staticmethod
    def due(step, every):
        """Log cadence: every step for an interval of 0 or 1, else the steps the interval divides."""
        return every <= 1 or step % every == 0

    def prints(self, step):
        return self.print_every == 0 or step % self.print_every == 0

    def saves(self, step):
        return bool(self.save_every) and step % self.save_every == 0


def _dt_limit_label(info, dt, dt_min):
    """Which bound set the step's dt, for the console line: adv | diff | max | min | fixed."""
    adv, diff = info["dt_adv"], info["dt_diff"]
    if adv is None or diff is None:
        return "fixed"
    label = "adv" if adv <= diff else "diff"
    if info["step_dt_max"] is not None and np.isclose(dt, float(info["step_dt_max"])):
        label = "max"
    if dt_min is not None and np.isclose(dt, float(dt_min)):
        label = "min"
    return label


def _starting_state(config, mesh, ctl, out, writer):
    """(fields, step, time) a run starts from: a saved state (the residual log is cut back to it) or CONFIG["initial"]
    (an older residual log of the case is removed).  solver.py:1433-1447."""
    source = _restart_source(config, out.states_dir)
    if source is None:
        if writer and out.residual_log.exists():
            out.residual_log.unlink()
        return initialize_fields(mesh, config["initial"]), 0, 0.0
    fields, step, time = load_state(source, mesh, config["initial"])
    if step is None:
        step = int((config.get("restart") or {}).get("step", 0) or 0)
    if time is None:
        per_step = ctl.dt_max if ctl.dt_max is not None else (ctl.dt or 0.0)
        time = step * per_step
    if writer:
        _cut_log_at(out.post_dir, out.case, step)
    return fields, step, time


def run_case(config):
    """Run a case configuration through the SIMPLE / PISO loop on the GPU
    (solver.run_case, solver.py:1367-1800).  Returns (fields, step, time)."""
    solver_cfg = config.get("solver", {})
    mesh = load_mesh(config["mesh"])
    ctl = StepControl(config)
    if ctl.scheme not in ("simple", "piso"):
        ctl.scheme = "simple"        # the reference treats anything but "piso" as SIMPLE (solver.py:1494, 1590)
    nu = config["physics"]["nu"]
    out = OutputControl(config, config["mesh"]["path"])
    for folder in (out.states_dir, out.post_dir):
        folder.mkdir(parents=True, exist_ok=True)
    slabs = _distributed_world() > 1
    writer = True                      # only rank 0 of a j-slab job touches the log files
    if slabs:
        import torch.distributed as dist
        writer = dist.get_rank() == 0
    if writer:
        _only_residual_log(out.post_dir, out.residual_log)
    fields, step, time = _starting_state(config, mesh, ctl, out, writer)

    turb_cfg = config.get("turbulence", {})
    turb_on = ctl.turb_on
    _validate_turbulence(fields, turb_cfg, config["boundary"])
    if "nu_t" not in fields:
        fields["nu_t"] = _ghosted(mesh)
    elif not turb_on:
        fields["nu_t"][...] = 0.0
    count_i, count_j = fields["u"].shape[0] - 2, fields["u"].shape[1] - 2

    # under an initialised torch.distributed job every rank calls run_case with the same CONFIG
    # and owns one j-slab; rank 0 writes the logs and states, every rank returns the whole fields
    if slabs:
        from .slab import slab_device
        slab_cfg = solver_cfg.get("slab", {})
        dev = slab_device(count_i, count_j, default_device_ordinal(solver_cfg), min_slab_rows=slab_cfg.get("min_rows"),
                          transport=slab_cfg.get("transport"))
    else:
        dev = Device(count_i, count_j, default_device_ordinal(solver_cfg))
    root = dev.rank == 0
    try:
        dev.upload_mesh(mesh)
        configure_device(dev, config, BoundaryGeometry.from_nodes(mesh["node"]), nu, turb_on)
        if turb_on:
            dev.set_turbulence_model(turb_cfg.get("model"), mesh, config["boundary"])
        dev.upload_fields(fields)
        dev.enable_phase_timing(out.timing)
        fields["p_corr"] = _ghosted(mesh)

        last_step = step + ctl.max_steps
        run_started = time_module.perf_counter()
        host_fresh = False
        wall = None
        while step < last_step:
            step_started = time_module.perf_counter()
            info = advance_one_step(dev, ctl, step)
            if info is None:
                break
            host_fresh = False
            dt, cont = info["dt"], info["cont"]
            spent = dict.fromkeys(_LOG_COLUMNS["timing"].split()[3:], 0.0)
            if out.timing:
                device_times = dev.phase_times(reset=True)
                spent.update({name: device_times[name] for name in _TIMED_PHASES})

            if root and cont is not None and out.due(step, out.residual_every):
                log_residual(out.post_dir, out.case, step, time, info["inner"], cont, info["mom_u"], info["mom_v"])
            if out.friction_on and out.due(step, out.friction_every):
                # edge kernel + device reductions: nothing but a few scalars crosses PCIe (cfd_wall_friction)
                fr = out.friction
                if wall is None:
                    wall = post_processing.WallFriction(mesh, fr.get("boundary", "j_min"), fr.get("range"))
                friction = post_processing.friction_coefficient_device(
                    dev, wall, nu=nu, rho=fr.get("rho", 1.0), u_ref=fr.get("u_ref"), use_nu_t=bool(fr.get("use_nu_t", False)),
                    flow_axis=fr.get("flow_axis", "i"), hydraulic_diameter=fr.get("hydraulic_diameter"),
                    pressure_gradient=config["physics"].get("pressure_gradient"))
                if root:
                    log_friction(out.post_dir, out.case, step, time, friction["cf"], friction["tau_avg"],
                                 bulk_velocity=friction.get("bulk_velocity"), re=friction.get("re"),
                                 cf_dpdx=friction.get("cf_dpdx"))

            if root and out.prints(step):
                print(_CONSOLE_LINE.format(step=step, time=time, dt=dt, limit=_dt_limit_label(info, dt, ctl.dt_min),
                                           wall=time_module.perf_counter() - run_started, scheme=ctl.scheme,
                                           inner=info["inner"], cont=float("nan") if cont is None else cont,
                                           mom_u=info["mom_u"], mom_v=info["mom_v"]), flush=True)
            if out.saves(step):
                save_started = time_module.perf_counter()
                if not host_fresh:
                    dev.download_fields(fields)
                    host_fresh = True
                if root:
                    print(f"saved {save_fields(mesh, fields, out.states_dir, step, time)}", flush=True)
                spent["save"] += time_module.perf_counter() - save_started

            spent["total"] = time_module.perf_counter() - step_started
            if out.timing and root:
                log_timing(out.post_dir, out.case, step, time, info["inner"], spent)
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
        if stats["transport_stalled"]:
            import warnings
            warnings.warn(f"{stats['transport_stalled']} transport solve(s) were asked for a tolerance below 1e-12 and "
                          "stalled between 1e-12 and 1e-6 after two restarts: accepted at the accuracy reached "
                          "(include/cfd_hip.h: transport_stalled)", RuntimeWarning, stacklevel=2)
    finally:
        dev.close()
    return fields, step, time
