"""Per-phase drop-ins for the reference's function-level API (README.md:196-205): the same names, argument
meaning, in-place effects on the caller's `fields` dict and return values as solver.py / turbulence.py of the
reference, each executed by the HIP kernels of libcfd_hip.so.

A call uploads the arrays it needs, runs its kernels and downloads what the reference function writes, so
these are for callers that drive the solver phase by phase (and for parity tests); `solver.run_case` does not
go through them -- its loop keeps every plane in HBM between phases.

One engine (GPU context + uploaded mesh) is kept per mesh dict, identified by the identity of its
`cell_volume` array: a mesh whose arrays are modified in place must be passed as a new dict.
There is no CPU path: without the HIP extension or a GPU every function raises.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .bc import BoundaryGeometry, BoundaryTables
from .device import Device

_WALLS = {side: {"type": "wall"} for side in ("i_min", "i_max", "j_min", "j_max")}
_MAX_ENGINES = 2
_engines = {}           # id(mesh["cell_volume"]) -> Engine, most recently used last
_turb_state = {"model": "k_omega"}      # the closure the last initialize() call validated (turbulence.py:10-13, 257)


def _turbulence_on(cfg):
    return bool(cfg) and bool(cfg.get("enabled", False)) and cfg.get("model") not in (None, "none")


class Engine:
    """GPU context of one mesh for the function-level API."""

    def __init__(self, mesh):
        self.mesh = mesh
        self.anchor = mesh["cell_volume"]           # keeps the id() of the key alive
        self.ni, self.nj = int(mesh["count_i"]), int(mesh["count_j"])
        self.dev = Device(self.ni, self.nj)
        self.dev.upload_mesh(mesh)
        self.dev.set_flux_kernels(True, False)      # nothing is carried from one call to the next
        self.geometry = BoundaryGeometry.from_nodes(mesh["node"]) if "node" in mesh else None
        self.bc_key = None
        self.model = None
        self.zeros = np.zeros((self.ni + 2, self.nj + 2))
        self.nut_zero = False                       # the nu_t plane in HBM is known to be all zeros
        self.boundary(_WALLS, 0.0, False)

    def close(self):
        self.dev.close()

    # ------------------------------------------------------------ parameters
    def boundary(self, bc_cfg, nu=0.0, turb_on=False):
        """Boundary tables of bc_cfg (kept while the same configuration comes again)."""
        if bc_cfg is None:
            return
        key = (repr(bc_cfg), float(nu) if turb_on else None, bool(turb_on))
        if key == self.bc_key:
            return
        if self.geometry is None:
            raise ValueError("mesh['node'] is needed to build the boundary tables")
        self.dev.set_boundary(BoundaryTables(bc_cfg, self.ni, self.nj, self.geometry, nu, turb_on))
        self.bc_key = key

    def viscosity(self, nu, rhie_chow=True, pressure_gradient=(0.0, 0.0)):
        """nu: scalar, or a (Ni, Nj) array of cell values (nu + nu_t of the caller, solver.py:147-230)."""
        dev = self.dev
        if np.isscalar(nu) or np.ndim(nu) == 0:
            dev.set_physics(float(nu), pressure_gradient, rhie_chow, False)
            dev.set_option("scalar_nu", 1)
            if not self.nut_zero:                   # the time-step kernel adds the nu_t plane to the scalar
                dev.upload(_lib.F_NUT, self.zeros)
                self.nut_zero = True
            return
        arr = np.asarray(nu, dtype=float)
        if arr.shape != (self.ni, self.nj):
            raise ValueError(f"nu must be a scalar or a {(self.ni, self.nj)} array, got {arr.shape}")
        plane = self.zeros.copy()
        plane[1:-1, 1:-1] = arr
        dev.set_physics(0.0, pressure_gradient, rhie_chow, False)        # 0.0 + x is x, bit for bit
        dev.set_option("scalar_nu", 0)
        dev.upload(_lib.F_NUT, plane)
        self.nut_zero = False

    def load(self, fields, names=("u", "v", "p"), p_field=None):
        for name in names:
            arr = p_field if (name == "p" and p_field is not None) else fields[name]
            self.dev.upload(_lib.FIELD_IDS[name], arr)

    def turbulence_model(self, cfg, bc_cfg):
        model = cfg.get("model", "k_omega")
        if model not in ("k_omega", "sst"):
            raise NotImplementedError(f"Unknown turbulence model: {model}")
        key = (model, repr(bc_cfg) if model == "sst" else None)
        if key != self.model:
            self.dev.set_turbulence_model(model, self.mesh, bc_cfg)
            self.model = key
        return model


def engine(mesh):
    key = id(mesh["cell_volume"])
    eng = _engines.pop(key, None)
    if eng is None or eng.anchor is not mesh["cell_volume"]:
        if eng is not None:
            eng.close()
        while len(_engines) >= _MAX_ENGINES:
            _engines.pop(next(iter(_engines))).close()
        eng = Engine(mesh)
    _engines[key] = eng
    return eng


def release():
    """Free every cached GPU context."""
    while _engines:
        _engines.popitem()[1].close()


# ---------------------------------------------------------------- solver.py phases
def apply_boundary_conditions(fields, mesh, bc_cfg):
    """solver.apply_boundary_conditions -> bc.apply_boundary_conditions (solver.py:422-424, bc.py:259-347):
    ghost cells of u, v, p from the boundary configuration; periodic pairs wrap every field of the dict."""
    eng = engine(mesh)
    dev = eng.dev
    eng.boundary(bc_cfg)
    eng.load(fields, ("u", "v", "p"))
    shape = (eng.ni + 2, eng.nj + 2)
    scalars = [n for n, a in fields.items() if n not in ("u", "v", "p") and getattr(a, "shape", None) == shape]
    slots = (_lib.F_K, _lib.F_OMEGA, _lib.F_NUT)           # the planes the periodic copy carries along
    dev.set_physics(0.0, (0.0, 0.0), True, True)
    eng.nut_zero = False
    try:
        for first in range(0, max(len(scalars), 1), len(slots)):
            batch = scalars[first:first + len(slots)]
            for fid, name in zip(slots, batch):
                dev.upload(fid, fields[name])
            dev.apply_flow_bc()                             # ghosts are functions of interior cells: repeating is harmless
            for fid, name in zip(slots, batch):
                dev.download(fid, out=fields[name])
    finally:
        dev.set_physics(0.0, (0.0, 0.0), True, False)
    for name in ("u", "v", "p"):
        dev.download(_lib.FIELD_IDS[name], out=fields[name])


def compute_adaptive_dt(fields, mesh, cfl, dt_max, dt_min=0.0, cfl_diff=None, nu_eff=None):
    """solver.compute_adaptive_dt (solver.py:547-582): global time step from the advective and diffusive limits."""
    eng = engine(mesh)
    eng.viscosity(0.0 if nu_eff is None else nu_eff)
    eng.load(fields, ("u", "v"))
    dt, _, _ = eng.dev.compute_dt(cfl, cfl_diff, dt_min, dt_max)
    return dt


def dt_limit_estimate(fields, mesh, cfl, cfl_diff, nu_eff):
    """solver._dt_limit_estimate (solver.py:585-610): (dt_adv_min, dt_diff_min)."""
    eng = engine(mesh)
    eng.viscosity(0.0 if nu_eff is None else nu_eff)
    eng.load(fields, ("u", "v"))
    _, dt_adv, dt_diff = eng.dev.compute_dt(cfl, cfl_diff, None, None)
    return dt_adv, dt_diff


def momentum_diagonal(mesh, nu, dt):
    """solver._momentum_diagonal (solver.py:439-452) for a scalar dt: the (Ni, Nj) diagonal a_P."""
    eng = engine(mesh)
    eng.viscosity(nu)
    eng.dev.momentum_diagonal(float(dt), 1.0)
    return eng.dev.download(_lib.F_AP, _lib.KIND_CELL)


def face_mass_flux(fields, mesh, a_p, use_rhie_chow=True, p_field=None):
    """solver._face_mass_flux (solver.py:455-513) with the caller's a_P."""
    eng = engine(mesh)
    eng.load(fields, ("u", "v", "p"), p_field)
    eng.dev.upload(_lib.F_AP, np.asarray(a_p, dtype=float), _lib.KIND_CELL)
    return eng.dev.face_flux(rhie_chow=use_rhie_chow)


def advect_scalar(phi, mesh, flux_i, flux_j):
    """solver.advect_scalar (solver.py:649-669): first-order upwind advection term, (Ni, Nj)."""
    eng = engine(mesh)
    eng.dev.upload(_lib.F_K, np.asarray(phi, dtype=float))
    return eng.dev.advect(_lib.F_K, flux_i, flux_j)


def grad_scalar(phi, mesh):
    """solver.grad_scalar (solver.py:322-358): Green-Gauss cell gradient, (Ni, Nj, 2)."""
    eng = engine(mesh)
    eng.dev.upload(_lib.F_K, np.asarray(phi, dtype=float))
    return eng.dev.grad(_lib.F_K)


def compute_advective_rhs(fields, mesh, nu, dt, use_rhie_chow=True, p_field=None):
    """solver.compute_advective_rhs (solver.py:672-691): {"u", "v"} ghost-shaped advective terms (zero ghosts)
    and the face mass fluxes {"flux_i", "flux_j"} they were formed with."""
    eng = engine(mesh)
    eng.viscosity(nu, rhie_chow=use_rhie_chow)
    eng.load(fields, ("u", "v", "p"), p_field)
    eng.dev.momentum_diagonal(float(dt), 1.0)
    flux_i, flux_j = eng.dev.face_flux(rhie_chow=use_rhie_chow)
    rhs = {}
    for name in ("u", "v"):
        out = np.zeros((eng.ni + 2, eng.nj + 2), dtype=float)
        out[1:-1, 1:-1] = eng.dev.advect(_lib.FIELD_IDS[name], flux_i, flux_j)
        rhs[name] = out
    rhs["flux_i"], rhs["flux_j"] = flux_i, flux_j
    return rhs


def solve_momentum(fields, mesh, dt, nu, adv_rhs, bc_cfg, solver_cfg=None, p_field=None, pressure_gradient=(0.0, 0.0),
                   extra_pressure_field=None):
    """solver.solve_momentum (solver.py:848-901): implicit predictor for u and v from the caller's advective terms;
    writes the interiors of fields["u"], fields["v"]; returns the diagonals and the RMS residuals."""
    solver_cfg = solver_cfg or {}
    eng = engine(mesh)
    eng.boundary(bc_cfg)
    eng.viscosity(nu, pressure_gradient=pressure_gradient)
    _set_tolerances(eng, mom=(solver_cfg.get("tol", 1.0e-8), solver_cfg.get("maxiter", 200)))
    eng.load(fields, ("u", "v", "p"), p_field)
    eng.dev.momentum_diagonal(float(dt), 1.0)
    eng.dev.momentum_rhs_from(dt, adv_rhs["u"][1:-1, 1:-1], adv_rhs["v"][1:-1, 1:-1], extra_pressure_field, pressure_gradient)
    mom_u, mom_v = eng.dev.momentum_solve()
    for name in ("u", "v"):
        fields[name][1:-1, 1:-1] = eng.dev.download(_lib.FIELD_IDS[name])[1:-1, 1:-1]
    diag = eng.dev.download(_lib.F_AP, _lib.KIND_CELL)
    return {"a_p_u": diag, "a_p_v": diag.copy(), "mom_u_l2": mom_u, "mom_v_l2": mom_v}


def solve_pressure(fields, mesh, dt, nu, bc_cfg, solver_cfg=None, use_rhie_chow=True, p_field=None, a_p_override=None,
                   amg_cache=None):
    """solver.solve_pressure (solver.py:904-1053): pressure-correction Poisson solve; returns fields["p_corr"].
    amg_cache: like the reference's pyamg cache (solver.py:821-833) a dict that, once filled by a call, keeps the
    first operator (matrix and multigrid hierarchy) for later calls with a system of the same shape."""
    solver_cfg = solver_cfg or {}
    eng = engine(mesh)
    eng.boundary(bc_cfg)
    eng.viscosity(nu, rhie_chow=use_rhie_chow)
    shape = (eng.ni * eng.nj, eng.ni * eng.nj)
    frozen = amg_cache is not None and amg_cache.get("ml") is eng and amg_cache.get("shape") == shape
    _set_tolerances(eng, p=(solver_cfg.get("tol", 1.0e-8), solver_cfg.get("maxiter", 200)), freeze=frozen)
    eng.load(fields, ("u", "v", "p"), p_field)
    if a_p_override is not None:
        eng.dev.upload(_lib.F_AP, np.asarray(a_p_override, dtype=float), _lib.KIND_CELL)
    else:
        eng.dev.momentum_diagonal(float(dt), 1.0)
    eng.dev.pressure_solve()
    if amg_cache is not None and not frozen:
        amg_cache["ml"], amg_cache["shape"] = eng, shape
    if "p_corr" not in fields:
        fields["p_corr"] = np.zeros((eng.ni + 2, eng.nj + 2), dtype=float)
    fields["p_corr"][1:-1, 1:-1] = eng.dev.download(_lib.F_PCORR)[1:-1, 1:-1]
    return fields["p_corr"]


def correct_velocity(fields, mesh, bc_cfg, a_p_u, a_p_v, u_prev, v_prev, p_prev, relax_u, relax_v, relax_p):
    """solver.correct_velocity (solver.py:1056-1087): ghost cells of p_corr, then the under-relaxed velocity and
    pressure update of the interiors of fields["u"], ["v"], ["p"]."""
    if "p_corr" not in fields:
        raise RuntimeError("Pressure correction field is missing.")
    a_p_u = np.asarray(a_p_u, dtype=float)
    if a_p_v is not a_p_u and not np.array_equal(a_p_u, a_p_v):
        raise NotImplementedError("the GPU path keeps one momentum diagonal: a_p_u and a_p_v must be equal "
                                  "(they are for every system solver.solve_momentum assembles, solver.py:878-883)")
    eng = engine(mesh)
    eng.boundary(bc_cfg)
    dev = eng.dev
    eng.load(fields, ("u", "v", "p"))
    dev.upload(_lib.F_PCORR, fields["p_corr"])
    dev.upload(_lib.F_AP, a_p_u, _lib.KIND_CELL)
    for fid, arr in ((_lib.F_U_PREV, u_prev), (_lib.F_V_PREV, v_prev), (_lib.F_P_PREV, p_prev)):
        dev.upload(fid, np.asarray(arr, dtype=float), _lib.KIND_CELL)
    dev.correct(relax_u, relax_v, relax_p)
    for name in ("u", "v", "p"):
        fields[name][1:-1, 1:-1] = dev.download(_lib.FIELD_IDS[name])[1:-1, 1:-1]
    pc = dev.download(_lib.F_PCORR)
    for a, b in ((0, 0), (0, -1), (-1, 0), (-1, -1)):        # the reference never writes the corner ghosts
        pc[a, b] = fields["p_corr"][a, b]
    fields["p_corr"][:, :] = pc


def continuity_residual(fields, mesh, nu, dt, use_rhie_chow=True, p_field=None):
    """solver.continuity_residual (solver.py:1205-1215): sqrt(mean((div F / V)^2)) of the face mass fluxes."""
    eng = engine(mesh)
    eng.viscosity(nu, rhie_chow=use_rhie_chow)
    eng.load(fields, ("u", "v", "p"), p_field)
    eng.dev.momentum_diagonal(float(dt), 1.0)
    return eng.dev.continuity()


def update_turbulence(fields, mesh, turb_cfg, bc_cfg, dt, nu, use_rhie_chow=True):
    """solver.update_turbulence (solver.py:1090-1153): one implicit k / omega transport step (k-omega or SST) and
    the turbulence boundary conditions; k and omega are REPLACED in the dict by new arrays like the reference's
    np.maximum clip does (solver.py:1150-1151), nu_t and the SST caches are written in place / added."""
    if not _turbulence_on(turb_cfg):
        initialize(fields, mesh, turb_cfg)
        return
    interval = int(turb_cfg.get("update_interval", 1))
    if interval > 1:
        counter = turb_cfg.get("_step_counter", 0) + 1
        turb_cfg["_step_counter"] = counter
        if counter % interval != 0:
            return
    model = turb_cfg.get("model")
    if model not in ("k_omega", "sst"):
        raise NotImplementedError(f"Unknown turbulence model: {model}")
    initialize(fields, mesh, turb_cfg)
    eng = engine(mesh)
    dev = eng.dev
    eng.boundary(bc_cfg, nu, True)
    dev.set_physics(float(nu), (0.0, 0.0), use_rhie_chow, True)
    dev.set_option("scalar_nu", 0)
    eng.turbulence_model(turb_cfg, bc_cfg)
    params = turb_cfg.get("params", {})
    dev.set_turbulence_clip(params.get("k_min", 1.0e-10), params.get("omega_min", 1.0e-10))
    ts = turb_cfg.get("solver", {})
    _set_tolerances(eng, turb=(ts.get("tol", 1.0e-8), ts.get("maxiter", 200)))
    if model == "sst":        # the caches travel inside `fields` (turbulence.py:473-521): none there, none on the device
        _lib.check(dev.lib.cfd_sst_set_cached(dev._h, 0))
    dev.upload_fields(fields)
    eng.nut_zero = False
    try:
        dev.turbulence_step(float(dt))
        for name in ("k", "omega"):
            fields[name] = dev.download(_lib.FIELD_IDS[name])
        dev.download(_lib.F_NUT, out=fields["nu_t"])
        if model == "sst" and dev.sst_cached():
            for name, fid in _lib.SST_FIELD_IDS.items():
                fields[name] = dev.download(fid)
    finally:
        dev.set_physics(float(nu), (0.0, 0.0), use_rhie_chow, False)


# ---------------------------------------------------------------- turbulence.py
def initialize(fields, mesh, cfg, bc_cfg=None):
    """turbulence.initialize (turbulence.py:215-259): validation and the nu_t field (host bookkeeping only)."""
    if not _turbulence_on(cfg):
        if "nu_t" not in fields:
            fields["nu_t"] = np.zeros((int(mesh["count_i"]) + 2, int(mesh["count_j"]) + 2), dtype=float)
        else:
            fields["nu_t"][:, :] = 0.0
        return fields
    if cfg.get("model") not in ("k_omega", "sst"):
        raise NotImplementedError(f"Unknown turbulence model: {cfg.get('model')}")
    if bc_cfg is not None:
        for boundary, segments in bc_cfg.items():
            if boundary == "pressure_reference" or segments is None:
                continue
            for seg in ([segments] if isinstance(segments, dict) else segments):
                if seg.get("type") == "inlet" and (("k" not in seg) or ("omega" not in seg)):
                    raise ValueError("Turbulence inlet requires 'k' and 'omega' values.")
    if "k" not in fields:
        raise ValueError("Initial k is missing; set it in initial scalars.")
    if "omega" not in fields:
        raise ValueError("Initial omega is missing; set it in initial scalars.")
    if "nu_t" not in fields:
        fields["nu_t"] = np.zeros((int(mesh["count_i"]) + 2, int(mesh["count_j"]) + 2), dtype=float)
    _turb_state["model"] = cfg.get("model")          # what sources() / effective_diffusivity() go by (turbulence.py:257)
    return fields


def apply_bcs(fields, mesh, bc_cfg, cfg, nu):
    """turbulence.apply_bcs (turbulence.py:291-341): periodic wraps, inlet / wall / outlet / symmetry values of k
    and omega (the wall rule also sets the first interior cells).  k-omega closure; the SST extras of
    turbulence.py:343-355 (wall distance, first F2) are produced by update_turbulence."""
    if not _turbulence_on(cfg):
        return
    eng = engine(mesh)
    eng.boundary(bc_cfg, nu, True)
    for name in ("k", "omega"):
        eng.dev.upload(_lib.FIELD_IDS[name], fields[name])
    eng.dev.apply_turb_bc()
    for name in ("k", "omega"):
        eng.dev.download(_lib.FIELD_IDS[name], out=fields[name])


def eddy_viscosity(fields, cfg):
    """turbulence.eddy_viscosity (turbulence.py:362-394): nu_t from k and omega, written into fields["nu_t"]:
    k / omega (k-omega), a1 k / max(a1 omega, S F2) with the caches _S, _F2 of a transport step, else a1 k / omega
    (SST).  The function has no mesh argument: it runs on the GPU context of the last mesh of this field shape."""
    if not _turbulence_on(cfg):
        fields["nu_t"][:, :] = 0.0
        return fields["nu_t"]
    model = cfg.get("model", "k_omega")
    eng = _engine_for_shape(fields["k"].shape)
    dev = eng.dev
    for name in ("k", "omega"):
        dev.upload(_lib.FIELD_IDS[name], fields[name])
    eng.model = None                                         # update_turbulence sets its closure up again
    _lib.check(dev.lib.cfd_set_turbulence_model(dev._h, _lib.TURB_SST if model == "sst" else _lib.TURB_KOMEGA))
    if model == "sst":
        limiter = fields.get("_F2") is not None and fields.get("_S") is not None
        if limiter:
            dev.upload(_lib.F_SST_S, fields["_S"])
            dev.upload(_lib.F_SST_F2, fields["_F2"])
        _lib.check(dev.lib.cfd_sst_set_cached(dev._h, int(limiter)))
    dev.set_physics(0.0, (0.0, 0.0), True, True)
    try:
        dev.eddy_viscosity()
    finally:
        dev.set_physics(0.0, (0.0, 0.0), True, False)
    eng.nut_zero = False
    dev.download(_lib.F_NUT, out=fields["nu_t"])
    return fields["nu_t"]


# ---------------------------------------------------------------- helpers
class _PlanesOnly:
    """GPU context of a field SHAPE, without a mesh: what the pointwise closures (eddy_viscosity, sources,
    effective_diffusivity -- none of them has a mesh argument in the reference) need."""

    def __init__(self, ni, nj):
        self.ni, self.nj = int(ni), int(nj)
        self.dev = Device(self.ni, self.nj)
        self.model = None
        self.nut_zero = False

    def close(self):
        self.dev.close()


def _engine_for_shape(shape):
    """The cached engine whose ghosted field shape matches, else a mesh-less context of that shape."""
    for eng in reversed(list(_engines.values())):
        if (eng.ni + 2, eng.nj + 2) == tuple(shape):
            return eng
    if len(shape) != 2 or min(shape) < 3:
        raise ValueError(f"ghosted (Ni+2, Nj+2) field expected, got shape {tuple(shape)}")
    while len(_engines) >= _MAX_ENGINES:
        _engines.pop(next(iter(_engines))).close()
    eng = _PlanesOnly(shape[0] - 2, shape[1] - 2)
    _engines[("planes", eng.ni, eng.nj)] = eng
    return eng


def sources(fields, grad_u, grad_v, mesh=None):
    """turbulence.sources (turbulence.py:401-444): (source_k, source_omega) on the interior cells from k, omega, nu_t
    and the caller's velocity gradients -- the Wilcox closure, one pointwise kernel (cfd_komega_sources).  The model
    is the one the last initialize() call saw, as in the reference (_turb_state).  The SST sources need the wall
    distance, the blending caches and the cell-centre differences of a transport step and exist only inside
    update_turbulence here (cfd_turbulence_step): asking for them stand-alone is an error, not a silent k-omega answer."""
    if _turb_state.get("model", "k_omega") == "sst":
        raise NotImplementedError("SST sources are evaluated inside update_turbulence (cfd_turbulence_step) only")
    eng = _engine_for_shape(np.shape(fields["k"]))
    dev = eng.dev
    for name in ("k", "omega", "nu_t"):
        dev.upload(_lib.FIELD_IDS[name], fields[name])
    eng.nut_zero = False
    return dev.komega_sources(grad_u, grad_v)


def effective_diffusivity(nu, nu_t, field="k", fields=None):
    """turbulence.effective_diffusivity (turbulence.py:541-577): nu + sigma nu_t (sigma_k = sigma_omega = 0.5) on the
    ghosted shape of nu_t; under SST the blended planes _D_k / _D_omega a transport step left in `fields`."""
    if fields is not None and _turb_state.get("model") == "sst":
        key = "_D_k" if field == "k" else "_D_omega"
        if key in fields:
            return fields[key]
    if field not in ("k", "omega"):
        raise ValueError(f"Unsupported field '{field}' for effective_diffusivity.")
    arr = np.asarray(nu_t, dtype=float)
    eng = _engine_for_shape(arr.shape)
    eng.dev.upload(_lib.F_NUT, arr)
    eng.nut_zero = False
    return eng.dev.effective_diffusivity(float(nu), 0.5)


def _set_tolerances(eng, mom=None, p=None, turb=None, freeze=False):
    cur = getattr(eng, "tolerances", {"mom": (1.0e-8, 200), "p": (1.0e-8, 200), "turb": (1.0e-8, 200)})
    if mom:
        cur["mom"] = mom
    if p:
        cur["p"] = p
    if turb:
        cur["turb"] = turb
    eng.tolerances = cur
    eng.dev.set_solver(cur["mom"][0], cur["mom"][1], cur["p"][0], cur["p"][1], cur["turb"][0], cur["turb"][1], freeze)
