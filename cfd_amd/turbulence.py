"""Drop-in for the reference's `turbulence` module on MI355X (turbulence.py:215-577 and the transport step
solver.update_turbulence drives, solver.py:1090-1153): same names and call shapes, executed by libcfd_hip.so.

    initialize(fields, mesh, cfg, bc_cfg=None)      validation, the nu_t field, the active model
    apply_bcs(fields, mesh, bc_cfg, cfg, nu)        k / omega boundary program
    eddy_viscosity(fields, cfg)                     nu_t from k, omega (k-omega; SST with or without the limiter caches)
    sources(fields, grad_u, grad_v, mesh=None)      Wilcox source terms from the caller's velocity gradients
    effective_diffusivity(nu, nu_t, field, fields)  nu + sigma nu_t, or the SST planes cached in `fields`
    update_turbulence(fields, mesh, turb_cfg, bc_cfg, dt, nu, use_rhie_chow=True)

eddy_viscosity, sources and effective_diffusivity are pointwise and have no mesh argument: they run on a GPU context of
the field SHAPE (created on first use).  `solver.run_case` goes through none of them: its transport step
(cfd_turbulence_step) evaluates sources and conductances inside fused kernels and keeps every plane in HBM.  The SST
source terms exist only there (sources() raises for SST instead of answering with another closure).
"""
from .phases import apply_bcs, eddy_viscosity, effective_diffusivity, initialize, sources, update_turbulence

__all__ = ["initialize", "apply_bcs", "eddy_viscosity", "sources", "effective_diffusivity", "update_turbulence"]
