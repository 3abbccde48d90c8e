"""Drop-in for the reference's `turbulence` module on MI355X (turbulence.py:215-394 and the transport step
solver.update_turbulence drives, solver.py:1090-1153): same names and call shapes, executed by libcfd_hip.so.

    initialize(fields, mesh, cfg, bc_cfg=None)      validation + the nu_t field
    apply_bcs(fields, mesh, bc_cfg, cfg, nu)        k / omega boundary program
    eddy_viscosity(fields, cfg)                     nu_t from k, omega (k-omega; SST with or without the limiter caches)
    update_turbulence(fields, mesh, turb_cfg, bc_cfg, dt, nu, use_rhie_chow=True)

turbulence.sources / effective_diffusivity have no stand-alone form here: the GPU evaluates them inside the
fused right-hand-side and conductance kernels of the transport step (cfd_turbulence_step) and never materialises
them; `solver.run_case` does not go through this module either (its loop keeps every plane in HBM).
"""
from .phases import apply_bcs, eddy_viscosity, initialize, update_turbulence

__all__ = ["initialize", "apply_bcs", "eddy_viscosity", "update_turbulence"]
