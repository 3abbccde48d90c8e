"""In-loop wall-friction diagnostic (post_processing.friction_coefficient_array,
post_processing.py:588-687 with the segment geometry of 103-170).

run_case uses WallFriction + friction_coefficient_device: the geometry of the boundary segment (O(perimeter),
from the mesh, once per run) goes to the GPU as a small table, an edge kernel gathers u, v, nu_t along the wall
from the planes in HBM and sums tau * length there (cfd_wall_friction), the bulk velocity is a device reduction
too -- no field is downloaded for the friction log.  friction_coefficient_array keeps the reference's call
shape for callers that hold the fields on the host.  Plotting is out of scope."""
from __future__ import annotations

import numpy as np

from .bc import _parse_index_range


def _wall_segment(mesh, boundary, index_range):
    node = mesh["node"]
    cc = mesh["cell_center"]
    ni, nj = node.shape[0] - 1, node.shape[1] - 1
    if boundary in ("i_min", "i_max"):
        lo, hi = _parse_index_range(index_range, nj - 1)
        idx = np.arange(lo, hi + 1)
        at_min = boundary == "i_min"
        line = node[0 if at_min else ni, lo:hi + 2]
        cell = cc[0 if at_min else ni - 1, idx]
        tang = line[1:] - line[:-1]
        normal = np.stack((tang[:, 1], -tang[:, 0]), axis=-1)
        i_int = np.full_like(idx, 1 if at_min else ni)
        j_int = idx + 1
        i_gh, j_gh = np.full_like(idx, 0 if at_min else ni + 1), j_int
        if at_min:
            fc, normal = mesh["face_center"][0, idx, 0, :], mesh["face_normal"][0, idx, 0, :]
    elif boundary in ("j_min", "j_max"):
        lo, hi = _parse_index_range(index_range, ni - 1)
        idx = np.arange(lo, hi + 1)
        at_min = boundary == "j_min"
        line = node[lo:hi + 2, 0 if at_min else nj]
        cell = cc[idx, 0 if at_min else nj - 1]
        tang = line[1:] - line[:-1]
        normal = np.stack((-tang[:, 1], tang[:, 0]), axis=-1)
        i_int = idx + 1
        j_int = np.full_like(idx, 1 if at_min else nj)
        i_gh, j_gh = i_int, np.full_like(idx, 0 if at_min else nj + 1)
        if at_min:
            fc, normal = mesh["face_center"][idx, 0, 1, :], mesh["face_normal"][idx, 0, 1, :]
    else:
        raise ValueError("boundary must be one of: i_min, i_max, j_min, j_max")
    if not at_min:
        fc = 0.5 * (line[:-1] + line[1:])
        normal = normal * np.where(np.sum(normal * (fc - cell), axis=-1) < 0.0, -1.0, 1.0)[:, None]
    seg_len = np.hypot(tang[:, 0], tang[:, 1])
    n_hat = normal / np.hypot(normal[:, 0], normal[:, 1])[:, None]
    return fc, cell, n_hat, seg_len, (i_int, j_int), (i_gh, j_gh)


def friction_coefficient_array(fields, mesh, boundary, index_range=None, nu=1.0e-3, rho=1.0, u_ref=None,
                               use_nu_t=False, flow_axis="i", hydraulic_diameter=None, pressure_gradient=None):
    if u_ref is None:
        vel = fields["u" if flow_axis == "i" else "v"][1:-1, 1:-1]
        volume = np.asarray(mesh["cell_volume"])
        if volume.shape != vel.shape:
            raise ValueError("cell_volume shape does not match velocity field.")
        u_ref = float(np.sum(vel * volume) / np.sum(volume))
    if u_ref == 0.0 or not np.isfinite(u_ref):
        raise ValueError("u_ref must be nonzero for friction coefficient.")
    u, v = fields["u"], fields["v"]
    ni, nj = mesh["node"].shape[0] - 1, mesh["node"].shape[1] - 1
    if u.shape != (ni + 2, nj + 2):
        raise ValueError("Velocity fields must include ghost cells for friction calc.")
    fc, cell, n_hat, seg_len, inner, ghost = _wall_segment(mesh, boundary, index_range)
    t_hat = np.stack((n_hat[:, 1], -n_hat[:, 0]), axis=-1)
    ut_cell = u[inner] * t_hat[:, 0] + v[inner] * t_hat[:, 1]
    ut_ghost = u[ghost] * t_hat[:, 0] + v[ghost] * t_hat[:, 1]
    dn = np.abs(np.sum((fc - cell) * n_hat, axis=-1))
    dn = np.where(dn > 1.0e-12, dn, 1.0e-12)
    nu_eff = float(nu)
    if use_nu_t and "nu_t" in fields:
        nu_eff = nu + fields["nu_t"][inner]
    tau = rho * nu_eff * ((ut_cell - ut_ghost) / (2.0 * dn))
    length = np.sum(seg_len)
    tau_avg = float(np.sum(tau * seg_len) / length) if length > 0.0 else 0.0
    cf = 2.0 * tau_avg / (float(rho) * float(u_ref) ** 2)
    re = None
    if hydraulic_diameter is not None:
        re = float(rho) * float(u_ref) * float(hydraulic_diameter) / float(nu)
    cf_dpdx = None
    if pressure_gradient is not None and hydraulic_diameter is not None:
        dp = float(pressure_gradient[0] if flow_axis == "i" else pressure_gradient[1])
        cf_dpdx = 2.0 * (-dp * float(hydraulic_diameter) / 4.0) / (float(rho) * float(u_ref) ** 2)
    s_face = np.concatenate(([0.0], np.cumsum(seg_len)))[:-1] + 0.5 * seg_len
    return {"cf": float(cf), "tau_avg": float(tau_avg), "bulk_velocity": float(u_ref), "re": re,
            "cf_dpdx": cf_dpdx, "tau": tau, "s": s_face}


class WallFriction:
    """Geometry of one boundary segment for the device-side friction diagnostic (built once per run)."""

    def __init__(self, mesh, boundary, index_range=None):
        from ._lib import SIDES
        fc, cell, n_hat, seg_len, inner, _ = _wall_segment(mesh, boundary, index_range)
        self.side = SIDES[boundary]
        along = inner[1] - 1 if boundary in ("i_min", "i_max") else inner[0] - 1
        self.lo, self.hi = int(along[0]), int(along[-1])
        self.t_hat = np.ascontiguousarray(np.stack((n_hat[:, 1], -n_hat[:, 0]), axis=-1))
        dn = np.abs(np.sum((fc - cell) * n_hat, axis=-1))
        self.dn = np.where(dn > 1.0e-12, dn, 1.0e-12)
        self.seg_len = np.ascontiguousarray(seg_len)
        self.length = float(np.sum(seg_len))
        self.s_face = np.concatenate(([0.0], np.cumsum(seg_len)))[:-1] + 0.5 * seg_len


def friction_coefficient_device(dev, wall, nu=1.0e-3, rho=1.0, u_ref=None, use_nu_t=False, flow_axis="i",
                                hydraulic_diameter=None, pressure_gradient=None, want_tau=False):
    """friction_coefficient_array (post_processing.py:588-687) on the fields a Device holds in HBM.
    want_tau adds the per-face shear stresses ("tau", "s") to the result, an O(segment) download."""
    from . import _lib
    if u_ref is None:
        u_ref = dev.bulk_velocity(_lib.F_U if flow_axis == "i" else _lib.F_V)
    if u_ref == 0.0 or not np.isfinite(u_ref):
        raise ValueError("u_ref must be nonzero for friction coefficient.")
    got = dev.wall_friction(wall.side, wall.lo, wall.hi, wall.t_hat, wall.dn, wall.seg_len, nu, rho, use_nu_t, want_tau)
    tau_avg = float(got[0] / wall.length) if wall.length > 0.0 else 0.0
    cf = 2.0 * tau_avg / (float(rho) * float(u_ref) ** 2)
    re = None
    if hydraulic_diameter is not None:
        re = float(rho) * float(u_ref) * float(hydraulic_diameter) / float(nu)
    cf_dpdx = None
    if pressure_gradient is not None and hydraulic_diameter is not None:
        dp = float(pressure_gradient[0] if flow_axis == "i" else pressure_gradient[1])
        cf_dpdx = 2.0 * (-dp * float(hydraulic_diameter) / 4.0) / (float(rho) * float(u_ref) ** 2)
    out = {"cf": float(cf), "tau_avg": tau_avg, "bulk_velocity": float(u_ref), "re": re, "cf_dpdx": cf_dpdx}
    if want_tau:
        out["tau"], out["s"] = got[2], wall.s_face
    return out
