"""Mesh generators and the mesh `.npz` data contract (host-side set-up).

Mirrors the public names of the reference's mesh module (mesh.py:82-183, 402-483):
`generate_rect_mesh`, `generate_annulus_mesh`, `generate_ramp_mesh`, `make_mesh`,
`precompute_from_nodes`, `ensure_single_mesh_file`.  The dict / .npz keys and shapes are
those of mesh.precompute_from_nodes (mesh.py:375-394), so meshes written by either code
are interchangeable.  For large synthetic meshes the metric planes are computed on the
GPU instead (cfd_amd.device.Device.mesh_from_nodes / mesh_from_rect) and this host
contract is bypassed.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np


def _graded(length, count, ratio):
    """Node coordinates of `count` cells on [0, length]; geometric growth `ratio`, or a
    list of (fraction, ratio) segments (mesh.py:4-79)."""
    if count <= 0:
        raise ValueError("count must be positive")
    if length <= 0.0:
        raise ValueError("length must be positive")
    if isinstance(ratio, (list, tuple)) and ratio and isinstance(ratio[0], (list, tuple)):
        parts = [(float(a), float(b)) for a, b in ratio]
        if abs(sum(a for a, _ in parts) - 1.0) > 1.0e-12:
            raise ValueError("segment fractions must sum to 1.0")
        if any(a <= 0.0 for a, _ in parts):
            raise ValueError("segment fraction must be positive")
        if any(b <= 0.0 for _, b in parts):
            raise ValueError("segment expansion ratio must be positive")
        edges = np.round(np.cumsum(np.array([a for a, _ in parts])) * count).astype(int)
        cells = np.diff(np.concatenate(([0], edges)))
        if cells.sum() != count:
            cells[-1] += count - cells.sum()
        if np.any(cells <= 0):
            raise ValueError("segment has zero cells; increase count or adjust fractions")
        coords, start = [0.0], 0.0
        for (frac, grow), n in zip(parts, cells):
            seg = _graded(length * frac, int(n), grow)
            coords.extend((seg[1:] + start).tolist())
            start += length * frac
        coords[-1] = length
        return np.array(coords, dtype=float)
    if ratio <= 0.0:
        raise ValueError("ratio must be positive")
    if count == 1:
        return np.array([0.0, length], dtype=float)
    if abs(ratio - 1.0) < 1.0e-12:
        return np.linspace(0.0, length, count + 1)
    w0 = length * (1.0 - ratio) / (1.0 - ratio ** count)
    x = np.concatenate(([0.0], np.cumsum(w0 * ratio ** np.arange(count, dtype=float))))
    x[-1] = length
    return x


def rect_node_axes(length_i, length_j, count_i, count_j, ratio_i=1.0, ratio_j=1.0):
    """The two 1-D node coordinate arrays of a rectangular mesh."""
    return _graded(length_i, count_i, ratio_i), _graded(length_j, count_j, ratio_j)


def generate_rect_nodes(length_i, length_j, count_i, count_j, ratio_i=1.0, ratio_j=1.0):
    xi, yj = rect_node_axes(length_i, length_j, count_i, count_j, ratio_i, ratio_j)
    node = np.empty((count_i + 1, count_j + 1, 2))
    node[..., 0] = xi[:, None]
    node[..., 1] = yj[None, :]
    return node


def generate_annulus_nodes(radius_inner, radius_outer, count_r, count_t, theta_min=0.0,
                           theta_max=np.pi, ratio_r=1.0, ratio_t=1.0):
    if radius_inner <= 0.0 or radius_outer <= 0.0:
        raise ValueError("radii must be positive")
    if radius_outer <= radius_inner:
        raise ValueError("radius_outer must be greater than radius_inner")
    if count_r <= 0 or count_t <= 0:
        raise ValueError("cell counts must be positive")
    if theta_max <= theta_min:
        raise ValueError("theta_max must be greater than theta_min")
    rad = _graded(radius_outer - radius_inner, count_r, ratio_r) + radius_inner
    ang = _graded(theta_max - theta_min, count_t, ratio_t) + theta_min
    node = np.empty((count_r + 1, count_t + 1, 2))
    node[..., 0] = rad[:, None] * np.cos(ang)[None, :]
    node[..., 1] = rad[:, None] * np.sin(ang)[None, :]
    return node


def generate_ramp_nodes(length_i, height_in, height_out, count_i, count_j, ramp_start, ramp_end,
                        ratio_i=1.0, ratio_j=1.0):
    if length_i <= 0.0:
        raise ValueError("length_i must be positive")
    if height_in <= 0.0 or height_out <= 0.0:
        raise ValueError("heights must be positive")
    if ramp_start < 0.0 or ramp_end < 0.0:
        raise ValueError("ramp positions must be non-negative")
    if ramp_end <= ramp_start:
        raise ValueError("ramp_end must be greater than ramp_start")
    if ramp_end > length_i:
        raise ValueError("ramp_end must be within the domain length")
    xi = _graded(length_i, count_i, ratio_i)
    eta = _graded(1.0, count_j, ratio_j)
    t = np.clip((xi - ramp_start) / (ramp_end - ramp_start), 0.0, 1.0)
    h = np.where(xi <= ramp_start, height_in, np.where(xi >= ramp_end, height_out,
                                                       height_in + t * (height_out - height_in)))
    node = np.zeros((count_i + 1, count_j + 1, 2))
    node[..., 0] = xi[:, None]
    node[..., 1] = eta[None, :] * h[:, None]
    return node


def _flip_to(n, direction):
    return n * np.where((n * direction).sum(axis=-1) < 0.0, -1.0, 1.0)[..., None]


def _weights(c_lo, c_hi, f):
    span = c_hi - c_lo
    den = span[..., 0] ** 2 + span[..., 1] ** 2
    den = np.where(den > 1.0e-14, den, 1.0)
    t = np.clip(((f - c_lo) * span).sum(axis=-1) / den, 0.0, 1.0)
    return 1.0 - t, t


def precompute_from_nodes(node, extra_fields=None):
    """Cell / face metrics of a curvilinear node grid, keys as in mesh.py:375-394."""
    node = np.asarray(node, dtype=float)
    if node.ndim != 3 or node.shape[2] != 2:
        raise ValueError("node must have shape (count_i+1, count_j+1, 2)")
    ni, nj = node.shape[0] - 1, node.shape[1] - 1
    sw, se, ne, nw = node[:-1, :-1], node[1:, :-1], node[1:, 1:], node[:-1, 1:]

    def cross(a, b):
        return a[..., 0] * b[..., 1] - b[..., 0] * a[..., 1]

    k0, k1, k2, k3 = cross(sw, se), cross(se, ne), cross(ne, nw), cross(nw, sw)
    signed = 0.5 * (k0 + k1 + k2 + k3)
    centre = np.empty((ni, nj, 2))
    for d in (0, 1):
        centre[..., d] = ((sw[..., d] + se[..., d]) * k0 + (se[..., d] + ne[..., d]) * k1
                          + (ne[..., d] + nw[..., d]) * k2 + (nw[..., d] + sw[..., d]) * k3) / (6.0 * signed)

    south, north = se - sw, ne - nw        # i-direction edges
    west, east = nw - sw, ne - se          # j-direction edges
    len_s, len_n = np.hypot(south[..., 0], south[..., 1]), np.hypot(north[..., 0], north[..., 1])
    len_w, len_e = np.hypot(west[..., 0], west[..., 1]), np.hypot(east[..., 0], east[..., 1])

    face_center = np.zeros((ni, nj, 2, 2))
    face_center[:, :, 0] = 0.5 * (sw + nw)
    face_center[:, :, 1] = 0.5 * (sw + se)
    face_size = np.stack((len_w, len_s), axis=-1)
    face_normal = np.zeros((ni, nj, 2, 2))
    face_normal[:, :, 0] = _flip_to(np.stack((west[..., 1], -west[..., 0]), axis=-1), 0.5 * (south + north))
    face_normal[:, :, 1] = _flip_to(np.stack((-south[..., 1], south[..., 0]), axis=-1), 0.5 * (west + east))

    fc_i = np.concatenate((face_center[:, :, 0], (0.5 * (node[ni, :-1] + node[ni, 1:]))[None]), axis=0)
    fc_j = np.concatenate((face_center[:, :, 1], (0.5 * (node[:-1, nj] + node[1:, nj]))[:, None]), axis=1)
    t = node[ni, 1:] - node[ni, :-1]
    last_i = _flip_to(np.stack((t[..., 1], -t[..., 0]), axis=-1), fc_i[ni] - centre[ni - 1])
    t = node[1:, nj] - node[:-1, nj]
    last_j = _flip_to(np.stack((-t[..., 1], t[..., 0]), axis=-1), fc_j[:, nj] - centre[:, nj - 1])
    fn_i = np.concatenate((face_normal[:, :, 0], last_i[None]), axis=0)
    fn_j = np.concatenate((face_normal[:, :, 1], last_j[:, None]), axis=1)

    wl_i = np.full((ni + 1, nj), 0.5)
    wr_i = np.full((ni + 1, nj), 0.5)
    wl_i[1:ni], wr_i[1:ni] = _weights(centre[:-1], centre[1:], fc_i[1:ni])
    ws_j = np.full((ni, nj + 1), 0.5)
    wn_j = np.full((ni, nj + 1), 0.5)
    ws_j[:, 1:nj], wn_j[:, 1:nj] = _weights(centre[:, :-1], centre[:, 1:], fc_j[:, 1:nj])

    def unit(n):
        mag = np.hypot(n[..., 0], n[..., 1])
        return n / np.where(mag > 1.0e-14, mag, 1.0)[..., None]

    ui, uj = unit(fn_i), unit(fn_j)
    d_i = np.zeros((ni + 1, nj))
    d_j = np.zeros((ni, nj + 1))
    d_i[1:ni] = np.abs(((centre[1:] - centre[:-1]) * ui[1:ni]).sum(axis=-1))
    d_j[:, 1:nj] = np.abs(((centre[:, 1:] - centre[:, :-1]) * uj[:, 1:nj]).sum(axis=-1))
    d_i[0] = 2.0 * np.abs(((centre[0] - fc_i[0]) * ui[0]).sum(axis=-1))
    d_i[ni] = 2.0 * np.abs(((centre[ni - 1] - fc_i[ni]) * ui[ni]).sum(axis=-1))
    d_j[:, 0] = 2.0 * np.abs(((centre[:, 0] - fc_j[:, 0]) * uj[:, 0]).sum(axis=-1))
    d_j[:, nj] = 2.0 * np.abs(((centre[:, nj - 1] - fc_j[:, nj]) * uj[:, nj]).sum(axis=-1))

    mesh = {
        "node": node,
        "cell_center": centre,
        "face_center": face_center,
        "cell_volume": np.abs(signed),
        "face_size": face_size,
        "face_normal": face_normal,
        "cell_spacing_i": 0.5 * (len_s + len_n),
        "cell_spacing_j": 0.5 * (len_w + len_e),
        "w_left_i": wl_i, "w_right_i": wr_i, "w_south_j": ws_j, "w_north_j": wn_j,
        "diff_d_i": np.where(d_i > 1.0e-14, d_i, 1.0e-14),
        "diff_d_j": np.where(d_j > 1.0e-14, d_j, 1.0e-14),
        "face_normal_i": fn_i, "face_normal_j": fn_j,
        "count_i": np.array(ni, dtype=int), "count_j": np.array(nj, dtype=int),
    }
    if extra_fields:
        mesh.update(extra_fields)
    return mesh


def _finish(node, out_path, extra_fields):
    mesh = precompute_from_nodes(node, extra_fields=extra_fields)
    if out_path:
        np.savez(out_path, **mesh)
    return mesh


def generate_rect_mesh(length_i, length_j, count_i, count_j, ratio_i=1.0, ratio_j=1.0,
                       out_path=None, extra_fields=None):
    return _finish(generate_rect_nodes(length_i, length_j, count_i, count_j, ratio_i, ratio_j),
                   out_path, extra_fields)


def generate_annulus_mesh(radius_inner, radius_outer, count_r, count_t, theta_min=0.0, theta_max=np.pi,
                          ratio_r=1.0, ratio_t=1.0, out_path=None, extra_fields=None):
    return _finish(generate_annulus_nodes(radius_inner, radius_outer, count_r, count_t, theta_min=theta_min,
                                          theta_max=theta_max, ratio_r=ratio_r, ratio_t=ratio_t),
                   out_path, extra_fields)


def generate_ramp_mesh(length_i, height_in, height_out, count_i, count_j, ramp_start, ramp_end,
                       ratio_i=1.0, ratio_j=1.0, out_path=None, extra_fields=None):
    return _finish(generate_ramp_nodes(length_i, height_in, height_out, count_i, count_j, ramp_start,
                                       ramp_end, ratio_i, ratio_j), out_path, extra_fields)


def ensure_single_mesh_file(mesh_path):
    """mesh.py:119-131."""
    mesh_path = Path(mesh_path)
    if not mesh_path.parent.exists():
        return
    others = [p for p in sorted(mesh_path.parent.glob("*.npz")) if p.resolve() != mesh_path.resolve()]
    if others:
        names = ", ".join(p.name for p in others)
        raise ValueError(f"Multiple mesh files found in {mesh_path.parent}: {names}. Remove extras.")


def make_mesh(mesh_cfg):
    """mesh.py:176-183: (re)generate the mesh file with the configured generator."""
    path = Path(mesh_cfg["path"])
    path.parent.mkdir(parents=True, exist_ok=True)
    ensure_single_mesh_file(path)
    if path.exists():
        path.unlink()
    return mesh_cfg["generator"](out_path=str(path), **mesh_cfg["params"])
