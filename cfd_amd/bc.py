"""Boundary-condition front end: CONFIG["boundary"] -> device tables.

The reference applies ghost-cell BCs with numpy slicing on every call
(bc.apply_boundary_conditions, bc.py:259-347; turbulence.apply_bcs, turbulence.py:291-341).
Here the segment list is resolved ONCE into per-boundary-index tables
(ghost = a * interior + b, symmetry reflection, periodic wrap) that edge kernels replay;
the k/omega program keeps the reference's segment order because wall segments rewrite
interior cells that later segments read at the corners.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .mesh import precompute_from_nodes

SIDE_NAMES = ("i_min", "i_max", "j_min", "j_max")
K_FLOOR = 1.0e-15          # turbulence.py:28
OMEGA_FLOOR = 1.0e-6       # turbulence.py:29
OMEGA_WALL_COEFF = 85.0    # turbulence.py:33


def _parse_index_range(index_range, max_index):
    """bc.py:4-13."""
    if index_range is None:
        return 0, max_index
    if len(index_range) != 2:
        raise ValueError("index_range must be (start, end)")
    start, end = index_range
    if start < 0 or end < start or end > max_index:
        raise ValueError("index_range out of bounds")
    return int(start), int(end)


def _normalize_segments(segments):
    if segments is None:
        return []
    if isinstance(segments, dict):
        return [segments]
    return list(segments)


def _periodic_segment(segments):
    """bc.py:47-58."""
    if not segments:
        return None
    types = {seg.get("type") for seg in segments}
    if "periodic" not in types:
        return None
    if types != {"periodic"}:
        raise ValueError("Periodic boundary cannot be mixed with other BC types.")
    if len(segments) != 1:
        raise ValueError("Periodic boundary must use a single segment.")
    return segments[0]


def get_periodic_pairs(bc_cfg, count_i, count_j):
    """bc.py:72-91: ((start, end) | None for i, same for j)."""
    segs = {s: _periodic_segment(_normalize_segments(bc_cfg.get(s))) for s in SIDE_NAMES}
    if (segs["i_min"] is None) != (segs["i_max"] is None):
        raise ValueError("Periodic BC must be specified on both i_min and i_max.")
    if (segs["j_min"] is None) != (segs["j_max"] is None):
        raise ValueError("Periodic BC must be specified on both j_min and j_max.")

    def match(a, b, nmax):
        ra = _parse_index_range(a.get("range"), nmax)
        rb = _parse_index_range(b.get("range"), nmax)
        if ra != rb:
            raise ValueError("Periodic boundary ranges must match on both sides.")
        if ra != (0, nmax):
            raise ValueError("Periodic boundaries must span the full boundary range.")
        return ra

    range_i = match(segs["i_min"], segs["i_max"], count_j - 1) if segs["i_min"] is not None else None
    range_j = match(segs["j_min"], segs["j_max"], count_i - 1) if segs["j_min"] is not None else None
    return range_i, range_j


class BoundaryGeometry:
    """Unit +i/+j face normal and signed cell-centre-to-face distance for every boundary
    face (bc._boundary_geometry, bc.py:133-193), from one-cell-thick strips of the node
    grid so that it costs O(perimeter) also for meshes that never exist on the host."""

    def __init__(self, strips):
        self.n_hat = {}
        self.s = {}
        for side, strip in strips.items():
            m = precompute_from_nodes(strip)
            node, cc = m["node"], m["cell_center"]
            if side == "i_min":
                fc, fn, cell = m["face_center"][0, :, 0, :], m["face_normal"][0, :, 0, :], cc[0]
            elif side == "j_min":
                fc, fn, cell = m["face_center"][:, 0, 1, :], m["face_normal"][:, 0, 1, :], cc[:, 0]
            elif side == "i_max":
                n0, n1 = node[1, :-1], node[1, 1:]
                fc = 0.5 * (n0 + n1)
                t = n1 - n0
                fn = np.stack((t[..., 1], -t[..., 0]), axis=-1)
                along = 0.5 * ((node[1, :-1] - node[0, :-1]) + (node[1, 1:] - node[0, 1:]))
                fn = fn * np.where(np.sum(fn * along, axis=-1) < 0.0, -1.0, 1.0)[..., None]
                cell = cc[0]
            else:
                n0, n1 = node[:-1, 1], node[1:, 1]
                fc = 0.5 * (n0 + n1)
                t = n1 - n0
                fn = np.stack((-t[..., 1], t[..., 0]), axis=-1)
                along = 0.5 * ((node[:-1, 1] - node[:-1, 0]) + (node[1:, 1] - node[1:, 0]))
                fn = fn * np.where(np.sum(fn * along, axis=-1) < 0.0, -1.0, 1.0)[..., None]
                cell = cc[:, 0]
            mag = np.hypot(fn[..., 0], fn[..., 1])
            nh = fn / mag[..., None]
            self.n_hat[side] = nh
            self.s[side] = np.sum((cell - fc) * nh, axis=-1)

    @classmethod
    def from_nodes(cls, node):
        node = np.asarray(node, dtype=float)
        return cls({"i_min": node[0:2], "i_max": node[-2:], "j_min": node[:, 0:2], "j_max": node[:, -2:]})

    @classmethod
    def from_rect_axes(cls, x, y):
        def grid(xs, ys):
            out = np.empty((len(xs), len(ys), 2))
            out[..., 0] = np.asarray(xs)[:, None]
            out[..., 1] = np.asarray(ys)[None, :]
            return out
        return cls({"i_min": grid(x[0:2], y), "i_max": grid(x[-2:], y),
                    "j_min": grid(x, y[0:2]), "j_max": grid(x, y[-2:])})


def wall_face_centers(node, bc_cfg):
    """Face centres (M, 2) of all boundary faces of type "wall", segment after segment
    (turbulence._wall_face_centers / _compute_wall_distance, turbulence.py:69-121)."""
    node = np.asarray(node, dtype=float)
    count_i, count_j = node.shape[0] - 1, node.shape[1] - 1
    out = []
    for boundary, segments in bc_cfg.items():
        if boundary == "pressure_reference" or segments is None:
            continue
        for seg in _normalize_segments(segments):
            if seg.get("type") != "wall":
                continue
            rng = seg.get("range")
            if boundary in ("i_min", "i_max"):
                start, end = (0, count_j - 1) if rng is None else rng
                col = 0 if boundary == "i_min" else count_i
                n0, n1 = node[col, start:end + 1, :], node[col, start + 1:end + 2, :]
            else:
                start, end = (0, count_i - 1) if rng is None else rng
                row = 0 if boundary == "j_min" else count_j
                n0, n1 = node[start:end + 1, row, :], node[start + 1:end + 2, row, :]
            out.append(0.5 * (n0 + n1))
    if not out:
        return np.zeros((0, 2))
    return np.ascontiguousarray(np.concatenate(out, axis=0))


def _expand(value, n):
    """bc.py:123-130."""
    if np.isscalar(value):
        return np.full(n, value, dtype=float)
    arr = np.asarray(value, dtype=float)
    if arr.shape != (n,):
        raise ValueError(f"value has shape {arr.shape}, expected {(n,)}")
    return arr


def pressure_dirichlet_masks(bc_cfg, count_i, count_j):
    """solver._pressure_dirichlet_masks (solver.py:245-276)."""
    masks = {"i_min": np.zeros(count_j, dtype=bool), "i_max": np.zeros(count_j, dtype=bool),
             "j_min": np.zeros(count_i, dtype=bool), "j_max": np.zeros(count_i, dtype=bool)}
    has = False
    for boundary, segments in bc_cfg.items():
        if boundary == "pressure_reference" or segments is None:
            continue
        for seg in _normalize_segments(segments):
            if seg.get("type") != "outlet" and "pressure" not in seg:
                continue
            if boundary not in masks:
                raise ValueError(f"Unknown boundary: {boundary}")
            nmax = (count_j if boundary[0] == "i" else count_i) - 1
            start, end = _parse_index_range(seg.get("range"), nmax)
            masks[boundary][start:end + 1] = True
            has = True
    return masks, has


class BoundaryTables:
    """Everything the device needs to replay the boundary conditions."""

    def __init__(self, bc_cfg, count_i, count_j, geometry, nu, turbulent):
        ni, nj = int(count_i), int(count_j)
        self.ni, self.nj = ni, nj
        range_i, range_j = get_periodic_pairs(bc_cfg, ni, nj)
        self.per_i, self.per_j = range_i is not None, range_j is not None
        side_len = {"i_min": nj, "i_max": nj, "j_min": ni, "j_max": ni}
        for side in bc_cfg:
            if side != "pressure_reference" and side not in side_len:
                raise ValueError(f"Unknown boundary: {side}")

        def has_p(sides):
            return any("pressure" in seg for s in sides for seg in _normalize_segments(bc_cfg.get(s)))
        self.wrap_p_i = self.per_i and not has_p(("i_min", "i_max"))      # bc.py:266-286
        self.wrap_p_j = self.per_j and not has_p(("j_min", "j_max"))

        # ---- flow tables ------------------------------------------------------------
        self.flow = []
        tabs = {}
        for side in SIDE_NAMES:
            n = side_len[side]
            tabs[side] = {"kind_uv": np.zeros(n, np.int32), "kind_p": np.zeros(n, np.int32),
                          **{k: np.zeros(n) for k in ("ua", "ub", "va", "vb", "pa", "pb", "nx", "ny")}}
        periodic_sides = set()
        if self.per_i:
            periodic_sides.update(("i_min", "i_max"))
        if self.per_j:
            periodic_sides.update(("j_min", "j_max"))
        for side in periodic_sides:
            tabs[side]["kind_uv"][:] = _lib.BC_PERIODIC
            tabs[side]["kind_p"][:] = _lib.BC_PERIODIC

        def dirichlet(t, pre, sl, value):
            t[pre + "a"][sl] = -1.0
            t[pre + "b"][sl] = 2.0 * _expand(value, sl.stop - sl.start)

        def neumann(t, pre, sl, grad, s):
            t[pre + "a"][sl] = 1.0
            t[pre + "b"][sl] = -(2.0 * s[sl] * _expand(grad, sl.stop - sl.start))

        for side, segments in bc_cfg.items():
            if side == "pressure_reference":
                continue
            t = tabs[side]
            s_geo = geometry.s[side]
            nmax = side_len[side] - 1
            if side in periodic_sides:
                for seg in _normalize_segments(segments):
                    lo, hi = _parse_index_range(seg.get("range"), nmax)
                    sl = slice(lo, hi + 1)
                    if "pressure" in seg:
                        t["kind_p"][sl] = _lib.BC_AFFINE
                        dirichlet(t, "p", sl, seg["pressure"])
                    elif "pressure_gradient" in seg:
                        t["kind_p"][sl] = _lib.BC_AFFINE
                        neumann(t, "p", sl, seg["pressure_gradient"], s_geo)
                continue
            if segments is None:
                continue
            for seg in _normalize_segments(segments):
                kind = seg["type"]
                lo, hi = _parse_index_range(seg.get("range"), nmax)
                sl = slice(lo, hi + 1)
                if kind == "inlet":
                    t["kind_uv"][sl] = _lib.BC_AFFINE
                    dirichlet(t, "u", sl, seg["velocity"][0])
                    dirichlet(t, "v", sl, seg["velocity"][1])
                    t["kind_p"][sl] = _lib.BC_AFFINE
                    neumann(t, "p", sl, seg.get("pressure_gradient", 0.0), s_geo)
                elif kind == "outlet":
                    g = seg.get("velocity_gradient", (0.0, 0.0))
                    t["kind_uv"][sl] = _lib.BC_AFFINE
                    neumann(t, "u", sl, g[0], s_geo)
                    neumann(t, "v", sl, g[1], s_geo)
                    t["kind_p"][sl] = _lib.BC_AFFINE
                    dirichlet(t, "p", sl, seg["pressure"])
                elif kind == "wall":
                    w = seg.get("velocity", (0.0, 0.0))
                    t["kind_uv"][sl] = _lib.BC_AFFINE
                    dirichlet(t, "u", sl, w[0])
                    dirichlet(t, "v", sl, w[1])
                    t["kind_p"][sl] = _lib.BC_AFFINE
                    neumann(t, "p", sl, seg.get("pressure_gradient", 0.0), s_geo)
                elif kind == "symmetry":
                    t["kind_uv"][sl] = _lib.BC_SYMMETRY
                    t["nx"][sl] = geometry.n_hat[side][sl, 0]
                    t["ny"][sl] = geometry.n_hat[side][sl, 1]
                    t["kind_p"][sl] = _lib.BC_AFFINE
                    neumann(t, "p", sl, 0.0, s_geo)
                else:
                    raise ValueError(f"Unknown BC type: {kind}")
        self.flow = [tabs[s] for s in SIDE_NAMES]

        # ---- pressure-correction BCs -------------------------------------------------
        masks, has = pressure_dirichlet_masks(bc_cfg, ni, nj)
        self.has_dirichlet = has
        self.p_masks = [np.ascontiguousarray(masks[s], dtype=np.int32) for s in SIDE_NAMES]
        self.pp_per_i = self.per_i and not (masks["i_min"].any() or masks["i_max"].any())
        self.pp_per_j = self.per_j and not (masks["j_min"].any() or masks["j_max"].any())
        ref = bc_cfg.get("pressure_reference", {})
        self.pin = tuple(int(q) for q in ref.get("cell", (0, 0)))
        self.pin_value = float(ref.get("value", 0.0))

        # ---- k / omega program (turbulence.py:291-341) --------------------------------
        self.turb_ops = []
        vals = []

        def push(op, field, side, lo, hi, a, values):
            off = sum(len(v) for v in vals)
            vals.append(np.asarray(values, dtype=float))
            self.turb_ops.append((op, field, _lib.SIDES[side], lo, hi, off, a))

        if turbulent:
            if self.per_i:
                self.turb_ops.append((_lib.TOP_WRAP_I, _lib.F_K, 0, 0, 0, 0, 0.0))
            if self.per_j:
                self.turb_ops.append((_lib.TOP_WRAP_J, _lib.F_K, 0, 0, 0, 0, 0.0))
            for side, segments in bc_cfg.items():
                if side == "pressure_reference" or segments is None:
                    continue
                nmax = side_len[side] - 1
                for seg in _normalize_segments(segments):
                    kind = seg.get("type")
                    lo, hi = _parse_index_range(seg.get("range"), nmax)
                    n = hi - lo + 1
                    sl = slice(lo, hi + 1)

                    def dir_op(field, value):
                        push(_lib.TOP_AFFINE_GHOST, field, side, lo, hi, -1.0, 2.0 * _expand(value, n))

                    def neu_op(field, grad):
                        push(_lib.TOP_AFFINE_GHOST, field, side, lo, hi, 1.0,
                             -(2.0 * geometry.s[side][sl] * _expand(grad, n)))

                    if kind == "inlet":
                        dir_op(_lib.F_K, seg["k"])
                        dir_op(_lib.F_OMEGA, seg["omega"])
                    elif kind == "wall":
                        y = np.maximum(np.abs(geometry.s[side][sl]), 1.0e-12)
                        ow = np.maximum(OMEGA_WALL_COEFF * nu / (y ** 2), OMEGA_FLOOR)
                        push(_lib.TOP_SET_INTERIOR, _lib.F_K, side, lo, hi, 0.0, np.full(n, K_FLOOR))
                        push(_lib.TOP_SET_INTERIOR, _lib.F_OMEGA, side, lo, hi, 0.0, ow)
                        dir_op(_lib.F_K, K_FLOOR)
                        dir_op(_lib.F_OMEGA, ow)
                    elif kind in ("outlet", "symmetry"):
                        if "k" in seg:
                            dir_op(_lib.F_K, seg["k"])
                        else:
                            neu_op(_lib.F_K, seg.get("k_gradient", 0.0))
                        if "omega" in seg:
                            dir_op(_lib.F_OMEGA, seg["omega"])
                        else:
                            neu_op(_lib.F_OMEGA, seg.get("omega_gradient", 0.0))
                    elif kind == "periodic":
                        continue
                    else:
                        raise ValueError(f"Unknown BC type for turbulence: {kind}")
        self.turb_values = np.concatenate(vals) if vals else np.zeros(0)

    # ---- j-slab view ---------------------------------------------------
    def slab(self, j_offset, nj_local):
        """Tables of the rows [j_offset, j_offset + nj_local) for one rank of a j-slab run.

        i_min / i_max tables and masks are sliced; j_min / j_max tables stay (the kernels skip a
        side that is a cut); the k / omega program keeps only the operations that touch this slab,
        re-based to local indices.  The pressure reference cell stays in global coordinates (the
        library resolves the owner)."""
        import copy
        if self.per_j:
            raise NotImplementedError("periodicity along j is not supported across j-slabs")
        j0, j1 = int(j_offset), int(j_offset) + int(nj_local)
        if j0 < 0 or j1 > self.nj or nj_local < 1:
            raise ValueError("slab outside the mesh")
        first, last = j0 == 0, j1 == self.nj
        out = copy.copy(self)
        out.nj = int(nj_local)
        out.flow = [{k: np.ascontiguousarray(v[j0:j1]) for k, v in t.items()} if side < 2 else t
                    for side, t in enumerate(self.flow)]
        out.p_masks = [np.ascontiguousarray(m[j0:j1]) if side < 2 else m for side, m in enumerate(self.p_masks)]
        ops, vals, off_new = [], [], 0
        for (op, field, side, lo, hi, off, a) in self.turb_ops:
            if op in (_lib.TOP_WRAP_I, _lib.TOP_WRAP_J):
                ops.append((op, field, side, lo, hi, 0, a))
                continue
            if side >= 2:
                if (side == 2 and not first) or (side == 3 and not last):
                    continue
                nlo, nhi, src = lo, hi, off
            else:
                glo, ghi = max(lo, j0), min(hi, j1 - 1)
                if glo > ghi:
                    continue
                nlo, nhi, src = glo - j0, ghi - j0, off + (glo - lo)
            n = nhi - nlo + 1
            vals.append(self.turb_values[src:src + n])
            ops.append((op, field, side, nlo, nhi, off_new, a))
            off_new += n
        out.turb_ops = ops
        out.turb_values = np.concatenate(vals) if vals else np.zeros(0)
        return out
