"""Drop-in for the reference's numba plug point (numba_kernels.py:5-83): the same two
signatures, executed by the HIP kernels of libcfd_hip.so on MI355X.

    advect_scalar_numba(phi, flux_i, flux_j, volume)                       -> (Ni, Nj)
    grad_scalar_numba(phi, w_left_i, w_right_i, w_south_j, w_north_j,
                      face_normal_i, face_normal_j, volume)                -> (Ni, Nj, 2)

These are convenience entry points for callers of the reference's function-level API:
each call uploads its arguments, runs one kernel and downloads the result.  The solver
itself never goes through them (its fused kernels keep everything in HBM).
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .device import Device

_cache = {}


def _device(ni, nj):
    key = (ni, nj)
    if key not in _cache:
        _cache.clear()
        _cache[key] = Device(ni, nj)
    return _cache[key]


def _upload_geometry(dev, volume, w_right_i, w_north_j, face_normal_i, face_normal_j):
    ni, nj = volume.shape
    ones_i, ones_j = np.ones((ni + 1, nj)), np.ones((ni, nj + 1))
    mesh = {"cell_volume": volume, "w_left_i": 1.0 - w_right_i, "w_right_i": w_right_i,
            "w_south_j": 1.0 - w_north_j, "w_north_j": w_north_j,
            "face_normal_i": face_normal_i, "face_normal_j": face_normal_j,
            "diff_d_i": ones_i, "diff_d_j": ones_j,
            "cell_spacing_i": np.ones((ni, nj)), "cell_spacing_j": np.ones((ni, nj))}
    dev.upload_mesh(mesh)


def advect_scalar_numba(phi, flux_i, flux_j, volume):
    """First-order upwind advection (numba_kernels.py:5-36)."""
    volume = np.asarray(volume, dtype=float)
    ni, nj = volume.shape
    dev = _device(ni, nj)
    half_i, half_j = np.full((ni + 1, nj), 0.5), np.full((ni, nj + 1), 0.5)
    _upload_geometry(dev, volume, half_i, half_j, np.zeros((ni + 1, nj, 2)), np.zeros((ni, nj + 1, 2)))
    dev.upload(_lib.F_K, np.asarray(phi, dtype=float))
    return dev.advect(_lib.F_K, flux_i, flux_j)


def grad_scalar_numba(phi, w_left_i, w_right_i, w_south_j, w_north_j, face_normal_i, face_normal_j, volume):
    """Green-Gauss cell gradient (numba_kernels.py:39-83)."""
    volume = np.asarray(volume, dtype=float)
    ni, nj = volume.shape
    if not (np.array_equal(w_left_i, 1.0 - np.asarray(w_right_i)) and
            np.array_equal(w_south_j, 1.0 - np.asarray(w_north_j))):
        raise ValueError("interpolation weights must be complementary (w_left = 1 - w_right)")
    dev = _device(ni, nj)
    _upload_geometry(dev, volume, np.asarray(w_right_i, dtype=float), np.asarray(w_north_j, dtype=float),
                     np.asarray(face_normal_i, dtype=float), np.asarray(face_normal_j, dtype=float))
    dev.upload(_lib.F_K, np.asarray(phi, dtype=float))
    return dev.grad(_lib.F_K)
