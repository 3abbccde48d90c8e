"""ctypes binding of libcfd_hip.so (include/cfd_hip.h).

The HIP extension is the product path: if the shared library is missing or a GPU call
fails, this module raises -- there is no CPU fallback anywhere in cfd_amd.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("CFD_HIP_LIB", _HERE / "libcfd_hip.so"))

F_U, F_V, F_P, F_K, F_OMEGA, F_NUT, F_PCORR, F_AP, F_AI, F_AJ, F_BU, F_BV = range(12)
F_SST_S, F_SST_F1, F_SST_F2, F_SST_DK, F_SST_DW, F_SST_DWALL = range(12, 18)
F_U_PREV, F_V_PREV, F_P_PREV = 18, 19, 20       # snapshot planes (u_prev, v_prev, p_prev of correct_velocity)
TURB_KOMEGA, TURB_SST = 0, 1
KIND_GHOSTED, KIND_CELL, KIND_FACE_I, KIND_FACE_J = range(4)
PROBE_ALL = 1000        # CFD_PROBE_ALL of include/cfd_hip.h
RCCL_ID_BYTES = 128     # CFD_RCCL_ID_BYTES of include/cfd_hip.h
GHOST_ROWS = 4          # CFD_GHOST_ROWS of include/cfd_hip.h (checked against cfd_ghost_rows() when the library loads)
SIDE_IMIN, SIDE_IMAX, SIDE_JMIN, SIDE_JMAX = range(4)
SIDES = {"i_min": 0, "i_max": 1, "j_min": 2, "j_max": 3}
BC_NONE, BC_AFFINE, BC_SYMMETRY, BC_PERIODIC = range(4)
TOP_WRAP_I, TOP_WRAP_J, TOP_SET_INTERIOR, TOP_AFFINE_GHOST = range(4)
FIELD_IDS = {"u": F_U, "v": F_V, "p": F_P, "k": F_K, "omega": F_OMEGA, "nu_t": F_NUT, "p_corr": F_PCORR}
# the SST caches the reference keeps inside `fields` (turbulence.py:447-534, 343-355)
SST_FIELD_IDS = {"_S": F_SST_S, "_F1": F_SST_F1, "_F2": F_SST_F2, "_D_k": F_SST_DK, "_D_omega": F_SST_DW,
                 "_d_wall": F_SST_DWALL}


class TurbOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("field", C.c_int32), ("side", C.c_int32), ("lo", C.c_int32),
                ("hi", C.c_int32), ("value_off", C.c_int32), ("a", C.c_double)]


class SolveStats(C.Structure):
    _fields_ = [("mom_iters", C.c_int32), ("p_iters", C.c_int32), ("k_iters", C.c_int32),
                ("p_solver", C.c_int32), ("mom_solver", C.c_int32), ("k_solver", C.c_int32), ("mom_relres", C.c_double * 2), ("p_relres", C.c_double),
                ("k_relres", C.c_double * 2), ("p_unconverged", C.c_int32), ("transport_stalled", C.c_int32)]


class HipError(RuntimeError):
    pass


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/cfd_hip.h declares
SIGNATURES = {
    "cfd_last_error": (C.c_char_p, []),
    "cfd_version": (C.c_int, []),
    "cfd_ghost_rows": (C.c_int, []),
    "cfd_col_offset": (C.c_int, []),
    "cfd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "cfd_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _vp, C.c_size_t]),
    "cfd_destroy": (C.c_int, [_vp]),
    "cfd_set_stream": (C.c_int, [_vp, _vp]),
    "cfd_synchronize": (C.c_int, [_vp]),
    "cfd_pitch": (C.c_int, [_vp]),
    "cfd_field_offset": (C.c_int64, [_vp, C.c_int]),
    "cfd_upload_mesh": (C.c_int, [_vp] + [_dp] * 9),
    "cfd_mesh_from_nodes": (C.c_int, [_vp, _dp]),
    "cfd_mesh_from_rect": (C.c_int, [_vp, _dp, _dp]),
    "cfd_download_mesh_plane": (C.c_int, [_vp, C.c_int, _dp]),
    "cfd_upload_field": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "cfd_download_field": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "cfd_set_physics": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]),
    "cfd_set_solver": (C.c_int, [_vp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]),
    "cfd_set_multigrid": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int]),
    "cfd_set_transport_solver": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_int]),
    "cfd_set_smoother_omega2": (C.c_int, [_vp, C.c_double]),
    "cfd_set_smoother_omegas": (C.c_int, [_vp, C.c_int, _dp]),
    "cfd_set_flux_kernels": (C.c_int, [_vp, C.c_int, C.c_int]),
    "cfd_set_pressure_nullspace": (C.c_int, [_vp, C.c_int]),
    "cfd_set_turbulence_model": (C.c_int, [_vp, C.c_int]),
    "cfd_upload_cell_centers": (C.c_int, [_vp, _dp]),
    "cfd_set_wall_faces": (C.c_int, [_vp, C.c_int, _dp]),
    "cfd_sst_get_cached": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "cfd_sst_set_cached": (C.c_int, [_vp, C.c_int]),
    "cfd_set_turbulence_clip": (C.c_int, [_vp, C.c_double, C.c_double]),
    "cfd_set_flow_bc": (C.c_int, [_vp, C.c_int, C.c_int, _ip, _ip] + [_dp] * 8),
    "cfd_set_periodicity": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cfd_set_pressure_bc": (C.c_int, [_vp, _ip, _ip, _ip, _ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]),
    "cfd_set_turb_bc": (C.c_int, [_vp, C.c_int, C.POINTER(TurbOp), C.c_int, _dp]),
    "cfd_apply_flow_bc": (C.c_int, [_vp]),
    "cfd_compute_dt": (C.c_int, [_vp, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, _dp]),
    "cfd_simple_iter": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, _dp]),
    "cfd_piso_step": (C.c_int, [_vp, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, _dp]),
    "cfd_turbulence_step": (C.c_int, [_vp, C.c_double]),
    "cfd_get_stats": (C.c_int, [_vp, C.POINTER(SolveStats)]),
    "cfd_bulk_velocity": (C.c_int, [_vp, C.c_int, _dp]),
    "cfd_bulk_velocity_noalloc": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "cfd_eddy_viscosity": (C.c_int, [_vp]),
    "cfd_komega_sources": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "cfd_effective_diffusivity": (C.c_int, [_vp, C.c_double, C.c_double, _dp]),
    "cfd_momentum_diagonal": (C.c_int, [_vp, C.c_double, C.c_double]),
    "cfd_face_flux": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp]),
    "cfd_grad": (C.c_int, [_vp, C.c_int, _dp]),
    "cfd_advect": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "cfd_momentum_rhs": (C.c_int, [_vp, C.c_double]),
    "cfd_momentum_rhs_from": (C.c_int, [_vp, C.c_double, _dp, _dp, _dp, C.c_double, C.c_double]),
    "cfd_wall_friction": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp]),
    "cfd_momentum_solve": (C.c_int, [_vp, _dp]),
    "cfd_pressure_solve": (C.c_int, [_vp]),
    "cfd_download_pressure_system": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "cfd_correct": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "cfd_snapshot": (C.c_int, [_vp]),
    "cfd_continuity": (C.c_int, [_vp, _dp]),
    "cfd_apply_turb_bc": (C.c_int, [_vp]),
    "cfd_spmv": (C.c_int, [_vp, C.c_int, _dp, _dp]),
    "cfd_vcycle": (C.c_int, [_vp, _dp, _dp]),
    "cfd_pressure_krylov": (C.c_int, [_vp, _dp, _dp, C.c_double, C.c_int, C.POINTER(C.c_int), _dp]),
    "cfd_time_kernel": (C.c_int, [_vp, C.c_int, C.c_int, _dp]),
    "cfd_probe": (C.c_int, [_vp, C.c_int]),
    "cfd_probe_read": (C.c_int, [_vp, C.POINTER(C.c_int), _dp, _dp]),
    "cfd_probe_read_id": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), _dp, _dp]),
    "cfd_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "cfd_host_counters": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "cfd_host_syncs_by_phase": (C.c_int, [C.POINTER(C.c_longlong), C.c_int]),
    "cfd_kernel_counts": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.c_int, C.c_int]),
    "cfd_phase_times": (C.c_int, [_vp, _dp, C.c_int]),
    "cfd_enable_phase_timing": (C.c_int, [_vp, C.c_int]),
}

# j-slab communication callbacks (cfd_halo_fn, cfd_allreduce_fn, cfd_allgather_fn)
HALO_FN = C.CFUNCTYPE(C.c_int, _vp, C.c_int, C.POINTER(_vp), C.c_int, C.c_int, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int, C.c_int)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _vp, C.c_size_t)
SIGNATURES.update({
    "cfd_workspace_bytes_slab": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "cfd_create_slab": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "cfd_set_comm": (C.c_int, [_vp, HALO_FN, ALLREDUCE_FN, ALLGATHER_FN, _vp]),
    "cfd_exchange_field": (C.c_int, [_vp, C.c_int]),
    "cfd_rccl_unique_id": (C.c_int, [_vp, C.c_size_t]),
    "cfd_set_comm_rccl": (C.c_int, [_vp, _vp, C.c_size_t]),
    "cfd_rccl_ranks": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "cfd_comm_counts": (C.c_int, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "cfd_comm_counts_by_phase": (C.c_int, [_vp, C.POINTER(C.c_longlong)]),
    "cfd_extra_device_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "cfd_level_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "cfd_level_smoother": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _dp, C.POINTER(C.c_int)]),
})

_lib = None


def load():
    """Load libcfd_hip.so; raises HipError if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipError(f"HIP extension not built: {LIB_PATH} is missing. Run "
                       f"`python -c 'import __graft_entry__ as g; g.build()'` (make -C cfd_amd/csrc).")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.cfd_ghost_rows() != GHOST_ROWS:
        raise HipError(f"libcfd_hip.so uses {lib.cfd_ghost_rows()} ghost rows, cfd_amd._lib expects {GHOST_ROWS}")
    _lib = lib
    return lib


LAST_CALLBACK_ERROR = None      # exception raised inside a communication callback (cfd_amd.slab)


def check(rc):
    global LAST_CALLBACK_ERROR
    if rc != 0:
        msg = load().cfd_last_error().decode("utf-8", "replace")
        cause, LAST_CALLBACK_ERROR = LAST_CALLBACK_ERROR, None
        if cause is not None:
            raise HipError(f"libcfd_hip error {rc}: {msg}: {cause!r}") from cause
        raise HipError(f"libcfd_hip error {rc}: {msg}")


def dptr(a):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], "need C-contiguous float64"
    return a.ctypes.data_as(_dp)


def iptr(a):
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_ip)
