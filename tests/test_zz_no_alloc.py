"""cfd_bulk_velocity_noalloc (the entry point that allocates nothing, include/cfd_hip.h) against cfd_bulk_velocity:
same bits, on two calls in a row of which the second has the smaller input (the static scratch array keeps the first
call's partial sums).  The second call also runs on a stream of the caller's and on planes with a pitch larger than
the library's own (a caller's allocation): neither changes tiles or summation order."""
import ctypes as C

import numpy as np
import pytest

from oracle import cfd_oracle as orc

pytestmark = pytest.mark.gpu


def _planes(torch, ni, nj, pitch, ng, vol, u, co):
    """Device planes in the library's layout: cell (i, j) at (j + ng) * pitch + i + co (CFD_GHOST_ROWS, CFD_COL_OFFSET)."""
    out = []
    for a in (u, vol):
        host = np.zeros((nj + 2 * ng, pitch))
        host[ng:ng + nj, co:co + ni] = a.T
        out.append(torch.from_numpy(host).cuda())
    return out


def test_no_alloc_entry_point_matches_bulk_velocity():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from cfd_amd import _lib as L
    from cfd_amd.device import Device
    lib = L.load()
    ng = lib.cfd_ghost_rows()
    rng = np.random.default_rng(3)
    keep = []
    # second call: fewer workgroups than the first left partial sums for; user stream; padded pitch
    for ni, nj, own_stream, pad in ((700, 520, False, 0), (96, 40, True, 64)):
        mesh = orc.rect_mesh(2.0, 1.0, ni, nj, 1.01, 1.03)
        u = np.zeros((ni + 2, nj + 2))
        u[1:-1, 1:-1] = rng.standard_normal((ni, nj)) + 0.3
        dev = Device(ni, nj)
        dev.upload_mesh(mesh)
        dev.upload(L.F_U, u)
        want = dev.bulk_velocity(L.F_U)
        pitch = dev.pitch + pad
        dev.close()
        f_pl, w_pl = _planes(torch, ni, nj, pitch, ng, mesh["cell_volume"], u[1:-1, 1:-1], lib.cfd_col_offset())
        out = torch.full((3,), float("nan"), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        user = torch.cuda.Stream() if own_stream else torch.cuda.current_stream()
        stream = user.cuda_stream
        assert not own_stream or stream != torch.cuda.current_stream().cuda_stream
        L.check(lib.cfd_bulk_velocity_noalloc(C.c_void_p(f_pl.data_ptr()), C.c_void_p(w_pl.data_ptr()), ni, nj, pitch,
                                              C.c_void_p(stream), C.c_void_p(out.data_ptr())))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert got[0] == want, (ni, nj, got, want)                      # same kernels, same order: same bits
        ref = float((u[1:-1, 1:-1] * mesh["cell_volume"]).sum() / mesh["cell_volume"].sum())
        assert abs(got[0] - ref) <= 1e-13 * abs(ref)
        assert abs(got[2] - mesh["cell_volume"].sum()) <= 1e-13 * got[2]
        keep.append((f_pl, w_pl, out, user))
    # sizes beyond the stated maximum are rejected with a clear error, nothing is launched
    rc = lib.cfd_bulk_velocity_noalloc(C.c_void_p(keep[0][0].data_ptr()), C.c_void_p(keep[0][1].data_ptr()), 40000, 8, 40032,
                                       None, C.c_void_p(keep[0][2].data_ptr()))
    assert rc != 0 and b"32768" in lib.cfd_last_error()
