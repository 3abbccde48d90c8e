"""CPU tests of the j-slab layer: partition, table slicing, and the three
collectives of cfd_amd.slab.SlabComm run by two gloo ranks on host tensors (the same code that
moves device views over RCCL; here the 'workspace' is a host tensor)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def test_partition():
    from cfd_amd.slab import partition
    assert partition(64, 1, 0) == (0, 64)
    assert partition(64, 4, 3) == (48, 16)
    assert partition(7, 1, 0) == (0, 7)            # single rank: no constraint
    with pytest.raises(ValueError):
        partition(66, 4, 0)                         # not divisible
    with pytest.raises(ValueError):
        partition(12, 4, 0)                         # odd slabs
    with pytest.raises(ValueError):
        partition(64, 4, 4)


def test_default_replication_threshold_follows_the_global_level_size():
    """A multigrid level is replicated once the global level has <= 2^18 cells: 8192^2 keeps 4 slab-distributed levels
    whatever the rank count (the round-2 constant 512 replicated a 4096 x 2048 level on every one of 8 ranks)."""
    from cfd_amd.slab import default_min_slab_rows
    assert default_min_slab_rows(8192, 8192, 2) == 512
    assert default_min_slab_rows(8192, 8192, 4) == 256
    assert default_min_slab_rows(8192, 8192, 8) == 128
    assert default_min_slab_rows(64, 256, 8) == 32            # small mesh: never distributed below the finest level
    assert default_min_slab_rows(2048, 2048, 2) == 512
    assert default_min_slab_rows(8192, 8192 * 8, 8) == 256     # weak-scaling channel 8192 x 65536: 6 distributed levels


def test_boundary_tables_slab():
    """Slab tables = the global tables restricted to the slab rows; k/omega ops re-based."""
    from cfd_amd import _lib as L
    from cfd_amd.bc import BoundaryGeometry, BoundaryTables
    from cfd_amd.mesh import rect_node_axes
    ni, nj = 6, 16
    x, y = rect_node_axes(length_i=2.0, length_j=1.0, count_i=ni, count_j=nj, ratio_i=1.0, ratio_j=1.1)
    geo = BoundaryGeometry.from_rect_axes(x, y)
    bc = {"i_min": [{"type": "inlet", "range": (0, 9), "velocity": (1.0, 0.0), "k": 1e-3, "omega": 5.0},
                    {"type": "wall", "range": (10, 15)}],
          "i_max": {"type": "outlet", "range": (0, 15), "pressure": 0.0},
          "j_min": {"type": "wall", "range": (0, 5)}, "j_max": {"type": "symmetry", "range": (0, 5)}}
    full = BoundaryTables(bc, ni, nj, geo, 1e-3, True)
    parts = [full.slab(j0, 4) for j0 in (0, 4, 8, 12)]
    for side in (0, 1):
        for key in full.flow[side]:
            assert np.array_equal(np.concatenate([p.flow[side][key] for p in parts]), full.flow[side][key])
        assert np.array_equal(np.concatenate([p.p_masks[side] for p in parts]), full.p_masks[side])
    # replay the k/omega programs: ops of all slabs together touch exactly what the global program touches
    def touched(tab, j0, njl):
        out = {}
        for (op, field, side, lo, hi, off, a) in tab.turb_ops:
            if op in (L.TOP_WRAP_I, L.TOP_WRAP_J):
                continue
            for q in range(lo, hi + 1):
                idx = q + j0 if side < 2 else q
                out.setdefault((op, field, side, idx), []).append((a, tab.turb_values[off + q - lo]))
        return out
    want = touched(full, 0, nj)
    got = {}
    for p, j0 in zip(parts, (0, 4, 8, 12)):
        for k, v in touched(p, j0, 4).items():
            got.setdefault(k, []).extend(v)
    assert got == want
    assert all(o[2] != 2 for o in parts[1].turb_ops + parts[2].turb_ops + parts[3].turb_ops)   # j_min only on the first
    assert all(o[2] != 3 for o in parts[0].turb_ops + parts[1].turb_ops + parts[2].turb_ops)   # j_max only on the last
    with pytest.raises(NotImplementedError):
        BoundaryTables({"j_min": {"type": "periodic", "range": (0, 5)}, "j_max": {"type": "periodic", "range": (0, 5)},
                        "i_min": {"type": "wall", "range": (0, 15)}, "i_max": {"type": "wall", "range": (0, 15)}},
                       ni, nj, geo, 1e-3, False).slab(0, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _comm_worker(rank, world, port, out_q):
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cfd_amd import _lib as L
        from cfd_amd.slab import SlabComm, gather_slabs, partition, slab_of
        pitch, njl, nplanes = 32, 6, 3
        ng = L.GHOST_ROWS
        rows = njl + 2 * ng
        ws = torch.zeros((nplanes * rows * pitch + 64) * 8, dtype=torch.uint8)
        comm = SlabComm(ws, rank, world)
        f = comm.f64
        planes = [comm.base + 8 * q * rows * pitch for q in range(nplanes)]

        def fill():
            for q in range(nplanes):
                v = f[q * rows * pitch:(q + 1) * rows * pitch].view(rows, pitch)
                v.fill_(-1.0)
                for r in range(ng, ng + njl):        # interior row j = r - ng -> code rank, plane, global row
                    v[r] = 1000.0 * rank + 100.0 * q + (rank * njl + r - ng) + 0.001 * torch.arange(pitch, dtype=torch.float64)
        depths = tuple(range(1, ng + 1))
        for depth in depths:
            fill()
            comm.halo(planes, pitch, njl, depth)
            for q in range(nplanes):
                v = f[q * rows * pitch:(q + 1) * rows * pitch].view(rows, pitch)
                col = 0.001 * torch.arange(pitch, dtype=torch.float64)
                for d in range(1, depth + 1):
                    lo, hi = v[ng - d], v[ng + njl + d - 1]
                    if rank > 0:
                        assert torch.equal(lo, 1000.0 * (rank - 1) + 100.0 * q + (rank * njl - d) + col), (depth, d)
                    else:
                        assert torch.all(lo == -1.0)
                    if rank < world - 1:
                        assert torch.equal(hi, 1000.0 * (rank + 1) + 100.0 * q + ((rank + 1) * njl + d - 1) + col)
                    else:
                        assert torch.all(hi == -1.0)
                if depth == 1:
                    assert torch.all(v[0] == -1.0) and torch.all(v[rows - 1] == -1.0)     # outer halo rows untouched
        # all-reduce sum / min on scalars at the end of the workspace
        s_off = nplanes * rows * pitch
        f[s_off:s_off + 3] = torch.tensor([1.0 + rank, 10.0 * (rank + 1), -float(rank)], dtype=torch.float64)
        comm.all_reduce(comm.base + 8 * s_off, 3, 0)
        assert f[s_off:s_off + 3].tolist() == [sum(1.0 + r for r in range(world)), sum(10.0 * (r + 1) for r in range(world)),
                                               -float(sum(range(world)))]
        f[s_off:s_off + 2] = torch.tensor([5.0 - rank, 2.0 + rank], dtype=torch.float64)
        comm.all_reduce(comm.base + 8 * s_off, 2, 1)
        assert f[s_off:s_off + 2].tolist() == [5.0 - (world - 1), 2.0]
        # all-gather of slab rows into a replicated plane (plane 2 used as destination)
        fill()
        src = planes[0] + 8 * ng * pitch
        dst = planes[2] + 8 * ng * pitch
        n_each = 2 * pitch                              # 2 rows per rank -> 2 * world rows <= njl
        comm.all_gather(src, dst, n_each)
        g = f[2 * rows * pitch:3 * rows * pitch].view(rows, pitch)
        for r in range(world):
            for k in range(2):
                assert g[ng + 2 * r + k, 0].item() == 1000.0 * r + (r * njl + k)
        assert (comm.n_halo, comm.n_reduce, comm.n_gather) == (len(depths), 2, 1)
        # out-of-workspace pointers are refused, and a failing callback reports instead of raising through C
        assert comm._reduce_cb(None, comm.base - 8, 1, 0) == 1 and isinstance(comm.error, ValueError)
        L.LAST_CALLBACK_ERROR = None
        # host slicing / gathering round trip for every array kind
        rng = np.random.default_rng(5)
        ni, njg = 5, njl * world
        j0, nl = partition(njg, world, rank) if njl % 2 == 0 else (rank * njl, njl)
        for kind, shape in ((L.KIND_GHOSTED, (ni + 2, njg + 2)), (L.KIND_CELL, (ni, njg)),
                            (L.KIND_FACE_I, (ni + 1, njg)), (L.KIND_FACE_J, (ni, njg + 1))):
            whole = rng.standard_normal(shape)
            mine = slab_of(whole, j0, nl, kind)
            back = gather_slabs(mine, kind, nl, world)
            assert np.array_equal(back, whole), kind
        dist.barrier()
        dist.destroy_process_group()
        out_q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001
        import traceback
        out_q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_slab_collectives_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_comm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = []
    for _ in procs:
        results.append(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        if p.is_alive():
            p.kill()
    bad = [r for r in results if r[1] != "ok"]
    assert not bad, "\n".join(f"rank {r}: {msg}" for r, msg in bad)


def test_rehearsal_env_gives_every_rank_a_host_of_its_own(monkeypatch):
    """One-GPU rehearsal of the RCCL transports: distinct NCCL_HOSTID per rank, loopback sockets, user settings kept."""
    import os
    from cfd_amd.slab import rehearsal_env
    monkeypatch.delenv("NCCL_HOSTID", raising=False)
    monkeypatch.delenv("NCCL_SOCKET_IFNAME", raising=False)
    monkeypatch.setenv("NCCL_IB_DISABLE", "0")
    ids = set()
    for rank in range(4):
        rehearsal_env(rank)
        ids.add(os.environ["NCCL_HOSTID"])
    assert len(ids) == 4
    assert os.environ["NCCL_SOCKET_IFNAME"] == "lo" and os.environ["NCCL_IB_DISABLE"] == "0"


def test_foreign_plane_describes_raw_device_doubles():
    """Planes the library allocated itself (SST caches) reach torch through __cuda_array_interface__ (no copy)."""
    from cfd_amd.slab import _ForeignPlane
    cai = _ForeignPlane(0x7F0000001000, 1234).__cuda_array_interface__
    assert cai["shape"] == (1234,) and cai["typestr"] == "<f8" and cai["data"] == (0x7F0000001000, False) and cai["version"] == 2


def test_abi_declares_the_rccl_transport():
    """The native transport entry points are part of the C ABI (and of the loaded library)."""
    from pathlib import Path
    header = (Path(__file__).resolve().parents[1] / "include" / "cfd_hip.h").read_text()
    for name in ("cfd_rccl_unique_id", "cfd_set_comm_rccl", "cfd_comm_counts", "cfd_comm_counts_by_phase"):
        assert f"int {name}(" in header
    assert "#define CFD_RCCL_ID_BYTES 128" in header
    from cfd_amd import _lib
    assert _lib.RCCL_ID_BYTES == 128
