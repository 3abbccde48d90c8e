"""j-slab decomposition of the outer step over several GPUs.

One process per GPU; rank r owns rows [r * nj_local, (r + 1) * nj_local) of the global mesh.
The library (csrc/comm.hip) needs three collectives (halo rows, all-reduce, all-gather).  Transports:

  * "rccl" (default when the process group's backend is nccl): the library's own RCCL communicator --
    the collectives are enqueued from C++ on the context's stream, torch.distributed only carries the
    128-byte unique id to the ranks once;
  * "torch": C callbacks into this module, implemented with torch.distributed -- backend "nccl" on
    zero-copy views of the device workspace, backend "gloo" with the same views staged through host
    memory (how the N > 1 logic is tested without RCCL, and on CPU tensors without any GPU).

On a one-GPU box several ranks can rehearse the RCCL transports on the same device when every rank poses
as a host of its own (`rehearsal_env`): RCCL then talks over its loopback socket transport.

The reference has no counterpart: its run_case (solver.py:1367-1451) is single-process.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def partition(nj_global, nranks, rank):
    """Rows of rank `rank`: equal slabs, even row count (2x multigrid aggregates never straddle a cut)."""
    nj_global, nranks, rank = int(nj_global), int(nranks), int(rank)
    if nranks < 1 or not 0 <= rank < nranks:
        raise ValueError("bad rank / nranks")
    if nj_global % nranks:
        raise ValueError(f"Nj = {nj_global} is not divisible by the {nranks} ranks")
    nj_local = nj_global // nranks
    if nranks > 1 and (nj_local % 2 or nj_local < 4):
        raise ValueError(f"j-slabs need an even number (>= 4) of rows per rank, got {nj_local}")
    return rank * nj_local, nj_local


class _ForeignPlane:
    """n device doubles at a raw address, for torch.as_tensor (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class SlabComm:
    """The three collectives on views of one flat workspace tensor.

    ws: torch uint8 tensor (device or host) that contains every pointer the library will pass.
    """

    def __init__(self, ws, rank, nranks, group=None, stream=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ws = ws
        self.base = ws.data_ptr()
        self.f64 = ws.view(torch.float64)
        self.rank, self.nranks = int(rank), int(nranks)
        self.group = group
        self.stream = stream
        backend = dist.get_backend(group) if self.nranks > 1 else "none"
        # gloo moves host memory only: stage device views through the host
        self.staged = ws.is_cuda and backend != "nccl"
        self.error = None
        self._plans = {}
        self.n_halo = self.n_reduce = self.n_gather = 0
        self._halo_c = _lib.HALO_FN(self._halo_cb)
        self._reduce_c = _lib.ALLREDUCE_FN(self._reduce_cb)
        self._gather_c = _lib.ALLGATHER_FN(self._gather_cb)

    # ------------------------------------------------------------------ views
    def view(self, ptr, n):
        off = int(ptr) - self.base
        if off < 0 or off % 8 or off + 8 * int(n) > self.ws.numel():
            if self.ws.is_cuda and int(ptr) % 8 == 0:
                # a device plane the library allocated itself (the SST caches): zero-copy view of foreign memory
                return self.torch.as_tensor(_ForeignPlane(int(ptr), int(n)), device=self.ws.device)
            raise ValueError("pointer outside the workspace")
        return self.f64[off // 8: off // 8 + int(n)]

    def _peer(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _ctx(self):
        import contextlib
        if self.stream is not None and self.ws.is_cuda:
            return self.torch.cuda.stream(self.stream)
        return contextlib.nullcontext()

    # ------------------------------------------------------------------ collectives
    def _halo_plan(self, plane_ptrs, pitch, nj_local, depth):
        """(send tensors, recv tensors, P2P ops) of one exchange; cached, the library repeats few patterns."""
        key = (tuple(plane_ptrs), pitch, nj_local, depth)
        plan = self._plans.get(key)
        if plan is not None:
            return plan
        dist = self.dist
        n = depth * pitch
        ng = _lib.GHOST_ROWS
        up, dn = self.rank + 1, self.rank - 1
        sends, recvs = [], []   # (tensor, peer)
        for p in plane_ptrs:
            def row(r, p=p):
                return self.view(p + 8 * r * pitch, n)
            if up < self.nranks:
                sends.append((row(ng + nj_local - depth), up))
                recvs.append((row(ng + nj_local), up))
            if dn >= 0:
                sends.append((row(ng), dn))
                recvs.append((row(ng - depth), dn))
        ops = None
        if not self.staged:
            ops = [dist.P2POp(dist.isend, t, self._peer(peer), self.group) for t, peer in sends]
            ops += [dist.P2POp(dist.irecv, t, self._peer(peer), self.group) for t, peer in recvs]
        plan = (sends, recvs, ops)
        if len(self._plans) < 4096:
            self._plans[key] = plan
        return plan

    def halo(self, plane_ptrs, pitch, nj_local, depth):
        """Rows [nj-d, nj) -> rows [-d, 0) of rank+1 ; rows [0, d) -> rows [nj, nj+d) of rank-1.
        Plane row r holds j = r - GHOST_ROWS."""
        dist = self.dist
        sends, recvs, ops = self._halo_plan(plane_ptrs, pitch, nj_local, depth)
        if not sends:
            return
        with self._ctx():
            if self.staged:
                n = depth * pitch
                s_host = [(t.cpu(), peer) for t, peer in sends]
                r_host = [(self.torch.empty(n, dtype=self.torch.float64), peer) for _, peer in recvs]
                ops = [dist.P2POp(dist.isend, t, self._peer(peer), self.group) for t, peer in s_host]
                ops += [dist.P2POp(dist.irecv, t, self._peer(peer), self.group) for t, peer in r_host]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
            if self.staged:
                for (dst, _), (src, _) in zip(recvs, r_host):
                    dst.copy_(src)
        self.n_halo += 1

    def all_reduce(self, ptr, n, op):
        dist = self.dist
        t = self.view(ptr, n)
        rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MIN
        with self._ctx():
            if self.staged:
                h = t.cpu()
                dist.all_reduce(h, op=rop, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=rop, group=self.group)
        self.n_reduce += 1

    def all_gather(self, src_ptr, dst_ptr, elems):
        dist = self.dist
        src = self.view(src_ptr, elems)
        dst = self.view(dst_ptr, elems * self.nranks)
        with self._ctx():
            if self.staged:
                h = src.cpu()
                parts = [self.torch.empty_like(h) for _ in range(self.nranks)]
                dist.all_gather(parts, h, group=self.group)
                dst.copy_(self.torch.cat(parts))
            elif self.ws.is_cuda:
                dist.all_gather_into_tensor(dst, src, group=self.group)
            else:
                parts = [dst[r * elems:(r + 1) * elems] for r in range(self.nranks)]
                dist.all_gather(parts, src.clone(), group=self.group)
        self.n_gather += 1

    # ------------------------------------------------------------------ C callbacks
    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except BaseException as e:      # must not propagate through the C frames
            self.error = e
            _lib.LAST_CALLBACK_ERROR = e
            return 1

    def _halo_cb(self, user, nplanes, planes, pitch, nj_local, depth):
        return self._guard(self.halo, [int(planes[q]) for q in range(nplanes)], pitch, nj_local, depth)

    def _reduce_cb(self, user, dev, n, op):
        return self._guard(self.all_reduce, int(dev), n, op)

    def _gather_cb(self, user, src, dst, elems):
        return self._guard(self.all_gather, int(src), int(dst), int(elems))

    def install(self, lib, handle):
        check(lib.cfd_set_comm(handle, self._halo_c, self._reduce_c, self._gather_c, None))


# A multigrid level is replicated on every rank (operator gathered once, right-hand side gathered once per V-cycle, no
# further messages) once the GLOBAL level has this few cells: such a level and everything below it is launch-latency
# work on one GPU (<= 2 MiB gathered), while keeping it distributed costs two exchanges per V-cycle.  Above it a level
# stays distributed: replicating a level of millions of cells would make every rank do the whole level's work.
REPLICATE_BELOW_CELLS = 1 << 18
DEFAULT_MIN_SLAB_ROWS = 512      # 8192^2 on two ranks under the rule above; kept as the fall-back of callers without a mesh size


def default_min_slab_rows(ni, nj_global, world):
    """Rows per slab of the coarsest level that stays slab-distributed under REPLICATE_BELOW_CELLS (>= 16, the library's
    floor; slab levels halve the rows while they stay even): 512 for 8192^2 on 2 ranks, 256 on 4, 128 on 8."""
    rows, ci, cj = int(nj_global) // int(world), int(ni), int(nj_global)
    while rows % 2 == 0 and rows // 2 >= 16 and ((ci + 1) // 2) * (cj // 2) > REPLICATE_BELOW_CELLS:
        rows, ci, cj = rows // 2, (ci + 1) // 2, cj // 2
    return max(16, rows)


def rehearsal_env(rank):
    """Several ranks on ONE GPU (functional rehearsal only): RCCL refuses two ranks of one host on one device,
    so each rank gets a host id of its own and RCCL uses its socket transport over loopback.  Call before the
    first RCCL communicator is created (torch's or the library's)."""
    import os
    os.environ["NCCL_HOSTID"] = f"rehearsal-host-{int(rank)}"
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")


def _rccl_probe(dev):
    """None if the library can reach RCCL on this rank (dlopen + ncclGetUniqueId), else the error text."""
    try:
        check(dev.lib.cfd_rccl_unique_id(C.create_string_buffer(_lib.RCCL_ID_BYTES), _lib.RCCL_ID_BYTES))
    except Exception as e:      # noqa: BLE001
        return str(e)
    return None


def attach_rccl(dev, group=None):
    """Give the slab context the library's own RCCL communicator (collective over `group`)."""
    import torch.distributed as dist
    ident = [None]
    if dev.rank == 0:
        # a failure here is broadcast like the id would be: no rank is left waiting in the collective below
        try:
            buf = C.create_string_buffer(_lib.RCCL_ID_BYTES)
            check(dev.lib.cfd_rccl_unique_id(buf, _lib.RCCL_ID_BYTES))
            ident[0] = buf.raw
        except Exception as e:      # noqa: BLE001
            ident[0] = ("failed", str(e))
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast_object_list(ident, src=src, group=group)
    if isinstance(ident[0], tuple):
        raise _lib.HipError("rank 0 could not create the RCCL id: " + ident[0][1])
    buf = C.create_string_buffer(ident[0], _lib.RCCL_ID_BYTES)
    check(dev.lib.cfd_set_comm_rccl(dev._h, buf, _lib.RCCL_ID_BYTES))


def slab_device(ni, nj_global, device=0, group=None, min_slab_rows=None, transport=None):
    """Device context of this rank's slab; torch.distributed must be initialised when world > 1.
    min_slab_rows: multigrid levels stay slab-distributed while a slab keeps at least this many rows
    (3 halo exchanges per level and V-cycle); coarser levels are replicated on every rank.
    transport: "rccl" (library-native RCCL), "torch" (torch.distributed callbacks); default: environment
    variable CFD_SLAB_TRANSPORT, else "rccl" under an nccl process group and "torch" otherwise."""
    import torch
    import torch.distributed as dist
    from .device import Device
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if min_slab_rows is None:
        min_slab_rows = default_min_slab_rows(ni, nj_global, world)
    min_slab_rows = max(16, int(min_slab_rows))
    rank = dist.get_rank(group) if world > 1 else 0
    j_off, nj_local = partition(nj_global, world, rank)
    if world == 1:
        dev = Device(ni, nj_global, device)
        dev.j_offset, dev.nj_global, dev.rank, dev.nranks = 0, int(nj_global), 0, 1
        return dev
    lib = _lib.load()
    need = lib.cfd_workspace_bytes_slab(int(ni), int(nj_local), int(world), min_slab_rows)
    if need == 0:
        raise _lib.HipError("cfd_workspace_bytes_slab rejected the slab shape")
    torch.cuda.set_device(device)
    ws = torch.empty(int(need), dtype=torch.uint8, device=f"cuda:{device}")
    if ws.data_ptr() % 256:
        raise _lib.HipError("torch workspace is not 256-byte aligned")
    stream = torch.cuda.Stream(device=device)
    dev = Device.__new__(Device)
    dev.lib = lib
    dev.ni, dev.nj = int(ni), int(nj_local)
    dev._h = C.c_void_p()
    dev.turbulent = False
    check(lib.cfd_create_slab(C.byref(dev._h), dev.ni, dev.nj, int(device), C.c_void_p(ws.data_ptr()), int(need),
                              int(rank), int(world), min_slab_rows))
    check(lib.cfd_set_stream(dev._h, C.c_void_p(stream.cuda_stream)))
    comm = SlabComm(ws, rank, world, group=group, stream=stream)
    dev._ws, dev._stream, dev.comm = ws, stream, comm     # keep alive as long as the context
    dev.j_offset, dev.nj_global, dev.rank, dev.nranks = j_off, int(nj_global), rank, world
    import os
    asked = transport or os.environ.get("CFD_SLAB_TRANSPORT")
    transport = asked or ("rccl" if dist.get_backend(group) == "nccl" else "torch")
    dev.transport_note = None
    if transport == "rccl" and not asked:
        # Default choice: if the library's own communicator cannot be built, all ranks agree to use the
        # torch.distributed callbacks instead (same RCCL underneath, host code between the kernels) and say so.
        # Two agreement points, so that no rank is left alone inside a collective: RCCL loadable by the library on
        # every rank (before the id is broadcast), then the communicator built on every rank.
        def all_ok(err):
            flag = torch.tensor([0 if err else 1], dtype=torch.int32, device=f"cuda:{device}")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            return int(flag.item()) == 1
        err = _rccl_probe(dev)
        if all_ok(err):
            try:
                attach_rccl(dev, group)
            except Exception as e:      # noqa: BLE001 - any failure of the native path leads to the same fall-back
                err = str(e)
            native = all_ok(err)
        else:
            native = False
        if not native:
            comm.install(lib, dev._h)       # cfd_set_comm releases a communicator this rank may have built
            transport = "torch"
            dev.transport_note = "library-native RCCL communicator not available (%s): torch.distributed callbacks" % \
                (err or "failed on another rank")
    elif transport == "rccl":
        attach_rccl(dev, group)
    elif transport == "torch":
        comm.install(lib, dev._h)
    else:
        raise ValueError(f"unknown slab transport {transport!r} (rccl | torch)")
    dev.transport = transport
    return dev


def slab_of(arr, j_offset, nj_local, kind=_lib.KIND_GHOSTED):
    """Slice of a global host array (reference layout, j = axis 1) that the slab context takes."""
    extra = {_lib.KIND_GHOSTED: 2, _lib.KIND_CELL: 0, _lib.KIND_FACE_I: 0, _lib.KIND_FACE_J: 1}[kind]
    return np.ascontiguousarray(np.asarray(arr)[:, j_offset:j_offset + nj_local + extra])


MESH_KINDS = {"cell_volume": _lib.KIND_CELL, "w_right_i": _lib.KIND_FACE_I, "w_left_i": _lib.KIND_FACE_I,
              "w_north_j": _lib.KIND_FACE_J, "w_south_j": _lib.KIND_FACE_J, "face_normal_i": _lib.KIND_FACE_I,
              "face_normal_j": _lib.KIND_FACE_J, "diff_d_i": _lib.KIND_FACE_I, "diff_d_j": _lib.KIND_FACE_J,
              "cell_spacing_i": _lib.KIND_CELL, "cell_spacing_j": _lib.KIND_CELL}


def slab_mesh(mesh, j_offset, nj_local):
    """The metric arrays Device.upload_mesh reads, restricted to one slab (cut faces keep their
    interior values because the global arrays were computed on the whole mesh)."""
    return {k: slab_of(mesh[k], j_offset, nj_local, kind) for k, kind in MESH_KINDS.items() if k in mesh}


def gather_slabs(mine, kind, nj_local, nranks, group=None):
    """Assemble the global host array from every rank's slab array (collective, all ranks get it)."""
    import torch.distributed as dist
    parts = [None] * nranks
    dist.all_gather_object(parts, mine, group=group)
    if kind == _lib.KIND_GHOSTED:
        # interior rows from their owner, the two physical ghost rows from the first / last rank
        body = [p[:, 1:-1] for p in parts]
        return np.concatenate([parts[0][:, :1]] + body + [parts[-1][:, -1:]], axis=1)
    if kind == _lib.KIND_FACE_J:
        return np.concatenate([p[:, :nj_local] for p in parts] + [parts[-1][:, nj_local:]], axis=1)
    return np.concatenate(parts, axis=1)
