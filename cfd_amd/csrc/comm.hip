// j-slab communication.  Two transports behind the same three shims (halo rows, all-reduce, all-gather),
// both ordered on Ctx::stream; no-ops on a single rank:
//   * native RCCL (cfd_set_comm_rccl): grouped ncclSend / ncclRecv to the two neighbours over xGMI,
//     ncclAllReduce, ncclAllGather, enqueued by the library itself -- no host code between the kernels;
//   * host callbacks (cfd_set_comm): whatever the embedding application uses (torch.distributed in
//     cfd_amd/slab.py; host-staged gloo where RCCL is not available).
// RCCL is resolved at run time (dlopen), so single-GPU users need no RCCL on the machine.
#include "internal.h"

#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace cfd {

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional: only cfd_rccl_ranks asks for it
};

RcclApi g_rccl;

template <class F>
bool bind(F &fn, const char *name) {
    fn = (F)dlsym(g_rccl.lib, name);
    if (!fn) set_error("RCCL library lacks %s", name);
    return fn != nullptr;
}

// The copy of RCCL the process already holds (e.g. the one PyTorch brought) is preferred over a second one.
int load_rccl() {
    if (g_rccl.lib) return CFD_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t q = 0; !h && q < sizeof(names) / sizeof(names[0]); ++q) h = dlopen(names[q], RTLD_NOW | RTLD_LOCAL);
    if (!h) { set_error("cannot load RCCL (librccl.so.1): %s", dlerror()); return CFD_E_STATE; }
    g_rccl.lib = h;
    const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                    bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.GroupStart, "ncclGroupStart") &&
                    bind(g_rccl.GroupEnd, "ncclGroupEnd") && bind(g_rccl.Send, "ncclSend") && bind(g_rccl.Recv, "ncclRecv") &&
                    bind(g_rccl.AllReduce, "ncclAllReduce") && bind(g_rccl.AllGather, "ncclAllGather") &&
                    bind(g_rccl.GetErrorString, "ncclGetErrorString");
    if (!ok) { g_rccl = RcclApi{}; return CFD_E_STATE; }
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.lib, "ncclCommCount");
    return CFD_OK;
}

int rccl_fail(const char *what, ncclResult_t r) {
    set_error("%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return CFD_E_STATE;
}
#define CFD_RCCL(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) return rccl_fail(#call, r_); } while (0)

int need_comm(const Ctx *c, const void *fn, const char *what) {
    if (fn) return CFD_OK;
    set_error("j-slab context (%d ranks) has no %s transport: call cfd_set_comm_rccl or cfd_set_comm first", c->comm.nranks, what);
    return CFD_E_STATE;
}

}  // namespace

int rccl_unique_id(void *id, size_t bytes) {
    if (!id || bytes < sizeof(ncclUniqueId)) { set_error("cfd_rccl_unique_id: buffer of %zu bytes needed", sizeof(ncclUniqueId)); return CFD_E_ARG; }
    CFD_TRY(load_rccl());
    ncclUniqueId u;
    CFD_RCCL(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
    return CFD_OK;
}

// Ranks of the library's own communicator as RCCL itself reports them (0: no native communicator): what a
// benchmark line quotes as proof that N processes really joined one communicator.
int rccl_ranks(Ctx *c, int *ranks) {
    *ranks = 0;
    if (!c->comm.rccl) return CFD_OK;
    if (!g_rccl.CommCount) { set_error("RCCL library lacks ncclCommCount"); return CFD_E_STATE; }
    CFD_RCCL(g_rccl.CommCount((ncclComm_t)c->comm.rccl, ranks));
    return CFD_OK;
}

void rccl_release(Ctx *c) {
    if (c->comm.rccl && g_rccl.CommDestroy) g_rccl.CommDestroy((ncclComm_t)c->comm.rccl);
    c->comm.rccl = nullptr;
}

// Collective over the ranks of the slab decomposition: every rank passes the id rank 0 generated.
int rccl_attach(Ctx *c, const void *id, size_t bytes) {
    if (c->comm.nranks <= 1) return CFD_OK;
    if (!id || bytes < sizeof(ncclUniqueId)) { set_error("cfd_set_comm_rccl: id of %zu bytes needed", sizeof(ncclUniqueId)); return CFD_E_ARG; }
    CFD_TRY(load_rccl());
    CFD_HIP(hipSetDevice(c->device));
    rccl_release(c);
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    CFD_RCCL(g_rccl.CommInitRank(&comm, c->comm.nranks, u, c->comm.rank));
    c->comm.rccl = comm;
    c->comm.halo = nullptr; c->comm.allreduce = nullptr; c->comm.allgather = nullptr; c->comm.user = nullptr;
    return CFD_OK;
}

// Rows [nj-d, nj) -> rows [-d, 0) of rank+1 ; rows [0, d) -> rows [nj, nj+d) of rank-1 (plane row r holds j = r - NG).
// Every plane's four messages go into one RCCL group: one launch per exchange, whatever the planes' geometries.
int halo_exchange_multi(Ctx *c, int n, const HaloSpec *specs) {
    if (c->comm.nranks <= 1 || n <= 0) return CFD_OK;
    for (int q = 0; q < n; ++q) {
        const Geo &g = *specs[q].g;
        if (!(g.cutlo || g.cuthi)) { set_error("halo exchange on a level without cuts"); return CFD_E_ARG; }
        if (specs[q].depth < 1 || specs[q].depth > NG || specs[q].depth > g.nj || !specs[q].plane) {
            set_error("halo depth %d not possible on %d rows", specs[q].depth, g.nj);
            return CFD_E_ARG;
        }
    }
    if (c->comm.rccl) {
        ++c->comm.n_halo; ++c->comm.by_phase[c->comm.phase][0];
        ncclComm_t comm = (ncclComm_t)c->comm.rccl;
        const int up = c->comm.rank + 1, dn = c->comm.rank - 1;
        CFD_RCCL(g_rccl.GroupStart());
        ncclResult_t r = ncclSuccess;
        for (int q = 0; q < n && r == ncclSuccess; ++q) {
            const Geo &g = *specs[q].g;
            const size_t P = (size_t)g.pitch, cnt = (size_t)specs[q].depth * P;
            const int depth = specs[q].depth;
            double *x = specs[q].plane;
            if (up < c->comm.nranks) {
                r = g_rccl.Send(x + (size_t)(NG + g.nj - depth) * P, cnt, ncclDouble, up, comm, c->stream);
                if (r == ncclSuccess) r = g_rccl.Recv(x + (size_t)(NG + g.nj) * P, cnt, ncclDouble, up, comm, c->stream);
            }
            if (dn >= 0 && r == ncclSuccess) {
                r = g_rccl.Send(x + (size_t)NG * P, cnt, ncclDouble, dn, comm, c->stream);
                if (r == ncclSuccess) r = g_rccl.Recv(x + (size_t)(NG - depth) * P, cnt, ncclDouble, dn, comm, c->stream);
            }
        }
        const ncclResult_t re = g_rccl.GroupEnd();   // always closed, also after a failed call inside
        if (r != ncclSuccess) return rccl_fail("ncclSend/ncclRecv (halo rows)", r);
        if (re != ncclSuccess) return rccl_fail("ncclGroupEnd (halo rows)", re);
        return CFD_OK;
    }
    // host callbacks take planes of one geometry and depth per call
    CFD_TRY(need_comm(c, (const void *)c->comm.halo, "halo"));
    for (int q = 0; q < n;) {
        double *pl[4];
        int m = 0;
        const Geo &g = *specs[q].g;
        const int depth = specs[q].depth;
        while (q < n && m < 4 && specs[q].depth == depth && specs[q].g->pitch == g.pitch && specs[q].g->nj == g.nj) pl[m++] = specs[q++].plane;
        ++c->comm.n_halo; ++c->comm.by_phase[c->comm.phase][0];
        if (c->comm.halo(c->comm.user, m, pl, g.pitch, g.nj, depth) != 0) {
            set_error("halo exchange callback failed");
            return CFD_E_STATE;
        }
    }
    return CFD_OK;
}

int halo_exchange(Ctx *c, const Geo &g, int depth, double *a, double *b, double *d, double *e) {
    if (c->comm.nranks <= 1 || !(g.cutlo || g.cuthi)) return CFD_OK;
    HaloSpec sp[4] = {{&g, depth, a}, {&g, depth, b}, {&g, depth, d}, {&g, depth, e}};
    int n = 0;
    while (n < 4 && sp[n].plane) ++n;
    return halo_exchange_multi(c, n, sp);
}

int all_reduce(Ctx *c, double *dev, int n, int op) {
    if (c->comm.nranks <= 1) return CFD_OK;
    ++c->comm.n_reduce; ++c->comm.by_phase[c->comm.phase][1];
    if (c->comm.rccl) {
        CFD_RCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMin, (ncclComm_t)c->comm.rccl, c->stream));
        return CFD_OK;
    }
    CFD_TRY(need_comm(c, (const void *)c->comm.allreduce, "all-reduce"));
    if (c->comm.allreduce(c->comm.user, dev, n, op) != 0) { set_error("all-reduce callback failed"); return CFD_E_STATE; }
    return CFD_OK;
}

int all_gather_rows(Ctx *c, const double *src, double *dst, size_t elems_each) {
    if (c->comm.nranks <= 1) {
        CFD_HIP(hipMemcpyAsync(dst, src, elems_each * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return CFD_OK;
    }
    ++c->comm.n_gather; ++c->comm.by_phase[c->comm.phase][2];
    if (c->comm.rccl) {
        CFD_RCCL(g_rccl.AllGather(src, dst, elems_each, ncclDouble, (ncclComm_t)c->comm.rccl, c->stream));
        return CFD_OK;
    }
    CFD_TRY(need_comm(c, (const void *)c->comm.allgather, "all-gather"));
    if (c->comm.allgather(c->comm.user, src, dst, elems_each) != 0) { set_error("all-gather callback failed"); return CFD_E_STATE; }
    return CFD_OK;
}

}  // namespace cfd
