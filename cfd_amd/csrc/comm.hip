// j-slab communication shims: thin, checked wrappers over the host-supplied callbacks
// (see include/cfd_hip.h, cfd_set_comm).  No-ops on a single rank.
#include "internal.h"

namespace cfd {

static int need_comm(const Ctx *c, const void *fn, const char *what) {
    if (fn) return CFD_OK;
    set_error("j-slab context (%d ranks) has no %s callback: call cfd_set_comm first", c->comm.nranks, what);
    return CFD_E_STATE;
}

int halo_exchange(Ctx *c, const Geo &g, int depth, double *a, double *b, double *d, double *e) {
    if (c->comm.nranks <= 1 || !(g.cutlo || g.cuthi)) return CFD_OK;
    CFD_TRY(need_comm(c, (const void *)c->comm.halo, "halo"));
    if (depth < 1 || depth > NG || depth > g.nj) { set_error("halo depth %d not possible on %d rows", depth, g.nj); return CFD_E_ARG; }
    double *pl[4] = {a, b, d, e};
    int n = 0;
    while (n < 4 && pl[n]) ++n;
    if (c->comm.halo(c->comm.user, n, pl, g.pitch, g.nj, depth) != 0) {
        set_error("halo exchange callback failed");
        return CFD_E_STATE;
    }
    return CFD_OK;
}

int all_reduce(Ctx *c, double *dev, int n, int op) {
    if (c->comm.nranks <= 1) return CFD_OK;
    CFD_TRY(need_comm(c, (const void *)c->comm.allreduce, "all-reduce"));
    if (c->comm.allreduce(c->comm.user, dev, n, op) != 0) { set_error("all-reduce callback failed"); return CFD_E_STATE; }
    return CFD_OK;
}

int all_gather_rows(Ctx *c, const double *src, double *dst, size_t elems_each) {
    if (c->comm.nranks <= 1) {
        CFD_HIP(hipMemcpyAsync(dst, src, elems_each * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        return CFD_OK;
    }
    CFD_TRY(need_comm(c, (const void *)c->comm.allgather, "all-gather"));
    if (c->comm.allgather(c->comm.user, src, dst, elems_each) != 0) { set_error("all-gather callback failed"); return CFD_E_STATE; }
    return CFD_OK;
}

}  // namespace cfd
