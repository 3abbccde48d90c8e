// cfd_bulk_velocity_noalloc: the volume-weighted mean of cfd_bulk_velocity for callers that cannot allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
// reference.  The per-workgroup partial sums live in a file-scope device array of fixed size instead of the
// context's workspace; the call allocates nothing, synchronises nothing and leaves its result in device memory.
//
// `scratch` below is a host-side stand-in for the device array: host code never passes that name to a kernel or to
// a memory routine.  Its device address is looked up once per GPU (hipGetSymbolAddress, with that GPU current)
// and kept in g_scratch[device]; each GPU holds its own copy of the array at its own address.
//
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two host threads) race
// on the partial sums.  Nothing has to start at zero: a call overwrites exactly the partials that its second kernel
// reads (blocks 0 .. nb-1 of both rows), so the contents a previous, larger call left behind are never looked at.
#include <atomic>
#include <mutex>

#include "internal.h"

namespace cfd {

constexpr int NA_MAX_BLOCKS = 2048;                    // = red_blocks() cap of launch_weighted_sum
constexpr int NA_MAX_SIDE = 32768;                     // cells per side the entry point accepts
constexpr int NA_SCRATCH_DOUBLES = 2 * MAX_PARTIALS;   // two rows of partial sums, laid out like Ctx::partials
constexpr int NA_MAX_DEVICES = 64;

static __device__ __align__(256) double scratch[NA_SCRATCH_DOUBLES];      // 64 KiB per GPU, held until the process ends

static std::atomic<double *> g_scratch[NA_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

// device address of this GPU's copy of `scratch`: looked up once per GPU, with it current (cfd_create does it, else the
// first call of the entry point on that GPU: that one call may load the code object and takes a mutex; every later
// call is one atomic load)
static int scratch_ptr(double **out) {
    int dev = -1;
    CFD_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= NA_MAX_DEVICES) { set_error("cfd_bulk_velocity_noalloc: device ordinal %d out of range", dev); return CFD_E_ARG; }
    double *ptr = g_scratch[dev].load(std::memory_order_acquire);
    if (!ptr) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        ptr = g_scratch[dev].load(std::memory_order_relaxed);
        if (!ptr) {
            CFD_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) { set_error("cfd_bulk_velocity_noalloc: hipGetSymbolAddress returned a null pointer"); return CFD_E_HIP; }
            g_scratch[dev].store(ptr, std::memory_order_release);
        }
    }
    *out = ptr;
    return CFD_OK;
}

// the arithmetic of k_weighted_sum / k_sum_finish (stencil.hip, linalg.hip), same tiles, same summation order
__device__ __forceinline__ double na_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double na_block_sum_256(double v) {
    __shared__ double sm[4];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    v = na_wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) sm[tid >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void k_na_weighted_sum(Geo g, const double *f, const double *w, double *partials, int tiles_x, int tiles) {
    double a = 0.0, b = 0.0;
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int i = (tile % tiles_x) * 64 + threadIdx.x, j = (tile / tiles_x) * 4 + threadIdx.y;
        if (i >= g.ni || j >= g.nj) continue;
        const size_t c = g.cell(i, j);
        a += f[c] * w[c];
        b += w[c];
    }
    a = na_block_sum_256(a);
    b = na_block_sum_256(b);
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        partials[blockIdx.x] = a;
        partials[MAX_PARTIALS + blockIdx.x] = b;
    }
}
// out = {sum f w / sum w, sum f w, sum w}
__global__ void k_na_finish(const double *partials, int n, double *out) {
    __shared__ double sm[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = 0; q < 2; ++q) {
        double v = 0.0;
        for (int t = tid; t < n; t += 256) v += partials[(size_t)q * MAX_PARTIALS + t];
        v = na_wave_sum(v);
        if (lane == 0) sm[q][wave] = v;
    }
    __syncthreads();
    if (tid == 0) {
        const double a = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]), b = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
        out[0] = a / b;
        out[1] = a;
        out[2] = b;
    }
}

int noalloc_prepare_device() {
    double *p = nullptr;
    return scratch_ptr(&p);
}

}  // namespace cfd

using namespace cfd;

extern "C" int cfd_bulk_velocity_noalloc(const double *field_dev, const double *volume_dev, int ni, int nj, int pitch,
                                         void *hip_stream, double *out_dev) {
    if (!field_dev || !volume_dev || !out_dev) { set_error("cfd_bulk_velocity_noalloc: null pointer"); return CFD_E_ARG; }
    if (ni < 1 || nj < 1 || ni > NA_MAX_SIDE || nj > NA_MAX_SIDE) {
        set_error("cfd_bulk_velocity_noalloc: %d x %d cells outside the supported 1 .. %d per side", ni, nj, NA_MAX_SIDE);
        return CFD_E_ARG;
    }
    if (pitch < ni + CO + 2 || pitch % 2) { set_error("cfd_bulk_velocity_noalloc: pitch %d too small for %d columns (need >= ni + %d, even)", pitch, ni, CO + 2); return CFD_E_ARG; }
    double *ptr = nullptr;
    CFD_TRY(scratch_ptr(&ptr));
    Geo g{};
    g.ni = ni; g.nj = nj; g.pitch = pitch; g.cutlo = 0; g.cuthi = 0;
    const int tx = (ni + 63) / 64;
    const int tiles = tx * ((nj + 3) / 4);
    const int nb = tiles < NA_MAX_BLOCKS ? tiles : NA_MAX_BLOCKS;
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_na_weighted_sum, dim3(nb), dim3(64, 4, 1), 0, st, g, field_dev, volume_dev, ptr, tx, tiles);
    hipLaunchKernelGGL(k_na_finish, dim3(1), dim3(256), 0, st, (const double *)ptr, nb, out_dev);
    CFD_HIP(hipGetLastError());
    return CFD_OK;
}
