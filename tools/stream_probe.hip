// Streaming probe for the marching kernels (gfx950): how fast can a wave that walks down a column
// strip read NR planes and write NW planes, with 8 or 16 bytes per lane, plain or nontemporal
// accesses, and different segment lengths / waves per CU?  Prints GB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Planes { const double *r[16]; double *w[4]; };

template <int NR, int NW, int VEC, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void k_march(Planes pl, int ni, int nj, int pitch, int Ls) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cols = 64 * VEC;
    const int nstrips = ni / cols, nsegs = nj / Ls;
    for (int item = blockIdx.x * 4 + wave; item < nstrips * nsegs; item += gridDim.x * 4) {
        const int strip = item % nstrips, seg = item / nstrips;
        const size_t base = (size_t)(seg * Ls) * pitch + (size_t)strip * cols + lane * VEC;
        for (int j = 0; j < Ls; j += UNROLL) {
            double acc[UNROLL][VEC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[u][v] = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    if (VEC == 2) {
                        dbl2 t = NT ? __builtin_nontemporal_load((const dbl2 *)(pl.r[q] + c)) : *(const dbl2 *)(pl.r[q] + c);
                        acc[u][0] += t.x; acc[u][VEC - 1] += t.y;
                    } else {
                        acc[u][0] += NT ? __builtin_nontemporal_load(pl.r[q] + c) : pl.r[q][c];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    if (VEC == 2) {
                        dbl2 t; t.x = acc[u][0] + q; t.y = acc[u][VEC - 1] - q;
                        if (NT) __builtin_nontemporal_store(t, (dbl2 *)(pl.w[q] + c)); else *(dbl2 *)(pl.w[q] + c) = t;
                    } else {
                        if (NT) __builtin_nontemporal_store(acc[u][0] + q, pl.w[q] + c); else pl.w[q][c] = acc[u][0] + q;
                    }
                }
            }
        }
    }
}

template <int NR, int NW, int VEC, bool NT, int UNROLL>
void run(const char *name, Planes pl, int n, int pitch, int Ls, int blocks, hipStream_t s) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int cols = 64 * VEC;
    const long items = (long)(n / cols) * (n / Ls);
    int nb = blocks > 0 ? blocks : (int)((items + 3) / 4);
    hipLaunchKernelGGL((k_march<NR, NW, VEC, NT, UNROLL>), dim3(nb), dim3(256), 0, s, pl, n, n, pitch, Ls);
    CHECK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_march<NR, NW, VEC, NT, UNROLL>), dim3(nb), dim3(256), 0, s, pl, n, n, pitch, Ls);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double gb = (double)(NR + NW) * 8.0 * n * n * reps / (ms * 1e-3) / 1e9;
    printf("%-44s R%2d W%d vec%d nt%d unroll%d Ls%3d blocks%6d : %7.1f us  %7.1f GB/s\n", name, NR, NW, VEC, (int)NT, UNROLL, Ls, nb,
           1e3 * ms / reps, gb);
    fflush(stdout);
}

int main() {
    const int n = 8192, pitch = 8192 + 32;
    const size_t elems = (size_t)(n + 4) * pitch;
    Planes pl{};
    for (int q = 0; q < 16; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); CHECK(hipMemset(p, 0, elems * 8)); pl.r[q] = p; }
    for (int q = 0; q < 4; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); pl.w[q] = p; }
    hipStream_t s; CHECK(hipStreamCreate(&s));
    for (int Ls : {16, 64}) {
        for (int blocks : {0, 2048, 1024}) {
            run<5, 3, 1, false, 2>("cg_pq-like", pl, n, pitch, Ls, blocks, s);
            run<5, 3, 2, false, 2>("cg_pq-like", pl, n, pitch, Ls, blocks, s);
            run<5, 3, 1, true, 2>("cg_pq-like", pl, n, pitch, Ls, blocks, s);
            run<5, 3, 2, true, 2>("cg_pq-like", pl, n, pitch, Ls, blocks, s);
        }
    }
    for (int Ls : {16, 64}) {
        run<13, 2, 1, false, 2>("mom_rhs-like", pl, n, pitch, Ls, 0, s);
        run<13, 2, 2, false, 2>("mom_rhs-like", pl, n, pitch, Ls, 0, s);
        run<13, 2, 2, false, 1>("mom_rhs-like", pl, n, pitch, Ls, 0, s);
        run<13, 2, 2, true, 1>("mom_rhs-like", pl, n, pitch, Ls, 0, s);
        run<4, 2, 1, false, 2>("mg_down-like", pl, n, pitch, Ls, 0, s);
        run<4, 2, 2, false, 2>("mg_down-like", pl, n, pitch, Ls, 0, s);
        run<4, 2, 2, false, 4>("mg_down-like", pl, n, pitch, Ls, 0, s);
        run<2, 1, 1, false, 4>("axpy-like", pl, n, pitch, Ls, 0, s);
        run<2, 1, 2, false, 4>("axpy-like", pl, n, pitch, Ls, 0, s);
        run<1, 1, 2, false, 4>("copy-like", pl, n, pitch, Ls, 0, s);
        run<1, 1, 2, true, 4>("copy-like", pl, n, pitch, Ls, 0, s);
    }
    return 0;
}
