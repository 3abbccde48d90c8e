// Does lockstep between the waves of a workgroup help the marching access pattern?  (DRAM page locality: adjacent
// strips of the same plane rows read at the same time.)  Kernel: NR planes read, NW written, one wave per 64-column
// strip marching down Ls rows, two rows of loads in flight; a workgroup = WPB adjacent strips; optional workgroup
// barrier every BAR rows.  hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe3.bin tools/stream_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Planes { const double *r[16]; double *w[4]; };
static hipStream_t s;

template <int NR, int NW, int BAR>
__global__ void k_march(Planes pl, int ni, int nj, int pitch, int Ls, int co) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int nstrips = ni / 64, nsegs = nj / Ls;
    // uniform trip count over the workgroup (the host sizes the grid so that every wave has an item each round)
    for (int item = blockIdx.x * wpb + wave; item < nstrips * nsegs; item += gridDim.x * wpb) {
        const int strip = item % nstrips, seg = item / nstrips;
        const size_t base = (size_t)(seg * Ls) * pitch + (size_t)strip * 64 + lane + co;
        for (int j = 0; j < Ls; j += 2) {
            double acc[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
                acc[u] = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) acc[u] += pl.r[q][c];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
#pragma unroll
                for (int q = 0; q < NW; ++q) pl.w[q][c] = acc[u] + q;
            }
            if (BAR > 0 && (j % BAR) == BAR - 2) __syncthreads();
        }
    }
}

template <int NR, int NW, int BAR>
static void run(Planes pl, int n, int pitch, int Ls, int wpb, int co) {
    const long items = (long)(n / 64) * (n / Ls);
    const int nb = (int)(items / wpb);           // n / 64 = 128 strips: divisible by every wpb used
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_march<NR, NW, BAR><<<dim3(nb), dim3(64 * wpb), 0, s>>>(pl, n, n, pitch, Ls, co);
    CHECK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) k_march<NR, NW, BAR><<<dim3(nb), dim3(64 * wpb), 0, s>>>(pl, n, n, pitch, Ls, co);
    CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
    const double gb = (double)(NR + NW) * n * n * 8.0 * reps / (ms * 1e-3) / 1e9;
    printf("R%2d W%d Ls%3d wpb%2d bar%2d co%2d : %7.1f us %7.1f GB/s\n", NR, NW, Ls, wpb, BAR, co, 1e3 * ms / reps, gb);
    fflush(stdout);
}

template <int NR, int NW>
static void shape(Planes pl, int n, int pitch) {
    for (int co : {0, 4, 8, 12, 16, 24})
        for (int wpb : {4, 16}) {
            run<NR, NW, 0>(pl, n, pitch, 32, wpb, co);
            run<NR, NW, 2>(pl, n, pitch, 32, wpb, co);
        }
}

int main() {
    const int n = 8192, pitch = 8192 + 32;
    const size_t elems = (size_t)(n + 8) * pitch;
    Planes pl{};
    for (int q = 0; q < 16; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); CHECK(hipMemset(p, 0, elems * 8)); pl.r[q] = p; }
    for (int q = 0; q < 4; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); pl.w[q] = p; }
    CHECK(hipStreamCreate(&s));
    shape<6, 2>(pl, n, pitch);
    return 0;
}
