// Streaming probe, round 2 (gfx950): what does the marching access pattern reach when the row loads go
// through an LDS-DMA ring (global_load_lds_dwordx4, D rows in flight per wave, no VGPR destination)
// instead of VGPR-destination loads one row ahead?  Compared with the register form (8 / 16 B per lane,
// plain / nontemporal) and with a flat grid-stride sweep of the same planes.  Prints GB/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/stream_probe2 tools/stream_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Planes { const double *r[16]; double *w[4]; };

// ---------------------------------------------------------------- LDS-DMA ring
// one wave = 128 columns (16 B per lane); slot = NR x 1 KiB; D slots per wave
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_nt(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N > 63 ? 63 : N) : "memory"); }

template <int NR, int NW, int D, bool NT, bool NTS>
__global__ __launch_bounds__(256) void k_ring(Planes pl, int ni, int nj, int pitch, int Ls) {
    extern __shared__ __align__(16) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned my = (unsigned)(size_t)lds + wave * (D * NR * 1024);      // LDS byte address of this wave's ring
    const int nstrips = ni / 128, nsegs = nj / Ls;
    for (int item = blockIdx.x * 4 + wave; item < nstrips * nsegs; item += gridDim.x * 4) {
        const int strip = item % nstrips, seg = item / nstrips;
        const size_t base = (size_t)(seg * Ls) * pitch + (size_t)strip * 128 + lane * 2;
        auto issue = [&](int j) {
            const unsigned slot = __builtin_amdgcn_readfirstlane(my + (unsigned)(j % D) * (NR * 1024));
            const size_t c = base + (size_t)j * pitch;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                if (NT) glds16_nt(pl.r[q] + c, slot + q * 1024);
                else glds16(pl.r[q] + c, slot + q * 1024);
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < D - 1; ++j) issue(j);
        for (int j = 0; j < Ls; ++j) {
            const int ahead = Ls - 1 - j < D - 1 ? Ls - 1 - j : D - 1;      // rows in flight behind row j after this issue
            if (j + D - 1 < Ls) issue(j + D - 1);
            if (ahead >= D - 1) wait_vm<(D - 1) * NR>();
            else if (D > 2 && ahead == D - 2) wait_vm<(D > 2 ? (D - 2) * NR : 0)>();
            else if (D > 3 && ahead == D - 3) wait_vm<(D > 3 ? (D - 3) * NR : 0)>();
            else wait_vm<0>();
            const char *sl = lds + wave * (D * NR * 1024) + (j % D) * (NR * 1024) + lane * 16;
            double a0 = 0.0, a1 = 0.0;
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const dbl2 t = *(const dbl2 *)(sl + q * 1024);
                a0 += t.x; a1 += t.y;
            }
            const size_t c = base + (size_t)j * pitch;
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                dbl2 t; t.x = a0 + q; t.y = a1 - q;
                if (NTS) __builtin_nontemporal_store(t, (dbl2 *)(pl.w[q] + c)); else *(dbl2 *)(pl.w[q] + c) = t;
            }
            if (NW == 0 && a0 + a1 == 123.456) pl.w[0][c] = a0;      // sink: keeps the reads alive
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

// ---------------------------------------------------------------- register form (round 1's probe) with PF rows of prefetch
template <int NR, int NW, int VEC, bool NT, bool NTS, int PF>
__global__ __launch_bounds__(256) void k_march(Planes pl, int ni, int nj, int pitch, int Ls) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cols = 64 * VEC;
    const int nstrips = ni / cols, nsegs = nj / Ls;
    for (int item = blockIdx.x * 4 + wave; item < nstrips * nsegs; item += gridDim.x * 4) {
        const int strip = item % nstrips, seg = item / nstrips;
        const size_t base = (size_t)(seg * Ls) * pitch + (size_t)strip * cols + lane * VEC;
        for (int j = 0; j < Ls; j += PF) {
            double acc[PF][2];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
                acc[u][0] = acc[u][1] = 0.0;
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    if (VEC == 2) {
                        dbl2 t = NT ? __builtin_nontemporal_load((const dbl2 *)(pl.r[q] + c)) : *(const dbl2 *)(pl.r[q] + c);
                        acc[u][0] += t.x; acc[u][1] += t.y;
                    } else {
                        acc[u][0] += NT ? __builtin_nontemporal_load(pl.r[q] + c) : pl.r[q][c];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const size_t c = base + (size_t)(j + u) * pitch;
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    if (VEC == 2) {
                        dbl2 t; t.x = acc[u][0] + q; t.y = acc[u][1] - q;
                        if (NTS) __builtin_nontemporal_store(t, (dbl2 *)(pl.w[q] + c)); else *(dbl2 *)(pl.w[q] + c) = t;
                    } else {
                        if (NTS) __builtin_nontemporal_store(acc[u][0] + q, pl.w[q] + c); else pl.w[q][c] = acc[u][0] + q;
                    }
                }
                if (NW == 0 && acc[u][0] + acc[u][1] == 123.456) pl.w[0][c] = acc[u][0];
            }
        }
    }
}

// ---------------------------------------------------------------- flat sweep: row-major, 16 B per lane, grid-stride
template <int NR, int NW, bool NT, bool NTS, int UN>
__global__ __launch_bounds__(256) void k_flat(Planes pl, size_t n2) {     // n2 = number of dbl2 elements
    const size_t stride = (size_t)gridDim.x * 256 * UN;
    for (size_t i0 = (size_t)blockIdx.x * 256 * UN + threadIdx.x; i0 < n2; i0 += stride) {
        dbl2 acc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            acc[u].x = 0.0; acc[u].y = 0.0;
            if (i < n2) {
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    const dbl2 t = NT ? __builtin_nontemporal_load((const dbl2 *)pl.r[q] + i) : ((const dbl2 *)pl.r[q])[i];
                    acc[u] += t;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            if (i < n2) {
#pragma unroll
                for (int q = 0; q < NW; ++q) {
                    dbl2 t = acc[u]; t.x += q;
                    if (NTS) __builtin_nontemporal_store(t, (dbl2 *)pl.w[q] + i); else ((dbl2 *)pl.w[q])[i] = t;
                }
                if (NW == 0 && acc[u].x + acc[u].y == 123.456) ((dbl2 *)pl.w[0])[i] = acc[u];
            }
        }
    }
}

static hipStream_t s;
static void report(const char *name, int NR, int NW, float ms, int reps, int n, const char *extra) {
    const double gb = (double)(NR + NW) * 8.0 * n * n * reps / (ms * 1e-3) / 1e9;
    printf("%-10s R%2d W%d %-36s : %7.1f us  %7.1f GB/s\n", name, NR, NW, extra, 1e3 * ms / reps, gb);
    fflush(stdout);
}
template <typename F>
static float time_it(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
    return ms;
}
#define TIME(...) const int reps = 10; const float ms = time_it([&]() { __VA_ARGS__; }, reps);

template <int NR, int NW, int D, bool NT, bool NTS>
void run_ring(Planes pl, int n, int pitch, int Ls, int blocks) {
    const size_t lds = (size_t)4 * D * NR * 1024;
    if (lds > 160 * 1024) return;
    CHECK(hipFuncSetAttribute((const void *)k_ring<NR, NW, D, NT, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long items = (long)(n / 128) * (n / Ls);
    const int nb = blocks > 0 ? blocks : (int)((items + 3) / 4);
    TIME(k_ring<NR, NW, D, NT, NTS><<<dim3(nb), dim3(256), lds, s>>>(pl, n, n, pitch, Ls));
    char extra[96];
    snprintf(extra, sizeof extra, "ring D%d nt%d nts%d Ls%3d blk%5d lds%3zuK", D, (int)NT, (int)NTS, Ls, nb, lds / 1024);
    report("ring", NR, NW, ms, reps, n, extra);
}
template <int NR, int NW, int VEC, bool NT, bool NTS, int PF>
void run_march(Planes pl, int n, int pitch, int Ls, int blocks) {
    const long items = (long)(n / (64 * VEC)) * (n / Ls);
    const int nb = blocks > 0 ? blocks : (int)((items + 3) / 4);
    TIME(k_march<NR, NW, VEC, NT, NTS, PF><<<dim3(nb), dim3(256), 0, s>>>(pl, n, n, pitch, Ls));
    char extra[96];
    snprintf(extra, sizeof extra, "march vec%d nt%d nts%d pf%d Ls%3d blk%5d", VEC, (int)NT, (int)NTS, PF, Ls, nb);
    report("march", NR, NW, ms, reps, n, extra);
}
template <int NR, int NW, bool NT, bool NTS, int UN>
void run_flat(Planes pl, int n, int pitch, int blocks) {
    const size_t n2 = (size_t)n * pitch / 2;
    TIME(k_flat<NR, NW, NT, NTS, UN><<<dim3(blocks), dim3(256), 0, s>>>(pl, n2));
    char extra[96];
    snprintf(extra, sizeof extra, "flat nt%d nts%d un%d blk%5d", (int)NT, (int)NTS, UN, blocks);
    report("flat", NR, NW, ms, reps, n, extra);
}

template <int NR, int NW>
void shape(Planes pl, int n, int pitch) {
    printf("---- shape R%d W%d\n", NR, NW);
    run_flat<NR, NW, false, false, 4>(pl, n, pitch, 2048);
    run_flat<NR, NW, true, false, 4>(pl, n, pitch, 2048);
    run_flat<NR, NW, true, true, 4>(pl, n, pitch, 2048);
    run_flat<NR, NW, false, false, 2>(pl, n, pitch, 8192);
    run_march<NR, NW, 1, false, false, 2>(pl, n, pitch, 32, 0);
    run_march<NR, NW, 2, false, false, 2>(pl, n, pitch, 32, 0);
    run_march<NR, NW, 2, true, false, 2>(pl, n, pitch, 32, 0);
    run_march<NR, NW, 2, true, true, 2>(pl, n, pitch, 32, 0);
    run_march<NR, NW, 2, true, true, 4>(pl, n, pitch, 32, 0);
    run_march<NR, NW, 2, true, true, 4>(pl, n, pitch, 32, 1024);
    run_ring<NR, NW, 2, false, false>(pl, n, pitch, 32, 0);
    run_ring<NR, NW, 3, false, false>(pl, n, pitch, 32, 0);
    run_ring<NR, NW, 4, false, false>(pl, n, pitch, 32, 0);
    run_ring<NR, NW, 4, true, false>(pl, n, pitch, 32, 0);
    run_ring<NR, NW, 4, true, true>(pl, n, pitch, 32, 0);
    run_ring<NR, NW, 4, true, true>(pl, n, pitch, 64, 0);
    run_ring<NR, NW, 4, true, true>(pl, n, pitch, 64, 512);
    if (NR <= 6) {
        run_ring<NR, NW, 6, false, false>(pl, n, pitch, 32, 0);
        run_ring<NR, NW, 6, true, true>(pl, n, pitch, 32, 0);
        run_ring<NR, NW, 6, true, true>(pl, n, pitch, 64, 512);
    }
}

int main() {
    const int n = 8192, pitch = 8192 + 32;
    const size_t elems = (size_t)(n + 8) * pitch;
    Planes pl{};
    for (int q = 0; q < 16; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); CHECK(hipMemset(p, 0, elems * 8)); pl.r[q] = p; }
    for (int q = 0; q < 4; ++q) { double *p; CHECK(hipMalloc(&p, elems * 8)); pl.w[q] = p; }
    CHECK(hipStreamCreate(&s));
    shape<1, 1>(pl, n, pitch);      // copy
    shape<5, 3>(pl, n, pitch);      // k_cg_pq
    shape<6, 2>(pl, n, pitch);      // k_chebk<1,4>
    shape<9, 4>(pl, n, pitch);      // k_chebk<2,4>
    shape<4, 1>(pl, n, pitch);      // k_mg_up_k-like
    shape<13, 2>(pl, n, pitch);     // momentum RHS
    shape<8, 0>(pl, n, pitch);      // reads only
    return 0;
}
