// Which cache-policy bits should the sweeps' stash stores / loads carry?  (round 3: torch's fill_ writes 6.7 TB/s, the `nt`
// stream of tools/micro/hbm_stream.hip 5.0-5.3 on the same box.)  Stash-shaped accesses (a wave-instruction = 4 x 256 B,
// rows np*16 B apart), 256 workgroups x 8 waves, 4 wave-instructions per array in flight; write-only and 2 reads : 2 writes.
//   policy 0: none   1: nt   2: sc1   3: sc0 sc1   4: nt sc1   5: nt sc0 sc1   6: sc0   7: nt sc0
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_policy.hip -o dbg/store_policy ; run: dbg/store_policy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define POL(P) ((P) == 0 ? "" : (P) == 1 ? " nt" : (P) == 2 ? " sc1" : (P) == 3 ? " sc0 sc1" : (P) == 4 ? " nt sc1" : (P) == 5 ? " nt sc0 sc1" : (P) == 6 ? " sc0" : " nt sc0")

template <int P> __device__ __forceinline__ void st(f32x4* p, f32x4 v) {
    if constexpr (P == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 6) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if constexpr (P == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}
template <int P> __device__ __forceinline__ f32x4 ld(const f32x4* p) {
    f32x4 v;
    if constexpr (P == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (P == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// MODE 2: write only (one array); MODE 3: 2 reads : 2 writes
template <int MODE, int LP, int SP>
__global__ __launch_bounds__(512) void stream(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c,
                                              f32x4* __restrict__ d, long np, int rows) {
    constexpr int DEPTH = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ngroups = np / 16;
    const long g0 = ngroups * blockIdx.x / gridDim.x, g1 = ngroups * (blockIdx.x + 1) / gridDim.x;
    for (long g = g0 + wave; g < g1; g += 8) {
        const long col = g * 16 + (lane & 15);
        for (int r = 0; r + 4 * (DEPTH - 1) < rows; r += 4 * DEPTH) {
            f32x4 v[DEPTH], w[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const long off = (long)(r + 4 * u + (lane >> 4)) * np + col;
                if (MODE == 3) { v[u] = ld<LP>(a + off); w[u] = ld<LP>(b + off); }
                else v[u] = f32x4{1.f, 2.f, 3.f, (float)r};
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const long off = (long)(r + 4 * u + (lane >> 4)) * np + col;
                if (MODE == 3) {
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // loads of step u landed (6 younger loads / stores behind them)
                    f32x4 x = v[u], y = w[u];
                    asm volatile("" : "+v"(x), "+v"(y));
                    st<SP>(c + off, x * y); st<SP>(d + off, x + y);
                } else st<SP>(c + off, v[u]);
            }
        }
    }
}

// torch-like fill: many small workgroups, linear, four 16-byte stores per thread
template <int SP>
__global__ __launch_bounds__(256) void fill_like(f32x4* __restrict__ c) {
    f32x4* p = c + (long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) st<SP>(p + 256 * u, f32x4{1.f, 2.f, 3.f, 4.f});
}
template <int SP>
double run_fill(f32x4* c, long total) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)(total / 1024);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((fill_like<SP>), dim3(grid), dim3(256), 0, 0, c);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill_like<SP>), dim3(grid), dim3(256), 0, 0, c);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)total * 16 / (ms / reps * 1e-3) / 1e12;
}

template <int MODE, int LP, int SP>
double run(const f32x4* a, const f32x4* b, f32x4* c, f32x4* d, long np, int rows) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream<MODE, LP, SP>), dim3(256), dim3(512), 0, 0, a, b, c, d, np, rows);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MODE, LP, SP>), dim3(256), dim3(512), 0, 0, a, b, c, d, np, rows);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * np * 16 * (MODE == 3 ? 4 : 1);
    return bytes / (ms / reps * 1e-3) / 1e12;
}

int main() {
    const long np = 100096; const int rows = 512;
    const size_t bytes = (size_t)rows * np * 16;
    f32x4 *a, *b, *c, *d;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    {   // do the four concurrent streams of a sweep collide on HBM channels?  The stash arrays sit a multiple of 2 MiB apart,
        // so all four are always at the same offset within a 2 MiB page: stagger b, c, d by 1, 2, 3 x D bytes.
        f32x4 *a2, *b2, *c2, *d2;
        const size_t slack = 3u << 20;
        hipMalloc(&a2, bytes + slack); hipMalloc(&b2, bytes + slack); hipMalloc(&c2, bytes + slack); hipMalloc(&d2, bytes + slack);
        hipMemset(a2, 0, bytes + slack); hipMemset(b2, 0, bytes + slack);
        printf("2r:2w (nt loads, nt stores), arrays staggered by D = 0 | 256 B | 1 KiB | 4 KiB | 16 KiB | 64 KiB | 256 KiB | 1 MiB / 3:\n");
        for (int rep = 0; rep < 2; ++rep) {
            printf("   ");
            for (long D : {0l, 256l, 1024l, 4096l, 16384l, 65536l, 262144l, 349440l}) {
                const long g = D / 16;
                printf(" %.2f", run<3, 1, 1>(a2, b2 + g, c2 + 2 * g, d2 + 3 * g, np, rows));
            }
            printf("\n");
        }
        printf("1r:2w-shaped check (write only, one array) at the same bases:");
        for (long D : {0l, 4096l, 65536l}) printf(" %.2f", run<2, 0, 1>(a2, b2, c2 + D / 16, d2, np, rows));
        printf("\n");
        hipFree(a2); hipFree(b2); hipFree(c2); hipFree(d2);
    }
    printf("stash-shaped streams of 0.82 GB arrays, TB/s; store policy: none | nt | sc1 | sc0 sc1 | nt sc1 | nt sc0 sc1 | sc0 | nt sc0\n");
#define W(SP) run<2, 0, SP>(a, b, c, d, np, rows)
    printf("write only           : %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", W(0), W(1), W(2), W(3), W(4), W(5), W(6), W(7));
#define M(LP, SP) run<3, LP, SP>(a, b, c, d, np, rows)
#define MROW(LP, name) printf("2r:2w, loads %-9s: %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", name, M(LP, 0), M(LP, 1), M(LP, 2), M(LP, 3), M(LP, 4), M(LP, 5), M(LP, 6), M(LP, 7))
    MROW(0, "none"); MROW(1, "nt"); MROW(2, "sc1"); MROW(3, "sc0 sc1"); MROW(4, "nt sc1"); MROW(5, "nt sc0 sc1");
#define F(SP) run_fill<SP>(c, (long)rows * np)
    printf("fill-like (50 048 WGs) : %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", F(0), F(1), F(2), F(3), F(4), F(5), F(6), F(7));
    hipMemsetAsync(c, 0, bytes, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipMemsetAsync(c, 0, bytes, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemsetAsync         : %.2f\n", bytes / (ms / 10 * 1e-3) / 1e12);
    return 0;
}
