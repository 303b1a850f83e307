// What does this box's HBM deliver to a hand-written stream in the SWEEPS' access shape?  
// 256 persistent workgroups x 8 waves, 16 B per lane, non-temporal, `depth` wave-instructions in flight per wave.
//   mode 0: copy (1 read : 1 write)   1: read only   2: write only   3: 2 reads : 2 writes (reverse / adjoint-forward sweep)
//   shape 0: linear (a wave-instruction = 1 KiB contiguous)
//   shape 1: stash (a wave-instruction = 4 x 256 B, the four feature-quad rows of a 16-column tile, rows np*16 B apart)
//   shape 2 (read only; `wgrad` argument): the weight-gradient GEMM's staging loads — a wave-instruction = 16 rows x 64 B
//            (4 lanes x 16 B), four instructions at +0, +64, +128, +192 B cover 256 B of each row; known byte count for
//            calibrating FETCH_SIZE on this access shape:  rocprofv3 --pmc FETCH_SIZE -- dbg/hbm_stream wgrad
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_stream.hip -o dbg/hbm_stream ; run: dbg/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int SHAPE, int DEPTH>
__global__ __launch_bounds__(512) void stream(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c,
                                              f32x4* __restrict__ d, long np, int rows, float* sink) {
    // SHAPE 0: arrays are `total` granules, walked linearly by (workgroup, wave).  SHAPE 1: arrays are [rows][np] granules;
    // a wave owns 16 columns and walks the rows four at a time (lane>>4 = row within the quad of rows), like a sweep's tail.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {0, 0, 0, 0};
    if (SHAPE == 0) {
        const long total = (long)rows * np;
        const long per = total / ((long)gridDim.x * 8);
        const long base = ((long)blockIdx.x * 8 + wave) * per;
        for (long i = lane; i + 64 * (DEPTH - 1) < per; i += 64 * DEPTH) {
            f32x4 v[DEPTH], w[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                if (MODE != 2) v[u] = __builtin_nontemporal_load(a + base + i + 64 * u);
                if (MODE == 3) w[u] = __builtin_nontemporal_load(b + base + i + 64 * u);
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                if (MODE == 0) __builtin_nontemporal_store(v[u], c + base + i + 64 * u);
                if (MODE == 1) acc += v[u];
                if (MODE == 2) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, (float)i}, c + base + i + 64 * u);
                if (MODE == 3) { __builtin_nontemporal_store(v[u] * w[u], c + base + i + 64 * u); __builtin_nontemporal_store(v[u] + w[u], d + base + i + 64 * u); }
            }
        }
    } else {
        const long ngroups = np / 16;
        const long g0 = ngroups * blockIdx.x / gridDim.x, g1 = ngroups * (blockIdx.x + 1) / gridDim.x;
        for (long g = g0 + wave; g < g1; g += 8) {
            const long col = g * 16 + (lane & 15);
            for (int r = 0; r + 4 * (DEPTH - 1) < rows; r += 4 * DEPTH) {
                f32x4 v[DEPTH], w[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const long off = (long)(r + 4 * u + (lane >> 4)) * np + col;
                    if (MODE != 2) v[u] = __builtin_nontemporal_load(a + off);
                    if (MODE == 3) w[u] = __builtin_nontemporal_load(b + off);
                }
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const long off = (long)(r + 4 * u + (lane >> 4)) * np + col;
                    if (MODE == 0) __builtin_nontemporal_store(v[u], c + off);
                    if (MODE == 1) acc += v[u];
                    if (MODE == 2) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, (float)r}, c + off);
                    if (MODE == 3) { __builtin_nontemporal_store(v[u] * w[u], c + off); __builtin_nontemporal_store(v[u] + w[u], d + off); }
                }
            }
        }
    }
    if (MODE == 1 && acc[0] == 123.456f) *sink = acc[1];
}

__global__ __launch_bounds__(512) void wgrad_shape(const f32x4* __restrict__ a, long np, int rows, float* sink) {
    // workgroup = 256 consecutive rows x a range of 16-column stages; thread: row group tid >> 2 (x2 passes), column group tid & 3
    const int tid = threadIdx.x;
    const long nstage = np / 16;
    const long s0 = nstage * blockIdx.x / gridDim.x, s1 = nstage * (blockIdx.x + 1) / gridDim.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int rb = 0; rb < rows; rb += 128) {
        const int row = rb + (tid >> 2);
        for (long st = s0; st < s1; ++st) {
            const f32x4* p = a + (long)row * np + st * 16 + (tid & 3);
            const f32x4 v0 = __builtin_nontemporal_load(p), v1 = __builtin_nontemporal_load(p + 4), v2 = __builtin_nontemporal_load(p + 8), v3 = __builtin_nontemporal_load(p + 12);
            acc += v0 + v1 + v2 + v3;
        }
    }
    if (acc[0] == 123.456f) *sink = acc[1];
}

template <int MODE, int SHAPE, int DEPTH>
double run(const f32x4* a, const f32x4* b, f32x4* c, f32x4* d, long np, int rows, float* sink, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream<MODE, SHAPE, DEPTH>), dim3(grid), dim3(512), 0, 0, a, b, c, d, np, rows, sink);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MODE, SHAPE, DEPTH>), dim3(grid), dim3(512), 0, 0, a, b, c, d, np, rows, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)rows * np * 16 * (MODE == 0 ? 2 : MODE == 3 ? 4 : 1);
    return bytes / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    // one stash array of the 8x256 / 100 096-column step: [8 layers x 64 feature quads = 512 rows][np] granules = 0.82 GB
    const long np = 100096; const int rows = 512;
    const size_t bytes = (size_t)rows * np * 16;
    f32x4 *a, *b, *c, *d; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    if (argc > 1) {                                     // calibration run: one launch, known bytes
        hipLaunchKernelGGL(wgrad_shape, dim3(256), dim3(512), 0, 0, a, np, rows, sink);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(wgrad_shape, dim3(256), dim3(512), 0, 0, a, np, rows, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("wgrad-shaped read of %.0f bytes: %.2f TB/s\n", (double)bytes, bytes / (ms * 1e-3) / 1e12);
        hipLaunchKernelGGL((stream<1, 1, 4>), dim3(256), dim3(512), 0, 0, a, b, c, d, np, rows, sink);
        hipLaunchKernelGGL((stream<1, 0, 4>), dim3(256), dim3(512), 0, 0, a, b, c, d, np, rows, sink);
        hipDeviceSynchronize();
        return 0;
    }
    printf("array %.2f GB each; TB/s (bytes read + written)\n", bytes / 1e9);
#define ROW(M, name) \
    printf("%-28s linear d2 %.2f d4 %.2f d8 %.2f | stash-shaped d2 %.2f d4 %.2f d8 %.2f | 1024 WGs linear d4 %.2f\n", name, \
           run<M, 0, 2>(a, b, c, d, np, rows, sink, 256), run<M, 0, 4>(a, b, c, d, np, rows, sink, 256), run<M, 0, 8>(a, b, c, d, np, rows, sink, 256), \
           run<M, 1, 2>(a, b, c, d, np, rows, sink, 256), run<M, 1, 4>(a, b, c, d, np, rows, sink, 256), run<M, 1, 8>(a, b, c, d, np, rows, sink, 256), \
           run<M, 0, 4>(a, b, c, d, np, rows, sink, 1024))
    ROW(0, "copy (1r:1w)");
    ROW(1, "read only");
    ROW(2, "write only");
    ROW(3, "2r:2w (rev / adj-fwd sweep)");
    return 0;
}
