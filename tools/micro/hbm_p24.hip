// Which memory shape should a 24-bit ("p24") stash array have?  (round 4)
// Same harness as hbm_stream.hip: 256 persistent workgroups x 8 waves, a wave owns 16 columns and walks the rows of its
// tile four at a time (lane >> 4 = row within the group of four feature-quad rows), non-temporal accesses, DEPTH
// wave-instructions in flight per wave.  A "unit row" = 4 values per column (one 16x16 accumulator tile register quad).
//   shape 0  f32   16 B per lane, rows np*16 B apart: today's stash (256-B segments, two full lines)
//   shape 1  p24r  12 B per lane (dwordx3), rows np*12 B apart: 192-B segments, every second line shared by two waves
//   shape 2  p24t  12 B per lane, tile-major: the 64 lanes of a wave write 768 contiguous bytes (six full lines)
//   shape 3  p24o  octet planes: per lane 16 B (hi 16 bits of 8 values) + 8 B (next 8 bits of 8 values) per TWO unit rows:
//                  256-B and 128-B segments, all full lines
// modes: w = write only, r = read only, rw = 1 read : 1 write, rev = the reverse sweep's mix with the p24 arrays in that
//        shape (reads C as f32 + S as p24, writes Q and R as p24), f32 row = today's 2r:2w.
// prints seconds per launch and TB/s of the bytes actually moved; the number that matters is the TIME for the same rows.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_p24.hip -o dbg/hbm_p24 ; run: dbg/hbm_p24
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void st16(char* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st12(char* p, f32x3 v) { asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8(char* p, f32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4 ld16(const char* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ f32x3 ld12(const char* p) { f32x3 v; asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ f32x2 ld8(const char* p) { f32x2 v; asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void waitall() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// byte address of (row group r4 = 4 unit rows, column group g, lane) in an array of the given shape; second plane for shape 3
template <int SHAPE>
__device__ __forceinline__ size_t addr(long r4, long g, int lane, long np, long ng) {
    const int q = lane >> 4, li = lane & 15;
    if (SHAPE == 0) return ((size_t)(r4 * 4 + q) * np + g * 16 + li) * 16;
    if (SHAPE == 1) return ((size_t)(r4 * 4 + q) * np + g * 16 + li) * 12;
    if (SHAPE == 2) return ((size_t)(r4 * ng + g) * 64 + lane) * 12;
    return 0;
}
// shape 3: r8 = a pair of row groups (8 unit rows = 2 tiles x 4 quarters): hi plane [r8*4+q][np][16 B], lo plane [r8*4+q][np][8 B]
__device__ __forceinline__ size_t addr_hi(long r8, long g, int lane, long np) { return ((size_t)(r8 * 4 + (lane >> 4)) * np + g * 16 + (lane & 15)) * 16; }
__device__ __forceinline__ size_t addr_lo(long r8, long g, int lane, long np) { return ((size_t)(r8 * 4 + (lane >> 4)) * np + g * 16 + (lane & 15)) * 8; }

// MODE 0 write, 1 read, 2 copy, 3 rev mix.  rows4 = number of row groups (4 unit rows each), even.
template <int MODE, int SHAPE, int DEPTH>
__global__ __launch_bounds__(512) void stream(char* a, char* b, char* c, char* d, char* a2, char* c2, char* d2, const char* cf32,
                                              long np, int rows4, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long ng = np / 16;
    const long g0 = ng * blockIdx.x / gridDim.x, g1 = ng * (blockIdx.x + 1) / gridDim.x;
    float acc = 0.f;
    for (long g = g0 + wave; g < g1; g += 8) {
        if (SHAPE != 3) {
            for (int r = 0; r + DEPTH - 1 < rows4; r += DEPTH) {
                f32x4 v4[DEPTH], w4[DEPTH]; f32x3 v3[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const size_t o = addr<SHAPE>(r + u, g, lane, np, ng);
                    if (MODE == 1 || MODE == 2 || MODE == 3) { if (SHAPE == 0) v4[u] = ld16(a + o); else v3[u] = ld12(a + o); }
                    if (MODE == 3) w4[u] = ld16(cf32 + addr<0>(r + u, g, lane, np, ng));
                }
                if (MODE != 0) waitall();
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const size_t o = addr<SHAPE>(r + u, g, lane, np, ng);
                    if (MODE == 1) acc += SHAPE == 0 ? v4[u][0] + v4[u][3] : v3[u][0] + v3[u][2];
                    if (MODE == 0) { if (SHAPE == 0) st16(c + o, f32x4{1.f, 2.f, 3.f, (float)r}); else st12(c + o, f32x3{1.f, 2.f, (float)r}); }
                    if (MODE == 2) { if (SHAPE == 0) st16(c + o, v4[u]); else st12(c + o, v3[u]); }
                    if (MODE == 3) {
                        if (SHAPE == 0) { st16(c + o, v4[u] * w4[u]); st16(d + o, v4[u] + w4[u]); }
                        else { st12(c + o, v3[u] * w4[u][0]); st12(d + o, v3[u] + w4[u][1]); }
                    }
                }
            }
        } else {
            for (int r = 0; r + DEPTH - 1 < rows4 / 2; r += DEPTH) {      // r counts PAIRS of row groups
                f32x4 vh[DEPTH], w4a[DEPTH], w4b[DEPTH]; f32x2 vl[DEPTH];
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    if (MODE != 0) { vh[u] = ld16(a + addr_hi(r + u, g, lane, np)); vl[u] = ld8(a2 + addr_lo(r + u, g, lane, np)); }
                    if (MODE == 3) { w4a[u] = ld16(cf32 + addr<0>(2 * (r + u), g, lane, np, ng)); w4b[u] = ld16(cf32 + addr<0>(2 * (r + u) + 1, g, lane, np, ng)); }
                }
                if (MODE != 0) waitall();
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const size_t oh = addr_hi(r + u, g, lane, np), ol = addr_lo(r + u, g, lane, np);
                    if (MODE == 1) acc += vh[u][0] + vh[u][3] + vl[u][1];
                    if (MODE == 0) { st16(c + oh, f32x4{1.f, 2.f, 3.f, (float)r}); st8(c2 + ol, f32x2{1.f, (float)r}); }
                    if (MODE == 2) { st16(c + oh, vh[u]); st8(c2 + ol, vl[u]); }
                    if (MODE == 3) {
                        st16(c + oh, vh[u] * w4a[u]); st8(c2 + ol, vl[u] * w4b[u][0]);
                        st16(d + oh, vh[u] + w4a[u]); st8(d2 + ol, vl[u] + w4b[u][1]);
                    }
                }
            }
        }
    }
    if (acc == 123.456f) *sink = acc;
}

static char *A, *B, *C, *D, *A2, *C2, *D2, *CF; static float* sink;
template <int MODE, int SHAPE, int DEPTH>
void run(const char* name, long np, int rows4) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-34s ", name);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream<MODE, SHAPE, DEPTH>), dim3(256), dim3(512), 0, 0, A, B, C, D, A2, C2, D2, CF, np, rows4, sink);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(hipGetLastError())); exit(1); }
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MODE, SHAPE, DEPTH>), dim3(256), dim3(512), 0, 0, A, B, C, D, A2, C2, D2, CF, np, rows4, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double unit = (double)rows4 * 4 * np;                  // granules of 4 values
    const double bpg = SHAPE == 0 ? 16 : 12;
    const double bytes = MODE == 0 || MODE == 1 ? unit * bpg : MODE == 2 ? 2 * unit * bpg : unit * (16 + 3 * bpg);
    printf("%8.3f ms  %6.2f TB/s  (%.2f GB)\n", ms, bytes / (ms * 1e-3) / 1e12, bytes / 1e9);
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); exit(1); }
}

int main() {
    // one stash array of the 8x256 / 100 096-column step: 8 layers x 64 feature quads = 512 unit rows = 128 row groups
    const long np = 100096; const int rows4 = 128;
    const size_t bytes = (size_t)rows4 * 4 * np * 16;
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipError_t me = hipSuccess;
    auto M = [&](char** p, size_t n) { if (hipMalloc(p, n) != hipSuccess) me = hipErrorOutOfMemory; };
    M(&A, bytes); M(&B, bytes); M(&C, bytes); M(&D, bytes); M(&CF, bytes); M(&A2, bytes / 2); M(&C2, bytes / 2); M(&D2, bytes / 2);
    if (hipMalloc(&sink, 4) != hipSuccess || me != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    hipMemset(A, 0, bytes); hipMemset(B, 0, bytes); hipMemset(CF, 0, bytes); hipMemset(A2, 0, bytes / 2);
    printf("A %p B %p C %p D %p CF %p A2 %p C2 %p D2 %p\n", A, B, C, D, CF, A2, C2, D2);
    printf("512 unit rows x %ld columns (f32: %.2f GB per array)\n", np, bytes / 1e9);
#define ALL(M, mname)                                                \
    run<M, 0, 4>(mname " f32  (16 B, rows)      d4", np, rows4);      \
    run<M, 0, 8>(mname " f32  (16 B, rows)      d8", np, rows4);      \
    run<M, 1, 4>(mname " p24r (12 B, rows)      d4", np, rows4);      \
    run<M, 1, 8>(mname " p24r (12 B, rows)      d8", np, rows4);      \
    run<M, 2, 4>(mname " p24t (12 B, tile-major) d4", np, rows4);     \
    run<M, 2, 8>(mname " p24t (12 B, tile-major) d8", np, rows4);     \
    run<M, 3, 2>(mname " p24o (16+8 B planes)   d2", np, rows4);      \
    run<M, 3, 4>(mname " p24o (16+8 B planes)   d4", np, rows4);
    ALL(0, "write") ALL(1, "read ") ALL(2, "copy ") ALL(3, "rev  ")
    return 0;
}
