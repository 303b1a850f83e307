// Do the vector ALU instructions of one wave issue beside the MFMAs of its SIMD partner (wave w + 4 of a 512-thread
// workgroup)?  Waves 0-3: VALU loop; waves 4-7: MFMA loop; timed alone and together.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/coissue.hip -o dbg/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int mode, int iters, int nvalu, int kind) {
    __shared__ __attribute__((aligned(16))) char lds[49152];
    const int wave = threadIdx.x >> 6;
    const bool domfma = wave >= 4;
    float acc = threadIdx.x;
    if (domfma && (mode & 2)) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)1.0f; }
        if constexpr (SHAPE == 16) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            if (mode & 4) {                                       // fragments from LDS, one tile (6 MFMAs) ahead
                const char* bp = lds + (threadIdx.x & 63) * 16;
                for (int it = 0; it < iters; ++it) {
                    bf16x8 f0 = *reinterpret_cast<const bf16x8*>(bp), f1 = *reinterpret_cast<const bf16x8*>(bp + 1024), f2 = *reinterpret_cast<const bf16x8*>(bp + 2048);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bf16x8 a0 = f0, a1 = f1, a2 = f2;
                        if (u < 3) { f0 = *reinterpret_cast<const bf16x8*>(bp + (u + 1) * 3072); f1 = *reinterpret_cast<const bf16x8*>(bp + (u + 1) * 3072 + 1024); f2 = *reinterpret_cast<const bf16x8*>(bp + (u + 1) * 3072 + 2048); }
                        __builtin_amdgcn_sched_barrier(0x76);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, c0, 0, 0, 0);
                    }
                }
            } else
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
                    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
                }
            }
            acc = c0[0] + c1[1] + c2[2] + c3[3];
        } else {
            f32x16 c0, c1;
            for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                }
            }
            acc = c0[0] + c1[1];
        }
    } else if (!domfma && (mode & 1)) {
        float x0 = acc, x1 = acc + 1, x2 = acc + 2, x3 = acc + 3, x4 = acc + 4, x5 = acc + 5, x6 = acc + 6, x7 = acc + 7;
#define F8 "v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\t" \
           "v_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7\n\t"
#define P8 "v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3\n\t" \
           "v_pk_fma_f32 %4, %4, %4, %4\n\tv_pk_fma_f32 %5, %5, %5, %5\n\tv_pk_fma_f32 %6, %6, %6, %6\n\tv_pk_fma_f32 %7, %7, %7, %7\n\t"
#define A8 "v_and_b32 %0, %0, %1\n\tv_and_b32 %1, %1, %2\n\tv_and_b32 %2, %2, %3\n\tv_and_b32 %3, %3, %4\n\t" \
           "v_and_b32 %4, %4, %5\n\tv_and_b32 %5, %5, %6\n\tv_and_b32 %6, %6, %7\n\tv_and_b32 %7, %7, %0\n\t"
#define D8 "v_dot2_f32_bf16 %0, %1, %2, %0\n\tv_dot2_f32_bf16 %1, %2, %3, %1\n\tv_dot2_f32_bf16 %2, %3, %4, %2\n\tv_dot2_f32_bf16 %3, %4, %5, %3\n\t" \
           "v_dot2_f32_bf16 %4, %5, %6, %4\n\tv_dot2_f32_bf16 %5, %6, %7, %5\n\tv_dot2_f32_bf16 %6, %7, %0, %6\n\tv_dot2_f32_bf16 %7, %0, %1, %7\n\t"
#define M8 "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %1, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %2, %2, %3, %4 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %3, %3, %4, %5 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
           "v_fma_mix_f32 %4, %4, %5, %6 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %5, %5, %6, %7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %6, %6, %7, %0 op_sel_hi:[0,0,1]\n\tv_fma_mix_f32 %7, %7, %0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define C8 "v_cvt_pk_f16_f32 %0, %0, %1\n\tv_cvt_pk_f16_f32 %1, %1, %2\n\tv_cvt_pk_f16_f32 %2, %2, %3\n\tv_cvt_pk_f16_f32 %3, %3, %4\n\t" \
           "v_cvt_pk_f16_f32 %4, %4, %5\n\tv_cvt_pk_f16_f32 %5, %5, %6\n\tv_cvt_pk_f16_f32 %6, %6, %7\n\tv_cvt_pk_f16_f32 %7, %7, %0\n\t"
#define B8 "v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_cvt_pk_bf16_f32 %1, %1, %2\n\tv_cvt_pk_bf16_f32 %2, %2, %3\n\tv_cvt_pk_bf16_f32 %3, %3, %4\n\t" \
           "v_cvt_pk_bf16_f32 %4, %4, %5\n\tv_cvt_pk_bf16_f32 %5, %5, %6\n\tv_cvt_pk_bf16_f32 %6, %6, %7\n\tv_cvt_pk_bf16_f32 %7, %7, %0\n\t"
#define H8 "v_cvt_f32_f16 %0, %1\n\tv_cvt_f32_f16 %1, %2\n\tv_cvt_f32_f16 %2, %3\n\tv_cvt_f32_f16 %3, %4\n\t" \
           "v_cvt_f32_f16 %4, %5\n\tv_cvt_f32_f16 %5, %6\n\tv_cvt_f32_f16 %6, %7\n\tv_cvt_f32_f16 %7, %0\n\t"
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 y0 = {x0, x1}, y1 = {x1, x2}, y2 = {x2, x3}, y3 = {x3, x4}, y4 = {x4, x5}, y5 = {x5, x6}, y6 = {x6, x7}, y7 = {x7, x0};
        for (int it = 0; it < iters; ++it) {
            for (int u = 0; u < nvalu; ++u) {                      // 48 instructions per trip
                if (kind == 0) asm volatile(F8 F8 F8 F8 F8 F8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 1) asm volatile(P8 P8 P8 P8 P8 P8 : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
                else if (kind == 3) asm volatile(D8 D8 D8 D8 D8 D8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 4) asm volatile(M8 M8 M8 M8 M8 M8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 5) asm volatile(C8 C8 C8 C8 C8 C8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 6) asm volatile(B8 B8 B8 B8 B8 B8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 7) asm volatile(H8 H8 H8 H8 H8 H8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                else if (kind == 2) asm volatile(A8 A8 A8 A8 A8 A8 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
        x0 += y0.x + y1.y + y2.x + y3.y + y4.x + y5.y + y6.x + y7.y;
        x0 += x4 + x5 + x6 + x7;
        acc = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <int SHAPE>
static float run(float* d, int mode, int iters, int nvalu, int kind = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, mode, iters, nvalu, kind);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(512), 0, 0, d, mode, iters, nvalu, kind);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    // per iteration: 24 (16x16x32) or 12 (32x32x16) MFMAs = 384 matrix-pipe cycles; VALU: 4 * nvalu instructions
    const char* kn[8] = {"v_fma_f32", "v_pk_fma_f32", "v_and_b32", "v_dot2_f32_bf16", "v_fma_mix_f32", "v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32", "v_cvt_f32_f16"};
    for (int kind = 0; kind < 8; ++kind)
        for (int nv : {1, 2}) {
            printf("%s x %d per 24 MFMAs (384 matrix cycles):\n", kn[kind], 48 * nv);
            printf("  16x16x32 independent acc: valu alone %.3f ms, mfma alone %.3f ms, both %.3f ms\n", run<16>(d, 1, iters, nv, kind), run<16>(d, 2, iters, nv, kind), run<16>(d, 3, iters, nv, kind));
            printf("  16x16x32 LDS frags + chains: valu alone %.3f ms, mfma alone %.3f ms, both %.3f ms\n", run<16>(d, 1, iters, nv, kind), run<16>(d, 6, iters, nv, kind), run<16>(d, 7, iters, nv, kind));
        }
    printf("  32x32x16: mfma alone %.3f ms, both(v_fma x48) %.3f ms\n", run<32>(d, 2, iters, 1, 0), run<32>(d, 3, iters, 1, 0));
    return 0;
}
