// How many vector-ALU instructions of the SAME wave fit in the shadow of its own MFMAs (one wave per SIMD, gfx950)?
// A 256-thread workgroup per CU; every wave runs   repeat { NM independent v_mfma_f32_32x32x16_f16 ; NV independent v_fma_f32 }   in program
// order with scheduling fences, and reports shader cycles per repeat.  (Round 6: the four-wave weight-gradient kernel, option wgrad_family = 3,
// dealt its operand split out between its MFMAs and measured NO overlap — this is the isolated question.)
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/selfissue.hip -o dbg/selfissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int DEP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((threadIdx.x + i) & 7); b[i] = (_Float16)1.0f; }
    f32x16 c[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) c[t][i] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(threadIdx.x + i);
    const float s = 1.0000001f, o = 0.5f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {                       // four groups per repeat: NM MFMAs then NV VALU each
#pragma unroll
            for (int m = 0; m < NM; ++m) c[DEP ? 0 : ((g * NM + m) & 3)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[DEP ? 0 : ((g * NM + m) & 3)], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], s, o);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int t = 0; t < 4; ++t) acc += c[t][0] + c[t][5];
    for (int i = 0; i < 8; ++i) acc += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NM, int NV, int DEP>
void run(float* out, unsigned long long* cyc, const char* what) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NM, NV, DEP>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NM, NV, DEP>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / iters / 4.0;            // s_memtime ticks per group (the counter runs at the shader clock on this chip: tools/micro/hwid)
    printf("%-34s %3d MFMA + %3d VALU per group: %7.1f clocks per group = %5.1f per MFMA (+ %5.1f over %d x 32)\n", what, NM, NV, per, per / NM, per - 32.0 * NM, NM);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<3, 0, 0>(out, cyc, "independent MFMAs");
    run<3, 6, 0>(out, cyc, "independent MFMAs");
    run<3, 12, 0>(out, cyc, "independent MFMAs");
    run<3, 18, 0>(out, cyc, "independent MFMAs");
    run<3, 24, 0>(out, cyc, "independent MFMAs");
    run<3, 36, 0>(out, cyc, "independent MFMAs");
    run<1, 0, 0>(out, cyc, "independent MFMAs");
    run<1, 4, 0>(out, cyc, "independent MFMAs");
    run<1, 6, 0>(out, cyc, "independent MFMAs");
    run<1, 8, 0>(out, cyc, "independent MFMAs");
    run<1, 12, 0>(out, cyc, "independent MFMAs");
    run<3, 0, 1>(out, cyc, "DEPENDENT MFMAs (one accumulator)");
    run<3, 12, 1>(out, cyc, "DEPENDENT MFMAs (one accumulator)");
    run<12, 0, 0>(out, cyc, "independent MFMAs");
    run<12, 44, 0>(out, cyc, "independent MFMAs");
    run<0, 12, 0>(out, cyc, "VALU only");
    return 0;
}
