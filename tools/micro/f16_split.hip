// Facts the fp16 hi/lo split ("fp16x3") depends on, checked on the device:
//  1. does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs (or flush them)?
//  2. does v_cvt_pk_f16_f32 round to nearest-even and produce subnormals?
//  3. hi + lo of the two-piece split: residual |v - hi - lo| / |v| over random values, with and without scaling
//  4. rate: 16x16x32 f16 MFMA against the bf16 one (same loop)
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/f16_split.hip -o dbg/f16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void mfma_sub(float* out, float aval, float bval) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.f; b[i] = (_Float16)0.f; }
    // A[row][k]: lane = (k/8)<<4 | row ; B[k][col]: lane = (k/8)<<4 | col.  Put one non-zero at k = 0 of every row / col.
    if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void cvt_test(const float* in, float* hi, float* lo, int n, float sc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = in[i] * sc;
    const f32x2 tv = {t, t};
    const f16x2 h = __builtin_convertvector(tv, f16x2);
    const float hf = (float)h[0];
    const float r = t - hf;
    const f32x2 rv = {r, r};
    const f16x2 l = __builtin_convertvector(rv, f16x2);
    hi[i] = hf; lo[i] = (float)l[0];
}
template <int F16>
__global__ __launch_bounds__(512) void rate(float* out, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    f16x8 a; bf16x8 ab;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(threadIdx.x + i); ab[i] = (__bf16)(float)(threadIdx.x + i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (F16) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, c3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* d; hipMalloc(&d, 1 << 22);
    float h;
    const float sub = ldexpf(1.f, -24), smalln = ldexpf(1.f, -14);
    struct { float a, b; const char* what; } cases[] = {
        {sub, 1024.f, "A = 2^-24 (smallest subnormal), B = 2^10 -> 2^-14 = 6.1035e-05 if honoured"},
        {1024.f, sub, "A = 2^10, B = 2^-24 -> 6.1035e-05 if honoured"},
        {sub * 3, 1.f, "A = 3 * 2^-24, B = 1 -> 1.788e-07 if honoured"},
        {smalln, smalln, "A = B = 2^-14 (normal) -> 2^-28 = 3.7253e-09"},
        {sub, sub, "A = B = 2^-24 -> 2^-48 = 3.5527e-15 if honoured"}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(mfma_sub, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("mfma_f16: %-80s got %.6e\n", c.what, h);
    }
    const int n = 1 << 16;
    float *in = (float*)malloc(n * 4), *hi = (float*)malloc(n * 4), *lo = (float*)malloc(n * 4);
    float *din, *dhi, *dlo; hipMalloc(&din, n * 4); hipMalloc(&dhi, n * 4); hipMalloc(&dlo, n * 4);
    srand(1);
    for (int range = 0; range < 3; ++range) {
        for (int i = 0; i < n; ++i) {
            const double u = rand() / (double)RAND_MAX, e = rand() / (double)RAND_MAX;
            in[i] = (float)((2 * u - 1) * (range == 0 ? 1.0 : range == 1 ? pow(2.0, -20.0 * e) : pow(2.0, -30.0 * e)));
        }
        hipMemcpy(din, in, n * 4, hipMemcpyHostToDevice);
        for (int s = 0; s < 2; ++s) {
            const float sc = s ? 32768.f : 1.f;
            hipLaunchKernelGGL(cvt_test, dim3(n / 256), dim3(256), 0, 0, din, dhi, dlo, n, sc);
            hipMemcpy(hi, dhi, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lo, dlo, n * 4, hipMemcpyDeviceToHost);
            double maxrel = 0, maxabs = 0; int nsub = 0, rne_bad = 0;
            for (int i = 0; i < n; ++i) {
                const double t = (double)in[i] * sc, res = t - hi[i] - lo[i];
                if (fabs(res) > maxabs) maxabs = fabs(res);
                if (t != 0 && fabs(res / t) > maxrel) maxrel = fabs(res / t);
                if (lo[i] != 0 && fabsf(lo[i]) < ldexpf(1.f, -14)) ++nsub;
                if (fabs((float)(_Float16)(float)t - hi[i]) != 0) ++rne_bad;   // host conversion is round-to-nearest-even
            }
            printf("split range %d scale %g: max |v-hi-lo|/|v| %.3e  max abs %.3e (x 2^-25 = %.3f)  subnormal lo pieces %d  hi != host RNE %d\n",
                   range, sc, maxrel, maxabs, maxabs / ldexp(1.0, -25), nsub, rne_bad);
        }
    }
    for (int f16 = 0; f16 < 2; ++f16) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 20000;
        if (f16) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, d, 10); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, d, 10);
        hipEventRecord(e0, 0);
        if (f16) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, d, iters); else hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, d, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * iters * 24 * 16384.0;
        printf("%s 16x16x32 MFMA, 256 WGs x 8 waves: %.1f TF\n", f16 ? "f16 " : "bf16", flop / (ms * 1e-3) / 1e12);
    }
    return 0;
}
