// Check of the LDS image the 24-bit weight-gradient GEMM stages its operands in (dudf_wgrad.hip, P24 build), before the kernel
// itself is built around it:
//   * producer lane (q = lane >> 4, li = lane & 15) of wave pw holds, for tiles T = 4 pw + t, the 4 features 16 T + 4 q + e of
//     column li of a 16-column stage and writes them as 4 fp16 = 8 bytes (ds_write_b64) into  image[k = li][feature]  with
//     rows of 576 bytes (512 + 64: four consecutive rows start 16 banks apart) and the 8-byte unit index XORed in its low
//     three bits with (k >> 1) & 7 (16 lanes x 8 bytes of one service group then cover all 32 write banks once);
//   * consumer lane l wants, for a 32-feature block b, feature 32 b + (l & 31) at the 8 columns 8 (l >> 5) .. + 7: two
//     ds_read_b64_tr_b16, each delivering 4 rows (columns k) x 16 features column-major to a 16-lane group;
//   * X (32 x 16) . Y^T (16 x 32) on v_mfma_f32_32x32x16_f16 must equal the host's sum over the 16 columns.
// Prints the max abs error (fp16-exact inputs: must be 0) and the LDS bank-conflict-free claim is left to SQ_LDS_BANK_CONFLICT.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/tr_image.hip -o dbg/tr_image
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROW = 576, PIECE = 16 * ROW;

__global__ __launch_bounds__(512) void k(const _Float16* __restrict__ X, const _Float16* __restrict__ Y, float* __restrict__ out) {
    // X, Y: [256 features][16 columns] fp16.  out: [8 x-blocks][8 y-blocks][32][32]
    __shared__ __attribute__((aligned(16))) char img[2 * PIECE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, li = lane & 15;
    const int oper = wave >> 2, pw = wave & 3;
    const _Float16* S = oper ? Y : X;
    const int s_li = (li >> 1) & 7;
    for (int t = 0; t < 4; ++t) {
        const int T = 4 * pw + t;
        f16x4 v;
        for (int e = 0; e < 4; ++e) v[e] = S[(16 * T + 4 * q + e) * 16 + li];
        const int unit = 4 * T + q;
        *reinterpret_cast<f16x4*>(img + oper * PIECE + li * ROW + ((unit & ~7) | ((unit ^ s_li) & 7)) * 8) = v;
    }
    __syncthreads();
    // consumer: wave w computes x-block bx = w, all 8 y-blocks
    const int kh = lane >> 5, gi = (lane >> 4) & 1, qp = (lane & 15) >> 2, p = lane & 3;
    auto frag = [&](int oper_, int b) -> f16x8 {
        f16x8 r;
        for (int h = 0; h < 2; ++h) {
            const int kk = 8 * kh + 4 * h + qp;                         // the row this lane addresses
            const int unit = 8 * b + 4 * gi + p;
            const char* a = img + oper_ * PIECE + kk * ROW + ((unit & ~7) | ((unit ^ ((kk >> 1) & 7)) & 7)) * 8;
            const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
            const f16x4 f = __builtin_bit_cast(f16x4, v);
            for (int e = 0; e < 4; ++e) r[4 * h + e] = f[e];
        }
        return r;
    };
    const f16x8 a = frag(0, wave);
    for (int by = 0; by < 8; ++by) {
        const f16x8 b = frag(1, by);
        f32x16 c;
        for (int e = 0; e < 16; ++e) c[e] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        // accumulator layout of the 32x32 MFMA as dudf_wgrad.hip writes it out: row o = (e & 3) + 8 (e >> 2) + 4 (lane >> 5), col i = lane & 31
        for (int e = 0; e < 16; ++e) {
            const int o = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), i = lane & 31;
            out[((wave * 8 + by) * 32 + o) * 32 + i] = c[e];
        }
    }
}

int main() {
    std::vector<_Float16> X(256 * 16), Y(256 * 16);
    srand(1);
    for (auto& v : X) v = (_Float16)((rand() % 65) - 32);
    for (auto& v : Y) v = (_Float16)((rand() % 33) - 16);
    _Float16 *dX, *dY; float* dO;
    hipMalloc(&dX, X.size() * 2); hipMalloc(&dY, Y.size() * 2); hipMalloc(&dO, 64 * 1024 * 4);
    hipMemcpy(dX, X.data(), X.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dY, Y.data(), Y.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, dX, dY, dO);
    std::vector<float> O(64 * 1024);
    if (hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED\n"); return 1; }
    double worst = 0; long bad = 0;
    for (int fx = 0; fx < 256; ++fx)
        for (int fy = 0; fy < 256; ++fy) {
            float ref = 0;
            for (int c = 0; c < 16; ++c) ref += (float)X[fx * 16 + c] * (float)Y[fy * 16 + c];
            const float got = O[(((fx >> 5) * 8 + (fy >> 5)) * 32 + (fx & 31)) * 32 + (fy & 31)];
            const double e = fabs((double)got - ref);
            if (e > worst) worst = e;
            if (e != 0) ++bad;
        }
    printf("tr image: max abs error %g, %ld of 65536 entries differ (expected 0 / 0)\n", worst, bad);
    return bad != 0;
}
