// Weight gradients of the DiffUDF training step: the part of `train_loss.backward()` (reference
// train.py:221) that contracts over POINTS.  SURVEY.md Appendix A.5:
//
//   dW_l[o][i] = sum_p  q_l[o][p] * A_{l-1}[i][p]  +  zbar_l[o][p] * s_{l-1}[i][p]        (l = 2..L)
//   db_l[o]    = sum_p  zbar_l[o][p]
//   dW_1[o][d] = sum_p  q_1[o][p] * gbar[p][d]  +  zbar_1[o][p] * x[p][d] ;  db_1 = sum_p zbar_1
//   dW_out[f]  = sum_p  A_L[f][p] + ybar[p] * s_L[f][p] ;                     db_out = sum_p ybar
//
// The sweeps leave q, A, zbar, s in HBM as [layer][feature/4][point][4] (dudf_internal.h).  That is
// the K-major layout a GEMM whose K dimension is the point index wants, so the hidden layers are one
// v_mfma_f32_32x32x2_f32 GEMM per layer (M = N = H, K = 2*n points) split over point ranges:
// a workgroup owns the whole HxH tile of one layer for a range of points, stages 32 points of the two
// operands through LDS per step ([fq][33][4] floats: the 33 makes the per-feature b32 reads
// conflict-free), and adds its partial tile into dtheta with float atomics at the end (each atomic
// wave-instruction is two 128-byte row segments — the full-rate shape; 256 KB per workgroup).
// The bias gradient rides along on the VALU beside the MFMAs: running row sums of zbar over the value columns
// (every column on the plain range, channel 0 of each quad on the Hessian range; padding columns hold zeros).
// The two thin layers (3 inputs / 1 output) are a bandwidth-bound VALU reduction.
#include "dudf_internal.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

template <int H> struct WG;
template <> struct WG<256> { static constexpr int WO = 4, WI = 2, MT = 2, NTL = 4; };
template <> struct WG<128> { static constexpr int WO = 4, WI = 2, MT = 1, NTL = 2; };
template <> struct WG<64>  { static constexpr int WO = 2, WI = 2, MT = 1, NTL = 1; };
template <> struct WG<32>  { static constexpr int WO = 1, WI = 1, MT = 1, NTL = 1; };

constexpr int KT = 32;          // points per LDS stage
constexpr int KTP = 33;         // padded

struct WgradArgs {
    const float *Q, *A, *Z, *S;         // stash arrays
    int64_t ncol_h;                     // columns [0, ncol_h) are Hessian quads: only channel 0 (col % 4 == 0) carries the bias
    float* dtheta;
    int64_t np, stash_layer, off_hid, hid_stride;
    int steps_total;                    // np / KT
    int L;
    int have_g;                         // 0: loss without df/dx terms (loss_s2) -> only the zbar*s pair
    int j0, nj;                         // hidden matrices [j0, j0 + nj) of this launch (matrix j = layer l = j + 2); grid.x = nj
    int Hs;                             // real layer width; the kernels' template H is the output TILE (<= 256):
                                        // blockIdx.z walks the (Hs/H)^2 tiles of a wider layer
    int remap_nsplit;                   // > 0: 1-D grid, the output tiles of one (layer, column split) share an XCD (see the body)
    unsigned long long* clk;            // profiling: (shader clock, 100 MHz reference) ticks of workgroup (0,0,0), or nullptr
    const unsigned* amax;               // [4][L] bit patterns of max |q_l|, |A_l|, |zbar_l| over all columns (fp16x3 kernel)
    int p24;                            // the operands are 24-bit tile-major arrays (DudfLayout::p24 bit 0: fixed point relative to a column scale)
    const float *fxS, *fxQ, *fxA, *fxZ; // p24: [L][np] — per layer and column the power of two 2^E of that array's values (dudf_sweep_common.h fx24_pack)
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <int H>
__device__ __forceinline__ void wgrad_hidden_body(const WgradArgs& a) {
    using W = WG<H>;
    constexpr int NTHR = 64 * W::WO * W::WI;
    constexpr int FQ = H / 4;                       // feature quads per operand
    constexpr int TILE = FQ * KTP * 4;              // floats per staged operand
    constexpr int F4 = FQ * KT;                     // float4 per operand per stage
    constexpr int NLD = (F4 + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [2 buffers][X tile | Y tile]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave / W::WI, wi = wave % W::WI;
    const int l32 = lane & 31, hh = lane >> 5;
    const int j = blockIdx.x + a.j0;                // hidden matrix index: layer l = j + 2
    const int nsplit = gridDim.y;
    const int s0 = (int)((int64_t)a.steps_total * blockIdx.y / nsplit);
    const int s1 = (int)((int64_t)a.steps_total * (blockIdx.y + 1) / nsplit);
    const int tz = a.Hs / H;                        // output tiles per side
    const int o_off = (blockIdx.z / tz) * H, i_off = (blockIdx.z % tz) * H;
    const int64_t xrow = (int64_t)(o_off / 4) * a.np * 4, yrow = (int64_t)(i_off / 4) * a.np * 4;

    f32x16 acc[W::MT][W::NTL];
    float bsum[W::MT];                              // bias gradient: running row sums of zbar (VALU, beside the MFMAs)
#pragma unroll
    for (int m = 0; m < W::MT; ++m) {
#pragma unroll
        for (int n = 0; n < W::NTL; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        bsum[m] = 0.f;
    }

    // pair 0: X = q_l (layer index j+1), Y = A_{l-1} (index j);  pair 1: X = zbar_l, Y = s_{l-1}
    const int64_t lstride = a.stash_layer;
    const float* X0 = a.Q + (int64_t)(j + 1) * lstride + xrow;
    const float* X1 = a.Z + (int64_t)(j + 1) * lstride + xrow;
    const float* Y0 = a.A + (int64_t)j * lstride + yrow;
    const float* Y1 = a.S + (int64_t)j * lstride + yrow;

    f32x4 rx[NLD], ry[NLD];
    auto issue = [&](int pair, int step) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int f = tid + NTHR * u;
            if (F4 % NTHR == 0 || f < F4) {
                const int fq = f / KT, pt = f % KT;
                const int64_t off = ((int64_t)fq * a.np + (int64_t)step * KT + pt) * 4;
                rx[u] = *reinterpret_cast<const f32x4*>((pair ? X1 : X0) + off);
                ry[u] = *reinterpret_cast<const f32x4*>((pair ? Y1 : Y0) + off);
            }
        }
    };
    auto commit = [&](float* buf) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int f = tid + NTHR * u;
            if (F4 % NTHR == 0 || f < F4) {
                const int fq = f / KT, pt = f % KT;
                *reinterpret_cast<f32x4*>(buf + (fq * KTP + pt) * 4) = rx[u];
                *reinterpret_cast<f32x4*>(buf + TILE + (fq * KTP + pt) * 4) = ry[u];
            }
        }
    };

    const int npair = a.have_g ? 2 : 1;
    const int nit = npair * (s1 - s0);              // (step, pair) iterations
    auto pair_of = [&](int it) { return a.have_g ? (it & 1) : 1; };
    auto step_of = [&](int it) { return s0 + (a.have_g ? (it >> 1) : it); };
    if (nit > 0) {
        issue(pair_of(0), step_of(0));
        commit(lds);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const float* buf = lds + (it & 1) * 2 * TILE;
        if (it + 1 < nit) issue(pair_of(it + 1), step_of(it + 1));      // next stage: global -> registers
        // bias mask of this stage's columns: k index 2*kk+hh is a value column always (plain range) or when
        // (2*kk+hh) % 4 == 0 (Hessian range: stages never straddle the two ranges)
        const float bflag = (pair_of(it) == 1 && wi == 0 && i_off == 0) ? 1.f : 0.f;
        const bool hstage = (int64_t)step_of(it) * KT < a.ncol_h;
        const float f_even = hstage ? (hh == 0 ? bflag : 0.f) : bflag;
        const float f_odd = hstage ? 0.f : bflag;
        const float* xa[W::MT];
        const float* yb[W::NTL];
#pragma unroll
        for (int m = 0; m < W::MT; ++m) {
            const int feat = (wo * W::MT + m) * 32 + l32;
            xa[m] = buf + ((feat >> 2) * KTP) * 4 + (feat & 3);
        }
#pragma unroll
        for (int n = 0; n < W::NTL; ++n) {
            const int feat = (wi * W::NTL + n) * 32 + l32;
            yb[n] = buf + TILE + ((feat >> 2) * KTP) * 4 + (feat & 3);
        }
#pragma unroll
        for (int kk = 0; kk < KT / 2; ++kk) {
            const int pt = 2 * kk + hh;             // MFMA k index = lane>>5
            float av[W::MT], bv[W::NTL];
#pragma unroll
            for (int m = 0; m < W::MT; ++m) av[m] = xa[m][pt * 4];
#pragma unroll
            for (int n = 0; n < W::NTL; ++n) bv[n] = yb[n][pt * 4];
#pragma unroll
            for (int m = 0; m < W::MT; ++m) {
#pragma unroll
                for (int n = 0; n < W::NTL; ++n) acc[m][n] = mfma32(av[m], bv[n], acc[m][n]);
                bsum[m] = fmaf(av[m], (kk & 1) ? f_odd : f_even, bsum[m]);
            }
        }
        // the other buffer was last read in iteration it-1 and every wave is past that iteration's barrier
        if (it + 1 < nit) commit(lds + ((it + 1) & 1) * 2 * TILE);
        __syncthreads();
    }

    // D layout 32x32: col = lane&31 (input index i), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (output index o)
    float* dW = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)o_off * a.Hs + i_off;
    float* dB = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)a.Hs * a.Hs + o_off;
    if (nit > 0) {
#pragma unroll
        for (int m = 0; m < W::MT; ++m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = (wo * W::MT + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
                for (int n = 0; n < W::NTL; ++n) {
                    const int i = (wi * W::NTL + n) * 32 + l32;
                    atomicAdd(dW + (int64_t)o * a.Hs + i, acc[m][n][e]);
                }
            }
            // lane (l32, hh) summed zbar[feature l32] over the points of parity hh
            const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
            if (wi == 0 && hh == 0 && i_off == 0) atomicAdd(dB + (wo * W::MT + m) * 32 + l32, tot);
        }
    }
}
template <int H>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_kernel(WgradArgs a) { wgrad_hidden_body<H>(a); }   // no packed fp32: see dudf_internal.h

// ---- the same GEMM on the bf16 matrix cores, at fp32 accuracy ("bf16x6") ---------------------------------------------
// Every fp32 operand is split EXACTLY into three bf16 pieces v = h + m + l (8+8+8 significand bits); a product a*b is
// the six piece products with combined order <= 2 (hh, hm, mh, hl, lh, mm), each exact in the MFMA and accumulated in
// fp32; the dropped ones (ml, lm, ll) are <= 2^-24 relative, the size of one fp32 rounding.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles, K = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles, K = 2) per 16 points:
// 2.67x less matrix-pipe time for the same 183 GFLOP.  The split (5.5 VALU ops per element, only of the fragments a
// wave consumes) runs beside the MFMAs.  tests/test_hip_parity.py holds the result to the SAME tolerances as fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

struct Split3 { bf16x8 h, m, l; };

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(f32x2 v) {              // one v_cvt_pk_bf16_f32: lo = bf16(v.x), hi = bf16(v.y)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 unpack(unsigned p) {              // the two bf16 back as exact floats
    return f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}

// 8 consecutive columns c0..c0+7 of one feature out of the swizzled [fq][16][4] image (column c sits at slot c ^ swz)
// -> three bf16 fragments, worked on as packed pairs (3 cvt_pk + 4 bit ops + 2 packed subtracts per pair), + the
// masked row sum for the bias gradient
__device__ __forceinline__ Split3 load_split_swz(const float* row, int c0, int swz, float& sum, float f_other) {
    u32x4 hp, mp, lp;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {row[((c0 + 2 * i) ^ swz) * 4], row[((c0 + 2 * i + 1) ^ swz) * 4]};
        const unsigned h = cvt_pk(v);
        const f32x2 r1 = v - unpack(h);
        const unsigned m = cvt_pk(r1);
        const f32x2 r2 = r1 - unpack(m);
        hp[i] = h; mp[i] = m; lp[i] = cvt_pk(r2);
        acc += ((i & 1) == 0 ? v.x : v.x * f_other) + v.y * f_other;     // Hessian range: only columns % 4 == 0 count
    }
    sum = acc;
    Split3 r;
    r.h = __builtin_bit_cast(bf16x8, hp); r.m = __builtin_bit_cast(bf16x8, mp); r.l = __builtin_bit_cast(bf16x8, lp);
    return r;
}

// Staging for this kernel: LDS-DMA (inline asm, see dudf_sweep.hip for why) into a 4-deep ring of 16-column stages.
// The image of one operand is [feature quad][16 columns][4 features] WITHOUT padding (a DMA wave-instruction writes
// 1 KiB linearly); bank conflicts of the per-feature b32 reads are removed by storing column c of quad fq at slot
// c ^ (fq & 7) — applied through the per-lane SOURCE address of the DMA and again on the read.  Three stages are in
// flight behind a counted vmcnt, because at bf16 rates a stage is consumed in ~1.5 us — less than an HBM round trip.
constexpr int KB = 16;          // columns per stage of the bf16 kernel
constexpr int NRING = 4;

template <int H>
__device__ __forceinline__ void wgrad_hidden_bf16_body(const WgradArgs& a) {
    using W = WG<H>;
    constexpr int NW_ = W::WO * W::WI;
    constexpr int FQ = H / 4;
    constexpr int OPER = FQ * KB * 4;                       // floats per operand image
    constexpr int STAGE = 2 * OPER;                         // X image | Y image
    constexpr int PIECES = (FQ * KB) / 64;                  // 1 KiB DMA pieces per operand image
    static_assert(PIECES % NW_ == 0, "every wave stages the same number of pieces");
    constexpr int PPW = PIECES / NW_;                       // pieces per wave and operand
    constexpr int PER_WAVE = 2 * PPW;                       // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) float lds[];     // [NRING][X image | Y image]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / W::WI, wi = wave % W::WI;
    const int l32 = lane & 31, hh = lane >> 5;
    const int j = blockIdx.x + a.j0;
    const int nsplit = gridDim.y;
    const int steps16 = a.steps_total * (KT / KB);
    const int s0 = (int)((int64_t)steps16 * blockIdx.y / nsplit);
    const int s1 = (int)((int64_t)steps16 * (blockIdx.y + 1) / nsplit);
    const int tz = a.Hs / H;
    const int o_off = (blockIdx.z / tz) * H, i_off = (blockIdx.z % tz) * H;
    const int64_t xrow = (int64_t)(o_off / 4) * a.np * 4, yrow = (int64_t)(i_off / 4) * a.np * 4;

    f32x16 acc[W::MT][W::NTL];
    float bsum[W::MT];
#pragma unroll
    for (int m = 0; m < W::MT; ++m) {
#pragma unroll
        for (int n = 0; n < W::NTL; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
        bsum[m] = 0.f;
    }
    const int64_t lstride = a.stash_layer;
    const float* X0 = a.Q + (int64_t)(j + 1) * lstride + xrow;
    const float* X1 = a.Z + (int64_t)(j + 1) * lstride + xrow;
    const float* Y0 = a.A + (int64_t)j * lstride + yrow;
    const float* Y1 = a.S + (int64_t)j * lstride + yrow;

    const int npair = a.have_g ? 2 : 1;
    const int nit = npair * (s1 - s0);
    auto pair_of = [&](int it) { return a.have_g ? (it & 1) : 1; };
    auto step_of = [&](int it) { return s0 + (a.have_g ? (it >> 1) : it); };

    // lane part of the DMA source: granule (fq_in_piece = lane/16, slot = lane%16) holds column slot ^ (fq & 7)
    auto issue = [&](int it) {
        const int pair = pair_of(it);
        const int64_t col0 = (int64_t)step_of(it) * KB;
        float* ring = lds + (it % NRING) * STAGE;
        const float* xs = pair ? X1 : X0;
        const float* ys = pair ? Y1 : Y0;
#pragma unroll
        for (int oper = 0; oper < 2; ++oper) {
#pragma unroll
            for (int v = 0; v < PPW; ++v) {
                const int piece = wave + NW_ * v;           // wave-uniform 1 KiB piece of this operand's image
                const int fq = piece * 4 + (lane >> 4);
                const int col = (lane & 15) ^ (fq & 7);
                const float* src = (oper ? ys : xs) + ((int64_t)fq * a.np + col0 + col) * 4;
                const unsigned dst = __builtin_amdgcn_readfirstlane(
                    (unsigned)(size_t)(__attribute__((address_space(3))) float*)(ring + oper * OPER + piece * 256));
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\t"
                             "s_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(dst) : "memory");    // nt: the stash is read once here
            }
        }
    };
    // prologue: three stages in flight
    for (int it = 0; it < NRING - 1 && it < nit; ++it) issue(it);
    if (nit > 0) {
        if (nit >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        // the ring slot of stage it+3 was read in iteration it-1; every wave is past that iteration's barrier
        if (it + NRING - 1 < nit) issue(it + NRING - 1);
        const float* buf = lds + (it % NRING) * STAGE;
        const float bflag = (pair_of(it) == 1 && wi == 0 && i_off == 0) ? 1.f : 0.f;
        const bool hstage = (int64_t)step_of(it) * KB < a.ncol_h;
        const float f_other = hstage ? 0.f : 1.f;
        const int c0 = 8 * hh;                                    // MFMA k = 8*(lane>>5) + jj -> column c0 + jj
        Split3 af[W::MT];
#pragma unroll
        for (int m = 0; m < W::MT; ++m) {
            const int feat = (wo * W::MT + m) * 32 + l32;
            float rs;
            af[m] = load_split_swz(buf + (feat >> 2) * (KB * 4) + (feat & 3), c0, (feat >> 2) & 7, rs, f_other);
            bsum[m] = fmaf(rs, bflag, bsum[m]);
        }
#pragma unroll
        for (int n = 0; n < W::NTL; ++n) {
            const int feat = (wi * W::NTL + n) * 32 + l32;
            float dummy;
            const Split3 bf = load_split_swz(buf + OPER + (feat >> 2) * (KB * 4) + (feat & 3), c0, (feat >> 2) & 7,
                                             dummy, 1.f);
#pragma unroll
            for (int m = 0; m < W::MT; ++m) {
                f32x16 c = acc[m][n];
                c = mfma_bf16(af[m].m, bf.m, c);              // smallest terms first
                c = mfma_bf16(af[m].l, bf.h, c);
                c = mfma_bf16(af[m].h, bf.l, c);
                c = mfma_bf16(af[m].m, bf.h, c);
                c = mfma_bf16(af[m].h, bf.m, c);
                c = mfma_bf16(af[m].h, bf.h, c);
                acc[m][n] = c;
            }
        }
        // stage it+1 has landed: all but the DMA pieces of the (up to) two younger stages are retired
        if (it + 1 < nit) {
            if (it + 3 < nit) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER_WAVE) : "memory");
            else if (it + 2 < nit) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    float* dW = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)o_off * a.Hs + i_off;
    float* dB = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)a.Hs * a.Hs + o_off;
    if (nit > 0) {
#pragma unroll
        for (int m = 0; m < W::MT; ++m) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = (wo * W::MT + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
                for (int n = 0; n < W::NTL; ++n) {
                    const int i = (wi * W::NTL + n) * 32 + l32;
                    atomicAdd(dW + (int64_t)o * a.Hs + i, acc[m][n][e]);
                }
            }
            const float tot = bsum[m] + __shfl_xor(bsum[m], 32);
            if (wi == 0 && hh == 0 && i_off == 0) atomicAdd(dB + (wo * W::MT + m) * 32 + l32, tot);
        }
    }
}
template <int H>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_bf16_kernel(WgradArgs a) { wgrad_hidden_bf16_body<H>(a); }   // no packed fp32: see dudf_internal.h

// ---- bf16x6 with a COOPERATIVE split (256 x 256 tiles) ----------------------------------------------------------------
// The kernel above is bound by vector-ALU issue, not by the matrix cores: every wave splits the fragments it consumes, so
// an X element is split by 2 waves and a Y element by 4 (7.5 VALU instructions per MFMA, PMC).  Here every element is
// split ONCE: a stage (16 columns x 256 features x 2 operands = 2048 16-byte granules) is dealt out 4 granules per lane
// — four consecutive columns of one feature quad, 64 contiguous bytes of the stash — loaded straight into registers
// (non-temporal; no raw copy in LDS), split, and written as bf16 pieces into an LDS image laid out in MFMA-FRAGMENT
// order ([operand][piece][32-feature block][column half][feature][8 columns]: a fragment is two lane-linear 512-byte
// halves, conflict-free for ds_read_b128; the halves are 528 bytes apart so that the 8-byte writes do not collide).  The image is double buffered (2 x 48 KiB); one barrier per stage; the next stage's split runs in
// front of this stage's MFMAs while the stage after that is in flight in registers.
// VAR bit 0: producer lanes are dealt out so that each ds_write_b64 wave-instruction is bank-conflict free (a 16-lane
//   service group spans offsets 16 a + 2 b + 4 c + 8 d dwords: fq bit 0 | column-group bit 0 | column half | fq bit 3);
//   the plain (tid >> 2) order put feature quads 0/2 and 1/3 of a group on the same banks (PMC r01: 29 % of LDS cycles).
// VAR bit 1: the split of the next stage is cut into four feature slices issued BETWEEN the four MFMA groups of this
//   stage instead of in front of them: the two waves of a SIMD then leave the post-barrier lockstep (both splitting,
//   matrix core idle) after the first slice — one wave's slice runs beside the other's MFMAs.
#ifndef DUDF_WG_NT
// Cache policy of the staging loads: no non-temporal hint.  A lane's four loads sit 64 bytes apart, so every 128-byte line is
// touched by two different instructions.  While round 3's workspace layout had the stash 64 bytes off the line grid (fixed:
// dudf_make_layout), the hinted loads fetched 4.17 GB per launch against 2.87 GB of operands and 3.04 GB without the hint
// (profiles/r03_wgrad_nt.txt); on the aligned layout both forms fetch 2.87 GB and take the same time.  A/B: -DDUDF_WG_NT_ON=1.
#if DUDF_WG_NT_ON
#define DUDF_WG_NT " nt"
#else
#define DUDF_WG_NT ""
#endif
#endif
#ifndef DUDF_WG_HREL
#define DUDF_WG_HREL 2             // flag-synchronised variant: hand the matrix pipe over this many MFMA groups before the end of a stage
#endif
#ifndef DUDF_WGRAD_DBG
#define DUDF_WGRAD_DBG 0           // timing experiments only (wrong results): 1 no loads, 2 no barrier, 4 no MFMA, 8 no split,
#endif                             // 16 no LDS fragment reads, 32 stage stamps, 64 no output atomics
#if DUDF_WGRAD_DBG & 32
// phase stamps of the LAST steady-state stage of one workgroup (timing experiments): [wave][stamp], shader-clock cycles
__device__ unsigned long long g_wstamp[8][12];
extern "C" int dudf_dbg_wstamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamp), sizeof(g_wstamp)); }
#define DUDF_WSTAMP(i) do { if constexpr (HOT) { __builtin_amdgcn_sched_barrier(0); if (it == DUDF_WSTAMP_IT) wst[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#ifndef DUDF_WSTAMP_IT
#define DUDF_WSTAMP_IT 60
#endif
#else
#define DUDF_WSTAMP(i) do { } while (0)
#endif
// SP = 1: the fp16 hi/lo split ("fp16x3": products hi*hi + hi*lo + lo*hi, see dudf_sweep_bf16.hip) — half the MFMAs, two
// pieces instead of three to convert, write and read back.  fp16's range is bought with per-layer powers of two: the
// sweeps leave max |q_l|, |A_l|, |zbar_l| over all columns in `amax` (|s_l| <= 1); each operand is scaled so that its
// largest element lands below 2^15, the two pairs of a layer (q A^T and zbar s^T accumulate into the same registers) are
// brought to a common product scale 2^P, and the accumulators are multiplied by 2^-P at the end.  Elements more than 2^16
// below their tensor's maximum lose relative (not absolute) accuracy: they are fp16 subnormals, which v_cvt_pk_f16_f32
// produces and the MFMA honours (profiles/r03_f16_split_facts.txt).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int dudf_exp_above(unsigned bits) {     // e with |v| < 2^e, clamped to a range that keeps every scale finite
    const int e = (int)((bits >> 23) & 255u) - 126;
    return e < -40 ? -40 : (e > 60 ? 60 : e);
}
__device__ __forceinline__ float dudf_pow2(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }

// P24 = 2: the fp32 row layout read the way the sweeps read it — a wave-instruction takes 256-byte pieces of FOUR feature-quad rows
// (lane = (q, li): row 16 pw + 4 t + q, column li) instead of 64-byte pieces of sixteen — into the same [column][feature] image
// and transposed fragment reads as P24 = 1, without the 24-bit decode ("row-major staging").
// P24 = 1: the operands arrive as 24-bit tile-major stash arrays (dudf_internal.h "p24": [layer][feature tile][16-column group]
// [64 lanes][3 dwords]) of FIXED-POINT values relative to a per-layer, per-column power of two 2^E (round 5; round 4 read 24-bit
// floats here): value = (t - 3) 2^E with t = 0x40000000 | the 24 bits.  The lane's column is fixed for a stage (lane = (q, li)), so
// ONE extra dword per lane and stage brings 2^E, and one FMA per value — t * g - 3 g with g = 2^E x the operand's layer scale —
// yields the number the fp16 split starts from.  A 16-column stage of an operand is then 16 contiguous 768-byte blocks, one per feature tile: a wave
// loads a block with ONE dwordx3 instruction (six full lines), lane (q, li) holding features 16 T + 4 q .. + 3 of column li.
// A lane therefore has many features of ONE column — the transpose of what the MFMA wants (one feature, 8 columns) — so the
// LDS image is [column k][feature] (rows of 576 bytes: 256 fp16 + 64, four consecutive rows start 16 banks apart; the
// 8-byte unit index of a row XORed in its low three bits with (k >> 1) & 7: a 16-lane ds_write_b64 service group covers
// the 32 write banks once) and the fragments come out of it through ds_read_b64_tr_b16, gfx950's transposing LDS read: a
// 16-lane group reads 4 rows (columns k) x 16 features and every lane receives its feature's four k — two of them per
// fragment (tools/micro/tr_image.hip checks image, swizzle and fragment maps against a host GEMM).
template <int H, int VAR, int SP = 0, int P24 = 0>
__device__ __forceinline__ void wgrad_hidden_bf16p_body(const WgradArgs& a) {
    using W = WG<H>;
    static_assert(H == 256, "256 x 256 output tiles");
    static_assert(P24 == 0 || (SP != 0 && (VAR & 8) != 0), "24-bit operands: the fp16x3, flag-synchronised build");
    constexpr int NPC = SP ? 2 : 3;                         // pieces per operand
    constexpr int NW_ = W::WO * W::WI;
    constexpr int FQ = H / 4;
    constexpr int HALFB = 32 * 16 + 16;                     // 32 features x 8 columns of a block, +16: the second column half
    constexpr int BLKB = 2 * HALFB;                         //   must not sit 512 B = 0 banks after the first (ds_write_b64)
    constexpr int ROWB = 576;                               // P24: one column's 256 fp16 + 64 bytes
    constexpr int PIECEB = P24 ? 16 * ROWB : (H / 32) * BLKB;   // 8.25 KiB (P24: 9)
    constexpr int OPERB = NPC * PIECEB;                     // 24 KiB (16.5; P24: 18)
    constexpr int BUFB = 2 * OPERB;                         // 48 KiB (33; P24: 36)
    extern __shared__ __attribute__((aligned(16))) char ldsb[];     // [2 buffers][X | Y][h | m | l][feature][16 columns]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave / W::WI, wi = wave % W::WI;
    // Layers wider than the tile (512: 2 x 2 tiles): the four tiles of one (layer, column split) read the same X rows twice
    // and the same Y rows twice.  On a 1-D grid whose blocks are dealt round-robin over the 8 XCDs, blocks with equal id % 8
    // share an XCD and its L2: a group's tiles get ids xcd + 8 * (4 * (group / 8) + tile), so the second reader of every row
    // hits L2 instead of HBM (PMC r03_b: 15.2 GB fetched per launch at 8x512 / 125 000 points against 7.2 GB of operands).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, nsplit = gridDim.y;
    if (a.remap_nsplit > 0) {
        const int id = blockIdx.x, slot = id >> 3, grp = (slot >> 2) * 8 + (id & 7);
        if (grp >= a.nj * a.remap_nsplit) return;               // the grid is padded to whole XCD rounds
        bz = slot & 3; bx = grp % a.nj; by = grp / a.nj; nsplit = a.remap_nsplit;
    }
    const int j = bx + a.j0;
    const int steps16 = a.steps_total * (KT / KB);
    const int s0 = (int)((int64_t)steps16 * by / nsplit);
    const int s1 = (int)((int64_t)steps16 * (by + 1) / nsplit);
    const int tz = a.Hs / H;
    const int o_off = (bz / tz) * H, i_off = (bz % tz) * H;
    const int64_t xrow = (int64_t)(o_off / 4) * a.np * 4, yrow = (int64_t)(i_off / 4) * a.np * 4;
    // (fp16x3 build only: the bf16x6 build sits at 256 registers and four more live scalars make it spill)
    const bool clk_on = SP != 0 && a.clk != nullptr && bx == 0 && by == 0 && bz == 0;
    const unsigned long long clk_t0 = clk_on ? __builtin_amdgcn_s_memtime() : 0ull, clk_r0 = clk_on ? __builtin_amdgcn_s_memrealtime() : 0ull;

    f32x16 acc[W::MT][W::NTL];
#pragma unroll
    for (int m = 0; m < W::MT; ++m)
#pragma unroll
        for (int n = 0; n < W::NTL; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;
    // (P24: a layer of a 24-bit array is 3/4 of stash_layer floats; H == Hs, so xrow == yrow == 0)
    const int64_t lstride = P24 == 1 ? a.stash_layer / 4 * 3 : a.stash_layer;
    const float* X0 = a.Q + (int64_t)(j + 1) * lstride + xrow;
    const float* X1 = a.Z + (int64_t)(j + 1) * lstride + xrow;
    const float* Y0 = a.A + (int64_t)j * lstride + yrow;
    const float* Y1 = a.S + (int64_t)j * lstride + yrow;

    const int npair = a.have_g ? 2 : 1;
    const int nit = npair * (s1 - s0);
    auto pair_of = [&](int it) { return a.have_g ? (it & 1) : 1; };
    auto step_of = [&](int it) { return s0 + (a.have_g ? (it >> 1) : it); };

    // producer role of this lane: operand, feature quad, group of four columns
    const int p_oper = wave / (NW_ / 2);                                  // wave-uniform: waves 0..3 stage X, 4..7 stage Y
    const int p_cg = tid & 3;
    const int p_fq = (VAR & 1) ? (((tid >> 2) & 1) | (((tid >> 4) & 3) << 1) | (((tid >> 3) & 1) << 3) | (((tid >> 6) & 3) << 4))
                               : ((tid >> 2) & (FQ - 1));
    // granule j of this lane = stage column p_cg + 4 j: one load instruction then reads 64 contiguous bytes per feature
    // quad (4 lanes x 16 B) and touches each 128-byte line twice instead of four times (PMC r01: 2x the ideal L2
    // requests; the loads alone held the kernel at 0.8 ms).  The image slot 4 p_cg + j therefore holds column p_cg + 4 j:
    // a permutation of the contraction index, the same for both operands, hence free.
    const int64_t p_goff = ((int64_t)p_fq * a.np + p_cg) * 4;             // floats, + col0 * 4 per stage
    // image of one piece: [32-feature block][column half][feature in block][8 columns] = the fragment order of the MFMA
    const int p_loff = p_oper * OPERB + (p_fq >> 3) * BLKB + (p_cg >> 1) * HALFB + ((p_fq & 7) * 4) * 16 + (p_cg & 1) * 8;   // + f * 16 + piece * PIECEB
    typedef unsigned u3_t __attribute__((ext_vector_type(3)));
    typedef typename std::conditional<P24 == 1, u3_t, f32x4>::type raw_t;      // P24 = 1: a granule is 3 dwords (four 24-bit values)
    struct RawSet { raw_t g0, g1, g2, g3; float sc; };                    // granule j: 4 features of column 4*cg + j (P24: of tile 4 pw + j, column li); sc (P24 = 1): the column's 2^E
    RawSet R0, R1, R2;                                                    // stage s travels in set s % 3, three stages ahead
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};                                    // bias gradient partial sums (X operand, zbar pair)
    f32x4 bacc1 = bacc, bacc2 = bacc, bacc3 = bacc;                       // P24: one quad of features per tile
    const float* P0 = p_oper ? Y0 : X0;                                   // this wave's operand for the (q, A) / (zbar, s) pair
    const float* P1 = p_oper ? Y1 : X1;
    // P24 = 1: the column scales of those operands (layer j + 1 for q / zbar, layer j for A / s)
    const float* F0 = P24 == 1 ? (p_oper ? a.fxA + (int64_t)j * a.np : a.fxQ + (int64_t)(j + 1) * a.np) : nullptr;
    const float* F1 = P24 == 1 ? (p_oper ? a.fxS + (int64_t)j * a.np : a.fxZ + (int64_t)(j + 1) * a.np) : nullptr;
    // fp16x3: this wave's operand scale for each pair, and the common product scale 2^P of the layer
    float sc0 = 1.f, sc1 = 1.f, inv_p = 1.f;
    if constexpr (SP != 0) {
        const int eq = dudf_exp_above(a.amax[0 * a.L + j + 1]), eA = dudf_exp_above(a.amax[1 * a.L + j]);
        const int ez = dudf_exp_above(a.amax[2 * a.L + j + 1]);
        const int esh = dudf_exp_above(a.amax[3 * a.L + j]);             // h | hdot^k of the Hessian quads; plain columns: |h| <= 1 (+ an ulp)
        const int es = esh > 1 ? esh : 1;
        const int P1l = 30 - eq - eA, P2l = 30 - ez - es;
        const int P = (a.have_g && P1l < P2l) ? P1l : P2l;
        // the pair with the larger admissible product gives the surplus back through its X operand
        sc0 = p_oper ? dudf_pow2(15 - eA) : dudf_pow2(15 - eq - (P1l - P));
        sc1 = p_oper ? dudf_pow2(15 - es) : dudf_pow2(15 - ez - (P2l - P));
        if (!a.have_g) sc0 = sc1;
        inv_p = dudf_pow2(-P);
    }
    // Inline asm + hand-counted vmcnt (as in the sweeps): with compiler-visible loads hipcc drains ALL stages in flight
    // (vmcnt(0)) at the loop head.  These twelve loads are the only vector-memory operations of the loop.
    // scalar base (operand, pair, stage: wave-uniform) + this lane's fixed 32-bit byte offset: no per-stage 64-bit address
    // arithmetic in vector registers (the launcher keeps FQ * np * 16 below 2^32)
    const unsigned p_voff = (unsigned)(p_goff * 4);
    const float bm_plain = (p_oper == 0 && i_off == 0) ? 1.f : 0.f;     // this lane's granules count towards the bias gradient ...
    const float bm_quad = ((P24 ? (lane & 3) : p_cg) == 0) ? bm_plain : 0.f;   // ... on a Hessian-quad stage (the value channel: column % 4 == 0)
    // P24 producer role: wave pw (0..3 of its operand) stages feature tiles 4 pw .. 4 pw + 3; lane = (q, li) as in the sweeps
    const int pw = wave & (NW_ / 2 - 1), p_q = lane >> 4, p_li = lane & 15;
    const int64_t ngrp = a.np >> 4;                                       // 16-column groups per row of tiles
    // P24 = 1: tile 4 pw (+ t) of the tile-major array; P24 = 2: row 16 pw + q (+ 4 t) of the fp32 row layout, column li
    const unsigned t_voff0 = P24 == 2 ? (unsigned)((((int64_t)(16 * pw + p_q)) * a.np + p_li) * 16)
                                      : (unsigned)(lane * 12 + (int64_t)(4 * pw) * ngrp * 768);   // (the launcher keeps a layer below 2^32 bytes)
    const unsigned t_vstep = P24 == 2 ? (unsigned)(4 * a.np * 16) : (unsigned)(ngrp * 768);
    const unsigned p_li4 = (unsigned)p_li * 4u;                            // P24 = 1: this lane's column scale inside a stage's 16
    // image write offsets of this lane (row li, unit 16 pw + 4 t + q with its low three bits swizzled): tiles with even / odd t
    const int p_sw = (p_li >> 1) & 7;
    const int t_w0 = p_oper * OPERB + p_li * ROWB + 128 * pw + 8 * (p_q ^ p_sw), t_w1 = p_oper * OPERB + p_li * ROWB + 128 * pw + 8 * ((4 + p_q) ^ p_sw);
    auto load_raw = [&](int it, RawSet& r) {
        if constexpr (P24 == 2) {
            const uint64_t g0 = (uint64_t)(size_t)(reinterpret_cast<const char*>(pair_of(it) ? P1 : P0) + (int64_t)step_of(it) * (KB * 16));
            const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)g0), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
            const uint64_t sbase = ((uint64_t)hi32 << 32) | lo32;
            asm volatile("global_load_dwordx4 %0, %4, %8" DUDF_WG_NT "\n\tglobal_load_dwordx4 %1, %5, %8" DUDF_WG_NT "\n\t"
                         "global_load_dwordx4 %2, %6, %8" DUDF_WG_NT "\n\tglobal_load_dwordx4 %3, %7, %8" DUDF_WG_NT
                         : "=&v"(r.g0), "=&v"(r.g1), "=&v"(r.g2), "=&v"(r.g3)
                         : "v"(t_voff0), "v"(t_voff0 + t_vstep), "v"(t_voff0 + 2 * t_vstep), "v"(t_voff0 + 3 * t_vstep), "s"(sbase) : "memory");
        } else if constexpr (P24 != 0) {
            const uint64_t g0 = (uint64_t)(size_t)(reinterpret_cast<const char*>(pair_of(it) ? P1 : P0) + (int64_t)step_of(it) * 768);
            const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)g0), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
            const uint64_t sbase = ((uint64_t)hi32 << 32) | lo32;
            const uint64_t f0 = (uint64_t)(size_t)((pair_of(it) ? F1 : F0) + (int64_t)step_of(it) * KB);
            const unsigned flo = __builtin_amdgcn_readfirstlane((unsigned)f0), fhi = __builtin_amdgcn_readfirstlane((unsigned)(f0 >> 32));
            const uint64_t fbase = ((uint64_t)fhi << 32) | flo;
            asm volatile("global_load_dwordx3 %0, %5, %9" DUDF_WG_NT "\n\tglobal_load_dwordx3 %1, %6, %9" DUDF_WG_NT "\n\t"
                         "global_load_dwordx3 %2, %7, %9" DUDF_WG_NT "\n\tglobal_load_dwordx3 %3, %8, %9" DUDF_WG_NT "\n\t"
                         "global_load_dword %4, %10, %11"
                         : "=&v"(r.g0), "=&v"(r.g1), "=&v"(r.g2), "=&v"(r.g3), "=&v"(r.sc)
                         : "v"(t_voff0), "v"(t_voff0 + t_vstep), "v"(t_voff0 + 2 * t_vstep), "v"(t_voff0 + 3 * t_vstep), "s"(sbase),
                           "v"(p_li4), "s"(fbase) : "memory");
        } else {
        const uint64_t g0 = (uint64_t)(size_t)((pair_of(it) ? P1 : P0) + (int64_t)step_of(it) * KB * 4);
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)g0), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
        const uint64_t sbase = ((uint64_t)hi32 << 32) | lo32;
        asm volatile("global_load_dwordx4 %0, %4, %5" DUDF_WG_NT "\n\tglobal_load_dwordx4 %1, %4, %5 offset:64" DUDF_WG_NT "\n\t"
                     "global_load_dwordx4 %2, %4, %5 offset:128" DUDF_WG_NT "\n\tglobal_load_dwordx4 %3, %4, %5 offset:192" DUDF_WG_NT
                     : "=&v"(r.g0), "=&v"(r.g1), "=&v"(r.g2), "=&v"(r.g3) : "v"(p_voff), "s"(sbase) : "memory");
        }
    };
    // outside the steady-state loop (prologue, last stages: conditional loads) the loads are ordinary ones: a conditional
    // asm load makes hipcc merge "loaded" and "not loaded" values with register copies — of registers still in flight
    auto load_raw_plain = [&](int it, RawSet& r) {
        if constexpr (P24 == 2) {
            const char* src = reinterpret_cast<const char*>(pair_of(it) ? P1 : P0) + (int64_t)step_of(it) * (KB * 16) + t_voff0;
            r.g0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
            r.g1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + t_vstep));
            r.g2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 2 * (size_t)t_vstep));
            r.g3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 3 * (size_t)t_vstep));
        } else if constexpr (P24 != 0) {
            const char* src = reinterpret_cast<const char*>(pair_of(it) ? P1 : P0) + (int64_t)step_of(it) * 768 + t_voff0;
            r.g0 = __builtin_nontemporal_load(reinterpret_cast<const u3_t*>(src));
            r.g1 = __builtin_nontemporal_load(reinterpret_cast<const u3_t*>(src + t_vstep));
            r.g2 = __builtin_nontemporal_load(reinterpret_cast<const u3_t*>(src + 2 * (size_t)t_vstep));
            r.g3 = __builtin_nontemporal_load(reinterpret_cast<const u3_t*>(src + 3 * (size_t)t_vstep));
            r.sc = (pair_of(it) ? F1 : F0)[(int64_t)step_of(it) * KB + p_li];
        } else {
        const float* src = (pair_of(it) ? P1 : P0) + p_goff + (int64_t)step_of(it) * KB * 4;
        r.g0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
        r.g1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + 4);
        r.g2 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + 8);
        r.g3 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src) + 12);
        }
    };
    constexpr int kSetLoads = P24 == 1 ? 5 : 4;          // loads per register set (P24 = 1: + the column scale)
    auto wait_raw = [&](RawSet& r, auto younger) {       // this set has landed; `younger` loads issued after it stay in flight
        if constexpr (P24 == 1) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r.g0), "+v"(r.g1), "+v"(r.g2), "+v"(r.g3), "+v"(r.sc) : "n"(decltype(younger)::value));
        else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.g0), "+v"(r.g1), "+v"(r.g2), "+v"(r.g3) : "n"(decltype(younger)::value));
    };
    // VAR bit 3: THREE image buffers and per-buffer counters in LDS instead of the workgroup barrier of every stage: a wave
    // starts stage `it` when all eight waves have written their part of image `it` (counter W) and have finished reading
    // the buffer its own slices are about to overwrite (counter R, image it - 2): waves may drift up to a stage apart, so the
    // SIMD partner that the issue arbitration serves first (the older wave) no longer idles a third of every stage at the
    // barrier while the younger one finishes.
    constexpr bool CS = (VAR & 8) != 0;
    // VAR bit 4 (with bit 3): FOUR image buffers.  With three, a wave must ask — a second poll per stage — whether every wave has
    // finished READING image it - 1 before it splits image it + 2 into that buffer.  With four the buffer of image it + 2 is the one
    // image it - 2 lived in, and the poll at the top of stage `it` (every wave has WRITTEN image it, which it does behind its reads
    // of image it - 2: LDS executes a wave's operations in order) already says so: one poll and one flag less per stage.
    constexpr int NB = (CS && (VAR & 16) != 0) ? 4 : 3;
    // Per-wave progress words in LDS (no atomics, no address registers).  A wave publishes with ds_write_addtid_b32 from lane 0
    // (address M0) and polls the whole block with one ds_read_addtid_b32 (lane i reads word i).  Inline asm: the poll is a
    // loop, and a compiler-visible loop inside the hand-counted stage loop makes hipcc spill registers that still have loads
    // in flight.
    const unsigned flag0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(ldsb + NB * BUFB);
    auto publish = [&](unsigned lds_addr, unsigned value) {
        unsigned keep, vt; uint64_t ex;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 %2, exec\n\ts_mov_b64 exec, 1\n\t"
                     "v_mov_b32 %1, %4\n\tds_write_addtid_b32 %1\n\ts_mov_b64 exec, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(vt), "=&s"(ex) : "s"(lds_addr), "s"(value) : "memory");
    };
    // one block of 24 words: lanes 0-7 = images written by waves 0..7, 8-15 = stages whose fragment reads are finished,
    // 16-23 = stages whose MFMAs are nearly all issued.  All three are plain stage counters.
    auto poll = [&](unsigned need_w, unsigned need_r, unsigned need_h03, unsigned need_h47) {
        unsigned keep, vt, t0, t1;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n"
                     ".Ldudf_poll%=:\n\t"
                     "ds_read_addtid_b32 %1\n\ts_waitcnt lgkmcnt(0)\n\t"
                     "v_cmp_le_u32 vcc, %5, %1\n\ts_and_b32 %2, vcc_lo, 0xff\n\t"
                     "v_cmp_le_u32 vcc, %6, %1\n\ts_and_b32 %3, vcc_lo, 0xff00\n\ts_or_b32 %2, %2, %3\n\t"
                     "v_cmp_le_u32 vcc, %7, %1\n\ts_and_b32 %3, vcc_lo, 0xf0000\n\ts_or_b32 %2, %2, %3\n\t"
                     "v_cmp_le_u32 vcc, %8, %1\n\ts_and_b32 %3, vcc_lo, 0xf00000\n\ts_or_b32 %2, %2, %3\n\t"
                     "s_cmp_eq_u32 %2, 0xffffff\n\t"
                     "s_cbranch_scc1 .Ldudf_done%=\n\ts_sleep 1\n\ts_branch .Ldudf_poll%=\n"
                     ".Ldudf_done%=:\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep), "=&v"(vt), "=&s"(t0), "=&s"(t1)
                     : "s"(flag0), "s"(need_w), "s"(need_r), "s"(need_h03), "s"(need_h47) : "memory", "vcc", "scc");
    };
    auto split_slice = [&](int it, const RawSet& r, int f, int bsel) {    // feature f of the lane's quad -> piece image buffer bsel
      if constexpr (P24 == 0) {
        char* dst = ldsb + bsel * BUFB + p_loff + 16 * f;
        if (f == 0) {
            // Hessian quads: only columns % 4 == 0 (the value channel) carry the bias — all four granules of the lanes
            // with p_cg == 0, none of the others
            // (branch-free: scalar selects times per-lane constants — a branch here would cut the hand-counted loop body
            // into several basic blocks)
            const float hs = ((int64_t)step_of(it) * KB < a.ncol_h) ? 1.f : 0.f;
            const float pf = pair_of(it) == 1 ? 1.f : 0.f;
            const float bm = pf * (bm_plain + hs * (bm_quad - bm_plain));
            bacc += bm * (r.g0 + r.g1 + r.g2 + r.g3);
        }
        const f32x2 v0 = {r.g0[f], r.g1[f]}, v1 = {r.g2[f], r.g3[f]};     // four columns of one feature -> 3 x 8 bytes
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (SP != 0) {
            // t = v 2^k -> hi = fp16(t), lo = fp16(t - hi): the residual as ONE v_fma_mix_f32 (v * 2^k - hi, exact; reads the
            // fp16 half in place and issues beside the SIMD partner's MFMAs like v_fma_f32 — tools/micro/coissue.hip)
            const float scl = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pair_of(it) ? sc1 : sc0)));
            const f16x2 h0 = __builtin_convertvector(v0 * scl, f16x2), h1 = __builtin_convertvector(v1 * scl, f16x2);
            f32x2 r0, r1;
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0.x) : "v"(v0.x), "s"(scl), "v"(h0));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r0.y) : "v"(v0.y), "s"(scl), "v"(h0));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r1.x) : "v"(v1.x), "s"(scl), "v"(h1));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1.y) : "v"(v1.y), "s"(scl), "v"(h1));
            const f16x2 l0 = __builtin_convertvector(r0, f16x2), l1 = __builtin_convertvector(r1, f16x2);
            *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
            *reinterpret_cast<u32x2*>(dst + PIECEB) = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
            return;
        }
        const unsigned h0 = cvt_pk(v0), h1 = cvt_pk(v1);
        const f32x2 r0 = v0 - unpack(h0), r1 = v1 - unpack(h1);
        const unsigned m0 = cvt_pk(r0), m1 = cvt_pk(r1);
        const f32x2 q0 = r0 - unpack(m0), q1 = r1 - unpack(m1);
        const unsigned l0 = cvt_pk(q0), l1 = cvt_pk(q1);
        *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(dst + PIECEB) = u32x2{m0, m1};
        *reinterpret_cast<u32x2*>(dst + 2 * PIECEB) = u32x2{l0, l1};
      }
    };
    // P24: tile t of this lane's four (features 16 (4 pw + t) + 4 q .. + 3 of column li): unpack, bias sums, fp16 hi / lo, two
    // 8-byte writes into row li of the [column][feature] image
    // (P24 = 1: `scl` = 2^E of this lane's column x the operand's layer scale, `m3` = -3 scl: v = (t - 3) scl is the value in the GEMM's
    //  scale, the bias sums run in that scale too and are divided by the layer scale at the end)
    auto split_tile = [&](int it, const raw_t& g, f32x4& bsum, int t, int bsel, const float bmask, const float scl, const float m3) {
        if constexpr (P24 != 0) {
            f32x4 v;
            f16x2 h0, h1;
            if constexpr (P24 == 1) {
                const unsigned d0 = g[0], d1 = g[1], d2 = g[2];
                const unsigned m = 0x00ffffffu, two = 0x40000000u;
                const unsigned t0 = (d0 & m) | two, t1 = (d1 & m) | two, t2 = (d2 & m) | two;          // v_and_or_b32
                const unsigned y = __builtin_amdgcn_perm(d1, d0, 0x0c0c0703u);                        // [d0.b3, d1.b3, 0, 0]
                const unsigned t3 = __builtin_amdgcn_perm(d2, y, 0x0c070100u) | two;                   // [.., .., d2.b3, 0] | 0x40 on top
                v = f32x4{__builtin_fmaf(__uint_as_float(t0), scl, m3), __builtin_fmaf(__uint_as_float(t1), scl, m3),
                          __builtin_fmaf(__uint_as_float(t2), scl, m3), __builtin_fmaf(__uint_as_float(t3), scl, m3)};
                h0 = __builtin_convertvector(f32x2{v[0], v[1]}, f16x2); h1 = __builtin_convertvector(f32x2{v[2], v[3]}, f16x2);
            } else {
                v = g;
            }
            bsum += bmask * v;
            const f32x2 v0 = {v[0], v[1]}, v1 = {v[2], v[3]};
            f32x2 r0, r1;
            if constexpr (P24 == 1) {
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0.x) : "v"(v0.x), "v"(h0));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r0.y) : "v"(v0.y), "v"(h0));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r1.x) : "v"(v1.x), "v"(h1));
                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1.y) : "v"(v1.y), "v"(h1));
            } else {
            // hi = fp16(v 2^k) in ONE instruction per value (v_fma_mixlo_f16 / v_fma_mixhi_f16: fp32 sources, fp16 result into one
            // half of the destination; the product with a power of two is exact, so this is the rounding of multiply + convert)
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h0) : "v"(v0.x), "s"(scl));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h0) : "v"(v0.y), "s"(scl));
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(v1.x), "s"(scl));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(v1.y), "s"(scl));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0.x) : "v"(v0.x), "s"(scl), "v"(h0));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r0.y) : "v"(v0.y), "s"(scl), "v"(h0));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r1.x) : "v"(v1.x), "s"(scl), "v"(h1));
            asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1.y) : "v"(v1.y), "s"(scl), "v"(h1));
            }
            const f16x2 l0 = __builtin_convertvector(r0, f16x2), l1 = __builtin_convertvector(r1, f16x2);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            char* dst = ldsb + bsel * BUFB + ((t & 1) ? t_w1 : t_w0) + 64 * (t >> 1);
            *reinterpret_cast<u32x2*>(dst) = u32x2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
            *reinterpret_cast<u32x2*>(dst + PIECEB) = u32x2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
        }
    };
    auto split_store = [&](int it, const RawSet& r, int bsel) {           // raw (stage it) -> piece image buffer bsel
        if constexpr (P24 != 0) {
            // once per stage: which of this lane's values count towards the bias gradient (zbar pair; on a Hessian-quad stage
            // only the value channel, column % 4 == 0), and this operand's power of two
            const float hs = ((int64_t)step_of(it) * KB < a.ncol_h) ? 1.f : 0.f;
            const float pf = pair_of(it) == 1 ? 1.f : 0.f;
            const float bmask = pf * (bm_plain + hs * (bm_quad - bm_plain));
            const float scl0 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(pair_of(it) ? sc1 : sc0)));
            const float scl = P24 == 1 ? r.sc * scl0 : scl0;              // P24 = 1: per lane — the column's 2^E times the layer scale (both powers of two)
            const float m3 = -3.0f * scl;
            split_tile(it, r.g0, bacc, 0, bsel, bmask, scl, m3); split_tile(it, r.g1, bacc1, 1, bsel, bmask, scl, m3);
            split_tile(it, r.g2, bacc2, 2, bsel, bmask, scl, m3); split_tile(it, r.g3, bacc3, 3, bsel, bmask, scl, m3);
        } else {
#pragma unroll
            for (int f = 0; f < 4; ++f) split_slice(it, r, f, bsel);
        }
    };
    // consumer role: fragment (32-feature block b, piece p) of an operand = 1 KiB, lane-linear
    const int c_lane = (lane >> 5) * HALFB + (lane & 31) * 16;
    // P24: this lane's two transposed reads of a fragment — rows k = 8 (lane >> 5) + ((lane & 15) >> 2) and k + 4, unit
    // 8 b + 4 ((lane >> 4) & 1) + (lane & 3) of block b, low three bits swizzled with the row's (k >> 1) & 7
    const int c_k = 8 * (lane >> 5) + ((lane & 15) >> 2), c_u = 4 * ((lane >> 4) & 1) + (lane & 3);
    const int c_t0 = c_k * ROWB + 8 * (c_u ^ ((c_k >> 1) & 7)), c_t1 = (c_k + 4) * ROWB + 8 * (c_u ^ (((c_k + 4) >> 1) & 7));
    auto frag_tr = [&](const char* base) -> u32x4 {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + c_t0)));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + c_t1)));
        return u32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto fragA = [&](const char* buf, int m, int pc) -> u32x4 {
        if constexpr ((DUDF_WGRAD_DBG & 16) != 0) { u32x4 z; asm volatile("" : "=v"(z)); return z; }     // timing only: no LDS read
        if constexpr (P24 != 0) return frag_tr(buf + pc * PIECEB + (wo * W::MT + m) * 64);
        else return *reinterpret_cast<const u32x4*>(buf + pc * PIECEB + (wo * W::MT + m) * BLKB + c_lane);
    };
    auto fragB = [&](const char* buf, int n, int pc) -> u32x4 {
        if constexpr ((DUDF_WGRAD_DBG & 16) != 0) { u32x4 z; asm volatile("" : "=v"(z)); return z; }
        if constexpr (P24 != 0) return frag_tr(buf + OPERB + pc * PIECEB + (wi * W::NTL + n) * 64);
        else return *reinterpret_cast<const u32x4*>(buf + OPERB + pc * PIECEB + (wi * W::NTL + n) * BLKB + c_lane);
    };

    if constexpr (CS) {
        if (tid < 128) reinterpret_cast<unsigned*>(ldsb + NB * BUFB)[tid] = 0;
        __syncthreads();
    }
    if constexpr (CS) {                                  // images 0 and 1 written, images 2, 3, 4 in flight (image k: set k % 3)
        if (nit > 0) {
            load_raw_plain(0, R0);
            if (nit > 1) load_raw_plain(1, R1);
            if (nit > 2) load_raw_plain(2, R2);
            split_store(0, R0, 0);
            publish(flag0 + 4u * (unsigned)wave, 1u);                          // one image written
            if (nit > 3) load_raw_plain(3, R0);
            if (nit > 1) {
                split_store(1, R1, 1);
                publish(flag0 + 4u * (unsigned)wave, 2u);                      // two
            }
            if (nit > 4) load_raw_plain(4, R1);
        }
    } else if (nit > 0) {
        load_raw_plain(0, R0);
        split_store(0, R0, 0);
        if (nit > 1) load_raw_plain(1, R1);
        if (nit > 2) load_raw_plain(2, R2);
        if (nit > 3) load_raw_plain(3, R0);
    }
    __syncthreads();
    // one stage: the next stage's registers -> pieces (its set is then refilled with the stage three further on, so three
    // stages = 96 KiB per CU stay in flight: with one, the kernel measured latency-bound at 2 TB/s), then 48 MFMAs
#if DUDF_WGRAD_DBG & 32
    unsigned long long wst[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto stage = [&](int it, RawSet& r, auto hot, auto bidx) {
        constexpr int BI = decltype(bidx)::value;             // it % 3 (the loops advance by three stages)
        const int bcur = !CS ? (it & 1) : (NB == 4 ? (it & 3) : BI), bnext = CS ? (BI + 1) % 3 : ((it + 1) & 1);
        const char* buf = ldsb + bcur * BUFB;
        constexpr bool HOT = decltype(hot)::value;
        DUDF_WSTAMP(0);
        if constexpr (CS) {
            // every wave has written its part of image `it` (generation g + 1 of W[BI]) and has finished reading image
            // it - 1, whose buffer this wave's split of image it + 2 is going to overwrite (R[(BI + 2) % 3])
            // (the second condition is checked in front of the split, behind this stage's MFMAs: the SIMD partner runs half
            //  a stage behind, and waiting for ITS reads here would stall this wave at the top of every stage)
            // + the SIMD partners alternate on the matrix pipe: waves 4-7 start the MFMAs of stage `it` when waves 0-3 have
            //   issued most of theirs, waves 0-3 start stage `it` when waves 4-7 have issued most of stage it - 1 (the flag is
            //   raised one MFMA group before the end, so that the hand-over latency is covered)
            const unsigned uit = (unsigned)it;
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(wave >= NW_ / 2));
            poll(uit + 1u, 0u, hi ? uit + 1u : 0u, hi ? 0u : uit);
        }
        constexpr int DBG = HOT ? DUDF_WGRAD_DBG : 0;
        constexpr bool IL = (VAR & 2) != 0;
        static_assert(!IL || W::NTL == 4, "one split slice per MFMA group");
        static_assert(!(CS && IL), "the flag-synchronised variant multiplies first and splits two images ahead");
        const bool more = HOT || it + 1 < nit;
        if constexpr (!IL && !CS) {
            if constexpr (HOT) {                  // steady state: the two younger sets (8 loads) stay in flight
                wait_raw(r, std::integral_constant<int, 2 * kSetLoads>{});
                split_store(it + 1, r, bnext);
                load_raw(it + 4, r);
            } else if (more) {
                split_store(it + 1, r, bnext);
                if (it + 4 < nit) load_raw_plain(it + 4, r);
            }
        }
        u32x4 af[W::MT][NPC], bn[NPC];
#pragma unroll
        for (int m = 0; m < W::MT; ++m)
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) af[m][pc] = fragA(buf, m, pc);
        DUDF_WSTAMP(1);
        if constexpr (!IL) {
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bn[pc] = fragB(buf, 0, pc);
        }
#pragma unroll
        for (int n = 0; n < W::NTL; ++n) {
            if constexpr (IL) {
                // this group's B fragments, then slice n of the next stage's split (its VALU covers the read latency: no
                // second fragment set in registers), then this group's MFMAs
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) bn[pc] = fragB(buf, n, pc);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (HOT && !(DBG & 1)) { if (n == 0) wait_raw(r, std::integral_constant<int, 2 * kSetLoads>{}); }
                if constexpr (!(DBG & 8)) { if (more) split_slice(it + 1, r, n, bnext); }
                __builtin_amdgcn_sched_barrier(0);
                DUDF_WSTAMP(2 + 2 * n);
            }
            u32x4 bc[NPC];
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bc[pc] = bn[pc];
            if constexpr (!IL) {
                if (n + 1 < W::NTL) {
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) bn[pc] = fragB(buf, n + 1, pc);
                    __builtin_amdgcn_sched_barrier(0x76);        // LDS reads and MFMAs keep their order: fragments one block ahead
                }
            }
#pragma unroll
            for (int m = 0; m < W::MT; ++m) {
                if constexpr ((DBG & 4) != 0) { asm volatile("" :: "v"(af[m][0]), "v"(af[m][1]), "v"(bc[0]), "v"(bc[1])); continue; }
                f32x16 c = acc[m][n];
                if constexpr (SP != 0) {                          // smallest terms first: lo*hi, hi*lo, hi*hi
                    auto H8 = [](u32x4 v) { return __builtin_bit_cast(f16x8, v); };
                    c = mfma_f16(H8(af[m][1]), H8(bc[0]), c);
                    c = mfma_f16(H8(af[m][0]), H8(bc[1]), c);
                    c = mfma_f16(H8(af[m][0]), H8(bc[0]), c);
                } else {
                    auto B8 = [](u32x4 v) { return __builtin_bit_cast(bf16x8, v); };
                    const bf16x8 bh = B8(bc[0]), bmid = B8(bc[1]), bl = B8(bc[NPC - 1]);
                    c = mfma_bf16(B8(af[m][1]), bmid, c);             // smallest terms first
                    c = mfma_bf16(B8(af[m][NPC - 1]), bh, c);
                    c = mfma_bf16(B8(af[m][0]), bl, c);
                    c = mfma_bf16(B8(af[m][1]), bh, c);
                    c = mfma_bf16(B8(af[m][0]), bmid, c);
                    c = mfma_bf16(B8(af[m][0]), bh, c);
                }
                acc[m][n] = c;
            }
            if constexpr (IL) __builtin_amdgcn_sched_barrier(0);
            if constexpr (CS) {
                if (n == W::NTL - DUDF_WG_HREL) {
                    __builtin_amdgcn_sched_barrier(0);
                    publish(flag0 + 64u + 4u * (unsigned)wave, (unsigned)it + 1u);                 // nearly done with the matrix pipe
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            DUDF_WSTAMP(3 + 2 * n);
        }
        if constexpr (IL) {
            if constexpr (HOT) { if constexpr (!(DBG & 1)) load_raw(it + 4, r); }
            else if (it + 4 < nit) load_raw_plain(it + 4, r);
        }
        if constexpr (CS) {
            // behind this stage's last fragment reads (LDS executes a wave's operations in order): image `it` read; then
            // the split of image it + 2 into the buffer image it - 1 lived in, and its flag
            __builtin_amdgcn_sched_barrier(0);
            const int BW = NB == 4 ? ((it + 2) & 3) : (BI + 2) % 3;
            if constexpr (NB == 3) {
                publish(flag0 + 32u + 4u * (unsigned)wave, (unsigned)it + 1u);                   // fragment reads of stage `it` done
                DUDF_WSTAMP(2);
                if (HOT || it + 2 < nit) poll(0u, (unsigned)it, 0u, 0u);                         // image it - 1 read by everybody: its buffer is free
            }
            DUDF_WSTAMP(4);
            if constexpr (HOT) {
                if constexpr (!(DBG & 1)) wait_raw(r, std::integral_constant<int, 2 * kSetLoads>{});         // (DBG: timing experiments, wrong results)
                DUDF_WSTAMP(6);
                if constexpr (!(DBG & 8)) split_store(it + 2, r, BW);
                DUDF_WSTAMP(8);
                if constexpr (!(DBG & 1)) load_raw(it + 5, r);
                publish(flag0 + 4u * (unsigned)wave, (unsigned)it + 3u);                         // images 0 .. it + 2 written
            } else if (it + 2 < nit) {
                split_store(it + 2, r, BW);
                if (it + 5 < nit) load_raw_plain(it + 5, r);
                publish(flag0 + 4u * (unsigned)wave, (unsigned)it + 3u);
            }
        }
        DUDF_WSTAMP(10);
        if constexpr (!(DBG & 2) && !CS) __syncthreads();
        DUDF_WSTAMP(11);
    };
    // VAR bit 2: static priority for waves 0-3 (their SIMD partners are waves 4-7).  Both waves of a SIMD leave every
    // barrier in lockstep, so their split slices (VALU, matrix core idle) and their MFMA groups (matrix core contended)
    // coincide and interleaving alone buys nothing; with one of them served first at every contended issue slot the other
    // falls behind by one slice and from then on runs its slices beside the partner's MFMAs.  No per-stage flips.
    if constexpr ((VAR & 4) != 0) { if (wave < NW_ / 2) __builtin_amdgcn_s_setprio(1); }
    int it = 0;                                                  // stage it+1 lives in set (it+1) % 3
    const int nhot = CS ? (nit >= 8 ? ((nit - 5) / 3) * 3 : 0) : (nit >= 7 ? ((nit - 4) / 3) * 3 : 0);
    // hipcc does not know the sets are in flight: enter (and leave) the hand-counted loop with everything landed, so that
    // the register copies it places on the loop's edges are harmless; inside, the sets stay put (tests/isa_contract.py
    // replays the loop body twice and fails on any instruction that touches a register with a load in flight)
    auto drain = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(R0.g0), "+v"(R0.g1), "+v"(R0.g2), "+v"(R0.g3), "+v"(R1.g0), "+v"(R1.g1),
                     "+v"(R1.g2), "+v"(R1.g3), "+v"(R2.g0), "+v"(R2.g1), "+v"(R2.g2), "+v"(R2.g3));
        if constexpr (P24 == 1) asm volatile("" : "+v"(R0.sc), "+v"(R1.sc), "+v"(R2.sc));
    };
    drain();
    using B0 = std::integral_constant<int, 0>; using B1 = std::integral_constant<int, 1>; using B2 = std::integral_constant<int, 2>;
    // register set of a stage: the image it splits — it + 1 (set (it + 1) % 3), or it + 2 with the flags (set (it + 2) % 3)
    RawSet& Sa = CS ? R2 : R1; RawSet& Sb = CS ? R0 : R2; RawSet& Sc = CS ? R1 : R0;
    for (; it < nhot; it += 3) {
        stage(it, Sa, std::true_type{}, B0{});
        stage(it + 1, Sb, std::true_type{}, B1{});
        stage(it + 2, Sc, std::true_type{}, B2{});
    }
    drain();
    for (; it < nit; it += 3) {
        stage(it, Sa, std::false_type{}, B0{});
        if (it + 1 < nit) stage(it + 1, Sb, std::false_type{}, B1{});
        if (it + 2 < nit) stage(it + 2, Sc, std::false_type{}, B2{});
    }
#if DUDF_WGRAD_DBG & 32
    if (bx == 3 && by == 10 && lane == 0)
        for (int i = 0; i < 12; ++i) g_wstamp[wave][i] = wst[i];
#endif

    float* dW = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)o_off * a.Hs + i_off;
    float* dB = a.dtheta + a.off_hid + (int64_t)j * a.hid_stride + (int64_t)a.Hs * a.Hs + o_off;
    if (nit > 0) {
        const int l32 = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int m = 0; m < W::MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int o = (wo * W::MT + m) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
#pragma unroll
                for (int n = 0; n < W::NTL; ++n) {
                    const int i = (wi * W::NTL + n) * 32 + l32;
#if DUDF_WGRAD_DBG & 64
                    asm volatile("" :: "v"(acc[m][n][e]));               // timing experiment: no output atomics
#else
                    atomicAdd(dW + (int64_t)o * a.Hs + i, SP != 0 ? acc[m][n][e] * inv_p : acc[m][n][e]);
#endif
                }
            }
        if (p_oper == 0 && i_off == 0) {                          // bias gradient: sum the four column groups of a quad first
            if constexpr (P24 != 0) {                             // ... P24: the 16 columns (lanes li) of features 16 (4 pw + t) + 4 q + e
                const float bun = P24 == 1 ? 1.0f / sc1 : 1.0f;          // P24 = 1: the sums ran in zbar's layer scale (a power of two)
                auto red = [&](float v, int f) {
                    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
                    if (p_li == 0) atomicAdd(dB + f, v * bun);
                };
                const f32x4 bs[4] = {bacc, bacc1, bacc2, bacc3};
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) red(bs[t][e], 16 * (4 * pw + t) + 4 * p_q + e);
            } else {
            auto red = [&](float v, int f) {
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                if (p_cg == 0) atomicAdd(dB + 4 * p_fq + f, v);
            };
            red(bacc.x, 0); red(bacc.y, 1); red(bacc.z, 2); red(bacc.w, 3);
            }
        }
    }
    if (clk_on && tid == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}
// built without packed fp32 instructions (dudf_internal.h, DUDF_NO_PK): the split slices of one wave then execute beside its
// SIMD partner's MFMAs (-5 % on this kernel).  The body is a forced-inline function: lambdas defined inside a function
// that carries the target attribute do not inherit it and would not be inlined.
template <int H, int VAR>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_bf16p_kernel(WgradArgs a) {
    wgrad_hidden_bf16p_body<H, VAR>(a);
}
template <int H, int VAR>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_f16p_kernel(WgradArgs a) {
    wgrad_hidden_bf16p_body<H, VAR, 1>(a);
}
// ... the fp32 rows staged 4 rows x 256 bytes per wave-instruction, [column][feature] image, transposed fragment reads
template <int H, int VAR>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_f16tr_kernel(WgradArgs a) {
    wgrad_hidden_bf16p_body<H, VAR, 1, 2>(a);
}
// ... reading 24-bit tile-major operands (dudf_internal.h "p24")
template <int H, int VAR>
__global__ __launch_bounds__(64 * WG<H>::WO * WG<H>::WI) DUDF_NO_PK void wgrad_hidden_f16p24_kernel(WgradArgs a) {
    wgrad_hidden_bf16p_body<H, VAR, 1, 1>(a);
}

// ---- first and last layer: thin reductions over columns (bandwidth-bound, VALU) -------------------------
//   dW_1[o][d] | db_1[o] = sum_c  q_1[o][c] * gbar[c][d]  +  zbar_1[o][c] * x4[c][d]      (d = 3 is the bias: x4[c][3])
//   dW_out[f]            = sum_c  A_L[f][c] * x4[c][3]   +  ybar[c] * s_L[f][c]
//   db_out               = sum_c  ybar[c]
#ifndef DUDF_WGSMALL_P24_PTS
#define DUDF_WGSMALL_P24_PTS 4096
#endif
struct WgradSmallArgs {
    const float *Q, *A, *Z, *S;
    const float *x4, *gbar, *ybar;      // x4 [np][4]; gbar [np][4]; ybar [np]
    float* dtheta;
    int64_t np, ncols, stash_layer, off_wo, off_bo;
    int H, L, have_g;
    int pts_per_block;
    float rho;                          // w0 / ww: d(loss)/d(W_1, b_1) = rho d(loss)/d(rho W_1, rho b_1)
    const float *fxS, *fxQ, *fxA, *fxZ; // 24-bit fixed-point arrays: [L][np] column scales (dudf_internal.h)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid.x = column ranges, grid.y = feature-quad groups; block = 256 threads = 4 waves; each wave loops over
// feature quads, lanes run over 64 consecutive columns (16-byte granules -> 1 KiB coalesced per load).
__device__ __forceinline__ void wgrad_small_body(const WgradSmallArgs& a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int FQ = a.H / 4;
    const int64_t p0 = (int64_t)blockIdx.x * a.pts_per_block;
    const int64_t p1 = (p0 + a.pts_per_block < a.ncols) ? p0 + a.pts_per_block : a.ncols;
    const float* Q0 = a.Q;                                          // layer index 0
    const float* Z0 = a.Z;
    const float* AL = a.A + (int64_t)(a.L - 1) * a.stash_layer;
    const float* SL = a.S + (int64_t)(a.L - 1) * a.stash_layer;
    const int fq_per = (FQ + gridDim.y - 1) / gridDim.y;
    const int fq_lo = blockIdx.y * fq_per, fq_hi = (fq_lo + fq_per < FQ) ? fq_lo + fq_per : FQ;
    for (int fq = fq_lo + wave; fq < fq_hi; fq += 4) {
        float w1[4][4], wo[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { wo[c] = 0.f; w1[c][0] = w1[c][1] = w1[c][2] = w1[c][3] = 0.f; }
#pragma unroll 4
        for (int64_t p = p0 + lane; p < p1; p += 64) {
            const int64_t off = ((int64_t)fq * a.np + p) * 4;
            // the stash is a stream (last reader of these rows); x4 / gbar / ybar are re-read per feature quad: cached
            const f32x4 z = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Z0 + off));
            const f32x4 sl = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(SL + off));
            const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x4 + p * 4);
            const float yb = a.ybar[p];
            f32x4 qv = {0, 0, 0, 0}, al = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
            if (a.have_g) {
                qv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(Q0 + off));
                al = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(AL + off));
                gb = *reinterpret_cast<const f32x4*>(a.gbar + p * 4);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int d = 0; d < 4; ++d) w1[c][d] += qv[c] * gb[d] + z[c] * xv[d];
                wo[c] += al[c] * xv[3] + yb * sl[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = 4 * fq + c;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float v = wave_sum(w1[c][d]);
                if (lane == 0) atomicAdd(d < 3 ? a.dtheta + f * 3 + d : a.dtheta + 3 * a.H + f, a.rho * v);
            }
            const float vo = wave_sum(wo[c]);
            if (lane == 0) atomicAdd(a.dtheta + a.off_wo + f, vo);
        }
    }
    if (wave == 0 && blockIdx.y == 0) {                               // db_out = sum ybar
        float s = 0.f;
        for (int64_t p = p0 + lane; p < p1; p += 64) s += a.ybar[p];
        s = wave_sum(s);
        if (lane == 0) atomicAdd(a.dtheta + a.off_bo, s);
    }
}
__global__ __launch_bounds__(256) DUDF_NO_PK void wgrad_small_kernel(WgradSmallArgs a) { wgrad_small_body(a); }

// The same reduction over the 24-bit tile-major fixed-point arrays (dudf_internal.h "p24" bit 0): grid.x = column ranges, grid.y =
// feature tiles; a wave walks 16-column groups, lane = (q, li) as in the sweeps: ONE dwordx3 per array and group (768 contiguous
// bytes per wave) + the column's scale 2^E; value = (t - 3) 2^E (dudf_sweep_common.h fx24_pack).
__device__ __forceinline__ f32x4 small_unpack24(const unsigned* g, const float sc) {
    typedef unsigned u3_t __attribute__((ext_vector_type(3)));
    const u3_t d = __builtin_nontemporal_load(reinterpret_cast<const u3_t*>(g));
    const unsigned d0 = d.x, d1 = d.y, d2 = d.z;
    const unsigned m = 0x00ffffffu, two = 0x40000000u;
    const unsigned y = __builtin_amdgcn_perm(d1, d0, 0x0c0c0703u);
    const unsigned t3 = __builtin_amdgcn_perm(d2, y, 0x0c070100u) | two;
    const float m3 = -3.0f * sc;
    return f32x4{__builtin_fmaf(__uint_as_float((d0 & m) | two), sc, m3), __builtin_fmaf(__uint_as_float((d1 & m) | two), sc, m3),
                 __builtin_fmaf(__uint_as_float((d2 & m) | two), sc, m3), __builtin_fmaf(__uint_as_float(t3), sc, m3)};
}
__global__ __launch_bounds__(256) DUDF_NO_PK void wgrad_small_p24_kernel(WgradSmallArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, li = lane & 15, T = blockIdx.y;
    const int64_t ng = a.np >> 4;
    const int64_t g0 = (int64_t)blockIdx.x * (a.pts_per_block >> 4);
    const int64_t gend = (a.ncols + 15) >> 4;
    const int64_t g1 = g0 + (a.pts_per_block >> 4) < gend ? g0 + (a.pts_per_block >> 4) : gend;
    const int64_t lbytes = a.stash_layer * 3;                            // bytes per layer of a 24-bit array
    auto tile = [&](const float* arr, int layer) { return reinterpret_cast<const char*>(arr) + (int64_t)layer * lbytes + (int64_t)T * ng * 768 + lane * 12; };
    const char* Q0 = tile(a.Q, 0); const char* Z0 = tile(a.Z, 0);
    const char* AL = tile(a.A, a.L - 1); const char* SL = tile(a.S, a.L - 1);
    float w1[4][4], wo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { wo[c] = 0.f; w1[c][0] = w1[c][1] = w1[c][2] = w1[c][3] = 0.f; }
    float sy = 0.f;
    const int nwave = (int)(blockDim.x >> 6);                           // 4; ONE in the deterministic mode (one adder per output)
#pragma unroll 4
    for (int64_t g = g0 + wave; g < g1; g += nwave) {
        const int64_t p = g * 16 + li;
        const f32x4 z = small_unpack24(reinterpret_cast<const unsigned*>(Z0 + g * 768), a.fxZ[p]);
        const f32x4 sl = small_unpack24(reinterpret_cast<const unsigned*>(SL + g * 768), a.fxS[(int64_t)(a.L - 1) * a.np + p]);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x4 + p * 4);
        const float yb = a.ybar[p];
        f32x4 qv = {0, 0, 0, 0}, al = {0, 0, 0, 0}, gb = {0, 0, 0, 0};
        if (a.have_g) {
            qv = small_unpack24(reinterpret_cast<const unsigned*>(Q0 + g * 768), a.fxQ[p]);
            al = small_unpack24(reinterpret_cast<const unsigned*>(AL + g * 768), a.fxA[(int64_t)(a.L - 1) * a.np + p]);
            gb = *reinterpret_cast<const f32x4*>(a.gbar + p * 4);
        }
        sy += yb;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int d = 0; d < 4; ++d) w1[c][d] += qv[c] * gb[d] + z[c] * xv[d];
            wo[c] += al[c] * xv[3] + yb * sl[c];
        }
    }
    auto sum16 = [](float v) { v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); return v; };
    // the block's waves add up in LDS first: one atomic per output and BLOCK instead of one per wave (400 blocks x 4 waves x 80
    // atomics onto 1280 addresses were a fifth of this kernel's time)
    __shared__ float red[4][4][21];                                      // [wave][lane quarter][20 sums + sum of ybar]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const float v = sum16(w1[c][d]);
            if (li == 0) red[wave][q][c * 4 + d] = v;
        }
        const float vo = sum16(wo[c]);
        if (li == 0) red[wave][q][16 + c] = vo;
    }
    {
        const float v = sum16(sy);
        if (li == 0) red[wave][q][20] = v;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 81; t += blockDim.x) {                 // (one wave per block in the deterministic mode)
        if (t < 80) {                                                    // (q, c, d | output row): the waves' partial sums -> one atomic
            const int qq = t / 20, k = t % 20;
            float v = 0.f;
            for (int w = 0; w < nwave; ++w) v += red[w][qq][k];
            if (k < 16) {
                const int c = k >> 2, d = k & 3, f = 16 * T + 4 * qq + c;
                atomicAdd(d < 3 ? a.dtheta + f * 3 + d : a.dtheta + 3 * a.H + f, a.rho * v);
            } else {
                atomicAdd(a.dtheta + a.off_wo + 16 * T + 4 * qq + (k - 16), v);
            }
        } else if (T == 0) {                                             // db_out = sum ybar (every quarter holds the same sum)
            float v = 0.f;
            for (int w = 0; w < nwave; ++w) v += red[w][0][20];
            atomicAdd(a.dtheta + a.off_bo, v);
        }
    }
}

template <int H>
int launch_hidden(const WgradArgs& a, hipStream_t st) {
    using W = WG<H>;
    constexpr int NTHR = 64 * W::WO * W::WI;
    const size_t smem = 4 * (size_t)(H / 4) * KTP * 4 * sizeof(float);   // 2 buffers x (X tile + Y tile)
    const size_t smem_bf = (size_t)NRING * 2 * (H / 4) * KB * 4 * sizeof(float);   // ring of 4 x (X image + Y image)
    const int nl = a.nj;
    if (nl <= 0) return 0;
    const int tz = a.Hs / H, ntz = tz * tz;              // output tiles of a layer wider than the 256 x 256 tile
    // one resident workgroup per CU, a single round.  dudf_set_wgrad_max_workgroups (default 256) caps the grid: with more
    // than one rank the engine sets 240, so that an RCCL kernel queued behind the previous layer group finds free CUs
    // beside this GEMM (its workgroups fill the register file of the CUs they run on)
    const int maxwg = dudf_wgrad_max_workgroups();
    int nsplit = maxwg / (nl * ntz);
    if (nsplit > a.steps_total) nsplit = a.steps_total;
    if (nsplit < 1 || dudf_deterministic()) nsplit = 1;      // deterministic: one workgroup per weight tile, one add per element
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_kernel<H>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    // option wgrad_family = 1 selects the f32-input MFMA kernel (A/B testing); default: the 16-bit cores at fp32 accuracy
    const bool use_f32 = dudf_opt_wgrad_family() == 1;
    dudf_note_products(PROF_WGRAD_HIDDEN, use_f32 ? 1 : 6);         // (the fp16x3 branch below overrides)
    if (a.p24 && (use_f32 || H != 256)) return DUDF_E_UNSUPPORTED;   // 24-bit operands: only the cooperative-split fp16x3 kernel reads them
    if (use_f32) {
        hipLaunchKernelGGL((wgrad_hidden_kernel<H>), dim3(nl, nsplit, ntz), dim3(NTHR), smem, st, a);
    } else {
        static bool attr2 = false;
        if (!attr2) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_bf16_kernel<H>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_bf);
            if (e != hipSuccess) return (int)e;
            attr2 = true;
        }
        // option wgrad_family = 2 keeps the per-wave split kernel for the 256-wide tiles (A/B testing)
        const bool per_wave = dudf_opt_wgrad_family() == 2;
        if constexpr (H == 256) {
            if (a.p24 && per_wave) return DUDF_E_UNSUPPORTED;
            if (!per_wave) {
                static bool attr3 = false;
                const size_t smem_p = 2 * 2 * 3 * (size_t)(H / 32) * 2 * (32 * 16 + 16);   // 2 buffers x (X | Y) x 3 pieces x blocks
                if ((int64_t)(a.Hs / 4) * a.np * 16 >= (1ll << 32)) {       // beyond the 32-bit lane byte offsets of the staging loads:
                    hipLaunchKernelGGL((wgrad_hidden_bf16_kernel<H>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_bf, st, a);   // per-wave split
                    return (int)hipGetLastError();
                }
                // The shipped body is VAR 9: conflict-free producer lanes + progress flags in LDS instead of the stage barrier (three
                // image buffers, MFMAs first, split two images ahead, SIMD partners alternating on the matrix pipe).  Its
                // barrier-synchronised predecessors (VAR 1, 3) and the static-priority variants are no longer instantiated:
                // DESIGN.md Appendix A has their numbers.
                constexpr int var = 9;
                if (a.p24) {                                                      // 24-bit tile-major operands: their own build
                    if (!(dudf_split_fp16() && a.amax && a.L <= 64 && var == 9 && ntz == 1)) return DUDF_E_UNSUPPORTED;
                    dudf_note_products(PROF_WGRAD_HIDDEN, 3);
                    static bool attr5 = false;
                    const size_t smem_t = 3 * (size_t)(2 * 2 * 16 * 576) + 512;    // three buffers x (X | Y) x 2 pieces x 16 rows of 576 B + the flags
                    const size_t smem_t4 = 4 * (size_t)(2 * 2 * 16 * 576) + 512;   // four (VAR bit 4)
                    if (!attr5) {
                        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_f16p24_kernel<H, 9>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_t);
                        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_f16p24_kernel<H, 25>),
                                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_t4);
                        if (e != hipSuccess) return (int)e;
                        attr5 = true;
                    }
                    if (dudf_opt_wgrad_buffers() == 4) hipLaunchKernelGGL((wgrad_hidden_f16p24_kernel<H, 25>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_t4, st, a);
                    else hipLaunchKernelGGL((wgrad_hidden_f16p24_kernel<H, 9>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_t, st, a);
                    return (int)hipGetLastError();
                }
                const bool tr = dudf_opt_wgrad_tr();
                if (tr && dudf_split_fp16() && a.amax && a.L <= 64 && var == 9 && ntz == 1 && !dudf_deterministic()) {
                    dudf_note_products(PROF_WGRAD_HIDDEN, 3);
                    static bool attr6 = false;
                    const size_t smem_t = 3 * (size_t)(2 * 2 * 16 * 576) + 512;
                    if (!attr6) {
                        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_f16tr_kernel<H, 9>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_t);
                        if (e != hipSuccess) return (int)e;
                        attr6 = true;
                    }
                    hipLaunchKernelGGL((wgrad_hidden_f16tr_kernel<H, 9>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_t, st, a);
                    return (int)hipGetLastError();
                }
                if (dudf_split_fp16() && a.amax && a.L <= 64 && var == 9) {       // fp16x3 (DUDF_SPLIT=bf16 keeps bf16x6)
                    dudf_note_products(PROF_WGRAD_HIDDEN, 3);
                    static bool attr4 = false;
                    // (three image buffers here: the four-buffer form of the body, VAR bit 4, is 2-3 % SLOWER with fp32 operands —
                    //  0.557 vs 0.543 ms at 256, 2.52 vs 2.46 ms at 512, profiles/r05_j_ab512.txt — and 2-3 % faster with 24-bit ones)
                    const size_t smem_h = 3 * (size_t)(2 * 2 * (H / 32) * 2 * (32 * 16 + 16)) + 512;   // three buffers x (X | Y) x 2 pieces + the flags
                    if (!attr4) {
                        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_f16p_kernel<H, 9>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
                        if (e != hipSuccess) return (int)e;
                        attr4 = true;
                    }
                    if (ntz == 4 && !dudf_deterministic()) {                       // 2 x 2 tiles: a group's tiles on one XCD
                        WgradArgs b = a;
                        b.remap_nsplit = nsplit;
                        const int groups = nl * nsplit, grid1 = ((groups + 7) / 8) * 32;
                        hipLaunchKernelGGL((wgrad_hidden_f16p_kernel<H, 9>), dim3(grid1), dim3(NTHR), smem_h, st, b);
                    } else {
                        hipLaunchKernelGGL((wgrad_hidden_f16p_kernel<H, 9>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_h, st, a);
                    }
                    return (int)hipGetLastError();
                }
                if (!attr3) {                                                       // bf16x6 (DUDF_SPLIT=bf16), same body
                    const size_t smem_cs = smem_p / 2 * 3 + 512;                     // three buffers + the flags
                    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_hidden_bf16p_kernel<H, 9>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_cs);
                    if (e != hipSuccess) return (int)e;
                    attr3 = true;
                }
                hipLaunchKernelGGL((wgrad_hidden_bf16p_kernel<H, 9>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_p / 2 * 3 + 512, st, a);
                return (int)hipGetLastError();
            }
        }
        hipLaunchKernelGGL((wgrad_hidden_bf16_kernel<H>), dim3(nl, nsplit, ntz), dim3(NTHR), smem_bf, st, a);
    }
    return (int)hipGetLastError();
}

}  // namespace

// dW / db of the layers [layer_begin, layer_end) in theta order: 0 = first layer (3 -> H), 1 .. L-1 = hidden matrices,
// L = output layer.  The two thin layers share one kernel (one pass over the columns): it runs when either is asked for.
int dudf_launch_wgrad(const DudfLayout& lo, float* ws, float* dtheta, int have_g, hipStream_t st, int layer_begin,
                      int layer_end) {
    if (layer_begin < 0) layer_begin = 0;
    if (layer_end > lo.L + 1) layer_end = lo.L + 1;
    WgradArgs a;
    a.Q = ws + lo.ws_Q; a.A = ws + lo.ws_A; a.Z = ws + lo.ws_Z; a.S = ws + lo.ws_S; a.ncol_h = lo.ncol_h;
    a.dtheta = dtheta; a.np = lo.np; a.stash_layer = lo.stash_layer;
    a.off_hid = lo.off_hid; a.hid_stride = lo.hid_stride; a.steps_total = (int)(lo.ncols / KT); a.L = lo.L;
    a.have_g = have_g; a.Hs = lo.H;
    a.amax = reinterpret_cast<const unsigned*>(ws + lo.ws_amax);
    a.clk = dudf_prof_clk(PROF_WGRAD_HIDDEN);
    a.remap_nsplit = 0;
    a.p24 = lo.p24 & 1;
    a.fxS = ws + lo.ws_fx[0]; a.fxQ = ws + lo.ws_fx[1]; a.fxA = ws + lo.ws_fx[2]; a.fxZ = ws + lo.ws_fx[3];
    const int hb = layer_begin < 1 ? 1 : layer_begin, he = layer_end > lo.L ? lo.L : layer_end;   // hidden matrices asked for
    a.j0 = hb - 1; a.nj = he > hb ? he - hb : 0;
    int rc = 0;
    if (a.nj > 0) {
        DudfProfScope prof(PROF_WGRAD_HIDDEN, st);
        switch (lo.H) {
            case 32: rc = launch_hidden<32>(a, st); break;
            case 64: rc = launch_hidden<64>(a, st); break;
            case 128: rc = launch_hidden<128>(a, st); break;
            case 256: rc = launch_hidden<256>(a, st); break;
            case 512: rc = launch_hidden<256>(a, st); break;      // 2 x 2 output tiles of 256 x 256
            default: return DUDF_E_BADCFG;
        }
    }
    if (rc) return rc;
    if (!(layer_begin <= 0 || layer_end >= lo.L + 1)) return 0;     // neither thin layer in range
    WgradSmallArgs s;
    s.Q = a.Q; s.A = a.A; s.Z = a.Z; s.S = a.S; s.x4 = ws + lo.ws_x4; s.gbar = ws + lo.ws_gbar; s.ybar = ws + lo.ws_ybar;
    s.dtheta = dtheta; s.np = lo.np; s.ncols = lo.ncols; s.stash_layer = lo.stash_layer;
    s.off_wo = lo.off_wo; s.off_bo = lo.off_bo; s.H = lo.H; s.L = lo.L; s.have_g = have_g; s.rho = lo.rho;
    s.fxS = a.fxS; s.fxQ = a.fxQ; s.fxA = a.fxA; s.fxZ = a.fxZ;
    s.pts_per_block = dudf_deterministic() ? (int)lo.ncols : 4096;     // with 16 feature-quad groups 4096 columns per block measured best (0.096 ms; 1024 x 4 groups: 0.156)
    const int grid = (int)((lo.ncols + s.pts_per_block - 1) / s.pts_per_block);
    DudfProfScope prof(PROF_WGRAD_SMALL, st);
    const int fqn = lo.H / 4;                                              // feature quads; a block's 4 waves take one each per round:
    const int gy = fqn >= 64 ? 16 : (fqn >= 4 ? fqn / 4 : 1);              // up to 16 groups -> more loads in flight per CU
    if (lo.p24 & 1) {                               // 24-bit tile-major arrays: one block row per feature tile, whole groups per block
        if (!dudf_deterministic()) {
            // (more blocks than the fp32 kernel: the decode + scale loads make a wave's chain longer.)  Large batches: 4096 columns
            // per block; the reference's batch (29 970 columns: 8 x 16 blocks, half the CUs idle, 74 us against 86 us at 100 000)
            // gets at least ~30 column blocks, in multiples of 256 columns
            s.pts_per_block = DUDF_WGSMALL_P24_PTS;
            const int64_t want = (lo.ncols / 30 + 255) / 256 * 256;
            if (want < s.pts_per_block) s.pts_per_block = want < 512 ? 512 : (int)want;
        }
        s.pts_per_block = (s.pts_per_block + 15) / 16 * 16;
        hipLaunchKernelGGL(wgrad_small_p24_kernel, dim3((unsigned)((lo.ncols + s.pts_per_block - 1) / s.pts_per_block), lo.H / 16),
                           dim3(dudf_deterministic() ? 64 : 256), 0, st, s);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(wgrad_small_kernel, dim3(grid, gy), dim3(256), 0, st, s);
    return (int)hipGetLastError();
}
