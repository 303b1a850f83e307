// The four SIREN sweeps of the DiffUDF training hot path as ONE kernel template (gfx950 / CDNA4).
//
//   SWEEP_FWD      value forward           z_l = W_l h_{l-1} + b_l, h_l = sin(w0 z_l), y          (reference src/model.py:131-135)
//   SWEEP_REV      input gradient          a_{l-1} = W_l^T (w0 c_l a_l), df/dx = a_0             (src/diff_operators.py:208-212)
//   SWEEP_ADJ_FWD  adjoint of SWEEP_REV    Q_l = W_l A_{l-1}, A_l = w0 c_l Q_l,  A_0 = gbar      (train.py:221 backward)
//   SWEEP_ADJ_REV  adjoint of SWEEP_FWD    zbar_l = w0 c_l hbar_l - e_l, hbar_{l-1} = W_l^T zbar_l
//
// Mapping (MI355X-first, nothing here is a translated CUDA tiling):
//   * one wave owns 16 points for a whole sweep.  A layer is  OUT[feature][point] = M[feature][k] * IN[k][point]
//     on v_mfma_f32_16x16x4_f32 (exact fp32): the weight rows are the A operand, the activations the B
//     operand.  The D tile (row = 4*(lane>>4)+reg = feature, col = lane&15 = point) of layer l is — after the
//     elementwise sin/cos epilogue applied in registers — directly the B operand of layer l+1
//     (B[k = lane>>4][j = lane&15], MFMA t of k-tile T consumes register t, i.e. k = 16T + 4*(lane>>4) + t,
//     and the A operand takes the matching columns with one 16-byte LDS read).  Activations never touch LDS.
//   * the 8 waves of a workgroup (128 points) walk the layers in lockstep and share the weights: every
//     32-row chunk of W_l (or of the pre-transposed W_l^T for the reverse sweeps) is staged ONCE per
//     workgroup global -> registers -> LDS (double buffered, one barrier per chunk) and read by all waves
//     as A operands.  A chunk completes two 16-feature output tiles per wave over the full K, so the
//     sin/cos + stash epilogue of chunk r overlaps the MFMAs of chunk r+1.
//   * what a later sweep needs (s_l, c_l, q_l, r_l/e_l, A_l, zbar_l) is stashed in HBM as one aligned
//     16-byte store per lane and tile ([layer][feature/4][point][4]); the same arrays are the operands of
//     the weight-gradient GEMM (dudf_wgrad.hip).
#include "dudf_internal.h"
#include "dudf_math.h"

namespace {

constexpr int NW = 4;                             // waves per workgroup (one per SIMD); two workgroups share a CU
constexpr int TILE = NW * 16;                     // points per workgroup pass

template <int H>
struct Geo {
    static constexpr int NT = H / 16;             // 16-feature tiles per activation vector
    static constexpr int NCH = H / 32;            // 32-row weight chunks per layer
    static constexpr int LDW = H + 4;             // padded LDS row stride in floats
    static constexpr int BUF = 32 * LDW;          // floats per LDS chunk buffer
    static constexpr int NTHR = 64 * NW;
    static constexpr int F4 = 8 * H;              // float4 per chunk
    static constexpr bool DMA = (H % 256 == 0);   // an LDS-DMA wave-instruction moves 1 KiB = a weight row (or half of one)
    static constexpr int WPSIMD = (H > 256) ? 1 : 2;   // H = 512: 2 x 128 activation registers -> one wave per SIMD
    static constexpr int NSTG = DMA ? 1 : (F4 + NTHR - 1) / NTHR;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- weight chunk staging: rows [32r, 32r+32) of a row-major HxH matrix -> LDS buffer with padded rows -----
// H == 256: LDS-DMA (global_load_lds_dwordx4): each wave-instruction moves one whole 1 KiB row, so the rows
//           can keep their +4-float padding; no staging registers, no ds_write.  Completion is the issuing
//           wave's vmcnt, published to the other waves by the chunk barrier.
// H <  256: global_load_dwordx4 -> registers (issue) ... ds_write_b128 (commit) one chunk later.
template <int H>
__device__ __forceinline__ void stage_issue(const float* __restrict__ M, int r, float* buf,
                                            f32x4 (&stg)[Geo<H>::NSTG], int tid) {
    using G = Geo<H>;
    if constexpr (G::DMA) {
        const int lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
        for (int i = 0; i < (32 / NW) * (H / 256); ++i) {
            const int row = wave * (32 / NW) + i / (H / 256);
            const int part = i % (H / 256);                  // 1 KiB piece of the row
            const float* g = M + (size_t)(32 * r + row) * H + part * 256 + lane * 4;
            // LDS byte address of the row, wave-uniform -> M0.  Inline asm on purpose: a builtin LDS-DMA makes
            // hipcc wait vmcnt(0) before the next ds_read of ANY LDS address (it cannot prove the two chunk
            // buffers distinct), which would expose the whole DMA latency at every chunk.  The matching wait is
            // dma_wait() in front of the chunk barrier.
            const unsigned l = __builtin_amdgcn_readfirstlane(
                (unsigned)(size_t)(__attribute__((address_space(3))) float*)(buf + row * G::LDW + part * 256));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(l) : "memory");
        }
    } else {
#pragma unroll
        for (int j = 0; j < G::NSTG; ++j) {
            const int f = tid + G::NTHR * j;
            if (G::F4 % G::NTHR == 0 || f < G::F4) {
                const int row = f / (H / 4), c4 = f % (H / 4);
                stg[j] = *reinterpret_cast<const f32x4*>(M + (size_t)(32 * r + row) * H + 4 * c4);
            }
        }
    }
}

template <int H>
__device__ __forceinline__ void stage_commit(float* buf, const f32x4 (&stg)[Geo<H>::NSTG], int tid) {
    using G = Geo<H>;
    if constexpr (!G::DMA) {
#pragma unroll
        for (int j = 0; j < G::NSTG; ++j) {
            const int f = tid + G::NTHR * j;
            if (G::F4 % G::NTHR == 0 || f < G::F4) {
                const int row = f / (H / 4), c4 = f % (H / 4);
                *reinterpret_cast<f32x4*>(buf + row * G::LDW + 4 * c4) = stg[j];
            }
        }
    }
}

// Stash addressing: `ub` is a WAVE-UNIFORM float offset (layer and tile folded in, lives in SGPRs),
// `vo` the lane's 32-bit float offset ((quarter*np + point)*4): global_load/store take the saddr form
// and no per-tile 64-bit address is kept in VGPRs.
#define DUDF_AT(arr, ub, vo) reinterpret_cast<f32x4*>((arr) + (ub) + (vo))
#define DUDF_CAT(arr, ub, vo) reinterpret_cast<const f32x4*>((arr) + (ub) + (vo))

// ---- quad (4 adjacent lanes = the 4 channels of one Hessian-path point) helpers: DPP, no LDS -------------
__device__ __forceinline__ float quad_bcast0(float v) {          // value of the quad's lane 0 (the value channel)
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x00, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {             // sum over the quad, in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    return v;
}

constexpr bool is_hess(int SW) { return SW >= 4; }      // channels that are not all "value" columns (quads, jets)
constexpr bool is_jet(int SW) { return SW == SWEEP_FWD_J; }
constexpr int base_of(int SW) { return SW & 3; }

// Elementwise tail of one 16-feature x 16-column tile.
// FL (compile-time, so the tail stays one basic block that can be interleaved with MFMAs):
//   SWEEP_FWD: bit0 = stash s_l, bit1 = stash c_l;  SWEEP_REV: bit0 = training (stash q_l, r_l);
//   SWEEP_ADJ_REV: bit0 = e_l exists (df/dx terms present)
//   Hessian variants: SWEEP_FWD_H / SWEEP_REV_H bit0 = training; the adjoint ones ignore FL.
// Hessian quads (lane&3 = channel, 0 = value, 1+k = d/dx_k):
//   FWD_H      z|zdot^k        -> h = s | hdot^k = w0 c zdot^k                      stash C = c, ZS = s|zdot^k, S = out
//   REV_H      a|adot^k        -> q = w0 c a | qdot^k = w0(-w0 s zdot^k a + c adot^k)   stash Q = out, R = a|adot^k
//   ADJ_FWD_H  Q|Qdot^k        -> A = w0 c Q + w0 sum_k cdot^k Qdot^k | Adot^k = w0 c Qdot^k          stash A = out,
//              E = w0 c sbar_rev - w0 s cbar_rev | zdotbar_rev^k = -w0 s chat^k, chat^k = w0 a Qdot^k,
//              cbar_rev = w0 a Q + w0 sum_k adot^k Qdot^k, sbar_rev = -w0 sum_k zdot^k chat^k
//   ADJ_REV_H  hbar|hdotbar^k  -> zbar = E + w0 c hbar - w0^2 s sum_k zdot^k hdotbar^k | zdotbar^k = E + w0 c hdotbar^k
// Third-order jets (SWEEP_FWD_J, query only, nothing stashed): a 16-column tile is ONE point — its columns carry the
// Taylor coefficients of  (s,r,t) -> z(x + s A + r B + t C)  for the monomials
//   0: 1 | 1: s  2: r  3: t | 4: ss  5: rr  6: tt  7: sr  8: st  9: rt | 10: sst  11: rrt  12: srt  13: stt  14: rtt | 15: 0
// (A, B, C = the three direction columns of x4).  The matmuls act on every coefficient alike; the sine composes them:
// with u = w0 (z - z_0), sin(w0 z) = s + c u - s u^2/2 - c u^3/6 + ..., i.e. for monomial m
//   h_m = c U_m - s [m](u^2/2) - c [m](u^3/6),     U_m = w0 z_m
// e.g. [ss] = U_s^2/2, [st] = U_s U_t, [sst](u^2/2) = U_ss U_t + U_s U_st, [sst](u^3/6) = U_s^2 U_t/2,
// [srt](u^2/2) = U_sr U_t + U_st U_r + U_rt U_s, [srt](u^3/6) = U_s U_r U_t.  kJetLane packs, per monomial, which
// lanes of the 16-group feed those products.  2 y_sst = d^3 f[A,A,C] etc. are the mixed third derivatives the
// curvature of the Hessian's eigenvector field needs (reference src/render_st.py:42-55), without the cancellation a
// polarisation of pure directional derivatives would suffer.
// word: bits 0-3/4-7/8-11 second-order source lanes (15 = the zero column), 12-13/14-15/16-17 the first-order factor
// paired with each, 18-19/20-21/22-23 factors a,b,c of the pure first-order product, 24-25 weight of F_a F_b in u^2/2
// (0, 1 = 1/2, 2 = 1), 26-27 weight of F_a F_b F_c in u^3/6, bit 28: value column.
constexpr unsigned jet_word(int sl0, int f0, int sl1, int f1, int sl2, int f2, int a, int b, int c, int w2, int w3,
                            int isval = 0) {
    return (unsigned)sl0 | ((unsigned)sl1 << 4) | ((unsigned)sl2 << 8) | ((unsigned)f0 << 12) | ((unsigned)f1 << 14) |
           ((unsigned)f2 << 16) | ((unsigned)a << 18) | ((unsigned)b << 20) | ((unsigned)c << 22) |
           ((unsigned)w2 << 24) | ((unsigned)w3 << 26) | ((unsigned)isval << 28);
}
__constant__ unsigned kJetLane[16] = {
    jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 0, 0, 1),                                   // value
    jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 0, 0), jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 0, 0),
    jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 0, 0),                                      // s, r, t
    jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 1, 0), jet_word(15, 0, 15, 0, 15, 0, 1, 1, 0, 1, 0),   // ss, rr
    jet_word(15, 0, 15, 0, 15, 0, 2, 2, 0, 1, 0), jet_word(15, 0, 15, 0, 15, 0, 0, 1, 0, 2, 0),   // tt, sr
    jet_word(15, 0, 15, 0, 15, 0, 0, 2, 0, 2, 0), jet_word(15, 0, 15, 0, 15, 0, 1, 2, 0, 2, 0),   // st, rt
    jet_word(4, 2, 8, 0, 15, 0, 0, 0, 2, 0, 1),                                        // sst: ss*t + st*s ; s s t / 2
    jet_word(5, 2, 9, 1, 15, 0, 1, 1, 2, 0, 1),                                        // rrt: rr*t + rt*r ; r r t / 2
    jet_word(7, 2, 8, 1, 9, 0, 0, 1, 2, 0, 2),                                         // srt: sr*t + st*r + rt*s ; s r t
    jet_word(6, 0, 8, 2, 15, 0, 2, 2, 0, 0, 1),                                        // stt: tt*s + st*t ; t t s / 2
    jet_word(6, 1, 9, 2, 15, 0, 2, 2, 1, 0, 1),                                        // rtt: tt*r + rt*t ; t t r / 2
    jet_word(15, 0, 15, 0, 15, 0, 0, 0, 0, 0, 0)};                                     // spare: stays zero

template <int SW, int FL>
__device__ __forceinline__ f32x4 epilogue(const SweepArgs& a, f32x4 acc, f32x4 o1, f32x4 o2, f32x4 o3, int64_t ub,
                                          unsigned vo, bool isv) {
    f32x4 out;
    if constexpr (SW == SWEEP_FWD) {
        f32x4 s, c;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float sv, cv;
            dudf_sincos(a.w0 * acc[t], &sv, &cv);
            s[t] = sv; c[t] = cv;
        }
        if constexpr (FL & 1) *DUDF_AT(a.S, ub, vo) = s;
        if constexpr (FL & 2) *DUDF_AT(a.C, ub, vo) = c;
        out = s;
    } else if constexpr (SW == SWEEP_REV) {          // acc = a_l, o1 = c_l, o2 = s_l
        out = a.w0 * o1 * acc;                       // q_l = w0 c_l a_l
        if constexpr (FL & 1) {
            *DUDF_AT(a.Q, ub, vo) = out;
            *DUDF_AT(a.R, ub, vo) = (a.w0 * a.w0) * o2 * acc;   // r_l = w0^2 s_l a_l
        }
    } else if constexpr (SW == SWEEP_ADJ_FWD) {      // acc = Q_l, o1 = c_l, o2 = r_l
        out = a.w0 * o1 * acc;                       // A_l = w0 c_l Q_l
        *DUDF_AT(a.A, ub, vo) = out;
        *DUDF_AT(a.E, ub, vo) = o2 * acc;            // e_l = r_l Q_l
    } else if constexpr (SW == SWEEP_ADJ_REV) {      // acc = hbar_l, o1 = c_l, o2 = e_l
        out = a.w0 * o1 * acc - o2;                  // zbar_l
        *DUDF_AT(a.Z, ub, vo) = out;
    } else if constexpr (SW == SWEEP_FWD_H) {
        f32x4 c, zs;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float sv, cv;
            dudf_sincos(a.w0 * quad_bcast0(acc[t]), &sv, &cv);
            c[t] = cv;
            zs[t] = isv ? sv : acc[t];
            out[t] = isv ? sv : a.w0 * cv * acc[t];
        }
        *DUDF_AT(a.C, ub, vo) = c;
        *DUDF_AT(a.ZS, ub, vo) = zs;
        if constexpr (FL & 1) *DUDF_AT(a.S, ub, vo) = out;
    } else if constexpr (SW == SWEEP_REV_H) {        // o1 = c, o2 = s|zdot^k
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float sv = quad_bcast0(o2[t]), a0 = quad_bcast0(acc[t]);
            out[t] = isv ? a.w0 * o1[t] * acc[t] : a.w0 * (o1[t] * acc[t] - a.w0 * sv * o2[t] * a0);
        }
        if constexpr (FL & 1) {
            *DUDF_AT(a.Q, ub, vo) = out;
            *DUDF_AT(a.R, ub, vo) = acc;
        }
    } else if constexpr (SW == SWEEP_ADJ_FWD_H) {    // o1 = c, o2 = s|zdot^k, o3 = a|adot^k
        f32x4 e;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float sv = quad_bcast0(o2[t]), a0 = quad_bcast0(o3[t]);
            const float cdot = -a.w0 * sv * o2[t];
            const float chat = a.w0 * a0 * acc[t];
            const float s1 = quad_sum(isv ? 0.f : cdot * acc[t]);
            const float s2 = quad_sum(isv ? 0.f : o3[t] * acc[t]);
            const float s3 = quad_sum(isv ? 0.f : o2[t] * chat);
            const float cbar = a.w0 * (o3[t] * acc[t] + s2);
            const float sbar = -a.w0 * s3;
            out[t] = a.w0 * (o1[t] * acc[t] + (isv ? s1 : 0.f));
            e[t] = isv ? a.w0 * (o1[t] * sbar - sv * cbar) : -a.w0 * sv * chat;
        }
        *DUDF_AT(a.A, ub, vo) = out;
        *DUDF_AT(a.E, ub, vo) = e;
    } else if constexpr (SW == SWEEP_FWD_J) {
        const int lane = threadIdx.x & 63, l0 = lane & 48;
        const unsigned jw = kJetLane[lane & 15];
        const float w2 = 0.5f * (float)((jw >> 24) & 3), w3 = 0.5f * (float)((jw >> 26) & 3);
        const bool isval = (jw >> 28) & 1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float sv, cv;
            dudf_sincos(a.w0 * __shfl(acc[t], l0), &sv, &cv);
            const float F[3] = {a.w0 * __shfl(acc[t], l0 + 1), a.w0 * __shfl(acc[t], l0 + 2), a.w0 * __shfl(acc[t], l0 + 3)};
            auto sel = [&](unsigned k) -> float { k &= 3; return k == 0 ? F[0] : (k == 1 ? F[1] : F[2]); };
            const float S0 = a.w0 * __shfl(acc[t], l0 + (int)(jw & 15)), S1 = a.w0 * __shfl(acc[t], l0 + (int)((jw >> 4) & 15)),
                        S2 = a.w0 * __shfl(acc[t], l0 + (int)((jw >> 8) & 15));
            const float fab = sel(jw >> 18) * sel(jw >> 20);
            const float p2 = S0 * sel(jw >> 12) + S1 * sel(jw >> 14) + S2 * sel(jw >> 16) + w2 * fab;
            const float p3 = w3 * fab * sel(jw >> 22);
            out[t] = isval ? sv : cv * (a.w0 * acc[t] - p3) - sv * p2;
        }
    } else {                                         // SWEEP_ADJ_REV_H: o1 = c, o2 = s|zdot^k, o3 = E
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float sv = quad_bcast0(o2[t]);
            const float st = quad_sum(isv ? 0.f : o2[t] * acc[t]);
            out[t] = o3[t] + a.w0 * o1[t] * acc[t] - (isv ? a.w0 * a.w0 * sv * st : 0.f);
        }
        *DUDF_AT(a.Z, ub, vo) = out;
    }
    return out;
}

template <int SW, int FL>
__device__ __forceinline__ void epilogue_loads(const SweepArgs& a, int64_t ub, unsigned vo, f32x4& o1, f32x4& o2,
                                               f32x4& o3) {
    o1 = f32x4{0, 0, 0, 0}; o2 = o1; o3 = o1;
    if constexpr (SW == SWEEP_REV) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        if constexpr (FL & 1) o2 = *DUDF_CAT(a.S, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_FWD) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = *DUDF_CAT(a.R, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_REV) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        if constexpr (FL & 1) o2 = *DUDF_CAT(a.E, ub, vo);            // no df/dx terms (loss_s2): e_l == 0
    } else if constexpr (SW == SWEEP_REV_H) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = *DUDF_CAT(a.ZS, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_FWD_H) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = *DUDF_CAT(a.ZS, ub, vo);
        o3 = *DUDF_CAT(a.R, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_REV_H) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = *DUDF_CAT(a.ZS, ub, vo);
        o3 = *DUDF_CAT(a.E, ub, vo);
    }
}

// The LDS-DMA pieces this wave issued have landed.  vmcnt counts every vector-memory operation of the wave in
// issue order, and the asm statements pin that order: after the DMA pieces of a chunk come exactly
// younger_ops<SW,FL>() compiler-issued operations (operand loads of the next tail + stash stores of the current
// one), so waiting for "all but that many" retires the DMA while those stay in flight.
// tests/test_isa_contract.py counts the instructions in the built code object and fails if this drifts.
template <int SW, int FL>
constexpr int younger_ops() {
    return SW == SWEEP_FWD ? 2 + 2 * ((FL & 1) + ((FL >> 1) & 1))      // 2 bias loads + s/c stores of 2 tiles
         : SW == SWEEP_REV ? ((FL & 1) ? 4 + 4 : 2)                     // c,s loads + q,r stores | c loads
         : SW == SWEEP_ADJ_FWD ? 4 + 4                                  // c,r loads + A,e stores
         : SW == SWEEP_ADJ_REV ? ((FL & 1) ? 4 + 2 : 2 + 2)             // c(,e) loads + zbar stores
         : SW == SWEEP_FWD_H ? 2 + 2 * (2 + (FL & 1))                   // bias + C,ZS(,S) stores
         : SW == SWEEP_REV_H ? 4 + ((FL & 1) ? 4 : 0)                   // c,zs loads + Q,R stores
         : SW == SWEEP_ADJ_FWD_H ? 6 + 4                                // c,zs,aa loads + A,E stores
         : SW == SWEEP_FWD_J ? 2                                        // bias loads only
         : 6 + 2;                                                       // c,zs,E loads + Z stores
}
template <int H, int N>
__device__ __forceinline__ void dma_wait() {
    if constexpr (Geo<H>::DMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// Two finished accumulator tiles whose elementwise tail has not run yet.  The tail of chunk r-1 is executed in
// the middle of chunk r's MFMA stream (same basic block), so sin/cos, stash traffic and MFMAs overlap inside
// one wave instead of serialising at every chunk barrier.
struct Pending {
    f32x4 acc0, acc1, o1a, o2a, o3a, o1b, o2b, o3b;
    int64_t ub0, ub1;
};

// One sweep over one 64-column tile.  `seed` replaces the per-column operand the prologue would read from HBM when
// the caller already has it in registers (fused step kernel: gbar[q] for SWEEP_ADJ_FWD, ybar for SWEEP_ADJ_REV);
// `res` returns what the tail produced: y in [0] (SWEEP_FWD, every lane), the a_0 rows in [0..2] (SWEEP_REV, lanes < 16).
template <int H, int SW, int FL, bool SEEDED = false>
__device__ __forceinline__ void sweep_tile(const SweepArgs& a, const int tile, float* lds, unsigned& gc, float seed,
                                           f32x4& res) {
    using G = Geo<H>;
    constexpr int BS = base_of(SW);
    constexpr bool HS = is_hess(SW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const bool isv = !HS || (is_jet(SW) ? li == 0 : (lane & 3) == 0);   // value channel (always, on the plain path)
    const int nhid = a.L - 1;                          // hidden x hidden layers
    constexpr bool kFwdDir = (BS == SWEEP_FWD || BS == SWEEP_ADJ_FWD);

    f32x4 in[G::NT], nxt[G::NT];
    f32x4 stg[G::NSTG];
    {
        const int64_t p = (int64_t)tile * TILE + wave * 16 + li;        // this lane's column
        // j-th hidden matrix this sweep multiplies by, and the 0-based layer index its output belongs to
        auto matrix = [&](int j) -> const float* {
            return kFwdDir ? a.theta + a.off_hid + (int64_t)j * a.hid_stride
                           : a.wt + (int64_t)(nhid - 1 - j) * H * H;
        };
        auto out_layer = [&](int j) -> int { return kFwdDir ? j + 1 : nhid - 1 - j; };
        auto stash_base = [&](int layer, int T) -> int64_t {                   // wave-uniform
            return (int64_t)layer * a.stash_layer + (int64_t)(16 * T) * a.np;
        };
        const unsigned vo = (unsigned)(((int64_t)q * a.np + p) * 4);           // this lane's granule

        // every wave of the workgroup is past its last LDS read of the previous tile before buffers are refilled
        __syncthreads();
        if (nhid > 0) stage_issue<H>(matrix(0), 0, lds + (gc & 1) * G::BUF, stg, tid);

        // ------------------------------ prologue: first activation vector ------------------------------
        Pending pend;
        {
            float b = 0.f, yb = 1.f;
            if constexpr (BS == SWEEP_FWD) b = a.x4[p * 4 + q];                       // (x,1) | (e_k,0): k=3 carries the bias
            if constexpr (BS == SWEEP_ADJ_FWD) b = SEEDED ? seed : ((q < 3) ? a.gbar[p * 4 + q] : 0.f);   // A_0 = gbar | Hbar columns
            if constexpr (BS == SWEEP_ADJ_REV) yb = SEEDED ? seed : a.ybar[p];
            if constexpr (SW == SWEEP_REV_H) yb = isv ? 1.f : 0.f;                    // adot_L^k = 0
            const int l0 = kFwdDir ? 0 : a.L - 1;
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const int64_t ub = stash_base(l0, T);
                f32x4 o1, o2, o3, acc;
                epilogue_loads<SW, FL>(a, ub, vo, o1, o2, o3);
                if constexpr (kFwdDir) {
                    acc = mfma16(a.w1b[(16 * T + li) * 4 + q], b, f32x4{0, 0, 0, 0});
                } else {
                    acc = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
                }
                if (T < G::NT - 2) {
                    in[T] = epilogue<SW, FL>(a, acc, o1, o2, o3, ub, vo, isv);
                } else if (T == G::NT - 2) {
                    pend.acc0 = acc; pend.o1a = o1; pend.o2a = o2; pend.o3a = o3; pend.ub0 = ub;
                } else {
                    pend.acc1 = acc; pend.o1b = o1; pend.o2b = o2; pend.o3b = o3; pend.ub1 = ub;
                }
            }
        }

        // ------------------------------ hidden x hidden layers ------------------------------
        // Order inside a chunk (it matters: hipcc does not see the asm LDS-DMA in its vmcnt bookkeeping, so no
        // compiler-visible vector-memory wait may sit between a DMA issue and dma_wait(), or it drains the DMA):
        //   first k-tile MFMAs -> tail of the previous chunk (waits its operands, issues its stash stores)
        //   -> DMA issue for the next chunk -> operand/bias loads for the next tail -> remaining MFMAs
        //   -> dma_wait + barrier.
        f32x4 bias0 = {0, 0, 0, 0}, bias1 = {0, 0, 0, 0};
        if constexpr (BS == SWEEP_FWD) {
            if (nhid > 0) {
                const float* bias = matrix(0) + (size_t)H * H;
                bias0 = *reinterpret_cast<const f32x4*>(bias + 4 * q);
                bias1 = *reinterpret_cast<const f32x4*>(bias + 16 + 4 * q);
            }
        }
        for (int j = 0; j < nhid; ++j) {
            const float* M = matrix(j);
            const float* Mn = (j + 1 < nhid) ? matrix(j + 1) : nullptr;
            const int lo = out_layer(j);
            stage_commit<H>(lds + (gc & 1) * G::BUF, stg, tid);
            if (j == 0) dma_wait<H, 0>();               // the prologue's loads and stores sit behind this DMA
            else dma_wait<H, younger_ops<SW, FL>()>();
            __syncthreads();
#pragma unroll
            for (int r = 0; r < G::NCH; ++r) {
                Pending cur;
                cur.ub0 = stash_base(lo, 2 * r); cur.ub1 = stash_base(lo, 2 * r + 1);
                // the bias loads issued during the previous chunk have landed: make the compiler wait for them
                // here, before this chunk's LDS-DMA goes in flight
                if constexpr (BS == SWEEP_FWD) asm volatile("" : "+v"(bias0), "+v"(bias1));
                if constexpr (HS) {                      // z = W h + b only in the value channel
                    cur.acc0 = isv ? bias0 : f32x4{0, 0, 0, 0};
                    cur.acc1 = isv ? bias1 : f32x4{0, 0, 0, 0};
                } else {
                    cur.acc0 = bias0; cur.acc1 = bias1;
                }
                const float* bp = lds + (gc & 1) * G::BUF + li * G::LDW + 4 * q;
                auto chunk_head = [&]() {
                    // operands of the previous chunk's tail have landed: the compiler puts its own vmcnt wait
                    // HERE, while no LDS-DMA of this chunk is in flight yet
                    if constexpr (BS != SWEEP_FWD)
                        asm volatile("" : "+v"(pend.o1a), "+v"(pend.o2a), "+v"(pend.o1b), "+v"(pend.o2b));
                    if constexpr (SW == SWEEP_ADJ_FWD_H || SW == SWEEP_ADJ_REV_H)
                        asm volatile("" : "+v"(pend.o3a), "+v"(pend.o3b));
                    float* nbuf = lds + ((gc + 1) & 1) * G::BUF;
                    if (r + 1 < G::NCH) stage_issue<H>(M, r + 1, nbuf, stg, tid);
                    else if (Mn) stage_issue<H>(Mn, 0, nbuf, stg, tid);
                    epilogue_loads<SW, FL>(a, cur.ub0, vo, cur.o1a, cur.o2a, cur.o3a);
                    epilogue_loads<SW, FL>(a, cur.ub1, vo, cur.o1b, cur.o2b, cur.o3b);
                    if constexpr (BS == SWEEP_FWD) {
                        const float* bias = (r + 1 < G::NCH) ? M + (size_t)H * H + 32 * (r + 1)
                                                             : (Mn ? Mn + (size_t)H * H : M + (size_t)H * H);
                        bias0 = *reinterpret_cast<const f32x4*>(bias + 4 * q);
                        bias1 = *reinterpret_cast<const f32x4*>(bias + 16 + 4 * q);
                    }
                    // loads stay up here; everything below (MFMAs, the tail's VALU and stash stores) is one
                    // freely interleavable region
                    __builtin_amdgcn_sched_barrier(0);
                    // tail of the previous chunk's two tiles (the previous layer's last two when r == 0)
                    const f32x4 e0 = epilogue<SW, FL>(a, pend.acc0, pend.o1a, pend.o2a, pend.o3a, pend.ub0, vo, isv);
                    const f32x4 e1 = epilogue<SW, FL>(a, pend.acc1, pend.o1b, pend.o2b, pend.o3b, pend.ub1, vo, isv);
                    if (r == 0) { in[G::NT - 2] = e0; in[G::NT - 1] = e1; }
                    else { nxt[2 * r - 2] = e0; nxt[2 * r - 1] = e1; }
                };
                // A operands are fetched one k-tile ahead of the MFMAs that consume them (LDS latency, doubled by the
                // 2-way bank conflict of the padded rows, stays behind 8 MFMAs instead of stalling every k-tile)
                f32x4 a0n = *reinterpret_cast<const f32x4*>(bp);
                f32x4 a1n = *reinterpret_cast<const f32x4*>(bp + 16 * G::LDW);
#pragma unroll
                for (int T = 0; T < G::NT; ++T) {
                    if (G::NT <= 2 && T == 0) chunk_head();     // tiny nets: k-tile 0 itself is one of the pending tiles
                    const f32x4 a0 = a0n, a1 = a1n;
                    if (T + 1 < G::NT) {
                        a0n = *reinterpret_cast<const f32x4*>(bp + 16 * (T + 1));
                        a1n = *reinterpret_cast<const f32x4*>(bp + 16 * G::LDW + 16 * (T + 1));
                        __builtin_amdgcn_sched_barrier(0x7F);   // everything but LDS ops may cross: the reads stay early
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        cur.acc0 = mfma16(a0[t], in[T][t], cur.acc0);
                        cur.acc1 = mfma16(a1[t], in[T][t], cur.acc1);
                    }
                    if (G::NT > 2 && T == 0) chunk_head();
                }
                pend = cur;
                if (r + 1 < G::NCH) {
                    ++gc;
                    stage_commit<H>(lds + (gc & 1) * G::BUF, stg, tid);
                    dma_wait<H, younger_ops<SW, FL>()>();
                    __syncthreads();
                }
            }
            ++gc;
#pragma unroll
            for (int T = 0; T < G::NT - 2; ++T) in[T] = nxt[T];
        }
        // flush the last pending pair
        in[G::NT - 2] = epilogue<SW, FL>(a, pend.acc0, pend.o1a, pend.o2a, pend.o3a, pend.ub0, vo, isv);
        in[G::NT - 1] = epilogue<SW, FL>(a, pend.acc1, pend.o1b, pend.o2b, pend.o3b, pend.ub1, vo, isv);

        // ------------------------------ tail ------------------------------
        if constexpr (BS == SWEEP_FWD) {                // y = W_out h_L + b_out (tangent channels: their own dot, unused)
            float part = 0.f;
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q);
                part += in[T][0] * w[0] + in[T][1] * w[1] + in[T][2] * w[2] + in[T][3] * w[3];
            }
            part += __shfl_xor(part, 16);               // the 4 lane quarters hold disjoint feature rows
            part += __shfl_xor(part, 32);
            if (isv) part += a.theta[a.off_bo];         // tangent / jet columns are derivatives: no constant term
            if (q == 0) a.y[p] = part;
            res[0] = part;
        } else if constexpr (BS == SWEEP_REV) {         // a_0 = W_1^T q_1 (rows 0..2 of a 16-row tile): df/dx | Hessian column
            f32x4 accg = {0, 0, 0, 0};
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 16 * T + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) accg = mfma16(w[t], in[T][t], accg);
            }
            if (q == 0) *reinterpret_cast<f32x4*>(a.g + p * 4) = f32x4{accg[0], accg[1], accg[2], 0.f};
            res = accg;
        }
    }
}

template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NW, Geo<H>::WPSIMD) void sweep_kernel(SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned gc = 0;                                   // running chunk counter: LDS buffer parity
    f32x4 res;
    for (int tile = a.tile0 + blockIdx.x; tile < a.tile0 + a.ntiles; tile += gridDim.x)
        sweep_tile<H, SW, FL>(a, tile, lds, gc, 0.f, res);
}

template <int H>
int launch_h(int which, const SweepArgs& a, hipStream_t st) {
    using G = Geo<H>;
    const size_t smem = 2 * G::BUF * sizeof(float);
    const int ntiles = a.ntiles;
    if (ntiles <= 0) return 0;
    const int slots = 256 * G::WPSIMD;                 // resident 4-wave workgroups: two per CU (one for H = 512)
    int grid = ntiles < slots ? ntiles : slots;
    if (grid < 1) grid = 1;
    hipError_t e = hipSuccess;
#define DUDF_GO(SW, FL)                                                                                     \
    do {                                                                                                    \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_kernel<H, SW, FL>),                \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((sweep_kernel<H, SW, FL>), dim3(grid), dim3(G::NTHR), smem, st, a);              \
    } while (0)
    switch (which) {
        case SWEEP_FWD:                                  // the stash-everything variant is the only one built: the
            if (!(a.store_s && a.store_c)) return DUDF_E_BADMODE;   // leaner ones made the register allocator spill
            DUDF_GO(SWEEP_FWD, 3);
            break;
        case SWEEP_REV:
            if (a.train) DUDF_GO(SWEEP_REV, 1); else DUDF_GO(SWEEP_REV, 0);
            break;
        case SWEEP_ADJ_FWD: DUDF_GO(SWEEP_ADJ_FWD, 0); break;
        case SWEEP_ADJ_REV:
            if (a.have_e) DUDF_GO(SWEEP_ADJ_REV, 1); else DUDF_GO(SWEEP_ADJ_REV, 0);
            break;
        // Hessian-quad variants: not built for H = 512 (hipcc 7.2 emits illegal AGPR operands for them at 512 registers)
        case SWEEP_FWD_H:
            if constexpr (H > 256) return DUDF_E_UNSUPPORTED;
            else { if (!a.store_s) return DUDF_E_BADMODE; DUDF_GO(SWEEP_FWD_H, 1); }
            break;
        case SWEEP_REV_H:
            if constexpr (H > 256) return DUDF_E_UNSUPPORTED;
            else { if (a.train) DUDF_GO(SWEEP_REV_H, 1); else DUDF_GO(SWEEP_REV_H, 0); }
            break;
        case SWEEP_ADJ_FWD_H:
            if constexpr (H > 256) return DUDF_E_UNSUPPORTED; else DUDF_GO(SWEEP_ADJ_FWD_H, 0);
            break;
        case SWEEP_ADJ_REV_H:
            if constexpr (H > 256) return DUDF_E_UNSUPPORTED; else DUDF_GO(SWEEP_ADJ_REV_H, 0);
            break;
        case SWEEP_FWD_J:
            if constexpr (H > 256) return DUDF_E_UNSUPPORTED; else DUDF_GO(SWEEP_FWD_J, 0);
            break;
        default: return DUDF_E_BADMODE;
    }
#undef DUDF_GO
    e = hipGetLastError();
    return (int)e;
}

}  // namespace

int dudf_launch_sweep(int which, int H, const SweepArgs& a, hipStream_t st) {
    DudfProfScope prof(PROF_SWEEP_FWD + (which & 3), st);
    switch (H) {
        case 32: return launch_h<32>(which, a, st);
        case 64: return launch_h<64>(which, a, st);
        case 128: return launch_h<128>(which, a, st);
        case 256: return launch_h<256>(which, a, st);
        case 512: return launch_h<512>(which, a, st);
        default: return DUDF_E_BADCFG;
    }
}
