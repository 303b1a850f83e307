// The SIREN sweeps (plain columns, Hessian quads, third-order jets; 256- and 128-wide layers) on the bf16 matrix cores at
// fp32 accuracy ("bf16x6", gfx950 / CDNA4).
//
// Same sweeps, same tails, same stash as dudf_sweep.hip (see there for what each sweep computes and which reference
// lines it replaces); what changes is how a hidden layer  OUT[feature][column] = M[feature][k] * IN[k][column]  is
// multiplied.  Every fp32 operand is split EXACTLY into three bf16 pieces v = h + m + l (8+8+8 significand bits,
// round-to-nearest at each step so the three pieces always hold all 24 bits); a product is the six partial products
// whose weight is >= 2^-16 (hh, hm, mh, hl, lh, mm — the dropped ml, lm, ll are below fp32 rounding), accumulated in
// fp32 by v_mfma_f32_16x16x32_bf16.  Six of those (16 cycles each, K = 32) replace eight v_mfma_f32_16x16x4_f32
// (32 cycles each, K = 4): 96 instead of 256 matrix-core cycles per 16x16 tile and 32 features.
//
// Mapping (output-stationary, K streamed):
//   * a wave owns 16 columns for a whole sweep; a workgroup is 8 waves = 128 columns, two waves per SIMD, one
//     workgroup per CU.  The accumulator tile of v_mfma_f32_16x16x32_bf16 has the layout of the f32 instruction
//     (row = 4*(lane>>4)+reg = feature, col = lane&15 = column), so all tails are shared (dudf_sweep_common.h) and the
//     stash layout is unchanged.
//   * a layer keeps ALL its 16 output tiles in accumulators (64 registers) and streams over the 8 k-blocks of 32
//     input features.  The B operand of k-block kb wants, per lane, 8 k of one column (k = 8*(lane>>4) + e).  K is a
//     contraction index, so a permutation applied to both operands is free: k-slot (g, e) is mapped to feature
//     32kb + 4g + e (e < 4) or 32kb + 16 + 4g + (e-4) — exactly the 4+4 accumulator registers a lane holds of tiles
//     2kb and 2kb+1 of the PREVIOUS layer.  So step kb runs the elementwise tail (bias, sin/cos or the adjoint
//     formulas, stash stores) of those two tiles of the previous layer, splits the 8 results into pieces and packs
//     them (v_cvt_pk_bf16_f32) into 12 registers: the B operand, used at once by 16 tiles x 6 MFMAs and then dead.
//     Activations never touch LDS, and the tail of step kb+1 overlaps the MFMAs of step kb inside a wave.
//   * the weights are pre-split once per step (pack kernel below) into an image in A-FRAGMENT ORDER: for every
//     (k-block, 16-row tile, piece) the 1 KiB that one ds_read_b128 wave-instruction fetches, lane L = (g<<4 | m)
//     holding M[row m][the 8 features of k-slots (g, 0..7)].  The 48 KiB of a k-block (16 tiles x 3 pieces at H = 256)
//     are contiguous, so LDS-DMA moves them verbatim in 1 KiB wave-instructions and the reads are lane-linear:
//     conflict-free without padding or swizzle.  Three buffers (144 KiB), fetched two steps ahead, one barrier per
//     k-block, hand-counted vmcnt as in the f32 kernel.  128 columns share every byte fetched from L2: at bf16 rates
//     the f32 kernel's 64-column workgroups would be bound by L2 -> LDS weight traffic, not by the matrix cores.
//   * the first layer (K = 3+1) and the output / df/dx matmuls stay on the fp32 MFMA.
#include "dudf_sweep_common.h"
#include <type_traits>

// Variants that were built, measured and NOT kept (tails of a pair half a step apart, one operand set for the fp16x3 adjoint
// forward sweep, one tile at a time, DMA pieces issued by the idle half of a partial pass, static / per-phase wave priorities,
// other fragment prefetch distances and feed / tail slots) are no longer in this file: DESIGN.md Appendix A has their numbers,
// `git log -- diffudf_amd/csrc/dudf_sweep_bf16.hip` (round 3) their code.
#ifndef DUDF_FWD_F16_KERNEL
#define DUDF_FWD_F16_KERNEL sweep_f16_np_kernel   // A/B: sweep_f16_kernel = the build WITH packed fp32 instructions
#endif
#ifndef DUDF_OCT
#define DUDF_OCT 1                           // a pass with a single 16-column group is shared by all eight waves (sweep_tile_oct)
#endif
#ifndef DUDF_W_RELAY_NT
#define DUDF_W_RELAY_NT 0                    // 512-wide kernel: 1 = non-temporal stores for the relay array as well (rounds 2-3)
#endif
#ifndef DUDF_TAIL_TOP
#define DUDF_TAIL_TOP (SP != 0 && BS == SWEEP_FWD)   // early half: tail in front of the step's DMA pieces
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one LDS atomic per wave: 64 lanes hitting the same LDS word serialise (measured: the per-layer publish of the quads' forward
// sweep cost 0.09 ms per launch that way), so the wave reduces first
__device__ __forceinline__ void lds_max_wave(unsigned* word, float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
#ifdef DUDF_LDSMAX_BRANCH
    if ((threadIdx.x & 63) == 0) atomicMax(word, __float_as_uint(v));
#else
    // lane 0 alone, WITHOUT a compiler-visible branch: `if (lane == 0)` ends the basic block, and the sweeps call this in the last
    // k-block step of every layer — the step then loses the interleaving of its tail with its MFMAs (round 5: the same shape of store
    // cost the reverse sweeps 12-18 %, profiles/r05_e_ab.txt)
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)word;
    uint64_t ex;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_max_u32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(ex) : "v"(addr), "v"(__float_as_uint(v)) : "memory");
#endif
}
constexpr int kMaxAmaxLayers = 64;                     // LDS words of the per-layer running maxima (deeper nets: no fp16x3 wgrad)
constexpr size_t kLdsCu = 160 * 1024;
constexpr int NWB = 8;                                 // waves per workgroup: two per SIMD
constexpr int TILEB = NWB * 16;                        // columns per workgroup pass

// SP = 0: exact three-piece bf16 split, six products ("bf16x6").  SP = 1: fp16 hi/lo split, three products ("fp16x3"):
// v * 2^k = hi + lo with two round-to-nearest fp16 pieces (|v 2^k - hi - lo| <= 2^-23 |v 2^k|; fp16 subnormals are
// produced by v_cvt_pk_f16_f32 and honoured by the MFMA — tools/micro/f16_split.hip, profiles/r03_f16_split_facts.txt),
// products hi*hi + hi*lo + lo*hi; the dropped lo*lo is <= 2^-22 of the product.  Half the matrix-core work, a third less
// LDS traffic, 2 conversions instead of 3 per value; the price is fp16's range: the weights are scaled per matrix by a power
// of two (pack kernel), the activations where their size is not known a priori (see `ColScale`).
template <int H, int SP = 0>
struct GeoB {
    static constexpr int NPC = SP ? 2 : 3;             // pieces per operand
    static constexpr int NT = H / 16;                  // 16-feature tiles per activation vector
    static constexpr int NKB = H / 32;                 // 32-feature k-blocks = weight chunks per layer
    static constexpr int FRAG = 1024;                  // bytes of one A fragment: 64 lanes x 8 x 16 bits
    static constexpr int CHUNKB = NT * NPC * FRAG;     // one k-block of a matrix: [tile][piece]
    static constexpr int IMGB = NKB * CHUNKB;          // one matrix: 6 (4) bytes per weight
    static constexpr int NDMA = NT * NPC / NWB;        // LDS-DMA wave-instructions per wave and chunk
    static constexpr int NTHR = 64 * NWB;
};


__device__ __forceinline__ f32x4 mfma_b(bf16x8 a, bf16x8 b, f32x4 c) {
#if DUDF_SWEEP_DBG & 4
    asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ unsigned cvt_pk(f32x2 v) {              // one v_cvt_pk_bf16_f32: lo = bf16(v.x), hi = bf16(v.y)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 unpack(unsigned p) {              // the two bf16 back as exact floats
    return f32x2{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
// 8 fp32 values (two accumulator tiles' registers of one lane) -> the three bf16x8 pieces of a B / A operand
__device__ __forceinline__ void split8(const f32x4 e0, const f32x4 e1, u32x4& h, u32x4& m, u32x4& l) {
    const f32x2 v[4] = {{e0[0], e0[1]}, {e0[2], e0[3]}, {e1[0], e1[1]}, {e1[2], e1[3]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned hp = cvt_pk(v[i]);
        const f32x2 r1 = v[i] - unpack(hp);                        // exact
        const unsigned mp = cvt_pk(r1);
        const f32x2 r2 = r1 - unpack(mp);                          // exact
        h[i] = hp; m[i] = mp; l[i] = cvt_pk(r2);
    }
}
__device__ __forceinline__ f32x4 mfma_h(f16x8 a, f16x8 b, f32x4 c) {
#if DUDF_SWEEP_DBG & 4
    asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c;
#endif
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// 8 fp32 values -> the two fp16x8 pieces hi = fp16(v), lo = fp16(v - hi) (the caller has scaled v into fp16's range)
__device__ __forceinline__ void split8h(const f32x4 e0, const f32x4 e1, u32x4& h, u32x4& l) {
    const f32x2 v[4] = {{e0[0], e0[1]}, {e0[2], e0[3]}, {e1[0], e1[1]}, {e1[2], e1[3]}};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f16x2 hp = __builtin_convertvector(v[i], f16x2);      // v_cvt_pk_f16_f32, round to nearest even
        // r = v - hi, exact.  v_fma_mix_f32 reads the fp16 half directly (no v_cvt_f32_f16) and issues beside the SIMD
        // partner's MFMAs like a plain v_fma_f32 (tools/micro/coissue.hip); hipcc folds `fma(v, 1, -hi)` back into
        // convert + subtract, hence the asm
        f32x2 r;
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r.x) : "v"(v[i].x), "v"(hp));
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r.y) : "v"(v[i].y), "v"(hp));
        h[i] = __builtin_bit_cast(unsigned, hp);
        l[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
    }
}
__device__ __forceinline__ f16x8 as_h(u32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 as_u(f32x4 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ f32x4 as_f(u32x4 v) { return __builtin_bit_cast(f32x4, v); }

// One chunk of a weight image -> LDS buffer, 1 KiB per wave-instruction (see dudf_sweep.hip for why this is inline asm).
// `lds_off` is the buffer's LDS byte offset, `voff` = lane * 16; this wave moves pieces wave*NDMA .. +NDMA-1.  One asm
// block: scalar base + lane offset addressing, M0 (the LDS destination) saved and restored once — under 2 instructions
// per piece instead of 11 through generic pointers.
template <int H, int SP = 0>
__device__ __forceinline__ void dma_issue(const char* __restrict__ chunk, unsigned lds_off, unsigned voff, int wave) {
    using G = GeoB<H, SP>;
    static_assert(G::NDMA == 6 || G::NDMA == 3 || G::NDMA == 4 || G::NDMA == 2, "asm below is written for 2, 3, 4 or 6 pieces per wave");
#if DUDF_SWEEP_DBG & 8
    return;
#endif
    const uint64_t g0 = (uint64_t)(size_t)chunk + (uint64_t)wave * (G::NDMA * G::FRAG);       // wave-uniform
    const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)g0), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
    const uint64_t sbase = ((uint64_t)hi32 << 32) | lo32;
    const unsigned l0 = __builtin_amdgcn_readfirstlane(lds_off + (unsigned)wave * (G::NDMA * G::FRAG));
    unsigned keep;
    if constexpr (G::NDMA == 6) {
        // the instruction offset is added to the global AND to the LDS address; past its 4 KiB reach: a second lane
        // offset and M0 + 4096
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %4, %2\n\t"
                     "global_load_lds_dwordx4 %4, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(l0), "v"(voff + 4096u) : "memory", "scc");
    } else if constexpr (G::NDMA == 4) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(l0) : "memory");
    } else if constexpr (G::NDMA == 3) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(l0) : "memory");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(l0) : "memory");
    }
}
template <int N>
__device__ __forceinline__ void dma_wait_b() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

#if DUDF_SWEEP_DBG & 128
// phase stamps (timing experiments): [sweep][wave][k-block][stamp] of one workgroup's first pass, layer 3
__device__ unsigned long long g_stamp[4][8][8][8];
extern "C" int dudf_dbg_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(g_stamp));
}
#define DUDF_STAMP(i) do { if (stamp_on && j == 3) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if (lane == 0) g_stamp[BS & 3][wave][kb][i] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define DUDF_STAMP(i) do { } while (0)
#endif
struct TailOps { f32x4 o1a, o2a, o3a, o1b, o2b, o3b, ba, bb; };   // operands of one pair of tiles (+ bias, forward sweeps)

// LDS offset of the per-layer running maxima: behind the three weight buffers — and behind the biases where the fp16x3 forward
// sweeps keep them (the quads' forward sweep has both)
template <int H, int SW, int SP>
__device__ __forceinline__ unsigned amax_lds_off(const SweepArgs& a) {
    return 3u * GeoB<H, SP>::CHUNKB + ((SP != 0 && base_of(SW) == SWEEP_FWD) ? (unsigned)(a.L * H * sizeof(float)) : 0u);
}

// One pass: the workgroup's waves 0..nact-1 take the 16-column groups g_first.. through a whole sweep.  Waves beyond
// nact (the last, partial pass of a workgroup's share) only keep the weight stream and the barriers going.
template <int H, int SW, int FL, int SP = 0, int P24 = 0>
__device__ __forceinline__ void sweep_tile_b(const SweepArgs& a, const int g_first, const int nact, char* lds, unsigned& gc,
                                             const bool stamp_on = false) {
    using G = GeoB<H, SP>;
    constexpr int NPC = G::NPC;
    static_assert(SP == 0 || SW <= SWEEP_FWD_J, "fp16x3: plain columns, Hessian quads, jets");
    static_assert(P24 == 0 || ((P24 == 6 || P24 == 7) && SP != 0 && H == 256 && !is_jet(SW)), "24-bit stash arrays (mask 6: R, E, C; 7: S, Q, A, Z as well): the fp16x3 training kernels of 256-wide layers");
    constexpr int BS = base_of(SW);
    constexpr bool HS = is_hess(SW);                   // quads: lane & 3 = channel (0 = value, 1 + k = tangent d/dx_k); jets:
                                                       // lane & 15 = Taylor monomial (0 = value), see dudf_sweep_common.h
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const bool isv = !HS || (is_jet(SW) ? li == 0 : (lane & 3) == 0);   // value channel (always, on the plain path)
    const int nhid = a.L - 1;                          // hidden x hidden layers (>= 1 here)
    constexpr bool kFwdDir = (BS == SWEEP_FWD || BS == SWEEP_ADJ_FWD);

    f32x4 acc[G::NT], prev[G::NT];                     // this layer's accumulators / the previous layer's, tails pending
    const int64_t p = (int64_t)(g_first + wave) * 16 + li;
    auto image = [&](int j) -> const char* {
        if constexpr (SP) return kFwdDir ? a.wimg16_f + (size_t)j * G::IMGB : a.wimg16_t + (size_t)(nhid - 1 - j) * G::IMGB;
        return kFwdDir ? a.wimg_f + (size_t)j * G::IMGB : a.wimg_t + (size_t)(nhid - 1 - j) * G::IMGB;
    };
    // fp16x3: the accumulators of matrix j hold 2^k_j (W h); `unscale` = 2^-k_j of the matrix whose outputs wait in `prev`
    // (1 for the first layer, which runs on the fp32 MFMA) — folded into the bias add of the forward tail
    float unscale = 1.f;
    auto unscale_of = [&](int j) -> float { return a.wsc[kFwdDir ? j : nhid - 1 - j]; };
    // fp16x3, B operand.  The forward sweep's is h_l = sin(.): |h| <= 1 fits fp16 as it stands (small values keep an absolute
    // error of 2^-25: fp16 subnormals are honoured).  The other sweeps' operands (q_l, A_l, zbar_l) have no a-priori size, so
    // every COLUMN gets its own power of two `sb`, fixed when the previous layer's accumulators are final — before any of
    // its tails has run — from a bound: |q_l| <= w0 max_f |a_l|, |A_l| <= w0 max_f |Q_l|, |zbar_l| <= w0 max_f |hbar_l| +
    // max_f |e_l| (the last term left per layer and column by the adjoint forward sweep, SweepArgs::ebound).  The scaled
    // column stays below 2^15; a bound that is loose by 2^m costs m of the 16 bits by which fp16's subnormal floor sits
    // below fp32's half ulp of the column maximum.  `unscale` (per lane = per column) turns the accumulators of the matrix
    // that consumed the scaled operand back into true values: 2^-k_j / sb.
    // Hessian quads: the forward sweep's tangent channels hdot^k = w0 c zdot^k are not bounded by 1 either, and the other
    // sweeps' tails couple the four channels of a quad — their bounds (set_scale) take two more per-column maxima: max_f
    // |zdot_l| of the tangent columns, left per layer by the quads' forward sweep (SweepArgs::zbound), and max_f |e_l|.
    constexpr bool kColScale = SP != 0 && (BS != SWEEP_FWD || HS);
    constexpr bool kTrackE = SP != 0 && BS == SWEEP_ADJ_FWD;   // (the fp16x3 adjoint reverse sweep therefore needs the fp16x3 adjoint forward one)
    float sb = 1.f, inv_sb = 1.f;                       // scale of the B operand being built (tails of `prev`) and its inverse
    auto colmax = [&](const f32x4 (&t)[G::NT]) -> float {      // max |.| over the 16 tiles' registers and the 4 lane quarters: per column
        float m = 0.f;
        // dudf_track is inline asm, and these are MFMA results: hipcc's hazard recogniser does not see an asm statement's
        // register reads, so the wait states between the last MFMA and the first read are spelled out (found the hard way:
        // the last layer's column scale came from stale accumulators)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int T = 0; T < G::NT; ++T) dudf_track(m, t[T]);
        m = fmaxf(m, __shfl_xor(m, 16));
        return fmaxf(m, __shfl_xor(m, 32));
    };
    // amax_true: max_f |accumulator| of this column (true values); extra: the bound of what the tail adds (adjoint reverse:
    // e_l); zb: max_f |zdot_l| of this column (quads, tangent channels).  Quads (tails in dudf_sweep_common.h::epilogue):
    //   forward          value: |sin| <= 1                  tangent k: w0 |zdot^k|
    //   reverse          value: w0 |a|                       tangent k: w0 (|adot^k| + w0 |zdot^k| |a|)           (a of the value channel)
    //   adjoint forward  value: w0 (|Q| + w0 sum_k |zdot^k| |Qdot^k|)                tangent k: w0 |Qdot^k|
    //   adjoint reverse  value: |E| + w0 |hbar| + w0^2 sum_k |zdot^k| |hdotbar^k|   tangent k: |E| + w0 |hdotbar^k|
    auto set_scale = [&](float amax_true, float extra, float zb) {
        float bound;
        if constexpr (!HS) { (void)zb; bound = a.w0 * amax_true + extra; }
        else if constexpr (is_jet(SW)) {
            // third-order jets: the tail of a monomial's column multiplies lower-order columns of the same point (epilogue,
            // SWEEP_FWD_J) — the same combination, evaluated on the columns' maxima, bounds it (|sin|, |cos| <= 1)
            const int l0 = lane & 48;
            const unsigned jw = kJetLane[lane & 15];
            const float w2 = 0.5f * (float)((jw >> 24) & 3), w3 = 0.5f * (float)((jw >> 26) & 3);
            const float F[3] = {a.w0 * __shfl(amax_true, l0 + 1), a.w0 * __shfl(amax_true, l0 + 2), a.w0 * __shfl(amax_true, l0 + 3)};
            auto sel = [&](unsigned k) -> float { k &= 3; return k == 0 ? F[0] : (k == 1 ? F[1] : F[2]); };
            const float S0 = a.w0 * __shfl(amax_true, l0 + (int)(jw & 15)), S1 = a.w0 * __shfl(amax_true, l0 + (int)((jw >> 4) & 15)),
                        S2 = a.w0 * __shfl(amax_true, l0 + (int)((jw >> 8) & 15));
            const float fab = sel(jw >> 18) * sel(jw >> 20);
            const float p2 = S0 * sel(jw >> 12) + S1 * sel(jw >> 14) + S2 * sel(jw >> 16) + w2 * fab;
            const float p3 = w3 * fab * sel(jw >> 22);
            bound = ((jw >> 28) & 1) ? 1.f : a.w0 * amax_true + p3 + p2;
        }
        else if constexpr (BS == SWEEP_FWD) bound = isv ? 1.f : a.w0 * amax_true;
        else if constexpr (BS == SWEEP_REV) {
            const float av = quad_bcast0(amax_true);
            bound = a.w0 * (amax_true + (isv ? 0.f : a.w0 * zb * av));
        } else {
            const float sk = quad_sum(isv ? 0.f : zb * amax_true);
            bound = extra + a.w0 * (amax_true + (isv ? a.w0 * sk : 0.f));
        }
        // 2^-10 of head room: the bound is formed in rounded fp32 and from the UNROUNDED e_l / zdot_l maxima, while the tails round
        // on their own path (and R, E come back rounded to 16 significant bits): a column whose bound sits an ulp under a power of two
        // could hand the fixed-point stash a value a grid step outside [-1, 1] 2^E, which fx24_pack would wrap to +5 2^E.
        // Once per layer and column, not per value; costs a bit of the column's precision in 0.14 % of the scale decisions.
        bound *= DUDF_FX_HEADROOM;
        unsigned E = (__float_as_uint(bound) >> 23) & 255u;    // bound < 2^(E - 126)
        E = E < 27u ? 27u : (E > 250u ? 250u : E);             // all-zero (padding) columns, infinities: any finite scale will do
        sb = __uint_as_float((268u - E) << 23);                // 2^(15 - (E - 126))
        inv_sb = __uint_as_float((E - 14u) << 23);
    };
    auto ebound_of = [&](int layer) -> float {                 // adjoint reverse sweep with df/dx terms: max_f |e_layer| of this column
        if constexpr (kColScale && BS == SWEEP_ADJ_REV && (HS || (FL & 1) != 0)) return a.ebound[(int64_t)layer * a.np + p];
        return 0.f;
    };
    auto zbound_of = [&](int layer) -> float {                 // quads behind the forward sweep: max_f |zdot_layer| of this column
        if constexpr (kColScale && HS && BS != SWEEP_FWD) return a.zbound[(int64_t)layer * a.nch + p];
        return 0.f;
    };
    auto store_zbound = [&](int layer, float m) {              // the quads' forward sweep leaves it (0 in the value columns)
        if constexpr (kColScale && HS && !is_jet(SW) && BS == SWEEP_FWD) a.zbound[(int64_t)layer * a.nch + p] = isv ? 0.f : m;   // (every lane, no branch: see store_fx)
    };
    float eb_next = 0.f, zb_next = 0.f;
    // running max |.| of what this sweep's tails store for the weight-gradient GEMM (q_l, A_l or zbar_l), per layer: lanes
    // -> one LDS word per layer (ds_max_u32 on the bit patterns: non-negative floats order like integers) -> HBM at kernel end
    constexpr int kRow = amax_row<SW, FL>();
    constexpr bool kFx = (P24 & 1) != 0;                // S, Q, A, Z as 24-bit fixed point relative to the column's bound (dudf_sweep_common.h)
    TailTrackT<kFx> tmax;
    // the power of two the readers of this sweep's fixed-point array multiply with, per layer and column (SweepArgs::fxs): 2^E of
    // set_scale, or 1 where no column scale exists (plain forward sweep: |sin| <= 1)
    // A fixed-point value cannot be a NaN or an infinity (their bit patterns decode to finite numbers), so the COLUMN SCALE carries
    // them: a column whose inputs are not finite — the network's output y (reverse sweep; the forward sweep checks its own y at
    // the end of the pass), the loss cotangents ybar / gbar (adjoint sweeps: a wrong `n_on_surface` hint arrives as NaN cotangents)
    // — stores NaN as its 2^E, and every reader's (t - 3) 2^E is NaN: d(theta) of a poisoned batch is NaN, as autograd's would be,
    // not finite garbage (tests/test_stash_formats_gpu.py, tests/test_api_gpu.py::test_on_surface_count_hint).
    bool poison = false;                                // (a divergent bool: one lane-mask SGPR pair, no vector register)
    auto nonfinite = [](float v) -> bool { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; };
    auto store_fx = [&](int layer) {
        if constexpr (kFx) {
            if constexpr (kColScale) tmax.fs = sb * 0x1p-15f;
#if !(DUDF_FX_DBG & 1)
            // (the plain columns' forward sweep has no column scale: its array is written once, at the end of the pass)
            // (every lane stores — the four lane quarters of a column the same number to the same address: a store under `if (q == 0)`
            //  splits the last k-block step of every layer into several basic blocks, and the reverse sweeps, which had no such
            //  store before, lost 12-18 % to it: profiles/r05_e_ab.txt)
            if constexpr (kColScale) a.fxs[(int64_t)layer * a.np + p] = poison ? __uint_as_float(0x7fc00000u) : inv_sb * 0x1p15f;
#endif
        }
    };
    unsigned* lds_amax = reinterpret_cast<unsigned*>(lds + amax_lds_off<H, SW, SP>(a));
    auto publish = [&](int layer) {
#ifdef DUDF_DBG_NOPUBLISH
        if constexpr (kRow >= 0) { asm volatile("" :: "v"(tmax.t)); tmax.t = 0.f; }
#else
        // (layers beyond the table share its last word: only networks of more than kMaxAmaxLayers layers have them, and their
        //  weight-gradient GEMM does not read the table — no uniform branch in the layer-switch step either)
        if constexpr (kRow >= 0) { lds_max_wave(lds_amax + (layer < kMaxAmaxLayers ? layer : kMaxAmaxLayers - 1), tmax.t); tmax.t = 0.f; }
#endif
    };
    // layer whose tail feeds matrix j (j == nhid: the last one, feeding the output stage), 0-based
    auto in_layer = [&](int j) -> int { return kFwdDir ? j : a.L - 1 - j; };
    auto bias_ptr = [&](int layer) -> const float* {   // forward sweep: b_{layer+1}
        return a.b1s + (size_t)layer * H;              // [L][H] biases as packed (row 0 = rho b_1)
    };
    auto stash_base = [&](int layer, int T) -> int64_t {              // wave-uniform, and told so: SGPR base + lane offset
        const int64_t v = (int64_t)layer * a.stash_layer + (int64_t)(16 * T) * a.np;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const unsigned vo = (unsigned)(((int64_t)q * a.np + p) * 16);     // this lane's granule, in bytes
    const LaneOff vl(vo, (is_hess(SW) && !is_jet(SW)) ? (unsigned)(((int64_t)q * a.np + (p >> 2)) * 16) : vo,   // C: one copy per quad
                     P24 ? (unsigned)(((p >> 4) * 64 + lane) * 12) : 0u,                                          // 24-bit tile-major arrays
                     !P24 ? 0u : (is_hess(SW) && !is_jet(SW)) ? (unsigned)(((p >> 6) * 64 + 16 * q + ((p >> 2) & 15)) * 12)   // C there: column p >> 2 of the quad region
                                                              : (unsigned)(((p >> 4) * 64 + lane) * 12));
    auto load_ops = [&](int layer, int kb, TailOps& o) {
        epilogue_loads<SW, FL, P24>(a, stash_base(layer, 2 * kb), vl, o.o1a, o.o2a, o.o3a);
        epilogue_loads<SW, FL, P24>(a, stash_base(layer, 2 * kb + 1), vl, o.o1b, o.o2b, o.o3b);
        if constexpr (BS == SWEEP_FWD && SP != 0) {
            // fp16x3: every layer's bias sits in LDS behind the weight buffers (sweep_body_b) — a vector-memory instruction
            // costs its wave ~100 cycles of issue when the CU's eight waves contend (s_memtime timeline: the two bias loads
            // held the waves that multiply first for 600 cycles of every step), an LDS read 4
            const float* lb = reinterpret_cast<const float*>(lds + 3 * G::CHUNKB) + layer * H + 32 * kb + 4 * q;
            o.ba = *reinterpret_cast<const f32x4*>(lb);
            o.bb = *reinterpret_cast<const f32x4*>(lb + 16);
        } else if constexpr (BS == SWEEP_FWD) {
            o.ba = *reinterpret_cast<const f32x4*>(bias_ptr(layer) + 32 * kb + 4 * q);
            o.bb = *reinterpret_cast<const f32x4*>(bias_ptr(layer) + 32 * kb + 16 + 4 * q);
        }
    };
    // tail of tiles 2kb, 2kb+1 of `layer`: fp32 results (and the stash stores the sweep owes)
    auto run_tail = [&](int layer, int kb, const f32x4 z0, const f32x4 z1, const TailOps& o, f32x4& e0, f32x4& e1) {
        const f32x4 zero = {0, 0, 0, 0};
        if constexpr (BS == SWEEP_FWD && SP != 0 && HS) {   // quads: the bias only in the value channel; `unscale` is this column's
            const f32x4 us = {unscale, unscale, unscale, unscale};
            e0 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, __builtin_elementwise_fma(z0, us, isv ? o.ba : zero), zero, zero, zero, stash_base(layer, 2 * kb), vl, isv, tmax);
            e1 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, __builtin_elementwise_fma(z1, us, isv ? o.bb : zero), zero, zero, zero, stash_base(layer, 2 * kb + 1), vl, isv, tmax);
        } else if constexpr (BS == SWEEP_FWD && SP != 0) {  // plain columns: z = 2^-k (2^k W h) + b, one FMA per value
            // (measured and dropped, round 4: w0 2/pi folded into this FMA and the biases, sin / cos from the argument in quarter
            //  turns — two instructions per value fewer, -2 % on this sweep, value error unchanged; but the ROUNDED constant
            //  w0 2/pi is off by 2e-8, the same way for every pre-activation of the network: a coherent error that the 12-step
            //  beetle trajectory amplified to 1e-3 where the reference's own, unbiased, fp32 roundings stay at 2e-7)
            e0 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, __builtin_elementwise_fma(z0, f32x4{unscale, unscale, unscale, unscale}, o.ba), zero, zero, zero, stash_base(layer, 2 * kb), vl, isv, tmax);
            e1 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, __builtin_elementwise_fma(z1, f32x4{unscale, unscale, unscale, unscale}, o.bb), zero, zero, zero, stash_base(layer, 2 * kb + 1), vl, isv, tmax);
        } else if constexpr (BS == SWEEP_FWD) {      // z = W h + b only in the value channel
            e0 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, z0 + (isv ? o.ba : zero), zero, zero, zero, stash_base(layer, 2 * kb), vl, isv, tmax);
            e1 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, z1 + (isv ? o.bb : zero), zero, zero, zero, stash_base(layer, 2 * kb + 1), vl, isv, tmax);
        } else if constexpr (kColScale) {            // accumulators -> true values first (2^-k_j / sb of the matrix that made them)
            e0 = epilogue<SW, FL, kTrackE, P24, false, TailTrackT<kFx>>(a, z0 * unscale, o.o1a, o.o2a, o.o3a, stash_base(layer, 2 * kb), vl, isv, tmax);
            e1 = epilogue<SW, FL, kTrackE, P24, false, TailTrackT<kFx>>(a, z1 * unscale, o.o1b, o.o2b, o.o3b, stash_base(layer, 2 * kb + 1), vl, isv, tmax);
        } else {
            e0 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, z0, o.o1a, o.o2a, o.o3a, stash_base(layer, 2 * kb), vl, isv, tmax);
            e1 = epilogue<SW, FL, false, P24, false, TailTrackT<kFx>>(a, z1, o.o1b, o.o2b, o.o3b, stash_base(layer, 2 * kb + 1), vl, isv, tmax);
        }
    };
    auto pin_ops = [&](TailOps& o) {                    // make the compiler wait for these loads HERE
        if constexpr (BS == SWEEP_FWD) asm volatile("" : "+v"(o.ba), "+v"(o.bb));
        else asm volatile("" : "+v"(o.o1a), "+v"(o.o2a), "+v"(o.o1b), "+v"(o.o2b));
        if constexpr (SW == SWEEP_ADJ_FWD_H || SW == SWEEP_ADJ_REV_H) asm volatile("" : "+v"(o.o3a), "+v"(o.o3b));
    };

    // chunk stream: chunk c = (matrix c / NKB, k-block c % NKB) lives in LDS buffer c % 3 and is fetched TWO steps
    // ahead.  vmcnt retires in order, so a one-step-ahead DMA would force every stash load and store of the previous
    // step to complete within one step as well; with two steps everything gets two (~3 us, an HBM round trip under load).
    const int total = nhid * G::NKB;
    auto chunk_src = [&](int c) -> const char* {       // c is wave-uniform
        const int j = c / G::NKB;
        return image(j) + (size_t)(c - j * G::NKB) * G::CHUNKB;
    };
    __syncthreads();                                   // every wave is past its last LDS read of the previous tile
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;   // LDS byte offset of the buffers
    const unsigned voff = (unsigned)lane * 16u;
    // Waves 0-3 issue ALL the DMA pieces of a chunk (their own six and their SIMD partner's), waves 4-7 none: a
    // vector-memory instruction stalls its wave's in-order issue for ~100 cycles when the CU's eight waves contend, and the
    // waves that multiply first (4-7, see `late` below) are the ones the step waits for — their chain loses the six pieces,
    // the other half, which idled at the barrier, takes them (-0.9 % on the step; the reverse assignment: no gain).
    // A partial pass with at most four active waves (the last pass of a workgroup's share: at 100 000 points a single wave in
    // 112 workgroups) leaves waves 4-7 idle: they take all the pieces then, the active waves none.
    auto dma2 = [&](const char* src, unsigned dst) {
        if (wave < NWB / 2) {
            const int w = wave & (NWB / 2 - 1);
            dma_issue<H, SP>(src, dst, voff, w); dma_issue<H, SP>(src, dst, voff, w + NWB / 2);
        }
    };
    dma2(chunk_src(0), lds0 + gc * G::CHUNKB);
    dma2(chunk_src(1), lds0 + ((gc + 1) % 3) * G::CHUNKB);
    if (wave >= nact) {                                // same DMA pieces, same barriers, nothing else
        dma_wait_b<0>();
        __syncthreads();
        for (int c = 0; c < total; ++c) {
            const bool more = c + 2 < total;
            if (more) dma2(chunk_src(c + 2), lds0 + ((gc + 2) % 3) * G::CHUNKB);
            gc = (gc + 1) % 3;
            if (more) dma_wait_b<2 * G::NDMA>();       // only this step's pieces may still be in flight (waves 4-7: nothing)
            else dma_wait_b<0>();
            __syncthreads();
        }
        return;
    }

    // ------------------------------ first layer (fp32, K = 3): pre-activations / incoming adjoints of 16 tiles -------
    {
        float b = 0.f, yb = 1.f;
        if constexpr (BS == SWEEP_FWD) b = (q < 3) ? a.x4[p * 4 + q] : 0.f;           // bias is added by the tail
        if constexpr (BS == SWEEP_ADJ_FWD) b = (q < 3) ? a.gbar[p * 4 + q] : 0.f;
        if constexpr (BS == SWEEP_ADJ_REV) yb = a.ybar[p];
        if constexpr (SW == SWEEP_REV_H) yb = isv ? 1.f : 0.f;                         // adot_L^k = 0
        if constexpr (kFx) {                                                           // poisoned column? (store_fx)
            if constexpr (BS == SWEEP_REV) poison = nonfinite(a.y[p]);
            if constexpr (BS == SWEEP_ADJ_REV) poison = nonfinite(a.ybar[p]);
            if constexpr (BS == SWEEP_ADJ_FWD) {
                int bad = nonfinite(b) ? 1 : 0;
                bad |= __shfl_xor(bad, 16); bad |= __shfl_xor(bad, 32);
                poison = bad != 0;
            }
        }
#pragma unroll
        for (int T = 0; T < G::NT; ++T) {
            if constexpr (kFwdDir) prev[T] = mfma16(a.w1b[(16 * T + li) * 4 + q], b, f32x4{0, 0, 0, 0});
            else prev[T] = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
        }
    }

    // ------------------------------ hidden x hidden layers ------------------------------
    // Step c:  [operands of tail c+1 have landed] -> DMA of chunk c+2 -> operand loads of tail c+3 -> 16 tiles x 6
    // MFMAs with B(c), interleaved with tail c+1 -> B(c+1) -> wait for chunk c+1 + barrier.  Tail t = pair t % NKB of
    // the layer feeding matrix t / NKB; the last step of a layer runs the first tail of the next one once its own
    // accumulators are final.
    // Tail operands: plain columns keep two sets (the next tail's, pinned at the top of the step, and the one after it,
    // loaded right behind the DMA: a full step ahead); the quad variants have a third operand array and no registers
    // for that — one set, refilled right after the tail that consumed it (about 0.8 step ahead).
    // (the fp16x3 adjoint forward sweep — two loads, two stores and the column scales — fits its registers with one set only)
    constexpr bool kOneSet = HS;
    TailOps ops_cur, ops_n1;
    u32x4 bp[NPC];                                     // B operand of the current step: bf16 h | m | l, or fp16 hi | lo
    auto split = [&](const f32x4 e0, const f32x4 e1, u32x4 (&o)[NPC]) {
        if constexpr (kColScale) split8h(e0 * sb, e1 * sb, o[0], o[1]);
        else if constexpr (SP) split8h(e0, e1, o[0], o[1]);
        else split8(e0, e1, o[0], o[1], o[2]);
    };
    // per layer and column max_f |e_l|: the adjoint forward sweep leaves it for the adjoint reverse sweep's column scale
    auto store_ebound = [&](int layer) {
        if constexpr (kTrackE) {                         // (every lane stores, no branch: see store_fx; the training launches always pass ebound)
            float m = fmaxf(tmax.e, __shfl_xor(tmax.e, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            a.ebound[(int64_t)layer * a.np + p] = m;
            tmax.e = 0.f;
        }
    };
    f32x4 fin0 = {0, 0, 0, 0}, fin1 = {0, 0, 0, 0};    // fp32 results of pair 0 of the layer after the last matrix
    load_ops(in_layer(0), 0, ops_cur);
    if constexpr (!kOneSet) load_ops(in_layer(0), 1, ops_n1);
    if constexpr (kColScale) {                         // first layer (fp32 MFMA / W_out^T ybar: true values): its column scale
        const float eb0 = ebound_of(in_layer(0)), zb0 = zbound_of(in_layer(0));
        if (nhid > 0) { eb_next = ebound_of(in_layer(1)); zb_next = zbound_of(in_layer(1)); }
        const float cm = colmax(prev);
        store_zbound(in_layer(0), cm);
        set_scale(cm, eb0, zb0);
    }
    store_fx(in_layer(0));
    {
        f32x4 e0, e1;
        run_tail(in_layer(0), 0, prev[0], prev[1], ops_cur, e0, e1);
        split(e0, e1, bp);
        if constexpr (kOneSet) load_ops(in_layer(0), 1, ops_cur);
    }
#pragma unroll
    for (int T = 0; T < G::NT; ++T) acc[T] = f32x4{0, 0, 0, 0};
    dma_wait_b<0>();                                   // once per tile: chunks 0 and 1 and everything above
    __syncthreads();
    constexpr int kYoung = younger_ops<SW, FL>() - ((SP != 0 && BS == SWEEP_FWD) ? 2 : 0);   // fp16x3 forward: the bias comes from LDS
    // The two waves of a SIMD (w, w + 4) leave every k-block barrier together.  With the same program order both run
    // their tail (vector ALU) at the same time and then collide on the matrix pipe: the step costs tail + MFMAs of both.
    // Waves 4-7 therefore run their MFMAs FIRST and the tail of the next step behind them (the tail only needs the
    // previous layer's accumulators, not this step's): each half's tail falls beside the other half's MFMAs.
    // (also for the Hessian-quad variants that keep their registers: the adjoint-forward quads and the jets would spill)
    constexpr bool kLateOk = !HS || (SW != SWEEP_ADJ_FWD_H && SW != SWEEP_FWD_J);
    #ifdef DUDF_LATE_FORCE                                 // tests/isa_contract.py: one half's program order at a time, branch-free
    const bool late = kLateOk && DUDF_LATE_FORCE;
#else
    const bool late = kLateOk && __builtin_amdgcn_readfirstlane((int)(wave >= NWB / 2)) != 0;
#endif
    // feed slot: after which tile's MFMAs a wave issues its DMA pieces and operand loads (-1: at the top of the step);
    // tail slot: after which tile's MFMAs it runs the tail of the next step.  A = waves 0-3, B = waves 4-7 (when `late`).
    constexpr int FA = -1, TA = 0, FB = (BS == SWEEP_FWD ? 15 : 7) * (G::NT - 1) / 15, TB = G::NT - 1;
    for (int j = 0; j < nhid; ++j) {
        const int lin = in_layer(j), lnx = in_layer(j + 1);
#pragma unroll
        for (int kb = 0; kb < G::NKB; ++kb) {
            const int c = j * G::NKB + kb;
            const char* bpl = lds + gc * G::CHUNKB + lane * 16;
            auto frag = [&](int T, int pc) -> u32x4 {
#if DUDF_SWEEP_DBG & 32
                u32x4 z; asm volatile("" : "=v"(z)); return z;
#endif
                return *reinterpret_cast<const u32x4*>(bpl + (T * NPC + pc) * G::FRAG);
            };
            DUDF_STAMP(0);
            if constexpr (!kOneSet) ops_cur = ops_n1;
            pin_ops(ops_cur);
            const bool more = c + 2 < total;
            auto load_after_next = [&](TailOps& o) {                 // operands of tail c+2
                if (kb + 2 < G::NKB) load_ops(lin, kb + 2, o);
                else load_ops(lnx, kb + 2 - G::NKB, o);
            };
            auto feed = [&]() {                                      // this step's DMA pieces, then the operand loads
                if (more) dma2(chunk_src(c + 2), lds0 + ((gc + 2) % 3) * G::CHUNKB);
                if constexpr (!kOneSet) load_after_next(ops_n1);          // one full step ahead
            };
            u32x4 nb[NPC];
            // fp16x3: the half that does not multiply first runs its tail at the very top of the step, in front of its DMA
            // pieces (fetched two steps ahead: no hurry) and of its first LDS fragment: the step is (tail of one half beside the
            // MFMAs of the other) twice, and neither tail should wait for anything
            constexpr bool kTailTop = DUDF_TAIL_TOP;
            if (kTailTop && !late && kb + 1 < G::NKB) {
                DUDF_STAMP(2);
                f32x4 e0, e1;
                run_tail(lin, kb + 1, prev[2 * kb + 2], prev[2 * kb + 3], ops_cur, e0, e1);
                split(e0, e1, nb);
                if constexpr (kOneSet) load_after_next(ops_cur);
                DUDF_STAMP(3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((FA < 0 && !late) || (FB < 0 && late)) feed();
            DUDF_STAMP(1);
            __builtin_amdgcn_sched_barrier(0);
            // A fragments travel two tiles (12 MFMAs, ~190 cycles) ahead of their use: with both waves of a SIMD and the
            // chunk DMA on the LDS, one tile of distance does not cover the read latency
            // (fp16x3: FOUR tiles = 12 MFMAs: a tile is only three MFMAs long)
            constexpr int AD = (SP && BS == SWEEP_FWD) ? 4 : 2;   // (the other fp16x3 sweeps need the registers: two tiles)
            u32x4 an[AD][NPC];
#pragma unroll
            for (int T = 0; T < AD; ++T)
#pragma unroll
                for (int pc = 0; pc < NPC; ++pc) an[T][pc] = frag(T, pc);
            // TP tiles per trip, their MFMA chains interleaved: a tile's products accumulate into ONE register quad, and a
            // wave that issues them back to back waits out each MFMA's latency (fp16x3: three-instruction chains — alone on
            // the pipe a wave reached one MFMA per ~34 cycles instead of 16)
            constexpr int TP = SP ? 2 : 1;
#pragma unroll
            for (int T = 0; T < G::NT; T += TP) {
                u32x4 af[TP][NPC];
#pragma unroll
                for (int u = 0; u < TP; ++u)
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) af[u][pc] = an[(T + u) % AD][pc];
                if (T + AD < G::NT) {
#pragma unroll
                    for (int u = 0; u < TP; ++u)
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) an[(T + u) % AD][pc] = frag(T + u + AD, pc);
                    // vector/scalar ALU and vector memory may move across, LDS reads and MFMAs may not: otherwise hipcc
                    // floats each tile's MFMAs up to its reads and every fragment is waited for just in time
                    __builtin_amdgcn_sched_barrier(0x76);
                }
                f32x4 cc[TP];
#pragma unroll
                for (int u = 0; u < TP; ++u) cc[u] = acc[T + u];
                if constexpr (SP) {                                     // smallest terms first: lo*hi, hi*lo, hi*hi
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_h(as_h(af[u][1]), as_h(bp[0]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_h(as_h(af[u][0]), as_h(bp[1]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_h(as_h(af[u][0]), as_h(bp[0]), cc[u]);
                } else {
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][1]), as_bf(bp[1]), cc[u]);   // smallest terms first
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][2]), as_bf(bp[0]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][0]), as_bf(bp[2]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][1]), as_bf(bp[0]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][0]), as_bf(bp[1]), cc[u]);
#pragma unroll
                    for (int u = 0; u < TP; ++u) cc[u] = mfma_b(as_bf(af[u][0]), as_bf(bp[0]), cc[u]);
                }
#pragma unroll
                for (int u = 0; u < TP; ++u) acc[T + u] = cc[u];
                const int Tl = T + TP - 1;                              // last tile of this trip
                if ((FA >= T && FA <= Tl && !late) || (FB >= T && FB <= Tl && late)) {   // the late half issues its DMA pieces (60-180 cycles
                    __builtin_amdgcn_sched_barrier(0);                  // of issue each) under the other half's MFMAs
                    DUDF_STAMP(6);
                    feed();
                    DUDF_STAMP(7);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (kb + 1 < G::NKB && ((TA >= T && TA <= Tl && !late && !kTailTop) || (TB >= T && TB <= Tl && late))) {   // the next step's B operand
                    if (T != 0) __builtin_amdgcn_sched_barrier(0);
                    DUDF_STAMP(2);
                    f32x4 e0, e1;
                    run_tail(lin, kb + 1, prev[2 * kb + 2], prev[2 * kb + 3], ops_cur, e0, e1);
                    split(e0, e1, nb);
                    if constexpr (kOneSet) load_after_next(ops_cur);
                    DUDF_STAMP(3);
                }
            }
            if (kb + 1 == G::NKB) {                                     // layer done: first tail of the next one
                if constexpr (kColScale) {
                    // this matrix consumed the operand scaled by sb: its accumulators are 2^k_j sb x the true values
                    unscale = unscale_of(j) * inv_sb;
                    const float ebl = eb_next, zbl = zb_next;
                    // a layer ahead: its latency hides behind 8 steps.  (No branch at the end of the stream: the last switches re-read the
                    // last layer's entry — a uniform branch would end the basic block of this step as well, see store_fx)
                    { const int jn = j + 2 <= nhid ? j + 2 : nhid; eb_next = ebound_of(in_layer(jn)); zb_next = zbound_of(in_layer(jn)); }
                    const float cm = colmax(acc) * unscale;
                    store_zbound(lnx, cm);
                    set_scale(cm, ebl, zbl);                            // ... and the scale of the operand its tails are about to build
                } else if constexpr (SP) unscale = unscale_of(j);
                store_fx(lnx);
                publish(lin);                                           // every tail of layer `lin` has run
                store_ebound(lin);
                run_tail(lnx, 0, acc[0], acc[1], ops_cur, fin0, fin1);
                split(fin0, fin1, nb);
                if constexpr (kOneSet) load_after_next(ops_cur);
#pragma unroll
                for (int T = 0; T < G::NT; ++T) { prev[T] = acc[T]; acc[T] = f32x4{0, 0, 0, 0}; }
            }
#pragma unroll
            for (int pc = 0; pc < NPC; ++pc) bp[pc] = nb[pc];
            gc = (gc + 1) % 3;
            DUDF_STAMP(4);
#if !(DUDF_SWEEP_DBG & 8)
            // chunk c+1 landed; c+2 and two steps' stash traffic stay in flight.  Only the waves that issued DMA pieces wait
            // (0-3: the barrier behind publishes the chunk to the others)
            // (tail at the top of the step: its stores are OLDER than this step's pieces — one step of stash traffic less)
            if (more) { if (wave < NWB / 2) dma_wait_b<(kTailTop ? 1 : 2) * kYoung + 2 * G::NDMA>(); }
            else dma_wait_b<0>();
#endif
            DUDF_STAMP(5);
#if !(DUDF_SWEEP_DBG & 16)
            __syncthreads();
#endif
        }
    }

    // ------------------------------ remaining tails of the last hidden layer + output stage (fp32) -------------------
    {
        const int lin = in_layer(nhid);
        float part = 0.f;
        f32x4 accg = {0, 0, 0, 0};
#pragma unroll
        for (int kb = 0; kb < G::NKB; ++kb) {
            f32x4 e0 = fin0, e1 = fin1;
            if (kb > 0) {
                if constexpr (!kOneSet) {
                    ops_cur = ops_n1;
                    if (kb + 1 < G::NKB) load_ops(lin, kb + 1, ops_n1);
                }
                run_tail(lin, kb, prev[2 * kb], prev[2 * kb + 1], ops_cur, e0, e1);
                if constexpr (kOneSet) { if (kb + 1 < G::NKB) load_ops(lin, kb + 1, ops_cur); }
            }
            if constexpr (BS == SWEEP_FWD) {
                const f32x4 w0v = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 32 * kb + 4 * q);
                const f32x4 w1v = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 32 * kb + 16 + 4 * q);
                part += e0[0] * w0v[0] + e0[1] * w0v[1] + e0[2] * w0v[2] + e0[3] * w0v[3];
                part += e1[0] * w1v[0] + e1[1] * w1v[1] + e1[2] * w1v[2] + e1[3] * w1v[3];
            } else if constexpr (BS == SWEEP_REV) {
                const f32x4 w0v = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 32 * kb + 4 * q);
                const f32x4 w1v = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 32 * kb + 16 + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) accg = mfma16(w0v[t], e0[t], accg);
#pragma unroll
                for (int t = 0; t < 4; ++t) accg = mfma16(w1v[t], e1[t], accg);
            }
        }
        publish(lin);
        store_ebound(lin);
        if constexpr (BS == SWEEP_FWD) {
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            if (isv) part += a.theta[a.off_bo];         // tangent columns are derivatives: no constant term
            if (q == 0) a.y[p] = part;
            if constexpr (kFx) {                        // a column whose output is not finite: its h_l scales become NaN (store_fx)
                if constexpr (kColScale) {
                    if (q == 0 && nonfinite(part))
                        for (int l = 0; l < a.L; ++l) a.fxs[(int64_t)l * a.np + p] = __uint_as_float(0x7fc00000u);
                } else if (q == 0) {                    // plain columns: |sin| <= 1, the scale is 1 in every layer
                    const float sc = nonfinite(part) ? __uint_as_float(0x7fc00000u) : 1.f;
                    for (int l = 0; l < a.L; ++l) a.fxs[(int64_t)l * a.np + p] = sc;
                }
            }
        } else if constexpr (BS == SWEEP_REV) {
            if constexpr (kFx) { if (poison) accg = f32x4{__uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), 0.f}; }   // df/dx of a poisoned column (its cos came back as a finite number)
            if (q == 0) *reinterpret_cast<f32x4*>(a.g + p * 4) = f32x4{accg[0], accg[1], accg[2], 0.f};
        }
    }
}

// ---- "oct mode": a pass that holds ONE 16-column group (the last pass of a workgroup's share: at 100 000 points 112 of the 256
// workgroups end with one) ---------------------------------------------------------------------------------------------------
// With sweep_tile_b a lone wave does everything serially: 8 tails and 8 x 48 MFMAs per layer, nothing to overlap them with
// (measured in situ: 0.19 ms of a 3.2 ms step; alone, a lone-wave pass takes 0.68 of a full one).  Here all eight waves share
// the group: wave w owns output tiles 2w, 2w+1 — exactly the pair whose post-tail values are the B operand of k-block w of
// the next matrix.  Per layer: [column scale: max over the waves' 32 features each, through LDS] -> every wave runs ONE
// tail pair (in parallel: an eighth of the lone wave's tail work) and publishes its B fragment (hi | lo, 2 KiB) in LDS ->
// 8 k-blocks of 6 MFMAs per wave, B(kb) read from LDS.  fp16x3, plain columns, H = 256.  No weight chunks through LDS here: a
// wave needs only ITS two tiles' fragments (4 KiB of a 32 KiB chunk), so it loads them from L2 straight into registers — all
// 32 fragments of a layer at the top of the layer, behind the exchange and the tail (a step of 6 MFMAs is far shorter than
// an LDS-DMA round trip: with the chunk stream the pass was bound by DMA latency).  Two barriers per layer, no hand-counted
// waits; the chunk buffers and `gc` are left alone.
constexpr int kOctBytes = 8 * 2 * 1024 + 1024 + 8 * 64 * 16;      // B fragments | column maxima (2 x 128 floats) | output-stage partials
template <int H, int SW, int FL, int P24 = 0>
__device__ __forceinline__ void sweep_tile_oct(const SweepArgs& a, const int g, char* lds, unsigned& gc, const unsigned oct_off) {
    using G = GeoB<H, 1>;
    static_assert(H == 256 && SW <= SWEEP_ADJ_REV, "fp16x3, plain columns, one k-block per wave");
    constexpr int NPC = 2;
    constexpr int BS = base_of(SW);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int nhid = a.L - 1;
    constexpr bool kFwdDir = (BS == SWEEP_FWD || BS == SWEEP_ADJ_FWD);
    constexpr bool kColScale = BS != SWEEP_FWD;
    constexpr bool kTrackE = BS == SWEEP_ADJ_FWD;
    constexpr int kRow = amax_row<SW, FL>();
    const int64_t p = (int64_t)g * 16 + li;
    const int T0 = 2 * wave;
    auto image = [&](int j) -> const char* {
        return kFwdDir ? a.wimg16_f + (size_t)j * G::IMGB : a.wimg16_t + (size_t)(nhid - 1 - j) * G::IMGB;
    };
    auto unscale_of = [&](int j) -> float { return a.wsc[kFwdDir ? j : nhid - 1 - j]; };
    auto in_layer = [&](int j) -> int { return kFwdDir ? j : a.L - 1 - j; };
    auto stash_base = [&](int layer, int T) -> int64_t {
        const int64_t v = (int64_t)layer * a.stash_layer + (int64_t)(16 * T) * a.np;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const LaneOff vo((unsigned)(((int64_t)q * a.np + p) * 16), (unsigned)(((int64_t)q * a.np + p) * 16),
                     P24 ? (unsigned)(((p >> 4) * 64 + lane) * 12) : 0u);
    char* ox = lds + oct_off;
    u32x4* Bx = reinterpret_cast<u32x4*>(ox);                          // [k-block][piece][lane]
    float* cmx = reinterpret_cast<float*>(ox + 8 * NPC * 1024);        // [wave][column]: max |accumulator|
    float* emx = cmx + 128;                                            // [wave][column]: max |e_l| (adjoint forward sweep)
    f32x4* red = reinterpret_cast<f32x4*>(ox + 8 * NPC * 1024 + 1024); // [wave][lane]: output-stage partial sums
    unsigned* lds_amax = reinterpret_cast<unsigned*>(lds + 3 * G::CHUNKB);

    (void)gc;
    __syncthreads();                                   // every wave is past its last LDS read of the previous pass

    // first layer (fp32, K = 3): my two tiles
    f32x4 prev[2], acc[2];
    bool poison = false;
    auto nonfinite = [](float v) -> bool { return (__float_as_uint(v) & 0x7f800000u) == 0x7f800000u; };
    {
        float b = 0.f, yb = 1.f;
        if constexpr (BS == SWEEP_FWD) b = (q < 3) ? a.x4[p * 4 + q] : 0.f;
        if constexpr (BS == SWEEP_ADJ_FWD) b = (q < 3) ? a.gbar[p * 4 + q] : 0.f;
        if constexpr (BS == SWEEP_ADJ_REV) yb = a.ybar[p];
        if constexpr ((P24 & 1) != 0) {                 // poisoned column? (sweep_tile_b: store_fx)
            if constexpr (BS == SWEEP_REV) poison = nonfinite(a.y[p]);
            if constexpr (BS == SWEEP_ADJ_REV) poison = nonfinite(yb);
            if constexpr (BS == SWEEP_ADJ_FWD) {
                int bad = nonfinite(b) ? 1 : 0;
                bad |= __shfl_xor(bad, 16); bad |= __shfl_xor(bad, 32);
                poison = bad != 0;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int T = T0 + u;
            if constexpr (kFwdDir) prev[u] = mfma16(a.w1b[(16 * T + li) * 4 + q], b, f32x4{0, 0, 0, 0});
            else prev[u] = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
        }
    }
    f32x4 o1[2], o2[2], o3[2], bs[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    auto load_ops = [&](int layer) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            epilogue_loads<SW, FL, P24>(a, stash_base(layer, T0 + u), vo, o1[u], o2[u], o3[u]);
            if constexpr (BS == SWEEP_FWD)
                bs[u] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(lds + 3 * G::CHUNKB) + layer * H + 16 * (T0 + u) + 4 * q);
        }
    };
    auto ebound_of = [&](int layer) -> float {
        if constexpr (BS == SWEEP_ADJ_REV && (FL & 1) != 0) return a.ebound[(int64_t)layer * a.np + p];
        return 0.f;
    };
    load_ops(in_layer(0));
    float eb = ebound_of(in_layer(0));
    float unscale = 1.f, sb = 1.f, inv_sb = 1.f;
    constexpr bool kFx = (P24 & 1) != 0;
    TailTrackT<kFx> tk;
    f32x4 e[2];
    for (int j = 0; j <= nhid; ++j) {
        const int lin = in_layer(j);
        if constexpr (kColScale) {                      // the column's scale from ALL 16 tiles: max over the eight waves
            float m = 0.f;
            __builtin_amdgcn_sched_barrier(0);          // MFMA results read by inline asm: explicit wait states (see colmax)
            asm volatile("s_nop 15\n\ts_nop 15");
            __builtin_amdgcn_sched_barrier(0);
            dudf_track(m, prev[0]); dudf_track(m, prev[1]);
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (q == 0) cmx[wave * 16 + li] = m;
            __syncthreads();
            float mm = 0.f;
#pragma unroll
            for (int w = 0; w < NWB; ++w) mm = fmaxf(mm, cmx[w * 16 + li]);
            const float bound = (a.w0 * (mm * unscale) + eb) * DUDF_FX_HEADROOM;   // (2^-10 of head room: see set_scale)
            unsigned E = (__float_as_uint(bound) >> 23) & 255u;
            E = E < 27u ? 27u : (E > 250u ? 250u : E);
            sb = __uint_as_float((268u - E) << 23);
            inv_sb = __uint_as_float((E - 14u) << 23);
        }
        if constexpr (kFx) {                            // fixed-point stash: this layer's column scale for the readers
            if constexpr (kColScale) tk.fs = sb * 0x1p-15f;
            if (wave == 0 && q == 0) a.fxs[(int64_t)lin * a.np + p] = poison ? __uint_as_float(0x7fc00000u) : (kColScale ? inv_sb * 0x1p15f : 1.f);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {                   // ONE tail pair per wave and layer
            f32x4 z = prev[u];
            if constexpr (BS == SWEEP_FWD) z = __builtin_elementwise_fma(z, f32x4{unscale, unscale, unscale, unscale}, bs[u]);
            else z *= unscale;
            e[u] = epilogue<SW, FL, kTrackE, P24, false, TailTrackT<kFx>>(a, z, o1[u], o2[u], o3[u], stash_base(lin, T0 + u), vo, true, tk);
        }
        if constexpr (kRow >= 0) { if (lin < kMaxAmaxLayers) lds_max_wave(lds_amax + lin, tk.t); tk.t = 0.f; }
        if constexpr (kTrackE) {
            float m = fmaxf(tk.e, __shfl_xor(tk.e, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            if (q == 0) emx[wave * 16 + li] = m;
            tk.e = 0.f;
        }
        if (j == nhid) break;
        // this wave's A fragments of matrix j: [k-block][tile][piece], 16 B per lane each
        u32x4 fr[G::NKB][2][NPC];
        {
            const char* img = image(j) + (size_t)lane * 16;
#pragma unroll
            for (int kb = 0; kb < G::NKB; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc)
                        fr[kb][u][pc] = *reinterpret_cast<const u32x4*>(img + (size_t)kb * G::CHUNKB + ((T0 + u) * NPC + pc) * G::FRAG);
        }
        u32x4 bh, bl;
        if constexpr (kColScale) split8h(e[0] * sb, e[1] * sb, bh, bl);
        else split8h(e[0], e[1], bh, bl);
        Bx[(wave * NPC + 0) * 64 + lane] = bh;          // = the B operand of k-block `wave` of matrix j
        Bx[(wave * NPC + 1) * 64 + lane] = bl;
        load_ops(in_layer(j + 1));                      // the next layer's tail operands: a whole k-loop ahead
        eb = ebound_of(in_layer(j + 1));
        __syncthreads();
        if constexpr (kTrackE) {                        // per layer and column max_f |e_l| for the adjoint reverse sweep
            if (wave == 0 && q == 0 && a.ebound) {
                float m = 0.f;
#pragma unroll
                for (int w = 0; w < NWB; ++w) m = fmaxf(m, emx[w * 16 + li]);
                a.ebound[(int64_t)lin * a.np + p] = m;
            }
        }
        acc[0] = f32x4{0, 0, 0, 0}; acc[1] = acc[0];
#pragma unroll
        for (int kb = 0; kb < G::NKB; ++kb) {
            const u32x4 b0 = Bx[(kb * NPC + 0) * 64 + lane], b1 = Bx[(kb * NPC + 1) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 cc = acc[u];
                cc = mfma_h(as_h(fr[kb][u][1]), as_h(b0), cc);    // smallest terms first: lo*hi, hi*lo, hi*hi
                cc = mfma_h(as_h(fr[kb][u][0]), as_h(b1), cc);
                cc = mfma_h(as_h(fr[kb][u][0]), as_h(b0), cc);
                acc[u] = cc;
            }
        }
        __syncthreads();                                // every wave has read B: the next layer may overwrite it
        prev[0] = acc[0]; prev[1] = acc[1];
        unscale = kColScale ? unscale_of(j) * inv_sb : unscale_of(j);
    }
    // output stage: the eight waves' partial results through LDS
    if constexpr (BS == SWEEP_FWD) {
        float part = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * (T0 + u) + 4 * q);
            part += e[u][0] * wv[0] + e[u][1] * wv[1] + e[u][2] * wv[2] + e[u][3] * wv[3];
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (q == 0) cmx[wave * 16 + li] = part;
        __syncthreads();
        if (wave == 0 && q == 0) {
            float y = a.theta[a.off_bo];
#pragma unroll
            for (int w = 0; w < NWB; ++w) y += cmx[w * 16 + li];
            a.y[p] = y;
            if constexpr (kFx) {
                if (nonfinite(y))
                    for (int l = 0; l < a.L; ++l) a.fxs[(int64_t)l * a.np + p] = __uint_as_float(0x7fc00000u);
            }
        }
    } else if constexpr (BS == SWEEP_REV) {
        f32x4 accg = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 16 * (T0 + u) + 4 * q);
#pragma unroll
            for (int t = 0; t < 4; ++t) accg = mfma16(wv[t], e[u][t], accg);
        }
        red[wave * 64 + lane] = accg;
        __syncthreads();
        if (wave == 0 && q == 0) {
            f32x4 sum = {0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < NWB; ++w) sum += red[w * 64 + lane];
            if constexpr (kFx) { if (poison) sum = f32x4{__uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), __uint_as_float(0x7fc00000u), 0.f}; }
            *reinterpret_cast<f32x4*>(a.g + p * 4) = f32x4{sum[0], sum[1], sum[2], 0.f};
        }
    } else if constexpr (kTrackE) {                     // the last layer's max |e_l|
        __syncthreads();
        if (wave == 0 && q == 0 && a.ebound) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < NWB; ++w) m = fmaxf(m, emx[w * 16 + li]);
            a.ebound[(int64_t)in_layer(nhid) * a.np + p] = m;
        }
    }
}

// (bid, nblk): this workgroup's index among the `nblk` that share the columns of `a` — the whole grid, or one of the two
// parts of a pair launch (sweep_pair_kernel)
template <int H, int SW, int FL, int SP = 0, int P24 = 0>
__device__ __forceinline__ void sweep_body_b(const SweepArgs& a, const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) char lds_b[];
    unsigned gc = 0;
    const bool clk_on = a.clk != nullptr && bid == 0;   // profiling: the clock this kernel runs at
    const unsigned long long clk_t0 = clk_on ? __builtin_amdgcn_s_memtime() : 0ull, clk_r0 = clk_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int kRow = amax_row<SW, FL>();
    unsigned* lds_amax = reinterpret_cast<unsigned*>(lds_b + amax_lds_off<H, SW, SP>(a));
    if constexpr (kRow >= 0) { if (threadIdx.x < kMaxAmaxLayers) lds_amax[threadIdx.x] = 0u; }   // (sweep_tile_b starts with a barrier)
    if constexpr (SP != 0 && base_of(SW) == SWEEP_FWD) {     // b_1 .. b_L behind the three weight buffers (read by the tails)
        float* lb = reinterpret_cast<float*>(lds_b + 3 * GeoB<H, SP>::CHUNKB);
        for (int i = threadIdx.x; i < a.L * H; i += 64 * NWB) lb[i] = a.b1s[i];      // [L][H] biases as packed (row 0 = rho b_1)
    }
    // balanced shares of 16-column groups; a share is walked in passes of 8 groups, the last one possibly partial —
    // a pass with one wave per SIMD (or a single wave) costs about half a full one, a whole extra round would cost all of it
    const int ng = a.ntiles * (TILE / 16), gbase = a.tile0 * (TILE / 16);
    const int g0 = (int)((int64_t)bid * ng / nblk), g1 = (int)((int64_t)(bid + 1) * ng / nblk);
    constexpr bool kOct = DUDF_OCT && SP != 0 && H == 256 && SW <= SWEEP_ADJ_REV;
    for (int g = g0; g < g1; g += NWB) {
        if constexpr (kOct) {
            if (g1 - g == 1) {                          // a pass with one group: all eight waves share it
                const unsigned oct_off = 3 * GeoB<H, SP>::CHUNKB + (base_of(SW) == SWEEP_FWD ? (unsigned)(a.L * H * sizeof(float)) : (unsigned)(kMaxAmaxLayers * sizeof(unsigned)));
                sweep_tile_oct<H, SW, FL, P24>(a, gbase + g, lds_b, gc, oct_off);
                continue;
            }
        }
        sweep_tile_b<H, SW, FL, SP, P24>(a, gbase + g, (g1 - g < NWB) ? g1 - g : NWB, lds_b, gc, (DUDF_SWEEP_DBG & 128) && bid == 100 && g == g0);
    }
    if constexpr (kRow >= 0) {
        __syncthreads();
        if ((int)threadIdx.x < a.L && (int)threadIdx.x < kMaxAmaxLayers && a.amax) {
            const unsigned v = lds_amax[threadIdx.x];
            if (v) atomicMax(a.amax + kRow * a.L + threadIdx.x, v);
        }
    }
    if (clk_on && threadIdx.x == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}
// Two code generations of the same body.  The packed fp32 instructions (v_pk_fma_f32 ...) halve the vector-ALU issue
// slots of a tail, but they do not execute beside the SIMD partner's MFMAs (tools/micro/coissue.hip: 48 v_pk_fma_f32 +
// 24 MFMAs take the SUM of their times, 48 v_fma_f32 + 24 MFMAs the maximum + 25 %).  The forward sweep, whose sin/cos
// tail is as long as its MFMA stream, is therefore built WITHOUT them and overlaps the two; the other sweeps (short
// tails, bound by the stash stream) keep them.
// (the Hessian-quad and jet variants too: +3 % / +5 % on the Hessian-frame and curvature queries)
template <int SW> constexpr bool sweep_no_pk() { return SW == SWEEP_FWD || SW >= SWEEP_FWD_H; }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_bf16_kernel(SweepArgs a) { sweep_body_b<H, SW, FL>(a, blockIdx.x, gridDim.x); }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) DUDF_NO_PK void sweep_bf16_np_kernel(SweepArgs a) { sweep_body_b<H, SW, FL>(a, blockIdx.x, gridDim.x); }
// the fp16x3 builds (own names: tests/isa_contract.py tells the two families apart by them)
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_f16_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1>(a, blockIdx.x, gridDim.x); }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) DUDF_NO_PK void sweep_f16_np_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1>(a, blockIdx.x, gridDim.x); }
// ... and the fp16x3 builds that keep stash arrays at 24 bits, tile-major (dudf_internal.h "p24"; training variants):
// f16r: R, E and C (mask 6, the default stash of 256-wide networks); f16p: S, Q, A, Z as well (mask 7, DUDF_STASH=17p24)
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_f16r_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1, 6>(a, blockIdx.x, gridDim.x); }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) DUDF_NO_PK void sweep_f16r_np_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1, 6>(a, blockIdx.x, gridDim.x); }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_f16p_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1, 7>(a, blockIdx.x, gridDim.x); }
template <int H, int SW, int FL>
__global__ __launch_bounds__(64 * NWB) DUDF_NO_PK void sweep_f16p_np_kernel(SweepArgs a) { sweep_body_b<H, SW, FL, 1, 7>(a, blockIdx.x, gridDim.x); }
// Pair launch (a batch with Hessian-path points: `loss_s1` with its eigenvector term, the reference's shipped recipe).  A sweep
// then has two column ranges — the quads (variant SWQ; fp16x3 or, SPQ = 0, bf16x6) and the plain columns (fp16x3, variant SWP) — which used to be
// two launches of <= 256 persistent workgroups each: at the reference's batch (29 970 points = 312 + 156 tiles of 128 columns)
// that is 1.2 rounds + 0.6 rounds, each rounded up by the tail of its own launch.  Here ONE grid carries both: the first
// `nbq` workgroups walk the quads, the rest the plain columns, and the host splits the 256 workgroups so that both parts
// finish together (launch_pair).  The two bodies are the ones above, unchanged.
template <int H, int SWQ, int FLQ, int SWP, int FLP, int SPQ, int P24 = 0>
__global__ __launch_bounds__(64 * NWB) DUDF_NO_PK void sweep_pair_kernel(SweepArgs aq, SweepArgs ap, int nbq) {
    if ((int)blockIdx.x < nbq) sweep_body_b<H, SWQ, FLQ, SPQ, P24>(aq, blockIdx.x, nbq);
    else sweep_body_b<H, SWP, FLP, 1, P24>(ap, (int)blockIdx.x - nbq, (int)gridDim.x - nbq);
}

// theta -> bf16x3 images in A-fragment order of W_l (forward sweeps) and W_l^T (reverse sweeps), l = 2..L
template <int H>
__device__ __forceinline__ void pack_bf16_body(const float* __restrict__ theta, char* __restrict__ img_f,
                                               char* __restrict__ img_t, int nhid, int64_t off_hid,
                                               int64_t hid_stride, int64_t block, int64_t nblocks) {
    using G = GeoB<H>;
    const int64_t total = (int64_t)2 * nhid * G::NKB * G::NT * 64;
    for (int64_t idx = block * 256 + threadIdx.x; idx < total; idx += nblocks * 256) {
        int64_t v = idx;
        const int lane = (int)(v & 63); v >>= 6;
        const int T = (int)(v % G::NT); v /= G::NT;
        const int kb = (int)(v % G::NKB); v /= G::NKB;
        const int j = (int)(v % nhid); v /= nhid;
        const int dir = (int)v;
        const int m = lane & 15, g = lane >> 4;
        const int row = 16 * T + m;
        const float* W = theta + off_hid + (int64_t)j * hid_stride;
        f32x4 e0, e1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f0 = 32 * kb + 4 * g + e, f1 = f0 + 16;
            e0[e] = dir == 0 ? W[(int64_t)row * H + f0] : W[(int64_t)f0 * H + row];
            e1[e] = dir == 0 ? W[(int64_t)row * H + f1] : W[(int64_t)f1 * H + row];
        }
        u32x4 h, mm, l;
        split8(e0, e1, h, mm, l);
        char* base = (dir == 0 ? img_f : img_t) + (size_t)j * G::IMGB + (size_t)kb * G::CHUNKB + (size_t)T * 3 * G::FRAG + lane * 16;
        *reinterpret_cast<u32x4*>(base) = h;
        *reinterpret_cast<u32x4*>(base + G::FRAG) = mm;
        *reinterpret_cast<u32x4*>(base + 2 * G::FRAG) = l;
    }
}
template <int H>
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ theta, char* __restrict__ img_f,
                                                        char* __restrict__ img_t, int nhid, int64_t off_hid,
                                                        int64_t hid_stride) {
    pack_bf16_body<H>(theta, img_f, img_t, nhid, off_hid, hid_stride, blockIdx.x, gridDim.x);
}

// theta -> fp16 hi/lo images of 2^k_j W_l and 2^k_j W_l^T in the same A-fragment order, k_j = 15 - (exponent of max |W_l|):
// the largest weight lands in [2^14, 2^15), weights down to 2^-18 of it keep two full pieces, smaller ones an absolute
// error of 2^-40 of the largest.  grid = (blocks per matrix, L - 1); every block reduces max |W_l| itself (256 KB from L2).
template <int H>
__device__ __forceinline__ void pack_f16_body(const float* __restrict__ theta, char* __restrict__ img_f,
                                              char* __restrict__ img_t, float* __restrict__ wsc, int nhid,
                                              int64_t off_hid, int64_t hid_stride, int j, int sub, int nsub) {
    using G = GeoB<H, 1>;
    const float* W = theta + off_hid + (int64_t)j * hid_stride;
    __shared__ float red[4];
    float mx = 0.f;
    for (int i = threadIdx.x; i < H * H / 4; i += 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(W)[i];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    int ex = 0;
    (void)frexpf(mx, &ex);                              // mx < 2^ex (0 -> 0; inf / nan: whatever, the step is lost anyway)
    ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
    const float sc = ldexpf(1.f, 15 - ex);
    if (sub == 0 && threadIdx.x == 0) { wsc[j] = ldexpf(1.f, ex - 15); wsc[nhid + j] = sc; }
    const int per = 2 * G::NKB * G::NT * 64;            // lane-items of this matrix: [dir][k-block][tile][lane]
    for (int idx = sub * 256 + threadIdx.x; idx < per; idx += nsub * 256) {
        int v = idx;
        const int lane = v & 63; v >>= 6;
        const int T = v % G::NT; v /= G::NT;
        const int kb = v % G::NKB; v /= G::NKB;
        const int dir = v;
        const int m = lane & 15, g = lane >> 4;
        const int row = 16 * T + m;
        f32x4 e0, e1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int f0 = 32 * kb + 4 * g + e, f1 = f0 + 16;
            e0[e] = sc * (dir == 0 ? W[(int64_t)row * H + f0] : W[(int64_t)f0 * H + row]);
            e1[e] = sc * (dir == 0 ? W[(int64_t)row * H + f1] : W[(int64_t)f1 * H + row]);
        }
        u32x4 h, l;
        split8h(e0, e1, h, l);
        char* base = (dir == 0 ? img_f : img_t) + (size_t)j * G::IMGB + (size_t)kb * G::CHUNKB + (size_t)T * 2 * G::FRAG + lane * 16;
        *reinterpret_cast<u32x4*>(base) = h;
        *reinterpret_cast<u32x4*>(base + G::FRAG) = l;
    }
}
template <int H>
__global__ __launch_bounds__(256) void pack_f16_kernel(const float* __restrict__ theta, char* __restrict__ img_f,
                                                       char* __restrict__ img_t, float* __restrict__ wsc, int nhid,
                                                       int64_t off_hid, int64_t hid_stride) {
    pack_f16_body<H>(theta, img_f, img_t, wsc, nhid, off_hid, hid_stride, blockIdx.y, blockIdx.x, gridDim.x);
}

// ---- everything a training forward needs in front of its sweeps, in ONE launch (the C ABI keeps its entry points; round 2
// launched pack, pack_bf16, x4 and two memsets separately: ~5 us each, 1.7 % of a 3.5 ms step).  Block roles by index range.
struct PrepArgs {
    const float* theta; const float* x;
    float *w1b, *b1s, *w1t16, *wt, *x4, *wsc;
    float rho;                                          // w0 / ww: the first layer as the kernels see it
    char *img16_f, *img16_t, *img_f, *img_t;
    unsigned* zero; int nzero;                          // the loss sums + ticket and the running maxima: nzero dwords from `zero`
    unsigned* zero2; int nzero2;
    int L, nsub;
    int64_t off_hid, hid_stride, n, n_h, ncol_h, np;
    int nb_f16, nb_bf16, nb_x4, nb_thin, nb_wt;         // blocks per role
};
template <int H>
__global__ __launch_bounds__(256) void prep_kernel(PrepArgs a) {
    int b = blockIdx.x;
    const int nhid = a.L - 1;
    if (b < a.nb_f16) { pack_f16_body<H>(a.theta, a.img16_f, a.img16_t, a.wsc, nhid, a.off_hid, a.hid_stride, b / a.nsub, b % a.nsub, a.nsub); return; }
    b -= a.nb_f16;
    if (b < a.nb_bf16) { pack_bf16_body<H>(a.theta, a.img_f, a.img_t, nhid, a.off_hid, a.hid_stride, b, a.nb_bf16); return; }
    b -= a.nb_bf16;
    if (b < a.nb_x4) {
        // x4: the layer-1 B operand of every column.  plain column: (x0,x1,x2,1); Hessian quad: channel 0 the same, channel
        // 1+k = (e_k, 0); padding: zeros (as make_x4_kernel, dudf_misc.hip)
        for (int64_t c = (int64_t)b * 256 + threadIdx.x; c < a.np; c += (int64_t)a.nb_x4 * 256) {
            f32x4 v = {0, 0, 0, 0};
            if (c < a.ncol_h) {
                const int64_t p = c >> 2; const int ch = (int)(c & 3);
                if (p < a.n_h) {
                    if (ch == 0) v = f32x4{a.x[p * 3], a.x[p * 3 + 1], a.x[p * 3 + 2], 1.f};
                    else v[ch - 1] = 1.f;
                }
            } else {
                const int64_t p = a.n_h + (c - a.ncol_h);
                if (p < a.n) v = f32x4{a.x[p * 3], a.x[p * 3 + 1], a.x[p * 3 + 2], 1.f};
            }
            *reinterpret_cast<f32x4*>(a.x4 + c * 4) = v;
        }
        return;
    }
    b -= a.nb_x4;
    if (b < a.nb_thin) {                                 // w1b[f][k] = k<3 ? W_1[f][k] : b_1[f];  w1t16[r][f] = r<3 ? W_1[f][r] : 0;  zeros
        const int n_thin = 16 * H > a.L * H ? 16 * H : a.L * H;
        for (int gid = b * 256 + threadIdx.x; gid < n_thin; gid += a.nb_thin * 256) {
            if (gid < 4 * H) {
                const int f = gid / 4, k = gid % 4;
                a.w1b[gid] = a.rho * (k < 3 ? a.theta[f * 3 + k] : a.theta[3 * H + f]);
                if (k == 3) a.b1s[f] = a.rho * a.theta[3 * H + f];
            }
            if (gid >= H && gid < a.L * H) {                 // b1s rows 1 .. L-1 = b_2 .. b_L (16 H >= L H is not guaranteed: see the loop bound)
                const int layer = gid / H, f = gid % H;
                a.b1s[gid] = a.theta[a.off_hid + (int64_t)(layer - 1) * a.hid_stride + (int64_t)H * H + f];
            }
            if (gid < 16 * H) {
                const int r = gid / H, f = gid % H;
                a.w1t16[gid] = r < 3 ? a.rho * a.theta[f * 3 + r] : 0.f;
            }
        }
        if (b == 0) {
            for (int i = threadIdx.x; i < a.nzero; i += 256) a.zero[i] = 0u;
            for (int i = threadIdx.x; i < a.nzero2; i += 256) a.zero2[i] = 0u;
        }
        return;
    }
    b -= a.nb_thin;
    {                                                    // wt[j][i][o] = W_{j+2}[o][i] (f32-input reverse sweeps)
        const int64_t n_wt = (int64_t)nhid * H * H;
        for (int64_t gid = (int64_t)b * 256 + threadIdx.x; gid < n_wt; gid += (int64_t)a.nb_wt * 256) {
            const int64_t j = gid / ((int64_t)H * H), rem = gid % ((int64_t)H * H);
            const int i = (int)(rem / H), o = (int)(rem % H);
            a.wt[gid] = a.theta[a.off_hid + j * a.hid_stride + (int64_t)o * H + i];
        }
    }
}

template <int H, int SW, int FL>
const void* sweep_kernel_ptr() {
    if constexpr (sweep_no_pk<SW>()) return reinterpret_cast<const void*>(&sweep_bf16_np_kernel<H, SW, FL>);
    else return reinterpret_cast<const void*>(&sweep_bf16_kernel<H, SW, FL>);
}
constexpr int kMaxLdsBiasLayers = 32;                  // fp16x3 forward sweep: b_1..b_L live in LDS (32 KiB at H = 256); deeper nets: bf16x6
template <int H>
int launch_b(int which, const SweepArgs& a0, hipStream_t st) {
    using G = GeoB<H>;
    SweepArgs a = a0;
    const size_t smem = 3 * G::CHUNKB + kMaxAmaxLayers * sizeof(unsigned);   // + the per-layer running maxima
    if (a.ntiles <= 0) return 0;
    const int ntb = (a.ntiles * TILE + TILEB - 1) / TILEB;
    int grid = ntb < 256 ? ntb : 256;                  // one resident 8-wave workgroup per CU
    hipError_t e = hipSuccess;
#define DUDF_GO_B(SW, FL)                                                                                   \
    do {                                                                                                    \
        if (SW <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + SW, 6);                                \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(sweep_kernel_ptr<H, SW, FL>(),                                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        if constexpr (sweep_no_pk<SW>())                                                                    \
            hipLaunchKernelGGL((sweep_bf16_np_kernel<H, SW, FL>), dim3(grid), dim3(G::NTHR), smem, st, a);  \
        else                                                                                                \
            hipLaunchKernelGGL((sweep_bf16_kernel<H, SW, FL>), dim3(grid), dim3(G::NTHR), smem, st, a);     \
    } while (0)
#define DUDF_GO_H(SW, FL, KERNEL, SMEM_MAX, SMEM)                                                           \
    do {                                                                                                    \
        if (SW <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + SW, 3);                                \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            (void)(SMEM_MAX);                                                                               \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL<H, SW, FL>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCu);               \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((KERNEL<H, SW, FL>), dim3(grid), dim3(G::NTHR), (SMEM), st, a);                  \
    } while (0)
    // fp16x3 (DUDF_SPLIT, DUDF_SPLIT_SWEEPS): the plain columns' four sweeps
    if (which <= SWEEP_ADJ_REV && ((a.split >> which) & 1)) {
        constexpr size_t w3 = 3 * GeoB<H, 1>::CHUNKB;
        constexpr size_t oct = (H == 256) ? kOctBytes : 0;                   // + the exchange area of the one-group pass
        const size_t smem_f = w3 + (size_t)a.L * H * sizeof(float) + oct;    // + the biases (forward sweep)
        constexpr size_t smem_fmax = w3 + kMaxLdsBiasLayers * H * sizeof(float) + oct;
        constexpr size_t smem_o = w3 + kMaxAmaxLayers * sizeof(unsigned) + oct;   // + the per-layer running maxima
        bool done = true;
        // 24-bit tile-major stash (SweepArgs::p24, H = 256): the training variants have a build of their own
#define DUDF_GO_HP(SW, FL, KERNEL, KERNELR, KERNELP, SMEM_MAX, SMEM)                                        \
        do {                                                                                                    \
            bool p_ = false;                                                                                    \
            if constexpr (H == 256) {                                                                           \
                if (a.p24 == 7) { DUDF_GO_H(SW, FL, KERNELP, SMEM_MAX, SMEM); p_ = true; }                      \
                else if (a.p24 == 6) { DUDF_GO_H(SW, FL, KERNELR, SMEM_MAX, SMEM); p_ = true; }                 \
            }                                                                                                   \
            if (!p_) { if (a.p24) return DUDF_E_UNSUPPORTED; DUDF_GO_H(SW, FL, KERNEL, SMEM_MAX, SMEM); }       \
        } while (0)
        // (a 24-bit workspace holds C as fixed point: only the training variants, which are built for it, may touch it)
        if (a.p24 && ((which == SWEEP_FWD && !(a.store_s && a.store_c)) || (which == SWEEP_REV && !a.train))) return DUDF_E_UNSUPPORTED;
        if (which == SWEEP_FWD && a.L <= kMaxLdsBiasLayers) {
            if (a.store_s && a.store_c) DUDF_GO_HP(SWEEP_FWD, 3, DUDF_FWD_F16_KERNEL, sweep_f16r_np_kernel, sweep_f16p_np_kernel, smem_fmax, smem_f);
            else if (a.store_c) DUDF_GO_H(SWEEP_FWD, 2, DUDF_FWD_F16_KERNEL, smem_fmax, smem_f);
            else if (!a.store_s) DUDF_GO_H(SWEEP_FWD, 0, DUDF_FWD_F16_KERNEL, smem_fmax, smem_f);
            else return DUDF_E_BADMODE;
        } else if (which == SWEEP_REV) {
            if (a.train) DUDF_GO_HP(SWEEP_REV, 1, sweep_f16_kernel, sweep_f16r_kernel, sweep_f16p_kernel, smem_o, smem_o); else DUDF_GO_H(SWEEP_REV, 0, sweep_f16_kernel, smem_o, smem_o);
        } else if (which == SWEEP_ADJ_FWD) {
            DUDF_GO_HP(SWEEP_ADJ_FWD, 0, sweep_f16_kernel, sweep_f16r_kernel, sweep_f16p_kernel, smem_o, smem_o);
        } else if (which == SWEEP_ADJ_REV && (!a.have_e || (a.ebound && ((a.split >> SWEEP_ADJ_FWD) & 1)))) {
            if (a.have_e) DUDF_GO_HP(SWEEP_ADJ_REV, 1, sweep_f16_kernel, sweep_f16r_kernel, sweep_f16p_kernel, smem_o, smem_o); else DUDF_GO_HP(SWEEP_ADJ_REV, 0, sweep_f16_kernel, sweep_f16r_kernel, sweep_f16p_kernel, smem_o, smem_o);
        } else {
            done = false;
        }
        if (done) return (int)hipGetLastError();
        if (a.p24) return DUDF_E_UNSUPPORTED;                  // a 24-bit workspace has no other kernels
    }
    // ... and the Hessian quads' sweeps (bit 5, DUDF_SPLIT_QUADS; the jets stay on bf16x6).  All of a workspace's or none:
    // the forward sweep leaves zbound for the other three, the adjoint forward sweep ebound for the adjoint reverse one.
    if (which >= SWEEP_FWD_H && which <= SWEEP_ADJ_REV_H && (a.split & 32) && a.zbound && a.L <= kMaxLdsBiasLayers) {
        constexpr size_t w3 = 3 * GeoB<H, 1>::CHUNKB;
        const size_t smem_q = w3 + kMaxAmaxLayers * sizeof(unsigned);
        const size_t smem_fq = w3 + (size_t)a.L * H * sizeof(float) + kMaxAmaxLayers * sizeof(unsigned);
        constexpr size_t smem_fqmax = w3 + kMaxLdsBiasLayers * H * sizeof(float) + kMaxAmaxLayers * sizeof(unsigned);
        bool done = true;
        if (a.p24 && ((which == SWEEP_FWD_H && !a.store_s) || (which == SWEEP_REV_H && !a.train))) return DUDF_E_UNSUPPORTED;
        if (which == SWEEP_FWD_H) { if (a.store_s) DUDF_GO_HP(SWEEP_FWD_H, 1, sweep_f16_np_kernel, sweep_f16r_np_kernel, sweep_f16p_np_kernel, smem_fqmax, smem_fq); else DUDF_GO_H(SWEEP_FWD_H, 0, sweep_f16_np_kernel, smem_fqmax, smem_fq); }
        else if (which == SWEEP_REV_H) { if (a.train) DUDF_GO_HP(SWEEP_REV_H, 1, sweep_f16_np_kernel, sweep_f16r_np_kernel, sweep_f16p_np_kernel, smem_q, smem_q); else DUDF_GO_H(SWEEP_REV_H, 0, sweep_f16_np_kernel, smem_q, smem_q); }
        else if (which == SWEEP_ADJ_FWD_H && a.ebound) DUDF_GO_HP(SWEEP_ADJ_FWD_H, 0, sweep_f16_np_kernel, sweep_f16r_np_kernel, sweep_f16p_np_kernel, smem_q, smem_q);
        else if (which == SWEEP_ADJ_REV_H && a.ebound) DUDF_GO_HP(SWEEP_ADJ_REV_H, 0, sweep_f16_np_kernel, sweep_f16r_np_kernel, sweep_f16p_np_kernel, smem_q, smem_q);
        else done = false;
        if (done) return (int)hipGetLastError();
    }
    if (a.p24 && which != SWEEP_FWD_J) return DUDF_E_UNSUPPORTED;     // (the jets stash nothing)
    if (which == SWEEP_FWD_J && (a.split & 32) && a.L <= kMaxLdsBiasLayers) {      // the third-order jets (curvature query): nothing stashed
        constexpr size_t w3 = 3 * GeoB<H, 1>::CHUNKB;
        const size_t smem_j = w3 + (size_t)a.L * H * sizeof(float) + kMaxAmaxLayers * sizeof(unsigned);
        constexpr size_t smem_jmax = w3 + kMaxLdsBiasLayers * H * sizeof(float) + kMaxAmaxLayers * sizeof(unsigned);
        DUDF_GO_H(SWEEP_FWD_J, 0, sweep_f16_np_kernel, smem_jmax, smem_j);
        return (int)hipGetLastError();
    }
#undef DUDF_GO_HP
#undef DUDF_GO_H
    switch (which) {
        case SWEEP_FWD:
            if (a.store_s && a.store_c) DUDF_GO_B(SWEEP_FWD, 3);
            else if (a.store_c) DUDF_GO_B(SWEEP_FWD, 2);          // value + df/dx query: only cos is read again
            else if (!a.store_s) DUDF_GO_B(SWEEP_FWD, 0);         // value-only query
            else return DUDF_E_BADMODE;
            break;
        case SWEEP_REV:
            if (a.train) DUDF_GO_B(SWEEP_REV, 1); else DUDF_GO_B(SWEEP_REV, 0);
            break;
        case SWEEP_ADJ_FWD: DUDF_GO_B(SWEEP_ADJ_FWD, 0); break;
        case SWEEP_ADJ_REV:
            if (a.have_e) DUDF_GO_B(SWEEP_ADJ_REV, 1); else DUDF_GO_B(SWEEP_ADJ_REV, 0);
            break;
        // Hessian quads: same kernel, the tails couple the 4 lanes of a quad by DPP
        case SWEEP_FWD_H:
            if (a.store_s) DUDF_GO_B(SWEEP_FWD_H, 1); else DUDF_GO_B(SWEEP_FWD_H, 0);   // queries do not need h | hdot again
            break;
        case SWEEP_REV_H:
            if (a.train) DUDF_GO_B(SWEEP_REV_H, 1); else DUDF_GO_B(SWEEP_REV_H, 0);
            break;
        case SWEEP_ADJ_FWD_H: DUDF_GO_B(SWEEP_ADJ_FWD_H, 0); break;
        case SWEEP_ADJ_REV_H: DUDF_GO_B(SWEEP_ADJ_REV_H, 0); break;
        case SWEEP_FWD_J: DUDF_GO_B(SWEEP_FWD_J, 0); break;      // third-order Taylor jets (curvature query), nothing stashed
        default: return DUDF_E_UNSUPPORTED;
    }
#undef DUDF_GO_B
    return (int)hipGetLastError();
}


// ====================================================================================================================
// 512-wide layers (BASELINE.json configs[2]: SIREN 8x512), plain columns, training variants.
// The scheme above keeps this layer's AND the previous layer's accumulators in registers (2 x NT x 4): at H = 512 that
// is 256 registers before anything else, i.e. one wave per SIMD and 64-column workgroups, which the weight stream cannot
// feed.  Here a wave keeps only THIS layer's 32 accumulator tiles (128 registers, two waves per SIMD, 128-column
// workgroups as above) and the previous layer's outputs travel through the stash arrays the sweep writes anyway:
//   * when a layer's accumulators are final its elementwise tails run in one burst (same `epilogue` as everywhere: bias,
//     sin/cos or adjoint formulas, stash stores) — the post-tail value of every tile is exactly what one of those stores
//     leaves behind (forward: h_l in S; reverse: q_l in Q; adjoint forward: A_l; adjoint reverse: zbar_l in Z);
//   * the next layer reads its B operand back, one k-block (two 16-byte loads per lane) ahead of its use — this wave's
//     own 2 KB per column, written a moment ago — splits it into the three bf16 pieces, and uses it for the 32 output
//     tiles in two half-steps of 16 tiles: a weight chunk stays 48 KiB ([k-block][half]: the image of a k-block is
//     [tile][piece], so a half is contiguous) and the three-buffer LDS-DMA stream is the one above;
//   * the read-back loads are inline asm like the DMA: inside the k-loop the compiler sees no vector-memory operation,
//     every wait is hand-counted (derivation at the waits); around a tail burst everything is drained once per layer.
// Cost against the register-resident scheme: one more stash unit READ per layer and sweep (largely served by L2 / the
// Infinity Cache: it is the unit just written), and the burst is not overlapped with this wave's own MFMAs.
// which stash array carries the post-tail values of sweep SW to the next layer.  Where the tail does not store them itself
// (queries: the reverse sweep without its training stores, the jets) the kernel stores them into S, which no later tail of
// the same sweep reads.
template <int SW, int FL>
__device__ __forceinline__ const float* wide_in(const SweepArgs& a) {
    constexpr int BS = base_of(SW);
    if constexpr (BS == SWEEP_FWD) return a.S;
    else if constexpr (BS == SWEEP_REV) return (FL & 1) ? a.Q : a.S;
    else if constexpr (BS == SWEEP_ADJ_FWD) return a.A;
    else return a.Z;
}
template <int SW, int FL>
constexpr bool wide_relay_store() { return (base_of(SW) == SWEEP_REV && !(FL & 1)) || SW == SWEEP_FWD_J; }

template <int SP>
struct GeoWT {
    static constexpr int NPC = SP ? 2 : 3;              // pieces per operand (fp16 hi | lo, or bf16 h | m | l)
    static constexpr int H = 512, NT = 32, NKB = 16, FRAG = 1024;
    static constexpr int HALFT = 16;                    // tiles per half-step
    static constexpr int CHUNKB = HALFT * NPC * FRAG;   // 48 (32) KiB: one (k-block, half) of a matrix
    static constexpr int IMGB = NKB * 2 * CHUNKB;       // one matrix (= GeoB<512, SP>::IMGB)
    static constexpr int NDMA = HALFT * NPC / NWB;      // 6 (4) LDS-DMA wave-instructions per wave and chunk
    static constexpr int NTHR = 64 * NWB;
};
using GeoW = GeoWT<0>;

// SP = 1: fp16x3 (see GeoB).  The B operand of a layer is read back from the stash AFTER the whole previous layer has been
// written, so its per-column scale is exact here: 2^15 over the column's largest |output| of the tail burst.
// P24 (0 or 6): R, E as 24-bit floats and C as 24-bit fixed point, tile-major (dudf_internal.h) — the arrays that are NOT the relay.
template <int SW, int FL, int SP = 0, int P24 = 0>
__device__ __forceinline__ void sweep_tile_w(const SweepArgs& a, const int g_first, const int nact, char* lds, unsigned& gc) {
    static_assert(P24 == 0 || (P24 == 6 && SP != 0 && !is_jet(SW)), "24-bit stash arrays in the 512-wide kernel: R, E, C of the fp16x3 training variants");
    using G = GeoWT<SP>;
    constexpr int H = G::H;
    constexpr int NPC = G::NPC;
    constexpr int BS = base_of(SW);
    constexpr bool HS = is_hess(SW);                   // Hessian quads / jets: the tails couple lanes (dudf_sweep_common.h)
    constexpr bool kColScale = SP != 0 && SW != SWEEP_FWD;   // (the quads' forward tangents are not bounded by 1)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const bool isv = !HS || (is_jet(SW) ? li == 0 : (lane & 3) == 0);
    const int nhid = a.L - 1;
    constexpr bool kFwdDir = (BS == SWEEP_FWD || BS == SWEEP_ADJ_FWD);
    const int64_t p = (int64_t)(g_first + wave) * 16 + li;
    auto image = [&](int j) -> const char* {
        if constexpr (SP) return kFwdDir ? a.wimg16_f + (size_t)j * G::IMGB : a.wimg16_t + (size_t)(nhid - 1 - j) * G::IMGB;
        return kFwdDir ? a.wimg_f + (size_t)j * G::IMGB : a.wimg_t + (size_t)(nhid - 1 - j) * G::IMGB;
    };
    auto unscale_of = [&](int j) -> float { return a.wsc[kFwdDir ? j : nhid - 1 - j]; };
    float unscale = 1.f, sb = 1.f, inv_sb = 1.f;        // accumulators -> true values | scale of the B operand being read back
    auto in_layer = [&](int j) -> int { return kFwdDir ? j : a.L - 1 - j; };
    auto bias_ptr = [&](int layer) -> const float* {
        return a.b1s + (size_t)layer * H;              // [L][H] biases as packed (row 0 = rho b_1)
    };
    auto stash_base = [&](int layer, int T) -> int64_t {
        const int64_t v = (int64_t)layer * a.stash_layer + (int64_t)(16 * T) * a.np;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (int64_t)(((uint64_t)hi << 32) | lo);
    };
    const unsigned vo = (unsigned)(((int64_t)q * a.np + p) * 16);
    const LaneOff vl(vo, (is_hess(SW) && !is_jet(SW)) ? (unsigned)(((int64_t)q * a.np + (p >> 2)) * 16) : vo,   // C: one copy per quad
                     P24 ? (unsigned)(((p >> 4) * 64 + lane) * 12) : 0u,                                          // 24-bit tile-major arrays
                     !P24 ? 0u : (is_hess(SW) && !is_jet(SW)) ? (unsigned)(((p >> 6) * 64 + 16 * q + ((p >> 2) & 15)) * 12)
                                                              : (unsigned)(((p >> 4) * 64 + lane) * 12));
    const int total2 = nhid * G::NKB * 2;
    auto chunk_src = [&](int c2) -> const char* {       // c2 = (matrix, k-block, half), wave-uniform
        const int j = c2 / (G::NKB * 2);
        return image(j) + (size_t)(c2 - j * G::NKB * 2) * G::CHUNKB;
    };
    auto dma = [&](int c2, unsigned buf) {
        // GeoB<256, SP> has the same chunk geometry (16 tiles x NPC pieces): reuse its issue code
        dma_issue<256, SP>(chunk_src(c2), (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + buf * G::CHUNKB,
                       (unsigned)lane * 16u, wave);
    };
    __syncthreads();                                   // every wave is past its last LDS read of the previous pass
    dma(0, gc);
    dma(1, (gc + 1) % 3);
    if (wave >= nact) {                                // idle waves of a partial pass: same DMA pieces, same barriers
        dma_wait_b<0>();
        __syncthreads();
        for (int j = 0; j < nhid; ++j) {
            for (int hs = 0; hs < G::NKB * 2; ++hs) {
                const int c2 = j * G::NKB * 2 + hs;
                const bool more = c2 + 2 < total2;
                if (more) dma(c2 + 2, (gc + 2) % 3);
                gc = (gc + 1) % 3;
                if (more) dma_wait_b<G::NDMA>(); else dma_wait_b<0>();
                __syncthreads();
            }
            dma_wait_b<0>();
            __syncthreads();                           // the barrier behind the active waves' tail burst
        }
        return;
    }

    f32x4 acc[G::NT];
    float part = 0.f;                                  // forward: y partial sums; reverse: df/dx accumulator
    f32x4 accg = {0, 0, 0, 0};
    // ---- the elementwise tails of all 32 tiles of `layer` (operands one pair ahead), stash stores, output stage ----
    constexpr int kRow = amax_row<SW, FL>();
    unsigned* lds_amax = reinterpret_cast<unsigned*>(lds + 3 * G::CHUNKB);
    auto tail_burst = [&](int layer, bool last) {
        TailTrack tmax;
        // operand ring: the stash operands of tile T + PD are requested when tile T has been consumed.  One tile of tail is
        // ~100 instructions, an HBM round trip ~2 us: with the operands only one tile ahead the burst waited for memory at
        // every tile (it took about as long as the layer's whole k-loop); the forward sweep only reads its bias (cached).
        // (three-operand tails — the quads' adjoint sweeps — get a ring of four: 128 accumulator registers leave no more)
        constexpr int PD = (BS == SWEEP_FWD) ? 2 : ((SW == SWEEP_ADJ_FWD_H || SW == SWEEP_ADJ_REV_H) ? 4 : 8);
        f32x4 o1[PD], o2[PD], o3[PD], bs[PD];
        auto ld = [&](int T, int s) {
            epilogue_loads<SW, FL, P24>(a, stash_base(layer, T), vl, o1[s], o2[s], o3[s]);
            if constexpr (BS == SWEEP_FWD) bs[s] = *reinterpret_cast<const f32x4*>(bias_ptr(layer) + 16 * T + 4 * q);
        };
        float cmax = 0.f;                              // fp16x3: largest |output| of this lane's rows of the column
#pragma unroll
        for (int T = 0; T < PD; ++T) ld(T, T);
        // (two halves of 16 tiles, each its own fully unrolled loop: as ONE loop of 32 the larger tails — the quads' adjoint forward
        //  sweep with 24-bit arrays — exceed hipcc's size limit for a forced unroll, and the rolled loop indexes acc[] dynamically)
        //  The jets' tail is too large even so; their loop stays the single rolled one it has been since round 3.)
        auto burst_range = [&](auto t0c, auto t1c) {
        constexpr int T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
#pragma unroll
        for (int T = T0; T < T1; ++T) {
            const int s = T % PD;
            f32x4 z = acc[T];
            const f32x4 zero4 = {0, 0, 0, 0};
            if constexpr (SP != 0 && SW == SWEEP_FWD) z = __builtin_elementwise_fma(z, f32x4{unscale, unscale, unscale, unscale}, bs[s]);
            else if constexpr (SW == SWEEP_FWD) z += bs[s];
            else if constexpr (BS == SWEEP_FWD) z = (SP != 0 ? z * unscale : z) + (isv ? bs[s] : zero4);   // the bias: value channel only
            else if constexpr (SP != 0) z *= unscale;
            // (RL: the array the next layer reads its operand back from keeps the default cache policy — DUDF_W_RELAY_NT=1: A/B)
            const f32x4 e = epilogue<SW, FL, false, P24, !DUDF_W_RELAY_NT>(a, z, o1[s], o2[s], o3[s], stash_base(layer, T), vl, isv, tmax);
            if constexpr (wide_relay_store<SW, FL>()) { if constexpr (DUDF_W_RELAY_NT) DUDF_ST(a.S, stash_base(layer, T), vo, e); else DUDF_ST_CACHED(a.S, stash_base(layer, T), vo, e); }
            if constexpr (kColScale) dudf_track(cmax, e);
            if (T + PD < G::NT) ld(T + PD, s);
            if (last) {
                if constexpr (BS == SWEEP_FWD) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q);
                    part += e[0] * wv[0] + e[1] * wv[1] + e[2] * wv[2] + e[3] * wv[3];
                } else if constexpr (BS == SWEEP_REV) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 16 * T + 4 * q);
#pragma unroll
                    for (int t = 0; t < 4; ++t) accg = mfma16(wv[t], e[t], accg);
                }
            }
            acc[T] = f32x4{0, 0, 0, 0};
        }
        };
        if constexpr (is_jet(SW)) {
            burst_range(std::integral_constant<int, 0>{}, std::integral_constant<int, G::NT>{});
        } else {
            burst_range(std::integral_constant<int, 0>{}, std::integral_constant<int, G::NT / 2>{});
            burst_range(std::integral_constant<int, G::NT / 2>{}, std::integral_constant<int, G::NT>{});
        }
        if constexpr (kRow >= 0) { if (layer < kMaxAmaxLayers) lds_max_wave(lds_amax + layer, tmax.t); }
        if constexpr (kColScale) {                     // the next layer's B operand = these outputs: scale the column below 2^15
            cmax = fmaxf(cmax, __shfl_xor(cmax, 16));
            cmax = fmaxf(cmax, __shfl_xor(cmax, 32));
            unsigned E = (__float_as_uint(cmax) >> 23) & 255u;         // cmax < 2^(E - 126)
            E = E < 27u ? 27u : (E > 250u ? 250u : E);
            sb = __uint_as_float((268u - E) << 23);
            inv_sb = __uint_as_float((E - 14u) << 23);
        }
    };
    // ---- first layer (fp32, K = 3): pre-activations / incoming adjoints of the 32 tiles, then their tails ----
    {
        float b = 0.f, yb = 1.f;
        if constexpr (BS == SWEEP_FWD) b = (q < 3) ? a.x4[p * 4 + q] : 0.f;
        if constexpr (BS == SWEEP_ADJ_FWD) b = (q < 3) ? a.gbar[p * 4 + q] : 0.f;
        if constexpr (BS == SWEEP_ADJ_REV) yb = a.ybar[p];
        if constexpr (SW == SWEEP_REV_H) yb = isv ? 1.f : 0.f;                         // adot_L^k = 0
#pragma unroll
        for (int T = 0; T < G::NT; ++T) {
            if constexpr (kFwdDir) acc[T] = mfma16(a.w1b[(16 * T + li) * 4 + q], b, f32x4{0, 0, 0, 0});
            else acc[T] = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
        }
        tail_burst(in_layer(0), false);
    }
    dma_wait_b<0>();                                   // chunks 0 and 1, and the burst's stores (read back below)
    __syncthreads();

    // read-back of the post-tail values of tiles 2kb, 2kb+1 of `layer`: two asm loads, scalar base + lane offset
    auto ld_in = [&](int layer, int kb, f32x4& x0, f32x4& x1) {
        const float* b0 = wide_in<SW, FL>(a) + stash_base(layer, 2 * kb);
        const float* b1 = b0 + 16 * a.np;              // next tile: 16 feature rows further (stash_base is linear in T)
        const uint64_t g0 = (uint64_t)(size_t)b0, g1 = (uint64_t)(size_t)b1;
        // (readfirstlane returns int: go through unsigned, or a low word with its top bit set sign-extends into the high word)
        const unsigned l0 = __builtin_amdgcn_readfirstlane((unsigned)g0), h0 = __builtin_amdgcn_readfirstlane((unsigned)(g0 >> 32));
        const unsigned l1 = __builtin_amdgcn_readfirstlane((unsigned)g1), h1 = __builtin_amdgcn_readfirstlane((unsigned)(g1 >> 32));
        const uint64_t s0 = ((uint64_t)h0 << 32) | l0, s1 = ((uint64_t)h1 << 32) | l1;
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %4"
                     : "=&v"(x0), "=&v"(x1) : "v"(vo), "s"(s0), "s"(s1) : "memory");
    };
    u32x4 bq[NPC];                                     // B operand of the current k-block
    for (int j = 0; j < nhid; ++j) {
        const int lin = in_layer(j);
        if constexpr (SP != 0) unscale = unscale_of(j) * inv_sb;     // what turns THIS matrix's accumulators into true values
        // read-back registers: `xa` carries the even k-blocks, `xb` the odd ones — the loop is unrolled by two so that a set
        // is never copied while its asm loads are in flight (a rolled loop would rotate them with v_mov at the back edge)
        f32x4 xa0, xa1, xb0, xb1;
#if DUDF_SWEEP_DBG & 128
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        ld_in(lin, 0, xa0, xa1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa0), "+v"(xa1));      // k-block 0: nothing to overlap it with yet
        auto kstep = [&](int kb, f32x4& c0, f32x4& c1, f32x4& n0, f32x4& n1, auto steady) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c2 = (j * G::NKB + kb) * 2 + h;
                const bool more = decltype(steady)::value || c2 + 2 < total2;    // compile-time in the steady loop: one basic block
                if (h == 0) {
                    // the read-back of this k-block was issued one k-block ago; younger than it: the 6 DMA pieces of the
                    // half-step in between
                    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(c0), "+v"(c1) : "n"(G::NDMA));
                    if constexpr (kColScale) split8h(c0 * sb, c1 * sb, bq[0], bq[1]);
                    else if constexpr (SP != 0) split8h(c0, c1, bq[0], bq[1]);
                    else split8(c0, c1, bq[0], bq[1], bq[2]);
                }
                if (more) dma(c2 + 2, (gc + 2) % 3);
                if (h == 0) ld_in(lin, kb + 1 < G::NKB ? kb + 1 : kb, n0, n1);   // the last one re-reads its own: uniform counts
                __builtin_amdgcn_sched_barrier(0);
                const char* bp = lds + gc * G::CHUNKB + lane * 16;
                auto frag = [&](int T, int pc) -> u32x4 {
                    return *reinterpret_cast<const u32x4*>(bp + (T * NPC + pc) * G::FRAG);
                };
                u32x4 an[2][NPC];
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) an[T][pc] = frag(T, pc);
#pragma unroll
                for (int T = 0; T < G::HALFT; ++T) {
                    u32x4 af[NPC];
#pragma unroll
                    for (int pc = 0; pc < NPC; ++pc) af[pc] = an[T & 1][pc];
                    if (T + 2 < G::HALFT) {
#pragma unroll
                        for (int pc = 0; pc < NPC; ++pc) an[T & 1][pc] = frag(T + 2, pc);
                        __builtin_amdgcn_sched_barrier(0x76);
                    }
                    f32x4 cc = acc[G::HALFT * h + T];
                    if constexpr (SP != 0) {                    // smallest terms first: lo*hi, hi*lo, hi*hi
                        cc = mfma_h(as_h(af[1]), as_h(bq[0]), cc);
                        cc = mfma_h(as_h(af[0]), as_h(bq[1]), cc);
                        cc = mfma_h(as_h(af[0]), as_h(bq[0]), cc);
                    } else {
                        const bf16x8 ah = as_bf(af[0]), am = as_bf(af[1]), al = as_bf(af[NPC - 1]);
                        cc = mfma_b(am, as_bf(bq[1]), cc);          // smallest terms first
                        cc = mfma_b(al, as_bf(bq[0]), cc);
                        cc = mfma_b(ah, as_bf(bq[NPC - 1]), cc);
                        cc = mfma_b(am, as_bf(bq[0]), cc);
                        cc = mfma_b(ah, as_bf(bq[1]), cc);
                        cc = mfma_b(ah, as_bf(bq[0]), cc);
                    }
                    acc[G::HALFT * h + T] = cc;
                }
                gc = (gc + 1) % 3;
                // chunk c2+1 has landed.  Issued after its DMA: h == 0: [this step: NDMA pieces + 2 read-back]; h == 1: [previous
                // step: 2 read-back] + [this step: NDMA pieces]  ->  NDMA + 2 younger operations either way
                if (more) dma_wait_b<G::NDMA + 2>(); else dma_wait_b<0>();
                __syncthreads();
            }
        };
        const int kb_steady = (j + 1 == nhid) ? G::NKB - 2 : G::NKB;   // the stream's last two half-chunks have no successor
#pragma unroll 1                                        // 384 MFMAs per iteration: the body stays inside the instruction cache
        for (int kb = 0; kb < kb_steady; kb += 2) {
            kstep(kb, xa0, xa1, xb0, xb1, std::true_type{});
            kstep(kb + 1, xb0, xb1, xa0, xa1, std::true_type{});
        }
        if (j + 1 == nhid) {
            kstep(G::NKB - 2, xa0, xa1, xb0, xb1, std::false_type{});
            kstep(G::NKB - 1, xb0, xb1, xa0, xa1, std::false_type{});
        }
        // the last k-block's (dummy) read-back is still in flight and nothing will consume it: keep its registers until it
        // has landed, or hipcc hands them to the burst below while the load is still writing them
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1));
#if DUDF_SWEEP_DBG & 128
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
        tail_burst(in_layer(j + 1), j + 1 == nhid);
#if DUDF_SWEEP_DBG & 128
        const unsigned long long tw2 = __builtin_amdgcn_s_memtime();
#endif
        dma_wait_b<0>();
        __syncthreads();
#if DUDF_SWEEP_DBG & 128
        if ((blockIdx.x == 100 || blockIdx.x == 101) && lane == 0 && (wave == 0 || wave == 4) && j < 8) {
            const unsigned long long tw3 = __builtin_amdgcn_s_memtime();
            unsigned long long* o = &g_stamp[SW & 3][(blockIdx.x - 100) * 2 + (wave >> 2)][j][0];
            o[0] = tw0; o[1] = tw1; o[2] = tw2; o[3] = tw3;
        }
#endif
    }
    if constexpr (BS == SWEEP_FWD) {
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (isv) part += a.theta[a.off_bo];             // tangent / jet columns are derivatives: no constant term
        if (q == 0) a.y[p] = part;
    } else if constexpr (BS == SWEEP_REV) {
        if (q == 0) *reinterpret_cast<f32x4*>(a.g + p * 4) = f32x4{accg[0], accg[1], accg[2], 0.f};
    }
}

template <int SW, int FL, int SP, int P24 = 0>
__device__ __forceinline__ void sweep_w_body(const SweepArgs& a) {
    extern __shared__ __attribute__((aligned(16))) char lds_w[];
    unsigned gc = 0;
    const bool clk_on = a.clk != nullptr && blockIdx.x == 0;
    const unsigned long long clk_t0 = clk_on ? __builtin_amdgcn_s_memtime() : 0ull, clk_r0 = clk_on ? __builtin_amdgcn_s_memrealtime() : 0ull;
    constexpr int kRow = amax_row<SW, FL>();
    unsigned* lds_amax = reinterpret_cast<unsigned*>(lds_w + 3 * GeoWT<SP>::CHUNKB);
    if constexpr (kRow >= 0) { if (threadIdx.x < kMaxAmaxLayers) lds_amax[threadIdx.x] = 0u; }
    // (measured and dropped: odd workgroups starting half a layer late, so that one half of the chip is in its compute phase
    //  — the k-loop — while the other is in its memory phase — the tail burst: +1.3 %)
    const int ng = a.ntiles * (TILE / 16), gbase = a.tile0 * (TILE / 16);
    const int g0 = (int)((int64_t)blockIdx.x * ng / gridDim.x), g1 = (int)((int64_t)(blockIdx.x + 1) * ng / gridDim.x);
    for (int g = g0; g < g1; g += NWB)
        sweep_tile_w<SW, FL, SP, P24>(a, gbase + g, (g1 - g < NWB) ? g1 - g : NWB, lds_w, gc);
    if constexpr (kRow >= 0) {
        __syncthreads();
        if ((int)threadIdx.x < a.L && (int)threadIdx.x < kMaxAmaxLayers && a.amax) {
            const unsigned v = lds_amax[threadIdx.x];
            if (v) atomicMax(a.amax + kRow * a.L + threadIdx.x, v);
        }
    }
    if (clk_on && threadIdx.x == 0) {
        a.clk[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
}
template <int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_w_kernel(SweepArgs a) { sweep_w_body<SW, FL, 0>(a); }
template <int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_w16_kernel(SweepArgs a) { sweep_w_body<SW, FL, 1>(a); }
// ... with R, E, C at 24 bits (stash mask 6: the training variants of a default training workspace)
template <int SW, int FL>
__global__ __launch_bounds__(64 * NWB) void sweep_w16r_kernel(SweepArgs a) { sweep_w_body<SW, FL, 1, 6>(a); }

int launch_w(int which, const SweepArgs& a, hipStream_t st) {
    using G = GeoW;
    const size_t smem = 3 * G::CHUNKB + kMaxAmaxLayers * sizeof(unsigned);
    if (a.ntiles <= 0) return 0;
    const int ntb = (a.ntiles * TILE + TILEB - 1) / TILEB;
    const int grid = ntb < 256 ? ntb : 256;
    hipError_t e = hipSuccess;
#define DUDF_GO_W(SW, FL)                                                                                   \
    do {                                                                                                    \
        if (SW <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + SW, 6);                                \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_w_kernel<SW, FL>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((sweep_w_kernel<SW, FL>), dim3(grid), dim3(G::NTHR), smem, st, a);               \
    } while (0)
#define DUDF_GO_W16(SW, FL)                                                                                 \
    do {                                                                                                    \
        if (SW <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + SW, 3);                                \
        static bool attr_done = false;                                                                      \
        const size_t smem16 = 3 * GeoWT<1>::CHUNKB + kMaxAmaxLayers * sizeof(unsigned);                     \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_w16_kernel<SW, FL>),               \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);               \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((sweep_w16_kernel<SW, FL>), dim3(grid), dim3(G::NTHR), smem16, st, a);           \
    } while (0)
#define DUDF_GO_W16R(SW, FL)                                                                                \
    do {                                                                                                    \
        if (SW <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + SW, 3);                                \
        static bool attr_done = false;                                                                      \
        const size_t smem16 = 3 * GeoWT<1>::CHUNKB + kMaxAmaxLayers * sizeof(unsigned);                     \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_w16r_kernel<SW, FL>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);               \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((sweep_w16r_kernel<SW, FL>), dim3(grid), dim3(G::NTHR), smem16, st, a);          \
    } while (0)
    // fp16x3 or bf16x6: plain columns by their bit of the split mask, quads and jets by bit 5 (DUDF_SPLIT_QUADS)
    const bool h16 = which <= SWEEP_ADJ_REV ? ((a.split >> which) & 1) != 0 : (a.split & 32) != 0;
    if (a.p24) {                                     // a training workspace with R, E, C at 24 bits: its training variants only
        if (a.p24 != 6 || !h16) return DUDF_E_UNSUPPORTED;
        switch (which) {
            case SWEEP_FWD: if (a.store_s && a.store_c) DUDF_GO_W16R(SWEEP_FWD, 3); else return DUDF_E_UNSUPPORTED; break;
            case SWEEP_REV: if (a.train) DUDF_GO_W16R(SWEEP_REV, 1); else return DUDF_E_UNSUPPORTED; break;
            case SWEEP_ADJ_FWD: DUDF_GO_W16R(SWEEP_ADJ_FWD, 0); break;
            case SWEEP_ADJ_REV: if (a.have_e) DUDF_GO_W16R(SWEEP_ADJ_REV, 1); else DUDF_GO_W16R(SWEEP_ADJ_REV, 0); break;
            case SWEEP_FWD_H: if (a.store_s) DUDF_GO_W16R(SWEEP_FWD_H, 1); else return DUDF_E_UNSUPPORTED; break;
            case SWEEP_REV_H: if (a.train) DUDF_GO_W16R(SWEEP_REV_H, 1); else return DUDF_E_UNSUPPORTED; break;
            case SWEEP_ADJ_FWD_H: DUDF_GO_W16R(SWEEP_ADJ_FWD_H, 0); break;
            case SWEEP_ADJ_REV_H: DUDF_GO_W16R(SWEEP_ADJ_REV_H, 0); break;
            default: return DUDF_E_UNSUPPORTED;
        }
        return (int)hipGetLastError();
    }
#define DUDF_W(SW, FL) do { if (h16) DUDF_GO_W16(SW, FL); else DUDF_GO_W(SW, FL); } while (0)
    // The stash array a layer's outputs travel through is written in every variant (wide_in): the forward sweeps always
    // store h_l (a value-only query: nothing else), the query variants of the reverse sweeps park q_l in S.
    switch (which) {
        case SWEEP_FWD: if (a.store_c) DUDF_W(SWEEP_FWD, 3); else DUDF_W(SWEEP_FWD, 1); break;
        case SWEEP_REV: if (a.train) DUDF_W(SWEEP_REV, 1); else DUDF_W(SWEEP_REV, 0); break;
        case SWEEP_ADJ_FWD: DUDF_W(SWEEP_ADJ_FWD, 0); break;
        case SWEEP_ADJ_REV: if (a.have_e) DUDF_W(SWEEP_ADJ_REV, 1); else DUDF_W(SWEEP_ADJ_REV, 0); break;
        case SWEEP_FWD_H: DUDF_W(SWEEP_FWD_H, 1); break;
        case SWEEP_REV_H: if (a.train) DUDF_W(SWEEP_REV_H, 1); else DUDF_W(SWEEP_REV_H, 0); break;
        case SWEEP_ADJ_FWD_H: DUDF_W(SWEEP_ADJ_FWD_H, 0); break;
        case SWEEP_ADJ_REV_H: DUDF_W(SWEEP_ADJ_REV_H, 0); break;
        case SWEEP_FWD_J: DUDF_W(SWEEP_FWD_J, 0); break;
        default: return DUDF_E_UNSUPPORTED;
    }
#undef DUDF_W
#undef DUDF_GO_W16
#undef DUDF_GO_W16R
#undef DUDF_GO_W
    return (int)hipGetLastError();
}

}  // namespace

// ... and with THESE stash flags; run_sweep asks before it opens the profiling scope of a launch.  (Round 2: the 512-wide kernel
// was built for the training variants only; now every variant is.)
bool dudf_sweep_bf16_handles(int which, int H, int L, const SweepArgs& a) {
    (void)a;
    return dudf_sweep_bf16_supported(which, H, L);
}

bool dudf_sweep_bf16_supported(int which, int H, int L) {
    return (H == 512 || H == 256 || H == 128) && L >= 2 && which >= SWEEP_FWD && which <= SWEEP_FWD_J;
}

namespace {
// Estimated duration of one part of a pair launch, in full plain-column passes: every workgroup walks ceil(ng / nb) groups of 16
// columns in passes of 8 (one per wave); a partial pass with up to one wave per SIMD costs about half a full one (measured: a
// lone wave 0.68, the passes are bound by latency, not by issue), more waves in proportion.  `c` = a pass of this variant
// against a pass of the plain fp16x3 sweep (quads on bf16x6: 1.3-1.65 measured at 29 970 points; 1.5 here).
double pair_est(int ng, int nb, double c) {
    const int m = (ng + nb - 1) / nb, full = m / NWB, rem = m % NWB;
    const double part = rem == 0 ? 0.0 : (rem <= NWB / 2 ? 0.55 : 0.55 + 0.45 * (rem - NWB / 2) / (NWB / 2));
    return c * (full + part);
}
constexpr size_t kPairSmemQ = 3 * GeoB<256, 0>::CHUNKB + kMaxAmaxLayers * sizeof(unsigned);                    // the quad body's LDS
constexpr size_t kPairSmemP = 3 * GeoB<256, 1>::CHUNKB + kMaxLdsBiasLayers * 256 * sizeof(float) + kOctBytes;     // the plain body's, at most
constexpr size_t kPairSmemMax = kPairSmemQ > kPairSmemP ? kPairSmemQ : kPairSmemP;
static_assert(kPairSmemMax <= 160 * 1024, "LDS of a CU");
template <int SWQ, int FLQ, int SWP, int FLP, int SPQ, int P24 = 0>
int launch_pair_t(const SweepArgs& aq, const SweepArgs& ap, size_t smem, int nbq, int nbp, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_pair_kernel<256, SWQ, FLQ, SWP, FLP, SPQ, P24>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLdsCu);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((sweep_pair_kernel<256, SWQ, FLQ, SWP, FLP, SPQ, P24>), dim3(nbq + nbp), dim3(GeoB<256>::NTHR), smem, st, aq, ap, nbq);
    return (int)hipGetLastError();
}
}  // namespace

// One launch for the quad columns (variant base + 4, bf16x6) AND the plain columns (variant base, fp16x3) of a training sweep
// at H = 256; DUDF_E_UNSUPPORTED when the combination has no pair kernel (the caller then launches them one after the other).
// Option pair_launch = 0 switches it off (A/B).
int dudf_launch_sweep_pair(int base, int H, const SweepArgs& aq0, const SweepArgs& ap0, hipStream_t st) {
    if (!dudf_opt_pair_launch() || H != 256 || base < SWEEP_FWD || base > SWEEP_ADJ_REV || aq0.ntiles <= 0 || ap0.ntiles <= 0) return DUDF_E_UNSUPPORTED;
    if (!((ap0.split >> base) & 1) || ap0.L < 2 || ap0.L > kMaxLdsBiasLayers) return DUDF_E_UNSUPPORTED;
    // the training variants only (a query has no plain columns beside its quads)
    if (base == SWEEP_FWD && !(aq0.store_s && ap0.store_s && ap0.store_c)) return DUDF_E_UNSUPPORTED;
    if (base == SWEEP_REV && !(aq0.train && ap0.train)) return DUDF_E_UNSUPPORTED;
    if (base == SWEEP_ADJ_REV && !(ap0.have_e && ap0.ebound && ((ap0.split >> SWEEP_ADJ_FWD) & 1))) return DUDF_E_UNSUPPORTED;
    const int ngq = aq0.ntiles * (TILE / 16), ngp = ap0.ntiles * (TILE / 16);
    const int tq = (ngq + NWB - 1) / NWB, tp = (ngp + NWB - 1) / NWB;          // passes of 8 groups
    int nbq, nbp;
    if (tq + tp <= 256) { nbq = tq; nbp = tp; }
    else {
        double best = 1e30; nbq = 128;
        for (int q = 1; q < 256; ++q) {
            const double eq = pair_est(ngq, q, 1.5), ep = pair_est(ngp, 256 - q, 1.0);
            const double t = (eq > ep ? eq : ep) + 1e-3 * (eq > ep ? eq - ep : ep - eq);
            if (t < best) { best = t; nbq = q; }
        }
        nbp = 256 - nbq;
    }
    DudfProfScope prof(PROF_SWEEP_FWD + base, st);
    dudf_note_products(PROF_SWEEP_FWD + base, 3);        // the plain columns of a pair launch are always fp16x3
    SweepArgs aq = aq0, ap = ap0;
    aq.clk = nullptr;
    ap.clk = dudf_prof_clk(PROF_SWEEP_FWD + base);
    constexpr size_t w3 = 3 * GeoB<256, 1>::CHUNKB;
    const size_t sp = w3 + (base == SWEEP_FWD ? (size_t)ap.L * 256 * sizeof(float) : kMaxAmaxLayers * sizeof(unsigned)) + kOctBytes;
    const bool q16 = (aq.split & 32) && aq.zbound && aq.ebound;       // the quads on fp16x3 as well (their LDS is then the smaller part)
    const size_t sq = q16 ? w3 + (base == SWEEP_FWD ? (size_t)aq.L * 256 * sizeof(float) : 0) + kMaxAmaxLayers * sizeof(unsigned) : kPairSmemQ;
    const size_t smem = sq > sp ? sq : sp;
    if (ap.p24 != aq.p24 || (ap.p24 && !q16)) return DUDF_E_UNSUPPORTED;   // (the 24-bit stash needs the quads on fp16x3 too: dudf_stash_p24_enabled)
    if (ap.p24 == 7) switch (base) {
        case SWEEP_FWD: return launch_pair_t<SWEEP_FWD_H, 1, SWEEP_FWD, 3, 1, 7>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_REV: return launch_pair_t<SWEEP_REV_H, 1, SWEEP_REV, 1, 1, 7>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_ADJ_FWD: return launch_pair_t<SWEEP_ADJ_FWD_H, 0, SWEEP_ADJ_FWD, 0, 1, 7>(aq, ap, smem, nbq, nbp, st);
        default: return launch_pair_t<SWEEP_ADJ_REV_H, 0, SWEEP_ADJ_REV, 1, 1, 7>(aq, ap, smem, nbq, nbp, st);
    }
    if (ap.p24 == 6) switch (base) {
        case SWEEP_FWD: return launch_pair_t<SWEEP_FWD_H, 1, SWEEP_FWD, 3, 1, 6>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_REV: return launch_pair_t<SWEEP_REV_H, 1, SWEEP_REV, 1, 1, 6>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_ADJ_FWD: return launch_pair_t<SWEEP_ADJ_FWD_H, 0, SWEEP_ADJ_FWD, 0, 1, 6>(aq, ap, smem, nbq, nbp, st);
        default: return launch_pair_t<SWEEP_ADJ_REV_H, 0, SWEEP_ADJ_REV, 1, 1, 6>(aq, ap, smem, nbq, nbp, st);
    }
    if (ap.p24) return DUDF_E_UNSUPPORTED;
    if (q16) switch (base) {
        case SWEEP_FWD: return launch_pair_t<SWEEP_FWD_H, 1, SWEEP_FWD, 3, 1>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_REV: return launch_pair_t<SWEEP_REV_H, 1, SWEEP_REV, 1, 1>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_ADJ_FWD: return launch_pair_t<SWEEP_ADJ_FWD_H, 0, SWEEP_ADJ_FWD, 0, 1>(aq, ap, smem, nbq, nbp, st);
        default: return launch_pair_t<SWEEP_ADJ_REV_H, 0, SWEEP_ADJ_REV, 1, 1>(aq, ap, smem, nbq, nbp, st);
    }
    switch (base) {
        case SWEEP_FWD: return launch_pair_t<SWEEP_FWD_H, 1, SWEEP_FWD, 3, 0>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_REV: return launch_pair_t<SWEEP_REV_H, 1, SWEEP_REV, 1, 0>(aq, ap, smem, nbq, nbp, st);
        case SWEEP_ADJ_FWD: return launch_pair_t<SWEEP_ADJ_FWD_H, 0, SWEEP_ADJ_FWD, 0, 0>(aq, ap, smem, nbq, nbp, st);
        default: return launch_pair_t<SWEEP_ADJ_REV_H, 0, SWEEP_ADJ_REV, 1, 0>(aq, ap, smem, nbq, nbp, st);
    }
}

int dudf_launch_sweep_bf16(int which, int H, const SweepArgs& a0, hipStream_t st) {
    DudfProfScope prof(PROF_SWEEP_FWD + (which & 3), st);
    SweepArgs a = a0;
    a.clk = (which <= SWEEP_ADJ_REV) ? dudf_prof_clk(PROF_SWEEP_FWD + which) : nullptr;   // plain columns only
    switch (H) {
        case 256: return launch_b<256>(which, a, st);
        case 128: return launch_b<128>(which, a, st);
        case 512: return launch_w(which, a, st);
        default: return DUDF_E_UNSUPPORTED;
    }
}

namespace {
template <int H>
int pack_b(const DudfLayout& lo, const float* theta, float* ws, hipStream_t st) {
    using G = GeoB<H>;
    char* img_f = reinterpret_cast<char*>(ws + lo.ws_wimg);
    char* img_t = img_f + (size_t)(lo.L - 1) * G::IMGB;
    const int64_t total = (int64_t)2 * (lo.L - 1) * G::NKB * G::NT * 64;
    hipLaunchKernelGGL(pack_bf16_kernel<H>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, theta, img_f, img_t,
                       lo.L - 1, lo.off_hid, lo.hid_stride);
    if (dudf_split_fp16()) {
        using G1 = GeoB<H, 1>;
        char* i16_f = reinterpret_cast<char*>(ws + lo.ws_wimg16);
        char* i16_t = i16_f + (size_t)(lo.L - 1) * G1::IMGB;
        hipLaunchKernelGGL(pack_f16_kernel<H>, dim3(H >= 256 ? 16 : 4, lo.L - 1), dim3(256), 0, st, theta, i16_f, i16_t,
                           ws + lo.ws_wsc, lo.L - 1, lo.off_hid, lo.hid_stride);
    }
    return (int)hipGetLastError();
}
}  // namespace

namespace {
template <int H>
int prep_b(const DudfLayout& lo, const float* theta, const float* x, float* ws, int need, hipStream_t st) {
    PrepArgs a;
    a.theta = theta; a.x = x;
    a.w1b = ws + lo.ws_w1b; a.b1s = ws + lo.ws_b1s; a.rho = lo.rho; a.w1t16 = ws + lo.ws_w1t16; a.wt = ws + lo.ws_wt; a.x4 = ws + lo.ws_x4; a.wsc = ws + lo.ws_wsc;
    a.img_f = reinterpret_cast<char*>(ws + lo.ws_wimg); a.img_t = a.img_f + (size_t)(lo.L - 1) * GeoB<H>::IMGB;
    a.img16_f = reinterpret_cast<char*>(ws + lo.ws_wimg16); a.img16_t = a.img16_f + (size_t)(lo.L - 1) * GeoB<H, 1>::IMGB;
    a.zero = reinterpret_cast<unsigned*>(ws + lo.ws_acc); a.nzero = 2 * DUDF_NACC;
    a.zero2 = reinterpret_cast<unsigned*>(ws + lo.ws_amax); a.nzero2 = 4 * lo.L;
    a.L = lo.L; a.off_hid = lo.off_hid; a.hid_stride = lo.hid_stride;
    a.n = lo.n; a.n_h = lo.n_h; a.ncol_h = lo.ncol_h; a.np = lo.np;
    const int nhid = lo.L - 1;
    a.nsub = H >= 256 ? 16 : 4;
    a.nb_f16 = (dudf_split_fp16() && nhid > 0) ? nhid * a.nsub : 0;
    a.nb_bf16 = ((need & 1) && nhid > 0) ? (int)(((int64_t)2 * nhid * GeoB<H>::NKB * GeoB<H>::NT * 64 + 255) / 256) : 0;
    a.nb_x4 = x ? (int)((lo.np + 255) / 256 < 1024 ? (lo.np + 255) / 256 : 1024) : 0;
    a.nb_thin = 2;
    a.nb_wt = ((need & 2) && nhid > 0) ? (int)(((int64_t)nhid * H * H + 255) / 256 < 2048 ? ((int64_t)nhid * H * H + 255) / 256 : 2048) : 0;
    const int grid = a.nb_f16 + a.nb_bf16 + a.nb_x4 + a.nb_thin + a.nb_wt;
    hipLaunchKernelGGL(prep_kernel<H>, dim3(grid), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}
}  // namespace

// Returns DUDF_E_UNSUPPORTED for widths without 16-bit weight images (the caller then packs with the separate kernels).
int dudf_launch_prep(const DudfLayout& lo, const float* theta, const float* x, float* ws, int need, hipStream_t st) {
    DudfProfScope prof(PROF_PACK, st);
    if (lo.H == 256) return prep_b<256>(lo, theta, x, ws, need, st);
    if (lo.H == 128) return prep_b<128>(lo, theta, x, ws, need, st);
    if (lo.H == 512) return prep_b<512>(lo, theta, x, ws, need, st);
    return DUDF_E_UNSUPPORTED;
}

int dudf_launch_pack_bf16(const DudfLayout& lo, const float* theta, float* ws, hipStream_t st) {
    if (lo.L < 2) return 0;
    DudfProfScope prof(PROF_PACK, st);
    if (lo.H == 256) return pack_b<256>(lo, theta, ws, st);
    if (lo.H == 128) return pack_b<128>(lo, theta, ws, st);
    if (lo.H == 512) return pack_b<512>(lo, theta, ws, st);
    return 0;
}

#if DUDF_FX_CHECK
// debug build only (not part of the C ABI): granules the fixed-point packers would have wrapped since the last reset — [0] S/Q/A/Z, [1] C
extern "C" int dudf_dbg_fx_violations(unsigned* out2, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_dudf_fx_bad), 2 * sizeof(unsigned));
    if (e == hipSuccess && reset) { const unsigned z[2] = {0u, 0u}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_dudf_fx_bad), z, sizeof(z)); }
    return (int)e;
}
#endif
