// Pieces shared by the two sweep kernels (dudf_sweep.hip: f32-input MFMA; dudf_sweep_bf16.hip: bf16x6 MFMA): stash
// addressing, the elementwise tails of every sweep variant, and the vector-memory bookkeeping of a chunk.
// Both kernels produce the 16x16 accumulator tile in the same register layout (row = 4*(lane>>4)+reg = feature,
// col = lane&15 = column), so the tails are identical.
#pragma once
#include "dudf_internal.h"
#include "dudf_math.h"

namespace {

constexpr int NW = 4;                             // waves per workgroup (one per SIMD)
constexpr int TILE = NW * 16;                     // columns per workgroup pass

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// dudf_sincos (dudf_math.h) on two values at once: the same operations in the same order — so the same bits — but the
// FMA/MUL chains are written on float2 so that hipcc emits the packed v_pk_fma_f32 / v_pk_mul_f32 (two lanes-worth of
// work per issue slot).  The forward sweep is bound by vector-ALU issue once its matmuls run on the bf16 cores.
typedef float dudf_f2 __attribute__((ext_vector_type(2)));
typedef int dudf_i2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dudf_sincos2(dudf_f2 x, dudf_f2& s_out, dudf_f2& c_out) {
#if DUDF_SWEEP_DBG & 64
    s_out = x; c_out = x * 0.5f; return;
#endif
    const dudf_f2 k = {rintf(x.x * 0.636619772367581343f), rintf(x.y * 0.636619772367581343f)};
    dudf_f2 r = __builtin_elementwise_fma(-k, (dudf_f2)(1.57079637050628662109375f), x);
    r = __builtin_elementwise_fma(-k, (dudf_f2)(-4.37113900018624283e-8f), r);
    const dudf_i2 n = {(int)k.x, (int)k.y};
    const dudf_f2 r2 = r * r;
    dudf_f2 ps = __builtin_elementwise_fma(r2, (dudf_f2)(-1.9515295891e-4f), (dudf_f2)(8.3321608736e-3f));
    ps = __builtin_elementwise_fma(r2, ps, (dudf_f2)(-1.6666654611e-1f));
    const dudf_f2 sr = __builtin_elementwise_fma(r * r2, ps, r);
    dudf_f2 pc = __builtin_elementwise_fma(r2, (dudf_f2)(2.443315711809948e-5f), (dudf_f2)(-1.388731625493765e-3f));
    pc = __builtin_elementwise_fma(r2, pc, (dudf_f2)(4.166664568298827e-2f));
    const dudf_f2 cr = __builtin_elementwise_fma(r2 * r2, pc, __builtin_elementwise_fma(r2, (dudf_f2)(-0.5f), (dudf_f2)(1.0f)));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float so, co;
        dudf_quadrant(n[i], sr[i], cr[i], &so, &co);
        s_out[i] = so; c_out[i] = co;
    }
}

// Stash addressing: `ub` is a WAVE-UNIFORM float offset (layer and tile folded in, lives in SGPRs),
// `vo` the lane's 32-bit float offset ((quarter*np + point)*4): global_load/store take the saddr form
// and no per-tile 64-bit address is kept in VGPRs.
// (`vo` in BYTES: "uniform base + zero-extended 32-bit lane offset" is what hipcc turns into the scalar-base addressing
//  mode; with a float offset it builds a 64-bit address per lane and access instead)
#define DUDF_AT(arr, ub, vo) reinterpret_cast<f32x4*>(reinterpret_cast<char*>((arr) + (ub)) + (vo))
#define DUDF_CAT(arr, ub, vo) reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>((arr) + (ub)) + (vo))
// the stash is a stream (written once, read once or twice, 14 GB per step): non-temporal accesses
#ifndef DUDF_NT_LD
#define DUDF_NT_LD 1
#endif
#ifndef DUDF_NT_ST
#define DUDF_NT_ST 1
#endif
#ifndef DUDF_SWEEP_DBG
#define DUDF_SWEEP_DBG 0    // timing experiments only (wrong results): 1 no stash stores, 2 no stash loads, 4 no MFMA (bf16 kernel),
#endif                      // 8 no weight DMA, 16 no k-block barrier, 32 no LDS fragment reads, 64 no sin/cos
#if DUDF_SWEEP_DBG & 1
#define DUDF_ST(arr, ub, vo, val) asm volatile("" :: "v"((f32x4)(val)))
#else
#if DUDF_NT_ST
#define DUDF_ST(arr, ub, vo, val) __builtin_nontemporal_store((f32x4)(val), DUDF_AT(arr, ub, vo))
#else
#define DUDF_ST(arr, ub, vo, val) (*DUDF_AT(arr, ub, vo) = (f32x4)(val))
#endif
#endif
#if DUDF_SWEEP_DBG & 2
__device__ __forceinline__ f32x4 dudf_dbg_any() { f32x4 z; asm volatile("" : "=v"(z)); return z; }
#define DUDF_LD(arr, ub, vo) dudf_dbg_any()
#else
#if DUDF_NT_LD
#define DUDF_LD(arr, ub, vo) __builtin_nontemporal_load(DUDF_CAT(arr, ub, vo))
#else
#define DUDF_LD(arr, ub, vo) (*DUDF_CAT(arr, ub, vo))
#endif
#endif

// ---- 24-bit tile-major stash arrays (dudf_internal.h, "p24") -----------------------------------------------------------------
// four fp32 values -> three dwords: round to nearest at bit 8 (an integer add on the bit pattern: a carry out of the mantissa
// lands in the exponent, as it should), then the top three bytes of each value, packed by v_perm_b32 (selector bytes 0-3 pick
// from the SECOND source, 4-7 from the first, 0x0c is the constant 0).  7 vector-ALU instructions per tile and array to write,
// 4 to read — beside an HBM-bound sweep.
typedef unsigned dudf_u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ dudf_u3 p24_pack(const f32x4 v) {
    const unsigned u0 = __float_as_uint(v[0]) + 0x80u, u1 = __float_as_uint(v[1]) + 0x80u,
                   u2 = __float_as_uint(v[2]) + 0x80u, u3 = __float_as_uint(v[3]) + 0x80u;
    dudf_u3 d;
    d.x = __builtin_amdgcn_perm(u1, u0, 0x05030201u);
    d.y = __builtin_amdgcn_perm(u2, u1, 0x06050302u);
    d.z = __builtin_amdgcn_perm(u3, u2, 0x07060503u);
    return d;
}
__device__ __forceinline__ f32x4 p24_unpack(const dudf_u3 d) {
    return f32x4{__uint_as_float(d.x << 8), __uint_as_float(__builtin_amdgcn_perm(d.y, d.x, 0x0504030cu)),
                 __uint_as_float(__builtin_amdgcn_perm(d.z, d.y, 0x0403020cu)), __uint_as_float(d.z & 0xffffff00u)};
}
// `ub` is the same wave-uniform float offset as for the fp32 arrays ((layer * H + 16 T) * np): the tile-major byte offset of
// (layer, T) is exactly 3 * ub; `vt` = the lane's ((p / 16) * 64 + lane) * 12
#define DUDF_AT24(arr, ub, vt) reinterpret_cast<dudf_u3*>(reinterpret_cast<char*>(arr) + 3 * (ub) + (vt))
#define DUDF_CAT24(arr, ub, vt) reinterpret_cast<const dudf_u3*>(reinterpret_cast<const char*>(arr) + 3 * (ub) + (vt))
#if DUDF_SWEEP_DBG & 1
#define DUDF_ST24(arr, ub, vt, val) asm volatile("" :: "v"(p24_pack((f32x4)(val))))
#else
#define DUDF_ST24(arr, ub, vt, val) __builtin_nontemporal_store(p24_pack((f32x4)(val)), DUDF_AT24(arr, ub, vt))
#endif
#if DUDF_SWEEP_DBG & 2
__device__ __forceinline__ dudf_u3 dudf_dbg_any3() { dudf_u3 z; asm volatile("" : "=v"(z)); return z; }
#define DUDF_LD24(arr, ub, vt) p24_unpack(dudf_dbg_any3())
#else
#define DUDF_LD24(arr, ub, vt) p24_unpack(__builtin_nontemporal_load(DUDF_CAT24(arr, ub, vt)))
#endif
// default cache policy (no `nt`): the 512-wide kernel reads an array it has just written back as the next layer's operand
// (sweep_tile_w: the "relay"); with the hint the line has left L2 by then (config 3: -2.3 % step time without it on the relay)
#if DUDF_SWEEP_DBG & 1
#define DUDF_ST_CACHED(arr, ub, vo, val) asm volatile("" :: "v"((f32x4)(val)))
#else
#define DUDF_ST_CACHED(arr, ub, vo, val) (*DUDF_AT(arr, ub, vo) = (f32x4)(val))
#endif
// The three dwords of a 24-bit granule as they come from memory (in .xyz of an f32x4): epilogue_loads must not unpack — the
// operands are requested two steps ahead between scheduling fences, and an unpack next to the load makes the wave wait for the
// data right there (measured: the adjoint reverse sweep got 9 % SLOWER reading 25 % fewer bytes); epilogue() unpacks at use.
#if DUDF_SWEEP_DBG & 2
#define DUDF_LD24RAW(arr, ub, vt) dudf_dbg_any()
#else
// (the constant fourth lane costs nothing: it folds; an UNDEFINED lane made hipcc spill three values inside the adjoint forward
//  sweep's k-block step instead of one)
__device__ __forceinline__ f32x4 dudf_raw3(const dudf_u3 d) { return f32x4{__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z), 0.f}; }
#define DUDF_LD24RAW(arr, ub, vt) dudf_raw3(__builtin_nontemporal_load(DUDF_CAT24(arr, ub, vt)))
#endif
__device__ __forceinline__ f32x4 p24_unpack_raw(const f32x4 r) { return p24_unpack(dudf_u3{__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2])}); }
// ---- C = cos(w0 z_l) as 24-bit FIXED POINT (DudfLayout::p24 bit 2; same tile-major granules) ------------------------------------
// |c| <= 1 needs no exponent: t = c + 3 lies in [2, 4], one binade, so the float add itself rounds c to nearest on a 2^-22 grid
// and the low 24 bits of t's pattern are the unsigned integer (c + 1) 2^22 in [0, 2^23] (t = 4 carries into the exponent's low
// bit, which is bit 23: still the right integer).  One v_add_f32 per value to write; to read, the top byte 0x40 comes back
// (t's patterns are 0x40000000 | u) and c = t - 3 is exact.  Absolute error <= 2^-23 = 1.2e-7 — the size of the polynomial
// pair's own error (9.4e-8) — where a 24-bit FLOAT (the "p24" arrays) would lose 2^-17 of every value; the 12-step beetle
// trajectory does not see it (profiles/r04_cround.txt: 2^-22 and 2^-23 grids both within 5e-7, like fp32).
// Debug build (-DDUDF_FX_CHECK=1, tools/fx_check.py): count the granules whose values the two fixed-point packers would WRAP
// (a value outside [-1, 1] of its grid reads back as ~ +5) — the precondition `set_scale`'s bound and |cos| <= 1 have to guarantee.  The counter is
// read by dudf_dbg_fx_violations (dudf_sweep_bf16.hip, debug build only); the branch is never taken unless the invariant breaks.
#ifndef DUDF_FX_CHECK
#define DUDF_FX_CHECK 0
#endif
#ifndef DUDF_FX_HEADROOM
#define DUDF_FX_HEADROOM 1.0009765625f      // factor on the column bound before its exponent is taken (set_scale); a debug build with < 1 is the positive control of the counters
#endif
#if DUDF_FX_CHECK
static __device__ unsigned g_dudf_fx_bad[2];          // [0] fx24_pack: |v fs| > 1, [1] c24_pack: |c| > 1
#define DUDF_FX_COUNT(i, cond) do { if (cond) atomicAdd(&g_dudf_fx_bad[i], 1u); } while (0)
#else
#define DUDF_FX_COUNT(i, cond) do { } while (0)
#endif
__device__ __forceinline__ dudf_u3 c24_pack(const f32x4 c) {
    DUDF_FX_COUNT(1, fabsf(c[0]) > 1.f || fabsf(c[1]) > 1.f || fabsf(c[2]) > 1.f || fabsf(c[3]) > 1.f);
    const unsigned u0 = __float_as_uint(c[0] + 3.0f), u1 = __float_as_uint(c[1] + 3.0f),
                   u2 = __float_as_uint(c[2] + 3.0f), u3 = __float_as_uint(c[3] + 3.0f);
    // values 0-2 in the low three bytes of a dword each, value 3 spread over the three top bytes: three of four values come back
    // with one bit-field insert, and the read side needs three constants instead of six.  (Packed back to back like the 24-bit
    // floats, the reverse sweep — at 256 registers — reloaded two spilled constants in every k-block step.)
    dudf_u3 d;
    d.x = __builtin_amdgcn_perm(u3, u0, 0x04020100u);
    d.y = __builtin_amdgcn_perm(u3, u1, 0x05020100u);
    d.z = __builtin_amdgcn_perm(u3, u2, 0x06020100u);
    return d;
}
__device__ __forceinline__ f32x4 c24_unpack(const dudf_u3 d) {
    const unsigned m = 0x00ffffffu, two = 0x40000000u;          // (the pattern of 2.0f: an inline constant)
    const unsigned t0 = (d.x & m) | two, t1 = (d.y & m) | two, t2 = (d.z & m) | two;     // v_and_or_b32
    const unsigned y = __builtin_amdgcn_perm(d.y, d.x, 0x0c0c0703u);                    // [d0.b3, d1.b3, 0, 0]
    const unsigned t3 = __builtin_amdgcn_perm(d.z, y, 0x0c070100u) | two;               // [.., .., d2.b3, 0] | 0x40 on top
    return f32x4{__uint_as_float(t0) - 3.0f, __uint_as_float(t1) - 3.0f, __uint_as_float(t2) - 3.0f, __uint_as_float(t3) - 3.0f};
}
// ---- S, Q, A, Z as 24-bit FIXED POINT relative to the column's bound (DudfLayout::p24 bit 0; same granules as C) --------------------
// The fp16x3 sweeps fix, per layer and column, a power of two 2^E above everything the tail of that column can produce (set_scale:
// it scales the B operand of the next matrix with it); fs = 2^-E brings the column's values into (-1, 1), and the same float-add
// trick as for C — t = fma(v, fs, 3) in (2, 4), low 24 bits of its pattern — stores them on a 2^-22 grid of that range: absolute
// error <= 2^(E-23), i.e. a 23-bit significand for the column's largest values and proportionally fewer bits for its small ones —
// which is how the weight-gradient GEMM weighs them anyway (it sums products over columns).  Against the 24-bit FLOAT of rounds
// 4 (2^-17 of every value) the noise on a column's dominant entries is 2^-6 of that.  The reader multiplies t - 3 by 2^E, which
// the sweep leaves per layer and column in a side array (SweepArgs::fxs).  |h| = |sin| <= 1 in the plain columns' forward sweep: fs = 1.
__device__ __forceinline__ dudf_u3 fx24_pack(const f32x4 v, const float fs) {
    DUDF_FX_COUNT(0, fabsf(v[0] * fs) > 1.f || fabsf(v[1] * fs) > 1.f || fabsf(v[2] * fs) > 1.f || fabsf(v[3] * fs) > 1.f);
    const unsigned u0 = __float_as_uint(__builtin_fmaf(v[0], fs, 3.0f)), u1 = __float_as_uint(__builtin_fmaf(v[1], fs, 3.0f)),
                   u2 = __float_as_uint(__builtin_fmaf(v[2], fs, 3.0f)), u3 = __float_as_uint(__builtin_fmaf(v[3], fs, 3.0f));
    dudf_u3 d;
    d.x = __builtin_amdgcn_perm(u3, u0, 0x04020100u);
    d.y = __builtin_amdgcn_perm(u3, u1, 0x05020100u);
    d.z = __builtin_amdgcn_perm(u3, u2, 0x06020100u);
    return d;
}
#ifndef DUDF_FX_DBG
#define DUDF_FX_DBG 0      // timing experiments only (wrong results): 1 = no column-scale side-array stores, 2 = the 24-bit FLOAT pack of round 4 instead of the fixed-point pack
#endif
#if DUDF_SWEEP_DBG & 1
#define DUDF_STF24(arr, ub, vt, val, fs) asm volatile("" :: "v"(fx24_pack((f32x4)(val), fs)))
#elif DUDF_FX_DBG & 2
#define DUDF_STF24(arr, ub, vt, val, fs) __builtin_nontemporal_store(p24_pack((f32x4)(val)), DUDF_AT24(arr, ub, vt))
#else
#define DUDF_STF24(arr, ub, vt, val, fs) __builtin_nontemporal_store(fx24_pack((f32x4)(val), fs), DUDF_AT24(arr, ub, vt))
#endif
__device__ __forceinline__ f32x4 c24_unpack_raw(const f32x4 r) { return c24_unpack(dudf_u3{__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2])}); }
#if DUDF_SWEEP_DBG & 1
#define DUDF_STC24(arr, ub, vt, val) asm volatile("" :: "v"(c24_pack((f32x4)(val))))
#else
#define DUDF_STC24(arr, ub, vt, val) __builtin_nontemporal_store(c24_pack((f32x4)(val)), DUDF_AT24(arr, ub, vt))
#endif
// a backward-only array in the format of this build: P (compile-time) = 24-bit tile-major, else fp32 rows
#define DUDF_STB(P, arr, ub, lo, val) do { if constexpr (P) DUDF_ST24(arr, ub, (lo).t, val); else DUDF_ST(arr, ub, (lo).v, val); } while (0)
// ... or, RL (compile-time): this array is the caller's relay — default cache policy
#define DUDF_STR(RL, P, arr, ub, lo, val) do { if constexpr (RL) DUDF_ST_CACHED(arr, ub, (lo).v, val); else DUDF_STB(P, arr, ub, lo, val); } while (0)
// S / Q / A / Z (the weight-gradient GEMM's operands): RL = the caller's relay (fp32, cached); P = 24-bit fixed point relative to the
// column's bound tk.fs; else fp32 rows
#define DUDF_STX(RL, P, arr, ub, lo, val) do { if constexpr (RL) DUDF_ST_CACHED(arr, ub, (lo).v, val); else if constexpr (P) DUDF_STF24(arr, ub, (lo).t, val, tk.fs); else DUDF_ST(arr, ub, (lo).v, val); } while (0)
#define DUDF_LDB(P, arr, ub, lo) ((P) ? DUDF_LD24RAW(arr, ub, (lo).t) : DUDF_LD(arr, ub, (lo).v))     // (P: raw, unpacked by epilogue())
#define DUDF_LDC(P, arr, ub, lo) ((P) ? DUDF_LD24RAW(arr, ub, (lo).ct) : DUDF_LD(arr, ub, (lo).c))    // C (fixed point when P; `c` == `v` in plain columns)

// ---- quad (4 adjacent lanes = the 4 channels of one Hessian-path point) helpers: DPP, no LDS -------------
// (the empty asm pins the DPP source to an ARCHITECTURAL vector register: in the 512-register kernels of the 512-wide
//  layers hipcc 7.2 keeps values in accumulation registers and then emits DPP moves that read them, which the assembler
//  rejects — "explicit register staging")
__device__ __forceinline__ float quad_bcast0(float v) {          // value of the quad's lane 0 (the value channel)
    asm volatile("" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x00, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {             // sum over the quad, in every lane
    asm volatile("" : "+v"(v));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    asm volatile("" : "+v"(v));
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

// Which of the wgrad operands a sweep's tail produces (row of SweepArgs::amax), or -1: the fp16x3 weight-gradient GEMM
// scales q_l, A_l, zbar_l per layer by a power of two taken from max |.| over all columns, so the tails keep a running
// maximum of what they store (2 v_max3_f32 per tile), published per layer through LDS and once per workgroup to HBM.
template <int SW, int FL>
constexpr int amax_row() {
#ifdef DUDF_DBG_NOTRACK_H                        // timing experiment only (wrong weight-gradient scales): no running maxima in the quad sweeps
    if (SW >= 4) return -1;
#endif
    return SW == SWEEP_FWD_H ? ((FL & 1) ? 3 : -1)      // h | hdot^k: the tangent channels are not bounded by 1 (plain columns: |h| <= 1)
         : (base_of(SW) == SWEEP_REV && SW != SWEEP_FWD_J) ? ((FL & 1) ? 0 : -1)
         : base_of(SW) == SWEEP_ADJ_FWD ? 1
         : base_of(SW) == SWEEP_ADJ_REV ? 2 : -1;
}
// running max |.| of the wgrad operand a tail stores (t) | of e_l (e: adjoint forward sweep) — and, in the builds that keep S, Q, A, Z
// as 24-bit fixed point (P24 bit 0), `fs`: the power of two that brings this column's stored operand into (-1, 1).  A second
// type, so that the other builds' register allocation does not see the extra field (a third float in this struct moved the
// adjoint forward sweep of the default build from 1 to 3 spills inside its k-block step: tests/test_isa_contract.py).
template <bool FX> struct TailTrackT { float t = 0.f, e = 0.f; };
template <> struct TailTrackT<true> { float t = 0.f, e = 0.f, fs = 1.f; };
typedef TailTrackT<false> TailTrack;
__device__ __forceinline__ void dudf_track(float& tmax, const f32x4 v) {
    // two v_max3_f32 with |.| source modifiers (fmaxf() would add IEEE canonicalisation instructions around every maximum:
    // +12 vector-ALU instructions per k-block step in the stash-bound sweeps)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(tmax) : "v"(v[0]), "v"(v[1]));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(tmax) : "v"(v[2]), "v"(v[3]));
}

// A lane's byte offset into a stash array: `v` for every array, `c` for C.  They differ in the Hessian-quad columns only: cos(w0 z_l)
// is the same number in all four channels of a quad (it is the VALUE channel's), so C keeps ONE copy per quad — at column
// (p >> 2) of the quad region, written and read by all four lanes (same bits, same granule: one 16-byte access serves four lanes) —
// instead of four: 3 of the quad sweeps' 16 stash units per column gone.
// `t`: the lane's byte offset inside a (layer, tile) block of a 24-bit tile-major array (0 where the build has none).
// `ct`: the same for C (its own: one granule per quad in the Hessian-quad columns, like `c`).
struct LaneOff {
    unsigned v, c, t, ct;
    __device__ __forceinline__ LaneOff(unsigned x) : v(x), c(x), t(0), ct(0) {}
    __device__ __forceinline__ LaneOff(unsigned x, unsigned y) : v(x), c(y), t(0), ct(0) {}
    __device__ __forceinline__ LaneOff(unsigned x, unsigned y, unsigned z) : v(x), c(y), t(z), ct(z) {}
    __device__ __forceinline__ LaneOff(unsigned x, unsigned y, unsigned z, unsigned w) : v(x), c(y), t(z), ct(w) {}
};

// RL: the array that carries this sweep's post-tail values (S / Q / A / Z by sweep) is stored with the default cache policy
// P24: which arrays are 24-bit tile-major in this build (DudfLayout::p24): bit 0 = S, Q, A, Z; bit 1 = R, E; bit 2 = C (fixed point)
template <int SW, int FL, bool TE = false, int P24 = 0, bool RL = false, class TK = TailTrack>
__device__ __forceinline__ f32x4 epilogue(const SweepArgs& a, f32x4 acc, f32x4 o1, f32x4 o2, f32x4 o3, int64_t ub,
                                          const LaneOff lo, bool isv, TK& tk) {
    const unsigned vo = lo.v;
    // operands that arrive as raw 24-bit granules (epilogue_loads): S in the reverse sweep, R in the adjoint forward sweeps,
    // E in the adjoint reverse sweeps
    if constexpr ((P24 & 4) != 0 && base_of(SW) != SWEEP_FWD) o1 = c24_unpack_raw(o1);      // C: every sweep behind the forward one
    if constexpr ((P24 & 1) != 0 && SW == SWEEP_REV && (FL & 1)) o2 = c24_unpack_raw(o2);   // S of the plain columns: fixed point with fs = 1
    if constexpr ((P24 & 2) != 0 && SW == SWEEP_ADJ_FWD) o2 = p24_unpack_raw(o2);
    if constexpr ((P24 & 2) != 0 && SW == SWEEP_ADJ_REV && (FL & 1)) o2 = p24_unpack_raw(o2);
    if constexpr ((P24 & 2) != 0 && (SW == SWEEP_ADJ_FWD_H || SW == SWEEP_ADJ_REV_H)) o3 = p24_unpack_raw(o3);
    f32x4 out;
    if constexpr (SW == SWEEP_FWD) {
        f32x4 s, c;
#pragma unroll
        for (int t = 0; t < 4; t += 2) {
            dudf_f2 sv, cv;
            dudf_sincos2(dudf_f2{a.w0 * acc[t], a.w0 * acc[t + 1]}, sv, cv);
            s[t] = sv.x; s[t + 1] = sv.y; c[t] = cv.x; c[t + 1] = cv.y;
        }
        if constexpr (FL & 1) DUDF_STX(RL, (P24 & 1) != 0, a.S, ub, lo, s);
        if constexpr (FL & 2) { if constexpr ((P24 & 4) != 0) DUDF_STC24(a.C, ub, lo.ct, c); else DUDF_ST(a.C, ub, vo, c); }
        out = s;
    } else if constexpr (SW == SWEEP_REV) {          // acc = a_l, o1 = c_l, o2 = s_l
        out = a.w0 * o1 * acc;                       // q_l = w0 c_l a_l
        if constexpr (FL & 1) {
            DUDF_STX(RL, (P24 & 1) != 0, a.Q, ub, lo, out);
            DUDF_STB((P24 & 2) != 0, a.R, ub, lo, (a.w0 * a.w0) * o2 * acc);   // r_l = w0^2 s_l a_l
        }
    } else if constexpr (SW == SWEEP_ADJ_FWD) {      // acc = Q_l, o1 = c_l, o2 = r_l
        out = a.w0 * o1 * acc;                       // A_l = w0 c_l Q_l
        DUDF_STX(RL, (P24 & 1) != 0, a.A, ub, lo, out);
        const f32x4 ev = o2 * acc;                   // e_l = r_l Q_l
        DUDF_STB((P24 & 2) != 0, a.E, ub, lo, ev);
        if constexpr (TE) dudf_track(tk.e, ev);      // (per column: what bounds zbar_l in the fp16x3 adjoint reverse sweep)
    } else if constexpr (SW == SWEEP_ADJ_REV) {      // acc = hbar_l, o1 = c_l, o2 = e_l
        out = a.w0 * o1 * acc - o2;                  // zbar_l
        DUDF_STX(RL, (P24 & 1) != 0, a.Z, ub, lo, out);
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
        // one copy per quad (LaneOff): the four lanes hold the same bits and write the same granule — no branch
        if constexpr ((P24 & 4) != 0) DUDF_STC24(a.C, ub, lo.ct, c); else DUDF_ST(a.C, ub, lo.c, c);
        DUDF_ST(a.ZS, ub, vo, zs);
        if constexpr (FL & 1) DUDF_STX(RL, (P24 & 1) != 0, a.S, ub, lo, out);
    } else if constexpr (SW == SWEEP_REV_H) {        // o1 = c, o2 = s|zdot^k
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float sv = quad_bcast0(o2[t]), a0 = quad_bcast0(acc[t]);
            out[t] = isv ? a.w0 * o1[t] * acc[t] : a.w0 * (o1[t] * acc[t] - a.w0 * sv * o2[t] * a0);
        }
        if constexpr (FL & 1) {
            DUDF_STX(RL, (P24 & 1) != 0, a.Q, ub, lo, out);
            DUDF_STB((P24 & 2) != 0, a.R, ub, lo, acc);
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
        DUDF_STX(RL, (P24 & 1) != 0, a.A, ub, lo, out);
        DUDF_STB((P24 & 2) != 0, a.E, ub, lo, e);
        if constexpr (TE) dudf_track(tk.e, e);
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
        DUDF_STX(RL, (P24 & 1) != 0, a.Z, ub, lo, out);
    }
    if constexpr (amax_row<SW, FL>() >= 0) dudf_track(tk.t, out);
    return out;
}

template <int SW, int FL, int P24 = 0>
__device__ __forceinline__ void epilogue_loads(const SweepArgs& a, int64_t ub, const LaneOff lo, f32x4& o1, f32x4& o2,
                                               f32x4& o3) {
    o1 = f32x4{0, 0, 0, 0}; o2 = o1; o3 = o1;
    const unsigned vo = lo.v;
    if constexpr (SW == SWEEP_REV) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        if constexpr (FL & 1) o2 = DUDF_LDB((P24 & 1) != 0, a.S, ub, lo);
    } else if constexpr (SW == SWEEP_ADJ_FWD) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        o2 = DUDF_LDB((P24 & 2) != 0, a.R, ub, lo);
    } else if constexpr (SW == SWEEP_ADJ_REV) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        if constexpr (FL & 1) o2 = DUDF_LDB((P24 & 2) != 0, a.E, ub, lo);      // no df/dx terms (loss_s2): e_l == 0
    } else if constexpr (SW == SWEEP_REV_H) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        o2 = DUDF_LD(a.ZS, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_FWD_H) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        o2 = DUDF_LD(a.ZS, ub, vo);
        o3 = DUDF_LDB((P24 & 2) != 0, a.R, ub, lo);
    } else if constexpr (SW == SWEEP_ADJ_REV_H) {
        o1 = DUDF_LDC((P24 & 4) != 0, a.C, ub, lo);
        o2 = DUDF_LD(a.ZS, ub, vo);
        o3 = DUDF_LDB((P24 & 2) != 0, a.E, ub, lo);
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

// Two finished accumulator tiles whose elementwise tail has not run yet.  The tail of chunk r-1 is executed in
// the middle of chunk r's MFMA stream (same basic block), so sin/cos, stash traffic and MFMAs overlap inside
// one wave instead of serialising at every chunk barrier.
struct Pending {
    f32x4 acc0, acc1, o1a, o2a, o3a, o1b, o2b, o3b;
    int64_t ub0, ub1;
};

}  // namespace
