// Branch-free fp32 sin/cos pair used by every SIREN sweep (device) and by the host accuracy test.
//
// Replaces `torch.sin(self.w0 * x)` (reference src/model.py:29-30) and the cos/-sin chains its
// autograd graph expands to.  SIREN arguments reach |w0*z| ~ 50 rad in the first layer, so the
// hardware v_sin_f32 (input in revolutions, fp32 pre-multiply) is not accurate enough; this is a
// two-constant Cody-Waite reduction by pi/2 done with FMAs (exact first step for |x| < 2^20)
// followed by the classic single-precision minimax polynomials on [-pi/4, pi/4].
// Measured vs fp64 (tests/test_host_math.py): max ABSOLUTE error 9.4e-8 for |x| <= 1e4 (about
// 1.5 ulp of 1.0); no Payne-Hanek path, so accuracy degrades gracefully beyond |x| ~ 1e5.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DUDF_HD __host__ __device__ __forceinline__
#else
#define DUDF_HD static inline
#endif

// Quadrant n (mod 4) applied to (sin r, cos r): odd n swaps the pair, bit 1 of n negates the sine, bit 1 of n + 1 the
// cosine.  Written on the bit patterns (exact: selections and sign flips only) so that the device code is a bit-field
// select per output and one three-input logic op for the sign (v_bfe_i32, 2 v_bfi_b32, v_lshlrev, v_add, 2 v_bitop3 on
// gfx950: 7 instructions per value instead of 12 compares / conditional moves).
DUDF_HD void dudf_quadrant(int n, float sr, float cr, float* s_out, float* c_out) {
    union { float f; unsigned u; } a, b, so, co;
    a.f = sr; b.f = cr;
    const unsigned t = (unsigned)n << 30;                        // bit 31 = bit 1 of n
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned odd = (unsigned)__builtin_amdgcn_sbfe(n, 0, 1);                       // all ones if n is odd
    const unsigned sa = __builtin_amdgcn_bitop3_b32(odd, b.u, a.u, 0xCA);                // (odd & b) | (~odd & a)
    const unsigned ca = __builtin_amdgcn_bitop3_b32(odd, a.u, b.u, 0xCA);
    so.u = __builtin_amdgcn_bitop3_b32(t, 0x80000000u, sa, 0x6A);                        // (t & sign) ^ sa
    co.u = __builtin_amdgcn_bitop3_b32(t + 0x40000000u, 0x80000000u, ca, 0x6A);          // bit 1 of n + 1
#else
    const unsigned odd = (unsigned)-(n & 1);
    const unsigned sa = (b.u & odd) | (a.u & ~odd);
    const unsigned ca = (a.u & odd) | (b.u & ~odd);
    so.u = sa ^ (t & 0x80000000u);
    co.u = ca ^ ((t + 0x40000000u) & 0x80000000u);
#endif
    *s_out = so.f; *c_out = co.f;
}

DUDF_HD void dudf_sincos(float x, float* s_out, float* c_out) {
    const float k = rintf(x * 0.636619772367581343f);            // nearest multiple of pi/2
    float r = fmaf(-k, 1.57079637050628662109375f, x);           // x - k*fl(pi/2): exact
    r = fmaf(-k, -4.37113900018624283e-8f, r);                   // - k*(pi/2 - fl(pi/2))
    const int n = (int)k;
    const float r2 = r * r;
    float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(r2, ps, -1.6666654611e-1f);
    const float sr = fmaf(r * r2, ps, r);                        // sin(r)
    float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(r2, pc, 4.166664568298827e-2f);
    const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));   // cos(r)
    dudf_quadrant(n, sr, cr, s_out, c_out);
}

