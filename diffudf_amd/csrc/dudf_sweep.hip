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

template <int H>
struct Geo {
    static constexpr int NT = H / 16;             // 16-feature tiles per activation vector
    static constexpr int NCH = H / 32;            // 32-row weight chunks per layer
    static constexpr int LDW = H + 4;             // padded LDS row stride in floats
    static constexpr int BUF = 32 * LDW;          // floats per LDS chunk buffer
    static constexpr int NTHR = 512;
    static constexpr int F4 = 8 * H;              // float4 per chunk
    static constexpr int NSTG = (F4 + NTHR - 1) / NTHR;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int H>
__device__ __forceinline__ void stage_issue(const float* __restrict__ M, int r, f32x4 (&stg)[Geo<H>::NSTG], int tid) {
    using G = Geo<H>;
#pragma unroll
    for (int j = 0; j < G::NSTG; ++j) {
        const int f = tid + G::NTHR * j;
        if (f < G::F4) {
            const int row = f / (H / 4), c4 = f % (H / 4);
            stg[j] = *reinterpret_cast<const f32x4*>(M + (size_t)(32 * r + row) * H + 4 * c4);
        }
    }
}

template <int H>
__device__ __forceinline__ void stage_commit(float* buf, const f32x4 (&stg)[Geo<H>::NSTG], int tid) {
    using G = Geo<H>;
#pragma unroll
    for (int j = 0; j < G::NSTG; ++j) {
        const int f = tid + G::NTHR * j;
        if (f < G::F4) {
            const int row = f / (H / 4), c4 = f % (H / 4);
            *reinterpret_cast<f32x4*>(buf + row * G::LDW + 4 * c4) = stg[j];
        }
    }
}

// Elementwise tail of one 16-feature x 16-point tile.  `so` = float offset of this lane's 16-byte
// granule inside a stash array (layer, tile, quarter, point already folded in).
// Stash addressing: `ub` is a WAVE-UNIFORM float offset (layer and tile folded in, lives in SGPRs),
// `vo` the lane's 32-bit float offset ((quarter*np + point)*4): global_load/store take the saddr form
// and no per-tile 64-bit address is kept in VGPRs.
#define DUDF_AT(arr, ub, vo) reinterpret_cast<f32x4*>((arr) + (ub) + (vo))
#define DUDF_CAT(arr, ub, vo) reinterpret_cast<const f32x4*>((arr) + (ub) + (vo))

template <int SW>
__device__ __forceinline__ f32x4 epilogue(const SweepArgs& a, f32x4 acc, f32x4 o1, f32x4 o2, int64_t ub, unsigned vo) {
    f32x4 out;
    if constexpr (SW == SWEEP_FWD) {
        f32x4 s, c;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float sv, cv;
            dudf_sincos(a.w0 * acc[t], &sv, &cv);
            s[t] = sv; c[t] = cv;
        }
        if (a.store_s) *DUDF_AT(a.S, ub, vo) = s;
        if (a.store_c) *DUDF_AT(a.C, ub, vo) = c;
        out = s;
    } else if constexpr (SW == SWEEP_REV) {          // acc = a_l, o1 = c_l, o2 = s_l
        out = a.w0 * o1 * acc;                       // q_l = w0 c_l a_l
        if (a.train) {
            *DUDF_AT(a.Q, ub, vo) = out;
            *DUDF_AT(a.R, ub, vo) = (a.w0 * a.w0) * o2 * acc;   // r_l = w0^2 s_l a_l
        }
    } else if constexpr (SW == SWEEP_ADJ_FWD) {      // acc = Q_l, o1 = c_l, o2 = r_l
        out = a.w0 * o1 * acc;                       // A_l = w0 c_l Q_l
        *DUDF_AT(a.A, ub, vo) = out;
        *DUDF_AT(a.E, ub, vo) = o2 * acc;                       // e_l = r_l Q_l
    } else {                                         // acc = hbar_l, o1 = c_l, o2 = e_l
        out = a.w0 * o1 * acc - o2;                  // zbar_l
        *DUDF_AT(a.Z, ub, vo) = out;
    }
    return out;
}

template <int SW>
__device__ __forceinline__ void epilogue_loads(const SweepArgs& a, int64_t ub, unsigned vo, f32x4& o1, f32x4& o2) {
    if constexpr (SW == SWEEP_REV) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = a.train ? *DUDF_CAT(a.S, ub, vo) : f32x4{0, 0, 0, 0};
    } else if constexpr (SW == SWEEP_ADJ_FWD) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = *DUDF_CAT(a.R, ub, vo);
    } else if constexpr (SW == SWEEP_ADJ_REV) {
        o1 = *DUDF_CAT(a.C, ub, vo);
        o2 = a.have_e ? *DUDF_CAT(a.E, ub, vo) : f32x4{0, 0, 0, 0};   // no df/dx terms (loss_s2): e_l == 0
    } else {
        o1 = f32x4{0, 0, 0, 0}; o2 = o1;
    }
}

template <int H, int SW>
__global__ __launch_bounds__(512, 2) void sweep_kernel(SweepArgs a) {
    using G = Geo<H>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int ntiles = (int)(a.np / DUDF_TILE_PTS);
    const int nhid = a.L - 1;                          // hidden x hidden layers
    constexpr bool kFwdDir = (SW == SWEEP_FWD || SW == SWEEP_ADJ_FWD);

    f32x4 in[G::NT], nxt[G::NT];
    f32x4 stg[G::NSTG];
    unsigned gc = 0;                                   // running chunk counter: LDS buffer parity

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t p = (int64_t)tile * DUDF_TILE_PTS + wave * 16 + li;
        const bool valid = p < a.n;
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

        if (nhid > 0) stage_issue<H>(matrix(0), 0, stg, tid);

        // ------------------------------ prologue: fills in[] ------------------------------
        if constexpr (SW == SWEEP_FWD || SW == SWEEP_ADJ_FWD) {
            float b;
            if constexpr (SW == SWEEP_FWD) b = (q < 3) ? (valid ? a.x[p * 3 + q] : 0.f) : 1.f;   // k=3 carries the bias
            else b = (q < 3) ? a.gbar[p * 4 + q] : 0.f;                                          // A_0 = gbar
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const int64_t ub = stash_base(0, T);
                f32x4 o1, o2;
                epilogue_loads<SW>(a, ub, vo, o1, o2);
                const float w = a.w1b[(16 * T + li) * 4 + q];
                f32x4 acc = mfma16(w, b, f32x4{0, 0, 0, 0});
                in[T] = epilogue<SW>(a, acc, o1, o2, ub, vo);
            }
        } else {
            float yb = 1.f;
            if constexpr (SW == SWEEP_ADJ_REV) yb = a.ybar[p];
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const int64_t ub = stash_base(a.L - 1, T);
                f32x4 o1, o2;
                epilogue_loads<SW>(a, ub, vo, o1, o2);
                f32x4 acc = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
                in[T] = epilogue<SW>(a, acc, o1, o2, ub, vo);
            }
        }

        // ------------------------------ hidden x hidden layers ------------------------------
        for (int j = 0; j < nhid; ++j) {
            const float* M = matrix(j);
            const float* Mn = (j + 1 < nhid) ? matrix(j + 1) : nullptr;
            const int lo = out_layer(j);
            stage_commit<H>(lds + (gc & 1) * G::BUF, stg, tid);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < G::NCH; ++r) {
                if (r + 1 < G::NCH) stage_issue<H>(M, r + 1, stg, tid);
                else if (Mn) stage_issue<H>(Mn, 0, stg, tid);
                const int64_t ub0 = stash_base(lo, 2 * r), ub1 = stash_base(lo, 2 * r + 1);
                f32x4 p0, p1, p2, p3;
                epilogue_loads<SW>(a, ub0, vo, p0, p1);
                epilogue_loads<SW>(a, ub1, vo, p2, p3);
                f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                if constexpr (SW == SWEEP_FWD) {
                    const float* bias = M + (size_t)H * H;
                    acc0 = *reinterpret_cast<const f32x4*>(bias + 32 * r + 4 * q);
                    acc1 = *reinterpret_cast<const f32x4*>(bias + 32 * r + 16 + 4 * q);
                }
                const float* bp = lds + (gc & 1) * G::BUF + li * G::LDW + 4 * q;
#pragma unroll
                for (int T = 0; T < G::NT; ++T) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(bp + 16 * T);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(bp + 16 * G::LDW + 16 * T);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc0 = mfma16(a0[t], in[T][t], acc0);
                        acc1 = mfma16(a1[t], in[T][t], acc1);
                    }
                }
                nxt[2 * r] = epilogue<SW>(a, acc0, p0, p1, ub0, vo);
                nxt[2 * r + 1] = epilogue<SW>(a, acc1, p2, p3, ub1, vo);
                if (r + 1 < G::NCH) {
                    ++gc;
                    stage_commit<H>(lds + (gc & 1) * G::BUF, stg, tid);
                    __syncthreads();
                }
            }
            ++gc;
#pragma unroll
            for (int T = 0; T < G::NT; ++T) in[T] = nxt[T];
        }

        // ------------------------------ tail ------------------------------
        if constexpr (SW == SWEEP_FWD) {                // y = W_out s_L + b_out
            float part = 0.f;
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q);
                part += in[T][0] * w[0] + in[T][1] * w[1] + in[T][2] * w[2] + in[T][3] * w[3];
            }
            part += __shfl_xor(part, 16);               // the 4 lane quarters hold disjoint feature rows
            part += __shfl_xor(part, 32);
            if (q == 0) a.y[p] = part + a.theta[a.off_bo];
        } else if constexpr (SW == SWEEP_REV) {         // df/dx = W_1^T q_1 (rows 0..2 of a 16-row tile)
            f32x4 accg = {0, 0, 0, 0};
#pragma unroll
            for (int T = 0; T < G::NT; ++T) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.w1t16 + li * H + 16 * T + 4 * q);
#pragma unroll
                for (int t = 0; t < 4; ++t) accg = mfma16(w[t], in[T][t], accg);
            }
            if (q == 0) *reinterpret_cast<f32x4*>(a.g + p * 4) = f32x4{accg[0], accg[1], accg[2], 0.f};
        }
    }
}

template <int H>
int launch_h(int which, const SweepArgs& a, hipStream_t st) {
    using G = Geo<H>;
    const size_t smem = 2 * G::BUF * sizeof(float);
    const int ntiles = (int)(a.np / DUDF_TILE_PTS);
    int grid = ntiles < 256 ? ntiles : 256;
    if (grid < 1) grid = 1;
    hipError_t e = hipSuccess;
#define DUDF_GO(SW)                                                                                         \
    do {                                                                                                    \
        static bool attr_done = false;                                                                      \
        if (!attr_done) {                                                                                   \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_kernel<H, SW>),                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                 \
            if (e != hipSuccess) return (int)e;                                                             \
            attr_done = true;                                                                               \
        }                                                                                                   \
        hipLaunchKernelGGL((sweep_kernel<H, SW>), dim3(grid), dim3(G::NTHR), smem, st, a);                  \
    } while (0)
    switch (which) {
        case SWEEP_FWD: DUDF_GO(SWEEP_FWD); break;
        case SWEEP_REV: DUDF_GO(SWEEP_REV); break;
        case SWEEP_ADJ_FWD: DUDF_GO(SWEEP_ADJ_FWD); break;
        case SWEEP_ADJ_REV: DUDF_GO(SWEEP_ADJ_REV); break;
        default: return DUDF_E_BADMODE;
    }
#undef DUDF_GO
    e = hipGetLastError();
    return (int)e;
}

}  // namespace

int dudf_launch_sweep(int which, int H, const SweepArgs& a, hipStream_t st) {
    DudfProfScope prof(PROF_SWEEP_FWD + which, st);
    switch (H) {
        case 32: return launch_h<32>(which, a, st);
        case 64: return launch_h<64>(which, a, st);
        case 128: return launch_h<128>(which, a, st);
        case 256: return launch_h<256>(which, a, st);
        default: return DUDF_E_BADCFG;
    }
}
