// The SIREN sweeps of the DiffUDF training hot path on the f32-INPUT matrix instruction, as one kernel template
// (gfx950 / CDNA4).  The headline configuration (256-wide layers) runs on the bf16x6 kernel, dudf_sweep_bf16.hip; this
// one serves the other widths (32/64/128/512, with their Hessian quads), the third-order jets, and DUDF_SWEEP=f32.
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
//   * the 4 waves of a workgroup (64 points; two workgroups per CU, one at H = 512) walk the layers in lockstep and
//     share the weights: every 32-row chunk of W_l (or of the pre-transposed W_l^T for the reverse sweeps) is staged
//     ONCE per workgroup into LDS (LDS-DMA for H = 256/512, through registers otherwise; double buffered, one barrier
//     per chunk) and read by all waves as A operands.  A chunk completes two 16-feature output tiles per wave over the full K, so the
//     sin/cos + stash epilogue of chunk r overlaps the MFMAs of chunk r+1.
//   * what a later sweep needs (s_l, c_l, q_l, r_l/e_l, A_l, zbar_l) is stashed in HBM as one aligned
//     16-byte store per lane and tile ([layer][feature/4][point][4]); the same arrays are the operands of
//     the weight-gradient GEMM (dudf_wgrad.hip).
#include "dudf_sweep_common.h"

namespace {


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

template <int H, int N>
__device__ __forceinline__ void dma_wait() {
    if constexpr (Geo<H>::DMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}


// One sweep over one 64-column tile.  `seed` replaces the per-column operand the prologue would read from HBM when
// the caller already has it in registers (fused step kernel: gbar[q] for SWEEP_ADJ_FWD, ybar for SWEEP_ADJ_REV);
// `res` returns what the tail produced: y in [0] (SWEEP_FWD, every lane), the a_0 rows in [0..2] (SWEEP_REV, lanes < 16).
// `tmax`: running max |.| of what the tails store for the weight-gradient GEMM (dudf_sweep_common.h, amax_row) — here one
// value for all layers of the sweep (an upper bound per layer is all the fp16x3 GEMM's scale needs).
template <int H, int SW, int FL, bool SEEDED = false>
__device__ __forceinline__ void sweep_tile(const SweepArgs& a, const int tile, float* lds, unsigned& gc, float seed,
                                           f32x4& res, TailTrack& tmax) {
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
        const unsigned vo = (unsigned)(((int64_t)q * a.np + p) * 16);          // this lane's granule, in bytes
        const LaneOff vl(vo, (is_hess(SW) && !is_jet(SW)) ? (unsigned)(((int64_t)q * a.np + (p >> 2)) * 16) : vo);   // C: one copy per quad

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
                epilogue_loads<SW, FL>(a, ub, vl, o1, o2, o3);
                if constexpr (kFwdDir) {
                    acc = mfma16(a.w1b[(16 * T + li) * 4 + q], b, f32x4{0, 0, 0, 0});
                } else {
                    acc = *reinterpret_cast<const f32x4*>(a.theta + a.off_wo + 16 * T + 4 * q) * yb;
                }
                if (T < G::NT - 2) {
                    in[T] = epilogue<SW, FL>(a, acc, o1, o2, o3, ub, vl, isv, tmax);
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
                    epilogue_loads<SW, FL>(a, cur.ub0, vl, cur.o1a, cur.o2a, cur.o3a);
                    epilogue_loads<SW, FL>(a, cur.ub1, vl, cur.o1b, cur.o2b, cur.o3b);
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
                    const f32x4 e0 = epilogue<SW, FL>(a, pend.acc0, pend.o1a, pend.o2a, pend.o3a, pend.ub0, vl, isv, tmax);
                    const f32x4 e1 = epilogue<SW, FL>(a, pend.acc1, pend.o1b, pend.o2b, pend.o3b, pend.ub1, vl, isv, tmax);
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
                        __builtin_amdgcn_sched_barrier(0x76);   // VALU/SALU/VMEM may cross, LDS reads and MFMAs may not: the reads stay early
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
        in[G::NT - 2] = epilogue<SW, FL>(a, pend.acc0, pend.o1a, pend.o2a, pend.o3a, pend.ub0, vl, isv, tmax);
        in[G::NT - 1] = epilogue<SW, FL>(a, pend.acc1, pend.o1b, pend.o2b, pend.o3b, pend.ub1, vl, isv, tmax);

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
    TailTrack tmax;
    for (int tile = a.tile0 + blockIdx.x; tile < a.tile0 + a.ntiles; tile += gridDim.x)
        sweep_tile<H, SW, FL>(a, tile, lds, gc, 0.f, res, tmax);
    if constexpr (amax_row<SW, FL>() >= 0) {            // one bound for every layer of this operand (fp16x3 weight-gradient GEMM)
        __shared__ unsigned s_amax;
        if (threadIdx.x == 0) s_amax = 0u;
        __syncthreads();
        atomicMax(&s_amax, __float_as_uint(tmax.t));
        __syncthreads();
        if ((int)threadIdx.x < a.L && a.amax && s_amax) atomicMax(a.amax + amax_row<SW, FL>() * a.L + threadIdx.x, s_amax);
    }
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
        // Hessian-quad variants (all widths incl. 512: DPP sources are pinned to architectural VGPRs, dudf_sweep_common.h)
        case SWEEP_FWD_H:
            if (!a.store_s) return DUDF_E_BADMODE;
            DUDF_GO(SWEEP_FWD_H, 1);
            break;
        case SWEEP_REV_H:
            if (a.train) DUDF_GO(SWEEP_REV_H, 1); else DUDF_GO(SWEEP_REV_H, 0);
            break;
        case SWEEP_ADJ_FWD_H:
            DUDF_GO(SWEEP_ADJ_FWD_H, 0);
            break;
        case SWEEP_ADJ_REV_H:
            DUDF_GO(SWEEP_ADJ_REV_H, 0);
            break;
        case SWEEP_FWD_J:
            DUDF_GO(SWEEP_FWD_J, 0);
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
    if (which <= SWEEP_ADJ_REV) dudf_note_products(PROF_SWEEP_FWD + which, 1);
    switch (H) {
        case 32: return launch_h<32>(which, a, st);
        case 64: return launch_h<64>(which, a, st);
        case 128: return launch_h<128>(which, a, st);
        case 256: return launch_h<256>(which, a, st);
        case 512: return launch_h<512>(which, a, st);
        default: return DUDF_E_BADCFG;
    }
}
