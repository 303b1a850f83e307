// Per-step training batch on the GPU — replaces reference src/dataset.py:14-70 `sampleTrainingData`
// (open3d RaycastingScene on the CPU) for triangle meshes small enough for a brute-force distance
// (the beetle has 2 053 triangles; 20 k queries x 2 k triangles is ~2.5 GFLOP, tens of microseconds).
//
// Batch = [on-surface | far | near]  (reference :52-68):
//   on   : a random point of the precomputed surface cloud, its triangle normal, sdf = 0
//   far  : uniform in [-1,1]^3, normal 0, sdf = distance to the mesh
//   near : one of this step's on-surface samples moved along its normal by N(0, 0.01), normal 0,
//          sdf = distance to the mesh
// The reference stores the SIGNED distance open3d returns; every consumer (loss_s1 / loss_s2,
// src/loss_functions.py:131-136) is even in it, so the unsigned distance is written.
//
// Point-cloud-only input (n_tri == 0; reference src/dataset.py:80-131 `sampleTrainingDataPC`): the far distance is the
// distance to the nearest CLOUD POINT (reference :72-78 `shortestDistance`, evaluated here as min |p - x| instead
// of its expanded |x|^2 - 2 p.x + |p|^2 form) and the near distance is |offset| (reference :108-110), no query.
//
// Random numbers are counter-based — a pure function of (seed, step, stream, GLOBAL sample index), the same
// splitmix64 construction as diffudf_amd/synth.py — so rank r of W produces exactly its slice of the global batch
// and the numpy restatement in oracle/sampler_oracle.py reproduces every sample.
#include "dudf_internal.h"

#ifndef DUDF_SAMPLE_DBG
#define DUDF_SAMPLE_DBG 0          // timing experiments (tools/build_dbg.sh): 1 no exact evaluations, 2 nothing behind pass 0, 3 no sphere setup, 4 no scans at all
#endif

namespace {

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t stream) {
    return splitmix64(seed * 0x100000001B3ull + stream);
}
__device__ __forceinline__ double uniform01(uint64_t key, uint64_t idx) {
    uint64_t b = splitmix64(idx ^ key);
    b = splitmix64(b + key);
    return (double)(b >> 11) * (1.0 / 9007199254740992.0);
}

struct SampleArgs {
    const float *tri, *pc_pos, *pc_nrm;
    float *x, *normals, *sdf;
    int64_t n_tri, n_pc;
    int64_t n_on, n_far, n_near;           // GLOBAL stratum sizes
    int64_t on0, on1, far0, far1, near0, near1;   // this rank's [begin, end) inside each stratum
    uint64_t k_on, k_fx, k_fy, k_fz, k_pick, k_n1, k_n2;
    uint64_t seed; const int64_t* step_dev;         // step_dev != nullptr: the keys are derived in the kernel from (seed, *step_dev)
};

__host__ __device__ __forceinline__ void sample_keys(SampleArgs& a, uint64_t seed, uint64_t step) {
    const uint64_t base = 1000ull * step;
    a.k_on = stream_key(seed, base + 400); a.k_fx = stream_key(seed, base + 401); a.k_fy = stream_key(seed, base + 402);
    a.k_fz = stream_key(seed, base + 403); a.k_pick = stream_key(seed, base + 404);
    a.k_n1 = stream_key(seed, base + 405); a.k_n2 = stream_key(seed, base + 406);
}

// squared distance from p to triangle (a,b,c): closest point by Voronoi regions of the triangle.  In fp64, like the
// oracle (and like nothing in fp32 can be: |p - c|^2 of coordinates ~1 carries 1e-7 absolute, 1e-4 of a near-surface
// distance of 1e-3); 2 k triangles x 2 k queries per step is noise for the fp64 vector pipe.
__device__ __forceinline__ double tri_dist2(double px, double py, double pz, const float* t) {
    if (DUDF_SAMPLE_DBG == 1) return px + t[0];
    const double ax = t[0], ay = t[1], az = t[2];
    const double abx = t[3] - ax, aby = t[4] - ay, abz = t[5] - az;
    const double acx = t[6] - ax, acy = t[7] - ay, acz = t[8] - az;
    const double apx = px - ax, apy = py - ay, apz = pz - az;
    const double d1 = abx * apx + aby * apy + abz * apz;
    const double d2 = acx * apx + acy * apy + acz * apz;
    double cx, cy, cz;                                  // closest point - a
    if (d1 <= 0.0 && d2 <= 0.0) { cx = cy = cz = 0.0; }
    else {
        const double bpx = apx - abx, bpy = apy - aby, bpz = apz - abz;
        const double d3 = abx * bpx + aby * bpy + abz * bpz;
        const double d4 = acx * bpx + acy * bpy + acz * bpz;
        if (d3 >= 0.0 && d4 <= d3) { cx = abx; cy = aby; cz = abz; }
        else {
            const double vc = d1 * d4 - d3 * d2;
            if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
                const double v = d1 / (d1 - d3);
                cx = v * abx; cy = v * aby; cz = v * abz;
            } else {
                const double cpx = apx - acx, cpy = apy - acy, cpz = apz - acz;
                const double d5 = abx * cpx + aby * cpy + abz * cpz;
                const double d6 = acx * cpx + acy * cpy + acz * cpz;
                if (d6 >= 0.0 && d5 <= d6) { cx = acx; cy = acy; cz = acz; }
                else {
                    const double vb = d5 * d2 - d1 * d6;
                    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
                        const double w = d2 / (d2 - d6);
                        cx = w * acx; cy = w * acy; cz = w * acz;
                    } else {
                        const double va = d3 * d6 - d5 * d4;
                        if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
                            const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
                            cx = abx + w * (acx - abx); cy = aby + w * (acy - aby); cz = abz + w * (acz - abz);
                        } else {
                            const double den = 1.0 / (va + vb + vc);
                            const double v = vb * den, w = vc * den;
                            cx = abx * v + acx * w; cy = aby * v + acy * w; cz = abz * v + acz * w;
                        }
                    }
                }
            }
        }
    }
    const double dx = apx - cx, dy = apy - cy, dz = apz - cz;
    return dx * dx + dy * dy + dz * dz;
}

constexpr int QSPLIT = 8;                 // lanes that share one query point (each scans every QSPLIT-th primitive)
constexpr int SPH_CAP = 2304;             // bounding spheres held in LDS at a time (36 KiB); larger meshes go through in chunks
constexpr int CAND_CAP = 12;              // per-lane list of triangles that survive the screen (indices into the chunk)
constexpr int SCAN_U = 8;                // spheres / points a lane reads ahead in the scans
constexpr int PC_TILE = 1024;             // cloud points per LDS tile (two tiles: the next one is loaded while this one is scanned)
static_assert(2 * PC_TILE * 16 <= SPH_CAP * 16, "the cloud tiles live in the spheres' LDS");
static_assert(PC_TILE % (QSPLIT * SCAN_U) == 0 && SPH_CAP % (QSPLIT * SCAN_U) == 0, "whole read-ahead groups per tile / chunk");

// Distances, organised for the GPU (round 3: QSPLIT lanes per point on every CU instead of one thread per point on 78 workgroups;
// round 5: the scan itself — a third of the reference recipe's stage-2 step — screened in fp32).  The answer is the minimum of
// EXACT fp64 evaluations (the Voronoi-region arithmetic above, as oracle/sampler_oracle.py does it), and every evaluation that can
// be skipped is skipped on an fp32 test that errs on the safe side, so the result is bit-identical to the brute-force scan
// (tools/check_sampler.py against the previous build; tests/test_beetle_gpu.py against the oracle):
//   spheres  — each workgroup that holds a query computes the bounding sphere of every triangle once into LDS (centre in fp32,
//              radius about THAT centre in fp64, rounded up);
//   pass 0   — an upper bound of the distance from the spheres alone, min |p - c| + r (the farthest point of a sphere is at least
//              as far as the nearest point of its triangle), rounded UP, and the triangle that attains it — which is then
//              evaluated exactly: the bound is now the distance to a nearby triangle;
//   pass 1   — a triangle whose sphere's NEAREST point, |p - c| - r rounded DOWN, lies beyond the bound cannot hold the minimum;
//              the few others go on a per-lane list and are evaluated together afterwards (an evaluation costs a wavefront the
//              same whether one lane or all of them take it; the branchy fp64 code inside the scan loop WAS the kernel's time).
// The fp32 distance to a centre is within 3 ulp of the true one (a difference of floats, three squares, a sum, a square root); the
// factors 1 +- 1e-6 (1e-5 on squares) cover that several times over.  Cloud-only input: the same screen on squared distances.
// The fp32 screens of the scans.  Contraction and the raw v_sqrt_f32 (1 ulp) are fine HERE — these are bounds whose safety factors
// cover any rounding — and only here: the kernel around them keeps fp contraction off for the oracle's arithmetic.
// SCAN_U spheres / points are read ahead of their use (a lane's chain LDS read -> distance -> compare, one primitive at a time,
// was most of the kernel); the arrays are padded to whole groups with primitives farther away than anything.
__device__ __forceinline__ void screen_bound(const float4* __restrict__ g, float px, float py, float pz, int t, float& ub0, int& arg) {
#pragma clang fp contract(fast)
    float4 sp[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) sp[u] = g[u * QSPLIT];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
        const float dx = px - sp[u].x, dy = py - sp[u].y, dz = pz - sp[u].z;
        const float uu = (__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * 1.000001f + sp[u].w) * 1.000001f;
        if (uu < ub0) { ub0 = uu; arg = t + u * QSPLIT; }
    }
}
// bit u set: primitive u of the group may lie within ub (|p - c| <= ub + r, tested on squares, safe side)
__device__ __forceinline__ unsigned screen_near(const float4* __restrict__ g, float px, float py, float pz, float ub) {
#pragma clang fp contract(fast)
    float4 sp[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) sp[u] = g[u * QSPLIT];
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
        const float dx = px - sp[u].x, dy = py - sp[u].y, dz = pz - sp[u].z;
        const float lim = ub + sp[u].w;
        m |= ((dx * dx + dy * dy + dz * dz) * 0.99999f > lim * lim * 1.00001f) ? 0u : (1u << u);
    }
    return m;
}
// the same for points and a bound ub2 of the SQUARED distance
__device__ __forceinline__ unsigned screen_near2(const float4* __restrict__ g, float px, float py, float pz, float ub2) {
#pragma clang fp contract(fast)
    float4 sp[SCAN_U];
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) sp[u] = g[u * QSPLIT];
    unsigned m = 0;
#pragma unroll
    for (int u = 0; u < SCAN_U; ++u) {
        const float dx = px - sp[u].x, dy = py - sp[u].y, dz = pz - sp[u].z;
        m |= ((dx * dx + dy * dy + dz * dz) * 0.99999f > ub2) ? 0u : (1u << u);
    }
    return m;
}

__global__ __launch_bounds__(256) void sample_batch_kernel(SampleArgs a) {
    // no a*b+c -> fma here: the oracle (numpy) rounds the product of normal and offset to fp32 before the add, and HIP's
    // __fmul_rn / __fadd_rn are plain operators that hipcc's default -ffp-contract=fast would fuse (measured: 48 of 999
    // near points off by one ulp)
#pragma clang fp contract(off)
    if (a.step_dev) sample_keys(a, a.seed, (uint64_t)*a.step_dev);     // graph replay: this launch's step lives in device memory
    __shared__ float4 ts[SPH_CAP];                      // spheres (centre, radius) of a chunk of triangles | two tiles of cloud points
    __shared__ unsigned short cand[CAND_CAP * 256];     // [entry][thread]
    const int64_t n_on_l = a.on1 - a.on0, n_far_l = a.far1 - a.far0, n_near_l = a.near1 - a.near0;
    const int64_t n_l = n_on_l + n_far_l + n_near_l;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gid / QSPLIT;
    const int part = (int)(gid % QSPLIT);
    const bool live = i < n_l;
    float px = 0.f, py = 0.f, pz = 0.f, nx = 0.f, ny = 0.f, nz = 0.f, known = 0.f;
    bool query = false;
    const bool cloud_only = a.n_tri == 0;
    if (live) {
        if (i < n_on_l) {
            const int64_t g = a.on0 + i;
            const int64_t c = (int64_t)(uniform01(a.k_on, (uint64_t)g) * (double)a.n_pc);
            px = a.pc_pos[c * 3]; py = a.pc_pos[c * 3 + 1]; pz = a.pc_pos[c * 3 + 2];
            nx = a.pc_nrm[c * 3]; ny = a.pc_nrm[c * 3 + 1]; nz = a.pc_nrm[c * 3 + 2];
        } else if (i < n_on_l + n_far_l) {
            const uint64_t g = (uint64_t)(a.far0 + (i - n_on_l));
            px = (float)(uniform01(a.k_fx, g) * 2.0 - 1.0);
            py = (float)(uniform01(a.k_fy, g) * 2.0 - 1.0);
            pz = (float)(uniform01(a.k_fz, g) * 2.0 - 1.0);
            query = true;
        } else {
            const uint64_t g = (uint64_t)(a.near0 + (i - n_on_l - n_far_l));
            const int64_t k = (int64_t)(uniform01(a.k_pick, g) * (double)a.n_on);       // which on-surface sample
            const int64_t c = (int64_t)(uniform01(a.k_on, (uint64_t)k) * (double)a.n_pc);
            const double u1 = uniform01(a.k_n1, g), u2 = uniform01(a.k_n2, g);
            const float off = (float)(0.01 * sqrt(-2.0 * log1p(-u1)) * cos(6.283185307179586476925 * u2));
            px = __fadd_rn(a.pc_pos[c * 3], __fmul_rn(a.pc_nrm[c * 3], off));
            py = __fadd_rn(a.pc_pos[c * 3 + 1], __fmul_rn(a.pc_nrm[c * 3 + 1], off));
            pz = __fadd_rn(a.pc_pos[c * 3 + 2], __fmul_rn(a.pc_nrm[c * 3 + 2], off));
            query = !cloud_only;
            known = fabsf(off);
        }
    }
    double best = 3.0e38;
    const double dpx = px, dpy = py, dpz = pz;
    const bool any_query = __syncthreads_or(query) != 0;          // a workgroup of on-surface points has nothing to measure
    if (any_query && !cloud_only) {
        float ub = 3.0e38f;                                       // fp32 upper bound of the answer, only ever rounded up
        for (int64_t c0 = 0; c0 < a.n_tri; c0 += SPH_CAP) {
            const int cnt = (int)((a.n_tri - c0 < SPH_CAP) ? a.n_tri - c0 : SPH_CAP);
            __syncthreads();
            for (int e = threadIdx.x; e < cnt && DUDF_SAMPLE_DBG != 3; e += blockDim.x) {
                const float* t = a.tri + (c0 + e) * 9;
                float v[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) v[k] = t[k];
                const float cx = (v[0] + v[3] + v[6]) * (1.f / 3.f), cy = (v[1] + v[4] + v[7]) * (1.f / 3.f), cz = (v[2] + v[5] + v[8]) * (1.f / 3.f);
                double r2 = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double dx = (double)v[3 * k] - cx, dy = (double)v[3 * k + 1] - cy, dz = (double)v[3 * k + 2] - cz;
                    r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
                }
                ts[e] = make_float4(cx, cy, cz, (float)(sqrt(r2) * 1.000001) + 1e-30f);   // >= the true radius about (cx, cy, cz)
            }
            const int cntp = (cnt + QSPLIT * SCAN_U - 1) / (QSPLIT * SCAN_U) * (QSPLIT * SCAN_U);
            if ((int)threadIdx.x < cntp - cnt) ts[cnt + threadIdx.x] = make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);   // padding
            __syncthreads();
            // pass 0: bound from the spheres, and who attains it
            float ub0 = 3.0e38f; int arg = 0;
            if (query && DUDF_SAMPLE_DBG != 4)
                for (int t = part; t < cntp; t += QSPLIT * SCAN_U) screen_bound(ts + t, px, py, pz, t, ub0, arg);
#pragma unroll
            for (int m = 1; m < QSPLIT; m <<= 1) {
                const float uo = __shfl_xor(ub0, m); const int ao = __shfl_xor(arg, m);
                if (uo < ub0 || (uo == ub0 && ao < arg)) { ub0 = uo; arg = ao; }
            }
            ub = fminf(ub, ub0);
            if (query) {                                          // the QSPLIT lanes of a point evaluate the same triangle: one path per point
                const double d2 = tri_dist2(dpx, dpy, dpz, a.tri + (c0 + arg) * 9);
                if (d2 < best) { best = d2; ub = fminf(ub, (float)sqrt(d2) * 1.000001f + 1e-37f); }
            }
            // pass 1: who can still beat the bound
            int len = 0;
            if (query && DUDF_SAMPLE_DBG != 2 && DUDF_SAMPLE_DBG != 4)
                for (int t0 = part; t0 < cntp; t0 += QSPLIT * SCAN_U) {
                    unsigned m = screen_near(ts + t0, px, py, pz, ub);
                    while (m) {
                        const int t = t0 + (__builtin_ctz(m)) * QSPLIT;
                        m &= m - 1;
                        if (t == arg || t >= cnt) continue;
                        if (len < CAND_CAP) { cand[len * 256 + threadIdx.x] = (unsigned short)t; ++len; }
                        else {                                                       // list full: evaluate on the spot (and tighten)
                            const double d2 = tri_dist2(dpx, dpy, dpz, a.tri + (c0 + t) * 9);
                            if (d2 < best) { best = d2; ub = fminf(ub, (float)sqrt(d2) * 1.000001f + 1e-37f); }
                        }
                    }
                }
            int maxlen = len;
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, m));
            for (int k = 0; k < maxlen; ++k) {
                // the lanes of a point pool their bounds first (the answer is their minimum anyway)
#pragma unroll
                for (int m = 1; m < QSPLIT; m <<= 1) ub = fminf(ub, __shfl_xor(ub, m));
                if (k < len) {
                    const int t = cand[k * 256 + threadIdx.x];
                    const float4 sp = ts[t];
                    const float dx = px - sp.x, dy = py - sp.y, dz = pz - sp.z;
                    const float lim = ub + sp.w;
                    if (!((dx * dx + dy * dy + dz * dz) * 0.99999f > lim * lim * 1.00001f)) {
                        const double d2 = tri_dist2(dpx, dpy, dpz, a.tri + (c0 + t) * 9);
                        if (d2 < best) { best = d2; ub = fminf(ub, (float)sqrt(d2) * 1.000001f + 1e-37f); }
                    }
                }
            }
#pragma unroll
            for (int m = 1; m < QSPLIT; m <<= 1) ub = fminf(ub, __shfl_xor(ub, m));
        }
    }
    if (any_query && cloud_only) {
        float4* tp = ts;                                                    // two tiles of PC_TILE points, xyz + pad
        float ub2 = 3.0e38f;                                                // fp32 upper bound of the best SQUARED distance
        const int64_t n_tiles = (a.n_pc + PC_TILE - 1) / PC_TILE;
        float4 nx4[PC_TILE / 256];
        auto fetch = [&](int64_t tile) {                                    // this thread's points of a tile -> registers
#pragma unroll
            for (int j = 0; j < PC_TILE / 256; ++j) {
                const int64_t g = tile * PC_TILE + j * 256 + threadIdx.x;
                nx4[j] = (g < a.n_pc) ? make_float4(a.pc_pos[g * 3], a.pc_pos[g * 3 + 1], a.pc_pos[g * 3 + 2], 0.f)
                                      : make_float4(3.0e18f, 3.0e18f, 3.0e18f, 0.f);      // padding: farther than anything
            }
        };
        fetch(0);
        for (int64_t tile = 0; tile < n_tiles; ++tile) {
            float4* cur = tp + (tile & 1) * PC_TILE;
#pragma unroll
            for (int j = 0; j < PC_TILE / 256; ++j) cur[j * 256 + threadIdx.x] = nx4[j];
            __syncthreads();                                                // (one barrier per tile: the other buffer was read a tile ago)
            if (tile + 1 < n_tiles) fetch(tile + 1);
            if (query)
                for (int t0 = part; t0 < PC_TILE; t0 += QSPLIT * SCAN_U) {
                    unsigned m = screen_near2(cur + t0, px, py, pz, ub2);
                    while (m) {                                                       // the fp64 distance decides
                        const float4 q = cur[t0 + (__builtin_ctz(m)) * QSPLIT];
                        m &= m - 1;
                        const double dx = dpx - q.x, dy = dpy - q.y, dz = dpz - q.z;
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        if (d2 < best) { best = d2; ub2 = (float)d2 * 1.00001f + 1e-37f; }
                    }
                }
#pragma unroll
            for (int m = 1; m < QSPLIT; m <<= 1) ub2 = fminf(ub2, __shfl_xor(ub2, m));
        }
    }
    // the QSPLIT lanes of a point are adjacent lanes of one wave (256 % QSPLIT == 0, 64 % QSPLIT == 0)
#pragma unroll
    for (int m = 1; m < QSPLIT; m <<= 1) best = fmin(best, __shfl_xor(best, m));
    if (live && part == 0) {
        a.x[i * 3] = px; a.x[i * 3 + 1] = py; a.x[i * 3 + 2] = pz;
        a.normals[i * 3] = nx; a.normals[i * 3 + 1] = ny; a.normals[i * 3 + 2] = nz;
        // off-surface samples keep sdf > 0: a far point that lies exactly on a triangle, or a near point whose offset rounds to 0, would
        // otherwise carry the "on surface" marker (sdf == 0) in the wrong stratum — the loss kernels' n_hess layout check turns that
        // into NaN for the rest of the run.  FLT_MIN as a distance is 0 to every later computation.
        const float sd = query ? (float)sqrt(best) : known;
        a.sdf[i] = (i < n_on_l) ? sd : fmaxf(sd, 1.17549435e-38f);
    }
}

}  // namespace

static int sample_batch_impl(const float* tri, int64_t n_tri, const float* pc_pos, const float* pc_nrm,
                             int64_t n_pc, int64_t n_on, int64_t n_far, int64_t n_near, uint64_t seed,
                             uint64_t step, const int64_t* step_dev, int rank, int world, float* x, float* normals, float* sdf,
                             void* stream) {
    if (n_tri < 0 || (n_tri > 0 && !tri) || n_pc <= 0 || world < 1 || rank < 0 || rank >= world || n_on < 0 || n_far < 0 || n_near < 0)
        return DUDF_E_BADCFG;
    if (n_near > 0 && n_on == 0) return DUDF_E_BADCFG;
    SampleArgs a;
    a.tri = tri; a.pc_pos = pc_pos; a.pc_nrm = pc_nrm; a.x = x; a.normals = normals; a.sdf = sdf;
    a.n_tri = n_tri; a.n_pc = n_pc; a.n_on = n_on; a.n_far = n_far; a.n_near = n_near;
    auto lo = [&](int64_t m) { return m * rank / world; };
    auto hi = [&](int64_t m) { return m * (rank + 1) / world; };
    a.on0 = lo(n_on); a.on1 = hi(n_on); a.far0 = lo(n_far); a.far1 = hi(n_far); a.near0 = lo(n_near); a.near1 = hi(n_near);
    a.seed = seed; a.step_dev = step_dev;
    sample_keys(a, seed, step);
    const int64_t n_l = (a.on1 - a.on0) + (a.far1 - a.far0) + (a.near1 - a.near0);
    if (n_l == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DudfProfScope prof(PROF_OTHER, st);
    hipLaunchKernelGGL(sample_batch_kernel, dim3((unsigned)((n_l * QSPLIT + 255) / 256)), dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int dudf_sample_batch(const float* tri, int64_t n_tri, const float* pc_pos, const float* pc_nrm,
                                 int64_t n_pc, int64_t n_on, int64_t n_far, int64_t n_near, uint64_t seed,
                                 uint64_t step, int rank, int world, float* x, float* normals, float* sdf,
                                 void* stream) {
    return sample_batch_impl(tri, n_tri, pc_pos, pc_nrm, n_pc, n_on, n_far, n_near, seed, step, nullptr, rank, world, x, normals, sdf, stream);
}

extern "C" int dudf_sample_batch_at(const float* tri, int64_t n_tri, const float* pc_pos, const float* pc_nrm,
                                    int64_t n_pc, int64_t n_on, int64_t n_far, int64_t n_near, uint64_t seed,
                                    const int64_t* step_dev, int rank, int world, float* x, float* normals, float* sdf,
                                    void* stream) {
    if (!step_dev) return DUDF_E_BADCFG;
    return sample_batch_impl(tri, n_tri, pc_pos, pc_nrm, n_pc, n_on, n_far, n_near, seed, 0, step_dev, rank, world, x, normals, sdf, stream);
}
