// CAP-UDF cell extraction on the GPU — replaces the Python triple loop of reference src/render_mc.py:201-256
// `extract_mesh_CAP(ndf, grad, resolution)` (133 M cells at 512^3, one `mcubes.marching_cubes` call per active cell).
// It consumes the (N,N,N) distance field and (N,N,N,3) direction field `dudf_grid_fields` leaves in HBM:
//
//   cell (i,j,k) is ACTIVE when min(ndf over its 8 corners) <= threshold (0.008, :205, :213);
//   corner sign = sign of dot(grad[corner 000], grad[corner]) (:224), res = +-ndf;  if any res < 0 (:230) the cell is
//   run through marching cubes at iso 0 on its 2x2x2 values (:231) and its vertices are shifted by (i,j,k) (:236-238),
//   cells being emitted in i, j, k loop order; finally v / (N-1) * 2 - 1 (:252).
//
// `mcubes` is PyMCubes 0.1.4: third-party, absent here — the per-cell marching cubes is this build's own (vertex at the
// linear-interpolation point of every sign-changing edge, table constructed by tools/gen_mc_table.py; conventions in
// oracle/capudf_oracle.py, which is the parity target; parity with PyMCubes itself is unpinned).
//
// Mapping: this is HBM-bound index/compaction work, no matrix cores.  One thread per cell, k fastest, 256 consecutive
// cells per workgroup, so the emission order of the reference (i, j, k loops) is the linear cell order:
//   pass 1  count : 8 coalesced row reads of ndf per cell (neighbouring cells and rows hit L1/L2: ~4 B/cell from HBM),
//                   the 8 direction vectors only for active cells (a thin shell around the surface); per-workgroup
//                   totals (cells, vertices, triangles)
//   pass 2  scan  : exclusive scan of the workgroup totals (one workgroup; <= 524 k entries at 512^3)
//   pass 3  emit  : the same per-cell evaluation, an in-workgroup exclusive scan (wave shuffles + 4 LDS words), float64
//                   vertices and int64 triangle indices written at their final offsets — deterministic, no atomics.
// The caller reads the three totals between pass 2 and pass 3 to size the outputs.
#include "dudf_internal.h"

namespace {

#define DUDF_MC_QUAL __constant__ const
#include "dudf_mc_table.h"          // kMcEdgeMask[256], kMcTri[256][1 + 3 * DUDF_MC_MAX_TRI], kMcEdgeCorner[12][2]

constexpr int CB = 256;                                  // cells per workgroup

struct CapArgs {
    const float* ndf; const float* grad;
    int64_t n, m;                                        // grid points / cells per side (m = n - 1)
    int64_t ncells;
    double threshold;
    uint32_t* blk;                                       // [nblocks][3] totals (pass 1)
    int64_t* off;                                        // [nblocks][3] exclusive offsets (pass 2)
    double* verts; int64_t* tris; int64_t* cells;        // outputs (pass 3)
};

struct CellEval { int idx; float res[8]; int i, j, k; };

// case index of a cell (0 = nothing to emit) and its signed corner values
__device__ __forceinline__ bool eval_cell(const CapArgs& a, int64_t cell, CellEval& ce) {
#pragma clang fp contract(off)                                   // the dot product below: products rounded, then added
    const int64_t m = a.m, n = a.n;
    const int64_t k = cell % m, ij = cell / m;
    const int64_t j = ij % m, i = ij / m;
    ce.i = (int)i; ce.j = (int)j; ce.k = (int)k; ce.idx = 0;
    const int64_t base = (i * n + j) * n + k;
    float v[8];
    float mn = 3.0e38f;
    bool nan = false;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t p = base + (int64_t)(c & 1) * n * n + (int64_t)((c >> 1) & 1) * n + ((c >> 2) & 1);
        v[c] = a.ndf[p];
        nan |= !(v[c] == v[c]);
        mn = fminf(mn, v[c]);
    }
    if (nan || (double)mn > a.threshold) return false;   // np.min(ndf_loc) > threshold: continue (float64 comparison)
    const float* g0 = a.grad + base * 3;
    const float gx = g0[0], gy = g0[1], gz = g0[2];
    int idx = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t p = base + (int64_t)(c & 1) * n * n + (int64_t)((c >> 1) & 1) * n + ((c >> 2) & 1);
        const float* g = a.grad + p * 3;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(gx, g[0]), __fmul_rn(gy, g[1])), __fmul_rn(gz, g[2]));
        const float r = d < 0.f ? -v[c] : v[c];
        ce.res[c] = r;
        idx |= (r < 0.f) ? (1 << c) : 0;
    }
    ce.idx = idx;
    return idx != 0;                                      // res.min() < 0
}

__device__ __forceinline__ unsigned pack_counts(int idx) {    // cells | vertices << 9 | triangles << 21
    if (!idx) return 0u;
    return 1u | ((unsigned)__popc((unsigned)kMcEdgeMask[idx]) << 9) | ((unsigned)kMcTri[idx][0] << 21);
}

__global__ __launch_bounds__(CB) void capudf_count_kernel(CapArgs a) {
    __shared__ unsigned part[CB / 64];
    const int64_t cell = (int64_t)blockIdx.x * CB + threadIdx.x;
    CellEval ce;
    unsigned w = 0;
    if (cell < a.ncells && eval_cell(a, cell, ce)) w = pack_counts(ce.idx);
    unsigned nc = w & 0x1ff, nv = (w >> 9) & 0xfff, nt = w >> 21;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { nc += __shfl_xor(nc, o); nv += __shfl_xor(nv, o); nt += __shfl_xor(nt, o); }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) part[wave] = nc | (nv << 9) | (nt << 21);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = 0, v = 0, t = 0;
        for (int q = 0; q < CB / 64; ++q) { c += part[q] & 0x1ff; v += (part[q] >> 9) & 0xfff; t += part[q] >> 21; }
        a.blk[(int64_t)blockIdx.x * 3] = c; a.blk[(int64_t)blockIdx.x * 3 + 1] = v; a.blk[(int64_t)blockIdx.x * 3 + 2] = t;
    }
}

// exclusive scan of the per-workgroup totals; totals -> out_counts (3 x int64)
__global__ __launch_bounds__(1024) void capudf_scan_kernel(const uint32_t* __restrict__ blk, int64_t* __restrict__ off,
                                                           int64_t nblocks, int64_t* __restrict__ out_counts) {
    __shared__ int64_t s[3][1024];
    const int t = threadIdx.x;
    const int64_t chunk = (nblocks + 1023) / 1024;
    const int64_t b0 = (int64_t)t * chunk, b1 = (b0 + chunk < nblocks) ? b0 + chunk : nblocks;
    int64_t sum[3] = {0, 0, 0};
    for (int64_t b = b0; b < b1; ++b)
        for (int q = 0; q < 3; ++q) sum[q] += blk[b * 3 + q];
    for (int q = 0; q < 3; ++q) s[q][t] = sum[q];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                  // inclusive Hillis-Steele over the 1024 partial sums
        int64_t add[3] = {0, 0, 0};
        if (t >= d) for (int q = 0; q < 3; ++q) add[q] = s[q][t - d];
        __syncthreads();
        for (int q = 0; q < 3; ++q) s[q][t] += add[q];
        __syncthreads();
    }
    int64_t run[3];
    for (int q = 0; q < 3; ++q) run[q] = s[q][t] - sum[q];
    for (int64_t b = b0; b < b1; ++b)
        for (int q = 0; q < 3; ++q) { off[b * 3 + q] = run[q]; run[q] += blk[b * 3 + q]; }
    if (t == 1023) for (int q = 0; q < 3; ++q) out_counts[q] = s[q][1023];
}

__global__ __launch_bounds__(CB) void capudf_emit_kernel(CapArgs a) {
    __shared__ unsigned part[CB / 64];
    const int64_t cell = (int64_t)blockIdx.x * CB + threadIdx.x;
    CellEval ce;
    unsigned w = 0;
    const bool live = cell < a.ncells && eval_cell(a, cell, ce);
    if (live) w = pack_counts(ce.idx);
    // exclusive scan of the three packed counts over the workgroup (the fields cannot overflow: <= 256 / 3072 / 1280)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    unsigned before = 0;
    for (int q = 0; q < wave; ++q) before += part[q];
    const unsigned exc = before + inc - w;
    if (!live) return;
    const int64_t* o3 = a.off + (int64_t)blockIdx.x * 3;
    const int64_t c0 = o3[0] + (exc & 0x1ff), v0 = o3[1] + ((exc >> 9) & 0xfff), t0 = o3[2] + (exc >> 21);
    if (a.cells) { a.cells[c0 * 3] = ce.i; a.cells[c0 * 3 + 1] = ce.j; a.cells[c0 * 3 + 2] = ce.k; }
    const unsigned mask = kMcEdgeMask[ce.idx];
    const double inv = (double)(a.n - 1);
    const double org[3] = {(double)ce.i, (double)ce.j, (double)ce.k};
    int slot[12];
    int nvert = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        slot[e] = nvert;
        if (mask >> e & 1) {
            const int ca = kMcEdgeCorner[e][0], cb = kMcEdgeCorner[e][1];
            const double va = (double)ce.res[ca], vb = (double)ce.res[cb];
            const double t = va / (va - vb);
            const int axis = e >> 2;                      // edges 0-3 run along axis 0, 4-7 along 1, 8-11 along 2
            double* out = a.verts + (v0 + nvert) * 3;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double local = (d == axis) ? t : (double)((ca >> d) & 1);
                out[d] = (local + org[d]) / inv * 2.0 - 1.0;
            }
            ++nvert;
        }
    }
    const int ntri = kMcTri[ce.idx][0];
    for (int q = 0; q < ntri; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int e = kMcTri[ce.idx][1 + 3 * q + d];
            int sl = 0;
#pragma unroll
            for (int x = 0; x < 12; ++x) sl = (x == e) ? slot[x] : sl;
            a.tris[(t0 + q) * 3 + d] = v0 + sl;
        }
}

int64_t cap_blocks(int64_t grid_n) { const int64_t m = grid_n - 1; return (m * m * m + CB - 1) / CB; }

int fill_args(const float* ndf, const float* grad, int64_t grid_n, double threshold, void* ws, size_t bytes, CapArgs* a) {
    if (grid_n < 2 || grid_n > 2048 || !ndf || !grad) return DUDF_E_BADCFG;
    const int64_t nb = cap_blocks(grid_n);
    if (!ws || bytes < dudf_capudf_workspace_bytes(grid_n) || (reinterpret_cast<uintptr_t>(ws) & 15)) return DUDF_E_WORKSPACE;
    a->ndf = ndf; a->grad = grad; a->n = grid_n; a->m = grid_n - 1; a->ncells = a->m * a->m * a->m; a->threshold = threshold;
    a->off = reinterpret_cast<int64_t*>(ws);
    a->blk = reinterpret_cast<uint32_t*>(a->off + nb * 3);
    a->verts = nullptr; a->tris = nullptr; a->cells = nullptr;
    return 0;
}

}  // namespace

extern "C" {

size_t dudf_capudf_workspace_bytes(int64_t grid_n) {
    if (grid_n < 2 || grid_n > 2048) return 0;
    const int64_t nb = cap_blocks(grid_n);
    return (size_t)(nb * 3 * (sizeof(int64_t) + sizeof(uint32_t)) + 64);
}

int dudf_capudf_count(const float* ndf, const float* grad, int64_t grid_n, double threshold, int64_t* out_counts,
                      void* workspace, size_t workspace_bytes, void* stream) {
    CapArgs a;
    int rc = fill_args(ndf, grad, grid_n, threshold, workspace, workspace_bytes, &a);
    if (rc) return rc;
    if (!out_counts) return DUDF_E_BADCFG;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DudfProfScope prof(PROF_OTHER, st);
    const int64_t nb = cap_blocks(grid_n);
    hipLaunchKernelGGL(capudf_count_kernel, dim3((unsigned)nb), dim3(CB), 0, st, a);
    hipLaunchKernelGGL(capudf_scan_kernel, dim3(1), dim3(1024), 0, st, a.blk, a.off, nb, out_counts);
    return (int)hipGetLastError();
}

int dudf_capudf_emit(const float* ndf, const float* grad, int64_t grid_n, double threshold, double* out_vertices,
                     int64_t* out_triangles, int64_t* out_cells, void* workspace, size_t workspace_bytes, void* stream) {
    CapArgs a;
    int rc = fill_args(ndf, grad, grid_n, threshold, workspace, workspace_bytes, &a);
    if (rc) return rc;
    if (!out_vertices || !out_triangles) return DUDF_E_BADCFG;
    a.verts = out_vertices; a.tris = out_triangles; a.cells = out_cells;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DudfProfScope prof(PROF_OTHER, st);
    hipLaunchKernelGGL(capudf_emit_kernel, dim3((unsigned)cap_blocks(grid_n)), dim3(CB), 0, st, a);
    return (int)hipGetLastError();
}

}  // extern "C"
