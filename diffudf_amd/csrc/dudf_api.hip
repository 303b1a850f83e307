// C-ABI glue: argument checks, workspace layout, kernel sequencing.  See include/dudf_hip.h.
#include "dudf_internal.h"
#include <stdio.h>
#include <string.h>
#include <vector>

// ---- per-kernel HIP-event timing ---------------------------------------------------------------------------
namespace {
struct ProfRec { int slot; hipEvent_t e0, e1; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t g_prof_open[PROF_NSLOTS];
const char* kProfNames[PROF_NSLOTS] = {"pack", "sweep_fwd", "sweep_rev", "sweep_adj_fwd", "sweep_adj_rev",
                                       "wgrad_hidden", "wgrad_small", "loss_fwd", "loss_bwd", "adam", "other"};
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}  // namespace

void dudf_prof_begin(int slot, hipStream_t st) {
    if (!g_prof_on) return;
    g_prof_open[slot] = prof_event();
    (void)hipEventRecord(g_prof_open[slot], st);
}
void dudf_prof_end(int slot, hipStream_t st) {
    if (!g_prof_on) return;
    hipEvent_t e1 = prof_event();
    (void)hipEventRecord(e1, st);
    g_prof_recs.push_back({slot, g_prof_open[slot], e1});
}

namespace {

int check_ws(const DudfLayout& lo, const void* ws, size_t bytes) {
    if (!ws || bytes < lo.total_bytes || (reinterpret_cast<uintptr_t>(ws) & 15)) return DUDF_E_WORKSPACE;
    if (lo.np > (1ll << 26)) return DUDF_E_BADCFG;          // 32-bit lane offsets inside a stash layer
    return 0;
}

SweepArgs make_sweep_args(const DudfLayout& lo, const float* theta, const float* x, float* ws) {
    SweepArgs a;
    a.theta = theta; a.w1b = ws + lo.ws_w1b; a.w1t16 = ws + lo.ws_w1t16; a.wt = ws + lo.ws_wt;
    a.x = x; a.y = ws + lo.ws_y; a.g = ws + lo.ws_g; a.ybar = ws + lo.ws_ybar; a.gbar = ws + lo.ws_gbar;
    a.S = ws + lo.ws_S; a.C = ws + lo.ws_C; a.Q = ws + lo.ws_Q; a.R = ws + lo.ws_R; a.E = ws + lo.ws_E; a.A = ws + lo.ws_A;
    a.Z = ws + lo.ws_Z;
    a.n = lo.n; a.np = lo.np; a.stash_layer = lo.stash_layer;
    a.off_hid = lo.off_hid; a.hid_stride = lo.hid_stride; a.off_wo = lo.off_wo; a.off_bo = lo.off_bo;
    a.L = lo.L; a.w0 = lo.w0;
    a.store_s = 0; a.store_c = 0; a.train = 0; a.have_e = 1;
    return a;
}

}  // namespace

extern "C" {

const char* dudf_version(void) { return "dudf_hip 0.1 (gfx950, fp32 MFMA 16x16x4 sweeps + 32x32x2 wgrad)"; }

int64_t dudf_theta_count(const dudf_net_cfg* cfg) {
    DudfLayout lo;
    if (dudf_make_layout(cfg, 1, &lo)) return -1;
    return lo.n_theta;
}

size_t dudf_workspace_bytes(const dudf_net_cfg* cfg, int64_t n) {
    DudfLayout lo;
    if (dudf_make_layout(cfg, n, &lo)) return 0;
    return lo.total_bytes;
}

int dudf_query(const dudf_net_cfg* cfg, const float* theta, const float* x, int64_t n, float* out_f, float* out_g,
               void* workspace, size_t workspace_bytes, void* stream) {
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    if (n <= 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    if ((rc = dudf_launch_pack(lo, theta, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.store_c = out_g ? 1 : 0;
    if ((rc = dudf_launch_sweep(SWEEP_FWD, lo.H, a, st))) return rc;
    if (out_g && (rc = dudf_launch_sweep(SWEEP_REV, lo.H, a, st))) return rc;
    return dudf_launch_copy_out(lo, ws, out_f, out_g, st);
}

int dudf_loss_forward(const dudf_net_cfg* cfg, int mode, const float* theta, const float* x, const float* normals,
                      const float* sdf, int64_t n_local, int64_t n_global, const double* weights, double alpha,
                      float* out_terms, void* workspace, size_t workspace_bytes, void* stream) {
    if (mode != DUDF_LOSS_S1 && mode != DUDF_LOSS_SIREN) return DUDF_E_BADMODE;
    if (mode == DUDF_LOSS_S1 && weights[2] != 0.0) return DUDF_E_UNSUPPORTED;   // Hessian term: not built yet
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n_local, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    if ((rc = dudf_launch_pack(lo, theta, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.store_s = 1; a.store_c = 1; a.train = 1;
    if ((rc = dudf_launch_sweep(SWEEP_FWD, lo.H, a, st))) return rc;
    if ((rc = dudf_launch_sweep(SWEEP_REV, lo.H, a, st))) return rc;
    return dudf_launch_loss_fwd(lo, mode, normals, sdf, n_global, weights, alpha, ws, out_terms, st);
}

int dudf_s2_forward_stats(const dudf_net_cfg* cfg, const float* theta, const float* x, const float* sdf,
                          int64_t n_local, double* stats, void* workspace, size_t workspace_bytes, void* stream) {
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n_local, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    if ((rc = dudf_launch_pack(lo, theta, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.store_s = 1; a.store_c = 1; a.train = 1;
    if ((rc = dudf_launch_sweep(SWEEP_FWD, lo.H, a, st))) return rc;
    return dudf_launch_s2_stats(lo, sdf, ws, stats, st);
}

int dudf_s2_terms(const double* stats, const double* weights, float* out_terms, void* stream) {
    return dudf_launch_s2_terms(stats, weights, out_terms, reinterpret_cast<hipStream_t>(stream));
}

int dudf_loss_backward(const dudf_net_cfg* cfg, int mode, const float* theta, const float* x, const float* normals,
                       const float* sdf, int64_t n_local, int64_t n_global, const double* weights, double alpha,
                       const float* cot, const double* stats, float* dtheta, int accumulate, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (mode != DUDF_LOSS_S1 && mode != DUDF_LOSS_SIREN && mode != DUDF_LOSS_S2) return DUDF_E_BADMODE;
    if (mode == DUDF_LOSS_S1 && weights[2] != 0.0) return DUDF_E_UNSUPPORTED;
    if (mode == DUDF_LOSS_S2 && !stats) return DUDF_E_BADMODE;
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n_local, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    const int have_g = (mode != DUDF_LOSS_S2);
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dtheta, 0, (size_t)lo.n_theta * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if ((rc = dudf_launch_loss_bwd(lo, mode, normals, sdf, n_global, weights, alpha, cot, stats, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.train = 1;
    if (have_g) {
        if ((rc = dudf_launch_sweep(SWEEP_ADJ_FWD, lo.H, a, st))) return rc;
    } else {
        a.have_e = 0;                                   // no df/dx terms: e_l == 0 in the reverse adjoint sweep
    }
    if ((rc = dudf_launch_sweep(SWEEP_ADJ_REV, lo.H, a, st))) return rc;
    return dudf_launch_wgrad(lo, x, ws, dtheta, have_g, st);
}

int dudf_fields_forward(const dudf_net_cfg* cfg, const float* theta, const float* x, int64_t n, float* out_f,
                        float* out_g, void* workspace, size_t workspace_bytes, void* stream) {
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    if ((rc = dudf_launch_pack(lo, theta, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.store_s = 1; a.store_c = 1; a.train = 1;
    if ((rc = dudf_launch_sweep(SWEEP_FWD, lo.H, a, st))) return rc;
    if ((rc = dudf_launch_sweep(SWEEP_REV, lo.H, a, st))) return rc;
    return dudf_launch_copy_out(lo, ws, out_f, out_g, st);
}

int dudf_fields_backward(const dudf_net_cfg* cfg, const float* theta, const float* x, int64_t n, const float* ybar,
                         const float* gbar, float* dtheta, int accumulate, void* workspace, size_t workspace_bytes,
                         void* stream) {
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float* ws = reinterpret_cast<float*>(workspace);
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dtheta, 0, (size_t)lo.n_theta * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    if ((rc = dudf_launch_copy_in(lo, ybar, gbar, ws, st))) return rc;
    SweepArgs a = make_sweep_args(lo, theta, x, ws);
    a.train = 1;
    const int have_g = gbar != nullptr;
    if (have_g) {
        if ((rc = dudf_launch_sweep(SWEEP_ADJ_FWD, lo.H, a, st))) return rc;
    } else {
        a.have_e = 0;
    }
    if ((rc = dudf_launch_sweep(SWEEP_ADJ_REV, lo.H, a, st))) return rc;
    return dudf_launch_wgrad(lo, x, ws, dtheta, have_g, st);
}

int dudf_adam_step(float* theta, const float* dtheta, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                   double beta1, double beta2, double eps, int64_t step, double grad_scale, void* stream) {
    if (n <= 0 || step < 1) return DUDF_E_BADCFG;
    return dudf_launch_adam(theta, dtheta, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale,
                            reinterpret_cast<hipStream_t>(stream));
}

int dudf_profile_enable(int on) {
    g_prof_on = (on != 0);
    return 0;
}

int dudf_profile_dump(char* buf, size_t buflen) {
    double tot[PROF_NSLOTS] = {0};
    long cnt[PROF_NSLOTS] = {0};
    for (auto& r : g_prof_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            tot[r.slot] += ms; cnt[r.slot] += 1;
        }
        g_prof_pool.push_back(r.e0); g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
    size_t off = 0;
    for (int i = 0; i < PROF_NSLOTS; ++i) {
        if (!cnt[i]) continue;
        int w = snprintf(buf + off, off < buflen ? buflen - off : 0, "%s %ld %.6f\n", kProfNames[i], cnt[i], tot[i]);
        if (w < 0 || off + (size_t)w >= buflen) return DUDF_E_WORKSPACE;
        off += (size_t)w;
    }
    if (off < buflen) buf[off] = 0;
    return 0;
}

int dudf_debug_read_stash(const dudf_net_cfg* cfg, int which, int layer, int64_t n, float* out, void* workspace,
                          size_t workspace_bytes, void* stream) {
    DudfLayout lo;
    int rc = dudf_make_layout(cfg, n, &lo);
    if (rc) return rc;
    if ((rc = check_ws(lo, workspace, workspace_bytes))) return rc;
    if (layer < 0 || layer >= lo.L) return DUDF_E_BADCFG;
    float* ws = reinterpret_cast<float*>(workspace);
    const int64_t offs[7] = {lo.ws_S, lo.ws_C, lo.ws_Q, lo.ws_E, lo.ws_A, lo.ws_Z, lo.ws_R};
    if (which < 0 || which > 6) return DUDF_E_BADMODE;
    return dudf_launch_read_stash(lo, ws + offs[which], layer, out, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
