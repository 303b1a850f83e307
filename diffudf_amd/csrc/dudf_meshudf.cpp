// MeshUDF marching cubes on the host (C++17, no GPU): the serial, topology-ordered sign propagation + Lewiner
// triangulation that the reference keeps in a Cython extension
// (reference src/marching_cubes/_marching_cubes_lewiner_cy.pyx:1116-1774 `marching_cubes_udf`, with `Cell` :87-855,
//  `the_big_switch` :1848 / `check_the_big_switch` :2125, `test_face` :2404, `test_internal` :2436, `compute_edge_vote`
//  :1777) — SURVEY.md §8(f) row 4.  Drop-in for that extension's entry point: same inputs (unsigned field, gradient
// field, the Lewiner look-up tables the CALLER owns — the reference passes them in as a `LutProvider` built from its own
// `_marching_cubes_lewiner_luts.py`; they are an argument here as well and are not part of this library), same outputs
// (vertices in x-y-z grid units, faces, normals, values) in the same order, bit for bit: the traversal, the vote
// arithmetic (float where the reference computes in float, double where it computes in double) and the vertex
// numbering are reproduced, the code is organised differently:
//   * one `resolve()` maps (case, configuration, ambiguity tests) to a (table, sub-index, triangle count) triple; the
//     reference's two 270-line if-chains (count / emit) become `count_existing()` and `emit()` over that triple;
//   * the seed scan and the breadth-first phase share one `sign_cube()` (the reference spells the block out twice);
//   * tables are addressed through a flat descriptor array in the order of `kLutNames` (diffudf_amd/marching_cubes.py).
// Built with -ffp-contract=off: the float dot products must round like the reference's separate multiplies and adds.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <new>
#include <vector>

namespace {

enum LutId {
    EDGESRELX, EDGESRELY, EDGESRELZ, CASESCLASSIC, CASES,
    TILING1, TILING2, TILING3_1, TILING3_2, TILING4_1, TILING4_2, TILING5, TILING6_1_1, TILING6_1_2, TILING6_2, TILING7_1,
    TILING7_2, TILING7_3, TILING7_4_1, TILING7_4_2, TILING8, TILING9, TILING10_1_1, TILING10_1_1_, TILING10_1_2,
    TILING10_2, TILING10_2_, TILING11, TILING12_1_1, TILING12_1_1_, TILING12_1_2, TILING12_2, TILING12_2_, TILING13_1,
    TILING13_1_, TILING13_2, TILING13_2_, TILING13_3, TILING13_3_, TILING13_4, TILING13_5_1, TILING13_5_2, TILING14,
    TEST3, TEST4, TEST6, TEST7, TEST10, TEST12, TEST13, SUBCONFIG13, N_LUTS
};

struct Lut {
    const int8_t* v = nullptr;
    int l1 = 1, l2 = 1;
    int at(int i) const { return v[i]; }
    int at(int i, int j) const { return v[i * l1 + j]; }
    int at(int i, int j, int k) const { return v[(i * l1 + j) * l2 + k]; }
};

constexpr double kEps = 2.220446049250313e-16;      // the reference's "FLT_EPSILON" is np.spacing(1.0), a double (:36)

struct Tiling { int lut, sub, nt; };                 // sub < 0: two-index table [config][3 nt]; else [config][sub][3 nt]

struct Result {
    std::vector<float> vertices, normals, values;
    std::vector<int> faces;
};

inline float sgn(float a) { return a > 0 ? 1.f : (a < 0 ? -1.f : 0.f); }
inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline bool nonzero(const float* a) { return (std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2])) > 0; }

class Mesher {
public:
    Mesher(const float* im, const float* grads, int nz, int ny, int nx, const Lut* luts, float avg_thresh, float max_thresh)
        : im_(im), g_(grads), nz_(nz), ny_(ny), nx_(nx), L_(luts) {
        const double voxel = 2.0 / (nx - 1);
        avg_lim_ = (float)((double)avg_thresh * voxel);
        max_lim_ = (float)((double)max_thresh * voxel);
        const size_t n = (size_t)nz * ny * nx;
        sign_.assign(n, 0.f); known_.assign(n, 0); visited_.assign(n, 0);
        slots_.assign(n * 4, -1);
        bx_ = nx - 2; by_ = ny - 2; bz_ = nz - 2;      // step 1: last cube index along an axis
    }

    void run(Result& out) {
        out_ = &out;
        for (int z = 0; z <= bz_; ++z)
            for (int y = 0; y <= by_; ++y)
                for (int x = 0; x <= bx_; ++x) {
                    if (visited_[id(z, y, x)] || !thin(z, y, x)) continue;
                    sign_cube(z, y, x, /*may_defer=*/false);
                    place(z, y, x);
                    const int c = L_[CASES].at(index_, 0);
                    visited_[id(z, y, x)] = 1;
                    if (c <= 0) continue;
                    emit(resolve(c, L_[CASES].at(index_, 1)));
                    push_neighbours(z, y, x);
                    flood();
                }
    }

private:
    // ---- grid helpers ------------------------------------------------------------------------------------------------
    size_t id(int z, int y, int x) const { return ((size_t)z * ny_ + y) * nx_ + x; }
    static constexpr int kDz[8] = {0, 0, 0, 0, 1, 1, 1, 1}, kDy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, kDx[8] = {0, 1, 1, 0, 0, 1, 1, 0};
    size_t corner(int z, int y, int x, int c) const { return id(z + kDz[c], y + kDy[c], x + kDx[c]); }

    bool thin(int z, int y, int x) const {              // the cube is close enough to the surface (:1215-1218)
        float s = 0.f, m = 0.f;
        float v[8];
        for (int c = 0; c < 8; ++c) v[c] = im_[corner(z, y, x, c)];
        s = v[0]; for (int c = 1; c < 8; ++c) s = s + v[c];
        m = v[7]; for (int c = 6; c >= 0; --c) m = v[c] > m ? v[c] : m;
        const float avg = (float)(0.125 * (double)s);
        return avg < avg_lim_ && m <= max_lim_;
    }

    void push_neighbours(int z, int y, int x) {          // :1387-1398 (note: cube indices below the LAST one only)
        if (x + 1 < bx_) queue_.push_back({z, y, x + 1});
        if (y + 1 < by_) queue_.push_back({z, y + 1, x});
        if (x - 1 >= 0) queue_.push_back({z, y, x - 1});
        if (y - 1 >= 0) queue_.push_back({z, y - 1, x});
        if (z - 1 >= 0) queue_.push_back({z - 1, y, x});
        if (z + 1 < bz_) queue_.push_back({z + 1, y, x});
    }

    // ---- pseudo-signs ------------------------------------------------------------------------------------------------
    static float edge_vote(const float* g1, const float* g2, int dz, int dy, int dx) {       // :1777-1808
        const float dsum = (float)dz + (float)dy + (float)dx;
        const int k = dz != 0 ? 0 : (dy != 0 ? 1 : 2);
        const float p1 = g1[k], p2 = g2[k];
        if (dsum > 0) return (p2 > 0 && p1 < 0) ? 1.f : dot3(g1, g2);
        return (p2 < 0 && p1 > 0) ? 1.f : dot3(g1, g2);
    }

    // Signs of the 8 corners of a cube: votes from already signed grid neighbours, then (for corners nobody voted on) the
    // side of an anchor gradient.  Returns false when the cube is put aside as "unsure" (only when may_defer).
    // In the breadth-first phase (`bfs`) a corner whose votes nearly cancel also puts the cube aside (:1584-1589) — queued as
    // unsure only while `widen`, skipped for now either way.
    bool sign_cube(int z, int y, int x, bool may_defer, bool bfs = false, bool widen = true) {
        static constexpr int kDirZ[6] = {1, -1, 0, 0, 0, 0}, kDirY[6] = {0, 0, 1, -1, 0, 0}, kDirX[6] = {0, 0, 0, 0, 1, -1};
        int votes[8];
        for (int c = 0; c < 8; ++c) {
            votes[c] = 0;
            const int zi = z + kDz[c], yi = y + kDy[c], xi = x + kDx[c];
            const size_t p = id(zi, yi, xi);
            float acc = 0.f;
            if (known_[p]) { votes[c] = 1; continue; }
            if (im_[p] == 0.0f) { votes[c] = 1; continue; }
            for (int d = 0; d < 6; ++d) {
                int reach = 1;
                for (int i = 1; i <= reach; ++i) {
                    const int cz = zi + i * kDirZ[d], cy = yi + i * kDirY[d], cx = xi + i * kDirX[d];
                    if (cz > bz_ || cz < 0 || cy > by_ || cy < 0 || cx > bx_ || cx < 0) break;
                    const size_t q = id(cz, cy, cx);
                    if (im_[q] == 0.0f) { if (i >= reach) ++reach; continue; }       // look one vertex further
                    if (sign_[q] == 0.0f) continue;
                    votes[c] += 1;
                    acc += sign_[q] * edge_vote(g_ + 3 * p, g_ + 3 * q, kDirZ[d], kDirY[d], kDirX[d]);
                }
            }
            if (bfs && votes[c] >= 1 && std::fabs((double)acc) / votes[c] < (double)0.707f && !queue_.empty()) {
                if (widen) unsure_.push_back({z, y, x});
                return false;
            }
            sign_[p] = sgn(acc);                          // provisional: used by the next corners, computed again later
        }
        bool all = true;
        for (int c = 0; c < 8; ++c) all = all && votes[c] >= 1;
        if (all) return true;
        // anchor: the first corner (order 0,1,3,2,4,5,7,6) with a known sign and a gradient, else the first with a gradient
        static constexpr int kOrder[8] = {0, 1, 3, 2, 4, 5, 7, 6};
        float asign = 1.f;
        int pick = -1;
        for (int k = 0; k < 8 && pick < 0; ++k) {
            const size_t p = corner(z, y, x, kOrder[k]);
            if (known_[p] && nonzero(g_ + 3 * p)) { pick = kOrder[k]; asign = sgn(sign_[p]); }
        }
        for (int k = 0; k < 8 && pick < 0; ++k)
            if (nonzero(g_ + 3 * corner(z, y, x, kOrder[k]))) pick = kOrder[k];
        if (pick >= 0) {
            const float* g = g_ + 3 * corner(z, y, x, pick);
            base_[0] = g[0]; base_[1] = g[1]; base_[2] = g[2];
        }                                                 // (no gradient anywhere: the previous cube's vector stays, as in :1346)
        for (int k = 0; k < 3; ++k) base_[k] = (float)((double)asign * (double)base_[k]);
        for (int c = 0; c < 8; ++c) {
            if (votes[c] != 0) continue;
            const size_t p = corner(z, y, x, c);
            const float s = dot3(base_, g_ + 3 * p);
            if (may_defer && std::fabs(s) < 0.707f) { unsure_.push_back({z, y, x}); return false; }
            sign_[p] = sgn(s);
        }
        return true;
    }

    // ---- the cube under the pen ------------------------------------------------------------------------------------------
    void place(int z, int y, int x) {                    // `Cell.set_cube` + the corner bookkeeping of :1356-1374
        cz_ = z; cy_ = y; cx_ = x;
        index_ = 0;
        for (int c = 0; c < 8; ++c) {
            const size_t p = corner(z, y, x, c);
            v_[c] = (double)(sign_[p] * im_[p]);
            known_[p] = 1;
            if (v_[c] > 0.0) index_ += 1 << c;
        }
        centre_done_ = false;
    }

    void prepare() {                                     // `prepare_for_adding_triangles` (:770-806)
        static constexpr int kPerm[8] = {0, 1, 3, 2, 4, 5, 7, 6};
        for (int i = 0; i < 8; ++i) vv_[i] = v_[kPerm[i]];
        double lo = 0.0, hi = 0.0;
        for (int i = 0; i < 8; ++i) { if (vv_[i] > hi) hi = vv_[i]; if (vv_[i] < lo) lo = vv_[i]; }
        vmax_ = hi - lo;
        const double* v = v_;
        const double g[8][3] = {{v[0] - v[1], v[0] - v[3], v[0] - v[4]}, {v[0] - v[1], v[1] - v[2], v[1] - v[5]},
                                {v[3] - v[2], v[1] - v[2], v[2] - v[6]}, {v[3] - v[2], v[0] - v[3], v[3] - v[7]},
                                {v[4] - v[5], v[4] - v[7], v[0] - v[4]}, {v[4] - v[5], v[5] - v[6], v[1] - v[5]},
                                {v[7] - v[6], v[5] - v[6], v[2] - v[6]}, {v[7] - v[6], v[4] - v[7], v[3] - v[7]}};
        std::memcpy(vg_, g, sizeof(g));
    }

    void centre() {                                      // `calculate_center_vertex` (:809-853)
        static constexpr int kX[8] = {0, 1, 1, 0, 0, 1, 1, 0}, kY[8] = {0, 0, 1, 1, 0, 0, 1, 1}, kZ[8] = {0, 0, 0, 0, 1, 1, 1, 1};
        double w[8], fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
        for (int i = 0; i < 8; ++i) w[i] = 1.0 / (kEps + std::fabs(v_[i]));
        for (int i = 0; i < 8; ++i) { fx += (double)kX[i] * w[i]; fy += (double)kY[i] * w[i]; fz += (double)kZ[i] * w[i]; ff += w[i]; }
        c12_[0] = cx_ + 1.0 * fx / ff; c12_[1] = cy_ + 1.0 * fy / ff; c12_[2] = cz_ + 1.0 * fz / ff;
        double gy = 0.0, gz = 0.0;
        for (int i = 0; i < 8; ++i) gy += w[i] * vg_[i][1];
        for (int i = 0; i < 8; ++i) gz += w[i] * vg_[i][2];
        // the reference assigns the z sum to the x component and never sets the z component (:846-851); kept, the normals
        // of centre vertices are part of the output
        g12_[0] = gz; g12_[1] = gy; g12_[2] = 0.0;
        centre_done_ = true;
    }

    size_t slot(int e) const {                           // `get_index_in_facelayer` (:678-766): 4 vertex slots per cube
        size_t i = id(cz_, cy_, cx_);
        int j = 0;
        size_t up = 0;
        if (e < 8) {
            if (e >= 4) { e -= 4; up = (size_t)nx_ * ny_; }
            if (e == 1) { i += 1; j = 1; }
            else if (e == 2) i += nx_;
            else if (e == 3) j = 1;
        } else if (e < 12) {
            j = 2;
            if (e == 9) i += 1;
            else if (e == 10) i += nx_ + 1;
            else if (e == 11) i += nx_;
        } else j = 3;
        return 4 * (i + up) + j;
    }

    int new_vertex(double x, double y, double z) {
        out_->vertices.push_back((float)x); out_->vertices.push_back((float)y); out_->vertices.push_back((float)z);
        out_->normals.push_back(0.f); out_->normals.push_back(0.f); out_->normals.push_back(0.f);
        out_->values.push_back(0.f);
        return (int)out_->values.size() - 1;
    }
    void add_face(int vi) {
        out_->faces.push_back(vi);
        if (vmax_ > out_->values[vi]) out_->values[vi] = (float)vmax_;
    }
    void add_normal(int vi, float gx, float gy, float gz) {
        out_->normals[3 * vi] += gx; out_->normals[3 * vi + 1] += gy; out_->normals[3 * vi + 2] += gz;
    }
    void add_corner_normal(int vi, int i, float w) { add_normal(vi, (float)(vg_[i][0] * w), (float)(vg_[i][1] * w), (float)(vg_[i][2] * w)); }

    void corner_of_triangle(int e) {                     // `_add_face_from_edge_index` (:590-676)
        const size_t s = slot(e);
        int vi = slots_[s];
        if (e == 12) {
            if (!centre_done_) centre();
            if (vi < 0) { vi = new_vertex(c12_[0], c12_[1], c12_[2]); slots_[s] = vi; }
            add_face(vi);
            add_normal(vi, (float)g12_[0], (float)g12_[1], (float)g12_[2]);
            return;
        }
        const int dx1 = L_[EDGESRELX].at(e, 0), dx2 = L_[EDGESRELX].at(e, 1);
        const int dy1 = L_[EDGESRELY].at(e, 0), dy2 = L_[EDGESRELY].at(e, 1);
        const int dz1 = L_[EDGESRELZ].at(e, 0), dz2 = L_[EDGESRELZ].at(e, 1);
        const int i1 = dz1 * 4 + dy1 * 2 + dx1, i2 = dz2 * 4 + dy2 * 2 + dx2;
        const double w1 = 1.0 / (kEps + std::fabs(vv_[i1])), w2 = 1.0 / (kEps + std::fabs(vv_[i2]));
        if (vi < 0) {
            double fx = 0.0, fy = 0.0, fz = 0.0, ff = 0.0;
            fx += (double)dx1 * w1; fy += (double)dy1 * w1; fz += (double)dz1 * w1; ff += w1;
            fx += (double)dx2 * w2; fy += (double)dy2 * w2; fz += (double)dz2 * w2; ff += w2;
            vi = new_vertex((double)cx_ + 1.0 * fx / ff, (double)cy_ + 1.0 * fy / ff, (double)cz_ + 1.0 * fz / ff);
            slots_[s] = vi;
        }
        add_face(vi);
        add_corner_normal(vi, i1, (float)w1);
        add_corner_normal(vi, i2, (float)w2);
    }

    int edge_of(const Tiling& t, int config, int k) const {
        return t.sub < 0 ? L_[t.lut].at(config, k) : L_[t.lut].at(config, t.sub, k);
    }
    int count_existing(const Tiling& t) {                // `check_triangles(2)`: distinct vertices that already exist
        prepare();
        int seen[36], ns = 0, n = 0;
        for (int k = 0; k < 3 * t.nt; ++k) {
            const int vi = slots_[slot(edge_of(t, config_, k))];
            bool dup = false;
            for (int q = 0; q < ns; ++q) dup = dup || seen[q] == vi;
            if (!dup && vi >= 0) ++n;
            seen[ns++] = vi;
        }
        return n;
    }
    void emit(const Tiling& t) {                          // `add_triangles(2)`
        prepare();
        for (int k = 0; k < 3 * t.nt; ++k) corner_of_triangle(edge_of(t, config_, k));
    }

    // ---- Lewiner's ambiguity tests --------------------------------------------------------------------------------------
    bool face_test(int face) const {                     // :2404-2433
        static constexpr int kF[7][4] = {{0, 0, 0, 0}, {0, 4, 5, 1}, {1, 5, 6, 2}, {2, 6, 7, 3}, {3, 7, 4, 0}, {0, 3, 2, 1}, {4, 7, 6, 5}};
        const int af = face < 0 ? -face : face;
        const double A = v_[kF[af][0]], B = v_[kF[af][1]], C = v_[kF[af][2]], D = v_[kF[af][3]];
        const double d = A * C - B * D;
        if (d > -kEps && d < kEps) return face >= 0;
        return face * A * d >= 0;
    }
    bool interior_test(int c, int config, int sub, int s) const {       // :2436-2570
        double t, At = 0.0, Bt = 0.0, Ct = 0.0, Dt = 0.0;
        const double* v = v_;
        if (c == 4 || c == 10) {
            const double a = (v[4] - v[0]) * (v[6] - v[2]) - (v[7] - v[3]) * (v[5] - v[1]);
            const double b = v[2] * (v[4] - v[0]) + v[0] * (v[6] - v[2]) - v[1] * (v[7] - v[3]) - v[3] * (v[5] - v[1]);
            t = -b / (2 * a + kEps);
            if (t < 0 || t > 1) return s > 0;
            At = v[0] + (v[4] - v[0]) * t; Bt = v[3] + (v[7] - v[3]) * t; Ct = v[2] + (v[6] - v[2]) * t; Dt = v[1] + (v[5] - v[1]) * t;
        } else {
            int e = -1;
            if (c == 6) e = L_[TEST6].at(config, 2);
            else if (c == 7) e = L_[TEST7].at(config, 4);
            else if (c == 12) e = L_[TEST12].at(config, 3);
            else if (c == 13) e = L_[TILING13_5_1].at(config, sub, 0);
            // per reference edge: the two ends (t = p / (p - q + eps)) and the three parallel edges (from, to) of B, C, D
            static constexpr int kE[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7},
                                              {3, 0, 2, 1, 6, 5, 7, 4}, {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2},
                                              {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0}, {0, 4, 3, 7, 2, 6, 1, 5},
                                              {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
            if (e >= 0 && e < 12) {
                const int* k = kE[e];
                t = v[k[0]] / (v[k[0]] - v[k[1]] + kEps);
                At = 0;
                Bt = v[k[2]] + (v[k[3]] - v[k[2]]) * t; Ct = v[k[4]] + (v[k[5]] - v[k[4]]) * t; Dt = v[k[6]] + (v[k[7]] - v[k[6]]) * t;
            }
        }
        const int test = (At >= 0 ? 1 : 0) + (Bt >= 0 ? 2 : 0) + (Ct >= 0 ? 4 : 0) + (Dt >= 0 ? 8 : 0);
        switch (test) {
            case 5: return (At * Ct - Bt * Dt < kEps) ? s > 0 : false;          // (falls off the end otherwise: 0)
            case 10: return (At * Ct - Bt * Dt >= kEps) ? s > 0 : false;
            case 7: case 11: case 13: case 14: case 15: return s < 0;
            default: return s > 0;
        }
    }

    // (case, configuration) -> which triangle list applies (`the_big_switch` / `check_the_big_switch`, :1848-2395)
    Tiling resolve(int c, int config) {
        config_ = config;
        const Lut* L = L_;
        auto ft = [&](int f) { return face_test(f); };
        switch (c) {
            case 1: return {TILING1, -1, 1};
            case 2: return {TILING2, -1, 2};
            case 3: return ft(L[TEST3].at(config)) ? Tiling{TILING3_2, -1, 4} : Tiling{TILING3_1, -1, 2};
            case 4: return interior_test(c, config, 0, L[TEST4].at(config)) ? Tiling{TILING4_1, -1, 2} : Tiling{TILING4_2, -1, 6};
            case 5: return {TILING5, -1, 3};
            case 6:
                if (ft(L[TEST6].at(config, 0))) return {TILING6_2, -1, 5};
                return interior_test(c, config, 0, L[TEST6].at(config, 1)) ? Tiling{TILING6_1_1, -1, 3} : Tiling{TILING6_1_2, -1, 9};
            case 7: {
                int sub = 0;
                if (ft(L[TEST7].at(config, 0))) sub += 1;
                if (ft(L[TEST7].at(config, 1))) sub += 2;
                if (ft(L[TEST7].at(config, 2))) sub += 4;
                switch (sub) {
                    case 0: return {TILING7_1, -1, 3};
                    case 1: return {TILING7_2, 0, 5};
                    case 2: return {TILING7_2, 1, 5};
                    case 3: return {TILING7_3, 0, 9};
                    case 4: return {TILING7_2, 2, 5};
                    case 5: return {TILING7_3, 1, 9};
                    case 6: return {TILING7_3, 2, 9};
                    default: return interior_test(c, config, sub, L[TEST7].at(config, 3)) ? Tiling{TILING7_4_2, -1, 9} : Tiling{TILING7_4_1, -1, 5};
                }
            }
            case 8: return {TILING8, -1, 2};
            case 9: return {TILING9, -1, 4};
            case 10: case 12: {
                const int T = c == 10 ? TEST10 : TEST12;
                const int t11 = c == 10 ? TILING10_1_1 : TILING12_1_1, t11_ = c == 10 ? TILING10_1_1_ : TILING12_1_1_;
                const int t12 = c == 10 ? TILING10_1_2 : TILING12_1_2, t2 = c == 10 ? TILING10_2 : TILING12_2, t2_ = c == 10 ? TILING10_2_ : TILING12_2_;
                if (ft(L[T].at(config, 0))) return ft(L[T].at(config, 1)) ? Tiling{t11_, -1, 4} : Tiling{t2, -1, 8};
                if (ft(L[T].at(config, 1))) return {t2_, -1, 8};
                return interior_test(c, config, 0, L[T].at(config, 2)) ? Tiling{t11, -1, 4} : Tiling{t12, -1, 8};
            }
            case 11: return {TILING11, -1, 4};
            case 13: {
                int sub = 0;
                for (int k = 0; k < 6; ++k) if (ft(L[TEST13].at(config, k))) sub += 1 << k;
                sub = L[SUBCONFIG13].at(sub);
                if (sub == 0) return {TILING13_1, -1, 4};
                if (sub <= 6) return {TILING13_2, sub - 1, 6};
                if (sub <= 18) return {TILING13_3, sub - 7, 10};
                if (sub <= 22) return {TILING13_4, sub - 19, 12};
                if (sub <= 26) return interior_test(c, config, sub - 23, L[TEST13].at(config, 6)) ? Tiling{TILING13_5_1, sub - 23, 6}
                                                                                                  : Tiling{TILING13_5_2, sub - 23, 10};
                if (sub <= 38) return {TILING13_3_, sub - 27, 10};
                if (sub <= 44) return {TILING13_2_, sub - 39, 6};
                if (sub == 45) return {TILING13_1_, -1, 4};
                return {TILING13_1, -1, 0};               // "impossible case 13": nothing is added
            }
            case 14: return {TILING14, -1, 4};
            default: return {TILING1, -1, 0};
        }
    }

    // ---- breadth-first phase (:1400-1774) -------------------------------------------------------------------------------
    struct Cube { int z, y, x; };
    void flood() {
        bool widen = true;                               // the reference's `unsure_cases_visit_neighbours`
        while (!queue_.empty() || !unsure_.empty() || !later_.empty()) {
            Cube cur;
            if (queue_.empty()) {
                if (unsure_.empty()) { cur = later_.front(); later_.pop_front(); }
                else {
                    cur = unsure_.front();
                    if (widen) {                         // first sign the neighbours of an unsure cube (no faces from them) ...
                        if (visited_[id(cur.z, cur.y, cur.x)]) { unsure_.pop_front(); continue; }
                        push_neighbours(cur.z, cur.y, cur.x);
                        widen = false;
                        continue;
                    }
                    unsure_.pop_front();                 // ... then the cube itself
                    widen = true;
                }
            } else { cur = queue_.front(); queue_.pop_front(); }
            const int z = cur.z, y = cur.y, x = cur.x;
            if (visited_[id(z, y, x)] || !thin(z, y, x)) continue;
            if (!sign_cube(z, y, x, widen && !queue_.empty(), true, widen)) continue;
            if (!widen) continue;
            place(z, y, x);
            const int c = L_[CASES].at(index_, 0);
            if (c <= 0) { visited_[id(z, y, x)] = 1; continue; }
            const bool plain = c == 1 || c == 2 || c == 5 || c == 8 || c == 9;
            if (!plain && (!queue_.empty() || !unsure_.empty())) { later_.push_back({z, y, x}); continue; }
            const Tiling t = resolve(c, L_[CASES].at(index_, 1));
            if (count_existing(t) >= 2) {
                visited_[id(z, y, x)] = 1;
                emit(t);
                push_neighbours(z, y, x);
            }
        }
    }

    const float* im_; const float* g_;
    int nz_, ny_, nx_, bx_, by_, bz_;
    const Lut* L_;
    float avg_lim_, max_lim_;
    std::vector<float> sign_;
    std::vector<uint8_t> known_, visited_;
    std::vector<int> slots_;
    std::deque<Cube> queue_, unsure_, later_;
    float base_[3] = {0.f, 0.f, 0.f};
    // current cube
    int cz_ = 0, cy_ = 0, cx_ = 0, index_ = 0, config_ = 0;
    double v_[8], vv_[8], vg_[8][3], vmax_ = 0.0, c12_[3], g12_[3];
    bool centre_done_ = false;
    Result* out_ = nullptr;
};

}  // namespace

extern "C" {

// Runs the extraction.  `lut_data` + `lut_offsets[n_luts]` + `lut_dims[n_luts][3]` describe the caller's tables in the
// order of `LutId` (n_luts must be 51).  Returns an opaque handle (nullptr on bad arguments / allocation failure).
void* dudf_meshudf_run(const float* udf, const float* grads, int nz, int ny, int nx, const signed char* lut_data,
                       const long long* lut_offsets, const int* lut_dims, int n_luts, float avg_thresh, float max_thresh) {
    if (!udf || !grads || !lut_data || !lut_offsets || !lut_dims || n_luts != N_LUTS || nz < 2 || ny < 2 || nx < 2) return nullptr;
    Lut luts[N_LUTS];
    for (int i = 0; i < N_LUTS; ++i) {
        luts[i].v = reinterpret_cast<const int8_t*>(lut_data) + lut_offsets[i];
        luts[i].l1 = lut_dims[3 * i + 1];
        luts[i].l2 = lut_dims[3 * i + 2];
    }
    Result* r = new (std::nothrow) Result();
    if (!r) return nullptr;
    try {
        Mesher m(udf, grads, nz, ny, nx, luts, avg_thresh, max_thresh);
        m.run(*r);
    } catch (...) { delete r; return nullptr; }
    return r;
}
void dudf_meshudf_sizes(const void* handle, long long* n_vertices, long long* n_face_indices) {
    const Result* r = static_cast<const Result*>(handle);
    *n_vertices = (long long)r->values.size();
    *n_face_indices = (long long)r->faces.size();
}
// vertices (x, y, z) float32 [n][3], faces int32 [n_face_indices], raw normal sums [n][3], values [n]; any may be null
void dudf_meshudf_copy(const void* handle, float* vertices, int* faces, float* normals, float* values) {
    const Result* r = static_cast<const Result*>(handle);
    if (vertices) std::memcpy(vertices, r->vertices.data(), r->vertices.size() * sizeof(float));
    if (faces) std::memcpy(faces, r->faces.data(), r->faces.size() * sizeof(int));
    if (normals) std::memcpy(normals, r->normals.data(), r->normals.size() * sizeof(float));
    if (values) std::memcpy(values, r->values.data(), r->values.size() * sizeof(float));
}
void dudf_meshudf_free(void* handle) { delete static_cast<Result*>(handle); }

}  // extern "C"
