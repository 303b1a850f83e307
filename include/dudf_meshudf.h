/* C ABI of libdudf_meshudf.so — MeshUDF marching cubes on the host (C++17, no GPU, no torch types).
 *
 * Replaces the reference's Cython extension entry point
 *   _marching_cubes_lewiner_cy.marching_cubes_udf(im, grads, luts, st=1, classic=0, avg_thresh, max_thresh, mask=None)
 *   (reference src/marching_cubes/_marching_cubes_lewiner_cy.pyx:1116-1774), called from
 *   src/marching_cubes/_marching_cubes_lewiner.py:113-115 (`udf_mc_lewiner`) <- src/render_mc.py:130-134.
 * Same traversal, same arithmetic widths, same vertex numbering: vertices / faces / normals / values come out bit for bit
 * (tests/test_meshudf.py against fixtures produced by the reference extension itself).
 *
 * The Lewiner look-up tables are an ARGUMENT, as they are in the reference (its `LutProvider` is built by the wrapper from
 * `_marching_cubes_lewiner_luts.py`, :163-187): `lut_data` holds the int8 tables back to back, table i starting at
 * `lut_offsets[i]` with dimensions `lut_dims[3 i .. 3 i + 2]` (missing dimensions = 1), in this order (n_luts = 51):
 *   EDGESRELX EDGESRELY EDGESRELZ CASESCLASSIC CASES TILING1 TILING2 TILING3_1 TILING3_2 TILING4_1 TILING4_2 TILING5
 *   TILING6_1_1 TILING6_1_2 TILING6_2 TILING7_1 TILING7_2 TILING7_3 TILING7_4_1 TILING7_4_2 TILING8 TILING9 TILING10_1_1
 *   TILING10_1_1_ TILING10_1_2 TILING10_2 TILING10_2_ TILING11 TILING12_1_1 TILING12_1_1_ TILING12_1_2 TILING12_2 TILING12_2_
 *   TILING13_1 TILING13_1_ TILING13_2 TILING13_2_ TILING13_3 TILING13_3_ TILING13_4 TILING13_5_1 TILING13_5_2 TILING14
 *   TEST3 TEST4 TEST6 TEST7 TEST10 TEST12 TEST13 SUBCONFIG13
 */
#ifndef DUDF_MESHUDF_H
#define DUDF_MESHUDF_H
#ifdef __cplusplus
extern "C" {
#endif

/* udf [nz][ny][nx] float32 (>= 0), grads [nz][ny][nx][3] float32 (component 0 along the first array axis).
 * Returns an opaque result handle, or NULL (bad arguments, n_luts != 51, allocation failure). */
void* dudf_meshudf_run(const float* udf, const float* grads, int nz, int ny, int nx, const signed char* lut_data,
                       const long long* lut_offsets, const int* lut_dims, int n_luts, float avg_thresh, float max_thresh);
/* number of vertices and of face INDICES (3 per triangle) */
void dudf_meshudf_sizes(const void* handle, long long* n_vertices, long long* n_face_indices);
/* vertices [n][3] float32 in (x, y, z) grid units, faces int32, normals [n][3] = the accumulated (unnormalised) gradient
 * sums of `Cell._normals`, values [n]; any pointer may be NULL */
void dudf_meshudf_copy(const void* handle, float* vertices, int* faces, float* normals, float* values);
void dudf_meshudf_free(void* handle);

#ifdef __cplusplus
}
#endif
#endif
