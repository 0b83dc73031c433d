/* disn.h -- C ABI of libdisn_hip.so: the MI355X (gfx950) replacement for the
 * hot path of Xharlie/DISN's dense SDF-grid inference + isosurface extraction.
 *
 * The reference has no FFI for this path; it crosses two informal boundaries
 *  which this header formalises:
 *   1. `sess.run([pred_sdf, ...], feed_dict)` once per point chunk
 *      (reference: demo/demo.py:312-326, test/create_sdf.py:224-289), and
 *   2. `os.system("./isosurface/computeMarchingCubes in.dist out.obj -i iso")`
 *      plus the .dist / OBJ file formats (reference: demo/demo.py:342-369).
 * Every entry point names the reference lines it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative DISN_E_* code and records a message retrievable with
 * disn_last_error(); no exceptions cross the boundary; no global state -- one
 * DisnCtx per (thread, device).  Pointers marked DEVICE are hipMalloc'ed
 * memory on the ctx's device and stay owned by the caller.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Calls are
 * asynchronous on `stream` unless stated otherwise.
 */
#ifndef DISN_H
#define DISN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISN_OK 0
#define DISN_E_ARG (-1)     /* bad argument (shape, null pointer, range) */
#define DISN_E_HIP (-2)     /* a HIP runtime call failed */
#define DISN_E_STATE (-3)   /* call order violated (e.g. emit before count) */
#define DISN_E_IO (-4)      /* file could not be read / written / parsed */
#define DISN_E_NOMEM (-5)

typedef struct DisnCtx DisnCtx;

/* ---- context ---------------------------------------------------------- */
/* Replaces: tf.Session creation + CUDA_VISIBLE_DEVICES device choice
 * (reference: demo/demo.py:86,142-151). */
int disn_create(int device_id, DisnCtx** out);
void disn_destroy(DisnCtx* ctx);
const char* disn_last_error(const DisnCtx* ctx);
const char* disn_version(void);

/* ---- two-stream decoder weights --------------------------------------- */
/* The 12 weight/bias pairs of the two decoders, fp32, each weight row-major
 * [Cin][Cout] (= TF HWIO with H=W=1).  Names follow the reference variables:
 *   g_* : scope "sdfprediction"          (reference: models/sdfnet.py:69-92)
 *   l_* : scope "sdfprediction_imgfeat"  (reference: models/sdfnet.py:171-190)
 *   f1c1/f1c2/f1c3 = fold1/conv{1,2,3}, f2c1/f2c2/f2c5 = fold2/conv{1,2,5}.
 * Pointers may be HOST memory: disn_sdf_set_weights copies and re-packs them. */
typedef struct DisnDecoderWeights {
  const float *g_f1c1_w, *g_f1c1_b; /* [3][64],    [64]  */
  const float *g_f1c2_w, *g_f1c2_b; /* [64][256],  [256] */
  const float *g_f1c3_w, *g_f1c3_b; /* [256][512], [512] */
  const float *g_f2c1_w, *g_f2c1_b; /* [1536][512] rows 0..511 point feat, 512..1535 global feat */
  const float *g_f2c2_w, *g_f2c2_b; /* [512][256], [256] */
  const float *g_f2c5_w, *g_f2c5_b; /* [256][1],   [1]   */
  const float *l_f1c1_w, *l_f1c1_b;
  const float *l_f1c2_w, *l_f1c2_b;
  const float *l_f1c3_w, *l_f1c3_b;
  const float *l_f2c1_w, *l_f2c1_b; /* [1984][512] rows 0..511 point feat, 512..1983 image feat */
  const float *l_f2c2_w, *l_f2c2_b;
  const float *l_f2c5_w, *l_f2c5_b;
} DisnDecoderWeights;

/* Replaces: tf.train.Saver.restore of the decoder variables
 * (reference: demo/demo.py:153-165).  Host pointers; synchronous. */
int disn_sdf_set_weights(DisnCtx* ctx, const DisnDecoderWeights* w);

/* Arithmetic of the decoder's four MFMA layers (no counterpart in the reference, whose TF graph is plain fp32).
 * Both stay within the 1e-4 tolerance on the SDF (profiles/precision_budget.json):
 *   DISN_DECODER_MIXED  (default) fold1/conv2, fold1/conv3, fold2/conv1: fp16 product + two fp8 (e4m3, K=64,
 *                       block-scaled MFMA) residual products; fold2/conv2: three bf16 products.
 *                       Activations of the first three layers beyond 448 lose their residual terms
 *                       (conversions saturate); weights must lie inside the fp16 range.
 *   DISN_DECODER_BF16X3 three bf16 products (hi/lo split of both operands) in every layer: 1.33x the matrix time;
 *                       the image term of the local stream as the fp32 four-corner blend for every point.
 * Takes effect at the next disn_sdf_grid / disn_sdf_points. */
#define DISN_DECODER_MIXED 0
#define DISN_DECODER_BF16X3 1
#define DISN_DECODER_MIXED4 2 /* the fp16 + 2 x fp8 scheme on fold2/conv2 as well: 0.89x the matrix time of MIXED, the
                                 smallest accuracy margin of the three -- meant to be chosen by disn_sdf_autotune */
int disn_sdf_set_decoder(DisnCtx* ctx, int decoder);

/* Accuracy self-check (no counterpart in the reference, whose graph is plain fp32 and needs none): n_points
 * sample points spread over the sdf_params box (half of them in patches of 8 x 4 points 1/256 of the box apart, the
 * arrangement of a dense grid's waves, half scattered: the mixed decoders evaluate the image term differently for the
 * two, see disn_sdf_grid) are evaluated through the AS-WRITTEN fp32 graph on the device
 * (reference models/model_normalization.py:169-206,241-251, models/sdfnet.py:69-92,171-190: projection, five
 * resized maps sampled and concatenated, tiled global feature, every 1x1 convolution as an exact fp32 GEMM, no
 * algebraic shortcut) and through the selected decoder; *max_abs_err = max |difference| of the two SDFs (after
 * the /10, i.e. in the units of the 1e-4 tolerance), *max_abs_sdf (may be NULL) = max |SDF|.  NaN/inf read as
 * +inf.  Needs disn_sdf_set_weights + disn_sdf_set_image.  SYNCHRONOUS; about 1 ms for 4096 points. */
int disn_sdf_selfcheck(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int n_points,
                       float* max_abs_err, float* max_abs_sdf, void* stream);
/* Decoder selection by measurement: runs the self-check for DISN_DECODER_MIXED4, _MIXED, _BF16X3 in this order
 * (cheapest first) and selects the first whose error x margin stays within `tolerance`; errs (may be NULL) is
 * indexed by decoder id, -1 = not measured.  If none qualifies, bf16x3 is selected and DISN_E_STATE returned. */
int disn_sdf_autotune(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int n_points,
                      float tolerance, float margin, int* decoder_out, float errs[3], void* stream);

/* Per-image constants.  feat_maps = the five VGG end points conv1_2, conv2_2,
 * conv3_3, conv4_3, conv5_3 as DEVICE fp32 NHWC at native resolution
 * [224,224,64] [112,112,128] [56,56,256] [28,28,512] [14,14,512];
 * global_feat = DEVICE fp32 [1024] (vgg fc8 embedding).
 * Replaces the per-chunk re-evaluation of reference
 * models/model_normalization.py:171-185 (five resize_bilinear to 137x137) and
 * the tiled global feature of models/sdfnet.py:78-82: both are per-image
 * constants and are computed once here. */
int disn_sdf_set_image(DisnCtx* ctx, const float* const feat_maps[5], const float* global_feat,
                       void* stream);

/* Regular-grid SDF evaluation.  Replaces the chunk loop
 * `for sp in range(SPLIT_SIZE): sess.run(...)` and the host-side
 * `/ SDF_WEIGHT` (reference: demo/demo.py:292-335): grid points are generated
 * on the device as np.linspace(lo,hi,res+1) per axis, x fastest
 * (demo/demo.py:297-305), projected (models/model_normalization.py:241-251),
 * sampled and decoded; the output is float32(float64(pred)/10).
 *   trans_mat : HOST float[12], row-major 4x3 (demo/demo.py:272-276)
 *   sdf_params: HOST double[6] = lo xyz, hi xyz (demo/demo.py:278)
 *   res       : sdf_res (grid has res+1 points per axis)
 *   z0,z1     : inclusive plane range to evaluate (z-slab); whole grid = 0,res
 *   sdf_out   : DEVICE float[(z1-z0+1)*(res+1)^2], x fastest then y then z
 * A grid point's value is the same bits for every z0,z1 it is evaluated under (z-slabs on several GPUs reproduce the
 * single launch).  With the mixed decoders it may differ in the last bits (<= 2e-6 after the /10) from what
 * disn_sdf_points returns for the same point, and from the same point of a grid of another res: the image term of the
 * local stream is evaluated over the pixel rectangle of the 32 points that share a wave (bf16 triple on the matrix
 * cores) when they project close together, and as the fp32 four-corner blend otherwise.  DISN_DECODER_BF16X3 always
 * blends in fp32 and is bit-identical between the two entry points. */
int disn_sdf_grid(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int res,
                  int z0, int z1, float* sdf_out, void* stream);

/* Arbitrary query points (the literal sess.run signature, reference:
 * demo/demo.py:317-325): sample_pc -> projection, sample_pc_rot -> decoder
 * input.  pts / pts_rot: DEVICE float[n][3] (may alias); out: DEVICE float[n]
 * = raw pred_sdf (NOT divided by 10, exactly what sess.run returns). */
int disn_sdf_points(DisnCtx* ctx, const float* trans_mat, const float* pts, const float* pts_rot,
                    int64_t n, float* out, void* stream);

/* As-written building blocks, exported for row-by-row parity tests
 * (SURVEY.md s8a rows a1-a4).  All DEVICE pointers. */
/* a1: reference models/model_normalization.py:241-251. xy_out float[n][2]. */
int disn_project_points(DisnCtx* ctx, const float* trans_mat, const float* pts, int64_t n,
                        float* xy_out, void* stream);
/* a2: tf.image.resize_bilinear(feat,(oh,ow)), align_corners=False, TF1 legacy
 * (reference call sites models/model_normalization.py:72,171-183). NHWC. */
int disn_resize_bilinear(DisnCtx* ctx, const float* in, int ih, int iw, int c, float* out, int oh,
                         int ow, void* stream);
/* a3+a4: tf.contrib.resampler.resampler on the five 137x137 maps + concat to
 * 1472 channels (reference: models/model_normalization.py:172-190).  Uses the
 * maps cached by disn_sdf_set_image.  feat_out float[n][1472]. */
int disn_sample_features(DisnCtx* ctx, const float* xy, int64_t n, float* feat_out, void* stream);

/* ---- once-per-image VGG-16 encoder (not the hot path) -- */
/* The slim vgg_16 variable set in the TensorFlow layout (HWIO): conv_w[i] = [3][3][Cin][Cout] of
 * conv1_1, conv1_2, conv2_1, ... conv5_3 (Cout 64,64,128,128,256,256,256,512 x6), conv_b[i] = [Cout];
 * fc_w = fc6 [7][7][512][4096], fc7 [1][1][4096][4096], fc8 [1][1][4096][1024] with their biases
 * (reference layer list: models/CNN/vgg.py:182-218).  HOST pointers; copied and packed. */
typedef struct DisnVggWeights {
  const float* conv_w[13];
  const float* conv_b[13];
  const float* fc_w[3];
  const float* fc_b[3];
} DisnVggWeights;
/* Replaces Saver.restore of the vgg_16 variables (reference demo/demo.py:153-165 for the SDF network,
 * :236-241 for the camera network).  slot 0 = SDF encoder, slot 1 = camera-estimator encoder.  Synchronous. */
int disn_vgg_set_weights(DisnCtx* ctx, int slot, const DisnVggWeights* w);
/* Replaces the encoder half of the graph (reference models/model_normalization.py:65-78;
 * cam_est/model_cam_old.py:70-78): img = DEVICE float[img_h][img_w][3], values in [0,1], BGR order, no mean
 * subtraction (demo/demo.py:263-264) -> TF1 legacy bilinear resize to 224x224 -> 13 x (conv3x3 SAME + ReLU),
 * 5 x maxpool 2x2, fc6 7x7 VALID + ReLU, fc7 + ReLU, fc8.  Exact fp32 with a fixed summation order
 * (deterministic run to run).  feat_maps: five DEVICE output buffers in the layout disn_sdf_set_image reads
 * (NHWC [224,224,64] ... [14,14,512]) or NULL (entries may be NULL too); global_feat: DEVICE float[1024] or NULL. */
int disn_vgg_forward(DisnCtx* ctx, int slot, const float* img, int img_h, int img_w, float* const feat_maps[5],
                     float* global_feat, void* stream);

/* ---- marching cubes ---------------------------------------------------- */
/* Slab description.  A whole grid is {nx,ny,nz, 0, nz, 0, nz}. */
typedef struct DisnMcGrid {
  int32_t nx, ny, nz;  /* points per axis PRESENT in the sdf buffer (x fastest) */
  int32_t z_offset;    /* global plane index of local plane 0 */
  int32_t nz_global;   /* planes of the whole grid (= sdf_res+1) */
  int32_t own0, own1;  /* local planes [own0,own1) whose edge vertices and voxel
                          layers this call owns; the rest is halo */
} DisnMcGrid;

/* Phase 1: classify + count.  Replaces MarchingCubes::compute's voxel pass of
 * `computeMarchingCubes <dist> <obj> -i iso` (reference: demo/demo.py:365-369;
 * spec SURVEY.md App. B.1-B.3).  SYNCHRONOUS (returns counts).
 * counts[0]=centre vertices, [1]=edge vertices, [2]=triangles, owned only. */
int disn_mc_count(DisnCtx* ctx, const float* sdf, const DisnMcGrid* grid, float iso,
                  uint64_t counts[3], void* stream);

/* Phase 2: emit.  Must follow disn_mc_count on the same ctx with the same sdf.
 *   bbox        : HOST double[6] lo xyz, hi xyz of the WHOLE grid (.dist header)
 *   centre_base : index given to this slab's first centre vertex
 *   edge_base   : index given to this slab's first edge vertex
 *                 (whole grid: centre_base=0, edge_base=counts[0])
 *   verts       : DEVICE double[(counts[0]+counts[1])*3]; centre vertices
 *                 first (descending k, then j, i), then edge vertices by key
 *   keys        : DEVICE int64[counts[0]+counts[1]] signed vertex keys or NULL
 *   tris        : DEVICE int32[counts[2]*3], indices already offset by the
 *                 bases, smallest index first, sorted lexicographically when
 *                 sort != 0. */
int disn_mc_emit(DisnCtx* ctx, const double* bbox, uint64_t centre_base, uint64_t edge_base,
                 double* verts, int64_t* keys, int32_t* tris, int sort, void* stream);

/* Sub-table id per voxel (uint16, 0x3ff = no surface) + edge flags in bits
 * 10..12, as left by the last disn_mc_count: DEVICE->DEVICE copy for
 * case-index parity tests.  out: DEVICE uint16[nx*ny*nz]. */
int disn_mc_debug_info(DisnCtx* ctx, uint16_t* out, void* stream);

/* Lexicographic sort of n triangles (int32 x3) in place; used by the slab
 * stitch to merge per-slab lists into the reference's global order
 * (SURVEY.md App. B.6).  max_index bounds the values (limits radix passes). */
int disn_sort_triangles(DisnCtx* ctx, int32_t* tris, uint64_t n, uint64_t max_index, void* stream);

/* Allocation-free variant of disn_sort_triangles.  It has NO counterpart in the reference; it exists for
 * callers that cannot allocate device memory: code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc.  Same arguments, same result (bit-identical order), but the sort's ping-pong buffer
 * and histograms live in a static device array of the library (12.5 MiB per GPU, held until the process
 * ends) instead of ctx-owned hipMalloc memory; the call itself performs no hipMalloc/hipFree.
 * Limit: n <= 1 048 576 triangles (DISN_E_ARG beyond it).  All calls on one GPU share that GPU's array:
 * calls that overlap in time on the same GPU (different streams / contexts / threads) race -- serialise them. */
int disn_sort_triangles_static(DisnCtx* ctx, int32_t* tris, uint64_t n, uint64_t max_index, void* stream);

/* ---- distance field of a triangle mesh (ground-truth generation) ---- */
/* The device half of the replacement for `computeDistanceField <obj> res res res [-s] [-m mode] [-g sigma]`
 * (closed Vega-FEM tool; reference call site preprocessing/create_point_sdf_grid.py:200-210, output format
 * :29-51).  Exact distance from every point of an nx x ny x nz grid spanning bbox (HOST double[6], lo xyz, hi xyz;
 * point i = lo + i (hi - lo) / (n - 1), x fastest) to the nearest triangle, float64 arithmetic, float32 result.
 *   geom    : DEVICE double[n_tris][9] = corners a, b, c; no degenerate triangles; for speed, consecutive triangles
 *             should be close in space (the host side orders them along a Morton curve)
 *   normals : NULL -> unsigned distance; else DEVICE double[n_tris][7][3] = face normal, pseudonormals of the
 *             vertices a, b, c and of the edges ab, ac, bc (Baerentzen & Aanaes 2005): the result carries the sign of
 *             (p - closest point) . pseudonormal of the closest feature -- negative inside a closed manifold surface
 *             (the tool's `-s -m 0`).  The tool's `-m 1` pipeline (Xu & Barbic 2014) is built from this call,
 *             disn_df_flood and disn_mc_* by disn_amd/distance_field.py.
 * Ties between triangles go to the lower index, so the result does not depend on the traversal.  SYNCHRONOUS. */
int disn_df_eval(DisnCtx* ctx, const double* geom, const double* normals, uint64_t n_tris, int nx, int ny, int nz,
                 const double* bbox, float* out, void* stream);
/* exterior[p] = 1 iff grid point p is connected to the boundary of the grid through points with field > level
 * (6-neighbourhood), else 0: the "remove the interior components of the offset surface" step of the same pipeline.
 * field: DEVICE float[nx*ny*nz], exterior: DEVICE uint8[nx*ny*nz].  SYNCHRONOUS. */
int disn_df_flood(DisnCtx* ctx, const float* field, float level, int nx, int ny, int nz, uint8_t* exterior, void* stream);

/* ---- file formats (host) ----------------------------------------------- */
/* .dist writer/reader.  Replaces to_binary (reference: demo/demo.py:342-353;
 * reader preprocessing/create_point_sdf_grid.py:29-51): int32 -res, res, res;
 * 6 x float64 bbox; (res+1)^3 float32, x fastest. HOST pointers. */
int disn_write_dist(const char* path, int res, const double* bbox, const float* values);
int disn_read_dist_header(const char* path, int* res, double* bbox);
int disn_read_dist(const char* path, float* values, size_t n_values);
/* Narrow-band .n.dist reader (the `-n` input mode of the reference tool; format of the shipped
 * fixture isosurface/model_normalized.obj.n.dist): dense header, int32 N,
 * N x (int32 i, j, k, float32 d).  disn_read_ndist expands it to a dense (res+1)^3 grid, x fastest.
 * Cells outside the band read +FLT_MAX (exterior) or -FLT_MAX (interior); the file has no flag for the
 * side, so it is inferred from the band: a missing cell takes the sign (relative to 0, i.e. the band is
 * assumed closed and meshed at iso ~ 0) of the last listed cell passed along +x, cells in front of a
 * line's first listed cell take that cell's sign, lines without listed cells are resolved along y, then z.
 * HOST pointers. */
int disn_read_ndist_header(const char* path, int* res, double* bbox, int64_t* n_cells);
int disn_read_ndist(const char* path, float* values, size_t n_values);
/* The same format written: the cells of a dense (res+1)^3 grid with |value| < band (the distance-field tool's
 * `-n -w band` output mode), in x-fastest order.  HOST pointers. */
int disn_write_ndist(const char* path, int res, const double* bbox, const float* values, float band);
/* OBJ writer in the reference tool's text format:
 * header comments, `v x y z` with 6 significant digits, `f a b c` 1-based. */
int disn_write_obj(const char* path, const double* verts, uint64_t n_verts, const int32_t* tris,
                   uint64_t n_tris);

/* CRC-32C (Castagnoli) of n bytes continuing from `crc` (0 to start): the checksum of the TensorFlow
 * tensor-bundle checkpoint format that replaces Saver.restore's own verification (reference
 * demo/demo.py:153-165); used by the checkpoint reader disn_amd/tf_bundle.py.  HOST pointer. */
uint32_t disn_crc32c(uint32_t crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* DISN_H */
