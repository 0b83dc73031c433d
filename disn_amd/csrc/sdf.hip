// sdf.hip -- dense SDF evaluation for gfx950: camera projection, VGG feature
// sampling and the two-stream point decoder of DISN, replacing the per-chunk
// `sess.run(pred_sdf)` of the reference (demo/demo.py:312-326) and the graph it
// executes (models/model_normalization.py:169-206,241-251;
// models/sdfnet.py:69-92,171-190).
//
// Design:
//  * Per image (disn_sdf_set_image): the five VGG maps are resized once to
//    137x137 (TF1 legacy bilinear) into F137[18769][1472]; because
//    fold2/conv1 of the local stream is linear before its ReLU, the image part
//    of that layer is applied to the MAP instead of to every point:
//    Pmap[18769][512] = F137 . W_l_f2c1[512:1984,:]  (fp32 MFMA GEMM).  Per
//    point only a 4-corner bilinear blend of 512 pre-projected values remains.
//    The global-feature part of the global stream's fold2/conv1 is a constant
//    vector per image and is folded into that layer's bias.
//  * Per point (sdf_decode): one wave owns 32 points and carries them through
//    both streams with every activation in registers.  Products are
//    W^T[32 features x 16 k] (A operand, streamed in fragment order) times
//    activations[16 k x 32 points] (B operand) on v_mfma_f32_32x32x16_bf16;
//    the 32x32 fp32 result has its point on the lane and its features in the
//    16 registers, which is exactly the next layer's B operand (no LDS, no
//    lane movement).  fp32 accuracy comes from a bf16 hi/lo split of both
//    operands (a_hi*w_hi + a_lo*w_hi + a_hi*w_lo, fp32 accumulate).
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(4))) float floatx4;

constexpr int IMG = 137;                 // FLAGS.img_h/img_w (reference demo/demo.py:43-44)
constexpr int NPIX = IMG * IMG;
constexpr int FEAT_C = 1472;             // 64+128+256+512+512
constexpr int LV_C[5] = {64, 128, 256, 512, 512};
constexpr int LV_HW[5] = {224, 112, 56, 28, 14};
constexpr int LV_OFF[5] = {0, 64, 192, 448, 960};

// small per-stream parameter block (floats)
constexpr int SP_W1 = 0;      // [64][4] = w[0][f], w[1][f], w[2][f], b[f]
constexpr int SP_B2 = 256;    // [256]
constexpr int SP_B3 = 512;    // [512]
constexpr int SP_B5 = 1024;   // [256]
constexpr int SP_W6 = 1280;   // [256]
constexpr int SP_B6 = 1536;   // [1]
constexpr int SP_SC = 1540;   // [4] (ints) E8M0 scale byte of the fp8 weight operands of fold1/conv2, fold1/conv3, fold2/conv1, fold2/conv2
constexpr int SP_MX_LDS = 1544;   // the mixed-precision kernel keeps [0, SP_MX_LDS) in LDS and reads B4 from global memory
constexpr int SP_B4 = 1544;   // [512] (global stream: includes the hoisted global-feature term)
constexpr int SP_SIZE = 2056;

// fragment pairs (hi 1 KiB + lo 1 KiB) per stream, in consumption order
constexpr int PAIRS = 8 * 4 + 16 * (16 + 32) + 16 * 16;   // 1056 pairs = 2.06 MiB

// ---- a1: projection (reference models/model_normalization.py:241-251) -----
__device__ __forceinline__ void project(const float* T, float x, float y, float z, float& u, float& v) {
  float a[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = __fadd_rn(__fmul_rn(x, T[c]), __fmul_rn(y, T[3 + c]));
    s = __fadd_rn(s, __fmul_rn(z, T[6 + c]));
    a[c] = __fadd_rn(s, T[9 + c]);
  }
  u = fminf(136.0f, fmaxf(0.0f, __fdiv_rn(a[0], a[2])));
  v = fminf(136.0f, fmaxf(0.0f, __fdiv_rn(a[1], a[2])));
}

struct Corners {
  int pix[4];    // pixel index (y*137+x), clamped
  float w[4];    // resampler weights in the reference's summation order
};

// a3: tf.contrib.resampler corner weights (call sites reference
// models/model_normalization.py:172-184); order: (fy,fx) (cy,cx) (cy,fx) (fy,cx)
__device__ __forceinline__ Corners corners(float x, float y) {
  Corners c;
  float fx = floorf(x), fy = floorf(y);
  float cx = fx + 1.0f, cy = fy + 1.0f;
  float dx = __fsub_rn(cx, x), dy = __fsub_rn(cy, y);
  int ifx = (int)fx, ify = (int)fy, icx = ifx + 1, icy = ify + 1;
  bool in = x > -1.0f && y > -1.0f && x < (float)IMG && y < (float)IMG;
  bool okfx = ifx >= 0 && ifx < IMG, okcx = icx >= 0 && icx < IMG;
  bool okfy = ify >= 0 && ify < IMG, okcy = icy >= 0 && icy < IMG;
  int xf = min(max(ifx, 0), IMG - 1), xc = min(max(icx, 0), IMG - 1);
  int yf = min(max(ify, 0), IMG - 1), yc = min(max(icy, 0), IMG - 1);
  float omdx = __fsub_rn(1.0f, dx), omdy = __fsub_rn(1.0f, dy);
  c.pix[0] = yf * IMG + xf; c.w[0] = (in && okfy && okfx) ? __fmul_rn(dx, dy) : 0.0f;
  c.pix[1] = yc * IMG + xc; c.w[1] = (in && okcy && okcx) ? __fmul_rn(omdx, omdy) : 0.0f;
  c.pix[2] = yc * IMG + xf; c.w[2] = (in && okcy && okfx) ? __fmul_rn(dx, omdy) : 0.0f;
  c.pix[3] = yf * IMG + xc; c.w[3] = (in && okfy && okcx) ? __fmul_rn(omdx, dy) : 0.0f;
  return c;
}

__global__ void k_project(const float* __restrict__ pts, long long n, const float* __restrict__ Tg, float* __restrict__ xy) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float T[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) T[q] = Tg[q];
  float u, v;
  project(T, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], u, v);
  xy[2 * i] = u;
  xy[2 * i + 1] = v;
}

// ---- a2: TF1 legacy resize_bilinear, NHWC (call sites reference
// models/model_normalization.py:72,171-183).  One thread per 4 channels.
__global__ void k_resize(const float* __restrict__ in, int ih, int iw, int c, float* __restrict__ out, int oh, int ow,
                         int out_stride, int out_off) {
  const int c4 = c >> 2;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)oh * ow * c4) return;
  int ch = (int)(t % c4) * 4;
  int ox = (int)((t / c4) % ow), oy = (int)(t / ((long long)c4 * ow));
  const float sy = __fdiv_rn((float)ih, (float)oh), sx = __fdiv_rn((float)iw, (float)ow);
  float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);
  int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  int yhi = min(ylo + 1, ih - 1), xhi = min(xlo + 1, iw - 1);
  float ly = __fsub_rn(fy, (float)ylo), lx = __fsub_rn(fx, (float)xlo);
  const floatx4 tl = *(const floatx4*)(in + ((size_t)ylo * iw + xlo) * c + ch);
  const floatx4 tr = *(const floatx4*)(in + ((size_t)ylo * iw + xhi) * c + ch);
  const floatx4 bl = *(const floatx4*)(in + ((size_t)yhi * iw + xlo) * c + ch);
  const floatx4 br = *(const floatx4*)(in + ((size_t)yhi * iw + xhi) * c + ch);
  floatx4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float top = __fadd_rn(tl[q], __fmul_rn(__fsub_rn(tr[q], tl[q]), lx));
    float bot = __fadd_rn(bl[q], __fmul_rn(__fsub_rn(br[q], bl[q]), lx));
    o[q] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ly));
  }
  *(floatx4*)(out + ((size_t)oy * ow + ox) * out_stride + out_off + ch) = o;
}

// ---- a3+a4: resampler + concat on the cached 137x137x1472 map ---------------
__global__ void k_sample_features(const float* __restrict__ f137, const float* __restrict__ xy, long long n,
                                  float* __restrict__ out, int out_stride, int out_off) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (FEAT_C / 4)) return;
  long long p = t / (FEAT_C / 4);
  int ch = (int)(t % (FEAT_C / 4)) * 4;
  Corners c = corners(xy[2 * p], xy[2 * p + 1]);
  floatx4 o = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    floatx4 v = *(const floatx4*)(f137 + (size_t)c.pix[q] * FEAT_C + ch);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(o[e], __fmul_rn(c.w[q], v[e]));
  }
  *(floatx4*)(out + (size_t)p * out_stride + out_off + ch) = o;
}

// ---- fp32 GEMM: C[M][N] = act(A[M][K] . B[K][N] + bias), exact fp32 on v_mfma_f32_32x32x2_f32.  128x128 tile
// per workgroup, 4 waves of 64x64.  Rows of A / C are lda / ldc floats apart (N % 128 == 0, K % 16 == 0).
// Used per image (Pmap) and by the as-written reference decoder of disn_sdf_selfcheck.
constexpr int GT = 128, GK = 16;
__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ A, int lda, const float* __restrict__ B,
                                                   const float* __restrict__ bias /*[N], added to every row*/,
                                                   float* __restrict__ C, int ldc, int M, int N, int K, int relu) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][GT + 4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.0f;
  for (int k0 = 0; k0 < K; k0 += GK) {
    // A tile: 128 rows x 16 k; thread loads 2 float4 along k
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int idx = threadIdx.x + q * 256;       // 512 float4
      int row = idx >> 2, kq = (idx & 3) * 4;
      floatx4 v = {0, 0, 0, 0};
      if (m0 + row < M) v = *(const floatx4*)(A + (size_t)(m0 + row) * lda + k0 + kq);
      As[kq + 0][row] = v[0]; As[kq + 1][row] = v[1]; As[kq + 2][row] = v[2]; As[kq + 3][row] = v[3];
    }
    // B tile: 16 k x 128 cols
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int idx = threadIdx.x + q * 256;
      int kr = idx >> 5, cq = (idx & 31) * 4;
      floatx4 v = *(const floatx4*)(B + (size_t)(k0 + kr) * N + n0 + cq);
      *(floatx4*)&Bs[kr][cq] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      float a0 = As[kk + h][wm + r], a1 = As[kk + h][wm + 32 + r];
      float b0 = Bs[kk + h][wn + r], b1 = Bs[kk + h][wn + 32 + r];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int row = m0 + wm + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        int col = n0 + wn + b * 32 + r;
        if (row < M) {
          float v = __fadd_rn(acc[a][b][g], bias[col]);
          C[(size_t)row * ldc + col] = relu ? fmaxf(v, 0.0f) : v;
        }
      }
}

// hoisted global-feature term: b4g[n] = b[n] + sum_k global[k] * Wg[k][n].  A workgroup takes 32 outputs; its 8
// groups of 32 threads each sum a 128-long slice of k in k order (fp32 fma), and the 8 partial sums are added in
// slice order: a fixed summation order, 16 x 8 x 128 B rows in flight instead of one serial 1024-step loop.
__global__ __launch_bounds__(256) void k_global_bias(const float* __restrict__ gfeat, const float* __restrict__ Wg /*[1024][512]*/,
                                                      const float* __restrict__ b, float* __restrict__ out) {
  __shared__ float part[8][32];
  const int n = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
  float s = 0.0f;
  for (int k = 128 * sl; k < 128 * sl + 128; ++k) s = fmaf(gfeat[k], Wg[(size_t)k * 512 + n], s);
  part[sl][threadIdx.x & 31] = s;
  __syncthreads();
  if (sl == 0) {
    float t = part[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < 8; ++q) t = __fadd_rn(t, part[q][threadIdx.x]);
    out[n] = __fadd_rn(t, b[n]);
  }
}

// ---- as-written fp32 reference decoder (disn_sdf_selfcheck) -----------------------------------------------------
// fold1/conv1 of both streams: h1[s][n][64] = relu(p . W1 + b1), plain fp32 in k order (reference models/sdfnet.py:74,176)
__global__ void k_ref_l1(const float* __restrict__ pts, int n, const float* __restrict__ w1g, const float* __restrict__ b1g,
                         const float* __restrict__ w1l, const float* __restrict__ b1l, float* __restrict__ h1g,
                         float* __restrict__ h1l) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 64) return;
  int p = t >> 6, f = t & 63;
  float x = pts[3 * p], y = pts[3 * p + 1], z = pts[3 * p + 2];
  float g = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, w1g[f]), __fmul_rn(y, w1g[64 + f])), __fmul_rn(z, w1g[128 + f])), b1g[f]);
  float l = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, w1l[f]), __fmul_rn(y, w1l[64 + f])), __fmul_rn(z, w1l[128 + f])), b1l[f]);
  h1g[t] = fmaxf(g, 0.0f);
  h1l[t] = fmaxf(l, 0.0f);
}
// the tiled global feature of models/sdfnet.py:78-82: dst[p][off + c] = g[c]
__global__ void k_ref_tile_global(const float* __restrict__ g, int n, float* __restrict__ dst, int stride, int off) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 256) return;
  int p = t >> 8, c = (t & 255) * 4;
  *(floatx4*)(dst + (size_t)p * stride + off + c) = *(const floatx4*)(g + c);
}
// fold2/conv5 of both streams, their sum, /10 in float64 (models/sdfnet.py:88,186; model_normalization.py:204; demo.py:335)
__global__ void k_ref_final(const float* __restrict__ h5g, const float* __restrict__ h5l, int n, const float* __restrict__ w6g,
                            const float* __restrict__ b6g, const float* __restrict__ w6l, const float* __restrict__ b6l,
                            float* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float g = 0.0f, l = 0.0f;
  for (int k = 0; k < 256; ++k) {
    g = fmaf(h5g[(size_t)p * 256 + k], w6g[k], g);
    l = fmaf(h5l[(size_t)p * 256 + k], w6l[k], l);
  }
  float pred = __fadd_rn(__fadd_rn(g, b6g[0]), __fadd_rn(l, b6l[0]));
  out[p] = (float)((double)pred / 10.0);
}
// res[0] = max |a - b|, res[1] = max |a| as float bits (non-negative floats order like unsigned ints); NaN counts as +inf
__global__ void k_max_abs_diff(const float* __restrict__ a, const float* __restrict__ b, int n, unsigned* __restrict__ res) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float d = fabsf(a[p] - b[p]), m = fabsf(a[p]);
  if (!(d == d)) d = __int_as_float(0x7f800000);
  if (!(m == m)) m = __int_as_float(0x7f800000);
  atomicMax(&res[0], (unsigned)__float_as_int(d));
  atomicMax(&res[1], (unsigned)__float_as_int(m));
}

// ---- the decoder -----------------------------------------------------------
struct DecodeArgs {
  const uint4* ws;          // fragment stream: [global stream 1056 pairs][local stream 1056 pairs], 2 KiB per pair
  const float* sp;          // small parameter blocks [2][SP_SIZE]
  const float* pmap;        // [NPIX][512]
  float T[12];
  const float* pts;         // points mode: projection input [n][3]
  const float* pts_rot;     // points mode: decoder input [n][3]
  long long n;
  int grid_mode, res, z0;   // grid mode: planes z0.. of a (res+1)^3 grid; tile -> points by grid_point()
  long long ntiles;         // tiles of 128 points
  // grid mode: the "core" [0,ic) x [0,jc) x [0,kc) (multiples of 8, 4, 4) is cut into bricks of 8 x 4 x 4 points,
  // one tile each; bricks are ordered in super-bricks of 2^sbi x 2^sbj x 2^sbk bricks; the rest of the slab
  // follows wave by wave (grid_point: r1 patches of the planes k >= kc, r2 runs of the rows j >= jc, r3 runs of the columns i >= ic)
  int n1, nk, ic, jc, kc, sbi, sbj, sbk, nsi, nsj;
  long long core_tiles;
  unsigned r1, r2, r3;      // waves of the three remainder regions
  double lo[3], hi[3], step[3];
  int scale_out;            // 1: out = float(double(pred)/10)  (reference demo/demo.py:335,351)
  float* out;
};

// Per-launch geometry, copied to LDS by the kernels: kept out of the scalar registers on purpose (the decoders run
// at the register limit, and kernel arguments that stay live across a tile cost spills).
struct Geom {
  double lo[3], hi[3], step[3];
  float T[12];
  int res, z0, n1, ic, jc, kc, sbi, sbj, sbk, nsi, nsj, core_tiles;
  unsigned r1, r2, r3;
};
constexpr int GEOM_BYTES = 192;
static_assert(sizeof(Geom) <= GEOM_BYTES, "Geom block");
__device__ __forceinline__ void geom_store(Geom* g, const DecodeArgs& a) {
  if (threadIdx.x == 0) {
    for (int q = 0; q < 3; ++q) { g->lo[q] = a.lo[q]; g->hi[q] = a.hi[q]; g->step[q] = a.step[q]; }
    for (int q = 0; q < 12; ++q) g->T[q] = a.T[q];
    g->res = a.res; g->z0 = a.z0; g->n1 = a.n1; g->ic = a.ic; g->jc = a.jc; g->kc = a.kc;
    g->sbi = a.sbi; g->sbj = a.sbj; g->sbk = a.sbk; g->nsi = a.nsi; g->nsj = a.nsj; g->core_tiles = (int)a.core_tiles;
    g->r1 = a.r1; g->r2 = a.r2; g->r3 = a.r3;
  }
}

// Grid mode: point (i, j, k) (k relative to z0) of lane `l32` (0..31) of wave `wave` in tile `tile`.
// A wave's 32 points are an 8 x 4 block of one plane (32-byte runs of the output): their projections share pixel
// corners (~7 distinct pixel rows per gather instruction instead of ~12 for 32 consecutive points of a grid line),
// and the 32 CUs of an XCD, which take
// consecutive tiles, work inside one 32^3 super-brick whose pixel rows stay in that XCD's L2 -- walking whole
// planes re-fetched the 38 MB pixel map from HBM for every plane.  Returns false for the padding of the last tile.
__device__ __forceinline__ bool grid_point(const Geom* g, unsigned t, int wave, int l32, int& i, int& j, int& k) {
  const unsigned core = (unsigned)g->core_tiles;
  if (t < core) {
    const unsigned sbi = g->sbi, sbj = g->sbj, sbk = g->sbk, nsi = g->nsi, nsj = g->nsj;
    const unsigned w = t & ((1u << (sbi + sbj + sbk)) - 1u), sidx = t >> (sbi + sbj + sbk);
    const unsigned li = w & ((1u << sbi) - 1u), lj = (w >> sbi) & ((1u << sbj) - 1u), lk = w >> (sbi + sbj);
    const unsigned s2 = sidx / nsi, si = sidx - s2 * nsi;
    const unsigned sk = s2 / nsj, sj = s2 - sk * nsj;
    i = (int)(8u * ((si << sbi) + li)) + (l32 & 7);
    j = (int)(4u * ((sj << sbj) + lj)) + (l32 >> 3);
    k = (int)(4u * ((sk << sbk) + lk)) + wave;
    return true;
  }
  // The rest of the slab, one wave at a time, in an enumeration under which a grid point shares its wave with the same
  // 31 others whatever the slab (so that its value does not depend on how the grid was cut: the image term's arithmetic
  // follows the wave's pixel rectangle, blend_mfma):
  //   B  planes k >= kc, (i < ic, j < jc): the same 8 x 4 patches as in the bricks, patch by patch;
  //   C  rows j >= jc of every plane: runs of 32 consecutive i;      D  columns i >= ic (j < jc): runs of 32 consecutive j.
  // Lanes beyond the end of a run repeat its last point and are not stored.
  const unsigned n1 = (unsigned)g->n1, ic = g->ic, jc = g->jc, kc = g->kc;
  const unsigned wb = g->r1, wc = g->r2, total = wb + wc + g->r3;
  unsigned q = (t - core) * 4u + (unsigned)wave;
  bool valid = q < total;
  if (!valid) q = total - 1u;
  if (q < wb) {
    const unsigned pi_n = ic >> 3, ppp = pi_n * (jc >> 2), kk = q / ppp, rem = q - kk * ppp, pj = rem / pi_n;
    k = (int)(kc + kk); j = (int)(4u * pj) + (l32 >> 3); i = (int)(8u * (rem - pj * pi_n)) + (l32 & 7);
  } else if (q - wb < wc) {
    q -= wb;
    const unsigned nseg = (n1 + 31u) >> 5, per_k = (n1 - jc) * nseg, kk = q / per_k, rem = q - kk * per_k, jj = rem / nseg;
    const unsigned ii = 32u * (rem - jj * nseg) + (unsigned)l32;
    valid = valid && ii < n1;
    k = (int)kk; j = (int)(jc + jj); i = (int)min(ii, n1 - 1u);
  } else {
    q -= wb + wc;
    const unsigned nseg = (jc + 31u) >> 5, per_k = (n1 - ic) * nseg, kk = q / per_k, rem = q - kk * per_k, ii = rem / nseg;
    const unsigned jj = 32u * (rem - ii * nseg) + (unsigned)l32;
    valid = valid && jj < jc;
    k = (int)kk; i = (int)(ic + ii); j = (int)min(jj, jc - 1u);
  }
  return valid;
}
// a0: np.linspace per axis in float64, cast to float32 (reference demo/demo.py:297-305)
__device__ __forceinline__ void grid_xyz(const Geom* g, int i, int j, int k, float& x, float& y, float& z) {
  const int res = g->res;
  k += g->z0;
  x = (float)(i == res ? g->hi[0] : __dadd_rn(__dmul_rn((double)i, g->step[0]), g->lo[0]));
  y = (float)(j == res ? g->hi[1] : __dadd_rn(__dmul_rn((double)j, g->step[1]), g->lo[1]));
  z = (float)(k == res ? g->hi[2] : __dadd_rn(__dmul_rn((double)k, g->step[2]), g->lo[2]));
}

// ---- weight pipeline: global -> LDS ring by LDS-DMA, shared by the 4 waves ----
// The fragment stream is consumed in stages of 16 pairs (32 KiB).  Four ring
// slots; while stage g is being multiplied, g+1 is in LDS (from the middle of g
// on), g+2 is in flight and g+3 is being issued.  The only rendezvous is in the
// MIDDLE of a stage (after pair 7): every wave waits for its own part of stage
// g+1 (counted vmcnt), all meet at one s_barrier, and the slot of stage g-1,
// which every wave has left by then, starts being refilled.  A refill period
// runs from the middle of one stage to the middle of the next: two 256 B
// LDS-DMA sub-pieces per wave in the MFMA shadows of every pair.  Fragment reads
// run three pairs ahead of their MFMAs and cross stage boundaries, so a stage
// start costs nothing.
constexpr int STAGE_PAIRS = 16;                          // fragment pairs per stage
constexpr int STAGE_BYTES = STAGE_PAIRS * 2048;          // 32 KiB
constexpr int NSLOT = 4;                                 // ring slots
constexpr int STAGES_PER_STREAM = PAIRS / STAGE_PAIRS;   // 66
constexpr int STAGES_PER_TILE = 2 * STAGES_PER_STREAM;   // 132
constexpr int SP_LDS_FLOATS = 2 * SP_SIZE;
constexpr int LDS_BYTES = NSLOT * STAGE_BYTES + SP_LDS_FLOATS * 4 + 192;   // ring | parameters | Geom
static_assert(PAIRS % STAGE_PAIRS == 0, "stream must be a whole number of stages");

#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Frags {
  bf16x8 hi[4], lo[4];   // register ring: pair p lives in slot p & 3
};

struct Pipe {
  const char* ws;        // global fragment stream: stages [ws, ws_end), then it repeats
  const char* ws_end;
  const char* src_next;  // stage that the next dma_setup issues (stages are set up in order)
  char* ring;            // LDS ring base
  unsigned ring32;       // its LDS byte address
  int g;                 // stage being consumed (counts across tiles)
  int wave, lane;
  const uint4* cur;      // this lane's view of stage g / g+1 in LDS
  const uint4* nxt;
  const char* dma_src;   // uniform: this wave's 8 KiB share of the stage being issued (g+3), first / second half
  const char* dma_src_hi;
  unsigned dma_dst_hi;   // LDS address of the second half of its destination
  unsigned voff;         // lane * 4

  __device__ __forceinline__ const uint4* slot_ptr(int stage) const {
    return (const uint4*)(ring + (stage & (NSLOT - 1)) * STAGE_BYTES) + lane;
  }
  // M0 = LDS base of the LDS-DMA.  hipcc has no use for M0 in this kernel (gfx9+ DS
  // instructions do not read it; `make check-m0` verifies it on the generated code), so it is
  // written only here, twice per stage, far ahead of the instructions that read it.
  __device__ __forceinline__ void dma_setup(int stage) {
    dma_src = src_next + wave * (STAGE_BYTES / 4);
    src_next += STAGE_BYTES;
    if (src_next == ws_end) src_next = ws;
    dma_src_hi = dma_src + 4096;
    unsigned dst = ring32 + (stage & (NSLOT - 1)) * STAGE_BYTES + wave * (STAGE_BYTES / 4);
    dma_dst_hi = dst + 4096;
    asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
  }
  __device__ __forceinline__ void dma_hi() const { asm volatile("s_mov_b32 m0, %0" ::"s"(dma_dst_hi) : "memory"); }
  // one 256 B sub-piece J of the wave's share (the instruction offset advances the global AND the
  // LDS address, so a half of 4 KiB shares one base).  Dword granularity and a scalar base on purpose:
  // the issue cost of an LDS-DMA grows with its size (a 1 KiB dwordx4 piece outlasts the 32-cycle shadow
  // of an MFMA and idles the matrix pipe), and every M0 write next to its use stalls the wave.  asm so
  // that hipcc does not count it (its own bookkeeping would drain vmcnt(0) before every ds_read)
  template <int J>
  __device__ __forceinline__ void dma_sub() const {
    asm volatile("global_load_lds_dword %0, %1 offset:%2"
                 :
                 : "v"(voff), "s"(J >= 16 ? dma_src_hi : dma_src), "i"((J & 15) * 256)
                 : "memory");
  }
  template <int A, int B>
  __device__ __forceinline__ void dma_range() const {   // sub-pieces A..B-1 (prologue only)
    if constexpr (A < B) {
      dma_sub<A>();
      dma_range<A + 1, B>();
    }
  }
  // The mixed-precision kernels move the same bytes as eight 1 KiB pieces per wave and stage (J = 0..7; voff = lane * 16
  // there): their K=64 MFMAs have 64-cycle shadows, which a 1 KiB piece fits, and a quarter of the issue slots is
  // spent on the refill (tools/group_cost.hip, modes 2 / 4).
  template <int J>
  __device__ __forceinline__ void dma4_sub() const {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2"
                 :
                 : "v"(voff), "s"(J >= 4 ? dma_src_hi : dma_src), "i"((J & 3) * 1024)
                 : "memory");
  }
  template <int A, int B>
  __device__ __forceinline__ void dma4_range() const {
    if constexpr (A < B) {
      dma4_sub<A>();
      dma4_range<A + 1, B>();
    }
  }
  // middle of stage g: stage g+1 must be in LDS for everyone; slot of g-1 is free afterwards.  PIECES = LDS-DMA
  // instructions per wave and stage (32 of 256 B or 8 of 1 KiB)
  template <int PIECES>
  __device__ __forceinline__ void mid_sync() {
    static_assert(PIECES == 32 || PIECES == 8, "piece count");
    if constexpr (PIECES == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // all but the pieces of stage g+2
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // no tail special case: past the end the refill just re-reads the (wrapping) stream into a slot
    // nobody consumes; the kernel drains vmcnt(0) before it exits
    dma_setup(g + 3);
  }
  __device__ __forceinline__ void advance() {
    ++g;
    cur = nxt;
    nxt = slot_ptr(g + 1);
  }
};

// single MFMAs in VGPR-accumulator and AGPR-accumulator form.  Inline asm so the
// accumulator file of each chain is ours to choose (the 512-wide fold2/conv1
// block owns all 256 AGPRs, everything else accumulates in VGPRs).  Hazards the
// compiler does not pad for an asm statement: a VALU-written operand needs two
// wait states before the MFMA reads it (s_nop 1 on the first MFMA of a pair, the
// only one that can follow VALU code); a result needs the drain below before
// any non-MFMA reader.
template <bool AGPR, bool LEAD>
__device__ __forceinline__ void mfma1(floatx16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (AGPR) {
    if constexpr (LEAD) asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (LEAD) asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}
// 8-pass MFMA result -> any other reader: 12+ wait states
__device__ __forceinline__ void drain_v(floatx16& acc) { asm("s_nop 15\n\ts_nop 3" : "+v"(acc)); }
__device__ __forceinline__ void drain_a(floatx16& acc) { asm("s_nop 15\n\ts_nop 3" : "+a"(acc)); }

// pair P of the current stage: w_lo*x_hi + w_hi*x_lo + w_hi*x_hi into acc, with the
// fragment reads of pair P+3 (possibly in the next stage), the pipeline's
// rendezvous / refill and the caller's filler work slotted between the MFMAs.
// extra(gap<M>) runs in the shadow of the M-th MFMA.  Measured budget (tools/mfma_cost.hip and the
// per-stage cycle counts): a shadow has ~16 free issue cycles; the fragment read + refill piece fill
// those of MFMAs 0 and 1, so filler VALU work belongs after MFMA 2, at most 4 ops per pair
template <int M>
using gap = std::integral_constant<int, M>;
template <int P, bool AGPR, class Extra>
__device__ __forceinline__ void pair_step(Pipe& pipe, Frags& fr, floatx16& acc, const bf16x8& xhi, const bf16x8& xlo,
                                          Extra extra) {
  constexpr int c = P & 3, n = (P + 3) & 3;
  constexpr int J = 2 * ((P + 8) & 15);   // refill period runs from the middle of one stage to the middle of the next
  const uint4* src = (P + 3 < STAGE_PAIRS) ? pipe.cur + (P + 3) * 128 : pipe.nxt + (P + 3 - STAGE_PAIRS) * 128;
  mfma1<AGPR, true>(acc, fr.lo[c], xhi);
  SCHED_FENCE();
  fr.hi[n] = as_bf16x8(src[0]);
  SCHED_FENCE();
  pipe.template dma_sub<J>();
  extra(gap<0>{});
  SCHED_FENCE();
  mfma1<AGPR, false>(acc, fr.hi[c], xlo);
  SCHED_FENCE();
  fr.lo[n] = as_bf16x8(src[64]);
  SCHED_FENCE();
  pipe.template dma_sub<J + 1>();
  if constexpr (J + 1 == 15) pipe.dma_hi();
  extra(gap<1>{});
  SCHED_FENCE();
  mfma1<AGPR, false>(acc, fr.hi[c], xhi);
  SCHED_FENCE();
  if constexpr (P == 7) pipe.template mid_sync<32>();
  extra(gap<2>{});
  SCHED_FENCE();
  if constexpr (P == STAGE_PAIRS - 1) pipe.advance();
}

template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ float relu1(float v) {
  float r;
  asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));   // one op (fmaxf adds a canonicalize)
  return r;
}
// ReLU + bf16 hi/lo split of accumulator elements e, e+1 (e even) into the two B fragments
__device__ __forceinline__ void cvt_pair(const floatx16& acc, int e, bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
  float v0 = relu1(acc[e]), v1 = relu1(acc[e + 1]);
  __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
  hi[e >> 3][e & 7] = h0;
  hi[e >> 3][(e & 7) + 1] = h1;
  lo[e >> 3][e & 7] = (__bf16)(v0 - (float)h0);
  lo[e >> 3][(e & 7) + 1] = (__bf16)(v1 - (float)h1);
}

// the same conversion in three parts of <= 4 VALU ops, so that it can be spread over MFMA shadows:
// T = 2 * (element pair) + part
struct CvtState {
  float v0, v1;
  __bf16 h0, h1;
};
template <int T>
__device__ __forceinline__ void cvt_part(const floatx16& acc, CvtState& st, bf16x8 (&hi)[2], bf16x8 (&lo)[2]) {
  constexpr int e = 2 * (T / 2), part = T % 2;
  if constexpr (part == 0) {   // 3 VALU ops (5 when the accumulator has to be read out of an AGPR)
    st.v0 = relu1(acc[e]);
    st.v1 = relu1(acc[e + 1]);
    st.h0 = (__bf16)st.v0;
    st.h1 = (__bf16)st.v1;
  } else {                     // 4 VALU ops, one of them two passes
    hi[e >> 3][e & 7] = st.h0;
    hi[e >> 3][(e & 7) + 1] = st.h1;
    lo[e >> 3][e & 7] = (__bf16)(st.v0 - (float)st.h0);
    lo[e >> 3][(e & 7) + 1] = (__bf16)(st.v1 - (float)st.h1);
  }
}

// accumulator initialised with a per-feature vector: reg g <-> feature base + (g&3) + 8(g>>2) + 4h
__device__ __forceinline__ floatx16 bias_init(const float* b, int base, int h) {
  floatx16 acc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    floatx4 v = *(const floatx4*)(b + base + 8 * q + 4 * h);
    acc[4 * q + 0] = v[0]; acc[4 * q + 1] = v[1]; acc[4 * q + 2] = v[2]; acc[4 * q + 3] = v[3];
  }
  return acc;
}

// one stage of fold1/conv3: 16 k-steps over the resident h2 fragments into one 32-feature block
__device__ __forceinline__ void l3_stage(Pipe& pipe, Frags& fr, floatx16& acc3, const bf16x8 (&h2hi)[16],
                                         const bf16x8 (&h2lo)[16]) {
  static_for<16>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    pair_step<P, false>(pipe, fr, acc3, h2hi[P], h2lo[P], [](auto) {});
  });
}

// one decoder stream for the wave's 32 points; sp = this stream's parameters in LDS
// LOCAL is a run-time (wave-uniform) flag on purpose: one copy of the ~50 KB stage code serves both
// streams, so the loop over tiles stays inside the 64 KB instruction cache
__device__ __forceinline__ float stream_pass(const bool LOCAL, Pipe& pipe, Frags& fr, const float* sp,
                                             const float* __restrict__ pmap, float px, float py, float pz, float u,
                                             float v) {
  const int lane = pipe.lane, h = lane >> 5;
  // fold2/conv1 accumulators (point part, 512 wide): bias (+ image term) first, while the
  // register file is still nearly empty so that many gather loads can be in flight
  floatx16 acc4[16];
  Corners cn;
  if (LOCAL) {
    asm volatile("" : "+v"(u), "+v"(v));   // keep hipcc from hoisting the gather above the previous stream
    cn = corners(u, v);
  }
  if (LOCAL) {
    // image part of fold2/conv1: bilinear blend of the pre-projected map.  The 16 loads of block
    // nb4+1 are issued before block nb4 is blended, so the memory latency is paid once, not 16 times
    // (the phase is then bound by the texture-address path: 1 MiB per 128 points at 64 B/clk)
    floatx4 t[2][16];
    auto gather = [&](floatx4 (&dst)[16], int nb4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* src = pmap + (size_t)cn.pix[c] * 512 + 32 * nb4 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[4 * c + q] = *(const floatx4*)(src + 8 * q);
      }
    };
    gather(t[0], 0);
    static_for<16>([&](auto nc) {
      constexpr int nb4 = decltype(nc)::value;
      if constexpr (nb4 + 1 < 16) gather(t[(nb4 + 1) & 1], nb4 + 1);
      // the blend of the four corners (fold2/conv1's bias was added to the map by disn_sdf_set_image)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc4[nb4][4 * q + e] = cn.w[0] * t[nb4 & 1][q][e];
#pragma unroll
      for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc4[nb4][4 * q + e] = fmaf(cn.w[c], t[nb4 & 1][4 * c + q][e], acc4[nb4][4 * q + e]);
      asm volatile("" : "+a"(acc4[nb4]));   // park the block in the accumulator file now, not at its first MFMA
    });
  } else {
#pragma unroll
    for (int nb4 = 0; nb4 < 16; ++nb4) {
      acc4[nb4] = bias_init(sp + SP_B4, 32 * nb4, h);
      asm volatile("" : "+a"(acc4[nb4]));
    }
  }
  // fold1/conv1 (3 -> 64) on the VALU, written straight into B-fragment order
  bf16x8 h1hi[4], h1lo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int f = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      floatx4 w = *(const floatx4*)(sp + SP_W1 + 4 * f);
      float t = fmaf(pz, w[2], fmaf(py, w[1], px * w[0])) + w[3];
      t = fmaxf(t, 0.0f);
      __bf16 hb = (__bf16)t;
      h1hi[ks][j] = hb;
      h1lo[ks][j] = (__bf16)(t - (float)hb);
    }
  }
  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks; all of h2 stays resident as B fragments
  bf16x8 h2hi[16], h2lo[16];
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    floatx16 acc2[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc2[b] = bias_init(sp + SP_B2, 32 * (4 * st + b), h);
    static_for<16>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      constexpr int b = P >> 2, ks = P & 3;
      pair_step<P, false>(pipe, fr, acc2[b], h1hi[ks], h1lo[ks], [](auto) {});
    });
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int nb2 = 4 * st + b;
      drain_v(acc2[b]);
      bf16x8 thi[2], tlo[2];
#pragma unroll
      for (int e = 0; e < 16; e += 2) cvt_pair(acc2[b], e, thi, tlo);
      h2hi[2 * nb2] = thi[0]; h2hi[2 * nb2 + 1] = thi[1];
      h2lo[2 * nb2] = tlo[0]; h2lo[2 * nb2 + 1] = tlo[1];
    }
  }
  // fold1/conv3 (256 -> 512) chained into fold2/conv1: each finished 32-feature
  // block of h3 is consumed at once as two k-steps of all sixteen accumulators.
  // Software-pipelined by one block: while block nb3 feeds fold2/conv1 (two
  // stages), block nb3+1 has already been accumulated (one stage) and its
  // ReLU + hi/lo split fills the gaps between the fold2/conv1 MFMAs.
  bf16x8 xhi[2], xlo[2];
  {
    floatx16 acc3 = bias_init(sp + SP_B3, 0, h);
    l3_stage(pipe, fr, acc3, h2hi, h2lo);
    drain_v(acc3);
#pragma unroll
    for (int e = 0; e < 16; e += 2) cvt_pair(acc3, e, xhi, xlo);
  }
  // the bias of the block after next is fetched from LDS behind an MFMA of the previous stage, so a
  // stage never starts with a wait on the LDS queue
  floatx16 bnext = bias_init(sp + SP_B3, 32, h);
#pragma unroll 1
  for (int nb3 = 0; nb3 < 15; ++nb3) {
    floatx16 acc3n = bnext;
    l3_stage(pipe, fr, acc3n, h2hi, h2lo);
    bf16x8 nhi[2], nlo[2];
    CvtState cs;
    // the split of block nb3+1 (16 parts) goes behind the third MFMA of pairs 1..8 of both stages;
    // the first one comes 6 MFMAs after the last write of acc3n, so no drain is needed
    static_for<16>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      pair_step<P, true>(pipe, fr, acc4[P >> 1], xhi[P & 1], xlo[P & 1], [&](auto m) {
        if constexpr (decltype(m)::value == 2 && P >= 1 && P <= 8) cvt_part<P - 1>(acc3n, cs, nhi, nlo);
      });
    });
    static_for<16>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      pair_step<P, true>(pipe, fr, acc4[8 + (P >> 1)], xhi[P & 1], xlo[P & 1], [&](auto m) {
        if constexpr (decltype(m)::value == 2 && P >= 1 && P <= 8) cvt_part<8 + P - 1>(acc3n, cs, nhi, nlo);
        if constexpr (decltype(m)::value == 2 && P == 11) bnext = bias_init(sp + SP_B3, 32 * (nb3 < 14 ? nb3 + 2 : 15), h);
      });
    });
    xhi[0] = nhi[0]; xhi[1] = nhi[1]; xlo[0] = nlo[0]; xlo[1] = nlo[1];
  }
  // last block: nothing left to convert
  static_for<16>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    pair_step<P, true>(pipe, fr, acc4[P >> 1], xhi[P & 1], xlo[P & 1], [](auto) {});
  });
  static_for<16>([&](auto pc) {
    constexpr int P = decltype(pc)::value;
    pair_step<P, true>(pipe, fr, acc4[8 + (P >> 1)], xhi[P & 1], xlo[P & 1], [](auto) {});
  });
  // fold2/conv2 (512 -> 256), consuming the fold2/conv1 accumulators block by block;
  // the split of block nb4+1 fills the gaps between the MFMAs of block nb4
  floatx16 acc5[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) acc5[nb] = bias_init(sp + SP_B5, 32 * nb, h);
  bf16x8 yhi[2][2], ylo[2][2];
  CvtState cs5;
  drain_a(acc4[0]);
#pragma unroll
  for (int e = 0; e < 16; e += 2) cvt_pair(acc4[0], e, yhi[0], ylo[0]);
  static_for<16>([&](auto nc) {
    constexpr int nb4 = decltype(nc)::value;
    constexpr int cur = nb4 & 1, nx = cur ^ 1, nn = nb4 < 15 ? nb4 + 1 : 15;
    static_for<16>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      pair_step<P, false>(pipe, fr, acc5[P >> 1], yhi[cur][P & 1], ylo[cur][P & 1], [&](auto m) {
        // block nn was last written a stage or more ago: no drain; one of its 16 parts per pair
        if constexpr (nb4 < 15 && decltype(m)::value == 2) cvt_part<P>(acc4[nn], cs5, yhi[nx], ylo[nx]);
      });
    });
  });
  // fold2/conv5 (256 -> 1): this lane holds half of its point's features
  float part = 0.0f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) {
    drain_v(acc5[nb]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      floatx4 w = *(const floatx4*)(sp + SP_W6 + 32 * nb + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) part = fmaf(fmaxf(acc5[nb][4 * q + e], 0.0f), w[e], part);
    }
  }
  part += __shfl_xor(part, 32, 64);
  return part + sp[SP_B6];
}

// persistent: one workgroup (4 waves x 32 points) per CU; tiles are dealt per XCD (below), so a workgroup may
// get no tile at all (e.g. 257 tiles on 256 CUs): it then only drains its prologue LDS-DMA and exits
__global__ __launch_bounds__(256, 1) void sdf_decode(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* sp_lds = (float*)(lds + NSLOT * STAGE_BYTES);
  for (int i = threadIdx.x; i < SP_LDS_FLOATS; i += 256) sp_lds[i] = a.sp[i];
  Geom* geom = (Geom*)(lds + NSLOT * STAGE_BYTES + SP_LDS_FLOATS * 4);
  geom_store(geom, a);
  __syncthreads();
  const long long ntiles = a.ntiles;
  Pipe pipe;
  pipe.ws = (const char*)a.ws;
  pipe.ws_end = pipe.ws + (size_t)STAGES_PER_TILE * STAGE_BYTES;
  pipe.src_next = pipe.ws;
  pipe.ring = lds;
  pipe.ring32 = (unsigned)(size_t)(lds_ptr_t)lds;
  pipe.g = 0;
  pipe.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  pipe.lane = threadIdx.x & 63;
  pipe.voff = pipe.lane * 4;
  pipe.dma_setup(0); pipe.dma_range<0, 16>(); pipe.dma_hi(); pipe.dma_range<16, 32>();
  pipe.dma_setup(1); pipe.dma_range<0, 16>(); pipe.dma_hi(); pipe.dma_range<16, 32>();
  pipe.dma_setup(2); pipe.dma_range<0, 16>(); pipe.dma_hi();   // second half: pairs 0..7 of stage 0
  asm volatile("s_waitcnt vmcnt(48)" ::: "memory");   // stage 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  pipe.cur = pipe.slot_ptr(0);
  pipe.nxt = pipe.slot_ptr(1);
  Frags fr;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    fr.hi[t] = as_bf16x8(pipe.cur[t * 128]);
    fr.lo[t] = as_bf16x8(pipe.cur[t * 128 + 64]);
  }
  // Blocks b and b + 8 share an XCD (workgroups are dealt round-robin over the 8 XCDs): every XCD gets one
  // contiguous eighth of the tiles, so that neighbouring tiles, which sample the same Pmap pixels, meet in the
  // same L2 instead of pulling them into all eight.
  long long tile_first = blockIdx.x, tile_end = ntiles, tile_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const long long per = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7;
    tile_first = xcd * per + (blockIdx.x >> 3);
    tile_end = min(ntiles, (xcd + 1) * per);
    tile_step = gridDim.x >> 3;
  }
  for (long long tile = tile_first; tile < tile_end; tile += tile_step) {
    long long p = tile * 128 + pipe.wave * 32 + (pipe.lane & 31);
    bool valid = p < a.n;
    if (!valid) p = a.n - 1;
    float x, y, z, xr, yr, zr;
    if (a.grid_mode) {
      int i, j, k;
      valid = grid_point(geom, (unsigned)tile, pipe.wave, pipe.lane & 31, i, j, k);
      p = ((long long)k * geom->n1 + j) * geom->n1 + i;
      grid_xyz(geom, i, j, k, x, y, z);
      xr = x; yr = y; zr = z;
    } else {
      x = a.pts[3 * p]; y = a.pts[3 * p + 1]; z = a.pts[3 * p + 2];
      xr = a.pts_rot[3 * p]; yr = a.pts_rot[3 * p + 1]; zr = a.pts_rot[3 * p + 2];
    }
    float u, v;
    project(geom->T, x, y, z, u, v);
    float pred = 0.0f;   // global stream first, then local: pred = g + l (reference models/model_normalization.py:204)
#pragma unroll 1
    for (int strm = 0; strm < 2; ++strm)
      pred = __fadd_rn(pred, stream_pass(strm != 0, pipe, fr, sp_lds + strm * SP_SIZE, a.pmap, xr, yr, zr, u, v));
    if (valid && pipe.lane < 32) a.out[p] = a.scale_out ? (float)((double)pred / 10.0) : pred;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup's LDS allocation
}

// =====================================================================================================
// Mixed-precision decoder ("mx", the default): fold1/conv2, fold1/conv3 and fold2/conv1 run as
//     W.x  ~=  f16(W).f16(x)                      v_mfma_f32_32x32x16_f16            (1 unit of matrix time)
//            + fp8(W - f16(W)).fp8(x)             v_mfma_scale_f32_32x32x64_f8f6f4   (0.5 units)
//            + fp8(W).fp8(x - f16(x))             v_mfma_scale_f32_32x32x64_f8f6f4   (0.5 units)
// instead of three bf16 products (3 units): the fp16 product leaves a 2^-12 relative residual on each operand,
// and a correction that is itself good to 2^-4 (fp8 e4m3) brings the sum to ~2^-16, the accuracy of the bf16
// triple (profiles/precision_budget.json).  The two residual products are block-scaled MFMAs at K = 64; their
// power-of-two scales (per layer, chosen when the weights are packed) ride in the instruction's E8M0 scale
// operands, so all three products accumulate into one fp32 accumulator.  fold2/conv2, whose input carries the
// image / global feature term and dominates the error budget, stays on the bf16 triple.
// Fragment stream, in 1 KiB UNITS (one ds_read_b128 per lane):  a "group" = 32 output features x 64 k =
// 4 units f16(W) (k-steps of 16) + 2 units fp8 residual + 2 units fp8 W = 8 units = 8 KiB; a bf16 "pair" =
// 2 units.  A stage = 32 units = 4 groups or 16 pairs.  The register ring holds 8 units; the slot an MFMA has
// just consumed is refilled with the unit 8 further on.
// MODE.FP16_OVFL is set for the kernel: f32->fp8/f16 conversions saturate (448 / 65504) instead of producing
// NaN / inf, so an activation beyond fp8's range only loses its correction terms.
typedef __attribute__((ext_vector_type(4))) int intx4;
typedef __attribute__((ext_vector_type(8))) int intx8;
typedef __attribute__((ext_vector_type(2))) short shortx2;

constexpr int UNITS_PER_STAGE = 32;
constexpr int XL_SHIFT = 11;    // x - f16(x) is scaled by 2^11 before its fp8 conversion (|x - f16(x)| * 2^11 <= x)

// LDS of the mixed-precision kernel: ring | small parameters [2][SP_MX_LDS] | B4 of the global stream [512] | parked operands
constexpr int MX_SP_BYTES = (2 * SP_MX_LDS + 512) * 4;
constexpr int MX_PARK_BYTES_PER_WAVE = 4096;   // two 8-register operands (fp8 residuals of h2 groups 2, 3) per lane
constexpr int MX_LDS_BYTES = NSLOT * STAGE_BYTES + MX_SP_BYTES + 4 * MX_PARK_BYTES_PER_WAVE + 192;   // ... | Geom
static_assert(MX_LDS_BYTES <= 160 * 1024, "LDS budget");

struct Ring8 {
  intx4 f[4];   // slots 0..3
  intx8 e[2];   // slots 4,5 and 6,7 (one 8-register fp8 operand each)
};

// The B operands of one group of 64 input features for this wave's 32 points.  They are made IN PLACE out of the
// two 32x32 fp32 accumulator blocks (a = features 0..31 of the group, b = 32..63) that hold the previous layer's
// result: step i (0..7) reads a[i], a[8+i], b[i], b[8+i] and writes back
//     a[i]   = f16 pair (a[i], a[8+i])       b[i]   = f16 pair (b[i], b[8+i])
//     a[8+i] = fp8 x 4 of the four values    b[8+i] = fp8 x 4 of their f16 residuals * 2^11
// so that afterwards a = [hi k-step 0 | hi k-step 1 | fp8(x) 8 regs], b = [hi k-step 2 | hi k-step 3 | fp8(x - f16 x)]
// with no register beyond the accumulators themselves.  The resulting k order (which accumulator element sits in
// which operand position) is matched by the weight packer: mx_hi_elem / mx_fp8_elem below.
typedef __attribute__((ext_vector_type(16))) int intx16;
struct XG {
  intx16 a, b;   // fp32 accumulator bits before the split, operand bits after it
};
__device__ __forceinline__ intx16 as_bits(const floatx16& v) { return __builtin_bit_cast(intx16, v); }
template <int S>
__device__ __forceinline__ intx4 xg_hi(const XG& x) {   // f16(x), k-step S of 4
  constexpr int o = 4 * (S & 1);
  return S < 2 ? __builtin_shufflevector(x.a, x.a, o, o + 1, o + 2, o + 3)
               : __builtin_shufflevector(x.b, x.b, o, o + 1, o + 2, o + 3);
}
__device__ __forceinline__ intx8 xg_x8(const XG& x) {
  return __builtin_shufflevector(x.a, x.a, 8, 9, 10, 11, 12, 13, 14, 15);
}
__device__ __forceinline__ intx8 xg_xl8(const XG& x) {
  return __builtin_shufflevector(x.b, x.b, 8, 9, 10, 11, 12, 13, 14, 15);
}
// accumulator element (block 0/1 = a/b, register g) behind operand positions; register g of a block <-> feature
// 32*block + (g&3) + 8*(g>>2) + 4*h of the group (h = lane >> 5)
__host__ __device__ constexpr int mx_hi_elem(int s, int j) {      // k-step s, element j -> 16*block + g
  return 16 * (s >> 1) + 4 * (s & 1) + (j >> 1) + 8 * (j & 1);
}
__host__ __device__ constexpr int mx_fp8_elem(int byte) {         // byte 0..31 of the lane's fp8 operand -> 16*block + g
  return 16 * ((byte >> 1) & 1) + (byte >> 2) + 8 * (byte & 1);
}
__host__ __device__ constexpr int mx_feature(int elem, int h) {   // -> feature 0..63 of the group
  return 32 * (elem >> 4) + ((elem & 15) & 3) + 8 * ((elem & 15) >> 2) + 4 * h;
}

template <int S>
__device__ __forceinline__ void ring_put(Ring8& rg, const uint4 v) {
  if constexpr (S < 4) {
    rg.f[S] = __builtin_bit_cast(intx4, v);
  } else {
    constexpr int q = (S - 4) >> 1, o = 4 * ((S - 4) & 1);
    rg.e[q][o + 0] = (int)v.x; rg.e[q][o + 1] = (int)v.y; rg.e[q][o + 2] = (int)v.z; rg.e[q][o + 3] = (int)v.w;
  }
}
template <int S>
__device__ __forceinline__ intx4 ring_get(const Ring8& rg) {
  if constexpr (S < 4) {
    return rg.f[S];
  } else {
    constexpr int q = (S - 4) >> 1, o = 4 * ((S - 4) & 1);
    return __builtin_shufflevector(rg.e[q], rg.e[q], o, o + 1, o + 2, o + 3);
  }
}

// stage unit U (slot U & 7) has just been consumed: refill its slot with unit U + 8
template <int U>
__device__ __forceinline__ void unit_read(Pipe& pipe, Ring8& rg) {
  const uint4* src = (U + 8 < UNITS_PER_STAGE) ? pipe.cur + (U + 8) * 64 : pipe.nxt + (U + 8 - UNITS_PER_STAGE) * 64;
  ring_put<U & 7>(rg, src[0]);
  SCHED_FENCE();
}
// shadows (0..5) of a group step that carry its two refill pieces (first <= second: the pieces are issued in order)
#ifndef MX_DMA_GAPS
#define MX_DMA_GAPS 4, 5
#endif
constexpr int MX_DMA_GAP[2] = {MX_DMA_GAPS};
static_assert(MX_DMA_GAP[0] <= MX_DMA_GAP[1] && MX_DMA_GAP[1] <= 5, "refill pieces in ascending order");
// LDS-DMA piece Q (0..7, 1 KiB per wave) of the refill period; the pieces of a stage must be issued in ascending order
// from the middle of the stage on (4, 5, 6, 7 | rendezvous | 0, 1, 2, 3): M0 moves to the second half after piece 3
template <int Q>
__device__ __forceinline__ void piece_dma(Pipe& pipe) {
  pipe.template dma4_sub<Q>();
  if constexpr (Q == 3) pipe.dma_hi();
  SCHED_FENCE();
}

template <bool AGPR, bool LEAD>
__device__ __forceinline__ void mfma_f16(floatx16& acc, const intx4& a, const intx4& b) {
  if constexpr (AGPR) {
    if constexpr (LEAD) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (LEAD) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}
template <bool AGPR, bool LEAD>
__device__ __forceinline__ void mfma_f16(intx16& acc, const intx4& a, const intx4& b) {   // accumulator held as bits (XG)
  static_assert(!AGPR, "operand groups accumulate in VGPRs");
  if constexpr (LEAD) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// fp8 e4m3 x fp8 e4m3, K = 64, product scaled by 2^(sa-127) * 2^(sb-127) (byte 0 of the scale registers)
template <bool AGPR>
__device__ __forceinline__ void mfma_fp8(floatx16& acc, const intx8& a, const intx8& b, int sa, int sb) {
  if constexpr (AGPR)
    asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+a"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
  else
    asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}

// Both residual products of a layer carry the same total scale: the weights are stored as fp8(W * 2^b) and
// fp8((W - f16 W) * 2^(b + 11)), the activations as fp8(x) and fp8((x - f16 x) * 2^11), so either product needs
// 2^-(b + 11): E8M0 bytes (127 - b) on the weight side and (127 - 11) on the activation side.
template <bool AGPR>
__device__ __forceinline__ void mfma_fp8(intx16& acc, const intx8& a, const intx8& b, int sa, int sb) {
  static_assert(!AGPR, "operand groups accumulate in VGPRs");
  asm("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb));
}
__device__ __forceinline__ void drain_v(intx16& acc) { asm("s_nop 15\n\ts_nop 3" : "+v"(acc)); }
struct Scales {
  int w;    // 127 - b of the layer
  int xl;   // 127 - XL_SHIFT
};

// group G (0..3) of the current stage: six MFMAs into acc.  extra(gap<M>) runs in the shadow of MFMA M; the
// shadows of the two K=64 MFMAs (M = 4, 5; 64 cycles each) are the ones with room for VALU work
template <int G, bool AGPR, class Acc, class Extra>
__device__ __forceinline__ void group_step_x(Pipe& pipe, Ring8& rg, Acc& acc, const XG& x, const intx8& xl8,
                                             const Scales& sc, Extra extra);
template <int G, bool AGPR, class Acc, class Extra>
__device__ __forceinline__ void group_step(Pipe& pipe, Ring8& rg, Acc& acc, const XG& x, const Scales& sc, Extra extra) {
  group_step_x<G, AGPR>(pipe, rg, acc, x, xg_xl8(x), sc, extra);
}
// the same with the fp8 residual operand given separately (it may have been parked in LDS)
template <int G, bool AGPR, class Acc, class Extra>
__device__ __forceinline__ void group_step_x(Pipe& pipe, Ring8& rg, Acc& acc, const XG& x, const intx8& xl8,
                                             const Scales& sc, Extra extra) {
  constexpr int U = 8 * G;
  constexpr int Q = (2 * G + 4) & 7;   // this group's two 1 KiB refill pieces: by default in the shadows of the K=64 MFMAs
  mfma_f16<AGPR, true>(acc, rg.f[0], xg_hi<0>(x));
  SCHED_FENCE();
  unit_read<U + 0>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 0) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 0) piece_dma<Q + 1>(pipe);
  extra(gap<0>{});
  SCHED_FENCE();
  mfma_f16<AGPR, false>(acc, rg.f[1], xg_hi<1>(x));
  SCHED_FENCE();
  unit_read<U + 1>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 1) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 1) piece_dma<Q + 1>(pipe);
  extra(gap<1>{});
  SCHED_FENCE();
  mfma_f16<AGPR, false>(acc, rg.f[2], xg_hi<2>(x));
  SCHED_FENCE();
  unit_read<U + 2>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 2) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 2) piece_dma<Q + 1>(pipe);
  extra(gap<2>{});
  SCHED_FENCE();
  mfma_f16<AGPR, false>(acc, rg.f[3], xg_hi<3>(x));
  SCHED_FENCE();
  unit_read<U + 3>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 3) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 3) piece_dma<Q + 1>(pipe);
  extra(gap<3>{});
  SCHED_FENCE();
  mfma_fp8<AGPR>(acc, rg.e[0], xg_x8(x), sc.w, sc.xl);
  SCHED_FENCE();
  unit_read<U + 4>(pipe, rg);
  unit_read<U + 5>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 4) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 4) piece_dma<Q + 1>(pipe);
  extra(gap<4>{});
  SCHED_FENCE();
  mfma_fp8<AGPR>(acc, rg.e[1], xl8, sc.w, sc.xl);
  SCHED_FENCE();
  unit_read<U + 6>(pipe, rg);
  unit_read<U + 7>(pipe, rg);
  if constexpr (MX_DMA_GAP[0] == 5) piece_dma<Q>(pipe);
  if constexpr (MX_DMA_GAP[1] == 5) piece_dma<Q + 1>(pipe);
  if constexpr (G == 1) pipe.template mid_sync<8>();
  extra(gap<5>{});
  SCHED_FENCE();
  if constexpr (G == 3) pipe.advance();
}

// bf16 pair P (0..15) of the current stage on the unit ring: units 2P (hi) and 2P+1 (lo)
template <int P, bool AGPR, class Extra>
__device__ __forceinline__ void pair_step8(Pipe& pipe, Ring8& rg, floatx16& acc, const bf16x8& xhi, const bf16x8& xlo,
                                           Extra extra) {
  constexpr int U = 2 * P;
  const bf16x8 whi = __builtin_bit_cast(bf16x8, ring_get<U & 7>(rg));
  const bf16x8 wlo = __builtin_bit_cast(bf16x8, ring_get<(U + 1) & 7>(rg));
  mfma1<AGPR, true>(acc, wlo, xhi);
  SCHED_FENCE();
  unit_read<U + 1>(pipe, rg);
  extra(gap<0>{});
  SCHED_FENCE();
  mfma1<AGPR, false>(acc, whi, xlo);
  SCHED_FENCE();
  if constexpr (P & 1) piece_dma<((P >> 1) + 4) & 7>(pipe);   // one 1 KiB piece per two pair steps
  extra(gap<1>{});
  SCHED_FENCE();
  mfma1<AGPR, false>(acc, whi, xhi);
  SCHED_FENCE();
  unit_read<U>(pipe, rg);
  if constexpr (P == 7) pipe.template mid_sync<8>();
  extra(gap<2>{});
  SCHED_FENCE();
  if constexpr (P == 15) pipe.advance();
}

__device__ __forceinline__ int cvt_pk_f16(float a, float b) {   // round to nearest even, packed
  int r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float res_lo(int hp, float a) {   // a - f16 in the low half of hp
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(a));
  return r;
}
__device__ __forceinline__ float res_hi(int hp, float a) {   // a - f16 in the high half of hp
  float r;
  asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "v"(a));
  return r;
}

// Packed f32 -> fp8 (e4m3) conversions into the low / high half of a register.  Inline asm on purpose: with the
// builtins hipcc (ROCm 7.2) merged the low-half and the high-half conversion of one register into a single
// instruction with the wrong sources when the value travelled through a vector element in between.
__device__ __forceinline__ int fp8_lo(float a, float b) {          // bytes 0,1 (the upper half keeps whatever it held)
  int r;
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int fp8_hi(int r, float a, float b) {   // bytes 2,3
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2 op_sel:[0,0,1]" : "+v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int fp8s_lo(float a, float b, float inv_scale) {   // fp8(a / inv_scale), fp8(b / inv_scale)
  int r;
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(inv_scale));
  return r;
}
__device__ __forceinline__ int fp8s_hi(int r, float a, float b, float inv_scale) {
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(r) : "v"(a), "v"(b), "s"(inv_scale));
  return r;
}
// The in-place split of an operand group (see XG), ReLU included, as a software pipeline of single VALU ops.
// The 16 "parts" of a group (part = 2 * I + block: step I = 0..7 on the two elements a[I], a[8+I] of block a or
// b[I], b[8+I] of block b) each go through four stages
//     A  r0, r1 = relu of the two elements            B  f16 pair -> a[I] / b[I];  fp8(r0, r1) -> half of a[8+I]
//     C  l0, l1 = r - f16(r)                          D  fp8(l * 2^11) -> half of b[8+I]
// and bundle t = 0..18 runs D(t-3), C(t-2), B(t-1), A(t): the (up to) 7 ops of a bundle do not depend on each other,
// so they issue back to back in an MFMA shadow (the same 7 ops as one dependent chain cost ~10 cycles each there).
// Program order keeps the in-place hazards right: part 2I+1 reads b[8+I] (stage A, bundle 2I+1) before part 2I
// writes its lower half (stage D, bundle 2I+3).
constexpr float XL_INV_SCALE = 1.0f / (float)(1 << XL_SHIFT);
struct CvtMx {
  float r[2][2];   // ReLU'd elements of the parts in stages A..C (two generations)
  float l[2];      // residuals between C and D
};
template <int PART, int STAGE, int J>
__device__ __forceinline__ void mx_uop(XG& x, CvtMx& st) {
  constexpr int I = PART >> 1, blk = PART & 1, gen = PART & 1;
  if constexpr (STAGE == 0) {
    if constexpr (J == 0) st.r[gen][0] = relu1(__int_as_float(blk ? x.b[I] : x.a[I]));
    else st.r[gen][1] = relu1(__int_as_float(blk ? x.b[8 + I] : x.a[8 + I]));
  } else if constexpr (STAGE == 1) {
    if constexpr (J == 0) {
      const int hp = cvt_pk_f16(st.r[gen][0], st.r[gen][1]);
      if constexpr (blk) x.b[I] = hp; else x.a[I] = hp;
    } else {
      if constexpr (blk) x.a[8 + I] = fp8_hi(x.a[8 + I], st.r[gen][0], st.r[gen][1]);
      else x.a[8 + I] = fp8_lo(st.r[gen][0], st.r[gen][1]);
    }
  } else if constexpr (STAGE == 2) {
    const int hp = blk ? x.b[I] : x.a[I];
    if constexpr (J == 0) st.l[0] = res_lo(hp, st.r[gen][0]);
    else st.l[1] = res_hi(hp, st.r[gen][1]);
  } else {
    if constexpr (blk) x.b[8 + I] = fp8s_hi(x.b[8 + I], st.l[0], st.l[1], XL_INV_SCALE);
    else x.b[8 + I] = fp8s_lo(st.l[0], st.l[1], XL_INV_SCALE);
  }
}
struct Uop { int part, stage, j; };
constexpr int MX_UOPS = 16 * 7;
constexpr Uop mx_uop_at(int n) {   // the n-th op of a group's split in pipelined order
  int c = 0;
  for (int t = 0; t < 16 + 3; ++t)
    for (int stage = 3; stage >= 0; --stage) {
      const int part = t - stage;
      if (part < 0 || part > 15) continue;
      for (int j = 0; j < (stage == 3 ? 1 : 2); ++j, ++c)
        if (c == n) return Uop{part, stage, j};
    }
  return Uop{-1, -1, -1};
}
template <int N0, int CNT>
__device__ __forceinline__ void mx_uops(XG& x, CvtMx& st) {   // ops N0 .. N0 + CNT - 1
  static_for<CNT>([&](auto ic) {
    constexpr Uop u = mx_uop_at(N0 + decltype(ic)::value);
    static_assert(u.part >= 0, "op index out of range");
    mx_uop<u.part, u.stage, u.j>(x, st);
  });
}
__device__ __forceinline__ void cvt_mx_all(XG& x) {
  CvtMx st;
  mx_uops<0, MX_UOPS>(x, st);
}
// ops of the split per MFMA shadow of a group_step (16 groups host one split: 7 ops per group)
#ifndef MX_CVT_Q
#define MX_CVT_Q 1, 1, 1, 1, 1, 2
#endif
constexpr int MX_Q[6] = {MX_CVT_Q};
constexpr int mx_q_start(int m) { int c = 0; for (int i = 0; i < m; ++i) c += MX_Q[i]; return c; }
static_assert(mx_q_start(6) == 7, "7 ops per group");

// The same split with the SOURCE in two other accumulator blocks (sa = features 0..31 of the group, sb = 32..63; they
// may live in AGPRs) and the operand group x as a separate destination: used where the source blocks cannot be
// overwritten in place (fold2/conv2 of the all-mixed decoder reads the fold2/conv1 accumulators out of the AGPR file).
// Five stages per part -- S0 read the two elements, S1 ReLU, S2 f16 pair + fp8 of the values, S3 residuals,
// S4 fp8 of the residuals -- and bundle t runs S4(t-4) .. S0(t); three generations of r cover the distance S0 -> S3.
struct CvtMx5 {
  float r[3][2];
  float l[2];
};
template <int PART, int STAGE, int J, class Src>
__device__ __forceinline__ void mx5_uop(const Src& sa, const Src& sb, XG& x, CvtMx5& st) {
  constexpr int I = PART >> 1, blk = PART & 1, gen = PART % 3;
  if constexpr (STAGE == 0) {
    st.r[gen][J] = blk ? sb[J == 0 ? I : 8 + I] : sa[J == 0 ? I : 8 + I];
    asm volatile("" : "+v"(st.r[gen][J]));   // the AGPR read happens here
  } else if constexpr (STAGE == 1) {
    st.r[gen][J] = relu1(st.r[gen][J]);
  } else if constexpr (STAGE == 2) {
    if constexpr (J == 0) {
      const int hp = cvt_pk_f16(st.r[gen][0], st.r[gen][1]);
      if constexpr (blk) x.b[I] = hp; else x.a[I] = hp;
    } else {
      if constexpr (blk) x.a[8 + I] = fp8_hi(x.a[8 + I], st.r[gen][0], st.r[gen][1]);
      else x.a[8 + I] = fp8_lo(st.r[gen][0], st.r[gen][1]);
    }
  } else if constexpr (STAGE == 3) {
    const int hp = blk ? x.b[I] : x.a[I];
    if constexpr (J == 0) st.l[0] = res_lo(hp, st.r[gen][0]);
    else st.l[1] = res_hi(hp, st.r[gen][1]);
  } else {
    if constexpr (blk) x.b[8 + I] = fp8s_hi(x.b[8 + I], st.l[0], st.l[1], XL_INV_SCALE);
    else x.b[8 + I] = fp8s_lo(st.l[0], st.l[1], XL_INV_SCALE);
  }
}
constexpr Uop mx5_uop_at(int n) {
  int c = 0;
  for (int t = 0; t < 16 + 4; ++t)
    for (int stage = 4; stage >= 0; --stage) {
      const int part = t - stage;
      if (part < 0 || part > 15) continue;
      for (int j = 0; j < (stage == 4 ? 1 : 2); ++j, ++c)
        if (c == n) return Uop{part, stage, j};
    }
  return Uop{-1, -1, -1};
}
template <int N0, int CNT, class Src>
__device__ __forceinline__ void mx5_uops(const Src& sa, const Src& sb, XG& x, CvtMx5& st) {
  static_for<CNT>([&](auto ic) {
    constexpr Uop u = mx5_uop_at(N0 + decltype(ic)::value);
    static_assert(u.part >= 0, "op index out of range");
    mx5_uop<u.part, u.stage, u.j>(sa, sb, x, st);
  });
}
// ops of this split per MFMA shadow of a group_step (8 groups host one split: 18 ops per group)
#ifndef MX5_CVT_Q
#define MX5_CVT_Q 2, 2, 2, 2, 5, 5
#endif
constexpr int MX5_Q[6] = {MX5_CVT_Q};
constexpr int mx5_q_start(int m) { int c = 0; for (int i = 0; i < m; ++i) c += MX5_Q[i]; return c; }
static_assert(mx5_q_start(6) == 18, "18 ops per group");

__device__ __forceinline__ int load_scale(const float* sp, int layer) {
  int w = ((const int*)(sp + SP_SC))[layer];
  asm volatile("" : "+v"(w));   // the MFMA takes its scales from VGPRs
  return w;
}

// one stage (4 groups) of fold1/conv3: K = 256 over the resident h2 into one 32-feature block.  The fp8 residual
// operands of groups 2 and 3 are parked in LDS (park: this lane's slot) and fetched one group ahead of their use,
// which takes 16 registers off the kernel's peak.
__device__ __forceinline__ intx8 park_load(const uint4* park, int t) {
  const uint4 lo = park[t * 128], hi = park[t * 128 + 64];
  intx8 r;
  r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
  r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
  return r;
}
__device__ __forceinline__ void park_store(uint4* park, int t, const XG& x) {
  uint4 lo, hi;
  lo.x = (unsigned)x.b[8]; lo.y = (unsigned)x.b[9]; lo.z = (unsigned)x.b[10]; lo.w = (unsigned)x.b[11];
  hi.x = (unsigned)x.b[12]; hi.y = (unsigned)x.b[13]; hi.z = (unsigned)x.b[14]; hi.w = (unsigned)x.b[15];
  park[t * 128] = lo;
  park[t * 128 + 64] = hi;
}
// ReLU + bf16 hi/lo split of one 32x32 accumulator block (16 elements = 8 pairs per lane) for the bf16 triple of
// fold2/conv2, as a software pipeline like the one above.  Pair p goes through
//     S0 read a0, a1 out of the AGPRs    S1 v = relu(a)            S2 hi = bf16 pair of v
//     S3 t = hi expanded to f32          S4 v -= t (exact)          S5 lo = bf16 pair of v
// and bundle t = 0..12 runs S5(t-5) .. S0(t): 9 independent instructions.  Fragments are kept as int vectors
// (element (e & 7) >> 1 of vector e >> 3 holds the pair e, e+1).
typedef __attribute__((ext_vector_type(2))) float floatx2;
struct CvtBf {
  float a[2];
  floatx2 v[4];   // four generations
  floatx2 t;
};
__device__ __forceinline__ int cvt_pk_bf16(float a, float b) {
  int r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int PAIR, int STAGE, int J, class AccT>
__device__ __forceinline__ void bf_uop(const AccT& acc, CvtBf& st, intx4 (&hi)[2], intx4 (&lo)[2]) {
  constexpr int e = 2 * PAIR, g = PAIR & 3;
  if constexpr (STAGE == 0) {
    st.a[J] = acc[e + J];
    asm volatile("" : "+v"(st.a[J]));   // the AGPR read happens here
  } else if constexpr (STAGE == 1) {
    st.v[g][J] = relu1(st.a[J]);
  } else if constexpr (STAGE == 2) {
    hi[e >> 3][(e & 7) >> 1] = cvt_pk_bf16(st.v[g][0], st.v[g][1]);
  } else if constexpr (STAGE == 3) {
    const int hp = hi[e >> 3][(e & 7) >> 1];
    if constexpr (J == 0) st.t[0] = __int_as_float(hp << 16);
    else st.t[1] = __int_as_float(hp & (int)0xffff0000);
  } else if constexpr (STAGE == 4) {
    asm("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(st.v[g]) : "v"(st.t));
  } else {
    lo[e >> 3][(e & 7) >> 1] = cvt_pk_bf16(st.v[g][0], st.v[g][1]);
  }
}
constexpr int BF_UOPS = 8 * 9;
constexpr int bf_stage_ops(int stage) { return (stage == 2 || stage >= 4) ? 1 : 2; }
constexpr Uop bf_uop_at(int n) {
  int c = 0;
  for (int t = 0; t < 8 + 5; ++t)
    for (int stage = 5; stage >= 0; --stage) {
      const int pair = t - stage;
      if (pair < 0 || pair > 7) continue;
      for (int j = 0; j < bf_stage_ops(stage); ++j, ++c)
        if (c == n) return Uop{pair, stage, j};
    }
  return Uop{-1, -1, -1};
}
template <int N0, int CNT, class AccT>
__device__ __forceinline__ void bf_uops(const AccT& acc, CvtBf& st, intx4 (&hi)[2], intx4 (&lo)[2]) {
  static_for<CNT>([&](auto ic) {
    constexpr Uop u = bf_uop_at(N0 + decltype(ic)::value);
    static_assert(u.part >= 0, "op index out of range");
    bf_uop<u.part, u.stage, u.j>(acc, st, hi, lo);
  });
}
// instructions of the split per MFMA shadow of a pair step: [pair parity][gap]; 9 per two pair steps
#ifndef BF_CVT_Q
#define BF_CVT_Q {2, 2, 1}, {2, 1, 1}
#endif
constexpr int BF_Q[2][3] = {BF_CVT_Q};
constexpr int bf_q_start(int P, int gap) {   // first op of pair step P's shadow `gap`
  int c = 0;
  for (int p = 0; p < P; ++p) c += BF_Q[p & 1][0] + BF_Q[p & 1][1] + BF_Q[p & 1][2];
  for (int i = 0; i < gap; ++i) c += BF_Q[P & 1][i];
  return c;
}
static_assert(bf_q_start(16, 0) == BF_UOPS, "72 instructions per block");

template <class Tail>
__device__ __forceinline__ void l3_stage_mx(Pipe& pipe, Ring8& rg, intx16& acc3, const XG (&h2)[4], const uint4* park,
                                            const Scales& sc, Tail tail) {
  intx8 p2, p3;
  group_step<0, false>(pipe, rg, acc3, h2[0], sc, [](auto) {});
  group_step<1, false>(pipe, rg, acc3, h2[1], sc, [&](auto mm) {
    if constexpr (decltype(mm)::value == 1) p2 = park_load(park, 0);
  });
  group_step_x<2, false>(pipe, rg, acc3, h2[2], p2, sc, [&](auto mm) {
    if constexpr (decltype(mm)::value == 1) p3 = park_load(park, 1);
  });
  group_step_x<3, false>(pipe, rg, acc3, h2[3], p3, sc, [&](auto mm) {
    if constexpr (decltype(mm)::value == 4) tail();   // e.g. the bias of the next block, one long shadow ahead
  });
}

// ---- image term of fold2/conv1 as a matrix product (local stream of the mixed kernels) ----------------------------
// A wave's 32 points are an 8 x 4 patch of one grid plane, and their projections fall into a small rectangle of
// pixels (4 x 4 on average at sdf_res 256).  Instead of gathering the four corner rows of every point (256 load
// instructions per wave that each touch 64 different cache lines: ~28 cycles apiece on the address path), the blend is
// written as  acc[512 features x 32 points] = Pmap^T[features x pixels] . S[pixels x points]  with S the sparse matrix of
// corner weights (<= 4 non-zeros per column), in chunks of 4 x 4 pixels = one K=16 bf16 MFMA step: lane (r, h) loads
// Pmap[pixel 8h + j][32 nb + r] for j = 0..7 -- 128 contiguous bytes per half wave -- which IS the A-operand layout,
// splits the eight values into bf16 hi + lo, and three products (lo.hi, hi.lo, hi.hi; fp32 accumulate, ~2^-17 relative,
// as in the bf16 triple) land in the accumulator block.  Pixel rows beyond the image carry weight 0 and are read from
// the zeroed tail of the Pmap buffer (PMAP_ROWS).  Waves whose rectangle needs more than BLEND_MAX_CHUNKS chunks
// (scattered points, very coarse grids) take the gather instead.  Every pixel of the rectangle enters every point's sum,
// most with weight 0: a non-finite map entry (features that overflowed fp32) spoils the wave's 32 points, not only the
// points that sample it.
#ifndef BLEND_MAX_CHUNKS
#define BLEND_MAX_CHUNKS 12
#endif
constexpr int PMAP_ROWS = (IMG + 4) * IMG + 8;   // pixel rows of the Pmap buffer incl. the zeroed tail
__device__ __forceinline__ void split_bf16x8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  intx4 h, l;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int hp = cvt_pk_bf16(v[2 * q], v[2 * q + 1]);
    const float r0 = __fsub_rn(v[2 * q], __int_as_float(hp << 16));
    const float r1 = __fsub_rn(v[2 * q + 1], __int_as_float(hp & (int)0xffff0000));
    h[q] = hp;
    l[q] = cvt_pk_bf16(r0, r1);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}
// min / max over lanes 0..31 of a wave whose lanes 32..63 repeat them, by DPP (five steps; ds_bpermute shuffles cost a
// dependent LDS round trip each): quads, half rows, rows, then row 0 into row 1; the result is read off lane 31.
template <bool MAX>
__device__ __forceinline__ int wave32_reduce(int v) {
#define DISN_RED_STEP(ctrl, rows) { const int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xf, false); v = MAX ? max(v, o) : min(v, o); }
  DISN_RED_STEP(0xB1, 0xf)    // quad_perm [1,0,3,2]
  DISN_RED_STEP(0x4E, 0xf)    // quad_perm [2,3,0,1]
  DISN_RED_STEP(0x141, 0xf)   // row_half_mirror
  DISN_RED_STEP(0x140, 0xf)   // row_mirror
  DISN_RED_STEP(0x142, 0xa)   // row_bcast15 into rows 1, 3
#undef DISN_RED_STEP
  return __builtin_amdgcn_readlane(v, 31);
}
// The split of one block's eight values (four pairs) as 20 single VALU ops in stage-major order -- hi pairs (RNE), their
// two expansions, the exact packed subtraction, lo pairs -- so that neighbours in the instruction stream are independent
// and a third of them fits the shadow of each of the previous block's three MFMAs.
struct BlendSplit {
  floatx2 e[4], r[4];
  intx4 hi, lo;
};
template <int OP>
__device__ __forceinline__ void blend_split_op(const floatx2 (&v)[4], BlendSplit& st) {
  constexpr int stage = OP >> 2, q = OP & 3;
  if constexpr (stage == 0) st.hi[q] = cvt_pk_bf16(v[q][0], v[q][1]);
  else if constexpr (stage == 1) st.e[q][0] = __int_as_float(st.hi[q] << 16);
  else if constexpr (stage == 2) st.e[q][1] = __int_as_float(st.hi[q] & (int)0xffff0000);
  else if constexpr (stage == 3) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(st.r[q]) : "v"(v[q]), "v"(st.e[q]));
  else st.lo[q] = cvt_pk_bf16(st.r[q][0], st.r[q][1]);
}
template <int N0, int N1>
__device__ __forceinline__ void blend_split_ops(const floatx2 (&v)[4], BlendSplit& st) {
  static_for<N1 - N0>([&](auto ic) { blend_split_op<N0 + decltype(ic)::value>(v, st); });
}
__device__ __forceinline__ void blend_mfma(floatx16 (&acc4)[16], const float* __restrict__ pmap, const Corners& cn,
                                           int ifx, int ify, int x0, int y0, int ncx, int nchunks, int lane) {
  const int l32 = lane & 31, hl = lane >> 5;
  floatx2 t[16][4];   // t[block][pair]: the lane's eight pixels of the chunk, feature 32 block + l32
  // byte offsets of the eight pixels from the chunk's first (rows are IMG pixels apart, pixels 2 KiB)
  // 32-bit byte offsets from the (uniform) map base: the map is 38 MB
  auto load_block = [&](floatx2 (&dst)[4], unsigned src, int nb4, int j0, int j1) {
#pragma unroll
    for (int j = j0; j < j1; ++j)
      dst[j >> 1][j & 1] = *(const float*)((const char*)pmap + (src + (unsigned)(((j >> 2) * IMG + (j & 3)) * 2048 + 128 * nb4)));
  };
  auto chunk_src = [&](int cx, int cy) {
    return (unsigned)(((y0 + 4 * cy + 2 * hl) * IMG + x0 + 4 * cx) * 2048 + 4 * l32);
  };
  {
    const unsigned src = chunk_src(0, 0);
#pragma unroll
    for (int nb4 = 0; nb4 < 16; ++nb4) load_block(t[nb4], src, nb4, 0, 8);
  }
  // the accumulators start at zero (the map carries the layer's bias); written while the first loads are in flight
#pragma unroll
  for (int nb4 = 0; nb4 < 16; ++nb4) {
#pragma unroll
    for (int e = 0; e < 16; ++e) acc4[nb4][e] = 0.0f;
    asm volatile("" : "+a"(acc4[nb4]));
  }
  int cx = 0, cy = 0;
  auto chunk = [&](auto prefetch) {
    constexpr bool PREFETCH = decltype(prefetch)::value;
    // S: the corner weights of this lane's point for the lane's eight pixels of the chunk
    float sv[8];
    const int ry0 = y0 + 4 * cy + 2 * hl - ify, rx0 = x0 + 4 * cx - ifx;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int ry = ry0 + rr;
      const float sf = ry == 0 ? cn.w[0] : (ry == 1 ? cn.w[2] : 0.0f);   // column fx: (fy,fx) / (cy,fx)
      const float sc = ry == 0 ? cn.w[3] : (ry == 1 ? cn.w[1] : 0.0f);   // column cx: (fy,cx) / (cy,cx)
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int rx = rx0 + cc;
        sv[4 * rr + cc] = rx == 0 ? sf : (rx == 1 ? sc : 0.0f);
      }
    }
    bf16x8 shi, slo;
    split_bf16x8(sv, shi, slo);
    if (++cx == ncx) { cx = 0; ++cy; }
    const unsigned nsrc = chunk_src(cx, cy);
    BlendSplit sp[2];
    blend_split_ops<0, 20>(t[0], sp[0]);
    static_for<16>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      BlendSplit& cu = sp[b & 1];
      BlendSplit& nx = sp[(b + 1) & 1];
      const bf16x8 ahi = __builtin_bit_cast(bf16x8, cu.hi), alo = __builtin_bit_cast(bf16x8, cu.lo);
      SCHED_FENCE();
      mfma1<true, true>(acc4[b], alo, shi);
      SCHED_FENCE();
      if constexpr (b < 15) blend_split_ops<0, 7>(t[b + 1], nx);
      if constexpr (PREFETCH) load_block(t[b], nsrc, b, 0, 3);
      SCHED_FENCE();
      mfma1<true, true>(acc4[b], ahi, slo);
      SCHED_FENCE();
      if constexpr (b < 15) blend_split_ops<7, 14>(t[b + 1], nx);
      if constexpr (PREFETCH) load_block(t[b], nsrc, b, 3, 6);
      SCHED_FENCE();
      mfma1<true, true>(acc4[b], ahi, shi);
      SCHED_FENCE();
      if constexpr (b < 15) blend_split_ops<14, 20>(t[b + 1], nx);
      if constexpr (PREFETCH) load_block(t[b], nsrc, b, 6, 8);
    });
    SCHED_FENCE();
  };
#pragma unroll 1
  for (int c = 1; c < nchunks; ++c) chunk(std::true_type{});
  chunk(std::false_type{});
}

template <bool LOCAL, bool L4MX>
__device__ __forceinline__ float stream_pass_mx(Pipe& pipe, Ring8& rg, const float* sp,
                                                const float* b4 /*LDS: global stream's fold2/conv1 bias*/, uint4* park,
                                                const float* __restrict__ pmap, float px, float py, float pz, float u,
                                                float v) {
  // half_of() = lane >> 5, recomputed from the DMA offset register at every use: addresses derived from a long-lived copy
  // cost registers that this kernel does not have
  auto half_of = [&]() {
    unsigned vo = pipe.voff;   // lane * 16 in this kernel
    asm volatile("" : "+v"(vo));
    return (int)(vo >> 9);
  };
  // fold2/conv1 accumulators: bias (+ image term), exactly as in the bf16 kernel
  floatx16 acc4[16];
  Corners cn;
  if constexpr (LOCAL) {
    asm volatile("" : "+v"(u), "+v"(v));
    cn = corners(u, v);
  }
  if constexpr (LOCAL) {
    // the wave's pixel rectangle (u, v lie in [0, 136]: project() clamps)
    const int ifx = (int)floorf(u), ify = (int)floorf(v);
    const int x0 = wave32_reduce<false>(ifx), x1 = wave32_reduce<true>(ifx) + 1;
    const int y0 = wave32_reduce<false>(ify), y1 = wave32_reduce<true>(ify) + 1;
    const int ncx = (x1 - x0 + 4) >> 2, nchunks = ncx * ((y1 - y0 + 4) >> 2);
    if (nchunks <= BLEND_MAX_CHUNKS) {
      unsigned vo = pipe.voff;   // lane * 16; re-derived here so that nothing lane-dependent is hoisted out of the tile loop
      asm volatile("" : "+v"(vo));
      blend_mfma(acc4, pmap, cn, ifx, ify, x0, y0, ncx, nchunks, (int)(vo >> 4));
    } else {
    floatx4 t[2][16];
    auto gather = [&](floatx4 (&dst)[16], int nb4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* src = pmap + (size_t)cn.pix[c] * 512 + 32 * nb4 + 4 * half_of();
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[4 * c + q] = *(const floatx4*)(src + 8 * q);
      }
    };
    gather(t[0], 0);
    static_for<16>([&](auto nc) {
      constexpr int nb4 = decltype(nc)::value;
      constexpr int tb = nb4 & 1;
      if constexpr (nb4 + 1 < 16) gather(t[(nb4 + 1) & 1], nb4 + 1);
      // the blend of the four corners (fold2/conv1's bias was added to the map by disn_sdf_set_image)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc4[nb4][4 * q + e] = cn.w[0] * t[tb][q][e];
#pragma unroll
      for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc4[nb4][4 * q + e] = fmaf(cn.w[c], t[tb][4 * c + q][e], acc4[nb4][4 * q + e]);
      asm volatile("" : "+a"(acc4[nb4]));
    });
    }
  } else {
#pragma unroll
    for (int nb4 = 0; nb4 < 16; ++nb4) {
      acc4[nb4] = bias_init(b4, 32 * nb4, half_of());
      asm volatile("" : "+a"(acc4[nb4]));
    }
  }
  if constexpr (LOCAL) {
    // The fragment ring (32 registers: the first eight units of the tile's first stage, read at the end of the previous
    // tile) is read again here instead of being carried through the image term, which wants the registers
    static_for<8>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value;
      ring_put<S>(rg, pipe.cur[S * 64]);
    });
  }
  asm volatile("; disn:pipelined_region_begin" ::: "memory");
  int xl_scale = 127 - XL_SHIFT;
  asm volatile("" : "+v"(xl_scale));
  // fold1/conv1 (3 -> 64) on the VALU, laid out like two accumulator blocks and split in place.  All LDS reads of a phase
  // are issued as one batch from one base address, fenced ahead of their uses: left to itself, hipcc at the register
  // limit issues each read just before its use and waits out the LDS latency 32 times (4 K cycles per tile and stream
  // in fold1/conv2's bias reads alone, tools/seg_patch_mx4.py)
  XG x1;
  XG h2[4];
  {
    const float* w1 = sp + SP_W1 + 16 * half_of();
    const float* b2 = sp + SP_B2 + 4 * half_of();
    floatx4 wa[16], wb[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const int f = (g & 3) + 8 * (g >> 2);
      wa[g] = *(const floatx4*)(w1 + 4 * f);
      wb[g] = *(const floatx4*)(w1 + 4 * (32 + f));
    }
    SCHED_FENCE();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      x1.a[g] = __float_as_int(fmaf(pz, wa[g][2], fmaf(py, wa[g][1], px * wa[g][0])) + wa[g][3]);
      x1.b[g] = __float_as_int(fmaf(pz, wb[g][2], fmaf(py, wb[g][1], px * wb[g][0])) + wb[g][3]);
    }
    SCHED_FENCE();
    // fold1/conv2's bias into its eight accumulator blocks (h2[2 st + p].a / .b = block 4 st + 2 p / + 1); the reads land
    // while x1 is split
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
      intx16& dst = (blk & 1) ? h2[blk >> 1].b : h2[blk >> 1].a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const floatx4 v = *(const floatx4*)(b2 + 32 * blk + 8 * q);
        dst[4 * q + 0] = __float_as_int(v[0]); dst[4 * q + 1] = __float_as_int(v[1]);
        dst[4 * q + 2] = __float_as_int(v[2]); dst[4 * q + 3] = __float_as_int(v[3]);
      }
    }
    SCHED_FENCE();
  }
  cvt_mx_all(x1);
  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks (one group each); h2 stays resident as 4 operand groups
  {
    const Scales sc = {load_scale(sp, 0), xl_scale};
    asm volatile("s_nop 7" ::: "memory");   // the split of h1 was written by VALU conversions just above
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      group_step<0, false>(pipe, rg, h2[2 * st].a, x1, sc, [](auto) {});
      group_step<1, false>(pipe, rg, h2[2 * st].b, x1, sc, [](auto) {});
      group_step<2, false>(pipe, rg, h2[2 * st + 1].a, x1, sc, [](auto) {});
      group_step<3, false>(pipe, rg, h2[2 * st + 1].b, x1, sc, [](auto) {});
      drain_v(h2[2 * st].b);
      cvt_mx_all(h2[2 * st]);
      drain_v(h2[2 * st + 1].b);
      drain_v(h2[2 * st + 1].b);
      cvt_mx_all(h2[2 * st + 1]);
    }
    park_store(park, 0, h2[2]);   // from here on b[8..15] of h2[2], h2[3] are dead in registers
    park_store(park, 1, h2[3]);
  }
  // fold1/conv3 (256 -> 512) chained into fold2/conv1 in groups of 64 features: while group m of h3 feeds the
  // sixteen fold2/conv1 accumulators (4 stages), blocks 2m+2, 2m+3 have already been accumulated (2 stages) and
  // their in-place split fills the shadows of the K=64 MFMAs.
  const Scales sc3 = {load_scale(sp, 1), xl_scale}, sc4 = {load_scale(sp, 2), xl_scale};
  XG xc;
  xc.a = as_bits(bias_init(sp + SP_B3, 0, half_of()));
  asm volatile("s_nop 7" ::: "memory");
  l3_stage_mx(pipe, rg, xc.a, h2, park, sc3, [&] { xc.b = as_bits(bias_init(sp + SP_B3, 32, half_of())); });
  l3_stage_mx(pipe, rg, xc.b, h2, park, sc3, [] {});
  drain_v(xc.b);
  drain_v(xc.b);
  cvt_mx_all(xc);
  asm volatile("s_nop 7" ::: "memory");
  // Two operand sets in ping-pong.  The chain is a loop of three double steps (m = 0..5) plus a tail (m = 6, 7)
  // so that the hot code stays small.  Step m: blocks 2m+2, 2m+3 of h3 are accumulated into nx (2 stages), then cu
  // feeds the sixteen fold2/conv1 accumulators (4 stages) while nx is split in the MFMA shadows.  The bias of a
  // block is read into its accumulator ahead of the stage that needs it: nx.b's in the last long shadow of nx.a's stage,
  // and the one of the NEXT step's first block into cu.a in the last two shadows of this step, where cu.a is dead
  // (k-steps 0, 1 were read by MFMAs 0, 1; the fp8 operand a[8..15] by MFMA 4, which has finished when MFMA 5,
  // chained on the same accumulator, has issued).
  XG xd;
  xd.a = as_bits(bias_init(sp + SP_B3, 64, half_of()));
  auto chain = [&](XG& cu, XG& nx, const int m, auto with_next, auto with_pre, auto tail) {
    constexpr bool NEXT = decltype(with_next)::value, PRE = decltype(with_pre)::value;
    if constexpr (NEXT) {
      l3_stage_mx(pipe, rg, nx.a, h2, park, sc3, [&] { nx.b = as_bits(bias_init(sp + SP_B3, 64 * m + 96, half_of())); });
      l3_stage_mx(pipe, rg, nx.b, h2, park, sc3, [] {});
    }
    CvtMx cs;
    const float* bnext = sp + SP_B3 + 64 * m + 128;
    static_for<4>([&](auto sc_) {
      constexpr int S = decltype(sc_)::value;
      static_for<4>([&](auto gc) {
        constexpr int G = decltype(gc)::value;
        constexpr int T = 4 * S + G;          // group T of the phase hosts ops 7T .. 7T+6 of nx's split
        group_step<G, true>(pipe, rg, acc4[T], cu, sc4, [&](auto mm) {
          constexpr int M = decltype(mm)::value;
          if constexpr (NEXT) mx_uops<7 * T + mx_q_start(M), MX_Q[M]>(nx, cs);
          tail(std::integral_constant<int, T>{}, mm);
          if constexpr (PRE && T == 15 && (M == 4 || M == 5)) {
            const int h4 = 4 * half_of();
#pragma unroll
            for (int q = 2 * (M - 4); q < 2 * (M - 4) + 2; ++q) {
              const floatx4 v = *(const floatx4*)(bnext + 8 * q + h4);
              cu.a[4 * q + 0] = __float_as_int(v[0]); cu.a[4 * q + 1] = __float_as_int(v[1]);
              cu.a[4 * q + 2] = __float_as_int(v[2]); cu.a[4 * q + 3] = __float_as_int(v[3]);
            }
          }
        });
      });
    });
  };
  auto no_tail = [](auto, auto) {};
#pragma unroll 1
  for (int mm = 0; mm < 3; ++mm) {
    chain(xc, xd, 2 * mm, std::true_type{}, std::true_type{}, no_tail);
    chain(xd, xc, 2 * mm + 1, std::true_type{}, std::true_type{}, no_tail);
  }
  chain(xc, xd, 6, std::true_type{}, std::false_type{}, no_tail);
  floatx16 acc5[8];
  if constexpr (L4MX) {
    // fold2/conv2 (512 -> 256) in the same mixed arithmetic: the fold2/conv1 accumulators are read out of the AGPR file
    // two blocks (64 features) at a time into an operand group (ping-pong ya / yb), whose split fills the MFMA
    // shadows of the previous group's eight group steps.  Blocks 0, 1 are final after the first stage of the last
    // chain step and are split in the shadows of its last two stages (xc is dead there).
    XG ya, yb;
    CvtMx5 c5;
    chain(xd, xc, 7, std::false_type{}, std::false_type{}, [&](auto tc, auto mc) {
      constexpr int T = decltype(tc)::value, M = decltype(mc)::value;
      if constexpr (T >= 8) mx5_uops<18 * (T - 8) + mx5_q_start(M), MX5_Q[M]>(acc4[0], acc4[1], ya, c5);
    });
    const Scales sc5 = {load_scale(sp, 3), xl_scale};
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc5[nb] = bias_init(sp + SP_B5, 32 * nb, half_of());
    asm volatile("s_nop 7" ::: "memory");
    static_for<8>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      XG& cu = (m & 1) ? yb : ya;
      XG& nx = (m & 1) ? ya : yb;
      static_for<8>([&](auto nc) {
        constexpr int nb5 = decltype(nc)::value;
        group_step<(nb5 & 3), false>(pipe, rg, acc5[nb5], cu, sc5, [&](auto mm) {
          constexpr int M = decltype(mm)::value;
          if constexpr (m < 7)
            mx5_uops<18 * nb5 + mx5_q_start(M), MX5_Q[M]>(acc4[m < 7 ? 2 * m + 2 : 0], acc4[m < 7 ? 2 * m + 3 : 0], nx, c5);
        });
      });
    });
  } else {
  chain(xd, xc, 7, std::false_type{}, std::false_type{}, no_tail);
  // fold2/conv2 (512 -> 256) on the bf16 triple, consuming the fold2/conv1 accumulators block by block (unrolled:
  // the accumulator file cannot be indexed by a loop counter)
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) acc5[nb] = bias_init(sp + SP_B5, 32 * nb, half_of());
  intx4 yhi[2][2], ylo[2][2];
  CvtBf cs5;
  drain_a(acc4[0]);
  bf_uops<0, BF_UOPS>(acc4[0], cs5, yhi[0], ylo[0]);
  static_for<16>([&](auto nc) {
    constexpr int nb4 = decltype(nc)::value;
    constexpr int cur = nb4 & 1, nx = cur ^ 1, nn = nb4 < 15 ? nb4 + 1 : 15;
    static_for<16>([&](auto pc) {
      constexpr int P = decltype(pc)::value;
      pair_step8<P, false>(pipe, rg, acc5[P >> 1], __builtin_bit_cast(bf16x8, yhi[cur][P & 1]),
                           __builtin_bit_cast(bf16x8, ylo[cur][P & 1]), [&](auto mm) {
        constexpr int M = decltype(mm)::value;
        if constexpr (nb4 < 15) bf_uops<bf_q_start(P, M), BF_Q[P & 1][M]>(acc4[nn], cs5, yhi[nx], ylo[nx]);
      });
    });
  });
  }
  // fold2/conv3 (256 -> 1) on the VALU: the lane's 128 weights in two batches of 16 reads (see fold1/conv1), four
  // independent partial sums
  drain_v(acc5[7]);
  float part;
  {
    const float* w6 = sp + SP_W6 + 4 * half_of();
    float ps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    static_for<2>([&](auto hc) {
      constexpr int H = decltype(hc)::value;
      floatx4 w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) w[r] = *(const floatx4*)(w6 + 32 * (4 * H + (r >> 2)) + 8 * (r & 3));
      SCHED_FENCE();
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) ps[e] = fmaf(relu1(acc5[4 * H + (r >> 2)][4 * (r & 3) + e]), w[r][e], ps[e]);
      SCHED_FENCE();
    });
    part = (ps[0] + ps[1]) + (ps[2] + ps[3]);
  }
  part += __shfl_xor(part, 32, 64);
  return part + sp[SP_B6];
}

// One launch per stream: LOCAL = false evaluates the global stream and stores its raw value g, LOCAL = true the local
// stream and stores g + l (reference models/model_normalization.py:204), scaled if asked.  Each launch walks one
// stream's 2.1 MiB of fragments, which stay in an XCD's 4 MiB L2; both streams in one launch (4.2 MiB, cyclic) missed
// on every line and drew the whole stream from the Infinity Cache for each round of tiles.
template <bool LOCAL, bool L4MX>
__global__ __launch_bounds__(256, 1) void sdf_decode_mx(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");   // FP16_OVFL: saturating f16 / fp8 conversions
  float* sp_lds = (float*)(lds + NSLOT * STAGE_BYTES);
  for (int i = threadIdx.x; i < 2 * SP_MX_LDS; i += 256)
    sp_lds[i] = a.sp[(i < SP_MX_LDS ? 0 : SP_SIZE - SP_MX_LDS) + i];   // both streams' blocks without their B4 tails
  for (int i = threadIdx.x; i < 512; i += 256) sp_lds[2 * SP_MX_LDS + i] = a.sp[SP_B4 + i];   // global stream's B4 (+ hoisted term)
  Geom* geom = (Geom*)(lds + NSLOT * STAGE_BYTES + MX_SP_BYTES + 4 * MX_PARK_BYTES_PER_WAVE);
  geom_store(geom, a);
  __syncthreads();
  uint4* park = (uint4*)(lds + NSLOT * STAGE_BYTES + MX_SP_BYTES + (threadIdx.x >> 6) * MX_PARK_BYTES_PER_WAVE) + (threadIdx.x & 63);
  const long long ntiles = a.ntiles;
  Pipe pipe;
  pipe.ws = (const char*)a.ws + (LOCAL ? (size_t)STAGES_PER_STREAM * STAGE_BYTES : 0);   // [global stream][local stream]
  pipe.ws_end = pipe.ws + (size_t)STAGES_PER_STREAM * STAGE_BYTES;
  pipe.src_next = pipe.ws;
  pipe.ring = lds;
  pipe.ring32 = (unsigned)(size_t)(lds_ptr_t)lds;
  pipe.g = 0;
  pipe.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  pipe.lane = threadIdx.x & 63;
  pipe.voff = pipe.lane * 16;   // 1 KiB LDS-DMA pieces: 16 bytes per lane
  pipe.dma_setup(0); pipe.dma4_range<0, 4>(); pipe.dma_hi(); pipe.dma4_range<4, 8>();
  pipe.dma_setup(1); pipe.dma4_range<0, 4>(); pipe.dma_hi(); pipe.dma4_range<4, 8>();
  pipe.dma_setup(2); pipe.dma4_range<0, 4>(); pipe.dma_hi();   // second half: groups 0, 1 of stage 0
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // stage 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  pipe.cur = pipe.slot_ptr(0);
  pipe.nxt = pipe.slot_ptr(1);
  Ring8 rg;
  static_for<8>([&](auto sc_) {
    constexpr int S = decltype(sc_)::value;
    ring_put<S>(rg, pipe.cur[S * 64]);
  });
  long long tile_first = blockIdx.x, tile_end = ntiles, tile_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {   // XCD-aware dealing, as in sdf_decode
    const long long per = (ntiles + 7) / 8;
    const int xcd = blockIdx.x & 7;
    tile_first = xcd * per + (blockIdx.x >> 3);
    tile_end = min(ntiles, (xcd + 1) * per);
    tile_step = gridDim.x >> 3;
  }
  for (long long tile = tile_first; tile < tile_end; tile += tile_step) {
    float pred;
    {
      float x, y, z, xr, yr, zr;
      if (a.grid_mode) {
        int i, j, k;
        grid_point(geom, (unsigned)tile, pipe.wave, pipe.lane & 31, i, j, k);
        grid_xyz(geom, i, j, k, x, y, z);
        xr = x; yr = y; zr = z;
      } else {
        long long p = tile * 128 + pipe.wave * 32 + (pipe.lane & 31);
        if (p >= a.n) p = a.n - 1;
        x = a.pts[3 * p]; y = a.pts[3 * p + 1]; z = a.pts[3 * p + 2];
        xr = a.pts_rot[3 * p]; yr = a.pts_rot[3 * p + 1]; zr = a.pts_rot[3 * p + 2];
      }
      float u, v;
      project(geom->T, x, y, z, u, v);
      pred = stream_pass_mx<LOCAL, L4MX>(pipe, rg, sp_lds + (LOCAL ? SP_MX_LDS : 0), sp_lds + 2 * SP_MX_LDS, park, a.pmap,
                                   xr, yr, zr, u, v);
    }
    long long p = tile * 128 + pipe.wave * 32 + (pipe.lane & 31);
    bool valid = p < a.n;
    if (a.grid_mode) {
      int i, j, k;
      valid = grid_point(geom, (unsigned)tile, pipe.wave, pipe.lane & 31, i, j, k);
      p = ((long long)k * geom->n1 + j) * geom->n1 + i;
    }
    if (valid && pipe.lane < 32) {
      if constexpr (LOCAL) {
        pred = __fadd_rn(a.out[p], pred);   // g + l
        a.out[p] = a.scale_out ? (float)((double)pred / 10.0) : pred;
      } else {
        a.out[p] = pred;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

struct SdfState {
  DevBuf ws, wsmx, wsmx4, sp, f137, pmap, wimg, wglob, b4g, b4l, tmat;
  DevBuf wraw, gfeat, chk;   // selfcheck: the fp32 weights as given, the image's global feature, work buffers
  size_t wraw_off[24] = {};  // float offsets of the 24 arrays (DisnDecoderWeights order) inside wraw
  int num_cu = 256;
  int decoder = DISN_DECODER_MIXED;
  bool have_weights = false, have_image = false, lds_attr_set = false, lds_attr_set_mx = false;
  bool mixed_ok = false;      // the mixed streams could be packed (MFMA-layer weights of the first three layers inside the fp16 range)
  bool mixed4_ok = false;     // ... and fold2/conv2's as well
  float mixed_bad_max = 0;    // ... else the offending max |w|
};

void disn_sdf_state_free(DisnCtx* ctx) {
  if (!ctx->sdf) return;
  SdfState* s = ctx->sdf;
  DevBuf* all[] = {&s->ws, &s->wsmx, &s->wsmx4, &s->sp, &s->f137, &s->pmap, &s->wimg, &s->wglob, &s->b4g, &s->b4l, &s->tmat,
                   &s->wraw, &s->gfeat, &s->chk};
  for (DevBuf* b : all) disn_release(*b);
  delete s;
  ctx->sdf = nullptr;
}

static SdfState* sdf_state(DisnCtx* ctx) {
  if (!ctx->sdf) {
    ctx->sdf = new SdfState();
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cu > 0)
      ctx->sdf->num_cu = cu;
  }
  return ctx->sdf;
}

static inline uint16_t bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// one fragment pair of W[K][N] (row-major [Cin][Cout], rows krow0.. = this layer's
// input features): n-block nb, k-step ks.  Lane (r,h) element j holds
// W[32(ks>>1) + 16(ks&1) + 8(j>>2) + 4h + (j&3)][32nb + r]  (the accumulator-as-operand k order).
static void pack_pair(std::vector<uint16_t>& out, const float* W, int N, int nb, int ks) {
  size_t base = out.size();
  out.resize(base + 1024);
  for (int lane = 0; lane < 64; ++lane) {
    int r = lane & 31, h = lane >> 5;
    for (int j = 0; j < 8; ++j) {
      int k = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
      float w = W[(size_t)k * N + 32 * nb + r];
      uint16_t hi = bf16_rne(w);
      uint16_t lo = bf16_rne(w - bf16_to_f32(hi));
      out[base + lane * 8 + j] = hi;
      out[base + 512 + lane * 8 + j] = lo;
    }
  }
}

// ---- fragment stream of the mixed-precision kernel (units of 1 KiB, see sdf_decode_mx) ----------------
static inline uint16_t f16_bits(float f) {
  _Float16 h = (_Float16)f;   // round to nearest even
  uint16_t b;
  memcpy(&b, &h, 2);
  return b;
}
static inline float f16_value(float f) { return (float)(_Float16)f; }
// OCP fp8 e4m3 (bias 7, max 448, no infinities), round to nearest even, saturating
static inline uint8_t e4m3_bits(float v) {
  if (v != v) return 0x7f;
  const uint8_t sgn = signbit(v) ? 0x80 : 0;
  float a = fabsf(v);
  if (a >= 448.0f) return sgn | 0x7e;
  if (a < 0.015625f) {   // below 2^-6: subnormal, step 2^-9
    int q = (int)nearbyintf(a * 512.0f);
    return sgn | (uint8_t)q;   // q == 8 is the smallest normal, whose code is 0x08 as well
  }
  int e;
  float m = frexpf(a, &e);   // a = m * 2^e, m in [0.5, 1)
  int q = (int)nearbyintf((m * 2.0f - 1.0f) * 8.0f);
  int E = e - 1 + 7;
  if (q == 8) { q = 0; ++E; }
  if (E > 15 || (E == 15 && q == 7)) return sgn | 0x7e;
  return sgn | (uint8_t)(E << 3) | (uint8_t)q;
}
// largest power-of-two exponent a with maxabs * 2^a <= 448
static inline int fp8_scale_exp(float maxabs) {
  if (!(maxabs > 0.0f)) return 0;
  int a = (int)floorf(log2f(448.0f / maxabs));
  while (ldexpf(maxabs, a) > 448.0f) --a;
  while (ldexpf(maxabs, a + 1) <= 448.0f) ++a;
  return a < -100 ? -100 : a > 100 ? 100 : a;
}
// one group: 32 output features (block nb) x 64 input features (k group kg) of W[K][N]
static void pack_group(std::vector<uint8_t>& out, const float* W, int N, int nb, int kg, int a_w) {
  const size_t base = out.size();
  out.resize(base + 8192);
  uint8_t* o = out.data() + base;
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int s = 0; s < 4; ++s)      // units 0..3: f16(W), k-step s; element j <-> the in-place operand order (XG)
      for (int j = 0; j < 8; ++j) {
        const int k = 64 * kg + mx_feature(mx_hi_elem(s, j), h);
        const uint16_t b = f16_bits(W[(size_t)k * N + 32 * nb + r]);
        memcpy(o + 1024 * s + lane * 16 + 2 * j, &b, 2);
      }
    for (int j = 0; j < 32; ++j) {   // units 4,5: fp8 of the f16 residual; units 6,7: fp8 of W; byte j of the lane's 32
      const int k = 64 * kg + mx_feature(mx_fp8_elem(j), h);
      const float w = W[(size_t)k * N + 32 * nb + r];
      const float wl = w - f16_value(w);
      o[4096 + 1024 * (j >> 4) + lane * 16 + (j & 15)] = e4m3_bits(ldexpf(wl, a_w + 11));   // |wl| <= 2^-11 |w|: no overflow
      o[6144 + 1024 * (j >> 4) + lane * 16 + (j & 15)] = e4m3_bits(ldexpf(w, a_w));
    }
  }
}
// scale exponents of one layer: rows 0..K-1 of W[.][N] (finite: checked by disn_sdf_set_weights)
static void layer_scale(const float* W, int K, int N, int* a_w, float* maxabs) {
  float mw = 0.0f;
  for (size_t i = 0; i < (size_t)K * N; ++i) mw = fmaxf(mw, fabsf(W[i]));
  *a_w = fp8_scale_exp(mw);
  *maxabs = mw;
}
static bool all_finite(const float* p, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (!(p[i] - p[i] == 0.0f)) return false;   // NaN and +-inf both fail
  return true;
}

static int upload(DisnCtx* ctx, DevBuf& b, const void* src, size_t bytes) {
  DISN_TRY(disn_ensure(ctx, b, bytes));
  DISN_HIP(ctx, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
  return DISN_OK;
}

extern "C" int disn_sdf_set_weights(DisnCtx* ctx, const DisnDecoderWeights* w) {
  if (!ctx) return DISN_E_ARG;
  if (!w) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_weights: null weights");
  const float* const* ptrs = (const float* const*)w;
  for (size_t i = 0; i < sizeof(DisnDecoderWeights) / sizeof(const float*); ++i)
    if (!ptrs[i]) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_weights: weight pointer %zu is null", i);
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  SdfState* s = sdf_state(ctx);
  s->have_weights = false;   // a failure below must not leave a half-packed set usable
  s->have_image = false;
  s->mixed_ok = s->mixed4_ok = true;
  struct Stream { const float *w1, *b1, *w2, *b2, *w3, *b3, *w4, *b4, *w5, *b5, *w6, *b6; };
  const Stream st[2] = {
      {w->g_f1c1_w, w->g_f1c1_b, w->g_f1c2_w, w->g_f1c2_b, w->g_f1c3_w, w->g_f1c3_b, w->g_f2c1_w, w->g_f2c1_b,
       w->g_f2c2_w, w->g_f2c2_b, w->g_f2c5_w, w->g_f2c5_b},
      {w->l_f1c1_w, w->l_f1c1_b, w->l_f1c2_w, w->l_f1c2_b, w->l_f1c3_w, w->l_f1c3_b, w->l_f2c1_w, w->l_f2c1_b,
       w->l_f2c2_w, w->l_f2c2_b, w->l_f2c5_w, w->l_f2c5_b}};
  for (int t = 0; t < 2; ++t) {
    const Stream& q = st[t];
    const float* arr[12] = {q.w1, q.b1, q.w2, q.b2, q.w3, q.b3, q.w4, q.b4, q.w5, q.b5, q.w6, q.b6};
    const size_t cnt[12] = {3 * 64, 64, 64 * 256, 256, 256 * 512, 512, (size_t)(t ? 1984 : 1536) * 512, 512, 512 * 256, 256, 256, 1};
    for (int i = 0; i < 12; ++i)
      if (!all_finite(arr[i], cnt[i]))
        return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_weights: %s stream, array %d holds a non-finite value", t ? "local" : "global", i);
  }
  for (int t = 0; t < 2; ++t) {
    const Stream& q = st[t];
    std::vector<uint16_t> frag;
    frag.reserve((size_t)PAIRS * 1024);
    for (int nb2 = 0; nb2 < 8; ++nb2)
      for (int ks = 0; ks < 4; ++ks) pack_pair(frag, q.w2, 256, nb2, ks);
    // fold1/conv3 runs one block ahead of fold2/conv1 (software pipeline in stream_pass)
    for (int ks = 0; ks < 16; ++ks) pack_pair(frag, q.w3, 512, 0, ks);
    for (int nb3 = 0; nb3 < 16; ++nb3) {
      if (nb3 < 15)
        for (int ks = 0; ks < 16; ++ks) pack_pair(frag, q.w3, 512, nb3 + 1, ks);
      for (int nb4 = 0; nb4 < 16; ++nb4)
        for (int sft = 0; sft < 2; ++sft) pack_pair(frag, q.w4, 512, nb4, 2 * nb3 + sft);  // rows 0..511 = point features
    }
    for (int nb4 = 0; nb4 < 16; ++nb4)
      for (int nb5 = 0; nb5 < 8; ++nb5)
        for (int sft = 0; sft < 2; ++sft) pack_pair(frag, q.w5, 256, nb5, 2 * nb4 + sft);
    if (frag.size() != (size_t)PAIRS * 1024) return disn_fail(ctx, DISN_E_STATE, "internal: fragment stream size");
    DISN_TRY(disn_ensure(ctx, s->ws, (size_t)2 * PAIRS * 2048));
    DISN_HIP(ctx, hipMemcpy((char*)s->ws.p + (size_t)t * PAIRS * 2048, frag.data(), frag.size() * sizeof(uint16_t),
                            hipMemcpyHostToDevice));
    // the same layers as units of the mixed-precision kernel: fold1/conv2, fold1/conv3, fold2/conv1 (point rows) as
    // f16 + fp8 groups, fold2/conv2 as the bf16 pairs packed above
    int sc_exp[4];
    {
      const float* lw[4] = {q.w2, q.w3, q.w4, q.w5};
      const int lk[4] = {64, 256, 512, 512}, ln[4] = {256, 512, 512, 256};
      for (int l = 0; l < 4; ++l) {
        float mx;
        layer_scale(lw[l], lk[l], ln[l], &sc_exp[l], &mx);
        if (!(mx < 65504.0f)) {   // beyond fp16: the f16 product cannot hold these weights
          s->mixed4_ok = false;
          if (l < 3) s->mixed_ok = false;
          s->mixed_bad_max = mx;
        }
      }
    }
    if (s->mixed_ok) {
      std::vector<uint8_t> mxs;
      mxs.reserve((size_t)PAIRS * 2048);
      for (int nb2 = 0; nb2 < 8; ++nb2) pack_group(mxs, q.w2, 256, nb2, 0, sc_exp[0]);
      for (int blk = 0; blk < 2; ++blk)
        for (int kg = 0; kg < 4; ++kg) pack_group(mxs, q.w3, 512, blk, kg, sc_exp[1]);
      for (int m = 0; m < 8; ++m) {
        if (m < 7)
          for (int blk = 2 * m + 2; blk < 2 * m + 4; ++blk)
            for (int kg = 0; kg < 4; ++kg) pack_group(mxs, q.w3, 512, blk, kg, sc_exp[1]);
        for (int nb4 = 0; nb4 < 16; ++nb4) pack_group(mxs, q.w4, 512, nb4, m, sc_exp[2]);
      }
      const size_t head = mxs.size();         // the three mixed layers; fold2/conv2 follows in either form
      const size_t f2c2_pairs = 16 * 16;      // the tail of the bf16 stream: fold2/conv2 as bf16 pairs
      const uint8_t* tail = (const uint8_t*)frag.data() + ((size_t)PAIRS - f2c2_pairs) * 2048;
      mxs.insert(mxs.end(), tail, tail + f2c2_pairs * 2048);
      if (mxs.size() != (size_t)PAIRS * 2048) return disn_fail(ctx, DISN_E_STATE, "internal: mixed fragment stream size");
      DISN_TRY(disn_ensure(ctx, s->wsmx, (size_t)2 * PAIRS * 2048));
      DISN_HIP(ctx, hipMemcpy((char*)s->wsmx.p + (size_t)t * PAIRS * 2048, mxs.data(), mxs.size(), hipMemcpyHostToDevice));
      if (s->mixed4_ok) {   // all-mixed decoder: fold2/conv2 as groups, k group m (64 rows) x output block nb5
        mxs.resize(head);
        for (int m = 0; m < 8; ++m)
          for (int nb5 = 0; nb5 < 8; ++nb5) pack_group(mxs, q.w5, 256, nb5, m, sc_exp[3]);
        if (mxs.size() != (size_t)PAIRS * 2048) return disn_fail(ctx, DISN_E_STATE, "internal: all-mixed fragment stream size");
        DISN_TRY(disn_ensure(ctx, s->wsmx4, (size_t)2 * PAIRS * 2048));
        DISN_HIP(ctx, hipMemcpy((char*)s->wsmx4.p + (size_t)t * PAIRS * 2048, mxs.data(), mxs.size(), hipMemcpyHostToDevice));
      }
    }
    std::vector<float> sp(SP_SIZE, 0.0f);
    for (int l = 0; l < 4; ++l) {
      const int e8m0 = 127 - sc_exp[l];   // the MFMA multiplies the product by 2^(byte - 127)
      memcpy(&sp[SP_SC + l], &e8m0, 4);
    }
    for (int f = 0; f < 64; ++f) {
      sp[SP_W1 + 4 * f + 0] = q.w1[0 * 64 + f];
      sp[SP_W1 + 4 * f + 1] = q.w1[1 * 64 + f];
      sp[SP_W1 + 4 * f + 2] = q.w1[2 * 64 + f];
      sp[SP_W1 + 4 * f + 3] = q.b1[f];
    }
    memcpy(&sp[SP_B2], q.b2, 256 * sizeof(float));
    memcpy(&sp[SP_B3], q.b3, 512 * sizeof(float));
    memcpy(&sp[SP_B4], q.b4, 512 * sizeof(float));
    memcpy(&sp[SP_B5], q.b5, 256 * sizeof(float));
    memcpy(&sp[SP_W6], q.w6, 256 * sizeof(float));
    sp[SP_B6] = q.b6[0];
    DISN_TRY(disn_ensure(ctx, s->sp, (size_t)2 * SP_SIZE * sizeof(float)));
    DISN_HIP(ctx, hipMemcpy((float*)s->sp.p + (size_t)t * SP_SIZE, sp.data(), sp.size() * sizeof(float),
                            hipMemcpyHostToDevice));
  }
  // per-image operands of the hoisted terms
  DISN_TRY(upload(ctx, s->wglob, w->g_f2c1_w + (size_t)512 * 512, (size_t)1024 * 512 * sizeof(float)));
  DISN_TRY(upload(ctx, s->wimg, w->l_f2c1_w + (size_t)512 * 512, (size_t)FEAT_C * 512 * sizeof(float)));
  DISN_TRY(upload(ctx, s->b4g, w->g_f2c1_b, 512 * sizeof(float)));
  DISN_TRY(upload(ctx, s->b4l, w->l_f2c1_b, 512 * sizeof(float)));   // folded into Pmap by disn_sdf_set_image
  {   // the weights as given, for the as-written fp32 reference of disn_sdf_selfcheck
    std::vector<float> raw;
    for (int t = 0; t < 2; ++t) {
      const Stream& q = st[t];
      const float* arr[12] = {q.w1, q.b1, q.w2, q.b2, q.w3, q.b3, q.w4, q.b4, q.w5, q.b5, q.w6, q.b6};
      const size_t cnt[12] = {3 * 64, 64, 64 * 256, 256, 256 * 512, 512, (size_t)(t ? 1984 : 1536) * 512, 512, 512 * 256, 256, 256, 1};
      for (int i = 0; i < 12; ++i) {
        raw.resize((raw.size() + 3) & ~(size_t)3);   // 16-byte aligned arrays (float4 loads)
        s->wraw_off[12 * t + i] = raw.size();
        raw.insert(raw.end(), arr[i], arr[i] + cnt[i]);
      }
    }
    DISN_TRY(upload(ctx, s->wraw, raw.data(), raw.size() * sizeof(float)));
  }
  s->have_weights = true;
  s->have_image = false;
  return DISN_OK;
}

extern "C" int disn_sdf_set_decoder(DisnCtx* ctx, int decoder) {
  if (!ctx) return DISN_E_ARG;
  if (decoder != DISN_DECODER_MIXED && decoder != DISN_DECODER_BF16X3 && decoder != DISN_DECODER_MIXED4)
    return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_decoder: unknown decoder %d", decoder);
  sdf_state(ctx)->decoder = decoder;
  return DISN_OK;
}

extern "C" int disn_resize_bilinear(DisnCtx* ctx, const float* in, int ih, int iw, int c, float* out, int oh, int ow,
                                    void* stream) {
  if (!ctx) return DISN_E_ARG;
  if (!in || !out || ih < 1 || iw < 1 || oh < 1 || ow < 1 || c < 4 || (c & 3))
    return disn_fail(ctx, DISN_E_ARG, "disn_resize_bilinear: bad argument (channels must be a multiple of 4)");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  long long n = (long long)oh * ow * (c / 4);
  k_resize<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(in, ih, iw, c, out, oh, ow, c, 0);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}

extern "C" int disn_sdf_set_image(DisnCtx* ctx, const float* const feat_maps[5], const float* global_feat,
                                  void* stream_) {
  if (!ctx) return DISN_E_ARG;
  SdfState* s = sdf_state(ctx);
  if (!s->have_weights) return disn_fail(ctx, DISN_E_STATE, "disn_sdf_set_image: call disn_sdf_set_weights first");
  if (!feat_maps || !global_feat) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_image: null argument");
  for (int l = 0; l < 5; ++l)
    if (!feat_maps[l]) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_set_image: feature map %d is null", l);
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  DISN_TRY(disn_ensure(ctx, s->f137, (size_t)NPIX * FEAT_C * sizeof(float)));
  if (s->pmap.cap < (size_t)PMAP_ROWS * 512 * sizeof(float)) {   // rows beyond the image: read with weight 0 by blend_mfma
    DISN_TRY(disn_ensure(ctx, s->pmap, (size_t)PMAP_ROWS * 512 * sizeof(float)));
    DISN_HIP(ctx, hipMemsetAsync((float*)s->pmap.p + (size_t)NPIX * 512, 0, (size_t)(PMAP_ROWS - NPIX) * 512 * sizeof(float), stream));
  }
  for (int l = 0; l < 5; ++l) {
    long long n = (long long)NPIX * (LV_C[l] / 4);
    k_resize<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(feat_maps[l], LV_HW[l], LV_HW[l], LV_C[l],
                                                              (float*)s->f137.p, IMG, IMG, FEAT_C, LV_OFF[l]);
  }
  dim3 grid(512 / GT, (NPIX + GT - 1) / GT);
  k_gemm_f32<<<grid, 256, 0, stream>>>((const float*)s->f137.p, FEAT_C, (const float*)s->wimg.p, (const float*)s->b4l.p,
                                       (float*)s->pmap.p, 512, NPIX, 512, FEAT_C, 0);
  k_global_bias<<<16, 256, 0, stream>>>(global_feat, (const float*)s->wglob.p, (const float*)s->b4g.p,
                                       (float*)s->sp.p + SP_B4);
  DISN_TRY(disn_ensure(ctx, s->gfeat, 1024 * sizeof(float)));   // kept for disn_sdf_selfcheck
  DISN_HIP(ctx, hipMemcpyAsync(s->gfeat.p, global_feat, 1024 * sizeof(float), hipMemcpyDeviceToDevice, stream));
  DISN_HIP(ctx, hipGetLastError());
  s->have_image = true;
  return DISN_OK;
}

static int launch_decode(DisnCtx* ctx, DecodeArgs& a, const float* trans_mat, hipStream_t stream) {
  SdfState* s = ctx->sdf;
  if (!s || !s->have_weights || !s->have_image)
    return disn_fail(ctx, DISN_E_STATE, "SDF evaluation needs disn_sdf_set_weights and disn_sdf_set_image first");
  if (!trans_mat) return disn_fail(ctx, DISN_E_ARG, "null trans_mat");
  for (int q = 0; q < 12; ++q) a.T[q] = trans_mat[q];
  a.ws = (const uint4*)s->ws.p;
  a.sp = (const float*)s->sp.p;
  a.pmap = (const float*)s->pmap.p;
  if (a.n <= 0) return DISN_OK;
  if (!a.grid_mode) a.ntiles = (a.n + 127) / 128;
  const long long tiles = a.ntiles;
  if (tiles * STAGES_PER_TILE > 0x7fffffffll) return disn_fail(ctx, DISN_E_ARG, "too many points for one launch");
  const unsigned blocks = (unsigned)(tiles < s->num_cu ? tiles : s->num_cu);
  if (s->decoder == DISN_DECODER_MIXED || s->decoder == DISN_DECODER_MIXED4) {
    const bool l4 = s->decoder == DISN_DECODER_MIXED4;
    if (!(l4 ? s->mixed4_ok : s->mixed_ok))
      return disn_fail(ctx, DISN_E_STATE, "the loaded weights reach |w| = %g, beyond the fp16 range of the mixed decoders: "
                       "select DISN_DECODER_BF16X3 (disn_sdf_set_decoder)", (double)s->mixed_bad_max);
    if (!s->lds_attr_set_mx) {
      const void* fns[4] = {(const void*)sdf_decode_mx<false, false>, (const void*)sdf_decode_mx<true, false>,
                            (const void*)sdf_decode_mx<false, true>, (const void*)sdf_decode_mx<true, true>};
      for (const void* f : fns) DISN_HIP(ctx, hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES));
      s->lds_attr_set_mx = true;
    }
    a.ws = (const uint4*)(l4 ? s->wsmx4.p : s->wsmx.p);
    if (l4) {
      sdf_decode_mx<false, true><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);   // out = g
      sdf_decode_mx<true, true><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);    // out = g + l (scaled)
    } else {
      sdf_decode_mx<false, false><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);
      sdf_decode_mx<true, false><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);
    }
    DISN_HIP(ctx, hipGetLastError());
    return DISN_OK;
  }
  if (!s->lds_attr_set) {   // once per context (the attribute is per device and function)
    DISN_HIP(ctx, hipFuncSetAttribute((const void*)sdf_decode, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    s->lds_attr_set = true;
  }
  sdf_decode<<<blocks, 256, LDS_BYTES, stream>>>(a);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}


// ---- accuracy self-check and decoder selection -------------------------------------------------------------------
// n sample points spread over the box (additive recurrence with the generalised golden ratio of dimension 3: low
// discrepancy, deterministic), evaluated (1) through the as-written fp32 graph -- projection, the five resized maps
// sampled and concatenated to 1472 channels, the tiled global feature, every 1x1 convolution as an exact-fp32
// MFMA GEMM with no algebraic shortcut (4.72 MFLOP per point) -- and (2) through the selected decoder kernel.
static int selfcheck_run(DisnCtx* ctx, SdfState* s, const float* trans_mat, const double* sdf_params, int n,
                         float* max_err, float* max_abs, hipStream_t stream) {
  const size_t N = (size_t)n;
  // work buffer layout (floats)
  size_t off = 0;
  auto take = [&](size_t cnt) { size_t o = off; off += (cnt + 63) & ~(size_t)63; return o; };
  const size_t o_pts = take(N * 3), o_xy = take(N * 2), o_h1g = take(N * 64), o_h1l = take(N * 64), o_h2 = take(N * 256);
  const size_t o_catg = take(N * 1536), o_catl = take(N * 1984), o_h4 = take(N * 512), o_h5g = take(N * 256), o_h5l = take(N * 256);
  const size_t o_ref = take(N), o_dec = take(N), o_res = take(64), o_tm = take(64);
  DISN_TRY(disn_ensure(ctx, s->chk, off * sizeof(float)));
  float* B = (float*)s->chk.p;
  std::vector<float> pts(N * 3);
  const double g = 1.22074408460575947536;   // x^4 = x + 1
  const double al[3] = {1.0 / g, 1.0 / (g * g), 1.0 / (g * g * g)};
  // First half: patches of 8 x 4 points 1/256 of the box apart around low-discrepancy anchors -- 32 consecutive points
  // are one wave of the decoder, whose pixel rectangle is then as small as in a dense grid, so the image term takes the
  // matrix-product path (blend_mfma); second half: scattered points (the gather path).
  const size_t npatch = N / 64;
  for (size_t i = 0; i < N; ++i) {
    const bool patch = i < 32 * npatch;
    const size_t a = patch ? i / 32 : i;
    const int off[3] = {(int)(i & 7), (int)((i >> 3) & 3), 0};
    for (int c = 0; c < 3; ++c) {
      double u = 0.5 + al[c] * (double)(a + 1);
      u -= floor(u);
      if (patch) u = u * (248.0 / 256.0) + off[c] / 256.0;
      pts[3 * i + c] = (float)(sdf_params[c] + u * (sdf_params[3 + c] - sdf_params[c]));
    }
  }
  DISN_HIP(ctx, hipMemcpyAsync(B + o_pts, pts.data(), N * 3 * sizeof(float), hipMemcpyHostToDevice, stream));
  DISN_HIP(ctx, hipMemcpyAsync(B + o_tm, trans_mat, 12 * sizeof(float), hipMemcpyHostToDevice, stream));
  DISN_HIP(ctx, hipStreamSynchronize(stream));   // pts is a local
  const float* W = (const float*)s->wraw.p;
  auto wp = [&](int t, int i) { return W + s->wraw_off[12 * t + i]; };   // i: w1 b1 w2 b2 w3 b3 w4 b4 w5 b5 w6 b6
  const unsigned nb = (unsigned)((N + 255) / 256);
  k_project<<<nb, 256, 0, stream>>>(B + o_pts, (long long)N, B + o_tm, B + o_xy);
  const long long tq = (long long)N * (FEAT_C / 4);
  k_sample_features<<<(unsigned)((tq + 255) / 256), 256, 0, stream>>>((const float*)s->f137.p, B + o_xy, (long long)N, B + o_catl, 1984, 512);
  k_ref_tile_global<<<(unsigned)((N * 256 + 255) / 256), 256, 0, stream>>>((const float*)s->gfeat.p, n, B + o_catg, 1536, 512);
  k_ref_l1<<<(unsigned)((N * 64 + 255) / 256), 256, 0, stream>>>(B + o_pts, n, wp(0, 0), wp(0, 1), wp(1, 0), wp(1, 1), B + o_h1g, B + o_h1l);
  auto gemm = [&](const float* A, int lda, const float* Wm, const float* bias, float* C, int ldc, int Nn, int K) {
    dim3 grid(Nn / GT, (unsigned)((N + GT - 1) / GT));
    k_gemm_f32<<<grid, 256, 0, stream>>>(A, lda, Wm, bias, C, ldc, n, Nn, K, 1);
  };
  for (int t = 0; t < 2; ++t) {
    float* cat = B + (t ? o_catl : o_catg);
    const int ck = t ? 1984 : 1536;
    gemm(B + (t ? o_h1l : o_h1g), 64, wp(t, 2), wp(t, 3), B + o_h2, 256, 256, 64);     // fold1/conv2
    gemm(B + o_h2, 256, wp(t, 4), wp(t, 5), cat, ck, 512, 256);                          // fold1/conv3 -> concat[:, 0:512]
    gemm(cat, ck, wp(t, 6), wp(t, 7), B + o_h4, 512, 512, ck);                            // fold2/conv1 on the concatenation
    gemm(B + o_h4, 512, wp(t, 8), wp(t, 9), B + (t ? o_h5l : o_h5g), 256, 256, 512);     // fold2/conv2
  }
  k_ref_final<<<nb, 256, 0, stream>>>(B + o_h5g, B + o_h5l, n, wp(0, 10), wp(0, 11), wp(1, 10), wp(1, 11), B + o_ref);
  DISN_HIP(ctx, hipGetLastError());
  // the selected decoder on the same points (points mode, scaled like the grid)
  DecodeArgs a{};
  a.grid_mode = 0;
  a.pts = B + o_pts;
  a.pts_rot = B + o_pts;
  a.n = n;
  a.scale_out = 1;
  a.out = B + o_dec;
  DISN_TRY(launch_decode(ctx, a, trans_mat, stream));
  DISN_HIP(ctx, hipMemsetAsync(B + o_res, 0, 2 * sizeof(unsigned), stream));
  k_max_abs_diff<<<nb, 256, 0, stream>>>(B + o_ref, B + o_dec, n, (unsigned*)(B + o_res));
  DISN_HIP(ctx, hipGetLastError());
  float res[2];
  DISN_HIP(ctx, hipMemcpyAsync(res, B + o_res, sizeof res, hipMemcpyDeviceToHost, stream));
  DISN_HIP(ctx, hipStreamSynchronize(stream));
  *max_err = res[0];
  if (max_abs) *max_abs = res[1];
  return DISN_OK;
}

static int selfcheck_args(DisnCtx* ctx, const char* who, const float* trans_mat, const double* sdf_params, int n) {
  SdfState* s = ctx->sdf;
  if (!s || !s->have_weights || !s->have_image)
    return disn_fail(ctx, DISN_E_STATE, "%s needs disn_sdf_set_weights and disn_sdf_set_image first", who);
  if (!trans_mat || !sdf_params) return disn_fail(ctx, DISN_E_ARG, "%s: null argument", who);
  if (n < 1 || n > (1 << 16)) return disn_fail(ctx, DISN_E_ARG, "%s: n_points must be in [1, 65536]", who);
  return DISN_OK;
}

extern "C" int disn_sdf_selfcheck(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int n_points,
                                  float* max_abs_err, float* max_abs_sdf, void* stream) {
  if (!ctx) return DISN_E_ARG;
  if (!max_abs_err) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_selfcheck: null output");
  DISN_TRY(selfcheck_args(ctx, "disn_sdf_selfcheck", trans_mat, sdf_params, n_points));
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  return selfcheck_run(ctx, ctx->sdf, trans_mat, sdf_params, n_points, max_abs_err, max_abs_sdf, (hipStream_t)stream);
}

extern "C" int disn_sdf_autotune(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int n_points, float tolerance,
                                 float margin, int* decoder_out, float errs[3], void* stream) {
  if (!ctx) return DISN_E_ARG;
  DISN_TRY(selfcheck_args(ctx, "disn_sdf_autotune", trans_mat, sdf_params, n_points));
  if (!(tolerance > 0.0f) || !(margin >= 1.0f)) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_autotune: tolerance must be > 0 and margin >= 1");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  SdfState* s = ctx->sdf;
  const int ladder[3] = {DISN_DECODER_MIXED4, DISN_DECODER_MIXED, DISN_DECODER_BF16X3};   // cheapest first
  const bool usable[3] = {s->mixed4_ok, s->mixed_ok, true};
  const int before = s->decoder;
  int chosen = -1;
  float e[3] = {-1.0f, -1.0f, -1.0f};   // -1: not measured
  for (int i = 0; i < 3 && chosen < 0; ++i) {
    if (!usable[i]) continue;
    s->decoder = ladder[i];
    int rc = selfcheck_run(ctx, s, trans_mat, sdf_params, n_points, &e[i], nullptr, (hipStream_t)stream);
    if (rc != DISN_OK) { s->decoder = before; return rc; }
    if (e[i] * margin <= tolerance) chosen = ladder[i];
  }
  if (errs) { errs[DISN_DECODER_MIXED4] = e[0]; errs[DISN_DECODER_MIXED] = e[1]; errs[DISN_DECODER_BF16X3] = e[2]; }
  if (chosen < 0) {
    s->decoder = DISN_DECODER_BF16X3;
    if (decoder_out) *decoder_out = DISN_DECODER_BF16X3;
    return disn_fail(ctx, DISN_E_STATE, "disn_sdf_autotune: no decoder arithmetic keeps %gx margin to the %g tolerance on these "
                     "weights and this image (measured max |error|: all-mixed %g, mixed %g, bf16x3 %g); bf16x3 stays selected",
                     (double)margin, (double)tolerance, (double)e[0], (double)e[1], (double)e[2]);
  }
  s->decoder = chosen;
  if (decoder_out) *decoder_out = chosen;
  return DISN_OK;
}

extern "C" int disn_sdf_grid(DisnCtx* ctx, const float* trans_mat, const double* sdf_params, int res, int z0, int z1,
                             float* sdf_out, void* stream) {
  if (!ctx) return DISN_E_ARG;
  if (!sdf_params || !sdf_out) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_grid: null argument");
  if (res < 1 || res > 2047 || z0 < 0 || z1 > res || z0 > z1)
    return disn_fail(ctx, DISN_E_ARG, "disn_sdf_grid: bad res/z range (res=%d z0=%d z1=%d)", res, z0, z1);
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  DecodeArgs a{};
  a.grid_mode = 1;
  a.res = res;
  a.z0 = z0;
  for (int q = 0; q < 3; ++q) {
    a.lo[q] = sdf_params[q];
    a.hi[q] = sdf_params[3 + q];
    a.step[q] = (sdf_params[3 + q] - sdf_params[q]) / (double)res;  // np.linspace: delta / div
  }
  a.n = (long long)(z1 - z0 + 1) * (res + 1) * (res + 1);
  // tiles: bricks of 8 x 4 x 4 points over the core of the slab, the rest wave by wave (grid_point)
  a.n1 = res + 1;
  a.nk = z1 - z0 + 1;
  a.ic = a.n1 & ~7; a.jc = a.n1 & ~3; a.kc = a.nk & ~3;
  const int nbi = a.ic / 8, nbj = a.jc / 4, nbk = a.kc / 4;
  auto pow2_div = [](int n, int cap) { int sh = 0; while (sh < cap && n > 0 && n % (2 << sh) == 0) ++sh; return sh; };
  a.sbi = pow2_div(nbi, 2); a.sbj = pow2_div(nbj, 3); a.sbk = pow2_div(nbk, 3);   // super-brick <= 32^3 points
  a.nsi = nbi >> a.sbi; a.nsj = nbj >> a.sbj;
  a.core_tiles = (long long)nbi * nbj * nbk;
  // waves of the remainder regions (grid_point): patches of the planes k >= kc, runs of 32 along the rows j >= jc, runs
  // of 32 along the columns i >= ic (j < jc)
  const long long r1 = (long long)(a.nk - a.kc) * nbi * nbj, r2 = (long long)a.nk * (a.n1 - a.jc) * ((a.n1 + 31) / 32);
  const long long r3 = (long long)a.nk * (a.n1 - a.ic) * ((a.jc + 31) / 32);
  if (a.core_tiles == 0) { a.nsi = a.nsj = 1; }
  a.r1 = (unsigned)r1; a.r2 = (unsigned)r2; a.r3 = (unsigned)r3;
  a.ntiles = a.core_tiles + (r1 + r2 + r3 + 3) / 4;
  a.scale_out = 1;
  a.out = sdf_out;
  return launch_decode(ctx, a, trans_mat, (hipStream_t)stream);
}

extern "C" int disn_sdf_points(DisnCtx* ctx, const float* trans_mat, const float* pts, const float* pts_rot, int64_t n,
                               float* out, void* stream) {
  if (!ctx) return DISN_E_ARG;
  if (n < 0 || (n && (!pts || !pts_rot || !out))) return disn_fail(ctx, DISN_E_ARG, "disn_sdf_points: bad argument");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  DecodeArgs a{};
  a.grid_mode = 0;
  a.pts = pts;
  a.pts_rot = pts_rot;
  a.n = n;
  a.scale_out = 0;
  a.out = out;
  return launch_decode(ctx, a, trans_mat, (hipStream_t)stream);
}

extern "C" int disn_project_points(DisnCtx* ctx, const float* trans_mat, const float* pts, int64_t n, float* xy_out,
                                   void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (!trans_mat || n < 0 || (n && (!pts || !xy_out))) return disn_fail(ctx, DISN_E_ARG, "disn_project_points: bad argument");
  if (n == 0) return DISN_OK;
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  SdfState* s = sdf_state(ctx);
  DISN_TRY(disn_ensure(ctx, s->tmat, 12 * sizeof(float)));
  DISN_HIP(ctx, hipMemcpyAsync(s->tmat.p, trans_mat, 12 * sizeof(float), hipMemcpyHostToDevice, stream));
  k_project<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(pts, n, (const float*)s->tmat.p, xy_out);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}

extern "C" int disn_sample_features(DisnCtx* ctx, const float* xy, int64_t n, float* feat_out, void* stream) {
  if (!ctx) return DISN_E_ARG;
  SdfState* s = ctx->sdf;
  if (!s || !s->have_image) return disn_fail(ctx, DISN_E_STATE, "disn_sample_features: call disn_sdf_set_image first");
  if (n < 0 || (n && (!xy || !feat_out))) return disn_fail(ctx, DISN_E_ARG, "disn_sample_features: bad argument");
  if (n == 0) return DISN_OK;
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  long long t = (long long)n * (FEAT_C / 4);
  k_sample_features<<<(unsigned)((t + 255) / 256), 256, 0, (hipStream_t)stream>>>((const float*)s->f137.p, xy, n, feat_out, FEAT_C, 0);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}
