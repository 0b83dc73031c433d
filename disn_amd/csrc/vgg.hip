// vgg.hip -- the once-per-image VGG-16 encoder of DISN on gfx950, in exact fp32.
//
// Replaces `vgg.vgg_16(ref_img, num_classes=1024, is_training=False, spatial_squeeze=False)` and the 137 -> 224
// resize in front of it (reference models/model_normalization.py:65-78; layer list: the slim mirror
// models/CNN/vgg.py:182-218; the camera estimator runs a second copy, cam_est/model_cam_old.py:70-78).
// It is NOT the hot path (once per image), but its end points feed every SDF value, so it is deterministic and
// plain fp32 by construction: every convolution is an implicit GEMM over NHWC activations
// (M = pixels, N = output channels, K = 9 taps x input channels, the TF variable layout HWIO is the B matrix as it
// lies) on v_mfma_f32_32x32x2_f32 with a fixed summation order -- no algorithm search, no Winograd transform, no
// atomics.  Layers with few pixels split K over workgroups into partial sums that a second kernel adds in a
// fixed order.
#include <string.h>

#include <vector>

#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(4))) float floatx4;

constexpr int GT = 128, GK = 16;

// TF1 legacy bilinear resize (align_corners=False, no half-pixel centres) of an HWC image with c <= 4 channels
// into a 4-channel map (channels c.. are zero).  Same arithmetic as sdf.hip's k_resize.
__global__ void k_resize_pad4(const float* __restrict__ in, int ih, int iw, int c, float* __restrict__ out, int oh, int ow) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= oh * ow) return;
  int ox = t % ow, oy = t / ow;
  const float sy = __fdiv_rn((float)ih, (float)oh), sx = __fdiv_rn((float)iw, (float)ow);
  float fy = __fmul_rn((float)oy, sy), fx = __fmul_rn((float)ox, sx);
  int ylo = (int)floorf(fy), xlo = (int)floorf(fx);
  int yhi = min(ylo + 1, ih - 1), xhi = min(xlo + 1, iw - 1);
  float ly = __fsub_rn(fy, (float)ylo), lx = __fsub_rn(fx, (float)xlo);
  floatx4 o = {0, 0, 0, 0};
  for (int q = 0; q < c; ++q) {
    float tl = in[((size_t)ylo * iw + xlo) * c + q], tr = in[((size_t)ylo * iw + xhi) * c + q];
    float bl = in[((size_t)yhi * iw + xlo) * c + q], br = in[((size_t)yhi * iw + xhi) * c + q];
    float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), lx));
    float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), lx));
    o[q] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ly));
  }
  *(floatx4*)(out + (size_t)t * 4) = o;
}

// 3x3 SAME convolution, stride 1, NHWC, as an implicit GEMM:  C[m][n] = sum_k A[m][k] B[k][n],
// m = y*W + x, k = tap*cin + ci (tap = 3*(dy+1) + (dx+1)), A[m][k] = in[y+dy][x+dx][ci] (0 outside the image),
// B = the HWIO weights [9*cin (zero rows up to kpad)][n].  cin is a power of two >= 4 (1 << cin_shift).
// 128 x TN tile per workgroup (TN = 128; 64 for the two 64-channel layers, which would waste half of a 128-wide tile),
// 4 waves of 64 x TN/2 (2 x TN/64 MFMA tiles of 32 x 32), 16 k per LDS stage.
// blockIdx.z = K split: split s takes stages [s*per, (s+1)*per) and writes its partial sums to
// part[s][M][N] (nsplit > 1) or the finished relu(acc + bias) to out (nsplit == 1).
template <int TN>
__global__ __launch_bounds__(256) void k_conv3x3(const float* __restrict__ in, const float* __restrict__ Wt,
                                                 const float* __restrict__ bias, float* __restrict__ out, int H, int W,
                                                 int cin_shift, int N, int kpad, int per, int nsplit) {
  __shared__ float As[GK][GT + 4];
  __shared__ float Bs[GK][TN + 4];
  constexpr int NB = TN / 64;   // MFMA tiles of a wave along N
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = H * W;
  const int m0 = blockIdx.y * GT, n0 = blockIdx.x * TN;
  const int wm = (w >> 1) * 64, wn = (w & 1) * (TN / 2);
  const int cin = 1 << cin_shift;
  // the two A rows this thread loads in every stage
  int py[2], px[2];
  bool pv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int m = m0 + ((threadIdx.x + q * 256) >> 2);
    pv[q] = m < M;
    py[q] = m / W;
    px[q] = m - py[q] * W;
  }
  const int kq = (threadIdx.x & 3) * 4;
  floatx16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][b][g] = 0.0f;
  const int s0 = blockIdx.z * per, s1 = min(kpad / GK, s0 + per);
  for (int st = s0; st < s1; ++st) {
    const int k0 = st * GK;
    {
      const int k = k0 + kq, tap = k >> cin_shift, ci = k & (cin - 1);
      const int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = (threadIdx.x + q * 256) >> 2;
        const int yy = py[q] + dy, xx = px[q] + dx;
        floatx4 v = {0, 0, 0, 0};
        if (pv[q] && tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W)
          v = *(const floatx4*)(in + (((size_t)yy * W + xx) << cin_shift) + ci);
        As[kq + 0][row] = v[0]; As[kq + 1][row] = v[1]; As[kq + 2][row] = v[2]; As[kq + 3][row] = v[3];
      }
    }
#pragma unroll
    for (int q = 0; q < TN / 64; ++q) {
      int idx = threadIdx.x + q * 256;
      int kr = idx / (TN / 4), cq = (idx % (TN / 4)) * 4;
      floatx4 v = {0, 0, 0, 0};
      if (n0 + cq < N) v = *(const floatx4*)(Wt + (size_t)(k0 + kr) * N + n0 + cq);
      *(floatx4*)&Bs[kr][cq] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GK; kk += 2) {
      const float a0 = As[kk + h][wm + r], a1 = As[kk + h][wm + 32 + r];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float bv = Bs[kk + h][wn + 32 * b + r];
        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0][b], 0, 0, 0);
        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1][b], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* dst = nsplit > 1 ? out + (size_t)blockIdx.z * M * N : out;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        int row = m0 + wm + a * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
        int col = n0 + wn + b * 32 + r;
        if (row < M && col < N) {
          float v = acc[a][b][g];
          if (nsplit == 1) v = fmaxf(__fadd_rn(v, bias[col]), 0.0f);
          dst[(size_t)row * N + col] = v;
        }
      }
}

// out[i] = act(part[0][i] + part[1][i] + ... (in this order) + bias[i % N]); 4 values per thread
__global__ void k_sum_splits(const float* __restrict__ part, int nsplit, size_t total, int N, const float* __restrict__ bias,
                             int relu, float* __restrict__ out) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= total) return;
  floatx4 s = *(const floatx4*)(part + i);
  for (int p = 1; p < nsplit; ++p) {
    floatx4 v = *(const floatx4*)(part + (size_t)p * total + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], v[e]);
  }
  floatx4 b = *(const floatx4*)(bias + (i % N));
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    s[e] = __fadd_rn(s[e], b[e]);
    if (relu) s[e] = fmaxf(s[e], 0.0f);
  }
  *(floatx4*)(out + i) = s;
}

// 2x2 max pooling, stride 2, VALID, NHWC; 4 channels per thread
__global__ void k_maxpool2(const float* __restrict__ in, int H, int W, int C, float* __restrict__ out) {
  const int oh = H / 2, ow = W / 2, c4 = C / 4;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)oh * ow * c4) return;
  int ch = (int)(t % c4) * 4;
  int ox = (int)((t / c4) % ow), oy = (int)(t / ((size_t)c4 * ow));
  const float* p = in + ((size_t)(2 * oy) * W + 2 * ox) * C + ch;
  floatx4 a = *(const floatx4*)p, b = *(const floatx4*)(p + C);
  floatx4 c = *(const floatx4*)(p + (size_t)W * C), d = *(const floatx4*)(p + (size_t)W * C + C);
  floatx4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
  *(floatx4*)(out + ((size_t)oy * ow + ox) * C + ch) = o;
}

// fully connected layer on one vector (fc6 = 7x7 VALID conv on the 7x7x512 map, fc7, fc8 = 1x1 convs on 1x1):
// part[s][n] = sum over rows [s*rows, (s+1)*rows) of x[k] W[k][n], k ascending, one fmaf per term.
// A thread owns 4 consecutive columns; a block 1024 columns of one row range.
__global__ __launch_bounds__(256) void k_fc_partial(const float* __restrict__ x, const float* __restrict__ Wt, int K, int N,
                                                    int rows, float* __restrict__ part) {
  const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int k0 = blockIdx.y * rows, k1 = min(K, k0 + rows);
  if (n >= N) return;
  floatx4 s = {0, 0, 0, 0};
  for (int k = k0; k < k1; ++k) {
    const float xv = x[k];
    const floatx4 wv = *(const floatx4*)(Wt + (size_t)k * N + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[e] = fmaf(xv, wv[e], s[e]);
  }
  *(floatx4*)(part + (size_t)blockIdx.y * N + n) = s;
}

// The same sum for the few outputs of a fully connected layer (1024 .. 4096) over many partials (128 .. 512): a
// workgroup takes 128 outputs; its 8 groups of 32 threads each add one eighth of the partials in order, and the eight
// group sums are added in group order -- a fixed association, 8 x fewer dependent loads per thread than one serial loop
// (which took 50 us for fc6's 196 partials on four workgroups).
__global__ __launch_bounds__(256) void k_sum_splits_fc(const float* __restrict__ part, int nsplit, int N, const float* __restrict__ bias,
                                                       int relu, float* __restrict__ out) {
  __shared__ floatx4 lds[8][32];
  const int q = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int n = (blockIdx.x * 32 + q) * 4;
  const int per = (nsplit + 7) / 8, p0 = g * per, p1 = min(nsplit, p0 + per);
  floatx4 s = {0, 0, 0, 0};
  if (n < N)
    for (int p = p0; p < p1; ++p) {
      const floatx4 v = *(const floatx4*)(part + (size_t)p * N + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = __fadd_rn(s[e], v[e]);
    }
  lds[g][q] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    floatx4 t = lds[0][q];
    for (int gg = 1; gg < 8; ++gg)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = __fadd_rn(t[e], lds[gg][q][e]);
    const floatx4 b = *(const floatx4*)(bias + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      t[e] = __fadd_rn(t[e], b[e]);
      if (relu) t[e] = fmaxf(t[e], 0.0f);
    }
    *(floatx4*)(out + n) = t;
  }
}

struct ConvLayer {
  int H, W, cin_shift, cout, kpad, nsplit, per, tn;
  size_t w_off, b_off;   // floats into the weight buffer
  int feat;              // >= 0: this layer's output is end point `feat`
  bool pool_after;
};

}  // namespace

struct VggNet {
  DevBuf weights, act[2], part, img224, fcx[2];
  std::vector<ConvLayer> conv;
  size_t fc_w_off[3], fc_b_off[3];
  bool loaded = false;
};
struct VggState {
  VggNet net[2];
  DevBuf feats[5];   // end points when the caller does not want them
};

void disn_vgg_state_free(DisnCtx* ctx) {
  if (!ctx->vgg) return;
  for (VggNet& n : ctx->vgg->net) {
    DevBuf* all[] = {&n.weights, &n.act[0], &n.act[1], &n.part, &n.img224, &n.fcx[0], &n.fcx[1]};
    for (DevBuf* b : all) disn_release(*b);
  }
  for (DevBuf& b : ctx->vgg->feats) disn_release(b);
  delete ctx->vgg;
  ctx->vgg = nullptr;
}

static const int VGG_BLOCKS[5][2] = {{2, 64}, {2, 128}, {3, 256}, {3, 512}, {3, 512}};
static const int FC_K[3] = {7 * 7 * 512, 4096, 4096};
static const int FC_N_DEFAULT[3] = {4096, 4096, 1024};
static const int FC_ROWS[3] = {128, 32, 16};   // rows of W per partial sum (disn_vgg_forward)

extern "C" int disn_vgg_set_weights(DisnCtx* ctx, int slot, const DisnVggWeights* w) {
  if (!ctx) return DISN_E_ARG;
  if (slot < 0 || slot > 1) return disn_fail(ctx, DISN_E_ARG, "disn_vgg_set_weights: slot must be 0 (SDF encoder) or 1 (camera encoder)");
  if (!w) return disn_fail(ctx, DISN_E_ARG, "disn_vgg_set_weights: null weights");
  for (int i = 0; i < 13; ++i)
    if (!w->conv_w[i] || !w->conv_b[i]) return disn_fail(ctx, DISN_E_ARG, "disn_vgg_set_weights: conv layer %d is null", i);
  for (int i = 0; i < 3; ++i)
    if (!w->fc_w[i] || !w->fc_b[i]) return disn_fail(ctx, DISN_E_ARG, "disn_vgg_set_weights: fc%d is null", 6 + i);
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  if (!ctx->vgg) ctx->vgg = new VggState();
  VggNet& net = ctx->vgg->net[slot];
  net.loaded = false;
  net.conv.clear();
  int cu = 256;
  (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  // layer table + packed host image of all parameters
  std::vector<float> host;
  int hw = 224, cin = 3, li = 0;
  size_t part_floats = 0;
  for (int b = 0; b < 5; ++b) {
    for (int l = 0; l < VGG_BLOCKS[b][0]; ++l, ++li) {
      ConvLayer c{};
      const int cout = VGG_BLOCKS[b][1];
      int cinp = cin < 4 ? 4 : cin;
      c.H = c.W = hw;
      c.cin_shift = 0;
      while ((1 << c.cin_shift) < cinp) ++c.cin_shift;
      c.cout = cout;
      c.kpad = (9 * cinp + GK - 1) / GK * GK;
      c.feat = (l == VGG_BLOCKS[b][0] - 1) ? b : -1;
      c.pool_after = c.feat >= 0;
      // K split so that small maps still fill the chip; every split gets whole 16-k stages
      c.tn = cout <= 64 ? 64 : GT;
      const int tiles = ((hw * hw + GT - 1) / GT) * ((cout + c.tn - 1) / c.tn), stages = c.kpad / GK;
      // (a few workgroups per CU: the kernel has no double buffering, co-resident workgroups hide its loads; but a split
      // pays for a 64 KB tile of partial sums, so it gets at least VGG_MIN_STAGES stages)
      static const int wg_per_cu = getenv("DISN_VGG_WG_PER_CU") ? atoi(getenv("DISN_VGG_WG_PER_CU")) : 2;
      static const int min_stages = getenv("DISN_VGG_MIN_STAGES") ? atoi(getenv("DISN_VGG_MIN_STAGES")) : 4;
      int want = tiles >= wg_per_cu * cu ? 1 : (wg_per_cu * cu + tiles - 1) / tiles;
      if (want > stages / min_stages) want = stages / min_stages;
      if (want < 1) want = 1;
      c.per = (stages + want - 1) / want;
      c.nsplit = (stages + c.per - 1) / c.per;
      if (c.nsplit > 1) part_floats = std::max(part_floats, (size_t)c.nsplit * hw * hw * cout);
      c.w_off = host.size();
      host.resize(host.size() + (size_t)c.kpad * cout, 0.0f);
      const float* src = w->conv_w[li];   // HWIO [3][3][cin][cout]
      for (int tap = 0; tap < 9; ++tap)
        for (int ci = 0; ci < cin; ++ci)
          memcpy(&host[c.w_off + ((size_t)tap * cinp + ci) * cout], src + ((size_t)tap * cin + ci) * cout, cout * sizeof(float));
      c.b_off = host.size();
      host.insert(host.end(), w->conv_b[li], w->conv_b[li] + cout);
      net.conv.push_back(c);
      cin = cout;
    }
    hw /= 2;
  }
  const int fc_n[3] = {FC_N_DEFAULT[0], FC_N_DEFAULT[1], FC_N_DEFAULT[2]};
  for (int i = 0; i < 3; ++i) {
    net.fc_w_off[i] = host.size();
    host.insert(host.end(), w->fc_w[i], w->fc_w[i] + (size_t)FC_K[i] * fc_n[i]);
    net.fc_b_off[i] = host.size();
    host.insert(host.end(), w->fc_b[i], w->fc_b[i] + fc_n[i]);
  }
  for (size_t i = 0; i < host.size(); ++i)
    if (!(host[i] - host[i] == 0.0f)) return disn_fail(ctx, DISN_E_ARG, "disn_vgg_set_weights: non-finite parameter");
  DISN_TRY(disn_ensure(ctx, net.weights, host.size() * sizeof(float)));
  DISN_HIP(ctx, hipMemcpy(net.weights.p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  const size_t act_floats = (size_t)224 * 224 * 64;
  DISN_TRY(disn_ensure(ctx, net.act[0], act_floats * sizeof(float)));
  DISN_TRY(disn_ensure(ctx, net.act[1], act_floats * sizeof(float)));
  // fc partial sums share the buffer: fc6 splits 25088 rows into 196 ranges of 128, fc7 4096 into 128, fc8 4096 into 512
  for (int i = 0; i < 3; ++i) part_floats = std::max(part_floats, (size_t)((FC_K[i] + FC_ROWS[i] - 1) / FC_ROWS[i]) * fc_n[i]);
  DISN_TRY(disn_ensure(ctx, net.part, part_floats * sizeof(float)));
  DISN_TRY(disn_ensure(ctx, net.img224, (size_t)224 * 224 * 4 * sizeof(float)));
  DISN_TRY(disn_ensure(ctx, net.fcx[0], 4096 * sizeof(float)));
  DISN_TRY(disn_ensure(ctx, net.fcx[1], 4096 * sizeof(float)));
  static const int FEAT_HW[5] = {224, 112, 56, 28, 14};
  for (int b = 0; b < 5; ++b)
    DISN_TRY(disn_ensure(ctx, ctx->vgg->feats[b], (size_t)FEAT_HW[b] * FEAT_HW[b] * VGG_BLOCKS[b][1] * sizeof(float)));
  net.loaded = true;
  return DISN_OK;
}

extern "C" int disn_vgg_forward(DisnCtx* ctx, int slot, const float* img, int img_h, int img_w, float* const feat_maps[5],
                                float* global_feat, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (slot < 0 || slot > 1 || !ctx->vgg || !ctx->vgg->net[slot].loaded)
    return disn_fail(ctx, DISN_E_STATE, "disn_vgg_forward: call disn_vgg_set_weights(slot %d) first", slot);
  if (!img || img_h < 1 || img_w < 1 || img_h > 4096 || img_w > 4096)
    return disn_fail(ctx, DISN_E_ARG, "disn_vgg_forward: bad image argument");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  VggNet& net = ctx->vgg->net[slot];
  const float* Wd = (const float*)net.weights.p;
  // reference models/model_normalization.py:65-72: legacy bilinear resize to 224 x 224 unless the image has that size
  k_resize_pad4<<<(224 * 224 + 255) / 256, 256, 0, stream>>>(img, img_h, img_w, 3, (float*)net.img224.p, 224, 224);
  const float* cur = (const float*)net.img224.p;
  int pp = 0;
  for (const ConvLayer& c : net.conv) {
    const int M = c.H * c.W;
    float* dst = c.feat >= 0 ? ((feat_maps && feat_maps[c.feat]) ? feat_maps[c.feat] : (float*)ctx->vgg->feats[c.feat].p)
                             : (float*)net.act[pp].p;
    dim3 grid((c.cout + c.tn - 1) / c.tn, (M + GT - 1) / GT, c.nsplit);
    if (c.tn == 64)
      k_conv3x3<64><<<grid, 256, 0, stream>>>(cur, Wd + c.w_off, Wd + c.b_off, c.nsplit > 1 ? (float*)net.part.p : dst, c.H, c.W,
                                              c.cin_shift, c.cout, c.kpad, c.per, c.nsplit);
    else
      k_conv3x3<GT><<<grid, 256, 0, stream>>>(cur, Wd + c.w_off, Wd + c.b_off, c.nsplit > 1 ? (float*)net.part.p : dst, c.H, c.W,
                                              c.cin_shift, c.cout, c.kpad, c.per, c.nsplit);
    if (c.nsplit > 1) {
      const size_t total = (size_t)M * c.cout;
      k_sum_splits<<<(unsigned)((total / 4 + 255) / 256), 256, 0, stream>>>((const float*)net.part.p, c.nsplit, total, c.cout,
                                                                            Wd + c.b_off, 1, dst);
    }
    cur = dst;
    if (c.feat < 0) pp ^= 1;
    if (c.pool_after) {
      float* pd = (float*)net.act[pp].p;
      const size_t n = (size_t)(c.H / 2) * (c.W / 2) * (c.cout / 4);
      k_maxpool2<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(cur, c.H, c.W, c.cout, pd);
      cur = pd;
      pp ^= 1;
    }
  }
  // cur = pool5 [7][7][512]; fc6 (7x7 VALID) + ReLU, fc7 + ReLU (dropout is the identity at inference), fc8 linear
  const int fc_n[3] = {FC_N_DEFAULT[0], FC_N_DEFAULT[1], FC_N_DEFAULT[2]};
  for (int i = 0; i < 3; ++i) {
    // row ranges sized so that every layer puts >= 512 workgroups on the chip (fc6: 196 ranges of 128 rows x 4 column
    // blocks; fc7: 128 x 4; fc8: 512 x 1)
    const int K = FC_K[i], N = fc_n[i], rows = FC_ROWS[i], ns = (K + rows - 1) / rows;
    dim3 grid((N / 4 + 255) / 256, ns);
    k_fc_partial<<<grid, 256, 0, stream>>>(cur, Wd + net.fc_w_off[i], K, N, rows, (float*)net.part.p);
    float* dst = i == 2 ? global_feat : (float*)net.fcx[i].p;
    if (!dst) dst = (float*)net.fcx[0].p;   // i == 2 without an output pointer: end points only
    k_sum_splits_fc<<<(N + 127) / 128, 256, 0, stream>>>((const float*)net.part.p, ns, N, Wd + net.fc_b_off[i], i < 2 ? 1 : 0, dst);
    cur = dst;
  }
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}
