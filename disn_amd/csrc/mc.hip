// mc.hip -- GPU marching cubes for gfx950, replacing the reference's
// `computeMarchingCubes <dist> <obj> -i iso` step (reference:
// demo/demo.py:355-369; algorithm: SURVEY.md Appendix B).
//
// HBM-bound scan/compaction.  Pipeline (all on the caller's stream):
//   mc_classify_walk  per grid point: 3 edge-slot flags + "voxel emits" -> info (u8), the voxel's sub-table
//                     id -> sub (u16, surface only), per-tile counters
//   mc_scan_tiles,    two-level exclusive scan of the tile counters, per-layer centre offsets, totals,
//   mc_scan           lists of the tiles that own anything
//   mc_vertices       wave scan of the slot flags -> vbase (u32, surface only) and the edge
//                     vertices, already in the reference's key order
//   mc_tri_records,   wave scan of triangle/centre counts -> one record per triangle, then one lane per
//   mc_tri_finish     triangle: final indices (smallest first) + centre vertices
//   tri_sort_*    LSD radix sort on the first index + run ordering (8-bit digits: histogram, per-digit row scan,
//                 stable scatter) -> lexicographic order
// Algorithmic bytes: 4 B/grid point + 24 B/vertex + 12 B/triangle.
// Compile with -ffp-contract=off: App. B specifies separately rounded
// products; the arithmetic intrinsics below make that explicit as well.
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include <mutex>
#include <type_traits>

#define DISN_MC_QUAL __device__
#include "../../include/disn_mc_tables.h"

namespace {

constexpr int BLK = 256;
constexpr int ITEMS = 4;
constexpr int TILE = BLK * ITEMS;
constexpr unsigned NOSUB = 0x3ffu;

// division of a 32-bit unsigned by a launch constant (Granlund-Montgomery round-up method):
// the grid kernels are instruction-bound, and two hardware divisions per point were half of them
struct FastDiv {
  unsigned m, s1, s2;
};
static FastDiv make_fastdiv(unsigned n) {   // n >= 1
  unsigned l = 0;
  while ((1ull << l) < n) ++l;
  FastDiv f;
  f.m = (unsigned)((((1ull << 32) * ((1ull << l) - n)) / n) + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l < 1 ? 0 : l - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned p, const FastDiv& f) {
  unsigned t = __umulhi(f.m, p);
  return (t + ((p - t) >> f.s1)) >> f.s2;
}

struct Dims {
  int nx, ny, nz;
  unsigned sxy;   // nx*ny: the caller's grid (sdf index p = (k ny + j) nx + i)
  unsigned P;     // nx*ny*nz
  int pitch;      // nx rounded up to a multiple of 4: row length of the per-point arrays (index q, see below)
  unsigned sxyq;  // pitch*ny
  unsigned Q;     // pitch*ny*nz
  int own0, own1;
  int z_offset;
  float iso;
  FastDiv dpitch, dny;
};
// grid coordinates of entry q of the per-point arrays
__device__ __forceinline__ void point_ijk(const Dims& d, unsigned q, int& i, int& j, int& k) {
  unsigned r = fdiv(q, d.dpitch);
  i = (int)(q - r * (unsigned)d.pitch);
  unsigned kk = fdiv(r, d.dny);
  j = (int)(r - kk * (unsigned)d.ny);
  k = (int)kk;
}

__device__ __forceinline__ float cval(const float* __restrict__ sdf, unsigned p, float iso) {
  float d = __fsub_rn(sdf[p], iso);  // B.1
  return d == 0.0f ? FLT_EPSILON : d;
}

// B.3 faceTest
__device__ int face_test(int f, const float* c) {
  const unsigned char* fv = DISN_MC_FACE[(f < 0 ? -f : f) - 1];
  float A = c[fv[0]], B = c[fv[1]], C = c[fv[2]], D = c[fv[3]];
  float ac = __fmul_rn(A, C), bd = __fmul_rn(B, D);
  return ((ac > bd) == (f > 0)) == (A > 0.0f);
}

// B.3 interiorTest
__device__ int interior_test(int s, const float* c) {
  float d10 = __fsub_rn(c[1], c[0]), d67 = __fsub_rn(c[6], c[7]);
  float d54 = __fsub_rn(c[5], c[4]), d23 = __fsub_rn(c[2], c[3]);
  float a = __fsub_rn(__fmul_rn(d10, d67), __fmul_rn(d54, d23));
  float b = __fsub_rn(__fadd_rn(__fsub_rn(__fmul_rn(c[0], d67), __fmul_rn(c[4], d23)), __fmul_rn(d10, c[7])),
                      __fmul_rn(d54, c[3]));
  double t = __ddiv_rn(-(double)b, __dadd_rn((double)a, (double)a));
  if (t < 0.0 || t > 1.0) return s > 0;
  double At = __dadd_rn((double)c[0], __dmul_rn((double)d10, t));
  double Bt = __dadd_rn((double)c[4], __dmul_rn((double)d54, t));
  double Ct = __dadd_rn((double)c[7], __dmul_rn((double)d67, t));
  double Dt = __dadd_rn((double)c[3], __dmul_rn((double)d23, t));
  int test = (At >= 0) | ((Bt >= 0) << 1) | ((Ct >= 0) << 2) | ((Dt >= 0) << 3);
  switch (test) {
    case 5: return (s > 0) ^ (__dmul_rn(Bt, Dt) <= __dmul_rn(At, Ct));
    case 10: return (s > 0) ^ (__dmul_rn(At, Ct) < __dmul_rn(Bt, Dt));
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    default: return s > 0;
  }
}

// B.2 + B.3: sub-table id, NOSUB when the voxel emits nothing
__device__ unsigned classify(const float* c) {
  int code = 0;
#pragma unroll
  for (int v = 0; v < 8; ++v) code |= (c[v] > 0.0f) << v;
  if (code == 0 || code == 255) return NOSUB;
  int cs = DISN_MC_CASE[code];
  if (cs == 0) return NOSUB;
  const signed char* ft = &DISN_MC_FACE_TESTS[7 * code];
  int n = ft[0], amb = 0;
  for (int t = 1; t <= n; ++t) amb |= face_test(ft[t], c) << (t - 1);
  int it = DISN_MC_INTERIOR_TEST[code];
  if (it) amb |= interior_test(it, c) << n;
  int sub = DISN_MC_AMBIGUITY[64 * cs + amb];
  if (sub < 0) return NOSUB;
  return DISN_MC_SUB_START[code] + sub;
}

__device__ void load_corners(const float* __restrict__ sdf, const Dims& d, unsigned p, float* c) {
  c[0] = cval(sdf, p, d.iso);
  c[1] = cval(sdf, p + 1, d.iso);
  c[2] = cval(sdf, p + 1 + d.nx, d.iso);
  c[3] = cval(sdf, p + d.nx, d.iso);
  c[4] = cval(sdf, p + d.sxy, d.iso);
  c[5] = cval(sdf, p + d.sxy + 1, d.iso);
  c[6] = cval(sdf, p + d.sxy + 1 + d.nx, d.iso);
  c[7] = cval(sdf, p + d.sxy + d.nx, d.iso);
}

// rare path of B.4/B.9: is voxel (i,j,k) inside the grid and active?
__device__ bool voxel_active(const float* __restrict__ sdf, const Dims& d, int i, int j, int k) {
  if (i < 0 || j < 0 || k < 0 || i >= d.nx - 1 || j >= d.ny - 1 || k >= d.nz - 1) return false;
  float c[8];
  load_corners(sdf, d, (unsigned)k * d.sxy + (unsigned)j * d.nx + i, c);
  return classify(c) != NOSUB;
}

// same-sign "crossed by underflow" edge: it owns a vertex only if one of the
// voxels around it is active (B.9)
__device__ bool underflow_edge_live(const float* __restrict__ sdf, const Dims& d, int i, int j, int k, int dir) {
  for (int a = -1; a <= 0; ++a)
    for (int b = -1; b <= 0; ++b) {
      bool live = dir == 0 ? voxel_active(sdf, d, i, j + a, k + b)
                : dir == 1 ? voxel_active(sdf, d, i + a, j, k + b)
                           : voxel_active(sdf, d, i + a, j + b, k);
      if (live) return true;
    }
  return false;
}

// block-wide exclusive scan of one unsigned per thread (BLK = 256 = 4 waves);
// returns the exclusive prefix, *total receives the block sum
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* lds /*[4]*/, unsigned* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned inc = wave_incl_scan(v, lane);
  __syncthreads();  // protect lds reuse between consecutive calls
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  unsigned pre = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < BLK / 64; ++q) {
    unsigned s = lds[q];
    if (q < w) pre += s;
    tot += s;
  }
  *total = tot;
  return pre + inc - v;
}

// counters per block: 0 E_all, 1 T_own, 2 C_own, 3 E_low, 4 E_own
constexpr int NCNT = 5;

// one grid point, cheap part: edge-slot flags (B.4, B.9) and the 8-bit corner code of its voxel
// (0 when the voxel does not exist).  c[] holds the shifted corner values; corners outside the grid are 1.0f.
__device__ __forceinline__ unsigned point_flags_code(const float* __restrict__ sdf, const Dims& d, int i, int j, int k,
                                                     const float* c, bool hx, bool hy, bool hz, unsigned* code) {
  unsigned flags = 0;
  const bool has[3] = {hx, hy, hz};
  const float cn[3] = {c[1], c[3], c[4]};
  unsigned cd = 0;
  if (hx && hy && hz) {
#pragma unroll
    for (int v = 0; v < 8; ++v) cd |= (unsigned)(c[v] > 0.0f) << v;
  }
  // An edge owns a vertex iff one of the (up to four) voxels around it is active (B.4: vertices are created by the
  // voxels that emit).  A sign change makes every voxel around the edge mixed, and a mixed voxel is active unless its
  // ambiguity lookup is impossible (sub < 0, B.3) -- which the tables allow only for the two fully alternating codes
  // 90 and 165 (case 13).  So the point's own voxel settles it whenever it exists and is not one of those two;
  // otherwise, and for same-sign products that underflowed to zero (B.9), the neighbours are classified.
  const bool own_settles = hx && hy && hz && cd != 90u && cd != 165u;
#pragma unroll
  for (int dir = 0; dir < 3; ++dir) {
    if (!has[dir]) continue;
    if (__fmul_rn(c[0], cn[dir]) <= 0.0f) {  // B.4
      bool live = (((c[0] > 0.0f) != (cn[dir] > 0.0f)) && own_settles) || underflow_edge_live(sdf, d, i, j, k, dir);
      if (live) flags |= 1u << dir;
    }
  }
  *code = cd;
  return flags;
}

// ---- per-point arrays: index space q -------------------------------------
// The caller's grid has rows of nx values (nx = sdf_res + 1 is odd); the arrays this file keeps per grid point --
// info (1 byte), sub (2 bytes, sparse), vbase (4 bytes, sparse) -- have rows of `pitch` = nx rounded up to a multiple
// of 4 entries, so that the classifier's runs of four points are whole aligned words: q = (k ny + j) pitch + i.
// Pad entries (nx <= i < pitch) hold 0 and own nothing.  q grows with the reference's vertex key, so ranks taken
// in q order are ranks in key order.  A TILE = 1024 consecutive q (the unit of the per-tile counters).
//   info[q]  bits 0..2: edge slots (x, y, z) owned by the point; bits 3..6: triangles its voxel emits (0..12);
//            bit 7: the voxel has a centre vertex -- everything the scans need, without a table look-up
//   sub[q]   sub-table id of the voxel, written only where the voxel emits
//   vbase[q] rank of the point's first edge slot, written only where a slot exists
constexpr unsigned INFO_NT_SHIFT = 3, INFO_NT_MASK = 15u, INFO_CENTRE_SHIFT = 7;
__device__ __forceinline__ unsigned sdf_index(const Dims& d, int i, int j, int k) {
  return (unsigned)k * d.sxy + (unsigned)j * (unsigned)d.nx + (unsigned)i;
}

// ---- mc_classify_walk: info, per-tile counters and centres per layer with every grid value fetched about once
// and almost no arithmetic away from the surface.
// A thread owns a column of 4 x 4 points (4 consecutive x, 4 consecutive rows) and walks L planes upwards: per
// step it loads the 5 x 5 values of the next plane above its points (two planes ahead of their use, so two loads
// are in flight per thread), 1.25 x (L+1)/L of the grid in total (the fifth value of a row is the next thread's
// first: the same cache line).  A step whose 50 values (this plane and the one above) all lie on one side of the
// iso value, none of them within 2^-60 of it -- no sign change, no product that could underflow (B.9) -- owns no
// vertex and no triangle: it is recognised with a min / max over the new plane (the shift v - iso is monotonic
// in v) and only writes its 16 "nothing here" bytes, one aligned word per row.  The other steps -- the surface, a
// few lanes of a wave at a time -- are handed to the whole wave: four such lanes' 4 x 4 points make 64 work
// items, each of which reads its corners again (cache hits) and takes the per-point path of the specification
// (edge flags B.4 / B.9, corner code, sub-table lookup B.3).  Counters go straight to the per-tile sums with
// global atomics (bsum and layer_centre start at zero): they fire only next to the surface.
// (Rounds 1-2 classified linear tiles of 1024 points, ~100 VALU instructions per point and 3.9x the grid in fabric
// traffic at sdf_res 1024; round 3 walked, but wrote 2-byte entries on the grid's own odd row length: ten
// partial-line stores per quiet step.)
constexpr int WROWS = 4;
constexpr float WALK_TINY = 8.67361737988403547e-19f;   // 2^-60: products of larger values cannot underflow
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load from a 4-byte aligned address

// raw values of plane k above the column (x0.., j0..); entries outside the grid copy `fill` (an existing value of
// the column: neutral for the min / max test)
__device__ __forceinline__ void walk_load_plane(const float* __restrict__ sdf, const Dims& d, int x0, int j0, int k, bool alive,
                                                float fill, float (&v)[WROWS + 1][ITEMS + 1]) {
  const bool fullx = x0 + ITEMS < d.nx;
#pragma unroll
  for (int r = 0; r <= WROWS; ++r) {
    const int j = j0 + r;
    if (!alive || k >= d.nz || j >= d.ny) {
#pragma unroll
      for (int t = 0; t <= ITEMS; ++t) v[r][t] = fill;
      continue;
    }
    const float* row = sdf + (size_t)k * d.sxy + (size_t)j * d.nx + x0;
    if (fullx) {
      const float4u q = *(const float4u*)row;
      v[r][0] = q[0]; v[r][1] = q[1]; v[r][2] = q[2]; v[r][3] = q[3];
      v[r][ITEMS] = row[ITEMS];
    } else {
#pragma unroll
      for (int t = 0; t <= ITEMS; ++t) v[r][t] = x0 + t < d.nx ? row[t] : fill;
    }
  }
}
__device__ __forceinline__ void walk_minmax(const float (&v)[WROWS + 1][ITEMS + 1], float& mn, float& mx) {
  float a[WROWS + 1], b[WROWS + 1];
#pragma unroll
  for (int r = 0; r <= WROWS; ++r) {
    a[r] = fminf(fminf(v[r][0], v[r][1]), fminf(fminf(v[r][2], v[r][3]), v[r][4]));
    b[r] = fmaxf(fmaxf(v[r][0], v[r][1]), fmaxf(fmaxf(v[r][2], v[r][3]), v[r][4]));
  }
  mn = fminf(fminf(a[0], a[1]), fminf(fminf(a[2], a[3]), a[4]));
  mx = fmaxf(fmaxf(b[0], b[1]), fmaxf(fmaxf(b[2], b[3]), b[4]));
}

// one grid point by the book (any position, boundary included): info entry, sub-table id and counters
__device__ __forceinline__ void walk_point(const float* __restrict__ sdf, const Dims& d, int i, int j, int k,
                                           uint8_t* __restrict__ info, uint16_t* __restrict__ subs,
                                           unsigned* __restrict__ bsum, unsigned* __restrict__ layer_centre) {
  const unsigned p = sdf_index(d, i, j, k);
  const unsigned q = (unsigned)k * d.sxyq + (unsigned)j * (unsigned)d.pitch + (unsigned)i;
  const bool hx = i < d.nx - 1, hy = j < d.ny - 1, hz = k < d.nz - 1;
  float c[8];
  c[0] = cval(sdf, p, d.iso);
  c[1] = hx ? cval(sdf, p + 1, d.iso) : 1.0f;
  c[3] = hy ? cval(sdf, p + d.nx, d.iso) : 1.0f;
  c[4] = hz ? cval(sdf, p + d.sxy, d.iso) : 1.0f;
  if (hx && hy && hz) {
    c[2] = cval(sdf, p + 1 + d.nx, d.iso);
    c[5] = cval(sdf, p + d.sxy + 1, d.iso);
    c[6] = cval(sdf, p + d.sxy + 1 + d.nx, d.iso);
    c[7] = cval(sdf, p + d.sxy + d.nx, d.iso);
  } else {
    c[2] = c[5] = c[6] = c[7] = 1.0f;
  }
  unsigned code;
  const unsigned fl = point_flags_code(sdf, d, i, j, k, c, hx, hy, hz, &code);
  const unsigned sub = (code != 0 && code != 255) ? classify(c) : (unsigned)NOSUB;
  unsigned nt = 0, nc = 0;
  if (sub != NOSUB) {
    const unsigned char* st = &DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub]];
    nt = st[0] / 3;   // 1..12
    nc = st[1];
    subs[q] = (uint16_t)sub;
  }
  info[q] = (uint8_t)(fl | (nt << INFO_NT_SHIFT) | (nc << INFO_CENTRE_SHIFT));
  const bool low = k < d.own0, owned = k >= d.own0 && k < d.own1;
  unsigned* bs = bsum + (size_t)(q / TILE) * NCNT;
  if (fl) {
    const unsigned ne = __popc(fl);
    atomicAdd(&bs[0], ne);
    if (low) atomicAdd(&bs[3], ne);
    if (owned) atomicAdd(&bs[4], ne);
  }
  if (owned && nt) {
    atomicAdd(&bs[1], nt);
    if (nc) {
      atomicAdd(&bs[2], 1u);
      atomicAdd(&layer_centre[k], 1u);
    }
  }
}

__global__ __launch_bounds__(BLK) void mc_classify_walk(const float* __restrict__ sdf, Dims d, uint8_t* __restrict__ info,
                                                         uint16_t* __restrict__ subs, unsigned* __restrict__ bsum,
                                                         unsigned* __restrict__ layer_centre, int nq, int ng, int L,
                                                         FastDiv dnq, FastDiv dng, unsigned nthreads) {
  const unsigned tid = blockIdx.x * BLK + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = tid < nthreads;
  int x0 = 0, j0 = 0, k0 = 0, k1 = 0;
  if (valid) {
    const unsigned t2 = fdiv(tid, dnq);
    const unsigned ch = fdiv(t2, dng);
    x0 = (int)(tid - t2 * (unsigned)nq) * ITEMS;   // nq = pitch / 4: the runs tile the padded row exactly
    j0 = (int)(t2 - ch * (unsigned)ng) * WROWS;
    k0 = (int)ch * L;
    k1 = min(d.nz, k0 + L);
  }
  const float fill = sdf[valid ? (size_t)k0 * d.sxy + (size_t)j0 * d.nx + x0 : 0];
  __shared__ unsigned queue[BLK / 64][128];   // per wave: surface points (q) waiting for the long path
  volatile unsigned* myq = queue[threadIdx.x >> 6];
  unsigned qn = 0;                             // wave-uniform
  float hi[WROWS + 1][ITEMS + 1], nn[WROWS + 1][ITEMS + 1];
  float lo_min, lo_max;
  walk_load_plane(sdf, d, x0, j0, k0, valid, fill, hi);
  walk_minmax(hi, lo_min, lo_max);
  walk_load_plane(sdf, d, x0, j0, k0 + 1, valid, fill, hi);
  for (int s = 0; s < L; ++s) {   // every lane of the wave stays in the loop: the surface steps are worked on together
    const int k = k0 + s;
    const bool alive = valid && k < k1;
    walk_load_plane(sdf, d, x0, j0, k + 2, alive && s + 1 < L, fill, nn);   // two planes ahead
    float hi_min, hi_max;
    walk_minmax(hi, hi_min, hi_max);
    const float cmin = __fsub_rn(fminf(lo_min, hi_min), d.iso), cmax = __fsub_rn(fmaxf(lo_max, hi_max), d.iso);
#ifdef WALK_FORCE_QUIET   // timing experiment only (wrong results)
    const bool quiet = cmin > -3e38f;
#else
    const bool quiet = cmin > WALK_TINY || cmax < -WALK_TINY;
#endif
    if (alive && quiet) {
      // 16 entries "no surface, no edge slot" (pad columns included): one aligned word per row
      const unsigned qrow = (unsigned)k * d.sxyq + (unsigned)j0 * (unsigned)d.pitch + (unsigned)x0;
#pragma unroll
      for (int r = 0; r < WROWS; ++r)
        if (j0 + r < d.ny) *(unsigned*)(info + qrow + (unsigned)r * (unsigned)d.pitch) = 0u;
    }
    unsigned long long need = __ballot(alive && !quiet);
    while (need) {   // wave-uniform: four columns at a time, one of their 64 points per lane
      int src[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        src[u] = need ? __ffsll((long long)need) - 1 : -1;
        if (need) need &= need - 1;
      }
      const int grp = lane >> 4, pt = lane & 15;
      const int from = grp == 0 ? src[0] : grp == 1 ? src[1] : grp == 2 ? src[2] : src[3];
      const int sl = from < 0 ? 0 : from;
      const int sx = __shfl(x0, sl, 64), sj = __shfl(j0, sl, 64), sk = __shfl(k, sl, 64);
      const int i = sx + (pt & 3), j = sj + (pt >> 2);
      // the point's own eight corners (those that exist) decide whether it is next to the surface at all
      bool active = false;
      unsigned q = 0;
      if (from >= 0 && j < d.ny) {   // i < pitch always
        q = (unsigned)sk * d.sxyq + (unsigned)j * (unsigned)d.pitch + (unsigned)i;
        if (i < d.nx) {
          const unsigned p = sdf_index(d, i, j, sk);
          const unsigned ox = i < d.nx - 1 ? 1u : 0u, oy = j < d.ny - 1 ? (unsigned)d.nx : 0u, oz = sk < d.nz - 1 ? d.sxy : 0u;
          const float a0 = sdf[p], a1 = sdf[p + ox], a2 = sdf[p + oy], a3 = sdf[p + ox + oy];
          const float b0 = sdf[p + oz], b1 = sdf[p + oz + ox], b2 = sdf[p + oz + oy], b3 = sdf[p + oz + ox + oy];
          const float mn = fminf(fminf(fminf(a0, a1), fminf(a2, a3)), fminf(fminf(b0, b1), fminf(b2, b3)));
          const float mx = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
          active = !(__fsub_rn(mn, d.iso) > WALK_TINY || __fsub_rn(mx, d.iso) < -WALK_TINY);
        }
        if (!active) info[q] = 0;   // quiet point or pad column
      }
      // surface points collect in the wave's queue and take the long path 64 at a time
      const unsigned long long act = __ballot(active);
      if (active) myq[qn + __popcll(act & ((1ull << lane) - 1ull))] = q;
      qn += __popcll(act);
      if (qn >= 64) {
        const unsigned pq = myq[lane];
        const unsigned rest = lane < (int)(qn - 64) ? myq[64 + lane] : 0u;
        int qi, qj, qk;
        point_ijk(d, pq, qi, qj, qk);
        walk_point(sdf, d, qi, qj, qk, info, subs, bsum, layer_centre);
        if (lane < (int)(qn - 64)) myq[lane] = rest;
        qn -= 64;
      }
    }
    lo_min = hi_min;
    lo_max = hi_max;
#pragma unroll
    for (int r = 0; r <= WROWS; ++r)
#pragma unroll
      for (int t = 0; t <= ITEMS; ++t) hi[r][t] = nn[r][t];
  }
  if (lane < (int)qn) {   // what is left in the queue
    int qi, qj, qk;
    point_ijk(d, myq[lane], qi, qj, qk);
    walk_point(sdf, d, qi, qj, qk, info, subs, bsum, layer_centre);
  }
}

// Two-level exclusive scan of the per-tile counters.  Level 1, one workgroup per chunk of SCAN_CHUNK
// tiles: boff[b][0..2] = prefix of (E_all, T_own, C_own) inside the chunk, csum[chunk][0..4] = chunk sums.
// Also compacts the tiles the emit kernels have to visit: actn[0] / actn[1] = number of tiles with an edge slot /
// with an owned triangle (both start at zero), their ids at act and act + nb.  A workgroup launched only to read
// its counter and leave still costs its launch (a million of them at sdf_res 1024).  The order of a list does not
// matter: every tile knows its own output positions.
constexpr int SCAN_CHUNK = BLK * ITEMS;   // 1024 tiles
__global__ __launch_bounds__(BLK) void mc_scan_tiles(const unsigned* __restrict__ bsum, unsigned nb,
                                                     unsigned* __restrict__ boff, unsigned* __restrict__ csum,
                                                     unsigned* __restrict__ actn, unsigned* __restrict__ act) {
  __shared__ unsigned lds[BLK / 64];
  __shared__ unsigned abase[2];
  const unsigned b0 = blockIdx.x * SCAN_CHUNK + threadIdx.x * ITEMS;
  unsigned v[ITEMS][NCNT];
#pragma unroll
  for (int t = 0; t < ITEMS; ++t)
#pragma unroll
    for (int n = 0; n < NCNT; ++n) v[t][n] = b0 + t < nb ? bsum[(size_t)(b0 + t) * NCNT + n] : 0u;
  {
    unsigned mine = 0;   // low half: tiles with slots, high half: tiles with owned triangles (<= 1024 per chunk)
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) mine += (v[t][0] != 0u ? 1u : 0u) | (v[t][1] != 0u ? 0x10000u : 0u);
    unsigned tot;
    const unsigned ex = block_excl_scan(mine, lds, &tot);
    if (threadIdx.x == 0) {
      abase[0] = (tot & 0xffffu) ? atomicAdd(&actn[0], tot & 0xffffu) : 0u;
      abase[1] = (tot >> 16) ? atomicAdd(&actn[1], tot >> 16) : 0u;
    }
    __syncthreads();
    unsigned* lv = act + abase[0] + (ex & 0xffffu);
    unsigned* lt = act + (size_t)nb + abase[1] + (ex >> 16);
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) {
      if (v[t][0] != 0u) *lv++ = b0 + t;
      if (v[t][1] != 0u) *lt++ = b0 + t;
    }
  }
#pragma unroll
  for (int n = 0; n < NCNT; ++n) {
    unsigned s = 0;
#pragma unroll
    for (int t = 0; t < ITEMS; ++t) s += v[t][n];
    unsigned tot;
    unsigned run = block_excl_scan(s, lds, &tot);
    if (threadIdx.x == 0) csum[(size_t)blockIdx.x * NCNT + n] = tot;
    if (n < 3) {
#pragma unroll
      for (int t = 0; t < ITEMS; ++t) {
        if (b0 + t < nb) boff[(size_t)(b0 + t) * 3 + n] = run;
        run += v[t][n];
      }
    }
  }
}

// Level 2, one workgroup of 1024 threads over the chunk sums (nb = number of chunks).  boff[c][0..2] =
// exclusive prefix of (E_all, T_own, C_own) over the chunks; totals[0..4] = sums of the five counters (u64),
// totals[5..6] = lengths of the two tile lists; layer[k] (in: centres of layer k) -> layer_asc[k] = centres in layers < k.
__global__ __launch_bounds__(1024) void mc_scan(const unsigned* __restrict__ bsum, unsigned nb, unsigned* __restrict__ boff,
                                                 unsigned long long* __restrict__ totals,
                                                 const unsigned* __restrict__ layer_centre, unsigned* __restrict__ layer_asc,
                                                 int nz, const unsigned* __restrict__ actn) {
  __shared__ unsigned long long part[1024];
  if (threadIdx.x < 2) totals[NCNT + threadIdx.x] = actn[threadIdx.x];
  const unsigned chunk = (nb + 1023) / 1024;
  const unsigned b0 = threadIdx.x * chunk, b1 = min(nb, b0 + chunk);
  for (int n = 0; n < NCNT; ++n) {
    unsigned long long s = 0;
    for (unsigned b = b0; b < b1; ++b) s += bsum[(size_t)b * NCNT + n];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (int o = 1; o < 1024; o <<= 1) {
      unsigned long long t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (threadIdx.x == 1023) totals[n] = part[1023];
    if (n < 3) {
      unsigned long long run = part[threadIdx.x] - s;
      for (unsigned b = b0; b < b1; ++b) {
        boff[(size_t)b * 3 + n] = (unsigned)run;
        run += bsum[(size_t)b * NCNT + n];
      }
    }
    __syncthreads();
  }
  {
    const int lchunk = (nz + 1023) / 1024;
    const int k0 = min(nz, (int)threadIdx.x * lchunk), k1 = min(nz, k0 + lchunk);
    unsigned long long s = 0;
    for (int k = k0; k < k1; ++k) s += layer_centre[k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      unsigned long long t = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    unsigned run = (unsigned)(part[threadIdx.x] - s);
    for (int k = k0; k < k1; ++k) {
      layer_asc[k] = run;
      run += layer_centre[k];
    }
    if (threadIdx.x == 1023) layer_asc[nz] = (unsigned)part[1023];
  }
}

struct EmitArgs {
  double lo[3], h[3];
  unsigned long long centre_base, edge_base;
  unsigned e_low;      // slots in planes below own0
  unsigned c_own;      // centre vertices owned
  double* verts;       // local output: [c_own centres][e_own edges]
  long long* keys;     // or null
  int* tris;
};

// p = sdf index of grid point (i, j, k)
__device__ __forceinline__ void edge_position(const float* __restrict__ sdf, const Dims& d, const EmitArgs& a, int i, int j,
                                              int k, unsigned p, int dir, double* pos) {
  const unsigned stride = dir == 0 ? 1u : dir == 1 ? (unsigned)d.nx : d.sxy;
  float ca = cval(sdf, p, d.iso), cb = cval(sdf, p + stride, d.iso);
  pos[0] = __dadd_rn(a.lo[0], __dmul_rn((double)i, a.h[0]));
  pos[1] = __dadd_rn(a.lo[1], __dmul_rn((double)j, a.h[1]));
  pos[2] = __dadd_rn(a.lo[2], __dmul_rn((double)(k + d.z_offset), a.h[2]));
  float diff = __fsub_rn(ca, cb);
  pos[dir] = __dadd_rn(pos[dir], __ddiv_rn(__dmul_rn((double)ca, a.h[dir]), (double)diff));  // B.4
}

// The emit kernels: ONE WAVE per tile, no workgroup barrier.  A lane takes 16 consecutive entries of the tile (one
// 16-byte load of info), a wave scan gives every lane its output positions, and the few entries that own
// something -- the surface crosses a row of 1024 points a handful of times -- go through the wave's LDS queue to
// a dense set of lanes.  (Round 3 ran a workgroup of 256 threads per tile: five barriers and a block scan for
// four or five surface points.)
constexpr int WPT = TILE / 64;   // entries per lane
__device__ __forceinline__ void load_info16(const uint8_t* __restrict__ info, const Dims& d, unsigned q0, unsigned* w) {
  if (q0 + WPT <= d.Q) {
    const uint4 v = *(const uint4*)(info + q0);   // q0 is a multiple of 16
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      w[c] = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (q0 + 4 * c + b < d.Q) w[c] |= (unsigned)info[q0 + 4 * c + b] << (8 * b);
    }
  }
}

// boff = prefix inside the chunk of SCAN_CHUNK tiles, coff = prefix over the chunks; tiles = the tiles with edge slots
__global__ __launch_bounds__(BLK) void mc_vertices(const float* __restrict__ sdf, Dims d, const uint8_t* __restrict__ info,
                                                    const unsigned* __restrict__ boff, const unsigned* __restrict__ coff,
                                                    unsigned* __restrict__ vbase, EmitArgs a, const unsigned* __restrict__ tiles,
                                                    unsigned ntiles) {
  // queue entry: position in the tile (10 bits) | flags << 10 | rank offset inside the tile << 13 (<= 3072)
  __shared__ unsigned queue[BLK / 64][TILE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned ti = blockIdx.x * (BLK / 64) + w;
  if (ti >= ntiles) return;
  const unsigned tile = tiles[ti];
  const unsigned q0 = tile * TILE + lane * WPT;
  unsigned wd[4];
  load_info16(info, d, q0, wd);
  unsigned mine = 0;   // low half: slots, high half: entries that own slots
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned f = wd[c] & 0x07070707u;
    mine += __popc(f) | (__popc((f | (f >> 1) | (f >> 2)) & 0x01010101u) << 16);
  }
  const unsigned inc = wave_incl_scan(mine, lane);
  const unsigned ntot = __shfl(inc, 63, 64) >> 16;
  unsigned vo = (inc - mine) & 0xffffu, e = (inc - mine) >> 16;
  volatile unsigned* myq = queue[w];
  if (mine) {
#pragma unroll
    for (int b = 0; b < WPT; ++b) {
      const unsigned fl = (wd[b >> 2] >> (8 * (b & 3))) & 7u;
      if (fl) {
        myq[e++] = (unsigned)(lane * WPT + b) | (fl << 10) | (vo << 13);
        vo += __popc(fl);
      }
    }
  }
  const unsigned tbase = boff[(size_t)tile * 3 + 0] + coff[(size_t)(tile / SCAN_CHUNK) * 3 + 0];
  for (unsigned x = lane; x < ntot; x += 64) {
    const unsigned ent = myq[x];
    const unsigned q = tile * TILE + (ent & 1023u), flags = (ent >> 10) & 7u, v0 = tbase + (ent >> 13);
    vbase[q] = v0;   // only slots that exist are ever looked up
    int i, j, k;
    point_ijk(d, q, i, j, k);
    if (k < d.own0 || k >= d.own1) continue;
    const unsigned p = sdf_index(d, i, j, k);
    unsigned long long out = (unsigned long long)a.c_own + (v0 - a.e_low);
#pragma unroll
    for (int dir = 0; dir < 3; ++dir) {
      if (!(flags >> dir & 1)) continue;
      double pos[3];
      edge_position(sdf, d, a, i, j, k, p, dir, pos);
      a.verts[3 * out + 0] = pos[0];
      a.verts[3 * out + 1] = pos[1];
      a.verts[3 * out + 2] = pos[2];
      if (a.keys)
        a.keys[out] = 3ll * (((long long)(k + d.z_offset) * d.ny + j) * d.nx + i) + dir;
      ++out;
    }
  }
}

// Triangles in two steps.  mc_tri_records (one wave per tile with owned triangles -- every centre vertex belongs to
// one of them): wave scan of (triangles, centres) per entry, then every emitting voxel leaves one 12-byte RECORD per
// triangle in the triangle's own slot of the output array: (q, sub-table id | triangle number << 10, centre rank).
// mc_tri_finish (one LANE per triangle, all lanes busy): turns the record into the three final vertex indices in
// place.  (Doing both per tile kept 4 or 5 lanes of a wave busy through a dozen dependent look-ups per triangle:
// 1.2 of the 4.6 ms at sdf_res 1024.)
__global__ __launch_bounds__(BLK) void mc_tri_records(Dims d, const uint8_t* __restrict__ info, const uint16_t* __restrict__ subs,
                                                       const unsigned* __restrict__ boff, const unsigned* __restrict__ coff,
                                                       int* __restrict__ tris, const unsigned* __restrict__ tiles,
                                                       unsigned ntiles) {
  const int lane = threadIdx.x & 63;
  const unsigned ti = blockIdx.x * (BLK / 64) + (threadIdx.x >> 6);
  if (ti >= ntiles) return;
  const unsigned tile = tiles[ti];
  const unsigned q0 = tile * TILE + lane * WPT;
  const unsigned own_lo = (unsigned)d.own0 * d.sxyq, own_hi = (unsigned)d.own1 * d.sxyq;   // owned layers as a range of q
  unsigned wd[4];
  load_info16(info, d, q0, wd);
  unsigned actmask = 0;   // bit b: entry b of this lane is an owned voxel that emits
  unsigned packed = 0;    // triangles | centres << 16 of this lane's entries
#pragma unroll
  for (int b = 0; b < WPT; ++b) {
    const unsigned e = (wd[b >> 2] >> (8 * (b & 3))) & 0xffu;
    if ((e >> INFO_NT_SHIFT) && q0 + b >= own_lo && q0 + b < own_hi) {
      actmask |= 1u << b;
      packed += ((e >> INFO_NT_SHIFT) & INFO_NT_MASK) | ((e >> INFO_CENTRE_SHIFT) << 16);
    }
  }
  unsigned off = wave_incl_scan(packed, lane) - packed;
  if (!actmask) return;
  const size_t ch = (size_t)(tile / SCAN_CHUNK) * 3;
  const unsigned tbase = boff[(size_t)tile * 3 + 1] + coff[ch + 1];
  const unsigned cbase = boff[(size_t)tile * 3 + 2] + coff[ch + 2];
#pragma unroll
  for (int b = 0; b < WPT; ++b) {
    if (!(actmask >> b & 1u)) continue;
    const unsigned e = (wd[b >> 2] >> (8 * (b & 3))) & 0xffu;
    const unsigned nt = (e >> INFO_NT_SHIFT) & INFO_NT_MASK, nc = e >> INFO_CENTRE_SHIFT;
    const unsigned sub = subs[q0 + b];
    int* rec = tris + 3 * (size_t)(tbase + (off & 0xffffu));
    for (unsigned tt = 0; tt < nt; ++tt) {
      rec[3 * tt + 0] = (int)(q0 + b);
      rec[3 * tt + 1] = (int)(sub | (tt << 10));
      rec[3 * tt + 2] = (int)(cbase + (off >> 16));
    }
    off += nt | (nc << 16);
  }
}

__global__ __launch_bounds__(BLK) void mc_tri_finish(const float* __restrict__ sdf, Dims d, const uint8_t* __restrict__ info,
                                                      const unsigned* __restrict__ vbase, const unsigned* __restrict__ layer_asc,
                                                      EmitArgs a, unsigned ntri) {
  const unsigned t = blockIdx.x * BLK + threadIdx.x;
  if (t >= ntri) return;
  int* o = a.tris + 3 * (size_t)t;
  const unsigned q = (unsigned)o[0], w1 = (unsigned)o[1], cb = (unsigned)o[2];
  const unsigned sub = w1 & 1023u, tt = w1 >> 10;
  const unsigned char* s = &DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub]];
  long long centre_idx = -1;
  if (s[1]) {
    // B.5/B.6: centres ordered by descending k, then ascending (j,i)
    int i, j, k;
    point_ijk(d, q, i, j, k);
    const unsigned in_layer = cb - layer_asc[k];
    const unsigned desc = (a.c_own - layer_asc[k + 1]) + in_layer;
    centre_idx = (long long)a.centre_base + desc;
    if (tt == 0) {   // the voxel's first triangle also writes the centre vertex: mean of ALL crossed edges (B.5)
      const unsigned p = sdf_index(d, i, j, k);
      double acc[3] = {0, 0, 0};
      int n = 0;
      for (int e = 0; e < 12; ++e) {
        int ca = DISN_MC_EDGE[e][0], dir = DISN_MC_EDGE[e][2];
        int di = DISN_MC_CORNER[ca][0], dj = DISN_MC_CORNER[ca][1], dk = DISN_MC_CORNER[ca][2];
        unsigned qa = q + di + dj * d.pitch + dk * d.sxyq;
        if (!(info[qa] >> dir & 1)) continue;
        double pos[3];
        edge_position(sdf, d, a, i + di, j + dj, k + dk, p + di + dj * d.nx + dk * d.sxy, dir, pos);
        acc[0] = __dadd_rn(acc[0], pos[0]);
        acc[1] = __dadd_rn(acc[1], pos[1]);
        acc[2] = __dadd_rn(acc[2], pos[2]);
        ++n;
      }
      a.verts[3 * (size_t)desc + 0] = __ddiv_rn(acc[0], (double)n);
      a.verts[3 * (size_t)desc + 1] = __ddiv_rn(acc[1], (double)n);
      a.verts[3 * (size_t)desc + 2] = __ddiv_rn(acc[2], (double)n);
      if (a.keys) a.keys[desc] = -(long long)(k + d.z_offset) * d.sxy + (long long)j * d.nx + i - 1;
    }
  }
  int v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int e = s[2 + 3 * tt + c];
    if (e == 12) {
      v[c] = (int)centre_idx;
    } else {
      int ca = DISN_MC_EDGE[e][0], dir = DISN_MC_EDGE[e][2];
      unsigned qa = q + DISN_MC_CORNER[ca][0] + DISN_MC_CORNER[ca][1] * d.pitch + DISN_MC_CORNER[ca][2] * d.sxyq;
      unsigned fl = (unsigned)info[qa] & 7u;
      v[c] = (int)(a.edge_base + (vbase[qa] - a.e_low) + __popc(fl & ((1u << dir) - 1u)));
    }
  }
  int m = 0;  // B.6: rotate, smallest index first
  if (v[1] < v[m]) m = 1;
  if (v[2] < v[m]) m = 2;
  const int r0 = v[m], r1 = v[m == 2 ? 0 : m + 1], r2 = v[m == 0 ? 2 : m - 1];
  o[0] = r0;
  o[1] = r1;
  o[2] = r2;
}

// the round-2 view of the classification for the parity tests: out[p] = sub-table id (0x3ff = none) | flags << 10
__global__ void mc_export_info(Dims d, const uint8_t* __restrict__ info, const uint16_t* __restrict__ subs,
                               uint16_t* __restrict__ out) {
  const unsigned q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= d.Q) return;
  int i, j, k;
  point_ijk(d, q, i, j, k);
  if (i >= d.nx) return;
  const unsigned b = info[q];
  out[sdf_index(d, i, j, k)] = (uint16_t)(((b >> INFO_NT_SHIFT) ? (unsigned)subs[q] : (unsigned)NOSUB) | ((b & 7u) << 10));
}

// ---- LSD radix sort of (a,b,c) int32 triples ------------------------------
constexpr int SORT_ITEMS = 8;
constexpr int SORT_TILE = BLK * SORT_ITEMS;

// `gate` (all sort kernels): null = always run; otherwise the launch is a no-op unless *gate != 0 (the fallback
// passes of a guarded sort are launched unconditionally -- no host round trip -- and run only when needed)
__global__ __launch_bounds__(BLK) void tri_sort_hist(const int* __restrict__ tris, unsigned long long n, int comp, int shift,
                                                      unsigned* __restrict__ hist, unsigned nb, const unsigned* gate) {
  __shared__ unsigned h[256];
  if (gate && *gate == 0) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long base = (unsigned long long)blockIdx.x * SORT_TILE;
  for (int r = 0; r < SORT_ITEMS; ++r) {
    unsigned long long t = base + r * BLK + threadIdx.x;
    if (t < n) atomicAdd(&h[((unsigned)tris[3 * t + comp] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];  // digit-major
}

// one workgroup per digit: exclusive scan of that digit's row hist[d][0..nb) in place,
// row sum to total[d]; the scatter kernel adds the 256-entry digit prefix itself
__global__ __launch_bounds__(BLK) void tri_sort_rowscan(unsigned* __restrict__ hist, unsigned nb,
                                                         unsigned* __restrict__ total, const unsigned* gate) {
  __shared__ unsigned lds[BLK / 64];
  if (gate && *gate == 0) return;
  unsigned* row = hist + (size_t)blockIdx.x * nb;
  unsigned carry = 0;
  for (unsigned b0 = 0; b0 < nb; b0 += BLK) {
    unsigned b = b0 + threadIdx.x;
    unsigned v = b < nb ? row[b] : 0u, tot;
    unsigned ex = block_excl_scan(v, lds, &tot);
    if (b < nb) row[b] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) total[blockIdx.x] = carry;
}

__global__ __launch_bounds__(BLK) void tri_sort_scatter(const int* __restrict__ in, int* __restrict__ out,
                                                         unsigned long long n, int comp, int shift,
                                                         const unsigned* __restrict__ hist, unsigned nb,
                                                         const unsigned* __restrict__ total, const unsigned* gate) {
  __shared__ unsigned basecnt[256];
  __shared__ unsigned wcnt[BLK / 64][256];
  __shared__ unsigned sl[BLK / 64];
  if (gate && *gate == 0) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  {
    unsigned tot;   // digit prefix over the 256 row totals, then this block's offset inside the row
    unsigned dig = block_excl_scan(total[threadIdx.x], sl, &tot);
    basecnt[threadIdx.x] = dig + hist[(size_t)threadIdx.x * nb + blockIdx.x];
  }
  unsigned long long base = (unsigned long long)blockIdx.x * SORT_TILE;
  for (int r = 0; r < SORT_ITEMS; ++r) {
#pragma unroll
    for (int q = 0; q < BLK / 64; ++q) wcnt[q][threadIdx.x] = 0;
    __syncthreads();
    unsigned long long t = base + r * BLK + threadIdx.x;
    const bool valid = t < n;
    int k0 = 0, k1 = 0, k2 = 0;
    unsigned dg = 0;
    if (valid) {
      k0 = in[3 * t];
      k1 = in[3 * t + 1];
      k2 = in[3 * t + 2];
      dg = ((unsigned)(comp == 0 ? k0 : comp == 1 ? k1 : k2) >> shift) & 255u;
    }
    // lanes of this wave holding the same digit
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      unsigned long long m = __ballot((dg >> b) & 1u);
      peers &= ((dg >> b) & 1u) ? m : ~m;
    }
    unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (valid && rank == 0) wcnt[w][dg] = __popcll(peers);
    __syncthreads();
    if (valid) {
      unsigned off = basecnt[dg] + rank;
      for (int q = 0; q < w; ++q) off += wcnt[q][dg];
      out[3 * (size_t)off] = k0;
      out[3 * (size_t)off + 1] = k1;
      out[3 * (size_t)off + 2] = k2;
    }
    __syncthreads();
    unsigned add = 0;
#pragma unroll
    for (int q = 0; q < BLK / 64; ++q) add += wcnt[q][threadIdx.x];
    basecnt[threadIdx.x] += add;
    __syncthreads();
  }
}

// After the stable sort on the first index: triangles with equal first index form short runs (a vertex is the
// smallest index of at most its valence many triangles).  Each thread finds its run and its rank inside it by
// (second, third, position) and writes the triple to its final place: the result is the full lexicographic order
// with a third of the radix passes.  Work is the sum of squared run lengths -- fine for meshes.  GUARD (the public
// sort entry points, whose input is arbitrary): a thread that has to look further than TIES_MAX_LOOK triples in
// either direction raises *flag and writes nothing; the caller's gated radix passes on the other two components
// then produce the result instead, so the work stays O(n) for any input.
constexpr unsigned TIES_MAX_LOOK = 64;
template <bool GUARD>
__global__ __launch_bounds__(BLK) void tri_sort_ties(const int* __restrict__ in, int* __restrict__ out, unsigned long long n,
                                                      unsigned* flag) {
  const unsigned long long t = (unsigned long long)blockIdx.x * BLK + threadIdx.x;
  if (t >= n) return;
  const int a = in[3 * t], b = in[3 * t + 1], c = in[3 * t + 2];
  unsigned long long lo = t, hi = t + 1;
  unsigned rank = 0, look = 0;
  while (lo > 0 && in[3 * (lo - 1)] == a) {
    if (GUARD && ++look > TIES_MAX_LOOK) { *flag = 1u; return; }
    --lo;
    const int b2 = in[3 * lo + 1], c2 = in[3 * lo + 2];
    rank += (b2 < b) || (b2 == b && c2 <= c);          // equal triples keep their order (stable)
  }
  look = 0;
  while (hi < n && in[3 * hi] == a) {
    if (GUARD && ++look > TIES_MAX_LOOK) { *flag = 1u; return; }
    const int b2 = in[3 * hi + 1], c2 = in[3 * hi + 2];
    rank += (b2 < b) || (b2 == b && c2 < c);
    ++hi;
  }
  int* o = out + 3 * (lo + rank);
  o[0] = a; o[1] = b; o[2] = c;
}

// dst = src when (*gate != 0) == want
__global__ void tri_copy_gated(const int* __restrict__ src, int* __restrict__ dst, unsigned long long n3, const unsigned* gate,
                               int want) {
  if ((*gate != 0) != (want != 0)) return;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n3;
       i += (unsigned long long)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

}  // namespace

struct McState {
  DevBuf info, subs, vbase, bsum, boff, csum, coff, layer, layer_asc, totals, act, sort_tmp, sort_hist;
  bool counted = false;
  long long walk_cap = 0;   // workgroups of mc_classify_walk the device holds at once (0 = not asked yet)
  const float* sdf = nullptr;
  Dims d{};
  int nz_global = 0;
  unsigned nb = 0;
  unsigned long long tot[NCNT + 2] = {0, 0, 0, 0, 0, 0, 0};   // the five counters; tiles with edge slots / with owned triangles
};

void disn_mc_state_free(DisnCtx* ctx) {
  if (!ctx->mc) return;
  McState* m = ctx->mc;
  DevBuf* all[] = {&m->info, &m->subs, &m->vbase, &m->bsum, &m->boff, &m->csum, &m->coff, &m->layer, &m->layer_asc, &m->totals, &m->act, &m->sort_tmp, &m->sort_hist};
  for (DevBuf* b : all) disn_release(*b);
  delete m;
  ctx->mc = nullptr;
}

static McState* mc_state(DisnCtx* ctx) {
  if (!ctx->mc) ctx->mc = new McState();
  return ctx->mc;
}

extern "C" int disn_mc_count(DisnCtx* ctx, const float* sdf, const DisnMcGrid* g, float iso, uint64_t counts[3],
                             void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (!sdf || !g || !counts) return disn_fail(ctx, DISN_E_ARG, "disn_mc_count: null argument");
  if (g->nx < 2 || g->ny < 2 || g->nz < 2) return disn_fail(ctx, DISN_E_ARG, "disn_mc_count: grid needs >= 2 points per axis");
  if (g->own0 < 0 || g->own1 > g->nz || g->own0 >= g->own1)
    return disn_fail(ctx, DISN_E_ARG, "disn_mc_count: bad owned plane range [%d,%d) of %d", g->own0, g->own1, g->nz);
  if (g->z_offset < 0 || g->z_offset + g->nz > g->nz_global)
    return disn_fail(ctx, DISN_E_ARG, "disn_mc_count: slab [%d,%d) outside global grid of %d planes", g->z_offset,
                     g->z_offset + g->nz, g->nz_global);
  const uint64_t P64 = (uint64_t)g->nx * g->ny * g->nz;
  if (P64 * 3 >= (1ull << 32)) return disn_fail(ctx, DISN_E_ARG, "disn_mc_count: grid too large for 32-bit slot indices");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  McState* m = mc_state(ctx);
  m->counted = false;
  Dims d;
  d.nx = g->nx; d.ny = g->ny; d.nz = g->nz;
  d.sxy = (unsigned)g->nx * g->ny;
  d.P = (unsigned)P64;
  d.pitch = (g->nx + ITEMS - 1) / ITEMS * ITEMS;
  const uint64_t Q64 = (uint64_t)d.pitch * g->ny * g->nz;   // <= 2 P
  d.sxyq = (unsigned)d.pitch * g->ny;
  d.Q = (unsigned)Q64;
  d.own0 = g->own0; d.own1 = g->own1;
  d.z_offset = g->z_offset;
  d.iso = iso;
  d.dpitch = make_fastdiv((unsigned)d.pitch);
  d.dny = make_fastdiv((unsigned)g->ny);
  const unsigned nb = (unsigned)((Q64 + TILE - 1) / TILE);
  DISN_TRY(disn_ensure(ctx, m->info, Q64 * sizeof(uint8_t)));
  DISN_TRY(disn_ensure(ctx, m->subs, Q64 * sizeof(uint16_t)));
  DISN_TRY(disn_ensure(ctx, m->bsum, ((size_t)nb * NCNT + 2) * sizeof(unsigned)));   // + the two list lengths
  DISN_TRY(disn_ensure(ctx, m->boff, (size_t)nb * 3 * sizeof(unsigned)));
  const unsigned nchunk = (nb + SCAN_CHUNK - 1) / SCAN_CHUNK;
  DISN_TRY(disn_ensure(ctx, m->csum, (size_t)nchunk * NCNT * sizeof(unsigned)));
  DISN_TRY(disn_ensure(ctx, m->coff, (size_t)nchunk * 3 * sizeof(unsigned)));
  DISN_TRY(disn_ensure(ctx, m->layer, (size_t)(g->nz + 1) * sizeof(unsigned)));
  DISN_TRY(disn_ensure(ctx, m->layer_asc, (size_t)(g->nz + 2) * sizeof(unsigned)));
  DISN_TRY(disn_ensure(ctx, m->totals, (NCNT + 2) * sizeof(unsigned long long)));
  DISN_TRY(disn_ensure(ctx, m->act, 2 * (size_t)nb * sizeof(unsigned)));
  DISN_HIP(ctx, hipMemsetAsync(m->layer.p, 0, (size_t)(g->nz + 1) * sizeof(unsigned), stream));
  DISN_HIP(ctx, hipMemsetAsync(m->bsum.p, 0, ((size_t)nb * NCNT + 2) * sizeof(unsigned), stream));
  unsigned* actn = (unsigned*)m->bsum.p + (size_t)nb * NCNT;
  {
    // columns of 4 x 4 points walking L planes.  Measured (tools/mc_walk_sweep.sh, sdf_res 256 / 512 / 1024): SHORT walks
    // win -- 4 planes at 1024 (count 1.37 ms against 2.85 ms with 129 planes), 2..4 at 512, 1 at 256.  The columns that
    // meet the surface do a hundred times the work of the others, and a long walk keeps that work in few waves; the
    // extra plane a walk reads at its seam ((L+1)/L) costs less than the imbalance.  So: the longest walk <= 4 planes
    // that still leaves four workgroups for every slot of the device.
    if (!m->walk_cap) {
      int per_cu = 0, cus = 0;
      DISN_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mc_classify_walk, BLK, 0));
      DISN_HIP(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
      m->walk_cap = (long long)(per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    }
    const int nq = d.pitch / ITEMS, ng = (g->ny + WROWS - 1) / WROWS;
    const long long per_chunk = (long long)nq * ng;
    int L = 4;
    while (L > 1 && (per_chunk * ((g->nz + L - 1) / L) + BLK - 1) / BLK < 4 * m->walk_cap) --L;
    if (const char* e = getenv("DISN_MC_WALK_L")) L = atoi(e) >= 1 ? atoi(e) : L;   // development switch: timing sweeps
    const long long nc = (g->nz + L - 1) / L;
    const long long nthreads = per_chunk * nc;
    mc_classify_walk<<<(unsigned)((nthreads + BLK - 1) / BLK), BLK, 0, stream>>>(
        sdf, d, (uint8_t*)m->info.p, (uint16_t*)m->subs.p, (unsigned*)m->bsum.p, (unsigned*)m->layer.p, nq, ng, L,
        make_fastdiv((unsigned)nq), make_fastdiv((unsigned)ng), (unsigned)nthreads);
  }
  mc_scan_tiles<<<nchunk, BLK, 0, stream>>>((const unsigned*)m->bsum.p, nb, (unsigned*)m->boff.p, (unsigned*)m->csum.p, actn,
                                            (unsigned*)m->act.p);
  mc_scan<<<1, 1024, 0, stream>>>((const unsigned*)m->csum.p, nchunk, (unsigned*)m->coff.p,
                                  (unsigned long long*)m->totals.p, (const unsigned*)m->layer.p,
                                  (unsigned*)m->layer_asc.p, g->nz, actn);
  DISN_HIP(ctx, hipGetLastError());
  DISN_HIP(ctx, hipMemcpyAsync(m->tot, m->totals.p, sizeof(m->tot), hipMemcpyDeviceToHost, stream));
  DISN_HIP(ctx, hipStreamSynchronize(stream));
  m->sdf = sdf;
  m->d = d;
  m->nz_global = g->nz_global;
  m->nb = nb;
  m->counted = true;
  counts[0] = m->tot[2];
  counts[1] = m->tot[4];
  counts[2] = m->tot[1];
  return DISN_OK;
}

// the sort proper on caller-chosen scratch: tmp = n*3 ints, hist = (nb*256 + 256 + 1) unsigned.  Every scratch word
// that is read has been written earlier in the same call, so neither buffer needs to start at zero.
// guard = false: marching-cubes output (runs of equal first index are bounded by the vertex valence);
// guard = true: arbitrary triangle lists (see tri_sort_ties).
constexpr size_t sort_hist_words(unsigned nb) { return (size_t)nb * 256 + 256 + 1; }
static int sort_tris_on(DisnCtx* ctx, int* tris, uint64_t n, uint64_t max_index, int* tmp, unsigned* hist,
                        hipStream_t stream, bool guard) {
  if (n < 2) return DISN_OK;
  int bits = 1;
  while (bits < 32 && (max_index >> bits)) ++bits;
  const int passes = (bits + 7) / 8;
  const unsigned nb = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
  unsigned* total = hist + (size_t)nb * 256;
  unsigned* flag = total + 256;
  int* a = tris;
  int* b = tmp;
  auto radix = [&](int comp, const unsigned* gate) {
    for (int ps = 0; ps < passes; ++ps) {
      tri_sort_hist<<<nb, BLK, 0, stream>>>(a, n, comp, 8 * ps, hist, nb, gate);
      tri_sort_rowscan<<<256, BLK, 0, stream>>>(hist, nb, total, gate);
      tri_sort_scatter<<<nb, BLK, 0, stream>>>(a, b, n, comp, 8 * ps, hist, nb, total, gate);
      int* t = a; a = b; b = t;
    }
  };
  const unsigned tb = (unsigned)((n + BLK - 1) / BLK);
  radix(0, nullptr);   // stable LSD radix sort on the first index only ...
  if (!guard) {
    tri_sort_ties<false><<<tb, BLK, 0, stream>>>(a, b, n, nullptr);   // ... then order the short runs
    if (b != tris) DISN_HIP(ctx, hipMemcpyAsync(tris, b, n * 3 * sizeof(int), hipMemcpyDeviceToDevice, stream));
    DISN_HIP(ctx, hipGetLastError());
    return DISN_OK;
  }
  DISN_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned), stream));
  tri_sort_ties<true><<<tb, BLK, 0, stream>>>(a, b, n, flag);
  int* short_runs = b;   // the result when no long run was met
  radix(2, flag);        // otherwise: the remaining passes of a full LSD sort, third index, second, first again
  radix(1, flag);
  radix(0, flag);
  const unsigned cb = tb < 4096 ? tb : 4096;
  if (short_runs != tris) tri_copy_gated<<<cb, BLK, 0, stream>>>(short_runs, tris, n * 3, flag, 0);
  if (a != tris) tri_copy_gated<<<cb, BLK, 0, stream>>>(a, tris, n * 3, flag, 1);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}

static int sort_tris(DisnCtx* ctx, McState* m, int* tris, uint64_t n, uint64_t max_index, hipStream_t stream, bool guard) {
  if (n < 2) return DISN_OK;
  const unsigned nb = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
  DISN_TRY(disn_ensure(ctx, m->sort_tmp, n * 3 * sizeof(int)));
  DISN_TRY(disn_ensure(ctx, m->sort_hist, sort_hist_words(nb) * sizeof(unsigned)));
  return sort_tris_on(ctx, tris, n, max_index, (int*)m->sort_tmp.p, (unsigned*)m->sort_hist.p, stream, guard);
}

// ---- allocation-free triangle sort (disn_sort_triangles_static) ------------
// For callers that cannot call hipMalloc (HIP graph capture, tight serving loops): the ping-pong buffer and the
// digit histograms live in a file-scope device array instead of ctx-owned hipMalloc memory.  One copy of the
// array exists per GPU the library runs on, at its own address, for the life of the process; `scratch` below is
// only the host-side handle of the symbol -- the device address is taken once per device with
// hipGetSymbolAddress (disn_mc_static_init, called by disn_create with that device current) and only that
// pointer is ever handed to kernels or copies.
constexpr uint64_t SORT_STATIC_MAX_TRIS = 1ull << 20;                                  // 1 Mi triangles
constexpr unsigned SORT_STATIC_MAX_NB = (unsigned)(SORT_STATIC_MAX_TRIS / SORT_TILE);   // 512 sort tiles
constexpr size_t SORT_STATIC_TMP_BYTES = SORT_STATIC_MAX_TRIS * 3 * sizeof(int);        // 12 MiB
constexpr size_t SORT_STATIC_HIST_BYTES = sort_hist_words(SORT_STATIC_MAX_NB) * sizeof(unsigned);
static __device__ __align__(256) unsigned char scratch[SORT_STATIC_TMP_BYTES + SORT_STATIC_HIST_BYTES];

constexpr int MAX_DEVICES = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[MAX_DEVICES];   // device address of `scratch` on each device, null until looked up

int disn_mc_static_init(DisnCtx* ctx) {
  if (ctx->device < 0 || ctx->device >= MAX_DEVICES)
    return disn_fail(ctx, DISN_E_ARG, "device id %d outside the static scratch table", ctx->device);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[ctx->device]) {
    DISN_HIP(ctx, hipSetDevice(ctx->device));
    void* ptr = nullptr;
    DISN_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return disn_fail(ctx, DISN_E_HIP, "hipGetSymbolAddress(scratch) returned null");
    g_scratch_ptr[ctx->device] = ptr;
  }
  ctx->static_scratch = g_scratch_ptr[ctx->device];
  return DISN_OK;
}

extern "C" int disn_sort_triangles_static(DisnCtx* ctx, int32_t* tris, uint64_t n, uint64_t max_index, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (n && !tris) return disn_fail(ctx, DISN_E_ARG, "disn_sort_triangles_static: null buffer");
  if (max_index >= (1ull << 31)) return disn_fail(ctx, DISN_E_ARG, "disn_sort_triangles_static: indices exceed int32");
  if (n > SORT_STATIC_MAX_TRIS)
    return disn_fail(ctx, DISN_E_ARG,
                     "disn_sort_triangles_static: %llu triangles exceed the static scratch limit of %llu "
                     "(use disn_sort_triangles, which allocates)",
                     (unsigned long long)n, (unsigned long long)SORT_STATIC_MAX_TRIS);
  if (!ctx->static_scratch) return disn_fail(ctx, DISN_E_STATE, "disn_sort_triangles_static: static scratch not initialised");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  unsigned char* base = (unsigned char*)ctx->static_scratch;   // the device pointer, never the symbol
  return sort_tris_on(ctx, tris, n, max_index, (int*)base, (unsigned*)(base + SORT_STATIC_TMP_BYTES), (hipStream_t)stream_,
                      true);
}

extern "C" int disn_mc_emit(DisnCtx* ctx, const double* bbox, uint64_t centre_base, uint64_t edge_base, double* verts,
                            int64_t* keys, int32_t* tris, int sort, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  McState* m = ctx->mc;
  if (!m || !m->counted) return disn_fail(ctx, DISN_E_STATE, "disn_mc_emit: call disn_mc_count first");
  if (!bbox) return disn_fail(ctx, DISN_E_ARG, "disn_mc_emit: null bbox");
  const uint64_t nv = m->tot[2] + m->tot[4], nt = m->tot[1];
  if ((nv && !verts) || (nt && !tris)) return disn_fail(ctx, DISN_E_ARG, "disn_mc_emit: null output buffer");
  const uint64_t max_index = (centre_base + m->tot[2] > edge_base + m->tot[0] ? centre_base + m->tot[2] : edge_base + m->tot[0]);
  if (max_index >= (1ull << 31)) return disn_fail(ctx, DISN_E_ARG, "disn_mc_emit: vertex indices exceed int32");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  const Dims& d = m->d;
  DISN_TRY(disn_ensure(ctx, m->vbase, (size_t)d.Q * sizeof(unsigned)));
  EmitArgs a;
  const int ng[3] = {d.nx, d.ny, m->nz_global};
  for (int q = 0; q < 3; ++q) {
    a.lo[q] = bbox[q];
    a.h[q] = (bbox[3 + q] - bbox[q]) / (double)(ng[q] - 1);
  }
  a.centre_base = centre_base;
  a.edge_base = edge_base;
  a.e_low = (unsigned)m->tot[3];
  a.c_own = (unsigned)m->tot[2];
  a.verts = verts;
  a.keys = (long long*)keys;
  a.tris = tris;
  const unsigned* act = (const unsigned*)m->act.p;
  const unsigned nvl = (unsigned)m->tot[NCNT], ntl = (unsigned)m->tot[NCNT + 1], wpb = BLK / 64;   // one wave per listed tile
  if (nvl)
    mc_vertices<<<(nvl + wpb - 1) / wpb, BLK, 0, stream>>>(m->sdf, d, (const uint8_t*)m->info.p, (const unsigned*)m->boff.p,
                                                           (const unsigned*)m->coff.p, (unsigned*)m->vbase.p, a, act, nvl);
  if (ntl) {
    mc_tri_records<<<(ntl + wpb - 1) / wpb, BLK, 0, stream>>>(d, (const uint8_t*)m->info.p, (const uint16_t*)m->subs.p,
                                                              (const unsigned*)m->boff.p, (const unsigned*)m->coff.p, tris,
                                                              act + (size_t)m->nb, ntl);
    mc_tri_finish<<<(unsigned)((nt + BLK - 1) / BLK), BLK, 0, stream>>>(m->sdf, d, (const uint8_t*)m->info.p,
                                                                        (const unsigned*)m->vbase.p,
                                                                        (const unsigned*)m->layer_asc.p, a, (unsigned)nt);
  }
  DISN_HIP(ctx, hipGetLastError());
  if (sort) DISN_TRY(sort_tris(ctx, m, tris, nt, max_index, stream, false));
  return DISN_OK;
}

extern "C" int disn_mc_debug_info(DisnCtx* ctx, uint16_t* out, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  McState* m = ctx->mc;
  if (!m || !m->counted) return disn_fail(ctx, DISN_E_STATE, "disn_mc_debug_info: call disn_mc_count first");
  if (!out) return disn_fail(ctx, DISN_E_ARG, "disn_mc_debug_info: null output");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  mc_export_info<<<(m->d.Q + BLK - 1) / BLK, BLK, 0, (hipStream_t)stream_>>>(m->d, (const uint8_t*)m->info.p,
                                                                               (const uint16_t*)m->subs.p, out);
  DISN_HIP(ctx, hipGetLastError());
  return DISN_OK;
}

extern "C" int disn_sort_triangles(DisnCtx* ctx, int32_t* tris, uint64_t n, uint64_t max_index, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (n && !tris) return disn_fail(ctx, DISN_E_ARG, "disn_sort_triangles: null buffer");
  if (max_index >= (1ull << 31)) return disn_fail(ctx, DISN_E_ARG, "disn_sort_triangles: indices exceed int32");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  return sort_tris(ctx, mc_state(ctx), tris, n, max_index, (hipStream_t)stream_, true);
}
