// df.hip -- distance field of a triangle mesh on a regular grid for gfx950: the GPU side of the replacement for the
// reference's ground-truth tool `computeDistanceField <obj> res res res -s -m 1 [-g sigma]` (closed Vega-FEM binary;
// call site preprocessing/create_point_sdf_grid.py:200-210).  The tool's pipeline for `-m 1` is the
// published one -- Xu & Barbic, "Signed Distance Fields for Polygon Soup Meshes", Graphics Interface 2014: unsigned
// field -> offset isosurface at sigma (marching cubes) -> exterior part only -> signed field of that closed surface
// (pseudonormal signs, Baerentzen & Aanaes 2005) -> shifted back by sigma.  This file holds its two device pieces;
// the host side (disn_amd/distance_field.py) strings them together with the library's own marching cubes.
//   disn_df_eval   exact distance from every grid point to the nearest triangle, float64 arithmetic, unsigned or
//                  signed by the pseudonormal of the closest feature
//   disn_df_flood  which grid points are connected to the box boundary through values above a level
// disn_df_eval: one workgroup per brick of 8^3 grid points.  Triangles come in chunks of 128 (the caller orders them
// along a space-filling curve) with a bounding sphere each.  A brick first bounds its points' distances from above with
// the spheres alone (ub = min over chunks of |centre - sphere centre| + sphere radius, + brick radius), then tests
// exactly only the chunks whose sphere can reach below that bound, re-tightened by the distances found so far -- a
// handful of chunks per brick instead of all of them.  The result is the minimum over ALL triangles of (squared
// distance, triangle index) in lexicographic order, so it does not depend on the culling or on the order of the tests.
// Compile with -ffp-contract=off (df_point_triangle.h fixes the operation sequence).
#include <float.h>
#include <math.h>

#include "common.h"

#define DF_QUAL __device__ __forceinline__
#include "df_point_triangle.h"

namespace {

constexpr int DF_CHUNK = 128;    // triangles per chunk
constexpr int DF_BR = 8;         // brick edge in grid points
constexpr int DF_BLK = 256;      // threads per brick: two points each

struct DfGrid {
  int nx, ny, nz;
  int bx, by, bz;     // bricks per axis
  double lo[3], h[3];
};

// bounding sphere of each chunk: centre of the vertices' box, radius to the farthest vertex
__global__ __launch_bounds__(DF_CHUNK) void k_df_spheres(const double* __restrict__ geom, unsigned n_tris,
                                                         double* __restrict__ spheres) {
  __shared__ double red[6][DF_CHUNK];
  const unsigned t = blockIdx.x * DF_CHUNK + threadIdx.x;
  const bool valid = t < n_tris;
  const unsigned tt = valid ? t : blockIdx.x * DF_CHUNK;   // the chunk's first triangle always exists
  const double* g = geom + (size_t)tt * 9;
  double mn[3], mx[3];
  for (int a = 0; a < 3; ++a) {
    mn[a] = fmin(g[a], fmin(g[3 + a], g[6 + a]));
    mx[a] = fmax(g[a], fmax(g[3 + a], g[6 + a]));
    red[a][threadIdx.x] = mn[a];
    red[3 + a][threadIdx.x] = mx[a];
  }
  __syncthreads();
  for (int s = DF_CHUNK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int a = 0; a < 3; ++a) {
        red[a][threadIdx.x] = fmin(red[a][threadIdx.x], red[a][threadIdx.x + s]);
        red[3 + a][threadIdx.x] = fmax(red[3 + a][threadIdx.x], red[3 + a][threadIdx.x + s]);
      }
    __syncthreads();
  }
  double c[3];
  for (int a = 0; a < 3; ++a) c[a] = 0.5 * (red[a][0] + red[3 + a][0]);
  __syncthreads();
  double r2 = 0.0;
  for (int v = 0; v < 3; ++v) {
    double d[3] = {g[3 * v] - c[0], g[3 * v + 1] - c[1], g[3 * v + 2] - c[2]};
    r2 = fmax(r2, df_dot(d, d));
  }
  red[0][threadIdx.x] = r2;
  __syncthreads();
  for (int s = DF_CHUNK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[0][threadIdx.x] = fmax(red[0][threadIdx.x], red[0][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* o = spheres + (size_t)blockIdx.x * 4;
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
    o[3] = sqrt(red[0][0]) * (1.0 + 1e-12);   // rounded up: the sphere must contain its vertices
  }
}

__device__ __forceinline__ double block_reduce(double v, double* lds, bool take_max) {
  // DF_BLK threads; lds holds DF_BLK / 64 doubles
  for (int o = 32; o > 0; o >>= 1) {
    const double u = __shfl_xor(v, o, 64);
    v = take_max ? fmax(v, u) : fmin(v, u);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = lds[0];
  for (int w = 1; w < DF_BLK / 64; ++w) r = take_max ? fmax(r, lds[w]) : fmin(r, lds[w]);
  return r;
}

template <bool SIGNED>
__global__ __launch_bounds__(DF_BLK) void k_df_eval(const double* __restrict__ geom, const double* __restrict__ normals,
                                                     unsigned n_tris, const double* __restrict__ spheres, unsigned n_chunks,
                                                     DfGrid g, float* __restrict__ out) {
  __shared__ double sgeom[DF_CHUNK * 9];
  __shared__ double ssph[DF_CHUNK * 4];   // per triangle of the staged chunk: centroid and radius to its farthest corner
  __shared__ double red[DF_BLK / 64];
  __shared__ unsigned cand[DF_BLK];
  __shared__ unsigned ncand;
  // this brick and this thread's two points
  const int b = blockIdx.x;
  const int bi = b % g.bx, bj = (b / g.bx) % g.by, bk = b / (g.bx * g.by);
  const int lx = threadIdx.x & 7, ly = (threadIdx.x >> 3) & 7, lz = threadIdx.x >> 6;
  const int i = bi * DF_BR + lx, j = bj * DF_BR + ly;
  const int k2[2] = {bk * DF_BR + lz, bk * DF_BR + lz + 4};
  double p[2][3];
  bool have[2];
  for (int u = 0; u < 2; ++u) {
    have[u] = i < g.nx && j < g.ny && k2[u] < g.nz;
    p[u][0] = g.lo[0] + (double)i * g.h[0];
    p[u][1] = g.lo[1] + (double)j * g.h[1];
    p[u][2] = g.lo[2] + (double)k2[u] * g.h[2];
  }
  // brick sphere: centre of its 8^3 points, radius to a corner point
  double bc[3], br2 = 0.0;
  const int b0[3] = {bi * DF_BR, bj * DF_BR, bk * DF_BR};
  for (int a = 0; a < 3; ++a) {
    bc[a] = g.lo[a] + ((double)b0[a] + 0.5 * (DF_BR - 1)) * g.h[a];
    const double e = 0.5 * (DF_BR - 1) * g.h[a];
    br2 += e * e;
  }
  const double brad = sqrt(br2) * (1.0 + 1e-12);
  double best[2] = {DBL_MAX, DBL_MAX}, sbest[2] = {DBL_MAX, DBL_MAX};   // squared distance; distance, rounded up
  unsigned bidx[2] = {0xffffffffu, 0xffffffffu};
  double ub = DBL_MAX;
  // exact tests of one chunk against this thread's two points, then the brick's new bound: its farthest point
  auto test_chunk = [&](unsigned ch) {
    const unsigned t0 = ch * DF_CHUNK;
    const unsigned nt = min((unsigned)DF_CHUNK, n_tris - t0);
    __syncthreads();
    for (unsigned x = threadIdx.x; x < nt * 9; x += DF_BLK) sgeom[x] = geom[(size_t)t0 * 9 + x];
    __syncthreads();
    if (threadIdx.x < nt) {
      const double* tg = sgeom + threadIdx.x * 9;
      double c[3], r2 = 0.0;
      for (int a = 0; a < 3; ++a) c[a] = (tg[a] + tg[3 + a] + tg[6 + a]) / 3.0;
      for (int v = 0; v < 3; ++v) {
        const double d[3] = {tg[3 * v] - c[0], tg[3 * v + 1] - c[1], tg[3 * v + 2] - c[2]};
        r2 = fmax(r2, df_dot(d, d));
      }
      double* o = ssph + threadIdx.x * 4;
      o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
      o[3] = sqrt(r2) * (1.0 + 1e-12);
    }
    __syncthreads();
    for (unsigned t = 0; t < nt; ++t) {
      const double* tg = sgeom + t * 9;
      const double* sp = ssph + t * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        // a dozen operations decide for most triangles: the whole triangle lies in its sphere, so it is farther
        // than |p - centre| - radius; strictly farther than the best so far -> it cannot win, not even a tie
        const double dc[3] = {p[u][0] - sp[0], p[u][1] - sp[1], p[u][2] - sp[2]};
        const double lim = sbest[u] + sp[3];
        if (df_dot(dc, dc) > lim * lim) continue;
        int region;
        double q[3];
        const double d2 = df_point_triangle(p[u], tg, tg + 3, tg + 6, &region, q);
        const unsigned idx = t0 + t;
        if (d2 < best[u] || (d2 == best[u] && idx < bidx[u])) {
          best[u] = d2;
          bidx[u] = idx;
          sbest[u] = sqrt(d2) * (1.0 + 1e-12);
        }
      }
    }
    double worst = 0.0;
    for (int u = 0; u < 2; ++u)
      if (have[u]) worst = fmax(worst, best[u]);
    worst = block_reduce(worst, red, true);
    if (worst < DBL_MAX) ub = fmin(ub, sqrt(worst) * (1.0 + 1e-9));
  };
  // phase A: the chunk whose sphere promises the smallest distance to the brick's centre is tested first, so that the
  // bound is tight before anything else is looked at (in list order the first chunks tested are usually far ones,
  // and a loose bound lets most of the mesh through: 4 ms per brick on a 26 K-triangle mesh, measured)
  unsigned first = 0;
  {
    double mine = DBL_MAX;
    unsigned mine_c = 0;
    for (unsigned c = threadIdx.x; c < n_chunks; c += DF_BLK) {
      const double* s = spheres + (size_t)c * 4;
      const double d[3] = {s[0] - bc[0], s[1] - bc[1], s[2] - bc[2]};
      const double v = sqrt(df_dot(d, d)) + s[3];
      if (v < mine) { mine = v; mine_c = c; }
    }
    const double lowest = block_reduce(mine, red, false);
    if (threadIdx.x == 0) ncand = 0xffffffffu;
    __syncthreads();
    if (mine == lowest) atomicMin(&ncand, mine_c);   // ties: the lowest chunk index, so the choice is deterministic
    __syncthreads();
    first = ncand;
    ub = (lowest + brad) * (1.0 + 1e-9);
    __syncthreads();
  }
  test_chunk(first);
  // phase B: the other chunks that can still reach below the bound
  for (unsigned c0 = 0; c0 < n_chunks; c0 += DF_BLK) {
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    const unsigned c = c0 + threadIdx.x;
    if (c < n_chunks && c != first) {
      const double* s = spheres + (size_t)c * 4;
      const double d[3] = {s[0] - bc[0], s[1] - bc[1], s[2] - bc[2]};
      const double lb = sqrt(df_dot(d, d)) - s[3] - brad;
      if (lb <= ub) cand[atomicAdd(&ncand, 1u)] = c;   // order inside the list does not matter: the result is a minimum
    }
    __syncthreads();
    const unsigned nc = ncand;
    for (unsigned e = 0; e < nc; ++e) {
      const unsigned ch = cand[e];
      // has the bound tightened enough to drop this chunk?  (every point of the brick already closer than it can be)
      const double* s = spheres + (size_t)ch * 4;
      const double d[3] = {s[0] - bc[0], s[1] - bc[1], s[2] - bc[2]};
      const double lb = sqrt(df_dot(d, d)) - s[3] - brad;
      if (lb > ub) continue;   // block-uniform
      test_chunk(ch);
    }
    __syncthreads();
  }
  for (int u = 0; u < 2; ++u) {
    if (!have[u]) continue;
    double dist = sqrt(best[u]);
    if (SIGNED && bidx[u] != 0xffffffffu) {   // no winner (non-finite input): leave the huge unsigned value, read nothing
      const double* tg = geom + (size_t)bidx[u] * 9;
      int region;
      double q[3];
      (void)df_point_triangle(p[u], tg, tg + 3, tg + 6, &region, q);
      const double* n = normals + (size_t)bidx[u] * 21 + 3 * region;
      const double d[3] = {p[u][0] - q[0], p[u][1] - q[1], p[u][2] - q[2]};
      if (df_dot(d, n) < 0.0) dist = -dist;
    }
    out[((size_t)k2[u] * g.ny + j) * g.nx + i] = (float)dist;
  }
}

// ---- exterior flood fill --------------------------------------------------
// ext[p] = 1 iff grid point p is connected to the boundary of the box through points with field > level
// (6-neighbourhood).  One thread per grid line sweeps it forwards and backwards; the host repeats x, y, z sweeps
// until nothing changes.
__global__ void k_flood_init(const float* __restrict__ f, float level, int nx, int ny, int nz, unsigned char* __restrict__ ext) {
  const size_t n = (size_t)nx * ny * nz;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(p % nx), j = (int)((p / nx) % ny), k = (int)(p / ((size_t)nx * ny));
    const bool wall = i == 0 || j == 0 || k == 0 || i == nx - 1 || j == ny - 1 || k == nz - 1;
    ext[p] = (wall && f[p] > level) ? 1 : 0;
  }
}
__global__ void k_flood_sweep(const float* __restrict__ f, float level, int n_lines, int n0, size_t s0, int n1, size_t s1,
                              int len, size_t stride, unsigned char* __restrict__ ext, unsigned* __restrict__ changed) {
  // line (a, b): start = a s0 + b s1, a < n0, b < n1, `len` points `stride` apart
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_lines) return;
  const size_t base = (size_t)(l % n0) * s0 + (size_t)(l / n0) * s1;
  bool any = false;
  for (int dir = 0; dir < 2; ++dir) {
    bool run = false;
    for (int t = 0; t < len; ++t) {
      const size_t p = base + (size_t)(dir ? len - 1 - t : t) * stride;
      if (f[p] > level) {
        if (ext[p]) run = true;
        else if (run) { ext[p] = 1; any = true; }
      } else {
        run = false;
      }
    }
  }
  if (any) atomicOr(changed, 1u);
}

}  // namespace

// The sphere and flag buffers are allocated per call and freed before returning: a distance field is a once-per-mesh
// preprocessing step, not a hot loop.
extern "C" int disn_df_eval(DisnCtx* ctx, const double* geom, const double* normals, uint64_t n_tris, int nx, int ny, int nz,
                            const double* bbox, float* out, void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (!geom || !bbox || !out) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: null argument");
  if (n_tris == 0) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: the mesh has no triangles");
  if (n_tris >= (1ull << 31)) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: too many triangles");
  if (nx < 2 || ny < 2 || nz < 2) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: grid needs >= 2 points per axis");
  for (int a = 0; a < 3; ++a)
    if (!(bbox[3 + a] > bbox[a])) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: empty bounding box");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  DfGrid g;
  g.nx = nx; g.ny = ny; g.nz = nz;
  g.bx = (nx + DF_BR - 1) / DF_BR; g.by = (ny + DF_BR - 1) / DF_BR; g.bz = (nz + DF_BR - 1) / DF_BR;
  const int n[3] = {nx, ny, nz};
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = bbox[a];
    g.h[a] = (bbox[3 + a] - bbox[a]) / (double)(n[a] - 1);
  }
  const uint64_t bricks = (uint64_t)g.bx * g.by * g.bz;
  if (bricks >= (1ull << 31)) return disn_fail(ctx, DISN_E_ARG, "disn_df_eval: grid too large");
  const unsigned n_chunks = (unsigned)((n_tris + DF_CHUNK - 1) / DF_CHUNK);
  DevBuf spheres;
  DISN_TRY(disn_ensure(ctx, spheres, (size_t)n_chunks * 4 * sizeof(double)));
  k_df_spheres<<<n_chunks, DF_CHUNK, 0, stream>>>(geom, (unsigned)n_tris, (double*)spheres.p);
  if (normals)
    k_df_eval<true><<<(unsigned)bricks, DF_BLK, 0, stream>>>(geom, normals, (unsigned)n_tris, (const double*)spheres.p, n_chunks, g, out);
  else
    k_df_eval<false><<<(unsigned)bricks, DF_BLK, 0, stream>>>(geom, nullptr, (unsigned)n_tris, (const double*)spheres.p, n_chunks, g, out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(stream);   // the sphere buffer is released below
  disn_release(spheres);
  if (e != hipSuccess) return disn_fail(ctx, DISN_E_HIP, "disn_df_eval: %s", hipGetErrorString(e));
  return DISN_OK;
}

extern "C" int disn_df_flood(DisnCtx* ctx, const float* field, float level, int nx, int ny, int nz, uint8_t* exterior,
                             void* stream_) {
  if (!ctx) return DISN_E_ARG;
  if (!field || !exterior) return disn_fail(ctx, DISN_E_ARG, "disn_df_flood: null argument");
  if (nx < 1 || ny < 1 || nz < 1) return disn_fail(ctx, DISN_E_ARG, "disn_df_flood: empty grid");
  DISN_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t)stream_;
  DevBuf flag;
  DISN_TRY(disn_ensure(ctx, flag, sizeof(unsigned)));
  const size_t sx = 1, sy = (size_t)nx, sz = (size_t)nx * ny;
  k_flood_init<<<4096, 256, 0, stream>>>(field, level, nx, ny, nz, exterior);
  // a round carries the flood along whole grid lines, so even a maze needs few rounds; the cap only bounds the loop
  const int max_rounds = 2 * (nx + ny + nz) + 16;
  int rc = 1;   // 1 = still flooding
  for (int round = 0; round < max_rounds; ++round) {
    hipError_t e = hipMemsetAsync(flag.p, 0, sizeof(unsigned), stream);
    if (e == hipSuccess) {
      k_flood_sweep<<<(ny * nz + 255) / 256, 256, 0, stream>>>(field, level, ny * nz, ny, sy, nz, sz, nx, sx, exterior, (unsigned*)flag.p);
      k_flood_sweep<<<(nx * nz + 255) / 256, 256, 0, stream>>>(field, level, nx * nz, nx, sx, nz, sz, ny, sy, exterior, (unsigned*)flag.p);
      k_flood_sweep<<<(nx * ny + 255) / 256, 256, 0, stream>>>(field, level, nx * ny, nx, sx, ny, sy, nz, sz, exterior, (unsigned*)flag.p);
      e = hipGetLastError();
    }
    unsigned changed = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&changed, flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { rc = disn_fail(ctx, DISN_E_HIP, "disn_df_flood: %s", hipGetErrorString(e)); break; }
    if (!changed) { rc = DISN_OK; break; }
  }
  disn_release(flag);
  if (rc == 1) rc = disn_fail(ctx, DISN_E_STATE, "disn_df_flood: no fixed point after %d rounds", max_rounds);
  return rc;
}
