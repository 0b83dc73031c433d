/* mc_oracle.c -- CPU restatement of the reference's isosurface step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (disn_amd/, the
 * C-ABI library) may call, link or import this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * What it restates: `computeMarchingCubes <in.dist> <out.obj> -i iso`, the
 * closed Vega-FEM tool the reference shells out to (reference:
 * demo/demo.py:355-369; binary isosurface/computeMarchingCubes, no source in
 * the tree).  The algorithm follows SURVEY.md Appendix B item by item (B.1
 * shift & nudge, B.2 voxel code, B.3 face/interior ambiguity tests, B.4 edge
 * vertices, B.5 centre vertex, B.6 ordering); each function cites its item.
 *
 * Pinning status: pinned by the reference's own fixture and known answers, not by
 * a golden mesh (the reference holds none: demo/result.obj is stripped, SURVEY.md s4;
 * prebuilt reference binaries are not executed by this build):
 *  - the reference's shipped data file isosurface/model_normalized.obj.n.dist
 *    (narrow-band grid, res 64) meshes to exactly the size the reference tool
 *    reports for it, 3800 v / 7604 t (tests/golden/ref_narrowband_res64.npz);
 *  - the known answers the survey recorded from the tool (SURVEY.md App. A.1:
 *    r=0.6 sphere on a res-8 grid -> 126 v / 248 t; s6: analytic 3-primitive field
 *    at res 64/128/256 -> 8978/17940, 35690/71380, 143683/287370).
 * All are checked in tests/test_mc_oracle.py.  Vertex positions and the face
 * ORDER follow SURVEY.md App. B.4-B.6 and are not covered by those counts.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: B.3/B.4 specify separately rounded f32 products.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/disn_mc_tables.h"

typedef struct {
  int64_t n_centre, n_edge, n_tri;
  double* verts;  /* (n_centre + n_edge) x 3 */
  int32_t* tris;  /* n_tri x 3, 0-based */
  int64_t* keys;  /* per vertex: signed key (B.4/B.5) */
  int64_t n_bad;  /* voxels whose ambiguity lookup was -1 (B.3) */
  int32_t* tri_layer; /* per triangle: k of the voxel that emitted it (slab tests) */
} McResult;

/* B.3 faceTest */
static int face_test(int f, const float* c) {
  const unsigned char* fv = DISN_MC_FACE[(f < 0 ? -f : f) - 1];
  float A = c[fv[0]], B = c[fv[1]], C = c[fv[2]], D = c[fv[3]];
  float ac = A * C, bd = B * D;
  return ((ac > bd) == (f > 0)) == (A > 0);
}

/* B.3 interiorTest */
static int interior_test(int s, const float* c) {
  float d10 = c[1] - c[0], d67 = c[6] - c[7], d54 = c[5] - c[4], d23 = c[2] - c[3];
  float a = d10 * d67 - d54 * d23;
  float b = ((c[0] * d67 - c[4] * d23) + d10 * c[7]) - d54 * c[3];
  double t = -(double)b / ((double)a + (double)a);
  if (t < 0.0 || t > 1.0) return s > 0;
  double At = (double)c[0] + (double)d10 * t;
  double Bt = (double)c[4] + (double)d54 * t;
  double Ct = (double)c[7] + (double)d67 * t;
  double Dt = (double)c[3] + (double)d23 * t;
  int test = (At >= 0) | ((Bt >= 0) << 1) | ((Ct >= 0) << 2) | ((Dt >= 0) << 3);
  switch (test) {
    case 5: return (s > 0) ^ (Bt * Dt <= At * Ct);
    case 10: return (s > 0) ^ (At * Ct < Bt * Dt);
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    default: return s > 0;
  }
}

/* B.2+B.3: sub-table id of a voxel, -1 = no surface, -2 = impossible ambiguity */
static int classify(const float* c) {
  int code = 0;
  for (int v = 0; v < 8; ++v) code |= (c[v] > 0.0f) << v;
  int cs = DISN_MC_CASE[code];
  if (cs == 0) return -1;
  const signed char* ft = &DISN_MC_FACE_TESTS[7 * code];
  int n = ft[0], amb = 0;
  for (int t = 1; t <= n; ++t) amb |= face_test(ft[t], c) << (t - 1);
  if (DISN_MC_INTERIOR_TEST[code]) amb |= interior_test(DISN_MC_INTERIOR_TEST[code], c) << n;
  int sub = DISN_MC_AMBIGUITY[64 * cs + amb];
  if (sub < 0) return -2;
  return DISN_MC_SUB_START[code] + sub;
}

static int tri_cmp(const void* a, const void* b) {
  const int32_t* x = (const int32_t*)a; const int32_t* y = (const int32_t*)b;
  for (int i = 0; i < 3; ++i) if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
  return 0;
}

void mc_oracle_free(McResult* r) {
  if (!r) return;
  free(r->verts); free(r->tris); free(r->keys); free(r->tri_layer); free(r);
}

/* val[k][j][i], x fastest (reference: demo/demo.py:342-353 .dist layout).
 * bbox = lo xyz, hi xyz (f64).  Returns NULL on allocation failure. */
McResult* mc_oracle_run(const float* val, int nx, int ny, int nz, const double* bbox, float iso) {
  const int64_t P = (int64_t)nx * ny * nz;
  const int64_t sxy = (int64_t)nx * ny;
  McResult* r = (McResult*)calloc(1, sizeof(McResult));
  float* c = (float*)malloc(sizeof(float) * P);
  int16_t* sub = (int16_t*)malloc(sizeof(int16_t) * P);     /* per voxel (indexed by lower grid point) */
  uint8_t* eflag = (uint8_t*)calloc(P, 1);                  /* 3 edge slots per grid point */
  int64_t* ebase = (int64_t*)malloc(sizeof(int64_t) * (P + 1));
  if (!r || !c || !sub || !eflag || !ebase) return NULL;
  double h[3] = {(bbox[3] - bbox[0]) / (nx - 1), (bbox[4] - bbox[1]) / (ny - 1), (bbox[5] - bbox[2]) / (nz - 1)};

  /* B.1 */
#pragma omp parallel for
  for (int64_t p = 0; p < P; ++p) {
    float d = val[p] - iso;
    if (d == 0.0f) d = FLT_EPSILON;
    c[p] = d;
  }
  /* B.2/B.3 per voxel */
  int64_t n_bad = 0;
#pragma omp parallel for reduction(+ : n_bad)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        int64_t p = k * sxy + (int64_t)j * nx + i;
        if (i == nx - 1 || j == ny - 1 || k == nz - 1) { sub[p] = -1; continue; }
        float cv[8];
        for (int v = 0; v < 8; ++v)
          cv[v] = c[p + DISN_MC_CORNER[v][0] + (int64_t)DISN_MC_CORNER[v][1] * nx + DISN_MC_CORNER[v][2] * sxy];
        int s = classify(cv);
        if (s == -2) { ++n_bad; s = -1; }
        sub[p] = (int16_t)s;
      }
  r->n_bad = n_bad;
  /* B.4: every crossed edge of every active voxel owns a vertex, keyed by its lower grid point + axis */
  for (int64_t p = 0; p < P; ++p) {   /* serial: neighbouring voxels OR into shared flags */
    if (sub[p] < 0) continue;
    for (int e = 0; e < 12; ++e) {
      int a = DISN_MC_EDGE[e][0], b = DISN_MC_EDGE[e][1], dir = DISN_MC_EDGE[e][2];
      int64_t pa = p + DISN_MC_CORNER[a][0] + (int64_t)DISN_MC_CORNER[a][1] * nx + DISN_MC_CORNER[a][2] * sxy;
      int64_t pb = p + DISN_MC_CORNER[b][0] + (int64_t)DISN_MC_CORNER[b][1] * nx + DISN_MC_CORNER[b][2] * sxy;
      float prod = c[pa] * c[pb];
      if (prod <= 0.0f) eflag[pa] |= (uint8_t)(1 << dir);
    }
  }
  int64_t n_edge = 0, n_centre = 0, n_tri = 0;
  for (int64_t p = 0; p < P; ++p) {
    ebase[p] = n_edge;
    n_edge += __builtin_popcount(eflag[p]);
    if (sub[p] >= 0) {
      const unsigned char* s = &DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub[p]]];
      n_tri += s[0] / 3;
      n_centre += s[1];
    }
  }
  ebase[P] = n_edge;
  r->n_edge = n_edge; r->n_centre = n_centre; r->n_tri = n_tri;
  int64_t nv = n_edge + n_centre;
  r->verts = (double*)malloc(sizeof(double) * 3 * (nv ? nv : 1));
  r->keys = (int64_t*)malloc(sizeof(int64_t) * (nv ? nv : 1));
  r->tris = (int32_t*)malloc(sizeof(int32_t) * 3 * (n_tri ? n_tri : 1));
  int32_t* tl = (int32_t*)malloc(sizeof(int32_t) * 4 * (n_tri ? n_tri : 1));   /* (a,b,c,layer) rows, sorted together */
  r->tri_layer = (int32_t*)malloc(sizeof(int32_t) * (n_tri ? n_tri : 1));
  if (!r->verts || !r->keys || !r->tris || !tl || !r->tri_layer) return NULL;

  /* B.4 positions; B.6: edge vertices follow the centre vertices, ascending key */
#pragma omp parallel for
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        int64_t p = k * sxy + (int64_t)j * nx + i;
        if (!eflag[p]) continue;
        int64_t vi = n_centre + ebase[p];
        const int64_t stride[3] = {1, nx, sxy};
        for (int dir = 0; dir < 3; ++dir) {
          if (!(eflag[p] >> dir & 1)) continue;
          float ca = c[p], cb = c[p + stride[dir]];
          double pos[3] = {bbox[0] + i * h[0], bbox[1] + j * h[1], bbox[2] + k * h[2]};
          float diff = ca - cb;
          pos[dir] += ((double)ca * h[dir]) / (double)diff;
          memcpy(&r->verts[3 * vi], pos, sizeof(pos));
          r->keys[vi] = 3 * p + dir;
          ++vi;
        }
      }
  /* B.5 centre vertices, B.6 order: descending k, then ascending j, i */
  {
    int64_t ci = 0;
    for (int k = nz - 2; k >= 0; --k)
      for (int j = 0; j < ny - 1; ++j)
        for (int i = 0; i < nx - 1; ++i) {
          int64_t p = k * sxy + (int64_t)j * nx + i;
          if (sub[p] < 0) continue;
          if (!DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub[p]] + 1]) continue;
          double acc[3] = {0, 0, 0};
          int cnt = 0;
          for (int e = 0; e < 12; ++e) {
            int a = DISN_MC_EDGE[e][0], dir = DISN_MC_EDGE[e][2];
            int64_t pa = p + DISN_MC_CORNER[a][0] + (int64_t)DISN_MC_CORNER[a][1] * nx + DISN_MC_CORNER[a][2] * sxy;
            if (!(eflag[pa] >> dir & 1)) continue;
            /* crossed edge of THIS voxel: its flag is set because this voxel is active */
            int64_t vi = n_centre + ebase[pa] + __builtin_popcount(eflag[pa] & ((1 << dir) - 1));
            for (int d = 0; d < 3; ++d) acc[d] += r->verts[3 * vi + d];
            ++cnt;
          }
          for (int d = 0; d < 3; ++d) r->verts[3 * ci + d] = acc[d] / cnt;
          r->keys[ci] = -(int64_t)k * sxy + (int64_t)j * nx + i - 1;
          ++ci;
        }
  }
  /* triangles: need each voxel's centre index -> second pass in the same order */
  {
    int64_t ci = 0, ti = 0;
    /* map voxel -> centre index through a scan in B.6 order */
    int64_t* cmap = (int64_t*)malloc(sizeof(int64_t) * P);
    if (!cmap) return NULL;
    for (int k = nz - 2; k >= 0; --k)
      for (int j = 0; j < ny - 1; ++j)
        for (int i = 0; i < nx - 1; ++i) {
          int64_t p = k * sxy + (int64_t)j * nx + i;
          cmap[p] = -1;
          if (sub[p] >= 0 && DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub[p]] + 1]) cmap[p] = ci++;
        }
    for (int64_t p = 0; p < P; ++p) {
      if (sub[p] < 0) continue;
      const unsigned char* s = &DISN_MC_TRI[DISN_MC_SUB_OFFSET[sub[p]]];
      int nidx = s[0];
      for (int t = 0; t < nidx; t += 3) {
        int32_t v[3];
        for (int q = 0; q < 3; ++q) {
          int e = s[2 + t + q];
          if (e == 12) { v[q] = (int32_t)cmap[p]; continue; }
          int a = DISN_MC_EDGE[e][0], dir = DISN_MC_EDGE[e][2];
          int64_t pa = p + DISN_MC_CORNER[a][0] + (int64_t)DISN_MC_CORNER[a][1] * nx + DISN_MC_CORNER[a][2] * sxy;
          v[q] = (int32_t)(n_centre + ebase[pa] + __builtin_popcount(eflag[pa] & ((1 << dir) - 1)));
        }
        /* B.6 rotate so the smallest index leads, orientation kept */
        int m = 0;
        if (v[1] < v[m]) m = 1;
        if (v[2] < v[m]) m = 2;
        tl[4 * ti + 0] = v[m];
        tl[4 * ti + 1] = v[(m + 1) % 3];
        tl[4 * ti + 2] = v[(m + 2) % 3];
        tl[4 * ti + 3] = (int32_t)(p / sxy);
        ++ti;
      }
    }
    free(cmap);
    qsort(tl, (size_t)n_tri, 4 * sizeof(int32_t), tri_cmp);   /* stable enough: ties are identical triples */
    for (int64_t t = 0; t < n_tri; ++t) {
      memcpy(&r->tris[3 * t], &tl[4 * t], 3 * sizeof(int32_t));
      r->tri_layer[t] = tl[4 * t + 3];
    }
    free(tl);
  }
  free(c); free(sub); free(eflag); free(ebase);
  return r;
}

/* flat accessors for ctypes */
int64_t mc_oracle_n_centre(const McResult* r) { return r->n_centre; }
int64_t mc_oracle_n_edge(const McResult* r) { return r->n_edge; }
int64_t mc_oracle_n_tri(const McResult* r) { return r->n_tri; }
int64_t mc_oracle_n_bad(const McResult* r) { return r->n_bad; }
void mc_oracle_copy_layers(const McResult* r, int32_t* layers) { memcpy(layers, r->tri_layer, sizeof(int32_t) * r->n_tri); }
void mc_oracle_copy(const McResult* r, double* verts, int32_t* tris, int64_t* keys) {
  int64_t nv = r->n_centre + r->n_edge;
  if (verts) memcpy(verts, r->verts, sizeof(double) * 3 * nv);
  if (tris) memcpy(tris, r->tris, sizeof(int32_t) * 3 * r->n_tri);
  if (keys) memcpy(keys, r->keys, sizeof(int64_t) * nv);
}
/* sub-table id per voxel (for case-index parity tests); out has nx*ny*nz int16, -1 = inactive */
void mc_oracle_classify(const float* val, int nx, int ny, int nz, float iso, int16_t* out) {
  const int64_t sxy = (int64_t)nx * ny;
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        int64_t p = k * sxy + (int64_t)j * nx + i;
        if (i == nx - 1 || j == ny - 1 || k == nz - 1) { out[p] = -1; continue; }
        float cv[8];
        for (int v = 0; v < 8; ++v) {
          float d = val[p + DISN_MC_CORNER[v][0] + (int64_t)DISN_MC_CORNER[v][1] * nx + DISN_MC_CORNER[v][2] * sxy] - iso;
          cv[v] = d == 0.0f ? FLT_EPSILON : d;
        }
        int s = classify(cv);
        out[p] = (int16_t)(s < 0 ? -1 : s);
      }
}
