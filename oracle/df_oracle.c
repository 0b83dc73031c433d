/* df_oracle.c -- CPU restatement of the distance-field computation of the reference's ground-truth tool
 * `computeDistanceField <obj> res res res [-s] [-m 0]` (closed Vega-FEM binary; call site
 * /root/reference/preprocessing/create_point_sdf_grid.py:200-210, output layout :29-51).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker of disn_df_eval, never part of the product.
 * PARITY UNPINNED against the tool: it is a closed binary that is not executed here and the reference holds neither
 * an input mesh nor a full output of it; what is restated is the DEFINITION the tool's help text gives (exact distance
 * to the nearest triangle; sign for manifold meshes) with the standard published sign rule (angle-weighted
 * pseudonormals, Baerentzen & Aanaes 2005).  Deliberately a different formulation from the GPU's region walk:
 * brute force over all triangles; per triangle the point is projected onto the plane and, when the projection falls
 * outside, the three edge segments are measured.
 *
 * gcc -O2 -fopenmp -ffp-contract=off */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void sub3(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* closest point on segment ab; *where = 0 at a, 1 at b, 2 strictly inside */
static double seg_dist2(const double* p, const double* a, const double* b, double* q, int* where) {
  double ab[3], ap[3], d[3];
  sub3(b, a, ab);
  sub3(p, a, ap);
  double t = dot3(ap, ab) / dot3(ab, ab);
  if (!(t > 0.0)) { t = 0.0; *where = 0; }
  else if (!(t < 1.0)) { t = 1.0; *where = 1; }
  else *where = 2;
  for (int i = 0; i < 3; ++i) q[i] = a[i] + t * ab[i];
  sub3(p, q, d);
  return dot3(d, d);
}

/* region: 0 face, 1 2 3 vertex a b c, 4 edge ab, 5 edge ac, 6 edge bc */
static double tri_dist2(const double* p, const double* a, const double* b, const double* c, double* q, int* region) {
  double ab[3], ac[3], bc[3], n[3], ap[3], bp[3], cp[3], t[3];
  sub3(b, a, ab); sub3(c, a, ac); sub3(c, b, bc);
  cross3(ab, ac, n);
  sub3(p, a, ap); sub3(p, b, bp); sub3(p, c, cp);
  /* inside the three edge planes? */
  cross3(ab, ap, t); const double s0 = dot3(t, n);
  cross3(bc, bp, t); const double s1 = dot3(t, n);
  double ca[3]; sub3(a, c, ca);
  cross3(ca, cp, t); const double s2 = dot3(t, n);
  if (s0 >= 0.0 && s1 >= 0.0 && s2 >= 0.0) {
    const double nn = dot3(n, n), h = dot3(ap, n) / nn;
    for (int i = 0; i < 3; ++i) q[i] = p[i] - h * n[i];
    *region = 0;
    return h * h * nn;
  }
  double best = INFINITY, qq[3];
  int w;
  const double* ends[3][2] = {{a, b}, {a, c}, {b, c}};
  const int vid[3][2] = {{1, 2}, {1, 3}, {2, 3}};
  for (int e = 0; e < 3; ++e) {
    const double d2 = seg_dist2(p, ends[e][0], ends[e][1], qq, &w);
    if (d2 < best) {
      best = d2;
      q[0] = qq[0]; q[1] = qq[1]; q[2] = qq[2];
      *region = w == 2 ? 4 + e : vid[e][w];
    }
  }
  return best;
}

/* geom [T][9]; normals [T][7][3] or NULL (unsigned); grid point (i,j,k) = lo + (i,j,k) h, x fastest; out float[nz][ny][nx] */
void df_oracle_eval(const double* geom, const double* normals, int64_t n_tris, int nx, int ny, int nz, const double* bbox,
                    float* out) {
  const double h[3] = {(bbox[3] - bbox[0]) / (nx - 1), (bbox[4] - bbox[1]) / (ny - 1), (bbox[5] - bbox[2]) / (nz - 1)};
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
  for (int k = 0; k < nz; ++k)
    for (int j = 0; j < ny; ++j)
      for (int i = 0; i < nx; ++i) {
        const double p[3] = {bbox[0] + i * h[0], bbox[1] + j * h[1], bbox[2] + k * h[2]};
        double best = INFINITY, bq[3] = {0, 0, 0};
        int64_t bt = -1;
        int breg = 0;
        for (int64_t t = 0; t < n_tris; ++t) {
          const double* g = geom + 9 * t;
          double q[3];
          int reg = 0;
          const double d2 = tri_dist2(p, g, g + 3, g + 6, q, &reg);
          if (d2 < best) { best = d2; bt = t; breg = reg; bq[0] = q[0]; bq[1] = q[1]; bq[2] = q[2]; }
        }
        double d = sqrt(best);
        if (normals && bt >= 0) {
          const double* n = normals + 21 * bt + 3 * breg;
          double v[3];
          sub3(p, bq, v);
          if (dot3(v, n) < 0.0) d = -d;
        }
        out[((size_t)k * ny + j) * nx + i] = (float)d;
      }
}
