// df_point_triangle.h -- closest point of a triangle to a point, float64, one fixed sequence of separately rounded
// operations (compile with -ffp-contract=off).  Used by the GPU distance-field kernels (csrc/df.hip) only; the CPU
// checker under tests/ measures the same distances with a different formulation.
//
// Algorithm: the Voronoi-region walk of Ericson, "Real-Time Collision Detection" (2005) s5.1.5.  region: 0 = face
// interior, 1 2 3 = vertex a b c, 4 = edge ab, 5 = edge ac, 6 = edge bc.  Returns the squared distance; q = the
// closest point.
#pragma once

#ifndef DF_QUAL
#define DF_QUAL static inline
#endif

DF_QUAL double df_dot(const double* x, const double* y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; }

DF_QUAL double df_point_triangle(const double* p, const double* a, const double* b, const double* c, int* region, double* q) {
  double ab[3], ac[3], ap[3], bp[3], cp[3], d[3];
  for (int t = 0; t < 3; ++t) { ab[t] = b[t] - a[t]; ac[t] = c[t] - a[t]; ap[t] = p[t] - a[t]; }
  const double d1 = df_dot(ab, ap), d2 = df_dot(ac, ap);
  if (d1 <= 0.0 && d2 <= 0.0) {
    *region = 1;
    for (int t = 0; t < 3; ++t) q[t] = a[t];
    return df_dot(ap, ap);
  }
  for (int t = 0; t < 3; ++t) bp[t] = p[t] - b[t];
  const double d3 = df_dot(ab, bp), d4 = df_dot(ac, bp);
  if (d3 >= 0.0 && d4 <= d3) {
    *region = 2;
    for (int t = 0; t < 3; ++t) q[t] = b[t];
    return df_dot(bp, bp);
  }
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
    const double v = d1 / (d1 - d3);
    *region = 4;
    for (int t = 0; t < 3; ++t) { q[t] = a[t] + v * ab[t]; d[t] = p[t] - q[t]; }
    return df_dot(d, d);
  }
  for (int t = 0; t < 3; ++t) cp[t] = p[t] - c[t];
  const double d5 = df_dot(ab, cp), d6 = df_dot(ac, cp);
  if (d6 >= 0.0 && d5 <= d6) {
    *region = 3;
    for (int t = 0; t < 3; ++t) q[t] = c[t];
    return df_dot(cp, cp);
  }
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
    const double w = d2 / (d2 - d6);
    *region = 5;
    for (int t = 0; t < 3; ++t) { q[t] = a[t] + w * ac[t]; d[t] = p[t] - q[t]; }
    return df_dot(d, d);
  }
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
    const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    *region = 6;
    for (int t = 0; t < 3; ++t) { q[t] = b[t] + w * (c[t] - b[t]); d[t] = p[t] - q[t]; }
    return df_dot(d, d);
  }
  const double denom = 1.0 / (va + vb + vc);
  const double v = vb * denom, w = vc * denom;
  *region = 0;
  for (int t = 0; t < 3; ++t) { q[t] = a[t] + ab[t] * v + ac[t] * w; d[t] = p[t] - q[t]; }
  return df_dot(d, d);
}
