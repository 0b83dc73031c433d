// hostio.cpp -- host-side file formats of the isosurface hop (include/disn.h):
// the .dist grid file the reference writes in to_binary (reference:
// demo/demo.py:342-353; reader preprocessing/create_point_sdf_grid.py:29-51)
// and the OBJ text the reference's tool emits.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>
#include <algorithm>
#include <atomic>
#include <thread>

#include "../../include/disn.h"

extern "C" int disn_write_dist(const char* path, int res, const double* bbox, const float* values) {
  if (!path || !bbox || !values || res < 1) return DISN_E_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return DISN_E_IO;
  int32_t hdr[3] = {-res, res, res};
  size_t n = (size_t)(res + 1) * (res + 1) * (res + 1);
  bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(bbox, sizeof(double), 6, f) == 6 &&
            fwrite(values, sizeof(float), n, f) == n;
  ok = (fclose(f) == 0) && ok;
  return ok ? DISN_OK : DISN_E_IO;
}

extern "C" int disn_read_dist_header(const char* path, int* res, double* bbox) {
  if (!path || !res || !bbox) return DISN_E_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return DISN_E_IO;
  int32_t hdr[3];
  bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && fread(bbox, sizeof(double), 6, f) == 6;
  fclose(f);
  if (!ok) return DISN_E_IO;
  // the first resolution is stored negated (marks the dense float32 format)
  int rx = hdr[0] < 0 ? -hdr[0] : hdr[0];
  if (rx < 1 || hdr[1] != rx || hdr[2] != rx) return DISN_E_IO;  // the path only produces cubic grids
  *res = rx;
  return DISN_OK;
}

extern "C" int disn_read_dist(const char* path, float* values, size_t n_values) {
  if (!path || !values) return DISN_E_ARG;
  FILE* f = fopen(path, "rb");
  if (!f) return DISN_E_IO;
  bool ok = fseek(f, 12 + 48, SEEK_SET) == 0 && fread(values, sizeof(float), n_values, f) == n_values;
  fclose(f);
  return ok ? DISN_OK : DISN_E_IO;
}

extern "C" int disn_read_ndist_header(const char* path, int* res, double* bbox, int64_t* n_cells) {
  if (!path || !res || !bbox || !n_cells) return DISN_E_ARG;
  int rc = disn_read_dist_header(path, res, bbox);
  if (rc != DISN_OK) return rc;
  FILE* f = fopen(path, "rb");
  if (!f) return DISN_E_IO;
  int32_t n = -1;
  bool ok = fseek(f, 60, SEEK_SET) == 0 && fread(&n, sizeof n, 1, f) == 1 && n >= 0;
  fclose(f);
  if (!ok) return DISN_E_IO;
  *n_cells = n;
  return DISN_OK;
}

extern "C" int disn_read_ndist(const char* path, float* values, size_t n_values) {
  if (!path || !values) return DISN_E_ARG;
  int res;
  double bbox[6];
  int64_t n;
  int rc = disn_read_ndist_header(path, &res, bbox, &n);
  if (rc != DISN_OK) return rc;
  const size_t n1 = (size_t)res + 1;
  if (n_values != n1 * n1 * n1) return DISN_E_ARG;
  // Cells that the file does not list lie outside the band.  The file carries no inside/outside flag of its own;
  // they get the sign of the side they are on, which the band itself tells: walking a grid line in +x from the box
  // wall (outside), the last listed cell passed says on which side of the surface we are.  Exterior cells read
  // +FLT_MAX, interior cells -FLT_MAX, so no surface appears between a band cell and a missing neighbour.
  const float unset = -0.0f;   // bit pattern 0x80000000 never occurs as a marker clash: listed values overwrite it
  std::vector<unsigned char> listed(n_values, 0);
  for (size_t q = 0; q < n_values; ++q) values[q] = unset;
  FILE* f = fopen(path, "rb");
  if (!f) return DISN_E_IO;
  bool ok = fseek(f, 64, SEEK_SET) == 0;
  struct Rec { int32_t i, j, k; float d; };
  std::vector<Rec> buf(4096);
  for (int64_t done = 0; ok && done < n;) {
    size_t want = (size_t)((n - done) < (int64_t)buf.size() ? (n - done) : (int64_t)buf.size());
    ok = fread(buf.data(), sizeof(Rec), want, f) == want;
    for (size_t q = 0; ok && q < want; ++q) {
      const Rec& r = buf[q];
      if (r.i < 0 || r.j < 0 || r.k < 0 || (size_t)r.i >= n1 || (size_t)r.j >= n1 || (size_t)r.k >= n1) { ok = false; break; }
      const size_t p = ((size_t)r.k * n1 + r.j) * n1 + r.i;
      values[p] = r.d;
      listed[p] = 1;
    }
    done += (int64_t)want;
  }
  fclose(f);
  if (!ok) return DISN_E_IO;
  for (size_t row = 0; row < n1 * n1; ++row) {
    float fill = 3.402823466e+38f;
    for (size_t i = 0; i < n1; ++i) {
      const size_t p = row * n1 + i;
      if (listed[p]) fill = values[p] < 0.0f ? -3.402823466e+38f : 3.402823466e+38f;
      else values[p] = fill;
    }
  }
  return DISN_OK;
}

// integer -> decimal, returns the end pointer (faces are most of an OBJ's lines; printf is the slow part)
static char* put_uint(char* p, unsigned v) {
  char tmp[12];
  int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}

extern "C" int disn_write_obj(const char* path, const double* verts, uint64_t n_verts, const int32_t* tris,
                              uint64_t n_tris) {
  if (!path || (n_verts && !verts) || (n_tris && !tris)) return DISN_E_ARG;
  FILE* f = fopen(path, "w");
  if (!f) return DISN_E_IO;
  fprintf(f, "# Generated by the TriMeshRef class\n# Number of vertices: %llu\n# Number of faces: %llu\n",
          (unsigned long long)n_verts, (unsigned long long)n_tris);
  // lines are formatted into per-chunk buffers by a few threads (text formatting, not the disk, bounds a
  // 6 MB OBJ) and written in order; the text is byte-identical to the serial loop
  const uint64_t CH = 1 << 15;
  const uint64_t nvc = (n_verts + CH - 1) / CH, ntc = (n_tris + CH - 1) / CH, nch = nvc + ntc;
  std::vector<std::vector<char>> out(nch);
  auto work = [&](uint64_t c) {
    std::vector<char>& b = out[c];
    if (c < nvc) {
      const uint64_t i0 = c * CH, i1 = std::min(n_verts, i0 + CH);
      b.resize((i1 - i0) * 80);
      char* p = b.data();
      for (uint64_t i = i0; i < i1; ++i)   // ostream default = %g, 6 significant digits
        p += snprintf(p, 80, "v %g %g %g\n", verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
      b.resize(p - b.data());
    } else {
      const uint64_t i0 = (c - nvc) * CH, i1 = std::min(n_tris, i0 + CH);
      b.resize((i1 - i0) * 40);
      char* p = b.data();
      for (uint64_t i = i0; i < i1; ++i) {
        *p++ = 'f';
        for (int q = 0; q < 3; ++q) {
          *p++ = ' ';
          p = put_uint(p, (unsigned)(tris[3 * i + q] + 1));
        }
        *p++ = '\n';
      }
      b.resize(p - b.data());
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
  if (nt > nch) nt = (unsigned)nch;
  if (nt <= 1) {
    for (uint64_t c = 0; c < nch; ++c) work(c);
  } else {
    std::atomic<uint64_t> next(0);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
      th.emplace_back([&] { for (uint64_t c; (c = next.fetch_add(1)) < nch;) work(c); });
    for (auto& t : th) t.join();
  }
  bool ok = true;
  for (uint64_t c = 0; c < nch && ok; ++c)
    ok = out[c].empty() || fwrite(out[c].data(), 1, out[c].size(), f) == out[c].size();
  ok = ok && !ferror(f);
  ok = (fclose(f) == 0) && ok;
  return ok ? DISN_OK : DISN_E_IO;
}
