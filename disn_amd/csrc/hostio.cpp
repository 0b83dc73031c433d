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

// narrow-band writer (the `-n -w band` output of the distance-field tool): the dense header, int32 N, then the N cells
// with |value| < band as (int32 i, j, k, float32 d), here in x-fastest order (the tool's own files list them in no
// particular order; readers must not depend on it)
extern "C" int disn_write_ndist(const char* path, int res, const double* bbox, const float* values, float band) {
  if (!path || !bbox || !values || res < 1 || !(band > 0.0f)) return DISN_E_ARG;
  const int n = res + 1;
  size_t count = 0;
  const size_t total = (size_t)n * n * n;
  for (size_t p = 0; p < total; ++p) count += fabsf(values[p]) < band;
  if (count > 0x7fffffffull) return DISN_E_ARG;
  FILE* f = fopen(path, "wb");
  if (!f) return DISN_E_IO;
  const int32_t hdr[3] = {-res, res, res}, cnt = (int32_t)count;
  bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(bbox, sizeof(double), 6, f) == 6 && fwrite(&cnt, sizeof cnt, 1, f) == 1;
  struct Cell { int32_t i, j, k; float d; };
  std::vector<Cell> buf;
  buf.reserve(1 << 16);
  for (int k = 0; ok && k < n; ++k)
    for (int j = 0; j < n; ++j) {
      const float* row = values + ((size_t)k * n + j) * n;
      for (int i = 0; i < n; ++i)
        if (fabsf(row[i]) < band) buf.push_back(Cell{i, j, k, row[i]});
      if (buf.size() >= (1 << 16) - (size_t)n || (k == n - 1 && j == n - 1)) {
        ok = ok && (buf.empty() || fwrite(buf.data(), sizeof(Cell), buf.size(), f) == buf.size());
        buf.clear();
      }
    }
  ok = (fclose(f) == 0) && ok;
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
  // they get the sign of the side they are on, which the band itself tells (it is assumed closed and iso ~ 0):
  // along a grid line a missing cell has the sign of the last listed cell passed, and the cells in front of the
  // line's first listed cell have that cell's sign (so a line that starts inside the surface at the box wall is
  // interior up to its first band cell).  Lines without any listed cell are resolved the same way along y, then
  // along z, from the cells resolved so far.  Exterior cells read +FLT_MAX, interior cells -FLT_MAX, so no
  // surface appears between a band cell and a missing neighbour.
  const float FMAX = 3.402823466e+38f;
  std::vector<uint64_t> known((n_values + 63) / 64, 0);   // bit p: values[p] is listed or already resolved
  auto is_known = [&](size_t p) { return (known[p >> 6] >> (p & 63)) & 1u; };
  auto set_known = [&](size_t p) { known[p >> 6] |= (uint64_t)1 << (p & 63); };
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
      set_known(p);
    }
    done += (int64_t)want;
  }
  fclose(f);
  if (!ok) return DISN_E_IO;
  // one sweep per axis: lines of `len` cells, cell t of line l at base(l) + t * stride
  auto sweep = [&](size_t stride, size_t nlines, auto base) {
    for (size_t l = 0; l < nlines; ++l) {
      const size_t b0 = base(l);
      size_t first = n1;
      for (size_t t = 0; t < n1; ++t)
        if (is_known(b0 + t * stride)) { first = t; break; }
      if (first == n1) continue;   // nothing to go by on this line (yet)
      float fill = values[b0 + first * stride] < 0.0f ? -FMAX : FMAX;
      for (size_t t = 0; t < n1; ++t) {
        const size_t p = b0 + t * stride;
        if (is_known(p)) fill = values[p] < 0.0f ? -FMAX : FMAX;
        else { values[p] = fill; set_known(p); }
      }
    }
  };
  sweep(1, n1 * n1, [&](size_t l) { return l * n1; });                                          // along x
  sweep(n1, n1 * n1, [&](size_t l) { return (l / n1) * n1 * n1 + (l % n1); });                  // along y
  sweep(n1 * n1, n1 * n1, [&](size_t l) { return l; });                                         // along z
  for (size_t p = 0; p < n_values; ++p)
    if (!is_known(p)) values[p] = FMAX;   // a file without any cell
  return DISN_OK;
}

// CRC-32C (Castagnoli, reflected polynomial 0x82f63b78), slicing by 8: the checksum of the tensor-bundle
// checkpoint format (table block trailers and BundleEntryProto.crc32c), used by disn_amd/tf_bundle.py.
static uint32_t g_crc_tab[8][256];
static std::atomic<int> g_crc_ready(0);
static void crc_init() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82f63b78u & (0u - (c & 1u)));
    g_crc_tab[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 255u];
  g_crc_ready.store(1, std::memory_order_release);
}
extern "C" uint32_t disn_crc32c(uint32_t crc, const void* data, size_t n) {
  if (!g_crc_ready.load(std::memory_order_acquire)) crc_init();   // idempotent: racing initialisers write the same table
  const unsigned char* p = (const unsigned char*)data;
  uint32_t c = ~crc;
  while (n && ((uintptr_t)p & 7)) { c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 255u]; --n; }
  while (n >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w ^= c;
    c = g_crc_tab[7][w & 255u] ^ g_crc_tab[6][(w >> 8) & 255u] ^ g_crc_tab[5][(w >> 16) & 255u] ^ g_crc_tab[4][(w >> 24) & 255u] ^
        g_crc_tab[3][(w >> 32) & 255u] ^ g_crc_tab[2][(w >> 40) & 255u] ^ g_crc_tab[1][(w >> 48) & 255u] ^ g_crc_tab[0][w >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ g_crc_tab[0][(c ^ *p++) & 255u];
  return ~c;
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
