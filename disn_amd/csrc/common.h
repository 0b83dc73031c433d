// common.h -- context, error plumbing and small device helpers shared by the
// HIP translation units of libdisn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <string>

#include "../../include/disn.h"

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct SdfState;  // sdf.hip
struct McState;   // mc.hip
struct VggState;  // vgg.hip

struct DisnCtx {
  int device = 0;
  std::string err;
  SdfState* sdf = nullptr;
  McState* mc = nullptr;
  VggState* vgg = nullptr;
  void* static_scratch = nullptr;   // this device's copy of mc.hip's static scratch array (disn_mc_static_init)
};

inline int disn_fail(DisnCtx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define DISN_HIP(ctx, expr)                                                              \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess)                                                                \
      return disn_fail((ctx), DISN_E_HIP, "%s failed: %s (%s:%d)", #expr,                \
                       hipGetErrorString(e_), __FILE__, __LINE__);                       \
  } while (0)

#define DISN_TRY(expr)            \
  do {                            \
    int r_ = (expr);              \
    if (r_ != DISN_OK) return r_; \
  } while (0)

// grow-only device scratch
inline int disn_ensure(DisnCtx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return DISN_OK;
  if (b.p) DISN_HIP(ctx, hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) return disn_fail(ctx, DISN_E_NOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  b.cap = want;
  return DISN_OK;
}
inline void disn_release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

void disn_sdf_state_free(DisnCtx*);
void disn_mc_state_free(DisnCtx*);
void disn_vgg_state_free(DisnCtx*);
int disn_mc_static_init(DisnCtx*);   // mc.hip: looks up the device address of the static scratch array

// ---- device helpers (wave = 64 lanes) ------------------------------------
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    unsigned t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
