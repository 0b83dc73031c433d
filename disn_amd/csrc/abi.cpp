// abi.cpp -- context lifetime of libdisn_hip.so (see include/disn.h).
#include "common.h"

extern "C" int disn_create(int device_id, DisnCtx** out) {
  if (!out) return DISN_E_ARG;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return DISN_E_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return DISN_E_HIP;
  DisnCtx* c = new DisnCtx();
  c->device = device_id;
  if (disn_mc_static_init(c) != DISN_OK) {   // per-device address of the allocation-free entry point's scratch
    fprintf(stderr, "disn_create: %s\n", c->err.c_str());
    delete c;
    return DISN_E_HIP;
  }
  *out = c;
  return DISN_OK;
}

extern "C" void disn_destroy(DisnCtx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  disn_sdf_state_free(ctx);
  disn_mc_state_free(ctx);
  disn_vgg_state_free(ctx);
  delete ctx;
}

extern "C" const char* disn_last_error(const DisnCtx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" const char* disn_version(void) { return "disn_hip 0.1 (gfx950)"; }
