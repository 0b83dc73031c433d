// gather_cost.hip -- texture-address-path cost of the decoder's 4-corner gather patterns (experiment).
// One workgroup of 4 waves; every wave issues REP x 16 global_load_dwordx4 with a given lane->address
// pattern over an L2-resident 38 MB map ([18769][512] floats) and reports cycles per load instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float floatx4;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(long long* out, const float* pmap, int rep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pt = lane & 31, h = lane >> 5;
  long long row;   // pixel row of this lane
  int col;         // float column offset inside the 512-float row
  if (MODE == 0) { row = 1000 + wave; col = lane * 4; }                                  // fully coalesced 1 KiB
  else if (MODE == 1) { row = 2000 + wave * 40 + pt; col = 4 * h; }                        // every point its own row
  else if (MODE == 2) { row = 2000 + wave * 40 + pt / 3; col = 4 * h; }                    // 3 points share a pixel
  else if (MODE == 3) { row = 2000 + wave * 40 + (lane >> 2); col = 4 * (lane & 3); }      // quads read 64 B contiguous
  else if (MODE == 4) { row = 2000 + wave * 40 + (lane >> 1); col = 4 * (lane & 1); }      // lane pairs read 32 B contiguous
  else if (MODE == 5) { row = 2000 + wave * 40 + (lane >> 3); col = 4 * (lane & 7); }      // 8 lanes read 128 B contiguous
  else { row = 2000 + wave * 40 + pt; col = 4 * h; }
  const float* base = pmap + row * 512 + col;
  floatx4 acc = {0, 0, 0, 0};
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < rep; ++r) {
    floatx4 t[16];
    const float* b = base + (r & 15) * 32;   // walk the 16 feature blocks of the row
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int q = 0; q < 4; ++q) t[4 * c + q] = __builtin_nontemporal_load((const floatx4*)(b + (c * 137) * 512 + 8 * q));
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += t[i];
  }
  long long t1 = clock64();
  __syncthreads();
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.0f) out[1] = 1;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, long long* dout, const float* dmap) {
  const int rep = 4096;
  long long hcyc = 0;
  for (int it = 0; it < 2; ++it) {
    k<MODE><<<dim3(1), dim3(256)>>>(dout, dmap, rep);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("mode %d: %s\n", MODE, hipGetErrorString(e)); return; }
    hipMemcpy(&hcyc, dout, 8, hipMemcpyDeviceToHost);
  }
  printf("mode %d  %-44s %7.1f cycles per load instruction per CU (4 waves x %d x 16 loads)\n", MODE, name,
         (double)hcyc / (rep * 16.0 * 4.0), rep);
}

int main() {
  long long* dout; float* dmap;
  size_t n = (size_t)18769 * 512;
  hipMalloc(&dout, 64); hipMalloc(&dmap, n * 4);
  hipMemset(dmap, 0, n * 4);
  run<0>("coalesced 1 KiB per wave", dout, dmap);
  run<1>("32 rows, lanes pt/pt+32 share 32 B", dout, dmap);
  run<2>("11 rows (3 points per pixel)", dout, dmap);
  run<3>("quads read 64 B contiguous", dout, dmap);
  run<4>("lane pairs read 32 B contiguous", dout, dmap);
  run<5>("8 lanes read 128 B contiguous", dout, dmap);
  return 0;
}
