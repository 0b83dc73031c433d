// mfma_probe.hip -- checks, with exact integer data, the lane maps of
// v_mfma_f32_32x32x16_bf16 that disn_amd/csrc/sdf.hip relies on:
//   A: lane(r,h) elem j = A[row r][k = 8h+j];  B: lane(r,h) elem j = B[k = 8h+j][col r]
//   D: lane(r,h) reg g  = D[row (g&3)+8(g>>2)+4h][col r]
// and the accumulator-as-B-operand chain: regs 8s..8s+7 of D are k-step s of the
// next product with physical k-slot (h,j) <-> row 16s+8(j>>2)+4h+(j&3).
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

__global__ void probe(const float* A, const float* B, const float* W2, float* D, float* E) {
  // A[32][16], B[16][32] row-major; W2[32 out][32 k] row-major; D[32][32]; E = W2 * D  [32][32]
  int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 16 + 8 * h + j]; b[j] = (__bf16)B[(8 * h + j) * 32 + r]; }
  floatx16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];
  floatx16 e = {0};
  for (int s = 0; s < 2; ++s) {
    bf16x8 x, w;
    for (int j = 0; j < 8; ++j) {
      x[j] = (__bf16)acc[8 * s + j];
      int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      w[j] = (__bf16)W2[r * 32 + k];
    }
    e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, e, 0, 0, 0);
  }
  for (int g = 0; g < 16; ++g) E[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = e[g];
}

int main() {
  std::vector<float> A(32 * 16), B(16 * 32), W2(32 * 32), D(32 * 32), E(32 * 32), Dr(32 * 32, 0), Er(32 * 32, 0);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 3 + k * 5) % 7 - 3);
  for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 2 + j * 7) % 5 - 2);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 32; ++k) W2[i * 32 + k] = (float)((i * 5 + k * 3) % 3 - 1);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) Dr[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 32; ++k) Er[i * 32 + j] += W2[i * 32 + k] * Dr[k * 32 + j];
  float *dA, *dB, *dW, *dD, *dE;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dW, W2.size() * 4);
  hipMalloc(&dD, D.size() * 4); hipMalloc(&dE, E.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W2.data(), W2.size() * 4, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dW, dD, dE);
  if (hipDeviceSynchronize() != hipSuccess) { printf("MFMA_PROBE launch FAILED\n"); return 2; }
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(E.data(), dE, E.size() * 4, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int i = 0; i < 1024; ++i) { bad1 += D[i] != Dr[i]; bad2 += E[i] != Er[i]; }
  printf("MFMA_PROBE first=%s (%d bad) chained=%s (%d bad)\n", bad1 ? "FAIL" : "PASS", bad1, bad2 ? "FAIL" : "PASS", bad2);
  return (bad1 || bad2) ? 1 : 0;
}
