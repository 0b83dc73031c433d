// f16_denorm_probe.hip -- does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  (experiment, not product)
// A = 2^-20 (fp16 subnormal, bits 0x0010) everywhere, B = 1024: every output element is 16 * 2^-10 = 2^-6 if
// subnormals take part, 0 if the matrix pipe flushes them.  Also with the MODE register's fp16 denorm bits cleared.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(4))) int intx4;
__global__ void k(float* out, int flush_mode) {
  if (flush_mode) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");   // FP_DENORM for f64/f16: flush in+out
  intx4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = 0x00100010; b[i] = 0x64006400; }
  floatx16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7" : "+v"(acc) : "v"(a), "v"(b));
  if (threadIdx.x == 0) out[flush_mode] = acc[0];
}
int main() {
  float* d; float h[2] = {-1, -1};
  if (hipMalloc(&d, 8) != hipSuccess) return 1;
  k<<<1, 64>>>(d, 0); k<<<1, 64>>>(d, 1);
  if (hipMemcpy(h, d, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  printf("f16 MFMA with subnormal A (2^-20) x 1024, K=16: default mode %g, fp16-denorm-flush mode %g (kept: 0.015625, flushed: 0)\n", h[0], h[1]);
  return 0;
}
