// cvt_probe.hip -- what the two-step packed fp8 conversions of the mixed-precision decoder leave in a register (experiment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) short shortx2;
__device__ __forceinline__ int any_int() { int r; asm("" : "=v"(r)); return r; }
__global__ void k(const float* x, int* out, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  float a = x[0], b = x[1], c = x[2], d = x[3];
  int x8 = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, any_int(), false);
  x8 = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, x8, true);
  shortx2 o = __builtin_bit_cast(shortx2, any_int());
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, b, 0.25f, false);
  o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, c, d, 0.25f, true);
  out[0] = x8;
  out[1] = __builtin_bit_cast(int, o);
  float m0, m1;
  asm volatile("v_max_f32 %0, 0, %1" : "=v"(m0) : "v"(x[4]));
  asm volatile("v_max_f32 %0, 0, %1" : "=v"(m1) : "v"(x[5]));
  out[2] = __builtin_amdgcn_cvt_pk_fp8_f32(m0, m1, 0, false);
  out[3] = __float_as_int(m0);
  out[4] = __float_as_int(m1);
}
int main() {
  float hx[6] = {1.0f, 2.0f, 3.0f, 4.0f, -0.0f, -1.5f};
  float* dx; int* dout;
  hipMalloc(&dx, sizeof hx); hipMalloc(&dout, 64);
  hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
  for (int ov = 0; ov < 2; ++ov) {
    k<<<1, 64>>>(dx, dout, ov);
    int h[5];
    hipMemcpy(h, dout, sizeof h, hipMemcpyDeviceToHost);
    printf("ovfl %d: fp8(1,2 | 3,4) = %08x (want 48444038)   scaled x4 = %08x (want 58545048)   fp8(relu(-0), relu(-1.5)) = %08x, relu bits %08x %08x\n",
           ov, h[0], h[1], h[2], h[3], h[4]);
  }
  return 0;
}
