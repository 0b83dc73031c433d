// group_cost.hip -- issue cost of one "group" of the mixed-precision decoder (4 x f16 K=16 MFMA + 2 x fp8 K=64 MFMA)
// with its fragment reads, LDS-DMA refill pieces and conversion VALU work placed in different shadows
// (experiment, not product).  One workgroup of 4 waves per CU, every CU busy; reports cycles per group (ideal 256).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(8))) int intx8;
typedef __attribute__((ext_vector_type(4))) int intx4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define MF(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define ME(acc, a, b, sa, sb) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define DMA1(off) asm volatile("global_load_lds_dword %0, %1 offset:" #off ::"v"(voff), "s"(src) : "memory")
#define DMA4(off) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #off ::"v"(voff4), "s"(src) : "memory")
#define VALU7() do { \
  asm volatile("v_max_f32 %0, 0, %1" : "=v"(f0) : "v"(f1)); asm volatile("v_max_f32 %0, 0, %1" : "=v"(f2) : "v"(f3)); \
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(i0) : "v"(f0), "v"(f2)); \
  asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(f1) : "v"(i0), "v"(f0)); \
  asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(f3) : "v"(i0), "v"(f2)); \
  asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "=v"(i1) : "v"(f0), "v"(f2)); \
  asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(i2) : "v"(f1), "v"(f3), "s"(half)); } while (0)

struct Res { long long cyc, wall; };

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(Res* out, const uint4* src_, int rep) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const char* src = (const char*)src_;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float half = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x3f000000 + (rep & 0)));
  for (int i = threadIdx.x; i < 4096; i += 256) ((uint4*)lds)[i] = src_[i];   // 64 KiB of fragments
  __syncthreads();
  const uint4* lp = (const uint4*)lds + lane;
  intx4 f[4], bh[4];
  intx8 e[2], b8[2];
  for (int i = 0; i < 4; ++i) { f[i] = __builtin_bit_cast(intx4, lp[i * 64]); bh[i] = __builtin_bit_cast(intx4, lp[(i + 8) * 64]); }
  for (int i = 0; i < 2; ++i)
    for (int q = 0; q < 8; ++q) { e[i][q] = (int)lp[(12 + i) * 64].x & 0x77777777; b8[i][q] = (int)lp[(14 + i) * 64].y & 0x77777777; }
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 4; ++q) { f[i][q] = (f[i][q] & 0x3fff3fff) | 0x20002000; bh[i][q] = (bh[i][q] & 0x3fff3fff) | 0x20002000; }
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.001f * (i + q);
  int sa = 118, sb = 120;
  float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f; int i0 = 0, i1 = 0, i2 = 0;
  asm volatile("" : "+v"(sa), "+v"(sb), "+v"(f1), "+v"(f3));
  const unsigned voff = lane * 4, voff4 = lane * 16;
  const unsigned ldsbase = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
  asm volatile("s_mov_b32 m0, %0" ::"s"(ldsbase + 65536 + wave * 8192));   // DMA lands in an area nobody reads
  src += wave * 8192;
  const uint4* rd = lp;   // fragment reads walk over the first 32 KiB
  long long w0 = wall_clock64(), t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      floatx16& c = acc[g];
      constexpr bool RD = MODE >= 1;
      const uint4* q = rd + g * 8 * 64;
#define READ_F(i) do { if (RD) { f[i] = __builtin_bit_cast(intx4, q[(i) * 64]); FENCE(); } } while (0)
#define READ_E(i) do { if (RD) { uint4 lo = q[(4 + 2 * (i)) * 64], hi = q[(5 + 2 * (i)) * 64]; e[i][0] = lo.x; e[i][1] = lo.y; e[i][2] = lo.z; e[i][3] = lo.w; e[i][4] = hi.x; e[i][5] = hi.y; e[i][6] = hi.z; e[i][7] = hi.w; FENCE(); } } while (0)
      MF(c, f[0], bh[0]); FENCE(); READ_F(0); if (MODE == 2 || MODE == 6) DMA1(0); FENCE();
      MF(c, f[1], bh[1]); FENCE(); READ_F(1); if (MODE == 2 || MODE == 6) DMA1(256); FENCE();
      MF(c, f[2], bh[2]); FENCE(); READ_F(2); if (MODE == 2 || MODE == 6) DMA1(512); FENCE();
      MF(c, f[3], bh[3]); FENCE(); READ_F(3); if (MODE == 2 || MODE == 6) DMA1(768); FENCE();
      ME(c, e[0], b8[0], sa, sb); FENCE(); READ_E(0);
      if (MODE == 2 || MODE == 6) { DMA1(1024); DMA1(1280); }
      if (MODE == 3) { DMA1(0); DMA1(256); DMA1(512); DMA1(768); }
      if (MODE == 4 || MODE == 5 || MODE == 7) DMA4(0);
      if (MODE == 7) VALU7();
      FENCE();
      ME(c, e[1], b8[1], sb, sa); FENCE(); READ_E(1);
      if (MODE == 2 || MODE == 6) { DMA1(1536); DMA1(1792); }
      if (MODE == 3) { DMA1(1024); DMA1(1280); DMA1(1536); DMA1(1792); }
      if (MODE == 4 || MODE == 5 || MODE == 7) DMA4(1024);
      if (MODE == 5 || MODE == 6) VALU7();
      FENCE();
    }
  }
  long long t1 = clock64(), w1 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = f0 + f1 + f2 + f3 + (float)(i0 + i1 + i2);
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
  if (s == 12345.678f) out[0].cyc = 0;
  if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].wall = w1 - w0; }
}

template <int MODE>
static void run(const char* name, const uint4* dsrc, Res* dres) {
  const int rep = 2000, blocks = 256;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 32768));
  k<MODE><<<blocks, 256, 65536 + 32768>>>(dres, dsrc, 100);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 256, 65536 + 32768>>>(dres, dsrc, rep);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  std::vector<Res> r(blocks);
  CK(hipMemcpy(r.data(), dres, blocks * sizeof(Res), hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0; for (auto& x : r) { cyc += x.cyc; wall += x.wall; } cyc /= blocks; wall /= blocks;
  printf("%-66s %6.1f cyc/group  clock %.2f GHz  %.3f ms\n", name, cyc / (rep * 4.0), cyc / (wall * 10.0), ms);
}

int main() {
  std::vector<unsigned> h(65536 / 4 * 2);
  srand(5); for (auto& v : h) v = ((unsigned)rand() << 16) ^ rand();
  uint4* dsrc; Res* dres;
  CK(hipMalloc(&dsrc, h.size() * 4)); CK(hipMalloc(&dres, 256 * sizeof(Res)));
  CK(hipMemcpy(dsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0>("0: 4 f16 + 2 fp8 MFMAs only", dsrc, dres);
  run<1>("1: + 8 ds_read_b128 (1 per f16 shadow, 2 per fp8 shadow)", dsrc, dres);
  run<2>("2: 1 + 8 dword LDS-DMA pieces (1 per f16 shadow, 2 per fp8 shadow)", dsrc, dres);
  run<3>("3: 1 + 8 dword LDS-DMA pieces, all in the fp8 shadows (4 + 4)", dsrc, dres);
  run<4>("4: 1 + 2 dwordx4 LDS-DMA pieces (1 per fp8 shadow)", dsrc, dres);
  run<5>("5: 4 + 7 conversion VALU ops in the last fp8 shadow", dsrc, dres);
  run<6>("6: 2 + 7 conversion VALU ops in the last fp8 shadow (current kernel)", dsrc, dres);
  run<7>("7: 4 with the 7 VALU ops in the first fp8 shadow", dsrc, dres);
  return 0;
}
