// shape_cost.hip -- the decoder's "group" (f16 product + two block-scaled fp8 residual products) in the two MFMA shapes,
// equal MACs per wave (65536 per iteration step), random operands, every CU busy, operands optionally re-read from LDS:
//   A: 4 x v_mfma_f32_32x32x16_f16  + 2 x v_mfma_scale_f32_32x32x64_f8f6f4   (32 features x 64 k x 32 points)
//   B: 8 x v_mfma_f32_16x16x32_f16  + 4 x v_mfma_scale_f32_16x16x128_f8f6f4  (16 features x 128 k x 2 x 16 points)
// Reports cycles per step (ideal 256), the clock the launch settles at and the wall time: at the power limit the wall
// time is what counts (experiment, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(8))) int intx8;
typedef __attribute__((ext_vector_type(4))) int intx4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define MF32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define ME32(acc, a, b, sa, sb) asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))
#define MF16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define ME16(acc, a, b, sa, sb) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc) : "v"(a), "v"(b), "v"(sa), "v"(sb))

struct Res { long long cyc, wall; };

// MODE bit 0: shape (0 = 32x32, 1 = 16x16); bit 1: weight operands re-read from LDS each step (8 x ds_read_b128)
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(Res* out, const uint4* src_, int rep) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) ((uint4*)lds)[i] = src_[i];   // 64 KiB of fragments
  __syncthreads();
  const uint4* lp = (const uint4*)lds + lane;
  constexpr bool S16 = MODE & 1, RD = MODE & 2;
  intx4 f[4], bh[2][4];
  intx8 e[2], b8[2][2];
  for (int i = 0; i < 4; ++i) {
    f[i] = __builtin_bit_cast(intx4, lp[i * 64]);
    for (int h = 0; h < 2; ++h) bh[h][i] = __builtin_bit_cast(intx4, lp[(8 + 4 * h + i) * 64]);
  }
  for (int i = 0; i < 2; ++i)
    for (int q = 0; q < 8; ++q) {
      e[i][q] = (int)lp[(20 + i) * 64 + (q & 1)].x & 0x77777777;
      for (int h = 0; h < 2; ++h) b8[h][i][q] = (int)lp[(24 + 2 * h + i) * 64].y & 0x77777777;
    }
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 4; ++q) {
    f[i][q] = (f[i][q] & 0x3fff3fff) | 0x20002000;
    for (int h = 0; h < 2; ++h) bh[h][i][q] = (bh[h][i][q] & 0x3fff3fff) | 0x20002000;
  }
  int sa = 118, sb = 120;
  asm volatile("" : "+v"(sa), "+v"(sb));
  floatx16 acc32[4];
  floatx4 acc16[8][2];
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) acc32[i][q] = 0.001f * (i + q);
  for (int i = 0; i < 8; ++i) for (int h = 0; h < 2; ++h) for (int q = 0; q < 4; ++q) acc16[i][h][q] = 0.001f * (i + q + h);
  long long w0 = wall_clock64(), t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < rep; ++r) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint4* q = lp + g * 8 * 64;
#define READ_F(i) do { if (RD) { f[i] = __builtin_bit_cast(intx4, q[(i) * 64]); FENCE(); } } while (0)
#define READ_E(i) do { if (RD) { uint4 lo = q[(4 + 2 * (i)) * 64], hi = q[(5 + 2 * (i)) * 64]; e[i][0] = lo.x; e[i][1] = lo.y; e[i][2] = lo.z; e[i][3] = lo.w; e[i][4] = hi.x; e[i][5] = hi.y; e[i][6] = hi.z; e[i][7] = hi.w; FENCE(); } } while (0)
      if constexpr (!S16) {
        floatx16& c = acc32[g];
        MF32(c, f[0], bh[0][0]); FENCE(); READ_F(0);
        MF32(c, f[1], bh[0][1]); FENCE(); READ_F(1);
        MF32(c, f[2], bh[0][2]); FENCE(); READ_F(2);
        MF32(c, f[3], bh[0][3]); FENCE(); READ_F(3);
        ME32(c, e[0], b8[0][0], sa, sb); FENCE(); READ_E(0);
        ME32(c, e[1], b8[0][1], sb, sa); FENCE(); READ_E(1);
      } else {
        floatx4& c0 = acc16[2 * g][0]; floatx4& c1 = acc16[2 * g][1];
        MF16(c0, f[0], bh[0][0]); FENCE(); MF16(c1, f[0], bh[1][0]); FENCE(); READ_F(0);
        MF16(c0, f[1], bh[0][1]); FENCE(); MF16(c1, f[1], bh[1][1]); FENCE(); READ_F(1);
        MF16(c0, f[2], bh[0][2]); FENCE(); MF16(c1, f[2], bh[1][2]); FENCE(); READ_F(2);
        MF16(c0, f[3], bh[0][3]); FENCE(); MF16(c1, f[3], bh[1][3]); FENCE(); READ_F(3);
        ME16(c0, e[0], b8[0][0], sa, sb); FENCE(); ME16(c1, e[0], b8[1][0], sa, sb); FENCE(); READ_E(0);
        ME16(c0, e[1], b8[0][1], sb, sa); FENCE(); ME16(c1, e[1], b8[1][1], sb, sa); FENCE(); READ_E(1);
      }
    }
  }
  long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc32[i][q];
  for (int i = 0; i < 8; ++i) for (int h = 0; h < 2; ++h) for (int q = 0; q < 4; ++q) s += acc16[i][h][q];
  if (s == 12345.678f) out[0].cyc = 0;
  if (threadIdx.x == 0) { out[blockIdx.x].cyc = t1 - t0; out[blockIdx.x].wall = w1 - w0; }
}

template <int MODE>
static void run(const char* name, const uint4* dsrc, Res* dres) {
  const int rep = 20000, blocks = 256;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  k<MODE><<<blocks, 256, 65536>>>(dres, dsrc, 100);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 256, 65536>>>(dres, dsrc, rep);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
  std::vector<Res> r(blocks);
  CK(hipMemcpy(r.data(), dres, blocks * sizeof(Res), hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0; for (auto& x : r) { cyc += x.cyc; wall += x.wall; } cyc /= blocks; wall /= blocks;
  printf("%-72s %6.1f cyc/step  clock %.2f GHz  %.3f ms\n", name, cyc / (rep * 4.0), cyc / (wall * 10.0), ms);
}

int main() {
  std::vector<unsigned> h(65536 / 4);
  srand(5); for (auto& v : h) v = ((unsigned)rand() << 16) ^ rand();
  uint4* dsrc; Res* dres;
  CK(hipMalloc(&dsrc, h.size() * 4)); CK(hipMalloc(&dres, 256 * sizeof(Res)));
  CK(hipMemcpy(dsrc, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (int round = 0; round < 2; ++round) {
    run<0>("A  32x32: 4 f16 K=16 + 2 fp8 K=64, operands in registers", dsrc, dres);
    run<1>("B  16x16: 8 f16 K=32 + 4 fp8 K=128, operands in registers", dsrc, dres);
    run<2>("A  32x32 + 8 ds_read_b128 of the weight operands per step", dsrc, dres);
    run<3>("B  16x16 + 8 ds_read_b128 of the weight operands per step", dsrc, dres);
  }
  return 0;
}
