s = s.replace("  unsigned voff;         // lane * 4\n", "  unsigned voff;         // lane * 4\n  long long tacc[8], tlast;\n", 1)
s = s.replace("#define SCHED_FENCE()", "#define TICK(i) { long long t_ = clock64(); pipe.tacc[i] += t_ - pipe.tlast; pipe.tlast = t_; }\n#define SCHED_FENCE()", 1)
# T0 after fold1/conv1
s = s.replace("  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks", "  if (LOCAL) TICK(0) else TICK(4)\n  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks", 1)
s = s.replace("  // fold1/conv3 (256 -> 512) chained into fold2/conv1: each finished", "  TICK(1)\n  // fold1/conv3 (256 -> 512) chained into fold2/conv1: each finished", 1)
s = s.replace("#pragma unroll 1\n  for (int nb3 = 0; nb3 < 15; ++nb3) {\n    floatx16 acc3n = bnext;\n    l3_stage(pipe, fr, acc3n, h2hi, h2lo);\n",
              "  TICK(2)\n#pragma unroll 1\n  for (int nb3 = 0; nb3 < 15; ++nb3) {\n    floatx16 acc3n = bnext;\n    l3_stage(pipe, fr, acc3n, h2hi, h2lo);\n    TICK(3)\n", 1)
s = s.replace("    xhi[0] = nhi[0]; xhi[1] = nhi[1]; xlo[0] = nlo[0]; xlo[1] = nlo[1];\n  }\n", "    xhi[0] = nhi[0]; xhi[1] = nhi[1]; xlo[0] = nlo[0]; xlo[1] = nlo[1];\n    TICK(5)\n  }\n", 1)
s = s.replace("  // fold2/conv2 (512 -> 256), consuming the fold2/conv1 accumulators block by block;", "  TICK(5)\n  // fold2/conv2 (512 -> 256), consuming the fold2/conv1 accumulators block by block;", 1)
s = s.replace("  // fold2/conv5 (256 -> 1): this lane holds half", "  TICK(6)\n  // fold2/conv5 (256 -> 1): this lane holds half", 1)
s = s.replace("  part += __shfl_xor(part, 32, 64);\n", "  part += __shfl_xor(part, 32, 64);\n  TICK(7)\n", 1)
s = s.replace("  Frags fr;\n#pragma unroll\n  for (int t = 0; t < 3; ++t) {", "  for (int i = 0; i < 8; ++i) pipe.tacc[i] = 0;\n  pipe.tlast = clock64();\n  long long w0 = wall_clock64();\n  Frags fr;\n#pragma unroll\n  for (int t = 0; t < 3; ++t) {", 1)
s = s.replace('''  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive''','''  __syncthreads();
  if (blockIdx.x == 7 && threadIdx.x == 0) { for (int i = 0; i < 8; ++i) ((long long*)a.out)[i] = pipe.tacc[i]; ((long long*)a.out)[8] = wall_clock64() - w0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive''',1)
