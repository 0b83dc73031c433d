# NOTE: written against the round-2 kernels (its anchors no longer match the source); the current kernels are profiled
# with tools/seg_patch_mx4.py / tools/seg_probe_mx4.py.
# per-segment cycle counters for sdf_decode_mx (applied to a copy of sdf.hip by tools/seg.sh with SEGPATCH=tools/seg_patch_mx.py)
def rep(a, b):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, 1)
rep("  unsigned voff;         // lane * 4\n", "  unsigned voff;         // lane * 4\n  long long tacc[8], tlast, twait, twait2;\n")
rep("    asm volatile(\"s_waitcnt vmcnt(32)\" ::: \"memory\");   // all but the 32 sub-pieces of stage g+2\n    __builtin_amdgcn_s_barrier();",
    "    long long tw0_ = clock64();\n    asm volatile(\"s_waitcnt vmcnt(32)\" ::: \"memory\");   // all but the 32 sub-pieces of stage g+2\n    long long tw1_ = clock64();\n    __builtin_amdgcn_s_barrier();\n    long long tw2_ = clock64(); twait += tw1_ - tw0_; twait2 += tw2_ - tw1_;")
rep("#define SCHED_FENCE()", "#define TICK(i) { long long t_ = clock64(); pipe.tacc[i] += t_ - pipe.tlast; pipe.tlast = t_; }\n#define SCHED_FENCE()")
i0 = s.find("__device__ __forceinline__ float stream_pass_mx(")
head, s = s[:i0], s[i0:]
rep("  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks (one group each); h2 stays resident as 4 operand groups", "  if (LOCAL) TICK(0) else TICK(4)\n  // fold1/conv2 (64 -> 256): 2 stages")
rep("  // fold1/conv3 (256 -> 512) chained into fold2/conv1 in groups of 64 features", "  TICK(1)\n  // fold1/conv3 (256 -> 512) chained")
rep("  // Two operand sets in ping-pong.", "  TICK(2)\n  // Two operand sets in ping-pong.")
rep("    CvtMx cs;\n    const float* bnext", "    TICK(3)\n    CvtMx cs;\n    const float* bnext")
rep("  // fold2/conv2 (512 -> 256) on the bf16 triple", "  TICK(5)\n  // fold2/conv2 (512 -> 256) on the bf16 triple")
rep("  float part = 0.0f;\n", "  TICK(6)\n  float part = 0.0f;\n")
rep("  part += __shfl_xor(part, 32, 64);\n", "  part += __shfl_xor(part, 32, 64);\n  TICK(7)\n")
rep("  Ring8 rg;\n", "  for (int i = 0; i < 8; ++i) pipe.tacc[i] = 0;\n  pipe.twait = 0; pipe.twait2 = 0;\n  pipe.tlast = clock64();\n  long long w0 = wall_clock64();\n  Ring8 rg;\n")
rep('''  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n}''', '''  __syncthreads();
  if (LOCAL == (SEG_LOCAL != 0) && blockIdx.x == 7 && threadIdx.x == 0) { for (int i = 0; i < 8; ++i) ((long long*)a.out)[i] = pipe.tacc[i]; ((long long*)a.out)[8] = wall_clock64() - w0; ((long long*)a.out)[9] = pipe.twait; ((long long*)a.out)[10] = pipe.twait2; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n}''')
s = head + s
# the f2c1 phase tick: after the 16 groups of each chain step
rep("    });\n  };\n#pragma unroll 1", "    });\n    TICK(5)\n  };\n#pragma unroll 1")
# which stream's kernel reports (the other launch is dropped for the global stream: it would overwrite the counters)
import os as _os
_loc = _os.environ.get("SEG_LOCAL", "1")
rep("#define SCHED_FENCE()", "#define SEG_LOCAL %s\n#define SCHED_FENCE()" % _loc)
if _loc == "0":
    rep("    sdf_decode_mx<true><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);    // out = g + l (scaled)", "")
