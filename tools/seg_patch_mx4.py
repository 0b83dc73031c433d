# per-segment cycle counters for sdf_decode_mx<LOCAL, true> as of round 3 (matrix-product image term, all-mixed layers):
# applied to a copy of sdf.hip by tools/seg.sh with SEGPATCH=tools/seg_patch_mx4.py SEGPROBE=tools/seg_probe_mx4.py.
# SEG_LOCAL=1 (default) reports the local stream's launch, 0 the global stream's.
def rep(a, b, where=None):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, 1)
rep("  unsigned voff;         // lane * 4\n", "  unsigned voff;         // lane * 4\n  long long tacc[12], tlast;\n")
rep("#define SCHED_FENCE()", "#define TICK(i) { long long t_ = clock64(); pipe.tacc[i] += t_ - pipe.tlast; pipe.tlast = t_; }\n#define SCHED_FENCE()")
i0 = s.find("__device__ __forceinline__ float stream_pass_mx(")
head, s = s[:i0], s[i0:]
rep("  floatx16 acc4[16];\n  Corners cn;\n", "  TICK(0)\n  floatx16 acc4[16];\n  Corners cn;\n")           # 0: tile set-up before the pass (grid point, projection) + store of the previous tile
rep('  asm volatile("; disn:pipelined_region_begin" ::: "memory");', '  TICK(1)\n  asm volatile("; disn:pipelined_region_begin" ::: "memory");')   # 1: image term / bias
rep("  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks", "  TICK(2)\n  // fold1/conv2 (64 -> 256): 2 stages of 4 blocks")   # 2: fold1/conv1 + its split
rep("  // fold1/conv3 (256 -> 512) chained into fold2/conv1 in groups of 64 features", "  TICK(3)\n  // fold1/conv3 (256 -> 512) chained into")   # 3: fold1/conv2
rep("  // Two operand sets in ping-pong.", "  TICK(4)\n  // Two operand sets in ping-pong.")   # 4: first two fold1/conv3 stages + split
rep("  chain(xc, xd, 6, std::true_type{}, std::false_type{}, no_tail);\n", "  chain(xc, xd, 6, std::true_type{}, std::false_type{}, no_tail);\n  TICK(5)\n")   # 5: chain steps 0..6 (14 + 28 stages)
rep("    const Scales sc5 = {load_scale(sp, 3), xl_scale};\n", "    TICK(6)\n    const Scales sc5 = {load_scale(sp, 3), xl_scale};\n")   # 6: chain step 7 (4 stages)
rep("  drain_v(acc5[7]);\n  float part;\n", "  TICK(7)\n  drain_v(acc5[7]);\n  float part;\n")   # 7: fold2/conv2 (16 stages)
rep("  part += __shfl_xor(part, 32, 64);\n  return part + sp[SP_B6];", "  part += __shfl_xor(part, 32, 64);\n  TICK(8)\n  return part + sp[SP_B6];")   # 8: last layer
rep("  Ring8 rg;\n", "  for (int i = 0; i < 12; ++i) pipe.tacc[i] = 0;\n  pipe.tlast = clock64();\n  long long w0 = wall_clock64();\n  Ring8 rg;\n")
rep('''  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n}''', '''  TICK(0)
  __syncthreads();
  if (LOCAL == (SEG_LOCAL != 0) && L4MX && blockIdx.x == 7 && threadIdx.x == 0) { for (int i = 0; i < 12; ++i) ((long long*)a.out)[i] = pipe.tacc[i]; ((long long*)a.out)[12] = wall_clock64() - w0; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n}''')
s = head + s
import os as _os
_loc = _os.environ.get("SEG_LOCAL", "1")
rep("#define SCHED_FENCE()", "#define SEG_LOCAL %s\n#define SCHED_FENCE()" % _loc)
if _loc == "0":
    rep("      sdf_decode_mx<true, true><<<blocks, 256, MX_LDS_BYTES, stream>>>(a);    // out = g + l (scaled)\n", "")
