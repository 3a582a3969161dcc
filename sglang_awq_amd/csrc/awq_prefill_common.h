// Pieces shared by the prefill kernels on the MFMA-fragment-major layout (awq_repacked_prefill.hip, awq_repacked_prefill_pc.hip).
#pragma once
#include "awq_repacked_gemv.h"

namespace awq {

__device__ __forceinline__ void mfma_tied(float4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ int pfp_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }   // [rows][128 halves]

// A value loaded at the top of a k-block but first needed late in it: without this, hipcc hoists the (pure) consumer
// up to the load and waits out the whole load latency at the top of the block (`s_waitcnt vmcnt` right behind the
// loads — a quarter of the kernel's wave-cycles).  The volatile pass-through cannot cross the sched_barriers.
__device__ __forceinline__ void pin_here(uint32_t& v) { asm volatile("" : "+v"(v)); }

struct ZsU { half2_t s2, z1024, z64; };
__device__ __forceinline__ ZsU zs_unpack(uint32_t zs) {
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  ZsU u;
  u.s2 = as_h2(pack_lo16(zs, zs));
  u.z1024 = as_h2(pack_hi16(zs, zs));
  u.z64 = u.z1024 - c960;                               // exact: (1024 + z) - 960
  return u;
}

constexpr int kPfBM = 128;                        // rows per workgroup tile
constexpr size_t kPfSplitHead = 4096;             // the workspace's counter head (zero between calls: other routes' tickets) is left alone
constexpr int kPfRsrcFlags = 0x00020000;          // raw buffer descriptor, 32-bit data format (gfx9 family)

}  // namespace awq
