// Internal launcher declarations shared by the .hip translation units and the C ABI (awq_capi.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdlib.h>
#include <stdint.h>

#include "../../include/awq_hip.h"
#include "awq_device.h"
#include "awq_dispatch.h"

namespace awq {

// A/B knobs of the laboratory build (make EXTRA=-DAWQ_LAB; tools/ab_*.sh, tools/kbench): environment variables that force a
// kernel variant.  The product library reads no environment: every lab_env() folds to its default, which is the measured choice
// recorded in DESIGN.md.
#ifdef AWQ_LAB
inline int lab_env(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int lab_env(const char*, int dflt) { return dflt; }
#endif

struct GemmArgs {
  const void* x;
  int64_t ldx;
  const int32_t* qweight;
  const void* scales;
  const int32_t* qzeros;
  const void* bias;   // may be null
  void* y;
  void* workspace;
  size_t workspace_bytes;
  int M, K, N, g;
  int dtype;
  int64_t tune;
  hipStream_t stream;
  // optional fusions of the repacked decode GEMV (awq_aux.h); all zero = the plain operator
  const void* norm_h = nullptr;      // x = rmsnorm(norm_h + norm_delta) * norm_w is built in the prologue (a.x is ignored)
  const void* norm_delta = nullptr;
  const void* norm_w = nullptr;
  void* norm_h_out = nullptr;        // norm_h + norm_delta, [M, K] with row stride ldx (must not alias norm_h)
  float norm_eps = 0.f;
  int silu_mul = 0;                  // 1: column groups alternate gate / up; y = silu(gate) * up, [M, N / 2]
  // AWQ-MoE decode (awq_aux_moe_gemv): `moe_slots` (token, expert) pairs, one grid row each; see RpFuse in awq_repacked_gemv.h
  const int* moe_expert_ids = nullptr;
  const float* moe_slot_scale = nullptr;
  int64_t moe_expert_stride = 0;     // bytes between the repacked weights of consecutive experts
  int moe_x_div = 1;                 // activation row of slot s = s / moe_x_div
  int moe_slots = 0;
  int moe_num_experts = 0;           // ids outside [0, moe_num_experts) are padded slots: zero output rows
};

int launch_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out, int64_t K,
                      int64_t N, int64_t g, int dtype, hipStream_t stream);

// generic VALU kernel: any legal shape / dtype
int launch_gemm_generic(const GemmArgs& a);

// skinny MFMA kernel (fp16 / bf16, M <= 16, N % 32 == 0, K % 32 == 0, g % 32 == 0)
constexpr int kSkinnyMaxM = 16;
bool skinny_supported(const GemmArgs& a);
size_t skinny_workspace_bytes(int64_t M, int64_t K, int64_t N);
int launch_gemm_skinny(const GemmArgs& a);

// tiled MFMA kernel for large M (fp16 / bf16, N % 64 == 0, K % 32 == 0, g % 32 == 0)
bool tiled_supported(const GemmArgs& a);
int launch_gemm_tiled(const GemmArgs& a);

// MFMA-fragment-major re-layout + the kernels on it (fp16 / bf16, K % 128 == 0, g % 128 == 0 or g in {32, 64}): decode GEMV, prefill tiles
size_t repacked_bytes(int64_t K, int64_t N, int64_t g);
bool repacked_supported(int64_t K, int64_t N, int64_t g, int dtype);
// the subset with the tuned kernels (fp16, g % 128 == 0): straight-line GEMV, hand-pipelined prefill, fused decode variants
bool repacked_fast(int64_t K, int64_t N, int64_t g, int dtype);
bool pipelined_addressable(const GemmArgs& a);                                 // spans fit the prefill kernel's 32-bit buffer offsets
int launch_gemv_repacked_splitk(const GemmArgs& a, const void* packed);   // 9..32 rows on narrow matrices, K split across workgroups, needs a.workspace (awq_repacked_splitk.hip)
size_t rps_workspace_bytes(int64_t M, int64_t K, int64_t N);
int launch_gemv_repacked_ext(const GemmArgs& a, const void* packed);          // bf16 / small groups, M <= 16 (awq_repacked_ext.hip)
int launch_gemm_repacked_tiled_ext(const GemmArgs& a, const void* packed);    // bf16 / small groups, large M
int launch_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* packed, int64_t K, int64_t N, int64_t g,
                  int dtype, hipStream_t stream);
int launch_gemv_repacked(const GemmArgs& a, const void* packed);
bool gemv_strip_geometry(int64_t K, int64_t N, int* G, int* nwg);          // strip width / workgroups of the one-strip-per-workgroup decode GEMV; false: rounds mode
int launch_gemv_repacked_loop(const GemmArgs& a, const void* packed);    // loop form of the straight-line GEMV: deep K, 9..16 rows on deep matrices (awq_repacked_loop.hip)
int launch_gemv_repacked_fused(const GemmArgs& a, const void* packed);   // norm prologue and / or SiLU-mul epilogue (awq_repacked_fused.hip)
int launch_gemv_repacked_moe(const GemmArgs& a, const void* packed);     // expert-indirect M = 1 GEMVs, one grid row per (token, expert) slot
int launch_gemm_repacked_tiled(const GemmArgs& a, const void* packed);   // any M, MFMA-bound prefill shapes
int launch_gemm_repacked_ksplit(const GemmArgs& a, const void* packed);      // 128 x 64 tiles, K split inside the workgroup (middle M)
int launch_gemm_repacked_pipelined(const GemmArgs& a, const void* packed);   // its hand-pipelined 128 x 256 form (awq_repacked_prefill.hip)
// under-filled launches with a workspace: the same tiles with K split over workgroups + a reduce launch; BAD_VARIANT where it does not apply
int launch_gemm_repacked_split_tiles(const GemmArgs& a, const void* packed);
bool pf_split_plan(const GemmArgs& a, int* NJ_out, int* S_out, int* kb_per_out);
size_t pf_split_workspace_bytes(int64_t M, int64_t K, int64_t N);
// the same kernel over expert-sorted 128-row blocks of (token, expert) pairs (AWQ-MoE prefill; awq_aux_moe_gemm_blocks)
int launch_gemm_repacked_moe_tiles(const GemmArgs& a, const void* packed_experts, const int* row_map, const int* block_expert, int num_blocks,
                                   int block_rows, const float* slot_scale, long long expert_stride, int x_div, bool silu_mul);

// Opt a kernel in to more than 64 KiB of dynamic LDS on the CURRENT device.  The attribute belongs to the function object of
// a device, and the shim serves several devices from one process: remember per (kernel, device) instead of once per process.
// Returns false when the runtime refuses (callers report AWQ_ERR_LAUNCH / fall back).
inline bool opt_in_dynamic_lds(const void* kernel, int bytes, unsigned long long (&done)[2]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return false;
  if (dev < 128 && (done[dev >> 6] >> (dev & 63) & 1ull)) return true;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  if (dev < 128) done[dev >> 6] |= 1ull << (dev & 63);
  return true;
}

}  // namespace awq
