// extern "C" entry points declared in include/awq_hip.h: argument validation, variant choice,
// launch.  No allocation, no synchronisation, no global mutable state beyond a once-only kernel
// attribute in the skinny launcher.
#include <cstdlib>

#include "awq_kernels.h"

namespace {

using namespace awq;

int check_common(const void* qweight, const void* scales, const void* qzeros, int64_t K, int64_t N, int64_t g, int dtype) {
  if (!qweight || !scales || !qzeros) return AWQ_ERR_NULL_POINTER;
  if (dtype != AWQ_DTYPE_F16 && dtype != AWQ_DTYPE_BF16 && dtype != AWQ_DTYPE_F32) return AWQ_ERR_BAD_DTYPE;
  if (K <= 0 || N <= 0 || g <= 0 || N % 8 != 0 || K % g != 0) return AWQ_ERR_BAD_SHAPE;
  if (K > INT32_MAX / 2 || N > INT32_MAX / 2 || (K / 8) * N > (int64_t)INT32_MAX * 4) return AWQ_ERR_BAD_SHAPE;
  // scales rows are read and outputs written 16 bytes at a time (as the reference op does,
  // awq_kernel.cu:142,161); qweight / qzeros words 4 bytes at a time in the generic paths.
  if ((((uintptr_t)scales) & 15) || (((uintptr_t)qweight) & 3) || (((uintptr_t)qzeros) & 3)) return AWQ_ERR_MISALIGNED;
  return AWQ_OK;
}

bool pow2_le32(int v) { return v >= 1 && v <= 32 && (v & (v - 1)) == 0; }

}  // namespace

// Kernel choice on the fragment-major layout, shared by awq_gemm_repacked and the op's on-the-fly re-layout (so the two
// routes give bit-identical results).  The table of routes and the measurements behind every threshold: awq_dispatch.h.
static int repacked_dispatch(const GemmArgs& a, const void* packed) {
  const int64_t M = a.M;
  // <= 32 rows: the streaming GEMV.  A few passes of it (32 rows each) beat the MFMA tiles while those leave most CUs idle and
  // are bound by their K loop — how many, route::gemv_passes_max says (two on 4096 x 22016, three on 4096 x 11008, four on narrow
  // matrices; round 2 ran passes up to 160 rows everywhere: 4096 x 22016 at 128 rows 78.6 us against 44 on tiles).  Where the
  // wide tiles are few (narrow matrices, up to 512 rows) 128 x 64 tiles with the K split inside the workgroup fill the
  // chip better (11008 x 4096 at M = 256: 142 -> 90 us); beyond that the hand-pipelined wide tiles.
  static const int env_mid = lab_env("AWQ_MID", 1);      // lab knob: 0 = never the 128 x 64 tiles
  if (M <= route::kGemvMaxRows) {
    if (M >= route::kSplitKMinRows && a.workspace != nullptr) {             // narrow matrix, many rows: wide strips with K split across workgroups
      const int rs = launch_gemv_repacked_splitk(a, packed);
      if (rs != AWQ_ERR_BAD_VARIANT) return rs;
    }
    const int rc = launch_gemv_repacked(a, packed);
    if (rc != AWQ_ERR_BAD_VARIANT || M <= 16) return rc;
    // 17..32 rows on a strip whose two-row-tile reduction scratch does not fit the CU's LDS: two passes of <= 16 rows
    GemmArgs c = a;
    c.M = 16;
    const int rc1 = launch_gemv_repacked(c, packed);
    if (rc1) return rc1;
    c.M = (int)M - 16;
    c.x = (const char*)a.x + (size_t)16 * a.ldx * 2;
    c.y = (char*)a.y + (size_t)16 * a.N * 2;
    return launch_gemv_repacked(c, packed);
  }
  if (!repacked_fast(a.K, a.N, a.g, a.dtype)) return launch_gemm_repacked_tiled(a, packed);   // bf16 / g in {32, 64}: the generic tiles from 33 rows on
  const bool aligned = a.ldx % 8 == 0 && (((uintptr_t)a.x) & 15) == 0;
  const int64_t wide_tiles = ((M + 127) / 128) * ((a.N + 255) / 256);
  static const int env_sk = lab_env("AWQ_PF_SK", 1);     // lab knob: 0 = never the split-K tile route
  if (env_sk != 0 && aligned && wide_tiles <= route::kPfSplitMaxWideTiles && M >= route::kPfSplitMinRows) {
    const int rs = launch_gemm_repacked_split_tiles(a, packed);      // under-filled tiles + workspace: K split over workgroups
    if (rs != AWQ_ERR_BAD_VARIANT) return rs;
  }
  if (env_mid != 0 && aligned && M >= route::kKsplitMinRows && wide_tiles <= route::kKsplitMaxWideTiles && repacked_fast(a.K, a.N, a.g, a.dtype)) return launch_gemm_repacked_ksplit(a, packed);
  static const int env_passes = lab_env("AWQ_PASSES_MAX", 0);      // lab knob: rows up to which passes run (0 = the cost rule)
  const int64_t passes_max_rows = env_passes > 0 ? env_passes : (int64_t)route::gemv_passes_max(a.K, a.N) * route::kGemvPassRows;
  if (M <= passes_max_rows) {
    const size_t eb2 = 2;
    for (int64_t m0 = 0; m0 < M; m0 += route::kGemvPassRows) {
      GemmArgs c = a;
      c.M = (int)(M - m0 < route::kGemvPassRows ? M - m0 : route::kGemvPassRows);
      c.x = (const char*)a.x + (size_t)m0 * a.ldx * eb2;
      c.y = (char*)a.y + (size_t)m0 * a.N * eb2;
      const int rc = launch_gemv_repacked(c, packed);
      if (rc) return rc;
    }
    return AWQ_OK;
  }
  return launch_gemm_repacked_tiled(a, packed);
}

extern "C" {

int awq_hip_abi_version(void) { return AWQ_HIP_ABI_VERSION; }

const char* awq_hip_build_info(void) {
  return "sglang_awq_amd libawq_hip: gfx950 (CDNA4, wave64), HIP " __VERSION__;
}

const char* awq_hip_status_string(int s) {
  switch (s) {
    case AWQ_OK: return "ok";
    case AWQ_ERR_NULL_POINTER: return "null pointer";
    case AWQ_ERR_BAD_SHAPE: return "bad shape (need N % 8 == 0, K % group_size == 0, positive dims, ldx >= K)";
    case AWQ_ERR_BAD_DTYPE: return "bad dtype (0 = fp16, 1 = bf16, 2 = fp32)";
    case AWQ_ERR_BAD_SPLIT_K: return "split_k_iters must be a power of two in [1, 32]";
    case AWQ_ERR_WORKSPACE: return "workspace missing or smaller than awq_gemm_workspace_bytes()";
    case AWQ_ERR_MISALIGNED: return "pointer not sufficiently aligned (16 bytes for scales / x / y / workspace)";
    case AWQ_ERR_BAD_VARIANT: return "requested kernel variant cannot run this shape / dtype";
    case AWQ_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    default: return "unknown awq status";
  }
}

int awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out, int64_t K, int64_t N,
                   int64_t group_size, int dtype, void* stream) {
  int rc = check_common(qweight, scales, qzeros, K, N, group_size, dtype);
  if (rc) return rc;
  if (!out) return AWQ_ERR_NULL_POINTER;
  if (((uintptr_t)out) & 15) return AWQ_ERR_MISALIGNED;
  return launch_dequantize(qweight, scales, qzeros, out, K, N, group_size, dtype, (hipStream_t)stream);
}

// Calls of the op beyond the decode range re-lay the weight out on the fly: one pass over the packed weight into the
// workspace (awq_repack, ~17 us at 4096 x 11008) buys the kernels of the fragment-major layout.  Measured against the
// checkpoint-layout kernels (11008 x 4096: M = 64 212 -> 63 us, M = 256 217 -> 118, M = 2048 311 -> 225; 4096 x 11008:
// M = 128 85 -> 61, M = 512 104 -> 89) it wins from the first row count the split-K decode kernel does not cover.
constexpr int64_t kRepackOnTheFlyMinM = route::kRepackOnTheFlyMinM;
constexpr size_t kWorkspaceHead = 4096;        // arrival counters of the split-K kernel live here

size_t awq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype) {
  if (M <= 0 || K <= 0 || N <= 0) return kWorkspaceHead;
  if (M >= kRepackOnTheFlyMinM && repacked_fast(K, N, group_size, dtype)) return kWorkspaceHead + repacked_bytes(K, N, group_size);
  return skinny_workspace_bytes(M, K, N);
}

int awq_gemm_ex(const void* x, int64_t ldx, const int32_t* qweight, const void* scales, const int32_t* qzeros,
                const void* bias, void* y, void* workspace, size_t workspace_bytes, int64_t M, int64_t K, int64_t N,
                int64_t group_size, int dtype, int split_k_iters, int variant, int64_t tune, void* stream) {
  int rc = check_common(qweight, scales, qzeros, K, N, group_size, dtype);
  if (rc) return rc;
  if (M < 0 || ldx < K || M > INT32_MAX / 2) return AWQ_ERR_BAD_SHAPE;
  if (!pow2_le32(split_k_iters)) return AWQ_ERR_BAD_SPLIT_K;
  if (M == 0) return AWQ_OK;
  if (!x || !y) return AWQ_ERR_NULL_POINTER;
  const int eb = dtype == AWQ_DTYPE_F32 ? 4 : 2;
  if ((((uintptr_t)x) | ((uintptr_t)y)) & (uintptr_t)(eb - 1)) return AWQ_ERR_MISALIGNED;

  GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = qweight; a.scales = scales; a.qzeros = qzeros; a.bias = bias; a.y = y;
  a.workspace = workspace; a.workspace_bytes = workspace_bytes;
  a.M = (int)M; a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = tune;
  a.stream = (hipStream_t)stream;

  switch (variant) {
    case AWQ_GEMM_GENERIC: return launch_gemm_generic(a);
    case AWQ_GEMM_SKINNY: return launch_gemm_skinny(a);
    case AWQ_GEMM_TILED: return tiled_supported(a) ? launch_gemm_tiled(a) : AWQ_ERR_BAD_VARIANT;
    case AWQ_GEMM_AUTO: break;
    default: return AWQ_ERR_BAD_VARIANT;
  }
  if (skinny_supported(a)) {
    if (!workspace || workspace_bytes < awq_gemm_workspace_bytes(M, K, N, group_size, dtype)) return AWQ_ERR_WORKSPACE;
    return launch_gemm_skinny(a);
  }
  if (M >= kRepackOnTheFlyMinM && repacked_fast(K, N, group_size, dtype) && workspace && (((uintptr_t)workspace) & 15) == 0 &&
      workspace_bytes >= kWorkspaceHead + repacked_bytes(K, N, group_size) && a.ldx % 8 == 0 && (((uintptr_t)x) & 15) == 0) {
    void* packed = (char*)workspace + kWorkspaceHead;
    rc = launch_repack(qweight, scales, qzeros, packed, K, N, group_size, dtype, a.stream);
    if (rc) return rc;
    // the re-laid-out copy occupies the front of the workspace: a route that wants scratch of its own (split-K tiles) gets what lies behind it
    GemmArgs c = a;
    const size_t used = (kWorkspaceHead + repacked_bytes(K, N, group_size) + 15) & ~(size_t)15;
    c.workspace = workspace_bytes > used ? (char*)workspace + used : nullptr;
    c.workspace_bytes = workspace_bytes > used ? workspace_bytes - used : 0;
    return repacked_dispatch(c, packed);
  }
  // 16 < M <= 48: the 128-row tiles of the prefill kernel would leave most CUs idle (N / 128 workgroups);
  // two or three passes of the decode kernel over 16-row slabs of x are faster (measured 82 us tiled vs
  // ~20 us per pass at 4096 x 11008).  Passes are stream-ordered, so they can share the workspace.
  if (M > kSkinnyMaxM && M <= 3 * kSkinnyMaxM) {
    GemmArgs c = a;
    c.M = kSkinnyMaxM;
    if (skinny_supported(c) && workspace && workspace_bytes >= awq_gemm_workspace_bytes(kSkinnyMaxM, K, N, group_size, dtype)) {
      const size_t eb2 = 2;   // skinny_supported() admitted fp16 / bf16 only
      for (int64_t m0 = 0; m0 < M; m0 += kSkinnyMaxM) {
        c.M = (int)(M - m0 < kSkinnyMaxM ? M - m0 : kSkinnyMaxM);
        c.x = (const char*)x + (size_t)m0 * ldx * eb2;
        c.y = (char*)y + (size_t)m0 * N * eb2;
        const int rc2 = launch_gemm_skinny(c);
        if (rc2) return rc2;
      }
      return AWQ_OK;
    }
  }
  if (tiled_supported(a)) return launch_gemm_tiled(a);
  return launch_gemm_generic(a);
}

int awq_gemm(const void* x, int64_t ldx, const int32_t* qweight, const void* scales, const int32_t* qzeros,
             const void* bias, void* y, void* workspace, size_t workspace_bytes, int64_t M, int64_t K, int64_t N,
             int64_t group_size, int dtype, int split_k_iters, void* stream) {
  return awq_gemm_ex(x, ldx, qweight, scales, qzeros, bias, y, workspace, workspace_bytes, M, K, N, group_size, dtype,
                     split_k_iters, AWQ_GEMM_AUTO, 0, stream);
}

size_t awq_repacked_bytes(int64_t K, int64_t N, int64_t group_size, int dtype) {
  return repacked_supported(K, N, group_size, dtype) ? repacked_bytes(K, N, group_size) : 0;
}

int awq_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* packed, int64_t K, int64_t N,
               int64_t group_size, int dtype, void* stream) {
  int rc = check_common(qweight, scales, qzeros, K, N, group_size, dtype);
  if (rc) return rc;
  if (!packed) return AWQ_ERR_NULL_POINTER;
  if (((uintptr_t)packed) & 15) return AWQ_ERR_MISALIGNED;
  return launch_repack(qweight, scales, qzeros, packed, K, N, group_size, dtype, (hipStream_t)stream);
}

size_t awq_gemm_repacked_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype) {
  if (!repacked_fast(K, N, group_size, dtype)) return 0;
  const size_t a = rps_workspace_bytes(M, K, N), b = pf_split_workspace_bytes(M, K, N);      // split-K GEMV (9 .. 32 rows) / split-K tiles (33 rows up)
  return a > b ? a : b;
}

int awq_gemm_repacked(const void* x, int64_t ldx, const void* packed, const void* bias, void* y, int64_t M, int64_t K,
                      int64_t N, int64_t group_size, int dtype, void* stream) {
  return awq_gemm_repacked_ws(x, ldx, packed, bias, y, nullptr, 0, M, K, N, group_size, dtype, stream);
}

int awq_gemm_repacked_ws(const void* x, int64_t ldx, const void* packed, const void* bias, void* y, void* workspace,
                         size_t workspace_bytes, int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype, void* stream) {
  if (!packed) return AWQ_ERR_NULL_POINTER;
  if (K <= 0 || N <= 0 || group_size <= 0 || N % 8 || K % group_size || M < 0 || ldx < K) return AWQ_ERR_BAD_SHAPE;
  if (M == 0) return AWQ_OK;
  if (!x || !y) return AWQ_ERR_NULL_POINTER;
  if ((((uintptr_t)packed) & 15) || (((uintptr_t)y) & 1)) return AWQ_ERR_MISALIGNED;
  // every kernel behind this entry indexes `packed` from (K, N, group_size) alone: refuse shapes the layout does not exist for
  if (!repacked_supported(K, N, group_size, dtype) || M > INT32_MAX / 2) return AWQ_ERR_BAD_VARIANT;
  GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = nullptr; a.scales = nullptr; a.qzeros = nullptr; a.bias = bias; a.y = y;
  a.workspace = workspace; a.workspace_bytes = workspace ? workspace_bytes : 0;      // optional: only the split-K route for 9..32 rows uses it
  a.M = (int)M; a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = 0;
  a.stream = (hipStream_t)stream;
  return repacked_dispatch(a, packed);
}

}  // extern "C"
