// Fused variants of the repacked decode GEMV for the decode harness (C entry in awq_aux.h):
//   * folded norm     y = inv_rms(v) * ((v * w) W), v = h + delta   (the reference's RMSNorm-with-residual before qkv_proj /
//                     gate_up_proj, python/sglang/srt/models/llama.py:277-290, carried through the GEMV's linearity:
//                     gemv_rp2_kernel<NORM>) — removes a ~4.7 us dependent launch per use;
//   * SiLU-mul epilogue  act = silu(gate) * up   (layers/activation.py SiluAndMul after gate_up_proj) on a copy
//                     repacked from column-interleaved gate / up groups, so both halves of a pair sit in one strip.
// Same kernel template as the plain operator (awq_repacked_gemv.h); only 16-wave, straight-line, one-row-tile
// instantiations exist, anything else returns AWQ_ERR_BAD_VARIANT and the caller runs the unfused sequence.
#include <cstdlib>

#include "../../include/awq_aux.h"
#include "awq_repacked_gemv.h"

namespace awq {

template <int EPI>
static bool fused_go(int G, const GemmArgs& a, const void* packed, int NG, int per_wave, int T, int nwg, size_t lds) {
  return rp_launch_g<16, true, 1, EPI>(G, a, packed, NG, per_wave, T, nwg, lds);
}

int launch_gemv_repacked_fused(const GemmArgs& a, const void* packed) {
  const bool norm = a.norm_h != nullptr;
  if (!norm && !a.silu_mul) return launch_gemv_repacked(a, packed);
  if (!repacked_fast(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.M > 32 || a.ldx % 8) return AWQ_ERR_BAD_VARIANT;   // fp16, g % 128 == 0 only
  if (a.M > 16) {                                                            // two row tiles: SiLU-mul epilogue only, 8 waves
    if (norm || !a.silu_mul || a.N % 32 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
    const int NG2 = rp_groups(a.N), KB2 = a.K / 128;
    int G2 = (NG2 + 255) / 256;
    if (G2 & 1) ++G2;
    if (G2 > kRpMaxG || G2 > NG2) return AWQ_ERR_BAD_VARIANT;
    const int pw = (KB2 + 7) / 8;
    const int T2 = (pw == 4 && rp_fits(8, 2, G2, 4)) ? 4 : 0;
    const size_t lds2 = (size_t)8 * a.M * 16 * G2 * sizeof(float);
    if (lds2 > (size_t)kRpMaxLds) return AWQ_ERR_BAD_VARIANT;
    if (!rp_launch_g<8, true, 2, 1>(G2, a, packed, NG2, pw, T2, (NG2 + G2 - 1) / G2, lds2)) return AWQ_ERR_BAD_VARIANT;
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
  }
  if (!norm && (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  if (a.silu_mul && a.N % 32) return AWQ_ERR_BAD_VARIANT;                  // whole (gate, up) pairs of 16-column groups
  constexpr int W = 16;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  int G = (NG + 255) / 256;
  if (a.silu_mul && (G & 1)) ++G;
  if (G > kRpMaxG && a.silu_mul && !norm && a.M <= 16 && (((uintptr_t)a.x) & 15) == 0) {
    // N > 32768 (e.g. the 57344-wide gate_up of a 70B model): rounds of 4-group strips on 8-wave workgroups, as the plain
    // launcher does, with the SiLU-mul epilogue (a strip holds two (gate, up) pairs)
    const int pw = (KB + 7) / 8;
    const size_t lds_r = (size_t)8 * a.M * 16 * 4 * sizeof(float);
    if (!rp_launch<4, 8, true, 1, 1>(a, packed, NG, pw, 0, (NG + 3) / 4, lds_r)) return AWQ_ERR_BAD_VARIANT;
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
  }
  if (G > kRpMaxG || G > NG) return AWQ_ERR_BAD_VARIANT;
  const int nwg = (NG + G - 1) / G;
  const int T = (KB + W - 1) / W;                                            // = per_wave: straight-line variants only
  if (norm) {
    if (!a.norm_delta || !a.norm_w || !a.norm_h_out || a.norm_h_out == a.norm_h) return AWQ_ERR_BAD_VARIANT;
    if ((((uintptr_t)a.norm_h) | ((uintptr_t)a.norm_delta) | ((uintptr_t)a.norm_w) | ((uintptr_t)a.norm_h_out)) & 15) return AWQ_ERR_BAD_VARIANT;
  }
  // gemv_rp2_kernel (norm folded through the GEMV and / or SiLU-mul epilogue) wherever it has an instantiation
  const bool ok = norm ? (a.silu_mul ? rp2_launch<1, true>(G, T, a, packed, NG, 2, nwg) : rp2_launch<0, true>(G, T, a, packed, NG, 2, nwg))
                       : rp2_launch<1, false>(G, T, a, packed, NG, 2, nwg);        // (!norm && !silu_mul went to the plain launcher above)
  if (ok) return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
  // no folded-norm instantiation (more than 4 staging chunks per lane): the caller runs the norm as its own launch.  The SiLU-mul
  // epilogue alone also exists on the round-1 kernel template (deeper K per wave than the straight-line form holds)
  if (norm || !rp_fits_fused(G, T, a.silu_mul)) return AWQ_ERR_BAD_VARIANT;
  const size_t lds = (size_t)W * a.M * 16 * G * sizeof(float);
  if (lds > (size_t)kRpMaxLds) return AWQ_ERR_BAD_VARIANT;
  if (!fused_go<1>(G, a, packed, NG, T, T, nwg, lds)) return AWQ_ERR_BAD_VARIANT;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// AWQ-MoE decode: `moe_slots` independent M = 1 GEMVs in one launch, grid row s on the repacked weight of expert
// moe_expert_ids[s] (gemv_rp2_kernel's expert indirection).  fp16, g % 128 == 0, >= 16 k-blocks.
int launch_gemv_repacked_moe(const GemmArgs& a, const void* packed) {
  if (!repacked_fast(a.K, a.N, a.g, a.dtype) || a.M != 1 || a.moe_slots < 1 || a.moe_slots > 65535 || !a.moe_expert_ids || a.moe_x_div < 1 ||
      a.ldx % 8 || (((uintptr_t)a.x) & 15) || (a.moe_expert_stride & 15))
    return AWQ_ERR_BAD_VARIANT;
  if (a.silu_mul && a.N % 32) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  int G = (NG + 255) / 256;
  if (a.silu_mul && (G & 1)) ++G;
  if (G > kRpMaxG || G > NG) return AWQ_ERR_BAD_VARIANT;
  const int nwg = (NG + G - 1) / G, T = (KB + 15) / 16;
  const bool ok = a.silu_mul ? rp2_launch<1, false>(G, T, a, packed, NG, 2, nwg) : rp2_launch<0, false>(G, T, a, packed, NG, 2, nwg);
  if (!ok) return AWQ_ERR_BAD_VARIANT;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq

extern "C" int awq_aux_moe_gemv(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                                int64_t num_experts, const int32_t* expert_ids, const float* slot_scale, void* y, int64_t slots,
                                int64_t K, int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream) {
  if (!x || !packed_experts || !expert_ids || !y) return AWQ_ERR_NULL_POINTER;
  if (K <= 0 || N <= 0 || group_size <= 0 || N % 8 || K % group_size || slots <= 0 || ldx < K || x_div < 1 || num_experts < 1 ||
      num_experts > INT32_MAX)
    return AWQ_ERR_BAD_SHAPE;
  if ((((uintptr_t)packed_experts) & 15) || (((uintptr_t)y) & 1)) return AWQ_ERR_MISALIGNED;
  awq::GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = nullptr; a.scales = nullptr; a.qzeros = nullptr; a.bias = nullptr; a.y = y;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.M = 1; a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = 0;
  a.stream = (hipStream_t)stream;
  a.silu_mul = silu_mul;
  a.moe_expert_ids = expert_ids; a.moe_slot_scale = slot_scale; a.moe_expert_stride = expert_stride_bytes; a.moe_x_div = x_div;
  a.moe_slots = (int)slots;
  a.moe_num_experts = (int)num_experts;
  return awq::launch_gemv_repacked_moe(a, packed_experts);
}

extern "C" int awq_aux_gemv_repacked_fused(const void* x, int64_t ldx, const void* packed, void* y, int64_t M, int64_t K, int64_t N,
                                           int64_t group_size, int dtype, const void* norm_h, const void* norm_delta,
                                           const void* norm_w, void* norm_h_out, float norm_eps, int silu_mul, void* stream) {
  if (!packed || !y || (!x && !norm_h)) return AWQ_ERR_NULL_POINTER;
  if (K <= 0 || N <= 0 || group_size <= 0 || N % 8 || K % group_size || M <= 0 || ldx < K) return AWQ_ERR_BAD_SHAPE;
  if ((((uintptr_t)packed) & 15) || (((uintptr_t)y) & 1)) return AWQ_ERR_MISALIGNED;
  awq::GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = nullptr; a.scales = nullptr; a.qzeros = nullptr; a.bias = nullptr; a.y = y;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.M = (int)M; a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = 0;
  a.stream = (hipStream_t)stream;
  a.norm_h = norm_h; a.norm_delta = norm_delta; a.norm_w = norm_w; a.norm_h_out = norm_h_out; a.norm_eps = norm_eps;
  a.silu_mul = silu_mul;
  return awq::launch_gemv_repacked_fused(a, packed);
}
