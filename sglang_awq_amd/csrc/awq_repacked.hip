// MFMA-fragment-major re-layout of AWQ weights + the decode GEMV that consumes it (gfx950).
//
// The reference keeps the AutoAWQ layout at load time (process_weights_after_loading is a no-op
// re-wrap, awq.py:429-432) and has a one-time re-layout only for its NVIDIA Marlin path
// (sgl-kernel/csrc/gemm/marlin/awq_marlin_repack.cu:13-253).  This is the CDNA4
// counterpart: an optional, one-time repack into the layout v_mfma_f32_16x16x32_f16 wants, so the
// decode kernel streams the weights linearly, needs no in-register transposes, no split-K across
// workgroups and 12 accumulator registers instead of 128.  The original-layout ops stay the
// drop-in boundary; numerics are identical (every W element is still (q - z) * s rounded once to
// fp16, products exact, fp32 accumulation, one final rounding).
//
// Layout (all little-endian dwords), NG = ceil(N / 16) column groups, KB = K / 128 k-blocks:
//   qw_r[NG][KB][64 lanes][4]   lane l = (q = l >> 4, r = l & 15), dword d: the 8 int4 of column
//                               n = 16 cg + r for rows k = 128 kb + 32 d + 8 q + j, j = 0..7; row j = 2t sits in
//                               nibble t, row j = 2t + 1 in nibble t + 4, so (w >> 4t) & 0x000f000f is the packed
//                               pair (k_2t, k_2t+1): the B fragment of one MFMA comes out of ONE dword in order.
//                               One wave-wide global_load_dwordx4 = 1 KiB = 128 rows x 16 columns, contiguous.
//   zs_r[NG][K / g][16]         per (column, quantisation group): low half = scale (fp16 bits), high half =
//                               fp16(1024 + zero).  Columns >= N are padded with scale 0.
// A workgroup owns G consecutive column groups (a strip of 16 G columns) for ALL of K; its 16 (or 8) waves take
// consecutive k-block ranges and are summed in fixed order through LDS; y is written directly.
// Kernels: awq_repacked_gemv.h (decode GEMV templates), awq_repacked_fused.hip (its fused variants),
// awq_repacked_prefill.hip (hand-pipelined prefill GEMM), awq_repacked_ext.hip (bf16 / small groups); this file: re-layout
// and launch heuristics.
#include <cstdlib>

#include "awq_repacked_gemv.h"

namespace awq {

// ------------------------------------------------------------------------------------------ repack
__global__ __launch_bounds__(256) void repack_qweight_kernel(const uint32_t* __restrict__ qw, uint32_t* __restrict__ out,
                                                             int K, int C, int NG) {
  const size_t total = (size_t)NG * (K / 128) * 256;          // dwords
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const size_t blk = idx >> 8;
    const int kb = (int)(blk % (K / 128));
    const int cg = (int)(blk / (K / 128));
    const int q = lane >> 4, r = lane & 15;
    const int n = cg * 16 + r;
    uint32_t w = 0;
    if (n < C * 8) {
      const int word = n >> 3, jn = n & 7;
      const int shift = ((jn & 1) << 4) + ((jn >> 1) << 2);    // 4 * {0,4,1,5,2,6,3,7}[jn]
      const int k0 = kb * 128 + d * 32 + q * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t nib = (qw[(size_t)(k0 + j) * C + word] >> shift) & 0xFu;
        w |= nib << (((j & 1) << 4) + ((j >> 1) << 2));       // row j: nibble j/2 (even j) or j/2 + 4 (odd j)
      }
    }
    out[idx] = w;
  }
}

// LDS-tiled form of the same re-layout: a workgroup takes one k-block (128 rows) x 32 column groups = 128 rows x 64 packed
// dwords, reads it with fully coalesced 256 B row segments into LDS, and every lane assembles its four output dwords of a
// column group from LDS and stores them as ONE dwordx4 (a wave writes 1 KiB contiguous).  The scattered 4-byte reads of
// the simple kernel above ran at ~1.3 TB/s (29-39 us at 4096 x 11008); this pass is on the prefill path of the awq_gemm
// op (on-the-fly re-layout) and on every model load.
__global__ __launch_bounds__(256) void repack_qweight_tiled_kernel(const uint32_t* __restrict__ qw, u32x4_t* __restrict__ out, int K, int C,
                                                                   int NG) {
  constexpr int RS = 65;                                   // LDS row stride in dwords (64 + 1: rows 8 apart hit different banks)
  __shared__ uint32_t tile[128 * RS];
  const int KB = K / 128;
  const int kb = blockIdx.x % KB, gt = blockIdx.x / KB;     // k-block, tile of 32 column groups
  const int w0 = gt * 64;                                   // first packed dword (8 columns each) of the tile
  const int t = threadIdx.x;
  // 128 rows x 16 dwordx4: thread t takes chunk (t & 15) of rows (t >> 4) + 16 i
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (t >> 4) + 16 * i, ch = t & 15;
    const int wd = w0 + ch * 4;
    const uint32_t* src = qw + (size_t)(kb * 128 + row) * C + wd;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (wd + 3 < C && (((uintptr_t)src) & 15) == 0) v = *(const u32x4_t*)src;
    else {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (wd + e < C) v[e] = src[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[row * RS + ch * 4 + e] = v[e];
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int shift = ((r & 1) << 4) + (((r & 7) >> 1) << 2);   // 4 * {0,4,1,5,2,6,3,7}[r & 7]
#pragma unroll
  for (int g8 = 0; g8 < 8; ++g8) {
    const int cgl = wave * 8 + g8, cg = gt * 32 + cgl;
    if (cg >= NG) break;
    const int wl = cgl * 2 + (r >> 3);                        // this lane's column lives in that dword of the row
    u32x4_t o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      uint32_t w = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t nib = (tile[(d * 32 + q * 8 + j) * RS + wl] >> shift) & 0xFu;
        w |= nib << (((j & 1) << 4) + ((j >> 1) << 2));        // row j: nibble j/2 (even j) or j/2 + 4 (odd j)
      }
      o[d] = w;
    }
    out[((size_t)cg * KB + kb) * 64 + lane] = o;
  }
}

__global__ __launch_bounds__(256) void repack_zs_kernel(const uint32_t* __restrict__ qz, const uint16_t* __restrict__ scales,
                                                        uint32_t* __restrict__ out, int groups, int C, int NG) {
  const size_t total = (size_t)NG * groups * 16;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(idx & 15);
    const int grp = (int)((idx >> 4) % groups);
    const int cg = (int)((idx >> 4) / groups);
    const int n = cg * 16 + r;
    uint32_t v = 0x64000000u;                                  // zero point 0, scale 0
    if (n < C * 8) {
      const int z = nibble_of_col(qz[(size_t)grp * C + (n >> 3)], n & 7);
      v = ((0x6400u | (uint32_t)z) << 16) | scales[(size_t)grp * C * 8 + n];
    }
    out[idx] = v;
  }
}

// ------------------------------------------------------------------------------------------ prefill GEMM
// Large M on the repacked layout: the hand-pipelined 128 x 256 kernel (awq_repacked_prefill.hip) for fp16 with g % 128 == 0,
// the compiler-scheduled form of the same decomposition for bf16 / small groups (awq_repacked_ext.hip).  The compiler-scheduled
// fp16 tiles this file used to carry for A/B (<2,2> 256 x 128: 665 TFLOP/s, <4,2> 8 waves: the same, <1,4> 128 x 256: 730
// against 890-960 for the pipelined kernel at 2048 x 4096 x 11008) were removed after round 1; DESIGN.md §5.4 keeps the numbers.
int launch_gemm_repacked_tiled(const GemmArgs& a, const void* packed) {
  if (!repacked_supported(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  if (!repacked_fast(a.K, a.N, a.g, a.dtype) || !pipelined_addressable(a)) return launch_gemm_repacked_tiled_ext(a, packed);     // bf16 / g in {32, 64}
  return launch_gemm_repacked_pipelined(a, packed);
}

// ------------------------------------------------------------------------------------------ host

size_t repacked_bytes(int64_t K, int64_t N, int64_t g) {
  if (K <= 0 || N <= 0 || g <= 0 || K % 128 || K % g) return 0;
  const size_t NG = (size_t)rp_groups(N);
  return NG * (size_t)(K / 128) * 1024 + NG * (size_t)(K / g) * 64;
}

bool repacked_supported(int64_t K, int64_t N, int64_t g, int dtype) {
  return (dtype == AWQ_DTYPE_F16 || dtype == AWQ_DTYPE_BF16) && K > 0 && N > 0 && g > 0 && K % 128 == 0 && (g % 128 == 0 || g == 64 || g == 32) &&
         K % g == 0 && N % 8 == 0 && rp_groups(N) >= 1;
}

bool repacked_fast(int64_t K, int64_t N, int64_t g, int dtype) {
  return dtype == AWQ_DTYPE_F16 && g % 128 == 0 && repacked_supported(K, N, g, dtype);
}

int launch_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* packed, int64_t K, int64_t N, int64_t g,
                  int dtype, hipStream_t stream) {
  if (!repacked_supported(K, N, g, dtype)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(N), C = (int)(N / 8);
  uint32_t* qw_r = (uint32_t*)packed;
  uint32_t* zs_r = qw_r + (size_t)NG * (K / 128) * 256;
  static const int env_simple = lab_env("AWQ_REPACK_SIMPLE", 0);      // lab knob
  if (env_simple)
    hipLaunchKernelGGL(repack_qweight_kernel, dim3(2048), dim3(256), 0, stream, (const uint32_t*)qweight, qw_r, (int)K, C, NG);
  else
    hipLaunchKernelGGL(repack_qweight_tiled_kernel, dim3((unsigned)((K / 128) * ((NG + 31) / 32))), dim3(256), 0, stream, (const uint32_t*)qweight,
                       (u32x4_t*)qw_r, (int)K, C, NG);
  hipLaunchKernelGGL(repack_zs_kernel, dim3(256), dim3(256), 0, stream, (const uint32_t*)qzeros, (const uint16_t*)scales, zs_r,
                     (int)(K / g), C, NG);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// diagnostic only (tools/kbench rstamps): when set, workgroups write 100 MHz-clock stamps into this device buffer
unsigned long long* g_rp_stamp_buffer = nullptr;
extern "C" void awq_debug_set_stamp_buffer(void* p) { g_rp_stamp_buffer = (unsigned long long*)p; }

static int rp_env(const char* name, int dflt) { return lab_env(name, dflt); }

bool gemv_strip_geometry(int64_t K, int64_t N, int* G, int* nwg) {
  (void)K;
  const int NG = rp_groups(N);
  int g = (NG + 255) / 256;
  if (g > route::kOneStripMaxGroups) return false;
  if (g > NG) g = NG;
  *G = g;
  *nwg = (NG + g - 1) / g;
  return true;
}

int launch_gemv_repacked(const GemmArgs& a, const void* packed) {
  if (!repacked_supported(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.M > 16 * kRpMaxMT || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  if (!repacked_fast(a.K, a.N, a.g, a.dtype)) {           // bf16 / g in {32, 64}: the generic kernel, 16 rows per launch
    const size_t eb = 2;
    for (int m0 = 0; m0 < a.M; m0 += 16) {
      GemmArgs c = a;
      c.M = a.M - m0 < 16 ? a.M - m0 : 16;
      c.x = (const char*)a.x + (size_t)m0 * a.ldx * eb;
      c.y = (char*)a.y + (size_t)m0 * a.N * eb;
      const int rc = launch_gemv_repacked_ext(c, packed);
      if (rc) return rc;
    }
    return AWQ_OK;
  }
  const int NG = rp_groups(a.N), KB = a.K / 128;
  const bool two_tiles = a.M > 16;
  const int MT = two_tiles ? 2 : 1;
  // A/B knobs for tools/kbench (0 / -1 = heuristic)
  static const int env_waves = rp_env("AWQ_RP_WAVES", 0), env_nt = rp_env("AWQ_RP_NT", -1), env_g = rp_env("AWQ_RP_G", 0),
                   env_t = rp_env("AWQ_RP_T", -1);
  // Strip width: one strip per CU while that needs <= 8 column groups; wider matrices (N > 32768) run several
  // rounds of 3-group strips on 8-wave workgroups, two of which are resident per CU so one's drain overlaps
  // the other's stream (145 -> 51 us at 8192 x 57344, 4.8 TB/s).
  int G = (NG + 255) / 256;
  bool rounds = false;
  if (G > route::kOneStripMaxGroups) { G = route::kRoundsGroups; rounds = true; }                       // measured 51 (G = 3) / 53 (2) / 54.5 (4) / 60 (1) us
  // (32 rows on wide strips: the reduction scratch, 8 waves x 32 x 16 G floats, goes past 64 KiB — that variant runs one
  // workgroup per CU anyway (190+ registers) and opts in to more of the CU's LDS; narrow rounds would need two resident)
  if (env_g >= 1 && env_g <= kRpMaxG) { G = env_g; rounds = (NG + G - 1) / G > 256; }
  if (G > NG) G = NG;
  const int nwg = (NG + G - 1) / G;
  // 16 waves + non-temporal weight loads for matrices that are streamed from HBM (measured 7.2 vs 8.3 us at
  // 4096 x 11008); small ones (< 12 MB packed, largely L2 / Infinity-Cache resident) run better with 8 waves and
  // default-policy loads (4.2 vs 4.7 us at 4096 x 4096).
  const bool big = (size_t)a.K * a.N / 2 >= route::kStreamedMinBytes;
  const bool nt = env_nt >= 0 ? env_nt != 0 : big;
  int W = env_waves == 16 || env_waves == 8 ? env_waves : (big && !rounds ? 16 : 8);
  if (two_tiles || (size_t)W * a.M * 16 * G * sizeof(float) > 64 * 1024) W = 8;
  // depth: every k-block of a wave in flight at once (straight-line, T = per_wave <= 6) when that fits the
  // register budget, else the double-buffered loop (T = 0); 16 waves fall back to 8 when neither fits
  auto depth = [&](int w) {
    const int pw = (KB + w - 1) / w;
    if (env_t == 0 || rounds) return 0;
    if (pw <= 6 && rp_fits(w, MT, G, pw) && (MT == 1 || pw == 4)) return pw;
    return rp_fits(w, MT, G, 0) ? 0 : -1;
  };
  int T = depth(W);
  // 16 waves, M <= 16: the restructured straight-line kernel (gemv_rp2_kernel: x and zs staged through wave-private LDS by the
  // same few loads, ring of two weight loads per wave, round-robin issue) wherever it has an instantiation; AWQ_RP2=0 /
  // AWQ_RP2_D=0 (every load up front) are A/B knobs for tools/kbench.
  // Ring depth: two loads in flight per wave (4096 x 11008: 6.73 -> 6.37 us at M = 1 against every load up front in
  // tools/gemv_lab; kbench at M = 4: 7.03 vs 7.14, 11008 x 4096 at M = 4: 8.36 vs 8.48).  AWQ_RP2_D=0: everything up front.
  // It also wins on small, largely cache-resident matrices as long as every one of the 16 waves has a k-block (KB >= 16):
  // 4096 x 4096 4.39 -> 4.07 us, 8192 x 1280 5.88 -> 4.72; with fewer k-blocks (1024 x 8192: 3.60 vs 3.81) the 8-wave form stays.
  static const int env_rp2 = rp_env("AWQ_RP2", 1), env_d = rp_env("AWQ_RP2_D", -1);
  const bool rp2_small = !big && !rounds && env_waves == 0 && env_nt < 0 && KB >= route::kRp2MinKBlocks;
  if (env_rp2 && !two_tiles && env_t != 0 && ((W == 16 && nt) || rp2_small)) {
    const int depth = env_d >= 0 ? env_d : 2;
    if (rp2_launch<0>(G, (KB + 15) / 16, a, packed, NG, depth, nwg)) return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
  }
  // what the straight-line form cannot hold (more than 8 staging chunks per lane: many rows on a deep matrix; more than 16 units per
  // wave: deep K, wide strips) runs its loop form, gemv_rp3_kernel (awq_repacked_loop.hip): same ring, x / zs staged per stage.
  // AWQ_RP3=0: the round-1 loop kernel instead (A/B)
  static const int env_rp3 = rp_env("AWQ_RP3", 1);
  // measured (profiles/r03_kbench_rp3_ab.txt): faster than the round-1 loop kernel at 13..16 rows on narrow strips (11008 x 4096 at 16
  // rows 14.4 -> 13.1 us, 8192 x 1280 10.4 -> 8.5), slower below and on wide strips (register pressure: G >= 4 spills at 16 waves)
  if (env_rp3 && !two_tiles && !rounds && KB >= route::kRp2MinKBlocks && env_t != 0 && env_waves == 0 && env_g == 0 && a.M >= route::kRp3MinRows && G <= route::kRp3MaxGroups) {
    const int rc = launch_gemv_repacked_loop(a, packed);
    if (rc != AWQ_ERR_BAD_VARIANT) return rc;
  }
  if (W == 16 && env_waves != 16 && (T < 0 || (T == 0 && (KB + 15) / 16 <= 6))) {   // straight-line did not fit: 8 waves measured better than the 16-wave loop
    W = 8;
    T = depth(W);
  }
  if (T < 0) return AWQ_ERR_BAD_VARIANT;
  const int per_wave = (KB + W - 1) / W;
  const size_t lds = (size_t)W * a.M * 16 * G * sizeof(float);
  if (lds > (size_t)(two_tiles ? kRpMaxLds : 64 * 1024)) return AWQ_ERR_BAD_VARIANT;
  bool launched;
  if (two_tiles) launched = rp_launch_g<8, true, 2>(G, a, packed, NG, per_wave, T, nwg, lds);
  else if (W == 16) launched = nt ? rp_launch_g<16, true, 1>(G, a, packed, NG, per_wave, T, nwg, lds) : rp_launch_g<16, false, 1>(G, a, packed, NG, per_wave, T, nwg, lds);
  else launched = nt ? rp_launch_g<8, true, 1>(G, a, packed, NG, per_wave, T, nwg, lds) : rp_launch_g<8, false, 1>(G, a, packed, NG, per_wave, T, nwg, lds);
  if (!launched) return AWQ_ERR_BAD_VARIANT;             // no such instantiation (e.g. a combination forced through the AWQ_RP_* knobs): nothing ran
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
