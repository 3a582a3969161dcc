// Decode GEMV on the MFMA-fragment-major layout for NARROW matrices at 9 .. 32 rows: wide strips, K split across workgroups.
//
// gemv_repacked_kernel / gemv_rp2_kernel give every workgroup one strip of columns and ALL of K, so every workgroup reads all of x:
// with N = 4096 (256 strips of 16 columns to fill the chip) and 32 rows that is 256 x 704 KB = 180 MB of L2 -> CU traffic behind
// 22.5 MB of weights (11008 x 4096 at M = 32: 23.5 us).  Here a workgroup owns G column groups (G = 4: 64 columns) and
// one of S slices of K: the x traffic drops by G (each workgroup reads M x K / S of it), the weights are still streamed once, and the
// S partial tiles of a strip meet through the workspace: fp32 partials written through to L2, one ticket per strip, the last
// workgroup to arrive adds them in slice order (deterministic; the hand-off of awq_gemm_skinny.hip / decode_attention_kernel — no
// spinning, the counters are left at zero).  Same per-element arithmetic as the other GEMV kernels; the fp32 summation ORDER differs
// (per slice, then over slices), so results agree with them to fp32 rounding of the sums, not bit for bit.
#include <cstdlib>

#include "awq_repacked_gemv.h"

namespace awq {

constexpr int kRpsWaves = 8;
constexpr size_t kRpsHead = 4096;                      // [<= 1024 strips] tickets at the head of the workspace (zero between calls)

template <int G, int MT>
__global__ __launch_bounds__(kRpsWaves * 64, 2) void gemv_rps_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                     const u32x4_t* __restrict__ qw_r, const uint32_t* __restrict__ zs_r,
                                                                     const void* __restrict__ bias, void* __restrict__ y, int M, int K,
                                                                     int N, int g, int NG, int kb_slice, float* __restrict__ ws_part,
                                                                     unsigned* __restrict__ ws_cnt, unsigned ws_part_bytes) {
  constexpr int W = kRpsWaves, SW = 16 * G;
  extern __shared__ __attribute__((aligned(16))) float red[];    // [W][M][16 G]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g;
  const int strip = blockIdx.x, slice = blockIdx.y, S = (int)gridDim.y;
  int cg0 = strip * G;
  if (cg0 + G > NG) cg0 = NG - G;                      // last strip overlaps its neighbour (same values written twice)
  int xr[kRpMaxMT];
#pragma unroll
  for (int mt = 0; mt < kRpMaxMT; ++mt) xr[mt] = mt * 16 + r < M ? mt * 16 + r : M - 1;

  float4_t acc[kRpMaxMT][kRpMaxG];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < G; ++c) acc[mt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // this workgroup's k-blocks [kb_lo, kb_hi), dealt to the waves in contiguous runs
  const int kb_lo = slice * kb_slice;
  int kb_hi = kb_lo + kb_slice;
  if (kb_hi > KB) kb_hi = KB;
  const int per_wave = (kb_hi - kb_lo + W - 1) / W;
  const int kb_begin = kb_lo + wave * per_wave;
  int kb_end = kb_begin + per_wave;
  if (kb_end > kb_hi) kb_end = kb_hi;

  {
    // steady state has no branch between a load and its use; only the prologue and the <= 3-block tail are conditional
    RpBlock A, B;
    int kb = kb_begin;
    if (kb < kb_end) rp_load<G, true, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
    if (kb + 1 < kb_end) rp_load<G, true, MT>(B, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 1, lane, xr);
    while (kb + 3 < kb_end) {
      prio_by_progress(kb - kb_begin, per_wave);
      rp_compute<G, MT>(A, acc);
      rp_load<G, true, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
      rp_compute<G, MT>(B, acc);
      rp_load<G, true, MT>(B, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 3, lane, xr);
      kb += 2;
    }
    prio_by_progress(kb - kb_begin, per_wave);
    if (kb < kb_end) rp_compute<G, MT>(A, acc);
    if (kb + 2 < kb_end) rp_load<G, true, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
    if (kb + 1 < kb_end) rp_compute<G, MT>(B, acc);
    if (kb + 2 < kb_end) rp_compute<G, MT>(A, acc);
  }
  __builtin_amdgcn_s_setprio(0);

  // D[m = 16 mt + 4 q + i][n = r] per column group -> LDS, summed over the waves in fixed order
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < G; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = mt * 16 + 4 * q + i;
        if (m < M) red[((size_t)wave * M + m) * SW + c * 16 + r] = acc[mt][c][i];
      }
  __syncthreads();

  // this slice's partial tile [M][16 G] -> workspace (write-through), then one ticket per workgroup
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ws_part, 0, (int)ws_part_bytes, 0x00020000);
  const unsigned tile_bytes = (unsigned)(M * SW) * 4u;
  const unsigned pbase = ((unsigned)strip * (unsigned)S + (unsigned)slice) * tile_bytes;
  for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
    float v = red[idx];                                // [wave 0][m][col]
#pragma unroll
    for (int w = 1; w < W; ++w) v += red[(size_t)w * M * SW + idx];
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, pbase + (unsigned)idx * 4u, 0, 16);
  }
  __shared__ unsigned ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&ws_cnt[strip], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != (unsigned)(S - 1)) return;
  if (threadIdx.x == 0) __hip_atomic_store(&ws_cnt[strip], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  const unsigned sbase = (unsigned)strip * (unsigned)S * tile_bytes;
  for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
    const int m = idx / SW, col = idx - m * SW;
    const int n = cg0 * 16 + col;
    float v = 0.f;
    for (int s2 = 0; s2 < S; ++s2)                     // slice order: deterministic
      v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, sbase + (unsigned)s2 * tile_bytes + (unsigned)idx * 4u, 0, 16));
    if (n < N) store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, v, bias, n);
  }
}

// Strip width and slice count.  Narrow matrices only (the one-strip-per-CU kernels already have few, wide strips when N is large).
bool rps_plan(const GemmArgs& a, int* G_out, int* S_out, int* kb_slice_out) {
  static const int env_on = lab_env("AWQ_RPS", 1);
  static const int env_g = lab_env("AWQ_RPS_G", 0), env_s = lab_env("AWQ_RPS_S", 0);
  static const int env_minm = lab_env("AWQ_RPS_MINM", 0);
  if (!env_on || !repacked_fast(a.K, a.N, a.g, a.dtype) || a.M > 32 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return false;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  // measured (tools/kbench rgemm, us, one-strip kernels -> this one): 11008 x 4096 at 32 / 16 rows 23.2 -> 17.3 / 14.5 -> 11.5;
  // 4096 x 4096 at 32 rows 10.6 -> 9.9, at 16 rows 6.7 -> 7.3 (not taken); more than two slices or four-group strips lose again
  // (also measured: 8192 x 1280 at 16 / 32 rows 10.4 -> 8.2 / 17.4 -> 14.1; no gain or a loss at K = 4096 (4096 x 4096, 4096 x 2560),
  // at N = 8192 (512 workgroups: two rounds) and with 3584 x 8192 — profiles/r02_kbench_splitk_ab.txt): deep and narrow only
  const int minm = env_minm ? env_minm : route::kSplitKMinRows;
  if (a.M < minm || NG < 8 || (!env_minm && (KB < route::kRpsMinKBlocks || NG > route::kRpsMaxGroups))) return false;
  const int G = env_g ? env_g : route::kRpsStripGroups;
  if (G != 2 && G != 4) return false;
  const int strips = (NG + G - 1) / G;
  // two slices once the strips alone fill half the chip; more only for very narrow matrices (at least one k-block per wave and slice)
  int S = env_s ? env_s : 2;
  if (!env_s) while (S < 8 && strips * S < 256 && KB / (S * 2) >= kRpsWaves) S *= 2;
  if (S < 2 || S > 16 || strips > 1024) return false;
  const int kb_slice = (KB + S - 1) / S;
  const size_t need = kRpsHead + (size_t)strips * S * a.M * 16 * G * sizeof(float);
  if (!a.workspace || (((uintptr_t)a.workspace) & 15) || a.workspace_bytes < need || need >= (size_t(1) << 31)) return false;
  *G_out = G; *S_out = (KB + kb_slice - 1) / kb_slice; *kb_slice_out = kb_slice;
  return *S_out >= 2;
}

size_t rps_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < 9 || M > 32 || K < 2048 || N > 8192 || N < 128) return 0;
  const size_t NG = (size_t)rp_groups(N);
  return kRpsHead + (NG + 3) * 16 * sizeof(float) * 16 * (size_t)M;       // strips x 16 G columns, <= 16 slices, M rows of fp32
}

template <int G>
static int rps_launch(const GemmArgs& a, const void* packed, int NG, int S, int kb_slice) {
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const int strips = (NG + G - 1) / G;
  const size_t lds = (size_t)kRpsWaves * a.M * 16 * G * sizeof(float);
  unsigned* cnt = (unsigned*)a.workspace;
  float* part = (float*)((char*)a.workspace + kRpsHead);
  const unsigned part_bytes = (unsigned)((size_t)strips * S * a.M * 16 * G * sizeof(float));
  static unsigned long long opted1[2] = {0ull, 0ull}, opted2[2] = {0ull, 0ull};
  if (a.M > 16) {
    if (lds > 64 * 1024) return AWQ_ERR_BAD_VARIANT;
    if (!opt_in_dynamic_lds((const void*)gemv_rps_kernel<G, 2>, (int)lds, opted2)) return AWQ_ERR_LAUNCH;
    hipLaunchKernelGGL((gemv_rps_kernel<G, 2>), dim3(strips, S), dim3(kRpsWaves * 64), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r,
                       a.bias, a.y, a.M, a.K, a.N, a.g, NG, kb_slice, part, cnt, part_bytes);
  } else {
    if (!opt_in_dynamic_lds((const void*)gemv_rps_kernel<G, 1>, (int)lds, opted1)) return AWQ_ERR_LAUNCH;
    hipLaunchKernelGGL((gemv_rps_kernel<G, 1>), dim3(strips, S), dim3(kRpsWaves * 64), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r,
                       a.bias, a.y, a.M, a.K, a.N, a.g, NG, kb_slice, part, cnt, part_bytes);
  }
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int launch_gemv_repacked_splitk(const GemmArgs& a, const void* packed) {
  int G = 0, S = 0, kb_slice = 0;
  if (!rps_plan(a, &G, &S, &kb_slice)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  return G == 4 ? rps_launch<4>(a, packed, NG, S, kb_slice) : rps_launch<2>(a, packed, NG, S, kb_slice);
}

}  // namespace awq
