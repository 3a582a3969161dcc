// The MFMA-fragment-major layout beyond fp16 / group sizes that are multiples of 128: bf16 scales and
// activations, and the small groups the reference's Triton path accepts (g in {32, 64}: awq_triton.py:10,250).
//
// Same layout as awq_repacked.hip — qw_r[NG][K / 128][64 lanes][4 dwords], one dword = the 8 int4 of one column over 8
// rows = one MFMA B fragment — with
//   zs_r[NG][K / g][16]: one word per (column, quantisation group): low half = the scale's 16 bits (fp16 or bf16), high
//                        half = fp16(1024 + zero).  A 128-row k-block spans GS = 128 / g groups when g < 128 (each k-step
//                        of 32 rows then has its own scale / zero), else all four k-steps share one word.
// Numerics are the reference's in either dtype (awq_kernel.cu:126-184 / awq_triton.py:103-104): (q - z) is an exact small
// integer; fp16: one v_pk_mul_f16 by the scale; bf16: exact fp32 product (an integer of < 5 bits times an 8-bit mantissa),
// one round-to-nearest-even conversion (v_cvt_pk_bf16_f32).  fp32 accumulation in the MFMA, one rounding of the sum.
//
// Kernels (less specialised than the fp16 / g % 128 ones they stand beside — runtime k loop, fragments of x straight from
// global memory): gemv_rpx_kernel for M <= 16 per launch (strip of G column groups per workgroup, K over 16 waves, partial
// sums added in wave order through LDS: deterministic), gemm_rpx_tiled_kernel for large M (128 x 256 tiles, B fragments
// from registers, x through double-buffered swizzled LDS).
#include "awq_repacked_gemv.h"

namespace awq {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ float4_t rpx_mfma(const u32x4_t& a, const u32x4_t& b, float4_t c) {
  if constexpr (DT == AWQ_DTYPE_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// one dword (8 weights of one column) + its (scale | 1024 + zero) word -> the 4 packed k-pairs of an MFMA B fragment
template <int DT>
__device__ __forceinline__ u32x4_t rpx_dequant(uint32_t w, uint32_t zs) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  const uint32_t magic = kMagicF16;
  const half2_t zh = as_h2(zs);
  const half2_t z1024 = __builtin_shufflevector(zh, zh, 1, 1);
  const half2_t z64 = z1024 - c960;                                     // exact: (1024 + z) - 960
  const uint32_t w8 = w >> 8;
  half2_t d[4];                                                         // (q - z), exact small integers in fp16
  d[0] = as_h2(and_or(w, kLoNib, magic)) - z1024;
  d[1] = __builtin_elementwise_fma(as_h2(and_or(w, kHiNib, magic)), sixteenth, -z64);
  d[2] = as_h2(and_or(w8, kLoNib, magic)) - z1024;
  d[3] = __builtin_elementwise_fma(as_h2(and_or(w8, kHiNib, magic)), sixteenth, -z64);
  u32x4_t f;
  if constexpr (DT == AWQ_DTYPE_F16) {
    const half2_t s2 = __builtin_shufflevector(zh, zh, 0, 0);
#pragma unroll
    for (int t = 0; t < 4; ++t) f[t] = as_u32(d[t] * s2);
  } else {
    const float s = __builtin_bit_cast(float, zs << 16);                // bf16 bits -> fp32
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float2_t p = {(float)d[t][0] * s, (float)d[t][1] * s};       // exact products
      f[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(p, bf16x2_t));   // one RNE rounding each
    }
  }
  return f;
}

// ------------------------------------------------------------------------------------------------ decode GEMV, M <= 16
// GS = scale words per k-block (1: g % 128 == 0; 2: g = 64; 4: g = 32).
template <int G, int DT, int GS>
__global__ __launch_bounds__(1024) void gemv_rpx_kernel(const uint16_t* __restrict__ x, int64_t ldx, const u32x4_t* __restrict__ qw_r,
                                                        const uint32_t* __restrict__ zs_r, const void* __restrict__ bias,
                                                        void* __restrict__ y, int M, int K, int N, int g, int NG, int T) {
  constexpr int W = 16;
  extern __shared__ __attribute__((aligned(16))) float red[];           // [W][M][16 G]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g;
  int cg0 = blockIdx.x * G;
  if (cg0 + G > NG) cg0 = NG - G;                                        // last strip overlaps its neighbour (same values written twice)
  const int xr = r < M ? r : M - 1;
  const uint16_t* xrow = x + (size_t)xr * ldx + q * 8;

  float4_t acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c) acc[c] = (float4_t){0.f, 0.f, 0.f, 0.f};

  u32x4_t wv[2][G], xa[2][4];
  uint32_t zs[2][G][GS];
  auto load = [&](int buf, int kb) {                                     // kb < KB
#pragma unroll
    for (int c = 0; c < G; ++c) wv[buf][c] = __builtin_nontemporal_load(qw_r + ((size_t)(cg0 + c) * KB + kb) * 64 + lane);
    const int grp = GS > 1 ? kb * GS : (kb * 128) / g;
#pragma unroll
    for (int c = 0; c < G; ++c)
#pragma unroll
      for (int s = 0; s < GS; ++s) zs[buf][c][s] = zs_r[((size_t)(cg0 + c) * groups + grp + s) * 16 + r];
#pragma unroll
    for (int d = 0; d < 4; ++d) xa[buf][d] = *(const u32x4_t*)(xrow + (size_t)kb * 128 + d * 32);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int c = 0; c < G; ++c)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        acc[c] = rpx_mfma<DT>(xa[buf][d], rpx_dequant<DT>(wv[buf][c][d], zs[buf][c][(d * GS) / 4]), acc[c]);
  };
  const int kb_begin = wave * T;
  int kb_end = kb_begin + T;
  if (kb_end > KB) kb_end = KB;
  // double-buffered over the wave's k-blocks; the loop body has no branch between a load and its use
  int kb = kb_begin;
  if (kb < kb_end) load(0, kb);
  while (kb + 2 < kb_end) {
    load(1, kb + 1);
    compute(0);
    load(0, kb + 2);
    compute(1);
    kb += 2;
  }
  if (kb + 1 < kb_end) {
    load(1, kb + 1);
    compute(0);
    compute(1);
  } else if (kb < kb_end) {
    compute(0);
  }

  const int SW = 16 * G;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < M) {                                                         // wave-uniform
      const int m = 4 * q + i;
#pragma unroll
      for (int c = 0; c < G; ++c)
        if (m < M) red[((size_t)wave * M + m) * SW + c * 16 + r] = acc[c][i];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
    const int m = idx / SW, col = idx - m * SW;
    const int n = cg0 * 16 + col;
    if (n >= N) continue;
    float v = red[(size_t)m * SW + col];
#pragma unroll
    for (int w = 1; w < W; ++w) v += red[((size_t)w * M + m) * SW + col];
    store_output<DT>(y, (size_t)m * N + n, v, bias, n);
  }
}

template <int G, int DT>
static bool rpx_launch_gs(int gs, const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int T, int nwg, size_t lds) {
#define RPX_GO(GS_)                                                                                                             \
  do {                                                                                                                          \
    auto kern = gemv_rpx_kernel<G, DT, GS_>;                                                                                    \
    static unsigned long long opted[2] = {0ull, 0ull};                                                                          \
    if (lds > 64 * 1024 && !opt_in_dynamic_lds((const void*)kern, kRpMaxLds, opted)) return false;                               \
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias, a.y, a.M, a.K, \
                       a.N, a.g, NG, T);                                                                                        \
    return true;                                                                                                                \
  } while (0)
  if (gs == 1) RPX_GO(1);
  if (gs == 2) RPX_GO(2);
  if (gs == 4) RPX_GO(4);
#undef RPX_GO
  return false;
}

template <int DT>
static bool rpx_launch_g(int G, int gs, const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int T, int nwg, size_t lds) {
  switch (G) {
    case 1: return rpx_launch_gs<1, DT>(gs, a, qw_r, zs_r, NG, T, nwg, lds);
    case 2: return rpx_launch_gs<2, DT>(gs, a, qw_r, zs_r, NG, T, nwg, lds);
    case 3: return rpx_launch_gs<3, DT>(gs, a, qw_r, zs_r, NG, T, nwg, lds);
    case 4: return rpx_launch_gs<4, DT>(gs, a, qw_r, zs_r, NG, T, nwg, lds);
    default: return false;
  }
}

// M <= 16 rows per launch; wider batches are cut into 16-row passes by the caller (launch_gemv_repacked)
int launch_gemv_repacked_ext(const GemmArgs& a, const void* packed) {
  if (!repacked_supported(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.M > 16 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  const int gs = a.g >= 128 ? 1 : 128 / a.g;
  // strips of <= 4 column groups (the fragment registers of two k-blocks per wave): one strip per CU up to N = 16384, rounds beyond
  int G = (NG + 255) / 256;
  if (G > 4) G = 4;
  if (G > NG) G = NG;
  const int nwg = (NG + G - 1) / G;
  const int T = (KB + 15) / 16;
  const size_t lds = (size_t)16 * a.M * 16 * G * sizeof(float);
  if (lds > (size_t)kRpMaxLds) return AWQ_ERR_BAD_VARIANT;
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * KB * 256;
  const bool ok = a.dtype == AWQ_DTYPE_F16 ? rpx_launch_g<AWQ_DTYPE_F16>(G, gs, a, qw_r, zs_r, NG, T, nwg, lds)
                                           : rpx_launch_g<AWQ_DTYPE_BF16>(G, gs, a, qw_r, zs_r, NG, T, nwg, lds);
  if (!ok) return AWQ_ERR_BAD_VARIANT;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------ large M: 128 x 256 tiles
// The decomposition of gemm_repacked_tiled_kernel<1, 4> (awq_repacked.hip), compiler-scheduled, for either dtype and any
// supported group size: 4 waves of 128 x 64, B fragments straight from the packed dwords in registers, x through
// double-buffered XOR-swizzled LDS, one barrier per 128-deep k-block, XCD-aware tile order.
__device__ __forceinline__ int rpx_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }   // [rows][128 halves]

template <int DT, int GS>
__global__ __launch_bounds__(256, 1) void gemm_rpx_tiled_kernel(const uint16_t* __restrict__ x, int64_t ldx, const u32x4_t* __restrict__ qw_r,
                                                                 const uint32_t* __restrict__ zs_r, const void* __restrict__ bias,
                                                                 void* __restrict__ y, int M, int K, int N, int g, int NG, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) unsigned char As[];      // 2 x 32 KiB
  constexpr int MI = 8, BMt = 128, BNt = 256, NT_ = 256, AL = BMt * 16 / NT_;
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g;
  const int nwg = nbx * nby, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int bm = (logical / nbx) * BMt;
  const int bn = (logical % nbx) * BNt;
  int cg[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = (bn + wn * 64) / 16 + j;
    cg[j] = c < NG ? c : NG - 1;                       // clamped: columns >= N are never stored
  }
  u32x4_t w_cur[4], w_nxt[4], a_st[AL];
  uint32_t zs_cur[4][GS], zs_nxt[4][GS];
  auto load_a = [&](int kb) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int c = tid + NT_ * i;
      const int row = c >> 4, chunk = c & 15;
      const int m = bm + row < M ? bm + row : M - 1;
      a_st[i] = *(const u32x4_t*)(x + (size_t)m * ldx + kb * 128 + chunk * 8);
    }
  };
  auto store_a = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int c = tid + NT_ * i;
      *(u32x4_t*)(As + buf * (BMt * 256) + rpx_off(c >> 4, c & 15)) = a_st[i];
    }
  };
  auto load_b = [&](u32x4_t (&w)[4], uint32_t (&zs)[4][GS], int kb) {
    const int grp = GS > 1 ? kb * GS : (kb * 128) / g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = qw_r[((size_t)cg[j] * KB + kb) * 64 + lane];
#pragma unroll
      for (int s = 0; s < GS; ++s) zs[j][s] = zs_r[((size_t)cg[j] * groups + grp + s) * 16 + r];
    }
  };
  float4_t acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mi][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  load_a(0);
  load_b(w_cur, zs_cur, 0);
  store_a(0);
  __syncthreads();
  for (int kb = 0; kb < KB; ++kb) {
    const int nxt = kb + 1 < KB ? kb + 1 : kb;         // clamped, unconditional prefetch (no branch between load and use)
    load_a(nxt);
    load_b(w_nxt, zs_nxt, nxt);
    __builtin_amdgcn_sched_barrier(0);                 // or hipcc sinks these loads to the end of the body, right in front of their use
    const unsigned char* Ab = As + (kb & 1) * (BMt * 256);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      u32x4_t af[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = *(const u32x4_t*)(Ab + rpx_off(mi * 16 + r, d * 4 + q));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4_t frag = rpx_dequant<DT>(w_cur[j][d], zs_cur[j][(d * GS) / 4]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][j] = rpx_mfma<DT>(af[mi], frag, acc[mi][j]);
      }
    }
    store_a((kb + 1) & 1);                             // nobody reads that buffer before the barrier
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w_cur[j] = w_nxt[j];
#pragma unroll
      for (int s = 0; s < GS; ++s) zs_cur[j][s] = zs_nxt[j][s];
    }
    __syncthreads();
  }
  // output tile through LDS (as awq_repacked_prefill.hip does): rounded (+ bias) as store_output<DT> would, parked wave-privately in the x
  // buffers (every wave is past the loop's last barrier: nobody reads them any more), stored as 16 bytes per lane — 128-byte row runs
  // instead of 4 rows x 32 bytes per store
  if constexpr (DT == 0 || DT == 1) {
    uint16_t* const tw = (uint16_t*)As + (size_t)wn * (64 * 64);
    const bool n_ok8 = (N % 8) == 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = bn + wn * 64 + j * 16 + r;
            tw[(mi * 16 + 4 * q + i) * 64 + j * 16 + r] = output_bits16<DT>(acc[half * 4 + mi][j][i], n < N ? bias : nullptr, n);
          }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int f = (p * 64 + lane) * 8;
        const int row = f >> 6, col = f & 63;
        const int m = bm + half * 64 + row, n = bn + wn * 64 + col;
        if (m < M && n < N) {
          if (n_ok8) {
            *(u32x4_t*)((uint16_t*)y + (size_t)m * N + n) = *(const u32x4_t*)(tw + f);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < N) ((uint16_t*)y)[(size_t)m * N + n + e] = tw[f + e];
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = bm + mi * 16 + 4 * q + i;
        if (m < M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = bn + wn * 64 + j * 16 + r;
            if (n < N) store_output<DT>(y, (size_t)m * N + n, acc[mi][j][i], bias, n);
          }
        }
      }
  }
}

template <int DT, int GS>
static bool rpx_tiled_go(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG) {
  const int nbx = (a.N + 255) / 256, nby = (a.M + 127) / 128;
  const size_t lds = 2 * 128 * 256;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_rpx_tiled_kernel<DT, GS>, (int)lds, opted)) return false;
  hipLaunchKernelGGL((gemm_rpx_tiled_kernel<DT, GS>), dim3(nbx * nby), dim3(256), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias,
                     a.y, a.M, a.K, a.N, a.g, NG, nbx, nby);
  return true;
}

int launch_gemm_repacked_tiled_ext(const GemmArgs& a, const void* packed) {
  if (!repacked_supported(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const int gs = a.g >= 128 ? 1 : 128 / a.g;
  bool ok = false;
  if (a.dtype == AWQ_DTYPE_F16) ok = gs == 1 ? rpx_tiled_go<AWQ_DTYPE_F16, 1>(a, qw_r, zs_r, NG) : gs == 2 ? rpx_tiled_go<AWQ_DTYPE_F16, 2>(a, qw_r, zs_r, NG)
                                                                                                         : rpx_tiled_go<AWQ_DTYPE_F16, 4>(a, qw_r, zs_r, NG);
  else ok = gs == 1 ? rpx_tiled_go<AWQ_DTYPE_BF16, 1>(a, qw_r, zs_r, NG) : gs == 2 ? rpx_tiled_go<AWQ_DTYPE_BF16, 2>(a, qw_r, zs_r, NG)
                                                                                    : rpx_tiled_go<AWQ_DTYPE_BF16, 4>(a, qw_r, zs_r, NG);
  if (!ok) return AWQ_ERR_LAUNCH;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
