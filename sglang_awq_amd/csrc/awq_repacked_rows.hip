// Decode GEMV on the MFMA-fragment-major layout for 17 .. 32 rows on WIDE matrices: x goes through LDS once per workgroup.
//
// gemv_repacked_kernel<MT = 2> lets every wave fetch the x fragments of its own k-blocks straight from L2: 8 dwordx4 loads of x next
// to G loads of weights per k-block and wave, so more than half of the CU's load slots carry activations it shares with nobody
// (4096 x 22016 at 32 rows: 22 us for 45 MB of weights).  Here the G x KS waves of a workgroup walk K together: iteration `it`
// covers k-blocks [it KS, it KS + KS); the x tile of those k-blocks (32 rows x 128 KS halves) is fetched ONCE by the whole
// workgroup into double-buffered, XOR-swizzled LDS and every wave reads its A fragments from there; wave (c, ks) streams the
// weights of column group cg0 + c for k-block it KS + ks through a register ring (RING loads in flight) and keeps MT x 4
// accumulators; one barrier per iteration.  With KS > 1 the KS partial sums of a column group meet in LDS at the end (fixed order).
// Same per-element arithmetic as the other GEMV kernels; the fp32 summation order differs (k-blocks interleaved over KS waves).
#include <cstdlib>

#include "awq_prefill_common.h"

namespace awq {

constexpr int kRtRing = 4;

// x tile [32 rows][KS * 128 halves]: row-major, 16-byte chunks of a row's k-block XOR-swizzled with the row index
template <int KS>
__device__ __forceinline__ int rt_off(int row, int kblk, int chunk) { return row * (KS * 256) + kblk * 256 + ((chunk ^ (row & 15)) << 4); }

template <int G, int KS, int MT>
__global__ __launch_bounds__(G * KS * 64) void gemv_rt_kernel(const uint16_t* __restrict__ x, int64_t ldx, const u32x4_t* __restrict__ qw_r,
                                                             const uint32_t* __restrict__ zs_r, const void* __restrict__ bias,
                                                             void* __restrict__ y, int M, int K, int N, int g, int NG) {
  constexpr int W = G * KS, NT = W * 64, ROWS = 16 * MT, TILE = ROWS * KS * 256;     // bytes of one x tile
  constexpr int CHUNKS = ROWS * KS * 16, XL = (CHUNKS + NT - 1) / NT;                // 16-byte chunks per tile / per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];               // 2 x tiles, then the reduction scratch
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = wave / KS, ks = wave - c * KS;
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g, kpg = g / 128;
  const int IT = (KB + KS - 1) / KS;
  int cg0 = blockIdx.x * G;
  if (cg0 + G > NG) cg0 = NG - G;                      // last strip overlaps its neighbour (same values written twice)

  // x tile chunks of this thread: chunk id = tid + NT i -> (row, k-block of the iteration, 16-byte chunk)
  // (exact extent: the clamped k-blocks of a ragged last iteration may point past the last row's end; out-of-range buffer loads return 0)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(((size_t)(M - 1) * ldx + K) * 2), kPfRsrcFlags);
  uint32_t xoff[XL];
  int xdst[XL];
  bool xok[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int id = tid + NT * i;
    xok[i] = id < CHUNKS;
    const int idc = xok[i] ? id : 0;
    const int row = idc / (KS * 16), rem = idc - row * (KS * 16), kblk = rem >> 4, chunk = rem & 15;
    const int mr = row < M ? row : M - 1;              // rows past M: a valid row's data, results never stored
    xoff[i] = (uint32_t)((size_t)mr * ldx + kblk * 128 + chunk * 8) * 2u;
    xdst[i] = rt_off<KS>(row, kblk, chunk);
  }
  // x tiles travel kRtRing iterations ahead of their use, in a ring of register sets (an L2 round trip is longer than an iteration)
  u32x4_t xs[kRtRing][XL];
  auto load_x = [&](int it, int slot) {                // (a k-block past K reads past the row: harmless, its weights are skipped)
    it = it < IT ? it : IT - 1;
#pragma unroll
    for (int i = 0; i < XL; ++i) xs[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[i], it * (KS * 256), 0);
  };
  auto store_x = [&](int buf, int slot) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      if (xok[i]) *(u32x4_t*)(lds + buf * TILE + xdst[i]) = xs[slot][i];
  };

  // weights of (column group cg0 + c, k-block it KS + ks): ring of kRtRing loads
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)(qw_r + (size_t)(cg0 + c) * KB * 64), 0, KB * 1024, kPfRsrcFlags);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(zs_r + (size_t)(cg0 + c) * groups * 16), 0, groups * 64, kPfRsrcFlags);
  const int lane16 = lane * 16, r4 = r * 4;
  u32x4_t wr[kRtRing];
  uint32_t zr[kRtRing];
  auto load_w = [&](int it, int slot) {
    int kb = it * KS + ks;
    kb = kb < KB ? kb : KB - 1;                        // clamped; a k-block past K is not computed
    wr[slot] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, kb * 1024, 0);
    zr[slot] = __builtin_amdgcn_raw_buffer_load_b32(rz, r4, (kb / kpg) * 64, 0);
  };

  float4_t acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // prologue: x tiles 0 .. kRtRing - 1 requested, tile 0 -> LDS (its slot then takes tile kRtRing), the first kRtRing weight loads
#pragma unroll
  for (int s = 0; s < kRtRing; ++s) load_x(s, s);
#pragma unroll
  for (int s = 0; s < kRtRing; ++s) load_w(s < IT ? s : IT - 1, s);
  store_x(0, 0);
  load_x(kRtRing, 0);
  __syncthreads();

  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  for (int it0 = 0; it0 < IT; it0 += kRtRing) {
#pragma unroll
    for (int s = 0; s < kRtRing; ++s) {                // ring slot = it % kRtRing, compile-time
      const int it = it0 + s;
      if (it < IT) {                                   // (workgroup-uniform)
        const unsigned char* Xb = lds + (it & 1) * TILE;
        if (it * KS + ks < KB) {                       // (wave-uniform: a ragged last iteration)
          const uint32_t zs = zr[s];
          const u32x4_t w = wr[s];
          const half2_t s2 = as_h2(pack_lo16(zs, zs));
          const half2_t z1024 = as_h2(pack_hi16(zs, zs));
          const half2_t z64 = z1024 - c960;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const u32x4_t frag = rp_dequant(w[d], z1024, z64, s2);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const u32x4_t a = *(const u32x4_t*)(Xb + rt_off<KS>(mt * 16 + r, ks, d * 4 + q));
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, frag), acc[mt], 0, 0, 0);
            }
          }
        }
        load_w(it + kRtRing < IT ? it + kRtRing : IT - 1, s);      // this slot's next occupant
        store_x((it + 1) & 1, (s + 1) % kRtRing);      // tile it + 1 (requested kRtRing iterations ago) -> the other buffer
        load_x(it + 1 + kRtRing, (s + 1) % kRtRing);
        __syncthreads();
      }
    }
  }

  // partial sums of the KS waves of a column group -> LDS, summed in ks order; D[m = 16 mt + 4 q + i][n = r]
  float* red = (float*)(lds + 2 * TILE);               // [W][ROWS][16]
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[(wave * ROWS + mt * 16 + 4 * q + i) * 16 + r] = acc[mt][i];
  __syncthreads();
  for (int idx = tid; idx < G * ROWS * 16; idx += NT) {
    const int cc = idx / (ROWS * 16), rem = idx - cc * (ROWS * 16), m = rem >> 4, col = rem & 15;
    const int n = (cg0 + cc) * 16 + col;
    if (m >= M || n >= N) continue;
    float v = red[((cc * KS) * ROWS + m) * 16 + col];
#pragma unroll
    for (int k2 = 1; k2 < KS; ++k2) v += red[((cc * KS + k2) * ROWS + m) * 16 + col];
    store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, v, bias, n);
  }
}

template <int G, int KS, int MT>
static int rt_go(const GemmArgs& a, const void* packed, int NG) {
  constexpr int W = G * KS, ROWS = 16 * MT;
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const size_t lds = 2 * (size_t)ROWS * KS * 256 + (size_t)W * ROWS * 16 * sizeof(float);
  const int nwg = (NG + G - 1) / G;
  hipLaunchKernelGGL((gemv_rt_kernel<G, KS, MT>), dim3(nwg), dim3(W * 64), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias,
                     a.y, a.M, a.K, a.N, a.g, NG);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// 17 .. 32 rows, wide matrices (at least 512 column groups), fp16, group size a multiple of 128.
int launch_gemv_repacked_rows(const GemmArgs& a, const void* packed) {
  static const int env_on = getenv("AWQ_RT") ? atoi(getenv("AWQ_RT")) : 1;
  static const int env_g = getenv("AWQ_RT_G") ? atoi(getenv("AWQ_RT_G")) : 0, env_ks = getenv("AWQ_RT_KS") ? atoi(getenv("AWQ_RT_KS")) : 0;
  static const int env_minm = getenv("AWQ_RT_MINM") ? atoi(getenv("AWQ_RT_MINM")) : 17;
  if (!env_on || !repacked_fast(a.K, a.N, a.g, a.dtype) || a.M < env_minm || a.M > 32 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  if ((int64_t)a.M * a.ldx * 2 >= (int64_t(1) << 31) || (int64_t)(a.K / 128) * 1024 >= (int64_t(1) << 31)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  if (NG < 512 || KB < 8) return AWQ_ERR_BAD_VARIANT;
  // strip width: about one workgroup per CU; waves per workgroup = G KS >= 6
  int G = env_g, KS = env_ks;
  if (!G) G = NG >= 1280 ? 6 : (NG >= 896 ? 4 : 3);
  if (!KS) KS = G >= 6 ? 1 : 2;
  const bool two = a.M > 16;
#define RT_CASE(GG, KK)                                                                      \
  if (G == GG && KS == KK) return two ? rt_go<GG, KK, 2>(a, packed, NG) : rt_go<GG, KK, 1>(a, packed, NG);
  RT_CASE(6, 1) RT_CASE(8, 1) RT_CASE(3, 2) RT_CASE(4, 2) RT_CASE(6, 2) RT_CASE(2, 4) RT_CASE(3, 4)
#undef RT_CASE
  return AWQ_ERR_BAD_VARIANT;
}

}  // namespace awq
