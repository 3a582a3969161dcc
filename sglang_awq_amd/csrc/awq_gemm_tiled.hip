// Tiled fused AWQ GEMM for prefill shapes (large M) on gfx950 — MFMA-bound.
//
// The reference's apply() (awq.py:446-447) writes the whole dequantised weight to HBM (90 MB at
// 4096 x 11008) and reads it back in a vendor GEMM.  Here the packed tile is dequantised once per
// workgroup straight into the LDS image the MFMA fragments are read from, so HBM only ever sees
// the 4-bit weights.
//
//   workgroup tile  128 (M) x 128 (N), K step 128 (= one g128 group), 4 waves as 2 x 2, each 64 x 64
//                   = 4 x 4 tiles of v_mfma_f32_16x16x32_{f16,bf16}; 64 MFMAs per wave per K step
//   x tile          128 x 128 halves (32 KiB): 16-byte global loads (16 lanes = one 256-byte row) ->
//                   LDS rows of 256 B with the 16-byte chunk index XOR (row & 15), so the fragment
//                   read (16 lanes = 16 different rows, same chunk) is conflict-free
//   W tile          128 rows x 16 packed words: thread (o = tid % 16, w = tid / 16) owns rows 8o..8o+7
//                   of word w; the packed dequantise + two v_perm_b32 per register pair give, for each
//                   of its 8 columns, the 8-deep k fragment (16 B), stored at Bs[n][chunk o ^ (n & 15)]
//                   — the same swizzled [row][k] image as the x tile, so A and B fragments are read
//                   identically with ds_read_b128
//   pipeline        the next K step's global loads are issued into registers before the MFMAs of the
//                   current one and written to LDS after them (one LDS buffer, two barriers per step;
//                   two workgroups per CU overlap each other's staging)
//   tile order      1-D grid, XCD-aware: blocks that share an XCD walk the N tiles of one M block, so
//                   the 1 MiB x panel stays in that XCD's L2
//
// Numerics as everywhere: W rounded per element like awq_dequantize, fp32 accumulation, one rounding,
// bias added after it (second rounding).
#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kTlBM = 128, kTlBN = 128, kTlBK = 128;

typedef float tl_float2 __attribute__((ext_vector_type(2)));
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ void tl_dequant_pairs(uint32_t w, const ZeroF16& zf, uint32_t zw, const u32x4_t& s, uint32_t (&P)[4]) {
  if constexpr (DT == AWQ_DTYPE_F16) {
    half2_t d[4];
    unpack_sub_f16(w, zf, d);
    P[0] = as_u32(d[0] * as_h2(s.x));
    P[1] = as_u32(d[1] * as_h2(s.y));
    P[2] = as_u32(d[2] * as_h2(s.z));
    P[3] = as_u32(d[3] * as_h2(s.w));
  } else {
    const uint32_t sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      tl_float2 f;
      f.x = (float)(nibble_of_col(w, 2 * t) - nibble_of_col(zw, 2 * t)) * __builtin_bit_cast(float, sv[t] << 16);
      f.y = (float)(nibble_of_col(w, 2 * t + 1) - nibble_of_col(zw, 2 * t + 1)) * __builtin_bit_cast(float, sv[t] & 0xffff0000u);
      P[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, tl_bf16x2));
    }
  }
}

template <int DT>
__device__ __forceinline__ float4_t tl_mfma(const u32x4_t& a, const u32x4_t& b, const float4_t& c) {
  if constexpr (DT == AWQ_DTYPE_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// byte offset of 16-byte chunk `chunk` (0..15) of row `row` in a [rows][128 halves] swizzled image
__device__ __forceinline__ int tl_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

template <int DT>
__global__ __launch_bounds__(256, 2) void gemm_tiled_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                            const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                            const uint32_t* __restrict__ qz, const void* __restrict__ bias,
                                                            void* __restrict__ y, int M, int K, int C, int g, int nbx, int nby) {
  __shared__ __attribute__((aligned(16))) unsigned char As[kTlBM * 256];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[kTlBN * 256];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int q = lane >> 4, r = lane & 15;
  const int N = C * 8;

  // XCD-aware tile order (bijective for any grid size): blocks with equal blockIdx % 8 share an XCD
  const int nwg = nbx * nby;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int bm = (logical / nbx) * kTlBM;
  const int bn = (logical % nbx) * kTlBN;

  // staging registers for the next K step
  u32x4_t a_st[8];
  uint32_t w_st[8];
  uint32_t zw_st;
  u32x4_t sc_st;
  const int o = tid & 15, wcol = tid >> 4;                    // W unit: rows 8o..8o+7, packed word wcol of the tile
  const int word = (bn >> 3) + wcol < C ? (bn >> 3) + wcol : C - 1;   // clamped: columns >= N are never stored

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 4, chunk = c & 15;
      const int m = bm + row < M ? bm + row : M - 1;          // clamped: rows >= M are never stored
      a_st[i] = *(const u32x4_t*)(x + (size_t)m * ldx + kt + chunk * 8);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) w_st[j] = qw[(size_t)(kt + 8 * o + j) * C + word];
    const int grp = (kt + 8 * o) / g;
    zw_st = qz[(size_t)grp * C + word];
    sc_st = *(const u32x4_t*)(scales + (size_t)grp * N + (size_t)word * 8);
  };

  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      *(u32x4_t*)(As + tl_off(c >> 4, c & 15)) = a_st[i];
    }
    ZeroF16 zf;
    if constexpr (DT == AWQ_DTYPE_F16) zf = make_zero_f16(zw_st);
    uint32_t P[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) tl_dequant_pairs<DT>(w_st[j], zf, zw_st, sc_st, P[j]);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const u32x4_t flo = {pack_lo16(P[0][tt], P[1][tt]), pack_lo16(P[2][tt], P[3][tt]),
                           pack_lo16(P[4][tt], P[5][tt]), pack_lo16(P[6][tt], P[7][tt])};
      const u32x4_t fhi = {pack_hi16(P[0][tt], P[1][tt]), pack_hi16(P[2][tt], P[3][tt]),
                           pack_hi16(P[4][tt], P[5][tt]), pack_hi16(P[6][tt], P[7][tt])};
      const int n0 = wcol * 8 + 2 * tt;
      *(u32x4_t*)(Bs + tl_off(n0, o)) = flo;
      *(u32x4_t*)(Bs + tl_off(n0 + 1, o)) = fhi;
    }
  };

  float4_t acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (float4_t){0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile();
  __syncthreads();

  for (int kt = 0; kt < K; kt += kTlBK) {
    const bool more = kt + kTlBK < K;
    // unconditional (clamped) prefetch: a conditional load would make hipcc drain vmcnt at the merge
    load_tile(more ? kt + kTlBK : kt);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4_t af[4], bf[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[mi] = *(const u32x4_t*)(As + tl_off(wm * 64 + mi * 16 + r, kk * 4 + q));
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bf[ni] = *(const u32x4_t*)(Bs + tl_off(wn * 64 + ni * 16 + r, kk * 4 + q));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = tl_mfma<DT>(af[mi], bf[ni], acc[mi][ni]);
    }
    __syncthreads();
    if (more) store_tile();
    __syncthreads();
  }

  // epilogue: D[m = 4q + i][n = r] per 16 x 16 tile
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = bm + wm * 64 + mi * 16 + 4 * q + i;
      if (m < M) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = bn + wn * 64 + ni * 16 + r;
          if (n < N) store_output<DT>(y, (size_t)m * N + n, acc[mi][ni][i], bias, n);
        }
      }
    }
}

bool tiled_supported(const GemmArgs& a) {
  if (a.dtype != AWQ_DTYPE_F16 && a.dtype != AWQ_DTYPE_BF16) return false;
  if (a.M < 1) return false;
  if (a.K % kTlBK || a.g % 8 || a.N % 8 || a.ldx % 8) return false;
  if ((((uintptr_t)a.x) | ((uintptr_t)a.scales)) & 15) return false;
  const long long nwg = (long long)((a.N + kTlBN - 1) / kTlBN) * ((a.M + kTlBM - 1) / kTlBM);
  return nwg > 0 && nwg < (1ll << 30);
}

int launch_gemm_tiled(const GemmArgs& a) {
  if (!tiled_supported(a)) return AWQ_ERR_BAD_VARIANT;
  const int C = a.N / 8;
  const int nbx = (a.N + kTlBN - 1) / kTlBN, nby = (a.M + kTlBM - 1) / kTlBM;
  dim3 grid(nbx * nby), block(256);
  if (a.dtype == AWQ_DTYPE_F16)
    hipLaunchKernelGGL(gemm_tiled_kernel<AWQ_DTYPE_F16>, grid, block, 0, a.stream, (const uint16_t*)a.x, a.ldx, (const uint32_t*)a.qweight,
                       (const uint16_t*)a.scales, (const uint32_t*)a.qzeros, a.bias, a.y, a.M, a.K, C, a.g, nbx, nby);
  else
    hipLaunchKernelGGL(gemm_tiled_kernel<AWQ_DTYPE_BF16>, grid, block, 0, a.stream, (const uint16_t*)a.x, a.ldx, (const uint32_t*)a.qweight,
                       (const uint16_t*)a.scales, (const uint32_t*)a.qzeros, a.bias, a.y, a.M, a.K, C, a.g, nbx, nby);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
