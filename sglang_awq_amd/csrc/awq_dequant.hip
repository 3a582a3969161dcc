// awq_dequantize for gfx950: int4 -> fp16 / bf16 / fp32, bit-exact with the reference op
// (sgl-kernel/csrc/gemm/awq_kernel.cu:126-221).
//
// HBM-bound: per packed word 4 B are read and 16 B (fp16/bf16) written.  One lane owns one packed
// column c and kRows consecutive rows, so a wave reads 256 contiguous bytes per row and writes
// 1 KiB contiguous per row (16 B per lane, fully coalesced); the group's zero word and scale
// vector are loaded once per lane and reused for its rows.  All kRows loads are issued before the
// first use so each wave keeps kRows * 256 B of reads in flight.
#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kDqRows = 8;     // rows per lane
constexpr int kDqWaves = 4;    // waves per workgroup, stacked along K

template <int DT>
__global__ __launch_bounds__(kDqWaves * 64) void dequant_kernel(const uint32_t* __restrict__ qw,
                                                                 const void* __restrict__ scales_v,
                                                                 const uint32_t* __restrict__ qz,
                                                                 void* __restrict__ out_v, int K, int C, int g) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.y * kDqWaves + wave) * kDqRows;
  if (c >= C || row0 >= K) return;
  const size_t N = (size_t)C * 8;

  uint32_t w[kDqRows];
#pragma unroll
  for (int j = 0; j < kDqRows; ++j) {
    const int row = row0 + j;
    w[j] = row < K ? qw[(size_t)row * C + c] : 0u;
  }

  int grp = row0 / g;
  int rem = row0 - grp * g;
  bool fresh = true;

  if constexpr (DT == AWQ_DTYPE_F16) {
    const half_t* scales = (const half_t*)scales_v;
    half_t* out = (half_t*)out_v;
    ZeroF16 z;
    u32x4_t s;
#pragma unroll
    for (int j = 0; j < kDqRows; ++j) {
      const int row = row0 + j;
      if (row >= K) break;
      if (fresh) {
        z = make_zero_f16(qz[(size_t)grp * C + c]);
        s = *(const u32x4_t*)(scales + (size_t)grp * N + (size_t)c * 8);
        fresh = false;
      }
      half2_t d[4];
      unpack_sub_f16(w[j], z, d);
      u32x4_t o;
      o.x = as_u32(d[0] * as_h2(s.x));
      o.y = as_u32(d[1] * as_h2(s.y));
      o.z = as_u32(d[2] * as_h2(s.z));
      o.w = as_u32(d[3] * as_h2(s.w));
      __builtin_nontemporal_store(o, (u32x4_t*)(out + (size_t)row * N + (size_t)c * 8));   // written once, 4 x the bytes read: keep it out of the caches (24.5 -> 19.7 us)
      if (++rem == g) { rem = 0; ++grp; fresh = true; }
    }
  } else if constexpr (DT == AWQ_DTYPE_BF16) {
    const uint16_t* scales = (const uint16_t*)scales_v;
    uint16_t* out = (uint16_t*)out_v;
    uint32_t zw = 0;
    u32x4_t s;
#pragma unroll
    for (int j = 0; j < kDqRows; ++j) {
      const int row = row0 + j;
      if (row >= K) break;
      if (fresh) {
        zw = qz[(size_t)grp * C + c];
        s = *(const u32x4_t*)(scales + (size_t)grp * N + (size_t)c * 8);
        fresh = false;
      }
      const uint32_t sv[4] = {s.x, s.y, s.z, s.w};
      uint32_t ov[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // (q - z) is an integer in [-15, 15]; times an 8-bit significand it is exact in fp32,
        // so one fp32 multiply + one RNE conversion equals the reference's bf16 hsub2 + hmul2.
        const float lo = (float)(nibble_of_col(w[j], 2 * t) - nibble_of_col(zw, 2 * t)) * bf16_bits_to_float((uint16_t)(sv[t] & 0xffffu));
        const float hi = (float)(nibble_of_col(w[j], 2 * t + 1) - nibble_of_col(zw, 2 * t + 1)) * bf16_bits_to_float((uint16_t)(sv[t] >> 16));
        ov[t] = (uint32_t)float_to_bf16_bits(lo) | ((uint32_t)float_to_bf16_bits(hi) << 16);
      }
      u32x4_t o = {ov[0], ov[1], ov[2], ov[3]};
      __builtin_nontemporal_store(o, (u32x4_t*)(out + (size_t)row * N + (size_t)c * 8));
      if (++rem == g) { rem = 0; ++grp; fresh = true; }
    }
  } else {
    const float* scales = (const float*)scales_v;
    float* out = (float*)out_v;
    uint32_t zw = 0;
    float4_t s0, s1;
#pragma unroll
    for (int j = 0; j < kDqRows; ++j) {
      const int row = row0 + j;
      if (row >= K) break;
      if (fresh) {
        zw = qz[(size_t)grp * C + c];
        s0 = *(const float4_t*)(scales + (size_t)grp * N + (size_t)c * 8);
        s1 = *(const float4_t*)(scales + (size_t)grp * N + (size_t)c * 8 + 4);
        fresh = false;
      }
      float4_t o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o0[e] = (float)(nibble_of_col(w[j], e) - nibble_of_col(zw, e)) * s0[e];
        o1[e] = (float)(nibble_of_col(w[j], e + 4) - nibble_of_col(zw, e + 4)) * s1[e];
      }
      __builtin_nontemporal_store(o0, (float4_t*)(out + (size_t)row * N + (size_t)c * 8));
      __builtin_nontemporal_store(o1, (float4_t*)(out + (size_t)row * N + (size_t)c * 8 + 4));
      if (++rem == g) { rem = 0; ++grp; fresh = true; }
    }
  }
}

int launch_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out, int64_t K,
                      int64_t N, int64_t g, int dtype, hipStream_t stream) {
  const int C = (int)(N / 8);
  dim3 grid((C + 63) / 64, (unsigned)((K + kDqWaves * kDqRows - 1) / (kDqWaves * kDqRows)));
  dim3 block(kDqWaves * 64);
  const uint32_t* qw = (const uint32_t*)qweight;
  const uint32_t* qz = (const uint32_t*)qzeros;
  switch (dtype) {
    case AWQ_DTYPE_F16:
      hipLaunchKernelGGL(dequant_kernel<AWQ_DTYPE_F16>, grid, block, 0, stream, qw, scales, qz, out, (int)K, C, (int)g);
      break;
    case AWQ_DTYPE_BF16:
      hipLaunchKernelGGL(dequant_kernel<AWQ_DTYPE_BF16>, grid, block, 0, stream, qw, scales, qz, out, (int)K, C, (int)g);
      break;
    case AWQ_DTYPE_F32:
      hipLaunchKernelGGL(dequant_kernel<AWQ_DTYPE_F32>, grid, block, 0, stream, qw, scales, qz, out, (int)K, C, (int)g);
      break;
    default:
      return AWQ_ERR_BAD_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
