// Generic fused AWQ GEMM (VALU, no MFMA): the catch-all behind awq_gemm for shapes and dtypes the
// MFMA kernels do not take (fp32 scales as in the reference's Triton GEMM test,
// test/srt/quant/test_awq_dequant.py:134-171; N % 32 != 0; K or g not multiples of 32).
// Same numerics as every other variant: W rounded to dtype per element, fp32 accumulation, one
// rounding of the sum, optional bias added with a second rounding.
//
// Workgroup = 64 packed columns (512 outputs) x kGenMT rows of x; its 4 waves take interleaved
// K rows and are summed in a fixed order through LDS (deterministic).
#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kGenMT = 4;
constexpr int kGenWaves = 4;

template <int DT>
__global__ __launch_bounds__(kGenWaves * 64) void gemm_generic_kernel(const void* __restrict__ x, int64_t ldx,
                                                                      const uint32_t* __restrict__ qw,
                                                                      const void* __restrict__ scales,
                                                                      const uint32_t* __restrict__ qz,
                                                                      const void* __restrict__ bias,
                                                                      void* __restrict__ y, int M, int K, int C, int g) {
  __shared__ float red[kGenWaves][kGenMT][64 * 8 + 8];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * kGenMT;
  const size_t N = (size_t)C * 8;
  constexpr int EB = ElemBytes<DT>::v;

  float acc[kGenMT][8];
#pragma unroll
  for (int m = 0; m < kGenMT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;

  if (c < C) {
    int cur = -1;
    uint32_t zw = 0;
    const char* sc = nullptr;
    for (int k = wave; k < K; k += kGenWaves) {
      const int grp = k / g;
      if (grp != cur) {
        cur = grp;
        zw = qz[(size_t)grp * C + c];
        sc = (const char*)scales + ((size_t)grp * N + (size_t)c * 8) * EB;
      }
      float wv[8];
      dequant_word<DT>(qw[(size_t)k * C + c], zw, sc, wv);
#pragma unroll
      for (int m = 0; m < kGenMT; ++m) {
        if (m0 + m < M) {
          const float xv = load_as_float<DT>(x, (size_t)(m0 + m) * ldx + k);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m][j] = __builtin_fmaf(xv, wv[j], acc[m][j]);
        }
      }
    }
  }
#pragma unroll
  for (int m = 0; m < kGenMT; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][m][lane * 8 + j] = acc[m][j];
  __syncthreads();
  for (int i = threadIdx.x; i < kGenMT * 512; i += kGenWaves * 64) {
    const int m = i / 512, nl = i % 512;
    const int n = blockIdx.x * 512 + nl;
    if (m0 + m < M && (size_t)n < N) {
      float v = red[0][m][nl];
#pragma unroll
      for (int w = 1; w < kGenWaves; ++w) v += red[w][m][nl];
      store_output<DT>(y, (size_t)(m0 + m) * N + n, v, bias, n);
    }
  }
}

int launch_gemm_generic(const GemmArgs& a) {
  const int C = a.N / 8;
  dim3 grid((C + 63) / 64, (a.M + kGenMT - 1) / kGenMT);
  dim3 block(kGenWaves * 64);
  if (a.M == 0) return AWQ_OK;
  const uint32_t* qw = (const uint32_t*)a.qweight;
  const uint32_t* qz = (const uint32_t*)a.qzeros;
  switch (a.dtype) {
    case AWQ_DTYPE_F16:
      hipLaunchKernelGGL(gemm_generic_kernel<AWQ_DTYPE_F16>, grid, block, 0, a.stream, a.x, a.ldx, qw, a.scales, qz, a.bias, a.y, a.M, a.K, C, a.g);
      break;
    case AWQ_DTYPE_BF16:
      hipLaunchKernelGGL(gemm_generic_kernel<AWQ_DTYPE_BF16>, grid, block, 0, a.stream, a.x, a.ldx, qw, a.scales, qz, a.bias, a.y, a.M, a.K, C, a.g);
      break;
    case AWQ_DTYPE_F32:
      hipLaunchKernelGGL(gemm_generic_kernel<AWQ_DTYPE_F32>, grid, block, 0, a.stream, a.x, a.ldx, qw, a.scales, qz, a.bias, a.y, a.M, a.K, C, a.g);
      break;
    default:
      return AWQ_ERR_BAD_DTYPE;
  }
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
