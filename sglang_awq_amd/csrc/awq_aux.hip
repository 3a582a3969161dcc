// Fused neighbours of the AWQ linears for the decode harness: the elementwise steps that
// sit either side of the four projections of a Llama layer (python/sglang/srt/models/llama.py:94-106,
// :188-199 — RMSNorm, rotary embedding + KV-cache write, SiluAndMul).  At batch 1 each of them is a handful
// of kilobytes, so as separate eager ops they are pure launch overhead; here each is one small launch.
// fp16 in / out, fp32 arithmetic.  These are harness helpers, not part of the drop-in op boundary.
#include "../../include/awq_aux.h"
#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

typedef _Float16 half8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nw; ++w) t += scratch[w];
  __syncthreads();
  return t;
}

// h[row] (+= delta[row], written back) ; out[row] = rmsnorm(h[row]) * w.   One workgroup per row, H % 8 == 0.
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(half_t* __restrict__ h, const half_t* __restrict__ delta,
                                                          const half_t* __restrict__ w, half_t* __restrict__ out, int H, float eps) {
  __shared__ float scratch[8];
  const size_t base = (size_t)blockIdx.x * H;
  float ss = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
    half8v v = *(const half8v*)(h + base + i);
    if (delta) {
      const half8v d = *(const half8v*)(delta + base + i);
      v = v + d;                                              // fp16 add, as the eager `h = h + o`
      *(half8v*)(h + base + i) = v;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
  }
  const float inv = __builtin_amdgcn_rsqf(block_sum(ss, scratch) / (float)H + eps);
  for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
    const half8v v = *(const half8v*)(h + base + i);
    const half8v ww = *(const half8v*)(w + i);
    half8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[e] * inv) * ww[e];   // round to fp16, then scale (eager order)
    *(half8v*)(out + base + i) = o;
  }
}

// In place: neox-style rotary embedding on the q and k heads of qkv[b], then k / v of this token into the
// caches [B, Hkv, S, D] at position pos[b].  One workgroup per (token, head); D / 2 threads.
__global__ void rope_kv_kernel(half_t* __restrict__ qkv, const int64_t* __restrict__ pos, const float* __restrict__ cos_t,
                               const float* __restrict__ sin_t, half_t* __restrict__ kc, half_t* __restrict__ vc, int Hq, int Hkv, int D,
                               int S) {
  const int b = blockIdx.y, head = blockIdx.x, t = threadIdx.x, half = D / 2;
  const int64_t p = pos[b];
  half_t* row = qkv + (size_t)b * (Hq + 2 * Hkv) * D + (size_t)head * D;
  if (head < Hq + Hkv) {
    const float c = cos_t[p * half + t], s = sin_t[p * half + t];
    const float x1 = (float)row[t], x2 = (float)row[t + half];
    const half_t o1 = (half_t)(x1 * c - x2 * s), o2 = (half_t)(x2 * c + x1 * s);
    row[t] = o1;
    row[t + half] = o2;
    if (head >= Hq) {
      half_t* dst = kc + (((size_t)b * Hkv + (head - Hq)) * S + p) * D;
      dst[t] = o1;
      dst[t + half] = o2;
    }
  } else {
    half_t* dst = vc + (((size_t)b * Hkv + (head - Hq - Hkv)) * S + p) * D;
    dst[t] = row[t];
    dst[t + half] = row[t + half];
  }
}

// act[b, i] = silu(gu[b, i]) * gu[b, I + i]   (SiluAndMul), I % 8 == 0
__global__ __launch_bounds__(256) void silu_mul_kernel(const half_t* __restrict__ gu, half_t* __restrict__ act, int I, size_t total8) {
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total8; v += (size_t)gridDim.x * blockDim.x) {
    const size_t b = (v * 8) / I, i = (v * 8) % I;
    const half8v g = *(const half8v*)(gu + b * 2 * I + i);
    const half8v u = *(const half8v*)(gu + b * 2 * I + I + i);
    half8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = (float)g[e];
      o[e] = (half_t)(x / (1.f + __expf(-x))) * u[e];
    }
    *(half8v*)(act + b * I + i) = o;
  }
}

}  // namespace awq

extern "C" {

int awq_aux_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t rows, int64_t H, float eps, void* stream) {
  if (!h || !w || !out) return AWQ_ERR_NULL_POINTER;
  if (rows <= 0 || H <= 0 || H % 8) return AWQ_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(awq::add_rmsnorm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (awq::half_t*)h,
                     (const awq::half_t*)delta, (const awq::half_t*)w, (awq::half_t*)out, (int)H, eps);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_aux_rope_kv(void* qkv, const int64_t* pos, const float* cos_t, const float* sin_t, void* k_cache, void* v_cache, int64_t B,
                    int64_t Hq, int64_t Hkv, int64_t D, int64_t S, void* stream) {
  if (!qkv || !pos || !cos_t || !sin_t || !k_cache || !v_cache) return AWQ_ERR_NULL_POINTER;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || D % 2 || D / 2 > 1024 || S <= 0) return AWQ_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(awq::rope_kv_kernel, dim3((unsigned)(Hq + 2 * Hkv), (unsigned)B), dim3((unsigned)(D / 2)), 0, (hipStream_t)stream,
                     (awq::half_t*)qkv, pos, cos_t, sin_t, (awq::half_t*)k_cache, (awq::half_t*)v_cache, (int)Hq, (int)Hkv, (int)D, (int)S);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_aux_silu_mul(const void* gate_up, void* act, int64_t rows, int64_t I, void* stream) {
  if (!gate_up || !act) return AWQ_ERR_NULL_POINTER;
  if (rows <= 0 || I <= 0 || I % 8) return AWQ_ERR_BAD_SHAPE;
  const size_t total8 = (size_t)rows * I / 8;
  const unsigned grid = (unsigned)((total8 + 255) / 256 < 2048 ? (total8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(awq::silu_mul_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const awq::half_t*)gate_up, (awq::half_t*)act,
                     (int)I, total8);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // extern "C"
