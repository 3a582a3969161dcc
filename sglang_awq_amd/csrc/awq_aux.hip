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
// One memory round trip: every thread requests its chunks of h, delta AND w up front and keeps h + delta in registers
// across the reduction (rows up to 256 x 8 x kNormChunks halves; longer rows take the strided two-pass loop).
constexpr int kNormChunks = 4;
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(half_t* __restrict__ h, const half_t* __restrict__ delta,
                                                          const half_t* __restrict__ w, half_t* __restrict__ out, int H, float eps) {
  __shared__ float scratch[8];
  const size_t base = (size_t)blockIdx.x * H;
  if (H <= 256 * 8 * kNormChunks) {
    half8v v[kNormChunks], ww[kNormChunks];
#pragma unroll
    for (int c = 0; c < kNormChunks; ++c) {
      const int i = (threadIdx.x + 256 * c) * 8;
      const int ic = i < H ? i : 0;                       // clamped: loaded, never used
      v[c] = *(const half8v*)(h + base + ic);
      if (delta) v[c] = v[c] + *(const half8v*)(delta + base + ic);     // fp16 add, as the eager `h = h + o`
      ww[c] = *(const half8v*)(w + ic);
    }
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < kNormChunks; ++c) {
      const int i = (threadIdx.x + 256 * c) * 8;
      if (i < H) {
        if (delta) *(half8v*)(h + base + i) = v[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += (float)v[c][e] * (float)v[c][e];
      }
    }
    const float inv = __builtin_amdgcn_rsqf(block_sum(ss, scratch) / (float)H + eps);
#pragma unroll
    for (int c = 0; c < kNormChunks; ++c) {
      const int i = (threadIdx.x + 256 * c) * 8;
      if (i < H) {
        half8v o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[c][e] * inv) * ww[c][e];   // round to fp16, then scale (eager order)
        *(half8v*)(out + base + i) = o;
      }
    }
    return;
  }
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

// Single-token attention for one (sequence, query head) per workgroup, with the rotary embedding and the
// KV-cache write of the new token folded in (what a rotary-embedding / cache-write launch + an SDPA launch would do in two).  The new
// token's k / v come straight from qkv (every workgroup of a GQA group rotates its own copy; the group's first
// head stores them), so no workgroup reads what another one writes in the same launch.
// Latency-bound at decode sizes (a few hundred KB per workgroup), so the structure minimises dependent round
// trips: the first block of K AND V rows is requested at kernel entry, before pos[b] is known (the cache is
// allocated to S rows, rows past pos are masked later); K and V of a position travel together and feed an
// online softmax kept per 16-lane position group (LP = D / 8 lanes x 16 B per cached row, 256 / LP groups,
// 4 positions per group in flight, the next block prefetched while the current one is reduced); the groups are
// merged once through LDS.  No per-position storage: any context length.
// DBG (laboratory build only, tools/time_attn.py with AWQ_ATTN_DBG): 1 = no merge across splits (wrong results), 2 = pos not loaded,
// 4 = every K / V row read is one of the first rows (cache-resident) — what each dependent piece of the launch costs.
template <int D, int DBG = 0>
__global__ __launch_bounds__(256) void decode_attention_kernel(const half_t* __restrict__ qkv, const int64_t* __restrict__ pos,
                                                               const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                               half_t* __restrict__ kc, half_t* __restrict__ vc,
                                                               half_t* __restrict__ out, int Hq, int Hkv, int S, float scale,
                                                               float* __restrict__ ws_part, unsigned* __restrict__ ws_cnt) {
  constexpr int LP = D / 8, NPG = 256 / LP, HALF = D / 2, U = 4;
  __shared__ float q_s[D];
  __shared__ __attribute__((aligned(16))) half_t knew_s[D];
  __shared__ __attribute__((aligned(16))) half_t vnew_s[D];
  __shared__ float m_s[NPG], l_s[NPG];
  __shared__ float part[NPG][D];

  const int b = blockIdx.y, h = blockIdx.x, t = threadIdx.x;
  const int rep = Hq / Hkv, kvh = h / rep;
  const bool owner = (h % rep) == 0;
  const int lp = t % LP, pg = t / LP;
  half_t* kbase = kc + ((size_t)b * Hkv + kvh) * (size_t)S * D;
  half_t* vbase = vc + ((size_t)b * Hkv + kvh) * (size_t)S * D;

  // Split-S ("flash-decoding"): gridDim.z workgroups share one (sequence, head); the context is cut into blocks of
  // BLK = 4 * NPG positions dealt round-robin (split sp takes blocks sp, sp + splits, ...), so the first block of every
  // split has an address that does not depend on pos.  Each split leaves an unnormalised (max, sum, weighted V) partial
  // in the workspace with write-through stores and the last one to arrive merges them in split order (the ticket
  // hand-off of awq_gemm_skinny.hip: no fences, deterministic).  One workgroup per head leaves 7/8 of the chip idle at
  // batch 1: 16 us per layer at context 1024, 44 us at 4096.
  constexpr int BLK = U * NPG;
  const int ns = (int)gridDim.z, sp = (int)blockIdx.z;

  // first block of K / V, requested before pos is known (rows < S exist; what lies past pos is masked below)
  half8v kv[U], vv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = sp * BLK + pg + u * NPG;
    const int jc = (DBG & 4) ? pg : (j < S ? j : S - 1);
    kv[u] = *(const half8v*)(kbase + (size_t)jc * D + lp * 8);
    vv[u] = *(const half8v*)(vbase + (size_t)jc * D + lp * 8);
  }

  const int64_t p_raw = (DBG & 2) ? (int64_t)(S - 64) : pos[b];                                     // cached positions 0 .. p-1, the new token is p
  // the host (GraphedDecoder) refuses to step past the cache; should a caller get here with p >= S anyway, clamp: the last
  // slot is overwritten and attended to, nothing outside the cache or the rotary tables is touched
  const int p = (int)(p_raw < 0 ? 0 : (p_raw < S ? p_raw : S - 1));
  const bool has_new = (p / BLK) % ns == sp;                         // the split whose block holds the new token stores it
  const half_t* row = qkv + (size_t)b * (Hq + 2 * Hkv) * D;
  if (t < HALF) {                                                    // q: rotate, round to fp16 as the unfused path does, pre-scale
    const float c = cos_t[(size_t)p * HALF + t], sn = sin_t[(size_t)p * HALF + t];
    const float x1 = (float)row[(size_t)h * D + t], x2 = (float)row[(size_t)h * D + t + HALF];
    q_s[t] = (float)(half_t)(x1 * c - x2 * sn) * scale;
    q_s[t + HALF] = (float)(half_t)(x2 * c + x1 * sn) * scale;
  } else if (t < D) {                                                // k of the new token
    const int i = t - HALF;
    const float c = cos_t[(size_t)p * HALF + i], sn = sin_t[(size_t)p * HALF + i];
    const half_t* kr = row + (size_t)(Hq + kvh) * D;
    const float x1 = (float)kr[i], x2 = (float)kr[i + HALF];
    const half_t o1 = (half_t)(x1 * c - x2 * sn), o2 = (half_t)(x2 * c + x1 * sn);
    knew_s[i] = o1;
    knew_s[i + HALF] = o2;
    if (owner && has_new) {
      kbase[(size_t)p * D + i] = o1;
      kbase[(size_t)p * D + i + HALF] = o2;
    }
  } else if (t < D + HALF) {                                         // v of the new token
    const int i = t - D;
    const half_t* vr = row + (size_t)(Hq + Hkv + kvh) * D;
    const half_t v1 = vr[i], v2 = vr[i + HALF];
    vnew_s[i] = v1;
    vnew_s[i + HALF] = v2;
    if (owner && has_new) {
      vbase[(size_t)p * D + i] = v1;
      vbase[(size_t)p * D + i + HALF] = v2;
    }
  }
  __syncthreads();

  float qf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = q_s[lp * 8 + e];
  const half8v knew = *(const half8v*)(knew_s + lp * 8), vnew = *(const half8v*)(vnew_s + lp * 8);

  float m = -INFINITY, l = 0.f;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int jb = sp * BLK; jb <= p; jb += ns * BLK) {
    const int j0 = jb + pg;
    half8v kn[U], vn[U];                                             // this split's next block, requested before this one is reduced
    const bool more = jb + ns * BLK <= p;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = j0 + ns * BLK + u * NPG;
        const int jc = (DBG & 4) ? pg : (j < p ? j : (p > 0 ? p - 1 : 0));
        kn[u] = *(const half8v*)(kbase + (size_t)jc * D + lp * 8);
        vn[u] = *(const half8v*)(vbase + (size_t)jc * D + lp * 8);
      }
    }
    float sc[U];
    float bm = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NPG;
      const half8v k8 = j == p ? knew : kv[u];
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d += qf[e] * (float)k8[e];
#pragma unroll
      for (int o = LP / 2; o > 0; o >>= 1) d += __shfl_xor(d, o);
      sc[u] = j <= p ? d : -INFINITY;
      bm = fmaxf(bm, sc[u]);
    }
    // (a group whose four positions all lie past pos keeps bm = m; with m = -inf that is exp(-inf - -inf): guard it)
    const float corr = bm == -INFINITY ? 0.f : __expf(m - bm);
    l *= corr;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= corr;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u * NPG;
      if (j <= p) {                                                  // rows past pos may hold anything (even NaN): never touch them
        const half8v v8 = j == p ? vnew : vv[u];
        const float w = __expf(sc[u] - bm);
        l += w;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * (float)v8[e];
      }
    }
    m = bm;
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) { kv[u] = kn[u]; vv[u] = vn[u]; }
    }
  }

  // merge the position groups (a group that saw no position has m = -inf, l = 0)
  if (lp == 0) { m_s[pg] = m; l_s[pg] = l; }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[pg][lp * 8 + e] = acc[e];
  __syncthreads();
  float mx = m_s[0];
#pragma unroll
  for (int g = 1; g < NPG; ++g) mx = fmaxf(mx, m_s[g]);
  float o = 0.f, lt = 0.f;
  if (t < D) {
#pragma unroll
    for (int g = 0; g < NPG; ++g) {
      const float f = m_s[g] == -INFINITY ? 0.f : __expf(m_s[g] - mx);   // (an empty split has mx = -inf too)
      o += part[g][t] * f;
      lt += l_s[g] * f;
    }
  }
  if (ns == 1 || (DBG & 1)) {
    if (t < D) out[((size_t)b * Hq + h) * D + t] = (half_t)(o / lt);
    return;
  }

  // partial of this split: [D] weighted V, then max, then sum — write-through, then one ticket per workgroup
  const size_t slot = ((size_t)b * Hq + h) * ns;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(ws_part, 0, (int)((size_t)gridDim.y * Hq * ns * (D + 2) * sizeof(float)), 0x00020000);
  const unsigned pbase = (unsigned)((slot + sp) * (D + 2) * sizeof(float));
  if (t < D) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), rsrc, pbase + (unsigned)t * 4u, 0, 16);
  if (t == 0) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mx), rsrc, pbase + (unsigned)D * 4u, 0, 16);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, lt), rsrc, pbase + (unsigned)D * 4u + 4u, 0, 16);
  }
  __shared__ unsigned ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) ticket = __hip_atomic_fetch_add(&ws_cnt[(size_t)b * Hq + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != (unsigned)(ns - 1)) return;
  if (t == 0) __hip_atomic_store(&ws_cnt[(size_t)b * Hq + h], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  if (t < D) {
    float pm[16], pl[16], po[16];                                     // ns <= 16: every load issued before the first use
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      const unsigned sb = (unsigned)((slot + (s2 < ns ? s2 : 0)) * (D + 2) * sizeof(float));
      po[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, sb + (unsigned)t * 4u, 0, 16));
      pm[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, sb + (unsigned)D * 4u, 0, 16));
      pl[s2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, sb + (unsigned)D * 4u + 4u, 0, 16));
    }
    float M2 = pm[0];
#pragma unroll
    for (int s2 = 1; s2 < 16; ++s2) if (s2 < ns) M2 = fmaxf(M2, pm[s2]);
    float oo = 0.f, ll = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      if (s2 < ns) {
        const float f = pm[s2] == -INFINITY ? 0.f : __expf(pm[s2] - M2);
        oo += po[s2] * f;
        ll += pl[s2] * f;
      }
    }
    out[((size_t)b * Hq + h) * D + t] = (half_t)(oo / ll);
  }
}

// out[t, :] = fp16(sum over k of fp32(y[t * top_k + k, :]))   (the combine step of an MoE layer: the pairs' fp16 outputs added in fp32, one
// rounding — what `y.view(T, top_k, K).sum(1, dtype=float32).half()` computes in two launches), K % 8 == 0
// ids != nullptr: pairs whose expert id lies outside [0, num_experts) (padded tokens) are skipped — their rows of y need not be initialised.
__global__ __launch_bounds__(256) void moe_sum_kernel(const half_t* __restrict__ y, half_t* __restrict__ out, int top_k, int K, size_t total8,
                                                      const int* __restrict__ ids, int num_experts) {
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total8; v += (size_t)gridDim.x * blockDim.x) {
    const size_t t = (v * 8) / K, c = (v * 8) % K;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < top_k; ++k) {
      if (ids != nullptr) {
        const int e = ids[t * top_k + k];
        if (e < 0 || e >= num_experts) continue;
      }
      const half8v h = *(const half8v*)(y + (t * top_k + k) * K + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)h[e];
    }
    half8v o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
    *(half8v*)(out + t * K + c) = o;
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

// Greedy sampling tail of a decode step in one launch: tokens[b] = argmax(logits[b, :]) (first maximum, as torch.argmax),
// pos[b] += 1.  One workgroup per sequence; replaces a reduce kernel + a copy + an add (three dependent tiny launches).
__global__ __launch_bounds__(1024) void argmax_advance_kernel(const half_t* __restrict__ logits, int64_t* __restrict__ tokens,
                                                              int64_t* __restrict__ pos, int V) {
  __shared__ float bv[16];
  __shared__ int bi[16];
  const int b = blockIdx.x, t = threadIdx.x;
  const half_t* row = logits + (size_t)b * V;
  float best = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = t * 8; i < V; i += 1024 * 8) {
    if (i + 8 <= V) {
      const half8v v = *(const half8v*)(row + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)v[e];
        if (f > best || (f == best && i + e < idx)) { best = f; idx = i + e; }
      }
    } else {
      for (int e = 0; i + e < V; ++e) {
        const float f = (float)row[i + e];
        if (f > best || (f == best && i + e < idx)) { best = f; idx = i + e; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o);
    const int oi = __shfl_xor(idx, o);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    tokens[b] = idx;
    pos[b] += 1;
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

constexpr int64_t kAttnMaxTickets = 16384;                          // (sequence, head) pairs a split-S call may have
constexpr size_t kAttnTicketBytes = (size_t)kAttnMaxTickets * sizeof(unsigned);

int awq_aux_decode_attention(const void* qkv, const int64_t* pos, const float* cos_t, const float* sin_t, void* k_cache, void* v_cache,
                             void* out, int64_t B, int64_t Hq, int64_t Hkv, int64_t D, int64_t S, float scale, int num_splits,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!qkv || !pos || !cos_t || !sin_t || !k_cache || !v_cache || !out) return AWQ_ERR_NULL_POINTER;
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || (D != 64 && D != 128) || S <= 0 || num_splits < 1 || num_splits > 16) return AWQ_ERR_BAD_SHAPE;
  float* part = nullptr;
  unsigned* cnt = nullptr;
  if (num_splits > 1) {
    if (!workspace || (((uintptr_t)workspace) & 15)) return AWQ_ERR_WORKSPACE;
    if (B * Hq > kAttnMaxTickets) return AWQ_ERR_BAD_VARIANT;       // (callers run one workgroup per head there: the grid already fills the chip)
    if (workspace_bytes < awq_aux_decode_attention_workspace_bytes(B, Hq, D, num_splits)) return AWQ_ERR_WORKSPACE;
    // [kAttnMaxTickets] tickets in a header of FIXED size, zero once at allocation and left zero by every call, then the partials.
    // (The header used to be B * Hq words: a call with a larger batch than an earlier one on the same workspace then found the
    // earlier call's partials where its tickets should have been zero — wrong results from batch 4 at 2..8 splits; round 3.)
    cnt = (unsigned*)workspace;
    part = (float*)((char*)workspace + kAttnTicketBytes);
  }
  const dim3 grid((unsigned)Hq, (unsigned)B, (unsigned)num_splits), block(256);
#define AWQ_ATTN_GO(DD)                                                                                                          \
  hipLaunchKernelGGL(awq::decode_attention_kernel<DD>, grid, block, 0, (hipStream_t)stream, (const awq::half_t*)qkv, pos, cos_t,  \
                     sin_t, (awq::half_t*)k_cache, (awq::half_t*)v_cache, (awq::half_t*)out, (int)Hq, (int)Hkv, (int)S, scale, part, cnt)
#ifdef AWQ_LAB
  const int dbg = awq::lab_env("AWQ_ATTN_DBG", 0);
#define AWQ_ATTN_DBG_GO(V)                                                                                                        \
  if (D == 128 && dbg == V) {                                                                                                     \
    hipLaunchKernelGGL((awq::decode_attention_kernel<128, V>), grid, block, 0, (hipStream_t)stream, (const awq::half_t*)qkv, pos, \
                       cos_t, sin_t, (awq::half_t*)k_cache, (awq::half_t*)v_cache, (awq::half_t*)out, (int)Hq, (int)Hkv, (int)S,  \
                       scale, part, cnt);                                                                                         \
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;                                                             \
  }
  AWQ_ATTN_DBG_GO(1) AWQ_ATTN_DBG_GO(2) AWQ_ATTN_DBG_GO(3) AWQ_ATTN_DBG_GO(4) AWQ_ATTN_DBG_GO(6) AWQ_ATTN_DBG_GO(7)
#undef AWQ_ATTN_DBG_GO
#endif
  if (D == 128) AWQ_ATTN_GO(128); else AWQ_ATTN_GO(64);
#undef AWQ_ATTN_GO
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

size_t awq_aux_decode_attention_workspace_bytes(int64_t B, int64_t Hq, int64_t D, int num_splits) {
  if (num_splits <= 1 || B <= 0 || Hq <= 0 || D <= 0) return 0;
  return kAttnTicketBytes + (size_t)B * Hq * num_splits * (D + 2) * sizeof(float);
}

int awq_aux_argmax_advance(const void* logits, int64_t* tokens, int64_t* pos, int64_t B, int64_t V, void* stream) {
  if (!logits || !tokens || !pos) return AWQ_ERR_NULL_POINTER;
  if (B <= 0 || V <= 0 || V >= (1ll << 31) || V % 8) return AWQ_ERR_BAD_SHAPE;      // rows of V halves must stay 16-byte aligned
  if (((uintptr_t)logits) & 15) return AWQ_ERR_MISALIGNED;
  hipLaunchKernelGGL(awq::argmax_advance_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, (const awq::half_t*)logits, tokens, pos,
                     (int)V);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_aux_moe_sum(const void* y, void* out, int64_t tokens, int64_t top_k, int64_t K, const int32_t* expert_ids, int64_t num_experts,
                    void* stream) {
  if (!y || !out) return AWQ_ERR_NULL_POINTER;
  if (tokens <= 0 || top_k <= 0 || top_k > 1024 || K <= 0 || K % 8 || (expert_ids && (num_experts < 1 || num_experts > INT32_MAX))) return AWQ_ERR_BAD_SHAPE;
  if ((((uintptr_t)y) | ((uintptr_t)out)) & 15) return AWQ_ERR_MISALIGNED;
  const size_t total8 = (size_t)tokens * K / 8;
  const unsigned grid = (unsigned)((total8 + 255) / 256 < 2048 ? (total8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(awq::moe_sum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const awq::half_t*)y, (awq::half_t*)out, (int)top_k, (int)K,
                     total8, expert_ids, (int)num_experts);
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
