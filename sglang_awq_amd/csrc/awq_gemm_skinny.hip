// Skinny fused AWQ GEMM for decode shapes (M <= 16) on gfx950 — HBM-bound, so the design goal is
// to keep every CU's share of qweight streaming in wide loads while the previous rows are being
// dequantised, and to keep the dequantised weights out of memory entirely.
//
// Decomposition
//   column tile   512 output columns = 16 lanes x one 16-byte load (4 packed words = 32 columns)
//   k-step        32 rows = 4 lane groups (q) x 8 rows (j); lane (q, r) loads rows k0+8q+j of
//                 chunk r: 8 global_load_dwordx4 per k-step, 256 contiguous bytes per row
//   wave          a contiguous run of k-steps, software-pipelined two deep: the loads of k-steps
//                 i+1 and i+2 are in flight while k-step i is dequantised and fed to MFMA (the
//                 compiler's counted vmcnt keeps the younger buffer outstanding); the 32 column
//                 accumulators (one v_mfma_f32_16x16x32 chain per logical column) live in AGPRs
//   workgroup     4 waves on the same column tile and adjacent K ranges, summed in a fixed order
//                 through LDS; the grid is sized to one workgroup per CU
//   grid          n_ct column tiles x S K-slices; S > 1 partials go to fp32 slabs in the
//                 workspace with write-through (sc1) stores, and the workgroup that draws the last
//                 ticket for a column tile adds them in slice order (sc1 loads) and writes y —
//                 deterministic, no float atomics, no fences, no second launch.
//
// MFMA mapping (v_mfma_f32_16x16x32_{f16,bf16}): A = x (row m = lane & 15, k = 8 (lane >> 4) + j),
// B = dequantised W (k = 8 q + j, column = lane & 15).  A lane's 8 rows x one logical column are
// exactly one B fragment, so after the packed dequantise (column pairs, see awq_device.h) two
// v_perm_b32 per register pair transpose them into k-pairs; nothing goes through LDS.  D[m][n]
// comes back with m = 4 q + i, n-lane = r.
#include <cstdlib>

#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kSkWaves = 4;
constexpr int kSkRowStride = 16 * 36;           // floats per (wave, m) row in LDS: 16 lanes x (32 + 4 pad)
constexpr size_t kSkCounterBytes = 4096;        // 1024 column-tile counters at the head of the workspace
constexpr size_t kSkSlabBudget = 32u << 20;     // default cap on fp32 partial slabs
constexpr int kNumCUs = 256;                    // MI355X

template <int DT> struct ZeroC;
template <> struct ZeroC<AWQ_DTYPE_F16> { ZeroF16 z; };
template <> struct ZeroC<AWQ_DTYPE_BF16> { uint32_t zw; };

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ ZeroC<DT> make_zero(uint32_t zw) {
  ZeroC<DT> r;
  if constexpr (DT == AWQ_DTYPE_F16) r.z = make_zero_f16(zw);
  else r.zw = zw;
  return r;
}

// P[t] = packed (W[2t], W[2t+1]) of one word, rounded to the storage dtype exactly as awq_dequantize.
template <int DT>
__device__ __forceinline__ void dequant_pairs(uint32_t w, const ZeroC<DT>& z, const u32x4_t& s, uint32_t (&P)[4]) {
  if constexpr (DT == AWQ_DTYPE_F16) {
    half2_t d[4];
    unpack_sub_f16(w, z.z, d);
    P[0] = as_u32(d[0] * as_h2(s.x));
    P[1] = as_u32(d[1] * as_h2(s.y));
    P[2] = as_u32(d[2] * as_h2(s.z));
    P[3] = as_u32(d[3] * as_h2(s.w));
  } else {
    const uint32_t sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float2_t f;
      f.x = (float)(nibble_of_col(w, 2 * t) - nibble_of_col(z.zw, 2 * t)) * __builtin_bit_cast(float, sv[t] << 16);
      f.y = (float)(nibble_of_col(w, 2 * t + 1) - nibble_of_col(z.zw, 2 * t + 1)) * __builtin_bit_cast(float, sv[t] & 0xffff0000u);
      P[t] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
    }
  }
}

template <int DT>
__device__ __forceinline__ float4_t mfma16(const u32x4_t& a, const u32x4_t& b, const float4_t& c) {
  if constexpr (DT == AWQ_DTYPE_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Everything one k-step (32 rows x 512 columns) needs, in registers.
struct KStep {
  u32x4_t R[8];     // 8 rows x 4 packed words of this lane's chunk
  u32x4_t SC[4];    // the group's 32 scales of this lane's columns
  u32x4_t ZW;       // the group's 4 zero words
  u32x4_t XA;       // x[m = r][k0 + 8q .. +8]
};

struct SkinnyCtx {
  const uint16_t* x; int64_t ldx;
  const uint32_t* qw; const uint16_t* scales; const uint32_t* qz;
  int C, N, g, chunk4, q, r;   // chunk4 / r already clamped for loading
};

// Every load is unconditional: lanes of a ragged last column tile are clamped onto the last valid
// chunk and MFMA rows m >= M onto row M - 1; what they compute is never stored.  (A `cond ? load : 0`
// form makes hipcc merge the two values with copies that wait vmcnt(0) inside the load block and
// de-pipelines the loop.)
template <int ABL = 0>
__device__ __forceinline__ void load_kstep(KStep& b, const SkinnyCtx& c, int kstep) {
  const int k0 = kstep * 32 + 8 * c.q;
#pragma unroll
  for (int j = 0; j < 8; ++j) b.R[j] = *(const u32x4_t*)(c.qw + (size_t)(k0 + j) * c.C + c.chunk4);
  if constexpr (ABL == 4 || ABL == 6) {   // timing-only: no x / zeros / scales loads
    b.XA = b.R[0]; b.ZW = b.R[1];
#pragma unroll
    for (int w = 0; w < 4; ++w) b.SC[w] = b.R[2 + w];
    return;
  }
  b.XA = *(const u32x4_t*)(c.x + (size_t)c.r * c.ldx + k0);
  const int grp = (kstep * 32) / c.g;
  b.ZW = *(const u32x4_t*)(c.qz + (size_t)grp * c.C + c.chunk4);
#pragma unroll
  for (int w = 0; w < 4; ++w) b.SC[w] = *(const u32x4_t*)(c.scales + (size_t)grp * c.N + (size_t)(c.chunk4 + w) * 8);
}

template <int DT, int ABL = 0>
__device__ __forceinline__ void compute_kstep(const KStep& b, float4_t (&acc)[4][8]) {
  if constexpr (ABL == 5 || ABL == 6) {   // timing-only: consume the registers, no dequantise / MFMA
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c][j][0] += __builtin_bit_cast(float, b.R[j][c] ^ b.XA[c] ^ b.SC[c][j & 3] ^ b.ZW[c]);
    return;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const ZeroC<DT> zc = make_zero<DT>(b.ZW[c]);
    uint32_t P[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) dequant_pairs<DT>(b.R[j][c], zc, b.SC[c], P[j]);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      const u32x4_t flo = {pack_lo16(P[0][tt], P[1][tt]), pack_lo16(P[2][tt], P[3][tt]),
                           pack_lo16(P[4][tt], P[5][tt]), pack_lo16(P[6][tt], P[7][tt])};
      const u32x4_t fhi = {pack_hi16(P[0][tt], P[1][tt]), pack_hi16(P[2][tt], P[3][tt]),
                           pack_hi16(P[4][tt], P[5][tt]), pack_hi16(P[6][tt], P[7][tt])};
      acc[c][2 * tt] = mfma16<DT>(b.XA, flo, acc[c][2 * tt]);
      acc[c][2 * tt + 1] = mfma16<DT>(b.XA, fhi, acc[c][2 * tt + 1]);
    }
  }
}

// diagnostic stamps (ABL == 3 only): global 100 MHz clock per workgroup and phase, written to a
// debug area no other code reads (never present in the shipped instantiations)
#define AWQ_STAMP(slot)                                                                         \
  do {                                                                                          \
    if (ABL >= 3 && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)

// T > 0: exactly T k-steps per wave, fully unrolled, every load issued before the first use (straight-line
// code is what lets hipcc emit exact counted vmcnt waits; across branches it falls back to near-full drains).
// T == 0: any number of k-steps, 2-deep pipelined loop.
template <int DT, int T, int ABL = 0>   // ABL: timing-only builds (1 = no split-K tail, >= 3 stamped); 0 ships
__global__ __launch_bounds__(kSkWaves * 64, 1) void gemm_skinny_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qz, const void* __restrict__ bias, void* __restrict__ y, float* __restrict__ slabs,
    unsigned* __restrict__ counters, int M, int K, int C, int g, int n_ct, int S, int steps_per_wave) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [kSkWaves][M][kSkRowStride] (+ 1 ticket word)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ct = blockIdx.x % n_ct;
  const int ks = blockIdx.x / n_ct;
  const int N = C * 8;
  SkinnyCtx cx;
  cx.x = x; cx.ldx = ldx; cx.qw = qw; cx.scales = scales; cx.qz = qz;
  const int q = lane >> 4, r = lane & 15;
  cx.C = C; cx.N = N; cx.g = g; cx.q = q;
  cx.r = r < M ? r : M - 1;                                  // x row this lane feeds to the MFMA
  const int chunk4 = (ct * 16 + r) * 4;                      // first packed word of this lane's 16-byte chunk
  cx.chunk4 = chunk4 < C ? chunk4 : C - 4;
  float* my_red = red + (size_t)wave * M * kSkRowStride;
  unsigned long long* dbg = (unsigned long long*)((char*)counters + (48u << 20));
  AWQ_STAMP(0);
  unsigned long long clk0 = 0;
  if (ABL >= 3) clk0 = __builtin_amdgcn_s_memtime();

  // this wave's k-steps: [k_begin, k_end) of K / 32
  const int k_begin = (ks * kSkWaves + wave) * steps_per_wave;
  int k_end = k_begin + steps_per_wave;
  if (k_end > K / 32) k_end = K / 32;

  float4_t acc[4][8];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[c][e] = (float4_t){0.f, 0.f, 0.f, 0.f};

  if constexpr (T > 0) {
    const int last = K / 32 - 1;
    KStep buf[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ks_t = k_begin + t;
      load_kstep<ABL>(buf[t], cx, ks_t < last ? ks_t : last);     // clamped: a k-step past the end is re-read, not skipped
    }
    // keep every load above the first dequantise: without this fence the machine scheduler sinks the
    // later k-steps' loads next to their uses to save registers, which serialises the stream
    __builtin_amdgcn_sched_barrier(0);
    AWQ_STAMP(1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (k_begin + t > last) buf[t].XA = (u32x4_t){0u, 0u, 0u, 0u};   // ... and weighted by x = 0 (a select on the value, not a conditional load)
      compute_kstep<DT, ABL>(buf[t], acc);
      __builtin_amdgcn_sched_barrier(0);   // k-steps in arrival order: do not start on a younger buffer early
    }
  } else {
    KStep A, B;
    if (k_begin < k_end) load_kstep<ABL>(A, cx, k_begin);
    if (k_begin + 1 < k_end) load_kstep<ABL>(B, cx, k_begin + 1);
    AWQ_STAMP(1);
    for (int i = k_begin; i < k_end; i += 2) {
      compute_kstep<DT, ABL>(A, acc);
      if (i + 2 < k_end) load_kstep<ABL>(A, cx, i + 2);
      if (i + 1 < k_end) {
        compute_kstep<DT, ABL>(B, acc);
        if (i + 3 < k_end) load_kstep<ABL>(B, cx, i + 3);
      }
    }
  }
  AWQ_STAMP(3);
  if (ABL >= 3 && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + 2] = __builtin_amdgcn_s_memtime() - clk0;   // shader-clock ticks, start -> loop end

  // D[m = 4q + i][n-lane r]: park this wave's 32 columns x M rows in its private LDS rows
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * q + i;
      if (m < M) {
        float4_t* dst = (float4_t*)(my_red + (size_t)m * kSkRowStride + r * 36 + 8 * c);
        dst[0] = (float4_t){acc[c][0][i], acc[c][1][i], acc[c][2][i], acc[c][3][i]};
        dst[1] = (float4_t){acc[c][4][i], acc[c][5][i], acc[c][6][i], acc[c][7][i]};
      }
    }
  __syncthreads();
  AWQ_STAMP(4);

  // cross-wave sum in fixed order, then either the output or this K-slice's fp32 slab.
  // One thread = 4 consecutive columns (float4) of one row m.
  const int Npad = n_ct * 512;
  const int nvec = M * 128;
  // buffer descriptor over the slab area (wave-uniform inputs only): lets the stores / loads carry sc1
  const __amdgpu_buffer_rsrc_t slab_rsrc = __builtin_amdgcn_make_buffer_rsrc(slabs, 0, (int)((size_t)S * M * Npad * sizeof(float)), 0x00020000);
  for (int v = threadIdx.x; v < nvec; v += kSkWaves * 64) {
    const int m = v >> 7, nl = (v & 127) * 4;
    const int off = m * kSkRowStride + (nl >> 5) * 36 + (nl & 31);
    float4_t acc = *(const float4_t*)(red + off);
#pragma unroll
    for (int w = 1; w < kSkWaves; ++w) acc += *(const float4_t*)(red + (size_t)w * M * kSkRowStride + off);
    const int n = ct * 512 + nl;
    if (S == 1 || ABL == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < N) store_output<DT>(y, (size_t)m * N + n + e, acc[e], bias, n + e);
    } else {
      // write-through (sc1) store: the slab leaves this XCD's L2, so no release fence is needed
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc), slab_rsrc,
                                             (unsigned)((((size_t)ks * M + m) * Npad + n) * sizeof(float)), 0, 16);
    }
  }
  AWQ_STAMP(5);
  if (S == 1 || ABL == 1) return;

  // Publish / ticket / reduce — the guide's counter-form hand-off with write-through payloads:
  // every storing wave drains its sc1 stores, the workgroup barrier orders them before ONE
  // relaxed agent-scope ticket; the workgroup that draws the last ticket reads every slab with sc1
  // loads (EVERY load of handed-off bytes bypasses the non-coherent caches), so neither a release
  // nor an acquire fence is needed and the result does not depend on workgroup placement.
  unsigned* ticket_word = (unsigned*)(red + (size_t)kSkWaves * M * kSkRowStride);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    *ticket_word = __hip_atomic_fetch_add(&counters[ct], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  AWQ_STAMP(6);
  if (*ticket_word != (unsigned)(S - 1)) return;
  if (threadIdx.x == 0)
    __hip_atomic_store(&counters[ct], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  // every slab of this column tile, summed in slice order (deterministic); loads are issued in
  // independent batches of 8 so the reducer pays one memory round trip per batch, not per slice
  const unsigned slab_stride_b = (unsigned)((size_t)M * Npad * sizeof(float));
  for (int v = threadIdx.x; v < nvec; v += kSkWaves * 64) {
    const int m = v >> 7, nl = (v & 127) * 4;
    const int n = ct * 512 + nl;
    if (n >= N) continue;
    const unsigned base = (unsigned)(((size_t)m * Npad + n) * sizeof(float));
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      u32x4_t t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, base + (unsigned)(s + u) * slab_stride_b, 0, 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += __builtin_bit_cast(float4_t, t[u]);
    }
    for (; s < S; ++s) acc += __builtin_bit_cast(float4_t, __builtin_amdgcn_raw_buffer_load_b128(slab_rsrc, base + (unsigned)s * slab_stride_b, 0, 16));
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (n + e < N) store_output<DT>(y, (size_t)m * N + n + e, acc[e], bias, n + e);
  }
  AWQ_STAMP(7);
}

// =================================================================================================
// ------------------------------------------------------------------------------------------ host
struct SkinnyPlan {
  int S, n_ct, steps_per_wave;
  size_t lds_bytes, slab_bytes;
};

static SkinnyPlan skinny_plan(int M, int K, int N, size_t slab_budget, int64_t tune) {
  SkinnyPlan p;
  p.n_ct = (N + 511) / 512;
  const int ksteps = K / 32;
  const int max_slices = (ksteps + kSkWaves - 1) / kSkWaves;
  // tune: bits 8-15 = S override (0 = heuristic)
  int S = (int)((tune >> 8) & 0xFF);
  if (S <= 0) {
    // one workgroup per CU (measured best at M <= 4: 242 workgroups, 3 k-steps per wave at the
    // 4096 x 11008 shape); with more rows the fp32 slabs grow with M, so fewer, longer slices
    S = kNumCUs / p.n_ct > 0 ? kNumCUs / p.n_ct : 1;
    if (M > 4 && S > 8) S = 8;
  }
  if (S < 1) S = 1;
  if (S > max_slices) S = max_slices;
  const size_t slab_row = (size_t)(M > 0 ? M : 1) * p.n_ct * 512 * sizeof(float);
  while (S > 1 && (size_t)S * slab_row > slab_budget) --S;
  p.steps_per_wave = (ksteps + S * kSkWaves - 1) / (S * kSkWaves);
  p.S = (ksteps + p.steps_per_wave * kSkWaves - 1) / (p.steps_per_wave * kSkWaves);   // drop empty trailing slices
  p.lds_bytes = (size_t)kSkWaves * M * kSkRowStride * sizeof(float) + 16;
  p.slab_bytes = p.S > 1 ? (size_t)p.S * slab_row : 0;
  return p;
}

bool skinny_supported(const GemmArgs& a) {
  if (a.dtype != AWQ_DTYPE_F16 && a.dtype != AWQ_DTYPE_BF16) return false;
  if (a.M < 1 || a.M > kSkinnyMaxM) return false;
  if (a.N % 32 || a.K % 32 || a.g % 32 || a.ldx % 8) return false;
  if ((a.N + 511) / 512 > (int)(kSkCounterBytes / sizeof(unsigned))) return false;
  if ((((uintptr_t)a.x) | ((uintptr_t)a.qweight) | ((uintptr_t)a.scales) | ((uintptr_t)a.qzeros)) & 15) return false;
  return true;
}

size_t skinny_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < 1 || M > kSkinnyMaxM) return kSkCounterBytes;
  const SkinnyPlan p = skinny_plan((int)M, (int)K, (int)N, kSkSlabBudget, 0);
  return kSkCounterBytes + p.slab_bytes;   // an S override through `tune` is clamped to the workspace passed in
}

template <int DT, int T, int ABL>
static void launch_T(const GemmArgs& a, const SkinnyPlan& p) {
  const int C = a.N / 8;
  dim3 grid(p.n_ct * p.S), block(kSkWaves * 64);
  unsigned* counters = (unsigned*)a.workspace;
  float* slabs = (float*)((char*)a.workspace + kSkCounterBytes);
  if (p.lds_bytes > 64 * 1024)   // dynamic LDS above 64 KiB (M >= 8) needs the ceiling raised (idempotent)
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<DT, T, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((gemm_skinny_kernel<DT, T, ABL>), grid, block, p.lds_bytes, a.stream, (const uint16_t*)a.x, a.ldx,
                     (const uint32_t*)a.qweight, (const uint16_t*)a.scales, (const uint32_t*)a.qzeros, a.bias, a.y, slabs,
                     counters, a.M, a.K, C, a.g, p.n_ct, p.S, p.steps_per_wave);
}

template <int DT, int ABL = 0>
static void launch_one(const GemmArgs& a, const SkinnyPlan& p) {
  switch (p.steps_per_wave) {
    case 1: launch_T<DT, 1, ABL>(a, p); break;
    case 2: launch_T<DT, 2, ABL>(a, p); break;
    case 3: launch_T<DT, 3, ABL>(a, p); break;
    case 4: launch_T<DT, 4, ABL>(a, p); break;
    default: launch_T<DT, 0, ABL>(a, p); break;
  }
}

int launch_gemm_skinny(const GemmArgs& a) {
  if (!skinny_supported(a)) return AWQ_ERR_BAD_VARIANT;
  if (a.workspace == nullptr || a.workspace_bytes < kSkCounterBytes) return AWQ_ERR_WORKSPACE;
  if (((uintptr_t)a.workspace) & 15) return AWQ_ERR_MISALIGNED;
  const SkinnyPlan p = skinny_plan(a.M, a.K, a.N, a.workspace_bytes - kSkCounterBytes, a.tune);
  if (p.lds_bytes > 160 * 1024) return AWQ_ERR_BAD_VARIANT;
  const int abl = (int)((a.tune >> 16) & 7);
  if (abl && a.dtype == AWQ_DTYPE_F16 && p.lds_bytes <= 64 * 1024) {   // timing-only builds (kbench): WRONG results for abl == 1
    if (abl == 1) launch_one<AWQ_DTYPE_F16, 1>(a, p);
    else {
      if (a.workspace_bytes < (60u << 20)) return AWQ_ERR_WORKSPACE;
      if (abl == 3) launch_one<AWQ_DTYPE_F16, 3>(a, p);
      else if (abl == 4) launch_one<AWQ_DTYPE_F16, 4>(a, p);
      else if (abl == 5) launch_one<AWQ_DTYPE_F16, 5>(a, p);
      else launch_one<AWQ_DTYPE_F16, 6>(a, p);
    }
    return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
  }
  if (a.dtype == AWQ_DTYPE_F16) launch_one<AWQ_DTYPE_F16>(a, p);
  else launch_one<AWQ_DTYPE_BF16>(a, p);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
