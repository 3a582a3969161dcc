// Skinny fused AWQ GEMM for decode shapes (M <= 16) on gfx950 — HBM-bound, so the design goal is
// to get every byte of qweight requested as early and as wide as possible and to keep the
// dequantised weights out of memory entirely.
//
// Decomposition
//   column tile   512 output columns = 16 lanes x one 16-byte load (4 packed words = 32 columns)
//   k-step        32 rows = 4 lane groups (q) x 8 rows (j); lane (q, r) loads rows k0+8q+j of
//                 chunk r, i.e. 8 global_load_dwordx4 per k-step and 256 contiguous bytes per row
//   wave          KT consecutive k-steps, all loads issued before the first use ("one shot":
//                 at M = 1 a wave only ever sees 1-3 k-steps of the 23 MB matrix, so the whole
//                 matrix is in flight across the chip almost immediately)
//   workgroup     4 waves on the same column tile and adjacent K ranges (x ROUNDS), summed in a
//                 fixed order through LDS
//   grid          n_ct column tiles x S K-slices; S > 1 partials go to fp32 slabs in the
//                 workspace and the last workgroup to arrive for a column tile (agent-scope
//                 release / ticket / acquire) adds them in slice order and writes y —
//                 deterministic, no float atomics, no second launch.
//
// MFMA mapping (v_mfma_f32_16x16x32_{f16,bf16}): A = x (row m = lane & 15, k = 8 (lane >> 4) + j),
// B = dequantised W (k = 8 q + j, column = lane & 15).  A lane's 8 rows x one logical column are
// exactly one B fragment, so after the packed dequantise (column pairs, see awq_device.h) two
// v_perm_b32 per register pair transpose them into k-pairs; nothing goes through LDS.  D[m][n]
// comes back with m = 4 q + i, n-lane = r.
#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kSkWaves = 4;
constexpr int kSkRowStride = 16 * 36;           // floats per (wave, m) row in LDS: 16 lanes x (32 + 4 pad)
constexpr size_t kSkCounterBytes = 4096;        // 1024 column-tile counters at the head of the workspace
constexpr size_t kSkSlabBudget = 32u << 20;     // default cap on fp32 partial slabs

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

template <int DT, int KT>
__global__ __launch_bounds__(kSkWaves * 64) void gemm_skinny_kernel(
    const uint16_t* __restrict__ x, int64_t ldx, const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
    const uint32_t* __restrict__ qz, const void* __restrict__ bias, void* __restrict__ y, float* __restrict__ slabs,
    unsigned* __restrict__ counters, int M, int K, int C, int g, int n_ct, int S, int rounds) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [kSkWaves][M][kSkRowStride] (+ 1 ticket word)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int ct = blockIdx.x % n_ct;
  const int ks = blockIdx.x / n_ct;
  const int N = C * 8;
  const int chunk4 = (ct * 16 + r) * 4;          // first packed word of this lane's 16-byte chunk
  const bool col_ok = chunk4 < C;
  const bool x_ok = r < M;
  float* my_red = red + (size_t)wave * M * kSkRowStride;

  for (int rd = 0; rd < rounds; ++rd) {
    // rows of this wave in this round: KT consecutive k-steps inside one quantisation group
    const int kbase = (((ks * rounds + rd) * kSkWaves + wave) * KT) * 32;
    const bool live = kbase < K;                 // wave-uniform (K % (32 KT) == 0 is a launch condition)

    u32x4_t R[KT][8];
    u32x4_t XA[KT];
    u32x4_t SC[4];
    u32x4_t ZW;
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    if (live) {
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const int k0 = kbase + t * 32 + 8 * q;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          R[t][j] = col_ok ? *(const u32x4_t*)(qw + (size_t)(k0 + j) * C + chunk4) : zero4;
        XA[t] = x_ok ? *(const u32x4_t*)(x + (size_t)r * ldx + k0) : zero4;
      }
      const int grp = kbase / g;
      ZW = col_ok ? *(const u32x4_t*)(qz + (size_t)grp * C + chunk4) : zero4;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        SC[c] = col_ok ? *(const u32x4_t*)(scales + (size_t)grp * N + (size_t)(chunk4 + c) * 8) : zero4;
    }

#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4_t acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = (float4_t){0.f, 0.f, 0.f, 0.f};
      if (live) {
        const ZeroC<DT> zc = make_zero<DT>(ZW[c]);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          uint32_t P[8][4];
#pragma unroll
          for (int j = 0; j < 8; ++j) dequant_pairs<DT>(R[t][j][c], zc, SC[c], P[j]);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
            const u32x4_t flo = {pack_lo16(P[0][tt], P[1][tt]), pack_lo16(P[2][tt], P[3][tt]),
                                 pack_lo16(P[4][tt], P[5][tt]), pack_lo16(P[6][tt], P[7][tt])};
            const u32x4_t fhi = {pack_hi16(P[0][tt], P[1][tt]), pack_hi16(P[2][tt], P[3][tt]),
                                 pack_hi16(P[4][tt], P[5][tt]), pack_hi16(P[6][tt], P[7][tt])};
            acc[2 * tt] = mfma16<DT>(XA[t], flo, acc[2 * tt]);
            acc[2 * tt + 1] = mfma16<DT>(XA[t], fhi, acc[2 * tt + 1]);
          }
        }
      }
      // D[m = 4q + i][n-lane r]: park the 8 columns of word c in this wave's private LDS rows
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = 4 * q + i;
        if (m < M) {
          float4_t* dst = (float4_t*)(my_red + (size_t)m * kSkRowStride + r * 36 + 8 * c);
          float4_t v0 = {acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
          float4_t v1 = {acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
          if (rd > 0) { v0 += dst[0]; v1 += dst[1]; }   // wave-private: no barrier needed
          dst[0] = v0;
          dst[1] = v1;
        }
      }
    }
  }
  __syncthreads();

  // cross-wave sum in fixed order, then either the output or this K-slice's fp32 slab
  const int Npad = n_ct * 512;
  for (int idx = threadIdx.x; idx < M * 512; idx += kSkWaves * 64) {
    const int m = idx >> 9, nl = idx & 511;
    const int off = m * kSkRowStride + (nl >> 5) * 36 + (nl & 31);
    float v = red[off];
#pragma unroll
    for (int w = 1; w < kSkWaves; ++w) v += red[(size_t)w * M * kSkRowStride + off];
    const int n = ct * 512 + nl;
    if (S == 1) {
      if (n < N) store_output<DT>(y, (size_t)m * N + n, v, bias, n);
    } else {
      slabs[((size_t)ks * M + m) * Npad + n] = v;
    }
  }
  if (S == 1) return;

  // publish the slab, take a ticket; the last arriver of this column tile reduces (guide §6 G16,
  // counter form).  Placement-independent: agent-scope release before the ticket, acquire after.
  unsigned* ticket_word = (unsigned*)(red + (size_t)kSkWaves * M * kSkRowStride);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *ticket_word = __hip_atomic_fetch_add(&counters[ct], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*ticket_word != (unsigned)(S - 1)) return;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(&counters[ct], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < M * 512; idx += kSkWaves * 64) {
    const int m = idx >> 9, nl = idx & 511;
    const int n = ct * 512 + nl;
    if (n >= N) continue;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += slabs[((size_t)s * M + m) * Npad + n];
    store_output<DT>(y, (size_t)m * N + n, v, bias, n);
  }
}

// ------------------------------------------------------------------------------------------ host
struct SkinnyPlan {
  int KT, rounds, S, n_ct;
  size_t lds_bytes, slab_bytes;
};

static SkinnyPlan skinny_plan(int M, int K, int N, int g, size_t slab_budget, int64_t tune) {
  SkinnyPlan p;
  p.n_ct = (N + 511) / 512;
  // tune: bits 0-3 = KT (1, 2, 4), bits 8-15 = S override, 0 = heuristic
  int kt = (int)(tune & 0xF);
  if (kt != 1 && kt != 2 && kt != 4) kt = 2;
  while (kt > 1 && (g % (32 * kt) != 0 || K % (32 * kt) != 0)) kt >>= 1;
  p.KT = kt;
  const int rows_per_round = kSkWaves * kt * 32;
  const int max_slices = (K + rows_per_round - 1) / rows_per_round;
  int S = (int)((tune >> 8) & 0xFF);
  if (S <= 0) {
    // enough workgroups to put ~2 on every CU, but keep slab traffic (write + read, fp32) under
    // ~1/8 of the packed-weight bytes: S * M * N * 8 <= K * N / 16
    const int want = (2 * 256 + p.n_ct - 1) / p.n_ct;
    const int traffic_cap = K / (128 * (M > 0 ? M : 1));
    S = want < traffic_cap ? want : traffic_cap;
  }
  if (S < 1) S = 1;
  if (S > max_slices) S = max_slices;
  const size_t slab_row = (size_t)(M > 0 ? M : 1) * p.n_ct * 512 * sizeof(float);
  while (S > 1 && (size_t)S * slab_row > slab_budget) --S;
  p.rounds = (max_slices + S - 1) / S;
  p.S = (max_slices + p.rounds - 1) / p.rounds;   // drop empty trailing slices
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
  const SkinnyPlan p = skinny_plan((int)M, (int)K, (int)N, 32, kSkSlabBudget, /*tune: KT=1*/ 1);
  // KT = 1 maximises the slice count the heuristic can pick; any tune override is clamped to the
  // workspace actually passed in.
  return kSkCounterBytes + p.slab_bytes;
}

template <int DT, int KT>
static void launch_one(const GemmArgs& a, const SkinnyPlan& p) {
  const int C = a.N / 8;
  dim3 grid(p.n_ct * p.S), block(kSkWaves * 64);
  unsigned* counters = (unsigned*)a.workspace;
  float* slabs = (float*)((char*)a.workspace + kSkCounterBytes);
  hipLaunchKernelGGL((gemm_skinny_kernel<DT, KT>), grid, block, p.lds_bytes, a.stream, (const uint16_t*)a.x, a.ldx,
                     (const uint32_t*)a.qweight, (const uint16_t*)a.scales, (const uint32_t*)a.qzeros, a.bias, a.y, slabs,
                     counters, a.M, a.K, C, a.g, p.n_ct, p.S, p.rounds);
}

int launch_gemm_skinny(const GemmArgs& a) {
  if (!skinny_supported(a)) return AWQ_ERR_BAD_VARIANT;
  if (a.workspace == nullptr || a.workspace_bytes < kSkCounterBytes) return AWQ_ERR_WORKSPACE;
  if (((uintptr_t)a.workspace) & 15) return AWQ_ERR_MISALIGNED;
  const SkinnyPlan p = skinny_plan(a.M, a.K, a.N, a.g, a.workspace_bytes - kSkCounterBytes, a.tune);
  if (p.lds_bytes > 160 * 1024) return AWQ_ERR_BAD_VARIANT;
  static bool attr_done = false;   // raise the dynamic-LDS ceiling once per process (idempotent)
  if (!attr_done) {
    const int big = 160 * 1024;
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_F16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_F16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_F16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_BF16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_BF16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    (void)hipFuncSetAttribute((const void*)gemm_skinny_kernel<AWQ_DTYPE_BF16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    attr_done = true;
  }
  if (a.dtype == AWQ_DTYPE_F16) {
    if (p.KT == 1) launch_one<AWQ_DTYPE_F16, 1>(a, p);
    else if (p.KT == 2) launch_one<AWQ_DTYPE_F16, 2>(a, p);
    else launch_one<AWQ_DTYPE_F16, 4>(a, p);
  } else {
    if (p.KT == 1) launch_one<AWQ_DTYPE_BF16, 1>(a, p);
    else if (p.KT == 2) launch_one<AWQ_DTYPE_BF16, 2>(a, p);
    else launch_one<AWQ_DTYPE_BF16, 4>(a, p);
  }
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
