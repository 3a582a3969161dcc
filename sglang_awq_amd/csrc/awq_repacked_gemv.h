// Decode GEMV on the MFMA-fragment-major layout (see awq_repacked.hip for the layout): kernel template, the
// register-fit table and the (G, T) dispatch, shared by the plain launcher (awq_repacked.hip) and the fused
// variants of the decode harness (awq_repacked_fused.hip) so the two sets of instantiations build in parallel.
#pragma once

#include "awq_device.h"
#include "awq_kernels.h"

namespace awq {

constexpr int kRpMaxG = 8;
constexpr int kRpMaxLds = 144 * 1024;      // dynamic LDS a one-workgroup-per-CU variant may opt in to (of the CU's 160 KiB)
inline int rp_groups(int64_t N) { return (int)((N + 15) / 16); }

// diagnostic only (tools/kbench rstamps): when set, workgroups write 100 MHz-clock stamps into this device buffer
extern unsigned long long* g_rp_stamp_buffer;

// 8 weights of one column (one dword) -> the 4 packed k-pairs of an MFMA B fragment
__device__ __forceinline__ u32x4_t rp_dequant(uint32_t w, half2_t z1024, half2_t z64, half2_t s2) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t magic = kMagicF16;
  const uint32_t w8 = w >> 8;
  const half2_t d0 = as_h2(and_or(w, kLoNib, magic)) - z1024;
  const half2_t d1 = __builtin_elementwise_fma(as_h2(and_or(w, kHiNib, magic)), sixteenth, -z64);
  const half2_t d2 = as_h2(and_or(w8, kLoNib, magic)) - z1024;
  const half2_t d3 = __builtin_elementwise_fma(as_h2(and_or(w8, kHiNib, magic)), sixteenth, -z64);
  return (u32x4_t){as_u32(d0 * s2), as_u32(d1 * s2), as_u32(d2 * s2), as_u32(d3 * s2)};
}

constexpr int kRpMaxMT = 2;   // MFMA row tiles: M <= 16 * MT
struct RpBlock {            // one k-block (128 rows) of a strip, in registers
  u32x4_t w[kRpMaxG];       // per column group: 4 dwords = 4 k-steps
  uint32_t zs[kRpMaxG];     // per column group: (1024 + z | s) of this lane's column
  u32x4_t xa[4][kRpMaxMT];  // x fragments of the 4 k-steps, per row tile
};

template <int G, bool NT, int MT, bool LOADX = true>
__device__ __forceinline__ void rp_load(RpBlock& b, const u32x4_t* __restrict__ qw_r, const uint32_t* __restrict__ zs_r,
                                        const uint16_t* __restrict__ x, int64_t ldx, int cg0, int KB, int groups, int g, int kb,
                                        int lane, const int (&xr)[kRpMaxMT]) {
  const int q = lane >> 4, r = lane & 15;
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const u32x4_t* p = qw_r + ((size_t)(cg0 + c) * KB + kb) * 64 + lane;
    b.w[c] = NT ? __builtin_nontemporal_load(p) : *p;      // streamed once: keep it out of the caches' way
  }
  const int grp = (kb * 128) / g;                      // g >= 128 here (smaller groups take the per-k-step path below)
#pragma unroll
  for (int c = 0; c < G; ++c) b.zs[c] = zs_r[((size_t)(cg0 + c) * groups + grp) * 16 + r];
  if constexpr (LOADX) {
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) b.xa[d][mt] = *(const u32x4_t*)(x + (size_t)xr[mt] * ldx + kb * 128 + d * 32 + q * 8);
  }
}

template <int G, int MT>
__device__ __forceinline__ void rp_compute(const RpBlock& b, float4_t (&acc)[kRpMaxMT][kRpMaxG]) {
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
#pragma unroll
  for (int c = 0; c < G; ++c) {
    const half2_t s2 = as_h2(pack_lo16(b.zs[c], b.zs[c]));
    const half2_t z1024 = as_h2(pack_hi16(b.zs[c], b.zs[c]));
    const half2_t z64 = z1024 - c960;                  // exact: (1024 + z) - 960
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const u32x4_t frag = rp_dequant(b.w[c][d], z1024, z64, s2);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)   // the dequantised fragment is shared by every row tile
        acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, b.xa[d][mt]), __builtin_bit_cast(half8_t, frag), acc[mt][c], 0, 0, 0);
    }
  }
}

// Optional fusions for the decode harness (awq_aux.h: awq_aux_gemv_repacked_fused).
//   PRO = chunks of 8 halves per lane (1, 2 or 4; M * T * 16 <= 64 * PRO): x is not read; every wave builds
//         x = rmsnorm(h + delta) * w for the columns of its own k-blocks (see the prologue in the kernel);
//         workgroup 0 also stores h + delta.  Same arithmetic as add_rmsnorm_kernel: fp16 add, fp32 sum of squares, fp16(v * inv) * w.
//   PRO < 0: no norm, x itself staged compactly through wave-private LDS (-PRO chunks per lane); chosen by the plain
//         launcher where it lets a straight-line depth fit that would otherwise spill (see the kernel).
//   EPI = 1: column groups alternate gate / up (repacked from column-interleaved tensors); the epilogue writes
//         act = fp16(silu(fp16 gate)) * fp16 up, [M, N / 2], instead of y.
struct RpFuse {
  const half_t* h;
  const half_t* delta;
  const half_t* w;
  half_t* h_out;
  float eps;
};

// T = k-blocks per wave when it is small enough to issue every load up front (straight-line code, exact
// counted waits); T == 0: any count, double-buffered loop.
template <int G, int T, int W, bool NT, int MT, int PRO = 0, int EPI = 0>
__global__ __launch_bounds__(W * 64, (W * 64) / 256) void gemv_repacked_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                          const u32x4_t* __restrict__ qw_r,
                                                                          const uint32_t* __restrict__ zs_r,
                                                                          const void* __restrict__ bias, void* __restrict__ y,
                                                                          int M, int K, int N, int g, int NG, int per_wave,
                                                                          unsigned long long* __restrict__ dbg, RpFuse fz) {
  static_assert(PRO == 0 || (T > 0 && MT == 1), "x through LDS (norm prologue or not) exists for the straight-line single-tile variants");
  extern __shared__ __attribute__((aligned(16))) float red[];    // [W][M][16 G]
#define RP_STAMP(slot) do { if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
  RP_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g;
  int cg0 = blockIdx.x * G;
  if (cg0 + G > NG) cg0 = NG - G;                      // last strip overlaps its neighbour (same values written twice)
  int xr[kRpMaxMT];
#pragma unroll
  for (int mt = 0; mt < kRpMaxMT; ++mt) xr[mt] = mt * 16 + r < M ? mt * 16 + r : M - 1;

  float4_t acc[kRpMaxMT][kRpMaxG];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < G; ++c) acc[mt][c] = (float4_t){0.f, 0.f, 0.f, 0.f};

  const int kb_begin = wave * per_wave;
  int kb_end = kb_begin + per_wave;
  if (kb_end > KB) kb_end = KB;

  if constexpr (T > 0) {
    RpBlock buf[T];
    // PRO != 0: this wave's x columns live in wave-private LDS, [M][T * 128 (+8 halves: rows 16 B apart in bank phase)]
    half_t* const x_lds = (half_t*)(red + (size_t)W * M * 16 * G) + (size_t)wave * M * (T * 128 + 8);
    constexpr int XS = T * 128 + 8;
    if constexpr (PRO > 0) {
      // Norm prologue.  Each wave owns the columns of its own k-blocks (T x 128) of every row: it loads h, delta
      // and w for them, adds, and contributes per-row sums of squares; ONE workgroup barrier later every wave has
      // the row sums, issues its weight loads, and normalises its own columns under their latency (wave-private
      // LDS round trip into the MFMA fragment order, no second barrier).  Two rules found by measurement:
      //  * the prologue loads must complete before any weight load is issued: with the weight stream of every
      //    workgroup of the XCD queued in the L2 these few KB take ~3 us instead of ~0.4;
      //  * no workgroup barrier after the weight loads: load ISSUE of the later waves is throttled by the saturated
      //    memory pipeline, so a barrier there holds every wave until nearly all of the strip has arrived.
      typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
      constexpr int CH = PRO;                                            // chunks of 8 halves per lane: M * T * 16 <= 64 * CH
      constexpr int WC = T * 16;                                         // chunks per row in this wave's column range
      float* part = (float*)((half_t*)(red + (size_t)W * M * 16 * G) + (size_t)W * M * (T * 128 + 8));   // [M][W]
      const int col0 = kb_begin * 128;                                   // first column owned by this wave
      h8_t hv[CH], dv[CH], wv[CH];
      int crow[CH], ccol[CH];
      bool cok[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        crow[i] = c / WC;
        ccol[i] = (c - crow[i] * WC) * 8;                                // column offset inside the wave's range
        cok[i] = crow[i] < M && col0 + ccol[i] < K;
        const int gr = cok[i] ? crow[i] : 0, gc = cok[i] ? col0 + ccol[i] : 0;     // clamped: loaded, then masked
        hv[i] = *(const h8_t*)(fz.h + (size_t)gr * ldx + gc);
        dv[i] = *(const h8_t*)(fz.delta + (size_t)gr * ldx + gc);
        wv[i] = *(const h8_t*)(fz.w + gc);
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        hv[i] = hv[i] + dv[i];                                           // fp16 add, as the eager h = h + delta
        float ss = 0.f;
        if (cok[i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) ss += (float)hv[i][e] * (float)hv[i][e];
        }
        // a row's WC chunks sit in WC consecutive lanes (WC = 16, 32, 48 or 64): segmented sum over aligned groups of 16,
        // then the group leaders of one row add up in LDS order below
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if ((lane & 15) == 0) part[(wave * CH + i) * 4 + (lane >> 4)] = ss;   // [W][CH][4 sixteen-lane groups]
      }
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int kb = kb_begin + t < KB ? kb_begin + t : KB - 1;         // clamped: re-read, then weighted by x = 0
        rp_load<G, NT, MT, false>(buf[t], qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
      }
      __builtin_amdgcn_sched_barrier(0);
      RP_STAMP(1);
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (cok[i]) {
          // sum of squares of row crow[i]: every (wave, chunk slot, 16-lane group) that belongs to that row, fixed order
          float tot = 0.f;
          for (int w2 = 0; w2 < W; ++w2)
#pragma unroll
            for (int i2 = 0; i2 < CH; ++i2) {
              const float4_t p4 = *(const float4_t*)(part + (w2 * CH + i2) * 4);
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4)
                if ((g4 * 16 + i2 * 64) / WC == crow[i]) tot += p4[g4];
            }
          const float inv = __builtin_amdgcn_rsqf(tot / (float)K + fz.eps);
          h8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)hv[i][e] * inv) * wv[i][e];
          *(h8_t*)(x_lds + (size_t)crow[i] * XS + ccol[i]) = o;
          if (blockIdx.x == 0) *(h8_t*)(fz.h_out + (size_t)crow[i] * ldx + col0 + ccol[i]) = hv[i];
        }
      }
    } else if constexpr (PRO < 0) {
      // x through wave-private LDS, no norm: the MFMA A layout replicates a row over 16 lanes (16 registers of x per
      // k-block whatever M is); loaded compactly instead (CH dwordx4 per lane for the wave's own M x T x 128 halves,
      // requested BEFORE its weights so they return first) and re-read per k-block as fragments, a narrow strip holds
      // 5 G registers per k-block instead of 5 G + 16 — deep enough straight-line code for K = 11008 at 16 waves.
      typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
      constexpr int CH = -PRO, WC = T * 16;
      const int col0 = kb_begin * 128;
      h8_t xv[CH];
      int crow[CH], ccol[CH];
      bool cok[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        crow[i] = c / WC;
        ccol[i] = (c - crow[i] * WC) * 8;
        cok[i] = crow[i] < M && col0 + ccol[i] < K;
        xv[i] = *(const h8_t*)((const half_t*)x + (cok[i] ? (size_t)crow[i] * ldx + col0 + ccol[i] : 0));
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int kb = kb_begin + t < KB ? kb_begin + t : KB - 1;         // clamped: re-read, then weighted by x = 0
        rp_load<G, NT, MT, false>(buf[t], qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
      }
      __builtin_amdgcn_sched_barrier(0);
      RP_STAMP(1);
#pragma unroll
      for (int i = 0; i < CH; ++i)
        if (cok[i]) *(h8_t*)(x_lds + (size_t)crow[i] * XS + ccol[i]) = xv[i];
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int kb = kb_begin + t < KB ? kb_begin + t : KB - 1;         // clamped: re-read, then weighted by x = 0
        rp_load<G, NT, MT, true>(buf[t], qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
      }
      __builtin_amdgcn_sched_barrier(0);
      RP_STAMP(1);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if constexpr (PRO != 0) {                            // fragments of this k-block only: 16 registers live at a time
#pragma unroll
        for (int d = 0; d < 4; ++d) buf[t].xa[d][0] = *(const u32x4_t*)(x_lds + (size_t)xr[0] * XS + t * 128 + d * 32 + q * 8);
      }
      if (kb_begin + t >= KB) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) buf[t].xa[d][mt] = (u32x4_t){0u, 0u, 0u, 0u};
      }
      rp_compute<G, MT>(buf[t], acc);
      __builtin_amdgcn_sched_barrier(0);
      if (t == 0) RP_STAMP(2);
    }
  } else {
    // steady state has no branch between a load and its use (exact counted waits); only the prologue and the
    // <= 3-block tail are conditional
    RpBlock A, B;
    int kb = kb_begin;
    if (kb < kb_end) rp_load<G, NT, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
    if (kb + 1 < kb_end) rp_load<G, NT, MT>(B, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 1, lane, xr);
    while (kb + 3 < kb_end) {
      rp_compute<G, MT>(A, acc);
      rp_load<G, NT, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
      rp_compute<G, MT>(B, acc);
      rp_load<G, NT, MT>(B, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 3, lane, xr);
      kb += 2;
    }
    if (kb < kb_end) rp_compute<G, MT>(A, acc);
    if (kb + 2 < kb_end) rp_load<G, NT, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
    if (kb + 1 < kb_end) rp_compute<G, MT>(B, acc);
    if (kb + 2 < kb_end) rp_compute<G, MT>(A, acc);
  }

  RP_STAMP(3);
  // D[m = 4q + i][n = r] per column group -> LDS, summed over the waves in fixed order
  const int SW = 16 * G;
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
  RP_STAMP(4);
  if constexpr (EPI == 1) {
    // strip = G / 2 (gate, up) pairs of column groups; output column = 16 * (pair index) + r of act[M, N / 2]
    const int SH = 8 * G, I = N / 2;
    for (int idx = threadIdx.x; idx < M * SH; idx += W * 64) {
      const int m = idx / SH, c = idx - m * SH;
      const int pair = c >> 4, r16 = c & 15;
      const int n = (cg0 / 2 + pair) * 16 + r16;
      if (n >= I) continue;
      const int cgate = pair * 32 + r16;
      float gv = red[(size_t)m * SW + cgate], uv = red[(size_t)m * SW + cgate + 16];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        gv += red[((size_t)w * M + m) * SW + cgate];
        uv += red[((size_t)w * M + m) * SW + cgate + 16];
      }
      const float xg = (float)(half_t)gv;                                // the unfused path rounds gate_up to fp16 first
      ((half_t*)y)[(size_t)m * I + n] = (half_t)(xg / (1.f + __expf(-xg))) * (half_t)uv;
    }
  } else {
    for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
      const int m = idx / SW, col = idx - m * SW;
      const int n = cg0 * 16 + col;
      if (n >= N) continue;
      float v = red[(size_t)m * SW + col];
#pragma unroll
      for (int w = 1; w < W; ++w) v += red[((size_t)w * M + m) * SW + col];
      store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, v, bias, n);
    }
  }
}

// Which (waves, row tiles, strip width G, straight-line depth T) instantiations fit their register budget
// (128 VGPRs at 16 waves, 256 at 8) without scratch — from hipcc's -Rpass-analysis=kernel-resource-usage
// (tools/rp_resources.py prints the table).  A spilling variant is never built nor chosen: G = 3, T = 4 at
// 16 waves spills 52 B / lane and ran 8192 x 10240 at 18.1 us instead of 13.0.
constexpr bool rp_fits(int W, int MT, int G, int T) {
  if (MT == 2) return T == 0 || (T == 4 && G <= 5);
  if (W == 16) return G <= (T == 0 ? 5 : T <= 2 ? 8 : T == 3 ? 4 : T == 4 ? 2 : T == 5 ? 1 : 0);
  return G <= (T <= 4 ? 8 : T == 5 ? 6 : 4);
}


// The fused variants exist for 16 waves, non-temporal loads, one row tile, T = 1..4.  The norm prologue's x
// fragments arrive late (from LDS, after the weight loads were issued), so it needs no more registers than the
// plain kernel (tools/rp_resources.py lists them too).
constexpr bool rp_fits_xl(int G, int T);
constexpr bool rp_fits_fused(int G, int T, int PRO, int EPI) {
  if (EPI && (G & 1)) return false;
  if (PRO > 0) return rp_fits_xl(G, T) && !(PRO >= 2 && G == 8 && T == 2);      // x fragments come lazily from LDS: 5 G registers per k-block
  return T >= 1 && T <= 4 && rp_fits(16, 1, G, T);
}

// x staged through wave-private LDS (PRO < 0), 16 waves: T k-blocks of 5 G registers each (tools/rp_resources.py)
constexpr bool rp_fits_xl(int G, int T) { return T >= 1 && T <= 8 && G * T <= 16; }

template <int G, int W, bool NT, int MT, int PRO, int EPI>
static void rp_launch(const GemmArgs& a, const void* packed, int NG, int per_wave, int T, int nwg, size_t lds) {
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const RpFuse fz = {(const half_t*)a.norm_h, (const half_t*)a.norm_delta, (const half_t*)a.norm_w, (half_t*)a.norm_h_out, a.norm_eps};
  dim3 grid(nwg), block(W * 64);
#define RP_GO(TT)                                                                                                                     \
  if constexpr ((PRO == 0 && EPI == 0) ? (TT <= 6 && rp_fits(W, MT, G, TT) && (MT == 1 || TT == 0 || TT == 4))                           \
                : (PRO < 0)            ? (W == 16 && NT && MT == 1 && !(EPI && (G & 1)) && rp_fits_xl(G, TT))                            \
                : (MT == 2)            ? (PRO == 0 && W == 8 && NT && !(G & 1) && rp_fits(8, 2, G, TT))   /* SiLU-mul epilogue, 17..32 rows */ \
                                       : (W == 16 ? (NT && MT == 1 && rp_fits_fused(G, TT, PRO, EPI))                                            \
                                                 : (W == 8 && NT && MT == 1 && PRO == 0 && EPI == 1 && TT == 0 && G == 4))) { /* SiLU epilogue in rounds mode */ \
    auto kern = gemv_repacked_kernel<G, TT, W, NT, MT, PRO, EPI>;                                                                       \
    if (lds > 64 * 1024) {                       /* one workgroup per CU: opt in to more of its 160 KiB of LDS, once */                  \
      static const hipError_t once = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kRpMaxLds);     \
      (void)once;                                                                                                                       \
    }                                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, block, lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias, a.y, a.M, a.K, a.N, a.g, NG,   \
                       per_wave, g_rp_stamp_buffer, fz);                                                                                \
  }
  switch (T) {
    case 1: RP_GO(1); break;
    case 2: RP_GO(2); break;
    case 3: RP_GO(3); break;
    case 4: RP_GO(4); break;
    case 5: RP_GO(5); break;
    case 6: RP_GO(6); break;
    case 7: RP_GO(7); break;
    case 8: RP_GO(8); break;
    default: RP_GO(0); break;
  }
#undef RP_GO
}

template <int W, bool NT, int MT, int PRO = 0, int EPI = 0>
static void rp_launch_g(int G, const GemmArgs& a, const void* packed, int NG, int per_wave, int T, int nwg, size_t lds) {
  switch (G) {
    case 1: rp_launch<1, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 2: rp_launch<2, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 3: rp_launch<3, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 4: rp_launch<4, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 5: rp_launch<5, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 6: rp_launch<6, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    case 7: rp_launch<7, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
    default: rp_launch<8, W, NT, MT, PRO, EPI>(a, packed, NG, per_wave, T, nwg, lds); break;
  }
}

}  // namespace awq
