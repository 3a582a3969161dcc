// Decode GEMV on the MFMA-fragment-major layout (see awq_repacked.hip for the layout): kernel template, the
// register-fit table and the (G, T) dispatch, shared by the plain launcher (awq_repacked.hip) and the fused
// variants of the decode harness (awq_repacked_fused.hip) so the two sets of instantiations build in parallel.
#pragma once

#include <cstdlib>

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
//   EPI = 1: column groups alternate gate / up (repacked from column-interleaved tensors); the epilogue writes
//         act = fp16(silu(fp16 gate)) * fp16 up, [M, N / 2], instead of y.
// (The RMSNorm in front of a linear is folded through gemv_rp2_kernel<NORM>; the round-1 form — a norm prologue in this kernel,
// every workgroup re-reading all of h + delta behind a barrier — measured 1.6 .. 6.5 us slower and was removed in round 3.)
struct RpFuse {
  const half_t* h;
  const half_t* delta;
  const half_t* w;
  half_t* h_out;
  float eps;
  // Expert indirection (AWQ-MoE decode, awq_aux_moe_gemv; gemv_rp2_kernel only): when expert_ids != nullptr the grid's y
  // dimension runs over (token, expert-slot) pairs; slot s uses the packed weight of expert expert_ids[s] (expert_stride
  // bytes apart), activation row s / x_div, output row s, and its fp32 sums are multiplied by slot_scale[s] (if given)
  // before the one rounding (the routed weight, applied where the reference's fused MoE applies it).
  const int* expert_ids;
  const float* slot_scale;
  long long expert_stride;
  int x_div;
  int num_experts;             // ids outside [0, num_experts) mark padded slots (the reference writes -1 for the padded tokens of a
                               // graph batch, layers/moe/topk.py:705-712): their output row is zero-filled, no weight is read
};

// Issue priority tied to a wave's progress (see gemv_rp2_kernel): done of total steps -> s_setprio 3 .. 0.  The waves that share a
// SIMD then finish together instead of one trailing the others into the workgroup's reduction.  RP_DYNPRIO=0: A/B build without.
#ifndef RP_DYNPRIO
#define RP_DYNPRIO 1
#endif
__device__ __forceinline__ void prio_by_progress(int done, int total) {   // wave-uniform arguments
#if RP_DYNPRIO
  const int b = done * 4;
  if (b < total) __builtin_amdgcn_s_setprio(3);
  else if (b < 2 * total) __builtin_amdgcn_s_setprio(2);
  else if (b < 3 * total) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// T = k-blocks per wave when it is small enough to issue every load up front (straight-line code, exact
// counted waits); T == 0: any count, double-buffered loop.
template <int G, int T, int W, bool NT, int MT, int EPI = 0>
__global__ __launch_bounds__(W * 64, (W * 64) / 256) void gemv_repacked_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                          const u32x4_t* __restrict__ qw_r,
                                                                          const uint32_t* __restrict__ zs_r,
                                                                          const void* __restrict__ bias, void* __restrict__ y,
                                                                          int M, int K, int N, int g, int NG, int per_wave,
                                                                          unsigned long long* __restrict__ dbg, RpFuse fz) {
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
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int kb = kb_begin + t < KB ? kb_begin + t : KB - 1;         // clamped: re-read, then weighted by x = 0
      rp_load<G, NT, MT, true>(buf[t], qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb, lane, xr);
    }
    __builtin_amdgcn_sched_barrier(0);
    RP_STAMP(1);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (kb_begin + t >= KB) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) buf[t].xa[d][mt] = (u32x4_t){0u, 0u, 0u, 0u};
      }
      prio_by_progress(t, T);
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
      prio_by_progress(kb - kb_begin, per_wave);
      rp_compute<G, MT>(A, acc);
      rp_load<G, NT, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
      rp_compute<G, MT>(B, acc);
      rp_load<G, NT, MT>(B, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 3, lane, xr);
      kb += 2;
    }
    prio_by_progress(kb - kb_begin, per_wave);
    if (kb < kb_end) rp_compute<G, MT>(A, acc);
    if (kb + 2 < kb_end) rp_load<G, NT, MT>(A, qw_r, zs_r, x, ldx, cg0, KB, groups, g, kb + 2, lane, xr);
    if (kb + 1 < kb_end) rp_compute<G, MT>(B, acc);
    if (kb + 2 < kb_end) rp_compute<G, MT>(A, acc);
  }
  __builtin_amdgcn_s_setprio(0);

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

// ------------------------------------------------------------------------------------------------------------------
// gemv_rp2_kernel — the straight-line decode GEMV for M <= 16 on 16-wave workgroups (one strip of G column groups per
// workgroup, T k-blocks per wave), restructured after per-wave stamps and issue-rate measurements (tools/gemv_lab,
// profiles/r02_gemv_lab_*.txt) showed the previous form (gemv_repacked_kernel with x through LDS) to be bound by vector-ALU
// issue, not by HBM: a wave's 13 VALU + 1 MFMA per packed dword cost ~70 cycles per dword when it runs alone on its SIMD and
// ~48 per dword per SIMD with four waves active, against ~27 GB/s per CU of arriving weights; and the CU admits only ~24 KiB
// of loads in flight, so with every load issued up front the four waves of a SIMD were served, and computed, one after
// the other.  Changes:
//   * every vector-memory instruction in the queue is a full 1 KiB weight load: x (M x T x 128 halves) and the wave's
//     (scale | 1024 + zero) words (G x T x 16 dwords) are fetched by the same CHS <= 8 per-lane-addressed loads and parked
//     in wave-private LDS (no barrier), instead of 1 + G T separate small loads per wave;
//   * weight loads go through a ring of D registers sets per wave (D = 2: two 1 KiB loads in flight per wave, 32 KiB per
//     CU, the next one issued before the current one is dequantised), and the first two issue rounds are separated by
//     workgroup barriers so the CU's queue holds round k of all sixteen waves before round k + 1 of any: all waves of a
//     SIMD progress together (4096 x 11008, M = 1: 6.73 -> 6.37 us in the lab harness);
//   * scale and zero are broadcast into the packed ops through op_sel (no v_perm), masks / magic are opaque compiler-visible
//     constants (v_and_or_b32 without asm boundary pads), addresses are one wave-uniform base + one per-lane offset:
//     ~530 instead of ~610 instructions per wave at G = 3, T = 2.
// Same arithmetic and the same summation order as gemv_repacked_kernel (wave w owns k-blocks [w T, w T + T); partial
// sums added in wave order through LDS), so results are bit-identical to it.
// M1: specialisation for one row (M == 1): no row arithmetic in the staging addresses, A fragments broadcast from row 0,
// one result element per column; CHS is then a function of (G, T) alone.
// NORM: the RMSNorm(+residual) in front of the linear (models/llama.py:277-290), folded through the GEMV's linearity:
//   y = (rmsnorm(v) * w) W = inv_rms(v) * ((v * w) W),   v = h + delta.
// A wave stages x' = fp16(v) * w for its own k-blocks (no dependence on other waves, no barrier before the weight loads),
// adds up its share of sum(v^2) on the side, and the epilogue — which already meets all waves in LDS — multiplies the fp32
// sums by inv_rms = rsqrt(sum(v^2) / K + eps) before the one rounding to fp16.  Against the eager order
// fp16(fp16(v * inv) * w) this moves where x is rounded (by at most an fp16 ulp of x per element; the tests restate this
// order in the oracle and bound the distance to the eager order); workgroup 0 stores v = h + delta (fp16 add, bit-exact).
// x' is staged as v * (w / 64) and the epilogue multiplies by 64 inv_rms: exact (powers of two) wherever v w and w / 64 are normal
// fp16 numbers, and finite for |v w| up to 4e6 — residual streams with massive activations (1e3..1e4 in a few channels of
// Llama-family models) times a norm weight above 1 would overflow the un-scaled product.
// The earlier norm prologue (round 1, inside gemv_repacked_kernel; removed) needed every workgroup to re-read all of h + delta and one
// workgroup barrier before its weight loads: +1.6 us at M = 1, +6.5 at M = 8; this form costs two extra staging loads.
#ifndef RP2_CMAJOR
#define RP2_CMAJOR 1          // unit i = (c = i / T, t = i % T): a wave's consecutive loads are contiguous; 0: t-major (round 2; same speed,
#endif                        // profiles/r03_kbench_next_hint_ab.txt)
#ifndef RP2_DYNPRIO
#define RP2_DYNPRIO 1         // a wave's issue priority falls as it gets through its units (s_setprio 3 .. 0); 0: A/B build, the static form of
#endif                        // round 2 (the later waves of a SIMD raised once).  profiles/r03_kbench_wave_priority_ab.txt
#ifndef RP2_EARLY
#define RP2_EARLY 1           // 0: A/B build, the replacement load is issued behind the unit's compute (round-2 first form)
#endif
template <int G, int T, int CHS, int D, int EPI, bool M1, bool NORM = false>
__global__ __launch_bounds__(1024) void gemv_rp2_kernel(const uint16_t* __restrict__ x, int64_t ldx, const u32x4_t* __restrict__ qw_r,
                                                        const uint32_t* __restrict__ zs_r, const void* __restrict__ bias,
                                                        void* __restrict__ y, int M, int K, int N, int groups, int gmul,
                                                        int gshift, int NG, RpFuse fz) {
  constexpr int W = 16;
  constexpr int L = G * T;
  constexpr int DD = (D == 0 || D > L) ? L : D;
  constexpr int XS = T * 128 + 8;                        // halves per staged x row (+8: rows 16 B apart in bank phase)
  extern __shared__ __attribute__((aligned(16))) float red[];    // [W][M][16 G] floats, then per wave: x rows, zs words
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128;                                // quantisation group of k-block kb: (kb * gmul) >> gshift (= kb / (g / 128), kb < 4096)
  int cg0 = blockIdx.x * G;
  if (cg0 + G > NG) cg0 = NG - G;                        // last strip overlaps its neighbour (same values written twice)
  const int kb0 = wave * T;
  if constexpr (M1) M = 1;
  float slot_scale = 1.f;
  if (fz.expert_ids != nullptr) {                        // AWQ-MoE: one (token, expert) pair per grid row (wave-uniform)
    const int slot = blockIdx.y;
    const int eid = fz.expert_ids[slot];
    if (eid < 0 || eid >= fz.num_experts) {             // padded slot (uniform over the workgroup: no barrier is left half-entered)
      const int width = EPI == 1 ? N / 2 : N, per = EPI == 1 ? 8 * G : 16 * G;
      const int n0 = EPI == 1 ? (cg0 / 2) * 16 : cg0 * 16;
      half_t* yr = (half_t*)y + (size_t)slot * width;
      for (int i = threadIdx.x; i < per; i += W * 64)
        if (n0 + i < width) yr[n0 + i] = (half_t)0.f;
      return;
    }
    const long long eoff = (long long)eid * fz.expert_stride;
    qw_r = (const u32x4_t*)((const unsigned char*)qw_r + eoff);
    zs_r = (const uint32_t*)((const unsigned char*)zs_r + eoff);
    x += (size_t)(slot / fz.x_div) * ldx;
    y = (unsigned char*)y + (size_t)slot * (EPI == 1 ? N / 2 : N) * 2;
    if (fz.slot_scale != nullptr) slot_scale = fz.slot_scale[slot];
  }
  const int xbytes = M * XS * 2;                         // staged x of one wave
  unsigned char* const stg = (unsigned char*)(red + (size_t)W * M * 16 * G) + (size_t)wave * (xbytes + G * T * 64 + 16);

  // staging chunks (16 B): 16 M T chunks of x (row, 8-half column chunk) and 4 G T chunks of zs words of (c, t).  M == 1:
  // one id space, lanes take ids lane + 64 i (x first, then zs) -> a single load at G = 3, T = 2.  M > 1: CHS loads of x
  // chunks, then one load of the (<= 64) zs chunks — no per-lane select between the two address computations.
  constexpr int NZ = G * T * 4, CHT = (M1 && !NORM) ? CHS : CHS + 1;
  const int nx = M * T * 16;
  u32x4_t sv[CHT];
  int sdst[CHT];
  const int dump = xbytes + G * T * 64;                  // 16-byte slot for lanes without a chunk: the ds_write stays unconditional
  auto zs_chunk = [&](int j, const unsigned char*& src, int& dst) {       // j in [0, NZ)
    const int c = j / (T * 4), rem = j - c * (T * 4), t = rem >> 2, part = rem & 3;
    int kbz = kb0 + t;
    kbz = kbz < KB ? kbz : KB - 1;
    src = (const unsigned char*)(zs_r + ((size_t)(cg0 + c) * groups + ((kbz * gmul) >> gshift)) * 16 + part * 4);
    dst = xbytes + ((c * T + t) * 16 + part * 4) * 4;
  };
  typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
  h8_t nh[NORM ? CHS : 1], nd[NORM ? CHS : 1], nw[NORM ? CHS : 1];        // NORM: h, delta, w chunks in place of x
  bool nin[NORM ? CHS : 1];
  float* const norm_part = (float*)((unsigned char*)(red + (size_t)W * M * 16 * G) + (size_t)W * (xbytes + G * T * 64 + 16));   // [W][M][T], then inv[M]
  if constexpr (NORM) {
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const int id = lane + 64 * i;
      const int row = id / (T * 16), cc = id - row * (T * 16);
      nin[i] = id < nx && (kb0 * 128 + cc * 8 < K);
      const size_t off = nin[i] ? (size_t)row * ldx + (size_t)kb0 * 128 + cc * 8 : 0;
      nh[i] = *(const h8_t*)(fz.h + off);
      nd[i] = *(const h8_t*)(fz.delta + off);
      nw[i] = *(const h8_t*)(fz.w + (nin[i] ? kb0 * 128 + cc * 8 : 0));
      sdst[i] = id < nx ? (row * XS + cc * 8) * 2 : dump;
    }
    const unsigned char* sz;
    int dz;
    zs_chunk(lane < NZ ? lane : NZ - 1, sz, dz);
    sv[CHS] = *(const u32x4_t*)sz;
    sdst[CHS] = lane < NZ ? dz : dump;
  } else if constexpr (M1) {
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const int id = lane + 64 * i;
      const bool isx = id < T * 16;
      int j = id - T * 16;
      j = j < 0 ? 0 : (j >= NZ ? NZ - 1 : j);
      const unsigned char* sz;
      int dz;
      zs_chunk(j, sz, dz);
      // k-blocks past K (ragged last wave) are whole k-blocks: their x chunks are read from column 0 instead (in bounds) and
      // their scale is forced to 0 after staging — no select on the loaded value, which would wait for the load right here
      const bool xin = isx && (kb0 * 128 + id * 8 < K);
      const unsigned char* sx = (const unsigned char*)(x + (xin ? (size_t)kb0 * 128 + id * 8 : 0));
      sv[i] = *(const u32x4_t*)(isx ? sx : sz);
      sdst[i] = id >= T * 16 + NZ ? dump : isx ? id * 16 : dz;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const int id = lane + 64 * i;
      const int row = id / (T * 16), cc = id - row * (T * 16);
      const bool xin = id < nx && (kb0 * 128 + cc * 8 < K);
      sv[i] = *(const u32x4_t*)(x + (xin ? (size_t)row * ldx + (size_t)kb0 * 128 + cc * 8 : 0));
      sdst[i] = id < nx ? (row * XS + cc * 8) * 2 : dump;
    }
    const unsigned char* sz;
    int dz;
    zs_chunk(lane < NZ ? lane : NZ - 1, sz, dz);
    sv[CHS] = *(const u32x4_t*)sz;
    sdst[CHS] = lane < NZ ? dz : dump;
  }
  // weights: wave-uniform base + lane offset; k-blocks past K are clamped (re-read, weighted by x = 0)
  const uint32_t loff = (uint32_t)lane * 16u;
  // DD loads in flight per wave, DD + 1 register sets: the load that replaces unit i is issued the moment unit i has arrived,
  // BEFORE its 4 x (13 VALU + MFMA) are computed (behind them it would wait ~300 cycles of this wave's issue time, up to four
  // times that with the SIMD's other waves in the way, and the wave would sit at one load in flight meanwhile)
  constexpr int RB = (RP2_EARLY && M1 && DD < L) ? DD + 1 : DD;      // (one-row form only: at M = 4 / 16 it measured 1 % slower, at M = 1 0-3 % faster)
  u32x4_t wbuf[RB];
  auto load_w = [&](int i) {
    const int t = RP2_CMAJOR ? i % T : i / G, c = RP2_CMAJOR ? i / T : i - t * G;
    int kb = kb0 + t;
    kb = kb < KB ? kb : KB - 1;
    const unsigned char* p = (const unsigned char*)(qw_r + ((size_t)(cg0 + c) * KB + kb) * 64) + loff;
    wbuf[i % RB] = __builtin_nontemporal_load((const u32x4_t*)p);      // streamed once: keep it out of the caches' way
  };
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    load_w(i);
    if (D != 0 && i < 2) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NORM) {
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const int id = lane + 64 * i;
      const int row = id / (T * 16), cc = id - row * (T * 16);
      const h8_t v = nh[i] + nd[i];                                      // fp16 add, as the eager h = h + delta
      if (blockIdx.x == 0 && nin[i]) *(h8_t*)(fz.h_out + (size_t)row * ldx + (size_t)kb0 * 128 + cc * 8) = v;
      // x' = v * (w / 64) (fp16 products; the power-of-two pre-scale is exact and keeps |x'| inside fp16 for residual streams with
      // massive activations: |v w| up to 4e6 instead of 65504); inv_rms * 64 comes in the epilogue
      const h8_t wsc = nw[i] * (half_t)0.015625f;
      sv[i] = __builtin_bit_cast(u32x4_t, v * wsc);
      float ss = 0.f;
      if (nin[i]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
      }
      // the 16 chunks of one (row, k-block) sit in one aligned group of 16 lanes: segmented sum, the group leader keeps it
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
      if ((lane & 15) == 0 && id < nx) norm_part[((size_t)wave * M + row) * T + (cc >> 4)] = ss;
    }
  }
#pragma unroll
  for (int i = 0; i < CHT; ++i) *(u32x4_t*)(stg + sdst[i]) = sv[i];
  if (kb0 + T > KB) {
    // ragged last wave (wave-uniform, rare): a k-block past K must contribute nothing.  Its weights were re-read from the last
    // valid k-block and its x from column 0 (both finite); overwrite its staged (scale | 1024 + zero) words with scale 0, so
    // every dequantised value is exactly 0.  A second, predicated LDS store behind the first: no select on loaded data.
    if (lane < NZ) {
      const int c = lane / (T * 4), rem = lane - c * (T * 4), t = rem >> 2, part = rem & 3;
      if (kb0 + t >= KB) *(u32x4_t*)(stg + xbytes + ((c * T + t) * 16 + part * 4) * 4) = (u32x4_t){0x64000000u, 0x64000000u, 0x64000000u, 0x64000000u};
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // VALU issue is arbitrated by priority, then age: the later waves of a SIMD (w >> 2 = 1, 2, 3) otherwise get the leftover
  // slots and finish last.  Round 2 raised the later half once (A/B build below); round 3 ties priority to progress (in the loop)
#if RP2_DYNPRIO == 0
  if (wave >= 12) __builtin_amdgcn_s_setprio(2);
  else if (wave >= 8) __builtin_amdgcn_s_setprio(1);
#endif
  uint32_t mlo = kLoNib, mhi = kHiNib, magic = kMagicF16;
  asm volatile("" : "+s"(mlo), "+s"(mhi));               // opaque: (w & m) | magic then selects v_and_or_b32 (one literal each)
  asm volatile("" : "+v"(magic));
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const half_t* x_lds = (const half_t*)stg + (M1 ? 0 : (size_t)(r < M ? r : M - 1) * XS);   // (M1: row 0 in either chunk layout)
  const uint32_t* zs_lds = (const uint32_t*)(stg + xbytes) + r;
  float4_t acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c) acc[c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t xa[4];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int t = RP2_CMAJOR ? i % T : i / G, c = RP2_CMAJOR ? i / T : i - t * G;
    if (RP2_CMAJOR ? (T > 1 || i == 0) : c == 0) {
#pragma unroll
      for (int d = 0; d < 4; ++d) xa[d] = *(const u32x4_t*)(x_lds + t * 128 + d * 32 + q * 8);
    }
#if RP2_DYNPRIO == 1
    // Four waves share a SIMD's issue port, arbitrated by priority, then age.  With fixed priorities some wave of the four always runs
    // ahead and one trails (stamps: the second-oldest wave finished its units 1.0 us after the youngest, raised one), and the trailing wave
    // cannot fill the port alone while the workgroup's reduction waits for it.  Priority tied to progress is a feedback loop: whoever is
    // behind outranks whoever is ahead, the four finish together.  4096 x 11008: 6.62 -> 6.32 us at one row, 8.05 -> 7.29 at eight.
    switch (3 - (i * 4) / L) {
      case 3: __builtin_amdgcn_s_setprio(3); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
#endif
    const half2_t zh = as_h2(zs_lds[(c * T + t) * 16]);
    const half2_t s2 = __builtin_shufflevector(zh, zh, 0, 0);          // folded into op_sel of the packed ops
    const half2_t z1024 = __builtin_shufflevector(zh, zh, 1, 1);
    const half2_t z64 = z1024 - c960;                                    // exact: (1024 + z) - 960
    u32x4_t w = wbuf[i % RB];
    if (RB > DD && i + DD < L) {
      asm volatile("" : "+v"(w));                        // unit i has arrived (the wait sits here) ...
      __builtin_amdgcn_sched_barrier(0);
      load_w(i + DD);                                    // ... its replacement leaves at once, into the spare register set
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const uint32_t ww = w[d], w8 = ww >> 8;
      const half2_t d0 = as_h2((ww & mlo) | magic) - z1024;
      const half2_t d1 = __builtin_elementwise_fma(as_h2((ww & mhi) | magic), sixteenth, -z64);
      const half2_t d2 = as_h2((w8 & mlo) | magic) - z1024;
      const half2_t d3 = __builtin_elementwise_fma(as_h2((w8 & mhi) | magic), sixteenth, -z64);
      const u32x4_t frag = {as_u32(d0 * s2), as_u32(d1 * s2), as_u32(d2 * s2), as_u32(d3 * s2)};
      acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, xa[d]), __builtin_bit_cast(half8_t, frag), acc[c], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (RB == DD && i + DD < L) {
      load_w(i + DD);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // D[m = 4q + i][n = r] per column group -> LDS, summed over the waves in fixed order
  const int SW = 16 * G;
  if constexpr (M1) {
    if (q == 0) {
#pragma unroll
      for (int c = 0; c < G; ++c) red[wave * SW + c * 16 + r] = acc[c][0];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < M) {                                       // wave-uniform
        const int m = 4 * q + i;
#pragma unroll
        for (int c = 0; c < G; ++c)
          if (m < M) red[((size_t)wave * M + m) * SW + c * 16 + r] = acc[c][i];
      }
    }
  }
  __syncthreads();
  float* const inv_rms = norm_part + (size_t)W * M * T;                  // [M]
  if constexpr (NORM) {
    if ((int)threadIdx.x < M) {
      float tot = 0.f;
      for (int w = 0; w < W; ++w)
        for (int t = 0; t < T; ++t) tot += norm_part[((size_t)w * M + threadIdx.x) * T + t];      // fixed order
      inv_rms[threadIdx.x] = __builtin_amdgcn_rsqf(tot / (float)K + fz.eps) * 64.f;       // (undoes the staging pre-scale)
    }
    __syncthreads();
  }
  if constexpr (EPI == 1) {
    // strip = G / 2 (gate, up) pairs of column groups; output column = 16 * (pair index) + r of act[M, N / 2]
    const int SH = 8 * G, I = N / 2;
    for (int idx = threadIdx.x; idx < M * SH; idx += W * 64) {
      const int m = M1 ? 0 : idx / SH, c = idx - m * SH;
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
      if constexpr (NORM) { gv *= inv_rms[m]; uv *= inv_rms[m]; }
      gv *= slot_scale; uv *= slot_scale;
      const float xg = (float)(half_t)gv;                                // the unfused path rounds gate_up to fp16 first
      ((half_t*)y)[(size_t)m * I + n] = (half_t)(xg / (1.f + __expf(-xg))) * (half_t)uv;
    }
  } else {
    for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
      const int m = M1 ? 0 : idx / SW, col = idx - m * SW;
      const int n = cg0 * 16 + col;
      if (n >= N) continue;
      float v = red[(size_t)m * SW + col];
#pragma unroll
      for (int w = 1; w < W; ++w) v += red[((size_t)w * M + m) * SW + col];
      if constexpr (NORM) v *= inv_rms[m];
      v *= slot_scale;                                   // 1.0 outside the MoE route: exact
      store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, v, bias, n);
    }
  }
}

// rp2 exists for T <= 8, G <= 8, G T <= 16 (unrolled length), staging of <= 8 chunks per lane; returns false when the
// (G, T, chunks) combination has no instantiation (caller falls back to gemv_repacked_kernel)
constexpr bool rp2_fits(int G, int T) { return G >= 1 && G <= kRpMaxG && T >= 1 && T <= 8 && G * T <= 16; }
// One row, plain operator: up to 32 units per wave (deep K at 16 waves — 70B down_proj, T = 14 — and wide strips on deep K — the
// 8192 x 28672 matrix, G = 7, T = 4); straight-line code of G T units, the same ring and staging.
constexpr bool rp2_fits_long(int G, int T) { return G >= 1 && G <= kRpMaxG && T >= 1 && T <= 16 && G * T <= 32 && !rp2_fits(G, T); }
inline int rp2_chunks(int M, int G, int T) { (void)G; const int n = (M * T * 16 + 63) / 64; return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 0; }   // x chunks per lane, M > 1
inline size_t rp2_lds(int M, int G, int T, bool norm = false) {
  return (size_t)16 * M * 16 * G * sizeof(float) + (size_t)16 * ((size_t)M * (T * 128 + 8) * 2 + (size_t)G * T * 64 + 16) +
         (norm ? ((size_t)16 * M * T + M) * sizeof(float) : 0);
}

template <int G, int T, int EPI, bool NORM>
static bool rp2_launch_t(const GemmArgs& a, const void* packed, int NG, int chunks, int depth, int nwg, size_t lds) {
  if constexpr ((!rp2_fits(G, T) && !(rp2_fits_long(G, T) && EPI == 0 && !NORM)) || (EPI == 1 && (G & 1))) {
    return false;
  } else {
    const u32x4_t* qw_r = (const u32x4_t*)packed;
    const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
    const int gk = a.g / 128;                            // k-blocks per quantisation group; kb / gk as a multiply-shift, exact for kb < 4096
    int lg = 0;
    while ((1 << lg) < gk) ++lg;
    const int gshift = 12 + lg, gmul = (int)(((1ll << gshift) + gk - 1) / gk);
    const RpFuse fz = {(const half_t*)a.norm_h, (const half_t*)a.norm_delta, (const half_t*)a.norm_w, (half_t*)a.norm_h_out, a.norm_eps,
                      a.moe_expert_ids, a.moe_slot_scale, (long long)a.moe_expert_stride, a.moe_x_div, a.moe_num_experts};
#define RP2_GO(CHS, DEP, ONE, NRM)                                                                                                 \
    do {                                                                                                                           \
      auto kern = gemv_rp2_kernel<G, T, CHS, DEP, EPI, ONE, NRM>;                                                                  \
      static unsigned long long opted[2] = {0ull, 0ull};                                                                           \
      if (lds > 64 * 1024 && !opt_in_dynamic_lds((const void*)kern, kRpMaxLds, opted)) return false;                                \
      hipLaunchKernelGGL(kern, dim3(nwg, a.moe_slots > 0 ? a.moe_slots : 1), dim3(1024), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, \
                         a.bias, a.y, a.M, a.K, a.N, a.K / a.g, gmul, gshift, NG, fz);                                              \
      return true;                                                                                                                 \
    } while (0)
    static const bool env_m1 = lab_env("AWQ_RP2_M1", 1) != 0;   // lab knob
    if constexpr (!rp2_fits(G, T)) {                     // the long form exists for one row only
      constexpr int C1 = (T * 16 + G * T * 4 + 63) / 64;            // <= 6 for G T <= 32, T <= 16
      constexpr int CH1 = C1 <= 1 ? 1 : C1 <= 2 ? 2 : C1 <= 4 ? 4 : 8;
      if (a.M == 1 && a.moe_slots == 0) {
#ifdef AWQ_LAB
        if (depth == 3) RP2_GO(CH1, 3, true, false);       // ring depth A/B (AWQ_RP2_D)
        if (depth == 4) RP2_GO(CH1, 4, true, false);
        if (depth == 6) RP2_GO(CH1, 6, true, false);
#endif
        RP2_GO(CH1, 2, true, false);
      }
    } else if constexpr (NORM) {                                // x chunk layout of the generic form at every M (chunks = x chunks per lane)
      if (a.M == 1 && chunks == 1) RP2_GO(1, 2, true, true);
      if (chunks == 1) RP2_GO(1, 2, false, true);
      if (chunks == 2) RP2_GO(2, 2, false, true);
      if (chunks == 4) RP2_GO(4, 2, false, true);
      // (8 chunks per lane would hold 3 x 8 staged vectors beside the weight ring: over the 128-register budget of a 16-wave
      // workgroup — tools/rp_resources.py showed scratch; those batches run the norm as its own launch)
    } else {
      if (a.M == 1 && env_m1) {
        constexpr int C1 = (T * 16 + G * T * 4 + 63) / 64;          // <= 5 for G T <= 16
        constexpr int CH1 = C1 <= 1 ? 1 : C1 <= 2 ? 2 : C1 <= 4 ? 4 : 8;
        if (depth == 0 && EPI == 0) RP2_GO(CH1, 0, true, false);
        RP2_GO(CH1, 2, true, false);
      }
      if (depth == 0 && EPI == 0) {
        if (chunks == 1) RP2_GO(1, 0, false, false);
        if (chunks == 2) RP2_GO(2, 0, false, false);
        if (chunks == 4) RP2_GO(4, 0, false, false);
        if (chunks == 8) RP2_GO(8, 0, false, false);
      } else {
        if (chunks == 1) RP2_GO(1, 2, false, false);
        if (chunks == 2) RP2_GO(2, 2, false, false);
        if (chunks == 4) RP2_GO(4, 2, false, false);
        if (chunks == 8) RP2_GO(8, 2, false, false);
      }
    }
#undef RP2_GO
    return false;
  }
}

template <int G, int EPI, bool NORM>
static bool rp2_launch_g(int T, const GemmArgs& a, const void* packed, int NG, int chunks, int depth, int nwg, size_t lds) {
  switch (T) {
    case 1: return rp2_launch_t<G, 1, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 2: return rp2_launch_t<G, 2, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 3: return rp2_launch_t<G, 3, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 4: return rp2_launch_t<G, 4, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 5: return rp2_launch_t<G, 5, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 6: return rp2_launch_t<G, 6, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 7: return rp2_launch_t<G, 7, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    case 8: return rp2_launch_t<G, 8, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
    default: break;
  }
  if constexpr (EPI == 0 && !NORM) {
    switch (T) {
      case 9: return rp2_launch_t<G, 9, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 10: return rp2_launch_t<G, 10, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 11: return rp2_launch_t<G, 11, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 12: return rp2_launch_t<G, 12, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 13: return rp2_launch_t<G, 13, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 14: return rp2_launch_t<G, 14, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 15: return rp2_launch_t<G, 15, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      case 16: return rp2_launch_t<G, 16, EPI, NORM>(a, packed, NG, chunks, depth, nwg, lds);
      default: break;
    }
  }
  return false;
}

// launches gemv_rp2_kernel if an instantiation exists for (G, T = per-wave k-blocks at 16 waves, M); false = nothing enqueued
template <int EPI, bool NORM = false>
static bool rp2_launch(int G, int T, const GemmArgs& a, const void* packed, int NG, int depth, int nwg) {
  const int chunks = rp2_chunks(a.M, G, T);
  const size_t lds = rp2_lds(a.M, G, T, NORM);
  if (!chunks || a.M > 16 || lds > (size_t)kRpMaxLds || a.K / 128 >= 4096 || a.g / 128 >= 4096) return false;
  switch (G) {
    case 1: return rp2_launch_g<1, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 2: return rp2_launch_g<2, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 3: return rp2_launch_g<3, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 4: return rp2_launch_g<4, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 5: return rp2_launch_g<5, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 6: return rp2_launch_g<6, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 7: return rp2_launch_g<7, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    case 8: return rp2_launch_g<8, EPI, NORM>(T, a, packed, NG, chunks, depth, nwg, lds);
    default: return false;
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


// The SiLU-mul epilogue variants of this kernel exist for 16 waves, non-temporal loads, one row tile, T = 1..4.
constexpr bool rp_fits_fused(int G, int T, int EPI) {
  if (EPI && (G & 1)) return false;
  return T >= 1 && T <= 4 && rp_fits(16, 1, G, T);
}

// returns false when no instantiation exists for (G, T, W, MT, EPI) or the LDS opt-in failed: nothing was enqueued
template <int G, int W, bool NT, int MT, int EPI>
static bool rp_launch(const GemmArgs& a, const void* packed, int NG, int per_wave, int T, int nwg, size_t lds) {
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const RpFuse fz = {(const half_t*)a.norm_h, (const half_t*)a.norm_delta, (const half_t*)a.norm_w, (half_t*)a.norm_h_out, a.norm_eps,
                      a.moe_expert_ids, a.moe_slot_scale, (long long)a.moe_expert_stride, a.moe_x_div, a.moe_num_experts};
  dim3 grid(nwg), block(W * 64);
#define RP_GO(TT)                                                                                                                     \
  if constexpr ((EPI == 0) ? (TT <= 6 && rp_fits(W, MT, G, TT) && (MT == 1 || TT == 0 || TT == 4))                                          \
                : (MT == 2) ? (W == 8 && NT && !(G & 1) && rp_fits(8, 2, G, TT))                /* SiLU-mul epilogue, 17..32 rows */            \
                            : (W == 16 ? (NT && MT == 1 && rp_fits_fused(G, TT, EPI))                                                           \
                                       : (W == 8 && NT && MT == 1 && EPI == 1 && TT == 0 && G == 4))) {  /* SiLU epilogue in rounds mode */       \
    auto kern = gemv_repacked_kernel<G, TT, W, NT, MT, EPI>;                                                                       \
    if (lds > 64 * 1024) {                       /* one workgroup per CU: opt in to more of its 160 KiB of LDS, once per device */       \
      static unsigned long long opted[2] = {0ull, 0ull};                                                                                \
      if (!opt_in_dynamic_lds((const void*)kern, kRpMaxLds, opted)) return false;                                                        \
    }                                                                                                                                   \
    hipLaunchKernelGGL(kern, grid, block, lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias, a.y, a.M, a.K, a.N, a.g, NG,   \
                       per_wave, g_rp_stamp_buffer, fz);                                                                                \
    return true;                                                                                                                        \
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
  return false;
}

template <int W, bool NT, int MT, int EPI = 0>
static bool rp_launch_g(int G, const GemmArgs& a, const void* packed, int NG, int per_wave, int T, int nwg, size_t lds) {
  switch (G) {
    case 1: return rp_launch<1, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 2: return rp_launch<2, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 3: return rp_launch<3, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 4: return rp_launch<4, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 5: return rp_launch<5, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 6: return rp_launch<6, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    case 7: return rp_launch<7, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
    default: return rp_launch<8, W, NT, MT, EPI>(a, packed, NG, per_wave, T, nwg, lds);
  }
}

}  // namespace awq
