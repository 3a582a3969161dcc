// gemv_rp3_kernel — the loop form of the straight-line decode GEMV (gemv_rp2_kernel, awq_repacked_gemv.h) for the shapes
// its unrolled form cannot hold: many k-blocks per wave (deep K: 7B down_proj from 6 rows, 70B down_proj, the 8192 x 28672
// matrix), 9..16 rows on deep matrices (staging of M x T k-blocks per wave past 8 chunks per lane), and row-mapped blocks of
// an AWQ mixture-of-experts (sorted (token, expert) pairs, 16 rows per block, awq_aux_moe_gemv_blocks).
//
// Same structure as gemv_rp2_kernel, per STAGE of TS k-blocks instead of once per wave:
//   * one workgroup = one strip of G column groups for ALL of K, 16 waves, wave w owns k-blocks [w T, w T + T) (contiguous:
//     the same summation order as every other GEMV of this layout — results are bit-identical to them);
//   * x (M x TS x 128 halves) and the (scale | 1024 + zero) words (G x TS x 16) of the NEXT stage are fetched by CHS + 1
//     per-lane-addressed loads issued at the top of the current stage and parked in wave-private LDS at its end (a wave's LDS
//     accesses complete in order: it has read the current stage's fragments by then — one buffer, no barrier);
//   * weights go through the same ring of two loads in flight per wave (three register sets: the replacement of a unit leaves
//     the moment the unit has arrived), continuing across stage boundaries; the first two issue rounds are separated by
//     workgroup barriers so the CU's queue interleaves the sixteen waves;
//   * k-blocks past the wave's range or past K contribute exactly 0 (x chunk stored as zeros, scale word forced to 0);
//   * the sixteen partial sums meet in LDS in wave order; the reduction scratch aliases the staging area behind a barrier.
// Reference semantics: AWQLinearMethod.apply, python/sglang/srt/layers/quantization/awq.py:434-451 (W rounded per element, fp32
// accumulation, one rounding of the sum, bias added after it).
#include <cstdlib>
#include <type_traits>

#include "../../include/awq_aux.h"
#include "awq_repacked_gemv.h"

namespace awq {

// ROWMAP (AWQ-MoE, 16-row blocks): grid row b serves rows row_map[16 b .. 16 b + 15] (indices of (token, expert) pairs sorted by
// expert, -1 = padding) with the weights of expert block_expert[b] (< 0: unused block, the workgroup leaves at once);
// activation row of pair p = p / x_div, output row = p, fp32 sums scaled by slot_scale[p] before the one rounding.
struct Rp3Moe {
  const int* row_map;
  const int* block_expert;
  const float* slot_scale;
  long long expert_stride;
  int x_div;
};

template <int G, int TS, int CHS, int EPI, bool M1, bool ROWMAP>
__global__ __launch_bounds__(1024) void gemv_rp3_kernel(const uint16_t* __restrict__ x, int64_t ldx, const u32x4_t* __restrict__ qw_r,
                                                        const uint32_t* __restrict__ zs_r, const void* __restrict__ bias,
                                                        void* __restrict__ y, int M, int K, int N, int groups, int gmul, int gshift,
                                                        int NG, int T, Rp3Moe moe) {
  constexpr int W = 16, L = G * TS, DD = 2, RB = 3;
  constexpr bool XJIT = !M1 && G * 4 + CHS * 4 >= 28;     // wide strips with many staging chunks: x fragments just in time (registers)
  static_assert(L >= 1 && L <= 16, "a stage is 1..16 units");
  constexpr int XS = TS * 128 + 8;                       // halves per staged x row (+8: rows 16 B apart in bank phase)
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128;
  int cg0 = blockIdx.x * G;
  if (cg0 + G > NG) cg0 = NG - G;                        // last strip overlaps its neighbour (same values written twice)
  if constexpr (M1) M = 1;
  const int* rows = nullptr;
  if constexpr (ROWMAP) {
    const int e = moe.block_expert[blockIdx.y];
    if (e < 0) return;                                   // unused block (uniform over the workgroup)
    qw_r = (const u32x4_t*)((const unsigned char*)qw_r + (long long)e * moe.expert_stride);
    zs_r = (const uint32_t*)((const unsigned char*)zs_r + (long long)e * moe.expert_stride);
    rows = moe.row_map + (size_t)blockIdx.y * 16;
    M = 16;
  }
  const int kb0 = wave * T;
  int kb_end = kb0 + T;
  kb_end = kb_end < KB ? kb_end : KB;                    // this wave's k-blocks: [kb0, kb_end), possibly empty
  const int S = (T + TS - 1) / TS;                       // stages (the same for every wave: the barriers below are uniform)

  const int xbytes = M * XS * 2;
  unsigned char* const stg = (unsigned char*)red + (size_t)wave * (xbytes + G * TS * 64 + 16);
  const int dump = xbytes + G * TS * 64;                 // 16-byte slot for lanes without a chunk

  // staging: CHS loads of x chunks (row, 8-half column chunk of the stage), one load of the (<= 64) zs chunks
  constexpr int NZ = G * TS * 4;
  static_assert(NZ <= 64, "zs words of a stage fit one load");
  const int nx = M * TS * 16;
  // Register diet (16 waves x 128 registers): per-lane chunk coordinates are recomputed from the lane id where they are used (behind an
  // opaque copy, or hipcc hoists them out of the stage loop and keeps 2-3 registers per chunk live across it); row offsets are 32-bit.
  u32x4_t sv[CHS + 1];
  uint32_t xoff[CHS];                                    // element offset of the chunk's source row (rows * ldx < 2^31: checked on the host)
#pragma unroll
  for (int i = 0; i < CHS; ++i) {
    const int id = lane + 64 * i;
    const int row = id / (TS * 16);
    int src_row = row < M ? row : 0;
    if constexpr (ROWMAP) {
      const int p = rows[row < 16 ? row : 0];
      src_row = p >= 0 ? p / moe.x_div : 0;              // padding rows read row 0 (finite or not: their outputs are never stored)
    }
    xoff[i] = (uint32_t)src_row * (uint32_t)ldx;
  }
  auto x_cc = [&](int l, int i) { const int id = l + 64 * i; return id - (id / (TS * 16)) * (TS * 16); };          // column chunk inside the stage
  auto x_dst = [&](int l, int i) { const int id = l + 64 * i, row = id / (TS * 16); return id < nx ? (row * XS + (id - row * (TS * 16)) * 8) * 2 : dump; };
  const int zc = lane < NZ ? lane : NZ - 1;              // zs chunk of this lane: (c, t, part)
  const int z_c = zc / (TS * 4), z_rem = zc - z_c * (TS * 4), z_t = z_rem >> 2, z_part = z_rem & 3;

  auto stage_load = [&](int s) {                         // issue the staging loads of stage s (clamped addresses: always in bounds)
    const int kbs = kb0 + s * TS;
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const int cc = x_cc(l, i);
      int kbx = kbs + (cc >> 4);
      kbx = kbx < KB ? kbx : KB - 1;
      sv[i] = *(const u32x4_t*)(x + xoff[i] + (uint32_t)kbx * 128u + (uint32_t)(cc & 15) * 8u);
    }
    int kbz = kbs + z_t;
    kbz = kbz < KB ? kbz : KB - 1;
    sv[CHS] = *(const u32x4_t*)(zs_r + ((size_t)(cg0 + z_c) * groups + ((kbz * gmul) >> gshift)) * 16 + z_part * 4);
  };
  auto stage_store = [&](int s) {                        // park them in this wave's LDS; k-blocks outside [kb0, kb_end) become exact zeros
    const int kbs = kb0 + s * TS;
    int l = lane;
    asm volatile("" : "+v"(l));
#pragma unroll
    for (int i = 0; i < CHS; ++i) {
      const bool in = kbs + (x_cc(l, i) >> 4) < kb_end;
      *(u32x4_t*)(stg + x_dst(l, i)) = in ? sv[i] : (u32x4_t){0u, 0u, 0u, 0u};
    }
    const bool zin = kbs + z_t < kb_end;
    const int zdst = l < NZ ? xbytes + ((z_c * TS + z_t) * 16 + z_part * 4) * 4 : dump;
    *(u32x4_t*)(stg + zdst) = zin ? sv[CHS] : (u32x4_t){0x64000000u, 0x64000000u, 0x64000000u, 0x64000000u};   // scale 0, zero point 0
  };

  // weights: unit u of stage s = (c = u / TS, t = u % TS): a wave's consecutive loads of a column group are contiguous.
  // load_rel(s, j, slot): unit j counted from unit 0 of stage s (j >= L runs into the following stages).
  const uint32_t loff = (uint32_t)lane * 16u;
  u32x4_t wbuf[RB];
  auto load_rel = [&](int s, int j, int slot) {
    const int ds = j / L, u = j - ds * L;
    const int c = u / TS, t = u - c * TS;
    int kb = kb0 + (s + ds) * TS + t;
    kb = kb < KB ? kb : KB - 1;                          // clamped: re-read, then multiplied by x = 0 and scale 0 (or never used)
    const unsigned char* p = (const unsigned char*)(qw_r + ((size_t)(cg0 + c) * KB + kb) * 64) + loff;
    wbuf[slot] = __builtin_nontemporal_load((const u32x4_t*)p);
  };

  stage_load(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    load_rel(0, i, i);
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
  }
  stage_store(0);
  __builtin_amdgcn_sched_barrier(0);

#if !RP_DYNPRIO
  if (wave >= 12) __builtin_amdgcn_s_setprio(2);
  else if (wave >= 8) __builtin_amdgcn_s_setprio(1);
#endif
  uint32_t mlo = kLoNib, mhi = kHiNib, magic = kMagicF16;
  asm volatile("" : "+s"(mlo), "+s"(mhi));               // opaque: (w & m) | magic then selects v_and_or_b32
  asm volatile("" : "+v"(magic));
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const half_t* x_lds = (const half_t*)stg + (M1 ? 0 : (size_t)(r < M ? r : M - 1) * XS);
  const uint32_t* zs_lds = (const uint32_t*)(stg + xbytes) + r;
  float4_t acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c) acc[c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  u32x4_t xa[4];

  // One stage.  PH = ring slot of the stage's unit 0: the ring advances by L units per stage, so with L % RB != 0 the slots
  // rotate from stage to stage; the registers of loads in flight cannot be moved (a move would wait for them), so the body
  // exists once per phase and the stage loop picks it (wave-uniform switch).
  auto stage_body = [&](auto ph, int s) {
    constexpr int PH = decltype(ph)::value;
    const bool more = s + 1 < S;                         // uniform
    prio_by_progress(s, S);                              // (stages done of stages: the waves of a SIMD finish together)
    if (more) stage_load(s + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < L; ++i) {
      const int c = i / TS, t = i - c * TS;
      if constexpr (!XJIT) {
        if (TS > 1 || i == 0) {
#pragma unroll
          for (int d = 0; d < 4; ++d) xa[d] = *(const u32x4_t*)(x_lds + t * 128 + d * 32 + q * 8);
        }
      }
      const half2_t zh = as_h2(zs_lds[(c * TS + t) * 16]);
      const half2_t s2 = __builtin_shufflevector(zh, zh, 0, 0);
      const half2_t z1024 = __builtin_shufflevector(zh, zh, 1, 1);
      const half2_t z64 = z1024 - c960;                  // exact: (1024 + z) - 960
      u32x4_t w = wbuf[(PH + i) % RB];
      asm volatile("" : "+v"(w));                        // unit i has arrived (the wait sits here) ...
      __builtin_amdgcn_sched_barrier(0);
      if (s + (i + DD) / L < S) load_rel(s, i + DD, (PH + i + DD) % RB);   // ... its replacement leaves at once, into the spare register set
      __builtin_amdgcn_sched_barrier(0);
      // XJIT (many rows on wide strips): the x fragment of a k-step is read from LDS just before its MFMA, one k-step ahead, instead of
      // holding the four of a k-block (16 registers) across the stage's units
      u32x4_t xn;
      if constexpr (XJIT) xn = *(const u32x4_t*)(x_lds + t * 128 + q * 8);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        u32x4_t xc;
        if constexpr (XJIT) {
          xc = xn;
          if (d < 3) xn = *(const u32x4_t*)(x_lds + t * 128 + (d + 1) * 32 + q * 8);
        } else {
          xc = xa[d];
        }
        const uint32_t ww = w[d], w8 = ww >> 8;
        const half2_t d0 = as_h2((ww & mlo) | magic) - z1024;
        const half2_t d1 = __builtin_elementwise_fma(as_h2((ww & mhi) | magic), sixteenth, -z64);
        const half2_t d2 = as_h2((w8 & mlo) | magic) - z1024;
        const half2_t d3 = __builtin_elementwise_fma(as_h2((w8 & mhi) | magic), sixteenth, -z64);
        const u32x4_t frag = {as_u32(d0 * s2), as_u32(d1 * s2), as_u32(d2 * s2), as_u32(d3 * s2)};
        acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, xc), __builtin_bit_cast(half8_t, frag), acc[c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) stage_store(s + 1);                        // behind this stage's last fragment reads (a wave's LDS accesses stay in order)
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (L % RB == 0) {
#pragma unroll 1
    for (int s = 0; s < S; ++s) stage_body(std::integral_constant<int, 0>{}, s);
  } else {
    int phase = 0;
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
      if (phase == 0) stage_body(std::integral_constant<int, 0>{}, s);
      else if (phase == 1) stage_body(std::integral_constant<int, 1>{}, s);
      else stage_body(std::integral_constant<int, 2>{}, s);
      phase = (phase + L) % RB;
    }
  }
  __builtin_amdgcn_s_setprio(0);

  // D[m = 4q + i][n = r] per column group -> LDS (aliasing the staging area: every wave is done with it behind this barrier),
  // summed over the waves in fixed order
  __syncthreads();
  const int SW = 16 * G;
  if constexpr (M1) {
    if (q == 0) {
#pragma unroll
      for (int c = 0; c < G; ++c) red[wave * SW + c * 16 + r] = acc[c][0];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * q + i;
#pragma unroll
      for (int c = 0; c < G; ++c)
        if (m < M) red[((size_t)wave * M + m) * SW + c * 16 + r] = acc[c][i];
    }
  }
  __syncthreads();
  if constexpr (EPI == 1) {
    // strip = G / 2 (gate, up) pairs of column groups; output column = 16 * (pair index) + r of act[M, N / 2]
    const int SH = 8 * G, I = N / 2;
    for (int idx = threadIdx.x; idx < M * SH; idx += W * 64) {
      const int m = M1 ? 0 : idx / SH, c = idx - m * SH;
      const int pair = c >> 4, r16 = c & 15;
      const int n = (cg0 / 2 + pair) * 16 + r16;
      if (n >= I) continue;
      int orow = m;
      if constexpr (ROWMAP) { orow = rows[m]; if (orow < 0) continue; }
      const int cgate = pair * 32 + r16;
      float gv = red[(size_t)m * SW + cgate], uv = red[(size_t)m * SW + cgate + 16];
#pragma unroll
      for (int w = 1; w < W; ++w) {
        gv += red[((size_t)w * M + m) * SW + cgate];
        uv += red[((size_t)w * M + m) * SW + cgate + 16];
      }
      if constexpr (ROWMAP) { if (moe.slot_scale != nullptr) { const float sc = moe.slot_scale[orow]; gv *= sc; uv *= sc; } }
      const float xg = (float)(half_t)gv;                                // the unfused path rounds gate_up to fp16 first
      ((half_t*)y)[(size_t)orow * I + n] = (half_t)(xg / (1.f + __expf(-xg))) * (half_t)uv;
    }
  } else {
    for (int idx = threadIdx.x; idx < M * SW; idx += W * 64) {
      const int m = M1 ? 0 : idx / SW, col = idx - m * SW;
      const int n = cg0 * 16 + col;
      if (n >= N) continue;
      int orow = m;
      if constexpr (ROWMAP) { orow = rows[m]; if (orow < 0) continue; }
      float v = red[(size_t)m * SW + col];
#pragma unroll
      for (int w = 1; w < W; ++w) v += red[((size_t)w * M + m) * SW + col];
      if constexpr (ROWMAP) { if (moe.slot_scale != nullptr) v *= moe.slot_scale[orow]; }
      store_output<AWQ_DTYPE_F16>(y, (size_t)orow * N + n, v, bias, n);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------- host
static size_t rp3_lds(int M, int G, int TS) {
  const size_t stage = (size_t)16 * ((size_t)M * (TS * 128 + 8) * 2 + (size_t)G * TS * 64 + 16);
  const size_t redb = (size_t)16 * M * 16 * G * sizeof(float);
  return stage > redb ? stage : redb;
}

// Instantiations: only what the dispatcher takes (awq_dispatch.h) — 9..16 rows (four staging chunks per lane, one k-block per stage) on
// strips of 1 or 2 column groups for the plain operator; 16-row blocks on strips of 6 / 3 / 2 / 1 column groups for the AWQ-MoE.  The
// template itself is general (stages of 1, 2 or 4 k-blocks, 1..16 rows, strips up to 8 groups: the round-3 A/B built all of them,
// profiles/r03_kbench_rp3_ab.txt); widths whose stage is not a whole number of ring turns need a loop body per ring phase and, beyond
// 2 groups at 16 rows, spill.
template <int G, int EPI, bool ROWMAP>
static bool rp3_go(const GemmArgs& a, const void* packed, int NG, int nwg, int ny, const Rp3Moe& moe) {
  if constexpr (EPI == 1 && (G & 1)) {
    return false;
  } else {
    constexpr int TS = 1, CHS = 4;
    const int M = ROWMAP ? 16 : a.M;
    const size_t lds = rp3_lds(M, G, TS);
    if (lds > (size_t)kRpMaxLds || M < 9 || M > 16) return false;
    const u32x4_t* qw_r = (const u32x4_t*)packed;
    const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
    const int KB = a.K / 128, T = (KB + 15) / 16;
    const int gk = a.g / 128;
    int lg = 0;
    while ((1 << lg) < gk) ++lg;
    const int gshift = 12 + lg, gmul = (int)(((1ll << gshift) + gk - 1) / gk);
    auto kern = gemv_rp3_kernel<G, TS, CHS, EPI, false, ROWMAP>;
    static unsigned long long opted[2] = {0ull, 0ull};
    if (lds > 64 * 1024 && !opt_in_dynamic_lds((const void*)kern, kRpMaxLds, opted)) return false;
    hipLaunchKernelGGL(kern, dim3(nwg, ny), dim3(1024), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r, a.bias, a.y, M, a.K, a.N,
                       a.K / a.g, gmul, gshift, NG, T, moe);
    return true;
  }
}

// The plain operator on the loop form: M <= 16, fp16, g % 128 == 0, at least 16 k-blocks; AWQ_ERR_BAD_VARIANT when the shape has no
// instantiation (the caller keeps its previous kernel).
int launch_gemv_repacked_loop(const GemmArgs& a, const void* packed) {
  if (!repacked_fast(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.M > 16 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N), KB = a.K / 128;
  if (KB < 16 || KB >= 4096 || a.g / 128 >= 4096) return AWQ_ERR_BAD_VARIANT;
  int G = 0, nwg = 0;
  if (!gemv_strip_geometry(a.K, a.N, &G, &nwg)) return AWQ_ERR_BAD_VARIANT;
  if (a.M < 9 || G > 2) return AWQ_ERR_BAD_VARIANT;      // what is instantiated (the dispatcher asks for 13..16 rows on <= 2 groups)
  if ((int64_t)a.M * a.ldx >= (int64_t(1) << 31)) return AWQ_ERR_BAD_VARIANT;                    // 32-bit activation row offsets in the kernel
  const Rp3Moe none = {nullptr, nullptr, nullptr, 0, 1};
  const bool ok = G == 1 ? rp3_go<1, 0, false>(a, packed, NG, nwg, 1, none) : rp3_go<2, 0, false>(a, packed, NG, nwg, 1, none);
  if (!ok) return AWQ_ERR_BAD_VARIANT;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq

// ------------------------------------------------------------------------------------------------------------- block alignment
// One workgroup: count the pairs of every expert (LDS atomics), pad each count to a multiple of the block size (16 rows for the GEMV
// route, 128 for the MFMA tile route), place every valid pair in its expert's run.  The order of the rows inside an expert's blocks
// is whatever the LDS atomics give — rows are independent, the outputs do not depend on it.
namespace awq {
constexpr int kAlignMaxExperts = 1024;
__global__ __launch_bounds__(1024) void moe_align_kernel(const int* __restrict__ ids, int P, int E, int* __restrict__ row_map,
                                                         int* __restrict__ block_expert, int B, int shift) {
  __shared__ int cnt[kAlignMaxExperts], start[kAlignMaxExperts + 1], cur[kAlignMaxExperts];
  const int t = threadIdx.x;
  const int pad = (1 << shift) - 1;
  for (int e = t; e < E; e += 1024) { cnt[e] = 0; cur[e] = 0; }
  for (int i = t; i < (B << shift); i += 1024) row_map[i] = -1;
  for (int b = t; b < B; b += 1024) block_expert[b] = -1;
  __syncthreads();
  for (int p = t; p < P; p += 1024) {
    const int e = ids[p];
    if (e >= 0 && e < E) atomicAdd(&cnt[e], 1);
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int e = 0; e < E; ++e) { start[e] = acc; acc += (cnt[e] + pad) & ~pad; }
    start[E] = acc;
  }
  __syncthreads();
  for (int e = t; e < E; e += 1024)
    for (int b = start[e] >> shift; b < (start[e + 1] >> shift); ++b) block_expert[b] = e;
  for (int p = t; p < P; p += 1024) {
    const int e = ids[p];
    if (e >= 0 && e < E) row_map[start[e] + atomicAdd(&cur[e], 1)] = p;
  }
}
}  // namespace awq

extern "C" int awq_aux_moe_align_blocks_n(const int32_t* ids, int64_t pairs, int64_t num_experts, int block_rows, int32_t* row_map,
                                          int32_t* block_expert, int64_t num_blocks, void* stream) {
  if (!ids || !row_map || !block_expert) return AWQ_ERR_NULL_POINTER;
  if (block_rows != 16 && block_rows != 64 && block_rows != 128) return AWQ_ERR_BAD_SHAPE;
  if (pairs <= 0 || num_experts < 1 || num_blocks < (pairs + block_rows - 1) / block_rows + num_experts || num_blocks > (1 << 24) || pairs > (1 << 28) ||
      num_blocks * block_rows >= (int64_t(1) << 31))
    return AWQ_ERR_BAD_SHAPE;
  if (num_experts > awq::kAlignMaxExperts) return AWQ_ERR_BAD_VARIANT;
  hipLaunchKernelGGL(awq::moe_align_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, (int)pairs, (int)num_experts, row_map,
                     block_expert, (int)num_blocks, block_rows == 16 ? 4 : block_rows == 64 ? 6 : 7);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

extern "C" int awq_aux_moe_align_blocks(const int32_t* ids, int64_t pairs, int64_t num_experts, int32_t* row_map, int32_t* block_expert,
                                        int64_t num_blocks, void* stream) {
  return awq_aux_moe_align_blocks_n(ids, pairs, num_experts, 16, row_map, block_expert, num_blocks, stream);
}

// AWQ-MoE over expert-sorted pairs in blocks of 128 rows on the MFMA tile kernel (include/awq_aux.h; awq_repacked_prefill.hip)
extern "C" int awq_aux_moe_gemm_blocks(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                                       const int32_t* row_map, const int32_t* block_expert, int64_t num_blocks, int block_rows, const float* slot_scale,
                                       void* y, int64_t K, int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream) {
  using namespace awq;
  if (!x || !packed_experts || !row_map || !block_expert || !y) return AWQ_ERR_NULL_POINTER;
  if (K <= 0 || N <= 0 || group_size <= 0 || N % 8 || K % group_size || num_blocks <= 0 || num_blocks > (1 << 20) || ldx < K || x_div < 1 ||
      (block_rows != 64 && block_rows != 128))
    return AWQ_ERR_BAD_SHAPE;
  if ((((uintptr_t)packed_experts) & 15) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 1) || (expert_stride_bytes & 15) || ldx % 8)
    return AWQ_ERR_MISALIGNED;
  if (!repacked_fast(K, N, group_size, dtype) || (silu_mul && N % 32)) return AWQ_ERR_BAD_VARIANT;
  if ((num_blocks * block_rows / x_div + 1) * ldx >= (int64_t(1) << 31)) return AWQ_ERR_BAD_VARIANT;     // 32-bit activation row offsets in the kernel
  const int64_t tiles = num_blocks * ((rp_groups(N) + 15) / 16);
  if (tiles >= (int64_t(1) << 31)) return AWQ_ERR_BAD_VARIANT;
  GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = nullptr; a.scales = nullptr; a.qzeros = nullptr; a.bias = nullptr; a.y = y;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.M = (int)(num_blocks * block_rows); a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = 0;
  a.stream = (hipStream_t)stream;
  return launch_gemm_repacked_moe_tiles(a, packed_experts, row_map, block_expert, (int)num_blocks, block_rows, slot_scale,
                                        (long long)expert_stride_bytes, x_div, silu_mul != 0);
}

// AWQ-MoE over expert-sorted (token, expert) pairs in blocks of 16 rows (include/awq_aux.h).
extern "C" int awq_aux_moe_gemv_blocks(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                                       const int32_t* row_map, const int32_t* block_expert, int64_t num_blocks, const float* slot_scale,
                                       void* y, int64_t K, int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream) {
  using namespace awq;
  if (!x || !packed_experts || !row_map || !block_expert || !y) return AWQ_ERR_NULL_POINTER;
  if (K <= 0 || N <= 0 || group_size <= 0 || N % 8 || K % group_size || num_blocks <= 0 || num_blocks > 65535 || ldx < K || x_div < 1)
    return AWQ_ERR_BAD_SHAPE;
  if ((((uintptr_t)packed_experts) & 15) || (((uintptr_t)x) & 15) || (((uintptr_t)y) & 1) || (expert_stride_bytes & 15) || ldx % 8)
    return AWQ_ERR_MISALIGNED;
  if (!repacked_fast(K, N, group_size, dtype) || K / 128 >= 4096 || group_size / 128 >= 4096 || (silu_mul && N % 32)) return AWQ_ERR_BAD_VARIANT;
  GemmArgs a;
  a.x = x; a.ldx = ldx; a.qweight = nullptr; a.scales = nullptr; a.qzeros = nullptr; a.bias = nullptr; a.y = y;
  a.workspace = nullptr; a.workspace_bytes = 0;
  a.M = 16; a.K = (int)K; a.N = (int)N; a.g = (int)group_size; a.dtype = dtype; a.tune = 0;
  a.stream = (hipStream_t)stream;
  const int NG = rp_groups(N);
  // Strip width.  A launch holds several 16-row blocks (one per active expert at least), so strips need not be one-per-CU narrow; every
  // workgroup stages all 16 activation rows, so wider strips mean less activation traffic.  6 column groups: a stage of 6 units is a
  // whole number of ring turns (one loop body, 98-100 registers; widths that are not a multiple of 3 need a body per ring phase and
  // spill at 16 rows: tools/rp_resources.py), even for the SiLU-mul pairs.
  int G = NG >= 6 ? 6 : (NG >= 3 && !silu_mul) ? 3 : NG >= 2 ? 2 : 1;
  if (silu_mul && (G & 1)) return AWQ_ERR_BAD_VARIANT;
  if ((num_blocks * 16 / x_div + 1) * ldx >= (int64_t(1) << 31)) return AWQ_ERR_BAD_VARIANT;     // 32-bit activation row offsets in the kernel
  const int nwg = (NG + G - 1) / G;
  const Rp3Moe moe = {row_map, block_expert, slot_scale, (long long)expert_stride_bytes, x_div};
  const int nb = (int)num_blocks;
  bool ok = false;
  if (silu_mul) ok = G == 6 ? rp3_go<6, 1, true>(a, packed_experts, NG, nwg, nb, moe) : rp3_go<2, 1, true>(a, packed_experts, NG, nwg, nb, moe);
  else ok = G == 6 ? rp3_go<6, 0, true>(a, packed_experts, NG, nwg, nb, moe) : G == 3 ? rp3_go<3, 0, true>(a, packed_experts, NG, nwg, nb, moe)
          : G == 2 ? rp3_go<2, 0, true>(a, packed_experts, NG, nwg, nb, moe) : rp3_go<1, 0, true>(a, packed_experts, NG, nwg, nb, moe);
  if (!ok) return AWQ_ERR_BAD_VARIANT;
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}
