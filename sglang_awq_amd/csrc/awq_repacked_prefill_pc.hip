// Prefill GEMM on the MFMA-fragment-major layout with the work split by ROLE inside the workgroup (gfx950).
//
// Why: counters and ablations of the single-role kernel (awq_repacked_prefill.hip; profiles/r02_prefill_ablation_counters.txt)
// show that a wave's VALU / LDS / VMEM instructions are NOT hidden behind its own MFMAs — a slot costs 16 cycles (MFMA) plus ~4
// per other instruction, 30 cycles per MFMA in that kernel — while tools/gemv_lab `overlap` shows that a wave issuing only MFMAs
// keeps the matrix pipe at full rate next to a second wave of the same SIMD issuing only VALU work.  So here
//   waves 0-3 (consumers, one per SIMD): 128 x 16 NJ accumulators each, MFMAs + the LDS reads of their A and B fragments, nothing else;
//   waves 4-7 (producers, one per SIMD): stream the packed weights, dequantise them into B fragments (the 13 VALU ops per dword of
//             rp_dequant, same numerics) and write them to a four-deep LDS ring per consumer; they also stage the x tiles
//             (global -> registers -> swizzled LDS), a quarter of a tile each.
// No workgroup barrier in the loop (a first version with one s_barrier per k-step ran 8 % slower than the single-role kernel:
// 520 cycles per step for the barrier + LDS round trip alone).  The two roles run decoupled and meet through four counters per
// pair in LDS, each written by one wave only and posted one step late, behind that wave's own s_waitcnt, so a posted value never
// runs ahead of the data it covers:
//   ready[p]  = k-steps whose fragments producer p has completely written        (consumer p reads step t + 1 when ready >= t + 2)
//   taken[c]  = k-steps whose fragments consumer c holds in registers             (producer reuses slot t & 3 when taken >= t - 3)
//   xready[p] = x tiles of which producer p has written its quarter               (consumers read tile kb + 1 when all >= kb + 2)
//   xdone[c]  = x tiles consumer c has finished reading                           (producers overwrite tile kb - 1 when all >= kb)
// A reader fetches the counter FIRST and the data right behind it (LDS executes one wave's instructions in order), keeps issuing
// MFMAs, and looks at the counter at the end of the step: if the data was not ready then (rare: the producer runs ahead), it spins
// and fetches again.  Spins are bounded (a bug shows up as a wrong result, not as a hung GPU).
// Workgroup tile 128 x 64 NJ, 512 threads, <= 256 registers per wave, 64 KiB (two x tiles) + 16 NJ KiB (ring) + 64 B of LDS.
#include <cstdlib>

#include "awq_prefill_common.h"

namespace awq {

#ifndef PC_ABL
#define PC_ABL 0            // diagnostic builds only (results wrong): 1 producers skip the dequantise, 2 producers skip the x staging, 4 consumers skip the MFMAs
#endif
constexpr int kPcThreads = 512, kPcXBytes = 2 * kPfBM * 256, kPcRing = 4;

__device__ __forceinline__ void pc_order() { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }   // no memory op moves across
__device__ __forceinline__ void pc_lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The counters are plain LDS words behind address-space-3 pointers (ds_read / ds_write, the compiler's own s_waitcnt at first use);
// pc_order() keeps every access on its side of the line.  (`volatile` would turn them into flat sc0 sc1 accesses with a full wait each.)
typedef __attribute__((address_space(3))) int lds_int;

// spin until *f >= need (wave-uniform; bounded)
__device__ __forceinline__ void pc_wait_ge(lds_int* f, int need) {
  for (int n = 0; n < (1 << 12); ++n) {
    asm volatile("" ::: "memory");
    if (__builtin_amdgcn_readfirstlane(*f) >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
}
__device__ __forceinline__ int pc_min4(lds_int* f) {        // (not yet waited for: the value is first used where the caller compares it)
  const int a = f[0], b = f[1], c = f[2], d = f[3];
  const int m0 = a < b ? a : b, m1 = c < d ? c : d;
  return m0 < m1 ? m0 : m1;
}
__device__ __forceinline__ void pc_wait4_ge(lds_int* f, int need) {
  for (int n = 0; n < (1 << 12); ++n) {
    asm volatile("" ::: "memory");
    if (__builtin_amdgcn_readfirstlane(pc_min4(f)) >= need) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int NJ>
__global__ __launch_bounds__(kPcThreads) void gemm_repacked_pc_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                      const u32x4_t* __restrict__ qw_r, const uint32_t* __restrict__ zs_r,
                                                                      const void* __restrict__ bias, void* __restrict__ y, int M, int K,
                                                                      int N, int g, int NG, int nbx, int nby, int cg_base, int ng_region) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int MI = 8, kCons = kPcRing * NJ * 1024;          // ring bytes per consumer
  unsigned char* Xs = lds;                           // [2][128 rows][128 halves], XOR-swizzled 16-byte chunks
  unsigned char* Bs = lds + kPcXBytes;               // [4 consumers][4 slots][NJ][64 lanes] x 16 B
  lds_int* F = (lds_int*)(lds + kPcXBytes + 4 * kCons);               // ready[4] taken[4] xready[4] xdone[4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PC_ROLE_BY_PARITY
  const int cw = wave >> 1;                          // probe: consumers are the even waves
  const bool is_consumer = (wave & 1) == 0;
#else
  const int cw = wave & 3;                           // consumer index; producer 4 + cw feeds consumer cw
  const bool is_consumer = wave < 4;
#endif
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g, kpg = g / 128, S = 4 * KB;

  const int nwg = nbx * nby, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int bm = (logical / nbx) * kPfBM;
  const int cg_tile = cg_base + (logical % nbx) * (4 * NJ) + cw * NJ;       // first column group of consumer cw
  const int cg_end = cg_base + ng_region < NG ? cg_base + ng_region : NG;
  unsigned char* Bc = Bs + cw * kCons + lane * 16;   // this (consumer, lane)'s place in its ring: + slot * NJ KiB + j KiB
  lds_int* f_ready = F + cw;
  lds_int* f_taken = F + 4 + cw;
  lds_int* f_xready = F + 8;
  lds_int* f_xdone = F + 12;

  if (tid < 16) F[tid] = 0;
  __syncthreads();                                   // the only workgroup barrier

  if (is_consumer) {
    // ------------------------------------------------------------------ consumer: MFMAs and fragment reads only
    float4_t acc[MI][NJ];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[mi][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t af[2][MI], bf[2][NJ];
    auto read_a = [&](u32x4_t (&dst)[MI], const unsigned char* Xb, int d, int mi) {
      dst[mi] = *(const u32x4_t*)(Xb + pfp_off(mi * 16 + r, d * 4 + q));
    };
    auto read_b = [&](u32x4_t (&dst)[NJ], int slot, int j) { dst[j] = *(const u32x4_t*)(Bc + slot * (NJ * 1024) + j * 1024); };

    pc_wait_ge(f_ready, 1);
    pc_wait4_ge(f_xready, 1);
    pc_order();
#pragma unroll
    for (int j = NJ - 1; j >= 0; --j) read_b(bf[0], 0, j);
#pragma unroll
    for (int mi = MI - 1; mi >= 0; --mi) read_a(af[0], Xs, 0, mi);
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned char* Xc = Xs + (kb & 1) * (kPfBM * 256);
      const unsigned char* Xn = Xs + ((kb + 1) & 1) * (kPfBM * 256);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int P = d & 1;                         // (4 steps per iteration: the register pairs are back in place at the back-edge)
        const int t = kb * 4 + d;
        const int slot_n = (t + 1) & (kPcRing - 1);
        int seen_ready = 0, seen_x = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            if constexpr (!(PC_ABL & 4)) mfma_tied(acc[mi][j], af[P][mi], bf[P][j]);
            else asm volatile("" :: "v"(af[P][mi]), "v"(bf[P][j]));
            const int n = j * MI + mi;               // MFMA index within the step (compile-time after unrolling)
            if (n == 0) {
              // the first MFMA waited for this step's fragments: they are all in registers (and, at d = 3, so is everything this
              // consumer reads from x tile kb)
              pc_order();
              *f_taken = t + 1;
              if (d == 3) f_xdone[cw] = kb + 1;
              pc_order();
              seen_ready = *f_ready;                 // counter first, data right behind it
              pc_order();
            }
            if (n >= 1 && n <= NJ) read_b(bf[P ^ 1], slot_n, NJ - n);
            if (n == 7 && d == 3) { pc_order(); seen_x = pc_min4(f_xready); pc_order(); }
            if (n >= 8 && n < 8 + MI) {
              const int mr = MI - 1 - (n - 8);
              if (d < 3) read_a(af[P ^ 1], Xc, d + 1, mr);
              else read_a(af[P ^ 1], Xn, 0, mr);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // what was fetched during this step is only good if the counters (fetched before it) already covered it
        if (t + 1 < S && __builtin_amdgcn_readfirstlane(seen_ready) < t + 2) {
          pc_wait_ge(f_ready, t + 2);
          pc_order();
#pragma unroll
          for (int j = NJ - 1; j >= 0; --j) read_b(bf[P ^ 1], slot_n, j);
        }
        if (d == 3 && kb + 1 < KB && __builtin_amdgcn_readfirstlane(seen_x) < kb + 2) {
          pc_wait4_ge(f_xready, kb + 2);
          pc_order();
#pragma unroll
          for (int mi = MI - 1; mi >= 0; --mi) read_a(af[P ^ 1], Xn, 0, mi);
        }
        pc_order();
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results must have left the pipe before the reads below
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = bm + mi * 16 + 4 * q + i;
        if (m < M) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = (cg_tile + j) * 16 + r;
            if (cg_tile + j < cg_end && n < N) store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, acc[mi][j][i], bias, n);
          }
        }
      }
    return;
  }

  // -------------------------------------------------------------------- producer: weights -> B fragments, x tiles -> LDS
  constexpr int AL = 8;                              // 16-byte chunks of an x tile per producer thread (2048 chunks / 256 threads)
  const int pt = cw * 64 + lane;                     // producer thread index 0 .. 255
  __amdgpu_buffer_rsrc_t rw[NJ];
  uint32_t zoff_s[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = cg_tile + j;
    const int cgj = c < cg_end ? c : cg_end - 1;     // clamped: groups outside the region are never stored
    rw[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(qw_r + (size_t)cgj * KB * 64), 0, KB * 1024, kPfRsrcFlags);
    zoff_s[j] = (uint32_t)cgj * (uint32_t)groups * 64u;
  }
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)zs_r, 0, 0x7fffffff, kPfRsrcFlags);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)bm * ldx), 0, 0x7fffffff, kPfRsrcFlags);
  const int lane16 = lane * 16, r4 = r * 4;
  uint32_t xoff[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int c = pt + 256 * i;
    const int row = c >> 4, chunk = c & 15;
    const int mr = bm + row < M ? row : M - 1 - bm;
    xoff[i] = (uint32_t)((size_t)mr * ldx + chunk * 8) * 2u;
  }
  u32x4_t a_st[AL], w_cur[NJ], w_nxt[NJ];
  uint32_t zs_nxt[NJ];
  ZsU zu[NJ];
  auto load_a = [&](int kb) {
#pragma unroll
    for (int i = 0; i < AL; ++i) a_st[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[i], kb * 256, 0);
  };
  auto store_a1 = [&](int buf, int i) {
    const int c = pt + 256 * i;
    *(u32x4_t*)(Xs + buf * (kPfBM * 256) + pfp_off(c >> 4, c & 15)) = a_st[i];
  };
  auto load_b = [&](u32x4_t (&w)[NJ], uint32_t (&zs)[NJ], int kb, int grp) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw[j], lane16, kb * 1024, 0);
      zs[j] = __builtin_amdgcn_raw_buffer_load_b32(rz, r4, zoff_s[j] + grp * 64, 0);
    }
  };

  load_a(0);
  load_b(w_cur, zs_nxt, 0, 0);
#pragma unroll
  for (int i = AL - 1; i >= 0; --i) store_a1(0, i);
  load_a(KB > 1 ? 1 : 0);                            // tile 1 travels during block 0
#pragma unroll
  for (int j = 0; j < NJ; ++j) zu[j] = zs_unpack(zs_nxt[j]);
  int grp_n = 0, cnt_n = 0;
  bool x_posted = false;                             // tile 0 is posted behind the first step's s_waitcnt
  for (int kb = 0; kb < KB; ++kb) {
    const int nxt = kb + 1 < KB ? kb + 1 : kb;
    const int nx2 = kb + 2 < KB ? kb + 2 : KB - 1;
    const int nbuf = (kb + 1) & 1;
    if (kb + 1 < KB && ++cnt_n == kpg) { cnt_n = 0; ++grp_n; }
    load_b(w_nxt, zs_nxt, nxt, grp_n);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int t = kb * 4 + d;
      // post what the previous step finished: its LDS writes are complete behind this wait (it is an old wait by now)
      pc_lds_done();
      *f_ready = t;
      if (d == 0) f_xready[cw] = kb + 1;            // tile kb: written in the prologue (kb = 0) or during steps (kb - 1, 2..3)
      pc_order();
      const int seen_taken = *f_taken;
      int seen_xd = 0;
      if (d == 2) seen_xd = pc_min4(f_xdone);
      pc_order();
      u32x4_t frag[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        frag[j] = (PC_ABL & 1) ? (u32x4_t){w_cur[j][d], w_cur[j][d], as_u32(zu[j].s2), as_u32(zu[j].z64)}
                               : rp_dequant(w_cur[j][d], zu[j].z1024, zu[j].z64, zu[j].s2);
      pc_order();
      if (t >= kPcRing && __builtin_amdgcn_readfirstlane(seen_taken) < t - (kPcRing - 1)) pc_wait_ge(f_taken, t - (kPcRing - 1));
      pc_order();
#pragma unroll
      for (int j = 0; j < NJ; ++j) *(u32x4_t*)(Bc + (t & (kPcRing - 1)) * (NJ * 1024) + j * 1024) = frag[j];
      // x tile kb + 1 -> the other buffer (which held tile kb - 1: every consumer must be done with it); youngest register first
      if (!(PC_ABL & 2) && kb + 1 < KB) {
        if (d == 2) {
          if (__builtin_amdgcn_readfirstlane(seen_xd) < kb) pc_wait4_ge(f_xdone, kb);
          pc_order();
#pragma unroll
          for (int i = AL / 2 - 1; i >= 0; --i) store_a1(nbuf, i);
        }
        if (d == 3) {
#pragma unroll
          for (int i = AL - 1; i >= AL / 2; --i) store_a1(nbuf, i);
          load_a(nx2);
        }
      }
      pc_order();
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      w_cur[j] = w_nxt[j];
      zu[j] = zs_unpack(zs_nxt[j]);
    }
  }
  (void)x_posted;
  pc_lds_done();
  *f_ready = S;
}

template <int NJ>
static int pc_launch_region(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int cg_base, int ng_region) {
  const int nbx = (ng_region + 4 * NJ - 1) / (4 * NJ), nby = (a.M + kPfBM - 1) / kPfBM;
  const size_t lds = kPcXBytes + kPcRing * 4 * NJ * 1024 + 64;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_pc_kernel<NJ>, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_repacked_pc_kernel<NJ>, dim3(nbx * nby), dim3(kPcThreads), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r,
                     a.bias, a.y, a.M, a.K, a.N, a.g, NG, nbx, nby, cg_base, ng_region);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// Column groups [0, gA) in wide tiles (128 x 256), the rest in narrow ones (128 x 192); the split is chosen by
// launch_gemm_repacked_pipelined's cost model (whole rounds of the 256 CUs in wide tiles).
int launch_gemm_repacked_pc(const GemmArgs& a, const void* packed, int gA) {
  if (!pipelined_addressable(a)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  if (gA > NG) gA = NG;
  if (gA > 0) {
    const int rc = pc_launch_region<4>(a, qw_r, zs_r, NG, 0, gA);
    if (rc) return rc;
  }
  if (gA < NG) return pc_launch_region<3>(a, qw_r, zs_r, NG, gA, NG - gA);
  return AWQ_OK;
}

}  // namespace awq
