// Prefill GEMM on the MFMA-fragment-major layout, software-pipelined by hand (gfx950).
//
// Same decomposition as gemm_repacked_tiled_kernel<1, 4> (awq_repacked.hip): workgroup tile 128 x 256, 4 waves of
// 128 x 64 = 8 x 4 MFMA 16x16x32 tiles, B never touches LDS (every wave dequantises the dwords of its own 64
// columns into fragments and reuses each for 8 row tiles), x through double-buffered XOR-swizzled LDS.  What
// changes is the instruction stream, after rocprofv3 counters on the compiler-scheduled kernel showed the matrix
// pipe busy 34 % of the time with one wave per SIMD alternating between VALU-only and MFMA-only runs:
//   * accumulators are tied in place in the AGPR file (inline-asm MFMA, "+a"): no v_accvgpr shuffling;
//   * the 13 VALU ops that turn the NEXT dword into a B fragment are cut into 8 stages of <= 2 ops, one stage
//     issued behind each of the 8 MFMAs that consume the CURRENT fragment — an MFMA 16x16x32 occupies the matrix
//     pipe for 16 cycles but the issue port for 8, which leaves room for exactly two VALU ops
//     (MI355X_MICROARCH.md, issue-cost row); `sched_barrier` pins every (MFMA, stage) pair;
//   * LDS reads of the next k-step's x fragments, the LDS writes of the next x tile and the unpacking of the next
//     k-block's scales / zeros ride in the same slots; the one barrier per k-block sits inside its last k-step so the
//     next block's first fragments are fetched behind it in the MFMA shadow.
// Numerics are those of rp_dequant (same operations in the same order per element).
#include <cstdlib>
#include <type_traits>

#include "awq_prefill_common.h"

namespace awq {

struct DqPipe {            // one dword on its way to becoming a fragment
  uint32_t w, w8, t0, t1, t2, t3;
  half2_t d0, d1, d2, d3;
  u32x4_t f;
};

template <int S>
__device__ __forceinline__ void dq_stage(DqPipe& p, half2_t z1024, half2_t z64, half2_t s2) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t magic = kMagicF16;
  if constexpr (S == 0) { p.w8 = p.w >> 8; p.t0 = and_or(p.w, kLoNib, magic); }
  if constexpr (S == 1) { p.t1 = and_or(p.w, kHiNib, magic); p.t2 = and_or(p.w8, kLoNib, magic); }
  if constexpr (S == 2) { p.t3 = and_or(p.w8, kHiNib, magic); p.d0 = as_h2(p.t0) - z1024; }
  if constexpr (S == 3) { p.d1 = __builtin_elementwise_fma(as_h2(p.t1), sixteenth, -z64); p.d2 = as_h2(p.t2) - z1024; }
  if constexpr (S == 4) { p.d3 = __builtin_elementwise_fma(as_h2(p.t3), sixteenth, -z64); p.f[0] = as_u32(p.d0 * s2); }
  if constexpr (S == 5) { p.f[1] = as_u32(p.d1 * s2); p.f[2] = as_u32(p.d2 * s2); }
  if constexpr (S == 6) { p.f[3] = as_u32(p.d3 * s2); }
}

#ifndef PF_ABL
#define PF_ABL 0            // diagnostic builds only (timing ablations; results are wrong): 1 no dequant stages, 2 no LDS fragment reads in the loop, 4 no x-tile loads / LDS writes, 8 no barrier
#endif
constexpr int kPfThreads = 256;
#ifndef PF_ROW_BLOCK
#define PF_ROW_BLOCK 4      // row tiles per super-row of the 2-D blocked tile order (see the kernel); 0: A/B build, column tiles of a row tile consecutive
#endif
#ifndef PF_LDS_EPILOGUE
#define PF_LDS_EPILOGUE 1   // 0: A/B build, the output tile is stored straight from the accumulators (round 1-2 form)
#endif

// AWQ-MoE prefill (MOE = true; awq_aux_moe_gemm_blocks): the rows are (token, expert) pairs grouped by expert into 128-row blocks that
// never straddle two experts (awq_aux_moe_align_blocks_n, the reference's moe_align_block_size step); a row tile takes its weights from
// expert block_expert[tile] (expert_stride bytes apart; -1 = unused tile: the workgroup leaves), row r of the tile reads activation
// row row_map[r] / x_div and writes output row row_map[r] (-1 = padding: read row 0, write nothing), with the fp32 sums multiplied by
// slot_scale[row_map[r]] before the one rounding (the routed weight, where the reference's fused MoE applies it).  EPI = 1: the
// column groups alternate gate / up (w13 repacked from interleaved tensors) and act = silu(fp16 gate) * fp16 up is written, N / 2 wide.
struct PfMoe {
  const int* row_map;
  const int* block_expert;
  const float* slot_scale;
  long long expert_stride;
  int x_div;
};

// Split-K over workgroups for under-filled launches (few tiles, 33 .. ~512 rows): S slices of kb_per k-blocks each; a workgroup writes its
// fp32 partial tile to part[slice][M][N] with plain stores and pf_splitk_reduce_kernel (a second launch: the kernel boundary orders them,
// fixed slice order, deterministic) adds the slices, the bias, and rounds once.  part == nullptr: one slice, the epilogue rounds and stores.
struct PfSplit {
  float* part;
  int S;
  int kb_per;
};

// NJ = column groups (16 columns) per wave: 4 -> 128 x 256 tiles, 3 -> 128 x 192 tiles, 2 -> 128 x 128 tiles.
// A launch covers the column groups [cg_base, cg_base + ng_region) of the matrix; the host cuts N into a region of 256-wide
// tiles that fills whole rounds of the 256 CUs and a remainder of 192-wide tiles, so that 2048 x 11008 costs 2 + 0.75 rounds
// instead of the 3 that 688 equal tiles pay for 2.69 (launch_gemm_repacked_pipelined).
template <int NJ, bool MOE = false, int EPI = 0, int MI = 8, bool SPLIT = false>
__global__ __launch_bounds__(kPfThreads, 1) void gemm_repacked_pipelined_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                                const u32x4_t* __restrict__ qw_r,
                                                                                const uint32_t* __restrict__ zs_r,
                                                                                const void* __restrict__ bias, void* __restrict__ y, int M,
                                                                                int K, int N, int g, int NG, int nbx, int nby, int cg_base,
                                                                                int ng_region, PfMoe moe, PfSplit split) {
  static_assert(!EPI || (MOE && NJ % 2 == 0), "SiLU-mul epilogue: whole (gate, up) pairs per wave");
  static_assert(MI == 8 || MI == 4, "128- or 64-row tiles");
  extern __shared__ __attribute__((aligned(16))) unsigned char As[];      // 2 x 32 KiB
  // MI = row tiles (16 rows) per wave: 8 -> 128-row workgroup tiles; 4 -> 64-row tiles (MoE layers whose experts hold ~64 rows each:
  // half the MFMA work per streamed weight, the 7 dequantise stages two to an MFMA slot instead of one)
  constexpr int BM = MI * 16, AL = MI;              // rows per workgroup tile; x-tile chunks (16 B) per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform on purpose: every weight / scale address below is
  const int q = lane >> 4, r = lane & 15;                          // an SGPR base + a 32-bit lane offset (no 64-bit VALU address math
  const int KBT = K / 128, groups = K / g, kpg = g / 128;          // in the loop: it is not hidden behind the MFMAs)

  // (SPLIT is a template parameter so that the unsplit instantiations compile exactly as before: the slice arithmetic, though a handful of
  // scalar instructions, cost the dense prefill 1-2 %)
  const int ntile = nbx * nby, nwg = SPLIT ? ntile * split.S : ntile, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical_all = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int slice = SPLIT ? logical_all / ntile : 0, logical = SPLIT ? logical_all - slice * ntile : logical_all;
  const int k0 = SPLIT ? slice * split.kb_per : 0;                  // this workgroup's k-blocks: [k0, k0 + KB)
  const int KB = SPLIT ? (KBT - k0 < split.kb_per ? KBT - k0 : split.kb_per) : KBT;
  // Tile order.  The workgroups an XCD runs together are 32 consecutive logical tiles.  As 32 column tiles of ONE row tile they share that row
  // tile's x (1 MB at K = 4096) in the XCD's 4 MB L2 and stream 32 different weight tiles (16 MB) past it; ordered row-fastest inside
  // super-rows of PF_ROW_BLOCK = 4 row tiles they cover 4 x 8 tiles: 4 MB of x and 4 MB of weights, each shared.  2048 x 11008 x 4096:
  // 185.0 -> 183.5 us, M = 4096 360 -> 352, M = 8192 690 -> 666 (1.11 PFLOP/s); never slower (profiles/r03_kbench_prefill_tile_order_ab.txt).
  // (The MoE and split-K launches keep the plain order: their row tiles belong to different experts / they have few tiles.)
  int bm, cg_tile;
  if constexpr (!MOE && !SPLIT && PF_ROW_BLOCK > 1) {
    const int sb_tiles = PF_ROW_BLOCK * nbx, sb = logical / sb_tiles, in_sb = logical - sb * sb_tiles;
    const int rows_sb = nby - sb * PF_ROW_BLOCK < PF_ROW_BLOCK ? nby - sb * PF_ROW_BLOCK : PF_ROW_BLOCK;
    bm = (sb * PF_ROW_BLOCK + in_sb % rows_sb) * BM;
    cg_tile = cg_base + (in_sb / rows_sb) * (4 * NJ) + wn * NJ;
  } else {
    bm = (logical / nbx) * BM;
    cg_tile = cg_base + (logical % nbx) * (4 * NJ) + wn * NJ;       // this wave's first column group
  }
  const int cg_end = cg_base + ng_region < NG ? cg_base + ng_region : NG;
  if constexpr (MOE) {
    const int eid = moe.block_expert[bm / BM];                  // uniform over the workgroup, read before any barrier
    if (eid < 0) return;
    qw_r = (const u32x4_t*)((const unsigned char*)qw_r + (long long)eid * moe.expert_stride);
    zs_r = (const uint32_t*)((const unsigned char*)zs_r + (long long)eid * moe.expert_stride);
  }

  // Every global load of the loop is a buffer load: SGPR descriptor + 32-bit lane offset + scalar offset (k-block, group), so the
  // k-block's addresses cost a few SALU ops instead of ~60 64-bit VALU ops per k-block in front of the first MFMA.
  constexpr int kRsrcFlags = kPfRsrcFlags;
  __amdgpu_buffer_rsrc_t rw[NJ];                       // this wave's column groups: [KB][64 lanes] dwordx4
  uint32_t zoff_s[NJ];                                 // byte offset of the group's first scale word in zs_r
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = cg_tile + j;
    const int cgj = c < cg_end ? c : cg_end - 1;       // clamped: groups outside the region are never stored
    rw[j] = __builtin_amdgcn_make_buffer_rsrc((void*)(qw_r + ((size_t)cgj * KBT + k0) * 64), 0, KB * 1024, kRsrcFlags);
    zoff_s[j] = (uint32_t)cgj * (uint32_t)groups * 64u;
  }
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)zs_r, 0, 0x7fffffff, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (MOE ? (size_t)0 : (size_t)bm * ldx) + (size_t)k0 * 128), 0, 0x7fffffff, kRsrcFlags);
  const int lane16 = lane * 16, r4 = r * 4;

  u32x4_t a_st[AL];
  u32x4_t w_cur[NJ], w_nxt[NJ];
  uint32_t zs_nxt[NJ];
  ZsU zu[NJ], zu_nxt[NJ];

  // this thread's AL rows / chunks of the x tile: byte offsets from the tile's first row
  uint32_t xoff[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int c = tid + kPfThreads * i;
    const int row = c >> 4, chunk = c & 15;
    int mr = bm + row < M ? row : M - 1 - bm;
    if constexpr (MOE) {
      const int pr = moe.row_map[bm + row];
      mr = pr >= 0 ? pr / moe.x_div : 0;
    }
    xoff[i] = (uint32_t)((size_t)mr * ldx + chunk * 8) * 2u;
  }
  auto load_a = [&](int kb) {
#pragma unroll
    for (int i = 0; i < AL; ++i) a_st[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[i], kb * 256, 0);
  };
  auto store_a1 = [&](int buf, int i) {
    const int c = tid + kPfThreads * i;
    *(u32x4_t*)(As + buf * (BM * 256) + pfp_off(c >> 4, c & 15)) = a_st[i];
  };
  auto load_b = [&](u32x4_t (&w)[NJ], uint32_t (&zs)[NJ], int kb, int grp) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      w[j] = __builtin_amdgcn_raw_buffer_load_b128(rw[j], lane16, kb * 1024, 0);
      zs[j] = __builtin_amdgcn_raw_buffer_load_b32(rz, r4, zoff_s[j] + grp * 64, 0);
    }
  };

  float4_t acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[mi][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // prologue: x tile 0 into LDS, k-block 0's weights / scales in registers, its first fragment dequantised
  load_a(0);
  int grp_n = k0 / kpg, cnt_n = k0 - grp_n * kpg;      // quantisation group of the next k-block, counted up (no division in the loop)
  load_b(w_cur, zs_nxt, 0, grp_n);
#pragma unroll
  for (int i = 0; i < AL; ++i) store_a1(0, i);
#pragma unroll
  for (int j = 0; j < NJ; ++j) zu[j] = zs_unpack(zs_nxt[j]);
  u32x4_t frag = rp_dequant(w_cur[0][0], zu[0].z1024, zu[0].z64, zu[0].s2);
  load_a(KB > 1 ? 1 : 0);                              // tile 1 travels while block 0 computes
  __syncthreads();
  u32x4_t af[MI], af_n[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) af[mi] = *(const u32x4_t*)(As + pfp_off(mi * 16 + r, q));

  // One barrier per k-block, in the middle of its last k-step: by then every wave has read all it needs from the
  // current x buffer (the d = 3 fragments are fetched during d = 2) and has written its share of the next tile into
  // the other buffer (during d = 2), so right behind the barrier the next block's first fragments can be fetched
  // in the shadow of the remaining MFMAs — no LDS latency is exposed at the block boundary.
  for (int kb = 0; kb < KB; ++kb) {
    const int nxt = kb + 1 < KB ? kb + 1 : kb;         // clamped, unconditional prefetch (no branch between load and use)
    const unsigned char* Ab = As + (kb & 1) * (BM * 256);
    const int nbuf = (kb + 1) & 1;
    const unsigned char* An = As + nbuf * (BM * 256);
    const int nx2 = kb + 2 < KB ? kb + 2 : KB - 1;
    if (kb + 1 < KB && ++cnt_n == kpg) { cnt_n = 0; ++grp_n; }
    load_b(w_nxt, zs_nxt, nxt, grp_n);
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        // the fragment after (d, j): (d, j + 1), then (d + 1, 0), then k-block kb + 1's (0, 0)
        DqPipe p;
        ZsU zn;
        if (j < NJ - 1) { p.w = w_cur[j + 1][d]; zn = zu[j + 1]; }
        else if (d < 3) { p.w = w_cur[0][d + 1]; zn = zu[0]; }
        else { uint32_t w0 = w_nxt[0][0]; pin_here(w0); p.w = w0; zn = zu_nxt[0]; }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          mfma_tied(acc[mi][j], af[mi], frag);
          if constexpr (!(PF_ABL & 1) && MI == 8) switch (mi) {      // (compile-time after unrolling)
            case 0: dq_stage<0>(p, zn.z1024, zn.z64, zn.s2); break;
            case 1: dq_stage<1>(p, zn.z1024, zn.z64, zn.s2); break;
            case 2: dq_stage<2>(p, zn.z1024, zn.z64, zn.s2); break;
            case 3: dq_stage<3>(p, zn.z1024, zn.z64, zn.s2); break;
            case 4: dq_stage<4>(p, zn.z1024, zn.z64, zn.s2); break;
            case 5: dq_stage<5>(p, zn.z1024, zn.z64, zn.s2); break;
            case 6: dq_stage<6>(p, zn.z1024, zn.z64, zn.s2); break;
            default: break;
          }
          if constexpr (!(PF_ABL & 1) && MI == 4) switch (mi) {
            case 0: dq_stage<0>(p, zn.z1024, zn.z64, zn.s2); dq_stage<1>(p, zn.z1024, zn.z64, zn.s2); break;
            case 1: dq_stage<2>(p, zn.z1024, zn.z64, zn.s2); dq_stage<3>(p, zn.z1024, zn.z64, zn.s2); break;
            case 2: dq_stage<4>(p, zn.z1024, zn.z64, zn.s2); dq_stage<5>(p, zn.z1024, zn.z64, zn.s2); break;
            default: dq_stage<6>(p, zn.z1024, zn.z64, zn.s2); break;
          }
          // passengers of the free slots
          // (read in the order 7 .. 0: the first consumer, row tile 0, then waits for the youngest read, and the compiler emits one
          // s_waitcnt per k-step instead of eight; the same for the x-tile registers written to LDS youngest first)
          if (!(PF_ABL & 2) && d < 3 && j == 1) af_n[MI - 1 - mi] = *(const u32x4_t*)(Ab + pfp_off((MI - 1 - mi) * 16 + r, (d + 1) * 4 + q));   // next k-step's x fragments
          // next x tile -> the other buffer, one 1 KiB write every fourth MFMA (eight in a row from all four waves
          // collide with the fragment reads), then the tile after it is requested at once: the 16 workgroups of a
          // row of tiles ask for the same fresh lines together, so they take about a k-block to arrive
          if constexpr (NJ == 4) {
            if (!(PF_ABL & 4) && ((d == 1 && j >= 2) || (d == 2 && j <= 1)) && (mi & 3) == 0)
              store_a1(nbuf, AL - 1 - (((d - 1) * 4 + j - 2) * (MI / 4) + (mi >> 2)));
            if (!(PF_ABL & 4) && d == 2 && j == 2 && mi == 0) load_a(nx2);
            if (d == 2 && j == 3 && mi == MI - 1) { pin_here(zs_nxt[0]); zu_nxt[0] = zs_unpack(zs_nxt[0]); }
          } else {
            // three fragments per k-step: the eight writes ride in (d, j) = (1, 2), (2, 0), (2, 2) behind MFMAs 0, 3, 6 (the j = 1
            // slots carry the fragment reads), the next tile is requested behind the last of them
            // (two per k-step: four writes each in (1, 0) and (2, 0), behind every second MFMA)
            const int slot = NJ == 3 ? ((d == 1 && j == 2) ? 0 : (d == 2 && j == 0) ? 1 : (d == 2 && j == 2) ? 2 : -1)
                                     : ((d == 1 && j == 0) ? 0 : (d == 2 && j == 0) ? 1 : -1);     // (compile-time after unrolling)
            constexpr int PER = NJ == 3 ? 3 : 2, CNT = (MI + PER - 1) / PER;
            if (!(PF_ABL & 4) && slot >= 0 && mi % PER == 0 && slot * CNT + mi / PER < AL) store_a1(nbuf, AL - 1 - (slot * CNT + mi / PER));
            if (!(PF_ABL & 4) && d == 2 && j == NJ - 1 && mi == MI - 1) load_a(nx2);
            if (d == 2 && j == 1 && mi == MI - 1) { pin_here(zs_nxt[0]); zu_nxt[0] = zs_unpack(zs_nxt[0]); }
          }
          if (!(PF_ABL & 8) && d == 3 && j == 0 && mi == 0) __syncthreads();
          if (!(PF_ABL & 2) && d == 3 && j == 1) af_n[MI - 1 - mi] = *(const u32x4_t*)(An + pfp_off((MI - 1 - mi) * 16 + r, q));    // next k-block's first fragments
          if (d == 3 && j < NJ - 1 && mi == MI - 1) { pin_here(zs_nxt[j + 1]); zu_nxt[j + 1] = zs_unpack(zs_nxt[j + 1]); }
          __builtin_amdgcn_sched_barrier(0);
        }
        frag = p.f;
      }
      if (!(PF_ABL & 2)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) af[mi] = af_n[mi];
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { w_cur[j] = w_nxt[j]; zu[j] = zu_nxt[j]; }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results must have left the pipe before the (compiler-scheduled) reads below

  if constexpr (MOE) {
    // values as the other routes round them (routed weight on the fp32 sums, SiLU-mul over the (gate, up) pairs of groups), parked wave-privately
    // in the idle x buffers 64 rows at a time, then every lane stores 16 bytes of its pair's output row (rows scatter by pair index)
    constexpr int WO = EPI == 1 ? NJ * 8 : NJ * 16;               // output columns of this wave
    const int width = EPI == 1 ? N / 2 : N;
    const int n_wave = EPI == 1 ? (cg_tile >> 1) * 16 : cg_tile * 16;         // first output column of this wave
    __syncthreads();
    half_t* const tw = (half_t*)As + (size_t)wn * (64 * WO);
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rl = mi * 16 + 4 * q + i;
          const int pr = moe.row_map[bm + half * 64 + rl];
          const float sc = (pr >= 0 && moe.slot_scale) ? moe.slot_scale[pr] : 1.f;
          if constexpr (EPI == 1) {
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
              const float xg = (float)(half_t)(acc[half * 4 + mi][j][i] * sc);       // the unfused path rounds gate_up to fp16 first
              tw[rl * WO + (j >> 1) * 16 + r] = (half_t)(xg / (1.f + __expf(-xg))) * (half_t)(acc[half * 4 + mi][j + 1][i] * sc);
            }
          } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) tw[rl * WO + j * 16 + r] = (half_t)(acc[half * 4 + mi][j][i] * sc);
          }
        }
#pragma unroll
      for (int p = 0; p < WO / 8; ++p) {
        const int f = (p * 64 + lane) * 8;
        const int row = f / WO, col = f - row * WO;
        const int pr = moe.row_map[bm + half * 64 + row];
        const int n = n_wave + col;
        // (a 16-column output group is complete or absent: cg_end and N / width are multiples of what a group covers; width % 8 == 0)
        const bool in = EPI == 1 ? (cg_tile + 2 * (col >> 4) + 1 < cg_end) : (cg_tile + (col >> 4) < cg_end);
        if (pr >= 0 && in && n < width) *(u32x4_t*)((half_t*)y + (size_t)pr * width + n) = *(const u32x4_t*)(tw + f);
      }
    }
  } else if constexpr (SPLIT) {
    // fp32 partial tile -> part[slice][M][N].  Straight from the accumulators a store instruction covers 4 rows x 64 bytes; through the
    // (now idle) x buffers of LDS, wave-private, 64 rows at a time, every lane stores 16 bytes of a 256-byte (NJ = 4) or 128-byte row run.
    static_assert(MI == 8, "two halves of 64 rows");
    constexpr int WC = NJ * 16;                                   // columns of this wave
    __syncthreads();                                              // every wave has read its last x fragments: the buffers are free
    float* const tw = (float*)As + (size_t)wn * (64 * WC);        // 64 rows x WC floats: 16 KiB per wave at NJ = 4 (4 waves: the 64 KiB)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) tw[(mi * 16 + 4 * q + i) * WC + j * 16 + r] = acc[half * 4 + mi][j][i];
      // (a wave's LDS accesses complete in order: no barrier between its own writes and reads)
#pragma unroll
      for (int p = 0; p < WC / 4; ++p) {
        const int f = (p * 64 + lane) * 4;
        const int row = f / WC, col = f - row * WC;
        const int m = bm + half * 64 + row, cg = cg_tile + (col >> 4), n = cg * 16 + (col & 15);
        const float4_t v = *(const float4_t*)(tw + f);
        if (m < M && cg < cg_end && n < N) *(float4_t*)(split.part + ((size_t)slice * M + m) * N + n) = v;
      }
    }
  } else if constexpr (MI == 8 && PF_LDS_EPILOGUE) {
    // the same route for the fp16 output tile: rounded (and biased) exactly as store_output does, parked in the idle x buffers, stored as
    // 16 bytes per lane (8 lanes per 128-byte row run of this wave's 64 columns; straight from the accumulators: 4 rows x 32 bytes a store)
    constexpr int WC = NJ * 16;
    __syncthreads();
    half_t* const tw = (half_t*)As + (size_t)wn * (64 * WC);
    const bool n_ok8 = (N % 8) == 0;                               // (16-byte stores need 8-column alignment; otherwise element stores below)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int n = (cg_tile + j) * 16 + r;
            half_t h = (half_t)acc[half * 4 + mi][j][i];
            if (bias && cg_tile + j < cg_end && n < N) h = h + ((const half_t*)bias)[n];
            tw[(mi * 16 + 4 * q + i) * WC + j * 16 + r] = h;
          }
#pragma unroll
      for (int p = 0; p < WC / 8; ++p) {
        const int f = (p * 64 + lane) * 8;
        const int row = f / WC, col = f - row * WC;
        const int m = bm + half * 64 + row, cg = cg_tile + (col >> 4), n = cg * 16 + (col & 15);
        if (m < M && cg < cg_end && n < N) {
          if (n_ok8) {
            *(u32x4_t*)((half_t*)y + (size_t)m * N + n) = *(const u32x4_t*)(tw + f);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (n + e < N) ((half_t*)y)[(size_t)m * N + n + e] = tw[f + e];
          }
        }
      }
    }
  } else {
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
  }
}

// y[m][n] = fp16(sum over slices of part[s][m][n]) (+ bias): slice order, one rounding — the second launch of the split-K tile route
__global__ __launch_bounds__(256) void pf_splitk_reduce_kernel(const float* __restrict__ part, const void* __restrict__ bias, void* __restrict__ y,
                                                               int S, size_t MN, int N) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (size_t)gridDim.x * 1024) {
    float4_t v = *(const float4_t*)(part + i);
    for (int s2 = 1; s2 < S; ++s2) v += *(const float4_t*)(part + (size_t)s2 * MN + i);
    const int n = (int)(i % (size_t)N);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_output<AWQ_DTYPE_F16>(y, i + j, v[j], bias, n + j);
  }
}

template <int NJ, int MI = 8>
static int pf_launch_region(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int cg_base, int ng_region) {
  constexpr int BM = MI * 16;
  const int nbx = (ng_region + 4 * NJ - 1) / (4 * NJ), nby = (a.M + BM - 1) / BM;
  const size_t lds = 2 * BM * 256;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_pipelined_kernel<NJ, false, 0, MI>, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_repacked_pipelined_kernel<NJ, false, 0, MI>), dim3(nbx * nby), dim3(kPfThreads), lds, a.stream, (const uint16_t*)a.x,
                     a.ldx, qw_r, zs_r, a.bias, a.y, a.M, a.K, a.N, a.g, NG, nbx, nby, cg_base, ng_region, PfMoe{nullptr, nullptr, nullptr, 0ll, 1},
                     PfSplit{nullptr, 1, 0});
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// AWQ-MoE over expert-sorted 128-row blocks: `num_blocks` row tiles x ceil(NG / 16) column tiles of 256 (the remainder tile is clamped)
template <int EPI, int MI>
static int pf_launch_moe(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int num_blocks, const PfMoe& moe) {
  const int nbx = (NG + 15) / 16, nby = num_blocks;
  const size_t lds = 2 * MI * 16 * 256;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_pipelined_kernel<4, true, EPI, MI>, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  hipLaunchKernelGGL((gemm_repacked_pipelined_kernel<4, true, EPI, MI>), dim3(nbx * nby), dim3(kPfThreads), lds, a.stream, (const uint16_t*)a.x,
                     a.ldx, qw_r, zs_r, nullptr, a.y, num_blocks * MI * 16, a.K, a.N, a.g, NG, nbx, nby, 0, NG, moe, PfSplit{nullptr, 1, 0});
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int launch_gemm_repacked_moe_tiles(const GemmArgs& a, const void* packed_experts, const int* row_map, const int* block_expert, int num_blocks,
                                   int block_rows, const float* slot_scale, long long expert_stride, int x_div, bool silu_mul) {
  if (!pipelined_addressable(a)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  if (silu_mul && (NG & 1)) return AWQ_ERR_BAD_VARIANT;
  const u32x4_t* qw_r = (const u32x4_t*)packed_experts;
  const uint32_t* zs_r = (const uint32_t*)packed_experts + (size_t)NG * (a.K / 128) * 256;
  const PfMoe moe = {row_map, block_expert, slot_scale, expert_stride, x_div};
  if (block_rows == 64) return silu_mul ? pf_launch_moe<1, 4>(a, qw_r, zs_r, NG, num_blocks, moe) : pf_launch_moe<0, 4>(a, qw_r, zs_r, NG, num_blocks, moe);
  if (block_rows != 128) return AWQ_ERR_BAD_SHAPE;
  return silu_mul ? pf_launch_moe<1, 8>(a, qw_r, zs_r, NG, num_blocks, moe) : pf_launch_moe<0, 8>(a, qw_r, zs_r, NG, num_blocks, moe);
}

// 32-bit offsets inside the kernel's buffer descriptors: a 128-row x tile, the scale words and one column group's strip of weights
// must each span less than 2 GiB (any real layer does; callers fall back to the generic tiles otherwise).
bool pipelined_addressable(const GemmArgs& a) {
  const int64_t NG = rp_groups(a.N);
  return a.ldx < (int64_t(1) << 22) && NG * (a.K / a.g) * 64 < (int64_t(1) << 31) && (int64_t)(a.K / 128) * 1024 < (int64_t(1) << 31);
}

// Split-K tile route: S slices x tiles workgroups write fp32 partials into the workspace (behind its 4096-byte counter head), a second
// launch adds them.  NJ = 2 (128 x 128) or 4 (128 x 256) tiles.
template <int NJ>
static int pf_launch_split(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int S, int kb_per) {
  const int nbx = (NG + 4 * NJ - 1) / (4 * NJ), nby = (a.M + kPfBM - 1) / kPfBM;
  const size_t lds = 2 * kPfBM * 256;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_pipelined_kernel<NJ, false, 0, 8, true>, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  float* part = (float*)((char*)a.workspace + kPfSplitHead);
  hipLaunchKernelGGL((gemm_repacked_pipelined_kernel<NJ, false, 0, 8, true>), dim3(nbx * nby * S), dim3(kPfThreads), lds, a.stream, (const uint16_t*)a.x,
                     a.ldx, qw_r, zs_r, nullptr, a.y, a.M, a.K, a.N, a.g, NG, nbx, nby, 0, NG, PfMoe{nullptr, nullptr, nullptr, 0ll, 1},
                     PfSplit{part, S, kb_per});
  const size_t MN = (size_t)a.M * a.N;
  const unsigned blocks = (unsigned)((MN / 4 + 255) / 256 < 2048 ? (MN / 4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(pf_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, a.stream, (const float*)part, a.bias, a.y, S, MN, a.N);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// Plan of the split-K tile route: false = not applicable (no workspace, too many tiles, too few k-blocks, N % 4)
bool pf_split_plan(const GemmArgs& a, int* NJ_out, int* S_out, int* kb_per_out) {
  if (!a.workspace || (((uintptr_t)a.workspace) & 15) || a.workspace_bytes <= kPfSplitHead || a.N % 4 || !pipelined_addressable(a)) return false;
  static const int env_s = lab_env("AWQ_PF_SK_S", 0), env_nj = lab_env("AWQ_PF_SK_NJ", 0);       // lab knobs: force slices / tile width (S = 1: off)
  if (env_s == 1) return false;
  const int NG = rp_groups(a.N), KBT = a.K / 128, nby = (a.M + kPfBM - 1) / kPfBM;
  const size_t slice_bytes = (size_t)a.M * a.N * sizeof(float);
  const int s_ws = (int)((a.workspace_bytes - kPfSplitHead) / slice_bytes);
  // Where it pays (profiles/r03_kbench_split_tiles_ab.txt): few wide tiles (<= 48, or <= 128 on deep matrices with >= 64 k-blocks), and from 65
  // rows unless the matrix is narrow (<= 32 wide tiles) — below that the GEMV passes win on wide matrices.  128 x 128 tiles while there are at
  // most 64 of them (fewer slices for the same number of workgroups: less partial traffic), 128 x 256 beyond.
  const int wide_tiles = ((NG + 15) / 16) * nby;
  if (!env_s && !env_nj) {
    const bool few = wide_tiles <= route::kPfSplitFewWideTiles || (KBT >= route::kPfSplitDeepBlocks && wide_tiles <= route::kPfSplitMaxWideTiles);
    if (!few || (a.M <= 64 && wide_tiles > route::kPfSplitNarrowWideTiles)) return false;
  }
  int best_nj = 0, best_s = 0;
  const int tiles2 = ((NG + 7) / 8) * nby;
  for (int pass = 0; pass < 2 && !best_nj; ++pass) {
    const int nj = env_nj ? env_nj : ((tiles2 <= route::kPfSplitNarrowTilesMax) == (pass == 0) ? 2 : 4);
    const int tiles = ((NG + 4 * nj - 1) / (4 * nj)) * nby;
    int S = env_s ? env_s : 256 / tiles;                  // fill the chip once
    if (S > route::kPfSplitMaxSlices) S = route::kPfSplitMaxSlices;
    if (S > s_ws) S = s_ws;
    if (S > KBT / route::kPfSplitMinBlocks) S = KBT / route::kPfSplitMinBlocks;
    if (S >= 2) { best_nj = nj; best_s = S; }
    if (env_nj) break;
  }
  if (!best_nj) return false;
  const int kb_per = (KBT + best_s - 1) / best_s;
  *NJ_out = best_nj; *S_out = (KBT + kb_per - 1) / kb_per; *kb_per_out = kb_per;
  return *S_out >= 2;
}

// Scratch the split-K tile route would use for this shape (0: not applicable), at most kPfSplitHead + 32 MiB (the budget of the Python
// shim's per-stream buffer): the plan with as many slices as fit.
size_t pf_split_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < route::kPfSplitMinRows || N % 4 || K % 128) return 0;
  const int64_t wide_tiles = ((M + 127) / 128) * ((N + 255) / 256);
  if (wide_tiles > route::kPfSplitMaxWideTiles) return 0;
  const size_t slice_bytes = (size_t)M * N * sizeof(float), budget = (size_t)32 << 20;
  int S = (int)(256 / wide_tiles) * 2;                      // (the narrow tiling may take twice the slices of the wide one; the plan decides)
  if (S > route::kPfSplitMaxSlices) S = route::kPfSplitMaxSlices;
  if (S > (int)(K / 128) / route::kPfSplitMinBlocks) S = (int)(K / 128) / route::kPfSplitMinBlocks;
  if ((size_t)S * slice_bytes > budget) S = (int)(budget / slice_bytes);
  return S >= 2 ? kPfSplitHead + (size_t)S * slice_bytes : 0;
}

int launch_gemm_repacked_split_tiles(const GemmArgs& a, const void* packed) {
  int NJ = 0, S = 0, kb_per = 0;
  if (!pf_split_plan(a, &NJ, &S, &kb_per)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  return NJ == 4 ? pf_launch_split<4>(a, qw_r, zs_r, NG, S, kb_per) : pf_launch_split<2>(a, qw_r, zs_r, NG, S, kb_per);
}

int launch_gemm_repacked_pipelined(const GemmArgs& a, const void* packed) {
  if (!pipelined_addressable(a)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  // Tile quantisation: nA column tiles of 256 (16 groups) + the remaining groups in tiles of 192 (12 groups, 0.75 of the time).
  // Cost in rounds of the 256 CUs: ceil(tiles_A / 256) + 0.84 ceil(tiles_B / 256) (+ a little for the second launch); the
  // smallest wins, ties go to fewer launches / wider tiles.  2048 x 11008: 32 x 16 = 512 wide tiles (2 rounds) + 15 x 16 = 240
  // narrow ones (0.75) instead of 688 wide ones (3 rounds).  AWQ_PF_SPLIT=0 keeps the single launch (A/B).
  static const bool env_split = lab_env("AWQ_PF_SPLIT", 1) != 0;
  const int nby = (a.M + kPfBM - 1) / kPfBM, nA_max = (NG + 15) / 16;
  int best_nA = nA_max;
  double best = 1e30;
  for (int nA = nA_max; nA >= 0 && env_split; --nA) {
    const int rest = NG - 16 * nA > 0 ? NG - 16 * nA : 0;
    const int nB = (rest + 11) / 12;
    if (nA == 0 && nB == 0) continue;
    // a round of narrow tiles measured 0.84 of a round of wide ones (62 vs 73.5 us at K = 4096: the x-tile staging does not shrink)
    const double cost = (double)((nA * nby + 255) / 256) + route::kNarrowRound192 * (double)((nB * nby + 255) / 256) + (nA > 0 && nB > 0 ? route::kSecondLaunch : 0.0);
    if (cost < best - 1e-9) { best = cost; best_nA = nA; }
  }
  const int gA = best_nA * 16 < NG ? best_nA * 16 : NG;
  // Few row tiles (M up to ~384 on a wide matrix): 128 x 128 tiles (NJ = 2) fill more CUs per round; a round of them measured
  // kNarrow2 of a round of wide ones.  Taken only when it beats the wide / 192-wide split.
  constexpr double kNarrow2 = route::kNarrowRound128;      // 43.5 us for 172 tiles against 73.5 for a round of wide ones (M = 256, 4096 x 11008)
  static const int env_nj2 = lab_env("AWQ_PF_NJ2", -1);      // lab knob: 0 never, 1 always
  const double cost2 = kNarrow2 * (double)((((NG + 7) / 8) * nby + 255) / 256);
  if (env_nj2 == 1 || (env_nj2 != 0 && env_split && cost2 < best - 1e-9)) return pf_launch_region<2>(a, qw_r, zs_r, NG, 0, NG);
#ifdef AWQ_LAB
  static const int env_mi4 = lab_env("AWQ_PF_MI4", 0);     // lab: 64-row tiles (two workgroups per CU); 1 = one launch of 256-wide tiles, 2 = with the split
  if (env_mi4 == 1) return pf_launch_region<4, 4>(a, qw_r, zs_r, NG, 0, NG);
  if (env_mi4 == 2) {
    if (gA > 0) {
      const int rc = pf_launch_region<4, 4>(a, qw_r, zs_r, NG, 0, gA);
      if (rc) return rc;
    }
    if (gA < NG) return pf_launch_region<3, 4>(a, qw_r, zs_r, NG, gA, NG - gA);
    return AWQ_OK;
  }
#endif
  if (gA > 0) {
    const int rc = pf_launch_region<4>(a, qw_r, zs_r, NG, 0, gA);
    if (rc) return rc;
  }
  if (gA < NG) return pf_launch_region<3>(a, qw_r, zs_r, NG, gA, NG - gA);
  return AWQ_OK;
}


// ------------------------------------------------------------------------------------------ middle M, few wide tiles
// 128 x 64 tiles, K split inside the workgroup.  The 128 x 256 tiles above give only N / 256 x M / 128 workgroups
// (16 per 128 rows for N = 4096), and 32-row GEMV passes re-stream the weight per pass.  Here a workgroup owns
// 128 rows x 64 columns and its four waves split every 128-deep K block by k-step (wave w takes rows
// k = 32 w .. 32 w + 31 of each block: dword w of the fragment-major dwordx4), so a wave still reuses each dequantised
// fragment for 8 row tiles; the four partial tiles are summed in fixed order through LDS at the end (no workspace,
// deterministic).  Four times the workgroups of the wide tile, the weight streamed once.
// Hand-pipelined like the 128 x 256 kernel.  Still ~1.1 us per K block against ~0.35 us of MFMA work (the compiler-scheduled
// form: 1.25 us; removing its x loads, LDS staging and barrier together only brought 48 us down to 33 at M = 128,
// 4096 x 11008), so it is dispatched only where it measured faster than the alternatives: 11008 x 4096 at M = 256 / 512:
// 142 -> 90 / 97 us; 4096 x 11008 at M = 128: 54 -> 40 us.
__global__ __launch_bounds__(kPfThreads, 1) void gemm_repacked_ksplit_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                             const uint32_t* __restrict__ qw_r,
                                                                             const uint32_t* __restrict__ zs_r,
                                                                             const void* __restrict__ bias, void* __restrict__ y, int M,
                                                                             int K, int N, int g, int NG, int nbx, int nby) {
  extern __shared__ __attribute__((aligned(16))) unsigned char As[];      // 2 x 32 KiB of x, later 4 x 32 KiB of partial tiles
  constexpr int MI = 8, AL = 8, BN = 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, r = lane & 15;
  const int KB = K / 128, groups = K / g;

  const int nwg = nbx * nby, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int bm = (logical / nbx) * kPfBM;
  const int bn = (logical % nbx) * BN;

  size_t woff[4], zoff[4];                              // dword offsets of this lane's weight dword / scale word, k-block 0
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int c = bn / 16 + j;
    if (c >= NG) c = NG - 1;                             // clamped: columns >= N are never stored
    woff[j] = ((size_t)c * KB * 64 + lane) * 4 + wave;
    zoff[j] = (size_t)c * groups * 16 + r;
  }
  const uint16_t* xrow[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int c = tid + kPfThreads * i;
    const int row = c >> 4, chunk = c & 15;
    const int m = bm + row < M ? bm + row : M - 1;
    xrow[i] = x + (size_t)m * ldx + chunk * 8;
  }
  // Two x tiles and two sets of weight dwords in flight (the loop is unrolled by two so their registers alternate): a
  // K block is only ~0.35 us of MFMA work here, a load takes ~1 us — with one tile in flight the kernel ran at the
  // latency-bound 32 KB per us per CU (1.25 us per K block).
  u32x4_t a_st[2][AL];
  uint32_t w_buf[2][4], zs_buf[2][4];
  auto load_a = [&](u32x4_t (&dst)[AL], int kb) {
#pragma unroll
    for (int i = 0; i < AL; ++i) dst[i] = *(const u32x4_t*)(xrow[i] + kb * 128);
  };
  auto store_a = [&](const u32x4_t (&src)[AL], int buf) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int c = tid + kPfThreads * i;
      *(u32x4_t*)(As + buf * (kPfBM * 256) + pfp_off(c >> 4, c & 15)) = src[i];
    }
  };
  auto load_b = [&](uint32_t (&w)[4], uint32_t (&zs)[4], int kb) {
    const int grp = (kb * 128) / g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = qw_r[woff[j] + (size_t)kb * 256];
      zs[j] = zs_r[zoff[j] + (size_t)grp * 16];
    }
  };

  float4_t acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[mi][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

  // Tile t travels in a_st[t & 1], is written to LDS buffer t & 1 during step t - 1 and consumed in step t; the weights of
  // block t sit in w_buf[t & 1] until step t - 1 has turned them into fragments.  One step = 32 MFMAs per wave (4 fragments
  // x 8 row tiles); behind each MFMA rides one of the 28 dequantise stages of the NEXT block's four fragments (as in the
  // 128 x 256 kernel above), the eight LDS writes of the next x tile ride in j = 1, the one barrier per block sits after
  // them, and the next block's x fragments are fetched behind it during j = 3 — no LDS or load latency at the block boundary.
  load_a(a_st[0], 0);
  load_b(w_buf[0], zs_buf[0], 0);
  load_b(w_buf[1], zs_buf[1], KB > 1 ? 1 : 0);
  store_a(a_st[0], 0);
  load_a(a_st[1], KB > 1 ? 1 : 0);
  u32x4_t frag[4], frag_n[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ZsU u = zs_unpack(zs_buf[0][j]);
    frag[j] = rp_dequant(w_buf[0][j], u.z1024, u.z64, u.s2);
  }
  __syncthreads();
  u32x4_t af[MI], af_n[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) af[mi] = *(const u32x4_t*)(As + pfp_off(mi * 16 + r, wave * 4 + q));

  auto step = [&](auto P_, int kb) {
    constexpr int P = decltype(P_)::value;
    const int k2 = kb + 2 < KB ? kb + 2 : KB - 1;       // clamped, unconditional prefetch
    // block kb + 1's weights (requested two steps ago) become this step's passengers; block kb's registers are free again
    uint32_t wn[4];
    ZsU zn[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      wn[j] = w_buf[P ^ 1][j];
      uint32_t z = zs_buf[P ^ 1][j];
      pin_here(wn[j]);
      pin_here(z);
      zn[j] = zs_unpack(z);
    }
    load_a(a_st[P], k2);                                 // (tile kb already sits in LDS buffer P, its fragments in af)
    load_b(w_buf[P], zs_buf[P], k2);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* An = As + (P ^ 1) * (kPfBM * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      DqPipe pp;
      pp.w = wn[j];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        mfma_tied(acc[mi][j], af[mi], frag[j]);
        switch (mi) {                                    // (compile-time after unrolling)
          case 0: dq_stage<0>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 1: dq_stage<1>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 2: dq_stage<2>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 3: dq_stage<3>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 4: dq_stage<4>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 5: dq_stage<5>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          case 6: dq_stage<6>(pp, zn[j].z1024, zn[j].z64, zn[j].s2); break;
          default: break;
        }
        if (j == 1) {                                    // tile kb + 1 (requested a full step ago) -> the other LDS buffer
          const int c = tid + kPfThreads * mi;
          *(u32x4_t*)(As + (P ^ 1) * (kPfBM * 256) + pfp_off(c >> 4, c & 15)) = a_st[P ^ 1][mi];
        }
        if (j == 2 && mi == 7) __syncthreads();
        if (j == 3) af_n[mi] = *(const u32x4_t*)(An + pfp_off(mi * 16 + r, wave * 4 + q));
        __builtin_amdgcn_sched_barrier(0);
      }
      frag_n[j] = pp.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) frag[j] = frag_n[j];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[mi] = af_n[mi];
  };
  for (int kb = 0; kb < KB; kb += 2) {
    step(std::integral_constant<int, 0>{}, kb);
    if (kb + 1 < KB) step(std::integral_constant<int, 1>{}, kb + 1);
  }
  __syncthreads();                                       // every wave is done with the x buffers: they become the reduction scratch
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");     // the last MFMAs' results must have left the pipe before the reads below

  // partial tiles -> LDS [wave][mi][j][i][lane] (conflict-free: consecutive lanes, consecutive floats), summed in wave order
  float* red = (float*)As;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[(((wave * MI + mi) * 4 + j) * 4 + i) * 64 + lane] = acc[mi][j][i];
  __syncthreads();
  // thread t finishes (mi = 2 * wave + {0, 1}, all j, i) for its lane: 32 outputs
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int mi = wave * 2 + h;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = bm + mi * 16 + 4 * q + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = bn + j * 16 + r;
        float v = red[(((0 * MI + mi) * 4 + j) * 4 + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) v += red[(((w * MI + mi) * 4 + j) * 4 + i) * 64 + lane];
        if (m < M && n < N) store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, v, bias, n);
      }
    }
  }
}

int launch_gemm_repacked_ksplit(const GemmArgs& a, const void* packed) {
  if (!repacked_fast(a.K, a.N, a.g, a.dtype) || a.M < 1 || a.ldx % 8 || (((uintptr_t)a.x) & 15)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const uint32_t* qw_r = (const uint32_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  const int nbx = (a.N + 63) / 64, nby = (a.M + kPfBM - 1) / kPfBM;
  const size_t lds = 4 * 8 * 4 * 4 * 64 * sizeof(float);                  // 128 KiB: four partial tiles (covers the 64 KiB of x buffers)
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_ksplit_kernel, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_repacked_ksplit_kernel, dim3(nbx * nby), dim3(kPfThreads), lds, a.stream, (const uint16_t*)a.x, a.ldx, qw_r, zs_r,
                     a.bias, a.y, a.M, a.K, a.N, a.g, NG, nbx, nby);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // namespace awq
