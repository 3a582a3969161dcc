// One place for the (M, K, N) -> kernel choices on the fragment-major layout and the measured crossover points behind them.
// Every threshold the launchers use is a named constant here; the launchers (awq_capi.hip: repacked_dispatch; awq_repacked.hip:
// launch_gemv_repacked; awq_repacked_splitk.hip: rps_plan; awq_repacked_prefill.hip: launch_gemm_repacked_pipelined) only apply them.
// fp16, group size a multiple of 128 unless noted; bf16 and g in {32, 64} take the generic kernels of awq_repacked_ext.hip at every M.
//
//  rows M        matrix                                   kernel                                        measured (us; source)
//  ------------  ---------------------------------------  --------------------------------------------  ------------------------------------------------
//  1..16         >= 16 k-blocks, streamed (>= 12 MiB) or   gemv_rp2_kernel (straight-line, 16 waves)      4096 x 11008 at 1 row 7.0 -> 6.3 (round 2, gemv_lab);
//                small; M T <= 32 staging chunks            ring of 2 loads per wave, x + zs staged once   4096 x 4096 4.39 -> 4.07; 1024 x 8192 (8 k-blocks) stays 8-wave
//  1             17..32 units per wave (G T <= 32, T <= 16) gemv_rp2_kernel, long form                     8192 x 28672 25.9 -> 24.0; 28672 x 8192 26.4 -> 23.9;
//                                                                                                         14336 x 4096 11.2 -> 8.0 (r03_kbench_long_rp2_ab.txt)
//  13..16        strips of <= 2 column groups, deep K      gemv_rp3_kernel (loop form of the ring)        11008 x 4096 at 16 rows 14.4 -> 13.1; 8192 x 1280 10.4 -> 8.5;
//                                                                                                         slower at 6..12 rows and on wide strips (r03_kbench_rp3_ab.txt)
//  9..32         K >= 8192 and N <= 4096, workspace given  gemv_rps_kernel (K split over workgroups)      11008 x 4096 at 12 / 16 / 32 rows 12.2 -> 10.6 / 14.7 -> 11.3 /
//                                                                                                         23.1 -> 17.2; not at K = 4096 or N = 8192 (r02_kbench_splitk_ab.txt)
//  2..32 (rest)  anything else, N <= 32768                 gemv_repacked_kernel (round 1: 8 / 16 waves,    17..32 rows: two row tiles per fragment, 8 waves
//                                                          straight-line or double-buffered loop)
//  1..32         N > 32768                                 gemv_repacked_kernel, rounds of 3-group strips  8192 x 57344: 145 -> 51 (G = 3; 53 / 54.5 / 60 at G = 2 / 4 / 1)
//  33..160       while passes cost less than the tiles     passes of <= 32 rows of the GEMV                a pass 10.3 / 13.2 / 19.7 / 22.8 us on 4096 x 4096 / x 11008 / x 22016 /
//                (gemv_passes_max: 2..4 passes)                                                           11008 x 4096; under-filled tiles ~1.12 us per k-block: 37-44 / 91 us
//                                                                                                         (r03_kbench_middle_rows_ab.txt: 4096 x 22016 at 128 rows 78.6 -> 44)
//  33..          few wide tiles, workspace given           gemm_repacked_pipelined_kernel with K split     11008 x 4096 at 128 / 256 rows 88.7 -> 26.4 / 88.9 -> 40.5; 8192 x 8192 at 128
//                (pf_split_plan)                           over workgroups + pf_splitk_reduce_kernel       rows 76.4 -> 35.2; 4096 x 11008 at 96 / 128 rows 40.4 -> 27.1 / 38.6 -> 28.2
//                                                                                                         (r03_kbench_split_tiles_ab.txt)
//  97..          <= 64 wide tiles, no workspace            gemm_repacked_ksplit_kernel (128 x 64 tiles)    11008 x 4096 at 256 / 512 rows 142 -> 90 / 97
//  161..         any                                       gemm_repacked_pipelined_kernel: 128 x 256 tiles 2048 x 4096 x 11008: 206.6 -> 200.3 with the wide + narrow
//                                                          for whole rounds + 128 x 192 for the rest,      split; 192 / 256 rows 51.5 -> 43 with 128 x 128 tiles
//                                                          or 128 x 128 when few row tiles                 (r02_kbench_prefill_split_ab.txt, …_nj2_ab.txt)
//  the torch op  M >= 33 without a cached copy             awq_repack into the workspace, then the above   11008 x 4096: M = 64 212 -> 63, M = 256 217 -> 118
//  AWQ-MoE       <= 12 (token, expert) pairs               gemv_rp2_kernel, one grid row per pair          ~17 per pair (E = 8, 4096 / 14336); blocks ~215: cross at ~13
//                more                                      gemv_rp3_kernel on expert-sorted 16-row blocks  (r03_time_moe.txt; threshold in sglang_awq_amd/moe.py)
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace awq {
namespace route {

// ---- rows -> route (awq_capi.hip: repacked_dispatch)
constexpr int kGemvMaxRows = 32;              // one launch of the streaming GEMV
constexpr int kGemvPassRows = 32;             // rows per pass when a batch runs as passes of the GEMV
constexpr int kGemvPassesMaxRows = 160;       // never more passes than this many rows' worth; fewer where the tiles are cheaper (gemv_passes_max)
// Passes of the 32-row GEMV against one launch of under-filled MFMA tiles (33 .. 160 rows).  Measured: a pass costs about
// 8 us + 0.26 us per MB of packed weight (2.5 x the per-MB term on deep matrices, K > 8192, where the 32-row kernel has no split-K form
// without a workspace); the tile kernel with at most one tile per CU takes about 1.12 us per k-block + 2 us (64 MFMAs per wave and block).
// Returns how many passes may run before the tiles win.
inline int gemv_passes_max(int64_t K, int64_t N) {
  const double mb = (double)K * (double)N / 2.0 / 1.0e6;
  const double t_pass = 8.0 + 0.26 * mb * (K > 8192 ? 2.5 : 1.0);
  const double t_tiles = 1.12 * (double)(K / 128) + 2.0;
  int p = (int)(t_tiles / t_pass + 0.35);
  return p < 1 ? 1 : (p > kGemvPassesMaxRows / kGemvPassRows ? kGemvPassesMaxRows / kGemvPassRows : p);
}
constexpr int kKsplitMinRows = 97;            // 128 x 64 K-split tiles from here ...
constexpr int kKsplitMaxWideTiles = 64;       // ... while the 128 x 256 tiling would give at most this many tiles
constexpr int kSplitKMinRows = 9;             // split-K GEMV (needs a workspace) from here
constexpr int kPfSplitMaxSlices = 8;          // split-K MFMA tiles (under-filled launches, workspace given): at most this many slices ...
constexpr int kPfSplitMinBlocks = 4;          // ... of at least this many k-blocks each
constexpr int kPfSplitMinRows = 33;           // ... from this many rows
constexpr int kPfSplitFewWideTiles = 48;      // ... when the 128 x 256 tiling gives at most this many tiles,
constexpr int kPfSplitDeepBlocks = 64;        // ... or, on matrices at least this many k-blocks deep,
constexpr int kPfSplitMaxWideTiles = 128;     // ... at most this many;
constexpr int kPfSplitNarrowWideTiles = 32;   // ... up to 64 rows only where even fewer (the GEMV passes win on wider matrices there)
constexpr int kPfSplitNarrowTilesMax = 64;    // 128 x 128 tiles while there are at most this many of them, 128 x 256 beyond
constexpr int64_t kRepackOnTheFlyMinM = 33;   // the op re-lays the weight out per call from here (no cached copy)

// ---- the GEMV family (awq_repacked.hip: launch_gemv_repacked)
constexpr size_t kStreamedMinBytes = 12u << 20;   // packed weight bytes from which a matrix is streamed from HBM: 16 waves, non-temporal loads
constexpr int kOneStripMaxGroups = 8;             // strip width (column groups) up to which one strip per CU is used; beyond: rounds
constexpr int kRoundsGroups = 3;                  // strip width in rounds mode
constexpr int kRp2MinKBlocks = 16;                // every one of the 16 waves has a k-block
constexpr int kRp3MinRows = 13;                   // loop form of the ring: from these rows ...
constexpr int kRp3MaxGroups = 2;                  // ... on strips of at most this many column groups

// ---- split-K GEMV plan (awq_repacked_splitk.hip: rps_plan)
constexpr int kRpsMinKBlocks = 64;                // K >= 8192
constexpr int kRpsMaxGroups = 256;                // N <= 4096
constexpr int kRpsStripGroups = 2;

// ---- prefill tile model (awq_repacked_prefill.hip), in rounds of the 256 CUs
constexpr double kNarrowRound192 = 0.84;          // a round of 128 x 192 tiles / a round of 128 x 256 tiles (62 vs 73.5 us at K = 4096)
constexpr double kNarrowRound128 = 0.65;          // a round of 128 x 128 tiles (43.5 us for 172 tiles at M = 256, 4096 x 11008)
constexpr double kSecondLaunch = 0.02;

}  // namespace route
}  // namespace awq
