// Prefill GEMM on the MFMA-fragment-major layout built on v_mfma_f32_32x32x16_f16 (gfx950).
//
// Same decomposition as gemm_repacked_pipelined_kernel (awq_repacked_prefill.hip): workgroup tile 128 x 64 CB x 4 waves, one wave
// per SIMD, B never touches LDS, x through double-buffered XOR-swizzled LDS, one barrier per 128-deep k-block, the 13 VALU ops that
// turn the NEXT packed dword into a B fragment riding in stages behind the MFMAs that consume the CURRENT one.  What changes is the
// matrix instruction: a wave issues the same VALU / LDS / VMEM instructions per k-block but HALF as many MFMAs, each holding the
// pipe 32 cycles — and tools/gemv_lab `hide` measures that a wave's own instruction stream hides ~6 independent VALU ops behind a
// 32x32x16 MFMA against 2 behind a 16x16x32 one (33 / 32.5 / 35.5 / 36.5 / 47 ticks for +0 / 2 / 4 / 6 / 8 ops; 19 / 19 / 23.5 / 26.5
// for +0 / 2 / 3 / 4).  The 16x16x32 kernel carries ~2.8 other instructions per MFMA, i.e. 5.6 per pair — over its budget, inside
// this one's.
//
// Operand mapping (lane = 32 h + c):
//   A (32 rows x 16 k):  row = c, k = 8 h .. 8 h + 7          -> one ds_read_b128 of the x tile, chunk 2 s + h of row 32 rb + c
//   B (16 k x 32 cols):  col = c, k = 8 h .. 8 h + 7          -> ONE packed dword of the existing layout: column group cg + c / 16,
//                        column c % 16, k-step d = s / 2, k-quad q = 2 (s % 2) + h.  A lane therefore fetches the 16 bytes of
//                        layout-lanes (h, c % 16) and (2 + h, c % 16) of "its" column group: two dwordx4 per 32-column block and
//                        k-block, every byte of the strip fetched once (four 256-byte segments per wave-load).
//   D (32 x 32):         col = c, row = 8 (v / 4) + 4 h + v % 4 for accumulator register v = 0 .. 15
// Numerics are those of rp_dequant (same operations in the same order per element); the k order inside an MFMA differs from the
// 16x16x32 kernel's, so results agree with it to fp32 summation order, not bit for bit.
#include <cstdlib>

#include "awq_prefill_common.h"

namespace awq {

typedef float float16v_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma32_tied(float16v_t& acc, const u32x4_t& a, const u32x4_t& b) {
  asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

struct Dq32 {              // one dword on its way to becoming a fragment: 13 VALU ops in four stages (4 + 3 + 3 + 3)
  uint32_t w, w8, t0, t1, t2, t3;
  half2_t d0, d1, d2, d3;
  u32x4_t f;
};
template <int S>
__device__ __forceinline__ void dq32_stage(Dq32& p, const ZsU& z) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t magic = kMagicF16;
  if constexpr (S == 0) { p.w8 = p.w >> 8; p.t0 = and_or(p.w, kLoNib, magic); p.t1 = and_or(p.w, kHiNib, magic); p.t2 = and_or(p.w8, kLoNib, magic); }
  if constexpr (S == 1) { p.t3 = and_or(p.w8, kHiNib, magic); p.d0 = as_h2(p.t0) - z.z1024; p.d1 = __builtin_elementwise_fma(as_h2(p.t1), sixteenth, -z.z64); }
  if constexpr (S == 2) { p.d2 = as_h2(p.t2) - z.z1024; p.d3 = __builtin_elementwise_fma(as_h2(p.t3), sixteenth, -z.z64); p.f[0] = as_u32(p.d0 * z.s2); }
  if constexpr (S == 3) { p.f[1] = as_u32(p.d1 * z.s2); p.f[2] = as_u32(p.d2 * z.s2); p.f[3] = as_u32(p.d3 * z.s2); }
}

constexpr int kP32Threads = 256;

// CB = 32-column blocks per wave: 2 -> 128 x 256 workgroup tiles, 1 -> 128 x 128.
template <int CB>
__global__ __launch_bounds__(kP32Threads, 1) void gemm_repacked_pipelined32_kernel(const uint16_t* __restrict__ x, int64_t ldx,
                                                                                   const u32x4_t* __restrict__ qw_r,
                                                                                   const uint32_t* __restrict__ zs_r,
                                                                                   const void* __restrict__ bias, void* __restrict__ y,
                                                                                   int M, int K, int N, int g, int NG, int nbx, int nby,
                                                                                   int cg_base, int ng_region) {
  extern __shared__ __attribute__((aligned(16))) unsigned char As[];      // 2 x 32 KiB
  constexpr int RB = 4, AL = 8;                      // 32-row blocks per wave; x-tile chunks (16 B) per thread
  const int tid = threadIdx.x, lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, c = lane & 31, r = lane & 15, gsel = (lane >> 4) & 1;
  const int KB = K / 128, groups = K / g, kpg = g / 128;

  const int nwg = nbx * nby, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rem = nwg & 7;
  const int logical = (xcd < rem ? xcd * (qd + 1) : rem * (qd + 1) + (xcd - rem) * qd) + (bid >> 3);
  const int bm = (logical / nbx) * kPfBM;
  const int cg_tile = cg_base + (logical % nbx) * (8 * CB) + wn * (2 * CB);       // this wave's first column group (2 per 32-column block)
  const int cg_end = cg_base + ng_region < NG ? cg_base + ng_region : NG;

  // buffer loads: one descriptor for the packed weights, one for the scale words, one for this tile's x rows; per-lane 32-bit offsets
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)qw_r, 0, 0x7fffffff, kPfRsrcFlags);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)zs_r, 0, 0x7fffffff, kPfRsrcFlags);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (size_t)bm * ldx), 0, 0x7fffffff, kPfRsrcFlags);
  uint32_t woff_a[CB], woff_b[CB], zoff[CB];         // byte offsets of this lane's two layout-lanes / its scale word, k-block 0
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) {
    int gi = cg_tile + 2 * cb + gsel;
    gi = gi < cg_end ? gi : cg_end - 1;              // clamped: groups outside the region are never stored
    woff_a[cb] = ((uint32_t)gi * (uint32_t)KB * 64u + (uint32_t)(h * 16 + r)) * 16u;
    woff_b[cb] = woff_a[cb] + 32u * 16u;             // layout-lane (2 + h, r)
    zoff[cb] = ((uint32_t)gi * (uint32_t)groups * 16u + (uint32_t)r) * 4u;
  }

  u32x4_t a_st[AL];
  u32x4_t wa[CB], wb[CB], wa_n[CB], wb_n[CB];        // even / odd k-steps of the current and the next k-block
  uint32_t zs_n[CB];
  ZsU zu[CB], zu_n[CB];

  uint32_t xoff[AL];
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int cc = tid + kP32Threads * i;
    const int row = cc >> 4, chunk = cc & 15;
    const int mr = bm + row < M ? row : M - 1 - bm;
    xoff[i] = (uint32_t)((size_t)mr * ldx + chunk * 8) * 2u;
  }
  auto load_a = [&](int kb) {
#pragma unroll
    for (int i = 0; i < AL; ++i) a_st[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff[i], kb * 256, 0);
  };
  auto store_a1 = [&](int buf, int i) {
    const int cc = tid + kP32Threads * i;
    *(u32x4_t*)(As + buf * (kPfBM * 256) + pfp_off(cc >> 4, cc & 15)) = a_st[i];
  };
  auto load_b = [&](u32x4_t (&a)[CB], u32x4_t (&b)[CB], uint32_t (&zs)[CB], int kb, int grp) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      a[cb] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff_a[cb], kb * 1024, 0);
      b[cb] = __builtin_amdgcn_raw_buffer_load_b128(rw, woff_b[cb], kb * 1024, 0);
      zs[cb] = __builtin_amdgcn_raw_buffer_load_b32(rz, zoff[cb], grp * 64, 0);
    }
  };
  // x fragment of 32-row block rb for k-step s (16 deep): chunk 2 s + h of row 32 rb + c
  auto read_a = [&](const unsigned char* Xb, int rb, int s) { return *(const u32x4_t*)(Xb + pfp_off(rb * 32 + c, 2 * s + h)); };

  float16v_t acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rb][cb][e] = 0.f;

  // prologue: x tile 0 into LDS, k-block 0's weights / scales in registers, its first fragment dequantised
  load_a(0);
  load_b(wa, wb, zs_n, 0, 0);
  int grp_n = 0, cnt_n = 0;
#pragma unroll
  for (int i = AL - 1; i >= 0; --i) store_a1(0, i);
#pragma unroll
  for (int cb = 0; cb < CB; ++cb) zu[cb] = zs_unpack(zs_n[cb]);
  u32x4_t frag = rp_dequant(wa[0][0], zu[0].z1024, zu[0].z64, zu[0].s2);
  load_a(KB > 1 ? 1 : 0);                            // tile 1 travels while block 0 computes
  __syncthreads();
  u32x4_t af[RB], af_n[RB];
#pragma unroll
  for (int rb = RB - 1; rb >= 0; --rb) af[rb] = read_a(As, rb, 0);

  for (int kb = 0; kb < KB; ++kb) {
    const int nxt = kb + 1 < KB ? kb + 1 : kb;       // clamped, unconditional prefetch
    const unsigned char* Ab = As + (kb & 1) * (kPfBM * 256);
    const int nbuf = (kb + 1) & 1;
    const unsigned char* An = As + nbuf * (kPfBM * 256);
    const int nx2 = kb + 2 < KB ? kb + 2 : KB - 1;
    if (kb + 1 < KB && ++cnt_n == kpg) { cnt_n = 0; ++grp_n; }
    load_b(wa_n, wb_n, zs_n, nxt, grp_n);
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) {
        // the fragment after (s, cb): (s, cb + 1), then (s + 1, 0), then k-block kb + 1's (0, 0)
        Dq32 p;
        ZsU zn;
        if (cb < CB - 1) { p.w = (s & 1) ? wb[cb + 1][s >> 1] : wa[cb + 1][s >> 1]; zn = zu[cb + 1]; }
        else if (s < 7) { p.w = ((s + 1) & 1) ? wb[0][(s + 1) >> 1] : wa[0][(s + 1) >> 1]; zn = zu[0]; }
        else { uint32_t w0 = wa_n[0][0]; pin_here(w0); p.w = w0; zn = zu_n[0]; }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          mfma32_tied(acc[rb][cb], af[rb], frag);
          switch (rb) {                              // (compile-time after unrolling)
            case 0: dq32_stage<0>(p, zn); break;
            case 1: dq32_stage<1>(p, zn); break;
            case 2: dq32_stage<2>(p, zn); break;
            default: dq32_stage<3>(p, zn); break;
          }
          const int n = (s * CB + cb) * RB + rb;     // MFMA index within the k-block: 0 .. 32 CB - 1
          // passengers: the next k-step's x fragments (last 32-row block first: one wait per k-step), the next x tile -> the other
          // buffer (youngest register first), the request for the tile after it, the unpacking of the next block's scales
          // (the fragment reads go behind the FIRST MFMAs of a k-step: they are needed a whole 32-column block — 128 matrix-pipe
          // cycles — later; the LDS writes of the next x tile go into the other block's slots)
          if (cb == 0 && s == 7 && rb == 0) __syncthreads();          // every wave's share of the next x tile is in LDS
          if (cb == 0 && s < 7) af_n[RB - 1 - rb] = read_a(Ab, RB - 1 - rb, s + 1);
          if (cb == 0 && s == 7) af_n[RB - 1 - rb] = read_a(An, RB - 1 - rb, 0);
          if (cb == CB - 1 && s >= 1 && s <= 4 && (rb == 0 || rb == 2)) store_a1(nbuf, AL - 1 - ((s - 1) * 2 + (rb >> 1)));
          if (cb == CB - 1 && s == 5 && rb == 1) load_a(nx2);
          if (s == 6 && rb == 3) { pin_here(zs_n[cb]); zu_n[cb] = zs_unpack(zs_n[cb]); }
          (void)n;
          __builtin_amdgcn_sched_barrier(0);
        }
        frag = p.f;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) af[rb] = af_n[rb];
    }
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) { wa[cb] = wa_n[cb]; wb[cb] = wb_n[cb]; zu[cb] = zu_n[cb]; }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs' results must have left the pipe before the (compiler-scheduled) reads below

#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int m = bm + rb * 32 + 8 * (v >> 2) + 4 * h + (v & 3);
      if (m < M) {
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int gi = cg_tile + 2 * cb + gsel;
          const int n = (cg_tile + 2 * cb) * 16 + c;
          if (gi < cg_end && n < N) store_output<AWQ_DTYPE_F16>(y, (size_t)m * N + n, acc[rb][cb][v], bias, n);
        }
      }
    }
}

template <int CB>
static int p32_launch_region(const GemmArgs& a, const u32x4_t* qw_r, const uint32_t* zs_r, int NG, int cg_base, int ng_region) {
  const int nbx = (ng_region + 8 * CB - 1) / (8 * CB), nby = (a.M + kPfBM - 1) / kPfBM;
  const size_t lds = 2 * kPfBM * 256;
  static unsigned long long opted[2] = {0ull, 0ull};
  if (!opt_in_dynamic_lds((const void*)gemm_repacked_pipelined32_kernel<CB>, (int)lds, opted)) return AWQ_ERR_LAUNCH;
  hipLaunchKernelGGL(gemm_repacked_pipelined32_kernel<CB>, dim3(nbx * nby), dim3(kP32Threads), lds, a.stream, (const uint16_t*)a.x, a.ldx,
                     qw_r, zs_r, a.bias, a.y, a.M, a.K, a.N, a.g, NG, nbx, nby, cg_base, ng_region);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

// 32-bit byte offsets into the whole packed tensor and the scale words
bool pipelined32_addressable(const GemmArgs& a) {
  const int64_t NG = rp_groups(a.N);
  return pipelined_addressable(a) && NG * (a.K / 128) * 1024 < (int64_t(1) << 31);
}

// Column groups [0, gA) in 128 x 256 tiles, the rest in 128 x 128 tiles.
int launch_gemm_repacked_pipelined32(const GemmArgs& a, const void* packed, int gA) {
  if (!pipelined32_addressable(a)) return AWQ_ERR_BAD_VARIANT;
  const int NG = rp_groups(a.N);
  const u32x4_t* qw_r = (const u32x4_t*)packed;
  const uint32_t* zs_r = (const uint32_t*)packed + (size_t)NG * (a.K / 128) * 256;
  if (gA > NG) gA = NG;
  if (gA > 0) {
    const int rc = p32_launch_region<2>(a, qw_r, zs_r, NG, 0, gA);
    if (rc) return rc;
  }
  if (gA < NG) return p32_launch_region<1>(a, qw_r, zs_r, NG, gA, NG - gA);
  return AWQ_OK;
}

}  // namespace awq
