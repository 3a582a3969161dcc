// chain_lab — laboratory probe (not product code): a CHAIN of dependent decode GEMVs whose consecutive launches overlap.
//
// tools/kbench (KBENCH_STREAMS=2) showed that independent launches of the decode GEMV issued on two streams run at 4.75 us instead of 6.2 us
// each: the next kernel's launch latency, argument fetch and first weight loads hide behind the previous kernel's tail.  The linears of a
// decode step depend on each other, so here each kernel
//   1. issues its first weight loads and stages its (scale | zero) words — none of that depends on the previous kernel;
//   2. waits (one lane polls, bounded; everyone else sits at a workgroup barrier) until the previous kernel's workgroups have all announced
//      their outputs on a device-side counter;
//   3. stages its activations with system-scope loads (another XCD's L2 must not serve last step's bytes), computes, reduces;
//   4. stores its outputs write-through (system scope), waits for them, and adds 1 to its own counter.
// Kernel i runs on stream i % 2; a kernel two launches back is on the same stream, so at most two kernels are ever resident and all of
// their workgroups fit the chip together (the one-row kernels need <= 64 VGPRs and <= 32 KiB of LDS: two workgroups per CU).
// Shapes alternate 4096 -> 12288 (strips of 3 column groups, 2 k-blocks per wave) and 12288 -> 4096 (1 column group, 6 k-blocks per wave):
// 256 workgroups each.  Every spin is bounded; a give-up raises an error flag and is reported.
//
//   tools/chain_lab [launches=64] [reps=50] [s_sleep(127) repetitions between polls=1]      (-DCHAIN_AGENT: sc1 instead of sc0 sc1)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/awq_hip.h"
#include "../sglang_awq_amd/csrc/awq_device.h"

using namespace awq;

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

struct ChainArgs {
  const uint16_t* x;          // activations = the previous kernel's y
  const u32x4_t* qw_r;
  const uint32_t* zs_r;
  uint16_t* y;
  int N, NG, KB;
  const unsigned* wait_cnt;   // previous kernel's arrival counter (nullptr: no wait)
  unsigned wait_val;
  unsigned* my_cnt;           // this kernel's arrival counter
  unsigned* zero_cnt;         // a counter far ahead in the ring, reset by workgroup 0
  unsigned* err;              // set when a poll gives up
  int sleeps;                 // s_sleep(127) repetitions between polls (~0.85 us each at 2.4 GHz... 127 x 64 cycles)
  int sys;                    // 1: system-scope activations / outputs + flag protocol; 0: plain loads / stores, no flags (kernel-boundary ordering)
};

__device__ __forceinline__ u32x4_t load_sys_b128(const void* p) {
  u32x4_t v;
#ifdef CHAIN_AGENT
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
#else
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
#endif
  return v;
}
__device__ __forceinline__ unsigned load_sys_b32(const void* p) {
  unsigned v;
#ifdef CHAIN_AGENT
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
#else
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
#endif
  return v;
}
__device__ __forceinline__ void store_sys_b16(void* p, uint32_t v) {
#ifdef CHAIN_AGENT
  asm volatile("global_store_short %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
#else
  asm volatile("global_store_short %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
#endif
}

// the one-row straight-line GEMV of awq_repacked_gemv.h (gemv_rp2_kernel<G, T, ..., M1>), re-ordered as described above
template <int G, int T>
__global__ __launch_bounds__(1024) void chain_gemv_kernel(ChainArgs a) {
  constexpr int W = 16, L = G * T, DD = 2, RB = 3;
  constexpr int XS = T * 128 + 8;
  constexpr int STG = XS * 2 + G * T * 64;
  extern __shared__ __attribute__((aligned(16))) float red[];
  __shared__ unsigned go;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, r = lane & 15;
  int cg0 = blockIdx.x * G;
  if (cg0 + G > a.NG) cg0 = a.NG - G;
  const int KB = a.KB, kb0 = wave * T;
  unsigned char* const stg = (unsigned char*)(red + (size_t)W * 16 * G) + (size_t)wave * STG;
  if (a.sys && blockIdx.x == 0 && threadIdx.x == 0 && a.zero_cnt) __hip_atomic_store(a.zero_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // 1. what does not depend on the previous kernel: zs words, first weight loads
  constexpr int NZ = G * T * 4;
  static_assert(NZ <= 64, "one load of zs chunks");
  const int zc = lane < NZ ? lane : NZ - 1;
  const int z_c = zc / (T * 4), z_j = zc % (T * 4);
  const u32x4_t zv = *(const u32x4_t*)((const unsigned char*)(a.zs_r + ((size_t)(cg0 + z_c) * KB + kb0) * 16) + z_j * 16);
  const unsigned char* wbase = (const unsigned char*)(a.qw_r + ((size_t)cg0 * KB + kb0) * 64);
  const uint32_t loff = (uint32_t)lane * 16u;
  u32x4_t wbuf[RB];
  auto load_w = [&](int i) {
    const int c = i / T, t = i % T;
    wbuf[i % RB] = __builtin_nontemporal_load((const u32x4_t*)(wbase + ((size_t)c * KB + t) * 1024 + loff));
  };
#pragma unroll
  for (int i = 0; i < DD; ++i) {
    load_w(i);
    __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
  }
  if (lane < NZ) *(u32x4_t*)(stg + XS * 2 + (z_c * T * 4 + z_j) * 16) = zv;

  // 2. wait for the previous kernel's outputs
  if (a.sys && a.wait_cnt) {
    if (threadIdx.x == 0) {
      unsigned ok = 0;
      for (int spin = 0; spin < (1 << 16); ++spin) {
        if (load_sys_b32(a.wait_cnt) >= a.wait_val) { ok = 1; break; }
        for (int z = 0; z < a.sleeps; ++z) __builtin_amdgcn_s_sleep(127);
      }
      if (!ok) atomicAdd(a.err, 1u);
      go = ok;
    }
    __syncthreads();
  }

  // 3. activations: this wave's T k-blocks of the one row
  constexpr int NXC = T * 16, CHX = (NXC + 63) / 64;
#pragma unroll
  for (int i = 0; i < CHX; ++i) {
    const int id = lane + 64 * i;
    if (id < NXC) {
      const unsigned char* sx = (const unsigned char*)(a.x + (size_t)kb0 * 128) + id * 16;
      const u32x4_t v = a.sys ? load_sys_b128(sx) : *(const u32x4_t*)sx;
      *(u32x4_t*)(stg + id * 16) = v;
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  if (wave >= 12) __builtin_amdgcn_s_setprio(2);
  else if (wave >= 8) __builtin_amdgcn_s_setprio(1);
  uint32_t mlo = kLoNib, mhi = kHiNib, magic = kMagicF16;
  asm volatile("" : "+s"(mlo), "+s"(mhi));
  asm volatile("" : "+v"(magic));
  float4_t acc[G];
#pragma unroll
  for (int c = 0; c < G; ++c) acc[c] = (float4_t){0.f, 0.f, 0.f, 0.f};
  const half2_t c960 = {(half_t)960.f, (half_t)960.f};
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const half_t* x_lds = (const half_t*)stg;
  const uint32_t* zs_lds = (const uint32_t*)(stg + XS * 2) + r;
  u32x4_t xa[4];
#pragma unroll
  for (int i = 0; i < L; ++i) {
    const int c = i / T, t = i % T;
    if (T > 1 || i == 0) {
#pragma unroll
      for (int d = 0; d < 4; ++d) xa[d] = *(const u32x4_t*)(x_lds + t * 128 + d * 32 + q * 8);
    }
    const half2_t zh = as_h2(zs_lds[(c * T + t) * 16]);
    const half2_t s2 = __builtin_shufflevector(zh, zh, 0, 0);
    const half2_t z1024 = __builtin_shufflevector(zh, zh, 1, 1);
    const half2_t z64 = z1024 - c960;
    u32x4_t w = wbuf[i % RB];
    asm volatile("" : "+v"(w));
    __builtin_amdgcn_sched_barrier(0);
    if (i + DD < L) load_w(i + DD);
    __builtin_amdgcn_sched_barrier(0);
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
  }
  __builtin_amdgcn_s_setprio(0);
  if (q == 0) {
#pragma unroll
    for (int c = 0; c < G; ++c) red[wave * (16 * G) + c * 16 + r] = acc[c][0];
  }
  __syncthreads();
  // 4. outputs (wave 0 holds all 16 G columns), then the announcement
  if (wave == 0) {
    if (lane < 16 * G) {
      float v = red[lane];
#pragma unroll
      for (int w2 = 1; w2 < W; ++w2) v += red[w2 * (16 * G) + lane];
      const int n = cg0 * 16 + lane;
      if (n < a.N) {
        const uint32_t hb = float_to_half_bits(v);
        if (a.sys) store_sys_b16(a.y + n, hb); else a.y[n] = (uint16_t)hb;
      }
    }
    if (a.sys) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(a.my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

struct Shape { int K, N, G, T, NG, KB, nwg; size_t lds; void* packed; };

static void launch(const Shape& s, const ChainArgs& a, hipStream_t st) {
  if (s.G == 3) hipLaunchKernelGGL((chain_gemv_kernel<3, 2>), dim3(s.nwg), dim3(1024), s.lds, st, a);
  else hipLaunchKernelGGL((chain_gemv_kernel<1, 6>), dim3(s.nwg), dim3(1024), s.lds, st, a);
}

int main(int argc, char** argv) {
  const int LCH = argc > 1 ? atoi(argv[1]) : 64, reps = argc > 2 ? atoi(argv[2]) : 50;
  const int sleeps = argc > 3 ? atoi(argv[3]) : 1;
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
  // 16 rotating weight sets per shape (more than 2 x the Infinity Cache in all)
  const int SETS = 16;
  Shape sh[2] = {{4096, 12288, 3, 2}, {12288, 4096, 1, 6}};
  std::vector<void*> packed[2];
  for (int k = 0; k < 2; ++k) {
    Shape& s = sh[k];
    s.NG = s.N / 16; s.KB = s.K / 128; s.nwg = (s.NG + s.G - 1) / s.G;
    s.lds = (size_t)16 * 16 * s.G * 4 + (size_t)16 * ((s.T * 128 + 8) * 2 + s.G * s.T * 64);
    const size_t pbytes = awq_repacked_bytes(s.K, s.N, 128, AWQ_DTYPE_F16);
    int32_t *qw, *qz; void* sc;
    CK(hipMalloc(&qw, (size_t)s.K * s.N / 2)); CK(hipMalloc(&qz, (size_t)(s.K / 128) * s.N / 2)); CK(hipMalloc(&sc, (size_t)(s.K / 128) * s.N * 2));
    std::vector<uint32_t> h((size_t)s.K * s.N / 8);
    std::vector<uint16_t> hs((size_t)(s.K / 128) * s.N);
    const float scale = s.K == 4096 ? 0.0024f : 0.00139f;      // 1 / (sqrt(K) * std(q - z) = 6.5): keeps |y| ~ |x| along the chain
    for (auto& v : hs) { _Float16 f = (_Float16)(scale * (0.8f + 0.4f * (float)((rnd() >> 40) * (1.0 / (1 << 24))))); memcpy(&v, &f, 2); }
    CK(hipMemcpy(sc, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < SETS; ++i) {
      for (auto& v : h) v = (uint32_t)(rnd() >> 16);
      CK(hipMemcpy(qw, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(qz, h.data(), (size_t)(s.K / 128) * s.N / 2, hipMemcpyHostToDevice));
      void* p; CK(hipMalloc(&p, pbytes));
      if (awq_repack(qw, sc, qz, p, s.K, s.N, 128, AWQ_DTYPE_F16, sa)) { fprintf(stderr, "repack failed\n"); return 1; }
      CK(hipStreamSynchronize(sa));
      packed[k].push_back(p);
    }
    CK(hipFree(qw)); CK(hipFree(qz)); CK(hipFree(sc));
  }
  // activations: buffer i holds the output of launch i (12288 halves is enough for either shape); buffer LCH = the chain's input
  std::vector<uint16_t*> act(LCH + 1);
  for (auto& p : act) { CK(hipMalloc(&p, 12288 * 2)); CK(hipMemset(p, 0, 12288 * 2)); }
  {
    std::vector<uint16_t> hx(4096);
    for (auto& v : hx) { _Float16 f = (_Float16)(-1.f + 2.f * (float)((rnd() >> 40) * (1.0 / (1 << 24)))); memcpy(&v, &f, 2); }
    CK(hipMemcpy(act[LCH], hx.data(), 4096 * 2, hipMemcpyHostToDevice));
  }
  constexpr int RING = 16;
  unsigned *cnt, *err;
  CK(hipMalloc(&cnt, RING * 64 * sizeof(unsigned))); CK(hipMemset(cnt, 0, RING * 64 * sizeof(unsigned)));     // one counter per 256-byte line
  CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipEvent_t e0, e1, ej;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ej));

  auto run_chain = [&](int mode) {      // 0: one stream, kernel-boundary ordering (plain loads / stores); 1: one stream + flag protocol; 2: two streams + flag protocol
    for (int i = 0; i < LCH; ++i) {
      const Shape& s = sh[i & 1];
      ChainArgs a;
      a.x = i == 0 ? act[LCH] : act[i - 1];
      a.qw_r = (const u32x4_t*)packed[i & 1][(i / 2) % SETS];
      a.zs_r = (const uint32_t*)a.qw_r + (size_t)s.NG * s.KB * 256;
      a.y = act[i]; a.N = s.N; a.NG = s.NG; a.KB = s.KB;
      a.sys = mode != 0;
      a.wait_cnt = (mode != 0 && i > 0) ? cnt + (size_t)((i - 1) % RING) * 64 : nullptr;
      a.wait_val = (unsigned)sh[(i - 1) & 1].nwg;
      a.my_cnt = cnt + (size_t)(i % RING) * 64;
      a.zero_cnt = cnt + (size_t)((i + 8) % RING) * 64;
      a.err = err;
      a.sleeps = sleeps;
      launch(s, a, (mode == 2 && (i & 1)) ? sb : sa);
    }
  };
  auto time_mode = [&](int mode) {
    CK(hipMemset(cnt, 0, RING * 64 * sizeof(unsigned)));
    CK(hipDeviceSynchronize());
    for (int w = 0; w < 3; ++w) {
      run_chain(mode);
      CK(hipDeviceSynchronize());
      CK(hipMemset(cnt, 0, RING * 64 * sizeof(unsigned)));
    }
    CK(hipDeviceSynchronize());
    float total = 0;
    for (int rpt = 0; rpt < reps; ++rpt) {
      CK(hipEventRecord(e0, sa));
      if (mode == 2) CK(hipStreamWaitEvent(sb, e0, 0));
      run_chain(mode);
      if (mode == 2) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
      CK(hipEventRecord(e1, sa));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
      CK(hipMemset(cnt, 0, RING * 64 * sizeof(unsigned)));     // every chain starts from zeroed counters (the in-kernel reset covers the ring within a chain)
      CK(hipDeviceSynchronize());
    }
    return total * 1e3 / (reps * LCH);
  };
  auto snapshot = [&]() {
    std::vector<uint16_t> all;
    for (int i = 0; i < LCH; ++i) {
      std::vector<uint16_t> h(sh[i & 1].N);
      CK(hipMemcpy(h.data(), act[i], h.size() * 2, hipMemcpyDeviceToHost));
      all.insert(all.end(), h.begin(), h.end());
    }
    return all;
  };
  const char* names[3] = {"one stream, kernel-boundary ordering (plain loads / stores)", "one stream + flag protocol (system-scope x / y)",
                          "two streams + flag protocol (consecutive kernels overlap)"};
  std::vector<uint16_t> ref;
  for (int mode = 0; mode < 3; ++mode) {
    const double us = time_mode(mode);
    run_chain(mode);
    if (mode == 2) { CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0)); }
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> got = snapshot();
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nonfinite = 0, zeros = 0;
    if (mode == 0) ref = got;
    for (size_t i = 0; i < got.size(); ++i) { diff += got[i] != ref[i]; nonfinite += (got[i] & 0x7c00) == 0x7c00; zeros += (got[i] & 0x7fff) == 0; }
    const double bytes = ((double)4096 * 12288 / 2 * 1.0625 + (double)12288 * 4096 / 2 * 1.0625) / 2;
    printf("%-64s %8.3f us per launch  (%6.1f GB/s of weights)  outputs differing from mode 0: %zu of %zu  non-finite %zu zeros %zu  poll give-ups %u\n", names[mode], us,
           bytes / us / 1e3, diff, got.size(), nonfinite, zeros, herr);
    CK(hipMemset(cnt, 0, RING * 64 * sizeof(unsigned)));
    CK(hipDeviceSynchronize());
  }
  return 0;
}
