// kbench — standalone timing harness for libawq_hip.so on a real MI355X (no Python in the loop).
//
//   tools/kbench gemm   M K N g dtype variant tune[,tune...] [sets=16] [iters=400] [graph=1]
//   tools/kbench dequant  K N g dtype [sets] [iters]
//   tools/kbench read   bytes_mb pattern [iters]      pure-read ceilings for the GEMV access pattern
//
// Weights rotate through `sets` distinct buffers so a 23 MB matrix is streamed from HBM rather than
// from the 256 MiB Infinity Cache (sets=1 gives the cache-hot number).  Times are HIP events around
// `iters` back-to-back launches on one stream (graph=1: the launches are replayed from a hipGraph).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "../include/awq_aux.h"
#include "../include/awq_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return rng_state;
}

static void fill_u32(void* dptr, size_t n_words) {
  std::vector<uint32_t> h(n_words);
  for (size_t i = 0; i < n_words; ++i) h[i] = (uint32_t)(rnd() >> 16);
  CK(hipMemcpy(dptr, h.data(), n_words * 4, hipMemcpyHostToDevice));
}

static uint16_t f2h(float f) {   // crude float -> half for small positive scales
  _Float16 h = (_Float16)f;
  uint16_t u; memcpy(&u, &h, 2); return u;
}
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

static void fill_scales(void* dptr, size_t n, int dtype, float lo, float hi) {
  if (dtype == AWQ_DTYPE_F32) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / (1 << 24)));
    CK(hipMemcpy(dptr, h.data(), n * 4, hipMemcpyHostToDevice));
  } else {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) {
      float v = lo + (hi - lo) * (float)((rnd() >> 40) * (1.0 / (1 << 24)));
      h[i] = dtype == AWQ_DTYPE_F16 ? f2h(v) : f2bf(v);
    }
    CK(hipMemcpy(dptr, h.data(), n * 2, hipMemcpyHostToDevice));
  }
}

// ------------------------------------------------------------------------------------------------
// pure-read kernels: how fast can this chip stream 23 MB in the shapes the GEMV uses?
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// pattern 0: fully linear, each lane 16 B, grid-stride
__global__ void read_linear(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    u32x4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// pattern 1: the skinny-GEMV pattern — [rows][row_words] matrix; a wave covers 32 rows x 256 B
// (lane (q, r): rows 8q+j, 16-byte chunk r), KT k-steps per wave, all loads issued up front.
template <int KT>
__global__ __launch_bounds__(256) void read_gemv_pattern(const uint32_t* __restrict__ qw, int K, int C, int n_ct, uint32_t* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, r = lane & 15;
  const int ct = blockIdx.x % n_ct, ks = blockIdx.x / n_ct;
  const int chunk4 = (ct * 16 + r) * 4;
  const int kbase = ((ks * 4 + wave) * KT) * 32;
  uint32_t acc = 0;
  if (chunk4 < C && kbase < K) {
    u32x4 R[KT][8];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) R[t][j] = *(const u32x4*)(qw + (size_t)(kbase + t * 32 + 8 * q + j) * C + chunk4);
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc ^= R[t][j].x ^ R[t][j].y ^ R[t][j].z ^ R[t][j].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// pattern 2: column strips — one workgroup owns 16 bytes (32 logical columns) of EVERY row
// (no split-K); lane = row.  Eight neighbouring strips share each 128-byte line (L2 reuse).
__global__ __launch_bounds__(1024) void read_strip_pattern(const uint32_t* __restrict__ qw, int K, int C, uint32_t* sink) {
  const int strip = blockIdx.x;   // 16-byte chunk index
  uint32_t acc = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    u32x4 v = *(const u32x4*)(qw + (size_t)k * C + strip * 4);
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// pattern 3: column strips with an XCD-aware strip assignment: blocks b and b+8 share an XCD (observed
// round-robin dispatch), so strips that share 128-byte lines are given to blocks of equal b % 8.
// W16 = 16-byte chunks per row owned by the workgroup (1 -> 32 columns, 2 -> 64 columns).
template <int W16>
__global__ __launch_bounds__(1024) void read_strip_xcd(const uint32_t* __restrict__ qw, int K, int C, int nstrips, uint32_t* sink) {
  const int b = blockIdx.x;
  const int per = (nstrips + 7) / 8;
  const int strip = (b % 8) * per + b / 8;
  uint32_t acc = 0;
  if (strip < nstrips && (b / 8) < per) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
#pragma unroll
      for (int u = 0; u < W16; ++u) {
        u32x4 v = *(const u32x4*)(qw + (size_t)k * C + (strip * W16 + u) * 4);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  if (acc == 0x12345678u) *sink = acc;
}

static int cmd_read(int argc, char** argv) {
  const int K = 4096, C = 1376;
  const int sets = argc > 2 ? atoi(argv[2]) : 16;
  const int iters = argc > 3 ? atoi(argv[3]) : 400;
  const size_t words = (size_t)K * C;
  std::vector<uint32_t*> bufs(sets);
  for (auto& b : bufs) { CK(hipMalloc(&b, words * 4)); fill_u32(b, words); }
  uint32_t* sink; CK(hipMalloc(&sink, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_it = [&](const char* name, auto launch) {
    for (int i = 0; i < 20; ++i) launch(bufs[i % sets]);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(bufs[i % sets]);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("read %-34s sets=%2d  %8.3f us/launch  %8.1f GB/s\n", name, sets, us, words * 4 / us / 1e3);
  };
  const int n_ct = (C / 4 + 15) / 16;
  for (int grid : {512, 1024, 2048, 4096})
    time_it(("linear grid=" + std::to_string(grid) + "x256").c_str(),
            [&](uint32_t* b) { hipLaunchKernelGGL(read_linear, dim3(grid), dim3(256), 0, st, (const u32x4*)b, words / 4, sink); });
  time_it("gemv-pattern KT=1 (704 wg)", [&](uint32_t* b) { hipLaunchKernelGGL(read_gemv_pattern<1>, dim3(n_ct * 32), dim3(256), 0, st, b, K, C, n_ct, sink); });
  time_it("gemv-pattern KT=2 (352 wg)", [&](uint32_t* b) { hipLaunchKernelGGL(read_gemv_pattern<2>, dim3(n_ct * 16), dim3(256), 0, st, b, K, C, n_ct, sink); });
  time_it("gemv-pattern KT=4 (176 wg)", [&](uint32_t* b) { hipLaunchKernelGGL(read_gemv_pattern<4>, dim3(n_ct * 8), dim3(256), 0, st, b, K, C, n_ct, sink); });
  time_it("strip-pattern 344 wg x 1024", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_pattern, dim3(C / 4), dim3(1024), 0, st, b, K, C, sink); });
  time_it("strip-xcd 16B 344 wg x 1024", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_xcd<1>, dim3(344), dim3(1024), 0, st, b, K, C, 344, sink); });
  time_it("strip-xcd 16B 344 wg x 512", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_xcd<1>, dim3(344), dim3(512), 0, st, b, K, C, 344, sink); });
  time_it("strip-xcd 32B 176 wg x 1024", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_xcd<2>, dim3(176), dim3(1024), 0, st, b, K, C, 172, sink); });
  time_it("strip-xcd 64B 88 wg x 1024", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_xcd<4>, dim3(88), dim3(1024), 0, st, b, K, C, 86, sink); });
  time_it("strip-pattern 344 wg x 512", [&](uint32_t* b) { hipLaunchKernelGGL(read_strip_pattern, dim3(C / 4), dim3(512), 0, st, b, K, C, sink); });
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct Weights { int32_t* qw; void* sc; int32_t* qz; };

static std::vector<Weights> make_weights(int sets, int K, int N, int g, int dtype) {
  std::vector<Weights> w(sets);
  const size_t eb = dtype == AWQ_DTYPE_F32 ? 4 : 2;
  for (auto& x : w) {
    CK(hipMalloc(&x.qw, (size_t)K * N / 8 * 4));
    CK(hipMalloc(&x.qz, (size_t)(K / g) * N / 8 * 4));
    CK(hipMalloc(&x.sc, (size_t)(K / g) * N * eb));
    fill_u32(x.qw, (size_t)K * N / 8);
    fill_u32(x.qz, (size_t)(K / g) * N / 8);
    fill_scales(x.sc, (size_t)(K / g) * N, dtype, 0.005f, 0.02f);
  }
  return w;
}

static int cmd_gemm(int argc, char** argv) {
  if (argc < 9) { fprintf(stderr, "usage: kbench gemm M K N g dtype variant tune[,tune..] [sets] [iters] [graph]\n"); return 2; }
  const int M = atoi(argv[2]), K = atoi(argv[3]), N = atoi(argv[4]), g = atoi(argv[5]), dtype = atoi(argv[6]), variant = atoi(argv[7]);
  std::vector<long long> tunes;
  for (char* tok = strtok(argv[8], ","); tok; tok = strtok(nullptr, ",")) tunes.push_back(strtoll(tok, nullptr, 0));
  const int sets = argc > 9 ? atoi(argv[9]) : 16;
  const int iters = argc > 10 ? atoi(argv[10]) : 400;
  const int use_graph = argc > 11 ? atoi(argv[11]) : 1;
  const size_t eb = dtype == AWQ_DTYPE_F32 ? 4 : 2;
  auto w = make_weights(sets, K, N, g, dtype);
  void *x, *y, *ws;
  CK(hipMalloc(&x, (size_t)M * K * eb)); fill_scales(x, (size_t)M * K, dtype, -1.f, 1.f);
  CK(hipMalloc(&y, (size_t)M * N * eb));
  const size_t ws_bytes = 4096 + (64u << 20);
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)K * N / 2 + (double)(K / g) * N / 2 + (double)(K / g) * N * eb + (double)M * K * eb + (double)M * N * eb;
  const double flops = 2.0 * M * K * N;
  for (long long tune : tunes) {
    auto launch = [&](int i) {
      int rc = awq_gemm_ex(x, K, w[i % sets].qw, w[i % sets].sc, w[i % sets].qz, nullptr, y, ws, ws_bytes, M, K, N, g, dtype, 1, variant, tune, st);
      if (rc) { fprintf(stderr, "awq_gemm_ex: %s\n", awq_hip_status_string(rc)); exit(1); }
    };
    for (int i = 0; i < 2 * sets; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    float ms = 0;
    if (use_graph) {
      hipGraph_t graph; hipGraphExec_t exec;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < sets; ++i) launch(i);
      CK(hipStreamEndCapture(st, &graph));
      CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      const int reps = (iters + sets - 1) / sets;
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= (float)(reps * sets);
      CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    } else {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) launch(i);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= (float)iters;
    }
    const double us = ms * 1e3;
    printf("gemm M=%d K=%d N=%d g=%d dt=%d variant=%d tune=0x%llx sets=%d graph=%d : %9.3f us  %8.1f GB/s  %8.2f TFLOP/s\n", M, K, N, g,
           dtype, variant, tune, sets, use_graph, us, bytes / us / 1e3, flops / us / 1e6);
    fflush(stdout);
  }
  return 0;
}

// one stamped launch (diagnostic build of the skinny kernel): per-workgroup phase times
static int cmd_stamps(int argc, char** argv) {
  if (argc < 8) { fprintf(stderr, "usage: kbench stamps M K N g dtype tune [sets]\n"); return 2; }
  const int M = atoi(argv[2]), K = atoi(argv[3]), N = atoi(argv[4]), g = atoi(argv[5]), dtype = atoi(argv[6]);
  long long tune = strtoll(argv[7], nullptr, 0);
  if (((tune >> 16) & 7) == 0) tune |= 0x30000;
  const int sets = argc > 8 ? atoi(argv[8]) : 16;
  auto w = make_weights(sets, K, N, g, dtype);
  void *x, *y, *ws;
  CK(hipMalloc(&x, (size_t)M * K * 2)); fill_scales(x, (size_t)M * K, dtype, -1.f, 1.f);
  CK(hipMalloc(&y, (size_t)M * N * 2));
  const size_t ws_bytes = 4096 + (64u << 20);
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int max_wg = 4096;
  std::vector<unsigned long long> h((size_t)max_wg * 8);
  const int reps = argc > 9 ? atoi(argv[9]) : 2 * sets + 3;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipMemsetAsync((char*)ws + (48u << 20), 0, (size_t)max_wg * 64, st));
    int rc = awq_gemm_ex(x, K, w[rep % sets].qw, w[rep % sets].sc, w[rep % sets].qz, nullptr, y, ws, ws_bytes, M, K, N, g, dtype, 1, AWQ_GEMM_SKINNY, tune, st);
    if (rc) { fprintf(stderr, "awq_gemm_ex: %s\n", awq_hip_status_string(rc)); return 1; }
    CK(hipStreamSynchronize(st));
  }
  CK(hipMemcpy(h.data(), (char*)ws + (48u << 20), (size_t)max_wg * 64, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; int nwg = 0;
  for (int b = 0; b < max_wg; ++b) if (h[(size_t)b * 8]) { nwg = b + 1; if (h[(size_t)b * 8] < t0) t0 = h[(size_t)b * 8]; }
  const char* names[8] = {"start", "loads issued", "loads landed", "compute done", "after barrier", "partial stored", "ticket drawn", "reduced (last arriver)"};
  printf("stamps: %d workgroups, times in us after the first workgroup started (100 MHz clock)\n", nwg);
  {   // slot 2 holds shader-clock ticks from kernel start to the end of the k-loop, slot 3 - slot 0 the same span in 10 ns units
    std::vector<double> mhz;
    for (int b = 0; b < nwg; ++b) if (h[(size_t)b * 8 + 2] && h[(size_t)b * 8 + 3] > h[(size_t)b * 8]) mhz.push_back((double)h[(size_t)b * 8 + 2] / ((double)(h[(size_t)b * 8 + 3] - h[(size_t)b * 8]) * 0.01));
    if (!mhz.empty()) { std::sort(mhz.begin(), mhz.end()); printf("  shader clock during the k-loop: median %.0f MHz (min %.0f, max %.0f)\n", mhz[mhz.size() / 2], mhz.front(), mhz.back()); }
  }
  for (int slot = 0; slot < 8; ++slot) {
    if (slot == 2) continue;
    std::vector<double> v;
    for (int b = 0; b < nwg; ++b) if (h[(size_t)b * 8 + slot]) v.push_back((double)(h[(size_t)b * 8 + slot] - t0) * 0.01);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-24s n=%4zu  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", names[slot], v.size(), v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  return 0;
}

// decode GEMV on the repacked (MFMA-fragment-major) layout
static int cmd_rgemm(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: kbench rgemm M K N g [sets] [iters] [graph] [fuse: 1 norm prologue, 2 silu-mul epilogue, 3 both]\n"); return 2; }
  const int M = atoi(argv[2]), K = atoi(argv[3]), N = atoi(argv[4]), g = atoi(argv[5]);
  const int sets = argc > 6 ? atoi(argv[6]) : 16;
  const int iters = argc > 7 ? atoi(argv[7]) : 400;
  const int use_graph = argc > 8 ? atoi(argv[8]) : 1;
  const int fuse = argc > 9 ? atoi(argv[9]) : 0;
  const size_t pbytes = awq_repacked_bytes(K, N, g, AWQ_DTYPE_F16);
  if (!pbytes) { fprintf(stderr, "shape not supported by the repacked path\n"); return 1; }
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<void*> packed(sets);
  {
    auto w = make_weights(1, K, N, g, AWQ_DTYPE_F16);
    for (int i = 0; i < sets; ++i) {
      CK(hipMalloc(&packed[i], pbytes));
      fill_u32(w[0].qw, (size_t)K * N / 8);       // fresh random nibbles per set
      int rc = awq_repack(w[0].qw, w[0].sc, w[0].qz, packed[i], K, N, g, AWQ_DTYPE_F16, st);
      if (rc) { fprintf(stderr, "awq_repack: %s\n", awq_hip_status_string(rc)); return 1; }
      CK(hipStreamSynchronize(st));
    }
  }
  void *x, *y;
  CK(hipMalloc(&x, (size_t)M * K * 2)); fill_scales(x, (size_t)M * K, AWQ_DTYPE_F16, -1.f, 1.f);
  CK(hipMalloc(&y, (size_t)M * N * 2));
  void *delta, *nw, *hout;
  CK(hipMalloc(&delta, (size_t)M * K * 2)); fill_scales(delta, (size_t)M * K, AWQ_DTYPE_F16, -1.f, 1.f);
  CK(hipMalloc(&nw, (size_t)K * 2)); fill_scales(nw, (size_t)K, AWQ_DTYPE_F16, 0.5f, 1.5f);
  CK(hipMalloc(&hout, (size_t)M * K * 2));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // optional scratch (the split-K route for 9..32 rows on narrow matrices): on unless KBENCH_WS=0
  void* ws = nullptr;
  const size_t ws_bytes = (getenv("KBENCH_WS") && atoi(getenv("KBENCH_WS")) == 0) ? 0 : awq_gemm_repacked_workspace_bytes(M, K, N, g, AWQ_DTYPE_F16);
  if (ws_bytes) { CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes)); }
  // KBENCH_STREAMS=2 (probe): the launches of a graph alternate between two streams with no dependency between them — how much of the
  // per-launch boundary + ramp disappears when the next kernel may start before the previous one has drained
  const int nstreams = getenv("KBENCH_STREAMS") ? atoi(getenv("KBENCH_STREAMS")) : 1;
  hipStream_t st2 = st; void* y2 = y;
  if (nstreams > 1) { CK(hipStreamCreate(&st2)); CK(hipMalloc(&y2, (size_t)M * N * 2)); }
  hipStream_t cur = st; void* ycur = y;
  auto launch = [&](int i) {
    if (nstreams > 1) { cur = (i & 1) ? st2 : st; ycur = (i & 1) ? y2 : y; }
    int rc = fuse ? awq_aux_gemv_repacked_fused(x, K, packed[i % sets], y, M, K, N, g, AWQ_DTYPE_F16, (fuse & 1) ? x : nullptr, delta, nw, hout,
                                                1e-5f, (fuse & 2) ? 1 : 0, cur)
                  : awq_gemm_repacked_ws(x, K, packed[i % sets], nullptr, ycur, ws, ws_bytes, M, K, N, g, AWQ_DTYPE_F16, cur);
    if (rc) { fprintf(stderr, "awq_gemm_repacked: %s\n", awq_hip_status_string(rc)); exit(1); }
  };
  for (int i = 0; i < 2 * sets; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  float ms = 0;
  if (use_graph) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    // launches per graph: at least 16 whatever `sets` is (a 1-launch graph times the replay overhead, not the kernel)
    const int glen = getenv("KBENCH_GLEN") ? atoi(getenv("KBENCH_GLEN")) : (sets < 16 ? 16 : sets);
    hipEvent_t fork, join;
    if (nstreams > 1) {
      CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
      CK(hipEventRecord(fork, st)); CK(hipStreamWaitEvent(st2, fork, 0));
    }
    for (int i = 0; i < glen; ++i) launch(i);
    if (nstreams > 1) { CK(hipEventRecord(join, st2)); CK(hipStreamWaitEvent(st, join, 0)); }
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    const int reps = (iters + glen - 1) / glen;
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)(reps * glen);
  } else {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= (float)iters;
  }
  const double us = ms * 1e3;
  const double bytes = (double)K * N / 2 + (double)(K / g) * N / 2 + (double)(K / g) * N * 2 + (double)M * K * 2 + (double)M * N * 2;
  printf("rgemm M=%d K=%d N=%d g=%d sets=%d graph=%d fuse=%d : %9.3f us  %8.1f GB/s (algorithmic)  %8.2f TFLOP/s\n", M, K, N, g, sets, use_graph, fuse, us,
         bytes / us / 1e3, 2.0 * M * K * N / us / 1e6);
  return 0;
}

extern "C" void awq_debug_set_stamp_buffer(void* p);   // diagnostic hook of libawq_hip.so (not in the public header)

static int cmd_rstamps(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: kbench rstamps M K N g [sets] [fuse]\n"); return 2; }
  const int M = atoi(argv[2]), K = atoi(argv[3]), N = atoi(argv[4]), g = atoi(argv[5]);
  const int sets = argc > 6 ? atoi(argv[6]) : 16;
  const int fuse = argc > 7 ? atoi(argv[7]) : 0;
  const size_t pbytes = awq_repacked_bytes(K, N, g, AWQ_DTYPE_F16);
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<void*> packed(sets);
  auto w = make_weights(1, K, N, g, AWQ_DTYPE_F16);
  for (int i = 0; i < sets; ++i) {
    CK(hipMalloc(&packed[i], pbytes));
    fill_u32(w[0].qw, (size_t)K * N / 8);
    if (awq_repack(w[0].qw, w[0].sc, w[0].qz, packed[i], K, N, g, AWQ_DTYPE_F16, st)) return 1;
    CK(hipStreamSynchronize(st));
  }
  void *x, *y; unsigned long long* dbg;
  CK(hipMalloc(&x, (size_t)M * K * 2)); fill_scales(x, (size_t)M * K, AWQ_DTYPE_F16, -1.f, 1.f);
  CK(hipMalloc(&y, (size_t)M * N * 2));
  void *delta, *nw, *hout;
  CK(hipMalloc(&delta, (size_t)M * K * 2)); fill_scales(delta, (size_t)M * K, AWQ_DTYPE_F16, -1.f, 1.f);
  CK(hipMalloc(&nw, (size_t)K * 2)); fill_scales(nw, (size_t)K, AWQ_DTYPE_F16, 0.5f, 1.5f);
  CK(hipMalloc(&hout, (size_t)M * K * 2));
  const int max_wg = 4096;
  CK(hipMalloc(&dbg, (size_t)max_wg * 64));
  awq_debug_set_stamp_buffer(dbg);
  std::vector<unsigned long long> h((size_t)max_wg * 8);
  for (int rep = 0; rep < 2 * sets + 3; ++rep) {
    CK(hipMemsetAsync(dbg, 0, (size_t)max_wg * 64, st));
    const int rc = fuse ? awq_aux_gemv_repacked_fused(x, K, packed[rep % sets], y, M, K, N, g, AWQ_DTYPE_F16, (fuse & 1) ? x : nullptr, delta,
                                                      nw, hout, 1e-5f, (fuse & 2) ? 1 : 0, st)
                        : awq_gemm_repacked(x, K, packed[rep % sets], nullptr, y, M, K, N, g, AWQ_DTYPE_F16, st);
    if (rc) { fprintf(stderr, "launch failed: %s\n", awq_hip_status_string(rc)); return 1; }
    CK(hipStreamSynchronize(st));
  }
  awq_debug_set_stamp_buffer(nullptr);
  CK(hipMemcpy(h.data(), dbg, (size_t)max_wg * 64, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull; int nwg = 0;
  for (int b = 0; b < max_wg; ++b) if (h[(size_t)b * 8]) { nwg = b + 1; if (h[(size_t)b * 8] < t0) t0 = h[(size_t)b * 8]; }
  const char* names[6] = {"start", "loads issued", "first k-block computed", "all k-blocks computed", "after barrier", "y stored"};
  printf("rstamps (wave 0 of each workgroup): %d workgroups, us after the first workgroup started\n", nwg);
  for (int slot = 0; slot < 6; ++slot) {
    std::vector<double> v;
    for (int b = 0; b < nwg; ++b) if (h[(size_t)b * 8 + slot]) v.push_back((double)(h[(size_t)b * 8 + slot] - t0) * 0.01);
    if (v.empty()) continue;
    std::sort(v.begin(), v.end());
    printf("  %-24s n=%4zu  min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", names[slot], v.size(), v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  }
  return 0;
}

static int cmd_dequant(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: kbench dequant K N g dtype [sets] [iters]\n"); return 2; }
  const int K = atoi(argv[2]), N = atoi(argv[3]), g = atoi(argv[4]), dtype = atoi(argv[5]);
  const int sets = argc > 6 ? atoi(argv[6]) : 8;
  const int iters = argc > 7 ? atoi(argv[7]) : 200;
  const size_t eb = dtype == AWQ_DTYPE_F32 ? 4 : 2;
  auto w = make_weights(sets, K, N, g, dtype);
  std::vector<void*> outs(sets);
  for (auto& o : outs) CK(hipMalloc(&o, (size_t)K * N * eb));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int i) {
    int rc = awq_dequantize(w[i % sets].qw, w[i % sets].sc, w[i % sets].qz, outs[i % sets], K, N, g, dtype, st);
    if (rc) { fprintf(stderr, "awq_dequantize: %s\n", awq_hip_status_string(rc)); exit(1); }
  };
  for (int i = 0; i < sets; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double bytes = (double)K * N / 2 + (double)(K / g) * N / 2 + (double)(K / g) * N * eb + (double)K * N * eb;
  printf("dequant K=%d N=%d g=%d dt=%d sets=%d : %9.3f us  %8.1f GB/s\n", K, N, g, dtype, sets, us, bytes / us / 1e3);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: kbench gemm|dequant|read ...\n"); return 2; }
  std::string cmd = argv[1];
  if (cmd == "gemm") return cmd_gemm(argc, argv);
  if (cmd == "dequant") return cmd_dequant(argc, argv);
  if (cmd == "stamps") return cmd_stamps(argc, argv);
  if (cmd == "rgemm") return cmd_rgemm(argc, argv);
  if (cmd == "rstamps") return cmd_rstamps(argc, argv);
  if (cmd == "read") return cmd_read(argc, argv);
  fprintf(stderr, "unknown command %s\n", argv[1]);
  return 2;
}
