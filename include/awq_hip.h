/*
 * awq_hip.h — C ABI of the MI355X (gfx950) AWQ int4 quantized-linear kernels.
 *
 * This is the drop-in boundary for ONE path of kvcache-ai/sglang_awq: the operator pair that
 * AWQLinearMethod.apply() runs (python/sglang/srt/layers/quantization/awq.py:434-451).
 * Plain pointers and sizes only; no torch types; nothing here allocates, frees or synchronises.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer on the GPU that owns `stream`; tensors are dense row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); the call only
 *     enqueues kernels on it — safe inside hipGraph stream capture (model_runner.py:2765-2771
 *     replays the captured decode graph, so the op may not sync or allocate);
 *   - dtype: AWQ_DTYPE_F16 / AWQ_DTYPE_BF16 / AWQ_DTYPE_F32 is the dtype of scales, activations
 *     and outputs (the reference op returns `scales.dtype`, awq_kernel.cu:198-199);
 *   - return value: 0 on success, a negative AWQ_ERR_* code otherwise (never throws, never aborts);
 *   - AWQ tensor format (AutoAWQ "GEMM" layout, consumed as stored on disk):
 *       qweight int32 [K, N/8]   logical column 8c+j in nibble {0,4,1,5,2,6,3,7}[j] of word c
 *       qzeros  int32 [K/g, N/8] same nibble order
 *       scales  dtype [K/g, N]
 *       W[k,n] = (q[k,n] - z[k/g,n]) * s[k/g,n], one rounding in `dtype`.
 */
#ifndef AWQ_HIP_H_
#define AWQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWQ_HIP_ABI_VERSION 2

enum awq_dtype { AWQ_DTYPE_F16 = 0, AWQ_DTYPE_BF16 = 1, AWQ_DTYPE_F32 = 2 };

enum awq_status {
  AWQ_OK = 0,
  AWQ_ERR_NULL_POINTER = -1,
  AWQ_ERR_BAD_SHAPE = -2,     /* N % 8 != 0, K % g != 0, non-positive dims, ldx < K */
  AWQ_ERR_BAD_DTYPE = -3,
  AWQ_ERR_BAD_SPLIT_K = -4,   /* split_k_iters not a power of two in [1, 32] (awq_triton.py:307-308) */
  AWQ_ERR_WORKSPACE = -5,     /* workspace missing / smaller than awq_gemm_workspace_bytes() */
  AWQ_ERR_MISALIGNED = -6,    /* a pointer is not 16-byte aligned (the op's uint4 loads: awq_kernel.cu:142,161) */
  AWQ_ERR_BAD_VARIANT = -7,
  AWQ_ERR_LAUNCH = -100       /* hipGetLastError() != hipSuccess after the launch */
};

/* ABI / build identification. */
int awq_hip_abi_version(void);
const char* awq_hip_build_info(void);               /* e.g. "gfx950 hipcc 7.2 ..." (static string) */
const char* awq_hip_status_string(int status);      /* static string for an awq_status value */

/*
 * awq_dequantize — replaces the op `sgl_kernel::awq_dequantize`
 *   schema   sgl-kernel/csrc/common_extension.cc:126-127
 *   host     sgl-kernel/csrc/gemm/awq_kernel.cu:186-221,  kernel :126-184
 *   python   sgl-kernel/python/sgl_kernel/gemm.py:8-11
 * out[K, N] (dtype) <- dequantise(qweight, scales, qzeros); bit-exact with the reference
 * (integer unpack exact, (q - z) exact, one round-to-nearest-even multiply).
 * group_size = K / scales.size(0) as the reference infers it (awq_kernel.cu:189).
 */
int awq_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out,
                   int64_t K, int64_t N, int64_t group_size, int dtype, void* stream);

/*
 * awq_gemm — the fused dequantise + GEMM the north star names `sgl_kernel.awq_gemm`; argument
 * order and meaning follow the only fused GEMM in the reference,
 *   awq_gemm_triton(input, qweight, scales, qzeros, split_k_iters)  awq_triton.py:284-339
 * and the numerics follow what AWQLinearMethod.apply computes (awq.py:446-447): W is rounded to
 * `dtype` element by element exactly as awq_dequantize would, products are accumulated in fp32,
 * the sum is rounded once to `dtype`.
 *
 *   x [M, K] with row stride `ldx` elements (ldx >= K; lets a row-parallel rank pass its K-slice
 *   of a wider activation without a copy: linear.py:1395-1399), y [M, N] dense.
 *   split_k_iters: power of two in [1, 32] (validated like the reference); a HINT — the kernel
 *   picks its own K partition; results do not depend on it and are run-to-run deterministic.
 *   workspace: device scratch of at least awq_gemm_workspace_bytes(...) bytes, 16-byte aligned,
 *   zero-filled ONCE when allocated (arrival counters live in it and every call leaves them
 *   zero again).  It may be shared by successive calls on one stream, not by concurrent streams.
 *   For M > 32 (fp16, K % 128 == 0, g % 128 == 0) awq_gemm_workspace_bytes() also
 *   covers one re-laid-out copy of the weight: the call then repacks on the fly and runs the kernel
 *   of awq_gemm_repacked; with less workspace it keeps to the checkpoint-layout kernels.
 *   bias (may be NULL): [N] in `dtype`, added AFTER the sum is rounded to `dtype`, with a second
 *   rounding — the reference's in-place `out.add_(bias)` (awq.py:449-450).
 */
size_t awq_gemm_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype);

int awq_gemm(const void* x, int64_t ldx, const int32_t* qweight, const void* scales,
             const int32_t* qzeros, const void* bias, void* y, void* workspace, size_t workspace_bytes,
             int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype, int split_k_iters,
             void* stream);

/*
 * Same as awq_gemm with the kernel variant forced (tests and A/B benchmarks).  variant:
 *   AWQ_GEMM_AUTO     the heuristic awq_gemm uses
 *   AWQ_GEMM_GENERIC  any shape / dtype, VALU, no MFMA
 *   AWQ_GEMM_SKINNY   M <= 32: dequantised tiles fed straight from registers to MFMA,
 *                     K split over waves and workgroups, in-launch deterministic reduction
 *   AWQ_GEMM_TILED    large M: LDS-staged MFMA tiles
 * `tune` packs variant-specific knobs (0 = defaults); see csrc/awq_capi.hip.
 * Returns AWQ_ERR_BAD_VARIANT if the variant cannot run the given shape.
 */
enum awq_gemm_variant { AWQ_GEMM_AUTO = 0, AWQ_GEMM_GENERIC = 1, AWQ_GEMM_SKINNY = 2, AWQ_GEMM_TILED = 3 };

int awq_gemm_ex(const void* x, int64_t ldx, const int32_t* qweight, const void* scales,
                const int32_t* qzeros, const void* bias, void* y, void* workspace, size_t workspace_bytes,
                int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype, int split_k_iters,
                int variant, int64_t tune, void* stream);

/*
 * Optional one-time re-layout for decode.  The reference re-lays AWQ weights out only for
 * its NVIDIA Marlin path (sgl-kernel/csrc/gemm/marlin/awq_marlin_repack.cu:13-253, called from
 * AWQMarlinLinearMethod.process_weights_after_loading); its plain AWQ method keeps the checkpoint layout
 * (awq.py:429-432).  awq_repack writes an MFMA-fragment-major copy of (qweight, qzeros, scales) into
 * `packed` (awq_repacked_bytes() bytes, 16-byte aligned); awq_gemm_repacked then computes the same result as
 * awq_gemm — same per-element rounding, fp32 accumulation, bias epilogue — from that copy: for M <= 32 a
 * streaming GEMV (linear weight stream, no cross-workgroup reduction, no workspace), for larger M an MFMA-bound
 * tiled kernel whose B fragments go from the copy straight to registers (an M <= 32 call whose reduction scratch
 * would not fit returns AWQ_ERR_BAD_VARIANT: use awq_gemm).
 * Supported: fp16 or bf16 (the dtype of the scales the copy is made from = the dtype of x and y), K % 128 == 0, group_size a
 * multiple of 128 or one of 32 / 64 (the group sizes of the reference's Triton path, awq_triton.py:250); otherwise
 * awq_repacked_bytes returns 0 and the other two return AWQ_ERR_BAD_VARIANT (callers keep using awq_gemm).  The hand-tuned kernels
 * (straight-line / loop-form GEMV, hand-pipelined prefill tiles) exist for fp16 with group_size % 128 == 0; bf16 and the small
 * groups run compiler-scheduled kernels on the same layout.
 */
size_t awq_repacked_bytes(int64_t K, int64_t N, int64_t group_size, int dtype);

int awq_repack(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* packed, int64_t K,
               int64_t N, int64_t group_size, int dtype, void* stream);

int awq_gemm_repacked(const void* x, int64_t ldx, const void* packed, const void* bias, void* y, int64_t M,
                      int64_t K, int64_t N, int64_t group_size, int dtype, void* stream);

/*
 * awq_gemm_repacked with an optional scratch buffer (>= awq_gemm_repacked_workspace_bytes(), 16-byte aligned, its first 4096 bytes
 * zero-filled ONCE when allocated — arrival counters, left at zero by every call — one buffer per stream of execution, the same rules
 * as awq_gemm's).  Two routes use it:
 *   9 .. 32 rows on a narrow matrix (N <= 8192): the one-strip-per-workgroup GEMV re-reads all of x in every workgroup; with the scratch
 *     that case runs wide strips with K split across workgroups (fp32 partials summed in fixed order by the last workgroup to arrive);
 *   33 rows and up where the MFMA tiling leaves most CUs without a tile (narrow or deep matrices, up to a few hundred rows): the tile
 *     kernel runs with K split over workgroups, fp32 partial tiles go to the scratch and a second small launch adds them in slice order,
 *     adds the bias and rounds once (11008 x 4096 at 128 rows: 89 -> 26 us).
 * Both are deterministic; sums agree with the no-workspace routes to fp32 rounding, not bit for bit.  workspace == NULL or too small:
 * exactly awq_gemm_repacked.  awq_gemm_repacked_workspace_bytes returns 0 where the scratch would not be used (at most 4096 + 32 MiB).
 */
size_t awq_gemm_repacked_workspace_bytes(int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype);

int awq_gemm_repacked_ws(const void* x, int64_t ldx, const void* packed, const void* bias, void* y, void* workspace,
                         size_t workspace_bytes, int64_t M, int64_t K, int64_t N, int64_t group_size, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AWQ_HIP_H_ */
