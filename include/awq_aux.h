/*
 * awq_aux.h — fused neighbours of the AWQ linears used by the decode harness.  NOT part of the
 * drop-in operator boundary (that is awq_hip.h): these are the elementwise steps either side of the four
 * projections of a Llama layer in the reference model (python/sglang/srt/models/llama.py:94-106, :188-199),
 * each as one small launch.  fp16 tensors, dense row-major, device pointers, hipStream_t as void*;
 * return 0 or a negative awq_status (awq_hip.h).
 */
#ifndef AWQ_AUX_H_
#define AWQ_AUX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* h[rows, H] += delta (if delta != NULL, written back) ; out = rmsnorm(h) * w   (layers/layernorm.py RMSNorm with residual) */
int awq_aux_add_rmsnorm(void* h, const void* delta, const void* w, void* out, int64_t rows, int64_t H, float eps, void* stream);

/* RoPE + KV-cache write + attention of ONE new token per sequence, in one launch (models/llama.py:188-199: rotary_emb, then
 * RadixAttention decode).  qkv[B, (Hq + 2 Hkv) D] is read only; k, v of the token are stored at pos[b] (which must be < S) and
 * out[B, Hq D] = softmax(scale q K^T) V over cache slots 0..pos[b].  D in {64, 128}, Hq % Hkv == 0.
 * num_splits (1..16) workgroups share each (sequence, head), splitting the context ("flash-decoding"; use > 1 when B * Hq is
 * well below the 256 CUs and the context is long); it needs `workspace` of awq_aux_decode_attention_workspace_bytes(...) bytes,
 * 16-byte aligned, zero-filled ONCE at allocation (tickets in a fixed 64 KiB header; every call leaves them zero — the same buffer
 * may serve calls of different batch sizes and split counts), one per stream of execution; B * Hq <= 16384 with num_splits > 1. */
size_t awq_aux_decode_attention_workspace_bytes(int64_t B, int64_t Hq, int64_t D, int num_splits);
int awq_aux_decode_attention(const void* qkv, const int64_t* pos, const float* cos_table, const float* sin_table, void* k_cache,
                             void* v_cache, void* out, int64_t B, int64_t Hq, int64_t Hkv, int64_t D, int64_t S, float scale,
                             int num_splits, void* workspace, size_t workspace_bytes, void* stream);

/* Greedy sampling tail of a decode step: tokens[b] = argmax(logits[b, :V]) (first maximum, fp16 logits, V % 8 == 0), pos[b] += 1. */
int awq_aux_argmax_advance(const void* logits, int64_t* tokens, int64_t* pos, int64_t B, int64_t V, void* stream);

/* out[tokens, K] = fp16(sum over k of fp32(y[tokens * top_k, K])): the combine step of an MoE layer over its (token, expert) pairs' outputs
 * (the reference's moe_sum_reduce, layers/moe/fused_moe_triton/fused_moe.py), one launch; fp16, K % 8 == 0, both 16-byte aligned.
 * expert_ids (may be NULL) [tokens * top_k]: pairs whose id lies outside [0, num_experts) — padded tokens — are skipped, so the rows of y
 * the block / tile routes never write need no initialisation. */
int awq_aux_moe_sum(const void* y, void* out, int64_t tokens, int64_t top_k, int64_t K, const int32_t* expert_ids, int64_t num_experts,
                    void* stream);

/* act[rows, I] = silu(gate_up[:, :I]) * gate_up[:, I:]   (layers/activation.py SiluAndMul) */
int awq_aux_silu_mul(const void* gate_up, void* act, int64_t rows, int64_t I, void* stream);

/* The repacked decode GEMV (awq_hip.h: awq_gemm_repacked; M <= 16, the SiLU-mul epilogue alone up to 32) with its
 * neighbours folded in:
 *   norm_h != NULL: x is ignored and x = RMSNorm(norm_h + norm_delta) * norm_w is built in the kernel's prologue;
 *                   norm_h_out = norm_h + norm_delta ([M, K], row stride ldx for all three; must not alias norm_h;
 *                   norm_delta must not be NULL — pass zeros).  M * K <= 32768, K <= 16384.
 *   silu_mul != 0:  (alone also for N > 32768, in rounds of narrow strips) `packed` was repacked from tensors whose 16-column groups alternate gate / up
 *                   (column 32 p + r = gate 16 p + r, 32 p + 16 + r = up 16 p + r); y = silu(gate) * up, [M, N / 2].
 * Returns AWQ_ERR_BAD_VARIANT (-7) for a shape without a fused instantiation: run the separate ops instead. */
int awq_aux_gemv_repacked_fused(const void* x, int64_t ldx, const void* packed, void* y, int64_t M, int64_t K, int64_t N,
                                int64_t group_size, int dtype, const void* norm_h, const void* norm_delta, const void* norm_w,
                                void* norm_h_out, float norm_eps, int silu_mul, void* stream);

/* AWQ-MoE decode step (the reference's AWQMoEMethod, python/sglang/srt/layers/quantization/awq.py:661-852, runs
 * NVIDIA Marlin MoE kernels; on ROCm it falls back to moe_wna16's Triton path).  `slots` independent one-row GEMVs in ONE launch:
 * slot s multiplies activation row s / x_div of x (row stride ldx) with the repacked weight of expert expert_ids[s] —
 * `packed_experts` holds the experts' awq_repack() outputs back to back, expert_stride_bytes apart (>= awq_repacked_bytes,
 * multiple of 16) — and writes output row s of y [slots, N] (or [slots, N / 2] with silu_mul = 1: SiLU(gate) * up, experts
 * repacked from gate / up column-interleaved tensors as for awq_aux_gemv_repacked_fused).  slot_scale (may be NULL): the fp32 sum
 * of slot s is multiplied by slot_scale[s] before its one rounding — the routed weight, applied where the reference's fused MoE
 * kernel applies it (mul_routed_weight).  fp16, group_size % 128 == 0; AWQ_ERR_BAD_VARIANT otherwise.  An id outside
 * [0, num_experts) marks a padded slot — the reference's routing writes -1 for the padded tokens of a graph batch
 * (layers/moe/topk.py:705-712) — : its output row is zero-filled on the device and no weight is read (no host check, no sync). */
int awq_aux_moe_gemv(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                     int64_t num_experts, const int32_t* expert_ids, const float* slot_scale, void* y, int64_t slots, int64_t K,
                     int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream);

/* The same step for batches of any size without a host synchronisation: the (token, expert) pairs are sorted by expert on the
 * device and cut into blocks of 16 rows that never straddle two experts (what the reference's moe_align_block_size prepares for
 * its fused MoE kernels, layers/moe/fused_moe_triton/moe_align_block_size.py).  Grid row b multiplies the activation rows of
 * the pairs row_map[16 b .. 16 b + 15] (pair index p -> activation row p / x_div; -1 = padding, nothing is stored for it) with
 * the repacked weight of expert block_expert[b] (< 0: unused block, skipped) and writes output row p of y [pairs, N] (or
 * [pairs, N / 2] with silu_mul); slot_scale[p] as in awq_aux_moe_gemv.  One launch streams each active expert once per 16 of its
 * rows.  fp16, group_size % 128 == 0; AWQ_ERR_BAD_VARIANT otherwise. */
/* The alignment step itself on the device, one small launch: ids [pairs] int32 -> row_map [16 num_blocks] (pair indices grouped by
 * expert, each expert's run padded with -1 to a multiple of 16) and block_expert [num_blocks] (-1 = unused block), num_blocks >=
 * ceil(pairs / 16) + num_experts (the static bound).  Ids outside [0, num_experts) are dropped.  The order of the rows inside an expert's run
 * is not specified (the outputs of awq_aux_moe_gemv_blocks do not depend on it).  num_experts <= 1024; AWQ_ERR_BAD_VARIANT beyond. */
int awq_aux_moe_align_blocks(const int32_t* ids, int64_t pairs, int64_t num_experts, int32_t* row_map, int32_t* block_expert,
                             int64_t num_blocks, void* stream);

int awq_aux_moe_gemv_blocks(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                            const int32_t* row_map, const int32_t* block_expert, int64_t num_blocks, const float* slot_scale,
                            void* y, int64_t K, int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream);

/* Prefill-sized batches (tens of rows per expert and more): the same two steps with blocks of 128 rows on the MFMA tile kernel of the
 * dense prefill path (128 x 256 tiles; the reference's fused MoE kernels run block_m 64..128 there,
 * layers/moe/fused_moe_triton/fused_moe.py) — each active expert is streamed once per 128 of its rows instead of once per 16.
 * awq_aux_moe_align_blocks_n: block_rows in {16, 64, 128}; row_map [block_rows * num_blocks], num_blocks >= ceil(pairs / block_rows) +
 * num_experts.  awq_aux_moe_gemm_blocks: same arguments and rounding points as awq_aux_moe_gemv_blocks over blocks of block_rows =
 * 128 or 64 rows (64: layers of many thinly loaded experts — half the MFMA work per streamed weight; silu_mul: N / 16 even). */
int awq_aux_moe_align_blocks_n(const int32_t* ids, int64_t pairs, int64_t num_experts, int block_rows, int32_t* row_map,
                               int32_t* block_expert, int64_t num_blocks, void* stream);
int awq_aux_moe_gemm_blocks(const void* x, int64_t ldx, int x_div, const void* packed_experts, int64_t expert_stride_bytes,
                            const int32_t* row_map, const int32_t* block_expert, int64_t num_blocks, int block_rows, const float* slot_scale,
                            void* y, int64_t K, int64_t N, int64_t group_size, int dtype, int silu_mul, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AWQ_AUX_H_ */
