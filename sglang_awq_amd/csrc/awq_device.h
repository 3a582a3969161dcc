// Device-side building blocks shared by the gfx950 AWQ kernels (wave64, CDNA4 only).
//
// AWQ packs 8 int4 per int32 with logical column j in nibble {0,4,1,5,2,6,3,7}[j]
// (reference: awq_triton.py:56-69).  Masking a word with 0x000f000f therefore yields logical
// columns (0,1) as the two 16-bit halves, (w >> 4) gives (2,3), (w >> 8) (4,5), (w >> 12) (6,7):
// packed-half pairs come out in natural column order with no shuffling.
//
// fp16 path: OR-ing 0x6400 into a 16-bit lane makes the half 1024 + n (n < 16 exact); leaving the
// nibble at bit 4 makes 1024 + 16 n, and fma(h, 1/16, -(64 + z)) = n - z exactly — one shift per
// word instead of three.  (q - z) is exact, the multiply by the scale is a single v_pk_mul_f16
// (round-to-nearest-even, fp16 denormals on), matching the reference's sub.f16x2 + mul.rn.f16x2
// (awq_kernel.cu:151-158) bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace awq {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr uint32_t kLoNib = 0x000f000fu;
constexpr uint32_t kHiNib = 0x00f000f0u;
constexpr uint32_t kMagicF16 = 0x64006400u;  // half2(1024, 1024)

// (a & mask) | orv in ONE instruction.  gfx9-family VOP3 cannot encode two 32-bit literals, so hipcc
// emits v_and + v_or for the literal form; with the mask in an SGPR and the OR value in a VGPR the
// fused v_and_or_b32 is encodable (one constant-bus operand).
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t orv) {
  uint32_t d;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(mask), "v"(orv));
  return d;
}

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// Zero-point constants of one packed qzeros word, in the encodings unpack_sub_f16 expects.
struct ZeroF16 {
  half2_t z01, z23, z45, z67;  // (1024+z0,1024+z1), (64+z2,64+z3), (1024+z4,1024+z5), (64+z6,64+z7)
};

__device__ __forceinline__ ZeroF16 make_zero_f16(uint32_t zw) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t z8 = zw >> 8;
  ZeroF16 z;
  z.z01 = as_h2((zw & kLoNib) | kMagicF16);
  z.z23 = as_h2((zw & kHiNib) | kMagicF16) * sixteenth;  // exact: (1024 + 16 z) / 16
  z.z45 = as_h2((z8 & kLoNib) | kMagicF16);
  z.z67 = as_h2((z8 & kHiNib) | kMagicF16) * sixteenth;
  return z;
}

// d[t] = (q - z) for logical columns (2t, 2t+1) of one packed word, exact small integers in fp16.
__device__ __forceinline__ void unpack_sub_f16(uint32_t w, const ZeroF16& z, half2_t (&d)[4]) {
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t w8 = w >> 8;
  const uint32_t magic = kMagicF16;
  d[0] = as_h2(and_or(w, kLoNib, magic)) - z.z01;
  d[1] = __builtin_elementwise_fma(as_h2(and_or(w, kHiNib, magic)), sixteenth, -z.z23);
  d[2] = as_h2(and_or(w8, kLoNib, magic)) - z.z45;
  d[3] = __builtin_elementwise_fma(as_h2(and_or(w8, kHiNib, magic)), sixteenth, -z.z67);
}

// Plain integer nibble of logical column j (used by the bf16 / fp32 / generic paths).
__device__ __forceinline__ int nibble_of_col(uint32_t w, int j) {
  // shift = 4 * {0,4,1,5,2,6,3,7}[j] = 16 * (j & 1) + 4 * (j >> 1)
  return (int)((w >> (((j & 1) << 4) + ((j >> 1) << 2))) & 0xFu);
}

__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }

// round-to-nearest-even float -> bf16 bits; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950.
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (bf16_t)f);
}

__device__ __forceinline__ uint16_t float_to_half_bits(float f) { return __builtin_bit_cast(uint16_t, (half_t)f); }
__device__ __forceinline__ float half_bits_to_float(uint16_t h) { return (float)__builtin_bit_cast(half_t, h); }

// v_perm_b32 selectors: build (a.lo16, b.lo16) / (a.hi16, b.hi16) from two packed pairs.
__device__ __forceinline__ uint32_t pack_lo16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t pack_hi16(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }


// ---------------------------------------------------------------------------------------------
// dtype-generic scalar helpers (generic kernel, epilogues).  DT is an awq_dtype value.
template <int DT> struct ElemBytes { static constexpr int v = (DT == 2) ? 4 : 2; };

template <int DT>
__device__ __forceinline__ float load_as_float(const void* p, size_t i) {
  if constexpr (DT == 0) return (float)((const half_t*)p)[i];
  else if constexpr (DT == 1) return bf16_bits_to_float(((const uint16_t*)p)[i]);
  else return ((const float*)p)[i];
}

// round a float once to the storage dtype and return it as float
template <int DT>
__device__ __forceinline__ float round_to_dtype(float v) {
  if constexpr (DT == 0) return (float)(half_t)v;
  else if constexpr (DT == 1) return bf16_bits_to_float(float_to_bf16_bits(v));
  else return v;
}

// y[i] = round(v) (+ bias[n], second rounding: the reference's in-place out.add_(bias), awq.py:449-450)
template <int DT>
__device__ __forceinline__ void store_output(void* y, size_t i, float v, const void* bias, int n) {
  if constexpr (DT == 0) {
    half_t h = (half_t)v;
    if (bias) h = h + ((const half_t*)bias)[n];
    ((half_t*)y)[i] = h;
  } else if constexpr (DT == 1) {
    float r = round_to_dtype<1>(v);
    if (bias) r = round_to_dtype<1>(r + bf16_bits_to_float(((const uint16_t*)bias)[n]));
    ((uint16_t*)y)[i] = float_to_bf16_bits(r);
  } else {
    ((float*)y)[i] = bias ? v + ((const float*)bias)[n] : v;
  }
}

// the 16-bit pattern store_output<DT> would write (DT = 0 fp16, 1 bf16): same operations, for epilogues that park a tile in LDS first
template <int DT>
__device__ __forceinline__ uint16_t output_bits16(float v, const void* bias, int n) {
  static_assert(DT == 0 || DT == 1, "16-bit outputs");
  if constexpr (DT == 0) {
    half_t h = (half_t)v;
    if (bias) h = h + ((const half_t*)bias)[n];
    return __builtin_bit_cast(uint16_t, h);
  } else {
    float r = round_to_dtype<1>(v);
    if (bias) r = round_to_dtype<1>(r + bf16_bits_to_float(((const uint16_t*)bias)[n]));
    return float_to_bf16_bits(r);
  }
}

// The 8 dequantised values of one packed word, each rounded to the storage dtype (as the reference
// materialises W before its matmul, awq.py:446), returned as floats.  sc points at the 8 scales.
template <int DT>
__device__ __forceinline__ void dequant_word(uint32_t w, uint32_t zw, const void* sc, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float d = (float)(nibble_of_col(w, j) - nibble_of_col(zw, j));
    o[j] = round_to_dtype<DT>(d * load_as_float<DT>(sc, j));
  }
}

}  // namespace awq
