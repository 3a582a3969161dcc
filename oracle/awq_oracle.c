/*
 * CPU oracle for the AWQ int4 quantized-linear path, plain C.
 *
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile (and __graft_entry__.build()) into
 * oracle/_build/libawq_oracle.so and loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The shipped package never links or loads it.
 *
 * Restates (does not copy) the reference arithmetic:
 *   awq_oracle_unpack      nibble order [0,4,1,5,2,6,3,7]  (awq_triton.py:56-69, :351-355)
 *   awq_oracle_dequantize  out[k,8c+j] = (nib(qw[k,c],P[j]) - nib(qz[k/g,c],P[j])) * s[k/g,8c+j]
 *                          one rounding in the scale dtype (awq_kernel.cu:126-184)
 *   awq_oracle_gemm        y = x @ dequant(W), W rounded to the scale dtype first (awq.py:446-447),
 *                          contraction carried in double, one rounding to the output dtype
 *
 * Parity pin: tests/test_oracle_golden.py checks this library bit-for-bit against the golden
 * vectors captured from the reference's own CPU-runnable functions (tests/golden/).
 *
 * dtype codes match include/awq_hip.h: 0 = fp16, 1 = bf16, 2 = fp32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static const int kNibbleOfCol[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* ---------------------------------------------------------------- scalar conversions */
static inline float bits_to_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static float half_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  if (exp == 0) {
    if (man == 0) return bits_to_f32(sign);
    /* subnormal: value = man * 2^-24 */
    float v = (float)man * 5.9604644775390625e-8f;
    return sign ? -v : v;
  }
  if (exp == 31) return bits_to_f32(sign | 0x7F800000u | (man << 13));
  return bits_to_f32(sign | ((exp + 112u) << 23) | (man << 13));
}

/* round-to-nearest-even double -> fp16 bits (single rounding) */
static uint16_t f64_to_half(double d) {
  if (isnan(d)) return 0x7E00u;
  uint16_t sign = signbit(d) ? 0x8000u : 0;
  double a = fabs(d);
  if (a >= 65520.0) return sign | 0x7C00u;          /* rounds to inf */
  if (a < 6.103515625e-05) {                        /* subnormal or zero: quantum 2^-24 */
    double q = nearbyint(a * 16777216.0);           /* ties-to-even in the default mode */
    return sign | (uint16_t)q;                      /* q == 1024 lands on the smallest normal */
  }
  int e;
  double m = frexp(a, &e);                          /* a = m * 2^e, m in [0.5, 1) */
  double q = nearbyint(m * 2048.0);                 /* 11 significant bits */
  if (q == 2048.0) { q = 1024.0; e += 1; }
  int biased = e - 1 + 15;
  return sign | (uint16_t)((biased << 10) + ((int)q - 1024));
}

static inline float bf16_to_f32(uint16_t b) { return bits_to_f32((uint32_t)b << 16); }

/* round-to-nearest-even double -> bf16 bits (single rounding; bf16 shares fp32's exponent range) */
static uint16_t f64_to_bf16(double d) {
  if (isnan(d)) return 0x7FC0u;
  uint16_t sign = signbit(d) ? 0x8000u : 0;
  double a = fabs(d);
  if (a == 0.0) return sign;
  if (a < 1.17549435082228750797e-38) {             /* bf16 subnormal: quantum 2^-133 */
    double q = nearbyint(ldexp(a, 133));
    return sign | (uint16_t)q;
  }
  int e;
  double m = frexp(a, &e);
  double q = nearbyint(m * 256.0);                  /* 8 significant bits */
  if (q == 256.0) { q = 128.0; e += 1; }
  int biased = e - 1 + 127;
  if (biased >= 255) return sign | 0x7F80u;
  return sign | (uint16_t)((biased << 7) + ((int)q - 128));
}

static inline int nib(uint32_t w, int p) { return (int)((w >> (4 * p)) & 0xFu); }

static int check_dims(int64_t K, int64_t N, int64_t g, int dtype) {
  if (K <= 0 || N <= 0 || g <= 0) return -1;
  if (N % 8 || K % g) return -1;
  if (dtype < 0 || dtype > 2) return -2;
  return 0;
}

/* ---------------------------------------------------------------- entry points */
int awq_oracle_unpack(const int32_t* packed, uint8_t* out, int64_t rows, int64_t cols_packed) {
  if (rows < 0 || cols_packed < 0) return -1;
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t c = 0; c < cols_packed; ++c) {
      uint32_t w = (uint32_t)packed[r * cols_packed + c];
      for (int j = 0; j < 8; ++j) out[(r * cols_packed + c) * 8 + j] = (uint8_t)nib(w, kNibbleOfCol[j]);
    }
  return 0;
}

static inline double scale_as_double(const void* scales, int64_t idx, int dtype) {
  if (dtype == 0) return (double)half_to_f32(((const uint16_t*)scales)[idx]);
  if (dtype == 1) return (double)bf16_to_f32(((const uint16_t*)scales)[idx]);
  return (double)((const float*)scales)[idx];
}

/* one dequantised element, already rounded to the storage dtype, returned as double */
static inline double dequant_rounded(int q, int z, double s, int dtype) {
  double p = (double)(q - z) * s;                   /* exact: |q-z| <= 15, s has <= 24 bits */
  if (dtype == 0) return (double)half_to_f32(f64_to_half(p));
  if (dtype == 1) return (double)bf16_to_f32(f64_to_bf16(p));
  return (double)(float)p;
}

int awq_oracle_dequantize(const int32_t* qweight, const void* scales, const int32_t* qzeros, void* out,
                          int64_t K, int64_t N, int64_t g, int dtype, int threads) {
  int rc = check_dims(K, N, g, dtype);
  if (rc) return rc;
  const int64_t nc = N / 8;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < K; ++k) {
    const int64_t grp = k / g;
    for (int64_t c = 0; c < nc; ++c) {
      uint32_t w = (uint32_t)qweight[k * nc + c];
      uint32_t zw = (uint32_t)qzeros[grp * nc + c];
      for (int j = 0; j < 8; ++j) {
        int p = kNibbleOfCol[j];
        int64_t n = c * 8 + j;
        double prod = (double)(nib(w, p) - nib(zw, p)) * scale_as_double(scales, grp * N + n, dtype);
        if (dtype == 0) ((uint16_t*)out)[k * N + n] = f64_to_half(prod);
        else if (dtype == 1) ((uint16_t*)out)[k * N + n] = f64_to_bf16(prod);
        else ((float*)out)[k * N + n] = (float)prod;
      }
    }
  }
  return 0;
}

/* y[M,N] (storage dtype) and, if y_exact != NULL, the un-rounded double result. */
int awq_oracle_gemm(const void* x, const int32_t* qweight, const void* scales, const int32_t* qzeros,
                    void* y, double* y_exact, int64_t M, int64_t K, int64_t N, int64_t g, int dtype,
                    int threads) {
  int rc = check_dims(K, N, g, dtype);
  if (rc) return rc;
  if (M < 0) return -1;
  const int64_t nc = N / 8;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel
  {
    double* acc = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1) * 8);
#pragma omp for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
      for (int64_t i = 0; i < M * 8; ++i) acc[i] = 0.0;
      for (int64_t k = 0; k < K; ++k) {
        const int64_t grp = k / g;
        uint32_t w = (uint32_t)qweight[k * nc + c];
        uint32_t zw = (uint32_t)qzeros[grp * nc + c];
        double wv[8];
        for (int j = 0; j < 8; ++j) {
          int p = kNibbleOfCol[j];
          wv[j] = dequant_rounded(nib(w, p), nib(zw, p), scale_as_double(scales, grp * N + c * 8 + j, dtype), dtype);
        }
        for (int64_t m = 0; m < M; ++m) {
          double xv = scale_as_double(x, m * K + k, dtype);
          for (int j = 0; j < 8; ++j) acc[m * 8 + j] += xv * wv[j];
        }
      }
      for (int64_t m = 0; m < M; ++m)
        for (int j = 0; j < 8; ++j) {
          int64_t o = m * N + c * 8 + j;
          double v = acc[m * 8 + j];
          if (y_exact) y_exact[o] = v;
          if (y) {
            if (dtype == 0) ((uint16_t*)y)[o] = f64_to_half(v);
            else if (dtype == 1) ((uint16_t*)y)[o] = f64_to_bf16(v);
            else ((float*)y)[o] = (float)v;
          }
        }
    }
    free(acc);
  }
  return 0;
}

int awq_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
