/*
 * oracle/corr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's spatial correlation sampler
 * (FlowNet-C style), forward and backward, float and double.
 *
 *   forward  follows /root/reference/models/correlation_ops/correlation.cpp:13-42  (correlate_patch)
 *                    and :80-129 (correlation_cpp_forward: output geometry, loop nest, OMP collapse(2))
 *   backward follows correlation.cpp:44-78 (correlate_patch_grad) and :131-183
 *                    (correlation_cpp_backward: OMP over the batch only, scatter-add into both grads)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (refign_amd/) never does: it fails loudly when
 * the HIP library is missing.
 *
 * Parity pin: tests/test_oracle_cpu.py checks this file against (a) the reference
 * C++ itself compiled into oracle/_ref/ by oracle/build_ref.py (when present) and
 * (b) the committed golden vectors in tests/golden/corr_*.npz that were produced
 * by that compiled reference.
 *
 * Layouts: inputs NCHW contiguous; output (B, patchH, patchW, oH, oW) contiguous.
 */
#include <stddef.h>
#include <string.h>

#define DEFINE_ORACLE(T, SUFFIX)                                                          \
  static void correlate_patch_##SUFFIX(const T* in1, const T* in2, T* dst, int C, int iH,  \
                                       int iW, int kH, int kW, int dilH, int dilW, int u,  \
                                       int v, int shiftU, int shiftV) {                    \
    /* correlation.cpp:25-41: channel-outer, kernel window inner, both taps bounds-checked \
       against the INPUT size (zero contribution outside).  Accumulates into *dst in the   \
       same order so the float result is bit-identical to the reference. */                \
    for (int c = 0; c < C; ++c) {                                                          \
      const T* p1 = in1 + (size_t)c * iH * iW;                                             \
      const T* p2 = in2 + (size_t)c * iH * iW;                                             \
      for (int i = 0; i < kH; ++i) {                                                       \
        int i1 = u + i * dilH, i2 = i1 + shiftU;                                           \
        if (i1 < 0 || i1 >= iH || i2 < 0 || i2 >= iH) continue;                            \
        for (int j = 0; j < kW; ++j) {                                                     \
          int j1 = v + j * dilW, j2 = j1 + shiftV;                                         \
          if (j1 < 0 || j1 >= iW || j2 < 0 || j2 >= iW) continue;                          \
          *dst += p1[(size_t)i1 * iW + j1] * p2[(size_t)i2 * iW + j2];                     \
        }                                                                                  \
      }                                                                                    \
    }                                                                                      \
  }                                                                                        \
                                                                                           \
  void oracle_corr_fwd_##SUFFIX(const T* in1, const T* in2, T* out, int B, int C, int iH,  \
                                int iW, int kH, int kW, int patchH, int patchW, int padH,  \
                                int padW, int dilH, int dilW, int dpH, int dpW, int dH,    \
                                int dW) {                                                  \
    /* correlation.cpp:93-100 */                                                           \
    const int radH = (patchH - 1) / 2, radW = (patchW - 1) / 2;                            \
    const int oH = (iH + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1;                       \
    const int oW = (iW + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;                       \
    memset(out, 0, sizeof(T) * (size_t)B * patchH * patchW * oH * oW);                     \
    int n, ph;                                                                             \
    _Pragma("omp parallel for collapse(2)")                                                \
    for (n = 0; n < B; ++n)                                                                \
      for (ph = 0; ph < patchH; ++ph)                                                      \
        for (int pw = 0; pw < patchW; ++pw)                                                \
          for (int h = 0; h < oH; ++h)                                                     \
            for (int w = 0; w < oW; ++w)                                                   \
              correlate_patch_##SUFFIX(                                                    \
                  in1 + (size_t)n * C * iH * iW, in2 + (size_t)n * C * iH * iW,            \
                  out + ((((size_t)n * patchH + ph) * patchW + pw) * oH + h) * oW + w, C,  \
                  iH, iW, kH, kW, dilH, dilW, -padH + h * dH, -padW + w * dW,              \
                  (ph - radH) * dpH, (pw - radW) * dpW);                                   \
  }                                                                                        \
                                                                                           \
  void oracle_corr_bwd_##SUFFIX(const T* in1, const T* in2, const T* gout, T* g1, T* g2,   \
                                int B, int C, int iH, int iW, int oH, int oW, int kH,      \
                                int kW, int patchH, int patchW, int padH, int padW,        \
                                int dilH, int dilW, int dpH, int dpW, int dH, int dW) {    \
    /* correlation.cpp:144-181: zeros_like both grads, OMP over n, loop ph,pw,h,w,         \
       then (correlate_patch_grad :60-77) c,i,j scatter-adds. */                           \
    const int radH = (patchH - 1) / 2, radW = (patchW - 1) / 2;                            \
    const size_t plane = (size_t)iH * iW;                                                  \
    memset(g1, 0, sizeof(T) * (size_t)B * C * plane);                                      \
    memset(g2, 0, sizeof(T) * (size_t)B * C * plane);                                      \
    int n;                                                                                 \
    _Pragma("omp parallel for")                                                            \
    for (n = 0; n < B; ++n) {                                                              \
      const T* a = in1 + (size_t)n * C * plane;                                            \
      const T* b = in2 + (size_t)n * C * plane;                                            \
      T* ga = g1 + (size_t)n * C * plane;                                                  \
      T* gb = g2 + (size_t)n * C * plane;                                                  \
      for (int ph = 0; ph < patchH; ++ph)                                                  \
        for (int pw = 0; pw < patchW; ++pw)                                                \
          for (int h = 0; h < oH; ++h)                                                     \
            for (int w = 0; w < oW; ++w) {                                                 \
              const T g =                                                                  \
                  gout[((((size_t)n * patchH + ph) * patchW + pw) * oH + h) * oW + w];     \
              const int u = -padH + h * dH, v = -padW + w * dW;                            \
              const int sU = (ph - radH) * dpH, sV = (pw - radW) * dpW;                    \
              for (int c = 0; c < C; ++c)                                                  \
                for (int i = 0; i < kH; ++i) {                                             \
                  int i1 = u + i * dilH, i2 = i1 + sU;                                     \
                  if (i1 < 0 || i1 >= iH || i2 < 0 || i2 >= iH) continue;                  \
                  for (int j = 0; j < kW; ++j) {                                           \
                    int j1 = v + j * dilW, j2 = j1 + sV;                                   \
                    if (j1 < 0 || j1 >= iW || j2 < 0 || j2 >= iW) continue;                \
                    const T v1 = a[c * plane + (size_t)i1 * iW + j1];                      \
                    const T v2 = b[c * plane + (size_t)i2 * iW + j2];                      \
                    gb[c * plane + (size_t)i2 * iW + j2] += g * v1;                        \
                    ga[c * plane + (size_t)i1 * iW + j1] += g * v2;                        \
                  }                                                                        \
                }                                                                          \
            }                                                                              \
    }                                                                                      \
  }

DEFINE_ORACLE(float, f32)
DEFINE_ORACLE(double, f64)
