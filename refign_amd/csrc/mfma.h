// refign_amd/csrc/mfma.h -- shared pieces of the hand-written gfx950 matrix-core kernels (mfma_gemm.hip, attn.hip,
// conv.hip): operand types, the 32x32x16 MFMA wrappers for bf16 / f16, LDS-DMA, packing and wave-half exchange.
//
// Register maps used throughout (v_mfma_f32_32x32x16_{bf16,f16}, wave64, lane l, g = l >> 5):
//   A operand  : 8 values  A[i = l & 31][k = 8 g + e],  e = 0..7      (one 16-byte register quad)
//   B operand  : 8 values  B[k = 8 g + e][j = l & 31]
//   C/D        : 16 floats D[i = (r & 3) + 8 (r >> 2) + 4 g][j = l & 31],  r = 0..15
// The instruction is a dot product over its 16 k-slots, so WHICH reduction index sits in slot (g, e) is free as long
// as the A and the B operand agree.  The kernels use that freedom instead of cross-lane shuffles: a C/D register
// block (a lane holds rows {0-3, 8-11, 16-19, 24-27} + 4 g of one column) is fed back as a B operand whose slot
// (g, e) means row {0-3, 8-11}[e] + 4 g (+ 16 for the second k-step), and the matching A operand is read from memory
// in that slot order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rfn {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// DT: 1 = bfloat16, 2 = float16 (the ABI's dtype codes; 0 = float32 is not a matrix-core input type here)
template <int DT> struct Elem;
template <> struct Elem<1> {
  using vec8 = bf16x8;
  using vec4 = bf16x4;
  using scalar = __bf16;
  static __device__ __forceinline__ f32x16 mma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Elem<2> {
  using vec8 = f16x8;
  using vec4 = f16x4;
  using scalar = _Float16;
  static __device__ __forceinline__ f32x16 mma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// 4 floats -> 4 x 16-bit (round to nearest even), as two dwords
template <int DT> __device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  using S = typename Elem<DT>::scalar;
  typename Elem<DT>::vec4 v = {(S)a, (S)b, (S)c, (S)d};
  return __builtin_bit_cast(u32x2, v);
}
template <int DT> __device__ __forceinline__ void unpack4(u32x2 p, float (&f)[4]) {
  auto v = __builtin_bit_cast(typename Elem<DT>::vec4, p);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
}
// two k-slot halves (4 + 4 values) -> one MFMA operand
template <int DT> __device__ __forceinline__ typename Elem<DT>::vec8 join8(u32x2 lo, u32x2 hi) {
  u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(typename Elem<DT>::vec8, q);
}

// One LDS-DMA instruction: global -> LDS, 16 bytes per lane, lane i lands at lds_wave_base + 16 i (the destination is
// wave-uniform base + lane order; the SOURCE address is per lane, which is where swizzles and gathers go).  Inline
// asm on purpose: hipcc does not track it, so the hand-off is ours -- `s_waitcnt vmcnt(N)` + barrier before the
// first ds_read of the data (see corr.hip for the measurement behind this choice).
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(sbase) : "memory", "m0");
#pragma clang diagnostic pop
}

// 4 bytes per lane (lane i lands at lds_wave_base + 4 i): small per-stage side data riding in the same ring
__device__ __forceinline__ void lds_dma4(const void* gsrc, void* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(gsrc), "s"(sbase) : "memory", "m0");
#pragma clang diagnostic pop
}

__device__ __forceinline__ void wait_dma_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Values the compiler loaded from global memory BEFORE a loop that issues LDS-DMA: make it wait for them here.  Otherwise
// it places its own `s_waitcnt vmcnt(0)` at their first use INSIDE the loop -- it does not see our DMA instructions, and
// the counter retires in order, so that wait drains the DMA just issued for the next stage, on every iteration: the ring
// never runs ahead (found in round 3 in all three attention kernels).
template <typename T> __device__ __forceinline__ void pin_loaded(const T& v) { asm volatile("" ::"v"(v)); }

// counted hand-off: at most N of this wave's vector-memory instructions still outstanding (they retire in order)
template <int N> __device__ __forceinline__ void wait_dma_upto() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// value of the partner lane (l ^ 32) combined with the own one
__device__ __forceinline__ float half_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// XCD-aware, bijective block remap (8 XCDs, dispatcher places block b on XCD b % 8): consecutive logical ids run on
// one XCD, so neighbouring tiles share that XCD's L2.  Speed only -- nothing depends on the placement.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// exact-erf GELU for 16-bit / 8-bit results: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7 -- two orders below half an
// ulp of a 16-bit result) -- one v_rcp, one v_exp, ten FMAs, no branches.  libm's erff is two branchy polynomials per
// call (both executed by a wave whose lanes disagree): ~110 instructions per element, which made the GELU-fused
// depthwise convolution VALU-bound (round 3: 197 us for 418 MB at the teacher's stage 3).
__device__ __forceinline__ float gelu_erf_fast(float z) {
  const float x = fabsf(z) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -1.44269504088896340736f);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float erf_abs = fmaf(-poly, e, 1.0f);
  return 0.5f * z * (1.0f + copysignf(erf_abs, z));
}

// ---- fp8 (OCP e4m3) helpers of the K5 teacher path (csrc/f8.hip; producers in layernorm.hip / dwconv.hip) --------------
struct f8e4m3 { unsigned char v; };          // storage tag: one e4m3 byte
constexpr float kF8Max = 448.f;
// 4 floats -> 4 e4m3 bytes (round to nearest even, saturating at +-448), byte e = value e
__device__ __forceinline__ unsigned quant4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -kF8Max, kF8Max);
  b = __builtin_amdgcn_fmed3f(b, -kF8Max, kF8Max);
  c = __builtin_amdgcn_fmed3f(c, -kF8Max, kF8Max);
  d = __builtin_amdgcn_fmed3f(d, -kF8Max, kF8Max);
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (unsigned)w;
}
__device__ __forceinline__ void dequant4(unsigned w, float (&f)[4]) {
  auto lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false);
  auto hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[0] = lo[0]; f[1] = lo[1]; f[2] = hi[0]; f[3] = hi[1];
}

}  // namespace rfn
