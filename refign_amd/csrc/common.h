// refign_amd/csrc/common.h -- shared helpers for the gfx950 kernels (error reporting, launch checks, math).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/refign_hip.h"

namespace rfn {

// Thread-local last-error message (rfn_last_error()).
char* err_buf();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RFN_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RFN_OK;
}

constexpr int kWave = 64;  // CDNA wavefront width

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// fp32 -> bf16 bits, round to nearest even (torch's conversion; NaN stays NaN): gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32, two values per instruction).  The integer formulation (add 0x7fff + lsb, NaN select) is 5-6
// instructions per element -- it made up a third of the instructions of the 16-bit element-wise kernels.
__device__ __forceinline__ unsigned bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ unsigned bf16x2_bits(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// zero-fill as a kernel on `st` (capi.hip: why not hipMemsetAsync)
int zero_async(void* p, size_t bytes, hipStream_t st);

}  // namespace rfn

#define RFN_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return rfn::fail(RFN_EINVAL, __VA_ARGS__); \
  } while (0)
