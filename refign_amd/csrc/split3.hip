// refign_amd/csrc/split3.hip -- operand preparation of the split-bf16 products (refign_amd/split32.py): an fp32 matrix x
// becomes hi = bf16(x), lo = bf16(x - hi), laid out as the three terms of  x . y ~ hi.hi' + hi.lo' + lo.hi'  in ONE pass
// (4 bytes read, 6 written per element).  Until round 6 this was eight torch element-wise launches per operand (cast, cast
// back, subtract, cast, zero-fill, three slice copies): 1.5 M launches per fp32-mode step, and the reason the matcher's
// decoders could not afford fp32-class accuracy inside the timed step.
//   term i of row r, column k  ->  out[i * term_stride + r * out_row_stride + k],  k < Kp (columns K .. Kp-1 are zero)
//   order 0: (hi, hi, lo)  -- the activation / gradient side;   order 1: (hi, lo, hi)  -- the weight side
// Side by side along the reduction index (Linear / convolution forward and data gradient): term_stride = Kp,
// out_row_stride = 3 Kp.  Stacked along the rows (weight gradients): term_stride = rows * out_row_stride.
#include "common.h"
#include "mfma.h"

namespace rfn {

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long xs, uint16_t* __restrict__ out, long os,
                                                     long ts, long rows, int K, int Kp, int order, int vec) {
  const int qpr = Kp >> 2;                               // quads per row
  const long total = rows * qpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / qpr;
    const int k0 = (int)(i - r * qpr) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* p = x + r * xs + k0;
    if (vec && k0 + 3 < K) {
      const f32x4 q = *(const f32x4*)p;
      v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k0 + e < K) v[e] = p[e];
    }
    float l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = v[e] - (float)(__bf16)v[e];
    const u32x2 hi = {bf16x2_bits(v[0], v[1]), bf16x2_bits(v[2], v[3])};
    const u32x2 lo = {bf16x2_bits(l[0], l[1]), bf16x2_bits(l[2], l[3])};
    uint16_t* o = out + r * os + k0;
    *(u32x2*)o = hi;
    *(u32x2*)(o + ts) = order ? lo : hi;
    *(u32x2*)(o + 2 * ts) = order ? hi : lo;
  }
}

}  // namespace rfn

extern "C" int rfn_split3_bf16(const float* x, long x_row_stride, void* out, long out_row_stride, long term_stride, long rows,
                               int K, int Kp, int order, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(x && out, "split3_bf16: null pointer");
  RFN_REQUIRE(rows > 0 && K > 0 && Kp >= K && Kp % 4 == 0 && x_row_stride >= K && out_row_stride >= Kp,
              "split3_bf16: rows=%ld K=%d Kp=%d strides %ld / %ld", rows, K, Kp, x_row_stride, out_row_stride);
  RFN_REQUIRE(out_row_stride % 4 == 0 && term_stride % 4 == 0 && (order == 0 || order == 1),
              "split3_bf16: out_row_stride / term_stride must be multiples of 4 elements, order 0 | 1");
  const int vec = (x_row_stride % 4 == 0) && ((size_t)x % 16 == 0);
  const long total = rows * (Kp / 4);
  const int blocks = (int)(total / 256 + 1 > 16384 ? 16384 : total / 256 + 1);
  hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, x_row_stride, (uint16_t*)out,
                     out_row_stride, term_stride, rows, K, Kp, order, vec);
  return check_launch("split3_bf16");
}
