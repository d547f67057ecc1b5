// tools/experiments/matrix_pipe_corr/split_f16.hip -- operand producer of the matrix-pipe correlation experiment (round 3):
// fp32 NCHW features -> split-fp16 chunk-major, optionally warped on the way.  Uses the bilinear-tap helpers of the product's
// warp.hip (included as source: this is a tools build, its copies of warp's entry points stay inside this library).
#include "../../../refign_amd/csrc/warp.hip"

namespace rfn {
// ---------------------------------------------------------------------------------------------------------------------
// fp32 NCHW features -> "split-fp16, chunk-major" operands of the matrix-core correlation kernel (csrc/corr_f16.hip):
//   out[b][chunk = c / 32][part][y][x][c % 32]   fp16,  part 0 = hi = fp16(v), part 1 = lo = fp16(v - hi)
// (hi + lo carries 22 significand bits of v: the three products hi.hi' + hi.lo' + lo.hi' reproduce an fp32 product to
// ~2^-21).  With `flow` the source is bilinearly warped on the way (warp(feature_source, flow) of uawarpc.py:149-152) --
// the warped fp32 map is never written.  One thread = one pixel x one 32-channel chunk: channel-plane reads are contiguous
// along x across the wave, the thread's 2 x 64 output bytes are contiguous.   grid (ceil(HW / 256), C / 32, B)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                        _Float16* __restrict__ out, int C, int H, int W) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int n = blockIdx.z, chunk = blockIdx.y, NC = C / 32;
  const float* src = x + ((size_t)n * C + (size_t)chunk * 32) * HW;
  Tap t;
  if (flow != nullptr) {
    const int gy = pix / W, gx = pix - gy * W;
    const float* fl = flow + (size_t)n * 2 * HW;
    t = bilinear_tap((float)gx, (float)gy, fl[pix], fl[HW + pix], H, W);
  }
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  unsigned hi[16], lo[16];
#pragma unroll
  for (int c = 0; c < 32; c += 2) {
    float v0, v1;
    if (flow != nullptr) {
      v0 = tap_sample(src + (size_t)c * HW, t);
      v1 = tap_sample(src + (size_t)(c + 1) * HW, t);
    } else {
      v0 = src[(size_t)c * HW + pix];
      v1 = src[(size_t)(c + 1) * HW + pix];
    }
    const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
    const h2 hh = {h0, h1};
    const h2 ll = {(_Float16)(v0 - (float)h0), (_Float16)(v1 - (float)h1)};
    hi[c / 2] = __builtin_bit_cast(unsigned, hh);
    lo[c / 2] = __builtin_bit_cast(unsigned, ll);
  }
  _Float16* ph = out + ((((size_t)n * NC + chunk) * 2 + 0) * HW + pix) * 32;
  _Float16* pl = out + ((((size_t)n * NC + chunk) * 2 + 1) * HW + pix) * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<uint4*>(ph + 8 * q) = make_uint4(hi[4 * q], hi[4 * q + 1], hi[4 * q + 2], hi[4 * q + 3]);
    *reinterpret_cast<uint4*>(pl + 8 * q) = make_uint4(lo[4 * q], lo[4 * q + 1], lo[4 * q + 2], lo[4 * q + 3]);
  }
}

}  // namespace rfn

extern "C" {
int rfx_split_f16(const float* x, const float* flow, void* out, int B, int C, int H, int W, rfn_stream_t stream) {
  RFN_REQUIRE(x && out, "rfx_split_f16: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && C % 32 == 0 && H > 0 && W > 0, "rfx_split_f16: sizes (C %% 32)");
  RFN_REQUIRE((long)H * W < 0x7fffffffL && B <= 65535, "rfx_split_f16: tensor too large");
  dim3 grid(cdiv((long)H * W, 256), C / 32, B);
  hipLaunchKernelGGL(split_f16_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, flow, (_Float16*)out, C, H, W);
  return check_launch("split_f16_kernel");
}

}
