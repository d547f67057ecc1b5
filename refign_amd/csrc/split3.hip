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

// The same split of a VIRTUAL channel concatenation of up to four fp32 maps of one (B, H, W) -- the inputs of the matcher's decoders
// are cat(correlation volume, flow, [feature,] log-variance) (uawarpc.py:136-160), parts that live in different layouts (NCHW from the
// correlation / up-sampling kernels, channels-last from the convolutions) -- straight into the (B, H, W, 3 Cp) operand of the
// convolution: one pass instead of torch.cat + a layout copy + the split.  A workgroup stages 64 pixels x all channels in LDS
// (reads run along the pixels of a channel plane, writes along the channels of a pixel).
struct CatParts {
  const float* p[4];
  long sb[4], sc[4], sh[4], sw[4];
  int c[4];
  int n;
};
constexpr int kCatMaxC = 96;

__global__ __launch_bounds__(256) void split3_cat_kernel(CatParts a, uint16_t* __restrict__ out, long total, int H, int W, int C,
                                                         int Cp) {
  __shared__ float tile[kCatMaxC][65];
  const int t = threadIdx.x, px = t & 63;
  const long gp = (long)blockIdx.x * 64 + px;
  const bool ok = gp < total;
  long b = 0;
  int y = 0, x = 0;
  if (ok) {
    b = gp / ((long)H * W);
    const int r = (int)(gp - b * (long)H * W);
    y = r / W;
    x = r - y * W;
  }
  int coff = 0;
  for (int part = 0; part < a.n; ++part) {
    const float* base = a.p[part] + b * a.sb[part] + (long)y * a.sh[part] + (long)x * a.sw[part];
    for (int c = t >> 6; c < a.c[part]; c += 4) tile[coff + c][px] = ok ? base[(long)c * a.sc[part]] : 0.f;
    coff += a.c[part];
  }
  for (int c = C + (t >> 6); c < Cp; c += 4) tile[c][px] = 0.f;
  __syncthreads();
  const int nq = Cp >> 2;
  for (int idx = t; idx < 64 * nq; idx += 256) {
    const int p2 = idx / nq, q = idx - p2 * nq;
    const long g2 = (long)blockIdx.x * 64 + p2;
    if (g2 >= total) continue;
    float v[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = tile[4 * q + e][p2];
      l[e] = v[e] - (float)(__bf16)v[e];
    }
    const u32x2 hi = {bf16x2_bits(v[0], v[1]), bf16x2_bits(v[2], v[3])};
    const u32x2 lo = {bf16x2_bits(l[0], l[1]), bf16x2_bits(l[2], l[3])};
    uint16_t* o = out + g2 * 3 * Cp + 4 * q;
    *(u32x2*)o = hi;
    *(u32x2*)(o + Cp) = hi;
    *(u32x2*)(o + 2 * Cp) = lo;
  }
}

}  // namespace rfn

extern "C" int rfn_split3_cat_bf16(const float* const* parts, const long* strides, const int* channels, int nparts, void* out, int B,
                                   int H, int W, int Cp, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(parts && strides && channels && out && nparts >= 1 && nparts <= 4, "split3_cat_bf16: 1..4 parts");
  CatParts a;
  int C = 0;
  for (int i = 0; i < 4; ++i) {
    a.p[i] = nullptr;
    a.sb[i] = a.sc[i] = a.sh[i] = a.sw[i] = 0;
    a.c[i] = 0;
  }
  for (int i = 0; i < nparts; ++i) {
    RFN_REQUIRE(parts[i] && channels[i] > 0, "split3_cat_bf16: part %d", i);
    a.p[i] = parts[i];
    a.sb[i] = strides[4 * i];
    a.sc[i] = strides[4 * i + 1];
    a.sh[i] = strides[4 * i + 2];
    a.sw[i] = strides[4 * i + 3];
    a.c[i] = channels[i];
    C += channels[i];
  }
  a.n = nparts;
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && Cp >= C && Cp % 4 == 0 && Cp <= kCatMaxC, "split3_cat_bf16: B=%d H=%d W=%d C=%d Cp=%d (<= %d)",
              B, H, W, C, Cp, kCatMaxC);
  const long total = (long)B * H * W;
  hipLaunchKernelGGL(split3_cat_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a, (uint16_t*)out,
                     total, H, W, C, Cp);
  return check_launch("split3_cat_bf16");
}

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
