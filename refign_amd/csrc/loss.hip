// refign_amd/csrc/loss.hip -- the segmentation loss of the student passes in one kernel: bilinear up-sampling of the class
// logits to the label resolution, pixel-weighted cross-entropy, and the gradient with respect to the LOW-resolution logits.
//
// Reference: models/segmentation_model.py:163-179 / :226-250 (source and mixed pass): `F.interpolate(logits, size,
// mode='bilinear', align_corners=False)` followed by models/losses.py:10-22 PixelWeightedCrossEntropyLoss =
// mean over ALL pixels of weight * CE(ignore_index).  Unfused, the (B, 19, 1080, 1920) tensor (158 MB in fp32) is written
// by the up-sampling, read and written by log_softmax, read by nll_loss, and the same again backwards -- 1.3 ms per
// student pass in six ATen kernels, between the forward and the backward of the pass (round 3, tools/aten_census.py).
// Here a workgroup owns a 16 x 16 tile of label pixels (32 KB of LDS: 4-5 workgroups per CU):
//   1. the low-resolution logits under the tile (its bilinear footprint, <= 10 x 10 cells x C) go to LDS;
//   2. every pixel interpolates its C logits (ATen's formula and operand order, optionally rounded to the 16-bit dtype
//      the unfused path would have stored), takes the log-sum-exp, adds weight * (lse - z[target]) to the loss and leaves
//      weight * (softmax - onehot) -- its gradient with respect to the up-sampled logits -- in LDS;
//   3. the transpose of the interpolation, separably: along x into (C, 16 rows, footprint columns), then along y, and the
//      footprint cells are ADDED to the low-resolution gradient (fp32 atomics: neighbouring tiles share their border
//      cells).
// Results are the SUMS (loss: one double; gradient: unscaled); the host divides by the pixel count and multiplies by the
// upstream gradient.  Scale factors >= 2 in both directions (the footprint bound above); anything else stays on ATen.
#include <hip/hip_bf16.h>

#include "common.h"
#include "mfma.h"

namespace rfn {

constexpr int kLossTH = 16, kLossTW = 16, kLossPX = kLossTH * kLossTW;     // label pixels per workgroup (one per thread)
constexpr int kLossFY = kLossTH / 2 + 2, kLossFX = kLossTW / 2 + 2;         // footprint bound at scale 2
constexpr int kLossSlots = 64;                                              // partial loss sums
constexpr int kLossMaxC = 19;                                               // classes (Cityscapes); LDS is sized for it

template <int DT> struct LossElem;
template <> struct LossElem<0> {
  using T = float;
  static __device__ __forceinline__ float ld(const void* p, long i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct LossElem<1> {
  using T = __bf16;
  static __device__ __forceinline__ float ld(const void* p, long i) { return (float)((const __bf16*)p)[i]; }
  static __device__ __forceinline__ float round(float v) { return (float)(__bf16)v; }
};
template <> struct LossElem<2> {
  using T = _Float16;
  static __device__ __forceinline__ float ld(const void* p, long i) { return (float)((const _Float16*)p)[i]; }
  static __device__ __forceinline__ float round(float v) { return (float)(_Float16)v; }
};

// ATen area_pixel_compute_source_index (align_corners = false, not cubic) + the neighbour / lambda of upsample_bilinear2d
__device__ __forceinline__ void src_index(int dst, float scale, int in, int& i0, int& i1, float& l1) {
  const float s = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  i0 = min((int)s, in - 1);
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

// FULLC: C == kLossMaxC, no per-class guards (with a run-time class count every class of every loop is a branch)
template <int DT, bool FULLC>
__global__ __launch_bounds__(256) void upsample_ce_kernel(const void* __restrict__ logits, const long* __restrict__ target,
                                                          const float* __restrict__ weight, float* __restrict__ grad_lo,
                                                          double* __restrict__ loss_sum, int C, int h, int w, int H, int W,
                                                          float sy, float sx, int ignore_index, int round16) {
  using E = LossElem<DT>;
  // the footprint logits (steps 1-2) and the x-reduced gradient (step 3) share their LDS
  __shared__ float tmp[kLossMaxC * kLossTH * kLossFX];         // [c][ty][fx]
  float* const lo = tmp;                                       // [c][fy][fx], dead before step 3a writes tmp
  __shared__ float pr[kLossMaxC * kLossPX];                    // [c][ty * 32 + tx]: gradient w.r.t. the up-sampled logits
  __shared__ int ry0[kLossTH], ry1[kLossTH], cx0[kLossTW], cx1[kLossTW];
  __shared__ float rl[kLossTH], cl[kLossTW];
  __shared__ float red[4];
  __shared__ int txlo[kLossFX], txhi[kLossFX], tylo[kLossFY], tyhi[kLossFY];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int Y0 = blockIdx.y * kLossTH, X0 = blockIdx.x * kLossTW;
  const int YN = min(kLossTH, H - Y0), XN = min(kLossTW, W - X0);       // live rows / columns of the tile
  if (tid < kLossTH) {
    int i0, i1;
    float l;
    src_index(min(Y0 + tid, H - 1), sy, h, i0, i1, l);
    ry0[tid] = i0;
    ry1[tid] = i1;
    rl[tid] = l;
  } else if (tid >= 64 && tid < 64 + kLossTW) {
    const int t = tid - 64;
    int i0, i1;
    float l;
    src_index(min(X0 + t, W - 1), sx, w, i0, i1, l);
    cx0[t] = i0;
    cx1[t] = i1;
    cl[t] = l;
  }
  __syncthreads();
  const int fy0 = ry0[0], fx0 = cx0[0];
  const int nfy = ry1[YN - 1] - fy0 + 1, nfx = cx1[XN - 1] - fx0 + 1;   // <= kLossFY, kLossFX for scales >= 2 (host checks)
  const long plane = (long)h * w;
  const char* lg = (const char*)logits;
  for (int i = tid; i < C * nfy * nfx; i += 256) {
    const int c = i / (nfy * nfx), r = i - c * nfy * nfx, fy = r / nfx, fx = r - fy * nfx;
    lo[(c * kLossFY + fy) * kLossFX + fx] = E::ld(lg, ((long)b * C + c) * plane + (long)(fy0 + fy) * w + fx0 + fx);
  }
  __syncthreads();
  // ---- 2. per pixel
  float lsum = 0.f;
#pragma unroll
  for (int k = 0; k < kLossPX / 256; ++k) {
    const int px = tid + 256 * k, ty = px / kLossTW, tx = px % kLossTW;
    const bool live = ty < YN && tx < XN;
    float z[kLossMaxC];
    float m = -3.0e38f;
    const int a0 = (ry0[ty] - fy0) * kLossFX, a1 = (ry1[ty] - fy0) * kLossFX, b0 = cx0[tx] - fx0, b1 = cx1[tx] - fx0;
    const float ly = rl[ty], hy = 1.f - ly, lx = cl[tx], hx = 1.f - lx;
#pragma unroll
    for (int c = 0; c < kLossMaxC; ++c) {
      if (FULLC || c < C) {
        const float* p = lo + c * kLossFY * kLossFX;
        float v = hy * (hx * p[a0 + b0] + lx * p[a0 + b1]) + ly * (hx * p[a1 + b0] + lx * p[a1 + b1]);
        if (round16) v = E::round(v);
        z[c] = v;
        m = fmaxf(m, v);
      }
    }
    long t = ignore_index;
    float wt = 0.f;
    if (live) {
      const long gi = ((long)b * H + Y0 + ty) * W + X0 + tx;
      t = target[gi];
      wt = weight != nullptr ? weight[gi] : 1.f;
    }
    const bool valid = live && t != ignore_index && t >= 0 && t < C;
    float S = 0.f, zt = 0.f;
#pragma unroll
    for (int c = 0; c < kLossMaxC; ++c) {
      if (FULLC || c < C) {
        if (valid && c == (int)t) zt = z[c];
        z[c] = __expf(z[c] - m);
        S += z[c];
      }
    }
    const float inv = valid ? wt / S : 0.f;
#pragma unroll
    for (int c = 0; c < kLossMaxC; ++c) {
      if (FULLC || c < C) pr[c * kLossPX + px] = z[c] * inv - ((valid && c == (int)t) ? wt : 0.f);
    }
    if (valid) lsum += wt * (m + __logf(S) - zt);                  // weight * (lse - z[target])
  }
  lsum = wave_sum(lsum);
  if ((tid & 63) == 0) red[tid >> 6] = lsum;
  __syncthreads();
  // kLossSlots partial sums (the host adds them): 8 160 tiles adding to ONE address were serialised in the L2 -- 435 us of
  // same-address atomics around 70 us of work
  if (tid == 0)
    atomicAdd(loss_sum + (blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) % kLossSlots,
              (double)((red[0] + red[1]) + (red[2] + red[3])));
  // the tile columns / rows whose footprint contains a cell are a contiguous run (the source index is monotone): its ends,
  // so that the sums below walk ~2 x scale pixels instead of the whole tile edge
  if (tid < nfx) {
    const int cell = fx0 + tid;
    int lo_ = XN, hi_ = -1;
    for (int tx = 0; tx < XN; ++tx)
      if (cx0[tx] == cell || cx1[tx] == cell) { lo_ = min(lo_, tx); hi_ = tx; }
    txlo[tid] = lo_;
    txhi[tid] = hi_;
  } else if (tid >= 64 && tid < 64 + nfy) {
    const int f = tid - 64, cell = fy0 + f;
    int lo_ = YN, hi_ = -1;
    for (int ty = 0; ty < YN; ++ty)
      if (ry0[ty] == cell || ry1[ty] == cell) { lo_ = min(lo_, ty); hi_ = ty; }
    tylo[f] = lo_;
    tyhi[f] = hi_;
  }
  __syncthreads();
  // ---- 3a. transpose of the interpolation along x: tmp[c][ty][fx] = sum_tx wx(tx, fx) pr[c][ty][tx]
  for (int i = tid; i < C * kLossTH * nfx; i += 256) {
    const int c = i / (kLossTH * nfx), r = i - c * kLossTH * nfx, ty = r / nfx, fx = r - ty * nfx;
    float acc = 0.f;
    if (ty < YN) {
      const float* p = pr + c * kLossPX + ty * kLossTW;
      const int cell = fx0 + fx;
      for (int tx = txlo[fx]; tx <= txhi[fx]; ++tx) {
        const float l = cl[tx];
        const float wgt = (cx0[tx] == cell ? 1.f - l : 0.f) + (cx1[tx] == cell ? l : 0.f);
        acc = fmaf(wgt, p[tx], acc);
      }
    }
    tmp[(c * kLossTH + ty) * kLossFX + fx] = acc;
  }
  __syncthreads();
  // ---- 3b. along y, and out
  for (int i = tid; i < C * nfy * nfx; i += 256) {
    const int c = i / (nfy * nfx), r = i - c * nfy * nfx, fy = r / nfx, fx = r - fy * nfx;
    const int cell = fy0 + fy;
    float acc = 0.f;
    for (int ty = tylo[fy]; ty <= tyhi[fy]; ++ty) {
      const float l = rl[ty];
      const float wgt = (ry0[ty] == cell ? 1.f - l : 0.f) + (ry1[ty] == cell ? l : 0.f);
      acc = fmaf(wgt, tmp[(c * kLossTH + ty) * kLossFX + fx], acc);
    }
    if (acc != 0.f) atomicAdd(grad_lo + ((long)b * C + c) * plane + (long)cell * w + fx0 + fx, acc);
  }
}

}  // namespace rfn

extern "C" {
using namespace rfn;

// loss_sum[0..63] <- 64 partial sums (their sum is the result) over all (b, y, x) of weight * CE(bilinear(logits)[b, :, y, x], target), grad_lo (B, C, h, w) fp32 <- the
// gradient of that sum with respect to the low-resolution logits; both are zeroed here.  logits: (B, C, h, w) contiguous,
// dtype 0 fp32 / 1 bf16 / 2 f16; target: (B, H, W) int64; weight: (B, H, W) fp32 or NULL.  round16: round the interpolated
// logits to `dtype` first (what an unfused 16-bit up-sampling stores).
int rfn_upsample_ce(const void* logits, const long* target, const float* weight, float* grad_lo, double* loss_sum, int B,
                    int C, int h, int w, int H, int W, int ignore_index, int dtype, int round16, rfn_stream_t stream) {
  RFN_REQUIRE(logits && target && grad_lo && loss_sum, "upsample_ce: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && C <= kLossMaxC && h > 0 && w > 0, "upsample_ce: B=%d C=%d (<= %d) h=%d w=%d", B, C, kLossMaxC, h, w);
  RFN_REQUIRE(H >= 2 * h && W >= 2 * w, "upsample_ce: %dx%d -> %dx%d (scale factors >= 2 only)", h, w, H, W);
  RFN_REQUIRE(dtype >= 0 && dtype <= 2, "upsample_ce: dtype %d (0 = f32, 1 = bf16, 2 = f16)", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero_async(grad_lo, (size_t)B * C * h * w * sizeof(float), s)) return rc;
  if (int rc = zero_async(loss_sum, kLossSlots * sizeof(double), s)) return rc;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;      // ATen: area_pixel_compute_scale with size=
  dim3 grid(cdiv(W, kLossTW), cdiv(H, kLossTH), B);
#define RFN_UCE(D)                                                                                                     \
  if (C == kLossMaxC)                                                                                                  \
    hipLaunchKernelGGL((upsample_ce_kernel<D, true>), grid, dim3(256), 0, s, logits, target, weight, grad_lo, loss_sum, \
                       C, h, w, H, W, sy, sx, ignore_index, round16 && D != 0);                                         \
  else                                                                                                                  \
    hipLaunchKernelGGL((upsample_ce_kernel<D, false>), grid, dim3(256), 0, s, logits, target, weight, grad_lo, loss_sum, \
                       C, h, w, H, W, sy, sx, ignore_index, round16 && D != 0)
  if (dtype == 0) RFN_UCE(0);
  else if (dtype == 1) RFN_UCE(1);
  else RFN_UCE(2);
#undef RFN_UCE
  return check_launch("upsample_ce");
}

}  // extern "C"
