// refign_amd/csrc/refine.hip -- adaptive label correction ("refine") for gfx950.
//
// Reference: DomainAdaptationSegmentationModel.refine / .eta (models/segmentation_model.py:438-491).
// Pure HBM-bound elementwise work (3 x 19 x H x W floats of algorithmic traffic + mask/cert), two kernels:
//   1. refine_entropy_kernel : per-image sum of the normalised entropy of the target logits.  A fixed number of
//      blocks per image grid-strides the pixels and writes one double partial each (no float atomics => the trust
//      score s is bit-reproducible run to run).
//   2. refine_blend_kernel   : every block re-sums the image's partials in a fixed order, s = mean^gamma, then per
//      pixel: two softmaxes, two argmaxes, static-class mask M, eps = s*max(P,M) (0 where the warp is invalid),
//      out = (1-eps) p_trg + eps p_ref.  One thread per pixel; the 19 class planes are strided by HW so a wave's
//      loads/stores are contiguous along w.
#include "common.h"

namespace rfn {

constexpr int kClasses = 19;
constexpr int kPartials = 256;  // blocks (=partials) per image in the entropy pass

__device__ __forceinline__ bool is_static_large(int c) {
  // static_large_classes = [0,1,2,3,4,8,9,10]  (segmentation_model.py:452)
  return (c <= 4) || (c >= 8 && c <= 10);
}

__global__ __launch_bounds__(256) void refine_entropy_kernel(const float* __restrict__ logits,
                                                             double* __restrict__ partials, int HW) {
  const int n = blockIdx.y;
  const float* p = logits + (size_t)n * kClasses * HW;
  double local = 0.0;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < HW; pix += kPartials * 256) {
    float v[kClasses];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
      v[c] = p[(size_t)c * HW + pix];
      m = fmaxf(m, v[c]);
    }
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
      v[c] -= m;
      sum += expf(v[c]);
    }
    const float lsum = logf(sum);
    float ent = 0.0f;
#pragma unroll
    for (int c = 0; c < kClasses; ++c) {
      const float lp = v[c] - lsum;           // log_softmax
      ent -= (expf(v[c]) / sum) * lp;         // softmax * log_softmax (segmentation_model.py:488-490)
    }
    local += (double)ent;
  }
  // block reduce (4 waves)
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[(size_t)n * kPartials + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void refine_blend_kernel(const float* __restrict__ lt, const float* __restrict__ lr,
                                                           const unsigned char* __restrict__ wmask,
                                                           const float* __restrict__ certs, float* __restrict__ out,
                                                           const double* __restrict__ partials, int HW, float gamma,
                                                           int flags) {
  const int n = blockIdx.y;
  __shared__ double red[4];
  __shared__ float s_trust;
  {
    double v = partials[(size_t)n * kPartials + threadIdx.x];  // kPartials == blockDim.x
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double mean_ent = (red[0] + red[1] + red[2] + red[3]) / (double)HW;
      const float eta_mean = (float)(mean_ent / log((double)kClasses));   // eta(): ent / log(dim)
      s_trust = powf(eta_mean, gamma);                                    // segmentation_model.py:449
    }
    __syncthreads();
  }
  const float s = s_trust;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const float* pt = lt + (size_t)n * kClasses * HW + pix;
  const float* pr = lr + (size_t)n * kClasses * HW + pix;
  float a[kClasses], b[kClasses];
  float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
  for (int c = 0; c < kClasses; ++c) {
    a[c] = pt[(size_t)c * HW];
    b[c] = pr[(size_t)c * HW];
    ma = fmaxf(ma, a[c]);
    mb = fmaxf(mb, b[c]);
  }
  float sa = 0.0f, sb = 0.0f;
#pragma unroll
  for (int c = 0; c < kClasses; ++c) {
    a[c] = expf(a[c] - ma);
    b[c] = expf(b[c] - mb);
    sa += a[c];
    sb += b[c];
  }
  int ia = 0, ib = 0;
  float pa = -1.0f, pb = -1.0f;
#pragma unroll
  for (int c = 0; c < kClasses; ++c) {
    a[c] = a[c] / sa;
    b[c] = b[c] / sb;
    if (a[c] > pa) { pa = a[c]; ia = c; }   // first maximum, like torch.argmax
    if (b[c] > pb) { pb = b[c]; ib = c; }
  }
  const bool M = !(flags & 1) && is_static_large(ia) && is_static_large(ib);   // :453-464
  const float P = (certs != nullptr && !(flags & 2)) ? certs[(size_t)n * HW + pix] : 0.5f;   // :466-473
  const bool valid = (wmask == nullptr) || (wmask[(size_t)n * HW + pix] != 0);             // :477-479
  float* po = out + (size_t)n * kClasses * HW + pix;
#pragma unroll
  for (int c = 0; c < kClasses; ++c) {
    // M only applies to channels 0-4 and 8-10 (:460-461)
    const float mc = (M && is_static_large(c)) ? 1.0f : 0.0f;
    const float eps = valid ? s * fmaxf(P, mc) : 0.0f;                                       // :475
    po[(size_t)c * HW] = (1.0f - eps) * a[c] + eps * b[c];                                   // :481
  }
}

// Majority label of every scale x scale window with its share (models/segmentation_model.py:637-668,
// downscale_label_ratio: one-hot -> avg_pool2d -> max over classes): out = the most frequent class of the window (the
// smallest index among equals, as torch.max), or `ignore` when that class is the ignore bin or covers less than
// min_ratio of the window.  Windows at the bottom / right edge may be partial (avg_pool2d ceil_mode: the divisor is the
// number of pixels inside the image).  One wave per window, histogram in registers via ballot.
__global__ __launch_bounds__(64) void label_majority_kernel(const long* __restrict__ gt, long* __restrict__ out, int H,
                                                            int W, int oh, int ow, int scale, int n_classes, int ignore,
                                                            float min_ratio) {
  const int cell = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const int cy = cell / ow, cx = cell % ow;
  const int y0 = cy * scale, x0 = cx * scale;
  const int hh = min(scale, H - y0), ww = min(scale, W - x0);
  const long* g = gt + (size_t)n * H * W;
  const int nb = n_classes + 1;                        // bin n_classes = ignore
  int cnt[33];
#pragma unroll
  for (int c = 0; c < 33; ++c) cnt[c] = 0;
  for (int i = lane; i < hh * ww; i += 64) {
    const int y = y0 + i / ww, x = x0 + i % ww;
    long v = g[(size_t)y * W + x];
    const int bin = (v == ignore) ? n_classes : (int)v;
#pragma unroll
    for (int c = 0; c < 33; ++c) cnt[c] += (c < nb && bin == c) ? 1 : 0;
  }
  int best = 0, bestc = -1;
#pragma unroll
  for (int c = 0; c < 33; ++c) {
    int t = cnt[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (c < nb && t > bestc) { bestc = t; best = c; }
  }
  if (lane == 0) {
    const float ratio = (float)bestc / (float)(hh * ww);
    out[((size_t)n * oh + cy) * ow + cx] = (best == n_classes || ratio < min_ratio) ? (long)ignore : (long)best;
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" {

unsigned long rfn_refine_workspace_bytes(int B) { return (unsigned long)(B > 0 ? B : 0) * kPartials * sizeof(double); }

int rfn_refine_f32(const float* logits_trg, const float* logits_ref, const unsigned char* warp_mask,
                   const float* certs, float* out, void* workspace, int B, int C, int H, int W, float gamma,
                   int flags, rfn_stream_t stream) {
  RFN_REQUIRE(logits_trg && logits_ref && out && workspace, "rfn_refine_f32: null pointer");
  RFN_REQUIRE(C == kClasses, "rfn_refine_f32: we assume cityscapes classes (C must be 19, got %d)", C);
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && B <= 65535 && (long)H * W < 0x7fffffffL, "rfn_refine_f32: bad size");
  const int HW = H * W;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(refine_entropy_kernel, dim3(kPartials, B), dim3(256), 0, st, logits_trg, (double*)workspace,
                     HW);
  if (int rc = check_launch("refine_entropy_kernel")) return rc;
  hipLaunchKernelGGL(refine_blend_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, st, logits_trg, logits_ref,
                     warp_mask, certs, out, (const double*)workspace, HW, gamma, flags);
  return check_launch("refine_blend_kernel");
}

int rfn_label_majority(const long* gt, long* out, int B, int H, int W, int scale, int n_classes, int ignore_index,
                       float min_ratio, rfn_stream_t stream) {
  RFN_REQUIRE(gt && out, "rfn_label_majority: null pointer");
  RFN_REQUIRE(B > 0 && B <= 65535 && H > 0 && W > 0 && scale > 1 && n_classes > 0 && n_classes <= 32,
              "rfn_label_majority: B=%d H=%d W=%d scale=%d classes=%d (<= 32)", B, H, W, scale, n_classes);
  const int oh = rfn::cdiv(H, scale), ow = rfn::cdiv(W, scale);
  hipLaunchKernelGGL(rfn::label_majority_kernel, dim3(oh * ow, B), dim3(64), 0, (hipStream_t)stream, gt, out, H, W, oh, ow,
                     scale, n_classes, ignore_index, min_ratio);
  return rfn::check_launch("label_majority_kernel");
}

}  // extern "C"
