// refign_amd/csrc/dacs.hip -- N4: the GPU-side data step of a Refign UDA iteration: DACS class-mix of (source, target)
// images / labels / pixel weights, colour jitter and Gaussian blur of the mixed image, as HIP kernels.
//
// Reference: helpers/dacs_transforms.py:14-24 (strong_transform: one_mix, then kornia ColorJitter if the coin allows, then
// kornia GaussianBlur2d), :43-78 (colour jitter / blur with ImageNet de-normalisation around the jitter), :81-112 (class
// masks, one_mix), called per sample from models/segmentation_model.py:525-582.  kornia 0.5.8 is a third-party dependency
// absent from this image: its published operators are restated (brightness additive, contrast about the image mean,
// saturation about the luma, hue as a rotation of the chroma plane; blur = separable normalised Gaussian, reflect border),
// and the kernels are pinned against the torch formulation of the same operators in refign_amd/uda.py (tests/test_dacs_gpu.py).
//
// All three are HBM-bound pixel maps.  Per sample the host passes (by value, as kernel arguments -- no host-to-device copy,
// no synchronisation) the chosen-class bit set, the jitter operator order and factors drawn from the torch generator, the
// hue matrix and the blur sigma:
//   dacs_mix_jitter_kernel   pass 0: (only when contrast is in the chain) the per-image mean the contrast operator needs, of
//                            the image as it stands in front of that operator; pass 1: mix + the whole jitter chain, mixed
//                            label and mixed weight.  One thread = 4 consecutive pixels (16-byte loads per plane).
//   dacs_blur_kernel         one separable pass (vertical or horizontal) with taps |r| <= 16: for sigma <= 1.15 every tap
//                            beyond is below 2^-120 of the centre and does not change a float sum, however wide kornia's
//                            kernel window (0.1 x the image extent) is.
#include <algorithm>

#include "common.h"

namespace rfn {

constexpr int kDacsMaxBatch = 8;
constexpr int kBlurR = 16;

struct DacsSample {
  int jitter;              // 0: no colour jitter
  int order[4];            // operator applied k-th: 0 brightness, 1 contrast, 2 saturation, 3 hue
  float factor[4];         // per operator (hue: unused)
  float hue[9];            // 3x3 RGB matrix of the hue rotation
};
struct DacsArgs {
  DacsSample s[kDacsMaxBatch];
  float mean[3], stdv[3];  // ImageNet normalisation of the images
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// jitter chain on one [0,1] RGB pixel; `upto` < 0: whole chain, else stop IN FRONT of the operator with that code
__device__ __forceinline__ void jitter_chain(const DacsSample& sp, float img_mean, int upto, float& r, float& g, float& b) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int op = sp.order[k];
    if (op == upto) return;
    const float f = sp.factor[op];
    if (op == 0) {
      r += f - 1.f; g += f - 1.f; b += f - 1.f;
    } else if (op == 1) {
      r = (r - img_mean) * f + img_mean; g = (g - img_mean) * f + img_mean; b = (b - img_mean) * f + img_mean;
    } else if (op == 2) {
      const float gray = 0.299f * r + 0.587f * g + 0.114f * b;
      r = (r - gray) * f + gray; g = (g - gray) * f + gray; b = (b - gray) * f + gray;
    } else {
      const float nr = sp.hue[0] * r + sp.hue[1] * g + sp.hue[2] * b;
      const float ng = sp.hue[3] * r + sp.hue[4] * g + sp.hue[5] * b;
      const float nb = sp.hue[6] * r + sp.hue[7] * g + sp.hue[8] * b;
      r = nr; g = ng; b = nb;
    }
    r = clamp01(r); g = clamp01(g); b = clamp01(b);
  }
}

template <int PASS>
__global__ __launch_bounds__(256) void dacs_mix_jitter_kernel(const float* __restrict__ src, const float* __restrict__ trg,
                                                              const long* __restrict__ gt, const long* __restrict__ pseudo,
                                                              const float* __restrict__ pweight, float* __restrict__ img,
                                                              long* __restrict__ lbl, float* __restrict__ wgt,
                                                              double* __restrict__ msum, long plane,
                                                              const long* __restrict__ class_bits, DacsArgs a) {
  // class_bits[n] (device data: the class set comes from torch.unique on the device, no host round trip): bit c = class c
  // is taken from the source image, bit 31 = the ignore label 255 is
  const int n = blockIdx.y;
  const DacsSample& sp = a.s[n];
  const unsigned bits = (unsigned)class_bits[n];
  const long q = (long)blockIdx.x * 256 + threadIdx.x;          // quad of 4 pixels
  const long p0 = 4 * q;
  float part = 0.f;
  if (p0 < plane) {
    const long* g4 = gt + n * plane + p0;
    float c[3][4];
    bool m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long l = g4[i];
      m[i] = ((l >= 0 && l < 31) ? (bits >> l) & 1u : (l == 255 ? bits >> 31 : 0u)) != 0u;
    }
    // img == nullptr: labels / weights only (the image half was mixed earlier in the step, with the same class set);
    // lbl == nullptr: image only (the pseudo-labels do not exist yet)
    if (img != nullptr) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float4 s4 = *reinterpret_cast<const float4*>(src + (n * 3 + ch) * plane + p0);
      const float4 t4 = *reinterpret_cast<const float4*>(trg + (n * 3 + ch) * plane + p0);
      const float sv[4] = {s4.x, s4.y, s4.z, s4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float mf = m[i] ? 1.f : 0.f;
        c[ch][i] = mf * sv[i] + (1.f - mf) * tv[i];               // one_mix's own arithmetic (mask * a + (1 - mask) * b)
      }
    }
    if (sp.jitter) {
      const float mu = PASS == 1 ? (float)(msum[n] / (double)(3 * plane)) : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float r = c[0][i] * a.stdv[0] + a.mean[0], g = c[1][i] * a.stdv[1] + a.mean[1], b = c[2][i] * a.stdv[2] + a.mean[2];
        jitter_chain(sp, mu, PASS == 0 ? 1 : -1, r, g, b);
        if (PASS == 0) part += (r + g) + b;
        c[0][i] = (r - a.mean[0]) / a.stdv[0]; c[1][i] = (g - a.mean[1]) / a.stdv[1]; c[2][i] = (b - a.mean[2]) / a.stdv[2];
      }
    }
    if (PASS == 1) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch)
        *reinterpret_cast<float4*>(img + (n * 3 + ch) * plane + p0) = make_float4(c[ch][0], c[ch][1], c[ch][2], c[ch][3]);
    }
    }
    if (PASS == 1 && lbl != nullptr) {
      const long* ps = pseudo + n * plane + p0;
      const float4 w4 = *reinterpret_cast<const float4*>(pweight + n * plane + p0);
      const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
      long lo[4];
      float wo[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = m[i] ? g4[i] : ps[i];
        wo[i] = m[i] ? 1.f : wv[i];
      }
      long* lp = lbl + n * plane + p0;
#pragma unroll
      for (int i = 0; i < 4; ++i) lp[i] = lo[i];
      *reinterpret_cast<float4*>(wgt + n * plane + p0) = make_float4(wo[0], wo[1], wo[2], wo[3]);
    }
  }
  if (PASS == 0) {
    __shared__ float red[4];
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(msum + n, (double)((red[0] + red[1]) + (red[2] + red[3])));
  }
}

__device__ __forceinline__ int reflect(int i, int n) {          // torch 'reflect' padding (no edge repeat)
  if (n == 1) return 0;
  const int period = 2 * (n - 1);
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}

struct BlurArgs {
  float w[kDacsMaxBatch][2 * kBlurR + 1];   // normalised taps per sample (host: float64 -> float32)
  int on[kDacsMaxBatch];
};

// VERT: taps along H, else along W.  One thread = one output pixel of one (sample, channel) plane.
template <bool VERT>
__global__ __launch_bounds__(256) void dacs_blur_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W,
                                                        BlurArgs a) {
  const int nc = blockIdx.z, n = nc / C;
  const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (px >= W || py >= H) return;
  const float* p = x + (long)nc * H * W;
  float acc = 0.f;
  if (!a.on[n]) {
    acc = p[(long)py * W + px];
  } else {
#pragma unroll 11
    for (int t = -kBlurR; t <= kBlurR; ++t) {
      const float v = VERT ? p[(long)reflect(py + t, H) * W + px] : p[(long)py * W + reflect(px + t, W)];
      acc = fmaf(a.w[n][t + kBlurR], v, acc);
    }
  }
  y[(long)nc * H * W + (long)py * W + px] = acc;
}

}  // namespace rfn

extern "C" {

int rfn_dacs_mix_jitter(const float* src, const float* trg, const long* gt_src, const long* pseudo_label,
                        const float* pseudo_weight, float* mixed_img, long* mixed_lbl, float* mixed_weight, double* mean_ws,
                        int B, int H, int W, const long* class_bits, const int* jitter_on, const int* order,
                        const float* factor, const float* hue, const float* mean3, const float* std3,
                        rfn_stream_t stream) {
  using namespace rfn;
  // either half may be left out: mixed_img == NULL -> labels / weights only (src / trg unused), mixed_lbl == NULL -> image only
  // (pseudo_label / pseudo_weight / mixed_weight unused); the masks of the two halves agree when gt_src / class_bits do
  RFN_REQUIRE(gt_src && mean_ws && (mixed_img || mixed_lbl), "dacs_mix_jitter: null pointer");
  RFN_REQUIRE(!mixed_img || (src && trg), "dacs_mix_jitter: image half without src / trg");
  RFN_REQUIRE(!mixed_lbl || (pseudo_label && pseudo_weight && mixed_weight), "dacs_mix_jitter: label half without pseudo-labels");
  RFN_REQUIRE(B > 0 && B <= kDacsMaxBatch && H > 0 && W > 0 && ((long)H * W) % 4 == 0, "dacs_mix_jitter: B=%d (<= %d) H=%d W=%d "
              "(H*W %% 4)", B, kDacsMaxBatch, H, W);
  RFN_REQUIRE(class_bits && jitter_on && order && factor && hue && mean3 && std3, "dacs_mix_jitter: null parameter array");
  DacsArgs a{};
  bool need_mean = false;
  for (int n = 0; n < B; ++n) {
    a.s[n].jitter = jitter_on[n];
    for (int k = 0; k < 4; ++k) {
      RFN_REQUIRE(order[4 * n + k] >= 0 && order[4 * n + k] < 4, "dacs_mix_jitter: operator code");
      a.s[n].order[k] = order[4 * n + k];
      a.s[n].factor[k] = factor[4 * n + k];
    }
    for (int k = 0; k < 9; ++k) a.s[n].hue[k] = hue[9 * n + k];
    need_mean = need_mean || (jitter_on[n] && mixed_img != nullptr);
  }
  for (int k = 0; k < 3; ++k) {
    a.mean[k] = mean3[k];
    a.stdv[k] = std3[k];
  }
  const long plane = (long)H * W;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(plane / 4, 256), (unsigned)B), block(256);
  if (need_mean) {
    if (int rc = zero_async(mean_ws, sizeof(double) * kDacsMaxBatch, st)) return rc;
    hipLaunchKernelGGL((dacs_mix_jitter_kernel<0>), grid, block, 0, st, src, trg, gt_src, pseudo_label, pseudo_weight,
                       mixed_img, mixed_lbl, mixed_weight, mean_ws, plane, class_bits, a);
  }
  hipLaunchKernelGGL((dacs_mix_jitter_kernel<1>), grid, block, 0, st, src, trg, gt_src, pseudo_label, pseudo_weight, mixed_img,
                     mixed_lbl, mixed_weight, mean_ws, plane, class_bits, a);
  return check_launch("dacs_mix_jitter_kernel");
}

int rfn_dacs_blur(const float* x, float* tmp, float* y, int B, int C, int H, int W, int ksize_y, int ksize_x, const int* blur_on,
                  const double* sigma_y, const double* sigma_x, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(x && tmp && y && blur_on && sigma_y && sigma_x, "dacs_blur: null pointer");
  RFN_REQUIRE(B > 0 && B <= kDacsMaxBatch && C > 0 && H > kBlurR && W > kBlurR, "dacs_blur: B=%d C=%d H=%d W=%d", B, C, H, W);
  RFN_REQUIRE(ksize_y > 0 && ksize_x > 0 && (ksize_y & 1) && (ksize_x & 1), "dacs_blur: odd window sizes (got %d, %d)", ksize_y,
              ksize_x);
  BlurArgs ay{}, ax{};
  for (int n = 0; n < B; ++n) {
    ay.on[n] = ax.on[n] = blur_on[n];
    if (!blur_on[n]) continue;
    RFN_REQUIRE(sigma_y[n] > 0 && sigma_y[n] <= 1.25 && sigma_x[n] > 0 && sigma_x[n] <= 1.25,
                "dacs_blur: sigma outside (0, 1.25]: the 33-tap window would truncate the kernel");
    for (int pass = 0; pass < 2; ++pass) {
      // kornia's window is ksize taps wide and normalised over the window; taps beyond +-16 are zero in fp32 for these sigmas
      const double s = pass ? sigma_x[n] : sigma_y[n];
      const int rad = std::min(kBlurR, (pass ? ksize_x : ksize_y) / 2);
      double g[2 * kBlurR + 1], sum = 0.0;
      for (int t = -kBlurR; t <= kBlurR; ++t) sum += (g[t + kBlurR] = (t < -rad || t > rad) ? 0.0 : exp(-(double)t * t / (2.0 * s * s)));
      for (int t = 0; t <= 2 * kBlurR; ++t) (pass ? ax : ay).w[n][t] = (float)(g[t] / sum);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)cdiv(W, 64), (unsigned)cdiv(H, 4), (unsigned)(B * C)), block(256);
  hipLaunchKernelGGL((dacs_blur_kernel<true>), grid, block, 0, st, x, tmp, C, H, W, ay);
  hipLaunchKernelGGL((dacs_blur_kernel<false>), grid, block, 0, st, (const float*)tmp, y, C, H, W, ax);
  return check_launch("dacs_blur_kernel");
}

}  // extern "C"
