// refign_amd/csrc/warp.hip -- bilinear warp kernels (gfx950).
//
// Reference behaviour restated from scratch:
//   warp()            helpers/matching_utils.py:11-49  (grid = base + flow, normalise with max(W-1,1),
//                      grid_sample bilinear / align_corners=True / zeros; mask = strictly inside (-1,1)^2)
//   tail of align()   models/segmentation_model.py:514-522 (bilinear align_corners=False upsampling of the
//                      quarter-res flow and log-variance, confidence, warp of the reference logits)
//
// HBM-bound gathers: one thread per output pixel computes the four taps once and streams the channels, so every
// global access of a wave is contiguous along w for the store and near-contiguous (smooth flow) for the loads.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include "common.h"

namespace rfn {

struct Tap {
  int o00, o01, o10, o11;
  float w00, w01, w10, w11;
  bool inside;  // strict-inequality validity of the NORMALISED coordinate (matching_utils.py:46-47)
};

__device__ __forceinline__ Tap bilinear_tap(float gx, float gy, float fx, float fy, int H, int W) {
#pragma clang fp contract(off)
  const float wm = (float)max(W - 1, 1), hm = (float)max(H - 1, 1);
  const float vx = 2.0f * (gx + fx) / wm - 1.0f;               // matching_utils.py:35
  const float vy = 2.0f * (gy + fy) / hm - 1.0f;               // matching_utils.py:36
  const float ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);      // ATen grid_sampler_unnormalize(align_corners)
  const float iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
  float x0f = floorf(ix), y0f = floorf(iy);
  const float tx = ix - x0f, ty = iy - y0f;
  x0f = fminf(fmaxf(x0f, -2.0f), (float)W);
  y0f = fminf(fmaxf(y0f, -2.0f), (float)H);
  if (!(ix == ix) || !(iy == iy)) { x0f = -2.0f; y0f = -2.0f; }
  const int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  Tap t;
  t.o00 = cy0 * W + cx0; t.o01 = cy0 * W + cx1; t.o10 = cy1 * W + cx0; t.o11 = cy1 * W + cx1;
  const float ax = 1.0f - tx, ay = 1.0f - ty;
  t.w00 = (vx0 && vy0) ? ax * ay : 0.0f;
  t.w01 = (vx1 && vy0) ? tx * ay : 0.0f;
  t.w10 = (vx0 && vy1) ? ax * ty : 0.0f;
  t.w11 = (vx1 && vy1) ? tx * ty : 0.0f;
  t.inside = (vx > -1.0f) && (vy > -1.0f) && (vx < 1.0f) && (vy < 1.0f);
  return t;
}

__device__ __forceinline__ float tap_sample(const float* __restrict__ src, const Tap& t) {
#pragma clang fp contract(off)
  // ATen grid_sampler_2d accumulation order: nw, ne, sw, se
  return src[t.o00] * t.w00 + src[t.o01] * t.w01 + src[t.o10] * t.w10 + src[t.o11] * t.w11;
}

// grid: (ceil(HW/256), channel groups, B)
template <int CG>
__global__ __launch_bounds__(256) void warp_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                   float* __restrict__ out, unsigned char* __restrict__ mask,
                                                   int C, int H, int W) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int n = blockIdx.z, c0 = blockIdx.y * CG;
  const int gy = pix / W, gx = pix - gy * W;
  const float* fl = flow + (size_t)n * 2 * HW;
  const Tap t = bilinear_tap((float)gx, (float)gy, fl[pix], fl[HW + pix], H, W);
  if (mask != nullptr && blockIdx.y == 0) mask[(size_t)n * HW + pix] = t.inside ? 1 : 0;
  const float* src = x + ((size_t)n * C + c0) * HW;
  float* dst = out + ((size_t)n * C + c0) * HW + pix;
  const int cend = min(CG, C - c0);
  for (int c = 0; c < cend; ++c) dst[(size_t)c * HW] = tap_sample(src + (size_t)c * HW, t);
}

// Backward of warp() = backward of grid_sample(bilinear, align_corners=True, zeros) composed with the flow -> grid map
// (ATen grid_sampler_2d_backward: gix = sum_c g [ (ne - nw) (1 - ty) + (se - sw) ty ], giy likewise; taps outside the
// image contribute zero).  d ix / d flow_x = ((W - 1) / 2) (2 / max(W - 1, 1)): 1 for W > 1, 0 for a one-pixel axis.
// One thread per pixel and channel group: grad_x is a scatter (float atomics, pre-zeroed), grad_flow the sum over the
// channel groups (atomics on 2 floats per pixel, pre-zeroed).
// grid: (ceil(HW/256), channel groups, B)
template <int CG>
__global__ __launch_bounds__(256) void warp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ flow,
                                                       const float* __restrict__ gout, float* __restrict__ gx_out,
                                                       float* __restrict__ gflow, int C, int H, int W) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W;
  if (pix >= HW) return;
  const int n = blockIdx.z, c0 = blockIdx.y * CG;
  const int gy = pix / W, gx = pix - gy * W;
  const float* fl = flow + (size_t)n * 2 * HW;
  const Tap t = bilinear_tap((float)gx, (float)gy, fl[pix], fl[HW + pix], H, W);
  // fractional parts and tap validity, recovered from the weights' construction
  const float fxv = fl[pix], fyv = fl[HW + pix];
  const float wm = (float)max(W - 1, 1), hm = (float)max(H - 1, 1);
  float ix, iy;
  {
#pragma clang fp contract(off)
    const float vx = 2.0f * ((float)gx + fxv) / wm - 1.0f, vy = 2.0f * ((float)gy + fyv) / hm - 1.0f;
    ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);
    iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
  }
  const float x0f = floorf(ix), y0f = floorf(iy);
  const float tx = ix - x0f, ty = iy - y0f, ax = 1.0f - tx, ay = 1.0f - ty;
  const bool finite = (ix == ix) && (iy == iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
  const int x0 = finite ? (int)x0f : -2, y0 = finite ? (int)y0f : -2;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const bool vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
  const bool v00 = vx0 && vy0, v01 = vx1 && vy0, v10 = vx0 && vy1, v11 = vx1 && vy1;
  const float* src = x + ((size_t)n * C + c0) * HW;
  const float* go = gout + ((size_t)n * C + c0) * HW + pix;
  float* gxp = gx_out ? gx_out + ((size_t)n * C + c0) * HW : nullptr;
  const int cend = min(CG, C - c0);
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < cend; ++c) {
    const float g = go[(size_t)c * HW];
    const float* s = src + (size_t)c * HW;
    const float nw = v00 ? s[t.o00] : 0.f, ne = v01 ? s[t.o01] : 0.f, sw = v10 ? s[t.o10] : 0.f, se = v11 ? s[t.o11] : 0.f;
    gix += g * ((ne - nw) * ay + (se - sw) * ty);
    giy += g * ((sw - nw) * ax + (se - ne) * tx);
    if (gxp) {
      float* d = gxp + (size_t)c * HW;
      if (v00) atomicAdd(d + t.o00, g * t.w00);
      if (v01) atomicAdd(d + t.o01, g * t.w01);
      if (v10) atomicAdd(d + t.o10, g * t.w10);
      if (v11) atomicAdd(d + t.o11, g * t.w11);
    }
  }
  if (gflow) {
    float* gf = gflow + (size_t)n * 2 * HW;
    if (W > 1 && finite) atomicAdd(gf + pix, gix);
    if (H > 1 && finite) atomicAdd(gf + HW + pix, giy);
  }
}

// ATen upsample_bilinear2d source index, align_corners=False: max(scale*(dst+0.5)-0.5, 0)
__device__ __forceinline__ void up_index(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.0f ? 0.0f : s;
  i0 = min((int)s, in_size - 1);
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - (float)i0;
  l0 = 1.0f - l1;
}

__device__ __forceinline__ float up_sample(const float* __restrict__ p, int w, int y0, int y1, int x0, int x1,
                                           float ly0, float ly1, float lx0, float lx1) {
#pragma clang fp contract(off)
  return ly0 * (lx0 * p[y0 * w + x0] + lx1 * p[y0 * w + x1]) + ly1 * (lx0 * p[y1 * w + x0] + lx1 * p[y1 * w + x1]);
}

// grid: (ceil(HW/256), 1, B)
__global__ __launch_bounds__(256) void align_tail_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ flow_q,
                                                         const float* __restrict__ logvar_q,
                                                         float* __restrict__ warped, unsigned char* __restrict__ mask,
                                                         float* __restrict__ cert, float* __restrict__ flow_up, int C,
                                                         int H, int W, int h, int w, float sy, float sx) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int HW = H * W, hw = h * w;
  if (pix >= HW) return;
  const int n = blockIdx.z;
  const int gy = pix / W, gx = pix - gy * W;
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  up_index(gy, sy, h, y0, y1, ly0, ly1);
  up_index(gx, sx, w, x0, x1, lx0, lx1);
  const float* fq = flow_q + (size_t)n * 2 * hw;
  const float fx = up_sample(fq, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
  const float fy = up_sample(fq + hw, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
  const float lv = up_sample(logvar_q + (size_t)n * hw, w, y0, y1, x0, x1, ly0, ly1, lx0, lx1);
  // matching_utils.py:55-56 with R = 1
  cert[(size_t)n * HW + pix] = 1.0f - expf(-1.0f / (2.0f * expf(lv)));
  if (flow_up != nullptr) {
    flow_up[(size_t)n * 2 * HW + pix] = fx;
    flow_up[(size_t)n * 2 * HW + HW + pix] = fy;
  }
  const Tap t = bilinear_tap((float)gx, (float)gy, fx, fy, H, W);
  mask[(size_t)n * HW + pix] = t.inside ? 1 : 0;
  const float* src = logits + (size_t)n * C * HW;
  float* dst = warped + (size_t)n * C * HW + pix;
  for (int c = 0; c < C; ++c) dst[(size_t)c * HW] = tap_sample(src + (size_t)c * HW, t);
}

// F.normalize(p=2, dim=1) for NCHW.  Generic form: one thread per pixel, channels strided by HW (coalesced across the
// wave), two passes over the channels.
__global__ __launch_bounds__(256) void l2norm_channels_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int C, int HW) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= HW) return;
  const float* p = x + (size_t)blockIdx.y * C * HW + pix;
  float* o = out + (size_t)blockIdx.y * C * HW + pix;
  float ss = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float v = p[(size_t)c * HW];
    ss = fmaf(v, v, ss);
  }
  const float d = fmaxf(sqrtf(ss), 1e-12f);
  for (int c = 0; c < C; ++c) o[(size_t)c * HW] = p[(size_t)c * HW] / d;
}

// The VGG pyramid widths (C = SL * CPT = 128, 256, 512): a workgroup is (256/SL pixels) x (SL channel slices), a thread
// keeps its CPT channels of one pixel in registers -- ONE pass over HBM, CPT independent loads in flight per thread
// (the generic kernel is a serial chain of C dependent-latency loads on 250 workgroups at level 2: 0.95 TB/s).
template <int SL, int CPT>
__global__ __launch_bounds__(256) void l2norm_channels_reg_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                  int HW) {
  constexpr int PX = 256 / SL;
  __shared__ float part[SL][PX];
  const int lane = threadIdx.x % PX, sl = threadIdx.x / PX;
  const int pix = blockIdx.x * PX + lane;
  const bool ok = pix < HW;
  const size_t base = ((size_t)blockIdx.y * SL * CPT + (size_t)sl * CPT) * HW + (ok ? pix : 0);
  float v[CPT];
  float ss = 0.0f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) v[k] = ok ? x[base + (size_t)k * HW] : 0.0f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) ss = fmaf(v[k], v[k], ss);
  part[sl][lane] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int t = 0; t < SL; ++t) tot += part[t][lane];
  const float d = fmaxf(sqrtf(tot), 1e-12f);
  if (ok) {
#pragma unroll
    for (int k = 0; k < CPT; ++k) out[base + (size_t)k * HW] = v[k] / d;
  }
}

// The same from the layout and precision the matcher's convolutions deliver under the reference's AMP recipe:
// channels-last 16-bit features (B, HW, C) -> NCHW fp32, L2-normalised over the channels.  Replaces a cast kernel, a
// strided NHWC -> NCHW copy (1.1 ms for 2 x 128 x 270 x 480: 0.25 TB/s) and the NCHW normalisation with one pass: a
// workgroup stages 32 pixels x C channels as fp32 in LDS (16-byte loads along the channels), eight lanes per pixel
// sum the squares, and the write-out runs along the pixels (32 consecutive floats per channel; LDS pitch C + 1 keeps
// both directions conflict-free).
template <typename T16>
__device__ __forceinline__ float to_f32(uint16_t raw);
template <>
__device__ __forceinline__ float to_f32<__half>(uint16_t raw) {
  return __half2float(__ushort_as_half(raw));
}
template <>
__device__ __forceinline__ float to_f32<__hip_bfloat16>(uint16_t raw) {
  return __uint_as_float((uint32_t)raw << 16);
}

template <typename T16>
__global__ __launch_bounds__(256) void l2norm_nhwc16_to_nchw_kernel(const uint16_t* __restrict__ x,
                                                                    float* __restrict__ out, int C, int HW) {
  extern __shared__ float tile[];                 // [32][C + 1] + inv[32]
  constexpr int PX = 32;
  const int P = C + 1, tid = threadIdx.x, pix0 = blockIdx.x * PX;
  const int npx = min(PX, HW - pix0);
  float* inv = tile + PX * P;
  const uint16_t* src = x + ((size_t)blockIdx.y * HW + pix0) * C;
  const int nvec = npx * C / 8;                    // C % 8 == 0: the tile is a contiguous run of 16-byte vectors
  for (int v = tid; v < nvec; v += 256) {
    const uint4 raw = *reinterpret_cast<const uint4*>(src + (size_t)v * 8);
    const int px = (v * 8) / C, c = (v * 8) % C;
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tile[px * P + c + 2 * k] = to_f32<T16>((uint16_t)(w[k] & 0xffffu));
      tile[px * P + c + 2 * k + 1] = to_f32<T16>((uint16_t)(w[k] >> 16));
    }
  }
  __syncthreads();
  {
    const int px = tid >> 3, part = tid & 7;
    float ss = 0.0f;
    if (px < npx)
      for (int c = part; c < C; c += 8) {
        const float v = tile[px * P + c];
        ss = fmaf(v, v, ss);
      }
    ss += __shfl_xor(ss, 1, 64);
    ss += __shfl_xor(ss, 2, 64);
    ss += __shfl_xor(ss, 4, 64);
    if (part == 0) inv[px] = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
  const int px = tid & 31;
  if (px < npx) {
    const float d = 1.0f / inv[px];                // divide, as F.normalize does (x / max(norm, eps))
    float* o = out + (size_t)blockIdx.y * C * HW + pix0 + px;
    for (int c = tid >> 5; c < C; c += 8) o[(size_t)c * HW] = tile[px * P + c] / d;
  }
}

// F.interpolate(mode='area') == adaptive average pooling: out[oy,ox] = mean of in[floor(oy*H/OH) .. ceil((oy+1)*H/OH))
// x the same along W (ATen start_index/end_index).  One thread per output element; windows are <= ~5x8 at 1080->256.
__global__ __launch_bounds__(256) void area_resize_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                          int W, int OH, int OW, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ox = idx % OW;
  const int oy = (idx / OW) % OH;
  const long plane = idx / ((long)OW * OH);
  const int y0 = (int)(((long)oy * H) / OH), y1 = (int)((((long)oy + 1) * H + OH - 1) / OH);
  const int x0 = (int)(((long)ox * W) / OW), x1 = (int)((((long)ox + 1) * W + OW - 1) / OW);
  const float* p = x + plane * H * W;
  float s = 0.0f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) s += p[(size_t)yy * W + xx];
  out[idx] = s / (float)((y1 - y0) * (x1 - x0));
}


// 2 x 2 / stride 2 max-pool of a channels-last 16-bit map (the pools of the VGG-16 pyramid, models/backbones/vgg.py:33-60:
// nn.MaxPool2d(2, 2), floor mode): a thread owns 8 channels of 2 adjacent output pixels -- 8 independent 16-byte loads in
// flight (ATen's channels-last pool: 0.54 ms for 4 x 1080 x 1920 x 64 = 2.4 TB/s).  Max of representable values: exact.
// Re-tiling of the uncertainty head's micro-image chain under autograd (align.py UncertaintyModule._patch_statistics_tiled;
// models/modules.py:528-545 runs one 3x3 valid convolution per s x s micro-image): the micro-images are the (k + 2) x (k + 2) tiles of
// one channels-last image, a valid 3x3 convolution of it holds every tile's k x k result at rows / columns t (k + 2) + i, i < k, and the
// next layer wants those as the k x k tiles of a (k h, k w) image.  Forward: dst (B, k h, k w, .) gathers from src (B, (k + 2) h - 2,
// (k + 2) w - 2, .); backward: the gradient of src gathers from the gradient of dst, zero at the dropped positions (windows that
// straddle two tiles) -- no atomics either way.  A pixel is U 16-byte units of any element type (ATen's strided copy of the 6-d view
// ran at 0.4 TB/s and came with two layout conversions per layer: 26 ms of a 125 ms matcher step, profiles/r04_matcher_census.txt).
// `Us`: 16-byte units from one source pixel to the next (>= U: the implicit-GEMM convolution pads its output channels to 64).
__global__ __launch_bounds__(256) void retile_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int h, int w, int k,
                                                          int U, int Us, int backward, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int k2 = k + 2, Hs = k2 * h - 2, Ws = k2 * w - 2, Hd = k * h, Wd = k * w;
  const int u = (int)(idx % U);
  long t = idx / U;
  if (!backward) {                                         // idx over dst (B, Hd, Wd, U)
    const int x = (int)(t % Wd);
    t /= Wd;
    const int y = (int)(t % Hd), b = (int)(t / Hd);
    const int sy = (y / k) * k2 + y % k, sx = (x / k) * k2 + x % k;
    dst[idx] = src[(((long)b * Hs + sy) * Ws + sx) * Us + u];
  } else {                                                 // idx over the gradient of src (B, Hs, Ws, U)
    const int x = (int)(t % Ws);
    t /= Ws;
    const int y = (int)(t % Hs), b = (int)(t / Hs);
    const int i = y % k2, j = x % k2;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i < k && j < k) v = src[(((long)b * Hd + (y / k2) * k + i) * Wd + (x / k2) * k + j) * Us + u];
    dst[idx] = v;
  }
}

template <typename S>
__global__ __launch_bounds__(256) void maxpool2x2_nhwc16_kernel(const S* __restrict__ x, S* __restrict__ y, int H, int W,
                                                                int OH, int OW, int CV, long total) {
  typedef S V8 __attribute__((ext_vector_type(8)));
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;                                // total = B * OH * ceil(OW / 2) * CV
  const int OWP = (OW + 1) / 2;
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int xp = (int)(t % OWP);
  t /= OWP;
  const int oy = (int)(t % OH), b = (int)(t / OH);
  const long rs = (long)W * CV * 8;
  const S* p = x + ((long)b * H + 2 * oy) * rs + (long)cv * 8;
  V8 v[2][4];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ix = min(4 * xp + c, W - 1);
      v[r][c] = *(const V8*)(p + r * rs + (long)ix * CV * 8);
    }
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const int ox = 2 * xp + o;
    if (ox >= OW) break;
    V8 m;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a = fmaxf((float)v[0][2 * o][e], (float)v[0][2 * o + 1][e]);
      const float c = fmaxf((float)v[1][2 * o][e], (float)v[1][2 * o + 1][e]);
      m[e] = (S)fmaxf(a, c);
    }
    *(V8*)(y + (((long)b * OH + oy) * OW + ox) * CV * 8 + (long)cv * 8) = m;
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" {

int rfn_area_resize_f32(const float* x, float* out, int planes, int H, int W, int OH, int OW, rfn_stream_t stream) {
  RFN_REQUIRE(x && out, "rfn_area_resize_f32: null pointer");
  RFN_REQUIRE(planes > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "rfn_area_resize_f32: non-positive size");
  const long total = (long)planes * OH * OW;
  hipLaunchKernelGGL(area_resize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, out, H, W, OH,
                     OW, total);
  return check_launch("area_resize_kernel");
}

int rfn_warp_f32(const float* x, const float* flow, float* out, unsigned char* mask, int B, int C, int H, int W,
                 rfn_stream_t stream) {
  RFN_REQUIRE(x && flow && out, "rfn_warp_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "rfn_warp_f32: non-positive size");
  RFN_REQUIRE((long)H * W < 0x7fffffffL && B <= 65535, "rfn_warp_f32: tensor too large");
  constexpr int CG = 32;
  dim3 grid(cdiv((long)H * W, 256), cdiv(C, CG), B);
  hipLaunchKernelGGL((warp_kernel<CG>), grid, dim3(256), 0, (hipStream_t)stream, x, flow, out, mask, C, H, W);
  return check_launch("warp_kernel");
}

int rfn_warp_bwd_f32(const float* x, const float* flow, const float* grad_out, float* grad_x, float* grad_flow, int B,
                     int C, int H, int W, rfn_stream_t stream) {
  RFN_REQUIRE(x && flow && grad_out && (grad_x || grad_flow), "rfn_warp_bwd_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "rfn_warp_bwd_f32: non-positive size");
  RFN_REQUIRE((long)H * W < 0x7fffffffL && B <= 65535, "rfn_warp_bwd_f32: tensor too large");
  hipStream_t s = (hipStream_t)stream;
  const size_t HW = (size_t)H * W;
  if (grad_x && hipMemsetAsync(grad_x, 0, (size_t)B * C * HW * sizeof(float), s) != hipSuccess)
    return fail(RFN_ELAUNCH, "rfn_warp_bwd_f32: memset");
  if (grad_flow && hipMemsetAsync(grad_flow, 0, (size_t)B * 2 * HW * sizeof(float), s) != hipSuccess)
    return fail(RFN_ELAUNCH, "rfn_warp_bwd_f32: memset");
  constexpr int CG = 32;
  dim3 grid(cdiv((long)HW, 256), cdiv(C, CG), B);
  hipLaunchKernelGGL((warp_bwd_kernel<CG>), grid, dim3(256), 0, s, x, flow, grad_out, grad_x, grad_flow, C, H, W);
  return check_launch("warp_bwd_kernel");
}

int rfn_align_tail_f32(const float* logits_ref, const float* flow_q, const float* logvar_q, float* warped,
                       unsigned char* mask, float* cert, float* flow_up, int B, int C, int H, int W, int h, int w,
                       rfn_stream_t stream) {
  RFN_REQUIRE(logits_ref && flow_q && logvar_q && warped && mask && cert, "rfn_align_tail_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0, "rfn_align_tail_f32: non-positive size");
  RFN_REQUIRE((long)H * W < 0x7fffffffL && B <= 65535, "rfn_align_tail_f32: tensor too large");
  dim3 grid(cdiv((long)H * W, 256), 1, B);
  // ATen area_pixel_compute_scale(align_corners=False, no explicit scale): input_size / output_size in float
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  hipLaunchKernelGGL(align_tail_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits_ref, flow_q, logvar_q,
                     warped, mask, cert, flow_up, C, H, W, h, w, sy, sx);
  return check_launch("align_tail_kernel");
}

int rfn_l2norm_channels_f32(const float* x, float* out, int B, int C, int HW, rfn_stream_t stream) {
  RFN_REQUIRE(x && out, "rfn_l2norm_channels_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && HW > 0 && B <= 65535, "rfn_l2norm_channels_f32: bad size");
  hipStream_t st = (hipStream_t)stream;
  if (C == 128)
    hipLaunchKernelGGL((l2norm_channels_reg_kernel<4, 32>), dim3(cdiv(HW, 64), B), dim3(256), 0, st, x, out, HW);
  else if (C == 256)
    hipLaunchKernelGGL((l2norm_channels_reg_kernel<8, 32>), dim3(cdiv(HW, 32), B), dim3(256), 0, st, x, out, HW);
  else if (C == 512)
    hipLaunchKernelGGL((l2norm_channels_reg_kernel<8, 64>), dim3(cdiv(HW, 32), B), dim3(256), 0, st, x, out, HW);
  else
    hipLaunchKernelGGL(l2norm_channels_kernel, dim3(cdiv(HW, 256), B), dim3(256), 0, st, x, out, C, HW);
  return check_launch("l2norm_channels_kernel");
}

int rfn_l2norm_channels_nhwc16_f32(const void* x, float* out, int B, int C, int HW, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && out, "rfn_l2norm_channels_nhwc16_f32: null pointer");
  RFN_REQUIRE(B > 0 && B <= 65535 && C > 0 && C % 8 == 0 && C <= 2048 && HW > 0,
              "rfn_l2norm_channels_nhwc16_f32: B=%d C=%d (multiple of 8, <= 2048) HW=%d", B, C, HW);
  RFN_REQUIRE(dtype == 1 || dtype == 2, "rfn_l2norm_channels_nhwc16_f32: dtype %d (1 = bf16, 2 = f16)", dtype);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = ((size_t)32 * (C + 1) + 32) * sizeof(float);
  dim3 grid(cdiv(HW, 32), B);
  if (dtype == 2) {
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)l2norm_nhwc16_to_nchw_kernel<__half>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(RFN_ELAUNCH, "l2norm_nhwc16_to_nchw_kernel: %zu bytes of LDS", lds);
    hipLaunchKernelGGL(l2norm_nhwc16_to_nchw_kernel<__half>, grid, dim3(256), lds, st, (const uint16_t*)x, out, C, HW);
  } else {
    if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)l2norm_nhwc16_to_nchw_kernel<__hip_bfloat16>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(RFN_ELAUNCH, "l2norm_nhwc16_to_nchw_kernel: %zu bytes of LDS", lds);
    hipLaunchKernelGGL(l2norm_nhwc16_to_nchw_kernel<__hip_bfloat16>, grid, dim3(256), lds, st, (const uint16_t*)x, out, C,
                       HW);
  }
  return check_launch("l2norm_nhwc16_to_nchw_kernel");
}

// y (B, H / 2, W / 2, C) = 2 x 2 / stride 2 max-pool (floor mode) of the channels-last 16-bit x (B, H, W, C), C % 8 == 0; dtype 1
// bf16, 2 f16
int rfn_maxpool2x2_nhwc16(const void* x, void* y, int B, int H, int W, int C, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && y, "maxpool2x2_nhwc16: null pointer");
  RFN_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "maxpool2x2_nhwc16: B=%d H=%d W=%d C=%d (C %% 8)", B, H, W, C);
  RFN_REQUIRE(dtype == 1 || dtype == 2, "maxpool2x2_nhwc16: dtype %d (1 = bf16, 2 = f16)", dtype);
  const int OH = H / 2, OW = W / 2, CV = C / 8;
  const long total = (long)B * OH * ((OW + 1) / 2) * CV;
  RFN_REQUIRE(total / 256 < 0x7fffffffL, "maxpool2x2_nhwc16: too large");
  const int grid = cdiv(total, 256);
  if (dtype == 1)
    hipLaunchKernelGGL(maxpool2x2_nhwc16_kernel<__bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x,
                       (__bf16*)y, H, W, OH, OW, CV, total);
  else
    hipLaunchKernelGGL(maxpool2x2_nhwc16_kernel<_Float16>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x,
                       (_Float16*)y, H, W, OH, OW, CV, total);
  return check_launch("maxpool2x2_nhwc16");
}

int rfn_retile_copy(const void* src, void* dst, int B, int h, int w, int k, int units16, int src_stride16, int backward,
                    rfn_stream_t stream) {
  RFN_REQUIRE(src && dst, "rfn_retile_copy: null pointer");
  RFN_REQUIRE(B > 0 && h > 0 && w > 0 && k > 0 && units16 > 0 && src_stride16 >= units16, "rfn_retile_copy: bad size");
  RFN_REQUIRE((((size_t)src | (size_t)dst) & 15) == 0, "rfn_retile_copy: pointers must be 16-byte aligned");
  const long Hs = (long)(k + 2) * h - 2, Ws = (long)(k + 2) * w - 2;
  const long total = backward ? (long)B * Hs * Ws * units16 : (long)B * k * h * k * w * units16;
  RFN_REQUIRE(total / 256 < 0x7fffffffL && Hs < 0x7fffffffL && Ws < 0x7fffffffL, "rfn_retile_copy: too large");
  hipLaunchKernelGGL(retile_copy_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const uint4*)src,
                     (uint4*)dst, h, w, k, units16, src_stride16, backward, total);
  return check_launch("retile_copy");
}

}  // extern "C"
