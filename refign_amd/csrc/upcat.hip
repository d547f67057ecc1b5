// refign_amd/csrc/upcat.hip -- multi-resolution fusion front end of the decode heads: bilinear up-sampling of the
// per-stage embeddings to the 1/4-resolution grid AND their channel concatenation in one pass.
//
// Reference: DAFormerHead.forward (models/heads/daformer.py:205-222) and SegFormerHead.forward
// (models/heads/segformer.py:86-104): for each of the 4 MiT stages  embed (Linear on tokens) -> reshape to NCHW ->
// F.interpolate(size of stage 1, bilinear, align_corners=False) -> torch.cat(dim=1).  Unfused that is 3 up-sampling
// kernels writing (n, 256, h, w) each plus a concat that reads and re-writes all (n, 1024, h, w): at the teacher's
// 44 x 135 x 240 maps 2.2 GB + 5.8 GB of traffic for a 2.9 GB result.  Here every output vector is produced once:
// read <= 4 neighbour vectors of its (small) source level, blend in fp32, write 16 bytes -- channels-last output,
// which is what the ASPP's convolutions and depthwise kernels want.
// Sources are the TOKEN maps (n, h_l*w_l, C_l) as the embedding Linear produces them (= channels-last).
// align_corners=False sampling as ATen (UpSample.h area_pixel_compute_source_index): src = max((dst + 0.5) * in/out -
// 0.5, 0), i1 = min(i0 + 1, in - 1).
#include <hip/hip_bf16.h>

#include <type_traits>

#include "common.h"

namespace rfn {

struct UpcatArgs {
  const void* src[4];
  int h[4], w[4], cv0[5];      // cv0: prefix sum of channel VECTORS (8 channels each) per level
  int nlev;
};

__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

// one thread = one 8-channel vector of kUpPX consecutive output pixels of a row (bf16: 16 bytes per pixel; fp32: 32 bytes): the
// 4 x kUpPX source loads are issued together -- with one pixel per thread the kernel was a chain of dependent latencies
// (index arithmetic -> 4 loads -> blend -> store) at 2.6 TB/s of a 6.7 TB/s write rate
constexpr int kUpPX = 4;

template <typename T>
__global__ __launch_bounds__(256) void upcat_nhwc_kernel(UpcatArgs a, T* __restrict__ out, int H, int W, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;                              // total = n * H * ceil(W / kUpPX) * CV
  const int CV = a.cv0[a.nlev], WQ = (W + kUpPX - 1) / kUpPX;
  const int cv = (int)(idx % CV);
  long pix = idx / CV;
  const int xq = (int)(pix % WQ);
  pix /= WQ;
  const int y = (int)(pix % H);
  const int n = (int)(pix / H);
  int l = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.nlev && cv >= a.cv0[i]) l = i;
  const int cl = cv - a.cv0[l], CVl = a.cv0[l + 1] - a.cv0[l];
  const int hl = a.h[l], wl = a.w[l];
  const T* s = reinterpret_cast<const T*>(a.src[l]) + (size_t)n * hl * wl * CVl * 8 + (size_t)cl * 8;
  const size_t st = (size_t)CVl * 8;
  const bool same = hl == H && wl == W;
  const float sy = fmaxf(((float)y + 0.5f) * ((float)hl / (float)H) - 0.5f, 0.0f);
  const int y0 = same ? y : (int)sy, y1 = min(y0 + 1, hl - 1);
  const float ly = same ? 0.f : sy - (float)y0;
  typedef typename std::conditional<sizeof(T) == 2, uint4, float4>::type V16;     // 16 bytes: 8 bf16 or 4 floats
  constexpr int NV = sizeof(T) == 2 ? 1 : 2;
  V16 v[kUpPX][4][NV];
  float lx[kUpPX];
#pragma unroll
  for (int p = 0; p < kUpPX; ++p) {
    const int x = min(xq * kUpPX + p, W - 1);
    const float sx = fmaxf(((float)x + 0.5f) * ((float)wl / (float)W) - 0.5f, 0.0f);
    const int x0 = same ? x : (int)sx, x1 = min(x0 + 1, wl - 1);
    lx[p] = same ? 0.f : sx - (float)x0;
    const T* p00 = s + ((size_t)y0 * wl + x0) * st;
    const T* p01 = s + ((size_t)y0 * wl + x1) * st;
    const T* p10 = s + ((size_t)y1 * wl + x0) * st;
    const T* p11 = s + ((size_t)y1 * wl + x1) * st;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      v[p][0][q] = reinterpret_cast<const V16*>(p00)[q];
      if (!same) {                                       // (wave-uniform: a wave's lanes are one level)
        v[p][1][q] = reinterpret_cast<const V16*>(p01)[q];
        v[p][2][q] = reinterpret_cast<const V16*>(p10)[q];
        v[p][3][q] = reinterpret_cast<const V16*>(p11)[q];
      }
    }
  }
  auto widen = [](const V16 (&t)[NV], float (&f)[8]) {
    if constexpr (sizeof(T) == 2) unpack8(t[0], f);
    else {
      f[0] = t[0].x; f[1] = t[0].y; f[2] = t[0].z; f[3] = t[0].w;
      f[4] = t[1].x; f[5] = t[1].y; f[6] = t[1].z; f[7] = t[1].w;
    }
  };
#pragma unroll
  for (int p = 0; p < kUpPX; ++p) {
    const int x = xq * kUpPX + p;
    if (x >= W) break;
    float r[8];
    if (same) {
      widen(v[p][0], r);
    } else {
      const float w00 = (1.0f - ly) * (1.0f - lx[p]), w01 = (1.0f - ly) * lx[p], w10 = ly * (1.0f - lx[p]), w11 = ly * lx[p];
      float v00[8], v01[8], v10[8], v11[8];
      widen(v[p][0], v00);
      widen(v[p][1], v01);
      widen(v[p][2], v10);
      widen(v[p][3], v11);
#pragma unroll
      for (int i = 0; i < 8; ++i) r[i] = w00 * v00[i] + w01 * v01[i] + w10 * v10[i] + w11 * v11[i];
    }
    T* o = out + (((size_t)n * H + y) * W + x) * (size_t)CV * 8 + (size_t)cv * 8;
    if constexpr (sizeof(T) == 2) {
      uint4 t;
      t.x = bf16x2_bits(r[0], r[1]);
      t.y = bf16x2_bits(r[2], r[3]);
      t.z = bf16x2_bits(r[4], r[5]);
      t.w = bf16x2_bits(r[6], r[7]);
      *reinterpret_cast<uint4*>(o) = t;
    } else {
      *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
      *reinterpret_cast<float4*>(o + 4) = make_float4(r[4], r[5], r[6], r[7]);
    }
  }
}

// Backward of the above: the gradient of every SOURCE vector in one pass over the concatenated gradient.  One thread =
// one 8-channel vector of one source pixel (ys, xs) of level l; it gathers over the output pixels whose bilinear
// footprint contains the source pixel -- a contiguous window of about 2 H/h_l x 2 W/w_l pixels -- with the weights the
// forward used, recomputed by the forward's own float arithmetic (a candidate row / column that does not reference the
// source pixel gets weight 0, so the window only has to be generous).  Reads the (n, H, W, sum C) gradient in place with
// the level's channel offset: no slicing, no per-level scatter (the library's up-sampling backward on channel slices of
// the fused gradient: 310 us per level and pass, plus the strided slice copies).  fp32 accumulation, one rounding.
struct UpcatBwdArgs {
  void* dst[4];
  int h[4], w[4], cv0[5];
  long vec0[5];                // prefix sum of n * h_l * w_l * (C_l / 8): thread ranges of the levels
  int nlev;
};

template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_nhwc_kernel(UpcatBwdArgs a, const T* __restrict__ g, int H, int W,
                                                             long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.nlev && idx >= a.vec0[i]) l = i;
  const int CV = a.cv0[a.nlev], CVl = a.cv0[l + 1] - a.cv0[l], hl = a.h[l], wl = a.w[l];
  long loc = idx - a.vec0[l];
  const int cl = (int)(loc % CVl);
  loc /= CVl;
  const int xs = (int)(loc % wl);
  loc /= wl;
  const int ys = (int)(loc % hl);
  const int n = (int)(loc / hl);
  const T* gb = g + (size_t)n * H * W * CV * 8 + (size_t)(a.cv0[l] + cl) * 8;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto add = [&](const T* p, float wgt) {
    float v[8];
    if constexpr (sizeof(T) == 2) unpack8(*reinterpret_cast<const uint4*>(p), v);
    else {
      const float4 u = *reinterpret_cast<const float4*>(p), t = *reinterpret_cast<const float4*>(p + 4);
      v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w; v[4] = t.x; v[5] = t.y; v[6] = t.z; v[7] = t.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(wgt, v[i], acc[i]);
  };
  if (hl == H && wl == W) {
    add(gb + ((size_t)ys * W + xs) * CV * 8, 1.0f);
  } else {
    const float ry = (float)hl / (float)H, rx = (float)wl / (float)W;       // the forward's scales
    const int ylo = max(0, (int)floorf(((float)ys - 0.5f) / ry - 0.5f) - 1);
    const int yhi = min(H - 1, (int)ceilf(((float)ys + 1.5f) / ry - 0.5f) + 1);
    const int xlo = max(0, (int)floorf(((float)xs - 0.5f) / rx - 0.5f) - 1);
    const int xhi = min(W - 1, (int)ceilf(((float)xs + 1.5f) / rx - 0.5f) + 1);
    for (int y = ylo; y <= yhi; ++y) {
      const float sy = fmaxf(((float)y + 0.5f) * ry - 0.5f, 0.0f);
      const int y0 = (int)sy, y1 = min(y0 + 1, hl - 1);
      const float ly = sy - (float)y0;
      const float wy = (y0 == ys ? 1.0f - ly : 0.0f) + (y1 == ys ? ly : 0.0f);
      if (wy == 0.0f) continue;
      const T* row = gb + (size_t)y * W * CV * 8;
      for (int x = xlo; x <= xhi; ++x) {
        const float sx = fmaxf(((float)x + 0.5f) * rx - 0.5f, 0.0f);
        const int x0 = (int)sx, x1 = min(x0 + 1, wl - 1);
        const float lx = sx - (float)x0;
        const float wx = (x0 == xs ? 1.0f - lx : 0.0f) + (x1 == xs ? lx : 0.0f);
        if (wx != 0.0f) add(row + (size_t)x * CV * 8, wy * wx);
      }
    }
  }
  T* o = reinterpret_cast<T*>(a.dst[l]) + (((size_t)n * hl + ys) * wl + xs) * (size_t)CVl * 8 + (size_t)cl * 8;
  if constexpr (sizeof(T) == 2) {
    uint4 t;
    t.x = bf16x2_bits(acc[0], acc[1]);
    t.y = bf16x2_bits(acc[2], acc[3]);
    t.z = bf16x2_bits(acc[4], acc[5]);
    t.w = bf16x2_bits(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(o) = t;
  } else {
    *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" {

int rfn_upsample_concat_nhwc(const void* src0, const void* src1, const void* src2, const void* src3, const int* hs,
                             const int* ws, const int* cs, int nlev, void* out, int n, int H, int W, int dtype,
                             rfn_stream_t stream) {
  RFN_REQUIRE(nlev >= 1 && nlev <= 4 && out && hs && ws && cs && n > 0 && H > 0 && W > 0,
              "rfn_upsample_concat_nhwc: bad arguments");
  UpcatArgs a;
  const void* srcs[4] = {src0, src1, src2, src3};
  a.nlev = nlev;
  a.cv0[0] = 0;
  for (int l = 0; l < 4; ++l) {
    a.src[l] = l < nlev ? srcs[l] : nullptr;
    a.h[l] = l < nlev ? hs[l] : 1;
    a.w[l] = l < nlev ? ws[l] : 1;
    if (l < nlev) {
      RFN_REQUIRE(srcs[l] && hs[l] > 0 && ws[l] > 0 && cs[l] > 0 && cs[l] % 8 == 0,
                  "rfn_upsample_concat_nhwc: level %d: null source or channels not a multiple of 8", l);
      a.cv0[l + 1] = a.cv0[l] + cs[l] / 8;
    } else {
      a.cv0[l + 1] = a.cv0[l];
    }
  }
  const long total = (long)n * H * ((W + kUpPX - 1) / kUpPX) * a.cv0[nlev];      // threads: kUpPX pixels of a row each
  RFN_REQUIRE(total / 256 < 0x7fffffffL, "rfn_upsample_concat_nhwc: too large");
  const int grid = cdiv(total, 256);
  if (dtype == 1)
    hipLaunchKernelGGL((upcat_nhwc_kernel<__hip_bfloat16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a,
                       (__hip_bfloat16*)out, H, W, total);
  else if (dtype == 0)
    hipLaunchKernelGGL((upcat_nhwc_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a, (float*)out, H, W,
                       total);
  else
    return fail(RFN_EINVAL, "rfn_upsample_concat_nhwc: dtype must be 0 (f32) or 1 (bf16)");
  return check_launch("upcat_nhwc_kernel");
}

int rfn_upsample_concat_nhwc_bwd(const void* grad_out, void* grad0, void* grad1, void* grad2, void* grad3, const int* hs,
                                 const int* ws, const int* cs, int nlev, int n, int H, int W, int dtype,
                                 rfn_stream_t stream) {
  RFN_REQUIRE(nlev >= 1 && nlev <= 4 && grad_out && hs && ws && cs && n > 0 && H > 0 && W > 0,
              "rfn_upsample_concat_nhwc_bwd: bad arguments");
  UpcatBwdArgs a;
  void* dsts[4] = {grad0, grad1, grad2, grad3};
  a.nlev = nlev;
  a.cv0[0] = 0;
  a.vec0[0] = 0;
  for (int l = 0; l < 4; ++l) {
    a.dst[l] = l < nlev ? dsts[l] : nullptr;
    a.h[l] = l < nlev ? hs[l] : 1;
    a.w[l] = l < nlev ? ws[l] : 1;
    if (l < nlev) {
      RFN_REQUIRE(dsts[l] && hs[l] > 0 && ws[l] > 0 && hs[l] <= H && ws[l] <= W && cs[l] > 0 && cs[l] % 8 == 0,
                  "rfn_upsample_concat_nhwc_bwd: level %d: null gradient, channels not a multiple of 8 or larger than the "
                  "output", l);
      a.cv0[l + 1] = a.cv0[l] + cs[l] / 8;
      a.vec0[l + 1] = a.vec0[l] + (long)n * hs[l] * ws[l] * (cs[l] / 8);
    } else {
      a.cv0[l + 1] = a.cv0[l];
      a.vec0[l + 1] = a.vec0[l];
    }
  }
  const long total = a.vec0[nlev];
  RFN_REQUIRE(total / 256 < 0x7fffffffL, "rfn_upsample_concat_nhwc_bwd: too large");
  const int grid = cdiv(total, 256);
  if (dtype == 1)
    hipLaunchKernelGGL((upcat_bwd_nhwc_kernel<__hip_bfloat16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a,
                       (const __hip_bfloat16*)grad_out, H, W, total);
  else if (dtype == 0)
    hipLaunchKernelGGL((upcat_bwd_nhwc_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a,
                       (const float*)grad_out, H, W, total);
  else
    return fail(RFN_EINVAL, "rfn_upsample_concat_nhwc_bwd: dtype must be 0 (f32) or 1 (bf16)");
  return check_launch("upcat_bwd_nhwc_kernel");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Token map (B, H*W, C) <-> non-overlapping r x r patches (B*Hr*Wr, r*r*C), Hr = H/r, Wr = W/r (ragged border dropped):
// the gather in front of / the scatter behind the spatial-reduction Linear of the MiT attention (refign_amd/conv.py).
// One thread moves one 8-channel vector; `inverse` scatters patches back into a token map whose ragged border (if any)
// the caller has zeroed.  The generic strided copy of the framework needed ~13 us for 5 MB (6-D index arithmetic per
// element); 560 of them per step.
// ---------------------------------------------------------------------------------------------------------------------
namespace rfn {

template <int VB>   // bytes per vector: 16 (bf16 x 8) or 32 (f32 x 8)
__global__ __launch_bounds__(256) void patchify_kernel(const char* __restrict__ src, char* __restrict__ dst, int H,
                                                       int W, int CV, int r, int Hr, int Wr, long total, int inverse) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  // patch-side linear index: (((b*Hr + i)*Wr + j)*r + ry)*r + rx)*CV + cv
  const int cv = (int)(idx % CV);
  long t = idx / CV;
  const int rx = (int)(t % r); t /= r;
  const int ry = (int)(t % r); t /= r;
  const int j = (int)(t % Wr); t /= Wr;
  const int i = (int)(t % Hr);
  const long b = t / Hr;
  const long tok = ((b * H + (long)i * r + ry) * W + (long)j * r + rx) * CV + cv;
  const char* s = inverse ? src + idx * VB : src + tok * VB;
  char* d = inverse ? dst + tok * VB : dst + idx * VB;
  if constexpr (VB == 16) {
    *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
  } else {
    *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
    *reinterpret_cast<uint4*>(d + 16) = *reinterpret_cast<const uint4*>(s + 16);
  }
}

// The same gather with the patch row in (c, ry, rx) order -- the (C, r, r) layout of one output channel of the convolution
// weight: the patch GEMM then multiplies by the parameter's own (Co, C r r) matrix and its weight gradient is ADDED
// straight into the parameter's .grad (coalesced atomics of the TN kernel) -- no permuted weight copy to refresh, no
// partial-sum + permuted-add passes (5 launches per spatial-reduction convolution and backward pass).
// Both sides of the move are 16-byte vectors that are contiguous across the lanes of a wave; the element transposition
// happens in LDS, which holds `np` patches in TOKEN order [p][ry][rx][c] (a first version with each thread writing its 8
// channels' runs of r elements straight to memory -- 4-byte stores 64 bytes apart -- ran at a quarter of the speed).
template <typename E, int R>
__global__ __launch_bounds__(256) void patchify_cmajor_kernel(const E* __restrict__ src, E* __restrict__ dst, int H, int W,
                                                              int C, int Hr, int Wr, int npatch, int np, int inverse,
                                                              unsigned magic_kv, unsigned magic_cve) {
  extern __shared__ uint4 patch_lds[];
  constexpr int VE = 16 / sizeof(E);
  const int K = R * R * C, KV = K / VE, CVE = C / VE;
  const int p0 = blockIdx.x * np;
  const int n = min(np, npatch - p0);
  const int nvec = n * KV;
  E* sm = reinterpret_cast<E*>(patch_lds);
  long* pbase = reinterpret_cast<long*>(sm + (size_t)np * K);     // token address of pixel (0, 0), channel 0 of each patch
  if ((int)threadIdx.x < n) {
    const int patch = p0 + threadIdx.x, j = patch % Wr, t = patch / Wr, i = t % Hr, b = t / Hr;
    pbase[threadIdx.x] = (((long)b * H + (long)i * R) * W + (long)j * R) * C;
  }
  // neighbouring lanes of the patch side are VE elements of a patch row apart: for R R > 8 that is the same channel at pixel
  // index rr + 8, C elements further -- the same bank.  XOR the pixel's 16-byte vector index with rr / 8.
  constexpr int SWM = (R * R >= 16) ? (R * R / 8 - 1) : 0;
  const int swm = (CVE % (SWM + 1) == 0) ? SWM : 0;
  // the two run-time divisors (vectors per patch, vectors per pixel) by multiplication: magic = 2^32 / d + 1, exact for
  // dividends < 2^32 / d (a workgroup's vector count; integer divisions were most of the first version's instructions)
  struct Tok { int lds; long addr; };
  auto token_side = [&](int v) -> Tok {          // v = (p, ry, rx, cv): LDS vector p KV + rr CVE + (cv ^ swizzle(rr))
    const int p = (int)__umulhi((unsigned)v, magic_kv), rem = v - p * KV;
    const int rr = (int)__umulhi((unsigned)rem, magic_cve), cv = rem - rr * CVE;
    const int ry = rr / R, rx = rr - ry * R;
    return Tok{p * KV + rr * CVE + (cv ^ ((rr >> 3) & swm)), pbase[p] + ((long)ry * W + rx) * C + cv * VE};
  };
  auto lds_elem = [&](int p, int k) -> int {     // patch-side element k = (c, ry, rx) of local patch p
    const int c = k / (R * R), rr = k - c * (R * R);
    const int cv = c / VE;
    return p * K + rr * C + ((cv ^ ((rr >> 3) & swm)) * VE) + (c - cv * VE);
  };
  __syncthreads();
  if (!inverse) {
    for (int v = threadIdx.x; v < nvec; v += 256) {
      const Tok t = token_side(v);
      patch_lds[t.lds] = *reinterpret_cast<const uint4*>(src + t.addr);
    }
    __syncthreads();
    for (int u = threadIdx.x; u < nvec; u += 256) {
      const int p = (int)__umulhi((unsigned)u, magic_kv), kv = u - p * KV;
      alignas(16) E o[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = sm[lds_elem(p, kv * VE + e)];
      *reinterpret_cast<uint4*>(dst + (long)(p0 + p) * K + (long)kv * VE) = *reinterpret_cast<const uint4*>(o);
    }
  } else {
    for (int u = threadIdx.x; u < nvec; u += 256) {
      const int p = (int)__umulhi((unsigned)u, magic_kv), kv = u - p * KV;
      alignas(16) E o[VE];
      *reinterpret_cast<uint4*>(o) = *reinterpret_cast<const uint4*>(src + (long)(p0 + p) * K + (long)kv * VE);
#pragma unroll
      for (int e = 0; e < VE; ++e) sm[lds_elem(p, kv * VE + e)] = o[e];
    }
    __syncthreads();
    for (int v = threadIdx.x; v < nvec; v += 256) {
      const Tok t = token_side(v);
      *reinterpret_cast<uint4*>(dst + t.addr) = patch_lds[t.lds];
    }
  }
}

template <typename E>
static int launch_patchify_cmajor(const void* src, void* dst, int B, int H, int W, int C, int r, int inverse, hipStream_t st) {
  const int Hr = H / r, Wr = W / r;
  const long npatch = (long)B * Hr * Wr;
  const long pbytes = (long)r * r * C * sizeof(E);
  if (C < 16) return fail(RFN_EINVAL, "rfn_patchify_tokens_cmajor: C >= 16 (got %d)", C);
  if (pbytes > 65536 - 8 || npatch >= (1L << 31) / (r * r * C))
    return fail(RFN_EINVAL, "rfn_patchify_tokens_cmajor: a patch has to fit 64 KB of LDS (r r C = %ld bytes)", pbytes);
  // ~16 KB of patches per workgroup, fewer while that leaves the grid under two workgroups per CU
  long np = std::min<long>(256, std::max<long>(1, 16384 / pbytes));
  while (np > 1 && cdiv(npatch, np) < 512) np >>= 1;
  const int grid = (int)cdiv(npatch, np);
  const size_t lds = (size_t)np * pbytes + (size_t)np * sizeof(long);
  constexpr int VE = 16 / sizeof(E);
  const unsigned magic_kv = (unsigned)((1ull << 32) / (unsigned)(r * r * C / VE) + 1);
  const unsigned magic_cve = (unsigned)((1ull << 32) / (unsigned)(C / VE) + 1);
#define RFN_PCM(R)                                                                                                     \
  hipLaunchKernelGGL((patchify_cmajor_kernel<E, R>), dim3(grid), dim3(256), lds, st, (const E*)src, (E*)dst, H, W, C, \
                     Hr, Wr, (int)npatch, (int)np, inverse, magic_kv, magic_cve)
  if (r == 2) RFN_PCM(2);
  else if (r == 4) RFN_PCM(4);
  else RFN_PCM(8);
#undef RFN_PCM
  return check_launch("patchify_cmajor_kernel");
}

}  // namespace rfn

extern "C" {

// patch rows in (c, ry, rx) order (see patchify_cmajor_kernel); r in {2, 4, 8} (the MiT spatial-reduction ratios)
int rfn_patchify_tokens_cmajor(const void* src, void* dst, int B, int H, int W, int C, int r, int dtype, int inverse,
                               rfn_stream_t stream) {
  RFN_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (r == 2 || r == 4 || r == 8) && r <= H && r <= W,
              "rfn_patchify_tokens_cmajor: bad arguments (C %% 8 == 0, r in {2, 4, 8})");
  RFN_REQUIRE(dtype == 0 || dtype == 1, "rfn_patchify_tokens_cmajor: dtype must be 0 (f32) or 1 (bf16 / f16)");
  if (dtype == 1) return rfn::launch_patchify_cmajor<unsigned short>(src, dst, B, H, W, C, r, inverse, (hipStream_t)stream);
  return rfn::launch_patchify_cmajor<unsigned int>(src, dst, B, H, W, C, r, inverse, (hipStream_t)stream);
}

int rfn_patchify_tokens(const void* src, void* dst, int B, int H, int W, int C, int r, int dtype, int inverse,
                        rfn_stream_t stream) {
  RFN_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && r > 0 && r <= H && r <= W,
              "rfn_patchify_tokens: bad arguments (C must be a multiple of 8)");
  RFN_REQUIRE(dtype == 0 || dtype == 1, "rfn_patchify_tokens: dtype must be 0 (f32) or 1 (bf16)");
  const int Hr = H / r, Wr = W / r, CV = C / 8;
  const long total = (long)B * Hr * Wr * r * r * CV;
  const int grid = rfn::cdiv(total, 256);
  if (dtype == 1)
    hipLaunchKernelGGL((rfn::patchify_kernel<16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       (char*)dst, H, W, CV, r, Hr, Wr, total, inverse);
  else
    hipLaunchKernelGGL((rfn::patchify_kernel<32>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       (char*)dst, H, W, CV, r, Hr, Wr, total, inverse);
  return rfn::check_launch("patchify_kernel");
}

}  // extern "C"
