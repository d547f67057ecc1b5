// refign_amd/csrc/reduce.hip -- out[i] (+)= sum_{s<S} x[s*n + i]: the parameter-gradient reductions of the training step.
//
// The Refign step runs ~1 750 of these per step (one per token-wise Linear per backward pass: bias gradient =
// column sum of grad_y (tokens x features), tokens = 8 160 ... 259 200, features = 64 ... 2 048; plus the reduction of
// the split-T weight-gradient partials).  The library reduction needed 11-21 us for a 5 MB (8160 x 320) bf16 matrix
// (0.25 TB/s, profiles/r01_step_shapes_elem.txt) and autograd followed every one with a separate `grad += g` kernel.
// Here: pure HBM streaming with 16-byte loads, fp32 accumulation, deterministic two-stage reduction, and the result is
// ADDED straight into the parameter's .grad view of the flat gradient buffer.
//   tall (S > 64):   stage 1: grid = stripes of rows; a workgroup is (n/8 column vectors) x (rows in flight), every lane
//                    owns 8 adjacent columns and walks rows with stride TY -> contiguous, fully coalesced reads;
//                    LDS tree over the TY row groups; one partial row per stripe in the workspace.
//                    stage 2: (stripes, n) partials, 32 columns x 8 stripe segments per workgroup.
//   flat (S <= 64):  one thread per 4 columns, loop over S.
#include <hip/hip_bf16.h>

#include <algorithm>

#include "common.h"

namespace rfn {

template <typename T>
__device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void load8<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

constexpr int kMaxSumStripes = 512;
constexpr int kFlatMaxS = 64;

// stage 1 of the tall case.  blockDim.x = CVB * TY (<= 256) with CVB = column vectors handled by this workgroup
// (blockIdx.y selects the column tile when n/8 > 256).  Rows [r0, r1) of stripe blockIdx.x.
template <typename T>
__global__ __launch_bounds__(256) void sum_rows_tall_kernel(const T* __restrict__ x, float* __restrict__ ws, long S,
                                                            long n, int cvb, int ty_count, long rows_per_stripe) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int cx = tid % cvb, ty = tid / cvb;
  const long cv = (long)blockIdx.y * cvb + cx;            // column vector index
  const bool active = ty < ty_count && cv * 8 < n;
  const long r0 = (long)blockIdx.x * rows_per_stripe, r1 = min(S, r0 + rows_per_stripe);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    const T* p = x + cv * 8;
    long r = r0 + ty;
    // 4 rows in flight per lane
    for (; r + 3L * ty_count < r1; r += 4L * ty_count) {
      float a[8], b[8], c[8], d[8];
      load8<T>(p + r * n, a);
      load8<T>(p + (r + ty_count) * n, b);
      load8<T>(p + (r + 2L * ty_count) * n, c);
      load8<T>(p + (r + 3L * ty_count) * n, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (a[j] + b[j]) + (c[j] + d[j]);
    }
    for (; r < r1; r += ty_count) {
      float a[8];
      load8<T>(p + r * n, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += a[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid * 8 + j] = acc[j];
  __syncthreads();
  if (ty == 0 && cv * 8 < n) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < ty_count; ++t) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += red[(t * cvb + cx) * 8 + j];
    }
    float* o = ws + (long)blockIdx.x * n + cv * 8;
    *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(s[4], s[5], s[6], s[7]);
  }
}

// flat case: thread = 4 adjacent columns, serial over S (fixed order).  `blk` = workgroup index within this job.
template <typename T>
__device__ __forceinline__ void sum_rows_flat_body(const T* __restrict__ x, float* __restrict__ out, int S, long n,
                                                   int accumulate, long blk) {
  const long i = (blk * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if constexpr (sizeof(T) == 4) {
#pragma unroll 4
    for (int s = 0; s < S; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + (long)s * n + i);
      s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
    }
  } else {
#pragma unroll 4
    for (int s = 0; s < S; ++s) {
      const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(x) + (long)s * n + i);
      s0 += __uint_as_float(v.x << 16); s1 += __uint_as_float(v.x & 0xffff0000u);
      s2 += __uint_as_float(v.y << 16); s3 += __uint_as_float(v.y & 0xffff0000u);
    }
  }
  float4* o = reinterpret_cast<float4*>(out + i);
  if (accumulate) {
    const float4 p = *o;
    s0 += p.x; s1 += p.y; s2 += p.z; s3 += p.w;
  }
  *o = make_float4(s0, s1, s2, s3);
}

template <typename T>
__global__ __launch_bounds__(256) void sum_rows_flat_kernel(const T* __restrict__ x, float* __restrict__ out, int S, long n,
                                                            int accumulate) {
  sum_rows_flat_body<T>(x, out, S, n, accumulate, blockIdx.x);
}

// stage 2 of the tall case: the (stripes, n) fp32 partials.  n is small here (<= 2048 columns), so the parallelism
// has to come from the stripe dimension too: a workgroup is 32 columns x 8 stripe segments (128-byte row reads),
// LDS combine in a fixed order.
__device__ __forceinline__ void sum_rows_partials_body(const float* __restrict__ ws, float* __restrict__ out,
                                                       int stripes, long n, int accumulate, long blk, float (*part)[32]) {
  const int lane = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const long col = blk * 32 + lane;
  float s = 0.0f;
  if (col < n) {
#pragma unroll 4
    for (int t = seg; t < stripes; t += 8) s += ws[(long)t * n + col];
  }
  part[seg][lane] = s;
  __syncthreads();
  if (seg == 0 && col < n) {
    float t = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
              ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
    if (accumulate) t += out[col];
    out[col] = t;
  }
}

__global__ __launch_bounds__(256) void sum_rows_partials_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                int stripes, long n, int accumulate) {
  __shared__ float part[8][32];
  sum_rows_partials_body(ws, out, stripes, n, accumulate, blockIdx.x, part);
}

// Both parameter gradients of one Linear in one launch: workgroups [0, nb_bias) finish the bias gradient (stage 2 of
// the tall column sum of grad_y), the rest reduce the (S, N*K) split-T weight-gradient partials.
template <typename T>
__global__ __launch_bounds__(256) void linear_param_grads_kernel(const float* __restrict__ ws, float* __restrict__ gb,
                                                                 int stripes, long n_b, int acc_b, int nb_bias,
                                                                 const T* __restrict__ wpart, float* __restrict__ gw,
                                                                 int S, long n_w, int acc_w) {
  __shared__ float part[8][32];
  if ((int)blockIdx.x < nb_bias) sum_rows_partials_body(ws, gb, stripes, n_b, acc_b, blockIdx.x, part);
  else sum_rows_flat_body<T>(wpart, gw, S, n_w, acc_w, (long)blockIdx.x - nb_bias);
}

struct TallPlan {
  int cvb, ty, gy, stripes;
  long rows_per_stripe;
};

static TallPlan plan_tall(long S, long n) {
  TallPlan p;
  const long cv = n / 8;
  p.cvb = (int)std::min<long>(cv, 256);
  p.ty = std::max(1, 256 / p.cvb);
  p.gy = (int)((cv + p.cvb - 1) / p.cvb);
  // enough workgroups to fill 256 CUs about three times over, at least 8*ty rows each
  long stripes = std::min<long>(kMaxSumStripes, std::max<long>(1, (256L * 3) / p.gy));
  stripes = std::max<long>(1, std::min<long>(stripes, S / (8L * p.ty)));
  p.rows_per_stripe = (S + stripes - 1) / stripes;
  p.stripes = (int)((S + p.rows_per_stripe - 1) / p.rows_per_stripe);
  return p;
}

template <typename T>
static int launch_sum_rows(const void* x, float* out, float* ws, long S, long n, int accumulate, hipStream_t st) {
  const int fgrid = cdiv(cdiv(n, 4), 256);
  if (S <= kFlatMaxS) {
    hipLaunchKernelGGL((sum_rows_flat_kernel<T>), dim3(fgrid), dim3(256), 0, st, (const T*)x, out, (int)S, n, accumulate);
    return check_launch("sum_rows_flat_kernel");
  }
  const TallPlan p = plan_tall(S, n);
  hipLaunchKernelGGL((sum_rows_tall_kernel<T>), dim3(p.stripes, p.gy), dim3(p.cvb * p.ty), 0, st, (const T*)x, ws, S, n,
                     p.cvb, p.ty, p.rows_per_stripe);
  if (int rc = check_launch("sum_rows_tall_kernel")) return rc;
  hipLaunchKernelGGL(sum_rows_partials_kernel, dim3(cdiv(n, 32)), dim3(256), 0, st, (const float*)ws, out, p.stripes, n,
                     accumulate);
  return check_launch("sum_rows_partials_kernel");
}

}  // namespace rfn

using namespace rfn;

extern "C" {

unsigned long rfn_sum_rows_workspace_bytes(long S, long n) {
  if (S <= kFlatMaxS || n <= 0 || n % 8 != 0) return 0;
  return (unsigned long)plan_tall(S, n).stripes * (unsigned long)n * sizeof(float);
}

// Bias gradient (column sum of grad_y (T, N)) and reduction of the (S, N*K) weight-gradient partials of one Linear:
// two launches (tall stage 1, then one combined kernel).  T must be > 64 (else call rfn_sum_rows twice).
int rfn_linear_param_grads(const void* grad_y, float* grad_bias, void* workspace, long T, long N, int acc_bias,
                           const void* w_partials, float* grad_weight, int S, long NK, int acc_weight, int dtype,
                           rfn_stream_t stream) {
  RFN_REQUIRE(grad_y && grad_bias && workspace && w_partials && grad_weight, "rfn_linear_param_grads: null pointer");
  RFN_REQUIRE(T > kFlatMaxS && N > 0 && N % 8 == 0 && S > 0 && S <= kFlatMaxS && NK > 0 && NK % 8 == 0,
              "rfn_linear_param_grads: need T > 64, 0 < S <= 64, N and N*K multiples of 8");
  RFN_REQUIRE(dtype == 0 || dtype == 1, "rfn_linear_param_grads: dtype must be 0 (f32) or 1 (bf16)");
  hipStream_t st = (hipStream_t)stream;
  const TallPlan p = plan_tall(T, N);
  const int nb_bias = cdiv(N, 32), nb_w = cdiv(cdiv(NK, 4), 256);
  if (dtype == 0) {
    hipLaunchKernelGGL((sum_rows_tall_kernel<float>), dim3(p.stripes, p.gy), dim3(p.cvb * p.ty), 0, st,
                       (const float*)grad_y, (float*)workspace, T, N, p.cvb, p.ty, p.rows_per_stripe);
    hipLaunchKernelGGL((linear_param_grads_kernel<float>), dim3(nb_bias + nb_w), dim3(256), 0, st,
                       (const float*)workspace, grad_bias, p.stripes, N, acc_bias, nb_bias, (const float*)w_partials,
                       grad_weight, S, NK, acc_weight);
  } else {
    hipLaunchKernelGGL((sum_rows_tall_kernel<__hip_bfloat16>), dim3(p.stripes, p.gy), dim3(p.cvb * p.ty), 0, st,
                       (const __hip_bfloat16*)grad_y, (float*)workspace, T, N, p.cvb, p.ty, p.rows_per_stripe);
    hipLaunchKernelGGL((linear_param_grads_kernel<__hip_bfloat16>), dim3(nb_bias + nb_w), dim3(256), 0, st,
                       (const float*)workspace, grad_bias, p.stripes, N, acc_bias, nb_bias,
                       (const __hip_bfloat16*)w_partials, grad_weight, S, NK, acc_weight);
  }
  return check_launch("linear_param_grads_kernel");
}

int rfn_sum_rows(const void* x, float* out, void* workspace, long S, long n, int x_dtype, int accumulate,
                 rfn_stream_t stream) {
  RFN_REQUIRE(x && out, "rfn_sum_rows: null pointer");
  RFN_REQUIRE(S > 0 && n > 0 && n % 8 == 0, "rfn_sum_rows: need S > 0 and n a positive multiple of 8 (got %ld, %ld)", S, n);
  RFN_REQUIRE(S <= kFlatMaxS || workspace, "rfn_sum_rows: workspace required for S > %d", kFlatMaxS);
  if (x_dtype == 0) return launch_sum_rows<float>(x, out, (float*)workspace, S, n, accumulate, (hipStream_t)stream);
  if (x_dtype == 1)
    return launch_sum_rows<__hip_bfloat16>(x, out, (float*)workspace, S, n, accumulate, (hipStream_t)stream);
  return fail(RFN_EINVAL, "rfn_sum_rows: x_dtype must be 0 (f32) or 1 (bf16)");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// Multi-tensor cast fp32 -> bf16: refresh of the cached bf16 copies of ~1 000 parameter tensors after an optimizer step
// / EMA update in ONE launch (torch._foreach_copy_ with a dtype change degenerates into one tiny kernel per tensor:
// 2 x 1 000 launches per step).  `table` (device memory, built once per parameter set by the host) holds one entry per
// chunk of <= kCastChunk elements: {src, dst, n}.  Round-to-nearest-even, as torch's conversion.
// ---------------------------------------------------------------------------------------------------------------------
namespace rfn {

struct CastChunk {
  const float* src;
  unsigned short* dst;
  long n;
};

__device__ __forceinline__ unsigned f32_to_bf16_bits(float f) { return bf16_bits(f); }   // common.h

__global__ __launch_bounds__(256) void multi_cast_f32_bf16_kernel(const CastChunk* __restrict__ table) {
  const CastChunk c = table[blockIdx.x];
  const bool aligned = (((size_t)c.src & 15) == 0) && (((size_t)c.dst & 7) == 0);
  long i = (long)threadIdx.x * 4;
  if (aligned) {
    for (; i + 3 < c.n; i += 256 * 4) {
      const float4 v = *reinterpret_cast<const float4*>(c.src + i);
      uint2 o;
      o.x = bf16x2_bits(v.x, v.y);
      o.y = bf16x2_bits(v.z, v.w);
      *reinterpret_cast<uint2*>(c.dst + i) = o;
    }
  }
  // tail (and unaligned tensors): scalar
  for (long j = aligned ? (c.n & ~3L) + threadIdx.x : threadIdx.x; j < c.n; j += 256)
    c.dst[j] = (unsigned short)f32_to_bf16_bits(c.src[j]);
}

// Layout-changing copies of a parameter set in one launch: dst (contiguous, up to 4-D, fp32 / bf16 / f16) <- a strided fp32 view
// of a parameter (the (Co, r, r, C) patch-GEMM form of a convolution weight, the tap-major depthwise weights, W^T of the patch
// GEMMs ...: torch._foreach_copy_ runs one tiny strided-copy kernel per tensor -- 240 of them after every optimiser / EMA
// step, on the serial tail of the training step).  A chunk = kPermChunk consecutive dst elements of one tensor.
struct PermChunk {
  const float* src;
  void* dst;
  long n1, n2, n3;             // extents of dims 1..3 of dst (dim 0 is implied)
  long s0, s1, s2, s3;         // strides of the source view, in elements
  long off, cnt;               // first dst element of the chunk, elements in it
  long dtype;                  // of dst: 0 fp32, 1 bf16, 2 f16
};
constexpr int kPermChunk = 8192;

__global__ __launch_bounds__(256) void multi_permute_cast_kernel(const PermChunk* __restrict__ table) {
  const PermChunk c = table[blockIdx.x];
  // 32-bit index arithmetic (a parameter tensor has < 2^31 elements): three 64-bit divisions per element were the whole
  // cost of this kernel (177 us per launch for ~100 MB)
  const unsigned n1 = (unsigned)c.n1, n2 = (unsigned)c.n2, n3 = (unsigned)c.n3, cnt = (unsigned)c.cnt;
  for (unsigned j = threadIdx.x; j < cnt; j += 256) {
    const unsigned i = (unsigned)c.off + j;
    const unsigned i3 = i % n3, r3 = i / n3, i2 = r3 % n2, r2 = r3 / n2, i1 = r2 % n1, i0 = r2 / n1;
    const float v = c.src[(long)i0 * c.s0 + (long)i1 * c.s1 + (long)i2 * c.s2 + (long)i3 * c.s3];
    if (c.dtype == 0) ((float*)c.dst)[i] = v;
    else if (c.dtype == 1) ((unsigned short*)c.dst)[i] = (unsigned short)bf16_bits(v);
    else ((_Float16*)c.dst)[i] = (_Float16)v;
  }
}

// EMA of a whole parameter set in one launch (models/segmentation_model.py:676-689: teacher <- m teacher + (1 - m) student)
// with the bf16 copy of the updated teacher weight written in the same pass where the chunk has one (dst16 != null).
struct EmaChunk {
  float* ema;
  const float* live;
  unsigned short* dst16;
  long n;
};

__global__ __launch_bounds__(256) void multi_ema_kernel(const EmaChunk* __restrict__ table, float m, float one_minus_m) {
  const EmaChunk c = table[blockIdx.x];
  const bool aligned = (((size_t)c.ema & 15) == 0) && (((size_t)c.live & 15) == 0) && (((size_t)c.dst16 & 7) == 0);
  long i = (long)threadIdx.x * 4;
  if (aligned) {
    for (; i + 3 < c.n; i += 256 * 4) {
      float4 e = *reinterpret_cast<const float4*>(c.ema + i);
      const float4 l = *reinterpret_cast<const float4*>(c.live + i);
      e.x = e.x * m + l.x * one_minus_m; e.y = e.y * m + l.y * one_minus_m;
      e.z = e.z * m + l.z * one_minus_m; e.w = e.w * m + l.w * one_minus_m;
      *reinterpret_cast<float4*>(c.ema + i) = e;
      if (c.dst16 != nullptr) {
        uint2 o;
        o.x = bf16x2_bits(e.x, e.y);
        o.y = bf16x2_bits(e.z, e.w);
        *reinterpret_cast<uint2*>(c.dst16 + i) = o;
      }
    }
  }
  for (long j = aligned ? (c.n & ~3L) + threadIdx.x : threadIdx.x; j < c.n; j += 256) {
    const float e = c.ema[j] * m + c.live[j] * one_minus_m;
    c.ema[j] = e;
    if (c.dst16 != nullptr) c.dst16[j] = (unsigned short)f32_to_bf16_bits(e);
  }
}

// Transposed 16-bit copies of a set of fp32 matrices in one launch (the W^T operands of the input-gradient GEMMs,
// refreshed after every optimiser / EMA update): one kTransposeTile x kTransposeTile tile per workgroup, rows read coalesced,
// transposed through LDS, columns written coalesced.  dst[k][n] = bf16(src[n][k]), src (N, K) row-major, dst (K, N) row-major.
struct TransposeTile {
  const float* src;
  unsigned short* dst;
  int N, K, n0, k0;
};

constexpr int kTransposeTile = 64;     // rfn_multi_transpose_tile()

__global__ __launch_bounds__(256) void multi_transpose_cast_kernel(const TransposeTile* __restrict__ table) {
  // 64 x 64 tile: rows of 64 floats read as 16-byte pieces (256 B per row), columns written as 16-byte pieces of 8 bf16 (128 B per
  // destination row).  (Round 5; 32 x 32 tiles with 2-byte stores: 0.94 ms for MiT-B5's 82 M weights, 64-byte write segments.)
  constexpr int T = kTransposeTile;
  __shared__ float tile[T][T + 1];
  const TransposeTile t = table[blockIdx.x];
  const bool vec_in = (t.K & 3) == 0 && (((size_t)t.src) & 15) == 0;
#pragma unroll
  for (int it = 0; it < T * T / 4 / 256; ++it) {
    const int idx = threadIdx.x + 256 * it, r = idx / (T / 4), c4 = idx % (T / 4);
    const int n = t.n0 + r, k = t.k0 + 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < t.N) {
      if (vec_in && k + 3 < t.K) {
        v = *reinterpret_cast<const float4*>(t.src + (size_t)n * t.K + k);
      } else {
        if (k < t.K) v.x = t.src[(size_t)n * t.K + k];
        if (k + 1 < t.K) v.y = t.src[(size_t)n * t.K + k + 1];
        if (k + 2 < t.K) v.z = t.src[(size_t)n * t.K + k + 2];
        if (k + 3 < t.K) v.w = t.src[(size_t)n * t.K + k + 3];
      }
    }
    tile[r][4 * c4] = v.x; tile[r][4 * c4 + 1] = v.y; tile[r][4 * c4 + 2] = v.z; tile[r][4 * c4 + 3] = v.w;
  }
  __syncthreads();
  const bool vec_out = (t.N & 7) == 0 && (((size_t)t.dst) & 15) == 0;
#pragma unroll
  for (int it = 0; it < T * T / 8 / 256; ++it) {
    const int idx = threadIdx.x + 256 * it, r = idx / (T / 8), c8 = idx % (T / 8);
    const int k = t.k0 + r, n = t.n0 + 8 * c8;
    if (k >= t.K || n >= t.N) continue;
    if (vec_out && n + 7 < t.N) {
      uint4 o;
      o.x = bf16x2_bits(tile[8 * c8][r], tile[8 * c8 + 1][r]);
      o.y = bf16x2_bits(tile[8 * c8 + 2][r], tile[8 * c8 + 3][r]);
      o.z = bf16x2_bits(tile[8 * c8 + 4][r], tile[8 * c8 + 5][r]);
      o.w = bf16x2_bits(tile[8 * c8 + 6][r], tile[8 * c8 + 7][r]);
      *reinterpret_cast<uint4*>(t.dst + (size_t)k * t.N + n) = o;
    } else {
      for (int e = 0; e < 8 && n + e < t.N; ++e)
        t.dst[(size_t)k * t.N + n + e] = (unsigned short)f32_to_bf16_bits(tile[8 * c8 + e][r]);
    }
  }
}

// AdamW over a whole parameter set in one launch (torch.optim.AdamW, decoupled weight decay, no amsgrad / maximize; the
// arithmetic of its fused implementation, fp32):  p -= lr wd p;  m = m + (1 - b1) (g - m);  v = b2 v + (1 - b2) g^2;
// p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps).  The host passes the per-group scalars of this step.
struct AdamChunk {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;          // low 56 bits: elements; the group index rides in the top byte
};
struct AdamGroups {
  float lr[8], beta1[8], beta2[8], eps[8], wd[8], bc1[8], bc2_sqrt[8], omb1[8], omb2[8];   // omb = 1 - beta, rounded from double
};

__global__ __launch_bounds__(256) void multi_adamw_kernel(const AdamChunk* __restrict__ table, AdamGroups gr) {
  const AdamChunk c = table[blockIdx.x];
  const int gi = (int)((unsigned long)c.n >> 56);
  const long n = c.n & ((1L << 56) - 1);
  const float lr = gr.lr[gi], b2 = gr.beta2[gi], eps = gr.eps[gi], wd = gr.wd[gi];
  const float step_size = lr / gr.bc1[gi], bc2s = gr.bc2_sqrt[gi], omb1 = gr.omb1[gi], omb2 = gr.omb2[gi];
  auto upd = [&](float& p, float g, float& m, float& v) {
    p -= lr * wd * p;
    m = m + omb1 * (g - m);
    v = b2 * v + omb2 * g * g;
    p -= step_size * m / (sqrtf(v) / bc2s + eps);
  };
  const bool aligned = ((((size_t)c.p | (size_t)c.g | (size_t)c.m | (size_t)c.v) & 15) == 0);
  long i = (long)threadIdx.x * 4;
  if (aligned) {
    for (; i + 3 < n; i += 256 * 4) {
      float4 p = *reinterpret_cast<const float4*>(c.p + i), m = *reinterpret_cast<const float4*>(c.m + i);
      float4 v = *reinterpret_cast<const float4*>(c.v + i);
      const float4 g = *reinterpret_cast<const float4*>(c.g + i);
      upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(c.p + i) = p;
      *reinterpret_cast<float4*>(c.m + i) = m;
      *reinterpret_cast<float4*>(c.v + i) = v;
    }
  }
  for (long j = aligned ? (n & ~3L) + threadIdx.x : threadIdx.x; j < n; j += 256) {
    float p = c.p[j], m = c.m[j], v = c.v[j];
    upd(p, c.g[j], m, v);
    c.p[j] = p; c.m[j] = m; c.v[j] = v;
  }
}

}  // namespace rfn

extern "C" {

int rfn_multi_transpose_cast_f32_bf16(const void* table, int ntiles, rfn_stream_t stream) {
  RFN_REQUIRE(table && ntiles > 0, "rfn_multi_transpose_cast_f32_bf16: empty table");
  hipLaunchKernelGGL(rfn::multi_transpose_cast_kernel, dim3(ntiles), dim3(256), 0, (hipStream_t)stream,
                     (const rfn::TransposeTile*)table);
  return rfn::check_launch("multi_transpose_cast_kernel");
}

int rfn_multi_adamw_f32(const void* table, int nchunks, const float* group_args, int ngroups, rfn_stream_t stream) {
  RFN_REQUIRE(table && nchunks > 0 && group_args && ngroups > 0 && ngroups <= 8, "rfn_multi_adamw_f32: bad arguments");
  rfn::AdamGroups gr{};
  for (int g = 0; g < ngroups; ++g) {                   // host array, 9 floats per group
    const float* a = group_args + 9 * g;
    gr.lr[g] = a[0]; gr.beta1[g] = a[1]; gr.beta2[g] = a[2]; gr.eps[g] = a[3]; gr.wd[g] = a[4]; gr.bc1[g] = a[5];
    gr.bc2_sqrt[g] = a[6]; gr.omb1[g] = a[7]; gr.omb2[g] = a[8];
  }
  hipLaunchKernelGGL(rfn::multi_adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const rfn::AdamChunk*)table, gr);
  return rfn::check_launch("multi_adamw_kernel");
}

int rfn_multi_ema_f32(const void* table, int nchunks, float momentum, rfn_stream_t stream) {
  RFN_REQUIRE(table && nchunks > 0, "rfn_multi_ema_f32: empty table");
  hipLaunchKernelGGL(rfn::multi_ema_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const rfn::EmaChunk*)table, momentum, 1.0f - momentum);
  return rfn::check_launch("multi_ema_kernel");
}

int rfn_multi_cast_chunk_elems(void) { return 16384; }

int rfn_multi_transpose_tile(void) { return rfn::kTransposeTile; }

int rfn_multi_permute_chunk_elems(void) { return rfn::kPermChunk; }

// table: nchunks rows of 12 int64 {src, dst, n1, n2, n3, s0, s1, s2, s3, off, cnt, dtype} (struct PermChunk above)
int rfn_multi_permute_cast_f32(const void* table, int nchunks, rfn_stream_t stream) {
  RFN_REQUIRE(table && nchunks > 0, "rfn_multi_permute_cast_f32: empty table");
  hipLaunchKernelGGL(rfn::multi_permute_cast_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const rfn::PermChunk*)table);
  return rfn::check_launch("multi_permute_cast_kernel");
}

int rfn_multi_cast_f32_bf16(const void* table, int nchunks, rfn_stream_t stream) {
  RFN_REQUIRE(table && nchunks > 0, "rfn_multi_cast_f32_bf16: empty table");
  hipLaunchKernelGGL(rfn::multi_cast_f32_bf16_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                     (const rfn::CastChunk*)table);
  return rfn::check_launch("multi_cast_f32_bf16_kernel");
}

}  // extern "C"
