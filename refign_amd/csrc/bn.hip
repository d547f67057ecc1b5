// refign_amd/csrc/bn.hip -- training-mode BatchNorm2d (+ ReLU) on channels-last 16-bit tensors, forward and backward:
// the norm + activation of every ConvBNReLU of the decode heads (models/modules.py:16-56; daformer.py:65-126,
// segformer.py:62-70), which run with BATCH statistics in the student AND in the EMA teacher.
//
//   forward   stats pass : sums[0][c] = sum_t x[t,c], sums[1][c] = sum_t x[t,c]^2          (fp64, pre-zeroed)
//             apply pass : y = relu?( (x - mean) * rstd * gamma + beta ), running statistics updated by block 0
//   backward  stats pass : sums[0][c] = sum_t g'[t,c], sums[1][c] = sum_t g'[t,c] * xhat[t,c],  g' = g * (y > 0)
//             apply pass : dx = gamma * rstd * ( g' - (sums[0] + xhat * sums[1]) / T )
// The two-pass split is the data dependence of BatchNorm itself (3 tensor passes forward, 5 backward -- what any
// implementation moves); what is fused away are the separate ReLU / ReLU-backward passes and the per-call
// normalisation-constant kernels.  Lanes run along channels (16-byte vectors), rows are strided over lanes.
//
// The forward sums are DOUBLES from the first addition to the atomics: the variance is E[x^2] - mean^2, which cancels
// catastrophically in fp32 once |mean| >> std over ~1e6 rows (torch's BatchNorm uses Welford / two passes for that
// reason); in fp64 the one-pass formula has 29 more bits than the 16-bit inputs can use, and the order of the atomics no
// longer shows in the fp32 constants derived from the sums.  The kernels stay HBM-bound (16 fp64 operations per 16-byte
// load).  The backward sums have no cancellation and stay fp32.
//
// Data parallelism (SyncBatchNorm, torch/nn/modules/_functions.py: the reference trains with `sync_batchnorm: True`): the
// statistics buffer is 2 C + 1 doubles, the last one the number of rows the sums were taken over -- the stats pass adds
// its own T there -- so that ONE all-reduce (SUM) of the buffer between the stats pass and the apply pass turns local
// statistics into global ones; the apply passes normalise with the count they find in the buffer, never with T.  The
// four phases are separate entry points for that (rfn_bn_stats_fwd / _apply_fwd / _stats_bwd / _apply_bwd); the
// parameter gradients are the LOCAL backward sums (read before the exchange), as in SyncBatchNorm.
#include <hip/hip_bf16.h>

#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace rfn {

template <int DT> __device__ __forceinline__ void load8(const uint16_t* p, float (&f)[8]) {
  const u32x4 v = *(const u32x4*)p;
  float a[4], b[4];
  unpack4<DT>(u32x2{v[0], v[1]}, a);
  unpack4<DT>(u32x2{v[2], v[3]}, b);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[i] = a[i];
    f[4 + i] = b[i];
  }
}
template <int DT> __device__ __forceinline__ void store8(uint16_t* p, const float (&f)[8]) {
  const u32x2 lo = pack4<DT>(f[0], f[1], f[2], f[3]), hi = pack4<DT>(f[4], f[5], f[6], f[7]);
  *(u32x4*)p = u32x4{lo[0], lo[1], hi[0], hi[1]};
}

// per-channel constants from the forward sums
__device__ __forceinline__ float batch_var(const double* sums, int C, int c, double invT, float& mean) {
  // invT = 1 / sums[2 C]: the (global) number of rows behind the sums
  const double m = sums[c] * invT;
  mean = (float)m;
  return (float)fmax(sums[C + c] * invT - m * m, 0.0);
}
__device__ __forceinline__ void norm_consts(const double* sums, int C, int c, double invT, float eps, float& mean, float& rstd) {
  rstd = rsqrtf(batch_var(sums, C, c, invT, mean) + eps);
}

// BWD = false: sums of x and x^2.  BWD = true: sums of g' and g' * xhat (fwd_sums give mean / rstd, gamma / beta the sign of y)
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       const double* __restrict__ fwd_sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, void* __restrict__ sums_, long T,
                                                       int C, int cvb, float eps, int relu, float slope) {
  using ST = typename std::conditional<BWD, float, double>::type;      // forward sums: doubles (header)
  ST* __restrict__ sums = (ST*)sums_;
  __shared__ ST red[2][256][8];
  const int CV = C / 8, pl = 256 / cvb;
  const int cv = blockIdx.x * cvb + threadIdx.x % cvb, rl = threadIdx.x / cvb;
  ST s0[8] = {0}, s1[8] = {0};
  if (cv < CV) {
    const int c0 = cv * 8;
    float mean[8], a[8], b[8];
    if (BWD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float r;
        norm_consts(fwd_sums, C, c0 + i, 1.0 / fwd_sums[2 * C], eps, mean[i], r);
        a[i] = r;                                                       // xhat = (x - mean) * rstd
        b[i] = (gamma ? gamma[c0 + i] : 1.f);
      }
    }
    float bt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bt[i] = (BWD && beta) ? beta[c0 + i] : 0.f;
    auto accumulate = [&](const float (&xv)[8], const float (&gv)[8]) {
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s0[i] += (ST)xv[i];
          s1[i] += (ST)xv[i] * (ST)xv[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[i] - mean[i]) * a[i];
          const float z = xh * b[i] + bt[i];
          const float gp = (relu && z <= 0.f) ? slope * gv[i] : gv[i];
          s0[i] += gp;
          s1[i] += gp * xh;
        }
      }
    };
    // four rows per iteration, their loads issued together: one row at a time (two 16-byte loads in flight per thread) ran at
    // 1.5-2.1 TB/s on the decode head's backward statistics (section 4.5 of DESIGN.md: memory-level parallelism per thread)
    const long stride = (long)gridDim.y * pl;
    long t = (long)blockIdx.y * pl + rl;
    for (; t + 3 * stride < T; t += 4 * stride) {
      float xv[4][8], gv[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        load8<DT>(x + (t + u * stride) * C + c0, xv[u]);
        if (BWD) load8<DT>(g + (t + u * stride) * C + c0, gv[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accumulate(xv[u], gv[u]);
    }
    for (; t < T; t += stride) {
      float xv[8], gv[8];
      load8<DT>(x + t * C + c0, xv);
      if (BWD) load8<DT>(g + t * C + c0, gv);
      accumulate(xv, gv);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    red[0][threadIdx.x][i] = s0[i];
    red[1][threadIdx.x][i] = s1[i];
  }
  __syncthreads();
  // thread (which, channel vector, element): sum over the row lanes, one atomic per channel
  for (int idx = threadIdx.x; idx < 2 * cvb * 8; idx += 256) {
    const int which = idx / (cvb * 8), rem = idx % (cvb * 8), v = rem / 8, e = rem % 8;
    const int cvg = blockIdx.x * cvb + v;
    if (cvg >= CV) continue;
    ST sum = 0;
    for (int r = 0; r < pl; ++r) sum += red[which][r * cvb + v][e];
    atomicAdd(sums + which * C + cvg * 8 + e, sum);
  }
  if (!BWD && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(sums + 2 * C, (ST)T);
}

template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       const double* __restrict__ fwd_sums, const float* __restrict__ bwd_sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       uint16_t* __restrict__ out, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, long T, int C, int cvb, float eps,
                                                       float momentum, int relu, float slope, long ldo) {
  // ldo: row pitch of `out` in elements (C for a dense result; larger when the result is a channel slice of a wider
  // channels-last tensor -- the branches of the ASPP write straight into their concatenation, daformer.py:110-118)
  const int CV = C / 8, pl = 256 / cvb;
  const int cv = blockIdx.x * cvb + threadIdx.x % cvb, rl = threadIdx.x / cvb;
  const double invTd = 1.0 / fwd_sums[2 * C];          // rows behind the statistics (all ranks')
  const float cnt = (float)fwd_sums[2 * C], invT = (float)invTd;
  if (!BWD && running_mean != nullptr && blockIdx.y == 0 && rl == 0 && cv < CV) {
    // running statistics: mean and UNBIASED variance of this batch, as nn.BatchNorm2d
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cv * 8 + i;
      float mean;
      const float var = batch_var(fwd_sums, C, c, invTd, mean);
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * (cnt / fmaxf(cnt - 1.f, 1.f));
    }
  }
  if (cv >= CV) return;
  const int c0 = cv * 8;
  float mean[8], rstd[8], gm[8], bt[8], k0[8], k1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    norm_consts(fwd_sums, C, c0 + i, invTd, eps, mean[i], rstd[i]);
    gm[i] = gamma ? gamma[c0 + i] : 1.f;
    bt[i] = beta ? beta[c0 + i] : 0.f;
    if (BWD) {
      k0[i] = bwd_sums[c0 + i] * invT;
      k1[i] = bwd_sums[C + c0 + i] * invT;
    }
  }
  for (long t = (long)blockIdx.y * pl + rl; t < T; t += (long)gridDim.y * pl) {
    float xv[8], o[8];
    load8<DT>(x + t * C + c0, xv);
    if (!BWD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = (xv[i] - mean[i]) * rstd[i] * gm[i] + bt[i];
        o[i] = (relu && z <= 0.f) ? slope * z : z;
      }
    } else {
      float gv[8];
      load8<DT>(g + t * C + c0, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        const float z = xh * gm[i] + bt[i];
        const float gp = (relu && z <= 0.f) ? slope * gv[i] : gv[i];
        o[i] = gm[i] * rstd[i] * (gp - k0[i] - xh * k1[i]);
      }
    }
    store8<DT>(out + t * ldo + c0, o);
  }
}

static inline int bn_cvb(int CV) { return CV >= 64 ? 64 : (CV >= 32 ? 32 : (CV >= 16 ? 16 : 8)); }

template <int DT>
static int bn_launch(bool bwd, bool apply, const void* x, const void* g, const double* fwd_sums, void* sums_or_bwd,
                     const float* gamma, const float* beta, void* out, float* rmean, float* rvar, long T, int C, float eps,
                     float momentum, int relu, float slope, hipStream_t s, long ldo = 0) {
  if (ldo == 0) ldo = C;
  const int CV = C / 8, cvb = bn_cvb(CV), gx = cdiv(CV, cvb), pl = 256 / cvb;
  // workgroups per launch.  A statistics workgroup ends with 2 x (its channels) atomics on the SAME 2C sums: 512 of them
  // instead of 2048 (129 600 x 256 backward statistics 68 -> 37 us, 1 296 000 x 256 forward 142 -> 116 us); the apply passes
  // 2048 in isolation (191 vs 166 us at 512) -- and 1024 in the step, where three streams share the CUs (end of round 4, together
  // with the GEMM tile threshold and the fused GELU backward: -1.5 ms, profiles/r04_knob_ab.txt).  RFN_BN_STATS_WGS / RFN_BN_WGS.
  static const long wgs_apply = 1024;
  static const long wgs_stats = 512;
  const long wgs = apply ? wgs_apply : wgs_stats;
  const int gy = (int)std::max<long>(1, std::min<long>(cdiv(T, (long)pl * 4), std::max<long>(1, wgs / gx)));
  dim3 grid(gx, gy), block(256);
  if (!apply) {
    if (bwd)
      hipLaunchKernelGGL((bn_stats_kernel<DT, true>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)g, fwd_sums,
                         gamma, beta, sums_or_bwd, T, C, cvb, eps, relu, slope);
    else
      hipLaunchKernelGGL((bn_stats_kernel<DT, false>), grid, block, 0, s, (const uint16_t*)x, nullptr, nullptr, nullptr,
                         nullptr, sums_or_bwd, T, C, cvb, eps, relu, slope);
  } else {
    if (bwd)
      hipLaunchKernelGGL((bn_apply_kernel<DT, true>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)g, fwd_sums,
                         (const float*)sums_or_bwd, gamma, beta, (uint16_t*)out, nullptr, nullptr, T, C, cvb, eps, momentum, relu, slope, ldo);
    else
      hipLaunchKernelGGL((bn_apply_kernel<DT, false>), grid, block, 0, s, (const uint16_t*)x, nullptr, fwd_sums, nullptr,
                         gamma, beta, (uint16_t*)out, rmean, rvar, T, C, cvb, eps, momentum, relu, slope, ldo);
  }
  return check_launch("bn kernel");
}

}  // namespace rfn

extern "C" {
using namespace rfn;

#define RFN_BN_DISPATCH(...) (dtype == 1 ? bn_launch<1>(__VA_ARGS__) : bn_launch<2>(__VA_ARGS__))

// activation code of the entry points (as rfn_gemm_nt): 0 none, 1 ReLU, 3 LeakyReLU(0.1)
static inline float bn_slope(int act) { return act == 3 ? 0.1f : 0.f; }

static int bn_check(const char* what, long T, int C, int dtype) {
  RFN_REQUIRE(T >= 1 && C > 0 && C % 8 == 0, "%s: T=%ld C=%d (C %% 8)", what, T, C);
  RFN_REQUIRE(dtype == 1 || dtype == 2, "%s: dtype %d (1 = bf16, 2 = f16)", what, dtype);
  return RFN_OK;
}

int rfn_bn_stats_fwd(const void* x, double* sums, long T, int C, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && sums, "bn_stats_fwd: null pointer");
  if (int rc = bn_check("bn_stats_fwd", T, C, dtype)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero_async(sums, (2 * (size_t)C + 1) * sizeof(double), s)) return rc;   // kernel, not a memset node (capi.hip)
  return RFN_BN_DISPATCH(false, false, x, nullptr, nullptr, sums, nullptr, nullptr, nullptr, nullptr, nullptr, T, C, 0.f, 0.f, 0, 0.f, s);
}

int rfn_bn_apply_fwd(const void* x, const float* gamma, const float* beta, void* y, const double* sums, float* running_mean,
                     float* running_var, long T, int C, float eps, float momentum, int relu, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && y && sums, "bn_apply_fwd: null pointer");
  if (int rc = bn_check("bn_apply_fwd", T, C, dtype)) return rc;
  return RFN_BN_DISPATCH(false, true, x, nullptr, sums, nullptr, gamma, beta, y, running_mean, running_var, T, C, eps, momentum,
                         relu != 0, bn_slope(relu), (hipStream_t)stream);
}

// rfn_bn_apply_fwd with the result written at a row pitch of ld_y elements (>= C, multiple of 8): y is a channel slice of a
// wider channels-last tensor
int rfn_bn_apply_fwd_ld(const void* x, const float* gamma, const float* beta, void* y, long ld_y, const double* sums,
                        float* running_mean, float* running_var, long T, int C, float eps, float momentum, int relu, int dtype,
                        rfn_stream_t stream) {
  RFN_REQUIRE(x && y && sums, "bn_apply_fwd_ld: null pointer");
  RFN_REQUIRE(ld_y >= C && ld_y % 8 == 0 && ((size_t)y & 15) == 0, "bn_apply_fwd_ld: pitch %ld (>= C = %d, %% 8, 16-byte base)", ld_y, C);
  if (int rc = bn_check("bn_apply_fwd_ld", T, C, dtype)) return rc;
  return RFN_BN_DISPATCH(false, true, x, nullptr, sums, nullptr, gamma, beta, y, running_mean, running_var, T, C, eps, momentum,
                         relu != 0, bn_slope(relu), (hipStream_t)stream, ld_y);
}

int rfn_bn_stats_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* gamma, const float* beta,
                     float* bwd_sums, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && fwd_sums && bwd_sums, "bn_stats_bwd: null pointer");
  if (int rc = bn_check("bn_stats_bwd", T, C, dtype)) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (int rc = zero_async(bwd_sums, 2 * (size_t)C * sizeof(float), s)) return rc;
  return RFN_BN_DISPATCH(true, false, x, grad_y, fwd_sums, bwd_sums, gamma, beta, nullptr, nullptr, nullptr, T, C, eps, 0.f, relu != 0, bn_slope(relu), s);
}

int rfn_bn_apply_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* bwd_sums, const float* gamma,
                     const float* beta, void* grad_x, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && fwd_sums && bwd_sums && grad_x, "bn_apply_bwd: null pointer");
  if (int rc = bn_check("bn_apply_bwd", T, C, dtype)) return rc;
  return RFN_BN_DISPATCH(true, true, x, grad_y, fwd_sums, (void*)bwd_sums, gamma, beta, grad_x, nullptr, nullptr, T, C, eps,
                         0.f, relu != 0, bn_slope(relu), (hipStream_t)stream);
}

// one rank: stats + apply back to back (`sums`: 2 C + 1 doubles)
int rfn_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, double* sums, float* running_mean,
                     float* running_var, long T, int C, float eps, float momentum, int relu, int dtype,
                     rfn_stream_t stream) {
  RFN_REQUIRE(T > 1, "bn_train_fwd: T=%ld", T);
  if (int rc = rfn_bn_stats_fwd(x, sums, T, C, dtype, stream)) return rc;
  return rfn_bn_apply_fwd(x, gamma, beta, y, sums, running_mean, running_var, T, C, eps, momentum, relu, dtype, stream);
}

int rfn_bn_train_bwd(const void* x, const void* grad_y, const double* fwd_sums, const float* gamma, const float* beta,
                     void* grad_x, float* bwd_sums, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream) {
  if (int rc = rfn_bn_stats_bwd(x, grad_y, fwd_sums, gamma, beta, bwd_sums, T, C, eps, relu, dtype, stream)) return rc;
  return rfn_bn_apply_bwd(x, grad_y, fwd_sums, bwd_sums, gamma, beta, grad_x, T, C, eps, relu, dtype, stream);
}

}  // extern "C"
