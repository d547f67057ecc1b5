// refign_amd/csrc/bn.hip -- training-mode BatchNorm2d (+ ReLU) on channels-last 16-bit tensors, forward and backward:
// the norm + activation of every ConvBNReLU of the decode heads (models/modules.py:16-56; daformer.py:65-126,
// segformer.py:62-70), which run with BATCH statistics in the student AND in the EMA teacher.
//
//   forward   stats pass : sums[0][c] = sum_t x[t,c], sums[1][c] = sum_t x[t,c]^2          (fp32 atomics, pre-zeroed)
//             apply pass : y = relu?( (x - mean) * rstd * gamma + beta ), running statistics updated by block 0
//   backward  stats pass : sums[0][c] = sum_t g'[t,c], sums[1][c] = sum_t g'[t,c] * xhat[t,c],  g' = g * (y > 0)
//             apply pass : dx = gamma * rstd * ( g' - (sums[0] + xhat * sums[1]) / T )
// The two-pass split is the data dependence of BatchNorm itself (3 tensor passes forward, 5 backward -- what any
// implementation moves); what is fused away are the separate ReLU / ReLU-backward passes and the per-call
// normalisation-constant kernels.  The sums are exactly what a data-parallel SyncBatchNorm all-reduces (one (2, C)
// vector per pass), should the host choose to.  Lanes run along channels (16-byte vectors), rows are strided over lanes.
#include <hip/hip_bf16.h>

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
__device__ __forceinline__ void norm_consts(const float* sums, int C, int c, float invT, float eps, float& mean, float& rstd) {
  mean = sums[c] * invT;
  const float var = fmaxf(sums[C + c] * invT - mean * mean, 0.f);
  rstd = rsqrtf(var + eps);
}

// BWD = false: sums of x and x^2.  BWD = true: sums of g' and g' * xhat (fwd_sums give mean / rstd, gamma / beta the sign of y)
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       const float* __restrict__ fwd_sums, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ sums, long T,
                                                       int C, int cvb, float eps, int relu) {
  __shared__ float red[2][256][8];
  const int CV = C / 8, pl = 256 / cvb;
  const int cv = blockIdx.x * cvb + threadIdx.x % cvb, rl = threadIdx.x / cvb;
  float s0[8] = {0}, s1[8] = {0};
  if (cv < CV) {
    const int c0 = cv * 8;
    float mean[8], a[8], b[8];
    if (BWD) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float r;
        norm_consts(fwd_sums, C, c0 + i, 1.f / (float)T, eps, mean[i], r);
        a[i] = r;                                                       // xhat = (x - mean) * rstd
        b[i] = (gamma ? gamma[c0 + i] : 1.f);
      }
    }
    for (long t = (long)blockIdx.y * pl + rl; t < T; t += (long)gridDim.y * pl) {
      float xv[8];
      load8<DT>(x + t * C + c0, xv);
      if (!BWD) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          s0[i] += xv[i];
          s1[i] += xv[i] * xv[i];
        }
      } else {
        float gv[8];
        load8<DT>(g + t * C + c0, gv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (xv[i] - mean[i]) * a[i];
          const float z = xh * b[i] + (beta ? beta[c0 + i] : 0.f);
          const float gp = (relu && z <= 0.f) ? 0.f : gv[i];
          s0[i] += gp;
          s1[i] += gp * xh;
        }
      }
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
    float sum = 0.f;
    for (int r = 0; r < pl; ++r) sum += red[which][r * cvb + v][e];
    atomicAdd(sums + which * C + cvg * 8 + e, sum);
  }
}

template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ g,
                                                       const float* __restrict__ fwd_sums, const float* __restrict__ bwd_sums,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       uint16_t* __restrict__ out, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, long T, int C, int cvb, float eps,
                                                       float momentum, int relu) {
  const int CV = C / 8, pl = 256 / cvb;
  const int cv = blockIdx.x * cvb + threadIdx.x % cvb, rl = threadIdx.x / cvb;
  const float invT = 1.f / (float)T;
  if (!BWD && running_mean != nullptr && blockIdx.y == 0 && rl == 0 && cv < CV) {
    // running statistics: mean and UNBIASED variance of this batch, as nn.BatchNorm2d
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cv * 8 + i;
      const float mean = fwd_sums[c] * invT;
      const float var = fmaxf(fwd_sums[C + c] * invT - mean * mean, 0.f);
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * var * ((float)T / fmaxf((float)T - 1.f, 1.f));
    }
  }
  if (cv >= CV) return;
  const int c0 = cv * 8;
  float mean[8], rstd[8], gm[8], bt[8], k0[8], k1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    norm_consts(fwd_sums, C, c0 + i, invT, eps, mean[i], rstd[i]);
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
        o[i] = relu ? fmaxf(z, 0.f) : z;
      }
    } else {
      float gv[8];
      load8<DT>(g + t * C + c0, gv);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (xv[i] - mean[i]) * rstd[i];
        const float z = xh * gm[i] + bt[i];
        const float gp = (relu && z <= 0.f) ? 0.f : gv[i];
        o[i] = gm[i] * rstd[i] * (gp - k0[i] - xh * k1[i]);
      }
    }
    store8<DT>(out + t * C + c0, o);
  }
}

static inline int bn_cvb(int CV) { return CV >= 64 ? 64 : (CV >= 32 ? 32 : (CV >= 16 ? 16 : 8)); }

template <int DT>
static int bn_launch(bool bwd, bool apply, const void* x, const void* g, const float* fwd_sums, float* sums_or_bwd,
                     const float* gamma, const float* beta, void* out, float* rmean, float* rvar, long T, int C, float eps,
                     float momentum, int relu, hipStream_t s) {
  const int CV = C / 8, cvb = bn_cvb(CV), gx = cdiv(CV, cvb), pl = 256 / cvb;
  const int gy = (int)std::max<long>(1, std::min<long>(cdiv(T, (long)pl * 4), (256L * 8) / gx));
  dim3 grid(gx, gy), block(256);
  if (!apply) {
    if (bwd)
      hipLaunchKernelGGL((bn_stats_kernel<DT, true>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)g, fwd_sums,
                         gamma, beta, sums_or_bwd, T, C, cvb, eps, relu);
    else
      hipLaunchKernelGGL((bn_stats_kernel<DT, false>), grid, block, 0, s, (const uint16_t*)x, nullptr, nullptr, nullptr,
                         nullptr, sums_or_bwd, T, C, cvb, eps, relu);
  } else {
    if (bwd)
      hipLaunchKernelGGL((bn_apply_kernel<DT, true>), grid, block, 0, s, (const uint16_t*)x, (const uint16_t*)g, fwd_sums,
                         sums_or_bwd, gamma, beta, (uint16_t*)out, nullptr, nullptr, T, C, cvb, eps, momentum, relu);
    else
      hipLaunchKernelGGL((bn_apply_kernel<DT, false>), grid, block, 0, s, (const uint16_t*)x, nullptr, fwd_sums, nullptr,
                         gamma, beta, (uint16_t*)out, rmean, rvar, T, C, cvb, eps, momentum, relu);
  }
  return check_launch("bn kernel");
}

}  // namespace rfn

extern "C" {
using namespace rfn;

int rfn_bn_train_fwd(const void* x, const float* gamma, const float* beta, void* y, float* sums, float* running_mean,
                     float* running_var, long T, int C, float eps, float momentum, int relu, int dtype,
                     rfn_stream_t stream) {
  RFN_REQUIRE(x && y && sums, "bn_train_fwd: null pointer");
  RFN_REQUIRE(T > 1 && C > 0 && C % 8 == 0, "bn_train_fwd: T=%ld C=%d (C %% 8)", T, C);
  RFN_REQUIRE(dtype == 1 || dtype == 2, "bn_train_fwd: dtype %d (1 = bf16, 2 = f16)", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(float), s) != hipSuccess) return fail(RFN_ELAUNCH, "bn_train_fwd: memset");
  int rc = dtype == 1 ? bn_launch<1>(false, false, x, nullptr, nullptr, sums, gamma, beta, nullptr, nullptr, nullptr, T, C, eps, momentum, relu, s)
                      : bn_launch<2>(false, false, x, nullptr, nullptr, sums, gamma, beta, nullptr, nullptr, nullptr, T, C, eps, momentum, relu, s);
  if (rc != RFN_OK) return rc;
  return dtype == 1 ? bn_launch<1>(false, true, x, nullptr, sums, nullptr, gamma, beta, y, running_mean, running_var, T, C, eps, momentum, relu, s)
                    : bn_launch<2>(false, true, x, nullptr, sums, nullptr, gamma, beta, y, running_mean, running_var, T, C, eps, momentum, relu, s);
}

int rfn_bn_train_bwd(const void* x, const void* grad_y, const float* fwd_sums, const float* gamma, const float* beta,
                     void* grad_x, float* bwd_sums, long T, int C, float eps, int relu, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && fwd_sums && grad_x && bwd_sums, "bn_train_bwd: null pointer");
  RFN_REQUIRE(T > 1 && C > 0 && C % 8 == 0, "bn_train_bwd: T=%ld C=%d (C %% 8)", T, C);
  RFN_REQUIRE(dtype == 1 || dtype == 2, "bn_train_bwd: dtype %d (1 = bf16, 2 = f16)", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(bwd_sums, 0, 2 * (size_t)C * sizeof(float), s) != hipSuccess) return fail(RFN_ELAUNCH, "bn_train_bwd: memset");
  int rc = dtype == 1 ? bn_launch<1>(true, false, x, grad_y, fwd_sums, bwd_sums, gamma, beta, nullptr, nullptr, nullptr, T, C, eps, 0.f, relu, s)
                      : bn_launch<2>(true, false, x, grad_y, fwd_sums, bwd_sums, gamma, beta, nullptr, nullptr, nullptr, T, C, eps, 0.f, relu, s);
  if (rc != RFN_OK) return rc;
  return dtype == 1 ? bn_launch<1>(true, true, x, grad_y, fwd_sums, bwd_sums, gamma, beta, grad_x, nullptr, nullptr, T, C, eps, 0.f, relu, s)
                    : bn_launch<2>(true, true, x, grad_y, fwd_sums, bwd_sums, gamma, beta, grad_x, nullptr, nullptr, T, C, eps, 0.f, relu, s);
}

}  // extern "C"
