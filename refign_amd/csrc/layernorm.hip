// refign_amd/csrc/layernorm.hip -- LayerNorm over the channel dim of (rows, C) token matrices, forward + backward.
//
// Where it sits in the reference: every MiT block has two LayerNorms (eps 1e-6) plus one per patch embedding, per
// spatial-reduction branch and per stage output (models/backbones/mix_transformer.py:135,188-207,234,369-419): 213
// LayerNorms per MiT-B5 forward.  Under bf16 autocast the library path runs them in fp32 and the surrounding linears in
// bf16, so every LayerNorm is followed by a separate fp32->bf16 cast pass and the residual stream stays fp32
// (14 bytes of traffic per element where 4-6 are needed; ~100 ms of a 600 ms step incl. the casts).  This kernel does
// the statistics in fp32 registers and reads/writes the activation dtype directly (fp32 or bf16 in, fp32 or bf16 out),
// so the residual stream can live in bf16.  Pure HBM-bound row reductions:
//   * one wave per row, lanes strided over C (C <= 1024): the row lives in registers (<= 16 values per lane), mean and
//     variance by two wave butterflies (two-pass variance, no E[x^2]-E[x]^2 cancellation);
//   * backward: the same pass produces dx and per-wave register partials of dgamma/dbeta over the wave's rows; partials
//     are combined per workgroup through LDS and written to a workspace row, reduced by a second tiny kernel in a
//     fixed order (deterministic, no atomics).
#include <hip/hip_bf16.h>

#include "common.h"
#include "mfma.h"

namespace rfn {

constexpr int kLnMaxPerLane = 16;   // C <= 1024
constexpr int kLnMaxBlocks = 256;   // workspace rows for dgamma/dbeta partials (one workgroup per CU)

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<__hip_bfloat16>(const __hip_bfloat16* p) {
  return __uint_as_float(((unsigned)*reinterpret_cast<const unsigned short*>(p)) << 16);
}
template <typename T>
__device__ __forceinline__ void st(T* p, float v);
template <>
__device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st<__hip_bfloat16>(__hip_bfloat16* p, float v) {
  *reinterpret_cast<unsigned short*>(p) = (unsigned short)bf16_bits(v);
}

template <>
__device__ __forceinline__ void st<f8e4m3>(f8e4m3* p, float v) { p->v = (unsigned char)(quant4(v, 0.f, 0.f, 0.f) & 0xffu); }

template <typename TI, typename TO, int NPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long rows, int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[NPL], b[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    g[i] = c < C ? gamma[c] : 0.0f;
    b[i] = c < C ? beta[c] : 0.0f;
  }
  const float invC = 1.0f / (float)C;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const TI* xr = x + r * C;
    float v[NPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = c < C ? ld<TI>(xr + c) : 0.0f;
      s += v[i];
    }
    const float mu = wave_sum(s) * invC;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      const float d = c < C ? v[i] - mu : 0.0f;
      q = fmaf(d, d, q);
    }
    const float rs = rsqrtf(wave_sum(q) * invC + eps);
    TO* yr = y + r * C;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      if (c < C) st<TO>(yr + c, fmaf((v[i] - mu) * rs, g[i], b[i]));
    }
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

// dx[r,c] = rstd * (g*gamma - (sum_c(g*gamma) + xhat * sum_c(g*gamma*xhat)) / C);  partial dgamma/dbeta per workgroup
template <typename TX, typename TG, int NPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const TX* __restrict__ x, const TG* __restrict__ gy,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, TX* __restrict__ dx,
                                                            float* __restrict__ ws, long rows, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gm[NPL], dg[NPL], db[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    gm[i] = c < C ? gamma[c] : 0.0f;
    dg[i] = 0.0f;
    db[i] = 0.0f;
  }
  const float invC = 1.0f / (float)C;
  for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
    const float mu = mean[r], rs = rstd[r];
    float xh[NPL], gg[NPL];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < C;
      const float xv = ok ? ld<TX>(x + r * C + c) : 0.0f;
      const float gv = ok ? ld<TG>(gy + r * C + c) : 0.0f;
      xh[i] = ok ? (xv - mu) * rs : 0.0f;
      gg[i] = gv * gm[i];
      s1 += gg[i];
      s2 = fmaf(gg[i], xh[i], s2);
      dg[i] = fmaf(gv, xh[i], dg[i]);
      db[i] += gv;
    }
    s1 = wave_sum(s1) * invC;
    s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
      const int c = lane + 64 * i;
      if (c < C) st<TX>(dx + r * C + c, rs * (gg[i] - s1 - xh[i] * s2));
    }
  }
  // combine the 4 waves' partials, one workspace row per workgroup: ws[block][2][C]
  __shared__ float red[4][2][64 * NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    red[wave][0][lane + 64 * i] = dg[i];
    red[wave][1][lane + 64 * i] = db[i];
  }
  __syncthreads();
  float* row = ws + (size_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, cc = c % C;
    row[c] = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Vectorised layout (C % 8 == 0, every MiT width): a lane owns 8 ADJACENT channels per vector (one 16-byte load for
// bf16, two for fp32) instead of 8 channels strided by 64 (eight 2-byte loads: 128 bytes per wave-load).  A row takes
// LPR = 8/16/32/64 lanes, so narrow rows share a wave (C = 64: 8 rows per wave) and reductions are shuffles inside the
// LPR-lane group.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float (&v)[8]);
template <>
__device__ __forceinline__ void ld8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8<__hip_bfloat16>(const __hip_bfloat16* p, float (&v)[8]) {
  const uint4 t = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void st8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void st8<__hip_bfloat16>(__hip_bfloat16* p, const float (&v)[8]) {
  uint4 t;
  t.x = bf16x2_bits(v[0], v[1]);
  t.y = bf16x2_bits(v[2], v[3]);
  t.z = bf16x2_bits(v[4], v[5]);
  t.w = bf16x2_bits(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = t;
}
template <>
__device__ __forceinline__ void st8<f8e4m3>(f8e4m3* p, const float (&v)[8]) {       // K5: e4m3 bytes, saturating RNE
  *reinterpret_cast<uint2*>(p) = make_uint2(quant4(v[0], v[1], v[2], v[3]), quant4(v[4], v[5], v[6], v[7]));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename TI, typename TO, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_vec_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, TO* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                long rows, int C, float eps, float oscale) {
  // oscale: the result is multiplied by it before it is stored (1 for everything but the e4m3 output of the K5 path,
  // where it is the quantisation scale of the consumer GEMM; x * 1.f is exact)
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  float g[NV][8], b[NV][8];
  bool act[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (sub + LPR * j) * 8;
    act[j] = c < C;
    ld8<float>(gamma + (act[j] ? c : 0), g[j]);      // two 16-byte loads each (c is a multiple of 8)
    ld8<float>(beta + (act[j] ? c : 0), b[j]);
  }
  const float invC = 1.0f / (float)C;
  for (long r0 = ((long)blockIdx.x * 4 + wave) * RPW; r0 < rows; r0 += (long)gridDim.x * 4 * RPW) {
    const long r = r0 + grp;
    const bool valid = r < rows;
    float v[NV][8];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (valid && act[j]) ld8<TI>(x + r * C + (sub + LPR * j) * 8, v[j]);
      else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][i] = 0.0f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[j][i];
    }
    const float mu = group_sum<LPR>(s) * invC;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = act[j] ? v[j][i] - mu : 0.0f;
        q = fmaf(d, d, q);
      }
    const float rs = rsqrtf(group_sum<LPR>(q) * invC + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (valid && act[j]) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = fmaf((v[j][i] - mu) * rs, g[j][i], b[j][i]) * oscale;
        st8<TO>(y + r * C + (sub + LPR * j) * 8, o);
      }
    if (valid && sub == 0 && mean != nullptr) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

template <typename TX, typename TG, int LPR, int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const TX* __restrict__ x, const TG* __restrict__ gy,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, TX* __restrict__ dx,
                                                                float* __restrict__ ws, long rows, int C,
                                                                const TX* __restrict__ add, const TG* __restrict__ gy2) {
  // gy2 (may be null): a second gradient of the LayerNorm's OUTPUT -- in a MiT attention block norm1(x) feeds the q projection AND
  // the spatial-reduction convolution (mix_transformer.py:142-150), two edges of the autograd graph whose gradients the engine
  // used to sum with an element-wise launch per block and backward pass -- summed in fp32 as the rows are loaded
  // add (may be null): a second gradient of x that arrives by another edge of the autograd graph -- the residual stream
  // x feeds the LayerNorm AND the residual add of the block (mix_transformer.py:203-207) -- summed here instead of in an
  // element-wise kernel of its own (2 per block and backward pass)
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane % LPR, grp = lane / LPR;
  float gm[NV][8], dg[NV][8], db[NV][8];
  bool act[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c = (sub + LPR * j) * 8;
    act[j] = c < C;
    ld8<float>(gamma + (act[j] ? c : 0), gm[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      dg[j][i] = 0.0f;
      db[j][i] = 0.0f;
    }
  }
  const float invC = 1.0f / (float)C;
  for (long r0 = ((long)blockIdx.x * 4 + wave) * RPW; r0 < rows; r0 += (long)gridDim.x * 4 * RPW) {
    const long r = r0 + grp;
    const bool valid = r < rows;
    const float mu = valid ? mean[r] : 0.0f, rs = valid ? rstd[r] : 0.0f;
    float xh[NV][8], gg[NV][8], av[NV][8];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float xv[8], gv[8];
      if (valid && act[j]) {
        ld8<TX>(x + r * C + (sub + LPR * j) * 8, xv);
        ld8<TG>(gy + r * C + (sub + LPR * j) * 8, gv);
        if (gy2 != nullptr) {
          float g2[8];
          ld8<TG>(gy2 + r * C + (sub + LPR * j) * 8, g2);
#pragma unroll
          for (int i = 0; i < 8; ++i) gv[i] += g2[i];
        }
        // the residual gradient is requested WITH the row, not after the two row reductions (a second exposed load latency
        // per row: the launch is a chain of <= 8 row iterations per wave)
        if (add != nullptr) ld8<TX>(add + r * C + (sub + LPR * j) * 8, av[j]);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { xv[i] = mu; gv[i] = 0.0f; }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xh[j][i] = (xv[i] - mu) * rs;
        gg[j][i] = gv[i] * gm[j][i];
        s1 += gg[j][i];
        s2 = fmaf(gg[j][i], xh[j][i], s2);
        dg[j][i] = fmaf(gv[i], xh[j][i], dg[j][i]);
        db[j][i] += gv[i];
      }
    }
    s1 = group_sum<LPR>(s1) * invC;
    s2 = group_sum<LPR>(s2) * invC;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (valid && act[j]) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = rs * (gg[j][i] - s1 - xh[j][i] * s2);
        if (add != nullptr) {
#pragma unroll
          for (int i = 0; i < 8; ++i) o[i] += av[j][i];
        }
        st8<TX>(dx + r * C + (sub + LPR * j) * 8, o);
      }
  }
  // the RPW row groups of a wave hold partials for the same channels: fold them, then the 4 waves through LDS
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        dg[j][i] += __shfl_xor(dg[j][i], o, 64);
        db[j][i] += __shfl_xor(db[j][i], o, 64);
      }
    }
  __shared__ float red[4][2][64 * 8 * NV];
  if (grp == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (sub + LPR * j) * 8 + i;
        red[wave][0][c] = dg[j][i];
        red[wave][1][c] = db[j][i];
      }
  }
  __syncthreads();
  float* row = ws + (size_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    const int which = c / C, cc = c % C;
    row[c] = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ ws,
                                                                   float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, int C, int nblocks,
                                                                   int accumulate) {
  // 64 columns x 4 row segments per workgroup: coalesced 256 B reads, fixed summation order
  __shared__ float part[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
  float s = 0.0f;
  if (col < 2 * C) {
#pragma unroll 8
    for (int b = seg; b < nblocks; b += 4) s += ws[(size_t)b * 2 * C + col];
  }
  part[seg][threadIdx.x & 63] = s;
  __syncthreads();
  if (seg == 0 && col < 2 * C) {
    const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    float* dst = col < C ? dgamma + col : dbeta + (col - C);
    *dst = accumulate ? *dst + t : t;       // accumulate: straight into the parameter's .grad (flat gradient buffer)
  }
}

static inline int ln_grid(long rows, int cap) { return (int)std::max<long>(1, std::min<long>(cdiv(rows, 4), cap)); }

template <typename TI, typename TO>
static int ln_fwd_dispatch(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd, long rows,
                           int C, float eps, hipStream_t st, float oscale = 1.f) {
  if (C % 8 == 0) {
#define RFN_LN_FWDV(LPR, NV)                                                                                         \
  hipLaunchKernelGGL((layernorm_fwd_vec_kernel<TI, TO, LPR, NV>),                                                    \
                     dim3((int)std::max<long>(1, std::min<long>(cdiv(rows, 4 * (64 / LPR)), 256 * 16))), dim3(256), \
                     0, st, (const TI*)x, g, b, (TO*)y, mean, rstd, rows, C, eps, oscale)
    if (C <= 64) RFN_LN_FWDV(8, 1);
    else if (C <= 128) RFN_LN_FWDV(16, 1);
    else if (C <= 256) RFN_LN_FWDV(32, 1);
    else if (C <= 512) RFN_LN_FWDV(64, 1);
    else RFN_LN_FWDV(64, 2);
#undef RFN_LN_FWDV
    return check_launch("layernorm_fwd_vec_kernel");
  }
  const int grid = ln_grid(rows, 256 * 16), npl = cdiv(C, 64);
#define RFN_LN_FWD(N)                                                                                             \
  hipLaunchKernelGGL((layernorm_fwd_kernel<TI, TO, N>), dim3(grid), dim3(256), 0, st, (const TI*)x, g, b, (TO*)y, \
                     mean, rstd, rows, C, eps)
  if (npl <= 1) RFN_LN_FWD(1);
  else if (npl <= 2) RFN_LN_FWD(2);
  else if (npl <= 4) RFN_LN_FWD(4);
  else if (npl <= 8) RFN_LN_FWD(8);
  else RFN_LN_FWD(16);
#undef RFN_LN_FWD
  return check_launch("layernorm_fwd_kernel");
}

template <typename TX, typename TG>
static int ln_bwd_dispatch(const void* x, const void* gy, const float* g, const float* mean, const float* rstd,
                           void* dx, float* dgamma, float* dbeta, float* ws, long rows, int C, int accumulate,
                           hipStream_t st, const void* add = nullptr, const void* gy2 = nullptr) {
  if ((add != nullptr || gy2 != nullptr) && C % 8 != 0) return fail(RFN_EINVAL, "rfn_layernorm_bwd_add: C %% 8 != 0");
  int grid = ln_grid(rows, kLnMaxBlocks);
  const int npl = cdiv(C, 64);
  if (C % 8 == 0) {
#define RFN_LN_BWDV(LPR, NV)                                                                                        \
  grid = (int)std::max<long>(1, std::min<long>(cdiv(rows, 4 * (64 / LPR)), kLnMaxBlocks));                         \
  hipLaunchKernelGGL((layernorm_bwd_vec_kernel<TX, TG, LPR, NV>), dim3(grid), dim3(256), 0, st, (const TX*)x,      \
                     (const TG*)gy, g, mean, rstd, (TX*)dx, ws, rows, C, (const TX*)add, (const TG*)gy2)
    if (C <= 64) { RFN_LN_BWDV(8, 1); }
    else if (C <= 128) { RFN_LN_BWDV(16, 1); }
    else if (C <= 256) { RFN_LN_BWDV(32, 1); }
    else if (C <= 512) { RFN_LN_BWDV(64, 1); }
    else { RFN_LN_BWDV(64, 2); }
#undef RFN_LN_BWDV
  } else {
#define RFN_LN_BWD(N)                                                                                              \
  hipLaunchKernelGGL((layernorm_bwd_kernel<TX, TG, N>), dim3(grid), dim3(256), 0, st, (const TX*)x, (const TG*)gy, \
                     g, mean, rstd, (TX*)dx, ws, rows, C)
  if (npl <= 1) RFN_LN_BWD(1);
  else if (npl <= 2) RFN_LN_BWD(2);
  else if (npl <= 4) RFN_LN_BWD(4);
  else if (npl <= 8) RFN_LN_BWD(8);
  else RFN_LN_BWD(16);
#undef RFN_LN_BWD
  }
  if (int rc = check_launch("layernorm_bwd_kernel")) return rc;
  hipLaunchKernelGGL(layernorm_bwd_reduce_kernel, dim3(cdiv(2L * C, 64)), dim3(256), 0, st, ws, dgamma, dbeta, C,
                     grid, accumulate);
  return check_launch("layernorm_bwd_reduce_kernel");
}

}  // namespace rfn

using namespace rfn;

extern "C" {

unsigned long rfn_layernorm_bwd_workspace_bytes(int C) {
  return (unsigned long)kLnMaxBlocks * 2ul * (unsigned long)(C > 0 ? C : 0) * sizeof(float);
}

// dtype codes: 0 = float32, 1 = bfloat16
int rfn_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                      long rows, int C, float eps, int in_dtype, int out_dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && gamma && beta && y && mean && rstd, "rfn_layernorm_fwd: null pointer");
  RFN_REQUIRE(rows > 0 && C > 0 && C <= 64 * kLnMaxPerLane, "rfn_layernorm_fwd: need 0 < C <= 1024 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == 0 && out_dtype == 0) return ln_fwd_dispatch<float, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, st);
  if (in_dtype == 0 && out_dtype == 1)
    return ln_fwd_dispatch<float, __hip_bfloat16>(x, gamma, beta, y, mean, rstd, rows, C, eps, st);
  if (in_dtype == 1 && out_dtype == 1)
    return ln_fwd_dispatch<__hip_bfloat16, __hip_bfloat16>(x, gamma, beta, y, mean, rstd, rows, C, eps, st);
  if (in_dtype == 1 && out_dtype == 0)
    return ln_fwd_dispatch<__hip_bfloat16, float>(x, gamma, beta, y, mean, rstd, rows, C, eps, st);
  return fail(RFN_EINVAL, "rfn_layernorm_fwd: dtype codes must be 0 (f32) or 1 (bf16)");
}

int rfn_layernorm_bwd_add(const void* x, const void* grad_y, const void* add, const float* gamma, const float* mean,
                          const float* rstd, void* grad_x, float* grad_gamma, float* grad_beta, void* workspace, long rows, int C,
                          int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && gamma && mean && rstd && grad_x && grad_gamma && grad_beta && workspace,
              "rfn_layernorm_bwd_add: null pointer");
  RFN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * kLnMaxPerLane, "rfn_layernorm_bwd_add: need C %% 8 == 0, C <= 1024 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (x_dtype == 0 && gy_dtype == 0)
    return ln_bwd_dispatch<float, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add);
  if (x_dtype == 0 && gy_dtype == 1)
    return ln_bwd_dispatch<float, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add);
  if (x_dtype == 1 && gy_dtype == 1)
    return ln_bwd_dispatch<__hip_bfloat16, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add);
  if (x_dtype == 1 && gy_dtype == 0)
    return ln_bwd_dispatch<__hip_bfloat16, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add);
  return fail(RFN_EINVAL, "rfn_layernorm_bwd_add: dtype codes must be 0 (f32) or 1 (bf16)");
}

// grad_y2 / add: either may be null
int rfn_layernorm_bwd_add2(const void* x, const void* grad_y, const void* grad_y2, const void* add, const float* gamma,
                           const float* mean, const float* rstd, void* grad_x, float* grad_gamma, float* grad_beta, void* workspace,
                           long rows, int C, int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && gamma && mean && rstd && grad_x && grad_gamma && grad_beta && workspace,
              "rfn_layernorm_bwd_add2: null pointer");
  RFN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * kLnMaxPerLane, "rfn_layernorm_bwd_add2: need C %% 8 == 0, C <= 1024 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (x_dtype == 0 && gy_dtype == 0)
    return ln_bwd_dispatch<float, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add, grad_y2);
  if (x_dtype == 0 && gy_dtype == 1)
    return ln_bwd_dispatch<float, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add, grad_y2);
  if (x_dtype == 1 && gy_dtype == 1)
    return ln_bwd_dispatch<__hip_bfloat16, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add, grad_y2);
  if (x_dtype == 1 && gy_dtype == 0)
    return ln_bwd_dispatch<__hip_bfloat16, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st, add, grad_y2);
  return fail(RFN_EINVAL, "rfn_layernorm_bwd_add2: dtype codes must be 0 (f32) or 1 (bf16)");
}

int rfn_layernorm_fwd_f8(const void* x_bf16, const float* gamma, const float* beta, void* y8, long rows, int C, float eps,
                         float out_q, rfn_stream_t stream) {
  RFN_REQUIRE(x_bf16 && gamma && beta && y8, "rfn_layernorm_fwd_f8: null pointer");
  RFN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C <= 64 * kLnMaxPerLane, "rfn_layernorm_fwd_f8: need C %% 8 == 0, C <= 1024 (got %d)", C);
  return ln_fwd_dispatch<__hip_bfloat16, f8e4m3>(x_bf16, gamma, beta, y8, nullptr, nullptr, rows, C, eps, (hipStream_t)stream,
                                                 out_q);
}

int rfn_layernorm_bwd(const void* x, const void* grad_y, const float* gamma, const float* mean, const float* rstd,
                      void* grad_x, float* grad_gamma, float* grad_beta, void* workspace, long rows, int C,
                      int x_dtype, int gy_dtype, int accumulate, rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && gamma && mean && rstd && grad_x && grad_gamma && grad_beta && workspace,
              "rfn_layernorm_bwd: null pointer");
  RFN_REQUIRE(rows > 0 && C > 0 && C <= 64 * kLnMaxPerLane, "rfn_layernorm_bwd: need 0 < C <= 1024 (got %d)", C);
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  if (x_dtype == 0 && gy_dtype == 0)
    return ln_bwd_dispatch<float, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st);
  if (x_dtype == 0 && gy_dtype == 1)
    return ln_bwd_dispatch<float, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st);
  if (x_dtype == 1 && gy_dtype == 1)
    return ln_bwd_dispatch<__hip_bfloat16, __hip_bfloat16>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st);
  if (x_dtype == 1 && gy_dtype == 0)
    return ln_bwd_dispatch<__hip_bfloat16, float>(x, grad_y, gamma, mean, rstd, grad_x, grad_gamma, grad_beta, ws, rows, C, accumulate, st);
  return fail(RFN_EINVAL, "rfn_layernorm_bwd: dtype codes must be 0 (f32) or 1 (bf16)");
}

}  // extern "C"
