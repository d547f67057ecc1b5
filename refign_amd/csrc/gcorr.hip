// refign_amd/csrc/gcorr.hip -- global feature correlation layer for gfx950.
//
// Reference: GlobalFeatureCorrelationLayer.forward (models/modules.py:294-308) = torch.bmm of the two flattened
// feature maps ('3D', H-first branch, modules.py:361-375) + mutual_matching (modules.py:310-333, eps 1e-5)
// + ReLU + L2-normalisation over the source axis (eps 1e-12).
//   S[b, s, t] = sum_c src[b,c,s] * trg[b,c,t],  s = hs*Ws+ws,  t = ht*Wt+wt,   out: (B, Ns, Ht, Wt)
// The level-4 problem is 256 x 256 x 512 per image (67 MFLOP): latency-, not throughput-bound.  Kernel 1 is a
// 64x64-tiled fp32 FMA GEMM (both operands are K-major with the M/N index contiguous, so global loads are 16-byte
// coalesced with no transposes); kernel 2 does the row/column maxima, the mutual-matching product, ReLU and the
// column L2 norm in place, one workgroup per image (the 256 KB score matrix stays L2-resident).
#include "common.h"

namespace rfn {

constexpr int kGT = 64;   // GEMM tile (M and N)
constexpr int kGK = 16;   // K chunk

__global__ __launch_bounds__(256) void gcorr_gemm_kernel(const float* __restrict__ src, const float* __restrict__ trg,
                                                         float* __restrict__ out, int C, int Ns, int Nt) {
  __shared__ __attribute__((aligned(16))) float sA[kGK][kGT];
  __shared__ __attribute__((aligned(16))) float sB[kGK][kGT];
  const int n = blockIdx.z;
  const int m0 = blockIdx.y * kGT, n0 = blockIdx.x * kGT;
  const float* A = src + (size_t)n * C * Ns;
  const float* Bm = trg + (size_t)n * C * Nt;
  const int tid = threadIdx.x;
  const int tm = (tid >> 4) * 4, tn = (tid & 15) * 4;   // 16 x 16 threads, 4 x 4 outputs each
  const int lk = tid >> 4, lx = (tid & 15) * 4;         // staging: row k, 4 consecutive columns
  float acc[4][4] = {};
  for (int k0 = 0; k0 < C; k0 += kGK) {
    __syncthreads();
    {
      const int k = k0 + lk;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
      if (k < C) {
        const float* pa = A + (size_t)k * Ns + m0 + lx;
        const float* pb = Bm + (size_t)k * Nt + n0 + lx;
        if ((Ns & 3) == 0 && m0 + lx + 3 < Ns) va = *reinterpret_cast<const float4*>(pa);
        else {
          if (m0 + lx + 0 < Ns) va.x = pa[0];
          if (m0 + lx + 1 < Ns) va.y = pa[1];
          if (m0 + lx + 2 < Ns) va.z = pa[2];
          if (m0 + lx + 3 < Ns) va.w = pa[3];
        }
        if ((Nt & 3) == 0 && n0 + lx + 3 < Nt) vb = *reinterpret_cast<const float4*>(pb);
        else {
          if (n0 + lx + 0 < Nt) vb.x = pb[0];
          if (n0 + lx + 1 < Nt) vb.y = pb[1];
          if (n0 + lx + 2 < Nt) vb.z = pb[2];
          if (n0 + lx + 3 < Nt) vb.w = pb[3];
        }
      }
      *reinterpret_cast<float4*>(&sA[lk][lx]) = va;
      *reinterpret_cast<float4*>(&sB[lk][lx]) = vb;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kGK; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(&sA[k][tm]);
      const float4 b = *reinterpret_cast<const float4*>(&sB[k][tn]);
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
  }
  float* O = out + (size_t)n * Ns * Nt;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + tm + i;
    if (m >= Ns) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + tn + j < Nt) O[(size_t)m * Nt + n0 + tn + j] = acc[i][j];
  }
}

// One workgroup (256 threads) per image; S is (Ns x Nt) row-major in `out`, processed in place.
__global__ __launch_bounds__(256) void gcorr_post_kernel(float* __restrict__ out, int Ns, int Nt, int cyclic) {
  __shared__ float rowmax[1024];   // max over t, per source position s   (corr4d_A_max, modules.py:321)
  __shared__ float colmax[1024];   // max over s, per target position t   (corr4d_B_max, modules.py:320)
  float* S = out + (size_t)blockIdx.x * Ns * Nt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (cyclic) {
    for (int t = tid; t < Nt; t += 256) {
      float m = -INFINITY;
      for (int s = 0; s < Ns; ++s) m = fmaxf(m, S[(size_t)s * Nt + t]);
      colmax[t] = m;
    }
    for (int s = wave; s < Ns; s += 4) {
      float m = -INFINITY;
      for (int t = lane; t < Nt; t += 64) m = fmaxf(m, S[(size_t)s * Nt + t]);
      m = wave_max(m);
      if (lane == 0) rowmax[s] = m;
    }
    __syncthreads();
  }
  const float eps = 1e-5f;
  for (int t = tid; t < Nt; t += 256) {
    float ss = 0.0f;
    const float cden = cyclic ? colmax[t] + eps : 1.0f;
    for (int s = 0; s < Ns; ++s) {
      float v = S[(size_t)s * Nt + t];
      if (cyclic) {
        const float ca = v / (rowmax[s] + eps);   // corr4d_A (modules.py:325)
        const float cb = v / cden;                // corr4d_B (modules.py:324)
        v = v * (ca * cb);                        // modules.py:331
      }
      v = fmaxf(v, 0.0f);
      ss = fmaf(v, v, ss);
      S[(size_t)s * Nt + t] = v;
    }
    const float d = fmaxf(sqrtf(ss), 1e-12f);
    for (int s = 0; s < Ns; ++s) S[(size_t)s * Nt + t] /= d;
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" int rfn_global_corr_layer_f32(const float* feature_source, const float* feature_target, float* out, int B,
                                         int C, int Hs, int Ws, int Ht, int Wt, int cyclic_consistency,
                                         rfn_stream_t stream) {
  RFN_REQUIRE(feature_source && feature_target && out, "rfn_global_corr_layer_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0 && B <= 65535,
              "rfn_global_corr_layer_f32: non-positive size");
  const int Ns = Hs * Ws, Nt = Ht * Wt;
  RFN_REQUIRE(Ns <= 1024 && Nt <= 1024, "rfn_global_corr_layer_f32: at most 1024 positions per map (got %d, %d)",
              Ns, Nt);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gcorr_gemm_kernel, dim3(cdiv(Nt, kGT), cdiv(Ns, kGT), B), dim3(256), 0, st, feature_source,
                     feature_target, out, C, Ns, Nt);
  if (int rc = check_launch("gcorr_gemm_kernel")) return rc;
  hipLaunchKernelGGL(gcorr_post_kernel, dim3(B), dim3(256), 0, st, out, Ns, Nt, cyclic_consistency);
  return check_launch("gcorr_post_kernel");
}
