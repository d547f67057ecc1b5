// refign_amd/csrc/gcorr.hip -- global feature correlation layer for gfx950.
//
// Reference: GlobalFeatureCorrelationLayer.forward (models/modules.py:294-308) = torch.bmm of the two flattened
// feature maps ('3D', H-first branch, modules.py:361-375) + mutual_matching (modules.py:310-333, eps 1e-5)
// + ReLU + L2-normalisation over the source axis (eps 1e-12).
//   S[b, s, t] = sum_c src[b,c,s] * trg[b,c,t],  s = hs*Ws+ws,  t = ht*Wt+wt,   out: (B, Ns, Ht, Wt)
// The level-4 problem is 256 x 256 x 512 per image (67 MFLOP): latency-, not throughput-bound.  Kernel 1 is a
// 64x64-tiled fp32 FMA GEMM (both operands are K-major with the M/N index contiguous, so global loads are 16-byte
// coalesced with no transposes); kernel 2 does the row/column maxima, the mutual-matching product, ReLU and the
// column L2 norm in place, one workgroup per 16 target positions (the 256 KB score matrix stays L2-resident).
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

// Mutual matching + ReLU + column L2 norm, in place on S (Ns x Nt row-major per image).  One workgroup per block of kCB = 16
// target positions (columns) of one image: 16 x B workgroups instead of round 1's one per image, whose threads each walked a
// whole column with dependent loads (208 us for 2 x 256 x 256: VERDICT r4).  A workgroup
//   1. (cyclic) reads the row maxima (modules.py:321 corr4d_A_max) gcorr_rowmax_kernel left in the workspace -- a launch of its
//      own, a wave per row: the in-place update below must not start while any workgroup still needs the raw scores of
//      another one's columns;
//   2. loads its 16 columns (thread = (row lane r, column c): rows r, r + 16, ...; 16 lanes = one 64-byte segment of a row)
//      into registers, reduces the column maximum over the 16 row lanes through LDS (modules.py:320 corr4d_B_max),
//   3. applies v * (v / (rowmax + eps)) * (v / (colmax + eps)) (modules.py:324-331), ReLU, sums the squares per column the same
//      way, divides and stores.
// NV = rows per thread (Ns <= 16 NV).
constexpr int kCB = 16;

// rowmax[img * Ns + s] = max_t S[img][s][t]: one wave per row (4 rows per workgroup)
__global__ __launch_bounds__(256) void gcorr_rowmax_kernel(const float* __restrict__ S, float* __restrict__ rowmax, long rows,
                                                           int Nt) {
  const int lane = threadIdx.x & 63;
  const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= rows) return;
  const float* row = S + (size_t)s * Nt;
  float m = -INFINITY;
  if ((Nt & 3) == 0) {
    for (int t = 4 * lane; t < Nt; t += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + t);
      m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
  } else {
    for (int t = lane; t < Nt; t += 64) m = fmaxf(m, row[t]);
  }
  m = wave_max(m);
  if (lane == 0) rowmax[s] = m;
}

template <int NV>
__global__ __launch_bounds__(256) void gcorr_post_kernel(float* __restrict__ out, const float* __restrict__ rowmax_all, int Ns,
                                                         int Nt, int cyclic) {
  __shared__ float rowmax[1024];
  __shared__ float red[16][kCB + 1];
  const int nblk = (Nt + kCB - 1) / kCB;
  const int img = blockIdx.x / nblk, t0 = (blockIdx.x % nblk) * kCB;
  float* S = out + (size_t)img * Ns * Nt;
  const int tid = threadIdx.x;
  if (cyclic)
    for (int s = tid; s < Ns; s += 256) rowmax[s] = rowmax_all[(size_t)img * Ns + s];
  const int r = tid >> 4, c = tid & 15, t = t0 + c;
  const bool live = t < Nt;
  float v[NV];
  float cm = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int s = r + 16 * i;
    v[i] = (live && s < Ns) ? S[(size_t)s * Nt + t] : -INFINITY;
    cm = fmaxf(cm, v[i]);
  }
  red[r][c] = cm;
  __syncthreads();                                     // (also: rowmax complete)
  const float eps = 1e-5f;
  float cden = 1.0f;
  if (cyclic) {
    float m = red[0][c];
#pragma unroll
    for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k][c]);
    cden = m + eps;
  }
  __syncthreads();
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int s = r + 16 * i;
    float x = (live && s < Ns) ? v[i] : 0.0f;
    if (cyclic && live && s < Ns) {
      const float ca = x / (rowmax[s] + eps);          // corr4d_A (modules.py:325)
      const float cb = x / cden;                       // corr4d_B (modules.py:324)
      x = x * (ca * cb);                               // modules.py:331
    }
    x = fmaxf(x, 0.0f);
    ss = fmaf(x, x, ss);
    v[i] = x;
  }
  red[r][c] = ss;
  __syncthreads();
  float tot = 0.0f;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += red[k][c];
  const float d = fmaxf(sqrtf(tot), 1e-12f);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int s = r + 16 * i;
    if (live && s < Ns) S[(size_t)s * Nt + t] = v[i] / d;
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" int rfn_global_corr_layer_f32(const float* feature_source, const float* feature_target, float* out,
                                         float* workspace, int B, int C, int Hs, int Ws, int Ht, int Wt,
                                         int cyclic_consistency, rfn_stream_t stream) {
  RFN_REQUIRE(feature_source && feature_target && out, "rfn_global_corr_layer_f32: null pointer");
  RFN_REQUIRE(workspace || !cyclic_consistency, "rfn_global_corr_layer_f32: mutual matching needs B * Hs * Ws floats of workspace");
  RFN_REQUIRE(B > 0 && C > 0 && Hs > 0 && Ws > 0 && Ht > 0 && Wt > 0 && B <= 65535,
              "rfn_global_corr_layer_f32: non-positive size");
  const int Ns = Hs * Ws, Nt = Ht * Wt;
  RFN_REQUIRE(Ns <= 1024 && Nt <= 1024, "rfn_global_corr_layer_f32: at most 1024 positions per map (got %d, %d)",
              Ns, Nt);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gcorr_gemm_kernel, dim3(cdiv(Nt, kGT), cdiv(Ns, kGT), B), dim3(256), 0, st, feature_source,
                     feature_target, out, C, Ns, Nt);
  if (int rc = check_launch("gcorr_gemm_kernel")) return rc;
  if (cyclic_consistency) {
    const long rows = (long)B * Ns;
    hipLaunchKernelGGL(gcorr_rowmax_kernel, dim3((unsigned)cdiv(rows, 4L)), dim3(256), 0, st, (const float*)out, workspace, rows,
                       Nt);
    if (int rc = check_launch("gcorr_rowmax_kernel")) return rc;
  }
  const unsigned blocks = (unsigned)(B * cdiv(Nt, kCB));
  if (Ns <= 256)
    hipLaunchKernelGGL(gcorr_post_kernel<16>, dim3(blocks), dim3(256), 0, st, out, (const float*)workspace, Ns, Nt,
                       cyclic_consistency);
  else
    hipLaunchKernelGGL(gcorr_post_kernel<64>, dim3(blocks), dim3(256), 0, st, out, (const float*)workspace, Ns, Nt,
                       cyclic_consistency);
  return check_launch("gcorr_post_kernel");
}
