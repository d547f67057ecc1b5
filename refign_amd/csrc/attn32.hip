// refign_amd/csrc/attn32.hip -- MiT's efficient self-attention (mix_transformer.py:137-164) for FLOAT32 tensors, the fp32
// parity mode's attention core: softmax(scale Q K^T) V and its backward, head dimension D = 64 (MiT-B1..B5) or 32 (MiT-B0), as
// ONE launch per pass with every
// product on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: true fp32 products and sums -- no operand splitting, no score
// matrix in memory).  Until round 6 the parity mode looped over (batch, head) with explicit split-bf16 GEMMs and a torch
// softmax: ~10^5 launches per teacher pass, which no hipGraph could hold (refign_amd/split32.py).
//
// Same formulation as the 16-bit kernels of attn.hip -- everything TRANSPOSED so that a lane owns one query (forward, dQ)
// or one key (dK / dV), and a probability block leaves one MFMA as a C/D register block and enters the next as its B
// operand without moving -- in the register maps of the 32x32x2 instruction (wave64, lane l, j = l & 31, g = l >> 5):
//   A operand : one value  A[i = j][k = g]
//   B operand : one value  B[k = g][j]
//   C/D       : 16 floats  D[i = crow(r, g)][j],  crow(r, g) = (r & 3) + 8 (r >> 2) + 4 g
// The instruction is a dot product over its two k-slots and an accumulation over calls, so WHICH reduction index sits in
// call s, slot g is free as long as A and B agree:
//   over d (S^T = K Q^T, dP^T = V dO^T, ...):   call s, slot g  <->  d = (D/2) g + s   (a lane's D/2 values are contiguous)
//   over rows of a C/D block (O^T += V^T P^T):  call r, slot g  <->  row crow(r, g)    (the block's own register order)
// Streamed operands (K / V blocks in the forward and dQ kernels, Q / dO blocks in the dK / dV kernel) sit in LDS as 32 rows
// of D floats with a pitch of D + 4: the 16 lanes of a ds_read_b128 service group then cover all 64 banks.
#include <cmath>

#include "common.h"
#include "mfma.h"

namespace rfn {

constexpr float kLog2e32 = 1.4426950408889634f;

__device__ __forceinline__ f32x16 mma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int crow(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

template <int D> struct Rows {
  f32x4 v[D / 32];
};
// 32 rows x D floats of a (rows, row stride) image -> D / 32 float4 per thread of a 256-thread workgroup; rows past the end: 0
template <int D> __device__ __forceinline__ Rows<D> rows_load(const float* base, long stride, int row0, int nrows, int t) {
  const int row = row0 + (t >> 3);
  Rows<D> r;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) r.v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (row < nrows) {
    const float* p = base + (long)row * stride + (t & 7) * 4;
#pragma unroll
    for (int i = 0; i < D / 32; ++i) r.v[i] = *(const f32x4*)(p + 32 * i);
  }
  return r;
}
template <int D> __device__ __forceinline__ void rows_store(float* lds, const Rows<D>& r, int t) {
  float* p = lds + (t >> 3) * (D + 4) + (t & 7) * 4;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) *(f32x4*)(p + 32 * i) = r.v[i];
}
// a lane's D / 2 contiguous values of one row (d = (D/2) g ...) from global memory, times `mul`; missing rows: 0
template <int D> __device__ __forceinline__ void row_frag(const float* rowptr, bool ok, int g, float mul, float (&f)[D / 2]) {
#pragma unroll
  for (int q4 = 0; q4 < D / 8; ++q4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *(const f32x4*)(rowptr + (D / 2) * g + 4 * q4);
#pragma unroll
    for (int e = 0; e < 4; ++e) f[4 * q4 + e] = v[e] * mul;
  }
}
// D (+)= X Y^T over d: X's rows from an LDS tile (A operand), Y's rows in registers (B operand, row_frag order)
template <int D> __device__ __forceinline__ f32x16 dot_d(const float* tile, int j, int g, const float (&y)[D / 2], f32x16 acc) {
  const float* p = tile + j * (D + 4) + (D / 2) * g;
#pragma unroll
  for (int q4 = 0; q4 < D / 8; ++q4) {
    const f32x4 a = *(const f32x4*)(p + 4 * q4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = mma32(a[e], y[4 * q4 + e], acc);
  }
  return acc;
}
// acc[d-block][i = d][j] += sum over the 32 tile rows of tile[row][32 db + i] * blk[row][j], blk = a C/D register block
template <int D> __device__ __forceinline__ void acc_rows(const float* tile, int j, int g, const f32x16& blk, f32x16 (&acc)[D / 32]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* p = tile + crow(r, g) * (D + 4) + j;
#pragma unroll
    for (int db = 0; db < D / 32; ++db) acc[db] = mma32(p[32 * db], blk[r], acc[db]);
  }
}
// a transposed accumulator pair (D[i = d][j = row]) -> row-major rows of D floats (store, or atomic add for partial sums)
template <bool ATOMIC, int DB>
__device__ __forceinline__ void store_rows(const f32x16 (&acc)[DB], float mul, float* rowptr, int g) {
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float* p = rowptr + 32 * db + 8 * c + 4 * g;
      if (ATOMIC) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(p + e, acc[db][4 * c + e] * mul);
      } else {
        *(f32x4*)p = f32x4{acc[db][4 * c] * mul, acc[db][4 * c + 1] * mul, acc[db][4 * c + 2] * mul, acc[db][4 * c + 3] * mul};
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// forward: grid (ceil(Nq / 128), B * heads), 4 waves x 32 queries; K / V stream through LDS in blocks of 32 keys
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_fwd_kernel(const float* __restrict__ Q, long qsb, long qsr,
                                                         const float* __restrict__ KV, long ksb, long ksr,
                                                         float* __restrict__ O, long osb, long osr, float* __restrict__ lse2,
                                                         int heads, int Nq, int Nkv, int nqpad, float scale) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][32 * (D + 4)];
  const int t = threadIdx.x, l = t & 63, j = l & 31, g = l >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int q = blockIdx.x * 128 + (t >> 6) * 32 + j;
  const bool ok = q < Nq;
  const float* kbase = KV + (long)b * ksb + hd * D;
  const float* vbase = kbase + heads * D;
  float qreg[D / 2];                                    // base-2 scores: scale * log2(e) rides on Q
  row_frag<D>(Q + (long)b * qsb + (long)(ok ? q : 0) * qsr + hd * D, ok, g, scale * kLog2e32, qreg);
  f32x16 o[D / 32];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int db = 0; db < D / 32; ++db) o[db][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const int nkb = (Nkv + 31) >> 5;
  Rows<D> kt = rows_load<D>(kbase, ksr, 0, Nkv, t), vt = rows_load<D>(vbase, ksr, 0, Nkv, t);
  rows_store<D>(lds[0][0], kt, t);
  rows_store<D>(lds[0][1], vt, t);
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) {
      kt = rows_load<D>(kbase, ksr, (kb + 1) * 32, Nkv, t);
      vt = rows_load<D>(vbase, ksr, (kb + 1) * 32, Nkv, t);
    }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    s = dot_d<D>(lds[cur][0], j, g, qreg, s);             // S^T[key][query]
    if (kb == nkb - 1 && (Nkv & 31)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kb * 32 + crow(r, g) >= Nkv) s[r] = -INFINITY;
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    const float mn = fmaxf(m, half_max(mx));
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - mn);
      ps += s[r];
    }
    lsum = lsum * alpha + ps;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int db = 0; db < D / 32; ++db) o[db][r] *= alpha;
    acc_rows<D>(lds[cur][1], j, g, s, o);                 // O^T[d][query] += V^T P^T
    if (kb + 1 < nkb) {
      rows_store<D>(lds[cur ^ 1][0], kt, t);
      rows_store<D>(lds[cur ^ 1][1], vt, t);
    }
    __syncthreads();
  }
  lsum = half_sum(lsum);
  if (ok) {
    store_rows<false, D / 32>(o, 1.f / lsum, O + (long)b * osb + (long)q * osr + hd * D, g);
    if (g == 0) lse2[(long)bh * nqpad + q] = m + log2f(lsum);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dQ (+ delta = rowsum(dO o O)): same walk as the forward, a lane owns a query
//   P^T = exp2(S^T - lse2),  dS^T = P^T o (dP^T - delta) * scale,  dQ^T[d][query] += K^T dS^T
// ---------------------------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn32_bwd_dq_kernel(const float* __restrict__ Q, long qsb, long qsr,
                                                            const float* __restrict__ KV, long ksb, long ksr,
                                                            const float* __restrict__ dO, const float* __restrict__ O,
                                                            long osb, long osr, const float* __restrict__ lse2,
                                                            float* __restrict__ delta, float* __restrict__ dQ, long dsb,
                                                            long dsr, int heads, int Nq, int Nkv, int nqpad, float scale) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][32 * (D + 4)];
  const int t = threadIdx.x, l = t & 63, j = l & 31, g = l >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int q = blockIdx.x * 128 + (t >> 6) * 32 + j;
  const bool ok = q < Nq;
  const float* kbase = KV + (long)b * ksb + hd * D;
  const float* vbase = kbase + heads * D;
  float qreg[D / 2], gor[D / 2];
  row_frag<D>(Q + (long)b * qsb + (long)(ok ? q : 0) * qsr + hd * D, ok, g, scale * kLog2e32, qreg);
  row_frag<D>(dO + (long)b * osb + (long)(ok ? q : 0) * osr + hd * D, ok, g, 1.f, gor);
  float dl = 0.f;
  {
    float orow[D / 2];
    row_frag<D>(O + (long)b * osb + (long)(ok ? q : 0) * osr + hd * D, ok, g, 1.f, orow);
#pragma unroll
    for (int s = 0; s < D / 2; ++s) dl = fmaf(gor[s], orow[s], dl);
  }
  dl = half_sum(dl);
  const float ls = ok ? lse2[(long)bh * nqpad + q] : INFINITY;      // (missing queries: P = 0)
  if (ok && g == 0) delta[(long)bh * nqpad + q] = dl;
  f32x16 dq[D / 32];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int db = 0; db < D / 32; ++db) dq[db][r] = 0.f;
  const int nkb = (Nkv + 31) >> 5;
  Rows<D> kt = rows_load<D>(kbase, ksr, 0, Nkv, t), vt = rows_load<D>(vbase, ksr, 0, Nkv, t);
  rows_store<D>(lds[0][0], kt, t);
  rows_store<D>(lds[0][1], vt, t);
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) {
      kt = rows_load<D>(kbase, ksr, (kb + 1) * 32, Nkv, t);
      vt = rows_load<D>(vbase, ksr, (kb + 1) * 32, Nkv, t);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
    s = dot_d<D>(lds[cur][0], j, g, qreg, s);             // S^T
    dp = dot_d<D>(lds[cur][1], j, g, gor, dp);            // dP^T = V dO^T
    const bool tail = kb == nkb - 1 && (Nkv & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = __builtin_amdgcn_exp2f(s[r] - ls);
      if (tail && kb * 32 + crow(r, g) >= Nkv) p = 0.f;
      s[r] = p * (dp[r] - dl) * scale;                 // dS^T
    }
    acc_rows<D>(lds[cur][0], j, g, s, dq);                // dQ^T += K^T dS^T
    if (kb + 1 < nkb) {
      rows_store<D>(lds[cur ^ 1][0], kt, t);
      rows_store<D>(lds[cur ^ 1][1], vt, t);
    }
    __syncthreads();
  }
  if (ok) store_rows<false, D / 32>(dq, 1.f, dQ + (long)b * dsb + (long)q * dsr + hd * D, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward, dK / dV: a lane owns a key; grid (ceil(Nkv / 128), B * heads, query chunks); Q / dO stream through LDS in
// blocks of 32 queries; the chunks of one key tile add their partial sums with fp32 atomics (one chunk: plain stores)
//   S = Q K^T,  dP = dO V^T,  P = exp2(c S - lse2),  dS = P o (dP - delta) * scale
//   dV^T[d][key] += dO^T P,   dK^T[d][key] += Q^T dS
// ---------------------------------------------------------------------------------------------------------------------
template <int D, bool ATOMIC>
__global__ __launch_bounds__(256) void attn32_bwd_dkv_kernel(const float* __restrict__ Q, long qsb, long qsr,
                                                             const float* __restrict__ KV, long ksb, long ksr,
                                                             const float* __restrict__ dO, long osb, long osr,
                                                             const float* __restrict__ lse2, const float* __restrict__ delta,
                                                             float* __restrict__ dKV, long gsb, long gsr, int heads, int Nq,
                                                             int Nkv, int nqpad, int blocks_per_chunk, float scale) {
  __shared__ __attribute__((aligned(16))) float lds[2][2][32 * (D + 4)];
  __shared__ float stat[2][2][32];                     // [buffer][lse2 / delta][query of the block]
  const int t = threadIdx.x, l = t & 63, j = l & 31, g = l >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int key = blockIdx.x * 128 + (t >> 6) * 32 + j;
  const bool ok = key < Nkv;
  const float* qbase = Q + (long)b * qsb + hd * D;
  const float* gbase = dO + (long)b * osb + hd * D;
  const float* krow = KV + (long)b * ksb + (long)(ok ? key : 0) * ksr + hd * D;
  float kreg[D / 2], vreg[D / 2];
  row_frag<D>(krow, ok, g, 1.f, kreg);
  row_frag<D>(krow + heads * D, ok, g, 1.f, vreg);
  f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int db = 0; db < D / 32; ++db) dk[db][r] = dv[db][r] = 0.f;
  const int nqb = (Nq + 31) >> 5;
  const int qb0 = blockIdx.z * blocks_per_chunk, qb1 = min(nqb, qb0 + blocks_per_chunk);
  const float c2 = scale * kLog2e32;
  auto stats_load = [&](int qb, float& a, float& d) {
    const int qq = qb * 32 + (t & 31);
    a = INFINITY;
    d = 0.f;
    if (t < 32 && qq < Nq) {
      a = lse2[(long)bh * nqpad + qq];
      d = delta[(long)bh * nqpad + qq];
    }
  };
  if (qb0 < qb1) {
    Rows<D> qt = rows_load<D>(qbase, qsr, qb0 * 32, Nq, t), gt = rows_load<D>(gbase, osr, qb0 * 32, Nq, t);
    float sa, sd;
    stats_load(qb0, sa, sd);
    rows_store<D>(lds[0][0], qt, t);
    rows_store<D>(lds[0][1], gt, t);
    if (t < 32) {
      stat[0][0][t] = sa;
      stat[0][1][t] = sd;
    }
    __syncthreads();
    for (int qb = qb0; qb < qb1; ++qb) {
      const int cur = (qb - qb0) & 1;
      if (qb + 1 < qb1) {
        qt = rows_load<D>(qbase, qsr, (qb + 1) * 32, Nq, t);
        gt = rows_load<D>(gbase, osr, (qb + 1) * 32, Nq, t);
        stats_load(qb + 1, sa, sd);
      }
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
      s = dot_d<D>(lds[cur][0], j, g, kreg, s);           // S[query][key]
      dp = dot_d<D>(lds[cur][1], j, g, vreg, dp);         // dP[query][key]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = crow(r, g);
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c2, -stat[cur][0][qi]));
        dp[r] = p * (dp[r] - stat[cur][1][qi]) * scale;   // dS
        s[r] = p;
      }
      acc_rows<D>(lds[cur][1], j, g, s, dv);              // dV^T += dO^T P
      acc_rows<D>(lds[cur][0], j, g, dp, dk);             // dK^T += Q^T dS
      if (qb + 1 < qb1) {
        rows_store<D>(lds[cur ^ 1][0], qt, t);
        rows_store<D>(lds[cur ^ 1][1], gt, t);
        if (t < 32) {
          stat[cur ^ 1][0][t] = sa;
          stat[cur ^ 1][1][t] = sd;
        }
      }
      __syncthreads();
    }
  }
  if (ok) {
    float* grow = dKV + (long)b * gsb + (long)key * gsr + hd * D;
    store_rows<ATOMIC, D / 32>(dk, 1.f, grow, g);
    store_rows<ATOMIC, D / 32>(dv, 1.f, grow + heads * D, g);
  }
}

template <int D>
static int attn32_bwd_launch(const float* Q, long qsb, long qsr, const float* KV, long ksb, long ksr, const float* dO, const float* O,
                             long osb, long osr, const float* lse2, float* delta, float* dQ, long dsb, long dsr, float* dKV,
                             long gsb, long gsr, int B, int heads, int Nq, int Nkv, int nqpad, int query_chunks, float scale,
                             hipStream_t st) {
  hipLaunchKernelGGL(attn32_bwd_dq_kernel<D>, dim3(cdiv(Nq, 128), B * heads), dim3(256), 0, st, Q, qsb, qsr, KV, ksb, ksr, dO, O,
                     osb, osr, lse2, delta, dQ, dsb, dsr, heads, Nq, Nkv, nqpad, scale);
  int rc = check_launch("attn32_bwd_dq");
  if (rc != RFN_OK) return rc;
  const int nqb = cdiv(Nq, 32);
  const int chunks = query_chunks > nqb ? nqb : query_chunks;
  const int per = cdiv(nqb, chunks);
  dim3 grid(cdiv(Nkv, 128), B * heads, cdiv(nqb, per));
  // (more than one chunk: dKV must be zero on entry -- the chunks add their partial sums)
  if (grid.z > 1)
    hipLaunchKernelGGL((attn32_bwd_dkv_kernel<D, true>), grid, dim3(256), 0, st, Q, qsb, qsr, KV, ksb, ksr, dO, osb, osr, lse2,
                       delta, dKV, gsb, gsr, heads, Nq, Nkv, nqpad, per, scale);
  else
    hipLaunchKernelGGL((attn32_bwd_dkv_kernel<D, false>), grid, dim3(256), 0, st, Q, qsb, qsr, KV, ksb, ksr, dO, osb, osr, lse2,
                       delta, dKV, gsb, gsr, heads, Nq, Nkv, nqpad, per, scale);
  return check_launch("attn32_bwd_dkv");
}

}  // namespace rfn

extern "C" {

using namespace rfn;

#define ATTN32_STRIDES_OK(...)                                                                                    \
  do {                                                                                                            \
    const long st_[] = {__VA_ARGS__};                                                                             \
    for (long v_ : st_) RFN_REQUIRE(v_ % 4 == 0, "attn32: strides must be multiples of 4 floats (got %ld)", v_);  \
  } while (0)

int rfn_attn32_fwd(const float* Q, long q_batch_stride, long q_row_stride, const float* KV, long kv_batch_stride,
                   long kv_row_stride, float* O, long o_batch_stride, long o_row_stride, float* lse2, int B, int heads,
                   int head_dim, int Nq, int Nkv, int nqpad, float scale, rfn_stream_t stream) {
  RFN_REQUIRE(Q && KV && O && lse2, "attn32_fwd: null pointer");
  RFN_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nkv > 0 && nqpad >= Nq && (head_dim == 32 || head_dim == 64),
              "attn32_fwd: B=%d heads=%d head_dim=%d Nq=%d Nkv=%d nqpad=%d", B, heads, head_dim, Nq, Nkv, nqpad);
  ATTN32_STRIDES_OK(q_row_stride, q_batch_stride, kv_row_stride, kv_batch_stride, o_row_stride, o_batch_stride);
  dim3 grid(cdiv(Nq, 128), B * heads);
  if (head_dim == 64)
    hipLaunchKernelGGL(attn32_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, Q, q_batch_stride, q_row_stride, KV,
                       kv_batch_stride, kv_row_stride, O, o_batch_stride, o_row_stride, lse2, heads, Nq, Nkv, nqpad, scale);
  else
    hipLaunchKernelGGL(attn32_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, Q, q_batch_stride, q_row_stride, KV,
                       kv_batch_stride, kv_row_stride, O, o_batch_stride, o_row_stride, lse2, heads, Nq, Nkv, nqpad, scale);
  return check_launch("attn32_fwd");
}

int rfn_attn32_bwd(const float* Q, long q_batch_stride, long q_row_stride, const float* KV, long kv_batch_stride,
                   long kv_row_stride, const float* dO, const float* O, long o_batch_stride, long o_row_stride,
                   const float* lse2, float* delta, float* dQ, long dq_batch_stride, long dq_row_stride, float* dKV,
                   long dkv_batch_stride, long dkv_row_stride, int B, int heads, int head_dim, int Nq, int Nkv, int nqpad,
                   int query_chunks, float scale, rfn_stream_t stream) {
  RFN_REQUIRE(Q && KV && dO && O && lse2 && delta && dQ && dKV, "attn32_bwd: null pointer");
  RFN_REQUIRE(B > 0 && heads > 0 && Nq > 0 && Nkv > 0 && nqpad >= Nq && query_chunks >= 1 && (head_dim == 32 || head_dim == 64),
              "attn32_bwd: B=%d heads=%d head_dim=%d Nq=%d Nkv=%d nqpad=%d chunks=%d", B, heads, head_dim, Nq, Nkv, nqpad,
              query_chunks);
  ATTN32_STRIDES_OK(q_row_stride, q_batch_stride, kv_row_stride, kv_batch_stride, o_row_stride, o_batch_stride, dq_row_stride,
                    dq_batch_stride, dkv_row_stride, dkv_batch_stride);
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 64)
    return attn32_bwd_launch<64>(Q, q_batch_stride, q_row_stride, KV, kv_batch_stride, kv_row_stride, dO, O, o_batch_stride,
                                 o_row_stride, lse2, delta, dQ, dq_batch_stride, dq_row_stride, dKV, dkv_batch_stride,
                                 dkv_row_stride, B, heads, Nq, Nkv, nqpad, query_chunks, scale, st);
  return attn32_bwd_launch<32>(Q, q_batch_stride, q_row_stride, KV, kv_batch_stride, kv_row_stride, dO, O, o_batch_stride,
                               o_row_stride, lse2, delta, dQ, dq_batch_stride, dq_row_stride, dKV, dkv_batch_stride,
                               dkv_row_stride, B, heads, Nq, Nkv, nqpad, query_chunks, scale, st);
}

}  // extern "C"
