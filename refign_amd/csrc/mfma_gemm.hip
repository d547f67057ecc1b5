// refign_amd/csrc/mfma_gemm.hip -- hand-written matrix-core GEMM for the token-wise Linear layers of MiT / the decode
// heads (mix_transformer.py:96-103,137-164; daformer.py:129-149) on gfx950.
//
//   Y[M,N] = epilogue( X[M,K] . W[N,K]^T )          bf16 or f16 in / out, fp32 accumulate ("NT": both operands K-major)
//
// forward : X = tokens, W = weight                   -> rfn_gemm_nt(x, weight, bias, ...)
// dgrad   : X = grad_y [T,N], W = weight^T [K,N]     (the host keeps a transposed 16-bit copy of every weight next to
//                                                     the plain one, refreshed once per optimizer step)
// wgrad   : dW[N,K] = grad_y[T,N]^T . x[T,K]         -> rfn_gemm_tn (reduction over the ROW index of both operands)
//
// NT kernel.  A workgroup (4 waves) owns a BM x BN tile of Y; a wave owns 64 x (BN/2) of it as 32x32x16 MFMA blocks
// computed TRANSPOSED, D[i = n][j = m] (A operand = W rows, B operand = X rows): a lane then holds, for ONE row m of Y,
// runs of 4 consecutive n -- bias, activation, residual and the 16-bit rounding happen in registers and the tile goes
// out as 16-byte row-contiguous stores (two lanes' runs joined by v_permlane32_swap), no LDS round trip.
// K is walked in 64-wide steps through a 2-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round
// trip, no ds_write pass); the DMA of step t+1 is in flight under the 16 MFMAs of step t, one barrier per step.
// LDS image of a tile: rows of 128 bytes (64 k), the 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7):
// every ds_read_b128 lane group ({0-3,12-15,20-27} ...) then covers 16 distinct 16-byte slots of the 256-byte bank
// row.  The DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE address.
#include <hip/hip_bf16.h>

#include "common.h"
#include "mfma.h"

namespace rfn {

struct GemmEpi {
  const uint16_t* bias;     // [N] or null, same 16-bit type as the operands
  const uint16_t* res;      // [M, ldy] residual added to the result, or null
  const float* rowscale;    // per-sample scale of the (acc + bias) term before the residual add, or null
  int rows_per_sample;      // sample of row m = m / rows_per_sample (for rowscale)
  int act;                  // 0 none, 1 ReLU, 2 GELU (erf)
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  return v;
}

template <int DT, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                      uint16_t* __restrict__ Y, int M, int N, int K, long ldx, long ldw,
                                                      long ldy, int tiles_n, GemmEpi epi) {
  using E = Elem<DT>;
  using vec8 = typename E::vec8;
  constexpr int IB = BN / 64;              // 32-wide n blocks per wave (waves are 2 (m) x 2 (n))
  constexpr int JB = BM / 64;              // 32-wide m blocks per wave
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
  constexpr int XI = BM / 32, WI = BN / 32;          // DMA instructions per wave and tile (8 rows each, 4 waves)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (wg / tiles_n) * BM, n0 = (wg % tiles_n) * BN;

  // ---- per-lane DMA sources: instruction q of this wave covers tile rows 8 (4 q + wave) .. + 7
  const int drow = lane >> 3, dchunk = lane & 7;
  const unsigned char* xsrc[XI];
  const unsigned char* wsrc[WI];
#pragma unroll
  for (int q = 0; q < XI; ++q) {
    const int r = 8 * (4 * q + wave) + drow;
    const int m = min(m0 + r, M - 1);
    xsrc[q] = (const unsigned char*)(X + (long)m * ldx) + 16 * (dchunk ^ ((r >> 1) & 7));
  }
#pragma unroll
  for (int q = 0; q < WI; ++q) {
    const int r = 8 * (4 * q + wave) + drow;
    const int n = min(n0 + r, N - 1);
    wsrc[q] = (const unsigned char*)(W + (long)n * ldw) + 16 * (dchunk ^ ((r >> 1) & 7));
  }
  auto issue = [&](int kt, int buf) {
    unsigned char* xs = smem + buf * STAGE;
    unsigned char* ws = xs + XBYTES;
#pragma unroll
    for (int q = 0; q < XI; ++q) lds_dma16(xsrc[q] + (long)kt * 128, xs + 1024 * (4 * q + wave));
#pragma unroll
    for (int q = 0; q < WI; ++q) lds_dma16(wsrc[q] + (long)kt * 128, ws + 1024 * (4 * q + wave));
  };

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (l & 31) of a 32-row block, chunk (2 ks + g) ^ swizzle(row); block offsets are multiples
  // of 32 rows and do not change the swizzle
  const int g = lane >> 5, frow = lane & 31, swz = (frow >> 1) & 7;
  const int xoff = (wm * (BM / 2) + frow) * 128, woff = XBYTES + (wn * (BN / 2) + frow) * 128;

  const int nk = K / 64;
  issue(0, 0);
  wait_dma_all();
  wg_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int coff = ((2 * ks + g) ^ swz) * 16;
      vec8 wf[IB], xf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i) wf[i] = *(const vec8*)(st + woff + i * 4096 + coff);
#pragma unroll
      for (int j = 0; j < JB; ++j) xf[j] = *(const vec8*)(st + xoff + j * 4096 + coff);
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = E::mma(wf[i], xf[j], acc[i][j]);
    }
    wait_dma_all();      // tile kt+1 has landed (this wave's part) ...
    wg_barrier();        // ... everybody's part, and everybody is done reading tile kt
  }

  // ---- epilogue, in registers: lane = one row m, register group k = 4 consecutive n
#pragma unroll
  for (int j = 0; j < JB; ++j) {
    const int m = m0 + wm * (BM / 2) + j * 32 + frow;
    const bool mok = m < M;
    const float rs = (epi.rowscale != nullptr && mok) ? epi.rowscale[m / epi.rows_per_sample] : 1.f;
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int nb = n0 + wn * (BN / 2) + i * 32;
      u32x2 pk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int n = nb + 8 * k + 4 * g;            // this lane's run: n .. n + 3
        float v[4] = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
        const bool ok = mok && n < N;               // N % 8 == 0: a run is all in or all out
        if (epi.bias != nullptr && ok) {
          float b[4];
          unpack4<DT>(*(const u32x2*)(epi.bias + n), b);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += b[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], epi.act);
        if (epi.res != nullptr && ok) {
          float rr[4];
          unpack4<DT>(*(const u32x2*)(epi.res + (long)m * ldy + n), rr);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = rr[e] + rs * v[e];
        }
        pk[k] = pack4<DT>(v[0], v[1], v[2], v[3]);
      }
      // join the runs of lane l (g = 0) and lane l + 32 (g = 1): afterwards lanes 0-31 hold n = nb + 16 p .. + 7 and
      // lanes 32-63 hold nb + 16 p + 8 .. + 15, 16 contiguous bytes each
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        u32x2 a = pk[2 * p], b = pk[2 * p + 1];
        auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
        const int n = nb + 16 * p + 8 * g;
        if (mok && n < N) {
          u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
          *(u32x4*)(Y + (long)m * ldy + n) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN kernel (weight gradient): P_s[N,K] = sum over the rows t of slab s of G[t, n] * X[t, k], fp32 partials per slab.
// Both operands are needed "transposed" (the reduction index is their ROW index), so the tiles are staged through LDS
// in the k-slot-major image [t / 4][column][4 t] by a register pass (8-byte ds_write per lane holding 4 rows of one
// column pair would need a transpose; instead each lane loads ONE column pair of 4 consecutive rows: 4 dword loads),
// and both MFMA operands are then plain 8-byte LDS reads: A[i = n][slots] = G^T, B[slots][j = k] = X.
// Slot order inside a 16-row k-step follows mfma.h: slot (g, e) = row {0-3, 8-11}[e] + 4 g.
// ---------------------------------------------------------------------------------------------------------------------
template <int DT, int BN, int BK>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ X,
                                                      float* __restrict__ P, int T, int N, int K, long ldg, long ldx,
                                                      int R, int tiles_k) {
  using E = Elem<DT>;
  constexpr int BT = 32;                   // rows of the reduction per stage (two 16-slot k-steps)
  constexpr int IB = BN / 64, JB = BK / 64;
  // LDS image per operand and stage: [BT / 4 quads][cols][4 rows] 16-bit = cols * 8 bytes per quad
  constexpr int GBYTES = (BT / 4) * BN * 8, XBYTES = (BT / 4) * BK * 8, STAGE = GBYTES + XBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = (N / BN) * tiles_k;
  const int slab = blockIdx.x / tiles, tile = blockIdx.x % tiles;   // slabs of one tile are far apart, tiles of one slab adjacent
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
  const long t0 = (long)slab * R;          // slab = R rows (R % 32 == 0), the last one may run past T: masked loads

  // register staging: thread handles column pair cp (2 columns = one dword per row) of quad q: 4 dword loads (rows
  // 4q .. 4q+3), transposed in registers into two 8-byte LDS writes (one per column)
  constexpr int GP = BN / 2, XP = BK / 2;              // column pairs per row
  constexpr int GITEMS = (BT / 4) * GP, XITEMS = (BT / 4) * XP;
  constexpr int GPT = GITEMS / 256, XPT = XITEMS / 256;   // items per thread
  static_assert(GITEMS % 256 == 0 && XITEMS % 256 == 0, "tile/threads");
  unsigned greg[GPT][4], xreg[XPT][4];
  auto load_stage = [&](int it) {
    const long tb = t0 + (long)it * BT;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / GP, cp = item % GP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        greg[u][r] = (tb + 4 * q + r < T) ? *(const unsigned*)(G + (tb + 4 * q + r) * ldg + n0 + 2 * cp) : 0u;
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / XP, cp = item % XP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        xreg[u][r] = (tb + 4 * q + r < T) ? *(const unsigned*)(X + (tb + 4 * q + r) * ldx + k0 + 2 * cp) : 0u;
    }
  };
  auto write_stage = [&](int buf) {
    unsigned char* gs = smem + buf * STAGE;
    unsigned char* xs = gs + GBYTES;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / GP, cp = item % GP;
      // even column: low halves of the 4 row dwords; odd column: high halves
      u32x4 o;
      o[0] = (greg[u][0] & 0xffffu) | (greg[u][1] << 16);
      o[1] = (greg[u][2] & 0xffffu) | (greg[u][3] << 16);
      o[2] = (greg[u][0] >> 16) | (greg[u][1] & 0xffff0000u);
      o[3] = (greg[u][2] >> 16) | (greg[u][3] & 0xffff0000u);
      *(u32x4*)(gs + (q * BN + 2 * cp) * 8) = o;
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / XP, cp = item % XP;
      u32x4 o;
      o[0] = (xreg[u][0] & 0xffffu) | (xreg[u][1] << 16);
      o[1] = (xreg[u][2] & 0xffffu) | (xreg[u][3] << 16);
      o[2] = (xreg[u][0] >> 16) | (xreg[u][1] & 0xffff0000u);
      o[3] = (xreg[u][2] >> 16) | (xreg[u][3] & 0xffff0000u);
      *(u32x4*)(xs + (q * BK + 2 * cp) * 8) = o;
    }
  };

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int g = lane >> 5, col = lane & 31;
  const int nit = (int)((min((long)R, (long)T - t0) + BT - 1) / BT);
  load_stage(0);
  write_stage(0);
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    if (it + 1 < nit) load_stage(it + 1);
    const unsigned char* gs = smem + buf * STAGE;
    const unsigned char* xs = gs + GBYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // slots (g, 0..3) = rows 16 ks + 4 g + 0..3 -> quad 4 ks + g; slots (g, 4..7) = rows 16 ks + 8 + 4 g .. -> quad 4 ks + 2 + g
      const int qa = 4 * ks + g, qb = 4 * ks + 2 + g;
      typename E::vec8 af[IB], bf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int c = wn * (BN / 2) + i * 32 + col;
        af[i] = join8<DT>(*(const u32x2*)(gs + (qa * BN + c) * 8), *(const u32x2*)(gs + (qb * BN + c) * 8));
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int c = wk * (BK / 2) + j * 32 + col;
        bf[j] = join8<DT>(*(const u32x2*)(xs + (qa * BK + c) * 8), *(const u32x2*)(xs + (qb * BK + c) * 8));
      }
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = E::mma(af[i], bf[j], acc[i][j]);
    }
    if (it + 1 < nit) write_stage(buf ^ 1);
    __syncthreads();
  }

  // D[i = n][j = k]: lane holds column k = .. + col, rows n = (r & 3) + 8 (r >> 2) + 4 g
  float* out = P + (long)slab * N * K;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int kk = k0 + wk * (BK / 2) + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        out[(long)n * K + kk] = acc[i][j][r];
      }
    }
}

template <int DT>
static int launch_nt(const void* X, const void* W, void* Y, long M, long N, long K, long ldx, long ldw, long ldy,
                     const GemmEpi& epi, hipStream_t s) {
  const bool wide = (N % 128 == 0);
  const int tiles_m = cdiv(M, 128), tiles_n = wide ? (int)(N / 128) : cdiv(N, 64);
  dim3 grid(tiles_m * tiles_n), block(256);
  if (wide)
    hipLaunchKernelGGL((gemm_nt_kernel<DT, 128, 128>), grid, block, 0, s, (const uint16_t*)X, (const uint16_t*)W,
                       (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, epi);
  else
    hipLaunchKernelGGL((gemm_nt_kernel<DT, 128, 64>), grid, block, 0, s, (const uint16_t*)X, (const uint16_t*)W,
                       (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, epi);
  return check_launch("gemm_nt");
}

template <int DT>
static int launch_tn(const void* G, const void* X, float* P, long T, long N, long K, long ldg, long ldx, int R,
                     hipStream_t s) {
  const int S = cdiv(T, R);
  dim3 block(256);
  if (N % 128 == 0 && K % 128 == 0) {
    dim3 grid((unsigned)((N / 128) * (K / 128) * S));
    hipLaunchKernelGGL((gemm_tn_kernel<DT, 128, 128>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X, P,
                       (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 128));
  } else {
    dim3 grid((unsigned)((N / 64) * (K / 64) * S));
    hipLaunchKernelGGL((gemm_tn_kernel<DT, 64, 64>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X, P,
                       (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 64));
  }
  return check_launch("gemm_tn");
}

}  // namespace rfn

extern "C" {

int rfn_gemm_nt(const void* X, const void* W, const void* bias, const void* res, const float* rowscale,
                int rows_per_sample, int act, void* Y, long M, long N, long K, long ldx, long ldw, long ldy, int dtype,
                rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(X && W && Y, "gemm_nt: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "gemm_nt: dtype %d (1 = bf16, 2 = f16)", dtype);
  RFN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0, "gemm_nt: M=%ld N=%ld K=%ld (K %% 64, N %% 8)", M, N, K);
  RFN_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldw >= K && ldy >= N,
              "gemm_nt: leading dimensions must be multiples of 8 elements");
  RFN_REQUIRE(M < (1L << 31) && N < (1L << 31), "gemm_nt: extent");
  RFN_REQUIRE(rowscale == nullptr || (res != nullptr && rows_per_sample > 0), "gemm_nt: rowscale needs res");
  RFN_REQUIRE(act >= 0 && act <= 2, "gemm_nt: act");
  GemmEpi epi{(const uint16_t*)bias, (const uint16_t*)res, rowscale, rows_per_sample > 0 ? rows_per_sample : 1, act};
  hipStream_t s = (hipStream_t)stream;
  return dtype == 1 ? launch_nt<1>(X, W, Y, M, N, K, ldx, ldw, ldy, epi, s)
                    : launch_nt<2>(X, W, Y, M, N, K, ldx, ldw, ldy, epi, s);
}

int rfn_gemm_tn(const void* G, const void* X, float* P, long T, long N, long K, long ldg, long ldx, int rows_per_slab,
                int dtype, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(G && X && P, "gemm_tn: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "gemm_tn: dtype %d", dtype);
  RFN_REQUIRE(T > 0 && T < (1L << 31) && rows_per_slab > 0 && rows_per_slab % 32 == 0,
              "gemm_tn: T=%ld rows_per_slab=%d (%% 32)", T, rows_per_slab);
  RFN_REQUIRE(N % 64 == 0 && K % 64 == 0 && ldg % 2 == 0 && ldx % 2 == 0, "gemm_tn: N=%ld K=%ld (%% 64)", N, K);
  hipStream_t s = (hipStream_t)stream;
  return dtype == 1 ? launch_tn<1>(G, X, P, T, N, K, ldg, ldx, rows_per_slab, s)
                    : launch_tn<2>(G, X, P, T, N, K, ldg, ldx, rows_per_slab, s);
}

}  // extern "C"
