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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace rfn {

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. N - 1
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct GemmEpi {
  // OUT32 kernels (fp32 result: the split-bf16 parity mode) read bias / res as fp32 through the same pointers
  const uint16_t* bias;     // [N] or null, same 16-bit type as the operands
  const uint16_t* res;      // [M, ldy] residual added to the result, or null
  const float* rowscale;    // per-sample scale of the (acc + bias) term before the residual add, or null
  int rows_per_sample;      // sample of row m = m / rows_per_sample (for rowscale)
  int act;                  // 0 none, 1 ReLU, 3 LeakyReLU(0.1)
};

// d/dz gelu(z) = Phi(z) + z phi(z), exact-erf GELU: erf by Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7) -- one v_rcp, one v_exp
// (shared by the erf and the density term), a dozen FMAs; used by the act = 4 epilogue (a 16-bit result)
__device__ __forceinline__ float gelu_grad(float z) {
  const float x = fabsf(z) * 0.70710678118654752440f;
  const float t = 1.0f / (1.0f + 0.3275911f * x);
  const float e = __expf(-x * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.0f - poly * e;
  const float cdf = 0.5f * (1.0f + (z < 0.f ? -erf_abs : erf_abs));
  return cdf + z * 0.3989422804014327f * e;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  // 1 = ReLU, 3 = LeakyReLU(0.1) (the matcher's decoders); GELU is NOT offered here: in the Mix-FFN it follows the
  // depthwise convolution (fused there, csrc/dwconv.hip), and its erf polynomial in this epilogue costs ~100 VGPRs
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 3) return v > 0.f ? v : 0.1f * v;
  return v;
}

// Implicit-GEMM view of a convolution over a channels-last tensor (dense GEMM: 1x1, stride 1 -- GATHER = false):
//   Y[m, n] = sum_{tap, c} X[b, oy s - p + ky d, ox s - p + kx d, c] * W[n, (tap, c)],   m = (b, oy, ox), zero padding.
// The reduction index k = tap * C + c is walked in 64-wide steps; a 16-byte DMA piece is 8 channels of ONE tap, so C % 8
// == 0 and the weight rows are [tap][c] zero-padded to a multiple of 64 (pieces past the last tap read the zero page).
// conv3x3.hip: RFN_OK / an error, or 1 = outside its domain
int launch_conv3x3_halo(const void* X, const void* W, const void* bias, void* Y, int B, int H, int Wd, int C, int N, long ldw,
                        long ldy, int act, int dtype, int out32, hipStream_t s);

struct ConvGeom {
  int H, W, C, OH, OW, KH, KW, stride, pad, dil;
  int C8;                  // C / 8
  unsigned c8_magic;       // ceil(2^32 / C8): piece index / C8 (exact for piece < 2^16)
  unsigned kw_magic;       // ceil(2^32 / KW)
  const void* zero;        // >= 16 bytes of zeros
  // transposed = 1: the DATA gradient of the convolution (rfn_conv2d_nhwc_dgrad).  X is then grad_y (B, H, W, C) = the
  // convolution's OUTPUT side, the result rows m are the convolution's INPUT pixels (b, iy, ix) (OH, OW = their extent) and
  //   DX[m, c] = sum_{tap, n} GY[b, (iy + pad - ky dil) / stride, (ix + pad - kx dil) / stride, n] * Wt[c, (tap, n)]
  // over the taps whose numerators are non-negative multiples of the stride (power of two: sshift) and land inside GY.
  int transposed, sshift;
};

// Persistent workgroups: a workgroup owns every G-th output tile (m-major, n fastest) and runs ONE software pipeline
// over all their K-steps through an NS-deep
// LDS ring: NS - 1 steps are in flight under the MFMAs of the current one -- also across tile boundaries, so the first
// steps of tile i+1 stream in under the last MFMAs and the whole epilogue of tile i.  The hand-off is a COUNTED
// `s_waitcnt vmcnt((NS - 2) * DMA instructions per step)` + one barrier per step; after an epilogue (its stores share the
// counter and need not retire in order with the loads) the wait is vmcnt(0).  Measured on MI355X (profiles/): K-steps of
// 64 (full 128-byte lines per row) with a 2-deep ring beat K-steps of 32 with a 4-deep ring by 1.3-1.6x on the K >= 512
// shapes -- half-line requests double the L2 request count -- so BK = 64, NS = 2 is what the host launches.

template <int DT, int BM, int BN, int BK, int NS, bool GATHER, int NW = 4, bool OUT32 = false>
__global__ __launch_bounds__(NW * 64) void gemm_nt_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W,
                                                      uint16_t* __restrict__ Y, int M, int N, int K, long ldx, long ldw,
                                                      long ldy, int tiles_n, int total_tiles, GemmEpi epi, ConvGeom cg) {
  using E = Elem<DT>;
  using vec8 = typename E::vec8;
  static_assert(BK == 32 || BK == 64, "K-step of 32 or 64");
  // profiling builds only (make FLAGS+=-DRFN_GEMM_PROFILE; tools/gemm_ablate.py): RFN_GEMM_ABLATE bit 1 no DMA, 2 no MFMA,
  // 4 no stores -- which phases of a launch overlap (round 3: none do, profiles/r03_gemm_ablation.txt)
#ifdef RFN_GEMM_PROFILE
  const int ablate = epi.act >> 8;
#else
  constexpr int ablate = 0;
#endif
  // NW waves as 2 (m) x NW / 2 (n); 8 waves (a 256 x 256 tile, wave tile 128 x 64) halve the LDS-DMA instructions a wave
  // issues per MFMA -- their issue cost, not the data volume, is what paces the 4-wave 128 x 128 tile (2 MFMAs per DMA)
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  constexpr int WNW = NW / 2;              // waves along n
  constexpr int IB = BN / (32 * WNW);      // 32-wide n blocks per wave
  constexpr int JB = BM / 64;              // 32-wide m blocks per wave
  constexpr int ROWB = BK * 2;             // bytes per LDS row
  constexpr int PPR = BK / 8;              // 16-byte pieces per row (4 or 8)
  constexpr int RPI = 64 / PPR;            // tile rows per DMA instruction (16 or 8)
  constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB, STAGE = XBYTES + WBYTES;
  constexpr int XI = BM / (NW * RPI), WI = BN / (NW * RPI);   // DMA instructions per wave and step
  constexpr int IPS = XI + WI;
  // piece c of row r sits at c ^ swizzle(r): rows of 128 B: (r >> 1) & 7, rows of 64 B: (r >> 2) & 3 -- either way every
  // ds_read_b128 lane group covers 16 distinct 16-byte slots of the 256-byte bank row
  auto swizzle = [](int r) { return BK == 64 ? (r >> 1) & 7 : (r >> 2) & 3; };
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  // tiles wg, wg + G, wg + 2 G, ...: at any time the resident workgroups work on CONSECUTIVE tiles, i.e. the n tiles of
  // one X row panel run side by side on one XCD (xcd_remap) and share the panel in its L2.  (Contiguous per-workgroup
  // ranges re-read the panel tiles_n times from HBM: the chip-wide working set of panels is far beyond 8 x 4 MB of L2
  // -- measured 1.6x slower on 81600 x 1280 -> 320.)
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int t_begin = wg;
  if (t_begin >= total_tiles) return;
  const int ntiles = (total_tiles - wg + G - 1) / G;
  const int nk = K / BK;

  // ---- per-lane DMA sources: instruction q of this wave covers tile rows RPI (4 q + wave) .. + RPI - 1; lane = (row,
  // 16-byte piece); the source piece of a lane is piece ^ swizzle(row), which does not depend on q
  const int drow = lane / PPR;
  const int dpiece = (lane % PPR) ^ swizzle(RPI * wave + drow);
  const unsigned char* xsrc[XI];
  const unsigned char* wsrc[WI];
  int iy0[GATHER ? XI : 1], ix0[GATHER ? XI : 1];
  auto setup = [&](int tile) {
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
    for (int q = 0; q < XI; ++q) {
      const int m = min(m0 + RPI * (NW * q + wave) + drow, M - 1);
      if constexpr (GATHER) {
        const int ohw = cg.OH * cg.OW;
        const int b = m / ohw, rem = m - b * ohw, oy = rem / cg.OW, ox = rem - oy * cg.OW;
        xsrc[q] = (const unsigned char*)(X + (long)b * cg.H * cg.W * cg.C);
        iy0[q] = cg.transposed ? oy + cg.pad : oy * cg.stride - cg.pad;
        ix0[q] = cg.transposed ? ox + cg.pad : ox * cg.stride - cg.pad;
      } else {
        xsrc[q] = (const unsigned char*)(X + (long)m * ldx) + 16 * dpiece;
      }
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) {
      const int n = min(n0 + RPI * (NW * q + wave) + drow, N - 1);
      wsrc[q] = (const unsigned char*)(W + (long)n * ldw) + 16 * dpiece;
    }
  };
  // (round 5) C % 64 == 0, forward gather: the 8 pieces of a K-step are 64 channels of ONE tap, and a tap lasts C / 64 K-steps --
  // the tap's source pointers (image bounds, row / column arithmetic: ~12 VALU per DMA piece, 4 VALU per MFMA on the 1024-channel
  // bottleneck convolution) are computed when the tap changes and ADVANCED by 128 bytes per K-step in between.  Measured: the
  // 40 x 135 x 240 x 1024 -> 256 bottleneck 6 520 -> 6 358 us (min 6 124 -> 5 966): the index arithmetic was mostly hidden under the
  // LDS-DMA issue that paces this kernel (8 DMA instructions per wave and K-step against 32 MFMAs).
  const bool fastc = GATHER && cg.transposed == 0 && (cg.C8 & 7) == 0;
  int f_tap = 0, f_cb = 0;
  const unsigned char* psrc[GATHER ? XI : 1];
  auto tap_sources = [&]() {
    if constexpr (GATHER) {
      const unsigned ky = cg.KW == 1 ? (unsigned)f_tap : __umulhi((unsigned)f_tap, cg.kw_magic), kx = (unsigned)f_tap - ky * cg.KW;
      const bool tap_ok = (int)ky < cg.KH;
      const int dy = (int)ky * cg.dil, dx = (int)kx * cg.dil;
#pragma unroll
      for (int q = 0; q < XI; ++q) {
        const int iy = iy0[q] + dy, ix = ix0[q] + dx;
        const bool ok = tap_ok && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
        psrc[q] = ok ? xsrc[q] + (long)(iy * cg.W + ix) * cg.C * 2 + 16 * dpiece : nullptr;
      }
    }
  };
  auto issue = [&](int kt, int buf) {
    unsigned char* xs = smem + buf * STAGE;
    unsigned char* ws = xs + XBYTES;
    if (ablate & 1) return;
    if constexpr (GATHER) {
      if (fastc) {
        if (kt == 0) {
          f_tap = 0;
          f_cb = 0;
          tap_sources();
        }
#pragma unroll
        for (int q = 0; q < XI; ++q)
          lds_dma16(psrc[q] != nullptr ? psrc[q] + (long)f_cb * 16 : (const unsigned char*)cg.zero, xs + 1024 * (NW * q + wave));
        f_cb += 8;
        if (f_cb == cg.C8) {
          f_cb = 0;
          ++f_tap;
          tap_sources();
        }
#pragma unroll
        for (int q = 0; q < WI; ++q) lds_dma16(wsrc[q] + (long)kt * ROWB, ws + 1024 * (NW * q + wave));
        return;
      }
      const unsigned j = (unsigned)(kt * PPR + dpiece);                     // 16-byte piece index along k
      const unsigned tap = cg.C8 == 1 ? j : __umulhi(j, cg.c8_magic), c8 = j - tap * cg.C8;
      const unsigned ky = cg.KW == 1 ? tap : __umulhi(tap, cg.kw_magic), kx = tap - ky * cg.KW;
      const bool tap_ok = (int)ky < cg.KH;
      const int dy = (int)ky * cg.dil, dx = (int)kx * cg.dil;
#pragma unroll
      for (int q = 0; q < XI; ++q) {
        int iy = iy0[q] + dy, ix = ix0[q] + dx;
        bool ok = tap_ok;
        if (cg.transposed) {                     // wave-uniform
          const int ty = iy0[q] - dy, tx = ix0[q] - dx, smask = (1 << cg.sshift) - 1;
          ok = ok && ty >= 0 && tx >= 0 && ((ty | tx) & smask) == 0;
          iy = ty >> cg.sshift;
          ix = tx >> cg.sshift;
        }
        ok = ok && (unsigned)iy < (unsigned)cg.H && (unsigned)ix < (unsigned)cg.W;
        const unsigned char* src = ok ? xsrc[q] + ((long)(iy * cg.W + ix) * cg.C + c8 * 8) * 2
                                      : (const unsigned char*)cg.zero;
        lds_dma16(src, xs + 1024 * (NW * q + wave));
      }
    } else {
#pragma unroll
      for (int q = 0; q < XI; ++q) lds_dma16(xsrc[q] + (long)kt * ROWB, xs + 1024 * (NW * q + wave));
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) lds_dma16(wsrc[q] + (long)kt * ROWB, ws + 1024 * (NW * q + wave));
  };

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (l & 31) of a 32-row block, piece (2 ks + g) ^ swizzle(row); block offsets are multiples
  // of 32 rows and do not change the swizzle
  const int g = lane >> 5, frow = lane & 31, swz = swizzle(frow);
  const int xoff = (wm * (BM / 2) + frow) * ROWB, woff = XBYTES + (wn * (BN / WNW) + frow) * ROWB;

  // producer cursor: the next K-step to issue, over all tiles of this workgroup
  const long S = (long)ntiles * nk;
  long p = 0;
  int p_tile = t_begin, p_kt = 0;
  auto produce = [&]() {
    if (p < S) {
      if (p_kt == 0) setup(p_tile);
      issue(p_kt, (int)(p % NS));
      ++p;
      if (++p_kt == nk) {
        p_kt = 0;
        p_tile += G;
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NS - 1; ++q) produce();

  int c_tile = t_begin, c_kt = 0;
  bool drained = false;                       // stores of an epilogue are outstanding: next wait is vmcnt(0)
  for (long s = 0; s < S; ++s) {
    // step s has landed once at most the two younger steps are still in flight (loads complete in order)
    if (NS == 2 || drained || s + (NS - 2) >= S) wait_dma_all();
    else wait_dma_upto<(NS - 2) * IPS>();
    wg_barrier();                             // ... everybody's part of it; and everybody is done reading step s - 1
    produce();                                // step s + 3 into the buffer step s - 1 occupied
    drained = false;
    const unsigned char* st = smem + (int)(s % NS) * STAGE;
    // fragment reads ONE k-sub-step ahead of the MFMAs (round 4, second half): left to itself hipcc re-uses one register set
    // and emits read, `s_waitcnt lgkmcnt(0)`, MFMA, read, wait, MFMA ... -- four LDS latencies in a row per K-step and wave,
    // which on the student's 64 x 64 tiles (one MFMA per sub-step) IS the K-step.  `sched_barrier` pins the order.
    // (one MFMA per sub-step -- the 64 x 64 tile: all four sub-steps' fragments up front, 32 registers it has to spare)
    constexpr int FD = (IB * JB == 1) ? BK / 16 : 2;     // fragment register sets
    vec8 wf[FD][IB], xf[FD][JB];
    auto read_frags = [&](int ks, int b) {
      const int coff = ((2 * ks + g) ^ swz) * 16;
#pragma unroll
      for (int i = 0; i < IB; ++i) wf[b][i] = *(const vec8*)(st + woff + i * 32 * ROWB + coff);
#pragma unroll
      for (int j = 0; j < JB; ++j) xf[b][j] = *(const vec8*)(st + xoff + j * 32 * ROWB + coff);
    };
#pragma unroll
    for (int ks = 0; ks < FD - 1; ++ks) read_frags(ks, ks);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + FD - 1 < BK / 16) read_frags(ks + FD - 1, (ks + FD - 1) % FD);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ablate & 2)) {
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
          for (int j = 0; j < JB; ++j) acc[i][j] = E::mma(wf[ks % FD][i], xf[ks % FD][j], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (++c_kt == nk) {
      // ---- epilogue.  Registers -> (bias, activation, 16-bit rounding) -> a per-wave LDS staging block -> row-contiguous
      // 16-byte stores: an instruction then writes 8 rows x 128 B (or 16 x 64 B) instead of 64 scattered 16-byte runs
      // (one cache-line request per run made the K <= 128 shapes store-bound).  The staging block lives in the LDS stage
      // that was just consumed, hence the barrier; the next steps' DMA (other stage) is in flight meanwhile.  `tile` is
      // laundered through an empty asm so that the address arithmetic below (loop-invariant for the step loop) is not
      // hoisted out of this block, where it would hold ~80 VGPRs.
      int tl = c_tile;
      asm volatile("" : "+s"(tl));
      const int m0 = (tl / tiles_n) * BM, n0 = (tl % tiles_n) * BN;
      constexpr int WN = BN / WNW;                     // columns of a wave's tile
      constexpr int PITCH = WN * 2 + 16;               // staging row pitch in bytes (16-byte aligned, 2-way at worst)
      constexpr int RPP = 64 / (WN / 8);               // rows per store instruction (8 pieces of 16 B per 64-column row)
      static_assert(NW * 32 * PITCH <= STAGE, "staging block fits the consumed stage");
      if constexpr (OUT32) {
        // fp32 result straight from the accumulators (parity mode: correctness path, not a fast one): a lane's run of 4
        // consecutive n of row m is one 16-byte store; bias / residual are fp32
        float* Y32 = (float*)Y;
        const float* bias32 = (const float*)epi.bias;
        const float* res32 = (const float*)epi.res;
#pragma unroll
        for (int j = 0; j < JB; ++j)
#pragma unroll
          for (int i = 0; i < IB; ++i) {
            const int m = m0 + wm * (BM / 2) + j * 32 + frow;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int n = n0 + wn * WN + i * 32 + 8 * k + 4 * g;
              f32x4 v = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
              if (m < M && n < N) {
                if (bias32 != nullptr) v += *(const f32x4*)(bias32 + n);
                if (epi.act != 0) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], epi.act);
                }
                if (res32 != nullptr || epi.rowscale != nullptr) {
                  const float rs = epi.rowscale != nullptr ? epi.rowscale[m / epi.rows_per_sample] : 1.f;
                  const f32x4 rr = res32 != nullptr ? *(const f32x4*)(res32 + (long)m * ldy + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                  v = rr + rs * v;
                }
                *(f32x4*)(Y32 + (long)m * ldy + n) = v;
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
        c_kt = 0;
        c_tile += G;
        drained = true;
        continue;
      }
      wg_barrier();                                    // every wave is done reading this stage's operands
      unsigned char* stg = const_cast<unsigned char*>(st) + wave * 32 * PITCH;
#pragma unroll
      for (int j = 0; j < JB; ++j) {
#pragma unroll
        for (int i = 0; i < IB; ++i) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cl = i * 32 + 8 * k + 4 * g;       // this lane's run: local columns cl .. cl + 3 of row frow
            const int n = n0 + wn * WN + cl;
            float v[4] = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
            if (epi.bias != nullptr && n < N) {          // N % 8 == 0: a run is all in or all out
              float b[4];
              unpack4<DT>(*(const u32x2*)(epi.bias + n), b);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += b[e];
            }
            if ((epi.act & 255) != 0) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], epi.act & 255);
            }
            *(u32x2*)(stg + frow * PITCH + cl * 2) = pack4<DT>(v[0], v[1], v[2], v[3]);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own staging writes (no other wave touches them)
#pragma unroll
        for (int it = 0; it < 32 / RPP; ++it) {
          const int row = it * RPP + lane / (WN / 8), piece = lane % (WN / 8);
          const int m = m0 + wm * (BM / 2) + j * 32 + row, n = n0 + wn * WN + piece * 8;
          u32x4 o = *(const u32x4*)(stg + row * PITCH + piece * 16);
          if (m < M && n < N) {
            if (epi.res != nullptr || epi.rowscale != nullptr) {
              const float rs = epi.rowscale != nullptr ? epi.rowscale[m / epi.rows_per_sample] : 1.f;
              const u32x4 rr = epi.res != nullptr ? *(const u32x4*)(epi.res + (long)m * ldy + n) : u32x4{0u, 0u, 0u, 0u};
              float a[4], b[4], c[4], d[4];
              unpack4<DT>(u32x2{o[0], o[1]}, a);
              unpack4<DT>(u32x2{o[2], o[3]}, b);
              unpack4<DT>(u32x2{rr[0], rr[1]}, c);
              unpack4<DT>(u32x2{rr[2], rr[3]}, d);
              u32x2 lo, hi;
              if (epi.act == 4) {
                // `res` carries the PRE-ACTIVATION z of an exact-erf GELU that sits in front of this layer's input: the result is
                // the gradient with respect to z, (rs * branch) * gelu'(z) -- the Mix-FFN's fc2 input gradient handed straight to
                // the depthwise backward (mix_transformer.py:99-102), no separate gelu_backward pass
                lo = pack4<DT>(rs * a[0] * gelu_grad(c[0]), rs * a[1] * gelu_grad(c[1]), rs * a[2] * gelu_grad(c[2]),
                               rs * a[3] * gelu_grad(c[3]));
                hi = pack4<DT>(rs * b[0] * gelu_grad(d[0]), rs * b[1] * gelu_grad(d[1]), rs * b[2] * gelu_grad(d[2]),
                               rs * b[3] * gelu_grad(d[3]));
              } else {
                lo = pack4<DT>(c[0] + rs * a[0], c[1] + rs * a[1], c[2] + rs * a[2], c[3] + rs * a[3]);
                hi = pack4<DT>(d[0] + rs * b[0], d[1] + rs * b[1], d[2] + rs * b[2], d[3] + rs * b[3]);
              }
              o = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            if (!(ablate & 4)) *(u32x4*)(Y + (long)m * ldy + n) = o;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging reads done before the next j block overwrites
      }
      c_kt = 0;
      c_tile += G;
      drained = true;
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
// GATHER: the weight gradient of a convolution (rfn_conv2d_nhwc_wgrad): row t of the X operand is the im2col row of output
// pixel t = (b, oy, ox), column k = (tap, c) reads x[b, oy s - p + ky d, ox s - p + kx d, c] (zero outside the image / past
// the last tap) -- no im2col buffer; a thread's column pair is fixed, so its (tap, c) is decomposed once.
struct WgradGeom {
  int H, W, C, OH, OW, KH, KW, stride, pad, dil;
  unsigned c_magic, kw_magic;                          // ceil(2^32 / d) for C, KW (dividends < 2^16)
};

template <int DT, int BN, int BK, bool GATHER = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ X,
                                                      float* __restrict__ P, int T, int N, int K, long ldg, long ldx,
                                                      int R, int tiles_k, int accumulate, float* __restrict__ gbias,
                                                      const float* __restrict__ rowscale, int rows_per_sample,
                                                      WgradGeom wg, int xcd) {
  using E = Elem<DT>;
  constexpr int BT = 32;                   // rows of the reduction per stage (two 16-slot k-steps)
  constexpr int IB = BN / 64, JB = BK / 64;
  // LDS image per operand and stage: [BT / 4 quads][cols][4 rows] 16-bit = cols * 8 bytes per quad
  constexpr int GBYTES = (BT / 4) * BN * 8, XBYTES = (BT / 4) * BK * 8, STAGE = GBYTES + XBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = (N / BN) * tiles_k;
  // slab-major logical order, XCD-aware (mfma.h xcd_remap: consecutive logical ids on ONE XCD): the tiles of a slab share
  // its rows of G and X (a slab of both operands fits the XCD's 4 MB L2), so each operand byte crosses the fabric once instead
  // of once per XCD.  xcd = 0 (RFN_GEMM_TN_XCD=0): dispatch order, as in the first two rounds.
  const int lid = xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int slab = lid / tiles, tile = lid % tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
  const long t0 = (long)slab * R;          // slab = R rows (R % 32 == 0), the last one may run past T: masked loads

  // register staging: thread handles column pair cp (2 columns = one dword per row) of quad q: 4 dword loads (rows
  // 4q .. 4q+3), transposed in registers into two 8-byte LDS writes (one per column)
  constexpr int GP = BN / 2, XP = BK / 2;              // column pairs per row
  constexpr int GITEMS = (BT / 4) * GP, XITEMS = (BT / 4) * XP;
  constexpr int GPT = GITEMS / 256, XPT = XITEMS / 256;   // items per thread
  static_assert(GITEMS % 256 == 0 && XITEMS % 256 == 0, "tile/threads");
  unsigned greg[GPT][4], xreg[XPT][4];
  // bias gradient = column sums of G: every thread stages the same column pair in every stage (256 % GP == 0), so it
  // keeps the running sums of its two columns; only the k-tile-0 workgroups of each (n tile, slab) contribute
  const bool do_bias = (gbias != nullptr) && (k0 == 0);
  float bsum[2] = {0.f, 0.f};
  // GATHER: (tap, channel) of this thread's X column pair (the same in every stage and for all its items)
  int gdy = 0, gdx = 0, gc = 0;
  bool gtap_ok = true;
  if constexpr (GATHER) {
    const unsigned k = (unsigned)(k0 + 2 * (threadIdx.x % XP));
    const unsigned tap = wg.C == 1 ? k : __umulhi(k, wg.c_magic);
    gc = (int)(k - tap * wg.C);
    const unsigned ky = wg.KW == 1 ? tap : __umulhi(tap, wg.kw_magic), kx = tap - ky * wg.KW;
    gtap_ok = (int)ky < wg.KH;
    gdy = (int)ky * wg.dil - wg.pad;
    gdx = (int)kx * wg.dil - wg.pad;
  }
  // (image, oy, ox) of the rows this thread stages, advanced by BT per stage (no division in the loop)
  int gb[GATHER ? XPT : 1][4], goy[GATHER ? XPT : 1][4], gox[GATHER ? XPT : 1][4];
  if constexpr (GATHER) {
#pragma unroll
    for (int u = 0; u < XPT; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long t = t0 + 4 * ((u * 256 + (int)threadIdx.x) / XP) + r;
        const long ohw = (long)wg.OH * wg.OW;
        gb[u][r] = (int)(t / ohw);
        const int rem = (int)(t - gb[u][r] * ohw);
        goy[u][r] = rem / wg.OW;
        gox[u][r] = rem - goy[u][r] * wg.OW;
      }
  }
  auto load_stage = [&](int it) {
    const long tb = t0 + (long)it * BT;
#pragma unroll
    for (int u = 0; u < GPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / GP, cp = item % GP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        greg[u][r] = (tb + 4 * q + r < T) ? *(const unsigned*)(G + (tb + 4 * q + r) * ldg + n0 + 2 * cp) : 0u;
      if (rowscale != nullptr) {
        // G <- diag(rowscale[row / rows_per_sample]) G (the stochastic-depth scale of the branch this gradient belongs
        // to): the two 16-bit values of each row dword, scaled in fp32 and rounded back like the eager g * mask
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long row = tb + 4 * q + r;
          const float sc = row < T ? rowscale[row / rows_per_sample] : 0.f;
          float f[4];
          unpack4<DT>(u32x2{greg[u][r], 0u}, f);
          greg[u][r] = pack4<DT>(f[0] * sc, f[1] * sc, 0.f, 0.f)[0];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < XPT; ++u) {
      const int item = u * 256 + threadIdx.x, q = item / XP, cp = item % XP;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (GATHER) {
          const long t = tb + 4 * q + r;
          const int iy = goy[u][r] * wg.stride + gdy, ix = gox[u][r] * wg.stride + gdx;
          const bool ok = t < T && gtap_ok && (unsigned)iy < (unsigned)wg.H && (unsigned)ix < (unsigned)wg.W;
          xreg[u][r] = ok ? *(const unsigned*)(X + (((long)gb[u][r] * wg.H + iy) * wg.W + ix) * wg.C + gc) : 0u;
          gox[u][r] += BT;                              // the same row of the NEXT stage
          while (gox[u][r] >= wg.OW) {
            gox[u][r] -= wg.OW;
            if (++goy[u][r] == wg.OH) {
              goy[u][r] = 0;
              ++gb[u][r];
            }
          }
        } else {
          xreg[u][r] = (tb + 4 * q + r < T) ? *(const unsigned*)(X + (tb + 4 * q + r) * ldx + k0 + 2 * cp) : 0u;
        }
      }
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
      if (do_bias) {
        float f[4];
        unpack4<DT>(u32x2{o[0], o[1]}, f);
        bsum[0] += (f[0] + f[1]) + (f[2] + f[3]);
        unpack4<DT>(u32x2{o[2], o[3]}, f);
        bsum[1] += (f[0] + f[1]) + (f[2] + f[3]);
      }
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
  float* out = accumulate ? P : P + (long)slab * N * K;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int kk = k0 + wk * (BK / 2) + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (accumulate) atomicAdd(out + (long)n * K + kk, acc[i][j][r]);     // 128-byte coalesced fp32 atomics
        else out[(long)n * K + kk] = acc[i][j][r];
      }
    }
  if (do_bias) {                                   // reduce the per-thread column sums over the row quads, one atomic per column
    float* red = (float*)smem;                     // the staging ring is free now (last __syncthreads above)
    red[threadIdx.x * 2] = bsum[0];
    red[threadIdx.x * 2 + 1] = bsum[1];
    __syncthreads();
    if (threadIdx.x < BN) {
      const int cp = threadIdx.x >> 1, e = threadIdx.x & 1;
      float sum = 0.f;
      for (int t = cp; t < 256; t += GP) sum += red[t * 2 + e];
      atomicAdd(gbias + n0 + threadIdx.x, sum);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// TN kernel, second generation (round 3): the same product and the same LDS image / MFMA operand order as above, but the
// operand tiles are fetched with 16-BYTE loads and transposed in registers.  128 threads stage the G tile, 128 the X tile;
// a thread owns ONE 4-row x 8-column block per stage: 4 x global_load_dwordx4 (a 16-lane group reads 256 contiguous bytes
// of a row), 16 v_perm_b32, 4 x ds_write_b128 -- a quarter of the load instructions per element of the first generation
// (4-byte loads), which was paced by load issue.  The 16-byte piece j (column pair j of the block) of block cv is stored at
// piece slot j ^ ((cv >> 1) & 3): the 8 lanes of a ds_write_b128 group then cover the eight 16-byte slots of a 128-byte
// bank row, and a fragment read stays a permutation inside its 256-byte span (conflict-free both ways).
// BT = rows of the reduction per stage: 32 for the 128 x 128 tile, 64 for the 64 x 64 tile (128 blocks per operand either
// way).  GATHER: im2col rows of a convolution (see WgradGeom); a thread's block is 8 channels of ONE tap of 4 consecutive
// output pixels.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned perm_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }  // {a.lo, b.lo}
__device__ __forceinline__ unsigned perm_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }  // {a.hi, b.hi}

template <int DT, int BN, int BK, int BT, bool GATHER>
__global__ __launch_bounds__(256) void gemm_tn2_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ X,
                                                       float* __restrict__ P, int T, int N, int K, long ldg, long ldx,
                                                       int R, int tiles_k, int accumulate, float* __restrict__ gbias,
                                                       const float* __restrict__ rowscale, int rows_per_sample,
                                                       WgradGeom wg, int xcd) {
  using E = Elem<DT>;
  constexpr int IB = BN / 64, JB = BK / 64;
  constexpr int NQ = BT / 4;                                     // row quads per stage
  constexpr int GBYTES = NQ * BN * 8, XBYTES = NQ * BK * 8, STAGE = GBYTES + XBYTES;
  static_assert(NQ * (BN / 8) == 128 && NQ * (BK / 8) == 128, "128 blocks per operand and stage");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = (N / BN) * tiles_k;
  // slab-major logical order, XCD-aware (mfma.h xcd_remap: consecutive logical ids on ONE XCD): the tiles of a slab share
  // its rows of G and X (a slab of both operands fits the XCD's 4 MB L2), so each operand byte crosses the fabric once instead
  // of once per XCD.  xcd = 0 (RFN_GEMM_TN_XCD=0): dispatch order, as in the first two rounds.
  const int lid = xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int slab = lid / tiles, tile = lid % tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
  const long t0 = (long)slab * R;

  // staging role: threads 0..127 the G tile, 128..255 the X tile; block (q, cv) = 4 rows x 8 columns
  const bool isx = threadIdx.x >= 128;
  const int bid = threadIdx.x & 127;
  const int cols8 = (isx ? BK : BN) / 8;
  const int cv = bid % cols8, q = bid / cols8;
  const int sw = (cv >> 1) & 3;
  const bool do_bias = (gbias != nullptr) && (k0 == 0) && !isx;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;

  // GATHER (X threads): (tap, channel) of the block's 8 columns, and the (image, oy, ox) of its 4 rows, advanced per stage
  int gdy = 0, gdx = 0, gc = 0;
  bool gtap_ok = true;
  int gb[4], goy[4], gox[4];
  if constexpr (GATHER) {
    const unsigned k = (unsigned)(k0 + 8 * cv);
    const unsigned tap = __umulhi(k, wg.c_magic);
    gc = (int)(k - tap * wg.C);
    const unsigned ky = wg.KW == 1 ? tap : __umulhi(tap, wg.kw_magic), kx = tap - ky * wg.KW;
    gtap_ok = (int)ky < wg.KH;
    gdy = (int)ky * wg.dil - wg.pad;
    gdx = (int)kx * wg.dil - wg.pad;
    const long ohw = (long)wg.OH * wg.OW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long t = t0 + 4 * q + r;
      gb[r] = (int)(t / ohw);
      const int rem = (int)(t - gb[r] * ohw);
      goy[r] = rem / wg.OW;
      gox[r] = rem - goy[r] * wg.OW;
    }
  }
  const uint16_t* base = isx ? X + k0 + 8 * cv : G + n0 + 8 * cv;
  const long ld = isx ? ldx : ldg;
  const long tend = min((long)T, t0 + R);             // a slab need not be a multiple of BT rows: mask at its end
  u32x4 reg[4];
  auto load_stage = [&](int it) {
    const long tb = t0 + (long)it * BT + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long t = tb + r;
      bool ok = t < tend;
      const uint16_t* src;
      if (GATHER && isx) {
        const int iy = goy[r] * wg.stride + gdy, ix = gox[r] * wg.stride + gdx;
        ok = ok && gtap_ok && (unsigned)iy < (unsigned)wg.H && (unsigned)ix < (unsigned)wg.W;
        src = X + (((long)gb[r] * wg.H + iy) * wg.W + ix) * wg.C + gc;
        gox[r] += BT;
        while (gox[r] >= wg.OW) {
          gox[r] -= wg.OW;
          if (++goy[r] == wg.OH) {
            goy[r] = 0;
            ++gb[r];
          }
        }
      } else {
        src = base + t * ld;
      }
      reg[r] = ok ? *(const u32x4*)src : u32x4{0u, 0u, 0u, 0u};
    }
    if (rowscale != nullptr && !isx) {
      // G <- diag(rowscale[row / rows_per_sample]) G, scaled in fp32 and rounded back like the eager g * mask
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = tb + r;
        const float sc = row < tend ? rowscale[row / rows_per_sample] : 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float f[4];
          unpack4<DT>(u32x2{reg[r][2 * h], reg[r][2 * h + 1]}, f);
          const u32x2 o = pack4<DT>(f[0] * sc, f[1] * sc, f[2] * sc, f[3] * sc);
          reg[r][2 * h] = o[0];
          reg[r][2 * h + 1] = o[1];
        }
      }
    }
  };
  auto write_stage = [&](int buf) {
    unsigned char* dst = smem + buf * STAGE + (isx ? GBYTES : 0) + (q * (isx ? BK : BN) + 8 * cv) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // column pair j: dword j of the four rows -> {even column: 4 rows, odd column: 4 rows}
      u32x4 o;
      o[0] = perm_lo(reg[0][j], reg[1][j]);
      o[1] = perm_lo(reg[2][j], reg[3][j]);
      o[2] = perm_hi(reg[0][j], reg[1][j]);
      o[3] = perm_hi(reg[2][j], reg[3][j]);
      *(u32x4*)(dst + 16 * (j ^ sw)) = o;
      if (do_bias) {
        float f[4];
        unpack4<DT>(u32x2{o[0], o[1]}, f);
        bsum[2 * j] += (f[0] + f[1]) + (f[2] + f[3]);
        unpack4<DT>(u32x2{o[2], o[3]}, f);
        bsum[2 * j + 1] += (f[0] + f[1]) + (f[2] + f[3]);
      }
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
  // byte offset of column c inside a quad's row of the image (swizzled 16-byte pieces inside 64-byte blocks)
  auto coff = [](int c) { return (c & ~7) * 8 + 16 * (((c >> 1) & 3) ^ ((c >> 4) & 3)) + 8 * (c & 1); };
  int aoff[IB], boff[JB];
#pragma unroll
  for (int i = 0; i < IB; ++i) aoff[i] = coff(wn * (BN / 2) + i * 32 + col);
#pragma unroll
  for (int j = 0; j < JB; ++j) boff[j] = coff(wk * (BK / 2) + j * 32 + col);

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
    for (int ks = 0; ks < BT / 16; ++ks) {
      const int qa = 4 * ks + g, qb = 4 * ks + 2 + g;
      typename E::vec8 af[IB], bf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i)
        af[i] = join8<DT>(*(const u32x2*)(gs + qa * BN * 8 + aoff[i]), *(const u32x2*)(gs + qb * BN * 8 + aoff[i]));
#pragma unroll
      for (int j = 0; j < JB; ++j)
        bf[j] = join8<DT>(*(const u32x2*)(xs + qa * BK * 8 + boff[j]), *(const u32x2*)(xs + qb * BK * 8 + boff[j]));
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = E::mma(af[i], bf[j], acc[i][j]);
    }
    if (it + 1 < nit) write_stage(buf ^ 1);
    __syncthreads();
  }

  float* out = accumulate ? P : P + (long)slab * N * K;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int kk = k0 + wk * (BK / 2) + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (accumulate) atomicAdd(out + (long)n * K + kk, acc[i][j][r]);
        else out[(long)n * K + kk] = acc[i][j][r];
      }
    }
  if (gbias != nullptr && k0 == 0) {             // the G threads' column sums over their row quads, one atomic per column
    float* red = (float*)smem;                   // [128 blocks][8]
    if (!isx) {
#pragma unroll
      for (int e = 0; e < 8; ++e) red[bid * 8 + e] = bsum[e];
    }
    __syncthreads();
    if (threadIdx.x < BN) {
      const int c8 = threadIdx.x >> 3, e = threadIdx.x & 7;
      float sum = 0.f;
      for (int qq = 0; qq < NQ; ++qq) sum += red[(qq * (BN / 8) + c8) * 8 + e];
      atomicAdd(gbias + n0 + threadIdx.x, sum);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// TN kernel, third generation (round 3): LDS-DMA staging + hardware transpose reads.
// The weight gradient needs both operands with the reduction index t as the slot index of the MFMA operands, i.e.
// "transposed" against their row-major storage; generations 1 and 2 transposed in registers on the way to LDS (16 v_perm,
// 4 ds_write_b128 per thread and stage) and were bound by exactly that -- per CU one workgroup-step per ~0.45 us whatever
// the number of slabs, the prefetch depth or the occupancy (PMC: 9 % MFMA, 53 % waiting; VALU and LDS pipe each about half
// of the step).  gfx950's `ds_read_b64_tr_b16` does the transpose in the LDS read: within a 16-lane group lane p supplies
// the address of 4 contiguous 16-bit elements and lane i receives element i % 4 of lanes i / 4 + 4 j, j = 0..3 (probed:
// tools/micro/tr_probe.hip).  With lane p pointing at row p / 4, columns 4 (p % 4) .. + 3 of a [4 rows][16 columns] block
// lane i gets rows 0..3 of column i: 4 consecutive t of ONE column -- half an MFMA operand.  So the tiles go to LDS ROW-MAJOR,
// as they lie in memory, by LDS-DMA (no registers, no VALU), and a fragment is two transpose reads.
// LDS image of an operand tile: [BT rows][cols], 16-byte pieces; piece c of row r at slot c ^ swz(r), swz(r) = 4 ((r >> 1) & 1)
// for 128-byte rows (64 columns), 4 (r & 3) for 256-byte rows (128 columns): the 16 pieces a 32-lane half of a transpose read
// touches (4 rows x 64 bytes) are 16 distinct slots of the 256-byte bank row.
// Rows past the end of the slab / of T are DMA'd from the zero page.  rowscale (stochastic depth): see SEG below -- the slab's
// scales are put in LDS before the loop (no global load inside the DMA loop: mfma.h pin_loaded).  Bias gradient: column sums
// of the G fragments (per sample, scaled like the products), by the waves with wk == 0 of the tiles with k0 == 0.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x2 lds_read_tr16(unsigned addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}

// GATHER: the weight gradient of a convolution -- row t of the X operand is the im2col row of output pixel t = (b, oy, ox),
// a 16-byte piece is 8 channels of ONE tap (C % 8 == 0): the DMA source of (row, piece) is x[b, oy s - p + ky d, ox s - p + kx d,
// c .. c + 7] or the zero page; a lane's pieces (tap, channel) are fixed, its rows advance by BT per stage.
// SEG (rowscale given): the slab's rows are walked sample by sample -- a stage never crosses a sample boundary (its rows
// past the end of the sample come from the zero page), the products of a sample go to a second accumulator block and are
// folded into the result with the sample's scale when the sample ends: sum_s scale_s (G_s^T X_s) in fp32.  (The first
// version scaled the G FRAGMENTS -- unpack, multiply, round, repack, 40 VALU instructions per 16-row step in every one of
// the K / 64 tiles that share a G panel: 8 160 x 320 x 1 280 took 42 us against 25 us without a scale.)
// (the kernel's body as a device function: `bid` of `nblocks` = this workgroup's place in ITS problem's grid -- the whole
// launch for gemm_tn3_kernel, a sub-range of it for gemm_tn3_group_kernel)
template <int DT, int BN, int BK, int BT, bool GATHER = false, bool SEG = false>
__device__ __forceinline__ void gemm_tn3_body(const uint16_t* __restrict__ G, const uint16_t* __restrict__ X,
                                              float* __restrict__ P, int T, int N, int K, long ldg, long ldx,
                                              int R, int tiles_k, int accumulate, float* __restrict__ gbias,
                                              const float* __restrict__ rowscale, int rows_per_sample,
                                              const void* zero, int xcd, const WgradGeom& wg, int bid, int nblocks) {
  using E = Elem<DT>;
  constexpr int IB = BN / 64, JB = BK / 64;
  constexpr int GROW = BN * 2, XROW = BK * 2;                      // bytes per LDS row
  constexpr int GBYTES = BT * GROW, XBYTES = BT * XROW, STAGE = GBYTES + XBYTES;
  constexpr int GPPR = BN / 8, XPPR = BK / 8;                      // 16-byte pieces per row
  constexpr int GI = BT * GPPR / 256, XI = BT * XPPR / 256;        // DMA instructions per wave and stage
  static_assert(GI >= 1 && XI >= 1 && (BN == 64 || BN == 128) && (BK == 64 || BK == 128), "tile");
  constexpr int kMaxScales = 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  __shared__ float sscale[kMaxScales];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int tiles = (N / BN) * tiles_k;
  const int lid = xcd ? xcd_remap(bid, nblocks) : bid;
  const int slab = lid / tiles, tile = lid % tiles;
  const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BK;
  const long t0 = (long)slab * R;
  const long tend = min((long)T, t0 + R);
  static_assert(!(SEG && GATHER), "sample segments: plain operands only");
  auto gswz = [](int r) { return BN == 64 ? ((r >> 1) & 1) << 2 : (r & 3) << 2; };
  auto xswz = [](int r) { return BK == 64 ? ((r >> 1) & 1) << 2 : (r & 3) << 2; };

  // the slab's stochastic-depth scales: sscale[i] = rowscale[s0 + i], s0 = sample of the slab's first row
  const long s0 = SEG ? t0 / rows_per_sample : 0;
  // end of the sample segment that starts at row a (a = t0 or a multiple of rows_per_sample)
  auto seg_end = [&](long a) { return SEG ? min(tend, (a / rows_per_sample + 1) * rows_per_sample) : tend; };
  int nit = 0;
  if constexpr (SEG) {
    const long s1 = (tend - 1) / rows_per_sample;
    if ((int)threadIdx.x <= (int)(s1 - s0) && threadIdx.x < kMaxScales) sscale[threadIdx.x] = rowscale[s0 + threadIdx.x];
    for (long a = t0; a < tend;) {
      const long e = seg_end(a);
      nit += (int)((e - a + BT - 1) / BT);
      a = e;
    }
  } else {
    nit = (int)((tend - t0 + BT - 1) / BT);
  }
  // GATHER: per X instruction u of this lane: (dy, dx, channel) of its piece, (image, oy, ox) of its row in stage 0
  int gdy[GATHER ? XI : 1], gdx[GATHER ? XI : 1], gch[GATHER ? XI : 1], gb[GATHER ? XI : 1], goy[GATHER ? XI : 1],
      gox[GATHER ? XI : 1];
  bool gok[GATHER ? XI : 1];
  if constexpr (GATHER) {
    const long ohw = (long)wg.OH * wg.OW;
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int q = XI * wave + u, r = q * (64 / XPPR) + lane / XPPR, c = lane % XPPR;
      const unsigned k = (unsigned)(k0 + 8 * (c ^ xswz(r)));
      const unsigned tap = __umulhi(k, wg.c_magic);
      gch[u] = (int)(k - tap * wg.C);
      const unsigned ky = wg.KW == 1 ? tap : __umulhi(tap, wg.kw_magic), kx = tap - ky * wg.KW;
      gok[u] = (int)ky < wg.KH;
      gdy[u] = (int)ky * wg.dil - wg.pad;
      gdx[u] = (int)kx * wg.dil - wg.pad;
      const long t = t0 + r;
      gb[u] = (int)(t / ohw);
      const int rem = (int)(t - gb[u] * ohw);
      goy[u] = rem / wg.OW;
      gox[u] = rem - goy[u] * wg.OW;
    }
  }
  // ---- DMA: instruction u of this wave moves 64 pieces = 64 / PPR rows of the tile; lane = (row, LDS piece slot), it fetches
  // source piece slot ^ swz(row)
  long itb = t0, iend = seg_end(t0);                 // issue cursor: first row of the next stage, end of its sample segment
  auto issue = [&](int it, int buf) {
    unsigned char* gs = smem + buf * STAGE;
    unsigned char* xs = gs + GBYTES;
    const long tb = SEG ? itb : t0 + (long)it * BT;
    const long tlim = SEG ? iend : tend;
    if constexpr (SEG) {                               // (stages are issued in order)
      itb += BT;
      if (itb >= iend) {
        itb = iend;
        iend = seg_end(iend);
      }
    }
#pragma unroll
    for (int u = 0; u < GI; ++u) {
      const int q = GI * wave + u, r = q * (64 / GPPR) + lane / GPPR, c = lane % GPPR;
      const long t = tb + r;
      const void* src = t < tlim ? (const void*)(G + t * ldg + n0 + 8 * (c ^ gswz(r))) : zero;
      lds_dma16(src, gs + q * 1024);
    }
#pragma unroll
    for (int u = 0; u < XI; ++u) {
      const int q = XI * wave + u, r = q * (64 / XPPR) + lane / XPPR, c = lane % XPPR;
      const long t = tb + r;
      const void* src;
      if constexpr (GATHER) {
        const int iy = goy[u] * wg.stride + gdy[u], ix = gox[u] * wg.stride + gdx[u];
        const bool ok = t < tend && gok[u] && (unsigned)iy < (unsigned)wg.H && (unsigned)ix < (unsigned)wg.W;
        src = ok ? (const void*)(X + (((long)gb[u] * wg.H + iy) * wg.W + ix) * wg.C + gch[u]) : zero;
        gox[u] += BT;                                  // the stage after this one (stages are issued in order)
        while (gox[u] >= wg.OW) {
          gox[u] -= wg.OW;
          if (++goy[u] == wg.OH) {
            goy[u] = 0;
            ++gb[u];
          }
        }
      } else {
        src = t < tlim ? (const void*)(X + t * ldx + k0 + 8 * (c ^ xswz(r))) : zero;
      }
      lds_dma16(src, xs + q * 1024);
    }
  };

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[IB];
#pragma unroll
  for (int i = 0; i < IB; ++i) bsum[i] = 0.f;
  const bool do_bias = gbias != nullptr && k0 == 0 && wk == 0;
  // SEG: result so far (finished samples, scaled); acc / bsum hold the running sample
  f32x16 tot[SEG ? IB : 1][SEG ? JB : 1];
  float btot[SEG ? IB : 1];
  if constexpr (SEG) {
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      btot[i] = 0.f;
#pragma unroll
      for (int j = 0; j < JB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
    }
  }
  long ctb = t0, cend = seg_end(t0);                   // compute cursor

  // ---- fragment addressing: lane = (16-lane group q16, p); group: column half (q16 & 1), k-slot half g = q16 >> 1
  const int g = lane >> 5, col = lane & 31, p = lane & 15, q16 = lane >> 4;
  const int frow = 8 * g + (p >> 2);                               // + 16 ks + 4 h
  const int fcol = 16 * (q16 & 1) + 4 * (p & 3);                   // + block offset; multiple of 4 -> byte offset 0 / 8 in a piece
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto gaddr = [&](int buf, int row, int c) {                      // byte address of element (row, c) of the G tile
    return lds0 + buf * STAGE + row * GROW + 16 * ((c >> 3) ^ gswz(row)) + 2 * (c & 7);
  };
  auto xaddr = [&](int buf, int row, int c) {
    return lds0 + buf * STAGE + GBYTES + row * XROW + 16 * ((c >> 3) ^ xswz(row)) + 2 * (c & 7);
  };

  issue(0, 0);
  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    wait_dma_all();
    wg_barrier();                                  // stage `it` has landed everywhere; everybody is done reading stage it - 1
    if (it + 1 < nit) issue(it + 1, buf ^ 1);
    // Transpose reads run AHEAD of the MFMAs (round 4, second half): the fragments of sub-step ks + 1 (64 x 64 tiles, one MFMA
    // per sub-step: of all the stage's sub-steps) are requested before the products of ks, and the wait counts what is allowed
    // to be in flight -- it used to be read, `lgkmcnt(0)`, MFMA, read ... : one LDS latency per sub-step and wave in a row.
    // The reads are inline asm: the compiler does not know that their results arrive asynchronously, and nothing ties the MFMAs
    // to the wait -- volatile asms keep their order, and the empty ones give the users of a register set a dependency on it.
    constexpr int NSUB = BT / 16;
    constexpr int FD = (IB * JB == 1) ? NSUB : 2;      // fragment register sets
    constexpr int NRD = 2 * (IB + JB);                 // reads per sub-step
    static_assert(NRD * (FD - 1) <= 15, "lgkmcnt is a 4-bit count");
    u32x2 a0[FD][IB], a1[FD][IB], b0[FD][JB], b1[FD][JB];
    auto read_frags = [&](int ks, int set) {
      const int row = 16 * ks + frow;
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int c = wn * (BN / 2) + i * 32 + fcol;
        a0[set][i] = lds_read_tr16(gaddr(buf, row, c));
        a1[set][i] = lds_read_tr16(gaddr(buf, row + 4, c));
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int c = wk * (BK / 2) + j * 32 + fcol;
        b0[set][j] = lds_read_tr16(xaddr(buf, row, c));
        b1[set][j] = lds_read_tr16(xaddr(buf, row + 4, c));
      }
    };
    static_for<0, FD - 1>([&](auto kc) { read_frags(decltype(kc)::value, decltype(kc)::value); });
    static_for<0, NSUB>([&](auto kc) {
      constexpr int ks = decltype(kc)::value, set = ks % FD;
      if constexpr (ks + FD - 1 < NSUB) read_frags(ks + FD - 1, (ks + FD - 1) % FD);
      constexpr int ahead = (NSUB - 1 - ks < FD - 1 ? NSUB - 1 - ks : FD - 1) * NRD;   // reads of younger sub-steps in flight
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(ahead) : "memory");
      typename E::vec8 af[IB], bf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i) asm volatile("" : "+v"(a0[set][i]), "+v"(a1[set][i]));
#pragma unroll
      for (int j = 0; j < JB; ++j) asm volatile("" : "+v"(b0[set][j]), "+v"(b1[set][j]));
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        if (do_bias) {
          float f[8];
          unpack4<DT>(a0[set][i], *(float(*)[4])&f[0]);
          unpack4<DT>(a1[set][i], *(float(*)[4])&f[4]);
          bsum[i] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        }
        af[i] = join8<DT>(a0[set][i], a1[set][i]);
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) bf[j] = join8<DT>(b0[set][j], b1[set][j]);
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = E::mma(af[i], bf[j], acc[i][j]);
    });
    if constexpr (SEG) {
      ctb += BT;
      if (ctb >= cend) {                               // the sample is complete: fold it in with its scale
        const float sc = sscale[min((long)(kMaxScales - 1), (cend - 1) / rows_per_sample - s0)];
#pragma unroll
        for (int i = 0; i < IB; ++i) {
          btot[i] = fmaf(sc, bsum[i], btot[i]);
          bsum[i] = 0.f;
#pragma unroll
          for (int j = 0; j < JB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              tot[i][j][r] = fmaf(sc, acc[i][j][r], tot[i][j][r]);
              acc[i][j][r] = 0.f;
            }
        }
        ctb = cend;
        cend = seg_end(cend);
      }
    }
  }
  if constexpr (SEG) {
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      bsum[i] = btot[i];
#pragma unroll
      for (int j = 0; j < JB; ++j) acc[i][j] = tot[i][j];
    }
  }

  float* out = accumulate ? P : P + (long)slab * N * K;
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int kk = k0 + wk * (BK / 2) + j * 32 + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * (BN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (accumulate) atomicAdd(out + (long)n * K + kk, acc[i][j][r]);
        else out[(long)n * K + kk] = acc[i][j][r];
      }
    }
  if (do_bias) {                                  // lanes (col, g = 0 / 1) hold the two k-slot halves of column col
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const float sum = half_sum(bsum[i]);
      if (g == 0) atomicAdd(gbias + n0 + wn * (BN / 2) + i * 32 + col, sum);
    }
  }
}

template <int DT, int BN, int BK, int BT, bool GATHER = false, bool SEG = false>
__global__ __launch_bounds__(256) void gemm_tn3_kernel(const uint16_t* __restrict__ G, const uint16_t* __restrict__ X,
                                                       float* __restrict__ P, int T, int N, int K, long ldg, long ldx,
                                                       int R, int tiles_k, int accumulate, float* __restrict__ gbias,
                                                       const float* __restrict__ rowscale, int rows_per_sample,
                                                       const void* zero, int xcd, WgradGeom wg = WgradGeom{}) {
  gemm_tn3_body<DT, BN, BK, BT, GATHER, SEG>(G, X, P, T, N, K, ldg, ldx, R, tiles_k, accumulate, gbias, rowscale,
                                             rows_per_sample, zero, xcd, wg, (int)blockIdx.x, (int)gridDim.x);
}

// Several weight gradients in ONE launch (round 5): the weight gradients of a MiT block's Linear layers are off the backward
// pass's dependency chain (only the data gradients feed the next layer), but as launches of their own they sit IN it -- six
// latency-bound launches of 15-25 us per block between the data-gradient GEMMs.  The host queues them (refign_amd/linear.py)
// and hands a whole block's worth over at once: workgroup b belongs to problem i with first[i] <= b < first[i + 1] and runs the
// body above on that problem's operands, 64 x 64 tiles, fp32 atomics into the parameters' views of the flat gradient buffer
// (+ the bias column sums).  The problems travel as kernel arguments (no host-to-device copy; a captured launch keeps them).
constexpr int kTnGroupMax = 8;
struct TnProblem {
  const uint16_t* G;
  const uint16_t* X;
  float* P;
  float* gbias;
  const float* rowscale;
  long ldg, ldx;
  int T, N, K, R, tiles_k, rows_per_sample, first, nblk;
};
struct TnGroup {
  TnProblem p[kTnGroupMax];
  int count;
};

template <int DT, bool SEG>
__global__ __launch_bounds__(256) void gemm_tn3_group_kernel(TnGroup grp, const void* zero, int xcd) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < kTnGroupMax; ++k)
    if (k < grp.count && (int)blockIdx.x >= grp.p[k].first) i = k;
  const TnProblem& q = grp.p[i];
  gemm_tn3_body<DT, 64, 64, 64, false, SEG>(q.G, q.X, q.P, q.T, q.N, q.K, q.ldg, q.ldx, q.R, q.tiles_k, 1, q.gbias, q.rowscale,
                                            q.rows_per_sample, zero, xcd, WgradGeom{}, (int)blockIdx.x - q.first, q.nblk);
}

__device__ uint4 g_zero_page[4];          // zero-initialised: DMA source of out-of-image / padding pieces

// gemm2.hip (its own translation unit: this file is built with MFMA results in VGPRs, gemm2.h pins its accumulators to AGPRs)
int launch_nt2(const void* X, const void* W, void* Y, long M, long N, long K, long ldx, long ldw, long ldy, const void* bias,
               const void* res, const float* rowscale, int rows_per_sample, int bn2, long t2, hipStream_t s);

template <int DT, bool GATHER>
static int launch_nt(const void* X, const void* W, void* Y, long M, long N, long K, long ldx, long ldw, long ldy,
                     const GemmEpi& epi, ConvGeom cg, hipStream_t s, bool out32 = false) {
  if (GATHER) {
    static void* zero_page = nullptr;        // looked up once (first call is an eager warm-up, never inside a capture)
    if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page)) != hipSuccess)
      return fail(RFN_ELAUNCH, "conv2d_nhwc: zero page symbol");
    cg.zero = zero_page;
  }
  // second-generation kernel (gemm2.h: software-pipelined K loop, stores dripped under the next tile's MFMAs) for the big
  // Linear layers whose column count is a multiple of 320 -- every GEMM of MiT-B5's stage 3 on the teacher's 40 views:
  // 1.2-1.5x on those shapes (profiles/r04_gemm2_v3_probe.txt); bit-identical results -- and, as 192 x 256 tiles, for the
  // other big problems with N % 256 == 0 (stages 2 / 4, the decode heads' 1 x 1 convolutions), which the 8-wave 256 x 256
  // tile of the first generation served.  
  if constexpr (!GATHER && DT == 1) {
    constexpr int g2 = 2;
    constexpr long g2_min = 200;
    // (192 x 256 from K = 256 on: at K = 128 the 8-wave tile is 3-7 % faster, at K >= 512 the new one 1.1-1.45 x --
    // profiles/r04_gemm2_256_ab.txt)
    const int bn2 = N % 320 == 0 ? 320 : (g2 >= 2 && N % 256 == 0 && K >= 256 ? 256 : 0);
    const long t2 = bn2 ? (long)cdiv(M, 192) * (N / bn2) : 0;
    if (g2 && bn2 && !out32 && K >= 192 && K % 64 == 0 && (epi.act & 255) == 0 && t2 >= g2_min &&
        (M + 192) * ldy * 2 < (1L << 32) && ldx < (1L << 22) && ldw < (1L << 22) && (((size_t)X | (size_t)W | (size_t)Y) & 15) == 0 &&
        (epi.res == nullptr || ((size_t)epi.res & 15) == 0)) {
      return launch_nt2(X, W, Y, M, N, K, ldx, ldw, ldy, epi.bias, epi.res, epi.rowscale, epi.rows_per_sample, bn2, t2, s);
    }
  }
  // tile: the largest of 128 x 128 (N % 128 == 0), 128 x 64, 64 x 64 that still gives ~1000 tiles (2-4 workgroups per CU
  // are resident; a launch of 300-600 big tiles is one and a bit rounds of a latency chain -- swept on replayed graphs in
  // round 3, profiles/r03_gemm_sweep.txt: 8160 x 320 -> 1280: 21.6 -> 18.1 us, 8160 x 1280 -> 320: 19.5 -> 17.5,
  // 2040 x 512 -> 2048: 13.7 -> 11.1).  ~2000 since the end of round 4: in the step, next to two other streams, the smaller
  // tiles of the 1000-2000 band win (profiles/r04_knob_sweep.txt, r04_knob_ab.txt).
  constexpr long min_tiles = 2000;
  int bn = (N % 128 == 0) ? 128 : 64, bm = 128;
  if ((long)cdiv(M, 128) * cdiv(N, bn) < min_tiles) {
    bn = 64;
    if ((long)cdiv(M, 128) * cdiv(N, 64) < min_tiles) bm = 64;
  }
  int ns = 2;
  // big problems whose n extent fills 256-wide tiles: 8 waves on a 256 x 256 tile (1 workgroup per CU); with a long
  // reduction (K >= 1024) already from 128 tiles on (20400 x 2048 -> 512: 74.8 -> 59.7 us)
  if (N % 256 == 0 && (long)cdiv(M, 256) * (N / 256) >= (K >= 1024 ? 128 : 256)) bm = bn = 256;
  const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn);
  const long total = (long)tiles_m * tiles_n;
  // workgroups per CU by LDS (ns-deep ring of (bm + bn) * 128 bytes; 160 KB per CU), at most 4
  const int ring = ns * (bm + bn) * 128;
  const int per_cu = std::max(1, std::min(160 * 1024 / ring, 4));
  // persistent (one pipeline across tiles) pays when a tile has only a few K-steps: the next tile's loads hide under the
  // epilogue.  With many K-steps per tile the plain one-tile-per-workgroup launch measured faster (dispatcher refills a CU
  // the moment a workgroup retires; its stores drain behind it).
  const bool persistent = K <= 256;
  const int slots = persistent ? 256 * per_cu : 0x7fffffff;
  dim3 grid((unsigned)std::min<long>(total, slots)), block(256);
#define RFN_NT(BM_, BN_, NS_)                                                                                            \
  hipLaunchKernelGGL((gemm_nt_kernel<DT, BM_, BN_, 64, NS_, GATHER>), grid, block, 0, s, (const uint16_t*)X,             \
                     (const uint16_t*)W, (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, (int)total, epi,  \
                     cg)
  if (out32) {                                  // fp32 result (split-bf16 parity mode): two tile shapes, no persistence
    bn = (N % 128 == 0) ? 128 : 64;
    bm = bn;
    const int tn = cdiv(N, bn);
    const long tot = (long)cdiv(M, bm) * tn;
    dim3 g32((unsigned)tot);
    if (bn == 128)
      hipLaunchKernelGGL((gemm_nt_kernel<DT, 128, 128, 64, 2, GATHER, 4, true>), g32, block, 0, s, (const uint16_t*)X,
                         (const uint16_t*)W, (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tn, (int)tot, epi, cg);
    else
      hipLaunchKernelGGL((gemm_nt_kernel<DT, 64, 64, 64, 2, GATHER, 4, true>), g32, block, 0, s, (const uint16_t*)X,
                         (const uint16_t*)W, (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tn, (int)tot, epi, cg);
    return check_launch("gemm_nt (fp32 result)");
  }
  const int key = bm * 10000 + bn * 10 + ns;
  switch (key) {
    case 1281282: RFN_NT(128, 128, 2); break;
    case 1280642: RFN_NT(128, 64, 2); break;
    case 641282: RFN_NT(64, 128, 2); break;
    case 640642: RFN_NT(64, 64, 2); break;
    case 2562562:
      grid = dim3((unsigned)std::min<long>(total, persistent ? 256 : 0x7fffffff));
      hipLaunchKernelGGL((gemm_nt_kernel<DT, 256, 256, 64, 2, GATHER, 8>), grid, dim3(512), 0, s, (const uint16_t*)X,
                         (const uint16_t*)W, (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, (int)total,
                         epi, cg);
      break;
    default: return fail(RFN_EINVAL, "gemm_nt: no kernel for tile %dx%d ring %d", bm, bn, ns);
  }
#undef RFN_NT
  return check_launch("gemm_nt");
}

template <int DT, bool GATHER = false>
static int launch_tn(const void* G, const void* X, float* P, long T, long N, long K, long ldg, long ldx, int R,
                     int accumulate, float* gbias, const float* rowscale, int rps, hipStream_t s,
                     WgradGeom wg = WgradGeom{}) {
  const int S = cdiv(T, R);
  dim3 block(256);
  static const int tn_xcd = 1;
  // (first-generation kernel = the fall-back for operands that are not 16-byte aligned; measured on the step: 189.2 ms with
  // it everywhere, 185.8 ms with the second generation)
  const bool vec = (GATHER ? wg.C % 8 == 0 : ldx % 8 == 0) && ((size_t)X & 15) == 0 && ldg % 8 == 0 && ((size_t)G & 15) == 0;
  static const int tn3 = 1;
  if (vec && tn3) {
    // third generation (LDS-DMA + transpose reads); the stochastic-depth scale needs whole samples per 8-row fragment and the
    // slab's scales in 64 LDS floats
    const long span = rowscale != nullptr ? ((long)R + rps - 1) / rps + 1 : 0;
    if (rowscale == nullptr || (!GATHER && span <= 64)) {
      static void* zero_page = nullptr;
      if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page)) != hipSuccess)
        return fail(RFN_ELAUNCH, "gemm_tn: zero page symbol");
      const bool big = N % 128 == 0 && K % 128 == 0;
      dim3 grid((unsigned)(big ? (N / 128) * (K / 128) * S : (N / 64) * (K / 64) * S));
#define RFN_TN3(BN_, BK_, BT_, SEG_)                                                                                      \
  hipLaunchKernelGGL((gemm_tn3_kernel<DT, BN_, BK_, BT_, GATHER && !SEG_, SEG_>), grid, block, 0, s, (const uint16_t*)G, \
                     (const uint16_t*)X, P, (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / BK_), accumulate, gbias,      \
                     rowscale, rps, zero_page, tn_xcd, wg)
      if (rowscale != nullptr) {
        if (big) RFN_TN3(128, 128, 32, true);
        else RFN_TN3(64, 64, 64, true);
      } else {
        if (big) RFN_TN3(128, 128, 32, false);
        else RFN_TN3(64, 64, 64, false);
      }
#undef RFN_TN3
      return check_launch("gemm_tn3");
    }
  }
  if (vec) {
    if (N % 128 == 0 && K % 128 == 0) {
      dim3 grid((unsigned)((N / 128) * (K / 128) * S));
      hipLaunchKernelGGL((gemm_tn2_kernel<DT, 128, 128, 32, GATHER>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X,
                         P, (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 128), accumulate, gbias, rowscale, rps, wg, tn_xcd);
    } else {
      dim3 grid((unsigned)((N / 64) * (K / 64) * S));
      hipLaunchKernelGGL((gemm_tn2_kernel<DT, 64, 64, 64, GATHER>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X,
                         P, (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 64), accumulate, gbias, rowscale, rps, wg, tn_xcd);
    }
    return check_launch("gemm_tn2");
  }
  if (N % 128 == 0 && K % 128 == 0) {
    dim3 grid((unsigned)((N / 128) * (K / 128) * S));
    hipLaunchKernelGGL((gemm_tn_kernel<DT, 128, 128, GATHER>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X, P,
                       (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 128), accumulate, gbias, rowscale, rps, wg, tn_xcd);
  } else {
    dim3 grid((unsigned)((N / 64) * (K / 64) * S));
    hipLaunchKernelGGL((gemm_tn_kernel<DT, 64, 64, GATHER>), grid, block, 0, s, (const uint16_t*)G, (const uint16_t*)X, P,
                       (int)T, (int)N, (int)K, ldg, ldx, R, (int)(K / 64), accumulate, gbias, rowscale, rps, wg, tn_xcd);
  }
  return check_launch("gemm_tn");
}

}  // namespace rfn

extern "C" {

int rfn_gemm_nt(const void* X, const void* W, const void* bias, const void* res, const float* rowscale,
                int rows_per_sample, int act, void* Y, long M, long N, long K, long ldx, long ldw, long ldy, int dtype,
                rfn_stream_t stream) {
  using namespace rfn;
#ifdef RFN_GEMM_PROFILE
  static const int prof = 0;
  const int act_arg = act;
#endif
  RFN_REQUIRE(X && W && Y, "gemm_nt: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "gemm_nt: dtype %d (1 = bf16, 2 = f16)", dtype);
  RFN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0, "gemm_nt: M=%ld N=%ld K=%ld (K %% 64, N %% 8)", M, N, K);
  RFN_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldw >= K && ldy >= N,
              "gemm_nt: leading dimensions must be multiples of 8 elements");
  RFN_REQUIRE(M < (1L << 31) && N < (1L << 31), "gemm_nt: extent");
  RFN_REQUIRE(rowscale == nullptr || rows_per_sample > 0, "gemm_nt: rowscale needs rows_per_sample");
  RFN_REQUIRE(act == 0 || act == 1 || act == 3 || (act == 4 && res != nullptr),
              "gemm_nt: act (0 none, 1 ReLU, 3 LeakyReLU 0.1, 4 = times gelu'(res), needs res)");
  GemmEpi epi{(const uint16_t*)bias, (const uint16_t*)res, rowscale, rows_per_sample > 0 ? rows_per_sample : 1, act};
#ifdef RFN_GEMM_PROFILE
  epi.act = act_arg | (prof << 8);
#endif
  hipStream_t s = (hipStream_t)stream;
  ConvGeom cg{};
  return dtype == 1 ? launch_nt<1, false>(X, W, Y, M, N, K, ldx, ldw, ldy, epi, cg, s)
                    : launch_nt<2, false>(X, W, Y, M, N, K, ldx, ldw, ldy, epi, cg, s);
}

int rfn_conv2d_nhwc(const void* X, const void* W, const void* bias, const void* res, int act, void* Y, int B, int H,
                    int Wd, int C, int N, int KH, int KW, int stride, int pad, int dil, long ldw, long ldy, int dtype,
                    rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(X && W && Y, "conv2d_nhwc: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "conv2d_nhwc: dtype %d (1 = bf16, 2 = f16)", dtype);
  RFN_REQUIRE(B > 0 && H > 0 && Wd > 0 && C > 0 && C % 8 == 0 && N > 0 && N % 8 == 0, "conv2d_nhwc: B=%d H=%d W=%d C=%d N=%d "
              "(C %% 8, N %% 8)", B, H, Wd, C, N);
  RFN_REQUIRE(KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d_nhwc: kernel %dx%d stride %d pad %d dil %d",
              KH, KW, stride, pad, dil);
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (Wd + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  RFN_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc: empty output");
  const long K = ((long)KH * KW * C + 63) / 64 * 64, M = (long)B * OH * OW;
  RFN_REQUIRE(ldw % 8 == 0 && ldw >= K && ldy % 8 == 0 && ldy >= N, "conv2d_nhwc: ldw=%ld (>= %ld, padded k) ldy=%ld", ldw,
              K, ldy);
  RFN_REQUIRE(M < (1L << 31) && K / 8 < 65536 && (long)H * Wd * C < (1L << 31), "conv2d_nhwc: extent");
  RFN_REQUIRE(act == 0 || act == 1 || act == 3, "conv2d_nhwc: act (0 none, 1 ReLU, 3 LeakyReLU 0.1)");
  GemmEpi epi{(const uint16_t*)bias, (const uint16_t*)res, nullptr, 1, act};
  ConvGeom cg{H, Wd, C, OH, OW, KH, KW, stride, pad, dil, C / 8, (unsigned)((0x100000000ULL + C / 8 - 1) / (C / 8)),
              (unsigned)((0x100000000ULL + KW - 1) / KW), nullptr, 0, 0};
  hipStream_t s = (hipStream_t)stream;
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1 && res == nullptr) {
    // halo-tiled form (conv3x3.hip): the input tile staged once per 64 channels instead of once per tap
    const int rc = launch_conv3x3_halo(X, W, bias, Y, B, H, Wd, C, N, ldw, ldy, act, dtype, 0, s);
    if (rc != 1) return rc;
  }
  return dtype == 1 ? launch_nt<1, true>(X, W, Y, M, N, K, 0, ldw, ldy, epi, cg, s)
                    : launch_nt<2, true>(X, W, Y, M, N, K, 0, ldw, ldy, epi, cg, s);
}

int rfn_gemm_nt_o32(const void* X, const void* W, const float* bias, const float* res, const float* rowscale,
                    int rows_per_sample, int act, float* Y, long M, long N, long K, long ldx, long ldw, long ldy,
                    rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(X && W && Y, "gemm_nt_o32: null operand");
  RFN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 8 == 0, "gemm_nt_o32: M=%ld N=%ld K=%ld (K %% 64, N %% 8)", M, N, K);
  RFN_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldy % 4 == 0 && ldx >= K && ldw >= K && ldy >= N, "gemm_nt_o32: leading dimensions");
  RFN_REQUIRE(M < (1L << 31) && N < (1L << 31), "gemm_nt_o32: extent");
  RFN_REQUIRE(rowscale == nullptr || rows_per_sample > 0, "gemm_nt_o32: rowscale needs rows_per_sample");
  RFN_REQUIRE(act == 0 || act == 1 || act == 3, "gemm_nt_o32: act");
  GemmEpi epi{(const uint16_t*)bias, (const uint16_t*)res, rowscale, rows_per_sample > 0 ? rows_per_sample : 1, act};
  ConvGeom cg{};
  return launch_nt<1, false>(X, W, Y, M, N, K, ldx, ldw, ldy, epi, cg, (hipStream_t)stream, true);
}

int rfn_conv2d_nhwc_o32(const void* X, const void* W, const float* bias, int act, float* Y, int B, int H, int Wd, int C, int N,
                        int KH, int KW, int stride, int pad, int dil, long ldw, long ldy, int transposed,
                        rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(X && W && Y, "conv2d_nhwc_o32: null operand");
  RFN_REQUIRE(B > 0 && H > 0 && Wd > 0 && C > 0 && C % 8 == 0 && N > 0 && N % 8 == 0, "conv2d_nhwc_o32: sizes (C %% 8, N %% 8)");
  RFN_REQUIRE(KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0 && (!transposed || (stride & (stride - 1)) == 0),
              "conv2d_nhwc_o32: geometry");
  RFN_REQUIRE(act == 0 || act == 1 || act == 3, "conv2d_nhwc_o32: act");
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (Wd + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  RFN_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc_o32: empty output");
  GemmEpi epi{(const uint16_t*)bias, nullptr, nullptr, 1, act};
  hipStream_t s = (hipStream_t)stream;
  if (!transposed) {           // forward: X (B, H, W, C) -> Y (B, OH, OW, N), W[n][(tap, c)]
    const long K = ((long)KH * KW * C + 63) / 64 * 64, M = (long)B * OH * OW;
    RFN_REQUIRE(ldw % 8 == 0 && ldw >= K && ldy % 4 == 0 && ldy >= N && M < (1L << 31) && K / 8 < 65536, "conv2d_nhwc_o32: ld");
    ConvGeom cg{H, Wd, C, OH, OW, KH, KW, stride, pad, dil, C / 8, (unsigned)((0x100000000ULL + C / 8 - 1) / (C / 8)),
                (unsigned)((0x100000000ULL + KW - 1) / KW), nullptr, 0, 0};
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && dil == 1) {       // halo-tiled form (conv3x3.hip)
      const int rc = launch_conv3x3_halo(X, W, bias, Y, B, H, Wd, C, N, ldw, ldy, act, 1, 1, s);
      if (rc != 1) return rc;
    }
    return launch_nt<1, true>(X, W, Y, M, N, K, 0, ldw, ldy, epi, cg, s, true);
  }
  // data gradient: X = grad_y (B, OH, OW, N), Y = dx (B, H, W, C), W = Wt[c][(tap, n)]
  const long K = ((long)KH * KW * N + 63) / 64 * 64, M = (long)B * H * Wd;
  RFN_REQUIRE(ldw % 8 == 0 && ldw >= K && ldy % 4 == 0 && ldy >= C && M < (1L << 31) && K / 8 < 65536, "conv2d_nhwc_o32: ld");
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  ConvGeom cg{OH, OW, N, H, Wd, KH, KW, stride, pad, dil, N / 8, (unsigned)((0x100000000ULL + N / 8 - 1) / (N / 8)),
              (unsigned)((0x100000000ULL + KW - 1) / KW), nullptr, 1, sshift};
  return launch_nt<1, true>(X, W, Y, M, C, K, 0, ldw, ldy, epi, cg, s, true);
}

int rfn_conv2d_nhwc_dgrad(const void* GY, const void* Wt, void* DX, int B, int H, int Wd, int C, int N, int KH, int KW,
                          int stride, int pad, int dil, long ldw, long ldy, int dtype, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(GY && Wt && DX, "conv2d_nhwc_dgrad: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "conv2d_nhwc_dgrad: dtype %d (1 = bf16, 2 = f16)", dtype);
  RFN_REQUIRE(B > 0 && H > 0 && Wd > 0 && C > 0 && C % 8 == 0 && N > 0 && N % 8 == 0, "conv2d_nhwc_dgrad: B=%d H=%d W=%d C=%d "
              "N=%d (C %% 8, N %% 8)", B, H, Wd, C, N);
  RFN_REQUIRE(KH > 0 && KW > 0 && stride > 0 && (stride & (stride - 1)) == 0 && dil > 0 && pad >= 0,
              "conv2d_nhwc_dgrad: kernel %dx%d stride %d (power of two) pad %d dil %d", KH, KW, stride, pad, dil);
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (Wd + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  RFN_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc_dgrad: empty output");
  // GEMM view: rows = input pixels (B, H, W), reduction = (tap, n) over grad_y's channels, result columns = C
  const long K = ((long)KH * KW * N + 63) / 64 * 64, M = (long)B * H * Wd;
  RFN_REQUIRE(ldw % 8 == 0 && ldw >= K && ldy % 8 == 0 && ldy >= C, "conv2d_nhwc_dgrad: ldw=%ld (>= %ld, padded k) ldy=%ld", ldw,
              K, ldy);
  RFN_REQUIRE(M < (1L << 31) && K / 8 < 65536 && (long)OH * OW * N < (1L << 31), "conv2d_nhwc_dgrad: extent");
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  GemmEpi epi{nullptr, nullptr, nullptr, 1, 0};
  ConvGeom cg{OH, OW, N, H, Wd, KH, KW, stride, pad, dil, N / 8, (unsigned)((0x100000000ULL + N / 8 - 1) / (N / 8)),
              (unsigned)((0x100000000ULL + KW - 1) / KW), nullptr, 1, sshift};
  hipStream_t s = (hipStream_t)stream;
  return dtype == 1 ? launch_nt<1, true>(GY, Wt, DX, M, C, K, 0, ldw, ldy, epi, cg, s)
                    : launch_nt<2, true>(GY, Wt, DX, M, C, K, 0, ldw, ldy, epi, cg, s);
}

int rfn_conv2d_nhwc_wgrad(const void* GY, const void* X, float* P, float* grad_bias, int B, int H, int Wd, int C, int N, int KH,
                          int KW, int stride, int pad, int dil, long ldg, long Kpad, int rows_per_slab, int accumulate,
                          int dtype, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(GY && X && P, "conv2d_nhwc_wgrad: null operand");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "conv2d_nhwc_wgrad: dtype %d", dtype);
  RFN_REQUIRE(B > 0 && H > 0 && Wd > 0 && C > 0 && C % 2 == 0 && N > 0 && N % 64 == 0, "conv2d_nhwc_wgrad: B=%d H=%d W=%d C=%d N=%d "
              "(C %% 2, N %% 64)", B, H, Wd, C, N);
  RFN_REQUIRE(KH > 0 && KW > 0 && stride > 0 && dil > 0 && pad >= 0, "conv2d_nhwc_wgrad: kernel %dx%d stride %d pad %d dil %d", KH,
              KW, stride, pad, dil);
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (Wd + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  RFN_REQUIRE(OH > 0 && OW > 0, "conv2d_nhwc_wgrad: empty output");
  const long T = (long)B * OH * OW;
  RFN_REQUIRE(Kpad % 64 == 0 && Kpad >= (long)KH * KW * C && ldg % 2 == 0 && ldg >= N, "conv2d_nhwc_wgrad: Kpad=%ld ldg=%ld", Kpad, ldg);
  RFN_REQUIRE(T < (1L << 31) && (long)B * H * Wd * C < (1L << 31) && rows_per_slab > 0 && rows_per_slab % 32 == 0,
              "conv2d_nhwc_wgrad: extent / rows_per_slab");
  auto magic = [](long d) { return (unsigned)((0x100000000ULL + d - 1) / d); };
  RFN_REQUIRE(Kpad < 65536, "conv2d_nhwc_wgrad: Kpad=%ld", Kpad);
  WgradGeom wg{H, Wd, C, OH, OW, KH, KW, stride, pad, dil, magic(C), magic(KW)};
  hipStream_t s = (hipStream_t)stream;
  return dtype == 1 ? launch_tn<1, true>(GY, X, P, T, N, Kpad, ldg, 0, rows_per_slab, accumulate, grad_bias, nullptr, 0, s, wg)
                    : launch_tn<2, true>(GY, X, P, T, N, Kpad, ldg, 0, rows_per_slab, accumulate, grad_bias, nullptr, 0, s, wg);
}

int rfn_gemm_tn(const void* G, const void* X, float* P, long T, long N, long K, long ldg, long ldx, int rows_per_slab,
                int accumulate, float* grad_bias, const float* rowscale, int rows_per_sample, int dtype,
                rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(G && X && P, "gemm_tn: null operand");
  RFN_REQUIRE(rowscale == nullptr || rows_per_sample > 0, "gemm_tn: rowscale needs rows_per_sample");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "gemm_tn: dtype %d", dtype);
  RFN_REQUIRE(T > 0 && T < (1L << 31) && rows_per_slab > 0 && rows_per_slab % 32 == 0,
              "gemm_tn: T=%ld rows_per_slab=%d (%% 32)", T, rows_per_slab);
  RFN_REQUIRE(N % 64 == 0 && K % 64 == 0 && ldg % 2 == 0 && ldx % 2 == 0, "gemm_tn: N=%ld K=%ld (%% 64)", N, K);
  hipStream_t s = (hipStream_t)stream;
  return dtype == 1 ? launch_tn<1>(G, X, P, T, N, K, ldg, ldx, rows_per_slab, accumulate, grad_bias, rowscale, rows_per_sample, s)
                    : launch_tn<2>(G, X, P, T, N, K, ldg, ldx, rows_per_slab, accumulate, grad_bias, rowscale, rows_per_sample, s);
}

int rfn_gemm_tn_grouped(int count, const void* const* G, const void* const* X, float* const* P, float* const* grad_bias,
                        const float* const* rowscale, const long* T, const long* N, const long* K, const long* ldg,
                        const long* ldx, const int* rows_per_slab, const int* rows_per_sample, int dtype, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(count > 0 && count <= kTnGroupMax, "gemm_tn_grouped: 1 .. %d problems (got %d)", kTnGroupMax, count);
  RFN_REQUIRE(G && X && P && grad_bias && rowscale && T && N && K && ldg && ldx && rows_per_slab && rows_per_sample,
              "gemm_tn_grouped: null array");
  RFN_REQUIRE(dtype == 1 || dtype == 2, "gemm_tn_grouped: dtype %d", dtype);
  TnGroup grp{};
  grp.count = count;
  long blocks = 0;
  const bool seg = rowscale[0] != nullptr;
  for (int i = 0; i < count; ++i) {
    RFN_REQUIRE(G[i] && X[i] && P[i], "gemm_tn_grouped: null operand in problem %d", i);
    RFN_REQUIRE((rowscale[i] != nullptr) == seg, "gemm_tn_grouped: problems with and without a row scale in one group");
    RFN_REQUIRE(T[i] > 0 && T[i] < (1L << 31) && rows_per_slab[i] > 0 && rows_per_slab[i] % 32 == 0,
                "gemm_tn_grouped: T=%ld rows_per_slab=%d (%% 32)", T[i], rows_per_slab[i]);
    RFN_REQUIRE(N[i] % 64 == 0 && K[i] % 64 == 0 && N[i] > 0 && K[i] > 0, "gemm_tn_grouped: N=%ld K=%ld (%% 64)", N[i], K[i]);
    RFN_REQUIRE(ldx[i] % 8 == 0 && ldg[i] % 8 == 0 && ((size_t)X[i] & 15) == 0 && ((size_t)G[i] & 15) == 0,
                "gemm_tn_grouped: operands must be 16-byte aligned with row pitches that are multiples of 8");
    const long S = cdiv(T[i], (long)rows_per_slab[i]);
    if (seg) {
      RFN_REQUIRE(rows_per_sample[i] > 0, "gemm_tn_grouped: rowscale needs rows_per_sample");
      const long span = ((long)rows_per_slab[i] + rows_per_sample[i] - 1) / rows_per_sample[i] + 1;
      RFN_REQUIRE(span <= 64, "gemm_tn_grouped: more than 64 samples per slab");
    }
    TnProblem& q = grp.p[i];
    q.G = (const uint16_t*)G[i];
    q.X = (const uint16_t*)X[i];
    q.P = P[i];
    q.gbias = grad_bias[i];
    q.rowscale = rowscale[i];
    q.ldg = ldg[i];
    q.ldx = ldx[i];
    q.T = (int)T[i];
    q.N = (int)N[i];
    q.K = (int)K[i];
    q.R = rows_per_slab[i];
    q.tiles_k = (int)(K[i] / 64);
    q.rows_per_sample = rows_per_sample[i];
    q.first = (int)blocks;
    q.nblk = (int)((N[i] / 64) * (K[i] / 64) * S);
    blocks += q.nblk;
    RFN_REQUIRE(blocks < (1L << 31), "gemm_tn_grouped: grid too large");
  }
  static void* zero_page = nullptr;
  if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page)) != hipSuccess)
    return fail(RFN_ELAUNCH, "gemm_tn_grouped: zero page symbol");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)blocks), block(256);
  if (dtype == 1) {
    if (seg) hipLaunchKernelGGL((gemm_tn3_group_kernel<1, true>), grid, block, 0, s, grp, (const void*)zero_page, 1);
    else hipLaunchKernelGGL((gemm_tn3_group_kernel<1, false>), grid, block, 0, s, grp, (const void*)zero_page, 1);
  } else {
    if (seg) hipLaunchKernelGGL((gemm_tn3_group_kernel<2, true>), grid, block, 0, s, grp, (const void*)zero_page, 1);
    else hipLaunchKernelGGL((gemm_tn3_group_kernel<2, false>), grid, block, 0, s, grp, (const void*)zero_page, 1);
  }
  return check_launch("gemm_tn3_group_kernel");
}

}  // extern "C"
