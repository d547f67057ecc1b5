// refign_amd/csrc/gemm2.h -- second-generation NT matrix-core GEMM for the big-M Linear layers of the MiT blocks and the
// decode heads (mix_transformer.py:79-103,137-164; daformer.py:129-149):  Y[M,N] = res + rs * act(X[M,K] . W[N,K]^T + b).
//
// What is different from gemm_nt_kernel (mfma_gemm.hip), and why (profiles/r03_gemm_ablation.txt: in that kernel the
// phases of a tile ADD -- DMA issue, fragment reads, MFMAs, a branchy LDS-staged epilogue -- nothing overlaps):
//   * ONE software pipeline per wave, written out by hand in program order: the fragments of k-sub-step ks + 1 are read
//     from LDS into the second register set under the MFMAs of sub-step ks; the LDS-DMA instructions of K-step s + 1 are
//     issued one at a time between MFMA groups of step s (slots D0..D3 per sub-step, X pieces first: they come from
//     HBM, the weight pieces are L2 hits); one barrier per K-step, placed between sub-steps 2 and 3, so that the first
//     fragments of step s + 1 are already in registers when step s + 1 begins -- also across a tile boundary: the next
//     tile's first fragments and its first stage are in flight under the epilogue.
//   * wave tiles of 128 x 128 (4 waves, one per SIMD, 256 accumulators in AGPRs: 0.5 LDS fragment reads per MFMA) or
//     128 x 64 (8 waves), selected by the template arguments; the accumulators are never zeroed -- the first sub-step of
//     a tile issues its MFMAs with C = 0.
//   * no global load in the steady state except the DMA: the bias tile rides in the ring (4 small LDS buffers), source
//     addresses are a scalar base (advanced per K-step with two SALU adds) plus a per-lane 32-bit offset that is fixed
//     per tile -- no vector address arithmetic per DMA.
//   * a branch-free epilogue, specialised at compile time on (bias, residual, activation): registers -> 16-byte stores
//     (two lanes' 4-column runs joined by v_permlane32_swap), rows masked at the bottom edge only.
// Layout conventions (LDS rows of 128 bytes = 64 k, 16-byte chunk c of row r at c ^ ((r >> 1) & 7); transposed MFMA
// blocks D[i = n][j = m]) are those of mfma_gemm.hip / mfma.h.
#pragma once
#include "mfma.h"

namespace rfn {

struct Gemm2Epi {
  const uint16_t* bias;     // [N] or null (BIAS kernels only)
  const uint16_t* res;      // [M, ldy] or null (RES kernels only)
  const float* rowscale;    // per-sample scale of act(acc + bias) before the residual add, or null (RES kernels only)
  int rows_per_sample;
  unsigned long long* trace;   // TRACE builds (tools/micro/gemm2_probe.hip): s_memtime stamps of workgroup 0, wave 0
};

// LDS-DMA, scalar base + per-lane 32-bit byte offset: global_load_lds_dwordx4 voff, s[base:base+1]
__device__ __forceinline__ void lds_dma16_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void lds_dma4_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}

template <int ACT> __device__ __forceinline__ float act2(float v) {
  if constexpr (ACT == 1) return fmaxf(v, 0.f);
  if constexpr (ACT == 3) return fmaxf(v, 0.1f * v);      // LeakyReLU(0.1)
  return v;
}

// DT 1 = bf16, 2 = f16.  BM x BN tile, WGM x WGN waves (wave tile BM / WGM x BN / WGN, both multiples of 32).
// D0..D3: LDS-DMA instructions a wave issues in sub-steps 0..3 (D3: after the barrier, for the step after next).
// ABL (profiling builds of tools/micro/gemm2_probe.hip only): bit 1 no DMA, 2 no MFMA, 4 no stores.
template <int DT, int BM, int BN, int WGM, int WGN, bool BIAS, bool RES, int ACT, int D0, int D1, int D2, int D3, int ABL = 0, bool TRACE = false>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_nt2_kernel(const uint16_t* __restrict__ X,
                                                                const uint16_t* __restrict__ W, uint16_t* __restrict__ Y,
                                                                int M, int N, int K, long ldx, long ldw, long ldy,
                                                                int tiles_n, int total_tiles, Gemm2Epi epi) {
  using E = Elem<DT>;
  using vec8 = typename E::vec8;
  constexpr int NW = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int JB = WM / 32, IB = WN / 32;
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile in 32 x 32 blocks");
  constexpr int XBYTES = BM * 128, WBYTES = BN * 128, STAGE = XBYTES + WBYTES;
  constexpr int XI = BM / (8 * NW), WI = BN / (8 * NW), NDMA = XI + WI;   // DMA instructions per wave and K-step
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "tile rows / DMA rows");
  static_assert(D0 + D1 + D2 + D3 == NDMA, "DMA slots");
  static_assert(NW % 2 == 0, "source swizzle independent of the instruction index");
  constexpr bool ACC_IN_AGPR = NW == 4 && IB * JB * 16 > 128;
  constexpr int BIASB = BN * 2;                     // bytes of one bias tile
  constexpr int NBIAS = 4;                          // bias buffers (the producer runs up to two tiles ahead)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + (BIAS ? NBIAS * BIASB : 0)];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  if (wg >= total_tiles) return;
  const int ntiles = (total_tiles - wg + G - 1) / G;
  const int nk = K >> 6;
  const int S = ntiles * nk;                        // K-steps of this workgroup, all tiles
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- producer: per-lane source offsets (bytes, relative to the tile's scalar base), fixed per tile
  // instruction q of this wave covers stage rows 8 (q NW + wave) .. + 7; lane = (row, 16-byte piece); the source piece is
  // piece ^ swizzle(row), swizzle(row) = (row >> 1) & 7 = 4 (wave & 1) | (drow >> 1): the same for every q
  const int drow = lane >> 3;
  const int spiece = (lane & 7) ^ ((4 * (wave & 1)) | (drow >> 1));
  unsigned xoff[XI], woff[WI];
#pragma unroll
  for (int q = 0; q < WI; ++q) woff[q] = (unsigned)(8 * (q * NW + wave) + drow) * (unsigned)(ldw * 2) + spiece * 16;
  // producer cursor (all scalar): step p, its k index, its tile as (tm, tn) advanced by G tiles without a division, the
  // tile's ordinal; xb / wb = scalar source bases of the current step.  Past the last step the producer re-issues the
  // last step's addresses into a free buffer (nobody reads it): the K-step body stays free of branches.
  const int g_tm = G / tiles_n, g_tn = G % tiles_n;
  int p = 0, p_kt = 0, p_tm = wg / tiles_n, p_tn = wg % tiles_n, p_tidx = 0;
  const unsigned char* xb = (const unsigned char*)X;
  const unsigned char* wb = (const unsigned char*)W;
  auto producer_begin_step = [&]() {                // once per step, before its first piece
    const bool live = p < S;
    if (live && p_kt == 0) {
      const int m0 = p_tm * BM, n0 = p_tn * BN;
      xb = (const unsigned char*)(X + (long)m0 * ldx);
      wb = (const unsigned char*)(W + (long)n0 * ldw);
      const int rmax = M - 1 - m0;                  // bottom edge: rows past M re-read row M - 1 (never stored)
#pragma unroll
      for (int q = 0; q < XI; ++q)
        xoff[q] = (unsigned)min(8 * (q * NW + wave) + drow, rmax) * (unsigned)(ldx * 2) + spiece * 16;
      if constexpr (BIAS) {
        if (wave * 128 < BN) {                      // 128 bias values (256 bytes) per instruction
          lds_dma4_sv((const unsigned char*)(epi.bias + n0), (unsigned)(wave * 256 + lane * 4),
                      lds0 + 2 * STAGE + (p_tidx & (NBIAS - 1)) * BIASB + wave * 256);
        }
      }
    } else if (live) {
      xb += 128;
      wb += 128;
    }
    if (live) {
      ++p;
      if (++p_kt == nk) {
        p_kt = 0;
        ++p_tidx;
        p_tm += g_tm;
        p_tn += g_tn;
        if (p_tn >= tiles_n) {
          p_tn -= tiles_n;
          ++p_tm;
        }
      }
    }
  };
  // piece q (0 .. NDMA - 1) of the producer's current step into ring buffer pbuf
  int pbuf = 0;
  auto piece = [&](int q) {
    const unsigned dst = lds0 + pbuf * STAGE + 1024 * ((q < XI ? q : q - XI) * NW + wave) + (q < XI ? 0 : XBYTES);
    if constexpr (ABL & 1) {
      asm volatile("" ::"v"(xoff[q < XI ? q : 0]), "v"(woff[q < XI ? 0 : q - XI]), "s"(dst), "s"(xb), "s"(wb));
    } else {
      if (q < XI) lds_dma16_sv(xb, xoff[q < XI ? q : 0], dst);
      else lds_dma16_sv(wb, woff[q < XI ? 0 : q - XI], dst);
    }
  };

  // ---- consumer: fragment addresses.  Row (l & 31) of a 32-row block, chunk (2 ks + g) ^ swizzle(row) = (2 ks) ^ (g ^ swz)
  const int g = lane >> 5, frow = lane & 31, swz = (frow >> 1) & 7;
  const unsigned xrd = (unsigned)((wm * WM + frow) * 128 + ((g ^ swz) << 4));
  const unsigned wrd = (unsigned)(XBYTES + (wn * WN + frow) * 128 + ((g ^ swz) << 4));
  vec8 wf0[IB], xf0[JB], wf1[IB], xf1[JB];
  auto read_frags = [&](vec8(&wf)[IB], vec8(&xf)[JB], int buf, int ks) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < IB; ++i) wf[i] = *(const vec8*)(st + ((wrd ^ (ks << 5)) + i * 4096));
#pragma unroll
    for (int j = 0; j < JB; ++j) xf[j] = *(const vec8*)(st + ((xrd ^ (ks << 5)) + j * 4096));
  };

  f32x16 acc[IB][JB];
  // one sub-step: IB * JB MFMAs; after every `per` of them one DMA instruction of the producer (pieces q_begin ..)
  auto mfma_group = [&](vec8(&wf)[IB], vec8(&xf)[JB], bool first, int q_begin, int q_count) {
    constexpr int NMF = IB * JB;
    const int per = q_count > 0 ? NMF / q_count : NMF;
    int issued = 0, q = q_begin;
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        if constexpr (ABL & 2) {
          asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
          if (first) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        } else if (first) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[i][j] = E::mma(wf[i], xf[j], z);
        } else {
          acc[i][j] = E::mma(wf[i], xf[j], acc[i][j]);
        }
        ++issued;
        if (q_count > 0 && issued % per == 0 && q < q_begin + q_count) {
          piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
    for (int qq = 0; qq < NDMA; ++qq)
      if (qq >= q && qq < q_begin + q_count) piece(qq);
  };

  // ---- prologue: all of step 0, the D3 head of step 1
  producer_begin_step();
  pbuf = 0;
#pragma unroll
  for (int q = 0; q < NDMA; ++q) piece(q);
  producer_begin_step();
  pbuf = 1;
#pragma unroll
  for (int q = 0; q < D3; ++q) piece(q);
  wait_dma_all();
  wg_barrier();
  read_frags(wf0, xf0, 0, 0);

  int c_tile = wg, c_tidx = 0, s = 0;
  int tcount = 0;
  auto stamp = [&]() {
    if constexpr (TRACE) {
      if (blockIdx.x == 0 && wave == 0 && tcount < 512) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (lane == 0) epi.trace[tcount] = t;
        ++tcount;
      }
    }
  };
  // one K-step (64 k) of the consumer; FIRST: the tile's first (accumulators start from C = 0)
  auto kstep = [&](bool first) {
    const int buf = s & 1;
    stamp();
    read_frags(wf1, xf1, buf, 1);
    mfma_group(wf0, xf0, first, D3, D0);            // sub-steps 0..2: the rest of step s + 1 streams in
    stamp();
    read_frags(wf0, xf0, buf, 2);
    mfma_group(wf1, xf1, false, D3 + D0, D1);
    stamp();
    read_frags(wf1, xf1, buf, 3);
    mfma_group(wf0, xf0, false, D3 + D0 + D1, D2);
    stamp();
    // hand-off: step s + 1 has landed (every wave's pieces), nobody reads stage `buf` any more
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    stamp();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();
    read_frags(wf0, xf0, buf ^ 1, 0);               // (past the last step: whatever the other buffer holds, unused)
    producer_begin_step();                          // step s + 2 into the buffer that just became free
    pbuf = buf;
    stamp();
    mfma_group(wf1, xf1, false, 0, D3);
    stamp();
    ++s;
  };

  for (int t = 0; t < ntiles; ++t) {
    kstep(true);
    for (int kt = 1; kt < nk; ++kt) kstep(false);
    {
      // ---- epilogue of tile c_tile
      const int m0 = (c_tile / tiles_n) * BM, n0 = (c_tile % tiles_n) * BN;
      // the wave's bias columns, once per tile: lane (g) holds columns 16 pr + 4 g .. + 3 and + 8 .. + 11 of block i
      u32x2 bva[BIAS ? IB : 1][2], bvb[BIAS ? IB : 1][2];
      if constexpr (BIAS) {
        const unsigned char* bl = smem + 2 * STAGE + (c_tidx & (NBIAS - 1)) * BIASB;
#pragma unroll
        for (int i = 0; i < IB; ++i)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int cl = wn * WN + i * 32 + 16 * pr + 4 * g;
            bva[i][pr] = *(const u32x2*)(bl + 2 * cl);
            bvb[i][pr] = *(const u32x2*)(bl + 2 * cl + 16);
          }
      }
#pragma unroll
      for (int j = 0; j < JB; ++j) {
        const int m = m0 + wm * WM + j * 32 + frow;
        const bool row_ok = m < M;
        float rs = 1.f;
        if constexpr (RES) {
          if (epi.rowscale != nullptr) rs = epi.rowscale[min(m, M - 1) / epi.rows_per_sample];
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            // this lane: rows k = 2 pr (columns cl .. cl + 3) and k = 2 pr + 1 (cl + 8 .. cl + 11), cl = 16 pr + 4 g
            float a[4], b[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if constexpr (ACC_IN_AGPR) {
                // 1 wave per SIMD: the accumulators live in AGPRs; read them HERE, one block at a time (left to itself
                // the compiler copies all 256 to VGPRs at the top of the epilogue and spills the loop's registers)
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(a[e]) : "a"(acc[i][j][8 * pr + e]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(b[e]) : "a"(acc[i][j][8 * pr + 4 + e]));
              } else {
                a[e] = acc[i][j][8 * pr + e];
                b[e] = acc[i][j][8 * pr + 4 + e];
              }
            }
            if constexpr (BIAS) {
              float ba[4], bb[4];
              unpack4<DT>(bva[i][pr], ba);
              unpack4<DT>(bvb[i][pr], bb);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                a[e] += ba[e];
                b[e] += bb[e];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a[e] = act2<ACT>(a[e]);
              b[e] = act2<ACT>(b[e]);
            }
            const int n_out = n0 + wn * WN + i * 32 + 16 * pr + 8 * g;   // after the swap: 8 consecutive columns
            if constexpr (RES) {
              // fp32 values change lanes: lower half keeps k = 2 pr and receives the upper half's k = 2 pr (columns
              // + 4 .. + 7); the upper half receives the lower half's k = 2 pr + 1 and keeps its own
              float lo[4], hi[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[e]), __float_as_uint(b[e]), false, false);
                lo[e] = __uint_as_float(r[0]);
                hi[e] = __uint_as_float(r[1]);
              }
              if (row_ok) {
                float rl[4] = {0.f, 0.f, 0.f, 0.f}, rh[4] = {0.f, 0.f, 0.f, 0.f};
                if (epi.res != nullptr) {
                  const u32x4 rr = *(const u32x4*)(epi.res + (long)m * ldy + n_out);
                  unpack4<DT>(u32x2{rr[0], rr[1]}, rl);
                  unpack4<DT>(u32x2{rr[2], rr[3]}, rh);
                }
                const u32x2 o0 = pack4<DT>(rl[0] + rs * lo[0], rl[1] + rs * lo[1], rl[2] + rs * lo[2], rl[3] + rs * lo[3]);
                const u32x2 o1 = pack4<DT>(rh[0] + rs * hi[0], rh[1] + rs * hi[1], rh[2] + rs * hi[2], rh[3] + rs * hi[3]);
                const u32x4 o = {o0[0], o0[1], o1[0], o1[1]};
                if constexpr (ABL & 4) asm volatile("" ::"v"(o));
                else *(u32x4*)(Y + (long)m * ldy + n_out) = o;
              }
            } else {
              const u32x2 pa = pack4<DT>(a[0], a[1], a[2], a[3]), pb = pack4<DT>(b[0], b[1], b[2], b[3]);
              auto r0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
              auto r1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
              const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
              if constexpr (ABL & 4) asm volatile("" ::"v"(o));
              else if (row_ok) *(u32x4*)(Y + (long)m * ldy + n_out) = o;
            }
          }
          __builtin_amdgcn_sched_barrier(0);        // one 32 x 32 block at a time: bounds the epilogue's live registers
        }
      }
      c_tile += G;
      ++c_tidx;
      stamp();
    }
  }
  wait_dma_all();
}

}  // namespace rfn
