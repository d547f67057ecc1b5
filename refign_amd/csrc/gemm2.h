// refign_amd/csrc/gemm2.h -- second-generation NT matrix-core GEMM for the big-M Linear layers of the MiT blocks and the
// decode heads (mix_transformer.py:79-103,137-164; daformer.py:129-149):  Y[M,N] = res + rs * act(X[M,K] . W[N,K]^T + b).
//
// What is different from gemm_nt_kernel (mfma_gemm.hip), and why.  In that kernel the phases of a tile ADD -- DMA issue,
// fragment reads, MFMAs, a branchy LDS-staged epilogue (profiles/r03_gemm_ablation.txt); s_memtime traces of the first
// version of this file (profiles/r04_gemm2_v1_trace.txt) showed where a tile's time goes on the K = 320 teacher shapes:
// 35-40 % in the epilogue, whose stores run at the chip's full HBM write rate while every CU is in its epilogue at once and
// block the wave at issue; 25 % of a K-step in barrier + wait + scalar bookkeeping during which a lone wave issues no MFMA.
//   * ONE software pipeline per wave, in program order: the fragments of k-sub-step ks + 1 are read from LDS into the
//     second register set under the MFMAs of sub-step ks; the LDS-DMA instructions of K-step s + 1 are issued one at a
//     time between MFMAs of step s; one barrier per K-step, between sub-steps 2 and 3, so that the first fragments of
//     step s + 1 are in registers when it begins -- also across a tile boundary.
//   * wave tiles up to 128 x 128 (4 waves, one per SIMD, accumulators in AGPRs: 0.5 LDS fragment reads per MFMA); the
//     accumulators are never zeroed -- the first sub-step of a tile issues its MFMAs with C = 0.
//   * the epilogue only CONVERTS: bias, activation, residual, rounding, the lane exchange that makes 16-byte runs -- the
//     packed tile stays in VGPRs and its stores are issued a few at a time between the MFMAs of the NEXT tile's first
//     NSK K-steps (bounds-checked buffer stores: rows past M fall off the end of the descriptor, no exec masking).  The
//     chip then writes while it computes instead of alternating.
//   * hand-off by COUNTED s_waitcnt: buffer/global memory instructions of a wave retire in order (the compiler's own
//     waitcnt insertion relies on that for gfx9-family targets), so `vmcnt(n)` with n = the stores issued after the
//     K-step's last DMA instruction means "my DMA has landed" while those stores are still in flight.
//   * no global load in the steady state except the DMA: the bias tile rides in LDS (two small buffers, fetched by the
//     previous tile's epilogue), source addresses are a scalar base (advanced with SALU adds) plus per-lane 32-bit
//     offsets; the producer's bookkeeping is branch-free scalar code placed in front of the barrier.
// Layout conventions (LDS rows of 128 bytes = 64 k, 16-byte chunk c of row r at c ^ ((r >> 1) & 7); transposed MFMA
// blocks D[i = n][j = m]) are those of mfma_gemm.hip / mfma.h.
#pragma once
#include <type_traits>

#include "mfma.h"

namespace rfn {

typedef __attribute__((ext_vector_type(4))) int i32x4;

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. N - 1 (register arrays need literal indices)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int V> using ic = std::integral_constant<int, V>;

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
  // (s_nop 3: with the s_mov that is the five states an SALU-written base needs before a vector-memory instruction reads it)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
__device__ __forceinline__ void lds_dma4_sv(const void* sbase, unsigned voff, unsigned lds_dst) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
#pragma clang diagnostic pop
}
// raw buffer descriptor (gfx9 family): base, stride 0, num_records bytes, 32-bit data format
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  return i32x4{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
// 16-byte store at rsrc.base + voff + IMM, dropped by the hardware when voff + IMM >= num_records
template <int IMM> __device__ __forceinline__ void buf_store16(u32x4 v, unsigned voff, i32x4 rsrc) {
  // (s_nop 1: a store of more than 8 bytes reads its data registers for two more states -- the compiler pads that for its
  // own stores, not for an asm statement; without it the next instruction's write to v[0] of the data reached HBM)
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc), "n"(IMM) : "memory");
}

__device__ __forceinline__ void buf_store16_dyn(u32x4 v, unsigned voff, i32x4 rsrc) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}

template <int ACT> __device__ __forceinline__ float act2(float v) {
  if constexpr (ACT == 1) return fmaxf(v, 0.f);
  if constexpr (ACT == 3) return fmaxf(v, 0.1f * v);      // LeakyReLU(0.1)
  return v;
}

// DT 1 = bf16, 2 = f16.  BM x BN tile, WGM x WGN waves (wave tile BM / WGM x BN / WGN, both multiples of 32).
// NSK: K-steps of the next tile that carry the stores of a tile (K / 64 >= NSK).  D0, D1, D3: LDS-DMA instructions a wave
// issues in sub-steps 0, 1 and 3 (3: behind the barrier, for the step after next; sub-step 2 carries stores only).
// ABL (profiling builds of tools/micro/gemm2_probe.hip only): bit 1 no DMA, 2 no MFMA, 4 no stores.
// RES kernels (residual / per-sample scale in the epilogue) do not hold the packed tile: the registers hold the tile's
// RESIDUAL instead, requested at the top of the tile's last K-step so that its HBM latency passes under that step's
// MFMAs; their stores are issued by the epilogue (K >= 128; NSK is ignored).
// (Tiles of 256 x 256 were built and dropped: 256 accumulators + a held tile + two fragment sets do not fit the two
// 256-entry register files -- the allocator shuffles accumulators through scratch; 192 x 320 / 192 x 256 fit.)
template <int DT, int BM, int BN, int WGM, int WGN, bool BIAS, bool RES, int ACT, int NSK, int D0, int D1, int D3,
          int ABL = 0, bool TRACE = false>
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
  static_assert(D0 + D1 + D3 == NDMA, "DMA slots");
  static_assert(NW % 2 == 0, "source swizzle independent of the instruction index");
  constexpr bool ACC_IN_AGPR = NW == 4 && IB * JB * 16 > 128;
  constexpr int NTILE16 = JB * IB * 2;              // 16-byte pieces of a wave's tile: (j, i, pr) = 8 columns of one row
  constexpr bool DRIP = !RES;
  constexpr int NPEND = DRIP ? NTILE16 : 0;         // pieces that wait in registers for the next tile's K-steps
  static_assert(!DRIP || NPEND % (2 * NSK) == 0, "stores per K-step split over sub-steps 2 and 3");
  constexpr int SPK = DRIP ? NPEND / NSK : 0, SPG = SPK / 2;   // stores per carrying K-step, per sub-step
  constexpr int BIASB = ((BN * 2 + 255) / 256) * 256;   // bytes of one bias buffer
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + (BIAS ? 2 * BIASB : 0)];

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
  const int g_tm = G / tiles_n, g_tn = G % tiles_n; // the next tile of a workgroup: + G, as (tm, tn) without a division

  // ---- producer (all scalar but the per-lane offsets): step p, its k index, its tile (tm, tn); xb / wb = source bases
  // instruction q of this wave covers stage rows 8 (q NW + wave) .. + 7; lane = (row, 16-byte piece); the source piece is
  // piece ^ swizzle(row), swizzle(row) = (row >> 1) & 7 = 4 (wave & 1) | (drow >> 1): the same for every q
  const int drow = lane >> 3;
  const int spiece = (lane & 7) ^ ((4 * (wave & 1)) | (drow >> 1));
  const unsigned ldx2 = (unsigned)(ldx * 2), ldw2 = (unsigned)(ldw * 2);
  unsigned xoff[XI], woff[WI];
#pragma unroll
  for (int q = 0; q < WI; ++q) woff[q] = (unsigned)(8 * (q * NW + wave) + drow) * ldw2 + spiece * 16;
  int p = 0, p_kt = 0, p_tm = wg / tiles_n, p_tn = wg % tiles_n;
  const unsigned char* xb = (const unsigned char*)X;
  const unsigned char* wb = (const unsigned char*)W;
  // once per step, before its first piece; branch-free.  Past the last step the state freezes: the pieces re-read the
  // last step's (valid) addresses into a buffer nobody reads.
  auto producer_begin_step = [&]() {
    const bool live = p < S, newt = p_kt == 0;
    const unsigned char* xt = (const unsigned char*)(X + (long)(p_tm * BM) * ldx);
    const unsigned char* wt = (const unsigned char*)(W + (long)(p_tn * BN) * ldw);
    xb = live ? (newt ? xt : xb + 128) : xb;
    wb = live ? (newt ? wt : wb + 128) : wb;
    // bottom edge: rows past M re-read row M - 1 (never stored); frozen past the last step (p_tm is then one tile too
    // far): row 0 of the last base
    const int rmax = max(M - 1 - p_tm * BM, 0);
#pragma unroll
    for (int q = 0; q < XI; ++q) xoff[q] = (unsigned)min(8 * (q * NW + wave) + drow, rmax) * ldx2 + spiece * 16;
    const bool wrap = live && (p_kt + 1 == nk);
    p += live ? 1 : 0;
    p_kt = live ? (wrap ? 0 : p_kt + 1) : p_kt;
    const int tn2 = p_tn + g_tn;
    const bool carry = tn2 >= tiles_n;
    p_tm = wrap ? p_tm + g_tm + (carry ? 1 : 0) : p_tm;
    p_tn = wrap ? (carry ? tn2 - tiles_n : tn2) : p_tn;
  };
  int pbuf = 0;                                     // ring buffer of the producer's current step
  auto piece = [&xoff, &woff, &xb, &wb, &pbuf, lds0, wave](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int qi = q < XI ? q : q - XI;
    const unsigned dst = lds0 + pbuf * STAGE + 1024 * (qi * NW + wave) + (q < XI ? 0 : XBYTES);
    if constexpr (ABL & 1) {
      asm volatile("" ::"v"(xoff[q < XI ? qi : 0]), "v"(woff[q < XI ? 0 : qi]), "s"(dst), "s"(xb), "s"(wb));
    } else if constexpr (q < XI) {
      lds_dma16_sv(xb, xoff[qi], dst);
    } else {
      lds_dma16_sv(wb, woff[qi], dst);
    }
  };
  // bias tile of tile column tn -> bias buffer `which` (128 values per instruction; waves beyond the tile repeat)
  auto bias_fetch = [&](int tn, int which) {
    if constexpr (BIAS) {
      constexpr int NI = (BN + 127) / 128;
      const int part = wave % NI;
      const int last = BN * 2 - 4;                  // clamp the tail of a tile that is not a multiple of 128 columns
      lds_dma4_sv((const unsigned char*)(epi.bias + (long)tn * BN), (unsigned)min(part * 256 + lane * 4, last),
                  lds0 + 2 * STAGE + which * BIASB + part * 256);
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

  // ---- the packed tile waiting for its stores: entry (j, i, pr) = 8 consecutive columns of row j * 32 + frow
  u32x4 pend[DRIP ? NPEND : 1];
  u32x4 rres[RES ? NTILE16 : 1];                    // RES: the tile's residual, same indexing
  float rs[RES ? JB : 1];                           // RES: the rows' per-sample scales
  unsigned rowoff[JB];                              // byte offset of (row, first column of the wave, lane half) in Y
  const unsigned ybytes = (unsigned)((long)M * ldy * 2);
  const i32x4 y_live = make_rsrc(Y, ybytes), y_null = make_rsrc(Y, 0u);
  i32x4 y_rsrc = y_null;                            // nothing is pending during the first tile
  auto store_entry = [&pend, &rowoff, &y_rsrc](auto ec) {
    constexpr int e = decltype(ec)::value;
    constexpr int j = e / (IB * 2), i = (e / 2) % IB, pr = e % 2;
    const u32x4 v = pend[DRIP ? e : 0];
    const unsigned ro = rowoff[j];
    if constexpr (!(ABL & 4)) buf_store16<i * 64 + pr * 32>(v, ro, y_rsrc);
    else asm volatile("" ::"v"(v), "v"(ro));
  };

  f32x16 acc[IB][JB];
  // one sub-step: IB * JB MFMAs; spread between them: NDM DMA instructions (pieces Q0 ..), then NST stores (entries E0 ..)
  auto mfma_group = [&](vec8(&wf)[IB], vec8(&xf)[JB], auto firstc, auto q0c, auto ndmc, auto e0c, auto nstc) {
    constexpr bool first = decltype(firstc)::value != 0;
    constexpr int Q0 = decltype(q0c)::value, NDM = decltype(ndmc)::value, E0 = decltype(e0c)::value, NST = decltype(nstc)::value;
    constexpr int NMF = IB * JB, NSLOT = NDM + NST;
    constexpr int NSL1 = NSLOT > 0 ? NSLOT : 1;
    constexpr int PER = NSLOT > 0 ? (NMF / NSL1 > 0 ? NMF / NSL1 : 1) : NMF + 1;
    constexpr int INLOOP = NSLOT < NMF / PER ? NSLOT : NMF / PER;    // slots placed between the MFMAs
    auto slot_fn = [&](auto sc) {
      constexpr int sl = decltype(sc)::value;
      if constexpr (sl < NDM) piece(ic<Q0 + sl>{});
      else store_entry(ic<E0 + sl - NDM>{});
    };
    static_for<0, NMF>([&](auto tc) {
      constexpr int t = decltype(tc)::value, j = t / IB, i = t % IB;
      if constexpr (ABL & 2) {
        asm volatile("" ::"v"(wf[i]), "v"(xf[j]));
        if constexpr (first) acc[i][j] = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      } else if constexpr (first) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[i][j] = E::mma(wf[i], xf[j], z);
      } else {
        acc[i][j] = E::mma(wf[i], xf[j], acc[i][j]);
      }
      if constexpr ((t + 1) % PER == 0 && (t + 1) / PER - 1 < INLOOP) {
        slot_fn(ic<(t + 1) / PER - 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    });
    static_for<INLOOP, NSLOT>(slot_fn);
  };

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

  // ---- prologue: all of step 0, the D3 head of step 1, the first tile's bias
  producer_begin_step();
  pbuf = 0;
  static_for<0, NDMA>(piece);
  producer_begin_step();
  pbuf = 1;
  static_for<0, D3>(piece);
  bias_fetch(wg % tiles_n, 0);
  wait_dma_all();
  wg_barrier();
  read_frags(wf0, xf0, 0, 0);

  // one K-step (64 k) of the consumer.  first: the tile's first (accumulators start from C = 0); carry >= 0: this step
  // carries stores carry * SPK .. + SPK - 1 of the previous tile
  // residual through a bounds-checked descriptor (no residual: zero records, every load returns 0)
  const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(RES && epi.res != nullptr ? epi.res : Y), 0, RES && epi.res != nullptr ? (int)ybytes : 0, 0x00020000);
  int c_tm = wg / tiles_n, c_tn = wg % tiles_n, c_tidx = 0, s = 0;
  auto set_rowoff = [&]() {                         // of the consumer's current tile
#pragma unroll
    for (int j = 0; j < JB; ++j)
      rowoff[j] = (unsigned)(c_tm * BM + wm * WM + j * 32 + frow) * (unsigned)(ldy * 2) + (unsigned)(c_tn * BN + wn * WN + 8 * g) * 2;
  };
  auto kstep = [&](auto firstc, auto carryc, auto lastc) {
    constexpr int carry = decltype(carryc)::value;
    constexpr int E0 = carry >= 0 ? carry * SPK : 0, NST = carry >= 0 ? SPG : 0;
    const int buf = s & 1;
    stamp();
    if constexpr (RES && decltype(lastc)::value != 0) {
      set_rowoff();
#pragma unroll
      for (int j = 0; j < JB; ++j)
        rs[j] = epi.rowscale != nullptr ? epi.rowscale[min(c_tm * BM + wm * WM + j * 32 + frow, M - 1) / epi.rows_per_sample] : 1.f;
      static_for<0, NTILE16>([&](auto ec) {
        constexpr int e = decltype(ec)::value, j = e / (IB * 2), i = (e / 2) % IB, pr = e % 2;
        rres[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, (int)(rowoff[j] + (unsigned)(i * 64 + pr * 32)), 0, 0));
      });
      if constexpr (ABL & 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    read_frags(wf1, xf1, buf, 1);
    mfma_group(wf0, xf0, firstc, ic<D3>{}, ic<D0>{}, ic<0>{}, ic<0>{});   // sub-steps 0, 1: the rest of step s + 1 streams in
    stamp();
    read_frags(wf0, xf0, buf, 2);
    mfma_group(wf1, xf1, ic<0>{}, ic<D3 + D0>{}, ic<D1>{}, ic<0>{}, ic<0>{});
    stamp();
    read_frags(wf1, xf1, buf, 3);
    producer_begin_step();                          // bookkeeping of step s + 2 (scalar; its pieces wait for the barrier)
    mfma_group(wf0, xf0, ic<0>{}, ic<0>{}, ic<0>{}, ic<E0>{}, ic<NST>{});   // sub-step 2: stores only -- they may stay in flight
    stamp();
    // hand-off: step s + 1 has landed (every wave's pieces: the DMA instructions are older than the NST stores, and
    // buffer / global instructions retire in order), nobody reads stage `buf` any more
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NST) : "memory");
    stamp();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp();
    read_frags(wf0, xf0, buf ^ 1, 0);               // (past the last step: whatever the other buffer holds, unused)
    pbuf = buf;                                     // step s + 2 goes into the buffer that just became free
    stamp();
    mfma_group(wf1, xf1, ic<0>{}, ic<0>{}, ic<D3>{}, ic<E0 + NST>{}, ic<NST>{});
    stamp();
    ++s;
  };

  for (int t = 0; t < ntiles; ++t) {
    if constexpr (DRIP) {
      kstep(ic<1>{}, ic<0>{}, ic<0>{});
      if constexpr (NSK >= 2) kstep(ic<0>{}, ic<1>{}, ic<0>{});
      if constexpr (NSK >= 3) kstep(ic<0>{}, ic<2>{}, ic<0>{});
      if constexpr (NSK >= 4) kstep(ic<0>{}, ic<3>{}, ic<0>{});
      static_assert(NSK <= 4, "at most 4 store-carrying K-steps");
      for (int kt = NSK; kt < nk; ++kt) kstep(ic<0>{}, ic<-1>{}, ic<0>{});
    } else {
      kstep(ic<1>{}, ic<-1>{}, ic<0>{});
      for (int kt = 1; kt + 1 < nk; ++kt) kstep(ic<0>{}, ic<-1>{}, ic<0>{});
      kstep(ic<0>{}, ic<-1>{}, ic<1>{});
    }
    {
      // ---- epilogue of tile (c_tm, c_tn): accumulators -> pend[], rowoff[]; fetch the next tile's bias
      const unsigned char* bl = smem + 2 * STAGE + (c_tidx & 1) * BIASB;
      if constexpr (ACC_IN_AGPR) {
        // the accumulators are read with inline asm below, which the compiler's hazard recogniser cannot see into: an
        // MFMA's result needs up to 18 wait states before a VALU may read it (found the hard way: rows 12-15 / 28-31 of
        // the blocks the last MFMAs wrote came back stale)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (DRIP) set_rowoff();             // (RES: set before the residual was requested)
      // column blocks outermost: a block's bias (lane g: columns 16 pr + 4 g .. + 3 and + 8 .. + 11) is 8 registers
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        u32x2 bva[2], bvb[2];
        if constexpr (BIAS) {
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int cl = wn * WN + i * 32 + 16 * pr + 4 * g;
            bva[pr] = *(const u32x2*)(bl + 2 * cl);
            bvb[pr] = *(const u32x2*)(bl + 2 * cl + 16);
          }
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
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
              unpack4<DT>(bva[pr], ba);
              unpack4<DT>(bvb[pr], bb);
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
            const int ent = (j * IB + i) * 2 + pr;  // after the swap: 8 consecutive columns from i * 32 + 16 pr + 8 g
            if constexpr (RES) {
              // as gemm_nt_kernel does it (bit-identical results): round act(acc + bias) to 16 bits, exchange, then add the
              // residual in fp32 and round again
              const u32x2 pa = pack4<DT>(a[0], a[1], a[2], a[3]), pb = pack4<DT>(b[0], b[1], b[2], b[3]);
              auto r0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
              auto r1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
              float lo[4], hi[4];
              unpack4<DT>(u32x2{r0[0], r1[0]}, lo);
              unpack4<DT>(u32x2{r0[1], r1[1]}, hi);
              const u32x4 rr = rres[RES ? ent : 0];
              float rl[4], rh[4];
              unpack4<DT>(u32x2{rr[0], rr[1]}, rl);
              unpack4<DT>(u32x2{rr[2], rr[3]}, rh);
              const float r_s = rs[RES ? j : 0];
              const u32x2 o0 = pack4<DT>(rl[0] + r_s * lo[0], rl[1] + r_s * lo[1], rl[2] + r_s * lo[2], rl[3] + r_s * lo[3]);
              const u32x2 o1 = pack4<DT>(rh[0] + r_s * hi[0], rh[1] + r_s * hi[1], rh[2] + r_s * hi[2], rh[3] + r_s * hi[3]);
              const u32x4 o = {o0[0], o0[1], o1[0], o1[1]};
              if constexpr (!(ABL & 4)) buf_store16_dyn(o, rowoff[j] + (unsigned)(i * 64 + pr * 32), y_live);
              else asm volatile("" ::"v"(o));
            } else {
              const u32x2 pa = pack4<DT>(a[0], a[1], a[2], a[3]), pb = pack4<DT>(b[0], b[1], b[2], b[3]);
              auto r0 = __builtin_amdgcn_permlane32_swap(pa[0], pb[0], false, false);
              auto r1 = __builtin_amdgcn_permlane32_swap(pa[1], pb[1], false, false);
              pend[DRIP ? ent : 0] = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
          }
          __builtin_amdgcn_sched_barrier(0);        // one 32 x 32 block at a time: bounds the epilogue's live registers
        }
      }
      // the next tile of this workgroup
      {
        const int tn2 = c_tn + g_tn;
        const bool carry = tn2 >= tiles_n;
        c_tm += g_tm + (carry ? 1 : 0);
        c_tn = carry ? tn2 - tiles_n : tn2;
        ++c_tidx;
      }
      if constexpr (BIAS) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this tile's bias has been read before its buffer's twin is refilled
        bias_fetch(min(c_tn, tiles_n - 1), c_tidx & 1);
      }
      // the next tile's first fragments, re-read here (they were requested behind the last barrier, but holding them
      // across the epilogue costs 32 registers that the packed tile needs)
      read_frags(wf0, xf0, s & 1, 0);
      y_rsrc = y_live;
      stamp();
    }
  }
  // the last tile's stores
  if constexpr (DRIP) static_for<0, NPEND>(store_entry);
  wait_dma_all();
}

}  // namespace rfn
