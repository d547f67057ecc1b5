// refign_amd/csrc/corr_mfma.hip -- patch-9 spatial correlation on the fp32 matrix pipe (gfx950 / MI355X).
//
// Same numbers as corr9_dma_kernel of corr.hip (reference: models/correlation_ops/correlation.cpp:13-42 for kernel 1,
// patch 9, stride 1, pad 0 -- the one parameterisation models/modules.py:268-270 uses; optional fused ReLU + L2
// normalisation over the 81 shifts, modules.py:272-273), but the products run on `v_mfma_f32_4x4x1_16B_f32` instead
// of packed VALU FMAs.  Why that instruction: it is 16 independent 4x4 outer products per issue, K = 1.  With
//     A block = 4 consecutive SOURCE pixels of one row,  B block = 4 consecutive TARGET pixels of one row
// a block computes the 16 pair products of 4 targets x 4 sources; the three source groups at -4, 0, +4 pixels cover
// the 12-pixel window the 4 targets need, so 36 of 48 products (75 %) are wanted ones -- against 14-28 % for the
// 32x32 / 16x16 fp32 shapes, whose square tiles fit a 9-wide band badly.  The rate is the same 256 flop / CU / clock as
// packed FMAs, so the arithmetic ceiling drops by a quarter; what is gained is operand delivery: an MFMA takes ONE
// dword per lane for each operand and every operand dword is reused by 9 (target) or up to 2 x 3 (source) MFMAs, i.e.
// 32 ds_read_b32 per 54 MFMAs per channel and wave = 0.15 LDS dwords per wanted product against 0.37 for the
// register-tiled VALU kernel, which was LDS-read-bound (960 LDS clocks against 648 VALU clocks per channel step).
// K = 1 also keeps the reference's summation order: one product is added to the accumulator per channel.
//
// Decomposition: a workgroup = two independent 16 x 32 tiles (as in corr9_dma_kernel: 510 tiles = 255 workgroups = one
// even round over 256 CUs at K4 level 1), 4 waves per tile, 2 waves per SIMD.  A wave owns 4 target rows x 32 pixels
// and ALL 81 shifts: lanes 0-31 are rows r0, r0+1 (MFMA p = 0, 1), lanes 32-63 rows r0+2, r0+3; within a half, lane
// 4 s + j is pixel 4 s + j.  Per channel: 2 target operands, 10 source rows x 3 groups = 30 source operands,
// 2 x 9 x 3 = 54 MFMAs into 216 accumulator registers.  All 81 values of a target pixel end up in ONE lane (the lane
// is the target, the register index the source), so the L2 norm needs no cross-lane traffic at all.
// The accumulator D_g[r] of lane j holds dx = 4 (g - 1) + r - j: a per-lane rotation of each 4-register group by j
// (two v_cndmask stages) lines the registers up with the output planes, after which every store instruction writes one
// plane and 2 x 128 contiguous bytes.
// Tiles arrive by LDS-DMA (global -> LDS, 16 B per lane) into a D-deep ring of CC-channel chunks: the DMA of chunk
// k + D - 1 is issued while chunk k is consumed, the hand-off is a counted `s_waitcnt vmcnt(N)` + a raw `s_barrier`.
// LDS rows have a 48-dword pitch so that the two halves of a ds_read_b32 (rows r and r + 2) fall into different bank
// halves.  Out-of-image and padding slots are loaded from a zero slot, so every wave issues the same number of DMA
// instructions for every chunk (which the counted waits need) and the ring is never initialised.
#include "common.h"

namespace rfn {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTH = 16, kTW = 32, kHalo = 4;
constexpr int kR2 = kTH + 2 * kHalo;       // source rows per tile
constexpr int kTilesPerWg = 2;

__device__ __attribute__((aligned(16))) float g_zero_slot[4];   // what out-of-image and padding slots are loaded from

// LDS-DMA with a per-lane 64-bit address (invalid lanes point at g_zero_slot)
__device__ __forceinline__ void lds_dma16_v(const float* gsrc, float* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  const unsigned m0v = __builtin_amdgcn_readfirstlane(base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "memory", "m0");
#pragma clang diagnostic pop
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// CC channels per chunk, D ring stages: the DMA of chunk k + D - 1 is issued while chunk k is consumed.
// P = target rows per half-wave: a wave owns 2 P rows x 32 pixels x 81 shifts in 108 P accumulator registers.
//   P = 2: 4 waves per tile, 2 waves per SIMD (<= 256 registers): 216 accumulators, the rest is operand look-ahead.
//   (P = 4, one 512-register wave per SIMD, does not work with this compiler: above 256 registers it selects the
//   AGPR form of the MFMAs for the whole kernel, and 432 accumulators do not fit 256 AGPRs -- ~1000 spills.)
// SP / TP: LDS row pitch of the source / target rows, chosen so that the two halves of a ds_read_b32 (rows r and r + P)
// fall into different bank halves (P * pitch = 32 mod 64).
template <bool FUSE, int CC, int D, int P>
__global__ __launch_bounds__(64 * (kTH / (2 * P)) * kTilesPerWg) void corr9_mfma_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, int C, int H, int W,
    int tilesX, int tilesY, int ntiles, int ablate, long long* __restrict__ trace) {
  constexpr int WPT = kTH / (2 * P);                 // waves per tile
  constexpr int SP = (P == 2) ? 48 : 40, TP = (P == 2) ? 48 : 40;
  static_assert((P * SP) % 64 == 32 && (P * TP) % 64 == 32, "bank halves");
  constexpr int SV = SP / 4, TV = TP / 4;            // 16-byte slots per LDS row
  constexpr int SRC_CH = kR2 * SP, TGT_CH = kTH * TP; // floats per channel
  constexpr int SRC_FLOATS = CC * SRC_CH, TGT_FLOATS = CC * TGT_CH;
  static_assert(SRC_FLOATS % 256 == 0 && TGT_FLOATS % 256 == 0, "whole DMA instructions per section");
  constexpr int SRC_INSTR = SRC_FLOATS / 256;        // wave-level DMA instructions (64 lanes x 4 floats)
  constexpr int TGT_INSTR = TGT_FLOATS / 256;
  constexpr int NINSTR = SRC_INSTR + TGT_INSTR;
  constexpr int K = (NINSTR + WPT - 1) / WPT;        // per wave; the last waves issue K - 1
  constexpr int KFULL = NINSTR - (K - 1) * WPT;      // waves [0, KFULL) issue K instructions
  constexpr int BUF = NINSTR * 256;                  // floats per tile and ring stage
  constexpr int NT = 64 * WPT;
  constexpr int NX = P + 8;                          // source rows a half-wave touches per channel
  static_assert((D - 1) * K < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) float ring[D * kTilesPerWg * BUF];

  const int half = __builtin_amdgcn_readfirstlane(threadIdx.x / NT);   // which tile of the workgroup (wave-uniform)
  const int tid = threadIdx.x % NT;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* const myring = ring + half * BUF;           // stage st at + st * kTilesPerWg * BUF

  const int tile = blockIdx.x * kTilesPerWg + half;
  const bool live = tile < ntiles;
  int bid = live ? tile : 0;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int h0 = ty * kTH, w0 = tx * kTW;
  const size_t plane = (size_t)H * W;
  const float* base1 = in1 + (size_t)n * C * plane;  // target features of this image
  const float* base2 = in2 + (size_t)n * C * plane;  // source features

  // DMA descriptors: address (chunk 0) of each of my K slots; slots outside the image and padding slots read zeros
  // from g_zero_slot with stride 0, so every wave issues the same number of instructions for every chunk (the counted
  // waits below rely on that) and the ring needs no initialisation
  const float* gsrc[K];
  unsigned gstride = 0;                              // bit k: slot k advances with the chunks
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int wi = wave + k * WPT;
    bool ok = live;
    const float* src = g_zero_slot;
    if (wi < SRC_INSTR) {
      const int slot = wi * 64 + lane;
      const int v = slot % SV, rr = (slot / SV) % kR2, c = slot / (SV * kR2);
      const int gy = h0 - kHalo + rr, gx = w0 - kHalo + 4 * v;
      ok = ok && v < (kTW + 2 * kHalo) / 4 && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
      if (ok) src = base2 + (size_t)c * plane + (long)gy * W + gx;
    } else {
      const int slot = (wi - SRC_INSTR) * 64 + lane;
      const int v = slot % TV, rr = (slot / TV) % kTH, c = slot / (TV * kTH);
      const int gy = h0 + rr, gx = w0 + 4 * v;
      ok = ok && v < kTW / 4 && gy < H && gx + 3 < W;
      if (ok) src = base1 + (size_t)c * plane + (long)gy * W + gx;
    }
    gsrc[k] = src;
    gstride |= ok ? (1u << k) : 0u;
  }
  const size_t chunk_stride = (size_t)CC * plane;

  auto issue = [&](int st) {
    float* stage = myring + st * (kTilesPerWg * BUF);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int wi = wave + k * WPT;
      if (wi < NINSTR) {                             // wave-uniform
        lds_dma16_v(gsrc[k], stage + wi * 256);
        gsrc[k] += (gstride & (1u << k)) ? chunk_stride : 0;
      }
    }
  };

  f32x4 acc[P][9][3];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int d = 0; d < 9; ++d)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[p][d][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int hh = lane >> 5, l31 = lane & 31;
  const int r0 = wave * 2 * P;                       // lanes 0-31: rows r0 + p, lanes 32-63: rows r0 + P + p
  const int src_lane = (r0 + P * hh) * SP + l31;     // my dword inside a channel block, before row / group offsets
  const int tgt_lane = SRC_FLOATS + (r0 + P * hh) * TP + l31;

  // profiling hook (RFN_CORR_TRACE): wave 0 of every workgroup records 100 MHz timestamps of its phases
  const bool tracing = trace != nullptr && threadIdx.x == 0;
  long long t_start = 0, t_wait = 0, t_loop = 0, c_start = 0;
  if (trace) { t_start = wall_clock64(); c_start = clock64(); }

  // The channel loop is ONE stream of source-row steps: step t = c * NX + x of a chunk issues the <= 3 P MFMAs of source
  // row x of channel c.  Operands are software-pipelined by hand across channel AND chunk boundaries: the three row
  // operands of step t + LA and the P target operands of the next channel are loaded while step t runs (an LDS read
  // takes ~130 clocks, a step <= 48), so a wave never starts cold; sched_barrier keeps the compiler from hoisting every
  // load of a chunk to the top (it would, and spill).  The hand-off of chunk ck + 1 (counted vmcnt wait + raw barrier:
  // __syncthreads() would carry a vmcnt(0) fence) sits in the MIDDLE of chunk ck, before the first look-ahead read
  // into it; the same barrier says that everybody is done with chunk ck - 1, whose stage then takes chunk ck + D - 1.
  constexpr int NSTEP = CC * NX;
  constexpr int LA = 4;                                // source-row steps of look-ahead
  constexpr int SYNC_T = NSTEP / 2 - 1;
  static_assert(NSTEP % (LA + 1) == 0 && CC % 2 == 0, "operand rings keep their phase across chunks");
  static_assert(SYNC_T + LA < NSTEP, "hand-off before the first read of the next chunk");
  constexpr int STAGE = kTilesPerWg * BUF;
  const int nchunks = C / CC;                          // >= D - 1 (checked by the launcher)
  const bool kfull = wave < KFULL;                     // this wave issues K (else K - 1) instructions per chunk
  float aq[LA + 1][3];
  float bq[2][P];
  auto load_row = [&](const float* sb, int t, float (&a)[3]) {
    const float* q = sb + (t / NX) * SRC_CH + (t % NX) * SP;
    a[0] = q[0]; a[1] = q[4]; a[2] = q[8];
  };

#pragma unroll
  for (int k = 0; k < D - 1; ++k) issue(k);
  if (kfull) wait_vmcnt<(D - 2) * K>(); else wait_vmcnt<(D - 2) * (K - 1)>();
  asm volatile("s_barrier" ::: "memory");
#pragma unroll
  for (int p = 0; p < P; ++p) bq[0][p] = myring[tgt_lane + p * TP];
#pragma unroll
  for (int t = 0; t < LA; ++t) load_row(myring + src_lane, t, aq[t]);

  int st = 0;
  for (int ck = 0; ck < nchunks; ++ck) {
    const int nxt = st + 1 == D ? 0 : st + 1;
    static_assert(SP == TP, "target rows at a constant offset from the source rows");
    const float* sb = myring + st * STAGE + src_lane;
    const float* tb = sb + SRC_FLOATS;
    const float* sbn = myring + nxt * STAGE + src_lane;
    const float* tbn = sbn + SRC_FLOATS;
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
      const int c = t / NX, x = t % NX;
      if (t == SYNC_T) {
        long long tw = 0;
        if (trace) tw = wall_clock64();
        if (ck + D - 2 < nchunks) {                    // chunks ck + 2 .. ck + D - 2 may still be in flight
          if (kfull) wait_vmcnt<(D - 3) * K>(); else wait_vmcnt<(D - 3) * (K - 1)>();
        } else {
          wait_vmcnt<0>();
        }
        asm volatile("s_barrier" ::: "memory");
        if (trace) t_wait += wall_clock64() - tw;
        if (ck + D - 1 < nchunks && !(ablate & 1)) issue(st == 0 ? D - 1 : st - 1);
      }
      if (t + LA < NSTEP) load_row(sb, t + LA, aq[(t + LA) % (LA + 1)]);
      else load_row(sbn, t + LA - NSTEP, aq[(t + LA) % (LA + 1)]);
      if (x == NX - 3) {
        const float* q = c + 1 < CC ? tb + (c + 1) * TGT_CH : tbn;
#pragma unroll
        for (int p = 0; p < P; ++p) bq[(c + 1) & 1][p] = q[p * TP];
      }
      const float(&a)[3] = aq[t % (LA + 1)];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int d = x - p;
        if (d >= 0 && d <= 8) {
          const float bp = bq[c & 1][p];
          acc[p][d][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bp, acc[p][d][0], 0, 0, 0);
          acc[p][d][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bp, acc[p][d][1], 0, 0, 0);
          acc[p][d][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bp, acc[p][d][2], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    st = nxt;
  }
  long long c_loop = 0;
  if (trace) { t_loop = wall_clock64(); c_loop = clock64() - c_start; }

  // ---- epilogue: rotate each accumulator group by j, pick the 9 planes, (ReLU + L2 norm), store ----
  // one target row at a time: lanes are pixels, the 81 values of a pixel sit in this lane's registers
  const int j = lane & 3;
  const bool j1 = (j & 1) != 0, j2 = (j & 2) != 0;
  const int wx = w0 + l31;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float o[9][9];
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      float R[3][4];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const f32x4 A = acc[p][d][g];
        const float t0 = j1 ? A[1] : A[0], t1 = j1 ? A[2] : A[1], t2 = j1 ? A[3] : A[2], t3 = j1 ? A[0] : A[3];
        R[g][0] = j2 ? t2 : t0; R[g][1] = j2 ? t3 : t1; R[g][2] = j2 ? t0 : t2; R[g][3] = j2 ? t1 : t3;
      }
      float* q = o[d];
      q[0] = R[0][0];
      q[1] = j < 3 ? R[0][1] : R[1][1];
      q[2] = j < 2 ? R[0][2] : R[1][2];
      q[3] = j < 1 ? R[0][3] : R[1][3];
      q[4] = R[1][0];
      q[5] = j < 3 ? R[1][1] : R[2][1];
      q[6] = j < 2 ? R[1][2] : R[2][2];
      q[7] = j < 1 ? R[1][3] : R[2][3];
      q[8] = R[2][0];
      if constexpr (FUSE) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          q[e] = fmaxf(q[e], 0.f);
          ss = fmaf(q[e], q[e], ss);
        }
      }
    }
    float sc = 1.f;
    if constexpr (FUSE) sc = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    const int h = h0 + r0 + p + P * hh;
    if (live && wx < W && h < H && !(ablate & 4)) {
      float* ob = out + (size_t)n * 81 * plane + (size_t)h * W + wx;
#pragma unroll
      for (int d = 0; d < 9; ++d)
#pragma unroll
        for (int e = 0; e < 9; ++e) ob[(size_t)(d * 9 + e) * plane] = o[d][e] * sc;
    }
  }
  if (trace) {
    const long long t_alu = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have been acknowledged
    const long long t_end = wall_clock64();
    if (tracing) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
      long long* r = trace + (size_t)blockIdx.x * 8;
      r[0] = t_start; r[1] = t_wait; r[2] = t_loop; r[3] = t_alu; r[4] = t_end; r[5] = hw; r[6] = __smid(); r[7] = c_loop;
    }
  }
}

// ---- wave-private variant: no workgroup barrier anywhere in the channel loop ------------------------------------------
// Each wave stages ITS OWN 12 source rows x 40 pixels + 4 target rows x 32 pixels of a channel (16 LDS rows of 48 floats
// = exactly 3 DMA instructions) into its own D-deep ring, so the hand-off of a channel is a counted `s_waitcnt vmcnt`
// of the wave itself.  What that buys: with tiles shared by the workgroup (kernel above) every chunk ends in an
// s_barrier at which all 8 waves stop, the matrix pipe drains, and all of them then issue their DMAs at once -- measured
// 46 % pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES) and 40 % of the wave cycles in waits; without barriers the waves
// drift apart and a SIMD always has a wave with MFMAs to issue.  What it costs: the source rows that neighbouring waves
// share are fetched (from L2) and held in LDS once per wave -- 3 KB instead of 1.9 KB per wave and channel.
// Every wave issues exactly 3 DMA instructions per channel, also past the last channel (all lanes then read the zero
// slot), so the channel loop has no branch and one wait count.
template <bool FUSE, int D>
__global__ __launch_bounds__(512) void corr9_mfma_wave_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, int C, int H, int W,
    int tilesX, int tilesY, int ntiles, int ablate, long long* __restrict__ trace) {
  constexpr int P = 2, NX = P + 8, LA = 4;
  constexpr int PITCH = 48, V = PITCH / 4;
  constexpr int NSRC = 2 * P + 8, NTGT = 2 * P;       // source / target rows of a wave (12 / 4)
  constexpr int CH = (NSRC + NTGT) * PITCH;           // floats per channel and wave: 768 = 3 DMA instructions
  static_assert(CH == 3 * 256 && NX % (LA + 1) == 0, "3 DMA instructions per channel; operand ring keeps its phase");
  __shared__ __attribute__((aligned(16))) float ring[8 * D * CH];

  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave of the workgroup, 0..7
  const int half = wv >> 2, wave = wv & 3, lane = threadIdx.x & 63;
  float* const wring = ring + wv * (D * CH);

  const int tile = blockIdx.x * kTilesPerWg + half;
  const bool live = tile < ntiles;
  int bid = live ? tile : 0;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int h0 = ty * kTH, w0 = tx * kTW;
  const int r0 = wave * 2 * P;                       // my target rows r0 .. r0 + 3 of the tile
  const size_t plane = (size_t)H * W;
  const float* base1 = in1 + (size_t)n * C * plane;  // target features of this image
  const float* base2 = in2 + (size_t)n * C * plane;  // source features

  const float* gsrc[3];
  unsigned gvalid = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int slot = k * 64 + lane;
    const int v = slot % V, row = slot / V;
    bool ok = live;
    const float* src = g_zero_slot;
    if (row < NSRC) {
      const int gy = h0 + r0 - kHalo + row, gx = w0 - kHalo + 4 * v;
      ok = ok && v < (kTW + 2 * kHalo) / 4 && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
      if (ok) src = base2 + (long)gy * W + gx;
    } else {
      const int gy = h0 + r0 + row - NSRC, gx = w0 + 4 * v;
      ok = ok && v < kTW / 4 && gy < H && gx + 3 < W;
      if (ok) src = base1 + (long)gy * W + gx;
    }
    gsrc[k] = src;
    gvalid |= ok ? (1u << k) : 0u;
  }

  // one DMA instruction (k-th of a channel) into `stage`; `real` = the channel exists
  auto issue_one = [&](int k, float* stage, bool real) {
    lds_dma16_v(real ? gsrc[k] : (const float*)g_zero_slot, stage + k * 256);
    gsrc[k] += (gvalid & (1u << k)) ? plane : 0;
  };

  f32x4 acc[P][9][3];
#pragma unroll
  for (int p = 0; p < P; ++p)
#pragma unroll
    for (int d = 0; d < 9; ++d)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[p][d][g] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int hh = lane >> 5, l31 = lane & 31;
  const int lane_off = P * hh * PITCH + l31;          // lanes 0-31: target rows 0, 1; lanes 32-63: target rows 2, 3

  const bool tracing = trace != nullptr && threadIdx.x == 0;
  long long t_start = 0, t_loop = 0, c_start = 0;
  if (trace) { t_start = wall_clock64(); c_start = clock64(); }

  float aq[LA + 1][3];
  float bc[P], bn[P];
  auto load_row = [&](const float* sb, int x, float (&a)[3]) {
    const float* q = sb + x * PITCH;
    a[0] = q[0]; a[1] = q[4]; a[2] = q[8];
  };

#pragma unroll
  for (int s = 0; s < D - 1; ++s)
#pragma unroll
    for (int k = 0; k < 3; ++k) issue_one(k, wring + s * CH, s < C);
  wait_vmcnt<(D - 2) * 3>();                           // channel 0 has landed
#pragma unroll
  for (int p = 0; p < P; ++p) bc[p] = wring[lane_off + (NSRC + p) * PITCH];
#pragma unroll
  for (int x = 0; x < LA; ++x) load_row(wring + lane_off, x, aq[x]);

  int st = 0;
  for (int c = 0; c < C; ++c) {
    const int nst = st + 1 == D ? 0 : st + 1;
    const int pst = st == 0 ? D - 1 : st - 1;          // stage of channel c - 1 == stage of channel c + D - 1
    const float* sb = wring + st * CH + lane_off;
    const float* sbn = wring + nst * CH + lane_off;
    float* const ist = wring + pst * CH;
    const bool more = c + D - 1 < C;
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      if (x >= 1 && x <= 3) issue_one(x - 1, ist, more);
      if (x == NX - LA) wait_vmcnt<(D - 2) * 3>();     // channel c + 1 has landed (look-ahead reads it from here on)
      if (x + LA < NX) load_row(sb, x + LA, aq[(x + LA) % (LA + 1)]);
      else load_row(sbn, x + LA - NX, aq[(x + LA) % (LA + 1)]);
      if (x == NX - 3) {
#pragma unroll
        for (int p = 0; p < P; ++p) bn[p] = sbn[(NSRC + p) * PITCH];
      }
      const float(&a)[3] = aq[x % (LA + 1)];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int d = x - p;
        if (d >= 0 && d <= 8) {
          acc[p][d][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0], bc[p], acc[p][d][0], 0, 0, 0);
          acc[p][d][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1], bc[p], acc[p][d][1], 0, 0, 0);
          acc[p][d][2] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[2], bc[p], acc[p][d][2], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) bc[p] = bn[p];
    st = nst;
  }
  wait_vmcnt<0>();                                     // the trailing zero-slot DMAs
  long long c_loop = 0;
  if (trace) { t_loop = wall_clock64(); c_loop = clock64() - c_start; }

  // ---- epilogue (as above): rotate each accumulator group by j, pick the 9 planes, (ReLU + L2 norm), store ----
  const int j = lane & 3;
  const bool j1 = (j & 1) != 0, j2 = (j & 2) != 0;
  const int wx = w0 + l31;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float o[9][9];
    float ss = 0.f;
#pragma unroll
    for (int d = 0; d < 9; ++d) {
      float R[3][4];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const f32x4 A = acc[p][d][g];
        const float t0 = j1 ? A[1] : A[0], t1 = j1 ? A[2] : A[1], t2 = j1 ? A[3] : A[2], t3 = j1 ? A[0] : A[3];
        R[g][0] = j2 ? t2 : t0; R[g][1] = j2 ? t3 : t1; R[g][2] = j2 ? t0 : t2; R[g][3] = j2 ? t1 : t3;
      }
      float* q = o[d];
      q[0] = R[0][0];
      q[1] = j < 3 ? R[0][1] : R[1][1];
      q[2] = j < 2 ? R[0][2] : R[1][2];
      q[3] = j < 1 ? R[0][3] : R[1][3];
      q[4] = R[1][0];
      q[5] = j < 3 ? R[1][1] : R[2][1];
      q[6] = j < 2 ? R[1][2] : R[2][2];
      q[7] = j < 1 ? R[1][3] : R[2][3];
      q[8] = R[2][0];
      if constexpr (FUSE) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          q[e] = fmaxf(q[e], 0.f);
          ss = fmaf(q[e], q[e], ss);
        }
      }
    }
    float sc = 1.f;
    if constexpr (FUSE) sc = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    const int h = h0 + r0 + p + P * hh;
    if (live && wx < W && h < H && !(ablate & 4)) {
      float* ob = out + (size_t)n * 81 * plane + (size_t)h * W + wx;
#pragma unroll
      for (int d = 0; d < 9; ++d)
#pragma unroll
        for (int e = 0; e < 9; ++e) ob[(size_t)(d * 9 + e) * plane] = o[d][e] * sc;
    }
  }
  if (trace) {
    const long long t_alu = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my stores have been acknowledged
    const long long t_end = wall_clock64();
    if (tracing) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hw));
      long long* r = trace + (size_t)blockIdx.x * 8;
      r[0] = t_start; r[1] = 0; r[2] = t_loop; r[3] = t_alu; r[4] = t_end; r[5] = hw; r[6] = __smid(); r[7] = c_loop;
    }
  }
}

}  // namespace

// Launcher used by corr.hip's dispatch.  Returns RFN_OK, or a positive value when the shape is outside what this kernel
// takes (the caller then uses the VALU kernels).
int launch_corr9_mfma(const float* in1, const float* in2, float* out, int B, int C, int H, int W, bool fuse,
                      hipStream_t st) {
  if ((W & 3) != 0 || (C % 8) != 0) return 1;
  if ((long)C * H * W * 4 >= (1L << 32)) return 1;                 // 32-bit per-image byte offsets
  static const int ablate = getenv("RFN_CORR_ABLATE") ? atoi(getenv("RFN_CORR_ABLATE")) : 0;
  // profiling only: RFN_CORR_TRACE=<file> makes every launch synchronous and dumps per-workgroup phase timestamps
  static const char* trace_path = getenv("RFN_CORR_TRACE");
  static long long* trace_buf = nullptr;
  const int tilesX = cdiv(W, kTW), tilesY = cdiv(H, kTH);
  const long ntiles = (long)B * tilesX * tilesY;
  const long blocks = (ntiles + kTilesPerWg - 1) / kTilesPerWg;
  if (blocks <= 0 || ntiles > 0x7fffffffL) return fail(RFN_EINVAL, "corr9: grid too large");
  const dim3 grid((unsigned)blocks);
  if (trace_path && !trace_buf && hipMalloc(&trace_buf, sizeof(long long) * 8 * 65536) != hipSuccess) trace_buf = nullptr;
  // RFN_CORR_TRACE_EVERY=N: trace (and synchronise) only every N-th launch, so that the traced launch runs in the
  // sustained conditions of N - 1 back-to-back launches before it
  static const int trace_every = getenv("RFN_CORR_TRACE_EVERY") ? atoi(getenv("RFN_CORR_TRACE_EVERY")) : 1;
  static long launch_no = 0;
  ++launch_no;
  long long* tr = (trace_path && blocks <= 65536 && launch_no % trace_every == 0) ? trace_buf : nullptr;
  static const int cfg = getenv("RFN_CORR_MFMA_CFG") ? atoi(getenv("RFN_CORR_MFMA_CFG")) : 0;   // tuning knob
#define RFN_LAUNCH(CC_, D_, P_)                                                                                   \
  {                                                                                                               \
    const dim3 block(64 * (kTH / (2 * P_)) * kTilesPerWg);                                                        \
    if (fuse)                                                                                                     \
      hipLaunchKernelGGL((corr9_mfma_kernel<true, CC_, D_, P_>), grid, block, 0, st, in1, in2, out, C, H, W,       \
                         tilesX, tilesY, (int)ntiles, ablate, tr);                                                \
    else                                                                                                          \
      hipLaunchKernelGGL((corr9_mfma_kernel<false, CC_, D_, P_>), grid, block, 0, st, in1, in2, out, C, H, W,      \
                         tilesX, tilesY, (int)ntiles, ablate, tr);                                                \
  }
#define RFN_LAUNCH_WAVE(D_)                                                                                       \
  {                                                                                                               \
    if (fuse)                                                                                                     \
      hipLaunchKernelGGL((corr9_mfma_wave_kernel<true, D_>), grid, dim3(512), 0, st, in1, in2, out, C, H, W,       \
                         tilesX, tilesY, (int)ntiles, ablate, tr);                                                \
    else                                                                                                          \
      hipLaunchKernelGGL((corr9_mfma_wave_kernel<false, D_>), grid, dim3(512), 0, st, in1, in2, out, C, H, W,      \
                         tilesX, tilesY, (int)ntiles, ablate, tr);                                                \
  }
  switch (cfg) {
    case 1: RFN_LAUNCH(2, 3, 2) break;
    case 2: RFN_LAUNCH(2, 5, 2) break;
    case 3: RFN_LAUNCH_WAVE(4) break;
    default: RFN_LAUNCH_WAVE(6) break;
  }
#undef RFN_LAUNCH_WAVE
#undef RFN_LAUNCH
  const int rc = check_launch("corr9_mfma_kernel");
  if (tr && rc == RFN_OK) {
    static long long host[8 * 65536];
    if (hipStreamSynchronize(st) == hipSuccess &&
        hipMemcpy(host, tr, sizeof(long long) * 8 * blocks, hipMemcpyDeviceToHost) == hipSuccess) {
      if (FILE* f = fopen(trace_path, "wb")) {
        fwrite(host, sizeof(long long) * 8, (size_t)blocks, f);
        fclose(f);
      }
    }
  }
  return rc;
}

}  // namespace rfn
