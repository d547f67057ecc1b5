// refign_amd/csrc/conv3x3.hip -- halo-tiled 3 x 3 convolution (stride 1, padding 1, no dilation) on the matrix pipe: VGG-16's layers
// (vgg.py:108-120), the DAFormer bottleneck (daformer.py:65-126) and every other dense 3 x 3 layer of the gradient-free passes with
// C % 64 == 0 and N % 64 == 0.   Y[b, y, x, n] = act( sum_{ky, kx, c} X[b, y + ky - 1, x + kx - 1, c] W[n, (ky, kx, c)] + bias[n] )
//
// Why a second convolution kernel.  The implicit-GEMM form (mfma_gemm.hip, GATHER) walks K = (tap, channel) in 64-wide steps and
// fetches, for EVERY tap, the tile's input rows again: nine LDS-DMA passes over (almost) the same pixels, and both operands per
// K-step -- 8 DMA instructions per wave against 32 MFMAs on the 256 x 256 tile, 24 against 32 on the 128 x 64 tile of VGG's
// 64-channel layers.  The launches are paced by the ISSUE of those instructions (profiles/r05: matrix pipe busy 38 % of the
// bottleneck convolution, 0.07-0.14 of the MFMA peak on conv1_2 / conv2_x).  Here a workgroup owns an 8 x 32 OUTPUT tile and BN
// output channels; per 64-channel chunk the 10 x 34 input HALO tile is staged ONCE (44 DMA instructions, double buffered) and
// the nine taps read it at shifted LDS addresses -- a tap costs one weight stage (BN rows x 128 B) and a ninth of a halo stage:
// 5.2 DMA instructions per 32 MFMAs at BN = 128, 6.4 at BN = 64.
//
// Layout.  LDS rows are 128 B (64 channels of one pixel / one filter row); the 16-byte piece c of row r lives at slot
// c ^ ((r >> 1) & 7) (mfma_gemm.hip's swizzle: any 16 CONSECUTIVE rows cover the 16 slots of a 256-byte bank row, whatever the
// first row -- which is what lets a tap start its 32-pixel fragment at any halo column).  DMA destinations are lane-linear, the
// swizzle goes into the per-lane SOURCE address; out-of-image halo pixels come from a zero page.
// Pipeline.  Steps s = (chunk, tap); weight stages in a ring of four, fetched three steps ahead; the next chunk's halo tile is
// fetched during the current chunk's taps 0-5 (2, 2, 2, 2, 2, 1 instructions per wave).  Hand-off: counted `s_waitcnt vmcnt`
// (loads retire in order: everything older than the two previous steps' instructions has landed) + one barrier per step.
// 8 waves = 4 x 2: a wave owns output rows 2 pw, 2 pw + 1 (two 32-pixel MFMA column blocks) x BN / 2 channels, computed transposed
// (D[channel][pixel]: a lane holds runs of 4 consecutive channels of one pixel); the epilogue adds the bias, applies the
// activation, rounds and stages the tile through LDS for row-contiguous 16-byte stores.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace rfn {

__device__ uint4 g_zero_page_c3[4];          // zero-initialised: DMA source of halo pixels outside the image

constexpr int kC3TH = 8, kC3TW = 32;
constexpr int kC3HW = kC3TW + 2, kC3HP = (kC3TH + 2) * kC3HW;      // 34 columns, 340 halo pixels
constexpr int kC3AI = 44;                                           // halo DMA instructions per chunk (8 rows each, 352 rows)
constexpr int kC3AB = kC3AI * 1024;                                 // bytes of a halo stage
constexpr int kC3NB = 4;                                            // weight ring depth

__device__ __forceinline__ void wait_dma_rt(int n) {               // wave-uniform n
  switch (n) {
    case 0: wait_dma_upto<0>(); break;
    case 1: wait_dma_upto<1>(); break;
    case 2: wait_dma_upto<2>(); break;
    case 3: wait_dma_upto<3>(); break;
    case 4: wait_dma_upto<4>(); break;
    case 5: wait_dma_upto<5>(); break;
    case 6: wait_dma_upto<6>(); break;
    case 7: wait_dma_upto<7>(); break;
    case 8: wait_dma_upto<8>(); break;
    case 9: wait_dma_upto<9>(); break;
    case 10: wait_dma_upto<10>(); break;
    case 11: wait_dma_upto<11>(); break;
    case 12: wait_dma_upto<12>(); break;
    default: wait_dma_upto<0>(); break;
  }
}

template <int I, int N, typename F> __device__ __forceinline__ void c3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    c3_static_for<I + 1, N>(f);
  }
}

// NA = halo stages: 2, or 1 for C <= 128 (one or two chunks per tile) -- with BN = 64 and 4 waves that is 76 KB of LDS, so TWO
// workgroups share a CU and one's prologue / chunk refill / epilogue runs under the other's taps
// OUT32: fp32 result, fp32 bias (the split-bf16 convolutions of the matcher's head: C = 3 Cp channels per pixel, split32.py).
// C % 32 == 0: a last half chunk reads zeros for its upper 32 channels; N % 8 == 0: filter rows past N are clamped, stores masked.
template <int DT, int BN, int NA, bool OUT32>
__global__ __launch_bounds__(NA == 1 ? 256 : 512) void conv3x3_halo_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                           const void* __restrict__ bias_, void* __restrict__ Y_, int H,
                                                           int W, int C, int N, long ldw, long ldy, int tiles_y, int tiles_x,
                                                           int act, const void* __restrict__ zero) {
  using E = Elem<DT>;
  using vec8 = typename E::vec8;
  // 8 waves = 4 (pixel rows 2 pw, 2 pw + 1) x 2 (channel halves): two waves per SIMD, one's fragment reads under the other's MFMAs.
  // NA == 1: 4 waves (all channels each), two such workgroups per CU.
  constexpr int NWV = NA == 1 ? 4 : 8;
  constexpr int WC = BN / (NWV / 4);          // channels of a wave
  constexpr int CB = WC / 32;                 // its 32-channel blocks (2 or 1)
  constexpr int NBI = BN / (8 * NWV);         // weight DMA instructions per wave and step (BN rows, 8 per instruction)
  constexpr int NAI = (kC3AI + NWV - 1) / NWV;   // halo DMA instructions of a wave: 6 (5 for waves 4-7) / 11
  constexpr int BSTAGE = BN * 128;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NA * kC3AB + kC3NB * BSTAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, j = lane & 31;
  const int pw = wave & 3, cw = NWV == 8 ? wave >> 2 : 0;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int NT = (N + BN - 1) / BN;
  const int nt = bid % NT;
  bid /= NT;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y, b = bid / tiles_y;
  const int y0 = ty * kC3TH, x0 = tx * kC3TW, n0 = nt * BN;
  const int nchunks = (C + 63) / 64, S = nchunks * 9;
  const bool half_tail = (C & 63) != 0;        // the last chunk holds 32 channels: source pieces 4-7 of its rows are zeros

  // ---- DMA sources: lane = (row of the instruction's 8, 16-byte piece); source piece = piece ^ swizzle(row) ----------------------
  // halo instructions wave + 8 k: waves 0-3 carry six of the 44, waves 4-7 five
  const int drow = lane >> 3, piece = lane & 7;
  const int na = NWV == 8 ? (wave < 4 ? 6 : 5) : 11;            // (scalar)
  unsigned aoff[NAI];
  unsigned ahigh = 0u, bhigh = 0u;
#pragma unroll
  for (int k = 0; k < NAI; ++k) {
    const int row = 8 * (wave + NWV * k) + drow;
    const int hy = row / kC3HW, hx = row - hy * kC3HW;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = row < kC3HP && y >= 0 && y < H && x >= 0 && x < W;
    const int sp = piece ^ ((row >> 1) & 7);
    aoff[k] = ok ? (unsigned)((((long)(b * H + y) * W + x) * C) * 2 + sp * 16) : 0xffffffffu;
    if (sp >= 4) ahigh |= 1u << k;            // (per lane: this instruction's piece is in the upper half of the row)
  }
  unsigned boff[NBI];
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    const int n = 8 * (wave + NWV * k) + drow;
    const int sp = piece ^ ((n >> 1) & 7);
    boff[k] = (unsigned)(((long)min(n0 + n, N - 1) * ldw) * 2 + sp * 16);
    if (sp >= 4) bhigh |= 1u << k;
  }
  const unsigned char* Xb = (const unsigned char*)X;
  const unsigned char* Wb = (const unsigned char*)Wt;
  auto issue_a = [&](int k, int chunk) {
    const bool cut = half_tail && chunk == nchunks - 1 && ((ahigh >> k) & 1u);
    const void* src = (aoff[k] != 0xffffffffu && !cut) ? (const void*)(Xb + aoff[k] + (long)chunk * 128) : zero;
    lds_dma16(src, smem + (chunk & (NA - 1)) * kC3AB + (wave + NWV * k) * 1024);
  };
  auto issue_b = [&](int s) {                // step s = 9 chunk + tap
    const int c = s / 9, t = s - 9 * c;
    const long koff = ((long)t * C + (long)c * 64) * 2;
    unsigned char* dst = smem + NA * kC3AB + (s & (kC3NB - 1)) * BSTAGE;
    const bool tail = half_tail && c == nchunks - 1;
#pragma unroll
    for (int k = 0; k < NBI; ++k) {
      const bool cut = tail && ((bhigh >> k) & 1u);
      lds_dma16(cut ? zero : (const void*)(Wb + boff[k] + koff), dst + (wave + NWV * k) * 1024);
    }
  };

  f32x16 acc[CB][2];
#pragma unroll
  for (int a = 0; a < CB; ++a)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][p][r] = 0.f;

  // ---- prologue: halo tile of chunk 0, weight stages of steps 0, 1, 2 ---------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < NAI - 1; ++k) issue_a(k, 0);
  if (NWV == 4 || na == 6) issue_a(NAI - 1, 0);
  issue_b(0);
  issue_b(1);
  issue_b(2);

  const int swzw = (j >> 1) & 7;
  for (int c = 0; c < nchunks; ++c) {
    const bool notlast = c + 1 < nchunks;
    const unsigned char* as = smem + (c & (NA - 1)) * kC3AB;
    c3_static_for<0, 9>([&](auto tt) {
      constexpr int t = decltype(tt)::value;
      constexpr int dy = t / 3, dx = t % 3;
      const int s = 9 * c + t;
      // everything older than what the two previous steps issued has landed (loads retire in order): this step's weight stage,
      // and on tap 0 the chunk's halo tile (its last piece went out on tap 5 of the previous chunk, before that stage).
      // Halo pieces of the next chunk: one per wave on taps 0-4, a sixth on tap 5 for waves 0-3.
      if (NA == 1 && t == 0 && c > 0) {
        // single halo stage, several chunks (C = 128): the stage is re-filled at the chunk boundary, in the open -- the CU's other
        // workgroup has the matrix pipe meanwhile
        wg_barrier();                        // everybody is done with the previous chunk's tile
#pragma unroll
        for (int k = 0; k < NAI; ++k) issue_a(k, c);
        wait_dma_all();
      } else if (s + 2 >= S) {
        wait_dma_all();
      } else {
        int en = t == 0 ? 0 : (t == 1 ? 1 : (t <= 5 ? 2 : 0));
        if (t == 6) en = 1 + (na == 6 ? 1 : 0);
        if (t == 7) en = na == 6 ? 1 : 0;
        wait_dma_rt(2 * NBI + ((NA == 2 && notlast) ? en : 0));
      }
      wg_barrier();
      if constexpr (NA == 2) {
        if (notlast) {
          if constexpr (t < 5) issue_a(t, c + 1);
          else if constexpr (t == 5) {
            if (na == 6) issue_a(5, c + 1);
          }
        }
      }
      if (s + 3 < S) issue_b(s + 3);
      const unsigned char* bs = smem + NA * kC3AB + (s & (kC3NB - 1)) * BSTAGE;
      const int hr0 = (2 * pw + dy) * kC3HW + j + dx, hr1 = hr0 + kC3HW;
      const int sw0 = (hr0 >> 1) & 7, sw1 = (hr1 >> 1) & 7;
      const unsigned char* x0p = as + hr0 * 128;
      const unsigned char* x1p = as + hr1 * 128;
      const unsigned char* wp = bs + (cw * WC + j) * 128;
      // fragment reads one k-sub-step ahead of the MFMAs (two register sets; sched_barrier pins the order)
      vec8 wf[2][CB], xf[2][2];
      auto read_frags = [&](int ks, int bsel) {
#pragma unroll
        for (int a = 0; a < CB; ++a) wf[bsel][a] = *(const vec8*)(wp + a * 32 * 128 + (((2 * ks + g) ^ swzw) * 16));
        xf[bsel][0] = *(const vec8*)(x0p + (((2 * ks + g) ^ sw0) * 16));
        xf[bsel][1] = *(const vec8*)(x1p + (((2 * ks + g) ^ sw1) * 16));
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) read_frags(ks + 1, (ks + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < CB; ++a)
#pragma unroll
          for (int p = 0; p < 2; ++p) acc[a][p] = E::mma(wf[ks & 1][a], xf[ks & 1][p], acc[a][p]);
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  }
  wg_barrier();                              // everybody is done with the stages: the staging block lies over the halo stages

  // ---- epilogue: + bias, activation, rounding -> LDS [pixel][channel] -> 16-byte row-contiguous stores (channels past N masked) ----
  if constexpr (!OUT32) {
    const uint16_t* bias = (const uint16_t*)bias_;
    uint16_t* Y = (uint16_t*)Y_;
    constexpr int PITCH = WC + 8;              // halfs
    static_assert(NWV * 64 * PITCH * 2 <= NA * kC3AB + kC3NB * BSTAGE, "staging block fits the stages");
    uint16_t* stg = (uint16_t*)smem + wave * 64 * PITCH;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = 32 * a + 8 * q + 4 * g;                // four consecutive output channels of the wave's share
          float bf[4] = {0.f, 0.f, 0.f, 0.f};
          if (bias != nullptr && n0 + cw * WC + ch < N) unpack4<DT>(*(const u32x2*)(bias + n0 + cw * WC + ch), bf);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float z = acc[a][p][4 * q + e] + bf[e];
            if (act == 1) z = fmaxf(z, 0.f);
            else if (act == 3) z = z > 0.f ? z : 0.1f * z;
            v[e] = z;
          }
          *(u32x2*)(stg + (p * 32 + j) * PITCH + ch) = pack4<DT>(v[0], v[1], v[2], v[3]);
        }
    __syncthreads();
    constexpr int OPR = WC / 8;                // 16-byte pieces per pixel (of this wave's channels)
    constexpr int RPI = 64 / OPR;              // pixels per store instruction
    const int opiece = lane % OPR, orow = lane / OPR;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int prow = it * RPI + orow;
      const int y = y0 + 2 * pw + (prow >> 5), x = x0 + (prow & 31);
      const u32x4 val = *(const u32x4*)(stg + prow * PITCH + opiece * 8);
      const int ch0 = n0 + cw * WC + opiece * 8;
      if (y < H && x < W && ch0 < N) *(u32x4*)(Y + ((long)(b * H + y) * W + x) * ldy + ch0) = val;
    }
  } else {
    const float* bias = (const float*)bias_;
    float* Y = (float*)Y_;
    constexpr int PITCH = WC + 4;              // floats
    static_assert(NWV * 64 * PITCH * 4 <= NA * kC3AB + kC3NB * BSTAGE, "staging block fits the stages");
    float* stg = (float*)smem + wave * 64 * PITCH;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int a = 0; a < CB; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = 32 * a + 8 * q + 4 * g;
          f32x4 bf = {0.f, 0.f, 0.f, 0.f};
          if (bias != nullptr && n0 + cw * WC + ch < N) bf = *(const f32x4*)(bias + n0 + cw * WC + ch);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float z = acc[a][p][4 * q + e] + bf[e];
            if (act == 1) z = fmaxf(z, 0.f);
            else if (act == 3) z = z > 0.f ? z : 0.1f * z;
            v[e] = z;
          }
          *(f32x4*)(stg + (p * 32 + j) * PITCH + ch) = v;
        }
    __syncthreads();
    constexpr int OPR = WC / 4;                // 16-byte pieces per pixel (of this wave's channels)
    constexpr int RPI = 64 / OPR;
    const int opiece = lane % OPR, orow = lane / OPR;
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int prow = it * RPI + orow;
      const int y = y0 + 2 * pw + (prow >> 5), x = x0 + (prow & 31);
      const f32x4 val = *(const f32x4*)(stg + prow * PITCH + opiece * 4);
      const int ch0 = n0 + cw * WC + opiece * 4;
      if (y < H && x < W && ch0 < N) *(f32x4*)(Y + ((long)(b * H + y) * W + x) * ldy + ch0) = val;
    }
  }
}

// Launch; the caller (rfn_conv2d_nhwc / rfn_conv2d_nhwc_o32) has checked the geometry (3 x 3, stride 1, padding 1, no dilation, no
// residual).  out32: fp32 result and bias.  Returns RFN_OK, an error, or 1 when the problem is outside this kernel's domain (the caller
// then takes the implicit-GEMM kernel).
int launch_conv3x3_halo(const void* X, const void* W, const void* bias, void* Y, int B, int H, int Wd, int C, int N, long ldw,
                        long ldy, int act, int dtype, int out32, hipStream_t s) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("RFN_CONV_HALO");
    enabled = (e == nullptr || e[0] != '0') ? 1 : 0;
  }
  if (!enabled) return 1;
  if (C % 32 != 0 || N % 8 != 0 || ldy % (out32 ? 4 : 8) != 0 || ldw % 8 != 0 || ldw < 9L * C) return 1;
  if ((long)B * H * Wd * C * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return 1;     // 32-bit byte offsets
  if ((((size_t)X | (size_t)W | (size_t)Y) & 15) != 0 || (bias != nullptr && ((size_t)bias & (out32 ? 15 : 7)) != 0)) return 1;
  if ((out32 && dtype != 1) || N < 32) return 1;           // (a 64-channel tile for fewer than 32 output channels is mostly padding)
  const int tiles_y = cdiv(H, kC3TH), tiles_x = cdiv(Wd, kC3TW);
  // small or narrow maps waste the 8 x 32 tile (the matcher's 16 x 16 and 32 x 32 levels): the implicit-GEMM kernel keeps those
  if ((long)B * H * Wd < 4096 || (double)H * Wd < 0.7 * (double)tiles_y * kC3TH * tiles_x * kC3TW) return 1;
  // where it wins (tools/experiments/conv3x3_check.py time, profiles/r06_conv3x3_halo.txt): few input channels (the halo tile is most
  // of a step's traffic), many pixels, or few output channels (the implicit-GEMM kernel is then on its small tiles); with N >= 256 and
  // C >= 256 on <= 130 000 pixels its 256 x 256 tile is 2-5 % ahead
  if (N >= 256 && C > 128 && (long)B * H * Wd < 400000) return 1;
  // C <= 128: a single halo stage (re-filled in the open at the chunk boundary), 64-channel tiles, 4-wave workgroups, two per CU --
  // 3-5 % ahead of the double-buffered 8-wave form at C = 128, level with it from C = 256 on (profiles/r06_conv3x3_halo.txt)
  const bool one_chunk = C <= 128;
  const int BN = (N % 128 == 0 && !one_chunk) ? 128 : 64;
  const long blocks = (long)B * tiles_y * tiles_x * cdiv(N, BN);
  if (blocks >= (1L << 31)) return 1;
  static void* zero_page = nullptr;          // looked up once (first call is an eager warm-up, never inside a capture)
  if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page_c3)) != hipSuccess)
    return fail(RFN_ELAUNCH, "conv3x3_halo: zero page symbol");
#define RFN_C3(DT_, BN_, NA_, O32_)                                                                                               \
  hipLaunchKernelGGL((conv3x3_halo_kernel<DT_, BN_, NA_, O32_>), dim3((unsigned)blocks), dim3(NA_ == 1 ? 256 : 512), 0, s,          \
                     (const uint16_t*)X, (const uint16_t*)W, bias, Y, H, Wd, C, N, ldw, ldy, tiles_y, tiles_x, act,                \
                     (const void*)zero_page)
  if (out32) {
    if (one_chunk) RFN_C3(1, 64, 1, true); else if (BN == 128) RFN_C3(1, 128, 2, true); else RFN_C3(1, 64, 2, true);
  } else if (dtype == 1) {
    if (one_chunk) RFN_C3(1, 64, 1, false); else if (BN == 128) RFN_C3(1, 128, 2, false); else RFN_C3(1, 64, 2, false);
  } else {
    if (one_chunk) RFN_C3(2, 64, 1, false); else if (BN == 128) RFN_C3(2, 128, 2, false); else RFN_C3(2, 64, 2, false);
  }
#undef RFN_C3
  return check_launch("conv3x3_halo_kernel");
}

}  // namespace rfn
