// refign_amd/csrc/corr_f16.hip -- round 3: the 9x9 local correlation (+ ReLU + L2 norm over the 81 shifts) on the MATRIX pipe.
//
//   out[b, dy * 9 + dx, y, x] = sum_c T[b, c, y, x] * S[b, c, y + dy - 4, x + dx - 4]      (zero outside the image)
// = LocalFeatureCorrelationLayer.forward (models/modules.py:266-274) = the sampler of models/correlation_ops/correlation.cpp
// :80-129 with patch 9, kernel 1, followed by relu and F.normalize over the shift dimension.
//
// Why a second kernel.  The fp32 VALU kernel (corr.hip) needs 5.4 GFLOP of fp32 FMAs per level-1 launch, and on this chip
// that arithmetic -- not the 349 MB it moves -- sets its time (105 us = 0.42 of the HBM roofline; three schedules of it
// landed on the same figure).  The fp16 matrix pipe is 16x wider.  Precision comes from the operand
// FORMAT: a feature v is stored as hi = fp16(v), lo = fp16(v - hi) (22 significand bits; csrc/warp.hip split_f16_kernel
// writes them, 4 bytes per feature like fp32), and a product is evaluated as th.sh + th.sl + tl.sh with fp32 accumulation:
// ~2^-21 relative per product, below the fp32 kernel's own summation noise.  The price is the banded shape: a 16-pixel
// target strip times a 32-pixel source window per vertical shift is two 16x16x32 MFMA blocks of which 9 diagonals are
// wanted (28 %), times three products -- 57 GFLOP issued for 5.4 wanted, 29 us of matrix pipe at its measured 2 PFLOP/s.
//
// Operand format (both inputs): [b][chunk = c / 32][part][y][x][32 channels] fp16 -- per (chunk, part) a plane of 64-byte
// pixels, so a tile row is one contiguous run and an MFMA operand (8 channels of one pixel) is one 16-byte LDS read.
//
// Workgroup = 8 waves = an 8 x 32 target tile; wave (rp, st) owns target rows 2 rp, 2 rp + 1 x pixels 16 st .. + 15 and keeps
// all their 2 x 9 x 2 result blocks (144 accumulators) in registers while the 32-channel chunks stream by.  A stage is ONE
// part of ONE chunk of the 16 x 48-pixel source window (48 KB), in a 2-deep LDS ring filled by LDS-DMA; stage (c, hi) feeds
// th.sh and tl.sh, stage (c, lo) feeds th.sl.  A source row's two operand blocks are read once per stage and used by both
// target rows of the wave (2 ds_read_b128 per 4-8 MFMAs).  The 16-byte piece kg of pixel p sits at slot kg ^ s(p),
// s(p) = (p ^ (p >> 2)) & 3, applied on the DMA's source side: the lane groups of a read cover all banks.
// Epilogue: the accumulator blocks are scattered to an LDS image [row][shift][pixel] (the 9 wanted diagonals of each
// 16 x 16 block), then one wave per tile row applies ReLU, the L2 norm over the 81 shifts and writes 128-byte runs.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace rfn {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kCfTH = 8, kCfTW = 32;
constexpr int kCfSR = kCfTH + 8, kCfSP = kCfTW + 16;          // source window: 16 rows x 48 pixels
constexpr int kCfStage = kCfSR * kCfSP * 64;                    // 49 152 bytes
constexpr int kCfPitch = 33;                                    // floats per (row, shift) of the output image

__device__ uint4 g_zero_page_cf[4];

__device__ __forceinline__ f32x4 mma16(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

__global__ __launch_bounds__(512) void corr9_f16_kernel(const _Float16* __restrict__ trg, const _Float16* __restrict__ src,
                                                        float* __restrict__ out, int NC, int H, int W, int tilesX, int tilesY,
                                                        int ntiles, const void* zero, int fuse, int ablate) {
  // LDS: two 48 KB stages | output image of HALF a tile [4 rows][81 shifts][33] | partial sums of squares [4 rows][4][32]
  constexpr int kOt = 2 * kCfStage, kSsq = kOt + 4 * 81 * kCfPitch * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kSsq + 4 * 4 * 32 * 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rp = wave >> 1, st = wave & 1, j = lane & 15, kg = lane >> 4;
  const long plane = (long)H * W * 64;                          // bytes of one (chunk, part) plane
  const int G = gridDim.x;

  // ---- DMA pieces of this lane (tile independent): 48 instructions of 1 KB per stage, 6 per wave;
  // piece = 16 bytes = (window row, window pixel, slot); slot holds channel group kgs = slot ^ s(pixel)
  int drow[6], dpx[6], drel[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int piece = (wave * 6 + q) * 64 + lane;
    const int pxl = piece >> 2, slot = piece & 3;
    drow[q] = pxl / kCfSP;
    dpx[q] = pxl - drow[q] * kCfSP;
    drel[q] = ((drow[q] - 4) * W + (dpx[q] - 4)) * 64 + 16 * (slot ^ ((dpx[q] ^ (dpx[q] >> 2)) & 3));
  }
  struct Tile {
    int b, y0, x0;
  };
  auto decode = [&](int t) {
    Tile r;
    r.x0 = (t % tilesX) * kCfTW;
    t /= tilesX;
    r.y0 = (t % tilesY) * kCfTH;
    r.b = t / tilesY;
    return r;
  };
  // stage s = (chunk s >> 1, part s & 1) of tile `tl` into ring buffer `buf`
  auto issue = [&](const Tile& tl, int s, int buf) {
    if (ablate & 1) return;                                     // profiling only (RFN_CORR_ABLATE): results are then meaningless
    const unsigned char* pl = (const unsigned char*)src + ((long)tl.b * NC * 2 + s) * plane + ((long)tl.y0 * W + tl.x0) * 64;
    unsigned char* dst = smem + buf * kCfStage + wave * 6 * 1024;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int gy = tl.y0 - 4 + drow[q], gx = tl.x0 - 4 + dpx[q];
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W && dpx[q] < kCfTW + 8;
      lds_dma16(ok ? (const void*)(pl + drel[q]) : zero, dst + q * 1024);
    }
  };
  // target operands of this wave: rows 2 rp, 2 rp + 1 of the tile, pixel 16 st + j, channels 8 kg .. + 7 of a chunk, both parts
  h8 cur[2][2], nxt[2][2];                                       // [row][part]
  auto load_t = [&](const Tile& tl, int chunk, h8 (&t)[2][2]) {
    const int ya = min(tl.y0 + 2 * rp, H - 1), yb = min(tl.y0 + 2 * rp + 1, H - 1), px = min(tl.x0 + 16 * st + j, W - 1);
    const unsigned char* base = (const unsigned char*)trg + ((long)tl.b * NC * 2 + 2 * chunk) * plane + 16 * kg;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
      t[0][part] = *(const h8*)(base + part * plane + ((long)ya * W + px) * 64);
      t[1][part] = *(const h8*)(base + part * plane + ((long)yb * W + px) * 64);
    }
  };

  // read addresses inside a stage: this lane's pixel of block 0 / 1 and its swizzled slot
  const int p0 = 16 * st + j, p1 = p0 + 16;
  const int o0 = p0 * 64 + 16 * (kg ^ ((p0 ^ (p0 >> 2)) & 3)), o1 = p1 * 64 + 16 * (kg ^ ((p1 ^ (p1 >> 2)) & 3));
  f32x4 acc[2][9][2];
  // one source row's two operand blocks feed both target rows of the wave (shift r for row a, r - 1 for row b)
  auto rows = [&](const unsigned char* sb, const h8 (&ta_)[2], const h8 (&tb_)[2], auto nterms) {
    constexpr int NT = decltype(nterms)::value;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const h8 b0 = *(const h8*)(sb + r * (kCfSP * 64) + o0), b1 = *(const h8*)(sb + r * (kCfSP * 64) + o1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (r < 9) {
          acc[0][r][0] = mma16(ta_[t], b0, acc[0][r][0]);
          acc[0][r][1] = mma16(ta_[t], b1, acc[0][r][1]);
        }
        if (r > 0) {
          acc[1][r - 1][0] = mma16(tb_[t], b0, acc[1][r - 1][0]);
          acc[1][r - 1][1] = mma16(tb_[t], b1, acc[1][r - 1][1]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);                         // keep one source row's operands live at a time
    }
  };

  // persistent workgroup: tiles wg, wg + G, ...; consecutive logical workgroups sit on one XCD (xcd_remap), i.e. the tiles
  // worked on at any one time by an XCD are neighbours and share their halos in its L2
  int t = xcd_remap(blockIdx.x, G);
  if (t >= ntiles) return;
  Tile tl = decode(t);
  issue(tl, 0, 0);
  load_t(tl, 0, cur);
  for (; t < ntiles; t += G) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int d = 0; d < 9; ++d)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[a][d][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // one iteration = one 32-channel chunk = two stages in straight-line code (one loop-carried copy of the accumulators)
    for (int chunk = 0; chunk < NC; ++chunk) {
      const int s = 2 * chunk;
      wait_dma_all();                                            // stage (chunk, hi): th.sh + tl.sh
      wg_barrier();
      issue(tl, s + 1, 1);
      if (chunk + 1 < NC) load_t(tl, chunk + 1, nxt);            // a whole chunk (0.8 us of MFMAs) ahead of its use
      if (!(ablate & 2)) rows(smem + (2 * rp) * (kCfSP * 64), cur[0], cur[1], std::integral_constant<int, 2>{});
      wait_dma_all();                                            // stage (chunk, lo): th.sl
      wg_barrier();
      if (chunk + 1 < NC) issue(tl, s + 2, 0);
      if (!(ablate & 2)) rows(smem + kCfStage + (2 * rp) * (kCfSP * 64), cur[0], cur[1], std::integral_constant<int, 1>{});
      if (chunk + 1 < NC) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int k = 0; k < 2; ++k) cur[a][k] = nxt[a][k];
      }
    }
    // the next tile's first stage and target operands start now and land under this tile's epilogue (ring buffer 0 was last
    // read in the final hi stage; the epilogue works in its own LDS region)
    const Tile done = tl;
    if (t + G < ntiles) {
      tl = decode(t + G);
      issue(tl, 0, 0);
      load_t(tl, 0, cur);
    }
    if (ablate & 4) {
      if (acc[0][0][0][0] == 123.456f) out[0] = 0.f;
      continue;
    }
    // ---- epilogue, half a tile (4 rows) at a time.  D block (shift dy, block blk): lane (j, kg) register e = target pixel
    // i = 4 kg + e x source pixel j + 16 blk of the strip's window, i.e. horizontal shift dx = j + 16 blk - i; 0 <= dx <= 8
    // are the wanted diagonals, the rest goes to a dump slot (branch-free scatter).
    float* ot = (float*)(smem + kOt);                            // [4][81][33]
    float* ssq = (float*)(smem + kSsq);                          // [4 rows][4 shift quarters][32 pixels]
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if ((rp >> 1) == hf) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int i = 4 * kg + e, dx = j + 16 * blk - i;
              // (row, shift dy * 9 + dx, pixel); unwanted entries land in the pad column 32 of the (row, dy * 9) line: either
              // way the address is a per-lane base + a compile-time multiple of dy
              const int line = (2 * (rp & 1) + a) * 81 * kCfPitch;
              float* p = ot + ((dx >= 0 && dx <= 8) ? line + dx * kCfPitch + 16 * st + i : line + 32);
#pragma unroll
              for (int d = 0; d < 9; ++d) p[d * 9 * kCfPitch] = acc[a][d][blk][e];
            }
      }
      wg_barrier();
      // 8 waves x 64 lanes over 4 rows x 32 pixels x 4 shift quarters (21, 20, 20, 20 shifts): row = wave >> 1,
      // quarter = 2 (wave & 1) + (lane >> 5)
      const int row = wave >> 1, px = lane & 31, qt = 2 * (wave & 1) + (lane >> 5);
      const int c0 = qt == 0 ? 0 : 1 + 20 * qt, cn = qt == 0 ? 21 : 20;
      const float* col = ot + (row * 81 + c0) * kCfPitch + px;
      float v[21];
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < 21; ++c) {
        v[c] = c < cn ? col[c * kCfPitch] : 0.f;
        if (fuse) v[c] = fmaxf(v[c], 0.f);
        ss = fmaf(v[c], v[c], ss);
      }
      float scale = 1.f;
      if (fuse) {
        ssq[(row * 4 + qt) * 32 + px] = ss;
        wg_barrier();
        const float* sp = ssq + row * 128 + px;
        scale = 1.f / fmaxf(sqrtf((sp[0] + sp[32]) + (sp[64] + sp[96])), 1e-12f);
      }
      const int y = done.y0 + 4 * hf + row, x = done.x0 + px;
      if (y < H && x < W) {
        float* o = out + (((long)done.b * 81 + c0) * H + y) * W + x;
        const long cs = (long)H * W;
#pragma unroll
        for (int c = 0; c < 21; ++c)
          if (c < cn) o[c * cs] = v[c] * scale;
      }
      wg_barrier();                                              // the image is rewritten by the other half / the next tile
    }
  }
}

}  // namespace rfn

extern "C" {

int rfn_local_corr_layer_f16split(const void* target_split, const void* source_split, float* out, int B, int C, int H, int W,
                                  int fuse, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(target_split && source_split && out, "rfn_local_corr_layer_f16split: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && C % 32 == 0 && H > 0 && W > 0, "rfn_local_corr_layer_f16split: sizes (C %% 32)");
  RFN_REQUIRE((long)H * W * 64 < (1L << 31), "rfn_local_corr_layer_f16split: plane too large");
  static void* zero_page = nullptr;
  if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page_cf)) != hipSuccess)
    return fail(RFN_ELAUNCH, "rfn_local_corr_layer_f16split: zero page symbol");
  const int tilesX = cdiv(W, kCfTW), tilesY = cdiv(H, kCfTH);
  const long tiles = (long)B * tilesX * tilesY;
  RFN_REQUIRE(tiles < 0x7fffffffL, "rfn_local_corr_layer_f16split: grid too large");
  static const int ablate = getenv("RFN_CORR_ABLATE") ? atoi(getenv("RFN_CORR_ABLATE")) : 0;     // profiling only
  static const int wgs = getenv("RFN_CORR_F16_WGS") ? atoi(getenv("RFN_CORR_F16_WGS")) : 256;     // one per CU (142 KB of LDS)
  hipLaunchKernelGGL(corr9_f16_kernel, dim3((unsigned)std::min<long>(tiles, wgs)), dim3(512), 0, (hipStream_t)stream,
                     (const _Float16*)target_split, (const _Float16*)source_split, out, C / 32, H, W, tilesX, tilesY,
                     (int)tiles, (const void*)zero_page, fuse, ablate);
  return check_launch("corr9_f16_kernel");
}

}  // extern "C"
