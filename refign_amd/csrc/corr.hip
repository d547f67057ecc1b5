// refign_amd/csrc/corr.hip -- spatial correlation sampler for gfx950 (MI355X), written from scratch.
//
// What it computes (reference: models/correlation_ops/correlation.cpp:13-42,80-129; the CUDA twin
// correlation_cuda_kernel.cu:26-88 computes the same numbers on NHWC copies):
//   out[n,ph,pw,h,w] = sum_c sum_{i<kH} sum_{j<kW} in1[n,c,u+i*dil,v+j*dil] * in2[n,c,u+i*dil+sU,v+j*dil+sV]
//   u = -pad + h*stride, v = -pad + w*stride, sU = (ph - (patchH-1)/2)*dil_patch, zero outside the image.
//
// Two forward kernels:
//   * corr9_tile_kernel -- the ONE parameterisation the hot path uses (kernel 1, patch 9, stride 1, pad 0;
//     models/modules.py:268-270).  HBM-bound by design: algorithmic traffic 4*(2C+81) bytes per pixel against
//     162*C flops per pixel (15 flop/B at C=128), so the only way to approach the HBM roofline on fp32 VALU is
//     register tiling.  Layout:
//       - a workgroup owns a TH x 64 pixel tile and walks the channels in chunks of CC; per chunk the
//         target tile (TH x 64) and the source tile with its 4-pixel halo ((TH+8) x 72) are staged in LDS with
//         16-byte coalesced global loads (NCHW rows are contiguous along w);
//       - a thread owns a strip of 4 consecutive pixels x 9 horizontal shifts x 3 vertical shifts
//         = 108 fp32 accumulators; per channel it issues 1 + 9 ds_read_b128 (40 dwords) for 108 FMAs;
//         the three vertical-shift groups of a tile are three sets of waves (wave-uniform, no divergence);
//       - lanes 16-31 / 48-63 of a wave take their strips rotated by 14 so that, with the 72-dword row pitch,
//         every ds_read_b128 lane group {0-3,12-15,20-27}/{4-11,16-19,28-31} hits 64 distinct banks
//         (MI355X LDS: b128 reads are serviced in those non-contiguous 16-lane groups);
//       - the 81 output planes are written once, 16 bytes per lane, contiguous along w.
//     Optional fusions selected by template flags:
//       FUSE  : ReLU + L2-normalisation over the 81 shifts (LocalFeatureCorrelationLayer, modules.py:272-273)
//               in the epilogue -- saves three full passes over the 81xHxW volume;
//       WARP  : the source tile is produced by bilinear-warping the un-warped source features with the flow
//               while staging (helpers/matching_utils.py:11-49) -- the warped feature map never exists in HBM.
//   * corr_generic_fwd_kernel -- any parameterisation, one thread per output element, float or double.
//
// Backward (reference: correlation.cpp:44-78,131-183; CUDA gather form correlation_cuda_kernel.cu:91-238):
//   * corr_k1_bwd_kernel     -- gather form for kernel 1 / stride 1 / pad 0 (deterministic, no atomics);
//   * corr_generic_bwd_kernel -- scatter with hardware float/double atomics for everything else.
#include <cstdlib>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "common.h"

namespace rfn {

// --------------------------------------------------------------------------------------------------------
// Tiled patch-9 forward
// --------------------------------------------------------------------------------------------------------
constexpr int kTW = 64;            // tile width in pixels
constexpr int kStrips = 16;        // 4-pixel strips per tile row
constexpr int kHalo = 4;           // (9-1)/2
constexpr int kPitch = kTW + 2 * kHalo;  // 72 dwords: LDS row pitch of BOTH tiles (same bank geometry)

// bilinear tap of the warp (matching_utils.py:35-43): returns the 4 clamped offsets and weights (0 for taps
// outside the image => zero padding).
struct WarpTap {
  int o00, o01, o10, o11;
  float w00, w01, w10, w11;
};

__device__ __forceinline__ WarpTap make_tap(float gx, float gy, float fx, float fy, int H, int W) {
#pragma clang fp contract(off)
  // normalise exactly as the reference does (2*v/max(W-1,1) - 1), then ATen's align_corners=True unnormalise
  const float wm = (float)max(W - 1, 1), hm = (float)max(H - 1, 1);
  float vx = 2.0f * (gx + fx) / wm - 1.0f;
  float vy = 2.0f * (gy + fy) / hm - 1.0f;
  float ix = ((vx + 1.0f) / 2.0f) * (float)(W - 1);
  float iy = ((vy + 1.0f) / 2.0f) * (float)(H - 1);
  float x0f = floorf(ix), y0f = floorf(iy);
  float tx = ix - x0f, ty = iy - y0f;
  // guard int conversion against huge / NaN flows
  x0f = fminf(fmaxf(x0f, -2.0f), (float)W);
  y0f = fminf(fmaxf(y0f, -2.0f), (float)H);
  if (!(ix == ix) || !(iy == iy)) { x0f = -2.0f; y0f = -2.0f; }
  int x0 = (int)x0f, y0 = (int)y0f, x1 = x0 + 1, y1 = y0 + 1;
  bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
  bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
  int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
  WarpTap t;
  t.o00 = cy0 * W + cx0; t.o01 = cy0 * W + cx1; t.o10 = cy1 * W + cx0; t.o11 = cy1 * W + cx1;
  float ax = 1.0f - tx, ay = 1.0f - ty;
  t.w00 = (vx0 && vy0) ? ax * ay : 0.0f;
  t.w01 = (vx1 && vy0) ? tx * ay : 0.0f;
  t.w10 = (vx0 && vy1) ? ax * ty : 0.0f;
  t.w11 = (vx1 && vy1) ? tx * ty : 0.0f;
  return t;
}

template <int TH, int CC, bool FUSE, bool WARP>
__global__ __launch_bounds__(TH * kStrips * 3) void corr9_tile_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, const float* __restrict__ flow,
    float* __restrict__ out, int C, int H, int W, int tilesX, int tilesY) {
  constexpr int NT = TH * kStrips * 3;
  constexpr int R2 = TH + 2 * kHalo;
  __shared__ __attribute__((aligned(16))) float smem[CC * (R2 + TH) * kPitch];
  float* s2 = smem;                      // [CC][R2][kPitch]  source tile + halo
  float* s1 = smem + CC * R2 * kPitch;   // [CC][TH][kPitch]  target tile

  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int h0 = ty * TH, w0 = tx * kTW;

  const int lane = tid & 63, wave = tid >> 6;
  constexpr int WPG = TH / 4;            // waves per vertical-shift group
  const int dyg = wave / WPG;            // 0..2 : vertical shifts dyg*3 .. dyg*3+2   (wave-uniform)
  const int q = lane >> 4, j = lane & 15;
  const int row = (wave % WPG) * 4 + q;
  const int strip = (q & 1) ? ((j + 14) & 15) : j;   // bank-conflict-free b128 lane groups, see header

  const size_t plane = (size_t)H * W;
  const float* p1 = in1 + (size_t)n * C * plane;
  const float* p2 = in2 + (size_t)n * C * plane;
  const bool vec_ok = (W & 3) == 0;

  float acc[3][9][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[a][b][i] = 0.0f;

  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();  // previous chunk fully consumed
    // ---- stage the target tile: CC x TH rows x 16 float4 ----
    for (int idx = tid; idx < CC * TH * kStrips; idx += NT) {
      const int v = idx % kStrips, r = (idx / kStrips) % TH, c = idx / (kStrips * TH);
      const int gy = h0 + r, gx = w0 + 4 * v, gc = c0 + c;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gc < C && gy < H) {
        const float* src = p1 + (size_t)gc * plane + (size_t)gy * W + gx;
        if (vec_ok && gx + 3 < W) {
          val = *reinterpret_cast<const float4*>(src);
        } else {
          if (gx + 0 < W) val.x = src[0];
          if (gx + 1 < W) val.y = src[1];
          if (gx + 2 < W) val.z = src[2];
          if (gx + 3 < W) val.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&s1[(c * TH + r) * kPitch + 4 * v]) = val;
    }
    // ---- stage the source tile with halo ----
    if constexpr (!WARP) {
      constexpr int V2 = kPitch / 4;  // 18 float4 per row
      for (int idx = tid; idx < CC * R2 * V2; idx += NT) {
        const int v = idx % V2, r = (idx / V2) % R2, c = idx / (V2 * R2);
        const int gy = h0 - kHalo + r, gx = w0 - kHalo + 4 * v, gc = c0 + c;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gc < C && gy >= 0 && gy < H) {
          const float* src = p2 + (size_t)gc * plane + (size_t)gy * W + gx;
          if (vec_ok && gx >= 0 && gx + 3 < W) {
            val = *reinterpret_cast<const float4*>(src);
          } else {
            if (gx + 0 >= 0 && gx + 0 < W) val.x = src[0];
            if (gx + 1 >= 0 && gx + 1 < W) val.y = src[1];
            if (gx + 2 >= 0 && gx + 2 < W) val.z = src[2];
            if (gx + 3 >= 0 && gx + 3 < W) val.w = src[3];
          }
        }
        *reinterpret_cast<float4*>(&s2[(c * R2 + r) * kPitch + 4 * v]) = val;
      }
    } else {
      // one (row, col) position per iteration, all CC channels: the bilinear taps are channel-independent
      const float* fl = flow + (size_t)n * 2 * plane;
      for (int pos = tid; pos < R2 * kPitch; pos += NT) {
        const int x = pos % kPitch, r = pos / kPitch;
        const int gy = h0 - kHalo + r, gx = w0 - kHalo + x;
        const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
        WarpTap t;
        if (inside) {
          const float fx = fl[(size_t)gy * W + gx], fy = fl[plane + (size_t)gy * W + gx];
          t = make_tap((float)gx, (float)gy, fx, fy, H, W);
        }
#pragma unroll
        for (int c = 0; c < CC; ++c) {
          float val = 0.0f;
          if (inside && c0 + c < C) {
            const float* src = p2 + (size_t)(c0 + c) * plane;
            // same accumulation order as ATen's grid_sampler_2d (nw, ne, sw, se)
            val = src[t.o00] * t.w00 + src[t.o01] * t.w01 + src[t.o10] * t.w10 + src[t.o11] * t.w11;
          }
          s2[(c * R2 + r) * kPitch + x] = val;
        }
      }
    }
    __syncthreads();

    // ---- 108 FMAs per channel per thread ----
#pragma unroll 2
    for (int c = 0; c < CC; ++c) {
      const float4 a = *reinterpret_cast<const float4*>(&s1[(c * TH + row) * kPitch + 4 * strip]);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        const float* rp = &s2[(c * R2 + row + dyg * 3 + dyi) * kPitch + 4 * strip];
        const float4 b0 = *reinterpret_cast<const float4*>(rp);
        const float4 b1 = *reinterpret_cast<const float4*>(rp + 4);
        const float4 b2 = *reinterpret_cast<const float4*>(rp + 8);
        const float bv[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
#pragma unroll
        for (int dx = 0; dx < 9; ++dx)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[dyi][dx][i] = fmaf(av[i], bv[i + dx], acc[dyi][dx][i]);
      }
    }
  }

  // ---- epilogue ----
  const int h = h0 + row, wx = w0 + 4 * strip;
  float scale[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (FUSE) {
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = fmaxf(acc[a][b][i], 0.0f);
          acc[a][b][i] = v;
          ss[i] = fmaf(v, v, ss[i]);
        }
    __syncthreads();  // tiles no longer needed: reuse LDS for the 3-way reduction
    float* red = smem;  // [3][TH][64]
    *reinterpret_cast<float4*>(&red[(dyg * TH + row) * kTW + 4 * strip]) = make_float4(ss[0], ss[1], ss[2], ss[3]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tot = red[(0 * TH + row) * kTW + 4 * strip + i] + red[(1 * TH + row) * kTW + 4 * strip + i] +
                        red[(2 * TH + row) * kTW + 4 * strip + i];
      scale[i] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);   // F.normalize: v / max(||v||, eps)
    }
  }
  if (h < H && wx < W) {
    float* obase = out + ((size_t)n * 81 + (size_t)(dyg * 3) * 9) * plane + (size_t)h * W + wx;
    const bool full = vec_ok && wx + 3 < W;
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi)
#pragma unroll
      for (int dx = 0; dx < 9; ++dx) {
        float* o = obase + (size_t)(dyi * 9 + dx) * plane;
        float r0 = acc[dyi][dx][0], r1 = acc[dyi][dx][1], r2 = acc[dyi][dx][2], r3 = acc[dyi][dx][3];
        if constexpr (FUSE) { r0 *= scale[0]; r1 *= scale[1]; r2 *= scale[2]; r3 *= scale[3]; }
        if (full) {
          *reinterpret_cast<float4*>(o) = make_float4(r0, r1, r2, r3);
        } else {
          o[0] = r0;
          if (wx + 1 < W) o[1] = r1;
          if (wx + 2 < W) o[2] = r2;
          if (wx + 3 < W) o[3] = r3;
        }
      }
  }
}


// --------------------------------------------------------------------------------------------------------
// Tiled patch-9 forward, LDS-DMA pipeline (the fast path when W % 4 == 0 and C % CC == 0)
//   Same tile/thread decomposition as corr9_tile_kernel, but the two tiles of a channel chunk are brought in by
//   `global_load_lds_dwordx4` (global -> LDS DMA, no VGPR round trip, no ds_write) into a 2-deep LDS ring, so the
//   HBM/L2 latency of chunk k+1 is hidden behind the 108 x CC FMAs per thread of chunk k and there is exactly one
//   barrier per chunk.  The DMA writes wave-uniform-base + lane*16, so the LDS image of a chunk is the linear
//   sequence of float4 "slots" (channel, row, 18 slots per 72-float row); rows 0..R2-1 are the source tile with
//   halo, rows R2..R2+TH-1 the target tile (slots 16,17 of those rows are never written).  Slots that fall outside
//   the image are never written either: the ring is zeroed once at kernel start and the out-of-image pattern is
//   the same for every chunk, which gives the zero padding of the reference for free.
// --------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. N - 1
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// One LDS-DMA instruction (global -> LDS, 16 bytes per lane, lane i lands at lds_wave_base + 16 i), issued through
// inline asm ON PURPOSE.  With the `__builtin_amdgcn_global_load_lds` builtin hipcc tracks the pending LDS write and,
// whenever it cannot prove that a later `ds_read` touches a different object, inserts `s_waitcnt vmcnt(0)` in front of
// it -- measured here: the wait landed inside the channel loop, i.e. the DMA of chunk k+1 never overlapped the FMAs of
// chunk k (DMA-only 62 us + FMA-only 81 us = 147 us total, profiles/r01_kbench_corr_ablation.txt).  Inline asm is
// invisible to that pass; the hand-off is then entirely ours: `s_waitcnt vmcnt(N)` + barrier before the first read.
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(sbase) : "memory", "m0");
#pragma clang diagnostic pop
}
// The same under a lane mask, WITHOUT control flow the compiler can see: exec is narrowed and restored inside the asm
// statement.  (An `if (ok) lds_dma16(...)` is a branch around the instruction; between the unrolled product steps of
// corr9_pipe2_kernel such branches split the chunk into basic blocks, and the products -- pure arithmetic -- then sink
// out of their steps: spills, and no pipeline left.)
// (scalar 64-bit base + 32-bit byte offset per lane: one address VGPR instead of two)
__device__ __forceinline__ void lds_dma16_masked(const void* sbase, unsigned voff, unsigned lds_wave_base,
                                                 unsigned long long mask) {
  unsigned long long saved;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  asm volatile("s_mov_b32 m0, %3\n\ts_and_saveexec_b64 %0, %4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "v"(voff), "s"(sbase), "s"(lds_wave_base), "s"(mask) : "memory", "m0", "scc");
#pragma clang diagnostic pop
}

// (Rounds 1-4 kept three earlier generations of this kernel in the library behind RFN_CORR_VARIANT -- the 2-stage LDS-DMA
// kernel `corr9_dma_kernel`, the first 4-stage ring `corr9_pipe_kernel` with its tile-tail race, and ~25 tile / chunk
// variants of them.  Round 5 removed them: one pipelined kernel below serves every map the tiled path takes, the round-1
// register-staged `corr9_tile_kernel` above everything else (odd widths, few channels, the fused-warp form, samples of 4 GB
// and more).  Their measurements stay in profiles/r01-r04_*corr*.)

// NTILE = 2: the workgroup is two independent halves, each owning its own tile (ids 2b, 2b+1 of the launch's tile
// order), its own LDS region and its own DMA stream; only the per-chunk barrier is shared.  This doubles the waves per
// CU (6-wave workgroups do not co-reside: their 2,2,1,1 wave placement over the SIMDs leaves no room for a second one
// at 168 VGPRs) while keeping the fine 16x32 tile granularity that fills 256 CUs evenly: K4 level 1 (2 x 270 x 480) is
// 510 tiles = 255 workgroups = one even round over the 256 CUs.
// ---- 4-stage variant, second take (round 4, second half) ---------------------------------------------------------
// What the counters said about corr9_pipe_kernel (profiles/r04_pmc_corr9.txt): it is VALU-bound, not LDS-bound
// (`ds_read_b128` moves 256 B / clock / CU on gfx950: 10 reads x 12 waves x 4 clocks = 480 clocks per channel against
// 3 waves x 66 VALU x 4 = 792 per SIMD), with 66 VALU instructions per channel where 54 are products (5 address
// computations + 7 moves), every `ds_read` waited for right behind its issue (the other two waves of the SIMD are the
// only latency cover) and 24 VGPRs holding wave-uniform LDS-DMA destinations as generic pointers.  Same decomposition,
// same products in the same order (identical results), but
//   * the wave index is a scalar (`readfirstlane`): DMA destinations live in SGPRs;
//   * a chunk's six (channel, vertical shift) row steps are unrolled with immediate offsets from two per-lane bases,
//     and software-pipelined: the three `ds_read_b128` of step s + 1 are issued into a second register set before the
//     18 packed products of step s (`sched_barrier` pins the order; hipcc's waitcnt pass then emits lgkmcnt(3));
//   * the DMA of the chunk three ahead is issued behind the first reads of a chunk, inside their latency.
//   * KSPLIT (NTILE == 2): the two wave groups of the workgroup walk the two HALVES of the channels of ONE tile, each through
//     its own ring, and the second group's 108 sums per lane are added to the first's through the LDS in front of the epilogue
//     (three rounds of 36 floats per lane in the ring's own memory).  For maps of at most one tile per CU: a chunk takes a CU
//     0.5 us with three waves and 0.85 us with six, so half the chunks per wave is ~15-20 % off the launch (not half: the
//     CU, not the wave, is what a chunk waits for).
//   * PAIR (round 5; K4 level 2: 2 x 135 x 240): a map whose width leaves at most HALF a tile column over (240 = 7 x 32 + 16)
//     spends a whole tile per row band on it -- 17 x 8 x 2 = 272 tiles, and the 16 CUs that get a second workgroup pace the
//     launch (83 us against 64.5 us for 256 tiles, profiles/r04_corr_ksplit.txt).  With PAIR the left-over column band of TWO
//     images shares one tile: strips 0..3 of the tile are image n's last columns, strips 4..7 image n + 1's (same rows), each
//     half with its own 4-pixel halo in the 48-float LDS row (2 x 24).  Only per-lane constants change (DMA source offsets,
//     the source-row base, the store address); the LDS image, the products and the hand-offs are the same.  `tilesX` then
//     counts the FULL tile columns, tiles [0, nreg) are regular and tiles [nreg, ntiles) are the (image pair, row band) edge
//     tiles: 2 x 17 x 7 + 17 = 255 workgroups -- one per CU, which also lets the channel split (KSPLIT) apply.
//   * JOIN (cross-workgroup channel split of tiny maps, launch_corr9_split; 1 = raw sums, 2 = ReLU + L2 norm): gridDim.y workgroups
//     share a tile, each writes the raw partial volume of its channel slice, takes a ticket, and the LAST arriver adds the slices in
//     slice order (deterministic), applies the epilogue and writes the result: the level in ONE launch (round 5; before: a second
//     kernel, 14 + 18 us for K4 level 3).  Publication across XCDs: slab stores, vmcnt(0), barrier, agent-scope release fence,
//     relaxed agent-scope ticket; the reducer takes an agent-scope acquire fence before it reads (the L2s of the XCDs are not
//     coherent with each other).  The tickets are zero on entry and the joining workgroup puts its ticket back to zero: the
//     caller zeroes them once per workspace (a memset node in front of every launch was tried first: under hipGraph replay the
//     level came out stale -- tests/test_align_gpu.py K2 golden -- while eager launches were right).
template <int TH, int TW, bool FUSE, int MINW, int NTILE, int DEPTH = 1, int NS = 4, bool KSPLIT = false, bool PAIR = false,
          int JOIN = 0>
__global__ __launch_bounds__(TH * (TW / 4) * 3 * NTILE, MINW) void corr9_pipe2_kernel(
    const float* __restrict__ in1, const float* __restrict__ in2, float* __restrict__ out, int C, int H, int W,
    int tilesX, int tilesY, int ntiles, int xcd_remap, int nreg, int Ctot, long part_stride, float* __restrict__ joined,
    unsigned* __restrict__ tickets) {
  // (C = the channels THIS workgroup walks, Ctot = the tensors' channel count: equal except in the cross-workgroup channel
  // split of tiny maps, launch_corr9_split, where blockIdx.y picks the slice [blockIdx.y C, (blockIdx.y + 1) C) and the raw
  // partial sums go to out + blockIdx.y * part_stride)
  static_assert(TW == 64 || TW == 32, "tile width 64 or 32");
  static_assert(!PAIR || TW == 32, "paired edge tiles: two 16-column halves with their halos fill the 48-float row of TW = 32");
  constexpr int CC = 2;
  constexpr int STRIPS = TW / 4;
  constexpr int RPW = 64 / STRIPS;
  constexpr int NT = TH * STRIPS * 3;
  constexpr int NW = NT / 64;
  constexpr int R2 = TH + 2 * kHalo;
  constexpr int ROWS = R2 + TH;
  constexpr int PITCH = (TW == 64) ? 72 : 48;        // (bank geometry: see corr9_pipe_kernel)
  constexpr int V = PITCH / 4;
  constexpr int VU2 = (TW + 2 * kHalo) / 4;
  constexpr int SLOTS = CC * ROWS * V;
  constexpr int NINSTR = (SLOTS + 63) / 64;
  constexpr int K = (NINSTR + NW - 1) / NW;
  constexpr int BUF = NINSTR * 64 * 4;
  // NS ring stages of [tile of the workgroup][BUF]: NS - 1 chunks are in flight while one is consumed.  Level 1 (12 waves per
  // CU, a chunk's arithmetic ~1 us): 4 stages.  The small-map instance (one 3-wave workgroup per CU, a chunk's arithmetic ~0.3 us)
  // is bound by chunks-in-flight / DMA latency with 4 (3 chunks per ~2 us = what it measured: 0.67 us per chunk): 8 stages.
  static_assert(NS >= 3 && NS * NTILE * BUF * 4 <= 160 * 1024, "ring fits the LDS");
  static_assert(!KSPLIT || ((NTILE == 2 || NTILE == 4) && NS * NTILE * BUF >= (NTILE - 1) * 36 * NT),
                "channel split: two or four wave groups, sums exchanged in the rings");
  __shared__ __attribute__((aligned(16))) float rings_all[NS * NTILE * BUF];
  constexpr unsigned RB = NTILE * BUF * 4;             // bytes from one stage to the next

  const int gw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: wave of the workgroup
  const int half = gw / NW, wave = gw % NW;
  const int lane = threadIdx.x & 63;
  // xcd_remap: workgroup b runs on XCD b % 8 (round-robin dispatch); give every XCD a contiguous band of tiles, so that the
  // halo rows two neighbouring tiles both fetch meet in ONE L2 (the launch is a single round: neighbours run side by side)
  int wg = blockIdx.x;
  if (xcd_remap) {
    const int nwg = gridDim.x, qq = nwg / 8, rr = nwg % 8, xcd = wg % 8, loc = wg / 8;
    wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int tile = KSPLIT ? wg : wg * NTILE + half;
  const bool live = tile < ntiles;
  int bid = live ? tile : 0;
  const bool edge = PAIR && bid >= nreg;               // (scalar) a left-over column band shared by images n and n + 1
  int tx, ty, n;
  if (edge) {
    bid -= nreg;
    tx = tilesX;
    ty = bid % tilesY;
    n = 2 * (bid / tilesY);
  } else {
    tx = bid % tilesX; bid /= tilesX;
    ty = bid % tilesY;
    n = bid / tilesY;
  }
  const int h0 = ty * TH, w0 = tx * TW;
  constexpr int WPG = TH / RPW;
  const int dyg = wave / WPG;
  const int q = lane / STRIPS, j = lane % STRIPS;
  const int row = (wave % WPG) * RPW + q;
  const int strip = (TW == 64) ? ((q & 1) ? ((j + 14) & 15) : j) : (j ^ ((((q & 3) == 1) || ((q & 3) == 2)) ? 4 : 0));

  const size_t plane = (size_t)H * W;
  const int Cw = KSPLIT ? C / NTILE : C;              // the channels this wave group walks (KSPLIT: group g takes [g Cw, (g+1) Cw))
  const size_t c0 = (size_t)blockIdx.y * C + (KSPLIT ? (size_t)half * Cw : 0);
  const float* p1 = in1 + ((size_t)n * Ctot + c0) * plane;
  const float* p2 = in2 + ((size_t)n * Ctot + c0) * plane;
  const float* const slab0 = out;                      // (JOIN) slice 0 of the partial volumes
  out += (size_t)blockIdx.y * part_stride;

  // LDS image of a chunk (differs from corr9_pipe_kernel's): the TARGET rows of both channels first, then the source rows --
  // [f1 c0 | f1 c1 | f2 c0 | f2 c1] -- so that the f1 / f2 border falls on a DMA instruction border (CC * TH * V slots = a
  // multiple of 64): every instruction then has ONE tensor as its source, i.e. a scalar 64-bit base (the sample's plane 0) +
  // a 32-bit byte offset per lane: half the address registers, and the loop has none to spare (a spilled pointer is reloaded
  // through `s_waitcnt vmcnt(0)`, which also waits for every DMA in flight).
  constexpr int F1SLOTS = CC * TH * V;
  static_assert(F1SLOTS % 64 == 0, "the target rows of a chunk must fill whole DMA instructions");
  constexpr int F2BASE = CC * TH * PITCH;              // float offset of the source rows in a ring buffer
  unsigned goff[K];
  unsigned long long gmask[K];                         // lanes of my k-th DMA instruction that have a source (scalar)
  const float* gbase[K];                               // scalar: p1 or p2
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int wi = wave + k * NW;
    const int slot = wi * 64 + lane;
    bool ok = live && (wi < NINSTR) && (slot < SLOTS);
    long off;
    if (wi * 64 < F1SLOTS) {                           // scalar
      const int v = slot % V, rr = (slot / V) % TH, c = slot / (V * TH);
      const int sel = (edge && v >= STRIPS / 2) ? 1 : 0;                   // second image of an edge tile
      const int gy = h0 + rr, gx = w0 + 4 * (v - sel * (STRIPS / 2));
      ok = ok && v < STRIPS && gy < H && gx + 3 < W;
      off = ((long)c + (long)sel * Ctot) * (long)plane + (long)gy * W + gx;
      gbase[k] = p1;
    } else {
      const int s2 = slot - F1SLOTS;
      const int v = s2 % V, rr = (s2 / V) % R2, c = s2 / (V * R2);
      // edge tile: vectors 0 .. V/2 - 1 = image n's columns w0 - 4 .. w0 + 19, vectors V/2 .. V - 1 the same of image n + 1
      const int sel = (edge && v >= V / 2) ? 1 : 0;
      const int gy = h0 - kHalo + rr, gx = w0 - kHalo + 4 * (v - sel * (V / 2));
      ok = ok && (edge || v < VU2) && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
      off = ((long)c + (long)sel * Ctot) * (long)plane + (long)gy * W + gx;
      gbase[k] = p2;
    }
    goff[k] = ok ? (unsigned)(off * (long)sizeof(float)) : 0u;
    gmask[k] = __builtin_amdgcn_ballot_w64(ok);
  }

  // one DMA instruction of mine (k-th of the chunk; `on` = all ones, or zero for the chunks past the tile's last) -- issued
  // ONE AT A TIME between the row steps of a chunk: right behind the barrier every wave of the SIMD is in the same phase,
  // and three vector-memory issues in a row there idle the VALU
  const unsigned chunk_bytes = (unsigned)(CC * plane * sizeof(float));
  auto lds_addr = [](float* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) float*)p; };
  auto issue1 = [&](unsigned ring, int k, unsigned long long on) {
    const int wi = wave + k * NW;                      // scalar
    lds_dma16_masked(gbase[k], goff[k], ring + (unsigned)wi * 1024u, gmask[k] & on);
    goff[k] += chunk_bytes;
  };
  const unsigned lds0 = lds_addr(rings_all + half * BUF);              // my tile's stage 0 (scalar); stage r at lds0 + r * RB
  auto issue = [&](unsigned ring) {
#pragma unroll
    for (int k = 0; k < K; ++k) issue1(ring, k, ~0ull);
  };

  f32x2 accp[3][4][4];
  float accs[3][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      accs[a][i] = 0.0f;
#pragma unroll
      for (int p = 0; p < 4; ++p) accp[a][i][p] = f32x2{0.0f, 0.0f};
    }

  const int a_off = row * PITCH + 4 * strip;                           // my 4 target pixels (channel 0 of the chunk)
  // my first source row (vertical shift dyg * 3); the second half of an edge tile reads the second image's columns, which
  // start V / 2 vectors into the row (a uniform shift per LDS lane group: strips 0..3 and 4..7 are never in one group)
  const int b_off = F2BASE + (row + dyg * 3) * PITCH + 4 * strip + ((edge && strip >= STRIPS / 2) ? 4 * (V / 2 - STRIPS / 2) : 0);

  // one (channel, vertical shift) step: 36 products of 4 target pixels with 12 source pixels = 16 packed + 4 single FMAs,
  // written as volatile asm: (a) products are pure arithmetic, and nothing else keeps instruction selection from emitting them
  // after the block's last side effect (= behind the next chunks' barriers, every loaded row spilled: seen); volatile asm
  // statements keep their order among themselves and with the DMA statements and barriers; (b) the broadcast of a target
  // pixel is an op_sel of the pair it was loaded in (hipcc moved pixel 3 into a fresh pair: one move per channel).
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto step = [&](const f32x4& a, const f32x4* b, int dyi) {
    const f32x2 ap[2] = {a.xy, a.zw};
    const f32x2 bp[6] = {b[0].xy, b[0].zw, b[1].xy, b[1].zw, b[2].xy, b[2].zw};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f32x2 bb = bp[(i + 2 * p + (i & 1)) / 2];
        if (i & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(accp[dyi][i][p]) : "v"(ap[i >> 1]), "v"(bb));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(accp[dyi][i][p]) : "v"(ap[i >> 1]), "v"(bb));
      }
      const int kb = (i & 1) ? i : i + 8;
      asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accs[dyi][i]) : "v"(a[i]), "v"(b[kb >> 2][kb & 3]));
    }
  };
  // LDS reads by byte address: per-lane base (of the ring in use) + immediate offset
  typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
  auto lds_ld = [](unsigned addr) { return *(lds_f32x4_ptr)(size_t)addr; };
  auto ldrow = [&](f32x4* b, unsigned pb, int st) {          // st = c * 3 + dyi, compile-time after unrolling
    const unsigned rp = pb + (unsigned)(((st / 3) * R2 + (st % 3)) * PITCH * 4);
    b[0] = lds_ld(rp);
    b[1] = lds_ld(rp + 16);
    b[2] = lds_ld(rp + 32);
  };
  // A chunk in two parts of three row steps.  `nxt`: the ring that takes the chunk three ahead (`on` = 0 for the tile's last
  // three chunks: nothing to fetch); its K DMA instructions go between the row steps of the part that has ISSUE set.  Each part
  // is ONE basic block: no run-time test in here.  The row registers live across the parts (the first row of part two is
  // requested under the last step of part one).
  static_assert(K <= 3 && CC == 2, "one DMA instruction per row step of a part");
  // DEPTH: how many row steps the source-row reads run ahead of the products (DEPTH + 1 register sets of three `ds_read_b128`).
  // One step covers the LDS latency when three waves share a SIMD (level 1); a 3-wave workgroup of the small-map instance has
  // a SIMD to itself per wave and registers to spare: two steps.
  static_assert(DEPTH == 1 || DEPTH == 2, "read-ahead of one or two row steps");
  f32x4 ra[CC], rb[DEPTH + 1][3];
  unsigned pa = 0, pb = 0;                             // my target-row / first-source-row byte addresses in the ring in use
  auto part = [&](int cur, int nxt, unsigned long long on, auto second, auto with_issue) {     // stage indices
    constexpr int S0 = decltype(second)::value ? 3 : 0;
    constexpr bool ISSUE = decltype(with_issue)::value;
    const unsigned nx = lds0 + (unsigned)nxt * RB;
    if constexpr (S0 == 0) {
      // (the ring's address passes through a volatile asm: otherwise the compiler keeps eight per-lane addresses, two per
      // ring, alive across the loop, and at 168 registers that is eight spills reloaded through `s_waitcnt vmcnt(0)`)
      unsigned sb = lds0 + (unsigned)cur * RB;
      asm volatile("" : "+s"(sb));
      pa = sb + (unsigned)a_off * 4u;
      pb = sb + (unsigned)b_off * 4u;
      ra[0] = lds_ld(pa);
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) ldrow(rb[d], pb, d);
      ra[1] = lds_ld(pa + (unsigned)(TH * PITCH * 4));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int st = S0; st < S0 + 3; ++st) {
      if (st + DEPTH < CC * 3) ldrow(rb[(st + DEPTH) % (DEPTH + 1)], pb, st + DEPTH);
      __builtin_amdgcn_sched_barrier(0);
      step(ra[st / 3], rb[st % (DEPTH + 1)], st % 3);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ISSUE) {
        if (st - S0 < K) {
          issue1(nx, st - S0, on);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };

  const int nchunks = Cw / CC;                         // multiple of NS (checked by the launcher)
  // chunk hand-off: my own DMA of the chunk has landed once at most NW_ newer instructions of mine are in flight (in-order
  // completion), the barrier extends that to every wave's and says that everybody is done with the ring about to be refilled
  auto handoff = [&](auto nwait) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(nwait)::value) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  static_assert((NS - 2) * (K - 1) <= 63, "vmcnt is a 6-bit count");
  const std::false_type first{}, no_issue{};
  const std::true_type second{}, do_issue{};
#pragma unroll
  for (int r = 0; r < NS - 1; ++r) issue(lds0 + r * RB);               // chunks 0 .. NS - 2
  // Zero padding: a slot without a source (outside the image, or the unused tail of a row) is never written by the DMA -- and
  // is the same slot in every chunk.  Each lane zeroes ITS slots of the stages once, AFTER the first chunks are on their way
  // (disjoint addresses): the first kernel zeroed all 120 KB and synchronised before its first DMA (~1 us).
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int wi = wave + k * NW;
    if (wi < NINSTR && !((gmask[k] >> lane) & 1)) {
#pragma unroll
      for (int r = 0; r < NS; ++r)
        *reinterpret_cast<float4*>(rings_all + (r * NTILE + half) * BUF + wi * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // visible to the others behind the first hand-off's barrier
  // barrier in FRONT of chunk c (chunk c landed; chunks c+1 .. c+NS-2 in flight), DMA of chunk c+NS-1 under its first part.
  // (Tried and dropped, profiles/r04_corr_pipe2.txt: the workgroup's second tile half a chunk out of phase -- the same barriers
  // in the MIDDLE of its chunks, so that one tile's waves multiply while the other's wait for the first rows of a new chunk:
  // 100 instead of 88 us; two instruction streams per CU cost more than the bubbles.)
  const std::integral_constant<int, (NS - 2) * (K - 1)> full_behind{};  // NS - 2 newer chunks, K or K - 1 instructions each
  for (int ck = 0; ck < nchunks - NS; ck += NS) {        // nchunks is a multiple of NS (checked by the launcher)
    static_for<0, NS>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      handoff(full_behind);
      part(r, (r + NS - 1) % NS, ~0ull, first, do_issue);
      part(r, (r + NS - 1) % NS, ~0ull, second, no_issue);
    });
  }
  // The tile's last NS chunks, with the waits THEIR queue needs: behind chunk n - NS + r only NS - 1 - r chunks are in flight -- a
  // constant count would let a wave read the last chunks before they have landed (the first 4-stage kernel did:
  // tools/micro/corr_race.py, 11 of 300 launches off by the last channels' products under memory load).
  static_for<0, NS>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    constexpr int behind = (r == 0 ? NS - 2 : NS - 1 - r) * (K - 1);
    handoff(std::integral_constant<int, behind>{});
    if constexpr (r == 0) part(0, NS - 1, ~0ull, first, do_issue);      // chunk n - 1 goes out under chunk n - NS
    else part(r, 0, 0ull, first, no_issue);
    part(r, 0, 0ull, second, no_issue);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (KSPLIT) {
    // the other wave groups' sums join the first's: [36][NT] floats per group and round (16 pairs + 4 singles of one vertical
    // shift), lane-contiguous (no bank conflict), in the rings' memory -- behind a barrier, because a slower wave may still be
    // reading its last chunk.  Group 0 adds groups 1, 2, 3 in that order (deterministic).
    const int t = (int)threadIdx.x - half * NT;
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (half != 0) {
        f32x2* red2 = reinterpret_cast<f32x2*>(rings_all + (half - 1) * 36 * NT);
        float* red1 = rings_all + (half - 1) * 36 * NT + 32 * NT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int p = 0; p < 4; ++p) red2[(i * 4 + p) * NT + t] = accp[a][i][p];
          red1[i * NT + t] = accs[a][i];
        }
      }
      __syncthreads();
      if (half == 0) {
#pragma unroll
        for (int hh = 1; hh < NTILE; ++hh) {
          const f32x2* red2 = reinterpret_cast<const f32x2*>(rings_all + (hh - 1) * 36 * NT);
          const float* red1 = rings_all + (hh - 1) * 36 * NT + 32 * NT;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int p = 0; p < 4; ++p) accp[a][i][p] += red2[(i * 4 + p) * NT + t];
            accs[a][i] += red1[i * NT + t];
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: ReLU + L2 norm over the 81 shifts on the register PAIRS (packed squares, the ReLU kept in the accumulators; the sum of squares is taken pair-wise, then across the three vertical-shift groups) ----
  auto get = [&](int dyi, int dx, int i) -> float {
    if (i & 1) return dx == 0 ? accs[dyi][i] : accp[dyi][i][(dx - 1) >> 1][(dx - 1) & 1];
    return dx == 8 ? accs[dyi][i] : accp[dyi][i][dx >> 1][dx & 1];
  };
  const bool img2 = edge && strip >= STRIPS / 2;
  const int h = h0 + row, wx = w0 + 4 * (img2 ? strip - STRIPS / 2 : strip);
  float scale[4] = {1.f, 1.f, 1.f, 1.f};
  if constexpr (FUSE) {
    f32x2 ssp[4];
    float ss[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ssp[i] = f32x2{0.f, 0.f};
      ss[i] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x2 v = __builtin_elementwise_max(accp[a][i][p], f32x2{0.f, 0.f});
          accp[a][i][p] = v;
          ssp[i] = __builtin_elementwise_fma(v, v, ssp[i]);
        }
        const float v = fmaxf(accs[a][i], 0.0f);
        accs[a][i] = v;
        ss[i] = fmaf(v, v, ss[i]);
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) ss[i] += ssp[i][0] + ssp[i][1];
    __syncthreads();
    float* red = rings_all + half * BUF;  // [3][TH][TW]: stage 0 of my tile
    *reinterpret_cast<float4*>(&red[(dyg * TH + row) * TW + 4 * strip]) = make_float4(ss[0], ss[1], ss[2], ss[3]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float tot = red[(0 * TH + row) * TW + 4 * strip + i] + red[(1 * TH + row) * TW + 4 * strip + i] +
                        red[(2 * TH + row) * TW + 4 * strip + i];
      scale[i] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
    }
  }
  if (live && (!KSPLIT || half == 0) && h < H && wx + 3 < W) {
    float* o = out + ((size_t)(n + (img2 ? 1 : 0)) * 81 + (size_t)(dyg * 3) * 9) * plane + (size_t)h * W + wx;
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi)
#pragma unroll
      for (int dx = 0; dx < 9; ++dx) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v val = {get(dyi, dx, 0) * scale[0], get(dyi, dx, 1) * scale[1], get(dyi, dx, 2) * scale[2], get(dyi, dx, 3) * scale[3]};
        *reinterpret_cast<f4v*>(o) = val;               // (non-temporal stores: measured, no change -- profiles/r06_corr_nt_store_ab.txt)
        o += plane;                                    // (a running pointer: one 64-bit add per store instead of a 64-bit multiply-add)
      }
  }
  if constexpr (JOIN != 0) {
    static_assert(!FUSE && !PAIR && KSPLIT, "JOIN: raw partial sums of one tile per workgroup");
    constexpr int PX = TH * TW, NTH = NT * NTILE;
    static_assert((81 * PX + PX) <= NS * NTILE * BUF, "JOIN: the tile's volume fits the rings");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // every wave's slab stores have left; the rings are free
    int* flag = reinterpret_cast<int*>(rings_all);
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned prev = __hip_atomic_fetch_add(&tickets[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = prev == gridDim.y - 1;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // nobody else touches this ticket in this launch: back to zero for the next one (the caller zeroes the tickets ONCE)
        __hip_atomic_store(&tickets[tile], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      flag[0] = last;
    }
    __syncthreads();
    const int last = flag[0];
    __syncthreads();
    if (!last) return;
    float* acc = rings_all;                            // [81][PX]
    float* scl = rings_all + 81 * PX;                  // [PX]
    const int S = (int)gridDim.y;
    const float* pbase = slab0 + (size_t)n * 81 * plane;
    for (int idx = threadIdx.x; idx < 81 * (PX / 4); idx += NTH) {
      const int d = idx / (PX / 4), sp = idx % (PX / 4), rr = sp / STRIPS, st = sp % STRIPS;
      const int hh = h0 + rr, ww = w0 + 4 * st;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hh < H && ww < W) {
        const float* q = pbase + (size_t)d * plane + (size_t)hh * W + ww;
        v = *reinterpret_cast<const float4*>(q);
#pragma unroll 4
        for (int sl = 1; sl < S; ++sl) {               // slice order: the same sum for every launch geometry
          const float4 t = *reinterpret_cast<const float4*>(q + (size_t)sl * part_stride);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
      }
      if (JOIN == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *reinterpret_cast<float4*>(&acc[d * PX + 4 * sp]) = v;
    }
    __syncthreads();
    if (JOIN == 2) {
      for (int px = threadIdx.x; px < PX; px += NTH) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {                  // the three vertical-shift groups of the one-kernel epilogue, in its order
          float ss = 0.f;
#pragma unroll
          for (int d = 0; d < 27; ++d) {
            const float r = acc[(k * 27 + d) * PX + px];
            ss = fmaf(r, r, ss);
          }
          tot = k == 0 ? ss : tot + ss;
        }
        scl[px] = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
      }
      __syncthreads();
    }
    float* obase = joined + (size_t)n * 81 * plane;
    for (int idx = threadIdx.x; idx < 81 * (PX / 4); idx += NTH) {
      const int d = idx / (PX / 4), sp = idx % (PX / 4), rr = sp / STRIPS, st = sp % STRIPS;
      const int hh = h0 + rr, ww = w0 + 4 * st;
      if (hh < H && ww < W) {
        float4 v = *reinterpret_cast<const float4*>(&acc[d * PX + 4 * sp]);
        if (JOIN == 2) {
          const float4 sc = *reinterpret_cast<const float4*>(&scl[4 * sp]);
          v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
        }
        *reinterpret_cast<float4*>(obase + (size_t)d * plane + (size_t)hh * W + ww) = v;
      }
    }
  }
}

// (The fp32-matrix-pipe formulation of this forward -- exact, measured slower: 166-171 vs 139 us, profiles/r02_corr_mfma_*.txt,
// r03_corr_f16_ablation.txt -- lives in tools/experiments/matrix_pipe_corr/, outside the product library.)

template <bool FUSE, bool WARP>
static int launch_corr9(const float* in1, const float* in2, const float* flow, float* out, int B, int C, int H,
                        int W, hipStream_t st) {
  if constexpr (!WARP) {
    // The pipelined kernel: W % 4 == 0 (16-byte DMA pieces), C a multiple of the ring's NS x 2 = 8 channels (16 with the
    // channel split), and -- its DMA offsets are 32-bit within a sample (a sample PAIR for edge tiles) -- samples below 4 GB.
    // Anything else takes the register-staged kernel below.
    const size_t sample = (size_t)C * H * W * sizeof(float);
    if ((W & 3) == 0 && (C % 8) == 0 && C >= 16 && sample < (1ull << 32)) {
      const int xcd = 1;      // XCD-local tile order: the halo rows two neighbours fetch meet in one L2 (DMA alone 65 -> 46 us)
      // K4 level 1 (2 x 270 x 480) and larger: two independent 16 x 32 tiles per 12-wave workgroup, 3 waves per SIMD
      const long pairs16 = ((long)B * cdiv(W, 32) * cdiv(H, 16) + 1) / 2;
      if (pairs16 >= 192) {
        const int tilesX = cdiv(W, 32), tilesY = cdiv(H, 16);
        const long ntiles = (long)B * tilesX * tilesY, blocks = (ntiles + 1) / 2;
        if (ntiles > 0x7fffffffL) return fail(RFN_EINVAL, "corr9: grid too large");
        hipLaunchKernelGGL((corr9_pipe2_kernel<16, 32, FUSE, 3, 2, 1, 4>), dim3((unsigned)blocks), dim3(16 * 8 * 3 * 2), 0, st,
                           in1, in2, out, C, H, W, tilesX, tilesY, (int)ntiles, xcd, 0, C, 0L, nullptr, nullptr);
        return check_launch("corr9_pipe2_kernel");
      }
      // Smaller maps: single 8 x 32 tiles, 3-wave workgroups.  Up to 256 of them (one workgroup per CU) the workgroup's second
      // wave group takes the second half of the tile's channels (KSPLIT: 136 tiles 59 -> 47 us, 256 tiles 64.5 -> 54.7 us,
      // K2 level 1 33.8 -> 28.6 us; profiles/r04_corr_ksplit.txt).  A map that needs MORE than 256 because its width leaves
      // half a tile column over (K4 level 2: 2 x 135 x 240 = 272 tiles) shares that column band between image pairs (PAIR)
      // when that brings it to <= 256: 255 tiles there.
      const int tilesY = cdiv(H, 8);
      const long nt8 = (long)B * cdiv(W, 32) * tilesY;
      const bool ksplit_ok = C >= 32 && C % 16 == 0;
      const int rem = W % 32;
      const long nreg = (long)B * (W / 32) * tilesY, npair = nreg + (long)(B / 2) * tilesY;
      // (round 5) FOUR wave groups per tile when the channels allow it (C % 32 == 0: whole ring rounds per group): twelve waves per
      // CU walk chunks at 9.2 wave-chunks / us against 7.1 with six (level 1 runs twelve) -- K4 level 2 64 -> ~50 us
      const bool ksplit4 = ksplit_ok && C >= 64 && C % 32 == 0;
      if (ksplit_ok && nt8 > 256 && rem > 0 && rem <= 16 && B % 2 == 0 && npair <= 256 && 2 * sample < (1ull << 32)) {
        if (ksplit4)
          hipLaunchKernelGGL((corr9_pipe2_kernel<8, 32, FUSE, 3, 4, 1, 4, true, true>), dim3((unsigned)npair), dim3(8 * 8 * 3 * 4), 0,
                             st, in1, in2, out, C, H, W, W / 32, tilesY, (int)npair, xcd, (int)nreg, C, 0L, nullptr, nullptr);
        else
          hipLaunchKernelGGL((corr9_pipe2_kernel<8, 32, FUSE, 3, 2, 1, 4, true, true>), dim3((unsigned)npair), dim3(8 * 8 * 3 * 2), 0,
                             st, in1, in2, out, C, H, W, W / 32, tilesY, (int)npair, xcd, (int)nreg, C, 0L, nullptr, nullptr);
        return check_launch("corr9_pipe2_kernel");
      }
      if (nt8 > 0x7fffffffL) return fail(RFN_EINVAL, "corr9: grid too large");
      if (ksplit_ok && nt8 <= 256) {
        if (ksplit4)
          hipLaunchKernelGGL((corr9_pipe2_kernel<8, 32, FUSE, 3, 4, 1, 4, true>), dim3((unsigned)nt8), dim3(8 * 8 * 3 * 4), 0, st,
                             in1, in2, out, C, H, W, cdiv(W, 32), tilesY, (int)nt8, xcd, 0, C, 0L, nullptr, nullptr);
        else
          hipLaunchKernelGGL((corr9_pipe2_kernel<8, 32, FUSE, 3, 2, 1, 4, true>), dim3((unsigned)nt8), dim3(8 * 8 * 3 * 2), 0, st,
                             in1, in2, out, C, H, W, cdiv(W, 32), tilesY, (int)nt8, xcd, 0, C, 0L, nullptr, nullptr);
        return check_launch("corr9_pipe2_kernel");
      }
      hipLaunchKernelGGL((corr9_pipe2_kernel<8, 32, FUSE, 3, 1, 1, 4>), dim3((unsigned)nt8), dim3(8 * 8 * 3), 0, st, in1, in2,
                         out, C, H, W, cdiv(W, 32), tilesY, (int)nt8, xcd, 0, C, 0L, nullptr, nullptr);
      return check_launch("corr9_pipe2_kernel");
    }
  }
  constexpr int TH = 8, CC = 8;
  const int tilesX = cdiv(W, kTW), tilesY = cdiv(H, TH);
  const long blocks = (long)B * tilesX * tilesY;
  if (blocks <= 0 || blocks > 0x7fffffffL) return fail(RFN_EINVAL, "corr9: grid too large");
  hipLaunchKernelGGL((corr9_tile_kernel<TH, CC, FUSE, WARP>), dim3((unsigned)blocks), dim3(TH * kStrips * 3), 0,
                     st, in1, in2, flow, out, C, H, W, tilesX, tilesY);
  return check_launch("corr9_tile_kernel");
}

// --------------------------------------------------------------------------------------------------------
// Channel split for small maps.  The tiled kernels walk the C channels of a tile serially: a 2 x 256 x 32 x 32 level
// is 8 tiles = 8 workgroups on 256 CUs and takes the same ~100 us as the 2 x 128 x 270 x 480 level (pure latency:
// 64 chunk hand-offs of ~1.5 us).  Here S workgroups share a tile, each over C / S channels, writing raw partial sums
// to a workspace (S, B, 81, H, W); the reduce kernel adds them in chunk order (deterministic) and applies the fused
// ReLU + L2-norm epilogue of the one-kernel path.
// --------------------------------------------------------------------------------------------------------
// 32 pixels x 8 shift lanes per workgroup: lane (px, k) adds the S partials of shifts k, k+8, ... (coalesced over the
// pixels) into an LDS tile [81][32]; three lanes per pixel then take the squares of the three vertical-shift groups in
// the one-kernel epilogue's order, and everybody scales and stores.
template <bool FUSE>
__global__ __launch_bounds__(256) void corr9_split_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                 int S, long part_stride, long plane, long total) {
  __shared__ float tile[81][33];
  __shared__ float ssq[3][32];
  const int px = threadIdx.x & 31, k = threadIdx.x >> 5;
  const long idx = (long)blockIdx.x * 32 + px;                   // (n, pixel)
  const bool ok = idx < total;
  const long n = ok ? idx / plane : 0, pix = ok ? idx % plane : 0;
  const float* p = part + n * 81 * plane + pix;
  for (int d = k; d < 81; d += 8) {
    float v = 0.0f;
    if (ok) {
      // chunk order, four loads in flight at a time (the adds stay sequential: same sum for every launch geometry)
      const float* q = p + (long)d * plane;
      v = q[0];
      int s = 1;
      for (; s + 3 < S; s += 4) {
        const float t0 = q[(long)s * part_stride], t1 = q[(long)(s + 1) * part_stride];
        const float t2 = q[(long)(s + 2) * part_stride], t3 = q[(long)(s + 3) * part_stride];
        v = (((v + t0) + t1) + t2) + t3;
      }
      for (; s < S; ++s) v += q[(long)s * part_stride];
    }
    tile[d][px] = FUSE ? fmaxf(v, 0.0f) : v;
  }
  __syncthreads();
  float scale = 1.0f;
  if constexpr (FUSE) {
    if (k < 3) {
      float ss = 0.0f;
#pragma unroll
      for (int d = 0; d < 27; ++d) {
        const float r = tile[k * 27 + d][px];
        ss = fmaf(r, r, ss);
      }
      ssq[k][px] = ss;
    }
    __syncthreads();
    scale = 1.0f / fmaxf(sqrtf(ssq[0][px] + ssq[1][px] + ssq[2][px]), 1e-12f);
  }
  if (ok) {
    float* o = out + n * 81 * plane + pix;
    for (int d = k; d < 81; d += 8) o[(long)d * plane] = tile[d][px] * scale;
  }
}

template <bool FUSE>
static int launch_corr9_split(const float* in1, const float* in2, float* out, float* workspace, int B, int C, int H, int W,
                              int S, hipStream_t st) {
  constexpr int TH = 8, TW = 32;
  const int Cc = C / S;
  const int tilesX = cdiv(W, TW), tilesY = cdiv(H, TH);
  const long ntiles = (long)B * tilesX * tilesY;
  if (ntiles <= 0 || ntiles > 0x7fffffffL) return fail(RFN_EINVAL, "corr9 split: grid too large");
  if ((size_t)C * H * W * sizeof(float) >= (1ull << 32)) return fail(RFN_EINVAL, "corr9 split: samples of 4 GB and more are not tiny maps");
  const long plane = (long)H * W, part_stride = (long)B * 81 * plane;
  if (Cc >= 64 && Cc % 32 == 0 && (W & 3) == 0) {
    // ONE launch: four wave groups per workgroup on the quarters of its slice, the tile's last workgroup joins the slices.
    // Tickets: the ntiles unsigned ints behind the partial volumes (rfn_local_corr_layer_split_workspace_bytes).
    unsigned* tickets = reinterpret_cast<unsigned*>(workspace + (size_t)S * part_stride);
    hipLaunchKernelGGL((corr9_pipe2_kernel<TH, TW, false, 3, 4, 1, 4, true, false, FUSE ? 2 : 1>), dim3((unsigned)ntiles, (unsigned)S),
                       dim3(TH * (TW / 4) * 3 * 4), 0, st, in1, in2, workspace, Cc, H, W, tilesX, tilesY, (int)ntiles, 0, 0, C,
                       part_stride, out, tickets);
    return check_launch("corr9_pipe2_kernel (channel split, joined in the launch)");
  }
  // slice blockIdx.y of the channels per workgroup (Cc % 8 == 0: whole rounds of the 4-stage ring), raw partial volumes
  hipLaunchKernelGGL((corr9_pipe2_kernel<TH, TW, false, 3, 1, 1, 4>), dim3((unsigned)ntiles, (unsigned)S), dim3(TH * (TW / 4) * 3),
                     0, st, in1, in2, workspace, Cc, H, W, tilesX, tilesY, (int)ntiles, 0, 0, C, part_stride, nullptr, nullptr);
  if (int rc = check_launch("corr9_pipe2_kernel (channel split)")) return rc;
  const long total = (long)B * plane;
  hipLaunchKernelGGL((corr9_split_reduce_kernel<FUSE>), dim3(cdiv(total, 32)), dim3(256), 0, st, workspace, out, S,
                     part_stride, plane, total);
  return check_launch("corr9_split_reduce_kernel");
}

// --------------------------------------------------------------------------------------------------------
// Generic forward: any kernel/patch/stride/pad/dilation, one thread per output element (coalesced along w).
// --------------------------------------------------------------------------------------------------------
struct CorrParams {
  int B, C, iH, iW, oH, oW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW;
};

// accumulation type: the element type itself for float / double (the CPU reference's `scalar_t` sums, bit for bit), fp32
// for half (the CUDA reference dispatches half too, correlation_cuda_kernel.cu:267; its sums are warp-shuffled partials in
// half -- here every product is formed and summed in fp32 and the result is rounded once)
template <typename T> struct CorrAcc { using type = T; };
template <> struct CorrAcc<__half> { using type = float; };

template <typename T>
__global__ __launch_bounds__(256) void corr_generic_fwd_kernel(const T* __restrict__ in1,
                                                               const T* __restrict__ in2, T* __restrict__ out,
                                                               CorrParams p, long total) {
  using A = typename CorrAcc<T>::type;
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int w = t % p.oW; t /= p.oW;
    const int h = t % p.oH; t /= p.oH;
    const int pw = t % p.patchW; t /= p.patchW;
    const int ph = t % p.patchH;
    const int n = t / p.patchH;
    const int u = -p.padH + h * p.dH, v = -p.padW + w * p.dW;
    const int sU = (ph - radH) * p.dpH, sV = (pw - radW) * p.dpW;
    const size_t plane = (size_t)p.iH * p.iW;
    const T* a = in1 + (size_t)n * p.C * plane;
    const T* b = in2 + (size_t)n * p.C * plane;
    A acc = 0;
    for (int c = 0; c < p.C; ++c) {
      for (int i = 0; i < p.kH; ++i) {
        const int i1 = u + i * p.dilH, i2 = i1 + sU;
        if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
        for (int jj = 0; jj < p.kW; ++jj) {
          const int j1 = v + jj * p.dilW, j2 = j1 + sV;
          if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
          acc += (A)a[c * plane + (size_t)i1 * p.iW + j1] * (A)b[c * plane + (size_t)i2 * p.iW + j2];
        }
      }
    }
    out[idx] = (T)acc;
  }
}

// --------------------------------------------------------------------------------------------------------
// Backward, gather form for kernel 1 / stride 1 / pad 0 (deterministic): one thread per (n,c,h,w).
//   g1[n,c,h,w] = sum_{ph,pw} gout[n,ph,pw,h,w]       * in2[n,c,h+sU,w+sV]
//   g2[n,c,y,x] = sum_{ph,pw} gout[n,ph,pw,y-sU,x-sV] * in1[n,c,y-sU,x-sV]
// --------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void corr_k1_bwd_kernel(const T* __restrict__ in1, const T* __restrict__ in2,
                                                          const T* __restrict__ gout, T* __restrict__ g1,
                                                          T* __restrict__ g2, CorrParams p, long total) {
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  const size_t plane = (size_t)p.iH * p.iW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int w = t % p.iW; t /= p.iW;
    const int h = t % p.iH; t /= p.iH;
    const int c = t % p.C;
    const int n = t / p.C;
    const T* a = in1 + ((size_t)n * p.C + c) * plane;
    const T* b = in2 + ((size_t)n * p.C + c) * plane;
    const T* go = gout + (size_t)n * p.patchH * p.patchW * plane;
    T s1 = 0, s2 = 0;
    for (int ph = 0; ph < p.patchH; ++ph) {
      const int sU = (ph - radH) * p.dpH;
      for (int pw = 0; pw < p.patchW; ++pw) {
        const int sV = (pw - radW) * p.dpW;
        const T* gp = go + (size_t)(ph * p.patchW + pw) * plane;
        const int y2 = h + sU, x2 = w + sV;
        if (y2 >= 0 && y2 < p.iH && x2 >= 0 && x2 < p.iW) s1 += gp[(size_t)h * p.iW + w] * b[(size_t)y2 * p.iW + x2];
        const int y1 = h - sU, x1 = w - sV;
        if (y1 >= 0 && y1 < p.iH && x1 >= 0 && x1 < p.iW)
          s2 += gp[(size_t)y1 * p.iW + x1] * a[(size_t)y1 * p.iW + x1];
      }
    }
    g1[idx] = s1;
    g2[idx] = s2;
  }
}

// --------------------------------------------------------------------------------------------------------
// Backward for the hot parameterisation (patch 9, kernel 1, stride 1, pad 0), f32: LDS-tiled gather.
//   MODE 1: g1[n,c,q] = sum_d gout[n,d,q]     * in2[n,c,q+d]      (weights at q, taps forward)
//   MODE 2: g2[n,c,q] = sum_d gout[n,d,q-d]   * in1[n,c,q-d]      (weights and taps mirrored)
// The 81 weights of an output pixel do not depend on the channel: a thread owns ONE pixel, gathers its 81 weights into
// registers once and then walks the channels; per channel chunk the (8+8) x (32+8) halo tile of the other input is
// staged in LDS and every FMA costs one conflict-free ds_read_b32 (lanes = consecutive pixels).  Deterministic.
// --------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void corr9_bwd_tile_kernel(const float* __restrict__ other,
                                                             const float* __restrict__ gout,
                                                             float* __restrict__ grad, int C, int H, int W,
                                                             int tilesX, int tilesY) {
  constexpr int TH = 8, TW = 32, CC = 8, RH = TH + 8, RW = TW + 8;
  __shared__ float tile[CC][RH][RW + 1];
  int bid = blockIdx.x;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int h0 = ty * TH, w0 = tx * TW;
  const int lx = threadIdx.x % TW, ly = threadIdx.x / TW;
  const int qy = h0 + ly, qx = w0 + lx;
  const bool inq = qy < H && qx < W;
  const size_t plane = (size_t)H * W;
  // my 81 weights
  float wgt[81];
  const float* go = gout + (size_t)n * 81 * plane;
#pragma unroll
  for (int dy = 0; dy < 9; ++dy)
#pragma unroll
    for (int dx = 0; dx < 9; ++dx) {
      const int d = dy * 9 + dx;
      float v = 0.0f;
      if (MODE == 1) {
        if (inq) v = go[(size_t)d * plane + (size_t)qy * W + qx];
      } else {
        const int py = qy - (dy - 4), px = qx - (dx - 4);
        if (inq && py >= 0 && py < H && px >= 0 && px < W) v = go[(size_t)d * plane + (size_t)py * W + px];
      }
      wgt[d] = v;
    }
  const float* src = other + (size_t)n * C * plane;
  float* dst = grad + (size_t)n * C * plane;
  for (int c0 = 0; c0 < C; c0 += CC) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < CC * RH * RW; idx += 256) {
      const int x = idx % RW, r = (idx / RW) % RH, c = idx / (RW * RH);
      const int gy = h0 - 4 + r, gx = w0 - 4 + x;
      float v = 0.0f;
      if (c0 + c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) v = src[(size_t)(c0 + c) * plane + (size_t)gy * W + gx];
      tile[c][r][x] = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int c = 0; c < CC; ++c) {
      float acc = 0.0f;
#pragma unroll
      for (int dy = 0; dy < 9; ++dy)
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          // tile origin is (h0-4, w0-4): pixel q sits at [ly+4][lx+4]; MODE 1 taps q+d, MODE 2 taps q-d
          const int r = (MODE == 1) ? ly + dy : ly + 8 - dy;
          const int x = (MODE == 1) ? lx + dx : lx + 8 - dx;
          acc = fmaf(wgt[dy * 9 + dx], tile[c][r][x], acc);
        }
      if (inq && c0 + c < C) dst[(size_t)(c0 + c) * plane + (size_t)qy * W + qx] = acc;
    }
  }
}

// Round 5: the same gather on register STRIPS, laid out like the forward.  A thread owns 4 adjacent pixels of a row and the
// weights of THREE vertical shifts (27 shifts x 4 pixels = 108 registers); the three wave groups of a 192-thread workgroup split
// the nine vertical shifts of an 8 x 32 tile.  Per channel and vertical shift a thread reads its 12 row values as three
// ds_read_b128 and does 36 FMAs (the one-pixel kernel above: one ds_read_b32 per FMA -- LDS-bound); the groups' partial sums of
// a chunk of 8 channels meet in LDS (group 0 adds groups 1, 2 in that order: deterministic) and leave as 16-byte stores.  The
// next chunk's halo tile is in flight to registers while a chunk is computed (double-buffered LDS, ONE barrier per chunk).
// Both gradients in ONE launch (blockIdx.y).  W % 4 == 0, C % 8 == 0; anything else takes the one-pixel kernel.
// K4 level 1 (2 x 128 x 270 x 480, both gradients): 997 us (two launches of the one-pixel kernel) -> 512 (strips, plain float
// loop) -> 324 (aligned register pairs) -> 308 us (conflict-free lane order) = 0.25 of 8 TB/s on 615 MB; matcher training step
// 100 -> 91-93 ms (profiles/r05_corr_backward.txt).
template <int MODE>
__device__ __forceinline__ void corr9_bwd_strip_body(const float* __restrict__ other, const float* __restrict__ gout,
                                                     float* __restrict__ grad, int C, int H, int W, int tilesX, int tilesY,
                                                     float* lds) {
  constexpr int TH = 8, TW = 32, CC = 8, RH = TH + 8, P4 = (TW + 8) / 4, NT = 192;
  constexpr int CHUNK4 = CC * RH * P4;                   // float4 pieces of a chunk's halo tile
  constexpr int NPF = (CHUNK4 + NT - 1) / NT;            // pieces a thread carries
  float4* tile = reinterpret_cast<float4*>(lds);                                  // [2][CC][RH][P4]
  constexpr int RP = 10;                                  // slots per row of the exchange buffer (same banking argument)
  float4* red = reinterpret_cast<float4*>(lds) + 2 * CHUNK4;                      // [2][2][CC][TH][RP]
  const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6;
  // lane -> (row, strip): ds_read_b128 is serviced in four groups of 16 lanes (quads {0,3,5,6}, {1,2,4,7} of each half wave),
  // and a group must touch 16 distinct 16-byte slots of the 256-byte bank row.  With rows of 10 slots the half rows start at
  // slot (10 r + 4 h) mod 16; the quads of a group get {(0,0),(0,1),(2,1),(4,1)}, {(2,0),(4,0),(6,0),(6,1)} and the same on the
  // odd rows -- start slots {0,4,8,12} + const for every vertical shift, channel and read of a row: conflict-free.
  const int quad = lane >> 2, q7 = quad & 7;
  const int ly = 2 * ((0x32130210u >> (4 * q7)) & 3) + (quad >> 3);
  const int strip = 4 * ((0xE8u >> q7) & 1) + (lane & 3);
  int bid = blockIdx.x;
  {                                                      // XCD-local tile order (neighbours share halo rows in one L2)
    const int nwg = gridDim.x, qq = nwg / 8, rr = nwg % 8, xcd = bid % 8, loc = bid / 8;
    bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
  }
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int h0 = ty * TH, w0 = tx * TW;
  const int qy = h0 + ly, qx0 = w0 + 4 * strip;
  const bool inq = qy < H && qx0 < W;                    // W % 4 == 0: a strip is all in or all out
  const size_t plane = (size_t)H * W;
  const float* go = gout + (size_t)n * 81 * plane;
  // Weights as REGISTER PAIRS for v_pk_fma_f32 (64-bit operands are even-aligned register pairs): a tap at an EVEN offset into the
  // 12 row values pairs pixels (0, 1) and (2, 3) with the natural halves of the three 16-byte reads; at an ODD offset pixels
  // (1, 2) pair up and pixels 0 and 3 go as single FMAs -- 22 VALU instructions per 36 FMAs, no register shuffling, no
  // unaligned LDS reads (the compiler's own pairing of a plain float loop: ~19 moves + 8 small LDS reads per row).
  f32x2 we[3][5][2];                                     // even horizontal shifts 0, 2, ..., 8: pixels (0, 1), (2, 3)
  f32x2 wo[3][4];                                        // odd shifts 1, 3, 5, 7: pixels (1, 2)
  float wo0[3][4], wo3[3][4];                            //                        pixels 0 and 3
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int dy = 3 * grp + a;
#pragma unroll
    for (int dx = 0; dx < 9; ++dx) {
      const int d = dy * 9 + dx;
      float wv[4];
      if (MODE == 1) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inq) v = *reinterpret_cast<const float4*>(go + (size_t)d * plane + (size_t)qy * W + qx0);
        wv[0] = v.x; wv[1] = v.y; wv[2] = v.z; wv[3] = v.w;
      } else {
        const int py = qy - (dy - 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int px = qx0 + i - (dx - 4);
          wv[i] = (inq && py >= 0 && py < H && px >= 0 && px < W) ? go[(size_t)d * plane + (size_t)py * W + px] : 0.0f;
        }
      }
      if (dx % 2 == 0) {
        we[a][dx / 2][0] = f32x2{wv[0], wv[1]};
        we[a][dx / 2][1] = f32x2{wv[2], wv[3]};
      } else {
        wo[a][dx / 2] = f32x2{wv[1], wv[2]};
        wo0[a][dx / 2] = wv[0];
        wo3[a][dx / 2] = wv[3];
      }
    }
  }
  const float* src = other + (size_t)n * C * plane;
  float* dst = grad + (size_t)n * C * plane;
  // this thread's pieces of a chunk's halo tile: (channel, row, 16-byte column) -- the same for every chunk
  int poff[NPF];                                         // offset of the piece within a channel chunk, or -1 (outside the image)
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const int idx = tid + NT * k;
    const int x4 = idx % P4, r = (idx / P4) % RH, c = idx / (P4 * RH);
    const int gy = h0 - 4 + r, gx = w0 - 4 + 4 * x4;
    poff[k] = (idx < CHUNK4 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (int)((size_t)c * plane + (size_t)gy * W + gx) : -1;
  }
  float4 pf[NPF];
  auto fetch = [&](int c0) {
    const float* s0 = src + (size_t)c0 * plane;
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      pf[k] = poff[k] >= 0 ? *reinterpret_cast<const float4*>(s0 + poff[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NPF; ++k)
      if (tid + NT * k < CHUNK4) tile[buf * CHUNK4 + tid + NT * k] = pf[k];
  };
  fetch(0);
  stage(0);
  __syncthreads();
  const int nchunks = C / CC;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) fetch((kc + 1) * CC);
    float4 part[CC];
    const float4* tb = tile + buf * CHUNK4;
#pragma unroll
    for (int c = 0; c < CC; ++c) {
      f32x2 e01 = {0.f, 0.f}, e23 = {0.f, 0.f}, o12 = {0.f, 0.f};
      float o0 = 0.f, o3 = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int dy = 3 * grp + a;
        const int r = (MODE == 1) ? ly + dy : ly + 8 - dy;
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f* rowp = reinterpret_cast<const v4f*>(tb + (c * RH + r) * P4 + strip);
        const v4f q0 = rowp[0], q1 = rowp[1], q2 = rowp[2];
        const f32x2 v2[6] = {f32x2{q0[0], q0[1]}, f32x2{q0[2], q0[3]}, f32x2{q1[0], q1[1]},
                             f32x2{q1[2], q1[3]}, f32x2{q2[0], q2[1]}, f32x2{q2[2], q2[3]}};
        const float v1[12] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3], q2[0], q2[1], q2[2], q2[3]};
#pragma unroll
        for (int dx = 0; dx < 9; ++dx) {
          const int o = (MODE == 1) ? dx : 8 - dx;       // pixel i reads row value i + o (same parity as dx)
          if (dx % 2 == 0) {
            e01 = __builtin_elementwise_fma(we[a][dx / 2][0], v2[o / 2], e01);
            e23 = __builtin_elementwise_fma(we[a][dx / 2][1], v2[o / 2 + 1], e23);
          } else {
            o12 = __builtin_elementwise_fma(wo[a][dx / 2], v2[(o + 1) / 2], o12);
            o0 = fmaf(wo0[a][dx / 2], v1[o], o0);
            o3 = fmaf(wo3[a][dx / 2], v1[o + 3], o3);
          }
        }
      }
      part[c] = make_float4(e01[0] + o0, e01[1] + o12[0], e23[0] + o12[1], e23[1] + o3);
    }
    float4* rb = red + buf * (2 * CC * TH * RP);
    if (grp != 0) {
#pragma unroll
      for (int c = 0; c < CC; ++c) rb[((grp - 1) * CC + c) * (TH * RP) + ly * RP + strip] = part[c];
    }
    if (kc + 1 < nchunks) stage(buf ^ 1);
    __syncthreads();
    if (grp == 0 && inq) {
      float* d0 = dst + (size_t)(kc * CC) * plane + (size_t)qy * W + qx0;
#pragma unroll
      for (int c = 0; c < CC; ++c) {
        const float4 p1 = rb[(0 * CC + c) * (TH * RP) + ly * RP + strip];
        const float4 p2 = rb[(1 * CC + c) * (TH * RP) + ly * RP + strip];
        float4 o;
        o.x = (part[c].x + p1.x) + p2.x;
        o.y = (part[c].y + p1.y) + p2.y;
        o.z = (part[c].z + p1.z) + p2.z;
        o.w = (part[c].w + p1.w) + p2.w;
        *reinterpret_cast<float4*>(d0 + (size_t)c * plane) = o;
      }
    }
  }
}

__global__ __launch_bounds__(192) void corr9_bwd_strip_kernel(const float* __restrict__ in1, const float* __restrict__ in2,
                                                              const float* __restrict__ gout, float* __restrict__ g1,
                                                              float* __restrict__ g2, int C, int H, int W, int tilesX,
                                                              int tilesY) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 8 * 16 * 40 + 2 * 2 * 8 * 8 * 40];
  if (blockIdx.y == 0) corr9_bwd_strip_body<1>(in2, gout, g1, C, H, W, tilesX, tilesY, lds);
  else corr9_bwd_strip_body<2>(in1, gout, g2, C, H, W, tilesX, tilesY, lds);
}

// Generic backward: scatter with hardware atomics, one thread per gout element (grads pre-zeroed by the caller).
template <typename T>
__global__ __launch_bounds__(256) void corr_generic_bwd_kernel(const T* __restrict__ in1,
                                                               const T* __restrict__ in2,
                                                               const T* __restrict__ gout, T* __restrict__ g1,
                                                               T* __restrict__ g2, CorrParams p, long total) {
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  const size_t plane = (size_t)p.iH * p.iW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int w = t % p.oW; t /= p.oW;
    const int h = t % p.oH; t /= p.oH;
    const int pw = t % p.patchW; t /= p.patchW;
    const int ph = t % p.patchH;
    const int n = t / p.patchH;
    const T g = gout[idx];
    const int u = -p.padH + h * p.dH, v = -p.padW + w * p.dW;
    const int sU = (ph - radH) * p.dpH, sV = (pw - radW) * p.dpW;
    const T* a = in1 + (size_t)n * p.C * plane;
    const T* b = in2 + (size_t)n * p.C * plane;
    T* ga = g1 + (size_t)n * p.C * plane;
    T* gb = g2 + (size_t)n * p.C * plane;
    for (int i = 0; i < p.kH; ++i) {
      const int i1 = u + i * p.dilH, i2 = i1 + sU;
      if (i1 < 0 || i1 >= p.iH || i2 < 0 || i2 >= p.iH) continue;
      for (int jj = 0; jj < p.kW; ++jj) {
        const int j1 = v + jj * p.dilW, j2 = j1 + sV;
        if (j1 < 0 || j1 >= p.iW || j2 < 0 || j2 >= p.iW) continue;
        const size_t o1 = (size_t)i1 * p.iW + j1, o2 = (size_t)i2 * p.iW + j2;
        for (int c = 0; c < p.C; ++c) {
          atomicAdd(&gb[c * plane + o2], g * a[c * plane + o1]);
          atomicAdd(&ga[c * plane + o1], g * b[c * plane + o2]);
        }
      }
    }
  }
}

// Generic backward, GATHER form (deterministic, any parameterisation): one thread per (n, c, y, x) collects, for every
// patch offset and kernel tap, the output position that touched this pixel -- h = (y + pad - i dil) / stride when that
// division is exact and in range (correlation.cpp:131-183 read from the gradient's side).  Sums in CorrAcc<T>, one rounding.
// Used for half, where the scatter kernel above would need 16-bit atomics and round after every add.
template <typename T>
__global__ __launch_bounds__(256) void corr_generic_bwd_gather_kernel(const T* __restrict__ in1, const T* __restrict__ in2,
                                                                      const T* __restrict__ gout, T* __restrict__ g1,
                                                                      T* __restrict__ g2, CorrParams p, long total) {
  using A = typename CorrAcc<T>::type;
  const int radH = (p.patchH - 1) / 2, radW = (p.patchW - 1) / 2;
  const size_t plane = (size_t)p.iH * p.iW, oplane = (size_t)p.oH * p.oW;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    long t = idx;
    const int x = t % p.iW; t /= p.iW;
    const int y = t % p.iH; t /= p.iH;
    const int c = t % p.C;
    const int n = t / p.C;
    const T* a = in1 + ((size_t)n * p.C + c) * plane;
    const T* b = in2 + ((size_t)n * p.C + c) * plane;
    const T* go = gout + (size_t)n * p.patchH * p.patchW * oplane;
    A s1 = 0, s2 = 0;
    for (int ph = 0; ph < p.patchH; ++ph) {
      const int sU = (ph - radH) * p.dpH;
      for (int pw = 0; pw < p.patchW; ++pw) {
        const int sV = (pw - radW) * p.dpW;
        const T* gp = go + (size_t)(ph * p.patchW + pw) * oplane;
        for (int i = 0; i < p.kH; ++i) {
          // g1: this pixel is input1's tap (i1 = y); its partner in input2 is (y + sU)
          const int hn = y + p.padH - i * p.dilH;
          const bool hok = hn >= 0 && hn % p.dH == 0 && hn / p.dH < p.oH;
          // g2: this pixel is input2's tap (i2 = y), so i1 = y - sU
          const int y1 = y - sU, hn2 = y1 + p.padH - i * p.dilH;
          const bool hok2 = y1 >= 0 && y1 < p.iH && hn2 >= 0 && hn2 % p.dH == 0 && hn2 / p.dH < p.oH;
          if (!hok && !hok2) continue;
          for (int jj = 0; jj < p.kW; ++jj) {
            if (hok) {
              const int wn = x + p.padW - jj * p.dilW;
              const int y2 = y + sU, x2 = x + sV;
              if (wn >= 0 && wn % p.dW == 0 && wn / p.dW < p.oW && y2 >= 0 && y2 < p.iH && x2 >= 0 && x2 < p.iW)
                s1 += (A)gp[(size_t)(hn / p.dH) * p.oW + wn / p.dW] * (A)b[(size_t)y2 * p.iW + x2];
            }
            if (hok2) {
              const int x1 = x - sV, wn2 = x1 + p.padW - jj * p.dilW;
              if (x1 >= 0 && x1 < p.iW && wn2 >= 0 && wn2 % p.dW == 0 && wn2 / p.dW < p.oW)
                s2 += (A)gp[(size_t)(hn2 / p.dH) * p.oW + wn2 / p.dW] * (A)a[(size_t)y1 * p.iW + x1];
            }
          }
        }
      }
    }
    g1[idx] = (T)s1;
    g2[idx] = (T)s2;
  }
}

static int fill_params(CorrParams& p, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                       int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW) {
  if (B <= 0 || C <= 0 || iH <= 0 || iW <= 0) return fail(RFN_EINVAL, "corr: non-positive tensor size");
  if (kH <= 0 || kW <= 0 || patchH <= 0 || patchW <= 0 || dH <= 0 || dW <= 0 || dilH <= 0 || dilW <= 0 ||
      dpH <= 0 || dpW <= 0 || padH < 0 || padW < 0)
    return fail(RFN_EINVAL, "corr: invalid kernel/patch/stride/dilation/padding");
  p = {B, C, iH, iW, 0, 0, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW};
  p.oH = (iH + 2 * padH - ((kH - 1) * dilH + 1)) / dH + 1;
  p.oW = (iW + 2 * padW - ((kW - 1) * dilW + 1)) / dW + 1;
  if (p.oH <= 0 || p.oW <= 0) return fail(RFN_EINVAL, "corr: empty output (%d x %d)", p.oH, p.oW);
  return RFN_OK;
}

static inline bool is_hot_param(const CorrParams& p) {
  return p.kH == 1 && p.kW == 1 && p.patchH == 9 && p.patchW == 9 && p.padH == 0 && p.padW == 0 && p.dH == 1 &&
         p.dW == 1 && p.dpH == 1 && p.dpW == 1;
}
static inline bool is_k1(const CorrParams& p) {
  return p.kH == 1 && p.kW == 1 && p.padH == 0 && p.padW == 0 && p.dH == 1 && p.dW == 1;
}

template <typename T>
static int corr_fwd_any(const T* in1, const T* in2, T* out, const CorrParams& p, hipStream_t st) {
  const long total = (long)p.B * p.patchH * p.patchW * p.oH * p.oW;
  const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 32);
  hipLaunchKernelGGL((corr_generic_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, in1, in2, out, p, total);
  return check_launch("corr_generic_fwd_kernel");
}

template <typename T>
static int corr_bwd_any(const T* in1, const T* in2, const T* gout, T* g1, T* g2, const CorrParams& p,
                        hipStream_t st) {
  if constexpr (std::is_same<T, float>::value) {
    if (is_hot_param(p)) {
      const int tilesX = cdiv(p.iW, 32), tilesY = cdiv(p.iH, 8);
      const long blocks = (long)p.B * tilesX * tilesY;
      if (blocks > 0x7fffffffL) return fail(RFN_EINVAL, "corr bwd: grid too large");
      if (p.iW % 4 == 0 && p.C % 8 == 0 && (size_t)p.C * p.iH * p.iW < (1ull << 31) &&
          (((size_t)in1 | (size_t)in2 | (size_t)gout | (size_t)g1 | (size_t)g2) & 15) == 0) {
        hipLaunchKernelGGL(corr9_bwd_strip_kernel, dim3((unsigned)blocks, 2), dim3(192), 0, st, in1, in2, gout, g1, g2, p.C, p.iH,
                           p.iW, tilesX, tilesY);
        return check_launch("corr9_bwd_strip_kernel");
      }
      hipLaunchKernelGGL((corr9_bwd_tile_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, in2, gout, g1, p.C,
                         p.iH, p.iW, tilesX, tilesY);
      if (int rc = check_launch("corr9_bwd_tile_kernel<1>")) return rc;
      hipLaunchKernelGGL((corr9_bwd_tile_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, in1, gout, g2, p.C,
                         p.iH, p.iW, tilesX, tilesY);
      return check_launch("corr9_bwd_tile_kernel<2>");
    }
  }
  if constexpr (std::is_same<T, __half>::value) {
    const long total = (long)p.B * p.C * p.iH * p.iW;
    const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 32);
    hipLaunchKernelGGL((corr_generic_bwd_gather_kernel<T>), dim3(grid), dim3(256), 0, st, in1, in2, gout, g1, g2, p, total);
    return check_launch("corr_generic_bwd_gather_kernel");
  } else {
  if (is_k1(p)) {
    const long total = (long)p.B * p.C * p.iH * p.iW;
    const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 32);
    hipLaunchKernelGGL((corr_k1_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, in1, in2, gout, g1, g2, p, total);
    return check_launch("corr_k1_bwd_kernel");
  }
  const size_t bytes = sizeof(T) * (size_t)p.B * p.C * p.iH * p.iW;
  if (hipMemsetAsync(g1, 0, bytes, st) != hipSuccess || hipMemsetAsync(g2, 0, bytes, st) != hipSuccess)
    return fail(RFN_ELAUNCH, "corr bwd: hipMemsetAsync failed");
  const long total = (long)p.B * p.patchH * p.patchW * p.oH * p.oW;
  const int grid = (int)std::min<long>(cdiv(total, 256), 256L * 32);
  hipLaunchKernelGGL((corr_generic_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, in1, in2, gout, g1, g2, p,
                     total);
  return check_launch("corr_generic_bwd_kernel");
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" {

int rfn_corr_fwd_f32(const float* in1, const float* in2, float* out, int B, int C, int iH, int iW, int kH,
                     int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW,
                     int dH, int dW, rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && out, "rfn_corr_fwd_f32: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  hipStream_t st = (hipStream_t)stream;
  if (is_hot_param(p)) return launch_corr9<false, false>(in1, in2, nullptr, out, B, C, iH, iW, st);
  return corr_fwd_any<float>(in1, in2, out, p, st);
}

int rfn_corr_fwd_f64(const double* in1, const double* in2, double* out, int B, int C, int iH, int iW, int kH,
                     int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW,
                     int dH, int dW, rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && out, "rfn_corr_fwd_f64: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  return corr_fwd_any<double>(in1, in2, out, p, (hipStream_t)stream);
}

int rfn_corr_bwd_f32(const float* in1, const float* in2, const float* grad_out, float* grad_in1,
                     float* grad_in2, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                     int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && grad_out && grad_in1 && grad_in2, "rfn_corr_bwd_f32: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  return corr_bwd_any<float>(in1, in2, grad_out, grad_in1, grad_in2, p, (hipStream_t)stream);
}

int rfn_corr_bwd_f64(const double* in1, const double* in2, const double* grad_out, double* grad_in1,
                     double* grad_in2, int B, int C, int iH, int iW, int kH, int kW, int patchH, int patchW,
                     int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && grad_out && grad_in1 && grad_in2, "rfn_corr_bwd_f64: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  return corr_bwd_any<double>(in1, in2, grad_out, grad_in1, grad_in2, p, (hipStream_t)stream);
}

int rfn_corr_fwd_f16(const void* in1, const void* in2, void* out, int B, int C, int iH, int iW, int kH, int kW, int patchH,
                     int patchW, int padH, int padW, int dilH, int dilW, int dpH, int dpW, int dH, int dW,
                     rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && out, "rfn_corr_fwd_f16: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  return corr_fwd_any<__half>((const __half*)in1, (const __half*)in2, (__half*)out, p, (hipStream_t)stream);
}

int rfn_corr_bwd_f16(const void* in1, const void* in2, const void* grad_out, void* grad_in1, void* grad_in2, int B, int C,
                     int iH, int iW, int kH, int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW, int dpH,
                     int dpW, int dH, int dW, rfn_stream_t stream) {
  RFN_REQUIRE(in1 && in2 && grad_out && grad_in1 && grad_in2, "rfn_corr_bwd_f16: null pointer");
  CorrParams p;
  if (int rc = fill_params(p, B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW))
    return rc;
  return corr_bwd_any<__half>((const __half*)in1, (const __half*)in2, (const __half*)grad_out, (__half*)grad_in1,
                              (__half*)grad_in2, p, (hipStream_t)stream);
}

int rfn_local_corr_layer_f32(const float* feature_target, const float* feature_source, const float* flow,
                             float* out, int B, int C, int H, int W, rfn_stream_t stream) {
  RFN_REQUIRE(feature_target && feature_source && out, "rfn_local_corr_layer_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "rfn_local_corr_layer_f32: non-positive size");
  hipStream_t st = (hipStream_t)stream;
  if (flow) return launch_corr9<true, true>(feature_target, feature_source, flow, out, B, C, H, W, st);
  return launch_corr9<true, false>(feature_target, feature_source, nullptr, out, B, C, H, W, st);
}

long rfn_local_corr_layer_split_workspace_bytes(int B, int H, int W, int splits) {
  if (B <= 0 || H <= 0 || W <= 0 || splits <= 0) return 0;
  const long tickets = (long)B * rfn::cdiv(W, 32) * rfn::cdiv(H, 8);
  return ((long)splits * B * 81 * H * W + tickets) * 4;
}

int rfn_local_corr_layer_split_f32(const float* feature_target, const float* feature_source, float* out,
                                   float* workspace, int B, int C, int H, int W, int splits, rfn_stream_t stream) {
  RFN_REQUIRE(feature_target && feature_source && out && workspace, "rfn_local_corr_layer_split_f32: null pointer");
  RFN_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && (W & 3) == 0, "rfn_local_corr_layer_split_f32: B=%d C=%d H=%d W=%d (W %% 4 == 0)",
              B, C, H, W);
  RFN_REQUIRE(splits >= 2 && splits <= 64 && C % splits == 0 && (C / splits) % 8 == 0,
              "rfn_local_corr_layer_split_f32: %d channels in %d chunks (chunks of a multiple of 8 channels)", C, splits);
  return launch_corr9_split<true>(feature_target, feature_source, out, workspace, B, C, H, W, splits, (hipStream_t)stream);
}

}  // extern "C"
