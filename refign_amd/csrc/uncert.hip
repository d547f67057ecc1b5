// refign_amd/csrc/uncert.hip -- front end of the UAWarpC UncertaintyModule (search size 9), fused, fp32 MFMA.
//
// Reference: UncertaintyModule.forward, models/modules.py:529-551: the (B,81,H,W) correlation volume is reshaped into
// B*H*W one-channel 9x9 micro-images that go through three VALID 3x3 convs + BN + LeakyReLU(0.1)
// (1->32: 9x9->7x7, 32->32: ->5x5, 32->16: ->3x3) and a 3x3 conv 16->6 (->1x1): six numbers per pixel.
// Unfused, the intermediates are (BHW,32,7,7), (BHW,32,5,5), (BHW,16,3,3): 813 + 415 + 75 MB per 1080x1920 image at
// level 1, written and re-read by library micro-convolutions plus NCHW<->NHWC transposes (11 ms of the 56 ms align).
// Here a workgroup takes 8 pixels at a time and keeps everything in LDS / registers:
//   L0 (1->32, 9 MAC per output)           VALU, lanes = output channels
//   L1 (32->32, K = 288)  M = 8*25 rows    v_mfma_f32_16x16x4_f32: exact fp32 (an fmaf chain), 13 M-tiles x 2 N-tiles
//   L2 (32->16, K = 288)  M = 8*9 rows     v_mfma_f32_16x16x4_f32, 5 M-tiles x 1 N-tile
//   L3 (16->6, K = 144)                    VALU
// The B operands (weights, BatchNorm folded in by the host) of L1/L2 live in registers for the whole kernel (72 VGPRs
// each: one dword per K-step); the A operands are single ds_read_b32 per MFMA from activation tiles stored
// [pixel][position][channel] with pitch 33/17 (bank spread).  K is ordered tap-major (k = tap*Cin + ci) so that an A
// read is `row base + compile-time offset`.
// MFMA lane maps (cdna_hip_programming.md §3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D[row = (l>>4)*4 + r][col = l&15], r = 0..3.
//
// Packed weight buffer (floats), prepared by refign_amd/align.py:
//   [0,288)      W0t[tap][c32]           [288,320)   b0[32]
//   [320,9536)   W1k[k288][n32]          [9536,9568) b1[32]
//   [9568,14176) W2k[k288][n16]          [14176,14192) b2[16]
//   [14192,15056) W3k[k144][c6]          [15056,15062) b3[6]          k144 = pos9*16 + ci
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace rfn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kUP = 8;                      // pixels per workgroup iteration
constexpr int kOffW0 = 0, kOffB0 = 288, kOffW1 = 320, kOffB1 = 9536, kOffW2 = 9568, kOffB2 = 14176, kOffW3 = 14192,
              kUncertWeights = 15062;

__device__ __forceinline__ float leaky(float v) { return v > 0.0f ? v : 0.1f * v; }

// K = 9 taps x 32 input channels = 72 MFMA steps on one accumulator.  The 8 A values of tap t+1 are fetched from LDS
// while the 8 MFMAs of tap t issue (one wave per SIMD here: nothing else hides the ds_read latency).  WIN = width of
// the input map (7 for L1, 5 for L2); `abase` = this lane's row base (+ its k-slice lk).
template <int WIN>
__device__ __forceinline__ void mfma_chain(const float* abase, const float (&wfrag)[72], f32x4& acc) {
  float a[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[0][j] = abase[4 * j];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap + 1 < 9) {
      const int ky = (tap + 1) / 3, kx = (tap + 1) - ky * 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) a[(tap + 1) & 1][j] = abase[(ky * WIN + kx) * 33 + 4 * j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap & 1][j], wfrag[tap * 8 + j], acc, 0, 0, 0);
    // keep the machine scheduler from sinking the prefetch back down to its uses: DS reads first, then the MFMAs
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
}

__global__ __launch_bounds__(256, 2) void uncert9_frontend_kernel(const float* __restrict__ corr,
                                                               const float* __restrict__ wts, float* __restrict__ out,
                                                               int HW, long npix, int ngroups, int ablate) {
  // 78 144 B: two workgroups per CU (160 KB), so one group's VALU phases / barriers hide under the other's MFMA
  // phases.  Storage is shared where lifetimes allow: s_in is dead once L0 has run and s_a1 is first written by L1;
  // s_a0 is dead once L1 has run, then holds s_a2 (written by L2) and the L3 weights (parked in registers between).
  __shared__ float smem[kUP * 49 * 33 + kUP * 25 * 33];
  float* const s_a0 = smem;                        // [px][pos49][ci32] pitch 33
  float* const s_a1 = smem + kUP * 49 * 33;        // [px][pos25][ci32] pitch 33
  float* const s_in = s_a1;                        // [px][81]
  float* const s_a2 = s_a0;                        // [px][pos9][ci16]  pitch 17
  float* const s_w3 = s_a0 + 1280;                 // [k144][c6] + b3[6]
  float* const s_part = s_a0 + 2176;               // [3][48] partial sums of L3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;

  // ---- per-kernel constants in registers -----------------------------------------------------------------
  // L0: thread = (channel c = tid & 31, pixel p = tid >> 5)
  const int c0 = tid & 31, p0 = tid >> 5;
  float w0[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w0[t] = wts[kOffW0 + t * 32 + c0];
  const float bias0 = wts[kOffB0 + c0];
  // L1 B fragments: wave -> N-tile (wave & 1); step s: k = 4 s + lk, n = ntile*16 + li
  const int nt1 = wave & 1;
  float wB[72];
#pragma unroll
  for (int s = 0; s < 72; ++s) wB[s] = wts[kOffW1 + (4 * s + lk) * 32 + nt1 * 16 + li];
  const float bias1 = wts[kOffB1 + nt1 * 16 + li];
  // L2 B fragments (N = 16: one tile)
  float wC[72];
#pragma unroll
  for (int s = 0; s < 72; ++s) wC[s] = wts[kOffW2 + (4 * s + lk) * 16 + li];
  const float bias2 = wts[kOffB2 + li];
  float w3r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) w3r[j] = (tid + 256 * j < 144 * 6 + 6) ? wts[kOffW3 + tid + 256 * j] : 0.0f;

  // the 8 x 81 correlation patches of a group: corr is (B,81,H,W), pixel index = b*HW + hw; element i of the group is
  // (pixel i & 7, shift i >> 3) -- 8 consecutive pixels of one shift plane.  Fetched one group ahead into registers.
  float pre[3];
  auto fetch_patches = [&](int g) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + 256 * j;
      const long pix = (long)g * kUP + (i & 7);
      float v = 0.0f;
      if (i < kUP * 81 && pix < npix) {
        const long b = pix / HW, hw = pix - b * HW;
        v = corr[(b * 81 + (i >> 3)) * HW + hw];
      }
      pre[j] = v;
    }
  };
  if ((int)blockIdx.x < ngroups) fetch_patches(blockIdx.x);

  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long pix0 = (long)g * kUP;
    __syncthreads();   // previous group fully consumed
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + 256 * j;
      if (i < kUP * 81) s_in[(i & 7) * 81 + (i >> 3)] = pre[j];
    }
    __syncthreads();
    if (g + (int)gridDim.x < ngroups) fetch_patches(g + gridDim.x);
    // ---- L0: 1 -> 32 channels, 9x9 -> 7x7 -----------------------------------------------------------------
    if (!(ablate & 2)) {
      const float* ip = s_in + p0 * 81;
      float* op = s_a0 + p0 * 49 * 33 + c0;
#pragma unroll 7
      for (int pos = 0; pos < 49; ++pos) {
        const int y = pos / 7, x = pos - y * 7;
        float acc = bias0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(w0[ky * 3 + kx], ip[(y + ky) * 9 + x + kx], acc);
        op[pos * 33] = leaky(acc);
      }
    }
    __syncthreads();
    // ---- L1: 32 -> 32, 7x7 -> 5x5; rows m = p*25 + y*5 + x (200 valid of 13 x 16) ---------------------------
    for (int mt = wave >> 1; mt < ((ablate & 1) ? 0 : 13); mt += 2) {
      const int m = min(mt * 16 + li, kUP * 25 - 1);       // my A row (clamped for the ragged last tile)
      const int p = m / 25, r = m - p * 25, y = r / 5, x = r - y * 5;
      const float* abase = s_a0 + (p * 49 + y * 7 + x) * 33 + lk;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mfma_chain<7>(abase, wB, acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mo = mt * 16 + lk * 4 + rr;              // D row
        if (mo < kUP * 25) {
          const int po = mo / 25, ro = mo - po * 25;
          s_a1[(po * 25 + ro) * 33 + nt1 * 16 + li] = leaky(acc[rr] + bias1);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (tid + 256 * j < 144 * 6 + 6) s_w3[tid + 256 * j] = w3r[j];     // s_a0 is dead: park the L3 weights there
    // ---- L2: 32 -> 16, 5x5 -> 3x3; rows m = p*9 + y*3 + x (72 valid of 5 x 16) ------------------------------
    // tiles 0..3 -> waves 0..3; the fifth tile goes to wave 3, which had 6 (not 7) L1 tiles: 8/8/7/8 MFMA tiles per wave
    for (int mt = wave; mt < ((ablate & 4) ? 0 : (wave == 3 ? 5 : wave + 1)); ++mt) {
      const int m = min(mt * 16 + li, kUP * 9 - 1);
      const int p = m / 9, r = m - p * 9, y = r / 3, x = r - y * 3;
      const float* abase = s_a1 + (p * 25 + y * 5 + x) * 33 + lk;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      mfma_chain<5>(abase, wC, acc);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mo = mt * 16 + lk * 4 + rr;
        if (mo < kUP * 9) s_a2[mo * 17 + li] = leaky(acc[rr] + bias2);     // mo = p*9 + pos
      }
    }
    __syncthreads();
    // ---- L3: 16 -> 6 on the 3x3 map -> one value per (pixel, output channel); K = 144 split over 4 thread groups -
    float part = 0.0f;
    const int o3 = tid % 48, ks = tid / 48;                 // output (p, c) = (o3 / 6, o3 % 6), k-slice
    if (tid < 192 && !(ablate & 8)) {
      const int p = o3 / 6, c = o3 - p * 6;
      const float* ap = s_a2 + p * 9 * 17;
#pragma unroll 6
      for (int kk = 0; kk < 36; ++kk) {
        const int k = ks * 36 + kk;
        part = fmaf(s_w3[k * 6 + c], ap[(k >> 4) * 17 + (k & 15)], part);
      }
      if (ks > 0) s_part[(ks - 1) * 48 + o3] = part;
    }
    __syncthreads();
    if (tid < 48) {
      const int p = o3 / 6, c = o3 - p * 6;
      const float acc = s_w3[144 * 6 + c] + ((part + s_part[o3]) + (s_part[48 + o3] + s_part[96 + o3]));
      const long pix = pix0 + p;
      if (pix < npix) {
        const long b = pix / HW, hw = pix - b * HW;
        out[(b * 6 + c) * HW + hw] = acc;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same front end with the two matrix layers on the f16 matrix pipe (round 5).  The reference runs these convolutions under
// autocast in its AMP recipe (fp16 operands, fp32 accumulation, fp16 activations: nothing in models/modules.py:529-551 forces
// fp32), so in the timed mode this is its precision map; the fp32 kernel above stays the parity mode's.  v_mfma_f32_16x16x32_f16:
// K = 32 per instruction = the 32 input channels of ONE tap, so a 16 x 16 tile of L1 / L2 is 9 MFMAs (fp32 form: 72 of twice the
// issue time) -- the matrix phases shrink from ~18 000 to ~1 200 clocks per group of 8 pixels and the kernel becomes its VALU
// layers (L0, L3) and barriers.  Activations live in LDS as f16, CHANNEL-GROUP-major: [kg = ci / 8][row][8 channels]; an A
// fragment (lane: row l & 15, channels 8 (l >> 4) .. + 7) is one ds_read_b128 at `plane kg + row * 16 B + tap offset`.  B
// fragments (36 VGPRs per layer) are converted from the same packed fp32 weights at kernel start.  47 KB of LDS: 3 workgroups/CU.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kRows0 = 400, kRows1 = 208;   // rows per channel-group plane (8 x 49 = 392, 8 x 25 = 200, rounded up to 16)

__global__ __launch_bounds__(256, 3) void uncert9_frontend_h_kernel(const float* __restrict__ corr,
                                                                  const float* __restrict__ wts, float* __restrict__ out,
                                                                  int HW, long npix, int ngroups) {
  __shared__ __attribute__((aligned(16))) _Float16 s_a0[4 * kRows0 * 8];   // [kg][p*49 + pos][8]
  __shared__ __attribute__((aligned(16))) _Float16 s_a1[4 * kRows1 * 8];   // [kg][p*25 + pos][8]
  __shared__ float s_in[kUP * 81];
  __shared__ float s_a2[kUP * 9 * 17];                                       // [p*9 + pos][ci16] pitch 17 (fp32: L3 is VALU)
  __shared__ float s_w3[144 * 6 + 6];
  __shared__ float s_part[3 * 48];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int c0 = tid & 31, p0 = tid >> 5;
  __shared__ float s_w0[288 + 32];                     // L0 weights + bias: re-read per group (no registers held across the loop)
  for (int i = tid; i < 320; i += 256) s_w0[i] = wts[kOffW0 + i];
  const int nt1 = wave & 1;
  h8 wB[9], wC[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wB[tap][e] = (_Float16)wts[kOffW1 + (tap * 32 + 8 * lk + e) * 32 + nt1 * 16 + li];
      wC[tap][e] = (_Float16)wts[kOffW2 + (tap * 32 + 8 * lk + e) * 16 + li];
    }
  const float bias1 = wts[kOffB1 + nt1 * 16 + li];
  const float bias2 = wts[kOffB2 + li];
  for (int i = tid; i < 144 * 6 + 6; i += 256) s_w3[i] = wts[kOffW3 + i];

  float pre[3];
  auto fetch_patches = [&](int g) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + 256 * j;
      const long pix = (long)g * kUP + (i & 7);
      float v = 0.0f;
      if (i < kUP * 81 && pix < npix) {
        const long b = pix / HW, hw = pix - b * HW;
        v = corr[(b * 81 + (i >> 3)) * HW + hw];
      }
      pre[j] = v;
    }
  };
  if ((int)blockIdx.x < ngroups) fetch_patches(blockIdx.x);

  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long pix0 = (long)g * kUP;
    __syncthreads();   // previous group fully consumed
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int i = tid + 256 * j;
      if (i < kUP * 81) s_in[(i & 7) * 81 + (i >> 3)] = pre[j];
    }
    __syncthreads();
    if (g + (int)gridDim.x < ngroups) fetch_patches(g + gridDim.x);
    // ---- L0: 1 -> 32 channels, 9x9 -> 7x7 (fp32 VALU, f16 result) --------------------------------------------
    {
      float w0[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) w0[t] = s_w0[t * 32 + c0];
      const float bias0 = s_w0[288 + c0];
      const float* ip = s_in + p0 * 81;
      _Float16* op = s_a0 + ((c0 >> 3) * kRows0 + p0 * 49) * 8 + (c0 & 7);
#pragma unroll 7
      for (int pos = 0; pos < 49; ++pos) {
        const int y = pos / 7, x = pos - y * 7;
        float acc = bias0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) acc = fmaf(w0[ky * 3 + kx], ip[(y + ky) * 9 + x + kx], acc);
        op[pos * 8] = (_Float16)leaky(acc);
      }
    }
    __syncthreads();
    // ---- L1: 32 -> 32, 7x7 -> 5x5; rows m = p*25 + y*5 + x (200 valid of 13 x 16); one MFMA per tap -----------
    for (int mt = wave >> 1; mt < 13; mt += 2) {
      const int m = min(mt * 16 + li, kUP * 25 - 1);
      const int p = m / 25, r = m - p * 25, y = r / 5, x = r - y * 5;
      const h8* abase = reinterpret_cast<const h8*>(s_a0) + lk * kRows0 + p * 49 + y * 7 + x;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(abase[(tap / 3) * 7 + tap % 3], wB[tap], acc, 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mo = mt * 16 + lk * 4 + rr;              // D row; D column = output channel nt1*16 + li
        if (mo < kUP * 25) {
          const int n = nt1 * 16 + li;
          s_a1[((n >> 3) * kRows1 + mo) * 8 + (n & 7)] = (_Float16)leaky(acc[rr] + bias1);
        }
      }
    }
    __syncthreads();
    // ---- L2: 32 -> 16, 5x5 -> 3x3; rows m = p*9 + y*3 + x (72 valid of 5 x 16) ------------------------------
    for (int mt = wave; mt < (wave == 3 ? 5 : wave + 1); ++mt) {
      const int m = min(mt * 16 + li, kUP * 9 - 1);
      const int p = m / 9, r = m - p * 9, y = r / 3, x = r - y * 3;
      const h8* abase = reinterpret_cast<const h8*>(s_a1) + lk * kRows1 + p * 25 + y * 5 + x;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(abase[(tap / 3) * 5 + tap % 3], wC[tap], acc, 0, 0, 0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mo = mt * 16 + lk * 4 + rr;
        if (mo < kUP * 9) s_a2[mo * 17 + li] = leaky(acc[rr] + bias2);
      }
    }
    __syncthreads();
    // ---- L3: 16 -> 6 on the 3x3 map (fp32 VALU, as in the fp32 kernel) ---------------------------------------
    float part = 0.0f;
    const int o3 = tid % 48, ks = tid / 48;
    if (tid < 192) {
      const int p = o3 / 6, c = o3 - p * 6;
      const float* ap = s_a2 + p * 9 * 17;
#pragma unroll 6
      for (int kk = 0; kk < 36; ++kk) {
        const int k = ks * 36 + kk;
        part = fmaf(s_w3[k * 6 + c], ap[(k >> 4) * 17 + (k & 15)], part);
      }
      if (ks > 0) s_part[(ks - 1) * 48 + o3] = part;
    }
    __syncthreads();
    if (tid < 48) {
      const int p = o3 / 6, c = o3 - p * 6;
      const float acc = s_w3[144 * 6 + c] + ((part + s_part[o3]) + (s_part[48 + o3] + s_part[96 + o3]));
      const long pix = pix0 + p;
      if (pix < npix) {
        const long b = pix / HW, hw = pix - b * HW;
        out[(b * 6 + c) * HW + hw] = acc;
      }
    }
  }
}

}  // namespace rfn

using namespace rfn;

extern "C" {

int rfn_uncertainty9_weights_len(void) { return kUncertWeights; }

int rfn_uncertainty9_frontend_f32(const float* corr, const float* weights, float* out, int B, int H, int W,
                                  rfn_stream_t stream) {
  RFN_REQUIRE(corr && weights && out, "rfn_uncertainty9_frontend_f32: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0, "rfn_uncertainty9_frontend_f32: non-positive size");
  const long npix = (long)B * H * W;
  const long ngroups = (npix + kUP - 1) / kUP;
  RFN_REQUIRE(ngroups < 0x7fffffffL, "rfn_uncertainty9_frontend_f32: too many pixels");
  // 2 resident workgroups per CU, grid-stride.  RFN_UNCERT_GRID / RFN_UNCERT_ABLATE: measurement knobs (tools/kbench.py)
  const char* eg = nullptr;
  const char* ea = nullptr;
  const int grid = (int)std::min<long>(ngroups, eg ? atol(eg) : 256L * 2);
  hipLaunchKernelGGL(uncert9_frontend_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, corr, weights, out, H * W,
                     npix, (int)ngroups, ea ? atoi(ea) : 0);
  return check_launch("uncert9_frontend_kernel");
}

int rfn_uncertainty9_frontend_f16mm(const float* corr, const float* weights, float* out, int B, int H, int W,
                                    rfn_stream_t stream) {
  RFN_REQUIRE(corr && weights && out, "rfn_uncertainty9_frontend_f16mm: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0, "rfn_uncertainty9_frontend_f16mm: non-positive size");
  const long npix = (long)B * H * W;
  const long ngroups = (npix + kUP - 1) / kUP;
  RFN_REQUIRE(ngroups < 0x7fffffffL, "rfn_uncertainty9_frontend_f16mm: too many pixels");
  const int grid = (int)std::min<long>(ngroups, 256L * 3);
  hipLaunchKernelGGL(uncert9_frontend_h_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, corr, weights, out, H * W, npix,
                     (int)ngroups);
  return check_launch("uncert9_frontend_h_kernel");
}

}  // extern "C"
