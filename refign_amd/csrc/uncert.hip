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
                                                               int HW, long npix, int ngroups) {
  // 81 624 B: two workgroups per CU (160 KB), so one group's VALU phases / barriers / global latency hide under the
  // other's MFMA phases.  s_in is dead once L0 has run and s_a1 is first written by L1; s_a0 is dead once L1 has
  // run and s_a2 is first written by L2 -- so they share storage.
  __shared__ float smem[kUP * 49 * 33 + kUP * 25 * 33 + 144 * 6 + 6];
  float* const s_a0 = smem;                        // [px][pos49][ci32] pitch 33
  float* const s_a1 = smem + kUP * 49 * 33;        // [px][pos25][ci32] pitch 33
  float* const s_w3 = s_a1 + kUP * 25 * 33;
  float* const s_in = s_a1;                        // [px][81]
  float* const s_a2 = s_a0;                        // [px][pos9][ci16]  pitch 17

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
  for (int i = tid; i < 144 * 6 + 6; i += 256) s_w3[i] = wts[kOffW3 + i];

  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const long pix0 = (long)g * kUP;
    __syncthreads();   // previous group fully consumed (also publishes s_w3 on the first trip)
    // ---- load the 8 x 81 correlation patches: corr is (B,81,H,W); pixel index = b*HW + hw ------------------
    for (int i = tid; i < kUP * 81; i += 256) {
      const int p = i & 7, d = i >> 3;                    // 8 consecutive pixels of one shift plane
      const long pix = pix0 + p;
      float v = 0.0f;
      if (pix < npix) {
        const long b = pix / HW, hw = pix - b * HW;
        v = corr[(b * 81 + d) * HW + hw];
      }
      s_in[p * 81 + d] = v;
    }
    __syncthreads();
    // ---- L0: 1 -> 32 channels, 9x9 -> 7x7 -----------------------------------------------------------------
    {
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
    for (int mt = wave >> 1; mt < 13; mt += 2) {
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
    // ---- L2: 32 -> 16, 5x5 -> 3x3; rows m = p*9 + y*3 + x (72 valid of 5 x 16) ------------------------------
    for (int mt = wave; mt < 5; mt += 4) {
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
    // ---- L3: 16 -> 6 on the 3x3 map -> one value per (pixel, output channel) ---------------------------------
    if (tid < kUP * 6) {
      const int p = tid / 6, c = tid - p * 6;
      float acc = s_w3[144 * 6 + c];
      const float* ap = s_a2 + p * 9 * 17;
#pragma unroll 4
      for (int pos = 0; pos < 9; ++pos)
#pragma unroll
        for (int ci = 0; ci < 16; ++ci) acc = fmaf(s_w3[(pos * 16 + ci) * 6 + c], ap[pos * 17 + ci], acc);
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
  const int grid = (int)std::min<long>(ngroups, 256L * 2);   // 2 resident workgroups per CU, grid-stride
  hipLaunchKernelGGL(uncert9_frontend_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, corr, weights, out, H * W,
                     npix, (int)ngroups);
  return check_launch("uncert9_frontend_kernel");
}

}  // extern "C"
