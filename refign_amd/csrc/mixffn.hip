// refign_amd/csrc/mixffn.hip -- Mix-FFN front half of the gradient-free MiT passes (mix_transformer.py:79-103): fc1, the depthwise
// 3x3 convolution and the exact GELU in ONE kernel,   a = gelu(dw3x3(x W1^T + b1) + bdw),   bf16 in / out, fp32 accumulation.
// The EMA teacher's 40 views made the 4C-wide hidden tensor the step's largest HBM customer: fc1 wrote it (209 MB per block at
// stage 3), the depthwise kernel read and re-wrote it (8.4 ms of the 73 ms the gradient-free half takes on its own:
// profiles/r06_teacher_half_kernel_stats.csv), fc2 read it.  Here the pre-activation never leaves the CU.
//
// A workgroup owns a SPATIAL tile of one view -- 8 x 32 tokens including a one-token halo, 6 x 30 without -- and 128 hidden channels
// (the depthwise convolution is per channel, so the hidden dimension tiles freely):
//   1. H^T[128 hidden][256 tokens] = W1[128 rows] . X[256 tokens]^T over C in chunks of 64 on the matrix pipe (32x32x16 bf16;
//      transposed so that a lane owns a TOKEN and four consecutive hidden channels per register group: 8-byte LDS writes);
//      operands through ONE LDS stage, the next chunk travelling in registers while a chunk is multiplied -- 68 KB of LDS and
//      <= 256 registers, so TWO workgroups share a CU and one's products run under the other's stencil phase.  The halo costs
//      1.42x the fc1 products of the interior -- cheaper than a round trip of the hidden tensor through HBM.
//   2. + b1, rounded to bf16 (what fc1 used to store), ZERO outside the image (the convolution's padding pads the hidden map, not
//      x), into an LDS tile [256 tokens][128 + 8 channels] laid over the ring.
//   3. the 3 x 3 stencil + bias + exact-erf GELU on the interior from LDS in strips of 6 outputs (3 x 8 halo values per strip; a
//      thread keeps its 8 channels' 72 weights in registers; packed fp32 math), 16-byte stores of the activation.
// Where the launch stands (profiles/r06_ffn_bench.txt, r06_pmc_ffn.txt; stage 3, 40 views: 215-230 us against 47 + 160 us for the
// fc1 GEMM + depthwise kernel back to back, 100 + 205 us each on its own): vector instructions 42 % of the launch, matrix pipe 20 %,
// the rest stalls -- each 64-channel chunk's operands arrive from L2 in ~1 us while its products take 0.5 us, and LDS (two
// workgroups per CU) has no room for a deeper ring.  Built, measured and removed this round: an fp32 hidden tile in LDS (no bf16
// unpacking in the stencil, one workgroup per CU: 341 us), a persistent grid with the CU's two workgroups started one phase apart
// (no change: they do not march in lockstep), an eight-wave workgroup at <= 128 registers (four waves per SIMD: 226 us, no change --
// occupancy is not what is missing).  In the step the kernel is worth 3.5 ms (same-box A/B, profiles/r06_fused_ffn_ab.txt): the
// hidden tensor's three trips through HBM were taken from the other two streams.
// fc2 (+ residual) stays the second-generation GEMM.  Numerics: the same roundings as the three-kernel path (bf16 hidden map, fp32
// stencil, bf16 activation); only the fp32 summation order of the stencil differs.
#include "common.h"
#include "mfma.h"

namespace rfn {

constexpr int kFfnTY = 8, kFfnTX = 32;                 // tile incl. halo (tokens)
constexpr int kFfnIY = kFfnTY - 2, kFfnIX = kFfnTX - 2;
constexpr int kFfnNH = 128;                            // hidden channels per workgroup
constexpr int kFfnKC = 64;                             // reduction chunk
constexpr int kFfnPitch = kFfnKC + 8;                  // LDS row pitch of an operand chunk (halfs): 144 B, conflict-free b128 reads
constexpr int kFfnXBytes = kFfnTY * kFfnTX * kFfnPitch * 2;          // 36 864
constexpr int kFfnWBytes = kFfnNH * kFfnPitch * 2;                   // 18 432
constexpr int kFfnStage = kFfnXBytes + kFfnWBytes;
constexpr int kFfnHPitch = kFfnNH + 8;                 // hidden tile row pitch (halfs): 272 B
constexpr int kFfnHBytes = kFfnTY * kFfnTX * kFfnHPitch * 2;        // 69 632
constexpr int kFfnLds = kFfnHBytes > kFfnStage ? kFfnHBytes : kFfnStage;   // the hidden tile lies over the operand stage
constexpr int kFfnStrip = 6;                           // interior outputs per stencil strip (30 = 5 strips per row)
static_assert(kFfnIX % kFfnStrip == 0, "strips tile the interior row");
typedef float f32p __attribute__((ext_vector_type(2)));

// exact-erf GELU on two adjacent channels in packed fp32 math.  erfc(z) = 2^q(z) with q a degree-7 polynomial fitted to log2(erfc)
// on [0, 3.3] (relative error of the GELU 7e-5 where |gelu| > 1e-3, absolute 1e-5 beyond: three orders below a bf16 step) -- ONE
// transcendental per value (v_exp_f32) where the Abramowitz-Stegun form of mfma.h needs a reciprocal too; the stencil phase of this
// kernel is bound by the vector ALU, and the GELU is two thirds of it.
__device__ __forceinline__ f32p gelu2(f32p x) {
  f32p z = f32p{fminf(fabsf(x[0]) * 0.70710678118654752440f, 3.3f), fminf(fabsf(x[1]) * 0.70710678118654752440f, 3.3f)};
  f32p q = z * 8.81649179e-05f + -0.000383410319f;
  q = q * z + -0.00249544169f;
  q = q * z + 0.0296828837f;
  q = q * z + -0.149101791f;
  q = q * z + -0.918289271f;
  q = q * z + -1.6279182f;
  q = q * z + 1.23649069e-07f;
  const f32p e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32p half = e * -0.5f + 0.5f;                   // 0.5 erf(|x| / sqrt 2)
  const f32p sg = {copysignf(half[0], x[0]), copysignf(half[1], x[1])};
  return x * (sg + 0.5f);
}

struct FfnRows {
  u32x4 x[8];      // a 16-byte piece of the X chunk's row of one halo token per halo row
  u32x4 w[4];      // the same piece of four W1 rows
};

__global__ __launch_bounds__(256, 2) void ffn_fc1_dw_gelu_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W1,
                                                              const uint16_t* __restrict__ b1, const float* __restrict__ wdw,
                                                              const float* __restrict__ bdw, uint16_t* __restrict__ A, int H,
                                                              int W, int C, int HID, int tiles_y, int tiles_x) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kFfnLds];
  const int t = threadIdx.x, wave = t >> 6, l = t & 63, j = l & 31, g = l >> 5;
  const int ntn = HID / kFfnNH;
  // the workgroups that share an X tile (the hidden slices of one spatial tile) are consecutive logical ids: keep them on ONE XCD
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = bid % ntn;
  bid /= ntn;
  const int tx = bid % tiles_x;
  bid /= tiles_x;
  const int ty = bid % tiles_y, v = bid / tiles_y;
  const int y0 = ty * kFfnIY - 1, x0 = tx * kFfnIX - 1, n0 = nt * kFfnNH;

  // ---- operand staging: 8 threads per 128-byte row piece (coalesced): thread t carries piece t & 7 of halo column t >> 3 in each
  // of the 8 halo rows, and of W1 rows (t >> 3) + 32 i -------------------------------------------------------------------------
  const int sub = t & 7, rb = t >> 3;
  const int mx = x0 + rb;
  const bool col_in = mx >= 0 && mx < W;
  const uint16_t* xcol = X + ((long)v * H * W + (col_in ? mx : 0)) * C + sub * 8;
  const uint16_t* wrow = W1 + (long)(n0 + rb) * C + sub * 8;
  auto fetch = [&](int k0, FfnRows& r) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int my = y0 + i;
      r.x[i] = (col_in && my >= 0 && my < H) ? *(const u32x4*)(xcol + (long)my * W * C + k0) : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = *(const u32x4*)(wrow + (long)32 * i * C + k0);
  };
  auto stash = [&](const FfnRows& r) {
    unsigned char* xs = smem + rb * (kFfnPitch * 2) + sub * 16;
    unsigned char* ws = xs + kFfnXBytes;
#pragma unroll
    for (int i = 0; i < 8; ++i) *(u32x4*)(xs + 32 * i * (kFfnPitch * 2)) = r.x[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(ws + 32 * i * (kFfnPitch * 2)) = r.w[i];
  };

  // ---- phase 1: H^T = W1 X^T.  Waves 2 (hidden) x 2 (tokens): wave owns hidden blocks 2 wn, 2 wn + 1 and token blocks 4 wm .. + 3
  const int wn = wave >> 1, wm = wave & 1;
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // ONE operand stage (68 KB of LDS and <= 256 registers: two workgroups per CU, eight waves to hide the stencil phase's
  // latencies): the next chunk travels in registers during the products, two barriers per chunk
  FfnRows rows;
  fetch(0, rows);
  const int nk = C / kFfnKC;
  for (int kc = 0; kc < nk; ++kc) {
    if (kc) __syncthreads();
    stash(rows);
    __syncthreads();
    if (kc + 1 < nk) fetch((kc + 1) * kFfnKC, rows);
    const unsigned char* xs = smem;
    const unsigned char* ws = xs + kFfnXBytes;
#pragma unroll
    for (int ks = 0; ks < kFfnKC / 16; ++ks) {
      bf16x8 wf[2], xf[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
        wf[a] = *(const bf16x8*)(ws + ((2 * wn + a) * 32 + j) * (kFfnPitch * 2) + (ks * 16 + 8 * g) * 2);
#pragma unroll
      for (int b = 0; b < 4; ++b)
        xf[b] = *(const bf16x8*)(xs + ((4 * wm + b) * 32 + j) * (kFfnPitch * 2) + (ks * 16 + 8 * g) * 2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = Elem<1>::mma(wf[a], xf[b], acc[a][b]);
    }
  }
  __syncthreads();

  // ---- phase 2: + b1, bf16, zero outside the image -> hidden tile [token][channel] over the (now idle) ring --------------------
  // acc[a][b][r] = H^T[hidden (2 wn + a) 32 + crow(r, g)][token (4 wm + b) 32 + j]
  uint16_t* ht = (uint16_t*)smem;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int tok = (4 * wm + b) * 32 + j;
    const int py = y0 + (tok >> 5), px = x0 + (tok & 31);
    const bool in = py >= 0 && py < H && px >= 0 && px < W;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hc = (2 * wn + a) * 32 + 8 * c + 4 * g;             // four consecutive hidden channels
        const u32x2 bb = *(const u32x2*)(b1 + n0 + hc);
        float bf[4];
        unpack4<1>(bb, bf);
        u32x2 pk = {0u, 0u};
        if (in) pk = pack4<1>(acc[a][b][4 * c] + bf[0], acc[a][b][4 * c + 1] + bf[1], acc[a][b][4 * c + 2] + bf[2],
                              acc[a][b][4 * c + 3] + bf[3]);
        *(u32x2*)(ht + tok * kFfnHPitch + hc) = pk;
      }
  }
  __syncthreads();

  // ---- phase 3: stencil + bias + GELU on the interior.  thread = (8-channel group t & 15, strip slot t >> 4); a strip = 6 outputs
  // of one interior row: its 3 x 8 halo values feed 6 outputs (4 LDS reads per output instead of 9), packed fp32 math ------------
  const int cg = t & 15;
  f32p wk[9][4], bk[4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const f32x4 w0 = *(const f32x4*)(wdw + (long)tap * HID + n0 + cg * 8), w1 = *(const f32x4*)(wdw + (long)tap * HID + n0 + cg * 8 + 4);
    wk[tap][0] = f32p{w0[0], w0[1]};
    wk[tap][1] = f32p{w0[2], w0[3]};
    wk[tap][2] = f32p{w1[0], w1[1]};
    wk[tap][3] = f32p{w1[2], w1[3]};
  }
  {
    const f32x4 c0 = *(const f32x4*)(bdw + n0 + cg * 8), c1 = *(const f32x4*)(bdw + n0 + cg * 8 + 4);
    bk[0] = f32p{c0[0], c0[1]};
    bk[1] = f32p{c0[2], c0[3]};
    bk[2] = f32p{c1[0], c1[1]};
    bk[3] = f32p{c1[2], c1[3]};
  }
  constexpr int kStripsPerRow = kFfnIX / kFfnStrip;
  for (int st = t >> 4; st < kFfnIY * kStripsPerRow; st += 16) {
    const int iy = st / kStripsPerRow, sx = (st - iy * kStripsPerRow) * kFfnStrip;
    const int oy = ty * kFfnIY + iy, ox0 = tx * kFfnIX + sx;
    if (oy >= H || ox0 >= W) continue;
    f32p s[kFfnStrip][4];
#pragma unroll
    for (int o = 0; o < kFfnStrip; ++o)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[o][e] = bk[e];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      f32p hv[kFfnStrip + 2][4];
#pragma unroll
      for (int c = 0; c < kFfnStrip + 2; ++c) {
        const u32x4 raw = *(const u32x4*)(ht + ((iy + dy) * kFfnTX + sx + c) * kFfnHPitch + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) hv[c][e] = f32p{__uint_as_float(raw[e] << 16), __uint_as_float(raw[e] & 0xffff0000u)};
      }
#pragma unroll
      for (int o = 0; o < kFfnStrip; ++o)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[o][e] = hv[o + dx][e] * wk[dy * 3 + dx][e] + s[o][e];
    }
    uint16_t* arow = A + ((long)v * H * W + (long)oy * W + ox0) * HID + n0 + cg * 8;
#pragma unroll
    for (int o = 0; o < kFfnStrip; ++o) {
      if (ox0 + o >= W) break;
      u32x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const f32p gz = gelu2(s[o][e]);
        pk[e] = bf16x2_bits(gz[0], gz[1]);
      }
      *(u32x4*)(arow + (long)o * HID) = pk;
    }
  }
}

}  // namespace rfn

extern "C" int rfn_ffn_fc1_dw_gelu_bf16(const void* x, const void* w1, const void* b1, const float* wdw_tap, const float* bdw, void* a,
                                        int views, int H, int W, int C, int HID, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(x && w1 && b1 && wdw_tap && bdw && a, "ffn_fc1_dw_gelu: null pointer");
  RFN_REQUIRE(views > 0 && H > 0 && W > 0 && C > 0 && C % kFfnKC == 0 && HID > 0 && HID % kFfnNH == 0,
              "ffn_fc1_dw_gelu: views=%d H=%d W=%d C=%d (%% 64) HID=%d (%% 128)", views, H, W, C, HID);
  const int tiles_y = cdiv(H, kFfnIY), tiles_x = cdiv(W, kFfnIX);
  const long blocks = (long)views * tiles_y * tiles_x * (HID / kFfnNH);
  RFN_REQUIRE(blocks < (1L << 31), "ffn_fc1_dw_gelu: %ld workgroups", blocks);
  hipLaunchKernelGGL(ffn_fc1_dw_gelu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (const uint16_t*)w1, (const uint16_t*)b1, wdw_tap, bdw, (uint16_t*)a, H, W, C, HID, tiles_y, tiles_x);
  return check_launch("ffn_fc1_dw_gelu");
}
