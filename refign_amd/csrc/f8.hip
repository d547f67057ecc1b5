// refign_amd/csrc/f8.hip -- K5 (BASELINE.json config 5): fp8 matrix-core path of the gradient-free EMA teacher.
//
// The EMA teacher of a Refign step (segmentation_model.py:204-209) runs MiT-B5 on 40 HRDA views per GPU with no
// gradient: its token-wise Linear layers (mix_transformer.py:96-103,137-164) and its attention core (:150-160) run here
// on the gfx950 fp8 matrix instruction v_mfma_f32_32x32x64_f8f6f4 (OCP e4m3, fp32 accumulate; 2x the bf16 MFMA rate, half
// the operand bytes).  There is no reference analogue: the reference's own recipe is 16-bit AMP.
//
//   rfn_gemm_nt_f8     Y[M,N] = epi( xs * ws[n] * (X8[M,K] . W8[N,K]^T) + bias )   X8, W8 e4m3; Y bf16 (+ residual, per-
//                      sample scale) or e4m3 (quantised in the epilogue for the next fp8 consumer)
//   rfn_quant_rows_f8  multi-tensor weight quantisation: bf16 rows -> e4m3 rows + one fp32 scale per row (amax / 448)
//   rfn_attn_pack_f8   K / V of one fp8 kv tensor -> per-64-key stages in MFMA operand order
//   rfn_attn_fwd_f8    O8 = softmax(scale Q8 K8^T) V8, fp32 softmax, probabilities re-quantised to e4m3 (x 256)
//
// Scaling scheme: weights carry one fp32 scale per OUTPUT ROW (amax -> 448); activations carry ONE power-of-two scale per
// producer site, a kernel argument (`x_scale` = what a stored byte must be multiplied by, `out_q` = what a value is
// multiplied by before it is stored): LayerNorm / GELU / attention outputs are O(1) tensors, q = 8 keeps |x| < 56 in
// range and pushes the subnormal threshold to 2e-3 (DESIGN.md section 4.3 states the measured agreement bound).
//
// GEMM kernel: the LDS-DMA ring, swizzled 128-byte rows, transposed 32x32 tiles, persistent schedule and LDS-staged
// epilogue of mfma_gemm.hip; a row of 128 bytes is now 128 k, walked by two K = 64 instructions whose operand is two
// ds_read_b128 (pieces 4 ks + 2 g and + 1 of the row).  K need not be a multiple of 128: pieces past K read a zero page.
#include <hip/hip_bf16.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "mfma.h"

namespace rfn {

typedef __attribute__((ext_vector_type(8))) int i32x8;

__device__ __forceinline__ f32x16 mma_f8(i32x8 a, i32x8 b, f32x16 c) {
  // scale operands 0 / 0: the backend selects the unscaled v_mfma_f32_32x32x64_f8f6f4 (cbsz = blgp = 0: e4m3 x e4m3)
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}
__device__ __forceinline__ i32x8 join32(u32x4 lo, u32x4 hi) {
  i32x8 r = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  return r;
}
struct F8Epi {
  const float* wscale;      // [N] fp32: dequantisation scale of weight row n
  float xscale;             // dequantisation scale of the activation bytes
  const uint16_t* bias;     // [N] bf16 or null
  const uint16_t* res;      // [M, ldy] bf16 residual (bf16 output only) or null
  const float* rowscale;    // per-sample scale of the branch before the residual add, or null
  int rows_per_sample;
  int act;                  // 0 none, 1 ReLU
  float outq;               // fp8 output: value * outq is stored
};

__device__ uint4 g_zero_page_f8[4];

template <int N> __device__ __forceinline__ void wait_dma_upto_f8() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// OUT8: e4m3 output (1 byte per element, ldy in bytes); otherwise bf16 (ldy in elements)
template <int BM, int BN, bool OUT8, int NW = 4>
__global__ __launch_bounds__(NW * 64) void gemm_nt_f8_kernel(const unsigned char* __restrict__ X,
                                                         const unsigned char* __restrict__ W, void* __restrict__ Yv, int M,
                                                         int N, int K, long ldx, long ldw, long ldy, int tiles_n,
                                                         int total_tiles, F8Epi epi, const void* zero) {
  constexpr int NS = 2;
  // waves as WMW (m) x WNW (n).  An e4m3 result row of a wave's tile should be a full 128-byte line: with e4m3 output the
  // waves are stacked along m only (4 waves) / 4 x 2 (8 waves), so that a wave owns >= 128 output columns where BN allows
  constexpr int WNW = (OUT8 && BM >= 128) ? NW / 4 : NW / 2, WMW = NW / WNW;
  constexpr int IB = BN / (32 * WNW), JB = BM / (32 * WMW);
  constexpr int ROWB = 128, PPR = 8, RPI = 8;
  constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB, STAGE = XBYTES + WBYTES;
  constexpr int XI = BM / (NW * RPI), WI = BN / (NW * RPI);
  auto swizzle = [](int r) { return (r >> 1) & 7; };
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int t_begin = wg;
  if (t_begin >= total_tiles) return;
  const int ntiles = (total_tiles - wg + G - 1) / G;
  const int nk = (K + ROWB - 1) / ROWB;

  const int drow = lane / PPR;
  const int dpiece = (lane % PPR) ^ swizzle(RPI * wave + drow);
  const unsigned char* xsrc[XI];
  const unsigned char* wsrc[WI];
  auto setup = [&](int tile) {
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
#pragma unroll
    for (int q = 0; q < XI; ++q) {
      const int m = min(m0 + RPI * (NW * q + wave) + drow, M - 1);
      xsrc[q] = X + (long)m * ldx + 16 * dpiece;
    }
#pragma unroll
    for (int q = 0; q < WI; ++q) {
      const int n = min(n0 + RPI * (NW * q + wave) + drow, N - 1);
      wsrc[q] = W + (long)n * ldw + 16 * dpiece;
    }
  };
  auto issue = [&](int kt, int buf) {
    unsigned char* xs = smem + buf * STAGE;
    unsigned char* ws = xs + XBYTES;
    const bool in = kt * ROWB + 16 * dpiece < K;          // K % 16 == 0: a piece is all in or all out
#pragma unroll
    for (int q = 0; q < XI; ++q)
      lds_dma16(in ? xsrc[q] + (long)kt * ROWB : (const unsigned char*)zero, xs + 1024 * (NW * q + wave));
#pragma unroll
    for (int q = 0; q < WI; ++q)
      lds_dma16(in ? wsrc[q] + (long)kt * ROWB : (const unsigned char*)zero, ws + 1024 * (NW * q + wave));
  };

  f32x16 acc[IB][JB];
#pragma unroll
  for (int i = 0; i < IB; ++i)
#pragma unroll
    for (int j = 0; j < JB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int g = lane >> 5, frow = lane & 31, swz = swizzle(frow);
  const int xoff = (wm * (BM / WMW) + frow) * ROWB, woff = XBYTES + (wn * (BN / WNW) + frow) * ROWB;

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
  produce();

  int c_tile = t_begin, c_kt = 0;
  for (long s = 0; s < S; ++s) {
    wait_dma_all();
    wg_barrier();
    produce();
    const unsigned char* st = smem + (int)(s % NS) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // k-slots of lane (row, g) in this instruction: the 32 bytes of pieces 4 ks + 2 g, 4 ks + 2 g + 1 of its row --
      // the A and the B operand agree, which is all the dot product needs
      const int c0 = ((4 * ks + 2 * g) ^ swz) * 16, c1 = ((4 * ks + 2 * g + 1) ^ swz) * 16;
      i32x8 wf[IB], xf[JB];
#pragma unroll
      for (int i = 0; i < IB; ++i)
        wf[i] = join32(*(const u32x4*)(st + woff + i * 32 * ROWB + c0), *(const u32x4*)(st + woff + i * 32 * ROWB + c1));
#pragma unroll
      for (int j = 0; j < JB; ++j)
        xf[j] = join32(*(const u32x4*)(st + xoff + j * 32 * ROWB + c0), *(const u32x4*)(st + xoff + j * 32 * ROWB + c1));
#pragma unroll
      for (int i = 0; i < IB; ++i)
#pragma unroll
        for (int j = 0; j < JB; ++j) acc[i][j] = mma_f8(wf[i], xf[j], acc[i][j]);
    }
    if (++c_kt == nk) {
      int tl = c_tile;
      asm volatile("" : "+s"(tl));
      const int m0 = (tl / tiles_n) * BM, n0 = (tl % tiles_n) * BN;
      constexpr int WN = BN / WNW;                     // columns of a wave's tile
      constexpr int EB = OUT8 ? 1 : 2;                 // bytes per output element
      constexpr int PITCH = WN * EB + 16;
      constexpr int PCS = WN * EB / 16;                // 16-byte pieces per staged row
      constexpr int RPP = 64 / PCS;                    // rows per store instruction
      static_assert(NW * 32 * PITCH <= STAGE, "staging block fits the consumed stage");
      wg_barrier();
      unsigned char* stg = const_cast<unsigned char*>(st) + wave * 32 * PITCH;
#pragma unroll
      for (int j = 0; j < JB; ++j) {
#pragma unroll
        for (int i = 0; i < IB; ++i) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int cl = i * 32 + 8 * k + 4 * g;
            const int n = n0 + wn * WN + cl;
            float v[4] = {acc[i][j][4 * k], acc[i][j][4 * k + 1], acc[i][j][4 * k + 2], acc[i][j][4 * k + 3]};
            if (n < N) {                                  // N % 8 == 0: a run is all in or all out
              const f32x4 sc = *(const f32x4*)(epi.wscale + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= sc[e] * epi.xscale;
              if (epi.bias != nullptr) {
                float b[4];
                unpack4<1>(*(const u32x2*)(epi.bias + n), b);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += b[e];
              }
            }
            if (epi.act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (OUT8) {
              *(unsigned*)(stg + frow * PITCH + cl) = quant4(v[0] * epi.outq, v[1] * epi.outq, v[2] * epi.outq, v[3] * epi.outq);
            } else {
              *(u32x2*)(stg + frow * PITCH + cl * 2) = pack4<1>(v[0], v[1], v[2], v[3]);
            }
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 32 / RPP; ++it) {
          const int row = it * RPP + lane / PCS, piece = lane % PCS;
          const int m = m0 + wm * (BM / WMW) + j * 32 + row, n = n0 + wn * WN + piece * (16 / EB);
          u32x4 o = *(const u32x4*)(stg + row * PITCH + piece * 16);
          if (m < M && n < N) {
            if constexpr (OUT8) {
              *(u32x4*)((unsigned char*)Yv + (long)m * ldy + n) = o;
            } else {
              uint16_t* Y = (uint16_t*)Yv;
              if (epi.res != nullptr || epi.rowscale != nullptr) {
                const float rs = epi.rowscale != nullptr ? epi.rowscale[m / epi.rows_per_sample] : 1.f;
                const u32x4 rr = epi.res != nullptr ? *(const u32x4*)(epi.res + (long)m * ldy + n) : u32x4{0u, 0u, 0u, 0u};
                float a[4], b[4], c[4], d[4];
                unpack4<1>(u32x2{o[0], o[1]}, a);
                unpack4<1>(u32x2{o[2], o[3]}, b);
                unpack4<1>(u32x2{rr[0], rr[1]}, c);
                unpack4<1>(u32x2{rr[2], rr[3]}, d);
                const u32x2 lo = pack4<1>(c[0] + rs * a[0], c[1] + rs * a[1], c[2] + rs * a[2], c[3] + rs * a[3]);
                const u32x2 hi = pack4<1>(d[0] + rs * b[0], d[1] + rs * b[1], d[2] + rs * b[2], d[3] + rs * b[3]);
                o = u32x4{lo[0], lo[1], hi[0], hi[1]};
              }
              *(u32x4*)(Y + (long)m * ldy + n) = o;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      c_kt = 0;
      c_tile += G;
    }
  }
}

template <bool OUT8>
static int launch_nt_f8(const void* X, const void* W, void* Y, long M, long N, long K, long ldx, long ldw, long ldy,
                        const F8Epi& epi, hipStream_t s) {
  static void* zero_page = nullptr;
  if (zero_page == nullptr && hipGetSymbolAddress(&zero_page, HIP_SYMBOL(g_zero_page_f8)) != hipSuccess)
    return fail(RFN_ELAUNCH, "gemm_nt_f8: zero page symbol");
  int bn = (N % 128 == 0) ? 128 : 64;
  int bm = ((long)cdiv(M, 128) * cdiv(N, bn) >= 256) ? 128 : 64;
  if (N % 256 == 0 && (long)cdiv(M, 256) * (N / 256) >= 256) bm = bn = 256;
  static const char* cfg_env = nullptr;       // "bm,bn": tile sweep (tools/f8_bench.py)
  if (cfg_env != nullptr) sscanf(cfg_env, "%d,%d", &bm, &bn);
  const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn);
  const long total = (long)tiles_m * tiles_n;
  const int ring = 2 * (bm + bn) * 128;
  const int per_cu = std::max(1, std::min(160 * 1024 / ring, 4));
  const bool persistent = K <= 512;                   // few K-steps per tile: the next tile's loads hide under the epilogue
  const int slots = persistent ? 256 * per_cu : 0x7fffffff;
  dim3 grid((unsigned)std::min<long>(total, slots)), block(256);
#define RFN_NT8(BM_, BN_)                                                                                                \
  hipLaunchKernelGGL((gemm_nt_f8_kernel<BM_, BN_, OUT8>), grid, block, 0, s, (const unsigned char*)X,                    \
                     (const unsigned char*)W, Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, (int)total, epi,        \
                     (const void*)zero_page)
  const int key = bm * 1000 + bn;
  switch (key) {
    case 128128: RFN_NT8(128, 128); break;
    case 128064: RFN_NT8(128, 64); break;
    case 64128: RFN_NT8(64, 128); break;
    case 64064: RFN_NT8(64, 64); break;
    case 256256:
      grid = dim3((unsigned)std::min<long>(total, persistent ? 256 : 0x7fffffff));
      hipLaunchKernelGGL((gemm_nt_f8_kernel<256, 256, OUT8, 8>), grid, dim3(512), 0, s, (const unsigned char*)X,
                         (const unsigned char*)W, Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tiles_n, (int)total, epi,
                         (const void*)zero_page);
      break;
    default: return fail(RFN_EINVAL, "gemm_nt_f8: no kernel for tile %dx%d", bm, bn);
  }
#undef RFN_NT8
  return check_launch("gemm_nt_f8");
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-tensor row quantisation (weights): one wave per row.  Table row = {src (bf16 row 0), dst (e4m3 row 0), scales,
// K | nrows << 32}; chunks of 4 rows, one workgroup each.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quant_rows_f8_kernel(const long* __restrict__ table, int nchunks) {
  const int chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  const long* e = table + (long)chunk * 4;
  const uint16_t* src = (const uint16_t*)e[0];
  unsigned char* dst = (unsigned char*)e[1];
  float* scales = (float*)e[2];
  const int K = (int)(e[3] & 0xffffffffL), nrows = (int)(e[3] >> 32);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave >= nrows) return;
  const uint16_t* row = src + (long)wave * K;
  float amax = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    float f[4];
    unpack4<1>(*(const u32x2*)(row + k), f);
    amax = fmaxf(fmaxf(fabsf(f[0]), fabsf(f[1])), fmaxf(fmaxf(fabsf(f[2]), fabsf(f[3])), amax));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax / kF8Max : 1.f;
  const float q = 1.f / sc;
  for (int k = lane * 4; k < K; k += 256) {
    float f[4];
    unpack4<1>(*(const u32x2*)(row + k), f);
    *(unsigned*)(dst + (long)wave * K + k) = quant4(f[0] * q, f[1] * q, f[2] * q, f[3] * q);
  }
  if (lane == 0) scales[wave] = sc;
}

// activation quantisation bf16 -> e4m3 with one scale (tests / entry of the fp8 chain)
__global__ __launch_bounds__(256) void quant_f8_kernel(const uint16_t* __restrict__ x, unsigned char* __restrict__ y,
                                                       long n4, float q) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float f[4];
    unpack4<1>(*(const u32x2*)(x + 4 * i), f);
    *(unsigned*)(y + 4 * i) = quant4(f[0] * q, f[1] * q, f[2] * q, f[3] * q);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention, fp8.  Everything transposed as in attn.hip: S^T = K Q^T (ONE K = 64 instruction per 32-key block: the head
// dimension is the reduction), O^T = V^T P^T (one instruction per 64-key stage and 32-wide d block: the 64 keys of a
// stage are the reduction), a lane owns one query.  The 32 probabilities a lane holds after the two S^T blocks of a stage
// (C/D registers r = 0..15 of block 0, then of block 1) ARE the 32 k-slots of its B operand for the PV product; the V
// pack stores V^T in that slot order.
//
// Pack of one 64-key stage (8 192 bytes), written by attn_pack_f8_kernel from the fp8 kv tensor (B, Nkv, 2 * heads * 64):
//   K: [kb (2 key blocks)][g][h][32 keys][16 B]   bytes d = 32 g + 16 h + 0..15 of key 32 kb + row
//   V: [db (2 d blocks)][g][h][32 d][16 B]        byte e of the piece = V[key = 32 h + (e & 3) + 8 (e >> 2) + 4 g][d = 32 db + row]
// Both are read as two conflict-free ds_read_b128 per operand.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kStage8 = 8192;

__global__ __launch_bounds__(256) void attn_pack_f8_kernel(const unsigned char* __restrict__ kv, long sb, long sr, int heads,
                                                           int Nkv, int nst, unsigned char* __restrict__ pack) {
  __shared__ __attribute__((aligned(16))) unsigned char vt[64][80];      // V tile [key][d], padded rows
  const int st = blockIdx.x, bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int t = threadIdx.x;
  const unsigned char* base = kv + (long)b * sb + hd * 64;
  const long voff = (long)heads * 64;                                      // V follows K inside a token
  unsigned char* out = pack + ((long)bh * nst + st) * kStage8;
  {   // K: one 16-byte piece per thread
    const int kb = t >> 7, g = (t >> 6) & 1, h = (t >> 5) & 1, row = t & 31;
    const int key = 64 * st + 32 * kb + row;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (key < Nkv) v = *(const u32x4*)(base + (long)key * sr + 32 * g + 16 * h);
    *(u32x4*)(out + t * 16) = v;
  }
  {   // V tile -> LDS
    const int key = t >> 2, c = t & 3;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (64 * st + key < Nkv) v = *(const u32x4*)(base + voff + (long)(64 * st + key) * sr + 16 * c);
    *(u32x4*)(&vt[key][16 * c]) = v;
  }
  __syncthreads();
  {
    const int db = t >> 7, g = (t >> 6) & 1, h = (t >> 5) & 1, row = t & 31;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned x = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) x |= (unsigned)vt[32 * h + e + 8 * q + 4 * g][32 * db + row] << (8 * e);
      w[q] = x;
    }
    *(u32x4*)(out + 4096 + t * 16) = u32x4{w[0], w[1], w[2], w[3]};
  }
}

constexpr float kLog2e8 = 1.4426950408889634f;
constexpr float kPScale = 256.f;            // probabilities (<= 1) are stored as p * 256 in e4m3

__global__ __launch_bounds__(256) void attn_fwd_f8_kernel(const unsigned char* __restrict__ Q, long qsb, long qsr,
                                                          const unsigned char* __restrict__ pack,
                                                          unsigned char* __restrict__ O, long osb, long osr, int heads,
                                                          int Nq, int Nkv, int nst, float c, float omul) {
  // c = softmax scale * q_scale * k_scale * log2(e) (scores in the base-2 domain); omul = v_scale * out_q / 256
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kStage8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 5, col = lane & 31;
  const int bh = blockIdx.y, b = bh / heads, hd = bh % heads;
  const int q = blockIdx.x * 128 + wave * 32 + col;
  const bool ok = q < Nq;

  i32x8 qf;
  {
    const unsigned char* p = Q + (long)b * qsb + (long)(ok ? q : 0) * qsr + hd * 64 + 32 * g;
    u32x4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
    if (ok) {
      lo = *(const u32x4*)p;
      hi = *(const u32x4*)(p + 16);
    }
    qf = join32(lo, hi);
  }
  pin_loaded(qf);                                          // mfma.h: no compiler wait inside the DMA loop
  const unsigned char* pbase = pack + (long)bh * nst * kStage8;
  auto issue = [&](int st, int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int piece = wave * 2 + u;                    // 8 x 1 KB per stage
      lds_dma16(pbase + (long)st * kStage8 + piece * 1024 + lane * 16, smem + buf * kStage8 + piece * 1024);
    }
  };
  f32x16 oacc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[0][r] = oacc[1][r] = 0.f;
  float mrun = -1e30f, lrun = 0.f;

  issue(0, 0);
  wait_dma_all();
  wg_barrier();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    const unsigned char* ks = smem + buf * kStage8;
    const unsigned char* vs = ks + 4096;
    const int key0 = st * 64;
    float p[2][16];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const unsigned char* kp = ks + kb * 2048 + (g * 64 + col) * 16;
      const i32x8 kf = join32(*(const u32x4*)kp, *(const u32x4*)(kp + 512));
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      const f32x16 sacc = mma_f8(kf, qf, z);
#pragma unroll
      for (int r = 0; r < 16; ++r) p[kb][r] = sacc[r];
    }
    if (key0 + 64 > Nkv) {
      asm volatile("; padded keys" ::: "memory");          // (a branch, not selects in every stage: csrc/attn.hip)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (key0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * g >= Nkv) p[kb][r] = -1e30f;
    }
    float mt = p[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, p[kb][r]);
    mt = half_max(mt) * c;
    const float mnew = fmaxf(mrun, mt);
    if (__any(mnew > mrun)) {
      const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
      lrun *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        oacc[0][r] *= alpha;
        oacc[1][r] *= alpha;
      }
      mrun = mnew;
    }
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        p[kb][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(p[kb][r], c, -mrun));
        ps += p[kb][r];
      }
    lrun += ps;
    i32x8 pb;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        pb[kb * 4 + k] = (int)quant4(p[kb][4 * k] * kPScale, p[kb][4 * k + 1] * kPScale, p[kb][4 * k + 2] * kPScale,
                                     p[kb][4 * k + 3] * kPScale);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const unsigned char* vp = vs + db * 2048 + (g * 64 + col) * 16;
      const i32x8 vf = join32(*(const u32x4*)vp, *(const u32x4*)(vp + 512));
      oacc[db] = mma_f8(vf, pb, oacc[db]);
    }
    wait_dma_all();
    wg_barrier();
  }
  const float l = half_sum(lrun);
  const float mul = omul / l;
  unsigned char* orow = O + (long)b * osb + (long)(ok ? q : 0) * osr + hd * 64;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    unsigned dw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dw[k] = quant4(oacc[db][4 * k] * mul, oacc[db][4 * k + 1] * mul, oacc[db][4 * k + 2] * mul, oacc[db][4 * k + 3] * mul);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      auto r = __builtin_amdgcn_permlane32_swap(dw[2 * pp], dw[2 * pp + 1], false, false);
      if (ok) *(u32x2*)(orow + db * 32 + 16 * pp + 8 * g) = u32x2{r[0], r[1]};
    }
  }
}

}  // namespace rfn

extern "C" {

int rfn_gemm_nt_f8(const void* X8, const void* W8, const float* wscale, float x_scale, const void* bias, const void* res,
                   const float* rowscale, int rows_per_sample, int act, void* Y, int out_f8, float out_q, long M, long N,
                   long K, long ldx, long ldw, long ldy, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(X8 && W8 && wscale && Y, "gemm_nt_f8: null operand");
  RFN_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0 && N % 16 == 0, "gemm_nt_f8: M=%ld N=%ld K=%ld (K %% 16, N %% 16)", M, N, K);
  RFN_REQUIRE(ldx % 16 == 0 && ldw % 16 == 0 && ldx >= K && ldw >= K && ldy >= N && ldy % (out_f8 ? 16 : 8) == 0,
              "gemm_nt_f8: leading dimensions (ldx, ldw %% 16 bytes; ldy %% 16 bytes)");
  RFN_REQUIRE(M < (1L << 31) && N < (1L << 31), "gemm_nt_f8: extent");
  RFN_REQUIRE(rowscale == nullptr || rows_per_sample > 0, "gemm_nt_f8: rowscale needs rows_per_sample");
  RFN_REQUIRE(act == 0 || act == 1, "gemm_nt_f8: act (0 none, 1 ReLU)");
  RFN_REQUIRE(!out_f8 || (res == nullptr && rowscale == nullptr), "gemm_nt_f8: residual needs a bf16 output");
  F8Epi epi{wscale, x_scale, (const uint16_t*)bias, (const uint16_t*)res, rowscale,
            rows_per_sample > 0 ? rows_per_sample : 1, act, out_q};
  hipStream_t s = (hipStream_t)stream;
  return out_f8 ? launch_nt_f8<true>(X8, W8, Y, M, N, K, ldx, ldw, ldy, epi, s)
                : launch_nt_f8<false>(X8, W8, Y, M, N, K, ldx, ldw, ldy, epi, s);
}

int rfn_quant_rows_f8(const void* table, int nchunks, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(table && nchunks > 0, "quant_rows_f8: empty table");
  hipLaunchKernelGGL(quant_rows_f8_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const long*)table, nchunks);
  return check_launch("quant_rows_f8");
}

int rfn_quant_f8(const void* x_bf16, void* y8, long n, float q, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(x_bf16 && y8 && n > 0 && n % 4 == 0, "quant_f8: n=%ld (%% 4)", n);
  const long n4 = n / 4;
  const int grid = (int)std::min<long>((n4 + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(quant_f8_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                     (unsigned char*)y8, n4, q);
  return check_launch("quant_f8");
}

int rfn_attn_pack_f8(const void* kv8, long batch_stride, long row_stride, int B, int heads, int Nkv, int nst, void* pack,
                     rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(kv8 && pack && B > 0 && heads > 0 && Nkv > 0 && nst * 64 >= Nkv, "attn_pack_f8: bad arguments");
  RFN_REQUIRE(row_stride % 16 == 0 && batch_stride % 16 == 0, "attn_pack_f8: strides (%% 16 bytes)");
  hipLaunchKernelGGL(attn_pack_f8_kernel, dim3(nst, B * heads), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)kv8, batch_stride, row_stride, heads, Nkv, nst, (unsigned char*)pack);
  return check_launch("attn_pack_f8");
}

int rfn_attn_fwd_f8(const void* q8, long q_batch_stride, long q_row_stride, const void* pack, void* o8, long o_batch_stride,
                    long o_row_stride, int B, int heads, int Nq, int Nkv, int nst, float scale, float q_scale, float k_scale,
                    float v_scale, float out_q, rfn_stream_t stream) {
  using namespace rfn;
  RFN_REQUIRE(q8 && pack && o8 && B > 0 && heads > 0 && Nq > 0 && Nkv > 0 && nst * 64 >= Nkv, "attn_fwd_f8: bad arguments");
  RFN_REQUIRE(q_row_stride % 16 == 0 && q_batch_stride % 16 == 0 && o_row_stride % 8 == 0 && o_batch_stride % 8 == 0,
              "attn_fwd_f8: strides");
  const float c = scale * q_scale * k_scale * kLog2e8, omul = v_scale * out_q / kPScale;
  hipLaunchKernelGGL(attn_fwd_f8_kernel, dim3(cdiv(Nq, 128), B * heads), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)q8, q_batch_stride, q_row_stride, (const unsigned char*)pack,
                     (unsigned char*)o8, o_batch_stride, o_row_stride, heads, Nq, Nkv, nst, c, omul);
  return check_launch("attn_fwd_f8");
}

}  // extern "C"
