// refign_amd/csrc/dwconv.hip -- depthwise 3x3 convolution on channels-last maps, forward + both backward passes (gfx950).
//
// Where it sits in the reference: the Mix-FFN of every MiT block runs fc1 -> DWConv 3x3 -> GELU -> fc2
// (models/backbones/mix_transformer.py:96-103,556-568: tokens are transposed to NCHW, nn.Conv2d(dim, dim, 3, 1, 1,
// groups=dim), transposed back), and the DAFormer head's ASPP has three dilated depthwise 3x3 branches on 1024 channels
// (models/heads/daformer.py:46-62).  On this stack the library path for those (MIOpen -> CK grouped conv) is a
// dense-conv kernel with group size 1 and was measured at 59 % of the whole training step; the op itself is a pure
// HBM-bound 9-tap stencil.
//
// Weights and weight gradients cross the ABI TAP-MAJOR, (9, C) fp32 (= weight.view(C, 9).t()), so that the 9 x VEC
// weights of a thread are nine contiguous vectors.
// Layout: x, y, grad tensors are (B, H, W, C) contiguous = the MiT token layout (B, N, C) itself, so no transposes at
// all; lanes run along C (16-byte vectors: 8 bf16 or 4 fp32 channels per lane => a wave reads 1 KiB contiguous), each
// thread produces 4 consecutive pixels along W and keeps its 9 x VEC weights in registers.  Halo re-reads between
// neighbouring threads/blocks are served by L1/L2.  Accumulation is fp32; weights/bias/weight-gradients stay fp32
// (master precision) while activations may be bf16.
//   forward       y[b,h,w,c]  = bias[c] + sum_{ky,kx} wgt[c,ky,kx] * x[b, h+(ky-1)d, w+(kx-1)d, c]
//   backward-data dx           = same stencil over gy with the taps flipped, no bias
//   backward-wgt  dw[c,ky,kx]  = sum_{b,h,w} gy[b,h,w,c] * x[b, h+(ky-1)d, w+(kx-1)d, c];   db[c] = sum gy
//                 (per-thread register partials over its pixel quads -> LDS tree over the block's pixel lanes ->
//                  one workspace row per stripe -> fixed-order reduction kernel: deterministic, no atomics)
#include <hip/hip_bf16.h>

#include <type_traits>

#include "common.h"
#include "mfma.h"

namespace rfn {

template <typename T>
struct VecIO;

template <>
struct VecIO<float> {
  static constexpr int N = 4;
  typedef float4 Raw;
  __device__ static Raw load_raw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  __device__ static void unpack(const Raw& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  typedef float Pair __attribute__((ext_vector_type(2)));
  __device__ static void unpack2(const Raw& t, Pair (&v)[2]) { v[0] = Pair{t.x, t.y}; v[1] = Pair{t.z, t.w}; }
  __device__ static void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ static void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

template <>
struct VecIO<__hip_bfloat16> {
  static constexpr int N = 8;
  typedef uint4 Raw;
  __device__ static Raw load_raw(const __hip_bfloat16* p) { return *reinterpret_cast<const uint4*>(p); }
  __device__ static void unpack(const Raw& t, float (&v)[8]) {
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  typedef float Pair __attribute__((ext_vector_type(2)));
  __device__ static void unpack2(const Raw& t, Pair (&v)[4]) {      // adjacent channels land in adjacent registers
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = Pair{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)};
  }
  __device__ static void load(const __hip_bfloat16* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static unsigned pack(float lo, float hi) { return bf16x2_bits(lo, hi); }   // common.h: one v_cvt_pk_bf16_f32
  __device__ static void store(__hip_bfloat16* p, const float (&v)[8]) {
    uint4 t;
    t.x = pack(v[0], v[1]); t.y = pack(v[2], v[3]); t.z = pack(v[4], v[5]); t.w = pack(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

// K5: e4m3 activations, 8 channels per lane (8-byte vectors); dequantisation / quantisation scales are kernel arguments
template <>
struct VecIO<f8e4m3> {
  static constexpr int N = 8;
  typedef uint2 Raw;
  __device__ static Raw load_raw(const f8e4m3* p) { return *reinterpret_cast<const uint2*>(p); }
  typedef float Pair __attribute__((ext_vector_type(2)));
  __device__ static void unpack2(const Raw& t, Pair (&v)[4]) {
    v[0] = __builtin_amdgcn_cvt_pk_f32_fp8((int)t.x, false);
    v[1] = __builtin_amdgcn_cvt_pk_f32_fp8((int)t.x, true);
    v[2] = __builtin_amdgcn_cvt_pk_f32_fp8((int)t.y, false);
    v[3] = __builtin_amdgcn_cvt_pk_f32_fp8((int)t.y, true);
  }
  __device__ static void store(f8e4m3* p, const float (&v)[8]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(quant4(v[0], v[1], v[2], v[3]), quant4(v[4], v[5], v[6], v[7]));
  }
};

constexpr int kPX = 4;   // pixels along W per thread

// A "quad" is 4 pixels of one image row spaced by the dilation: w0, w0+d, w0+2d, w0+3d.  Their 3x3 dilated taps fall on
// the 6 columns w0-d .. w0+4d, so 6 loads per row feed 4 outputs for ANY dilation (for d = 1 it is 4 adjacent pixels).
// Per row there are d phases x ceil(ceil(W/d)/4) quads.
__device__ __forceinline__ int quads_per_row(int W, int dil) { return dil * (((W + dil - 1) / dil + kPX - 1) / kPX); }

__device__ __forceinline__ void quad_coords(long quad, int WQ, int H, int dil, int& b, int& h, int& w0) {
  const int wq = (int)(quad % WQ);
  const long t = quad / WQ;
  h = (int)(t % H);
  b = (int)(t / H);
  w0 = (wq % dil) + (wq / dil) * kPX * dil;
}

// Thread layout (all three kernels): blockDim = 256 = cvb channel-vectors (fastest, so a wave reads contiguous
// channels) x pl pixel lanes.  A thread keeps ONE channel vector for its whole life -- its 9 x V weights are loaded
// once -- and walks "quads" (4 consecutive pixels of one image row) with stride gridDim.y * pl.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// zero a packed load when its tap is outside the image (true zero padding; 4 selects instead of one multiply per FMA)
__device__ __forceinline__ float4 mask_raw(const float4& t, bool ok) {
  return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ uint4 mask_raw(const uint4& t, bool ok) { return ok ? t : make_uint4(0u, 0u, 0u, 0u); }
__device__ __forceinline__ uint2 mask_raw(const uint2& t, bool ok) { return ok ? t : make_uint2(0u, 0u); }

// ACT: the Mix-FFN applies GELU (exact erf) right after this convolution (mix_transformer.py:99-101): with ACT the
// activation is computed on the fp32 accumulators and written to `ya`; the pre-activation goes to `y` only if that
// pointer is given (the backward needs it, a gradient-free pass does not) -- the separate GELU kernel (one more read
// and write of the 4C-wide hidden tensor) disappears.
// STATS: the BatchNorm that follows a depthwise convolution of the decode heads (daformer.py:10-62: DepthwiseSeparable ASPP
// branch = depthwise 3x3 -> BN -> ReLU -> 1x1 -> BN -> ReLU) needs the per-channel sum and sum of squares of THIS kernel's
// result: a thread owns one channel vector for its whole life, so it adds up what it stores (the ROUNDED values: what the
// statistics pass of csrc/bn.hip would have read back -- 2.65 GB for the teacher's 42 maps, 0.5 ms, per branch), the block
// folds its pixel lanes in LDS and adds to the fp64 buffer of csrc/bn.hip (sum x, sum x^2, rows).
// STATS = 2: statistics only, nothing is stored; BNE: the result is BatchNorm(batch statistics from `sums`) + ReLU of the
// convolution -- the two passes of the gradient-free form (the EMA teacher's decode head): statistics pass, then convolution
// + normalisation + activation in one pass; 3 tensor passes (read, read, write) instead of the 5 of convolution (read,
// write), statistics (read), BatchNorm (read, write).  BnEpi: gamma / beta (may be null), eps, relu, and the running buffers
// the blocks of the first row of the grid update (momentum), as csrc/bn.hip's apply pass does.
struct BnEpi {
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float eps, momentum;
  int relu;
};

template <typename T, bool FLIP, bool ACT = false, int STATS = 0, bool BNE = false>
__global__ __launch_bounds__(256) void dwconv3x3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wgt,
                                                            const float* __restrict__ bias, T* __restrict__ y, int B,
                                                            int H, int W, int C, int dil, int cvb,
                                                            T* __restrict__ ya = nullptr, float xs = 1.f,
                                                            float oq = 1.f, int sliced = 0,
                                                            double* __restrict__ sums = nullptr, BnEpi bn = BnEpi{}) {
  // xs / oq (e4m3 activations only): stored input bytes mean xs * value -- folded into the weights; outputs are stored as
  // value * oq
  constexpr bool F8 = std::is_same<T, f8e4m3>::value;
  constexpr int V = VecIO<T>::N, V2 = V / 2;
  const int CV = C / V, WQ = quads_per_row(W, dil);
  // Two launch geometries.
  //  * grid (channel blocks, quad chunks), rows in image order: the first-generation mapping.
  //  * SLICED (gridDim.y == 1, sliced != 0): a 1-D grid of 8 j blocks; block id runs on XCD id % 8 (the dispatcher's
  //    round-robin) and owns channel slice id % 8 (cvb = CV / 8 vectors) -- so the three uses of an input row (output
  //    rows h - d, h, h + d) are made by ONE XCD and the second and third find the row in that XCD's L2 -- provided they
  //    come soon enough: the blocks of an XCD walk the rows in lockstep (the grid is exactly the resident blocks), and
  //    for a dilated convolution in residue-class order r, r + d, r + 2 d, ... (h_slots = d ceil(H / d) slots per image),
  //    so the uses are 1 and 2 rows apart whatever the dilation.  With the first mapping a row's uses land on different
  //    XCDs / megabytes apart: each is a fabric read (measured round 3: tools/kbench.py --only dw).
  int pl, cv, qfirst, qstride;
  bool active;
  if (sliced) {
    pl = blockDim.x / cvb;
    active = (int)threadIdx.x < cvb * pl;
    cv = (blockIdx.x & 7) * cvb + threadIdx.x % cvb;
    qfirst = (blockIdx.x >> 3) * pl + threadIdx.x / cvb;
    qstride = (gridDim.x >> 3) * pl;
  } else {
    pl = 256 / cvb;
    cv = blockIdx.x * cvb + threadIdx.x % cvb;
    active = cv < CV;
    qfirst = blockIdx.y * pl + threadIdx.x / cvb;
    qstride = gridDim.y * pl;
  }
  if (STATS == 0 && !active) return;
  if (!active) cv = 0;                                   // (STATS: idle threads stay for the block reduction)
  const int Hd = (H + dil - 1) / dil, h_slots = sliced ? dil * Hd : H;
  const int c0 = cv * V;
  float st0[STATS ? V : 1], st1[STATS ? V : 1];
  if constexpr (STATS != 0) {
#pragma unroll
    for (int i = 0; i < V; ++i) st0[i] = st1[i] = 0.f;
  }
  float bsc[BNE ? V : 1], bsh[BNE ? V : 1];              // y = relu(conv * bsc + bsh)
  if constexpr (BNE) {
    const double cnt = sums[2 * C], inv = 1.0 / cnt;
    const bool first = sliced ? (blockIdx.x >> 3) == 0 : blockIdx.y == 0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = c0 + i;
      const double m = sums[c] * inv;
      const float var = (float)fmax(sums[C + c] * inv - m * m, 0.0), mean = (float)m;
      const float g = bn.gamma != nullptr ? bn.gamma[c] : 1.f, be = bn.beta != nullptr ? bn.beta[c] : 0.f;
      bsc[i] = rsqrtf(var + bn.eps) * g;
      bsh[i] = be - mean * bsc[i];
      if (first && active && threadIdx.x / cvb == 0 && bn.running_mean != nullptr) {
        const float n = (float)cnt;
        bn.running_mean[c] = (1.f - bn.momentum) * bn.running_mean[c] + bn.momentum * mean;
        bn.running_var[c] = (1.f - bn.momentum) * bn.running_var[c] + bn.momentum * var * (n / fmaxf(n - 1.f, 1.f));
      }
    }
  }
  // weights, bias and accumulators live as adjacent-channel PAIRS: every multiply-add below is one v_pk_fma_f32
  f32x2 wr[9][V2];   // tap-major weights (9, C): one contiguous fp32 vector per tap
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float* wp = wgt + (size_t)(FLIP ? 8 - k : k) * C + c0;
#pragma unroll
    for (int i = 0; i < V; i += 4) {
      const float4 t4 = *reinterpret_cast<const float4*>(wp + i);
      wr[k][i / 2] = f32x2{t4.x, t4.y};
      wr[k][i / 2 + 1] = f32x2{t4.z, t4.w};
      if constexpr (F8) {
        wr[k][i / 2] *= xs;
        wr[k][i / 2 + 1] *= xs;
      }
    }
  }
  f32x2 bs[V2];
#pragma unroll
  for (int i = 0; i < V2; ++i)
    bs[i] = (bias != nullptr) ? f32x2{bias[c0 + 2 * i], bias[c0 + 2 * i + 1]} : f32x2{0.0f, 0.0f};
  const long nquads = active ? (long)B * h_slots * WQ : 0;
  for (long quad = qfirst; quad < nquads; quad += qstride) {
    int b, h, w0;
    quad_coords(quad, WQ, h_slots, dil, b, h, w0);
    if (sliced && dil > 1) {                            // slot -> row of residue class slot / Hd
      h = h / Hd + (h % Hd) * dil;
      if (h >= H) continue;
    }
    f32x2 acc[kPX][V2];
#pragma unroll
    for (int p = 0; p < kPX; ++p)
#pragma unroll
      for (int i = 0; i < V2; ++i) acc[p][i] = bs[i];
    const T* xb = x + (size_t)b * H * W * C + c0;
    // branch-free window: all 18 loads (3 rows x 6 columns, clamped addresses) are issued back to back and the
    // out-of-image taps are zeroed by a select on the packed data -- per-tap `if`s made hipcc wait for each load in turn
    typename VecIO<T>::Raw raw[3][kPX + 2];            // kept packed (bf16: 4 VGPRs per 8 channels) until used
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = h + (ky - 1) * dil;
      const bool rowok = yy >= 0 && yy < H;
      const T* xr = xb + (size_t)min(max(yy, 0), H - 1) * W * C;
#pragma unroll
      for (int j = 0; j < kPX + 2; ++j) {
        const int xx = w0 + (j - 1) * dil;
        raw[ky][j] = mask_raw(VecIO<T>::load_raw(xr + (size_t)min(max(xx, 0), W - 1) * C),
                              rowok && xx >= 0 && xx < W);
      }
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int j = 0; j < kPX + 2; ++j) {
        f32x2 v[V2];
        VecIO<T>::unpack2(raw[ky][j], v);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int p = j - kx;                        // output pixel fed by column j through tap kx
          if (p < 0 || p >= kPX) continue;
#pragma unroll
          for (int i = 0; i < V2; ++i) acc[p][i] = __builtin_elementwise_fma(wr[ky * 3 + kx][i], v[i], acc[p][i]);
        }
      }
    const size_t obase = ((size_t)b * H + h) * W * C + c0;
#pragma unroll
    for (int p = 0; p < kPX; ++p)
      if (w0 + p * dil < W) {
        float o[V];
#pragma unroll
        for (int i = 0; i < V2; ++i) { o[2 * i] = acc[p][i].x; o[2 * i + 1] = acc[p][i].y; }
        if constexpr (BNE) {
          // the statistics were taken of the ROUNDED convolution result (what the unfused path stores and reads back)
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float r = sizeof(T) == 2 ? __uint_as_float(bf16_bits(o[i]) << 16) : o[i];
            const float z = fmaf(r, bsc[i], bsh[i]);
            o[i] = (bn.relu && z <= 0.f) ? 0.f : z;
          }
        }
        if (STATS != 2 && (!ACT || y != nullptr)) VecIO<T>::store(y + obase + (size_t)(w0 + p * dil) * C, o);
        if constexpr (STATS != 0) {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            const float r = sizeof(T) == 2 ? __uint_as_float(bf16_bits(o[i]) << 16) : o[i];
            st0[i] += r;
            st1[i] = fmaf(r, r, st1[i]);
          }
        }
        if (ACT) {
#pragma unroll
          for (int i = 0; i < V; ++i) {
            // fp32 activations (parity mode): libm's erff; 16-bit / e4m3 results: mfma.h gelu_erf_fast
            if constexpr (sizeof(T) == 4) o[i] = 0.5f * o[i] * (1.f + erff(o[i] * 0.70710678118654752440f));
            else o[i] = gelu_erf_fast(o[i]) * (F8 ? oq : 1.f);
          }
          VecIO<T>::store(ya + obase + (size_t)(w0 + p * dil) * C, o);
        }
      }
  }
  if constexpr (STATS != 0) {
    __shared__ float sred[2][256][V + 1];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      sred[0][threadIdx.x][i] = active ? st0[i] : 0.f;
      sred[1][threadIdx.x][i] = active ? st1[i] : 0.f;
    }
    __syncthreads();
    // thread (which, channel vector of the block, element): fold the pixel lanes, one fp64 atomic per channel and block
    const int cvbase = sliced ? (blockIdx.x & 7) * cvb : blockIdx.x * cvb;
    for (int idx = threadIdx.x; idx < 2 * cvb * V; idx += blockDim.x) {
      const int which = idx / (cvb * V), rem = idx % (cvb * V), v = rem / V, e = rem % V;
      if (cvbase + v >= CV) continue;
      float sum = 0.f;
      for (int r = 0; r < pl; ++r) sum += sred[which][r * cvb + v][e];
      atomicAdd(sums + which * C + (cvbase + v) * V + e, (double)sum);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) atomicAdd(sums + 2 * C, (double)B * H * W);
  }
}

// backward-weight, stage 1: register partials over the thread's quads, LDS tree over the block's pixel lanes, one
// partial row per (stripe = blockIdx.y) in the workspace: ws[stripe][k][C], k = 0..8 taps, 9 = bias.  No atomics.
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_bwd_weight_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                   float* __restrict__ ws, int B, int H, int W, int C,
                                                                   int dil, int cvb) {
  constexpr int V = VecIO<T>::N;
  const int CV = C / V, WQ = quads_per_row(W, dil);
  const int pl = 256 / cvb;
  const int cvi = threadIdx.x % cvb, pli = threadIdx.x / cvb;
  const int cv = blockIdx.x * cvb + cvi;
  const bool active = cv < CV;
  const int c0 = cv * V;
  constexpr int V2 = V / 2;
  f32x2 aw[9][V2], ab[V2];          // adjacent-channel pairs: every multiply-add is one v_pk_fma_f32
#pragma unroll
  for (int i = 0; i < V2; ++i) {
    ab[i] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 9; ++k) aw[k][i] = f32x2{0.0f, 0.0f};
  }
  if (active) {
    const long nquads = (long)B * H * WQ;
    for (long quad = (long)blockIdx.y * pl + pli; quad < nquads; quad += (long)gridDim.y * pl) {
      int b, h, w0;
      quad_coords(quad, WQ, H, dil, b, h, w0);
      f32x2 g[kPX][V2];
      const T* gp = gy + (((size_t)b * H + h) * W) * C + c0;
      {
        typename VecIO<T>::Raw graw[kPX];
#pragma unroll
        for (int p = 0; p < kPX; ++p)     // pixels past the row end contribute zero (select on the packed data)
          graw[p] = mask_raw(VecIO<T>::load_raw(gp + (size_t)min(w0 + p * dil, W - 1) * C), w0 + p * dil < W);
#pragma unroll
        for (int p = 0; p < kPX; ++p) {
          VecIO<T>::unpack2(graw[p], g[p]);
#pragma unroll
          for (int i = 0; i < V2; ++i) ab[i] += g[p][i];
        }
      }
      const T* xb = x + (size_t)b * H * W * C + c0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int yy = h + (ky - 1) * dil;
        const bool rowok = yy >= 0 && yy < H;
        const T* xr = xb + (size_t)min(max(yy, 0), H - 1) * W * C;
        typename VecIO<T>::Raw raw[kPX + 2];
#pragma unroll
        for (int j = 0; j < kPX + 2; ++j) {
          const int xx = w0 + (j - 1) * dil;
          raw[j] = mask_raw(VecIO<T>::load_raw(xr + (size_t)min(max(xx, 0), W - 1) * C), rowok && xx >= 0 && xx < W);
        }
#pragma unroll
        for (int j = 0; j < kPX + 2; ++j) {
          f32x2 v[V2];
          VecIO<T>::unpack2(raw[j], v);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int p = j - kx;
            if (p < 0 || p >= kPX) continue;
#pragma unroll
            for (int i = 0; i < V2; ++i) aw[ky * 3 + kx][i] = __builtin_elementwise_fma(g[p][i], v[i], aw[ky * 3 + kx][i]);
          }
        }
      }
    }
  }
  // fold the block's pixel lanes: two passes of 5 of the 10 rows (9 taps + bias) through 40 KB of LDS, every thread sums
  // (10 one-row passes with two barriers each and only the first pixel lane summing were a 3-5 us tail)
  __shared__ float red[5][256][V];
  float* wrow = ws + (size_t)blockIdx.y * 10 * C;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
      const int k = pass * 5 + kk;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const f32x2 t = (k < 9) ? aw[k < 9 ? k : 0][i / 2] : ab[i / 2];
        red[kk][threadIdx.x][i] = (i & 1) ? t.y : t.x;
      }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 5 * cvb * V; o += 256) {
      const int kk = o / (cvb * V), rem = o - kk * cvb * V, cvj = rem / V, i = rem - cvj * V;
      if (blockIdx.x * cvb + cvj >= CV) continue;
      float sum = 0.0f;
      for (int p = 0; p < pl; ++p) sum += red[kk][p * cvb + cvj][i];
      wrow[(size_t)(pass * 5 + kk) * C + (blockIdx.x * cvb + cvj) * V + i] = sum;
    }
  }
}

// stage 2: dw[k][c] = sum over stripes (fixed order => deterministic); k = 9 -> bias gradient.
// flags bit0: accumulate into dw/db (they are the parameters' .grad); bit1: dw in parameter layout (C,1,3,3) = [c][k]
// instead of tap-major [k][c]
__global__ __launch_bounds__(256) void dwconv3x3_bwd_weight_reduce_kernel(const float* __restrict__ ws,
                                                                          float* __restrict__ dw,
                                                                          float* __restrict__ db, int C, int stripes,
                                                                          int flags) {
  // 32 columns x 8 stripe segments per workgroup (the stripe dimension has to supply parallelism: 10*C columns are
  // only 50 workgroups of 256 at C = 1280), LDS combine in a fixed order
  __shared__ float part[8][32];
  const int lane = threadIdx.x & 31, seg = threadIdx.x >> 5;
  const int idx = blockIdx.x * 32 + lane;
  float s = 0.0f;
  if (idx < 10 * C) {
#pragma unroll 4
    for (int t = seg; t < stripes; t += 8) s += ws[(size_t)t * 10 * C + idx];
  }
  part[seg][lane] = s;
  __syncthreads();
  if (seg != 0 || idx >= 10 * C) return;
  s = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
      ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
  float* dst;
  if (idx < 9 * C) {
    const int k = idx / C, c = idx - k * C;
    dst = (flags & 2) ? dw + c * 9 + k : dw + idx;
  } else {
    if (db == nullptr) return;
    dst = db + (idx - 9 * C);
  }
  *dst = (flags & 1) ? *dst + s : s;
}

static inline int pick_cvb(int CV) { return CV >= 64 ? 64 : (CV >= 32 ? 32 : (CV >= 16 ? 16 : 8)); }

// the SLICED geometry of dwconv3x3_fwd_kernel: one channel slice of CV / 8 vectors per XCD (at least 8 vectors = 128-byte
// runs per pixel), and exactly the blocks that are resident at once -- 2 per CU (216 VGPRs: 2 waves per SIMD), 64 per XCD
// (or twice that: no difference measured) -- so that an XCD's blocks advance through the rows together.
// RFN_DWCONV_SLICED=0: first-generation grid everywhere.
struct SlicedGeom {
  bool on;
  int cvb, grid;
};
// Measured (round 3, tools/kbench.py --only dw, first-generation grid -> sliced): ASPP 40 x 135 x 240 x 1024, dilation 6:
// 1884 -> 1451 us (1422 with 128 blocks per XCD, 1706 with 32); fp32, dilation 12: 361 -> 272 us; 4 x 34 x 60 x 1280: 19.1 ->
// 14.4 us; Mix-FFN + GELU 40 x 34 x 60 x 1280: 160 -> 149 us.  NOT for slices under 16 vectors (512 channels: 128-byte runs
// per pixel, 214 -> 240 us).
static inline SlicedGeom sliced_geom(int CV, long nslots) {
  static const int enabled = 1;
  // workgroups per XCD: 128 was the optimum of the isolated launches; inside the step (three streams share the CUs) 64 is
  // 0.6 ms per step better (152.8 vs 153.4, twice), 32 is 7 ms worse
  static const int per_xcd = 64;
  if (!enabled || CV % 8 != 0 || CV / 8 < 16 || CV / 8 > 64) return SlicedGeom{false, 0, 0};
  const int cvb = CV / 8, pl = 256 / cvb;
  const int j = (int)std::max<long>(1, std::min<long>(cdiv(nslots, pl), per_xcd));
  return SlicedGeom{true, cvb, 8 * j};
}
constexpr int kMaxStripes = 128;

template <typename T>
static int launch_fwd_gelu(const void* x, const float* w, const float* bias, void* y, void* ya, int B, int H, int W,
                           int C, hipStream_t st, float xs = 1.f, float oq = 1.f) {
  constexpr int V = VecIO<T>::N;
  const int CV = C / V;
  const long nquads = (long)B * H * ((W + kPX - 1) / kPX);
  if (SlicedGeom sg = sliced_geom(CV, nquads); sg.on) {
    hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false, true>), dim3(sg.grid), dim3(256), 0, st, (const T*)x, w, bias,
                       (T*)y, B, H, W, C, 1, sg.cvb, (T*)ya, xs, oq, 1);
    return check_launch("dwconv3x3_fwd_kernel<gelu, sliced>");
  }
  const int cvb = pick_cvb(CV), gx = cdiv(CV, cvb), pl = 256 / cvb;
  const int gy = (int)std::max<long>(1, std::min<long>(cdiv(nquads, pl), (256L * 16) / gx));
  hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false, true>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, w, bias,
                     (T*)y, B, H, W, C, 1, cvb, (T*)ya, xs, oq);
  return check_launch("dwconv3x3_fwd_kernel<gelu>");
}

// STATS 1: convolution + statistics; 2: statistics only; BNE: convolution + BatchNorm + activation from complete statistics
template <typename T, int STATS, bool BNE>
static int launch_fwd_stats(const void* x, const float* w, const float* bias, void* y, double* sums, int B, int H, int W, int C,
                            int dil, hipStream_t st, BnEpi bn = BnEpi{}) {
  constexpr int V = VecIO<T>::N;
  const int CV = C / V;
  const int WQ = dil * (((W + dil - 1) / dil + kPX - 1) / kPX);
  if (SlicedGeom sg = sliced_geom(CV, (long)B * dil * ((H + dil - 1) / dil) * WQ); sg.on) {
    hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false, false, STATS, BNE>), dim3(sg.grid), dim3(256), 0, st, (const T*)x, w, bias,
                       (T*)y, B, H, W, C, dil, sg.cvb, (T*)nullptr, 1.f, 1.f, 1, sums, bn);
    return check_launch("dwconv3x3_fwd_kernel<sliced, stats / bn>");
  }
  const int cvb = pick_cvb(CV), gx = cdiv(CV, cvb), pl = 256 / cvb;
  const long nquads = (long)B * H * WQ;
  const int gy = (int)std::max<long>(1, std::min<long>(cdiv(nquads, pl), (256L * 16) / gx));
  hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false, false, STATS, BNE>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, w, bias,
                     (T*)y, B, H, W, C, dil, cvb, (T*)nullptr, 1.f, 1.f, 0, sums, bn);
  return check_launch("dwconv3x3_fwd_kernel<stats / bn>");
}

template <typename T>
static int launch_fwd(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int dil,
                      int flip, hipStream_t st) {
  constexpr int V = VecIO<T>::N;
  const int CV = C / V;
  const int WQ = dil * (((W + dil - 1) / dil + kPX - 1) / kPX);
  if (SlicedGeom sg = sliced_geom(CV, (long)B * dil * ((H + dil - 1) / dil) * WQ); sg.on) {
    if (flip)
      hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, true>), dim3(sg.grid), dim3(256), 0, st, (const T*)x, w, bias, (T*)y, B,
                         H, W, C, dil, sg.cvb, (T*)nullptr, 1.f, 1.f, 1);
    else
      hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false>), dim3(sg.grid), dim3(256), 0, st, (const T*)x, w, bias, (T*)y,
                         B, H, W, C, dil, sg.cvb, (T*)nullptr, 1.f, 1.f, 1);
    return check_launch("dwconv3x3_fwd_kernel<sliced>");
  }
  const int cvb = pick_cvb(CV), gx = cdiv(CV, cvb), pl = 256 / cvb;
  const long nquads = (long)B * H * WQ;
  const int gy = (int)std::max<long>(1, std::min<long>(cdiv(nquads, pl), (256L * 16) / gx));
  if (flip)
    hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, true>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, w, bias, (T*)y, B,
                       H, W, C, dil, cvb);
  else
    hipLaunchKernelGGL((dwconv3x3_fwd_kernel<T, false>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, w, bias, (T*)y,
                       B, H, W, C, dil, cvb);
  return check_launch("dwconv3x3_fwd_kernel");
}

template <typename T>
static int launch_bwd_weight(const void* x, const void* gy, float* dw, float* db, float* ws, int B, int H, int W, int C,
                             int dil, int flags, hipStream_t st) {
  constexpr int V = VecIO<T>::N;
  // narrow channel blocks on LARGE maps: with 16 vectors (256 contiguous bytes per pixel) a block has 16 pixel lanes instead
  // of 4 -- four times the workgroups for the same partial-sum volume (the workspace row of a stripe is 10 C floats however
  // the channels are split).  Measured (tools/kbench.py --only dw, forward + backward): 40 x 135 x 240 x 256: 1 393 -> 1 070 us;
  // per-call census of the step (tools/abi_census.py): 4 x 68 x 120 x 512: 47 -> 24 us, 4 x 135 x 240 x 256: 78 -> 48 us; the
  // student's stage 3 / 4 maps (4 x 34 x 60 x 1280: 27 vs 26 us) keep the wide blocks.  RFN_DWCONV_WGRAD_CVB forces a width.
  static const int force_cvb = 0;
  const int CV = C / V;
  const long nquads = (long)B * H * (dil * (((W + dil - 1) / dil + kPX - 1) / kPX));
  const int want_cvb = force_cvb ? force_cvb : (nquads >= 8000 ? 16 : 0);
  const int cvb = (want_cvb >= 8 && want_cvb <= 64 && (want_cvb & (want_cvb - 1)) == 0 && CV >= want_cvb) ? want_cvb : pick_cvb(CV);
  const int gx = cdiv(CV, cvb), pl = 256 / cvb;
  const int stripes = (int)std::max<long>(1, std::min<long>(std::min<long>(kMaxStripes, cdiv(nquads, pl)),
                                                            std::max<long>(1, (256L * 8) / gx)));
  hipLaunchKernelGGL((dwconv3x3_bwd_weight_kernel<T>), dim3(gx, stripes), dim3(256), 0, st, (const T*)x, (const T*)gy,
                     ws, B, H, W, C, dil, cvb);
  if (int rc = check_launch("dwconv3x3_bwd_weight_kernel")) return rc;
  hipLaunchKernelGGL(dwconv3x3_bwd_weight_reduce_kernel, dim3(cdiv(10L * C, 32)), dim3(256), 0, st, ws, dw, db, C,
                     stripes, flags);
  return check_launch("dwconv3x3_bwd_weight_reduce_kernel");
}


// ---------------------------------------------------------------------------------------------------------------------
// Three dilated depthwise branches of ONE input in one pass -- the EMA teacher's ASPP (daformer.py:46-62,65-126: dilations
// 6 / 12 / 18 on the 40 x 135 x 240 x 1024 concatenated feature map; round 3 read that 2.65 GB map six times: three statistics
// passes and three convolution + BatchNorm + ReLU passes, each bound by L2 -> CU traffic: 4.5 loads per output, nothing of a
// dilated window is shared through L1).  Dilations g, 2g, 3g are plain / 2- / 3-dilated 3x3 convolutions of the g x g PHASE
// sub-images (pixels (py + g i, px + g j)): a workgroup takes one phase of one image for 32 channels -- at most 23 x 40 pixels
// x 64 bytes -- into LDS ONCE (every input byte is read from L2 / HBM exactly once) and computes all three branches from there:
// statistics pass (sum, sum of squares of the rounded results of the three branches, nothing stored) and apply pass (convolution
// + BatchNorm(batch statistics) + ReLU, three outputs).  76 KB of LDS: TWO workgroups per CU, one loading while the other
// computes (the first version -- 64 channels, 151 KB, one workgroup of 4 waves per CU -- was latency-bound end to end: 47 us per
// workgroup, slower than six single-branch passes).  The two 32-channel halves of a 64-channel group are taken by workgroups
// that the dispatcher places on the SAME XCD 8 blocks apart, so that the 128-byte lines they share cross the fabric once.
// LDS pixel pitch 80 bytes: the 16 lanes of a ds_read_b128 group (4 channel vectors x 4 quads, 4 pixels apart) cover all banks.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTriPitch = 80, kTriMaxPix = 944, kTriScratch = 4 * 4 * 16 * 4;

struct TriArgs {
  const float* w;            // [3][9][C] tap-major fp32
  const float* bias;         // [3][C] or null
  double* sums;              // [3][2 C + 1]
  const float* gamma[3];     // apply pass (may be null)
  const float* beta[3];
  float* running_mean[3];    // may be null
  float* running_var[3];
  float eps[3], momentum[3];
  __hip_bfloat16* y[3];
  int relu, ablate;
};

template <int M, int MODE>
__device__ __forceinline__ void tri_branch(const unsigned char* __restrict__ img, float* __restrict__ scratch, const TriArgs& a,
                                           int b, int B, int H, int W, int C, int g, int py, int px, int Hs, int Ws, int c0, int cv,
                                           int pl, bool first_block, bool stat_block) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int k = M - 1;
  const int c = c0 + cv * 8;
  f2 wr[9][4], bs[4];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* wp = a.w + ((size_t)k * 9 + t) * C + c;
    const float4 lo = *reinterpret_cast<const float4*>(wp), hi = *reinterpret_cast<const float4*>(wp + 4);
    wr[t][0] = f2{lo.x, lo.y}; wr[t][1] = f2{lo.z, lo.w}; wr[t][2] = f2{hi.x, hi.y}; wr[t][3] = f2{hi.z, hi.w};
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    bs[i] = a.bias != nullptr ? f2{a.bias[(size_t)k * C + c + 2 * i], a.bias[(size_t)k * C + c + 2 * i + 1]} : f2{0.f, 0.f};
  float bsc[8], bsh[8], st0[8], st1[8];
  if constexpr (MODE == 1) {
    const double* sm = a.sums + (size_t)k * (2 * C + 1);
    const double cnt = sm[2 * C], inv = 1.0 / cnt;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double m = sm[c + i] * inv;
      const float var = (float)fmax(sm[C + c + i] * inv - m * m, 0.0), mean = (float)m;
      const float ga = a.gamma[k] != nullptr ? a.gamma[k][c + i] : 1.f, be = a.beta[k] != nullptr ? a.beta[k][c + i] : 0.f;
      bsc[i] = rsqrtf(var + a.eps[k]) * ga;
      bsh[i] = be - mean * bsc[i];
      if (stat_block && pl == 0 && a.running_mean[k] != nullptr) {
        const float n = (float)cnt;
        a.running_mean[k][c + i] = (1.f - a.momentum[k]) * a.running_mean[k][c + i] + a.momentum[k] * mean;
        a.running_var[k][c + i] = (1.f - a.momentum[k]) * a.running_var[k][c + i] + a.momentum[k] * var * (n / fmaxf(n - 1.f, 1.f));
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) st0[i] = st1[i] = 0.f;
  }
  const int QW = (Ws + 3) >> 2, nq = Hs * QW;
  for (int q = pl; q < nq; q += 64) {
    const int ys = q / QW, x0 = 4 * (q - ys * QW);
    f2 acc[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[p][i] = bs[i];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = ys + (ky - 1) * M;
      const bool rowok = yy >= 0 && yy < Hs;
      const unsigned char* rowp = img + (size_t)min(max(yy, 0), Hs - 1) * Ws * kTriPitch + cv * 16;
#pragma unroll
      for (int o = 0; o < 4 + 2 * M; ++o) {
        const int xx = x0 - M + o;
        uint4 raw = *reinterpret_cast<const uint4*>(rowp + min(max(xx, 0), Ws - 1) * kTriPitch);
        if (!(rowok && xx >= 0 && xx < Ws)) raw = make_uint4(0u, 0u, 0u, 0u);
        f2 v[4];
        VecIO<__hip_bfloat16>::unpack2(raw, v);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int d = o - p;                           // column offset of this load relative to output p, + M
          if (d != 0 && d != M && d != 2 * M) continue;
          const int kx = d / M;
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[p][i] = __builtin_elementwise_fma(wr[ky * 3 + kx][i], v[i], acc[p][i]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (x0 + p >= Ws) continue;
      float o[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // the unfused path stores the convolution result in bf16 before the statistics / BatchNorm read it back
        o[2 * i] = __uint_as_float(bf16_bits(acc[p][i].x) << 16);
        o[2 * i + 1] = __uint_as_float(bf16_bits(acc[p][i].y) << 16);
      }
      if constexpr (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          st0[i] += o[i];
          st1[i] = fmaf(o[i], o[i], st1[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float z = fmaf(o[i], bsc[i], bsh[i]);
          o[i] = (a.relu && z <= 0.f) ? 0.f : z;
        }
        VecIO<__hip_bfloat16>::store(a.y[k] + (((size_t)b * H + py + g * ys) * W + px + g * (x0 + p)) * C + c, o);
      }
    }
  }
  if constexpr (MODE == 0) {
    // lanes cv + 4 j of a wave hold partial sums of the same 8 channels: fold j, then the four waves through LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int sft = 4; sft < 64; sft <<= 1) {
        st0[i] += __shfl_xor(st0[i], sft, 64);
        st1[i] += __shfl_xor(st1[i], sft, 64);
      }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        scratch[(wave * 4 + lane) * 16 + i] = st0[i];
        scratch[(wave * 4 + lane) * 16 + 8 + i] = st1[i];
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int v = threadIdx.x >> 4, e = threadIdx.x & 15;
      const float sum = scratch[(0 * 4 + v) * 16 + e] + scratch[(1 * 4 + v) * 16 + e] + scratch[(2 * 4 + v) * 16 + e] +
                        scratch[(3 * 4 + v) * 16 + e];
      atomicAdd(a.sums + (size_t)k * (2 * C + 1) + (e >> 3) * C + c0 + v * 8 + (e & 7), (double)sum);
    }
    if (first_block && threadIdx.x == 0) atomicAdd(a.sums + (size_t)k * (2 * C + 1) + 2 * C, (double)B * H * W);
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void dwconv3x3_tri_kernel(const __hip_bfloat16* __restrict__ x, TriArgs a, int B, int H, int W,
                                                               int C, int g, int nitems) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kTriMaxPix * kTriPitch + kTriScratch];
  // block L runs on XCD L % 8 (the dispatcher's round-robin; speed only): the two halves of item i are blocks 8 apart on one XCD
  const int L = blockIdx.x, local = L >> 3, half = local & 1, item = (local >> 1) * 8 + (L & 7);
  if (item >= nitems) return;
  const int gg = g * g, phase = item % gg, rest = item / gg, n64 = C / 64, b = rest / n64, c0 = (rest % n64) * 64 + half * 32;
  const int py = phase / g, px = phase % g;
  const int Hs = (H - py + g - 1) / g, Ws = (W - px + g - 1) / g;
  const int cv = threadIdx.x & 3, pl = threadIdx.x >> 2;
  const __hip_bfloat16* xb = x + (size_t)b * H * W * C + c0 + cv * 8;
  const int npix = Hs * Ws;
  // all of the thread's loads first (up to 15 in flight), then the LDS writes
  constexpr int NL = (kTriMaxPix + 63) / 64;
  uint4 v[NL];
#pragma unroll
  for (int it = 0; it < NL; ++it) {
    const int p = min(pl + 64 * it, npix - 1), ys = p / Ws, xs = p - ys * Ws;
    v[it] = (a.ablate & 1) ? uint4{0, 0, 0, 0} : *reinterpret_cast<const uint4*>(xb + ((size_t)(py + g * ys) * W + (px + g * xs)) * C);
  }
#pragma unroll
  for (int it = 0; it < NL; ++it) {
    const int p = pl + 64 * it;
    if (p < npix) *reinterpret_cast<uint4*>(lds + (size_t)p * kTriPitch + cv * 16) = v[it];
  }
  __syncthreads();
  float* scratch = reinterpret_cast<float*>(lds + kTriMaxPix * kTriPitch);
  const bool first_block = item == 0 && half == 0, stat_block = phase == 0 && b == 0;
  if (a.ablate & 2) return;
  tri_branch<1, MODE>(lds, scratch, a, b, B, H, W, C, g, py, px, Hs, Ws, c0, cv, pl, first_block, stat_block);
  if (a.ablate & 4) return;
  tri_branch<2, MODE>(lds, scratch, a, b, B, H, W, C, g, py, px, Hs, Ws, c0, cv, pl, first_block, stat_block);
  tri_branch<3, MODE>(lds, scratch, a, b, B, H, W, C, g, py, px, Hs, Ws, c0, cv, pl, first_block, stat_block);
}

static int tri_domain(int B, int H, int W, int C, int g) {
  return B > 0 && g >= 1 && g <= H && g <= W && C % 64 == 0 && cdiv(H, g) * cdiv(W, g) <= kTriMaxPix &&
         (long)g * g * (C / 64) * B < (1L << 26);
}

}  // namespace rfn

using namespace rfn;

extern "C" {
// Three dilated depthwise 3x3 branches (dilations g, 2 g, 3 g; padding = dilation) of one bf16 NHWC input in ONE pass each:
//   rfn_dwconv3x3_tri_stats        sums3 [3][2 C + 1] doubles <- (sum, sum of squares, rows) of the three rounded results;
//   rfn_dwconv3x3_tri_bn_act_fwd   y[k] = act(bn_k(conv_k(x))) with the statistics in sums3 (a SyncBatchNorm all-reduces them in
//                                  between).  weight3: [3][9][C] tap-major fp32, bias3: [3][C] or NULL; gamma / beta / running_mean /
//                                  running_var / y / eps / momentum: HOST arrays of 3.  C % 64 == 0, ceil(H / g) ceil(W / g) <= 944.
// rfn_dwconv3x3_tri_usable: 1 when a shape is inside that domain.
int rfn_dwconv3x3_tri_usable(int B, int H, int W, int C, int g) { return tri_domain(B, H, W, C, g); }

int rfn_dwconv3x3_tri_stats(const void* x, const float* weight3, const float* bias3, double* sums3, int B, int H, int W, int C, int g,
                            rfn_stream_t stream) {
  RFN_REQUIRE(x && weight3 && sums3, "rfn_dwconv3x3_tri_stats: null pointer");
  RFN_REQUIRE(tri_domain(B, H, W, C, g), "rfn_dwconv3x3_tri_stats: B=%d H=%d W=%d C=%d g=%d outside the kernel's domain", B, H, W, C, g);
  hipStream_t st = (hipStream_t)stream;
  if (int rc = zero_async(sums3, 3 * (2 * (size_t)C + 1) * sizeof(double), st)) return rc;
  TriArgs a{};
  a.w = weight3, a.bias = bias3, a.sums = sums3;
  a.ablate = 0;
  const long nitems = (long)g * g * (C / 64) * B;
  hipLaunchKernelGGL((dwconv3x3_tri_kernel<0>), dim3((unsigned)(16 * cdiv(nitems, 8))), dim3(256), 0, st, (const __hip_bfloat16*)x, a,
                     B, H, W, C, g, (int)nitems);
  return check_launch("dwconv3x3_tri_kernel<stats>");
}

int rfn_dwconv3x3_tri_bn_act_fwd(const void* x, const float* weight3, const float* bias3, const float* const* gamma3,
                                 const float* const* beta3, const double* sums3, float* const* running_mean3,
                                 float* const* running_var3, void* const* y3, int B, int H, int W, int C, int g, const float* eps3,
                                 const float* momentum3, int relu, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight3 && sums3 && y3 && gamma3 && beta3 && running_mean3 && running_var3 && eps3 && momentum3,
              "rfn_dwconv3x3_tri_bn_act_fwd: null pointer");
  RFN_REQUIRE(tri_domain(B, H, W, C, g), "rfn_dwconv3x3_tri_bn_act_fwd: B=%d H=%d W=%d C=%d g=%d outside the kernel's domain", B, H, W, C, g);
  TriArgs a{};
  a.w = weight3, a.bias = bias3, a.sums = const_cast<double*>(sums3), a.relu = relu;
  a.ablate = 0;
  for (int k = 0; k < 3; ++k) {
    RFN_REQUIRE(y3[k], "rfn_dwconv3x3_tri_bn_act_fwd: null output %d", k);
    a.gamma[k] = gamma3[k], a.beta[k] = beta3[k], a.running_mean[k] = running_mean3[k], a.running_var[k] = running_var3[k];
    a.eps[k] = eps3[k], a.momentum[k] = momentum3[k], a.y[k] = (__hip_bfloat16*)y3[k];
  }
  const long nitems = (long)g * g * (C / 64) * B;
  hipLaunchKernelGGL((dwconv3x3_tri_kernel<1>), dim3((unsigned)(16 * cdiv(nitems, 8))), dim3(256), 0, (hipStream_t)stream,
                     (const __hip_bfloat16*)x, a, B, H, W, C, g, (int)nitems);
  return check_launch("dwconv3x3_tri_kernel<apply>");
}


int rfn_dwconv3x3_nhwc_fwd(const void* x, const float* weight, const float* bias, void* y, int B, int H, int W, int C,
                           int dilation, int dtype, int flip, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight && y, "rfn_dwconv3x3_nhwc_fwd: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "rfn_dwconv3x3_nhwc_fwd: bad size");
  if (dtype == 0) {
    RFN_REQUIRE(C % 4 == 0, "rfn_dwconv3x3_nhwc_fwd: C must be a multiple of 4 for f32 (got %d)", C);
    return launch_fwd<float>(x, weight, bias, y, B, H, W, C, dilation, flip, (hipStream_t)stream);
  }
  if (dtype == 1) {
    RFN_REQUIRE(C % 8 == 0, "rfn_dwconv3x3_nhwc_fwd: C must be a multiple of 8 for bf16 (got %d)", C);
    return launch_fwd<__hip_bfloat16>(x, weight, bias, y, B, H, W, C, dilation, flip, (hipStream_t)stream);
  }
  return fail(RFN_EINVAL, "rfn_dwconv3x3_nhwc_fwd: dtype must be 0 (f32) or 1 (bf16)");
}

// rfn_dwconv3x3_nhwc_fwd (bf16) that also leaves the BatchNorm statistics of its result in `sums` (2 C + 1 doubles: sum,
// sum of squares, rows -- the buffer of rfn_bn_stats_fwd, zeroed here): the statistics pass over the result is not needed
int rfn_dwconv3x3_nhwc_fwd_stats(const void* x, const float* weight, const float* bias, void* y, double* sums, int B, int H,
                                 int W, int C, int dilation, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight && y && sums, "rfn_dwconv3x3_nhwc_fwd_stats: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "rfn_dwconv3x3_nhwc_fwd_stats: bad size");
  RFN_REQUIRE(dtype == 1 && C % 8 == 0, "rfn_dwconv3x3_nhwc_fwd_stats: bf16 (dtype 1), C %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = zero_async(sums, (2 * (size_t)C + 1) * sizeof(double), st)) return rc;
  return launch_fwd_stats<__hip_bfloat16, 1, false>(x, weight, bias, y, sums, B, H, W, C, dilation, st);
}

// Gradient-free depthwise 3x3 -> BatchNorm(batch statistics) -> ReLU in two passes over the INPUT (bf16):
//   rfn_dwconv3x3_nhwc_stats        sums <- (sum, sum of squares, rows) of the rounded convolution result, nothing stored;
//   rfn_dwconv3x3_bn_act_nhwc_fwd   y = act(bn(conv(x))) with the statistics in `sums` (between the two a SyncBatchNorm
//                                   all-reduces `sums`); running_mean / running_var (may be NULL) updated as rfn_bn_apply_fwd.
int rfn_dwconv3x3_nhwc_stats(const void* x, const float* weight, const float* bias, double* sums, int B, int H, int W, int C,
                             int dilation, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight && sums, "rfn_dwconv3x3_nhwc_stats: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "rfn_dwconv3x3_nhwc_stats: bad size");
  RFN_REQUIRE(dtype == 1 && C % 8 == 0, "rfn_dwconv3x3_nhwc_stats: bf16 (dtype 1), C %% 8 == 0");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = zero_async(sums, (2 * (size_t)C + 1) * sizeof(double), st)) return rc;
  return launch_fwd_stats<__hip_bfloat16, 2, false>(x, weight, bias, nullptr, sums, B, H, W, C, dilation, st);
}

int rfn_dwconv3x3_bn_act_nhwc_fwd(const void* x, const float* weight, const float* bias, const float* gamma, const float* beta,
                                  const double* sums, float* running_mean, float* running_var, void* y, int B, int H, int W,
                                  int C, int dilation, float eps, float momentum, int relu, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight && sums && y, "rfn_dwconv3x3_bn_act_nhwc_fwd: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "rfn_dwconv3x3_bn_act_nhwc_fwd: bad size");
  RFN_REQUIRE(dtype == 1 && C % 8 == 0 && (relu == 0 || relu == 1), "rfn_dwconv3x3_bn_act_nhwc_fwd: bf16, C %% 8 == 0, relu 0 / 1");
  BnEpi bn{gamma, beta, running_mean, running_var, eps, momentum, relu};
  return launch_fwd_stats<__hip_bfloat16, 0, true>(x, weight, bias, y, const_cast<double*>(sums), B, H, W, C, dilation,
                                                   (hipStream_t)stream, bn);
}

int rfn_dwconv3x3_gelu_nhwc_fwd(const void* x, const float* weight, const float* bias, void* y_pre, void* y_act, int B,
                                int H, int W, int C, int dtype, rfn_stream_t stream) {
  RFN_REQUIRE(x && weight && y_act, "rfn_dwconv3x3_gelu_nhwc_fwd: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "rfn_dwconv3x3_gelu_nhwc_fwd: bad size");
  if (dtype == 0) {
    RFN_REQUIRE(C % 4 == 0, "rfn_dwconv3x3_gelu_nhwc_fwd: C must be a multiple of 4 for f32 (got %d)", C);
    return launch_fwd_gelu<float>(x, weight, bias, y_pre, y_act, B, H, W, C, (hipStream_t)stream);
  }
  if (dtype == 1) {
    RFN_REQUIRE(C % 8 == 0, "rfn_dwconv3x3_gelu_nhwc_fwd: C must be a multiple of 8 for bf16 (got %d)", C);
    return launch_fwd_gelu<__hip_bfloat16>(x, weight, bias, y_pre, y_act, B, H, W, C, (hipStream_t)stream);
  }
  return fail(RFN_EINVAL, "rfn_dwconv3x3_gelu_nhwc_fwd: dtype must be 0 (f32) or 1 (bf16)");
}

int rfn_dwconv3x3_gelu_nhwc_fwd_f8(const void* x8, const float* weight, const float* bias, void* y8, int B, int H, int W, int C,
                                   float x_scale, float out_q, rfn_stream_t stream) {
  RFN_REQUIRE(x8 && weight && y8, "rfn_dwconv3x3_gelu_nhwc_fwd_f8: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "rfn_dwconv3x3_gelu_nhwc_fwd_f8: bad size (C %% 8)");
  return launch_fwd_gelu<f8e4m3>(x8, weight, bias, nullptr, y8, B, H, W, C, (hipStream_t)stream, x_scale, out_q);
}

unsigned long rfn_dwconv3x3_bwd_weight_workspace_bytes(int C) {
  return (unsigned long)kMaxStripes * 10ul * (unsigned long)(C > 0 ? C : 0) * sizeof(float);
}

int rfn_dwconv3x3_nhwc_bwd_weight(const void* x, const void* grad_y, float* grad_weight, float* grad_bias,
                                  void* workspace, int B, int H, int W, int C, int dilation, int dtype, int flags,
                                  rfn_stream_t stream) {
  RFN_REQUIRE(x && grad_y && grad_weight && workspace, "rfn_dwconv3x3_nhwc_bwd_weight: null pointer");
  RFN_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && dilation > 0, "rfn_dwconv3x3_nhwc_bwd_weight: bad size");
  if (dtype == 0) {
    RFN_REQUIRE(C % 4 == 0, "rfn_dwconv3x3_nhwc_bwd_weight: C must be a multiple of 4 for f32");
    return launch_bwd_weight<float>(x, grad_y, grad_weight, grad_bias, (float*)workspace, B, H, W, C, dilation,
                                    flags, (hipStream_t)stream);
  }
  if (dtype == 1) {
    RFN_REQUIRE(C % 8 == 0, "rfn_dwconv3x3_nhwc_bwd_weight: C must be a multiple of 8 for bf16");
    return launch_bwd_weight<__hip_bfloat16>(x, grad_y, grad_weight, grad_bias, (float*)workspace, B, H, W, C, dilation,
                                             flags, (hipStream_t)stream);
  }
  return fail(RFN_EINVAL, "rfn_dwconv3x3_nhwc_bwd_weight: dtype must be 0 (f32) or 1 (bf16)");
}

}  // extern "C"
