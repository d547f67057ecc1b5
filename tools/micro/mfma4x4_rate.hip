// tools/micro/mfma4x4_rate.hip -- issue rate of v_mfma_f32_4x4x1_16B_f32 on gfx950: NACC independent accumulators,
// W waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 mfma4x4_rate.hip -o /tmp/mfma4x4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void rate(float* out, int iters, long long* cyc) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<NACC>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep == 1) {
      const double n = (double)iters * NACC * (threads / 64 / 4.0);   // MFMAs per SIMD
      printf("NACC %3d  waves/SIMD %.0f: %.2f clk64 ticks per MFMA per SIMD, %.3f ms -> %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n",
             NACC, threads / 256.0, (double)c / n, ms, ms * 1e6 / n, 512.0 * n * 4 * 256 / (ms * 1e-3) / 1e12);
    }
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<8>(256); run<8>(512); run<27>(256); run<27>(512); run<54>(256); run<54>(512); run<54>(1024); run<16>(1024);
  return 0;
}
