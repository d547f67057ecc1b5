// tools/micro/cu_fetch_rate.hip -- how fast ONE CU can pull bytes from L2 / MALL / HBM: every workgroup streams its own slice of
// a buffer (a) through LDS-DMA (global_load_lds_dwordx4, the GEMM / correlation / attention staging path) and (b) through plain
// global_load_dwordx4 into registers, with W waves per workgroup and one workgroup per CU.  Footprints: 1 MB per XCD slice set
// (L2-resident after the first pass), 64 MB (MALL), 2 GB (HBM).  Prints GB/s per CU and chip-wide.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/cu_fetch_rate.hip -o refign_amd/lib/ab/cu_fetch_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e));                                \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_wave_base) {
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_wave_base;
  const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(sbase) : "memory", "m0");
}

// each wave: `iters` rounds of `UNROLL` x 1 KB, wrapping inside its workgroup's slice of `slice` bytes
template <int UNROLL>
__global__ __launch_bounds__(1024) void dma_kernel(const char* __restrict__ buf, size_t slice, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = buf + (size_t)blockIdx.x * slice;
  size_t off = (size_t)wave * UNROLL * 1024;
  char* dst = smem + wave * UNROLL * 1024;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) lds_dma16(base + off + u * 1024 + lane * 16, dst + u * 1024);
    off += (size_t)nw * UNROLL * 1024;
    if (off + UNROLL * 1024 > slice) off = (size_t)wave * UNROLL * 1024;
    // keep at most 2 rounds in flight (what a 2-3 stage ring does)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UNROLL) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = *(float*)smem;
}

template <int UNROLL>
__global__ __launch_bounds__(1024) void reg_kernel(const char* __restrict__ buf, size_t slice, int iters, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const char* base = buf + (size_t)blockIdx.x * slice;
  size_t off = (size_t)wave * UNROLL * 1024;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = *(const float4*)(base + off + u * 1024 + lane * 16);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
    off += (size_t)nw * UNROLL * 1024;
    if (off + UNROLL * 1024 > slice) off = (size_t)wave * UNROLL * 1024;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f && sink) sink[blockIdx.x] = acc.x;
}

template <typename K>
static double run(K kernel, const char* buf, size_t slice, int wgs, int waves, int iters, size_t lds, float* sink) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(wgs), dim3(waves * 64), lds, 0, buf, slice, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3;
}

int main() {
  const int wgs = 256;
  const size_t total = (size_t)2 << 30;
  char* buf;
  float* sink;
  CHECK(hipMalloc(&buf, total));
  CHECK(hipMemset(buf, 1, total));
  CHECK(hipMalloc(&sink, 4096));
  constexpr int U = 8;
  printf("# one workgroup per CU (256), UNROLL %d KB per wave and round, <= 2 rounds in flight per wave\n", U);
  printf("# path     slice/WG   waves   GB/s per CU   TB/s chip\n");
  for (size_t slice : {(size_t)128 << 10, (size_t)1 << 20, (size_t)8 << 20}) {
    for (int waves : {4, 8, 16}) {
      const size_t per_round = (size_t)waves * U * 1024;
      const int iters = (int)(((size_t)64 << 20) / per_round);     // 64 MB per workgroup
      const double bytes = (double)iters * per_round * wgs;
      double t = run(dma_kernel<U>, buf, slice, wgs, waves, iters, (size_t)waves * U * 1024, sink);
      printf("lds-dma   %6zu KB   %4d   %10.1f   %8.2f\n", slice >> 10, waves, bytes / t / wgs / 1e9, bytes / t / 1e12);
      t = run(reg_kernel<U>, buf, slice, wgs, waves, iters, 0, sink);
      printf("vgpr      %6zu KB   %4d   %10.1f   %8.2f\n", slice >> 10, waves, bytes / t / wgs / 1e9, bytes / t / 1e12);
      fflush(stdout);
    }
  }
  // few CUs active: is the limit per CU or chip-wide?
  for (int n : {8, 32, 64}) {
    const int waves = 8;
    const size_t per_round = (size_t)waves * U * 1024;
    const int iters = (int)(((size_t)64 << 20) / per_round);
    const double bytes = (double)iters * per_round * n;
    double t = run(dma_kernel<U>, buf, (size_t)1 << 20, n, waves, iters, (size_t)waves * U * 1024, sink);
    printf("lds-dma   1 MB slices, only %3d workgroups, 8 waves: %8.1f GB/s per CU\n", n, bytes / t / n / 1e9);
    t = run(reg_kernel<U>, buf, (size_t)1 << 20, n, waves, iters, 0, sink);
    printf("vgpr      1 MB slices, only %3d workgroups, 8 waves: %8.1f GB/s per CU\n", n, bytes / t / n / 1e9);
  }
  return 0;
}
