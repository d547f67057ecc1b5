// tools/micro/tr_probe.hip -- what ds_read_b64_tr_b16 returns.  LDS holds the element index (uint16) of a [64 rows][64 cols]
// row-major image; every lane issues ONE transpose read at the address of (row, col) chosen by `mode` and the host prints
// the 4 values each lane received, decoded back to (row, col).
//   hipcc -O2 --offload-arch=gfx950 tools/micro/tr_probe.hip -o refign_amd/lib/ab/tr_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t img[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) img[i] = (uint16_t)i;     // value = row * 64 + col
  __syncthreads();
  const int l = threadIdx.x, p = l & 15, q = l >> 4;
  int row, col;
  if (mode == 0) {          // lane p of a 16-lane group points at row p / 4, cols 4 (p % 4) .. + 3 of a [4][16] block;
    row = 4 * (q >> 1) + p / 4;             // group q takes the block at cols 16 (q & 1), rows 4 (q >> 1)
    col = 16 * (q & 1) + 4 * (p % 4);
  } else if (mode == 1) {   // lane p points at row p (16 rows), 4 contiguous cols of column block q
    row = p;
    col = 4 * q;
  } else {                  // every lane its own row l, cols 0..3
    row = l;
    col = 0;
  }
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) uint16_t*)(img + row * 64 + col);
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[2 * l] = r[0];
  out[2 * l + 1] = r[1];
}

int main() {
  unsigned* d;
  hipMalloc(&d, 128 * 4);
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    unsigned h[128];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (lane: 4 x (row,col) received)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        const unsigned v = (h[2 * l + j / 2] >> (16 * (j & 1))) & 0xffff;
        printf(" (%2u,%2u)", v / 64, v % 64);
      }
      printf("\n");
    }
  }
  return 0;
}
