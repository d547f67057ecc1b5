// tools/micro/wg_placement.hip -- where does the dispatcher put the workgroups of a grid that does not fill the chip?
// Every workgroup records the XCD / shader engine / CU it runs on (s_getreg HW_ID, XCC_ID) and spins long enough for the whole
// grid to be resident at once; the host prints how many CUs hold 0, 1, 2, ... workgroups, for the launch shapes of the
// correlation kernels' smaller levels (workgroups, threads, LDS bytes).
//   hipcc -O3 --offload-arch=gfx950 tools/micro/wg_placement.hip -o refign_amd/lib/ab/wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void where_kernel(unsigned* out, long spin) {
  extern __shared__ char lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) lds[0] = (char)hw;
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

int main() {
  struct Cfg { int wgs, threads, lds; const char* what; };
  const Cfg cfgs[] = {
      {272, 192, 36864, "level 2, single 8x32 tiles (3 waves, 36 KB)"},
      {272, 384, 73728, "level 2, channel split (6 waves, 72 KB)"},
      {272, 384, 90112, "level 2, channel split, LDS padded to 88 KB (one workgroup per CU)"},
      {136, 384, 73728, "level 2, two tiles per workgroup (6 waves, 72 KB)"},
      {68, 768, 147456, "level 2, four tiles per workgroup (12 waves, 144 KB)"},
      {128, 192, 36864, "K2 level 1, single tiles"},
      {128, 384, 73728, "K2 level 1, channel split"},
      {255, 768, 147456, "level 1 (12 waves, 144 KB)"},
      {640, 256, 32768, "student GEMM 8160 x 320 -> 320: 640 tiles of 64 x 64 (4 waves, 32 KB)"},
  };
  unsigned* d;
  (void)hipMalloc(&d, 4096 * 2 * sizeof(unsigned));
  (void)hipFuncSetAttribute((const void*)where_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (const Cfg& c : cfgs) {
    (void)hipMemset(d, 0xff, 4096 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(where_kernel, dim3(c.wgs), dim3(c.threads), c.lds, 0, d, 3000L);   // 100 MHz counter: 30 us
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(2 * c.wgs);
    (void)hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_xcd;
    for (int i = 0; i < c.wgs; ++i) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 15;
      const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
      per_xcd[xcc]++;
    }
    std::map<int, int> hist;
    for (auto& kv : per_cu) hist[kv.second]++;
    printf("%-78s %4d workgroups on %3zu CUs:", c.what, c.wgs, per_cu.size());
    for (auto& kv : hist) printf("  %d CUs x %d", kv.second, kv.first);
    printf("   | per XCD:");
    for (auto& kv : per_xcd) printf(" %d", kv.second);
    printf("\n");
  }
  return 0;
}
