// refign_amd/csrc/gemm2.hip -- launch side of the second-generation NT GEMM (gemm2.h), in a translation unit of its own:
// gemm2.h pins its accumulators to AGPRs by hand, the other matrix-core files are built with `-amdgpu-mfma-vgpr-form`
// (Makefile), and LLVM's AGPR-copy rewrite pass does not survive the two together.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "gemm2.h"

namespace rfn {

int launch_nt2(const void* X, const void* W, void* Y, long M, long N, long K, long ldx, long ldw, long ldy, const void* bias,
               const void* res, const float* rowscale, int rows_per_sample, int bn2, long t2, hipStream_t s) {
  Gemm2Epi e2{(const uint16_t*)bias, (const uint16_t*)res, rowscale, rows_per_sample, nullptr};
  const bool res2 = res != nullptr || rowscale != nullptr;
  const int tn = (int)(N / bn2);
  // persistent grid: the tiles are dealt round-robin, so the launch takes ceil(t2 / G) rounds whatever G <= 256 is -- the
  // smallest G with the same round count leaves the other CUs to the streams that run next to the teacher (425 tiles: 213
  // workgroups of 2 tiles instead of 256 of which 87 run one)
  const long rounds = cdiv(t2, 256L);
  dim3 grid((unsigned)cdiv(t2, rounds)), block(256);
#define RFN_G2(BN_, NSK_, D3_, BIAS_, RES_)                                                                              \
  hipLaunchKernelGGL((gemm_nt2_kernel<1, 192, BN_, 2, 2, BIAS_, RES_, 0, NSK_, 4, 4, D3_>), grid, block, 0, s,           \
                     (const uint16_t*)X, (const uint16_t*)W, (uint16_t*)Y, (int)M, (int)N, (int)K, ldx, ldw, ldy, tn,    \
                     (int)t2, e2)
#define RFN_G2E(BN_, NSK_, D3_)                                                                                          \
  do {                                                                                                                   \
    if (bias != nullptr) {                                                                                               \
      if (res2) RFN_G2(BN_, NSK_, D3_, true, true);                                                                      \
      else RFN_G2(BN_, NSK_, D3_, true, false);                                                                          \
    } else {                                                                                                             \
      if (res2) RFN_G2(BN_, NSK_, D3_, false, true);                                                                     \
      else RFN_G2(BN_, NSK_, D3_, false, false);                                                                         \
    }                                                                                                                    \
  } while (0)
  if (bn2 == 320) RFN_G2E(320, 3, 8);
  else RFN_G2E(256, 2, 6);
#undef RFN_G2E
#undef RFN_G2
  return check_launch("gemm_nt2");
}

}  // namespace rfn
