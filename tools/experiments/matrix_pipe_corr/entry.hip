// tools/experiments/matrix_pipe_corr/entry.hip -- C entry point of the fp32-matrix-pipe forward (was RFN_CORR_VARIANT=30 of
// the product's rfn_corr_fwd_f32 until round 3)
#include "common.h"
namespace rfn {
int launch_corr9_mfma(const float* in1, const float* in2, float* out, int B, int C, int H, int W, bool fuse, hipStream_t st);
}
extern "C" int rfx_corr9_mfma(const float* in1, const float* in2, float* out, int B, int C, int H, int W, int fuse, void* stream) {
  return rfn::launch_corr9_mfma(in1, in2, out, B, C, H, W, fuse != 0, (hipStream_t)stream);
}
