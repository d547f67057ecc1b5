// refign_amd/csrc/capi.hip -- error plumbing + version of the C ABI (include/refign_hip.h).
#include <cstdint>

#include "common.h"

namespace rfn {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Zero-fill as a KERNEL on the caller's stream.  hipMemsetAsync becomes a memset NODE when the stream is being captured
// into a hipGraph.  With an experimental loss kernel of round 2 (fused up-sampling + cross-entropy; not kept) whose entry
// point zeroed a 467 KB and an 8 KB buffer that way, the captured student pass was right on its first replay and
// produced garbage gradients from the second replay on, in that buffer and in neighbouring pool tensors; with the same
// buffers zeroed by a kernel the gradients were right on every replay.  A kernel node is ordered like every other
// kernel of the pass, so the entry points that run inside captured passes (attention dK/dV accumulator, BatchNorm
// statistics) zero with this.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t n_words) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n_words; i += stride) {
    if (i + 4 <= n_words) {
      *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t j = i; j < n_words; ++j) p[j] = 0u;
    }
  }
}

int zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return RFN_OK;
  if ((reinterpret_cast<uintptr_t>(p) & 15) || (bytes & 3)) {
    // not 16-byte aligned / not whole words: no caller does this; keep the semantics anyway
    return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? RFN_OK : fail(RFN_ELAUNCH, "zero_async: hipMemsetAsync failed");
  }
  const size_t words = bytes / 4;
  const int blocks = (int)((words / 4 + 255) / 256 > 2048 ? 2048 : (words / 4 + 255) / 256);
  hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, (uint32_t*)p, words);
  return check_launch("zero_fill_kernel");
}

}  // namespace rfn

extern "C" {
int rfn_abi_version(void) { return RFN_ABI_VERSION; }
const char* rfn_last_error(void) { return rfn::err_buf(); }
}
