// refign_amd/csrc/capi.hip -- error plumbing + version of the C ABI (include/refign_hip.h).
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

}  // namespace rfn

extern "C" {
int rfn_abi_version(void) { return RFN_ABI_VERSION; }
const char* rfn_last_error(void) { return rfn::err_buf(); }
}
