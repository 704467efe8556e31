// Error plumbing and library identity for libsglk.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace sglk {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(SGLK_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return SGLK_OK;
}

// Compute units of the current device (cached per device id); 256 on MI355X.
int num_cus() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

int set_max_dyn_lds(const void* fn, int bytes, unsigned long long* done_mask, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 64;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && ((__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return SGLK_OK;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return fail(SGLK_ELAUNCH, "%s: cannot reserve %d B of LDS: %s", what, bytes, hipGetErrorString(e));
  if (tracked) __atomic_fetch_or(done_mask, 1ull << dev, __ATOMIC_RELEASE);
  return SGLK_OK;
}

}  // namespace sglk

extern "C" {
const char* sglk_last_error(void) { return sglk::g_err; }
const char* sglk_version(void) { return "0.1.0"; }
const char* sglk_arch(void) { return "gfx950"; }
int sglk_abi_version(void) { return SGLK_ABI_VERSION; }
}
