// capi.hip -- library-level entry points of libmau_hip.so: version, error reporting, device check.
#include <stdarg.h>
#include <string.h>
#include "mau_common.h"

namespace mau {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return MAU_ERR_HIP;
  }
  return MAU_OK;
}

}  // namespace mau

extern "C" {

int mau_abi_version(void) { return MAU_ABI_VERSION; }

const char* mau_last_error(void) { return mau::g_err; }

int mau_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    mau::set_error("no HIP device");
    return MAU_ERR_DEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mau::set_error("hipGetDeviceProperties failed");
    return MAU_ERR_DEVICE;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mau::set_error("device %d is %s, libmau_hip is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
    return MAU_ERR_DEVICE;
  }
  return MAU_OK;
}

}  // extern "C"
