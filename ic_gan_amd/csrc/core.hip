// Library-wide helpers of libicgan_hip.so (error reporting, version).
#include "icg_common.h"

// per calling thread: the hipError_t behind the last ICG_ERR_LAUNCH this thread received (no cross-thread races; a caller reads
// it right after the failing call, on the same thread)
thread_local int g_icg_last_hip_error = 0;

extern "C" const char* icg_strerror(int code) {
  switch (code) {
    case ICG_OK: return "ok";
    case ICG_ERR_ARG: return "invalid argument (null pointer, unsupported shape or misaligned buffer)";
    case ICG_ERR_LAUNCH: return "HIP kernel launch failed (see icg_last_hip_error)";
    case ICG_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown icg error";
  }
}
extern "C" int icg_last_hip_error(void) { return g_icg_last_hip_error; }
extern "C" int icg_version(void) { return 100; }
