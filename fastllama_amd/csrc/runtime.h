// runtime.h -- error/diagnostic helpers shared by the translation units of libfastllama_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace fl {
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char *what);
void warn(const char *fmt, ...) __attribute__((format(printf, 1, 2)));   // one line to the warning handler (fl_set_warn_handler; default stderr)
int ensure_device();  // FL_OK, or FL_ENODEV when no HIP device is visible (there is no CPU fallback)
}  // namespace fl
