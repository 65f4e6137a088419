// Error plumbing and version entry points of the C-ABI (include/yolov3_amd.h).
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void yolo_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int yolo_abi_version(void) { return YOLO_ABI_VERSION; }
extern "C" const char* yolo_last_error(void) { return g_err; }
extern "C" int yolo_abi_dtype(void) {
#ifdef YOLO_FP16
  return YOLO_DTYPE_FP16;
#else
  return YOLO_DTYPE_BF16;
#endif
}
