#include "stt_common.h"

static thread_local char g_last_error[512] = "";

int stt_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* stt_last_error(void) { return g_last_error; }
extern "C" int stt_abi_version(void) { return 5; }
