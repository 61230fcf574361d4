// grip_set_error / grip_last_error for the HOST-ONLY sanitizer build of leaderboard.cpp + bpe.cpp (`make sanitize`); the shipped library's
// versions live in tower.hip.  Not part of libgrip_amd.so.
#include <stdarg.h>

#include "host_common.h"

static thread_local char g_err[512] = "";
void grip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char* grip_last_error(void) { return g_err; }
