// Diagnostics entry points of the C-ABI (include/falnet_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/falnet_hip.h"

static thread_local char g_err[512] = "";

void falnet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int falnet_version(void) { return 100; }
extern "C" const char* falnet_last_error(void) { return g_err; }
extern "C" int falnet_channel_pad(int dtype) { (void)dtype; return 32; }
