// Diagnostics entry points of the C-ABI (include/falnet_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime_api.h>
#include "../../include/falnet_hip.h"

static thread_local char g_err[512] = "";

void falnet_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// major * 100 + layout revision of the descriptor structs of include/falnet_hip.h (falnet_conv_t, falnet_wgrad_t, falnet_reduce_t ...):
// bumped whenever a struct gains / moves a field or an entry point changes meaning; the ctypes binding and the autotune cache
// header check it.  300: round 3 (falnet_conv_t.variant 17/18, falnet_wgrad_t deterministic flag, falnet_adam_step_dev finite guard).
extern "C" int falnet_version(void) { return 600; }
extern "C" const char* falnet_last_error(void) { return g_err; }

// Deterministic mode (process-wide): see include/falnet_hip.h.  Read by the launchers of conv.hip / losses.hip.
static int g_deterministic = 0;
int falnet_deterministic() { return g_deterministic; }
int* falnet_replay_depth() {
    static thread_local int depth = 0;
    return &depth;
}
extern "C" int falnet_set_deterministic(int on) {
    g_deterministic = on ? 1 : 0;
    return 0;
}
extern "C" int falnet_get_deterministic(void) { return g_deterministic; }
extern "C" int falnet_channel_pad(int dtype) { (void)dtype; return 32; }

// Launch functions make the device of the caller's stream current themselves (common.h: falnet_enter_stream); this sets it
// explicitly for callers that pass the NULL stream from a fresh thread.
extern "C" int falnet_set_device(int device) {
    const hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the failure is reported HERE: do not leave it for the next launch's hipGetLastError
        falnet_set_error("falnet_set_device(%d): %s", device, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
