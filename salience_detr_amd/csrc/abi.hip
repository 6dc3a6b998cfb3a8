// ABI bookkeeping of libsalience_hip.so: version + thread-local error text.
#include "common.h"

namespace sdetr {
char *error_buffer()
{
    static thread_local char buf[512] = {0};
    return buf;
}

static thread_local int g_last_forward_kernel = 0;
void note_forward_kernel(int which) { g_last_forward_kernel = which; }
static thread_local int g_last_backward_kernel = 0;
void note_backward_kernel(int which) { g_last_backward_kernel = which; }
}  // namespace sdetr

extern "C" int sdetr_abi_version(void) { return SDETR_ABI_VERSION; }
extern "C" const char *sdetr_last_error(void) { return sdetr::error_buffer(); }
extern "C" int sdetr_msda_last_kernel(void) { return sdetr::g_last_forward_kernel; }
extern "C" int sdetr_msda_last_backward_kernel(void) { return sdetr::g_last_backward_kernel; }
