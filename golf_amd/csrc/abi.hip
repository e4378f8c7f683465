// ABI bookkeeping for libgolf_hip.so (error text, version).
#include "common.h"

namespace golf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace golf

extern "C" int golf_abi_version(void) { return GOLF_ABI_VERSION; }
extern "C" const char* golf_last_error(void) { return golf::err_buf(); }
extern "C" const char* golf_target_arch(void) { return "gfx950"; }
