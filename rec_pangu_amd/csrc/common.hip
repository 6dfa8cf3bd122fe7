#include "common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>

static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_launches{0};

int rp_fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

void rp_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

// bumped whenever an entry point's signature changes or one is added: rec_pangu_amd/hip.py refuses a library of another
// version (its ctypes table would call through the wrong prototypes silently).  106 = round 6.
extern "C" int rp_version(void) { return 106; }
extern "C" const char *rp_last_error(void) { return g_err; }
extern "C" uint64_t rp_launch_count(void) { return g_launches.load(std::memory_order_relaxed); }
