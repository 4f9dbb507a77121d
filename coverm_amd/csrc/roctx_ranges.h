// roctx ranges around the host-side phases of the product (window upload, inflate round, parse / extraction, cov_finish), so that a
// `rocprofv3 --marker-trace --kernel-trace` timeline shows which phase a kernel or a copy belongs to.  The marker library
// (librocprofiler-sdk-roctx, or the older libroctx64) is bound at run time and only when a profiler is present (rocprofv3 sets
// ROCP_TOOL_LIBRARIES) or COVERM_ROCTX=1 asks for it: an ordinary run never loads it and a range costs one predictable branch.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>

namespace covr {

struct Roctx {
    typedef int (*push_t)(const char *);
    typedef int (*pop_t)();
    push_t push = nullptr; pop_t pop = nullptr;
    Roctx() {
        const char *on = getenv("COVERM_ROCTX");
        if (on ? atoi(on) == 0 : getenv("ROCP_TOOL_LIBRARIES") == nullptr) return;
        void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (push_t)dlsym(h, "roctxRangePushA"); pop = (pop_t)dlsym(h, "roctxRangePop");
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
inline Roctx &roctx() { static Roctx r; return r; }

struct Range {     // RAII: the range ends where the scope does, on every way out
    bool on;
    explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) (void)roctx().push(name); }
    ~Range() { if (on) (void)roctx().pop(); }
    Range(const Range &) = delete; Range &operator=(const Range &) = delete;
};

}  // namespace covr
