// trace_ranges.hpp — profiler ranges around the driver's phases (query, GPU enqueue, batch, launch set, top-K, collect): the
// role of the reference's nvtx::ScopedRange (src/hpc_helpers/nvtx_markers.cuh:17-57; runpeakbenchmark.sh:37-39 wraps the
// binary in nsys).  `rocprofv3 --marker-trace -- align ...` shows them as a timeline next to the kernels.
//
// The ROCTX library is looked up at run time (dlopen; rocprofiler-sdk's first, roctracer's as a fallback): no link-time
// dependency, nothing to stub in builds without ROCm (tests/host/fake_gpu), and a process that runs without a profiler pays
// one well-predicted branch per range.  CUDASW4_AMD_TRACE_RANGES=0 keeps the library from being loaded at all.
#pragma once
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace swh {

struct TraceApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    TraceApi() {
        const char* e = getenv("CUDASW4_AMD_TRACE_RANGES");
        if (e && e[0] == '0') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void* h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};

inline TraceApi& trace_api() {
    static TraceApi api;
    return api;
}

// a range on the calling thread, closed when the object goes out of scope; the name is formatted only when a tool listens
struct TraceRange {
    bool on;
    template <class... Args>
    explicit TraceRange(const char* fmt, Args... args) : on(trace_api().push != nullptr) {
        if (!on) return;
        char buf[160];
        if constexpr (sizeof...(Args) == 0) snprintf(buf, sizeof(buf), "%s", fmt);
        else snprintf(buf, sizeof(buf), fmt, args...);
        trace_api().push(buf);
    }
    ~TraceRange() { if (on) trace_api().pop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

}  // namespace swh
