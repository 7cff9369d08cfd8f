// fake_hip.cpp — TEST INFRASTRUCTURE (never shipped, never measured): a host-memory stand-in for the ~35 HIP runtime
// entry points the C++ host driver (cudasw4_amd/csrc/host/search_driver.cpp) calls, with TWO devices, so that the
// driver's multi-GPU path — per-device hipSetDevice in worker threads, per-device streams / events / buffers, shard-to-
// device mapping, host merge — runs on a CPU-only box with two DISTINCT device ordinals (every GPU
// test so far used devices=[0, 0, ...]).  Everything executes synchronously at enqueue time; what the fake checks is
// AFFINITY: every stream, event and device allocation belongs to the device that was current when it was created, and
// every call that uses one must be made while that device is current on the calling thread.  Violations are counted and
// described (fake_hip_violations / fake_hip_violation_text).
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

namespace {
constexpr int kDevices = 2;
thread_local int t_device = 0;
std::mutex g_mu;
std::map<const void*, std::pair<size_t, int>> g_allocs;  // device allocations: base -> (bytes, device)
std::atomic<int> g_violations{0};
std::string g_text;
std::atomic<long> g_calls[kDevices];

struct FakeStream { int device; };
struct FakeEvent { int device; double t; };

void violation(const char* what, int owner) {
    g_violations++;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_text.size() < 4000) g_text += std::string(what) + ": object of device " + std::to_string(owner) + " used while device " + std::to_string(t_device) + " is current; ";
}
int owner_of(const void* p) {  // device of the allocation that holds p, -1: host memory
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_allocs.upper_bound(p);
    if (it == g_allocs.begin()) return -1;
    --it;
    const char* b = static_cast<const char*>(it->first);
    return (static_cast<const char*>(p) >= b && static_cast<const char*>(p) < b + it->second.first) ? it->second.second : -1;
}
void check_ptr(const char* what, const void* p) {
    const int o = owner_of(p);
    if (o >= 0 && o != t_device) violation(what, o);
}
void check_stream(const char* what, hipStream_t s) {
    if (!s) return;
    const int o = reinterpret_cast<FakeStream*>(s)->device;
    if (o != t_device) violation(what, o);
    g_calls[o < kDevices ? o : 0]++;
}
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" {
int fake_hip_violations() { return g_violations.load(); }
const char* fake_hip_violation_text() { return g_text.c_str(); }
long fake_hip_stream_calls(int device) { return g_calls[device].load(); }
int fake_hip_owner_of(const void* p) { return owner_of(p); }
int fake_hip_current_device() { return t_device; }

hipError_t hipGetDeviceCount(int* n) { *n = kDevices; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= kDevices) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake hip error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* pi, hipDeviceAttribute_t attr, int) {
    *pi = attr == hipDeviceAttributeCanUseStreamWaitValue ? 1 : attr == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
    return hipSuccess;
}
hipError_t hipDeviceGetPCIBusId(char* s, int len, int device) { snprintf(s, size_t(len), "0000:%02x:00.0", 0xc1 + device * 0x10); return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 1; *hi = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = size_t(64) << 30; *t = size_t(64) << 30; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
    *p = calloc(1, n ? n : 1);
    std::lock_guard<std::mutex> lk(g_mu);
    g_allocs[*p] = {n ? n : 1, t_device};
    return hipSuccess;
}
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> lk(g_mu); g_allocs.erase(p); }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return hipSuccess; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(new FakeStream{t_device}); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<FakeStream*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { check_stream("hipStreamSynchronize", s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned) { check_stream("hipStreamWaitEvent", s); return hipSuccess; }
hipError_t hipStreamWaitValue32(hipStream_t s, void* ptr, uint32_t value, unsigned, uint32_t) {
    check_stream("hipStreamWaitValue32", s);
    // everything ran synchronously: the value must be there already, or the real thing would hang
    if (*static_cast<uint32_t*>(ptr) < value) { violation("hipStreamWaitValue32 would never be released", t_device); }
    return hipSuccess;
}

hipError_t hipStreamWriteValue32(hipStream_t s, void* ptr, uint32_t value, unsigned) {
    check_stream("hipStreamWriteValue32", s);
    *static_cast<uint32_t*>(ptr) = value;
    return hipSuccess;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new FakeEvent{t_device, 0.0}); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<FakeEvent*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    check_stream("hipEventRecord", s);
    FakeEvent* fe = reinterpret_cast<FakeEvent*>(e);
    if (fe->device != t_device) violation("hipEventRecord (event)", fe->device);
    fe->t = now_ms();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // everything ran at enqueue time
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = float(reinterpret_cast<FakeEvent*>(b)->t - reinterpret_cast<FakeEvent*>(a)->t); return hipSuccess; }

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t s) {
    check_stream("hipMemcpyAsync", s);
    check_ptr("hipMemcpyAsync dst", dst);
    check_ptr("hipMemcpyAsync src", src);
    memcpy(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind k) { return hipMemcpyAsync(dst, src, n, k, nullptr); }
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t s) {
    check_stream("hipMemsetAsync", s);
    check_ptr("hipMemsetAsync", dst);
    memset(dst, v, n);
    return hipSuccess;
}
}
