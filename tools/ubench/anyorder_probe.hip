// anyorder_probe.hip — does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on gfx950?
// And: does a stream created with a CU mask confine a kernel to those CUs (hipExtStreamCreateWithCUMask)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/anyorder_probe.hip -o tools/ubench/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(unsigned long long cycles, unsigned long long* stamps, int slot) {
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}
__global__ void where(unsigned* cu_seen) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { atomicAdd(&cu_seen[(xcc & 0xf) * 64 + ((id >> 8) & 0xf) + 16 * ((id >> 13) & 0x7)], 1u); }
}

int main() {
    unsigned long long* d; CK(hipMalloc(&d, 64));
    unsigned long long h[8];
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const unsigned long long ms100 = 10000000ull;  // wall_clock64 ticks at 100 MHz: 100 ms
    for (int mode = 0; mode < 3; mode++) {
        CK(hipMemset(d, 0, 64));
        unsigned long long c = ms100 / 5; int s0 = 0, s1 = 1;
        void* a0[] = {&c, &d, &s0}; void* a1[] = {&c, &d, &s1};
        // A: few workgroups, long.  B: right behind it in the same stream.
        CK(hipExtLaunchKernel((const void*)spin, dim3(4), dim3(64), a0, 0, s, nullptr, nullptr, mode == 2 ? hipExtAnyOrderLaunch : 0));
        CK(hipExtLaunchKernel((const void*)spin, dim3(4), dim3(64), a1, 0, s, nullptr, nullptr, mode >= 1 ? hipExtAnyOrderLaunch : 0));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
        printf("mode %d (%s): A [%.2f, %.2f] ms  B [%.2f, %.2f] ms  -> %s\n", mode,
               mode == 0 ? "plain, plain" : mode == 1 ? "plain, anyorder" : "anyorder, anyorder",
               0.0, (h[1] - h[0]) / 1e5, (double)(h[2] - h[0]) / 1e5, (double)(h[3] - h[0]) / 1e5,
               h[2] < h[1] ? "OVERLAP" : "serial");
    }
    // CU mask: 8 CUs (one per 32)
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0u);
    for (size_t i = 0; i < mask.size(); i++) mask[i] = 0x1u;
    hipStream_t sm;
    hipError_t e = hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask.size(), mask.data());
    printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        unsigned* seen; CK(hipMalloc(&seen, 8 * 64 * 4)); CK(hipMemset(seen, 0, 8 * 64 * 4));
        hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, sm, seen);
        CK(hipStreamSynchronize(sm));
        std::vector<unsigned> hs(8 * 64); CK(hipMemcpy(hs.data(), seen, 8 * 64 * 4, hipMemcpyDeviceToHost));
        int used = 0; for (unsigned v : hs) used += v != 0;
        printf("masked stream: %d distinct (xcc, cu) slots saw workgroups (mask has %zu bits set)\n", used, mask.size());
        CK(hipMemset(seen, 0, 8 * 64 * 4));
        hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, seen);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(hs.data(), seen, 8 * 64 * 4, hipMemcpyDeviceToHost));
        used = 0; for (unsigned v : hs) used += v != 0;
        printf("plain stream : %d distinct (xcc, cu) slots\n", used);
    }
    return 0;
}
