// side_launch_probe.hip — when does a small side launch (the giants) start next to a persistent grid that fills the GPU?
// Mimics the driver: work stream: [small kernel] -> event `fork` -> [hog: a persistent grid that takes every workgroup slot];
// side stream (high priority): wait(fork) -> [side: 5 workgroups, 43 KB of LDS each].  The host enqueues the side launch first.
// Variants: plain (race), and with a device-side handshake: the side kernel counts its started workgroups in signal memory and
// the work stream waits for the count (hipStreamWaitValue32) before the hog.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/side_launch_probe.hip -o tools/ubench/side_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_BYTES>
__global__ void __launch_bounds__(256) spin(unsigned long long ticks, unsigned long long* stamps, int slot, unsigned* started) {
    __shared__ unsigned char lds[LDS_BYTES];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(&stamps[2 * slot], t0);
        if (started) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) atomicMax(&stamps[2 * slot + 1], wall_clock64());
    if (lds[(threadIdx.x + 1) & 255] == 77 && ticks == 1) stamps[15] = 1;  // keep the LDS array alive
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    unsigned long long* d; CK(hipMalloc(&d, 128));
    unsigned long long h[16];
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int canWait = 0; CK(hipDeviceGetAttribute(&canWait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("priority range [%d, %d], hipDeviceAttributeCanUseStreamWaitValue = %d\n", lo, hi, canWait);
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    for (int variant = 0; variant < 3; variant++) {
        // 0: side stream created AFTER the work stream; 1: BEFORE; 2: after, with the handshake
        hipStream_t work, side, extra[3];
        if (variant == 1) CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
        CK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
        for (auto& x : extra) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        if (variant != 1) CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
        hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        const bool handshake = variant == 2;
        int beside = 0, behind = 0, infront = 0;
        for (int r = 0; r < reps; r++) {
            unsigned long long init[16]; for (int i = 0; i < 16; i++) init[i] = (i & 1) ? 0ull : ~0ull;
            CK(hipMemcpy(d, init, 128, hipMemcpyHostToDevice));
            *sig = 0;
            const unsigned long long ms = 100000ull;  // 100 MHz ticks per ms
            hipLaunchKernelGGL(spin<1024>, dim3(64), dim3(256), 0, work, ms / 5, d, 0, (unsigned*)nullptr);   // "profile build"
            CK(hipEventRecord(fork, work));
            CK(hipStreamWaitEvent(side, fork, 0));
            hipLaunchKernelGGL(spin<44000>, dim3(5), dim3(256), 0, side, 20 * ms, d, 1, handshake ? sig : (unsigned*)nullptr);   // giants
            if (handshake) CK(hipStreamWaitValue32(work, sig, 5, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(spin<63000>, dim3(1024), dim3(256), 0, work, 15 * ms, d, 2, (unsigned*)nullptr);                 // bulk: 2 per CU resident, 2 rounds
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 128, hipMemcpyDeviceToHost));
            const double t0 = (double)h[0];
            const double s0 = (h[2] - t0) / 1e5, s1 = (h[3] - t0) / 1e5, b0 = (h[4] - t0) / 1e5, b1 = (h[5] - t0) / 1e5;
            const char* what = s0 < b0 + 1.0 && s1 > b0 ? (s0 <= b0 ? "side first, overlapping" : "beside") : s0 >= b1 - 1.0 ? "BEHIND the bulk" : s1 <= b0 ? "IN FRONT (serial)" : "late";
            if (s0 >= b1 - 1.0) behind++; else if (s1 <= b0 + 0.01) infront++; else if (s0 < b0 + 1.0) beside++;
            if (r < 3) printf("  variant %d rep %d: side [%.2f, %.2f] bulk [%.2f, %.2f] ms -> %s\n", variant, r, s0, s1, b0, b1, what);
        }
        printf("variant %d (%s): beside %d, behind %d, in front %d of %d\n", variant,
               variant == 0 ? "side stream created last" : variant == 1 ? "side stream created first" : "created last + start handshake", beside, behind, infront, reps);
        CK(hipStreamDestroy(work)); CK(hipStreamDestroy(side)); for (auto& x : extra) CK(hipStreamDestroy(x));
    }
    return 0;
}
