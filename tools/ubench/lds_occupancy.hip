// lds_occupancy.hip — how many 256-thread workgroups with N bytes of static LDS does a gfx950 CU hold?  (the LDS allocation
// granule decides whether three 53.8 KB profile tiles fit the 160 KB of a CU; sw_dp_kernel.hpp: SWK_WAVES3_MAX_R)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/lds_occupancy.hip -o tools/ubench/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BYTES>
__global__ void __launch_bounds__(256, 3) k(unsigned* out) {
    __shared__ unsigned char lds[BYTES];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[out[0] & 255];
}
// what the hardware really holds: every workgroup counts itself in, notes the largest count it sees, spins 20 ms, counts out
template <int BYTES>
__global__ void __launch_bounds__(256, 3) live(unsigned* ctr) {
    __shared__ unsigned char lds[BYTES];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned now = atomicAdd(&ctr[0], 1u) + 1u + (lds[ctr[2] & 255] & 0u);
        atomicMax(&ctr[1], now);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 2000000ull) __builtin_amdgcn_s_sleep(32);
        atomicSub(&ctr[0], 1u);
    }
}
template <int BYTES>
void probe() {
    int n = -1, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 256, 0);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned* d; hipMalloc(&d, 16); hipMemset(d, 0, 16);
    hipLaunchKernelGGL(live<BYTES>, dim3(cus * 4), dim3(256), 0, 0, d);
    unsigned h[4] = {0, 0, 0, 0};
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipFree(d);
    printf("static LDS %6d B: occupancy API %d workgroups per CU (%s); live at once on %d CUs: %u = %.2f per CU\n", BYTES, n,
           hipGetErrorString(e), cus, h[1], double(h[1]) / cus);
}
int main() {
    probe<48408>(); probe<53248>(); probe<53760>(); probe<53761>(); probe<53784>(); probe<54272>(); probe<54528>();
    probe<54612>(); probe<54613>(); probe<55040>(); probe<59160>();
    return 0;
}
