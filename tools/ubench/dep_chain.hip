// dep_chain.hip — how many waves per SIMD does a DEPENDENT chain of packed ops need to keep the VALU busy?
// Each lane runs `ILP` independent chains of v_pk_maximum3_f16 / v_pk_add_f16 (each instruction reads the result of
// the previous one of its chain); waves per SIMD are set by the grid (1 workgroup of 256 threads = 1 wave per SIMD).
//   hipcc --offload-arch=gfx950 -O3 dep_chain.hip -o dep_chain && ./dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;

template <int ILP>
__global__ void __launch_bounds__(256) k_chain(unsigned* out, unsigned seed) {
    unsigned a[ILP];
    unsigned b = seed + threadIdx.x, c = seed * 3u + 1u;
    for (int i = 0; i < ILP; i++) a[i] = seed + i + threadIdx.x;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int u = 0; u < 24 / ILP; u++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#pragma unroll
            for (int i = 0; i < ILP; i++) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        }
    }
    unsigned r = 0;
    for (int i = 0; i < ILP; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int ILP>
void run(int cus, unsigned* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; wps++) {
        const int grid = cus * wps;
        hipLaunchKernelGGL(k_chain<ILP>, dim3(grid), dim3(256), 0, 0, out, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chain<ILP>, dim3(grid), dim3(256), 0, 0, out, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)grid * 256.0 * ITER * 48.0;
        printf("ILP %d  waves/SIMD %d: %8.1f lanes/clk/CU @2.4GHz  (%.3f ms)\n", ILP, wps, instr / (ms * 1e-3) / (cus * 2.4e9), ms);
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    unsigned* out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * prop.multiProcessorCount * 8));
    run<1>(prop.multiProcessorCount, out);
    run<2>(prop.multiProcessorCount, out);
    run<3>(prop.multiProcessorCount, out);
    run<4>(prop.multiProcessorCount, out);
    return 0;
}
