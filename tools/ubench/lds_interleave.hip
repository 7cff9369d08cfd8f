// lds_interleave.hip — design experiment for the packed DP kernels: where should the (scoreA, scoreB)
// interleave of the two subjects of a group happen?
//
//   scheme PERM : 2 x ds_read_b128 per 8 query rows + one v_perm_b32 per cell pair      (shipping kernel)
//   scheme D16  : ds_read_u16_d16 + ds_read_u16_d16_hi per cell pair, no VALU interleave
//
// Both run the same 8 packed VALU ops per cell pair (a stand-in for the recurrence) on R=32 rows per
// lane with per-lane "letters" that change every step.  Reports G cell-pairs/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned u32;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr int R = 32;
constexpr int STEPS = 2048;

__device__ __forceinline__ u32 pk_add(u32 a, u32 b) { return __builtin_bit_cast(u32, (f16x2)(__builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b))); }
__device__ __forceinline__ u32 pk_max3(u32 a, u32 b, u32 c) {
    return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)), __builtin_bit_cast(f16x2, c)));
}

// the recurrence stand-in: 4 adds + 3.5 max3 per cell pair, same dependency shape as the DP step
template <class GETS>
__device__ __forceinline__ void cells(u32 (&H)[R], u32 (&E)[R], u32& F, u32& maxv, u32 diag, u32 gop, u32 gex, GETS&& gets) {
#pragma unroll
    for (int r = 0; r < R; r++) {
        const u32 s = gets(r);
        const u32 t = pk_add(diag, s);
        diag = H[r];
        const u32 h = pk_max3(t, E[r], F);
        const u32 hg = pk_add(h, gop);
        E[r] = pk_max3(pk_add(E[r], gex), hg, 0u);
        F = pk_max3(pk_add(F, gex), hg, 0u);
        H[r] = h;
        if (r & 1) maxv = pk_max3(maxv, H[r - 1], h);
    }
}

// PERM: tile[c][chunk k][lane l][16 B], 16-bit entries, row stride 1024 B
__global__ void __launch_bounds__(256) k_perm(u32* out, const unsigned char* letters, u32 gop, u32 gex) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[16 + 21 * 1024];
    for (int i = threadIdx.x; i < (21 * 1024) / 4; i += 256) ((u32*)(tile + 16))[i] = i * 2654435761u & 0x03ff03ffu;
    __syncthreads();
    const int lane16 = threadIdx.x & 15;
    u32 H[R], E[R];
    for (int r = 0; r < R; r++) { H[r] = 0; E[r] = 0; }
    u32 F = 0, maxv = 0;
    const unsigned char* lp = letters + (blockIdx.x * 256 + threadIdx.x) * 2;
    for (int step = 0; step < STEPS; step++) {
        const u32 ca = lp[(step * 2) & 1023] % 21, cb = lp[(step * 2 + 1) & 1023] % 21;
        const unsigned char* pa = tile + 16 + ca * 1024 + lane16 * 16;
        const unsigned char* pb = tile + 16 + cb * 1024 + lane16 * 16;
        u32 wa[16], wb[16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 a = *(const uint4*)(pa + k * 256), b = *(const uint4*)(pb + k * 256);
            wa[4 * k] = a.x; wa[4 * k + 1] = a.y; wa[4 * k + 2] = a.z; wa[4 * k + 3] = a.w;
            wb[4 * k] = b.x; wb[4 * k + 1] = b.y; wb[4 * k + 2] = b.z; wb[4 * k + 3] = b.w;
        }
        cells(H, E, F, maxv, F, gop, gex, [&](int r) { return __builtin_amdgcn_perm(wb[r >> 1], wa[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u); });
    }
    out[blockIdx.x * 256 + threadIdx.x] = maxv ^ H[3] ^ E[5];
}

// D16: tile[c][row r][parity p][lane l] 16-bit entries: address = c*2048 + r*64 + p*32 + l*2
// (two DPP rows of a 32-lane half never share a bank: parity picks the upper/lower 8 banks of the line).
// The d16 loads are inline asm (hipcc only emits zero-extending ds_read_u16 + a VALU merge), software
// pipelined by hand: the loads of step t+1 are in flight while step t computes.
#define D16_LOAD(REG, PA, PB, OFF)                                                                \
    asm volatile("ds_read_u16_d16 %0, %1 offset:%3\n\tds_read_u16_d16_hi %0, %2 offset:%3"          \
                 : "+v"(REG) : "v"(PA), "v"(PB), "i"(OFF))

template <int R0>
__device__ __forceinline__ void d16_issue(u32 (&s)[R], u32 pa, u32 pb) {
    if constexpr (R0 < R) {
        D16_LOAD(s[R0], pa, pb, R0 * 64);
        d16_issue<R0 + 1>(s, pa, pb);
    }
}

__device__ __forceinline__ void d16_wait(u32 (&s)[R]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]),
                   "+v"(s[8]), "+v"(s[9]), "+v"(s[10]), "+v"(s[11]), "+v"(s[12]), "+v"(s[13]), "+v"(s[14]), "+v"(s[15]));
    asm volatile("; d16 wait (second half)"
                 : "+v"(s[16]), "+v"(s[17]), "+v"(s[18]), "+v"(s[19]), "+v"(s[20]), "+v"(s[21]), "+v"(s[22]), "+v"(s[23]),
                   "+v"(s[24]), "+v"(s[25]), "+v"(s[26]), "+v"(s[27]), "+v"(s[28]), "+v"(s[29]), "+v"(s[30]), "+v"(s[31]));
    __builtin_amdgcn_sched_barrier(0);
}

__global__ void __launch_bounds__(256) k_d16(u32* out, const unsigned char* letters, u32 gop, u32 gex) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[21 * 1024];
    for (int i = threadIdx.x; i < 21 * 1024; i += 256) tile[i] = (unsigned short)((i * 2654435761u) & 0x03ff);
    __syncthreads();
    const int lane16 = threadIdx.x & 15;
    const int parity = (threadIdx.x >> 4) & 1;
    u32 H[R], E[R];
    for (int r = 0; r < R; r++) { H[r] = 0; E[r] = 0; }
    u32 F = 0, maxv = 0;
    const unsigned char* lp = letters + (blockIdx.x * 256 + threadIdx.x) * 2;
    const u32 base = (u32)(size_t)tile + parity * 32 + lane16 * 2;  // LDS byte address
    auto addr = [&](int step, int which) { return base + (u32)(lp[(step * 2 + which) & 1023] % 21) * 2048u; };
    u32 sA[R], sB[R];
    for (int r = 0; r < R; r++) { sA[r] = 0; sB[r] = 0; }
    d16_issue<0>(sA, addr(0, 0), addr(0, 1));
    for (int step = 0; step < STEPS; step += 2) {
        d16_wait(sA);
        d16_issue<0>(sB, addr(step + 1, 0), addr(step + 1, 1));
        cells(H, E, F, maxv, F, gop, gex, [&](int r) { return sA[r]; });
        d16_wait(sB);
        d16_issue<0>(sA, addr(step + 2, 0), addr(step + 2, 1));
        cells(H, E, F, maxv, F, gop, gex, [&](int r) { return sB[r]; });
    }
    d16_wait(sA);
    out[blockIdx.x * 256 + threadIdx.x] = maxv ^ H[3] ^ E[5] ^ sA[0];
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    u32* out;
    unsigned char* letters;
    CHECK(hipMalloc(&out, sizeof(u32) * 256 * cus * 8));
    std::vector<unsigned char> h(256 * cus * 8 * 2 + 2048);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(rand() % 21);
    CHECK(hipMalloc(&letters, h.size()));
    CHECK(hipMemcpy(letters, h.data(), h.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int wgs = 2; wgs <= 4; wgs++) {
        for (int which = 0; which < 2; which++) {
            const int grid = cus * wgs;
            auto launch = [&]() {
                if (which == 0) hipLaunchKernelGGL(k_perm, dim3(grid), dim3(256), 0, 0, out, letters, 0xcb80cb80u, 0xbc00bc00u);
                else hipLaunchKernelGGL(k_d16, dim3(grid), dim3(256), 0, 0, out, letters, 0xcb80cb80u, 0xbc00bc00u);
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double pairs = (double)grid * 256 * STEPS * R;
            printf("%-5s WGs/CU=%d  %.3f ms  %.1f G cell-pairs/s  (= %.0f GCUPS packed)\n", which ? "D16" : "PERM", wgs, ms,
                   pairs / ms / 1e6, 2 * pairs / ms / 1e6);
        }
    }
    return 0;
}
