// mix_rate.hip — does the "fast" class of VALU ops (v_add_u32 / v_add_f32, ~100 lanes/clk/CU) keep its rate
// when interleaved with the "slow" class (v_max3_*, ~59 lanes/clk/CU), and does int differ from float?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 16384;

#define MIX_KERNEL(NAME, ADD, MAX3)                                                                   \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {                       \
        unsigned a[8], b = seed + threadIdx.x, c = seed * 3u + 1u;                                    \
        for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;                                    \
        for (int it = 0; it < ITER; it++) {                                                           \
            _Pragma("unroll") for (int i = 0; i < 8; i++) {                                           \
                asm volatile(ADD " %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
                asm volatile(MAX3 " %0, %0, %1, %2" : "+v"(a[(i + 3) & 7]) : "v"(b), "v"(c));         \
                asm volatile(ADD " %0, %0, %1" : "+v"(a[(i + 5) & 7]) : "v"(c));                      \
                asm volatile(MAX3 " %0, %0, %1, 0" : "+v"(a[(i + 6) & 7]) : "v"(b));                  \
            }                                                                                         \
        }                                                                                             \
        unsigned r = 0;                                                                               \
        for (int i = 0; i < 8; i++) r ^= a[i];                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                               \
    }
MIX_KERNEL(k_int, "v_add_u32", "v_max3_i32")
MIX_KERNEL(k_flt, "v_add_f32", "v_max3_f32")
MIX_KERNEL(k_int_fadd, "v_add_f32", "v_max3_i32")
MIX_KERNEL(k_flt_iadd, "v_add_u32", "v_max3_f32")
MIX_KERNEL(k_pk16, "v_pk_add_u16", "v_pk_maximum3_f16")
MIX_KERNEL(k_f16add_pkmax, "v_add_f16", "v_pk_maximum3_f16")

// ratio mixes closer to the DP loops: per unit NA fast ops + NS slow ops, all independent accumulators
#define RATIO_KERNEL(NAME, BODY)                                                                      \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {                       \
        unsigned a[16], b = seed + threadIdx.x, c = seed * 3u + 1u;                                   \
        unsigned long long p[4];                                                                      \
        for (int i = 0; i < 16; i++) a[i] = seed + i + threadIdx.x;                                   \
        for (int i = 0; i < 4; i++) p[i] = seed * 7ull + i + threadIdx.x;                              \
        const unsigned long long pb = ((unsigned long long)b << 32) | c;                              \
        for (int it = 0; it < ITER; it++) { BODY }                                                    \
        unsigned r = 0;                                                                               \
        for (int i = 0; i < 16; i++) r ^= a[i];                                                       \
        for (int i = 0; i < 4; i++) r ^= (unsigned)p[i] ^ (unsigned)(p[i] >> 32);                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                               \
    }
#define FADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKMAX3H(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKADDH(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(0x05040100));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define ANDB(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define FMUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
// f32 cell x2: 8 adds + 7 max3
RATIO_KERNEL(k_f32_8a7m, FADD(0) FMAX3(8) FADD(1) FMAX3(9) FADD(2) FMAX3(10) FADD(3) FMAX3(11) FADD(4) FMAX3(12) FADD(5) FMAX3(13) FADD(6) FMAX3(14) FADD(7))
// f32 cell x2 with packed adds where legal: 2 pk + 4 scalar adds + 7 max3
RATIO_KERNEL(k_f32_2pk4a7m, PKFADD(0) FMAX3(8) FADD(1) FMAX3(9) FADD(2) FMAX3(10) PKFADD(1) FMAX3(11) FADD(4) FMAX3(12) FADD(5) FMAX3(13) FMAX3(14))
// all adds packed (two subjects per lane): 4 pk adds + 7 max3
RATIO_KERNEL(k_f32_4pk7m, PKFADD(0) FMAX3(8) FMAX3(9) PKFADD(1) FMAX3(10) FMAX3(11) PKFADD(2) FMAX3(12) FMAX3(13) PKFADD(3) FMAX3(14))
// pk_add_f32 alone
RATIO_KERNEL(k_pkfadd, PKFADD(0) PKFADD(1) PKFADD(2) PKFADD(3) PKFADD(0) PKFADD(1) PKFADD(2) PKFADD(3))
// f16x2 cell pair as shipped: 4 pk add + 3.5 pk max3 + 1 perm (x2)
RATIO_KERNEL(k_h2_shipped, PKADDH(0) PKMAX3H(8) PKADDH(1) PKADDH(2) PKMAX3H(9) PKADDH(3) PKMAX3H(10) PERM(4) PKADDH(5) PKMAX3H(11) PKADDH(6) PKADDH(7) PKMAX3H(12) PKADDH(0) PKMAX3H(13) PERM(1) PKMAX3H(14))
// can mov / and / fmul ride along with the packed ops?
RATIO_KERNEL(k_pkmax_mov, PKMAX3H(8) MOV(0) PKMAX3H(9) MOV(1) PKMAX3H(10) MOV(2) PKMAX3H(11) MOV(3))
RATIO_KERNEL(k_pkmax_and, PKMAX3H(8) ANDB(0) PKMAX3H(9) ANDB(1) PKMAX3H(10) ANDB(2) PKMAX3H(11) ANDB(3))
RATIO_KERNEL(k_pkmax_fmul, PKMAX3H(8) FMUL(0) PKMAX3H(9) FMUL(1) PKMAX3H(10) FMUL(2) PKMAX3H(11) FMUL(3))
RATIO_KERNEL(k_pkmax_fadd, PKMAX3H(8) FADD(0) PKMAX3H(9) FADD(1) PKMAX3H(10) FADD(2) PKMAX3H(11) FADD(3))
RATIO_KERNEL(k_pkmax_pkfadd, PKMAX3H(8) PKFADD(0) PKMAX3H(9) PKFADD(1) PKMAX3H(10) PKFADD(2) PKMAX3H(11) PKFADD(3))

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned* out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct { const char* name; void (*k)(unsigned*, unsigned); } ks[] = {
        {"int : v_add_u32 + v_max3_i32", k_int}, {"flt : v_add_f32 + v_max3_f32", k_flt},
        {"v_add_f32 + v_max3_i32", k_int_fadd}, {"v_add_u32 + v_max3_f32", k_flt_iadd},
        {"v_pk_add_u16 + v_pk_maximum3_f16", k_pk16}, {"v_add_f16 + v_pk_maximum3_f16", k_f16add_pkmax}};
    for (auto& en : ks) {
        for (int wpc : {2, 4}) {
            const int grid = cus * wpc;
            hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 1u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr = (double)grid * 256.0 * ITER * 32.0;
            printf("%-36s waves/CU=%2d  %8.1f Glane-instr/s  %6.1f lanes/clk/CU@2.4G (1:1 mix)\n", en.name, wpc * 4,
                   instr / ms / 1e6, instr / (ms * 1e-3) / (cus * 2.4e9));
        }
    }
    struct { const char* name; void (*k)(unsigned*, unsigned); int instr; double cells; } rs[] = {
        {"f32 x2 cells: 8 add_f32 + 7 max3_f32", k_f32_8a7m, 15, 2}, {"f32 x2 cells: 2 pk_add + 4 add + 7 max3", k_f32_2pk4a7m, 13, 2},
        {"f32 x2 cells: 4 pk_add_f32 + 7 max3", k_f32_4pk7m, 11, 2}, {"v_pk_add_f32 alone (8)", k_pkfadd, 8, 0},
        {"f16x2 x2 pairs: 8 pk_add + 7 pk_max3 + 2 perm", k_h2_shipped, 17, 4}, {"4 pk_max3_f16 + 4 v_mov", k_pkmax_mov, 8, 0},
        {"4 pk_max3_f16 + 4 v_and", k_pkmax_and, 8, 0}, {"4 pk_max3_f16 + 4 v_mul_f32", k_pkmax_fmul, 8, 0},
        {"4 pk_max3_f16 + 4 v_add_f32", k_pkmax_fadd, 8, 0}, {"4 pk_max3_f16 + 4 v_pk_add_f32", k_pkmax_pkfadd, 8, 0}};
    for (auto& en : rs) {
        const int grid = cus * 4;
        hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 1u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 2u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)grid * 256.0 * ITER * en.instr;
        const double cells = (double)grid * 256.0 * ITER * en.cells;
        printf("%-48s %6.1f lanes/clk/CU  %8.0f G cells/s\n", en.name, instr / (ms * 1e-3) / (cus * 2.4e9), cells / ms / 1e6);
    }
    return 0;
}
