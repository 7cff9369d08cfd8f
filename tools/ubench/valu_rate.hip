// valu_rate.hip — instruction-issue microbenchmark for gfx950: how many lane-operations per second do
// the instructions of the Smith-Waterman inner loop sustain?  The answer is the VALU roofline that
// bench.py / DESIGN.md price the DP kernel against (the kernel is VALU-issue bound, not HBM bound).
//
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
//
// Each kernel runs ITER iterations of 32 independent instructions of one kind (8 accumulators x 4)
// on every lane of (256 CUs x WAVES_PER_CU) waves and reports G lane-instr/s and instr/clk/CU at the
// nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITER = 16384;

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY4(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

#define DEFINE_KERNEL(NAME, ASM3)                                                        \
    __global__ void __launch_bounds__(256) NAME(unsigned* out, unsigned seed) {          \
        unsigned a[8];                                                                   \
        unsigned b = seed + threadIdx.x, c = seed * 3u + 1u;                             \
        for (int i = 0; i < 8; i++) a[i] = seed + i + threadIdx.x;                       \
        for (int it = 0; it < ITER; it++) {                                              \
            BODY4(ASM3)                                                                  \
        }                                                                                \
        unsigned r = 0;                                                                  \
        for (int i = 0; i < 8; i++) r ^= a[i];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                  \
    }

#define OP_PK_ADD_U16(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MAX_I16(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_SUB_U16C(i) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(b));
#define OP_PK_ADD_F16(i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MAX_F16(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MAX3_F16(i) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(0x05040100));
#define OP_ADD_U32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX_I32(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX3_I32(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX3_F32(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_FMA_F32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PK_FMA_F32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(unsigned long long*)&a[i & 6]) : "v"(*(unsigned long long*)&a[(i & 6) ^ 2]));
#define OP_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
#define OP_ADD_DPP(i) asm volatile("v_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX_DPP(i) asm volatile("v_max_i32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PK_MAD_I16(i) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX_I16(i) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD_U16_SDWA(i) asm volatile("v_add_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:BYTE_2" : "+v"(a[i]) : "v"(b));

#define OP_ADD_F32(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX_F32(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MUL_F32(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD_F16(i) asm volatile("v_add_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX_F16(i) asm volatile("v_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD_U16(i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_SUB_U16(i) asm volatile("v_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX_U16(i) asm volatile("v_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MIN_I32(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX_U32(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_SUB_U32(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_AND_B32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_XOR_B32(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_LSHL_B32(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));
#define OP_MOV_B32(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
#define OP_ADD3_U32(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAX3_I16(i) asm volatile("v_max3_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MED3_I32(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a[i]) : "v"(b));
#define OP_LSHL_OR(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(b));
#define OP_AND_OR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_PK_ADD_I16(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MAX_U16(i) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MIN_F16(i) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_MUL_F16(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_PK_FMA_F16(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define OP_MAD_I32_I24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
DEFINE_KERNEL(k_add_f32, OP_ADD_F32)
DEFINE_KERNEL(k_max_f32, OP_MAX_F32)
DEFINE_KERNEL(k_mul_f32, OP_MUL_F32)
DEFINE_KERNEL(k_add_f16, OP_ADD_F16)
DEFINE_KERNEL(k_max_f16, OP_MAX_F16)
DEFINE_KERNEL(k_add_u16, OP_ADD_U16)
DEFINE_KERNEL(k_sub_u16, OP_SUB_U16)
DEFINE_KERNEL(k_max_u16, OP_MAX_U16)
DEFINE_KERNEL(k_min_i32, OP_MIN_I32)
DEFINE_KERNEL(k_max_u32, OP_MAX_U32)
DEFINE_KERNEL(k_sub_u32, OP_SUB_U32)
DEFINE_KERNEL(k_and_b32, OP_AND_B32)
DEFINE_KERNEL(k_xor_b32, OP_XOR_B32)
DEFINE_KERNEL(k_lshl_b32, OP_LSHL_B32)
DEFINE_KERNEL(k_mov_b32, OP_MOV_B32)
DEFINE_KERNEL(k_cndmask, OP_CNDMASK)
DEFINE_KERNEL(k_add3_u32, OP_ADD3_U32)
DEFINE_KERNEL(k_max3_i16, OP_MAX3_I16)
DEFINE_KERNEL(k_med3_i32, OP_MED3_I32)
DEFINE_KERNEL(k_bfi, OP_BFI)
DEFINE_KERNEL(k_alignbit, OP_ALIGNBIT)
DEFINE_KERNEL(k_lshl_or, OP_LSHL_OR)
DEFINE_KERNEL(k_and_or, OP_AND_OR)
DEFINE_KERNEL(k_pk_add_i16, OP_PK_ADD_I16)
DEFINE_KERNEL(k_pk_max_u16, OP_PK_MAX_U16)
DEFINE_KERNEL(k_pk_min_f16, OP_PK_MIN_F16)
DEFINE_KERNEL(k_pk_mul_f16, OP_PK_MUL_F16)
DEFINE_KERNEL(k_pk_fma_f16, OP_PK_FMA_F16)
DEFINE_KERNEL(k_mad_i32_i24, OP_MAD_I32_I24)
DEFINE_KERNEL(k_pk_add_u16, OP_PK_ADD_U16)
DEFINE_KERNEL(k_pk_max_i16, OP_PK_MAX_I16)
DEFINE_KERNEL(k_pk_sub_u16_clamp, OP_PK_SUB_U16C)
DEFINE_KERNEL(k_pk_add_f16, OP_PK_ADD_F16)
DEFINE_KERNEL(k_pk_max_f16, OP_PK_MAX_F16)
DEFINE_KERNEL(k_pk_maximum3_f16, OP_PK_MAX3_F16)
DEFINE_KERNEL(k_perm_b32, OP_PERM)
DEFINE_KERNEL(k_add_u32, OP_ADD_U32)
DEFINE_KERNEL(k_max_i32, OP_MAX_I32)
DEFINE_KERNEL(k_max3_i32, OP_MAX3_I32)
DEFINE_KERNEL(k_max3_f32, OP_MAX3_F32)
DEFINE_KERNEL(k_fma_f32, OP_FMA_F32)
DEFINE_KERNEL(k_mov_dpp, OP_MOV_DPP)
DEFINE_KERNEL(k_add_u32_dpp, OP_ADD_DPP)
DEFINE_KERNEL(k_max_i32_dpp, OP_MAX_DPP)
DEFINE_KERNEL(k_pk_mad_i16, OP_PK_MAD_I16)
DEFINE_KERNEL(k_max_i16, OP_MAX_I16)
DEFINE_KERNEL(k_add_u16_sdwa, OP_ADD_U16_SDWA)

// LDS read rate: every lane reads 16 bytes at lane*16 + k*1024 (the DP kernel's conflict-free pattern)
__global__ void __launch_bounds__(256) k_ds_read_b128(unsigned* out, unsigned seed) {
    __shared__ uint4 lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = make_uint4(i, seed, i, seed);
    __syncthreads();
    uint4 acc = make_uint4(0, 0, 0, 0);
    unsigned base = (threadIdx.x & 63) + (seed & 1);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            const uint4 v = lds[(base + k * 64 + it) & 4095];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

typedef void (*kern_t)(unsigned*, unsigned);

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, cus, prop.clockRate / 1000);
    struct Entry { const char* name; kern_t k; };
    std::vector<Entry> entries = {
        {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_sub_u16 clamp", k_pk_sub_u16_clamp},
        {"v_pk_add_f16", k_pk_add_f16}, {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_maximum3_f16", k_pk_maximum3_f16},
        {"v_perm_b32", k_perm_b32}, {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32}, {"v_max3_i32", k_max3_i32},
        {"v_max3_f32", k_max3_f32}, {"v_fma_f32", k_fma_f32}, {"v_mov_b32_dpp", k_mov_dpp},
        {"v_add_u32_dpp", k_add_u32_dpp}, {"v_max_i32_dpp", k_max_i32_dpp}, {"v_pk_mad_i16", k_pk_mad_i16},
        {"v_max_i16", k_max_i16}, {"v_add_u16_sdwa", k_add_u16_sdwa}, {"ds_read_b128", k_ds_read_b128},
        {"v_add_f32", k_add_f32}, {"v_max_f32", k_max_f32}, {"v_mul_f32", k_mul_f32}, {"v_add_f16", k_add_f16},
        {"v_max_f16", k_max_f16}, {"v_add_u16", k_add_u16}, {"v_sub_u16", k_sub_u16}, {"v_max_u16", k_max_u16},
        {"v_min_i32", k_min_i32}, {"v_max_u32", k_max_u32}, {"v_sub_u32", k_sub_u32}, {"v_and_b32", k_and_b32},
        {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshl_b32}, {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask},
        {"v_add3_u32", k_add3_u32}, {"v_max3_i16", k_max3_i16}, {"v_med3_i32", k_med3_i32}, {"v_bfi_b32", k_bfi},
        {"v_alignbit_b32", k_alignbit}, {"v_lshl_or_b32", k_lshl_or}, {"v_and_or_b32", k_and_or},
        {"v_pk_add_i16", k_pk_add_i16}, {"v_pk_max_u16", k_pk_max_u16}, {"v_pk_min_f16", k_pk_min_f16},
        {"v_pk_mul_f16", k_pk_mul_f16}, {"v_pk_fma_f16", k_pk_fma_f16}, {"v_mad_i32_i24", k_mad_i32_i24},
    };
    const int wgs_per_cu_list[] = {2, 4};
    unsigned* out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 256 * cus * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-22s %8s %14s %16s\n", "instruction", "waves/CU", "Glane-instr/s", "lanes/clk/CU@2.4G");
    for (auto& en : entries) {
        for (int wpc : wgs_per_cu_list) {
            const int grid = cus * wpc;
            hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 1u);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(en.k, dim3(grid), dim3(256), 0, 0, out, 2u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr = (double)grid * 256.0 * ITER * 32.0;
            const double rate = instr / (ms * 1e-3);
            printf("%-22s %8d %14.1f %16.1f   (%.3f ms)\n", en.name, wpc * 4, rate / 1e9, rate / (cus * 2.4e9), ms);
        }
    }
    return 0;
}
