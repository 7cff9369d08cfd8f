// Codegen probe: which packed 16-bit / DPP instructions does hipcc emit for gfx950?
#include <hip/hip_runtime.h>
typedef short    s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void k_i16(const int* in, int* out) {
    int tid = threadIdx.x;
    s16x2 a = __builtin_bit_cast(s16x2, in[tid]);
    s16x2 b = __builtin_bit_cast(s16x2, in[tid + 64]);
    u16x2 c = __builtin_bit_cast(u16x2, in[tid + 128]);
    s16x2 t = a + b;
    s16x2 m = __builtin_elementwise_max(t, b);
    u16x2 g = {11, 11};
    u16x2 s = __builtin_elementwise_sub_sat(c, g);
    s16x2 r = __builtin_elementwise_max(m, __builtin_bit_cast(s16x2, s));
    out[tid] = __builtin_bit_cast(int, r);
}
__global__ void k_f16(const int* in, int* out) {
    int tid = threadIdx.x;
    f16x2 a = __builtin_bit_cast(f16x2, in[tid]);
    f16x2 b = __builtin_bit_cast(f16x2, in[tid + 64]);
    f16x2 c = __builtin_bit_cast(f16x2, in[tid + 128]);
    f16x2 t = a + b;
    f16x2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(t, b), c);
    f16x2 z = {0, 0};
    f16x2 r = __builtin_elementwise_max(m, z);
    out[tid] = __builtin_bit_cast(int, r);
}
__global__ void k_dpp(const int* in, int* out) {
    int tid = threadIdx.x;
    int x = in[tid];
    int y = in[tid + 64];
    // row_shr:1 = 0x111, wave_shr:1 = 0x138
    int a = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
    int b = __builtin_amdgcn_update_dpp(y, x, 0x111, 0xf, 0xf, false);
    int c = __builtin_amdgcn_update_dpp(y, x, 0x138, 0xf, 0xf, false);
    int d = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true) + 16;
    int e = __builtin_amdgcn_perm(x, y, 0x05040100);
    out[tid] = a ^ b ^ c ^ d ^ e;
}
