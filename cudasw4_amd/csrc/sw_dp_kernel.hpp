// sw_dp_kernel.hpp — the Smith-Waterman DP-matrix fill for gfx950 (MI355X), all four arithmetic kinds.
//
// Replaces the reference's NW_local_affine_{single,multi}_pass_{half2,dpx_s16,dpx_s32,float}
// kernels (half2_kernels.cuh:798-1125, dpx_s16_kernels.cuh:765-1056, dpx_s32_kernels.cuh:777-1017,
// float_kernels.cuh:788-1023).  NOT a translation: the reference keeps the SUBJECT in registers and
// streams the query through a 32-lane warp with a 37 KB pair table gathered from shared memory.
// Here the orientation is transposed and built around CDNA4's DPP rows:
//
//   * one alignment group == one DPP row of 16 lanes; a wave64 runs 4 independent groups (for the few long
//     subjects of a real DB: group == the whole wave, 64 lanes, wave_shr:1);
//   * the QUERY is tiled in stripes of LANES*R rows; lane l of a group owns R consecutive query rows
//     and keeps their H (previous column) and E (horizontal gap) state in VGPRs;
//   * subject letters stream through the row: at step t lane l works on subject column t-l
//     (anti-diagonal wavefront).  H and F of a lane's bottom row go to lane l+1 with
//     v_mov_b32_dpp row_shr:1 (lane 0 is fed by bound_ctrl zero-fill or by the previous stripe);
//   * the substitution scores come from a per-query PROFILE tile in LDS: for subject letter c the
//     lane reads its R scores as contiguous 16-byte chunks (ds_read_b128), laid out so that the
//     bank of an access depends on the lane only -> conflict-free for any mix of letters;
//   * packed kinds run two subjects per group in the two 16-bit halves (v_pk_add_f16 / v_pk_maximum3_f16; the int16
//     kind adds with v_pk_add_u16 / v_pk_sub_u16 and compares its biased bit patterns with the same fp16 maximum);
//     with 16-lane groups their profile entries are WIDE words (score, 1), so that one v_pk_fma_f16 / v_pk_mad_u16
//     with op_sel pairs the two subjects' scores AND adds them to the diagonal (Arith::add_pair);
//   * the recurrence runs in a column-offset frame (dp_step<OFFS>): every value of subject column j is kept raised by
//     |gex| * (j mod K + LANES), which takes the "+ gex" out of the horizontal gap state, and lane-local row r by a
//     further |gex| * (r mod P) (row classes), which takes it out of the vertical gap state except where the class
//     wraps: 5.8 instead of 8.5 instructions per cell pair; the plain form remains for gap-extension scores too
//     large for any K;
//   * queries longer than one stripe are processed stripe after stripe by the same group; the H/F
//     row at the stripe border is spilled to a small global scratch (branch-free: 8 bytes stored per
//     step, 32 bytes loaded per four steps), the analogue of the reference's devTempHcol2/devTempEcol2;
//   * persistent workgroups pull batches of groups from an atomic counter, longest subjects first.
//
// Padding is self-neutralising exactly as in the reference: query rows >= Q and
// subject columns >= len use letter code 20 whose scores are all negative.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace swk {

typedef uint32_t u32;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kGroup = 16;        // lanes per alignment group in the standard shape (one DPP row)
constexpr int kThreads = 256;     // workgroup size: 4 waves = 16 row groups or 4 wave-wide groups
constexpr int kPadLetter = 20;    // code used for padding rows / columns
constexpr int kLetters = 21;

enum { F16X2 = 0, I16X2 = 1, I32 = 2, F32 = 3 };

// DPP controls (GFX9 encoding)
constexpr int DPP_ROW_SHL1 = 0x101;   // lane i <- lane i+1 (within a row of 16)
constexpr int DPP_ROW_SHR1 = 0x111;   // lane i <- lane i-1
constexpr int DPP_ROW_ROR1 = 0x121;
constexpr int DPP_QUAD_SHR1 = 0x90;   // quad_perm [0,0,1,2]: lane i <- lane i-1 within its quad (lane 0 of a quad: itself)
constexpr int DPP_WAVE_SHL1 = 0x130;  // the same across all 64 lanes of the wave
constexpr int DPP_WAVE_SHR1 = 0x138;

// The group shape: LANES = 16 (one DPP row per alignment group, 4 groups per wave: the bulk of a DB),
// LANES = 64 (the whole wave is one anti-diagonal pipeline: long subjects, 4x the lanes per alignment so
// that the few giant sequences of a real DB do not become the tail of the scan), or
// LANES = 8 (half a DPP row, 8 groups per wave: SHORT QUERIES.  A query of Q rows gives every lane Q/8 instead of Q/16
// rows, so the per-step work that does not scale with the rows (lane exchange, letter handling, window upkeep: ~10
// instructions) is spread over twice the cells, and the pipeline fill per subject is 7 instead of 15 steps.  The
// row_shr:1 exchange crosses from lane 7 into lane 8, the head of the second group of the row, which costs one select
// per exchanged value (prev_lane).  With several stripes the shape pays off for short subjects only: half the fill per
// stripe, but twice the stripes.), or
// LANES = 4 (a DPP quad, 16 groups per wave: VERY SHORT QUERIES, single-stripe only.  A 48-residue query gives an 8-lane
// group 6 rows per lane — (6.5 * 6 + 19) / 6 = 9.67 instructions per cell pair, SQ_INSTS_VALU says the same — and a quad 12:
// 8.08.  The exchange is a quad_perm, every group's head selects its boundary value like lane 8 above.)
template <int LANES>
struct Shift {
    static_assert(LANES == 4 || LANES == 8 || LANES == 16 || LANES == 64, "group = a DPP quad, half a DPP row, a DPP row or the whole wave");
    static constexpr int kShr1 = LANES <= 16 ? DPP_ROW_SHR1 : DPP_WAVE_SHR1;
    static constexpr int kShl1 = LANES <= 16 ? DPP_ROW_SHL1 : DPP_WAVE_SHL1;
};

template <int CTRL, bool ZERO_FILL>
__device__ __forceinline__ u32 dpp(u32 old, u32 src) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, 0xf, ZERO_FILL);
}

// The value of the previous lane of the group; the group's head lane takes `headval` (ZERO_FILL: headval must be 0 and
// the head is filled by bound_ctrl).  8-lane groups: lane 8 of a DPP row is a head too but has a source lane, so it
// is selected explicitly (`head` is loop-invariant: one v_cndmask with an SGPR mask).
template <int LANES, bool ZERO_FILL>
__device__ __forceinline__ u32 prev_lane(u32 headval, u32 src, bool head) {
    if constexpr (LANES == 8) {
        const u32 v = dpp<DPP_ROW_SHR1, false>(headval, src);
        return head ? headval : v;
    } else if constexpr (LANES == 4) {
        const u32 v = dpp<DPP_QUAD_SHR1, false>(headval, src);
        return head ? headval : v;
    } else {
        (void)head;
        return dpp<Shift<LANES>::kShr1, ZERO_FILL>(headval, src);
    }
}

// ------------------------------------------------------------------------------------------------
// Arithmetic traits.  All state lives in 32-bit registers; E and F are kept CLAMPED at zero
// (E' = max(E,0), F' = max(F,0)), which leaves H = max(diag+s, E', F') unchanged and makes the
// explicit max(.,0) of the reference recurrence (half2_kernels.cuh:176) free.
// ------------------------------------------------------------------------------------------------
#ifndef SWK_F32_WINDOW
#define SWK_F32_WINDOW 0
#endif
template <int KIND>
struct Arith;

// int16 kind.  State is kept BIASED by 1024 in unsigned 16-bit fields: H~ = H + 1024, E~ = E' + 1024, F~ likewise.
// Adds/subtracts are integer (v_pk_add_u16 / v_pk_sub_u16); every maximum is taken with the fp16 comparator
// v_pk_maximum3_f16 ON THE INTEGER BIT PATTERNS: for patterns in [0x0400, 0x7BFF] (positive normal halves) the
// fp16 order equals the integer order and the instruction returns one of its inputs bit-exactly, so the
// 3-input maximum the int16 ISA lacks comes for free (8.5 instead of 10 instructions per cell pair).
//   * diag+s can dip below 1024 (a denormal pattern, possibly flushed): it then loses against E~ >= 1024
//     anyway, so the result is unaffected;
//   * patterns >= 0x7C00 (inf/NaN, H >= 30720) only arise after the running maximum has passed the
//     25000 overflow limit (H grows by at most max(s) per anti-diagonal), and a NaN maximum is itself
//     >= the limit, so such subjects are always flagged and re-scored in 32 bits.
template <>
struct Arith<I16X2> {
    static constexpr bool kPacked = true;
    static constexpr int kSubjects = 2;
    static constexpr int kLimit = 25000;  // kernels.cuh:5 MAX_ACC_SHORT
    static constexpr bool kWindow = true;  // StripeState: zero levels and maxima as windows over a quad
    static constexpr int kBias = 1024;
    static constexpr u32 kZero = (u32)kBias | ((u32)kBias << 16);  // the value 0 in both halves
    // gap scores are <= 0; magnitudes are subtracted (|g| <= 1000 keeps E~ - |g| non-negative)
    static __host__ __device__ u32 encode_gap(int g) { u32 m = (u32)(-g) & 0xffffu; return m | (m << 16); }
    static __host__ __device__ u32 encode_score(int s) { return (u32)(uint16_t)(int16_t)s; }
    static __device__ __forceinline__ u32 add(u32 a, u32 b) {
        return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
    }
    static __device__ __forceinline__ u32 max3(u32 a, u32 b, u32 c) {
        return __builtin_bit_cast(u32, __builtin_elementwise_maximum(
            __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
            __builtin_bit_cast(f16x2, c)));
    }
    static __device__ __forceinline__ u32 max2(u32 a, u32 b) {
        return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)));
    }
    static __device__ __forceinline__ u32 cell_h(u32 t, u32 e, u32 f) { return max3(t, e, f); }
    static __device__ __forceinline__ u32 gap(u32 a, u32 g) {
        return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, g)));
    }
    static __device__ __forceinline__ u32 gap_state(u32 ext, u32 open) { return max3(ext, open, kZero); }
    static __device__ __forceinline__ u32 fold2(u32 m, u32 a, u32 b) { return max3(m, a, b); }
    static __device__ __forceinline__ int score_lo(u32 v) { return (int)(v & 0xffffu) - kBias; }
    static __device__ __forceinline__ int score_hi(u32 v) { return (int)(v >> 16) - kBias; }
    // column-offset recurrence (dp_step<OFFS>): the zero level of column offset k, +a, star -> true values
    static __host__ __device__ u32 zero_at(int a, int k) { u32 z = (u32)(kBias + a * k) & 0xffffu; return z | (z << 16); }
    static __host__ __device__ u32 pos_word(int a) { u32 m = (u32)a & 0xffffu; return m | (m << 16); }
    // the zero level v (sw_stream_kernel.hpp: levels with a base and jumps) in both halves
    static __device__ __forceinline__ u32 level_word(int v) { const u32 z = (u32)(kBias + v) & 0xffffu; return z | (z << 16); }
    static __device__ __forceinline__ u32 true_of(u32 m, u32 z) { return gap(m, z); }  // plain unsigned integers
    static __device__ __forceinline__ u32 true_max(u32 a, u32 b) {
        return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
    }
    static __device__ __forceinline__ int true_lo(u32 v) { return (int)(v & 0xffffu); }
    static __device__ __forceinline__ int true_hi(u32 v) { return (int)(v >> 16); }
    // wide profile words (score, 1): (wa.lo * wb.hi + c.lo, wa.hi * wb.lo + c.hi) = c + (score A, score B) in ONE
    // v_pk_mad_u16 (op_sel swaps the halves of wb) — replaces the v_perm_b32 + add of the two-rows-per-word layout
    static constexpr u32 kOne = 1u;
    static __device__ __forceinline__ u32 add_pair(u32 wa, u32 wb, u32 c) {
        u32 d;
        asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(wa), "v"(wb), "v"(c));
        return d;
    }
};

template <>
struct Arith<F16X2> {
    static constexpr bool kPacked = true;
    static constexpr int kSubjects = 2;
    static constexpr int kLimit = 2048;  // kernels.cuh:4 MAX_ACC_HALF2
    static constexpr bool kWindow = true;
    static constexpr u32 kZero = 0u;
    static __host__ __device__ u32 half_bits(int v) {
        // exact conversion of a small integer |v| < 2048 to IEEE binary16 bits
        if (v == 0) return 0;
        u32 sign = v < 0 ? 0x8000u : 0u;
        u32 a = (u32)(v < 0 ? -v : v);
        int e = 0;
        while ((a >> (e + 1)) != 0) e++;
        u32 mant = (a << (10 - e)) & 0x3ffu;
        return sign | ((u32)(e + 15) << 10) | mant;
    }
    static __host__ __device__ u32 encode_gap(int g) { u32 h = half_bits(g); return h | (h << 16); }
    static __host__ __device__ u32 encode_score(int s) { return half_bits(s); }
    static __device__ __forceinline__ u32 add(u32 a, u32 b) {
        return __builtin_bit_cast(u32, (f16x2)(__builtin_bit_cast(f16x2, a) + __builtin_bit_cast(f16x2, b)));
    }
    static __device__ __forceinline__ u32 max3(u32 a, u32 b, u32 c) {
        return __builtin_bit_cast(u32, __builtin_elementwise_maximum(
            __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)),
            __builtin_bit_cast(f16x2, c)));
    }
    static __device__ __forceinline__ u32 max2(u32 a, u32 b) {
        return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b)));
    }
    static __device__ __forceinline__ u32 cell_h(u32 t, u32 e, u32 f) { return max3(t, e, f); }
    static __device__ __forceinline__ u32 gap(u32 a, u32 g) { return add(a, g); }  // not clamped yet
    static __device__ __forceinline__ u32 gap_state(u32 ext, u32 open) { return max3(ext, open, 0u); }
    static __device__ __forceinline__ u32 fold2(u32 m, u32 a, u32 b) { return max3(m, a, b); }  // one v_pk_maximum3_f16 per two rows
    static __device__ __forceinline__ int score_lo(u32 v) { return (int)(float)__builtin_bit_cast(f16x2, v).x; }
    static __device__ __forceinline__ int score_hi(u32 v) { return (int)(float)__builtin_bit_cast(f16x2, v).y; }
    static __host__ __device__ u32 zero_at(int a, int k) { u32 h = half_bits(a * k); return h | (h << 16); }
    static __host__ __device__ u32 pos_word(int a) { u32 h = half_bits(a); return h | (h << 16); }
    // the zero level v (|v| <= 2048: exact) in both halves, by the hardware conversion
    static __device__ __forceinline__ u32 level_word(int v) {
        const u32 h = (u32)__builtin_bit_cast(unsigned short, (_Float16)(float)v);
        return h | (h << 16);
    }
    static __device__ __forceinline__ u32 true_of(u32 m, u32 z) {
        return __builtin_bit_cast(u32, (f16x2)(__builtin_bit_cast(f16x2, m) - __builtin_bit_cast(f16x2, z)));
    }
    static __device__ __forceinline__ u32 true_max(u32 a, u32 b) { return max2(a, b); }
    static __device__ __forceinline__ int true_lo(u32 v) { return score_lo(v); }
    static __device__ __forceinline__ int true_hi(u32 v) { return score_hi(v); }
    // wide profile words (score, 1.0): one v_pk_fma_f16 with op_sel (exact: a product with 1.0, one rounding of an integer sum)
    static constexpr u32 kOne = 0x3c00u;
    static __device__ __forceinline__ u32 add_pair(u32 wa, u32 wb, u32 c) {
        // spelled out: from a shufflevector hipcc folds the swap into op_sel for three words of a 16-byte LDS chunk but
        // rotates the first one with an extra v_alignbit_b32
        u32 d;
        asm("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(wa), "v"(wb), "v"(c));
        return d;
    }
};

template <>
struct Arith<I32> {
    static constexpr bool kPacked = false;
    static constexpr int kSubjects = 1;
    static constexpr int kLimit = 0x7fffffff;
    static constexpr bool kWindow = true;
    static constexpr u32 kZero = 0u;
    static __host__ __device__ u32 encode_gap(int g) { return (u32)g; }
    static __host__ __device__ u32 encode_score(int s) { return (u32)s; }
    // v_max3_i32 spelled out: hipcc's own selection mixes signed/unsigned 2- and 3-input forms here
    // (4.4 max instructions per cell instead of 3.5).  The compiler puts a wait state behind each inline-asm result that
    // is read at once; the plain-C++ form has none but spills more under the occupancy bound (5.6 vs 6.0 TCUPS)
    static __device__ __forceinline__ u32 max3(u32 a, u32 b, u32 c) {
        u32 d;
        asm("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    static __device__ __forceinline__ u32 max3_zero(u32 a, u32 b) {
        u32 d;
        asm("v_max3_i32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    static __device__ __forceinline__ u32 add(u32 a, u32 b) { return a + b; }
    static __device__ __forceinline__ u32 max2(u32 a, u32 b) { return (u32)((int)a > (int)b ? (int)a : (int)b); }
    static __device__ __forceinline__ u32 cell_h(u32 t, u32 e, u32 f) { return max3(t, e, f); }
    static __device__ __forceinline__ u32 gap(u32 a, u32 g) { return a + g; }
    static __device__ __forceinline__ u32 gap_state(u32 ext, u32 open) { return max3_zero(ext, open); }
    static __device__ __forceinline__ u32 fold2(u32 m, u32 a, u32 b) { return max3(m, a, b); }
    static __device__ __forceinline__ int score_lo(u32 v) { return (int)v; }
    static __device__ __forceinline__ int score_hi(u32) { return 0; }
    static __host__ __device__ u32 zero_at(int a, int k) { return (u32)(a * k); }
    static __host__ __device__ u32 pos_word(int a) { return (u32)a; }
    static __device__ __forceinline__ u32 true_of(u32 m, u32 z) { return m - z; }
    static __device__ __forceinline__ u32 true_max(u32 a, u32 b) { return max2(a, b); }
    static __device__ __forceinline__ int true_lo(u32 v) { return (int)v; }
    static __device__ __forceinline__ int true_hi(u32) { return 0; }
};

template <>
struct Arith<F32> {
    static constexpr bool kPacked = false;
    static constexpr int kSubjects = 1;
    static constexpr int kLimit = 0x7fffffff;
    // the fp32 kernels (168 VGPRs for three waves per SIMD) spill in their loops with the 7 extra window registers
    // (7.9 -> 1.6 TCUPS): they add a per step to P zero levels and P maxima instead
    static constexpr bool kWindow = SWK_F32_WINDOW != 0;
    static constexpr u32 kZero = 0u;
    static __host__ __device__ u32 encode_gap(int g) { return __builtin_bit_cast(u32, (float)g); }
    static __host__ __device__ u32 encode_score(int s) { return __builtin_bit_cast(u32, (float)s); }
    static __device__ __forceinline__ float f(u32 v) { return __builtin_bit_cast(float, v); }
    static __device__ __forceinline__ u32 u(float v) { return __builtin_bit_cast(u32, v); }
    static __device__ __forceinline__ u32 add(u32 a, u32 b) { return u(f(a) + f(b)); }
    static __device__ __forceinline__ u32 max2(u32 a, u32 b) { return u(__builtin_fmaxf(f(a), f(b))); }
    static __device__ __forceinline__ u32 max3(u32 a, u32 b, u32 c) { return u(__builtin_fmaxf(__builtin_fmaxf(f(a), f(b)), f(c))); }
    static __device__ __forceinline__ u32 cell_h(u32 t, u32 e, u32 ff) { return u(__builtin_fmaxf(__builtin_fmaxf(f(t), f(e)), f(ff))); }
    static __device__ __forceinline__ u32 gap(u32 a, u32 g) { return u(f(a) + f(g)); }
    static __device__ __forceinline__ u32 gap_state(u32 ext, u32 open) { return u(__builtin_fmaxf(__builtin_fmaxf(f(ext), f(open)), 0.0f)); }
    static __device__ __forceinline__ u32 fold2(u32 m, u32 a, u32 b) { return u(__builtin_fmaxf(__builtin_fmaxf(f(m), f(a)), f(b))); }  // v_max3_f32
    static __device__ __forceinline__ int score_lo(u32 v) { return (int)f(v); }
    static __device__ __forceinline__ int score_hi(u32) { return 0; }
    static __host__ __device__ u32 zero_at(int a, int k) { return __builtin_bit_cast(u32, (float)(a * k)); }
    static __host__ __device__ u32 pos_word(int a) { return __builtin_bit_cast(u32, (float)a); }
    static __device__ __forceinline__ u32 true_of(u32 m, u32 z) { return u(f(m) - f(z)); }
    static __device__ __forceinline__ u32 true_max(u32 a, u32 b) { return max2(a, b); }
    static __device__ __forceinline__ int true_lo(u32 v) { return (int)f(v); }
    static __device__ __forceinline__ int true_hi(u32) { return 0; }
};

// ------------------------------------------------------------------------------------------------
// Profile tile geometry (shared by the profile builder and the kernel).
//   NW     32-bit words a lane reads per letter per step (ceil(R/2) packed, R scalar)
//   NCH    16-byte chunks per lane per letter
//   letter row  = NCH chunks-rows of 256 bytes: chunk k of lane l at  k*256 + l*16
//   tile        = 21 letter rows, preceded by 16 bytes (lane addresses carry a +16 bias, see Step)
// ------------------------------------------------------------------------------------------------
// How the NW words a lane reads per letter are cut into 16-byte LDS slots ("chunks").  LDS serves ds_read_b128 and
// ds_read_b96 of this layout (one 16-byte slot per lane, letter rows a multiple of 256 bytes apart) without bank
// conflicts for ANY mix of letter rows across the groups of a wave, but not ds_read_b32 / ds_read_b64: those are
// processed 64 / 32 lanes at a time, and the lanes of different groups then meet in the same banks at different
// addresses (measured on ragged subjects, where the groups read different letters: 11-37 % of the LDS cycles of the
// NW mod 4 = 1 kernels were conflicts, 4-12 % for NW mod 4 = 2; the identical subjects of the peak DB hid it).  So a
// word count that is no multiple of 4 is cut into 4-word chunks and one to three 3-WORD chunks (42 = 9 x 4 + 2 x 3,
// 45 = 9 x 4 + 3 x 3, 43 = 10 x 4 + 3): the same number of reads and of LDS bytes as full chunks plus a short tail, no
// register holds a word nobody needs (reading the short tail as b96 and dropping a word cost 2.5 % in the 256-VGPR
// kernels), and every read is conflict-free.  NW = 1, 2, 5 cannot be cut that way: their tail is a b96 read of which
// one or two words are used.
template <int NW>
struct Chunks {
    static constexpr int kRem = NW % 4;
    static constexpr int kWant3 = (4 - kRem) % 4;
    static constexpr bool kSplit = NW >= 3 * kWant3;
    static constexpr int kN3 = kSplit ? kWant3 : 1;                      // 3-word chunks (not kSplit: ONE partly used one)
    static constexpr int kN4 = kSplit ? (NW - 3 * kWant3) / 4 : NW / 4;  // 4-word chunks, in front
    static constexpr int kCount = kN4 + kN3;
    static constexpr int first(int c) { return c < kN4 ? 4 * c : 4 * kN4 + 3 * (c - kN4); }  // first word of chunk c
    static constexpr int words(int c) { return c < kN4 ? 4 : (kSplit ? 3 : kRem); }            // words of chunk c in use
    static constexpr int chunk_of(int w) { return w < 4 * kN4 ? w / 4 : kN4 + (w - 4 * kN4) / 3; }
    static constexpr bool starts_chunk(int w) { return first(chunk_of(w)) == w; }
    // progressive reads: `lead` chunks are issued at the top of a step, chunk c + lead when the chain reaches the first
    // word of chunk c; a word that is consumed `ahead` words before the chain reaches it must have been issued at least
    // `slack` rows earlier (the LDS latency hides behind them)
    static constexpr bool lead_ok(int lead, int ahead, int slack) {
        for (int r = 0; r < NW; r++) {
            const int need = chunk_of(r + ahead < NW ? r + ahead : NW - 1);
            if (need >= lead && first(need - lead) + slack > r) return false;
        }
        return true;
    }
    static constexpr int lead_for(int ahead, int slack) {
        int lead = 1;
        while (lead < kCount && !lead_ok(lead, ahead, slack)) lead++;
        return lead;
    }
};

template <int KIND, int R, int LANES = kGroup>
struct Geometry {
    static constexpr bool kPacked = Arith<KIND>::kPacked;
    // packed kinds, 16-lane groups: WIDE words (score of the row, 1) — the pair (score A, score B) of a cell pair and its
    // addition to the diagonal are ONE packed multiply-add (Arith::add_pair).  Wave-wide groups keep two query rows per
    // word (an odd R leaves the upper half of the last word unused) and pair the scores with v_perm_b32.
    static constexpr bool kWide = kPacked && LANES <= 16;
    static constexpr int NW = (kPacked && !kWide) ? (R + 1) / 2 : R;
    using C = Chunks<NW>;
    static constexpr int NCH = C::kCount;
    // chunk k of lane l at k*kChunkRowBytes + l*16; 8-lane groups keep the 16 slots of a DPP row: the first group of a row
    // reads slots 0..7, the second one slots 8..15 (dp_step: laneStep), which the profile builder fills with a copy —
    // sharing slots 0..7 made the two groups meet in the same banks with different letter rows
    static constexpr int kChunkRowBytes = (LANES <= 16 ? 16 : LANES) * 16;
    static constexpr int kRowBytes = NCH * kChunkRowBytes;
    static constexpr int kTileBytes = kLetters * kRowBytes;
    static constexpr int kStripeRows = LANES * R;
    // a letter travels as (letter * kLetterUnits) in one byte; its row offset is that byte << kLetterShift
    static constexpr int kLetterShift = LANES <= 16 ? 8 : 10;
    static constexpr int kLetterUnits = kRowBytes >> kLetterShift;
    static_assert(kPadLetter * kLetterUnits < 256, "letter offset must fit a byte");
};

struct ScanParams {
    const int8_t* chars;
    const uint64_t* offsets;
    const int32_t* lengths;
    const int32_t* positions;  // optional indirection list (overflow re-score); nullptr -> first_pos + i
    const int32_t* count_ptr;  // optional device-side count; nullptr -> n
    int32_t first_pos;
    int32_t n;
    const unsigned char* profile;  // nstripes tiles of Geometry::kTileBytes
    int32_t nstripes;
    u32 gop, gex;              // Arith<KIND>::encode_gap
    float* scores;
    int32_t* ids;
    int64_t id_offset;
    int32_t* ovf_pos;
    int32_t* ovf_count;
    int32_t ovf_check;
    u32* scratch;              // stripe-border spill: per (workgroup, group) border_region_words(lcap) words
    int32_t lcap;
    const u32* zeros;          // >= 64 bytes of the kind's zero pattern (border of the first stripe)
    u32* work_counter;         // zeroed before the launch: next batch to hand out
    int32_t gex_mag;           // OFFS kernels: a = -gex; then gop holds encode_gap(gop - gex) and the profile s + a
    int32_t renorm_quads;      // OFFS: K/4 — every K columns a lane lowers its frame by a*K (0: never); a power of two
    u32 renorm_word;           // encode_gap(-a*K)
    u32 wrap_class;            // OFFS: encode_gap(-a*P), P = frame_classes(...) of the launched kernel (dp_step: row classes)
    u32 wrap_last;             // OFFS: encode_gap(-a*((R-1) % P + 1))
    int32_t* stat_count;       // optional.  Scan launches (streamed kernels): += subjects listed only because the slot before them
                               // scored at or above the zero-level jump (sw_set_dirty_counter).  Re-score launches: += subjects whose exact score is >= stat_limit, i.e. the
    int32_t stat_limit;        // reference's notion of an overflow (half2_kernels.cuh:1087-1109), for the printed statistic
    // Start handshake (side launches that must run BESIDE a grid that fills the GPU): every workgroup counts itself in
    // work_counter[1] when it becomes resident, and the one that completes start_quorum adds 1 to *start_signal (system
    // scope; signal memory a stream can wait on: hipStreamWaitValue32).  nullptr: no handshake.
    u32* start_signal;
    u32 start_quorum;
    // Overflow lists that are re-scored WHILE they are filled (sw_rescore_service).  claim: the list's entries are taken
    // with a compare-and-swap (-1: not written yet, >= 0: a subject to re-score, -2: taken), so that a service launch that
    // runs beside the producing launch and the ordinary re-score launch behind it never score an entry twice; the producer
    // publishes an entry with an agent-scope store (its plain stores would sit in its XCD's L2).  service: the list's length
    // (count_ptr) is polled per batch instead of read once, a batch is taken as soon as its first entry is there, and the
    // workgroup leaves when *done_flag has reached done_value (the producer has finished) and nothing is left for it.
    int32_t* claim;            // == positions, writable; nullptr: plain list
    int32_t service;
    const u32* done_flag;
    u32 done_value;
    // Tail hand-over (sw_set_dry_signal): the workgroup that finds the work counter dry — the first one that has nothing
    // left to take, while the others are still busy with their last batches — stores dry_value in *dry_signal (system
    // scope, signal memory).  A caller that ordered the NEXT query's launch behind that value on another stream gets a
    // grid that fills the slots this launch's workgroups leave one by one, instead of one that shares the CUs with it
    // from the start.  nullptr: none.
    u32* dry_signal;
    u32 dry_value;
    // Streamed subjects (sw_stream_kernel.hpp): batches a workgroup may claim at once (<= 1: sw_scan_kernel), the most columns
    // a round of several slots may have (the border scratch of multi-stripe queries is sized for it) and the most its
    // a * columns + jumps may add up to (the frame is not lowered inside such a round); the level of column -LANES (fp16
    // starts at the bottom of its exact range); what a lane's zero levels rise by at a slot border, and from which score of
    // the slot before on a slot is flagged (its lanes may have kept values above the raised levels)
    int32_t stream_slots;
    int32_t stream_cols;
    int32_t stream_room;
    int32_t level_base;
    int32_t jump;
    u32 jump_word;
    int32_t jump_limit;
};

constexpr int32_t kListEmpty = -1, kListTaken = -2;

typedef u32 u32x3 __attribute__((ext_vector_type(3)));

// chunk c of a lane's words for one letter (Chunks<NW>); c folds to a constant in the unrolled callers
template <int NW, int CHUNK_ROW_BYTES>
__device__ __forceinline__ void lds_read_chunk(u32 (&dst)[NW], const unsigned char* p, int c) {
    using C = Chunks<NW>;
    const int w = C::first(c);
    if (c < C::kN4) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + c * CHUNK_ROW_BYTES);
        dst[w + 0] = v.x; dst[w + 1] = v.y; dst[w + 2] = v.z; dst[w + 3] = v.w;
    } else if (c < C::kCount) {
        const u32x3 v = *reinterpret_cast<const u32x3*>(p + c * CHUNK_ROW_BYTES);
        if constexpr (C::kSplit) {
            dst[w + 0] = v.x; dst[w + 1] = v.y; dst[w + 2] = v.z;
        } else {
            // NW = 1, 2, 5: one or two of the three words are used.  x passes through an empty asm that also takes the
            // unused words: a pure data dependency (no ordering against the hand-scheduled step) that keeps the compiler
            // from narrowing the load to a conflicting ds_read_b32 / b64 again
            u32 x = v.x;
            if constexpr (C::kRem == 1) asm("" : "+v"(x) : "v"(v.y), "v"(v.z));
            if constexpr (C::kRem == 2) asm("" : "+v"(x) : "v"(v.z));
            dst[w] = x;
            if constexpr (C::kRem == 2) dst[w + 1] = v.y;
        }
    }
}

template <int NW, int CHUNK_ROW_BYTES>
__device__ __forceinline__ void lds_read_words(u32 (&dst)[NW], const unsigned char* p) {
#pragma unroll
    for (int c = 0; c < Chunks<NW>::kCount; c++) lds_read_chunk<NW, CHUNK_ROW_BYTES>(dst, p, c);
}

// Row classes of the column-offset frame (dp_step<OFFS>): lane-local row r belongs to class r mod P and is kept raised
// by a further a * (r mod P).  P = 1 is the plain column frame.
// Packed single-stripe kernels up to this many rows per lane are bound to 168 VGPRs = three waves per SIMD (two waves
// fill 93.5 % of the issue slots at best, three 97 %: tools/ubench/dep_chain.hip); the multi-stripe kernels of the same
// height would spill (their border state), taller ones need the registers.
#ifndef SWK_WAVES3_MAX_R
#define SWK_WAVES3_MAX_R 36
#endif
#ifndef SWK_WAVES3_MAX_R_MULTI
// Packed MULTI-stripe kernels up to this many rows per lane are bound to 168 VGPRs as well (round 4).  With the stripe
// border in LDS rings (Border<LANES>) their loops run without a single scratch access at that bound (tools/loop_spills.py;
// what the code object reports as spills is set-up code), the 43 KB tile plus 8 KB of rings still fits three workgroups
// per CU, and the third wave is worth +6 % at the same stripe height (850 / 1000-residue queries: 10.8 / 11.0 -> 11.5 /
// 11.6 TCUPS) — enough for the planner to prefer 32-row stripes over 40..48-row ones for most long queries
// (sw_api.hip: plan_query; peak benchmark 11.53 -> 11.67 TCUPS).  0: none.
#define SWK_WAVES3_MAX_R_MULTI 32
#endif
#ifndef SWK_MULTI_SCALAR_LOOP
// multi-stripe kernels: the quad counters in scalar registers too (wave-uniform by construction).  Rounds 1-3 kept them in
// vector registers (the scalar form measured 1 % slower then); with the block loop of the LDS border rings the scalar
// form is the faster one (11.39 -> 11.44 TCUPS) and frees registers
#define SWK_MULTI_SCALAR_LOOP 1
#endif
#ifndef SWK_WAVES4_MAX_R
#define SWK_WAVES4_MAX_R 22   // ... and up to this many to 128 VGPRs = four waves per SIMD (+1 % at R = 17..22, -0.4 % at 24)
#endif
#ifndef SWK_CLASSES_PACKED
#define SWK_CLASSES_PACKED 8
#endif
#ifndef SWK_CLASSES_PACKED_SMALL
#define SWK_CLASSES_PACKED_SMALL 4
#endif
#ifndef SWK_CLASSES_SCALAR
#define SWK_CLASSES_SCALAR 4
#endif
constexpr int frame_classes(bool packed, int R, int lanes, bool multi) {
    (void)lanes;
    // packed kinds: 8 classes for the tall kernels (two waves per SIMD anyway: 0.125 instead of 0.25 wrap subtractions per
    // cell pair for 8 more registers, +0.8 % on the peak benchmark); the single-stripe kernels up to R = SWK_WAVES3_MAX_R and
    // everything below R = 25 four, which keeps those at three (or more) waves per SIMD.  32-bit kinds: four (register-bound by their occupancy).
    const int want = packed ? ((R > SWK_WAVES3_MAX_R || (multi && R >= 25 && R > SWK_WAVES3_MAX_R_MULTI)) ? SWK_CLASSES_PACKED : SWK_CLASSES_PACKED_SMALL)
                            : SWK_CLASSES_SCALAR;
    int P = want;
    while (P > 1 && 2 * P > R) P--;  // at least two rows per class, so that the running maximum still folds two rows per max3
    return P;
}

// the words of two letters, chunk by chunk (A's chunk k, B's chunk k, ...): LDS answers in order, so the first rows'
// score words arrive after two reads instead of after all of A's
template <int NW, int CHUNK_ROW_BYTES>
__device__ __forceinline__ void lds_read_chunk2(u32 (&da)[NW], u32 (&db)[NW], const unsigned char* pa, const unsigned char* pb, int c) {
    lds_read_chunk<NW, CHUNK_ROW_BYTES>(da, pa, c);
    lds_read_chunk<NW, CHUNK_ROW_BYTES>(db, pb, c);
    __builtin_amdgcn_sched_barrier(0);  // keep this issue order (the scheduler would sort the reads by register)
}

template <int NW, int CHUNK_ROW_BYTES>
__device__ __forceinline__ void lds_read_words2(u32 (&da)[NW], u32 (&db)[NW], const unsigned char* pa, const unsigned char* pb) {
#pragma unroll
    for (int c = 0; c < Chunks<NW>::kCount; c++) lds_read_chunk2<NW, CHUNK_ROW_BYTES>(da, db, pa, pb, c);
}

// the same for one letter (32-bit kinds)
template <int NW, int CHUNK_ROW_BYTES>
__device__ __forceinline__ void lds_read_chunk1(u32 (&da)[NW], const unsigned char* pa, int c) {
    lds_read_chunk<NW, CHUNK_ROW_BYTES>(da, pa, c);
    __builtin_amdgcn_sched_barrier(0);
}

// Per-group DP state that lives across the steps of one stripe.
template <int KIND, int R, int P = 1>
struct StripeState {
    u32 H[R];      // H(row, column-1)
    u32 E[R];      // horizontal gap state entering the current column (clamped)
    u32 upH_prev;  // H(row0-1, column-1): diagonal input of the lane's first row
    u32 Hlast;     // H of the lane's bottom row after the last step
    u32 Fout;      // vertical gap state leaving the lane's bottom row after the last step
    u32 yA, yB;    // LDS byte address (+16 bias) of the lane's chunk for the current letter(s); packed kinds with
                   // 16-lane groups keep both in yA (B's address in the upper half)
    // OFFS: windows over the four steps of a quad.  With j0 the column the lane works on in the quad's first step,
    //   Zc[i]   = zero level of column j0 raised by a*i: step q uses Zc[q + k] for class k (or class + 1) — the zero levels
    //             of consecutive columns and classes are the same numbers, so nothing is added per step;
    //   maxv[d] = running maximum of all cells whose frame is a*(c_j0 + d): step q folds class c into maxv[q + c].
    // Both move up by 4a once per quad (2P + 7 additions instead of 8P per quad).  The plain form uses maxv[0] only.
    u32 maxv[P + 3];
    u32 Zc[P + 4];
};

// One anti-diagonal step of one lane: R cells (or R cell pairs).
//   BYTE    which byte of the letter words feeds lane 0 in this step
//   MULTI   stripe borders in play: lane 0 takes (inH, inF) — the previous stripe's bottom row at this
//           column — instead of the zero boundary; the caller stores lane 15's (Hlast, Fout) afterwards
//
// OFFS — the column-offset form of the recurrence.  With a = -gex all values of column j are kept as
// X* = X + a*c_j (c_j = j + LANES).  Then E*(i,j+1) = max3(E*(i,j), H* + (gop + a), Z_{j+1}) needs no addition of
// its own (E decays by a per column, the frame rises by a per column), F*(i+1,j) = max3(F*, H* + (gop + a), Z_{j+1}) - a,
// the diagonal term is H*(i-1,j-1) + (s + a) with the +a folded into the profile, and Z_j = a*c_j is the zero level
// of the column: 7.5 instead of 8.5 instructions per cell pair (6.5 instead of 7.5 per cell).  Price: one register
// (Z) and two instructions per step (Z += a; running maximum += a), and magnitudes
// that grow with the column index: the launcher picks OFFS only while a * columns stays well inside the exact range
// of the kind, and a subject whose bound maxscore + a * columns reaches the limit is flagged like an overflow.
// `first` (MULTI): the stripe has no predecessor, lane 0's boundary is the zero level instead of the border row.
// ZFILL = false (sw_stream_kernel.hpp): the head lane's vertical gap state is the column's zero level (class 0) instead of the
// bound_ctrl zero fill — levels that start below zero, and separator columns, which rebuild a lane's state from it.
template <int KIND, int R, int LANES, int BYTE, bool MULTI, bool OFFS = false, int P = 1, bool ZFILL = true>
__device__ __forceinline__ void dp_step(StripeState<KIND, R, P>& st, const unsigned char* tile,
                                        u32 lettersA, u32 lettersB, u32 gop, u32 gex, u32 inH, u32 inF,
                                        u32 apos = 0, bool first = false, u32 wrapP = 0, u32 wrapLast = 0, bool head = false,
                                        u32 laneStep = 0) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, LANES>;
    // what a lane adds to the LDS address it receives: one 16-byte slot.  8-lane groups: the second group of a DPP row
    // works in slots 8..15 of the chunk row (the profile builder fills them with a copy), so its head lane adds 128 bytes
    // more — a per-lane constant (laneStep) in place of the literal, no instruction more.  With both groups in slots
    // 0..7 they met in the same banks with different letter rows: every second LDS cycle of the 8-lane kernels was a
    // conflict on ragged subjects.
    const u32 step2 = LANES < 16 ? laneStep : 0x00100010u;
    const u32 step1 = LANES < 16 ? laneStep : 16u;
    constexpr u32 kSel = 0x0c0c000cu | ((u32)BYTE << 8);  // letter byte BYTE -> bits 15:8
    constexpr int kPostShift = G::kLetterShift - 8;        // row offset = byte << kLetterShift

    // 32-bit kinds, column-offset form: progressive LDS reads like the wide kernels (SWK_PROG_SCALAR=0: all at the top)
#ifndef SWK_PROG_SCALAR
#define SWK_PROG_SCALAR 1
#endif
    constexpr bool kProgScalar = SWK_PROG_SCALAR != 0;
    // subject letter(s): shift along the group, lane 0 takes the next letter of its subject
    u32 wa[G::NW];
    u32 wb[G::NW];
    if constexpr (A::kPacked && LANES <= 16) {
        // both LDS addresses (< 64 KB) travel in one register: one permute, one DPP move and one add for the pair
        constexpr u32 kSel2 = ((u32)(4 + BYTE) << 24) | 0x000c000cu | ((u32)BYTE << 8);  // B's byte -> 31:24, A's -> 15:8
        const u32 inj = __builtin_amdgcn_perm(lettersB, lettersA, kSel2);
        st.yA = prev_lane<LANES, false>(inj, st.yA, head) + step2;
        if constexpr (!OFFS) lds_read_words2<G::NW, G::kChunkRowBytes>(wa, wb, tile + (st.yA & 0xffffu), tile + (st.yA >> 16));
    } else {
        const u32 injA = __builtin_amdgcn_perm(0u, lettersA, kSel) << kPostShift;
        st.yA = prev_lane<LANES, false>(injA, st.yA, head) + step1;
        if constexpr (A::kPacked || !OFFS || !kProgScalar) lds_read_words<G::NW, G::kChunkRowBytes>(wa, tile + st.yA);
        if constexpr (A::kPacked) {
            const u32 injB = __builtin_amdgcn_perm(0u, lettersB, kSel) << kPostShift;
            st.yB = prev_lane<LANES, false>(injB, st.yB, head) + step1;
            lds_read_words<G::NW, G::kChunkRowBytes>(wb, tile + st.yB);
        }
    }

    if constexpr (OFFS) {
        // Row classes: lane-local row r (class p = r mod P) is kept raised by a further a*p.  F then needs its "- a" only
        // where the class wraps (every P rows, and after the lane's last row: by a*(p+1), back to class 0), E and F of a
        // row share the zero level Zc[Q+p+1], and the running maximum is kept per class (rows r and r+P fold into one
        // max3; StripeState explains the windows Zc / maxv and the step index Q).  What a lane passes on (Hlast) stays
        // in the frame of its last row's class; the profile entry of a row carries s + a*(1 + class - class of the row
        // above) (sw_build_profile_kernel), which makes row 0 consistent.
        constexpr int kLastClass = (R - 1) % P;
        constexpr int Q = A::kWindow ? BYTE : 0;  // step within the quad == which letter byte feeds lane 0
        u32 upH, F;
        if constexpr (MULTI) {
            // the stripe above's bottom row at this column; in the first stripe the IN ring holds the column's zero level as
            // H (the local-alignment boundary row 0's diagonal expects: first_stripe_pair) and the kind's unraised zero
            // pattern as F, which lies below every zero level — "no vertical gap" as well as any other value down there
            (void)first;
            upH = prev_lane<LANES, false>(inH, st.Hlast, head);
            F = prev_lane<LANES, false>(inF, st.Fout, head);
        } else {
            const u32 bH = st.Zc[Q + kLastClass];  // the local-alignment boundary H = 0 as row 0's diagonal expects it
            upH = prev_lane<LANES, false>(bH, st.Hlast, head);
            // any F below the column's zero level is "no vertical gap": bound_ctrl zero fill (pattern 0 is below every
            // zero level of every kind, and the fp16 comparator of the int16 kind orders +0.0 below all its patterns)
            if constexpr (ZFILL) F = prev_lane<LANES, true>(0u, st.Fout, head);
            else F = prev_lane<LANES, false>(st.Zc[Q], st.Fout, head);
        }
        u32 diag = st.upH_prev;
        st.upH_prev = upH;
        // the running maxima live in moving frames (st.maxv): the rows' maxima fold straight into the accumulator of
        // their frame (converted back to true scores once per stripe)
        u32 m[P];
#pragma unroll
        for (int c = 0; c < P; c++) m[c] = A::kWindow ? st.maxv[Q + c] : A::add(st.maxv[c], apos);
        // The rows are software-pipelined by hand: the chain h -> h+G -> max3 -> next row's h is serial, and on
        // gfx950 a packed op that reads the result of the instruction right before it costs a wait state, so the
        // independent work (score lookup and diagonal term of the rows ahead, the E update, the maximum) is
        // written in between the links of the chain.
        auto score = [&](int r) -> u32 {
            if constexpr (G::kWide) return 0u;
            else if constexpr (A::kPacked) return __builtin_amdgcn_perm(wb[r >> 1], wa[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u);
            else return wa[r];
        };
        // diagonal term of a row: d + score (the wide kernels pair and add in one instruction, see tq below)
        auto diag_term = [&](int, u32 d, u32 sc) -> u32 { return A::add(d, sc); };
        // Wide words: the diagonal terms run kAhead rows ahead of the chain and the scheduler may only reorder within
        // four rows (it would otherwise compute all of them first, in register order, and so wait for the LAST LDS
        // chunk at the top of the step): a row needs its score words only when the chain is kAhead rows away.
#ifndef SWK_LOOKAHEAD
#define SWK_LOOKAHEAD 12
#endif
        constexpr int kAhead = G::kWide ? (SWK_LOOKAHEAD < R ? SWK_LOOKAHEAD : R) : 1;
        // ... and the LDS reads themselves are issued progressively: the chunks the first kAhead + 4 rows need at the top
        // of the step, chunk k + kChunks0 when the chain reaches row 4k — a few chunks are in flight instead of all
        // (2R registers), which is what lets R go up to 48
        using CH = Chunks<G::NW>;
        constexpr int kChunksAll = CH::kCount;
        // with full chunks: the chunks of the first kAhead + 4 rows at the top (a read is issued at least four rows before
        // its first word is used), as many more as it takes where 3-word chunks shift the rows
        constexpr int kChunks0 = G::kWide ? CH::lead_for(kAhead, 4) : 1;
        static_assert(!G::kWide || kChunks0 >= kChunksAll || CH::lead_ok(kChunks0, kAhead, 4), "a score word would be used before its read is issued");
        const unsigned char* const pa = tile + (st.yA & 0xffffu);
        const unsigned char* const pb = tile + (st.yA >> 16);
        u32 tq[G::kWide ? R : 1];
        if constexpr (G::kWide) {
#pragma unroll
            for (int k = 0; k < kChunks0 && k < kChunksAll; k++) lds_read_chunk2<G::NW, G::kChunkRowBytes>(wa, wb, pa, pb, k);
#pragma unroll
            for (int r = 0; r < kAhead; r++) tq[r] = A::add_pair(wa[r], wb[r], r == 0 ? diag : st.H[r - 1]);
        }
        constexpr bool kProg1 = !A::kPacked && kProgScalar;
        // the score of row r + 2 is picked up when the chain is at row r: rows 0..7 at the top of the step when the chunks
        // are full ones, the chunk after those when the chain reaches a chunk's first row
        constexpr int kChunks0S = kProg1 ? CH::lead_for(2, 6) : 1;
        if constexpr (kProg1) {
#pragma unroll
            for (int k = 0; k < kChunks0S && k < kChunksAll; k++) lds_read_chunk1<G::NW, G::kChunkRowBytes>(wa, tile + st.yA, k);
        }
        u32 s_next = score(0);
        u32 t_next = G::kWide ? 0u : diag_term(0, diag, s_next);
        s_next = R > 1 ? score(1) : 0u;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int c = r % P;
            const u32 zop = st.Zc[Q + c + 1];
            if constexpr (G::kWide) {
                if (r % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                if (CH::starts_chunk(r) && CH::chunk_of(r) + kChunks0 < kChunksAll)
                    lds_read_chunk2<G::NW, G::kChunkRowBytes>(wa, wb, pa, pb, CH::chunk_of(r) + kChunks0);
                t_next = tq[r];
            }
            if constexpr (kProg1) {
                if (r % 4 == 0) __builtin_amdgcn_sched_barrier(0);
                if (CH::starts_chunk(r) && CH::chunk_of(r) + kChunks0S < kChunksAll)
                    lds_read_chunk1<G::NW, G::kChunkRowBytes>(wa, tile + st.yA, CH::chunk_of(r) + kChunks0S);
            }
            const u32 t = t_next;
            const u32 h = A::cell_h(t, st.E[r], F);
            const u32 s1 = s_next;
            if (r + 2 < R) s_next = score(r + 2);
            const u32 hg = A::gap(h, gop);  // gop + a
            if constexpr (G::kWide) {
                if (r + kAhead < R) tq[r + kAhead] = A::add_pair(wa[r + kAhead], wb[r + kAhead], st.H[r + kAhead - 1]);
            } else {
                if (r + 1 < R) t_next = diag_term(r + 1, st.H[r], s1);  // the row's old H is the next row's diagonal
            }
            const u32 fm = A::max3(F, hg, zop);
            st.E[r] = A::max3(st.E[r], hg, zop);
            if (c == P - 1) F = A::gap(fm, wrapP);          // class P-1 -> class 0: lower by a*P
            else if (r == R - 1) F = A::gap(fm, wrapLast);  // the lane's last row: back to class 0 for the next lane
            else F = fm;                                     // next class: frame rises by a while F decays by a
            if ((r / P) & 1) m[c] = A::fold2(m[c], st.H[r - P], h);
            else if (r + P >= R) m[c] = A::max2(m[c], h);
            st.H[r] = h;
        }
#pragma unroll
        for (int c = 0; c < P; c++) st.maxv[Q + c] = m[c];
        if constexpr (!A::kWindow) {
#pragma unroll
            for (int k = (kLastClass == 0 ? 0 : 1); k <= P; k++) st.Zc[k] = A::add(st.Zc[k], apos);
        }
        st.Hlast = st.H[R - 1];
        st.Fout = F;
        return;
    }

    // row above the lane's first row: from the neighbouring lane, or from the stripe border
    u32 upH, F;
    if constexpr (MULTI) {
        upH = prev_lane<LANES, false>(inH, st.Hlast, head);  // lane 0 keeps `old` == the border value
        F = prev_lane<LANES, false>(inF, st.Fout, head);
    } else if constexpr (A::kZero == 0u) {
        upH = prev_lane<LANES, true>(0u, st.Hlast, head);  // bound_ctrl zero fill == the local-alignment boundary
        F = prev_lane<LANES, true>(0u, st.Fout, head);
    } else {
        upH = prev_lane<LANES, false>(A::kZero, st.Hlast, head);
        F = prev_lane<LANES, false>(A::kZero, st.Fout, head);
    }
    u32 diag = st.upH_prev;
    st.upH_prev = upH;

    u32 maxv = st.maxv[0];
#pragma unroll
    for (int r = 0; r < R; r++) {
        u32 t;
        if constexpr (G::kWide) {
            t = A::add_pair(wa[r], wb[r], diag);
        } else if constexpr (A::kPacked) {
            // (score of subject A, score of subject B) for query row r
            t = A::add(diag, __builtin_amdgcn_perm(wb[r >> 1], wa[r >> 1], (r & 1) ? 0x07060302u : 0x05040100u));
        } else {
            t = A::add(diag, wa[r]);
        }
        diag = st.H[r];
        const u32 h = A::cell_h(t, st.E[r], F);
        const u32 hg = A::gap(h, gop);
        st.E[r] = A::gap_state(A::gap(st.E[r], gex), hg);
        F = A::gap_state(A::gap(F, gex), hg);
        st.H[r] = h;
        // running maximum: folded two rows at a time (a 3-input max where the ISA has one)
        if (r & 1) maxv = A::fold2(maxv, st.H[r - 1], h);
        else if (r == R - 1) maxv = A::max2(maxv, h);
    }
    st.maxv[0] = maxv;
    st.Hlast = st.H[R - 1];
    st.Fout = F;
}

// Copy one profile tile (global, L2-resident) into LDS.  The tile starts 16 bytes into the LDS
// array because lane addresses carry a +16 bias (lane 0 injects `offset`, lanes >0 add 16 per hop).
template <int TILE_BYTES>
__device__ __forceinline__ void load_tile(unsigned char* lds, const unsigned char* gsrc) {
    static_assert(TILE_BYTES % 16 == 0, "tile must be 16-byte granular");
    const uint4* src = reinterpret_cast<const uint4*>(gsrc);
    uint4* dst = reinterpret_cast<uint4*>(lds + 16);
    for (int i = threadIdx.x; i < TILE_BYTES / 16; i += kThreads) dst[i] = src[i];
}

// maximum over the 8 lanes of a half row, left in lanes 0..3 (and 8..11): half-row mirror, then the two quad swaps
template <class MAX>
__device__ __forceinline__ u32 half_row_max(u32 v, MAX&& mx) {
    v = mx(v, dpp<0x141, false>(v, v));  // row_half_mirror: lane i <-> 7 - i
    v = mx(v, dpp<0xB1, false>(v, v));   // quad_perm [1,0,3,2]
    v = mx(v, dpp<0x4E, false>(v, v));   // quad_perm [2,3,0,1]
    return v;
}

// ... and over the 4 lanes of a quad, left in all of them
template <class MAX>
__device__ __forceinline__ u32 quad_max(u32 v, MAX&& mx) {
    v = mx(v, dpp<0xB1, false>(v, v));   // quad_perm [1,0,3,2]
    v = mx(v, dpp<0x4E, false>(v, v));   // quad_perm [2,3,0,1]
    return v;
}

template <int KIND, int LANES>
__device__ __forceinline__ u32 group_max(u32 v) {
    using A = Arith<KIND>;
    if constexpr (LANES == 8) return half_row_max(v, [](u32 a, u32 b) { return A::max2(a, b); });
    if constexpr (LANES == 4) return quad_max(v, [](u32 a, u32 b) { return A::max2(a, b); });
    v = A::max2(v, dpp<0x128, false>(v, v));  // row_ror:8
    v = A::max2(v, dpp<0x124, false>(v, v));  // row_ror:4
    v = A::max2(v, dpp<0x122, false>(v, v));  // row_ror:2
    v = A::max2(v, dpp<0x121, false>(v, v));  // row_ror:1
    if constexpr (LANES == 64) {
        v = A::max2(v, (u32)__shfl_xor((int)v, 16));
        v = A::max2(v, (u32)__shfl_xor((int)v, 32));
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// The scan kernel.  Persistent workgroups stride over batches of groups (LANES = 16: 16 groups, 32 or 16
// subjects per batch; LANES = 64: 4 groups), longest subjects first.  MULTI == the query needs more than
// one stripe.
// ------------------------------------------------------------------------------------------------
// Stripe border of multi-stripe queries (the analogue of the reference's devTempHcol2 / devTempEcol2,
// half2_kernels.cuh:335-338).  The bottom row of a stripe — one (H, F) pair = 8 bytes per subject column — goes from the
// group's LAST lane to lane 0 of the same group one stripe later.  Round 4: through LDS rings and block transfers.
//   * the last lane stores its pair into the group's OUT ring in LDS every step (one ds_write_b64; the other lanes hit
//     the group's dummy slot — stores of several lanes to ONE address do not serialise), lane 0 takes the previous stripe's
//     pair from the group's IN ring (one ds_read_b64; the other lanes read the same address and ignore the value): no
//     global access, no staging registers and no register copies in the step.  Round 6: the groups' rings are 16 bytes off
//     the 256-byte bank period and the dummy slots are per group, not per lane — the two groups a 64-bit LDS access serves
//     in one pass no longer meet in the same banks (SQ_LDS_BANK_CONFLICT of the R = 32 kernel: 2.38e9 -> 0.16e9 cycles per
//     launch of 2.8e10 LDS cycles; what is left are the steps on which a group's walking store meets its neighbour's dummy
//     slot);
//   * every kBlockCols = 2 * LANES steps the group moves one BLOCK: each lane copies 16 bytes of the OUT ring to the
//     global scratch (a fully coalesced 16 * LANES-byte burst: whole lines, written once) and 16 bytes of the block after
//     next from the scratch into a register, which it drops into the IN ring one block later.
// (Rounds 1-3: every lane issued 2 x 16-byte loads and stores per four steps, only lane 0's / the last lane's addresses
// walked the real array, the rest hit junk slots and the zeros array: 16 staging registers, ~3 register copies per step,
// and write-backs of junk lines.)  The scratch is indexed by the PRODUCER's step ("position"): the last lane emits column
// c at position c + LANES - 1, so the consumer's block b starts 8 * (LANES - 1) bytes into producer block b.
#ifndef SWK_BORDER_PEND
#define SWK_BORDER_PEND 1   // 1: the block after next is loaded one block ahead into 4 registers; 0: loaded where it is needed (exposed latency once per block)
#endif
template <int LANES>
struct Border {
    static constexpr int kBlockCols = 2 * LANES;                 // positions per block: 16 bytes per lane
    static constexpr int kQuadsPerBlock = LANES / 2;
    static constexpr int kBlockBytes = 8 * kBlockCols;
    static constexpr int kBlockWords = 2 * kBlockCols;
    // LDS per group: IN ring and OUT ring, one block each; per workgroup one dummy slot per group (what they hold is never read)
    static constexpr int kInBytes = kBlockBytes, kOutBytes = kBlockBytes, kDummyBytes = 8 * (kThreads / LANES) + 32;   // one 8-byte dummy slot per group
    // + 16: a ds_read_b64 / ds_write_b64 of a wave is served 32 lanes at a time, i.e. two 16-lane groups per pass, and the
    // groups walk their rings in step — with rings 2^k bytes apart both groups' pairs sat in the same two banks on every step
    // (rounds 4-5: SQ_LDS_BANK_CONFLICT = 6.7 % of the multi-stripe kernels' LDS cycles, 0 in the single-stripe ones)
    static constexpr int kGroupBytes = kInBytes + kOutBytes + 16;
    static constexpr int ring_bytes(int groups) { return groups * kGroupBytes + kDummyBytes; }
    static_assert(kGroupBytes % 16 == 0, "rings are read and written 16 bytes at a time");
    // blocks of the global array for subjects of up to `steps` steps: the partial last block, the tail block behind it,
    // and one more that the consumer's look-ahead reads
    static constexpr int blocks(int steps) { return (steps + kBlockCols - 1) / kBlockCols + 3; }
};
template <int LANES>
constexpr int border_region_words(int lcap) { return Border<LANES>::blocks(lcap) * Border<LANES>::kBlockWords; }  // per group

// Minimum waves per SIMD the register allocator must leave room for (2nd __launch_bounds__ argument).
// Packed kinds: 1 (unconstrained) is best — forcing 3-4 waves spills the multi-stripe kernels (-3..4 %).
// 32-bit kinds: v_add_f32/v_add_u32 co-issue with v_max3_* mostly ACROSS waves (tools/ubench/mix_rate.hip:
// 99 lanes/clk/CU at 4 waves/SIMD, 80 at 2), so they want occupancy more than registers.
#ifndef SWK_MIN_WAVES_SCALAR
#define SWK_MIN_WAVES_SCALAR 0
#endif
#ifndef SWK_I32_WAVES3_MAX_R
#define SWK_I32_WAVES3_MAX_R 32
#endif
#ifndef SWK_I32_WAVES3_MAX_R_MULTI
#define SWK_I32_WAVES3_MAX_R_MULTI 32
#endif
#ifndef SWK_F32_WAVES3_MAX_R
#define SWK_F32_WAVES3_MAX_R 48
#endif
#ifndef SWK_F32_WAVES3_MAX_R_MULTI
#define SWK_F32_WAVES3_MAX_R_MULTI 48
#endif
constexpr int min_waves_of(int KIND, int R, int LANES, bool MULTI) {
    const bool packed = KIND == F16X2 || KIND == I16X2;
    // packed kinds: 2 waves/SIMD (256 VGPRs) for the tall kernels; up to SWK_WAVES3_MAX_R rows a third wave is asked for
    // (168 VGPRs): two waves cover each other's wait states only ~92 % of the time, three reach the issue peak
    if (packed && LANES <= 16 && MULTI && R <= SWK_WAVES3_MAX_R_MULTI) return 3;
    if (packed) return (LANES <= 16 && !MULTI && R <= SWK_WAVES4_MAX_R) ? 4 : (LANES <= 16 && !MULTI && R <= SWK_WAVES3_MAX_R) ? 3 : 2;
    if (SWK_MIN_WAVES_SCALAR > 0) return SWK_MIN_WAVES_SCALAR;
    // int32 above 32 rows per lane: two waves per SIMD (the registers of the taller stripes; its add/max3 mix cannot
    // co-issue anyway).  Up to 32 rows the third wave is worth more than the spills it causes in the multi-stripe kernels
    // from R = 24 up (two-stripe queries of 850 / 1000 residues: 6.36 / 6.42 with three waves, 6.12 / 6.22 TCUPS with two)
    if (KIND == I32 && LANES <= 16 && R > (MULTI ? SWK_I32_WAVES3_MAX_R_MULTI : SWK_I32_WAVES3_MAX_R)) return 2;
    if (KIND == F32 && LANES <= 16 && R > (MULTI ? SWK_F32_WAVES3_MAX_R_MULTI : SWK_F32_WAVES3_MAX_R)) return 2;
    // 4 would spill the multi-stripe R = 14..16 kernels; the wave-wide shape's 43 KB tiles cap it at 3 anyway
    return (R <= 16 && !MULTI && LANES <= 16) ? 4 : 3;
}
template <int KIND, int R, int LANES, bool MULTI>
constexpr int min_waves() { return min_waves_of(KIND, R, LANES, MULTI); }
// the vector registers a wave of that kernel may take: its slot in a SIMD's 512-entry register file (allocated in eights)
constexpr int vgpr_slot_of(int KIND, int R, int LANES, bool MULTI) { return (512 / min_waves_of(KIND, R, LANES, MULTI)) & ~7; }

template <int KIND, int R, int LANES, bool MULTI, bool OFFS = false>
__global__ void __launch_bounds__(kThreads, (min_waves<KIND, R, LANES, MULTI>())) sw_scan_kernel(const ScanParams p) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, LANES>;
    constexpr int kGroups = kThreads / LANES;
    using BD = Border<LANES>;
    constexpr int SHL1 = Shift<LANES>::kShl1;
    constexpr int kQuadsPerLetterBlock = LANES;  // a lane holds 4 letters: LANES quads per reload
    constexpr int P = OFFS ? frame_classes(A::kPacked, R, LANES, MULTI) : 1;  // row classes of the column-offset frame
    __shared__ __attribute__((aligned(16))) unsigned char lds[16 + G::kTileBytes];
    __shared__ __attribute__((aligned(16))) unsigned char rings[MULTI ? BD::ring_bytes(kGroups) : 16];

    const int tid = threadIdx.x;
    const int lane = tid & (LANES - 1);  // position in the alignment group
    const int group = tid / LANES;
    const bool head = lane == 0;
    // 8-lane groups: the two groups of a DPP row use the two halves of the row's 16 profile slots (dp_step: laneStep)
    // (4-lane groups likewise: the four groups of a row in slots 0..3, 4..7, 8..11, 12..15)
    const int slot = LANES < 16 ? (tid & 15) : lane;
    const u32 laneStep = (A::kPacked ? 0x00100010u : 16u) * ((LANES < 16 && head) ? u32(slot + 1) : 1u);
    if (p.start_signal && tid == 0) {
        // this workgroup holds its registers and LDS now: whoever the caller ordered behind the signal cannot take them
        if (atomicAdd(p.work_counter + 1, 1u) + 1u == p.start_quorum)
            __hip_atomic_fetch_add(p.start_signal, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // A re-score launch is the tail of its scan: a handful of subjects, each one group's walk, started BEHIND the bulk grid
    // whose older waves win the SIMD's arbitration — measured 0.27 us per step of a wave-wide fp32 group against 0.14 for
    // the same kernel launched ahead of the bulk grid.  Raised priority gives the walk its issue slots.
    if (p.positions) __builtin_amdgcn_s_setprio(2);
    const bool service = p.service != 0;  // uniform
    const int n = (p.count_ptr && !service) ? *p.count_ptr : p.n;  // service: the list's capacity; its length is polled per batch
    constexpr int kSubjPerBatch = kGroups * A::kSubjects;
    const int nbatches = (n + kSubjPerBatch - 1) / kSubjPerBatch;
    if (!service && (int)blockIdx.x >= nbatches) return;  // workgroup-uniform (device-side count of the re-score path)

    if constexpr (!MULTI) {
        load_tile<G::kTileBytes>(lds, p.profile);
        __syncthreads();
    }

    // Stripe border of this group (Border<LANES>): its rings in LDS and its block array in the global scratch
    unsigned char* const ringIn = rings + (MULTI ? group * BD::kGroupBytes : 0);
    unsigned char* const ringOut = ringIn + BD::kInBytes;
    u32* const gBorder = MULTI ? p.scratch + ((size_t)blockIdx.x * kGroups + group) * (size_t)border_region_words<LANES>(p.lcap) : nullptr;

    // OFFS: a lane starts every stripe "at column -lane": zero level a*(LANES - lane), +a per step
    const u32 apos = OFFS ? A::pos_word(p.gex_mag) : 0u;
    const u32 apos4 = OFFS ? A::pos_word(4 * p.gex_mag) : 0u;
    // frame lowering exists in the packed kernels only: a 32-bit frame has room for any subject (a second copy of the
    // loop body would only cost the occupancy-bounded 32-bit kernels registers)
    constexpr bool kLowers = OFFS && A::kPacked;
    const int rq = kLowers ? p.renorm_quads : 0;
    const u32 zstart = OFFS ? A::zero_at(p.gex_mag, LANES - lane) : A::kZero;
    const u32 zbefore = OFFS ? A::zero_at(p.gex_mag, LANES - lane - 1) : A::kZero;  // the column before

    __shared__ int next_batch, batch_avail;
    // an entry of a claimed list: wait (briefly: the producer stores it right after it has counted it) until it is written,
    // then take it; -1: somebody else has it
    auto claim_entry = [&](int i) -> int {
        for (int spin = 0; spin < (1 << 20); spin++) {
            const int v = __hip_atomic_load(p.claim + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == kListTaken) return -1;
            if (v >= 0) return atomicCAS(p.claim + i, v, kListTaken) == v ? v : -1;
            __builtin_amdgcn_s_sleep(2);
        }
        return -1;
    };
    for (;;) {
        // dynamic batch distribution: one atomic per workgroup per batch
        __syncthreads();
        if (tid == 0) {
            int b = (int)atomicAdd(p.work_counter, 1u), avail = n;
            if (service) {
                // the batch is there as soon as its first entry is; the producer's end (done_flag) ends the wait — what the
                // service has not taken by then is the ordinary re-score launch's
                avail = 0;
                for (u32 spin = 0; b < nbatches; spin++) {
                    avail = __hip_atomic_load(p.count_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (avail > b * kSubjPerBatch) break;
                    const u32 done = __hip_atomic_load(p.done_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((int32_t)(done - p.done_value) >= 0 || spin > 600000u) { b = nbatches; break; }  // (the bound, ~2 s: never hang a GPU on a lost flag; what is left goes to the ordinary re-score launch)
                    __builtin_amdgcn_s_sleep(127);
                }
                avail = min(avail, n);
            }
            if (b == nbatches && p.dry_signal) __hip_atomic_store(p.dry_signal, p.dry_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            next_batch = b;
            batch_avail = avail;
        }
        __syncthreads();
        const int b = next_batch;
        if (b >= nbatches) break;
        const int navail = batch_avail;
        const int batch = service ? b : nbatches - 1 - b;  // DB is length-sorted ascending: longest first; a service walks its list as it grows
        const int i0 = batch * kSubjPerBatch + group * A::kSubjects;
        const int i1 = i0 + 1;
        bool valid0 = i0 < navail;
        bool valid1 = A::kPacked && (i1 < navail);
        int pos0 = 0, pos1 = 0, len0 = 0, len1 = 0;
        const int8_t* s0 = p.chars;
        const int8_t* s1 = p.chars;
        if (p.claim) {  // (32-bit kinds only: one subject per group; read by the group's lane 0, broadcast below)
            int v = -1;
            if (valid0 && lane == 0) v = claim_entry(i0);
            if constexpr (LANES == 64) v = __builtin_amdgcn_readfirstlane(v);
            else v = __shfl(v, (tid & 63) & ~(LANES - 1));
            valid0 = v >= 0;
            pos0 = valid0 ? v : 0;
            if (valid0) {
                len0 = p.lengths[pos0];
                s0 = p.chars + (p.offsets[pos0] - p.offsets[0]);
            }
            valid1 = false;
        } else {
            if (valid0) {
                pos0 = p.positions ? p.positions[i0] : p.first_pos + i0;
                len0 = p.lengths[pos0];
                s0 = p.chars + (p.offsets[pos0] - p.offsets[0]);
            }
            if (valid1) {
                pos1 = p.positions ? p.positions[i1] : p.first_pos + i1;
                len1 = p.lengths[pos1];
                s1 = p.chars + (p.offsets[pos1] - p.offsets[0]);
            }
        }
        int lmax = len0 > len1 ? len0 : len1;
        if constexpr (LANES <= 16) {  // the 4 (8) groups of a wave run in lock-step
            if constexpr (LANES <= 4) lmax = max(lmax, __shfl_xor(lmax, 4));
            if constexpr (LANES <= 8) lmax = max(lmax, __shfl_xor(lmax, 8));
            lmax = max(lmax, __shfl_xor(lmax, 16));
            lmax = max(lmax, __shfl_xor(lmax, 32));
        }
        // the last lane finishes column lmax-1 at step lmax+LANES-2
        int nquads = (lmax + LANES - 1 + 3) >> 2;
        // the border arrays hold lcap columns (sized from the caller's max_subject_len): never walk past them,
        // even if a caller under-reports the bound (scores of such subjects are then wrong, memory is not)
        if constexpr (MULTI) nquads = min(nquads, (p.lcap - 4) >> 2);
        // the same in every lane of the wave (lmax was reduced over its groups): saying so turns the quad loop, its
        // letter-reload test and the frame-lowering segments into scalar control flow (the counter otherwise lives in a
        // VGPR, with a compare, an exec update and a masked branch per quad).  Single-stripe kernels: +0.5 %; the
        // multi-stripe kernels schedule worse with it (4 instructions fewer per quad, yet -1 %), so they keep the vector loop
        if constexpr (!MULTI || SWK_MULTI_SCALAR_LOOP) nquads = __builtin_amdgcn_readfirstlane(nquads);
        const int len0pad = (len0 + 3) & ~3, len1pad = (len1 + 3) & ~3;

        u32 maxv = OFFS ? 0u : A::kZero;  // OFFS tracks true scores (unbiased), the plain form the kind's own zero
        for (int stripe = 0; stripe < p.nstripes; stripe++) {
            const bool first = stripe == 0;
            if constexpr (MULTI) {
                __syncthreads();
                load_tile<G::kTileBytes>(lds, p.profile + (size_t)stripe * G::kTileBytes);
                __syncthreads();
            }
            StripeState<KIND, R, P> st;
            {
                // zero levels of the column before the lane's first one, per class (zc[k] = zbefore + a*k)
                u32 zc[P + 5];
                zc[0] = zbefore; zc[1] = zstart;
#pragma unroll
                for (int k = 2; k < P + 5; k++) zc[k] = A::add(zc[k - 1], apos);
#pragma unroll
                for (int r = 0; r < R; r++) { st.H[r] = zc[r % P]; st.E[r] = zc[r % P + 1]; }
                st.upH_prev = zc[(R - 1) % P]; st.Hlast = zc[(R - 1) % P]; st.Fout = zbefore;
#pragma unroll
                for (int k = 0; k < P + 4; k++) st.Zc[k] = zc[k + 1];
                // OFFS: true score -> frame of the column before the first
#pragma unroll
                for (int d = 0; d < P + 3; d++) st.maxv[d] = OFFS ? A::add(maxv, zc[A::kWindow ? d + 1 : d]) : maxv;
            }
            st.yA = ((u32)(kPadLetter * G::kLetterUnits) << G::kLetterShift) + 16u * (u32)(slot + 1);
            st.yB = st.yA;
            if constexpr (A::kPacked && LANES <= 16) st.yA |= st.yA << 16;  // (address for subject B, address for subject A)

            // subject letters: lane l holds letters 4*LANES*blk + 4l .. +3 of each subject, premultiplied
            // by kLetterUnits so that a byte << kLetterShift is the byte offset of the letter's profile row
            auto fetch = [&](const int8_t* s, int lenpad, int blk) -> u32 {
                const int j = blk * (4 * LANES) + lane * 4;
                u32 w = 0x14141414u;
                if (j < lenpad) w = *reinterpret_cast<const u32*>(s + j);
                return w * (u32)G::kLetterUnits;
            };
            u32 nextA = fetch(s0, len0pad, 0);
            u32 nextB = A::kPacked ? fetch(s1, len1pad, 0) : 0u;
            u32 lettersA = 0, lettersB = 0;

            // Stripe border (MULTI only; Border<LANES>).  Lane 0 walks the IN ring, the last lane the OUT ring; the other
            // lanes read the ring's first pair over and over and write into their dummy slots.
            const bool last = stripe + 1 == p.nstripes;
            const unsigned char* inPtr = ringIn;
            unsigned char* outPtr = (lane == LANES - 1) ? ringOut : rings + kGroups * BD::kGroupBytes + 8 * group;
            const u32 walkIn = 32u;   // bytes per quad; every lane reads along with lane 0 (one address per group: a broadcast)
            const u32 walkOut = (lane == LANES - 1) ? 32u : 0u;
            uint4 pend = make_uint4(A::kZero, A::kZero, A::kZero, A::kZero);   // this lane's 16 bytes of the block after the current one
            uint2 nxt = make_uint2(A::kZero, A::kZero);                          // the pair of the step to come
            // consumer block b = positions b * kBlockCols + (LANES - 1) ...: 8 * (LANES - 1) bytes into producer block b
            const u32* const gIn = MULTI ? gBorder + 2 * (LANES - 1) + 4 * lane : nullptr;
            // What lane 0 takes in the FIRST stripe, where no stripe lies above: columns j, j + 1 -> (H, F, H, F) with H the
            // local-alignment boundary in the column's frame — zero level a * (j mod K + LANES) raised to the class of the lane's
            // last row, the frame row 0's diagonal term expects (OFFS; plain form: the kind's zero) — and F "no vertical gap"
            auto first_stripe_pairs = [&](int j) -> uint4 {
                if constexpr (OFFS) {
                    constexpr int kLastClass = (R - 1) % P;
                    const int K = 4 * rq;   // frame period (0: the frame is never lowered)
                    const int j0 = K > 0 ? (j & (K - 1)) : j, j1 = K > 0 ? ((j + 1) & (K - 1)) : j + 1;
                    return make_uint4(A::zero_at(p.gex_mag, j0 + LANES + kLastClass), A::kZero, A::zero_at(p.gex_mag, j1 + LANES + kLastClass), A::kZero);
                } else {
                    return make_uint4(A::kZero, A::kZero, A::kZero, A::kZero);
                }
            };
            if constexpr (MULTI) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (!first) {
                    const uint4 b0 = *reinterpret_cast<const uint4*>(gIn);
#if SWK_BORDER_PEND
                    pend = *reinterpret_cast<const uint4*>(gIn + BD::kBlockWords);
#endif
                    *reinterpret_cast<uint4*>(ringIn + 16 * lane) = b0;
                } else {
                    *reinterpret_cast<uint4*>(ringIn + 16 * lane) = first_stripe_pairs(2 * lane);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                nxt = *reinterpret_cast<const uint2*>(inPtr);
            }
            // end of block `blk` (all of its steps are done): the OUT ring's first block goes to the scratch, the IN ring
            // takes the next block, the load of the one after that is issued
            auto block_end = [&](int blk) {
                // every address is rebuilt here from the thread index (kept opaque, so that the compiler does not hoist a
                // dozen loop-invariant address registers into a loop that runs at the register limit)
                int t = tid;
                asm volatile("" : "+v"(t));
                const int ln = t & (LANES - 1), grp = t / LANES;
                unsigned char* const rIn = rings + grp * BD::kGroupBytes + 16 * ln;
                u32* const gb = p.scratch + ((size_t)blockIdx.x * kGroups + grp) * (size_t)border_region_words<LANES>(p.lcap) + 4 * ln;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (!last) {
                    const uint4 v = *reinterpret_cast<const uint4*>(rIn + BD::kInBytes);
                    *reinterpret_cast<uint4*>(gb + (size_t)blk * BD::kBlockWords) = v;
                }
                outPtr -= walkOut * BD::kQuadsPerBlock;
                inPtr -= walkIn * BD::kQuadsPerBlock;
                if (first) {
                    *reinterpret_cast<uint4*>(rIn) = first_stripe_pairs((blk + 1) * BD::kBlockCols + 2 * ln);
                } else {
#if SWK_BORDER_PEND
                    *reinterpret_cast<uint4*>(rIn) = pend;
                    pend = *reinterpret_cast<const uint4*>(gb + 2 * (LANES - 1) + (size_t)(blk + 2) * BD::kBlockWords);
#else
                    *reinterpret_cast<uint4*>(rIn) = *reinterpret_cast<const uint4*>(gb + 2 * (LANES - 1) + (size_t)(blk + 1) * BD::kBlockWords);
#endif
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                nxt = *reinterpret_cast<const uint2*>(inPtr);   // what the last step prefetched was the old block's
            };

            auto quad = [&](int q, auto lower_tag) {
                constexpr bool LOWER = decltype(lower_tag)::value;
                if ((q & (kQuadsPerLetterBlock - 1)) == 0) {
                    lettersA = nextA; lettersB = nextB;
                    nextA = fetch(s0, len0pad, q / kQuadsPerLetterBlock + 1);
                    if constexpr (A::kPacked) nextB = fetch(s1, len1pad, q / kQuadsPerLetterBlock + 1);
                }
                // OFFS on long subjects: the frame of column j is a*(j mod K + LANES), i.e. a lane lowers everything it
                // holds by a*K right before it enters a column that is a multiple of K (lane l at step m*K + l).  The
                // values it receives for that column come from a lane that has already done so, the ones it passed
                // on last belong to the column before: the frame stays a function of the column alone, which is what
                // keeps the stripe borders consistent.  16 of K steps pay 2R + 4 extra instructions.
                const int lower_lane = 4 * (q & (rq - 1));
                auto lower_frame = [&](int k) {
                    const u32 gw = lane == k ? p.renorm_word : 0u;
#pragma unroll
                    for (int r = 0; r < R; r++) { st.H[r] = A::gap(st.H[r], gw); st.E[r] = A::gap(st.E[r], gw); }
                    st.upH_prev = A::gap(st.upH_prev, gw);
#pragma unroll
                    for (int k = 0; k < P + 4; k++) st.Zc[k] = A::gap(st.Zc[k], gw);
#pragma unroll
                    for (int d = 0; d < P + 3; d++) st.maxv[d] = A::gap(st.maxv[d], gw);
                };
                // one step: lane 0's pair of the stripe above from the IN ring, the last lane's own pair into the OUT ring.
                // The pair is requested ONE STEP AHEAD (nxt): the chain of a step starts with it (row 0's diagonal and F),
                // and an LDS read issued at the top of the step would be waited for right there.
                auto border_step = [&](auto byte_tag) {
                    constexpr int BYTE = decltype(byte_tag)::value;
                    const uint2 in = nxt;
                    if constexpr (MULTI) {
                        if constexpr (BYTE == 3) {
                            inPtr += walkIn;
                            nxt = *reinterpret_cast<const uint2*>(inPtr);   // first pair of the next quad (re-read after a block transfer)
                        } else {
                            nxt = *reinterpret_cast<const uint2*>(inPtr + 8 * (BYTE + 1));
                        }
                    }
                    dp_step<KIND, R, LANES, BYTE, MULTI, OFFS, P>(st, lds, lettersA, lettersB, p.gop, p.gex, in.x, in.y, apos, first, p.wrap_class, p.wrap_last, head, laneStep);
                    if constexpr (MULTI) *reinterpret_cast<uint2*>(outPtr + 8 * BYTE) = make_uint2(st.Hlast, st.Fout);
                };
                if constexpr (LOWER) lower_frame(lower_lane + 0);
                border_step(std::integral_constant<int, 0>{});
                if constexpr (LOWER) lower_frame(lower_lane + 1);
                border_step(std::integral_constant<int, 1>{});
                if constexpr (LOWER) lower_frame(lower_lane + 2);
                border_step(std::integral_constant<int, 2>{});
                if constexpr (LOWER) lower_frame(lower_lane + 3);
                border_step(std::integral_constant<int, 3>{});
                if constexpr (MULTI) outPtr += walkOut;
                lettersA = dpp<SHL1, true>(0u, lettersA);
                if constexpr (A::kPacked) lettersB = dpp<SHL1, true>(0u, lettersB);
                if constexpr (OFFS && A::kWindow) {  // the windows move on by four columns
#pragma unroll
                    for (int k = 0; k < P + 4; k++) st.Zc[k] = A::add(st.Zc[k], apos4);
#pragma unroll
                    for (int d = 0; d < P + 3; d++) st.maxv[d] = A::add(st.maxv[d], apos4);
                }
            };
            // the quads in which lanes lower their frame (the first LANES/4 of every K/4 quads but the first) run a second
            // copy of the loop body, so that the others pay nothing for it
            if constexpr (!MULTI) {
                const int seg = (kLowers && rq > 0) ? rq : nquads;
                for (int q0 = 0; q0 < nquads; q0 += seg) {
                    const int qend = min(nquads, q0 + seg);
                    int q = q0;
                    if constexpr (kLowers) {
                        if (q0 > 0) {
                            const int qlow = min(qend, q0 + LANES / 4);
                            for (; q < qlow; q++) quad(q, std::true_type{});
                        }
                    }
                    for (; q < qend; q++) quad(q, std::false_type{});
                }
            } else {
                // multi-stripe: block by block (Border<LANES>: kQuadsPerBlock quads, then the block transfer); the lowering
                // period K/4 is a multiple of the block and the LANES/4 lowering quads fit in one
                static_assert(BD::kQuadsPerBlock >= LANES / 4, "the lowering quads of a period must lie in one block");
                for (int q0 = 0; q0 < nquads; q0 += BD::kQuadsPerBlock) {
                    const int qend = min(nquads, q0 + BD::kQuadsPerBlock);
                    int q = q0;
                    if constexpr (kLowers) {
                        if (rq > 0 && q0 > 0 && (q0 & (rq - 1)) == 0) {
                            const int qlow = min(qend, q0 + LANES / 4);
                            for (; q < qlow; q++) quad(q, std::true_type{});
                        }
                    }
                    for (; q < qend; q++) quad(q, std::false_type{});
                    if (qend == q0 + BD::kQuadsPerBlock) block_end(q0 / BD::kQuadsPerBlock);
                }
            }
            if constexpr (MULTI) {
                // The last lane has emitted positions 0 .. 4*nquads - 1; the next stripe's lane 0 reads positions up to
                // 4*nquads + LANES - 2 (its columns up to 4*nquads - 1): LANES pairs of "no value" (the kind's unraised zero
                // pattern) go behind the last emitted one — into the OUT ring where they fall into the partial block, straight
                // to the scratch where they fall into the block behind it — and the partial block leaves the OUT ring.
                if (!last) {
                    const int done = nquads & ~(BD::kQuadsPerBlock - 1);           // quads in complete (flushed) blocks
                    const int slot = 4 * (nquads - done) + lane;                     // this lane's "no value" pair: position 4 * done + slot
                    u32* const g = gBorder + (size_t)(done / BD::kQuadsPerBlock) * BD::kBlockWords;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (slot < BD::kBlockCols) *reinterpret_cast<uint2*>(ringOut + 8 * slot) = make_uint2(A::kZero, A::kZero);
                    else *reinterpret_cast<uint2*>(g + 2 * slot) = make_uint2(A::kZero, A::kZero);   // in the block behind: the flush below does not touch it
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    *reinterpret_cast<uint4*>(g + 4 * lane) = *reinterpret_cast<const uint4*>(ringOut + 16 * lane);
                }
            }
            if constexpr (OFFS) {
                if constexpr (A::kWindow) {
                    // accumulator d and zero level d are in the same frame -> true scores
                    maxv = A::true_of(st.maxv[0], st.Zc[0]);
#pragma unroll
                    for (int d = 1; d < P + 3; d++) maxv = A::true_max(maxv, A::true_of(st.maxv[d], st.Zc[d]));
                } else {
                    // Zc[c + 1] is the NEXT column's level of class c + 1: 2a above the accumulator's frame
                    maxv = A::true_of(st.maxv[0], A::gap(A::gap(st.Zc[1], p.gex), p.gex));
#pragma unroll
                    for (int c = 1; c < P; c++)
                        maxv = A::true_max(maxv, A::true_of(st.maxv[c], A::gap(A::gap(st.Zc[c + 1], p.gex), p.gex)));
                }
            } else {
                maxv = st.maxv[0];
            }
            if constexpr (MULTI) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        }

        int sc0, sc1, guard = 0;
        if constexpr (OFFS) {
            // group maximum of true scores
            if constexpr (LANES == 8) {
                maxv = half_row_max(maxv, [](u32 a, u32 b) { return A::true_max(a, b); });
            } else if constexpr (LANES == 4) {
                maxv = quad_max(maxv, [](u32 a, u32 b) { return A::true_max(a, b); });
            } else {
                maxv = A::true_max(maxv, dpp<0x128, false>(maxv, maxv));
                maxv = A::true_max(maxv, dpp<0x124, false>(maxv, maxv));
                maxv = A::true_max(maxv, dpp<0x122, false>(maxv, maxv));
                maxv = A::true_max(maxv, dpp<0x121, false>(maxv, maxv));
            }
            if constexpr (LANES == 64) {
                maxv = A::true_max(maxv, (u32)__shfl_xor((int)maxv, 16));
                maxv = A::true_max(maxv, (u32)__shfl_xor((int)maxv, 32));
            }
            sc0 = A::true_lo(maxv);
            sc1 = A::true_hi(maxv);
            // no value of the alignment exceeded score + a * (columns + LANES): inside the exact range below the limit
            guard = p.gex_mag * ((rq > 0 && 4 * nquads > 4 * rq ? 4 * rq : 4 * nquads) + 2 * LANES + 4 + P);
        } else {
            maxv = group_max<KIND, LANES>(maxv);
            sc0 = A::score_lo(maxv);
            sc1 = A::score_hi(maxv);
        }
        if (lane == 0) {
            if (valid0) {
                if (A::kPacked && p.ovf_check && sc0 >= A::kLimit - guard) {
                    __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), pos0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    p.scores[pos0] = (float)sc0;
                    if (p.stat_count && sc0 >= p.stat_limit) atomicAdd(p.stat_count, 1);
                }
                p.ids[pos0] = (int32_t)(p.id_offset + pos0);
            }
            if (valid1) {
                if (p.ovf_check && sc1 >= A::kLimit - guard) {
                    __hip_atomic_store(p.ovf_pos + atomicAdd(p.ovf_count, 1), pos1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    p.scores[pos1] = (float)sc1;
                }
                p.ids[pos1] = (int32_t)(p.id_offset + pos1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Profile builder: tile[stripe][letter][chunk][lane][4 words] from the encoded query and the matrix.
// Replaces the per-block pair-table construction of the reference (half2_kernels.cuh:57-65).
// ------------------------------------------------------------------------------------------------
template <int KIND, int R, int LANES>
__global__ void sw_build_profile_kernel(const int8_t* __restrict__ query, int32_t qlen,
                                        const int8_t* __restrict__ matrix21, int32_t pad_row, int32_t nstripes,
                                        unsigned char* __restrict__ profile, int32_t shift) {
    using A = Arith<KIND>;
    using G = Geometry<KIND, R, LANES>;
    constexpr int kWordsPerRow = G::kRowBytes / 4;
    constexpr int kWordsPerChunkRow = G::kChunkRowBytes / 4;
    const size_t total = (size_t)nstripes * kLetters * kWordsPerRow;
    u32* out = reinterpret_cast<u32*>(profile);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int word = (int)(i % kWordsPerRow);
        const int letter = (int)((i / kWordsPerRow) % kLetters);
        const int stripe = (int)(i / ((size_t)kWordsPerRow * kLetters));
        const int chunk = word / kWordsPerChunkRow, lane = ((word % kWordsPerChunkRow) / 4) % LANES, sub = word % 4;
        const int w = G::C::first(chunk) + sub;  // word index within the lane's NW words (Chunks: 4- and 3-word chunks)
        u32 v = 0;
        if (sub < G::C::words(chunk) && w < G::NW) {
            auto entry = [&](int row_in_lane) -> u32 {
                if (row_in_lane >= R) return 0u;  // unused upper half of an odd R's last word
                const int64_t row = (int64_t)stripe * G::kStripeRows + lane * R + row_in_lane;
                // matrix21: (query letters + one padding row `pad_row`) x 21 subject letters
                const int qc = row < qlen ? (int)query[row] : pad_row;
                // OFFS kernels (shift = a): the diagonal step raises the frame by a per column and by a per row class;
                // the row above lane-local row 0 is the previous lane's last row (dp_step<OFFS>)
                const int P = frame_classes(A::kPacked, R, LANES, nstripes > 1);  // the scan kernel's (MULTI == more than one stripe)
                const int cls = row_in_lane % P, above = (row_in_lane == 0 ? R - 1 : row_in_lane - 1) % P;
                return A::encode_score((int)matrix21[qc * kLetters + letter] + shift * (1 + cls - above));
            };
            if constexpr (G::kWide) v = w < R ? (entry(w) | (A::kOne << 16)) : 0u;
            else if constexpr (A::kPacked) v = entry(2 * w) | (entry(2 * w + 1) << 16);
            else v = entry(w);
        }
        out[i] = v;
    }
}

}  // namespace swk
