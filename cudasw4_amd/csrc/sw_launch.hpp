// sw_launch.hpp — per-kind launch tables shared by the kind translation units and sw_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>

#include "sw_dp_kernel.hpp"
#include "sw_stream_kernel.hpp"

namespace swk {

// Rows-per-lane values that are compiled.  The query planner (sw_api.hip: plan_query) only picks these.
constexpr int kRowsGranule = 1;
// standard shape (16-lane groups)
#ifndef SWK_MAX_ROWS_PACKED
#define SWK_MAX_ROWS_PACKED 48
#endif
constexpr int kMaxRowsPacked = SWK_MAX_ROWS_PACKED;  // stripe = 768 query rows, 63 KB tile of wide words (two workgroups per CU, LDS addresses stay below 64 KB); 256 VGPRs (2 waves/SIMD)
#ifndef SWK_MAX_ROWS_SCALAR
#define SWK_MAX_ROWS_SCALAR 36
#endif
constexpr int kMaxRowsScalar = SWK_MAX_ROWS_SCALAR;  // fp32: its add/max3 co-issue wants three waves per SIMD: a single stripe up to 576 query rows (47 KB tile, three workgroups per CU; the 567-residue query 7.55 -> 8.71 TCUPS against two stripes of 18 rows), several stripes up to 32 rows per lane (kMaxRowsScalarMulti)
#ifndef SWK_MAX_ROWS_SCALAR_MULTI
#define SWK_MAX_ROWS_SCALAR_MULTI 32
#endif
constexpr int kMaxRowsScalarMulti = SWK_MAX_ROWS_SCALAR_MULTI;
#ifndef SWK_MAX_ROWS_I32
#define SWK_MAX_ROWS_I32 48
#endif
constexpr int kMaxRowsI32 = SWK_MAX_ROWS_I32;
// packed multi-stripe kernels up to this height keep three waves per SIMD (sw_dp_kernel.hpp: SWK_WAVES3_MAX_R_MULTI)
constexpr int kWaves3MaxRowsPackedMulti = SWK_WAVES3_MAX_R_MULTI;  // int32 cannot co-issue, so like the packed kinds it trades the third wave for taller stripes (fewer stripes, less per-step overhead)
// long-subject shape (64-lane groups)
constexpr int kMaxRowsPackedLong = 16;  // stripe = 1024 query rows, 43 KB tile
constexpr int kMaxRowsScalarLong = 8;   // stripe = 512 query rows, 43 KB tile
// short groups (8 and 4 lanes: single-stripe queries of up to 256 / 128 residues): 32 rows per lane
constexpr int kMaxRowsShortGroups = 32;
constexpr int max_rows(int kind, int lanes) {
    const bool packed = kind == F16X2 || kind == I16X2;
    if (lanes < 16) return kMaxRowsShortGroups;
    return lanes == 16 ? (packed ? kMaxRowsPacked : kind == I32 ? kMaxRowsI32 : kMaxRowsScalar) : (packed ? kMaxRowsPackedLong : kMaxRowsScalarLong);
}

struct KindLaunch {
    // return hipSuccess or the launch error; (R, lanes) must be a compiled combination, else hipErrorInvalidValue
    // offs: the column-offset form of the recurrence (sw_dp_kernel.hpp: dp_step<OFFS>); needs a profile built with shift = a
    // reserve > 0: the grid is capped at the workgroups the device holds of this kernel at once, minus `reserve` (slots left
    // free for small launches of other streams while the persistent grid runs: sw_set_grid_reserve)
    hipError_t (*scan)(int R, int lanes, bool multi, bool offs, int grid, int reserve, hipStream_t stream, const ScanParams& p);
    hipError_t (*profile)(int R, int lanes, const int8_t* query, int32_t qlen, const int8_t* matrix21, int32_t pad_row,
                          int32_t nstripes, unsigned char* out, int32_t shift, hipStream_t stream);
    size_t (*tile_bytes)(int R, int lanes);
    // an empty launch from the kind's translation unit: the runtime loads a TU's code object (a few MB of kernels) on
    // the first launch of any of its kernels; sw_ctx_create pays that once instead of the first query (~25 ms per kind)
    hipError_t (*warm)(hipStream_t stream);
    bool packed;
};

const KindLaunch& launch_f16x2();
const KindLaunch& launch_i16x2();
const KindLaunch& launch_i32();
const KindLaunch& launch_f32();

// ---- helpers used by the kind TUs ----
// workgroups of `kernel` the current device holds at once (asked once per instantiation and device); 0: unknown
template <class K>
int resident_workgroups(K kernel) {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kThreads, 0) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu <= 0 || cus <= 0) {
            (void)hipGetLastError();
            v = -1;
        } else {
            v = per_cu * cus;
        }
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? v : 0;
}

template <class K>
hipError_t launch_scan_k(K kernel, int grid, int reserve, hipStream_t stream, const ScanParams& p) {
    if (reserve > 0) {
        const int resident = resident_workgroups(kernel);
        if (resident > 0 && grid > resident - reserve) {
            ScanParams q = p;
            grid = resident - reserve > 1 ? resident - reserve : 1;
            if (q.start_quorum > (u32)grid) q.start_quorum = (u32)grid;
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, stream, q);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, stream, p);
    return hipGetLastError();
}

// Which instantiations exist (round 6: the kernel set was cut to what the planner can reach — 641 kernels and 35 MB of
// code objects per packed kind before):
//   * packed kinds run the column-offset recurrence only: a launch that cannot (gap extension too large for any frame
//     period, no overflow list to flag into) is served by its 32-bit kind, bit-identically (sw_api.hip: packed_fallback);
//   * packed kinds on 16-lane groups: the streamed kernels (sw_stream_kernel.hpp) for every single-stripe R and for the
//     multi-stripe launches whose subjects are short enough for rounds of several slots to pay; sw_scan_kernel for the other
//     multi-stripe launches;
//   * 8- and 4-lane groups: single-stripe kernels only (queries of up to 256 / 128 residues).
template <int KIND, int R, int LANES, bool OFFS>
hipError_t launch_scan_ro(bool multi, int grid, int reserve, hipStream_t stream, const ScanParams& p) {
    // a query that needs more than one stripe always gets R > max/2 from the planner
    constexpr int kMaxR = max_rows(KIND, LANES);
    constexpr bool kPacked = Arith<KIND>::kPacked;
    if constexpr (R > kMaxR || (kPacked && !OFFS)) {
        return hipErrorInvalidValue;
    } else if constexpr (kPacked && LANES == 16) {
        if (p.positions || p.claim || p.service || p.count_ptr || p.stream_slots < 1) return hipErrorInvalidValue;   // (32-bit kinds' business)
        if (multi) {
            if constexpr (2 * R > kMaxR) {
                // several stripes: the streamed kernel where rounds of several slots pay (short subjects: sw_api.hip: stream_plan),
                // the one-pair-at-a-time kernel otherwise — a round of ONE slot costs the streamed kernel its round set-up
                // (claim, slot widths, two barriers) on top of what sw_scan_kernel does
                if (p.stream_slots > 1) return launch_scan_k(sw_scan_stream_kernel<KIND, R, LANES, true>, grid, reserve, stream, p);
                return launch_scan_k(sw_scan_kernel<KIND, R, LANES, true, OFFS>, grid, reserve, stream, p);
            } else {
                return hipErrorInvalidValue;
            }
        } else {
            return launch_scan_k(sw_scan_stream_kernel<KIND, R, LANES, false>, grid, reserve, stream, p);
        }
    } else {
        if (multi) {
            if constexpr (2 * R > kMaxR && LANES >= 16) {   // (short groups: single-stripe queries only)
                return launch_scan_k(sw_scan_kernel<KIND, R, LANES, true, OFFS>, grid, reserve, stream, p);
            } else {
                return hipErrorInvalidValue;
            }
        } else {
            return launch_scan_k(sw_scan_kernel<KIND, R, LANES, false, OFFS>, grid, reserve, stream, p);
        }
    }
}

template <int KIND, int R, int LANES>
hipError_t launch_scan_r(bool multi, bool offs, int grid, int reserve, hipStream_t stream, const ScanParams& p) {
    return offs ? launch_scan_ro<KIND, R, LANES, true>(multi, grid, reserve, stream, p)
                : launch_scan_ro<KIND, R, LANES, false>(multi, grid, reserve, stream, p);
}

template <int KIND, int R, int LANES>
hipError_t launch_profile_r(const int8_t* query, int32_t qlen, const int8_t* matrix21, int32_t pad_row, int32_t nstripes,
                            unsigned char* out, int32_t shift, hipStream_t stream) {
    if constexpr (R > max_rows(KIND, LANES)) {
        return hipErrorInvalidValue;
    } else {
        const size_t total = (size_t)nstripes * kLetters * (Geometry<KIND, R, LANES>::kRowBytes / 4);
        const int grid = (int)std::min<size_t>((total + 255) / 256, 65536);  // grid-stride loop covers the rest
        hipLaunchKernelGGL((sw_build_profile_kernel<KIND, R, LANES>), dim3(grid), dim3(256), 0, stream, query, qlen,
                           matrix21, pad_row, nstripes, out, shift);
        return hipGetLastError();
    }
}

template <int KIND, int R, int LANES>
constexpr size_t tile_bytes_r() {
    if constexpr (R > max_rows(KIND, LANES)) return 0;
    else return (size_t)Geometry<KIND, R, LANES>::kTileBytes;
}

#define SWK_FOR_EACH_R_PACKED(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37) X(38) X(39) X(40) X(41) X(42) X(43) X(44) X(45) X(46) X(47) X(48)
#define SWK_FOR_EACH_R_I32(X) SWK_FOR_EACH_R_PACKED(X)
#define SWK_FOR_EACH_R_SCALAR(X) \
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32)

#define SWK_DEFINE_KIND(FN, KIND, FOR_EACH_R)                                                                      \
    static hipError_t FN##_scan(int R, int lanes, bool multi, bool offs, int grid, int reserve, hipStream_t stream, \
                                const ScanParams& p) {                                                              \
        if (lanes == 16) { switch (R) { FOR_EACH_R(SWK_CASE_SCAN16_##KIND) } }                                      \
        else if (lanes == 64) { switch (R) { FOR_EACH_R(SWK_CASE_SCAN64_##KIND) } }                                 \
        else if (lanes == 8) { switch (R) { FOR_EACH_R(SWK_CASE_SCAN8_##KIND) } }                                   \
        else if (lanes == 4) { switch (R) { FOR_EACH_R(SWK_CASE_SCAN4_##KIND) } }                                   \
        return hipErrorInvalidValue;                                                                                \
    }                                                                                                               \
    static hipError_t FN##_profile(int R, int lanes, const int8_t* q, int32_t qlen, const int8_t* m, int32_t pr,    \
                                   int32_t ns, unsigned char* out, int32_t shift, hipStream_t s) {                  \
        if (lanes == 16) { switch (R) { FOR_EACH_R(SWK_CASE_PROF16_##KIND) } }                                      \
        else if (lanes == 64) { switch (R) { FOR_EACH_R(SWK_CASE_PROF64_##KIND) } }                                 \
        else if (lanes == 8) { switch (R) { FOR_EACH_R(SWK_CASE_PROF8_##KIND) } }                                   \
        else if (lanes == 4) { switch (R) { FOR_EACH_R(SWK_CASE_PROF4_##KIND) } }                                   \
        return hipErrorInvalidValue;                                                                                \
    }                                                                                                               \
    static size_t FN##_tile_bytes(int R, int lanes) {                                                               \
        if (lanes == 16) { switch (R) { FOR_EACH_R(SWK_CASE_TILE16_##KIND) } }                                      \
        else if (lanes == 64) { switch (R) { FOR_EACH_R(SWK_CASE_TILE64_##KIND) } }                                 \
        else if (lanes == 8) { switch (R) { FOR_EACH_R(SWK_CASE_TILE8_##KIND) } }                                   \
        else if (lanes == 4) { switch (R) { FOR_EACH_R(SWK_CASE_TILE4_##KIND) } }                                   \
        return 0;                                                                                                   \
    }                                                                                                               \
    static __global__ void FN##_warm_kernel() {}                                                                    \
    static hipError_t FN##_warm(hipStream_t s) {                                                                    \
        hipLaunchKernelGGL(FN##_warm_kernel, dim3(1), dim3(64), 0, s);                                              \
        return hipGetLastError();                                                                                   \
    }                                                                                                               \
    const KindLaunch& FN() {                                                                                        \
        static const KindLaunch k{FN##_scan, FN##_profile, FN##_tile_bytes, FN##_warm, Arith<KIND>::kPacked};       \
        return k;                                                                                                   \
    }

}  // namespace swk

