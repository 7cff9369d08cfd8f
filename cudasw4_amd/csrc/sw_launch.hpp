// sw_launch.hpp — per-kind launch tables shared by the kind translation units and sw_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "sw_dp_kernel.hpp"

namespace swk {

// Rows-per-lane values that are compiled.  The query planner (sw_api.hip: plan_query) only picks these.
constexpr int kRowsGranule = 2;
constexpr int kMaxRowsPacked = 32;  // stripe = 512 query rows
constexpr int kMaxRowsScalar = 16;  // stripe = 256 query rows (32-bit profile entries)

struct KindLaunch {
    // returns hipSuccess or the launch error; R must be a compiled value, else hipErrorInvalidValue
    hipError_t (*scan)(int R, bool multi, int grid, hipStream_t stream, const ScanParams& p);
    hipError_t (*profile)(int R, const int8_t* query, int32_t qlen, const int8_t* matrix21, int32_t nstripes,
                          unsigned char* out, hipStream_t stream);
    size_t (*tile_bytes)(int R);
    int max_rows;
};

const KindLaunch& launch_f16x2();
const KindLaunch& launch_i16x2();
const KindLaunch& launch_i32();
const KindLaunch& launch_f32();

// ---- helpers used by the kind TUs ----
template <int KIND, int R>
hipError_t launch_scan_r(bool multi, int grid, hipStream_t stream, const ScanParams& p) {
    // a query that needs more than one stripe always gets R > max/2 from the planner
    constexpr int kMaxR = Arith<KIND>::kPacked ? kMaxRowsPacked : kMaxRowsScalar;
    if (multi) {
        if constexpr (2 * R > kMaxR) hipLaunchKernelGGL((sw_scan_kernel<KIND, R, true>), dim3(grid), dim3(kThreads), 0, stream, p);
        else return hipErrorInvalidValue;
    } else {
        hipLaunchKernelGGL((sw_scan_kernel<KIND, R, false>), dim3(grid), dim3(kThreads), 0, stream, p);
    }
    return hipGetLastError();
}

template <int KIND, int R>
hipError_t launch_profile_r(const int8_t* query, int32_t qlen, const int8_t* matrix21, int32_t nstripes,
                            unsigned char* out, hipStream_t stream) {
    const int total = nstripes * kLetters * (Geometry<KIND, R>::kRowBytes / 4);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL((sw_build_profile_kernel<KIND, R>), dim3(grid), dim3(256), 0, stream, query, qlen, matrix21,
                       nstripes, out);
    return hipGetLastError();
}

#define SWK_FOR_EACH_R_PACKED(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32)
#define SWK_FOR_EACH_R_SCALAR(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)

#define SWK_DEFINE_KIND(FN, KIND, FOR_EACH_R, MAXR)                                                          \
    static hipError_t FN##_scan(int R, bool multi, int grid, hipStream_t stream, const ScanParams& p) {       \
        switch (R) {                                                                                          \
            FOR_EACH_R(SWK_CASE_SCAN_##KIND)                                                                  \
        }                                                                                                     \
        return hipErrorInvalidValue;                                                                          \
    }                                                                                                         \
    static hipError_t FN##_profile(int R, const int8_t* q, int32_t qlen, const int8_t* m, int32_t ns,         \
                                   unsigned char* out, hipStream_t s) {                                       \
        switch (R) {                                                                                          \
            FOR_EACH_R(SWK_CASE_PROF_##KIND)                                                                  \
        }                                                                                                     \
        return hipErrorInvalidValue;                                                                          \
    }                                                                                                         \
    static size_t FN##_tile_bytes(int R) {                                                                    \
        switch (R) {                                                                                          \
            FOR_EACH_R(SWK_CASE_TILE_##KIND)                                                                  \
        }                                                                                                     \
        return 0;                                                                                             \
    }                                                                                                         \
    const KindLaunch& FN() {                                                                                  \
        static const KindLaunch k{FN##_scan, FN##_profile, FN##_tile_bytes, MAXR};                            \
        return k;                                                                                             \
    }

}  // namespace swk
