// sw_stream_launch.hpp — launch table of the stream kernels of one kind (included by sw_stream_<kind>.hip).
#pragma once
#include "sw_launch.hpp"
#include "sw_stream_kernel.hpp"

namespace swk {

template <int KIND, int R>
hipError_t launch_stream_r(int grid, hipStream_t stream, const ScanParams& p) {
    if constexpr (R > max_rows(Arith<KIND>::kPacked, kGroup)) {
        return hipErrorInvalidValue;
    } else {
        hipLaunchKernelGGL((sw_stream_kernel<KIND, R>), dim3(grid), dim3(kThreads), 0, stream, p);
        return hipGetLastError();
    }
}

#define SWK_DEFINE_STREAM(FN, KIND, FOR_EACH_R)                                             \
    hipError_t FN(int R, int grid, hipStream_t stream, const ScanParams& p) {               \
        switch (R) { FOR_EACH_R(SWK_CASE_STREAM_##KIND) }                                   \
        return hipErrorInvalidValue;                                                        \
    }

}  // namespace swk
