// sw_stream_i16x2.hip — instantiations of the stream kernel (sw_stream_kernel.hpp) for kind I16X2.
#include "sw_stream_launch.hpp"

namespace swk {
#define SWK_CASE_STREAM_I16X2(R) case R: return launch_stream_r<I16X2, R>(grid, stream, p);
SWK_DEFINE_STREAM(stream_i16x2, I16X2, SWK_FOR_EACH_R_PACKED)
}  // namespace swk
