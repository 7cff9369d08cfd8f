// sw_stream_i32.hip — instantiations of the stream kernel (sw_stream_kernel.hpp) for kind I32.
#include "sw_stream_launch.hpp"

namespace swk {
#define SWK_CASE_STREAM_I32(R) case R: return launch_stream_r<I32, R>(grid, stream, p);
SWK_DEFINE_STREAM(stream_i32, I32, SWK_FOR_EACH_R_SCALAR)
}  // namespace swk
