// sw_stream_f32.hip — instantiations of the stream kernel (sw_stream_kernel.hpp) for kind F32.
#include "sw_stream_launch.hpp"

namespace swk {
#define SWK_CASE_STREAM_F32(R) case R: return launch_stream_r<F32, R>(grid, stream, p);
SWK_DEFINE_STREAM(stream_f32, F32, SWK_FOR_EACH_R_SCALAR)
}  // namespace swk
