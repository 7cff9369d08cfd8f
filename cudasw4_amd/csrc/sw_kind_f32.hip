// sw_kind_f32.hip — instantiations of the DP scan kernel for kind F32 (one TU per kind so that the
// four kinds compile in parallel).
#include "sw_launch.hpp"

namespace swk {
#define SWK_CASE_SCAN_F32(R) case R: return launch_scan_r<F32, R>(multi, grid, stream, p);
#define SWK_CASE_PROF_F32(R) case R: return launch_profile_r<F32, R>(q, qlen, m, ns, out, s);
#define SWK_CASE_TILE_F32(R) case R: return (size_t)Geometry<F32, R>::kTileBytes;
SWK_DEFINE_KIND(launch_f32, F32, SWK_FOR_EACH_R_SCALAR, kMaxRowsScalar)
}  // namespace swk
