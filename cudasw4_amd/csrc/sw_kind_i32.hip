// sw_kind_i32.hip — instantiations of the DP scan kernel for kind I32 (one TU per kind so that the
// four kinds compile in parallel).
#include "sw_launch.hpp"

namespace swk {
#define SWK_CASE_SCAN_I32(R) case R: return launch_scan_r<I32, R>(multi, grid, stream, p);
#define SWK_CASE_PROF_I32(R) case R: return launch_profile_r<I32, R>(q, qlen, m, ns, out, s);
#define SWK_CASE_TILE_I32(R) case R: return (size_t)Geometry<I32, R>::kTileBytes;
SWK_DEFINE_KIND(launch_i32, I32, SWK_FOR_EACH_R_SCALAR, kMaxRowsScalar)
}  // namespace swk
