// sw_kind_f16x2.hip — instantiations of the DP scan kernel for kind F16X2 (one TU per kind so that the
// four kinds compile in parallel).
#include "sw_launch.hpp"

namespace swk {
#define SWK_CASE_SCAN_F16X2(R) case R: return launch_scan_r<F16X2, R>(multi, grid, stream, p);
#define SWK_CASE_PROF_F16X2(R) case R: return launch_profile_r<F16X2, R>(q, qlen, m, ns, out, s);
#define SWK_CASE_TILE_F16X2(R) case R: return (size_t)Geometry<F16X2, R>::kTileBytes;
SWK_DEFINE_KIND(launch_f16x2, F16X2, SWK_FOR_EACH_R_PACKED, kMaxRowsPacked)
}  // namespace swk
