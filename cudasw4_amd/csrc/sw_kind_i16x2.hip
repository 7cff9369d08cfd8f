// sw_kind_i16x2.hip — instantiations of the DP scan kernel for kind I16X2 (one TU per kind so that the
// four kinds compile in parallel).
#include "sw_launch.hpp"

namespace swk {
#define SWK_CASE_SCAN16_I16X2(R) case R: return launch_scan_r<I16X2, R, 16>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_SCAN64_I16X2(R) case R: return launch_scan_r<I16X2, R, 64>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF16_I16X2(R) case R: return launch_profile_r<I16X2, R, 16>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_PROF64_I16X2(R) case R: return launch_profile_r<I16X2, R, 64>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE16_I16X2(R) case R: return tile_bytes_r<I16X2, R, 16>();
#define SWK_CASE_TILE64_I16X2(R) case R: return tile_bytes_r<I16X2, R, 64>();
#define SWK_CASE_SCAN8_I16X2(R) case R: return launch_scan_r<I16X2, R, 8>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF8_I16X2(R) case R: return launch_profile_r<I16X2, R, 8>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE8_I16X2(R) case R: return tile_bytes_r<I16X2, R, 8>();
#define SWK_CASE_SCAN4_I16X2(R) case R: return launch_scan_r<I16X2, R, 4>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF4_I16X2(R) case R: return launch_profile_r<I16X2, R, 4>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE4_I16X2(R) case R: return tile_bytes_r<I16X2, R, 4>();
SWK_DEFINE_KIND(launch_i16x2, I16X2, SWK_FOR_EACH_R_PACKED)
}  // namespace swk
