// sw_kind_f32.hip — instantiations of the DP scan kernel for kind F32 (one TU per kind so that the
// four kinds compile in parallel).
#include "sw_launch.hpp"

namespace swk {
#define SWK_CASE_SCAN16_F32(R) case R: return launch_scan_r<F32, R, 16>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_SCAN64_F32(R) case R: return launch_scan_r<F32, R, 64>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF16_F32(R) case R: return launch_profile_r<F32, R, 16>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_PROF64_F32(R) case R: return launch_profile_r<F32, R, 64>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE16_F32(R) case R: return tile_bytes_r<F32, R, 16>();
#define SWK_CASE_TILE64_F32(R) case R: return tile_bytes_r<F32, R, 64>();
#define SWK_CASE_SCAN8_F32(R) case R: return launch_scan_r<F32, R, 8>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF8_F32(R) case R: return launch_profile_r<F32, R, 8>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE8_F32(R) case R: return tile_bytes_r<F32, R, 8>();
#define SWK_CASE_SCAN4_F32(R) case R: return launch_scan_r<F32, R, 4>(multi, offs, grid, reserve, stream, p);
#define SWK_CASE_PROF4_F32(R) case R: return launch_profile_r<F32, R, 4>(q, qlen, m, pr, ns, out, shift, s);
#define SWK_CASE_TILE4_F32(R) case R: return tile_bytes_r<F32, R, 4>();
SWK_DEFINE_KIND(launch_f32, F32, SWK_FOR_EACH_R_I32)
}  // namespace swk
