// Temporary: entry points whose implementation lands later in this round.
#include "geometry.hpp"
#include "stabilizer.hpp"
namespace gtx {
struct Stabilizer::Impl {};
Stabilizer::Stabilizer(gtx_ctx*, const gtx_stab_config&) { fail(GTX_ERR_UNSUPPORTED, "stabilizer not built yet"); }
Stabilizer::~Stabilizer() = default;
void Stabilizer::set_ref_frame(const uint8_t*, int, int, const float*, int) {}
void Stabilizer::set_ref_gray_dev(const void*, int, int, const float*, int) {}
void Stabilizer::stabilize(const uint8_t*, int, int, const float*, int, double*, int*, int*) {}
void Stabilizer::stabilize_gray_dev(const void*, int, int, const float*, int, double*, int*, int*) {}
void Stabilizer::keypoints(int, int, int*, float*, int*, int*, uint8_t*) {}
void Stabilizer::matches(int, int*, int*, int*, int*) {}
void warp_frame(gtx_ctx*, const uint8_t*, int, int, const double*, uint8_t*) { fail(GTX_ERR_UNSUPPORTED, "warp_frame not built yet"); }
}  // namespace gtx
