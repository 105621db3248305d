// ttx_tt_spec64.hip -- the shape-specialised contraction kernels of the r = 64 family (ttx_tt_spec.inc), a translation unit of
// their own so that the families compile in parallel; entry points spec_fwd_64 / spec_bwd_64, called by ttx_tt.hip.
#include "ttx_tt_common.h"
#define TTX_SPEC_GROUP 64
namespace ttx {
#include "ttx_tt_spec.inc"
}  // namespace ttx
