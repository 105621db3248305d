// ttx_tt_spec16.hip -- the shape-specialised contraction kernels of the r = 16 family (ttx_tt_spec.inc), a translation unit of
// their own so that the families compile in parallel; entry points spec_fwd_16 / spec_bwd_16, called by ttx_tt.hip.
#include "ttx_tt_common.h"
#define TTX_SPEC_GROUP 16
namespace ttx {
#include "ttx_tt_spec.inc"
}  // namespace ttx
