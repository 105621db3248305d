// ttx_tt_spec32.hip -- the shape-specialised contraction kernels of the r = 32 family (ttx_tt_spec.inc), a translation unit of
// their own so that the families compile in parallel; entry points spec_fwd_32 / spec_bwd_32, called by ttx_tt.hip.
#include "ttx_tt_common.h"
#define TTX_SPEC_GROUP 32
namespace ttx {
#include "ttx_tt_spec.inc"
}  // namespace ttx
