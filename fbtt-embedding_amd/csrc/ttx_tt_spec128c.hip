// ttx_tt_spec128c.hip -- the shape-specialised contraction kernels for r = 128, q = [4,8,8] (ttx_tt_spec.inc), a translation
// unit of their own so that the shapes compile in parallel; entry points spec_fwd_128c / spec_bwd_128c, called by ttx_tt.hip.
#include "ttx_tt_common.h"
#define TTX_SPEC_GROUP 1283
namespace ttx {
#include "ttx_tt_spec.inc"
}  // namespace ttx
