/* ttx_test_hooks.h -- test / ablation knobs of libttx.  TEST BUILD ONLY.
 *
 * These entry points exist in libttx_hooks.so, the same sources as libttx.so compiled with -DTTX_TEST_HOOKS
 * (__graft_entry__.build()); the product library exports none of them and compiles every knob as a constant at its
 * default (tests/test_module_cpu.py::test_product_library_has_no_test_knobs).  The knobs are plain globals of the test
 * library -- not per stream, not thread-safe: parity tests of code paths small shapes would not reach (the generic kernels,
 * their block walk, odd chunk sizes) and A/B timing (scripts/ablate*.py, phase_times*.py).  The Python shim routes through
 * the test library only while a knob is away from its default (tt_embeddings.debug_skip / set_chunk / debug_lds_budget ...).
 */
#ifndef TTX_TEST_HOOKS_H
#define TTX_TEST_HOOKS_H
#include "ttx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* tuning knob (bench A/B, tests): indices per work-group chunk; 0 = heuristic */
int ttx_set_chunk(int32_t indices_per_chunk);
/* LDS budget in bytes (<= 163840; 0 = that) of the generic kernels' tile search.  The generic contraction kernels walk a
 * core_1 slice in K blocks x column passes sized to the budget (csrc/ttx_tt_generic.inc), so a small budget drives small
 * shapes through the walk that ranks >= 80 need.  Set it before sizing workspaces / plans. */
int ttx_debug_lds_budget(int32_t bytes);
/* bit mask: bits 0..7 kernel phases to skip (results INVALID), bit 8 force the generic kernels, bits 9..12 leave out launches,
 * bit 15 shape-specialised kernels for exact shapes only; bit 16 reduce_apply with a work-group per small slice instead of a wave
 * per slice (round 6; results stay valid up to the order of addition).  0 = normal operation. */
int ttx_debug_skip(int32_t mask);
/* A/B knob (scripts/bench_cache.py): 1 = ttx_cache_forward uses the one-group-per-lookup kernel for every D */
int ttx_debug_cache_fwd(int32_t lookup_groups);
/* experiment (round 6): lookups per chunk (a multiple of 32, <= 1024) of bwd32_kernel -- the backward of q = [4,4,4], ranks [32,32] at
 * >= 131072 lookups with eight lookups per wave on v_mfma_f32_32x32x2 and persistent work-groups; 0 = spec_bwd_kernel (the product's) */
int ttx_debug_bwd32(int32_t lookups_per_chunk);
/* (scripts/phase_times.py) device buffer receiving 16 int64 wall-clock stamps per backward work-group; NULL = off */
int ttx_debug_stamps(void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif
