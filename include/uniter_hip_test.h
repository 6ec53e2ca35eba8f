/* uniter_hip_test.h — test / tuning hooks of libuniter_hip.so.  NOT part of the drop-in boundary (include/uniter_hip.h): these
 * switches exist for tests/native/test_kernels.cpp, the probes under tests/native/ and tests/test_gpu_parity.py, which force a tile,
 * a launch order or a stream arrangement to compare it with the default.  Process-wide, not thread-safe, no stability promise. */
#ifndef UNITER_HIP_TEST_H
#define UNITER_HIP_TEST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Force the GEMM tile (index into the build's tile table, uniter_gemm_tile_count(); -1 = tuned table / cost model) and the weight-
 * gradient split-K factor (-1 = heuristic) for every following GEMM call of the process. */
int uniter_gemm_debug_force(int cfg, int splits);
/* 0x100: uniter_gemm_bias_gelu_fwd writes gelu'(u) where it documents u, and uniter_gemm_dgrad_gelu takes that tensor as its `u`
 * argument and multiplies by it — the form uniter_encoder_forward / _backward use between FFN1 and the FFN2 data gradient
 * (csrc/common.cuh, UH_ACT_SAVE_GRAD); 0 (default): the documented forms. */
int uniter_gemm_debug_act_flags(int flags);
/* 0: uniter_encoder_backward runs the weight-gradient GEMMs and bias column sums on the caller's stream instead of the library's
 * weight-gradient stream (default 1). */
int uniter_encoder_debug_side_stream(int enable);
/* 1: the dependent kernels of a forward / backward call are dispatched as an overlapped kernel chain (include/uniter_hip.h
 * "Overlapped kernel chains"); default 0 = in-order launches. */
int uniter_encoder_debug_chain(int enable);
/* 0: uniter_encoder_autotune keeps the isolated per-GEMM winners; 1 (default): it then re-picks every GEMM's tile among its fastest
 * candidates by timing a short forward + backward stack. */
int uniter_encoder_debug_tune_in_situ(int enable);

#ifdef __cplusplus
}
#endif
#endif
