/* lcr_hip_debug.h — test / tuning hooks of liblcr_hip.so.  NOT part of the product ABI (include/lcr_hip.h): process-global switches
 * that force a kernel form so that tests can hold two forms of the same op against each other and the bench tools can time them.
 * Nothing in lcr-net_amd/ calls them outside `functional.py`'s explicit A/B setters; a reference-side binding never needs them. */
#ifndef LCR_HIP_DEBUG_H
#define LCR_HIP_DEBUG_H

#ifdef __cplusplus
extern "C" {
#endif

/* tools/gemm_bench.py: 0 = heuristic tile choice, 1..5 = force 128x128 / 128x64 / 128x32 / 64x64 / 64x128. */
void lcr_gemm_debug_force_tile(int tile);
/* K-deep fp32 form (v_mfma_f32_32x32x2_f32, LDS-direct loads): -1 = environment LCR_GEMM_DEEP, 0 = off, 1 = heuristic, 2 = wherever legal. */
void lcr_gemm_debug_deep(int mode);
/* stream-K form of the K-deep contractions — persistent workgroups with equal contiguous (tile, K-step) ranges, partial tiles parked and
 * folded in ascending workgroup order by whichever workgroup parks a tile's last piece, nobody waits (+3..7 % on those shapes, opt-in):
 * -1 = environment LCR_GEMM_STREAMK (default 0 = off, 1 = heuristic), 0 = never, 2 = whenever the kernel is applicable. */
void lcr_gemm_debug_streamk(int mode);
/* GroupNorm apply: 1 = force the general (non-vectorised) kernel. */
void lcr_groupnorm_debug_general(int on);
/* KPConv aggregation: 1 = force 64-bit gather offsets where the host could guarantee 32-bit ones (bit-identity test of the two forms). */
void lcr_kpconv_debug_off64(int on);

#ifdef __cplusplus
}
#endif
#endif
