/*
 * recoder_hip_probe.h -- tuning probes and test switches of librecoder_hip.so: NOT part of the drop-in
 * boundary (include/recoder_hip.h).  tools/probes/ and a few tests use them; nothing in the product path does.
 */
#ifndef RECODER_HIP_PROBE_H
#define RECODER_HIP_PROBE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* Phase stamps: a device buffer of 8 uint64 per workgroup (per user row for RK_PROBE_ENC) of the largest
 * grid, NULL switches the probe off (the default).
 *   RK_PROBE_GEMM   the fp32 GEMM kernels of csrc/gemm.hip     (tools/gemm_probe.py)
 *   RK_PROBE_DW3    the fp16-pair dW kernel of csrc/dw3.hip    (tools/dw3_probe.py)
 *   RK_PROBE_ENC    the encoder forward                        (tools/probes/enc_phase_probe.py)
 *   RK_PROBE_PLANES the plane kernels of csrc/decode16.hip     (tools/probes/planes_phase_probe.py) */
enum { RK_PROBE_GEMM = 0, RK_PROBE_DW3 = 1, RK_PROBE_ENC = 2, RK_PROBE_PLANES = 3 };
int rk_probe_buffer(int32_t which, unsigned long long *buffer);

/* A/B and tuning switches, set inside one process before the first launch they affect (the defaults are
 * the product; none of them is read from the environment):
 *   RK_TUNE_LINEAR_PAIR   1  dX and dW of a hidden layer's backward as ONE launch
 *   RK_TUNE_PLANES_TILE   0  rows of the decode tile of csrc/decode16.hip: 64 / 128, 0 = by shape
 *   RK_TUNE_DZ_FUSED      1  the fused decode + loss + dZ launch where it applies (rk_plan_t.decode_dz_fused_ok)
 *   RK_TUNE_DW_ENC_FUSED  1  dW || encoder backward in one launch (rk_plan_t.dw_encode_bwd_fused_ok); 0: never (the step then
 *                            leaves the register-resident fused decode too)
 *   RK_TUNE_DW_BF16X3     0  dW on bf16 triples (no operand range) instead of fp16 pairs
 *   RK_TUNE_ADAM_DE_SIDE  -  (the decoder table's Adam sweep as a launch of its own behind dW on dw_stream: removed in
 *                            round 6 -- +1.2 % at C2 in round 3, excluded by the lazy sweeps since; the index stays reserved)
 *   RK_TUNE_PG_TILE       0  decode tile of csrc/pgemm.hip: 256 (256 x 256), 1282 (128 x 256), 0 = by batch size
 *   RK_TUNE_DZ_TN         0  column tiles (of 32 hidden units) per workgroup of rk_decode_bwd_dz_planes: 2/4/7/8
 *   RK_TUNE_DZ_SPLITS     0  cap of rk_decode_bwd_dz's split-K (multiple of 8)
 *   RK_TUNE_PAIR_ORDER    0  1: the second GEMM's tiles first in rk_linear_bwd's paired launch
 *   RK_TUNE_GRAPH_EVENT_NODES 0  1: bench brackets inside a captured graph as event-record nodes
 *   RK_TUNE_MF_FDEC       1  MatrixFactorization steps on rk_fdec_loss_dz + rk_pg_dw_dz_reduce (rk_plan_t.mf_fdec_ok); 0: the
 *                            round-3 pair rk_decode_loss_dz_planes + rk_decode_bwd_dw2_dz_reduce
 *   RK_TUNE_DW_RING       2  LDS stages of the 64 x 128 dW tiles' ring k-loop (csrc/pgemm.h: counted vmcnt waits, raw
 *                            s_barrier, asm transpose reads): 2 (default: the LDS footprint of the two-stage loop, the next
 *                            tile's DMA now really under the MFMAs), 4 = a deeper ring (3 / 4 / 6 stages measured: faster
 *                            alone -- 18.0 -> 13.2 us at C2's shape --, slower in the merged launches: every workgroup range
 *                            pays the LDS; only 4 is still instantiated); 0 = the
 *                            two-stage loop as the compiler schedules it (vmcnt(0) before the first transpose read)
 *   RK_TUNE_DW_ONES       1  whole steps on the fused decode, h % 32 != 0: the decoder bias gradient as output column h of the dW
 *                            tiles (a ones column in the Z image's padding) instead of a column-sum range over the dO image
 *   RK_TUNE_FDEC_STREAM   -  (round 5's streaming form of the fused decode: removed in round 6, the index stays reserved --
 *                            tools/probes/patches/r06_fdec_streaming_form.patch) */
enum { RK_TUNE_LINEAR_PAIR = 0, RK_TUNE_PLANES_TILE = 1, RK_TUNE_DZ_FUSED = 2, RK_TUNE_DW_ENC_FUSED = 3,
       RK_TUNE_DW_BF16X3 = 4, RK_TUNE_ADAM_DE_SIDE = 5, RK_TUNE_PG_TILE = 6, RK_TUNE_DZ_TN = 7,
       RK_TUNE_DZ_SPLITS = 8, RK_TUNE_PAIR_ORDER = 9, RK_TUNE_GRAPH_EVENT_NODES = 10, RK_TUNE_FDEC_STREAM = 11,
       RK_TUNE_MF_FDEC = 12, RK_TUNE_DW_RING = 13, RK_TUNE_DW_ONES = 14, RK_TUNE_COUNT = 15 };
int rk_tune(int32_t knob, int32_t value);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
