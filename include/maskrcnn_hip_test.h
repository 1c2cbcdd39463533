/*
 * maskrcnn_hip_test.h — TEST AND MEASUREMENT entry points of libmaskrcnn_hip.so.
 *
 * Not part of the drop-in surface (include/maskrcnn_hip.h): these are what tests/, tools/ and bench.py use to run one
 * convolution of the kernel family on caller data, to time a layer shape, to read the live per-kernel profile of a
 * predict, and to switch kernel-selection policy for A/B comparisons.  The switches are PROCESS-WIDE and not
 * thread-safe; a production host never calls anything declared here.
 */
#ifndef MASKRCNN_HIP_TEST_H
#define MASKRCNN_HIP_TEST_H

#include "maskrcnn_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Live per-kernel profile of the convolution family during predict (bench.py roofline leg): when
 * enabled every conv launch is bracketed by HIP events on the model's stream.  tile: 0 = the
 * 128x128 kernel, 1 = 128x64, 2 = 128x32, 3 = 128x128 run by four waves of 32x128 (split modes, K >= 2048), 4 = 256x256 ping-pong,
 * 5 = the persistent halo tiles of the 3x3 layers of the split modes (kernels_conv_halo.hip), 6 = halo tiles with the fused
 * bottleneck tail (a 3x3 `branch2b` and the 1x1 `branch2c` behind it in one launch; flops of both layers).  enable(1) opens a measurement window (totals reset);
 * enable(0) closes it and the totals stay readable — the events cost ~2 % (fp32) / ~13 % (fp16) of a step,
 * so bench.py opens the window for the first steps of its timed region only.
 * total_flops is ALGORITHMIC work (2*M*N*K of the convolution, padding excluded). */
MRCNN_API int mrcnn_model_conv_profile_enable(mrcnn_model* model, int on);
MRCNN_API int mrcnn_model_conv_profile_get(mrcnn_model* model, int tile, int64_t* launches, double* total_ms,
                                           double* total_flops);
/* The same window split in two groups: 1 = the BACKBONE convolutions (conv1 and the res2..res5 stages: SURVEY.md section 8d's 344.9 GFLOP per image,
 * the subset north_star's ">= 50 % MFMA roofline" is worded on), 0 = every other convolution (FPN, RPN, heads). */
MRCNN_API int mrcnn_model_conv_profile_group(mrcnn_model* model, int group, int64_t* launches, double* total_ms, double* total_flops);
/* The same totals broken down by GEMM shape (M = images*OH*OW, N = output columns, K = taps*Cin): writes at
 * most `capacity` records, *count = number of distinct shapes seen (call with capacity 0 to size the buffer). */
typedef struct mrcnn_conv_shape_stat {
    int32_t M, N, K, tile;
    int64_t launches;
    double total_ms, total_flops;
    double total_bytes;      /* ALGORITHMIC bytes: every operand of the layer across HBM once (inputs read, filters, residual, outputs stored) */
} mrcnn_conv_shape_stat;
/* ALGORITHMIC bytes of the launches counted in tile class `tile` (as mrcnn_model_conv_profile_get's totals): what a memory-bound class is priced
 * against (bench.py: roofline.by_tile_class[*].bound / frac_of_hbm). */
MRCNN_API int mrcnn_model_conv_profile_bytes(mrcnn_model* model, int tile, double* total_bytes);
MRCNN_API int mrcnn_model_conv_profile_shapes(mrcnn_model* model, mrcnn_conv_shape_stat* out, int capacity, int* count);

/* ---------------------------------------------------------------------------------------------
 * Convolution micro-benchmark hook (bench.py roofline leg): runs one convolution of the trunk's
 * kernel family on synthetic data resident in HBM and reports the average kernel time measured
 * with HIP events on the launching stream.
 * --------------------------------------------------------------------------------------------- */
MRCNN_API int mrcnn_bench_conv(int batch, int h, int w, int cin, int cout, int ksize, int stride,
                               int iters, float* avg_ms, double* flops);
/* Same with an explicit element type (MRCNN_F32 | MRCNN_F16, fp32 accumulate). */
MRCNN_API int mrcnn_bench_conv_dtype(int batch, int h, int w, int cin, int cout, int ksize, int stride,
                                     int iters, int dtype, float* avg_ms, double* flops);

/* What the matrix cores of THIS board sustain right now: every wave of the chip on back-to-back MFMAs (kind MRCNN_F16: v_mfma_f32_32x32x16_f16,
 * MRCNN_F32: v_mfma_f32_32x32x2_f32) whose register operands change from one instruction to the next, no operand traffic, for `seconds`
 * (<= 30); *tflops = the rate over the last third of the run, *mhz_equivalent (optional) = the shader clock it means at back-to-back issue.
 * bench.py runs it in-process before the timed loop (`roofline.sustained_peak_live`): boxes of one pool differ by several per cent under
 * the power cap, and this is what separates a kernel change from a box change. */
MRCNN_API int mrcnn_bench_mfma_probe(double seconds, int kind, double* tflops, double* mhz_equivalent);

/* One convolution of the engine's kernel family on caller (host) data — the unit the parity tests of the kernels use:
 * in (B,H,W,Cin) NHWC fp32, filters (Cout, k, k, Cin) fp32 (k = 1 | 3, 'same' padding k/2), optional per-channel
 * scale/shift (folded BatchNorm + bias), optional residual (B,OH,OW,Cout), act 0 none | 1 ReLU | 2 sigmoid;
 * dtype = compute mode (inputs are converted to it on the host, round-to-nearest); out (B,OH,OW,Cout) fp32
 * (MRCNN_F16: the fp16 values the layer stores, widened).  mrcnn_debug_set switches kernel-selection policy knobs for A/B tests
 * ("conv_pp" 0|1: the 256-row ping-pong fp16 kernels; "conv_pp_min_tiles", "conv_pp_min_kt", "conv_pp_min_fill", "conv_pp_split",
 * "conv_pp_dbg"; "conv_tn4" -1|0|1: split modes, 128x128 tile as 4 waves of 32x128 by policy | never | always; "conv_min_blocks": the grid
 * size below which the N tile is narrowed; "conv_direct" 0|1|2|3: epilogue without block barriers never | fp16 tensors straight from the accumulators | + fp32 tensors through wave-private LDS tiles | + the fp16 tensors of the 128-column kernel (default 3; all four bit-identical); "mask_fused" 0|1: the mask head's
 * deconvolution + selected-class 1x1 as two launches over a materialised tensor | fused — results within fp32 summation noise): every choice must give
 * bit-identical results — the tile shape depends on the batch size and per-image results must not.  Further knobs: "conv_halo" 0|1 the
 * persistent halo kernel of the 3x3 layers of the split modes (its K order is its own: results differ from "0" by summation noise);
 * "halo_geo" 0|1: its round-3 tile geometries | two-row tiles, region-sized staging, conflict-free LDS pitch (bit-identical);
 * "conv_tail" 0|1: bottleneck tails as two launches | one fused launch where the grid fills the chip (bit-identical; default 0: measured slower);
 * "conv_stem" 0|1|2: conv1 and the max-pool as two launches | one fused launch (default; split modes: bit-identical to 0) | fp16 tensors: the fused launch in
 * round 4's four-K-group form (bit-identical to 0; the default compact form — two K groups per kernel row, half the MFMAs — differs from it by summation noise);
 * "halo_lat" 0|1|2: 3x3 layers on grids under 3/8 of the chip (single images at C5 / P5): the eight-wave 64 x 128 tiles | 64 x 64 tiles on
 * four waves with deep prefetch (default) | that form for every grid under 3/4 (bit-identical);
 * "halo_n64" 0|1|2: 64-column 3x3 layers of the split modes on the 64-column 128-row kernel | on the halo kernel as 128 x 64 tiles | and as 256 x 64 tiles where
 * four image rows x 64 columns tile the level and the grid fills the chip (default 2; 1 and 2 are bit-identical, "0" differs by summation noise: its K order is
 * the 128-row kernel's);
 * "halo_rounds" 0|1: 128 x 128 halo tiles whenever the grid covers 3/4 of the chip | 64 x 128 where they shorten a nearly empty last round (default 1; bit-identical);
 * "mask_sel_wave" 0|1: the mask head's deconvolution + selected-class dot through the block-staged epilogue (128-column partial sums) | straight from the
 * accumulators (64-column partial sums; default 1) — the two differ by summation noise;
 * "conv_scfuse" 0|1: a ResNet stage's shortcut convolution as its own launch | inside the launch of the `branch2c` that adds it (default 1; bit-identical);
 * "conv_kchunk" 0|1: the long-K (>= 2048) 1x1 layers of the split modes as one running sum | as 4 / 8 canonical K chunks folded in order
 * (default 1; the two differ by summation noise — the chunk count is a property of the layer, never of the batch);
 * "conv_ksplit" 0|1 and "conv_ksplit_below" n: chunked layers whose widest-tile grid has fewer than n (256) blocks give every chunk its
 * own block, the last one to finish folds the partial sums | never (bit-identical to each other).
 * "conv_min_blocks_split" n: the same threshold for the split modes alone (default 256 = one block per CU; fp16 tensors keep 448; bit-identical);
 * box path (kernels_boxes.hip; none changes an output bit): "proposal_rank_sort" 0|1: the K pre-NMS candidates ordered by one bitonic-sort block per
 * image | by rank counting over the whole chip (default 1); "nms_col_splits" n: column splits of the suppression-matrix grid (0 = by policy: as many
 * as the chip's wave slots hold, at least 4); "nms_class_fast" 0|1: DetectionLayer's per-class limit tested per chunk as count + 64 | as count + the
 * chunk's own alive candidates of the class, classes at the limit dropping out (default 1).
 * Measurement-only environment variable read by mrcnn_model_load: MRCNN_CU_MASK_PROBE="w0,...,w7" (hexadecimal) creates the handle's stream on that subset of the
 * CUs (hipExtStreamCreateWithCUMask) — tools/dual_stream_probe.py's question whether two half batches on disjoint halves of the chip beat one batch on all of it (no).
 * The switches are PROCESS-WIDE test / measurement knobs: not thread-safe; a choice captured in a hipGraph stays captured.
 * ARMING (round 6): mrcnn_debug_set, the MRCNN_* environment overrides of the switches' defaults and the measurement hooks (MRCNN_CU_MASK_PROBE,
 * MRCNN_BNECK_DBG, MRCNN_SPLIT_EXP, MRCNN_BENCH_RESIDUAL) work only in a process started with MRCNN_TEST_KNOBS=1 in its environment (tests/conftest.py
 * and the tools set it; read once, at the library's first use).  Anywhere else — every production host — mrcnn_debug_set returns MRCNN_ERR_UNSUPPORTED
 * and the overrides are ignored: the alternative forms stay in the library as the tests' bit-identity anchors, out of a host's reach
 * (tests/test_host.py::test_the_knobs_are_out_of_a_production_hosts_reach).  Forms that lost their A/B and anchor nothing are removed instead
 * (round 6: the level-parallel FPN / RPN region of round 5). */
MRCNN_API int mrcnn_conv2d_nhwc(const float* in, int batch, int h, int w, int cin, const float* filters, int cout,
                                int ksize, int stride, const float* scale, const float* shift, const float* residual,
                                int act, int dtype, float* out);
MRCNN_API int mrcnn_debug_set(const char* key, int value);

/* An identity ResNet bottleneck block of the fp16 mode on caller (host) data: x (B,H,W,4C) fp32 (rounded to fp16 on the host),
 * w1 (C,4C), w2 (C,3,3,C), w3 (4C,C) fp32 (rounded to fp16), folded BatchNorm scale / shift per layer; out (B,H,W,4C) = the fp16
 * values the block stores, widened.  fused = 1: ONE persistent launch with both mid tensors on chip (kernels_bneck.hip; needs
 * C in {64,128,256}, W % 16 == 0, H % 16 == 0 (C = 256: H % 8 == 0)); fused = 2: the same with every operand staged through LDS (C = 256
 * otherwise streams its filter fragments straight into registers); fused = 0: the three launches of the convolution family.
 * All must agree bit for bit (tests/test_gpu_bneck.py).  iters > 0: *avg_ms = average time of `iters` further runs (HIP events).
 * "conv_bneck" 0|1|2|3 of mrcnn_debug_set: the engine's identity bottlenecks of the fp16 mode as three launches | fused where the grid
 * fills the chip (default) | fused, all-LDS form | fused at every grid size (bit-identical, all four). */
MRCNN_API int mrcnn_bottleneck_nhwc(const float* x, int batch, int h, int w, int cmid, const float* w1, const float* w2, const float* w3,
                                    const float* s1, const float* h1, const float* s2, const float* h2, const float* s3, const float* h3,
                                    int fused, int iters, float* out, float* avg_ms);
/* The stage-ENTRY block of C2 (res2a; fp16 mode): x (B,H,W,C) with C = 64, w1 (C,C), w2 (C,3,3,C), w3 (4C,C), ws (4C,C) = the shortcut convolution
 * `branch1`; bn[8] = scale / shift of branch2a, 2b, 2c, branch1 in that order; out (B,H,W,4C).  fused = 1: one launch, 0: the four launches — bit for bit. */
MRCNN_API int mrcnn_bottleneck_first_nhwc(const float* x, int batch, int h, int w, int cmid, const float* w1, const float* w2, const float* w3, const float* ws,
                                          const float* const bn[8], int fused, int iters, float* out, float* avg_ms);

/* The consecutive identity blocks of a C = 256 stage (C4's 22 in ResNet-101) on caller data: x (B,H,W,1024), stacked operands w1 (n,256,1024),
 * w2 (n,256,3,3,256), w3 (n,1024,256), bn = {s1,h1,s2,h2: (n,256); s3,h3: (n,1024)}.  form 1: ONE launch in which a tile's block l waits for its
 * neighbour tiles' block l - 1 (kernels_bneck.hip, STAGE form); form 0: one fused launch per block.  Bit-identical (tests/test_gpu_bneck.py).
 * out (B,H,W,1024) = the last block's output; *status_flag (optional) = the launch's flag word (bit 0 fp16 range, bit 1 "no progress").
 * "conv_bneck_stage" 0|1 of mrcnn_debug_set: the engine's C4 identity blocks of the fp16 mode one launch per block (default) | one launch per stage
 * (measured equal: profiles/r06_bneck_stage_ab.txt). */
MRCNN_API int mrcnn_bottleneck_stage_nhwc(const float* x, int batch, int h, int w, int nlayers, const float* w1, const float* w2, const float* w3,
                                          const float* const bn[6], int form, int iters, float* out, float* avg_ms, int* status_flag);


#ifdef __cplusplus
}
#endif

#endif /* MASKRCNN_HIP_TEST_H */
