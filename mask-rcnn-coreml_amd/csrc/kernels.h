// kernels.h — host-side launchers of the gfx950 kernels (one namespace, no torch, no templates in the API).
#pragma once
#include <map>
#include <tuple>
#include <vector>

#include "common.h"

namespace mrcnn {

// ================================================================================================
// Box path (kernels_boxes.hip): top-k → decode → NMS, detection filtering.  -ffp-contract=off.
// ================================================================================================

// Workspace of the proposal path for a batch of images; all pointers are device memory.
struct ProposalWorkspace {
    int B = 0, A = 0, K = 0, Kpad = 0, max_keep = 0, nblk = 0, W = 0;
    uint32_t* keys = nullptr;        // [B][A]      order keys of the foreground scores
    uint32_t* hist = nullptr;        // [B][3][4096] radix-select histograms (12+12+8 bits)
    uint32_t* blockhist = nullptr;   // [B][nblk][256] per-block histogram of the last 8 bits
    uint32_t* state = nullptr;       // [B][8]      {prefix, k_remaining, threshold key, need_ties, n_gt, slot counter}
    uint64_t* cand = nullptr;        // [B][Kpad]   (~key << 32 | index), sorted ascending
    int32_t* topk_idx = nullptr;     // [B][K]      anchor index of the i-th best score
    float* boxes = nullptr;          // [B][K][4]   decoded + clipped boxes in score order
    uint64_t* nms_mask = nullptr;    // [B][K][W]   bit j of word w of row i: box 64w+j (> i) is suppressed by i
    int32_t* keep_idx = nullptr;     // [B][max_keep]
    int32_t* keep_count = nullptr;   // [B]
    static size_t bytes(int B, int A, int K, int max_keep);
    void bind(void* base, int B, int A, int K, int max_keep);
};

// ProposalLayer.evaluate for B images.  probs (B, A, 2) / deltas (B, A, 4) with batch strides in
// elements; anchors (A,4); rois out: B × max_proposals rows of `row_stride` floats (first 4 written,
// whole rows zeroed beyond the kept count — ProposalLayer.swift:181-192).
void proposal_forward(hipStream_t s, const ProposalWorkspace& ws, const float* probs, long probs_sB,
                      const float* deltas, long deltas_sB, const float* anchors, const float std4[4],
                      float nms_thr, float* rois, long rois_sB, long row_stride);

struct DetectionWorkspace {
    int B = 0, N = 0, max_det = 0, W = 0, Npad = 0;
    int32_t* count = nullptr;        // [B]        rows surviving the score/background filter
    int32_t* src = nullptr;          // [B][N]     original ROI index of filtered row k
    float* boxes = nullptr;          // [B][N][4]  refined, clipped
    float* score = nullptr;          // [B][N]
    int32_t* cls = nullptr;          // [B][N]
    uint64_t* nms_mask = nullptr;    // [B][N][W]
    int32_t* keep_idx = nullptr;     // [B][N]
    int32_t* keep_count = nullptr;   // [B]
    static size_t bytes(int B, int N, int max_det);
    void bind(void* base, int B, int N, int max_det);
};

// DetectionLayer.evaluate for B images: rois (B, N, 4) rows of roi_stride, cls6 (B, N, 6) contiguous
// rows, out: B × max_det rows of row_stride floats.
void detection_forward(hipStream_t s, const DetectionWorkspace& ws, const float* rois, long rois_sB,
                       long roi_stride, const float* cls6, long cls_sB, const float std4[4],
                       float score_thr, float nms_thr, int num_classes_hint, float* out, long out_sB,
                       long row_stride);

// ================================================================================================
// ROIAlign (kernels_roialign.hip).  -ffp-contract=off.
// ================================================================================================
struct PyramidMaps {
    const void* data[4];
    int H[4], W[4];
    long sB[4];          // batch stride in elements
    float mul[4] = {1.f, 1.f, 1.f, 1.f};   // NHWC kernel: every sample of level l is multiplied by mul[l] — exact powers of two that bring
                                           // levels stored at different split exponents to the output's (engine.h: SplitGroup)
};
// layout_nhwc = 1: maps are (B,H,W,C) and out is (B,n,P,P,C); 0: maps (B,C,H,W), out (B,n,C,P,P).
// dtype: element type of the maps and of the output (MRCNN_F16 only with layout_nhwc = 1).
void roi_align_forward(hipStream_t s, const PyramidMaps& maps, int C, int layout_nhwc, const float* rois,
                       long rois_sB, long roi_stride, int n_rois, int B, int pool, double image_w,
                       double image_h, void* out, long out_sB, long out_row_stride, int dtype,
                       int32_t* row_flags = nullptr);   // NHWC only: [B][n_rois] 1 iff every fp32 sample of the row != 0

// `.scaleFit` letterbox of an RGB8 image into an H×W canvas (content nh×nw at offset (py,px)).
void letterbox_forward(hipStream_t s, const uint8_t* src, int h, int w, uint8_t* dst, int H, int W, int nh, int nw, int py, int px);
// The same resampling fused with the mean subtraction, straight into the stem's padded staging tensor (B images of h×w).
void preprocess_scalefit_forward(hipStream_t s, const uint8_t* src, int B, int h, int w, int H, int W, int nh, int nw, int py, int px, int pad,
                                 const float mean[3], void* out, int dtype);
// Full-resolution binary instance masks from the 28×28 sigmoid masks (resize to box + threshold).
void paste_masks_forward(hipStream_t s, const float* det, long det_stride, const float* masks, int n, int S, int H, int W,
                         float thr, uint8_t* out);

// ================================================================================================
// Convolution family + element-wise helpers (kernels_conv.hip)
// ================================================================================================
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

struct ConvDesc {
    // element type of activations / filters / residual: MRCNN_F32 or MRCNN_F16 (fp32 accumulate)
    int dtype = MRCNN_F32;
    int out_f32 = 0;             // with MRCNN_F16: store the output(s) as fp32
    // element type of the filters; -1 = same as dtype.  (dtype F32, wdtype F16) selects the split mode:
    // fp32 tensors, each convolution as two fp16 MFMA passes over a hi/lo split of the activations.
    int wdtype = -1;
    // input NHWC (channel stride 1), arbitrary outer strides in elements
    const void* in = nullptr;
    int B = 0, H = 0, W = 0, Cin = 0;
    long in_sB = 0, in_sH = 0, in_sW = 0;
    // filter: packed [Npad][KH*KW*Cin], k contiguous (tap-major, channel-minor)
    const void* wgt = nullptr;
    // the same filters re-tiled for the halo kernel (kernels_conv_halo.hip: conv_halo_pack); nullptr = not available
    const void* wgt_halo = nullptr;
    // fp16 mode: the same filters in MFMA-fragment order for the fused bottleneck (kernels_bneck.hip: bneck_pack_frag); nullptr = not available
    const void* wgt_frag = nullptr;
    // fp16 mode, 3x3 stride-1 layers: the filters in the stream order of k_conv3x3_h (kernels_conv3x3_h.hip: conv3x3h_pack); nullptr = not available
    const void* wgt_c3h = nullptr;
    int prefer_c3h = 0;          // run on that kernel even without a fused head (the engine's A/B of the head fusion: "conv_c3h" 3)
    int KH = 1, KW = 1, stride = 1, padH = 0, padW = 0;
    // epilogue: y = act(acc*scale[n] + shift[n] + residual)
    const float* scale = nullptr;
    const float* shift = nullptr;
    // `res` MAY ALIAS `out` (the engine writes every bottleneck block's output in place over its shortcut).  Contract of every
    // epilogue form: the residual element at an address is loaded by the SAME thread that stores the output element there, and
    // before that store; no epilogue may prefetch residual elements another thread (or a later pass) stores over.  Pinned by
    // tests/test_gpu_conv_kernels.py::test_every_epilogue_is_safe_in_place_over_its_residual.
    const void* res = nullptr;
    long res_sB = 0, res_sH = 0, res_sW = 0;
    int res_shift = 0;           // residual read at (oh >> res_shift, ow >> res_shift): nearest 2× upsample when 1
    int act = ACT_NONE;
    // output NHWC: address = b*out_sB + (oh*OW+ow)*out_sP + n  (rows of the image contiguous)
    void* out = nullptr;
    int OH = 0, OW = 0, Cout = 0, Npad = 0;
    long out_sB = 0, out_sP = 0;
    // optional second output: columns >= n_split go to out2 (column n - n_split)
    void* out2 = nullptr;
    int n_split = 0;
    long out2_sB = 0, out2_sP = 0;
    // transposed-conv 2x2 stride 2 scatter: column n = q*Cout + co, q = dy*2+dx → pixel (2oh+dy, 2ow+dx)
    int deconv2 = 0;
    long out_sH = 0, out_sW = 0;  // only used when deconv2
    // algorithmic reduction length per output (defaults to KH*KW*Cin; conv1 pads 147 → 224)
    int algo_k = 0;
    // profile bookkeeping only: 1 = a backbone convolution (conv1, res2..res5), 0 = anything else (FPN, RPN, heads)
    int group = 0;
    // deconv2 only — selected-class dot instead of the store (the mask head's last two layers fused): for input image
    // (= ROI row) b with class sel_cid[b] >= 0, every output pixel P contributes
    //   sel_partial[(b * 4*OH*OW + P) * (Cout/128) + h] = Σ_{co in 128-channel part h} y[P][co] * sel_w[sel_cid[b]][co]
    // (y = the activated output, rounded to fp16 first when dtype is MRCNN_F16); nothing is written to `out`.
    const float* sel_w = nullptr;
    const int32_t* sel_cid = nullptr;
    float* sel_partial = nullptr;
    // halo kernel only — a 1x1 head fused into this layer's epilogue (the RPN's class / box heads on the shared 3x3 layer): the
    // layer's own output is not stored; head_out gets head columns [0, head_split), head_out2 the columns [head_split, head_cols),
    // each + head_bias.  head_w = conv_halo_pack_head of the [32][Cout] head filters.  conv_halo_head_eligible says whether the
    // layer can run fused (a property of its geometry only); conv_forward fails loudly when it cannot.
    const void* head_w = nullptr;
    const float* head_bias = nullptr;
    float* head_out = nullptr; float* head_out2 = nullptr;
    long head_out_sB = 0, head_out_sP = 0, head_out2_sB = 0, head_out2_sP = 0;
    int head_split = 0, head_cols = 0;
    float head_mul = 1.f;        // the head's sums are multiplied by this before the bias (2^-e of this layer's output group: exact)
};

// Live per-kernel profile of the conv family: when a profiler is active on the calling thread every
// conv_forward launch is bracketed by HIP events on its own stream; collect() (after the stream
// has been synchronised) folds the elapsed times into per-tile-shape totals.
struct ConvProfile {
    struct Slot { long launches = 0; double ms = 0, flops = 0, bytes = 0; };       // bytes: ALGORITHMIC bytes (conv_algorithmic_bytes: every operand once)
    Slot by_tile[9];                 // 8: 16 x 16 halo tiles x 256 columns of the fp16 mode's 3x3 layers (kernels_conv3x3_h.hip); 7: a whole identity bottleneck of the fp16 mode in one launch (kernels_bneck.hip; flops of its three layers); 0: 128x128, 1: 128x64, 2: 128x32, 3: 128x128 run by 4 waves of 32x128 (split modes, long K), 4: 256x256 ping-pong,
                                     // 5: persistent halo tiles (3x3 stride 1, split modes), 6: halo tiles with the fused bottleneck tail (3x3 + 1x1)
    std::vector<hipEvent_t> pool;
    struct Shape { int M, N, K, tile; bool operator<(const Shape& o) const { return std::tie(M, N, K, tile) < std::tie(o.M, o.N, o.K, o.tile); } };
    std::map<Shape, Slot> by_shape;  // per GEMM shape (M = images·OH·OW, N = output columns, K = taps·Cin)
    struct Pending { int tile; double flops; int e0, e1; Shape shape; int group = 0; double bytes = 0; };
    Slot by_group[2];                // ConvDesc::group: 0 = everything else, 1 = the backbone convolutions (C1..C5: north_star's roofline target is worded on them)
    std::vector<Pending> pending;
    int used = 0;
    bool active = false;
    void reset();
    void collect();
    ~ConvProfile();
};
void conv_set_profiler(ConvProfile* p);   // thread-local; nullptr disables
// One-time per-process set-up; idempotent, called at model / layer creation.
void boxes_one_time_init();
// While set (thread-local), every conv launch ORs 1 into *device_flag when one of its outputs leaves the fp16
// range (|v| >= 65504, inf or NaN): the watchdog of the fp16-MFMA modes, whose next layer reads it through fp16.
void conv_set_range_flag(int* device_flag);
// Device scratch of the convolution family (the partial sums + tile counters of shared-tile K chunks, the fused bottleneck tail's
// parking buffer).  OWNED by whoever launches — a Model (allocated at load, freed with it), a stand-alone test entry (for the call) —
// and handed to the launches of the calling thread like the range flag.  Without one the kernels that need it are not chosen
// (a chunked layer sums its chunks in one block: the same bits), so nothing is ever allocated behind a caller's back or inside a
// stream capture, and nothing outlives its owner.
struct ConvScratch {
    DevBuf ks_buf, ks_cnt, park;
    void alloc();            // synchronous; call outside any stream capture
    static size_t ks_bytes();
};
void conv_set_scratch(ConvScratch* c);    // thread-local; nullptr = none
// Picks the tile shape from Cout; returns the N tile it will use so that callers can pad weights.
int conv_n_tile(int Cout);
// sc != nullptr: `sc` is the 1x1 convolution whose output is d's residual (a ResNet stage's shortcut): computed inside d's launch where the pair
// qualifies (bit-identical to the two launches; the shortcut tensor is then not written), as its own launch before d otherwise
void conv_forward(hipStream_t s, const ConvDesc& d, const ConvDesc* sc = nullptr);
// Test / measurement switches of the kernel choice ("conv_pp" 0|1, "conv_pp_split" 0|1, "conv_pp_min_tiles", "conv_pp_min_kt",
// "conv_pp_min_fill" percent, "conv_pp_dbg" ablation bits); false = unknown key.
bool conv_debug_set(const char* key, int value);
bool boxes_debug_set(const char* key, int value);    // kernels_boxes.hip: "proposal_rank_sort", "nms_col_splits", "nms_class_fast" (no output bit depends on them)
// Halo kernel (kernels_conv_halo.hip): 3x3 stride-1 layers of the split modes.  conv_halo_pack re-tiles [Npad][9][Cin] fp16
// filters (device) into its granule layout; conv_halo_eligible says whether a layer can run on it (a property of the layer's
// geometry and mode only — never of the batch: the K order of the kernel differs from the 128-row kernel's).
struct ConvArgs;
void conv_halo_pack(hipStream_t s, const void* wgt_std, int Npad, int Cin, DevBuf& out, int taps = 9);
void conv_halo_pack_head(hipStream_t s, const void* wgt_std, int Npad, int Cin, DevBuf& out);
bool conv_halo_eligible(const ConvDesc& d);
bool conv_halo_head_eligible(const ConvDesc& d);
bool conv_halo_enabled();                   // the run-time switch mrcnn_debug_set("conv_halo") / MRCNN_HALO
bool conv_halo_debug_set(const char* key, int value);     // "halo_geo" 0 = the round-3 tile geometries (A/B, bit-identity tests) | 1
bool conv_halo_packable(int KH, int KW, int Cin, int Npad);   // the filter shapes conv_halo_eligible can accept (re-tile only those)
int conv_halo_forward(hipStream_t s, ConvArgs a, const ConvDesc& d, int parts, int n_cus, const ConvArgs* tail = nullptr, const void* tail_w = nullptr);
bool conv_halo_tail_packable(int KH, int KW, int Cin, int Npad);       // the 1x1 filters the fused bottleneck tail accepts (re-tiled with one tap)
bool conv_halo_tail_geometry_ok(int H, int W);                          // ... and the 3x3 tile geometries it is instantiated for
// A ResNet bottleneck's tail — d3: the 3x3 `branch2b` (halo-eligible, 256 output columns, ReLU), d1: the 1x1 `branch2c` that reads
// d3's output (256 -> 1024, + shortcut, ReLU; d1.wgt_halo = its filters re-tiled with one tap) — as ONE persistent launch in which the
// 256-column tensor between them never exists (kernels_conv_halo.hip: TAIL).  Bit-identical to conv_forward(d3); conv_forward(d1),
// which is what runs when the pair does not qualify, the grid would not fill the chip, or mrcnn_debug_set("conv_tail", 0).
bool conv_tail_fusable(const ConvDesc& d3, const ConvDesc& d1);
int conv_sel_part_cols();       // selected-class mode: output columns per partial sum the next conv_forward will leave (64: wave-private form, 128: block-staged)
void conv_forward_tail(hipStream_t s, const ConvDesc& d3, const ConvDesc& d1, const ConvDesc* sc = nullptr);      // sc: d1's shortcut convolution (conv_forward)

// An identity ResNet bottleneck of the fp16 mode — da: the 1x1 `branch2a` (4C -> C, ReLU), db: the 3x3 `branch2b` reading da's output
// (C -> C, ReLU), dc: the 1x1 `branch2c` reading db's output (C -> 4C, + shortcut = da's input, ReLU), C in {64, 128, 256} — as ONE
// persistent launch (kernels_bneck.hip): the two mid tensors never leave the chip.  BIT-IDENTICAL to conv_forward(da); conv_forward(db);
// conv_forward(dc), which is what conv_bneck_forward runs when the triple does not qualify or mrcnn_debug_set("conv_bneck", 0).  The fused
// form needs dc.out != da.in (a tile reads halo pixels its neighbours own); whether a triple qualifies is a property of the layers'
// geometry only, never of the batch.
bool conv_bneck_fusable(const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc);
// The stage-ENTRY block of C2 in the fp16 mode (res2a: stride 1, 64 -> 64 -> 64 -> 256, shortcut = the 1x1 convolution ds = `branch1` of the
// block's input, whose output is dc's residual) as one launch of the same kernel (FIRST form): bit-identical to conv_forward(da);
// conv_forward(ds); conv_forward(db); conv_forward(dc), which run when the block does not qualify / the grid under-fills the chip / "conv_bneck" 0.
bool conv_bneck_first_fusable(const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc, const ConvDesc& ds);
void conv_bneck_first_forward(hipStream_t s, const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc, const ConvDesc& ds);
bool conv_bneck_enabled();
void conv_bneck_forward(hipStream_t s, const ConvDesc& da, const ConvDesc& db, const ConvDesc& dc);
bool bneck_geometry_ok(int C, int H, int W);
// w2f / w3f: W2 / W3 in MFMA-fragment order (bneck_pack_frag) — used by the C = 256 form, whose waves stream their filter fragments
// straight from L2 into registers; nullptr selects the form with every operand staged through LDS
void bneck_launch(hipStream_t s, int C, const void* x, void* y, int B, int H, int W, const void* w1, const void* w2, const void* w3,
                  const float* s1, const float* h1, const float* s2, const float* h2, const float* s3, const float* h3, int* range_flag, int n_cus,
                  const void* w2f = nullptr, const void* w3f = nullptr, const void* w1f = nullptr,
                  const void* ws = nullptr, const float* ss = nullptr, const float* hs = nullptr);      // ws / ss / hs: the stage-entry form (C = 64)
// The identity blocks of a C = 256 stage (C4) as ONE launch — kernels_bneck.hip, STAGE form: a tile's block l starts when its <= 9 neighbour
// tiles have published block l - 1 (per-tile counters `done`, device-scope stores / loads), no launch gap or drain between blocks; BIT-IDENTICAL
// to nlayers calls of bneck_launch.  layers_dev: nlayers records of bneck_layer_record_bytes() bytes written by bneck_layer_record (host) and
// copied to the device; pp0 = the stage's input, pp1 = its ping-pong partner (result in pp[nlayers & 1]); done: >= B * tiles counters.
// range_flag (required): bit 0 as everywhere, bit 1 = a tile waited ~0.1 s for a neighbour (the grid was not resident as a whole): results invalid.
void bneck_stage_launch(hipStream_t s, const void* layers_dev, int nlayers, void* pp0, void* pp1, int B, int H, int W, unsigned* done, int* range_flag, int n_cus);
size_t bneck_layer_record_bytes();
void bneck_layer_record(void* dst, const void* w1f, const void* w2f, const void* w3f, const float* s1, const float* h1, const float* s2, const float* h2,
                        const float* s3, const float* h3);
struct BneckTriple { ConvDesc a, b, c; };
// conv_bneck_forward over the consecutive identity blocks of a stage: ONE launch where every block takes the fragment-streaming form, the grid
// fills the chip and mrcnn_debug_set("conv_bneck_stage", 1) (opt-in: measured equal); the per-block launches otherwise.  layers_dev / done as above (owned by the caller).
void conv_bneck_stage_forward(hipStream_t s, const BneckTriple* blocks, int n, const void* layers_dev, unsigned* done);
// [N][K] fp16 filters (K contiguous; N % 32 == 0, K % 16 == 0) -> 1-KB granules [N/32][K/16][lane 0..63][8]: lane (l31, kk) of granule (nt, kg)
// holds filter row 32 nt + l31, k = 16 kg + 8 kk .. + 7 — the first MFMA operand of v_mfma_f32_32x32x16_f16, one coalesced 16-B load per lane
void bneck_pack_frag(hipStream_t s, const void* wgt_std, int N, int K, DevBuf& out);
bool bneck_frag_wanted(int KH, int KW, int Cin, int Cout);      // the filter shapes the C = 256 form reads in fragment order

// The 3x3 stride-1 'same' layers of the fp16 mode with 256 | 512 output columns (kernels_conv3x3_h.hip): 16 x 16 halo tiles resident in LDS per
// 64-channel block, filter fragments streamed into registers, K order (channel block, tap, group) — its own: whether a layer runs here is
// decided by the layer alone (conv3x3h_eligible), never by the batch.  mrcnn_debug_set("conv_c3h", 0) sends such layers back to the
// 128-row / ping-pong kernels (their tap-major order: results differ by summation noise).
void conv3x3h_pack(hipStream_t s, const void* wgt_std, int N, int Cin, DevBuf& out);
bool conv3x3h_packable(int KH, int KW, int Cin, int Cout, int Npad);
bool conv3x3h_eligible(const ConvDesc& d);
void conv3x3h_launch(hipStream_t s, const ConvDesc& d, int* range_flag, int n_cus);
int conv_c3h_mode();                        // mrcnn_debug_set("conv_c3h") / MRCNN_C3H: 0 never | 1 where the RPN heads ride along (default) | 2 every eligible layer | 3 as 1 with the heads as their own launch

// The stem in the split modes and the fp16 mode (kernels_conv_stem.hip): conv1 — described by d exactly as for conv_forward (7 row taps of 32 "channels"
// on the zero-padded NHWC4 input, 64 output columns, ReLU) — and the 3x3 stride-2 'same' max-pool behind it in ONE persistent launch;
// conv1's output tensor is neither written nor read.  pooled: (B, PH, PW, 64) fp32.  Bit-identical to conv_forward(d) +
// maxpool3x3s2_forward.  conv_stem_eligible: whether d is that layer in a split mode or the fp16 mode (a property of the layer, not of the batch);
// mrcnn_debug_set("conv_stem", 0) sends callers back to the two launches.
bool conv_stem_eligible(const ConvDesc& d);
bool conv_stem_enabled();
void conv_stem_forward(hipStream_t s, const ConvDesc& d, void* pooled, int PH, int PW);
void conv_stem_launch(hipStream_t s, const void* in, int B, int Hp, int Wp, const void* wgt, const float* scale, const float* shift, int CH, int CW,
                      void* out, int PH, int PW, int parts, int* range_flag, int n_cus, bool compact = true);

// uint8 RGB (B,H,W,3) → fp32 (B, H+2*pad, W+2*pad, 4) minus mean, zero border, channel 3 = 0.
void preprocess_forward(hipStream_t s, const uint8_t* rgb, int B, int H, int W, int pad, const float mean[3],
                        void* out, int dtype);   // fp32 → NHWC4, fp16 → NHWC8
// 3×3 stride-2 max pool, Keras 'same' (window clipped at the bottom/right edge). NHWC, C % 4 == 0.
void maxpool3x3s2_forward(hipStream_t s, const void* in, int B, int H, int W, int C, void* out, int OH, int OW, int dtype);
// softmax over consecutive pairs: logits (n,2) → probs (n,2)
void softmax_pairs_forward(hipStream_t s, const float* logits, float* probs, long n_pairs);
// row softmax: logits rows of ld floats, first nc columns → probs (n, nc) contiguous
void softmax_rows_forward(hipStream_t s, const float* logits, long ld, int nc, long n, float* probs);
// copy bbox columns: src rows of ld floats starting at column c0, ncols columns → dst (n, ncols)
void copy_columns_forward(hipStream_t s, const float* src, long ld, int c0, int ncols, long n, float* dst);
// TimeDistributedClassifierLayer post-processing: probs (n,nc), bbox (n,nc*4) → rows (dy,dx,dh,dw,id,score)
void classifier_postprocess_forward(hipStream_t s, const float* probs, const float* bbox, int nc, long n,
                                    float* out, long out_row_stride);
// layout changes between Core ML's CHW and the engine's HWC
void nchw_to_nhwc_forward(hipStream_t s, const float* in, long n, int C, int H, int W, void* out, int dtype);
void nhwc_to_nchw_forward(hipStream_t s, const float* in, long n, int C, int H, int W, float* out);
// strided row gather: dst[i][0..len) = src[i*src_stride .. +len)
void copy_rows_forward(hipStream_t s, const float* src, long src_stride, long n, long len, float* dst, long dst_stride);

// TimeDistributedMaskLayer pieces -----------------------------------------------------------------
struct MaskSelectWorkspace {
    int32_t* flags = nullptr;      // [B][D] row is all-nonzero
    int32_t* mapping = nullptr;    // [B][D] compact index → row
    int32_t* kept = nullptr;       // [B]
    int32_t* sel_cid = nullptr;    // [B][D] class per row for the fused tail (mask_select_classes), -1 = not written
};
// flags/mapping of MultiArrayBatchProvider(removeZeros:true): pooled (B, D, row_len) contiguous rows.
// pooled == nullptr: ws.flags already holds the predicate (written by roi_align_forward's row_flags from the fp32
// samples — the fused engine path, also in fp16 mode where the stored rows are rounded); only the compaction runs.
void mask_valid_rows_forward(hipStream_t s, const void* pooled, long pooled_sB, long row_stride, long row_len,
                             int D, int B, const MaskSelectWorkspace& ws, int dtype);
// feat (B, D, HW, C) = ReLU(deconv) NHWC; w (nc, C), bias (nc); detections rows det_stride;
// out rows out_stride (>= HW).  Writes exactly what TimeDistributedMaskLayer.swift:58-89 writes.
void mask_select_forward(hipStream_t s, const void* feat, long feat_sB, int HW, int C, const float* w,
                         const float* bias, int nc, const float* det, long det_sB, long det_stride, int D, int B,
                         const MaskSelectWorkspace& ws, float* out, long out_sB, long out_stride, int dtype);
// The same result from the partial dots a deconv2 convolution in selected-class mode left (ConvDesc::sel_partial):
//   mask_select_classes: sel_cid[b*D + row] = the class the layer would use for that row, -1 for rows it does not write;
//   mask_select_from_partials: sigmoid(Σ_h partial[...][h] + bias[class]) into the rows the layer writes, zero padding as above.
void mask_select_classes(hipStream_t s, const float* det, long det_sB, long det_stride, int D, int B, int nc,
                         const MaskSelectWorkspace& ws, int32_t* sel_cid);     // sel_cid = ws.sel_cid in the engine
void mask_select_from_partials(hipStream_t s, const float* partial, int parts, int HW, const float* bias, int nc, const float* det,
                               long det_sB, long det_stride, int D, int B, const MaskSelectWorkspace& ws, float* out, long out_sB,
                               long out_stride);
// Variant for precomputed per-class masks (n_kept-compacted or in-place), used by the stand-alone layer:
// masks (B, D, nc, HW) in place (row r of the batch = detection r).
void mask_select_from_full_forward(hipStream_t s, const float* masks, long masks_sB, int HW, int nc,
                                   const float* det, long det_sB, long det_stride, int D, int B,
                                   const MaskSelectWorkspace& ws, float* out, long out_sB, long out_stride);

}  // namespace mrcnn
