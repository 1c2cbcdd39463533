// kernels_boxes.hip — the proposal path and the detection path on gfx950.
//
// Replaces (reference, Sources/Mask-RCNN-CoreML/):
//   ProposalLayer.evaluate              ProposalLayer.swift:103-195
//   sortedIndices / indexed / broadcastedIndices / elementWiseMultiply   Utils.swift:28-38,56-66,112-148,173-180
//   applyBoxDeltas / clip               BoxUtils.swift:32-80
//   nonMaxSupression / IOU              Utils.swift:185-246
//   DetectionLayer.evaluate             DetectionLayer.swift:107-276
//
// Design (HBM-bound integer/byte work — no MFMA here):
//   * top-k = 3-pass radix select (12+12+8 bits) over monotone uint32 keys, coalesced 16-B loads,
//     LDS histograms, then an order-preserving tie compaction (ties → lowest anchor index, the
//     total order the parity tests pin) and one in-LDS bitonic sort of the K survivors per image;
//   * gather of the K anchors / deltas as 16-B rows, decode + clip in registers;
//   * NMS = all-pairs suppression bit-matrix (64×64 tiles, one wave per tile, IoU in fp64 exactly
//     like the CGRect code) + a chunked greedy scan: a wave resolves 64 candidates at a time from
//     the diagonal word with ballot/readlane, then the block ORs the kept rows into the removed
//     set.  Early exit at maxProposals kept.
// Compiled with -ffp-contract=off (see device_math.h).
#include <mutex>
#include <string>

#include "device_math.h"
#include "kernels.h"

namespace mrcnn {

static constexpr int CHUNK = 1024;   // scores per block in the select passes (256 threads × 4)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static int boxes_env_int(const char* name, int dflt) { const char* e = knob_env(name); return e && *e ? atoi(e) : dflt; }      // (honoured only with MRCNN_TEST_KNOBS=1)
// Run-time switches (A/B and bit-identity tests; mrcnn_debug_set): none of them changes an output bit
static int g_rank_sort = boxes_env_int("MRCNN_RANK_SORT", 1);       // "proposal_rank_sort": 1 rank counting over the chip (k_rank_decode), 0 the one-block bitonic sort
static int g_nms_splits = boxes_env_int("MRCNN_NMS_SPLITS", 0);     // "nms_col_splits": column splits of k_nms_mask's grid; 0 = by policy (nms_col_splits)
static int g_nms_fast = boxes_env_int("MRCNN_NMS_FAST", 1);         // "nms_class_fast": 1 the per-class limit test on the chunk's own candidates, 0 the round-4 test (count + 64)
bool boxes_debug_set(const char* key, int value)
{
    const std::string k = key;
    if (k == "proposal_rank_sort") g_rank_sort = value;
    else if (k == "nms_col_splits") g_nms_splits = value;
    else if (k == "nms_class_fast") g_nms_fast = value;
    else return false;
    return true;
}
// Column splits of the suppression-matrix launch.  Row block rb owns the column chunks rb .. W-1, so with a fixed split the blocks
// of row 0 walk W / 16 chunks one after the other while the chip idles (single image, W = 94: six rounds, 79 us): as many splits as
// keep the launch within the chip's wave slots — every wave at most one chunk on a single image (22 us) — and never fewer than four.
static int nms_col_splits(int W, int B)
{
    if (g_nms_splits > 0) return g_nms_splits;
    if (W < 16) return 1;
    const long full = (W + 3) / 4, fit = 16384 / ((long)W * B * 4);
    return (int)(fit < 4 ? 4 : fit > full ? full : fit);
}
static inline int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// ------------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------------
size_t ProposalWorkspace::bytes(int B, int A, int K, int max_keep)
{
    const int nblk = (A + CHUNK - 1) / CHUNK, W = (K + 63) / 64, Kpad = next_pow2(K);
    size_t n = 0;
    n += align_up((size_t)B * (size_t)(nblk * CHUNK) * 4, 256);   // keys (padded to whole chunks)
    n += align_up((size_t)B * 3 * 4096 * 4, 256);                  // hist
    n += align_up((size_t)B * nblk * 256 * 4, 256);                // blockhist
    n += align_up((size_t)B * 8 * 4, 256);                         // state
    n += align_up((size_t)B * Kpad * 8, 256);                      // cand
    n += align_up((size_t)B * K * 4, 256);                         // topk_idx
    n += align_up((size_t)B * K * 16, 256);                        // boxes
    n += align_up((size_t)B * K * W * 8, 256);                     // nms_mask
    n += align_up((size_t)B * max_keep * 4, 256);                  // keep_idx
    n += align_up((size_t)B * 4, 256);                             // keep_count
    return n;
}

void ProposalWorkspace::bind(void* base, int B_, int A_, int K_, int max_keep_)
{
    B = B_; A = A_; K = K_; max_keep = max_keep_;
    nblk = (A + CHUNK - 1) / CHUNK; W = (K + 63) / 64; Kpad = next_pow2(K);
    char* p = (char*)base;
    auto take = [&](size_t n) { char* r = p; p += align_up(n, 256); return r; };
    keys = (uint32_t*)take((size_t)B * (size_t)(nblk * CHUNK) * 4);
    hist = (uint32_t*)take((size_t)B * 3 * 4096 * 4);
    blockhist = (uint32_t*)take((size_t)B * nblk * 256 * 4);
    state = (uint32_t*)take((size_t)B * 8 * 4);
    cand = (uint64_t*)take((size_t)B * Kpad * 8);
    topk_idx = (int32_t*)take((size_t)B * K * 4);
    boxes = (float*)take((size_t)B * K * 16);
    nms_mask = (uint64_t*)take((size_t)B * K * W * 8);
    keep_idx = (int32_t*)take((size_t)B * max_keep * 4);
    keep_count = (int32_t*)take((size_t)B * 4);
}

// state words
enum { ST_PREFIX = 0, ST_KREM = 1, ST_T = 2, ST_NEED = 3, ST_NGT = 4, ST_SLOT = 5, ST_TIEBIN = 6 };

// ------------------------------------------------------------------------------------------------
// pass 0: foreground score → key, histogram of the top 12 bits
// probs are (A,2) pairs; the object probability is the odd element (ProposalLayer.swift:124).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_keys_hist0(const float* __restrict__ probs, long probs_sB, int A,
                                                    uint32_t* __restrict__ keys, long keys_sB,
                                                    uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[4096];
    const int b = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) lh[i] = 0;
    __syncthreads();
    const float* p = probs + (size_t)b * probs_sB;
    const int i0 = blockIdx.x * CHUNK + t * 4;
    uint32_t k[4];
    if (i0 + 3 < A && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + (size_t)i0 * 2);
        const float4 v1 = *reinterpret_cast<const float4*>(p + (size_t)i0 * 2 + 4);
        k[0] = order_key(v0.y); k[1] = order_key(v0.w); k[2] = order_key(v1.y); k[3] = order_key(v1.w);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) k[j] = (i0 + j < A) ? order_key(p[(size_t)(i0 + j) * 2 + 1]) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (i0 + j < A) atomicAdd(&lh[k[j] >> 20], 1u);
    *reinterpret_cast<uint4*>(keys + (size_t)b * keys_sB + i0) = make_uint4(k[0], k[1], k[2], k[3]);
    __syncthreads();
    uint32_t* gh = hist + (size_t)b * 3 * 4096;
    for (int i = t; i < 4096; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// ------------------------------------------------------------------------------------------------
// scan of a pass histogram from the top: finds the bin holding the k-th largest key
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_select_scan(uint32_t* __restrict__ hist, uint32_t* __restrict__ state,
                                                     int pass, int K)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const int nbins = pass == 2 ? 256 : 4096, per = nbins / 256, bits = pass == 2 ? 8 : 12;
    const uint32_t* h = hist + ((size_t)b * 3 + pass) * 4096;
    uint32_t* st = state + (size_t)b * 8;
    uint32_t s = 0;
    for (int i = 0; i < per; ++i) s += h[t * per + i];
    // above[t] = keys in the groups above group t (suffix sum over the 256 groups: wave scan from the top + wave totals);
    // the group holding the k-th largest key is the one with above < k <= above + own — found by its own thread
    // (a single thread walking the groups from the top cost 12 us per pass)
    __shared__ uint32_t wtot[4];
    __shared__ int s_g;
    const int lane = t & 63, wv = t >> 6;
    uint32_t inc = s;                                  // inclusive suffix sum inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_down(inc, o);
        if (lane + o < 64) inc += v;
    }
    if (lane == 0) wtot[wv] = inc;
    if (t == 0) s_g = 0;                              // group 0 takes what no group above it holds
    __syncthreads();
    uint32_t above_t = inc - s;
    for (int w = wv + 1; w < 4; ++w) above_t += wtot[w];
    const uint32_t k = pass == 0 ? (uint32_t)K : st[ST_KREM];
    if (t > 0 && above_t < k && above_t + s >= k) s_g = t;         // at most one group qualifies (k >= 1)
    __syncthreads();
    if (t == s_g) {
        const int g = t;
        uint32_t above = above_t;
        int bin = (g + 1) * per - 1;
        for (; bin > g * per; --bin) {
            if (above + h[bin] >= k) break;
            above += h[bin];
        }
        const uint32_t prefix = pass == 0 ? 0u : st[ST_PREFIX];
        st[ST_PREFIX] = (prefix << bits) | (uint32_t)bin;
        st[ST_KREM] = k - above;
        if (pass == 2) {
            st[ST_T] = (prefix << bits) | (uint32_t)bin;
            st[ST_NEED] = k - above;                  // how many keys equal to T are taken (lowest indices)
            st[ST_NGT] = (uint32_t)K - (k - above);   // keys strictly greater than T
            st[ST_TIEBIN] = (uint32_t)bin;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// passes 1/2: histogram of the next digit among keys matching the prefix found so far
// ------------------------------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void k_hist_pass(const uint32_t* __restrict__ keys, long keys_sB, int A,
                                                   const uint32_t* __restrict__ state, uint32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ blockhist, int nblk)
{
    constexpr int NB = PASS == 1 ? 4096 : 256;
    __shared__ uint32_t lh[NB];
    const int b = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < NB; i += 256) lh[i] = 0;
    __syncthreads();
    const uint32_t prefix = state[(size_t)b * 8 + ST_PREFIX];
    const int i0 = blockIdx.x * CHUNK + t * 4;
    const uint4 kv = *reinterpret_cast<const uint4*>(keys + (size_t)b * keys_sB + i0);
    const uint32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j >= A) continue;
        if (PASS == 1) {
            if ((k[j] >> 20) == prefix) atomicAdd(&lh[(k[j] >> 8) & 0xFFFu], 1u);
        } else {
            if ((k[j] >> 8) == prefix) atomicAdd(&lh[k[j] & 0xFFu], 1u);
        }
    }
    __syncthreads();
    uint32_t* gh = hist + ((size_t)b * 3 + PASS) * 4096;
    for (int i = t; i < NB; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
    if (PASS == 2) blockhist[((size_t)b * nblk + blockIdx.x) * 256 + t] = lh[t];
}

// ------------------------------------------------------------------------------------------------
// compaction: keys > T in any order, keys == T in index order (first `need` of them)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compact(const uint32_t* __restrict__ keys, long keys_sB, int A,
                                                 uint32_t* __restrict__ state, const uint32_t* __restrict__ blockhist,
                                                 int nblk, uint64_t* __restrict__ cand, int Kpad)
{
    __shared__ uint32_t red[256];
    __shared__ uint32_t wsum[4], wsumg[4];
    __shared__ uint32_t s_gbase;
    const int b = blockIdx.y, t = threadIdx.x, blk = blockIdx.x;
    uint32_t* st = state + (size_t)b * 8;
    const uint32_t T = st[ST_T], need = st[ST_NEED], ngt = st[ST_NGT], tiebin = st[ST_TIEBIN];
    // ties in the blocks before this one
    uint32_t s = 0;
    for (int i = t; i < blk; i += 256) s += blockhist[((size_t)b * nblk + i) * 256 + tiebin];
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    const uint32_t tie_base = red[0];
    const int i0 = blk * CHUNK + t * 4;
    const uint4 kv = *reinterpret_cast<const uint4*>(keys + (size_t)b * keys_sB + i0);
    const uint32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
    uint32_t nt = 0, ng = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        nt += (i0 + j < A && k[j] == T) ? 1u : 0u;
        ng += (i0 + j < A && k[j] > T) ? 1u : 0u;
    }
    // exclusive scans of nt (ties, in index order) and ng (keys above T) over the 256 threads (wave scan + 4 wave totals)
    const int lane = t & 63, wv = t >> 6;
    uint32_t inc = nt, incg = ng;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o), vg = __shfl_up(incg, o);
        if (lane >= o) { inc += v; incg += vg; }
    }
    if (lane == 63) { wsum[wv] = inc; wsumg[wv] = incg; }
    __syncthreads();
    uint32_t wbase = 0, wbaseg = 0;
    for (int i = 0; i < wv; ++i) { wbase += wsum[i]; wbaseg += wsumg[i]; }
    // ONE reservation per block for its keys above T (their order is free: they are sorted next) — one atomic per key
    // serialised 6 000 updates of a single word per image (56 us)
    if (t == 0) {
        const uint32_t total = wsumg[0] + wsumg[1] + wsumg[2] + wsumg[3];
        s_gbase = total ? atomicAdd(&st[ST_SLOT], total) : 0u;
    }
    __syncthreads();
    uint32_t rank = tie_base + wbase + inc - nt;
    uint32_t slot = s_gbase + wbaseg + incg - ng;
    uint64_t* c = cand + (size_t)b * Kpad;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (i0 + j >= A) continue;
        const uint64_t v = ((uint64_t)(~k[j]) << 32) | (uint32_t)(i0 + j);
        if (k[j] > T) {
            c[slot++] = v;
        } else if (k[j] == T) {
            if (rank < need) c[ngt + rank] = v;
            ++rank;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bitonic sort of the K candidates (64-bit keys: score key << 32 | anchor index, all distinct), then gather + decode + clip.
// One 1024-thread block per image.  k_sort_decode<E>: Kpad = 1024·E, thread t owns the E consecutive elements t·E ..:
// a compare-exchange distance below E stays in the thread's registers, below 64·E inside the wave (lane shuffle), and only
// the distances from 64·E up (10 of the 91 stages at Kpad = 8192) go through LDS and a block barrier — the all-LDS form
// (kept for Kpad < 1024) spent 138 us of its stages' 91 barriers on 8 of the chip's 256 CUs.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_sorted(const uint64_t key, int i, int b, int K, const float* dl, const float* anchors, float4 stdv,
                                              int32_t* topk_idx, float* boxes)
{
    const uint32_t idx = (uint32_t)(key & 0xFFFFFFFFull);
    topk_idx[(size_t)b * K + i] = (int32_t)idx;
    float4 d = *reinterpret_cast<const float4*>(dl + (size_t)idx * 4);
    const float4 an = *reinterpret_cast<const float4*>(anchors + (size_t)idx * 4);
    d.x = d.x * stdv.x; d.y = d.y * stdv.y; d.z = d.z * stdv.z; d.w = d.w * stdv.w;   // Utils.swift:173-180
    *reinterpret_cast<float4*>(boxes + ((size_t)b * K + i) * 4) = decode_clip_box(an, d);
}

template <int E>
__global__ __launch_bounds__(1024) void k_sort_decode(const uint64_t* __restrict__ cand, int K, int Kpad,
                                                      const float* __restrict__ deltas, long deltas_sB,
                                                      const float* __restrict__ anchors, float4 stdv,
                                                      int32_t* __restrict__ topk_idx, float* __restrict__ boxes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = reinterpret_cast<uint64_t*>(smem);          // [E][1024]: element (t, r) at r·1024 + t (conflict-free columns)
    const int b = blockIdx.x, t = threadIdx.x;
    const uint64_t* c = cand + (size_t)b * Kpad;
    uint64_t v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) { const int i = t * E + r; v[r] = i < K ? c[i] : ~0ull; }
    for (int k = 2; k <= Kpad; k <<= 1) {
        int j = k >> 1;
        for (; j >= 64 * E; j >>= 1) {                         // partner in another wave: through LDS
#pragma unroll
            for (int r = 0; r < E; ++r) a[r * 1024 + t] = v[r];
            __syncthreads();
            const int tp = t ^ (j / E);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint64_t y = a[r * 1024 + tp];
                const int i = t * E + r;
                const bool take_min = ((i & j) == 0) == ((i & k) == 0);
                v[r] = take_min ? (v[r] < y ? v[r] : y) : (v[r] > y ? v[r] : y);
            }
            __syncthreads();
        }
        for (; j >= E; j >>= 1) {                              // partner in another lane of this wave
            const int d = j / E;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint32_t ylo = __shfl_xor((uint32_t)v[r], d), yhi = __shfl_xor((uint32_t)(v[r] >> 32), d);
                const uint64_t y = ((uint64_t)yhi << 32) | ylo;
                const int i = t * E + r;
                const bool take_min = ((i & j) == 0) == ((i & k) == 0);
                v[r] = take_min ? (v[r] < y ? v[r] : y) : (v[r] > y ? v[r] : y);
            }
        }
        // partner in this thread's registers: the distance is a compile-time constant once unrolled (a run-time register
        // index would send v[] to scratch memory)
#pragma unroll
        for (int jj = E >> 1; jj > 0; jj >>= 1) {
            if (jj <= (k >> 1)) {
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    if ((r & jj) == 0) {
                        const int i = t * E + r;
                        const bool up = (i & k) == 0;
                        const uint64_t x = v[r], y = v[r | jj];
                        const bool sw = (x > y) == up;
                        v[r] = sw ? y : x;
                        v[r | jj] = sw ? x : y;
                    }
                }
            }
        }
    }
    const float* dl = deltas + (size_t)b * deltas_sB;
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const int i = t * E + r;
        if (i < K) decode_sorted(v[r], i, b, K, dl, anchors, stdv, topk_idx, boxes);
    }
}

__global__ __launch_bounds__(1024) void k_sort_decode_lds(const uint64_t* __restrict__ cand, int K, int Kpad,
                                                          const float* __restrict__ deltas, long deltas_sB,
                                                          const float* __restrict__ anchors, float4 stdv,
                                                          int32_t* __restrict__ topk_idx, float* __restrict__ boxes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = reinterpret_cast<uint64_t*>(smem);
    const int b = blockIdx.x, t = threadIdx.x;
    const uint64_t* c = cand + (size_t)b * Kpad;
    for (int i = t; i < Kpad; i += 1024) a[i] = i < K ? c[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= Kpad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < Kpad; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    const float* dl = deltas + (size_t)b * deltas_sB;
    for (int i = t; i < K; i += 1024) decode_sorted(a[i], i, b, K, dl, anchors, stdv, topk_idx, boxes);
}

// ------------------------------------------------------------------------------------------------
// The same order by RANK COUNTING on the whole chip (round 5): the K candidate keys are distinct (score key << 32 | anchor
// index), so the position of key i in the sorted order is the number of keys below it — K x K compares with no dependency
// between them, against the 91 dependent stages of the bitonic network in one block per image (91 us on 1 .. 8 of the chip's
// 256 CUs whatever the batch).  A block owns 64 keys (one per lane); its four waves each count over a quarter of every
// 2048-key chunk staged in LDS (wave-uniform addresses: broadcast reads), and wave 0 adds the four counts and writes the
// decoded box at its rank.  Same keys, same order, same decode: the outputs are the bitonic kernel's, bit for bit
// (tests/test_gpu_boxes.py; "proposal_rank_sort" 0 / MRCNN_RANK_SORT=0 keeps the old kernel).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rank_decode(const uint64_t* __restrict__ cand, int K, int Kpad,
                                                     const float* __restrict__ deltas, long deltas_sB,
                                                     const float* __restrict__ anchors, float4 stdv,
                                                     int32_t* __restrict__ topk_idx, float* __restrict__ boxes)
{
    constexpr int CH = 2048;
    __shared__ __attribute__((aligned(16))) uint64_t sk[CH];
    __shared__ int s_cnt[4][64];
    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(t >> 6));
    const uint64_t* c = cand + (size_t)b * Kpad;
    const int i = blockIdx.x * 64 + lane;
    const uint64_t mine = i < K ? c[i] : 0ull;
    int cnt = 0;
    for (int base = 0; base < K; base += CH) {
        if (base) __syncthreads();
#pragma unroll
        for (int e = t; e < CH; e += 256) sk[e] = base + e < K ? c[base + e] : ~0ull;        // (padding: never below a key)
        __syncthreads();
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(sk + wv * (CH / 4));
#pragma unroll 8
        for (int j = 0; j < CH / 8; ++j) {
            const ulonglong2 v = p[j];
            cnt += (v.x < mine ? 1 : 0) + (v.y < mine ? 1 : 0);
        }
    }
    s_cnt[wv][lane] = cnt;
    __syncthreads();
    if (wv == 0 && i < K) {
        const int rank = s_cnt[0][lane] + s_cnt[1][lane] + s_cnt[2][lane] + s_cnt[3][lane];
        decode_sorted(mine, rank, b, K, deltas + (size_t)b * deltas_sB, anchors, stdv, topk_idx, boxes);
    }
}

// ------------------------------------------------------------------------------------------------
// NMS part 1: suppression bit-matrix.  Bit j of mask[i][cb] ⇔ column box (cb*64+j) > i, same class, IoU > thr; only the
// words cb >= (i / 64) are computed.  Block (rb, split) = 4 waves holding the 64 rows rb*64.. in registers (one row per
// lane); wave w of split s walks the column chunks cb = rb + 4 s + w, + 4·gridDim.y, ... (a wave-private LDS slot holds
// the chunk's 64 boxes: no block barrier).  Per chunk two phases:
//   1. every (row, column) pair through the exact FLOAT disjointness test (6 VALU per pair on pre-sorted corners,
//      column box broadcast from LDS) → a 64-bit candidate word per row;
//   2. each lane walks ITS candidates only (ctz loop) through the fp64 IoU decision (iou_exceeds: the reference's
//      Double arithmetic on per-box extents computed once, the division only in borderline cases).
// Round 1 ran one 64-thread block per (rb, cb) pair and sent EVERY column through the fp64 IoU with its division whenever
// one of the 64 rows overlapped it (almost always, with a handful of lanes active): 256 us per batch of 8 x 6000 boxes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, long boxes_sB,
                                                  const int32_t* __restrict__ cls, long cls_sB,
                                                  const int32_t* __restrict__ n_dev, int n_const, float thr,
                                                  uint64_t* __restrict__ mask, long mask_sB, int W)
{
    __shared__ double s_ext[4][5][64];   // standardized extents + area of the wave's column chunk (RectExt, one array per field)
    __shared__ float4 s_srt[4][64];      // (min y, min x, max y, max x)
    __shared__ int32_t s_cls[4][64];
    const int rb = blockIdx.x, b = blockIdx.z, lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = n_dev ? n_dev[b] : n_const;
    if (rb * 64 >= n) return;
    const int nW = (n + 63) / 64;
    const float* bx = boxes + (size_t)b * boxes_sB;
    const int i = rb * 64 + lane;
    const bool in_i = i < n;
    const float4 me = in_i ? *reinterpret_cast<const float4*>(bx + (size_t)i * 4) : make_float4(0, 0, 0, 0);
    const float4 ms = make_float4(fminf(me.x, me.z), fminf(me.y, me.w), fmaxf(me.x, me.z), fmaxf(me.y, me.w));
    const RectExt me_ext = rect_ext(me);
    const int mycls = (cls && in_i) ? cls[(size_t)b * cls_sB + i] : 0;
    const bool pretest = thr >= 0.0f;     // IOU() == 0 > thr holds for a caller-supplied thr < 0: then every pair is a candidate
    const double thr_mid = iou_threshold_midpoint(thr);
    const int stride = 4 * (int)gridDim.y;
    for (int cb = rb + 4 * (int)blockIdx.y + wv; cb < nW; cb += stride) {
        const int cj = cb * 64 + lane;
        const float4 cv = cj < n ? *reinterpret_cast<const float4*>(bx + (size_t)cj * 4) : make_float4(0, 0, 0, 0);
        {
            const RectExt ce = rect_ext(cv);
            s_ext[wv][0][lane] = ce.minx; s_ext[wv][1][lane] = ce.maxx; s_ext[wv][2][lane] = ce.miny; s_ext[wv][3][lane] = ce.maxy;
            s_ext[wv][4][lane] = ce.area;
        }
        s_srt[wv][lane] = make_float4(fminf(cv.x, cv.z), fminf(cv.y, cv.w), fmaxf(cv.x, cv.z), fmaxf(cv.y, cv.w));
        if (cls) s_cls[wv][lane] = cj < n ? cls[(size_t)b * cls_sB + cj] : 0;
        const int jn = min(64, n - cb * 64);
        // phase 1 — exact float pre-test: disjoint (or merely touching) boxes have intersection 0, hence IoU 0
        uint64_t cand = 0;
        for (int j = 0; j < jn; ++j) {
            const float4 c = s_srt[wv][j];
            bool hit = !pretest || !(fminf(c.z, ms.z) <= fmaxf(c.x, ms.x) || fminf(c.w, ms.w) <= fmaxf(c.y, ms.y));
            if (cls) hit = hit && s_cls[wv][j] == mycls;
            cand |= hit ? 1ull << j : 0ull;
        }
        if (cb == rb) cand &= ~((2ull << lane) - 1ull);        // only columns behind the row itself (gj > i)
        // phase 2 — the reference's fp64 IoU on the candidates
        uint64_t bits = 0;
        while (cand != 0ull) {
            const int j = __builtin_ctzll(cand);
            cand &= cand - 1;
            const RectExt ce = {s_ext[wv][0][j], s_ext[wv][1][j], s_ext[wv][2][j], s_ext[wv][3][j], s_ext[wv][4][j]};
            if (iou_exceeds(ce, me_ext, thr, thr_mid)) bits |= 1ull << j;            // IOU(anchorA = candidate, anchorB = selected)
        }
        if (in_i) mask[(size_t)b * mask_sB + (size_t)i * W + cb] = bits;
    }
}

// ------------------------------------------------------------------------------------------------
// NMS part 2: greedy scan (nonMaxSupression, Utils.swift:185-218).  One 256-thread block per image,
// candidates in chunks of 64.  For chunk c the "already suppressed" word is the OR of word c of the
// rows kept so far — up to max_keep independent 8-B loads spread over the block, ONE memory latency
// per chunk (an incremental removed[] array needed up to 64 dependent row sweeps per chunk and made
// the scan 8x slower), requested one chunk AHEAD (see the loop).  Wave 0 then resolves the chunk from
// the diagonal word with ballot/readlane.
// per_class_max > 0: candidates carry classes; a class stops selecting after per_class_max keeps
// (each class is its own nonMaxSupression call in DetectionLayer.swift:170-183).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nms_scan(const float* __restrict__ boxes, long boxes_sB,
                                                  const int32_t* __restrict__ cls, long cls_sB,
                                                  const int32_t* __restrict__ n_dev, int n_const,
                                                  const uint64_t* __restrict__ mask, long mask_sB, int W,
                                                  int max_keep, int per_class_max,
                                                  int32_t* __restrict__ keep_idx, long keep_sB,
                                                  int32_t* __restrict__ keep_count, int class_fast)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int32_t* kept = reinterpret_cast<int32_t*>(smem);                           // [max_keep] rows kept so far
    int32_t* kept_cls = reinterpret_cast<int32_t*>(smem + (size_t)max_keep * 4);  // [max_keep] (per-class mode)
    constexpr int NCLS = 1024;
    __shared__ int cls_cnt[NCLS];            // keeps per class id < NCLS (larger ids: counted from kept_cls)
    __shared__ uint64_t s_part[4];
    __shared__ int s_kc;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int n = n_dev ? n_dev[b] : n_const;
    const int nW = (n + 63) / 64;
    if (per_class_max > 0)
        for (int i = t; i < NCLS; i += 256) cls_cnt[i] = 0;
    if (t == 0) s_kc = 0;
    __syncthreads();
    const float* bx = boxes + (size_t)b * boxes_sB;
    const uint64_t* mk = mask + (size_t)b * mask_sB;
    int32_t* kidx = keep_idx + (size_t)b * keep_sB;
    // The two rendezvous of a chunk exchange LDS data only (s_part, s_kc, kept[]): a plain barrier behind an LDS wait —
    // __syncthreads() would also drain vmcnt, i.e. wait for the requests issued ahead for the next chunk.
#define NMS_LDS_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // Software pipeline: everything chunk c+1 needs from memory is requested BEFORE chunk c is resolved — word c+1 of the rows
    // kept up to chunk c-1 (spread over the block), and, in wave 0, word c+1 and the diagonal word of the 64 rows of chunks c
    // and c+1 themselves (which of chunk c's rows get kept is only known after the resolve: their words are fetched for all
    // 64 and OR-ed in under the kept mask).  The memory latency of a chunk hides behind the previous chunk's resolve.
    uint64_t part = 0;          // this thread's share of OR_{rows kept before chunk c-1 was resolved} word c
    uint64_t prev_w = 0;        // wave 0: word c of row (c-1)*64 + lane
    uint64_t prev_kept = 0;     // wave 0 (uniform): which rows of chunk c-1 were kept
    uint64_t diag = (wv == 0 && lane < n) ? mk[(size_t)lane * W] : 0ull;       // wave 0: word c of row c*64 + lane
    for (int c = 0; c < nW; ++c) {
        const int kc0 = s_kc;
        uint64_t part_n = 0, prev_w_n = 0, diag_n = 0;
        if (c + 1 < nW) {
            for (int k = t; k < kc0; k += 256) part_n |= mk[(size_t)kept[k] * W + c + 1];
            if (wv == 0) {
                const int row = c * 64 + lane, rown = row + 64;
                if (row < n) prev_w_n = mk[(size_t)row * W + c + 1];
                if (rown < n) diag_n = mk[(size_t)rown * W + c + 1];
            }
        }
        uint32_t lo = (uint32_t)part, hi = (uint32_t)(part >> 32);
        if (wv == 0 && ((prev_kept >> lane) & 1ull)) { lo |= (uint32_t)prev_w; hi |= (uint32_t)(prev_w >> 32); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo |= __shfl_xor(lo, o); hi |= __shfl_xor(hi, o); }
        if (lane == 0) s_part[wv] = ((uint64_t)hi << 32) | lo;
        NMS_LDS_BARRIER
        if (wv == 0) {
            const uint64_t removed = s_part[0] | s_part[1] | s_part[2] | s_part[3];
            const int row = c * 64 + lane;
            const bool inr = row < n;
            const bool ok = inr && rect_selectable(*reinterpret_cast<const float4*>(bx + (size_t)(inr ? row : 0) * 4));
            const int mycls = (cls && inr) ? cls[(size_t)b * cls_sB + row] : 0;
            // The resolve is a serial chain over the chunk's surviving candidates: kept on the SCALAR unit (wave-uniform mask,
            // s_ff1, v_readlane for the picked lane's diagonal word / class / class count) — a cross-lane __shfl per step
            // (ds_bpermute: an LDS round trip) and an LDS read-modify-write of the class counter made a detection chunk cost 9 us.
            // Class counts of the chunk's own classes live in registers (every lane carries the count of ITS class, all lanes of a
            // class updated together) and go back to LDS once per chunk.
            uint64_t m = __ballot(ok) & ~removed;
            m = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) |
                (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m);          // (the builtin returns a SIGNED int)
            const uint32_t dlo_v = (uint32_t)(diag & 0xFFFFFFFFull), dhi_v = (uint32_t)(diag >> 32);
            const bool small_cls = (unsigned)mycls < (unsigned)NCLS;
            int mycnt = (per_class_max > 0 && inr && small_cls) ? cls_cnt[mycls] : 0;
            uint64_t km = 0;            // the chunk's kept lanes; their indices are written once, in parallel, after the resolve
            int kc = kc0;
            // Wave-parallel resolve (the common case): the greedy rule "candidate i is kept iff no KEPT earlier candidate of the
            // chunk suppresses it" is evaluated in rounds over the whole chunk at once.  P = the transposed 64x64 diagonal block
            // (lane i: the alive earlier candidates that would suppress i — six shuffle-exchange stages); per round every
            // undecided candidate whose potential suppressors are all decided settles (kept iff none of them is kept): the number
            // of rounds is the depth of the longest suppression chain in the chunk (typically 2-4) instead of one serial step per
            // kept candidate.  The per-class limit couples candidates through a running count: chunks in which a class could reach
            // it (count + 64 > limit), or with class ids beyond the counter table, take the serial loop below.
            // (round 5: a class can gain at most as many keeps as it has ALIVE candidates in this chunk — the round-4 test assumed 64 of them and
            //  sent every chunk behind a class's 36th keep down the serial loop: 5 us per chunk of a 1000-row detection scan)
            int gain = 64;
            if (class_fast && per_class_max > 0) {
                // candidates of a class that already holds its limit are refused whatever happens (and a refused candidate suppresses nothing)
                m &= ~__ballot(inr && small_cls && mycnt >= per_class_max);
                gain = 0;
                uint64_t todo = m;
                while (todo != 0ull) {                              // one step per distinct class among the alive candidates
                    const int l = __builtin_ctzll(todo);
                    const int c = __builtin_amdgcn_readlane(mycls, l);
                    const uint64_t eq = __ballot(mycls == c) & m;
                    if (mycls == c) gain = __popcll(eq);
                    todo &= ~eq;
                }
            }
            const bool fast = per_class_max <= 0 || __ballot(inr && ((m >> lane) & 1ull || !class_fast) && (!small_cls || mycnt + gain > per_class_max)) == 0ull;
            if (fast) {
                const bool alive = (m >> lane) & 1ull;
                uint64_t P = alive ? diag : 0ull;
                // 64x64 bit-matrix transpose across the lanes: stage s swaps the off-diagonal s x s blocks of every 2s x 2s block
                // (cm = the columns c with (c & s) != 0)
#define NMS_TRANSPOSE_STAGE(S, CM)                                                                             \
                {                                                                                              \
                    const uint32_t ylo = __shfl_xor((uint32_t)P, S), yhi = __shfl_xor((uint32_t)(P >> 32), S); \
                    const uint64_t y = ((uint64_t)yhi << 32) | ylo;                                            \
                    P = (lane & S) ? ((P & CM) | ((y >> S) & ~CM)) : ((P & ~CM) | ((y << S) & CM));            \
                }
                NMS_TRANSPOSE_STAGE(32, 0xFFFFFFFF00000000ull) NMS_TRANSPOSE_STAGE(16, 0xFFFF0000FFFF0000ull)
                NMS_TRANSPOSE_STAGE(8, 0xFF00FF00FF00FF00ull) NMS_TRANSPOSE_STAGE(4, 0xF0F0F0F0F0F0F0F0ull)
                NMS_TRANSPOSE_STAGE(2, 0xCCCCCCCCCCCCCCCCull) NMS_TRANSPOSE_STAGE(1, 0xAAAAAAAAAAAAAAAAull)
#undef NMS_TRANSPOSE_STAGE
                uint64_t und = m, kept_m = 0;
                while (und != 0ull) {
                    const bool und_i = (und >> lane) & 1ull;
                    const bool ready = und_i && (P & und) == 0ull;
                    const bool keep_i = ready && (P & kept_m) == 0ull;
                    const uint64_t nk = __ballot(keep_i), nd = __ballot(ready);
                    kept_m |= nk;
                    und &= ~nd;
                }
                const int room = max_keep - kc0;                      // > 0: the chunk loop ends once max_keep is reached
                const bool mine = (kept_m >> lane) & 1ull;
                km = __ballot(mine && __popcll(kept_m & ((1ull << lane) - 1ull)) < room);
                kc = kc0 + __popcll(km);
                if (per_class_max > 0 && ((km >> lane) & 1ull)) atomicAdd(&cls_cnt[mycls], 1);      // small_cls holds on this path
            } else {
                while (m != 0ull && kc < max_keep) {
                    const int i = __builtin_ctzll(m);
                    m &= m - 1;
                    bool take = true;
                    if (per_class_max > 0) {
                        const int ci = __builtin_amdgcn_readlane(mycls, i);
                        int cnt;
                        if ((unsigned)ci < (unsigned)NCLS) cnt = __builtin_amdgcn_readlane(mycnt, i);
                        else {
                            cnt = __popcll(__ballot(mycls == ci) & km);              // kept in this chunk
                            for (int base = 0; base < kc0; base += 64) {            // ... and in the earlier ones
                                const bool eq = (base + lane < kc0) && kept_cls[base + lane] == ci;
                                cnt += __popcll(__ballot(eq));
                            }
                        }
                        take = cnt < per_class_max;
                        if (take && mycls == ci) ++mycnt;
                    }
                    if (take) {
                        ++kc;
                        km |= 1ull << i;
                        const uint32_t dlo = __builtin_amdgcn_readlane(dlo_v, i), dhi = __builtin_amdgcn_readlane(dhi_v, i);
                        m &= ~(((uint64_t)dhi << 32) | dlo);
                    }
                }
                if (per_class_max > 0 && inr && small_cls) cls_cnt[mycls] = mycnt;      // lanes of one class write the same value
            }
            if ((km >> lane) & 1ull) {
                const int slot = kc0 + __popcll(km & ((1ull << lane) - 1ull));
                kidx[slot] = row;
                kept[slot] = row;
                if (per_class_max > 0) kept_cls[slot] = mycls;
            }
            prev_kept = km;
            if (lane == 0) s_kc = kc;
        }
        NMS_LDS_BARRIER
        if (s_kc >= max_keep) break;
        part = part_n; prev_w = prev_w_n; diag = diag_n;
    }
#undef NMS_LDS_BARRIER
    if (t == 0) keep_count[b] = s_kc;
}

// ------------------------------------------------------------------------------------------------
// ProposalLayer output copy + zero padding (ProposalLayer.swift:178-192)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_write_rois(const float* __restrict__ boxes, long boxes_sB,
                                                    const int32_t* __restrict__ keep_idx, long keep_sB,
                                                    const int32_t* __restrict__ keep_count, int max_keep,
                                                    float* __restrict__ rois, long rois_sB, long row_stride)
{
    const int b = blockIdx.x;
    const int nk = keep_count[b];
    float* o = rois + (size_t)b * rois_sB;
    const float* bx = boxes + (size_t)b * boxes_sB;
    for (long e = threadIdx.x; e < (long)max_keep * row_stride; e += 256) {
        const long i = e / row_stride, j = e % row_stride;
        float v = 0.0f;
        if (i < nk) {
            if (j < 4) v = bx[(size_t)keep_idx[(size_t)b * keep_sB + i] * 4 + j];
            else continue;    // the reference writes only the 4 coordinates of kept rows
        }
        o[e] = v;
    }
}

void proposal_forward(hipStream_t s, const ProposalWorkspace& ws, const float* probs, long probs_sB,
                      const float* deltas, long deltas_sB, const float* anchors, const float std4[4],
                      float nms_thr, float* rois, long rois_sB, long row_stride)
{
    const int B = ws.B, A = ws.A, K = ws.K;
    const long keys_sB = (long)ws.nblk * CHUNK;
    HIP_CHECK(hipMemsetAsync(ws.hist, 0, (size_t)B * 3 * 4096 * 4, s));
    HIP_CHECK(hipMemsetAsync(ws.state, 0, (size_t)B * 8 * 4, s));
    dim3 g(ws.nblk, B);
    // the roctx ranges carry the reference's signpost names for the step each group of launches replaces (ProposalLayer.swift:122-186)
    trace_push("Proposal-StridedSlice");     // foreground scores -> order keys (fused with the first histogram)
    hipLaunchKernelGGL(k_keys_hist0, g, dim3(256), 0, s, probs, probs_sB, A, ws.keys, keys_sB, ws.hist);
    trace_pop();
    trace_push("Proposal-Sorting");          // radix select of the top K + sort of the survivors
    hipLaunchKernelGGL(k_select_scan, dim3(B), dim3(256), 0, s, ws.hist, ws.state, 0, K);
    hipLaunchKernelGGL(k_hist_pass<1>, g, dim3(256), 0, s, ws.keys, keys_sB, A, ws.state, ws.hist, ws.blockhist, ws.nblk);
    hipLaunchKernelGGL(k_select_scan, dim3(B), dim3(256), 0, s, ws.hist, ws.state, 1, K);
    hipLaunchKernelGGL(k_hist_pass<2>, g, dim3(256), 0, s, ws.keys, keys_sB, A, ws.state, ws.hist, ws.blockhist, ws.nblk);
    hipLaunchKernelGGL(k_select_scan, dim3(B), dim3(256), 0, s, ws.hist, ws.state, 2, K);
    hipLaunchKernelGGL(k_compact, g, dim3(256), 0, s, ws.keys, keys_sB, A, ws.state, ws.blockhist, ws.nblk, ws.cand, ws.Kpad);
    const size_t sort_lds = (size_t)ws.Kpad * 8;
    MRCNN_REQUIRE(sort_lds <= 160 * 1024 - 1024, MRCNN_ERR_UNSUPPORTED,
                  "preNMSMaxProposals %d exceeds the in-LDS sort capacity (max 16384)", K);
    boxes_one_time_init();
    const float4 stdv = make_float4(std4[0], std4[1], std4[2], std4[3]);
#define MRCNN_SORT(KERNEL) hipLaunchKernelGGL(KERNEL, dim3(B), dim3(1024), sort_lds, s, ws.cand, K, ws.Kpad, deltas, deltas_sB, anchors, stdv, ws.topk_idx, ws.boxes)
    if (g_rank_sort)
        hipLaunchKernelGGL(k_rank_decode, dim3((K + 63) / 64, B), dim3(256), 0, s, ws.cand, K, ws.Kpad, deltas, deltas_sB, anchors, stdv, ws.topk_idx, ws.boxes);
    else
    switch (ws.Kpad / 1024) {
    case 1: MRCNN_SORT(k_sort_decode<1>); break;
    case 2: MRCNN_SORT(k_sort_decode<2>); break;
    case 4: MRCNN_SORT(k_sort_decode<4>); break;
    case 8: MRCNN_SORT(k_sort_decode<8>); break;
    case 16: MRCNN_SORT(k_sort_decode<16>); break;
    default: MRCNN_SORT(k_sort_decode_lds); break;          // Kpad < 1024
    }
#undef MRCNN_SORT
    trace_pop();                             // (k_sort_decode also covers Proposal-Gathering and Proposal-Compute: gather, x std, decode, clip)
    const long boxes_sB = (long)K * 4, mask_sB = (long)K * ws.W;
    trace_push("Proposal-NMS");
    hipLaunchKernelGGL(k_nms_mask, dim3(ws.W, nms_col_splits(ws.W, B), B), dim3(256), 0, s, ws.boxes, boxes_sB, (const int32_t*)nullptr, 0L,
                       (const int32_t*)nullptr, K, nms_thr, ws.nms_mask, mask_sB, ws.W);
    hipLaunchKernelGGL(k_nms_scan, dim3(B), dim3(256), (size_t)ws.max_keep * 4, s, ws.boxes, boxes_sB, (const int32_t*)nullptr, 0L,
                       (const int32_t*)nullptr, K, ws.nms_mask, mask_sB, ws.W, ws.max_keep, 0, ws.keep_idx,
                       (long)ws.max_keep, ws.keep_count, g_nms_fast);
    trace_pop();
    trace_push("Proposal-Copy");
    hipLaunchKernelGGL(k_write_rois, dim3(B), dim3(256), 0, s, ws.boxes, boxes_sB, ws.keep_idx, (long)ws.max_keep,
                       ws.keep_count, ws.max_keep, rois, rois_sB, row_stride);
    trace_pop();
    HIP_CHECK(hipGetLastError());
}

// ================================================================================================
// DetectionLayer
// ================================================================================================
size_t DetectionWorkspace::bytes(int B, int N, int /*max_det*/)
{
    const int W = (N + 63) / 64;
    size_t n = 0;
    n += align_up((size_t)B * 4, 256) * 2;              // count, keep_count
    n += align_up((size_t)B * N * 4, 256) * 4;          // src, score, cls, keep_idx
    n += align_up((size_t)B * N * 16, 256);             // boxes
    n += align_up((size_t)B * N * W * 8, 256);          // nms_mask
    return n;
}

void DetectionWorkspace::bind(void* base, int B_, int N_, int max_det_)
{
    B = B_; N = N_; max_det = max_det_; W = (N + 63) / 64; Npad = next_pow2(N);
    char* p = (char*)base;
    auto take = [&](size_t n) { char* r = p; p += align_up(n, 256); return r; };
    count = (int32_t*)take((size_t)B * 4);
    keep_count = (int32_t*)take((size_t)B * 4);
    src = (int32_t*)take((size_t)B * N * 4);
    score = (float*)take((size_t)B * N * 4);
    cls = (int32_t*)take((size_t)B * N * 4);
    keep_idx = (int32_t*)take((size_t)B * N * 4);
    boxes = (float*)take((size_t)B * N * 16);
    nms_mask = (uint64_t*)take((size_t)B * N * W * 8);
}

// score >= threshold (vDSP_vthres + vDSP_vcmprs, DetectionLayer.swift:238-276), classId > 0 (:136-140),
// order-preserving compaction, then ×std, applyBoxDeltas, clip (:156-164).
__global__ __launch_bounds__(1024) void k_det_filter_decode(const float* __restrict__ rois, long rois_sB, long roi_stride,
                                                            const float* __restrict__ cls6, long cls_sB, int N,
                                                            float4 stdv, float score_thr, int32_t* __restrict__ count,
                                                            int32_t* __restrict__ src, float* __restrict__ boxes,
                                                            float* __restrict__ score, int32_t* __restrict__ cls)
{
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const float* r = rois + (size_t)b * rois_sB;
    const float* c = cls6 + (size_t)b * cls_sB;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 1024) {
        const int i = i0 + t;
        bool keep = false;
        float sc = 0.0f, cid = 0.0f;
        if (i < N) {
            cid = c[(size_t)i * 6 + 4];
            sc = c[(size_t)i * 6 + 5];
            const float gated = sc >= score_thr ? sc : 0.0f;
            keep = gated != 0.0f && cid > 0.0f;
        }
        const uint64_t bal = __ballot(keep);
        const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(bal);
        __syncthreads();
        int base = s_base;
        for (int w = 0; w < wv; ++w) base += wsum[w];
        if (keep) {
            const int k = base + wpre;
            float4 box = make_float4(r[(size_t)i * roi_stride], r[(size_t)i * roi_stride + 1], r[(size_t)i * roi_stride + 2],
                                     r[(size_t)i * roi_stride + 3]);
            float4 d = make_float4(c[(size_t)i * 6], c[(size_t)i * 6 + 1], c[(size_t)i * 6 + 2], c[(size_t)i * 6 + 3]);
            d.x = d.x * stdv.x; d.y = d.y * stdv.y; d.z = d.z * stdv.z; d.w = d.w * stdv.w;
            *reinterpret_cast<float4*>(boxes + ((size_t)b * N + k) * 4) = decode_clip_box(box, d);
            src[(size_t)b * N + k] = i;
            score[(size_t)b * N + k] = sc;
            cls[(size_t)b * N + k] = (int32_t)cid;
        }
        __syncthreads();
        if (t == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wsum[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (t == 0) count[b] = s_base;
}

// Top maxDetections of the NMS survivors by score (DetectionLayer.swift:186-209; ties keep the
// nmsBoxIds order = class ascending, then ROI order), row write-out and zero padding (:211-231).
__global__ __launch_bounds__(1024) void k_det_finalize(const float* __restrict__ boxes, const float* __restrict__ score,
                                                       const int32_t* __restrict__ cls, const int32_t* __restrict__ keep_idx,
                                                       const int32_t* __restrict__ keep_count, int N, int Npad, int max_det,
                                                       float* __restrict__ out, long out_sB, long row_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* a = reinterpret_cast<uint64_t*>(smem);
    const int b = blockIdx.x, t = threadIdx.x;
    const int nk = keep_count[b];
    for (int i = t; i < Npad; i += 1024) {
        uint64_t v = ~0ull;
        if (i < nk) {
            const int k = keep_idx[(size_t)b * N + i];
            const uint32_t sk = ~order_key(score[(size_t)b * N + k]);
            v = ((uint64_t)sk << 32) | ((uint64_t)((uint32_t)cls[(size_t)b * N + k] & 0xFFFFu) << 16) | (uint32_t)(k & 0xFFFF);
        }
        a[i] = v;
    }
    __syncthreads();
    for (int k = 2; k <= Npad; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < Npad; i += 1024) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    const int nd = nk < max_det ? nk : max_det;
    float* o = out + (size_t)b * out_sB;
    for (long e = t; e < (long)max_det * row_stride; e += 1024) {
        const long i = e / row_stride, j = e % row_stride;
        float v = 0.0f;
        if (i < nd) {
            if (j >= 6) continue;
            const int k = (int)(a[i] & 0xFFFFull);
            if (j < 4) v = boxes[((size_t)b * N + k) * 4 + j];
            else if (j == 4) v = (float)cls[(size_t)b * N + k];
            else v = score[(size_t)b * N + k];
        }
        o[e] = v;
    }
}

void detection_forward(hipStream_t s, const DetectionWorkspace& ws, const float* rois, long rois_sB,
                       long roi_stride, const float* cls6, long cls_sB, const float std4[4],
                       float score_thr, float nms_thr, int /*num_classes_hint*/, float* out, long out_sB,
                       long row_stride)
{
    const int B = ws.B, N = ws.N;
    MRCNN_REQUIRE(N <= 65535, MRCNN_ERR_UNSUPPORTED, "DetectionLayer: more than 65535 regions (%d)", N);
    const float4 stdv = make_float4(std4[0], std4[1], std4[2], std4[3]);
    hipLaunchKernelGGL(k_det_filter_decode, dim3(B), dim3(1024), 0, s, rois, rois_sB, roi_stride, cls6, cls_sB, N, stdv,
                       score_thr, ws.count, ws.src, ws.boxes, ws.score, ws.cls);
    const long boxes_sB = (long)N * 4, mask_sB = (long)N * ws.W;
    hipLaunchKernelGGL(k_nms_mask, dim3(ws.W, ws.W >= 16 ? 4 : 1, B), dim3(256), 0, s, ws.boxes, boxes_sB, ws.cls, (long)N, ws.count, 0,
                       nms_thr, ws.nms_mask, mask_sB, ws.W);
    boxes_one_time_init();
    const size_t scan_lds = (size_t)N * 8;
    MRCNN_REQUIRE(scan_lds <= 64 * 1024, MRCNN_ERR_UNSUPPORTED, "DetectionLayer: too many regions (%d)", N);
    hipLaunchKernelGGL(k_nms_scan, dim3(B), dim3(256), scan_lds, s, ws.boxes, boxes_sB, ws.cls, (long)N, ws.count, 0,
                       ws.nms_mask, mask_sB, ws.W, N, ws.max_det, ws.keep_idx, (long)N, ws.keep_count, g_nms_fast);
    hipLaunchKernelGGL(k_det_finalize, dim3(B), dim3(1024), (size_t)ws.Npad * 8, s, ws.boxes, ws.score, ws.cls, ws.keep_idx,
                       ws.keep_count, N, ws.Npad, ws.max_det, out, out_sB, row_stride);
    HIP_CHECK(hipGetLastError());
}

// Large dynamic-LDS opt-ins of the box kernels, once per process (also called at model / layer creation so that the first
// predict may already run inside a caller's stream capture).
void boxes_one_time_init()
{
    // per device (the attribute lives on the device's code object) and safe against concurrent first callers
    static std::mutex mu;
    static bool done[64] = {};
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (dev >= 0 && dev < 64 && done[dev]) return;
    HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_decode<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_sort_decode<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_nms_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_det_finalize, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    if (dev >= 0 && dev < 64) done[dev] = true;
}

}  // namespace mrcnn
