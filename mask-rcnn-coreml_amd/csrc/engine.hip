// engine.hip — artefact loading, BatchNorm folding / weight packing, and the static schedules of the
// three models.  Replaces the Core ML graph executors of MaskRCNN.mlmodel, Classifier.mlmodel and
// Mask.mlmodel (reference: Sources/maskrcnn/Python/Conversion/task.py:69-116 emits the specs; the
// layer list itself is the Matterport layout of the un-vendored `maskrcnn` package).
//
// HBM layout: all activations NHWC fp32 in one arena, one dense tensor per layer output (a batch of
// 8 × 1024² images needs ≈9 GB of the 288 GB); weights are packed once at load as [cout][tap][cin]
// fp32 with BatchNorm folded into a per-channel scale/shift; the pyramid P2..P5 stays resident for
// both ROIAlign passes (the reference re-uploads 89 MB of textures per call,
// PyramidROIAlignLayer.swift:110-118).
#include <dlfcn.h>

#include "engine.h"

#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <fstream>

namespace mrcnn {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
const char* last_error() { return g_last_error.c_str(); }

// ---- roctx (run-time bound) ----------------------------------------------------------------------------------------
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char* e = getenv("MRCNN_ROCTX");
        if (e && atoi(e) == 0) return;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            auto pu = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            auto po = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (pu && po) { push = pu; pop = po; return; }
        }
    }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void trace_push(const char* name) { if (roctx().push) (void)roctx().push(name); }
void trace_pop() { if (roctx().pop) (void)roctx().pop(); }

void fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error{code, buf};
}

void require_gpu()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail(MRCNN_ERR_HIP, "no HIP device visible (%s): libmaskrcnn_hip has no CPU fallback",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIP_CHECK(hipGetDeviceProperties(&p, dev));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
        fail(MRCNN_ERR_HIP, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, p.gcnArchName);
}

// ------------------------------------------------------------------------------------------------
// .mrcw
// ------------------------------------------------------------------------------------------------
static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else {
            exp = 113;
            while ((man & 0x400u) == 0) { man <<= 1; --exp; }
            man &= 0x3FFu;
            f = sign | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) f = sign | 0x7F800000u | (man << 13);
    else f = sign | ((exp + 112) << 23) | (man << 13);
    float r;
    memcpy(&r, &f, 4);
    return r;
}

void MrcwFile::load(const std::string& p)
{
    path = p;
    std::ifstream in(p, std::ios::binary | std::ios::ate);
    MRCNN_REQUIRE(in.good(), MRCNN_ERR_IO, "cannot open model artefact '%s'", p.c_str());
    const std::streamsize sz = in.tellg();
    in.seekg(0);
    buf.resize((size_t)sz);
    in.read(reinterpret_cast<char*>(buf.data()), sz);
    MRCNN_REQUIRE(in.good() && sz >= 16 && memcmp(buf.data(), "MRCW", 4) == 0, MRCNN_ERR_IO, "'%s' is not a .mrcw file", p.c_str());
    size_t pos = 4;
    auto need = [&](size_t n) { MRCNN_REQUIRE(pos + n <= buf.size(), MRCNN_ERR_IO, "'%s': truncated header", p.c_str()); };
    auto rd = [&](void* dst, size_t n) { need(n); memcpy(dst, buf.data() + pos, n); pos += n; };
    uint32_t ver, n_meta, n_t;
    rd(&ver, 4); rd(&n_meta, 4); rd(&n_t, 4);
    MRCNN_REQUIRE(ver == 1, MRCNN_ERR_IO, "'%s': unsupported .mrcw version %u", p.c_str(), ver);
    for (uint32_t i = 0; i < n_meta; ++i) {
        uint16_t kl; rd(&kl, 2);
        need(kl);
        std::string k((const char*)buf.data() + pos, kl); pos += kl;
        uint8_t t; rd(&t, 1);
        if (t == 0) { int64_t v; rd(&v, 8); ints[k] = v; }
        else if (t == 1) { double v; rd(&v, 8); doubles[k] = v; }
        else { uint32_t sl; rd(&sl, 4); need(sl); strings[k] = std::string((const char*)buf.data() + pos, sl); pos += sl; }
    }
    struct Ent { std::string name; MrcwTensor t; uint64_t off; };
    std::vector<Ent> ents;
    for (uint32_t i = 0; i < n_t; ++i) {
        uint16_t nl; rd(&nl, 2);
        need(nl);
        Ent e;
        e.name = std::string((const char*)buf.data() + pos, nl); pos += nl;
        uint8_t dt, nd; rd(&dt, 1); rd(&nd, 1);
        e.t.dtype = dt;
        e.t.dims.resize(nd);
        for (int d = 0; d < nd; ++d) rd(&e.t.dims[d], 4);
        uint64_t nb; rd(&e.off, 8); rd(&nb, 8);
        e.t.nbytes = (size_t)nb;
        ents.push_back(std::move(e));
    }
    const size_t base = (pos + 63) / 64 * 64;
    for (auto& e : ents) {
        MRCNN_REQUIRE(base + e.off + e.t.nbytes <= buf.size(), MRCNN_ERR_IO, "'%s': tensor %s out of bounds", p.c_str(), e.name.c_str());
        MRCNN_REQUIRE(e.t.dtype == 0 || e.t.dtype == 2, MRCNN_ERR_IO, "'%s': tensor %s has unsupported dtype", p.c_str(), e.name.c_str());
        e.t.data = buf.data() + base + e.off;
        tensors[e.name] = e.t;
    }
}
int64_t MrcwFile::get_int(const std::string& k) const
{
    auto it = ints.find(k);
    MRCNN_REQUIRE(it != ints.end(), MRCNN_ERR_IO, "'%s': missing integer metadata key '%s'", path.c_str(), k.c_str());
    return it->second;
}
int64_t MrcwFile::get_int(const std::string& k, int64_t dflt) const
{
    auto it = ints.find(k);
    return it == ints.end() ? dflt : it->second;
}
double MrcwFile::get_double(const std::string& k, double dflt) const
{
    auto it = doubles.find(k);
    if (it != doubles.end()) return it->second;
    auto ii = ints.find(k);
    return ii == ints.end() ? dflt : (double)ii->second;
}
std::string MrcwFile::get_string(const std::string& k) const
{
    auto it = strings.find(k);
    MRCNN_REQUIRE(it != strings.end(), MRCNN_ERR_IO, "'%s': missing string metadata key '%s'", path.c_str(), k.c_str());
    return it->second;
}
const MrcwTensor& MrcwFile::tensor(const std::string& name) const
{
    auto it = tensors.find(name);
    MRCNN_REQUIRE(it != tensors.end(), MRCNN_ERR_IO, "'%s': missing tensor '%s'", path.c_str(), name.c_str());
    return it->second;
}
std::vector<float> MrcwFile::floats(const std::string& name) const
{
    const MrcwTensor& t = tensor(name);
    std::vector<float> v(t.count());
    if (t.dtype == 0) memcpy(v.data(), t.data, v.size() * 4);
    else {
        const uint16_t* h = reinterpret_cast<const uint16_t*>(t.data);
        for (size_t i = 0; i < v.size(); ++i) v[i] = half_to_float(h[i]);
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
static void upload(DevBuf& d, const std::vector<float>& h)
{
    d.alloc(h.size() * 4);
    HIP_CHECK(hipMemcpy(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
}

static uint16_t float_to_half(float f)   // round to nearest even
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            // overflow → inf
    if (x < 0x38800000u) {                                              // subnormal half / zero
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;                                      // 14..24
        const uint32_t half_m = m >> shift, rem = m & ((1u << shift) - 1), mid = 1u << (shift - 1);
        uint32_t r = half_m;
        if (rem > mid || (rem == mid && (half_m & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}

// filter weights in the compute dtype (the artefact stores fp16, so the fp16 path is lossless)
static void upload_w(DevBuf& d, const std::vector<float>& h, int dtype, bool must_be_exact = false, const char* what = "")
{
    if (dtype != MRCNN_F16 && dtype != MRCNN_F32X3) { upload(d, h); return; }
    std::vector<uint16_t> hh(h.size());
    for (size_t i = 0; i < h.size(); ++i) {
        hh[i] = float_to_half(h[i]);
        if (must_be_exact) {
            const _Float16 back = __builtin_bit_cast(_Float16, hh[i]);
            MRCNN_REQUIRE((float)back == h[i], MRCNN_ERR_UNSUPPORTED,
                          "%s: filter value %.9g is not fp16-representable; MRCNN_F32S needs the fp16 filters the converter writes", what, h[i]);
        }
    }
    d.alloc(hh.size() * 2);
    HIP_CHECK(hipMemcpy(d.p, hh.data(), hh.size() * 2, hipMemcpyHostToDevice));
}

// scale/shift of conv(+bias)(+BN):  y = acc*scale + shift
static void fold_bn(const MrcwFile& f, const std::string& conv, const std::string& bn, int Cout, int Npad,
                    std::vector<float>& scale, std::vector<float>& shift)
{
    const std::vector<float> bias = f.floats(conv + "/bias");
    MRCNN_REQUIRE((int)bias.size() == Cout, MRCNN_ERR_IO, "%s/bias has %zu entries, expected %d", conv.c_str(), bias.size(), Cout);
    scale.assign(Npad, 0.f);
    shift.assign(Npad, 0.f);
    if (bn.empty()) {
        for (int o = 0; o < Cout; ++o) { scale[o] = 1.f; shift[o] = bias[o]; }
        return;
    }
    const std::vector<float> g = f.floats(bn + "/gamma"), be = f.floats(bn + "/beta"), mu = f.floats(bn + "/mean"),
                             var = f.floats(bn + "/variance");
    const float eps = (float)f.get_double("bn_eps", 1e-3);
    for (int o = 0; o < Cout; ++o) {
        const float sc = g[o] / sqrtf(var[o] + eps);
        scale[o] = sc;
        shift[o] = (bias[o] - mu[o]) * sc + be[o];
    }
}

// kernel [O][I][KH][KW] (Core ML layout) → [Npad][KH][KW][I]
PackedConv pack_conv_oihw(const MrcwFile& f, const std::string& conv, const std::string& bn, int dtype)
{
    const MrcwTensor& t = f.tensor(conv + "/kernel");
    MRCNN_REQUIRE(t.dims.size() == 4, MRCNN_ERR_IO, "%s/kernel is not 4-D", conv.c_str());
    const int O = t.dims[0], I = t.dims[1], KH = t.dims[2], KW = t.dims[3];
    const std::vector<float> k = f.floats(conv + "/kernel");
    PackedConv pc;
    pc.dtype = mode_act(dtype); pc.wdtype = mode_wgt(dtype);
    pc.Cin = I; pc.Cout = O; pc.KH = KH; pc.KW = KW;
    const int bn_tile = conv_n_tile(O);
    pc.Npad = (O + bn_tile - 1) / bn_tile * bn_tile;
    std::vector<float> w((size_t)pc.Npad * KH * KW * I, 0.f);
    for (int o = 0; o < O; ++o)
        for (int i = 0; i < I; ++i)
            for (int y = 0; y < KH; ++y)
                for (int x = 0; x < KW; ++x)
                    w[(((size_t)o * KH + y) * KW + x) * I + i] = k[(((size_t)o * I + i) * KH + y) * KW + x];
    upload_w(pc.wgt, w, pc.wdtype, pc.wdtype != pc.dtype, "MRCNN_F32S / MRCNN_F32X3");
    if (pc.wdtype != pc.dtype && conv_halo_packable(KH, KW, I, pc.Npad)) {
        conv_halo_pack(nullptr, pc.wgt.p, pc.Npad, I, pc.wgt_halo);           // split modes: also in the halo kernel's tiling
        HIP_CHECK(hipStreamSynchronize(nullptr));
    } else if (pc.wdtype != pc.dtype && conv_halo_tail_packable(KH, KW, I, pc.Npad) && O == pc.Npad) {
        conv_halo_pack(nullptr, pc.wgt.p, pc.Npad, I, pc.wgt_halo, 1);        // ... the 256 -> 1024 1x1 layers for the fused bottleneck tail
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    if (pc.dtype == MRCNN_F16 && pc.wdtype == MRCNN_F16 && conv3x3h_packable(KH, KW, I, O, pc.Npad)) {
        conv3x3h_pack(nullptr, pc.wgt.p, O, I, pc.wgt_c3h);
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    if (pc.dtype == MRCNN_F16 && pc.wdtype == MRCNN_F16 && O == pc.Npad && bneck_frag_wanted(KH, KW, I, O)) {
        bneck_pack_frag(nullptr, pc.wgt.p, O, KH * KW * I, pc.wgt_frag);      // fp16 mode: C4's identity bottlenecks stream these straight into registers
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    std::vector<float> sc, sh;
    fold_bn(f, conv, bn, O, pc.Npad, sc, sh);
    upload(pc.scale, sc);
    upload(pc.shift, sh);
    pc.h_scale = sc; pc.h_shift = sh;
    return pc;
}

// conv1: 7×7 stride 2 on 3 channels.  The input is staged as zero-padded NHWC4, so one kernel row
// (7 taps × 4 channels = 28 floats, padded to 32) is a contiguous 128-B run: taps = 7 (rows),
// "Cin" = 32.  Packed [64][7][32] with k = kw*4 + ci.
static PackedConv pack_conv1(const MrcwFile& f, int dtype)
{
    const MrcwTensor& t = f.tensor("conv1/kernel");
    MRCNN_REQUIRE(t.dims.size() == 4 && t.dims[1] == 3 && t.dims[2] == 7 && t.dims[3] == 7, MRCNN_ERR_IO, "conv1/kernel must be [O,3,7,7]");
    const int O = t.dims[0];
    const std::vector<float> k = f.floats("conv1/kernel");
    PackedConv pc;
    pc.dtype = mode_act(dtype); pc.wdtype = mode_wgt(dtype);
    const int px = pc.dtype == MRCNN_F16 ? 8 : 4;   // channels per staged pixel (NHWC8 / NHWC4): 16 B either way
    const int row = 8 * px;                         // one kernel row = 7 taps padded to 8 pixels = 128 B
    pc.Cin = row; pc.Cout = O; pc.KH = 7; pc.KW = 1;
    const int bn_tile = conv_n_tile(O);
    pc.Npad = (O + bn_tile - 1) / bn_tile * bn_tile;
    std::vector<float> w((size_t)pc.Npad * 7 * row, 0.f);
    for (int o = 0; o < O; ++o)
        for (int ci = 0; ci < 3; ++ci)
            for (int y = 0; y < 7; ++y)
                for (int x = 0; x < 7; ++x)
                    w[((size_t)o * 7 + y) * row + x * px + ci] = k[(((size_t)o * 3 + ci) * 7 + y) * 7 + x];
    upload_w(pc.wgt, w, pc.wdtype, pc.wdtype != pc.dtype, "MRCNN_F32S / MRCNN_F32X3");
    std::vector<float> sc, sh;
    fold_bn(f, "conv1", "bn_conv1", O, pc.Npad, sc, sh);
    upload(pc.scale, sc);
    upload(pc.shift, sh);
    pc.h_scale = sc; pc.h_shift = sh;
    return pc;
}

// Several [O_i][I] inner-product / 1×1 kernels stacked along N (no BN).
static PackedConv pack_stacked_1x1(const MrcwFile& f, const std::vector<std::string>& names, int dtype)
{
    PackedConv pc;
    pc.dtype = mode_act(dtype); pc.wdtype = mode_wgt(dtype);
    int O = 0, I = -1;
    for (auto& n : names) {
        const MrcwTensor& t = f.tensor(n + "/kernel");
        MRCNN_REQUIRE(t.dims.size() == 2 || (t.dims.size() == 4 && t.dims[2] == 1 && t.dims[3] == 1), MRCNN_ERR_IO,
                      "%s/kernel is not an inner product / 1x1 kernel", n.c_str());
        MRCNN_REQUIRE(I < 0 || I == (int)t.dims[1], MRCNN_ERR_IO, "%s: input size mismatch", n.c_str());
        I = t.dims[1];
        O += t.dims[0];
    }
    pc.Cin = I; pc.Cout = O; pc.KH = pc.KW = 1;
    const int bn_tile = conv_n_tile(O);
    pc.Npad = (O + bn_tile - 1) / bn_tile * bn_tile;
    std::vector<float> w((size_t)pc.Npad * I, 0.f), sc(pc.Npad, 0.f), sh(pc.Npad, 0.f);
    int o0 = 0;
    for (auto& n : names) {
        const std::vector<float> k = f.floats(n + "/kernel"), b = f.floats(n + "/bias");
        memcpy(w.data() + (size_t)o0 * I, k.data(), k.size() * 4);
        for (size_t o = 0; o < b.size(); ++o) { sc[o0 + o] = 1.f; sh[o0 + o] = b[o]; }
        o0 += (int)b.size();
    }
    upload_w(pc.wgt, w, pc.wdtype, pc.wdtype != pc.dtype, "MRCNN_F32S / MRCNN_F32X3");
    upload(pc.scale, sc);
    upload(pc.shift, sh);
    pc.h_scale = sc; pc.h_shift = sh;
    return pc;
}

// ConvTranspose 2×2 stride 2, kernel [I][O][2][2] → GEMM rows n = (dy*2+dx)*O + co.
static PackedConv pack_deconv2(const MrcwFile& f, const std::string& name, int dtype)
{
    const MrcwTensor& t = f.tensor(name + "/kernel");
    MRCNN_REQUIRE(t.dims.size() == 4 && t.dims[2] == 2 && t.dims[3] == 2, MRCNN_ERR_IO, "%s/kernel must be [I,O,2,2]", name.c_str());
    const int I = t.dims[0], O = t.dims[1];
    const std::vector<float> k = f.floats(name + "/kernel"), b = f.floats(name + "/bias");
    PackedConv pc;
    pc.dtype = mode_act(dtype); pc.wdtype = mode_wgt(dtype);
    pc.Cin = I; pc.Cout = O; pc.KH = pc.KW = 1;
    pc.Npad = 4 * O;
    MRCNN_REQUIRE(pc.Npad % conv_n_tile(pc.Npad) == 0, MRCNN_ERR_SHAPE, "%s: 4*O must be a multiple of the N tile", name.c_str());
    std::vector<float> w((size_t)pc.Npad * I), sc(pc.Npad, 1.f), sh(pc.Npad);
    for (int qd = 0; qd < 4; ++qd)
        for (int o = 0; o < O; ++o) {
            for (int i = 0; i < I; ++i) w[((size_t)qd * O + o) * I + i] = k[(((size_t)i * O + o) * 2 + (qd >> 1)) * 2 + (qd & 1)];
            sh[(size_t)qd * O + o] = b[o];
        }
    upload_w(pc.wgt, w, pc.wdtype, pc.wdtype != pc.dtype, "MRCNN_F32S / MRCNN_F32X3");
    upload(pc.scale, sc);
    upload(pc.shift, sh);
    pc.h_scale = sc; pc.h_shift = sh;
    return pc;
}

// A convolution's own copy of its scale / shift (= the folded BatchNorm + bias until exponents are applied)
void init_scaled_op(ScaledOp& op, const PackedConv* pc, int g_in, int g_out)
{
    op.pc = pc; op.g_in = g_in; op.g_out = g_out;
    upload(op.scale, pc->h_scale);
    upload(op.shift, pc->h_shift);
}

// Dense NHWC conv helper (in: B×H×W×Cin, out: B×OH×OW×Cout).
void run_conv_dense(hipStream_t s, const PackedConv& pc, const void* in, int B, int H, int W, void* out, int stride,
                    int pad, int act, const void* res, int out_f32, const ScaledOp* sop)
{
    ConvDesc d;
    d.dtype = pc.dtype; d.wdtype = pc.wdtype; d.out_f32 = out_f32;
    d.in = in; d.B = B; d.H = H; d.W = W; d.Cin = pc.Cin;
    d.in_sW = pc.Cin; d.in_sH = (long)W * pc.Cin; d.in_sB = (long)H * W * pc.Cin;
    d.wgt = pc.wgt.p; d.KH = pc.KH; d.KW = pc.KW; d.stride = stride; d.padH = d.padW = pad;
    d.wgt_halo = pc.wgt_halo.p;
    d.scale = sop ? sop->scale.as<float>() : pc.scale.as<float>(); d.shift = sop ? sop->shift.as<float>() : pc.shift.as<float>();
    d.OH = (H + 2 * pad - pc.KH) / stride + 1;
    d.OW = (W + 2 * pad - pc.KW) / stride + 1;
    d.Cout = pc.Cout; d.Npad = pc.Npad;
    d.out = out; d.out_sP = pc.Cout; d.out_sB = (long)d.OH * d.OW * pc.Cout;
    d.act = act;
    if (res) { d.res = res; d.res_sB = d.out_sB; d.res_sW = pc.Cout; d.res_sH = (long)d.OW * pc.Cout; }
    conv_forward(s, d);
}

// ------------------------------------------------------------------------------------------------
// Classifier head
// ------------------------------------------------------------------------------------------------
void ClassifierHead::load(const MrcwFile& f, int capacity_rows, int dtype_)
{
    nc = (int)f.get_int("num_classes");
    cap = capacity_rows;
    mode = dtype_;
    dtype = mode_act(mode);
    const MrcwTensor& k1 = f.tensor("mrcnn_class_conv1/kernel");
    MRCNN_REQUIRE(k1.dims.size() == 4, MRCNN_ERR_IO, "mrcnn_class_conv1/kernel must be 4-D");
    C = k1.dims[1]; pool = k1.dims[2];
    fc1 = pack_conv_oihw(f, "mrcnn_class_conv1", "mrcnn_class_bn1", mode);    // [1024][7][7][256] == rows of the NHWC pooled vector
    fc1.Cin = fc1.Cin * fc1.KH * fc1.KW; fc1.KH = fc1.KW = 1;           // as an inner product over K = 12544
    fc2 = pack_conv_oihw(f, "mrcnn_class_conv2", "mrcnn_class_bn2", mode);
    fc3 = pack_stacked_1x1(f, {"mrcnn_class_logits", "mrcnn_bbox_fc"}, mode);
    MRCNN_REQUIRE(fc3.Cout == 5 * nc, MRCNN_ERR_IO, "classifier output size %d != 5*num_classes", fc3.Cout);
    Arena ar;
    for (int pass = 0; pass < 2; ++pass) {
        ar.off = 0;
        h1 = ar.alloc_e((size_t)cap * fc1.Cout, dtype);
        h2 = ar.alloc_e((size_t)cap * fc2.Cout, dtype);
        lb = ar.alloc_f((size_t)cap * fc3.Cout);
        probs = ar.alloc_f((size_t)cap * nc);
        bbox = ar.alloc_f((size_t)cap * nc * 4);
        cls6 = ar.alloc_f((size_t)cap * 6);
        stage_in = ar.alloc_e((size_t)cap * pool * pool * C, dtype);
        if (pass == 0) { arena.alloc(ar.off); ar.base = arena.as<char>(); }
    }
    init_scaled_op(sop[0], &fc1, -1, -1);
    init_scaled_op(sop[1], &fc2, -1, -1);
    init_scaled_op(sop[2], &fc3, -1, -1);
}

void ClassifierHead::forward(hipStream_t s, const void* pooled_nhwc, int n, float* cls6_out, long cls6_stride)
{
    MRCNN_REQUIRE(n <= cap, MRCNN_ERR_SHAPE, "classifier head: %d rows exceed capacity %d", n, cap);
    if (n <= 0) return;
    // rows are "pixels" of a 1×n image
    run_conv_dense(s, fc1, pooled_nhwc, 1, 1, n, h1, 1, 0, ACT_RELU, nullptr, 0, &sop[0]);
    if (observe) observe(s, grp[1], h1, (size_t)n * fc1.Cout);
    run_conv_dense(s, fc2, h1, 1, 1, n, h2, 1, 0, ACT_RELU, nullptr, 0, &sop[1]);
    if (observe) observe(s, grp[2], h2, (size_t)n * fc2.Cout);
    run_conv_dense(s, fc3, h2, 1, 1, n, lb, 1, 0, ACT_NONE, nullptr, 1, &sop[2]);
    TraceRange tr("TimeDistributedClassifierLayer-ProcessOutput");
    softmax_rows_forward(s, lb, fc3.Cout, nc, n, probs);
    copy_columns_forward(s, lb, fc3.Cout, nc, 4 * nc, n, bbox);
    if (cls6_out) classifier_postprocess_forward(s, probs, bbox, nc, n, cls6_out, cls6_stride);
}

// ------------------------------------------------------------------------------------------------
// Mask head
// ------------------------------------------------------------------------------------------------
void MaskHead::load(const MrcwFile& f, int capacity_rows, int dtype_)
{
    nc = (int)f.get_int("num_classes");
    cap = capacity_rows;
    mode = dtype_;
    dtype = mode_act(mode);
    for (int i = 0; i < 4; ++i)
        conv[i] = pack_conv_oihw(f, "mrcnn_mask_conv" + std::to_string(i + 1), "mrcnn_mask_bn" + std::to_string(i + 1), mode);
    C = conv[0].Cin;
    deconv = pack_deconv2(f, "mrcnn_mask_deconv", mode);
    final_full = pack_conv_oihw(f, "mrcnn_mask", "", mode);
    upload(final_w, f.floats("mrcnn_mask/kernel"));
    upload(final_b, f.floats("mrcnn_mask/bias"));
    const size_t hw = (size_t)pool * pool;
    Arena ar;
    for (int pass = 0; pass < 2; ++pass) {
        ar.off = 0;
        t0 = ar.alloc_e((size_t)cap * hw * C, dtype);
        t1 = ar.alloc_e((size_t)cap * hw * C, dtype);
        feat = ar.alloc_e((size_t)cap * hw * 4 * deconv.Cout, dtype);
        full = ar.alloc_f((size_t)cap * hw * 4 * nc);
        stage_in = ar.alloc_e((size_t)cap * hw * C, dtype);
        if (pass == 0) { arena.alloc(ar.off); ar.base = arena.as<char>(); }
    }
    for (int i = 0; i < 4; ++i) init_scaled_op(sop[i], &conv[i], -1, -1);
    init_scaled_op(sop[4], &deconv, -1, -1);
}

// process-wide A/B switch of the fused mask tail (tests, tools/e2e_ab.py): mrcnn_debug_set("mask_fused", 0 | 1)
static int g_fuse_mask_tail = 1;
bool engine_debug_set(const char* key, int value)
{
    if (std::string(key) == "mask_fused") { g_fuse_mask_tail = value; return true; }
    return false;
}

void MaskHead::forward_features(hipStream_t s, const void* pooled_nhwc, int n, const int32_t* sel_cid, float* sel_partial)
{
    MRCNN_REQUIRE(n <= cap, MRCNN_ERR_SHAPE, "mask head: %d rows exceed capacity %d", n, cap);
    if (n <= 0) return;
    const size_t te = (size_t)n * pool * pool * C;
    run_conv_dense(s, conv[0], pooled_nhwc, n, pool, pool, t0, 1, 1, ACT_RELU, nullptr, 0, &sop[0]);
    if (observe) observe(s, grp[1], t0, te);
    run_conv_dense(s, conv[1], t0, n, pool, pool, t1, 1, 1, ACT_RELU, nullptr, 0, &sop[1]);
    if (observe) observe(s, grp[2], t1, te);
    run_conv_dense(s, conv[2], t1, n, pool, pool, t0, 1, 1, ACT_RELU, nullptr, 0, &sop[2]);
    if (observe) observe(s, grp[3], t0, te);
    run_conv_dense(s, conv[3], t0, n, pool, pool, t1, 1, 1, ACT_RELU, nullptr, 0, &sop[3]);
    if (observe) observe(s, grp[4], t1, te);
    ConvDesc d;
    const int Co = deconv.Cout;
    d.dtype = dtype;
    d.in = t1; d.B = n; d.H = pool; d.W = pool; d.Cin = deconv.Cin;
    d.in_sW = deconv.Cin; d.in_sH = (long)pool * deconv.Cin; d.in_sB = (long)pool * pool * deconv.Cin;
    d.wdtype = deconv.wdtype;
    d.wgt = deconv.wgt.p; d.scale = sop[4].scale.as<float>(); d.shift = sop[4].shift.as<float>();
    d.OH = pool; d.OW = pool; d.Cout = Co; d.Npad = deconv.Npad;
    d.deconv2 = 1; d.act = ACT_RELU;
    d.out = feat; d.out_sW = Co; d.out_sH = (long)2 * pool * Co; d.out_sB = (long)4 * pool * pool * Co;
    d.out_sP = Co;
    if (sel_partial) { d.sel_w = final_w.as<float>(); d.sel_cid = sel_cid; d.sel_partial = sel_partial; }   // feat is not written
    conv_forward(s, d);
}

void MaskHead::forward_full(hipStream_t s, int n)
{
    if (n <= 0) return;
    ConvDesc d;
    const int P2 = 2 * pool;
    d.dtype = dtype; d.out_f32 = 1;
    d.in = feat; d.B = n; d.H = P2; d.W = P2; d.Cin = final_full.Cin;
    d.in_sW = d.Cin; d.in_sH = (long)P2 * d.Cin; d.in_sB = (long)P2 * P2 * d.Cin;
    d.wdtype = final_full.wdtype;
    d.wgt = final_full.wgt.p; d.scale = final_full.scale.as<float>(); d.shift = final_full.shift.as<float>();
    d.OH = P2; d.OW = P2; d.Cout = nc; d.Npad = final_full.Npad;
    d.act = ACT_SIGMOID;
    d.out = full; d.out_sP = nc; d.out_sB = (long)P2 * P2 * nc;
    conv_forward(s, d);
}

// ------------------------------------------------------------------------------------------------
// timers
// ------------------------------------------------------------------------------------------------
void StageTimer::begin(hipStream_t s)
{
    if (!enabled) return;
    names.clear();
    if (ev.empty()) {
        ev.resize(16);
        for (auto& e : ev) HIP_CHECK(hipEventCreate(&e));
    }
    HIP_CHECK(hipEventRecord(ev[0], s));
}
void StageTimer::mark(hipStream_t s, const char* name)
{
    if (!enabled) return;
    names.push_back(name);
    HIP_CHECK(hipEventRecord(ev[names.size()], s));
}
void StageTimer::finish()
{
    if (!enabled || names.empty()) return;
    HIP_CHECK(hipEventSynchronize(ev[names.size()]));
    ms.clear();
    for (size_t i = 0; i < names.size(); ++i) {
        float t = 0;
        HIP_CHECK(hipEventElapsedTime(&t, ev[i], ev[i + 1]));
        ms[names[i]] = t;
    }
}
StageTimer::~StageTimer()
{
    for (auto& e : ev) (void)hipEventDestroy(e);
}

// ------------------------------------------------------------------------------------------------
// Model
// ------------------------------------------------------------------------------------------------
void Model::drop_graphs()
{
    for (auto& kv : graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    graphs.clear();
}

Model::~Model()
{
    drop_graphs();
    for (auto& sl : pipe) {
        if (sl.ev_in) (void)hipEventDestroy(sl.ev_in);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
    }
    if (pipe_in) (void)hipStreamDestroy(pipe_in);
    if (pipe_out) (void)hipStreamDestroy(pipe_out);
    if (ev_p0) (void)hipEventDestroy(ev_p0);
    if (ev_p1) (void)hipEventDestroy(ev_p1);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

static void read_std(const MrcwFile& f, const std::string& prefix, float out[4])
{
    const float dflt[4] = {0.1f, 0.1f, 0.2f, 0.2f};
    const int64_t cnt = f.get_int(prefix + "bboxStdDev_count", 0);
    for (int i = 0; i < 4; ++i) out[i] = dflt[i];
    if (cnt == 4)
        for (int i = 0; i < 4; ++i) out[i] = (float)f.get_double(prefix + "bboxStdDev_" + std::to_string(i), dflt[i]);
}

void Model::load(int kind_, const std::string& path, int max_batch_, int dtype_)
{
    require_gpu();
    kind = kind_;
    max_batch = max_batch_ > 0 ? max_batch_ : 1;
    file.load(path);
    mode = dtype_;
    if (dtype_ == MRCNN_DEFAULT) {
        // the mode the artefact is prepared for (include/maskrcnn_hip.h): stored split exponents -> the three-part split, else exact fp32
        bool stored = false;
        if (kind == MRCNN_MODEL_MASKRCNN)
            for (auto& kv : file.ints) stored = stored || kv.first.compare(0, 10, "split_exp.") == 0;
        mode = stored ? MRCNN_F32X3 : MRCNN_F32;
        mode_defaulted = true;
    }
    dtype = mode_act(mode);
    const char* want = kind == MRCNN_MODEL_MASKRCNN ? "MaskRCNN" : kind == MRCNN_MODEL_CLASSIFIER ? "Classifier" : "Mask";
    MRCNN_REQUIRE(file.get_string("kind") == want, MRCNN_ERR_IO, "'%s' holds a %s model, expected %s", path.c_str(),
                  file.get_string("kind").c_str(), want);
    if (const char* cm = knob_env("MRCNN_CU_MASK_PROBE")) {
        fprintf(stderr, "libmaskrcnn_hip: MRCNN_CU_MASK_PROBE — this handle's stream runs on a subset of the CUs (measurement only)\n");
        // measurement only (tools/dual_stream_probe.py): the handle's stream on a subset of the CUs — 8 hexadecimal words, "w0,w1,...,w7"
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int n = 0;
        for (const char* p = cm; *p && n < 8; ++n) { words[n] = (uint32_t)strtoul(p, nullptr, 16); p = strchr(p, ','); if (!p) { ++n; break; } ++p; }
        HIP_CHECK(hipExtStreamCreateWithCUMask(&stream, 8, words));
    } else
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    own_stream = true;
    if (const char* e = getenv("MRCNN_GRAPH")) use_graph = atoi(e) != 0;
    if (const char* e = knob_env("MRCNN_FUSE_MASK_TAIL")) fuse_mask_tail = atoi(e) != 0;
    boxes_one_time_init();
    range_flag.alloc(sizeof(int));
    HIP_CHECK(hipMemset(range_flag.p, 0, sizeof(int)));
    nc = (int)file.get_int("num_classes");
    // split modes: the shared-tile K chunks of the long-K 1x1 layers on under-filled grids (single images) and the opt-in fused
    // bottleneck tail need device scratch — owned by this handle from load to destroy (kernels.h: ConvScratch)
    if ((mode == MRCNN_F32S || mode == MRCNN_F32X3) && kind != MRCNN_MODEL_MASK) conv_scratch.alloc();
    if (kind == MRCNN_MODEL_CLASSIFIER) { cls_head.load(file, max_batch, mode); return; }
    if (kind == MRCNN_MODEL_MASK) { mask_head.load(file, max_batch, mode); return; }
    build_maskrcnn();
}

void Model::build_maskrcnn()
{
    const MrcwFile& f = file;
    arch = f.get_string("architecture");
    MRCNN_REQUIRE(arch == "resnet101" || arch == "resnet50", MRCNN_ERR_UNSUPPORTED, "architecture '%s'", arch.c_str());
    H = (int)f.get_int("image_height"); W = (int)f.get_int("image_width");
    MRCNN_REQUIRE(H % 64 == 0 && W % 64 == 0 && H > 0 && W > 0, MRCNN_ERR_SHAPE, "input_image_shape %dx%d must be multiples of 64", H, W);
    na = (int)f.get_int("num_anchors_per_location", 3);
    mean[0] = (float)f.get_double("mean_r", 123.7); mean[1] = (float)f.get_double("mean_g", 116.8); mean[2] = (float)f.get_double("mean_b", 103.9);
    pre_nms = (int)f.get_int("ProposalLayer.preNMSMaxProposals", 6000);
    max_prop = (int)f.get_int("ProposalLayer.maxProposals", 1000);
    prop_nms_thr = (float)f.get_double("ProposalLayer.nmsIOUThreshold", 0.7);
    read_std(f, "ProposalLayer.", prop_std);
    max_det = (int)f.get_int("DetectionLayer.maxDetections", 100);
    det_score_thr = (float)f.get_double("DetectionLayer.scoreThreshold", 0.7);
    det_nms_thr = (float)f.get_double("DetectionLayer.nmsIOUThreshold", 0.3);
    read_std(f, "DetectionLayer.", det_std);
    cls_pool = (int)f.get_int("PyramidROIAlignLayer.classifier.poolSize", 7);
    mask_pool = (int)f.get_int("PyramidROIAlignLayer.mask.poolSize", 14);
    roi_img_w = f.get_double("PyramidROIAlignLayer.classifier.imageWidth", (double)W);
    roi_img_h = f.get_double("PyramidROIAlignLayer.classifier.imageHeight", (double)H);

    // ---- sub-artefacts from the config singleton (ProposalLayer.swift:68, TimeDistributed*Layer.swift:41/49)
    const char* ap = mrcnn_config_get_anchors_path();
    const char* cp = mrcnn_config_get_classifier_path();
    const char* mp = mrcnn_config_get_mask_path();
    MRCNN_REQUIRE(ap, MRCNN_ERR_CONFIG, "MaskRCNNConfig.anchorsURL must be set before the MaskRCNN model is loaded");
    MRCNN_REQUIRE(cp, MRCNN_ERR_CONFIG, "MaskRCNNConfig.compiledClassifierModelURL must be set before the MaskRCNN model is loaded");
    MRCNN_REQUIRE(mp, MRCNN_ERR_CONFIG, "MaskRCNNConfig.compiledMaskModelURL must be set before the MaskRCNN model is loaded");

    const int strides[5] = {4, 8, 16, 32, 64};
    int fh[5], fw[5];
    A = 0;
    long lvl_off[5];
    for (int l = 0; l < 5; ++l) {
        fh[l] = (H + strides[l] - 1) / strides[l];
        fw[l] = (W + strides[l] - 1) / strides[l];
        lvl_off[l] = A;
        A += fh[l] * fw[l] * na;
    }
    K = A < pre_nms ? A : pre_nms;
    {   // anchors.bin: A×4 little-endian float32 (task.py:173-176)
        std::ifstream in(ap, std::ios::binary | std::ios::ate);
        MRCNN_REQUIRE(in.good(), MRCNN_ERR_IO, "cannot open anchors file '%s'", ap);
        const std::streamsize sz = in.tellg();
        MRCNN_REQUIRE(sz == (std::streamsize)A * 16, MRCNN_ERR_IO, "anchors file '%s' has %lld bytes, expected %lld (%d anchors x 4 float32)", ap,
                      (long long)sz, (long long)A * 16, A);
        std::vector<float> h((size_t)A * 4);
        in.seekg(0);
        in.read(reinterpret_cast<char*>(h.data()), sz);
        upload(anchors, h);
    }
    {
        MrcwFile cf; cf.load(cp);
        MRCNN_REQUIRE(cf.get_string("kind") == "Classifier", MRCNN_ERR_IO, "'%s' is not a Classifier artefact", cp);
        cls_head.load(cf, max_batch * max_prop, mode);
        MRCNN_REQUIRE(cls_head.nc == nc, MRCNN_ERR_IO, "Classifier num_classes %d != %d", cls_head.nc, nc);
        MrcwFile mf; mf.load(mp);
        MRCNN_REQUIRE(mf.get_string("kind") == "Mask", MRCNN_ERR_IO, "'%s' is not a Mask artefact", mp);
        mask_head.load(mf, max_batch * max_det, mode);
        MRCNN_REQUIRE(mask_head.nc == nc, MRCNN_ERR_IO, "Mask num_classes %d != %d", mask_head.nc, nc);
    }

    // ---- trunk weights ----------------------------------------------------------------------------
    convs["conv1"] = pack_conv1(f, mode);
    std::vector<std::vector<std::string>> blocks(6);
    {
        const int n4 = arch == "resnet101" ? 22 : 5;
        blocks[2] = {"a", "b", "c"};
        blocks[3] = {"a", "b", "c", "d"};
        blocks[4] = {"a"};
        for (int i = 0; i < n4; ++i) blocks[4].push_back(std::string(1, (char)('b' + i)));
        blocks[5] = {"a", "b", "c"};
    }
    for (int st = 2; st <= 5; ++st)
        for (auto& b : blocks[st]) {
            const std::string p = std::to_string(st) + b;
            for (const char* br : {"2a", "2b", "2c"}) convs["res" + p + "_branch" + br] = pack_conv_oihw(f, "res" + p + "_branch" + br, "bn" + p + "_branch" + br, mode);
            if (b == "a") convs["res" + p + "_branch1"] = pack_conv_oihw(f, "res" + p + "_branch1", "bn" + p + "_branch1", mode);
        }
    for (const char* n : {"fpn_c5p5", "fpn_c4p4", "fpn_c3p3", "fpn_c2p2", "fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5", "rpn_conv_shared"})
        convs[n] = pack_conv_oihw(f, n, "", mode);
    convs["rpn_heads"] = pack_stacked_1x1(f, {"rpn_class_raw", "rpn_bbox_pred"}, mode);
    MRCNN_REQUIRE(convs["rpn_heads"].Cout == 6 * na, MRCNN_ERR_IO, "RPN head width %d != 6*anchors_per_location", convs["rpn_heads"].Cout);
    if (mode == MRCNN_F16 && convs["rpn_heads"].Npad == 32 && convs["rpn_heads"].Cin % 16 == 0) {       // fp16 mode: heads fused into the shared 3x3 layer (kernels_conv3x3_h.hip)
        bneck_pack_frag(nullptr, convs["rpn_heads"].wgt.p, 32, convs["rpn_heads"].Cin, rpn_head_frag);
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }
    if (convs["rpn_heads"].wdtype != convs["rpn_heads"].dtype && convs["rpn_heads"].Npad == 32) {       // split modes: heads fused into the shared 3x3 layer
        conv_halo_pack_head(nullptr, convs["rpn_heads"].wgt.p, 32, convs["rpn_heads"].Cin, rpn_head_frag);
        HIP_CHECK(hipStreamSynchronize(nullptr));
    }

    // ---- activation plan (pass 0 sizes the arena, pass 1 binds pointers and records the ops) ------
    const int Bm = max_batch;
    Arena ar;
    Model* const self = this;
    for (int pass = 0; pass < 2; ++pass) {
        ar.off = 0;
        trunk_ops.clear();
        taps.clear();
        sgroups.clear();
        sops.clear();
        stage_tabs.clear();
        const bool real = pass == 1;
        const int dt = dtype;
        auto T = [&](int h, int w, int c) { Tensor4 t; t.H = h; t.W = w; t.C = c; t.p = ar.alloc_e((size_t)Bm * h * w * c, dt); return t; };
        auto add = [&](Op op) { if (real) trunk_ops.push_back(std::move(op)); };
        // split groups (engine.h: SplitGroup): g_in / g_out of every convolution; a residual rides in the output's group
        const int g_img = new_split_group("image", true);        // pixel - mean: written by the pre-processing kernel, exponent 0
        const int g_zero = new_split_group("outputs", true);      // logits, box deltas, probabilities: consumed by fp32 arithmetic
        auto make_desc = [&](const std::string& name, const Tensor4& in, const Tensor4& out, int stride, int pad, int act,
                             const Tensor4* res, int res_shift, int g_in, int g_out) {
            const PackedConv* pc = &convs.at(name);
            ScaledOp* const so = real ? new_scaled_op(pc, g_in, g_out) : nullptr;
            ConvDesc d;
            d.dtype = dt;
            d.in = in.p; d.H = in.H; d.W = in.W; d.Cin = pc->Cin;
            d.in_sW = in.C; d.in_sH = (long)in.W * in.C; d.in_sB = in.sB();
            d.wdtype = pc->wdtype;
            d.wgt = pc->wgt.p; d.KH = pc->KH; d.KW = pc->KW; d.stride = stride; d.padH = d.padW = pad;
            d.wgt_halo = pc->wgt_halo.p;
            d.wgt_frag = pc->wgt_frag.p;
            d.wgt_c3h = pc->wgt_c3h.p;
            d.scale = so ? so->scale.as<float>() : nullptr; d.shift = so ? so->shift.as<float>() : nullptr;
            d.OH = out.H; d.OW = out.W; d.Cout = pc->Cout; d.Npad = pc->Npad;
            d.out = out.p; d.out_sP = out.C; d.out_sB = out.sB();
            d.act = act;
            d.group = name.compare(0, 3, "res") == 0 ? 1 : 0;          // conv profile: the backbone subset (C2..C5; conv1 is tagged below)
            if (res) { d.res = res->p; d.res_sB = res->sB(); d.res_sH = (long)res->W * res->C; d.res_sW = res->C; d.res_shift = res_shift; }
            return d;
        };
        auto conv_op = [&](const std::string& name, const Tensor4& in, const Tensor4& out, int stride, int pad, int act,
                           const Tensor4* res, int res_shift, int g_in, int g_out) {
            const ConvDesc d = make_desc(name, in, out, stride, pad, act, res, res_shift, g_in, g_out);
            const size_t per_image = (size_t)out.sB();
            add([d, self, g_out, per_image](hipStream_t s, int batch) {
                ConvDesc x = d; x.B = batch; conv_forward(s, x);
                if (self->calib_phase) self->observe_split(s, g_out, d.out, per_image * batch);
            });
        };
        // a bottleneck's tail: branch2b (3x3) and the branch2c (1x1 + shortcut) behind it — one fused launch where the pair
        // qualifies and the grid fills the chip (conv_forward_tail: bit-identical to the two launches), two launches otherwise
        // and while a calibration pass needs the tensor between them
        // dsc: the stage's shortcut convolution (first block) — computed inside branch2c's launch where the pair qualifies (conv_forward),
        // so that its tensor is neither written nor read back; its own launch in a calibration pass (which observes its output)
        auto tail_op = [&](const std::string& n3, const std::string& n1, const Tensor4& in3, const Tensor4& mid, const Tensor4& out, const Tensor4& res,
                           int g_in, int g_mid, int g_out, const ConvDesc* dsc) {
            const ConvDesc d3 = make_desc(n3, in3, mid, 1, 1, ACT_RELU, nullptr, 0, g_in, g_mid);
            const ConvDesc d1 = make_desc(n1, mid, out, 1, 0, ACT_RELU, &res, 0, g_mid, g_out);
            const size_t pm = (size_t)mid.sB(), po = (size_t)out.sB();
            const bool has_sc = dsc != nullptr;
            const ConvDesc dS = has_sc ? *dsc : ConvDesc();
            add([d3, d1, dS, has_sc, self, g_mid, g_out, pm, po](hipStream_t s, int batch) {
                ConvDesc x3 = d3, x1 = d1, xs = dS;
                x3.B = x1.B = xs.B = batch;
                if (self->calib_phase) {
                    if (has_sc) { conv_forward(s, xs); self->observe_split(s, g_out, dS.out, po * batch); }
                    conv_forward(s, x3); self->observe_split(s, g_mid, d3.out, pm * batch);
                    conv_forward(s, x1); self->observe_split(s, g_out, d1.out, po * batch);
                } else conv_forward_tail(s, x3, x1, has_sc ? &xs : nullptr);
            });
        };

        // fp16 mode — an identity bottleneck (every non-first block of a stage) as ONE persistent launch with both branch tensors on
        // chip (kernels_bneck.hip; bit-identical to the three launches, which conv_bneck_forward runs when the block does not qualify
        // or behind mrcnn_debug_set("conv_bneck", 0)).  The fused form reads halo pixels of x that neighbouring tiles own, so it cannot
        // write in place: such stages ping-pong between two block-output tensors.
        // Round 6: the consecutive identity blocks of a stage are ONE op — where every block takes the fragment-streaming form (C4) they
        // run as ONE launch whose tiles wait for their neighbours' previous block (conv_bneck_stage_forward; bit-identical to the
        // per-block launches, which it falls back to).  The launch reads its per-block operands from a table on the device.
        std::vector<BneckTriple> stage_blocks;
        auto bneck_op = [&](const std::string& na, const std::string& nb, const std::string& nc, const Tensor4& xin, const Tensor4& ta, const Tensor4& tb,
                            const Tensor4& out, int g_in, int g_a, int g_b, int g_out) {
            BneckTriple t;
            t.a = make_desc(na, xin, ta, 1, 0, ACT_RELU, nullptr, 0, g_in, g_a);
            t.b = make_desc(nb, ta, tb, 1, 1, ACT_RELU, nullptr, 0, g_a, g_b);
            t.c = make_desc(nc, tb, out, 1, 0, ACT_RELU, &xin, 0, g_b, g_out);
            stage_blocks.push_back(t);
        };
        auto bneck_stage_flush = [&](int tiles_per_image) {
            if (stage_blocks.empty()) return;
            const std::vector<BneckTriple> blocks = stage_blocks;
            stage_blocks.clear();
            void* tab = nullptr;
            unsigned* done = nullptr;
            bool frag = blocks.size() >= 2;
            for (auto& t : blocks) frag = frag && t.a.wgt_frag && t.b.wgt_frag && t.c.wgt_frag;
            if (real && frag) {
                const size_t rb = bneck_layer_record_bytes();
                std::vector<unsigned char> h(rb * blocks.size());
                for (size_t i = 0; i < blocks.size(); ++i)
                    bneck_layer_record(h.data() + i * rb, blocks[i].a.wgt_frag, blocks[i].b.wgt_frag, blocks[i].c.wgt_frag, blocks[i].a.scale, blocks[i].a.shift,
                                       blocks[i].b.scale, blocks[i].b.shift, blocks[i].c.scale, blocks[i].c.shift);
                stage_tabs.emplace_back(new DevBuf(h.size() + (size_t)Bm * tiles_per_image * sizeof(unsigned)));
                HIP_CHECK(hipMemcpy(stage_tabs.back()->p, h.data(), h.size(), hipMemcpyHostToDevice));
                tab = stage_tabs.back()->p;
                done = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(tab) + h.size());
            }
            add([blocks, tab, done](hipStream_t s, int batch) {
                std::vector<BneckTriple> b = blocks;
                for (auto& t : b) t.a.B = t.b.B = t.c.B = batch;
                conv_bneck_stage_forward(s, b.data(), (int)b.size(), tab, done);
            });
        };

        d_rgb = (uint8_t*)ar.alloc_b((size_t)Bm * H * W * 3);
        // C1: zero-padded NHWC4 (fp32) / NHWC8 (fp16) staging — 16 B per pixel either way — so the 7×7/2 conv
        // is 7 row-taps of one contiguous 128-B run each; then the 3×3/2 max pool
        const int Hp = H + 6, Wp = W + 6;
        const int pxc = dt == MRCNN_F16 ? 8 : 4;
        void* x0 = ar.alloc_b((size_t)Bm * Hp * Wp * 16 + 256);
        stem_in = x0;
        {
            const float m3[3] = {mean[0], mean[1], mean[2]};
            uint8_t* src = d_rgb;
            const int h = H, w = W;
            add([=](hipStream_t s, int batch) { preprocess_forward(s, src, batch, h, w, 3, m3, x0, dt); });
        }
        Tensor4 c1 = T(H / 2, W / 2, 64);
        Tensor4 x_after_stem;
        const int g_c1 = new_split_group("C1");                  // conv1's output and its max-pooled version (max-pool commutes with the scaling)
        {
            const PackedConv* pc = &convs.at("conv1");
            ConvDesc d;
            d.dtype = dt;
            d.in = x0; d.H = Hp; d.W = Wp; d.Cin = 8 * pxc;
            d.in_sW = pxc; d.in_sH = (long)Wp * pxc; d.in_sB = (long)Hp * Wp * pxc;
            d.wdtype = pc->wdtype;
            d.wgt = pc->wgt.p; d.KH = 7; d.KW = 1; d.stride = 2; d.padH = d.padW = 0;
            ScaledOp* const so = real ? new_scaled_op(pc, g_img, g_c1) : nullptr;
            d.scale = so ? so->scale.as<float>() : nullptr; d.shift = so ? so->shift.as<float>() : nullptr;
            d.OH = c1.H; d.OW = c1.W; d.Cout = pc->Cout; d.Npad = pc->Npad;
            d.out = c1.p; d.out_sP = c1.C; d.out_sB = c1.sB(); d.act = ACT_RELU;
            d.algo_k = 147;   // 7*7*3 real taps (the packed row is padded to 7*32)
            d.group = 1;
            // split modes: conv1 and the max-pool behind it as ONE persistent launch (kernels_conv_stem.hip) — c1 is then neither
            // written nor read; bit-identical to the two launches, which remain for the other modes and behind "conv_stem" 0
            const size_t per_image = (size_t)c1.sB();
            const int g = g_c1;
            Tensor4 xo = T(H / 4, W / 4, 64);
            const bool stem = conv_stem_eligible(d);
            add([d, self, g, per_image, stem, xo, c1, dt](hipStream_t s, int batch) {
                ConvDesc x = d; x.B = batch;
                if (stem && conv_stem_enabled()) {
                    conv_stem_forward(s, x, xo.p, xo.H, xo.W);
                    if (self->calib_phase) self->observe_split(s, g, xo.p, (size_t)xo.sB() * batch);     // (max over the pooled tensor = max over c1: every element of c1 lies in a window)
                    return;
                }
                conv_forward(s, x);
                if (self->calib_phase) self->observe_split(s, g, d.out, per_image * batch);
                maxpool3x3s2_forward(s, c1.p, batch, c1.H, c1.W, c1.C, xo.p, xo.H, xo.W, dt);
            });
            x_after_stem = xo;
        }
        Tensor4 x = x_after_stem;
        Tensor4 Cf[6];
        const int f1s[6] = {0, 0, 64, 128, 256, 512}, f3s[6] = {0, 0, 256, 512, 1024, 2048};
        int g_x = g_c1;                                           // group of the running tensor x
        int g_C[6] = {-1, -1, -1, -1, -1, -1};
        for (int st = 2; st <= 5; ++st) {
            Tensor4 stage_ta, stage_tb, stage_main, stage_alt;
            bool stage_fused = false, x_is_main = true;
            // every block output of a stage shares ONE group: the blocks add their shortcut in place (res == out)
            const int g_stage = new_split_group("C" + std::to_string(st));
            for (auto& b : blocks[st]) {
                const std::string p = std::to_string(st) + b;
                const bool first = b == "a";
                const int stride = (first && st > 2) ? 2 : 1;
                const int oh = x.H / stride, ow = x.W / stride;
                if (first) { stage_ta = T(oh, ow, f1s[st]); stage_tb = T(oh, ow, f1s[st]); }      // the branch tensors are reused by every block of the stage
                if (first && mode == MRCNN_F16 && blocks[st].size() > 1 && bneck_geometry_ok(f1s[st], oh, ow)) {
                    stage_fused = true;
                    stage_alt = T(oh, ow, f3s[st]);
                }
                const int g_a = new_split_group("res" + p + "_branch2a"), g_b = new_split_group("res" + p + "_branch2b");
                Tensor4 ta = stage_ta;
                Tensor4 tb = stage_tb;
                if (!first && stage_fused) {
                    const Tensor4 to = x_is_main ? stage_alt : stage_main;
                    bneck_op("res" + p + "_branch2a", "res" + p + "_branch2b", "res" + p + "_branch2c", x, ta, tb, to, g_x, g_a, g_b, g_stage);
                    x = to;
                    x_is_main = !x_is_main;
                    g_x = g_stage;
                    continue;
                }
                if (first && mode == MRCNN_F16 && stride == 1 && x.C == f1s[st] && f1s[st] == 64 && bneck_geometry_ok(f1s[st], oh, ow)) {
                    // fp16 mode, the entry block of C2: branch2a + 2b + 2c AND the shortcut convolution branch1 in one launch (kernels_bneck.hip, FIRST
                    // form; bit-identical to the four launches, which conv_bneck_first_forward runs where the block does not qualify)
                    const Tensor4 sc1 = T(oh, ow, f3s[st]);         // the shortcut tensor of the four-launch form (unused by the fused one)
                    const Tensor4 out1 = T(oh, ow, f3s[st]);
                    const ConvDesc da = make_desc("res" + p + "_branch2a", x, ta, 1, 0, ACT_RELU, nullptr, 0, g_x, g_a);
                    const ConvDesc db = make_desc("res" + p + "_branch2b", ta, tb, 1, 1, ACT_RELU, nullptr, 0, g_a, g_b);
                    const ConvDesc ds = make_desc("res" + p + "_branch1", x, sc1, 1, 0, ACT_NONE, nullptr, 0, g_x, g_stage);
                    const ConvDesc dc = make_desc("res" + p + "_branch2c", tb, out1, 1, 0, ACT_RELU, &sc1, 0, g_b, g_stage);
                    add([da, db, dc, ds](hipStream_t s, int batch) {
                        ConvDesc a = da, b = db, c = dc, e = ds;
                        a.B = b.B = c.B = e.B = batch;
                        conv_bneck_first_forward(s, a, b, c, e);
                    });
                    x = out1;
                    stage_main = out1;
                    g_x = g_stage;
                    continue;
                }
                conv_op("res" + p + "_branch2a", x, ta, stride, 0, ACT_RELU, nullptr, 0, g_x, g_a);
                Tensor4 sc = x;
                ConvDesc dsc;
                if (first) {
                    sc = T(oh, ow, f3s[st]);
                    dsc = make_desc("res" + p + "_branch1", x, sc, stride, 0, ACT_NONE, nullptr, 0, g_x, g_stage);
                }
                // The block's output overwrites its shortcut IN PLACE (branch2c reads a residual element and writes the output
                // element at the same address, from the same thread; nothing reads the shortcut afterwards): a stage then cycles
                // through x + two branch tensors — 200 MB for C4 at batch 8, inside the 256 MB Infinity Cache — instead of
                // streaming a fresh 134 MB tensor per block through HBM.
                Tensor4 to = sc;
                tail_op("res" + p + "_branch2b", "res" + p + "_branch2c", ta, tb, to, sc, g_a, g_b, g_stage, first ? &dsc : nullptr);
                x = to;
                if (first) stage_main = to;
                g_x = g_stage;
            }
            bneck_stage_flush((x.H / (f1s[st] == 256 ? 8 : 16)) * (x.W / 16));
            Cf[st] = x;
            g_C[st] = g_stage;
        }
        // FPN: lateral 1×1 (+ nearest 2× upsample of the level above, fused as a shifted residual), then 3×3
        Tensor4 L5 = T(Cf[5].H, Cf[5].W, 256), L4 = T(Cf[4].H, Cf[4].W, 256), L3 = T(Cf[3].H, Cf[3].W, 256), L2 = T(Cf[2].H, Cf[2].W, 256);
        const int g_L = new_split_group("fpn_lateral");           // L5..L2 add each other (top-down): one group
        conv_op("fpn_c5p5", Cf[5], L5, 1, 0, ACT_NONE, nullptr, 0, g_C[5], g_L);
        conv_op("fpn_c4p4", Cf[4], L4, 1, 0, ACT_NONE, &L5, 1, g_C[4], g_L);
        conv_op("fpn_c3p3", Cf[3], L3, 1, 0, ACT_NONE, &L4, 1, g_C[3], g_L);
        conv_op("fpn_c2p2", Cf[2], L2, 1, 0, ACT_NONE, &L3, 1, g_C[2], g_L);
        const Tensor4 Ls[4] = {L2, L3, L4, L5};
        const char* pn[4] = {"fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5"};
        const char* tn[4] = {"P2", "P3", "P4", "P5"};
        for (int l = 0; l < 4; ++l) {
            P[l] = T(Ls[l].H, Ls[l].W, 256);
            g_P[l] = new_split_group(tn[l]);
            conv_op(pn[l], Ls[l], P[l], 1, 1, ACT_NONE, nullptr, 0, g_L, g_P[l]);
            taps[tn[l]] = {P[l].p, P[l].sB(), dt, g_P[l]};
            MRCNN_REQUIRE(P[l].H == fh[l] && P[l].W == fw[l], MRCNN_ERR_SHAPE, "pyramid level %d shape mismatch", l + 2);
        }
        // RPN on P2..P6 (P6 = P5 sub-sampled by 2: read in place through doubled strides)
        rpn_logits = ar.alloc_f((size_t)Bm * A * 2);
        rpn_probs = ar.alloc_f((size_t)Bm * A * 2);
        rpn_deltas = ar.alloc_f((size_t)Bm * A * 4);
        taps["rpn_probs"] = {rpn_probs, (long)A * 2, MRCNN_F32};
        taps["rpn_deltas"] = {rpn_deltas, (long)A * 4, MRCNN_F32};
        // the shared layer's 512-channel tensor: ONE buffer, sized for the largest level, reused level after level on the model's stream
        // (round 5's per-level copies served the level-parallel region, measured slower and removed in round 6: profiles/r05_level_parallel_ab.txt)
        void* const rpn_feat = ar.alloc_e((size_t)Bm * fh[0] * fw[0] * 512, dt);
        for (int l = 0; l < 5; ++l) {
            const Tensor4& src = P[l < 4 ? l : 3];
            const int sub = l < 4 ? 1 : 2;
            const PackedConv* pc = &convs.at("rpn_conv_shared");
            g_rpn[l] = new_split_group("rpn_feat_P" + std::to_string(l + 2));
            ScaledOp* const so_d = real ? new_scaled_op(pc, g_P[l < 4 ? l : 3], g_rpn[l]) : nullptr;
            ConvDesc d;
            d.dtype = dt;
            d.in = src.p; d.H = fh[l]; d.W = fw[l]; d.Cin = 256;
            d.in_sW = (long)sub * src.C; d.in_sH = (long)sub * src.W * src.C; d.in_sB = src.sB();
            d.wdtype = pc->wdtype;
            d.wgt = pc->wgt.p; d.KH = 3; d.KW = 3; d.stride = 1; d.padH = d.padW = 1;
            d.wgt_halo = pc->wgt_halo.p;
            d.wgt_c3h = pc->wgt_c3h.p;
            d.scale = so_d ? so_d->scale.as<float>() : nullptr; d.shift = so_d ? so_d->shift.as<float>() : nullptr;
            d.OH = fh[l]; d.OW = fw[l]; d.Cout = 512; d.Npad = pc->Npad;
            d.out = rpn_feat; d.out_sP = 512; d.out_sB = (long)fh[l] * fw[l] * 512; d.act = ACT_RELU;
            const PackedConv* hc = &convs.at("rpn_heads");
            ScaledOp* const so_e = real ? new_scaled_op(hc, g_rpn[l], g_zero) : nullptr;     // the separate head launch undoes the exponent in its scale
            ConvDesc e;
            e.dtype = dt; e.out_f32 = 1;      // the box path consumes fp32 (ProposalLayer.swift:108-109)
            e.in = rpn_feat; e.H = fh[l]; e.W = fw[l]; e.Cin = 512;
            e.in_sW = 512; e.in_sH = (long)fw[l] * 512; e.in_sB = (long)fh[l] * fw[l] * 512;
            e.wdtype = hc->wdtype;
            e.wgt = hc->wgt.p; e.scale = so_e ? so_e->scale.as<float>() : nullptr; e.shift = so_e ? so_e->shift.as<float>() : nullptr;
            e.OH = fh[l]; e.OW = fw[l]; e.Cout = hc->Cout; e.Npad = hc->Npad;
            e.out = rpn_logits + lvl_off[l] * 2; e.out_sP = 2 * na; e.out_sB = (long)A * 2;
            e.out2 = rpn_deltas + lvl_off[l] * 4; e.out2_sP = 4 * na; e.out2_sB = (long)A * 4; e.n_split = 2 * na;
            // Levels whose geometry qualifies (a property of the level, not of the batch: P2..P4 at 1024²) run the two heads INSIDE
            // the shared 3x3 layer's epilogue: the 512-channel tensor — 1.07 GB at P2, batch 8 — is neither
            // written nor read back, and the head launches disappear.  Switched off with the halo kernel itself: the two launches.
            ConvDesc f = d;
            f.B = 1;
            f.head_w = rpn_head_frag.p; f.head_bias = hc->shift.as<float>();
            f.head_out = rpn_logits + lvl_off[l] * 2; f.head_out_sP = 2 * na; f.head_out_sB = (long)A * 2;
            f.head_out2 = rpn_deltas + lvl_off[l] * 4; f.head_out2_sP = 4 * na; f.head_out2_sB = (long)A * 4;
            f.head_split = 2 * na; f.head_cols = 6 * na;
            // fp16 mode: the same fusion on the halo-tile kernel of kernels_conv3x3_h.hip, on the levels with >= 16384 pixels per image (P2, P3 at 1024^2 — P4's 16 tiles per image would leave half the chip idle at batch 8:
            // a property of the level, like the split modes' rule); bit-identical to that kernel followed by the separate head launch
            const bool fuse16 = mode == MRCNN_F16 && rpn_head_frag.p && d.wgt_c3h && (long)fh[l] * fw[l] >= 16384 && conv3x3h_eligible(f);
            const bool fuse = fuse16 || (mode != MRCNN_F16 && rpn_head_frag.p && d.wgt_halo && conv_halo_head_eligible(f));
            const int g_r = g_rpn[l];
            const size_t per_image = (size_t)fh[l] * fw[l] * 512;
            add([d, e, f, fuse, fuse16, self, g_r, per_image](hipStream_t s, int batch) {
                // (a calibration pass needs the 512-channel tensor: it runs the two-launch form)
                const int c3h = conv_c3h_mode();
                if (fuse && (fuse16 ? (c3h == 1 || c3h == 2) : conv_halo_enabled()) && !self->calib_phase) {
                    ConvDesc x = f; x.B = batch;
                    x.head_mul = ldexpf(1.0f, -self->sgroups[(size_t)g_r].exp);       // the fused heads see 2^e * relu(...): undone on their sums (exact)
                    conv_forward(s, x);
                    return;
                }
                ConvDesc x = d; x.B = batch;
                x.prefer_c3h = fuse16 && c3h == 3;               // (the fused form's own 3x3 kernel followed by the separate head launch: bit-identical to the fusion)
                conv_forward(s, x);
                if (self->calib_phase) self->observe_split(s, g_r, d.out, per_image * batch);
                ConvDesc y = e; y.B = batch; conv_forward(s, y);
            });
        }
        {
            float* lg = rpn_logits; float* pr = rpn_probs; const long per = A;
            add([=](hipStream_t s, int batch) { softmax_pairs_forward(s, lg, pr, per * batch); });
        }
        rois = ar.alloc_f((size_t)Bm * max_prop * 4);
        pooled = ar.alloc_e((size_t)Bm * max_prop * cls_pool * cls_pool * 256, dt);
        cls6 = ar.alloc_f((size_t)Bm * max_prop * 6);
        detections = ar.alloc_f((size_t)Bm * max_det * 6);
        pooled_mask = ar.alloc_e((size_t)Bm * max_det * mask_pool * mask_pool * 256, dt);
        mask_out = ar.alloc_f((size_t)Bm * max_det * 4 * mask_pool * mask_pool);
        taps["rois"] = {rois, (long)max_prop * 4, MRCNN_F32};
        g_pooled = new_split_group("pooled");
        g_pooled_mask = new_split_group("pooled_mask");
        taps["pooled"] = {pooled, (long)max_prop * cls_pool * cls_pool * 256, dt, g_pooled};
        taps["cls6"] = {cls6, (long)max_prop * 6, MRCNN_F32};
        taps["detections"] = {detections, (long)max_det * 6, MRCNN_F32};
        taps["pooled_mask"] = {pooled_mask, (long)max_det * mask_pool * mask_pool * 256, dt, g_pooled_mask};
        taps["mask"] = {mask_out, (long)max_det * 4 * mask_pool * mask_pool, MRCNN_F32};
        void* pws = ar.alloc_b(ProposalWorkspace::bytes(Bm, A, K, max_prop));
        void* dws = ar.alloc_b(DetectionWorkspace::bytes(Bm, max_prop, max_det));
        msel_ws.flags = (int32_t*)ar.alloc_b((size_t)Bm * max_det * 4);
        msel_ws.mapping = (int32_t*)ar.alloc_b((size_t)Bm * max_det * 4);
        msel_ws.kept = (int32_t*)ar.alloc_b((size_t)Bm * 4);
        msel_ws.sel_cid = (int32_t*)ar.alloc_b((size_t)Bm * max_det * 4);
        mask_partial = ar.alloc_f((size_t)Bm * max_det * 4 * mask_pool * mask_pool * (256 / 64));        // (up to four partial sums per output pixel: conv_sel_part_cols)
        if (real) {
            prop_ws.bind(pws, Bm, A, K, max_prop);
            det_ws.bind(dws, Bm, max_prop, max_det);
        }
        if (real) {
            // the heads' layers: pooled → h1 → h2 → logits (0);  pooled_mask → t1..t4 → deconvolution output (0: its consumer is an fp32 dot)
            cls_head.grp[0] = g_pooled; cls_head.grp[1] = new_split_group("cls_h1"); cls_head.grp[2] = new_split_group("cls_h2");
            mask_head.grp[0] = g_pooled_mask;
            for (int i = 1; i <= 4; ++i) mask_head.grp[i] = new_split_group("mask_t" + std::to_string(i));
            for (int i = 0; i < 3; ++i) { cls_head.sop[i].g_in = cls_head.grp[i]; cls_head.sop[i].g_out = i < 2 ? cls_head.grp[i + 1] : g_zero; }
            for (int i = 0; i < 5; ++i) { mask_head.sop[i].g_in = mask_head.grp[i]; mask_head.sop[i].g_out = i < 4 ? mask_head.grp[i + 1] : g_zero; }
            cls_head.observe = [self](hipStream_t s, int g, const void* x, size_t n) { if (self->calib_phase) self->observe_split(s, g, x, n); };
            mask_head.observe = cls_head.observe;
        }
        if (pass == 0) { arena.alloc(ar.off); ar.base = arena.as<char>(); }
    }
    calib_buf.alloc(sgroups.size() * 32);
    HIP_CHECK(hipMemset(calib_buf.p, 0, sgroups.size() * 32));
    const bool split_mode = mode == MRCNN_F32S || mode == MRCNN_F32X3;
    if (split_mode) {
        // exponents stored with the artefact (convert.py --calibrate: "split_exp.<group>" metadata): the drop-in prediction() needs no call
        int found = 0;
        for (auto& g : sgroups) {
            const auto it = file.ints.find("split_exp." + g.name);
            if (it == file.ints.end() || g.fixed) continue;
            MRCNN_REQUIRE(it->second >= -60 && it->second <= 60, MRCNN_ERR_IO, "stored split exponent of group '%s' out of range", g.name.c_str());
            g.exp = (int)it->second;
            ++found;
        }
        if (found) { apply_split_exponents(); split_calibrated = true; exponents_from_artefact = true; }
    }
    if (const char* e = knob_env("MRCNN_SPLIT_EXP")) {          // measurement / tests: one exponent for every non-fixed group (split modes only)
        if (split_mode) {
            for (auto& g : sgroups) if (!g.fixed) g.exp = atoi(e);
            apply_split_exponents();
        } else fprintf(stderr, "libmaskrcnn_hip: MRCNN_SPLIT_EXP ignored: compute mode %d carries no split exponents\n", mode);
    }
    taps["cls_probs"] = {cls_head.probs, (long)max_prop * nc, MRCNN_F32};
    taps["cls_bbox"] = {cls_head.bbox, (long)max_prop * nc * 4, MRCNN_F32};
}

// ------------------------------------------------------------------------------------------------
// scale-aware split (engine.h: SplitGroup)
// ------------------------------------------------------------------------------------------------
int Model::new_split_group(const std::string& name, bool fixed)
{
    SplitGroup g;
    g.name = name; g.fixed = fixed;
    sgroups.push_back(g);
    return (int)sgroups.size() - 1;
}

ScaledOp* Model::new_scaled_op(const PackedConv* pc, int g_in, int g_out)
{
    sops.emplace_back(new ScaledOp);
    init_scaled_op(*sops.back(), pc, g_in, g_out);
    return sops.back().get();
}

// max |x| of a tensor (phase 1) / how many of its non-zero elements the three-part split cannot carry exactly (phase 2):
// slot = [uint32 bits of the maximum | pad | small | inexact | counted] (32 B per group)
__global__ __launch_bounds__(256) void k_split_observe(const float* __restrict__ x, size_t n, int phase, float small_thr, unsigned* __restrict__ slot)
{
    const size_t stride = (size_t)gridDim.x * 256;
    float mx = 0.f;
    unsigned long long small = 0, inexact = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float a = fabsf(x[i]);
        if (phase == 1) mx = fmaxf(mx, a);
        else if (a != 0.f) { small += a < small_thr; inexact += a < 0.5f; }
    }
    __shared__ float s_mx[256];
    __shared__ unsigned long long s_a[256], s_b[256];
    s_mx[threadIdx.x] = mx; s_a[threadIdx.x] = small; s_b[threadIdx.x] = inexact;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_mx[threadIdx.x] = fmaxf(s_mx[threadIdx.x], s_mx[threadIdx.x + w]);
            s_a[threadIdx.x] += s_a[threadIdx.x + w];
            s_b[threadIdx.x] += s_b[threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (phase == 1) atomicMax(slot, __float_as_uint(s_mx[0]));        // non-negative floats order like their bit patterns (NaN / inf sort above everything: reported)
        else {
            unsigned long long* c = reinterpret_cast<unsigned long long*>(slot + 2);
            atomicAdd(c, s_a[0]);
            atomicAdd(c + 1, s_b[0]);
            if (blockIdx.x == 0) atomicAdd(c + 2, (unsigned long long)n);
        }
    }
}

void Model::observe_split(hipStream_t s, int group, const void* x, size_t n)
{
    if (!calib_phase || group < 0 || n == 0 || dtype != MRCNN_F32) return;
    const SplitGroup& g = sgroups[(size_t)group];
    if (g.fixed) return;
    // phase 2 sees the stored (scaled) tensor: "small" = below 2^-8 of the group's maximum, "inexact" = below 0.5
    const float small_thr = ldexpf(g.absmax, g.exp - 8);
    const int grid = (int)((n + 256 * 16 - 1) / (256 * 16) < 2048 ? (n + 256 * 16 - 1) / (256 * 16) : 2048);
    hipLaunchKernelGGL(k_split_observe, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, static_cast<const float*>(x), n, calib_phase, small_thr,
                       calib_buf.as<unsigned>() + (size_t)group * 8);
    HIP_CHECK(hipGetLastError());
}

void Model::apply_split_exponents()
{
    HIP_CHECK(hipStreamSynchronize(stream));
    drop_graphs();                                   // the fused heads' multiplier and the samplers' are launch arguments
    auto ex = [&](int g) { return g >= 0 ? sgroups[(size_t)g].exp : 0; };
    auto refresh = [&](ScaledOp& op) {
        const int ds = ex(op.g_out) - ex(op.g_in), eo = ex(op.g_out);
        std::vector<float> sc(op.pc->h_scale), sh(op.pc->h_shift);
        for (auto& v : sc) v = ldexpf(v, ds);        // exact: a power of two (a folded BatchNorm scale is nowhere near the fp32 range limits)
        for (auto& v : sh) v = ldexpf(v, eo);
        HIP_CHECK(hipMemcpy(op.scale.p, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(op.shift.p, sh.data(), sh.size() * 4, hipMemcpyHostToDevice));
    };
    for (auto& op : sops) refresh(*op);
    for (auto& op : cls_head.sop) refresh(op);
    for (auto& op : mask_head.sop) refresh(op);
}

int Model::split_exponent_for(float absmax)
{
    if (!(absmax > 0.f)) return 0;
    int k = 0;
    (void)frexpf(absmax, &k);                      // absmax = m * 2^k, 0.5 <= m < 1
    const int e = 12 - k;
    return e < -24 ? -24 : (e > 48 ? 48 : e);
}

void Model::measure_split_groups(const uint8_t* rgb, int batch, int h, int w, int memspace, bool fit, bool resident)
{
    const size_t G = sgroups.size();
    std::vector<unsigned> raw(G * 8);
    // true magnitudes.  Every exponent 0 — or, when an activation of THIS model leaves the fp16 range there (the watchdog fails the
    // pass: its maxima would be garbage), one uniform negative exponent that keeps it inside
    int base = 0;
    for (;;) {
        for (auto& g : sgroups) g.exp = g.fixed ? 0 : base;
        apply_split_exponents();
        HIP_CHECK(hipMemset(calib_buf.p, 0, G * 32));
        calib_phase = 1;
        bool tripped = false;
        try {
            predict(rgb, batch, h, w, memspace, nullptr, nullptr, true, fit, resident);
        } catch (const Error& e) {
            calib_phase = 0;
            if (e.code != MRCNN_ERR_UNSUPPORTED || base <= -36) throw;
            tripped = true;
        } catch (...) {
            calib_phase = 0;
            throw;
        }
        calib_phase = 0;
        if (!tripped) break;
        base -= 6;
    }
    HIP_CHECK(hipMemcpy(raw.data(), calib_buf.p, G * 32, hipMemcpyDeviceToHost));
    for (size_t g = 0; g < G; ++g) {
        float mx;
        memcpy(&mx, &raw[g * 8], 4);
        mx = ldexpf(mx, -sgroups[g].exp);               // stored = 2^e * value
        sgroups[g].absmax = mx;
        MRCNN_REQUIRE(mx == mx && mx < 3.0e38f, MRCNN_ERR_UNSUPPORTED, "%s: tensor group '%s' holds inf / NaN on this batch (an fp32 engine would produce them too)",
                      resident ? "range recovery of a predict (the batch was measured where it sits, after an activation left the fp16 range)" : "calibrate_split",
                      sgroups[g].name.c_str());
    }
}

void Model::calibrate_split(const uint8_t* rgb, int batch, int h, int w, int memspace, bool apply)
{
    MRCNN_REQUIRE(kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "calibrate_split called on a non-MaskRCNN model");
    MRCNN_REQUIRE(mode == MRCNN_F32S || mode == MRCNN_F32X3, MRCNN_ERR_UNSUPPORTED,
                  "calibrate_split: only the split modes (MRCNN_F32S, MRCNN_F32X3) carry activations through fp16 parts; this model is scale-invariant as it is");
    MRCNN_REQUIRE(batch >= 1 && batch <= max_batch, MRCNN_ERR_SHAPE, "batch %d outside 1..%d", batch, max_batch);
    const size_t G = sgroups.size();
    std::vector<unsigned> raw(G * 8);
    std::vector<int> saved(G);
    for (size_t g = 0; g < G; ++g) saved[g] = sgroups[g].exp;
    try {
        // ---- phase 1: true magnitudes ----
        measure_split_groups(rgb, batch, h, w, memspace, false, false);
        // ---- exponents: max |a| * 2^e in [2^11, 2^12) — 16x head room below the fp16 range, 0.5 = 2^-13 of the maximum exact ----
        std::vector<int> chosen(G, 0);
        for (size_t g = 0; g < G; ++g)
            if (!sgroups[g].fixed) chosen[g] = split_exponent_for(sgroups[g].absmax);
        // ---- phase 2: with the chosen exponents — verifies them (range watchdog) and counts what a split still cannot carry.
        //      (The images are resident from phase 1: no second copy, and a device-side caller's buffer is read once.) ----
        for (size_t g = 0; g < G; ++g) sgroups[g].exp = chosen[g];
        apply_split_exponents();
        HIP_CHECK(hipMemset(calib_buf.p, 0, G * 32));
        calib_phase = 2;
        try {
            predict(rgb, batch, h, w, memspace, nullptr, nullptr, true, false, true);
        } catch (...) {
            calib_phase = 0;
            throw;
        }
        calib_phase = 0;
        HIP_CHECK(hipMemcpy(raw.data(), calib_buf.p, G * 32, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < G; ++g) {
            unsigned long long c[3];
            memcpy(c, &raw[g * 8 + 2], 24);
            sgroups[g].small = (long long)c[0]; sgroups[g].inexact = (long long)c[1]; sgroups[g].counted = (long long)c[2];
        }
        if (!apply) {
            for (size_t g = 0; g < G; ++g) sgroups[g].exp = saved[g];
            apply_split_exponents();
        } else split_calibrated = true;
    } catch (...) {
        for (size_t g = 0; g < G; ++g) sgroups[g].exp = saved[g];
        apply_split_exponents();
        throw;
    }
}

bool Model::recover_range(int batch, int h, int w, bool fit)
{
    if (!(mode == MRCNN_F32S || mode == MRCNN_F32X3) || kind != MRCNN_MODEL_MASKRCNN) return false;
    const size_t G = sgroups.size();
    std::vector<int> before(G);
    for (size_t g = 0; g < G; ++g) before[g] = sgroups[g].exp;
    try {
        measure_split_groups(nullptr, batch, h, w, MRCNN_DEVICE, fit, true);
    } catch (const Error& e) {
        for (size_t g = 0; g < G; ++g) sgroups[g].exp = before[g];
        apply_split_exponents();
        // (the caller asked for a predict, not for a calibration: say what happened in ITS terms — ADVICE r5)
        fail(e.code, "predict: an activation left the calibrated fp16 range and the batch could not be measured for a range recovery (%s); the split "
             "exponents are unchanged, the results of this call are not valid", e.msg.c_str());
    } catch (...) {
        for (size_t g = 0; g < G; ++g) sgroups[g].exp = before[g];
        apply_split_exponents();
        throw;
    }
    // never raise an exponent: what earlier batches needed still fits
    for (size_t g = 0; g < G; ++g) {
        if (sgroups[g].fixed) { sgroups[g].exp = 0; continue; }
        const int need = split_exponent_for(sgroups[g].absmax);
        sgroups[g].exp = need < before[g] ? need : before[g];
    }
    apply_split_exponents();
    return true;
}

void Model::predict(const uint8_t* rgb, int batch, int h, int w, int memspace, float* det_out, float* masks_out, bool sync, bool fit, bool resident)
{
    MRCNN_REQUIRE(kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "predict called on a non-MaskRCNN model");
    MRCNN_REQUIRE((rgb || resident) && ((det_out && masks_out) || calib_phase), MRCNN_ERR_INVALID, "null buffer");
    MRCNN_REQUIRE(fit ? (h > 0 && w > 0 && h < 32768 && w < 32768) : (h == H && w == W), MRCNN_ERR_SHAPE,
                  "image is %dx%d, the model expects %dx%d (mrcnn_maskrcnn_predict_scalefit letterboxes any size)", h, w, H, W);
    MRCNN_REQUIRE(batch >= 1 && batch <= max_batch, MRCNN_ERR_SHAPE, "batch %d outside 1..%d", batch, max_batch);
    hipStream_t s = stream;
    int fitgeo[6] = {h, w, 0, 0, 0, 0};
    if (fit) {
        MRCNN_REQUIRE(mrcnn_letterbox_geometry(h, w, H, W, &fitgeo[2], &fitgeo[3], &fitgeo[4], &fitgeo[5]) == MRCNN_OK, MRCNN_ERR_INVALID, "bad letterbox geometry");
        const size_t need = (size_t)batch * h * w * 3;
        if (fit_src.bytes < need) { HIP_CHECK(hipStreamSynchronize(s)); fit_src.alloc(need); drop_graphs(); }
    }
    const size_t img_bytes = (size_t)batch * h * w * 3;
    uint8_t* const img_dst = fit ? fit_src.as<uint8_t>() : d_rgb;
    if (!ev_p0) { HIP_CHECK(hipEventCreate(&ev_p0)); HIP_CHECK(hipEventCreate(&ev_p1)); }
    hipStreamCaptureStatus outer_capture = hipStreamCaptureStatusNone;
    HIP_CHECK(hipStreamIsCapturing(s, &outer_capture));
    const bool timed_call = sync && outer_capture == hipStreamCaptureStatusNone;
    if (timed_call) HIP_CHECK(hipEventRecord(ev_p0, s));
    if (!resident) HIP_CHECK(hipMemcpyAsync(img_dst, rgb, img_bytes, memspace == MRCNN_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));

    hipStreamCaptureStatus caller_capture = hipStreamCaptureStatusNone;
    if (use_graph) HIP_CHECK(hipStreamIsCapturing(s, &caller_capture));
    // (the scale-fit geometry is a launch argument of the first kernel: such calls are not replayed from a captured graph)
    const bool plain = !use_graph || fit || timer.enabled || conv_profile.active || calib_phase || caller_capture != hipStreamCaptureStatusNone;
    if (plain) {
        enqueue_pipeline(s, batch, fit ? fitgeo : nullptr);
    } else {
        GraphSlot& g = graphs[batch];
        if (!g.exec && g.eager_runs++ == 0) {
            enqueue_pipeline(s, batch);              // first call at this batch size: also runs the one-time host set-up
        } else {
            if (!g.exec) {
                HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                try {
                    enqueue_pipeline(s, batch);
                } catch (...) {
                    hipGraph_t dead = nullptr;
                    (void)hipStreamEndCapture(s, &dead);
                    if (dead) (void)hipGraphDestroy(dead);
                    throw;
                }
                HIP_CHECK(hipStreamEndCapture(s, &g.graph));
                HIP_CHECK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
            }
            HIP_CHECK(hipGraphLaunch(g.exec, s));
            ++graph_launches;
        }
    }

    const int HW = 4 * mask_pool * mask_pool;
    // (the recovery path of collect() reads device images and writes host records: the direction of the record copies follows the pointers)
    const hipMemcpyKind back = hipMemcpyDefault;
    if (det_out) HIP_CHECK(hipMemcpyAsync(det_out, detections, (size_t)batch * max_det * 6 * 4, back, s));
    if (masks_out) HIP_CHECK(hipMemcpyAsync(masks_out, mask_out, (size_t)batch * max_det * HW * 4, back, s));
    if (timed_call) HIP_CHECK(hipEventRecord(ev_p1, s));
    if (sync) {
        HIP_CHECK(hipStreamSynchronize(s));
        if (timed_call && !calib_phase) {              // (calibration / recovery passes are not the host's predicts)
            float ms = 0;
            HIP_CHECK(hipEventElapsedTime(&ms, ev_p0, ev_p1));
            gpu_busy_ms += ms;
            ++predict_calls;
        }
        timer.finish();
        if (conv_profile.active) conv_profile.collect();
        if (mode != MRCNN_F32) {
            int tripped = 0;
            HIP_CHECK(hipMemcpy(&tripped, range_flag.p, sizeof(int), hipMemcpyDeviceToHost));
            if (tripped & 2)
                fail(MRCNN_ERR_HIP, "a whole-stage bottleneck launch made no progress for ~0.1 s (its grid was not resident as a whole: are compute units masked off or "
                 "held by another process?): the results of this call are not valid; mrcnn_debug_set(\"conv_bneck_stage\", 0) runs one launch per block");
            if (tripped && calib_phase)
                fail(MRCNN_ERR_UNSUPPORTED, "an activation left the fp16 range during a calibration pass");       // (the caller steps the exponents down)
            if (tripped) {
                ++range_overflows;
                // split modes: this batch needs smaller exponents than the calibration images did — measure it where it sits, lower them,
                // compute it again.  The reference's fp32 path handles any such input (Conversion/task.py:90).
                if (outer_capture == hipStreamCaptureStatusNone && recover_range(batch, h, w, fit)) {
                    ++range_recoveries;
                    enqueue_pipeline(s, batch, fit ? fitgeo : nullptr);
                    if (det_out) HIP_CHECK(hipMemcpyAsync(det_out, detections, (size_t)batch * max_det * 6 * 4, back, s));
                    if (masks_out) HIP_CHECK(hipMemcpyAsync(masks_out, mask_out, (size_t)batch * max_det * HW * 4, back, s));
                    HIP_CHECK(hipStreamSynchronize(s));
                    timer.finish();
                    if (conv_profile.active) conv_profile.collect();
                    HIP_CHECK(hipMemcpy(&tripped, range_flag.p, sizeof(int), hipMemcpyDeviceToHost));
                    if (!tripped) return;
                }
                fail(MRCNN_ERR_UNSUPPORTED, "an activation left the fp16 range (|v| >= 65504) in compute mode %s: the results of this call are "
                     "not valid; load the model with MRCNN_F32", mode == MRCNN_F16 ? "MRCNN_F16" : (mode == MRCNN_F32S ? "MRCNN_F32S" : "MRCNN_F32X3"));
            }
        }
    }
}

// ---- pipelined host entry ------------------------------------------------------------------------------------------------
void Model::submit(const uint8_t* rgb_host, int batch, int h, int w)
{
    MRCNN_REQUIRE(kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "submit called on a non-MaskRCNN model");
    MRCNN_REQUIRE(rgb_host, MRCNN_ERR_INVALID, "null buffer");
    MRCNN_REQUIRE(h == H && w == W, MRCNN_ERR_SHAPE, "image is %dx%d, the model expects %dx%d", h, w, H, W);
    MRCNN_REQUIRE(batch >= 1 && batch <= max_batch, MRCNN_ERR_SHAPE, "batch %d outside 1..%d", batch, max_batch);
    MRCNN_REQUIRE(pipe_submitted - pipe_collected < 2, MRCNN_ERR_INVALID, "two batches are in flight already: call mrcnn_maskrcnn_collect first");
    // the stage timer and the conv profile hold ONE batch's events: a second submission would re-record what the first one's collect reads
    MRCNN_REQUIRE(pipe_submitted == pipe_collected || !(timer.enabled || conv_profile.active), MRCNN_ERR_INVALID,
                  "stage timing / the conv profile are on: collect the batch in flight before submitting the next");
    PipeSlot& sl = pipe[pipe_submitted & 1];
    const int HW = 4 * mask_pool * mask_pool;
    if (!pipe_in) {
        HIP_CHECK(hipStreamCreateWithFlags(&pipe_in, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&pipe_out, hipStreamNonBlocking));
    }
    if (!sl.ev_in) {
        HIP_CHECK(hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        sl.rgb.alloc((size_t)max_batch * H * W * 3);
        sl.det.alloc((size_t)max_batch * max_det * 6 * 4);
        sl.mask.alloc((size_t)max_batch * max_det * HW * 4);
        sl.flag.alloc(sizeof(int));
    }
    hipStream_t s = stream;
    // the images cross PCIe on the copy stream (under whatever the compute stream is doing: the previous batch's predict) ...
    HIP_CHECK(hipMemcpyAsync(sl.rgb.p, rgb_host, (size_t)batch * H * W * 3, hipMemcpyHostToDevice, pipe_in));
    HIP_CHECK(hipEventRecord(sl.ev_in, pipe_in));
    // ... and the compute stream picks them up when they have arrived
    HIP_CHECK(hipStreamWaitEvent(s, sl.ev_in, 0));
    HIP_CHECK(hipMemcpyAsync(d_rgb, sl.rgb.p, (size_t)batch * H * W * 3, hipMemcpyDeviceToDevice, s));
    enqueue_pipeline(s, batch);
    HIP_CHECK(hipMemcpyAsync(sl.det.p, detections, (size_t)batch * max_det * 6 * 4, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(sl.mask.p, mask_out, (size_t)batch * max_det * HW * 4, hipMemcpyDeviceToDevice, s));
    if (mode != MRCNN_F32) HIP_CHECK(hipMemcpyAsync(sl.flag.p, range_flag.p, sizeof(int), hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipEventRecord(sl.ev_done, s));
    sl.batch = batch;
    sl.busy = true;
    ++pipe_submitted;
}

void Model::collect(float* det_host, float* masks_host, int* batch_out)
{
    MRCNN_REQUIRE(det_host && masks_host, MRCNN_ERR_INVALID, "null buffer");
    MRCNN_REQUIRE(pipe_submitted > pipe_collected, MRCNN_ERR_INVALID, "nothing was submitted");
    PipeSlot& sl = pipe[pipe_collected & 1];
    const int HW = 4 * mask_pool * mask_pool;
    int tripped = 0;
    // whatever happens below, the slot is handed back: a HIP error here must not leave it "in flight" for ever
    struct Release {
        Model* m; PipeSlot* sl;
        ~Release() { sl->busy = false; ++m->pipe_collected; }
    } release{this, &sl};
    HIP_CHECK(hipStreamWaitEvent(pipe_out, sl.ev_done, 0));
    HIP_CHECK(hipMemcpyAsync(det_host, sl.det.p, (size_t)sl.batch * max_det * 6 * 4, hipMemcpyDeviceToHost, pipe_out));
    HIP_CHECK(hipMemcpyAsync(masks_host, sl.mask.p, (size_t)sl.batch * max_det * HW * 4, hipMemcpyDeviceToHost, pipe_out));
    if (mode != MRCNN_F32) HIP_CHECK(hipMemcpyAsync(&tripped, sl.flag.p, sizeof(int), hipMemcpyDeviceToHost, pipe_out));
    HIP_CHECK(hipStreamSynchronize(pipe_out));
    if (pipe_submitted == pipe_collected + 1) {      // nothing newer has been enqueued: the stage timer / conv profile hold THIS batch's events
        timer.finish();
        if (conv_profile.active) conv_profile.collect();
    }
    if (batch_out) *batch_out = sl.batch;
    ++predict_calls;
    if (tripped & 2)
        fail(MRCNN_ERR_HIP, "a whole-stage bottleneck launch made no progress for ~0.1 s (its grid was not resident as a whole: are compute units masked off or "
         "held by another process?): the results of this call are not valid; mrcnn_debug_set(\"conv_bneck_stage\", 0) runs one launch per block");
    if (tripped) {
        ++range_overflows;
        if (mode == MRCNN_F32S || mode == MRCNN_F32X3) {
            // this slot's images are still on the device: compute the batch again through the synchronous entry, which lowers the exponents
            // it needs (a newer batch already in flight was computed with the old ones and is checked at its own collect)
            --predict_calls;
            HIP_CHECK(hipStreamSynchronize(stream));                // (a newer batch in flight has finished: its records wait in its own slot)
            HIP_CHECK(hipMemcpyAsync(d_rgb, sl.rgb.p, (size_t)sl.batch * H * W * 3, hipMemcpyDeviceToDevice, stream));
            if (recover_range(sl.batch, H, W, false)) {
                ++range_recoveries;
                predict(nullptr, sl.batch, H, W, MRCNN_DEVICE, det_host, masks_host, true, false, true);
                return;
            }
        }
        fail(MRCNN_ERR_UNSUPPORTED, "an activation left the fp16 range (|v| >= 65504) in the fp16 compute mode: the results of this batch are "
             "not valid; load the model with MRCNN_F32X3 or MRCNN_F32");
    }
}

void Model::enqueue_pipeline(hipStream_t s, int batch, const int* fit)
{
    int* const rflag = mode != MRCNN_F32 ? range_flag.as<int>() : nullptr;
    if (rflag) HIP_CHECK(hipMemsetAsync(rflag, 0, sizeof(int), s));
    // the thread-local hand-overs of the convolution family (range flag, scratch, profiler) point into THIS model: whatever way the
    // function is left — a launch may throw — they are cleared, so that nothing dangles once the handle is destroyed
    struct LaunchContext {
        ~LaunchContext() { conv_set_profiler(nullptr); conv_set_range_flag(nullptr); conv_set_scratch(nullptr); }
    } launch_context;
    conv_set_range_flag(rflag);
    conv_set_scratch(conv_scratch.ks_buf.p ? &conv_scratch : nullptr);
    timer.begin(s);
    conv_set_profiler(conv_profile.active ? &conv_profile : nullptr);
    {
        TraceRange tr("MaskRCNN-Trunk");          // the Core ML graph itself: no signpost in the reference
        size_t first = 0;
        if (fit) {      // .scaleFit: letterbox + mean subtraction in one kernel, instead of trunk_ops[0] (d_rgb → stem_in)
            preprocess_scalefit_forward(s, fit_src.as<uint8_t>(), batch, fit[0], fit[1], H, W, fit[2], fit[3], fit[4], fit[5], 3, mean, stem_in, dtype);
            first = 1;
        }
        for (size_t i = first; i < trunk_ops.size(); ++i) trunk_ops[i](s, batch);
    }
    timer.mark(s, "Trunk");
    // ProposalLayer
    ProposalWorkspace pw = prop_ws; pw.B = batch;
    {
        TraceRange tr("Proposal-Eval");
        proposal_forward(s, pw, rpn_probs, (long)A * 2, rpn_deltas, (long)A * 4, anchors.as<float>(), prop_std, prop_nms_thr, rois,
                         (long)max_prop * 4, 4);
    }
    timer.mark(s, "Proposal-Eval");
    // PyramidROIAlign (classifier)
    PyramidMaps maps;
    for (int l = 0; l < 4; ++l) { maps.data[l] = P[l].p; maps.H[l] = P[l].H; maps.W[l] = P[l].W; maps.sB[l] = P[l].sB(); }
    // the pyramid levels may sit at different split exponents: the sampler brings each to the exponent of its output (exact)
    auto level_muls = [&](int g_out) { for (int l = 0; l < 4; ++l) maps.mul[l] = ldexpf(1.0f, sgroups[(size_t)g_out].exp - sgroups[(size_t)g_P[l]].exp); };
    const long prow = (long)cls_pool * cls_pool * 256;
    {
        TraceRange tr("PyramidROIAlign-Eval");
        level_muls(g_pooled);
        roi_align_forward(s, maps, 256, 1, rois, (long)max_prop * 4, 4, max_prop, batch, cls_pool, roi_img_w, roi_img_h, pooled,
                          (long)max_prop * prow, prow, dtype);
        if (calib_phase) observe_split(s, g_pooled, pooled, (size_t)batch * max_prop * prow);
    }
    timer.mark(s, "PyramidROIAlign-Eval");
    // TimeDistributedClassifier
    {
        TraceRange tr("TimeDistributedClassifierLayer-Eval");
        cls_head.forward(s, pooled, batch * max_prop, cls6, 6);
    }
    timer.mark(s, "TimeDistributedClassifierLayer-Eval");
    // DetectionLayer
    DetectionWorkspace dw = det_ws; dw.B = batch;
    {
        TraceRange tr("Detection-Eval");
        detection_forward(s, dw, rois, (long)max_prop * 4, 4, cls6, (long)max_prop * 6, det_std, det_score_thr, det_nms_thr, nc,
                          detections, (long)max_det * 6, 6);
    }
    timer.mark(s, "Detection-Eval");
    // PyramidROIAlign (mask) on the detections' boxes
    const long mrow = (long)mask_pool * mask_pool * 256;
    // ... which also evaluates the mask layer's removeZeros predicate on the fp32 samples (before an fp16 store rounds them)
    {
        TraceRange tr("PyramidROIAlign-Eval");
        level_muls(g_pooled_mask);
        roi_align_forward(s, maps, 256, 1, detections, (long)max_det * 6, 6, max_det, batch, mask_pool, roi_img_w, roi_img_h, pooled_mask,
                          (long)max_det * mrow, mrow, dtype, msel_ws.flags);
        if (calib_phase) observe_split(s, g_pooled_mask, pooled_mask, (size_t)batch * max_det * mrow);
    }
    timer.mark(s, "PyramidROIAlign-Eval-Mask");
    // TimeDistributedMask
    TraceRange tr_mask("TimeDistributedMask-Eval");
    const int HW = 4 * mask_pool * mask_pool;
    mask_valid_rows_forward(s, nullptr, (long)max_det * mrow, mrow, mrow, max_det, batch, msel_ws, dtype);
    // Rows the reference's mask layer never writes (an invalid row below the kept count,
    // TimeDistributedMaskLayer.swift:58-89) hold whatever Core ML's buffer held; here they are defined: zero.
    HIP_CHECK(hipMemsetAsync(mask_out, 0, (size_t)batch * max_det * HW * sizeof(float), s));
    const bool fused_tail = fuse_mask_tail && g_fuse_mask_tail && mask_head.deconv.Cout % 128 == 0;
    if (fused_tail) {
        // the deconvolution's epilogue takes the dot with the selected class's 1x1 filter: its 256-channel output (642 MB of
        // fp32 per batch of 8) is never written nor read back
        mask_select_classes(s, detections, (long)max_det * 6, 6, max_det, batch, nc, msel_ws, msel_ws.sel_cid);
        mask_head.forward_features(s, pooled_mask, batch * max_det, msel_ws.sel_cid, mask_partial);
        mask_select_from_partials(s, mask_partial, mask_head.deconv.Cout / conv_sel_part_cols(), HW, mask_head.final_b.as<float>(), nc, detections,
                                  (long)max_det * 6, 6, max_det, batch, msel_ws, mask_out, (long)max_det * HW, HW);
    } else {
        mask_head.forward_features(s, pooled_mask, batch * max_det);
        mask_select_forward(s, mask_head.feat, (long)max_det * HW * mask_head.deconv.Cout, HW, mask_head.deconv.Cout,
                            mask_head.final_w.as<float>(), mask_head.final_b.as<float>(), nc, detections, (long)max_det * 6, 6, max_det,
                            batch, msel_ws, mask_out, (long)max_det * HW, HW, dtype);
    }
    timer.mark(s, "TimeDistributedMask-Eval");
}

void Model::read_tensor(const std::string& name, int image, float* dst, int64_t cap, int64_t* count)
{
    MRCNN_REQUIRE(kind == MRCNN_MODEL_MASKRCNN, MRCNN_ERR_INVALID, "taps exist on the MaskRCNN model only");
    MRCNN_REQUIRE(image >= 0 && image < max_batch, MRCNN_ERR_INVALID, "image index %d out of range", image);
    HIP_CHECK(hipStreamSynchronize(stream));
    if (name == "topk_idx" || name == "keep_idx") {
        const bool tk = name == "topk_idx";
        const long n = tk ? K : max_prop;
        if (count) *count = n;
        MRCNN_REQUIRE(dst && cap >= n, MRCNN_ERR_SHAPE, "tensor '%s' needs %ld floats", name.c_str(), n);
        std::vector<int32_t> tmp((size_t)n);
        HIP_CHECK(hipMemcpy(tmp.data(), (tk ? prop_ws.topk_idx : prop_ws.keep_idx) + (size_t)image * n, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; ++i) dst[i] = (float)tmp[(size_t)i];
        return;
    }
    if (name == "boxes_sorted") {
        const long n = (long)K * 4;
        if (count) *count = n;
        MRCNN_REQUIRE(dst && cap >= n, MRCNN_ERR_SHAPE, "tensor '%s' needs %ld floats", name.c_str(), n);
        HIP_CHECK(hipMemcpy(dst, prop_ws.boxes + (size_t)image * n, (size_t)n * 4, hipMemcpyDeviceToHost));
        return;
    }
    if (name == "mask_row_flags") {
        const long n = max_det;
        if (count) *count = n;
        MRCNN_REQUIRE(dst && cap >= n, MRCNN_ERR_SHAPE, "tensor '%s' needs %ld floats", name.c_str(), n);
        std::vector<int32_t> tmp((size_t)n);
        HIP_CHECK(hipMemcpy(tmp.data(), msel_ws.flags + (size_t)image * n, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; ++i) dst[i] = (float)tmp[(size_t)i];
        return;
    }
    if (name == "keep_count") {
        if (count) *count = 1;
        MRCNN_REQUIRE(dst && cap >= 1, MRCNN_ERR_SHAPE, "tensor '%s' needs 1 float", name.c_str());
        int32_t v = 0;
        HIP_CHECK(hipMemcpy(&v, prop_ws.keep_count + image, 4, hipMemcpyDeviceToHost));
        dst[0] = (float)v;
        return;
    }
    auto it = taps.find(name);
    MRCNN_REQUIRE(it != taps.end(), MRCNN_ERR_INVALID, "unknown tensor '%s'", name.c_str());
    const long n = it->second.per_image;
    if (count) *count = n;
    MRCNN_REQUIRE(dst && cap >= n, MRCNN_ERR_SHAPE, "tensor '%s' needs %ld floats, buffer holds %lld", name.c_str(), n, (long long)cap);
    if (it->second.dtype == MRCNN_F16) {
        std::vector<uint16_t> tmp((size_t)n);
        HIP_CHECK(hipMemcpy(tmp.data(), static_cast<const uint16_t*>(it->second.base) + (size_t)image * n, (size_t)n * 2, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; ++i) dst[i] = half_to_float(tmp[(size_t)i]);
    } else {
        HIP_CHECK(hipMemcpy(dst, static_cast<const float*>(it->second.base) + (size_t)image * n, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    // a tensor of a split group is stored as 2^e * value: hand out the value (exact)
    const int e = it->second.group >= 0 ? sgroups[(size_t)it->second.group].exp : 0;
    if (e != 0) for (long i = 0; i < n; ++i) dst[i] = ldexpf(dst[i], -e);
}

}  // namespace mrcnn
