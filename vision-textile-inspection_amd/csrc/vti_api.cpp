// C ABI of libvti.so (declared in include/vti.h): context lifetime, plan execution on the
// caller's HIP stream, error reporting.  Replaces the `ultralytics.YOLO` object the reference
// builds at measurement.py:145 and calls at measurement.py:208-210.
//
// Error convention: every entry point returns a vti_status and records a message; nothing
// throws across the boundary and nothing aborts (measurement.py:207-216 expects predict
// failures to be survivable).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "vti_internal.h"

using namespace vti;

struct vti_ctx {
    Plan plan;
    std::string err;
    int device = -1;
    void* d_wpk = nullptr;     // packed weights (device)
    float* d_bias = nullptr;   // biases (device)
    std::vector<float> alpha;  // per conv: accumulator scale of its packed weights (VTI_H2; 1 otherwise)
    char* ws = nullptr;        // caller-owned workspace
    size_t ws_bytes = 0;
    size_t act_bytes = 0;      // activations part; NMS scratch follows
    const void* last_input = nullptr;
    void* last_proto = nullptr;
    // side streams for the independent branches (proto chain, head levels); created with the weights
    hipStream_t side[kNumLanes] = {};
    hipEvent_t ev_fork[kNumLanes] = {};
    hipEvent_t ev_join[kNumLanes] = {};
    bool multi_stream = false;
};

static std::string g_create_err;
static const int kMaskSlotsPerFrame = VTI_MASK_SLOTS_PER_FRAME;   // mask work-list capacity: max_batch * 512 instances per call

static int32_t fail(vti_ctx* c, int32_t code, const std::string& msg) {
    if (c) c->err = msg; else g_create_err = msg;
    return code;
}
static int32_t hip_fail(vti_ctx* c, hipError_t e, const char* what) {
    return fail(c, VTI_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define VTI_HIP(c, call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail((c), e_, (what)); } while (0)

#ifdef VTI_STAMPS
// diagnostic build: medians of the intervals between the in-kernel s_memtime stamps of one launch, to stderr
static void report_stamps(const std::vector<unsigned long long>& h, size_t nwg, bool pk) {
    const char* names[16] = {"start", "c0:top", "c0:A staged", "c0:B staged", "c0:barrier", "c0:mfma done",
                             "c1:top", "c1:A staged", "c1:B staged", "c1:barrier", "c1:mfma done", "epilogue start", "end", "stage2: mfma done", "stage2: stores issued", ""};
    const char* pkn[16] = {"start", "c0:before barrier", "c0:after barrier", "c0:sub-tile A done", "c0:sub-tile B done",
                           "c1:before barrier", "c1:after barrier", "c1:sub-tile A done", "c1:sub-tile B done", "", "", "epilogue start", "end", "", "", ""};
    if (pk) for (int i = 0; i < 16; ++i) names[i] = pkn[i];
    fprintf(stderr, "[stamps] %zu workgroups; median cycles since previous stamp (100 MHz s_memtime ticks x clock)\n", nwg);
    int prev = 0;
    const int order[13] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 12};     // 13: inside the fused stage, before `end` (14 / 15 hold s_memrealtime)
    for (int oi = 0; oi < 13; ++oi) {
        const int i = order[oi];
        std::vector<long long> d;
        for (size_t w = 0; w < nwg; ++w) if (h[w * 16 + i] && h[w * 16 + prev]) d.push_back((long long)(h[w * 16 + i] - h[w * 16 + prev]));
        if (d.empty()) continue;
        std::sort(d.begin(), d.end());
        fprintf(stderr, "[stamps] %-16s median %8lld  p10 %8lld  p90 %8lld\n", names[i], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        prev = i;
    }
    std::vector<long long> tot;
    for (size_t w = 0; w < nwg; ++w) tot.push_back((long long)(h[w * 16 + 12] - h[w * 16]));
    std::sort(tot.begin(), tot.end());
    fprintf(stderr, "[stamps] whole workgroup   median %8lld\n", tot[tot.size() / 2]);
    std::vector<double> clk;            // in-kernel shader clock: s_memtime cycles per 100 MHz s_memrealtime tick
    for (size_t w = 0; w < nwg; ++w)
        if (h[w * 16 + 15] > h[w * 16 + 14]) clk.push_back((double)(h[w * 16 + 12] - h[w * 16]) / (double)(h[w * 16 + 15] - h[w * 16 + 14]) * 0.1);
    if (!clk.empty()) { std::sort(clk.begin(), clk.end()); fprintf(stderr, "[stamps] in-kernel clock    median %.3f GHz (p10 %.3f, p90 %.3f)\n", clk[clk.size() / 2], clk[clk.size() / 10], clk[clk.size() * 9 / 10]); }
}
#endif

extern "C" {

int32_t vti_create(const vti_desc* desc, vti_ctx** out) {
    if (!desc || !out) return fail(nullptr, VTI_ERR_ARG, "vti_create: null argument");
    *out = nullptr;
    vti_ctx* c = new (std::nothrow) vti_ctx();
    if (!c) return fail(nullptr, VTI_ERR_NOMEM, "vti_create: out of host memory");
    std::string e;
    try { e = c->plan.build(*desc); } catch (const std::exception& ex) { e = ex.what(); }
    if (!e.empty()) { delete c; return fail(nullptr, VTI_ERR_ARG, "vti_create: " + e); }
    c->act_bytes = c->plan.ws_bytes;
    *out = c;
    return VTI_OK;
}

void vti_destroy(vti_ctx* c) {
    if (!c) return;
    if (c->d_wpk) (void)hipFree(c->d_wpk);
    if (c->d_bias) (void)hipFree(c->d_bias);
    for (int l = 1; l < kNumLanes; ++l) {
        if (c->side[l]) (void)hipStreamDestroy(c->side[l]);
        if (c->ev_fork[l]) (void)hipEventDestroy(c->ev_fork[l]);
        if (c->ev_join[l]) (void)hipEventDestroy(c->ev_join[l]);
    }
    delete c;
}

const char* vti_last_error(const vti_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int32_t vti_num_convs(const vti_ctx* c) { return c ? (int32_t)c->plan.convs.size() : 0; }

int32_t vti_conv_at(const vti_ctx* c, int32_t i, vti_conv_info* o) {
    if (!c || !o || i < 0 || i >= (int32_t)c->plan.convs.size()) return VTI_ERR_ARG;
    const ConvRow& r = c->plan.convs[i];
    memset(o, 0, sizeof *o);
    snprintf(o->name, sizeof o->name, "%s", r.name.c_str());
    o->c1 = r.c1; o->c2 = r.c2; o->k = r.k; o->s = r.s; o->kind = r.kind;
    o->h_in = r.h_in; o->w_in = r.w_in; o->h_out = r.h_out; o->w_out = r.w_out;
    o->macs = r.macs();
    for (const Op& op : c->plan.ops) {
        if (op.kind != OP_CONV && op.kind != OP_CONV0) continue;
        if (op.conv == i) {
            o->tile_h = op.cfg.TH; o->tile_w = op.cfg.TW; o->waves_n = op.cfg.WN; o->nrep = op.cfg.NREP;
            o->lds_bytes = (int32_t)op.cfg.lds; o->persistent = op.cfg.pk;   // 1: conv3_pk, 2: conv1_pk, 3: bneck_pk
        } else if (op.tail == i) {      // the C2f's closing 1x1 inside its bottleneck's kernel (bneck_pk tail)
            o->tile_h = op.cfg.TH; o->tile_w = op.cfg.TW; o->waves_n = 1; o->nrep = 2; o->lds_bytes = 0; o->fused = 1; o->persistent = 1;
        } else if (op.pair == i) {      // second 3x3 of a fused Bottleneck: runs inside the first one's kernel (bneck_pk)
            o->tile_h = op.cfg.TH; o->tile_w = op.cfg.TW; o->waves_n = 1; o->nrep = op.cfg.NREP; o->lds_bytes = 0; o->fused = 1;
            o->persistent = 1;
        } else if (op.fold == i) {      // ConvTranspose folded into the following 3x3 (convfold_kernel): never materialised
            o->tile_h = op.cfg.TH; o->tile_w = op.cfg.TW; o->persistent = op.cfg.pk; o->waves_n = 4; o->nrep = op.cfg.NREP; o->lds_bytes = 0;   // `fused` stays 0: that flag means "runs inside the PREVIOUS row's kernel"
        } else if (op.fused_l1 == i) {  // layer 1 inside the stem's kernel (stem_l1_kernel: 16 x 20 output tiles, 2 n-tiles)
            stem_l1_tile(&o->tile_h, &o->tile_w); o->waves_n = 1; o->nrep = 2; o->lds_bytes = 0; o->fused = 1;
        } else if (op.fused == i) {     // runs inside its producer's kernel, on that kernel's geometry
            o->tile_h = op.cfg.TH; o->tile_w = op.cfg.TW; o->waves_n = 1; o->nrep = op.cfg.ntiles2;
            if (op.fused_l1 >= 0) stem_l1_tile(&o->tile_h, &o->tile_w);
            o->lds_bytes = 0; o->fused = 1;
        }
    }
    return VTI_OK;
}

int32_t vti_num_anchors(const vti_ctx* c) { return c ? c->plan.num_anchors : 0; }
int64_t vti_fused_params(const vti_ctx* c) { return c ? c->plan.fused_params : 0; }
int64_t vti_macs_per_frame(const vti_ctx* c) { return c ? c->plan.macs : 0; }
int64_t vti_workspace_bytes(const vti_ctx* c) {
    if (!c) return 0;
    const vti_desc& d = c->plan.desc;
    return (int64_t)(c->plan.ws_bytes + nms_workspace_bytes(d.max_batch, c->plan.num_anchors) +
                     masks_workspace_bytes(d.max_batch * kMaskSlotsPerFrame, d.H, d.W));
}
int32_t vti_num_launches(const vti_ctx* c) {
    if (!c) return 0;
    int32_t n = 0;
    for (const Op& op : c->plan.ops) n += (op.kind != OP_FORK && op.kind != OP_JOIN);
    return n;
}

int32_t vti_load_weights(vti_ctx* c, const void* blob, size_t nbytes, int32_t device) {
    if (!c) return VTI_ERR_ARG;
    std::vector<uint8_t> wpk;
    std::vector<float> bias;
    std::string e;
    try { e = pack_weights(c->plan, blob, nbytes, wpk, bias, c->alpha); } catch (const std::exception& ex) { e = ex.what(); }
    if (!e.empty()) return fail(c, VTI_ERR_WEIGHTS, e);
    VTI_HIP(c, hipSetDevice(device), "hipSetDevice");
    if (c->d_wpk) { (void)hipFree(c->d_wpk); c->d_wpk = nullptr; }
    if (c->d_bias) { (void)hipFree(c->d_bias); c->d_bias = nullptr; }
    VTI_HIP(c, hipMalloc(&c->d_wpk, wpk.size()), "hipMalloc(weights)");
    VTI_HIP(c, hipMalloc((void**)&c->d_bias, bias.size() * sizeof(float)), "hipMalloc(bias)");
    VTI_HIP(c, hipMemcpy(c->d_wpk, wpk.data(), wpk.size(), hipMemcpyHostToDevice), "hipMemcpy(weights)");
    VTI_HIP(c, hipMemcpy(c->d_bias, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(bias)");
    c->device = device;
    // side streams (VTI_SINGLE_STREAM=1 keeps everything on the caller's stream)
    const char* ss = getenv("VTI_SINGLE_STREAM");
    if (!(ss && ss[0] == '1') && !c->side[1]) {
        bool ok = true;
        for (int l = 1; l < kNumLanes && ok; ++l) {
            ok = hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_fork[l], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_join[l], hipEventDisableTiming) == hipSuccess;
        }
        c->multi_stream = ok;
    }
    return VTI_OK;
}

int32_t vti_set_workspace(vti_ctx* c, void* dev_ws, size_t nbytes) {
    if (!c) return VTI_ERR_ARG;
    if (!dev_ws || ((uintptr_t)dev_ws & 255)) return fail(c, VTI_ERR_ARG, "vti_set_workspace: pointer must be 256-B aligned");
    if ((int64_t)nbytes < vti_workspace_bytes(c)) return fail(c, VTI_ERR_NOMEM, "vti_set_workspace: workspace too small");
    c->ws = (char*)dev_ws;
    c->ws_bytes = nbytes;
    return VTI_OK;
}

// A ctx is bound to the device its weights were uploaded to (vti_load_weights); every launching entry point refuses to run
// with another device current instead of launching there with pointers of the wrong GPU.
static int32_t check_device(vti_ctx* c, const char* fn) {
    if (!c || c->device < 0) return VTI_OK;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != c->device)
        return fail(c, VTI_ERR_STATE, std::string(fn) + ": ctx is bound to device " + std::to_string(c->device) +
                    " but the current device is " + std::to_string(cur) + " (hipSetDevice / torch.cuda.set_device it first)");
    return VTI_OK;
}

static int32_t check_ready(vti_ctx* c, int32_t B, const char* fn) {
    if (!c) return VTI_ERR_ARG;
    if (int32_t rc = check_device(c, fn)) return rc;
    if (B < 0 || B > c->plan.desc.max_batch) return fail(c, VTI_ERR_ARG, std::string(fn) + ": B out of range (0..max_batch)");
    if (!c->d_wpk) return fail(c, VTI_ERR_STATE, std::string(fn) + ": weights not loaded");
    if (!c->ws) return fail(c, VTI_ERR_STATE, std::string(fn) + ": workspace not set");
    return VTI_OK;
}

// Ultralytics LetterBox geometry (auto=False here: the model size HxW is fixed at vti_create).
static void letterbox_geom(int H0, int W0, int H, int W, int& new_h, int& new_w, int& top, int& left) {
    const double r = std::min((double)H / H0, (double)W / W0);
    new_w = (int)std::nearbyint(W0 * r);
    new_h = (int)std::nearbyint(H0 * r);
    const double dw = (W - new_w) / 2.0, dh = (H - new_h) / 2.0;
    top = (int)std::nearbyint(dh - 0.1);
    left = (int)std::nearbyint(dw - 0.1);
}

int32_t vti_letterbox(vti_ctx* c, const uint8_t* frames, int32_t B, int32_t H0, int32_t W0, uint8_t* out, void* stream) {
    if (!c || !frames || !out || H0 < 1 || W0 < 1 || B < 0) return fail(c, VTI_ERR_ARG, "vti_letterbox: bad argument");
    int nh, nw, top, left;
    letterbox_geom(H0, W0, c->plan.desc.H, c->plan.desc.W, nh, nw, top, left);
    if (int32_t drc = check_device(c, "vti_letterbox")) return drc;
    VTI_HIP(c, launch_letterbox(frames, B, H0, W0, out, c->plan.desc.H, c->plan.desc.W, nh, nw, top, left, (hipStream_t)stream),
            "letterbox kernel");
    return VTI_OK;
}

static void* buf_ptr(vti_ctx* c, int buf, const void* input, void* proto) {
    if (buf == 0) return (void*)input;
    if (buf == c->plan.proto_buf_c) return proto;
    return c->ws + c->plan.bufs[buf].off;
}

// Delay units (x 1024 cycles) of the staggered persistent 3x3 launches.  h2 only by default: -2.6 ... -4.2 % per forward on five of
// seven boxes (A/B on one box at a time), +0.8 % on the two fastest; fp16: +0.2 ... +1 % on a fast box, fp32: nothing.  VTI_PK_STAGGER=n overrides.
static int stagger_units(int dtype) {
    static const int v = getenv("VTI_PK_STAGGER") ? std::max(0, std::min(255, atoi(getenv("VTI_PK_STAGGER")))) : -1;
    return v >= 0 ? v : (dtype == VTI_H2 ? 10 : 0);
}

static int pk_linear_map() {       // A/B aid: VTI_PK_LINEAR_MAP=1 restores the linear pixel -> column-tile map of the persistent 3x3 kernels
    static const int v = getenv("VTI_PK_LINEAR_MAP") && getenv("VTI_PK_LINEAR_MAP")[0] == '1';
    return v;
}

// One conv launch's parameter block from its table row + geometry + tensor views.
static void fill_conv_params(int conv_elem_size, ConvParams& p, const ConvRow& r, const ConvCfg& g, int B, const void* in, int in_ld,
                             int in_coff, void* out, int out_ld, int out_coff, const void* res, int res_ld,
                             int res_coff, const void* wpk, const float* bias, bool out_f32, int swap_rb) {
    const bool deconv = r.kind == 2;
    memset(&p, 0, sizeof p);
    p.in = in; p.out = out; p.wpk = wpk; p.bias = bias;
    p.B = B; p.Hin = r.h_in; p.Win = r.w_in;
    p.Hout = deconv ? r.h_in : r.h_out; p.Wout = deconv ? r.w_in : r.w_out;
    p.Cin = r.c1; p.in_ld = in_ld; p.in_coff = in_coff;
    p.Cout = g.gemm_n; p.out_ld = out_ld; p.out_coff = out_coff;
    if (res) { p.res = res; p.res_ld = res_ld; p.res_coff = res_coff; p.has_res = 1; }
    p.TH = g.TH; p.TW = g.TW;
    p.tiles_y = (p.Hout + g.TH - 1) / g.TH; p.tiles_x = (p.Wout + g.TW - 1) / g.TW;
    p.WN = g.WN;
    p.nt = g.threads;
    p.pk_lin = pk_linear_map();
    p.act = r.kind == 0; p.out_f32 = out_f32 ? 1 : 0;
    p.deconv_c = deconv ? r.c2 : 0;
    p.swap_rb = swap_rb ? 1 : 0;
    p.nchunks = g.nchunks; p.ntiles_n = g.ntiles_n;
    p.scalar_store = (g.gemm_n % 4 || out_ld % 4 || out_coff % 4) ? 1 : 0;
    const bool conv0 = r.c1 == 3;
    const int ks = deconv ? 1 : r.k, st = deconv ? 1 : r.s;
    const unsigned PW = conv0 ? (unsigned)g.TW : (unsigned)((g.TW - 1) * st + ks);
    p.pw_magic = (unsigned)((0x100000000ull + PW - 1) / PW);
    const unsigned RWD = ((unsigned)(2 * g.TW + 1) * 3 + 6) >> 2;      // stem: dwords per u8 patch row
    p.rw_magic = (unsigned)((0x100000000ull + RWD - 1) / RWD);
    p.tw_magic = (unsigned)((0x100000000ull + (unsigned)g.TW - 1) / (unsigned)g.TW);
    p.wpk_bytes = (unsigned)packed_conv_bytes(r, conv0, g);
    if (g.pk == 2) {  // persistent 1x1 kernel: tiles are runs of TH * 80 pixels of the flattened [B*H*W] index space
        const size_t es = conv_elem_size;
        const size_t npx = (size_t)B * p.Hout * p.Wout;
        const size_t ib = npx * in_ld * es;
        const size_t ob = (deconv ? 4 * npx : npx) * out_ld * (out_f32 ? 4 : es);
        p.pk = (ib < 0x80000000ull && ob < 0x80000000ull && !res) ? 2 : 0;
        p.in_bytes = (unsigned)ib; p.out_bytes = (unsigned)ob; p.res_bytes = 0;
        p.pk_depth = g.pk_depth; p.pk_wstat = g.pk_wstat; p.pk_cps = g.pk_cps;
        p.pk_tiles = (int)((npx + (size_t)g.TH * 80 - 1) / ((size_t)g.TH * 80));
        const int gy = g.ntiles_n / (g.WN * g.NREP);
        int G = std::min(p.pk_tiles, std::max(1, 256 * g.pk_wgpc / gy));
        if (const char* cap = getenv("VTI_PK_MAX_WGS")) G = std::max(1, std::min(G, atoi(cap)));
        p.pk_xcd = G >= 8 ? 1 : 0;
        if (p.pk_xcd) G &= ~7;
        p.pk_wgs = G;
    } else if (g.pk) {   // persistent kernel: workgroups along x walk the B * tiles_y * tiles_x tiles
        const size_t es = conv_elem_size;
        const size_t ib = (size_t)B * r.h_in * r.w_in * in_ld * es, ob = (size_t)B * p.Hout * p.Wout * out_ld * (out_f32 ? 4 : es);
        const size_t rb = res ? (size_t)B * p.Hout * p.Wout * res_ld * es : 0;
        p.pk = (ib < 0x80000000ull && ob < 0x80000000ull && rb < 0x80000000ull) ? g.pk : 0;   // 2^31 marks out-of-range lanes
        p.in_bytes = (unsigned)ib; p.out_bytes = (unsigned)ob; p.res_bytes = (unsigned)rb;
        p.pk_tiles = B * p.tiles_y * p.tiles_x;
        p.pk_depth = g.pk_depth; p.pk_wstat = g.pk_wstat;
        const int gy = g.ntiles_n / (g.WN * g.NREP);
        int G = std::min(p.pk_tiles, std::max(1, 256 * g.pk_wgpc / gy));
        if (const char* cap = getenv("VTI_PK_MAX_WGS")) G = std::max(1, std::min(G, atoi(cap)));   // tests: force many tiles per workgroup
        p.pk_xcd = G >= 8 ? 1 : 0;
        if (p.pk_xcd) G &= ~7;
        p.pk_wgs = G;
        // launches that fill the chip start the upper half of their workgroups late (conv_pk.hip: pk_stagger_wait): eligibility here,
        // the caller scales it by stagger_units(dtype)
        p.pk_stagger = ((g.pk == 1 || g.pk == 4) && p.pk_tiles >= std::max(1, 256 * g.pk_wgpc / gy) && G >= 16) ? 1 : 0;
    }
}

static int32_t forward_impl(vti_ctx* c, const uint8_t* input, int32_t B, int32_t swap_rb, float* pred, void* proto, float* best, void* stream);

int32_t vti_forward(vti_ctx* c, const uint8_t* input, int32_t B, int32_t swap_rb, float* pred, void* proto, void* stream) {
    return forward_impl(c, input, B, swap_rb, pred, proto, nullptr, stream);
}

int32_t vti_forward_scored(vti_ctx* c, const uint8_t* input, int32_t B, int32_t swap_rb, float* pred, void* proto, float* anchor_best,
                           void* stream) {
    if (!anchor_best) return fail(c, VTI_ERR_ARG, "vti_forward_scored: null pointer");
    return forward_impl(c, input, B, swap_rb, pred, proto, anchor_best, stream);
}

static int32_t forward_impl(vti_ctx* c, const uint8_t* input, int32_t B, int32_t swap_rb, float* pred, void* proto, float* best, void* stream) {
    int32_t rc = check_ready(c, B, "vti_forward");
    if (rc) return rc;
    if (!input || !pred || !proto) return fail(c, VTI_ERR_ARG, "vti_forward: null pointer");
    if (B == 0) return VTI_OK;
    const Plan& P = c->plan;
    hipStream_t main_st = (hipStream_t)stream;
    const int dt = P.desc.dtype;
    static const bool list_ops = getenv("VTI_LIST_OPS") != nullptr;      // developer aid: launch order, to label a kernel trace
    if (list_ops) {
        int i = 0;
        for (const Op& op : P.ops) {
            if (op.kind == OP_FORK || op.kind == OP_JOIN) continue;
            const bool cv = op.kind == OP_CONV || op.kind == OP_CONV0;
            fprintf(stderr, "[op %2d] lane %d %s%s%s\n", i++, op.lane,
                    cv ? P.convs[op.conv].name.c_str() : op.kind == OP_POOL ? "sppf_pool" : op.kind == OP_UP2 ? "upsample2x" : "decode",
                    cv && op.fused_l1 >= 0 ? (" + " + P.convs[op.fused_l1].name).c_str() : cv && op.fold >= 0 ? (" (folded: " + P.convs[op.fold].name + ")").c_str() : cv && op.pair >= 0 ? (" + " + P.convs[op.pair].name + (op.tail >= 0 ? " + " + P.convs[op.tail].name : "")).c_str() : "",
                    cv && op.fused >= 0 ? (" + " + P.convs[op.fused].name).c_str() : "");
        }
    }
    for (const Op& op : P.ops) {
        hipStream_t st = (c->multi_stream && op.lane > 0) ? c->side[op.lane] : main_st;
        switch (op.kind) {
        case OP_FORK:
            if (c->multi_stream) {
                VTI_HIP(c, hipEventRecord(c->ev_fork[op.lane], main_st), "fork record");
                VTI_HIP(c, hipStreamWaitEvent(c->side[op.lane], c->ev_fork[op.lane], 0), "fork wait");
            }
            break;
        case OP_JOIN:
            if (c->multi_stream) {
                VTI_HIP(c, hipEventRecord(c->ev_join[op.lane], c->side[op.lane]), "join record");
                VTI_HIP(c, hipStreamWaitEvent(main_st, c->ev_join[op.lane], 0), "join wait");
            }
            break;
        case OP_CONV0:
        case OP_CONV: {
            const ConvRow& r = P.convs[op.conv];
            const ConvCfg& g = op.cfg;
            const Buf& ib = P.bufs[op.in.buf];
            const Buf& ob = P.bufs[op.out.buf];
            if (op.fused_l1 >= 0) {     // stem + layer 1 in one kernel
                const ConvRow& r1 = P.convs[op.fused_l1];
                const Buf& o1 = P.bufs[op.out2.buf];
                ConvParams q;
                memset(&q, 0, sizeof q);
                q.in = input; q.B = B; q.Hin = r.h_in; q.Win = r.w_in; q.Hout = r1.h_out; q.Wout = r1.w_out;
                q.Cin = 16; q.Cout = r1.c2; q.ntiles_n = 2; q.act = 1; q.swap_rb = swap_rb ? 1 : 0;
                q.out = buf_ptr(c, op.out2.buf, input, proto); q.out_ld = o1.C; q.out_coff = op.out2.coff;
                q.wpk = (const char*)c->d_wpk + g.wpk_off2; q.bias = c->d_bias + g.bias_off2;     // layer 1
                q.w0 = (const char*)c->d_wpk + g.wpk_off; q.bias0 = c->d_bias + g.bias_off;       // stem
                q.alpha = c->alpha[op.fused_l1]; q.alpha0 = c->alpha[op.conv]; q.alpha2 = op.fused >= 0 ? c->alpha[op.fused] : 1.f;
                if (op.fused >= 0) {    // + the 1x1 conv after layer 1: layer 1 itself is not stored (out2 = that conv's view)
                    q.out = nullptr;
                    q.w2 = (const char*)c->d_wpk + g.wpk_off3; q.bias2 = c->d_bias + g.bias_off3;
                    q.out2 = buf_ptr(c, op.out2.buf, input, proto);
                    q.Cout2 = g.gemm_n2; q.ntiles2 = g.ntiles2; q.out2_ld = o1.C; q.out2_coff = op.out2.coff;
                    q.act2 = 1; q.out2_bstride = r1.h_out * r1.w_out;
                    q.scalar_store2 = (g.gemm_n2 % 4 || o1.C % 4 || op.out2.coff % 4) ? 1 : 0;
                }
                stem_l1_tile(&q.TH, &q.TW);
                q.tiles_y = (q.Hout + q.TH - 1) / q.TH; q.tiles_x = (q.Wout + q.TW - 1) / q.TW; q.WN = 1;
                q.scalar_store = (q.out_ld % 4 || q.out_coff % 4) ? 1 : 0;
#ifdef VTI_STAMPS
                if (const char* so = getenv("VTI_STAMP_OP")) {
                    if (r.name == so) {
                        const size_t nwg = (size_t)stem_l1_grid(dt, B * q.tiles_y * q.tiles_x);
                        unsigned long long* d_st = nullptr;
                        if (hipMalloc((void**)&d_st, nwg * 16 * 8) == hipSuccess) {
                            (void)hipMemset(d_st, 0, nwg * 16 * 8);
                            q.stamps = d_st;
                            (void)launch_stem_l1(dt, q, st);
                            (void)hipStreamSynchronize(st);
                            std::vector<unsigned long long> h(nwg * 16);
                            (void)hipMemcpy(h.data(), d_st, nwg * 16 * 8, hipMemcpyDeviceToHost);
                            (void)hipFree(d_st);
                            fprintf(stderr, "[stamps] op %s + layer 1 (1: patch staged, 2: stem done, 3: layer-1 MFMAs done, 12: end)\n", so);
                            report_stamps(h, nwg, false);
                            q.stamps = nullptr;
                        }
                    }
                }
#endif
                VTI_HIP(c, launch_stem_l1(dt, q, st), "stem + layer 1");
                break;
            }
            if (op.fold >= 0) {         // ConvTranspose2d(2,2) + 3x3 + fused 1x1 as four 2x2 convs on the low-resolution map
                const ConvRow& ru = P.convs[op.fold];
                const Buf& o2 = P.bufs[op.out2.buf];
                ConvParams q;
                memset(&q, 0, sizeof q);
                q.in = buf_ptr(c, op.in.buf, input, proto); q.B = B; q.Hin = ru.h_in; q.Win = ru.w_in; q.Hout = ru.h_in; q.Wout = ru.w_in;
                q.Cin = ru.c1; q.in_ld = ib.C; q.in_coff = op.in.coff; q.Cout = g.gemm_n; q.act = 1; q.fold = 1; q.pk_lin = pk_linear_map();
                q.TH = g.TH; q.TW = g.TW; q.tiles_y = (q.Hout + g.TH - 1) / g.TH; q.tiles_x = (q.Wout + g.TW - 1) / g.TW; q.WN = 4; q.nt = g.threads;
                q.nchunks = g.nchunks; q.ntiles_n = g.ntiles_n;
                q.pw_magic = (unsigned)((0x100000000ull + (unsigned)(g.TW + 2) - 1) / (unsigned)(g.TW + 2));
                q.tw_magic = (unsigned)((0x100000000ull + (unsigned)g.TW - 1) / (unsigned)g.TW);
                q.wpk = (const char*)c->d_wpk + g.wpk_off; q.bias = c->d_bias + g.bias_off; q.wpk_bytes = (unsigned)packed_fold_bytes(g);
                q.w2 = (const char*)c->d_wpk + g.wpk_off2; q.bias2 = c->d_bias + g.bias_off2;
                q.alpha = c->alpha[op.conv]; q.alpha2 = c->alpha[op.fused]; q.alpha0 = 1.f;
                q.out2 = buf_ptr(c, op.out2.buf, input, proto);
                q.Cout2 = g.gemm_n2; q.ntiles2 = g.ntiles2; q.out2_ld = o2.C; q.out2_coff = op.out2.coff;
                q.act2 = P.convs[op.fused].kind == 0; q.out2_f32 = op.out2_f32 ? 1 : 0;
                q.scalar_store2 = (g.gemm_n2 % 4 || o2.C % 4 || op.out2.coff % 4) ? 1 : 0;
                q.out2_bstride = 4 * q.Hout * q.Wout;
                q.nat2 = op.nat2;           // f32 proto (h2 engine): stage-2 weights are packed with natural rows
                if (g.pk) {             // persistent schedule: composed weights resident in LDS, tiles walked per XCD
                    q.pk = 1; q.pk_depth = g.pk_depth; q.in_bytes = (unsigned)((size_t)B * q.Hin * q.Win * q.in_ld * P.esize);
                    q.pk_tiles = B * q.tiles_y * q.tiles_x;
                    int G = std::min(q.pk_tiles, 256);
                    if (const char* cap = getenv("VTI_PK_MAX_WGS")) G = std::max(1, std::min(G, atoi(cap)));
                    q.pk_xcd = G >= 8 ? 1 : 0;
                    if (q.pk_xcd) G &= ~7;
                    q.pk_wgs = G;
                    VTI_HIP(c, launch_conv_pk_fold(dt, q, g.lds, st), r.name.c_str());
                } else {
                    VTI_HIP(c, launch_convfold(dt, q, g.lds, st), r.name.c_str());
                }
                break;
            }
            ConvParams p;
            fill_conv_params(P.esize, p, r, g, B, buf_ptr(c, op.in.buf, input, proto), ib.C, op.in.coff,
                             buf_ptr(c, op.out.buf, input, proto), ob.C, op.out.coff,
                             op.has_res ? buf_ptr(c, op.res.buf, input, proto) : nullptr,
                             op.has_res ? P.bufs[op.res.buf].C : 0, op.res.coff,
                             (const char*)c->d_wpk + g.wpk_off, c->d_bias + g.bias_off, op.out_f32, swap_rb);
            p.alpha = c->alpha[op.conv]; p.alpha0 = 1.f;
            p.pk_stagger *= stagger_units(dt);
            p.alpha2 = op.fused >= 0 ? c->alpha[op.fused] : op.pair >= 0 ? c->alpha[op.pair] : 1.f;
            if (op.fused >= 0) {
                const Buf& o2 = P.bufs[op.out2.buf];
                p.w2 = (const char*)c->d_wpk + g.wpk_off2;
                p.bias2 = c->d_bias + g.bias_off2;
                p.out2 = buf_ptr(c, op.out2.buf, input, proto);
                p.Cout2 = g.gemm_n2; p.ntiles2 = g.ntiles2; p.out2_ld = o2.C; p.out2_coff = op.out2.coff;
                p.act2 = P.convs[op.fused].kind == 0; p.out2_f32 = op.out2_f32 ? 1 : 0;
                p.scalar_store2 = (g.gemm_n2 % 4 || o2.C % 4 || op.out2.coff % 4) ? 1 : 0;
                p.nat2 = op.nat2;
                p.out2_bstride = r.h_out * r.w_out;
                if (op.pred_mode) {     // class / coefficient towers write their rows of the anchor-major pred [B, A, no] directly
                    const int no = 4 + P.desc.nc + P.desc.nm;
                    p.out2 = pred + (size_t)op.pred_a0 * no;
                    p.out2_ld = no; p.out2_coff = op.pred_cbase; p.out2_f32 = 1; p.out2_bstride = P.num_anchors;
                    p.act2 = op.pred_mode == 2 ? 2 : op.pred_mode == 3 ? 3 : 0;
                    p.dfl_stride = (float)op.dfl_stride;
                    p.scalar_store2 = (g.gemm_n2 % 4 || no % 4 || op.pred_cbase % 4) ? 1 : 0;
                    p.best = (op.pred_mode == 2 && best) ? best + (size_t)op.pred_a0 * 2 : nullptr;     // class towers: (max, class) per anchor
                }
            }
            if (op.pair >= 0) {         // fused Bottleneck: second conv's weights; p.out / p.res already are the second conv's views
                p.w2 = (const char*)c->d_wpk + g.wpk_off2;
                p.bias2 = c->d_bias + g.bias_off2;
                if (p.pk != 3) return fail(c, VTI_ERR_UNSUPPORTED, "fused bottleneck needs the persistent kernel (tensor too large?)");
                if (op.tail >= 0) {     // + the C2f's closing 1x1: the patch carries [y0 | y1] (in_coff = y0's offset), y2 is not stored
                    const Buf& o2 = P.bufs[op.out2.buf];
                    if (dt == VTI_F16) p.in_coff -= r.c1;          // fp16: one 64-byte slot = [y0 | y1]; h2: the patch stays y1, y0 is read directly
                    p.alpha0 = c->alpha[op.tail];
                    p.w0 = (const char*)c->d_wpk + g.wpk_off3; p.bias0 = c->d_bias + g.bias_off3;
                    p.out2 = buf_ptr(c, op.out2.buf, input, proto); p.out2_ld = o2.C; p.out2_coff = op.out2.coff; p.Cout2 = P.convs[op.tail].c2;
                    const size_t o2b = (size_t)B * p.Hout * p.Wout * o2.C * P.esize;
                    if (o2b >= 0x80000000ull) return fail(c, VTI_ERR_UNSUPPORTED, "fused bottleneck tail: output tensor too large");
                    p.out2_bytes = (unsigned)o2b;
                }
            }
            if (op.up_C > 0) {
                const Buf& ub = P.bufs[op.up_src.buf];
                p.in2 = buf_ptr(c, op.up_src.buf, input, proto); p.in2_ld = ub.C; p.in2_coff = op.up_src.coff; p.up_C = op.up_C;
                p.in2_bytes = (unsigned)((size_t)B * ub.H * ub.W * ub.C * P.esize);
                if (p.pk != 2) return fail(c, VTI_ERR_UNSUPPORTED, "folded upsample needs the persistent 1x1 kernel (tensor too large?)");
            }
            const bool deconv = r.kind == 2;
            const int ks = deconv ? 1 : r.k, s = deconv ? 1 : r.s;
#ifdef VTI_STAMPS
            if (const char* so = getenv("VTI_STAMP_OP")) {     // diagnostic build: stamp this op of the forward (fused ops included)
                if (r.name == so) {
                    const int NTBs = g.WN * g.NREP;
                    const size_t nwg = (p.pk ? (size_t)p.pk_wgs : (size_t)B * p.tiles_y * p.tiles_x) * ((g.ntiles_n + NTBs - 1) / NTBs);
                    unsigned long long* d_st = nullptr;
                    if (hipMalloc((void**)&d_st, nwg * 16 * 8) == hipSuccess) {
                        (void)hipMemset(d_st, 0, nwg * 16 * 8);
                        p.stamps = d_st;
                        (void)launch_conv(dt, ks, s, g.NREP, op.kind == OP_CONV0 ? 1 : 0, p, g.lds, st);
                        (void)hipStreamSynchronize(st);
                        std::vector<unsigned long long> h(nwg * 16);
                        (void)hipMemcpy(h.data(), d_st, nwg * 16 * 8, hipMemcpyDeviceToHost);
                        (void)hipFree(d_st);
                        fprintf(stderr, "[stamps] op %s\n", so);
                        report_stamps(h, nwg, p.pk != 0);
                        p.stamps = nullptr;
                        break;
                    }
                }
            }
#endif
            VTI_HIP(c, launch_conv(dt, ks, s, g.NREP, op.kind == OP_CONV0 ? 1 : 0, p, g.lds, st), r.name.c_str());
            break;
        }
        case OP_POOL: {
            const Buf& ib = P.bufs[op.in.buf];
            PoolParams p;
            p.in = buf_ptr(c, op.in.buf, input, proto); p.out = p.in ? (void*)p.in : nullptr;
            p.B = B; p.H = ib.H; p.W = ib.W; p.C = op.in.C; p.ld = ib.C; p.in_coff = op.in.coff; p.out_coff = op.out.coff;
            VTI_HIP(c, launch_sppf_pool(dt, p, st), "sppf pool");
            break;
        }
        case OP_UP2: {
            const Buf& ib = P.bufs[op.in.buf];
            const Buf& ob = P.bufs[op.out.buf];
            Up2Params p;
            p.in = buf_ptr(c, op.in.buf, input, proto); p.out = buf_ptr(c, op.out.buf, input, proto);
            p.B = B; p.H = ib.H; p.W = ib.W; p.C = op.in.C; p.in_ld = ib.C; p.in_coff = op.in.coff;
            p.out_ld = ob.C; p.out_coff = op.out.coff;
            VTI_HIP(c, launch_upsample2x(dt, p, st), "upsample2x");
            break;
        }
        case OP_DECODE: {
            DecodeParams p;
            memset(&p, 0, sizeof p);
            int a0 = 0;
            for (int l = 0; l < 3; ++l) {
                const Level& lv = P.levels[l];
                p.box[l] = (const float*)(c->ws + P.bufs[lv.box_buf].off);
                p.cls[l] = (const float*)(c->ws + P.bufs[lv.cls_buf].off);
                p.mc[l] = (const float*)(c->ws + P.bufs[lv.mc_buf].off);
                p.H[l] = lv.H; p.W[l] = lv.W; p.stride[l] = lv.stride; p.a0[l] = a0;
                a0 += lv.H * lv.W;
            }
            p.B = B; p.A = P.num_anchors; p.nc = P.desc.nc; p.nm = P.desc.nm; p.reg_max = P.desc.reg_max;
            p.pred = pred;
            if (P.pred_scatter) VTI_HIP(c, launch_box_decode(p, st), "box decode");
            else VTI_HIP(c, launch_decode(p, st), "decode");
            break;
        }
        }
    }
    // a plan whose class towers do not write pred themselves (VTI_NO_SCATTER ...) derives the pairs from the finished rows
    if (best && !P.pred_scatter)
        VTI_HIP(c, launch_anchor_best(pred, B, P.num_anchors, P.desc.nc, P.desc.nm, best, main_st), "anchor pairs");
    c->last_input = input;
    c->last_proto = proto;
    return VTI_OK;
}

int32_t vti_nms_scored(vti_ctx* c, const float* pred, const float* anchor_best, int32_t B, float conf, double iou, int32_t max_det,
                       int32_t agnostic, float* dets, int32_t* counts, void* stream) {
    if (!c) return VTI_ERR_ARG;
    if (B < 0 || B > c->plan.desc.max_batch) return fail(c, VTI_ERR_ARG, "vti_nms_scored: B out of range");
    if (!c->ws) return fail(c, VTI_ERR_STATE, "vti_nms_scored: workspace not set");
    if (!pred || !anchor_best || !dets || !counts || max_det < 1) return fail(c, VTI_ERR_ARG, "vti_nms_scored: bad argument");
    if (int32_t drc = check_device(c, "vti_nms_scored")) return drc;
    VTI_HIP(c, launch_nms(pred, anchor_best, B, c->plan.num_anchors, c->plan.desc.nc, c->plan.desc.nm, conf, iou, max_det, agnostic,
                          dets, counts, c->ws + c->act_bytes, (hipStream_t)stream), "nms kernel");
    return VTI_OK;
}

int32_t vti_nms(vti_ctx* c, const float* pred, int32_t B, float conf, double iou, int32_t max_det, int32_t agnostic,
                float* dets, int32_t* counts, void* stream) {
    if (!c) return VTI_ERR_ARG;
    if (B < 0 || B > c->plan.desc.max_batch) return fail(c, VTI_ERR_ARG, "vti_nms: B out of range");
    if (!c->ws) return fail(c, VTI_ERR_STATE, "vti_nms: workspace not set");
    if (!pred || !dets || !counts || max_det < 1) return fail(c, VTI_ERR_ARG, "vti_nms: bad argument");
    if (int32_t drc = check_device(c, "vti_nms")) return drc;
    VTI_HIP(c, launch_nms(pred, nullptr, B, c->plan.num_anchors, c->plan.desc.nc, c->plan.desc.nm, conf, iou, max_det, agnostic,
                          dets, counts, c->ws + c->act_bytes, (hipStream_t)stream), "nms kernel");
    return VTI_OK;
}

int32_t vti_masks(vti_ctx* c, const float* dets, const int32_t* counts, const void* proto, int32_t B, int32_t max_det,
                  int32_t mode, int32_t packing, uint8_t* masks, int32_t capacity, int32_t* offsets, void* stream) {
    if (!c || !dets || !counts || !proto || !offsets || B < 0 || max_det < 1 || capacity < 0 || (capacity && !masks))
        return fail(c, VTI_ERR_ARG, "vti_masks: bad argument");
    if ((mode != VTI_MASK_LOGIT && mode != VTI_MASK_SIGMOID) || (packing != VTI_PACK_U8 && packing != VTI_PACK_BITS))
        return fail(c, VTI_ERR_ARG, "vti_masks: bad mode/packing");
    const vti_desc& d = c->plan.desc;
    if (!c->ws) return fail(c, VTI_ERR_STATE, "vti_masks: workspace not set");
    if (B > d.max_batch) return fail(c, VTI_ERR_ARG, "vti_masks: B out of range");
    if (capacity > d.max_batch * kMaskSlotsPerFrame)
        return fail(c, VTI_ERR_UNSUPPORTED, "vti_masks: capacity above max_batch*512 instances per call");
    void* mws = c->ws + c->act_bytes + nms_workspace_bytes(d.max_batch, c->plan.num_anchors);
    if (int32_t drc = check_device(c, "vti_masks")) return drc;
    VTI_HIP(c, launch_masks(d.dtype, dets, counts, proto, B, max_det, d.nm, d.H / 4, d.W / 4, d.H, d.W, mode, packing, masks,
                            capacity, offsets, mws, (hipStream_t)stream), "mask kernel");
    return VTI_OK;
}

int32_t vti_scale_boxes(vti_ctx* c, const float* dets, const int32_t* counts, int32_t B, int32_t max_det, int32_t H0,
                        int32_t W0, float* xyxy, void* stream) {
    if (!c || !dets || !counts || !xyxy || B < 0 || max_det < 1 || H0 < 1 || W0 < 1)
        return fail(c, VTI_ERR_ARG, "vti_scale_boxes: bad argument");
    const vti_desc& d = c->plan.desc;
    if (int32_t drc = check_device(c, "vti_scale_boxes")) return drc;
    VTI_HIP(c, launch_scale_boxes(dets, counts, B, max_det, d.nm, d.H, d.W, H0, W0, xyxy, (hipStream_t)stream), "scale_boxes kernel");
    return VTI_OK;
}

int32_t vti_predict(vti_ctx* c, const uint8_t* frames, int32_t B, int32_t H0, int32_t W0, int32_t swap_rb, float conf,
                    double iou, int32_t max_det, int32_t agnostic, int32_t mask_mode, int32_t packing,
                    uint8_t* input_scratch, float* pred, void* proto, float* dets, int32_t* counts, uint8_t* masks,
                    int32_t capacity, int32_t* offsets, float* xyxy, void* stream) {
    int32_t rc = check_ready(c, B, "vti_predict");
    if (rc) return rc;
    const vti_desc& d = c->plan.desc;
    const uint8_t* input = frames;
    if (H0 != d.H || W0 != d.W) {
        if (!input_scratch) return fail(c, VTI_ERR_ARG, "vti_predict: frames need letterboxing but dev_input_scratch is NULL");
        rc = vti_letterbox(c, frames, B, H0, W0, input_scratch, stream);
        if (rc) return rc;
        input = input_scratch;
    }
    // the (max score, class) pairs travel from the class towers to the NMS filter through the library's own workspace
    float* best = c->ws ? nms_workspace_best(c->ws + c->act_bytes, d.max_batch, c->plan.num_anchors) : nullptr;
    if (!best) return fail(c, VTI_ERR_STATE, "vti_predict: workspace not set");
    if ((rc = vti_forward_scored(c, input, B, swap_rb, pred, proto, best, stream))) return rc;
    if ((rc = vti_nms_scored(c, pred, best, B, conf, iou, max_det, agnostic, dets, counts, stream))) return rc;
    if ((rc = vti_masks(c, dets, counts, proto, B, max_det, mask_mode, packing, masks, capacity, offsets, stream))) return rc;
    if (xyxy && (rc = vti_scale_boxes(c, dets, counts, B, max_det, H0, W0, xyxy, stream))) return rc;
    return VTI_OK;
}

int32_t vti_mask_to_frame(vti_ctx* c, const uint8_t* masks, int32_t n, int32_t H, int32_t W, int32_t H0, int32_t W0,
                          uint8_t* bitmaps, int32_t* nonzero, void* stream) {
    if (!c || n < 0 || H < 1 || W < 1 || H0 < 1 || W0 < 1 || (n && (!masks || !bitmaps || !nonzero)))
        return fail(c, VTI_ERR_ARG, "vti_mask_to_frame: bad argument");
    if (int32_t drc = check_device(c, "vti_mask_to_frame")) return drc;
    VTI_HIP(c, launch_mask_to_frame(masks, n, H, W, H0, W0, bitmaps, nonzero, (hipStream_t)stream), "mask_to_frame kernel");
    return VTI_OK;
}

int32_t vti_union_envelope(vti_ctx* c, const uint8_t* bitmaps, const int32_t* select, int32_t nsel, int32_t H0, int32_t W0,
                           uint8_t* uni, int32_t* envelope, void* stream) {
    if (!c || nsel < 0 || H0 < 1 || W0 < 1 || !uni || !envelope || (nsel && (!bitmaps || !select)))
        return fail(c, VTI_ERR_ARG, "vti_union_envelope: bad argument");
    if (int32_t drc = check_device(c, "vti_union_envelope")) return drc;
    VTI_HIP(c, launch_union_envelope(bitmaps, select, nsel, H0, W0, uni, envelope, (hipStream_t)stream), "union_envelope kernel");
    return VTI_OK;
}

int32_t vti_mask_stats(vti_ctx* c, const uint8_t* bitmaps, int32_t n, int32_t H0, int32_t W0, int64_t* stats, void* stream) {
    if (!c || n < 0 || H0 < 1 || W0 < 1 || (n && (!bitmaps || !stats)))
        return fail(c, VTI_ERR_ARG, "vti_mask_stats: bad argument");
    if (int32_t drc = check_device(c, "vti_mask_stats")) return drc;
    VTI_HIP(c, launch_mask_stats(bitmaps, n, H0, W0, (long long*)stats, (hipStream_t)stream), "mask_stats kernel");
    return VTI_OK;
}

int32_t vti_mask_stats_bits(vti_ctx* c, const uint8_t* masks_bits, int32_t n, const int32_t* n_live, int32_t H, int32_t W,
                            int32_t H0, int32_t W0, int64_t* stats, void* stream) {
    if (n < 0 || H < 1 || W < 32 || (W & 31) || H0 < 1 || W0 < 1 || (n && (!masks_bits || !stats)))
        return fail(c, VTI_ERR_ARG, "vti_mask_stats_bits: bad argument (W must be a multiple of 32)");
    if (int32_t drc = check_device(c, "vti_mask_stats_bits")) return drc;
    VTI_HIP(c, launch_mask_stats_bits(masks_bits, n, n_live, H, W, H0, W0, (long long*)stats, (hipStream_t)stream), "mask_stats_bits kernel");
    return VTI_OK;
}

int32_t vti_envelope_bits(vti_ctx* c, const uint8_t* masks_bits, const int32_t* offsets, const float* dets, int32_t B,
                          int32_t max_det, int32_t capacity, int32_t cls, int32_t H0, int32_t W0, int32_t* envelope, void* stream) {
    if (!c || B < 0 || max_det < 1 || capacity < 0 || H0 < 1 || W0 < 1 || (B && (!masks_bits || !offsets || !dets || !envelope)))
        return fail(c, VTI_ERR_ARG, "vti_envelope_bits: bad argument");
    const vti_desc& d = c->plan.desc;
    if (int32_t drc = check_device(c, "vti_envelope_bits")) return drc;
    VTI_HIP(c, launch_envelope_bits(masks_bits, offsets, dets, B, max_det, d.nm, capacity, cls, d.H, d.W, H0, W0, envelope,
                                    (hipStream_t)stream), "envelope_bits kernel");
    return VTI_OK;
}

int32_t vti_pixels_to_world(vti_ctx* c, const double* uv, int32_t n, const double* K, const double* dist, const double* R,
                            const double* t, double* xyz, int32_t* valid, void* stream) {
    if (n < 0 || !K || !dist || !R || !t || (n && (!uv || !xyz || !valid)))
        return fail(c, VTI_ERR_ARG, "vti_pixels_to_world: bad argument");
    VTI_HIP(c, launch_pixels_to_world(uv, n, K, dist, R, t, xyz, valid, (hipStream_t)stream), "pixels_to_world kernel");
    return VTI_OK;
}

int32_t vti_kmeans1d2(vti_ctx* c, const double* values, const int32_t* counts, int32_t B, int32_t max_n, int32_t max_iters,
                      int32_t* labels, double* centers, void* stream) {
    if (B < 0 || max_n < 1 || max_n > 1024 || max_iters < 0 || (B && (!values || !counts || !labels || !centers)))
        return fail(c, VTI_ERR_ARG, "vti_kmeans1d2: bad argument (max_n <= 1024)");
    VTI_HIP(c, launch_kmeans1d2(values, counts, B, max_n, max_iters, labels, centers, (hipStream_t)stream), "kmeans1d2 kernel");
    return VTI_OK;
}

int32_t vti_debug_conv_output(vti_ctx* c, int32_t i, int32_t B, float* out, void* stream) {
    int32_t rc = check_ready(c, B, "vti_debug_conv_output");
    if (rc) return rc;
    if (i < 0 || i >= (int32_t)c->plan.convs.size() || !out) return fail(c, VTI_ERR_ARG, "vti_debug_conv_output: bad argument");
    const View& v = c->plan.conv_out[i];
    if (v.buf < 0) return fail(c, VTI_ERR_UNSUPPORTED, "vti_debug_conv_output: this conv is fused into the next one; its output is never materialised");
    const Buf& b = c->plan.bufs[v.buf];
    const void* src = buf_ptr(c, v.buf, c->last_input, c->last_proto);
    if (!src) return fail(c, VTI_ERR_STATE, "vti_debug_conv_output: run vti_forward first");
    const int is_f32 = (b.elem == EL_F32 || (b.elem == EL_T && c->plan.desc.dtype == VTI_F32)) ? 1
                       : (b.elem == EL_T && c->plan.desc.dtype == VTI_H2) ? 2 : 0;      // 2: h2 pairs
    VTI_HIP(c, launch_debug_nchw(is_f32, src, B, b.H, b.W, v.C, b.C, v.coff, out, (hipStream_t)stream), "debug copy");
    return VTI_OK;
}

int32_t vti_debug_conv2d(int32_t dtype, const void* dev_in, int32_t B, int32_t H, int32_t W, int32_t in_ld, int32_t in_coff,
                         int32_t c1, const float* host_w, const float* host_b, int32_t c2, int32_t k, int32_t s, int32_t kind,
                         const void* dev_res, int32_t res_ld, int32_t res_coff, void* dev_out, int32_t out_ld, int32_t out_coff,
                         int32_t out_f32, int32_t swap_rb, int32_t tile_h, int32_t tile_w, int32_t waves_n, int32_t nrep,
                         int32_t iters, float* ms_out, int32_t* cfg_out, void* stream) {
    if (!dev_in || !host_w || !host_b || !dev_out || B < 1 || H < 1 || W < 1 || c1 < 1 || c2 < 1 || iters < 1)
        return fail(nullptr, VTI_ERR_ARG, "vti_debug_conv2d: bad argument");
    if ((dtype != VTI_F16 && dtype != VTI_F32 && dtype != VTI_H2) || (kind < 0 || kind > 2))
        return fail(nullptr, VTI_ERR_ARG, "vti_debug_conv2d: bad dtype/kind");
    const bool conv0 = (c1 == 3);   // u8 HWC3 input, the stem conv
    if (conv0 ? !(k == 3 && s == 2 && kind == 0) : (c1 % 16 != 0))
        return fail(nullptr, VTI_ERR_ARG, "vti_debug_conv2d: c1 must be 3 (stem) or a multiple of 16");
    if (!((k == 1 && s == 1) || (k == 3 && (s == 1 || s == 2)) || (kind == 2 && k == 2 && s == 2)))
        return fail(nullptr, VTI_ERR_ARG, "vti_debug_conv2d: unsupported kernel/stride");
    ConvRow r;
    r.name = "debug"; r.c1 = c1; r.c2 = c2; r.k = k; r.s = s; r.kind = kind; r.h_in = H; r.w_in = W;
    if (kind == 2) { r.h_out = 2 * H; r.w_out = 2 * W; }
    else { r.h_out = (H + 2 * (k / 2) - k) / s + 1; r.w_out = (W + 2 * (k / 2) - k) / s + 1; }
    ConvCfg g;
    choose_conv_cfg(dtype, r, conv0, B, g, tile_h, tile_w, waves_n, nrep);
    if (g.TH == 0) return fail(nullptr, VTI_ERR_ARG, "vti_debug_conv2d: no launch configuration fits");
    if (cfg_out) { cfg_out[0] = g.TH; cfg_out[1] = g.TW; cfg_out[2] = g.WN; cfg_out[3] = g.NREP; cfg_out[4] = (int32_t)g.lds * (g.pk ? -1 : 1); }   // negative LDS size = persistent kernel
    std::vector<uint8_t> wpk(packed_conv_bytes(r, conv0, g));
    std::vector<float> bias((size_t)g.ntiles_n * 16);
    float alpha = 1.f;
    pack_conv(dtype, r, conv0, g, host_w, host_b, wpk.data(), bias.data(), 0, &alpha);
    void* d_w = nullptr; float* d_b = nullptr;
    hipStream_t st = (hipStream_t)stream;
    VTI_HIP(nullptr, hipMalloc(&d_w, wpk.size()), "hipMalloc");
    hipError_t e = hipMalloc((void**)&d_b, bias.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(d_w, wpk.data(), wpk.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) {
        ConvParams p;
        fill_conv_params(dtype == VTI_F16 ? 2 : 4, p, r, g, B, dev_in, in_ld, in_coff, dev_out, out_ld, out_coff, dev_res, res_ld, res_coff, d_w, d_b,
                         out_f32 != 0, swap_rb);
        p.alpha = alpha; p.alpha2 = p.alpha0 = 1.f;
        p.pk_stagger *= stagger_units(dtype);
        const int ks = kind == 2 ? 1 : k, ss = kind == 2 ? 1 : s;
#ifdef VTI_STAMPS
        {   // diagnostic build: one stamped launch, medians of the phase intervals to stderr
            const int NTB = g.WN * g.NREP;
            const size_t nwg = (p.pk ? (size_t)p.pk_wgs : (size_t)B * p.tiles_y * p.tiles_x) * ((g.ntiles_n + NTB - 1) / NTB);
            unsigned long long* d_st = nullptr;
            if (hipMalloc((void**)&d_st, nwg * 16 * 8) == hipSuccess) {
                (void)hipMemset(d_st, 0, nwg * 16 * 8);
                (void)launch_conv(dtype, ks, ss, g.NREP, conv0 ? 1 : 0, p, g.lds, st);   // warm caches/icache
                p.stamps = d_st;
                (void)launch_conv(dtype, ks, ss, g.NREP, conv0 ? 1 : 0, p, g.lds, st);
                (void)hipStreamSynchronize(st);
                std::vector<unsigned long long> h(nwg * 16);
                (void)hipMemcpy(h.data(), d_st, nwg * 16 * 8, hipMemcpyDeviceToHost);
                p.stamps = nullptr;
                (void)hipFree(d_st);
                report_stamps(h, nwg, p.pk != 0);
            }
        }
#endif
        e = launch_conv(dtype, ks, ss, g.NREP, conv0 ? 1 : 0, p, g.lds, st);   // warm-up / the checked run
        if (e == hipSuccess) e = hipEventRecord(e0, st);
        for (int i = 1; i < iters && e == hipSuccess; ++i) e = launch_conv(dtype, ks, ss, g.NREP, conv0 ? 1 : 0, p, g.lds, st);
        if (e == hipSuccess) e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        float ms = 0.f;
        if (e == hipSuccess && iters > 1) e = hipEventElapsedTime(&ms, e0, e1);
        if (ms_out) *ms_out = iters > 1 ? ms / (iters - 1) : 0.f;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_w);
    if (d_b) (void)hipFree(d_b);
    if (e != hipSuccess) return hip_fail(nullptr, e, "vti_debug_conv2d");
    return VTI_OK;
}

}  // extern "C"
