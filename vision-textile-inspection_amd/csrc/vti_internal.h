// Internal declarations shared by the host-side plan/API (vti_api.cpp, plan.cpp, weights.cpp)
// and the gfx950 kernels (conv.hip, misc.hip, post.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vti.h"

namespace vti {

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) is a per-device setting: the "already done" flags of the launchers are kept
// per device, so a process that holds contexts on several GPUs sets it on each of them.
constexpr int kMaxDevices = 64;
inline int current_device_slot() { int d = 0; (void)hipGetDevice(&d); return (unsigned)d < (unsigned)kMaxDevices ? d : 0; }

// ---- plan ---------------------------------------------------------------------------
enum ElemKind { EL_T = 0, EL_F32 = 1, EL_U8 = 2 };   // EL_T = ctx dtype (fp16, fp32, or h2 = split-fp16 pairs, 4 bytes)

struct Buf {             // one NHWC activation tensor [max_batch, H, W, C]
    int C, H, W;
    int elem;            // ElemKind
    size_t off;          // byte offset inside the workspace
    size_t bytes;        // for max_batch frames
};

struct View {            // channel slice of a Buf (concat-free C2f / neck)
    int buf = -1, coff = 0, C = 0;
};

struct ConvRow {         // fused conv table, Ultralytics order
    std::string name;
    int c1, c2, k, s, kind;
    int h_in, w_in, h_out, w_out;
    int64_t macs() const {
        const int64_t hw = (kind == 2) ? (int64_t)h_in * w_in : (int64_t)h_out * w_out;
        return hw * c1 * c2 * k * k;
    }
    int64_t fused_params() const { return (int64_t)c1 * c2 * k * k + c2; }
};

enum OpKind { OP_CONV0, OP_CONV, OP_POOL, OP_UP2, OP_DECODE, OP_FORK, OP_JOIN };
constexpr int kNumLanes = 10;         // lane 0 = the caller's stream; 1..9 = ctx-owned side streams

struct ConvCfg {         // launch geometry chosen at plan time
    int TH = 0, TW = 0;  // output tile (pixels)
    int WN = 1;          // waves along Cout (1,2,4); 4/WN waves along pixels
    int NREP = 1;        // 16-wide cout tiles per wave
    int nchunks = 0;     // K chunks (32 ch fp16 / 16 ch fp32)
    int ntiles_n = 0;    // ceil(gemmN/16)
    int gemm_n = 0;      // Cout (deconv: 4*Cout)
    size_t lds = 0;
    size_t wpk_off = 0;  // byte offset of this conv's packed weights
    size_t bias_off = 0; // float offset of this conv's bias
    // fused second stage (a 1x1 conv applied to this conv's register tile), 0 tiles = none
    int ntiles2 = 0, gemm_n2 = 0;
    size_t wpk_off2 = 0, bias_off2 = 0;
    size_t wpk_off3 = 0, bias_off3 = 0;   // stem_l1_kernel with a fused 1x1 third conv: that conv's stage-2 pack (offsets 2 = layer 1)
    // persistent LDS-DMA kernel (conv_pk.hip): TW = 20, TH = 4 * M-waves; wgpc = co-resident workgroups per CU
    int pk = 0, pk_wgpc = 1;         // pk: 1 = conv3_pk, 2 = conv1_pk (TH = compute waves along M, TW = 80), 3 = bneck_pk (fused 3x3 -> 3x3 pair), 4 = conv3_pk stride 2
    int pk_depth = 2, pk_wstat = 0;  // conv1_pk: stage-ring depth, weights stationary in LDS
    int pk_cps = 1;                  // conv1_pk: K chunks per step
    int threads = 256;               // per-tile kernel: 256, or 512 (fused towers on wide maps: one 8-wave workgroup per CU, 16 x 40 tiles)
};

struct Op {
    OpKind kind;
    int conv = -1;       // index into convs (OP_CONV0 / OP_CONV)
    View in, out, res;
    bool has_res = false;
    bool out_f32 = false;
    int fused = -1;      // conv index of a 1x1 conv fused into this op's epilogue (its own op is dropped)
    int tail = -1;       // bneck_pk only: conv index of the C2f's closing 1x1 computed in the same kernel (out2 = its output; y2 is not stored)
    int pair = -1;       // conv index of the SECOND 3x3 of a fused C2f Bottleneck (bneck_pk): this op is the first; out/res are the second's
    int fold = -1;       // conv index of a ConvTranspose2d(2,2) FOLDED into this 3x3 conv (convfold_kernel): `in` is then the deconv's input
    int fused_l1 = -1;   // OP_CONV0 only: conv index of layer 1 computed by the same kernel (stem_l1_kernel); out/out2 = layer 1's view
    int pred_mode = 0, pred_cbase = 0, pred_a0 = 0;   // fused stage writes into pred instead of a head buffer (1 raw, 2 sigmoid, 3 DFL boxes)
    int dfl_stride = 0;      // pred_mode 3: the level's stride (box tower: DFL expectation + dist2bbox in the epilogue)
    int nat2 = 0;            // fused stage with an fp32 NHWC output: natural channel order (lane group g owns channels 16n+4g..+3: one
                             // store instruction then covers 64 contiguous bytes per pixel instead of four 16-byte pieces 64 B apart)
    View out2;           // where the fused conv writes
    bool out2_f32 = false;
    View up_src; int up_C = 0;   // conv1_pk only: channels [0, up_C) of `in` are read as the nearest-2x upsample of up_src (no UP2 op)
    int lane = 0;        // stream the op is enqueued on (OP_FORK/OP_JOIN: the side lane that starts/finishes)
    ConvCfg cfg;
};

struct Level { int C, H, W, stride; int box_buf, cls_buf, mc_buf; };

struct Plan {
    vti_desc desc;
    int esize = 2;                       // bytes per EL_T element
    std::vector<ConvRow> convs;
    std::vector<View> conv_out;          // where each conv's result lives (debug hook)
    std::vector<Buf> bufs;
    std::vector<Op> ops;
    std::vector<Level> levels;
    int proto_buf_c = 0;                 // npr
    View proto_src;                      // input of proto.cv3 (so cv3 writes straight to the caller's proto)
    int num_anchors = 0;
    bool pred_scatter = false;           // class/coefficient towers write pred directly; decode handles boxes only
    bool dfl_fused = false;              // ... and the box towers decode their own boxes: no decode op at all
    size_t ws_bytes = 0;
    size_t wpk_bytes = 0, bias_floats = 0;
    int64_t macs = 0, fused_params = 0;
    std::string build(const vti_desc& d); // returns "" or an error message
};

// ---- kernel parameter blocks ------------------------------------------------------------
struct ConvParams {
    const void* in; void* out; const void* res; const void* wpk; const float* bias;
    int B, Hin, Win, Hout, Wout;         // Hout/Wout: conv output grid (deconv: == Hin/Win, stores on the 2x grid)
    int Cin, in_ld, in_coff;
    int Cout;                            // GEMM N
    int out_ld, out_coff, res_ld, res_coff;
    int TH, TW, tiles_y, tiles_x, WN;
    int act, out_f32, deconv_c, swap_rb, nchunks, ntiles_n, has_res, scalar_store;
    unsigned pw_magic, rw_magic, tw_magic;   // ceil(2^32 / {PW, raw-row-bytes, TW}): division-free indexing
    unsigned wpk_bytes;                  // bytes of this conv's packed weights (buffer-load range check)
    // fused 1x1 second stage: out2 = act2(W2 . silu(conv + bias) + bias2), never touching HBM in between
    const void* w2; const float* bias2; void* out2;
    int Cout2, ntiles2, out2_ld, out2_coff, act2 /*0 none, 1 SiLU, 2 sigmoid, 3 DFL + dist2bbox*/, out2_f32, scalar_store2, nat2, out2_bstride;
    int nt;           // threads per workgroup of the per-tile kernel (256 / 512)
    float* best;      // act2 == 2 (class scores into pred): also (max score, its first class) per anchor -> best[(b * out2_bstride + pixel) * 2], or null
    float dfl_stride;
    // persistent kernel (conv_pk.hip): tile count, workgroups along x, XCD-contiguous tile ranges, tensor sizes
    int pk, pk_tiles, pk_wgs, pk_xcd, pk_depth, pk_wstat, pk_lin /* 1: linear pixel -> column-tile map (A/B aid) */;
    int pk_cps;                          // conv1_pk: K chunks per step (1 fp16, 2 for 4-byte storage)
    int pk_stagger;                      // conv3_pk: the upper half of every XCD's workgroups starts its compute waves this many x 1024 cycles late (0: off)
    unsigned in_bytes, out_bytes, res_bytes, out2_bytes;
    // conv1_pk: channels [0, up_C) come from in2 [B, Hout/2, Wout/2, in2_ld] at (y >> 1, x >> 1): the neck's Upsample + Concat folded into the loads
    const void* in2; int in2_ld, in2_coff, up_C; unsigned in2_bytes;
    int fold;                            // convfold_kernel: stage-2 output grid is 2 Hout x 2 Wout, stage-1 bias = p.bias[border class][64]
    const void* w0; const float* bias0;  // stem_l1_kernel: the stem's packed weights / bias (wpk/bias = layer 1, w2/bias2 = the fused 1x1 third conv)
    // VTI_H2 only: accumulator scales 1 / (SW * 16) of the weights behind wpk / w2 / w0 (weights.cpp: emit_packed); 1 otherwise
    float alpha, alpha2, alpha0;
    unsigned long long* stamps;          // diagnostic build only (VTI_STAMPS): 16 s_memtime slots per workgroup
};

// dtype: VTI_F16/VTI_F32; mode 0 = NHWC conv, 1 = conv0 (u8 input, im2col K=27->32)
hipError_t launch_conv(int dtype, int ks, int stride, int nrep, int mode, const ConvParams& p,
                       size_t lds_bytes, hipStream_t st);
bool conv_fusable(int nrep, int nrep2);   // is there a (3x3 NREP) + (1x1 NREP2) fused instantiation
size_t conv_lds_bytes(int ks, int stride, int mode, int TH, int TW, int WN, int NREP);
bool conv_cfg_fits(int ks, int stride, int mode, int TH, int TW, int WN, int NREP, int threads = 256);
// stem (3->16, k3 s2) + layer 1 (16->32, k3 s2) in one kernel: 16 x 20 layer-1 output tiles
hipError_t launch_stem_l1(int dtype, const ConvParams& p, hipStream_t st);
size_t stem_l1_lds_bytes(int dtype);
void stem_l1_tile(int* th, int* tw);
int stem_l1_grid(int dtype, int ntiles);     // workgroups of the (persistent) launch
void pack_stem_toeplitz(int dtype, const ConvRow& r0, const float* w, const float* b, uint8_t* dst_w, float* dst_b, float* alpha = nullptr);
size_t packed_stem_toeplitz_bytes(int dtype);
// ConvTranspose2d(C,C,2,2) folded into the following 3x3 conv (+ its fused 1x1): four 2x2 convs on the low-resolution map
hipError_t launch_convfold(int dtype, const ConvParams& p, size_t lds_bytes, hipStream_t st);
size_t convfold_lds_bytes(int TH, int TW);
bool convfold_supported(int c_in, int c_mid, int c_out, int ntiles2);
void pack_conv_fold(int dtype, const ConvRow& rU, const ConvRow& rV, const ConvCfg& c, const float* wU, const float* bU,
                    const float* wV, const float* bV, uint8_t* dst_w, float* dst_b, float* alpha = nullptr);
size_t packed_fold_bytes(const ConvCfg& c);
hipError_t launch_conv_pk_fold(int dtype, const ConvParams& p, size_t lds_bytes, hipStream_t st);   // the same on the persistent schedule
size_t conv_pk_fold_lds_bytes(int nchunks, int depth);
int conv_pk_fold_depth(int nchunks);
void pack_conv_l1pairs(int dtype, const ConvRow& r1, const float* w, const float* b, uint8_t* dst_w, float* dst_b, float* alpha = nullptr);
size_t packed_l1pairs_bytes(int dtype);
// conv_pk.hip: persistent 3x3/s1 kernel
hipError_t launch_conv_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st);
size_t conv_pk_lds_bytes(int TH, int WN, int NREP, int nchunks);
size_t conv_pk_lds_bytes(int TH, int WN, int NREP, int nchunks, int depth, int wstat = 0);   // wstat: all K chunks' weights resident in LDS
int conv_pk_depth(int TH, int WN, int NREP, int nchunks, int wstat = 0);
// stride-2 3x3 on the persistent schedule (ConvCfg.pk == 4)
hipError_t launch_conv_pk2(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st);
size_t conv_pk2_lds_bytes(int TH, int WN, int NREP, int nchunks, int depth, int wstat = 0);
bool conv_pk2_fits(int TH, int WN, int NREP, int nchunks, int wstat = 0);
bool conv_pk2_instantiated(int nrep, int wn);
int conv_pk2_depth(int TH, int WN, int NREP, int nchunks, int wstat = 0);
bool conv_pk_fits(int TH, int WN, int NREP, int nchunks, int wstat = 0);
bool conv_pk_instantiated(int nrep, int wn);
hipError_t launch_conv1_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st);
size_t conv1_pk_lds_bytes(int nwm, int WN, int NREP, int nchunks, int depth, int wstat, int cps = 1);   // cps: K chunks per step
bool conv1_pk_fits(int nwm, int WN, int NREP, int nchunks, int depth, int wstat, int cps = 1);
bool conv1_pk_instantiated(int nrep, int wn);
hipError_t launch_bneck_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st);
size_t bneck_pk_lds_bytes(int TH, int NREP);
size_t bneck_pk_lds_bytes(int TH, int NREP, int depth);
int bneck_pk_depth(int TH, int NREP);
bool bneck_pk_fits(int TH, int NREP);

struct PoolParams { const void* in; void* out; int B, H, W, C, ld, in_coff, out_coff; };
hipError_t launch_sppf_pool(int dtype, const PoolParams& p, hipStream_t st);

struct Up2Params { const void* in; void* out; int B, H, W, C, in_ld, in_coff, out_ld, out_coff; };
hipError_t launch_upsample2x(int dtype, const Up2Params& p, hipStream_t st);

struct DecodeParams {
    const float* box[3]; const float* cls[3]; const float* mc[3];
    int H[3], W[3], stride[3], a0[3];
    int B, A, nc, nm, reg_max;
    float* pred;                          // [B, 4+nc+nm, A]
};
hipError_t launch_decode(const DecodeParams& p, hipStream_t st);
hipError_t launch_box_decode(const DecodeParams& p, hipStream_t st);   // DFL + dist2bbox only (pred_scatter plans)

hipError_t launch_debug_nchw(int elem_is_f32, const void* src, int B, int H, int W, int C, int ld, int coff,
                             float* dst, hipStream_t st);

// post-processing (post.hip)
hipError_t launch_letterbox(const uint8_t* frames, int B, int H0, int W0, uint8_t* out, int H, int W,
                            int new_h, int new_w, int top, int left, hipStream_t st);
size_t nms_workspace_bytes(int B, int A);
// `best`: optional [B, A, 2] floats (max class score, its first class index as a float) that the producer of `pred` wrote beside it
// (vti_forward_scored): the candidate filter then reads 8 bytes per anchor instead of the nc class scores
hipError_t launch_nms(const float* pred, const float* best, int B, int A, int nc, int nm, float conf, double iou, int max_det,
                      int agnostic, float* dets, int* counts, void* ws, hipStream_t st);
hipError_t launch_anchor_best(const float* pred, int B, int A, int nc, int nm, float* best, hipStream_t st);   // the same pairs from pred itself
float* nms_workspace_best(void* ws, int B, int A);       // [B, A, 2] floats at the end of the NMS workspace (vti_predict's own pair buffer)
hipError_t launch_masks(int dtype, const float* dets, const int* counts, const void* proto, int B, int max_det,
                        int nm, int Hp, int Wp, int H, int W, int mode, int packing, uint8_t* masks,
                        int capacity, int* offsets, void* ws, hipStream_t st);
size_t masks_workspace_bytes(int capacity, int H, int W);
hipError_t launch_scale_boxes(const float* dets, const int* counts, int B, int max_det, int nm, int H, int W,
                              int H0, int W0, float* xyxy, hipStream_t st);
hipError_t launch_mask_to_frame(const uint8_t* masks, int n, int H, int W, int H0, int W0, uint8_t* bitmaps,
                                int* nonzero, hipStream_t st);
hipError_t launch_union_envelope(const uint8_t* bitmaps, const int* select, int nsel, int H0, int W0,
                                 uint8_t* uni, int* envelope, hipStream_t st);
hipError_t launch_mask_stats(const uint8_t* bitmaps, int n, int H0, int W0, long long* stats, hipStream_t st);

// consumer.hip: reductions on bit-packed masks, pixel -> world geometry, 1-D 2-means
hipError_t launch_mask_stats_bits(const uint8_t* bits, int n, const int* n_live, int H, int W, int H0, int W0, long long* stats,
                                  hipStream_t st);
hipError_t launch_envelope_bits(const uint8_t* bits, const int* offsets, const float* dets, int B, int max_det, int nm,
                                int capacity, int cls, int H, int W, int H0, int W0, int* envelope, hipStream_t st);
hipError_t launch_pixels_to_world(const double* uv, int n, const double* K, const double* dist, const double* R,
                                  const double* t, double* xyz, int* valid, hipStream_t st);
hipError_t launch_kmeans1d2(const double* values, const int* counts, int B, int max_n, int max_iters, int* labels,
                            double* centers, hipStream_t st);

// plan.cpp: launch geometry for one conv (tile, wave split, LDS) -- th/tw/wn/nrep > 0 force a choice
void choose_conv_cfg(int dtype, const ConvRow& r, bool conv0, int max_batch, ConvCfg& c,
                     int th = 0, int tw = 0, int wn = 0, int nrep = 0, bool allow_pk = true);
// weights.cpp: one conv's weights -> fragment order; dst_w has cfg.nchunks*ntiles_n*taps KiB, dst_b ntiles_n*16 floats
void pack_conv(int dtype, const ConvRow& r, bool conv0, const ConvCfg& c, const float* w, const float* b,
               uint8_t* dst_w, float* dst_b, int cin_off = 0, float* alpha = nullptr);      // cin_off: the conv's input channels sit at K positions cin_off.. of the chunk; alpha: see ConvParams
size_t packed_conv_bytes(const ConvRow& r, bool conv0, const ConvCfg& c);
// fused second stage: the 1x1 conv `r2` packed against the accumulator layout of a producer with nrep1 cout tiles
void pack_conv_stage2(int dtype, const ConvRow& r2, int nrep1, const float* w, const float* b, uint8_t* dst_w, float* dst_b,
                      bool natural_rows = false, float* alpha = nullptr);
size_t packed_stage2_bytes(int dtype, const ConvRow& r2, int nrep1);
// weights.cpp: parse VTIW1 + pack into MFMA fragment order (host memory)
std::string pack_weights(const Plan& plan, const void* blob, size_t nbytes,
                         std::vector<uint8_t>& wpk, std::vector<float>& bias, std::vector<float>& alpha);

}  // namespace vti
