/*
 * vti.h -- C ABI of libvti.so: MI355X (gfx950) YOLOv8-seg inference hot path.
 *
 * Drop-in boundary for the reference's `ultralytics.YOLO` object protocol
 * (RishWijewardhena/vision-textile-inspection):
 *     model = YOLO(path)                                   measurement.py:145
 *     model.predict(img, conf=, iou=, max_det=, imgsz=)    measurement.py:208-210,
 *                                                          Utils/check_model.py:331-337
 *     Results.boxes.{xyxy,cls,conf}, Results.masks.data    measurement.py:74-75,242-245
 * and for measurement.py's mask post-processing (measurement.py:70-86,160-185,300-330).
 *
 * The reference has no FFI of its own; the binding a maintainer adds is the ctypes
 * shim shown in INTEGRATION.md (shipped as vti_amd/ in this repo).
 *
 * Conventions
 *  - Every function returns VTI_OK (0) or a negative vti_status; it never aborts or
 *    exits the process (measurement.py:207-216 catches every predict exception and
 *    keeps running).  vti_last_error(ctx) gives a message for the last failure.
 *  - All `dev_*` pointers are DEVICE pointers owned by the caller (torch-ROCm tensors'
 *    data_ptr()).  The library never allocates caller-visible memory; its only device
 *    allocation is the packed weight image made by vti_load_weights.
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *    Calls enqueue work on it and return; no hidden synchronisation.
 *  - One ctx per (device, host thread) (main.py:187-211: single-threaded use).  A ctx is bound to the device passed to
 *    vti_load_weights; entry points that launch work return VTI_ERR_STATE when another device is current.
 *  - Multi-GPU is one process per GPU above this ABI: frames are independent, so the only exchange steps
 *    are a scatter of uint8 frames and a gather of detections + consumer reductions, done by the host shim with
 *    torch.distributed (backend "nccl" = RCCL over xGMI; vti_amd/dataparallel.py).  The library itself opens no communicator
 *    and exports no vti_dp_* entry points: there is no collective inside the model.
 *  - Tensor layouts (T = fp16 for VTI_F16, fp32 for VTI_F32 and VTI_H2):
 *      frames   u8  [B,H0,W0,3]           camera frames, any channel order (see swap_rb)
 *      input    u8  [B,H,W,3]             letterboxed frames (H,W multiples of 32)
 *      pred     f32 [B,A,4+nc+nm]         decoded head output, ANCHOR-MAJOR: Ultralytics' [B,4+nc+nm,A] transposed, so that
 *                                         an anchor's box, class scores and mask coefficients are one contiguous row (the head
 *                                         towers write it and NMS reads it row-wise; the Python shim hands out the
 *                                         [B,4+nc+nm,A] view of the same memory)
 *                                          (cx,cy,w,h in letterboxed px; sigmoid class scores; coeffs)
 *      proto    T   [B,H/4,W/4,nm]        mask prototypes, NHWC
 *      dets     f32 [B,max_det,6+nm]      rows x1,y1,x2,y2,conf,cls,coeff[nm]; conf-descending;
 *                                          boxes in letterboxed px (vti_scale_boxes maps to frame px)
 *      counts   i32 [B]                   detections per frame
 *      masks    u8  [cap,H,W] (VTI_PACK_U8, 0/1 per byte) or u8 [cap,H,W/8] (VTI_PACK_BITS,
 *               LSB-first); instance i of frame b lives in slot offsets[b]+i
 *      offsets  i32 [B+1]                 exclusive prefix sum of counts (written by vti_masks)
 */
#ifndef VTI_H
#define VTI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vti_ctx vti_ctx;

typedef enum {
    VTI_OK = 0,
    VTI_ERR_ARG = -1,        /* bad argument / shape */
    VTI_ERR_STATE = -2,      /* weights or workspace not set */
    VTI_ERR_WEIGHTS = -3,    /* container does not match the model description */
    VTI_ERR_HIP = -4,        /* HIP runtime error (message in vti_last_error) */
    VTI_ERR_NOMEM = -5,      /* workspace too small */
    VTI_ERR_UNSUPPORTED = -6
} vti_status;

/* Storage type of weights and activations.  VTI_H2 = split-fp16: every element is the fp16 pair (hi, lo) of value * 16
 * (22-23 significant bits) and every product runs on the fp16 matrix pipe (two 16x16x32 MFMAs per 16 channels, fp32
 * accumulation): the results meet the reference tolerance (mask IoU >= 0.999, |d box| < 1e-3) like VTI_F32 at ~4x its matrix rate.
 * proto is f32 for VTI_H2 and VTI_F32, fp16 for VTI_F16. */
enum { VTI_F16 = 0, VTI_F32 = 1, VTI_H2 = 2 };
enum { VTI_MASK_LOGIT = 0,     /* current Ultralytics: crop, bilinear upsample, > 0.0 */
       VTI_MASK_SIGMOID = 1 }; /* Ultralytics 8.0.x : sigmoid, crop, upsample, > 0.5  */
enum { VTI_PACK_U8 = 0, VTI_PACK_BITS = 1 };
/* vti_masks work-list size: one call handles at most max_batch * VTI_MASK_SLOTS_PER_FRAME instances (capacity above that is
 * VTI_ERR_UNSUPPORTED); vti_workspace_bytes() is sized for it.  Ultralytics' default max_det is 300, the reference's 200. */
#define VTI_MASK_SLOTS_PER_FRAME 512

/* Model description: replaces what YOLO(model_path) reads out of the .pt (measurement.py:145). */
typedef struct {
    char scale;        /* 'n','s','m','l','x' */
    int32_t nc;        /* classes (reference: 2 -- stitch=0, fabric=1, config.py:69-70) */
    int32_t nm;        /* mask coefficients, 32 */
    int32_t reg_max;   /* DFL bins, 16 */
    int32_t H, W;      /* letterboxed input size, multiples of 32 */
    int32_t max_batch; /* largest B any call will pass */
    int32_t dtype;     /* VTI_F16 / VTI_F32 / VTI_H2: storage type of weights and activations */
} vti_desc;

/* One row of the fused conv table (replaces walking model.model[*] of the unpickled net). */
typedef struct {
    char name[48];               /* Ultralytics tensor-name prefix, e.g. "model.2.m.0.cv1" */
    int32_t c1, c2, k, s, kind;  /* kind: 0 conv+BN+SiLU (folded), 1 conv+bias, 2 ConvTranspose2d+bias */
    int32_t h_in, w_in, h_out, w_out;
    int64_t macs;                /* multiply-accumulates per frame */
    int32_t tile_h, tile_w;      /* launch geometry: output pixels per workgroup tile */
    int32_t waves_n, nrep;       /* waves along Cout, 16-wide cout tiles per wave */
    int32_t lds_bytes;
    int32_t fused;               /* 1: this 1x1 conv runs inside the previous conv's kernel (register-level fusion) */
    int32_t persistent;          /* 1: persistent LDS-DMA kernel (conv_pk.hip): tile_h/4 * waves_n compute waves + as many loader waves */
} vti_conv_info;

/* ---- lifetime -------------------------------------------------------------------- */
/* Builds the network plan on the host.  Touches no GPU: usable on a CPU-only box. */
int32_t vti_create(const vti_desc* desc, vti_ctx** out);
void vti_destroy(vti_ctx* ctx);
const char* vti_last_error(const vti_ctx* ctx);   /* ctx may be NULL: last create error */

/* ---- plan introspection (host only) ------------------------------------------------ */
int32_t vti_num_convs(const vti_ctx* ctx);
int32_t vti_conv_at(const vti_ctx* ctx, int32_t i, vti_conv_info* out);
int32_t vti_num_anchors(const vti_ctx* ctx);
int64_t vti_fused_params(const vti_ctx* ctx);     /* incl. the frozen DFL arange, as model.info() */
int64_t vti_macs_per_frame(const vti_ctx* ctx);   /* conv + deconv MACs */
int64_t vti_workspace_bytes(const vti_ctx* ctx);  /* for max_batch frames */
int32_t vti_num_launches(const vti_ctx* ctx);     /* kernels per vti_forward call */

/* ---- device setup ---------------------------------------------------------------- */
/* Parses a VTIW1 container held in HOST memory (fused OIHW fp32 + bias per conv, table
 * order), repacks it into MFMA fragment order and uploads it to `device`. */
int32_t vti_load_weights(vti_ctx* ctx, const void* host_blob, size_t nbytes, int32_t device);
/* Caller-owned device scratch of at least vti_workspace_bytes(), 256-B aligned. */
int32_t vti_set_workspace(vti_ctx* ctx, void* dev_ws, size_t nbytes);

/* ---- the hot path: stages of predict() ------------------------------------------- */
/* U1 LetterBox: resize (OpenCV u8 INTER_LINEAR fixed point; the rounded 2x2 box mean OpenCV substitutes -- INTER_AREA --
 * when the frame is exactly twice the resized size) + pad 114 to HxW.  A frame already HxW is copied through. */
int32_t vti_letterbox(vti_ctx* ctx, const uint8_t* dev_frames, int32_t B, int32_t H0, int32_t W0,
                      uint8_t* dev_input, void* stream);
/* U2-U5 network forward.  swap_rb=1 reproduces Ultralytics' channel flip of ndarray sources. */
int32_t vti_forward(vti_ctx* ctx, const uint8_t* dev_input, int32_t B, int32_t swap_rb,
                    float* dev_pred, void* dev_proto, void* stream);
/* U6 non_max_suppression (class-aware unless agnostic), torchvision.ops.nms semantics. */
int32_t vti_nms(vti_ctx* ctx, const float* dev_pred, int32_t B, float conf, double iou,
                int32_t max_det, int32_t agnostic, float* dev_dets, int32_t* dev_counts, void* stream);
/* U5 -> U6 hand-over without re-reading the class scores: vti_forward_scored also writes, for every anchor, the pair (best class
 * score, index of the first class that has it, as a float) into dev_anchor_best (f32 [B, A, 2]) -- from the class towers' epilogue,
 * where the scores are in registers -- and vti_nms_scored takes its candidates (score > conf) from those 8 bytes per anchor instead of
 * scanning nc scores per anchor of dev_pred (Ultralytics: x[:, 4:4+nc].amax(1) > conf_thres, the first thing non_max_suppression
 * does).  Same detections as vti_forward + vti_nms, bit for bit; the pairs must belong to the dev_pred they are passed with. */
int32_t vti_forward_scored(vti_ctx* ctx, const uint8_t* dev_input, int32_t B, int32_t swap_rb,
                           float* dev_pred, void* dev_proto, float* dev_anchor_best, void* stream);
int32_t vti_nms_scored(vti_ctx* ctx, const float* dev_pred, const float* dev_anchor_best, int32_t B, float conf, double iou,
                       int32_t max_det, int32_t agnostic, float* dev_dets, int32_t* dev_counts, void* stream);
/* U7 process_mask(upsample=True) + threshold.  Writes slots [0, min(offsets[B], capacity)) of dev_masks completely; slots
 * beyond that are left untouched (no whole-buffer memset: the cost follows the number of instances, not the capacity). */
int32_t vti_masks(vti_ctx* ctx, const float* dev_dets, const int32_t* dev_counts, const void* dev_proto,
                  int32_t B, int32_t max_det, int32_t mode, int32_t packing,
                  uint8_t* dev_masks, int32_t capacity, int32_t* dev_offsets, void* stream);
/* U8 scale_boxes + clip: letterboxed px -> frame px, writes f32 [B,max_det,4]. */
int32_t vti_scale_boxes(vti_ctx* ctx, const float* dev_dets, const int32_t* dev_counts, int32_t B,
                        int32_t max_det, int32_t H0, int32_t W0, float* dev_xyxy, void* stream);
/* All of the above on one stream (the scored pair of entry points, with the pairs in the workspace).  dev_input_scratch
 * (u8 [B,H,W,3]) may be NULL when H0xW0 == HxW. */
int32_t vti_predict(vti_ctx* ctx, const uint8_t* dev_frames, int32_t B, int32_t H0, int32_t W0,
                    int32_t swap_rb, float conf, double iou, int32_t max_det, int32_t agnostic,
                    int32_t mask_mode, int32_t packing, uint8_t* dev_input_scratch,
                    float* dev_pred, void* dev_proto, float* dev_dets, int32_t* dev_counts,
                    uint8_t* dev_masks, int32_t capacity, int32_t* dev_offsets, float* dev_xyxy,
                    void* stream);

/* ---- measurement.py's mask post-processing on device ------ */
/* A4 get_instance_mask_as_bitmap (measurement.py:70-86): u8 0/1 masks [n,H,W] ->
 * cv2.INTER_NEAREST resize to H0xW0, (>0); nonzero[i] = count of set pixels. */
int32_t vti_mask_to_frame(vti_ctx* ctx, const uint8_t* dev_masks, int32_t n, int32_t H, int32_t W,
                          int32_t H0, int32_t W0, uint8_t* dev_bitmaps, int32_t* dev_nonzero, void* stream);
/* A5+A6 _combine_masks + _fabric_lower_envelope (measurement.py:160-185): OR of the selected
 * bitmaps and, per column, the largest y with a set pixel (-1 if none). */
int32_t vti_union_envelope(vti_ctx* ctx, const uint8_t* dev_bitmaps, const int32_t* dev_select, int32_t nsel,
                           int32_t H0, int32_t W0, uint8_t* dev_union, int32_t* dev_envelope, void* stream);
/* A7 moments / extents (measurement.py:302-318): per bitmap i64 {m00, m10, m01, min_col, max_col}
 * (min/max = -1 when empty). */
int32_t vti_mask_stats(vti_ctx* ctx, const uint8_t* dev_bitmaps, int32_t n, int32_t H0, int32_t W0,
                       int64_t* dev_stats, void* stream);

/* The same reductions straight from the BIT-PACKED masks vti_masks writes (VTI_PACK_BITS, u8 [n,H,W/8]); the nearest resize of
 * A4 is folded into integer weights, so no [n,H0,W0] bitmap is materialised.  Results are identical to
 * vti_mask_to_frame followed by vti_mask_stats / vti_union_envelope.
 * A4+A7: i64 {m00, m10, m01, min_col, max_col} per instance (measurement.py:70-86,302-318).  dev_n_live (may be NULL) points
 * at the number of live slots of a fixed-capacity buffer (&dev_offsets[B] of vti_masks): slots at and beyond it are not read
 * and report the empty mask {0,0,0,-1,-1}. */
int32_t vti_mask_stats_bits(vti_ctx* ctx, const uint8_t* dev_masks_bits, int32_t n, const int32_t* dev_n_live,
                            int32_t H, int32_t W, int32_t H0, int32_t W0, int64_t* dev_stats, void* stream);
/* A4+A5+A6, batched: envelope i32 [B,W0] = per frame and frame column the largest row covered by any of the frame's instances of
 * class `cls` (cls < 0: every instance), -1 if none (measurement.py:70-86,160-185; the reference selects FABRIC_CLASS_ID,
 * config.py:70).  dev_offsets / dev_dets / capacity as written by vti_masks / vti_nms; H, W are the ctx's. */
int32_t vti_envelope_bits(vti_ctx* ctx, const uint8_t* dev_masks_bits, const int32_t* dev_offsets, const float* dev_dets,
                          int32_t B, int32_t max_det, int32_t capacity, int32_t cls, int32_t H0, int32_t W0,
                          int32_t* dev_envelope, void* stream);

/* ---- measurement geometry, float64 as the reference; ctx may be NULL ------------------------ */
/* pixel_to_world_using_camera_plane (measurement.py:50-65) for n points: cv2.undistortPoints (5 fixed-point iterations of the
 * k1,k2,p1,p2,k3 model) + ray / fabric-plane intersection with the plane of compute_camera_plane (measurement.py:44-48).
 * dev_uv f64 [n,2]; host_K f64[9] row-major, host_dist f64[5], host_R f64[9] row-major, host_t f64[3] in HOST memory
 * (camera_calibration.json / extrinsics.json); dev_xyz f64 [n,3] world metres; dev_valid i32 [n] (0 where the reference
 * returns None: |n . ray| < 1e-9). */
int32_t vti_pixels_to_world(vti_ctx* ctx, const double* dev_uv, int32_t n, const double* host_K, const double* host_dist,
                            const double* host_R, const double* host_t, double* dev_xyz, int32_t* dev_valid, void* stream);
/* kmeans_1d_two_clusters (measurement.py:88-113), batched: dev_values f64 [B,max_n] (counts[b] valid entries per row, max_n <=
 * 1024) -> dev_labels i32 [B,max_n] (0 beyond counts[b]), dev_centers f64 [B,2]; same iteration and exit rules, means summed in
 * numpy's pairwise order. */
int32_t vti_kmeans1d2(vti_ctx* ctx, const double* dev_values, const int32_t* dev_counts, int32_t B, int32_t max_n,
                      int32_t max_iters, int32_t* dev_labels, double* dev_centers, void* stream);

/* ---- per-layer access for parity tests ------------------------------------------- */
/* Copies the activation written by conv `i` of the last vti_forward into dev_out as
 * f32 NCHW [B,c2,h_out,w_out] (test hook; not on the hot path). */
int32_t vti_debug_conv_output(vti_ctx* ctx, int32_t i, int32_t B, float* dev_out, void* stream);

/* Runs ONE convolution of the engine's conv family on caller tensors (kernel unit tests and
 * micro-benchmarks; synchronous, allocates its own packed weights -- not on the hot path).
 * dev_in: T NHWC [B,H,W,in_ld] (or u8 [B,H,W,3] when c1==3, the stem conv); host_w: f32 OIHW
 * (kind 2: IOHW) in HOST memory; dev_out: T (or f32 if out_f32) NHWC with row pitch out_ld.
 * tile_h/tile_w/waves_n/nrep = 0 lets the planner choose; iters > 1 times iters-1 launches with
 * HIP events into *ms_out; cfg_out[5] receives {tile_h, tile_w, waves_n, nrep, lds_bytes}. */
int32_t vti_debug_conv2d(int32_t dtype, const void* dev_in, int32_t B, int32_t H, int32_t W, int32_t in_ld,
                         int32_t in_coff, int32_t c1, const float* host_w, const float* host_b, int32_t c2,
                         int32_t k, int32_t s, int32_t kind, const void* dev_res, int32_t res_ld, int32_t res_coff,
                         void* dev_out, int32_t out_ld, int32_t out_coff, int32_t out_f32, int32_t swap_rb,
                         int32_t tile_h, int32_t tile_w, int32_t waves_n, int32_t nrep, int32_t iters,
                         float* ms_out, int32_t* cfg_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VTI_H */
