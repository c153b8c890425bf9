// Wavefront-level pre/post-processing kernels for gfx950 (all HBM/L2-bound byte and fp32 work):
//   U1 LetterBox (OpenCV u8 INTER_LINEAR fixed point + 114 border)
//   U6 non_max_suppression (Ultralytics filter + torchvision.ops.nms greedy semantics)
//   U7 process_mask (coeff x proto, crop, bilinear upsample, threshold) with u8 or bit-packed output
//   U8 scale_boxes + clip
//   A4-A7 measurement.py's mask post-processing (nearest resize, OR, lower envelope, moments)
// Reference call sites: measurement.py:208-210 (predict), 70-86, 160-185, 300-330.
// Compiled with -ffp-contract=off so fp32 expressions round exactly as written (the CPU
// libraries they restate do not fuse multiply-adds in these formulas).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "vti_internal.h"

namespace vti {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================
// U1 LetterBox
// =====================================================================================
struct LinTap { int s0, s1, a0, a1; };

// OpenCV resize() coefficient for destination index d (imgproc/resize.cpp): fx computed in double,
// cast to float, floored; 11-bit fixed-point weights via cvRound (round half to even).
__device__ __forceinline__ LinTap lin_tap(int d, int ssize, double scale, bool horizontal) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    LinTap t;
    if (horizontal) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
        t.s0 = s; t.s1 = min(s + 1, ssize - 1);
    } else {
        t.s0 = min(max(s, 0), ssize - 1);
        t.s1 = min(max(s + 1, 0), ssize - 1);
    }
    t.a0 = (int)rintf((1.f - f) * 2048.f);
    t.a1 = (int)rintf(f * 2048.f);
    return t;
}

__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ frames, int B, int H0, int W0,
                                                        uint8_t* __restrict__ out, int H, int W, int new_h, int new_w,
                                                        int top, int left) {
    const long total = (long)B * H * W;
    const double scale_x = 1.0 / ((double)new_w / (double)W0);
    const double scale_y = 1.0 / ((double)new_h / (double)H0);
    const bool resize = (new_w != W0) || (new_h != H0);
    // OpenCV's resize() turns INTER_LINEAR into INTER_AREA when both scales are exactly 2 (hal::resize: is_area_fast &&
    // iscale_x == 2 && iscale_y == 2): the u8 result is the rounded 2x2 box mean, not the bilinear tap pair
    const bool area2 = W0 == 2 * new_w && H0 == 2 * new_h;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        long r = i / W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        uint8_t* o = out + i * 3;
        const int dy = y - top, dx = x - left;
        if (dy < 0 || dy >= new_h || dx < 0 || dx >= new_w) {
            o[0] = o[1] = o[2] = 114;
            continue;
        }
        const uint8_t* src = frames + (size_t)b * H0 * W0 * 3;
        if (!resize) {
            const uint8_t* s = src + ((size_t)dy * W0 + dx) * 3;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
            continue;
        }
        if (area2) {
            const uint8_t* s0 = src + ((size_t)(2 * dy) * W0 + 2 * dx) * 3;
            const uint8_t* s1 = s0 + (size_t)W0 * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = (uint8_t)((s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2);
            continue;
        }
        const LinTap tx = lin_tap(dx, W0, scale_x, true);
        const LinTap ty = lin_tap(dy, H0, scale_y, false);
        const uint8_t* r0 = src + (size_t)ty.s0 * W0 * 3;
        const uint8_t* r1 = src + (size_t)ty.s1 * W0 * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[tx.s0 * 3 + c] * tx.a0 + r0[tx.s1 * 3 + c] * tx.a1;
            const int h1 = r1[tx.s0 * 3 + c] * tx.a0 + r1[tx.s1 * 3 + c] * tx.a1;
            int v = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            o[c] = (uint8_t)v;
        }
    }
}

hipError_t launch_letterbox(const uint8_t* frames, int B, int H0, int W0, uint8_t* out, int H, int W, int new_h,
                            int new_w, int top, int left, hipStream_t st) {
    const long total = (long)B * H * W;
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(letterbox_kernel, dim3(grid), dim3(256), 0, st, frames, B, H0, W0, out, H, W, new_h, new_w, top, left);
    return hipGetLastError();
}

// =====================================================================================
// U6 non_max_suppression -- one 1024-thread workgroup per frame
// =====================================================================================
constexpr int NMS_THREADS = 1024;
constexpr int NMS_LDS_BOX = 2048;         // sorted boxes/areas/keep list live in LDS up to this many candidates
constexpr int NMS_LDS_KEYS = 2048;        // == NMS_LDS_BOX: beyond this many candidates everything lives in global scratch
constexpr float NMS_MAX_WH = 7680.0f;     // Ultralytics class offset
constexpr int NMS_MAX_NMS = 30000;        // Ultralytics max_nms: only the 30000 best-scoring candidates enter the greedy pass

static inline size_t nms_pow2(size_t n) { size_t p = 1; while (p < n) p <<= 1; return p; }
static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct NmsWsLayout { size_t keys, keys2, boxes, area, cls, keep, kid, kcls, per_frame; };
static NmsWsLayout nms_layout(int A) {
    NmsWsLayout l;
    size_t off = 0;
    l.keys = off; off += align256(nms_pow2((size_t)A) * 8);
    l.keys2 = off; off += align256(nms_pow2((size_t)A) * 8);
    l.boxes = off; off += align256((size_t)A * 16);
    l.area = off; off += align256((size_t)A * 4);
    l.cls = off; off += align256((size_t)A * 4);
    l.keep = off; off += align256((size_t)A * 4);
    l.kid = off; off += align256((size_t)A * 4);
    l.kcls = off; off += align256((size_t)A * 4);
    l.per_frame = off;
    return l;
}
size_t nms_workspace_bytes(int B, int A) { return nms_layout(A).per_frame * (size_t)B + align256((size_t)B * 4) + align256((size_t)B * A * 8); }
float* nms_workspace_best(void* ws, int B, int A) { return (float*)((char*)ws + nms_layout(A).per_frame * (size_t)B + align256((size_t)B * 4)); }

// Stage 1 (whole chip): best class score per anchor (first maximal index); anchors above `conf` are
// appended to the frame's key list.  Arrival order is arbitrary -- the sort below restores the order.
// With `best_out` the kernel only records the (score, class) pair of every anchor -- launch_anchor_best: the pairs a plan without
// fused class towers cannot write from its epilogue.
__global__ __launch_bounds__(256) void nms_scan_kernel(const float* __restrict__ pred, int A, int nc, int nm, float conf,
                                                       char* ws, NmsWsLayout L, int* __restrict__ ncand, float* __restrict__ best_out) {
    const int b = blockIdx.y;
    const int no = 4 + nc + nm;
    float best;
    int j = 0;
    int a;
    if (((no | nc) & 3) == 0) {
        // anchor-major rows: four lanes share an anchor and read its class scores as 16-byte pieces (a wave instruction
        // covers 16 rows x 64 contiguous bytes instead of 64 rows x 4 bytes); (value, first index) max across the quad
        a = blockIdx.x * 64 + (threadIdx.x >> 2);
        const int q = threadIdx.x & 3;
        best = -INFINITY; j = 0x7fffffff;
        if (a < A) {
            const float4* row = (const float4*)(pred + ((size_t)b * A + a) * no + 4);
            constexpr int MAXV = 8;                 // up to 128 classes: every load of the lane is issued before the first compare
            float4 v[MAXV];
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int i = q + 4 * u;
                v[u] = row[i < (nc >> 2) ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < MAXV; ++u) {
                const int i = q + 4 * u;
                if (i >= (nc >> 2)) continue;
                if (v[u].x > best) { best = v[u].x; j = 4 * i; }
                if (v[u].y > best) { best = v[u].y; j = 4 * i + 1; }
                if (v[u].z > best) { best = v[u].z; j = 4 * i + 2; }
                if (v[u].w > best) { best = v[u].w; j = 4 * i + 3; }
            }
            for (int i = q + 4 * MAXV; i < (nc >> 2); i += 4) {
                const float4 w = row[i];
                if (w.x > best) { best = w.x; j = 4 * i; }
                if (w.y > best) { best = w.y; j = 4 * i + 1; }
                if (w.z > best) { best = w.z; j = 4 * i + 2; }
                if (w.w > best) { best = w.w; j = 4 * i + 3; }
            }
        }
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
            const float ob = __shfl_xor(best, o);
            const int oj = __shfl_xor(j, o);
            if (ob > best || (ob == best && oj < j)) { best = ob; j = oj; }
        }
        if (q != 0 || a >= A) return;
    } else {
        a = blockIdx.x * 256 + threadIdx.x;
        if (a >= A) return;
        const float* P = pred + ((size_t)b * A + a) * no;      // pred is anchor-major: [B, A, 4+nc+nm]
        best = P[4];
        for (int c = 1; c < nc; ++c) {
            const float v = P[4 + c];
            if (v > best) { best = v; j = c; }
        }
    }
    if (best_out) {
        *(float2*)(best_out + ((size_t)b * A + a) * 2) = make_float2(best, (float)(j == 0x7fffffff ? 0 : j));
        return;
    }
    if (best > conf) {
        char* wsb = ws + (size_t)b * L.per_frame;
        const int idx = atomicAdd(&ncand[b], 1);
        // ascending key order == score descending, then anchor ascending (stable sort of the
        // anchor-ordered candidate list, as torchvision's stable descending sort).
        ((unsigned long long*)(wsb + L.keys))[idx] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned)a;
        ((int*)(wsb + L.cls))[a] = j;
    }
}

#ifdef VTI_STAMPS   // diagnostic build only: per-frame phase stamps of nms_kernel
__device__ unsigned long long g_nms_stamps[4096 * 8];
#define NMS_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (tid == 0 && b < 4096) {                                                                    \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
            g_nms_stamps[b * 8 + (i)] = t_;                                                            \
        }                                                                                              \
    } while (0)
#else
#define NMS_STAMP(i) do { } while (0)
#endif

// Stage 2: one workgroup per frame -- sort, greedy suppression, output rows.
// The body is instantiated per storage case (keys / boxes in LDS or in global scratch) and force-inlined
// at call sites where each pointer has ONE origin, so the compiler emits ds_* / global_* instructions
// instead of flat_* ones (a pointer chosen by `cond ? lds : global` is a flat pointer: every access then
// waits on both memory counters and costs several hundred cycles).
template <typename KeyP, typename BoxP, typename FltP, typename IntP>
__device__ __forceinline__ void nms_body(const float* __restrict__ P, int A, int nc, int nm, int n, double iou, int max_det,
                                         int agnostic, float* __restrict__ D, int* __restrict__ counts, int b,
                                         const unsigned long long* __restrict__ g_keys_in, KeyP keys, BoxP boxes, FltP area,
                                         IntP keep, IntP kid, IntP kcls, unsigned char* suppressed,
                                         const int* __restrict__ cls_of) {
    const int tid = threadIdx.x;
    // 2. bitonic sort of the keys
    int Pn = 1;
    while (Pn < n) Pn <<= 1;
    for (int i = tid; i < Pn; i += NMS_THREADS) keys[i] = i < n ? g_keys_in[i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= Pn; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < Pn; i += NMS_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
    NMS_STAMP(1);
    // Ultralytics: `x = x[x[:, 4].argsort(descending=True)[:max_nms]]` -- the keys are already in that order
    // (reachable only with more than 30000 anchors, e.g. 1280x1280 inputs: A = 33600)
    if (n > NMS_MAX_NMS) n = NMS_MAX_NMS;
    // 3. boxes in sorted order: xywh -> xyxy, + class offset, areas (all fp32 as torch computes them)
    for (int i = tid; i < n; i += NMS_THREADS) {
        const int a = (int)(keys[i] & 0xffffffffu);
        const float* Pa = P + (size_t)a * (4 + nc + nm);
        const float cx = Pa[0], cy = Pa[1], w = Pa[2], h = Pa[3];
        const float dw = w / 2.0f, dh = h / 2.0f;
        const float off = agnostic ? 0.0f : (float)cls_of[a] * NMS_MAX_WH;
        f32x4 bx;
        bx[0] = (cx - dw) + off; bx[1] = (cy - dh) + off; bx[2] = (cx + dw) + off; bx[3] = (cy + dh) + off;
        boxes[i] = bx;
        area[i] = (bx[2] - bx[0]) * (bx[3] - bx[1]);
        suppressed[i] = 0;
    }
    __syncthreads();
    NMS_STAMP(2);
    // 4. greedy suppression in score order, 64 candidates per round (same decisions as the sequential
    // loop, 3 barriers per 64 candidates instead of one per kept box):
    //   A. all threads: bit (i, j) of the 64x64 intra-block matrix = "i suppresses j" (j > i)
    //   B. wave 0: walk the block's still-alive candidates in order with 64-bit masks
    //   C. all threads: the block's kept boxes suppress every later candidate
    __shared__ unsigned long long s_rows[64];
    __shared__ int s_kept;
    auto iou_gt = [&](const f32x4& bi, float ai, const f32x4& bj, float aj) -> bool {
        const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
        const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
        const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
        const float inter = w * h;
        const float ovr = inter / (ai + aj - inter);
        return (double)ovr > iou;
    };
    int kept = 0;
    if (tid == 0) s_kept = 0;
    for (int base = 0; base < n && kept < max_det; base += 64) {
        const int cnt = min(64, n - base);
        if (tid < 64) s_rows[tid] = 0;
        __syncthreads();
        // A: thread -> (row i, 4 columns)
        for (int e = tid; e < 64 * 16; e += NMS_THREADS) {
            const int i = e >> 4, jg = (e & 15) * 4;
            if (i >= cnt || jg + 3 <= i) continue;
            const f32x4 bi = boxes[base + i];
            const float ai = area[base + i];
            unsigned long long m = 0;
            f32x4 bj4[4];
            float aj4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {          // the 4 operands first (clamped index), then the tests: no wait per pair
                const int jc = jg + k < cnt ? jg + k : cnt - 1;
                bj4[k] = boxes[base + jc]; aj4[k] = area[base + jc];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = jg + k;
                if (j > i && j < cnt && iou_gt(bi, ai, bj4[k], aj4[k])) m |= 1ull << j;
            }
            if (m) atomicOr(&s_rows[i], m);
        }
        __syncthreads();
        // B: wave 0, lane l <-> candidate base+l
        if (tid < 64) {
            const bool alive_l = tid < cnt && !suppressed[base + tid];
            unsigned long long alive = __ballot(alive_l);
            int k_ = kept;
            // lane l holds row l of the matrix; the walk fetches row i with v_readlane (i is wave-uniform) instead of a
            // dependent LDS read per kept box (~100 cycles each, 64 in a row: a third of the greedy phase on crowded frames)
            const unsigned long long myrow = s_rows[tid];
            const int row_lo = (int)(unsigned)myrow, row_hi = (int)(unsigned)(myrow >> 32);
            while (alive && k_ < max_det) {
                const int i = __ffsll((long long)alive) - 1;
                if (tid == 0) keep[k_] = base + i;
                ++k_;
                const unsigned long long ri = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(row_lo, i) |
                                              ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(row_hi, i) << 32);
                alive &= ~(ri | (1ull << i));
            }
            if (tid == 0) s_kept = k_;
        }
        __syncthreads();
        const int kept0 = kept;           // keep[kept0 .. kept) are this block's kept candidates, in order
        kept = s_kept;
        // C: later candidates vs this block's kept boxes, four at a time (operands of a group loaded together; a candidate
        // is suppressed iff ANY kept box of the block overlaps it, so the order of the tests does not matter).  With fewer
        // candidates left than threads, `parts` threads share a candidate and split the kept boxes between them.
        if (kept < max_det) {
            const int nk = kept - kept0;
            const int rest = n - (base + 64);
            int parts = 1;
            while (parts < 8 && rest * parts * 2 <= NMS_THREADS) parts <<= 1;
            const int part = tid & (parts - 1), lane_j = tid / parts, jstep = NMS_THREADS / parts;
            for (int j = base + 64 + lane_j; j < n; j += jstep) {
                if (suppressed[j]) continue;
                const f32x4 bj = boxes[j];
                const float aj = area[j];
                bool sup = false;
                for (int t = 4 * part; t < nk && !sup; t += 4 * parts) {
                    int ii[4];
                    f32x4 bk[4];
                    float ak[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) ii[u] = keep[kept0 + (t + u < nk ? t + u : nk - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { bk[u] = boxes[ii[u]]; ak[u] = area[ii[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) sup = sup | iou_gt(bk[u], ak[u], bj, aj);
                }
                if (sup) suppressed[j] = 1;
            }
        }
        __syncthreads();
    }
    NMS_STAMP(3);
    // 5. output rows [x1,y1,x2,y2,conf,cls,coeffs], zero the unused tail.  Anchor id and class of every
    // kept row are gathered once, then each wave streams whole rows (no per-element division, one level of
    // dependent global loads).
    for (int k = tid; k < kept; k += NMS_THREADS) {
        const int a = (int)(keys[keep[k]] & 0xffffffffu);
        kid[k] = a;
        kcls[k] = cls_of[a];
    }
    __syncthreads();
    const int row = 6 + nm;
    const unsigned row_magic = (unsigned)((0x100000000ull + (unsigned)row - 1) / (unsigned)row);
    // element e of the kept rows: (k, f) = (e / row, e % row).  Every element issues the same two UNCONDITIONAL loads (centre and
    // size of its axis for the box fields, the same word twice otherwise) and a thread issues all of its group before using
    // any: a load behind a data-dependent branch makes hipcc wait for it on the spot (one DRAM round trip per element: 39 k
    // cycles for 217 rows, the crowded frame that sets this kernel's duration).
    const int nel = kept * row, ntot = max_det * row;
    const int no_ = 4 + nc + nm;
    constexpr int OU = 8;
    for (int e0 = tid; e0 < nel; e0 += OU * NMS_THREADS) {
        float x1[OU], x2[OU];
        int fj[OU];
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int e = e0 + u * NMS_THREADS;
            const int ec = e < nel ? e : 0;
            const int k = (int)__umulhi((unsigned)ec, row_magic), f = ec - k * row;
            const int a = kid[k], j = kcls[k];
            const int off1 = f < 4 ? (f & 1) : f == 4 ? 4 + j : f == 5 ? 0 : 4 + nc + (f - 6);
            const int off2 = f < 4 ? 2 + (f & 1) : off1;
            const float* Pa = P + (size_t)a * no_;
            x1[u] = Pa[off1]; x2[u] = Pa[off2];
            fj[u] = (f << 16) | j;
        }
#pragma unroll
        for (int u = 0; u < OU; ++u) {
            const int e = e0 + u * NMS_THREADS;
            const int f = fj[u] >> 16, j = fj[u] & 0xffff;
            const float d_ = x2[u] / 2.0f;
            const float v = f < 2 ? x1[u] - d_ : f < 4 ? x1[u] + d_ : f == 5 ? (float)j : x1[u];
            if (e < nel) D[e] = v;
        }
    }
    for (int e = nel + tid; e < ntot; e += NMS_THREADS) D[e] = 0.f;
    __syncthreads();
    NMS_STAMP(4);
    if (tid == 0) counts[b] = kept;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ pred, int A, int nc, int nm,
                                                          float conf, double iou, int max_det, int agnostic,
                                                          float* __restrict__ dets, int* __restrict__ counts,
                                                          char* ws, NmsWsLayout L, int* __restrict__ ncand,
                                                          const float* __restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* lds_keys = (unsigned long long*)smem;             // NMS_LDS_KEYS entries
    f32x4* lds_boxes = (f32x4*)(smem + NMS_LDS_KEYS * 8);                 // NMS_LDS_BOX entries
    float* lds_area = (float*)(lds_boxes + NMS_LDS_BOX);
    int* lds_keep = (int*)(lds_area + NMS_LDS_BOX);                       // keep | kept anchor ids | kept classes
    unsigned char* suppressed = (unsigned char*)(lds_keep + 3 * NMS_LDS_BOX);  // A bytes

    const int b = blockIdx.x, tid = threadIdx.x;
    const int no = 4 + nc + nm;
    const float* P = pred + (size_t)b * no * A;
    char* wsb = ws + (size_t)b * L.per_frame;
    unsigned long long* g_keys = (unsigned long long*)(wsb + L.keys);
    const int* cls_of = (const int*)(wsb + L.cls);
    float* D = dets + (size_t)b * max_det * (6 + nm);
    __shared__ int s_cnt;
    if (best) {
        // Stage 1 here, from the per-anchor (max score, class) pairs written beside pred (8 bytes per anchor instead of nc scores):
        // anchors above `conf` go to the frame's key list.  Arrival order is arbitrary -- the sort restores the order.
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        const float2* bp = (const float2*)best + (size_t)b * A;
        int* cls_w = (int*)(wsb + L.cls);
        for (int a0 = 0; a0 < A; a0 += 4 * NMS_THREADS) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int a = a0 + tid + u * NMS_THREADS; v[u] = bp[a < A ? a : A - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int a = a0 + tid + u * NMS_THREADS;
                if (a < A && v[u].x > conf) {
                    const int idx = atomicAdd(&s_cnt, 1);
                    g_keys[idx] = ((unsigned long long)(~__float_as_uint(v[u].x)) << 32) | (unsigned)a;
                    cls_w[a] = (int)v[u].y;
                }
            }
        }
        __syncthreads();                          // the block's global writes are visible to the block behind the barrier
        if (tid == 0) ncand[b] = s_cnt;
    }
    const int n = best ? s_cnt : ncand[b];
    if (n == 0) {
        if (tid == 0) counts[b] = 0;
        for (int i = tid; i < max_det * (6 + nm); i += NMS_THREADS) D[i] = 0.f;
        return;
    }
    NMS_STAMP(0);
    if (n <= NMS_LDS_BOX && max_det <= NMS_LDS_BOX) {
        nms_body(P, A, nc, nm, n, iou, max_det, agnostic, D, counts, b, g_keys, lds_keys, lds_boxes, lds_area, lds_keep,
                 lds_keep + NMS_LDS_BOX, lds_keep + 2 * NMS_LDS_BOX, suppressed, cls_of);
    } else {
        // pathological candidate counts: everything but the suppressed flags in global scratch.  The keys are
        // sorted in a second scratch array so the unsorted input is not aliased.
        int* g_int = (int*)(wsb + L.keep);          // A ints: keep list; kept ids/classes reuse boxes' tail? no: own arrays below
        nms_body(P, A, nc, nm, n, iou, max_det, agnostic, D, counts, b, g_keys, (unsigned long long*)(wsb + L.keys2),
                 (f32x4*)(wsb + L.boxes), (float*)(wsb + L.area), g_int, (int*)(wsb + L.kid), (int*)(wsb + L.kcls), suppressed,
                 cls_of);
    }
}

hipError_t launch_anchor_best(const float* pred, int B, int A, int nc, int nm, float* best, hipStream_t st) {
    if (B == 0) return hipSuccess;
    const int apb = (((4 + nc + nm) | nc) & 3) == 0 ? 64 : 256;
    hipLaunchKernelGGL(nms_scan_kernel, dim3((A + apb - 1) / apb, B), dim3(256), 0, st, pred, A, nc, nm, 0.0f, (char*)nullptr, nms_layout(A),
                       (int*)nullptr, best);
    return hipGetLastError();
}

hipError_t launch_nms(const float* pred, const float* best, int B, int A, int nc, int nm, float conf, double iou, int max_det,
                      int agnostic, float* dets, int* counts, void* ws, hipStream_t st) {
    if (B == 0) return hipSuccess;
    const size_t lds = (size_t)NMS_LDS_KEYS * 8 + (size_t)NMS_LDS_BOX * 32 + (((size_t)A + 15) & ~(size_t)15);
    if (lds > 150 * 1024) return hipErrorInvalidValue;   // > ~39k anchors: unsupported
    static bool attr_set_dev[kMaxDevices] = {};
    bool& attr_set = attr_set_dev[current_device_slot()];
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const NmsWsLayout L = nms_layout(A);
    int* ncand = (int*)((char*)ws + L.per_frame * (size_t)B);
    if (!best) {
        hipError_t e = hipMemsetAsync(ncand, 0, sizeof(int) * (size_t)B, st);
        if (e != hipSuccess) return e;
        const int apb = (((4 + nc + nm) | nc) & 3) == 0 ? 64 : 256;        // anchors per block: quad-per-anchor / lane-per-anchor
        hipLaunchKernelGGL(nms_scan_kernel, dim3((A + apb - 1) / apb, B), dim3(256), 0, st, pred, A, nc, nm, conf, (char*)ws, L, ncand, (float*)nullptr);
    }
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(NMS_THREADS), lds, st, pred, A, nc, nm, conf, iou, max_det, agnostic,
                       dets, counts, (char*)ws, L, ncand, best);
#ifdef VTI_STAMPS
    {
        (void)hipStreamSynchronize(st);
        static int calls = 0;
        if (++calls == 3) {
            std::vector<unsigned long long> h(4096 * 8);
            std::vector<int> hc(B), hn(B);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_nms_stamps), h.size() * 8);
            (void)hipMemcpy(hc.data(), counts, B * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hn.data(), ncand, B * 4, hipMemcpyDeviceToHost);
            const char* nm_[4] = {"sort", "boxes", "greedy", "output"};
            int worst = 0;
            for (int b = 1; b < B; ++b) if (h[b * 8 + 4] - h[b * 8] > h[worst * 8 + 4] - h[worst * 8]) worst = b;
            fprintf(stderr, "[nms stamps] slowest frame %d: candidates %d kept %d\n", worst, hn[worst], hc[worst]);
            for (int i = 0; i < 4; ++i) {
                std::vector<long long> d;
                for (int b = 0; b < B; ++b) if (hn[b] > 0) d.push_back((long long)(h[b * 8 + i + 1] - h[b * 8 + i]));
                std::sort(d.begin(), d.end());
                fprintf(stderr, "[nms stamps] %-7s median %8lld  max %8lld  slowest-frame %8lld cycles\n", nm_[i],
                        d.empty() ? 0 : d[d.size() / 2], d.empty() ? 0 : d.back(), (long long)(h[worst * 8 + i + 1] - h[worst * 8 + i]));
            }
        }
    }
#endif
    return hipGetLastError();
}

// =====================================================================================
// U7 process_mask
// =====================================================================================
constexpr int MT = 64;          // output tile edge
constexpr int ML = MT / 4 + 3;  // low-res rows/cols needed by one tile at the fixed 1/4 scale (+ slack)

// Plan: exclusive prefix sums of the counts, then one work item per (instance slot, 64x64 output tile)
// whose low-res footprint can intersect the instance's crop box.  The live slots are zeroed first (mask_clear_kernel).
// item = slot * tiles + tile.  One workgroup; B and the detection counts are small.
__global__ __launch_bounds__(256) void mask_offsets_kernel(const int* __restrict__ counts, int B, int max_det,
                                                           int* __restrict__ offsets, int* __restrict__ nitems) {
    extern __shared__ int s_off[];      // B + 1 prefix sums
    for (int b = threadIdx.x; b < B; b += 256) {
        int c = counts[b];
        s_off[b + 1] = c < 0 ? 0 : (c > max_det ? max_det : c);
    }
    if (threadIdx.x == 0) { s_off[0] = 0; *nitems = 0; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int b = 0; b < B; ++b) s_off[b + 1] += s_off[b];
    __syncthreads();
    for (int b = threadIdx.x; b <= B; b += 256) offsets[b] = s_off[b];
}

// frame and instance of output slot `slot` (largest b with offsets[b] <= slot)
__device__ __forceinline__ void mask_slot_owner(const int* __restrict__ offsets, int B, int slot, int& b, int& inst) {
    int lo = 0, hi = B;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offsets[mid] <= slot) lo = mid; else hi = mid; }
    b = lo; inst = slot - offsets[lo];
}
// tiles [tx0, tx1] x [ty0, ty1] that get a work item: a tile can be non-zero only if one of its bilinear taps lies inside the box
// (in 1/4-res pixels); the box is expanded by 2 low-res pixels (= 8 output px) to be safe on every side.
__device__ __forceinline__ bool mask_tile_rect(const float* __restrict__ d, int H, int W, int& tx0, int& tx1, int& ty0, int& ty1) {
    const int tiles_x = (W + MT - 1) / MT, tiles_y = (H + MT - 1) / MT;
    const float x1 = d[0] - 8.f, y1 = d[1] - 8.f, x2 = d[2] + 8.f, y2 = d[3] + 8.f;
    tx0 = (int)floorf(x1 / MT); tx1 = (int)floorf(x2 / MT); ty0 = (int)floorf(y1 / MT); ty1 = (int)floorf(y2 / MT);
    tx0 = max(tx0, 0); ty0 = max(ty0, 0); tx1 = min(tx1, tiles_x - 1); ty1 = min(ty1, tiles_y - 1);
    return tx1 >= tx0 && ty1 >= ty0;
}

// Zeroes the live slots (everything a work item does not cover must read 0).  Slots at and beyond offsets[B] are not touched: a
// 4096-slot buffer holding three instances costs three slots of writes, not a memset of the whole buffer.  Whole slots, full
// lines: zeroing only the bytes outside each slot's tile rectangle was tried and is slower (partial-line writes: 109 MB took
// 50 us where this takes ~23 us for 181 MB).  One workgroup per slot.
// FUSED_OFFSETS (B <= 1024): the launch also does mask_offsets_kernel's job -- every workgroup sums the clamped counts for itself
// (that is the number of live slots), workgroup 0 writes the prefix sums; one launch less in front of the tile kernel.
template <bool FUSED_OFFSETS>
__global__ __launch_bounds__(256) void mask_clear_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int* __restrict__ nitems,
                                                         int B, int max_det, int slot_bytes, int capacity, uint8_t* __restrict__ masks) {
    const int slot = blockIdx.x;
    int total;
    if constexpr (FUSED_OFFSETS) {
        __shared__ int s_c[1025];
        __shared__ int s_tot;
        if (threadIdx.x == 0) s_tot = 0;
        __syncthreads();
        int part = 0;
        for (int b = threadIdx.x; b < B; b += 256) {
            const int c = counts[b], cc = c < 0 ? 0 : (c > max_det ? max_det : c);
            part += cc;
            if (slot == 0) s_c[b + 1] = cc;
        }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_tot, part);
        __syncthreads();
        total = s_tot;
        if (slot == 0) {
            if (threadIdx.x == 0) {
                s_c[0] = 0; *nitems = 0;
                for (int b = 0; b < B; ++b) s_c[b + 1] += s_c[b];
            }
            __syncthreads();
            for (int b = threadIdx.x; b <= B; b += 256) offsets[b] = s_c[b];
        }
    } else {
        total = offsets[B];
    }
    if (slot >= min(total, capacity)) return;
    uint4* b4 = (uint4*)(masks + (size_t)slot * slot_bytes);       // slot_bytes % 16 == 0 (checked by the launcher)
    const int nvec = slot_bytes >> 4;
    for (int v = threadIdx.x; v < nvec; v += 256) b4[v] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void mask_plan_kernel(const float* __restrict__ dets, const int* __restrict__ offsets, int B,
                                                        int max_det, int row, int H, int W, int capacity,
                                                        int2* __restrict__ items, int* __restrict__ nitems) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    const int total = min(offsets[B], capacity);
    if (slot >= total) return;
    int lo, inst;
    mask_slot_owner(offsets, B, slot, lo, inst);
    const float* d = dets + ((size_t)lo * max_det + inst) * row;
    int tx0, tx1, ty0, ty1;
    if (!mask_tile_rect(d, H, W, tx0, tx1, ty0, ty1)) return;
    const int cnt = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
    int base = atomicAdd(nitems, cnt);
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) items[base++] = make_int2((slot << 12) | (ty << 6) | tx, (int)(((unsigned)lo << 16) | (unsigned)inst));   // tiles < 64 per side
}

// Bilinear 4x upsample + threshold of ONE instance's low-res tile `lw` (pitch ML + 1) for the 16 output rows of a wave; a LANE IS A
// COLUMN (taps c0 / c1, weight lx1).  F.interpolate(bilinear, align_corners=False) at the fixed 1/4 scale: src = 0.25*(dst+0.5)-0.5
// clamped at 0, so for dst >= 2 the source index is (dst-2)>>2 with fraction {0.125,0.375,0.625,0.875}[(dst-2)&3] and for dst < 2 it
// is index 0, fraction 0 (all exact in fp32).  The horizontal blend is done once per low-res row (6 per wave), a pixel then costs one
// vertical blend, a + w (b - a) as one FMA, and the 64-lane compare mask of a row IS that row's 8 output bytes, moved to lane r's
// registers with v_writelane (the builtin, so the compiler fills the VALU->SGPR->writelane wait states with the next rows' blends).
// `ro[j]`: offset of low-res row (first row >> 2) - 1 + j (clamped to the image) inside `lw`; `top`: the wave starts at image row 0.
template <bool BYTES>
__device__ __forceinline__ void mask_rows16(const float* lw, const int (&ro)[6], int c0, int c1, float lx1, bool top, float thr,
                                            unsigned& lo, unsigned& hi, uint8_t* __restrict__ out_bytes, int W, int H, int yw, int xg) {
    float hb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float a = lw[ro[j] + c0], b = lw[ro[j] + c1];
        hb[j] = __builtin_fmaf(b - a, lx1, a);
    }
    unsigned l_ = 0u, h_ = 0u;
    constexpr float FR[4] = {0.125f, 0.375f, 0.625f, 0.875f};
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += 8) {
        unsigned ml[8], mh[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = r0 + q, j = (r + 2) >> 2;
            float w1 = FR[(r + 2) & 3];
            if (r < 2 && top) w1 = 0.0f;                                  // rows 0, 1 of the image clamp to source row 0
            const float v = __builtin_fmaf(hb[j + 1] - hb[j], w1, hb[j]);
            if constexpr (BYTES) {
                if (xg < W && yw + r < H) out_bytes[(size_t)(yw + r) * W + xg] = v > thr ? (uint8_t)1 : (uint8_t)0;
            } else {
                const unsigned long long m = __builtin_amdgcn_ballot_w64(v > thr);
                ml[q] = (unsigned)m; mh[q] = (unsigned)(m >> 32);         // wave-uniform (SGPRs)
            }
        }
        if constexpr (!BYTES) {
            // the compares write SGPRs that v_writelane reads: the hardware needs wait states in between and the compiler's hazard
            // recogniser does not look inside inline asm (seen as wrong bits in exactly the rows whose compare sat next to its
            // writelane), so eight rows' masks are moved by one block behind one s_nop
            if (r0 == 0)
                asm("s_nop 4\n\tv_writelane_b32 %0, %2, 0\n\tv_writelane_b32 %1, %3, 0\n\tv_writelane_b32 %0, %4, 1\n\tv_writelane_b32 %1, %5, 1\n\t"
                    "v_writelane_b32 %0, %6, 2\n\tv_writelane_b32 %1, %7, 2\n\tv_writelane_b32 %0, %8, 3\n\tv_writelane_b32 %1, %9, 3\n\t"
                    "v_writelane_b32 %0, %10, 4\n\tv_writelane_b32 %1, %11, 4\n\tv_writelane_b32 %0, %12, 5\n\tv_writelane_b32 %1, %13, 5\n\t"
                    "v_writelane_b32 %0, %14, 6\n\tv_writelane_b32 %1, %15, 6\n\tv_writelane_b32 %0, %16, 7\n\tv_writelane_b32 %1, %17, 7"
                    : "+v"(l_), "+v"(h_)
                    : "s"(ml[0]), "s"(mh[0]), "s"(ml[1]), "s"(mh[1]), "s"(ml[2]), "s"(mh[2]), "s"(ml[3]), "s"(mh[3]),
                      "s"(ml[4]), "s"(mh[4]), "s"(ml[5]), "s"(mh[5]), "s"(ml[6]), "s"(mh[6]), "s"(ml[7]), "s"(mh[7]));
            else
                asm("s_nop 4\n\tv_writelane_b32 %0, %2, 8\n\tv_writelane_b32 %1, %3, 8\n\tv_writelane_b32 %0, %4, 9\n\tv_writelane_b32 %1, %5, 9\n\t"
                    "v_writelane_b32 %0, %6, 10\n\tv_writelane_b32 %1, %7, 10\n\tv_writelane_b32 %0, %8, 11\n\tv_writelane_b32 %1, %9, 11\n\t"
                    "v_writelane_b32 %0, %10, 12\n\tv_writelane_b32 %1, %11, 12\n\tv_writelane_b32 %0, %12, 13\n\tv_writelane_b32 %1, %13, 13\n\t"
                    "v_writelane_b32 %0, %14, 14\n\tv_writelane_b32 %1, %15, 14\n\tv_writelane_b32 %0, %16, 15\n\tv_writelane_b32 %1, %17, 15"
                    : "+v"(l_), "+v"(h_)
                    : "s"(ml[0]), "s"(mh[0]), "s"(ml[1]), "s"(mh[1]), "s"(ml[2]), "s"(mh[2]), "s"(ml[3]), "s"(mh[3]),
                      "s"(ml[4]), "s"(mh[4]), "s"(ml[5]), "s"(mh[5]), "s"(ml[6]), "s"(mh[6]), "s"(ml[7]), "s"(mh[7]));
        }
    }
    lo = l_; hi = h_;
}

// Generic coefficient count (nm != 32): one work item per (instance slot, tile), vector-ALU dots.  The nm == 32 layout of every
// YOLOv8-seg checkpoint runs masks_group_kernel below; this kernel is the reference-shaped fallback and shares its upsample.
template <typename T>
__global__ __launch_bounds__(256, 6) void masks_kernel(const float* __restrict__ dets, const int* __restrict__ offsets,
                                                    const T* __restrict__ proto, int B, int max_det, int nm, int Hp,
                                                    int Wp, int H, int W, int mode, int packing,
                                                    uint8_t* __restrict__ masks, const int2* __restrict__ items,
                                                    const int* __restrict__ nitems) {
    __shared__ float coef[64];
    __shared__ float low[ML][ML + 1];
    const int tid = threadIdx.x;
    const int n = *nitems;
    // XCD x (= blockIdx & 7: workgroups are dealt round-robin to the 8 XCDs) walks its own contiguous eighth of the work list,
    // round-robin over its workgroups: at any time an XCD works on ~1 frame, whose prototypes are fetched into ONE L2 instead
    // of eight, and the 8-byte row pieces that share a 128-byte output line are written through one L2.  (A contiguous run
    // per workgroup instead spreads the chip over all 64 frames at once: 333 -> 363 us.)
    const int nx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int per = (n + 7) >> 3;
    const int it0 = xcd * per + jx, it1 = min(n, (xcd + 1) * per);
    int2 itm_next = it0 < it1 ? items[it0] : make_int2(0, 0);
    for (int it = it0; it < it1; it += nx) {
        const int2 itm = itm_next;
        {   // the next item's record is fetched a whole item ahead (clamped index: nothing consumes it in this iteration)
            const int itn = it + nx;
            itm_next = items[itn < it1 ? itn : it];
        }
        const int item = __builtin_amdgcn_readfirstlane(itm.x);          // block-uniform: keep it scalar
        const int code = __builtin_amdgcn_readfirstlane(itm.y);
        const int tx = item & 63, ty = (item >> 6) & 63, slot = item >> 12;      // packed by mask_plan_kernel: no divisions here
        const int b = (code >> 16) & 0x7fff, inst = code & 0xffff;
        __syncthreads();                  // previous iteration is done with the shared tiles
        const int y0 = ty * MT, x0 = tx * MT;
        const int row = 6 + nm;
        const float* d = dets + ((size_t)b * max_det + inst) * row;
        if (tid < nm) coef[tid] = d[6 + tid];

        // torch: area_pixel_compute_scale<float>(in, out) = (float)in / out ; src = scale*(dst+0.5)-0.5, clamped at 0
        const float sh = (float)Hp / (float)H, sw = (float)Wp / (float)W;
        float fy0 = sh * ((float)y0 + 0.5f) - 0.5f; fy0 = fy0 < 0.f ? 0.f : fy0;
        float fx0 = sw * ((float)x0 + 0.5f) - 0.5f; fx0 = fx0 < 0.f ? 0.f : fx0;
        const int ly0 = (int)fy0, lx0 = (int)fx0;     // first low-res row/col this tile touches
        // crop box in prototype pixels: boxes * (mw/iw) etc. in fp32 (torch multiplies an f32 tensor by a python float)
        const float wr = (float)((double)Wp / (double)W), hr = (float)((double)Hp / (double)H);
        const float bx1 = d[0] * wr, by1 = d[1] * hr, bx2 = d[2] * wr, by2 = d[3] * hr;
        __syncthreads();
        for (int e = tid; e < ML * ML; e += 256) {
            const int r = e / ML, c = e - r * ML;
            const int py = ly0 + r, px = lx0 + c;
            float v = 0.f;
            if (py < Hp && px < Wp) {
                const float fr = (float)py, fc = (float)px;
                if (fc >= bx1 && fc < bx2 && fr >= by1 && fr < by2) {
                    const T* pp = proto + ((size_t)(b * Hp + py) * Wp + px) * nm;
                    float acc = 0.f;
                    constexpr int PV = 16 / sizeof(T);          // 16-B pieces
                    typedef T pvec __attribute__((ext_vector_type(PV)));
                    // summation order is free here (torch's sgemm sums in its own order anyway)
                    if (nm % PV == 0) {
                        for (int k = 0; k < nm; k += PV) {
                            const pvec v = *(const pvec*)(pp + k);
#pragma unroll
                            for (int j = 0; j < PV; ++j) acc = __builtin_fmaf(coef[k + j], (float)v[j], acc);
                        }
                    } else {
                        for (int k = 0; k < nm; ++k) acc = __builtin_fmaf(coef[k], (float)pp[k], acc);
                    }
                    v = mode == VTI_MASK_SIGMOID ? 1.0f / (1.0f + expf(-acc)) : acc;
                }
            }
            low[r][c] = v;
        }
        __syncthreads();

        const float thr = mode == VTI_MASK_SIGMOID ? 0.5f : 0.0f;
        // a wave owns 16 rows of the tile (mask_rows16)
        const int wv = tid >> 6, ln = tid & 63;
        const int yw = __builtin_amdgcn_readfirstlane(y0 + 16 * wv);      // first row of this wave
        if (yw >= H) continue;
        const int xg = x0 + ln;
        int k0 = xg >= 2 ? (xg - 2) >> 2 : 0;
        k0 = k0 < Wp - 1 ? k0 : Wp - 1;                                   // columns past W (partial tile) are masked below
        const int k1 = k0 + (k0 < Wp - 1 ? 1 : 0);
        const float lx1 = xg >= 2 ? 0.125f + 0.25f * (float)((xg - 2) & 3) : 0.0f;
        const int rb = (yw >> 2) - 1;                                     // low-res row of (yw - 2) >> 2
        int ro[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            int rr = rb + j;
            rr = rr < 0 ? 0 : (rr > Hp - 1 ? Hp - 1 : rr);                // == min(iy + 1, Hp - 1) for the lower tap
            ro[j] = (rr - ly0) * (ML + 1);
        }
        const int valid = W - x0;                                         // 32 or >= 64 (W is a multiple of 32)
        const unsigned long long cmask = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
        unsigned lo = 0u, hi = 0u;
        if (packing == VTI_PACK_U8) mask_rows16<true>(&low[0][0], ro, k0 - lx0, k1 - lx0, lx1, yw == 0, thr, lo, hi, masks + (size_t)slot * H * W, W, H, yw, xg);
        else mask_rows16<false>(&low[0][0], ro, k0 - lx0, k1 - lx0, lx1, yw == 0, thr, lo, hi, nullptr, W, H, yw, xg);
        lo &= (unsigned)cmask; hi &= (unsigned)(cmask >> 32);            // columns past W (partial tile)
        if (packing != VTI_PACK_U8 && ln < 16 && yw + ln < H) {           // lane r stores row r: 8 (4) bytes of bits
            const int wb = W >> 3;
            unsigned* o = (unsigned*)(masks + ((size_t)slot * H + yw + ln) * wb + (x0 >> 3));     // 4-byte aligned (W % 32 == 0)
            o[0] = lo;
            if (valid >= 64) o[1] = hi;
        }
    }
}

// The same upsample with a LANE AS A ROW: one wave produces the whole 64 x 64 tile of one instance, every lane the 64 bits of its
// row -- no transposition of compare masks (the 32 v_writelane per 16 rows of mask_rows16 go away) and 3 vector instructions per
// 64 pixels: the blend, the compare, and v_addc (bits = 2 bits + carry-in) that shifts the compare's own bit into the lane's word;
// pixels are taken from 31 down to 0 so that pixel 0 ends in bit 0.  The vertical blend comes first here (18 low-res columns per
// lane), the horizontal one second with compile-time taps and weights: bilinear interpolation is separable, so this is the same
// number as torch's horizontal-first order up to the last rounding (a pixel can differ only where the blended logit is within
// ~1e-7 relative of the threshold).  `lrow`: &low[u][row of (y - 2) >> 2][first column]; the next low-res row is LP floats on.  The
// caller's low-res tile replicates the image's last row / column beyond the border, so no tap needs clamping.  `first`: the tile
// starts at image column 0 (pixels 0, 1 take the first column with weight 0).
constexpr int MLP = ML + 2;           // low-res row pitch in the grouped kernel: odd, so the 17 rows a wave reads fall in 17 banks
__device__ __forceinline__ void mask_tile_rows64(const float* lrow, float wy, bool first, float thr, unsigned& lo, unsigned& hi) {
    float vv[18];
#pragma unroll
    for (int j = 0; j < 18; ++j) {
        const float a = lrow[j], b = lrow[j + MLP];
        vv[j] = __builtin_fmaf(b - a, wy, a);
    }
    unsigned w[2] = {0u, 0u};
    if (thr == 0.0f) {
        // logits against 0 (wave-uniform): the bit is the SIGN of -v, shifted in by v_alignbit -- two instructions per 64 pixels.
        // -v = fma(vv[j] - vv[j+1], wx, -vv[j]) is the exact negation of the blend below (round-to-nearest is symmetric), and it is
        // never -0: a zero sum of non-zero terms is +0, cropped taps are +0 and x - x is +0, so v = 0 gives bit 0 as v > 0 does.
        // (A NaN logit -- non-finite coefficients -- has no defined sign: the low-res tile is written without NaNs.)
#pragma unroll
        for (int j = 0; j < 18; ++j) vv[j] = -vv[j];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int k = 31; k >= 0; --k) {
                const int px = 32 * hf + k;
                const int jj = ((px - 2) >> 2) + 1;                       // 0 .. 16
                const float wx = 0.125f + 0.25f * (float)((px + 2) & 3);
                float u = __builtin_fmaf(vv[jj + 1] - vv[jj], wx, vv[jj]);
                if (px < 2) u = first ? vv[1] + 0.0f : u;                 // (-0) + (+0) = +0: a cropped tap must not read as negative
                w[hf] = __builtin_amdgcn_alignbit(w[hf], __builtin_bit_cast(unsigned, u), 31);     // (w << 1) | sign(u)
            }
        }
    } else {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int k = 31; k >= 0; --k) {
                const int px = 32 * hf + k;
                const int jj = ((px - 2) >> 2) + 1;                       // 0 .. 16
                const float wx = 0.125f + 0.25f * (float)((px + 2) & 3);
                float v = __builtin_fmaf(vv[jj + 1] - vv[jj], wx, vv[jj]);
                if (px < 2) v = first ? vv[1] : v;
                asm("v_cmp_gt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w[hf]) : "v"(v), "v"(thr) : "vcc");
            }
        }
    }
    lo = w[0]; hi = w[1];
}

#ifdef VTI_STAMPS   // diagnostic build only: cycles per phase of masks_group_kernel, summed over wave 0 of every block
__device__ unsigned long long g_mask_acc[8];
#define MASK_T(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
                       macc[i] += t_ - mprev; mprev = t_; } while (0)
#else
#define MASK_T(i) do { } while (0)
#endif

// ---- nm == 32 (every YOLOv8-seg checkpoint): instances GROUPED per (frame, 64x64 output tile) ----
// The per-(instance, tile) kernel above fetches the tile's 19 x 19 x 32 prototype footprint once per instance that touches the
// tile (~23 KB per item: its time was the CU's L1, one access per cycle) and runs the 361 dots on the vector ALU.  Here a work item
// is a (frame, tile) pair -- a static, frame-major list, no plan kernel, no atomics -- and `coeff x proto` is what it is in the
// reference: a small GEMM (ops.process_mask: masks_in @ protos.view(c, -1)).  A workgroup
//   1. lists the frame's instances whose crop box can reach the tile (same rectangle rule as mask_tile_rect; list order =
//      detection order, built with ballots),
//   2. loads the prototype footprint ONCE, straight into MFMA operand registers (point = MFMA column; a wave owns 6 of the 23
//      16-point column tiles; 4 lanes cover a point's 64 contiguous bytes),
//   3. per group of 16 listed instances: coefficient rows -> the other MFMA operand (fp16 prototypes: c = s * (h + l) with
//      h = half(c/s), l = half(c/s - h), s a power of two that is 1 unless |c| >= 3e4 -- two 16x16x32 MFMAs per column tile,
//      products exact, fp32 accumulation, as the split v_dot2 path it replaces; fp32 prototypes: eight exact 16x16x4 steps),
//      crop, low-res logits -> LDS, then the bilinear 4x upsample + threshold + bit packing of the kernel above, instance by instance.
// HBM/L2 traffic: prototypes x (19/16)^2 instead of x (instances per tile); the dots leave the vector ALU.
constexpr int MG = 16;                          // instances per MFMA group
constexpr int MPTS = ML * ML;                   // 361 low-res points per tile
constexpr int MTILES = (MPTS + 15) / 16;        // 23 column tiles
constexpr int MJ = (MTILES + 3) / 4;            // column tiles per wave
constexpr int MROUND = 64;                      // instances examined per list round (one per lane of wave 0)

template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 4 : 3) void masks_group_kernel(const float* __restrict__ dets, const int* __restrict__ offsets,
                                                          const T* __restrict__ proto, int B, int max_det, int Hp, int Wp,
                                                          int H, int W, int mode, int packing, uint8_t* __restrict__ masks,
                                                          int capacity) {
    constexpr bool F16 = sizeof(T) == 2;
    constexpr int ROW = 6 + 32;
    typedef typename std::conditional<F16, half8, f32x4>::type opv;      // one MFMA operand register set
    constexpr int NOP = F16 ? 1 : 2;                                       // fp32: channels 4g..4g+3 and 16+4g..16+4g+3 of lane group g
    __shared__ uint4 s_co[MROUND][8];          // listed instances' coefficient rows, ready as MFMA operand pieces (128 B each)
    __shared__ float4 s_box[MROUND];           // ... crop boxes in prototype pixels
    __shared__ float s_scale[MROUND];
    __shared__ unsigned short s_list[MROUND];  // ... index inside the round
    __shared__ int s_n;
    __shared__ float low[MG][ML][MLP];          // behind the other arrays: the lane-as-row reads start one float before a row
    const int tid = threadIdx.x, ln = tid & 63, wv = tid >> 6;
    const int li = ln & 15, lg = ln >> 4;
    const int tiles_x = (W + MT - 1) / MT, tiles_y = (H + MT - 1) / MT, tiles = tiles_x * tiles_y;
    const int n = B * tiles;
    // XCD x (= blockIdx & 7) walks its own contiguous eighth of the frame-major list, round-robin over its workgroups: at any
    // time an XCD works on about one frame, whose prototypes and dets rows sit in ONE L2
    const int nx = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int per = (n + 7) >> 3;
    const int it1 = min(n, (xcd + 1) * per);
    const float wr = (float)((double)Wp / (double)W), hr = (float)((double)Hp / (double)H);       // as in masks_kernel
    const float sh = (float)Hp / (float)H, sw = (float)Wp / (float)W;
    const float thr = mode == VTI_MASK_SIGMOID ? 0.5f : 0.0f;
    // point of this lane in column tile j of this wave: low-res row r (bits 5..9) and column c (0..4) inside the 19 x 19 footprint;
    // points past 361 (last tile) are (18, 19): the pad column, read for nothing and written to a pad element
    int prc[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        const int m = wv + 4 * j, pt = 16 * m + li;
        const bool live = m < MTILES && pt < MPTS;
        const int r = live ? pt / ML : ML - 1, c = live ? pt - r * ML : ML;
        prc[j] = (r << 5) | c;
    }
#ifdef VTI_STAMPS
    unsigned long long macc[6] = {0, 0, 0, 0, 0, 0}, mprev, mitems = 0, mpairs = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(mprev)::"memory");
#endif
    for (int it = xcd * per + jx; it < it1; it += nx) {
        const int b = it / tiles, t = it - b * tiles;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * MT, x0 = tx * MT;
        const int off_b = offsets[b];
        const int n_b = min(offsets[b + 1], capacity) - off_b;            // instances of this frame that own a slot
        if (n_b <= 0) continue;                                            // block-uniform
        float fy0 = sh * ((float)y0 + 0.5f) - 0.5f; fy0 = fy0 < 0.f ? 0.f : fy0;
        float fx0 = sw * ((float)x0 + 0.5f) - 0.5f; fx0 = fx0 < 0.f ? 0.f : fx0;
        const int ly0 = (int)fy0, lx0 = (int)fx0;
        // ---- prototype footprint -> MFMA operand registers (issued before the list is known: the latency runs under the list build)
        opv pb[MJ][NOP];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            int q = prc[j];
            asm volatile("" : "+v"(q));                                    // unpack here, every time: hoisted copies cost registers (spills)
            const int py = ly0 + (q >> 5), px = lx0 + (q & 31);
            const int pyc = py < Hp ? py : Hp - 1, pxc = px < Wp ? px : Wp - 1;
            const T* pp = proto + ((size_t)(b * Hp + pyc) * Wp + pxc) * 32;
            if constexpr (F16) pb[j][0] = *(const half8*)(pp + 8 * lg);
            else { pb[j][0] = *(const f32x4*)(pp + 4 * lg); pb[j][1] = *(const f32x4*)(pp + 16 + 4 * lg); }
        }
        MASK_T(0);                                                         // item decoded, prototype loads issued
        const int wvy = __builtin_amdgcn_readfirstlane(y0 + 16 * wv);     // first output row of this wave's quarter of the tile
        const int xg = x0 + ln;
        int kc0 = xg >= 2 ? (xg - 2) >> 2 : 0;
        kc0 = kc0 < Wp - 1 ? kc0 : Wp - 1;
        const int kc1 = kc0 + (kc0 < Wp - 1 ? 1 : 0) - lx0;
        kc0 -= lx0;
        const float lx1 = xg >= 2 ? 0.125f + 0.25f * (float)((xg - 2) & 3) : 0.0f;
        const int rbase = (wvy >> 2) - 1;
        int ro[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            int rr = rbase + j;
            rr = rr < 0 ? 0 : (rr > Hp - 1 ? Hp - 1 : rr);
            ro[j] = (rr - ly0) * MLP;
        }
        const int valid = W - x0;
        const unsigned long long cmask = valid >= 64 ? ~0ull : ((1ull << valid) - 1ull);
        // lane as a row (bit packing): output row y0 + ln, its upper low-res row and vertical weight; column tap 0 of the tile
        const int yr = y0 + ln;
        const int lrow_off = ((yr >= 2 ? (yr - 2) >> 2 : 0) - ly0) * MLP + (tx > 0 ? 0 : -1);
        const float wy = yr >= 2 ? 0.125f + 0.25f * (float)((yr - 2) & 3) : 0.0f;
        for (int i0 = 0; i0 < n_b; i0 += MROUND) {
            // ---- wave 0: the round's instances, one per lane.  The whole 152-byte dets row is loaded before the box is tested (one
            // memory round trip instead of box -> list -> coefficients); rows of instances that reach the tile go to LDS in detection
            // order (ballot), already in MFMA operand form.
            __syncthreads();                                               // previous round / item is done with the shared arrays
            if (wv == 0) {
                const int i = i0 + ln;
                const float* d = dets + ((size_t)b * max_det + (i < n_b ? i : n_b - 1)) * ROW;
                float2 raw[ROW / 2];                                       // rows are 152 bytes: 8-byte aligned
#pragma unroll
                for (int k = 0; k < ROW / 2; ++k) raw[k] = *(const float2*)(d + 2 * k);
                const float dd[4] = {raw[0].x, raw[0].y, raw[1].x, raw[1].y};
                int tx0, tx1, ty0, ty1;
                const bool hit = i < n_b && mask_tile_rect(dd, H, W, tx0, tx1, ty0, ty1) && tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1;
                const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
                if (ln == 0) s_n = __builtin_popcountll(bal);
                if (hit) {
                    const int pos = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                    s_list[pos] = (unsigned short)ln;
                    s_box[pos] = make_float4(dd[0] * wr, dd[1] * hr, dd[2] * wr, dd[3] * hr);     // crop box in prototype pixels (fp32, as torch)
                    float cf[32];
#pragma unroll
                    for (int k = 0; k < 16; ++k) { cf[2 * k] = raw[3 + k].x; cf[2 * k + 1] = raw[3 + k].y; }
                    float scale = 1.0f;
                    if constexpr (F16) {
                        float mx = 0.f;
#pragma unroll
                        for (int k = 0; k < 32; ++k) mx = fmaxf(mx, fabsf(cf[k]));
                        float inv = 1.0f;
                        if (!(mx < 3.0e4f)) {                              // out of half range: c = 2^e * c', |c'| < 2^14
                            const int e = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127 - 13;
                            const int ec = e < -100 ? -100 : (e > 100 ? 100 : e);
                            scale = __builtin_bit_cast(float, (unsigned)(ec + 127) << 23);
                            inv = __builtin_bit_cast(float, (unsigned)(127 - ec) << 23);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {                      // piece q = channels 8q .. 8q+7: high halves, then the residuals
                            half8 hh, hl;
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float c = cf[8 * q + k] * inv;
                                const half_t h = (half_t)c;
                                hh[k] = h;
                                hl[k] = (half_t)(c - (float)h);
                            }
                            s_co[pos][q] = __builtin_bit_cast(uint4, hh);
                            s_co[pos][4 + q] = __builtin_bit_cast(uint4, hl);
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q)                        // piece q = channels 4q .. 4q+3
                            s_co[pos][q] = __builtin_bit_cast(uint4, (f32x4){cf[4 * q], cf[4 * q + 1], cf[4 * q + 2], cf[4 * q + 3]});
                    }
                    s_scale[pos] = scale;
                }
            }
            __syncthreads();
            const int nlist = s_n;
            MASK_T(1);                                                     // list built
#ifdef VTI_STAMPS
            ++mitems; mpairs += nlist;
#endif
            for (int g0 = 0; g0 < nlist; g0 += MG) {
                const int ng = min(MG, nlist - g0);
                // ---- coefficient operand: lane (li, lg) holds instance g0 + li's channels of lane group lg (rows >= ng: stale, unused)
                opv ca[2];
                ca[0] = __builtin_bit_cast(opv, s_co[g0 + li][lg]);
                ca[1] = __builtin_bit_cast(opv, s_co[g0 + li][4 + lg]);
                // rows 4 lg + r of the MFMA result are instances g0 + 4 lg + r: their boxes and scales
                float4 rb[4];
                float rs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int u = 4 * lg + r;
                    rb[r] = s_box[g0 + u];
                    rs[r] = s_scale[g0 + u];
                }
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (F16) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca[0], pb[j][0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ca[1], pb[j][0], acc, 0, 0, 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[q][k], pb[j][q][k], acc, 0, 0, 0);
                    }
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = F16 ? acc[r] * rs[r] : acc[r];
                    if (mode == VTI_MASK_SIGMOID) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + expf(-v[r]));
                    }
                    // branch-free stores: rows >= ng of `low` are never read, points that do not exist write their row's pad element
                    int q = prc[j];
                    asm volatile("" : "+v"(q));
                    // points beyond the image's last row / column take that row's / column's value (they were loaded from the clamped
                    // address): the border replication of F.interpolate's taps, done once here instead of in every tap
                    const int pr = q >> 5, pc = q & 31, py = min(ly0 + pr, Hp - 1), px = min(lx0 + pc, Wp - 1);
                    const float pfc = (float)px, pfr = (float)py;
                    const int pla = pr * MLP + pc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool inb = pfc >= rb[r].x && pfc < rb[r].z && pfr >= rb[r].y && pfr < rb[r].w;
                        (&low[4 * lg + r][0][0])[pla] = (inb && v[r] == v[r]) ? v[r] : 0.f;       // cropped, or NaN (non-finite coefficients): +0
                    }
                }
                MASK_T(2);                                                 // coefficients loaded, MFMAs, low-res tiles written
                __syncthreads();
                MASK_T(3);
                // ---- upsample + threshold + store
                if (packing != VTI_PACK_U8) {
                    // bits: a wave takes every fourth listed instance and produces its whole tile, a lane is a row (mask_tile_rows64)
                    for (int u = wv; u < ng; u += 4) {
                        const int slot = off_b + i0 + (int)s_list[g0 + u];
                        unsigned lo, hi;
                        mask_tile_rows64(&low[u][0][0] + lrow_off, wy, tx == 0, thr, lo, hi);
                        lo &= (unsigned)cmask; hi &= (unsigned)(cmask >> 32);
                        if (yr < H) {
                            unsigned* o = (unsigned*)(masks + ((size_t)slot * H + yr) * (W >> 3) + (x0 >> 3));
                            if (valid >= 64 && !(W & 63)) *(uint2*)o = make_uint2(lo, hi);      // rows of W/8 bytes with W % 64 == 0: 8-byte aligned
                            else {
                                o[0] = lo;
                                if (valid >= 64) o[1] = hi;
                            }
                        }
                    }
                } else if (wvy < H) {
                    // bytes: the wave owns 16 rows of the tile, a lane is a column (mask_rows16: coalesced byte stores)
                    for (int u = 0; u < ng; ++u) {
                        const int slot = off_b + i0 + (int)s_list[g0 + u];
                        unsigned lo = 0u, hi = 0u;
                        mask_rows16<true>(&low[u][0][0], ro, kc0, kc1, lx1, wvy == 0, thr, lo, hi, masks + (size_t)slot * H * W, W, H, wvy, xg);
                    }
                }
                MASK_T(4);                                                 // this wave's bands upsampled and stored
                if (g0 + MG < nlist) __syncthreads();                      // the next group overwrites `low`
                MASK_T(5);
            }
        }
    }
#ifdef VTI_STAMPS
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(&g_mask_acc[i], macc[i]);
        atomicAdd(&g_mask_acc[6], mitems); atomicAdd(&g_mask_acc[7], mpairs);
    }
#endif
}

size_t masks_workspace_bytes(int capacity, int H, int W) {
    const size_t tiles = (size_t)((H + MT - 1) / MT) * ((W + MT - 1) / MT);
    return 256 + ((((size_t)capacity * tiles * 8) + 255) & ~(size_t)255);      // nitems | items (the nm != 32 kernel's work list)
}

hipError_t launch_masks(int dtype, const float* dets, const int* counts, const void* proto, int B, int max_det, int nm,
                        int Hp, int Wp, int H, int W, int mode, int packing, uint8_t* masks, int capacity,
                        int* offsets, void* ws, hipStream_t st) {
    if (B == 0) return hipSuccess;
    if (Hp * 4 != H || Wp * 4 != W || nm > 64) return hipErrorInvalidValue;   // tile geometry assumes stride-4 prototypes
    int* nitems = (int*)ws;
    int2* items = (int2*)((char*)ws + 256);
    if (max_det > 65535 || B > 32767 || H > 64 * MT || W > 64 * MT || capacity > (1 << 19)) return hipErrorInvalidValue;    // item record fields
    const size_t slot_bytes = (size_t)H * (packing == VTI_PACK_U8 ? W : W / 8);      // H, W multiples of 32: a multiple of 128
    if (capacity > 0 && ((slot_bytes & 15) || ((uintptr_t)masks & 15))) return hipErrorInvalidValue;
    if (capacity > 0 && B <= 1024) {
        hipLaunchKernelGGL(mask_clear_kernel<true>, dim3(capacity), dim3(256), 0, st, counts, offsets, nitems, B, max_det, (int)slot_bytes, capacity, masks);
    } else {
        hipLaunchKernelGGL(mask_offsets_kernel, dim3(1), dim3(256), (size_t)(B + 1) * sizeof(int), st, counts, B, max_det, offsets, nitems);
        if (capacity <= 0) return hipGetLastError();
        // (running the clear on a side stream next to the plan was tried: the two event hops cost more than the 13 us they hide)
        hipLaunchKernelGGL(mask_clear_kernel<false>, dim3(capacity), dim3(256), 0, st, counts, offsets, nitems, B, max_det, (int)slot_bytes, capacity, masks);
    }
    // persistent blocks walk the work list: exactly as many as are resident at once (a second round of late blocks would run
    // on a mostly empty chip), a multiple of 8 for the per-XCD partition
    static int per_cu_dev[kMaxDevices][4] = {};
    int* per_cu = per_cu_dev[current_device_slot()];
    const int kidx = (dtype == VTI_F16 ? 0 : 2) + (nm == 32 ? 0 : 1);
    if (!per_cu[kidx]) {
        int nb = 0;
        const void* fn = dtype == VTI_F16 ? (nm == 32 ? (const void*)masks_group_kernel<half_t> : (const void*)masks_kernel<half_t>)
                                          : (nm == 32 ? (const void*)masks_group_kernel<float> : (const void*)masks_kernel<float>);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0) != hipSuccess || nb < 1) nb = 4;
        per_cu[kidx] = nb > 8 ? 8 : nb;
    }
    int grid = 256 * per_cu[kidx];
    if (const char* e = getenv("VTI_MASK_WGS_PER_CU")) grid = 256 * std::max(1, std::min(per_cu[kidx], atoi(e)));   // experiments: fewer resident workgroups
    if (nm == 32) {
        // the (frame, tile) list is static: no plan, no work-list memory
        if (dtype == VTI_F16)
            hipLaunchKernelGGL(masks_group_kernel<half_t>, dim3(grid), dim3(256), 0, st, dets, offsets, (const half_t*)proto, B, max_det, Hp, Wp,
                               H, W, mode, packing, masks, capacity);
        else
            hipLaunchKernelGGL(masks_group_kernel<float>, dim3(grid), dim3(256), 0, st, dets, offsets, (const float*)proto, B, max_det, Hp, Wp,
                               H, W, mode, packing, masks, capacity);
#ifdef VTI_STAMPS
        {
            (void)hipStreamSynchronize(st);
            static int calls = 0;
            if (++calls == 3) {
                unsigned long long h[8];
                (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mask_acc), sizeof h);
                const char* nm_[6] = {"decode + issue protos", "list build", "coeffs + MFMA + low", "barrier", "upsample + store", "barrier (next group)"};
                fprintf(stderr, "[mask stamps] %llu (frame, tile) items, %llu (instance, tile) pairs over %d calls (wave 0 of each block; s_memtime ticks per item)\n", h[6], h[7], calls);
                for (int i = 0; i < 6; ++i) fprintf(stderr, "[mask stamps] %-22s %9.1f\n", nm_[i], (double)h[i] / (double)h[6]);
            }
        }
#endif
        return hipGetLastError();
    }
    hipLaunchKernelGGL(mask_plan_kernel, dim3((capacity + 255) / 256), dim3(256), 0, st, dets, offsets, B, max_det, 6 + nm, H, W,
                       capacity, items, nitems);
    if (dtype == VTI_F16)
        hipLaunchKernelGGL(masks_kernel<half_t>, dim3(grid), dim3(256), 0, st, dets, offsets, (const half_t*)proto, B, max_det, nm, Hp, Wp,
                           H, W, mode, packing, masks, items, nitems);
    else
        hipLaunchKernelGGL(masks_kernel<float>, dim3(grid), dim3(256), 0, st, dets, offsets, (const float*)proto, B, max_det, nm, Hp, Wp,
                           H, W, mode, packing, masks, items, nitems);
    return hipGetLastError();
}

// =====================================================================================
// U8 scale_boxes + clip_boxes
// =====================================================================================
__global__ void scale_boxes_kernel(const float* __restrict__ dets, const int* __restrict__ counts, int B, int max_det,
                                   int row, float padx, float pady, float gain, float W0, float H0,
                                   float* __restrict__ xyxy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * max_det) return;
    const int b = i / max_det, k = i - b * max_det;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < counts[b]) {
        const float* d = dets + (size_t)i * row;
        o.x = fminf(fmaxf((d[0] - padx) / gain, 0.f), W0);
        o.y = fminf(fmaxf((d[1] - pady) / gain, 0.f), H0);
        o.z = fminf(fmaxf((d[2] - padx) / gain, 0.f), W0);
        o.w = fminf(fmaxf((d[3] - pady) / gain, 0.f), H0);
    }
    ((float4*)xyxy)[i] = o;
}

hipError_t launch_scale_boxes(const float* dets, const int* counts, int B, int max_det, int nm, int H, int W, int H0,
                              int W0, float* xyxy, hipStream_t st) {
    if (B * max_det == 0) return hipSuccess;
    const double gain = std::min((double)H / H0, (double)W / W0);
    const double padx = nearbyint((W - W0 * gain) / 2 - 0.1), pady = nearbyint((H - H0 * gain) / 2 - 0.1);
    hipLaunchKernelGGL(scale_boxes_kernel, dim3((B * max_det + 255) / 256), dim3(256), 0, st, dets, counts, B, max_det,
                       6 + nm, (float)padx, (float)pady, (float)gain, (float)W0, (float)H0, xyxy);
    return hipGetLastError();
}

// =====================================================================================
// A4-A7: measurement.py's mask post-processing
// =====================================================================================
// A4 measurement.py:70-86: cv2.resize(INTER_NEAREST) to the frame, (>0) -> u8, count_nonzero
__global__ __launch_bounds__(256) void mask_to_frame_kernel(const uint8_t* __restrict__ masks, int n, int H, int W, int H0,
                                                            int W0, uint8_t* __restrict__ bitmaps,
                                                            int* __restrict__ nonzero) {
    const int inst = blockIdx.y;
    const double ify = 1.0 / ((double)H0 / (double)H), ifx = 1.0 / ((double)W0 / (double)W);
    const long total = (long)H0 * W0;
    int local = 0;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W0), y = (int)(i / W0);
        const int sy = min((int)floor((double)y * ify), H - 1);
        const int sx = min((int)floor((double)x * ifx), W - 1);
        const uint8_t v = masks[((size_t)inst * H + sy) * W + sx] > 0 ? 1 : 0;
        bitmaps[(size_t)inst * total + i] = v;
        local += v;
    }
    // wave reduction, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&nonzero[inst], local);
}

hipError_t launch_mask_to_frame(const uint8_t* masks, int n, int H, int W, int H0, int W0, uint8_t* bitmaps,
                                int* nonzero, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(nonzero, 0, sizeof(int) * (size_t)n, st);
    if (e != hipSuccess) return e;
    const long total = (long)H0 * W0;
    const int gx = (int)((total + 255) / 256 < 512 ? (total + 255) / 256 : 512);
    hipLaunchKernelGGL(mask_to_frame_kernel, dim3(gx, n), dim3(256), 0, st, masks, n, H, W, H0, W0, bitmaps, nonzero);
    return hipGetLastError();
}

// A5+A6 measurement.py:160-185: OR of the selected bitmaps; per column the largest y that is set (-1 if none).
// Workgroup = 64 columns x 4 row groups.
__global__ __launch_bounds__(256) void union_envelope_kernel(const uint8_t* __restrict__ bitmaps,
                                                             const int* __restrict__ select, int nsel, int H0, int W0,
                                                             uint8_t* __restrict__ uni, int* __restrict__ envelope) {
    __shared__ int red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    int env = -1;
    if (col < W0) {
        for (int y = rg; y < H0; y += 4) {
            uint8_t v = 0;
            for (int s = 0; s < nsel; ++s) v |= bitmaps[((size_t)select[s] * H0 + y) * W0 + col];
            v = v ? 1 : 0;
            uni[(size_t)y * W0 + col] = v;
            if (v) env = y;
        }
    }
    red[rg][threadIdx.x & 63] = env;
    __syncthreads();
    if (rg == 0 && col < W0) envelope[col] = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
}

hipError_t launch_union_envelope(const uint8_t* bitmaps, const int* select, int nsel, int H0, int W0, uint8_t* uni,
                                 int* envelope, hipStream_t st) {
    hipLaunchKernelGGL(union_envelope_kernel, dim3((W0 + 63) / 64), dim3(256), 0, st, bitmaps, select, nsel, H0, W0, uni, envelope);
    return hipGetLastError();
}

// A7 measurement.py:302-318: cv2.moments of the binary image (m00, m10, m01) and min/max occupied column.
__global__ __launch_bounds__(1024) void mask_stats_kernel(const uint8_t* __restrict__ bitmaps, int H0, int W0,
                                                          long long* __restrict__ stats) {
    __shared__ long long s_m00[16], s_m10[16], s_m01[16];
    __shared__ int s_min[16], s_max[16];
    const int inst = blockIdx.x, tid = threadIdx.x;
    const uint8_t* m = bitmaps + (size_t)inst * H0 * W0;
    long long m00 = 0, m10 = 0, m01 = 0;
    int mn = 0x7fffffff, mx = -1;
    const long total = (long)H0 * W0;
    for (long i = tid; i < total; i += 1024) {
        if (m[i]) {
            const int x = (int)(i % W0), y = (int)(i / W0);
            m00 += 1; m10 += x; m01 += y;
            mn = min(mn, x); mx = max(mx, x);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        m00 += __shfl_down(m00, o); m10 += __shfl_down(m10, o); m01 += __shfl_down(m01, o);
        mn = min(mn, __shfl_down(mn, o)); mx = max(mx, __shfl_down(mx, o));
    }
    if ((tid & 63) == 0) { s_m00[tid >> 6] = m00; s_m10[tid >> 6] = m10; s_m01[tid >> 6] = m01; s_min[tid >> 6] = mn; s_max[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) { m00 += s_m00[w]; m10 += s_m10[w]; m01 += s_m01[w]; mn = min(mn, s_min[w]); mx = max(mx, s_max[w]); }
        long long* o = stats + (size_t)inst * 5;
        o[0] = m00; o[1] = m10; o[2] = m01; o[3] = m00 ? mn : -1; o[4] = mx;
    }
}

hipError_t launch_mask_stats(const uint8_t* bitmaps, int n, int H0, int W0, long long* stats, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(mask_stats_kernel, dim3(n), dim3(1024), 0, st, bitmaps, H0, W0, stats);
    return hipGetLastError();
}

}  // namespace vti
