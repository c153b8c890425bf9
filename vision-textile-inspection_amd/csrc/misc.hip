// Non-GEMM pieces of the YOLOv8-seg forward pass for gfx950: SPPF max-pools, nearest-2x
// upsample into a concat slice, and the Detect/Segment decode (DFL + dist2bbox + sigmoid).
// SURVEY.md section 8 rows U2 (SPPF, Upsample) and U3/U4 (decode); reference call site
// measurement.py:208-210.  All are HBM/L2-bound streaming kernels: 16 B per lane, coalesced
// along the NHWC channel axis.
#include "vti_internal.h"

namespace vti {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// h2 (VTI_H2, conv_dev.h): one element = the fp16 pair (hi | lo << 16) of value * 16; here it only has to be compared and copied
struct h2e { unsigned u; };
__device__ __forceinline__ float h2_val(unsigned u) {       // monotone in the element's value (the common scale does not matter)
    return (float)__builtin_bit_cast(half_t, (unsigned short)(u & 0xffffu)) + (float)__builtin_bit_cast(half_t, (unsigned short)(u >> 16));
}

template <typename T> struct V16;
template <> struct V16<half_t> { typedef half8 vec; static constexpr int N = 8; };
template <> struct V16<float> { typedef f32x4 vec; static constexpr int N = 4; };
template <> struct V16<h2e> { typedef u32x4 vec; static constexpr int N = 4; };

__device__ __forceinline__ half8 vmax(half8 a, half8 b) { return __builtin_elementwise_max(a, b); }   // v_pk_max_f16
__device__ __forceinline__ f32x4 vmax(f32x4 a, f32x4 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ u32x4 vmax(u32x4 a, u32x4 b) {       // h2 pairs: the pair with the larger decoded value, bits unchanged
    u32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = h2_val(b[j]) > h2_val(a[j]) ? b[j] : a[j];
    return r;
}
template <typename T> __device__ __forceinline__ typename V16<T>::vec vlowest() {
    typename V16<T>::vec v;
    if constexpr (sizeof(T) == 4 && !__is_same(T, float)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = 0xfc00u;             // hi = -inf, lo = +0
    } else {
#pragma unroll
        for (int j = 0; j < V16<T>::N; ++j) v[j] = (T)(-INFINITY);
    }
    return v;
}

// ---- SPPF: three chained MaxPool2d(5,1,2) == max over 5x5 / 9x9 / 13x13 windows clipped to the
// image (the implicit -inf padding makes chaining and direct windows identical).
template <typename T>
__global__ __launch_bounds__(256) void sppf_pool_kernel(const PoolParams p) {
    using vec = typename V16<T>::vec;
    constexpr int N = V16<T>::N;
    const int cv = p.C / N;
    const long total = (long)p.B * p.H * p.W * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int x = (int)(r % p.W); r /= p.W;
        const int y = (int)(r % p.H);
        const int b = (int)(r / p.H);
        const T* in = (const T*)p.in;
        vec m5 = vlowest<T>(), m9 = vlowest<T>(), m13 = vlowest<T>();
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)p.H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)p.W) continue;
                const vec v = *(const vec*)(in + ((size_t)(b * p.H + yy) * p.W + xx) * p.ld + p.in_coff + c * N);
                const int ad = max(abs(dy), abs(dx));
                m13 = vmax(m13, v);
                if (ad <= 4) m9 = vmax(m9, v);
                if (ad <= 2) m5 = vmax(m5, v);
            }
        }
        T* o = (T*)p.out + ((size_t)(b * p.H + y) * p.W + x) * p.ld + p.out_coff + c * N;
        *(vec*)o = m5;
        *(vec*)(o + p.C) = m9;
        *(vec*)(o + 2 * p.C) = m13;
    }
}

// Fast path: one workgroup per (frame, group of G 16-byte channel pieces); the whole H x W map of that group sits in
// LDS and the three chained 5x5 pools run as separable row/column passes -- exactly the reference's chained
// MaxPool2d(5,1,2) (max is exact, so fp16/fp32 results are bit-identical to the chained form).  G consecutive lanes
// touch G x 16 contiguous bytes of one pixel (G = 4: 64-byte segments instead of one 16-byte piece per 1-KiB pixel row).
template <typename T, int G>
__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(const PoolParams p) {
    using vec = typename V16<T>::vec;
    constexpr int N = V16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int HW = p.H * p.W;
    vec* cur = (vec*)smem;              // [HW][G]
    vec* tmp = cur + HW * G;            // [HW][G]
    const int cg = p.C / (N * G);
    const int b = blockIdx.x / cg, c = blockIdx.x - b * cg;
    const T* in = (const T*)p.in + (size_t)b * HW * p.ld + p.in_coff + c * (N * G);
    T* out = (T*)p.out + (size_t)b * HW * p.ld + p.out_coff + c * (N * G);
    {   // all loads first (a load consumed inside the loop body is one memory round trip per iteration)
        constexpr int MAXIT = 8;
        if (HW * G <= MAXIT * 256) {
            vec r[MAXIT];
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int i = threadIdx.x + it * 256;
                const int ic = i < HW * G ? i : 0;
                r[it] = *(const vec*)(in + (size_t)(ic / G) * p.ld + (ic % G) * N);
            }
#pragma unroll
            for (int it = 0; it < MAXIT; ++it) {
                const int i = threadIdx.x + it * 256;
                if (i < HW * G) cur[i] = r[it];
            }
        } else {
            for (int i = threadIdx.x; i < HW * G; i += 256) cur[i] = *(const vec*)(in + (size_t)(i / G) * p.ld + (i % G) * N);
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = threadIdx.x; i < HW * G; i += 256) {   // row pass: max over x-2..x+2
            const int pix = i / G, v_ = i - pix * G;
            const int y = pix / p.W, x = pix - y * p.W;
            vec m = cur[i];
            for (int d = -2; d <= 2; ++d) {
                const int xx = x + d;
                if (d == 0 || (unsigned)xx >= (unsigned)p.W) continue;
                const vec v = cur[(y * p.W + xx) * G + v_];
                m = vmax(m, v);                              // v_pk_max_f16 / v_max_f32: 4 instructions per 16-byte piece
            }
            tmp[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < HW * G; i += 256) {   // column pass + store this pool level
            const int pix = i / G, v_ = i - pix * G;
            const int y = pix / p.W, x = pix - y * p.W;
            vec m = tmp[i];
            for (int d = -2; d <= 2; ++d) {
                const int yy = y + d;
                if (d == 0 || (unsigned)yy >= (unsigned)p.H) continue;
                const vec v = tmp[(yy * p.W + x) * G + v_];
                m = vmax(m, v);                              // v_pk_max_f16 / v_max_f32: 4 instructions per 16-byte piece
            }
            cur[i] = m;
            *(vec*)(out + (size_t)pix * p.ld + pass * p.C + v_ * N) = m;
        }
        __syncthreads();
    }
}

hipError_t launch_sppf_pool(int dtype, const PoolParams& p, hipStream_t st) {
    const int N = dtype == VTI_F16 ? 8 : 4;
    const long total = (long)p.B * p.H * p.W * (p.C / N);
    if (total == 0) return hipSuccess;
    const size_t lds1 = (size_t)2 * p.H * p.W * 16;
    if (lds1 * 4 <= 64 * 1024 && p.C % (4 * N) == 0) {      // four 16-byte pieces per workgroup: 64-byte global segments
        const int grid = p.B * (p.C / (4 * N));
        if (dtype == VTI_F16) hipLaunchKernelGGL((sppf_pool_lds_kernel<half_t, 4>), dim3(grid), dim3(256), lds1 * 4, st, p);
        else if (dtype == VTI_H2) hipLaunchKernelGGL((sppf_pool_lds_kernel<h2e, 4>), dim3(grid), dim3(256), lds1 * 4, st, p);
        else hipLaunchKernelGGL((sppf_pool_lds_kernel<float, 4>), dim3(grid), dim3(256), lds1 * 4, st, p);
        return hipGetLastError();
    }
    if (lds1 <= 64 * 1024) {
        const int grid = p.B * (p.C / N);
        if (dtype == VTI_F16) hipLaunchKernelGGL((sppf_pool_lds_kernel<half_t, 1>), dim3(grid), dim3(256), lds1, st, p);
        else if (dtype == VTI_H2) hipLaunchKernelGGL((sppf_pool_lds_kernel<h2e, 1>), dim3(grid), dim3(256), lds1, st, p);
        else hipLaunchKernelGGL((sppf_pool_lds_kernel<float, 1>), dim3(grid), dim3(256), lds1, st, p);
        return hipGetLastError();
    }
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == VTI_F16) hipLaunchKernelGGL(sppf_pool_kernel<half_t>, dim3(grid), dim3(256), 0, st, p);
    else if (dtype == VTI_H2) hipLaunchKernelGGL(sppf_pool_kernel<h2e>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(sppf_pool_kernel<float>, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---- nn.Upsample(scale_factor=2, mode="nearest") written into a channel slice of the concat buffer
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const Up2Params p) {
    using vec = typename V16<T>::vec;
    constexpr int N = V16<T>::N;
    const int cv = p.C / N;
    const int Ho = 2 * p.H, Wo = 2 * p.W;
    const long total = (long)p.B * Ho * Wo * cv;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long r = i / cv;
        const int x = (int)(r % Wo); r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const vec v = *(const vec*)((const T*)p.in + ((size_t)(b * p.H + (y >> 1)) * p.W + (x >> 1)) * p.in_ld + p.in_coff + c * N);
        *(vec*)((T*)p.out + ((size_t)(b * Ho + y) * Wo + x) * p.out_ld + p.out_coff + c * N) = v;
    }
}

hipError_t launch_upsample2x(int dtype, const Up2Params& p, hipStream_t st) {
    const long total = (long)p.B * 4 * p.H * p.W * (p.C / (dtype == VTI_F16 ? 8 : 4));
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == VTI_F16) hipLaunchKernelGGL(upsample2x_kernel<half_t>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---- Detect/Segment inference decode, fp32 throughout:
//   DFL: softmax over reg_max bins per side, expectation with arange weights -> l,t,r,b
//   dist2bbox(xywh=True): x1y1 = anchor - lt, x2y2 = anchor + rb, (cxcy, wh) * stride
//   class scores -> sigmoid; mask coefficients copied.
// One workgroup per (frame, 64 consecutive anchors of one level).  The anchors' rows of the three
// head tensors ([anchor][64 | nc | nm], anchor-major) are contiguous, so they are loaded with
// coalesced dwords into an LDS tile [64][no_in+1]; the anchor-major pred [B,A,4+nc+nm] rows are then
// written as one contiguous run.
constexpr int DEC_TA = 64;

__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p, int tiles0, int tiles1, int tiles_per_frame) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.x / tiles_per_frame;
    int t = blockIdx.x - b * tiles_per_frame;
    int l = 0;
    if (t >= tiles0 + tiles1) { l = 2; t -= tiles0 + tiles1; } else if (t >= tiles0) { l = 1; t -= tiles0; }
    const int W = p.W[l], HW = p.H[l] * W;
    const int la0 = t * DEC_TA;
    const int na = min(DEC_TA, HW - la0);
    const int nbox = 4 * p.reg_max, nin = nbox + p.nc + p.nm, pitch = nin + 1;
    const int tid = threadIdx.x;
    const size_t pix0 = (size_t)b * HW + la0;
    // coalesced loads: each source is one contiguous run of na rows
    const float* src = p.box[l] + pix0 * nbox;
    for (int i = tid; i < na * nbox; i += 256) { const int a = i / nbox, k = i - a * nbox; tile[a * pitch + k] = src[i]; }
    src = p.cls[l] + pix0 * p.nc;
    for (int i = tid; i < na * p.nc; i += 256) { const int a = i / p.nc, k = i - a * p.nc; tile[a * pitch + nbox + k] = src[i]; }
    src = p.mc[l] + pix0 * p.nm;
    for (int i = tid; i < na * p.nm; i += 256) { const int a = i / p.nm, k = i - a * p.nm; tile[a * pitch + nbox + p.nc + k] = src[i]; }
    __syncthreads();
    // DFL expectation: thread = (anchor, side)
    float dist = 0.f;
    const int a_ = tid >> 2, side = tid & 3;
    if (a_ < na) {
        const float* v = tile + a_ * pitch + side * 16;
        float e[16];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = v[k]; mx = fmaxf(mx, e[k]); }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = expf(e[k] - mx); sum += e[k]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) dist += (e[k] / sum) * (float)k;
    }
    __syncthreads();
    if (a_ < na) tile[a_ * pitch + side] = dist;     // l,t,r,b overwrite the first 4 box logits
    __syncthreads();
    const int no = 4 + p.nc + p.nm;
    float* out = p.pred + ((size_t)b * p.A + p.a0[l] + la0) * no;       // anchor-major pred [B, A, no]: rows are contiguous
    const float st = (float)p.stride[l];
    for (int i = tid; i < no * DEC_TA; i += 256) {
        const int a = i / no, ch = i - a * no;
        if (a >= na) continue;
        const float* r = tile + a * pitch;
        float v;
        if (ch < 4) {
            const int la = la0 + a;
            const int gy = la / W, gx = la - gy * W;
            const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
            const float x1 = ax - r[0], y1 = ay - r[1], x2 = ax + r[2], y2 = ay + r[3];
            v = ch == 0 ? ((x1 + x2) / 2.0f) * st : ch == 1 ? ((y1 + y2) / 2.0f) * st : ch == 2 ? (x2 - x1) * st : (y2 - y1) * st;
        } else if (ch < 4 + p.nc) {
            v = 1.0f / (1.0f + expf(-r[nbox + ch - 4]));
        } else {
            v = r[nbox + ch - 4];
        }
        out[(size_t)a * no + ch] = v;
    }
}

hipError_t launch_decode(const DecodeParams& p, hipStream_t st) {
    if ((long)p.B * p.A == 0) return hipSuccess;
    if (p.reg_max != 16) return hipErrorInvalidValue;
    int tiles[3];
    for (int l = 0; l < 3; ++l) tiles[l] = (p.H[l] * p.W[l] + DEC_TA - 1) / DEC_TA;
    const int tpf = tiles[0] + tiles[1] + tiles[2];
    const size_t lds = (size_t)DEC_TA * (4 * p.reg_max + p.nc + p.nm + 1) * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;     // nc up to ~150 classes
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)(p.B * tpf)), dim3(256), lds, st, p, tiles[0], tiles[1], tpf);
    return hipGetLastError();
}

// ---- box-only decode for pred_scatter plans: DFL + dist2bbox on the box tower's logits; class scores and
// mask coefficients were already written into pred by the towers' fused 1x1 stage.
__global__ __launch_bounds__(256) void box_decode_kernel(const DecodeParams p, int tiles0, int tiles1, int tiles_per_frame) {
    __shared__ float tile[DEC_TA * 65];
    __shared__ float dist4[DEC_TA * 4];
    const int b = blockIdx.x / tiles_per_frame;
    int t = blockIdx.x - b * tiles_per_frame;
    int l = 0;
    if (t >= tiles0 + tiles1) { l = 2; t -= tiles0 + tiles1; } else if (t >= tiles0) { l = 1; t -= tiles0; }
    const int W = p.W[l], HW = p.H[l] * W;
    const int la0 = t * DEC_TA;
    const int na = min(DEC_TA, HW - la0);
    const int tid = threadIdx.x;
    const float* src = p.box[l] + ((size_t)b * HW + la0) * 64;
    for (int i = tid; i < na * 64; i += 256) tile[(i >> 6) * 65 + (i & 63)] = src[i];
    __syncthreads();
    const int a_ = tid >> 2, side = tid & 3;
    if (a_ < na) {
        const float* v = tile + a_ * 65 + side * 16;
        float e[16];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = v[k]; mx = fmaxf(mx, e[k]); }
        float sum = 0.f, dist = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = expf(e[k] - mx); sum += e[k]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) dist += (e[k] / sum) * (float)k;
        dist4[a_ * 4 + side] = dist;
    }
    __syncthreads();
    const int no = 4 + p.nc + p.nm;
    const float st = (float)p.stride[l];
    const int a = tid >> 2, ch = tid & 3;           // 64 anchors x 4 box values: 16 contiguous bytes per anchor
    if (a < na) {
        const float* r = dist4 + a * 4;
        const int la = la0 + a;
        const int gy = la / W, gx = la - gy * W;
        const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
        const float x1 = ax - r[0], y1 = ay - r[1], x2 = ax + r[2], y2 = ay + r[3];
        const float v = ch == 0 ? ((x1 + x2) / 2.0f) * st : ch == 1 ? ((y1 + y2) / 2.0f) * st : ch == 2 ? (x2 - x1) * st : (y2 - y1) * st;
        p.pred[((size_t)b * p.A + p.a0[l] + la) * no + ch] = v;
    }
}

hipError_t launch_box_decode(const DecodeParams& p, hipStream_t st) {
    if ((long)p.B * p.A == 0) return hipSuccess;
    if (p.reg_max != 16) return hipErrorInvalidValue;
    int tiles[3];
    for (int l = 0; l < 3; ++l) tiles[l] = (p.H[l] * p.W[l] + DEC_TA - 1) / DEC_TA;
    const int tpf = tiles[0] + tiles[1] + tiles[2];
    hipLaunchKernelGGL(box_decode_kernel, dim3((unsigned)(p.B * tpf)), dim3(256), 0, st, p, tiles[0], tiles[1], tpf);
    return hipGetLastError();
}

// ---- test hook: NHWC channel slice -> f32 NCHW
template <typename T>
__global__ void debug_nchw_kernel(const T* src, int B, int H, int W, int C, int ld, int coff, float* dst) {
    const long total = (long)B * C * H * W;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W);
        long r = i / W;
        const int y = (int)(r % H); r /= H;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        if constexpr (sizeof(T) == 4 && !__is_same(T, float)) dst[i] = h2_val(src[((size_t)(b * H + y) * W + x) * ld + coff + c].u) * (1.0f / 16.0f);
        else dst[i] = (float)src[((size_t)(b * H + y) * W + x) * ld + coff + c];
    }
}

hipError_t launch_debug_nchw(int elem_is_f32, const void* src, int B, int H, int W, int C, int ld, int coff,
                             float* dst, hipStream_t st) {
    const long total = (long)B * C * H * W;
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (elem_is_f32 == 2) hipLaunchKernelGGL(debug_nchw_kernel<h2e>, dim3(grid), dim3(256), 0, st, (const h2e*)src, B, H, W, C, ld, coff, dst);
    else if (elem_is_f32) hipLaunchKernelGGL(debug_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, B, H, W, C, ld, coff, dst);
    else hipLaunchKernelGGL(debug_nchw_kernel<half_t>, dim3(grid), dim3(256), 0, st, (const half_t*)src, B, H, W, C, ld, coff, dst);
    return hipGetLastError();
}

}  // namespace vti
