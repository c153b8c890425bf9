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

template <typename T> struct V16;
template <> struct V16<half_t> { typedef half8 vec; static constexpr int N = 8; };
template <> struct V16<float> { typedef f32x4 vec; static constexpr int N = 4; };

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
        vec m5, m9, m13;
#pragma unroll
        for (int j = 0; j < N; ++j) m5[j] = m9[j] = m13[j] = (T)(-INFINITY);
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)p.H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)p.W) continue;
                const vec v = *(const vec*)(in + ((size_t)(b * p.H + yy) * p.W + xx) * p.ld + p.in_coff + c * N);
                const int ad = max(abs(dy), abs(dx));
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    m13[j] = v[j] > m13[j] ? v[j] : m13[j];
                    if (ad <= 4) m9[j] = v[j] > m9[j] ? v[j] : m9[j];
                    if (ad <= 2) m5[j] = v[j] > m5[j] ? v[j] : m5[j];
                }
            }
        }
        T* o = (T*)p.out + ((size_t)(b * p.H + y) * p.W + x) * p.ld + p.out_coff + c * N;
        *(vec*)o = m5;
        *(vec*)(o + p.C) = m9;
        *(vec*)(o + 2 * p.C) = m13;
    }
}

hipError_t launch_sppf_pool(int dtype, const PoolParams& p, hipStream_t st) {
    const long total = (long)p.B * p.H * p.W * (p.C / (dtype == VTI_F16 ? 8 : 4));
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == VTI_F16) hipLaunchKernelGGL(sppf_pool_kernel<half_t>, dim3(grid), dim3(256), 0, st, p);
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
// One thread per (frame, anchor); stores are coalesced along the anchor axis of pred [B,no,A].
__global__ __launch_bounds__(256) void decode_kernel(const DecodeParams p) {
    const long total = (long)p.B * p.A;
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i >= total) return;
    const int a = (int)(i % p.A);
    const int b = (int)(i / p.A);
    int l = 0;
    if (a >= p.a0[2]) l = 2; else if (a >= p.a0[1]) l = 1;
    const int la = a - p.a0[l];
    const int W = p.W[l], HW = p.H[l] * W;
    const int gy = la / W, gx = la - gy * W;
    const size_t pix = (size_t)b * HW + la;
    const int no = 4 + p.nc + p.nm;
    float* out = p.pred + (size_t)b * no * p.A + a;

    const float* bx = p.box[l] + pix * (4 * p.reg_max);
    float dist[4];
    for (int s = 0; s < 4; ++s) {
        float v[16];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[k] = bx[s * 16 + k]; mx = fmaxf(mx, v[k]); }
        float sum = 0.f, acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const float e = expf(v[k] - mx); sum += e; v[k] = e; }
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += (v[k] / sum) * (float)k;
        dist[s] = acc;
    }
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
    const float st = (float)p.stride[l];
    out[0 * (size_t)p.A] = ((x1 + x2) / 2.0f) * st;
    out[1 * (size_t)p.A] = ((y1 + y2) / 2.0f) * st;
    out[2 * (size_t)p.A] = (x2 - x1) * st;
    out[3 * (size_t)p.A] = (y2 - y1) * st;
    const float* cl = p.cls[l] + pix * p.nc;
    for (int c = 0; c < p.nc; ++c) out[(size_t)(4 + c) * p.A] = 1.0f / (1.0f + expf(-cl[c]));
    const float* mc = p.mc[l] + pix * p.nm;
    for (int c = 0; c < p.nm; ++c) out[(size_t)(4 + p.nc + c) * p.A] = mc[c];
}

hipError_t launch_decode(const DecodeParams& p, hipStream_t st) {
    const long total = (long)p.B * p.A;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
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
        dst[i] = (float)src[((size_t)(b * H + y) * W + x) * ld + coff + c];
    }
}

hipError_t launch_debug_nchw(int elem_is_f32, const void* src, int B, int H, int W, int C, int ld, int coff,
                             float* dst, hipStream_t st) {
    const long total = (long)B * C * H * W;
    if (total == 0) return hipSuccess;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (elem_is_f32) hipLaunchKernelGGL(debug_nchw_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)src, B, H, W, C, ld, coff, dst);
    else hipLaunchKernelGGL(debug_nchw_kernel<half_t>, dim3(grid), dim3(256), 0, st, (const half_t*)src, B, H, W, C, ld, coff, dst);
    return hipGetLastError();
}

}  // namespace vti
