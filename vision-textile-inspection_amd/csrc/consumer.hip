// Consumer-side reductions of measurement.py on the GPU, second generation:
//   * A4 + A7 and A4 + A5 + A6 straight from the BIT-PACKED masks vti_masks writes: the nearest resize to the frame
//     (measurement.py:79, cv2.INTER_NEAREST) is folded into integer weight tables, so no [n, H0, W0] bitmap is ever
//     materialised and the consumer's payload shrinks from n*H*W bytes to n*5 + W0 integers (what a multi-GPU gather ships);
//   * N3: batched pixel -> world (cv2.undistortPoints' 5-iteration inverse of the k1,k2,p1,p2,k3 model + ray/plane
//     intersection, measurement.py:44-65) and the 1-D 2-means of measurement.py:88-113, in float64 as the reference.
// All integer results are exact; the float64 ones follow the reference's operation order (-ffp-contract=off).
#include <climits>

#include "vti_internal.h"

namespace vti {

// cv2.resize(INTER_NEAREST): src index of destination index d = min(floor(d * (1 / (dst / src))), src - 1), in double.
__device__ __forceinline__ int nn_src(int d, double inv_scale, int ssize) {
    const int s = (int)floor((double)d * inv_scale);
    return s < ssize - 1 ? s : ssize - 1;
}

// ---- A4 + A7: per-instance moments / column extents of the frame-sized bitmap, from the bit-packed mask -------------
// Destination pixel (y, x) of the H0 x W0 bitmap copies source pixel (sy(y), sx(x)); so with
//   cx[sx] = #{x : sx(x) = sx}, xs[sx] = sum of those x, cy[sy], ys[sy] likewise,
//   m00 = sum_set cy*cx,  m10 = sum_set cy*xs,  m01 = sum_set ys*cx,  min/max col = first/last x of the extreme set sx
// (over set source pixels that have at least one destination row and column).  One workgroup per instance.
__global__ __launch_bounds__(256) void mask_stats_bits_kernel(const unsigned* __restrict__ bits, const int* __restrict__ n_live,
                                                              int H, int W, int H0, int W0, long long* __restrict__ stats) {
    extern __shared__ int tab[];            // cx[W] xs[W] xf[W] xl[W] cy[H] ys[H]
    int* cx = tab; int* xs = cx + W; int* xf = xs + W; int* xl = xf + W; int* cy = xl + W; int* ys = cy + H;
    const int tid = threadIdx.x, slot = blockIdx.x;
    if (n_live && slot >= *n_live) {        // a dead slot of a fixed-capacity buffer: the empty-mask answer, nothing read
        if (tid < 5) stats[(size_t)slot * 5 + tid] = tid < 3 ? 0 : -1;
        return;
    }
    const bool ident = H0 == H && W0 == W;
    if (!ident) {
        for (int i = tid; i < W; i += 256) { cx[i] = 0; xs[i] = 0; xf[i] = INT_MAX; xl[i] = -1; }
        for (int i = tid; i < H; i += 256) { cy[i] = 0; ys[i] = 0; }
        __syncthreads();
        const double ifx = 1.0 / ((double)W0 / (double)W), ify = 1.0 / ((double)H0 / (double)H);
        for (int x = tid; x < W0; x += 256) {
            const int s = nn_src(x, ifx, W);
            atomicAdd(&cx[s], 1); atomicAdd(&xs[s], x); atomicMin(&xf[s], x); atomicMax(&xl[s], x);
        }
        for (int y = tid; y < H0; y += 256) {
            const int s = nn_src(y, ify, H);
            atomicAdd(&cy[s], 1); atomicAdd(&ys[s], y);
        }
        __syncthreads();
    }
    const int wpr = W >> 5;                 // 32-bit words per mask row (W is a multiple of 32)
    // a slot is H * W / 8 bytes, a multiple of 128 (H, W multiples of 32), and `bits` is 16-byte aligned (checked by the launcher):
    // 16-byte loads, four of them in flight per thread before the first word is looked at (the loop was one dependent 4-byte load
    // per trip: latency-bound at a third of what the masks' L2 / HBM residency gives)
    const uint4* m4 = (const uint4*)(bits + (size_t)slot * H * wpr);
    const int n4 = (H * wpr) >> 2;
    long long m00 = 0, m10 = 0, m01 = 0;
    int mn = INT_MAX, mx = -1;
    auto word = [&](unsigned w, int sy, int x0) {
        if (!w) return;
        if (ident) {
            const int pc = __popc(w);
            // sum of the set bit positions: bit k of a position contributes 2^k times the population of its mask
            const int ps = __popc(w & 0xAAAAAAAAu) + 2 * __popc(w & 0xCCCCCCCCu) + 4 * __popc(w & 0xF0F0F0F0u) +
                           8 * __popc(w & 0xFF00FF00u) + 16 * __popc(w & 0xFFFF0000u);
            m00 += pc; m10 += (long long)x0 * pc + ps; m01 += (long long)sy * pc;
            mn = min(mn, x0 + __ffs(w) - 1); mx = max(mx, x0 + 31 - __clz(w));
        } else {
            const int wy = cy[sy];
            if (!wy) return;
            long long rc = 0, rx = 0;
            while (w) {
                const int sx = x0 + __ffs(w) - 1;
                w &= w - 1;
                if (!cx[sx]) continue;
                rc += cx[sx]; rx += xs[sx];
                mn = min(mn, xf[sx]); mx = max(mx, xl[sx]);
            }
            m00 += rc * wy; m10 += rx * wy; m01 += rc * ys[sy];
        }
    };
    constexpr int U = 4;
    for (int i0 = tid; i0 < n4; i0 += 256 * U) {
        uint4 q[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 256 * u;
            q[u] = i < n4 ? m4[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!(q[u].x | q[u].y | q[u].z | q[u].w)) continue;
            const int i = (i0 + 256 * u) << 2;
            int sy = i / wpr, c = i - sy * wpr;                           // word column inside the row; the four words may cross a row end
            const unsigned ww[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                word(ww[k], sy, c << 5);
                if (++c == wpr) { c = 0; ++sy; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        m00 += __shfl_down(m00, o); m10 += __shfl_down(m10, o); m01 += __shfl_down(m01, o);
        mn = min(mn, __shfl_down(mn, o)); mx = max(mx, __shfl_down(mx, o));
    }
    __shared__ long long r00[4], r10[4], r01[4];
    __shared__ int rmn[4], rmx[4];
    if ((tid & 63) == 0) { r00[tid >> 6] = m00; r10[tid >> 6] = m10; r01[tid >> 6] = m01; rmn[tid >> 6] = mn; rmx[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) { m00 += r00[w]; m10 += r10[w]; m01 += r01[w]; mn = min(mn, rmn[w]); mx = max(mx, rmx[w]); }
        long long* o = stats + (size_t)slot * 5;
        o[0] = m00; o[1] = m10; o[2] = m01; o[3] = m00 ? mn : -1; o[4] = mx;
    }
}

hipError_t launch_mask_stats_bits(const uint8_t* bits, int n, const int* n_live, int H, int W, int H0, int W0, long long* stats,
                                  hipStream_t st) {
    if (n == 0) return hipSuccess;
    if ((W & 31) || (H & 31) || ((uintptr_t)bits & 15) || (size_t)(4 * W + 2 * H) * 4 > 60 * 1024) return hipErrorInvalidValue;      // 16-byte loads
    hipLaunchKernelGGL(mask_stats_bits_kernel, dim3(n), dim3(256), (size_t)(4 * W + 2 * H) * 4, st, (const unsigned*)bits, n_live, H, W, H0, W0, stats);
    return hipGetLastError();
}

// ---- A4 + A5 + A6, batched: per frame the lower envelope of the union of its instances of class `cls` -----------------
// (measurement.py:160-185 on the frame-sized bitmaps of measurement.py:70-86).  envelope[b, x] = the largest frame row y whose
// source pixel (sy(y), sx(x)) is set in ANY selected instance, -1 if none.  yl[sy] = last frame row that maps to sy (-1: none),
// monotone in sy, so the answer is yl of the largest set source row.  A mask is zero outside its box grown by 8 px (two
// prototype pixels of bilinear reach, the same bound mask_plan_kernel uses), so only those rows are read.
__global__ __launch_bounds__(256) void envelope_bits_kernel(const unsigned* __restrict__ bits, const int* __restrict__ offsets,
                                                            const float* __restrict__ dets, int max_det, int row, int capacity,
                                                            int cls, int H, int W, int H0, int W0, int* __restrict__ envelope) {
    extern __shared__ int yl[];             // [H]
    __shared__ int red[4][64];
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int i = tid; i < H; i += 256) yl[i] = -1;
    __syncthreads();
    const double ify = 1.0 / ((double)H0 / (double)H), ifx = 1.0 / ((double)W0 / (double)W);
    for (int y = tid; y < H0; y += 256) atomicMax(&yl[nn_src(y, ify, H)], y);
    __syncthreads();
    const int x = blockIdx.x * 64 + (tid & 63), rg = tid >> 6;
    const int sx = nn_src(x < W0 ? x : W0 - 1, ifx, W);
    const int wpr = W >> 5;
    const int s0 = offsets[b], s1 = min(offsets[b + 1], capacity);
    int env = -1;
    // the frame's instances of the wanted class, listed first (a thread per instance; the order is irrelevant to a max): the row
    // search below then runs over those only -- walking all instances with one dependent class load each was the kernel's time
    __shared__ int s_sel[256][3];
    __shared__ int s_nsel;
    for (int r0 = s0; r0 < s1; r0 += 256) {
        __syncthreads();                                       // the previous round's list has been consumed
        if (tid == 0) s_nsel = 0;
        __syncthreads();
        if (r0 + tid < s1) {
            const float* d = dets + ((size_t)b * max_det + (r0 + tid - s0)) * row;
            if (cls < 0 || (int)d[5] == cls) {
                int ya = (int)floorf(d[1] - 8.f), yb = (int)ceilf(d[3] + 8.f);
                ya = max(ya, 0); yb = min(yb, H - 1);
                const int pos = atomicAdd(&s_nsel, 1);
                s_sel[pos][0] = r0 + tid; s_sel[pos][1] = ya; s_sel[pos][2] = yb;
            }
        }
        __syncthreads();
        const int nsel = s_nsel;
        for (int j = 0; j < nsel; ++j) {
            const int s = s_sel[j][0], ya = s_sel[j][1], yb = s_sel[j][2];
            const unsigned* m = bits + (size_t)s * H * wpr + (sx >> 5);
            for (int sy = yb - rg; sy >= ya; sy -= 4) {           // top of the search first: the first hit of a lane is its largest
                if ((m[(size_t)sy * wpr] >> (sx & 31)) & 1u) { env = max(env, yl[sy]); if (yl[sy] >= 0) break; }
            }
        }
    }
    red[rg][tid & 63] = env;
    __syncthreads();
    if (rg == 0 && x < W0) envelope[(size_t)b * W0 + x] = max(max(red[0][tid], red[1][tid]), max(red[2][tid], red[3][tid]));
}

hipError_t launch_envelope_bits(const uint8_t* bits, const int* offsets, const float* dets, int B, int max_det, int nm,
                                int capacity, int cls, int H, int W, int H0, int W0, int* envelope, hipStream_t st) {
    if (B == 0) return hipSuccess;
    if ((W & 31) || ((uintptr_t)bits & 3) || (size_t)H * 4 > 60 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(envelope_bits_kernel, dim3((W0 + 63) / 64, B), dim3(256), (size_t)H * 4, st, (const unsigned*)bits, offsets, dets,
                       max_det, 6 + nm, capacity, cls, H, W, H0, W0, envelope);
    return hipGetLastError();
}

// ---- N3a: pixel -> world on the fabric plane (measurement.py:44-65) ----------------------------------------------------
// cv2.undistortPoints(pts, K, dist, P=None) with its default criteria (COUNT 5): x0 = (u - cx) / fx, then 5 fixed-point steps
//   r2 = x^2 + y^2; icdist = 1 / (1 + ((k3 r2 + k2) r2 + k1) r2); dX = 2 p1 x y + p2 (r2 + 2 x^2); dY = p1 (r2 + 2 y^2) + 2 p2 x y
//   x = (x0 - dX) icdist; y = (y0 - dY) icdist          (k4..k6, s1..s4, tilt = 0 for the 5-coefficient model of
// camera_calibration.json; OpenCV leaves the loop if icdist < 0).  Then the ray (x, y, 1) meets the plane n.X + d = 0:
//   s = -d / (n . ray); X_cam = s ray; X_world = R^T (X_cam - t); no point when |n . ray| < 1e-9.
struct GeomParams { double fx, fy, cx, cy, k1, k2, p1, p2, k3; double R[9]; double t[3]; double n[3]; double d; };

__global__ __launch_bounds__(256) void pixels_to_world_kernel(const double* __restrict__ uv, int n, GeomParams g,
                                                               double* __restrict__ xyz, int* __restrict__ valid) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double u = uv[2 * i], v = uv[2 * i + 1];
    const double ifx = 1.0 / g.fx, ify = 1.0 / g.fy;
    double x = (u - g.cx) * ifx, y = (v - g.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + ((g.k3 * r2 + g.k2) * r2 + g.k1) * r2);
        if (icdist < 0) { x = (u - g.cx) * ifx; y = (v - g.cy) * ify; break; }
        const double dX = 2.0 * g.p1 * x * y + g.p2 * (r2 + 2.0 * x * x);
        const double dY = g.p1 * (r2 + 2.0 * y * y) + 2.0 * g.p2 * x * y;
        x = (x0 - dX) * icdist;
        y = (y0 - dY) * icdist;
    }
    const double denom = (g.n[0] * x + g.n[1] * y) + g.n[2];            // n . (x, y, 1)
    const bool ok = fabs(denom) >= 1e-9;
    const double s = -g.d / denom;
    const double c0 = s * x - g.t[0], c1 = s * y - g.t[1], c2 = s - g.t[2];
    double* o = xyz + 3 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 3; ++k)                                          // R^T row k = column k of R
        o[k] = ok ? (g.R[k] * c0 + g.R[3 + k] * c1) + g.R[6 + k] * c2 : 0.0;
    valid[i] = ok ? 1 : 0;
}

hipError_t launch_pixels_to_world(const double* uv, int n, const double* K, const double* dist, const double* R,
                                  const double* t, double* xyz, int* valid, hipStream_t st) {
    if (n == 0) return hipSuccess;
    GeomParams g;
    g.fx = K[0]; g.fy = K[4]; g.cx = K[2]; g.cy = K[5];
    g.k1 = dist[0]; g.k2 = dist[1]; g.p1 = dist[2]; g.p2 = dist[3]; g.k3 = dist[4];
    for (int i = 0; i < 9; ++i) g.R[i] = R[i];
    for (int i = 0; i < 3; ++i) { g.t[i] = t[i]; g.n[i] = R[3 * i + 2]; }       // plane normal = third column of R (measurement.py:46)
    g.d = -((g.n[0] * g.t[0] + g.n[1] * g.t[1]) + g.n[2] * g.t[2]);            // measurement.py:47
    hipLaunchKernelGGL(pixels_to_world_kernel, dim3((n + 255) / 256), dim3(256), 0, st, uv, n, g, xyz, valid);
    return hipGetLastError();
}

// ---- N3b: kmeans_1d_two_clusters (measurement.py:88-113), one wave per frame ------------------------------------------
// numpy's float64 `mean` = pairwise sum / count; the loop's exit test compares the new centres with `==`, so the sums are
// formed exactly as numpy forms them (8 interleaved partial sums per block of <= 128, halves above that).
__device__ __forceinline__ double np_pairwise_leaf(const double* a, int n) {       // n <= 128
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// numpy recurses on halves above 128 elements (the first half a multiple of 8 long): sum(a, n) = sum(a, n2) + sum(a + n2, n - n2).
// Evaluated here with an explicit frame stack (depth <= log2(n / 128) + 1).
__device__ double np_pairwise_sum(const double* a, int n) {
    struct Fr { int off, n, state; double left; };
    Fr fr[12];
    fr[0] = {0, n, 0, 0.0};
    int sp = 1;
    double ret = 0.0;
    while (sp > 0) {
        Fr& f = fr[sp - 1];
        if (f.n <= 128) { ret = np_pairwise_leaf(a + f.off, f.n); --sp; continue; }
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.state == 0) { f.state = 1; fr[sp] = {f.off, n2, 0, 0.0}; ++sp; }
        else if (f.state == 1) { f.left = ret; f.state = 2; fr[sp] = {f.off + n2, f.n - n2, 0, 0.0}; ++sp; }
        else { ret = f.left + ret; --sp; }
    }
    return ret;
}

__global__ __launch_bounds__(64) void kmeans1d2_kernel(const double* __restrict__ values, const int* __restrict__ counts, int max_n,
                                                       int max_iters, int* __restrict__ labels, double* __restrict__ centers) {
    extern __shared__ double sm[];          // vals[max_n] | group0[max_n] | group1[max_n]
    double* vals = sm; double* g0 = sm + max_n; double* g1 = g0 + max_n;
    __shared__ int lab[2][1024];
    const int b = blockIdx.x, lane = threadIdx.x;
    int n = counts[b];
    n = n < 0 ? 0 : (n > max_n ? max_n : n);
    const double* v = values + (size_t)b * max_n;
    int* L = labels + (size_t)b * max_n;
    for (int i = lane; i < n; i += 64) { vals[i] = v[i]; lab[0][i] = 0; }
    for (int i = lane; i < max_n; i += 64) L[i] = 0;
    __syncthreads();
    if (lane != 0) return;                  // <= a few hundred values: the order-exact sums are serial anyway
    double c0, c1;
    if (n < 2) {                            // measurement.py:90-91: all zeros, both centres = mean (NaN for an empty input, as numpy)
        const double m = n ? vals[0] : __builtin_nan("");
        centers[2 * b] = m; centers[2 * b + 1] = m;
        return;
    }
    c0 = vals[0]; c1 = vals[0];
    for (int i = 1; i < n; ++i) { c0 = fmin(c0, vals[i]); c1 = fmax(c1, vals[i]); }
    int cur = 0;                            // lab[cur] = `labels`, lab[cur ^ 1] = `new_labels`
    for (int it = 0; it < max_iters; ++it) {
        int* nl = lab[cur ^ 1];
        int n1 = 0, k0 = 0, k1 = 0;
        for (int i = 0; i < n; ++i) {
            const int l = fabs(vals[i] - c1) < fabs(vals[i] - c0) ? 1 : 0;
            nl[i] = l; n1 += l;
            if (l) g1[k1++] = vals[i]; else g0[k0++] = vals[i];
        }
        if (n1 == 0 || n1 == n) break;
        const double nc0 = np_pairwise_sum(g0, k0) / (double)k0, nc1 = np_pairwise_sum(g1, k1) / (double)k1;
        if (nc0 == c0 && nc1 == c1) break;
        c0 = nc0; c1 = nc1; cur ^= 1;
    }
    for (int i = 0; i < n; ++i) L[i] = lab[cur][i];
    centers[2 * b] = c0; centers[2 * b + 1] = c1;
}

hipError_t launch_kmeans1d2(const double* values, const int* counts, int B, int max_n, int max_iters, int* labels,
                            double* centers, hipStream_t st) {
    if (B == 0) return hipSuccess;
    if (max_n < 1 || max_n > 1024) return hipErrorInvalidValue;     // LDS tables; the reference caps detections at 200 (config.py:73)
    hipLaunchKernelGGL(kmeans1d2_kernel, dim3(B), dim3(64), (size_t)max_n * 3 * sizeof(double), st, values, counts, max_n, max_iters,
                       labels, centers);
    return hipGetLastError();
}

}  // namespace vti
