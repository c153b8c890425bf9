// Implicit-GEMM convolution for gfx950 (MI355X): NHWC activations, MFMA 16x16x32 f16 (or the
// exact-f32 16x16x4 form), fp32 accumulate, fused bias + SiLU (+ residual) epilogue.
//
// Implements the Conv-BN-SiLU / Conv2d / ConvTranspose2d(2,2) rows of the YOLOv8-seg table
//  that run behind the reference's model.predict()
// (measurement.py:208-210).
//
// Mapping to the hardware
//  * GEMM view: M = output pixels of one TH x TW tile of one frame, N = output channels,
//    K = taps x input channels, walked in chunks of KC = 32 (fp16) / 16 (fp32) channels.
//  * The MFMA "A" operand (rows) carries the WEIGHTS and the "B" operand (cols) the PIXELS, so an
//    accumulator lane ends up with 4 consecutive output channels of one pixel: the epilogue
//    stores 8 B (fp16) / 16 B (fp32) per lane straight into the NHWC tensor, no LDS transpose.
//  * Input patch (tile + halo) is staged once per chunk into LDS as 4 planes of [pixel][16 B]
//    (plane q = channels q*VEC..q*VEC+VEC-1 of the chunk): the lane->(pixel = lane&15,
//    k-group = lane>>4) operand map then reads 16 consecutive 16-B slots per k-group --
//    conflict-free ds_read_b128 -- and every tap is a constant byte offset.
//  * Weights are pre-packed on the host in fragment order (weights.cpp) and staged per chunk as a
//    straight 1-KiB-per-fragment copy.
//  * 256 threads = 4 waves; WN of them split N, 4/WN split M; each wave owns a 5 x NREP grid of
//    16x16 accumulators (80 pixels x 16*NREP channels).
#include "vti_internal.h"

namespace vti {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Tr;
template <> struct Tr<half_t> { typedef half8 vec; static constexpr int VEC = 8, KC = 32; };
template <> struct Tr<float> { typedef f32x4 vec; static constexpr int VEC = 4, KC = 16; };

__device__ __forceinline__ f32x4 mma(half8 w, half8 x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, c, 0, 0, 0);
}
// f32: lane l feeds k = 4*(l>>4)+i to the i-th 16x16x4 step on both operands, so the k order is
// a permutation shared by weights and pixels (exact f32 FMA chain either way).
__device__ __forceinline__ f32x4 mma(f32x4 w, f32x4 x, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], x[3], c, 0, 0, 0);
    return c;
}

// SiLU.  fp32 engine (parity mode): IEEE exp + division, as the CPU reference computes it.
// fp16 engine: v_exp_f32 + v_rcp_f32 (each ~1 ulp in f32, far below the fp16 rounding that follows);
// the accurate form costs ~30 VALU instructions per element and dominated the kernel's issue slots.
template <bool FAST> __device__ __forceinline__ float silu(float x) {
    if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    else return x / (1.0f + expf(-x));
}

constexpr int MREP = 5;

// Diagnostic build (make STAMPS=1 -> libvti_stamps.so, used only by tools/): s_memtime stamps of
// workgroup phases, written by wave 0 to a buffer nothing else reads.  Compiled out of the product.
#ifdef VTI_STAMPS
#define VTI_STAMP(i)                                                                              \
    do {                                                                                          \
        if (p.stamps && tid == 0) {                                                               \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;              \
        }                                                                                         \
    } while (0)
#else
#define VTI_STAMP(i) do { } while (0)
#endif

__host__ __device__ inline int patch_dim(int t, int ks, int s, int mode) { return mode == 1 ? t : (t - 1) * s + ks; }

size_t stem_lds_bytes(int TH, int TW);

size_t conv_lds_bytes(int ks, int stride, int mode, int TH, int TW, int WN, int NREP) {
    if (mode == 1) return stem_lds_bytes(TH, TW);
    const int npix = patch_dim(TH, ks, stride, mode) * patch_dim(TW, ks, stride, mode);
    const size_t plane = (size_t)((npix + 15) & ~15) * 16;
    const int taps = mode == 1 ? 1 : ks * ks;
    size_t raw = 0;
    if (mode == 1) raw = (((size_t)(2 * TH + 1) * (2 * TW + 1) * 3) + 15) & ~(size_t)15;   // u8 input patch
    return 4 * plane + (size_t)WN * NREP * taps * 1024 + raw;
}

// ---- shared epilogue: bias, SiLU, residual, stores.
// With the row permutation of weights.cpp, accumulator lane-group g = lane>>4 of a wave holds, over its
// NREP n-tiles, the 4*NREP CONSECUTIVE output channels crun .. crun+4*NREP-1 of its pixel
// (tile n, element j <-> channel crun + 4n + j).  fp16 outputs therefore go out as 16-byte stores (two tiles
// at a time) that tile whole 128-byte lines; residuals are read the same way.
template <typename T, int NREP>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x4 (&acc)[MREP][NREP], const bool (&pvalid)[MREP],
                                              const int (&opy)[MREP], const int (&opx)[MREP], int b, int nt0, int wn,
                                              int lane) {
    // Bias (and the residual of a whole pixel) are loaded up front: a load inside the store loop would make
    // every block wait on vmcnt(0), i.e. on all earlier STORES as well.
    constexpr bool FAST = sizeof(T) == 2;
    const int crun = (nt0 + wn * NREP) * 16 + (lane >> 4) * 4 * NREP;
    f32x4 bias_r[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        const int cb = crun + 4 * n < p.ntiles_n * 16 ? crun + 4 * n : 0;     // bias is padded to whole groups
        bias_r[n] = *(const f32x4*)(p.bias + cb);
    }
    if (!p.scalar_store && !p.out_f32 && !p.deconv_c) {
        // common case: T output, vector stores, plain NHWC addressing
        const bool has_res = __builtin_amdgcn_readfirstlane(p.has_res) != 0;
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            if (!pvalid[m]) continue;
            const size_t opix = ((size_t)(b * p.Hout + opy[m])) * p.Wout + opx[m];
            T* op = (T*)p.out + opix * p.out_ld + p.out_coff + crun;
            f32x4 res_r[NREP];
            if (has_res) {
                const T* rp = (const T*)p.res + opix * p.res_ld + p.res_coff + crun;
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    res_r[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (crun + 4 * n < p.Cout) {
                        if constexpr (sizeof(T) == 2) {
                            const half4 r = *(const half4*)(rp + 4 * n);
#pragma unroll
                            for (int j = 0; j < 4; ++j) res_r[n][j] = (float)r[j];
                        } else {
                            res_r[n] = *(const f32x4*)(rp + 4 * n);
                        }
                    }
                }
            }
            f32x4 v[NREP];
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
                v[n] = acc[m][n] + bias_r[n];
                if (p.act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[n][j] = silu<FAST>(v[n][j]);
                }
                if (has_res) v[n] += res_r[n];
            }
            if constexpr (sizeof(T) == 2) {
                if constexpr (NREP % 2 == 0) {     // runs start 16-B aligned: one 16-byte store per tile pair
#pragma unroll
                    for (int n = 0; n < NREP; n += 2) {
                        if (crun + 4 * n + 8 <= p.Cout) {
                            half8 hv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v[n][j]; hv[4 + j] = (half_t)v[n + 1][j]; }
                            *(half8*)(op + 4 * n) = hv;
                        } else if (crun + 4 * n + 4 <= p.Cout) {
                            half4 hv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[n][j];
                            *(half4*)(op + 4 * n) = hv;
                        }
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        if (crun + 4 * n >= p.Cout) continue;
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[n][j];
                        *(half4*)(op + 4 * n) = hv;
                    }
                }
            } else {
#pragma unroll
                for (int n = 0; n < NREP; ++n)
                    if (crun + 4 * n < p.Cout) *(f32x4*)(op + 4 * n) = v[n];
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 2 && NREP % 2 == 0) {
        if (p.deconv_c && !p.scalar_store && !p.out_f32 && !p.has_res) {
            // ConvTranspose2d(2,2): this lane's whole channel run lies in ONE (dy,dx) plane (planner keeps
            // 16*NREP | Cout), so it goes out as 16-byte stores to output pixel (2y+dy, 2x+dx)
            const int qd = crun / p.deconv_c, co = crun - qd * p.deconv_c;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if (!pvalid[m] || crun >= p.Cout) continue;
                const size_t opix = ((size_t)(b * 2 * p.Hout + 2 * opy[m] + (qd >> 1))) * (2 * p.Wout) + 2 * opx[m] + (qd & 1);
                T* op = (T*)p.out + opix * p.out_ld + p.out_coff + co;
#pragma unroll
                for (int n = 0; n < NREP; n += 2) {
                    half8 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a0 = acc[m][n][j] + bias_r[n][j], a1 = acc[m][n + 1][j] + bias_r[n + 1][j];
                        if (p.act) { a0 = silu<FAST>(a0); a1 = silu<FAST>(a1); }
                        hv[j] = (half_t)a0; hv[4 + j] = (half_t)a1;
                    }
                    *(half8*)(op + 4 * n) = hv;
                }
            }
            return;
        }
    }
    // general case: fp32 head outputs, ragged channel counts (scalar stores), ConvTranspose scatter
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        if (!pvalid[m]) continue;
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            const int cout0 = crun + 4 * n;
            if (cout0 >= p.Cout) continue;
            f32x4 v = acc[m][n] + bias_r[n];
            if (p.act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = silu<FAST>(v[j]);
            }
            size_t opix;
            int co = cout0;
            if (p.deconv_c) {
                const int q = cout0 / p.deconv_c;
                co = cout0 - q * p.deconv_c;
                opix = ((size_t)(b * 2 * p.Hout + 2 * opy[m] + (q >> 1))) * (2 * p.Wout) + 2 * opx[m] + (q & 1);
            } else {
                opix = ((size_t)(b * p.Hout + opy[m])) * p.Wout + opx[m];
            }
            if (p.has_res) {
                const T* rp = (const T*)p.res + opix * p.res_ld + p.res_coff + co;
                if constexpr (sizeof(T) == 2) {
                    const half4 r = *(const half4*)rp;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
                } else {
                    v += *(const f32x4*)rp;
                }
            }
            const size_t o = opix * p.out_ld + p.out_coff + co;
            if (p.scalar_store) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (cout0 + j < p.Cout) {
                        if (p.out_f32) ((float*)p.out)[o + j] = v[j];
                        else ((T*)p.out)[o + j] = (T)v[j];
                    }
                }
            } else if (p.out_f32 || sizeof(T) == 4) {
                *(f32x4*)((float*)p.out + o) = v;
            } else {
                half4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[j];
                *(half4*)((half_t*)p.out + o) = hv;
            }
        }
    }
}

// Register-staged operand prefetch: a thread owns up to AR input-patch pieces and BR weight pieces
// (16 B each) of a chunk.  issue() only starts the loads; commit() writes them to LDS.  The next chunk is
// issued before the MFMA loop of the current one, so HBM/L2 latency hides under MFMA.
// Loads are raw buffer loads: 32-bit per-piece offsets computed once per tile, a scalar offset per chunk,
// and the hardware range check returns zeros for halo pixels outside the image (offset 0xFFFFFFFF).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename V> __device__ __forceinline__ V buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(V, v);
}

template <typename T, int KS, int S, int NREP, int WN, int NREP2 = 0>
__global__ __launch_bounds__(256, 2) void conv_kernel(const ConvParams p) {   // 2 waves/SIMD: VGPR + AGPR <= 256
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC;
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int NTB = WN * NREP;
    constexpr int AR = (S == 2 ? 12 : 8);
    constexpr int BR = (NTB * TAPS * 64 + 255) / 256;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    const int PH = (p.TH - 1) * S + KS, PW = (p.TW - 1) * S + KS;
    const int npix = PH * PW;
    const int plane_bytes = ((npix + 15) & ~15) * 16;
    char* smA = smem;
    char* smB = smem + 4 * plane_bytes;
    const int nt0 = blockIdx.y * NTB;
    const int tile_px = p.TH * p.TW;
    VTI_STAMP(0);

    // per-lane pixel bookkeeping for the MREP pixel tiles of this wave
    int abase[MREP], opy[MREP], opx[MREP];
    bool pvalid[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = (wm * MREP + m) * 16 + (lane & 15);
        const bool v = pp < tile_px;
        const int pc = v ? pp : 0;
        const int py = (int)__umulhi((unsigned)pc, p.tw_magic), px = pc - py * p.TW;
        opy[m] = oy0 + py; opx[m] = ox0 + px;
        pvalid[m] = v && opy[m] < p.Hout && opx[m] < p.Wout;
        abase[m] = (lane >> 4) * plane_bytes + ((py * S) * PW + px * S) * 16;
    }

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- operand sources as buffer resources (wave-uniform bases)
    const size_t frame_elems = (size_t)p.Hin * p.Win * p.in_ld;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.in + (size_t)b * frame_elems), 0, (int)(frame_elems * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)p.wpk_bytes, 0x00020000);

    // item -> (pixel, 16-B channel piece q): 8 consecutive lanes take 8 consecutive pixels of ONE plane (a
    // conflict-free 128-B ds_write run); a wave instruction still covers 16 pixels x 64 B of global memory.
    // item i = tid + 256*u  ->  pixel = (tid>>5)*8 + (tid&7) + 64*u, q = (tid>>3)&3 (same q for every u).
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const int q = (tid >> 3) & 3;
    const int pix0 = (tid >> 5) * 8 + (tid & 7);
    const int ldsA0 = q * plane_bytes + pix0 * 16;          // + u * 1024 per piece
    const int cvalid = (p.Cin - q * VEC + KC - 1) / KC;     // chunks in which this lane's channel piece exists
    unsigned aoff[AR];
#pragma unroll
    for (int u = 0; u < AR; ++u) {
        const int pix = pix0 + 64 * u;
        const int py = (int)__umulhi((unsigned)pix, p.pw_magic), px = pix - py * PW;
        const int y = iy0 + py, x = ix0 + px;
        const bool ok = pix < npix && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
        aoff[u] = ok ? (unsigned)(((y * p.Win + x) * p.in_ld + p.in_coff + q * VEC) * (int)sizeof(T)) : OOB;
    }
    // NREP == 5 kernels are at the register limit: they fetch the weight pieces synchronously inside commit()
    // (short-lived registers, L2-resident data) instead of carrying them across the MFMA loop.
    constexpr bool BPRE = NREP < 5;
    vec ra[AR], rb[BR];
    auto loadB = [&](int c) {
        const unsigned sB = (unsigned)(((size_t)c * p.ntiles_n + nt0) * (TAPS * 1024));
#pragma unroll
        for (int u = 0; u < BR; ++u)
            if (tid + u * 256 < NTB * TAPS * 64) rb[u] = buf_load16<vec>(rsB, (unsigned)(tid + u * 256) * 16u, sB);
    };
    auto issue = [&](int c) {
        const bool qok = c < cvalid;
#pragma unroll
        for (int u = 0; u < AR; ++u)
            if (pix0 + 64 * u < ((npix + 7) & ~7))
                ra[u] = buf_load16<vec>(rsA, qok ? aoff[u] : OOB, (unsigned)(c * KC * (int)sizeof(T)));
        if constexpr (BPRE) loadB(c);
    };
    auto commit = [&](int c) {
        if constexpr (!BPRE) loadB(c);
#pragma unroll
        for (int u = 0; u < AR; ++u)
            if (pix0 + 64 * u < npix) *(vec*)(smA + ldsA0 + u * 1024) = ra[u];
#pragma unroll
        for (int u = 0; u < BR; ++u)
            if (tid + u * 256 < NTB * TAPS * 64) *(vec*)(smB + (tid + u * 256) * 16) = rb[u];
    };

    issue(0);
    for (int c = 0; c < p.nchunks; ++c) {
        if (c < 2) VTI_STAMP(1 + 5 * c);
        commit(c);                          // waits for this chunk's loads, fills LDS
        if (c < 2) VTI_STAMP(3 + 5 * c);
        __syncthreads();
        if (c < 2) VTI_STAMP(4 + 5 * c);
        if (c + 1 < p.nchunks) issue(c + 1);   // in flight during the MFMA loop below
        // ---- MFMA over the taps of this chunk, software pipelined over the flat (tap, pixel-tile)
        // sequence: pixel fragments are read two steps ahead and the next tap's weight fragments one
        // whole tap ahead, so LDS latency hides under the 4-5 MFMAs of each step.
        {
            constexpr int NSTEP = TAPS * MREP;
            // NREP == 5 (80-channel towers) sits at the register limit: it keeps one weight-fragment set and a
            // 2-deep pixel queue so that two waves still fit per SIMD (VGPR + AGPR <= 256).
            constexpr int XD = NREP >= 5 ? 2 : 3;          // pixel fragments in flight
            constexpr int WD = NREP >= 5 ? 1 : 2;          // weight fragment sets
            vec xq[XD];
            vec wq[WD][NREP];
            auto ldx = [&](int s_) -> vec {
                const int tp = s_ / MREP, mm = s_ % MREP;
                const int toff = ((tp / KS) * PW + (tp % KS)) * 16;
                return *(const vec*)(smA + abase[mm] + toff);
            };
            auto ldw = [&](int tp, vec (&w)[NREP]) {
#pragma unroll
                for (int n = 0; n < NREP; ++n)
                    w[n] = *(const vec*)(smB + ((wn * NREP + n) * TAPS + tp) * 1024 + lane * 16);
            };
            ldw(0, wq[0]);
            xq[0] = ldx(0);
            if (XD > 2 && NSTEP > 1) xq[1] = ldx(1);
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                const int tp = s_ / MREP, mm = s_ % MREP;
                if (s_ + XD - 1 < NSTEP) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
                if (WD == 2 && mm == 0 && tp + 1 < TAPS) ldw(tp + 1, wq[(tp + 1) % WD]);
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch reads ahead of this step's MFMAs
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[mm][n] = mma(wq[tp % WD][n], xq[s_ % XD], acc[mm][n]);
                __builtin_amdgcn_sched_barrier(0);
                if (WD == 1 && mm == MREP - 1 && tp + 1 < TAPS) ldw(tp + 1, wq[0]);   // reload after the tap's last use
            }
        }
        if (c < 2) VTI_STAMP(5 + 5 * c);
        if (c + 1 < p.nchunks) __syncthreads();   // everyone is done reading LDS before it is refilled
    }

    VTI_STAMP(11);
    if constexpr (NREP2 == 0) {
        conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, b, nt0, wn, lane);
    } else {
        // ---- fused 1x1 second stage on the register tile (WN == 1: this wave holds every mid channel
        // of its 80 pixels).  silu(acc + bias) in fp16/fp32 IS the MFMA pixel operand of the next GEMM:
        // lane group g of cout tile n holds channels 16n+4g+j, and the stage-2 weights are packed with
        // exactly that K order (weights.cpp: pack_conv_stage2), so nothing moves between lanes or LDS.
        static_assert(WN == 1, "fused stage needs the whole Cout in one wave");
        constexpr bool FAST = sizeof(T) == 2;
        constexpr int KT = sizeof(T) == 2 ? (NREP + 1) / 2 : NREP;
        const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.w2, 0, (int)(KT * p.ntiles2 * 1024), 0x00020000);
        f32x4 bias1[NREP];
#pragma unroll
        for (int n = 0; n < NREP; ++n) bias1[n] = *(const f32x4*)(p.bias + (lane >> 4) * 4 * NREP + 4 * n);
        f32x4 acc2[MREP][NREP2];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP2; ++n) acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t2 = 0; t2 < KT; ++t2) {
            vec w2[NREP2];
#pragma unroll
            for (int n = 0; n < NREP2; ++n)
                w2[n] = buf_load16<vec>(rsW2, (unsigned)(((t2 * p.ntiles2 + n) * 64 + lane) * 16), 0u);
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                vec x;
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int n1 = 2 * t2 + h;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v = 0.f;
                            if (n1 < NREP) {
                                v = acc[m][n1 < NREP ? n1 : 0][j] + bias1[n1 < NREP ? n1 : 0][j];
                                if (p.act) v = silu<FAST>(v);
                            }
                            x[h * 4 + j] = (T)v;
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[m][t2][j] + bias1[t2][j];
                        if (p.act) v = silu<FAST>(v);
                        x[j] = v;
                    }
                }
#pragma unroll
                for (int n = 0; n < NREP2; ++n) acc2[m][n] = mma(w2[n], x, acc2[m][n]);
            }
        }
        // second-stage epilogue: bias2 (+SiLU for proto.cv3), T or fp32 output
        f32x4 bias2[NREP2];
        const int crun2 = (lane >> 4) * 4 * NREP2;      // this lane's consecutive output-channel run
#pragma unroll
        for (int n = 0; n < NREP2; ++n) bias2[n] = *(const f32x4*)(p.bias2 + crun2 + 4 * n);
        if (p.pred_mode) {
            // class scores (exact sigmoid: they are compared with `conf`) / mask coefficients go straight
            // into pred [B, no, A]: channel-major, so 16 lanes (= 16 anchors) of a tile share a 64-B segment
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if (!pvalid[m]) continue;
                float* o = p.pred + ((size_t)b * p.pred_no + p.pred_cbase) * p.pred_A + p.pred_a0 + opy[m] * p.Wout + opx[m];
#pragma unroll
                for (int n = 0; n < NREP2; ++n) {
                    const f32x4 v = acc2[m][n] + bias2[n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int ch = crun2 + 4 * n + j;
                        if (ch < p.Cout2) {
                            float r = v[j];
                            if (p.pred_mode == 2) {     // sigmoid: exact in the fp32 parity engine, hw-rate (~1 ulp f32) in fp16
                                if constexpr (FAST) r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(r * -1.4426950408889634f));
                                else r = 1.0f / (1.0f + expf(-r));
                            }
                            o[(size_t)ch * p.pred_A] = r;
                        }
                    }
                }
            }
        } else
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            if (!pvalid[m]) continue;
            const size_t o0 = (((size_t)(b * p.Hout + opy[m])) * p.Wout + opx[m]) * p.out2_ld + p.out2_coff;
#pragma unroll
            for (int n = 0; n < NREP2; ++n) {
                const int cout0 = crun2 + 4 * n;
                if (cout0 >= p.Cout2) continue;
                f32x4 v = acc2[m][n] + bias2[n];
                if (p.act2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = silu<FAST>(v[j]);
                }
                const size_t o = o0 + cout0;
                if (p.scalar_store2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cout0 + j < p.Cout2) {
                            if (p.out2_f32) ((float*)p.out2)[o + j] = v[j];
                            else ((T*)p.out2)[o + j] = (T)v[j];
                        }
                    }
                } else if (p.out2_f32 || sizeof(T) == 4) {
                    *(f32x4*)((float*)p.out2 + o) = v;
                } else {
                    half4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[j];
                    *(half4*)((half_t*)p.out2 + o) = hv;
                }
            }
        }
    }
    VTI_STAMP(12);
}

// ---- stem conv (model.0): u8 HWC3 frame -> /255 -> 3x3 stride-2 conv, K = 27 padded to 32.
// The (2TH+1) x (2TW+1) x 3-byte input patch is copied to LDS with aligned dword loads; an exact
// 256-entry table gives T(v/255.0f) (what torch computes for `im.float()/255`, then .half()); every
// lane gathers its MFMA pixel-operand fragment (k = (kh*3+kw)*3 + channel) straight from the LDS
// bytes, so no im2col image is materialised.  Weights are read as fragments from L2 (1 KiB/wave).
template <typename T, int NREP>
__global__ __launch_bounds__(256) void stem_kernel(const ConvParams p) {
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC;
    constexpr int NCH = 32 / KC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* lut = (T*)smem;                                  // 256 entries, 1 KiB reserved
    unsigned* raw32 = (unsigned*)(smem + 1024);
    const unsigned char* raw = (const unsigned char*)raw32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    const int RH = 2 * p.TH + 1, RWB = (2 * p.TW + 1) * 3;
    const int RWD = (RWB + 6) >> 2;                     // dwords per patch row incl. alignment slack
    const int pitch = RWD * 4;
    const int x0b = (ox0 * 2 - 1) * 3, a0 = x0b & ~3, shift = x0b - a0;
    const int y0 = oy0 * 2 - 1;
    const uint8_t* inb = (const uint8_t*)p.in + (size_t)b * p.Hin * p.Win * 3;
    const int rowbytes = p.Win * 3;                     // multiple of 4 in the product (W % 32 == 0)
    for (int i = tid; i < RH * RWD; i += 256) {
        const int ry = (int)__umulhi((unsigned)i, p.rw_magic), rd = i - ry * RWD;
        const int y = y0 + ry, gx = a0 + 4 * rd;
        unsigned v = 0;
        if ((unsigned)y < (unsigned)p.Hin) {
            if ((rowbytes & 3) == 0) {      // aligned rows: a dword is wholly inside or outside the row
                if (gx >= 0 && gx < rowbytes) v = *(const unsigned*)(inb + (size_t)y * rowbytes + gx);
            } else {                        // ragged width (tests only): assemble byte by byte
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (gx + k >= 0 && gx + k < rowbytes) v |= (unsigned)inb[(size_t)y * rowbytes + gx + k] << (8 * k);
            }
        }
        raw32[i] = v;
    }
    lut[tid] = (T)((float)tid / 255.0f);
    __syncthreads();

    const int tile_px = p.TH * p.TW;
    const int nt0 = blockIdx.y * NREP;
    int opy[MREP], opx[MREP], rbase[MREP];
    bool pvalid[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = (wave * MREP + m) * 16 + (lane & 15);
        const bool v = pp < tile_px;
        const int pc = v ? pp : 0;
        const int py = (int)__umulhi((unsigned)pc, p.tw_magic), px = pc - py * p.TW;
        opy[m] = oy0 + py; opx[m] = ox0 + px;
        pvalid[m] = v && opy[m] < p.Hout && opx[m] < p.Wout;
        rbase[m] = (py * 2) * pitch + (px * 2) * 3 + shift;
    }
    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int off[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int k = c * KC + (lane >> 4) * VEC + j;
            const int tap = k / 3, chn = k - tap * 3;
            const int kh = tap / 3, kw = tap - kh * 3;
            off[j] = k < 27 ? kh * pitch + kw * 3 + (p.swap_rb ? 2 - chn : chn) : -1;
        }
        vec w[NREP];
#pragma unroll
        for (int n = 0; n < NREP; ++n) {
            const int nt = nt0 + n < p.ntiles_n ? nt0 + n : 0;
            w[n] = ((const vec*)p.wpk)[((size_t)c * p.ntiles_n + nt) * 64 + lane];
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            vec x;
#pragma unroll
            for (int j = 0; j < VEC; ++j) x[j] = off[j] >= 0 ? lut[raw[rbase[m] + off[j]]] : (T)0;
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = mma(w[n], x, acc[m][n]);
        }
    }
    conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, b, nt0, 0, lane);
}

size_t stem_lds_bytes(int TH, int TW) { return 1024 + (size_t)(2 * TH + 1) * ((((2 * TW + 1) * 3 + 6) >> 2) * 4); }

// Host-side check that a geometry fits the kernel's fixed register staging arrays.
bool conv_cfg_fits(int ks, int stride, int mode, int TH, int TW, int WN, int NREP) {
    if (mode == 1) return WN == 1 && (2 * TH + 1) * (((2 * TW + 1) * 3 + 6) >> 2) < 65536;
    const int AR = stride == 2 ? 12 : 8;
    const int npix = patch_dim(TH, ks, stride, mode) * patch_dim(TW, ks, stride, mode);
    if (((npix + 7) / 8) * 32 > 256 * AR) return false;     // input pieces per thread
    return ks == 1 || WN * NREP <= 5;                        // 3x3 instantiations cover WN*NREP <= 5
}

template <typename T, int KS, int S, int NREP, int WN, int NREP2 = 0>
static hipError_t launch_one(const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
    auto k = conv_kernel<T, KS, S, NREP, WN, NREP2>;
    static size_t lds_ok = 64 * 1024;
    if (lds > lds_ok) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lds_ok = 160 * 1024;
    }
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);
    return hipGetLastError();
}

// 3x3 kernels exist for WN*NREP <= 5 (the weight prefetch array is sized by it); 1x1 for every split.
template <typename T, int KS, int S>
static hipError_t launch_ks(int nrep, int wn, const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
#define VTI_L(N, W) if (nrep == N && wn == W) return launch_one<T, KS, S, N, W>(p, grid, lds, st);
    VTI_L(1, 1) VTI_L(2, 1) VTI_L(3, 1) VTI_L(4, 1) VTI_L(5, 1) VTI_L(1, 2) VTI_L(2, 2) VTI_L(1, 4)
    if constexpr (KS == 1) { VTI_L(3, 2) VTI_L(4, 2) VTI_L(5, 2) VTI_L(2, 4) VTI_L(3, 4) VTI_L(4, 4) VTI_L(5, 4) }
#undef VTI_L
    return hipErrorInvalidValue;
}

// (3x3 s1, NREP mid tiles) + fused (1x1, NREP2 out tiles): the pairs YOLOv8-seg's heads / proto need
bool conv_fusable(int nrep, int nrep2) {
    return (nrep == 2 && nrep2 == 2) || (nrep == 3 && nrep2 == 2) || (nrep == 4 && (nrep2 == 1 || nrep2 == 2 || nrep2 == 4)) ||
           (nrep == 5 && nrep2 == 5) || (nrep == 4 && nrep2 == 5);
}

template <typename T>
static hipError_t launch_fused(int nrep, int nrep2, const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
#define VTI_F(N, N2) if (nrep == N && nrep2 == N2) return launch_one<T, 3, 1, N, 1, N2>(p, grid, lds, st);
    VTI_F(2, 2) VTI_F(3, 2) VTI_F(4, 1) VTI_F(4, 2) VTI_F(4, 4) VTI_F(4, 5) VTI_F(5, 5)
#undef VTI_F
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_t(int ks, int stride, int nrep, int mode, const ConvParams& p, dim3 grid, size_t lds,
                           hipStream_t st) {
    if (p.ntiles2 > 0) {
        if (!(ks == 3 && stride == 1 && p.WN == 1 && mode == 0)) return hipErrorInvalidValue;
        return launch_fused<T>(nrep, p.ntiles2, p, grid, lds, st);
    }
    if (mode == 1) {
        switch (nrep) {
#define VTI_STEM(N) case N: hipLaunchKernelGGL((stem_kernel<T, N>), grid, dim3(256), lds, st, p); return hipGetLastError();
            VTI_STEM(1) VTI_STEM(2) VTI_STEM(3) VTI_STEM(4) VTI_STEM(5)
#undef VTI_STEM
            default: return hipErrorInvalidValue;
        }
    }
    if (ks == 1 && stride == 1) return launch_ks<T, 1, 1>(nrep, p.WN, p, grid, lds, st);
    if (ks == 3 && stride == 1) return launch_ks<T, 3, 1>(nrep, p.WN, p, grid, lds, st);
    if (ks == 3 && stride == 2) return launch_ks<T, 3, 2>(nrep, p.WN, p, grid, lds, st);
    return hipErrorInvalidValue;
}

hipError_t launch_conv(int dtype, int ks, int stride, int nrep, int mode, const ConvParams& p, size_t lds_bytes,
                       hipStream_t st) {
    const int NTB = p.WN * nrep;
    dim3 grid((unsigned)(p.B * p.tiles_y * p.tiles_x), (unsigned)((p.ntiles_n + NTB - 1) / NTB));
    if (grid.x == 0) return hipSuccess;
    // host-side shape guard: the pixel tile must fit the 4/WN waves x 5 x 16 pixels
    if (p.TH * p.TW > (4 / p.WN) * MREP * 16 || (p.WN != 1 && p.WN != 2 && p.WN != 4)) return hipErrorInvalidValue;
    if (dtype == VTI_F16) return launch_t<half_t>(ks, stride, nrep, mode, p, grid, lds_bytes, st);
    return launch_t<float>(ks, stride, nrep, mode, p, grid, lds_bytes, st);
}

}  // namespace vti
