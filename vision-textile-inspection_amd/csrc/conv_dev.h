// Device-side pieces shared by the conv kernels (conv.hip: per-tile kernel, conv_pk.hip: persistent
// LDS-DMA kernel): MFMA wrappers, SiLU, the shared epilogue and the fused 1x1 second stage.
#pragma once
#include "vti_internal.h"

namespace vti {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ---- h2: split-fp16 storage (VTI_H2).  One element is 4 bytes: the fp16 pair (hi | lo << 16) with hi = fp16(v * H2_SX),
// lo = fp16(v * H2_SX - hi): 22-23 significant bits in fp16 MFMA operands.  Every layout (64-byte pixel slots of 16 channels,
// 1-KiB weight fragments of 4 channels x 16 B per lane, KC = 16) is the fp32 engine's, so the data movement of every kernel is
// shared with it; what differs is the matrix instruction -- two 16x16x32 f16 MFMAs per 16 channels instead of four 16x16x4 f32
// ones (4x the rate) -- and the encode in the epilogues.  Weights are stored as pairs of w * SW (SW a per-conv power of two that
// puts max|w| in [2^13, 2^14): lo parts stay normal numbers); the epilogue multiplies the accumulator by alpha = 1 / (SW * H2_SX).
// Why not plain fp16: profiles/r03_precision_ablation.txt -- the north-star gate (mask IoU >= 0.999, |d box| < 1e-3, same kept set)
// needs >= 20 significant bits in EVERY stored tensor and weight of these networks.
struct h2_t { unsigned u; };
struct h2x4 { u32x4 u; };
constexpr float H2_SX = 16.0f;          // activation scale: |v| < 4094 representable, lo parts normal down to |v| ~ 8e-3

template <typename T> struct Tr;
template <> struct Tr<half_t> { typedef half8 vec; static constexpr int VEC = 8, KC = 32; static constexpr bool H16 = true, F32 = false, H2 = false; };
template <> struct Tr<float> { typedef f32x4 vec; static constexpr int VEC = 4, KC = 16; static constexpr bool H16 = false, F32 = true, H2 = false; };
template <> struct Tr<h2_t> { typedef h2x4 vec; static constexpr int VEC = 4, KC = 16; static constexpr bool H16 = false, F32 = false, H2 = true; };

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// Written as explicit fmas so that each half is ONE instruction (v_fma_mixlo_f16 / v_fma_mixhi_f16 with the fp16 hi as the fp16
// addend: v * 16 - hi is exact in f32, then rounded to fp16) -- the plain casts cost six (the file is compiled -ffp-contract=off).
__device__ __forceinline__ unsigned h2_enc(float v) {
    half2v p;
    p[0] = (half_t)__builtin_fmaf(v, H2_SX, 0.0f);
    p[1] = (half_t)__builtin_fmaf(v, H2_SX, -(float)p[0]);
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ float h2_dec(unsigned u) {
    const half2v p = __builtin_bit_cast(half2v, u);
    return ((float)p[0] + (float)p[1]) * (1.0f / H2_SX);      // hi + lo is exact in f32 (<= 24 significant bits)
}
// one element from a float; element j of an operand vector
template <typename T> __device__ __forceinline__ T to_T(float v) {
    if constexpr (Tr<T>::H2) return h2_t{h2_enc(v)};
    else return (T)v;
}
template <typename T> __device__ __forceinline__ void vset(typename Tr<T>::vec& x, int j, T e) {
    if constexpr (Tr<T>::H2) x.u[j] = e.u;
    else x[j] = e;
}
// 4 channels of a 4-byte element type <-> f32x4 (float: a bit cast)
template <typename T> __device__ __forceinline__ u32x4 pack4(f32x4 v) {
    if constexpr (Tr<T>::H2) return (u32x4){h2_enc(v[0]), h2_enc(v[1]), h2_enc(v[2]), h2_enc(v[3])};
    else return __builtin_bit_cast(u32x4, v);
}
template <typename T> __device__ __forceinline__ f32x4 unpack4(u32x4 u) {
    if constexpr (Tr<T>::H2) return (f32x4){h2_dec(u[0]), h2_dec(u[1]), h2_dec(u[2]), h2_dec(u[3])};
    else return __builtin_bit_cast(f32x4, u);
}
// accumulator -> pre-activation: h2 accumulators carry the weight and activation scales (alpha = 1 / (SW * H2_SX))
template <typename T> __device__ __forceinline__ f32x4 acc_bias(f32x4 acc, f32x4 bias, float alpha) {
    if constexpr (Tr<T>::H2) return acc * (f32x4){alpha, alpha, alpha, alpha} + bias;
    else return acc + bias;
}

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

// h2: the weight fragment holds (hi | lo << 16) pairs of 4 channels; the pixel fragment the same.  With WH = the hi halves
// duplicated into both halves of each dword and WL = the lo halves duplicated,
//   mfma(WH, X) = sum wh * (xh + xl),  mfma(WL, X) = sum wl * xh  with WL = (lo, 0):  fp32 accumulation; wl * xl (< 2^-22) is dropped.
__device__ __forceinline__ f32x4 mma(h2x4 w, h2x4 x, f32x4 c) {
    u32x4 wh, wl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        wh[i] = __builtin_amdgcn_perm(w.u[i], w.u[i], 0x01000100u);
        wl[i] = w.u[i] >> 16;
    }
    const half8 xv = __builtin_bit_cast(half8, x.u);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wh), xv, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, wl), xv, c, 0, 0, 0);
    return c;
}

// The MFMA loop of one K chunk for h2, over the flat (tap, pixel-tile) sequence of a wave's M x NREP register tile (the loop every
// 3x3 kernel runs; ldx(s) reads the pixel fragment of step s = tap * M + m, ldw(tap, n) the raw weight fragment of n-tile n).
// Same software pipeline as the fp16 / fp32 loops -- pixel fragments XD steps ahead, the next tap's weights one tap ahead -- plus
// what h2 needs: a raw fragment (hi | lo << 16 pairs) becomes the two MFMA operands WH = (hi, hi) / WL = (lo, 0) by 8 VALU
// instructions.  (WL faces xh only: the wl * xl products are below 2^-22 of the sum, and with zeros in half of the second MFMA's
// multipliers the chip holds a higher clock under this loop -- 120 -> 113 us on the 64 -> 64 layer at 80x80, A/B on one box.)  Done where the
// fragment is used, those 8 * NREP instructions per tap run in the open (measured: 16 perms in front of every 20 MFMAs at NREP = 2);
// here the NEXT tap's fragments are prepared one v_perm per MFMA in steps 1 .. M-1 of the current tap, where they issue in the
// matrix pipe's shadow (an MFMA blocks vector issue for 8 of its 16 cycles).
template <int NREP, int M, int TAPS, int XD, class LDX, class LDW>
__device__ __forceinline__ void h2_taps(f32x4 (&acc)[M][NREP], LDX ldx, LDW ldw) {
    constexpr int NSTEP = TAPS * M;
    constexpr unsigned SEL_H = 0x01000100u;
    h2x4 xq[XD];
    u32x4 wraw[NREP];
    u32x4 wh[2][NREP], wl[2][NREP];                     // prepared operands of the current / next tap (by tap parity)
#pragma unroll
    for (int n = 0; n < NREP; ++n) wraw[n] = ldw(0, n).u;
#pragma unroll
    for (int i = 0; i < XD - 1; ++i) xq[i] = ldx(i);
#pragma unroll
    for (int n = 0; n < NREP; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            wh[0][n][i] = __builtin_amdgcn_perm(wraw[n][i], wraw[n][i], SEL_H);
            wl[0][n][i] = wraw[n][i] >> 16;
        }
#pragma unroll
    for (int s_ = 0; s_ < NSTEP; ++s_) {
        const int tp = s_ / M, mm = s_ % M, cur = tp & 1, nxt = cur ^ 1;
#ifdef H2_EXP_NOXREAD       // experiment (tools/ variants only): how much of a step is the pixel-fragment LDS read
        if (s_ + XD - 1 < NSTEP && s_ < XD) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
#else
        if (s_ + XD - 1 < NSTEP) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
#endif
        if (mm == 0 && tp + 1 < TAPS) {
#pragma unroll
            for (int n = 0; n < NREP; ++n) wraw[n] = ldw(tp + 1, n).u;
        }
        __builtin_amdgcn_sched_barrier(0);
        const half8 xv = __builtin_bit_cast(half8, xq[s_ % XD].u);
#pragma unroll
        for (int j = 0; j < 2 * NREP; ++j) {
            const int n = j >> 1;
#ifdef H2_EXP_SKIP          // experiment (wrong results): every other tap runs without its WL MFMA = 3 MFMAs per 32 channel-taps
            if (!((j & 1) && (tp & 1)))
#endif
            acc[mm][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, (j & 1) ? wl[cur][n] : wh[cur][n]), xv, acc[mm][n], 0, 0, 0);
            const int k = (mm - 1) * 2 * NREP + j;      // this MFMA's share of the next tap's preparation
#ifdef H2_EXP_NOPERM         // experiment: next tap's operands = the raw fragments (wrong results), no v_perm
            if (mm == 1 && j == 0 && tp + 1 < TAPS) {
#pragma unroll
                for (int n2 = 0; n2 < NREP; ++n2) { wh[nxt][n2] = wraw[n2]; wl[nxt][n2] = wraw[n2]; }
            }
            if (false) {
#else
            if (mm >= 1 && k < 8 * NREP && tp + 1 < TAPS) {
#endif
                const int pn = k >> 3, r = k & 7, i = r & 3;
                if (r < 4) wh[nxt][pn][i] = __builtin_amdgcn_perm(wraw[pn][i], wraw[pn][i], SEL_H);
                else wl[nxt][pn][i] = wraw[pn][i] >> 16;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA, then one VALU: keep the interleave
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The same loop for LARGE register tiles (NREP = 5: 100 accumulator registers): h2_taps keeps two prepared operand sets of NREP
// fragments (16 NREP registers), which spills there.  Here the work of a chunk is walked n-tile-major inside a tap -- unit u = (tap, n) =
// 2 M MFMAs on one weight fragment -- so that only ONE fragment is prepared ahead (under the current unit's MFMAs) and one raw fragment
// is in flight behind it; the M pixel fragments of a tap stay in registers for its NREP units and the next tap's are read during the
// tap's last unit.
template <int NREP, int M, int TAPS, class LDX, class LDW>
__device__ __forceinline__ void h2_taps_nmajor(f32x4 (&acc)[M][NREP], LDX ldx, LDW ldw) {
    constexpr int NU = TAPS * NREP;
    h2x4 x[2][M];
    u32x4 wraw[2], wh[2], wl[2];
#pragma unroll
    for (int m = 0; m < M; ++m) x[0][m] = ldx(m);
    wraw[0] = ldw(0, 0).u;
    if (NU > 1) wraw[1] = ldw(1 / NREP, 1 % NREP).u;
#pragma unroll
    for (int i = 0; i < 4; ++i) { wh[0][i] = __builtin_amdgcn_perm(wraw[0][i], wraw[0][i], 0x01000100u); wl[0][i] = wraw[0][i] >> 16; }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int tp = u / NREP, n = u % NREP, cur = u & 1, nxt = cur ^ 1;
        if (n == NREP - 1 && tp + 1 < TAPS) {
#pragma unroll
            for (int m = 0; m < M; ++m) x[(tp + 1) & 1][m] = ldx((tp + 1) * M + m);
        }
        const u32x4 wnext = wraw[nxt];
        if (u + 2 < NU) wraw[cur] = ldw((u + 2) / NREP, (u + 2) % NREP).u;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2 * M; ++j) {
            const int m = j >> 1;
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, (j & 1) ? wl[cur] : wh[cur]),
                                                               __builtin_bit_cast(half8, x[tp & 1][m].u), acc[m][n], 0, 0, 0);
            if (j < 8 && u + 1 < NU) {
                const int i = j & 3;
                if (j < 4) wh[nxt][i] = __builtin_amdgcn_perm(wnext[i], wnext[i], 0x01000100u);
                else wl[nxt][i] = wnext[i] >> 16;
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// SiLU.  fp32 engine (parity mode): IEEE exp + division, as the CPU reference computes it.
// fp16 engine: v_exp_f32 + v_rcp_f32 (each ~1 ulp in f32, far below the fp16 rounding that follows);
// the accurate form costs ~30 VALU instructions per element and dominated the kernel's issue slots.
template <bool FAST> __device__ __forceinline__ float silu(float x) {
    if constexpr (FAST) return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
    else return x / (1.0f + expf(-x));
}

// Four values at once: the multiplies and the add as packed-f32 instructions (v_pk_mul_f32 / v_pk_add_f32), which
// halves the non-transcendental instruction count of the activation (PMC: the epilogue was VALU-issue bound at
// ~12 instructions per value).  Same arithmetic per element as silu<FAST>.
template <bool FAST> __device__ __forceinline__ f32x4 silu4(f32x4 x) {
    if constexpr (FAST) {
        const f32x4 y = x * (f32x4){-1.4426950408889634f, -1.4426950408889634f, -1.4426950408889634f, -1.4426950408889634f};
        f32x4 e;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_exp2f(y[j]);
        e = e + (f32x4){1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_rcpf(e[j]);
        return x * e;
    } else {
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = silu<false>(x[j]);
        return r;
    }
}

constexpr int MREP = 5;

// Diagnostic build (make STAMPS=1 -> libvti_stamps.so, used only by tools/): s_memtime stamps of
// workgroup phases, written by wave 0 to a buffer nothing else reads.  Compiled out of the product.
#ifdef VTI_STAMPS
// slots 0 and 12 (start / end of the workgroup) also record s_memrealtime (100 MHz) in slots 14 / 15: the in-kernel shader clock is
// (t12 - t0) / (rt15 - rt14) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
#define VTI_STAMP(i)                                                                              \
    do {                                                                                          \
        if (p.stamps && tid == 0) {                                                               \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;              \
            if ((i) == 0 || (i) == 12) {                                                          \
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
                p.stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + ((i) == 0 ? 14 : 15)] = t_; \
            }                                                                                     \
        }                                                                                         \
    } while (0)
#else
#define VTI_STAMP(i) do { } while (0)
#endif

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
    constexpr bool FAST = !Tr<T>::F32;
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
                            res_r[n] = unpack4<T>(*(const u32x4*)(rp + 4 * n));
                        }
                    }
                }
            }
            f32x4 v[NREP];
#pragma unroll
            for (int n = 0; n < NREP; ++n) {
                v[n] = acc_bias<T>(acc[m][n], bias_r[n], p.alpha);
                if (p.act) v[n] = silu4<FAST>(v[n]);
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
                    if (crun + 4 * n < p.Cout) *(u32x4*)(op + 4 * n) = pack4<T>(v[n]);
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
            f32x4 v = acc_bias<T>(acc[m][n], bias_r[n], p.alpha);
            if (p.act) v = silu4<FAST>(v);
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
                    v += unpack4<T>(*(const u32x4*)rp);
                }
            }
            const size_t o = opix * p.out_ld + p.out_coff + co;
            if (p.scalar_store) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (cout0 + j < p.Cout) {
                        if (p.out_f32) ((float*)p.out)[o + j] = v[j];
                        else if constexpr (Tr<T>::H2) ((unsigned*)p.out)[o + j] = h2_enc(v[j]);
                        else ((T*)p.out)[o + j] = (T)v[j];
                    }
                }
            } else if (p.out_f32 || Tr<T>::F32) {
                *(f32x4*)((float*)p.out + o) = v;
            } else if constexpr (Tr<T>::H2) {
                *(u32x4*)((unsigned*)p.out + o) = pack4<T>(v);
            } else {
                half4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[j];
                *(half4*)((half_t*)p.out + o) = hv;
            }
        }
    }
}

// Loads are raw buffer loads: 32-bit per-piece offsets computed once per tile, a scalar offset per chunk,
// and the hardware range check returns zeros for halo pixels outside the image (offset 0xFFFFFFFF).
template <typename V> __device__ __forceinline__ V buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return __builtin_bit_cast(V, v);
}

// ---- fused 1x1 second stage on the register tile of a 3x3 conv (WN == 1).
// Operands of the fused stage that do not depend on the tile: a persistent kernel loads them once (stage2_preload) instead of once
// per tile, where one compute wave per SIMD would wait for each of them in the open.
template <typename T, int NREP, int NREP2> struct Stage2Regs {
    using vec = typename Tr<T>::vec;
    static constexpr int KT = sizeof(T) == 2 ? (NREP + 1) / 2 : NREP;
    vec w2[KT][NREP2];
    f32x4 bias1[NREP];          // FOLD: the interior border class
    f32x4 bias2[NREP2];
};

// FOLD (convfold_kernel): opy/opx are coordinates on the 2 Hout x 2 Wout output grid and the stage-1 bias depends on the output
// pixel's border class (p.bias = [3 x 3][16 NREP] floats, weights.cpp: pack_conv_fold).
template <typename T, int NREP, int NREP2>
__device__ __forceinline__ void stage2_preload(const ConvParams& p, int lane, bool fold, Stage2Regs<T, NREP, NREP2>& r) {
    using vec = typename Tr<T>::vec;
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w2, 0, (int)(Stage2Regs<T, NREP, NREP2>::KT * p.ntiles2 * 1024), 0x00020000);
#pragma unroll
    for (int t2 = 0; t2 < Stage2Regs<T, NREP, NREP2>::KT; ++t2)
#pragma unroll
        for (int n = 0; n < NREP2; ++n) r.w2[t2][n] = buf_load16<vec>(rsW2, (unsigned)(((t2 * p.ntiles2 + n) * 64 + lane) * 16), 0u);
    const float* b1 = p.bias + (fold ? 4 * (16 * NREP) : 0) + (lane >> 4) * 4 * NREP;      // fold: class (1, 1) = interior
#pragma unroll
    for (int n = 0; n < NREP; ++n) r.bias1[n] = *(const f32x4*)(b1 + 4 * n);
    const int crun2 = p.nat2 ? (lane >> 4) * 4 : (lane >> 4) * 4 * NREP2, cstep2 = p.nat2 ? 16 : 4;
#pragma unroll
    for (int n = 0; n < NREP2; ++n) r.bias2[n] = *(const f32x4*)(p.bias2 + crun2 + cstep2 * n);
}

template <typename T, int NREP, int NREP2, bool FOLD = false, bool PRE = false>
__device__ __forceinline__ void conv_stage2(const ConvParams& p, f32x4 (&acc)[MREP][NREP], const bool (&pvalid)[MREP],
                                            const int (&opy)[MREP], const int (&opx)[MREP], int b, int lane,
                                            const Stage2Regs<T, NREP, NREP2>* pre = nullptr, bool interior = false) {
    const int OW = FOLD ? 2 * p.Wout : p.Wout;        // width of the grid the second stage stores on
    using vec = typename Tr<T>::vec;
    // ---- fused 1x1 second stage on the register tile (WN == 1: this wave holds every mid channel
    // of its 80 pixels).  silu(acc + bias) in fp16/fp32 IS the MFMA pixel operand of the next GEMM:
    // lane group g of cout tile n holds channels 16n+4g+j, and the stage-2 weights are packed with
    // exactly that K order (weights.cpp: pack_conv_stage2), so nothing moves between lanes or LDS.
    constexpr bool FAST = !Tr<T>::F32;
    constexpr int KT = sizeof(T) == 2 ? (NREP + 1) / 2 : NREP;
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.w2, 0, (int)(KT * p.ntiles2 * 1024), 0x00020000);
    f32x4 bias1[FOLD ? MREP : 1][NREP];
    if (PRE && (!FOLD || interior)) {               // (wave-uniform) every pixel of the tile takes the preloaded vector
#pragma unroll
        for (int m = 0; m < (FOLD ? MREP : 1); ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) bias1[m][n] = pre->bias1[n];
    } else if constexpr (FOLD) {
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const int cy = opy[m] == 0 ? 0 : (opy[m] == 2 * p.Hout - 1 ? 2 : 1), cx = opx[m] == 0 ? 0 : (opx[m] == OW - 1 ? 2 : 1);
            const float* bt = p.bias + (cy * 3 + cx) * (16 * NREP) + (lane >> 4) * 4 * NREP;
#pragma unroll
            for (int n = 0; n < NREP; ++n) bias1[m][n] = *(const f32x4*)(bt + 4 * n);
        }
    } else {
#pragma unroll
        for (int n = 0; n < NREP; ++n) bias1[0][n] = *(const f32x4*)(p.bias + (lane >> 4) * 4 * NREP + 4 * n);
    }
    f32x4 acc2[MREP][NREP2];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP2; ++n) acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t2 = 0; t2 < KT; ++t2) {
        vec w2[NREP2];
#pragma unroll
        for (int n = 0; n < NREP2; ++n) {
            if constexpr (PRE) w2[n] = pre->w2[t2][n];
            else w2[n] = buf_load16<vec>(rsW2, (unsigned)(((t2 * p.ntiles2 + n) * 64 + lane) * 16), 0u);
        }
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            vec x;
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int n1 = 2 * t2 + h;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (n1 < NREP) {
                        v = acc_bias<T>(acc[m][n1 < NREP ? n1 : 0], bias1[FOLD ? m : 0][n1 < NREP ? n1 : 0], p.alpha);
                        if (p.act) v = silu4<FAST>(v);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[h * 4 + j] = (T)v[j];
                }
            } else {
                f32x4 v = acc_bias<T>(acc[m][t2], bias1[FOLD ? m : 0][t2], p.alpha);
                if (p.act) v = silu4<FAST>(v);
                if constexpr (Tr<T>::H2) x.u = pack4<T>(v);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j] = v[j];
                }
            }
#pragma unroll
            for (int n = 0; n < NREP2; ++n) acc2[m][n] = mma(w2[n], x, acc2[m][n]);
        }
    }
    [[maybe_unused]] const int tid = threadIdx.x;
    VTI_STAMP(13);
    // second-stage epilogue: bias2 (+SiLU for proto.cv3 / sigmoid for class scores), T or fp32 output
    f32x4 bias2[NREP2];
    // channel of (tile n, element j): permuted rows give the lane a consecutive run of 4*NREP2 channels (16-byte fp16 pairs);
    // natural rows (nat2, fp32 outputs) give lane group g channels 16n + 4g + j
    const int crun2 = p.nat2 ? (lane >> 4) * 4 : (lane >> 4) * 4 * NREP2;
    const int cstep2 = p.nat2 ? 16 : 4;
#pragma unroll
    for (int n = 0; n < NREP2; ++n) {
        if constexpr (PRE) bias2[n] = pre->bias2[n];
        else bias2[n] = *(const f32x4*)(p.bias2 + crun2 + cstep2 * n);
    }
    if constexpr (NREP2 == 4) {
        if (p.act2 == 3) {
            // ---- box tower: DFL + dist2bbox here.  Natural channel order: tile n is side n (l,t,r,b) and lane
            // group g = lane >> 4 holds bins 4g..4g+3 of this lane's pixel; the softmax over a side's 16 bins is a 4-value
            // partial per lane plus a butterfly over the 4 lane groups (lanes l, l^16, l^32, l^48).  Lane group g then
            // stores component g of (cx, cy, w, h) * stride: 4 lanes x 4 B = the first 16 bytes of the anchor's pred row.
            const int g = lane >> 4;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                float d4[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f32x4 v = acc_bias<T>(acc2[m][n], bias2[n], p.alpha2);
                    float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    mx = fmaxf(mx, __shfl_xor(mx, 16));
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float e[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (FAST) e[j] = __builtin_amdgcn_exp2f((v[j] - mx) * 1.4426950408889634f);
                        else e[j] = expf(v[j] - mx);
                    }
                    float s = (e[0] + e[1]) + (e[2] + e[3]);
                    float w = (e[0] * (float)(4 * g) + e[1] * (float)(4 * g + 1)) + (e[2] * (float)(4 * g + 2) + e[3] * (float)(4 * g + 3));
                    s += __shfl_xor(s, 16); w += __shfl_xor(w, 16);
                    s += __shfl_xor(s, 32); w += __shfl_xor(w, 32);
                    d4[n] = w / s;
                }
                const float ax = (float)opx[m] + 0.5f, ay = (float)opy[m] + 0.5f;
                const float x1 = ax - d4[0], y1 = ay - d4[1], x2 = ax + d4[2], y2 = ay + d4[3];
                const float comp = g == 0 ? ((x1 + x2) / 2.0f) * p.dfl_stride : g == 1 ? ((y1 + y2) / 2.0f) * p.dfl_stride
                                 : g == 2 ? (x2 - x1) * p.dfl_stride : (y2 - y1) * p.dfl_stride;
                if (pvalid[m])
                    ((float*)p.out2)[((size_t)b * p.out2_bstride + (size_t)opy[m] * OW + opx[m]) * p.out2_ld + p.out2_coff + g] = comp;
            }
            return;
        }
    }
    if constexpr (sizeof(T) == 2 && NREP2 % 2 == 0) {
        // fp16 output with permuted rows (proto.cv3): the lane's 4*NREP2 channels are consecutive, so tile pairs go out as
        // 16-byte stores and the four lane groups of a pixel write 64 contiguous bytes per instruction
        if (!p.nat2 && !p.out2_f32 && !p.scalar_store2 && (p.Cout2 & 7) == 0 && ((p.out2_ld | p.out2_coff) & 7) == 0) {
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                if (!pvalid[m]) continue;
                half_t* op = (half_t*)p.out2 + ((size_t)b * p.out2_bstride + (size_t)opy[m] * OW + opx[m]) * p.out2_ld + p.out2_coff + crun2;
#pragma unroll
                for (int n = 0; n < NREP2; n += 2) {
                    if (crun2 + 4 * n >= p.Cout2) continue;
                    f32x4 v0 = acc_bias<T>(acc2[m][n], bias2[n], p.alpha2), v1 = acc_bias<T>(acc2[m][n + 1], bias2[n + 1], p.alpha2);
                    if (p.act2 == 1) { v0 = silu4<FAST>(v0); v1 = silu4<FAST>(v1); }
                    half8 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v0[j]; hv[4 + j] = (half_t)v1[j]; }
                    *(half8*)(op + 4 * n) = hv;
                }
            }
            return;
        }
    }
    const bool want_best = p.act2 == 2 && p.best != nullptr;            // wave-uniform
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        // out2_bstride = pixels per frame of the destination: Hout*Wout, or the anchor count when the towers write into pred
        const size_t pix = (size_t)b * p.out2_bstride + (size_t)opy[m] * OW + opx[m];
        const size_t o0 = pix * p.out2_ld + p.out2_coff;
        // (max score, first class that has it) of this pixel, for the NMS candidate filter (post.hip: nms_kernel reads these 8 bytes
        // instead of the nc scores): strict > over the lane's ascending channels, then the lane groups' partial results with ties going
        // to the lower class -- the same pair nms_scan_kernel derives from the stored row.  No pvalid branch around the shuffles.
        float bv = -INFINITY;
        int bc = 0x7fffffff;
        if (!want_best && !pvalid[m]) continue;
#pragma unroll
        for (int n = 0; n < NREP2; ++n) {
            const int cout0 = crun2 + cstep2 * n;
            if (cout0 >= p.Cout2) continue;
            f32x4 v = acc_bias<T>(acc2[m][n], bias2[n], p.alpha2);
            if (p.act2 == 1) v = silu4<FAST>(v);
            else if (p.act2 == 2) {         // class scores: sigmoid (exact in the fp32 parity engine, hw-rate ~1 ulp f32 in fp16)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (FAST) v[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[j] * -1.4426950408889634f));
                    else v[j] = 1.0f / (1.0f + expf(-v[j]));
                }
                if (want_best) {
                    if ((p.Cout2 & 3) == 0) {       // (wave-uniform) whole quads only: no per-class bound test
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (v[j] > bv) { bv = v[j]; bc = cout0 + j; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (cout0 + j < p.Cout2 && v[j] > bv) { bv = v[j]; bc = cout0 + j; }
                    }
                }
            }
            if (!pvalid[m]) continue;
            const size_t o = o0 + cout0;
            if (p.scalar_store2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (cout0 + j < p.Cout2) {
                        if (p.out2_f32) ((float*)p.out2)[o + j] = v[j];
                        else if constexpr (Tr<T>::H2) ((unsigned*)p.out2)[o + j] = h2_enc(v[j]);
                        else ((T*)p.out2)[o + j] = (T)v[j];
                    }
                }
            } else if (p.out2_f32 || Tr<T>::F32) {
                *(f32x4*)((float*)p.out2 + o) = v;
            } else if constexpr (Tr<T>::H2) {
                *(u32x4*)((unsigned*)p.out2 + o) = pack4<T>(v);
            } else {
                half4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[j];
                *(half4*)((half_t*)p.out2 + o) = hv;
            }
        }
        if (want_best) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ob = __shfl_xor(bv, o);
                const int oc = __shfl_xor(bc, o);
                if (ob > bv || (ob == bv && oc < bc)) { bv = ob; bc = oc; }
            }
            if (lane < 16 && pvalid[m]) *(float2*)(p.best + pix * 2) = make_float2(bv, (float)(bc == 0x7fffffff ? 0 : bc));
        }
    }
}

}  // namespace vti
