// Implicit-GEMM convolution for gfx950 (MI355X): NHWC activations, MFMA 16x16x32 f16 (or the
// exact-f32 16x16x4 form), fp32 accumulate, fused bias + SiLU (+ residual) epilogue.
//
// This file: the per-tile kernel family (`conv_kernel`: 3x3 s1/s2, 1x1, ConvTranspose; with the register-level fused 1x1
// second stage of the head towers / proto), the standalone stem and the fused stem + layer-1 kernel.  The persistent
// LDS-DMA kernels for 3x3/s1 and 1x1 convs live in conv_pk.hip; both share conv_dev.h.  launch_conv() dispatches.
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
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_dev.h"

namespace vti {

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

// Register-staged operand prefetch: a thread owns up to AR input-patch pieces and BR weight pieces
// (16 B each) of a chunk.  issue() only starts the loads; commit() writes them to LDS.  The next chunk is
// issued before the MFMA loop of the current one, so HBM/L2 latency hides under MFMA.
// NT = 512 (fused head towers on the 80-wide maps): ONE 8-wave workgroup per CU on a 16 x 40 tile instead of two 4-wave workgroups on
// 16 x 20 tiles -- the same 2 waves per SIMD, but a staged weight chunk (45 KB for the 80-channel class tower) now serves 640 pixels
// instead of 320 (the towers' weights were re-staged from L2 once per 320 pixels: 690 MB per launch), and with half the weight pieces
// per thread even the NREP = 5 kernels can prefetch them into registers under the MFMA loop instead of fetching them in commit().
template <typename T, int KS, int S, int NREP, int WN, int NREP2 = 0, int NT = 256>
__global__ __launch_bounds__(NT, NT == 512 ? 1 : 2) void conv_kernel(const ConvParams p) {   // 2 waves/SIMD: VGPR + AGPR <= 256
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC;
    constexpr int TAPS = KS * KS, PAD = KS / 2;
    constexpr int NTB = WN * NREP;
    constexpr int PP = NT / 4;                               // pixels of the input patch per staging pass (4 lanes per pixel)
    constexpr int AR = NT == 512 ? 6 : (S == 2 ? 12 : 8);
    constexpr int BR = (NTB * TAPS * 64 + NT - 1) / NT;
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
    // item i = tid + NT*u  ->  pixel = (tid>>5)*8 + (tid&7) + (NT/4)*u, q = (tid>>3)&3 (same q for every u).
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const int q = (tid >> 3) & 3;
    const int pix0 = (tid >> 5) * 8 + (tid & 7);
    const int ldsA0 = q * plane_bytes + pix0 * 16;          // + u * PP * 16 per piece
    const int cvalid = (p.Cin - q * VEC + KC - 1) / KC;     // chunks in which this lane's channel piece exists
    unsigned aoff[AR];
#pragma unroll
    for (int u = 0; u < AR; ++u) {
        const int pix = pix0 + PP * u;
        const int py = (int)__umulhi((unsigned)pix, p.pw_magic), px = pix - py * PW;
        const int y = iy0 + py, x = ix0 + px;
        const bool ok = pix < npix && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
        aoff[u] = ok ? (unsigned)(((y * p.Win + x) * p.in_ld + p.in_coff + q * VEC) * (int)sizeof(T)) : OOB;
    }
    // NREP == 5 kernels are at the register limit: they fetch the weight pieces synchronously inside commit()
    // (short-lived registers, L2-resident data) instead of carrying them across the MFMA loop.
    constexpr bool BPRE = NREP < 5 || NT == 512;
    // Register sets of staged operands.  1x1 convs on the small maps are a chain of short K chunks (20 MFMAs each) whose cost is the
    // memory round trip per chunk: they keep TWO chunks in flight (their staging registers are few: no taps, no halo).
    constexpr int PD = (KS == 1 && BPRE) ? 2 : 1;
    vec ra[PD][AR], rb[PD][BR];
    auto loadB = [&](int c, vec (&rbs)[BR]) {
        const unsigned sB = (unsigned)(((size_t)c * p.ntiles_n + nt0) * (TAPS * 1024));
#pragma unroll
        for (int u = 0; u < BR; ++u)
            if (tid + u * NT < NTB * TAPS * 64) rbs[u] = buf_load16<vec>(rsB, (unsigned)(tid + u * NT) * 16u, sB);
    };
    auto issue = [&](int c, vec (&ras)[AR], vec (&rbs)[BR]) {
        const bool qok = c < cvalid;
#pragma unroll
        for (int u = 0; u < AR; ++u)
            if (pix0 + PP * u < ((npix + 7) & ~7))
                ras[u] = buf_load16<vec>(rsA, qok ? aoff[u] : OOB, (unsigned)(c * KC * (int)sizeof(T)));
        if constexpr (BPRE) loadB(c, rbs);
    };
    auto commit = [&](int c, vec (&ras)[AR], vec (&rbs)[BR]) {
        if constexpr (!BPRE) loadB(c, rbs);
#pragma unroll
        for (int u = 0; u < AR; ++u)
            if (pix0 + PP * u < npix) *(vec*)(smA + ldsA0 + u * (PP * 16)) = ras[u];
#pragma unroll
        for (int u = 0; u < BR; ++u)
            if (tid + u * NT < NTB * TAPS * 64) *(vec*)(smB + (tid + u * NT) * 16) = rbs[u];
    };
    // one K chunk: commit its operands (register set SET), refill that set with chunk c + PD, MFMA over the taps
    auto chunk = [&](int c, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        if (c < 2) VTI_STAMP(1 + 5 * c);
        commit(c, ra[SET], rb[SET]);        // waits for this chunk's loads, fills LDS
        if (c < 2) VTI_STAMP(3 + 5 * c);
        __syncthreads();
        if (c < 2) VTI_STAMP(4 + 5 * c);
        if (c + PD < p.nchunks) issue(c + PD, ra[SET], rb[SET]);   // in flight during the MFMA loop(s) below
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
            // (h2_taps keeps two prepared operand sets: 16 NREP registers more than the plain loop, which spills at NREP = 5 and at
            // NREP = 4 with 256 threads -- those keep the plain loop, whose operand preparation sits in front of each tap's MFMAs; the
            // register-lean h2_taps_nmajor of the persistent kernel was SLOWER here: 235 -> 286 us on the P3 class tower, the fused
            // stage's 100 + 100 accumulators spill either way and its two pixel-fragment sets add to it)
            if constexpr (Tr<T>::H2 && TAPS > 1 && (NREP <= 3 || (NREP == 4 && NT == 512))) {
                auto ldw1 = [&](int tp, int n) -> vec { return *(const vec*)(smB + ((wn * NREP + n) * TAPS + tp) * 1024 + lane * 16); };
                h2_taps<NREP, MREP, TAPS, 3>(acc, ldx, ldw1);
            } else {
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
        }
        if (c < 2) VTI_STAMP(5 + 5 * c);
        if (c + 1 < p.nchunks) __syncthreads();   // everyone is done reading LDS before it is refilled
    };

    issue(0, ra[0], rb[0]);
    if constexpr (PD == 2) {
        if (p.nchunks > 1) issue(1, ra[1], rb[1]);
        for (int c = 0; c < p.nchunks; c += 2) {
            chunk(c, std::integral_constant<int, 0>{});
            if (c + 1 < p.nchunks) chunk(c + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int c = 0; c < p.nchunks; ++c) chunk(c, std::integral_constant<int, 0>{});
    }

    VTI_STAMP(11);
    if constexpr (NREP2 == 0) {
        conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, b, nt0, wn, lane);
    } else {
        static_assert(WN == 1, "fused stage needs the whole Cout in one wave");
        conv_stage2<T, NREP, NREP2>(p, acc, pvalid, opy, opx, b, lane);
    }
    VTI_STAMP(12);
}

// ---- ConvTranspose2d(C, C, 2, 2) folded into the 3x3 conv that follows it, with that conv's fused 1x1 stage
// (proto.upsample -> proto.cv2 -> proto.cv3; algebra and weight composition: weights.cpp, pack_conv_fold).
// A workgroup owns a TH x TW tile of the LOW-resolution map (<= 80 pixels) and stages its (TH+2) x (TW+2) patch exactly as the
// 3x3 kernel does; wave w is output PHASE (py, px) = (w >> 1, w & 1): it computes the 64 mid channels of output pixels
// (2y+py, 2x+px) for all the tile's (y, x) as a 2x2 conv whose window starts at patch row py, column px -- 4 taps x C deep
// instead of 9 taps x C at 4x the pixels -- and then runs the 1x1 stage on its register tile (conv_stage2<FOLD>).
// ConvParams: in/Hin/Win/Cin = the deconv's input; Hout/Wout = Hin/Win (tile grid); wpk = [chunk][16 n-tiles: phase-major][4 taps]
// fragments; bias = [3][3][64] border-class table; w2/bias2/out2 = the fused 1x1 on the 2Hout x 2Wout grid.
size_t convfold_lds_bytes(int TH, int TW) {
    const int npix = (TH + 2) * (TW + 2);
    return 4 * (size_t)((npix + 15) & ~15) * 16 + (size_t)16 * 4 * 1024;
}
bool convfold_supported(int c_in, int c_mid, int c_out, int ntiles2) {
    return c_in == c_mid && c_mid % 16 == 0 && c_out == 64 && c_in % 16 == 0 && c_in <= 64 && (ntiles2 == 1 || ntiles2 == 2 || ntiles2 == 4);
}

// NT = 512 (the h2 plan's choice): TWO groups of four phase waves on an 8 x 20 tile share one staged weight chunk.  The composed weights are
// 64 KB per 16-channel chunk there -- 256 KB per tile for the h2 / fp32 engines, which cannot stay resident: 1.3 GB of L2 -> LDS staging per
// launch with 80-pixel tiles.  160-pixel tiles halve that; alone the kernel takes the same 272 us (staging is not its bound), beside the
// other branches of the multi-stream forward it costs them less (plan.cpp).
template <typename T, int NREP2, int NT = 256>
__global__ __launch_bounds__(NT, NT == 512 ? 1 : 2) void convfold_kernel(const ConvParams p) {
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC;
    constexpr int NREP = 4, TAPS = 4, NTB = 16, AR = 8;
    constexpr int BR = NTB * TAPS * 64 / NT;                  // weight pieces per thread and chunk (16 / 8)
    constexpr int PPP = NT / 4;                               // patch pixels per staging pass
    constexpr unsigned OOB = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, mgrp = tid >> 8;      // wave = output phase, mgrp = which 80 pixels of the tile
    const int py = wave >> 1, px = wave & 1;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int b = t / p.tiles_y;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;                // low-resolution tile origin
    const int PH = p.TH + 2, PW = p.TW + 2;
    const int npix = PH * PW;
    const int plane_bytes = ((npix + 15) & ~15) * 16;
    char* smA = smem;
    char* smB = smem + 4 * plane_bytes;
    const int tile_px = p.TH * p.TW;

    int abase[MREP], opy[MREP], opx[MREP];
    bool pvalid[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = (mgrp * MREP + m) * 16 + (lane & 15);
        const bool v = pp < tile_px;
        const int pc = v ? pp : 0;
        const int ly = (int)__umulhi((unsigned)pc, p.tw_magic), lx = pc - ly * p.TW;
        pvalid[m] = v && oy0 + ly < p.Hout && ox0 + lx < p.Wout;
        opy[m] = 2 * (oy0 + ly) + py; opx[m] = 2 * (ox0 + lx) + px;         // this wave's output pixel of low-res pixel (ly, lx)
        abase[m] = (lane >> 4) * plane_bytes + ((ly + py) * PW + lx + px) * 16;   // window origin: patch (ly + py, lx + px)
    }
    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m)
#pragma unroll
        for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const size_t frame_elems = (size_t)p.Hin * p.Win * p.in_ld;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.in + (size_t)b * frame_elems), 0, (int)(frame_elems * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)p.wpk_bytes, 0x00020000);
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int q = (tid >> 3) & 3;
    const int pix0 = (tid >> 5) * 8 + (tid & 7);
    const int ldsA0 = q * plane_bytes + pix0 * 16;
    const int cvalid = (p.Cin - q * VEC + KC - 1) / KC;
    unsigned aoff[AR];
#pragma unroll
    for (int u = 0; u < AR; ++u) {
        const int pix = pix0 + PPP * u;
        const int ry = (int)__umulhi((unsigned)pix, p.pw_magic), rx = pix - ry * PW;
        const int y = iy0 + ry, x = ix0 + rx;
        const bool ok = pix < npix && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
        aoff[u] = ok ? (unsigned)(((y * p.Win + x) * p.in_ld + p.in_coff + q * VEC) * (int)sizeof(T)) : OOB;
    }
    vec ra[AR];
    auto issue = [&](int c) {
        const bool qok = c < cvalid;
#pragma unroll
        for (int u = 0; u < AR; ++u)
            if (pix0 + PPP * u < ((npix + 7) & ~7))
                ra[u] = buf_load16<vec>(rsA, qok ? aoff[u] : OOB, (unsigned)(c * KC * (int)sizeof(T)));
    };
    // the 64 KB of weight fragments of a chunk are fetched inside commit() (short-lived registers, L2-resident data) instead of
    // being carried across the MFMA loop: 16 pieces per thread would not fit beside the accumulators at 2 workgroups per CU
    auto commit = [&](int c) {
        const unsigned sB = (unsigned)((size_t)c * (NTB * TAPS * 1024));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            vec rb[BR / 2];
#pragma unroll
            for (int u = 0; u < BR / 2; ++u) rb[u] = buf_load16<vec>(rsB, (unsigned)(tid + (h * (BR / 2) + u) * NT) * 16u, sB);
            if (h == 0) {
#pragma unroll
                for (int u = 0; u < AR; ++u)
                    if (pix0 + PPP * u < npix) *(vec*)(smA + ldsA0 + u * (PPP * 16)) = ra[u];
            }
#pragma unroll
            for (int u = 0; u < BR / 2; ++u) *(vec*)(smB + (tid + (h * (BR / 2) + u) * NT) * 16) = rb[u];
        }
    };

    issue(0);
    for (int c = 0; c < p.nchunks; ++c) {
        commit(c);
        __syncthreads();
        if (c + 1 < p.nchunks) issue(c + 1);
        {
            constexpr int NSTEP = TAPS * MREP;
            vec xq[3];
            vec wq[2][NREP];
            auto ldx = [&](int s_) -> vec {
                const int tp = s_ / MREP, mm = s_ % MREP;
                return *(const vec*)(smA + abase[mm] + ((tp >> 1) * PW + (tp & 1)) * 16);
            };
            auto ldw = [&](int tp, vec (&w)[NREP]) {
#pragma unroll
                for (int n = 0; n < NREP; ++n) w[n] = *(const vec*)(smB + ((wave * NREP + n) * TAPS + tp) * 1024 + lane * 16);
            };
            if constexpr (Tr<T>::H2) {
                auto ldw1 = [&](int tp, int n) -> vec { return *(const vec*)(smB + ((wave * NREP + n) * TAPS + tp) * 1024 + lane * 16); };
                h2_taps<NREP, MREP, TAPS, 3>(acc, ldx, ldw1);
            } else {
            ldw(0, wq[0]);
            xq[0] = ldx(0);
            xq[1] = ldx(1);
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                const int tp = s_ / MREP, mm = s_ % MREP;
                if (s_ + 2 < NSTEP) xq[(s_ + 2) % 3] = ldx(s_ + 2);
                if (mm == 0 && tp + 1 < TAPS) ldw(tp + 1, wq[(tp + 1) % 2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[mm][n] = mma(wq[tp % 2][n], xq[s_ % 3], acc[mm][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
        }
        if (c + 1 < p.nchunks) __syncthreads();
    }
    conv_stage2<T, NREP, NREP2, true>(p, acc, pvalid, opy, opx, b, lane);
}

template <typename T>
static hipError_t launch_convfold_t(const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
    if (p.nt == 512) {                  // 8 x 20 tiles, two pixel groups per phase (the plan's choice for 4-byte storage)
        if constexpr (sizeof(T) == 4) {
#define VTI_FOLD8(N2)                                                                                                  \
            if (p.ntiles2 == N2) {                                                                                     \
                auto k = convfold_kernel<T, N2, 512>;                                                                  \
                static bool done_dev[kMaxDevices] = {};                                                                \
                bool& done = done_dev[current_device_slot()];                                                          \
                if (!done) {                                                                                           \
                    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                    if (e != hipSuccess) return e;                                                                     \
                    done = true;                                                                                       \
                }                                                                                                      \
                hipLaunchKernelGGL(k, grid, dim3(512), lds, st, p);                                                    \
                return hipGetLastError();                                                                              \
            }
            VTI_FOLD8(1) VTI_FOLD8(2) VTI_FOLD8(4)
#undef VTI_FOLD8
        }
        return hipErrorInvalidValue;
    }
#define VTI_FOLD(N2)                                                                                                   \
    if (p.ntiles2 == N2) {                                                                                             \
        auto k = convfold_kernel<T, N2>;                                                                               \
        static bool done_dev[kMaxDevices] = {};                                                                        \
        bool& done = done_dev[current_device_slot()];                                                                  \
        if (!done) {                                                                                                   \
            hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) return e;                                                                             \
            done = true;                                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL(k, grid, dim3(256), lds, st, p);                                                            \
        return hipGetLastError();                                                                                      \
    }
    VTI_FOLD(1) VTI_FOLD(2) VTI_FOLD(4)
#undef VTI_FOLD
    return hipErrorInvalidValue;
}

hipError_t launch_convfold(int dtype, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    dim3 grid((unsigned)(p.B * p.tiles_y * p.tiles_x));
    if (grid.x == 0) return hipSuccess;
    // host-side shape guards for the kernel's fixed staging arrays: 80 pixels per workgroup, <= 8 patch pieces per thread
    const int nthr = p.nt == 512 ? 512 : 256;
    if (p.TH * p.TW > (nthr / 256) * MREP * 16 || ((p.TH + 2) * (p.TW + 2) + 7) / 8 * 32 > nthr * 8 || p.Cout != 256 || !p.out2 || !p.fold)
        return hipErrorInvalidValue;
    if (dtype == VTI_F16) return launch_convfold_t<half_t>(p, grid, lds_bytes, st);
    if (dtype == VTI_H2) return launch_convfold_t<h2_t>(p, grid, lds_bytes, st);
    return launch_convfold_t<float>(p, grid, lds_bytes, st);
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
    if ((rowbytes & 3) == 0 && RH * RWD <= 8 * 256) {
        // aligned rows, straight-line: all (unconditional, clamped) loads first, masks applied afterwards -- a load inside a
        // branch is waited for at the join, one memory round trip per loop iteration
        unsigned vv[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = tid + it * 256;
            const int ry = (int)__umulhi((unsigned)i, p.rw_magic), rd = i - ry * RWD;
            const int y = y0 + ry, gx = a0 + 4 * rd;
            const bool ok = i < RH * RWD && (unsigned)y < (unsigned)p.Hin && gx >= 0 && gx < rowbytes;
            vv[it] = *(const unsigned*)(inb + (ok ? (size_t)y * rowbytes + gx : (size_t)0));
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int i = tid + it * 256;
            const int ry = (int)__umulhi((unsigned)i, p.rw_magic), rd = i - ry * RWD;
            const int y = y0 + ry, gx = a0 + 4 * rd;
            const bool ok = (unsigned)y < (unsigned)p.Hin && gx >= 0 && gx < rowbytes;
            if (i < RH * RWD) raw32[i] = ok ? vv[it] : 0u;
        }
    } else
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
    lut[tid] = to_T<T>((float)tid / 255.0f);
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
            for (int j = 0; j < VEC; ++j) vset<T>(x, j, off[j] >= 0 ? lut[raw[rbase[m] + off[j]]] : to_T<T>(0.f));
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = mma(w[n], x, acc[m][n]);
        }
    }
    conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, b, nt0, 0, lane);
}

// ---- stem + layer 1 in one kernel (model.0: 3 -> 16, k3 s2; model.1: 16 -> 32, k3 s2; n-scale channel counts).
// The 320x320x16 tensor between them is the largest activation of the net (210 MB written + read per 64 frames);
// here it only ever exists as a (2TH+1) x (2TW+1) x 16 tile in LDS.
//   phase 1  u8 patch (4TH+3) x (4TW+3) x 3 -> LDS (aligned dwords), exact /255 table          [as stem_kernel]
//   phase 2  stem on the tile's (2TH+1) x (2TW+1) stem pixels (halo recompute ~6 %): per 16-pixel m-tile one K=27->32
//            MFMA, bias + SiLU, rounded to T, stored pixel-major [pixel][16 ch]; stem pixels outside the stem map are
//            layer 1's zero padding
//   phase 3  layer 1 as implicit GEMM straight from that LDS tile: Cin = 16, so one 32-deep fp16 MFMA step covers TWO taps
//            (lane group g -> tap 2s + (g >> 1), channels 8 (g & 1) ..; fp32: one tap per 16-deep step); 5 (9) steps
//   phase 4  the shared epilogue (bias, SiLU, NHWC stores) on layer 1's 8 x 40 output tile -- or, when the plan fused the 1x1
//            conv that is layer 1's only consumer (model.2.cv1), that conv on layer 1's register tile (conv_stage2): then
//            layer 1's tensor does not reach HBM either and only the third conv's output is stored
// ConvParams roles: in = u8 frames [B, Hin, Win, 3]; wpk / bias / out / Cout / Hout / Wout describe LAYER 1;
// w0 / bias0 = the stem's packed weights / bias; w2 / bias2 / out2 (non-null) = the fused third conv.
constexpr int SL_TH = 8, SL_TW = 40;       // 35 patch rows of 489 B: the u8 reads are DRAM-bound on segment length (20-wide tiles: 252 B, 2x slower staging)
constexpr int SL_SH = 2 * SL_TH + 1, SL_SW = 2 * SL_TW + 1, SL_NSP = SL_SH * SL_SW;       // stem pixels per tile
constexpr int SL_RH = 4 * SL_TH + 3, SL_RWB = (4 * SL_TW + 3) * 3, SL_RWD = (SL_RWB + 6) >> 2;
// Row pitch of the converted patch in elements.  The stem reads it as aligned 32-element windows (see phase 2): window J of a row
// starts at element 24 J, so the last window (J = 20) reaches element 511 and the pitch must cover it with FINITE values (zero
// filled past the 492 real ones: they meet zero weights); 16-byte aligned rows.  fp16: 520 (1040 B: consecutive stem rows then sit
// 32 B apart modulo the 256-B bank row, so two rows of one 16-window block do not collide); fp32: 512, what still fits 160 KiB.
constexpr int sl_pitch(int es) { return es == 2 ? 520 : 512; }
constexpr int SL_NJ = (SL_SW + 3) / 4, SL_NWIN = SL_SH * SL_NJ;       // 4-pixel windows per stem row / per tile
static_assert(24 * (SL_NJ - 1) + 32 <= 512 && SL_RWD * 4 <= 512, "stem windows must stay inside a patch row");

// h2: the u8 patch is staged as plain fp16 INTEGERS (0..255 are exact; the 1/255 lives in the stem's weights, which carry the hi/lo
// split alone: two K = 32 MFMAs per window and input row, no lo plane of x, half the patch bytes) -- see phase 2.
size_t stem_l1_lds_bytes(int dtype) {
    const size_t es = dtype == VTI_F16 ? 2 : 4, pes = dtype == VTI_F32 ? 4 : 2;
    return (((size_t)SL_RH * sl_pitch((int)pes) * pes + 15) & ~(size_t)15) + (size_t)SL_NSP * 16 * es;
}

template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 2 : 1) void stem_l1_kernel(const ConvParams p) {     // fp16: the LDS image fits twice per CU -> <= 256 registers
    using vec = typename Tr<T>::vec;
    constexpr int ES = (int)sizeof(T);
    // the patch's element type: T, except h2 -> plain fp16 holding the bytes' integer values (the stem's weights are w / 255 as hi / lo planes)
    constexpr bool XH = Tr<T>::H2;
    using PT = typename std::conditional<XH, half_t, T>::type;
    using pvec = typename Tr<PT>::vec;
    constexpr int VEC = Tr<PT>::VEC, KC = Tr<PT>::KC, PES = (int)sizeof(PT);
    constexpr int NCH = 32 / KC;                        // stem K chunks (27 -> 32)
    constexpr int NWP = XH ? 2 : 1;                     // weight planes per (window pixel, row, chunk): h2 = hi, lo
    constexpr int NS1 = sizeof(T) == 2 ? 5 : 9;         // layer-1 K steps
    constexpr bool FAST = !Tr<T>::F32;
    constexpr int SL_PITCH = sl_pitch(PES);
    constexpr int CP_BYTES = (SL_RH * SL_PITCH * PES + 15) & ~15;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PT* cp = (PT*)smem;                                 // the input patch, already v/255 in T (h2: the integer v in fp16): [row][SL_PITCH]
    char* sout = smem + CP_BYTES;                       // [stem pixel][16 ch] of T

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
    // PERSISTENT: a workgroup walks tiles blockIdx.x, + gridDim.x, ... (one or two workgroups per CU: launch_stem_l1), and the u8 patch of
    // the NEXT tile is loaded into registers at the top of phase 2, a whole tile of compute ahead of its use: the per-tile kernel
    // spent 7.5 k of its 32 k cycles per tile (h2; stamps) waiting for exactly those loads at the top of every tile.
    const int ntiles = p.B * p.tiles_y * p.tiles_x;
    if ((int)blockIdx.x >= ntiles) return;
    const int Hs = (p.Hin - 1) / 2 + 1, Ws = (p.Win - 1) / 2 + 1;       // stem map (k3 s2 p1)
    const int rowbytes = p.Win * 3;
    auto tile_geom = [&](int tt, int& b_, int& oy0_, int& ox0_) {
        const int tx = tt % p.tiles_x, r = tt / p.tiles_x;
        b_ = r / p.tiles_y; oy0_ = (r % p.tiles_y) * SL_TH; ox0_ = tx * SL_TW;
    };
    VTI_STAMP(0);
    // ---- phase 1: u8 -> T(v / 255) while staging.  fp16: v * (1/255) rounds to the same half as v / 255 for all 256 byte
    // values (checked exhaustively), so no table and no division; fp32 divides (what torch's `im.float() / 255` does).
    // wave w stages rows w, w+4, ...; a row is SL_RWD dwords = RC lane-chunks: all row arithmetic is scalar, the column test
    // is done once per chunk, and every load of the wave is issued (branch-free) before the first one is consumed
    constexpr int RC = (SL_RWD + 63) / 64, NR = (SL_RH + 3) / 4;
    unsigned vv[NR][RC];
    const bool aligned = (rowbytes & 3) == 0;
    // straight-line code: every load is unconditional (clamped address) and NOTHING consumes a result here -- a select or a branch
    // per load makes the compiler wait for each one (18 serialised round trips: 20 k cycles per workgroup, measured).  The masks
    // are re-derived where the values are consumed (top of the tile's iteration).
    auto issue_raw = [&](int tt) {
        int b_, oy0_, ox0_;
        tile_geom(tt, b_, oy0_, ox0_);
        const int iy0_ = 2 * (2 * oy0_ - 1) - 1, a0_ = ((2 * (2 * ox0_ - 1) - 1) * 3) & ~3;
        const uint8_t* inb_ = (const uint8_t*)p.in + (size_t)b_ * p.Hin * p.Win * 3;
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int ry = wave + 4 * it;
            const int y = iy0_ + ry;
            const bool row_ok = ry < SL_RH && (unsigned)y < (unsigned)p.Hin;
#pragma unroll
            for (int ch = 0; ch < RC; ++ch) {
                const int rd = lane + 64 * ch;
                const int gx = a0_ + 4 * rd;
                const bool ok = row_ok && rd < SL_RWD && gx >= 0 && gx < rowbytes;
                vv[it][ch] = *(const unsigned*)(inb_ + (ok ? (size_t)y * rowbytes + gx : (size_t)0));
            }
        }
    };
    if (aligned) issue_raw((int)blockIdx.x);
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
    int b, oy0, ox0;                                    // layer-1 output tile origin
    tile_geom(t, b, oy0, ox0);
    const int sy0 = 2 * oy0 - 1, sx0 = 2 * ox0 - 1;     // stem-map origin of the tile's stem pixels
    const int iy0 = 2 * sy0 - 1;
    const int x0b = (2 * sx0 - 1) * 3, a0 = x0b & ~3;           // x0b = 12 ox0 - 9, so x0b - a0 == 3 for every tile (phase 2 relies on it)
    const uint8_t* inb = (const uint8_t*)p.in + (size_t)b * p.Hin * p.Win * 3;
    // (per tile, not per workgroup: kept across the tile loop these ~190 registers push the kernel to one wave per SIMD for fp16 and into
    // scratch for h2; re-fetched they are L2 hits whose latency hides under the patch conversion)
    // Every weight fragment of the three convs is fetched NOW (fp16: 48 + 40 + 8 registers), so that the round trips hide under the
    // patch staging: loaded where they are used, each phase opened with an exposed L2 latency (the stem phase alone spent ~6 k of
    // its 12 k cycles per workgroup waiting for its 12 fragments).
    // the pointers are laundered through an empty asm each iteration so that the loop-invariant loads are NOT hoisted out of the tile loop
    const void* w0p = p.w0;
    const void* w1p = p.wpk;
    asm volatile("" : "+s"(w0p), "+s"(w1p));
    pvec wA[4][3][NCH * NWP];
    {
        const pvec* w0 = (const pvec*)w0p + (size_t)(p.swap_rb ? 1 : 0) * (4 * 3 * NCH * NWP * 64);
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int c = 0; c < NCH * NWP; ++c) wA[pp][kh][c] = w0[(size_t)((pp * 3 + kh) * NCH * NWP + c) * 64 + lane];
    }
    vec w1all[NS1][2];
#pragma unroll
    for (int s1 = 0; s1 < NS1; ++s1)
#pragma unroll
        for (int n = 0; n < 2; ++n) w1all[s1][n] = ((const vec*)w1p)[((size_t)s1 * 2 + n) * 64 + lane];
    const f32x4 bias0_4 = *(const f32x4*)(p.bias0 + g * 4);
    Stage2Regs<T, 2, 2> s2r;
    if (p.out2) stage2_preload<T, 2, 2>(p, lane, false, s2r);
    {
        if (aligned) {
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int ry = wave + 4 * it;
                const int y = iy0 + ry;
                const bool row_ok = ry < SL_RH && (unsigned)y < (unsigned)p.Hin;
#pragma unroll
                for (int ch = 0; ch < RC; ++ch) {
                    const int rd = lane + 64 * ch;
                    const int gx = a0 + 4 * rd;
                    const bool ok = row_ok && rd < SL_RWD && gx >= 0 && gx < rowbytes;
                    vv[it][ch] = ok ? vv[it][ch] : 0u;
                }
            }
        } else {                    // ragged width (tests only): assemble byte by byte
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int ry = wave + 4 * it;
                const int y = iy0 + ry;
                const bool row_ok = ry < SL_RH && (unsigned)y < (unsigned)p.Hin;
#pragma unroll
                for (int ch = 0; ch < RC; ++ch) {
                    const int rd = lane + 64 * ch;
                    const int gx = a0 + 4 * rd;
                    unsigned v = 0;
                    if (row_ok && rd < SL_RWD) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (gx + k >= 0 && gx + k < rowbytes) v |= (unsigned)inb[(size_t)y * rowbytes + gx + k] << (8 * k);
                    }
                    vv[it][ch] = v;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NR; ++it) {
            const int ry = wave + 4 * it;
#pragma unroll
            for (int ch = 0; ch < RC; ++ch) {
                const int rd = lane + 64 * ch;
                if (ry >= SL_RH || rd >= SL_RWD) continue;
                const unsigned v = vv[it][ch];
                PT* o = cp + (size_t)ry * SL_PITCH + 4 * rd;
                if constexpr (XH) {
                    half4 hv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hv[k] = (half_t)(float)((v >> (8 * k)) & 0xffu);      // exact
                    *(half4*)o = hv;
                } else if constexpr (sizeof(T) == 2) {
                    half4 hv;
#pragma unroll
                    for (int k = 0; k < 4; ++k) hv[k] = (half_t)((float)((v >> (8 * k)) & 0xffu) * (1.0f / 255.0f));
                    *(half4*)o = hv;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = to_T<T>((float)((v >> (8 * k)) & 0xffu) / 255.0f);
                }
            }
        }
    }
    // elements [4 SL_RWD, SL_PITCH) of every row: read by the last stem window of the row, must be finite (zero weights meet them)
    for (int i = tid; i < SL_RH * ((SL_PITCH - 4 * SL_RWD) / 4); i += 256) {
        const int ry = i / ((SL_PITCH - 4 * SL_RWD) / 4), c4 = i - ry * ((SL_PITCH - 4 * SL_RWD) / 4);
        PT* o = cp + (size_t)ry * SL_PITCH + 4 * SL_RWD + 4 * c4;
        o[0] = to_T<PT>(0.f); o[1] = to_T<PT>(0.f); o[2] = to_T<PT>(0.f); o[3] = to_T<PT>(0.f);
    }
    __syncthreads();
    VTI_STAMP(1);
    if (aligned && t + (int)gridDim.x < ntiles) issue_raw(t + (int)gridDim.x);      // next tile's patch: in flight under phases 2-4
    // ---- phase 2: stem as a Toeplitz GEMM -- no gathers.  Output pixel sx of a stem row reads patch elements 6 sx + 3 .. 6 sx + 11
    // of input rows 2 sy + kh (the tile's byte offset inside its first dword is always 3: 12 ox0 - 9), so the FOUR pixels 4J .. 4J+3
    // read inside the ALIGNED 32-element window [24 J, 24 J + 32) of each of the three rows.  One MFMA column = one window (16 windows
    // of consecutive (row, J) per block), K = the window's 32 elements (one aligned ds_read_b128 per lane and input row), and the
    // weights are expanded on the host into 4 x 3 banded fragments W'[p][kh][co][k] = w[co][kh][k - 6 p - 3] (weights.cpp:
    // pack_stem_toeplitz; both channel orders, selected by swap_rb).  12 MFMAs per 64 pixels instead of 4, but the 8 scalar LDS
    // gathers + packing per lane and m-tile that bounded this phase (15 k of ~30 k cycles per workgroup) are gone.
    {
        const f32x4 bias4 = bias0_4;
        constexpr int NBLK = (SL_NWIN + 15) / 16;
        for (int blk = wave; blk < NBLK; blk += 4) {
            const int wi = blk * 16 + (lane & 15);
            const bool wvalid = wi < SL_NWIN;
            const int wc = wvalid ? wi : SL_NWIN - 1;
            // rows fastest: the 16 windows of a block are 16 different stem rows, whose output pixels sy * 81 + 4 J + p fall on 4
            // different bank groups of the [pixel][32 B] tile (consecutive J of one row would all hit the same one: 16-way conflict)
            const int J = wc / SL_SH, sy = wc - J * SL_SH;
            const PT* base = cp + (size_t)(2 * sy) * SL_PITCH + 24 * J + g * VEC;
            pvec x[3][NCH];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int c = 0; c < NCH; ++c) x[kh][c] = *(const pvec*)(base + kh * SL_PITCH + c * KC);
            f32x4 acc[4];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                acc[pp] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int wp = 0; wp < NWP; ++wp) acc[pp] = mma(wA[pp][kh][c * NWP + wp], x[kh][c], acc[pp]);   // h2: (wh + wl) . v
            }
            const bool row_in = (unsigned)(sy0 + sy) < (unsigned)Hs;
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const int sx = 4 * J + pp;
                f32x4 v = silu4<FAST>(acc_bias<T>(acc[pp], bias4, p.alpha0));
                if (!(row_in && (unsigned)(sx0 + sx) < (unsigned)Ws)) v = (f32x4){0.f, 0.f, 0.f, 0.f};     // layer 1's zero padding
                if (wvalid && sx < SL_SW) {
                    T* o = (T*)(sout + ((size_t)(sy * SL_SW + sx) * 16 + g * 4) * ES);
                    if constexpr (sizeof(T) == 2) {
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[j];
                        *(half4*)o = hv;
                    } else {
                        *(u32x4*)o = pack4<T>(v);
                    }
                }
            }
        }
    }
    __syncthreads();
    VTI_STAMP(2);
    // ---- phase 3: layer 1 from the LDS tile
    int opy[MREP], opx[MREP], sbase[MREP];
    bool pvalid[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = (wave * MREP + m) * 16 + (lane & 15);           // < 320 = TH * TW
        const int py = pp / SL_TW, px = pp - py * SL_TW;
        opy[m] = oy0 + py; opx[m] = ox0 + px;
        pvalid[m] = opy[m] < p.Hout && opx[m] < p.Wout;
        sbase[m] = ((2 * py) * SL_SW + 2 * px) * 16 * ES;
    }
    f32x4 acc[MREP][2];
#pragma unroll
    for (int m = 0; m < MREP; ++m) { acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < NS1; ++s) {
        const int tap = sizeof(T) == 2 ? 2 * s + (g >> 1) : s;
        const int tc = tap < 9 ? tap : 8;                               // the 10th half-step multiplies by zero weights
        const int toff = ((tc / 3) * SL_SW + (tc % 3)) * 16 * ES + (sizeof(T) == 2 ? (g & 1) * 8 : g * 4) * ES;
        vec w1[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) w1[n] = w1all[s][n];
#pragma unroll
        for (int m = 0; m < MREP; ++m) {
            const vec x = *(const vec*)(sout + sbase[m] + toff);
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = mma(w1[n], x, acc[m][n]);
        }
    }
    VTI_STAMP(3);
    // ---- phase 4: layer 1's epilogue, or (model.2.cv1 fused) the 1x1 conv on layer 1's register tile
    if (p.out2) conv_stage2<T, 2, 2, false, true>(p, acc, pvalid, opy, opx, b, lane, &s2r);
    else conv_epilogue<T, 2>(p, acc, pvalid, opy, opx, b, 0, 0, lane);
    VTI_STAMP(12);
    }       // tiles of this workgroup
}

void stem_l1_tile(int* th, int* tw) { *th = SL_TH; *tw = SL_TW; }

// persistent grid: as many workgroups as fit the chip at once (one per CU; two where the LDS image fits twice: fp16)
int stem_l1_grid(int dtype, int ntiles) {
    const int wgpc = (int)std::min<size_t>(2, (160 * 1024) / stem_l1_lds_bytes(dtype));
    const char* np = getenv("VTI_STEM_NOT_PERSISTENT");          // A/B aid: one workgroup per tile, as before
    if (np && np[0] == '1') return ntiles;
    return std::min(ntiles, 256 * std::max(1, wgpc));
}

hipError_t launch_stem_l1(int dtype, const ConvParams& p, hipStream_t st) {
    dim3 grid((unsigned)stem_l1_grid(dtype, p.B * p.tiles_y * p.tiles_x));
    if (grid.x == 0) return hipSuccess;
    const size_t lds = stem_l1_lds_bytes(dtype);
    if (dtype == VTI_F16) {
        static bool attr16_dev[kMaxDevices] = {};
        bool& attr16 = attr16_dev[current_device_slot()];
        if (!attr16) {
            hipError_t e = hipFuncSetAttribute((const void*)stem_l1_kernel<half_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr16 = true;
        }
        hipLaunchKernelGGL(stem_l1_kernel<half_t>, grid, dim3(256), lds, st, p);
    } else if (dtype == VTI_H2) {
        static bool attrh2_dev[kMaxDevices] = {};
        bool& attrh2 = attrh2_dev[current_device_slot()];
        if (!attrh2) {
            hipError_t e = hipFuncSetAttribute((const void*)stem_l1_kernel<h2_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attrh2 = true;
        }
        hipLaunchKernelGGL(stem_l1_kernel<h2_t>, grid, dim3(256), lds, st, p);
    } else {
        static bool attr_done_dev[kMaxDevices] = {};
        bool& attr_done = attr_done_dev[current_device_slot()];
        if (!attr_done) {
            hipError_t e = hipFuncSetAttribute((const void*)stem_l1_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_done = true;
        }
        hipLaunchKernelGGL(stem_l1_kernel<float>, grid, dim3(256), lds, st, p);
    }
    return hipGetLastError();
}

size_t stem_lds_bytes(int TH, int TW) { return 1024 + (size_t)(2 * TH + 1) * ((((2 * TW + 1) * 3 + 6) >> 2) * 4); }

// Host-side check that a geometry fits the kernel's fixed register staging arrays.
bool conv_cfg_fits(int ks, int stride, int mode, int TH, int TW, int WN, int NREP, int threads) {
    if (mode == 1) return WN == 1 && (2 * TH + 1) * (((2 * TW + 1) * 3 + 6) >> 2) < 65536;
    if (threads != 256 && !(threads == 512 && ks == 3 && stride == 1 && WN == 1)) return false;
    const int AR = threads == 512 ? 6 : (stride == 2 ? 12 : 8);
    const int npix = patch_dim(TH, ks, stride, mode) * patch_dim(TW, ks, stride, mode);
    if (((npix + 7) / 8) * 32 > threads * AR) return false;     // input pieces per thread
    return ks == 1 || WN * NREP <= 5;                        // 3x3 instantiations cover WN*NREP <= 5
}

template <typename T, int KS, int S, int NREP, int WN, int NREP2 = 0, int NT = 256>
static hipError_t launch_one(const ConvParams& p, dim3 grid, size_t lds, hipStream_t st) {
    auto k = conv_kernel<T, KS, S, NREP, WN, NREP2, NT>;
    static size_t lds_ok_dev[kMaxDevices] = {};
    size_t& lds_ok = lds_ok_dev[current_device_slot()];
    if (lds > 64 * 1024 && lds > lds_ok) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        lds_ok = 160 * 1024;
    }
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, st, p);
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
#define VTI_F(N, N2) if (nrep == N && nrep2 == N2) return p.nt == 512 ? launch_one<T, 3, 1, N, 1, N2, 512>(p, grid, lds, st) : launch_one<T, 3, 1, N, 1, N2>(p, grid, lds, st);
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
    if (p.pk == 4) return launch_conv_pk2(dtype, nrep, p, lds_bytes, st);
    if (p.pk == 3) return launch_bneck_pk(dtype, nrep, p, lds_bytes, st);
    if (p.pk == 2) return launch_conv1_pk(dtype, nrep, p, lds_bytes, st);
    if (p.pk) return launch_conv_pk(dtype, nrep, p, lds_bytes, st);
    const int NTB = p.WN * nrep;
    dim3 grid((unsigned)(p.B * p.tiles_y * p.tiles_x), (unsigned)((p.ntiles_n + NTB - 1) / NTB));
    if (grid.x == 0) return hipSuccess;
    // host-side shape guard: the pixel tile must fit the 4/WN waves x 5 x 16 pixels
    const int nwaves = p.nt == 512 ? 8 : 4;
    if ((p.nt != 256 && p.nt != 512) || (p.nt == 512 && p.ntiles2 == 0)) return hipErrorInvalidValue;     // 512 threads: fused towers only
    if (p.TH * p.TW > (nwaves / p.WN) * MREP * 16 || (p.WN != 1 && p.WN != 2 && p.WN != 4)) return hipErrorInvalidValue;
    if (dtype == VTI_F16) return launch_t<half_t>(ks, stride, nrep, mode, p, grid, lds_bytes, st);
    if (dtype == VTI_H2) return launch_t<h2_t>(ks, stride, nrep, mode, p, grid, lds_bytes, st);
    return launch_t<float>(ks, stride, nrep, mode, p, grid, lds_bytes, st);
}

}  // namespace vti
