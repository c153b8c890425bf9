// Persistent implicit-GEMM convolutions for gfx950 (MI355X) with LDS-DMA operand prefetch: conv3_pk (3x3 / stride 1, described
// first) and conv1_pk (1x1, further down).
//
// Same GEMM view, MFMA operand roles, weight packing and epilogue contract as conv.hip (Conv-BN-SiLU rows of
// the YOLOv8-seg table, SURVEY.md section 8 U2-U5, behind measurement.py:208-210), different schedule:
//  * One workgroup per CU slot walks a strided list of output tiles (XCD k owns a contiguous range of tiles, so
//    neighbouring tiles -- which share halo rows -- meet in one L2).  Nothing is re-derived per tile except the
//    per-lane source offsets of its input patch.
//  * A workgroup is (TH/4) * WN COMPUTE waves plus as many LOADER waves.  Loader waves issue every operand load as
//    `buffer_load_dwordx4 ... lds` (no VGPR staging): while the compute waves run the MFMAs of step s, the patch
//    (and, for K > 2 chunks, the weight chunk) of step s+1 is in flight into the other stage buffer; a step is one
//    (tile, 32-channel chunk) pair and the chain crosses tile seams, so a tile's first chunk loads under the previous
//    tile's MFMAs and epilogue.  Out-of-image halo pixels are out-of-range buffer offsets: the hardware writes zeros.
//    One raw s_barrier per step: loaders arrive after `s_waitcnt vmcnt(0)` on their own DMA, compute waves after the
//    step's MFMAs (so the stage about to be refilled has no readers left).  Compute waves never wait on vmcnt: their
//    stores stay in flight.  (Issuing a 1-KiB DMA piece costs the issuing wave ~130 cycles; with one compute wave per
//    SIMD that cost has to live in other waves.)
//  * Weights of a conv with K <= 2 chunks -- or of any conv whose planner found LDS room for all chunks of a workgroup's n-group
//    (pk_wstat: typically after splitting N across workgroups) -- are loaded ONCE per workgroup and stay in LDS for all its tiles.
//  * LDS patch image: pixel-major, one 64-byte slot per pixel and chunk, 24 slots per patch row (20-wide tiles
//    + halo, padded to a multiple of 8).  A DMA wave-instruction fills 16 consecutive slots from 16 pixels x 64
//    contiguous bytes of global memory (4 lanes per pixel: whole 64-B segments, not 16-B fragments).  The 16-byte
//    piece j of slot s holds channel piece j ^ (2 * bit2(s)) -- the swizzle goes on the per-lane SOURCE address,
//    the LDS image itself is lane-linear -- which makes the MFMA pixel-operand reads (16 consecutive slots per
//    k-group, ds_read_b128) conflict-free for every tap: the row pitch is a multiple of 8 slots, so bit 2 of a
//    slot index only depends on the tap's dx, and each lane keeps three precomputed addresses (dx = 0,1,2) with
//    dy as an immediate offset.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_dev.h"

namespace vti {

constexpr int PK_PWP = 24;        // slots per patch row
constexpr int PK_TW = 20;         // tile width (pixels)
constexpr int PK_ROWS = 4;        // tile rows per M-wave: 4 x 20 = 80 pixels = MREP m-tiles
constexpr int PK_MAXD = 9;        // patch DMA instructions per wave and step (host checks)
constexpr int PK_MAXD2 = 13;      // ... of the stride-2 kernel (its patch is ~4x the output tile)

// `depth` patch stages (2..4): the loaders run depth - 1 steps ahead.  Weights: K <= 2 chunks stay resident (1 or 2 buffers);
// more chunks travel with the patches, one buffer per stage.
static size_t pk_stage_bytes(int TH, int S) { return S == 2 ? (size_t)(2 * TH + 1) * 2 * PK_PWP * 64 : (size_t)(TH + 2) * PK_PWP * 64; }
// wstat: the weights of ALL K chunks of the workgroup's n-group stay in LDS (any chunk count; K <= 2 chunks are always resident)
size_t conv_pk2_lds_bytes(int TH, int WN, int NREP, int nchunks, int depth, int wstat) {     // stride 2
    const int nwbuf = wstat ? nchunks : nchunks > 2 ? depth : (nchunks > 1 ? 2 : 1);
    return depth * pk_stage_bytes(TH, 2) + (size_t)nwbuf * WN * NREP * 9 * 1024 + (size_t)WN * NREP * 16 * 4;
}
bool conv_pk2_instantiated(int nrep, int wn) { return (nrep == 1 && wn == 4) || (nrep == 2 && wn == 2) || (nrep == 4 && wn == 1) || (nrep == 2 && wn == 1) || (nrep == 1 && wn == 2); }
bool conv_pk2_fits(int TH, int WN, int NREP, int nchunks, int wstat) {
    if (TH % PK_ROWS || !conv_pk2_instantiated(NREP, WN)) return false;
    const int ncomp = (TH / PK_ROWS) * WN;
    if (ncomp < 1 || ncomp > 4) return false;
    const int ndma = (int)(pk_stage_bytes(TH, 2) / 1024);
    if ((ndma + ncomp - 1) / ncomp > PK_MAXD2) return false;
    return conv_pk2_lds_bytes(TH, WN, NREP, nchunks, 2, wstat) <= 160 * 1024;
}
int conv_pk2_depth(int TH, int WN, int NREP, int nchunks, int wstat) {
    if (!conv_pk2_fits(TH, WN, NREP, nchunks, wstat)) return 0;
    // default 2: in an A/B on one box the deeper rings made the whole forward ~0.7 % SLOWER (VTI_PK_DEPTH=3/4 to re-measure)
    const char* cap = getenv("VTI_PK_DEPTH");
    const int maxd = cap ? std::max(2, std::min(4, atoi(cap))) : 2;
    const int ncomp = (TH / PK_ROWS) * WN;
    const int per_step = ((int)(pk_stage_bytes(TH, 2) / 1024) + ncomp - 1) / ncomp + (nchunks > 2 && !wstat ? (WN * NREP * 9 + ncomp - 1) / ncomp : 0);
    int d = 2;
    while (d < maxd && conv_pk2_lds_bytes(TH, WN, NREP, nchunks, d + 1, wstat) <= 160 * 1024 && per_step * (d - 1) <= 63) ++d;
    return d;
}

size_t conv_pk_lds_bytes(int TH, int WN, int NREP, int nchunks, int depth, int wstat) {
    const size_t stage = (size_t)(TH + 2) * PK_PWP * 64;
    const int nwbuf = wstat ? nchunks : nchunks > 2 ? depth : (nchunks > 1 ? 2 : 1);
    return depth * stage + (size_t)nwbuf * WN * NREP * 9 * 1024 + (size_t)WN * NREP * 16 * 4;
}
size_t conv_pk_lds_bytes(int TH, int WN, int NREP, int nchunks) { return conv_pk_lds_bytes(TH, WN, NREP, nchunks, 2, 0); }

// deepest ring (<= 4) that fits the 160 KiB of LDS and the counted-wait range; 0 = the geometry does not fit at all
int conv_pk_depth(int TH, int WN, int NREP, int nchunks, int wstat) {
    if (!conv_pk_fits(TH, WN, NREP, nchunks, wstat)) return 0;
    // default 2: in an A/B on one box the deeper rings made the whole forward ~0.7 % SLOWER (VTI_PK_DEPTH=3/4 to re-measure)
    const char* cap = getenv("VTI_PK_DEPTH");
    const int maxd = cap ? std::max(2, std::min(4, atoi(cap))) : 2;
    const int ncomp = (TH / PK_ROWS) * WN;
    const int per_step = ((TH + 2) * PK_PWP / 16 + ncomp - 1) / ncomp + (nchunks > 2 && !wstat ? (WN * NREP * 9 + ncomp - 1) / ncomp : 0);
    int d = 2;
    while (d < maxd && conv_pk_lds_bytes(TH, WN, NREP, nchunks, d + 1, wstat) <= 160 * 1024 && per_step * (d - 1) <= 63) ++d;
    return d;
}

bool conv_pk_fits(int TH, int WN, int NREP, int nchunks, int wstat) {
    if (TH % PK_ROWS) return false;
    const int ncomp = (TH / PK_ROWS) * WN;        // compute waves; as many loader waves beside them
    if (ncomp < 1 || ncomp > 4) return false;
    const int ndma = (TH + 2) * PK_PWP / 16;
    if ((ndma + ncomp - 1) / ncomp > PK_MAXD) return false;
    return conv_pk_lds_bytes(TH, WN, NREP, nchunks, 2, wstat) <= 160 * 1024;
}

// LDS-DMA: 64 lanes x 16 B from per-lane buffer offsets to LDS [lds_addr, lds_addr + 1 KiB), lane-linear.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }
// counted wait with a run-time count: waits until at most min(n, N) of the wave's youngest vector-memory operations are pending
template <int N> struct WaitVm { static __device__ __forceinline__ void go(int n) { if (n >= N) wait_vm<N>(); else WaitVm<N - 1>::go(n); } };
template <> struct WaitVm<0> { static __device__ __forceinline__ void go(int) { wait_vm<0>(); } };

// De-phase the two halves of a persistent 3x3 launch: the compute waves of the upper half of every XCD's workgroups (blockIdx.x >> 3
// in the upper half of its range) start `units` x 1024 cycles late (~10 k cycles: most of a tile of the small-channel layers), the
// loader waves at once.  Measured on the whole forward (bs 64, h2), A/B on one box at a time: -2.4 ... -4.7 % on the eight boxes whose
// default forward took 3.70-3.89 ms, +0.6 ... 0.8 % on the three that took 3.51-3.57 ms.  No single kernel gets faster (a hot loop of one
// layer is 0-4 % slower staggered, and subsets of the launches give nothing): the chip does once the forward's MFMA-dense launches stop
// moving all CUs in lockstep -- DESIGN.md section 7 has the experiments.  Only launches that fill the chip are staggered
// (vti_api.cpp: fill_conv_params, stagger_units); VTI_PK_STAGGER=0 turns it off.
__device__ __forceinline__ void pk_stagger_wait(int units) {
    if (units > 0 && (int)(blockIdx.x >> 3) * 2 >= (int)(gridDim.x >> 3))
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(16);
}

// FOLD: the ConvTranspose2d(2,2) -> 3x3 fold of conv.hip's convfold_kernel on this schedule: TH = 4, WN = 4 -- compute wave wn is
// output phase (py, px) = (wn >> 1, wn & 1) and runs the 2x2 window that starts at patch (py, px) over the SAME 4 x 20 low-resolution
// pixels as its three siblings (4 taps instead of 9; both K chunks of the 128 KB of composed weights stay in LDS for the whole
// launch, where the per-tile kernel re-stages 64 KB per chunk and tile); its epilogue is the fused 1x1 stage on the 2x grid.
// S = 2: the stride-2 3x3 convs (model.3/5/7/16/19) on the same schedule.  The patch is (2 TH + 1) x 41 input pixels; in LDS a patch
// row keeps its EVEN columns in slots 0..20 and its ODD columns in slots 24..43 (row pitch 48 slots), so the 16 consecutive output
// pixels of an MFMA operand read 16 consecutive slots for every tap (dx = 0: even plane at px, 1: odd plane at px, 2: even plane at
// px + 1) -- conflict-free with the same source-side swizzle as stride 1 (an interleaved image would put them 128 B apart: 8-way).
template <typename T, int NREP, int WN, int NREP2 = 0, bool FOLD = false, int S = 1>
__global__ __launch_bounds__(512) void conv3_pk(const ConvParams p) {
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC, ES = (int)sizeof(T);
    constexpr int TAPS = FOLD ? 4 : 9, NTB = WN * NREP;
    constexpr int PWP = S == 2 ? 2 * PK_PWP : PK_PWP;       // slots per patch row
    constexpr int MAXD = S == 2 ? PK_MAXD2 : PK_MAXD;
    static_assert(S == 1 || (S == 2 && !FOLD && NREP2 == 0), "stride 2: plain epilogue only");
    constexpr int WCHUNK = NTB * TAPS * 1024;
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool FAST = !Tr<T>::F32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(!FOLD || (WN == 4 && NREP2 > 0), "fold: one compute wave per output phase, fused 1x1 epilogue");
    const int ncomp = (p.TH / PK_ROWS) * WN;                // compute waves; the other blockDim/64 - ncomp waves load
    const int nld = (int)(blockDim.x >> 6) - ncomp;
    const int PH = S == 2 ? 2 * p.TH + 1 : p.TH + 2;
    const int stage_bytes = PH * PWP * 64;
    const int ndma = PH * PWP / 16;
    const int D = p.pk_depth;                               // patch stages (ring depth)
    const bool stream_w = p.nchunks > 2 && !p.pk_wstat;        // pk_wstat: all chunks of this n-group resident (the planner found room)
    const int wbuf_off = D * stage_bytes;
    const int bias_off = wbuf_off + (stream_w ? D : p.nchunks) * WCHUNK;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nt0 = blockIdx.y * NTB;

    // bias of this workgroup's channels -> LDS (read back as 16-B pieces in the epilogue)
    for (int i = tid; i < NTB * 16; i += (int)blockDim.x) ((float*)(smem + bias_off))[i] = p.bias[nt0 * 16 + i];
    __syncthreads();

    // ---- this workgroup's tiles: XCD k (= blockIdx.x & 7) owns tiles [k * per, (k+1) * per)
    int t, tend, tstride;
    if (p.pk_xcd) {
        const int per = (p.pk_tiles + 7) >> 3, k = blockIdx.x & 7;
        t = k * per + (int)(blockIdx.x >> 3);
        tend = min((k + 1) * per, p.pk_tiles);
        tstride = (int)(gridDim.x >> 3);
    } else {
        t = blockIdx.x; tend = p.pk_tiles; tstride = (int)gridDim.x;
    }
    if (t >= tend) return;
    auto tile_coords = [&](int tt, int& b, int& oy0, int& ox0) {
        const int tx = tt % p.tiles_x, r = tt / p.tiles_x;
        const int ty = r % p.tiles_y;
        b = r / p.tiles_y; oy0 = ty * p.TH; ox0 = tx * PK_TW;
    };
    const int nsteps = ((tend - t + tstride - 1) / tstride) * p.nchunks;

    if (wave >= ncomp) {
        // =================== loader waves: every LDS-DMA of the workgroup ===================
        // Step s = (tile, chunk) pair, stage s % D.  The loaders run D - 1 steps ahead: DMA(s + D - 1) is issued right after
        // barrier(s) -- the compute waves finished reading that stage (step s - 1) before they arrived there -- and before
        // barrier(s) they wait with a COUNTED vmcnt for everything up to DMA(s) while the younger steps stay in flight
        // (every loader wave issues the same number of DMA instructions in every step).  With steps of ~1.3 k MFMA cycles
        // and ~2-4 k cycles of loaded memory latency a single step of look-ahead leaves the compute waves waiting at every barrier.
        const int lw = wave - ncomp;
        // piece i = lw + u * nld fills slots 16i .. 16i+15; lane l -> slot 16i + (l >> 2), 16-B position l & 3,
        // which holds channel piece q (source-side swizzle)
        const int q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const int cvalid = (p.Cin - q * VEC + KC - 1) / KC;         // chunks in which this lane's channel piece exists
        int dyx[MAXD];
#pragma unroll
        for (int u = 0; u < MAXD; ++u) {
            const int s = (lw + u * nld) * 16 + (lane >> 2);
            if constexpr (S == 2) {
                const int py = (int)(((unsigned)s * 1366u) >> 16), r = s - py * PWP;     // s / 48 for s < 4096
                const int plane = r >= PK_PWP ? 1 : 0, idx = r - plane * PK_PWP;           // even columns first, then the odd ones
                dyx[u] = idx < PK_TW + 1 - plane ? (py << 8) | (2 * idx + plane) : -1;
            } else {
                const int py = (int)(((unsigned)s * 2731u) >> 16), px = s - py * PK_PWP;    // s / 24 for s < 4096
                dyx[u] = px < PK_TW + 2 ? (py << 8) | px : -1;
            }
        }
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)p.wpk_bytes, 0x00020000);
        unsigned voff[MAXD];
        auto setup_voff = [&](int tt) {
            int b, oy0, ox0;
            tile_coords(tt, b, oy0, ox0);
#pragma unroll
            for (int u = 0; u < MAXD; ++u) {
                const int y = S * oy0 - 1 + (dyx[u] >> 8), x = S * ox0 - 1 + (dyx[u] & 255);
                const bool ok = dyx[u] >= 0 && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
                voff[u] = ok ? (unsigned)((((b * p.Hin + y) * p.Win + x) * p.in_ld + p.in_coff + q * VEC) * ES) : OOB;
            }
        };
        auto issue_patch = [&](int c, int stage) {
            const bool qok = c < cvalid;
            const unsigned dst = lds0 + stage * stage_bytes + lw * 1024;
#pragma unroll
            for (int u = 0; u < MAXD; ++u)
                if (lw + u * nld < ndma)
                    dma16(rsA, qok ? voff[u] : OOB, (unsigned)(c * KC * ES), dst + u * nld * 1024);
        };
        auto issue_weights = [&](int c, int wb) {
            const unsigned src = (unsigned)(((size_t)c * p.ntiles_n + nt0) * (TAPS * 1024));
            const unsigned dst = lds0 + wbuf_off + wb * WCHUNK;
            for (int f = lw; f < NTB * TAPS; f += nld) dma16(rsB, (unsigned)lane * 16u, src + f * 1024, dst + f * 1024);
        };
        if (!stream_w) issue_weights(0, 0);                 // resident weights of chunk 0 first: the oldest operations
        int per_step = 0;
#pragma unroll
        for (int u = 0; u < MAXD; ++u) per_step += (lw + u * nld < ndma) ? 1 : 0;
        if (stream_w && lw < NTB * TAPS) per_step += (NTB * TAPS - lw + nld - 1) / nld;
        int it = t, ic = 0;                                 // (tile, chunk) of the next step to issue
        setup_voff(t);
        auto issue_next = [&](int s) {
            const int stage = s % D;
            issue_patch(ic, stage);
            if (stream_w) issue_weights(ic, stage);
            if (++ic == p.nchunks) {
                ic = 0; it += tstride;
                if (it < tend) setup_voff(it);
            }
        };
        const int ahead = min(D - 1, nsteps);
        for (int s = 0; s < ahead; ++s) issue_next(s);
        // chunk 1's resident weights go out AFTER the first patches: step 0 starts as soon as chunk 0's weights and patch(0) are in,
        // without waiting for the second half of the weights (36 of 72 KB on the 64-channel layers: ~2 k cycles of every launch)
        int w1cnt = 0;
        if (!stream_w) {
            for (int c = 1; c < p.nchunks; ++c) {
                issue_weights(c, c);
                if (lw < NTB * TAPS) w1cnt += (NTB * TAPS - lw + nld - 1) / nld;
            }
        }
        for (int s = 0; s < nsteps; ++s) {
            const int inflight = min(D - 2, nsteps - 1 - s);
            WaitVm<63>::go(min(63, inflight * per_step + (s == 0 ? w1cnt : 0)));
            __builtin_amdgcn_s_barrier();
            if (s + D - 1 < nsteps) issue_next(s + D - 1);
        }
        return;
    }

    // =================== compute waves: MFMA + epilogue ===================
    const int wn = wave % WN, wm = wave / WN;
    const int fpy = FOLD ? (wn >> 1) : 0, fpx = FOLD ? (wn & 1) : 0;       // fold: this wave's output phase
    // MFMA pixel operand: m-tile m, lane l -> tile pixel pp = 16m + (l & 15) of this wave's 4 x 20 rows
    int xa[MREP][3], ry[MREP], rx[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        // column tile m < 4 = the first 16 pixels of tile row m, column tile 4 = the last 4 pixels of the four rows: 16 pixels cut from
        // the LINEAR 80-pixel run wrap into the next patch row (+5 slots), which puts lanes 0-3 and 12-15 of a ds_read_b128 group on the
        // same bank quarter with the same swizzle bit -- 3 of 5 column tiles were 2-way conflicted (SQ_LDS_BANK_CONFLICT 25-33 % of the
        // LDS-active cycles); row runs are conflict-free, the 4 x 4 block stays 2-way.  (p.pk_lin: the old map, for A/B runs.)
        const int pp = m * 16 + (lane & 15), li = lane & 15;
        const int py = p.pk_lin ? (pp * 205) >> 12 : (m < 4 ? m : li >> 2);      // pp / 20 for pp < 80
        const int px = p.pk_lin ? pp - py * PK_TW : (m < 4 ? li : 16 + (li & 3));
        ry[m] = wm * PK_ROWS + py; rx[m] = px;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            // fold: window column b (= dx < 2) of phase column fpx is patch column px + fpx + b, window row a is patch row py + fpy + a
            const int s = S == 2 ? (2 * ry[m]) * PWP + (dx == 1 ? PK_PWP + px : px + (dx >> 1))      // input (2 y + dy, 2 x + dx), planes
                                 : (ry[m] + fpy) * PK_PWP + px + fpx + dx;
            xa[m][dx] = (s * 64 + (lane >> 4) * 16) ^ ((s & 4) << 3);
        }
    }
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)p.res_bytes, 0x00020000);
    const bool fast_epi = NREP2 == 0 && !p.scalar_store && !p.out_f32 && !p.deconv_c &&
                          (sizeof(T) != 2 || NREP % 2 || (p.Cout & 7) == 0);    // a 16-B pair must not straddle Cout
    int b, oy0, ox0;
    tile_coords(t, b, oy0, ox0);
    int step = 0;
    // fused stage operands that are the same for every tile: loaded once per launch (fold only; the other fused ops run per tile)
    Stage2Regs<T, NREP, NREP2 ? NREP2 : 1> s2r;
    if constexpr (FOLD) stage2_preload<T, NREP, NREP2>(p, lane, true, s2r);
    VTI_STAMP(0);
    pk_stagger_wait(p.pk_stagger);
    while (true) {
        f32x4 acc[MREP][NREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < p.nchunks; ++c, ++step) {
            const int cur = step % D;
            __builtin_amdgcn_s_barrier();               // the loaders waited for DMA(step) before they arrived
            asm volatile("" ::: "memory");
            if (step < 2) VTI_STAMP(1 + 4 * step);
            // ---- MFMA over the 9 taps of this chunk (software pipelined as in conv.hip)
            {
                const char* sx = smem + cur * stage_bytes;
                const char* sw = smem + wbuf_off + (stream_w ? cur : c) * WCHUNK + wn * (NREP * TAPS * 1024) + lane * 16;
                constexpr int NSTEP = TAPS * MREP;
                // pixel fragments in flight: a step is NREP MFMAs (16 cycles each), an LDS read takes ~100+ cycles with 8 waves on
                // the CU, so small register tiles need a deeper queue to keep the matrix pipe fed
                constexpr int XD = NREP >= 5 ? 3 : NREP == 4 ? 5 : NREP == 3 ? 5 : NREP == 2 ? 6 : 8, WD = 2;
                vec xq[XD];
                vec wq[WD][NREP];
                auto ldx = [&](int s_) -> vec {
                    const int tp = s_ / MREP, mm = s_ % MREP;
                    if constexpr (FOLD) return *(const vec*)(sx + xa[mm][tp & 1] + (tp >> 1) * (PK_PWP * 64));
                    else return *(const vec*)(sx + xa[mm][tp % 3] + (tp / 3) * (PWP * 64));
                };
                auto ldw = [&](int tp, vec (&w)[NREP]) {
#pragma unroll
                    for (int n = 0; n < NREP; ++n) w[n] = *(const vec*)(sw + (n * TAPS + tp) * 1024);
                };
                if constexpr (Tr<T>::H2) {
                    auto ldw1 = [&](int tp, int n) -> vec { return *(const vec*)(sw + (n * TAPS + tp) * 1024); };
#ifndef H2_XD
#define H2_XD (NREP >= 4 ? 3 : 4)
#endif
                    if constexpr (NREP >= 5) h2_taps_nmajor<NREP, MREP, TAPS>(acc, ldx, ldw1);
                    else h2_taps<NREP, MREP, TAPS, H2_XD>(acc, ldx, ldw1);
                } else {
                ldw(0, wq[0]);
#pragma unroll
                for (int i = 0; i < XD - 1; ++i) xq[i] = ldx(i);
#pragma unroll
                for (int s_ = 0; s_ < NSTEP; ++s_) {
                    const int tp = s_ / MREP, mm = s_ % MREP;
                    if (s_ + XD - 1 < NSTEP) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
                    if (mm == 0 && tp + 1 < TAPS) ldw(tp + 1, wq[(tp + 1) % WD]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < NREP; ++n) acc[mm][n] = mma(wq[tp % WD][n], xq[s_ % XD], acc[mm][n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
            }
            if (step < 2) VTI_STAMP(3 + 4 * step);
        }
        VTI_STAMP(11);
        // ---- epilogue of tile t
        if (fast_epi) {
            const int crun = (nt0 + wn * NREP) * 16 + (lane >> 4) * 4 * NREP;
            const char* sb = smem + bias_off + (wn * NREP * 16 + (lane >> 4) * 4 * NREP) * 4;
            f32x4 bias_r[NREP];
#pragma unroll
            for (int n = 0; n < NREP; ++n) bias_r[n] = *(const f32x4*)(sb + n * 16);
            auto epi = [&](auto has_res_c) {
            constexpr bool has_res = decltype(has_res_c)::value;
            // residual of the WHOLE tile first (fp16, register tiles up to NREP = 2: 20 registers): loading it per m-tile costs
            // one memory round trip per m-tile (16 -> 16 at 160x160: 58 us with, 43 us without a residual)
            constexpr bool RES_UPFRONT = has_res && sizeof(T) == 2 && NREP <= 2;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 rr[RES_UPFRONT ? MREP : 1][RES_UPFRONT ? NREP : 1];
            if constexpr (RES_UPFRONT) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) {
                    const int gy = oy0 + ry[m], gx = ox0 + rx[m];
                    const bool pv = gy < p.Hout && gx < p.Wout;
                    const int opix = (b * p.Hout + gy) * p.Wout + gx;
                    const unsigned rb = (unsigned)((opix * p.res_ld + p.res_coff + crun) * ES);
#pragma unroll
                    for (int n = 0; n < NREP; ++n)
                        rr[m][n] = __builtin_amdgcn_raw_buffer_load_b64(rsR, (pv && crun + 4 * n < p.Cout) ? rb + n * 8 : OOB, 0u, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int gy = oy0 + ry[m], gx = ox0 + rx[m];
                const bool pv = gy < p.Hout && gx < p.Wout;
                const int opix = (b * p.Hout + gy) * p.Wout + gx;
                const unsigned ob = (unsigned)((opix * p.out_ld + p.out_coff + crun) * ES);
                f32x4 v[NREP];
                if constexpr (RES_UPFRONT) {
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        const half4 r = __builtin_bit_cast(half4, rr[m][n]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[n][j] = (float)r[j];
                    }
                } else if constexpr (has_res) {
                    const unsigned rb = (unsigned)((opix * p.res_ld + p.res_coff + crun) * ES);
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        const bool cv = pv && crun + 4 * n < p.Cout;
                        if constexpr (sizeof(T) == 2) {
                            const u32x2 r2 = __builtin_amdgcn_raw_buffer_load_b64(rsR, cv ? rb + n * 8 : OOB, 0u, 0);
                            const half4 r = __builtin_bit_cast(half4, r2);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[n][j] = (float)r[j];
                        } else {
                            v[n] = unpack4<T>(buf_load16<u32x4>(rsR, cv ? rb + n * 16 : OOB, 0u));
                        }
                    }
                }
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    f32x4 a = acc_bias<T>(acc[m][n], bias_r[n], p.alpha);
                    if (p.act) a = silu4<FAST>(a);
                    if constexpr (has_res) v[n] += a; else v[n] = a;
                }
                if constexpr (sizeof(T) == 2 && NREP % 2 == 0) {
#pragma unroll
                    for (int n = 0; n < NREP; n += 2) {
                        half8 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v[n][j]; hv[4 + j] = (half_t)v[n + 1][j]; }
                        const bool cv = pv && crun + 4 * n + 8 <= p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rsO, cv ? ob + n * 8 : OOB, 0u, 0);
                    }
                } else if constexpr (sizeof(T) == 2) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[n][j];
                        const bool cv = pv && crun + 4 * n < p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rsO, cv ? ob + n * 8 : OOB, 0u, 0);
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        const bool cv = pv && crun + 4 * n < p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b128(pack4<T>(v[n]), rsO, cv ? ob + n * 16 : OOB, 0u, 0);
                    }
                }
            }
            };
            if (p.has_res) epi(std::true_type{}); else epi(std::false_type{});
        } else {
            int opy[MREP], opx[MREP];
            bool pvalid[MREP];
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                opy[m] = oy0 + ry[m]; opx[m] = ox0 + rx[m];
                pvalid[m] = opy[m] < p.Hout && opx[m] < p.Wout;
            }
            if constexpr (FOLD) {
#pragma unroll
                for (int m = 0; m < MREP; ++m) { opy[m] = 2 * opy[m] + fpy; opx[m] = 2 * opx[m] + fpx; }
                // no pixel of an interior tile lies on the first / last output row or column: one bias vector for all of them
                const bool interior = oy0 > 0 && oy0 + p.TH < p.Hout && ox0 > 0 && ox0 + PK_TW < p.Wout;
                conv_stage2<T, NREP, NREP2, true, true>(p, acc, pvalid, opy, opx, b, lane, &s2r, interior);
            } else if constexpr (NREP2 == 0) conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, b, nt0, wn, lane);
            else conv_stage2<T, NREP, NREP2>(p, acc, pvalid, opy, opx, b, lane);
        }
        VTI_STAMP(12);
        const int tn = t + tstride;
        if (tn >= tend) break;
        t = tn;
        tile_coords(t, b, oy0, ox0);
    }
}

// =====================================================================================================
// Persistent 1x1 convolution (C2f cv1/cv2, SPPF, ConvTranspose2d(2,2) as a 4-way 1x1 GEMM): HBM-bound layers.
// Same roles as conv3_pk (compute waves + loader waves, LDS-DMA, swizzled 64-B pixel slots), but
//  * a tile is a run of (compute waves along M) x 80 consecutive pixels of the flattened [B*H*W] index space;
//  * the stage ring is `pk_depth` deep (up to 8): the loaders run pk_depth-1 steps ahead, so a CU keeps
//    ~100 KB of loads in flight (Little's law at the ~4.5 us loaded latency measured on the 3x3 kernel) instead
//    of one 20-KB stage; they wait for the OLDEST stage with a counted s_waitcnt (every loader wave issues the
//    same number of DMA pieces per step -- surplus pieces are all-lanes-out-of-range writes into a dummy slot);
//  * weights stay in LDS for the whole launch when all K chunks of the workgroup's n-group fit (`pk_wstat`),
//    otherwise the chunk of a step travels with its pixels in the ring.
// One raw s_barrier per (tile, chunk) step.
//  * CPS chunks per step (4-byte storage: 2): a K chunk of the fp32 / h2 engines is 16 channels, i.e. 10 NREP (h2: 20 NREP) MFMAs
//    per wave between two barriers -- on the K = 256..512 layers of the 20x20 / 40x40 maps (16-32 steps per tile) the barrier, the
//    operand reads' latency and, for h2, the operand preparation in front of every step's MFMAs were 3/4 of a step (stamps:
//    2.2 k cycles per step for 640 cycles of MFMAs).  A step now carries two chunks (two 64-B slot images per stage): half the
//    barriers, and h2 prepares the second chunk's operands in the shadow of the first chunk's MFMAs.

constexpr int PK1_MAXP = 5;       // pixel DMA pieces per loader wave, chunk and step (80 px = 5 pieces per M-wave, WN >= 1)

size_t conv1_pk_lds_bytes(int nwm, int WN, int NREP, int nchunks, int depth, int wstat, int cps) {
    const size_t stage = (size_t)nwm * 80 * 64 * cps;
    const size_t wch = (size_t)WN * NREP * 1024;
    return (size_t)depth * stage + (wstat ? (size_t)nchunks * wch : (size_t)depth * wch * cps) + 1024 /*dummy*/ + (size_t)WN * NREP * 64;
}

template <typename T, int NREP, int WN, int CPS>
__global__ __launch_bounds__(512) void conv1_pk(const ConvParams p) {
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, KC = Tr<T>::KC, ES = (int)sizeof(T);
    constexpr int NTB = WN * NREP;
    constexpr int WCH = NTB * 1024;
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool FAST = !Tr<T>::F32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwm = p.TH;                                   // compute waves along M (p.TH is reused for it)
    const int ncomp = nwm * WN;
    const int nld = (int)(blockDim.x >> 6) - ncomp;
    const int D = p.pk_depth;
    const int tile_px = nwm * 80;
    const int img_bytes = tile_px * 64;                     // one chunk's slot image
    const int stage_bytes = img_bytes * CPS;
    const int npieces = tile_px / 16;
    const int wbuf_off = D * stage_bytes;
    const int dummy_off = wbuf_off + (p.pk_wstat ? p.nchunks : D * CPS) * WCH;
    const int spt = (p.nchunks + CPS - 1) / CPS;            // steps per tile
    const int bias_off = dummy_off + 1024;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nt0 = blockIdx.y * NTB;
    const int total_px = p.B * p.Hout * p.Wout;

    for (int i = tid; i < NTB * 16; i += (int)blockDim.x) ((float*)(smem + bias_off))[i] = p.bias[nt0 * 16 + i];
    __syncthreads();

    int t, tend, tstride;
    if (p.pk_xcd) {
        const int per = (p.pk_tiles + 7) >> 3, k = blockIdx.x & 7;
        t = k * per + (int)(blockIdx.x >> 3);
        tend = min((k + 1) * per, p.pk_tiles);
        tstride = (int)(gridDim.x >> 3);
    } else {
        t = blockIdx.x; tend = p.pk_tiles; tstride = (int)gridDim.x;
    }
    if (t >= tend) return;
    const int ntiles_mine = (tend - t + tstride - 1) / tstride;
    const int nsteps = ntiles_mine * spt;

    if (wave >= ncomp) {
        // =================== loader waves ===================
        const int lw = wave - ncomp;
        const int q = (lane & 3) ^ (((lane >> 4) & 1) << 1);            // channel piece of this lane's 16-B position
        const int cvalid = (p.Cin - q * VEC + KC - 1) / KC;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)p.wpk_bytes, 0x00020000);
        const int ppl = (npieces + nld - 1) / nld;                      // pixel pieces per loader and step (<= PK1_MAXP)
        const int wpl = p.pk_wstat ? 0 : (NTB + nld - 1) / nld;         // weight pieces per loader and step
        const int per_step = CPS * (ppl + wpl);
        const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in2, 0, (int)p.in2_bytes, 0x00020000);
        // per-lane source offsets of this loader's pieces, recomputed when the chain moves to the next tile: vo1 = the conv's
        // own input; vo2 = the low-resolution source of the folded Upsample (pixel (y >> 1, x >> 1) of the same frame)
        unsigned vo1[PK1_MAXP], vo2[PK1_MAXP];
        int cur_ti = -1;
        auto issue = [&](int s) {                                       // DMA of step s of this workgroup's chain
            const int ti = s / spt, cs = s - ti * spt;
            if (ti != cur_ti) {
                cur_ti = ti;
                const int pix0 = (t + ti * tstride) * tile_px;
                const int hw = p.Hout * p.Wout;
#pragma unroll
                for (int u = 0; u < PK1_MAXP; ++u) {
                    const int piece = lw + u * nld;
                    const int px = pix0 + piece * 16 + (lane >> 2);
                    const bool ok = piece < npieces && px < total_px;
                    vo1[u] = ok ? (unsigned)((px * p.in_ld + p.in_coff + q * VEC) * ES) : OOB;
                    vo2[u] = OOB;
                    if (p.up_C > 0 && ok) {
                        const int bb = px / hw, r = px - bb * hw;
                        const int y = r / p.Wout, x = r - y * p.Wout;
                        const int px2 = (bb * (p.Hout >> 1) + (y >> 1)) * (p.Wout >> 1) + (x >> 1);
                        vo2[u] = (unsigned)((px2 * p.in2_ld + p.in2_coff + q * VEC) * ES);
                    }
                }
            }
            const int slot = s % D;
#pragma unroll
            for (int cc = 0; cc < CPS; ++cc) {
                // a chunk past the last one (odd chunk count) is issued all the same -- every step must issue the same number of DMA
                // instructions for the counted wait -- with every lane out of range: zeros, which the compute waves do not read
                const int c = cs * CPS + cc;
                const bool live = c < p.nchunks;
                const bool qok = c < cvalid && live;
                const bool from_up = c * KC < p.up_C;                  // chunk granularity: up_C is a multiple of KC
                const unsigned dst = lds0 + slot * stage_bytes + cc * img_bytes;
#pragma unroll
                for (int u = 0; u < PK1_MAXP; ++u) {
                    if (u >= ppl) break;
                    const int piece = lw + u * nld;
                    const unsigned vo = qok ? (from_up ? vo2[u] : vo1[u]) : OOB;
                    const unsigned ld = piece < npieces ? dst + piece * 1024 : lds0 + dummy_off;
                    if (from_up) dma16(rsA2, vo, (unsigned)(c * KC * ES), ld);
                    else dma16(rsA, vo, (unsigned)(c * KC * ES), ld);
                }
                if (wpl) {
                    const unsigned src = (unsigned)(((size_t)(live ? c : 0) * p.ntiles_n + nt0) * 1024);
                    const unsigned wd = lds0 + wbuf_off + (slot * CPS + cc) * WCH;
                    for (int u = 0; u < wpl; ++u) {
                        const int f = lw + u * nld;
                        dma16(rsB, (f < NTB && live) ? (unsigned)lane * 16u : OOB, src + f * 1024, f < NTB ? wd + f * 1024 : lds0 + dummy_off);
                    }
                }
            }
        };
        if (p.pk_wstat) {                                               // every chunk of this n-group, once
            for (int f = lw; f < p.nchunks * NTB; f += nld) {
                const int c = f / NTB, n = f - c * NTB;
                dma16(rsB, (unsigned)lane * 16u, (unsigned)(((size_t)c * p.ntiles_n + nt0 + n) * 1024), lds0 + wbuf_off + f * 1024);
            }
        }
        const int ahead = min(D - 1, nsteps);
        for (int s = 0; s < ahead; ++s) issue(s);
        for (int s = 0; s < nsteps; ++s) {
            // steps s+1 .. min(s+D-2, nsteps-1) may stay in flight; everything older (incl. the stationary weights) is waited for
            const int inflight = min(D - 2, nsteps - 1 - s);
            WaitVm<63>::go(min(63, inflight * per_step));
            __builtin_amdgcn_s_barrier();
            if (s + D - 1 < nsteps) issue(s + D - 1);                   // refills the slot the compute waves left before this barrier
        }
        return;
    }

    // =================== compute waves ===================
    const int wn = wave % WN, wm = wave / WN;
    int xa[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int sl = wm * 80 + m * 16 + (lane & 15);
        xa[m] = (sl * 64 + (lane >> 4) * 16) ^ ((sl & 4) << 3);
    }
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const bool fast_epi = !p.scalar_store && !p.out_f32 && !p.deconv_c && !p.has_res &&
                          (sizeof(T) != 2 || NREP % 2 || (p.Cout & 7) == 0);
    const int crun = (nt0 + wn * NREP) * 16 + (lane >> 4) * 4 * NREP;
    int s = 0;
    for (int ti = 0; ti < ntiles_mine; ++ti) {
        f32x4 acc[MREP][NREP];
#pragma unroll
        for (int m = 0; m < MREP; ++m)
#pragma unroll
            for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cs = 0; cs < spt; ++cs, ++s) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int slot = s % D;
            const int ncc = min(CPS, p.nchunks - cs * CPS);         // live chunks of this step (wave-uniform)
            if constexpr (Tr<T>::H2) {
                // units u = (chunk of the step, n-tile): raw fragment read two units ahead, WH / WL prepared during the previous unit's
                // 2 x MREP MFMAs (one VALU per MFMA), pixel fragments of the next chunk read under the current chunk's last unit
                constexpr int NU = CPS * NREP;
                const int nu = ncc * NREP;
                auto wptr = [&](int u) -> const char* {
                    const int cc = u / NREP, n = u - cc * NREP;
                    return smem + wbuf_off + (p.pk_wstat ? cs * CPS + cc : slot * CPS + cc) * WCH + (wn * NREP + n) * 1024 + lane * 16;
                };
                u32x4 wraw[2], wh[2], wl[2];
                h2x4 x[2][MREP];
                wraw[0] = *(const u32x4*)wptr(0);
                if (nu > 1) wraw[1] = *(const u32x4*)wptr(1);
#pragma unroll
                for (int m = 0; m < MREP; ++m) x[0][m] = *(const h2x4*)(smem + slot * stage_bytes + xa[m]);
#pragma unroll
                for (int i = 0; i < 4; ++i) { wh[0][i] = __builtin_amdgcn_perm(wraw[0][i], wraw[0][i], 0x01000100u); wl[0][i] = wraw[0][i] >> 16; }
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    if (u < nu) {
                        const int cc = u / NREP, n = u % NREP, cur = u & 1, nxt = cur ^ 1;
                        if (n == 0 && cc + 1 < CPS && cc + 1 < ncc) {
#pragma unroll
                            for (int m = 0; m < MREP; ++m) x[(cc + 1) & 1][m] = *(const h2x4*)(smem + slot * stage_bytes + (cc + 1) * img_bytes + xa[m]);
                        }
                        u32x4 wnext = wraw[nxt];                       // raw fragment of unit u + 1 (read at unit u - 1)
                        if (u + 2 < nu) wraw[cur] = *(const u32x4*)wptr(u + 2);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 2 * MREP; ++j) {
                            const int m = j >> 1;
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, (j & 1) ? wl[cur] : wh[cur]),
                                                                               __builtin_bit_cast(half8, x[cc & 1][m].u), acc[m][n], 0, 0, 0);
                            if (j < 8 && u + 1 < nu) {
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
            } else {
#pragma unroll
                for (int cc = 0; cc < CPS; ++cc) {
                    if (cc < ncc) {
                        const int c = cs * CPS + cc;
                        const char* sx = smem + slot * stage_bytes + cc * img_bytes;
                        const char* sw = smem + wbuf_off + (p.pk_wstat ? c : slot * CPS + cc) * WCH + wn * (NREP * 1024) + lane * 16;
                        vec w[NREP];
#pragma unroll
                        for (int n = 0; n < NREP; ++n) w[n] = *(const vec*)(sw + n * 1024);
                        vec x[MREP];
#pragma unroll
                        for (int m = 0; m < MREP; ++m) x[m] = *(const vec*)(sx + xa[m]);
#pragma unroll
                        for (int m = 0; m < MREP; ++m)
#pragma unroll
                            for (int n = 0; n < NREP; ++n) acc[m][n] = mma(w[n], x[m], acc[m][n]);
                    }
                }
            }
        }
        const int pix0 = (t + ti * tstride) * tile_px + wm * 80;
        if (fast_epi) {
            const char* sb = smem + bias_off + (wn * NREP * 16 + (lane >> 4) * 4 * NREP) * 4;
            f32x4 bias_r[NREP];
#pragma unroll
            for (int n = 0; n < NREP; ++n) bias_r[n] = *(const f32x4*)(sb + n * 16);
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int opix = pix0 + m * 16 + (lane & 15);
                const bool pv = opix < total_px;
                const unsigned ob = (unsigned)((opix * p.out_ld + p.out_coff + crun) * ES);
                f32x4 v[NREP];
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    v[n] = acc_bias<T>(acc[m][n], bias_r[n], p.alpha);
                    if (p.act) v[n] = silu4<FAST>(v[n]);
                }
                if constexpr (sizeof(T) == 2 && NREP % 2 == 0) {
#pragma unroll
                    for (int n = 0; n < NREP; n += 2) {
                        half8 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v[n][j]; hv[4 + j] = (half_t)v[n + 1][j]; }
                        const bool cv = pv && crun + 4 * n + 8 <= p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rsO, cv ? ob + n * 8 : OOB, 0u, 0);
                    }
                } else if constexpr (sizeof(T) == 2) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        half4 hv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[n][j];
                        const bool cv = pv && crun + 4 * n < p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rsO, cv ? ob + n * 8 : OOB, 0u, 0);
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NREP; ++n) {
                        const bool cv = pv && crun + 4 * n < p.Cout;
                        __builtin_amdgcn_raw_buffer_store_b128(pack4<T>(v[n]), rsO, cv ? ob + n * 16 : OOB, 0u, 0);
                    }
                }
            }
        } else {
            int opy[MREP], opx[MREP];
            bool pvalid[MREP];
            const int hw = p.Hout * p.Wout;
            int bfr = 0;
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int opix = pix0 + m * 16 + (lane & 15);
                pvalid[m] = opix < total_px;
                const int oc = pvalid[m] ? opix : 0;
                const int bb = oc / hw, r = oc - bb * hw;
                opy[m] = r / p.Wout; opx[m] = r - opy[m] * p.Wout;
                // the shared epilogue addresses pixels as (b * Hout + y) * Wout + x with ONE frame index per call:
                // fold the frame into y (rows of later frames follow the rows of frame 0 in memory)
                opy[m] += bb * p.Hout;
            }
            conv_epilogue<T, NREP>(p, acc, pvalid, opy, opx, bfr, nt0, wn, lane);
        }
    }
}

template <typename T, int NREP, int WN, int NREP2 = 0, bool FOLD = false, int S = 1>
static hipError_t launch_pk_one(const ConvParams& p, dim3 grid, int threads, size_t lds, hipStream_t st) {
    auto k = conv3_pk<T, NREP, WN, NREP2, FOLD, S>;
    static bool attr_done_dev[kMaxDevices] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(threads), lds, st, p);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_pk_t(int nrep, const ConvParams& p, dim3 grid, int threads, size_t lds, hipStream_t st) {
    if (p.ntiles2 > 0) {
        if (p.WN != 1) return hipErrorInvalidValue;
#define VTI_F(N, N2) if (nrep == N && p.ntiles2 == N2) return launch_pk_one<T, N, 1, N2>(p, grid, threads, lds, st);
        VTI_F(2, 2) VTI_F(3, 2) VTI_F(4, 1) VTI_F(4, 2) VTI_F(4, 4) VTI_F(4, 5) VTI_F(5, 5)
#undef VTI_F
        return hipErrorInvalidValue;
    }
#define VTI_L(N, W) if (nrep == N && p.WN == W) return launch_pk_one<T, N, W>(p, grid, threads, lds, st);
    VTI_L(1, 1) VTI_L(2, 1) VTI_L(3, 1) VTI_L(4, 1) VTI_L(5, 1) VTI_L(1, 2) VTI_L(2, 2) VTI_L(3, 2) VTI_L(4, 2) VTI_L(1, 4) VTI_L(2, 4)
#undef VTI_L
    return hipErrorInvalidValue;
}

bool conv_pk_instantiated(int nrep, int wn) {
    return (wn == 1 && nrep >= 1 && nrep <= 5) || (wn == 2 && nrep >= 1 && nrep <= 4) || (wn == 4 && nrep >= 1 && nrep <= 2);
}

template <typename T, int NREP, int WN>
static hipError_t launch_pk1_one(const ConvParams& p, dim3 grid, int threads, size_t lds, hipStream_t st) {
    constexpr int CPS = sizeof(T) == 2 ? 1 : 2;            // chunks per step: the planner sets pk_cps to the same value
    if (p.pk_cps != CPS) return hipErrorInvalidValue;
    auto k = conv1_pk<T, NREP, WN, CPS>;
    static bool attr_done_dev[kMaxDevices] = {};
    bool& attr_done = attr_done_dev[current_device_slot()];
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(threads), lds, st, p);
    return hipGetLastError();
}

bool conv1_pk_instantiated(int nrep, int wn) { return conv_pk_instantiated(nrep, wn) || (wn == 4 && nrep == 4); }

bool conv1_pk_fits(int nwm, int WN, int NREP, int nchunks, int depth, int wstat, int cps) {
    const int ncomp = nwm * WN;
    if (nwm < 1 || ncomp > 4 || depth < 2 || depth > 8 || cps < 1 || cps > 2) return false;
    const int npieces = nwm * 5;
    if ((npieces + ncomp - 1) / ncomp > PK1_MAXP) return false;
    const int per_step = cps * ((npieces + ncomp - 1) / ncomp + (wstat ? 0 : (WN * NREP + ncomp - 1) / ncomp));
    if (per_step * (depth - 2) > 63) return false;                     // counted s_waitcnt range
    return conv1_pk_lds_bytes(nwm, WN, NREP, nchunks, depth, wstat, cps) <= 160 * 1024;
}

template <typename T>
static hipError_t launch_pk1_t(int nrep, const ConvParams& p, dim3 grid, int threads, size_t lds, hipStream_t st) {
#define VTI_L(N, W) if (nrep == N && p.WN == W) return launch_pk1_one<T, N, W>(p, grid, threads, lds, st);
    VTI_L(1, 1) VTI_L(2, 1) VTI_L(3, 1) VTI_L(4, 1) VTI_L(5, 1) VTI_L(1, 2) VTI_L(2, 2) VTI_L(3, 2) VTI_L(4, 2) VTI_L(1, 4) VTI_L(2, 4) VTI_L(4, 4)
#undef VTI_L
    return hipErrorInvalidValue;
}

// 1x1: p.TH = compute waves along M, p.TW = 80 (pixels per wave); workgroups as for the 3x3 kernel
hipError_t launch_conv1_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    const int NTB = p.WN * nrep;
    if (!conv1_pk_fits(p.TH, p.WN, nrep, p.nchunks, p.pk_depth, p.pk_wstat, p.pk_cps)) return hipErrorInvalidValue;
    if (p.ntiles_n % NTB || p.pk_wgs < 1 || (p.pk_xcd && p.pk_wgs % 8) || p.has_res) return hipErrorInvalidValue;
    if ((size_t)p.in_bytes >= 0x80000000u || (size_t)p.out_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.pk_tiles == 0) return hipSuccess;
    const int threads = 2 * p.TH * p.WN * 64;
    dim3 grid((unsigned)p.pk_wgs, (unsigned)(p.ntiles_n / NTB));
    if (dtype == VTI_F16) return launch_pk1_t<half_t>(nrep, p, grid, threads, lds_bytes, st);
    if (dtype == VTI_H2) return launch_pk1_t<h2_t>(nrep, p, grid, threads, lds_bytes, st);
    return launch_pk1_t<float>(nrep, p, grid, threads, lds_bytes, st);
}

// fold on the persistent schedule: 4 x 20 low-resolution pixels per tile, 4 phase waves + 4 loader waves, 2 stages of a 6 x 24 slot
// patch + both K chunks of the composed weights (2 x 16 n-tiles x 4 taps KiB) + the (unused) bias slot
size_t conv_pk_fold_lds_bytes(int nchunks, int depth) { return depth * (size_t)6 * PK_PWP * 64 + (size_t)(nchunks > 1 ? 2 : 1) * 16 * 4 * 1024 + 16 * 16 * 4; }
int conv_pk_fold_depth(int nchunks) {
    const char* cap = getenv("VTI_PK_FOLD_DEPTH");          // steps here are only ~1.3 k MFMA cycles long: a third stage keeps a patch in flight
    const int maxd = cap ? std::max(2, std::min(4, atoi(cap))) : 3;
    int d = 2;
    while (d < maxd && conv_pk_fold_lds_bytes(nchunks, d + 1) <= 160 * 1024) ++d;
    return d;
}

hipError_t launch_conv_pk_fold(int dtype, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    if (p.TH != 4 || p.TW != PK_TW || p.WN != 4 || p.nchunks > 2 || p.ntiles_n != 16 || !p.fold || !p.out2) return hipErrorInvalidValue;
    if (p.pk_wgs < 1 || (p.pk_xcd && p.pk_wgs % 8) || (size_t)p.in_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.pk_depth < 2 || p.pk_depth > 4 || lds_bytes < conv_pk_fold_lds_bytes(p.nchunks, p.pk_depth) || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (p.pk_tiles == 0) return hipSuccess;
    dim3 grid((unsigned)p.pk_wgs, 1);
#define VTI_FP(N2)                                                                                              \
    if (p.ntiles2 == N2) {                                                                                      \
        if (dtype == VTI_F16) return launch_pk_one<half_t, 4, 4, N2, true>(p, grid, 512, lds_bytes, st);        \
        if (dtype == VTI_H2) return launch_pk_one<h2_t, 4, 4, N2, true>(p, grid, 512, lds_bytes, st);           \
        return launch_pk_one<float, 4, 4, N2, true>(p, grid, 512, lds_bytes, st);                               \
    }
    VTI_FP(1) VTI_FP(2) VTI_FP(4)
#undef VTI_FP
    return hipErrorInvalidValue;
}

// stride-2 3x3 on the persistent schedule (p.pk == 4)
hipError_t launch_conv_pk2(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    const int NTB = p.WN * nrep;
    if (p.TW != PK_TW || !conv_pk2_fits(p.TH, p.WN, nrep, p.nchunks, p.pk_wstat) || p.ntiles2 > 0) return hipErrorInvalidValue;
    if (p.pk_depth < 2 || p.pk_depth > 4 || lds_bytes < conv_pk2_lds_bytes(p.TH, p.WN, nrep, p.nchunks, p.pk_depth, p.pk_wstat) || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (p.ntiles_n % NTB || p.pk_wgs < 1 || (p.pk_xcd && p.pk_wgs % 8)) return hipErrorInvalidValue;
    if ((size_t)p.in_bytes >= 0x80000000u || (size_t)p.out_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.pk_tiles == 0) return hipSuccess;
    const int threads = 2 * (p.TH / PK_ROWS) * p.WN * 64;
    dim3 grid((unsigned)p.pk_wgs, (unsigned)(p.ntiles_n / NTB));
#define VTI_L2(N, W)                                                                                           \
    if (nrep == N && p.WN == W) {                                                                              \
        if (dtype == VTI_F16) return launch_pk_one<half_t, N, W, 0, false, 2>(p, grid, threads, lds_bytes, st); \
        if (dtype == VTI_H2) return launch_pk_one<h2_t, N, W, 0, false, 2>(p, grid, threads, lds_bytes, st);    \
        return launch_pk_one<float, N, W, 0, false, 2>(p, grid, threads, lds_bytes, st);                        \
    }
    VTI_L2(1, 4) VTI_L2(2, 2) VTI_L2(4, 1) VTI_L2(2, 1) VTI_L2(1, 2)
#undef VTI_L2
    return hipErrorInvalidValue;
}

// grid.x workgroups (p.pk_wgs, a multiple of 8 when p.pk_xcd) x n-groups; (TH/4) * WN compute + as many loader waves
hipError_t launch_conv_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    const int NTB = p.WN * nrep;
    if (p.TH % PK_ROWS || p.TW != PK_TW || !conv_pk_fits(p.TH, p.WN, nrep, p.nchunks, p.pk_wstat)) return hipErrorInvalidValue;
    if (p.pk_depth < 2 || p.pk_depth > 4 || lds_bytes < conv_pk_lds_bytes(p.TH, p.WN, nrep, p.nchunks, p.pk_depth, p.pk_wstat) || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (p.ntiles_n % NTB || p.pk_wgs < 1 || (p.pk_xcd && p.pk_wgs % 8)) return hipErrorInvalidValue;
    if ((size_t)p.in_bytes >= 0x80000000u || (size_t)p.out_bytes >= 0x80000000u) return hipErrorInvalidValue;
    if (p.pk_tiles == 0) return hipSuccess;
    const int threads = 2 * (p.TH / PK_ROWS) * p.WN * 64;   // compute waves + as many loader waves
    dim3 grid((unsigned)p.pk_wgs, (unsigned)(p.ntiles_n / NTB));
    if (dtype == VTI_F16) return launch_pk_t<half_t>(nrep, p, grid, threads, lds_bytes, st);
    if (dtype == VTI_H2) return launch_pk_t<h2_t>(nrep, p, grid, threads, lds_bytes, st);
    return launch_pk_t<float>(nrep, p, grid, threads, lds_bytes, st);
}

// =====================================================================================================
// Fused C2f Bottleneck (Ultralytics Bottleneck(c, c, shortcut, k=(3,3), e=1.0): SURVEY section 8 U2): y = [x +] silu(W2 * silu(W1 * x + b1) + b2)
// in ONE persistent kernel -- the 3x3 -> 3x3 pair of model.{2,4,15,...}.m.j.  The intermediate tensor lives only as a (TH+2) x 22
// tile in LDS (halo recompute: conv 1 runs on the tile grown by one pixel, 7 instead of 5 m-tiles per wave), the shortcut is read
// from the input patch that is in LDS anyway, so per pair one read of x (+ halo) and one write of y reach memory instead of
// x, t, t, x, y.  Channel counts of one K chunk (C <= 32 fp16 / 16 fp32: the 160x160 and 80x80 blocks of the n model, where
// these layers are HBM-bound); both convs' weights stay in LDS for the whole launch.
//   * loader waves (as many as compute waves): patch (TH+4) x 24 slots of tile t+1 by LDS-DMA into the other stage while tile t computes
//   * compute wave wm, phase 1: conv 1 on R1 pixels [112 wm, 112 wm + 112) of the (TH+2) x 22 region, bias + SiLU, rounded to T and
//     written to the T image (same swizzled 64-B pixel slots as a DMA'd patch; pixels outside the feature map are conv 2's ZERO padding)
//   * barrier; phase 2: conv 2 on its 4 x 20 pixels from the T image, bias + SiLU (+ shortcut from the input stage), 16-byte stores
// Two workgroup barriers per tile.  ConvParams: wpk/bias = conv 1, w2/bias2 = conv 2; res must be the input view when has_res.
constexpr int BN_MREP1 = 7;

size_t bneck_pk_lds_bytes(int TH, int NREP, int depth) {
    return depth * (size_t)(TH + 4) * PK_PWP * 64 + (size_t)(TH + 2) * PK_PWP * 64 + 2 * (size_t)NREP * 9 * 1024 + 2 * (size_t)NREP * 64;
}
size_t bneck_pk_lds_bytes(int TH, int NREP) { return bneck_pk_lds_bytes(TH, NREP, 2); }
// patch stages: as many (<= 4) as fit -- these layers are HBM-bound and one 15-30 KB patch in flight per CU is ~1/3 of what
// Little's law asks for at ~2 us of loaded latency
int bneck_pk_depth(int TH, int NREP) {
    // default 2: in an A/B on one box the deeper rings made the whole forward ~0.7 % SLOWER (VTI_PK_DEPTH=3/4 to re-measure)
    const char* cap = getenv("VTI_PK_DEPTH");
    const int maxd = cap ? std::max(2, std::min(4, atoi(cap))) : 2;
    const int nwm = TH / PK_ROWS;
    const int per_step = ((TH + 4) * PK_PWP / 16 + nwm - 1) / nwm;
    int d = 2;
    while (d < maxd && bneck_pk_lds_bytes(TH, NREP, d + 1) <= 160 * 1024 && per_step * (d - 1) <= 63) ++d;
    return d;
}
bool bneck_pk_fits(int TH, int NREP) {
    if (TH % PK_ROWS || TH < 8 || TH > 16 || NREP < 1 || NREP > 2) return false;
    const int nwm = TH / PK_ROWS;
    if (((TH + 2) * (PK_TW + 2) + 15) / 16 > BN_MREP1 * nwm) return false;          // conv-1 m-tiles per wave
    if (((TH + 4) * PK_PWP / 16 + nwm - 1) / nwm > PK_MAXD) return false;            // patch DMA pieces per loader wave
    return bneck_pk_lds_bytes(TH, NREP) <= 160 * 1024;
}

// TAIL (fp16, C = 16: the n = 1 C2f of model.2): the C2f's closing 1x1 conv over [y0 | y1 | y2] runs here too.  The patch then carries
// y0 AND y1 (one full 64-byte slot per pixel; conv 1's weights sit at K positions 16..31), and after conv 2's epilogue the wave holds
// y2 of its pixels in registers: out = silu(Wa . [y0, y1](from the patch) + Wb . y2(from registers) + b) as two more MFMA steps per
// (m-tile, n-tile) -- y2 is neither written nor re-read, the whole Y tensor is read once, and model.2.cv2's own launch disappears.
// ConvParams: in_coff = y0's offset, w0 = [Wa tile 0, Wa tile 1, Wb tile 0, Wb tile 1] fragments, bias0 = the 1x1's bias, out2 = its output.
template <typename T, int NREP, bool TAIL = false>
__global__ __launch_bounds__(512) void bneck_pk(const ConvParams p) {
    static_assert(!TAIL || ((sizeof(T) == 2 || Tr<T>::H2) && NREP == 1), "tail: fp16 / h2, 16 channels");
    // h2 tail: a 64-byte slot holds 16 channels, so the patch carries y1 only (no in_coff shift); y0 of the tile's own pixels is read
    // straight from global memory as an MFMA operand (five 16-byte loads per lane and tile, issued at the top of the tile), and the 1x1
    // is three steps per (m-tile, n-tile): Wa0 . y0 + Wa1 . y1(patch) + Wb . y2(registers), one common weight scale (alpha0).
    constexpr bool TAILH2 = TAIL && Tr<T>::H2;
    using vec = typename Tr<T>::vec;
    constexpr int VEC = Tr<T>::VEC, ES = (int)sizeof(T);
    constexpr int TAPS = 9, R1W = PK_TW + 2;
    constexpr unsigned OOB = 0x80000000u;
    constexpr bool FAST = !Tr<T>::F32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwm = p.TH / PK_ROWS;                         // compute waves; as many loader waves
    const int PH = p.TH + 4, R1H = p.TH + 2;
    const int stage_bytes = PH * PK_PWP * 64;
    const int D = p.pk_depth;                               // patch stages
    const int timg_off = D * stage_bytes;
    const int w1_off = timg_off + R1H * PK_PWP * 64;
    const int w2_off = w1_off + NREP * TAPS * 1024;
    const int bias_off = w2_off + NREP * TAPS * 1024;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    for (int i = tid; i < NREP * 16; i += (int)blockDim.x) {
        ((float*)(smem + bias_off))[i] = p.bias[i];
        ((float*)(smem + bias_off))[NREP * 16 + i] = p.bias2[i];
    }
    for (int i = tid; i < R1H * PK_PWP * 4; i += (int)blockDim.x) ((f32x4*)(smem + timg_off))[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    int t, tend, tstride;
    if (p.pk_xcd) {
        const int per = (p.pk_tiles + 7) >> 3, k = blockIdx.x & 7;
        t = k * per + (int)(blockIdx.x >> 3);
        tend = min((k + 1) * per, p.pk_tiles);
        tstride = (int)(gridDim.x >> 3);
    } else {
        t = blockIdx.x; tend = p.pk_tiles; tstride = (int)gridDim.x;
    }
    if (t >= tend) return;
    auto tile_coords = [&](int tt, int& b, int& oy0, int& ox0) {
        const int tx = tt % p.tiles_x, r = tt / p.tiles_x;
        const int ty = r % p.tiles_y;
        b = r / p.tiles_y; oy0 = ty * p.TH; ox0 = tx * PK_TW;
    };
    const int ntiles_mine = (tend - t + tstride - 1) / tstride;

    if (wave >= nwm) {
        // =================== loader waves ===================
        const int lw = wave - nwm, nld = nwm;
        const int ndma = PH * PK_PWP / 16;
        const int q = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const bool qok = q * VEC < ((TAIL && !TAILH2) ? 2 * p.Cin : p.Cin);   // one chunk: channel pieces beyond Cin are written as zeros (fp16 tail: y0 | y1)
        int dyx[PK_MAXD];
#pragma unroll
        for (int u = 0; u < PK_MAXD; ++u) {
            const int s = (lw + u * nld) * 16 + (lane >> 2);
            const int py = (int)(((unsigned)s * 2731u) >> 16), px = s - py * PK_PWP;
            dyx[u] = (py << 8) | px;                        // every column of the 24-slot row is a patch pixel here (20 + 2 x 2 halo)
        }
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)p.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, NREP * TAPS * 1024, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, NREP * TAPS * 1024, 0x00020000);
        auto issue_patch = [&](int tt, int stage) {
            int b, oy0, ox0;
            tile_coords(tt, b, oy0, ox0);
            const unsigned dst = lds0 + stage * stage_bytes + lw * 1024;
#pragma unroll
            for (int u = 0; u < PK_MAXD; ++u) {
                if (lw + u * nld >= ndma) break;
                const int y = oy0 - 2 + (dyx[u] >> 8), x = ox0 - 2 + (dyx[u] & 255);
                const bool ok = qok && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
                const unsigned vo = ok ? (unsigned)((((b * p.Hin + y) * p.Win + x) * p.in_ld + p.in_coff + q * VEC) * ES) : OOB;
                dma16(rsA, vo, 0u, dst + u * nld * 1024);
            }
        };
        for (int f = lw; f < NREP * TAPS; f += nld) {
            dma16(rsW1, (unsigned)lane * 16u, (unsigned)(f * 1024), lds0 + w1_off + f * 1024);
            dma16(rsW2, (unsigned)lane * 16u, (unsigned)(f * 1024), lds0 + w2_off + f * 1024);
        }
        int per_step = 0;
#pragma unroll
        for (int u = 0; u < PK_MAXD; ++u) per_step += (lw + u * nld < ndma) ? 1 : 0;
        // the loaders run D - 1 tiles ahead; before barrier A(s) they wait (counted) for everything up to patch(s)
        const int ahead = min(D - 1, ntiles_mine);
        for (int s = 0; s < ahead; ++s) issue_patch(t + s * tstride, s % D);
        for (int s = 0; s < ntiles_mine; ++s) {
            WaitVm<63>::go(min(63, min(D - 2, ntiles_mine - 1 - s) * per_step));
            __builtin_amdgcn_s_barrier();                   // A: patch(s) has landed; every compute wave left stage (s - 1) % D and the T image
            if (s + D - 1 < ntiles_mine) issue_patch(t + (s + D - 1) * tstride, (s + D - 1) % D);
            __builtin_amdgcn_s_barrier();                   // B
        }
        return;
    }

    // =================== compute waves ===================
    const int wm = wave;
    // phase 1 operand addresses: R1 pixel pp -> (r1y, r1x); patch slot of tap (dy, dx) = (r1y + dy) * 24 + r1x + dx
    int xa1[BN_MREP1][3], r1y[BN_MREP1], r1x[BN_MREP1];
    bool v1[BN_MREP1];
    int tw1[BN_MREP1];                                      // T-image byte address of this lane's channel run
#pragma unroll
    for (int m = 0; m < BN_MREP1; ++m) {
        // column tiles of the (TH + 2) x 22 region: a row's first 16 pixels, then blocks of 2 rows x its last 6 pixels (12 lanes) -- when
        // these fit the waves' 7 tiles each (TH = 16, 12); otherwise the linear cut (whose row wraps are 2-way LDS conflicts: see conv3_pk)
        const int ct = wm * BN_MREP1 + m, li = lane & 15;
        const bool rowrun = !p.pk_lin && R1H + (R1H + 1) / 2 <= BN_MREP1 * (p.TH / PK_ROWS);
        int yy, xx;
        if (rowrun) {
            const int r6 = (li * 43) >> 8;                                           // li / 6 for li < 16
            yy = ct < R1H ? ct : 2 * (ct - R1H) + r6;
            xx = ct < R1H ? li : 16 + li - 6 * r6;
            v1[m] = ct < R1H || (li < 12 && yy < R1H);
            if (!v1[m]) { yy = 0; xx = 0; }
        } else {
            const int pp = ct * 16 + li;
            v1[m] = pp < R1H * R1W;
            const int pc = v1[m] ? pp : 0;
            yy = (int)(((unsigned)pc * 2979u) >> 16); xx = pc - yy * R1W;            // pc / 22 for pc < 8192
        }
        r1y[m] = yy; r1x[m] = xx;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int s = yy * PK_PWP + xx + dx;
            xa1[m][dx] = (s * 64 + g * 16) ^ ((s & 4) << 3);
        }
        const int st = yy * PK_PWP + xx;
        // lane group g owns channels 4 NREP g ..: fp16 NREP 2 / fp32 -> 16-byte piece g; fp16 NREP 1 -> half of piece g >> 1
        const int piece = (sizeof(T) == 2 && NREP == 1) ? (g >> 1) : g;
        tw1[m] = timg_off + ((st * 64 + piece * 16) ^ ((st & 4) << 3)) + ((sizeof(T) == 2 && NREP == 1) ? (g & 1) * 8 : 0);
    }
    // phase 2 operand addresses (as conv3_pk, on the T image whose origin is the tile origin - 1)
    int xa2[MREP][3], ry[MREP], rx[MREP], ra[MREP];
    [[maybe_unused]] int xc[MREP];
#pragma unroll
    for (int m = 0; m < MREP; ++m) {
        const int pp = m * 16 + (lane & 15), li = lane & 15;                      // row runs + one 4 x 4 block: see conv3_pk
        const int py = p.pk_lin ? (pp * 205) >> 12 : (m < 4 ? m : li >> 2);
        const int px = p.pk_lin ? pp - py * PK_TW : (m < 4 ? li : 16 + (li & 3));
        ry[m] = wm * PK_ROWS + py; rx[m] = px;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int s = ry[m] * PK_PWP + px + dx;
            xa2[m][dx] = timg_off + ((s * 64 + g * 16) ^ ((s & 4) << 3));
        }
        const int sr = (ry[m] + 2) * PK_PWP + px + 2;       // the pixel itself in the input patch (shortcut)
        const int piece = ((sizeof(T) == 2 && NREP == 1) ? (g >> 1) : g) + ((TAIL && !TAILH2) ? 2 : 0);      // fp16 tail: y1 is the slot's upper half
        ra[m] = ((sr * 64 + piece * 16) ^ ((sr & 4) << 3)) + ((sizeof(T) == 2 && NREP == 1) ? (g & 1) * 8 : 0);
        xc[m] = (sr * 64 + g * 16) ^ ((sr & 4) << 3);       // tail: the pixel's [y0 | y1] as an MFMA operand (k-group g = piece g)
    }
    [[maybe_unused]] vec wA[2], wB[2];
    [[maybe_unused]] f32x4 b3r[2];
    [[maybe_unused]] u32x4 th[TAILH2 ? 6 : 1], tl[TAILH2 ? 6 : 1];   // h2 tail: prepared operands of [Wa0 n0, Wa0 n1, Wa1 n0, Wa1 n1, Wb n0, Wb n1], kernel-invariant
    if constexpr (TAILH2) {
#pragma unroll
        for (int f = 0; f < 6; ++f) {
            const u32x4 raw = ((const u32x4*)p.w0)[f * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) { th[f][i] = __builtin_amdgcn_perm(raw[i], raw[i], 0x01000100u); tl[f][i] = raw[i] >> 16; }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) b3r[n] = *(const f32x4*)(p.bias0 + g * 8 + 4 * n);
    } else if constexpr (TAIL) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            wA[n] = ((const vec*)p.w0)[n * 64 + lane];
            wB[n] = ((const vec*)p.w0)[(2 + n) * 64 + lane];
            b3r[n] = *(const f32x4*)(p.bias0 + g * 8 + 4 * n);      // permuted rows (NREP2 = 2): lane group g owns channels 8 g .. 8 g + 7
        }
    }
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, TAILH2 ? (int)p.in_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO2 = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, TAIL ? (int)p.out2_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const int crun = g * 4 * NREP;
    f32x4 b1r[NREP], b2r[NREP];
#pragma unroll
    for (int n = 0; n < NREP; ++n) {
        b1r[n] = *(const f32x4*)(smem + bias_off + (crun + 4 * n) * 4);
        b2r[n] = *(const f32x4*)(smem + bias_off + (NREP * 16 + crun + 4 * n) * 4);
    }
    const bool has_res = __builtin_amdgcn_readfirstlane(p.has_res) != 0;
    int step = 0;
    VTI_STAMP(0);
    while (true) {
        int b, oy0, ox0;
        tile_coords(t, b, oy0, ox0);
        const char* sx = smem + (step % D) * stage_bytes;
        [[maybe_unused]] u32x4 y0r[TAILH2 ? MREP : 1];
        if constexpr (TAILH2) {                             // y0 of this wave's pixels: consumed in the tail, a whole tile of MFMAs later
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int gy = oy0 + ry[m], gx = ox0 + rx[m];
                const bool pv = gy < p.Hout && gx < p.Wout;
                const int opix = (b * p.Hout + gy) * p.Wout + gx;
                y0r[m] = buf_load16<u32x4>(rsY, pv ? (unsigned)((opix * p.in_ld + p.in_coff - p.Cin + g * 4) * ES) : OOB, 0u);
            }
        }
        if (step == 2) VTI_STAMP(1);
        __builtin_amdgcn_s_barrier();                       // A
        asm volatile("" ::: "memory");
        if (step == 2) VTI_STAMP(2);
        // ---- phase 1: conv 1 on the grown tile
        {
            f32x4 acc[BN_MREP1][NREP];
#pragma unroll
            for (int m = 0; m < BN_MREP1; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* sw = smem + w1_off + lane * 16;
            constexpr int NSTEP = TAPS * BN_MREP1;
            constexpr int XD = NREP == 2 ? 6 : 8;           // pixel fragments in flight (a step is only NREP MFMAs long)
            vec xq[XD];
            vec wq[2][NREP];
            auto ldx = [&](int s_) -> vec {
                const int tp = s_ / BN_MREP1, mm = s_ % BN_MREP1;
                return *(const vec*)(sx + xa1[mm][tp % 3] + (tp / 3) * (PK_PWP * 64));
            };
            auto ldw = [&](int tp, vec (&w)[NREP]) {
#pragma unroll
                for (int n = 0; n < NREP; ++n) w[n] = *(const vec*)(sw + (n * TAPS + tp) * 1024);
            };
            if constexpr (Tr<T>::H2) {
                auto ldw1 = [&](int tp, int n) -> vec { return *(const vec*)(sw + (n * TAPS + tp) * 1024); };
                h2_taps<NREP, BN_MREP1, TAPS, 4>(acc, ldx, ldw1);
            } else {
            ldw(0, wq[0]);
#pragma unroll
            for (int i = 0; i < XD - 1; ++i) xq[i] = ldx(i);
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                const int tp = s_ / BN_MREP1, mm = s_ % BN_MREP1;
                if (s_ + XD - 1 < NSTEP) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
                if (mm == 0 && tp + 1 < TAPS) ldw(tp + 1, wq[(tp + 1) % 2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[mm][n] = mma(wq[tp % 2][n], xq[s_ % XD], acc[mm][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (step == 2) VTI_STAMP(3);
            // bias + SiLU, rounded to T, into the T image; a pixel outside the feature map is conv 2's zero padding
#pragma unroll
            for (int m = 0; m < BN_MREP1; ++m) {
                const int gy = oy0 - 1 + r1y[m], gx = ox0 - 1 + r1x[m];
                const bool inside = (unsigned)gy < (unsigned)p.Hout && (unsigned)gx < (unsigned)p.Wout;
                f32x4 v[NREP];
#pragma unroll
                for (int n = 0; n < NREP; ++n) {
                    v[n] = silu4<FAST>(acc_bias<T>(acc[m][n], b1r[n], p.alpha));
                    if (!inside) v[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (!v1[m]) continue;
                if constexpr (sizeof(T) == 2 && NREP == 2) {
                    half8 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v[0][j]; hv[4 + j] = (half_t)v[1][j]; }
                    *(half8*)(smem + tw1[m]) = hv;
                } else if constexpr (sizeof(T) == 2) {
                    half4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[0][j];
                    *(half4*)(smem + tw1[m]) = hv;
                } else {
                    static_assert(sizeof(T) == 2 || NREP == 1, "fp32: one 16-channel chunk");
                    *(u32x4*)(smem + tw1[m]) = pack4<T>(v[0]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's T-image writes have been performed (a raw s_barrier does not wait)
        if (step == 2) VTI_STAMP(4);
        __builtin_amdgcn_s_barrier();                       // B: the T image is complete
        asm volatile("" ::: "memory");
        if (step == 2) VTI_STAMP(5);
        // ---- phase 2: conv 2 on the tile, from the T image
        {
            f32x4 acc[MREP][NREP];
#pragma unroll
            for (int m = 0; m < MREP; ++m)
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const char* sw = smem + w2_off + lane * 16;
            constexpr int NSTEP = TAPS * MREP;
            constexpr int XD = NREP == 2 ? 6 : 8;
            vec xq[XD];
            vec wq[2][NREP];
            auto ldx = [&](int s_) -> vec {
                const int tp = s_ / MREP, mm = s_ % MREP;
                return *(const vec*)(smem + xa2[mm][tp % 3] + (tp / 3) * (PK_PWP * 64));
            };
            auto ldw = [&](int tp, vec (&w)[NREP]) {
#pragma unroll
                for (int n = 0; n < NREP; ++n) w[n] = *(const vec*)(sw + (n * TAPS + tp) * 1024);
            };
            if constexpr (Tr<T>::H2) {
                auto ldw1 = [&](int tp, int n) -> vec { return *(const vec*)(sw + (n * TAPS + tp) * 1024); };
                h2_taps<NREP, MREP, TAPS, 4>(acc, ldx, ldw1);
            } else {
            ldw(0, wq[0]);
#pragma unroll
            for (int i = 0; i < XD - 1; ++i) xq[i] = ldx(i);
#pragma unroll
            for (int s_ = 0; s_ < NSTEP; ++s_) {
                const int tp = s_ / MREP, mm = s_ % MREP;
                if (s_ + XD - 1 < NSTEP) xq[(s_ + XD - 1) % XD] = ldx(s_ + XD - 1);
                if (mm == 0 && tp + 1 < TAPS) ldw(tp + 1, wq[(tp + 1) % 2]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NREP; ++n) acc[mm][n] = mma(wq[tp % 2][n], xq[s_ % XD], acc[mm][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (step == 2) VTI_STAMP(6);
#pragma unroll
            for (int m = 0; m < MREP; ++m) {
                const int gy = oy0 + ry[m], gx = ox0 + rx[m];
                const bool pv = gy < p.Hout && gx < p.Wout;
                const int opix = (b * p.Hout + gy) * p.Wout + gx;
                const unsigned ob = (unsigned)((opix * p.out_ld + p.out_coff + crun) * ES);
                f32x4 v[NREP];
#pragma unroll
                for (int n = 0; n < NREP; ++n) v[n] = silu4<FAST>(acc_bias<T>(acc[m][n], b2r[n], p.alpha2));
                if (has_res) {                              // the shortcut: this pixel of the input patch, still in its stage
                    if constexpr (sizeof(T) == 2 && NREP == 2) {
                        const half8 r = *(const half8*)(sx + ra[m]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { v[0][j] += (float)r[j]; v[1][j] += (float)r[4 + j]; }
                    } else if constexpr (sizeof(T) == 2) {
                        const half4 r = *(const half4*)(sx + ra[m]);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[0][j] += (float)r[j];
                    } else {
                        v[0] += unpack4<T>(*(const u32x4*)(sx + ra[m]));
                    }
                }
                if constexpr (TAILH2) {
                    const u32x4 xB = pack4<T>(v[0]);                       // y2 as it would have been stored
                    const u32x4 xA1 = *(const u32x4*)(sx + xc[m]);        // y1: this pixel's slot of the input patch
                    const half8 x0 = __builtin_bit_cast(half8, y0r[m]), x1 = __builtin_bit_cast(half8, xA1), x2 = __builtin_bit_cast(half8, xB);
                    f32x4 a3[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, th[n]), x0, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, tl[n]), x0, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, th[2 + n]), x1, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, tl[2 + n]), x1, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, th[4 + n]), x2, a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, tl[4 + n]), x2, a, 0, 0, 0);
                        a3[n] = silu4<FAST>(acc_bias<T>(a, b3r[n], p.alpha0));
                    }
                    const unsigned ob2 = (unsigned)((opix * p.out2_ld + p.out2_coff + g * 8) * ES);
                    __builtin_amdgcn_raw_buffer_store_b128(pack4<T>(a3[0]), rsO2, pv ? ob2 : OOB, 0u, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(pack4<T>(a3[1]), rsO2, pv ? ob2 + 16 : OOB, 0u, 0);
                } else if constexpr (TAIL) {
                    // y2 of this pixel (rounded to T as it would have been stored) is the K operand of the second tail step:
                    // elements 0..3 = channels 4 g .. 4 g + 3 (pack_conv_stage2's K order for a one-tile producer), 4..7 = 0
                    vec xB;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xB[j] = (T)v[0][j]; xB[4 + j] = (T)0; }
                    const vec xA = *(const vec*)(sx + xc[m]);
                    f32x4 a3[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        a3[n] = mma(wA[n], xA, (f32x4){0.f, 0.f, 0.f, 0.f});
                        a3[n] = mma(wB[n], xB, a3[n]);
                        a3[n] = silu4<FAST>(a3[n] + b3r[n]);
                    }
                    half8 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hv[j] = (half_t)a3[0][j]; hv[4 + j] = (half_t)a3[1][j]; }
                    const unsigned ob2 = (unsigned)((opix * p.out2_ld + p.out2_coff + g * 8) * ES);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rsO2, pv ? ob2 : OOB, 0u, 0);
                } else if constexpr (sizeof(T) == 2 && NREP == 2) {
                    half8 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hv[j] = (half_t)v[0][j]; hv[4 + j] = (half_t)v[1][j]; }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hv), rsO, pv ? ob : OOB, 0u, 0);
                } else if constexpr (sizeof(T) == 2) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    half4 hv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) hv[j] = (half_t)v[0][j];
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rsO, pv ? ob : OOB, 0u, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(pack4<T>(v[0]), rsO, pv ? ob : OOB, 0u, 0);
                }
            }
        }
        if (step == 2) VTI_STAMP(7);
        const int tn = t + tstride;
        if (tn >= tend) break;
        t = tn; ++step;
    }
    VTI_STAMP(12);
}

template <typename T, int NREP, bool TAIL = false>
static hipError_t launch_bneck_one(const ConvParams& p, dim3 grid, int threads, size_t lds, hipStream_t st) {
    auto k = bneck_pk<T, NREP, TAIL>;
    static bool done_dev[kMaxDevices] = {};
    bool& done = done_dev[current_device_slot()];
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL(k, grid, dim3(threads), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_bneck_pk(int dtype, int nrep, const ConvParams& p, size_t lds_bytes, hipStream_t st) {
    if (p.TW != PK_TW || !bneck_pk_fits(p.TH, nrep) || p.nchunks != 1 || p.Cin != 16 * nrep || p.Cout != 16 * nrep || !p.w2 || !p.bias2)
        return hipErrorInvalidValue;
    if (p.Hin != p.Hout || p.Win != p.Wout || p.pk_wgs < 1 || (p.pk_xcd && p.pk_wgs % 8)) return hipErrorInvalidValue;
    if ((size_t)p.in_bytes >= 0x80000000u || (size_t)p.out_bytes >= 0x80000000u) return hipErrorInvalidValue;
    const int tail = p.w0 != nullptr;                       // the C2f's closing 1x1 in the same kernel (in_coff is then y0's offset)
    if (tail && ((dtype != VTI_F16 && dtype != VTI_H2) || nrep != 1 || !p.out2 || !p.bias0 || p.Cout2 != 32 || ((p.out2_ld | p.out2_coff) & 7) || (p.in_coff & 7)))
        return hipErrorInvalidValue;
    if (tail && dtype == VTI_H2 && p.in_coff < p.Cin) return hipErrorInvalidValue;      // h2: in_coff is y1's offset, y0 sits Cin channels below it
    // shortcut = the input (fp16 tail: in_coff was moved down to y0, the shortcut y1 sits Cin above it)
    if (p.has_res && (p.res != p.in || p.res_ld != p.in_ld || p.res_coff != p.in_coff + ((tail && dtype == VTI_F16) ? p.Cin : 0))) return hipErrorInvalidValue;
    if ((p.out_ld | p.out_coff) & (dtype == VTI_F16 ? 7 : 3)) return hipErrorInvalidValue;                              // 16-byte stores
    if (p.pk_depth < 2 || p.pk_depth > 4 || lds_bytes < bneck_pk_lds_bytes(p.TH, nrep, p.pk_depth) || lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    if (p.pk_tiles == 0) return hipSuccess;
    const int threads = 2 * (p.TH / PK_ROWS) * 64;
    dim3 grid((unsigned)p.pk_wgs, 1);
    if (dtype == VTI_F16) {
        if (nrep == 1 && tail) return launch_bneck_one<half_t, 1, true>(p, grid, threads, lds_bytes, st);
        if (nrep == 1) return launch_bneck_one<half_t, 1>(p, grid, threads, lds_bytes, st);
        if (nrep == 2) return launch_bneck_one<half_t, 2>(p, grid, threads, lds_bytes, st);
    } else if (nrep == 1) {
        if (dtype == VTI_H2 && tail) return launch_bneck_one<h2_t, 1, true>(p, grid, threads, lds_bytes, st);
        if (dtype == VTI_H2) return launch_bneck_one<h2_t, 1>(p, grid, threads, lds_bytes, st);
        return launch_bneck_one<float, 1>(p, grid, threads, lds_bytes, st);
    }
    return hipErrorInvalidValue;
}


}  // namespace vti
