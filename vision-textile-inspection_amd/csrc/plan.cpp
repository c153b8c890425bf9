// Host-side network plan for YOLOv8-seg: the fused conv table, the concat-free NHWC buffer
// graph and per-conv launch geometry.  Pure host code (no GPU calls) so it can be built and
// checked on a CPU-only box.
//
// What it stands in for: the module list Ultralytics unpickles from the .pt behind
// `YOLO(model_path)` (reference: measurement.py:145) -- yolov8-seg.yaml scaled by
// depth/width/max_channels.  Concats are never
// materialised: every producer writes straight into its consumer's channel slice.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "vti_internal.h"

namespace vti {

namespace {

struct Scale { char tag; double depth, width; int maxc; };
const Scale kScales[] = {{'n', 0.33, 0.25, 1024}, {'s', 0.33, 0.50, 1024}, {'m', 0.67, 0.75, 768},
                         {'l', 1.00, 1.00, 512},  {'x', 1.00, 1.25, 512}};

int make_divisible(double x, int d) { return (int)std::ceil(x / d) * d; }
// Python round(): half to even.
int py_round(double x) { return (int)std::nearbyint(x); }

struct Builder {
    Plan& P;
    int maxB;
    std::vector<Op>* cur;      // op group being filled (groups are ordered/laned at the end of build)
    int lane = 0;
    explicit Builder(Plan& p) : P(p), maxB(p.desc.max_batch), cur(&p.ops) {}

    int new_buf(int C, int H, int W, int elem = EL_T) {
        Buf b;
        b.C = C; b.H = H; b.W = W; b.elem = elem; b.off = 0;
        const size_t es = elem == EL_T ? (size_t)P.esize : elem == EL_F32 ? 4 : 1;
        b.bytes = (size_t)maxB * H * W * C * es;
        P.bufs.push_back(b);
        return (int)P.bufs.size() - 1;
    }
    View whole(int buf) { View v; v.buf = buf; v.coff = 0; v.C = P.bufs[buf].C; return v; }
    View slice(int buf, int coff, int C) { View v; v.buf = buf; v.coff = coff; v.C = C; return v; }

    // Adds one row to the conv table and the op that runs it.
    void conv(const std::string& name, View in, View out, int k, int s, int kind, bool out_f32 = false,
              const View* res = nullptr) {
        const Buf& ib = P.bufs[in.buf];
        ConvRow r;
        r.name = name; r.c1 = in.C; r.c2 = out.C; r.k = k; r.s = s; r.kind = kind;
        r.h_in = ib.H; r.w_in = ib.W;
        if (kind == 2) { r.h_out = 2 * ib.H; r.w_out = 2 * ib.W; }
        else { r.h_out = (ib.H + 2 * (k / 2) - k) / s + 1; r.w_out = (ib.W + 2 * (k / 2) - k) / s + 1; }
        P.convs.push_back(r);
        P.conv_out.push_back(out);
        Op op;
        op.kind = (ib.elem == EL_U8) ? OP_CONV0 : OP_CONV;
        op.conv = (int)P.convs.size() - 1;
        op.in = in; op.out = out; op.out_f32 = out_f32;
        if (res) { op.res = *res; op.has_res = true; }
        op.lane = lane;
        cur->push_back(op);
    }

    // C2f(c1, c2, n, shortcut): cv1 -> 2c; n x Bottleneck(c, c, 3x3, 3x3) chained on the last
    // chunk; cv2 over all (2+n)c channels.  Y holds every chunk so `cat` is free.
    void c2f(int idx, View in, View out, int n, bool shortcut) {
        const Buf& ib = P.bufs[in.buf];
        const int c = out.C / 2;
        const int Y = new_buf((2 + n) * c, ib.H, ib.W);
        char nm[64];
        snprintf(nm, sizeof nm, "model.%d.cv1", idx);
        conv(nm, in, slice(Y, 0, 2 * c), 1, 1, 0);
        for (int j = 0; j < n; ++j) {
            const int T = new_buf(c, ib.H, ib.W);
            View prev = slice(Y, (1 + j) * c, c);
            snprintf(nm, sizeof nm, "model.%d.m.%d.cv1", idx, j);
            conv(nm, prev, whole(T), 3, 1, 0);
            snprintf(nm, sizeof nm, "model.%d.m.%d.cv2", idx, j);
            conv(nm, whole(T), slice(Y, (2 + j) * c, c), 3, 1, 0, false, shortcut ? &prev : nullptr);
        }
        snprintf(nm, sizeof nm, "model.%d.cv2", idx);
        conv(nm, whole(Y), out, 1, 1, 0);
    }

    void up2(View in, View out) {
        Op op; op.kind = OP_UP2; op.in = in; op.out = out; op.lane = lane;
        cur->push_back(op);
    }
};

}  // namespace

// Pick the launch geometry for one conv.  Templates fix MREP=5 (80 pixels per wave along M);
// WN in {1,2,4} splits the 4 waves between pixels and couts; the pixel tile is TH x TW.
// Geometry for the persistent LDS-DMA kernel (conv_pk.hip; 3x3 stride 1): 20-wide tiles of 4 rows per M-wave.
// Cost model (seconds, rough): a layer takes max(memory time, compute-wave time) plus a start-up term.
//  * memory: patch reads incl. halo rows/columns (once per n-group) + output writes at ~5 TB/s;
//  * compute: per tile and wave nchunks x 45 x NREP MFMAs of 16 cycles + ~1100 x NREP epilogue cycles, times the
//    waves that share the busiest SIMD, times the rounds of tiles per workgroup slot, at ~1.9 GHz;
//  * LDS operand reads per MFMA (1/NREP + 1/5) stretch the MFMA phase once they pass ~0.5.
// 4-byte storage (fp32 / h2): K chunks are 16 channels, so a layer has twice the chunks and twice the weight bytes of the fp16 engine and
// its weights rarely fit beside the patches; streamed with every patch they are 60-70 % of a step's LDS-DMA bytes.  The search therefore
// also tries `wstat` -- ALL chunks of a workgroup's n-group resident (which favours splitting N across workgroups: a 2-tile slice of a
// 64 -> 64 layer is 72 KB) -- and prices the L2 -> LDS ingest (patches once per n-group + weights per tile, streamed, or per workgroup,
// resident) at 8 TB/s: measured, 173 MB of ingest pass in 30 us, so it seldom binds; what does decide these layers is whether TWO
// workgroups fit a CU (`solo` below).
static bool choose_pk_cfg(int dtype, const ConvRow& r, int max_batch, ConvCfg& c, int fth, int fwn, int fnrep) {
    const char* no = getenv("VTI_NO_PK");
    if (no && no[0] == '1') return false;
    if (!(r.k == 3 && r.s == 1 && r.kind != 2) || r.w_out < 20) return false;
    const int esize = dtype == VTI_F16 ? 2 : 4;
    static const bool wide16 = getenv("VTI_PK_WIDE16") && getenv("VTI_PK_WIDE16")[0] == '1';      // A/B aid: the same search for fp16
    const bool wide = dtype != VTI_F16 || wide16;              // 4-byte elements: ingest-priced search with resident-weight variants
    const double mf = dtype == VTI_H2 ? 2.0 : dtype == VTI_F32 ? 8.0 : 1.0;      // matrix-pipe cycles per (n-tile, m-tile, tap) in units of 16
    static const bool no_wstat = getenv("VTI_NO_PK_WSTAT") && getenv("VTI_NO_PK_WSTAT")[0] == '1';
    double best = 1e30;
    bool found = false;
    const int tiles_x = (r.w_out + 19) / 20;
    for (int WN = 1; WN <= 4; WN *= 2) {
        if (fwn && WN != fwn) continue;
        for (int NREP = 1; NREP <= 5; ++NREP) {
            if (fnrep && NREP != fnrep) continue;
            if (!conv_pk_instantiated(NREP, WN)) continue;
            static const bool h2_no_n5 = getenv("VTI_H2_NO_N5") && getenv("VTI_H2_NO_N5")[0] == '1';      // A/B aid (h2 at NREP = 5 runs h2_taps_nmajor)
            if (dtype == VTI_H2 && NREP == 5 && !fnrep && h2_no_n5) continue;
            const int NTB = WN * NREP;
            const int gy = (c.ntiles_n + NTB - 1) / NTB;
            const double n_eff = (double)c.ntiles_n / (gy * NTB);
            if (n_eff < 0.74 && !fnrep) continue;
            for (int NWM = 1; NWM <= 4; ++NWM) {
                const int TH = 4 * NWM, ncomp = NWM * WN;
                if (fth && TH != fth) continue;
                for (int wstat = (wide && !no_wstat && c.nchunks > 2) ? 1 : 0; wstat >= 0; --wstat) {
                    if (ncomp > 4 || !conv_pk_fits(TH, WN, NREP, c.nchunks, wstat)) continue;
                    const size_t lds = conv_pk_lds_bytes(TH, WN, NREP, c.nchunks, 2, wstat);
                    const int wgpc = (int)std::min<size_t>(2, (160 * 1024) / lds);
                    const int tiles_y = (r.h_out + TH - 1) / TH;
                    const long NT = (long)max_batch * tiles_y * tiles_x;
                    long G = std::min<long>(NT, std::max(1, 256 * wgpc / gy));
                    if (G >= 8) G &= ~7L;
                    const long rounds = (NT + G - 1) / G;
                    const int simd_load = (ncomp * wgpc + 3) / 4;
                    const double lds_reads = (1.0 / NREP + 0.2) / mf;
                    // NREP = 5 keeps 100 accumulator registers and a shallow operand queue: measured 12 % behind two n-groups of 3
                    // (tools/pk_sweep.py: 64 -> 80 at 80 x 80: 61 vs 54 us), hence the factor (fitted so that the model flips where the sweep does)
                    // 4-byte storage: a lone compute wave keeps its SIMD's matrix pipe ~70 % busy inside the MFMA phase (operand preparation,
                    // LDS reads and waits issue in order with the MFMAs: stamps), a second resident workgroup fills those slots -- measured
                    // (tools/pk_sweep.py h2): 8 x 20 tiles with 66 KB of LDS, i.e. two workgroups per CU, beat every 1-per-CU geometry by
                    // 1.25-1.5x on the 32- and 64-output-channel layers
                    const double solo = wide ? ((ncomp * wgpc > 4) ? 1.1 : 1.45) : 1.0;
                    const double mfma_cyc = (double)c.nchunks * 45 * NREP * 16 * mf * solo * std::max(1.0, lds_reads / 0.5) * (NREP == 5 ? 1.6 : 1.0);
                    const double t_comp = rounds * (mfma_cyc * simd_load + 1100.0 * NREP) / 1.9e9;   // a SIMD partner's MFMAs hide the epilogue
                    const double bytes = (double)NT * ((double)gy * (TH + 2) * 22 * r.c1 + (double)TH * 20 * r.c2) * esize;
                    const double t_mem = bytes / 5.0e12;
                    double t_ing = 0.0;
                    if (wide) {
                        const double patch = (double)(TH + 2) * 24 * 64 * c.nchunks, wbytes = (double)NTB * 9 * 1024 * c.nchunks;
                        const bool resident = wstat || c.nchunks <= 2;
                        t_ing = ((double)NT * gy * (patch + (resident ? 0.0 : wbytes)) + (resident ? (double)G * gy * wbytes : 0.0)) / 8.0e12;
                    }
                    // every further n-group walks the whole K range again for its 16 NTB channels (patch DMA, pixel-operand reads, barriers): measured
                    // on 256 -> 80 at 20x20, five 1-tile groups 61 us against 50 us for two groups of three
                    const double cost = (std::max(std::max(t_comp, t_mem), t_ing) + 0.15 * std::min(t_comp, t_mem)) * (wide ? 1.0 + 0.06 * (gy - 1) : 1.0) + 4e-6;
                    if (cost < best) {
                        best = cost; found = true;
                        c.TH = TH; c.TW = 20; c.WN = WN; c.NREP = NREP; c.lds = lds; c.pk = 1; c.pk_wgpc = wgpc; c.pk_wstat = wstat;
                    }
                }
            }
        }
    }
    if (found) {
        c.ntiles_n = (c.ntiles_n + c.WN * c.NREP - 1) / (c.WN * c.NREP) * (c.WN * c.NREP);   // whole workgroup n-groups
        // ring depth: a deeper ring (loaders further ahead) when it costs no co-resident workgroup
        c.pk_depth = 2;
        const int dmax = conv_pk_depth(c.TH, c.WN, c.NREP, c.nchunks, c.pk_wstat);
        for (int dd = 3; dd <= dmax; ++dd) {
            const size_t l = conv_pk_lds_bytes(c.TH, c.WN, c.NREP, c.nchunks, dd, c.pk_wstat);
            if ((int)std::min<size_t>(2, (160 * 1024) / l) >= c.pk_wgpc) { c.pk_depth = dd; c.lds = l; }
        }
    }
    return found;
}

// Stride-2 3x3 convs on the persistent schedule (conv3_pk<..., S = 2>): same cost model, the patch is (2 TH + 1) x 41 input pixels.
static bool choose_pk2_cfg(int dtype, const ConvRow& r, int max_batch, ConvCfg& c, int fth, int fwn, int fnrep) {
    const char* no = getenv("VTI_NO_PK2");
    if (no && no[0] == '1') return false;
    if (!(r.k == 3 && r.s == 2 && r.kind == 0) || r.w_out < 20) return false;
    const int esize = dtype == VTI_F16 ? 2 : 4;
    static const bool wide16 = getenv("VTI_PK_WIDE16") && getenv("VTI_PK_WIDE16")[0] == '1';
    const bool wide = dtype != VTI_F16 || wide16;
    const double mf = dtype == VTI_H2 ? 2.0 : dtype == VTI_F32 ? 8.0 : 1.0;
    static const bool no_wstat = getenv("VTI_NO_PK_WSTAT") && getenv("VTI_NO_PK_WSTAT")[0] == '1';
    double best = 1e30;
    bool found = false;
    const int tiles_x = (r.w_out + 19) / 20;
    for (int WN = 1; WN <= 4; WN *= 2) {
        if (fwn && WN != fwn) continue;
        for (int NREP = 1; NREP <= 4; ++NREP) {
            if (fnrep && NREP != fnrep) continue;
            if (!conv_pk2_instantiated(NREP, WN)) continue;
            const int NTB = WN * NREP;
            const int gy = (c.ntiles_n + NTB - 1) / NTB;
            const double n_eff = (double)c.ntiles_n / (gy * NTB);
            if (n_eff < 0.74 && !fnrep) continue;
            for (int NWM = 1; NWM <= 4; ++NWM) {
                const int TH = 4 * NWM, ncomp = NWM * WN;
                if (fth && TH != fth) continue;
                for (int wstat = (wide && !no_wstat && c.nchunks > 2) ? 1 : 0; wstat >= 0; --wstat) {
                    if (ncomp > 4 || !conv_pk2_fits(TH, WN, NREP, c.nchunks, wstat)) continue;
                    const int tiles_y = (r.h_out + TH - 1) / TH;
                    const long NT = (long)max_batch * tiles_y * tiles_x;
                    long G = std::min<long>(NT, std::max(1, 256 / gy));
                    if (G >= 8) G &= ~7L;
                    const long rounds = (NT + G - 1) / G;
                    const int simd_load = (ncomp + 3) / 4;
                    const double lds_reads = (1.0 / NREP + 0.2) / mf;
                    const double mfma_cyc = (double)c.nchunks * 45 * NREP * 16 * mf * std::max(1.0, lds_reads / 0.5);
                    const double t_comp = rounds * (mfma_cyc * simd_load + 1100.0 * NREP) / 1.9e9;
                    const double bytes = (double)NT * ((double)gy * (2 * TH + 1) * 41 * r.c1 + (double)TH * 20 * r.c2) * esize;
                    const double t_mem = bytes / 5.0e12;
                    double t_ing = 0.0;
                    if (wide) {
                        const double patch = (double)(2 * TH + 1) * 48 * 64 * c.nchunks, wbytes = (double)NTB * 9 * 1024 * c.nchunks;
                        const bool resident = wstat || c.nchunks <= 2;
                        t_ing = ((double)NT * gy * (patch + (resident ? 0.0 : wbytes)) + (resident ? (double)G * gy * wbytes : 0.0)) / 8.0e12;
                    }
                    const double cost = std::max(std::max(t_comp, t_mem), t_ing) + 0.15 * std::min(t_comp, t_mem) + 4e-6;
                    if (cost < best) {
                        best = cost; found = true;
                        c.TH = TH; c.TW = 20; c.WN = WN; c.NREP = NREP; c.pk = 4; c.pk_wgpc = 1; c.pk_wstat = wstat;
                    }
                }
            }
        }
    }
    if (found) {
        c.ntiles_n = (c.ntiles_n + c.WN * c.NREP - 1) / (c.WN * c.NREP) * (c.WN * c.NREP);
        c.pk_depth = conv_pk2_depth(c.TH, c.WN, c.NREP, c.nchunks, c.pk_wstat);
        c.lds = conv_pk2_lds_bytes(c.TH, c.WN, c.NREP, c.nchunks, c.pk_depth, c.pk_wstat);
    }
    return found;
}

// Geometry for the persistent 1x1 kernel (conv1_pk): tiles of (M-waves x 80) consecutive pixels, a deep stage ring,
// weights stationary in LDS when all K chunks of the n-group fit in 64 KB.  These layers are HBM-bound: prefer one
// n-group (the pixels are read once), then the deepest ring, then the fewest rounds.
static bool choose_pk1_cfg(int esize, const ConvRow& r, int max_batch, ConvCfg& c, int fth, int fwn, int fnrep) {
    const int cps = esize == 2 ? 1 : 2;        // K chunks per step (conv_pk.hip: launch_pk1_one instantiates the same value per type)
    const char* no = getenv("VTI_NO_PK1");
    if (no && no[0] == '1') return false;
    const bool deconv = r.kind == 2;
    if (!((r.k == 1 && r.s == 1) || deconv)) return false;
    // measured (tools/pk1_sweep.py, bs=64): 15-27 % faster than the per-tile kernel on the 160/80/40-wide maps, slower on the
    // 20 x 20 maps (a tile's worth of pixels per CU: nothing to pipeline) and on the ConvTranspose scatter epilogue
    const bool forced = fth || fwn || fnrep;
    const char* all1 = getenv("VTI_PK1_ALL");
    if (!forced && !(all1 && all1[0] == '1') && (deconv || r.h_out * r.w_out < 1600)) return false;
    const long total_px = (long)max_batch * (deconv ? r.h_in * r.w_in : r.h_out * r.w_out);
    double best = 1e30;
    bool found = false;
    for (int WN = 1; WN <= 4; WN *= 2) {
        if (fwn && WN != fwn) continue;
        for (int NREP = 1; NREP <= 5; ++NREP) {
            if (fnrep && NREP != fnrep) continue;
            if (!conv1_pk_instantiated(NREP, WN)) continue;
            if (deconv && r.c2 % (16 * NREP)) continue;      // a lane's channel run must stay inside one (dy,dx) plane
            const int NTB = WN * NREP;
            const int gy = (c.ntiles_n + NTB - 1) / NTB;
            const double n_eff = (double)c.ntiles_n / (gy * NTB);
            if (n_eff < 0.74 && !fnrep) continue;
            for (int nwm = 1; nwm <= 4; ++nwm) {
                if (fth && nwm != fth) continue;
                const int ncomp = nwm * WN;
                if (ncomp > 4) continue;
                // stationary weights up to 96 KB: streaming them with the pixels takes LDS-DMA ingest from the pixels (model.12.cv1, 384 -> 128 at
                // 40x40: 8 of every 18 KB per step; 36.5 -> 32.6 us with 96 KB stationary and a 6-deep ring).  VTI_PK1_WSTAT_KB: tools/pk1_sweep.py
                static const size_t wstat_max = getenv("VTI_PK1_WSTAT_KB") ? (size_t)atoi(getenv("VTI_PK1_WSTAT_KB")) * 1024 : 96 * 1024;
                const int wstat = (size_t)c.nchunks * NTB * 1024 <= wstat_max ? 1 : 0;
                // one workgroup per CU with the deepest ring that fits -- or (4-byte storage) two co-resident workgroups with half the LDS each:
                // the same bytes in flight per CU, and one workgroup's MFMAs + epilogue run under the other's waits (as for conv3_pk: `solo`)
                static const bool no_w2 = getenv("VTI_PK1_NO_WGPC2") && getenv("VTI_PK1_NO_WGPC2")[0] == '1';
                static const bool force_w2 = getenv("VTI_PK1_FORCE_WGPC2") && getenv("VTI_PK1_FORCE_WGPC2")[0] == '1';     // A/B aid
                for (int wgpc = (esize == 4 && force_w2) ? 2 : 1; wgpc <= ((esize == 4 && !no_w2) ? 2 : 1); ++wgpc) {
                    int depth = 0;
                    for (int dd = 8; dd >= 2; --dd)
                        if (conv1_pk_fits(nwm, WN, NREP, c.nchunks, dd, wstat, cps) &&
                            conv1_pk_lds_bytes(nwm, WN, NREP, c.nchunks, dd, wstat, cps) <= (size_t)(160 * 1024) / wgpc) { depth = dd; break; }
                    if (!depth) continue;
                    const long NT = (total_px + nwm * 80 - 1) / (nwm * 80);
                    long G = std::min<long>(NT, std::max(1, 256 * wgpc / gy));
                    if (G >= 8) G &= ~7L;
                    const long rounds = (NT + G - 1) / G;
                    const double solo = esize == 4 ? ((ncomp * wgpc > 4) ? 1.1 : 1.45) : 1.0;
                    const double step_cyc = 5.0 * NREP * 16 * solo * (esize == 4 ? 2.0 : 1.0) + 250.0 / cps;
                    const double t_comp = rounds * (c.nchunks * step_cyc + 1100.0 * NREP) * ((ncomp * wgpc + 3) / 4) / 1.9e9;
                    const double bytes = (double)total_px * ((double)gy * r.c1 + (double)c.gemm_n) * esize +
                                         (wstat ? 0.0 : (double)NT * gy * c.nchunks * NTB * 1024 * 0.25);   // streamed weights (L2)
                    const double inflight = (double)wgpc * (depth - 1) * nwm * 80 * 64 * cps;          // bytes in flight per CU
                    static const double inflight_full = getenv("VTI_PK1_INFLIGHT_KB") ? atof(getenv("VTI_PK1_INFLIGHT_KB")) * 1e3 : 50e3;       // bytes in flight per CU that saturate the ingest (measured: ~50 KB)
                    const double t_mem = bytes / (5.0e12 * std::min(1.0, inflight / inflight_full));
                    const double cost = std::max(t_comp, t_mem) + 0.15 * std::min(t_comp, t_mem) + 4e-6;
                    if (cost < best) {
                        best = cost; found = true;
                        c.TH = nwm; c.TW = 80; c.WN = WN; c.NREP = NREP; c.pk = 2; c.pk_wgpc = wgpc; c.pk_depth = depth; c.pk_wstat = wstat; c.pk_cps = cps;
                        c.lds = conv1_pk_lds_bytes(nwm, WN, NREP, c.nchunks, depth, wstat, cps);
                    }
                }
            }
        }
    }
    if (found) c.ntiles_n = (c.ntiles_n + c.WN * c.NREP - 1) / (c.WN * c.NREP) * (c.WN * c.NREP);
    return found;
}

void choose_conv_cfg(int dtype, const ConvRow& r, bool conv0, int max_batch, ConvCfg& c, int fth, int ftw, int fwn,
                     int fnrep, bool allow_pk) {
    const bool f16 = dtype == VTI_F16;
    const int KC = f16 ? 32 : 16;
    const bool deconv = r.kind == 2;
    c.gemm_n = deconv ? 4 * r.c2 : r.c2;
    c.ntiles_n = (c.gemm_n + 15) / 16;      // rounded up to whole NREP groups once NREP is chosen (below)
    c.nchunks = conv0 ? (32 / KC) : (r.c1 + KC - 1) / KC;
    c.pk = 0;
    if (allow_pk && !conv0 && (!ftw || ftw == 80) && (!fth || fth <= 4) && choose_pk1_cfg(f16 ? 2 : 4, r, max_batch, c, fth, fwn, fnrep)) return;
    c.pk_wstat = 0;
    if (allow_pk && !conv0 && (!ftw || ftw == 20) && (!fth || fth % 4 == 0) && choose_pk_cfg(dtype, r, max_batch, c, fth, fwn, fnrep)) {
        return;
    }
    if (allow_pk && !conv0 && (!ftw || ftw == 20) && (!fth || fth % 4 == 0) && choose_pk2_cfg(dtype, r, max_batch, c, fth, fwn, fnrep)) return;
    const int ks = deconv ? 1 : r.k, st = deconv ? 1 : r.s;
    const int Ho = deconv ? r.h_in : r.h_out, Wo = deconv ? r.w_in : r.w_out;
    c.TH = c.TW = 0;

    double best = -1;
    for (int WN = 1; WN <= 4; WN *= 2) {
        if (fwn && WN != fwn) continue;
        for (int NREP = 1; NREP <= 5; ++NREP) {
            if (fnrep && NREP != fnrep) continue;
            if (deconv && r.c2 % (16 * NREP)) continue;  // a lane's channel run must stay inside one (dy,dx) plane
            const int BN = WN * NREP;                    // n-tiles per workgroup
            const int gy = (c.ntiles_n + BN - 1) / BN;
            const double n_eff = (double)c.ntiles_n / (gy * BN);
            if (n_eff < 0.74 && !fnrep) continue;
            const int BM = (4 / WN) * 80;
            for (int TW = std::min(Wo, BM); TW >= 1; --TW) {   // widest first: ties keep row-contiguous tiles
                if (ftw && TW != ftw) continue;
                int TH = std::min(Ho, BM / TW);
                if (fth) { if (fth * TW > BM) continue; TH = fth; }
                if (TH < 1) continue;
                const int tiles = ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW);
                const double m_eff = (double)Ho * Wo / ((double)tiles * BM);
                const int PH = conv0 ? TH : (TH - 1) * st + ks, PW = conv0 ? TW : (TW - 1) * st + ks;
                const size_t lds = conv_lds_bytes(ks, st, conv0 ? 1 : 0, TH, TW, WN, NREP);
                if (lds > ((fth || ftw) ? 160u : 80u) * 1024) continue;
                if (!conv_cfg_fits(ks, st, conv0 ? 1 : 0, TH, TW, WN, NREP)) continue;
                // score: MFMA efficiency, mild preference for compact input patches (halo re-reads),
                // for bigger per-wave register tiles (LDS traffic per MFMA ~ 1/NREP + 1/5), for long
                // contiguous tile rows (coalescing) and for >= 2 workgroups per CU of LDS.
                const double halo = conv0 ? (double)(4 * TH * TW) / ((2 * TH + 1) * (2 * TW + 1))
                                          : (double)(TH * TW * st * st) / (PH * PW);
                const double lds_traffic = 1.0 / NREP + 1.0 / 5;
                const double rowb = std::min(1.0, (double)TW * 64.0 / 1024.0);
                double score = m_eff * n_eff * (0.6 + 0.4 * halo) * (0.8 + 0.2 * rowb) / (0.35 + lds_traffic);
                if (lds > 64 * 1024) score *= 0.9;
                if (ks == 1 && WN >= 2) score *= 1.12;      // measured: 1x1 / deconv prefer more, smaller workgroups
                const double wgs = (double)tiles * gy * max_batch;
                if (wgs < 512) score *= 0.5 + 0.5 * wgs / 512;
                if (score > best) {
                    best = score;
                    c.TH = TH; c.TW = TW; c.WN = WN; c.NREP = NREP; c.lds = lds;
                }
            }
        }
    }
    if (c.TH) c.ntiles_n = (c.ntiles_n + c.NREP - 1) / c.NREP * c.NREP;   // whole channel groups (row permutation)
}

std::string Plan::build(const vti_desc& d) {
    desc = d;
    const Scale* sc = nullptr;
    for (const Scale& s : kScales) if (s.tag == d.scale) sc = &s;
    if (!sc) return "unknown scale (expected one of n,s,m,l,x)";
    if (d.nc < 1 || d.nm < 1 || d.reg_max < 1 || d.nm % 4 || d.nm > 64) return "bad nc/nm/reg_max";
    if (d.reg_max != 16) return "only reg_max=16 is supported";
    if (d.H < 32 || d.W < 32 || d.H % 32 || d.W % 32) return "H and W must be positive multiples of 32";
    if (d.max_batch < 1) return "max_batch must be >= 1";
    if (d.dtype != VTI_F16 && d.dtype != VTI_F32 && d.dtype != VTI_H2) return "dtype must be VTI_F16, VTI_F32 or VTI_H2";
    esize = d.dtype == VTI_F16 ? 2 : 4;

    auto ch = [&](int c) { return make_divisible(std::min(c, sc->maxc) * sc->width, 8); };
    auto rep = [&](int n) { return std::max(py_round(n * sc->depth), 1); };
    const int c0 = ch(64), c1 = ch(128), c2 = ch(256), c3 = ch(512), c4 = ch(1024);
    const int r0 = rep(3), r1 = rep(6), r2 = rep(6), r3 = rep(3), rn = rep(3);
    const int npr = ch(256);
    const int c_box = std::max(std::max(16, c2 / 4), 4 * d.reg_max);
    const int c_cls = std::max(c2, std::min(d.nc, 100));
    const int c_mc = std::max(c2 / 4, d.nm);
    for (int c : {c0, c1, c2, c3, c4, npr, c_box, c_cls, c_mc})
        if (c % 16) return "channel count not a multiple of 16 for this scale";

    Builder b(*this);
    // Op groups.  Frames are independent and so are the branches that hang off P3/P4/P5: the proto chain
    // and the three head levels run on side streams (lanes) forked when their input is ready and joined
    // before the decode, so the many small 40x40 / 20x20 kernels overlap instead of queueing.
    std::vector<Op> g_p3, g_proto, g_head[3], g_p4, g_p5;
    b.cur = &g_p3;
    const int H = d.H, W = d.W;
    const int in_u8 = b.new_buf(3, H, W, EL_U8);         // caller's letterboxed frames (not in workspace)
    const int B0 = b.new_buf(c0, H / 2, W / 2);
    const int B1 = b.new_buf(c1, H / 4, W / 4);
    const int B2 = b.new_buf(c1, H / 4, W / 4);
    const int B3 = b.new_buf(c2, H / 8, W / 8);
    const int CAT14 = b.new_buf(c3 + c2, H / 8, W / 8);   // [up(x12), x4]
    const int B5 = b.new_buf(c3, H / 16, W / 16);
    const int CAT11 = b.new_buf(c4 + c3, H / 16, W / 16); // [up(x9), x6]
    const int B7 = b.new_buf(c4, H / 32, W / 32);
    const int B8 = b.new_buf(c4, H / 32, W / 32);
    const int SP = b.new_buf(2 * c4, H / 32, W / 32);     // SPPF: [cv1, p5, p9, p13]
    const int CAT20 = b.new_buf(c3 + c4, H / 32, W / 32); // [conv19(p4), x9]
    const int CAT17 = b.new_buf(c2 + c3, H / 16, W / 16); // [conv16(p3), x12]
    const int P3 = b.new_buf(c2, H / 8, W / 8);
    const int P4 = b.new_buf(c3, H / 16, W / 16);
    const int P5 = b.new_buf(c4, H / 32, W / 32);

    b.conv("model.0", b.whole(in_u8), b.whole(B0), 3, 2, 0);
    b.conv("model.1", b.whole(B0), b.whole(B1), 3, 2, 0);
    b.c2f(2, b.whole(B1), b.whole(B2), r0, true);
    b.conv("model.3", b.whole(B2), b.whole(B3), 3, 2, 0);
    const View x4 = b.slice(CAT14, c3, c2);
    b.c2f(4, b.whole(B3), x4, r1, true);
    b.conv("model.5", x4, b.whole(B5), 3, 2, 0);
    const View x6 = b.slice(CAT11, c4, c3);
    b.c2f(6, b.whole(B5), x6, r2, true);
    b.conv("model.7", x6, b.whole(B7), 3, 2, 0);
    b.c2f(8, b.whole(B7), b.whole(B8), r3, true);
    // SPPF
    const View x9 = b.slice(CAT20, c3, c4);
    b.conv("model.9.cv1", b.whole(B8), b.slice(SP, 0, c4 / 2), 1, 1, 0);
    { Op op; op.kind = OP_POOL; op.in = b.slice(SP, 0, c4 / 2); op.out = b.slice(SP, c4 / 2, 3 * (c4 / 2)); b.cur->push_back(op); }
    b.conv("model.9.cv2", b.whole(SP), x9, 1, 1, 0);
    // neck
    b.up2(x9, b.slice(CAT11, 0, c4));
    const View x12 = b.slice(CAT17, c2, c3);
    b.c2f(12, b.whole(CAT11), x12, rn, false);
    b.up2(x12, b.slice(CAT14, 0, c3));
    b.c2f(15, b.whole(CAT14), b.whole(P3), rn, false);
    b.cur = &g_p4;
    b.conv("model.16", b.whole(P3), b.slice(CAT17, 0, c2), 3, 2, 0);
    b.c2f(18, b.whole(CAT17), b.whole(P4), rn, false);
    b.cur = &g_p5;
    b.conv("model.19", b.whole(P4), b.slice(CAT20, 0, c3), 3, 2, 0);
    b.c2f(21, b.whole(CAT20), b.whole(P5), rn, false);
    // Segment head
    const int feat[3] = {P3, P4, P5};
    const int strides[3] = {8, 16, 32};
    num_anchors = 0;
    // lanes: level 0 is one lane (its kernels fill the chip); the small level-1/2 towers get a lane each
    // VTI_P3_LANES=1 (A/B aid): the three P3 towers on three lanes instead of one
    static const bool p3l = getenv("VTI_P3_LANES") && getenv("VTI_P3_LANES")[0] == '1';
    const int tower_lane[3][3] = {{2, p3l ? 8 : 2, p3l ? 9 : 2}, {3, 4, 5}, {6, 7, 0}};
    for (int l = 0; l < 3; ++l) {
        b.cur = &g_head[l];
        const Buf fb = bufs[feat[l]];
        Level lv; lv.C = fb.C; lv.H = fb.H; lv.W = fb.W; lv.stride = strides[l];
        const char* towers[3] = {"cv2", "cv3", "cv4"};
        const int cmid[3] = {c_box, c_cls, c_mc};
        const int cout[3] = {4 * d.reg_max, d.nc, d.nm};
        int outs[3];
        for (int t = 0; t < 3; ++t) {
            b.lane = tower_lane[l][t];
            const int t1 = b.new_buf(cmid[t], fb.H, fb.W), t2 = b.new_buf(cmid[t], fb.H, fb.W);
            outs[t] = b.new_buf(cout[t], fb.H, fb.W, EL_F32);
            char nm[64];
            snprintf(nm, sizeof nm, "model.22.%s.%d.0", towers[t], l);
            b.conv(nm, b.whole(feat[l]), b.whole(t1), 3, 1, 0);
            snprintf(nm, sizeof nm, "model.22.%s.%d.1", towers[t], l);
            b.conv(nm, b.whole(t1), b.whole(t2), 3, 1, 0);
            snprintf(nm, sizeof nm, "model.22.%s.%d.2", towers[t], l);
            b.conv(nm, b.whole(t2), b.whole(outs[t]), 1, 1, 1, true);
        }
        lv.box_buf = outs[0]; lv.cls_buf = outs[1]; lv.mc_buf = outs[2];
        levels.push_back(lv);
        num_anchors += fb.H * fb.W;
    }
    b.cur = &g_proto;
    b.lane = 1;
    {
        const int pc1 = b.new_buf(npr, H / 8, W / 8), pup = b.new_buf(npr, H / 4, W / 4);
        // the caller's proto is T for the fp16 / fp32 engines and f32 for the h2 engine (its pairs are an internal storage format)
        const bool proto_f32 = d.dtype == VTI_H2;
        const int pc2 = b.new_buf(npr, H / 4, W / 4), pout = b.new_buf(d.nm, H / 4, W / 4, proto_f32 ? EL_F32 : EL_T);
        b.conv("model.22.proto.cv1", b.whole(P3), b.whole(pc1), 3, 1, 0);
        b.conv("model.22.proto.upsample", b.whole(pc1), b.whole(pup), 2, 2, 2);
        b.conv("model.22.proto.cv2", b.whole(pup), b.whole(pc2), 3, 1, 0);
        b.conv("model.22.proto.cv3", b.whole(pc2), b.whole(pout), 1, 1, 0, proto_f32);
        proto_buf_c = pout;   // replaced by the caller's proto pointer at run time
    }
    {   // final op order: lane 0 carries backbone + neck (+ the P5 head); side lanes fork off it
        auto sync = [&](OpKind k, int lane) { Op op; op.kind = k; op.lane = lane; ops.push_back(op); };
        auto add = [&](const std::vector<Op>& g) { ops.insert(ops.end(), g.begin(), g.end()); };
        add(g_p3);
        sync(OP_FORK, 1); sync(OP_FORK, 2);
        if (p3l) { sync(OP_FORK, 8); sync(OP_FORK, 9); }
        add(g_proto); add(g_head[0]);
        add(g_p4);
        sync(OP_FORK, 3); sync(OP_FORK, 4); sync(OP_FORK, 5);
        add(g_head[1]);
        add(g_p5);
        sync(OP_FORK, 6); sync(OP_FORK, 7);
        add(g_head[2]);
        for (int l = 1; l <= (p3l ? 9 : 7); ++l) sync(OP_JOIN, l);
        Op op; op.kind = OP_DECODE; ops.push_back(op);
    }

    // workspace layout: plain bump allocation, 256-B aligned; buffer 0 (u8 input) and the
    // proto output belong to the caller.
    size_t off = 0;
    for (size_t i = 0; i < bufs.size(); ++i) {
        if ((int)i == in_u8 || (int)i == proto_buf_c) { bufs[i].off = 0; continue; }
        bufs[i].off = off;
        off += (bufs[i].bytes + 255) & ~(size_t)255;
    }
    ws_bytes = off;

    // fuse "3x3 conv -> 1x1 conv" pairs (head towers .1 -> .2, proto cv2 -> cv3): the 1x1 runs on the
    // producer's register tile, its op disappears and the intermediate never reaches HBM.
    {
        const char* nf = getenv("VTI_NO_FUSE");
        for (size_t i = 0; !(nf && nf[0] == '1') && i + 1 < ops.size(); ++i) {
            Op& a = ops[i];
            const Op& b2 = ops[i + 1];
            if (a.kind != OP_CONV || b2.kind != OP_CONV || a.lane != b2.lane) continue;
            const ConvRow& ra = convs[a.conv];
            const ConvRow& rb = convs[b2.conv];
            if (!(ra.k == 3 && ra.s == 1 && ra.kind == 0 && rb.k == 1 && rb.s == 1 && rb.kind != 2)) continue;
            if (a.has_res || b2.has_res || a.out_f32) continue;
            if (b2.in.buf != a.out.buf || b2.in.coff != a.out.coff || b2.in.C != a.out.C || a.out.C != bufs[a.out.buf].C) continue;
            if (ra.c2 % 16 || !conv_fusable(ra.c2 / 16, (rb.c2 + 15) / 16)) continue;
            bool other_reader = false;          // the intermediate must have no other consumer
            for (size_t j = 0; j < ops.size(); ++j)
                if (j != i + 1 && (ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                    (ops[j].in.buf == a.out.buf || (ops[j].has_res && ops[j].res.buf == a.out.buf))) other_reader = true;
            if (other_reader) continue;
            a.fused = b2.conv; a.out2 = b2.out; a.out2_f32 = b2.out_f32;
            conv_out[a.conv].buf = -1;          // not materialised any more
            ops.erase(ops.begin() + i + 1);
        }
    }

    // Fold "ConvTranspose2d(2,2) -> 3x3 conv (+ fused 1x1)" (proto.upsample -> proto.cv2 -> proto.cv3) into ONE kernel of four
    // 2x2 convs on the low-resolution map (weights.cpp: pack_conv_fold): the deconv's 4x larger output tensor is neither
    // written nor read and the pair costs 2.25x fewer MACs.
    {
        const char* nf = getenv("VTI_NO_FOLD");
        for (size_t i = 0; !(nf && nf[0] == '1') && i + 1 < ops.size(); ++i) {
            const Op& u = ops[i];
            Op& v = ops[i + 1];
            if (u.kind != OP_CONV || v.kind != OP_CONV || u.lane != v.lane || v.fused < 0 || u.fused >= 0) continue;
            const ConvRow& ru = convs[u.conv];
            const ConvRow& rv = convs[v.conv];
            if (!(ru.kind == 2 && ru.k == 2 && ru.s == 2 && rv.kind == 0 && rv.k == 3 && rv.s == 1) || u.has_res || v.has_res) continue;
            if (v.in.buf != u.out.buf || v.in.coff != u.out.coff || v.in.C != u.out.C || u.out.C != bufs[u.out.buf].C) continue;
            if (u.in.coff != 0 || u.in.C != bufs[u.in.buf].C) continue;
            if (!convfold_supported(ru.c1, ru.c2, rv.c2, (convs[v.fused].c2 + 15) / 16)) continue;
            bool other_reader = false;
            for (size_t j = 0; j < ops.size(); ++j)
                if (j != i + 1 && (ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                    (ops[j].in.buf == u.out.buf || (ops[j].has_res && ops[j].res.buf == u.out.buf))) other_reader = true;
            if (other_reader) continue;
            v.fold = u.conv;
            v.in = u.in;
            conv_out[u.conv].buf = -1;          // the upsampled tensor does not exist any more
            ops.erase(ops.begin() + i);
        }
    }

    // Fuse the two 3x3 convs of a C2f Bottleneck (m.j.cv1 -> m.j.cv2 [+ shortcut]) into one persistent kernel when their channels
    // fit one K chunk (bneck_pk: the intermediate lives in LDS, the shortcut comes from the input patch).
    {
        const char* nb = getenv("VTI_NO_BNECK");
        const int KC = d.dtype == VTI_F16 ? 32 : 16;
        size_t pk_limit = 0x80000000ull;
        if (const char* pl = getenv("VTI_PK_LIMIT_BYTES")) pk_limit = (size_t)atoll(pl);
        for (size_t i = 0; !(nb && nb[0] == '1') && i + 1 < ops.size(); ++i) {
            Op& a = ops[i];
            const Op& b2 = ops[i + 1];
            if (a.kind != OP_CONV || b2.kind != OP_CONV || a.lane != b2.lane || a.fused >= 0 || b2.fused >= 0 || a.fold >= 0 || b2.fold >= 0) continue;
            const ConvRow& ra = convs[a.conv];
            const ConvRow& rb = convs[b2.conv];
            if (!(ra.k == 3 && ra.s == 1 && ra.kind == 0 && rb.k == 3 && rb.s == 1 && rb.kind == 0)) continue;
            if (ra.c1 != ra.c2 || rb.c1 != rb.c2 || ra.c1 != rb.c1 || ra.c1 % 16 || ra.c1 > KC || ra.c1 > 32) continue;
            if (a.has_res || a.out_f32 || b2.out_f32 || ra.w_out < 20 || ra.h_out < 8) continue;
            if (b2.in.buf != a.out.buf || b2.in.coff != a.out.coff || b2.in.C != a.out.C || a.out.C != bufs[a.out.buf].C) continue;
            if (b2.has_res && (b2.res.buf != a.in.buf || b2.res.coff != a.in.coff || b2.res.C != a.in.C)) continue;
            if (bufs[a.in.buf].bytes >= pk_limit || bufs[b2.out.buf].bytes >= pk_limit) continue;
            const int esz = d.dtype == VTI_F16 ? 2 : 4;
            if (((bufs[b2.out.buf].C | b2.out.coff) * esz) % 16) continue;        // 16-byte stores of a lane's channel run
            bool other_reader = false;
            for (size_t j = 0; j < ops.size(); ++j)
                if (j != i + 1 && (ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                    (ops[j].in.buf == a.out.buf || (ops[j].has_res && ops[j].res.buf == a.out.buf))) other_reader = true;
            if (other_reader) continue;
            a.pair = b2.conv;
            a.out = b2.out; a.has_res = b2.has_res; a.res = b2.res;
            conv_out[a.conv].buf = -1;          // the intermediate is never materialised
            ops.erase(ops.begin() + i + 1);
            // ... and, for an n = 1 C2f with c = 16 (fp16: [y0 | y1] is exactly one 64-byte slot), its closing 1x1 over [y0 | y1 | y2]:
            // the tail of bneck_pk.  Y = [y0 | y1 | y2] in one buffer; the pair reads y1 (shortcut too) and would write y2.
            const char* nt = getenv("VTI_NO_BNECK_TAIL");
            if (!(nt && nt[0] == '1') && (d.dtype == VTI_F16 || d.dtype == VTI_H2) && ra.c1 == 16 && i + 1 < ops.size()) {
                Op& a2 = ops[i];
                const Op& c3 = ops[i + 1];
                const int Y = a2.in.buf, c = ra.c1;
                if (c3.kind == OP_CONV && c3.lane == a2.lane && c3.fused < 0 && !c3.has_res && !c3.out_f32 && c3.up_C == 0) {
                    const ConvRow& rc = convs[c3.conv];
                    bool ok = rc.k == 1 && rc.s == 1 && rc.kind == 0 && rc.c1 == 3 * c && rc.c2 == 32 && bufs[Y].C == 3 * c &&
                              c3.in.buf == Y && c3.in.coff == 0 && c3.in.C == 3 * c && a2.in.coff == c && a2.out.buf == Y && a2.out.coff == 2 * c &&
                              a2.has_res && ((bufs[c3.out.buf].C | c3.out.coff) % 8) == 0 && bufs[c3.out.buf].bytes < pk_limit;
                    for (size_t j = 0; ok && j < ops.size(); ++j)     // nobody else may read y2 (it will not exist)
                        if (j != i && j != i + 1 && (ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                            (ops[j].in.buf == Y || (ops[j].has_res && ops[j].res.buf == Y)) &&
                            !(ops[j].in.buf == Y && ops[j].in.coff + ops[j].in.C <= 2 * c && !(ops[j].has_res && ops[j].res.buf == Y))) ok = false;
                    if (ok) {
                        a2.tail = c3.conv; a2.out2 = c3.out;
                        conv_out[a2.pair].buf = -1;             // y2 lives in registers only
                        ops.erase(ops.begin() + i + 1);
                    }
                }
            }
        }
    }

    // stem + layer 1 in one kernel (n-scale channel counts: 3 -> 16 -> 32): the 320x320x16 tensor stays in LDS
    {
        const char* nsf = getenv("VTI_NO_STEM_FUSE");
        if (!(nsf && nsf[0] == '1') && ops.size() >= 2 && ops[0].kind == OP_CONV0 && ops[1].kind == OP_CONV) {
            Op& a = ops[0];
            const Op& b1 = ops[1];
            const ConvRow& r0 = convs[a.conv];
            const ConvRow& r1 = convs[b1.conv];
            bool other_reader = false;
            for (size_t j = 2; j < ops.size(); ++j)
                if ((ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                    (ops[j].in.buf == a.out.buf || (ops[j].has_res && ops[j].res.buf == a.out.buf))) other_reader = true;
            if (r0.c2 == 16 && r0.k == 3 && r0.s == 2 && r1.c1 == 16 && r1.c2 == 32 && r1.k == 3 && r1.s == 2 && r1.kind == 0 &&
                !b1.has_res && !b1.out_f32 && b1.in.buf == a.out.buf && b1.in.coff == 0 && b1.in.C == 16 && a.lane == b1.lane && !other_reader) {
                a.fused_l1 = b1.conv; a.out2 = b1.out;
                conv_out[a.conv].buf = -1;          // the stem's output is never materialised
                ops.erase(ops.begin() + 1);
                // ... and the 1x1 conv that is layer 1's only consumer (model.2.cv1) runs on layer 1's register tile as a fused
                // second stage (conv_stage2<T, 2, 2>): layer 1's 160x160x32 tensor is neither written nor read back either
                const char* ns3 = getenv("VTI_NO_STEM_FUSE3");
                if (!(ns3 && ns3[0] == '1') && ops.size() >= 2 && ops[1].kind == OP_CONV) {
                    Op& s0 = ops[0];
                    const Op& c1 = ops[1];
                    const ConvRow& rc = convs[c1.conv];
                    bool other = false;
                    for (size_t j = 2; j < ops.size(); ++j)
                        if ((ops[j].kind == OP_CONV || ops[j].kind == OP_UP2 || ops[j].kind == OP_POOL) &&
                            (ops[j].in.buf == s0.out2.buf || (ops[j].has_res && ops[j].res.buf == s0.out2.buf))) other = true;
                    if (rc.k == 1 && rc.s == 1 && rc.kind == 0 && rc.c1 == 32 && rc.c2 == 32 && conv_fusable(2, 2) && !c1.has_res && !c1.out_f32 &&
                        c1.in.buf == s0.out2.buf && c1.in.coff == s0.out2.coff && c1.in.C == 32 && bufs[s0.out2.buf].C == 32 &&
                        c1.lane == s0.lane && !other) {
                        conv_out[s0.fused_l1].buf = -1;     // layer 1's output is not materialised
                        s0.fused = c1.conv; s0.out2 = c1.out;
                        ops.erase(ops.begin() + 1);
                    }
                }
            }
        }
    }

    // If every class / mask-coefficient tower ends in a fused 1x1, that stage writes its result straight into
    // pred (sigmoid in the epilogue) and the decode kernel only has to turn the box logits into boxes.
    {
        const char* ns = getenv("VTI_NO_PRED_SCATTER");
        int found = 0;
        for (Op& op : ops) {
            if (op.kind != OP_CONV || op.fused < 0) continue;
            const std::string& nm2 = convs[op.fused].name;      // model.22.cv3.<l>.2 / model.22.cv4.<l>.2
            if (nm2.rfind("model.22.cv3.", 0) == 0 || nm2.rfind("model.22.cv4.", 0) == 0) ++found;
        }
        pred_scatter = found == 6 && !(ns && ns[0] == '1');
        if (pred_scatter) {
            int a0[3] = {0, 0, 0};
            for (int l = 1; l < 3; ++l) a0[l] = a0[l - 1] + levels[l - 1].H * levels[l - 1].W;
            for (Op& op : ops) {
                if (op.kind != OP_CONV || op.fused < 0) continue;
                const std::string& nm2 = convs[op.fused].name;
                const bool is_cls = nm2.rfind("model.22.cv3.", 0) == 0, is_mc = nm2.rfind("model.22.cv4.", 0) == 0;
                if (!is_cls && !is_mc) continue;
                const int l = nm2[13] - '0';
                op.pred_mode = is_cls ? 2 : 1;
                op.pred_cbase = is_cls ? 4 : 4 + d.nc;
                op.pred_a0 = a0[l];
                conv_out[op.fused].buf = -1;     // lives in pred only
            }
            // fp16 engine: the box towers' fused 1x1 stage also does the DFL expectation + dist2bbox and writes the 4 box
            // values of its anchors -- the fp32 logits [B, HW, 64] and the decode kernel disappear.  (The fp32 parity engine keeps
            // the separate decode: its softmax sums in the reference's order.)
            const char* nd = getenv("VTI_NO_DFL_FUSE");
            int nbox = 0;
            for (Op& op : ops)
                if (op.kind == OP_CONV && op.fused >= 0 && convs[op.fused].name.rfind("model.22.cv2.", 0) == 0 && convs[op.fused].c2 == 64) ++nbox;
            dfl_fused = nbox == 3 && d.dtype != VTI_F32 && d.reg_max == 16 && !(nd && nd[0] == '1');
            if (dfl_fused) {
                for (Op& op : ops) {
                    if (op.kind != OP_CONV || op.fused < 0 || convs[op.fused].name.rfind("model.22.cv2.", 0) != 0) continue;
                    const int l = convs[op.fused].name[13] - '0';
                    op.pred_mode = 3; op.pred_cbase = 0; op.pred_a0 = a0[l]; op.dfl_stride = levels[l].stride;
                    conv_out[op.fused].buf = -1;
                }
                for (size_t i = 0; i < ops.size(); ++i)
                    if (ops[i].kind == OP_DECODE) { ops.erase(ops.begin() + i); break; }
            }
        }
    }

    // launch geometry + packed-weight offsets
    macs = 0; fused_params = d.reg_max;
    size_t woff = 0, boff = 0;
    for (Op& op : ops) {
        if (op.kind != OP_CONV && op.kind != OP_CONV0) continue;
        const ConvRow& r = convs[op.conv];
        macs += r.macs(); fused_params += r.fused_params();
        // The persistent kernels address whole tensors through ONE buffer resource with 32-bit offsets and use bit 31 as the
        // "out of range" marker, so a tensor of 2 GiB or more (at max_batch) keeps its convs on the per-tile kernel, which makes a
        // resource per frame.  Decided here, at plan time, so that nothing downstream (the folded Upsample) relies on a kernel
        // the run-time size check would refuse.  VTI_PK_LIMIT_BYTES lowers the limit (tests of this fall-back).
        size_t pk_limit = 0x80000000ull;
        if (const char* pl = getenv("VTI_PK_LIMIT_BYTES")) pk_limit = (size_t)atoll(pl);
        const bool pk_ok = bufs[op.in.buf].bytes < pk_limit && bufs[op.out.buf].bytes < pk_limit &&
                           (!op.has_res || bufs[op.res.buf].bytes < pk_limit);
        if (op.pair >= 0) {         // bneck_pk: 16 x 20 tiles (4 compute + 4 loader waves); 8 x 20 when that leaves CUs without a tile
            const ConvRow& rb = convs[op.pair];
            macs += rb.macs(); fused_params += rb.fused_params();
            const int KC = d.dtype == VTI_F16 ? 32 : 16;
            op.cfg = ConvCfg();
            const long tiles16 = (long)d.max_batch * ((r.h_out + 15) / 16) * ((r.w_out + 19) / 20);
            op.cfg.TH = tiles16 >= 256 ? 16 : 8; op.cfg.TW = 20; op.cfg.WN = 1; op.cfg.NREP = r.c2 / 16;
            op.cfg.nchunks = (r.c1 + KC - 1) / KC; op.cfg.gemm_n = r.c2; op.cfg.ntiles_n = r.c2 / 16;
            op.cfg.pk = 3; op.cfg.pk_wgpc = 1;
            op.cfg.pk_depth = bneck_pk_depth(op.cfg.TH, op.cfg.NREP);
            op.cfg.lds = bneck_pk_lds_bytes(op.cfg.TH, op.cfg.NREP, op.cfg.pk_depth);
            if (!bneck_pk_fits(op.cfg.TH, op.cfg.NREP) || op.cfg.nchunks != 1) return "no launch configuration for the fused bottleneck " + r.name;
        }
        else if (op.fold >= 0) {         // convfold_kernel: 4 x 20 low-resolution pixels per workgroup, wave = output phase, 4 n-tiles each
            const ConvRow& ru = convs[op.fold];
            macs += ru.macs(); fused_params += ru.fused_params();
            const int KC = d.dtype == VTI_F16 ? 32 : 16;
            op.cfg = ConvCfg();
            op.cfg.TH = 4; op.cfg.TW = 20; op.cfg.WN = 4; op.cfg.NREP = r.c2 / 16;
            op.cfg.nchunks = (ru.c1 + KC - 1) / KC; op.cfg.gemm_n = 4 * r.c2; op.cfg.ntiles_n = 4 * (r.c2 / 16);
            // h2: 8 x 20 tiles with 8-wave workgroups, two pixel groups sharing every staged weight chunk (conv.hip).  Alone the kernel takes the
            // same 272 us as the 4-wave form, but inside the multi-stream forward, where the proto chain runs beside the P3 towers and the trunk,
            // the forward is 1.4 % shorter (3.79 vs 3.84 ms, two interleaved rounds on one box: tools/fwd_ab.py).  VTI_NO_FOLD512=1 / VTI_FOLD512=1.
            const char* f8 = getenv("VTI_FOLD512");
            const char* nf8 = getenv("VTI_NO_FOLD512");
            const bool fold8 = f8 ? f8[0] == '1' : (d.dtype == VTI_H2 && !(nf8 && nf8[0] == '1'));
            if (d.dtype != VTI_F16 && fold8 && ru.h_in >= 8) { op.cfg.TH = 8; op.cfg.threads = 512; }
            op.cfg.lds = convfold_lds_bytes(op.cfg.TH, op.cfg.TW);
            // persistent schedule (weights stay in LDS): fp16 only (the fp32 engine's 4 chunks of 64 KB do not fit), tensors < 2 GiB
            const char* npf = getenv("VTI_NO_PK_FOLD");
            if (pk_ok && op.cfg.nchunks <= 2 && !(npf && npf[0] == '1')) { op.cfg.pk = 1; op.cfg.pk_wgpc = 1; op.cfg.pk_depth = conv_pk_fold_depth(op.cfg.nchunks);
                                                                          op.cfg.lds = conv_pk_fold_lds_bytes(op.cfg.nchunks, op.cfg.pk_depth); }
        }
        else if (op.fused_l1 >= 0) choose_conv_cfg(d.dtype, r, true, d.max_batch, op.cfg, 0, 0, 1, 1);   // one 16-channel n-tile: stem_l1_kernel's weight indexing
        else if (op.fused >= 0) {   // whole Cout in one wave; the per-tile kernel (2 workgroups per CU) hides the long fused epilogue better
            const char* pf = getenv("VTI_PK_FUSED");
            // VTI_PK_FUSED: 1 = every fused op on the persistent schedule, 2 = those whose register tile leaves room for the fused stage
            // (NREP <= 4: the 80-channel class towers spill there)
            // h2: the 64-channel box towers run the fused stage on the persistent schedule by default (A/B on one box: 150 -> 137 us at P3,
            // 41 -> 30 us at P5; the 32-channel coefficient towers lose 1-2 us there and stay on the per-tile kernel)
            const bool pkf = pf ? (pf[0] == '1' || (pf[0] == '2' && r.c2 / 16 <= 4)) : (d.dtype == VTI_H2 && r.c2 / 16 == 4);
            choose_conv_cfg(d.dtype, r, false, d.max_batch, op.cfg, 0, 0, 1, r.c2 / 16, pk_ok && pkf);
            // 8-wave workgroups on 16 x 40 tiles where such tiles cover the map as well as the chosen ones (the 80-wide level: most
            // of the towers' time): half the weight staging per pixel (conv.hip, NT = 512)
            const char* n5 = getenv("VTI_NO_T512");
            if (!(n5 && n5[0] == '1') && !op.cfg.pk && op.cfg.TH && r.k == 3 && r.s == 1 && r.kind == 0) {
                const int TH = 16, TW = 40;
                const double eff_now = (double)r.h_out * r.w_out / ((double)((r.h_out + op.cfg.TH - 1) / op.cfg.TH) * ((r.w_out + op.cfg.TW - 1) / op.cfg.TW) * op.cfg.TH * op.cfg.TW);
                const double eff_512 = (double)r.h_out * r.w_out / ((double)((r.h_out + TH - 1) / TH) * ((r.w_out + TW - 1) / TW) * TH * TW);
                const size_t lds = conv_lds_bytes(3, 1, 0, TH, TW, 1, op.cfg.NREP);
                const size_t wgs = (size_t)((r.h_out + TH - 1) / TH) * ((r.w_out + TW - 1) / TW) * d.max_batch;
                if (eff_512 >= eff_now - 1e-9 && lds <= 160 * 1024 && wgs >= 512 && conv_cfg_fits(3, 1, 0, TH, TW, 1, op.cfg.NREP, 512)) {
                    op.cfg.TH = TH; op.cfg.TW = TW; op.cfg.threads = 512; op.cfg.lds = lds;
                }
            }
        }
        else choose_conv_cfg(d.dtype, r, op.kind == OP_CONV0, d.max_batch, op.cfg, 0, 0, 0, 0, pk_ok);
        if (op.cfg.TH == 0) return "no launch configuration for conv " + r.name;
        op.nat2 = (op.fused >= 0 && op.fused_l1 < 0 && (op.pred_mode || op.out2_f32)) ? 1 : 0;
        op.cfg.wpk_off = woff;
        op.cfg.bias_off = boff;
        if (op.fold >= 0) { woff += packed_fold_bytes(op.cfg); boff += (size_t)9 * r.c2; }      // bias: [3 x 3 border classes][Cout]
        else if (op.kind == OP_CONV0 && op.fused_l1 >= 0) { woff += packed_stem_toeplitz_bytes(d.dtype); boff += 16; }
        else { woff += packed_conv_bytes(r, op.kind == OP_CONV0, op.cfg); boff += (size_t)op.cfg.ntiles_n * 16; }
        if (op.pair >= 0) {         // second conv of the pair: the same packing, its own slot
            op.cfg.wpk_off2 = woff; op.cfg.bias_off2 = boff;
            woff += packed_conv_bytes(convs[op.pair], false, op.cfg);
            boff += (size_t)op.cfg.ntiles_n * 16;
        }
        if (op.tail >= 0) {         // the C2f's closing 1x1 inside bneck_pk: Wa (2 fragments) + Wb (2 fragments), 32 biases
            const ConvRow& rt = convs[op.tail];
            macs += rt.macs(); fused_params += rt.fused_params();
            op.cfg.wpk_off3 = woff; op.cfg.bias_off3 = boff;
            woff += (d.dtype == VTI_F16 ? 4 : 6) * 1024; boff += 32;      // h2: Wa is two 16-channel chunks
        }
        if (op.fused_l1 >= 0) {
            const ConvRow& r1 = convs[op.fused_l1];
            macs += r1.macs(); fused_params += r1.fused_params();
            op.cfg.wpk_off2 = woff; op.cfg.bias_off2 = boff;
            woff += packed_l1pairs_bytes(d.dtype);
            boff += 32;
        }
        if (op.fused >= 0) {
            const ConvRow& r2 = convs[op.fused];
            macs += r2.macs(); fused_params += r2.fused_params();
            op.cfg.ntiles2 = (r2.c2 + 15) / 16; op.cfg.gemm_n2 = r2.c2;
            if (op.fused_l1 >= 0) {     // stage 2 of the stem kernel sits on layer 1's register tile (2 n-tiles); offsets 2 are layer 1's
                op.cfg.wpk_off3 = woff; op.cfg.bias_off3 = boff;
                woff += packed_stage2_bytes(d.dtype, r2, 2);
            } else {
                op.cfg.wpk_off2 = woff; op.cfg.bias_off2 = boff;
                woff += packed_stage2_bytes(d.dtype, r2, op.cfg.NREP);
            }
            boff += (size_t)op.cfg.ntiles2 * 16;
        }
    }
    wpk_bytes = woff; bias_floats = boff;

    // Fold "nearest-2x Upsample -> Concat -> 1x1 conv" (neck layers 10-12, 13-15): when the consumer runs on conv1_pk its
    // loader reads the upsampled channels straight from the low-resolution tensor at (y >> 1, x >> 1); the UP2 op
    // and its write + re-read of a 4x larger tensor disappear.
    {
        const char* nu = getenv("VTI_NO_UPFUSE");
        const int KC = d.dtype == VTI_F16 ? 32 : 16;
        for (size_t i = 0; !(nu && nu[0] == '1') && i < ops.size(); ++i) {
            if (ops[i].kind != OP_UP2) continue;
            const View uo = ops[i].out;
            int consumer = -1, readers = 0;
            for (size_t j = 0; j < ops.size(); ++j) {
                if (j == i) continue;
                const Op& o = ops[j];
                auto overlaps = [&](const View& v) { return v.buf == uo.buf && v.coff < uo.coff + uo.C && uo.coff < v.coff + v.C; };
                if ((o.kind == OP_CONV || o.kind == OP_UP2 || o.kind == OP_POOL) && (overlaps(o.in) || (o.has_res && overlaps(o.res)))) {
                    ++readers; consumer = (int)j;
                }
            }
            if (readers != 1 || consumer < (int)i) continue;
            Op& cv = ops[consumer];
            const ConvRow& r = convs[cv.conv];
            if (cv.kind != OP_CONV || cv.cfg.pk != 2 || r.k != 1 || cv.in.coff != uo.coff || uo.C % KC || uo.C >= cv.in.C || cv.lane != ops[i].lane) continue;
            cv.up_src = ops[i].in; cv.up_C = uo.C;
            ops.erase(ops.begin() + i);
            --i;
        }
    }
    return "";
}

}  // namespace vti
