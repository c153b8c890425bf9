// VTIW1 container parsing and repacking into MFMA fragment order (host side).
//
// Stands in for the state-dict the reference obtains by unpickling its .pt
// (measurement.py:145); tensor names follow Ultralytics' `model.{i}...` prefixes so a
// converter can fill the container from a real checkpoint.
//
// Packed layout per conv (consumed by conv.hip):
//   wpk[chunk][ntile][tap][lane 0..63][VEC]   VEC = 8 fp16 / 4 fp32 (16 B per lane)
//   element = W[cout = perm(ntile, lane&15)][cin = chunk*KC + (lane>>4)*VEC + j][tap]
// i.e. one (chunk, ntile, tap) fragment is the 1 KiB a wave loads as the MFMA "A" operand
// (rows = output channels), zero padded past Cout / Cin.
// Row permutation: a wave owns NREP consecutive n-tiles (a "group" of 16*NREP channels).  Row r of tile n
// carries channel group*16*NREP + (r>>2)*4*NREP + n*4 + (r&3), so that accumulator lane-group g = r>>2 ends
// up with the 4*NREP CONSECUTIVE channels [g*4*NREP, (g+1)*4*NREP) of its pixel: the epilogue then stores
// (and reads residuals) in 16-byte pieces that tile whole 128-byte lines instead of scattered 8-byte ones.
static inline int perm_cout(int nt, int r, int nrep) {
    return (nt / nrep) * 16 * nrep + (r >> 2) * 4 * nrep + (nt % nrep) * 4 + (r & 3);
}
#include <algorithm>
#include <cmath>
#include <cstring>

#include "vti_internal.h"

namespace vti {

namespace {
struct Hdr { char magic[4]; uint32_t version; char scale[4]; uint32_t nc, nm, reg_max, n_convs; char pad[36]; };
struct Rec { char name[48]; uint32_t c1, c2, k, s, kind; char pad[12]; };
static_assert(sizeof(Hdr) == 64 && sizeof(Rec) == 80, "VTIW1 record sizes");
}  // namespace

// Writes n packed elements in the engine's storage type.  h2 (split-fp16 pairs, conv_dev.h): the values are scaled by SW, the power
// of two that puts max|v| in [2^13, 2^14) (so that the lo halves of all but vanishing weights are NORMAL fp16 numbers), stored as
// (hi | lo << 16) with hi = fp16(v SW), lo = fp16(v SW - hi); *alpha = 1 / (SW * 16) is what the kernel multiplies its accumulator
// by (16 = H2_SX, the activation scale of conv_dev.h).
static void emit_packed(int dtype, const std::vector<float>& v, uint8_t* dst, float* alpha) {
    if (dtype == VTI_F16) { for (size_t i = 0; i < v.size(); ++i) ((_Float16*)dst)[i] = (_Float16)v[i]; if (alpha) *alpha = 1.f; return; }
    if (dtype == VTI_F32) { memcpy(dst, v.data(), v.size() * 4); if (alpha) *alpha = 1.f; return; }
    float mx = 0.f;
    for (float x : v) mx = std::max(mx, std::fabs(x));
    int e = 0;
    if (mx > 0.f) (void)std::frexp(mx, &e);            // mx = m 2^e, m in [0.5, 1)
    const float sw = mx > 0.f ? std::ldexp(1.0f, 14 - e) : 1.0f;
    for (size_t i = 0; i < v.size(); ++i) {
        const float sv = v[i] * sw;
        const _Float16 hi = (_Float16)sv;
        const _Float16 lo = (_Float16)(sv - (float)hi);
        uint16_t hb, lb;
        memcpy(&hb, &hi, 2); memcpy(&lb, &lo, 2);
        ((uint32_t*)dst)[i] = (uint32_t)hb | ((uint32_t)lb << 16);
    }
    if (alpha) *alpha = 1.0f / (sw * 16.0f);
}

size_t packed_conv_bytes(const ConvRow& r, bool conv0, const ConvCfg& c) {
    const int taps = (conv0 || r.kind == 2) ? 1 : r.k * r.k;
    return (size_t)c.nchunks * c.ntiles_n * taps * 64 * 16;
}

void pack_conv(int dtype, const ConvRow& r, bool conv0, const ConvCfg& c, const float* w, const float* b,
               uint8_t* dst, float* bd, int cin_off, float* alpha) {
    const bool f16 = dtype == VTI_F16;
    const int KC = f16 ? 32 : 16, VEC = f16 ? 8 : 4;
    const bool deconv = r.kind == 2;
    const int taps = (conv0 || deconv) ? 1 : r.k * r.k;
    std::vector<float> pv((size_t)c.nchunks * c.ntiles_n * taps * 64 * VEC);
    for (int ck = 0; ck < c.nchunks; ++ck)
        for (int nt = 0; nt < c.ntiles_n; ++nt)
            for (int tap = 0; tap < taps; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < VEC; ++j) {
                        const int ng = perm_cout(nt, lane & 15, c.NREP);
                        const int kk = ck * KC + (lane >> 4) * VEC + j;
                        float v = 0.f;
                        if (ng < c.gemm_n) {
                            if (conv0) {
                                if (kk < 27) {   // k = (kh*3+kw)*3 + channel
                                    const int t = kk / 3, chn = kk % 3;
                                    v = w[((size_t)ng * 3 + chn) * 9 + t];
                                }
                            } else if (deconv) {
                                if (kk < r.c1) {   // gemm column = (dy*2+dx)*c2 + co ; torch IOHW
                                    const int q = ng / r.c2, co = ng % r.c2;
                                    v = w[((size_t)kk * r.c2 + co) * 4 + q];
                                }
                            } else if (kk >= cin_off && kk - cin_off < r.c1) {
                                v = w[((size_t)ng * r.c1 + (kk - cin_off)) * taps + tap];
                            }
                        }
                        const size_t e = ((((size_t)ck * c.ntiles_n + nt) * taps + tap) * 64 + lane) * VEC + j;
                        pv[e] = v;
                    }
    emit_packed(dtype, pv, dst, alpha);
    for (int n = 0; n < c.ntiles_n * 16; ++n) bd[n] = n < c.gemm_n ? b[deconv ? n % r.c2 : n] : 0.f;
}

// Fused 1x1 stage.  The producer's accumulator lane (g = lane>>4) holds, for its cout tile n, channels
// 16n + 4g + j (j<4).  fp16: one 32-deep MFMA step t consumes tiles 2t and 2t+1, so operand element jj of
// lane group g is mid-channel 32t + 16(jj>>2) + 4g + (jj&3); the weights are packed with that same map.
// fp32: one 16-deep step per tile with element i = channel 16t + 4g + i, i.e. the standard chunk packing.
size_t packed_stage2_bytes(int dtype, const ConvRow& r2, int nrep1) {
    const int kt = dtype == VTI_F16 ? (nrep1 + 1) / 2 : nrep1;
    return (size_t)kt * ((r2.c2 + 15) / 16) * 1024;
}

// The producer's accumulator lane (g = lane>>4) holds for its cout tile n the mid channels
// g*4*nrep1 + n*4 + j (the row permutation above).  fp16: one 32-deep MFMA step t consumes tiles 2t, 2t+1,
// so operand element jj of lane group g is mid channel g*4*nrep1 + (2t + (jj>>2))*4 + (jj&3);
// fp32: one 16-deep step per tile, element jj = g*4*nrep1 + t*4 + jj.  Output rows use the same
// permutation with nrep2 = ceil(c2/16) (one group).
// natural_rows: output channel of tile n2, fragment row r is 16*n2 + r (the transposed pred stage, where the
// weights are the MFMA column operand and a lane's column IS its channel).
void pack_conv_stage2(int dtype, const ConvRow& r2, int nrep1, const float* w, const float* b, uint8_t* dst, float* bd,
                      bool natural_rows, float* alpha) {
    const bool f16 = dtype == VTI_F16;
    const int kt = f16 ? (nrep1 + 1) / 2 : nrep1, VEC = f16 ? 8 : 4;
    const int nt2 = (r2.c2 + 15) / 16;
    std::vector<float> pv((size_t)kt * nt2 * 64 * VEC);
    for (int t = 0; t < kt; ++t)
        for (int n2 = 0; n2 < nt2; ++n2)
            for (int lane = 0; lane < 64; ++lane)
                for (int jj = 0; jj < VEC; ++jj) {
                    const int g = lane >> 4, co = natural_rows ? n2 * 16 + (lane & 15) : perm_cout(n2, lane & 15, nt2);
                    const int n1 = f16 ? 2 * t + (jj >> 2) : t;
                    const int cm = g * 4 * nrep1 + n1 * 4 + (f16 ? (jj & 3) : jj);
                    const float v = (n1 < nrep1 && co < r2.c2 && cm < r2.c1) ? w[(size_t)co * r2.c1 + cm] : 0.f;
                    const size_t e = (((size_t)t * nt2 + n2) * 64 + lane) * VEC + jj;
                    pv[e] = v;
                }
    emit_packed(dtype, pv, dst, alpha);
    for (int n = 0; n < nt2 * 16; ++n) bd[n] = n < r2.c2 ? b[n] : 0.f;
}

// ConvTranspose2d(C, M, 2, 2, bias) [rU, IOHW] folded into the 3x3 conv that follows it [rV: M -> O, pad 1]
// (proto.upsample -> proto.cv2).  With U[Y][X] = bU + WU[:, :, Y&1, X&1]^T P[Y>>1][X>>1] and zero padding of U,
//   V[2y+py][2x+px] = bV' + sum_{a,b in {0,1}} Weff[py][px][a][b] P[y-1+py+a][x-1+px+b]
//   Weff[py][px][a][b][o][c] = sum over the 3x3 taps (dy,dx) whose source row/column falls on window row a / column b of
//                              sum_m WV[o][m][dy][dx] WU[c][m][r][q],   (r, q) = parity of (py+dy, px+dx),
// i.e. four 2x2 convolutions on the LOW-resolution map (K = 4 C instead of 9 M at 4x the pixels: 2.25x fewer MACs, and the
// deconv's output tensor never exists).  P outside the map reads as zero; what remains of the padding is the deconv BIAS of
// taps that fall outside U, which depends on the output pixel's border class (first / interior / last row x column):
//   bias[cy][cx][o] = bV[o] + sum_{dy valid for cy, dx valid for cx} sum_m WV[o][m][dy][dx] bU[m].
// Exact in real arithmetic; in floating point the composed weights are rounded once (instead of rounding U), which stays
// inside the per-layer tolerance of the parity tests (tests/test_gpu_forward.py).
// Packed as a conv with gemm N = 4 phases x O (phase-major), 4 taps (a, b), rows permuted per phase group.
size_t packed_fold_bytes(const ConvCfg& c) { return (size_t)c.nchunks * c.ntiles_n * 4 * 1024; }

void pack_conv_fold(int dtype, const ConvRow& rU, const ConvRow& rV, const ConvCfg& c, const float* wU, const float* bU,
                    const float* wV, const float* bV, uint8_t* dst_w, float* dst_b, float* alpha) {
    const int C = rU.c1, M = rU.c2, O = rV.c2;
    // tap (dy in -1..1) of output parity py -> (window row a, deconv parity r): source row 2y+py+dy = 2 (y-1+py+a) + r
    auto split = [](int py, int dy, int& a, int& r) {
        const int t = py + dy;                     // -1 .. 2
        const int lo = t >= 0 ? t / 2 : -1;         // floor(t / 2)
        r = t - 2 * lo;
        a = lo + 1 - py;                            // window row 0 is low-res row y - 1 + py
    };
    std::vector<float> weff((size_t)4 * O * C * 4, 0.f);         // [(phase, o)][c][a][b]  (OIHW with k = 2)
    std::vector<double> acc((size_t)O * C);
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px)
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    std::fill(acc.begin(), acc.end(), 0.0);
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            int aa, r, bb, q;
                            split(py, dy, aa, r); split(px, dx, bb, q);
                            if (aa != a || bb != b) continue;
                            for (int o = 0; o < O; ++o)
                                for (int m = 0; m < M; ++m) {
                                    const double wv = wV[(((size_t)o * M + m) * 3 + (dy + 1)) * 3 + (dx + 1)];
                                    for (int ci = 0; ci < C; ++ci)
                                        acc[(size_t)o * C + ci] += wv * (double)wU[(((size_t)ci * M + m) * 2 + r) * 2 + q];
                                }
                        }
                    const int ph = py * 2 + px;
                    for (int o = 0; o < O; ++o)
                        for (int ci = 0; ci < C; ++ci)
                            weff[((((size_t)ph * O + o) * C + ci) * 2 + a) * 2 + b] = (float)acc[(size_t)o * C + ci];
                }
    ConvRow syn;
    syn.name = "fold"; syn.c1 = C; syn.c2 = 4 * O; syn.k = 2; syn.s = 1; syn.kind = 0;
    syn.h_in = syn.w_in = syn.h_out = syn.w_out = 0;
    ConvCfg sc = c;
    sc.gemm_n = 4 * O;
    std::vector<float> zero_b((size_t)4 * O, 0.f), bias_sink((size_t)c.ntiles_n * 16);
    pack_conv(dtype, syn, false, sc, weff.data(), zero_b.data(), dst_w, bias_sink.data(), 0, alpha);
    // bias table [cy][cx][O]: class 0 = first row/column (tap -1 outside), 1 = interior, 2 = last (tap +1 outside)
    for (int cy = 0; cy < 3; ++cy)
        for (int cx = 0; cx < 3; ++cx)
            for (int o = 0; o < O; ++o) {
                double v = bV[o];
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if ((cy == 0 && dy < 0) || (cy == 2 && dy > 0) || (cx == 0 && dx < 0) || (cx == 2 && dx > 0)) continue;
                        for (int m = 0; m < M; ++m) v += (double)wV[(((size_t)o * M + m) * 3 + (dy + 1)) * 3 + (dx + 1)] * (double)bU[m];
                    }
                dst_b[((size_t)cy * 3 + cx) * O + o] = (float)v;
            }
}

// The stem (3 -> 16, k3 s2) inside stem_l1_kernel as a Toeplitz GEMM (conv.hip, phase 2): for output pixel p (0..3) of a 4-pixel
// window and kernel row kh, fragment row co holds W'[k] = w[co][chn][kh][kw] at window element k = 6 p + 3 + 3 kw + m, where m is
// the MEMORY position of the channel inside the pixel (m = chn, or 2 - chn when the kernel flips the channel order: swap_rb), and
// zero elsewhere.  Layout [swap 0..1][p][kh][chunk][lane][VEC]; natural row order (one n-tile: 16 stem channels).
size_t packed_stem_toeplitz_bytes(int dtype) { return (size_t)2 * 4 * 3 * (dtype == VTI_F16 ? 1 : 2) * 1024; }

void pack_stem_toeplitz(int dtype, const ConvRow& r0, const float* w, const float* b, uint8_t* dst, float* bd, float* alpha) {
    // h2: the kernel's pixel operand is the byte's INTEGER value in plain fp16 (exact), so the fragments use the fp16 K layout and carry
    // w / 255 as two planes: hi = fp16(w' SW), lo = fp16(w' SW - hi); alpha = 1 / SW (the operand is not scaled by 16 here)
    const bool f16 = dtype == VTI_F16 || dtype == VTI_H2;
    const int KC = f16 ? 32 : 16, VEC = f16 ? 8 : 4, NCH = 32 / KC;
    std::vector<float> pv((size_t)2 * 4 * 3 * NCH * 64 * VEC);
    for (int sw = 0; sw < 2; ++sw)
        for (int pp = 0; pp < 4; ++pp)
            for (int kh = 0; kh < 3; ++kh)
                for (int c = 0; c < NCH; ++c)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < VEC; ++j) {
                            const int co = lane & 15, k = c * KC + (lane >> 4) * VEC + j;
                            const int t = k - 6 * pp - 3;
                            float v = 0.f;
                            if (t >= 0 && t < 9 && co < r0.c2) {
                                const int kw = t / 3, m = t % 3, chn = sw ? 2 - m : m;
                                v = w[(((size_t)co * 3 + chn) * 3 + kh) * 3 + kw];
                                if (dtype == VTI_H2) v = (float)((double)v / 255.0);
                            }
                            const size_t e = ((((((size_t)sw * 4 + pp) * 3 + kh) * NCH + c) * 64) + lane) * VEC + j;
                            pv[e] = v;
                        }
    if (dtype == VTI_H2) {
        float mx = 0.f;
        for (float x : pv) mx = std::max(mx, std::fabs(x));
        int e = 0;
        if (mx > 0.f) (void)std::frexp(mx, &e);
        const float sc = mx > 0.f ? std::ldexp(1.0f, 14 - e) : 1.0f;
        const size_t frag = (size_t)64 * VEC;                      // elements of one (sw, pp, kh) fragment
        for (size_t f = 0; f < pv.size() / frag; ++f)
            for (size_t i = 0; i < frag; ++i) {
                const float sv = pv[f * frag + i] * sc;
                const _Float16 hi = (_Float16)sv, lo = (_Float16)(sv - (float)hi);
                ((_Float16*)dst)[(2 * f) * frag + i] = hi;         // plane 0
                ((_Float16*)dst)[(2 * f + 1) * frag + i] = lo;     // plane 1
            }
        if (alpha) *alpha = 1.0f / sc;
    } else {
        emit_packed(dtype, pv, dst, alpha);
    }
    for (int n = 0; n < 16; ++n) bd[n] = n < r0.c2 ? b[n] : 0.f;
}

// Layer 1 (16 -> 32, k3 s2) inside stem_l1_kernel: K = 9 taps x 16 channels walked in MFMA steps that pair taps.
// fp16: step s, lane group g -> tap 2s + (g >> 1), channels 8 (g & 1) + j (j < 8); the 10th half-step is zero.
// fp32: step s = tap s, channels 4 g + j (j < 4).  Layout [step][ntile 0..1][lane][VEC]; rows permuted as for NREP = 2.
size_t packed_l1pairs_bytes(int dtype) { return (size_t)(dtype == VTI_F16 ? 5 : 9) * 2 * 1024; }

void pack_conv_l1pairs(int dtype, const ConvRow& r1, const float* w, const float* b, uint8_t* dst, float* bd, float* alpha) {
    const bool f16 = dtype == VTI_F16;
    const int NS = f16 ? 5 : 9, VEC = f16 ? 8 : 4;
    std::vector<float> pv((size_t)NS * 2 * 64 * VEC);
    for (int s = 0; s < NS; ++s)
        for (int nt = 0; nt < 2; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < VEC; ++j) {
                    const int g = lane >> 4, co = perm_cout(nt, lane & 15, 2);
                    const int tap = f16 ? 2 * s + (g >> 1) : s;
                    const int ch = f16 ? (g & 1) * 8 + j : g * 4 + j;
                    const float v = (tap < 9 && co < r1.c2) ? w[((size_t)co * r1.c1 + ch) * 9 + tap] : 0.f;
                    const size_t e = (((size_t)s * 2 + nt) * 64 + lane) * VEC + j;
                    pv[e] = v;
                }
    emit_packed(dtype, pv, dst, alpha);
    for (int n = 0; n < 32; ++n) bd[n] = n < r1.c2 ? b[n] : 0.f;
}

std::string pack_weights(const Plan& plan, const void* blob, size_t nbytes, std::vector<uint8_t>& wpk,
                         std::vector<float>& bias, std::vector<float>& alpha) {
    const uint8_t* p = (const uint8_t*)blob;
    if (!blob || nbytes < sizeof(Hdr)) return "weights: container too small";
    Hdr h;
    memcpy(&h, p, sizeof h);
    if (memcmp(h.magic, "VTIW", 4) != 0 || h.version != 1) return "weights: not a VTIW1 container";
    if (h.scale[0] != plan.desc.scale || (int)h.nc != plan.desc.nc || (int)h.nm != plan.desc.nm ||
        (int)h.reg_max != plan.desc.reg_max)
        return "weights: container scale/nc/nm/reg_max do not match the model description";
    if (h.n_convs != plan.convs.size()) return "weights: conv count does not match the plan";

    wpk.assign(plan.wpk_bytes, 0);
    bias.assign(plan.bias_floats, 0.f);
    alpha.assign(plan.convs.size(), 1.f);       // per conv: accumulator scale of its packed weights (1 unless dtype == VTI_H2)

    // conv index -> op (cfg); a conv fused into its producer's epilogue maps to that producer
    std::vector<const Op*> op_of(plan.convs.size(), nullptr), host_of(plan.convs.size(), nullptr), l1_host(plan.convs.size(), nullptr);
    for (const Op& op : plan.ops)
        if (op.kind == OP_CONV || op.kind == OP_CONV0) {
            op_of[op.conv] = &op;
            if (op.fused >= 0) host_of[op.fused] = &op;
            if (op.fused_l1 >= 0) l1_host[op.fused_l1] = &op;
        }

    std::vector<const Op*> tail_host(plan.convs.size(), nullptr);      // a C2f's closing 1x1 computed inside its bottleneck's kernel
    for (const Op& op : plan.ops)
        if (op.kind == OP_CONV && op.tail >= 0) tail_host[op.tail] = &op;
    std::vector<const Op*> pair_host(plan.convs.size(), nullptr);      // second 3x3 of a fused Bottleneck -> the op of the first
    for (const Op& op : plan.ops)
        if (op.kind == OP_CONV && op.pair >= 0) pair_host[op.pair] = &op;
    std::vector<const Op*> fold_host(plan.convs.size(), nullptr);      // deconv index -> the 3x3 op it is folded into
    for (const Op& op : plan.ops)
        if ((op.kind == OP_CONV) && op.fold >= 0) fold_host[op.fold] = &op;
    std::vector<float> fold_w, fold_b;                                  // the folded deconv's tensors, kept until its 3x3 arrives

    size_t off = sizeof(Hdr);
    for (size_t i = 0; i < plan.convs.size(); ++i) {
        const ConvRow& r = plan.convs[i];
        if (off + sizeof(Rec) > nbytes) return "weights: truncated container";
        Rec rec;
        memcpy(&rec, p + off, sizeof rec);
        off += sizeof rec;
        char nm[49];
        memcpy(nm, rec.name, 48); nm[48] = 0;
        if (r.name != nm) return std::string("weights: expected conv '") + r.name + "' but found '" + nm + "'";
        if ((int)rec.c1 != r.c1 || (int)rec.c2 != r.c2 || (int)rec.k != r.k || (int)rec.s != r.s || (int)rec.kind != r.kind)
            return "weights: shape mismatch for " + r.name;
        const size_t nw = (size_t)r.c1 * r.c2 * r.k * r.k;
        if (off + 4 * (nw + r.c2) > nbytes) return "weights: truncated container";
        std::vector<float> w(nw), b(r.c2);
        memcpy(w.data(), p + off, 4 * nw); off += 4 * nw;
        memcpy(b.data(), p + off, 4 * (size_t)r.c2); off += 4 * (size_t)r.c2;

        if (tail_host[i]) {
            // out = silu(Wa . [y0 | y1] + Wb . y2 + b): Wa = the first 2c input channels as an ordinary one-chunk 1x1 (rows permuted for
            // two n-tiles), Wb = the last c against the producer's accumulator layout (one n-tile), 2 + 2 fragments
            const ConvCfg& hc = tail_host[i]->cfg;
            const int cb = plan.convs[tail_host[i]->conv].c1, ca = r.c1 - cb;       // c (y2), 2c (y0 | y1)
            std::vector<float> wa((size_t)r.c2 * ca), wb((size_t)r.c2 * cb), zb(r.c2, 0.f), sink(64);
            for (int o = 0; o < r.c2; ++o) {
                for (int c2 = 0; c2 < ca; ++c2) wa[(size_t)o * ca + c2] = w[(size_t)o * r.c1 + c2];
                for (int c2 = 0; c2 < cb; ++c2) wb[(size_t)o * cb + c2] = w[(size_t)o * r.c1 + ca + c2];
            }
            ConvRow ra = r; ra.c1 = ca;
            ConvRow rb2 = r; rb2.c1 = cb;
            ConvCfg ac; ac.nchunks = 1; ac.ntiles_n = 2; ac.NREP = 2; ac.gemm_n = r.c2;
            uint8_t* dst = wpk.data() + hc.wpk_off3;
            if (plan.desc.dtype == VTI_H2) {
                // h2: Wa as two 16-channel chunks [chunk][n-tile] + Wb [n-tile], laid out as the fp32 packing does and then encoded TOGETHER,
                // so that the three partial products of the tail share one scale (alpha[i])
                ac.nchunks = 2;
                std::vector<float> tmp((size_t)6 * 256);            // 6 fragments of 64 lanes x 4 channels
                pack_conv(VTI_F32, ra, false, ac, wa.data(), zb.data(), (uint8_t*)tmp.data(), sink.data());
                pack_conv_stage2(VTI_F32, rb2, 1, wb.data(), b.data(), (uint8_t*)(tmp.data() + 4 * 256), bias.data() + hc.bias_off3, false);
                emit_packed(VTI_H2, tmp, dst, &alpha[i]);
                continue;
            }
            pack_conv(plan.desc.dtype, ra, false, ac, wa.data(), zb.data(), dst, sink.data());
            pack_conv_stage2(plan.desc.dtype, rb2, 1, wb.data(), b.data(), dst + 2 * 1024, bias.data() + hc.bias_off3, false);
            continue;
        }
        if (pair_host[i]) {
            const ConvCfg& hc = pair_host[i]->cfg;
            pack_conv(plan.desc.dtype, r, false, hc, w.data(), b.data(), wpk.data() + hc.wpk_off2, bias.data() + hc.bias_off2, 0, &alpha[i]);
            continue;
        }
        if (fold_host[i]) { fold_w = w; fold_b = b; continue; }          // packed together with the 3x3 that follows
        if (op_of[i] && op_of[i]->fold >= 0) {
            const Op& op = *op_of[i];
            pack_conv_fold(plan.desc.dtype, plan.convs[op.fold], r, op.cfg, fold_w.data(), fold_b.data(), w.data(), b.data(),
                           wpk.data() + op.cfg.wpk_off, bias.data() + op.cfg.bias_off, &alpha[i]);
            continue;
        }
        if (l1_host[i]) {      // layer 1 computed inside the stem's kernel
            const ConvCfg& hc = l1_host[i]->cfg;
            pack_conv_l1pairs(plan.desc.dtype, r, w.data(), b.data(), wpk.data() + hc.wpk_off2, bias.data() + hc.bias_off2, &alpha[i]);
            continue;
        }
        if (host_of[i]) {
            const ConvCfg& hc = host_of[i]->cfg;
            if (host_of[i]->fused_l1 >= 0)      // third conv of the stem kernel: K order of layer 1's two n-tiles
                pack_conv_stage2(plan.desc.dtype, r, 2, w.data(), b.data(), wpk.data() + hc.wpk_off3, bias.data() + hc.bias_off3, false, &alpha[i]);
            else
                pack_conv_stage2(plan.desc.dtype, r, hc.NREP, w.data(), b.data(), wpk.data() + hc.wpk_off2, bias.data() + hc.bias_off2,
                                 host_of[i]->nat2 != 0, &alpha[i]);
            continue;
        }
        const Op& op = *op_of[i];
        if (op.tail >= 0 && plan.desc.dtype == VTI_F16) {          // first 3x3 of a bottleneck with the fused tail: its input y1 is the UPPER half of the [y0 | y1] slot
            pack_conv(plan.desc.dtype, r, false, op.cfg, w.data(), b.data(), wpk.data() + op.cfg.wpk_off, bias.data() + op.cfg.bias_off, r.c1);
            continue;
        }
        if (op.kind == OP_CONV0 && op.fused_l1 >= 0) {       // the stem inside stem_l1_kernel: banded (Toeplitz) fragments
            pack_stem_toeplitz(plan.desc.dtype, r, w.data(), b.data(), wpk.data() + op.cfg.wpk_off, bias.data() + op.cfg.bias_off, &alpha[i]);
            continue;
        }
        pack_conv(plan.desc.dtype, r, op.kind == OP_CONV0, op.cfg, w.data(), b.data(), wpk.data() + op.cfg.wpk_off,
                  bias.data() + op.cfg.bias_off, 0, &alpha[i]);
    }
    if (off != nbytes) return "weights: trailing bytes in container";
    return "";
}

}  // namespace vti
