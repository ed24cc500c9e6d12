// Whole-network entry points of the C ABI: a non-Python host builds an NCSN++ score network from the reference's
// state_dict tensors and evaluates it with one call.
//
//   storm_ncsnpp_create(cfg, weight_ptrs[n], ...)   replaces NCSNpp.__init__ + load_state_dict (ncsnpp.py:38-273):
//       enumerates the modules in the reference's registration order, packs every tensor into the engine's arena
//       (conv weights -> [tap][CoutP][CinP] in the operand dtype, NIN matrices transposed, all Dense_0 of the residual
//       blocks concatenated into one matrix, Conv_1 + Conv_2 biases pre-added);
//   storm_ncsnpp_forward(handle, ...)               replaces NCSNpp.forward (ncsnpp.py:281-450): plans the fused op
//       program for (B, F, T) once (liveness-based workspace reuse) and runs it through storm_program_run.
//
// Host code only (plus two trivial fill kernels).  The planner mirrors, op for op, what tests/py_planner.py restates in
// Python (tests compare the two op lists bit for bit).
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstring>
#include "common.h"

namespace storm {
namespace graph {

constexpr long long ALIGN = 256;
static inline long long up(long long x, long long m) { return (x + m - 1) / m * m; }

enum Kind { GFP, LINEAR, CONV3, RES, COMBINE, ATTN, GN };
struct Module { Kind kind; int i = 0, o = 0, c = 0, n = 0; bool resample = false; };

struct Cfg {
    int nf = 128; std::vector<int> ch_mult{1, 2, 2, 2}; int num_res_blocks = 1; std::vector<int> attn_res{0};
    int image_size = 256, input_channels = 4; bool discriminative = false;
    bool conditional() const { return !discriminative; }
    int total() const { return discriminative ? 2 : input_channels; }
};

static bool in_list(const std::vector<int>& v, int x) { for (int e : v) if (e == x) return true; return false; }

// all_modules in registration order (ncsnpp.py:153-273)
static std::vector<Module> module_list(const Cfg& c) {
    const int nf = c.nf, nres = (int)c.ch_mult.size(), total = c.total();
    std::vector<int> all_res;
    for (int i = 0; i < nres; ++i) all_res.push_back(c.image_size / (1 << i));
    std::vector<Module> m;
    auto add = [&](Kind k, int i, int o, int cc, int n, bool rs) { Module x; x.kind = k; x.i = i; x.o = o; x.c = cc; x.n = n; x.resample = rs; m.push_back(x); };
    add(GFP, 0, 0, 0, nf, false);
    if (c.conditional()) { add(LINEAR, 2 * nf, 4 * nf, 0, 0, false); add(LINEAR, 4 * nf, 4 * nf, 0, 0, false); }
    add(CONV3, total, nf, 0, 0, false);
    std::vector<int> hs_c{nf};
    int in_ch = nf;
    for (int lvl = 0; lvl < nres; ++lvl) {
        for (int b = 0; b < c.num_res_blocks; ++b) {
            const int out_ch = nf * c.ch_mult[lvl];
            add(RES, in_ch, out_ch, 0, 0, false);
            in_ch = out_ch;
            if (in_list(c.attn_res, all_res[lvl])) add(ATTN, 0, 0, in_ch, 0, false);
            hs_c.push_back(in_ch);
        }
        if (lvl != nres - 1) {
            add(RES, in_ch, in_ch, 0, 0, true);
            add(COMBINE, total, in_ch, 0, 0, false);
            hs_c.push_back(in_ch);
        }
    }
    in_ch = hs_c.back();
    add(RES, in_ch, in_ch, 0, 0, false); add(ATTN, 0, 0, in_ch, 0, false); add(RES, in_ch, in_ch, 0, 0, false);
    for (int lvl = nres - 1; lvl >= 0; --lvl) {
        for (int b = 0; b < c.num_res_blocks + 1; ++b) {
            const int out_ch = nf * c.ch_mult[lvl];
            add(RES, in_ch + hs_c.back(), out_ch, 0, 0, false);
            hs_c.pop_back();
            in_ch = out_ch;
        }
        if (in_list(c.attn_res, all_res[lvl])) add(ATTN, 0, 0, in_ch, 0, false);
        add(GN, 0, 0, in_ch, 0, false);
        add(CONV3, in_ch, total, 0, 0, false);
        if (lvl != 0) add(RES, in_ch, in_ch, 0, 0, true);
    }
    return m;
}

struct Tensor { std::string name; std::vector<long long> shape; long long numel() const { long long n = 1; for (auto s : shape) n *= s; return n; } };

// the reference state_dict, in its order
static std::vector<Tensor> state_dict(const Cfg& c) {
    std::vector<Tensor> t;
    const int total = c.total();
    auto add = [&](const std::string& n, std::vector<long long> s) { t.push_back(Tensor{n, std::move(s)}); };
    add("output_layer.weight", {2, total, 1, 1}); add("output_layer.bias", {2});
    const auto mods = module_list(c);
    for (size_t idx = 0; idx < mods.size(); ++idx) {
        const Module& p = mods[idx];
        const std::string k = "all_modules." + std::to_string(idx) + ".";
        switch (p.kind) {
            case GFP: add(k + "W", {p.n}); break;
            case LINEAR: add(k + "weight", {p.o, p.i}); add(k + "bias", {p.o}); break;
            case CONV3: add(k + "weight", {p.o, p.i, 3, 3}); add(k + "bias", {p.o}); break;
            case GN: add(k + "weight", {p.c}); add(k + "bias", {p.c}); break;
            case COMBINE: add(k + "Conv_0.weight", {p.o, p.i, 1, 1}); add(k + "Conv_0.bias", {p.o}); break;
            case ATTN:
                add(k + "GroupNorm_0.weight", {p.c}); add(k + "GroupNorm_0.bias", {p.c});
                for (int j = 0; j < 4; ++j) { add(k + "NIN_" + std::to_string(j) + ".W", {p.c, p.c}); add(k + "NIN_" + std::to_string(j) + ".b", {p.c}); }
                break;
            case RES:
                add(k + "GroupNorm_0.weight", {p.i}); add(k + "GroupNorm_0.bias", {p.i});
                add(k + "Conv_0.weight", {p.o, p.i, 3, 3}); add(k + "Conv_0.bias", {p.o});
                add(k + "Dense_0.weight", {p.o, 4 * c.nf}); add(k + "Dense_0.bias", {p.o});
                add(k + "GroupNorm_1.weight", {p.o}); add(k + "GroupNorm_1.bias", {p.o});
                add(k + "Conv_1.weight", {p.o, p.o, 3, 3}); add(k + "Conv_1.bias", {p.o});
                if (p.i != p.o || p.resample) { add(k + "Conv_2.weight", {p.o, p.i, 1, 1}); add(k + "Conv_2.bias", {p.o}); }
                break;
        }
    }
    return t;
}

// ---- packed parameter arena -----------------------------------------------------------------------------------
struct Entry { std::string key; enum K { CONV, NIN, F32, F32SUM, DENSE_W, DENSE_B } kind; std::vector<std::string> src; long long shape[3]; long long off, bytes; };
struct Layout {
    int dtype, esize, per16;
    std::vector<Entry> entries; std::map<std::string, int> index;
    long long size = 0, dense_rows = 0;
    std::map<int, long long> dense_off;
    void add(const std::string& key, Entry::K kind, std::vector<std::string> src, long long s0, long long s1, long long s2, int es) {
        Entry e; e.key = key; e.kind = kind; e.src = std::move(src); e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2;
        e.off = size; e.bytes = es * s0 * s1 * s2;
        index[key] = (int)entries.size(); entries.push_back(e);
        size = up(size + e.bytes, ALIGN);
    }
    void conv(const std::string& n, int Cout, int Cin, int taps) { add(n, Entry::CONV, {n}, taps, up(Cout, 32), up(Cin, 2 * per16), esize); }
    void nin(const std::string& n, int C) { add(n, Entry::NIN, {n}, 1, up(C, 32), C, esize); }
    void f32(const std::string& n, long long a, long long b = 1) { add(n, Entry::F32, {n}, a, b, 1, 4); }
    const Entry& at(const std::string& k) const { return entries[index.at(k)]; }
    bool has(const std::string& k) const { return index.count(k) != 0; }
    long long off(const std::string& k) const { return at(k).off; }
};

static Layout make_layout(const Cfg& c, int dtype) {
    Layout L; L.dtype = dtype; L.esize = dtype == STORM_F32 ? 4 : 2; L.per16 = 16 / L.esize;
    const int total = c.total();
    L.f32("output_layer.weight", 2, total); L.f32("output_layer.bias", 2);
    std::vector<std::string> dense;
    const auto mods = module_list(c);
    for (size_t idx = 0; idx < mods.size(); ++idx) {
        const Module& p = mods[idx];
        const std::string k = "all_modules." + std::to_string(idx) + ".";
        switch (p.kind) {
            case GFP: L.f32(k + "W", p.n); break;
            case LINEAR: L.f32(k + "weight", p.o, p.i); L.f32(k + "bias", p.o); break;
            case CONV3: L.conv(k + "weight", p.o, p.i, 9); L.f32(k + "bias", p.o); break;
            case GN: L.f32(k + "weight", p.c); L.f32(k + "bias", p.c); break;
            case COMBINE: L.conv(k + "Conv_0.weight", p.o, p.i, 1); L.f32(k + "Conv_0.bias", p.o); break;
            case ATTN:
                L.f32(k + "GroupNorm_0.weight", p.c); L.f32(k + "GroupNorm_0.bias", p.c);
                for (int j = 0; j < 4; ++j) { L.nin(k + "NIN_" + std::to_string(j) + ".W", p.c); L.f32(k + "NIN_" + std::to_string(j) + ".b", p.c); }
                break;
            case RES:
                L.f32(k + "GroupNorm_0.weight", p.i); L.f32(k + "GroupNorm_0.bias", p.i);
                L.conv(k + "Conv_0.weight", p.o, p.i, 9); L.f32(k + "Conv_0.bias", p.o);
                L.f32(k + "GroupNorm_1.weight", p.o); L.f32(k + "GroupNorm_1.bias", p.o);
                L.conv(k + "Conv_1.weight", p.o, p.o, 9);
                if (p.i != p.o || p.resample) {
                    L.conv(k + "Conv_2.weight", p.o, p.i, 1);
                    L.add(k + "bias12", Entry::F32SUM, {k + "Conv_1.bias", k + "Conv_2.bias"}, p.o, 1, 1, 4);
                } else L.f32(k + "Conv_1.bias", p.o);
                if (c.conditional()) { L.dense_off[(int)idx] = L.dense_rows; L.dense_rows += p.o; dense.push_back(k + "Dense_0"); }
                break;
        }
    }
    if (c.conditional()) {
        std::vector<std::string> w, b;
        for (auto& s : dense) { w.push_back(s + ".weight"); b.push_back(s + ".bias"); }
        L.add("dense.weight", Entry::DENSE_W, w, L.dense_rows, 4 * c.nf, 1, 4);
        L.add("dense.bias", Entry::DENSE_B, b, L.dense_rows, 1, 1, 4);
    }
    return L;
}

// ---- op program --------------------------------------------------------------------------------------------------
enum { BUF_WS = 0, BUF_PARAMS, BUF_IN0, BUF_IN1, BUF_IN2, BUF_T, BUF_OUT, N_BUFS };

// first-fit offset allocator with a coalescing free list (activations are reused aggressively)
struct Arena {
    std::vector<std::pair<long long, long long>> free_; long long top = 0; std::map<long long, long long> live;
    long long alloc(long long nbytes) {
        const long long n = up(nbytes > 1 ? nbytes : 1, ALIGN);
        for (size_t k = 0; k < free_.size(); ++k)
            if (free_[k].second >= n) {
                const long long off = free_[k].first;
                if (free_[k].second == n) free_.erase(free_.begin() + (long)k);
                else { free_[k].first += n; free_[k].second -= n; }
                live[off] = n; return off;
            }
        long long off;
        if (!free_.empty() && free_.back().first + free_.back().second == top) { off = free_.back().first; free_.pop_back(); top = off + n; }
        else { off = top; top += n; }
        live[off] = n; return off;
    }
    void release(long long off) {
        const long long n = live.at(off); live.erase(off);
        free_.push_back({off, n});
        std::sort(free_.begin(), free_.end());
        std::vector<std::pair<long long, long long>> m;
        for (auto& f : free_) { if (!m.empty() && m.back().first + m.back().second == f.first) m.back().second += f.second; else m.push_back(f); }
        free_.swap(m);
    }
};

struct Act { long long off = 0; int H = 0, W = 0, C = 0; long long part = -1; int tiles = 0; bool valid = false; };
struct Seg {                       // one K-segment of a conv op
    bool a_is_act = true; Act a; int a_buf = -1; long long a_off = 0; int a_C = 0;
    bool has_b = false; Act b;
    bool w_par = true; std::string w_key; long long w_off = 0;
    int CinP = 0, rows = 0, taps = 1; long long w_bstride = 0, w_tapstride = -1;
    long long gn = -1; bool gn_silu = true;
};

struct Program {
    const Cfg& cfg; const Layout& lay; int B, F, T; bool fuse_stats = true, fuse_apply = true, fused_attention = true;
    int dtype, esize; std::vector<storm_op> ops; Arena arena; long long flops = 0, ws_bytes = 0;
    long long stats_off = 0, stats_bytes = 0, stats_cursor = 0, dense_out = 0;
    // HIP-graph replay of the evaluation (storm_ncsnpp_set_graph): the maximal runs of ops that touch only the workspace and the weight
    // arena, instantiated once per workspace address.  state: 0 = first call (runs eagerly: lazy kernel attributes / code-object loads
    // must not happen inside a capture), 1 = captured, -1 = capture failed on this runtime (eager from then on).
    struct GraphSeg { int first, last; hipGraphExec_t exec; };
    struct GraphSet {
        int state = 0; std::vector<GraphSeg> segs; unsigned long long used = 0;
        unsigned long long epoch = switch_epoch();              // the launchers' switch table as it stood when this set was created
        ~GraphSet() { for (auto& sg : segs) if (sg.exec) (void)hipGraphExecDestroy(sg.exec); }
    };
    std::map<void*, std::shared_ptr<GraphSet>> graphs;
    Program(const Cfg& c, const Layout& l, int B_, int F_, int T_) : cfg(c), lay(l), B(B_), F(F_), T(T_), dtype(l.dtype), esize(l.esize) {}

    storm_op& op(int code) {
        storm_op o; memset(&o, 0, sizeof(o)); o.code = code;
        for (int j = 0; j < STORM_OP_NPTR; ++j) o.p[j].buf = -1;
        ops.push_back(o); return ops.back();
    }
    static void ref(storm_op& o, int j, int buf, long long off) { o.p[j].buf = buf; o.p[j].off = off; }
    void ws(storm_op& o, int j, long long off) { ref(o, j, BUF_WS, off); }
    void par(storm_op& o, int j, const std::string& key) { ref(o, j, BUF_PARAMS, lay.off(key)); }
    Act new_act(int H, int W, int C, int es = 0) { Act a; a.off = arena.alloc((long long)B * H * W * C * (es ? es : esize)); a.H = H; a.W = W; a.C = C; a.valid = true; return a; }
    void free_act(const Act& a) { arena.release(a.off); if (a.part >= 0) arena.release(a.part); }
    long long new_stats(int G) { const long long off = stats_cursor; stats_cursor += up((long long)B * G * 2 * 8, ALIGN); return off; }

    // GroupNorm (+SiLU) (+FIR resample of h and raw x): returns (out, raw)
    std::pair<Act, Act> gn(const Act& xa, const Act* xb, const std::string& wkey, const std::string& bkey, bool silu, int resample) {
        const int Cc = xa.C + (xb ? xb->C : 0), G = std::min(Cc / 4, 32);
        const long long st = new_stats(G);
        if (xa.part >= 0 && (!xb || xb->part >= 0)) {
            storm_op& o = op(STORM_OP_GN_FINALIZE);
            ws(o, 0, xa.part); if (xb) ws(o, 1, xb->part); ws(o, 2, st);
            o.i[0] = xa.C; o.i[1] = xa.tiles; o.i[2] = xb ? xb->C : 0; o.i[3] = xb ? xb->tiles : 0; o.i[4] = B; o.i[5] = G;
        } else {
            storm_op& o = op(STORM_OP_GN_STATS);
            ws(o, 0, xa.off); if (xb) ws(o, 1, xb->off); ws(o, 2, st);
            o.i[0] = xa.C; o.i[1] = xb ? xb->C : 0; o.i[2] = B; o.i[3] = (long long)xa.H * xa.W; o.i[4] = G;
        }
        const int OH = resample == 1 ? 2 * xa.H : (resample == 2 ? xa.H / 2 : xa.H), OW = resample == 1 ? 2 * xa.W : (resample == 2 ? xa.W / 2 : xa.W);
        Act out = new_act(OH, OW, Cc), raw;
        if (resample) raw = new_act(OH, OW, Cc);
        storm_op& o = op(STORM_OP_GN_APPLY);
        ws(o, 0, xa.off); if (xb) ws(o, 1, xb->off); ws(o, 2, st); par(o, 3, wkey); par(o, 4, bkey); ws(o, 5, out.off);
        if (raw.valid) ws(o, 6, raw.off);
        o.i[0] = xa.C; o.i[1] = xb ? xb->C : 0; o.i[2] = B; o.i[3] = xa.H; o.i[4] = xa.W; o.i[5] = G; o.i[6] = silu; o.i[7] = resample;
        o.f[0] = 1e-6f;
        return {out, raw};
    }
    // GroupNorm as a per-(batch, channel) affine table for a conv that fuses the apply (+SiLU) into its operand load
    long long gn_affine(const Act& xa, const Act* xb, const std::string& wkey, const std::string& bkey) {
        const int Cc = xa.C + (xb ? xb->C : 0), G = std::min(Cc / 4, 32);
        const long long st = new_stats(G), ss = arena.alloc((long long)B * Cc * 2 * 4);
        storm_op& o = op(STORM_OP_GN_FINALIZE);
        ws(o, 0, xa.part); if (xb) ws(o, 1, xb->part); ws(o, 2, st); par(o, 3, wkey); par(o, 4, bkey); ws(o, 5, ss);
        o.i[0] = xa.C; o.i[1] = xa.tiles; o.i[2] = xb ? xb->C : 0; o.i[3] = xb ? xb->tiles : 0; o.i[4] = B; o.i[5] = G; o.i[6] = (long long)xa.H * xa.W;
        o.f[0] = 1e-6f;
        return ss;
    }
    Seg wseg(const Act& a, const std::string& key, int taps, const Act* b = nullptr, long long gnss = -1) {
        const Entry& e = lay.at(key);
        Seg s; s.a = a; s.has_b = b != nullptr; if (b) s.b = *b; s.w_par = true; s.w_key = key; s.CinP = (int)e.shape[2]; s.rows = (int)e.shape[1]; s.taps = taps; s.gn = gnss;
        return s;
    }
    struct ConvOpt { int outC = 0; const char* bias_key = nullptr; std::string bias; bool has_tb = false; long long tb_off = 0; int tb_stride = 0;
                     const Act* skip = nullptr; float scale = 1.0f; bool out_f32 = false; long long out_bstride = -1, src0_bstride = -1; bool want_part = false; };
    Act conv(const std::vector<Seg>& segs, int Cout, int H, int W, const ConvOpt& k) {
        const int outC = k.outC ? k.outC : (int)up(Cout, 8);
        Act out = new_act(H, W, outC, k.out_f32 ? 4 : 0);
        storm_op& o = op(STORM_OP_CONV);
        o.i[0] = (long long)segs.size(); o.i[1] = B; o.i[2] = H; o.i[3] = W; o.i[4] = outC; o.i[5] = Cout; o.i[7] = k.out_f32;
        for (size_t g = 0; g < segs.size(); ++g) {
            const Seg& s = segs[g];
            int Ca, Cb = 0;
            if (s.a_is_act) { ws(o, 3 * (int)g, s.a.off); Ca = s.a.C; } else { ref(o, 3 * (int)g, s.a_buf, s.a_off); Ca = s.a_C; }
            if (s.has_b) { ws(o, 3 * (int)g + 1, s.b.off); Cb = s.b.C; }
            if (s.w_par) par(o, 3 * (int)g + 2, s.w_key); else ws(o, 3 * (int)g + 2, s.w_off);
            const int q = 8 + 7 * (int)g;
            o.i[q] = Ca; o.i[q + 1] = Cb; o.i[q + 2] = s.CinP; o.i[q + 3] = s.rows; o.i[q + 4] = s.taps; o.i[q + 5] = s.w_bstride;
            o.i[q + 6] = s.w_tapstride >= 0 ? s.w_tapstride : (long long)s.CinP * s.rows;
            flops += 2LL * B * H * W * Cout * (Ca + Cb) * s.taps;
        }
        o.i[22] = k.src0_bstride; o.i[23] = k.out_bstride;
        if (segs[0].gn >= 0) { ws(o, 11, segs[0].gn); o.f[1] = segs[0].gn_silu ? 1.0f : 0.0f; }
        ws(o, 6, out.off);
        if (!k.bias.empty()) par(o, 7, k.bias);
        if (k.has_tb) { ref(o, 8, BUF_WS, k.tb_off); o.i[6] = k.tb_stride; }
        if (k.skip) ws(o, 9, k.skip->off);
        o.f[0] = k.scale;
        if (k.want_part) {
            bool any9 = false; for (auto& s : segs) any9 = any9 || s.taps == 9;
            out.tiles = any9 ? ((W + 31) / 32) * ((H + 7) / 8) : (int)(((long long)H * W + 255) / 256);
            out.part = arena.alloc((long long)B * out.tiles * outC * 2 * 4);
            ws(ops.back(), 10, out.part);
        }
        {   // few-tile 3x3 layers: scratch for storm_conv's split of K over workgroups - live for this op only (the stream orders
            // every later writer of the region behind it)
            storm_conv_args q;
            memset(&q, 0, sizeof(q));
            q.nseg = (int)segs.size(); q.B = B; q.H = H; q.W = W; q.outC = outC; q.Cout = Cout; q.out_f32 = k.out_f32; q.dtype = dtype;
            for (size_t g = 0; g < segs.size(); ++g) {
                const Seg& s = segs[g];
                storm_conv_seg& sg = q.seg[g];
                sg.src_a = &q; sg.w = &q;                       // (shape query: the pointers only have to be non-NULL)
                sg.Ca = s.a_is_act ? s.a.C : s.a_C; sg.Cb = s.has_b ? s.b.C : 0; if (s.has_b) sg.src_b = &q;
                sg.CinP = s.CinP; sg.w_rows = s.rows; sg.ntaps = s.taps; sg.w_bstride = s.w_bstride;
                sg.w_tapstride = s.w_tapstride >= 0 ? s.w_tapstride : (long long)s.CinP * s.rows;
                sg.bstride_a = (long long)H * W * sg.Ca; sg.bstride_b = (long long)H * W * sg.Cb;
            }
            const long long nb = storm_conv_splitk_bytes(&q);
            if (nb > 0) {                                       // f[2] = the slabs the scratch holds (the interpreter sizes it from that)
                const long long off = arena.alloc(nb); ws(ops.back(), 12, off); arena.release(off);
                ops.back().f[2] = (float)(nb / ((long long)B * H * W * outC * 4));
            }
        }
        return out;
    }

    // ResnetBlockBigGANpp.forward (layerspp.py:242-274) as 6-7 fused ops
    Act resblock(int idx, const Module& p, const Act& xa, const Act* xb, int resample) {
        const std::string k = "all_modules." + std::to_string(idx) + ".";
        const int o = p.o;
        const float inv = (float)(1.0 / std::sqrt(2.0));
        const bool fuse = fuse_apply && xa.part >= 0 && (!xb || xb->part >= 0);
        ConvOpt c0; c0.bias = k + "Conv_0.bias";
        if (cfg.conditional()) { c0.has_tb = true; c0.tb_off = dense_out + 4 * lay.dense_off.at(idx); c0.tb_stride = (int)lay.dense_rows; }
        Act u, xr, a2;
        if (fuse && !resample) {
            const long long ss0 = gn_affine(xa, xb, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias");
            c0.want_part = true;
            u = conv({wseg(xa, k + "Conv_0.weight", 9, xb, ss0)}, o, xa.H, xa.W, c0);
            arena.release(ss0);
        } else {
            auto pr = gn(xa, xb, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias", true, resample);
            xr = pr.second;
            c0.want_part = fuse_stats;
            u = conv({wseg(pr.first, k + "Conv_0.weight", 9)}, o, pr.first.H, pr.first.W, c0);
            free_act(pr.first);
        }
        Seg s1; long long ss1 = -1;
        if (fuse) { ss1 = gn_affine(u, nullptr, k + "GroupNorm_1.weight", k + "GroupNorm_1.bias"); s1 = wseg(u, k + "Conv_1.weight", 9, nullptr, ss1); }
        else { auto pr = gn(u, nullptr, k + "GroupNorm_1.weight", k + "GroupNorm_1.bias", true, 0); a2 = pr.first; free_act(u); s1 = wseg(a2, k + "Conv_1.weight", 9); }
        Act out;
        ConvOpt c1; c1.scale = inv; c1.want_part = fuse_stats;
        if (lay.has(k + "Conv_2.weight")) {
            Seg s2 = xr.valid ? wseg(xr, k + "Conv_2.weight", 1) : wseg(xa, k + "Conv_2.weight", 1, xb);
            c1.bias = k + "bias12";
            out = conv({s1, s2}, o, u.H, u.W, c1);
        } else {
            c1.bias = k + "Conv_1.bias"; c1.skip = &xa;
            out = conv({s1}, o, u.H, u.W, c1);
        }
        if (fuse) { arena.release(ss1); free_act(u); } else free_act(a2);
        if (xr.valid) free_act(xr);
        return out;
    }

    // AttnBlockpp.forward (layerspp.py:75-91)
    Act attnblock(int idx, const Module& p, const Act& x) {
        const std::string k = "all_modules." + std::to_string(idx) + ".";
        const int Cc = p.c, Lp = x.H * x.W, Lp8 = (int)up(Lp, 8);
        auto pr = gn(x, nullptr, k + "GroupNorm_0.weight", k + "GroupNorm_0.bias", false, 0);
        Act h = pr.first, hl = h; hl.H = 1; hl.W = Lp; hl.part = -1;
        ConvOpt cq; cq.bias = k + "NIN_0.b";
        Act q = conv({wseg(hl, k + "NIN_0.W", 1)}, Cc, 1, Lp, cq);
        ConvOpt ck; ck.bias = k + "NIN_1.b";
        Act kk = conv({wseg(hl, k + "NIN_1.W", 1)}, Cc, 1, Lp, ck);
        Seg sv; sv.a_is_act = false; sv.a_buf = BUF_PARAMS; sv.a_off = lay.off(k + "NIN_2.W"); sv.a_C = Cc; sv.w_par = false; sv.w_off = h.off;
        sv.CinP = Cc; sv.rows = Lp; sv.taps = 1; sv.w_bstride = (long long)Lp * Cc;
        ConvOpt cv; cv.outC = Lp8; cv.src0_bstride = 0;
        Act vT = conv({sv}, Lp, 1, Cc, cv);
        free_act(h);
        Act o;
        const float scale = (float)std::pow((double)Cc, -0.5);
        if (fused_attention && storm_attention_supported(Cc, dtype)) {
            o = new_act(1, Lp, Cc);
            storm_op& a = op(STORM_OP_ATTENTION);
            ws(a, 0, q.off); ws(a, 1, kk.off); ws(a, 2, vT.off); par(a, 3, k + "NIN_2.b"); ws(a, 4, o.off);
            a.i[0] = B; a.i[1] = Lp; a.i[2] = Cc; a.i[3] = Lp8; a.f[0] = scale;
            const long long nb = storm_attention_scratch_bytes(B, Lp, Cc, dtype);       // key-range split of small calls: scratch live for this op only
            if (nb > 0) { const long long off = arena.alloc(nb); ws(ops.back(), 5, off); ops.back().i[4] = nb; arena.release(off); }
            flops += 4LL * B * Lp * Lp * Cc;
            free_act(q); free_act(kk); free_act(vT);
        } else {
            Seg ss; ss.a = q; ss.w_par = false; ss.w_off = kk.off; ss.CinP = Cc; ss.rows = Lp; ss.taps = 1; ss.w_bstride = (long long)Lp * Cc;
            ConvOpt cs; cs.outC = Lp8; cs.scale = scale; cs.out_f32 = true;
            Act S = conv({ss}, Lp, 1, Lp, cs);
            free_act(q); free_act(kk);
            Act P = new_act(1, Lp, Lp8);
            storm_op& sm = op(STORM_OP_SOFTMAX);
            ws(sm, 0, S.off); ws(sm, 1, P.off); sm.i[0] = (long long)B * Lp; sm.i[1] = Lp; sm.i[2] = Lp8;
            free_act(S);
            Seg sp; sp.a = P; sp.w_par = false; sp.w_off = vT.off; sp.CinP = Lp8; sp.rows = Cc; sp.taps = 1; sp.w_bstride = (long long)Cc * Lp8;
            ConvOpt co; co.bias = k + "NIN_2.b";
            o = conv({sp}, Cc, 1, Lp, co);
            free_act(P); free_act(vT);
        }
        Act xl = x; xl.H = 1; xl.W = Lp; xl.part = -1;
        ConvOpt c3; c3.bias = k + "NIN_3.b"; c3.skip = &xl; c3.scale = (float)(1.0 / std::sqrt(2.0)); c3.want_part = fuse_stats;
        Act out = conv({wseg(o, k + "NIN_3.W", 1)}, Cc, 1, Lp, c3);
        free_act(o);
        out.H = x.H; out.W = x.W;
        return out;
    }

    int build() {
        const auto mods = module_list(cfg);
        const int nres = (int)cfg.ch_mult.size(), total = cfg.total();
        if (F % (1 << (nres - 1)) || T % (1 << (nres - 1))) { set_error("storm_ncsnpp: spectrogram %dx%d must be divisible by %d", F, T, 1 << (nres - 1)); return STORM_ERR_INVALID; }
        int n_gn = 0;
        for (auto& m : mods) n_gn += m.kind == RES ? 2 : (m.kind == ATTN || m.kind == GN ? 1 : 0);
        stats_bytes = n_gn * up((long long)B * 32 * 2 * 8, ALIGN);
        stats_off = arena.alloc(stats_bytes); stats_cursor = stats_off;
        { storm_op& o = op(STORM_OP_MEMSET); ws(o, 0, stats_off); o.i[0] = stats_bytes; }
        const int n_in = total / 2;
        // the input pyramid (input_skip): a function of the network input alone, so every level is built here, ahead of the U-Net - the packing and
        // up to three FIR x2 down steps per launch (csrc/pyramid.hip; `ncsnpp`: one launch, `ncsnpplarge` with its six steps: two)
        std::vector<Act> ips;
        for (int lvl = 0; lvl < nres; ++lvl) ips.push_back(new_act(F >> lvl, T >> lvl, 8));
        for (int l0 = 0; l0 == 0 || l0 < nres - 1; l0 += 3) {
            const int nl = std::min(4, nres - l0);
            storm_op& o = op(STORM_OP_INPUT_PYRAMID);
            if (l0 == 0) for (int j = 0; j < n_in; ++j) ref(o, j, BUF_IN0 + j, 0);
            for (int k = 0; k < nl; ++k) ws(o, 3 + k, ips[(size_t)(l0 + k)].off);
            o.i[0] = l0 == 0 ? n_in : 0; o.i[1] = B; o.i[2] = F >> l0; o.i[3] = T >> l0; o.i[4] = nl;
        }
        Act x0 = ips[0];
        int midx = 1;
        if (cfg.conditional()) {
            const long long temb = arena.alloc((long long)B * 4 * cfg.nf * 4);
            storm_op& o = op(STORM_OP_TEMB);
            ref(o, 0, BUF_T, 0); par(o, 1, "all_modules.0.W"); par(o, 2, "all_modules.1.weight"); par(o, 3, "all_modules.1.bias");
            par(o, 4, "all_modules.2.weight"); par(o, 5, "all_modules.2.bias"); ws(o, 6, temb); o.i[0] = B; o.i[1] = cfg.nf;
            dense_out = arena.alloc((long long)B * lay.dense_rows * 4);
            storm_op& d = op(STORM_OP_DENSE);
            ws(d, 0, temb); par(d, 1, "dense.weight"); par(d, 2, "dense.bias"); ws(d, 3, dense_out); d.i[0] = B; d.i[1] = lay.dense_rows; d.i[2] = 4 * cfg.nf;
            midx = 3;
        }
        Act ip = x0;
        std::vector<Act> hs;
        { const std::string k = "all_modules." + std::to_string(midx) + "."; ConvOpt c; c.bias = k + "bias"; c.want_part = fuse_stats;
          hs.push_back(conv({wseg(x0, k + "weight", 9)}, cfg.nf, F, T, c)); ++midx; }
        for (int lvl = 0; lvl < nres; ++lvl) {
            for (int r = 0; r < cfg.num_res_blocks; ++r) {
                Act h = resblock(midx, mods[midx], hs.back(), nullptr, 0); ++midx;
                if (in_list(cfg.attn_res, h.H)) { Act h2 = attnblock(midx, mods[midx], h); ++midx; free_act(h); h = h2; }
                hs.push_back(h);
            }
            if (lvl != nres - 1) {
                Act h = resblock(midx, mods[midx], hs.back(), nullptr, 2); ++midx;
                free_act(ip); ip = ips[(size_t)lvl + 1];
                const std::string kk = "all_modules." + std::to_string(midx) + ".";
                ConvOpt c; c.bias = kk + "Conv_0.bias"; c.skip = &h; c.want_part = fuse_stats;
                Act hc = conv({wseg(ip, kk + "Conv_0.weight", 1)}, h.C, h.H, h.W, c); ++midx;
                free_act(h);
                hs.push_back(hc);
            }
        }
        free_act(ip);
        Act h = hs.back();
        Act h1 = resblock(midx, mods[midx], h, nullptr, 0); ++midx;
        Act h2 = attnblock(midx, mods[midx], h1); ++midx;
        free_act(h1);
        h = resblock(midx, mods[midx], h2, nullptr, 0); ++midx;
        free_act(h2);
        std::vector<Act> phs;
        for (int lvl = nres - 1; lvl >= 0; --lvl) {
            for (int r = 0; r < cfg.num_res_blocks + 1; ++r) {
                Act skip = hs.back(); hs.pop_back();
                Act hn = resblock(midx, mods[midx], h, &skip, 0); ++midx;
                free_act(h); free_act(skip);
                h = hn;
            }
            if (in_list(cfg.attn_res, h.H)) { Act hn = attnblock(midx, mods[midx], h); ++midx; free_act(h); h = hn; }
            const std::string kg = "all_modules." + std::to_string(midx) + ".", kc = "all_modules." + std::to_string(midx + 1) + ".";
            Act ph;
            ConvOpt c; c.outC = 8; c.bias = kc + "bias";
            if (fuse_apply && h.part >= 0) {
                const long long ssp = gn_affine(h, nullptr, kg + "weight", kg + "bias");
                ph = conv({wseg(h, kc + "weight", 9, nullptr, ssp)}, total, h.H, h.W, c);
                arena.release(ssp);
            } else {
                auto pr = gn(h, nullptr, kg + "weight", kg + "bias", true, 0);
                ph = conv({wseg(pr.first, kc + "weight", 9)}, total, h.H, h.W, c);
                free_act(pr.first);
            }
            midx += 2;
            phs.push_back(ph);                                   // (coarsest first; the up chain and the head run in ONE launch at the end)
            if (lvl != 0) { Act hn = resblock(midx, mods[midx], h, nullptr, 1); ++midx; free_act(h); h = hn; }
        }
        if (!hs.empty() || midx != (int)mods.size()) { set_error("storm_ncsnpp: planner walked %d of %zu modules", midx, mods.size()); return STORM_ERR_INVALID; }
        free_act(h);
        // output pyramid (output_skip) + head: p = ph_0 + up(ph_1 + up(ph_2 + ...)), then output_layer / t - one launch (csrc/pyramid.hip)
        storm_op& o = op(STORM_OP_OUTPUT_PYRAMID);
        for (int k = 0; k < nres; ++k) ws(o, k, phs[(size_t)(nres - 1 - k)].off);
        if (cfg.conditional()) ref(o, 8, BUF_T, 0);
        par(o, 9, "output_layer.weight"); par(o, 10, "output_layer.bias"); ref(o, 11, BUF_OUT, 0);
        o.i[0] = total; o.i[1] = B; o.i[2] = F; o.i[3] = T; o.i[4] = 0; o.i[5] = nres;
        for (const Act& a : phs) free_act(a);
        ws_bytes = arena.top;
        return STORM_OK;
    }
};

__global__ void add_f32_kernel(const float* a, const float* b, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

}  // namespace graph
}  // namespace storm

using namespace storm;
using namespace storm::graph;

struct storm_ncsnpp {
    Cfg cfg; int dtype; Layout layout; void* arena = nullptr; bool owns_arena = false;
    bool fuse_stats = true, fuse_apply = true, fused_attention = true;
    // planned programs by (B, F, T): a ragged stream produces one per bucket and tail batch size, so the cache is bounded
    // (least recently used out) and guarded - a handle may be shared by host threads driving different streams
    // (the fourth key: STORM_BATCH_INVARIANT - the planner sizes the split-K / attention scratch by the rules that switch changes)
    std::map<std::tuple<int, int, int, int>, std::shared_ptr<Program>> programs;
    std::map<std::tuple<int, int, int, int>, unsigned long long> last_use;
    unsigned long long tick = 0;
    // programs whose op list has been handed out through storm_ncsnpp_program (profilers hold the raw pointer): pinned for the life of
    // the handle, whatever the cache evicts or storm_ncsnpp_set_fusion drops
    std::vector<std::shared_ptr<Program>> exported;
    std::mutex mu;
    static constexpr size_t MAX_PROGRAMS = 64;
    int graph_mode = -1;             // storm_ncsnpp_set_graph: 0 eager, 1 replay, -1 = the library's rule (graph_wanted)
    hipStream_t cap_stream = nullptr;   // captures are recorded on a private stream (the caller's may be the legacy default stream, which cannot capture)
    unsigned long long gtick = 0;
    std::atomic<long long> graph_launches{0};   // hipGraphLaunch calls so far (storm_ncsnpp_graph_launches: did the replay really run?)
    static constexpr size_t MAX_GRAPH_WS = 4;   // workspaces (addresses) with instantiated graphs per program
    // grouped evaluation (storm_ncsnpp_forward_group): per list of (B, T) the problems' programs, their workspace offsets and the grouped
    // launches; the tables of those launches hold absolute pointers, so their device copy is tied to a workspace address and to the caller's
    // input / output tensors - they are rebuilt and re-uploaded whenever one of those changes
    struct GroupPlan {
        std::vector<std::shared_ptr<Program>> progs;
        std::vector<long long> ws_off;
        long long blob_bytes = 0, ws_bytes = 0;
        std::vector<GroupOp> gops;
        char* host_blob = nullptr;              // PINNED host image of the tables: the per-call upload is a true asynchronous copy
        std::vector<const void*> built_for;     // the workspace / arena addresses the tables were built with (grouped ops reference nothing else)
        unsigned long long epoch = 0, used = 0;
        // the image is read by an ASYNCHRONOUS copy: before it is rewritten (a caller pointer changed) the last copy that was enqueued from it
        // must have executed - the host may be a whole evaluation ahead of the device
        hipEvent_t copied{}; bool have_event = false, copy_pending = false;
        ~GroupPlan() { if (have_event) { (void)hipEventSynchronize(copied); (void)hipEventDestroy(copied); } if (host_blob) (void)hipHostFree(host_blob); }
    };
    std::map<std::vector<int>, std::shared_ptr<GroupPlan>> groups;
    static constexpr size_t MAX_GROUPS = 16;
    std::atomic<long long> group_launches{0};   // grouped kernel launches so far (storm_ncsnpp_group_launches: did the grouping really happen?)
};

static int to_cfg(const storm_ncsnpp_config* c, Cfg& out) {
    STORM_CHECK(c != nullptr, "storm_ncsnpp: null config");
    STORM_CHECK(c->nf > 0 && c->nf % 8 == 0, "storm_ncsnpp: nf=%d must be a positive multiple of 8", c->nf);
    STORM_CHECK(c->n_levels >= 1 && c->n_levels <= 8 && c->n_attn >= 0 && c->n_attn <= 4 && c->num_res_blocks >= 1, "storm_ncsnpp: bad config");
    out.nf = c->nf; out.ch_mult.assign(c->ch_mult, c->ch_mult + c->n_levels); out.num_res_blocks = c->num_res_blocks;
    out.attn_res.assign(c->attn_resolutions, c->attn_resolutions + c->n_attn); out.image_size = c->image_size;
    out.input_channels = c->discriminative ? 2 : c->input_channels; out.discriminative = c->discriminative != 0;
    STORM_CHECK(out.total() >= 2 && out.total() <= 6 && out.total() % 2 == 0, "storm_ncsnpp: input_channels=%d", c->input_channels);
    return STORM_OK;
}

extern "C" int storm_ncsnpp_num_tensors(const storm_ncsnpp_config* c) {
    Cfg cfg; if (to_cfg(c, cfg) != STORM_OK) return -1;
    return (int)state_dict(cfg).size();
}

extern "C" int storm_ncsnpp_tensor_info(const storm_ncsnpp_config* c, int i, char* name, int name_len, int* ndim, long long* shape4) {
    Cfg cfg; if (int rc = to_cfg(c, cfg)) return rc;
    const auto sd = state_dict(cfg);
    STORM_CHECK(i >= 0 && i < (int)sd.size() && name && name_len > 0 && ndim && shape4, "storm_ncsnpp_tensor_info: bad arguments");
    snprintf(name, (size_t)name_len, "%s", sd[i].name.c_str());
    *ndim = (int)sd[i].shape.size();
    for (int d = 0; d < 4; ++d) shape4[d] = d < *ndim ? sd[i].shape[d] : 1;
    return STORM_OK;
}

extern "C" long long storm_ncsnpp_arena_bytes(const storm_ncsnpp_config* c, int dtype) {
    Cfg cfg; if (to_cfg(c, cfg) != STORM_OK) return -1;
    return make_layout(cfg, dtype).size;
}

// weights: device fp32 tensors of the reference state_dict, in ITS order (storm_ncsnpp_tensor_info); arena: device buffer of
// storm_ncsnpp_arena_bytes() bytes owned by the caller, or NULL (the handle allocates and owns one).
extern "C" int storm_ncsnpp_create(const storm_ncsnpp_config* c, const void* const* weights, int n_weights, int dtype, void* arena,
                                   storm_stream_t s, storm_ncsnpp** out) {
    STORM_CHECK(out != nullptr && weights != nullptr, "storm_ncsnpp_create: null argument");
    STORM_CHECK(dtype == STORM_F32 || dtype == STORM_BF16 || dtype == STORM_F16, "storm_ncsnpp_create: dtype %d", dtype);
    Cfg cfg; if (int rc = to_cfg(c, cfg)) return rc;
    const auto sd = state_dict(cfg);
    STORM_CHECK(n_weights == (int)sd.size(), "storm_ncsnpp_create: %d weight tensors given, the configuration has %zu", n_weights, sd.size());
    std::map<std::string, int> by_name;
    for (size_t i = 0; i < sd.size(); ++i) { by_name[sd[i].name] = (int)i; STORM_CHECK(weights[i] != nullptr, "storm_ncsnpp_create: weight %zu (%s) is NULL", i, sd[i].name.c_str()); }
    storm_ncsnpp* h = new storm_ncsnpp();
    h->cfg = cfg; h->dtype = dtype; h->layout = make_layout(cfg, dtype);
    hipStream_t st = (hipStream_t)s;
    if (arena) h->arena = arena;
    else {
        const long long need = h->layout.size;
        if (hipMalloc(&h->arena, (size_t)need) != hipSuccess) { delete h; set_error("storm_ncsnpp_create: hipMalloc(%lld) failed", need); return STORM_ERR_HIP; }
        h->owns_arena = true;
    }
    char* base = static_cast<char*>(h->arena);
    int rc = hipMemsetAsync(base, 0, (size_t)h->layout.size, st) == hipSuccess ? STORM_OK : STORM_ERR_HIP;
    auto W = [&](const std::string& n) { return static_cast<const float*>(weights[by_name.at(n)]); };
    auto T_ = [&](const std::string& n) -> const Tensor& { return sd[by_name.at(n)]; };
    for (const Entry& e : h->layout.entries) {
        if (rc != STORM_OK) break;
        char* dst = base + e.off;
        switch (e.kind) {
            case Entry::CONV: { const Tensor& t = T_(e.src[0]);
                rc = storm_pack_conv_weight(W(e.src[0]), dst, (int)t.shape[0], (int)t.shape[1], (int)(t.shape[2] * t.shape[3]), (int)e.shape[1], (int)e.shape[2], dtype, s); break; }
            case Entry::NIN: { const Tensor& t = T_(e.src[0]);                // [Cin][Cout]
                rc = storm_pack_matrix(W(e.src[0]), dst, (int)t.shape[1], (int)t.shape[0], 1, (int)e.shape[1], (int)e.shape[2], dtype, s); break; }
            case Entry::F32:
                if (hipMemcpyAsync(dst, W(e.src[0]), (size_t)e.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = STORM_ERR_HIP;
                break;
            case Entry::F32SUM: { const int n = (int)e.shape[0];
                hipLaunchKernelGGL(add_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, st, W(e.src[0]), W(e.src[1]), reinterpret_cast<float*>(dst), n); break; }
            case Entry::DENSE_W: case Entry::DENSE_B: {
                long long off = 0;
                for (const std::string& sname : e.src) {
                    const long long nb = T_(sname).numel() * 4;
                    if (hipMemcpyAsync(dst + off, W(sname), (size_t)nb, hipMemcpyDeviceToDevice, st) != hipSuccess) rc = STORM_ERR_HIP;
                    off += nb;
                }
                if (off != e.bytes) { set_error("storm_ncsnpp_create: dense table size mismatch"); rc = STORM_ERR_INVALID; }
                break; }
        }
    }
    if (rc == STORM_OK && hipGetLastError() != hipSuccess) { set_error("storm_ncsnpp_create: launch failed"); rc = STORM_ERR_HIP; }
    if (rc != STORM_OK) { if (h->owns_arena) (void)hipFree(h->arena); delete h; return rc; }
    *out = h;
    return STORM_OK;
}

extern "C" void storm_ncsnpp_destroy(storm_ncsnpp* h) {
    if (!h) return;
    h->programs.clear(); h->exported.clear();               // (graph executables go with their programs)
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->owns_arena && h->arena) (void)hipFree(h->arena);
    delete h;
}

extern "C" int storm_ncsnpp_set_fusion(storm_ncsnpp* h, int fuse_stats, int fuse_apply, int fused_attention) {
    STORM_CHECK(h != nullptr, "storm_ncsnpp_set_fusion: null handle");
    h->fuse_stats = fuse_stats != 0; h->fuse_apply = fuse_apply != 0 && fuse_stats != 0; h->fused_attention = fused_attention != 0;
    std::lock_guard<std::mutex> lk(h->mu);
    h->programs.clear(); h->last_use.clear();                  // (and with them their recorded graphs)
    return STORM_OK;
}

// HIP-graph replay of the evaluations of this handle: 0 = eager launches, 1 = replay (the first call per (shape, workspace address) runs
// eagerly, the second records, later ones are one hipGraphLaunch per recorded run + the few ops that read the caller's tensors),
// -1 = the library's rule (graph_wanted: small batches).  A replayed evaluation launches exactly the kernels the eager one does.
extern "C" int storm_ncsnpp_set_graph(storm_ncsnpp* h, int mode) {
    STORM_CHECK(h != nullptr && mode >= -1 && mode <= 1, "storm_ncsnpp_set_graph: bad arguments");
    std::lock_guard<std::mutex> lk(h->mu);
    h->graph_mode = mode;
    return STORM_OK;
}

extern "C" long long storm_ncsnpp_graph_launches(storm_ncsnpp* h) { return h ? h->graph_launches.load(std::memory_order_relaxed) : -1; }

// The returned program stays alive for as long as the caller holds the shared_ptr (an eviction by another thread only drops
// the cache's reference).
static int get_program(storm_ncsnpp* h, int B, int F, int T, std::shared_ptr<Program>* out) {
    STORM_CHECK(h != nullptr && B > 0 && F > 0 && T > 0, "storm_ncsnpp: bad shape B=%d F=%d T=%d", B, F, T);
    std::lock_guard<std::mutex> lk(h->mu);
    auto key = std::make_tuple(B, F, T, storm::switches().batch_invariant != 0 ? 1 : 0);
    auto it = h->programs.find(key);
    if (it == h->programs.end()) {
        std::shared_ptr<Program> p(new Program(h->cfg, h->layout, B, F, T));
        p->fuse_stats = h->fuse_stats; p->fuse_apply = h->fuse_apply && h->fuse_stats; p->fused_attention = h->fused_attention;
        const int rc = p->build();
        if (rc != STORM_OK) return rc;
        if (h->programs.size() >= storm_ncsnpp::MAX_PROGRAMS) {
            auto old = h->last_use.begin();
            for (auto u = h->last_use.begin(); u != h->last_use.end(); ++u) if (u->second < old->second) old = u;
            h->programs.erase(old->first); h->last_use.erase(old);
        }
        it = h->programs.emplace(key, p).first;
    }
    h->last_use[key] = ++h->tick;
    *out = it->second;
    return STORM_OK;
}

extern "C" long long storm_ncsnpp_workspace_bytes(storm_ncsnpp* h, int B, int F, int T) {
    std::shared_ptr<Program> p;
    if (get_program(h, B, F, T, &p) != STORM_OK) return -1;
    return p->ws_bytes;
}

// the planned op list (for profilers: storm_program_run_timed / storm_program_kernel_name); owned by the handle and valid until the
// handle is destroyed: a program handed out here is pinned (neither the LRU eviction nor storm_ncsnpp_set_fusion frees it)
extern "C" int storm_ncsnpp_program(storm_ncsnpp* h, int B, int F, int T, const storm_op** ops, int* n_ops, long long* flops) {
    std::shared_ptr<Program> p;
    if (int rc = get_program(h, B, F, T, &p)) return rc;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        bool have = false;
        for (const auto& e : h->exported) have = have || e.get() == p.get();
        if (!have) h->exported.push_back(p);
    }
    if (ops) *ops = p->ops.data();
    if (n_ops) *n_ops = (int)p->ops.size();
    if (flops) *flops = p->flops;
    return STORM_OK;
}

// Unpin a program handed out by storm_ncsnpp_program (the op list may be freed from now on - by an eviction, a fusion change or the
// handle's destruction, whichever comes first): a profiler sweeping many shapes calls this when it is done with one.
extern "C" int storm_ncsnpp_release_program(storm_ncsnpp* h, const storm_op* ops) {
    STORM_CHECK(h != nullptr && ops != nullptr, "storm_ncsnpp_release_program: null argument");
    std::lock_guard<std::mutex> lk(h->mu);
    for (auto it = h->exported.begin(); it != h->exported.end(); ++it)
        if ((*it)->ops.data() == ops) { h->exported.erase(it); return STORM_OK; }
    STORM_CHECK(false, "storm_ncsnpp_release_program: not an op list of this handle");
}

extern "C" const void* storm_ncsnpp_arena(storm_ncsnpp* h) { return h ? h->arena : nullptr; }

// ---- one evaluation: eager launches, or HIP-graph replay ---------------------------------------------------------------------
// ops [a, b) of the program; `negate` is a run-time argument of THE CALL: the cached op list is shared by every caller of the
// handle, so the output head (the last op) runs from a local copy instead of being patched in place
static int run_range(const Program& p, int a, int b, void* const* bufs, int dtype, int negate, storm_stream_t s) {
    const int n = (int)p.ops.size();
    const int be = b == n ? n - 1 : b;
    if (be > a) if (int rc = storm_program_run(p.ops.data() + a, be - a, bufs, N_BUFS, dtype, s)) return rc;
    if (b == n && a < n) {
        storm_op head = p.ops[n - 1];
        head.i[4] = negate ? 1 : 0;
        return storm_program_run(&head, 1, bufs, N_BUFS, dtype, s);
    }
    return STORM_OK;
}

// The library's rule when nobody chose (storm_ncsnpp_set_graph(h, -1)): eager launches.  MEASURED (round 5, profiles/r05a_*): at ONE utterance
// per call - the reference's own operating point (enhancement.py:66-72), where an evaluation is 117 launches of 5 - 160 us - the kernels
// of an evaluation are busy 3325 us of its 3331 us wall: the C launch loop (0.5 ms of host time per evaluation) already runs ahead of
// the GPU and the queue never drains, so replay moves nothing (batch 1 / 2 / 4 / 16: 3.332 vs 3.328, 4.52 vs 4.54, 6.70 vs 6.70,
// 21.16 vs 21.11 ms per evaluation; ragged stream 5.53 vs 5.47 utt/s).  What a one-utterance call lacks is work per launch, not
// launches per second (conv_splitk_slices' small-call rule).  Replay stays available for hosts whose launch thread is slower.
static int graph_wanted(const Program&) { return 0; }

// maximal runs (>= 4 ops) of ops that reference nothing but the workspace and the weight arena: their kernel arguments depend on
// (program, workspace address) only, so one instantiated graph serves every later call with that workspace
static std::vector<std::pair<int, int>> graph_ranges(const Program& p) {
    std::vector<std::pair<int, int>> r;
    const int n = (int)p.ops.size();
    int a = -1;
    for (int k = 0; k <= n; ++k) {
        bool internal = k < n - 1;                              // (the output head carries `negate`: always eager)
        if (internal)
            for (int j = 0; j < STORM_OP_NPTR; ++j) { const int b = p.ops[k].p[j].buf; internal = internal && (b < 0 || b == BUF_WS || b == BUF_PARAMS); }
        if (internal) { if (a < 0) a = k; }
        else { if (a >= 0 && k - a >= 4) r.push_back({a, k}); a = -1; }
    }
    return r;
}

static int forward_replay(storm_ncsnpp* h, Program& p, void* const* bufs, int negate, storm_stream_t s) {
    std::shared_ptr<Program::GraphSet> g;
    bool capture = false;
    {
        std::lock_guard<std::mutex> lk(h->mu);
        auto it = p.graphs.find(bufs[BUF_WS]);
        if (it != p.graphs.end() && it->second->epoch != switch_epoch()) {      // a switch changed since: the recorded kernel selection is stale
            p.graphs.erase(it);
            it = p.graphs.end();
        }
        if (it == p.graphs.end()) {
            if (p.graphs.size() >= storm_ncsnpp::MAX_GRAPH_WS) {
                auto old = p.graphs.begin();
                for (auto u = p.graphs.begin(); u != p.graphs.end(); ++u) if (u->second->used < old->second->used) old = u;
                p.graphs.erase(old);
            }
            it = p.graphs.emplace(bufs[BUF_WS], std::make_shared<Program::GraphSet>()).first;
        }
        g = it->second;
        g->used = ++h->gtick;
        if (g->state == 0) g->state = 2;                        // this call runs eagerly (warm-up), the next one records
        else if (g->state == 2) {
            capture = true;
            if (h->cap_stream == nullptr && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) { h->cap_stream = nullptr; g->state = -1; capture = false; }
        }
        if (capture) {
            // recorded under the handle's lock (once per program and workspace): nothing executes here, the launches below do
            g->state = 1;
            for (const auto& r : graph_ranges(p)) {
                hipGraph_t gr = nullptr; hipGraphExec_t ex = nullptr;
                bool ok = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
                const int rc = ok ? storm_program_run(p.ops.data() + r.first, r.second - r.first, bufs, N_BUFS, h->dtype, (storm_stream_t)h->cap_stream) : STORM_ERR_HIP;
                if (ok) ok = hipStreamEndCapture(h->cap_stream, &gr) == hipSuccess && gr != nullptr;
                ok = ok && rc == STORM_OK && hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0) == hipSuccess;
                if (gr) (void)hipGraphDestroy(gr);
                if (!ok) { (void)hipGetLastError(); g->segs.clear(); g->state = -1; break; }
                g->segs.push_back({r.first, r.second, ex});
            }
        }
    }
    const int n = (int)p.ops.size();
    if (g->state != 1) return run_range(p, 0, n, bufs, h->dtype, negate, s);
    int k = 0;
    for (const auto& sg : g->segs) {
        if (int rc = run_range(p, k, sg.first, bufs, h->dtype, negate, s)) return rc;
        STORM_HIP(hipGraphLaunch(sg.exec, (hipStream_t)s));
        h->graph_launches.fetch_add(1, std::memory_order_relaxed);
        k = sg.last;
    }
    return run_range(p, k, n, bufs, h->dtype, negate, s);
}

// parts: n_parts device pointers to complex64 [B][F][T] tensors (x, y[, y_denoised]) - torch.cat([x, y], 1) of the reference
// never materialises; t: fp32 [B] (NULL for a discriminative net); out: complex64 [B][F][T]; negate: score = -dnn(...)
// (model.py:131-132) folded into the output head.
extern "C" int storm_ncsnpp_forward(storm_ncsnpp* h, const void* const* parts, int n_parts, const float* t, void* out, void* ws,
                                    long long ws_bytes, int B, int F, int T, int negate, storm_stream_t s) {
    std::shared_ptr<Program> p;
    if (int rc = get_program(h, B, F, T, &p)) return rc;
    STORM_CHECK(parts && out && ws, "storm_ncsnpp_forward: null pointer");
    STORM_CHECK(n_parts == h->cfg.total() / 2, "storm_ncsnpp_forward: %d complex input channels given, the network takes %d", n_parts, h->cfg.total() / 2);
    STORM_CHECK(ws_bytes >= p->ws_bytes, "storm_ncsnpp_forward: workspace %lld < %lld bytes", ws_bytes, p->ws_bytes);
    STORM_CHECK(!h->cfg.conditional() || t != nullptr, "storm_ncsnpp_forward: a score network needs t");
    void* bufs[N_BUFS] = {nullptr};
    bufs[BUF_WS] = ws; bufs[BUF_PARAMS] = h->arena;
    for (int j = 0; j < n_parts; ++j) { STORM_CHECK(parts[j] != nullptr, "storm_ncsnpp_forward: input %d is NULL", j); bufs[BUF_IN0 + j] = const_cast<void*>(parts[j]); }
    bufs[BUF_T] = const_cast<float*>(t); bufs[BUF_OUT] = out;
    int mode = switches().graph >= 0 ? switches().graph : h->graph_mode;
    if (mode < 0) mode = graph_wanted(*p);
    if (mode == 0) return run_range(*p, 0, (int)p->ops.size(), bufs, h->dtype, negate, s);
    return forward_replay(h, *p, bufs, negate, s);
}

// ---- grouped evaluation: P micro-batches of different (B, T) in ONE call -------------------------------------------------------------------
// BASELINE.json configs[4]: a stream of 2 - 10 s utterances micro-batched by padded frame count is 2 - 3 rows per launch - the deep levels of
// such a call are a handful of pixel tiles.  All micro-batches run the same op sequence, so op k of every problem is launched together where
// a grouped kernel exists (the conv_pipe family: program.hip / conv_pipe.hip), problem by problem otherwise; every row still computes
// exactly what its own micro-batch's call computes in the kernels that serve it.
static int get_group(storm_ncsnpp* h, int P, const int* B, const int* T, int F, std::shared_ptr<storm_ncsnpp::GroupPlan>* out) {
    STORM_CHECK(h != nullptr && P >= 1 && B && T, "storm_ncsnpp_forward_group: bad arguments");
    std::vector<int> key;
    key.push_back(F); key.push_back(storm::switches().batch_invariant != 0 ? 1 : 0);
    for (int g = 0; g < P; ++g) { key.push_back(B[g]); key.push_back(T[g]); }
    {
        std::lock_guard<std::mutex> lk(h->mu);
        auto it = h->groups.find(key);
        if (it != h->groups.end() && it->second->epoch == switch_epoch()) { it->second->used = ++h->gtick; *out = it->second; return STORM_OK; }
        if (it != h->groups.end()) h->groups.erase(it);
    }
    std::shared_ptr<storm_ncsnpp::GroupPlan> gp(new storm_ncsnpp::GroupPlan());
    gp->epoch = switch_epoch();
    for (int g = 0; g < P; ++g) {
        std::shared_ptr<Program> p;
        if (int rc = get_program(h, B[g], F, T[g], &p)) return rc;
        STORM_CHECK(gp->progs.empty() || p->ops.size() == gp->progs[0]->ops.size(), "storm_ncsnpp_forward_group: op lists of different lengths");
        gp->progs.push_back(p);
    }
    std::vector<const storm_op*> ops((size_t)P);
    for (int g = 0; g < P; ++g) ops[(size_t)g] = gp->progs[(size_t)g]->ops.data();
    gp->blob_bytes = up(program_group_blob_bytes(ops.data(), (int)gp->progs[0]->ops.size(), P, h->dtype), ALIGN);
    long long off = gp->blob_bytes;
    for (int g = 0; g < P; ++g) { gp->ws_off.push_back(off); off += up(gp->progs[(size_t)g]->ws_bytes, ALIGN); }
    gp->ws_bytes = off;
    if (gp->blob_bytes > 0) STORM_HIP(hipHostMalloc(reinterpret_cast<void**>(&gp->host_blob), (size_t)gp->blob_bytes));
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->groups.size() >= storm_ncsnpp::MAX_GROUPS) {
        auto old = h->groups.begin();
        for (auto u = h->groups.begin(); u != h->groups.end(); ++u) if (u->second->used < old->second->used) old = u;
        h->groups.erase(old);
    }
    gp->used = ++h->gtick;
    h->groups[key] = gp;
    *out = gp;
    return STORM_OK;
}

extern "C" long long storm_ncsnpp_group_workspace_bytes(storm_ncsnpp* h, int P, const int* B, const int* T, int F) {
    std::shared_ptr<storm_ncsnpp::GroupPlan> gp;
    if (get_group(h, P, B, T, F, &gp) != STORM_OK) return -1;
    return gp->ws_bytes;
}

// parts: P * n_parts device pointers (problem-major) to complex64 [B_p][F][T_p]; t: P pointers to fp32 [B_p] (NULL entries for a
// discriminative net); out: P pointers to complex64 [B_p][F][T_p]; ws: >= storm_ncsnpp_group_workspace_bytes of the same shape list.
extern "C" int storm_ncsnpp_forward_group(storm_ncsnpp* h, int P, const int* B, const int* T, int F, const void* const* parts, int n_parts,
                                          const float* const* t, void* const* out, void* ws, long long ws_bytes, int negate, storm_stream_t s) {
    std::shared_ptr<storm_ncsnpp::GroupPlan> gp;
    if (int rc = get_group(h, P, B, T, F, &gp)) return rc;
    STORM_CHECK(parts && out && ws, "storm_ncsnpp_forward_group: null pointer");
    STORM_CHECK(n_parts == h->cfg.total() / 2, "storm_ncsnpp_forward_group: %d complex input channels given, the network takes %d", n_parts, h->cfg.total() / 2);
    STORM_CHECK(ws_bytes >= gp->ws_bytes, "storm_ncsnpp_forward_group: workspace %lld < %lld bytes", ws_bytes, gp->ws_bytes);
    STORM_CHECK(!h->cfg.conditional() || t != nullptr, "storm_ncsnpp_forward_group: a score network needs t");
    std::vector<void*> bufs((size_t)P * N_BUFS, nullptr);
    std::vector<void* const*> bufp((size_t)P);
    std::vector<const storm_op*> ops((size_t)P);
    std::vector<const void*> sig;                           // what the tables of the grouped launches depend on: the workspace and the weight arena
    sig.push_back(ws); sig.push_back(h->arena);
    for (int g = 0; g < P; ++g) {
        void** b = bufs.data() + (size_t)g * N_BUFS;
        b[BUF_WS] = static_cast<char*>(ws) + gp->ws_off[(size_t)g]; b[BUF_PARAMS] = h->arena;
        for (int j = 0; j < n_parts; ++j) {
            STORM_CHECK(parts[g * n_parts + j] != nullptr, "storm_ncsnpp_forward_group: input %d of problem %d is NULL", j, g);
            b[BUF_IN0 + j] = const_cast<void*>(parts[g * n_parts + j]);
        }
        b[BUF_T] = t ? const_cast<float*>(t[g]) : nullptr; b[BUF_OUT] = out[g];
        bufp[(size_t)g] = b;
        ops[(size_t)g] = gp->progs[(size_t)g]->ops.data();
    }
    const int n_ops = (int)gp->progs[0]->ops.size();
    if (gp->blob_bytes > 0) {
        std::lock_guard<std::mutex> lk(h->mu);               // (one builder; the upload is ordered on the caller's stream before the launches below)
        if (gp->built_for != sig) {
            if (gp->copy_pending) { STORM_HIP(hipEventSynchronize(gp->copied)); gp->copy_pending = false; }
            gp->gops.assign((size_t)n_ops, GroupOp());
            const int n = program_group_build(ops.data(), n_ops, bufp.data(), N_BUFS, P, h->dtype, gp->host_blob, gp->blob_bytes, gp->gops.data(), n_ops, BUF_PARAMS + 1);
            if (n < 0) return n;
            gp->gops.resize((size_t)n);
            gp->built_for = sig;
        }
        // the tables travel with every call: 14 problems x 32 layers are ~1.5 MB, one asynchronous copy ahead of the evaluation's ~100 launches
        // (a cached device copy would be wrong as soon as two callers alternate workspaces or tensors on one handle)
        if (!gp->gops.empty()) {
            STORM_HIP(hipMemcpyAsync(ws, gp->host_blob, (size_t)gp->blob_bytes, hipMemcpyHostToDevice, (hipStream_t)s));
            if (!gp->have_event) { STORM_HIP(hipEventCreate(&gp->copied)); gp->have_event = true; }
            STORM_HIP(hipEventRecord(gp->copied, (hipStream_t)s));
            gp->copy_pending = true;
        }
    }
    h->group_launches.fetch_add((long long)gp->gops.size(), std::memory_order_relaxed);
    return program_run_group(ops.data(), n_ops, bufp.data(), N_BUFS, P, h->dtype, static_cast<const char*>(ws), gp->gops.data(), (int)gp->gops.size(), negate, s);
}

extern "C" long long storm_ncsnpp_group_launches(storm_ncsnpp* h) { return h ? h->group_launches.load(std::memory_order_relaxed) : -1; }
