// Program interpreter: executes a host-planned op list (one NCSN++ forward) with a single
// C-ABI call, so the ~300 kernel launches of a score evaluation cost no Python dispatch.
// Field conventions are mirrored by storm_amd/backbones/plan.py (class Op).
#include <cstring>
#include <vector>
#include "common.h"
#include "conv_params.h"

using namespace storm;

static inline void* resolve(const storm_ref& r, void* const* bufs, int n_bufs, bool& ok) {
    if (r.buf < 0) return nullptr;
    if (r.buf >= n_bufs || bufs[r.buf] == nullptr) { ok = false; return nullptr; }
    return static_cast<char*>(bufs[r.buf]) + r.off;
}

// storm_conv_args of a STORM_OP_CONV op (pointers resolved by the caller; NULL pointers when only the shape matters)
static void conv_args_of(const storm_op& op, void* const* p, int dtype, storm_conv_args& a) {
    const int64_t* i = op.i;
    memset(&a, 0, sizeof(a));
    a.nseg = (int)i[0]; a.B = (int)i[1]; a.H = (int)i[2]; a.W = (int)i[3];
    a.outC = (int)i[4]; a.Cout = (int)i[5]; a.tbias_stride = (int)i[6]; a.out_f32 = (int)i[7];
    const long long hw = (long long)a.H * a.W;
    for (int g = 0; g < 2; ++g) {
        storm_conv_seg& sgm = a.seg[g];
        const int64_t* q = i + 8 + 7 * g;
        sgm.src_a = p[3 * g]; sgm.src_b = p[3 * g + 1]; sgm.w = p[3 * g + 2];
        sgm.Ca = (int)q[0]; sgm.Cb = (int)q[1]; sgm.CinP = (int)q[2]; sgm.w_rows = (int)q[3];
        sgm.ntaps = (int)q[4]; sgm.w_bstride = q[5]; sgm.w_tapstride = q[6];
        sgm.bstride_a = hw * sgm.Ca; sgm.bstride_b = hw * sgm.Cb;
    }
    if (i[22] >= 0) a.seg[0].bstride_a = i[22];
    a.out = p[6]; a.bias = (const float*)p[7]; a.tbias = (const float*)p[8]; a.skip = p[9];
    a.out_bstride = i[23] >= 0 ? i[23] : hw * a.outC;
    a.skip_bstride = hw * a.outC;
    a.scale = op.f[0];
    a.dtype = dtype;
    a.gn_part = (float*)p[10];
    a.seg[0].gn_ss = (const float*)p[11];
    a.seg[0].gn_silu = op.f[1] != 0.f;
    if (p[12] != nullptr) {                                 // split-K scratch: f[2] fp32 slabs [B][H][W][outC] (storm_conv_splitk_bytes at plan time)
        a.splitk_ws = p[12];
        a.splitk_ws_bytes = (long long)op.f[2] * a.B * hw * a.outC * 4;
    }
}

static int run_ops(const storm_op* ops, int n_ops, void* const* bufs, int n_bufs, int dtype, storm_stream_t s,
                   hipEvent_t* ev) {
    STORM_CHECK(ops && bufs && n_ops >= 0, "storm_program_run: bad arguments");
    hipStream_t st = (hipStream_t)s;
    for (int k = 0; k < n_ops; ++k) {
        if (ev) STORM_HIP(hipEventRecord(ev[k], st));
        const storm_op& op = ops[k];
        bool ok = true;
        void* p[STORM_OP_NPTR];
        for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(op.p[j], bufs, n_bufs, ok);
        STORM_CHECK(ok, "storm_program_run: op %d (code %d) references a missing buffer", k, op.code);
        const int64_t* i = op.i;
        int rc = STORM_OK;
        switch (op.code) {
            case STORM_OP_MEMSET:
                STORM_HIP(hipMemsetAsync(p[0], 0, (size_t)i[0], st));
                break;
            case STORM_OP_PACK_INPUT: {
                const float* in[3] = {(const float*)p[0], (const float*)p[1], (const float*)p[2]};
                rc = storm_pack_input(in, (int)i[0], p[3], (int)i[1], (int)i[2], (int)i[3], dtype, s);
                break;
            }
            case STORM_OP_TEMB:
                rc = storm_time_embedding((const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3],
                                          (const float*)p[4], (const float*)p[5], (float*)p[6], (int)i[0], (int)i[1], s);
                break;
            case STORM_OP_DENSE:
                rc = storm_dense((const float*)p[0], (const float*)p[1], (const float*)p[2], (float*)p[3], (int)i[0],
                                 (int)i[1], (int)i[2], s);
                break;
            case STORM_OP_CONV: {
                storm_conv_args a;
                conv_args_of(op, p, dtype, a);
                rc = storm_conv(&a, s);
                break;
            }
            case STORM_OP_GN_STATS:
                rc = storm_gn_stats(p[0], (int)i[0], p[1], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (double*)p[2], dtype, s);
                break;
            case STORM_OP_GN_FINALIZE:
                if (p[5] != nullptr)
                    rc = storm_gn_finalize_ss((const float*)p[0], (int)i[0], (int)i[1], (const float*)p[1], (int)i[2], (int)i[3],
                                              (int)i[4], (int)i[5], (long long)i[6], (const float*)p[3], (const float*)p[4],
                                              op.f[0], (double*)p[2], (float*)p[5], s);
                else
                    rc = storm_gn_finalize((const float*)p[0], (int)i[0], (int)i[1], (const float*)p[1], (int)i[2], (int)i[3],
                                           (int)i[4], (int)i[5], (double*)p[2], s);
                break;
            case STORM_OP_GN_APPLY:
                rc = storm_gn_apply(p[0], (int)i[0], p[1], (int)i[1], (int)i[2], (int)i[3], (int)i[4], (int)i[5],
                                    (const double*)p[2], (const float*)p[3], (const float*)p[4], op.f[0], (int)i[6],
                                    (int)i[7], p[5], p[6], dtype, s);
                break;
            case STORM_OP_FIR_UP:
                rc = storm_fir_up2(p[0], p[1], p[2], (int)i[0], (int)i[1], (int)i[2], (int)i[3], dtype, s);
                break;
            case STORM_OP_FIR_DOWN:
                rc = storm_fir_down2(p[0], p[1], (int)i[0], (int)i[1], (int)i[2], (int)i[3], dtype, s);
                break;
            case STORM_OP_SOFTMAX:
                rc = storm_softmax_rows((const float*)p[0], p[1], (long long)i[0], (int)i[1], (int)i[2], dtype, s);
                break;
            case STORM_OP_ATTENTION:
                rc = storm_attention_ws(p[0], p[1], p[2], (const float*)p[3], p[4], (int)i[0], (int)i[1], (int)i[2], (int)i[3],
                                        (long long)i[1] * i[2], (long long)i[1] * i[2], (long long)i[2] * i[3], (long long)i[1] * i[2],
                                        op.f[0], dtype, p[5], (long long)i[4], s);
                break;
            case STORM_OP_OUTPUT_HEAD:
                rc = storm_output_head(p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], (int)i[0],
                                       (float*)p[4], (int)i[1], (int)i[2], (int)i[3], (int)i[4], dtype, s);
                break;
            case STORM_OP_INPUT_PYRAMID: {
                const float* in[3] = {(const float*)p[0], (const float*)p[1], (const float*)p[2]};
                rc = storm_input_pyramid(i[0] > 0 ? in : nullptr, (int)i[0], p + 3, (int)i[4], (int)i[1], (int)i[2], (int)i[3], dtype, s);
                break;
            }
            case STORM_OP_OUTPUT_PYRAMID:
                rc = storm_output_pyramid(p, (int)i[5], (const float*)p[8], (const float*)p[9], (const float*)p[10], (int)i[0],
                                          (float*)p[11], (int)i[1], (int)i[2], (int)i[3], (int)i[4], dtype, s);
                break;
            default:
                STORM_CHECK(false, "storm_program_run: op %d has unknown code %d", k, op.code);
        }
        if (rc != STORM_OK) return rc;
    }
    if (ev) STORM_HIP(hipEventRecord(ev[n_ops], st));
    return STORM_OK;
}


// ---- grouped evaluation (common.h) ---------------------------------------------------------------------------------------------------------
namespace {
// op k of the P lists groups when it is the same 16-bit 3x3 convolution of the conv_pipe family in every problem
bool group_candidate(const storm_op* const* ops, int k, int P, int dtype) {
    if (dtype != STORM_BF16 && dtype != STORM_F16) return false;
    if (switches().conv_variant >= 0) return false;          // a forced kernel family (tests, A/B) is honoured problem by problem
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_CONV) return false;
        if ((int)o.i[4] <= 128 || (int)o.i[7] != 0 || (int)o.i[8 + 4] != 9) return false;      // outC, out_f32, taps of segment 0
        if (o.i[4] != ops[0][k].i[4] || o.i[5] != ops[0][k].i[5] || o.i[0] != ops[0][k].i[0]) return false;
    }
    return true;
}
long long op_tiles(const storm_op& o) { return (long long)o.i[1] * cdiv(o.i[2], 8) * cdiv(o.i[3], 32); }   // B x 8-row x 32-pixel tiles
// the output pyramid's 3x3 convolutions to <= 4 planes (conv_narrow.hip): 4 per evaluation, one 8-wave workgroup per CU walking 20 x 32-pixel tiles
bool narrow_candidate(const storm_op* const* ops, int k, int P, int dtype) {
    if (dtype != STORM_BF16 && dtype != STORM_F16) return false;
    if (switches().conv_variant >= 0) return false;
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_CONV || (int)o.i[4] != 8 || (int)o.i[0] != 1 || (int)o.i[8 + 4] != 9 || (int)o.i[7] != 0) return false;
        if (o.i[5] != ops[0][k].i[5] || o.i[8] != ops[0][k].i[8] || o.i[9] != 0) return false;
    }
    return true;
}
// the 8-channel-input convolutions (conv_thin.hip: the stem and the three input-skip 1x1s)
bool thin_candidate(const storm_op* const* ops, int k, int P, int dtype) {
    if (dtype != STORM_BF16 && dtype != STORM_F16) return false;
    if (switches().conv_variant >= 0) return false;
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_CONV || (int)o.i[0] != 1 || (int)o.i[8] != 8 || (int)o.i[9] != 0 || (int)o.i[4] < 64 || (int)o.i[7] != 0) return false;
        if (o.i[4] != ops[0][k].i[4] || o.i[8 + 4] != ops[0][k].i[8 + 4]) return false;
    }
    return true;
}
long long narrow_tiles(const storm_op& o) { return (long long)o.i[1] * cdiv(o.i[2], 20) * cdiv(o.i[3], 32); }
long long align256(long long v) { return (v + 255) / 256 * 256; }
}  // namespace

// the GroupNorm finalizes (45 per evaluation, ~6 us each whatever the problem's size): same group count in every problem
static bool fin_candidate(const storm_op* const* ops, int k, int P) {
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_GN_FINALIZE || o.i[5] != ops[0][k].i[5] || o.i[0] + o.i[2] != ops[0][k].i[0] + ops[0][k].i[2]) return false;
    }
    return true;
}

// the fused attention of the bottleneck (16-bit): STORM_OP_ATTENTION (q, k, vT, bias, out, scratch; B, L, C, ldv; scale)
static bool attn_candidate(const storm_op* const* ops, int k, int P, int dtype) {
    if (dtype != STORM_BF16 && dtype != STORM_F16) return false;
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_ATTENTION || o.i[2] != ops[0][k].i[2] || o.f[0] != ops[0][k].f[0]) return false;
    }
    return true;
}
static long long attn_items(const storm_op* const* ops, int k, int P) {
    long long n = 0;
    for (int g = 0; g < P; ++g) n += ops[g][k].i[0] * attn_query_blocks((int)ops[g][k].i[1]);
    return n;
}

static bool fir_candidate(const storm_op* const* ops, int k, int P) {
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if ((o.code != STORM_OP_FIR_UP && o.code != STORM_OP_FIR_DOWN) || o.code != ops[0][k].code || o.i[3] != ops[0][k].i[3]) return false;
    }
    return true;
}

// GroupNorm-apply + SiLU + FIR x2 of h and x (the up / down resblocks' first op: STORM_OP_GN_APPLY with resample 1 / 2): same channels, groups and
// affine parameters in every problem
static bool apply_candidate(const storm_op* const* ops, int k, int P, int dtype, GnApplyGroupPlan* plan) {
    if (dtype != STORM_BF16 && dtype != STORM_F16) return false;
    const storm_op& o0 = ops[0][k];
    if (o0.code != STORM_OP_GN_APPLY || (o0.i[7] != 1 && o0.i[7] != 2) || o0.i[6] == 0 || P > 64) return false;
    int B[64], H[64], W[64];
    for (int g = 0; g < P; ++g) {
        const storm_op& o = ops[g][k];
        if (o.code != STORM_OP_GN_APPLY || o.i[0] != o0.i[0] || o.i[1] != o0.i[1] || o.i[5] != o0.i[5] || o.i[6] != o0.i[6] || o.i[7] != o0.i[7] || o.f[0] != o0.f[0]) return false;
        if (o.p[6].buf < 0) return false;                    // (the resampling form always writes both tensors)
        B[g] = (int)o.i[2]; H[g] = (int)o.i[3]; W[g] = (int)o.i[4];
    }
    GnApplyGroupPlan pl;
    if (!gn_apply_group_plan((int)o0.i[7], (int)(o0.i[0] + o0.i[1]), P, B, H, W, dtype, pl)) return false;
    if (plan) *plan = pl;
    return true;
}

long long storm::program_group_blob_bytes(const storm_op* const* ops, int n_ops, int P, int dtype) {
    long long n = 0;
    if (P < 2) return 0;
    for (int k = 0; k < n_ops; ++k) {
        if (fir_candidate(ops, k, P)) {
            long long items = 0;
            for (int g = 0; g < P; ++g) items += ops[g][k].i[0];
            n += align256((long long)P * sizeof(FirProblem)) + align256(items * 8);
            continue;
        }
        if (attn_candidate(ops, k, P, dtype)) { n += align256((long long)P * sizeof(AttnProblem)) + align256(attn_items(ops, k, P) * (long long)sizeof(AttnItem)); continue; }
        if (fin_candidate(ops, k, P)) {
            long long items = 0;
            for (int g = 0; g < P; ++g) items += ops[g][k].i[4];
            n += align256((long long)P * sizeof(GnFinProblem)) + align256(items * 8);
            continue;
        }
        { GnApplyGroupPlan pl; if (apply_candidate(ops, k, P, dtype, &pl)) { n += align256((long long)P * sizeof(GnApplyProblem)) + align256(pl.items * 8); continue; } }
        if (narrow_candidate(ops, k, P, dtype)) {
            long long t = 0;
            for (int g = 0; g < P; ++g) t += narrow_tiles(ops[g][k]);
            n += conv_narrow_group_bytes(P) + align256(t * (long long)sizeof(pipe::GroupTile));
            continue;
        }
        if (thin_candidate(ops, k, P, dtype)) { n += conv_thin_group_bytes(P); continue; }
        if (!group_candidate(ops, k, P, dtype)) continue;
        long long t = 0;
        for (int g = 0; g < P; ++g) t += op_tiles(ops[g][k]);
        n += align256((long long)P * sizeof(pipe::PipeParams)) + align256(t * (long long)sizeof(pipe::GroupTile));
    }
    return n;
}

int storm::program_group_build(const storm_op* const* ops, int n_ops, void* const* const* bufs, int n_bufs, int P, int dtype, char* host_blob,
                               long long blob_bytes, GroupOp* gops, int max_gops, int stable_bufs) {
    int n = 0;
    long long off = 0;
    std::vector<storm_conv_args> args((size_t)P);
    for (int k = 0; k < n_ops; ++k) {
        bool stable = true;                                  // (tables are rebuilt only when a stable buffer moves: nothing else may be in them)
        for (int g = 0; g < P && stable; ++g)
            for (int j = 0; j < STORM_OP_NPTR; ++j) stable = stable && ops[g][k].p[j].buf < stable_bufs;
        if (!stable) continue;
        if (fir_candidate(ops, k, P)) {
            long long items = 0;
            for (int g = 0; g < P; ++g) items += ops[g][k].i[0];
            const long long tab = align256((long long)P * sizeof(FirProblem)), til = align256(items * 8);
            STORM_CHECK(off + tab + til <= blob_bytes && n < max_gops && items < 65536, "storm_program_group: table blob too small");
            FirProblem* t = reinterpret_cast<FirProblem*>(host_blob + off);
            int* it = reinterpret_cast<int*>(host_blob + off + tab);
            const int up = ops[0][k].code == STORM_OP_FIR_UP ? 1 : 2;
            long long ni = 0;
            int max_blocks = 0;
            for (int g = 0; g < P; ++g) {
                const storm_op& o = ops[g][k];
                bool ok = true;
                void* p[STORM_OP_NPTR];
                for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(o.p[j], bufs[g], n_bufs, ok);
                STORM_CHECK(ok, "storm_program_group: op %d of problem %d references a missing buffer", k, g);
                memset(&t[g], 0, sizeof(FirProblem));
                // run_ops: FIR_UP (x, add, out; B, H, W, C) / FIR_DOWN (x, out; B, H, W, C)
                const int nb = up == 1 ? fir_group_problem(1, p[0], p[1], p[2], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], t[g])
                                       : fir_group_problem(2, p[0], nullptr, p[1], (int)o.i[0], (int)o.i[1], (int)o.i[2], (int)o.i[3], t[g]);
                if (nb > max_blocks) max_blocks = nb;
                for (int b = 0; b < (int)o.i[0]; ++b) { it[2 * ni] = g; it[2 * ni + 1] = b; ++ni; }
            }
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 4; go.outC = (int)ops[0][k].i[3]; go.bn = (max_blocks << 2) | up; go.table_off = off; go.tiles_off = off + tab; go.ntiles = ni;
            off += tab + til;
            continue;
        }
        if (attn_candidate(ops, k, P, dtype)) {
            const long long items = attn_items(ops, k, P);
            const long long tab = align256((long long)P * sizeof(AttnProblem)), til = align256(items * (long long)sizeof(AttnItem));
            STORM_CHECK(off + tab + til <= blob_bytes && n < max_gops && items < (1LL << 31), "storm_program_group: table blob too small");
            AttnProblem* t = reinterpret_cast<AttnProblem*>(host_blob + off);
            AttnItem* it = reinterpret_cast<AttnItem*>(host_blob + off + tab);
            long long ni = 0;
            const void* bias = nullptr;
            for (int g = 0; g < P; ++g) {
                const storm_op& o = ops[g][k];
                bool ok = true;
                void* p[STORM_OP_NPTR];
                for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(o.p[j], bufs[g], n_bufs, ok);
                STORM_CHECK(ok, "storm_program_group: op %d of problem %d references a missing buffer", k, g);
                const int B = (int)o.i[0], L = (int)o.i[1], Cc = (int)o.i[2], ldv = (int)o.i[3];
                AttnProblem& q = t[g];
                memset(&q, 0, sizeof(q));
                q.q = p[0]; q.k = p[1]; q.vT = p[2]; q.out = p[4]; q.L = L; q.ldv = ldv;
                q.q_bs = (long long)L * Cc; q.k_bs = (long long)L * Cc; q.v_bs = (long long)Cc * ldv; q.o_bs = (long long)L * Cc;     // (run_ops' strides)
                bias = p[3];
                const int nq = attn_query_blocks(L);
                // long rows first inside the list would balance better; the order is the problems' (deterministic, the result does not depend on it)
                for (int b = 0; b < B; ++b)
                    for (int qb = 0; qb < nq; ++qb) { AttnItem& a = it[ni++]; a.problem = g; a.b = b; a.qblock = qb; a.pad_ = 0; }
            }
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 5; go.outC = (int)ops[0][k].i[2]; go.bn = 0; go.table_off = off; go.tiles_off = off + tab; go.ntiles = ni;
            go.aux = bias; go.faux = ops[0][k].f[0];
            off += tab + til;
            continue;
        }
        if (fin_candidate(ops, k, P)) {
            long long items = 0;
            for (int g = 0; g < P; ++g) items += ops[g][k].i[4];
            const long long tab = align256((long long)P * sizeof(GnFinProblem)), til = align256(items * 8);
            STORM_CHECK(off + tab + til <= blob_bytes && n < max_gops && items < 65536, "storm_program_group: table blob too small");
            GnFinProblem* t = reinterpret_cast<GnFinProblem*>(host_blob + off);
            int* it = reinterpret_cast<int*>(host_blob + off + tab);
            long long ni = 0;
            for (int g = 0; g < P; ++g) {
                const storm_op& o = ops[g][k];
                bool ok = true;
                void* p[STORM_OP_NPTR];
                for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(o.p[j], bufs[g], n_bufs, ok);
                STORM_CHECK(ok, "storm_program_group: op %d of problem %d references a missing buffer", k, g);
                GnFinProblem& q = t[g];
                memset(&q, 0, sizeof(q));
                q.pa = (const float*)p[0]; q.pb = (const float*)p[1]; q.stats = (double*)p[2]; q.gamma = (const float*)p[3]; q.beta = (const float*)p[4];
                q.ss = (float*)p[5]; q.count = p[5] != nullptr ? (long long)o.i[6] : 0; q.Ca = (int)o.i[0]; q.tiles_a = (int)o.i[1]; q.Cb = (int)o.i[2];
                q.tiles_b = (int)o.i[3]; q.eps = p[5] != nullptr ? o.f[0] : 0.f;
                for (int b = 0; b < (int)o.i[4]; ++b) { it[2 * ni] = g; it[2 * ni + 1] = b; ++ni; }
            }
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 1; go.outC = (int)ops[0][k].i[5]; go.bn = 0; go.table_off = off; go.tiles_off = off + tab; go.ntiles = ni;
            off += tab + til;
            continue;
        }
        { GnApplyGroupPlan pl;
          if (apply_candidate(ops, k, P, dtype, &pl)) {
            const long long tab = align256((long long)P * sizeof(GnApplyProblem)), til = align256(pl.items * 8);
            STORM_CHECK(off + tab + til <= blob_bytes && n < max_gops, "storm_program_group: table blob too small");
            GnApplyProblem* t = reinterpret_cast<GnApplyProblem*>(host_blob + off);
            GnFinItem* it = reinterpret_cast<GnFinItem*>(host_blob + off + tab);
            const storm_op& o0 = ops[0][k];
            long long ni = 0;
            const void *gamma = nullptr, *beta = nullptr;
            for (int g = 0; g < P; ++g) {
                const storm_op& o = ops[g][k];
                bool ok = true;
                void* p[STORM_OP_NPTR];
                for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(o.p[j], bufs[g], n_bufs, ok);
                STORM_CHECK(ok, "storm_program_group: op %d of problem %d references a missing buffer", k, g);
                // run_ops: GN_APPLY (xa, xb, stats, gamma, beta, out_act, out_raw; Ca, Cb, B, H, W, G, silu, resample; eps)
                GnApplyProblem& q = t[g];
                memset(&q, 0, sizeof(q));
                q.xa = p[0]; q.xb = p[1]; q.stats = (const double*)p[2]; q.out_act = p[5]; q.out_raw = p[6]; q.H = (int)o.i[3]; q.W = (int)o.i[4];
                gamma = p[3]; beta = p[4];
                ni += gn_apply_group_problem((int)o0.i[7], (int)(o0.i[0] + o0.i[1]), (int)o.i[2], dtype, pl, g, q, it + ni);
            }
            STORM_CHECK(ni == pl.items, "storm_program_group: GroupNorm-apply items %lld != %lld", ni, pl.items);
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 6; go.outC = (int)o0.i[0]; go.bn = (int)o0.i[1]; go.table_off = off; go.tiles_off = off + tab; go.ntiles = ni;
            go.aux = gamma; go.aux2 = beta; go.faux = o0.f[0];
            go.x[0] = (int)o0.i[5]; go.x[1] = (int)o0.i[7]; go.x[2] = (pl.rows_per_strip << 1) | (pl.share ? 1 : 0); go.x[3] = pl.max_cols;
            off += tab + til;
            continue;
          } }
        const bool narrow = narrow_candidate(ops, k, P, dtype), thin = !narrow && thin_candidate(ops, k, P, dtype);
        if (!narrow && !thin && !group_candidate(ops, k, P, dtype)) continue;
        long long t = 0;
        for (int g = 0; g < P; ++g) {
            bool ok = true;
            void* p[STORM_OP_NPTR];
            for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = resolve(ops[g][k].p[j], bufs[g], n_bufs, ok);
            STORM_CHECK(ok, "storm_program_group: op %d of problem %d references a missing buffer", k, g);
            conv_args_of(ops[g][k], p, dtype, args[(size_t)g]);
            args[(size_t)g].splitk_ws = nullptr; args[(size_t)g].splitk_ws_bytes = 0;      // (a grouped launch never splits K)
            t += narrow ? narrow_tiles(ops[g][k]) : op_tiles(ops[g][k]);
        }
        if (thin) {
            const long long img = conv_thin_group_bytes(P);
            STORM_CHECK(off + img <= blob_bytes && n < max_gops, "storm_program_group: table blob too small");
            const long long nt = conv_thin_group_prepare(args.data(), P, host_blob + off);
            if (nt <= 0) continue;
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 3; go.outC = P; go.bn = args[0].seg[0].ntaps; go.table_off = off; go.tiles_off = off; go.ntiles = nt;
            off += img;
            continue;
        }
        if (narrow) {
            const long long tabn = conv_narrow_group_bytes(P), tiln = align256(t * (long long)sizeof(pipe::GroupTile));
            STORM_CHECK(off + tabn + tiln <= blob_bytes && n < max_gops, "storm_program_group: table blob too small");
            const long long gotn = conv_narrow_group_prepare(args.data(), P, host_blob + off, reinterpret_cast<pipe::GroupTile*>(host_blob + off + tabn), t);
            if (gotn != t) continue;
            GroupOp& go = gops[n++];
            go.k = k; go.kind = 2; go.outC = args[0].seg[0].Ca; go.bn = (args[0].seg[0].gn_ss != nullptr ? 2 : 0) | (args[0].seg[0].gn_silu ? 1 : 0);
            go.table_off = off; go.tiles_off = off + tabn; go.ntiles = t;
            off += tabn + tiln;
            continue;
        }
        const long long tab = align256((long long)P * sizeof(pipe::PipeParams)), til = align256(t * (long long)sizeof(pipe::GroupTile));
        STORM_CHECK(off + tab + til <= blob_bytes && n < max_gops, "storm_program_group: table blob too small");
        const long long got = conv_pipe_group_prepare(args.data(), P, reinterpret_cast<pipe::PipeParams*>(host_blob + off),
                                                      reinterpret_cast<pipe::GroupTile*>(host_blob + off + tab), t);
        if (got != t) continue;                              // outside the pipelined kernel's coverage: runs problem by problem
        GroupOp& go = gops[n++];
        go.k = k; go.outC = args[0].outC; go.table_off = off; go.tiles_off = off + tab; go.ntiles = t;
        go.bn = t * cdiv(args[0].outC, 256) >= 512 ? 256 : 128;   // the ladder's rule for the pipelined kernel's two tiles, on the GROUP's tile count
        go.kind = 0;
        off += tab + til;
    }
    return n;
}

int storm::program_run_group(const storm_op* const* ops, int n_ops, void* const* const* bufs, int n_bufs, int P, int dtype, const char* dev_blob,
                             const GroupOp* gops, int n_gops, int negate, storm_stream_t s) {
    int gi = 0;
    for (int k = 0; k < n_ops; ++k) {
        if (gi < n_gops && gops[gi].k == k) {
            const GroupOp& go = gops[gi++];
            if (go.kind == 6) {
                GnApplyGroupPlan pl;
                pl.rows_per_strip = go.x[2] >> 1; pl.share = go.x[2] & 1; pl.max_cols = go.x[3]; pl.items = go.ntiles;
                if (int rc = launch_gn_apply_group(go.x[1], reinterpret_cast<const GnApplyProblem*>(dev_blob + go.table_off), dev_blob + go.tiles_off, pl, go.outC, go.bn, go.x[0],
                                                   static_cast<const float*>(go.aux), static_cast<const float*>(go.aux2), go.faux, dtype, (hipStream_t)s)) return rc;
                continue;
            }
            if (go.kind == 5) {
                if (int rc = launch_attention_group(reinterpret_cast<const AttnProblem*>(dev_blob + go.table_off), reinterpret_cast<const AttnItem*>(dev_blob + go.tiles_off),
                                                    (int)go.ntiles, static_cast<const float*>(go.aux), go.outC, go.faux, dtype, (hipStream_t)s)) return rc;
                continue;
            }
            if (go.kind == 4) {
                if (int rc = launch_fir_group(go.bn & 3, reinterpret_cast<const FirProblem*>(dev_blob + go.table_off), dev_blob + go.tiles_off, (int)go.ntiles, go.bn >> 2, go.outC,
                                              dtype, (hipStream_t)s)) return rc;
                continue;
            }
            if (go.kind == 3) {
                if (int rc = launch_conv_thin_group(dev_blob + go.table_off, go.outC, go.ntiles, go.bn, dtype, (hipStream_t)s)) return rc;
                continue;
            }
            if (go.kind == 2) {
                storm_conv_args a0;
                memset(&a0, 0, sizeof(a0));
                a0.dtype = dtype; a0.seg[0].Ca = go.outC; a0.seg[0].gn_ss = (go.bn & 2) ? reinterpret_cast<const float*>(uintptr_t(16)) : nullptr; a0.seg[0].gn_silu = go.bn & 1;
                if (int rc = launch_conv_narrow_group(a0, dev_blob + go.table_off, reinterpret_cast<const pipe::GroupTile*>(dev_blob + go.tiles_off), go.ntiles, (hipStream_t)s)) return rc;
                continue;
            }
            if (go.kind == 1) {
                if (int rc = launch_gn_finalize_group(reinterpret_cast<const GnFinProblem*>(dev_blob + go.table_off), dev_blob + go.tiles_off, (int)go.ntiles, go.outC,
                                                      (hipStream_t)s)) return rc;
                continue;
            }
            if (int rc = launch_conv_pipe_group(reinterpret_cast<const pipe::PipeParams*>(dev_blob + go.table_off),
                                                reinterpret_cast<const pipe::GroupTile*>(dev_blob + go.tiles_off), go.ntiles, go.outC, go.bn, dtype,
                                                (hipStream_t)s)) return rc;
            continue;
        }
        for (int g = 0; g < P; ++g) {
            if (k == n_ops - 1 && (ops[g][k].code == STORM_OP_OUTPUT_HEAD || ops[g][k].code == STORM_OP_OUTPUT_PYRAMID)) {
                storm_op head = ops[g][k];
                head.i[4] = negate ? 1 : 0;
                if (int rc = run_ops(&head, 1, bufs[g], n_bufs, dtype, s, nullptr)) return rc;
            } else if (int rc = run_ops(ops[g] + k, 1, bufs[g], n_bufs, dtype, s, nullptr)) return rc;
        }
    }
    return STORM_OK;
}

extern "C" int storm_program_run(const storm_op* ops, int n_ops, void* const* bufs, int n_bufs, int dtype,
                                 storm_stream_t s) {
    return run_ops(ops, n_ops, bufs, n_bufs, dtype, s, nullptr);
}

// Profiling variant: brackets every op with HIP events ON THE LAUNCH STREAM and returns the
// elapsed milliseconds per op (host array ms[n_ops]); synchronises the stream at the end.
extern "C" int storm_program_run_timed(const storm_op* ops, int n_ops, void* const* bufs, int n_bufs, int dtype,
                                       storm_stream_t s, float* ms) {
    STORM_CHECK(ms != nullptr && n_ops > 0, "storm_program_run_timed: bad arguments");
    hipEvent_t* ev = new hipEvent_t[n_ops + 1];
    int created = 0, rc = STORM_OK;
    for (; created <= n_ops; ++created)
        if (hipEventCreate(&ev[created]) != hipSuccess) { storm::set_error("hipEventCreate failed"); rc = STORM_ERR_HIP; break; }
    if (rc == STORM_OK) rc = run_ops(ops, n_ops, bufs, n_bufs, dtype, s, ev);
    if (rc == STORM_OK && hipEventSynchronize(ev[n_ops]) != hipSuccess) { storm::set_error("hipEventSynchronize failed"); rc = STORM_ERR_HIP; }
    if (rc == STORM_OK)
        for (int k = 0; k < n_ops; ++k)
            if (hipEventElapsedTime(&ms[k], ev[k], ev[k + 1]) != hipSuccess) { storm::set_error("hipEventElapsedTime failed"); rc = STORM_ERR_HIP; break; }
    for (int k = 0; k < created; ++k) (void)hipEventDestroy(ev[k]);
    delete[] ev;
    return rc;
}

// Name of the kernel op k of the program launches (conv ops; "" otherwise): the dispatch decision depends on shapes, dtype
// and which optional pointers are present, not on their values.
extern "C" const char* storm_program_kernel_name(const storm_op* ops, int k, int dtype) {
    if (ops == nullptr || k < 0 || ops[k].code != STORM_OP_CONV) return "";
    void* p[STORM_OP_NPTR];
    for (int j = 0; j < STORM_OP_NPTR; ++j) p[j] = ops[k].p[j].buf >= 0 ? reinterpret_cast<void*>(uintptr_t(16)) : nullptr;
    storm_conv_args a;
    conv_args_of(ops[k], p, dtype, a);
    return storm_conv_kernel_name(&a);
}
