// frontier_model.cpp -- ORACLE (test infrastructure only; never on the product path).
//
// A scalar CPU model of the BUILD's algorithm (stcsp-solver_amd/csrc/engine.hip): frontier
// search over immutable domain-bitset blocks, propagation to the full GAC fixpoint, candidate
// records, state table, raw edge log, the same sharded stepping interface. It exists for two
// jobs the reference-faithful restatement (ref_dfs.cpp) cannot do:
//   (1) checking the device kernels at the granularity they work at -- the GAC fixpoint of a
//       node is unique, so a kernel's propagated block must equal this model's bit for bit --
//       and running the engine's host-side constraint-set manager + bytecode compiler (it links
//       the product's cset.cpp and interprets the SAME flat program the device executes);
//   (2) standing in for the HIP engine in the world_size-2 gloo tests of the multi-shard
//       driver on machines without a GPU.
// Its automaton is checked against ref_dfs.cpp (and so against the reference's recorded
// outputs) in tests/test_oracle.py.
//
// Reference behaviour it must agree with: see the citations in engine.hip; the propagator here
// is the textbook definition (a value stays iff some satisfying tuple of current values
// contains it) rather than the reference's bounds-only AC-3 (src/solveralgorithm.cpp:476-523).
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../stcsp-solver_amd/csrc/cset.hpp"
#include "../stcsp-solver_amd/csrc/okfix.hpp"
#include "stcsp_engine.h"

using namespace stcsp;

namespace {

struct Model {
    SetManager mgr;
    FlatProgram prog;
    stcsp_options opt{};
    int N = 0, K = 2, NK = 0, NS = 0, CS = 0, KL = 0, sig_len = 0;
    std::vector<uint32_t> init;
    std::deque<std::vector<uint32_t>> open;               // node records (NS words)
    std::vector<std::vector<uint32_t>> outbox;            // per peer: concatenated candidates
    std::vector<uint32_t> keys;                           // state keys, KL words each
    std::map<std::vector<uint32_t>, uint32_t> table;      // key -> local state index
    std::vector<int64_t> e_src, e_dst;
    std::vector<int32_t> e_val;
    stcsp_counters ctr{};
    bool truncated = false, begun = false;
    std::chrono::steady_clock::time_point t0;
    std::string err;
    // export storage
    std::vector<int32_t> r_cid, r_sig;
    std::vector<uint8_t> r_fail, r_issig;
    std::vector<int64_t> r_esrc, r_edst;
    std::vector<int32_t> r_eval;

    int recompile() {
        int rc = mgr.compile(prog);
        if (rc != STCSP_OK) err = mgr.error;
        return rc;
    }
    int create(const stcsp_problem *p, const stcsp_options *o) {
        if (o) opt = *o;
        if (opt.world <= 0) opt.world = 1;
        int rc = mgr.init(p, opt.world > 1);
        if (rc != STCSP_OK) {
            err = mgr.error;
            return rc;
        }
        N = mgr.N;
        K = mgr.K;
        NK = N * K;
        for (int v = 0; v < N; v++)
            if ((long long)mgr.ub[v] - mgr.lb[v] + 1 > 32) {
                err = "domain wider than 32 values";
                return STCSP_E_UNSUPPORTED;
            }
        sig_len = mgr.n_sig + mgr.n_until_cons;
        KL = 1 + sig_len;
        NS = node_stride(N, K);
        CS = cand_stride(N, K, sig_len);
        init.resize(N);
        for (int v = 0; v < N; v++) {
            int w = mgr.ub[v] - mgr.lb[v] + 1;
            init[v] = w >= 32 ? 0xffffffffu : ((1u << w) - 1u);
        }
        outbox.resize(opt.world);
        return recompile();
    }

    // scalar interpreter of the device bytecode (same semantics as eval_program in engine.hip)
    int eval(const ConDesc &C, const int *vals) {
        int stk[256];
        int sp = 0, t = 0;
        bool valid = true;
        uint32_t dead = 0;
        const int32_t *code = prog.code.data();
        for (int pc = C.code_off;;) {
            int w = code[pc++], op = w & 255, arg = w >> 8;
            switch (op) {
                case OP_END: return t;
                case OP_CONST: stk[sp++] = t; t = code[pc++]; break;
                case OP_VAR: stk[sp++] = t; t = vals[arg]; break;
                case OP_ARR: {
                    int off = mgr.array_off[arg], size = mgr.array_off[arg + 1] - off;
                    bool inr = (unsigned)t < (unsigned)size;
                    if (!inr && dead == 0) valid = false;
                    t = inr ? mgr.array_data[off + t] : 0;
                    break;
                }
                case OP_ABS: t = t < 0 ? (int)(0u - (unsigned)t) : t; break;
                case OP_NOT: t = (t == 0); break;
                case OP_MASK_T:
                case OP_MASK_F: {
                    int v = arg == 0 ? t : stk[sp - arg];
                    bool live = (op == OP_MASK_T) ? (v != 0) : (v == 0);
                    dead = (dead << 1) | (live ? 0u : 1u);
                    break;
                }
                case OP_MASK_POP: dead >>= 1; break;
                case OP_SEL_IF: {
                    int b = t, a = stk[sp - 1], c = stk[sp - 2];
                    sp -= 2;
                    t = c ? a : b;
                    break;
                }
                case OP_SEL_AND: { int a = stk[--sp]; t = a ? t : 0; break; }
                case OP_SEL_OR: { int a = stk[--sp]; t = a ? 1 : t; break; }
                case OP_SEL_IMPLY: { int a = stk[--sp]; t = (a == 0) ? 1 : (a <= t); break; }
                default: {
                    int b = t, a = stk[--sp], r = 0;
                    switch (op) {
                        case OP_ADD: r = (int)((unsigned)a + (unsigned)b); break;
                        case OP_SUB: r = (int)((unsigned)a - (unsigned)b); break;
                        case OP_MUL: r = (int)((unsigned)a * (unsigned)b); break;
                        case OP_DIV: r = (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : a / b; break;
                        case OP_MOD: r = (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : a % b; break;
                        case OP_LT: r = a < b; break;
                        case OP_GT: r = a > b; break;
                        case OP_LE: r = a <= b; break;
                        case OP_GE: r = a >= b; break;
                        case OP_EQ: r = a == b; break;
                        case OP_NE: r = a != b; break;
                        default: break;
                    }
                    t = (C.uses_valid && !valid) ? 0 : r;
                }
            }
        }
    }

    // GAC on one point constraint at one point: keep exactly the values that occur in some
    // satisfying tuple of the current domains
    bool revise_point(const ConDesc &C, int p, uint32_t *blk, std::vector<int> &changed) {
        int s = C.scope_len;
        const int32_t *sc = prog.scope.data() + C.scope_off;
        std::vector<uint32_t> D(s), supp(s, 0);
        std::vector<int> vals(s), bit(s);
        for (int j = 0; j < s; j++) {
            D[j] = blk[p * N + sc[j]];
            if (!D[j]) return false;
            bit[j] = __builtin_ctz(D[j]);
            vals[j] = mgr.lb[sc[j]] + bit[j];
        }
        ctr.revisions++;
        for (;;) {
            ctr.evaluations++;
            if (eval(C, vals.data()))
                for (int j = 0; j < s; j++) supp[j] |= 1u << bit[j];
            bool all = true;
            for (int j = 0; j < s && all; j++) all = supp[j] == D[j];
            if (all) break;
            int j = 0;  // odometer, first scope variable fastest
            for (; j < s; j++) {
                uint32_t rest = D[j] & ~((2u << bit[j]) - 1u);
                if (bit[j] < 31 && rest) {
                    bit[j] = __builtin_ctz(rest);
                    vals[j] = mgr.lb[sc[j]] + bit[j];
                    break;
                }
                bit[j] = __builtin_ctz(D[j]);
                vals[j] = mgr.lb[sc[j]] + bit[j];
            }
            if (j == s) break;
        }
        for (int j = 0; j < s; j++) {
            if (!supp[j]) return false;
            if (supp[j] != D[j]) {
                blk[p * N + sc[j]] = supp[j];
                changed.push_back(sc[j]);
            }
        }
        return true;
    }

    // propagate a block to the fixpoint under constraint set `set`
    bool propagate(int set, uint32_t expire, uint32_t *blk) {
        const SetDesc &S = prog.sets[set];
        std::vector<uint8_t> dirty(S.ncons, 1);
        ctr.gac_calls++;
        for (;;) {
            int ci = -1;
            for (int i = 0; i < S.ncons; i++)
                if (dirty[i]) {
                    ci = i;
                    break;
                }
            if (ci < 0) return true;
            dirty[ci] = 0;
            const ConDesc &C = prog.cons[S.con_begin + ci];
            std::vector<int> changed;
            bool self_again = false;
            if (C.type == CT_NEXT) {
                int sh = mgr.lb[C.x] - mgr.lb[C.y];
                ctr.revisions++;
                for (int p = 0; p + 1 < K; p++) {
                    uint32_t &DX = blk[p * N + C.x], &DY = blk[(p + 1) * N + C.y];
                    uint32_t Yal = sh >= 0 ? (sh < 32 ? DY >> sh : 0u) : (-sh < 32 ? DY << -sh : 0u);
                    uint32_t m = DX & Yal;
                    if (!m) return false;
                    uint32_t newY = sh >= 0 ? (m << sh) : (m >> -sh);
                    if (m != DX) { DX = m; changed.push_back(C.x); self_again = true; }
                    if (newY != DY) { DY = newY; changed.push_back(C.y); self_again = true; }
                }
            } else if (C.type == CT_POINT) {
                for (int p = 0; p < C.npoints; p++)
                    if (!revise_point(C, p, blk, changed)) return false;
            } else if (C.type == CT_UNTIL) {
                ctr.revisions++;
                if (!((expire >> C.until_ordinal) & 1u)) {
                    uint32_t DX = blk[C.x], DY = blk[C.y];
                    if (__builtin_popcount(DX) == 1 && __builtin_popcount(DY) == 1) {
                        int vx = mgr.lb[C.x] + __builtin_ctz(DX), vy = mgr.lb[C.y] + __builtin_ctz(DY);
                        if (vx != 1 && vy != 1) return false;
                    }
                }
            }
            for (int v : changed) {
                const uint32_t *row = prog.varcons.data() + S.varcons_off + (size_t)v * S.cw;
                for (int i = 0; i < S.ncons; i++)
                    if ((row[i / 32] >> (i % 32)) & 1u) dirty[i] = 1;
            }
            if (C.type == CT_POINT || !self_again) dirty[ci] = 0;
        }
    }

    unsigned long long key_hash(const uint32_t *kw) const {
        return stcsp::key_hash(kw, KL);
    }

    // expand one node: propagate, classify, emit children / a candidate
    int expand(std::vector<uint32_t> node) {
        uint32_t h0 = node[0], h1 = node[1], expire = node[3];
        int set = (int)node[2];
        uint32_t *blk = node.data() + 4;
        ctr.search_nodes++;
        if (!propagate(set, expire, blk)) {
            ctr.fails++;
            return STCSP_OK;
        }
        int bvar = -1;
        for (int v = 0; v < N && bvar < 0; v++)
            if (__builtin_popcount(blk[v]) > 1) bvar = v;
        if (bvar >= 0) {
            uint32_t D = blk[bvar];
            int lo = __builtin_ctz(D), hi = 31 - __builtin_clz(D), mid = lo + (hi - lo) / 2;
            uint32_t lowmask = mid >= 31 ? 0xffffffffu : ((2u << mid) - 1u);
            std::vector<uint32_t> a = node, b = node;
            a[4 + bvar] = D & lowmask;
            b[4 + bvar] = D & ~lowmask;
            // LIFO: the lower half is explored first, like the reference's DFS
            open.push_back(std::move(b));
            open.push_back(std::move(a));
            return STCSP_OK;
        }
        ctr.leaves++;
        int next_set = set;
        if (!mgr.sets[set]->self_loop) {
            std::vector<int> fv;
            for (int v : mgr.sets[set]->first_vars) fv.push_back(mgr.lb[v] + __builtin_ctz(blk[v]));
            size_t before = mgr.sets.size();
            size_t trans_before = mgr.sets[set]->trans.size();
            next_set = mgr.transition(set, fv);
            if (next_set < 0) {
                err = mgr.error;
                return next_set;
            }
            if (mgr.sets.size() != before || mgr.sets[set]->trans.size() != trans_before) {
                int rc = recompile();
                if (rc != STCSP_OK) return rc;
            }
        }
        std::vector<uint32_t> rec(CS, 0u);
        rec[0] = h0;
        rec[1] = h1;
        rec[2] = (uint32_t)mgr.sets[next_set]->tag;
        uint32_t new_expire = expire;
        for (int j = 0; j < mgr.n_sig; j++) {
            int v = mgr.sig_vars[j];
            rec[kCandHdr + j] = (uint32_t)(mgr.lb[v] + __builtin_ctz(blk[v]));
        }
        for (int u = 0; u < mgr.n_until_cons; u++) {
            int y = mgr.until_y[u];
            bool ex = (expire >> u) & 1u;
            if (!ex && mgr.lb[y] + __builtin_ctz(blk[y]) == 1) {
                ex = true;
                new_expire |= 1u << u;
            }
            rec[kCandHdr + mgr.n_sig + u] = ex;
        }
        rec[3] = new_expire;
        uint32_t *vals = rec.data() + kCandHdr + sig_len, *nb = vals + N;
        for (int v = 0; v < N; v++) vals[v] = (uint32_t)(mgr.lb[v] + __builtin_ctz(blk[v]));
        for (int p = 0; p < K; p++)
            for (int v = 0; v < N; v++) nb[p * N + v] = (p + 1 < K) ? blk[(p + 1) * N + v] : init[v];
        std::vector<uint32_t> kw(KL);
        kw[0] = rec[2];
        for (int j = 0; j < sig_len; j++) kw[1 + j] = rec[kCandHdr + j];
        unsigned long long hh = key_hash(kw.data());
        rec[4] = (uint32_t)hh;
        rec[5] = (uint32_t)(hh >> 32);
        int owner = key_owner(hh, opt.world, KL, kw[0]);  // device_types.hpp: the root key of a signature-less model stays on shard 0
        outbox[owner].insert(outbox[owner].end(), rec.begin(), rec.end());
        return STCSP_OK;
    }

    int commit(const uint32_t *recs, int64_t count) {
        for (int64_t i = 0; i < count; i++) {
            const uint32_t *rec = recs + (size_t)i * CS;
            std::vector<uint32_t> kw(KL);
            kw[0] = rec[2];
            for (int j = 0; j < sig_len; j++) kw[1 + j] = rec[kCandHdr + j];
            auto it = table.find(kw);
            uint32_t idx;
            bool is_new = false;
            if (it == table.end()) {
                idx = (uint32_t)(keys.size() / KL);
                keys.insert(keys.end(), kw.begin(), kw.end());
                table.emplace(kw, idx);
                is_new = true;
            } else {
                idx = it->second;
            }
            e_src.push_back((int64_t)(((uint64_t)rec[1] << 32) | rec[0]));
            e_dst.push_back(((int64_t)opt.rank << STCSP_GID_SHIFT) | idx);
            const uint32_t *vals = rec + kCandHdr + sig_len;
            for (int v = 0; v < N; v++) e_val.push_back((int32_t)vals[v]);
            if (is_new) {
                int set = mgr.find_tag((int32_t)rec[2]);
                if (set < 0) {
                    err = "candidate names an unknown constraint set";
                    return STCSP_E_INTERNAL;
                }
                std::vector<uint32_t> node(NS, 0u);
                uint64_t gid = ((uint64_t)opt.rank << STCSP_GID_SHIFT) | idx;
                node[0] = (uint32_t)gid;
                node[1] = (uint32_t)(gid >> 32);
                node[2] = (uint32_t)set;
                node[3] = rec[3];
                memcpy(node.data() + 4, vals + N, (size_t)NK * 4);
                open.push_back(std::move(node));
            }
        }
        return STCSP_OK;
    }

    int begin() {
        open.clear();
        for (auto &o : outbox) o.clear();
        keys.clear();
        table.clear();
        e_src.clear();
        e_dst.clear();
        e_val.clear();
        ctr = stcsp_counters{};
        truncated = false;
        if (opt.rank == 0) {
            std::vector<uint32_t> key(KL, 0u);
            key[0] = sig_len == 0 ? 0u : kRootTag;
            keys = key;
            table.emplace(key, 0u);
            std::vector<uint32_t> node(NS, 0u);
            for (int p = 0; p < K; p++)
                for (int v = 0; v < N; v++) node[4 + p * N + v] = init[v];
            open.push_back(std::move(node));
        }
        begun = true;
        t0 = std::chrono::steady_clock::now();
        return STCSP_OK;
    }
    bool over_budget() {
        if (opt.max_search_nodes && ctr.search_nodes >= opt.max_search_nodes) return true;
        if (opt.time_limit_s > 0 && (ctr.search_nodes & 255) == 0 &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > opt.time_limit_s)
            return true;
        return false;
    }
    // expand_local returns early once it has expanded budget_rounds * kRoundNodes nodes and holds >= budget_min_open
    // open nodes (the engine's launch rounds have no counterpart here; a "round" stands for kRoundNodes expansions)
    int64_t budget_rounds = 0, budget_min_open = 0;
    static constexpr int64_t kRoundNodes = 8;
    int expand_local(int64_t *left) {
        int64_t done = 0;
        while (!open.empty()) {
            if (over_budget()) {
                truncated = true;
                open.clear();
                break;
            }
            if (budget_rounds > 0 && done >= budget_rounds * kRoundNodes && (int64_t)open.size() >= budget_min_open) break;
            std::vector<uint32_t> node = std::move(open.back());
            open.pop_back();
            int rc = expand(std::move(node));
            if (rc != STCSP_OK) return rc;
            done++;
        }
        if (left) *left = (int64_t)open.size();
        return STCSP_OK;
    }
    // frontier redistribution: the oldest open nodes leave as transfer records (xfer_stride words each)
    std::vector<uint32_t> donated;
    int donate(int64_t want, void **ptr, int64_t *count) {
        const int TS = xfer_stride(N, K);
        donated.clear();
        int64_t n = 0;
        while (n < want && !open.empty()) {
            const std::vector<uint32_t> &node = open.front();
            std::vector<uint32_t> rec(TS, 0u);
            rec[0] = node[0];
            rec[1] = node[1];
            rec[2] = (uint32_t)mgr.sets[node[2] & stcsp::kSetMask]->tag;
            rec[3] = node[3];
            rec[4] = node[2] >> stcsp::kSetBits;
            memcpy(rec.data() + kXferHdr, node.data() + 4, (size_t)NK * 4);
            donated.insert(donated.end(), rec.begin(), rec.end());
            open.pop_front();
            n++;
        }
        *ptr = donated.data();
        *count = n;
        return STCSP_OK;
    }
    int adopt(const uint32_t *recs, int64_t count) {
        const int TS = xfer_stride(N, K);
        for (int64_t i = 0; i < count; i++) {
            const uint32_t *rec = recs + (size_t)i * TS;
            const int set = mgr.find_tag((int32_t)rec[2]);
            if (set < 0) {
                err = "adopted node names an unknown constraint set";
                return STCSP_E_INTERNAL;
            }
            std::vector<uint32_t> node(NS, 0u);
            node[0] = rec[0];
            node[1] = rec[1];
            node[2] = (uint32_t)set;  // (this model revises every item of a node: no dirty seed)
            node[3] = rec[3];
            memcpy(node.data() + 4, rec + kXferHdr, (size_t)NK * 4);
            open.push_back(std::move(node));
        }
        return STCSP_OK;
    }
    int solve() {
        int rc = begin();
        if (rc != STCSP_OK) return rc;
        for (;;) {
            rc = expand_local(nullptr);
            if (rc != STCSP_OK) return rc;
            if (outbox[0].empty()) break;
            std::vector<uint32_t> recs;
            recs.swap(outbox[0]);
            rc = commit(recs.data(), (int64_t)(recs.size() / CS));
            if (rc != STCSP_OK) return rc;
        }
        ctr.seconds_search = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return STCSP_OK;
    }
    int export_result(stcsp_result *res) {
        size_t ns = keys.size() / KL;
        r_cid.assign(ns, 0);
        r_sig.assign(ns * (size_t)sig_len, 0);
        for (size_t i = 0; i < ns; i++) {
            uint32_t tag = keys[i * KL];
            r_cid[i] = tag == kRootTag ? 0 : (int32_t)tag;
            for (int j = 0; j < sig_len; j++) r_sig[i * sig_len + j] = (int32_t)keys[i * KL + 1 + j];
        }
        r_esrc = e_src;
        r_edst = e_dst;
        r_eval = e_val;
        r_fail.assign(ns, 0);
        stcsp_counters c = ctr;
        if (opt.world == 1) {
            std::vector<uint8_t> alive;
            ok_fixpoint((int64_t)ns, r_esrc, r_edst, r_fail, alive);
            size_t w = 0;
            for (size_t k = 0; k < alive.size(); k++)
                if (alive[k]) {
                    r_esrc[w] = r_esrc[k];
                    r_edst[w] = r_edst[k];
                    if (w != k) memmove(&r_eval[w * N], &r_eval[k * N], (size_t)N * 4);
                    w++;
                }
            r_esrc.resize(w);
            r_edst.resize(w);
            r_eval.resize(w * N);
            int64_t ok_states = 0;
            for (size_t v = 1; v < ns; v++) ok_states += !r_fail[v];
            c.dominance = (int64_t)w - ok_states;
        }
        r_issig.assign(mgr.is_sig.begin(), mgr.is_sig.end());
        memset(res, 0, sizeof *res);
        res->n_states = (int64_t)ns;
        res->sig_len = sig_len;
        res->n_sig_vars = mgr.n_sig;
        res->n_until = mgr.n_until;
        res->n_until_cons = mgr.n_until_cons;
        res->state_cid = r_cid.data();
        res->state_sig = r_sig.data();
        res->state_fail = r_fail.data();
        res->n_edges = (int64_t)r_esrc.size();
        res->edge_src = r_esrc.data();
        res->edge_dst = r_edst.data();
        res->edge_values = r_eval.data();
        res->n_vars = N;
        res->n_constraint_sets = (int32_t)mgr.sets.size();
        res->var_is_signature = r_issig.data();
        res->root_final = mgr.n_until_cons == 0;
        res->truncated = truncated;
        res->counters = c;
        return STCSP_OK;
    }
};

thread_local std::string g_err;

}  // namespace

struct stcsp_fmodel {
    Model m;
    std::vector<int32_t> blob;
    std::vector<std::vector<uint32_t>> handed;  // per peer: outbox buffer handed to the driver
};

extern "C" {

int stcsp_fmodel_create(const stcsp_problem *p, const stcsp_options *o, stcsp_fmodel **out) {
    std::unique_ptr<stcsp_fmodel> h(new stcsp_fmodel());
    int rc = h->m.create(p, o);
    if (rc != STCSP_OK) {
        g_err = h->m.err;
        return rc;
    }
    *out = h.release();
    return STCSP_OK;
}
int stcsp_fmodel_solve(stcsp_fmodel *h, stcsp_result *res) {
    if (h->m.opt.world != 1) return STCSP_E_STATE;
    int rc = h->m.solve();
    if (rc != STCSP_OK) return rc;
    return h->m.export_result(res);
}
int stcsp_fmodel_export(stcsp_fmodel *h, stcsp_result *res) { return h->m.export_result(res); }
void stcsp_fmodel_destroy(stcsp_fmodel *h) { delete h; }
const char *stcsp_fmodel_last_error(const stcsp_fmodel *h) { return h ? h->m.err.c_str() : g_err.c_str(); }
int stcsp_fmodel_begin(stcsp_fmodel *h) { return h->m.begin(); }
int stcsp_fmodel_expand_local(stcsp_fmodel *h, int64_t *left) { return h->m.expand_local(left); }
int stcsp_fmodel_candidate_bytes(const stcsp_fmodel *h) { return h->m.CS * 4; }
int stcsp_fmodel_outbox(stcsp_fmodel *h, int peer, void **ptr, int64_t *count) {
    if (peer < 0 || peer >= h->m.opt.world) return STCSP_E_INVALID;
    h->handed.resize(h->m.opt.world);
    h->handed[peer].swap(h->m.outbox[peer]);  // valid until the next outbox(peer) call
    h->m.outbox[peer].clear();
    *ptr = h->handed[peer].data();
    *count = (int64_t)(h->handed[peer].size() / h->m.CS);
    return STCSP_OK;
}
int stcsp_fmodel_commit(stcsp_fmodel *h, const void *recs, int64_t count) { return h->m.commit((const uint32_t *)recs, count); }
int stcsp_fmodel_set_expand_budget(stcsp_fmodel *h, int64_t max_rounds, int64_t min_open) {
    h->m.budget_rounds = max_rounds;
    h->m.budget_min_open = min_open;
    return STCSP_OK;
}
int stcsp_fmodel_node_bytes(const stcsp_fmodel *h) { return xfer_stride(h->m.N, h->m.K) * 4; }
int stcsp_fmodel_donate(stcsp_fmodel *h, int64_t want, void **ptr, int64_t *count) { return h->m.donate(want, ptr, count); }
int stcsp_fmodel_adopt(stcsp_fmodel *h, const void *recs, int64_t count) { return h->m.adopt((const uint32_t *)recs, count); }
int stcsp_fmodel_finish(stcsp_fmodel *h) {
    h->m.ctr.seconds_search = std::chrono::duration<double>(std::chrono::steady_clock::now() - h->m.t0).count();
    return STCSP_OK;
}
int stcsp_fmodel_counters(stcsp_fmodel *h, stcsp_counters *out) {
    *out = h->m.ctr;
    return STCSP_OK;
}
int stcsp_fmodel_sets_blob(stcsp_fmodel *h, const int32_t **words, int64_t *n_words) {
    h->blob.clear();
    h->blob.push_back((int32_t)h->m.mgr.sets.size());
    for (size_t i = 0; i < h->m.mgr.sets.size(); i++) {
        std::vector<int32_t> w = h->m.mgr.serialise_set((int)i);
        h->blob.push_back((int32_t)w.size());
        h->blob.insert(h->blob.end(), w.begin(), w.end());
    }
    *words = h->blob.data();
    *n_words = (int64_t)h->blob.size();
    return STCSP_OK;
}
int stcsp_fmodel_sets_import(stcsp_fmodel *h, const int32_t *words, int64_t n) {
    size_t before = h->m.mgr.sets.size();
    int64_t pos = 1;
    for (int32_t i = 0; i < words[0]; i++) {
        int32_t len = words[pos++];
        int rc = h->m.mgr.import_set(words + pos, (size_t)len);
        if (rc < 0) {
            h->m.err = h->m.mgr.error;
            return rc;
        }
        pos += len;
    }
    if (h->m.mgr.sets.size() != before) return h->m.recompile();
    return STCSP_OK;
}
// Kernel-granularity check: propagate one domain block (N*K words, point-major) under
// constraint set `set` to the GAC fixpoint in place. Returns 1 consistent, 0 wiped out.
int stcsp_fmodel_propagate(stcsp_fmodel *h, int set, uint32_t expire, uint32_t *block) {
    if (set < 0 || set >= (int)h->m.prog.sets.size()) return STCSP_E_INVALID;
    return h->m.propagate(set, expire, block) ? 1 : 0;
}

// Test hook for the product's ahead-of-need translation (cset.cpp SetManager::pretranslate, compiled into this library):
// runs it on this model's set registry and compiles the flat program; reports the sets known afterwards, how many of them
// have a direct (tuple-indexed) transition table and the size of all those tables. Returns the number of transitions added.
int stcsp_fmodel_pretranslate(stcsp_fmodel *h, long long max_tuples, int max_sets, int *n_sets, int *n_direct, long long *table_entries) {
    const int added = h->m.mgr.pretranslate(max_tuples, max_sets);
    if (added < 0) {
        h->m.err = h->m.mgr.error;
        return added;
    }
    const int rc = h->m.recompile();  // (the model interprets the flat program: it must know the new sets)
    if (rc != STCSP_OK) return rc;
    const stcsp::FlatProgram &prog = h->m.prog;
    int direct = 0;
    for (const stcsp::SetDesc &sd : prog.sets) direct += sd.trans_count < 0;
    if (n_sets) *n_sets = (int)h->m.mgr.sets.size();
    if (n_direct) *n_direct = direct;
    if (table_entries) *table_entries = (long long)prog.tdirect.size();
    return added;
}

// Test / sizing hook: byte sizes of the flat program's sections [sets, sweep records, itemrows, wavefront item records, scope,
// strides, code, cons, tables, transition lists + values, direct transition tables], as the engine would lay them out.
int stcsp_fmodel_program_sizes(stcsp_fmodel *h, long long *out, int n) {
    const stcsp::FlatProgram &g = h->m.prog;
    long long witems = 0;
    for (const stcsp::SetDesc &sd : g.sets) witems += sd.nitems - sd.nsmall;
    const long long v[11] = {(long long)(g.sets.size() * sizeof(stcsp::SetDesc)), (long long)(g.items.size() * 16), (long long)(g.itemrows.size() * 4),
                             witems * (long long)sizeof(stcsp::ItemDesc), (long long)(g.scope.size() * 4), (long long)(g.strides.size() * 4),
                             (long long)(g.code.size() * 4), (long long)(g.cons.size() * sizeof(stcsp::ConDesc)), (long long)((g.tables.size() + g.stables.size()) * 4),
                             (long long)(g.trans.size() * sizeof(stcsp::TransDesc) + g.transvals.size() * 4), (long long)(g.tdirect.size() * 4)};
    for (int i = 0; i < n && i < 11; i++) out[i] = v[i];
    return 11;
}

}  // extern "C"
