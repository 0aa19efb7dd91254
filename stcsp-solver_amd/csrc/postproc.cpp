// postproc.cpp -- host post-processing of the raw automaton and the solutions.dot writer.
//
// Behavioural mirror of
//   src/graph.cpp:357-418   graphTraverse      (final / valid flags, drop edges into invalid)
//   src/graph.cpp:304-355   adversarialTraverse  + checkVertexOutEdge  (167-189)
//   src/graph.cpp:247-302   adversarialTraverse2 + checkVertexOutEdge2 (193-244)
//   src/graph.cpp:420-442   renumberVertex
//   src/solveralgorithm.cpp:709-730 solverOut, src/graph.cpp:41-101,145-154 vertexOut/edgeOut
// on flat arrays (CSR) instead of hash_map<int, slist<Edge*>*> per vertex. All passes are
// monotone fixpoints, so the result does not depend on the reference's pointer-ordered
// std::set worklists. O(V+E), once per solve: host plumbing (SURVEY.md section 8f-2 lists the
// device version as a "next" row).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "okfix.hpp"
#include "stcsp_host.h"

namespace stcsp {

struct Automaton {
    int n_vars = 0, sig_len = 0, n_sig_vars = 0, n_until = 0;
    std::vector<std::string> names;
    std::vector<uint8_t> is_sig;
    std::vector<int> var_lb, var_ub;
    int64_t n_states = 0;
    std::vector<int32_t> cid, sig;
    std::vector<uint8_t> fail, valid, final_;
    std::vector<int64_t> id;  // printed vertex id (renumberVertex)
    std::vector<int64_t> esrc, edst;
    std::vector<int32_t> eval;
    std::vector<uint8_t> ealive;
    std::vector<int64_t> out_off, out_edge;  // CSR by src, edges sorted by (dst, insertion)
    std::vector<int64_t> in_off, in_edge;    // CSR by dst

    void build_csr() {
        int64_t E = (int64_t)esrc.size();
        out_off.assign(n_states + 1, 0);
        in_off.assign(n_states + 1, 0);
        for (int64_t e = 0; e < E; e++) {
            out_off[esrc[e] + 1]++;
            in_off[edst[e] + 1]++;
        }
        for (int64_t v = 0; v < n_states; v++) {
            out_off[v + 1] += out_off[v];
            in_off[v + 1] += in_off[v];
        }
        out_edge.resize(E);
        in_edge.resize(E);
        std::vector<int64_t> oc(out_off.begin(), out_off.end() - 1), ic(in_off.begin(), in_off.end() - 1);
        for (int64_t e = 0; e < E; e++) {
            out_edge[oc[esrc[e]]++] = e;
            in_edge[ic[edst[e]]++] = e;
        }
        for (int64_t v = 0; v < n_states; v++)
            std::stable_sort(out_edge.begin() + out_off[v], out_edge.begin() + out_off[v + 1],
                             [&](int64_t a, int64_t b) { return edst[a] < edst[b]; });
    }

    // graphTraverse (graph.cpp:357-418)
    void traverse(int root_final) {
        std::vector<int64_t> work;
        for (int64_t v = 0; v < n_states; v++) {
            if (v != 0) {
                bool fin = true;
                for (int c = n_sig_vars; fin && c < n_sig_vars + n_until; c++) fin = sig[v * sig_len + c] == 1;
                final_[v] = fin;
                valid[v] = fin;
                if (fin) work.push_back(v);
            } else {
                final_[v] = (uint8_t)root_final;
                valid[v] = final_[v];
            }
        }
        while (!work.empty()) {  // backward reachability over the parent map
            int64_t v = work.back();
            work.pop_back();
            for (int64_t i = in_off[v]; i < in_off[v + 1]; i++) {
                int64_t e = in_edge[i];
                if (!ealive[e]) continue;
                int64_t u = esrc[e];
                if (u != v && !valid[u]) {
                    valid[u] = 1;
                    work.push_back(u);
                }
            }
        }
        for (int64_t e = 0; e < (int64_t)esrc.size(); e++)
            if (ealive[e] && (valid[esrc[e]] || esrc[e] == 0) && !valid[edst[e]]) ealive[e] = 0;
    }

    // checkVertexOutEdge (graph.cpp:167-189)
    bool covers_all_values(int64_t v, int i) const {
        for (int c = var_lb[i]; c <= var_ub[i]; c++) {
            bool exist = false;
            for (int64_t k = out_off[v]; k < out_off[v + 1] && !exist; k++) {
                int64_t e = out_edge[k];
                if (ealive[e] && eval[e * n_vars + i] == c && valid[edst[e]]) exist = true;
            }
            if (!exist) return false;
        }
        return true;
    }
    void requeue_valid_parents(int64_t v, std::set<int64_t> &todo) const {
        for (int64_t k = in_off[v]; k < in_off[v + 1]; k++) {
            int64_t e = in_edge[k];
            if (ealive[e] && esrc[e] != v && valid[esrc[e]]) todo.insert(esrc[e]);
        }
    }
    // adversarialTraverse (graph.cpp:304-355)
    int adversarial(int i) {
        if (i < 0 || i >= n_vars) return STCSP_E_INVALID;
        std::set<int64_t> todo;
        for (int64_t v = 0; v < n_states; v++) todo.insert(v);
        while (!todo.empty()) {
            int64_t v = *todo.begin();
            todo.erase(todo.begin());
            if (!covers_all_values(v, i)) {
                valid[v] = 0;
                requeue_valid_parents(v, todo);
            }
        }
        for (int64_t e = 0; e < (int64_t)esrc.size(); e++)
            if (ealive[e] && valid[esrc[e]] && !valid[edst[e]]) ealive[e] = 0;
        return valid[0];
    }
    // checkVertexOutEdge2 (graph.cpp:193-244)
    bool simultaneous_check(int64_t v, int op, int ava) {
        std::map<int, std::set<int>> seen;
        size_t op_nums = (size_t)(var_ub[op] - var_lb[op] + 1);
        bool node_valid = false;
        for (int64_t k = out_off[v]; k < out_off[v + 1]; k++) {
            int64_t e = out_edge[k];
            if (!ealive[e] || !valid[edst[e]]) continue;
            std::set<int> &s = seen[eval[e * n_vars + ava]];
            s.insert(eval[e * n_vars + op]);
            if (s.size() == op_nums) node_valid = true;
        }
        if (node_valid)
            for (int64_t k = out_off[v]; k < out_off[v + 1]; k++) {
                int64_t e = out_edge[k];
                if (ealive[e] && valid[edst[e]] && seen[eval[e * n_vars + ava]].size() != op_nums) ealive[e] = 0;
            }
        return node_valid;
    }
    // adversarialTraverse2 (graph.cpp:247-302)
    int adversarial2(int op, int ava) {
        if (op < 0 || op >= n_vars || ava < 0 || ava >= n_vars) return STCSP_E_INVALID;
        std::vector<uint8_t> was_valid(valid);  // the parent map is built from valid vertices only
        std::set<int64_t> todo;
        for (int64_t v = 0; v < n_states; v++) todo.insert(v);
        while (!todo.empty()) {
            int64_t v = *todo.begin();
            todo.erase(todo.begin());
            if (!simultaneous_check(v, op, ava)) {
                valid[v] = 0;
                for (int64_t k = in_off[v]; k < in_off[v + 1]; k++) {
                    int64_t e = in_edge[k];
                    int64_t u = esrc[e];
                    if (u != v && was_valid[u] && valid[u]) todo.insert(u);
                }
            }
        }
        if (valid[0])
            for (int64_t e = 0; e < (int64_t)esrc.size(); e++)
                if (ealive[e] && valid[esrc[e]] && !valid[edst[e]]) ealive[e] = 0;
        return valid[0];
    }
    // renumberVertex (graph.cpp:420-442): stack-based walk from the root; unreachable
    // vertices keep their table id.
    void renumber() {
        std::vector<uint8_t> seen(n_states, 0);
        std::vector<int64_t> stack{0};
        int64_t next = 0;
        while (!stack.empty()) {
            int64_t v = stack.back();
            stack.pop_back();
            if (seen[v]) continue;
            seen[v] = 1;
            id[v] = next++;
            int64_t last = -1;
            for (int64_t k = out_off[v]; k < out_off[v + 1]; k++) {
                int64_t e = out_edge[k];
                if (!ealive[e] || edst[e] == last) continue;
                last = edst[e];
                stack.push_back(last);
            }
        }
    }

    std::string name_line() const {
        std::string s = "#";
        for (auto &n : names) s += " " + n;
        return s;
    }
    std::string sig_name_line() const {
        std::string s = "#";
        for (int v = 0; v < n_vars; v++)
            if (is_sig[v]) s += " " + names[v];
        return s;
    }
    std::string sig_text(int64_t v, const char *sep) const {
        if (v == 0) return "S";
        std::string s;
        int n = n_sig_vars + n_until;
        for (int i = 0; i < n; i++) {
            s += std::to_string(sig[v * sig_len + i]);
            if (i != n - 1) s += sep;
        }
        return s;
    }

    // solverOut / graphOut / vertexOut / edgeOut
    int write_dot(const char *path) const {
        FILE *fp = fopen(path, "w");
        if (!fp) return STCSP_E_INVALID;
        fprintf(fp, "# Number of nodes = %lld\n", (long long)n_states);
        fprintf(fp, "%s\n%s\n", name_line().c_str(), sig_name_line().c_str());
        fprintf(fp, "digraph \"StCSP\" {\n");
        if (valid[0]) {
            std::vector<uint8_t> seen(n_states, 0);
            // iterative version of the reference's recursive vertexOut
            struct Frame {
                int64_t v, k;
            };
            std::vector<Frame> st;
            auto open = [&](int64_t v) {
                seen[v] = 1;
                fprintf(fp, "%lld [shape=%s, label=\"%d: %s\"];\n", (long long)id[v], final_[v] ? "doublecircle" : "circle",
                        cid[v], sig_text(v, ", ").c_str());
                st.push_back(Frame{v, out_off[v]});
            };
            open(0);
            std::string line;
            while (!st.empty()) {
                Frame &f = st.back();
                while (f.k < out_off[f.v + 1] && !ealive[out_edge[f.k]]) f.k++;
                if (f.k >= out_off[f.v + 1]) {
                    st.pop_back();
                    continue;
                }
                int64_t d = edst[out_edge[f.k]];
                int64_t v = f.v;
                while (f.k < out_off[v + 1] && edst[out_edge[f.k]] == d) {  // one destination bucket
                    int64_t e = out_edge[f.k++];
                    if (!ealive[e]) continue;
                    line.clear();
                    for (int i = 0; i < n_vars; i++) {
                        line += std::to_string(eval[e * n_vars + i]);
                        if (i != n_vars - 1) line += ", ";
                    }
                    fprintf(fp, "%lld -> %lld [label=\"%s\"];\n", (long long)id[v], (long long)id[d], line.c_str());
                }
                if (!seen[d]) open(d);  // invalidates f
            }
        }
        fprintf(fp, "}\n");
        fclose(fp);
        return STCSP_OK;
    }

    // SURVEY.md Appendix A.7 (normative script: tests/canon.py)
    std::string canonical(int64_t *n_live_states, int64_t *n_live_edges) const {
        std::string out = name_line() + "\n" + sig_name_line() + "\n";
        int64_t ns = 0, ne = 0;
        if (!valid[0]) {
            out += "EMPTY\n";
        } else {
            std::vector<int64_t> num(n_states, -1), order;
            std::map<int, int> cidmap;
            auto sorted_out = [&](int64_t u) {
                std::vector<int64_t> es;
                for (int64_t k = out_off[u]; k < out_off[u + 1]; k++)
                    if (ealive[out_edge[k]]) es.push_back(out_edge[k]);
                std::sort(es.begin(), es.end(), [&](int64_t a, int64_t b) {
                    int c = 0;
                    for (int i = 0; i < n_vars && c == 0; i++) {
                        int x = eval[a * n_vars + i], y = eval[b * n_vars + i];
                        c = (x < y) ? -1 : (x > y);
                    }
                    if (c) return c < 0;
                    return a < b;
                });
                return es;
            };
            num[0] = 0;
            order.push_back(0);
            for (size_t q = 0; q < order.size(); q++) {
                int64_t u = order[q];
                for (int64_t e : sorted_out(u)) {
                    int64_t v = edst[e];
                    if (num[v] < 0) {
                        num[v] = (int64_t)order.size();
                        order.push_back(v);
                    }
                }
            }
            for (int64_t u : order) {
                if (!cidmap.count(cid[u])) {
                    int k = (int)cidmap.size();
                    cidmap[cid[u]] = k;
                }
                out += "S " + std::to_string(num[u]) + " " + std::to_string(cidmap[cid[u]]) + " " +
                       std::to_string((int)final_[u]) + " " + sig_text(u, " ") + "\n";
                for (int64_t e : sorted_out(u)) {
                    out += "E " + std::to_string(num[u]) + " " + std::to_string(num[edst[e]]);
                    for (int i = 0; i < n_vars; i++) out += " " + std::to_string(eval[e * n_vars + i]);
                    out += "\n";
                    ne++;
                }
            }
            ns = (int64_t)order.size();
        }
        if (n_live_states) *n_live_states = ns;
        if (n_live_edges) *n_live_edges = ne;
        return out;
    }
};

struct Merged {
    std::vector<int32_t> cid, sig, eval;
    std::vector<uint8_t> fail, issig;
    std::vector<int64_t> esrc, edst;
    stcsp_result res{};
};

}  // namespace stcsp

using stcsp::Automaton;

struct stcsp_automaton {
    Automaton a;
    int root_final;
};
struct stcsp_merged {
    stcsp::Merged m;
};

extern "C" {

int stcsp_automaton_build(const stcsp_problem *p, const stcsp_result *r, stcsp_automaton **out) {
    if (!p || !r || !out) return STCSP_E_INVALID;
    stcsp_automaton *h = new stcsp_automaton();
    Automaton &a = h->a;
    a.n_vars = r->n_vars;
    a.sig_len = r->sig_len;
    a.n_sig_vars = r->n_sig_vars;
    a.n_until = r->n_until;
    for (int v = 0; v < a.n_vars; v++) {
        a.names.push_back(p->var_names && p->var_names[v] ? p->var_names[v] : ("v" + std::to_string(v)));
        a.var_lb.push_back(p->var_lb[v]);
        a.var_ub.push_back(p->var_ub[v]);
    }
    a.is_sig.assign(r->var_is_signature, r->var_is_signature + a.n_vars);
    a.n_states = r->n_states;
    a.cid.assign(r->state_cid, r->state_cid + r->n_states);
    a.sig.assign(r->state_sig, r->state_sig + r->n_states * (int64_t)r->sig_len);
    a.fail.assign(r->state_fail, r->state_fail + r->n_states);
    a.valid.assign(r->n_states, 0);
    a.final_.assign(r->n_states, 0);
    a.id.resize(r->n_states);
    for (int64_t v = 0; v < r->n_states; v++) a.id[v] = v;
    a.esrc.assign(r->edge_src, r->edge_src + r->n_edges);
    a.edst.assign(r->edge_dst, r->edge_dst + r->n_edges);
    a.eval.assign(r->edge_values, r->edge_values + r->n_edges * (int64_t)r->n_vars);
    a.ealive.assign(r->n_edges, 1);
    for (int64_t e = 0; e < r->n_edges; e++)  // tolerate raw logs (STCSP_F_KEEP_RAW_EDGES)
        if (a.fail[a.edst[e]]) a.ealive[e] = 0;
    a.build_csr();
    h->root_final = r->root_final;
    *out = h;
    return STCSP_OK;
}

void stcsp_automaton_free(stcsp_automaton *a) { delete a; }

int stcsp_automaton_traverse(stcsp_automaton *a) {
    if (!a) return STCSP_E_INVALID;
    a->a.traverse(a->root_final);
    return STCSP_OK;
}
int stcsp_automaton_adversarial(stcsp_automaton *a, int var_index) { return a ? a->a.adversarial(var_index) : STCSP_E_INVALID; }
int stcsp_automaton_adversarial2(stcsp_automaton *a, int op, int ava) { return a ? a->a.adversarial2(op, ava) : STCSP_E_INVALID; }
int stcsp_automaton_import_flags(stcsp_automaton *a, const uint8_t *valid, const uint8_t *final_flags, const uint8_t *alive) {
    if (!a || !valid || !final_flags || (!alive && !a->a.esrc.empty())) return STCSP_E_INVALID;
    Automaton &g = a->a;
    g.valid.assign(valid, valid + g.n_states);
    g.final_.assign(final_flags, final_flags + g.n_states);
    for (size_t e = 0; e < g.esrc.size(); e++) g.ealive[e] = g.ealive[e] && alive[e];
    return STCSP_OK;
}
int stcsp_automaton_flags(const stcsp_automaton *a, uint8_t *valid, uint8_t *final_flags, uint8_t *alive) {
    if (!a) return STCSP_E_INVALID;
    const Automaton &g = a->a;
    if (valid) memcpy(valid, g.valid.data(), g.valid.size());
    if (final_flags) memcpy(final_flags, g.final_.data(), g.final_.size());
    if (alive) memcpy(alive, g.ealive.data(), g.ealive.size());
    return STCSP_OK;
}
int stcsp_automaton_renumber(stcsp_automaton *a) {
    if (!a) return STCSP_E_INVALID;
    a->a.renumber();
    return STCSP_OK;
}
int stcsp_automaton_write_dot(const stcsp_automaton *a, const char *path) { return a ? a->a.write_dot(path) : STCSP_E_INVALID; }

char *stcsp_automaton_canonical(const stcsp_automaton *a, size_t *len) {
    if (!a) return nullptr;
    std::string s = a->a.canonical(nullptr, nullptr);
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.c_str(), s.size() + 1);
    if (len) *len = s.size();
    return p;
}
int64_t stcsp_automaton_num_states(const stcsp_automaton *a) { return a->a.n_states; }
int64_t stcsp_automaton_num_live_states(const stcsp_automaton *a) {
    int64_t ns = 0;
    a->a.canonical(&ns, nullptr);
    return ns;
}
int64_t stcsp_automaton_num_live_edges(const stcsp_automaton *a) {
    int64_t ne = 0;
    a->a.canonical(nullptr, &ne);
    return ne;
}

// Sharded runs: shard r's result holds its local states (index = local id) and the raw edge
// log of the leaves it committed, with global ids (owner << STCSP_GID_SHIFT | local) in
// edge_src / edge_dst. The root is local state 0 of shard 0.
int stcsp_merge_shards(const stcsp_result *const *shards, int n, stcsp_merged **out) {
    if (!shards || n <= 0 || !out) return STCSP_E_INVALID;
    stcsp_merged *h = new stcsp_merged();
    stcsp::Merged &m = h->m;
    std::vector<int64_t> off(n + 1, 0);
    for (int r = 0; r < n; r++) off[r + 1] = off[r] + shards[r]->n_states;
    const stcsp_result *s0 = shards[0];
    int sl = s0->sig_len, nv = s0->n_vars;
    stcsp_counters ctr{};
    int truncated = 0, nsets = 0;
    for (int r = 0; r < n; r++) {
        const stcsp_result *s = shards[r];
        m.cid.insert(m.cid.end(), s->state_cid, s->state_cid + s->n_states);
        m.sig.insert(m.sig.end(), s->state_sig, s->state_sig + s->n_states * (int64_t)sl);
        for (int64_t e = 0; e < s->n_edges; e++) {
            int64_t gs = s->edge_src[e], gd = s->edge_dst[e];
            m.esrc.push_back(off[gs >> STCSP_GID_SHIFT] + (gs & (((int64_t)1 << STCSP_GID_SHIFT) - 1)));
            m.edst.push_back(off[gd >> STCSP_GID_SHIFT] + (gd & (((int64_t)1 << STCSP_GID_SHIFT) - 1)));
        }
        m.eval.insert(m.eval.end(), s->edge_values, s->edge_values + s->n_edges * (int64_t)nv);
        ctr.search_nodes += s->counters.search_nodes;
        ctr.gac_calls += s->counters.gac_calls;
        ctr.fails += s->counters.fails;
        ctr.leaves += s->counters.leaves;
        ctr.revisions += s->counters.revisions;
        ctr.evaluations += s->counters.evaluations;
        if (s->counters.levels > ctr.levels) ctr.levels = s->counters.levels;
        if (s->counters.seconds_search > ctr.seconds_search) ctr.seconds_search = s->counters.seconds_search;
        if (s->counters.seconds_export > ctr.seconds_export) ctr.seconds_export = s->counters.seconds_export;
        truncated |= s->truncated;
        if (s->n_constraint_sets > nsets) nsets = s->n_constraint_sets;
    }
    std::vector<uint8_t> alive;
    stcsp::ok_fixpoint(off[n], m.esrc, m.edst, m.fail, alive);
    size_t w = 0;
    for (size_t e = 0; e < alive.size(); e++)
        if (alive[e]) {
            m.esrc[w] = m.esrc[e];
            m.edst[w] = m.edst[e];
            std::copy(m.eval.begin() + e * nv, m.eval.begin() + (e + 1) * nv, m.eval.begin() + w * nv);
            w++;
        }
    m.esrc.resize(w);
    m.edst.resize(w);
    m.eval.resize(w * nv);
    int64_t ok_states = 0;
    for (int64_t v = 1; v < off[n]; v++) ok_states += !m.fail[v];
    ctr.dominance = (int64_t)w - ok_states;
    m.issig.assign(s0->var_is_signature, s0->var_is_signature + nv);
    m.res = *s0;
    m.res.n_states = off[n];
    m.res.state_cid = m.cid.data();
    m.res.state_sig = m.sig.data();
    m.res.state_fail = m.fail.data();
    m.res.n_edges = (int64_t)w;
    m.res.edge_src = m.esrc.data();
    m.res.edge_dst = m.edst.data();
    m.res.edge_values = m.eval.data();
    m.res.var_is_signature = m.issig.data();
    m.res.n_constraint_sets = nsets;
    m.res.truncated = truncated;
    m.res.counters = ctr;
    *out = h;
    return STCSP_OK;
}
const stcsp_result *stcsp_merged_result(const stcsp_merged *m) { return m ? &m->m.res : nullptr; }
void stcsp_merged_free(stcsp_merged *m) { delete m; }

}  // extern "C"
