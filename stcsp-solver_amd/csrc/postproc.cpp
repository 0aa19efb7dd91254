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
#include <new>
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

// Append-only text buffer with hand-rolled integer formatting: the reference prints one fprintf
// per edge (src/graph.cpp:92-101; 82 MB of text on partialorder_14), which dominates the wall
// time once the search itself takes milliseconds.
struct TextOut {
    std::vector<char> buf;
    size_t n = 0;
    FILE *fp = nullptr;  // nullptr: keep everything in memory
    explicit TextOut(FILE *f = nullptr, size_t cap = 1u << 22) : buf(cap), fp(f) {}
    void room(size_t need) {
        if (n + need <= buf.size()) return;
        if (fp) {
            fwrite(buf.data(), 1, n, fp);
            n = 0;
            if (need > buf.size()) buf.resize(need * 2);
        } else {
            buf.resize(std::max(buf.size() * 2, n + need));
        }
    }
    void str(const char *s, size_t len) {
        room(len);
        memcpy(buf.data() + n, s, len);
        n += len;
    }
    void str(const std::string &s) { str(s.data(), s.size()); }
    void lit(const char *s) { str(s, strlen(s)); }
    void ch(char c) {
        room(1);
        buf[n++] = c;
    }
    void num(long long v) {
        room(24);
        char tmp[24];
        int k = 0;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do {
            tmp[k++] = (char)('0' + u % 10);
            u /= 10;
        } while (u);
        if (v < 0) buf[n++] = '-';
        while (k) buf[n++] = tmp[--k];
    }
    void flush() {
        if (fp && n) {
            fwrite(buf.data(), 1, n, fp);
            n = 0;
        }
    }
    std::string take() { return std::string(buf.data(), n); }
};

struct Automaton {
    int n_vars = 0, sig_len = 0, n_sig_vars = 0, n_until = 0;
    std::vector<std::string> names;
    std::vector<uint8_t> is_sig;
    std::vector<int> var_lb, var_ub;
    int64_t n_states = 0;
    int64_t table_states = -1;  // "# Number of nodes" when it differs from n_states (binary reload)
    std::vector<int32_t> cid, sig;
    std::vector<uint8_t> fail, valid, final_;
    std::vector<int64_t> id;  // printed vertex id (renumberVertex)
    std::vector<int32_t> cid_print;  // order_by_label(): constraint-set ids renumbered by first visit of renumber()
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

    // State indices and the order of the edge arrays depend on the scheduling of the search, so the
    // (dst, insertion) order above differs from run to run. For output files the out-edges of every
    // state are re-ordered by content instead: destination buckets by their smallest label, edges
    // inside a bucket by label (a label determines its destination, so this is a total order) --
    // the same input then always gives the same solutions.dot / binary file.
    void order_by_label() {
        auto less = [&](int64_t a, int64_t b) {
            const int32_t *x = &eval[a * n_vars], *y = &eval[b * n_vars];
            for (int i = 0; i < n_vars; i++)
                if (x[i] != y[i]) return x[i] < y[i];
            return a < b;
        };
        std::vector<int64_t> rank(n_states, -1), touched;
        for (int64_t v = 0; v < n_states; v++) {
            auto b = out_edge.begin() + out_off[v], e = out_edge.begin() + out_off[v + 1];
            if (e - b < 2) continue;
            std::sort(b, e, less);
            int64_t next = 0;
            touched.clear();
            for (auto it = b; it != e; ++it)
                if (rank[edst[*it]] < 0) {
                    rank[edst[*it]] = next++;
                    touched.push_back(edst[*it]);
                }
            std::stable_sort(b, e, [&](int64_t x, int64_t y) { return rank[edst[x]] < rank[edst[y]]; });
            for (int64_t d : touched) rank[d] = -1;
        }
        label_ordered = true;
    }
    bool label_ordered = false;
    int32_t printed_cid(int64_t v) const {
        return (label_ordered && cid[v] >= 0 && (size_t)cid[v] < cid_print.size() && cid_print[cid[v]] >= 0) ? cid_print[cid[v]] : cid[v];
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
        int32_t next_cid = 0;
        cid_print.clear();
        while (!stack.empty()) {
            int64_t v = stack.back();
            stack.pop_back();
            if (seen[v]) continue;
            seen[v] = 1;
            id[v] = next++;
            if (label_ordered && cid[v] >= 0) {  // set ids in the order this walk meets them (the search's
                if ((size_t)cid[v] >= cid_print.size()) cid_print.resize(cid[v] + 1, -1);  // discovery order is scheduling-dependent)
                if (cid_print[cid[v]] < 0) cid_print[cid[v]] = next_cid++;
            }
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

    // label text of one edge ("v0, v1, ..., vN-1"): per variable the text of every domain value
    // is formatted once, then an edge label is a run of short copies
    struct LabelTable {
        // Consecutive variables are grouped while the product of their domain sizes stays small
        // and the joint text fits 16 bytes; a group's text is one table lookup + one 16-byte copy
        // (partialorder_14: 31 values per label -> 8 copies).
        struct Piece {
            char text[16];
            uint32_t len;
        };
        struct Group {
            int first, count;  // variables [first, first + count)
            int base;          // index of the group's first piece
            int combos;
        };
        std::vector<Piece> piece;
        std::vector<Group> groups;
        const Automaton &a;
        const char *sep;
        LabelTable(const Automaton &au, const char *sep_) : a(au), sep(sep_) {
            const size_t seplen = strlen(sep);
            auto width = [&](int i) { return (long long)a.var_ub[i] - a.var_lb[i] + 1; };
            auto maxlen = [&](int i) {
                return std::max(std::to_string(a.var_lb[i]).size(), std::to_string(a.var_ub[i]).size()) + (i != a.n_vars - 1 ? seplen : 0);
            };
            for (int i = 0; i < a.n_vars;) {
                Group g{i, 0, (int)piece.size(), 1};
                size_t len = 0;
                while (i < a.n_vars && width(i) <= 4096 && (long long)g.combos * width(i) <= 4096 && len + maxlen(i) <= sizeof(Piece::text)) {
                    g.combos *= (int)width(i);
                    len += maxlen(i);
                    g.count++;
                    i++;
                }
                if (g.count == 0) {  // a variable that does not fit a table: formatted directly
                    g.count = -1;
                    i++;
                    groups.push_back(g);
                    continue;
                }
                for (int c = 0; c < g.combos; c++) {  // mixed radix, first variable most significant
                    std::string t;
                    int rem = c, div = g.combos;
                    for (int k = 0; k < g.count; k++) {
                        const int v = g.first + k;
                        div /= (int)width(v);
                        t += std::to_string(a.var_lb[v] + rem / div);
                        rem %= div;
                        if (v != a.n_vars - 1) t += sep;
                    }
                    Piece pc{};
                    memcpy(pc.text, t.data(), t.size());
                    pc.len = (uint32_t)t.size();
                    piece.push_back(pc);
                }
                groups.push_back(g);
            }
        }
        void put(TextOut &o, int64_t e) const {
            const int32_t *row = &a.eval[e * a.n_vars];
            o.room((size_t)a.n_vars * 16 + 32);
            char *w = o.buf.data() + o.n;
            for (const Group &g : groups) {
                bool direct = g.count < 0;
                int idx = 0;
                if (!direct)
                    for (int k = 0; k < g.count; k++) {
                        const int v = g.first + k;
                        const int d = row[v] - a.var_lb[v], wd = a.var_ub[v] - a.var_lb[v] + 1;
                        direct |= (d < 0) | (d >= wd);  // value outside the declared bounds
                        idx = idx * wd + d;
                    }
                if (!direct) {
                    const Piece &pc = piece[g.base + idx];
                    memcpy(w, pc.text, sizeof pc.text);
                    w += pc.len;
                    continue;
                }
                o.n = (size_t)(w - o.buf.data());
                const int cnt = g.count < 0 ? 1 : g.count;
                for (int k = 0; k < cnt; k++) {
                    o.num(row[g.first + k]);
                    if (g.first + k != a.n_vars - 1) o.lit(sep);
                }
                o.room((size_t)a.n_vars * 16 + 32);
                w = o.buf.data() + o.n;
            }
            o.n = (size_t)(w - o.buf.data());
        }
    };

    // solverOut / graphOut / vertexOut / edgeOut
    int write_dot(const char *path) const {
        FILE *fp = fopen(path, "w");
        if (!fp) return STCSP_E_INVALID;
        TextOut o(fp);
        o.lit("# Number of nodes = ");
        o.num(table_states >= 0 ? table_states : n_states);
        o.ch('\n');
        o.str(name_line());
        o.ch('\n');
        o.str(sig_name_line());
        o.lit("\ndigraph \"StCSP\" {\n");
        if (valid[0]) {
            LabelTable labels(*this, ", ");
            std::vector<uint8_t> seen(n_states, 0);
            // iterative version of the reference's recursive vertexOut
            struct Frame {
                int64_t v, k;
            };
            std::vector<Frame> st;
            auto open = [&](int64_t v) {
                seen[v] = 1;
                o.num(id[v]);
                o.lit(final_[v] ? " [shape=doublecircle, label=\"" : " [shape=circle, label=\"");
                o.num(printed_cid(v));
                o.lit(": ");
                o.str(sig_text(v, ", "));
                o.lit("\"];\n");
                st.push_back(Frame{v, out_off[v]});
            };
            open(0);
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
                    if (f.k + 6 < out_off[v + 1]) {  // label rows are visited in CSR order, i.e. at random
                        const char *nx = (const char *)&eval[out_edge[f.k + 6] * n_vars];
                        __builtin_prefetch(nx);
                        __builtin_prefetch(nx + 64);
                    }
                    if (!ealive[e]) continue;
                    o.num(id[v]);
                    o.lit(" -> ");
                    o.num(id[d]);
                    o.lit(" [label=\"");
                    labels.put(o, e);
                    o.lit("\"];\n");
                }
                if (!seen[d]) open(d);  // invalidates f
            }
        }
        o.lit("}\n");
        o.flush();
        const bool bad = ferror(fp) != 0;
        return (fclose(fp) != 0 || bad) ? STCSP_E_INVALID : STCSP_OK;
    }

    // ---- compact binary form of the printed automaton ----------
    // Little-endian. Holds what solutions.dot holds -- the states reachable from a valid root over
    // kept edges, with their printed ids -- at ~(4 + n_vars) bytes per edge instead of ~120 of text.
    //   char  magic[8] = "STCSPAUT"; u32 version = 1; u32 n_vars, sig_len, n_sig_vars, n_until;
    //   u32 value_bytes (1: u8 offset from the variable's lb, 4: i32); u64 table_states; u64 n_states; u64 n_edges;
    //   per variable: i32 lb, i32 ub, u8 is_signature, u16 name_len, name
    //   u32 id[n_states]; i32 cid[n_states]; u8 final[n_states]; i32 sig[n_states*sig_len]; u32 out_degree[n_states];
    //   u32 dst[n_edges] (index into the state arrays, grouped by source in state order); values[n_edges*n_vars]
    static constexpr const char *kMagic = "STCSPAUT";
    int write_binary(const char *path) const {
        FILE *fp = fopen(path, "wb");
        if (!fp) return STCSP_E_INVALID;
        std::vector<int64_t> order;  // kept states in discovery order (index 0 = root)
        std::vector<int64_t> dense(n_states, -1);
        if (valid[0]) {
            dense[0] = 0;
            order.push_back(0);
            for (size_t q = 0; q < order.size(); q++)
                for (int64_t k = out_off[order[q]]; k < out_off[order[q] + 1]; k++) {
                    const int64_t e = out_edge[k];
                    if (ealive[e] && dense[edst[e]] < 0) {
                        dense[edst[e]] = (int64_t)order.size();
                        order.push_back(edst[e]);
                    }
                }
        }
        bool narrow = true;
        for (int i = 0; i < n_vars; i++) narrow = narrow && (long long)var_ub[i] - var_lb[i] < 256;
        std::vector<uint32_t> ids, outdeg, dsts;
        std::vector<int32_t> cids, sigs;
        std::vector<uint8_t> fins, vals8;
        std::vector<int32_t> vals32;
        for (int64_t v : order) {
            ids.push_back((uint32_t)id[v]);
            cids.push_back(printed_cid(v));
            fins.push_back(final_[v]);
            for (int c = 0; c < sig_len; c++) sigs.push_back(sig[v * sig_len + c]);
            uint32_t deg = 0;
            for (int64_t k = out_off[v]; k < out_off[v + 1]; k++) {
                const int64_t e = out_edge[k];
                if (!ealive[e]) continue;
                deg++;
                dsts.push_back((uint32_t)dense[edst[e]]);
                for (int i = 0; i < n_vars; i++) {
                    const int32_t x = eval[e * n_vars + i];
                    if (narrow && (x < var_lb[i] || x > var_ub[i])) {  // cannot happen for engine results
                        fclose(fp);
                        return STCSP_E_INVALID;
                    }
                    if (narrow) vals8.push_back((uint8_t)(x - var_lb[i]));
                    else vals32.push_back(x);
                }
            }
            outdeg.push_back(deg);
        }
        auto put = [&](const void *ptr, size_t bytes) { if (bytes) fwrite(ptr, 1, bytes, fp); };
        auto u32 = [&](uint32_t x) { put(&x, 4); };
        auto u64 = [&](uint64_t x) { put(&x, 8); };
        put(kMagic, 8);
        u32(1);
        u32((uint32_t)n_vars);
        u32((uint32_t)sig_len);
        u32((uint32_t)n_sig_vars);
        u32((uint32_t)n_until);
        u32(narrow ? 1 : 4);
        u64((uint64_t)n_states);
        u64(order.size());
        u64(dsts.size());
        for (int i = 0; i < n_vars; i++) {
            int32_t b[2] = {var_lb[i], var_ub[i]};
            put(b, 8);
            put(&is_sig[i], 1);
            uint16_t nl = (uint16_t)names[i].size();
            put(&nl, 2);
            put(names[i].data(), nl);
        }
        put(ids.data(), ids.size() * 4);
        put(cids.data(), cids.size() * 4);
        put(fins.data(), fins.size());
        put(sigs.data(), sigs.size() * 4);
        put(outdeg.data(), outdeg.size() * 4);
        put(dsts.data(), dsts.size() * 4);
        if (narrow) put(vals8.data(), vals8.size());
        else put(vals32.data(), vals32.size() * 4);
        const bool bad = ferror(fp) != 0;
        return (fclose(fp) != 0 || bad) ? STCSP_E_INVALID : STCSP_OK;
    }
    int read_binary(const char *path) {
        FILE *fp = fopen(path, "rb");
        if (!fp) return STCSP_E_INVALID;
        bool ok = true;
        auto get = [&](void *ptr, size_t bytes) { if (bytes && fread(ptr, 1, bytes, fp) != bytes) ok = false; };
        char magic[8] = {0};
        uint32_t h[6] = {0};
        uint64_t table = 0, ns = 0, ne = 0;
        get(magic, 8);
        get(h, sizeof h);
        get(&table, 8);
        get(&ns, 8);
        get(&ne, 8);
        if (!ok || memcmp(magic, kMagic, 8) != 0 || h[0] != 1 || (h[5] != 1 && h[5] != 4) || h[1] > (1u << 20) || h[2] > (1u << 20) ||
            ns > (1ull << 32) || ne > (1ull << 36)) {
            fclose(fp);
            return STCSP_E_INVALID;
        }
        // the counts must be covered by what is left of the file BEFORE anything is sized from them (a
        // truncated or hostile header would otherwise ask for hundreds of GB)
        {
            const long here = ftell(fp);
            long size = -1;
            if (here >= 0 && fseek(fp, 0, SEEK_END) == 0) size = ftell(fp);
            if (here < 0 || size < here || fseek(fp, here, SEEK_SET) != 0) {
                fclose(fp);
                return STCSP_E_INVALID;
            }
            const unsigned __int128 left = (unsigned __int128)(size - here);
            const unsigned __int128 need = (unsigned __int128)h[1] * 11                     // per variable: bounds, flag, name length
                                           + (unsigned __int128)ns * (4 + 4 + 1 + 4)        // ids, cids, final, outdeg
                                           + (unsigned __int128)ns * h[2] * 4               // signatures
                                           + (unsigned __int128)ne * 4                      // dsts
                                           + (unsigned __int128)ne * h[1] * (h[5] == 1 ? 1 : 4);  // labels
            if (need > left) {
                fclose(fp);
                return STCSP_E_INVALID;
            }
        }
        n_vars = (int)h[1];
        sig_len = (int)h[2];
        n_sig_vars = (int)h[3];
        n_until = (int)h[4];
        names.clear();
        var_lb.clear();
        var_ub.clear();
        is_sig.clear();
        for (int i = 0; i < n_vars && ok; i++) {
            int32_t b[2] = {0, 0};
            uint8_t sg = 0;
            uint16_t nl = 0;
            get(b, 8);
            get(&sg, 1);
            get(&nl, 2);
            std::string nm(nl, ' ');
            get(&nm[0], nl);
            var_lb.push_back(b[0]);
            var_ub.push_back(b[1]);
            is_sig.push_back(sg);
            names.push_back(nm);
        }
        table_states = (int64_t)table;
        n_states = (int64_t)std::max<uint64_t>(ns, 1);  // an EMPTY automaton keeps an invalid root
        std::vector<uint32_t> ids(ns), outdeg(ns), dsts(ne);
        cid.assign(n_states, 0);
        final_.assign(n_states, 0);
        sig.assign((size_t)n_states * sig_len, 0);
        get(ids.data(), ns * 4);
        get(cid.data(), ns * 4);
        get(final_.data(), ns);
        get(sig.data(), ns * sig_len * 4);
        get(outdeg.data(), ns * 4);
        get(dsts.data(), ne * 4);
        eval.assign((size_t)ne * n_vars, 0);
        if (h[5] == 1) {
            std::vector<uint8_t> v8((size_t)ne * n_vars);
            get(v8.data(), v8.size());
            for (size_t k = 0; k < v8.size(); k++) eval[k] = var_lb[k % n_vars] + v8[k];
        } else {
            get(eval.data(), eval.size() * 4);
        }
        fclose(fp);
        uint64_t sum = 0;
        for (uint32_t d : outdeg) sum += d;
        for (uint32_t d : dsts) ok = ok && d < ns;
        if (!ok || sum != ne) return STCSP_E_INVALID;
        id.assign(n_states, 0);
        for (uint64_t v = 0; v < ns; v++) id[v] = ids[v];
        fail.assign(n_states, 0);
        valid.assign(n_states, ns ? 1 : 0);
        esrc.clear();
        edst.assign(dsts.begin(), dsts.end());
        for (uint64_t v = 0; v < ns; v++) esrc.insert(esrc.end(), outdeg[v], (int64_t)v);
        ealive.assign(ne, 1);
        build_csr();
        return STCSP_OK;
    }

    // SURVEY.md Appendix A.7 (normative script: tests/canon.py)
    std::string canonical(int64_t *n_live_states, int64_t *n_live_edges, bool want_text = true) const {
        TextOut out(nullptr, want_text ? (1u << 22) : 4096);
        out.str(name_line());
        out.ch('\n');
        out.str(sig_name_line());
        out.ch('\n');
        int64_t ns = 0, ne = 0;
        if (!valid[0]) {
            out.lit("EMPTY\n");
        } else {
            std::vector<int64_t> num(n_states, -1), order;
            std::map<int, int> cidmap;
            std::vector<int64_t> sorted_edges, sorted_off{0};  // per state in `order`: live out-edges sorted by label
            auto sort_out = [&](int64_t u) {
                const size_t b = sorted_edges.size();
                for (int64_t k = out_off[u]; k < out_off[u + 1]; k++)
                    if (ealive[out_edge[k]]) sorted_edges.push_back(out_edge[k]);
                std::sort(sorted_edges.begin() + b, sorted_edges.end(), [&](int64_t a, int64_t b2) {
                    const int32_t *x = &eval[a * n_vars], *y = &eval[b2 * n_vars];
                    for (int i = 0; i < n_vars; i++)
                        if (x[i] != y[i]) return x[i] < y[i];
                    return a < b2;
                });
                sorted_off.push_back((int64_t)sorted_edges.size());
            };
            num[0] = 0;
            order.push_back(0);
            for (size_t q = 0; q < order.size(); q++) {
                sort_out(order[q]);
                for (int64_t k = sorted_off[q]; k < sorted_off[q + 1]; k++) {
                    int64_t v = edst[sorted_edges[k]];
                    if (num[v] < 0) {
                        num[v] = (int64_t)order.size();
                        order.push_back(v);
                    }
                }
            }
            ns = (int64_t)order.size();
            ne = (int64_t)sorted_edges.size();
            if (want_text) {
                LabelTable labels(*this, " ");
                for (size_t q = 0; q < order.size(); q++) {
                    const int64_t u = order[q];
                    if (!cidmap.count(cid[u])) {
                        int k = (int)cidmap.size();
                        cidmap[cid[u]] = k;
                    }
                    out.lit("S ");
                    out.num(num[u]);
                    out.ch(' ');
                    out.num(cidmap[cid[u]]);
                    out.ch(' ');
                    out.num((int)final_[u]);
                    out.ch(' ');
                    out.str(sig_text(u, " "));
                    out.ch('\n');
                    for (int64_t k = sorted_off[q]; k < sorted_off[q + 1]; k++) {
                        const int64_t e = sorted_edges[k];
                        out.lit("E ");
                        out.num(num[u]);
                        out.ch(' ');
                        out.num(num[edst[e]]);
                        out.ch(' ');
                        labels.put(out, e);
                        out.ch('\n');
                    }
                }
            }
        }
        if (n_live_states) *n_live_states = ns;
        if (n_live_edges) *n_live_edges = ne;
        return out.take();
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
int stcsp_automaton_order_by_label(stcsp_automaton *a) {
    if (!a) return STCSP_E_INVALID;
    a->a.order_by_label();
    return STCSP_OK;
}
int stcsp_automaton_renumber(stcsp_automaton *a) {
    if (!a) return STCSP_E_INVALID;
    a->a.renumber();
    return STCSP_OK;
}
int stcsp_automaton_write_dot(const stcsp_automaton *a, const char *path) { return a ? a->a.write_dot(path) : STCSP_E_INVALID; }
int stcsp_automaton_write_binary(const stcsp_automaton *a, const char *path) { return a && path ? a->a.write_binary(path) : STCSP_E_INVALID; }
int stcsp_automaton_read_binary(const char *path, stcsp_automaton **out) {
    if (!path || !out) return STCSP_E_INVALID;
    stcsp_automaton *h = nullptr;
    try {  // no exception may cross the C boundary
        h = new stcsp_automaton();
        h->root_final = 0;
        int rc = h->a.read_binary(path);
        if (rc != STCSP_OK) {
            delete h;
            return rc;
        }
    } catch (const std::bad_alloc &) {
        delete h;
        return STCSP_E_NOMEM;
    } catch (...) {
        delete h;
        return STCSP_E_INVALID;
    }
    *out = h;
    return STCSP_OK;
}

char *stcsp_automaton_canonical(const stcsp_automaton *a, size_t *len) {
    if (!a) return nullptr;
    std::string s = a->a.canonical(nullptr, nullptr);
    char *p = (char *)malloc(s.size() + 1);
    memcpy(p, s.c_str(), s.size() + 1);
    if (len) *len = s.size();
    return p;
}
int64_t stcsp_automaton_num_states(const stcsp_automaton *a) { return a->a.table_states >= 0 ? a->a.table_states : a->a.n_states; }
int64_t stcsp_automaton_num_live_states(const stcsp_automaton *a) {
    int64_t ns = 0;
    a->a.canonical(&ns, nullptr, false);
    return ns;
}
int64_t stcsp_automaton_num_live_edges(const stcsp_automaton *a) {
    int64_t ne = 0;
    a->a.canonical(nullptr, &ne, false);
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
    {
        // Solver::seenConstraints holds the initial set and every set some leaf translated to, i.e. the distinct
        // set ids of the table's states (a shard's own registry may hold more: sets it translated ahead of need)
        std::set<int32_t> used(m.cid.begin(), m.cid.end());
        used.insert(0);
        nsets = (int32_t)used.size();
    }
    m.res.n_constraint_sets = nsets;
    m.res.truncated = truncated;
    m.res.counters = ctr;
    *out = h;
    return STCSP_OK;
}
const stcsp_result *stcsp_merged_result(const stcsp_merged *m) { return m ? &m->m.res : nullptr; }
void stcsp_merged_free(stcsp_merged *m) { delete m; }

}  // extern "C"
