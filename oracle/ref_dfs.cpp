// ref_dfs.cpp -- ORACLE (test infrastructure, never shipped, never on the product path).
//
// A CPU restatement of the reference's propagation + search algorithm for the path
// solverSolve() covers (reference src/solveralgorithm.cpp:945-1005 and everything it calls),
// written from scratch against the flat stcsp_problem descriptor of include/stcsp_engine.h.
// It deliberately keeps the REFERENCE's algorithm -- interval domains, the under-propagating
// AC-3 arc queue with the reference's exact queue discipline, brute-force support search in
// the reference's variable order, recursive DFS with bisection, per-leaf constraint
// translation, dominance detection -- so that not only the automaton but also the
// reference's counters (dominance / table states / fails, search nodes, arc revisions,
// validate calls) are reproduced. Those counters plus the canonical
// sha256 of the automaton for the reference's 26 example instances are what pin it
// (tests/golden/reference_golden.json, values recorded from the reference itself).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// Each function cites the reference file:line whose behaviour it restates.
#include <pthread.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#include "../stcsp-solver_amd/csrc/okfix.hpp"
#include "stcsp_engine.h"

namespace {

enum { TYPE_NEXT = 0, TYPE_POINT = 1, TYPE_UNTIL = 2, TYPE_AT = 3 };  // constraint.h:33-36

struct Expr {  // constraint.h:24-31
    int token, num, var, arr;
    Expr *left, *right;
};

struct Con;
struct Arc {  // solver.h:14-18
    Con *con;
    int var;
    bool inqueue;
};

struct Con {  // constraint.h:38-49
    Expr *root = nullptr;
    int type = TYPE_POINT;
    bool has_first = false;
    int until_ordinal = -1;  // position among the UNTIL constraints of its set
    std::vector<int> scope;  // REVERSED first-occurrence order (constraint.cpp:309-314)
    std::vector<Arc> arcs;   // first-occurrence order (constraint.cpp:225-235)
};

// One constraint set = one entry of Solver::seenConstraints, with the per-variable constraint
// lists that solverConstraintQueuePush builds while linking (constraint.cpp:201-222).
struct ConSet {
    std::vector<std::unique_ptr<Con>> cons;
    std::vector<std::vector<Con *>> var_cons;  // Variable::constraints for this set
    std::vector<std::unique_ptr<Expr>> pool;
};

struct State {  // graph.h:47-56
    int cid;
    std::vector<int> sig;
    bool fail = false;
};

// ---- CPU baseline on all host cores (SURVEY 8(d) baseline (ii), bench.py only): the reference's algorithm
// with the automaton's STATES as the unit of parallel work. Every worker is a complete Oracle (its own
// domains, trail, arc queue and constraint-set copies: Arc::inqueue is mutable); they share the state table and
// a task queue. A leaf that opens a new state does not recurse into it (solveralgorithm.cpp:857) but queues it,
// with the time-advanced window, for any worker; every leaf edge is logged and the ok/fail bookkeeping
// (:865-909), which depends on the depth-first order, is replaced by the order-independent fixpoint the engine
// uses (okfix.hpp). Same automaton; the tree can differ slightly (which parent opens a state is a race).
struct Shared {
    std::mutex mu;
    std::condition_variable cv;
    std::map<std::vector<int32_t>, int> gsets;      // serialised set content -> global set id
    std::vector<std::vector<int32_t>> gset_words;   // global set id -> content
    std::map<std::pair<int, std::vector<int>>, int> table;
    std::vector<State> states;
    struct Task {
        int vertex, gcid;
        std::vector<int> lb, ub, expire;
    };
    std::deque<Task> queue;
    int active = 0;
    std::atomic<bool> stop{false};
};

struct Oracle {
    Shared *sh = nullptr;            // non-null: worker of a parallel run
    std::vector<int> g_of_local;     // this worker's seen[] index -> global set id (-1: not asked yet)
    std::map<int, int> local_of_g;
    // model
    int N = 0, K = 2;
    std::vector<int> lb, ub;
    std::vector<std::vector<int>> arrays;
    // variable state (variable.h:10-26)
    std::vector<int> curLB, curUB;  // [var*K + point]
    std::vector<int> pval;          // propagateValue
    std::vector<uint8_t> is_sig, is_until;
    int num_sig = 0, num_until = 0, n_until_cons = 0;
    bool has_first = false;  // Solver::hasFirst (sticky)
    // constraint sets
    std::vector<std::unique_ptr<ConSet>> seen;  // Solver::seenConstraints
    ConSet *cur = nullptr;                      // Solver::constrQueue
    int cid = 0;                                // Solver::constraintID
    std::vector<int> expire;                    // Constraint::expire by UNTIL ordinal
    // trail (util.cpp:94-146)
    std::vector<std::pair<int *, int>> trail;
    // arc queue
    std::deque<Arc *> queue;
    // automaton
    std::vector<State> states;
    std::map<std::pair<int, std::vector<int>>, int> table;  // VertexTable (graph.h:64)
    std::vector<int64_t> e_src, e_dst;
    std::vector<int32_t> e_val;
    // counters / limits
    stcsp_counters ctr{};
    int64_t max_nodes = 0;
    double time_limit = 0;
    std::chrono::steady_clock::time_point t0;
    bool stop = false;
    // result storage
    std::vector<int32_t> r_cid, r_sig;
    std::vector<uint8_t> r_fail, r_issig;
    std::string error;

    // ---- trail: backup / levelUp / levelDown (util.cpp:94-146)
    void backup(int *addr) { trail.emplace_back(addr, *addr); }
    void level_up() { trail.emplace_back(nullptr, 0); }
    void level_down() {
        while (trail.back().first != nullptr) {
            *trail.back().first = trail.back().second;
            trail.pop_back();
        }
        trail.pop_back();
    }

    int &LB(int v, int p) { return curLB[v * K + p]; }
    int &UB(int v, int p) { return curUB[v * K + p]; }

    // ---- tree utilities
    Expr *mk(ConSet &s, int token, int num, int var, int arr, Expr *l, Expr *r) {
        s.pool.emplace_back(new Expr{token, num, var, arr, l, r});
        return s.pool.back().get();
    }
    Expr *import_tree(ConSet &s, const stcsp_problem *p, int idx) {
        if (idx < 0) return nullptr;
        const stcsp_node &n = p->nodes[idx];
        Expr *l = import_tree(s, p, n.left);
        Expr *r = import_tree(s, p, n.right);
        return mk(s, n.token, n.num, n.var, n.arr, l, r);
    }

    // constraintNodeHasFirst (constraint.cpp:240-250)
    static bool tree_has_first(const Expr *e) {
        if (!e) return false;
        if (e->token == STCSP_T_FIRST || e->token == STCSP_T_AT) return true;
        if (e->token == STCSP_T_VAR || e->token == STCSP_T_CONST) return false;
        return tree_has_first(e->left) || tree_has_first(e->right);
    }
    // constraintVarLinkRe (constraint.cpp:201-216)
    void link_vars(ConSet &s, Con *c, const Expr *e) {
        if (!e) return;
        if (e->token == STCSP_T_VAR) {
            for (int v : c->scope)
                if (v == e->var) return;
            c->scope.push_back(e->var);
            s.var_cons[e->var].push_back(c);
        } else {
            link_vars(s, c, e->left);
            link_vars(s, c, e->right);
        }
    }
    // solverConstraintQueuePush (constraint.cpp:254-318)
    void push_constraint(ConSet &s, Expr *root) {
        s.cons.emplace_back(new Con());
        Con *c = s.cons.back().get();
        c->root = root;
        if (root->token == STCSP_T_UNTIL_CON) {
            int rv = root->right->var;
            if (!is_until[rv]) {
                is_until[rv] = 1;
                num_until++;
            }
            c->type = TYPE_UNTIL;
            int ord = 0;
            for (auto &o : s.cons)
                if (o.get() != c && o->type == TYPE_UNTIL) ord++;
            c->until_ordinal = ord;
        } else if (root->right && root->right->token == STCSP_T_NEXT) {
            int lv = root->left->var;
            if (!is_sig[lv]) {
                is_sig[lv] = 1;
                num_sig++;
            }
            c->type = TYPE_NEXT;
        } else if (root->right && root->right->token == STCSP_T_AT) {
            c->type = TYPE_AT;
        } else {
            c->type = TYPE_POINT;
        }
        if (tree_has_first(root)) {
            has_first = true;
            c->has_first = true;
        }
        link_vars(s, c, root);
        for (int v : c->scope) c->arcs.push_back(Arc{c, v, false});  // constraintArcLink
        for (size_t i = 0, n = c->scope.size(); i < n / 2; i++) std::swap(c->scope[i], c->scope[n - 1 - i]);
    }

    // ---- constant folding, constraintNodeValue (constraint.cpp:335-439) incl. its quirks
    struct Lifted {
        bool unknown;
        int v;
    };
    Lifted fold(const Expr *e) {
        if (!e) return {true, 0};
        switch (e->token) {
            case STCSP_T_VAR: return {true, 0};
            case STCSP_T_CONST: return {false, e->num};
            case STCSP_T_FIRST: return fold(e->right);
            case STCSP_T_NEXT: return {true, 0};
            case STCSP_T_ARR: {
                Lifted r = fold(e->right);
                if (r.unknown) return r;
                const std::vector<int> &a = arrays[e->arr];
                return {false, (r.v >= 0 && r.v < (int)a.size()) ? a[r.v] : 0};
            }
            case STCSP_T_ABS: {
                Lifted r = fold(e->right);
                if (r.unknown) return r;
                return {false, std::abs(r.v)};
            }
            case STCSP_T_IF: {
                Lifted c = fold(e->left);
                if (c.unknown) return c;
                return c.v ? fold(e->right->left) : fold(e->right->right);
            }
            case STCSP_T_NOT: {
                Lifted r = fold(e->right);
                if (r.unknown) return r;
                return {false, r.v == 0 ? 1 : 0};
            }
            case STCSP_T_AND: {
                Lifted l = fold(e->left);
                if (l.unknown) return l;
                if (l.v == 0) return {false, 0};
                return fold(e->right);
            }
            case STCSP_T_OR: {  // quirk (constraint.cpp:400-414): left != 1 -> 1
                Lifted l = fold(e->left);
                if (l.unknown) return l;
                if (l.v != 1) return {false, 1};
                return fold(e->right);
            }
            default: break;
        }
        Lifted l = fold(e->left), r = fold(e->right);
        if (l.unknown || r.unknown) return {true, 0};
        long long a = l.v, b = r.v;
        switch (e->token) {
            case STCSP_T_LT_OP: return {false, a < b};
            case STCSP_T_GT_OP: return {false, a < b};  // quirk (constraint.cpp:425)
            case STCSP_T_LE_OP: return {false, a <= b};
            case STCSP_T_GE_OP: return {false, a >= b};
            case STCSP_T_EQ_OP: return {false, a == b};
            case STCSP_T_NE_OP: return {false, a != b};
            case STCSP_T_ADD: return {false, (int)(uint32_t)(a + b)};
            case STCSP_T_SUB: return {false, (int)(uint32_t)(a - b)};
            case STCSP_T_MUL: return {false, (int)(uint32_t)((uint64_t)a * (uint64_t)b)};
            case STCSP_T_DIV: return {false, (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : (int)(a / b)};
            case STCSP_T_MOD: return {false, (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : (int)(a % b)};
            default: return {false, 0};
        }
    }
    // constraintNodeTautology (constraint.cpp:443-462)
    bool tautology(const Expr *root) {
        Lifted l = fold(root->left), r = fold(root->right);
        if (l.unknown || r.unknown) return false;
        switch (root->token) {
            case STCSP_T_LT_CON: return l.v < r.v;
            case STCSP_T_GT_CON: return l.v > r.v;
            case STCSP_T_LE_CON: return l.v <= r.v;
            case STCSP_T_GE_CON: return l.v >= r.v;
            case STCSP_T_EQ_CON: return l.v == r.v;
            case STCSP_T_NE_CON: return l.v != r.v;
            case STCSP_T_IMPLY_CON: return l.v <= r.v;
            case STCSP_T_UNTIL_CON: return r.v == 1;
            default: return false;
        }
    }

    // ---- per-leaf translation (constraint.cpp:466-548)
    // constraintNodeTranslateFirst: inside a `first`, variables become their time-0 values
    Expr *translate_first(ConSet &s, const Expr *e) {
        if (!e) return nullptr;
        if (e->token == STCSP_T_VAR) return mk(s, STCSP_T_CONST, LB(e->var, 0), -1, -1, nullptr, nullptr);
        Expr *l = translate_first(s, e->left);
        Expr *r = translate_first(s, e->right);
        return mk(s, e->token, e->num, -1, e->arr, l, r);
    }
    // constraintNodeTranslate
    Expr *translate(ConSet &s, const Expr *e) {
        if (!e) return nullptr;
        if (e->token == STCSP_T_FIRST) {
            Expr *sub = translate_first(s, e->right);
            Lifted v = fold(sub);
            if (v.unknown) return sub;  // "cannot be completely evaluated": keeps the subtree
            return mk(s, STCSP_T_CONST, v.v, -1, -1, nullptr, nullptr);
        }
        if (e->token == STCSP_T_EQ_CON && e->right && e->right->token == STCSP_T_AT) {
            // constraintNodeTranslateAT (constraint.cpp:484-505): X == Y@k -> X == Y@(k-1),
            // or X == first Y when k == 1
            int x = e->left->var, y = e->right->left->var, k = e->right->right->num;
            Expr *l = mk(s, STCSP_T_VAR, 0, x, -1, nullptr, nullptr);
            Expr *yv = mk(s, STCSP_T_VAR, 0, y, -1, nullptr, nullptr);
            Expr *r = (k == 1) ? mk(s, STCSP_T_FIRST, 0, -1, -1, nullptr, yv)
                               : mk(s, STCSP_T_AT, 0, -1, -1, yv, mk(s, STCSP_T_CONST, k - 1, -1, -1, nullptr, nullptr));
            return mk(s, e->token, 0, -1, -1, l, r);
        }
        Expr *l = translate(s, e->left);
        Expr *r = translate(s, e->right);
        return mk(s, e->token, e->num, e->var, e->arr, l, r);
    }
    // constraintNodeEq / constraintQueueEq (constraint.cpp:551-576): token, num, var, shape
    static bool tree_eq(const Expr *a, const Expr *b) {
        if (!a && !b) return true;
        if (!a || !b) return false;
        if (a->token != b->token || a->num != b->num || a->var != b->var) return false;
        return tree_eq(a->left, b->left) && tree_eq(a->right, b->right);
    }
    static bool set_eq(const ConSet &a, const ConSet &b) {
        if (a.cons.size() != b.cons.size()) return false;
        for (size_t i = 0; i < a.cons.size(); i++)
            if (!tree_eq(a.cons[i]->root, b.cons[i]->root)) return false;
        return true;
    }

    // ---- parallel runs: a constraint set travels between workers as its serialised trees
    static void ser_tree(const Expr *e, std::vector<int32_t> &out) {
        if (!e) {
            out.push_back(INT32_MIN);
            return;
        }
        out.push_back(e->token);
        out.push_back(e->num);
        out.push_back(e->var);
        out.push_back(e->arr);
        ser_tree(e->left, out);
        ser_tree(e->right, out);
    }
    static std::vector<int32_t> ser_set(const ConSet &s) {
        std::vector<int32_t> out;
        for (auto &c : s.cons) ser_tree(c->root, out);
        return out;
    }
    Expr *deser_tree(ConSet &s, const std::vector<int32_t> &w, size_t &pos) {
        if (w[pos] == INT32_MIN) {
            pos++;
            return nullptr;
        }
        int token = w[pos], num = w[pos + 1], var = w[pos + 2], arr = w[pos + 3];
        pos += 4;
        Expr *l = deser_tree(s, w, pos);
        Expr *r = deser_tree(s, w, pos);
        return mk(s, token, num, var, arr, l, r);
    }
    // global id of this worker's set `local` (registering its content when nobody has met it yet)
    int global_set(int local) {
        if ((int)g_of_local.size() <= local) g_of_local.resize(local + 1, -1);
        if (g_of_local[local] >= 0) return g_of_local[local];
        std::vector<int32_t> words = ser_set(*seen[local]);
        std::lock_guard<std::mutex> lk(sh->mu);
        auto it = sh->gsets.find(words);
        int g;
        if (it == sh->gsets.end()) {
            g = (int)sh->gset_words.size();
            sh->gsets.emplace(words, g);
            sh->gset_words.push_back(words);
        } else {
            g = it->second;
        }
        g_of_local[local] = g;
        local_of_g[g] = local;
        return g;
    }
    // this worker's copy of global set g (built from the registry when another worker discovered it)
    int local_set(int g) {
        auto it = local_of_g.find(g);
        if (it != local_of_g.end()) return it->second;
        std::vector<int32_t> words;
        {
            std::lock_guard<std::mutex> lk(sh->mu);
            words = sh->gset_words[g];
        }
        std::unique_ptr<ConSet> ns(new ConSet());
        ns->var_cons.resize(N);
        size_t pos = 0;
        while (pos < words.size()) push_constraint(*ns, deser_tree(*ns, words, pos));
        int local = (int)seen.size();
        seen.push_back(std::move(ns));
        if ((int)g_of_local.size() <= local) g_of_local.resize(local + 1, -1);
        g_of_local[local] = g;
        local_of_g[g] = local;
        return local;
    }

    // ---- expression evaluation, solverValidateRe (solveralgorithm.cpp:336-424)
    int eval(const Expr *e, bool &valid) {
        if (!e) return 0;
        switch (e->token) {
            case STCSP_T_VAR: return pval[e->var];
            case STCSP_T_ARR: {
                int i = eval(e->right, valid);
                const std::vector<int> &a = arrays[e->arr];
                if (i < 0 || i >= (int)a.size()) {
                    valid = false;
                    return 0;
                }
                return a[i];
            }
            case STCSP_T_CONST: return e->num;
            case STCSP_T_ABS: {
                int v = eval(e->right, valid);
                return v < 0 ? (int)(0u - (unsigned)v) : v;
            }
            case STCSP_T_IF: return eval(e->left, valid) ? eval(e->right->left, valid) : eval(e->right->right, valid);
            case STCSP_T_FIRST: return eval(e->right, valid);
            case STCSP_T_AT: return eval(e->left, valid);
            case STCSP_T_NOT: return eval(e->right, valid) == 0 ? 1 : 0;
            case STCSP_T_AND: return eval(e->left, valid) ? eval(e->right, valid) : 0;
            case STCSP_T_OR: return eval(e->left, valid) ? 1 : eval(e->right, valid);
            case STCSP_T_IMPLY_CON: {
                int l = eval(e->left, valid);
                if (l == 0) return 1;
                return l <= eval(e->right, valid);
            }
            default: break;
        }
        int l = eval(e->left, valid);
        int r = eval(e->right, valid);
        if (!valid) return 0;
        switch (e->token) {
            case STCSP_T_LT_CON: case STCSP_T_LT_OP: return l < r;
            case STCSP_T_GT_CON: case STCSP_T_GT_OP: return l > r;
            case STCSP_T_LE_CON: case STCSP_T_LE_OP: return l <= r;
            case STCSP_T_GE_CON: case STCSP_T_GE_OP: return l >= r;
            case STCSP_T_EQ_CON: case STCSP_T_EQ_OP: return l == r;
            case STCSP_T_NE_CON: case STCSP_T_NE_OP: return l != r;
            case STCSP_T_ADD: return (int)((unsigned)l + (unsigned)r);
            case STCSP_T_SUB: return (int)((unsigned)l - (unsigned)r);
            case STCSP_T_MUL: return (int)((unsigned)l * (unsigned)r);
            // the reference traps on /0; the engine defines x/0 = x%0 = 0 and so does this
            case STCSP_T_DIV: return (r == 0 || (l == INT32_MIN && r == -1)) ? 0 : l / r;
            case STCSP_T_MOD: return (r == 0 || (l == INT32_MIN && r == -1)) ? 0 : l % r;
            default: return 0;
        }
    }
    // validate (solveralgorithm.cpp:428-431)
    bool validate(const Con *c) {
        ctr.evaluations++;
        // (time-boxed runs only: a single support search over wide domains can take minutes; once the box is spent every
        // search ends as "supported" and the run is reported truncated -- its automaton is not used)
        if (time_limit > 0) {
            if ((ctr.evaluations & 0xfffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_limit) stop = true;
            if (stop) return true;
        }
        bool valid = true;
        return eval(c->root, valid) != 0;
    }
    // findSupportRe (solveralgorithm.cpp:435-464): nested loops over the INTERVALS of the other
    // scope variables, in reversed first-occurrence order, first support exits
    bool find_support(const Con *c, int var, int point, size_t index) {
        int v = c->scope[index];
        bool last = index + 1 == c->scope.size();
        if (v == var) return last ? validate(c) : find_support(c, var, point, index + 1);
        bool supported = false;
        int lo = LB(v, point), hi = UB(v, point);
        for (int x = lo; !supported && x <= hi; x++) {
            pval[v] = x;
            supported = last ? validate(c) : find_support(c, var, point, index + 1);
        }
        return supported;
    }
    // enforcePointConsistencyAt (solveralgorithm.cpp:476-523): bounds tightening by support
    bool revise_point_at(const Con *c, int var, bool &change, int point) {
        bool supported = false;
        int lo = LB(var, point), hi = UB(var, point);
        for (int x = lo; !supported && x <= hi; x++) {
            pval[var] = x;
            supported = find_support(c, var, point, 0);
        }
        if (!supported) return false;
        if (lo != pval[var]) {
            change = true;
            backup(&LB(var, point));
            LB(var, point) = pval[var];
        }
        supported = false;
        lo = LB(var, point);
        for (int x = hi; !supported && x > lo; x--) {
            pval[var] = x;
            supported = find_support(c, var, point, 0);
        }
        if (!supported) {
            if (hi != lo) {
                change = true;
                backup(&UB(var, point));
                UB(var, point) = lo;
            }
        } else if (hi != pval[var]) {
            change = true;
            backup(&UB(var, point));
            UB(var, point) = pval[var];
        }
        return true;
    }
    // enforcePointConsistency (solveralgorithm.cpp:527-539)
    bool revise_point(const Con *c, int var, bool &change) {
        if (c->has_first) return revise_point_at(c, var, change, 0);
        for (int p = 0; p < K; p++)
            if (!revise_point_at(c, var, change, p)) return false;
        return true;
    }
    // enforceNextConsistency (solveralgorithm.cpp:544-593) for X == next Y
    bool revise_next(const Con *c, int var, bool &change) {
        int X = c->root->left->var, Y = c->root->right->right->var;
        if (var == Y) {
            for (int p = 1; p < K; p++) {
                if (LB(Y, p) < LB(X, p - 1)) { change = true; backup(&LB(Y, p)); LB(Y, p) = LB(X, p - 1); }
                if (UB(Y, p) > UB(X, p - 1)) { change = true; backup(&UB(Y, p)); UB(Y, p) = UB(X, p - 1); }
                if (LB(Y, p) > UB(Y, p)) return false;
            }
        } else {
            for (int p = 0; p < K - 1; p++) {
                if (LB(var, p) < LB(Y, p + 1)) { change = true; backup(&LB(var, p)); LB(var, p) = LB(Y, p + 1); }
                if (UB(var, p) > UB(Y, p + 1)) { change = true; backup(&UB(var, p)); UB(var, p) = UB(Y, p + 1); }
                if (LB(var, p) > UB(var, p)) return false;
            }
        }
        return true;
    }
    // enforceUntilConsistency (solveralgorithm.cpp:598-614): a check, never prunes
    bool revise_until(const Con *c) {
        if (expire[c->until_ordinal]) return true;
        int X = c->root->left->var, Y = c->root->right->var;
        if (LB(X, 0) == UB(X, 0) && LB(Y, 0) == UB(Y, 0) && LB(Y, 0) != 1 && LB(X, 0) != 1) return false;
        return true;
    }
    // generalisedArcConsistent (solveralgorithm.cpp:617-706)
    bool gac() {
        ctr.gac_calls++;
        for (Arc *a : queue) a->inqueue = false;
        queue.clear();
        for (auto &c : cur->cons)
            for (Arc &a : c->arcs) {
                a.inqueue = true;
                queue.push_back(&a);
            }
        bool consistent = true, change = false;
        while (consistent && !queue.empty()) {
            Arc *arc = queue.front();
            queue.pop_front();
            arc->inqueue = false;
            ctr.revisions++;
            Con *c = arc->con;
            switch (c->type) {
                case TYPE_NEXT: consistent = revise_next(c, arc->var, change); break;
                case TYPE_POINT: consistent = revise_point(c, arc->var, change); break;
                case TYPE_UNTIL: consistent = revise_until(c); break;
                default: break;  // CONSTR_AT: lazily handled by translation (:658-662)
            }
            if (consistent && change) {
                // only arcs of the variable's OTHER constraints are re-queued (:679-693)
                for (Con *o : cur->var_cons[arc->var]) {
                    if (o == c) continue;
                    for (Arc &a : o->arcs)
                        if (!a.inqueue) {
                            a.inqueue = true;
                            queue.push_back(&a);
                        }
                }
            }
            change = false;
        }
        for (Arc *a : queue) a->inqueue = false;
        queue.clear();
        return consistent;
    }

    bool over_budget() {
        if (sh && sh->stop.load(std::memory_order_relaxed)) stop = true;
        if (stop) return true;
        if (max_nodes && ctr.search_nodes >= max_nodes) stop = true;
        if (time_limit > 0 && (ctr.search_nodes & 1023) == 0) {
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s > time_limit) stop = true;
        }
        if (stop && sh) sh->stop.store(true, std::memory_order_relaxed);
        return stop;
    }

    // solverSolveRe (solveralgorithm.cpp:733-942)
    bool search(int vertex) {
        if (over_budget()) return false;
        ctr.search_nodes++;
        int var = -1;  // solverGetFirstUnboundVar (solver.cpp:41-53)
        for (int v = 0; v < N; v++)
            if (LB(v, 0) < UB(v, 0)) {
                var = v;
                break;
            }
        bool ok = false;
        if (var < 0) {
            ctr.leaves++;
            level_up();
            ConSet *saved = cur;
            if (has_first) {  // constraint translation + set identification (:755-805)
                std::unique_ptr<ConSet> ns(new ConSet());
                ns->var_cons.resize(N);
                std::vector<Expr *> roots;
                for (auto &c : cur->cons) {
                    Expr *t = translate(*ns, c->root);  // constraintTranslate (constraint.cpp:540-548)
                    if (!tautology(t)) roots.push_back(t);
                }
                for (Expr *t : roots) push_constraint(*ns, t);
                backup(&cid);
                int found = -1;
                for (size_t i = 0; i < seen.size() && found < 0; i++)
                    if (set_eq(*ns, *seen[i])) found = (int)i;
                if (found < 0) {
                    found = (int)seen.size();
                    seen.push_back(std::move(ns));
                }
                cid = found;
                cur = seen[found].get();  // same content as the freshly translated queue
            }
            // signature (:812-837): signature variables in varQueue order, then one sticky
            // 0/1 flag per UNTIL constraint
            std::vector<int> sig;
            for (int v = 0; v < N; v++)
                if (is_sig[v]) sig.push_back(LB(v, 0));
            for (auto &c : cur->cons)
                if (c->type == TYPE_UNTIL) {
                    int &ex = expire[c->until_ordinal];
                    backup(&ex);
                    if (ex == 1) {
                        sig.push_back(1);
                    } else if (LB(c->root->right->var, 0) == 1) {
                        ex = 1;
                        sig.push_back(1);
                    } else {
                        sig.push_back(0);
                    }
                }
            if (sh) {
                // parallel run: look the state up in the shared table; a new state becomes a task
                const int g = global_set(cid);
                auto key = std::make_pair(g, sig);
                int dst;
                bool is_new = false;
                {
                    std::lock_guard<std::mutex> lk(sh->mu);
                    auto it = sh->table.find(key);
                    if (it == sh->table.end()) {
                        dst = (int)sh->states.size();
                        sh->states.push_back(State{g, sig, false});
                        sh->table.emplace(key, dst);
                        is_new = true;
                    } else {
                        dst = it->second;
                    }
                }
                if (is_new) {
                    Shared::Task t;
                    t.vertex = dst;
                    t.gcid = g;
                    t.lb.resize((size_t)N * K);
                    t.ub.resize((size_t)N * K);
                    for (int v = 0; v < N; v++) {  // variableAdvanceOneTimeStep (variable.cpp:94-108)
                        for (int p = 0; p < K - 1; p++) {
                            t.lb[v * K + p] = LB(v, p + 1);
                            t.ub[v * K + p] = UB(v, p + 1);
                        }
                        t.lb[v * K + K - 1] = lb[v];
                        t.ub[v * K + K - 1] = ub[v];
                    }
                    t.expire = expire;
                    {
                        std::lock_guard<std::mutex> lk(sh->mu);
                        sh->queue.push_back(std::move(t));
                    }
                    sh->cv.notify_one();
                } else {
                    ctr.dominance++;
                }
                cur = saved;
                e_src.push_back(vertex);  // every leaf edge is logged (this worker's log); okfix decides later
                e_dst.push_back(dst);
                for (int v = 0; v < N; v++) e_val.push_back(LB(v, 0));
                level_down();
                return !stop;
            }
            auto key = std::make_pair(cid, sig);
            auto it = table.find(key);
            int dst;
            if (it == table.end()) {  // new state (:842-864)
                dst = (int)states.size();
                states.push_back(State{cid, sig, false});
                table.emplace(key, dst);
                for (int v = 0; v < N; v++) {  // variableAdvanceOneTimeStep (variable.cpp:94-108)
                    for (int p = 0; p < K; p++) {
                        backup(&LB(v, p));
                        backup(&UB(v, p));
                    }
                    for (int p = 0; p < K - 1; p++) {
                        LB(v, p) = LB(v, p + 1);
                        UB(v, p) = UB(v, p + 1);
                    }
                    LB(v, K - 1) = lb[v];
                    UB(v, K - 1) = ub[v];
                }
                if (gac()) {
                    ok = search(dst);
                } else {
                    ctr.fails++;
                    ok = false;
                }
            } else {
                dst = it->second;
                if (states[dst].fail) {
                    ok = false;
                } else {
                    ctr.dominance++;
                    ok = true;
                }
            }
            cur = saved;
            level_down();
            if (stop) return false;  // budget hit below: leave the partial automaton untouched
            if (ok) {  // edgeNew + vertexAddEdge (graph.cpp:78-89, 33-38)
                e_src.push_back(vertex);
                e_dst.push_back(dst);
                for (int v = 0; v < N; v++) e_val.push_back(LB(v, 0));
            } else {
                states[dst].fail = true;
            }
        } else {
            // variableSplitLower / variableSplitUpper (variable.cpp:52-67)
            level_up();
            backup(&UB(var, 0));
            UB(var, 0) = LB(var, 0) + (UB(var, 0) - LB(var, 0)) / 2;
            if (gac()) ok |= search(vertex); else ctr.fails++;
            level_down();
            level_up();
            backup(&LB(var, 0));
            LB(var, 0) = LB(var, 0) + (UB(var, 0) - LB(var, 0)) / 2 + 1;
            if (!stop) {
                if (gac()) ok |= search(vertex); else ctr.fails++;
            }
            level_down();
        }
        return ok;
    }

    int setup(const stcsp_problem *p, const stcsp_options *o) {
        if (!p || p->n_vars <= 0 || p->prefix_k <= 0) {
            error = "invalid problem";
            return STCSP_E_INVALID;
        }
        N = p->n_vars;
        K = p->prefix_k;
        lb.assign(p->var_lb, p->var_lb + N);
        ub.assign(p->var_ub, p->var_ub + N);
        for (int a = 0; a < p->n_arrays; a++)
            arrays.emplace_back(p->array_data + p->array_off[a], p->array_data + p->array_off[a + 1]);
        curLB.resize((size_t)N * K);
        curUB.resize((size_t)N * K);
        for (int v = 0; v < N; v++)
            for (int q = 0; q < K; q++) {
                LB(v, q) = lb[v];
                UB(v, q) = ub[v];
            }
        pval.assign(N, 0);
        is_sig.assign(N, 0);
        is_until.assign(N, 0);
        std::unique_ptr<ConSet> s0(new ConSet());
        s0->var_cons.resize(N);
        for (int c = 0; c < p->n_constraints; c++) push_constraint(*s0, import_tree(*s0, p, p->constraint_root[c]));
        for (auto &c : s0->cons)
            if (c->type == TYPE_UNTIL) n_until_cons++;
        expire.assign(n_until_cons, 0);
        seen.push_back(std::move(s0));  // solveralgorithm.cpp:948
        cur = seen[0].get();
        cid = 0;
        if (o) {
            max_nodes = o->max_search_nodes;
            time_limit = o->time_limit_s;
        }
        return STCSP_OK;
    }

    // solverSolve (solveralgorithm.cpp:945-971)
    void run() {
        t0 = std::chrono::steady_clock::now();
        states.push_back(State{0, {}, false});
        table.emplace(std::make_pair(0, std::vector<int>()), 0);
        level_up();
        if (gac())
            search(0);
        else
            ctr.fails++;
        level_down();
        ctr.seconds_search = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }

    // worker of a parallel run: take states from the shared queue until none is left and nobody is working
    void work() {
        t0 = std::chrono::steady_clock::now();
        for (;;) {
            Shared::Task t;
            {
                std::unique_lock<std::mutex> lk(sh->mu);
                sh->cv.wait(lk, [&] { return !sh->queue.empty() || sh->active == 0 || sh->stop.load(); });
                if (sh->stop.load() || (sh->queue.empty() && sh->active == 0)) {
                    sh->cv.notify_all();
                    return;
                }
                if (sh->queue.empty()) continue;
                t = std::move(sh->queue.front());
                sh->queue.pop_front();
                sh->active++;
            }
            // install the state: its constraint set, the time-advanced window, the until flags
            const int local = local_set(t.gcid);
            cur = seen[local].get();
            cid = local;
            curLB = t.lb;
            curUB = t.ub;
            expire = t.expire;
            trail.clear();
            level_up();
            if (gac()) {
                search(t.vertex);
            } else {
                ctr.fails++;
            }
            level_down();
            {
                std::lock_guard<std::mutex> lk(sh->mu);
                sh->active--;
            }
            sh->cv.notify_all();
        }
    }

    void fill(stcsp_result *r) {
        memset(r, 0, sizeof *r);
        int sl = num_sig + n_until_cons;
        r->n_states = (int64_t)states.size();
        r->sig_len = sl;
        r->n_sig_vars = num_sig;
        r->n_until = num_until;
        r->n_until_cons = n_until_cons;
        r_cid.resize(states.size());
        r_sig.assign(states.size() * (size_t)sl, 0);
        r_fail.resize(states.size());
        for (size_t i = 0; i < states.size(); i++) {
            r_cid[i] = states[i].cid;
            r_fail[i] = states[i].fail;
            for (size_t j = 0; j < states[i].sig.size() && j < (size_t)sl; j++) r_sig[i * sl + j] = states[i].sig[j];
        }
        r_issig.assign(is_sig.begin(), is_sig.end());
        r->state_cid = r_cid.data();
        r->state_sig = r_sig.data();
        r->state_fail = r_fail.data();
        r->n_edges = (int64_t)e_src.size();
        r->edge_src = e_src.data();
        r->edge_dst = e_dst.data();
        r->edge_values = e_val.data();
        r->n_vars = N;
        r->n_constraint_sets = (int32_t)seen.size();
        r->var_is_signature = r_issig.data();
        r->root_final = n_until_cons == 0;
        r->truncated = stop;
        r->counters = ctr;
    }
};

struct ThreadArg {
    Oracle *o;
};
void *thread_main(void *p) {
    static_cast<ThreadArg *>(p)->o->run();
    return nullptr;
}
void *worker_main(void *p) {
    static_cast<ThreadArg *>(p)->o->work();
    return nullptr;
}

}  // namespace

struct stcsp_oracle {
    Oracle o;
    // parallel runs
    const stcsp_problem *problem = nullptr;
    stcsp_options options{};
    Shared shared;
    std::vector<std::unique_ptr<Oracle>> workers;
    std::vector<int32_t> p_cid, p_sig, p_val;
    std::vector<uint8_t> p_fail, p_issig;
    std::vector<int64_t> p_src, p_dst;
};

extern "C" {

// Same shape as stcsp_engine_create/solve/destroy so the parity tests drive both alike.
int stcsp_oracle_create(const stcsp_problem *problem, const stcsp_options *options, stcsp_oracle **out) {
    std::unique_ptr<stcsp_oracle> h(new stcsp_oracle());
    int rc = h->o.setup(problem, options);
    if (rc != STCSP_OK) return rc;
    h->problem = problem;  // (the caller keeps the problem alive as long as the oracle: bench.py / tests do)
    if (options) h->options = *options;
    *out = h.release();
    return STCSP_OK;
}

// The reference's algorithm on `threads` host cores (see struct Shared). Same result layout as solve().
int stcsp_oracle_solve_parallel(stcsp_oracle *h, int threads, stcsp_result *result) {
    if (!h || !result || threads < 1 || !h->problem) return STCSP_E_INVALID;
    Shared &sh = h->shared;
    sh.table.clear();
    sh.states.clear();
    sh.queue.clear();
    sh.gsets.clear();
    sh.gset_words.clear();
    sh.active = 0;
    sh.stop = false;
    h->workers.clear();
    for (int t = 0; t < threads; t++) {
        h->workers.emplace_back(new Oracle());
        Oracle &w = *h->workers.back();
        int rc = w.setup(h->problem, &h->options);
        if (rc != STCSP_OK) return rc;
        w.sh = &sh;
    }
    Oracle &w0 = *h->workers[0];
    w0.global_set(0);  // the model's own set is global set 0 (identical in every clone)
    for (auto &w : h->workers) {
        w->g_of_local.assign(1, 0);
        w->local_of_g[0] = 0;
    }
    // root: Signature({}, 0) with the initial domains (solveralgorithm.cpp:951-968)
    sh.states.push_back(State{0, {}, false});
    sh.table.emplace(std::make_pair(0, std::vector<int>()), 0);
    Shared::Task root;
    root.vertex = 0;
    root.gcid = 0;
    root.lb = w0.curLB;
    root.ub = w0.curUB;
    root.expire = w0.expire;
    sh.queue.push_back(std::move(root));
    auto t0 = std::chrono::steady_clock::now();
    std::vector<pthread_t> th(threads);
    std::vector<ThreadArg> args(threads);
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, (size_t)256 << 20);
    for (int t = 0; t < threads; t++) {
        args[t].o = h->workers[t].get();
        if (pthread_create(&th[t], &attr, worker_main, &args[t]) != 0) return STCSP_E_NOMEM;
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], nullptr);
    pthread_attr_destroy(&attr);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // merge the workers' edge logs, then the order-independent ok/fail fixpoint
    const int N = w0.N;
    h->p_src.clear();
    h->p_dst.clear();
    h->p_val.clear();
    stcsp_counters ctr{};
    for (auto &w : h->workers) {
        h->p_src.insert(h->p_src.end(), w->e_src.begin(), w->e_src.end());
        h->p_dst.insert(h->p_dst.end(), w->e_dst.begin(), w->e_dst.end());
        h->p_val.insert(h->p_val.end(), w->e_val.begin(), w->e_val.end());
        ctr.search_nodes += w->ctr.search_nodes;
        ctr.gac_calls += w->ctr.gac_calls;
        ctr.fails += w->ctr.fails;
        ctr.leaves += w->ctr.leaves;
        ctr.revisions += w->ctr.revisions;
        ctr.evaluations += w->ctr.evaluations;
    }
    const int64_t ns = (int64_t)sh.states.size();
    h->p_fail.assign((size_t)ns, 0);
    std::vector<uint8_t> alive;
    stcsp::ok_fixpoint(ns, h->p_src, h->p_dst, h->p_fail, alive);
    size_t wq = 0;
    for (size_t e = 0; e < alive.size(); e++)
        if (alive[e]) {
            h->p_src[wq] = h->p_src[e];
            h->p_dst[wq] = h->p_dst[e];
            if (wq != e) memmove(&h->p_val[wq * N], &h->p_val[e * N], (size_t)N * 4);
            wq++;
        }
    h->p_src.resize(wq);
    h->p_dst.resize(wq);
    h->p_val.resize(wq * N);
    const int sl = w0.num_sig + w0.n_until_cons;
    h->p_cid.resize((size_t)ns);
    h->p_sig.assign((size_t)ns * sl, 0);
    int64_t ok_states = 0;
    for (int64_t i = 0; i < ns; i++) {
        h->p_cid[i] = sh.states[i].cid;
        for (size_t j = 0; j < sh.states[i].sig.size() && j < (size_t)sl; j++) h->p_sig[i * sl + j] = sh.states[i].sig[j];
        if (i) ok_states += !h->p_fail[i];
    }
    ctr.dominance = (int64_t)wq - ok_states;
    ctr.seconds_search = secs;
    h->p_issig.assign(w0.is_sig.begin(), w0.is_sig.end());
    memset(result, 0, sizeof *result);
    result->n_states = ns;
    result->sig_len = sl;
    result->n_sig_vars = w0.num_sig;
    result->n_until = w0.num_until;
    result->n_until_cons = w0.n_until_cons;
    result->state_cid = h->p_cid.data();
    result->state_sig = h->p_sig.data();
    result->state_fail = h->p_fail.data();
    result->n_edges = (int64_t)wq;
    result->edge_src = h->p_src.data();
    result->edge_dst = h->p_dst.data();
    result->edge_values = h->p_val.data();
    result->n_vars = N;
    result->n_constraint_sets = (int32_t)sh.gset_words.size();
    result->var_is_signature = h->p_issig.data();
    result->root_final = w0.n_until_cons == 0;
    result->truncated = sh.stop.load();
    result->counters = ctr;
    return STCSP_OK;
}

int stcsp_oracle_solve(stcsp_oracle *h, stcsp_result *result) {
    if (!h || !result) return STCSP_E_INVALID;
    // The reference raises RLIMIT_STACK to unlimited (stcsp.y:184-191) because the recursion
    // reaches tens of thousands of frames (42,862 on digitinvader9); here the DFS runs on a
    // thread with a large lazily-committed stack instead.
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, (size_t)8 << 30);
    pthread_t th;
    ThreadArg arg{&h->o};
    if (pthread_create(&th, &attr, thread_main, &arg) != 0) {
        pthread_attr_setstacksize(&attr, (size_t)1 << 30);
        if (pthread_create(&th, &attr, thread_main, &arg) != 0) return STCSP_E_NOMEM;
    }
    pthread_join(th, nullptr);
    pthread_attr_destroy(&attr);
    h->o.fill(result);
    return STCSP_OK;
}

void stcsp_oracle_destroy(stcsp_oracle *h) { delete h; }

}  // extern "C"
