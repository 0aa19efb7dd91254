// cset.cpp -- see cset.hpp.
#include "cset.hpp"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>

namespace stcsp {

// constraintNodeHasFirst (reference src/constraint.cpp:240-250)
static bool tree_has_first(const Tree *t) {
    if (!t) return false;
    if (t->token == STCSP_T_FIRST || t->token == STCSP_T_AT) return true;
    if (t->token == STCSP_T_VAR || t->token == STCSP_T_CONST) return false;
    return tree_has_first(t->left) || tree_has_first(t->right);
}
static bool tree_has_arr(const Tree *t) {
    if (!t) return false;
    if (t->token == STCSP_T_ARR) return true;
    return tree_has_arr(t->left) || tree_has_arr(t->right);
}
// constraintVarLinkRe (src/constraint.cpp:201-216): distinct identifiers, first-occurrence order
static void collect_scope(const Tree *t, std::vector<int> &scope) {
    if (!t) return;
    if (t->token == STCSP_T_VAR) {
        if (std::find(scope.begin(), scope.end(), t->var) == scope.end()) scope.push_back(t->var);
    } else {
        collect_scope(t->left, scope);
        collect_scope(t->right, scope);
    }
}
static void collect_first_vars(const Tree *t, bool under_first, std::set<int> &out) {
    if (!t) return;
    if (t->token == STCSP_T_VAR) {
        if (under_first) out.insert(t->var);
        return;
    }
    bool u = under_first || t->token == STCSP_T_FIRST;
    collect_first_vars(t->left, u, out);
    collect_first_vars(t->right, u, out);
}
// constraintNodeEq (src/constraint.cpp:551-561): token, num, var and shape
static bool tree_eq(const Tree *a, const Tree *b) {
    if (!a && !b) return true;
    if (!a || !b) return false;
    if (a->token != b->token || a->num != b->num || a->var != b->var) return false;
    return tree_eq(a->left, b->left) && tree_eq(a->right, b->right);
}
static bool set_eq(const HostSet &a, const HostSet &b) {  // constraintQueueEq (:564-576)
    if (a.cons.size() != b.cons.size()) return false;
    for (size_t i = 0; i < a.cons.size(); i++)
        if (!tree_eq(a.cons[i].root, b.cons[i].root)) return false;
    return true;
}
static void serialise_tree(const Tree *t, std::vector<int32_t> &out) {
    if (!t) {
        out.push_back(-1);
        return;
    }
    out.push_back(t->token);
    out.push_back(t->num);
    out.push_back(t->var);
    out.push_back(t->arr);
    serialise_tree(t->left, out);
    serialise_tree(t->right, out);
}
static Tree *deserialise_tree(const int32_t *&p, const int32_t *end, TreeArena &arena) {
    if (p >= end) return nullptr;
    int tok = *p++;
    if (tok < 0) return nullptr;
    if (p + 3 > end) return nullptr;
    int num = *p++, var = *p++, arr = *p++;
    Tree *l = deserialise_tree(p, end, arena);
    Tree *r = deserialise_tree(p, end, arena);
    return arena.make(tok, num, var, arr, l, r);
}

// solverConstraintQueuePush (src/constraint.cpp:254-318)
int SetManager::push_constraint(HostSet &s, Tree *root) {
    HostCon c;
    c.root = root;
    if (!root || !is_constraint_root(root->token) || !root->left || !root->right) {
        error = "constraint root is not a constraint operator";
        return STCSP_E_INVALID;
    }
    if (root->token == STCSP_T_UNTIL_CON) {
        if (root->left->token != STCSP_T_VAR || root->right->token != STCSP_T_VAR) {
            error = "until operands must be variables after normalisation";
            return STCSP_E_INVALID;
        }
        c.type = CT_UNTIL;
        c.x = root->left->var;
        c.y = root->right->var;
        if (!is_until[c.y]) {
            is_until[c.y] = 1;
            n_until++;
        }
        int ord = 0;
        for (auto &o : s.cons) ord += o.type == CT_UNTIL;
        c.until_ordinal = ord;
    } else if (root->right->token == STCSP_T_NEXT) {
        if (root->left->token != STCSP_T_VAR || !root->right->right || root->right->right->token != STCSP_T_VAR) {
            error = "next constraint is not of the normalised form X == next Y";
            return STCSP_E_INVALID;
        }
        c.type = CT_NEXT;
        c.x = root->left->var;
        c.y = root->right->right->var;
        if (!is_sig[c.x]) {
            is_sig[c.x] = 1;
            n_sig++;
        }
    } else if (root->right->token == STCSP_T_AT) {
        c.type = CT_AT;
    } else {
        c.type = CT_POINT;
    }
    if (tree_has_first(root)) {
        has_first = true;
        c.has_first = true;
    }
    collect_scope(root, c.scope);
    s.cons.push_back(c);
    return STCSP_OK;
}

void SetManager::finish_set(HostSet &s) {
    std::set<int> fv;
    bool any_first_or_at = false;
    for (auto &c : s.cons) {
        collect_first_vars(c.root, false, fv);
        any_first_or_at |= tree_has_first(c.root);
    }
    s.first_vars.assign(fv.begin(), fv.end());
    s.self_loop = !any_first_or_at;
}

// Hash of exactly what set_eq / tree_eq compare -- token, num, var and the shape; NOT Tree::arr, which constraintNodeEq
// (src/constraint.cpp:551-561) ignores: two translated sets that differ only in the array a node indexes are ONE set to the
// reference's seenConstraints lookup, so they must land in one bucket here (serialise_tree, with arr, is the wire format
// of sharded runs only).
static void hash_tree(const Tree *t, uint64_t &h) {
    auto mix = [&](int32_t w) {
        h ^= (uint32_t)w;
        h *= 1099511628211ull;
    };
    if (!t) {
        mix(-1);
        return;
    }
    mix(t->token);
    mix(t->num);
    mix(t->var);
    hash_tree(t->left, h);
    hash_tree(t->right, h);
}
static uint64_t set_content_hash(const HostSet &s) {
    uint64_t h = 1469598103934665603ull;
    for (auto &c : s.cons) hash_tree(c.root, h);
    return h;
}

int SetManager::register_set(std::unique_ptr<HostSet> s) {
    finish_set(*s);
    int idx = (int)sets.size();
    const uint64_t h = set_content_hash(*s);
    set_by_hash.emplace(h, idx);
    if (sharded) {  // (what set_eq compares: shards that meet array-variants of one set agree on its tag)
        int32_t tag = (int32_t)((h ^ (h >> 31)) & 0x3fffffff);
        // The tag must depend on the set's content only (shards discover sets in different orders), so a
        // collision cannot be probed away: it is reported (2^-30 per pair of sets).
        if (idx == 0) tag = 0;  // the initial set is set 0 everywhere
        else if (tag == 0) tag = 0x3fffffff;
        if (find_tag(tag) >= 0) {
            error = "two constraint sets share the content tag " + std::to_string(tag) + " (sharded runs identify sets by a 30-bit content hash)";
            return STCSP_E_UNSUPPORTED;
        }
        s->tag = tag;
    } else {
        s->tag = idx;
    }
    sets.push_back(std::move(s));
    return idx;
}

int SetManager::find_tag(int32_t tag) const {
    for (size_t i = 0; i < sets.size(); i++)
        if (sets[i]->tag == tag) return (int)i;
    return -1;
}

int SetManager::init(const stcsp_problem *p, bool sharded_tags) {
    if (!p || p->n_vars <= 0 || p->prefix_k <= 0 || !p->var_lb || !p->var_ub || p->n_constraints < 0) {
        error = "invalid problem descriptor";
        return STCSP_E_INVALID;
    }
    sharded = sharded_tags;
    if (const char *ev = getenv("STCSP_FRESH_INIT")) fresh_init = atoi(ev) != 0;
    if (const char *ev = getenv("STCSP_SPLIT_WIDE")) split_mode = atoi(ev);  // tuning / A-B: 0 keeps wide conditionals interpreted
    N = p->n_vars;
    K = p->prefix_k;
    lb.assign(p->var_lb, p->var_lb + N);
    ub.assign(p->var_ub, p->var_ub + N);
    for (int v = 0; v < N; v++)
        if (lb[v] > ub[v]) {
            error = "variable with empty domain";
            return STCSP_E_INVALID;
        }
    {
        long long widest = 1;
        for (int v = 0; v < N; v++) widest = std::max(widest, (long long)ub[v] - (long long)lb[v] + 1);
        W = widest <= 32 ? 1 : (widest <= 64 ? 2 : (widest <= 128 ? 4 : 0));
    }
    array_off.assign(1, 0);
    for (int a = 0; a < p->n_arrays; a++) {
        arrays.elements.emplace_back(p->array_data + p->array_off[a], p->array_data + p->array_off[a + 1]);
        array_data.insert(array_data.end(), arrays.elements.back().begin(), arrays.elements.back().end());
        array_off.push_back((int32_t)array_data.size());
    }
    is_sig.assign(N, 0);
    is_until.assign(N, 0);
    std::unique_ptr<HostSet> s0(new HostSet());
    for (int c = 0; c < p->n_constraints; c++) {
        int root = p->constraint_root[c];
        if (root < 0 || root >= p->n_nodes) {
            error = "constraint root out of range";
            return STCSP_E_INVALID;
        }
        int rc = push_constraint(*s0, unflatten(p, root, s0->arena));
        if (rc != STCSP_OK) return rc;
    }
    for (auto &c : s0->cons)
        if (c.type == CT_UNTIL) {
            n_until_cons++;
            until_x.push_back(c.x);
            until_y.push_back(c.y);
        }
    for (int v = 0; v < N; v++)
        if (is_sig[v]) sig_vars.push_back(v);
    const int r0 = register_set(std::move(s0));
    if (r0 < 0) return r0;
    if (!has_first) sets[0]->self_loop = true;  // no translation at all (solveralgorithm.cpp:755)
    return STCSP_OK;
}

// constraintNodeTranslateFirst (src/constraint.cpp:466-480)
Tree *SetManager::translate_first(HostSet &dst, const Tree *t, const std::map<int, int> &vals) {
    if (!t) return nullptr;
    if (t->token == STCSP_T_VAR) return dst.arena.constant(vals.at(t->var));
    Tree *l = translate_first(dst, t->left, vals);
    Tree *r = translate_first(dst, t->right, vals);
    return dst.arena.make(t->token, t->num, -1, t->arr, l, r);
}
// constraintNodeTranslate (src/constraint.cpp:509-537)
Tree *SetManager::translate(HostSet &dst, const Tree *t, const std::map<int, int> &vals) {
    if (!t) return nullptr;
    if (t->token == STCSP_T_FIRST) {
        Tree *sub = translate_first(dst, t->right, vals);
        Lifted v = fold(sub, arrays);
        return v.unknown ? sub : dst.arena.constant(v.value);
    }
    if (t->token == STCSP_T_EQ_CON && t->right && t->right->token == STCSP_T_AT) {
        // constraintNodeTranslateAT (:484-505)
        int x = t->left->var, y = t->right->left->var, k = t->right->right->num;
        Tree *r = (k == 1) ? dst.arena.make(STCSP_T_FIRST, 0, -1, -1, nullptr, dst.arena.variable(y))
                           : dst.arena.make(STCSP_T_AT, 0, -1, -1, dst.arena.variable(y), dst.arena.constant(k - 1));
        return dst.arena.make(t->token, 0, -1, -1, dst.arena.variable(x), r);
    }
    Tree *l = translate(dst, t->left, vals);
    Tree *r = translate(dst, t->right, vals);
    return dst.arena.make(t->token, t->num, t->var, t->arr, l, r);
}

int SetManager::transition(int set, const std::vector<int> &first_vals) {
    HostSet &s = *sets[set];
    if (s.self_loop) return set;
    auto it = s.trans.find(first_vals);
    if (it != s.trans.end()) return it->second;
    if (first_vals.size() != s.first_vars.size()) {
        error = "transition: wrong number of captured values";
        return STCSP_E_INTERNAL;
    }
    std::map<int, int> vals;
    for (size_t i = 0; i < first_vals.size(); i++) vals[s.first_vars[i]] = first_vals[i];
    std::unique_ptr<HostSet> ns(new HostSet());
    std::vector<Tree *> roots;
    for (auto &c : s.cons) {
        Tree *t = translate(*ns, c.root, vals);
        if (!is_tautology(t, arrays)) roots.push_back(t);
    }
    for (Tree *t : roots) {
        int rc = push_constraint(*ns, t);
        if (rc != STCSP_OK) return rc;
    }
    int found = -1;  // seenConstraints lookup (solveralgorithm.cpp:786-800): structural equality, found through a content hash
    {
        auto range = set_by_hash.equal_range(set_content_hash(*ns));
        for (auto it2 = range.first; it2 != range.second; ++it2)
            if (set_eq(*ns, *sets[it2->second]) && (found < 0 || it2->second < found)) found = it2->second;
    }
    if (found < 0) {
        found = register_set(std::move(ns));
        if (found < 0) return found;
    }
    sets[set]->trans[first_vals] = found;  // (s may dangle after register_set: re-index)
    return found;
}

int SetManager::pretranslate(long long max_tuples, int max_sets, long long max_total_tuples, double max_seconds) {
    int added = 0;
    // max_tuples bounds ONE set's capture space; the work of the whole call is bounded too -- every set enumerates its tuples
    // even when all of them map onto known sets, so thousands of sets x tens of thousands of tuples would be hours of host time.
    // When either budget runs out the call stops where it is: the sets not finished keep direct = false (the device asks on demand).
    long long total_left = max_total_tuples > 0 ? max_total_tuples : (1ll << 62);
    const auto t_start = std::chrono::steady_clock::now();
    auto out_of_time = [&]() { return max_seconds > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > max_seconds; };
    for (size_t si = 0; si < sets.size() && (int)sets.size() < max_sets; si++) {  // sets.size() grows while we go
        if (sets[si]->self_loop) continue;
        if (total_left <= 0 || out_of_time()) break;
        const std::vector<int> fv = sets[si]->first_vars;  // (copy: sets may be re-allocated by transition())
        // Values a leaf can show for each captured variable: a `first` constraint over ONE captured variable and nothing
        // else (first giveTo < 2; first seen3 == 0) is enforced at time 0, so only the values it admits are ever
        // captured -- partialorder_14's 458,752 tuples shrink to 2.
        std::vector<std::vector<int>> cand(fv.size());
        long long full = 1, tuples = 1;
        for (size_t k = 0; k < fv.size(); k++) {
            const int v = fv[k];
            full *= (long long)ub[v] - lb[v] + 1;
            if (full > kDirectTransMax) full = kDirectTransMax + 1;
            for (int x = lb[v]; x <= ub[v]; x++) {
                bool ok = true;
                for (auto &c : sets[si]->cons) {
                    if (!c.has_first || !ok) continue;
                    std::set<int> cf;
                    collect_first_vars(c.root, false, cf);
                    if (cf.size() != 1 || *cf.begin() != v) continue;
                    std::vector<int> all;
                    collect_scope(c.root, all);  // every variable of the tree, captured or not
                    if (all.size() != 1) continue;
                    HostSet scratch;
                    std::map<int, int> vm;
                    vm[v] = x;
                    Tree *t = translate(scratch, c.root, vm);
                    std::vector<int> sc;
                    collect_scope(t, sc);
                    if (sc.empty() && !is_tautology(t, arrays)) ok = false;
                }
                if (ok) cand[k].push_back(x);
            }
            tuples *= (long long)cand[k].size();
            if (tuples > max_tuples) break;
        }
        if (tuples > max_tuples) continue;
        if (tuples > total_left) break;  // (the whole set or nothing: a half-enumerated set gains no table)
        total_left -= tuples;
        if (tuples > 0) {
            std::vector<size_t> pos(fv.size(), 0);
            std::vector<int> vals(fv.size());
            long long done = 0;
            for (;;) {
                if ((++done & 255) == 0 && out_of_time()) return added;  // (this set stays direct = false)
                for (size_t k = 0; k < fv.size(); k++) vals[k] = cand[k][pos[k]];
                // a tuple that violates one of the set's own `first` constraints is never seen at a leaf (the constraint
                // is enforced at time 0): its translation -- a set with a false constant constraint -- is not worth a set
                bool reachable = true;
                {
                    std::map<int, int> vm;
                    for (size_t k = 0; k < fv.size(); k++) vm[fv[k]] = vals[k];
                    HostSet scratch;
                    for (auto &c : sets[si]->cons) {
                        if (!c.has_first) continue;
                        Tree *t = translate(scratch, c.root, vm);
                        std::vector<int> sc;
                        collect_scope(t, sc);
                        if (sc.empty() && !is_tautology(t, arrays)) reachable = false;
                    }
                }
                if (reachable && !sets[si]->trans.count(vals)) {
                    const int ns = transition((int)si, vals);
                    if (ns < 0) return ns;
                    added++;
                    if ((int)sets.size() >= max_sets) return added;
                }
                // next tuple: the LAST captured variable runs fastest, values ascending -- the order in which the reference's
                // DFS meets the leaves of a state (first unbound variable in queue order bisected first, lower half first:
                // solver.cpp:41-53, solveralgorithm.cpp:911-939). It matters when two translations differ only in the ARRAY a
                // node indexes: constraintNodeEq ignores the array (constraint.cpp:551-561), so the reference keeps whichever
                // of the two its DFS translated first, and every later leaf runs under that one.
                size_t k = fv.size();
                while (k > 0) {
                    if (pos[k - 1] + 1 < cand[k - 1].size()) {
                        pos[k - 1]++;
                        break;
                    }
                    pos[k - 1] = 0;
                    k--;
                }
                if (k == 0) break;
            }
        }
        // every tuple a leaf can show has its transition: the device looks it up in a table indexed by the tuple (when the
        // FULL space is small enough for a table; else in the list, which then holds the reachable tuples only)
        sets[si]->direct = full <= kDirectTransMax;
    }
    return added;
}

std::vector<int32_t> SetManager::serialise_set(int set) const {
    std::vector<int32_t> out;
    const HostSet &s = *sets[set];
    out.push_back(s.tag);
    out.push_back((int32_t)s.cons.size());
    for (auto &c : s.cons) serialise_tree(c.root, out);
    return out;
}

int SetManager::import_set(const int32_t *words, size_t n) {
    if (n < 2) return STCSP_E_INVALID;
    int32_t tag = words[0];
    int ncons = words[1];
    const int32_t *p = words + 2, *end = words + n;
    std::unique_ptr<HostSet> ns(new HostSet());
    for (int i = 0; i < ncons; i++) {
        Tree *t = deserialise_tree(p, end, ns->arena);
        int rc = push_constraint(*ns, t);
        if (rc != STCSP_OK) return rc;
    }
    // a known tag must stand for the same content here as on the sending shard
    const int found = find_tag(tag);
    if (found >= 0) {
        if (set_eq(*ns, *sets[found])) return found;
        error = "constraint-set tag " + std::to_string(tag) + " names different sets on two shards";
        return STCSP_E_INTERNAL;
    }
    for (size_t i = 0; i < sets.size(); i++)
        if (set_eq(*ns, *sets[i])) {
            error = "constraint set known here under tag " + std::to_string(sets[i]->tag) + ", on the sending shard under " + std::to_string(tag);
            return STCSP_E_INTERNAL;
        }
    int idx = register_set(std::move(ns));
    if (idx < 0) return idx;
    if (sets[idx]->tag != tag) {
        error = "constraint-set tag mismatch between shards";
        return STCSP_E_INTERNAL;
    }
    return idx;
}

// ------------------------------------------------------------------ bytecode compiler
// Postfix program with the evaluation order of solverValidateRe (reference
// src/solveralgorithm.cpp:336-424): children before parent, left before right. Short-circuit
// forms (if / and / or / ->) are evaluated in full and SELECTED afterwards -- lanes of a
// wavefront run one instruction stream -- so when the tree contains array lookups, guard
// markers record which sub-programs the reference would really have evaluated: only a LIVE
// out-of-range lookup clears `valid` (:347-349), and once cleared every later arithmetic /
// relational node yields 0 (:396-397).
int SetManager::compile_expr(const Tree *t, const std::vector<int> &scope, bool guards, std::vector<int32_t> &code,
                             int &depth, int &max_depth, int &mask_depth) {
    if (!t) {
        error = "malformed constraint tree";
        return STCSP_E_INVALID;
    }
    auto emit = [&](int op, int arg = 0) { code.push_back((arg << 8) | op); };
    auto push = [&]() {
        depth++;
        if (depth > max_depth) max_depth = depth;
    };
    auto guarded = [&](int op_mask, int d, const Tree *sub) -> int {
        if (guards) {
            if (++mask_depth > 31) {
                error = "conditional nesting deeper than 31";
                return STCSP_E_UNSUPPORTED;
            }
            emit(op_mask, d);
        }
        int rc = compile_expr(sub, scope, guards, code, depth, max_depth, mask_depth);
        if (guards) {
            emit(OP_MASK_POP);
            mask_depth--;
        }
        return rc;
    };
    int rc;
    switch (t->token) {
        case STCSP_T_CONST:
            emit(OP_CONST);
            code.push_back(t->num);
            push();
            return STCSP_OK;
        case STCSP_T_VAR: {
            auto it = std::find(scope.begin(), scope.end(), t->var);
            emit(OP_VAR, (int)(it - scope.begin()));
            push();
            return STCSP_OK;
        }
        case STCSP_T_ARR:
            if ((rc = compile_expr(t->right, scope, guards, code, depth, max_depth, mask_depth))) return rc;
            emit(OP_ARR, t->arr);
            return STCSP_OK;
        case STCSP_T_ABS:
        case STCSP_T_NOT:
            if ((rc = compile_expr(t->right, scope, guards, code, depth, max_depth, mask_depth))) return rc;
            emit(t->token == STCSP_T_ABS ? OP_ABS : OP_NOT);
            return STCSP_OK;
        case STCSP_T_FIRST: return compile_expr(t->right, scope, guards, code, depth, max_depth, mask_depth);  // :364-365
        case STCSP_T_AT: return compile_expr(t->left, scope, guards, code, depth, max_depth, mask_depth);      // :366-367
        case STCSP_T_IF: {
            if (!t->right || t->right->token != STCSP_T_THEN) {
                error = "if without then/else";
                return STCSP_E_INVALID;
            }
            if ((rc = compile_expr(t->left, scope, guards, code, depth, max_depth, mask_depth))) return rc;
            if ((rc = guarded(OP_MASK_T, 0, t->right->left))) return rc;
            if ((rc = guarded(OP_MASK_F, 1, t->right->right))) return rc;
            emit(OP_SEL_IF);
            depth -= 2;
            return STCSP_OK;
        }
        case STCSP_T_AND:
        case STCSP_T_OR:
        case STCSP_T_IMPLY_CON: {
            if ((rc = compile_expr(t->left, scope, guards, code, depth, max_depth, mask_depth))) return rc;
            if ((rc = guarded(t->token == STCSP_T_OR ? OP_MASK_F : OP_MASK_T, 0, t->right))) return rc;
            emit(t->token == STCSP_T_AND ? OP_SEL_AND : (t->token == STCSP_T_OR ? OP_SEL_OR : OP_SEL_IMPLY));
            depth -= 1;
            return STCSP_OK;
        }
        default: break;
    }
    int op = 0;
    switch (t->token) {
        case STCSP_T_ADD: op = OP_ADD; break;
        case STCSP_T_SUB: op = OP_SUB; break;
        case STCSP_T_MUL: op = OP_MUL; break;
        case STCSP_T_DIV: op = OP_DIV; break;
        case STCSP_T_MOD: op = OP_MOD; break;
        case STCSP_T_LT_OP: case STCSP_T_LT_CON: op = OP_LT; break;
        case STCSP_T_GT_OP: case STCSP_T_GT_CON: op = OP_GT; break;
        case STCSP_T_LE_OP: case STCSP_T_LE_CON: op = OP_LE; break;
        case STCSP_T_GE_OP: case STCSP_T_GE_CON: op = OP_GE; break;
        case STCSP_T_EQ_OP: case STCSP_T_EQ_CON: op = OP_EQ; break;
        case STCSP_T_NE_OP: case STCSP_T_NE_CON: op = OP_NE; break;
        default:
            error = "expression token not supported inside a point constraint";
            return STCSP_E_UNSUPPORTED;
    }
    if ((rc = compile_expr(t->left, scope, guards, code, depth, max_depth, mask_depth))) return rc;
    if ((rc = compile_expr(t->right, scope, guards, code, depth, max_depth, mask_depth))) return rc;
    emit(op);
    depth -= 1;
    return STCSP_OK;
}

// Host evaluation of a constraint tree on one tuple (scope position -> value): the semantics of
// solverValidateRe (reference src/solveralgorithm.cpp:336-424), used to tabulate constraints.
int SetManager::eval_tree(const Tree *t, const std::vector<int> &scope, const int *vals, bool &valid) const {
    if (!t) return 0;
    switch (t->token) {
        case STCSP_T_VAR: return vals[std::find(scope.begin(), scope.end(), t->var) - scope.begin()];
        case STCSP_T_CONST: return t->num;
        case STCSP_T_ARR: {
            int i = eval_tree(t->right, scope, vals, valid);
            const std::vector<int> &a = arrays.elements[t->arr];
            if (i < 0 || i >= (int)a.size()) {
                valid = false;
                return 0;
            }
            return a[i];
        }
        case STCSP_T_ABS: {
            int v = eval_tree(t->right, scope, vals, valid);
            return v < 0 ? (int)(0u - (unsigned)v) : v;
        }
        case STCSP_T_IF:
            return eval_tree(t->left, scope, vals, valid) ? eval_tree(t->right->left, scope, vals, valid)
                                                          : eval_tree(t->right->right, scope, vals, valid);
        case STCSP_T_FIRST: return eval_tree(t->right, scope, vals, valid);
        case STCSP_T_AT: return eval_tree(t->left, scope, vals, valid);
        case STCSP_T_NOT: return eval_tree(t->right, scope, vals, valid) == 0;
        case STCSP_T_AND: return eval_tree(t->left, scope, vals, valid) ? eval_tree(t->right, scope, vals, valid) : 0;
        case STCSP_T_OR: return eval_tree(t->left, scope, vals, valid) ? 1 : eval_tree(t->right, scope, vals, valid);
        case STCSP_T_IMPLY_CON: {
            int l = eval_tree(t->left, scope, vals, valid);
            return l == 0 ? 1 : (l <= eval_tree(t->right, scope, vals, valid));
        }
        default: break;
    }
    int l = eval_tree(t->left, scope, vals, valid);
    int r = eval_tree(t->right, scope, vals, valid);
    if (!valid) return 0;
    switch (t->token) {
        case STCSP_T_LT_CON: case STCSP_T_LT_OP: return l < r;
        case STCSP_T_GT_CON: case STCSP_T_GT_OP: return l > r;
        case STCSP_T_LE_CON: case STCSP_T_LE_OP: return l <= r;
        case STCSP_T_GE_CON: case STCSP_T_GE_OP: return l >= r;
        case STCSP_T_EQ_CON: case STCSP_T_EQ_OP: return l == r;
        case STCSP_T_NE_CON: case STCSP_T_NE_OP: return l != r;
        case STCSP_T_ADD: return (int)((unsigned)l + (unsigned)r);
        case STCSP_T_SUB: return (int)((unsigned)l - (unsigned)r);
        case STCSP_T_MUL: return (int)((unsigned)l * (unsigned)r);
        case STCSP_T_DIV: return (r == 0 || (l == INT32_MIN && r == -1)) ? 0 : l / r;
        case STCSP_T_MOD: return (r == 0 || (l == INT32_MIN && r == -1)) ? 0 : l % r;
        default: return 0;
    }
}

// Tabulate a point constraint over the product of its variables' INITIAL domains.
//   small  (arity <= 4, <= kSmallMaxRows rows): one 32-bit row of allowed values of the "word"
//          variable (the widest one) per tuple of the other variables -- one lane revises it;
//   bitmap (product <= kBitmapMaxBits): one bit per tuple -- the wavefront revises it with a
//          lookup in place of interpreting the postfix program; up to kBitmapMaxBitsDevice when the
//          engine tabulates on the device (the entry is left `pending`);
//   otherwise the program is interpreted.
void SetManager::build_entry(const HostCon &c, TableEntry &e) {
    e = TableEntry();
    const int s = (int)c.scope.size();
    if (c.type != CT_POINT || s == 0) return;
    std::vector<int> size(s);
    long long product = 1;
    const long long limit = device_tabulation ? kBitmapMaxBitsDevice : kBitmapMaxBits;
    for (int j = 0; j < s; j++) {
        size[j] = ub[c.scope[j]] - lb[c.scope[j]] + 1;
        product *= size[j];
        if (product > limit) return;
    }
    std::vector<int> vals(s), bit(s, 0);
    auto holds = [&]() {
        for (int j = 0; j < s; j++) vals[j] = lb[c.scope[j]] + bit[j];
        bool valid = true;
        return eval_tree(c.root, c.scope, vals.data(), valid) != 0;
    };
    int w = 0;
    for (int j = 1; j < s; j++)
        if (size[j] > size[w]) w = j;
    long long rows = product / size[w];
    if (W == 1 && s <= 4 && rows <= kSmallMaxRows) {
        std::vector<int> others;
        for (int j = 0; j < s; j++)
            if (j != w) others.push_back(j);
        ItemDesc it{};
        it.type = IT_SMALL;
        it.arity = s;
        it.toff = 0;  // relative to the entry's words; rows are fetched four at a time (128-bit reads): the
                      // table is placed on a 4-word boundary and every run of r1 rows is padded to a multiple of 4
        it.r1 = others.size() > 0 ? size[others[0]] : 1;
        it.r2 = others.size() > 1 ? size[others[1]] : 1;
        int r3 = others.size() > 2 ? size[others[2]] : 1;
        it.aux = (int32_t)rows;  // number of table rows = r1 * r2 * r3
        // idx[] carries scope positions here; the caller turns them into block word indices
        it.idx[0] = w;
        for (size_t k = 0; k < 3; k++) it.idx[1 + k] = k < others.size() ? others[k] : -1;
        for (int b3 = 0; b3 < r3; b3++)
            for (int b2 = 0; b2 < it.r2; b2++)
                for (int b1 = 0; b1 < it.r1; b1++) {
                    if (others.size() > 0) bit[others[0]] = b1;
                    if (others.size() > 1) bit[others[1]] = b2;
                    if (others.size() > 2) bit[others[2]] = b3;
                    uint32_t row = 0;
                    for (int b0 = 0; b0 < size[w]; b0++) {
                        bit[w] = b0;
                        if (holds()) row |= 1u << b0;
                    }
                    e.words.push_back(row);
                    if (b1 == it.r1 - 1)
                        for (int pad = it.r1; pad < small_row_stride(it.r1); pad++) e.words.push_back(0u);
                }
        e.small = it;
        e.is_small = true;
        return;
    }
    long long st = 1;
    for (int j = 0; j < s; j++) {
        e.strides.push_back((int32_t)st);
        st *= size[j];
    }
    e.bitmap = true;
    e.words.assign((size_t)((product + 31) / 32), 0u);
    if (product > kBitmapMaxBits) {  // the device fills it in (engine.hip: k_tabulate), then store_tabulated()
        // k_tabulate's interpreter keeps its operand stack in 32 registers per thread: an expression that nests deeper
        // stays interpreted by the wavefront revision (whose stack spills to LDS) instead of being tabulated wrongly
        std::vector<int32_t> scratch;
        int depth = 0, max_depth = 0, mask_depth = 0;
        if (compile_expr(c.root, c.scope, tree_has_arr(c.root), scratch, depth, max_depth, mask_depth) != STCSP_OK || max_depth > kTabulateMaxStack) {
            e = TableEntry();
            return;
        }
        e.pending = true;
        return;
    }
    uint32_t *bm = e.words.data();
    for (long long t = 0; t < product; t++) {
        long long rem = t;
        for (int j = 0; j < s; j++) {
            bit[j] = (int)(rem % size[j]);
            rem /= size[j];
        }
        if (holds()) bm[t >> 5] |= 1u << (t & 31);
    }
    long long allowed = 0;
    for (uint32_t wd : e.words) allowed += __builtin_popcount(wd);
    e.n_forbidden = product - allowed <= kFewForbidden ? (int32_t)(product - allowed) : -1;
}

// ------------------------------------------------------------------ wide constraints with conditionals
// A point constraint over more variables than a tuple bitmap can hold (juggling `_nosym`: A == if B0 eq 1 then next B0 else if
// B1 eq 1 then ... over 13 variables of 7 values) would be interpreted tuple by tuple. Evaluation is pure and total here (no
// array look-ups in the tree: nothing clears `valid`; x / 0 is defined), so a conditional can be lifted out of ANY context:
//     T[if c then t else e]  ==  if c then T[t] else T[e]
// and as a constraint (non-zero = satisfied) that is the conjunction of (if c then T[t] else 1) and (if c then 1 else T[e]).
// Repeated until every branch fits a bitmap, the constraint becomes a handful of guarded branches, each over the variables of
// its guards and ONE branch body -- tabulated like any other constraint. Only the compiled program changes: the set keeps the
// constraint as written (identity, translation, serialisation), and the conjunction has the same solutions, so leaves are
// checked exactly as before; propagation on the branches is GAC per branch (sound; the search stays complete).
static Tree *clone_tree(const Tree *t, TreeArena &a, const Tree *hole = nullptr, const Tree *fill = nullptr) {
    if (!t) return nullptr;
    if (t == hole) return clone_tree(fill, a);
    Tree *l = clone_tree(t->left, a, hole, fill);
    Tree *r = clone_tree(t->right, a, hole, fill);
    return a.make(t->token, t->num, t->var, t->arr, l, r);
}
static const Tree *first_if(const Tree *t) {  // pre-order: the outermost conditional first
    if (!t) return nullptr;
    if (t->token == STCSP_T_IF && t->right && t->right->token == STCSP_T_THEN) return t;
    if (const Tree *l = first_if(t->left)) return l;
    return first_if(t->right);
}

bool SetManager::split_wide(const Tree *body, std::vector<Guard> &guards, long long limit, std::vector<Tree *> &out, int &budget) {
    std::vector<int> scope;
    for (const Guard &g : guards) collect_scope(g.cond, scope);
    collect_scope(body, scope);
    long long product = 1;
    for (int v : scope) product = std::min<long long>(product * ((long long)ub[v] - lb[v] + 1), limit + 1);
    const Tree *f = product <= split_target ? nullptr : first_if(body);
    if (!f) {
        if (product > limit) return false;
        Tree *t = clone_tree(body, piece_arena);
        for (size_t k = guards.size(); k-- > 0;) {
            Tree *one = piece_arena.constant(1);
            Tree *th = piece_arena.make(STCSP_T_THEN, 0, -1, -1, guards[k].taken ? t : one, guards[k].taken ? one : t);
            t = piece_arena.make(STCSP_T_IF, 0, -1, -1, clone_tree(guards[k].cond, piece_arena), th);
        }
        out.push_back(t);
        return true;
    }
    if (--budget < 0) return false;
    TreeArena scratch;  // the two bodies with the conditional replaced by one of its branches
    const Tree *bt = clone_tree(body, scratch, f, f->right->left);
    const Tree *be = clone_tree(body, scratch, f, f->right->right);
    guards.push_back(Guard{f->left, true});
    bool ok = split_wide(bt, guards, limit, out, budget);
    guards.back().taken = false;
    ok = ok && split_wide(be, guards, limit, out, budget);
    guards.pop_back();
    return ok;
}

const std::vector<Tree *> &SetManager::pieces_of(const HostCon &c, const std::vector<int32_t> &key) {
    auto it = piece_cache.find(key);
    if (it != piece_cache.end()) return it->second;
    std::vector<Tree *> out;
    if (split_mode > 0 && c.type == CT_POINT && !tree_has_arr(c.root) && first_if(c.root)) {
        std::vector<Guard> guards;
        int budget = 64;  // conditionals lifted per constraint (each adds one branch)
        const long long limit = device_tabulation ? kBitmapMaxBitsDevice : kBitmapMaxBits;
        split_target = split_mode >= 2 ? 0 : limit;
        if (!split_wide(c.root, guards, limit, out, budget) || out.size() < 2) out.clear();
    }
    return piece_cache.emplace(key, std::move(out)).first->second;
}

void SetManager::store_tabulated(const std::vector<int32_t> &key, const uint32_t *words, size_t n, long long product) {
    auto it = table_cache.find(key);
    if (it == table_cache.end() || it->second.words.size() != n) return;
    TableEntry &e = it->second;
    std::copy(words, words + n, e.words.begin());
    e.pending = false;
    long long allowed = 0;  // (bits beyond the product in the last word are zero: k_tabulate masks them)
    for (uint32_t wd : e.words) allowed += __builtin_popcount(wd);
    e.n_forbidden = product - allowed <= kFewForbidden ? (int32_t)(product - allowed) : -1;
}

int SetManager::compile(FlatProgram &out) {
    out = FlatProgram();
    n_split = 0;
    n_wide_conditional = 0;
    struct Placed {  // where a cached table sits in THIS program image
        int32_t off, stride_off;
    };
    std::map<std::vector<int32_t>, Placed> placed;
    std::map<std::vector<int32_t>, std::pair<int32_t, int32_t>> code_cache;
    for (size_t si = 0; si < sets.size(); si++) {
        HostSet &s = *sets[si];
        SetDesc sd{};
        sd.con_begin = (int32_t)out.cons.size();
        sd.ncons = (int32_t)s.cons.size();
        sd.cw = (sd.ncons + 31) / 32;
        if (sd.cw < 1) sd.cw = 1;
        if (sd.cw > 64) {
            error = "more than 2048 constraints in one set";
            return STCSP_E_UNSUPPORTED;
        }
        if (sd.cw > out.max_cw) out.max_cw = sd.cw;
        sd.varcons_off = (int32_t)out.varcons.size();
        out.varcons.resize(out.varcons.size() + (size_t)N * sd.cw, 0u);
        sd.self_loop = s.self_loop;
        sd.nfirst = (int32_t)s.first_vars.size();
        if (sd.nfirst > 64) {
            error = "more than 64 variables under `first` in one constraint set";
            return STCSP_E_UNSUPPORTED;
        }
        sd.first_off = (int32_t)out.firstvars.size();
        out.firstvars.insert(out.firstvars.end(), s.first_vars.begin(), s.first_vars.end());
        {
            int32_t st = 1;  // mixed radix over the captured variables, first one fastest
            for (int v : s.first_vars) {
                out.fstrides.push_back(st);
                st *= ub[v] - lb[v] + 1;
            }
        }
        long long direct_tuples = 1;
        for (int v : s.first_vars) direct_tuples = std::min<long long>(direct_tuples * ((long long)ub[v] - lb[v] + 1), kDirectTransMax + 1);
        // (all the tables of a program together stay below kDirectTransTotalMax entries: the sets beyond that keep their lists)
        if (s.direct && direct_tuples <= kDirectTransMax && (long long)out.tdirect.size() + direct_tuples <= kDirectTransTotalMax) {
            // transition table indexed by the captured tuple (-1: no leaf can have it, or not translated yet -> the host is asked)
            const long long tuples = direct_tuples;
            sd.trans_begin = (int32_t)out.tdirect.size();
            sd.trans_count = -1;
            out.tdirect.resize(out.tdirect.size() + (size_t)tuples, -1);
            for (auto &kv : s.trans) {
                long long idx = 0, stv = 1;
                for (size_t k = 0; k < s.first_vars.size(); k++) {
                    idx += (long long)(kv.first[k] - lb[s.first_vars[k]]) * stv;
                    stv *= (long long)ub[s.first_vars[k]] - lb[s.first_vars[k]] + 1;
                }
                out.tdirect[(size_t)sd.trans_begin + (size_t)idx] = kv.second;
            }
        } else {
            sd.trans_begin = (int32_t)out.trans.size();
            sd.trans_count = (int32_t)s.trans.size();
            for (auto &kv : s.trans) {
                TransDesc td{(int32_t)out.transvals.size(), kv.second};
                out.transvals.insert(out.transvals.end(), kv.first.begin(), kv.first.end());
                out.trans.push_back(td);
            }
        }
        sd.tag = s.tag;
        std::vector<ItemDesc> small_items, wave_items;
        std::vector<uint32_t> nxt((size_t)N * K * 2, 0u);  // eager X == next Y partners per block word (SetDesc::next_off)
        bool any_next = false;
        // the constraints the PROGRAM propagates: the set's own, except that a conditional constraint too wide for a bitmap
        // is replaced by its guarded branches (split_wide)
        std::vector<HostCon> pcons;
        for (const HostCon &c0 : s.cons) {
            bool split = false;
            if (split_mode > 0 && c0.type == CT_POINT && !c0.scope.empty()) {
                std::vector<int32_t> key;
                serialise_tree(c0.root, key);
                long long product = 1;
                for (int v : c0.scope) product = std::min<long long>(product * ((long long)ub[v] - lb[v] + 1), kBitmapMaxBitsDevice + 1);
                if (product > kWideConditional && first_if(c0.root)) n_wide_conditional++;
                bool try_split = product > (device_tabulation ? kBitmapMaxBitsDevice : kBitmapMaxBits);  // (no bitmap can hold it)
                if (!try_split && split_mode >= 2) {
                    auto ce = table_cache.find(key);
                    if (ce == table_cache.end()) {
                        ce = table_cache.emplace(key, TableEntry()).first;
                        build_entry(c0, ce->second);
                    }
                    try_split = !ce->second.is_small;
                }
                if (try_split) {
                    for (Tree *piece : pieces_of(c0, key)) {
                        HostCon pc;
                        pc.root = piece;
                        pc.type = CT_POINT;
                        pc.has_first = c0.has_first;
                        collect_scope(piece, pc.scope);
                        pcons.push_back(pc);
                        split = true;
                    }
                }
            }
            if (!split) pcons.push_back(c0);
            n_split += split;
        }
        sd.ncons = (int32_t)pcons.size();
        sd.cw = std::max(1, (sd.ncons + 31) / 32);
        if (sd.cw > 64) {
            error = "more than 2048 constraints in one set";
            return STCSP_E_UNSUPPORTED;
        }
        if (sd.cw > out.max_cw) out.max_cw = sd.cw;
        out.varcons.resize((size_t)sd.varcons_off + (size_t)N * sd.cw, 0u);
        for (size_t ci = 0; ci < pcons.size(); ci++) {
            HostCon &c = pcons[ci];
            ConDesc cd{};
            cd.n_forbidden = -1;
            cd.type = c.type;
            cd.npoints = c.has_first ? 1 : K;
            cd.scope_off = (int32_t)out.scope.size();
            cd.scope_len = (int32_t)c.scope.size();
            if (cd.scope_len > kMaxScope) {
                error = "constraint over more than 64 variables";
                return STCSP_E_UNSUPPORTED;
            }
            out.scope.insert(out.scope.end(), c.scope.begin(), c.scope.end());
            cd.x = c.x;
            cd.y = c.y;
            cd.until_ordinal = c.until_ordinal;
            cd.code_off = (int32_t)out.code.size();
            if (c.type == CT_POINT) {
                bool guards = tree_has_arr(c.root);
                cd.uses_valid = guards;
                // identical constraints (the same constraint in two sets) share their program
                std::vector<int32_t> key;
                serialise_tree(c.root, key);
                auto hit = code_cache.find(key);
                if (hit != code_cache.end()) {
                    cd.code_off = hit->second.first;
                    cd.code_len = hit->second.second;
                } else {
                    int depth = 0, max_depth = 0, mask_depth = 0;
                    int rc = compile_expr(c.root, c.scope, guards, out.code, depth, max_depth, mask_depth);
                    if (rc != STCSP_OK) return rc;
                    out.code.push_back(OP_END);
                    if (max_depth > out.max_stack) out.max_stack = max_depth;
                    cd.code_len = (int32_t)out.code.size() - cd.code_off;
                    code_cache.emplace(key, std::make_pair(cd.code_off, cd.code_len));
                }
            } else {
                cd.code_len = 0;
            }
            // work items of this constraint (one per enforced time point)
            const int con_abs = (int)out.cons.size();
            if (c.type == CT_NEXT) {
                for (int p = 0; p + 1 < K; p++) {
                    // an arc whose two words have no eager partner of that kind yet is kept consistent
                    // by close_next() after every change of either word; only the others become items
                    const int wx = p * N + c.x, wy = (p + 1) * N + c.y;
                    int sh = lb[c.x] - lb[c.y];
                    sh = sh > 32 ? 32 : (sh < -32 ? -32 : sh);
                    auto side_free = [&](int w, uint32_t side) {  // no eager partner on that side of word w yet, and a free entry
                        for (int k = 0; k < 2; k++) {
                            const uint32_t e = nxt[(size_t)w * 2 + k];
                            if (e && ((e >> 24) & 1u) == side) return false;
                        }
                        return nxt[(size_t)w * 2] == 0 || nxt[(size_t)w * 2 + 1] == 0;
                    };
                    if (W == 1 && side_free(wx, 0u) && side_free(wy, 1u)) {
                        auto put = [&](int w, int partner, uint32_t side) {
                            const size_t k = nxt[(size_t)w * 2] == 0 ? 0 : 1;
                            nxt[(size_t)w * 2 + k] = (uint32_t)(partner + 1) | (uint32_t)(sh + 64) << 16 | side << 24;
                        };
                        put(wx, wy, 0u);
                        put(wy, wx, 1u);
                        any_next = true;
                        continue;
                    }
                    ItemDesc it{};
                    it.type = IT_NEXT;
                    it.point = p;
                    it.con = con_abs;
                    it.arity = 2;
                    it.idx[0] = p * N + c.x;
                    it.idx[1] = (p + 1) * N + c.y;
                    it.aux = lb[c.x] - lb[c.y];
                    if (W > 1) it.aux = std::max(-128, std::min(128, it.aux));  // (a shift beyond the widest domain empties both sides anyway)
                    (W > 1 ? wave_items : small_items).push_back(it);
                }
            } else if (c.type == CT_UNTIL) {
                ItemDesc it{};
                it.type = IT_UNTIL;
                it.con = con_abs;
                it.arity = 2;
                it.idx[0] = c.x;
                it.idx[1] = c.y;
                it.aux = c.until_ordinal;
                (W > 1 ? wave_items : small_items).push_back(it);
            } else if (c.type == CT_POINT && !c.scope.empty()) {
                // identical constraints (e.g. the same constraint in two sets, or the same one at the next compile)
                // share their tables: table_cache holds the tabulated form, `placed` its position in this image
                std::vector<int32_t> key;
                serialise_tree(c.root, key);
                auto ce = table_cache.find(key);
                if (ce == table_cache.end()) {
                    ce = table_cache.emplace(key, TableEntry()).first;
                    build_entry(c, ce->second);
                }
                const TableEntry &te = ce->second;
                std::vector<ItemDesc> proto;
                const bool is_small = te.is_small;
                cd.bitmap_off = -1;
                cd.stride_off = 0;
                cd.n_forbidden = -1;
                if (te.is_small || te.bitmap) {
                    auto pl = placed.find(key);
                    if (pl == placed.end()) {
                        std::vector<uint32_t> &dst = te.is_small ? out.stables : out.tables;
                        while (dst.size() & 3) dst.push_back(0u);
                        Placed np{(int32_t)dst.size(), (int32_t)out.strides.size()};
                        dst.insert(dst.end(), te.words.begin(), te.words.end());
                        out.strides.insert(out.strides.end(), te.strides.begin(), te.strides.end());
                        pl = placed.emplace(key, np).first;
                        if (te.pending) {
                            TabulateTodo td;
                            td.key = key;
                            td.tables_off = np.off;
                            td.product = 1;
                            for (int v : c.scope) td.product *= (long long)ub[v] - lb[v] + 1;
                            td.code_off = cd.code_off;
                            td.code_len = cd.code_len;
                            td.uses_valid = cd.uses_valid;
                            td.scope.assign(c.scope.begin(), c.scope.end());
                            out.todo.push_back(td);
                        }
                    }
                    if (te.is_small) {
                        ItemDesc it = te.small;
                        it.toff = pl->second.off;
                        proto.push_back(it);
                    } else {
                        cd.bitmap_off = pl->second.off;
                        cd.stride_off = pl->second.stride_off;
                        cd.n_forbidden = te.n_forbidden;
                    }
                }
                for (int p = 0; p < cd.npoints; p++) {
                    ItemDesc it{};
                    if (is_small) {
                        it = proto[0];
                        for (int k = 0; k < 4; k++) it.idx[k] = it.idx[k] >= 0 ? p * N + c.scope[it.idx[k]] : 0;
                    } else {
                        // a wavefront-revised item carries what the revision needs of its ConDesc, so that the
                        // device reads ONE record (a single LDS round trip) before it gathers the scope
                        it.type = IT_WAVE;
                        it.arity = cd.scope_len;
                        it.idx[0] = cd.scope_off;
                        it.idx[1] = cd.bitmap_off;
                        it.idx[2] = cd.stride_off;
                        it.idx[3] = cd.n_forbidden;
                        it.toff = cd.code_off;
                        it.r1 = cd.uses_valid;
                        it.r2 = cd.code_len;
                    }
                    it.point = p;
                    it.con = con_abs;
                    (is_small ? small_items : wave_items).push_back(it);
                }
            } else {
                cd.bitmap_off = -1;
            }
            out.cons.push_back(cd);
            if (c.type != CT_AT)  // AT constraints are never revised (solveralgorithm.cpp:658-662)
                for (int v : c.scope) out.varcons[sd.varcons_off + (size_t)v * sd.cw + ci / 32] |= 1u << (ci % 32);
        }
        sd.next_off = -1;
        if (any_next) {
            sd.next_off = (int32_t)out.nextpart.size();
            out.nextpart.insert(out.nextpart.end(), nxt.begin(), nxt.end());
        }
        // items: lane-revised ones first, then the wavefront-revised ones
        sd.item_begin = (int32_t)out.items.size();
        sd.witem_begin = 0;
        for (const SetDesc &prev : out.sets) sd.witem_begin += prev.nitems - prev.nsmall;
        sd.nsmall = (int32_t)small_items.size();
        sd.nitems = (int32_t)(small_items.size() + wave_items.size());
        sd.iw = std::max(1, (sd.nitems + 31) / 32);
        if (sd.iw > 64) {
            error = "more than 2048 propagation items in one set";
            return STCSP_E_UNSUPPORTED;
        }
        if (sd.iw > out.max_iw) out.max_iw = sd.iw;
        out.items.insert(out.items.end(), small_items.begin(), small_items.end());
        out.items.insert(out.items.end(), wave_items.begin(), wave_items.end());
        sd.itemrows_off = (int32_t)out.itemrows.size();
        // A set without `first` / `@` constraints says the same about EVERY state under it, so what its constraints make of the
        // initial domains is worked out once (engine.hip fresh_init) and the new time point of a fresh state starts from that:
        // its point items have no work until an arc from the point before changes one of their words (the closure marks them).
        bool own_init = fresh_init && W == 1 && sets.size() * (size_t)N <= 4096 && s.first_vars.empty();
        for (const HostCon &c : pcons) own_init = own_init && !c.has_first && c.type != CT_AT;
        out.set_fresh_init.push_back(own_init ? 1 : 0);
        // rows [0, N*K): the items that read block word (p, v); row N*K: the items with work in a FRESH state under the same set
        // (dirty seed N*K + 1, dev_propagate.hpp process_node) -- those that read the new time point K-1, until checks (their
        // expire bits change with the state) and constraints with a `first` (enforced at point 0 only, which they now see for
        // the first time); every other item saw the same domains one point later in the leaf the state comes from
        // ... row N*K + 1: the wavefront-revised items revise_batch can take (a tuple bitmap, at most kBatchArity scope variables)
        out.itemrows.resize(out.itemrows.size() + ((size_t)N * K + 2) * sd.iw, 0u);
        for (int i = 0; i < sd.nitems; i++) {
            const ItemDesc &it = out.items[sd.item_begin + i];
            auto mark = [&](int word) { out.itemrows[sd.itemrows_off + (size_t)word * sd.iw + i / 32] |= 1u << (i % 32); };
            bool fresh;
            if (it.type == IT_NEXT || it.type == IT_UNTIL) {
                mark(it.idx[0]);
                mark(it.idx[1]);
                fresh = it.type == IT_UNTIL || it.point + 1 == K - 1;
            } else {
                const ConDesc &cd = out.cons[it.con];
                for (int j = 0; j < cd.scope_len; j++) mark(it.point * N + out.scope[cd.scope_off + j]);
                fresh = (it.point == K - 1 && !own_init) || cd.npoints < K;
            }
            if (fresh) mark(N * K);
            if (i >= sd.nsmall && it.type == IT_WAVE && it.idx[1] >= 0 && it.arity <= kBatchArity) mark(N * K + 1);
        }
        out.sets.push_back(sd);
    }
    return STCSP_OK;
}

}  // namespace stcsp
