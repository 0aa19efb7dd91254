// cvalue.hpp -- constant folding of constraint trees ("lifted int" evaluation) and the
// tautology test built on it. Behavioural mirror of constraintNodeValue /
// constraintNodeTautology (reference src/constraint.cpp:335-462), including its two
// deterministic quirks, because the results decide which constraints survive a `first`
// translation and therefore the constraint-set ids printed in the automaton:
//   * `gt` folds with `<`            (src/constraint.cpp:425)
//   * `or` folds to 1 when left != 1 (src/constraint.cpp:400-414)
// Cases where the reference reads uninitialised memory (tokens with no switch arm, e.g. a
// `->` or `@` under `first`) fold to 0 here; out-of-range constant array indices fold to 0
// and x/0, x%0 fold to 0 (the reference would fault).
#pragma once
#include <cstdlib>
#include <vector>
#include "tree.hpp"

namespace stcsp {

struct Lifted {
    bool unknown;  // LiftedInt::tag
    int value;
};

struct ArrayTable {
    std::vector<std::vector<int>> elements;
};

inline Lifted fold(const Tree *t, const ArrayTable &arrays) {
    Lifted res{false, 0};
    if (!t) return Lifted{true, 0};
    switch (t->token) {
        case STCSP_T_VAR: return Lifted{true, 0};
        case STCSP_T_CONST: return Lifted{false, t->num};
        case STCSP_T_FIRST: return fold(t->right, arrays);
        case STCSP_T_NEXT: return Lifted{true, 0};
        case STCSP_T_ARR: {
            Lifted r = fold(t->right, arrays);
            if (r.unknown) return r;
            const std::vector<int> &e = arrays.elements[t->arr];
            res.value = (r.value >= 0 && r.value < (int)e.size()) ? e[r.value] : 0;
            return res;
        }
        case STCSP_T_ABS: {
            Lifted r = fold(t->right, arrays);
            if (r.unknown) return r;
            res.value = std::abs(r.value);
            return res;
        }
        case STCSP_T_IF: {
            Lifted c = fold(t->left, arrays);
            if (c.unknown) return c;
            return c.value ? fold(t->right->left, arrays) : fold(t->right->right, arrays);
        }
        case STCSP_T_NOT: {
            Lifted r = fold(t->right, arrays);
            if (r.unknown) return r;
            res.value = (r.value == 0) ? 1 : 0;
            return res;
        }
        case STCSP_T_AND: {
            Lifted l = fold(t->left, arrays);
            if (l.unknown) return l;
            if (l.value == 0) return Lifted{false, 0};
            return fold(t->right, arrays);
        }
        case STCSP_T_OR: {
            Lifted l = fold(t->left, arrays);
            if (l.unknown) return l;
            if (l.value != 1) return Lifted{false, 1};  // reference quirk, see header
            return fold(t->right, arrays);
        }
        default: break;
    }
    Lifted l = fold(t->left, arrays);
    Lifted r = fold(t->right, arrays);
    if (l.unknown || r.unknown) return Lifted{true, 0};
    long long a = l.value, b = r.value;
    switch (t->token) {
        case STCSP_T_LT_OP: res.value = a < b; break;
        case STCSP_T_GT_OP: res.value = a < b; break;  // reference quirk, see header
        case STCSP_T_LE_OP: res.value = a <= b; break;
        case STCSP_T_GE_OP: res.value = a >= b; break;
        case STCSP_T_EQ_OP: res.value = a == b; break;
        case STCSP_T_NE_OP: res.value = a != b; break;
        case STCSP_T_ADD: res.value = (int)(uint32_t)(a + b); break;
        case STCSP_T_SUB: res.value = (int)(uint32_t)(a - b); break;
        case STCSP_T_MUL: res.value = (int)(uint32_t)((uint64_t)a * (uint64_t)b); break;
        case STCSP_T_DIV: res.value = (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : (int)(a / b); break;
        case STCSP_T_MOD: res.value = (b == 0 || (a == INT32_MIN && b == -1)) ? 0 : (int)(a % b); break;
        default: res.value = 0; break;
    }
    return res;
}

inline bool is_tautology(const Tree *root, const ArrayTable &arrays) {
    Lifted l = fold(root->left, arrays);
    Lifted r = fold(root->right, arrays);
    if (l.unknown || r.unknown) return false;
    switch (root->token) {
        case STCSP_T_LT_CON: return l.value < r.value;
        case STCSP_T_GT_CON: return l.value > r.value;
        case STCSP_T_LE_CON: return l.value <= r.value;
        case STCSP_T_GE_CON: return l.value >= r.value;
        case STCSP_T_EQ_CON: return l.value == r.value;
        case STCSP_T_NE_CON: return l.value != r.value;
        case STCSP_T_IMPLY_CON: return l.value <= r.value;
        case STCSP_T_UNTIL_CON: return r.value == 1;
        default: return false;
    }
}

}  // namespace stcsp
