// tree.hpp -- host-side constraint tree (the role of ConstraintNode, reference
// src/constraint.h:24-31) shared by the front end and the engine's constraint-set manager.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "stcsp_engine.h"

namespace stcsp {

struct Tree {
    int token = 0;
    int num = 0;
    int var = -1;
    int arr = -1;
    Tree *left = nullptr;
    Tree *right = nullptr;
};

// Bump arena: trees are created in bulk (per constraint set) and dropped together.
class TreeArena {
public:
    Tree *make(int token, int num = 0, int var = -1, int arr = -1, Tree *l = nullptr, Tree *r = nullptr) {
        pool_.emplace_back(new Tree{token, num, var, arr, l, r});
        return pool_.back().get();
    }
    Tree *constant(int v) { return make(STCSP_T_CONST, v); }
    Tree *variable(int v) { return make(STCSP_T_VAR, 0, v); }
    size_t size() const { return pool_.size(); }

private:
    std::vector<std::unique_ptr<Tree>> pool_;
};

inline bool is_constraint_root(int t) { return t >= STCSP_T_LT_CON && t <= STCSP_T_UNTIL_CON; }

// Rebuild pointer trees from the flat ABI form.
inline Tree *unflatten(const stcsp_problem *p, int idx, TreeArena &arena) {
    if (idx < 0) return nullptr;
    const stcsp_node &n = p->nodes[idx];
    return arena.make(n.token, n.num, n.var, n.arr, unflatten(p, n.left, arena), unflatten(p, n.right, arena));
}

inline int flatten(const Tree *t, std::vector<stcsp_node> &out) {
    if (!t) return -1;
    int me = (int)out.size();
    out.push_back(stcsp_node{t->token, t->num, t->var, t->arr, -1, -1});
    int l = flatten(t->left, out);
    int r = flatten(t->right, out);
    out[me].left = l;
    out[me].right = r;
    return me;
}

}  // namespace stcsp
