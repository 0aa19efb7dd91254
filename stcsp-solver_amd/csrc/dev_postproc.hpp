// dev_postproc.hpp -- the reference's post-search graph passes as HBM-bound sweeps over the
// compacted automaton that the device export leaves in HBM.
//
//   src/graph.cpp:357-418   graphTraverse         final / valid flags, backward reachability
//                                                 from the final states, drop edges into invalid
//   src/graph.cpp:304-355   adversarialTraverse   + checkVertexOutEdge  (167-189)
//   src/graph.cpp:247-302   adversarialTraverse2  + checkVertexOutEdge2 (193-244)
//
// The reference drives each pass from a pointer-ordered std::set worklist over per-vertex
// hash_map<int, slist<Edge*>>; every pass is a monotone fixpoint (flags only ever go
// valid -> invalid, or invalid -> valid in graphTraverse), so the flags at the fixpoint do not
// depend on the visiting order. Here each round is one coalesced sweep over the structure-of-
// arrays edge list (src, dst: 8 B each; one 4 B label word) plus one sweep over the states; the
// host only reads the `changed` word between rounds. The host twin (postproc.cpp) is the checker
// in tests/test_postproc_gpu.py.
//
// Layout: states are table indices (root = 0); edge e = (src[e], dst[e], values[e*N .. e*N+N)).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace stcsp {
namespace dev {

// final[v] = every `until` flag of the signature is 1 (graph.cpp:364-379); the root's flag is
// given (no UNTIL constraint in the model, solveralgorithm.cpp:956-964). valid starts as final.
__global__ void k_trav_init(uint32_t n_states, const uint32_t *keys, int KL, int c0, int c1, int root_final, uint8_t *valid,
                            uint8_t *fin) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_states) return;
    uint8_t f = 1;
    if (s == 0) {
        f = (uint8_t)(root_final != 0);
    } else {
        for (int c = c0; c < c1; c++)
            if (keys[(size_t)s * KL + 1 + c] != 1u) f = 0;
    }
    fin[s] = f;
    valid[s] = f;
}

// One backward-reachability sweep (graph.cpp:381-402): a live edge into a valid state makes
// its source valid. Flags are read live, so a sweep can carry validity across several edges.
__global__ void k_trav_back(uint32_t E, const long long *src, const long long *dst, const uint8_t *alive, uint8_t *valid,
                            uint32_t *changed) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !alive[e]) return;
    const long long u = src[e], v = dst[e];
    if (valid[v] && !valid[u]) {
        valid[u] = 1;
        *changed = 1u;
    }
}

// Drop the edges from a kept state into an invalid one. graphTraverse also walks the root's
// edges when the root itself is not valid (graph.cpp:404-417: `root_rule`); the adversarial
// passes only look at valid sources (graph.cpp:343-353, 290-300).
__global__ void k_kill_into_invalid(uint32_t E, const long long *src, const long long *dst, uint8_t *alive, const uint8_t *valid,
                                    int root_rule) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !alive[e]) return;
    const long long u = src[e];
    if ((valid[u] || (root_rule && u == 0)) && !valid[dst[e]]) alive[e] = 0;
}

// adversarialTraverse: cover[v] = set of values of variable `var` on live edges into valid states.
__global__ void k_adv_cover(uint32_t E, const long long *src, const long long *dst, const int32_t *values, int N, int var, int lb,
                            const uint8_t *alive, const uint8_t *valid, uint32_t *cover) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !alive[e] || !valid[dst[e]]) return;
    atomicOr(&cover[src[e]], 1u << (values[(size_t)e * N + var] - lb));
}
// checkVertexOutEdge: a state stays valid only if every value of the variable is offered.
__global__ void k_adv_check(uint32_t n_states, const uint32_t *cover, uint32_t full, uint8_t *valid, uint32_t *changed) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_states) return;
    if (valid[s] && cover[s] != full) {
        valid[s] = 0;
        *changed = 1u;
    }
}

// adversarialTraverse2: cover2[v][a] = set of values of `op` seen together with value a of `ava`.
__global__ void k_adv2_cover(uint32_t E, const long long *src, const long long *dst, const int32_t *values, int N, int op, int ava,
                             int lb_op, int lb_ava, int wa, const uint8_t *alive, const uint8_t *valid, uint32_t *cover2) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !alive[e] || !valid[dst[e]]) return;
    const int32_t *row = values + (size_t)e * N;
    atomicOr(&cover2[(size_t)src[e] * wa + (row[ava] - lb_ava)], 1u << (row[op] - lb_op));
}
// checkVertexOutEdge2, first half: the state is kept iff some value of `ava` sees every value of `op`.
__global__ void k_adv2_check(uint32_t n_states, const uint32_t *cover2, int wa, uint32_t full, uint8_t *valid, uint8_t *node_ok,
                             uint32_t *changed) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_states) return;
    bool any = false;
    for (int a = 0; a < wa; a++) any |= cover2[(size_t)s * wa + a] == full;
    node_ok[s] = (uint8_t)any;
    if (!any && valid[s]) {
        valid[s] = 0;
        *changed = 1u;
    }
}
// second half: in a kept state the edges of the incomplete `ava` classes are removed (graph.cpp:231-241).
__global__ void k_adv2_kill(uint32_t E, const long long *src, const long long *dst, const int32_t *values, int N, int ava, int lb_ava,
                            int wa, uint32_t full, uint8_t *alive, const uint8_t *valid, const uint8_t *node_ok,
                            const uint32_t *cover2) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E || !alive[e] || !valid[dst[e]]) return;
    const long long u = src[e];
    if (node_ok[u] && cover2[(size_t)u * wa + (values[(size_t)e * N + ava] - lb_ava)] != full) alive[e] = 0;
}

}  // namespace dev
}  // namespace stcsp
