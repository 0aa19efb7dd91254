// okfix.hpp -- the ok/fail bookkeeping of the reference's DFS restated as an order-independent
// fixpoint over the raw leaf-edge log.
//
// Reference (src/solveralgorithm.cpp:857-874, 904-909): a leaf's edge is kept iff the
// destination state is "ok"; a state is marked fail iff no leaf below it was ok; a known,
// not-failed destination (even one still open on the DFS stack) counts as ok. That is the
// greatest fixpoint "ok(v) <=> v has an out-edge to an ok state". A frontier search logs EVERY
// leaf edge and then deletes, until nothing changes, every non-root state without a live
// out-edge together with the edges into it. Kept edges == reference edges, deleted states ==
// reference Vertex::fail states. The root is never marked (the reference ignores the root's
// return value, solveralgorithm.cpp:967-971).
#pragma once
#include <cstdint>
#include <vector>

namespace stcsp {

// edges are (src,dst) over dense state ids [0,n_states); on return fail[v] in {0,1} and
// alive[e] in {0,1}.
inline void ok_fixpoint(int64_t n_states, const std::vector<int64_t> &src, const std::vector<int64_t> &dst,
                        std::vector<uint8_t> &fail, std::vector<uint8_t> &alive) {
    const int64_t E = (int64_t)src.size();
    fail.assign((size_t)n_states, 0);
    alive.assign((size_t)E, 1);
    std::vector<int64_t> outdeg((size_t)n_states, 0), in_off((size_t)n_states + 1, 0), in_edge((size_t)E);
    for (int64_t e = 0; e < E; e++) {
        outdeg[src[e]]++;
        in_off[dst[e] + 1]++;
    }
    for (int64_t v = 0; v < n_states; v++) in_off[v + 1] += in_off[v];
    {
        std::vector<int64_t> cur(in_off.begin(), in_off.end() - 1);
        for (int64_t e = 0; e < E; e++) in_edge[cur[dst[e]]++] = e;
    }
    std::vector<int64_t> work;
    for (int64_t v = 1; v < n_states; v++)
        if (outdeg[v] == 0) {
            fail[v] = 1;
            work.push_back(v);
        }
    while (!work.empty()) {
        int64_t v = work.back();
        work.pop_back();
        for (int64_t i = in_off[v]; i < in_off[v + 1]; i++) {
            int64_t e = in_edge[i];
            if (!alive[e]) continue;
            alive[e] = 0;
            int64_t u = src[e];
            if (--outdeg[u] == 0 && u != 0 && !fail[u]) {
                fail[u] = 1;
                work.push_back(u);
            }
        }
    }
}

}  // namespace stcsp
