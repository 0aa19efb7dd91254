/*
 * stcsp_host.h -- host-side plumbing around the engine seam (plain C++ behind a C-ABI so the
 * Python tests, bench.py and the `stcsp` CLI share one implementation).
 *
 *   front end  : .csp text -> built model (stcsp_problem)
 *                replaces src/stcsp.l + src/stcsp.y:56-174 (hand-written: no lex/yacc in the
 *                image) and src/solver.cpp:138-159 solverParse +
 *                src/solveralgorithm.cpp:16-332 solverAddConstr / constraintNormalise.
 *   post-proc  : raw automaton -> pruned, renumbered automaton -> solutions.dot
 *                replaces src/graph.cpp:357-442 (graphTraverse, renumberVertex),
 *                src/graph.cpp:167-355 (adversarialTraverse{,2}) and
 *                src/solveralgorithm.cpp:709-730 + src/graph.cpp:41-101,145-154 (dot writer).
 *
 * None of this is on the GPU hot path (SURVEY.md section 8 marks it plumbing / "next").
 */
#ifndef STCSP_HOST_H
#define STCSP_HOST_H

#include <stddef.h>
#include "stcsp_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- front end ---------------- */
typedef struct stcsp_model stcsp_model;

/* Parse + normalise + flatten. prefix_k <= 0 selects the reference default 2
 * (src/solver.cpp:204). On error returns a negative code; stcsp_host_last_error() has the
 * message the reference would have logged / printed (e.g. "Line 3: syntax error"). */
int stcsp_model_load_file(const char *path, int prefix_k, stcsp_model **out);
int stcsp_model_load_text(const char *text, int prefix_k, stcsp_model **out);
const stcsp_problem *stcsp_model_problem(const stcsp_model *model);
void stcsp_model_free(stcsp_model *model);
const char *stcsp_host_last_error(void);

/* Pretty-print constraint `index` of the model the way constraintNodeLogPrint does
 * (src/constraint.cpp:91-154), for diagnostics. Returns a malloc'd string (stcsp_host_free). */
char *stcsp_model_constraint_string(const stcsp_model *model, int index);

/* ---------------- automaton post-processing ---------------- */
typedef struct stcsp_automaton stcsp_automaton;

/* Copies the raw result (so the engine may be destroyed afterwards). */
int stcsp_automaton_build(const stcsp_problem *problem, const stcsp_result *result, stcsp_automaton **out);
void stcsp_automaton_free(stcsp_automaton *a);

/* graphTraverse (src/graph.cpp:357-418): final flags from the until part of the signature,
 * valid = can reach a final state, edges into invalid states deleted. */
int stcsp_automaton_traverse(stcsp_automaton *a);
/* adversarialTraverse (src/graph.cpp:304-355); the reference hard-codes var_index = 5.
 * Returns root->valid (0/1) or a negative error. */
int stcsp_automaton_adversarial(stcsp_automaton *a, int var_index);
/* adversarialTraverse2 (src/graph.cpp:247-302); the reference hard-codes opponent=5, avatar=6. */
int stcsp_automaton_adversarial2(stcsp_automaton *a, int opponent_index, int avatar_index);
/* Take the flags computed on the device by stcsp_engine_postprocess() (stcsp_engine.h) instead of
 * running the three passes above on the host: valid / final per state, alive per edge, indexed like
 * the stcsp_result the automaton was built from. */
int stcsp_automaton_import_flags(stcsp_automaton *a, const uint8_t *state_valid, const uint8_t *state_final,
                                 const uint8_t *edge_alive);
/* Read the current flags back (any pointer may be NULL): [n_states], [n_states], [n_edges]. */
int stcsp_automaton_flags(const stcsp_automaton *a, uint8_t *state_valid, uint8_t *state_final, uint8_t *edge_alive);
/* Make the output order independent of the search's scheduling: out-edges of every state ordered by
 * content (destination buckets by smallest label, edges by label). Call before renumber / write_dot /
 * write_binary when the files must be reproducible byte for byte; O(E log E) label comparisons. */
int stcsp_automaton_order_by_label(stcsp_automaton *a);
/* renumberVertex (src/graph.cpp:420-442). */
int stcsp_automaton_renumber(stcsp_automaton *a);

/* solverOut (src/solveralgorithm.cpp:709-730). Same line formats; the body order follows this
 * implementation's own deterministic DFS (byte-identical order would need
 * __gnu_cxx::hash_map iteration order: SURVEY.md parity level L3, not a goal). */
int stcsp_automaton_write_dot(const stcsp_automaton *a, const char *path);

/* Compact binary form of the same printed automaton (layout: postproc.cpp, "STCSPAUT" v1):
 * ~(4 + n_vars) bytes per edge instead of ~4*n_vars of text -- the exchange format for consumers
 * that do not need Graphviz. read_binary gives an automaton on which write_dot / canonical /
 * num_* work (traverse / adversarial passes have already been applied to what it holds). */
int stcsp_automaton_write_binary(const stcsp_automaton *a, const char *path);
int stcsp_automaton_read_binary(const char *path, stcsp_automaton **out);

/* Canonical text of the automaton: BFS from the root over edges
 * sorted by label, states renumbered in discovery order, constraint-set ids renumbered by
 * first appearance. sha256 of this text is the parity object. malloc'd; stcsp_host_free. */
char *stcsp_automaton_canonical(const stcsp_automaton *a, size_t *len);

int64_t stcsp_automaton_num_states(const stcsp_automaton *a);      /* table size (incl. failed) */
int64_t stcsp_automaton_num_live_states(const stcsp_automaton *a); /* reachable + printed        */
int64_t stcsp_automaton_num_live_edges(const stcsp_automaton *a);

/* Merge the per-shard results of a sharded run (global state ids, see stcsp_engine.h) into
 * one result with dense ids; runs the ok-fixpoint over the union. The merged result is owned
 * by the returned handle. */
typedef struct stcsp_merged stcsp_merged;
int stcsp_merge_shards(const stcsp_result *const *shards, int n_shards, stcsp_merged **out);
const stcsp_result *stcsp_merged_result(const stcsp_merged *m);
void stcsp_merged_free(stcsp_merged *m);

void stcsp_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* STCSP_HOST_H */
