/*
 * stcsp_engine.h -- C-ABI of the MI355X stream-CSP propagation + search engine.
 *
 * This is the drop-in boundary for the reference's
 *     double solverSolve(Solver *solver, bool testing)
 * (reference: src/solveralgorithm.h:11, defined src/solveralgorithm.cpp:945-1005; the
 * narrowest cut is lines 966-971 = "levelUp; if (GAC) solverSolveRe(root) else numFails++").
 *
 * What crosses the boundary is exactly what solverSolve reads from / leaves in `Solver`
 * (src/solver.h:22-49), flattened to plain-old-data:
 *   in : varQueue (lb/ub, order = branching order = edge-label order), arrayQueue,
 *        constrQueue (the normalised constraint trees, in queue order), prefixK
 *   out: the automaton the search leaves in solver->graph (src/graph.h:47-72): the state
 *        table (signature, constraint-set id, fail flag), the labelled edges, and the
 *        counters numFails / numDominance / numNodes (src/solver.h:28-39).
 * Everything else in solverSolve (graphTraverse, adversarial passes, renumberVertex,
 * solutions.dot, the stats line; lines 972-1005) stays on the host side of this ABI
 * (see stcsp_host.h).
 *
 * Plain C types only. Every function returns 0 on success or a negative STCSP_E_* code;
 * the library never calls exit() (the reference exit(1)s on every error, e.g.
 * src/solver.cpp:33-36).
 */
#ifndef STCSP_ENGINE_H
#define STCSP_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- expression-tree tokens (mirror of the yacc tokens used in ConstraintNode::token,
 *      src/constraint.h:24-31; numbering is this ABI's own, not y.tab.h's) ---- */
enum stcsp_token {
    STCSP_T_CONST = 1, /* CONSTANT            num = value                       */
    STCSP_T_VAR,       /* IDENTIFIER          var = variable index              */
    STCSP_T_ARR,       /* ARR_IDENTIFIER      arr = array index, right = index  */
    STCSP_T_FIRST,     /* first e             right = e                         */
    STCSP_T_NEXT,      /* next  e             right = e                         */
    STCSP_T_FBY,       /* a fby b             (never present after normalise)   */
    STCSP_T_AT,        /* e @ k               left = e, right = CONST k         */
    STCSP_T_ABS,       /* abs e               right = e                         */
    STCSP_T_NOT,       /* not e               right = e                         */
    STCSP_T_IF,        /* if c then a else b  left = c, right = THEN(a, b)      */
    STCSP_T_THEN,
    STCSP_T_AND,
    STCSP_T_OR,
    STCSP_T_ADD,
    STCSP_T_SUB,
    STCSP_T_MUL,
    STCSP_T_DIV,
    STCSP_T_MOD,
    STCSP_T_LT_OP, /* lt gt le ge eq ne : expression-level comparisons */
    STCSP_T_GT_OP,
    STCSP_T_LE_OP,
    STCSP_T_GE_OP,
    STCSP_T_EQ_OP,
    STCSP_T_NE_OP,
    STCSP_T_LT_CON, /* <  >  <=  >=  ==  !=  ->  until : constraint roots */
    STCSP_T_GT_CON,
    STCSP_T_LE_CON,
    STCSP_T_GE_CON,
    STCSP_T_EQ_CON,
    STCSP_T_NE_CON,
    STCSP_T_IMPLY_CON,
    STCSP_T_UNTIL_CON
};

/* One node of a flattened constraint tree (ConstraintNode, src/constraint.h:24-31).
 * left/right index into stcsp_problem::nodes, -1 = NULL. */
typedef struct stcsp_node {
    int32_t token;
    int32_t num;
    int32_t var;   /* -1 unless token == STCSP_T_VAR */
    int32_t arr;   /* -1 unless token == STCSP_T_ARR */
    int32_t left;
    int32_t right;
} stcsp_node;

/* The built model, as solverSolve finds it (src/solver.h:22-49). */
typedef struct stcsp_problem {
    int32_t n_vars;               /* varQueue->size(), aux vars (_V%d) included, in queue order */
    int32_t prefix_k;             /* Solver::prefixK (-k, default 2)                           */
    const int32_t *var_lb;        /* [n_vars] Variable::lb  (src/variable.h:13)                */
    const int32_t *var_ub;        /* [n_vars] Variable::ub                                      */
    const char *const *var_names; /* [n_vars] may be NULL (only the dot writer needs names)    */
    int32_t n_arrays;             /* arrayQueue->size()                                         */
    const int32_t *array_off;     /* [n_arrays + 1] offsets into array_data                     */
    const int32_t *array_data;    /* Array::elements, concatenated (src/variable.h:54-59)       */
    int32_t n_nodes;
    const stcsp_node *nodes;         /* all constraint trees                                    */
    int32_t n_constraints;           /* constrQueue->size()                                     */
    const int32_t *constraint_root;  /* [n_constraints] root node of each, in constrQueue order */
} stcsp_problem;

typedef struct stcsp_options {
    int32_t device;           /* HIP device ordinal (ignored by the CPU oracle)                    */
    int32_t rank;             /* this shard (0 when not sharded)                                   */
    int32_t world;            /* number of shards; states are owned by hash(key) % world          */
    int32_t batch_nodes;      /* max open search nodes expanded per kernel launch (0 = default)    */
    int64_t max_search_nodes; /* stop after this many node expansions (0 = unlimited)              */
    double time_limit_s;      /* stop after this many seconds of search (0 = unlimited); result is
                                 then partial and stcsp_result::truncated is set                   */
    int32_t flags;            /* STCSP_F_*                                                         */
    int32_t reserved;
} stcsp_options;

#define STCSP_F_KEEP_RAW_EDGES 1 /* also keep edges into failed states in the result (debug)       */
#define STCSP_F_NO_EXPORT 2      /* solve() leaves the automaton on the device; call
                                    stcsp_engine_export() to copy it out (bench: HBM-resident)     */
#define STCSP_F_PROFILE 4        /* bracket every k_expand launch with HIP events (roofline)       */
#define STCSP_F_STEPPED 8        /* run the sharded pipeline (leaves emit successor candidates, the
                                    owner commits them: begin/expand_local/outbox/commit/finish) even
                                    with world == 1 -- measures/tests that pipeline on a single GPU   */

typedef struct stcsp_counters {
    int64_t search_nodes; /* node expansions = propagation-to-fixpoint + classification; the
                             reference's unit is one solverSolveRe call (solveralgorithm.cpp:733) */
    int64_t gac_calls;    /* generalisedArcConsistent calls (solveralgorithm.cpp:617)             */
    int64_t fails;        /* Solver::numFails  (solveralgorithm.cpp:862,923,936,970)              */
    int64_t dominance;    /* Solver::numDominance (solveralgorithm.cpp:871)                        */
    int64_t leaves;       /* leaf cases reached (solveralgorithm.cpp:739)                          */
    int64_t revisions;    /* arc (reference) / constraint-point (engine) revisions                 */
    int64_t evaluations;  /* constraint-tuple evaluations (validate calls, solveralgorithm.cpp:428)*/
    int64_t levels;       /* engine only: kernel launch rounds                                     */
    double seconds_search;/* wall time of the search phase (automaton resident on device / in RAM) */
    double seconds_export;/* wall time of copying the automaton out + ok-fixpoint                  */
    /* engine only, with STCSP_F_PROFILE: HIP-event time of the dominant kernel (k_expand), summed
       over its launches, on the stream it is launched on */
    double seconds_expand_kernel;
    int64_t expand_launches;
    int64_t wave_revisions; /* engine only: revisions done by a whole wavefront (bitmap / bytecode) */
    int64_t sweeps;         /* engine only: lane-per-item sweeps over the small constraints         */
    int64_t skipped_revisions; /* engine only: revisions skipped because the product to refute
                                  exceeded the per-revision budget (sound, see engine.hip)          */
    int64_t translation_stops; /* engine only: times the device stopped for the host to translate a
                                  constraint set (constraintTranslate, constraint.cpp:540-548)       */
} stcsp_counters;

/* The automaton as the search leaves it in solver->graph, before graphTraverse.
 * State 0 is the root (Signature({},0), solveralgorithm.cpp:951-954); its signature row is
 * all zeros and is not meaningful. Arrays are owned by the engine and stay valid until the
 * next solve()/export() on it or stcsp_engine_destroy(). */
typedef struct stcsp_result {
    int64_t n_states;           /* vertexTable->size(): failed states included                    */
    int32_t sig_len;            /* numSignVar + (#UNTIL constraints)                               */
    int32_t n_sig_vars;         /* Solver::numSignVar                                              */
    int32_t n_until;            /* Solver::numUntil (distinct right-hand vars of `until`)          */
    int32_t n_until_cons;       /* number of UNTIL constraints (one signature flag each)           */
    const int32_t *state_cid;   /* [n_states] Signature::constraintID                              */
    const int32_t *state_sig;   /* [n_states * sig_len] Signature::sigValues                       */
    const uint8_t *state_fail;  /* [n_states] Vertex::fail                                         */
    int64_t n_edges;            /* edges whose destination is not failed (== reference edges)      */
    const int64_t *edge_src;    /* [n_edges] state index                                           */
    const int64_t *edge_dst;    /* [n_edges]                                                       */
    const int32_t *edge_values; /* [n_edges * n_vars] Edge::values: time-0 value of EVERY variable */
    int32_t n_vars;
    int32_t n_constraint_sets;  /* seenConstraints->size()                                         */
    const uint8_t *var_is_signature; /* [n_vars] Variable::isSignature after classification        */
    int32_t root_final;         /* solveralgorithm.cpp:956-964: no UNTIL constraint in the model   */
    int32_t truncated;          /* 1 if a node/time limit stopped the search early                 */
    stcsp_counters counters;
} stcsp_result;

enum stcsp_error {
    STCSP_OK = 0,
    STCSP_E_INVALID = -1,      /* malformed problem descriptor                                     */
    STCSP_E_UNSUPPORTED = -2,  /* feature outside the bitset path (e.g. a domain wider than the
                                  engine's word budget: aux vars of / and % get [INT_MIN,INT_MAX],
                                  solveralgorithm.cpp:316-322)                                     */
    STCSP_E_DEVICE = -3,       /* HIP runtime error / no device                                    */
    STCSP_E_NOMEM = -4,        /* a device pool overflowed and could not grow                      */
    STCSP_E_INTERNAL = -5,     /* watchdog / invariant violation inside a kernel                   */
    STCSP_E_STATE = -6         /* call sequence error                                              */
};

typedef struct stcsp_engine stcsp_engine;

/* Build an engine for one model on one device: classifies the constraints
 * (solverConstraintQueuePush, src/constraint.cpp:254-318), compiles them for the device and
 * allocates the frontier / state-table / edge-log pools. */
int stcsp_engine_create(const stcsp_problem *problem, const stcsp_options *options, stcsp_engine **out);

/* Run the whole search (solveralgorithm.cpp:966-971 and everything it calls) and fill *result
 * with the raw automaton. Blocking. May be called repeatedly (the reference's -t loop,
 * src/solver.cpp:295-349, re-solves the same model). */
int stcsp_engine_solve(stcsp_engine *engine, stcsp_result *result);

/* Copy the device-resident automaton of the last solve() out (only needed with
 * STCSP_F_NO_EXPORT, or per shard in sharded mode). */
int stcsp_engine_export(stcsp_engine *engine, stcsp_result *result);

/* ---- post-search graph passes on the device -------------------
 * Run on the compacted automaton that the last solve()/export() left in HBM; the edge indices of
 * the returned flags are those of that stcsp_result. Replaces, in this order,
 *   graphTraverse         src/graph.cpp:357-418 (called at src/solveralgorithm.cpp:974)   always
 *   adversarialTraverse   src/graph.cpp:304-355 (-a, solveralgorithm.cpp:975-978); the reference
 *                         hard-codes variable index 5 (graph.cpp:329)
 *   adversarialTraverse2  src/graph.cpp:247-302 (-z, solveralgorithm.cpp:980-983); the reference
 *                         hard-codes opponent 5 / avatar 6 (graph.cpp:275)
 * Unsharded engines only (sharded runs post-process the merged automaton on the host). */
typedef struct stcsp_post_options {
    int32_t adversarial_var;  /* variable index for adversarialTraverse, -1 = pass not run          */
    int32_t adversarial2_op;  /* opponent variable for adversarialTraverse2, -1 = pass not run      */
    int32_t adversarial2_ava; /* avatar variable                                                    */
    int32_t reserved;
} stcsp_post_options;

typedef struct stcsp_post_result {
    int64_t n_states, n_edges;  /* sizes of the arrays below (== the exported stcsp_result)          */
    const uint8_t *state_valid; /* [n_states] Vertex::valid after the passes                         */
    const uint8_t *state_final; /* [n_states] Vertex::final                                          */
    const uint8_t *edge_alive;  /* [n_edges] 0 = the reference removed this edge from its EdgeMap    */
    int32_t adver1, adver2;     /* what the reference prints as "adver1: %d; " / "adver2: %d" (the
                                   root's valid flag after the pass), -1 when the pass was not run   */
    int32_t rounds[3];          /* sweeps until the fixpoint: traverse, adversarial, adversarial2    */
    double seconds;             /* wall time of the passes incl. copying the flags out               */
} stcsp_post_result;

int stcsp_engine_postprocess(stcsp_engine *engine, const stcsp_post_options *options, stcsp_post_result *out);

void stcsp_engine_destroy(stcsp_engine *engine);

/* Message of the last error on this engine (or of the last failed create when engine==NULL). */
const char *stcsp_engine_last_error(const stcsp_engine *engine);

/* ------------------------------------------------------------------------------------------
 * Sharded stepping interface (one engine per GPU, options.world > 1). There is no reference
 * counterpart (the reference is single-threaded); this is the seam the multi-GPU driver uses:
 *
 *   begin();                                   rank owning the root seeds it
 *   loop:
 *     expand_local()                           run this shard's open search nodes until only
 *                                              leaf successor candidates remain, bucketed by
 *                                              owner = hash(cid, signature) % world
 *     outbox(peer) -> device ptr, count        fixed-size candidate records for `peer`
 *     <driver moves records between shards: RCCL all-to-all-v over xGMI>
 *     commit(device ptr, count)                lookup-or-insert each candidate's state, log its
 *                                              edge, open the successor node if the state is new.
 *                                              Asynchronous: the records must stay valid until the
 *                                              next expand_local() / finish() returns
 *     stop when every shard has no open node and no candidate
 *   export() per shard; stcsp_merge_shards() on the gathering rank.
 *
 * State ids in sharded mode are global: gid = ((int64)owner << 40) | local_index.
 * ------------------------------------------------------------------------------------------ */
int stcsp_engine_begin(stcsp_engine *engine);
int stcsp_engine_expand_local(stcsp_engine *engine, int64_t *open_nodes_left);
int stcsp_engine_candidate_bytes(const stcsp_engine *engine); /* record stride */
int stcsp_engine_outbox(stcsp_engine *engine, int peer, void **device_ptr, int64_t *count);
int stcsp_engine_commit(stcsp_engine *engine, const void *device_records, int64_t count);
int stcsp_engine_finish(stcsp_engine *engine); /* closes the timed search phase */

/* Frontier redistribution ("load-imbalance-triggered redistribution of branch nodes",
 * the work unit being the branch case of solverSolveRe, src/solveralgorithm.cpp:911-939). Open search nodes are
 * self-contained records, so any shard can expand any of them:
 *
 *   set_expand_budget(max_rounds, min_open)  expand_local() returns early -- with open nodes left -- once it has
 *                                            run max_rounds launch rounds AND holds at least min_open open nodes
 *                                            (0, 0 = run the local frontier dry, the default). This is what lets
 *                                            the driver see an imbalance, and what makes the start "expand on the
 *                                            root's shard for a few rounds, then scatter".
 *   donate(want) -> device ptr, count        removes up to `want` open nodes from the OLDEST end of the local
 *                                            frontier (the shallowest nodes = the biggest subtrees) and returns them
 *                                            as fixed-size transfer records (node_bytes() each: src state gid,
 *                                            constraint-set TAG, until-expire bits, dirty seed, domain block);
 *                                            valid until the next donate / expand_local
 *   <driver moves the records: the same all-to-all-v as the candidates>
 *   adopt(device ptr, count)                 pushes received records onto the local frontier (the sender's
 *                                            constraint sets must have been imported first: sets_import)
 */
int stcsp_engine_set_expand_budget(stcsp_engine *engine, int64_t max_rounds, int64_t min_open);
int stcsp_engine_node_bytes(const stcsp_engine *engine); /* transfer record stride */
int stcsp_engine_donate(stcsp_engine *engine, int64_t want, void **device_ptr, int64_t *count);
int stcsp_engine_adopt(stcsp_engine *engine, const void *device_records, int64_t count);
/* counters of the last / current solve without exporting the automaton */
int stcsp_engine_counters(stcsp_engine *engine, stcsp_counters *out);
/* Constraint-set registry exchange: every shard must know a set before it can open a state that
 * uses it. sets_blob returns this shard's registry serialised as int32 words (valid until the
 * next call); sets_import registers the sets of another shard's blob (idempotent). */
int stcsp_engine_sets_blob(stcsp_engine *engine, const int32_t **words, int64_t *n_words);
int stcsp_engine_sets_import(stcsp_engine *engine, const int32_t *words, int64_t n_words);

/* Kernel-granularity check (tests, diagnostics): propagate `count` caller-provided domain blocks
 * (N*K words each, point-major: word[p*N + v], bit i <=> value lb[v] + i) under constraint set `set`
 * with the production device code of one search node (the propagation of generalisedArcConsistent,
 * src/solveralgorithm.cpp:617-706, followed by the classification of solverSolveRe, :738). The blocks are
 * overwritten by their propagated form; outcome[i] = 0 wiped out, 1 branch node, 2 leaf whose
 * constraint-set translation is not known yet, 3 leaf. *skipped (may be NULL) receives the number of
 * revisions the device skipped for their enumeration budget (0 => every block is at its GAC fixpoint).
 * Unsharded engines, between solves. */
int stcsp_engine_propagate(stcsp_engine *engine, int32_t set, uint32_t expire, uint32_t *blocks, int64_t count,
                           int32_t *outcome, int64_t *skipped);

#define STCSP_GID_SHIFT 40

#ifdef __cplusplus
}
#endif
#endif /* STCSP_ENGINE_H */
