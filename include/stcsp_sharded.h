/* stcsp_sharded.h -- the multi-GPU superstep loop behind the C-ABI.
 *
 * The reference is a single-threaded program (no threads, no MPI / NCCL), so there is
 * no reference interface to mirror here. The unit of data parallelism is the open search node (one call of
 * solverSolveRe, reference src/solveralgorithm.cpp:733; the branch case :911-939 is the work that moves
 * between GPUs), the one shared structure is the automaton's state table (VertexTable, src/graph.h:64),
 * sharded by owner = hash(constraint set, signature) % world.
 *
 * stcsp_engine_solve_sharded() runs the whole sharded search of ONE rank (one engine = one GPU): expand the
 * local frontier, exchange a small table of counts, redistribute open nodes when the load is lopsided,
 * exchange the leaf successor candidates (all-to-all-v), commit what arrived -- until no rank has open
 * nodes or candidates. No Python, no PyTorch in the loop: the host language only provides the TRANSPORT,
 * three collectives over the ranks:
 *
 *   libstcsp_rccl.so   stcsp_transport_rccl_create(): RCCL over xGMI, one process per GPU
 *                      (ncclAllGather for the count table, grouped ncclSend / ncclRecv for the records, all
 *                      on the engine's HIP stream)
 *   libstcsp_hip.so    stcsp_transport_local_create(): ranks = host threads of ONE process, one engine
 *                      each (on the same or on different GPUs of the node); records move with
 *                      hipMemcpyPeerAsync, the count table through host memory. This is what the `stcsp`
 *                      CLI uses (--shards N) and what the tests drive with several engines on one GPU.
 *
 * Any other transport (MPI, a host language's own channels) fills the same struct.
 */
#ifndef STCSP_SHARDED_H
#define STCSP_SHARDED_H
#include <stdint.h>
#include "stcsp_engine.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct stcsp_transport {
    void *self;
    int32_t rank, world;
    /* Every rank contributes n int64 words (host memory); `all` receives world * n, rank-major. Blocking. */
    int (*all_gather_i64)(void *self, const int64_t *mine, int32_t n, int64_t *all);
    /* The same for n bytes of host memory (n equal on every rank: callers pad). Blocking. */
    int (*all_gather_bytes)(void *self, const void *mine, int64_t n, void *all);
    /* All-to-all-v of DEVICE buffers of 32-bit words: send_words[p] words go to peer p, recv_words[p] words
     * arrive from peer p, peers back to back in both buffers. `stream` is the engine's hipStream_t: the
     * exchange is ordered behind the work already enqueued on it, and when the call returns the received
     * words are either in place or will be before anything enqueued on `stream` afterwards runs. The send
     * buffer may be reused after the next collective call of this transport. */
    int (*all_to_all_v)(void *self, const void *send, const int64_t *send_words, void *recv, const int64_t *recv_words,
                        void *stream);
    const char *(*last_error)(void *self);
} stcsp_transport;

typedef struct stcsp_sharded_options {
    int64_t budget_rounds;   /* launch rounds per superstep once a rank holds enough open nodes to share (0: 8) */
    int64_t share_per_rank;  /* ... "enough" = this many open nodes per rank (0: 64)                              */
    int64_t max_supersteps;  /* safety net (0: 1,000,000)                                                       */
} stcsp_sharded_options;

typedef struct stcsp_sharded_stats {
    int64_t supersteps;
    int64_t nodes_donated, nodes_adopted; /* open search nodes this rank gave away / took over                 */
    int64_t candidates_sent, candidates_received;
    double seconds_collectives;           /* wall time inside the transport's calls                            */
    double seconds_expand;                /* ... inside expand_local (launch rounds + their synchronisations)   */
    double seconds_pack;                  /* ... packing the outboxes / serialising the set registry            */
    double seconds_commit;                /* ... enqueueing commit / adopt (the kernels themselves run behind)  */
} stcsp_sharded_stats;

/* The engine must have been created with options.rank / options.world equal to the transport's (world 1: with
 * STCSP_F_STEPPED). Blocking; every rank of the group must call it. Afterwards stcsp_engine_export() gives this
 * shard's states and raw leaf edges; stcsp_merge_shards() (stcsp_host.h) joins the shards. When any rank's engine
 * fails, EVERY rank returns an error at the same superstep (the status travels with the count table): nobody is
 * left waiting in a collective. */
int stcsp_engine_solve_sharded(stcsp_engine *engine, const stcsp_transport *transport, const stcsp_sharded_options *options,
                               stcsp_sharded_stats *stats);

/* In-process transport group: `world` transports that belong together, one per host thread / engine. */
typedef struct stcsp_local_group stcsp_local_group;
int stcsp_local_group_create(int32_t world, stcsp_local_group **out);
const stcsp_transport *stcsp_local_group_transport(stcsp_local_group *group, int32_t rank);
void stcsp_local_group_destroy(stcsp_local_group *group);

/* ---- libstcsp_rccl.so (links librccl; kept apart so that the engine library itself has no RCCL dependency) ---- */
#define STCSP_RCCL_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls this and hands the bytes to the other ranks (file, pipe, environment, MPI ...). */
int stcsp_rccl_unique_id(void *id_out);
/* ncclCommInitRank on `device` + the transport over it. */
int stcsp_transport_rccl_create(const void *unique_id, int32_t rank, int32_t world, int32_t device, stcsp_transport **out);
void stcsp_transport_rccl_destroy(stcsp_transport *transport);

#ifdef __cplusplus
}
#endif
#endif /* STCSP_SHARDED_H */
