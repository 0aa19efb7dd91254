// dev_layout.hpp -- device-visible structures of the engine: control-block layout, the device plan
// (frontier segment stack + next-round arguments), kernel context. Included by engine.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "device_types.hpp"
#include "stcsp_engine.h"
namespace stcsp {
namespace dev {


constexpr int R = kRegions;
constexpr int CST = kCursorStride;
constexpr int kStatSlots = 64;
constexpr int kStatWords = 48;
enum { ST_NODES = 0, ST_FAILS, ST_LEAVES, ST_REVS, ST_EVALS, ST_REQUEUE, ST_NEWSTATES, ST_WAVEREVS, ST_SWEEPS, ST_SKIPPED,  // (the first kMirrorCounters: Progress::counters)
       ST_CYC_LOAD, ST_CYC_SWEEP, ST_CYC_WAVE, ST_CYC_CLASSIFY, ST_CYC_COMMIT, ST_CYC_TOTAL,
       ST_BATCHES, ST_BATCH_ITEMS, ST_CYC_BATCH, ST_BATCH_REFUSED, ST_CYC_BATCH_AB, ST_CYC_BATCH_DE, ST_BATCH_TUPLES,  // revise_batch (phases build)
       ST_CYC_STAGE, ST_BLOCKS, ST_CYC_FINAL, ST_ROUNDS_FINAL, ST_CYC_BLOCK,
       ST_CYC_RV_SETUP, ST_CYC_RV_LOOP, ST_CYC_RV_WB, ST_CYC_CLOSE, ST_CYC_LEAF, ST_RV_BLOCKS, ST_RV_OPEN, ST_RV_LANES,
       ST_CYC_RV_DIGITS, ST_CYC_RV_EVAL_BITMAP, ST_CYC_RV_EVAL_CODE, ST_CYC_RV_SUPPORT, ST_RV_BLOCKS_CODE,
       ST_ROOTS, ST_CYC_ROOT_SWEEP, ST_CYC_ROOT_WAVE, ST_ROOT_BATCHES, ST_ROOT_CYC_BATCH, ST_ROOT_RV_BLOCKS, ST_ROOT_REFUSED };  // first nodes of states (phases build)
static_assert(ST_ROOT_REFUSED < kStatWords, "statistics row");
// per-wavefront LDS words behind the block copy: [0] rows examined by the current node's sweeps, [1 + ST_x]
// the wavefront's work counters of this launch (flushed to the global statistics once per launch)
constexpr int kLdsStatWords = 12;
// LDS words of one wavefront: [lane-enumerated values | expression stack] (general revisions only) + the
// AND-accumulator copy of the block + the counters. LITE kernels (no general revision) keep the last part only.
// ... + the sibling stack of the chained expansions (dev_kernels.hpp expand_node): kSibDepth node records.
constexpr int kSibDepth = 4;  // (default; Ctx::sib_depth is what a launch uses: the big-workgroup variant trades depth for LDS)
__host__ __device__ inline int wave_sib_offset(int NK, int stack_slots, bool lite) {
    return (lite ? 0 : (kMaxLowVars + stack_slots) * 64) + ((NK + kLdsStatWords + 63) & ~63);
}
__host__ __device__ inline int wave_scratch_words(int NK, int stack_slots, bool lite, int sib_depth) {
    return wave_sib_offset(NK, stack_slots, lite) + sib_depth * ((4 + NK + 3) & ~3);
}
constexpr int kMissStride = 66;  // set, nfirst, 64 values
constexpr uint32_t kPending = 0xffffffffu;

// control block (u32 words; every cursor on its own 64-byte line)
constexpr int kSlotCursors = 16;  // slot tickets of a round are dealt by this many counters (one address would serialise ~200 atomics per us)
struct CtlLayout {
    int out0, cand0, edge0, misc0, slotcur0, words;  // out0: TWO sets of R cursors (round parity)
    __host__ __device__ CtlLayout(int world) {
        out0 = 0;
        cand0 = 2 * R * CST;
        edge0 = cand0 + world * R * CST;
        misc0 = edge0 + R * CST;
        slotcur0 = misc0 + 8 * CST;
        words = slotcur0 + kSlotCursors * CST;
    }
    __host__ __device__ int out(int parity, int r) const { return out0 + (parity * R + r) * CST; }
};

// The frontier bookkeeping lives on the device: the last workgroup of every k_expand launch
// accounts the round (finalize_round) and plans the next one (plan_next), so the host enqueues
// bursts of rounds and synchronises once per burst.
constexpr int kMaxSegments = 4096;
enum PlanStatus : int { PS_RUN = 0, PS_DONE, PS_NEED_ARENA, PS_NEED_EDGES, PS_NEED_STATES, PS_NEED_TABLE, PS_HOST, PS_OUTBOX_FULL, PS_STACK_FULL };
struct DevSegment {
    unsigned long long base;  // word offset into the arena
    unsigned cap;             // node slots per region
    int count[R];
    int pad;
};
struct Plan {
    // Gate of the planned round: (id of the ONE launch that may execute it) << 32 | its number of slots, written and read
    // as one 8-byte word. A launch is dealt more workgroups than a small round has slots; such a surplus workgroup may be
    // scheduled late -- when the GPU is shared with another process, or busy with the export streams, AFTER the last
    // working workgroup of its launch has planned the next round. It must not mistake that plan for its own: it then
    // reads either the old gate (no slot for it: leaves) or the new one (another launch's id: leaves).
    unsigned long long gate;
    int status, parity, nslots, sp;
    int take[R], count[R];
    unsigned long long in_base, out_base, arena_top, arena_words, slot_cap;
    unsigned in_cap, out_cap, edge_cap, state_cap, cand_cap, done_blocks;
    int chunk_r, world;
    int chain;                                 // expansions per slot in the planned round
    int chain_small, chain_big, chain_thresh;  // policy: chain_small while a round has <= chain_thresh nodes
    int chain_heavy;                           // a slot stops chaining after this many cycles in one launch
    long long rounds, open_total;
    unsigned states_seen;   // ... and the number of states in the table then
    unsigned edge_seen[R];  // edge-log cursors at the end of the last accounted round (the host streams the log out while the search runs)
    DevSegment stack[kMaxSegments];
};
enum { MISC_NSTATES = 0, MISC_NMISS = 1, MISC_ERROR = 2 };
enum { ERR_WATCHDOG = 1, ERR_TABLE_SPIN = 2, ERR_EDGE_OVERFLOW = 3, ERR_STATE_OVERFLOW = 4, ERR_UNKNOWN_SET = 5,
       ERR_EMPTY_DOMAIN = 6, ERR_OUT_OVERFLOW = 7, ERR_CAND_OVERFLOW = 8,
       ERR_ADOPT_OVERFLOW = 9, ERR_COMMIT_OUT_OVERFLOW = 10, ERR_FLUSH_OVERFLOW = 11 };

// Progress mirror in pinned HOST memory (streaming export): finalize_round of launch g writes the edge-log cursors and
// the state count as of the end of launch g, each word tagged with g (high half), so the host can ship finished parts
// of the log WHILE a burst of launches is still running. No fence orders the words (a system-scope release would
// write the whole L2 back, every round): a snapshot is valid when all its tags agree. (It is shipped one launch
// late: only when launch g + 1 is seen to have finalized has launch g ended and written its caches back.)
struct Progress {
    unsigned long long edge_seen[R];  // tag << 32 | cursor
    unsigned long long states_seen;   // tag << 32 | count
    // number of the round whose launch has STARTED (written by its first workgroup): launches of one stream run one after the
    // other, so the round before it has ENDED -- its snapshot may be shipped now, a whole round earlier than when the next
    // snapshot shows up
    unsigned long long started;
    // What the host needs to decide about the next burst, written by whoever planned last (finalize_round of the last round run,
    // or k_replan): read AFTER the burst's end has been seen (event), so no tags -- and no device-to-host copy of the plan, the
    // control block and the statistics between two bursts (three small asynchronous copies cost ~50-80 us of every burst
    // boundary under the ROCm 7.2 runtime, where bursts are short: engine.hip create()).
    long long rounds, open_total;
    int status;  // PlanStatus
    int pad;
    // ... and, with the verdict PS_DONE, the work counters of the solve summed over their slots (ST_NODES .. ST_SKIPPED): the end of
    // an unsharded search needs no copy either (the edge-log cursors and the state count are the words above)
    unsigned long long counters[16];
};
constexpr int kMirrorCounters = 10;  // ST_NODES .. ST_SKIPPED

struct ImgOff {
    int sets, cons, scope, strides, items, sweep, nextpart, itemrows, tables, var_lb, var_init, sig_vars, until_y, firstvars, trans, transvals,
        arr_off, code, stables, fstrides, words;  // stables: row tables of the lane-revised items (FlatProgram::stables)
    int hot_words;  // the image's first hot_words words are the sections every node touches (see upload_program)
    int init_stride;  // var_init holds one row of N words per constraint set (N) or one row for all of them (0): the domains a fresh time point starts from
};

struct Ctx {
    // ---- words 0..63: what the node loops of k_expand read (they see the context through a lane-striped
    // register copy, dev_kernels.hpp ctx_from: one VGPR covers these)
    int N, K, NK, NS, CS, ES, KL, sig_len, n_sig, n_until_cons, world, rank, nsets, stack_slots;
    int sharded;  // leaves emit successor candidates for their owner (world > 1, or STCSP_F_STEPPED) instead of committing in place
    int stage_words;  // image words every workgroup copies into LDS: o.words (whole image), a prefix of whole hot sections, or 0
    // the compiled program: one contiguous image of 32-bit words (sections at the offsets in `o`),
    // staged into LDS by every workgroup of k_expand when it fits; the bytecode and the
    // user arrays (potentially large) stay in global memory
    const uint32_t *img;
    ImgOff o;
    uint32_t slot_mask;
    unsigned long long *slots;
    uint32_t *state_keys;
    uint32_t state_cap;
    uint32_t edge_cap;  // records per region
    uint32_t *ctl;
    uint32_t *edges;
    Plan *plan;
    uint32_t *arena;
    uint32_t *cand;  // outbox [owner][region] x cand_cap records (sharded runs)
    int *miss;
    int miss_cap;
    int max_iw;  // widest dirty mask (words) over the program's constraint sets
    // State table = `slot_mask + 1` ENTRIES of 2^tab_shift words (one or more whole 128-byte lines): the state's key
    // [set tag, signature...] in words [0, KL) and the slot word {hi: generation of the solve that wrote it, lo: state index or
    // kPending} in the last two -- "is this the state I am looking for?" is decided by ONE line read (round 3: slot word, then
    // the key out of state_keys: two dependent round trips at every leaf). An entry whose generation is not the running
    // solve's is free: no memset of the table between solves.
    int tab_shift;
    uint32_t tab_gen;
    // ---- beyond word 63: general wavefront revisions and diagnostics only
    int budget_bitmap, budget_code;  // odometer steps one wavefront revision may take (bitmap lookup / bytecode); longer ones are skipped
    int sib_depth;                   // node records on a wavefront's sibling stack (kSibDepth; 2 under the big-workgroup variant)
    int W;                           // bitset words per (variable, time point): 1, or 2 / 4 for domains of up to 64 / 128 values (NK = W*N*K)
    const int *code;
    const int *arr_data;
    const int *tdirect;  // direct transition tables (SetDesc::trans_count < 0): one look-up per leaf, kept out of the LDS-staged image
    unsigned long long *stats;
    Progress *progress;  // null: no mirror
};
static_assert(offsetof(Ctx, budget_bitmap) <= 64 * 4, "the node loops' part of Ctx must fit one lane-striped register");
__host__ __device__ inline int table_entry_shift(int KL) { return KL <= 30 ? 5 : (KL <= 62 ? 6 : 7); }  // 32 / 64 / 128 words per entry

struct ExpandArgs {  // per-round view, read from the device plan by every wavefront
    const uint32_t *in_base;
    uint32_t in_cap;
    uint32_t *out_base;
    uint32_t out_cap;
    uint32_t *cand_base;
    uint32_t cand_cap;
    int parity;
};

struct CommitArgs {
    const uint32_t *cand_base;  // contiguous array of candidate records
    long long total;
};

}  // namespace dev
}  // namespace stcsp
