// device_types.hpp -- data layout shared by the host-side constraint-set compiler (cset.cpp)
// and the HIP kernels (engine.hip). Everything is 32-bit words in HBM.
//
// DOMAIN BLOCK of one open search node (the build's replacement for Variable::currLB/currUB,
// reference src/variable.h:19-20): N*K bitset words, point-major:
//       word[p * N + v]  bit i  <=>  value (lb[v] + i) is in the domain of v at time point p
// (|D| <= 32 => one word per (v,p); W = 1).  A node record is a 4-word header + the block:
//       [0] src state id (low 32)   [1] src state id (high 32: owner rank in sharded mode)
//       [2] constraint-set index     [3] until-expire bits (Constraint::expire, one per UNTIL)
//       [4 ...] block
// padded to a multiple of 4 words (16 B) so lanes can move it with aligned wide accesses.
#pragma once
#include <cstdint>

namespace stcsp {

constexpr int kRegions = 32;        // cursor shards per segment (spreads allocation atomics)
constexpr int kMaxDomRegs = 4;      // N*K <= 64 * kMaxDomRegs words live in VGPRs, lane-striped
constexpr int kCompactSweepItems = 128;  // sets with more small items than this sweep over a compacted dirty list
constexpr int kMaxLowVars = 6;      // lane-enumerated scope variables per revision (2^6 = 64; value bits packed 5 x 6 in a register)
constexpr int kMaxScope = 64;       // scope variables per constraint (one lane each)
constexpr int kCursorStride = 16;   // words between cursors (64 B: one cursor per cache line)
constexpr uint32_t kRootTag = 0x7fffffffu;
// node header word 2 = constraint-set ordinal (low kSetBits bits) | dirty seed (the rest): seed 0 = revise every item (fresh state /
// root), kSeedNone = nothing to revise (a re-queued fixpoint), else 1 + the variable whose time-0 domain the parent bisected
// (N <= 256, so 12 bits hold it). 20 bits of ordinal: a million constraint sets (round 3: 16 / 16 bits, 65,535 sets).
constexpr int kSetBits = 20;
constexpr uint32_t kSetMask = (1u << kSetBits) - 1u;
constexpr uint32_t kSeedNone = (1u << (32 - kSetBits)) - 1u;
// odometer steps (blocks of <= 64 tuples, ~1.1 k cycles each when the bitmap is in HBM) one bitmap revision may take; longer
// ones are skipped. 4096 made juggling_b4_f5_nosym 15.8 ms (1.5 with 64) and digitinvader9 23.4 (20.8): a revision that
// long rarely prunes enough to pay for itself. 16 is faster still but propagates less than the reference does
// (791 instead of 781 nodes on juggling_b4_f5_nosym).
constexpr unsigned long long kBudgetBitmapIters = 32;
constexpr unsigned long long kBudgetCodeIters = 32;      // ... and an interpreted one
constexpr int kCandHdr = 8;                 // header words of a candidate record
constexpr int kSmallMaxRows = 64;            // table rows a single lane may scan
constexpr long long kBitmapMaxBits = 1ll << 22;  // largest tuple bitmap the HOST tabulates per constraint (~0.1 us per tuple)
constexpr long long kDirectTransMax = 1ll << 22;  // entries of one constraint set's direct transition table (4 B each)
constexpr long long kDirectTransTotalMax = 1ll << 26;  // ... and of all the tables of one program (256 MB)
constexpr long long kWideConditional = 1ll << 20;  // a conditional constraint over more tuples than this marks a program whose search is shallow and bushy (engine.hip: chain policy)
constexpr int kBatchArity = 16;             // scope variables of an item revise_batch can take (one row of 16 lanes scans a scope)
constexpr int kTabulateMaxStack = 32;        // operand-stack entries of k_tabulate's per-thread interpreter (deeper programs are not tabulated on the device)
constexpr long long kBitmapMaxBitsDevice = 1ll << 28;  // ... and the largest one at all: bigger ones up to here are tabulated on the device

enum ConType : int32_t { CT_NEXT = 0, CT_POINT = 1, CT_UNTIL = 2, CT_AT = 3 };

// bytecode: one word per instruction, (arg << 8) | op; OP_CONST is followed by its immediate.
enum Op : int32_t {
    OP_END = 0,
    OP_CONST,
    OP_VAR,   // arg = scope index
    OP_ARR,   // arg = array id; index on top of stack
    OP_ABS,
    OP_NOT,
    OP_ADD,
    OP_SUB,
    OP_MUL,
    OP_DIV,
    OP_MOD,
    OP_LT,
    OP_GT,
    OP_LE,
    OP_GE,
    OP_EQ,
    OP_NE,
    OP_MASK_T,  // arg = stack depth d: open a guarded region, live iff stack[top-d] != 0
    OP_MASK_F,  //                      ... live iff stack[top-d] == 0
    OP_MASK_POP,
    OP_SEL_IF,     // [c a b] -> c ? a : b
    OP_SEL_AND,    // [a b]   -> a ? b : 0
    OP_SEL_OR,     // [a b]   -> a ? 1 : b     (raw b, solveralgorithm.cpp:378-384)
    OP_SEL_IMPLY   // [a b]   -> a == 0 ? 1 : (a <= b)
};

struct ConDesc {          // one constraint of one constraint set
    int32_t type;         // ConType
    int32_t npoints;      // time points to enforce: 1 if the constraint has `first`/@, else K
    int32_t scope_off;    // into scope[]: variable ids, first-occurrence order
    int32_t scope_len;
    int32_t code_off;     // into code[]
    int32_t code_len;
    int32_t x, y;         // NEXT: X == next Y ; UNTIL: X until Y
    int32_t until_ordinal;
    int32_t uses_valid;   // program contains array lookups => track `valid` + liveness
    int32_t bitmap_off;   // >= 0: into tables[], satisfying-tuple bitmap over the full initial
                          //       product (bit index = sum_j bitpos_j * stride_j); -1: interpret code
    int32_t stride_off;   // into strides[] (parallel to the scope) when bitmap_off >= 0
    int32_t n_forbidden;  // bitmap constraints: number of violating tuples of the full product when that is
                          // at most kFewForbidden, else -1. A value can only lose its support when the
                          // product of the OTHER domains fits inside the forbidden set, so a revision with
                          // (product of all domains) / (largest domain) > n_forbidden cannot prune and is skipped.
    int32_t pad1;
};
constexpr int kFewForbidden = 64;

// A propagation work item = one constraint at one time point. Items are what the dirty mask
// tracks; items [0, nsmall) of a set are "small" (one lane revises one item), the rest are
// revised by the whole wavefront.
enum ItemType : int32_t { IT_NEXT = 0, IT_UNTIL = 1, IT_SMALL = 2, IT_WAVE = 3 };
struct ItemDesc {
    int32_t type;
    int32_t point;
    int32_t con;          // index into cons[] (absolute)
    int32_t arity;        // IT_SMALL: 1..4 (word variable first); IT_WAVE: scope length
    int32_t idx[4];       // block word indices p*N+v.  NEXT: idx[0] = (p,X), idx[1] = (p+1,Y)
                          // IT_WAVE: the constraint's scope_off, bitmap_off, stride_off, n_forbidden (ConDesc)
    int32_t toff;         // IT_SMALL: into tables[], one 32-bit row per tuple of variables 1..3; IT_WAVE: code_off
    int32_t r1, r2;       // IT_SMALL: radices (initial domain sizes) of variables 1 and 2; IT_WAVE: r1 = uses_valid, r2 = code_len
    int32_t aux;          // NEXT: lbX - lbY ; UNTIL: ordinal ; SMALL: number of table rows
};

// IT_SMALL tables: row (b1, b2, b3) lives at toff + small_row_stride(r1) * (b2 + r2 * b3) + b1
inline constexpr int small_row_stride(int r1) { return (r1 + 3) & ~3; }

// What the lane-per-item sweep needs of an ItemDesc, packed into one 16-byte record so that a
// lane fetches its item with a single 128-bit LDS read:
//   x = idx[0] | idx[1] << 8 | idx[2] << 16 | idx[3] << 24      (block word indices, N*K <= 256)
//   y = type | arity << 2 | r1 << 5 | r2 << 11 | aux << 17       (aux signed: NEXT shift clamped to
//                                                                  [-32, 32], UNTIL ordinal)
//   z = toff
inline void pack_sweep_item(const ItemDesc &it, uint32_t out[4]) {
    auto byte = [](int32_t v) { return (uint32_t)(v < 0 ? 0 : v) & 0xffu; };
    out[0] = byte(it.idx[0]) | byte(it.idx[1]) << 8 | byte(it.idx[2]) << 16 | byte(it.idx[3]) << 24;
    int32_t aux = it.type == IT_SMALL ? 0 : it.aux;
    if (it.type == IT_NEXT) aux = aux > 32 ? 32 : (aux < -32 ? -32 : aux);
    out[1] = ((uint32_t)it.type & 3u) | ((uint32_t)it.arity & 7u) << 2 | ((uint32_t)it.r1 & 63u) << 5 | ((uint32_t)it.r2 & 63u) << 11 |
             (uint32_t)aux << 17;
    out[2] = (uint32_t)it.toff;
    out[3] = 0u;
}

struct SetDesc {          // one constraint set (entry of Solver::seenConstraints)
    int32_t con_begin;    // into cons[]
    int32_t ncons;
    int32_t varcons_off;  // into varcons[]: [N][cw] bitmask rows, cw = (ncons + 31) / 32
    int32_t cw;
    int32_t self_loop;    // translation maps the set to itself whatever the leaf values
    int32_t nfirst;       // variables whose time-0 value the translation reads
    int32_t first_off;    // into firstvars[]
    int32_t trans_begin;  // into trans[]: known (values -> next set) transitions of this set; into tdirect[] when trans_count < 0
    int32_t trans_count;  // < 0: the transitions are a table indexed by the captured tuple (fstrides[first_off + j] = stride of variable j)
    int32_t tag;          // id stored in state keys (ordinal when unsharded, content hash when sharded)
    int32_t item_begin;   // into items[]
    int32_t nitems;
    int32_t nsmall;       // items [0, nsmall) are lane-revised
    int32_t iw;           // dirty-mask words = (nitems + 31) / 32
    int32_t itemrows_off; // into itemrows[]: [N*K][iw] rows: items that read block word (p,v)
    int32_t next_off;     // into nextpart[]: [N*K][2] partner entries of the X == next Y arcs that are kept
                          // consistent eagerly (close_next) instead of being items; -1: none in this set
    int32_t witem_begin;  // number of wavefront-revised items of the sets before this one: the device image holds
                          // the ItemDesc records of those items only (item i >= nsmall -> record witem_begin + i - nsmall)
    int32_t pad[3];
};
// nextpart[word] = two entries; entry: 0 = none, else (partner block word + 1) | (lbX - lbY + 64) << 16 |
// side << 24 (shift clamped to [-32, 32]; side 0: this word is X[p] and the partner Y[p+1], side 1: this
// word is Y[p+1] and the partner X[p]). Entry 0 is used first; entry 1 only when the word is on both sides
// of arcs (possible for K > 2 only).

struct TransDesc {
    int32_t vals_off;     // into transvals[]: nfirst values
    int32_t next_set;     // set index
};

#if defined(__HIPCC__)
#define STCSP_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define STCSP_HD inline
#endif

// Hash of a state key (constraint-set tag, signature words): table slot, slot tag and, in sharded
// runs, the owner shard = (h >> 40) % world. Every word is mixed on its own (position-salted) and the
// terms are XORed, so that a wavefront can hash a key with one lane per word and a cross-lane XOR
// instead of a serial chain of 64-bit multiplies (dev_propagate.hpp) -- every implementation (device,
// k_rehash, host root insertion, oracle/frontier_model.cpp) goes through these two functions.
STCSP_HD unsigned long long key_term(int j, uint32_t w) {
    unsigned long long t = ((unsigned long long)(uint32_t)(j + 1) << 32 | w) * 0xBF58476D1CE4E5B9ull;
    t ^= t >> 29;
    t *= 0x94D049BB133111EBull;
    return t ^ (t >> 31);
}
STCSP_HD unsigned long long mix_final(unsigned long long h) {
    h *= 0x94D049BB133111EBull;
    h ^= h >> 32;
    return h;
}
constexpr unsigned long long kHashSeed = 0x9E3779B97F4A7C15ull;
STCSP_HD unsigned long long key_hash(const uint32_t *key, int kl) {
    unsigned long long h = kHashSeed;
    for (int j = 0; j < kl; j++) h ^= key_term(j, key[j]);
    return mix_final(h);
}

// Owner shard of a state key. A model without any signature word (no next/first/fby/until) has one
// state per constraint set and its root key is the plain (tag 0), which a leaf of set 0 must find
// again: that key lives where begin() put the root, on shard 0, not where its hash points.
STCSP_HD int key_owner(unsigned long long h, int world, int kl, uint32_t tag) {
    if (kl == 1 && tag == 0u) return 0;
    return (int)((h >> 40) % (unsigned)world);
}

// record strides in words (all multiples of 4)
STCSP_HD int node_stride(int N, int K) { return (4 + N * K + 3) & ~3; }
STCSP_HD int cand_stride(int N, int K, int sig_len) { return (kCandHdr + sig_len + N + N * K + 3) & ~3; }
STCSP_HD int edge_stride(int N) { return (4 + N + 3) & ~3; }
// transfer record of an open search node (frontier redistribution between shards):
//   [0,1] src state gid  [2] constraint-set TAG (ordinals differ between shards)  [3] until-expire bits
//   [4] dirty seed  [5..7] spare  [8..) the N*K-word block
constexpr int kXferHdr = 8;
STCSP_HD int xfer_stride(int N, int K) { return (kXferHdr + N * K + 3) & ~3; }
// candidate record: [0,1] src gid  [2] next set tag  [3] expire bits  [4,5] key hash  [6,7] spare
//                   [8..) signature, then N edge-label values, then the N*K time-advanced block

}  // namespace stcsp
