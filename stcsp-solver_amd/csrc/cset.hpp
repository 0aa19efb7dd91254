// cset.hpp -- host side of the engine: constraint-set registry, per-leaf translation and the
// compiler from constraint trees to the flat device program (device_types.hpp).
//
// Reference behaviour reproduced here (all of it runs per LEAF in the reference; the engine
// runs it once per distinct (set, captured `first` values) and caches the transition):
//   src/constraint.cpp:254-318  solverConstraintQueuePush  -- classify NEXT/POINT/UNTIL/AT,
//                                hasFirst, signature / until variables, scope
//   src/constraint.cpp:466-548  constraintTranslate & helpers -- `first e` -> constant,
//                                X == Y@k -> X == Y@(k-1) | X == first Y, drop tautologies
//   src/constraint.cpp:551-576  constraintQueueEq / constraintNodeEq -- set identity
//   src/solveralgorithm.cpp:755-805  the per-leaf translation / seenConstraints lookup
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "cvalue.hpp"
#include "device_types.hpp"
#include "tree.hpp"

namespace stcsp {

struct HostCon {
    Tree *root = nullptr;
    int type = CT_POINT;
    bool has_first = false;
    std::vector<int> scope;  // first-occurrence order (the reference's arc order)
    int x = -1, y = -1;
    int until_ordinal = -1;
};

struct HostSet {
    std::vector<HostCon> cons;
    std::vector<int> first_vars;  // sorted ids of the variables that occur under a `first`
    bool self_loop = false;
    int32_t tag = 0;
    std::map<std::vector<int>, int> trans;  // captured first-var values -> next set index
    bool direct = false;  // the whole space of captured value tuples has been enumerated ahead of need (pretranslate): the
                          // device then looks the transition up in a table indexed by the tuple, not in a list
    TreeArena arena;
};

// Tabulated form of one point constraint. Kept across compiles (keyed by the serialised tree): a recompile after
// a new constraint set only tabulates the constraints it has not seen yet.
struct TableEntry {
    bool is_small = false;        // rows for the lane-per-item sweep (ItemDesc proto, toff relative to `words`)
    ItemDesc small{};
    bool bitmap = false;          // one bit per tuple of the full initial product
    bool pending = false;         // bitmap still to be tabulated -- on the device (engine.hip k_tabulate)
    std::vector<uint32_t> words;
    std::vector<int32_t> strides;
    int32_t n_forbidden = -1;
};
// A bitmap the device has to fill in before the program can run (products the host would take seconds for)
struct TabulateTodo {
    std::vector<int32_t> key;     // table_cache key
    int32_t tables_off = 0;       // where its words sit in FlatProgram::tables
    long long product = 0;
    int32_t code_off = 0, code_len = 0, uses_valid = 0;
    std::vector<int32_t> scope;   // variable ids, scope order (= stride order)
};

struct FlatProgram {
    std::vector<TabulateTodo> todo;
    std::vector<SetDesc> sets;
    std::vector<ConDesc> cons;
    std::vector<int32_t> scope, code, firstvars, transvals;
    std::vector<int32_t> fstrides;  // parallel to firstvars: mixed-radix stride of the variable in its set's direct transition table
    std::vector<int32_t> tdirect;   // direct transition tables (SetDesc::trans_begin when trans_count < 0): next set or -1 per tuple
    std::vector<uint32_t> varcons;
    std::vector<TransDesc> trans;
    std::vector<ItemDesc> items;
    std::vector<uint32_t> itemrows, tables, nextpart;
    std::vector<uint32_t> stables;  // row tables of the lane-revised items (ItemDesc::toff of IT_SMALL items points in here): a section of
                                    // their own, so that a few hundred words every sweep reads are staged in LDS even when the tuple
                                    // bitmaps in `tables` run to megabytes (digitinvader9: 268 words beside 4.9 MB)
    std::vector<int32_t> strides;
    // Per set: 1 when a fresh state under the set may start its new time point from the set's own PROPAGATED initial domains (the
    // engine works them out once per set: engine.hip fresh_init) -- the point items over the new point are then not dirty in a fresh
    // state (dirty row N*K); all zero when the table of nsets x N words would not stay small.
    std::vector<uint8_t> set_fresh_init;
    int max_stack = 1;
    int max_cw = 1;
    int max_iw = 1;
};

class SetManager {
public:
    int N = 0, K = 2;
    std::vector<int> lb, ub;
    ArrayTable arrays;
    std::vector<int32_t> array_off, array_data;
    std::vector<uint8_t> is_sig, is_until;
    int n_sig = 0, n_until = 0, n_until_cons = 0;
    bool has_first = false;  // Solver::hasFirst (sticky)
    bool sharded = false;    // tags are content hashes instead of ordinals
    std::vector<int> sig_vars, until_x, until_y;
    std::vector<std::unique_ptr<HostSet>> sets;
    std::multimap<uint64_t, int> set_by_hash;  // content hash -> set index (constraintQueueEq decides; the hash only finds candidates)
    std::string error;
    std::map<std::vector<int32_t>, TableEntry> table_cache;
    // A point constraint too wide for a tuple bitmap whose tree holds an `if` is replaced -- in the compiled PROGRAM only, the set's
    // identity keeps the constraint as written -- by the conjunction of its branches (split_wide): serialised tree -> branch trees.
    // An empty entry: the constraint stays as it is (interpreted).
    std::map<std::vector<int32_t>, std::vector<Tree *>> piece_cache;
    TreeArena piece_arena;
    int split_mode = 1;            // 0: never; 1: conditional constraints no bitmap can hold; 2: every conditional constraint that is not lane-revised
    bool fresh_init = true;        // STCSP_FRESH_INIT=0: fresh states start from the plain initial domains, every item over the new point dirty
    int n_split = 0;               // constraints the last compile() replaced by their guarded branches
    int n_wide_conditional = 0;    // ... and conditional constraints over more than kWideConditional tuples it saw, split or not (engine: chain policy)
    long long split_target = 0;    // a branch at most this big is not split further (set by compile)
    // Bitset words per (variable, time point): 1 while every variable has at most 32 values, else 2 (<= 64) or 4 (<= 128).
    // Programs compiled for W > 1 have no lane-revised items and no eager arcs: X == next Y, until and point constraints are
    // all wavefront-revised items (device: dev_wide.hpp). 0: some variable is wider than 128 values (no bitset path).
    int W = 1;
    bool device_tabulation = false;  // products in (kBitmapMaxBits, kBitmapMaxBitsDevice] become bitmaps filled in by the device
    void store_tabulated(const std::vector<int32_t> &key, const uint32_t *words, size_t n, long long product);

    int init(const stcsp_problem *p, bool sharded_tags);
    // next set for a leaf of `set` whose first-variables have the given values (in
    // first_vars order); creates and registers the set / transition when unseen.
    int transition(int set, const std::vector<int> &first_vals);
    // Translate AHEAD of need: for every set whose captured variables span at most `max_tuples` value tuples,
    // register the transition of every tuple (and so on for the sets this creates, up to `max_sets` sets), so
    // that the device never has to stop for a translation of such a set (models with `first x` inside
    // arithmetic have |D| sets). Returns the number of transitions added, < 0 on error.
    int pretranslate(long long max_tuples, int max_sets, long long max_total_tuples = 0, double max_seconds = 0);
    int compile(FlatProgram &out);
    int find_tag(int32_t tag) const;
    // serialised registry exchange for sharded runs (every shard must know every set)
    std::vector<int32_t> serialise_set(int set) const;
    int import_set(const int32_t *words, size_t n);

private:
    int push_constraint(HostSet &s, Tree *root);
    void finish_set(HostSet &s);
    int register_set(std::unique_ptr<HostSet> s);
    Tree *translate(HostSet &dst, const Tree *t, const std::map<int, int> &vals);
    Tree *translate_first(HostSet &dst, const Tree *t, const std::map<int, int> &vals);
    int eval_tree(const Tree *t, const std::vector<int> &scope, const int *vals, bool &valid) const;
    void build_entry(const HostCon &c, TableEntry &e);
    struct Guard {
        const Tree *cond;
        bool taken;
    };
    bool split_wide(const Tree *body, std::vector<Guard> &guards, long long limit, std::vector<Tree *> &out, int &budget);
    const std::vector<Tree *> &pieces_of(const HostCon &c, const std::vector<int32_t> &key);
    int compile_expr(const Tree *t, const std::vector<int> &scope, bool guards, std::vector<int32_t> &code, int &depth,
                     int &max_depth, int &mask_depth);
};

}  // namespace stcsp
