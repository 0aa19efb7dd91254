// dev_kernels.hpp -- the kernels: round-based k_expand with its device-side round bookkeeping,
// state-table commit (table_commit / k_commit), outbox packing,
// rehash and the export kernels. Included by engine.hip only.
#pragma once
#include "dev_propagate.hpp"
#include "dev_wide.hpp"
namespace stcsp {
namespace dev {
// ------------------------------------------------------------------ k_expand (round-based)
// expand ONE open node (slot `gw` of this round) with one wavefront
// The kernel context as the node loops see it: a register copy rebuilt from two VGPRs that hold its words
// lane-striped (word w in hot[w >> 6], lane w & 63). Reading a field is a v_readlane (a few cycles, no counter
// to wait for) where the copy in memory costs a scalar load per use -- ~40 per node, each ~200 cycles behind
// s_waitcnt lgkmcnt(0), which also drains every LDS read in flight. The copy is rebuilt at the top of every
// node and again before its outputs are written (the empty asm keeps the compiler from merging the rebuilds
// into one set of 70 scalars that is live, and spilled, across the whole slot).
#ifndef STCSP_CTX_REBUILDS
#define STCSP_CTX_REBUILDS 2
#endif
constexpr int kCtxWords = (int)(sizeof(Ctx) / 4);
static_assert(kCtxWords <= 128 && sizeof(Ctx) % 4 == 0, "Ctx must fit two lane-striped registers");
__device__ __forceinline__ Ctx ctx_from(const uint32_t (&hot)[2]) {
    uint32_t h0 = hot[0], h1 = hot[1];
    asm volatile("" : "+v"(h0), "+v"(h1));
    Ctx cr;
    uint32_t *dst = (uint32_t *)&cr;
#pragma unroll
    for (int w = 0; w < kCtxWords; w++) dst[w] = rdlane(w < 64 ? h0 : h1, w & 63);
    return cr;
}

// the dynamic LDS of the launch, by name: the sibling stack is addressed through it (word offsets) so that its
// accesses are LDS instructions whatever the compiler can or cannot infer about a pointer into it
extern __shared__ __attribute__((aligned(16))) int stcsp_lds[];

// Which parked sibling a slot goes on with when its path has ended. In a DRY round (fewer slots than resident wavefronts: the
// round lasts as long as its longest slot) of the general kernels the OLDEST one: the root of the biggest subtree still waiting,
// the one the rest of the search is most likely to wait for (juggling `_nosym` 10-25 % faster). Otherwise the youngest
// (depth-first: the stack stays short, fewer children travel through the frontier -- digitinvader9's wide rounds lose 7 % with
// oldest-first; the LITE kernels' chains are four expansions long).
#ifndef STCSP_SIB_OLDEST
#define STCSP_SIB_OLDEST 1
#endif
// A new state under the constraint set of the state it comes from starts with the dirty seed N*K + 1 ("fresh"): only the items
// that read the fresh time point are dirty (process_node). 0: every item, as before round 4.
#ifndef STCSP_FRESH_SEED
#define STCSP_FRESH_SEED 1
#endif
template <int DR, int L, bool CS, bool LITE, int W = 1>
__device__ void expand_node(const uint32_t (&hot)[2], const ExpandArgs &a, const Img<L> &P, int gw, int lane, int *lds_vals, int *lds_stk, int *ldom,
                            int sib_off, WaveEnv<DR> &env, bool dry = false) {
    const Ctx c0 = ctx_from(hot);
    const Ctx &c = c0;
    const int r = gw % R, i = gw / R;
    const int take_r = kload(c.plan, (int)(offsetof(Plan, take) / 4) + r);
    if (i >= take_r) return;
    const int count_r = kload(c.plan, (int)(offsetof(Plan, count) / 4) + r);
    // outputs go to another cursor shard than the input's, or a subtree would stay in the region
    // of its root forever; for every i exactly one input region maps to each output region, so
    // an output region receives from at most max(take) wavefronts
    const int ro = (i + r) % R;
    const CtlLayout L_(c.world);
    const unsigned long long t_a = PHASE_NOW();
    (void)t_a;
    const uint32_t *node = a.in_base + ((size_t)r * a.in_cap + (size_t)(count_r - 1 - i)) * c.NS;
    Dom<DR> dom;
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        dom.r[q] = idx < c.NK ? node[4 + idx] : 0u;
    }
    // header word 2: constraint set (low 16 bits) | dirty seed (high 16 bits): 0 = revise every
    // item (fresh state / root), kSeedNone = nothing to revise (re-queued fixpoint), else 1 + the
    // variable whose time-0 domain the parent just bisected -- the parent block was at its
    // fixpoint, so only items reading that word can have lost supports
    NodeHdr hd;
    hd.h0 = rflu(node[0]);
    hd.h1 = rflu(node[1]);
    const uint32_t w2 = rflu(node[2]);
    hd.set = (int)(w2 & kSetMask);
    hd.seed = w2 >> kSetBits;
    hd.expire = rflu(node[3]);
    // A wavefront does not stop after one expansion: it keeps the lower child of a bisection (or the
    // first node of a state its leaf just opened) in registers and expands it too, up to `chain`
    // expansions per slot -- depth-first inside the slot, breadth-first across slots. The planner
    // picks `chain` per round: long chains while the frontier is smaller than the machine (each
    // round costs ~20 us whatever its size), short ones when there are more slots than wavefronts.
    // Chains pay off while the fixed cost of a round (launch gap, image staging, bookkeeping: ~15 us)
    // is comparable to an expansion; after an expensive expansion (digitinvader / juggling revisions
    // run for 100+ us) continuing would only serialise work other wavefronts could take next round,
    // so a slot also stops chaining once it has used `chain_cycles` of this launch.
    const int chain = kload(c.plan, (int)(offsetof(Plan, chain) / 4));
    const unsigned long long chain_cycles = (unsigned long long)(unsigned)kload(c.plan, (int)(offsetof(Plan, chain_heavy) / 4));
    const unsigned long long t_slot = __builtin_amdgcn_s_memtime();
    uint32_t *out_region = a.out_base + (size_t)ro * a.out_cap * c.NS;
    const unsigned long long t_b = PHASE_NOW();
    (void)t_b;
#ifdef STCSP_PHASES
    if (lane == 0) add_stats(c, gw, ST_CYC_LOAD, t_b - t_a);
#endif
    // Sibling stack (LDS, kSibDepth node records): the upper child of a bisection waits here instead of going to the
    // frontier; when the chain's path ends (failure, or a leaf into a known state) before the slot has used its
    // expansions, the slot goes on with the youngest sibling -- depth-first inside the slot. Whatever is left when the
    // slot stops goes to the frontier in one piece. Small rounds (fewer nodes than wavefronts, where a round is one
    // chain of dependent expansions long) get more out of each launch this way: partialorder_14 40 -> 31 rounds.
    // Only chains of more than two expansions use it: parking a child in LDS for a single step is pure overhead.
    int sd = 0;
#ifdef STCSP_PHASES
    // (hipcc 7.2 fails on this one instantiation of the instrumented build -- "Illegal instruction detected:
    // V_CMP_NE_U32_e32 0, $src_shared_base" -- with the stack in it; the product build is not affected)
    const bool use_sib = chain > 2 && !(DR == 4 && L && !LITE);  // (round 3: <4,true,false,false> fails the same way; round 4: the wide kernels too)
#else
    const bool use_sib = chain > 2;
#endif
    // the siblings still waiting + `extra` more records go to the frontier: one cursor bump for all of them
    auto flush_siblings = [&](const Ctx &c, uint32_t extra, uint32_t &pos) -> bool {
        pos = 0;
        if (sd == 0 && extra == 0) return true;
        if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], (uint32_t)sd + extra);
        pos = rflu(pos);
        if (pos + (uint32_t)sd + extra > a.out_cap) {
            env.err = max(env.err, (unsigned)ERR_FLUSH_OVERFLOW);
            return false;
        }
        uint32_t *dst = out_region + (size_t)(pos + extra) * c.NS;  // [pos, pos + extra): the caller's records
        for (int w = lane; w < sd * c.NS; w += 64) dst[w] = (uint32_t)stcsp_lds[sib_off + w];
        STCSP_REJOIN();
        return true;
    };
    for (int step = 1;; step++) {
        BranchOut bo;     // outputs of this expansion only (nothing of them is carried round the loop)
        BranchOutWide bow;  // (W > 1: dev_wide.hpp)
        LeafOut<DR> lo;
        sd = rfl(sd);
        // The header is wave-uniform, but values carried round a loop whose exits the compiler cannot
        // prove uniform are treated as divergent (VGPRs, vector instead of scalar descriptor loads:
        // +30..60 VGPRs and a wavefront of occupancy per SIMD). Pin them to SGPRs every iteration.
        hd.h0 = rflu(hd.h0);
        hd.h1 = rflu(hd.h1);
        hd.set = rfl(hd.set);
        hd.seed = rflu(hd.seed);
        hd.expire = rflu(hd.expire);
        const unsigned long long t_p = PHASE_NOW();
#if STCSP_CTX_REBUILDS >= 2
        int oc;
        {
            const Ctx cn = ctx_from(hot);
            if constexpr (W > 1)
                oc = process_node_wide<DR, W, L>(cn, P, lane, lds_vals, lds_stk, ldom, dom, hd, gw, env, bow, lo);
            else
                oc = process_node<DR, L, CS, LITE>(cn, P, lane, lds_vals, lds_stk, ldom, dom, hd, gw, env, bo, lo);
        }
        const Ctx ce = ctx_from(hot);
        const Ctx &c = ce;
#elif STCSP_CTX_REBUILDS == 1
        static_assert(W == 1, "tuning builds: one-word domains only");
        const Ctx ce = ctx_from(hot);
        const Ctx &c = ce;
        const int oc = process_node<DR, L, CS, LITE>(c, P, lane, lds_vals, lds_stk, ldom, dom, hd, gw, env, bo, lo);
#else
        const int oc = process_node<DR, L, CS, LITE>(c, P, lane, lds_vals, lds_stk, ldom, dom, hd, gw, env, bo, lo);
#endif
        const bool last = step >= chain || __builtin_amdgcn_s_memtime() - t_slot > chain_cycles;
        const unsigned long long t_c = PHASE_NOW();
        (void)t_p;
        (void)t_c;
#ifdef STCSP_PHASES
        // everything after process_node (child stores / commit / candidate) is charged to "commit"
        struct PhaseEnd {
            const Ctx &c; int gw, lane; unsigned long long tp, tc;
            __device__ ~PhaseEnd() {
                if (lane == 0) {
                    const unsigned long long td = PHASE_NOW();
                    add_stats(c, gw, ST_CYC_CLASSIFY, tc - tp);  // whole process_node (incl. sweeps + wavefront revisions)
                    add_stats(c, gw, ST_CYC_COMMIT, td - tc);
                    add_stats(c, gw, ST_CYC_TOTAL, td - tp);
                }
            }
        } phase_end{c, gw, lane, t_p, t_c};
#endif
        // the path of this chain ended here: go on with the youngest sibling, or hand the rest to the frontier
#define STCSP_PATH_END()                                                                   \
    {                                                                                      \
        if (sd == 0) return;                                                               \
        if (last) {                                                                        \
            uint32_t fpos;                                                                 \
            flush_siblings(c, 0u, fpos);                                                   \
            return;                                                                        \
        }                                                                                  \
        sd--;                                                                              \
        const int sib = sib_off + (STCSP_SIB_OLDEST && !LITE && dry ? 0 : sd * c.NS);      \
        hd.h0 = rflu((uint32_t)stcsp_lds[sib]);                                            \
        hd.h1 = rflu((uint32_t)stcsp_lds[sib + 1]);                                        \
        const uint32_t sw2 = rflu((uint32_t)stcsp_lds[sib + 2]);                           \
        hd.set = (int)(sw2 & kSetMask);                                                    \
        hd.seed = sw2 >> kSetBits;                                                         \
        hd.expire = rflu((uint32_t)stcsp_lds[sib + 3]);                                    \
        _Pragma("unroll") for (int q = 0; q < DR; q++) {                                   \
            const int idx = q * 64 + lane;                                                 \
            dom.r[q] = idx < c.NK ? (uint32_t)stcsp_lds[sib + 4 + idx] : 0u;               \
        }                                                                                  \
        if (STCSP_SIB_OLDEST && !LITE && dry) { /* the younger ones move down (reads run ahead of the writes) */ \
            for (int w = lane; w < sd * c.NS; w += 64) {                                   \
                const int x = stcsp_lds[sib_off + c.NS + w];                               \
                stcsp_lds[sib_off + w] = x;                                                \
            }                                                                              \
            STCSP_REJOIN();                                                                \
        }                                                                                  \
        continue;                                                                          \
    }
        if (oc == OC_FAIL) STCSP_PATH_END();
        if (oc == OC_BRANCH) {
            const int bvar = W > 1 ? bow.bvar : bo.bvar;
            const uint32_t cw2 = (uint32_t)hd.set | ((uint32_t)(bvar + 1) << kSetBits);
            Dom<DR> child = dom;
            if constexpr (W > 1) {
                // the bisection point falls into one chunk of the variable: chunks below it go to the lower child whole, those above to the upper one
                const int NK1 = c.N * c.K;
#pragma unroll
                for (int ch = 0; ch < W; ch++) {
                    const int wi = ch * NK1 + bvar, rel = bow.mid - 32 * ch;
                    const uint32_t Dc = dom.get(wi);
                    const uint32_t lowmask = rel >= 31 ? 0xffffffffu : (rel < 0 ? 0u : ((2u << rel) - 1u));
                    child.set(wi, Dc & ~lowmask, lane);
                    dom.set(wi, Dc & lowmask, lane);
                }
            } else {
                child.set(bo.bvar, bo.D & ~bo.lowmask, lane);  // upper half: waits on the sibling stack, or goes to the frontier
                dom.set(bo.bvar, bo.D & bo.lowmask, lane);     // lower half: next in the chain, or stored too
            }
            if (use_sib && !last && sd < c.sib_depth) {
                const int sb = sib_off + sd * c.NS;
                if (lane < 4) stcsp_lds[sb + lane] = (int)(lane == 0 ? hd.h0 : (lane == 1 ? hd.h1 : (lane == 2 ? cw2 : hd.expire)));
#pragma unroll
                for (int q = 0; q < DR; q++) {
                    const int idx = q * 64 + lane;
                    if (idx < c.NK) stcsp_lds[sb + 4 + idx] = (int)child.r[q];
                }
                STCSP_REJOIN();
                sd++;
                hd.seed = (uint32_t)(bvar + 1);
                continue;
            }
            if (last) {  // both children and the waiting siblings
                uint32_t pos;
                if (!flush_siblings(c, 2u, pos)) return;
                store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, child, lane);
                store_node<DR>(out_region + (size_t)(pos + 1) * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, dom, lane);
                return;
            }
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], 1u);
            pos = rflu(pos);
            if (pos + 1u > a.out_cap) {
                env.err = max(env.err, (unsigned)ERR_OUT_OVERFLOW);
                return;
            }
            store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, child, lane);
            hd.seed = (uint32_t)(bvar + 1);
            continue;
        }
        if (oc == OC_MISS) {  // park the (propagated) node again until the host has translated the set
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], 1u);
            pos = rflu(pos);
            if (pos + 1 > a.out_cap) {
                env.err = max(env.err, (unsigned)ERR_OUT_OVERFLOW);
                return;
            }
            store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, (uint32_t)hd.set | (kSeedNone << kSetBits), hd.expire, dom, lane);
            STCSP_PATH_END();
        }
        // leaf
        // sharded runs: a leaf whose successor state belongs to another shard becomes a candidate
        // record for its owner (header, signature, edge label, block); one that belongs to this
        // shard is committed right here like in an unsharded run
        if (c.sharded && (lo.owner != c.rank || c.sharded == 2)) {  // (2: tests send every leaf through the exchange, own ones too)
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.cand0 + (lo.owner * R + ro) * CST], 1u);
            pos = rflu(pos);
            if (pos + 1 > a.cand_cap) {
                env.err = max(env.err, (unsigned)ERR_CAND_OVERFLOW);
                return;
            }
            uint32_t *rec = a.cand_base + ((size_t)(lo.owner * R + ro) * a.cand_cap + pos) * c.CS;
            if (lane < 6)
                rec[lane] = lane == 0 ? hd.h0
                          : (lane == 1 ? hd.h1
                          : (lane == 2 ? lo.next_tag : (lane == 3 ? lo.new_expire : (lane == 4 ? (uint32_t)lo.h : (uint32_t)(lo.h >> 32)))));
            if (lane >= 1 && lane <= c.sig_len) rec[kCandHdr + lane - 1] = lo.kw;
            uint32_t *vals = rec + kCandHdr + c.sig_len;
            uint32_t *blk = vals + c.N;
#pragma unroll
            for (int q = 0; q < DR; q++) {
                int idx = q * 64 + lane;
                if (idx < c.N) vals[idx] = lo.evals[q];
                if (idx < c.NK) blk[idx] = lo.nblk[q];
            }
            STCSP_REJOIN();
            STCSP_PATH_END();
        }
        // commit right here, the leaf's data never leaves the registers
#ifdef STCSP_X_NOCOMMIT
        STCSP_PATH_END();
#endif
        CommitOut co = table_commit<DR>(c, lane, ro, lo.kw, lo.h, hd.h0, hd.h1, lo.next_set, lo.next_tag, lo.evals);
        if (!co.ok) {
            env.err = max(env.err, co.err);
            return;
        }
        if (!co.is_new) STCSP_PATH_END();
        env.n_new++;
        if (last) {
            env.err = max(env.err, emit_state_node<DR>(c, lane, ro, a.out_base, a.out_cap, a.parity, co, lo.new_expire, lo.nblk,
                                                       (STCSP_FRESH_SEED && W == 1 && co.set == hd.set) ? (uint32_t)(c.N * c.K + 1) : 0u));
            uint32_t fpos;
            flush_siblings(c, 0u, fpos);
            return;
        }
        // the leaf opened a new state: its first node is next in the chain
        const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | co.idx;
        hd.h0 = (uint32_t)gid;
        hd.h1 = (uint32_t)(gid >> 32);
        // (a state that runs under the SAME constraint set as the one it comes from starts with the fresh point's items dirty only:
        // process_node, kSeedFresh)
        hd.seed = (STCSP_FRESH_SEED && W == 1 && co.set == hd.set) ? (uint32_t)(c.N * c.K + 1) : 0u;
        hd.set = co.set;
        hd.expire = lo.new_expire;
#pragma unroll
        for (int q = 0; q < DR; q++) dom.r[q] = lo.nblk[q];
    }
#undef STCSP_PATH_END
}

__device__ __forceinline__ uint32_t ald(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Reductions of the round bookkeeping (non-negative counts, sums below 2^31): DPP scans as in
// wave_scan_add, the result is taken from lane 63. All 64 lanes are active at every call.
__device__ __forceinline__ int wave_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ long long wave_sum64(long long v) { return (long long)wave_sum((int)v); }

// The round bookkeeping below is executed by ONE wavefront (lane r looks after cursor region r),
// so that its global-memory reads go out in parallel: a handful of round trips per round.

// Plan the next round from the top of the segment stack. Leaves status != PS_RUN when there is
// nothing to do or the host has to act first (grow a pool, translate a constraint set, look at
// an error).
__device__ __forceinline__ void set_gate(Plan *p, unsigned launch_id, int nslots) {
    __hip_atomic_store(&p->gate, (unsigned long long)launch_id << 32 | (unsigned)nslots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// no round is planned: no upcoming launch may pass (the id of a launch that has already run, no slots)
__device__ __forceinline__ void close_gate(Plan *p, unsigned next_launch) { set_gate(p, next_launch - 1u, 0); }
// `next_launch`: id of the launch that is to execute the round planned here
__device__ void plan_next(const Ctx &c, Plan *p, int lane, unsigned next_launch) {
    const CtlLayout L(c.world);
    const bool rl = lane < R;
    uint32_t flag = 0;
    if (lane < 2) flag = ald(&c.ctl[L.misc0 + (lane == 0 ? MISC_ERROR : MISC_NMISS) * CST]);
    if (__ballot(flag != 0)) {
        if (lane == 0) {
            p->status = PS_HOST;
            close_gate(p, next_launch);
        }
        return;
    }
    int sp = rfl(p->sp);
    unsigned long long arena_top = p->arena_top;
    int cnt = 0;
    while (sp > 0) {  // drop exhausted segments from the top
        cnt = rl ? p->stack[sp - 1].count[lane] : 0;
        if (wave_sum64(cnt) != 0) break;
        arena_top = p->stack[sp - 1].base;
        sp--;
    }
    if (lane == 0) {
        p->sp = sp;
        p->arena_top = arena_top;
    }
    if (sp == 0) {
        if (lane == 0) {
            p->status = PS_DONE;
            close_gate(p, next_launch);
        }
        return;
    }
    int chunk = p->chunk_r;
    int mc = 0;  // sharded: the fullest outbox cursor
    if (c.sharded) {
        for (int k = lane; k < c.world * R; k += 64) mc = max(mc, (int)ald(&c.ctl[L.cand0 + k * CST]));
        mc = wave_max(mc);
        // The outbox bounds the round, not the batch: a slot may meet a leaf in every expansion of its chain and all of a region's
        // slots may send to one owner, so a region takes at most (room / longest chain) nodes. (The outboxes are sized for 4,096
        // nodes per region whatever the batch: engine.hip create.)
        const int longest = p->chain_small > p->chain_big ? p->chain_small : p->chain_big;
        const int room = ((int)p->cand_cap - mc) / longest;
        if (room < chunk) chunk = (room >= 256 || mc == 0) ? max(room, 0) : 0;  // (no dribbling: a nearly full outbox is full)
    }
    const int take = cnt < chunk ? cnt : chunk;
    const int maxtake = wave_max(take);
    const long long taken = wave_sum64(take);
    // expansions per slot this round (see expand_node): a slot emits at most chain + 1 nodes and
    // chain leaves, all into one cursor region that receives from at most maxtake slots
    const int chain = taken <= (long long)p->chain_thresh ? p->chain_small : p->chain_big;
    const int proc = chain;  // nodes one slot may expand in the launch (= leaves, edges, new states it may produce)
    const unsigned out_cap = (unsigned)(proc + 2) * (unsigned)maxtake;
    int status = PS_RUN;
    if (arena_top + (unsigned long long)R * out_cap * c.NS > p->arena_words) status = PS_NEED_ARENA;
    const unsigned max_edges = (unsigned)wave_max(rl ? (int)ald(&c.ctl[L.edge0 + lane * CST]) : 0);
    const unsigned long long ns = rflu(lane == 0 ? ald(&c.ctl[L.misc0 + MISC_NSTATES * CST]) : 0u);
    if (status == PS_RUN && (unsigned long long)max_edges + (unsigned long long)proc * maxtake > p->edge_cap) status = PS_NEED_EDGES;
    if (status == PS_RUN && ns + proc * taken > p->state_cap) status = PS_NEED_STATES;
    if (status == PS_RUN && (ns + proc * taken) * 2 > p->slot_cap) status = PS_NEED_TABLE;
    if (status == PS_RUN && c.sharded && (maxtake == 0 || (unsigned long long)mc + (unsigned long long)proc * maxtake > p->cand_cap))
        status = PS_OUTBOX_FULL;  // no room for even one node per region: the exchange has to empty the outboxes first
    if (status != PS_RUN) {
        if (lane == 0) {
            p->status = status;
            close_gate(p, next_launch);
        }
        return;
    }
    const int parity = rfl(p->parity) ^ 1;
    if (rl) {
        p->take[lane] = take;
        p->count[lane] = cnt;
        c.ctl[L.out(parity, lane)] = 0u;
    }
    if (lane == 0) {
        p->in_base = p->stack[sp - 1].base;
        p->in_cap = p->stack[sp - 1].cap;
        p->out_base = arena_top;
        p->out_cap = out_cap;
        p->nslots = R * maxtake;
        p->parity = parity;
        p->chain = chain;
        p->status = PS_RUN;
        set_gate(p, next_launch, R * maxtake);
    }
}

// Account a finished output segment: read its cursors, push it if non-empty.
__device__ void push_output(const Ctx &c, Plan *p, bool consumed_input, int lane) {
    const CtlLayout L(c.world);
    const bool rl = lane < R;
    const int parity = rfl(p->parity);
    const int sp = rfl(p->sp);
    const int tcount = rl ? (int)ald(&c.ctl[L.out(parity, lane)]) : 0;
    const long long total = wave_sum64(tcount);
    long long taken = 0;
    if (consumed_input) {
        const int tk = rl ? p->take[lane] : 0;
        if (rl) p->stack[sp - 1].count[lane] -= tk;
        taken = wave_sum64(tk);
    }
    if (lane == 0) p->open_total += total - taken;
    if (total > 0) {
        if (sp >= kMaxSegments) {
            if (lane == 0) p->status = PS_STACK_FULL;
            return;
        }
        if (rl) p->stack[sp].count[lane] = tcount;
        if (lane == 0) {
            p->stack[sp].base = p->out_base;
            p->stack[sp].cap = p->out_cap;
            p->sp = sp + 1;
            p->arena_top = p->out_base + (unsigned long long)R * p->out_cap * c.NS;
        }
    }
}

// End of a k_expand round, by the last working workgroup: push_output + plan_next in one pass. The
// two functions above go through ~8 dependent global round trips (5.4 us per round, measured); here
// every value the decision needs is requested up front -- cursors and pool fill levels by the lanes
// that own them, the plan's scalars once -- so the common case costs two round trips. The segment
// being consumed is known without re-reading it (plan.count holds its counts as planned).
struct Verdict {  // what finalize_round decided (wave-uniform): mirrored to the host without reading the plan back
    int status;
    long long rounds, open_total;
};
__device__ Verdict finalize_round(const Ctx &c, Plan *p, int lane, unsigned next_launch, int parity0) {
    const CtlLayout L(c.world);
    const bool rl = lane < R;
    // ---- one batch of independent loads (the round's parity comes from the launch's prologue: the cursor addresses below
    // depend on it, reading it here would put a round trip in front of the batch)
    int sp = p->sp;
    const int tcount = rl ? (int)ald(&c.ctl[L.out(parity0, lane)]) : 0;
    const int tk = rl ? p->take[lane] : 0;
    const int cnt_planned = rl ? p->count[lane] : 0;
    uint32_t flag = 0;
    if (lane < 2) flag = ald(&c.ctl[L.misc0 + (lane == 0 ? MISC_ERROR : MISC_NMISS) * CST]);
    const int edges_r = rl ? (int)ald(&c.ctl[L.edge0 + lane * CST]) : 0;
    const uint32_t ns_l = lane == 0 ? ald(&c.ctl[L.misc0 + MISC_NSTATES * CST]) : 0u;
    int mc = 0;
    if (c.sharded)
        for (int k = lane; k < c.world * R; k += 64) mc = max(mc, (int)ald(&c.ctl[L.cand0 + k * CST]));
    const unsigned long long out_base = p->out_base, arena_words = p->arena_words, slot_cap = p->slot_cap;
    const unsigned out_cap = p->out_cap, edge_cap = p->edge_cap, state_cap = p->state_cap, cand_cap = p->cand_cap;
    const int chunk = p->chunk_r, chain_small = p->chain_small, chain_big = p->chain_big, chain_thresh = p->chain_thresh;
    const long long open_total = p->open_total, rounds = p->rounds;
    unsigned long long arena_top = p->arena_top;
    // ---- account the finished round (push_output)
    const long long total = wave_sum64(tcount), taken0 = wave_sum64(tk);
    int cnt = cnt_planned - tk;  // what is left of the segment this round consumed from
    if (rl) p->stack[sp - 1].count[lane] = cnt;
    Verdict vd{PS_RUN, rounds + 1, open_total + total - taken0};
    if (lane == 0) {
        p->open_total = vd.open_total;
        p->rounds = vd.rounds;
    }
    if (rl) p->edge_seen[lane] = (unsigned)edges_r;
    if (lane == 0) p->states_seen = ns_l;
    if (c.progress) {
        const unsigned long long tag = (unsigned long long)(rounds + 1) << 32;
        if (rl) __hip_atomic_store(&c.progress->edge_seen[lane], tag | (unsigned)edges_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (lane == 0) __hip_atomic_store(&c.progress->states_seen, tag | ns_l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (total > 0) {
        if (sp >= kMaxSegments) {
            if (lane == 0) {
                p->sp = sp;
                p->status = PS_STACK_FULL;
                close_gate(p, next_launch);
            }
            vd.status = PS_STACK_FULL;
            return vd;
        }
        if (rl) p->stack[sp].count[lane] = tcount;
        if (lane == 0) {
            p->stack[sp].base = out_base;
            p->stack[sp].cap = out_cap;
        }
        sp++;
        arena_top = out_base + (unsigned long long)R * out_cap * c.NS;
        cnt = tcount;  // the new segment is the top of the stack
    }
    // ---- plan the next round (plan_next)
    unsigned long long in_base = total > 0 ? out_base : 0ull;
    unsigned in_cap = out_cap;
    bool in_known = total > 0;
    if (__ballot(flag != 0)) {
        if (lane == 0) {
            p->sp = sp;
            p->arena_top = arena_top;
            p->status = PS_HOST;
            close_gate(p, next_launch);
        }
        vd.status = PS_HOST;
        return vd;
    }
    while (sp > 0 && wave_sum64(cnt) == 0) {  // drop exhausted segments from the top (rare: extra round trips)
        arena_top = p->stack[sp - 1].base;
        sp--;
        cnt = (sp > 0 && rl) ? p->stack[sp - 1].count[lane] : 0;
        in_known = false;
    }
    if (lane == 0) {
        p->sp = sp;
        p->arena_top = arena_top;
    }
    if (sp == 0) {
        if (lane == 0) {
            p->status = PS_DONE;
            close_gate(p, next_launch);
        }
        vd.status = PS_DONE;
        return vd;
    }
    if (!in_known) {
        in_base = p->stack[sp - 1].base;
        in_cap = p->stack[sp - 1].cap;
    }
    int chunk_now = chunk;
    const int mcw = c.sharded ? wave_max(mc) : 0;
    if (c.sharded) {  // the outbox bounds the round (see plan_next)
        const int longest = chain_small > chain_big ? chain_small : chain_big;
        const int room = ((int)cand_cap - mcw) / longest;
        if (room < chunk_now) chunk_now = (room >= 256 || mcw == 0) ? max(room, 0) : 0;
    }
    const int take = cnt < chunk_now ? cnt : chunk_now;
    const int maxtake = wave_max(take);
    const long long taken = wave_sum64(take);
    const int chain = taken <= (long long)chain_thresh ? chain_small : chain_big;
    const int proc = chain;  // see plan_next
    const unsigned new_cap = (unsigned)(proc + 2) * (unsigned)maxtake;
    int status = PS_RUN;
    if (arena_top + (unsigned long long)R * new_cap * c.NS > arena_words) status = PS_NEED_ARENA;
    const unsigned max_edges = (unsigned)wave_max(edges_r);
    const unsigned long long ns = rflu(ns_l);
    if (status == PS_RUN && (unsigned long long)max_edges + (unsigned long long)proc * maxtake > edge_cap) status = PS_NEED_EDGES;
    if (status == PS_RUN && ns + proc * taken > state_cap) status = PS_NEED_STATES;
    if (status == PS_RUN && (ns + proc * taken) * 2 > slot_cap) status = PS_NEED_TABLE;
    if (status == PS_RUN && c.sharded && (maxtake == 0 || (unsigned long long)mcw + (unsigned long long)proc * maxtake > cand_cap)) status = PS_OUTBOX_FULL;
    if (status != PS_RUN) {
        if (lane == 0) {
            p->status = status;
            close_gate(p, next_launch);
        }
        vd.status = status;
        return vd;
    }
    const int parity = parity0 ^ 1;
    if (rl) {
        p->take[lane] = take;
        p->count[lane] = cnt;
        c.ctl[L.out(parity, lane)] = 0u;
    }
    if (lane == 0) {
        p->in_base = in_base;
        p->in_cap = in_cap;
        p->out_base = arena_top;
        p->out_cap = new_cap;
        p->nslots = R * maxtake;
        p->parity = parity;
        p->chain = chain;
        p->status = PS_RUN;
        set_gate(p, next_launch, R * maxtake);
    }
    return vd;
}

// the planner's verdict for the host (Progress::status / rounds / open_total [/ counters]); the values come in registers -- reading
// the plan back would be three more dependent round trips at the end of every round
__device__ __forceinline__ void mirror_plan(const Ctx &c, const Verdict &vd, int lane) {
    if (!c.progress) return;
    if (vd.status == PS_DONE && lane < kMirrorCounters) {
        // the search is over: lane k sums work counter k over its slots (every wavefront of the solve has flushed its counters
        // and WAITED for those atomics before its workgroup's barrier: flush_env)
        unsigned long long acc = 0;
        for (int sl = 0; sl < kStatSlots; sl++) acc += __hip_atomic_load(&c.stats[sl * kStatWords + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&c.progress->counters[lane], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    STCSP_REJOIN();
    if (lane == 0) {
        if (vd.status == PS_DONE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the counters before the verdict that announces them)
        __hip_atomic_store(&c.progress->rounds, vd.rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&c.progress->open_total, vd.open_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&c.progress->status, vd.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    STCSP_REJOIN();
}
__global__ void k_replan(Ctx c, unsigned next_launch) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        plan_next(c, c.plan, threadIdx.x, next_launch);
        // (rare: once per burst at most -- the plan words are read back, lane 0's stores drained first)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Verdict vd;
        vd.status = rfl(__hip_atomic_load(&c.plan->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        vd.rounds = __hip_atomic_load(&c.plan->rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        vd.open_total = __hip_atomic_load(&c.plan->open_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mirror_plan(c, vd, threadIdx.x);
    }
}
// sharded commit: open an output segment of `cap` slots per region / close it again
__global__ void k_open_segment(Ctx c, unsigned cap) {
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        Plan *p = c.plan;
        const CtlLayout L(c.world);
        const int parity = rfl(p->parity) ^ 1;
        if (threadIdx.x < R) c.ctl[L.out(parity, threadIdx.x)] = 0u;
        if (threadIdx.x == 0) {
            p->out_base = p->arena_top;
            p->out_cap = cap;
            p->parity = parity;
        }
    }
}
__global__ void k_close_segment(Ctx c) {
    if (blockIdx.x == 0 && threadIdx.x < 64) push_output(c, c.plan, false, threadIdx.x);
}

// Each workgroup first stages the program image into LDS (when L), then its four wavefronts
// loop over the round's node slots with a grid stride; the last workgroup to finish accounts
// the round and plans the next one.
#ifndef STCSP_EXPAND_WAVES
#define STCSP_EXPAND_WAVES 1
#endif
// Register budget: 4 wavefronts/SIMD for the 1- and 2-register blocks, 3 for the 4-register one (the
// allocator lands a few registers above those limits otherwise and loses a whole wavefront per SIMD;
// the handful of spills this forces sit in cold paths). Ctx is read through a pointer (scalar loads on demand): passing it by value kept ~130 SGPRs
// live/spilled and cost a wavefront of occupancy per SIMD.
#ifndef STCSP_LITE_WAVES
#define STCSP_LITE_WAVES 6
#endif
#ifndef STCSP_GEN_WAVES
#define STCSP_GEN_WAVES 4
#endif
#ifndef STCSP_WIDE_WAVES
#define STCSP_WIDE_WAVES 4
#endif
// BIG: one workgroup of 1024 threads (16 wavefronts = 4 per SIMD) per CU shares ONE staged copy of the program -- for LITE
// programs whose tables do not fit beside four 256-thread workgroups' copies (the synthetic 64 x 32 family: 122 KB of sweep
// records, dirty rows and tables). Same code; the workgroup size is read from blockDim.
#ifndef STCSP_BIG_WAVES
#define STCSP_BIG_WAVES 16  // wavefronts of a big workgroup
#endif
template <int DR, int L, bool CS, bool LITE, bool BIG = false, int W = 1>
__global__ __launch_bounds__(BIG ? STCSP_BIG_WAVES * 64 : 256, BIG ? 1 : (STCSP_EXPAND_WAVES > 1 ? STCSP_EXPAND_WAVES : (LITE && DR == 1 ? STCSP_LITE_WAVES : (DR <= 2 ? STCSP_GEN_WAVES : STCSP_WIDE_WAVES)))) void k_expand(const Ctx *__restrict__ cp, const Plan *__restrict__ plan_arg, unsigned launch_id, uint32_t tab_gen) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    // the planned round's gate in ONE 8-byte read (Plan::gate): is it this launch's round, and how many slots has it? The plan
    // pointer is a kernel argument of its own (not read through *cp), so this is the first and only load a workgroup without
    // work waits for. A planner that stops (done, pool full, host needed) leaves the gate on the launch that has just run: the
    // rest of the burst fails this test -- no separate look at the status word.
    static_assert(offsetof(Plan, gate) == 0, "the gate is the plan's first word");
    // ---- ONE batch of independent loads before anything is waited for: the gate, the plan words of the round (through the plan
    // ARGUMENT, not through the pointer inside *cp), the context words the prologue needs and the register copy of the context.
    // (Round 3's prologue was a chain of a dozen dependent scalar loads -- gate, then *cp, then cp->progress, then cp->plan, then
    // plan->rounds, then stage_words, then img, ... -- eight of them first touches of a cache line after the launch boundary, on
    // the critical path of every round. The empty asm below pins the batch in front of the gate test.)
    const kptr pk = (kptr)(const __attribute__((address_space(1))) int *)plan_arg;
    const unsigned long long gate = ((const __attribute__((address_space(4))) unsigned long long *)(const __attribute__((address_space(1))) unsigned long long *)plan_arg)[0];
    auto pl = [&](size_t off) { return (uint32_t)pk[(int)(off / 4)]; };
    const uint32_t p_in_lo = pl(offsetof(Plan, in_base)), p_in_hi = pl(offsetof(Plan, in_base) + 4);
    const uint32_t p_out_lo = pl(offsetof(Plan, out_base)), p_out_hi = pl(offsetof(Plan, out_base) + 4);
    const uint32_t p_in_cap = pl(offsetof(Plan, in_cap)), p_out_cap = pl(offsetof(Plan, out_cap)), p_cand_cap = pl(offsetof(Plan, cand_cap));
    const uint32_t p_parity = pl(offsetof(Plan, parity)), p_rounds = pl(offsetof(Plan, rounds));
    const int c_stage_words = c.stage_words, c_NK = c.NK, c_stack_slots = c.stack_slots, c_sib_depth = c.sib_depth, c_world = c.world;
    const uint32_t *const c_img = c.img;
    uint32_t *const c_arena = c.arena, *const c_cand = c.cand, *const c_ctl = c.ctl;
    Progress *const c_progress = c.progress;
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t hot[2];
#pragma unroll
    for (int q = 0; q < 2; q++) hot[q] = q * 64 + lane < kCtxWords ? ((const uint32_t *)cp)[q * 64 + lane] : 0u;
    asm volatile("" ::"s"(p_in_lo), "s"(p_in_hi), "s"(p_out_lo), "s"(p_out_hi), "s"(p_in_cap), "s"(p_out_cap), "s"(p_cand_cap), "s"(p_parity), "s"(p_rounds),
                 "s"(c_stage_words), "s"(c_NK), "s"(c_stack_slots), "s"(c_sib_depth), "s"(c_world), "s"(c_img), "s"(c_arena), "s"(c_cand), "s"(c_ctl), "s"(c_progress));
    if ((unsigned)(gate >> 32) != launch_id) return;  // another launch's round (this workgroup is late, or the burst ran past a stop)
    const int n_slots = (int)(unsigned)gate;
    const int wpb = BIG ? STCSP_BIG_WAVES : 4;  // wavefronts per workgroup
    // workgroups without a node slot leave at once; the ticket below counts the working ones only
    if ((int)blockIdx.x * wpb >= n_slots) return;
    const unsigned n_working = (unsigned)min((n_slots + wpb - 1) / wpb, (int)gridDim.x);
    if (blockIdx.x == 0 && threadIdx.x == 0 && c_progress)  // (the streaming export's "the round before this one has ended")
        __hip_atomic_store(&c_progress->started, (unsigned long long)(p_rounds + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int img_words = (c_stage_words + 3) & ~3;  // L: the whole image; else a prefix of hot sections (or 0)
    const unsigned long long t_k0 = PHASE_NOW();
    (void)t_k0;
    if (img_words) {
        const uint4 *src = (const uint4 *)c_img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += (BIG ? STCSP_BIG_WAVES * 64 : 256)) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = wave_scratch_words(c_NK, c_stack_slots, LITE, c_sib_depth);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    int *ldom = LITE ? lds_vals : lds_stk + c_stack_slots * 64;  // NK-word AND-accumulator of this wavefront, then its counters
    const int sib_off = img_words + wib * per_wave + wave_sib_offset(c_NK, c_stack_slots, LITE);  // word offset in the launch's LDS
    Img<L> P{c_img, (const uint32_t *)smem, c_stage_words};
    ExpandArgs a;
    a.in_base = c_arena + ((unsigned long long)p_in_hi << 32 | p_in_lo);
    a.in_cap = p_in_cap;
    a.out_base = c_arena + ((unsigned long long)p_out_hi << 32 | p_out_lo);
    a.out_cap = p_out_cap;
    a.cand_base = c_cand;
    a.cand_cap = p_cand_cap;
    a.parity = (int)p_parity;
    const int total_waves = gridDim.x * wpb;
    const unsigned long long t_k1 = PHASE_NOW();
    (void)t_k1;
    WaveEnv<DR> env;
    {
        // the state table's generation changes with every solve; the device copy of the context does not have to: the launch
        // brings it along and it goes straight into the register copy the node loops read
        constexpr int gw_ = (int)(offsetof(Ctx, tab_gen) / 4);
        static_assert(gw_ < 64, "tab_gen sits in the first register of the context copy");
        if (lane == gw_) hot[0] = tab_gen;
    }
#ifdef STCSP_STATIC_SLOTS
    for (int gw = blockIdx.x * wpb + wib; gw < n_slots; gw += total_waves) expand_node<DR, L, CS, LITE, W>(hot, a, P, gw, lane, lds_vals, lds_stk, ldom, sib_off, env);
#else
    // Slots: the first one by position, every further one by ticket -- slots differ widely in cost (a chain of up to `chain`
    // expansions, each anything between a failed sweep and a leaf with a new state), and with a fixed stride the round waits for
    // the wavefront whose share happened to be the dearest. The ticket for the NEXT slot is requested before the current one is
    // expanded (its latency disappears behind the node load); kSlotCursors counters deal interleaved tickets.
    {
        const CtlLayout L_(c_world);
        const int ncur = min(kSlotCursors, (int)gridDim.x);  // (a grid smaller than the counters: every residue needs a workgroup)
        const int cur = (int)blockIdx.x % ncur;
        uint32_t *cursor = c_ctl + L_.slotcur0 + cur * CST;
        for (int gw = blockIdx.x * wpb + wib; gw < n_slots;) {
            unsigned ticket = 0;
            if (lane == 0) ticket = atomicAdd(cursor, 1u);
            expand_node<DR, L, CS, LITE, W>(hot, a, P, gw, lane, lds_vals, lds_stk, ldom, sib_off, env, n_slots <= total_waves);
            gw = total_waves + (int)rflu(ticket) * ncur + cur;
        }
    }
#endif
    flush_env<DR>(c, env, blockIdx.x * wpb + wib, lane);
    __syncthreads();
#ifdef STCSP_PHASES
    if (threadIdx.x == 0) {
        add_stats(c, blockIdx.x, ST_CYC_STAGE, t_k1 - t_k0);
        add_stats(c, blockIdx.x, ST_BLOCKS, 1);
        add_stats(c, blockIdx.x, ST_CYC_BLOCK, PHASE_NOW() - t_k0);
    }
#endif
    if (wib == 0) {
        // No fence on either side of the ticket: what the finalizing wavefront reads of the other workgroups are cursors and
        // counters, all of them written by returning agent-scope atomics that completed before the workgroup's barrier
        // above; the node and edge records themselves are only read by LATER launches (a kernel boundary away). An
        // agent-scope fence here costs 2-3.5 us on the critical path of every round (MI355X_MICROARCH.md, fence table).
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(&c.plan->done_blocks, 1u);
        if (rflu(t) == n_working - 1) {  // last working workgroup: every cursor of this round is final
            if (lane == 0) __hip_atomic_store(&c.plan->done_blocks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef STCSP_STATIC_SLOTS
            // (every other wavefront of the launch has drawn its last ticket: the counters start the next round at zero)
            if (lane < kSlotCursors) __hip_atomic_store(&c.ctl[CtlLayout(c.world).slotcur0 + lane * CST], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            const unsigned long long t_k2 = PHASE_NOW();
            (void)t_k2;
            const Verdict vd = finalize_round(c, c.plan, lane, launch_id + 1u, a.parity);
            mirror_plan(c, vd, lane);
#ifdef STCSP_PHASES
            if (lane == 0) {
                add_stats(c, 0, ST_CYC_FINAL, PHASE_NOW() - t_k2);
                add_stats(c, 0, ST_ROUNDS_FINAL, 1);
            }
#endif
        }
    }
}

// ------------------------------------------------------------------ k_probe (tests / diagnostics)
// process_node on caller-provided blocks: the kernel-granularity check behind stcsp_engine_propagate.
// One wavefront per block, every item dirty (seed 0); the propagated block and the outcome go back.
template <int DR, int L, bool CS, bool LITE>
__global__ __launch_bounds__(256, (DR <= 2 ? 4 : 3)) void k_probe(const Ctx *__restrict__ cp, uint32_t *blocks, int n, int set, uint32_t expire,
                                                                  int *outcome) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int img_words = (c.stage_words + 3) & ~3;
    if (img_words) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = wave_scratch_words(c.NK, c.stack_slots, LITE, c.sib_depth);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    int *ldom = LITE ? lds_vals : lds_stk + c.stack_slots * 64;
    Img<L> P{c.img, (const uint32_t *)smem, c.stage_words};
    WaveEnv<DR> env;
    for (int gw = blockIdx.x * 4 + wib; gw < n; gw += gridDim.x * 4) {
        uint32_t *blk = blocks + (size_t)gw * c.NK;
        Dom<DR> dom;
#pragma unroll
        for (int q = 0; q < DR; q++) {
            const int idx = q * 64 + lane;
            dom.r[q] = idx < c.NK ? blk[idx] : 0u;
        }
        NodeHdr hd;
        hd.h0 = hd.h1 = 0u;
        hd.set = rfl(set >= 0 ? set : gw);  // (set < 0: block i under constraint set i, and no work counters -- engine.hip fresh_init)
        hd.seed = 0u;
        hd.expire = rflu(expire);
        BranchOut bo;
        LeafOut<DR> lo;
        const int oc = process_node<DR, L, CS, LITE>(c, P, lane, lds_vals, lds_stk, ldom, dom, hd, gw, env, bo, lo);
#pragma unroll
        for (int q = 0; q < DR; q++) {
            const int idx = q * 64 + lane;
            if (idx < c.NK) blk[idx] = dom.r[q];
        }
        if (lane == 0) outcome[gw] = oc;
    }
    if (set >= 0) flush_env<DR>(c, env, blockIdx.x * 4 + wib, lane);
}

// ------------------------------------------------------------------ commit
// Lookup-or-insert the state (set tag, signature) held lane-striped in `kw` (lane j = key word j)
// and append the edge record (label `vals`, lane-striped like the domain block).
// Role of vertexTableGetVertex / vertexNew + vertexTableAddVertex / edgeNew + vertexAddEdge
// (reference src/graph.cpp:14-38, 78-89, 108-123).
template <int DR>
__device__ CommitOut table_commit(const Ctx &c, int lane, int ro, uint32_t kw, unsigned long long h, uint32_t s0, uint32_t s1,
                                  int set, uint32_t tag, const uint32_t (&vals)[DR]) {
    const CtlLayout L(c.world);
    uint32_t *misc = c.ctl + L.misc0;
    CommitOut out;
    out.idx = 0;
    out.is_new = false;
    out.ok = false;
    out.set = set;
    out.err = 0;
    const uint32_t gen = c.tab_gen;
    const int esz = 1 << c.tab_shift;
    uint32_t pos = (uint32_t)h & c.slot_mask;
    uint32_t idx = 0;
    bool is_new = false;
    // the edge slot is needed whatever the lookup finds: request it now, use it after the probe
    uint32_t e = 0;
    if (lane == 0) e = atomicAdd(&c.ctl[L.edge0 + ro * CST], 1u);
    for (unsigned probes = 0;; probes++) {
        uint32_t *ent = (uint32_t *)c.slots + ((size_t)pos << c.tab_shift);
        unsigned long long *sw = (unsigned long long *)(ent + esz - 2);
        // ONE round trip: the slot word and the key of the entry (same 128-byte line), the slot word requested first -- a
        // publisher writes the key, drains its stores, then the index, so a reader that sees the index sees the key
        unsigned long long sv = 0;
        if (lane == 0) sv = __hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t other = 0;
        if (lane < c.KL) other = __hip_atomic_load(&ent[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t lo = rflu((uint32_t)sv), hi = rflu((uint32_t)(sv >> 32));
        if (hi != gen) {  // free (empty, or left over from an earlier solve): claim it
            bool claimed = false;
            if (lane == 0) claimed = atomicCAS(sw, sv, ((unsigned long long)gen << 32) | kPending) == sv;
            if (!__ballot(claimed)) {  // somebody else was faster: look at the entry again
                if (probes > c.slot_mask) {
                    out.err = ERR_STATE_OVERFLOW;
                    return out;
                }
                continue;
            }
            // claimed: allocate the state, publish its key (in the entry; and in state_keys for the export), then the index
            uint32_t ni = 0;
            if (lane == 0) ni = atomicAdd(&misc[MISC_NSTATES * CST], 1u);
            ni = rflu(ni);
            if (ni >= c.state_cap) {
                out.err = ERR_STATE_OVERFLOW;
                return out;
            }
            if (lane < c.KL) {
                __hip_atomic_store(&ent[lane], kw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.state_keys[(size_t)ni * c.KL + lane] = kw;  // (read by later launches and the export kernels only)
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(sw, ((unsigned long long)gen << 32) | ni, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            STCSP_REJOIN();
            idx = ni;
            is_new = true;
            break;
        }
        unsigned spins = 0;
        while (lo == kPending) {  // another wavefront is publishing this entry: read slot word and key again
            __builtin_amdgcn_s_sleep(2);
            unsigned long long t = 0;
            if (lane == 0) t = __hip_atomic_load(sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < c.KL) other = __hip_atomic_load(&ent[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo = rflu((uint32_t)t);
            if (++spins > (1u << 22)) {
                out.err = ERR_TABLE_SPIN;
                return out;
            }
        }
        if (!__ballot(lane < c.KL && other != kw)) {
            idx = lo;
            break;
        }
        // another state's entry. (Belt and braces: a key that differs is read once more before the probe moves on -- a
        // duplicate state would be a wrong automaton, a second read of a colliding entry costs a round trip on a path that
        // tables at most half full rarely take.)
        if (lane < c.KL) other = __hip_atomic_load(&ent[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!__ballot(lane < c.KL && other != kw)) {
            idx = lo;
            break;
        }
        pos = (pos + 1) & c.slot_mask;
        if (probes > c.slot_mask) {
            out.err = ERR_STATE_OVERFLOW;
            return out;
        }
    }
    // edge record: src (global id), dst (local index), label = time-0 value of every variable
    e = rflu(e);
    if (e >= c.edge_cap) {
        out.err = ERR_EDGE_OVERFLOW;
        return out;
    }
    uint32_t *er = c.edges + ((size_t)ro * c.edge_cap + e) * c.ES;
    if (lane < 4) er[lane] = lane == 0 ? s0 : (lane == 1 ? s1 : (lane == 2 ? idx : 0u));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int k = q * 64 + lane;
        if (k < c.N) er[4 + k] = vals[q];
    }
    STCSP_REJOIN();
    out.idx = idx;
    out.is_new = is_new;
    out.ok = true;
    if (is_new) {
        if (set < 0) {  // sharded: the record names the set by tag
            for (int t = 0; t < c.nsets && set < 0; t++)
                if ((uint32_t)kload(c.img, c.o.sets + t * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4)) == tag) set = t;
            if (set < 0) {
                out.err = ERR_UNKNOWN_SET;
                out.ok = false;
            }
        }
        out.set = set;
    }
    return out;
}

// new state: open its first search node in the round's output segment
template <int DR>
__device__ unsigned emit_state_node(const Ctx &c, int lane, int ro, uint32_t *out_base, uint32_t out_cap, int parity,
                                    const CommitOut &co, uint32_t expire, const uint32_t (&blk)[DR], uint32_t seed) {
    const CtlLayout L(c.world);
    uint32_t np = 0;
    if (lane == 0) np = atomicAdd(&c.ctl[L.out(parity, ro)], 1u);
    np = rflu(np);
    if (np + 1 > out_cap) return ERR_OUT_OVERFLOW;
    uint32_t *dst = out_base + ((size_t)ro * out_cap + np) * c.NS;
    const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | co.idx;
    if (lane < 4) dst[lane] = lane == 0 ? (uint32_t)gid : (lane == 1 ? (uint32_t)(gid >> 32) : (lane == 2 ? ((uint32_t)co.set | seed << kSetBits) : expire));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int k = q * 64 + lane;
        if (k < c.NK) dst[4 + k] = blk[q];
    }
    STCSP_REJOIN();
    return 0u;
}

// ------------------------------------------------------------------ k_commit (sharded runs)
template <int DR>
__global__ __launch_bounds__(256) void k_commit(Ctx c, CommitArgs a) {
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long long gw = (long long)blockIdx.x * 4 + wib;
    if (gw >= a.total) return;
    const int ro = (int)(gw % R);  // cursor shard for this wavefront's outputs (edge record, new node)
    const uint32_t *rec = a.cand_base + (size_t)gw * c.CS;
    const Plan *p = c.plan;
    uint32_t hw = lane < 6 ? rec[lane] : 0u;  // one coalesced header read
    const uint32_t s0 = rdlane(hw, 0), s1 = rdlane(hw, 1), tag = rdlane(hw, 2), expire = rdlane(hw, 3);
    const unsigned long long h = ((unsigned long long)rdlane(hw, 5) << 32) | rdlane(hw, 4);  // computed by k_expand
    uint32_t kw = 0;
    if (lane == 0) kw = tag;
    if (lane >= 1 && lane <= c.sig_len) kw = rec[kCandHdr + lane - 1];
    const uint32_t *pv = rec + kCandHdr + c.sig_len, *pb = pv + c.N;
    uint32_t vals[DR], blk[DR];
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int k = q * 64 + lane;
        vals[q] = k < c.N ? pv[k] : 0u;
        blk[q] = k < c.NK ? pb[k] : 0u;
    }
    CommitOut co = table_commit<DR>(c, lane, ro, kw, h, s0, s1, -1, tag, vals);
    unsigned err = co.ok ? 0u : co.err;
    if (co.ok && co.is_new) {
        if (lane == 0) add_stats(c, (int)(gw & 0x7fffffff), ST_NEWSTATES, 1);
        err = emit_state_node<DR>(c, lane, ro, c.arena + p->out_base, p->out_cap, p->parity, co, expire, blk, 0u);
        if (err == ERR_OUT_OVERFLOW) err = ERR_COMMIT_OUT_OVERFLOW;
    }
    if (err && lane == 0) atomicMax(&c.ctl[CtlLayout(c.world).misc0 + MISC_ERROR * CST], (uint32_t)err);
}

// gather the R regions of one owner's outbox into a contiguous array (for the all-to-all)
__global__ void k_pack(const uint32_t *cand_base, uint32_t cand_cap, int CS, const uint32_t *ctl, int cursor_base,
                       uint32_t *dst) {
    __shared__ uint32_t pref[R + 1];
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int r = 0; r < R; r++) {
            pref[r] = acc;
            acc += ctl[cursor_base + r * CST];
        }
        pref[R] = acc;
    }
    __syncthreads();
    const size_t total_words = (size_t)pref[R] * CS;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += (size_t)gridDim.x * blockDim.x) {
        uint32_t recno = (uint32_t)(w / CS), off = (uint32_t)(w % CS);
        int r = 0;
        while (recno >= pref[r + 1]) r++;
        dst[w] = cand_base[((size_t)r * cand_cap + (recno - pref[r])) * CS + off];
    }
}

// ... the same for EVERY owner's outbox in one launch (blockIdx.y = owner): a world of 8 used to cost eight small launches per superstep
struct PackAllArgs {
    uint32_t before[64];  // records in front of owner p's part of the packed buffer
};
__global__ void k_pack_all(const uint32_t *cand_base, uint32_t cand_cap, int CS, const uint32_t *ctl, int cand0, PackAllArgs a, uint32_t *dst_base) {
    __shared__ uint32_t pref[R + 1];
    const int peer = blockIdx.y;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int r = 0; r < R; r++) {
            pref[r] = acc;
            acc += ctl[cand0 + (peer * R + r) * CST];
        }
        pref[R] = acc;
    }
    __syncthreads();
    const uint32_t *src = cand_base + (size_t)peer * R * cand_cap * CS;
    uint32_t *dst = dst_base + (size_t)a.before[peer] * CS;
    const size_t total_words = (size_t)pref[R] * CS;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += (size_t)gridDim.x * blockDim.x) {
        uint32_t recno = (uint32_t)(w / CS), off = (uint32_t)(w % CS);
        int r = 0;
        while (recno >= pref[r + 1]) r++;
        dst[w] = src[((size_t)r * cand_cap + (recno - pref[r])) * CS + off];
    }
}

// ------------------------------------------------------------------ frontier redistribution (sharded runs)
// k_donate: the top `take[r]` records of every region of ONE frontier segment leave as transfer records
// (device_types.hpp xfer_stride: constraint set named by tag, dirty seed in a word of its own); the segment's
// counts and the open-node total of the device plan are reduced by what left. One wavefront per record.
struct DonateArgs {
    unsigned long long seg_base;  // word offset of the segment in the arena
    unsigned seg_cap;
    int seg;                      // its index in the plan's segment stack
    int count[R], take[R];
    uint32_t pref[R + 1];         // prefix sums of take[]
};
__global__ __launch_bounds__(256) void k_donate(Ctx c, DonateArgs a, uint32_t *out) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t gw = blockIdx.x * 4 + wib;
    if (blockIdx.x == 0 && threadIdx.x < R) {  // account for what leaves
        c.plan->stack[a.seg].count[threadIdx.x] = a.count[threadIdx.x] - a.take[threadIdx.x];
        if (threadIdx.x == 0) c.plan->open_total -= (long long)a.pref[R];
    }
    if (gw >= a.pref[R]) return;
    int r = 0;
#pragma unroll
    for (int step = R / 2; step >= 1; step >>= 1)
        if (gw >= a.pref[r + step]) r += step;
    const uint32_t i = gw - a.pref[r];
    const uint32_t *node = c.arena + a.seg_base + ((size_t)r * a.seg_cap + (size_t)(a.count[r] - 1 - (int)i)) * c.NS;
    const int TS = xfer_stride(c.N, c.K * c.W);
    uint32_t *rec = out + (size_t)gw * TS;
    const uint32_t w2 = node[2];
    const uint32_t tag = (uint32_t)kload(c.img, c.o.sets + (int)(w2 & kSetMask) * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4));
    if (lane < kXferHdr) rec[lane] = lane == 0 ? node[0] : (lane == 1 ? node[1] : (lane == 2 ? tag : (lane == 3 ? node[3] : (lane == 4 ? w2 >> kSetBits : 0u))));
    for (int k = lane; k < c.NK; k += 64) rec[kXferHdr + k] = node[4 + k];
}
// k_adopt: received transfer records become open nodes of the segment k_open_segment has just opened
__global__ __launch_bounds__(256) void k_adopt(Ctx c, const uint32_t *recs, long long total) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const long long gw = (long long)blockIdx.x * 4 + wib;
    if (gw >= total) return;
    const CtlLayout L(c.world);
    const Plan *p = c.plan;
    const int TS = xfer_stride(c.N, c.K * c.W);
    const uint32_t *rec = recs + (size_t)gw * TS;
    const uint32_t tag = rec[2];
    int set = -1;
    for (int t = 0; t < c.nsets && set < 0; t++)
        if ((uint32_t)kload(c.img, c.o.sets + t * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4)) == tag) set = t;
    const int ro = (int)(gw % R);
    uint32_t np = 0;
    if (lane == 0) np = atomicAdd(&c.ctl[L.out(p->parity, ro)], 1u);
    np = rflu(np);
    if (set < 0 || np + 1 > p->out_cap) {
        if (lane == 0) atomicMax(&c.ctl[L.misc0 + MISC_ERROR * CST], (uint32_t)(set < 0 ? ERR_UNKNOWN_SET : ERR_ADOPT_OVERFLOW));
        return;
    }
    uint32_t *dst = c.arena + p->out_base + ((size_t)ro * p->out_cap + np) * c.NS;
    if (lane < 4) dst[lane] = lane == 0 ? rec[0] : (lane == 1 ? rec[1] : (lane == 2 ? ((uint32_t)set | rec[4] << kSetBits) : rec[3]));
    for (int k = lane; k < c.NK; k += 64) dst[4 + k] = rec[kXferHdr + k];
}

// ------------------------------------------------------------------ k_tabulate
// Tuple bitmap of one point constraint, filled in on the device: one THREAD per tuple of the full initial
// product (mixed radix over the scope, first variable fastest -- the stride order of ConDesc::stride_off), each
// running the constraint's postfix program with the semantics of solverValidateRe (reference
// src/solveralgorithm.cpp:336-424; the same interpreter as eval_program, scalar, stack in registers/scratch). The
// host tabulates products up to kBitmapMaxBits (~0.1 us per tuple); an 8-ary constraint over 11-value domains has
// 35 M tuples -- seconds on the host, about a millisecond here -- and without its bitmap every revision of it
// interprets the program block by block (digitinvader9: 88 % of all revision cycles).
struct TabArgs {
    const int *code;       // the program
    const int *arr_off;    // array table (offsets / data)
    const int *arr_data;
    long long product;
    int scope_len, uses_valid;
    int size[kMaxScope > 16 ? 16 : kMaxScope];  // (constraints over more than 16 variables are never tabulated: 2^16 < product)
    int lb[16];
};
__global__ __launch_bounds__(256) void k_tabulate(TabArgs a, uint32_t *bitmap) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    int vals[16];
    {
        uint32_t rem = t < a.product ? (uint32_t)t : 0u;  // product <= kBitmapMaxBitsDevice = 2^28
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < a.scope_len) {
                const uint32_t q = rem / (uint32_t)a.size[j];
                vals[j] = a.lb[j] + (int)(rem - q * (uint32_t)a.size[j]);
                rem = q;
            }
    }
    int stk[kTabulateMaxStack];  // (cset.cpp build_entry leaves deeper programs to the interpreter of the wavefront revision)
    static_assert(kTabulateMaxStack == 32, "the stack indices below wrap at 32");
    int sp = 0, tos = 0;
    bool valid = true;
    uint32_t dead = 0;
    for (int pc = 0;;) {
        const int w = a.code[pc++], op = w & 255, arg = w >> 8;
        if (op == OP_END) break;
        switch (op) {
            case OP_CONST: stk[sp++ & 31] = tos; tos = a.code[pc++]; break;
            case OP_VAR: stk[sp++ & 31] = tos; tos = vals[arg & 15]; break;
            case OP_ARR: {
                const int off = a.arr_off[arg], size = a.arr_off[arg + 1] - off;
                const bool inr = (unsigned)tos < (unsigned)size;
                if (!inr && dead == 0) valid = false;
                tos = inr ? a.arr_data[off + tos] : 0;
                break;
            }
            case OP_ABS: tos = tos < 0 ? (int)(0u - (unsigned)tos) : tos; break;
            case OP_NOT: tos = (tos == 0); break;
            case OP_MASK_T:
            case OP_MASK_F: {
                const int v = arg == 0 ? tos : stk[(sp - arg) & 31];
                const bool live = (op == OP_MASK_T) ? (v != 0) : (v == 0);
                dead = (dead << 1) | (live ? 0u : 1u);
                break;
            }
            case OP_MASK_POP: dead >>= 1; break;
            case OP_SEL_IF: {
                const int b = tos, x = stk[(sp - 1) & 31], cnd = stk[(sp - 2) & 31];
                sp -= 2;
                tos = cnd ? x : b;
                break;
            }
            case OP_SEL_AND: { const int x = stk[--sp & 31]; tos = x ? tos : 0; break; }
            case OP_SEL_OR: { const int x = stk[--sp & 31]; tos = x ? 1 : tos; break; }
            case OP_SEL_IMPLY: { const int x = stk[--sp & 31]; tos = (x == 0) ? 1 : (x <= tos); break; }
            default: {
                const int b = tos, x = stk[--sp & 31];
                int r = 0;
                switch (op) {
                    case OP_ADD: r = (int)((unsigned)x + (unsigned)b); break;
                    case OP_SUB: r = (int)((unsigned)x - (unsigned)b); break;
                    case OP_MUL: r = (int)((unsigned)x * (unsigned)b); break;
                    case OP_DIV: r = (b == 0 || (x == INT_MIN && b == -1)) ? 0 : x / b; break;
                    case OP_MOD: r = (b == 0 || (x == INT_MIN && b == -1)) ? 0 : x % b; break;
                    case OP_LT: r = x < b; break;
                    case OP_GT: r = x > b; break;
                    case OP_LE: r = x <= b; break;
                    case OP_GE: r = x >= b; break;
                    case OP_EQ: r = x == b; break;
                    case OP_NE: r = x != b; break;
                    default: break;
                }
                tos = (a.uses_valid && !valid) ? 0 : r;
            }
        }
    }
    const unsigned long long m = __ballot(t < a.product && tos != 0);
    const int lane = threadIdx.x & 63;
    const long long w0 = (t - lane) >> 5;  // the wavefront's two bitmap words
    if (lane == 0 && t - lane < a.product) bitmap[w0] = (uint32_t)m;
    if (lane == 32 && t - lane + 32 < a.product) bitmap[w0 + 1] = (uint32_t)(m >> 32);
}

// Start of a solve in ONE launch: zero the control block, the statistics and the out-degree mirror; the plan header (with the
// pool capacities), the root's table entry, its key and its search node come out of a pinned host buffer the kernel reads
// directly. (Round 3: three memsets and seven small host-to-device copies, each its own command in front of the first k_expand.)
struct BeginArgs {
    uint32_t *ctl;
    int ctl_words, nstates_word;
    unsigned long long *stats;
    int stats_words;
    uint32_t *sdeg;
    unsigned long long sdeg_words;
    uint32_t *plan_dst;
    int plan_words;          // header + first segment, in 32-bit words
    const uint32_t *stage;   // pinned host memory: [plan words | root entry (esz words) | root node (NS words)]
    uint32_t *entry_dst;     // null: this rank does not hold the root
    int esz;
    uint32_t *keys_dst;
    int KL;
    uint32_t *node_dst;
    int NS;
};
__global__ __launch_bounds__(256) void k_begin(BeginArgs a) {
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = tid; i < (unsigned long long)a.ctl_words; i += nth) a.ctl[i] = (a.entry_dst && (int)i == a.nstates_word) ? 1u : 0u;
    for (unsigned long long i = tid; i < (unsigned long long)a.stats_words; i += nth) a.stats[i] = 0ull;
    for (unsigned long long i = tid; i < a.sdeg_words; i += nth) a.sdeg[i] = 0u;
    for (unsigned long long i = tid; i < (unsigned long long)a.plan_words; i += nth) a.plan_dst[i] = a.stage[i];
    if (a.entry_dst) {
        const uint32_t *ent = a.stage + a.plan_words, *node = ent + a.esz;
        for (unsigned long long i = tid; i < (unsigned long long)a.esz; i += nth) a.entry_dst[i] = ent[i];
        for (unsigned long long i = tid; i < (unsigned long long)a.KL; i += nth) a.keys_dst[i] = ent[i];
        for (unsigned long long i = tid; i < (unsigned long long)a.NS; i += nth) a.node_dst[i] = node[i];
    }
}

// re-insert every state into a larger table
__global__ void k_rehash(Ctx c, uint32_t n_states) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    const uint32_t *key = c.state_keys + (size_t)i * c.KL;
    const unsigned long long h = key_hash(key, c.KL);
    const int esz = 1 << c.tab_shift;
    uint32_t pos = (uint32_t)h & c.slot_mask;
    const unsigned long long want = ((unsigned long long)c.tab_gen << 32) | i;
    for (;;) {  // (the new table is zeroed: generation 0 is never a solve's)
        uint32_t *ent = (uint32_t *)c.slots + ((size_t)pos << c.tab_shift);
        if (atomicCAS((unsigned long long *)(ent + esz - 2), 0ull, want) == 0ull) {
            for (int j = 0; j < c.KL; j++) ent[j] = key[j];
            return;
        }
        pos = (pos + 1) & c.slot_mask;
    }
}

// ------------------------------------------------------------------ export (unsharded runs)
// The reference's ok/fail bookkeeping (src/solveralgorithm.cpp:857-874, 904-909) as an
// edge-parallel fixpoint on the device (the host twin is okfix.hpp): repeatedly mark every
// non-root state without a live out-edge as failed and kill the edges into it. Then the live
// edges are compacted into structure-of-arrays buffers, so the host copies exactly the result
// arrays of the C-ABI (no per-edge work on the host).
struct EdgeView {
    const uint32_t *edges;
    uint32_t edge_cap;
    int ES, N;
    uint32_t pref[R + 1];  // prefix sums of the per-region record counts
};
__device__ __forceinline__ const uint32_t *edge_at(const EdgeView &v, uint32_t e) {
    int r = 0;
#pragma unroll
    for (int step = R / 2; step >= 1; step >>= 1)
        if (e >= v.pref[r + step]) r += step;
    return v.edges + ((size_t)r * v.edge_cap + (e - v.pref[r])) * v.ES;
}
__global__ void k_post_outdeg(EdgeView v, uint32_t *outdeg, uint8_t *alive) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.pref[R]) return;
    const uint32_t *er = edge_at(v, e);
    alive[e] = 1;
    atomicAdd(&outdeg[er[0]], 1u);  // unsharded: the global id is the local index
}
// First round of the ok-fixpoint when the whole log was streamed: every state's flag (0 / 1) goes to the device array AND straight to
// the pinned host array, the "somebody failed" word to pinned memory too -- no memsets in front, no copies behind.
__global__ void k_post_mark_host(uint32_t n_states, const uint32_t *outdeg, uint8_t *fail, uint8_t *hfail, uint32_t *hchanged) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_states) return;
    const uint8_t f = (s != 0 && outdeg[s] == 0) ? 1 : 0;  // the root is never marked (solveralgorithm.cpp:967-971)
    fail[s] = f;
    hfail[s] = f;
    if (f) *hchanged = 1u;
}
__global__ void k_post_mark(uint32_t n_states, const uint32_t *outdeg, uint8_t *fail, uint32_t *changed) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s == 0 || s >= n_states) return;  // the root is never marked (solveralgorithm.cpp:967-971)
    if (!fail[s] && outdeg[s] == 0) {
        fail[s] = 1;
        *changed = 1u;
    }
}
__global__ void k_post_kill(EdgeView v, uint8_t *alive, const uint8_t *fail, uint32_t *outdeg) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= v.pref[R] || !alive[e]) return;
    const uint32_t *er = edge_at(v, e);
    if (fail[er[2]]) {
        alive[e] = 0;
        atomicSub(&outdeg[er[0]], 1u);
    }
}
// Compaction of the live edges into the structure-of-arrays result (a streaming pass: HBM-bound). One
// wavefront takes 64 consecutive raw records; its live ones get consecutive output slots (one atomic per
// wavefront), so their labels form ONE contiguous span of nlive * N words that the lanes write word by word
// (consecutive lanes -> consecutive addresses) while reading the source records through a rank -> record
// table in LDS (consecutive lanes read consecutive label words of one record).
__global__ __launch_bounds__(256) void k_post_compact(EdgeView v, const uint8_t *alive, uint32_t *counter, long long *osrc,
                                                       long long *odst, int32_t *oval) {
    __shared__ unsigned long long tab[4][64];  // per wavefront: word offset of the record of live rank r
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < v.pref[R] && alive[e];
    const unsigned long long m = __ballot(live);
    const int nlive = __popcll(m);
    if (nlive == 0) return;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(counter, (uint32_t)nlive);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (live) {
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        const uint32_t *er = edge_at(v, e);
        tab[wib][rank] = (unsigned long long)(er - v.edges);
        osrc[base + rank] = (long long)(((unsigned long long)er[1] << 32) | er[0]);
        odst[base + rank] = (long long)er[2];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int N = v.N, total = nlive * N;
    const int q64 = 64 / N, r64 = 64 % N;
    int r = lane / N, k = lane - r * N;
    int32_t *out = oval + (size_t)base * N;
    for (int w = lane; w < total; w += 64) {
        out[w] = (int32_t)v.edges[tab[wib][r] + 4 + k];
        r += q64;
        k += r64;
        if (k >= N) {
            k -= N;
            r++;
        }
    }
}

// Streaming export: the edge records [from[r], from[r] + n[r]) of every region -- written by launches that have
// completed -- are transposed into the structure-of-arrays result at positions base, base + 1, ... while the
// search goes on (another stream); the host copies each chunk out as soon as it is staged. Whether an edge
// survives the ok/fail fixpoint is only known at the end: if no state fails (the usual case) the streamed arrays
// ARE the result, otherwise the engine falls back to the compacting export.
struct StreamView {
    const uint32_t *edges;
    uint32_t edge_cap;
    int ES, N;
    uint32_t from[R];
    uint32_t pref[R + 1];  // prefix sums of the per-region record counts of this chunk
};
// ... and the keys of the states [from, to) split into the result's constraint-set id and signature arrays
// (`cid`, `sig`: pinned host memory the device writes straight into; a state's key never changes once committed)
__global__ __launch_bounds__(256) void k_stream_keys(const uint32_t *keys, int KL, int sl, uint32_t from, uint32_t to, int32_t *cid, int32_t *sig) {
    const unsigned long long w = (unsigned long long)from * KL + blockIdx.x * 256ull + threadIdx.x;
    if (w >= (unsigned long long)to * KL) return;
    const uint32_t i = (uint32_t)(w / (unsigned)KL);
    const int j = (int)(w - (unsigned long long)i * KL);
    const uint32_t v = keys[w];
    if (j == 0)
        cid[i] = v == kRootTag ? 0 : (int32_t)v;
    else if (j - 1 < sl)
        sig[(size_t)i * sl + (j - 1)] = (int32_t)v;
}
// (hsrc / hdst / hval: the pinned host arrays, or null. The LAST chunk of an export, when it is small, is written to the host by the
// kernel itself beside the device copy -- one command instead of a kernel and three copies behind it; big chunks go by the copy
// engine: a kernel that writes megabytes over the link holds its wave slots at the link's speed.)
__global__ __launch_bounds__(256) void k_stream_edges(StreamView v, unsigned long long base, unsigned long long dst_or, long long *osrc, long long *odst,
                                                      int32_t *oval, uint32_t *sdeg, long long *hsrc, long long *hdst, int32_t *hval) {
    __shared__ unsigned long long tab[4][64];
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const uint32_t e0 = e - (uint32_t)lane;
    if (e0 >= v.pref[R]) return;
    const bool in = e < v.pref[R];
    if (in) {
        int r = 0;
#pragma unroll
        for (int step = R / 2; step >= 1; step >>= 1)
            if (e >= v.pref[r + step]) r += step;
        const uint32_t *er = v.edges + ((size_t)r * v.edge_cap + v.from[r] + (e - v.pref[r])) * v.ES;
        tab[wib][lane] = (unsigned long long)(er - v.edges);
        const long long es = (long long)(((unsigned long long)er[1] << 32) | er[0]);
        const long long ed = (long long)(dst_or | er[2]);  // sharded runs: the local index becomes a global id (rank bits)
        osrc[base + e] = es;
        odst[base + e] = ed;
        if (hsrc) {
            hsrc[base + e] = es;
            hdst[base + e] = ed;
        }
        if (sdeg) atomicAdd(&sdeg[er[0]], 1u);         // unsharded: out-degrees for the ok-fixpoint's first round (the global id is the local index)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int nrec = (int)min(64u, v.pref[R] - e0);
    const int N = v.N, total = nrec * N;
    const int q64 = 64 / N, r64 = 64 % N;
    int r = lane / N, k = lane - r * N;
    int32_t *out = oval + (size_t)(base + e0) * N;
    int32_t *hout = hval ? hval + (size_t)(base + e0) * N : nullptr;
    for (int w = lane; w < total; w += 64) {
        const int32_t val = (int32_t)v.edges[tab[wib][r] + 4 + k];
        out[w] = val;
        if (hout) hout[w] = val;
        r += q64;
        k += r64;
        if (k >= N) {
            k -= N;
            r++;
        }
    }
}

}  // namespace dev
}  // namespace stcsp
