// dev_kernels.hpp -- the kernels: round-based k_expand with its device-side round bookkeeping,
// the experimental persistent k_persist, state-table commit (table_commit / k_commit), outbox packing,
// rehash and the export kernels. Included by engine.hip only.
#pragma once
#include "dev_propagate.hpp"
namespace stcsp {
namespace dev {
// ------------------------------------------------------------------ k_expand (round-based)
// expand ONE open node (slot `gw` of this round) with one wavefront
template <int DR, bool L, bool CS>
__device__ void expand_node(const Ctx &c, const ExpandArgs &a, const Img<L> &P, int gw, int lane, int *lds_vals, int *lds_stk) {
    const int r = gw % R, i = gw / R;
    const int take_r = kload(c.plan, (int)(offsetof(Plan, take) / 4) + r);
    if (i >= take_r) return;
    const int count_r = kload(c.plan, (int)(offsetof(Plan, count) / 4) + r);
    // outputs go to another cursor shard than the input's, or a subtree would stay in the region
    // of its root forever; for every i exactly one input region maps to each output region, so
    // an output region receives from at most max(take) wavefronts
    const int ro = (i + r) % R;
    const CtlLayout L_(c.world);
    uint32_t *misc = c.ctl + L_.misc0;
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
    // item (fresh state / root), 0xffff = nothing to revise (re-queued fixpoint), else 1 + the
    // variable whose time-0 domain the parent just bisected -- the parent block was at its
    // fixpoint, so only items reading that word can have lost supports
    NodeHdr hd;
    hd.h0 = rflu(node[0]);
    hd.h1 = rflu(node[1]);
    const uint32_t w2 = rflu(node[2]);
    hd.set = (int)(w2 & 0xffffu);
    hd.seed = w2 >> 16;
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
    for (int step = 1;; step++) {
        BranchOut bo;     // outputs of this expansion only (nothing of them is carried round the loop)
        LeafOut<DR> lo;
        // The header is wave-uniform, but values carried round a loop whose exits the compiler cannot
        // prove uniform are treated as divergent (VGPRs, vector instead of scalar descriptor loads:
        // +30..60 VGPRs and a wavefront of occupancy per SIMD). Pin them to SGPRs every iteration.
        hd.h0 = rflu(hd.h0);
        hd.h1 = rflu(hd.h1);
        hd.set = rfl(hd.set);
        hd.seed = rflu(hd.seed);
        hd.expire = rflu(hd.expire);
        const unsigned long long t_p = PHASE_NOW();
        const int oc = process_node<DR, L, CS>(c, P, lane, lds_vals, lds_stk, dom, hd, gw, bo, lo);
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
        if (oc == OC_FAIL) return;
        if (oc == OC_BRANCH) {
            const uint32_t n_out = last ? 2u : 1u;
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], n_out);
            pos = rflu(pos);
            if (pos + n_out > a.out_cap) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_OUT_OVERFLOW);
                return;
            }
            const uint32_t cw2 = (uint32_t)hd.set | ((uint32_t)(bo.bvar + 1) << 16);
            Dom<DR> child = dom;
            child.set(bo.bvar, bo.D & ~bo.lowmask, lane);  // upper half: always to the frontier
            store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, child, lane);
            dom.set(bo.bvar, bo.D & bo.lowmask, lane);     // lower half: next in the chain, or stored too
            if (last) {
                store_node<DR>(out_region + (size_t)(pos + 1) * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, dom, lane);
                return;
            }
            hd.seed = (uint32_t)(bo.bvar + 1);
            continue;
        }
        if (oc == OC_MISS) {  // park the (propagated) node again until the host has translated the set
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], 1u);
            pos = rflu(pos);
            if (pos + 1 > a.out_cap) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_OUT_OVERFLOW);
                return;
            }
            store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, (uint32_t)hd.set | 0xffff0000u, hd.expire, dom, lane);
            return;
        }
        // leaf
        // sharded runs: a leaf whose successor state belongs to another shard becomes a candidate
        // record for its owner (header, signature, edge label, block); one that belongs to this
        // shard is committed right here like in an unsharded run
        if (c.sharded && lo.owner != c.rank) {
            uint32_t pos = 0;
            if (lane == 0) pos = atomicAdd(&c.ctl[L_.cand0 + (lo.owner * R + ro) * CST], 1u);
            pos = rflu(pos);
            if (pos + 1 > a.cand_cap) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_CAND_OVERFLOW);
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
            return;
        }
        // commit right here, the leaf's data never leaves the registers
        CommitOut co = table_commit<DR>(c, lane, ro, lo.kw, lo.h, hd.h0, hd.h1, lo.next_set, lo.next_tag, lo.evals, gw);
        if (!(co.ok && co.is_new)) return;
        if (last) {
            emit_state_node<DR>(c, lane, ro, a.out_base, a.out_cap, a.parity, co, lo.new_expire, lo.nblk);
            return;
        }
        // the leaf opened a new state: its first node is next in the chain
        const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | co.idx;
        hd.h0 = (uint32_t)gid;
        hd.h1 = (uint32_t)(gid >> 32);
        hd.set = co.set;
        hd.seed = 0;
        hd.expire = lo.new_expire;
#pragma unroll
        for (int q = 0; q < DR; q++) dom.r[q] = lo.nblk[q];
    }
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
__device__ void plan_next(const Ctx &c, Plan *p, int lane) {
    const CtlLayout L(c.world);
    const bool rl = lane < R;
    uint32_t flag = 0;
    if (lane < 2) flag = ald(&c.ctl[L.misc0 + (lane == 0 ? MISC_ERROR : MISC_NMISS) * CST]);
    if (__ballot(flag != 0)) {
        if (lane == 0) p->status = PS_HOST;
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
        if (lane == 0) p->status = PS_DONE;
        return;
    }
    const int chunk = p->chunk_r;
    const int take = cnt < chunk ? cnt : chunk;
    const int maxtake = wave_max(take);
    const long long taken = wave_sum64(take);
    // expansions per slot this round (see expand_node): a slot emits at most chain + 1 nodes and
    // chain leaves, all into one cursor region that receives from at most maxtake slots
    const int chain = taken <= (long long)p->chain_thresh ? p->chain_small : p->chain_big;
    const unsigned out_cap = (unsigned)(chain + 2) * (unsigned)maxtake;
    int status = PS_RUN;
    if (arena_top + (unsigned long long)R * out_cap * c.NS > p->arena_words) status = PS_NEED_ARENA;
    const unsigned max_edges = (unsigned)wave_max(rl ? (int)ald(&c.ctl[L.edge0 + lane * CST]) : 0);
    const unsigned long long ns = rflu(lane == 0 ? ald(&c.ctl[L.misc0 + MISC_NSTATES * CST]) : 0u);
    if (status == PS_RUN && (unsigned long long)max_edges + (unsigned long long)chain * maxtake > p->edge_cap) status = PS_NEED_EDGES;
    if (status == PS_RUN && ns + chain * taken > p->state_cap) status = PS_NEED_STATES;
    if (status == PS_RUN && (ns + chain * taken) * 2 > p->slot_cap) status = PS_NEED_TABLE;
    if (status == PS_RUN && c.sharded) {
        int mc = 0;
        for (int k = lane; k < c.world * R; k += 64) mc = max(mc, (int)ald(&c.ctl[L.cand0 + k * CST]));
        mc = wave_max(mc);
        if ((unsigned long long)mc + maxtake > p->cand_cap) status = PS_OUTBOX_FULL;  // a slot ends at its first leaf
    }
    if (status != PS_RUN) {
        if (lane == 0) p->status = status;
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
__device__ void finalize_round(const Ctx &c, Plan *p, int lane) {
    const CtlLayout L(c.world);
    const bool rl = lane < R;
    // ---- one batch of independent loads
    const int parity0 = p->parity;
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
    if (lane == 0) {
        p->open_total = open_total + total - taken0;
        p->rounds = rounds + 1;
    }
    if (total > 0) {
        if (sp >= kMaxSegments) {
            if (lane == 0) {
                p->sp = sp;
                p->status = PS_STACK_FULL;
            }
            return;
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
        }
        return;
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
        if (lane == 0) p->status = PS_DONE;
        return;
    }
    if (!in_known) {
        in_base = p->stack[sp - 1].base;
        in_cap = p->stack[sp - 1].cap;
    }
    const int take = cnt < chunk ? cnt : chunk;
    const int maxtake = wave_max(take);
    const long long taken = wave_sum64(take);
    const int chain = taken <= (long long)chain_thresh ? chain_small : chain_big;
    const unsigned new_cap = (unsigned)(chain + 2) * (unsigned)maxtake;
    int status = PS_RUN;
    if (arena_top + (unsigned long long)R * new_cap * c.NS > arena_words) status = PS_NEED_ARENA;
    const unsigned max_edges = (unsigned)wave_max(edges_r);
    const unsigned long long ns = rflu(ns_l);
    if (status == PS_RUN && (unsigned long long)max_edges + (unsigned long long)chain * maxtake > edge_cap) status = PS_NEED_EDGES;
    if (status == PS_RUN && ns + chain * taken > state_cap) status = PS_NEED_STATES;
    if (status == PS_RUN && (ns + chain * taken) * 2 > slot_cap) status = PS_NEED_TABLE;
    if (status == PS_RUN && c.sharded && (unsigned long long)wave_max(mc) + maxtake > cand_cap) status = PS_OUTBOX_FULL;
    if (status != PS_RUN) {
        if (lane == 0) p->status = status;
        return;
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
    }
}

__global__ void k_replan(Ctx c) {
    if (blockIdx.x == 0 && threadIdx.x < 64) plan_next(c, c.plan, threadIdx.x);
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
template <int DR, bool L, bool CS>
__global__ __launch_bounds__(256, (STCSP_EXPAND_WAVES > 1 ? STCSP_EXPAND_WAVES : (DR <= 2 ? 4 : 3))) void k_expand(const Ctx *__restrict__ cp) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    if (kload(c.plan, (int)(offsetof(Plan, status) / 4)) != PS_RUN) return;  // the burst ran past the end
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int n_slots = kload(c.plan, (int)(offsetof(Plan, nslots) / 4));
    // workgroups without a node slot leave at once; the ticket below counts the working ones only
    if ((int)blockIdx.x * 4 >= n_slots) return;
    const unsigned n_working = (unsigned)min((n_slots + 3) / 4, (int)gridDim.x);
    const int img_words = (c.stage_words + 3) & ~3;  // L: the whole image; else a prefix of hot sections (or 0)
    const unsigned long long t_k0 = PHASE_NOW();
    (void)t_k0;
    if (img_words) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = (kMaxLowVars + c.stack_slots) * 64 + ((c.NK + kLdsStatWords + 63) & ~63);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    Img<L> P{c.img, (const uint32_t *)smem, c.stage_words};
    ExpandArgs a;
    {
        const Plan *p = c.plan;
        a.in_base = c.arena + p->in_base;
        a.in_cap = p->in_cap;
        a.out_base = c.arena + p->out_base;
        a.out_cap = p->out_cap;
        a.cand_base = c.cand;
        a.cand_cap = p->cand_cap;
        a.parity = p->parity;
    }
    const int total_waves = gridDim.x * 4;
    const unsigned long long t_k1 = PHASE_NOW();
    (void)t_k1;
    int *lstat = lds_stk + c.stack_slots * 64 + c.NK;  // this wavefront's counters (process_node: lstat_add)
    if (lane < kLdsStatWords) lstat[lane] = 0;
    for (int gw = blockIdx.x * 4 + wib; gw < n_slots; gw += total_waves) expand_node<DR, L, CS>(c, a, P, gw, lane, lds_vals, lds_stk);
    flush_lds_stats(c, lstat, blockIdx.x * 4 + wib, lane);
    __syncthreads();
#ifdef STCSP_PHASES
    if (threadIdx.x == 0) {
        add_stats(c, blockIdx.x, ST_CYC_STAGE, t_k1 - t_k0);
        add_stats(c, blockIdx.x, ST_BLOCKS, 1);
        add_stats(c, blockIdx.x, ST_CYC_BLOCK, PHASE_NOW() - t_k0);
    }
#endif
    if (wib == 0) {
        unsigned t = 0;
        if (lane == 0) {
            __threadfence();
            t = atomicAdd(&c.plan->done_blocks, 1u);
        }
        if (rflu(t) == n_working - 1) {  // last working workgroup: every cursor of this round is final
            if (lane == 0) c.plan->done_blocks = 0;
            __threadfence();
            const unsigned long long t_k2 = PHASE_NOW();
            (void)t_k2;
            finalize_round(c, c.plan, lane);
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
template <int DR, bool L, bool CS>
__global__ __launch_bounds__(256, (DR <= 2 ? 4 : 3)) void k_probe(const Ctx *__restrict__ cp, uint32_t *blocks, int n, int set, uint32_t expire,
                                                                  int *outcome) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int img_words = (c.stage_words + 3) & ~3;
    if (img_words) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = (kMaxLowVars + c.stack_slots) * 64 + ((c.NK + kLdsStatWords + 63) & ~63);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    Img<L> P{c.img, (const uint32_t *)smem, c.stage_words};
    int *lstat = lds_stk + c.stack_slots * 64 + c.NK;
    if (lane < kLdsStatWords) lstat[lane] = 0;
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
        hd.set = rfl(set);
        hd.seed = 0u;
        hd.expire = rflu(expire);
        BranchOut bo;
        LeafOut<DR> lo;
        const int oc = process_node<DR, L, CS>(c, P, lane, lds_vals, lds_stk, dom, hd, gw, bo, lo);
#pragma unroll
        for (int q = 0; q < DR; q++) {
            const int idx = q * 64 + lane;
            if (idx < c.NK) blk[idx] = dom.r[q];
        }
        if (lane == 0) outcome[gw] = oc;
    }
    flush_lds_stats(c, lstat, blockIdx.x * 4 + wib, lane);
}

// ------------------------------------------------------------------ k_persist (experimental, opt-in)
// STATUS: correct (parity-tested) but slower than the round-based default at full occupancy:
// measured on partialorder_14, 64 wavefronts run at the round-based per-node cost, 5,120 take
// ~190 ms -- the agent-scope loads of thousands of idle pollers serialise on the ring's hot cache
// lines (~21 M operations/s whatever the back-off) and the producers' atomics queue behind them.
// A competitive version needs sharded rings / per-CU wake-ups.
//
// Persistent work-queue variant of the search (unsharded runs): no rounds, no host in the loop.
// Every wavefront runs depth-first: after a bisection it keeps the lower child in registers and
// puts the upper child on its PRIVATE stack (its own slice of HBM); after a leaf that opened a
// new state it continues with that state's first node. Work is shared through a bounded
// multi-producer/multi-consumer ring (sequence-number protocol): a busy wavefront pushes a child
// there instead of on its private stack while the ring is "hungry", idle wavefronts pop from it.
// Termination: PQ_PENDING counts tasks (= ring items) that were pushed and are not finished yet;
// a pusher increments it, the wavefront that popped a task decrements it once the task and all of
// its private descendants are done; idle wavefronts leave when it is 0. A failed pop touches no
// counter. Every spin is bounded and a global abort word ends the launch.
__device__ __forceinline__ uint32_t aldw(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astw(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int DR>
__device__ __forceinline__ void ring_store(const Ctx &c, uint32_t *rec, const NodeHdr &hd, const Dom<DR> &dom, int lane) {
    // agent-scope (write-through) stores: the record is read by another CU
    if (lane < 4) astw(&rec[lane], lane == 0 ? hd.h0 : (lane == 1 ? hd.h1 : (lane == 2 ? ((uint32_t)hd.set | (hd.seed << 16)) : hd.expire)));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        if (idx < c.NK) astw(&rec[4 + idx], dom.r[q]);
    }
}
template <int DR>
__device__ __forceinline__ void ring_load(const Ctx &c, const uint32_t *rec, NodeHdr &hd, Dom<DR> &dom, int lane) {
    uint32_t hw = lane < 4 ? aldw(&rec[lane]) : 0u;
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        dom.r[q] = idx < c.NK ? aldw(&rec[4 + idx]) : 0u;
    }
    hd.h0 = rdlane(hw, 0);
    hd.h1 = rdlane(hw, 1);
    const uint32_t w2 = rdlane(hw, 2);
    hd.set = (int)(w2 & 0xffffu);
    hd.seed = w2 >> 16;
    hd.expire = rdlane(hw, 3);
}

// push one node record on the shared ring. The producer takes its slot with ONE fetch-add (a
// CAS loop here turns into an O(contenders^2) retry storm when many wavefronts share at once);
// the sharing policy keeps the ring far from full, so the slot is normally free at once --
// otherwise wait (bounded) for the consumer of the previous lap. false = gave up (abort set).
template <int DR>
__device__ bool q_push(const Ctx &c, const NodeHdr &hd, const Dom<DR> &dom, int lane) {
    uint32_t pos = 0;
    int ok = 1;
    if (lane == 0) {
        atomicAdd(&c.pq[PQ_PENDING], 1u);
        pos = atomicAdd(&c.pq[PQ_TAIL], 1u);
        unsigned spins = 0;
        while (aldw(&c.seq[pos & c.qmask]) != pos) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 24)) {
                atomicMax(&c.pq[PQ_ABORT], (uint32_t)AB_QUEUE_FULL);
                ok = 0;
                break;
            }
        }
    }
    if (!rfl(ok)) return false;
    pos = rflu(pos);
    ring_store<DR>(c, c.ring + (size_t)(pos & c.qmask) * c.NS, hd, dom, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing lane is in this wavefront
    if (lane == 0) astw(&c.seq[pos & c.qmask], pos + 1);
    return true;
}

// pop one node record; false when the ring looks empty OR another consumer won the race (the
// caller backs off; no hot retry). No counter is touched either way: the popped task stays
// counted in PQ_PENDING until its wavefront has finished it.
template <int DR>
__device__ bool q_pop(const Ctx &c, NodeHdr &hd, Dom<DR> &dom, int lane) {
    uint32_t pos = 0;
    int got = 0;
    if (lane == 0) {
        pos = aldw(&c.pq[PQ_HEAD]);
        if (aldw(&c.seq[pos & c.qmask]) == pos + 1) got = atomicCAS(&c.pq[PQ_HEAD], pos, pos + 1) == pos;
    }
    if (!rfl(got)) return false;
    pos = rflu(pos);
    ring_load<DR>(c, c.ring + (size_t)(pos & c.qmask) * c.NS, hd, dom, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) astw(&c.seq[pos & c.qmask], pos + c.qmask + 1);  // slot free for the next lap
    return true;
}

#ifndef STCSP_PERSIST_WAVES
#define STCSP_PERSIST_WAVES 5
#endif
template <int DR, bool L>
__global__ __launch_bounds__(256, STCSP_PERSIST_WAVES) void k_persist(const Ctx *__restrict__ cp) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int img_words = (c.stage_words + 3) & ~3;  // L: the whole image; else a prefix of hot sections (or 0)
    const unsigned long long t_k0 = PHASE_NOW();
    (void)t_k0;
    if (img_words) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = (kMaxLowVars + c.stack_slots) * 64 + ((c.NK + kLdsStatWords + 63) & ~63);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    Img<L> P{c.img, (const uint32_t *)smem, c.stage_words};
    const int wid = blockIdx.x * 4 + wib;
    int *lstat = lds_stk + c.stack_slots * 64 + c.NK;  // this wavefront's counters (process_node: lstat_add)
    if (lane < kLdsStatWords) lstat[lane] = 0;
    uint32_t *mystack = c.pstack + (size_t)wid * c.pstk_cap * c.NS;
    const CtlLayout L_(c.world);
    uint32_t *misc = c.ctl + L_.misc0;

    int sp = 0;            // private stack depth
    bool have = false;     // a node is in registers
    bool counted = false;  // this wavefront is counted in PQ_ACTIVE
    Dom<DR> dom;
    NodeHdr hd{};
    unsigned long long dbg_idle = 0, dbg_busy = 0, dbg_t = __builtin_amdgcn_s_memtime();
    unsigned dbg_push = 0, dbg_pop = 0, dbg_polls = 0, dbg_ppop = 0;
    unsigned polls = 0, nodes_done = 0;
    // pollers are staggered: each wavefront starts at its own point of the back-off range
    unsigned backoff = 1u + ((unsigned)wid * 2654435761u >> 26);  // 1..64 us
    uint32_t last_tail = 0;
    bool empty_seen = false;
    for (;;) {
        if (!have) {
            if (sp > 0) {  // next sibling from the private stack (own stores: plain accesses)
                sp--;
                dbg_ppop++;
                const uint32_t *rec = mystack + (size_t)sp * c.NS;
                uint32_t hw = lane < 4 ? rec[lane] : 0u;
#pragma unroll
                for (int q = 0; q < DR; q++) {
                    int idx = q * 64 + lane;
                    dom.r[q] = idx < c.NK ? rec[4 + idx] : 0u;
                }
                hd.h0 = rdlane(hw, 0);
                hd.h1 = rdlane(hw, 1);
                const uint32_t w2 = rdlane(hw, 2);
                hd.set = (int)(w2 & 0xffffu);
                hd.seed = w2 >> 16;
                hd.expire = rdlane(hw, 3);
                have = true;
            } else {
                if (counted) {  // the task I popped (and everything below it that I kept) is done
                    if (lane == 0) atomicSub(&c.pq[PQ_PENDING], 1u);
                    counted = false;
                    unsigned long long now = __builtin_amdgcn_s_memtime();
                    dbg_busy += now - dbg_t;
                    dbg_t = now;
                }
                // Idle polling must be gentle: thousands of wavefronts hammering the same L2 lines
                // with agent-scope loads starve the producers. While the tail has not moved since
                // the ring was last seen empty there is nothing to pop, so ONE load per poll
                // suffices; polls back off exponentially (1 us .. ~0.2 ms).
                bool try_pop = true;
                if (empty_seen) {
                    uint32_t tl = 0;
                    if (lane == 0) tl = aldw(&c.pq[PQ_TAIL]);
                    tl = rflu(tl);
                    try_pop = tl != last_tail;
                }
                dbg_polls++;
                if (try_pop && q_pop<DR>(c, hd, dom, lane)) {
                    dbg_pop++;
                    {
                        unsigned long long now = __builtin_amdgcn_s_memtime();
                        dbg_idle += now - dbg_t;
                        dbg_t = now;
                    }
                    counted = true;
                    have = true;
                    polls = 0;
                    backoff = 1u + ((unsigned)(wid + nodes_done) * 2654435761u >> 28);  // 1..16 us after work
                    empty_seen = false;
                } else {
                    if (try_pop || (polls & 7u) == 7u) {
                        // (a lost pop race also lands here: the check below re-reads head/tail)
                        // done when no task is pending anywhere
                        uint32_t act = 0, hd_ = 0, tl = 0, stop = 0;
                        if (lane == 0) {
                            act = aldw(&c.pq[PQ_PENDING]);
                            hd_ = aldw(&c.pq[PQ_HEAD]);
                            tl = aldw(&c.pq[PQ_TAIL]);
                            stop = aldw(&c.pq[PQ_ABORT]) | aldw(&misc[MISC_ERROR * CST]);
                        }
                        act = rflu(act);
                        hd_ = rflu(hd_);
                        tl = rflu(tl);
                        if (rflu(stop)) break;
                        if (act == 0) break;
                        if (hd_ == tl) {
                            empty_seen = true;
                            last_tail = tl;
                        } else {
                            empty_seen = false;  // somebody is mid-push/pop: look again soon
                        }
                    }
                    if (++polls > (1u << 22)) {  // ~10 minutes of nothing: give up loudly
                        if (lane == 0) atomicMax(&c.pq[PQ_ABORT], (uint32_t)AB_SPIN);
                        break;
                    }
                    for (unsigned k = 0; k < backoff; k++) __builtin_amdgcn_s_sleep(40);  // ~1 us each
                    if (backoff < (unsigned)c.park_cap && backoff < kMaxBackoff) backoff <<= 1;
                    continue;
                }
            }
        }
        // ---- one search node
        if ((nodes_done & 63u) == 63u) {  // a pool overflowed / somebody aborted: stop producing
            uint32_t stop = 0;
            if (lane == 0) stop = aldw(&c.pq[PQ_ABORT]) | aldw(&misc[MISC_ERROR * CST]);
            if (rflu(stop)) break;
        }
        BranchOut bo;
        LeafOut<DR> lo;
        const int oc = process_node<DR, L, false>(c, P, lane, lds_vals, lds_stk, dom, hd, wid + (int)nodes_done, bo, lo);
        nodes_done++;
        if (oc == OC_FAIL) {
            have = false;
        } else if (oc == OC_BRANCH) {
            // upper child: shared ring while it is hungry (or my stack is full), else private stack
            Dom<DR> up = dom;
            up.set(bo.bvar, bo.D & ~bo.lowmask, lane);
            NodeHdr uh = hd;
            uh.seed = (uint32_t)(bo.bvar + 1);
            bool shared = sp >= c.pstk_cap;
            if (!shared && ((nodes_done & 3u) == 0 || nodes_done < 32u)) {
                // share only while there are idle wavefronts that the ring cannot feed yet:
                // in steady state (everybody busy) nothing goes through the shared words at all
                int want = 0;
                if (lane == 0) {
                    const int ql = (int)(aldw(&c.pq[PQ_TAIL]) - aldw(&c.pq[PQ_HEAD]));
                    const int busy = (int)aldw(&c.pq[PQ_PENDING]) - ql;  // tasks held by wavefronts
                    want = ql < c.hungry - busy;                         // hungry = wavefronts in the grid

                }
                shared = rfl(want) != 0;
            }
            if (shared) shared = q_push<DR>(c, uh, up, lane);
            if (shared) dbg_push++;
            if (!shared) {
                if (sp >= c.pstk_cap) {
                    if (lane == 0) atomicMax(&c.pq[PQ_ABORT], (uint32_t)AB_QUEUE_FULL);
                    break;
                }
                store_node<DR>(mystack + (size_t)sp * c.NS, c, uh.h0, uh.h1, (uint32_t)uh.set | (uh.seed << 16), uh.expire, up, lane);
                sp++;
            }
            // continue with the lower child in registers
            dom.set(bo.bvar, bo.D & bo.lowmask, lane);
            hd.seed = (uint32_t)(bo.bvar + 1);
            have = true;
        } else if (oc == OC_MISS) {
            uint32_t pi = 0;
            if (lane == 0) pi = atomicAdd(&c.pq[PQ_PARKED], 1u);
            pi = rflu(pi);
            if ((int)pi >= c.park_cap) {
                if (lane == 0) atomicMax(&c.pq[PQ_ABORT], (uint32_t)AB_PARK_FULL);
                break;
            }
            store_node<DR>(c.parked + (size_t)pi * c.NS, c, hd.h0, hd.h1, (uint32_t)hd.set | 0xffff0000u, hd.expire, dom, lane);
            have = false;
        } else {
            const int ro = (wid + (int)nodes_done) % R;
            CommitOut co = table_commit<DR>(c, lane, ro, lo.kw, lo.h, hd.h0, hd.h1, lo.next_set, lo.next_tag, lo.evals, wid);
            if (!co.ok) break;  // pool overflow: MISC_ERROR is set
            if (co.is_new) {
                // new state: go on with its first node right here
                const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | co.idx;
                hd.h0 = (uint32_t)gid;
                hd.h1 = (uint32_t)(gid >> 32);
                hd.set = co.set;
                hd.seed = 0;
                hd.expire = lo.new_expire;
#pragma unroll
                for (int q = 0; q < DR; q++) dom.r[q] = lo.nblk[q];
                have = true;
            } else {
                have = false;
            }
        }
    }
    if (counted && lane == 0) atomicSub(&c.pq[PQ_PENDING], 1u);
    flush_lds_stats(c, lstat, wid, lane);
    if (lane == 0) {
        dbg_idle += __builtin_amdgcn_s_memtime() - dbg_t;
        add_stats(c, wid, ST_QPUSH, dbg_push);
        add_stats(c, wid, ST_QPOP, dbg_pop);
        add_stats(c, wid, ST_POLLS, dbg_polls);
        add_stats(c, wid, ST_IDLE_CYC, dbg_idle);
        add_stats(c, wid, ST_BUSY_CYC, dbg_busy);
        add_stats(c, wid, ST_PSTACK_POP, dbg_ppop);
        add_stats(c, wid, ST_WAVES_WORKED, nodes_done ? 1 : 0);
    }
}

// ------------------------------------------------------------------ commit
// Lookup-or-insert the state (set tag, signature) held lane-striped in `kw` (lane j = key word j)
// and append the edge record (label `vals`, lane-striped like the domain block).
// Role of vertexTableGetVertex / vertexNew + vertexTableAddVertex / edgeNew + vertexAddEdge
// (reference src/graph.cpp:14-38, 78-89, 108-123).
template <int DR>
__device__ CommitOut table_commit(const Ctx &c, int lane, int ro, uint32_t kw, unsigned long long h, uint32_t s0, uint32_t s1,
                                  int set, uint32_t tag, const uint32_t (&vals)[DR], int stat_slot) {
    const CtlLayout L(c.world);
    uint32_t *misc = c.ctl + L.misc0;
    CommitOut out;
    out.idx = 0;
    out.is_new = false;
    out.ok = false;
    out.set = set;
    const uint32_t htag = (uint32_t)(h >> 32) | 0x80000000u;
    uint32_t pos = (uint32_t)h & c.slot_mask;
    uint32_t idx = 0;
    bool is_new = false;
    // the edge slot is needed whatever the lookup finds: request it now, use it after the probe
    uint32_t e = 0;
    if (lane == 0) e = atomicAdd(&c.ctl[L.edge0 + ro * CST], 1u);
    for (unsigned probes = 0;; probes++) {
        unsigned long long sv = 0;
        bool claimed = false;
        if (lane == 0) {
            sv = __hip_atomic_load(&c.slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (sv == 0) {
                unsigned long long want = ((unsigned long long)htag << 32) | kPending;
                unsigned long long old = atomicCAS(&c.slots[pos], 0ull, want);
                claimed = old == 0;
                sv = old;
            }
        }
        uint32_t lo = rflu((uint32_t)sv), hi = rflu((uint32_t)(sv >> 32));
        if (__ballot(claimed)) {
            // claimed: allocate the state, publish its key, then publish the index
            uint32_t ni = 0;
            if (lane == 0) ni = atomicAdd(&misc[MISC_NSTATES * CST], 1u);
            ni = rflu(ni);
            if (ni >= c.state_cap) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_STATE_OVERFLOW);
                return out;
            }
            if (lane < c.KL) __hip_atomic_store(&c.state_keys[(size_t)ni * c.KL + lane], kw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0)
                __hip_atomic_store(&c.slots[pos], ((unsigned long long)htag << 32) | ni, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            idx = ni;
            is_new = true;
            break;
        }
        if (hi == htag) {
            unsigned spins = 0;
            while (lo == kPending) {  // another wavefront is publishing this slot
                __builtin_amdgcn_s_sleep(2);
                unsigned long long t = 0;
                if (lane == 0) t = __hip_atomic_load(&c.slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lo = rflu((uint32_t)t);
                if (++spins > (1u << 22)) {
                    if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_TABLE_SPIN);
                    return out;
                }
            }
            // no acquire fence: every access to a key word is an agent-scope (sc1, L1-bypassing)
            // atomic, the publisher drained its stores before the index became visible
            uint32_t other = 0;
            if (lane < c.KL) other = __hip_atomic_load(&c.state_keys[(size_t)lo * c.KL + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!__ballot(lane < c.KL && other != kw)) {
                idx = lo;
                break;
            }
        }
        pos = (pos + 1) & c.slot_mask;
        if (probes > c.slot_mask) {
            if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_STATE_OVERFLOW);
            return out;
        }
    }
    // edge record: src (global id), dst (local index), label = time-0 value of every variable
    e = rflu(e);
    if (e >= c.edge_cap) {
        if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_EDGE_OVERFLOW);
        return out;
    }
    uint32_t *er = c.edges + ((size_t)ro * c.edge_cap + e) * c.ES;
    if (lane < 4) er[lane] = lane == 0 ? s0 : (lane == 1 ? s1 : (lane == 2 ? idx : 0u));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int k = q * 64 + lane;
        if (k < c.N) er[4 + k] = vals[q];
    }
    out.idx = idx;
    out.is_new = is_new;
    out.ok = true;
    if (is_new) {
        if (set < 0) {  // sharded: the record names the set by tag
            for (int t = 0; t < c.nsets && set < 0; t++)
                if ((uint32_t)kload(c.img, c.o.sets + t * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4)) == tag) set = t;
            if (set < 0) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_UNKNOWN_SET);
                out.ok = false;
            }
        }
        out.set = set;
        if (lane == 0) add_stats(c, stat_slot, ST_NEWSTATES, 1);
    }
    return out;
}

// new state: open its first search node in the round's output segment
template <int DR>
__device__ void emit_state_node(const Ctx &c, int lane, int ro, uint32_t *out_base, uint32_t out_cap, int parity,
                                const CommitOut &co, uint32_t expire, const uint32_t (&blk)[DR]) {
    const CtlLayout L(c.world);
    uint32_t np = 0;
    if (lane == 0) np = atomicAdd(&c.ctl[L.out(parity, ro)], 1u);
    np = rflu(np);
    if (np + 1 > out_cap) {
        if (lane == 0) atomicMax(&c.ctl[L.misc0 + MISC_ERROR * CST], (uint32_t)ERR_OUT_OVERFLOW);
        return;
    }
    uint32_t *dst = out_base + ((size_t)ro * out_cap + np) * c.NS;
    const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | co.idx;
    if (lane < 4) dst[lane] = lane == 0 ? (uint32_t)gid : (lane == 1 ? (uint32_t)(gid >> 32) : (lane == 2 ? (uint32_t)co.set : expire));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int k = q * 64 + lane;
        if (k < c.NK) dst[4 + k] = blk[q];
    }
}

// ------------------------------------------------------------------ k_commit (sharded runs)
template <int DR>
__global__ __launch_bounds__(256) void k_commit(Ctx c, CommitArgs a) {
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
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
    CommitOut co = table_commit<DR>(c, lane, ro, kw, h, s0, s1, -1, tag, vals, (int)(gw & 0x7fffffff));
    if (co.ok && co.is_new) emit_state_node<DR>(c, lane, ro, c.arena + p->out_base, p->out_cap, p->parity, co, expire, blk);
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

// re-insert every state into a larger table
__global__ void k_rehash(Ctx c, uint32_t n_states) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_states) return;
    const unsigned long long h = key_hash(c.state_keys + (size_t)i * c.KL, c.KL);
    const uint32_t htag = (uint32_t)(h >> 32) | 0x80000000u;
    uint32_t pos = (uint32_t)h & c.slot_mask;
    const unsigned long long want = ((unsigned long long)htag << 32) | i;
    while (atomicCAS(&c.slots[pos], 0ull, want) != 0ull) pos = (pos + 1) & c.slot_mask;
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
__global__ __launch_bounds__(256) void k_post_compact(EdgeView v, const uint8_t *alive, uint32_t *counter, long long *osrc,
                                                       long long *odst, int32_t *oval) {
    __shared__ uint32_t wcount[4], base;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const bool live = e < v.pref[R] && alive[e];
    const unsigned long long m = __ballot(live);
    if (lane == 0) wcount[wib] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) base = atomicAdd(counter, wcount[0] + wcount[1] + wcount[2] + wcount[3]);
    __syncthreads();
    if (!live) return;
    uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wib; w++) pos += wcount[w];
    const uint32_t *er = edge_at(v, e);
    osrc[pos] = (long long)(((unsigned long long)er[1] << 32) | er[0]);
    odst[pos] = (long long)er[2];
    for (int k = 0; k < v.N; k++) oval[(size_t)pos * v.N + k] = (int32_t)er[4 + k];
}

}  // namespace dev
}  // namespace stcsp
