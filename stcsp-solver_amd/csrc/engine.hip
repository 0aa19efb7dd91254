// engine.hip -- the MI355X (gfx950) stream-CSP propagation + search engine.
//
// Replaces the reference's recursive DFS (src/solveralgorithm.cpp:733-942 solverSolveRe) and its
// arc-queue propagator (:435-706) by a frontier search resident in HBM:
//
//   * every open search-tree node is one immutable record: a 4-word header + the packed domain
//     bitsets of all N variables at all K look-ahead points (device_types.hpp). There is no
//     trail (src/util.cpp:94-146): children are new records.
//   * k_expand: ONE 64-lane wavefront per open node. The domain block lives in VGPRs
//     (lane-striped, read with v_readlane / ds_bpermute, no LDS round trip), propagation runs
//     to the fixpoint of generalised arc consistency over the node's constraint set:
//       - X == next Y arcs (enforceNextConsistency, :544-593): shifted word AND;
//       - point constraints (enforcePointConsistency / findSupport / validate, :428-539):
//         lanes enumerate tuples of the scope variables' current domains, evaluate the
//         constraint's postfix program (the role of solverValidateRe, :336-424), and one
//         __ballot per 64 tuples feeds per-(variable,value) support masks;
//       - until constraints (enforceUntilConsistency, :598-614): check only.
//     Then the node is classified like solverSolveRe does: failed / branch (bisect the first
//     unbound variable: variableSplitLower/Upper, src/variable.cpp:52-67) / leaf.
//     A leaf emits a successor CANDIDATE: next constraint-set (per-leaf translation,
//     :755-805, looked up in a device transition table the host fills on demand), signature
//     (:812-837), edge label, time-advanced block (variableAdvanceOneTimeStep,
//     src/variable.cpp:94-108).
//   * k_commit: one wavefront per candidate does lookup-or-insert of (set, signature) in an
//     open-addressing hash table in HBM (the role of VertexTable, src/graph.h:64;
//     vertexTableGetVertex/AddVertex, src/graph.cpp:108-123) with a 64-bit CAS claim,
//     appends the edge record (edgeNew/vertexAddEdge, src/graph.cpp:33-38,78-89) and, for a
//     new state, opens its first search node.
//   * ok/fail bookkeeping (:857-874, 904-909) is the order-independent fixpoint of okfix.hpp,
//     run on the exported edge log.
//
// Integer bit work: no MFMA anywhere. Bounded by wave-level ALU/latency, then HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cset.hpp"
#include "device_types.hpp"
#include "okfix.hpp"
#include "stcsp_engine.h"

using namespace stcsp;

namespace {

constexpr int R = kRegions;
constexpr int CST = kCursorStride;
constexpr int kStatSlots = 64;
constexpr int kStatWords = 24;
enum { ST_NODES = 0, ST_FAILS, ST_LEAVES, ST_REVS, ST_EVALS, ST_REQUEUE, ST_NEWSTATES, ST_WAVEREVS, ST_SWEEPS, ST_SKIPPED,
       ST_CYC_LOAD, ST_CYC_SWEEP, ST_CYC_WAVE, ST_CYC_CLASSIFY, ST_CYC_COMMIT, ST_CYC_TOTAL,
       ST_QPUSH, ST_QPOP, ST_POLLS, ST_IDLE_CYC, ST_BUSY_CYC, ST_PSTACK_POP, ST_WAVES_WORKED };
constexpr int kMissStride = 66;  // set, nfirst, 64 values
constexpr uint32_t kPending = 0xffffffffu;

// control block (u32 words; every cursor on its own 64-byte line)
struct CtlLayout {
    int out0, cand0, edge0, misc0, words;  // out0: TWO sets of R cursors (round parity)
    __host__ __device__ CtlLayout(int world) {
        out0 = 0;
        cand0 = 2 * R * CST;
        edge0 = cand0 + world * R * CST;
        misc0 = edge0 + R * CST;
        words = misc0 + 8 * CST;
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
    int status, parity, nslots, sp;
    int take[R], count[R];
    unsigned long long in_base, out_base, arena_top, arena_words, slot_cap;
    unsigned in_cap, out_cap, edge_cap, state_cap, cand_cap, done_blocks;
    int chunk_r, world;
    long long rounds, open_total;
    DevSegment stack[kMaxSegments];
};
enum { MISC_NSTATES = 0, MISC_NMISS = 1, MISC_ERROR = 2 };
enum { ERR_WATCHDOG = 1, ERR_TABLE_SPIN = 2, ERR_EDGE_OVERFLOW = 3, ERR_STATE_OVERFLOW = 4, ERR_UNKNOWN_SET = 5,
       ERR_EMPTY_DOMAIN = 6, ERR_OUT_OVERFLOW = 7, ERR_CAND_OVERFLOW = 8 };

struct ImgOff {
    int sets, cons, scope, strides, items, itemrows, tables, var_lb, var_init, sig_vars, until_y, firstvars, trans, transvals,
        arr_off, words;
};

struct Ctx {
    int N, K, NK, NS, CS, ES, KL, sig_len, n_sig, n_until_cons, world, rank, nsets, stack_slots;
    // the compiled program: one contiguous image of 32-bit words (sections at the offsets in `o`),
    // staged into LDS by every workgroup of k_expand when it fits; the bytecode and the
    // user arrays (potentially large) stay in global memory
    const uint32_t *img;
    ImgOff o;
    const int *code;
    const int *arr_data;
    unsigned long long *slots;
    uint32_t slot_mask;
    uint32_t *state_keys;
    uint32_t state_cap;
    uint32_t *ctl;
    uint32_t *edges;
    uint32_t edge_cap;  // records per region
    int *miss;
    int miss_cap;
    unsigned long long *stats;
    Plan *plan;
    uint32_t *arena;
    uint32_t *cand;  // outbox [owner][region] x cand_cap records (sharded runs)
    // persistent mode (k_persist): shared ring of node records + per-wavefront private stacks
    uint32_t *pq;      // control words, one per 64-byte line: see PQ_*
    uint32_t *ring;    // qcap node records
    uint32_t *seq;     // qcap sequence numbers (bounded MPMC queue)
    uint32_t *pstack;  // [wavefront][pstk_cap] node records
    uint32_t *parked;  // nodes waiting for a constraint-set translation
    uint32_t qmask;    // qcap - 1
    int pstk_cap, park_cap, hungry;
};
enum { PQ_HEAD = 0, PQ_TAIL = 16, PQ_PENDING = 32, PQ_ABORT = 48, PQ_PARKED = 64, PQ_WORDS = 80 };
#ifndef STCSP_MAX_BACKOFF
#define STCSP_MAX_BACKOFF 128
#endif
constexpr unsigned kMaxBackoff = STCSP_MAX_BACKOFF;
enum { AB_NONE = 0, AB_QUEUE_FULL = 1, AB_PARK_FULL = 2, AB_SPIN = 3, AB_DEVICE_ERROR = 4 };

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

// ------------------------------------------------------------------ device helpers
// View of the program image: L = true -> the workgroup's LDS copy, false -> global memory.
// u(): wave-uniform read (scalar load / broadcast LDS read), v(): per-lane read.
template <bool L>
struct Img {
    const uint32_t *p;
    __device__ __forceinline__ int v(int off) const { return (int)p[off]; }
    __device__ __forceinline__ int u(int off) const;
};
// The compiled program (bytecode, descriptors, tables) is read-only for the lifetime of a launch
// and indexed wave-uniformly: reading it through the constant address space makes hipcc emit
// scalar loads (s_load_dword through the scalar cache) instead of 64-lane vector loads.
typedef const __attribute__((address_space(4))) int *kptr;
__device__ __forceinline__ int kload(const void *base, int idx) {
    return ((kptr)(const __attribute__((address_space(1))) int *)base)[idx];
}
template <>
__device__ __forceinline__ int Img<true>::u(int off) const { return __builtin_amdgcn_readfirstlane((int)p[off]); }
template <>
__device__ __forceinline__ int Img<false>::u(int off) const { return kload(p, off); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t rflu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// position of the k-th (0-based) set bit of m
__device__ __forceinline__ int select_kth(uint32_t m, int k) {
    for (int i = 0; i < k; i++) m &= m - 1;
    return __ffs((int)m) - 1;
}

// The node's domain block, lane-striped over DR VGPRs: word idx lives in r[idx >> 6], lane idx & 63.
template <int DR>
struct Dom {
    uint32_t r[DR];
    __device__ __forceinline__ uint32_t get(int idx) const {  // idx wave-uniform
        uint32_t v = r[0];
#pragma unroll
        for (int q = 1; q < DR; q++)
            if ((idx >> 6) == q) v = r[q];
        return rdlane(v, idx & 63);
    }
    __device__ __forceinline__ uint32_t gather(int idx) const {  // idx per lane
        uint32_t out = 0;
#pragma unroll
        for (int q = 0; q < DR; q++) {
            uint32_t t = (uint32_t)__shfl((int)r[q], idx & 63, 64);
            if ((idx >> 6) == q) out = t;
        }
        return out;
    }
    __device__ __forceinline__ void set(int idx, uint32_t val, int lane) {  // idx wave-uniform
#pragma unroll
        for (int q = 0; q < DR; q++)
            if ((idx >> 6) == q && lane == (idx & 63)) r[q] = val;
    }
};

#ifdef STCSP_PHASES
#define PHASE_NOW() __builtin_amdgcn_s_memtime()
#else
#define PHASE_NOW() 0ull
#endif
struct WaveStats {
    unsigned revs = 0, wave_revs = 0, sweeps = 0, skipped = 0;
    unsigned long long cyc_sweep = 0, cyc_wave = 0;
    unsigned long long evals = 0;
};

// Evaluate one constraint program on this lane's tuple (the role of solverValidateRe,
// reference src/solveralgorithm.cpp:336-424). varinfo/curval are per-lane registers indexed by
// scope position: varinfo = 1 + slot for lane-enumerated variables (value in lds_vals), 0 for
// wave-uniform ones (value in curval).
template <bool L>
__device__ int eval_program(const Ctx &c, const Img<L> &P, int pc, bool uses_valid, int lane, uint32_t varinfo, int curval,
                            const int *lds_vals, int *lds_stk) {
    int t = 0, sp = 0;
    bool valid = true;
    uint32_t dead = 0;
    for (;;) {
        int w = kload(c.code, pc++);
        int op = w & 255, arg = w >> 8;
        switch (op) {
            case OP_END: return t;
            case OP_CONST:
                lds_stk[sp * 64 + lane] = t;
                sp++;
                t = kload(c.code, pc++);
                break;
            case OP_VAR: {
                lds_stk[sp * 64 + lane] = t;
                sp++;
                uint32_t info = rdlane(varinfo, arg);
                if (info)
                    t = lds_vals[(info - 1) * 64 + lane];
                else
                    t = (int)rdlane((uint32_t)curval, arg);
                break;
            }
            case OP_ARR: {
                int off = P.u(c.o.arr_off + arg), size = P.u(c.o.arr_off + arg + 1) - off;
                bool inr = (unsigned)t < (unsigned)size;
                if (!inr && dead == 0) valid = false;
                t = inr ? c.arr_data[off + t] : 0;
                break;
            }
            case OP_ABS: t = t < 0 ? (int)(0u - (unsigned)t) : t; break;
            case OP_NOT: t = (t == 0); break;
            case OP_MASK_T:
            case OP_MASK_F: {
                int v = arg == 0 ? t : lds_stk[(sp - arg) * 64 + lane];
                bool live = (op == OP_MASK_T) ? (v != 0) : (v == 0);
                dead = (dead << 1) | (live ? 0u : 1u);
                break;
            }
            case OP_MASK_POP: dead >>= 1; break;
            case OP_SEL_IF: {
                int b = t, a = lds_stk[(sp - 1) * 64 + lane], cnd = lds_stk[(sp - 2) * 64 + lane];
                sp -= 2;
                t = cnd ? a : b;
                break;
            }
            case OP_SEL_AND: {
                int a = lds_stk[--sp * 64 + lane];
                t = a ? t : 0;
                break;
            }
            case OP_SEL_OR: {
                int a = lds_stk[--sp * 64 + lane];
                t = a ? 1 : t;
                break;
            }
            case OP_SEL_IMPLY: {
                int a = lds_stk[--sp * 64 + lane];
                t = (a == 0) ? 1 : (a <= t);
                break;
            }
            default: {
                int b = t, a = lds_stk[--sp * 64 + lane], r = 0;
                switch (op) {
                    case OP_ADD: r = (int)((unsigned)a + (unsigned)b); break;
                    case OP_SUB: r = (int)((unsigned)a - (unsigned)b); break;
                    case OP_MUL: r = (int)((unsigned)a * (unsigned)b); break;
                    case OP_DIV: r = (b == 0 || (a == INT_MIN && b == -1)) ? 0 : a / b; break;
                    case OP_MOD: r = (b == 0 || (a == INT_MIN && b == -1)) ? 0 : a % b; break;
                    case OP_LT: r = a < b; break;
                    case OP_GT: r = a > b; break;
                    case OP_LE: r = a <= b; break;
                    case OP_GE: r = a >= b; break;
                    case OP_EQ: r = a == b; break;
                    case OP_NE: r = a != b; break;
                    default: break;
                }
                t = (uses_valid && !valid) ? 0 : r;
                break;
            }
        }
    }
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// position of the k-th (0-based) set bit of m, branch-free (k < popcount(m))
__device__ __forceinline__ int select_kth_fast(uint32_t m, int k) {
    int pos = 0, cnt;
    cnt = __popc(m & 0xffffu);
    if (k >= cnt) { k -= cnt; pos += 16; m >>= 16; }
    cnt = __popc(m & 0xffu);
    if (k >= cnt) { k -= cnt; pos += 8; m >>= 8; }
    cnt = __popc(m & 0xfu);
    if (k >= cnt) { k -= cnt; pos += 4; m >>= 4; }
    cnt = __popc(m & 0x3u);
    if (k >= cnt) { k -= cnt; pos += 2; m >>= 2; }
    if (k >= (int)(m & 1u)) pos += 1;
    return pos;
}
// x / d for 0 <= x < 64, 1 <= d <= 64 via one reciprocal (exact: (x + 0.5) / d is never within
// 1/128 of an integer, far above float error)
__device__ __forceinline__ int small_div(int x, int d) {
    return (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}

// Enforce one point constraint at one time point: afterwards every remaining value of every
// scope variable has a supporting tuple (generalised arc consistency on this constraint; the
// reference tightens bounds only, solveralgorithm.cpp:476-523 -- this prunes at least as much).
// Returns false when a domain is wiped out. Rows of changed block words are OR-ed into `dirtyw`.
template <int DR, bool L>
__device__ bool revise_point(const Ctx &c, const Img<L> &G, const SetDesc &S, const ConDesc &C, int item, int p, Dom<DR> &dom,
                             int lane, uint32_t &dirtyw, int *lds_vals, int *lds_stk, int *ldom, WaveStats &ws, uint32_t *ctl_misc) {
    const int s = C.scope_len;
    // per-lane view of scope variable j = lane
    int var = 0;
    if (lane < s) var = G.v(c.o.scope + C.scope_off + lane);
    uint32_t D = dom.gather(p * c.N + var);
    if (lane >= s) D = 0;
    const int n = lane < s ? __popc(D) : 1;
    if (__ballot(lane < s && n == 0)) return false;
    const int vlb = lane < s ? G.v(c.o.var_lb + var) : 0;
    const bool use_bitmap = C.bitmap_off >= 0;
    const int mystride = (use_bitmap && lane < s) ? G.v(c.o.strides + C.stride_off + lane) : 0;

    // --- split the NON-SINGLETON scope variables: up to kMaxLowVars of them whose domain sizes
    // multiply to <= 64 are enumerated ACROSS LANES (lane index = mixed-radix tuple index), the rest
    // ("high") are stepped wave-uniformly by an odometer. Singletons are just constants.
    uint32_t varinfo = 0;   // scope lane j: 1 + slot if low
    int pairbase = 0;       // scope lane j: first pair lane of low var j
    int pst = 1, pn = 1, pk = 0;  // pair lane: stride / radix / digit it stands for
    int lane_part = 0;      // tuple lane: bitmap index contribution of the low variables
    int P = 1, nlow = 0, npairs = 0;
    unsigned long long highmask = 0, lowmask = 0;
    int maxn = 1;
    for (unsigned long long m = __ballot(lane < s && n > 1); m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const int nj = (int)rdlane((uint32_t)n, j);
        if (nlow < kMaxLowVars && P * nj <= 64) {
            const uint32_t Dj = rdlane(D, j);
            const int q = small_div(lane, P);
            const int digit = q - nj * small_div(q, nj);
            const int bitpos = select_kth_fast(Dj, digit);
            lds_vals[nlow * 64 + lane] = (int)rdlane((uint32_t)vlb, j) + bitpos;
            if (use_bitmap) lane_part += bitpos * (int)rdlane((uint32_t)mystride, j);
            if (lane == j) {
                varinfo = 1 + nlow;
                pairbase = npairs;
            }
            if (lane >= npairs && lane < npairs + nj) {
                pst = P;
                pn = nj;
                pk = lane - npairs;
            }
            lowmask |= 1ull << j;
            nlow++;
            P *= nj;
            npairs += nj;
        } else {
            highmask |= 1ull << j;
            if (nj > maxn) maxn = nj;
        }
    }
    const bool active = lane < P;
    const bool pairlane = lane < npairs;
    // pair lane (q,k): the set of tuple lanes whose digit of low variable q equals k is periodic
    // in the lane index -- build it arithmetically (no ballots)
    unsigned long long M = 0;
    if (pairlane) {
        M = ((1ull << pst) - 1ull) << (pk * pst);
        int sh = pst * pn;
#pragma unroll
        for (int it = 0; it < 6; it++) {
            if (sh < 64) M |= M << sh;
            sh <<= 1;
        }
        if (P < 64) M &= (1ull << P) - 1ull;
    }
    // Budget. Pruning a value needs the WHOLE product of the other variables refuted; when the
    // wave-uniform part of that product (the odometer range) is larger than the budget the
    // revision could never finish, so it is skipped outright. This keeps propagation sound (no
    // value is ever removed without proof) and the search complete: at a leaf every variable is a
    // singleton, the product is 1 and the constraint is checked exactly -- the same argument that
    // makes the reference's weaker, bounds-only propagation (solveralgorithm.cpp:476-523) yield
    // the same automaton.
    {
        const unsigned long long budget = use_bitmap ? kBudgetBitmapIters : kBudgetCodeIters;
        unsigned long long total_hi = 1;
        for (unsigned long long hm = highmask; hm && total_hi <= budget; hm &= hm - 1)
            total_hi *= (unsigned long long)rdlane((uint32_t)n, __ffsll((long long)hm) - 1);
        if (total_hi > budget) {
            if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
            ws.skipped++;
            return true;
        }
    }
    bool hit = false;   // pair lanes: this (low var, digit) has a support
    uint32_t hs = 0;    // scope lanes (high vars): supported value bits
    int digit_h = 0;    // scope lanes (high vars): odometer digit
    int curbit = lane < s ? (__ffs((int)D) - 1) : 0;
    int curval = vlb + curbit;
    const bool is_high = (highmask >> lane) & 1ull;
    // bitmap index contribution of the singleton variables (constant for this revision)
    int base_sum = 0;
    if (use_bitmap) base_sum = wave_sum((lane < s && n == 1) ? curbit * mystride : 0);
    ws.revs++;
    ws.wave_revs++;
    const unsigned nact = (unsigned)P;
    // stage A: maxn "diagonal" probes (high variable j takes its (it mod n_j)-th value): every
    // value of every high variable appears once, so loose constraints finish here.
    // stage B: exhaustive odometer over the high variables, early exit once all is supported.
    bool any_sat = false;
    unsigned long long iters = 0;
    int stage_a_left = highmask ? maxn : 1;
    bool stage_b = false;
    for (;;) {
        if (stage_a_left > 0) {
            if (is_high) {
                int it = maxn - stage_a_left;
                curbit = select_kth_fast(D, it - n * small_div(it, n));
                curval = vlb + curbit;
            }
            stage_a_left--;
        } else if (!stage_b) {
            stage_b = true;  // first exhaustive tuple block: all high digits 0
            if (is_high) {
                digit_h = 0;
                curbit = __ffs((int)D) - 1;
                curval = vlb + curbit;
            }
        }
        int res;
        if (use_bitmap) {
            int bit = lane_part + base_sum;
            for (unsigned long long hm = highmask; hm; hm &= hm - 1) {
                const int j = __ffsll((long long)hm) - 1;
                bit += (int)rdlane((uint32_t)curbit, j) * (int)rdlane((uint32_t)mystride, j);
            }
            res = active ? (int)(((uint32_t)G.v(c.o.tables + C.bitmap_off + (bit >> 5)) >> (bit & 31)) & 1u) : 0;
        } else {
            res = eval_program<L>(c, G, C.code_off, C.uses_valid != 0, lane, varinfo, curval, lds_vals, lds_stk);
        }
        ws.evals += nact;
        const unsigned long long sm = __ballot(active && res != 0);
        if (sm) {
            any_sat = true;
            if (pairlane && (M & sm)) hit = true;
            if (is_high) hs |= 1u << curbit;
        }
        if (__ballot((pairlane && !hit) || (is_high && hs != D)) == 0) break;  // everything supported
        if (stage_a_left > 0) continue;
        if (!highmask) break;  // no high variables: the lanes covered the whole product
        if (!stage_b) continue;
        // advance the odometer (wave-uniform carry chain over the high variables)
        bool carry = true;
        for (unsigned long long hm = highmask; hm && carry; hm &= hm - 1) {
            const int j = __ffsll((long long)hm) - 1;
            int dj = (int)rdlane((uint32_t)digit_h, j) + 1;
            const int nj = (int)rdlane((uint32_t)n, j);
            if (dj == nj)
                dj = 0;
            else
                carry = false;
            if (lane == j) {
                digit_h = dj;
                curbit = select_kth_fast(D, dj);
                curval = vlb + curbit;
            }
        }
        if (carry) break;  // wrapped around: product exhausted
        if (++iters > (1ull << 22)) {
            if (lane == 0) atomicMax(&ctl_misc[MISC_ERROR * CST], (uint32_t)ERR_WATCHDOG);
            return false;
        }
    }
    // --- write back. No satisfying tuple at all: wipe-out. Otherwise singletons are supported by
    // construction and only the enumerated / stepped variables can lose values.
    if (!any_sat) return false;
    const unsigned long long hitmask = __ballot(pairlane && hit);
    for (unsigned long long m = lowmask | highmask; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const uint32_t Dj = rdlane(D, j);
        uint32_t newD;
        if ((lowmask >> j) & 1ull) {
            const uint32_t dig = (uint32_t)(hitmask >> (int)rdlane((uint32_t)pairbase, j));
            bool keep = false;
            if (lane < 32 && ((Dj >> lane) & 1u)) keep = (dig >> __popc(Dj & ((1u << lane) - 1u))) & 1u;
            newD = (uint32_t)__ballot(keep);
        } else {
            newD = rdlane(hs, j);
        }
        if (newD == 0) return false;
        if (newD != Dj) {
            const int vj = (int)rdlane((uint32_t)var, j);
            dom.set(p * c.N + vj, newD, lane);
            if (lane == 0) ldom[p * c.N + vj] = (int)newD;  // keep the sweep's LDS copy of the block current
            if (lane < S.iw) dirtyw |= (uint32_t)G.v(c.o.itemrows + S.itemrows_off + (p * c.N + vj) * S.iw + lane);
        }
    }
    // one revision is a fixpoint for this constraint at this point: no need to revisit it for
    // its own changes (supports are whole tuples of surviving values)
    if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
    return true;
}

template <bool L>
__device__ __forceinline__ void load_set(const Ctx &c, const Img<L> &P, int set, SetDesc &S) {
    int *dst = (int *)&S;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(SetDesc) / 4); i++) dst[i] = P.u(c.o.sets + set * (int)(sizeof(SetDesc) / 4) + i);
}
template <bool L>
__device__ __forceinline__ void load_con(const Ctx &c, const Img<L> &P, int idx, ConDesc &C) {
    int *dst = (int *)&C;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(ConDesc) / 4); i++) dst[i] = P.u(c.o.cons + idx * (int)(sizeof(ConDesc) / 4) + i);
}

__device__ __forceinline__ void add_stats(const Ctx &c, int gw, int which, unsigned long long v) {
    if (v) atomicAdd(&c.stats[(gw % kStatSlots) * kStatWords + which], v);
}

template <int DR>
__device__ __forceinline__ void store_node(uint32_t *dst, const Ctx &c, uint32_t h0, uint32_t h1, uint32_t h2, uint32_t h3,
                                           const Dom<DR> &dom, int lane) {
    if (lane < 4) dst[lane] = lane == 0 ? h0 : (lane == 1 ? h1 : (lane == 2 ? h2 : h3));
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        if (idx < c.NK) dst[4 + idx] = dom.r[q];
    }
}

enum Outcome : int { OC_FAIL = 0, OC_BRANCH, OC_MISS, OC_LEAF };
struct NodeHdr {
    uint32_t h0, h1;  // src state (global id)
    int set;          // constraint set index
    uint32_t seed;    // dirty seed (see k_expand)
    uint32_t expire;  // until-expire bits
};
struct BranchOut {
    int bvar;
    uint32_t D, lowmask;  // children: D & lowmask, D & ~lowmask at word (0, bvar)
};
template <int DR>
struct LeafOut {
    uint32_t kw;            // lane j: key word j = [next set tag, signature...]
    unsigned long long h;   // key hash
    int next_set, owner;
    uint32_t next_tag, new_expire;
    uint32_t evals[DR];     // edge label (Edge::values), lane-striped
    uint32_t nblk[DR];      // time-advanced block, lane-striped
};
struct CommitOut {
    uint32_t idx;  // local state index
    bool is_new, ok;
    int set;
};
template <int DR>
__device__ CommitOut table_commit(const Ctx &c, int lane, int ro, uint32_t kw, unsigned long long h, uint32_t s0, uint32_t s1,
                                  int set, uint32_t tag, const uint32_t (&vals)[DR], int stat_slot);
template <int DR>
__device__ void emit_state_node(const Ctx &c, int lane, int ro, uint32_t *out_base, uint32_t out_cap, int parity,
                                const CommitOut &co, uint32_t expire, const uint32_t (&blk)[DR]);

// ------------------------------------------------------------------ one search node
// Propagate the block in `dom` to its fixpoint under the node's constraint set and classify the
// node like solverSolveRe does: failed / branch / leaf (or "miss": a leaf whose constraint-set
// translation the host has not provided yet). Outputs stay in registers; the callers (the
// round-based k_expand and the persistent k_persist) decide where children and leaves go.
template <int DR, bool L>
__device__ int process_node(const Ctx &c, const Img<L> &P, int lane, int *lds_vals, int *lds_stk, Dom<DR> &dom,
                            const NodeHdr &hd, int gw, BranchOut &bo, LeafOut<DR> &lo) {
    const CtlLayout L_(c.world);
    uint32_t *misc = c.ctl + L_.misc0;
    const unsigned long long t_start = PHASE_NOW();
    (void)t_start;
    const int set = hd.set;
    const uint32_t seed = hd.seed, expire = hd.expire;
    SetDesc S;
    load_set<L>(c, P, set, S);

    // ---- propagate to the GAC fixpoint (role of generalisedArcConsistent, :617-706). Work items
    // are (constraint, time point) pairs; the dirty mask is lane-striped (lane w holds word w).
    // Items [0, nsmall) -- X == next Y arcs, until checks and small extensional point constraints --
    // are revised ONE ITEM PER LANE against a snapshot of the block, their prunings ANDed together
    // through an LDS copy (a Jacobi sweep); the remaining items are revised by the whole
    // wavefront one at a time. Monotone propagators: any fair order reaches the same fixpoint.
    WaveStats ws;
    const unsigned long long t_loaded = PHASE_NOW();
    (void)t_loaded;
    int *ldom = lds_stk + c.stack_slots * 64;  // NK-word AND-accumulator of this wavefront
    uint32_t dirtyw = 0;
    if (lane < S.iw) {
        if (seed == 0) {
            int left = S.nitems - lane * 32;
            dirtyw = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
        } else if (seed != 0xffffu) {
            dirtyw = (uint32_t)P.v(c.o.itemrows + S.itemrows_off + (int)(seed - 1) * S.iw + lane);  // word (0, seed var)
        }
    }
    uint32_t smallmask = 0;
    if (lane < S.iw) {
        int left = S.nsmall - lane * 32;
        smallmask = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
    }
    bool consistent = true;
    unsigned guard = 0;
    // LDS copy of the block (AND-accumulator of the sweeps); kept equal to `dom` between sweeps.
    // Only this wavefront touches it and a wavefront's LDS operations execute in order, so
    // wavefront-scope fences (compiler ordering only) are enough.
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        if (idx < c.NK) ldom[idx] = (int)dom.r[q];
    }
    while (consistent) {
        if (__ballot((dirtyw & smallmask) != 0)) {
            const unsigned long long t_sw = PHASE_NOW();
            // ---- lane-parallel sweep over the dirty small items
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool lfail = false;
            ws.sweeps++;
            const int npass = (S.nsmall + 63) >> 6;
            for (int t = 0; t < npass; t++) {
                const int item = t * 64 + lane;
                const uint32_t dw0 = rdlane(dirtyw, (2 * t) & 63), dw1 = rdlane(dirtyw, (2 * t + 1) & 63);
                const uint32_t dw = lane < 32 ? dw0 : dw1;
                const bool isd = item < S.nsmall && ((dw >> (item & 31)) & 1u);
                unsigned long long dmask = __ballot(isd);
                if (!dmask) continue;
                ws.revs += (unsigned)__popcll(dmask);
                ItemDesc it;
                {
                    const int ioff = c.o.items + (S.item_begin + (isd ? item : 0)) * (int)(sizeof(ItemDesc) / 4);
                    int *dst = (int *)&it;
#pragma unroll
                    for (int k = 0; k < (int)(sizeof(ItemDesc) / 4); k++) dst[k] = P.v(ioff + k);
                }
                // gathers are executed by every lane (cross-lane reads need the source lanes active)
                uint32_t D0 = dom.gather(it.idx[0]), D1 = dom.gather(it.idx[1]);
                uint32_t D2 = dom.gather(it.idx[2]), D3 = dom.gather(it.idx[3]);
                if (isd) {
                    if (it.type == IT_NEXT) {
                        // X == next Y <=> X[p] == Y[p+1]; bit i of X is value lbX + i = bit i + sh of Y
                        const int sh = it.aux;
                        uint32_t Yal = sh >= 0 ? (sh < 32 ? D1 >> sh : 0u) : (-sh < 32 ? D1 << -sh : 0u);
                        uint32_t m = D0 & Yal;
                        uint32_t newY = sh >= 0 ? (sh < 32 ? m << sh : 0u) : (-sh < 32 ? m >> -sh : 0u);
                        if (m == 0) lfail = true;
                        if (m != D0) atomicAnd((unsigned *)&ldom[it.idx[0]], m);
                        if (newY != D1) atomicAnd((unsigned *)&ldom[it.idx[1]], newY);
                    } else if (it.type == IT_UNTIL) {
                        if (!((expire >> it.aux) & 1u) && __popc(D0) == 1 && __popc(D1) == 1) {
                            int vx = P.v(c.o.var_lb + it.idx[0]) + __ffs((int)D0) - 1, vy = P.v(c.o.var_lb + it.idx[1]) + __ffs((int)D1) - 1;
                            if (vx != 1 && vy != 1) lfail = true;
                        }
                    } else {
                        // small extensional constraint: one row of allowed word-variable values per
                        // tuple of the other (<= 3) variables; scan the rows of the current product
                        if (it.arity < 2) D1 = 1u;
                        if (it.arity < 3) D2 = 1u;
                        if (it.arity < 4) D3 = 1u;
                        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                        const int tab = c.o.tables + it.toff;
                        unsigned nev = 0;
                        for (uint32_t m3 = D3; m3; m3 &= m3 - 1) {
                            const int b3 = __ffs((int)m3) - 1;
                            for (uint32_t m2 = D2; m2; m2 &= m2 - 1) {
                                const int b2 = __ffs((int)m2) - 1;
                                const int base = it.r1 * (b2 + it.r2 * b3);
                                for (uint32_t m1 = D1; m1; m1 &= m1 - 1) {
                                    const int b1 = __ffs((int)m1) - 1;
                                    const uint32_t row = (uint32_t)P.v(tab + base + b1) & D0;
                                    nev++;
                                    if (row) {
                                        s0 |= row;
                                        s1 |= 1u << b1;
                                        s2 |= 1u << b2;
                                        s3 |= 1u << b3;
                                    }
                                }
                            }
                        }
                        ws.evals += nev;
                        if (s0 == 0) lfail = true;
                        if (s0 != D0) atomicAnd((unsigned *)&ldom[it.idx[0]], s0);
                        if (it.arity > 1 && s1 != D1) atomicAnd((unsigned *)&ldom[it.idx[1]], s1);
                        if (it.arity > 2 && s2 != D2) atomicAnd((unsigned *)&ldom[it.idx[2]], s2);
                        if (it.arity > 3 && s3 != D3) atomicAnd((unsigned *)&ldom[it.idx[3]], s3);
                    }
                }
            }
            dirtyw &= ~smallmask;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (__ballot(lfail)) {
                consistent = false;
                break;
            }
            // read the intersection back; every changed word re-dirties the items that read it
#pragma unroll
            for (int q = 0; q < DR; q++) {
                int idx = q * 64 + lane;
                uint32_t nd = idx < c.NK ? (uint32_t)ldom[idx] : dom.r[q];
                if (__ballot(idx < c.NK && nd == 0)) consistent = false;
                unsigned long long cm = __ballot(nd != dom.r[q]);
                dom.r[q] = nd;
                while (cm) {
                    int l = __ffsll((long long)cm) - 1;
                    cm &= cm - 1;
                    if (lane < S.iw) dirtyw |= (uint32_t)P.v(c.o.itemrows + S.itemrows_off + (q * 64 + l) * S.iw + lane);
                }
            }
            if (++guard > (1u << 20)) {
                if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_WATCHDOG);
                consistent = false;
            }
            ws.cyc_sweep += PHASE_NOW() - t_sw;
            continue;
        }
        const unsigned long long t_wv = PHASE_NOW();
        unsigned long long dm = __ballot(dirtyw != 0);
        if (!dm) break;
        int wl = __ffsll((long long)dm) - 1;
        uint32_t word = rdlane(dirtyw, wl);
        int b = __ffs((int)word) - 1;
        int item = wl * 32 + b;
        if (lane == wl) dirtyw &= ~(1u << b);
        const int ibase = c.o.items + (S.item_begin + item) * (int)(sizeof(ItemDesc) / 4);
        const int ipoint = P.u(ibase + 1), icon = P.u(ibase + 2);
        ConDesc C;
        load_con<L>(c, P, icon, C);
        consistent = revise_point<DR, L>(c, P, S, C, item, ipoint, dom, lane, dirtyw, lds_vals, lds_stk, ldom, ws, misc);
        ws.cyc_wave += PHASE_NOW() - t_wv;
        if (++guard > (1u << 20)) {
            if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_WATCHDOG);
            consistent = false;
        }
    }
    if (lane == 0) {
        add_stats(c, gw, ST_NODES, 1);
        add_stats(c, gw, ST_REVS, ws.revs);
        add_stats(c, gw, ST_EVALS, ws.evals);
        add_stats(c, gw, ST_WAVEREVS, ws.wave_revs);
        add_stats(c, gw, ST_SWEEPS, ws.sweeps);
        add_stats(c, gw, ST_SKIPPED, ws.skipped);
#ifdef STCSP_PHASES
        add_stats(c, gw, ST_CYC_LOAD, t_loaded - t_start);
        add_stats(c, gw, ST_CYC_SWEEP, ws.cyc_sweep);
        add_stats(c, gw, ST_CYC_WAVE, ws.cyc_wave);
#endif
    }
    const unsigned long long t_prop = PHASE_NOW();
    (void)t_prop;
    if (!consistent) {
        if (lane == 0) add_stats(c, gw, ST_FAILS, 1);
        return OC_FAIL;
    }

    // ---- classify (solverGetFirstUnboundVar, src/solver.cpp:41-53): first variable, in queue
    // order, whose time-0 domain is not a singleton
    int bvar = -1;
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        unsigned long long m = __ballot(idx < c.N && __popc(dom.r[q]) > 1);
        if (bvar < 0 && m) bvar = q * 64 + __ffsll((long long)m) - 1;
    }
    if (bvar >= 0) {
        // bisect [lb,ub] of the branching variable (variableSplitLower/Upper, variable.cpp:52-67)
        const uint32_t D = dom.get(bvar);
        const int lo_ = __ffs((int)D) - 1, hi_ = 31 - __clz((int)D);
        const int mid = lo_ + (hi_ - lo_) / 2;
        bo.bvar = bvar;
        bo.D = D;
        bo.lowmask = (mid >= 31) ? 0xffffffffu : ((2u << mid) - 1u);
        return OC_BRANCH;
    }

    // ---- leaf: every variable has a single time-0 value (solveralgorithm.cpp:739-910)
    // (1) next constraint set: per-leaf translation (:755-805) via the transition table
    int next_set = set;
    if (!S.self_loop) {
        int fv = 0;
        if (lane < S.nfirst) {
            int v = P.v(c.o.firstvars + S.first_off + lane);
            fv = v;
        }
        uint32_t fd = dom.gather(fv);  // time-0 word of that variable
        int fval = (lane < S.nfirst) ? P.v(c.o.var_lb + fv) + __ffs((int)fd) - 1 : 0;
        next_set = -1;
        for (int t = 0; t < S.trans_count && next_set < 0; t++) {
            int voff = P.u(c.o.trans + (S.trans_begin + t) * 2);
            bool ne = lane < S.nfirst && P.v(c.o.transvals + voff + lane) != fval;
            if (!__ballot(ne)) next_set = P.u(c.o.trans + (S.trans_begin + t) * 2 + 1);
        }
        if (next_set < 0) {
            // unknown transition: park the node again and tell the host which translation is needed
            uint32_t mi = 0;
            if (lane == 0) mi = atomicAdd(&misc[MISC_NMISS * CST], 1u);
            mi = rflu(mi);
            if ((int)mi < c.miss_cap) {
                int *rec = c.miss + (size_t)mi * kMissStride;
                if (lane == 0) {
                    rec[0] = set;
                    rec[1] = S.nfirst;
                }
                if (lane < S.nfirst) rec[2 + lane] = fval;
            }
            if (lane == 0) add_stats(c, gw, ST_REQUEUE, 1);
            return OC_MISS;
        }
    }
    const uint32_t next_tag = (uint32_t)P.u(c.o.sets + next_set * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4));
    // (2) signature (:812-837): signature variables in queue order, then one sticky flag per until
    uint32_t new_expire = expire;
    uint32_t kw = 0;  // lane j holds key word j: [tag, sig...]
    {
        int sv = 0;
        if (lane >= 1 && lane <= c.n_sig) sv = P.v(c.o.sig_vars + lane - 1);
        uint32_t sd = dom.gather(sv);
        if (lane >= 1 && lane <= c.n_sig) kw = (uint32_t)(P.v(c.o.var_lb + sv) + __ffs((int)sd) - 1);
        for (int u = 0; u < c.n_until_cons; u++) {
            int y = P.u(c.o.until_y + u);
            uint32_t DY = dom.get(y);
            bool ex = (expire >> u) & 1u;
            if (!ex && P.u(c.o.var_lb + y) + __ffs((int)DY) - 1 == 1) {
                ex = true;
                new_expire |= 1u << u;
            }
            if (lane == 1 + c.n_sig + u) kw = ex ? 1u : 0u;
        }
        if (lane == 0) kw = next_tag;
    }
    // (3) owner shard = hash(key) % world
    unsigned long long h = kHashSeed;
    for (int j = 0; j < c.KL; j++) h = mix64(h, rdlane(kw, j));
    h = mix_final(h);
    lo.kw = kw;
    lo.h = h;
    lo.next_set = next_set;
    lo.next_tag = next_tag;
    lo.new_expire = new_expire;
    lo.owner = (int)((h >> 40) % (unsigned)c.world);
    // edge label (Edge::values) and the time-advanced block (variableAdvanceOneTimeStep,
    // variable.cpp:94-108: point p <- point p+1, last point <- [lb,ub]), lane-striped
#pragma unroll
    for (int q = 0; q < DR; q++) {
        int idx = q * 64 + lane;
        lo.evals[q] = idx < c.N ? (uint32_t)(P.v(c.o.var_lb + idx) + __ffs((int)dom.r[q]) - 1) : 0u;
        uint32_t shifted = dom.gather(idx + c.N < c.NK ? idx + c.N : 0);
        uint32_t nb = 0;
        if (idx < c.NK) {
            int p = idx / c.N, v = idx - p * c.N;
            nb = (p + 1 < c.K) ? shifted : (uint32_t)P.v(c.o.var_init + v);
        }
        lo.nblk[q] = nb;
    }
    if (lane == 0) add_stats(c, gw, ST_LEAVES, 1);
    return OC_LEAF;
}

// ------------------------------------------------------------------ k_expand (round-based)
// expand ONE open node (slot `gw` of this round) with one wavefront
template <int DR, bool L>
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
    BranchOut bo;
    LeafOut<DR> lo;
    const int oc = process_node<DR, L>(c, P, lane, lds_vals, lds_stk, dom, hd, gw, bo, lo);
    if (oc == OC_FAIL) return;
    uint32_t *out_region = a.out_base + (size_t)ro * a.out_cap * c.NS;
    if (oc == OC_BRANCH) {
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(a.parity, ro)], 2u);
        pos = rflu(pos);
        if (pos + 2 > a.out_cap) {
            if (lane == 0) atomicMax(&misc[MISC_ERROR * CST], (uint32_t)ERR_OUT_OVERFLOW);
            return;
        }
        Dom<DR> child = dom;
        child.set(bo.bvar, bo.D & bo.lowmask, lane);
        const uint32_t cw2 = (uint32_t)hd.set | ((uint32_t)(bo.bvar + 1) << 16);
        store_node<DR>(out_region + (size_t)pos * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, child, lane);
        child.set(bo.bvar, bo.D & ~bo.lowmask, lane);
        store_node<DR>(out_region + (size_t)(pos + 1) * c.NS, c, hd.h0, hd.h1, cw2, hd.expire, child, lane);
        return;
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
    if (c.world == 1) {
        // unsharded: commit right here, the leaf's data never leaves the registers
        CommitOut co = table_commit<DR>(c, lane, ro, lo.kw, lo.h, hd.h0, hd.h1, lo.next_set, lo.next_tag, lo.evals, gw);
        if (co.ok && co.is_new) emit_state_node<DR>(c, lane, ro, a.out_base, a.out_cap, a.parity, co, lo.new_expire, lo.nblk);
        return;
    }
    // sharded: candidate record for the owner: header, signature, edge label, block
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
}

__device__ __forceinline__ uint32_t ald(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ long long wave_sum64(long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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
    const unsigned out_cap = 3u * (unsigned)maxtake;
    int status = PS_RUN;
    if (arena_top + (unsigned long long)R * out_cap * c.NS > p->arena_words) status = PS_NEED_ARENA;
    const unsigned max_edges = (unsigned)wave_max(rl ? (int)ald(&c.ctl[L.edge0 + lane * CST]) : 0);
    const unsigned long long ns = rflu(lane == 0 ? ald(&c.ctl[L.misc0 + MISC_NSTATES * CST]) : 0u);
    if (status == PS_RUN && (unsigned long long)max_edges + maxtake > p->edge_cap) status = PS_NEED_EDGES;
    if (status == PS_RUN && ns + taken > p->state_cap) status = PS_NEED_STATES;
    if (status == PS_RUN && (ns + taken) * 2 > p->slot_cap) status = PS_NEED_TABLE;
    if (status == PS_RUN && c.world > 1) {
        int mc = 0;
        for (int k = lane; k < c.world * R; k += 64) mc = max(mc, (int)ald(&c.ctl[L.cand0 + k * CST]));
        mc = wave_max(mc);
        if ((unsigned long long)mc + chunk > p->cand_cap) status = PS_OUTBOX_FULL;
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

__device__ void finalize_round(const Ctx &c, Plan *p, int lane) {
    push_output(c, p, true, lane);
    if (lane == 0) p->rounds++;
    __threadfence();
    if (rfl(p->status) == PS_RUN) plan_next(c, p, lane);
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
// Ctx is read through a pointer (scalar loads on demand): passing it by value kept ~130 SGPRs
// live/spilled and cost a wavefront of occupancy per SIMD.
template <int DR, bool L>
__global__ __launch_bounds__(256, STCSP_EXPAND_WAVES) void k_expand(const Ctx *__restrict__ cp) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    if (kload(c.plan, (int)(offsetof(Plan, status) / 4)) != PS_RUN) return;  // the burst ran past the end
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int n_slots = kload(c.plan, (int)(offsetof(Plan, nslots) / 4));
    // workgroups without a node slot leave at once; the ticket below counts the working ones only
    if ((int)blockIdx.x * 4 >= n_slots) return;
    const unsigned n_working = (unsigned)min((n_slots + 3) / 4, (int)gridDim.x);
    const int img_words = L ? ((c.o.words + 3) & ~3) : 0;
    if (L) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = (kMaxLowVars + c.stack_slots) * 64 + ((c.NK + 63) & ~63);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    Img<L> P{L ? (const uint32_t *)smem : c.img};
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
    for (int gw = blockIdx.x * 4 + wib; gw < n_slots; gw += total_waves) expand_node<DR, L>(c, a, P, gw, lane, lds_vals, lds_stk);
    __syncthreads();
    if (wib == 0) {
        unsigned t = 0;
        if (lane == 0) {
            __threadfence();
            t = atomicAdd(&c.plan->done_blocks, 1u);
        }
        if (rflu(t) == n_working - 1) {  // last working workgroup: every cursor of this round is final
            if (lane == 0) c.plan->done_blocks = 0;
            __threadfence();
            finalize_round(c, c.plan, lane);
        }
    }
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
    const int img_words = L ? ((c.o.words + 3) & ~3) : 0;
    if (L) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int per_wave = (kMaxLowVars + c.stack_slots) * 64 + ((c.NK + 63) & ~63);
    int *lds_vals = smem + img_words + wib * per_wave;
    int *lds_stk = lds_vals + kMaxLowVars * 64;
    Img<L> P{L ? (const uint32_t *)smem : c.img};
    const int wid = blockIdx.x * 4 + wib;
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
        const int oc = process_node<DR, L>(c, P, lane, lds_vals, lds_stk, dom, hd, wid + (int)nodes_done, bo, lo);
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
    unsigned long long h = kHashSeed;
    for (int j = 0; j < c.KL; j++) h = mix64(h, c.state_keys[(size_t)i * c.KL + j]);
    h = mix_final(h);
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

// ------------------------------------------------------------------ host side
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        return hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &v) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess) return e;
        if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
};

thread_local std::string g_create_error;

}  // namespace

struct stcsp_engine {
    SetManager mgr;
    FlatProgram prog;
    stcsp_options opt{};
    int device = 0;
    hipStream_t stream = nullptr;
    Ctx ctx{};
    int DR = 1;
    CtlLayout L{1};
    size_t lds_bytes = 0;
    int chunk_r = 0;  // max nodes taken per region per launch
    int max_blocks = 256 * 4;  // k_expand grid (workgroups): set from the occupancy query
    int persist_blocks = 256 * 4;  // k_persist grid: must all be resident

    DevBuf<int> d_arr_data, d_code, d_miss;
    DevBuf<uint32_t> d_state_keys, d_ctl, d_edges, d_arena, d_cand, d_pack, d_img;
    bool img_in_lds = false;
    DevBuf<unsigned long long> d_slots, d_stats;
    uint32_t *h_ctl = nullptr;  // pinned
    int *h_miss = nullptr;      // pinned
    uint32_t cand_cap = 0;
    DevBuf<Plan> d_plan;
    DevBuf<uint32_t> d_pq, d_ring, d_seq, d_pstack, d_parked;  // persistent mode
    bool persist = false;    // STCSP_PERSIST=1: unsharded solves without budgets use k_persist (experimental)
    DevBuf<Ctx> d_ctx;       // device copy of ctx for k_expand (re-uploaded before a burst)
    Ctx *h_ctx = nullptr;    // pinned staging copy
    Plan *h_plan = nullptr;  // pinned mirror of the plan header (everything before the stack)
    int burst = 8;           // rounds enqueued per host synchronisation
    std::vector<uint32_t> edge_count = std::vector<uint32_t>(R, 0);
    uint32_t n_states = 0;
    bool begun = false, finished = false;
    bool truncated = false;
    std::chrono::steady_clock::time_point t_begin;
    double seconds_search = 0, seconds_export = 0;
    long long levels = 0;
    std::string err;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    double seconds_expand_kernel = 0;
    long long expand_launches = 0;
    stcsp_counters snap{};
    std::vector<int32_t> blob;

    // export storage
    std::vector<int32_t> r_cid, r_sig, r_eval;
    std::vector<uint8_t> r_fail, r_issig;
    std::vector<int64_t> r_esrc, r_edst;
    // device-side export (unsharded): compacted SoA on the device, pinned mirrors on the host
    DevBuf<long long> d_osrc, d_odst;
    DevBuf<int32_t> d_oval;
    DevBuf<uint8_t> d_alive, d_fail;
    DevBuf<uint32_t> d_outdeg, d_post;  // d_post: [0] changed flag, [1] live-edge counter
    long long *h_osrc = nullptr, *h_odst = nullptr;
    int32_t *h_oval = nullptr;
    uint32_t *h_keys = nullptr;
    uint8_t *h_fail = nullptr;
    size_t h_edge_cap = 0, h_state_cap = 0;

    ~stcsp_engine() {
        for (auto &e : ev_pool) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        if (h_ctl) (void)hipHostFree(h_ctl);
        if (h_plan) (void)hipHostFree(h_plan);
        if (h_ctx) (void)hipHostFree(h_ctx);
        if (h_osrc) (void)hipHostFree(h_osrc);
        if (h_odst) (void)hipHostFree(h_odst);
        if (h_oval) (void)hipHostFree(h_oval);
        if (h_keys) (void)hipHostFree(h_keys);
        if (h_fail) (void)hipHostFree(h_fail);
        if (h_miss) (void)hipHostFree(h_miss);
        if (stream) (void)hipStreamDestroy(stream);
    }

    int fail(int code, const char *fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
#define HIPCHK(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) return fail(STCSP_E_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

    int upload_program() {
        int rc = mgr.compile(prog);
        if (rc != STCSP_OK) return fail(rc, "%s", mgr.error.c_str());
        // one contiguous image; every section starts on a 16-byte boundary
        std::vector<uint32_t> img;
        ImgOff o{};
        auto put = [&](const void *data, size_t bytes) {
            while (img.size() & 3) img.push_back(0u);
            int off = (int)img.size();
            size_t words = (bytes + 3) / 4;
            img.resize(img.size() + words, 0u);
            if (bytes) memcpy(img.data() + off, data, bytes);
            return off;
        };
        std::vector<uint32_t> init(ctx.N);
        for (int v = 0; v < ctx.N; v++) {
            int w = mgr.ub[v] - mgr.lb[v] + 1;
            init[v] = w >= 32 ? 0xffffffffu : ((1u << w) - 1u);
        }
        o.sets = put(prog.sets.data(), prog.sets.size() * sizeof(SetDesc));
        o.cons = put(prog.cons.data(), prog.cons.size() * sizeof(ConDesc));
        o.scope = put(prog.scope.data(), prog.scope.size() * 4);
        o.strides = put(prog.strides.data(), prog.strides.size() * 4);
        o.items = put(prog.items.data(), prog.items.size() * sizeof(ItemDesc));
        o.itemrows = put(prog.itemrows.data(), prog.itemrows.size() * 4);
        o.var_lb = put(mgr.lb.data(), mgr.lb.size() * 4);
        o.var_init = put(init.data(), init.size() * 4);
        o.sig_vars = put(mgr.sig_vars.data(), mgr.sig_vars.size() * 4);
        o.until_y = put(mgr.until_y.data(), mgr.until_y.size() * 4);
        o.firstvars = put(prog.firstvars.data(), prog.firstvars.size() * 4);
        o.trans = put(prog.trans.data(), prog.trans.size() * sizeof(TransDesc));
        o.transvals = put(prog.transvals.data(), prog.transvals.size() * 4);
        o.arr_off = put(mgr.array_off.data(), mgr.array_off.size() * 4);
        o.tables = put(prog.tables.data(), prog.tables.size() * 4);  // last: the part that may be big
        while (img.size() & 3) img.push_back(0u);
        o.words = (int)img.size();
        HIPCHK(d_img.upload(img));
        HIPCHK(d_code.upload(prog.code));
        ctx.img = d_img.p;
        ctx.o = o;
        ctx.code = d_code.p;
        ctx.nsets = (int)prog.sets.size();
        ctx.stack_slots = prog.max_stack + 2;
        const size_t scratch = (size_t)4 * ((kMaxLowVars + ctx.stack_slots) * 64 + ((ctx.NK + 63) & ~63)) * sizeof(int);
        if (scratch > 160 * 1024) return fail(STCSP_E_UNSUPPORTED, "expression stack too deep for LDS");
        // stage the image in LDS when image + scratch leave room for >= 2 workgroups per CU
        img_in_lds = (size_t)o.words * 4 + scratch <= 64 * 1024;
        if (const char *ev = getenv("STCSP_IMG_LDS")) img_in_lds = img_in_lds && atoi(ev) != 0;  // tuning switch
        lds_bytes = scratch + (img_in_lds ? (size_t)o.words * 4 : 0);
        // Grid = exactly the workgroups that are resident at once: wavefronts take node slots with
        // a static grid stride, so a workgroup that has to wait for a free CU slot would start its
        // share only when another one has finished all of its own (a 2x tail).
        {
            int per_cu = 0;
            hipError_t e;
            const void *fn;
            switch (DR) {
                case 1: fn = img_in_lds ? (const void *)k_expand<1, true> : (const void *)k_expand<1, false>; break;
                case 2: fn = img_in_lds ? (const void *)k_expand<2, true> : (const void *)k_expand<2, false>; break;
                default: fn = img_in_lds ? (const void *)k_expand<4, true> : (const void *)k_expand<4, false>; break;
            }
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes);
            hipDeviceProp_t prop;
            if (e == hipSuccess && per_cu > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess)
                max_blocks = per_cu * prop.multiProcessorCount;
            if (const char *ev = getenv("STCSP_BLOCKS")) if (atoi(ev) > 0) max_blocks = atoi(ev);
            // the persistent kernel has its own register footprint
            switch (DR) {
                case 1: fn = img_in_lds ? (const void *)k_persist<1, true> : (const void *)k_persist<1, false>; break;
                case 2: fn = img_in_lds ? (const void *)k_persist<2, true> : (const void *)k_persist<2, false>; break;
                default: fn = img_in_lds ? (const void *)k_persist<4, true> : (const void *)k_persist<4, false>; break;
            }
            per_cu = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes);
            if (e == hipSuccess && per_cu > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess)
                persist_blocks = per_cu * prop.multiProcessorCount;
            if (const char *ev = getenv("STCSP_PBLOCKS")) if (atoi(ev) > 0) persist_blocks = atoi(ev);
        }
        return STCSP_OK;
    }

    int create(const stcsp_problem *p, const stcsp_options *o) {
        if (o) opt = *o;
        if (opt.world <= 0) opt.world = 1;
        if (opt.rank < 0 || opt.rank >= opt.world) return fail(STCSP_E_INVALID, "rank %d outside world %d", opt.rank, opt.world);
        int rc = mgr.init(p, opt.world > 1);
        if (rc != STCSP_OK) return fail(rc, "%s", mgr.error.c_str());
        const int N = mgr.N, K = mgr.K;
        for (int v = 0; v < N; v++) {
            long long width = (long long)mgr.ub[v] - (long long)mgr.lb[v] + 1;
            if (width > 32)
                return fail(STCSP_E_UNSUPPORTED, "variable %d has %lld values; this engine packs at most 32 per bitset word", v, width);
        }
        if ((long long)N * K > 64 * kMaxDomRegs)
            return fail(STCSP_E_UNSUPPORTED, "N*K = %d exceeds the %d-word register-resident block", N * K, 64 * kMaxDomRegs);
        if (mgr.n_until_cons > 32) return fail(STCSP_E_UNSUPPORTED, "more than 32 until constraints");
        if (1 + mgr.n_sig + mgr.n_until_cons > 64) return fail(STCSP_E_UNSUPPORTED, "signature longer than 63 words");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(STCSP_E_DEVICE, "no HIP device available");
        device = opt.device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&stream));
        L = CtlLayout(opt.world);
        ctx.N = N;
        ctx.K = K;
        ctx.NK = N * K;
        ctx.NS = node_stride(N, K);
        ctx.sig_len = mgr.n_sig + mgr.n_until_cons;
        ctx.n_sig = mgr.n_sig;
        ctx.n_until_cons = mgr.n_until_cons;
        ctx.KL = 1 + ctx.sig_len;
        ctx.CS = cand_stride(N, K, ctx.sig_len);
        ctx.ES = edge_stride(N);
        ctx.world = opt.world;
        ctx.rank = opt.rank;
        DR = (N * K + 63) / 64;
        if (DR == 3) DR = 4;
        HIPCHK(d_arr_data.upload(mgr.array_data));
        ctx.arr_data = d_arr_data.p;
        rc = upload_program();
        if (rc != STCSP_OK) return rc;
        // pools
        int batch = opt.batch_nodes > 0 ? opt.batch_nodes : 65536;
        chunk_r = std::max(1, (batch + R - 1) / R);
        HIPCHK(d_ctl.alloc(L.words));
        HIPCHK(hipHostMalloc((void **)&h_ctl, L.words * sizeof(uint32_t)));
        ctx.ctl = d_ctl.p;
        ctx.miss_cap = 4096;
        HIPCHK(d_miss.alloc((size_t)ctx.miss_cap * kMissStride));
        HIPCHK(hipHostMalloc((void **)&h_miss, (size_t)ctx.miss_cap * kMissStride * sizeof(int)));
        ctx.miss = d_miss.p;
        HIPCHK(d_stats.alloc(kStatSlots * kStatWords));
        ctx.stats = d_stats.p;
        // STCSP_SMALL_POOLS=1 (tests): start with tiny pools so that every growth path is exercised
        const bool small_pools = getenv("STCSP_SMALL_POOLS") && atoi(getenv("STCSP_SMALL_POOLS")) != 0;
        rc = alloc_table(small_pools ? 64u : 1u << 20);
        if (rc != STCSP_OK) return rc;
        rc = alloc_states(small_pools ? 16u : 1u << 19);
        if (rc != STCSP_OK) return rc;
        rc = alloc_edges(small_pools ? 8u : 1u << 15);
        if (rc != STCSP_OK) return rc;
        // outbox: [owner][region] x cand_cap records. Unsharded: emptied after every launch.
        cand_cap = (uint32_t)(opt.world > 1 ? std::max(4 * chunk_r, 4096) : chunk_r);
        HIPCHK(d_cand.alloc((size_t)opt.world * R * cand_cap * ctx.CS));
        if (opt.world > 1) HIPCHK(d_pack.alloc((size_t)R * cand_cap * ctx.CS));
        // arena of node segments (grown on demand)
        HIPCHK(d_arena.alloc(small_pools ? (size_t)4 * R * ctx.NS : (size_t)8 * R * chunk_r * ctx.NS));
        HIPCHK(d_plan.alloc(1));
        HIPCHK(d_ctx.alloc(1));
        HIPCHK(hipHostMalloc((void **)&h_ctx, sizeof(Ctx)));
        HIPCHK(hipHostMalloc((void **)&h_plan, sizeof(Plan)));
        if (const char *ev = getenv("STCSP_BURST")) burst = std::max(1, atoi(ev));
        if (const char *ev = getenv("STCSP_PERSIST")) persist = atoi(ev) != 0;
        sync_ctx();
        return STCSP_OK;
    }

    int alloc_table(uint32_t slots) {
        HIPCHK(d_slots.alloc(slots));
        HIPCHK(hipMemsetAsync(d_slots.p, 0, (size_t)slots * sizeof(unsigned long long), stream));
        ctx.slots = d_slots.p;
        ctx.slot_mask = slots - 1;
        return STCSP_OK;
    }
    int alloc_states(uint32_t cap) {
        DevBuf<uint32_t> nb;
        HIPCHK(nb.alloc((size_t)cap * ctx.KL));
        if (d_state_keys.p && n_states)
            HIPCHK(hipMemcpyAsync(nb.p, d_state_keys.p, (size_t)n_states * ctx.KL * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::swap(d_state_keys.p, nb.p);
        std::swap(d_state_keys.n, nb.n);
        ctx.state_keys = d_state_keys.p;
        ctx.state_cap = cap;
        return STCSP_OK;
    }
    int alloc_edges(uint32_t cap) {
        DevBuf<uint32_t> nb;
        HIPCHK(nb.alloc((size_t)R * cap * ctx.ES));
        if (d_edges.p)
            for (int r = 0; r < R; r++)
                if (edge_count[r])
                    HIPCHK(hipMemcpyAsync(nb.p + (size_t)r * cap * ctx.ES, d_edges.p + (size_t)r * ctx.edge_cap * ctx.ES,
                                          (size_t)edge_count[r] * ctx.ES * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::swap(d_edges.p, nb.p);
        std::swap(d_edges.n, nb.n);
        ctx.edges = d_edges.p;
        ctx.edge_cap = cap;
        return STCSP_OK;
    }
    void sync_ctx() {
        ctx.plan = d_plan.p;
        ctx.arena = d_arena.p;
        ctx.cand = d_cand.p;
        ctx.pq = d_pq.p;
        ctx.ring = d_ring.p;
        ctx.seq = d_seq.p;
        ctx.pstack = d_pstack.p;
        ctx.parked = d_parked.p;
    }
    int flush_ctx() {
        sync_ctx();
        if (memcmp(h_ctx, &ctx, sizeof(Ctx)) != 0) {  // pools / program moved: refresh the device copy
            HIPCHK(hipStreamSynchronize(stream));      // (the staging copy may still be in flight)
            memcpy(h_ctx, &ctx, sizeof(Ctx));
            HIPCHK(hipMemcpyAsync(d_ctx.p, h_ctx, sizeof(Ctx), hipMemcpyHostToDevice, stream));
        }
        return STCSP_OK;
    }
    static constexpr size_t kPlanHeader = offsetof(Plan, stack);

    // pool growth (between bursts, when the device plan asks for it)
    int grow_edges() {
        if (ctx.edge_cap > 0x3fffffffu) return fail(STCSP_E_NOMEM, "edge log too large");
        int rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        return alloc_edges(ctx.edge_cap * 2);
    }
    int grow_states() {
        if (ctx.state_cap > 0x3fffffffu) return fail(STCSP_E_NOMEM, "state pool too large");
        int rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        return alloc_states(ctx.state_cap * 2);
    }
    int grow_table() {
        uint64_t slots = ((uint64_t)ctx.slot_mask + 1) * 2;
        if (slots > (1ull << 31)) return fail(STCSP_E_NOMEM, "state table too large");
        int rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        rc = alloc_table((uint32_t)slots);
        if (rc != STCSP_OK) return rc;
        if (n_states) hipLaunchKernelGGL(k_rehash, dim3((n_states + 255) / 256), dim3(256), 0, stream, ctx, n_states);
        HIPCHK(hipGetLastError());
        return STCSP_OK;
    }
    int grow_arena(size_t min_words) {
        size_t want = std::max(d_arena.n * 2, min_words);
        DevBuf<uint32_t> nb;
        if (nb.alloc(want) != hipSuccess) return fail(STCSP_E_NOMEM, "cannot grow the frontier arena to %zu MiB", want * 4 >> 20);
        HIPCHK(hipMemcpyAsync(nb.p, d_arena.p, (size_t)h_plan->arena_top * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::swap(d_arena.p, nb.p);
        std::swap(d_arena.n, nb.n);
        return STCSP_OK;
    }
    // push the host-side capacities into the device plan (after any growth)
    int push_caps() {
        sync_ctx();
        h_plan->arena_words = d_arena.n;
        h_plan->slot_cap = (unsigned long long)ctx.slot_mask + 1;
        h_plan->edge_cap = ctx.edge_cap;
        h_plan->state_cap = ctx.state_cap;
        h_plan->cand_cap = cand_cap;
        HIPCHK(hipMemcpyAsync(&d_plan.p->arena_words, &h_plan->arena_words, sizeof(unsigned long long) * 2, hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemcpyAsync(&d_plan.p->edge_cap, &h_plan->edge_cap, sizeof(unsigned) * 3, hipMemcpyHostToDevice, stream));
        return STCSP_OK;
    }
    int read_plan() {
        HIPCHK(hipMemcpyAsync(h_plan, d_plan.p, kPlanHeader, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        levels = h_plan->rounds;
        return STCSP_OK;
    }
    int replan() {
        hipLaunchKernelGGL(k_replan, dim3(1), dim3(64), 0, stream, ctx);
        HIPCHK(hipGetLastError());
        return STCSP_OK;
    }

    int begin() {
        HIPCHK(hipSetDevice(device));
        std::fill(edge_count.begin(), edge_count.end(), 0u);
        n_states = 0;
        truncated = false;
        levels = 0;
        finished = false;
        ev_used = 0;
        seconds_expand_kernel = 0;
        expand_launches = 0;
        HIPCHK(hipMemsetAsync(d_ctl.p, 0, L.words * sizeof(uint32_t), stream));
        for (int i = 0; i < L.words; i++) h_ctl[i] = 0;
        HIPCHK(hipMemsetAsync(d_stats.p, 0, kStatSlots * kStatWords * sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(d_slots.p, 0, ((size_t)ctx.slot_mask + 1) * sizeof(unsigned long long), stream));
        memset(h_plan, 0, sizeof(Plan));
        h_plan->chunk_r = chunk_r;
        h_plan->world = opt.world;
        if (opt.rank == 0) {
            // root state: Signature({}, 0) (solveralgorithm.cpp:951-954) = local state 0 of shard 0.
            // With an empty signature a leaf of set 0 must find it again, so the key is the plain
            // (tag 0); otherwise a reserved tag keeps it apart from a state with an all-zero signature.
            std::vector<uint32_t> key(ctx.KL, 0u);
            key[0] = ctx.sig_len == 0 ? 0u : kRootTag;
            unsigned long long h = kHashSeed;
            for (int j = 0; j < ctx.KL; j++) h = mix64(h, key[j]);
            h = mix_final(h);
            uint32_t htag = (uint32_t)(h >> 32) | 0x80000000u;
            unsigned long long slot = ((unsigned long long)htag << 32) | 0u;
            HIPCHK(hipMemcpyAsync(d_state_keys.p, key.data(), ctx.KL * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_slots.p + ((uint32_t)h & ctx.slot_mask), &slot, sizeof slot, hipMemcpyHostToDevice, stream));
            uint32_t one = 1;
            HIPCHK(hipMemcpyAsync(d_ctl.p + L.misc0 + MISC_NSTATES * CST, &one, sizeof one, hipMemcpyHostToDevice, stream));
            n_states = 1;
            // root search node: initial domains at every point (variable.cpp:24-29), set 0
            std::vector<uint32_t> node(ctx.NS, 0u);
            for (int p = 0; p < ctx.K; p++)
                for (int v = 0; v < ctx.N; v++) {
                    int w = mgr.ub[v] - mgr.lb[v] + 1;
                    node[4 + p * ctx.N + v] = w >= 32 ? 0xffffffffu : ((1u << w) - 1u);
                }
            HIPCHK(hipMemcpyAsync(d_arena.p, node.data(), ctx.NS * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            h_plan->sp = 1;
            h_plan->stack[0].base = 0;
            h_plan->stack[0].cap = 1;
            h_plan->stack[0].count[0] = 1;
            h_plan->arena_top = (unsigned long long)R * ctx.NS;
            h_plan->open_total = 1;
        }
        h_plan->status = PS_DONE;
        HIPCHK(hipMemcpyAsync(d_plan.p, h_plan, kPlanHeader + sizeof(DevSegment), hipMemcpyHostToDevice, stream));
        int rc = push_caps();
        if (rc != STCSP_OK) return rc;
        HIPCHK(hipStreamSynchronize(stream));
        begun = true;
        t_begin = std::chrono::steady_clock::now();
        return STCSP_OK;
    }

    double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); }

    int read_ctl() {
        HIPCHK(hipMemcpyAsync(h_ctl, d_ctl.p, L.words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (int r = 0; r < R; r++) edge_count[r] = h_ctl[L.edge0 + r * CST];
        n_states = h_ctl[L.misc0 + MISC_NSTATES * CST];
        uint32_t e = h_ctl[L.misc0 + MISC_ERROR * CST];
        if (e) {
            static const char *names[] = {"", "watchdog", "table spin", "edge overflow", "state overflow", "unknown set",
                                          "empty domain", "frontier overflow", "candidate overflow"};
            return fail(e == ERR_WATCHDOG ? STCSP_E_INTERNAL : STCSP_E_NOMEM, "device reported error %u (%s)", e, e < 9 ? names[e] : "?");
        }
        return STCSP_OK;
    }

    template <int DRT>
    void launch_expand() {
        if (img_in_lds)
            hipLaunchKernelGGL((k_expand<DRT, true>), dim3(max_blocks), dim3(256), lds_bytes, stream, (const Ctx *)d_ctx.p);
        else
            hipLaunchKernelGGL((k_expand<DRT, false>), dim3(max_blocks), dim3(256), lds_bytes, stream, (const Ctx *)d_ctx.p);
    }

    int service_misses() {
        uint32_t nm = h_ctl[L.misc0 + MISC_NMISS * CST];
        if (!nm) return STCSP_OK;
        uint32_t n = std::min<uint32_t>(nm, (uint32_t)ctx.miss_cap);
        HIPCHK(hipMemcpyAsync(h_miss, d_miss.p, (size_t)n * kMissStride * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < n; i++) {
            const int *rec = h_miss + (size_t)i * kMissStride;
            std::vector<int> vals(rec + 2, rec + 2 + rec[1]);
            int ns = mgr.transition(rec[0], vals);
            if (ns < 0) return fail(ns, "%s", mgr.error.c_str());
        }
        uint32_t zero = 0;
        HIPCHK(hipMemcpyAsync(d_ctl.p + L.misc0 + MISC_NMISS * CST, &zero, sizeof zero, hipMemcpyHostToDevice, stream));
        return upload_program();
    }

    // Enqueue bursts of rounds until the device plan stops: done, outbox full (sharded), or
    // truncated by a budget. Pool growth and constraint-set translation are served in between.
    int run_rounds() {
        int rc = replan();
        if (rc != STCSP_OK) return rc;
        const bool prof = opt.flags & STCSP_F_PROFILE;
        for (;;) {
            if ((rc = flush_ctx())) return rc;
            for (int k = 0; k < burst; k++) {
                if (prof) {
                    if (ev_used == ev_pool.size()) {
                        hipEvent_t e0, e1;
                        HIPCHK(hipEventCreate(&e0));
                        HIPCHK(hipEventCreate(&e1));
                        ev_pool.emplace_back(e0, e1);
                    }
                    HIPCHK(hipEventRecord(ev_pool[ev_used].first, stream));
                }
                switch (DR) {
                    case 1: launch_expand<1>(); break;
                    case 2: launch_expand<2>(); break;
                    default: launch_expand<4>(); break;
                }
                HIPCHK(hipGetLastError());
                if (prof) HIPCHK(hipEventRecord(ev_pool[ev_used++].second, stream));
                expand_launches++;
            }
            rc = read_plan();
            if (rc != STCSP_OK) return rc;
            switch (h_plan->status) {
                case PS_RUN: break;
                case PS_DONE:
                case PS_OUTBOX_FULL: return STCSP_OK;
                case PS_NEED_ARENA: {
                    size_t need = (size_t)h_plan->arena_top + (size_t)R * 3 * chunk_r * ctx.NS;
                    if ((rc = grow_arena(need)) || (rc = push_caps()) || (rc = replan())) return rc;
                    break;
                }
                case PS_NEED_EDGES:
                    if ((rc = grow_edges()) || (rc = push_caps()) || (rc = replan())) return rc;
                    break;
                case PS_NEED_STATES:
                    if ((rc = grow_states()) || (rc = push_caps()) || (rc = replan())) return rc;
                    break;
                case PS_NEED_TABLE:
                    if ((rc = grow_table()) || (rc = push_caps()) || (rc = replan())) return rc;
                    break;
                case PS_HOST:
                    if ((rc = read_ctl())) return rc;  // reports device errors
                    if ((rc = service_misses())) return rc;
                    sync_ctx();
                    if ((rc = replan())) return rc;
                    break;
                case PS_STACK_FULL: return fail(STCSP_E_NOMEM, "frontier segment stack deeper than %d", kMaxSegments);
                default: return fail(STCSP_E_INTERNAL, "unknown plan status %d", h_plan->status);
            }
            if (over_budget()) {
                truncated = true;
                return STCSP_OK;
            }
        }
    }

    bool over_budget() {
        if (opt.time_limit_s > 0 && elapsed() > opt.time_limit_s) return true;
        if (opt.max_search_nodes > 0) {
            // cheap upper bound without a device read: every launch expands at most R*chunk_r nodes
            if (levels * (long long)R * chunk_r >= opt.max_search_nodes) {
                std::vector<unsigned long long> st(kStatSlots * kStatWords);
                if (hipMemcpy(st.data(), d_stats.p, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                    long long nodes = 0;
                    for (int s = 0; s < kStatSlots; s++) nodes += (long long)st[s * kStatWords + ST_NODES];
                    if (nodes >= opt.max_search_nodes) return true;
                }
            }
        }
        return false;
    }

    template <int DRT>
    void launch_persist() {
        if (img_in_lds)
            hipLaunchKernelGGL((k_persist<DRT, true>), dim3(persist_blocks), dim3(256), lds_bytes, stream, (const Ctx *)d_ctx.p);
        else
            hipLaunchKernelGGL((k_persist<DRT, false>), dim3(persist_blocks), dim3(256), lds_bytes, stream, (const Ctx *)d_ctx.p);
    }
    // Persistent mode. Returns 1 when the run has to be redone round-based (a fixed-size pool
    // overflowed: the round-based path grows pools between launches), 0 on success, < 0 on error.
    int solve_persistent() {
        const uint32_t qcap = 1u << 16;
        const int pstk = 64, park_cap = 4096;
        const size_t nwaves = (size_t)persist_blocks * 4;
        if (!d_ring.p) {
            HIPCHK(d_pq.alloc(PQ_WORDS));
            HIPCHK(d_ring.alloc((size_t)qcap * ctx.NS));
            HIPCHK(d_seq.alloc(qcap));
            HIPCHK(d_parked.alloc((size_t)park_cap * ctx.NS));
        }
        if (d_pstack.n < nwaves * pstk * ctx.NS) HIPCHK(d_pstack.alloc(nwaves * pstk * ctx.NS));
        ctx.qmask = qcap - 1;
        ctx.pstk_cap = pstk;
        ctx.park_cap = park_cap;
        ctx.hungry = (int)nwaves;  // wavefronts in the grid (idle = hungry - active)
        int rc = begin();  // common initialisation: cursors, statistics, table, root state 0
        if (rc != STCSP_OK) return rc;
        std::vector<uint32_t> seqs(qcap);
        // the ring starts with the root node (written to the arena by begin())
        std::vector<uint32_t> pending((size_t)ctx.NS);
        HIPCHK(hipMemcpy(pending.data(), d_arena.p, (size_t)ctx.NS * sizeof(uint32_t), hipMemcpyDeviceToHost));
        size_t n_pending = 1;
        for (;;) {
            // (re)load the ring with the pending nodes: slots [0, n) full, the rest empty
            for (uint32_t i = 0; i < qcap; i++) seqs[i] = i < n_pending ? i + 1 : i;
            std::vector<uint32_t> pq(PQ_WORDS, 0u);
            pq[PQ_TAIL] = (uint32_t)n_pending;
            pq[PQ_PENDING] = (uint32_t)n_pending;
            HIPCHK(hipMemcpyAsync(d_seq.p, seqs.data(), qcap * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_pq.p, pq.data(), PQ_WORDS * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(d_ring.p, pending.data(), n_pending * ctx.NS * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            if ((rc = flush_ctx())) return rc;
            HIPCHK(hipStreamSynchronize(stream));  // staging vectors are pageable
            const bool prof = opt.flags & STCSP_F_PROFILE;
            if (prof) {
                if (ev_used == ev_pool.size()) {
                    hipEvent_t e0, e1;
                    HIPCHK(hipEventCreate(&e0));
                    HIPCHK(hipEventCreate(&e1));
                    ev_pool.emplace_back(e0, e1);
                }
                HIPCHK(hipEventRecord(ev_pool[ev_used].first, stream));
            }
            switch (DR) {
                case 1: launch_persist<1>(); break;
                case 2: launch_persist<2>(); break;
                default: launch_persist<4>(); break;
            }
            HIPCHK(hipGetLastError());
            if (prof) HIPCHK(hipEventRecord(ev_pool[ev_used++].second, stream));
            expand_launches++;
            levels++;
            HIPCHK(hipMemcpyAsync(pq.data(), d_pq.p, PQ_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_ctl, d_ctl.p, L.words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            const uint32_t dev_err = h_ctl[L.misc0 + MISC_ERROR * CST], ab = pq[PQ_ABORT];
            if (dev_err == ERR_EDGE_OVERFLOW || dev_err == ERR_STATE_OVERFLOW || ab == AB_QUEUE_FULL || ab == AB_PARK_FULL)
                return 1;  // fixed-size pool too small for this instance: redo with growing pools
            if (dev_err || ab) return fail(STCSP_E_INTERNAL, "persistent kernel stopped: device error %u, abort %u", dev_err, ab);
            n_pending = pq[PQ_PARKED];
            if (n_pending == 0) break;
            // leaves waiting for a constraint-set translation: translate, then run them again
            if ((rc = service_misses())) return rc;
            pending.resize(n_pending * ctx.NS);
            HIPCHK(hipMemcpy(pending.data(), d_parked.p, pending.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        return 0;
    }

    int solve_unsharded() {
        int rc;
        if (persist && opt.time_limit_s <= 0 && opt.max_search_nodes <= 0) {
            rc = solve_persistent();
            if (rc < 0) return rc;
            if (rc == 0) return finish();
            // fall through: redo round-based
        }
        rc = begin();
        if (rc != STCSP_OK) return rc;
        rc = run_rounds();
        if (rc != STCSP_OK) return rc;
        return finish();
    }

    int read_counters(stcsp_counters &ctr) {
        std::vector<unsigned long long> st(kStatSlots * kStatWords);
        HIPCHK(hipMemcpy(st.data(), d_stats.p, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long tot[kStatWords] = {0};
        for (int s = 0; s < kStatSlots; s++)
            for (int k = 0; k < kStatWords; k++) tot[k] += st[s * kStatWords + k];
        ctr = stcsp_counters{};
        ctr.search_nodes = (int64_t)(tot[ST_NODES] - tot[ST_REQUEUE]);
        ctr.gac_calls = (int64_t)tot[ST_NODES];
        ctr.fails = (int64_t)tot[ST_FAILS];
        ctr.leaves = (int64_t)tot[ST_LEAVES];
        ctr.revisions = (int64_t)tot[ST_REVS];
        ctr.evaluations = (int64_t)tot[ST_EVALS];
        ctr.wave_revisions = (int64_t)tot[ST_WAVEREVS];
        ctr.sweeps = (int64_t)tot[ST_SWEEPS];
        ctr.skipped_revisions = (int64_t)tot[ST_SKIPPED];
        if (getenv("STCSP_DEBUG") && tot[ST_POLLS])
            fprintf(stderr, "[persist] pushes %llu pops %llu private pops %llu polls %llu waves that worked %llu; idle Mcycles %.1f busy Mcycles %.1f\n",
                    (unsigned long long)tot[ST_QPUSH], (unsigned long long)tot[ST_QPOP], (unsigned long long)tot[ST_PSTACK_POP],
                    (unsigned long long)tot[ST_POLLS], (unsigned long long)tot[ST_WAVES_WORKED], tot[ST_IDLE_CYC] / 1e6, tot[ST_BUSY_CYC] / 1e6);
#ifdef STCSP_PHASES
        if (tot[ST_NODES])
            fprintf(stderr, "[phases] cycles/node: load %.0f sweep %.0f wave %.0f classify+emit %.0f commit %.0f total %.0f (nodes %llu)\n",
                    (double)tot[ST_CYC_LOAD] / tot[ST_NODES], (double)tot[ST_CYC_SWEEP] / tot[ST_NODES], (double)tot[ST_CYC_WAVE] / tot[ST_NODES],
                    (double)tot[ST_CYC_CLASSIFY] / tot[ST_NODES], (double)tot[ST_CYC_COMMIT] / tot[ST_NODES],
                    (double)tot[ST_CYC_TOTAL] / tot[ST_NODES], (unsigned long long)tot[ST_NODES]);
#endif
        ctr.levels = levels;
        ctr.seconds_search = finished ? seconds_search : elapsed();
        ctr.seconds_expand_kernel = seconds_expand_kernel;
        ctr.expand_launches = expand_launches;
        return STCSP_OK;
    }
    int finish() {
        HIPCHK(hipStreamSynchronize(stream));
        seconds_search = elapsed();
        finished = true;
        // every enqueued launch counts (a burst may run a few no-op launches past the end), so that
        // the average agrees with rocprofv3's per-kernel average
        for (size_t i = 0; i < ev_used; i++) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, ev_pool[i].first, ev_pool[i].second));
            seconds_expand_kernel += ms * 1e-3;
        }
        ev_used = 0;
        int rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        return read_counters(snap);
    }

    // ---- sharded stepping
    int expand_local(int64_t *left) {
        if (!begun) return fail(STCSP_E_STATE, "expand_local before begin");
        int rc = run_rounds();
        if (rc != STCSP_OK) return rc;
        rc = read_ctl();  // outbox cursors for outbox()
        if (rc != STCSP_OK) return rc;
        if (left) *left = truncated ? 0 : (int64_t)h_plan->open_total;
        return STCSP_OK;
    }
    int outbox(int peer, void **ptr, int64_t *count) {
        if (peer < 0 || peer >= opt.world) return fail(STCSP_E_INVALID, "peer out of range");
        uint32_t total = 0;
        for (int r = 0; r < R; r++) total += h_ctl[L.cand0 + (peer * R + r) * CST];
        // pack this peer's regions; each peer gets its own slice of the pack buffer
        if (d_pack.n < (size_t)opt.world * R * cand_cap * ctx.CS) HIPCHK(d_pack.alloc((size_t)opt.world * R * cand_cap * ctx.CS));
        uint32_t *dst = d_pack.p + (size_t)peer * R * cand_cap * ctx.CS;
        if (total) {
            hipLaunchKernelGGL(k_pack, dim3(std::min<uint32_t>(1024, (total * ctx.CS + 255) / 256)), dim3(256), 0, stream,
                               d_cand.p + (size_t)peer * R * cand_cap * ctx.CS, cand_cap, ctx.CS, d_ctl.p, L.cand0 + peer * R * CST, dst);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(stream));
        *ptr = dst;
        *count = total;
        return STCSP_OK;
    }
    int commit(const void *records, int64_t count) {
        if (!begun) return fail(STCSP_E_STATE, "commit before begin");
        // the outbox has been handed over: empty it
        HIPCHK(hipMemsetAsync(d_ctl.p + L.cand0, 0, (size_t)(L.edge0 - L.cand0) * sizeof(uint32_t), stream));
        for (int i = L.cand0; i < L.edge0; i++) h_ctl[i] = 0;
        if (count <= 0) {
            HIPCHK(hipStreamSynchronize(stream));
            return STCSP_OK;
        }
        // room for `count` more edges / states and for the segment of new nodes
        int rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        rc = read_plan();
        if (rc != STCSP_OK) return rc;
        const long long per_region = (count + R - 1) / R + 1;
        uint32_t max_edges = 0;
        for (int r = 0; r < R; r++) max_edges = std::max(max_edges, edge_count[r]);
        while ((long long)max_edges + per_region > (long long)ctx.edge_cap)
            if ((rc = alloc_edges(ctx.edge_cap * 2))) return rc;
        while ((long long)n_states + count > (long long)ctx.state_cap)
            if ((rc = alloc_states(ctx.state_cap * 2))) return rc;
        while (((long long)n_states + count) * 2 > (long long)ctx.slot_mask + 1)
            if ((rc = grow_table())) return rc;
        const unsigned cap = (unsigned)per_region;
        if ((size_t)h_plan->arena_top + (size_t)R * cap * ctx.NS > d_arena.n)
            if ((rc = grow_arena((size_t)h_plan->arena_top + (size_t)R * cap * ctx.NS))) return rc;
        if ((rc = push_caps())) return rc;
        hipLaunchKernelGGL(k_open_segment, dim3(1), dim3(64), 0, stream, ctx, cap);
        CommitArgs ca{};
        ca.cand_base = (const uint32_t *)records;
        ca.total = count;
        {
            dim3 grid((unsigned)((count + 3) / 4)), block(256);
            switch (DR) {
                case 1: hipLaunchKernelGGL((k_commit<1>), grid, block, 0, stream, ctx, ca); break;
                case 2: hipLaunchKernelGGL((k_commit<2>), grid, block, 0, stream, ctx, ca); break;
                default: hipLaunchKernelGGL((k_commit<4>), grid, block, 0, stream, ctx, ca); break;
            }
        }
        hipLaunchKernelGGL(k_close_segment, dim3(1), dim3(64), 0, stream, ctx);
        HIPCHK(hipGetLastError());
        rc = read_ctl();
        if (rc != STCSP_OK) return rc;
        return read_plan();
    }

    // unsharded export: ok-fixpoint + compaction on the device, result arrays land in pinned memory
    int export_device(stcsp_result *res, stcsp_counters &ctr, size_t &E_out) {
        const int KL = ctx.KL, sl = ctx.sig_len, N = ctx.N;
        size_t E = 0;
        EdgeView v{};
        v.edges = d_edges.p;
        v.edge_cap = ctx.edge_cap;
        v.ES = ctx.ES;
        v.N = N;
        for (int r = 0; r < R; r++) {
            v.pref[r] = (uint32_t)E;
            E += edge_count[r];
        }
        v.pref[R] = (uint32_t)E;
        if (E > 0xfffffff0ull) return fail(STCSP_E_NOMEM, "edge log too large for the device export");
        if (d_alive.n < E) HIPCHK(d_alive.alloc(E + E / 4 + 256));
        if (d_osrc.n < E) {
            size_t cap = E + E / 4 + 256;
            HIPCHK(d_osrc.alloc(cap));
            HIPCHK(d_odst.alloc(cap));
            HIPCHK(d_oval.alloc(cap * N));
        }
        if (d_fail.n < n_states) {
            HIPCHK(d_fail.alloc((size_t)n_states + n_states / 4 + 256));
            HIPCHK(d_outdeg.alloc((size_t)n_states + n_states / 4 + 256));
        }
        if (!d_post.p) HIPCHK(d_post.alloc(4));
        if (h_edge_cap < E) {
            if (h_osrc) (void)hipHostFree(h_osrc);
            if (h_odst) (void)hipHostFree(h_odst);
            if (h_oval) (void)hipHostFree(h_oval);
            h_edge_cap = E + E / 4 + 256;
            HIPCHK(hipHostMalloc((void **)&h_osrc, h_edge_cap * sizeof(long long)));
            HIPCHK(hipHostMalloc((void **)&h_odst, h_edge_cap * sizeof(long long)));
            HIPCHK(hipHostMalloc((void **)&h_oval, h_edge_cap * N * sizeof(int32_t)));
        }
        if (h_state_cap < n_states) {
            if (h_keys) (void)hipHostFree(h_keys);
            if (h_fail) (void)hipHostFree(h_fail);
            h_state_cap = (size_t)n_states + n_states / 4 + 256;
            HIPCHK(hipHostMalloc((void **)&h_keys, h_state_cap * KL * sizeof(uint32_t)));
            HIPCHK(hipHostMalloc((void **)&h_fail, h_state_cap));
        }
        const bool dbg = getenv("STCSP_DEBUG") != nullptr;
        auto tA = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!dbg) return;
            (void)hipStreamSynchronize(stream);
            auto tB = std::chrono::steady_clock::now();
            fprintf(stderr, "[export] %-22s %.3f ms\n", what, std::chrono::duration<double>(tB - tA).count() * 1e3);
            tA = tB;
        };
        HIPCHK(hipMemcpyAsync(h_keys, d_state_keys.p, (size_t)n_states * KL * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemsetAsync(d_fail.p, 0, n_states, stream));
        HIPCHK(hipMemsetAsync(d_outdeg.p, 0, (size_t)n_states * sizeof(uint32_t), stream));
        HIPCHK(hipMemsetAsync(d_post.p, 0, 4 * sizeof(uint32_t), stream));
        const unsigned eb = (unsigned)((E + 255) / 256), sb = (n_states + 255) / 256;
        uint32_t live = 0;
        if (E) {
            hipLaunchKernelGGL(k_post_outdeg, dim3(eb), dim3(256), 0, stream, v, d_outdeg.p, d_alive.p);
            for (int it = 0;; it++) {
                hipLaunchKernelGGL(k_post_mark, dim3(sb), dim3(256), 0, stream, n_states, (const uint32_t *)d_outdeg.p, d_fail.p, d_post.p);
                uint32_t changed = 0;
                HIPCHK(hipMemcpyAsync(&changed, d_post.p, sizeof changed, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if (!changed) break;
                HIPCHK(hipMemsetAsync(d_post.p, 0, sizeof(uint32_t), stream));
                hipLaunchKernelGGL(k_post_kill, dim3(eb), dim3(256), 0, stream, v, d_alive.p, (const uint8_t *)d_fail.p, d_outdeg.p);
                if (it > (int)n_states + 8) return fail(STCSP_E_INTERNAL, "ok-fixpoint did not converge");
            }
            lap("fixpoint");
            hipLaunchKernelGGL(k_post_compact, dim3(eb), dim3(256), 0, stream, v, (const uint8_t *)d_alive.p, d_post.p + 1, d_osrc.p, d_odst.p, d_oval.p);
            HIPCHK(hipGetLastError());
            lap("compact");
            HIPCHK(hipMemcpyAsync(&live, d_post.p + 1, sizeof live, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            HIPCHK(hipMemcpyAsync(h_osrc, d_osrc.p, (size_t)live * sizeof(long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_odst, d_odst.p, (size_t)live * sizeof(long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(h_oval, d_oval.p, (size_t)live * N * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        } else {
            // no edges at all: every non-root state is failed
            hipLaunchKernelGGL(k_post_mark, dim3(sb), dim3(256), 0, stream, n_states, (const uint32_t *)d_outdeg.p, d_fail.p, d_post.p);
        }
        HIPCHK(hipMemcpyAsync(h_fail, d_fail.p, n_states, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        lap("D2H result arrays");
        r_cid.assign(n_states, 0);
        r_sig.assign((size_t)n_states * std::max(sl, 0), 0);
        int64_t ok_states = 0;
        for (uint32_t i = 0; i < n_states; i++) {
            uint32_t tag = h_keys[(size_t)i * KL];
            r_cid[i] = (tag == kRootTag) ? 0 : (int32_t)tag;
            for (int j = 0; j < sl; j++) r_sig[(size_t)i * sl + j] = (int32_t)h_keys[(size_t)i * KL + 1 + j];
            if (i) ok_states += !h_fail[i];
        }
        ctr.dominance = (int64_t)live - ok_states;  // every ok non-root state is entered by exactly one creating leaf
        lap("host key split");
        res->state_fail = h_fail;
        res->edge_src = (const int64_t *)h_osrc;
        res->edge_dst = (const int64_t *)h_odst;
        res->edge_values = h_oval;
        E_out = live;
        return STCSP_OK;
    }

    int export_result(stcsp_result *res) {
        auto t0 = std::chrono::steady_clock::now();
        HIPCHK(hipSetDevice(device));
        const int KL = ctx.KL, sl = ctx.sig_len, N = ctx.N, ES = ctx.ES;
        if (opt.world == 1 && !(opt.flags & STCSP_F_KEEP_RAW_EDGES) && !getenv("STCSP_HOST_EXPORT")) {
            stcsp_counters ctr{};
            int rcc = read_counters(ctr);
            if (rcc != STCSP_OK) return rcc;
            memset(res, 0, sizeof *res);
            size_t E = 0;
            rcc = export_device(res, ctr, E);
            if (rcc != STCSP_OK) return rcc;
            r_issig.assign(mgr.is_sig.begin(), mgr.is_sig.end());
            res->n_states = n_states;
            res->sig_len = sl;
            res->n_sig_vars = mgr.n_sig;
            res->n_until = mgr.n_until;
            res->n_until_cons = mgr.n_until_cons;
            res->state_cid = r_cid.data();
            res->state_sig = r_sig.data();
            res->n_edges = (int64_t)E;
            res->n_vars = N;
            res->n_constraint_sets = (int32_t)mgr.sets.size();
            res->var_is_signature = r_issig.data();
            res->root_final = mgr.n_until_cons == 0;
            res->truncated = truncated;
            seconds_export = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (getenv("STCSP_DEBUG")) fprintf(stderr, "[export] whole export_result    %.3f ms\n", seconds_export * 1e3);
            ctr.seconds_export = seconds_export;
            res->counters = ctr;
            return STCSP_OK;
        }
        std::vector<uint32_t> keys((size_t)n_states * KL);
        if (n_states) HIPCHK(hipMemcpy(keys.data(), d_state_keys.p, keys.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        r_cid.assign(n_states, 0);
        r_sig.assign((size_t)n_states * std::max(sl, 0), 0);
        for (uint32_t i = 0; i < n_states; i++) {
            uint32_t tag = keys[(size_t)i * KL];
            r_cid[i] = (tag == kRootTag) ? 0 : (int32_t)tag;
            for (int j = 0; j < sl; j++) r_sig[(size_t)i * sl + j] = (int32_t)keys[(size_t)i * KL + 1 + j];
        }
        size_t E = 0;
        for (int r = 0; r < R; r++) E += edge_count[r];
        r_esrc.resize(E);
        r_edst.resize(E);
        r_eval.resize(E * N);
        std::vector<uint32_t> buf;
        size_t e = 0;
        for (int r = 0; r < R; r++) {
            if (!edge_count[r]) continue;
            buf.resize((size_t)edge_count[r] * ES);
            HIPCHK(hipMemcpy(buf.data(), d_edges.p + (size_t)r * ctx.edge_cap * ES, buf.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (uint32_t k = 0; k < edge_count[r]; k++, e++) {
                const uint32_t *er = buf.data() + (size_t)k * ES;
                int64_t src = (int64_t)(((uint64_t)er[1] << 32) | er[0]);
                int64_t dst = ((int64_t)opt.rank << STCSP_GID_SHIFT) | er[2];
                r_esrc[e] = src;
                r_edst[e] = dst;
                memcpy(&r_eval[e * N], er + 4, (size_t)N * sizeof(int32_t));
            }
        }
        stcsp_counters ctr{};
        {
            int rcc = read_counters(ctr);
            if (rcc != STCSP_OK) return rcc;
        }
        r_fail.assign(n_states, 0);
        if (opt.world == 1) {
            // ok-fixpoint over the raw leaf-edge log (okfix.hpp); sharded runs do it after the merge
            std::vector<uint8_t> alive;
            ok_fixpoint(n_states, r_esrc, r_edst, r_fail, alive);
            if (!(opt.flags & STCSP_F_KEEP_RAW_EDGES)) {
                size_t w = 0;
                for (size_t k = 0; k < E; k++)
                    if (alive[k]) {
                        r_esrc[w] = r_esrc[k];
                        r_edst[w] = r_edst[k];
                        if (w != k) memmove(&r_eval[w * N], &r_eval[k * N], (size_t)N * sizeof(int32_t));
                        w++;
                    }
                r_esrc.resize(w);
                r_edst.resize(w);
                r_eval.resize(w * N);
                E = w;
            }
            int64_t ok_states = 0, live = 0;
            for (uint32_t v = 1; v < n_states; v++) ok_states += !r_fail[v];
            for (size_t k = 0; k < alive.size(); k++) live += alive[k];
            ctr.dominance = live - ok_states;  // every ok non-root state is entered by exactly one creating leaf
        }
        r_issig.assign(mgr.is_sig.begin(), mgr.is_sig.end());
        memset(res, 0, sizeof *res);
        res->n_states = n_states;
        res->sig_len = sl;
        res->n_sig_vars = mgr.n_sig;
        res->n_until = mgr.n_until;
        res->n_until_cons = mgr.n_until_cons;
        res->state_cid = r_cid.data();
        res->state_sig = r_sig.data();
        res->state_fail = r_fail.data();
        res->n_edges = (int64_t)E;
        res->edge_src = r_esrc.data();
        res->edge_dst = r_edst.data();
        res->edge_values = r_eval.data();
        res->n_vars = N;
        res->n_constraint_sets = (int32_t)mgr.sets.size();
        res->var_is_signature = r_issig.data();
        res->root_final = mgr.n_until_cons == 0;
        res->truncated = truncated;
        seconds_export = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ctr.seconds_export = seconds_export;
        res->counters = ctr;
        return STCSP_OK;
    }
};

// ------------------------------------------------------------------ C-ABI
extern "C" {

int stcsp_engine_create(const stcsp_problem *problem, const stcsp_options *options, stcsp_engine **out) {
    if (!problem || !out) {
        g_create_error = "null argument";
        return STCSP_E_INVALID;
    }
    std::unique_ptr<stcsp_engine> e(new stcsp_engine());
    int rc = e->create(problem, options);
    if (rc != STCSP_OK) {
        g_create_error = e->err;
        return rc;
    }
    *out = e.release();
    return STCSP_OK;
}

int stcsp_engine_solve(stcsp_engine *e, stcsp_result *result) {
    if (!e || !result) return STCSP_E_INVALID;
    if (e->opt.world != 1) return e->fail(STCSP_E_STATE, "solve() is the unsharded entry point; use the stepping calls when world > 1");
    int rc = e->solve_unsharded();
    if (rc != STCSP_OK) return rc;
    if (e->opt.flags & STCSP_F_NO_EXPORT) {
        memset(result, 0, sizeof *result);
        result->truncated = e->truncated;
        result->counters = e->snap;
        return STCSP_OK;
    }
    return e->export_result(result);
}

int stcsp_engine_export(stcsp_engine *e, stcsp_result *result) {
    if (!e || !result) return STCSP_E_INVALID;
    if (!e->begun) return e->fail(STCSP_E_STATE, "export before a solve");
    return e->export_result(result);
}

void stcsp_engine_destroy(stcsp_engine *e) { delete e; }

const char *stcsp_engine_last_error(const stcsp_engine *e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int stcsp_engine_begin(stcsp_engine *e) { return e ? e->begin() : STCSP_E_INVALID; }
int stcsp_engine_expand_local(stcsp_engine *e, int64_t *left) { return e ? e->expand_local(left) : STCSP_E_INVALID; }
int stcsp_engine_candidate_bytes(const stcsp_engine *e) { return e ? e->ctx.CS * 4 : STCSP_E_INVALID; }
int stcsp_engine_outbox(stcsp_engine *e, int peer, void **ptr, int64_t *count) {
    if (!e || !ptr || !count) return STCSP_E_INVALID;
    return e->outbox(peer, ptr, count);
}
int stcsp_engine_commit(stcsp_engine *e, const void *records, int64_t count) { return e ? e->commit(records, count) : STCSP_E_INVALID; }
int stcsp_engine_finish(stcsp_engine *e) { return e ? e->finish() : STCSP_E_INVALID; }
int stcsp_engine_counters(stcsp_engine *e, stcsp_counters *out) {
    if (!e || !out) return STCSP_E_INVALID;
    if (!e->begun) return e->fail(STCSP_E_STATE, "counters before a solve");
    return e->read_counters(*out);
}
int stcsp_engine_sets_blob(stcsp_engine *e, const int32_t **words, int64_t *n_words) {
    if (!e || !words || !n_words) return STCSP_E_INVALID;
    e->blob.clear();
    e->blob.push_back((int32_t)e->mgr.sets.size());
    for (size_t i = 0; i < e->mgr.sets.size(); i++) {
        std::vector<int32_t> w = e->mgr.serialise_set((int)i);
        e->blob.push_back((int32_t)w.size());
        e->blob.insert(e->blob.end(), w.begin(), w.end());
    }
    *words = e->blob.data();
    *n_words = (int64_t)e->blob.size();
    return STCSP_OK;
}
int stcsp_engine_sets_import(stcsp_engine *e, const int32_t *words, int64_t n) {
    if (!e || !words || n < 1) return STCSP_E_INVALID;
    size_t before = e->mgr.sets.size();
    int64_t pos = 1;
    for (int32_t i = 0; i < words[0]; i++) {
        if (pos >= n) return e->fail(STCSP_E_INVALID, "truncated set blob");
        int32_t len = words[pos++];
        if (pos + len > n) return e->fail(STCSP_E_INVALID, "truncated set blob");
        int rc = e->mgr.import_set(words + pos, (size_t)len);
        if (rc < 0) return e->fail(rc, "%s", e->mgr.error.c_str());
        pos += len;
    }
    if (e->mgr.sets.size() != before) return e->upload_program();
    return STCSP_OK;
}

}  // extern "C"
