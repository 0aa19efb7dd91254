// dev_wide.hpp -- search nodes whose variables have MORE than 32 values (Variable::currLB/currUB take any width,
// reference src/variable.h:19-20; aux variables of arithmetic under next / fby get hull bounds,
// src/solveralgorithm.cpp:182-184): W = 2 or 4 bitset words per (variable, time point), |D| <= 128.
//
// Block layout (chunk-major): word(c, p, v) = c * N*K + p * N + v holds values lb[v] + 32 c ... lb[v] + 32 c + 31 of
// variable v at look-ahead point p. Everything that only MOVES blocks (node load / store, sibling stack, frontier, candidate
// and transfer records, commit) sees a block of Ctx::NK = W * N * K words and is shared with the one-word kernels; this file
// holds what has to know the chunks: propagation, classification, the leaf.
//
// Propagation here is the reference's own strength -- BOUNDS consistency (enforcePointConsistencyAt,
// src/solveralgorithm.cpp:476-523: new lb = least supported value, new ub = greatest supported value) -- with the support
// search (findSupportRe, :435-464) spread over the wavefront: the tuples of the other variables' current domains are
// enumerated 64 at a time, one `__ballot` per block, first support exits. Programs compiled for W > 1 have no lane-revised
// items and no eager arcs (cset.cpp compile): X == next Y, until and point constraints are all revised one at a time by the
// whole wavefront. Sound pruning + exact leaves => the same automaton as the reference.
#pragma once
#include "dev_propagate.hpp"
namespace stcsp {
namespace dev {

// ---- a W-word domain held by one lane
template <int W>
struct WDom {
    uint32_t w[W];
    __device__ __forceinline__ int count() const {
        int n = 0;
#pragma unroll
        for (int c = 0; c < W; c++) n += __popc(w[c]);
        return n;
    }
    __device__ __forceinline__ bool empty() const {
        uint32_t o = 0;
#pragma unroll
        for (int c = 0; c < W; c++) o |= w[c];
        return o == 0;
    }
    __device__ __forceinline__ int lowest() const {  // bit position of the least value (domain not empty)
        int r = 0;
        bool found = false;
#pragma unroll
        for (int c = 0; c < W; c++)
            if (!found && w[c]) {
                r = 32 * c + __ffs((int)w[c]) - 1;
                found = true;
            }
        return r;
    }
    __device__ __forceinline__ int highest() const {
        int r = 0;
#pragma unroll
        for (int c = 0; c < W; c++)
            if (w[c]) r = 32 * c + 31 - __clz((int)w[c]);
        return r;
    }
    __device__ __forceinline__ int kth(int k) const {  // position of the k-th (0-based) set bit, k < count()
        int r = 0;
        bool found = false;
#pragma unroll
        for (int c = 0; c < W; c++) {
            const int n = __popc(w[c]);
            if (!found && k < n) {
                r = 32 * c + select_kth_fast(w[c], k);
                found = true;
            }
            if (!found) k -= n;
        }
        return r;
    }
    __device__ __forceinline__ void clear(int bit) {
#pragma unroll
        for (int c = 0; c < W; c++)
            if ((bit >> 5) == c) w[c] &= ~(1u << (bit & 31));
    }
    __device__ __forceinline__ void only(int bit) {
#pragma unroll
        for (int c = 0; c < W; c++) w[c] = (bit >> 5) == c ? (1u << (bit & 31)) : 0u;
    }
    __device__ __forceinline__ bool operator!=(const WDom &o) const {
        bool ne = false;
#pragma unroll
        for (int c = 0; c < W; c++) ne = ne || w[c] != o.w[c];
        return ne;
    }
};
// the domain of block word `idx` (= p * N + v, per lane) out of the wavefront's LDS copy of the block
template <int W>
__device__ __forceinline__ WDom<W> wload(const int *ldom, int NK1, int idx) {
    WDom<W> d;
#pragma unroll
    for (int c = 0; c < W; c++) d.w[c] = (uint32_t)ldom[c * NK1 + idx];
    return d;
}
template <int W>
__device__ __forceinline__ WDom<W> wbroadcast(const WDom<W> &d, int srclane) {  // srclane wave-uniform
    WDom<W> o;
#pragma unroll
    for (int c = 0; c < W; c++) o.w[c] = rdlane(d.w[c], srclane);
    return o;
}

// Is there a tuple of the scope variables' domains `D` (lane j < s holds variable j's) that satisfies the constraint?
// (findSupportRe, src/solveralgorithm.cpp:435-464, with lanes over tuples.) Up to kMaxLowVars open variables are
// enumerated across lanes -- the product of their sizes in trips of 64, every lane decoding its own tuple -- the others by a
// wave-uniform odometer. Evaluation = tuple bitmap look-up or the postfix program. A search longer than `budget` trips is
// given up as "supported" (never prune without proof; leaves are exact: their product is 1).
template <int W, int L>
__device__ bool exists_support_wide(const Ctx &c, const Img<L> &G, const ConDesc &C, int lane, const WDom<W> &D, int vlb, int mystride,
                                    int *lds_vals, int *lds_stk, unsigned long long &n_evals, bool &gave_up) {
    const int s = C.scope_len;
    const bool use_bitmap = C.bitmap_off >= 0;
    const int n = lane < s ? D.count() : 1;
    const unsigned long long openm = __ballot(lane < s && n > 1);
    unsigned long long lowmask = 0, highmask = 0;
    unsigned Plow = 1;
    int nlow = 0;
    unsigned long long nsteps = 1;
    const unsigned long long budget = (unsigned long long)(unsigned)(use_bitmap ? c.budget_bitmap : c.budget_code) * 32ull;
    for (unsigned long long m = openm; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const unsigned nj = rdlane((uint32_t)n, j);
        if (nlow < kMaxLowVars && Plow * nj <= 65536u) {
            lowmask |= 1ull << j;
            Plow *= nj;
            nlow++;
        } else {
            highmask |= 1ull << j;
            if (nsteps <= budget) nsteps *= nj;
        }
    }
    const unsigned trips = (Plow + 63u) >> 6;
    if (nsteps > budget || nsteps * trips > budget) {
        gave_up = true;
        return true;
    }
    // scope lane j: 1 + slot for a lane-enumerated variable (the interpreter reads its value from lds_vals), else 0
    const uint32_t varinfo = ((lowmask >> lane) & 1ull) ? 1u + (uint32_t)__popcll(lowmask & ((1ull << lane) - 1ull)) : 0u;
    int curbit = lane < s ? D.lowest() : 0;  // singletons: their bit; odometer variables: the current bit
    int curval = vlb + curbit;
    int digit_h = 0;
    for (;;) {
        // wave-uniform part of the bitmap index: singletons and odometer variables
        int base_sum = 0;
        if (use_bitmap) base_sum = wave_sum((lane < s && !((lowmask >> lane) & 1ull)) ? curbit * mystride : 0);
        for (unsigned trip = 0; trip < trips; trip++) {
            const unsigned t = trip * 64u + (unsigned)lane;
            const bool active = t < Plow;
            unsigned u = active ? t : 0u;
            int lane_part = 0, slot = 0;
            for (unsigned long long m = lowmask; m; m &= m - 1, slot++) {
                const int j = __ffsll((long long)m) - 1;
                const unsigned nj = rdlane((uint32_t)n, j);
                const WDom<W> Dj = wbroadcast<W>(D, j);
                const unsigned q = u / nj;
                const int bitpos = Dj.kth((int)(u - q * nj));
                u = q;
                if (use_bitmap)
                    lane_part += bitpos * (int)rdlane((uint32_t)mystride, j);
                else
                    lds_vals[slot * 64 + lane] = (int)rdlane((uint32_t)vlb, j) + bitpos;
            }
            int res;
            if (use_bitmap) {
                const int bit = base_sum + lane_part;
                res = active ? (int)((((uint32_t)G.vc(c.o.tables + C.bitmap_off + (bit >> 5))) >> (bit & 31)) & 1u) : 0;
            } else {
                res = eval_program<L>(c, G, C.code_off, C.code_len, C.uses_valid != 0, lane, varinfo, curval, lds_vals, lds_stk);
            }
            n_evals += min(Plow - trip * 64u, 64u);
            if (__ballot(active && res != 0)) return true;
        }
        if (!highmask) return false;
        bool carry = true;  // advance the odometer
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
                curbit = D.kth(dj);
                curval = vlb + curbit;
            }
        }
        if (carry) return false;
    }
}

// One point constraint at one time point, bounds consistency for every scope variable (enforcePointConsistencyAt,
// src/solveralgorithm.cpp:476-523). Returns false on a wipe-out; `changedm`: scope positions whose domain shrank.
template <int W, int L>
__device__ bool revise_bounds_wide(const Ctx &c, const Img<L> &G, const ConDesc &C, int p, int lane, int *ldom, int *lds_vals, int *lds_stk,
                                   unsigned long long &changedm, unsigned long long &n_evals, unsigned &n_skipped) {
    const int s = C.scope_len, NK1 = c.N * c.K;
    int var = 0;
    if (lane < s) var = G.v(c.o.scope + C.scope_off + lane);
    const int word = p * c.N + var;
    WDom<W> D = wload<W>(ldom, NK1, word);
    if (lane >= s)
        for (int k = 0; k < W; k++) D.w[k] = k == 0 ? 1u : 0u;
    if (__ballot(lane < s && D.empty())) return false;
    const int vlb = lane < s ? G.v(c.o.var_lb + var) : 0;
    const int mystride = (C.bitmap_off >= 0 && lane < s) ? G.v(c.o.strides + C.stride_off + lane) : 0;
    const WDom<W> D_in = D;
    bool gave_up = false;
    const unsigned long long openm = __ballot(lane < s && D.count() > 1);
    if (!openm) {  // every variable fixed: the single tuple is checked (this is what makes leaves exact)
        const bool ok = exists_support_wide<W, L>(c, G, C, lane, D, vlb, mystride, lds_vals, lds_stk, n_evals, gave_up);
        changedm = 0;
        return ok;
    }
    for (unsigned long long m = openm; m; m &= m - 1) {
        const int j0 = __ffsll((long long)m) - 1;
        // least supported value, then greatest (values in between keep their place: bounds only, like the reference)
        for (int side = 0; side < 2; side++) {
            for (;;) {
                const WDom<W> Dj = wbroadcast<W>(D, j0);
                if (Dj.empty()) return false;
                const int lo = Dj.lowest(), hi = Dj.highest();
                if (side == 1 && lo == hi) break;  // the least value is supported already
                const int b = side == 0 ? lo : hi;
                WDom<W> T = D;
                if (lane == j0) T.only(b);
                gave_up = false;
                const bool sup = exists_support_wide<W, L>(c, G, C, lane, T, vlb, mystride, lds_vals, lds_stk, n_evals, gave_up);
                if (gave_up) n_skipped++;
                if (sup) break;
                if (lane == j0) D.clear(b);
            }
        }
    }
    const bool ch = lane < s && (D != D_in);
    changedm = __ballot(ch);
    if (ch) {
#pragma unroll
        for (int k = 0; k < W; k++) ldom[k * NK1 + word] = (int)D.w[k];
    }
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// X == next Y at point p (enforceNextConsistency, src/solveralgorithm.cpp:544-593): X[p] and Y[p+1] hold the same value;
// bit i of X is value lbX + i = bit i + sh of Y (sh = lbX - lbY). One lane per value, two ballots per 64 values.
template <int W>
__device__ bool revise_next_wide(const Ctx &c, int wx, int wy, int sh, int lane, int *ldom, bool &chx, bool &chy) {
    const int NK1 = c.N * c.K;
    constexpr int B = 32 * W;
    uint32_t nx[W], ny[W];
    uint32_t anyx = 0;
    bool cx = false, cy = false;
#pragma unroll
    for (int pass = 0; pass < (B + 63) / 64; pass++) {
        const int i = pass * 64 + lane, k = i + sh;
        const bool xb = i < B && (((uint32_t)ldom[(i >> 5) * NK1 + wx] >> (i & 31)) & 1u);
        const bool yb = k >= 0 && k < B && (((uint32_t)ldom[((k >= 0 && k < B ? k : 0) >> 5) * NK1 + wy] >> (k & 31)) & 1u);
        const unsigned long long keep = __ballot(xb && yb);
        nx[2 * pass] = (uint32_t)keep;
        if (2 * pass + 1 < W) nx[2 * pass + 1] = (uint32_t)(keep >> 32);
        anyx |= (uint32_t)keep | (uint32_t)(keep >> 32);
    }
    if (!anyx) return false;
    // Y keeps exactly the images of what X kept
#pragma unroll
    for (int pass = 0; pass < (B + 63) / 64; pass++) {
        const int k = pass * 64 + lane, i = k - sh;
        bool xb = false;
        if (k < B && i >= 0 && i < B) {
            uint32_t wv = 0;
#pragma unroll
            for (int q = 0; q < W; q++)
                if ((i >> 5) == q) wv = nx[q];
            xb = (wv >> (i & 31)) & 1u;
        }
        const unsigned long long keep = __ballot(xb);
        ny[2 * pass] = (uint32_t)keep;
        if (2 * pass + 1 < W) ny[2 * pass + 1] = (uint32_t)(keep >> 32);
    }
#pragma unroll
    for (int q = 0; q < W; q++) {
        const uint32_t ox = (uint32_t)ldom[q * NK1 + wx], oy = (uint32_t)ldom[q * NK1 + wy];
        cx = cx || ox != nx[q];
        cy = cy || oy != ny[q];
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < W; q++) {
            ldom[q * NK1 + wx] = (int)nx[q];
            ldom[q * NK1 + wy] = (int)ny[q];
        }
    }
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    chx = cx;
    chy = cy;
    return true;
}

struct BranchOutWide {
    int bvar, mid;  // children: values at bit positions <= mid / > mid of variable bvar at point 0
};

template <int DR, int W, int L>
__device__ int process_node_wide(const Ctx &c, const Img<L> &P, int lane, int *lds_vals, int *lds_stk, int *ldom, Dom<DR> &dom, const NodeHdr &hd,
                                 int gw, WaveEnv<DR> &S, BranchOutWide &bo, LeafOut<DR> &lo) {
    const int set = hd.set;
    const uint32_t seed = hd.seed, expire = hd.expire;
    const int NK1 = c.N * c.K;
    if (set != S.set) load_env<DR, L>(c, P, set, lane, S);
    uint32_t dirtyw = 0;
    if (lane < S.iw) {
        if (seed == 0) {
            const int left = S.nitems - lane * 32;
            dirtyw = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
        } else if (seed != kSeedNone) {
            dirtyw = (uint32_t)P.v(S.rows_abs + (int)(seed - 1) * S.iw + lane);  // word (0, seed variable)
        }
    }
#pragma unroll
    for (int q = 0; q < DR; q++) {
        const int idx = q * 64 + lane;
        if (idx < c.NK) ldom[idx] = (int)dom.r[q];
    }
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool consistent = true;
    unsigned guard = 0;
    auto mark_word = [&](int word) {  // the items that read block word (p, v)
        if (lane < S.iw) dirtyw |= (uint32_t)P.v(S.rows_abs + word * S.iw + lane);
    };
    while (consistent) {
        const unsigned long long dm = __ballot(dirtyw != 0);
        if (!dm) break;
        const int wl = __ffsll((long long)dm) - 1;
        const uint32_t word = rdlane(dirtyw, wl);
        const int b = __ffs((int)word) - 1;
        const int item = wl * 32 + b;
        if (lane == wl) dirtyw &= ~(1u << b);
        const int ibase = S.items_abs + item * (int)(sizeof(ItemDesc) / 4);
        const uint32_t irec = lane < (int)(sizeof(ItemDesc) / 4) ? (uint32_t)P.v(ibase + lane) : 0u;
#define STCSP_ID(f) (int) rdlane(irec, (int)(offsetof(ItemDesc, f) / 4))
        const int itype = STCSP_ID(type), ipoint = STCSP_ID(point);
        if (itype == IT_NEXT) {
            bool chx = false, chy = false;
            const int wx = STCSP_ID(idx[0]), wy = STCSP_ID(idx[1]);
            consistent = revise_next_wide<W>(c, wx, wy, STCSP_ID(aux), lane, ldom, chx, chy);
            S.n_revs++;
            if (consistent) {
                if (chx) mark_word(wx);
                if (chy) mark_word(wy);
                if (lane == wl) dirtyw &= ~(1u << b);  // the arc itself is at its fixpoint
            }
        } else if (itype == IT_UNTIL) {
            // X until Y (enforceUntilConsistency, :598-614): a check at point 0, never a pruning
            const int x = STCSP_ID(idx[0]), y = STCSP_ID(idx[1]), ord = STCSP_ID(aux);
            S.n_revs++;
            if (!((expire >> ord) & 1u)) {
                const WDom<W> DX = wload<W>(ldom, NK1, x), DY = wload<W>(ldom, NK1, y);
                if (DX.count() == 1 && DY.count() == 1) {
                    const int vx = P.u(c.o.var_lb + x) + DX.lowest(), vy = P.u(c.o.var_lb + y) + DY.lowest();
                    if (rfl(vx) != 1 && rfl(vy) != 1) consistent = false;
                }
            }
        } else {
            ConDesc C;
            C.scope_len = STCSP_ID(arity);
            C.scope_off = STCSP_ID(idx[0]);
            C.bitmap_off = STCSP_ID(idx[1]);
            C.stride_off = STCSP_ID(idx[2]);
            C.n_forbidden = STCSP_ID(idx[3]);
            C.code_off = STCSP_ID(toff);
            C.uses_valid = STCSP_ID(r1);
            C.code_len = STCSP_ID(r2);
            unsigned long long changedm = 0;
            consistent = revise_bounds_wide<W, L>(c, P, C, ipoint, lane, ldom, lds_vals, lds_stk, changedm, S.n_evals, S.n_skipped);
            S.n_revs++;
            S.n_wave_revs++;
            for (unsigned long long m = changedm; consistent && m; m &= m - 1) {
                const int j = __ffsll((long long)m) - 1;
                mark_word(ipoint * c.N + P.u(c.o.scope + C.scope_off + j));  // (re-dirties this item too: bounds of the others may follow)
            }
        }
#undef STCSP_ID
        if (++guard > (1u << 20)) {
            S.err = max(S.err, (unsigned)ERR_WATCHDOG);
            consistent = false;
        }
    }
    S.n_nodes++;
    // the registers follow the LDS copy (the callers store / bisect the block from them)
#pragma unroll
    for (int q = 0; q < DR; q++) {
        const int idx = q * 64 + lane;
        dom.r[q] = idx < c.NK ? (uint32_t)ldom[idx] : 0u;
    }
    if (!consistent) {
        S.n_fails++;
        return OC_FAIL;
    }
    // ---- classify (solverGetFirstUnboundVar, src/solver.cpp:41-53)
    int bvar = -1;
    for (int v0 = 0; v0 < c.N && bvar < 0; v0 += 64) {
        const int v = v0 + lane;
        const bool open = v < c.N && wload<W>(ldom, NK1, v < c.N ? v : 0).count() > 1;
        const unsigned long long m = __ballot(open);
        if (m) bvar = v0 + __ffsll((long long)m) - 1;
    }
    if (bvar >= 0) {
        // variableSplitLower/Upper (variable.cpp:52-67): [lb, lb + (ub - lb) / 2] and the rest
        const WDom<W> D = wload<W>(ldom, NK1, bvar);
        const int lo_ = rfl(D.lowest()), hi_ = rfl(D.highest());
        bo.bvar = bvar;
        bo.mid = lo_ + (hi_ - lo_) / 2;
        return OC_BRANCH;
    }
    // ---- leaf (solveralgorithm.cpp:739-910): every variable has one time-0 value
    auto value_of = [&](int v) -> int {  // per lane: time-0 value of variable v
        return P.v(c.o.var_lb + v) + wload<W>(ldom, NK1, v).lowest();
    };
    int next_set = set;
    if (!S.self_loop) {
        const int fv = lane < S.nfirst ? P.v(c.o.firstvars + S.first_off + lane) : 0;
        const int fbit = wload<W>(ldom, NK1, fv).lowest();
        const int fval = lane < S.nfirst ? P.v(c.o.var_lb + fv) + fbit : 0;
        next_set = -1;
        if (S.trans_count < 0) {
            const int fs = lane < S.nfirst ? P.v(c.o.fstrides + S.first_off + lane) : 0;
            const int idx = wave_sum(lane < S.nfirst ? fbit * fs : 0);
            next_set = kload(c.tdirect, S.trans_begin + idx);
        }
        for (int t = 0; t < S.trans_count && next_set < 0; t++) {
            const int voff = P.u(c.o.trans + (S.trans_begin + t) * 2);
            const int tv = P.v(c.o.transvals + voff + (lane < S.nfirst ? lane : 0));
            const bool ne = lane < S.nfirst && tv != fval;
            if (!__ballot(ne)) next_set = P.u(c.o.trans + (S.trans_begin + t) * 2 + 1);
        }
        if (next_set < 0) {  // unknown transition: the host translates (see process_node)
            uint32_t mi = 0;
            if (lane == 0) mi = atomicAdd(&c.ctl[CtlLayout(c.world).misc0 + MISC_NMISS * CST], 1u);
            mi = rflu(mi);
            if ((int)mi < c.miss_cap) {
                int *rec = c.miss + (size_t)mi * kMissStride;
                if (lane == 0) {
                    rec[0] = set;
                    rec[1] = S.nfirst;
                }
                if (lane < S.nfirst) rec[2 + lane] = fval;
            }
            STCSP_REJOIN();
            S.n_requeue++;
            return OC_MISS;
        }
    }
    const uint32_t next_tag = (uint32_t)P.u(c.o.sets + next_set * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4));
    uint32_t new_expire = expire;
    uint32_t kw = 0;
    {
        const bool sl = lane >= 1 && lane <= c.n_sig;
        const int sv = sl ? P.v(c.o.sig_vars + lane - 1) : 0;
        const int val = value_of(sv);
        if (sl) kw = (uint32_t)val;
        for (int u = 0; u < c.n_until_cons; u++) {
            const int y = P.u(c.o.until_y + u);
            const int vy = rfl(P.u(c.o.var_lb + y) + wload<W>(ldom, NK1, y).lowest());
            bool ex = (expire >> u) & 1u;
            if (!ex && vy == 1) {
                ex = true;
                new_expire |= 1u << u;
            }
            if (lane == 1 + c.n_sig + u) kw = ex ? 1u : 0u;
        }
        if (lane == 0) kw = next_tag;
    }
    unsigned long long h;
    {
        const unsigned long long t = lane < c.KL ? key_term(lane, kw) : 0ull;
        h = mix_final(kHashSeed ^ wave_xor64(t));
    }
    lo.kw = kw;
    lo.h = h;
    lo.next_set = next_set;
    lo.next_tag = next_tag;
    lo.new_expire = new_expire;
    lo.owner = key_owner(h, c.world, c.KL, next_tag);
#pragma unroll
    for (int q = 0; q < DR; q++) {
        const int idx = q * 64 + lane;
        lo.evals[q] = idx < c.N ? (uint32_t)value_of(idx) : 0u;
        uint32_t nb = 0;
        if (idx < c.NK) {  // variableAdvanceOneTimeStep (variable.cpp:94-108), chunk by chunk
            const int ch = idx / NK1, rem = idx - ch * NK1;
            const int p = rem / c.N, v = rem - p * c.N;
            nb = (p + 1 < c.K) ? (uint32_t)ldom[ch * NK1 + (p + 1) * c.N + v] : (uint32_t)P.v(c.o.var_init + ch * c.N + v);
        }
        lo.nblk[q] = nb;
    }
    S.n_leaves++;
    return OC_LEAF;
}

}  // namespace dev
}  // namespace stcsp
