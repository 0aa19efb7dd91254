// dev_rows.hpp -- k_expand_rows: up to FOUR open search nodes per wavefront, one per DPP row of 16 lanes: a wavefront
// takes ONE slot of the round and walks that node's SUBTREE with its four rows.
//
// Why: under the one-node-per-wavefront kernel (dev_kernels.hpp k_expand) a partialorder_14 node keeps ~14 of 64 lanes
// busy in a sweep and the rest of the node (classification, signature, hash, table probe, record writes) is wave-uniform
// scalar work: ~1,250 instructions per 0.6 KB node, the kernel is bound by instruction issue (profiles/r02_p14_pmc.json).
// Programs whose node fits a quarter wavefront run here instead: every 16-lane row owns its own node -- block, dirty
// mask, sibling stack, chain -- and runs the same code; what used to be wave-uniform (SGPR) state is row-uniform
// (the 16 lanes of a row hold equal values), control flow diverges between rows through the exec mask, cross-lane traffic
// stays inside a row: DPP row rotations for reductions, ds_swizzle / ds_bpermute for broadcasts, 16-bit slices of
// __ballot for votes. One instruction stream now serves up to four nodes.
//
// Which four: the nodes a wavefront holds are always from ONE subtree -- the slot's node, then its two children, then four
// grandchildren ... (upper children wait on a stack in LDS that all four rows push to and pop from at the end of every
// step). Sister rows therefore run the same phases at the same time (same constraint set, same depth, same kind of
// node), which is what makes the sharing pay: four UNRELATED nodes per wavefront (the first version of this kernel: one
// slot per row) execute the union of all code paths in every step and came out at 1,077 instructions per node on
// partialorder_14 against 1,246 for one node per wavefront -- no gain. A slot's wavefront expands up to 1, 2, 4, 4 ...
// nodes in its steps (rows_processed), so a launch also advances a subtree by 11 nodes where the chained kernel does 4:
// fewer rounds for the parts of a search that are narrower than the machine.
//
// Eligible programs (engine.hip upload_program): LITE (every wavefront-revised constraint is a tuple bitmap with at most
// one violating tuple: revise_point's two shortcut exits are all that is needed), N*K <= 16*DQ block words,
// 1 + signature length <= 16*KT key words, at most 4 dirty-mask words per set (<= 128 work items).
//
// Layout of a row: block word w lives in register d[w >> 4] of lane (w & 15) of the row; its AND-accumulator copy in
// LDS (ldom) is what per-lane code reads domains from (one ds_read where a register gather would be DQ bpermutes);
// dirty word j in lane j; key word j in kw[j >> 4] of lane (j & 15).
//
// Same frontier format, same device plan, same finalize_round as k_expand: the host side does not know the difference.
// Reference roles are those of dev_propagate.hpp / dev_kernels.hpp (solverSolveRe, src/solveralgorithm.cpp:733-942;
// generalisedArcConsistent :617-706; enforcePointConsistencyAt :476-523; enforceNextConsistency :544-593).
#pragma once
#include "dev_kernels.hpp"
namespace stcsp {
namespace dev {

constexpr int kRowStack = 16;     // node records on a wavefront's shared stack (rows_alive(6) - 4 = 16)
constexpr int kRowStatWords = 16; // per-wavefront counters in LDS: [ST_x] as in flush_env, [15] sticky error code
constexpr int kRowMaxIW = 4;      // dirty-mask words per set

// LDS words of one wavefront of k_expand_rows: counters + 4 rows x block copy + the shared stack
__host__ __device__ inline int rows_ldom_words(int DQ) { return DQ * 16; }
__host__ __device__ inline int rows_wave_words(int DQ, int NS) { return kRowStatWords + 4 * rows_ldom_words(DQ) + kRowStack * NS; }

// lane (rbase | j) of v, j row-uniform per lane
__device__ __forceinline__ uint32_t row_get(uint32_t v, int rbase, int j) { return (uint32_t)__builtin_amdgcn_ds_bpermute((rbase | j) << 2, (int)v); }
// ... for a compile-time j: ds_swizzle in bit-mask mode (lane' = (lane & 0x10) | J inside each group of 32), no address register
template <int J>
__device__ __forceinline__ uint32_t row_get_c(uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x10 | (J << 5)); }
// the 16 vote bits of this lane's row
__device__ __forceinline__ uint32_t row_ballot(bool p, int rbase) { return (uint32_t)(__ballot(p) >> rbase) & 0xffffu; }
__device__ __forceinline__ bool row_any(bool p, int rbase) { return row_ballot(p, rbase) != 0u; }
__device__ __forceinline__ uint32_t row_or(uint32_t v) {
    v |= (uint32_t)row_ror((int)v, 8);
    v |= (uint32_t)row_ror((int)v, 4);
    v |= (uint32_t)row_ror((int)v, 2);
    v |= (uint32_t)row_ror((int)v, 1);
    return v;
}
__device__ __forceinline__ uint32_t row_xor(uint32_t v) {
    v ^= (uint32_t)row_ror((int)v, 8);
    v ^= (uint32_t)row_ror((int)v, 4);
    v ^= (uint32_t)row_ror((int)v, 2);
    v ^= (uint32_t)row_ror((int)v, 1);
    return v;
}
// inclusive prefix sum inside the row (row_shr shifts zeros in at the row's start)
__device__ __forceinline__ int row_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return v;
}

// the constraint set a row works under (row-uniform unless marked per lane); reloaded when the row's set changes
template <int DQ>
struct RowEnv {
    int set = -1;
    int self_loop = 0, nfirst = 0, first_off = 0, trans_begin = 0, trans_count = 0, nitems = 0, nsmall = 0, iw = 1;
    int rows_abs = 0, sweep_abs = 0, items_abs = 0, next_abs = -1;
    uint32_t smallmask = 0;  // per lane: bits of the lane-revised items in dirty word lane16
    uint32_t e0[DQ] = {};    // per lane: eager-arc partner entry 0 of block word q*16 + lane16
};

template <int DQ, bool L>
__device__ __forceinline__ void load_row_env(const Ctx &c, const Img<L> &P, int set, int lane16, RowEnv<DQ> &E) {
    constexpr int W = (int)(sizeof(SetDesc) / 4);
    const int sb = c.o.sets + set * W;
#define STCSP_SD(f) P.v(sb + (int)(offsetof(SetDesc, f) / 4))
    E.set = set;
    E.self_loop = STCSP_SD(self_loop);
    E.nfirst = STCSP_SD(nfirst);
    E.first_off = STCSP_SD(first_off);
    E.trans_begin = STCSP_SD(trans_begin);
    E.trans_count = STCSP_SD(trans_count);
    E.nitems = STCSP_SD(nitems);
    E.nsmall = STCSP_SD(nsmall);
    E.iw = STCSP_SD(iw);
    const int item_begin = STCSP_SD(item_begin), next_off = STCSP_SD(next_off);
    E.rows_abs = c.o.itemrows + STCSP_SD(itemrows_off);
    E.sweep_abs = c.o.sweep + item_begin * 4;
    E.items_abs = c.o.items + (STCSP_SD(witem_begin) - E.nsmall) * (int)(sizeof(ItemDesc) / 4);
    E.next_abs = next_off >= 0 ? c.o.nextpart + next_off : -1;
#undef STCSP_SD
    const int left = E.nsmall - lane16 * 32;
    E.smallmask = (lane16 < E.iw && left > 0) ? (left >= 32 ? 0xffffffffu : ((1u << left) - 1u)) : 0u;
#pragma unroll
    for (int q = 0; q < DQ; q++) {
        const int idx = q * 16 + lane16;
        E.e0[q] = (E.next_abs >= 0 && idx < c.NK) ? (uint32_t)P.v(E.next_abs + idx * 2) : 0u;
    }
}

// OR the dirty rows of this lane's changed block words into the row's lane-striped dirty mask: every lane fetches the
// rows of ITS changed words (iw <= kRowMaxIW words each), a DPP OR-reduction spreads them over the row, lane j keeps word j
template <int DQ, bool L>
__device__ __forceinline__ void row_mark_dirty(const Img<L> &P, const RowEnv<DQ> &E, int lane16, const bool (&chg)[DQ], int maxiw, uint32_t &dirty) {
    uint32_t acc[kRowMaxIW] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < DQ; q++)
        if (chg[q]) {
            const int rb = E.rows_abs + (q * 16 + lane16) * E.iw;
#pragma unroll
            for (int j = 0; j < kRowMaxIW; j++)
                if (j < E.iw) acc[j] |= (uint32_t)P.v(rb + j);
        }
#pragma unroll
    for (int j = 0; j < kRowMaxIW; j++)
        if (j < maxiw) {  // maxiw: wave-uniform bound over the program's sets
            const uint32_t all = row_or(acc[j]);
            if (lane16 == j) dirty |= all;
        }
}

template <int DQ, int KT, bool L>
__global__ __launch_bounds__(256, 4) void k_expand_rows(const Ctx *__restrict__ cp) {
    const Ctx &c = *cp;
    extern __shared__ __attribute__((aligned(16))) int smem[];
    if (kload(c.plan, (int)(offsetof(Plan, status) / 4)) != PS_RUN) return;  // the burst ran past the end
    const int lane = threadIdx.x & 63, wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane16 = lane & 15, rbase = lane & 48, row = lane >> 4;
    const int n_slots = kload(c.plan, (int)(offsetof(Plan, nslots) / 4));
    const int total_waves = gridDim.x * 4;
    // workgroups without a slot leave at once; the ticket below counts the working ones only
    if ((int)blockIdx.x * 4 >= n_slots) return;
    const unsigned n_working = (unsigned)min((n_slots + 3) / 4, (int)gridDim.x);
    const int img_words = (c.stage_words + 3) & ~3;
    if (img_words) {
        const uint4 *src = (const uint4 *)c.img;
        uint4 *dst = (uint4 *)smem;
        for (int k = threadIdx.x; k < img_words / 4; k += 256) dst[k] = src[k];
        __syncthreads();
    }
    const int N = c.N, K = c.K, NK = c.NK, NS = c.NS, KL = c.KL;
    const int per_wave = rows_wave_words(DQ, NS);
    const int wave_off = img_words + wib * per_wave;  // word offset in the launch's LDS
    int *wstat = smem + wave_off;                      // counters of this wavefront
    if (lane < kRowStatWords) wstat[lane] = 0;
    int *ldom = smem + wave_off + kRowStatWords + row * rows_ldom_words(DQ);
    const int stk_off = wave_off + kRowStatWords + 4 * rows_ldom_words(DQ);  // the wavefront's stack of node records
    Img<L> P{c.img, (const uint32_t *)smem, c.stage_words};
    const CtlLayout L_(c.world);
    // round arguments (wave-uniform)
    auto pl = [&](size_t off) { return (uint32_t)kload(c.plan, (int)(off / 4)); };
    auto pl64 = [&](size_t off) { return (unsigned long long)pl(off) | (unsigned long long)pl(off + 4) << 32; };
    const uint32_t *in_base = c.arena + pl64(offsetof(Plan, in_base));
    const uint32_t in_cap = pl(offsetof(Plan, in_cap));
    uint32_t *out_base = c.arena + pl64(offsetof(Plan, out_base));
    const uint32_t out_cap = pl(offsetof(Plan, out_cap));
    const uint32_t cand_cap = pl(offsetof(Plan, cand_cap));
    const int parity = (int)pl(offsetof(Plan, parity));
    const int chain = min((int)pl(offsetof(Plan, chain)), kRowMaxChain);
    const unsigned long long chain_cycles = (unsigned long long)pl(offsetof(Plan, chain_heavy));
    const int maxiw = c.max_iw;  // wave-uniform bound of the sets' dirty-mask widths

    auto stat = [&](int which, unsigned v) {
        if (lane16 == 0 && v) atomicAdd((unsigned *)&wstat[which], v);
    };
    auto raise = [&](unsigned code) {
        if (lane16 == 0) atomicMax((unsigned *)&wstat[15], code);
    };
    auto store_row_node = [&](uint32_t *dst, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, const uint32_t (&blk)[DQ]) {
        if (lane16 < 4) dst[lane16] = lane16 == 0 ? a0 : (lane16 == 1 ? a1 : (lane16 == 2 ? a2 : a3));
#pragma unroll
        for (int q = 0; q < DQ; q++) {
            const int idx = q * 16 + lane16;
            if (idx < NK) dst[4 + idx] = blk[q];
        }
    };
    // a 4-bit mask of the rows for which `p` (row-uniform) holds, and this row's rank among them
    auto rows_of = [&](bool p) -> uint32_t {
        const unsigned long long b = __ballot(p);
        return (uint32_t)(b & 1ull) | (uint32_t)((b >> 15) & 2ull) | (uint32_t)((b >> 30) & 4ull) | (uint32_t)((b >> 45) & 8ull);
    };
    const uint32_t below = (1u << row) - 1u;

    // ---- row state
    RowEnv<DQ> E;
    uint32_t d[DQ];
#pragma unroll
    for (int q = 0; q < DQ; q++) d[q] = 0u;
    uint32_t h0 = 0, h1 = 0, seed = 0, expire = 0;  // node header (row-uniform)
    int set = 0;
    const int wave_g = (int)blockIdx.x * 4 + wib;

    for (int gw = wave_g; gw < n_slots; gw += total_waves) {
        // ---- slot gw = (region r = gw % R, i = gw / R), valid when i < take[r]; outputs go to another cursor shard than the
        // input's (a subtree would stay in the region of its root forever otherwise)
        const int r = gw % R, i = gw / R;
        if (i >= kload(c.plan, (int)(offsetof(Plan, take) / 4) + r)) continue;
        const int count_r = kload(c.plan, (int)(offsetof(Plan, count) / 4) + r);
        const int ro = (i + r) % R;
        uint32_t *const out_region = out_base + (size_t)ro * out_cap * NS;
        bool have = row == 0;  // the slot's node starts in row 0; the other rows join as the subtree widens
        if (have) {
            const uint32_t *node = in_base + ((size_t)r * in_cap + (size_t)(count_r - 1 - i)) * NS;
            const uint32_t hw = lane16 < 4 ? node[lane16] : 0u;
#pragma unroll
            for (int q = 0; q < DQ; q++) {
                const int idx = q * 16 + lane16;
                d[q] = idx < NK ? node[4 + idx] : 0u;
            }
            h0 = row_get_c<0>(hw);
            h1 = row_get_c<1>(hw);
            const uint32_t w2 = row_get_c<2>(hw);
            expire = row_get_c<3>(hw);
            set = (int)(w2 & 0xffffu);
            seed = w2 >> 16;
        }
        int sp = 0;  // records on the wavefront's stack (wave-uniform)
        const unsigned long long t_slot = __builtin_amdgcn_s_memtime();
        for (int step = 1;; step++) {
            bool cont = false, extra = false;  // after its node this row holds: the next node of its chain (d, header) / an upper child (child)
            uint32_t child[DQ];
#pragma unroll
            for (int q = 0; q < DQ; q++) child[q] = 0u;
            if (have) {
                // =========================================================== one node (process_node of the row)
                if (set != E.set) load_row_env<DQ, L>(c, P, set, lane16, E);
                uint32_t dirty = 0;
                if (lane16 < E.iw) {
                    if (seed == 0) {
                        const int left = E.nitems - lane16 * 32;
                        dirty = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
                    } else if (seed != 0xffffu) {
                        dirty = (uint32_t)P.v(E.rows_abs + (int)(seed - 1) * E.iw + lane16);
                    }
                }
#pragma unroll
                for (int q = 0; q < DQ; q++) {
                    const int idx = q * 16 + lane16;
                    if (idx < NK) ldom[idx] = (int)d[q];
                }
                bool need_close = false;
                if (E.next_abs >= 0) {
                    if (seed == 0)
                        need_close = true;  // fresh state: the time shift may have broken any arc
                    else if (seed != 0xffffu)
                        need_close = P.v(E.next_abs + (int)(seed - 1) * 2) != 0;  // the bisected word has an eager partner
                }
                bool consistent = true;
                unsigned guard = 0;
                while (consistent) {
                    if (need_close) {
                        // ---- eager X == next Y arcs: every word intersects itself with its partner's (shifted) domain
                        need_close = false;
                        for (int pass = 0; pass < 64 && consistent; pass++) {
                            bool chg[DQ];
                            bool any = false, wipe = false;
#pragma unroll
                            for (int q = 0; q < DQ; q++) {
                                const int idx = q * 16 + lane16;
                                const bool in = idx < NK;
                                auto allowed = [&](uint32_t e) -> uint32_t {
                                    if (!e) return 0xffffffffu;
                                    const uint32_t partner = (uint32_t)ldom[(int)(e & 0xffffu) - 1];
                                    const int sh = (int)((e >> 16) & 0xffu) - 64;
                                    const bool up = ((e >> 24) & 1u) ? sh >= 0 : sh < 0;
                                    const int a = sh >= 0 ? sh : -sh;
                                    return a >= 32 ? 0u : (up ? partner << a : partner >> a);
                                };
                                uint32_t nd = d[q] & allowed(E.e0[q]);
                                if (K > 2 && in) nd &= allowed((uint32_t)P.v(E.next_abs + idx * 2 + 1));
                                chg[q] = in && nd != d[q];
                                wipe = wipe || (in && nd == 0u);
                                any = any || chg[q];
                                d[q] = nd;
                            }
                            // (all partners were read before any word of this pass is written back)
#pragma unroll
                            for (int q = 0; q < DQ; q++)
                                if (chg[q]) ldom[q * 16 + lane16] = (int)d[q];
                            if (row_any(wipe, rbase)) consistent = false;
                            const bool changed = row_any(any, rbase);
                            if (changed) row_mark_dirty<DQ, L>(P, E, lane16, chg, maxiw, dirty);
                            if (!changed || K == 2) break;
                        }
                        if (!consistent) break;
                    }
                    const uint32_t dsm = dirty & E.smallmask;
                    if (row_any(dsm != 0u, rbase)) {
                        // ---- lane-parallel sweep over the row's dirty small items, compacted: lane k of pass t takes the
                        // (16 t + k)-th dirty item (prefix counts over the row's dirty words, 4-step search)
                        stat(ST_SWEEPS, 1u);
                        const int cnt = __popc(dsm);
                        const int incl = row_scan_add(cnt);
                        const int excl = incl - cnt;
                        const int total_dirty = (int)row_get_c<15>((uint32_t)incl);
                        bool lfail = false;
                        unsigned nev = 0;
                        for (int t = 0; t * 16 < total_dirty; t++) {
                            const int k = t * 16 + lane16;
                            int w = 0;  // the last lane of the row whose exclusive prefix is <= k owns the k-th dirty bit
#pragma unroll
                            for (int stp = 8; stp >= 1; stp >>= 1) {
                                const int cand = w + stp;
                                const int e = (int)row_get((uint32_t)excl, rbase, cand & 15);
                                if (cand < 16 && e <= k) w = cand;
                            }
                            const uint32_t word = row_get(dsm, rbase, w);
                            const int first = (int)row_get((uint32_t)excl, rbase, w);
                            const bool isd = k < total_dirty;
                            const int item = isd ? w * 32 + select_kth_fast(word, k - first) : 0;
                            const uint4 sw = P.v4(E.sweep_abs + item * 4);
                            const int i0 = (int)(sw.x & 255u), i1 = (int)((sw.x >> 8) & 255u), i2 = (int)((sw.x >> 16) & 255u), i3 = (int)(sw.x >> 24);
                            const int type = (int)(sw.y & 3u), arity = (int)((sw.y >> 2) & 7u), r1 = (int)((sw.y >> 5) & 63u), r2 = (int)((sw.y >> 11) & 63u);
                            const int aux = (int)sw.y >> 17, toff = (int)sw.z;
                            if (isd) {
                                const uint32_t D0 = (uint32_t)ldom[i0];
                                uint32_t D1 = (uint32_t)ldom[i1], D2 = (uint32_t)ldom[i2], D3 = (uint32_t)ldom[i3];
                                if (type == IT_NEXT) {
                                    const int sh = aux;
                                    const uint32_t Yal = sh >= 0 ? (sh < 32 ? D1 >> sh : 0u) : (-sh < 32 ? D1 << -sh : 0u);
                                    const uint32_t m = D0 & Yal;
                                    const uint32_t newY = sh >= 0 ? (sh < 32 ? m << sh : 0u) : (-sh < 32 ? m >> -sh : 0u);
                                    if (m == 0) lfail = true;
                                    if (m != D0) atomicAnd((unsigned *)&ldom[i0], m);
                                    if (newY != D1) atomicAnd((unsigned *)&ldom[i1], newY);
                                } else if (type == IT_UNTIL) {
                                    if (!((expire >> aux) & 1u) && __popc(D0) == 1 && __popc(D1) == 1) {
                                        const int vx = P.v(c.o.var_lb + i0) + __ffs((int)D0) - 1, vy = P.v(c.o.var_lb + i1) + __ffs((int)D1) - 1;
                                        if (vx != 1 && vy != 1) lfail = true;
                                    }
                                } else {
                                    if (arity < 2) D1 = 1u;
                                    if (arity < 3) D2 = 1u;
                                    if (arity < 4) D3 = 1u;
                                    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                                    const int tab = c.o.tables + toff;
                                    const int r1p = (r1 + 3) & ~3;
                                    for (uint32_t m3 = D3; m3; m3 &= m3 - 1) {
                                        const int b3 = __ffs((int)m3) - 1;
                                        for (uint32_t m2 = D2; m2; m2 &= m2 - 1) {
                                            const int b2 = __ffs((int)m2) - 1;
                                            const int base = tab + r1p * (b2 + r2 * b3);
                                            uint32_t any = 0;
                                            for (int c4 = 0; c4 < r1p; c4 += 4) {
                                                const uint32_t nib = (D1 >> c4) & 15u;
                                                if (!nib) continue;
                                                const uint4 rr = P.v4c(base + c4);
                                                const uint32_t rows4[4] = {rr.x, rr.y, rr.z, rr.w};
                                                uint32_t got = 0;
#pragma unroll
                                                for (int kk = 0; kk < 4; kk++) {
                                                    const uint32_t rv = ((nib >> kk) & 1u) ? (rows4[kk] & D0) : 0u;
                                                    s0 |= rv;
                                                    got |= (rv ? 1u : 0u) << kk;
                                                }
                                                nev += (unsigned)__popc(nib);
                                                s1 |= got << c4;
                                                any |= got;
                                            }
                                            if (any) {
                                                s2 |= 1u << b2;
                                                s3 |= 1u << b3;
                                            }
                                        }
                                    }
                                    if (s0 == 0) lfail = true;
                                    if (s0 != D0) atomicAnd((unsigned *)&ldom[i0], s0);
                                    if (arity > 1 && s1 != D1) atomicAnd((unsigned *)&ldom[i1], s1);
                                    if (arity > 2 && s2 != D2) atomicAnd((unsigned *)&ldom[i2], s2);
                                    if (arity > 3 && s3 != D3) atomicAnd((unsigned *)&ldom[i3], s3);
                                }
                            }
                        }
                        stat(ST_REVS, (unsigned)total_dirty);
                        if (nev) atomicAdd((unsigned *)&wstat[ST_EVALS], nev);
                        dirty &= ~E.smallmask;
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        if (row_any(lfail, rbase)) {
                            consistent = false;
                            break;
                        }
                        // read the intersection back; every changed word re-dirties the items that read it
                        bool chg[DQ];
                        bool any = false, wipe = false, part = false;
#pragma unroll
                        for (int q = 0; q < DQ; q++) {
                            const int idx = q * 16 + lane16;
                            const uint32_t nd = idx < NK ? (uint32_t)ldom[idx] : d[q];
                            wipe = wipe || (idx < NK && nd == 0u);
                            chg[q] = nd != d[q];
                            any = any || chg[q];
                            part = part || (chg[q] && E.e0[q] != 0u);
                            d[q] = nd;
                        }
                        if (row_any(wipe, rbase)) consistent = false;
                        if (row_any(any, rbase)) row_mark_dirty<DQ, L>(P, E, lane16, chg, maxiw, dirty);
                        need_close = row_any(part, rbase);  // a word with an eager partner changed
                        if (++guard > (1u << 20)) {
                            raise(ERR_WATCHDOG);
                            consistent = false;
                        }
                        continue;
                    }
                    // ---- a dirty wavefront-revised item: tuple bitmap with at most one violating tuple (LITE programs)
                    const uint32_t dmask = row_ballot(dirty != 0u, rbase);
                    if (!dmask) break;
                    const int wl = __ffs((int)dmask) - 1;
                    const uint32_t word = row_get(dirty, rbase, wl);
                    const int b = __ffs((int)word) - 1;
                    const int item = wl * 32 + b;
                    if (lane16 == wl) dirty &= ~(1u << b);
                    const int ib = E.items_abs + item * (int)(sizeof(ItemDesc) / 4);
                    const int ipoint = P.v(ib + (int)(offsetof(ItemDesc, point) / 4));
                    const int s = P.v(ib + (int)(offsetof(ItemDesc, arity) / 4));
                    const int scope_off = P.v(ib + (int)(offsetof(ItemDesc, idx) / 4));
                    const int bitmap_off = P.v(ib + (int)(offsetof(ItemDesc, idx) / 4) + 1);
                    const int stride_off = P.v(ib + (int)(offsetof(ItemDesc, idx) / 4) + 2);
                    const int n_forbidden = P.v(ib + (int)(offsetof(ItemDesc, idx) / 4) + 3);
                    // scan the scope 16 variables at a time: open variables, bitmap index of the singletons, the last open one
                    int nopen = 0, fixed_part = 0;
                    int oj_var = 0, oj_stride = 0;
                    uint32_t oj_D = 0;
                    bool wipe = false;
                    for (int base = 0; base < s; base += 16) {
                        const int j = base + lane16;
                        const bool in = j < s;
                        const int var = in ? P.v(c.o.scope + scope_off + j) : 0;
                        const uint32_t D = in ? (uint32_t)ldom[ipoint * N + var] : 1u;
                        const int mystride = in ? P.v(c.o.strides + stride_off + j) : 0;
                        const int n = __popc(D);
                        wipe = wipe || n == 0;
                        const uint32_t open = row_ballot(in && n > 1, rbase);
                        nopen += __popc(open);
                        fixed_part += row_sum((in && n == 1) ? (__ffs((int)D) - 1) * mystride : 0);
                        if (open) {
                            const int oj = 31 - __clz((int)open);  // (only used when exactly one variable is open)
                            oj_var = (int)row_get((uint32_t)var, rbase, oj);
                            oj_stride = (int)row_get((uint32_t)mystride, rbase, oj);
                            oj_D = row_get(D, rbase, oj);
                        }
                    }
                    if (row_any(wipe, rbase)) {
                        consistent = false;
                        break;
                    }
                    // a value loses its support only when the whole product of the OTHER domains is forbidden: with at most one
                    // violating tuple that needs every other variable fixed
                    if (n_forbidden == 0 || nopen >= 2 || bitmap_off < 0 || n_forbidden > 1) {
                        if (n_forbidden > 1 || bitmap_off < 0) stat(ST_SKIPPED, 1u);  // (not a LITE item: cannot happen; skipping is sound)
                        continue;
                    }
                    stat(ST_REVS, 1u);
                    stat(ST_WAVEREVS, 1u);
                    const int tab = c.o.tables + bitmap_off;
                    if (nopen == 0) {
                        stat(ST_EVALS, 1u);
                        if (!((((uint32_t)P.vc(tab + (fixed_part >> 5))) >> (fixed_part & 31)) & 1u)) {
                            consistent = false;
                            break;
                        }
                        continue;
                    }
                    uint32_t newD = 0;
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int v = half * 16 + lane16;
                        const bool has = (oj_D >> v) & 1u;
                        const int bit = fixed_part + v * oj_stride;
                        const bool sat = has && ((((uint32_t)P.vc(tab + ((has ? bit : 0) >> 5))) >> (bit & 31)) & 1u);
                        newD |= row_ballot(sat, rbase) << (16 * half);
                    }
                    stat(ST_EVALS, (unsigned)__popc(oj_D));
                    if (newD == 0u) {
                        consistent = false;
                        break;
                    }
                    if (newD != oj_D) {
                        const int w = ipoint * N + oj_var;
                        bool chg[DQ];
                        bool part = false;
#pragma unroll
                        for (int q = 0; q < DQ; q++) {
                            chg[q] = (q * 16 + lane16) == w;
                            if (chg[q]) {
                                d[q] = newD;
                                ldom[w] = (int)newD;
                                part = E.e0[q] != 0u;
                            }
                        }
                        row_mark_dirty<DQ, L>(P, E, lane16, chg, maxiw, dirty);
                        if (lane16 == wl) dirty &= ~(1u << b);  // a revision is a fixpoint for its own constraint
                        need_close = row_any(part, rbase);
                    }
                    if (++guard > (1u << 20)) {
                        raise(ERR_WATCHDOG);
                        consistent = false;
                    }
                }
                stat(ST_NODES, 1u);
                if (!consistent) {
                    stat(ST_FAILS, 1u);
                } else {
                    // ---- classify: first variable, in queue order, whose time-0 domain is not a singleton
                    int bvar = -1;
#pragma unroll
                    for (int q = DQ - 1; q >= 0; q--) {
                        const int idx = q * 16 + lane16;
                        const uint32_t m = row_ballot(idx < N && __popc(d[q]) > 1, rbase);
                        if (m) bvar = q * 16 + __ffs((int)m) - 1;
                    }
                    if (bvar >= 0) {
                        // ---- bisect (variableSplitLower/Upper): the lower half stays in this row, the upper half is handed
                        // to an idle sister row or to the stack at the end of the step
                        const uint32_t D = (uint32_t)ldom[bvar];
                        const int lo_ = __ffs((int)D) - 1, hi_ = 31 - __clz((int)D);
                        const int mid = lo_ + (hi_ - lo_) / 2;
                        const uint32_t lowmask = (mid >= 31) ? 0xffffffffu : ((2u << mid) - 1u);
#pragma unroll
                        for (int q = 0; q < DQ; q++) {
                            const bool at = (q * 16 + lane16) == bvar;
                            child[q] = at ? (D & ~lowmask) : d[q];
                            if (at) d[q] = D & lowmask;
                        }
                        seed = (uint32_t)(bvar + 1);  // (both halves: same header)
                        cont = true;
                        extra = true;
                    } else {
                        // ---- leaf: every variable has a single time-0 value (solveralgorithm.cpp:739-910)
                        // (1) next constraint set through the transition table
                        int next_set = set;
                        bool miss = false;
                        if (!E.self_loop) {
                            next_set = -1;
                            for (int t = 0; t < E.trans_count && next_set < 0; t++) {
                                const int voff = P.v(c.o.trans + (E.trans_begin + t) * 2);
                                bool ne = false;
                                for (int j = lane16; j < E.nfirst; j += 16) {
                                    const int fv = P.v(c.o.firstvars + E.first_off + j);
                                    const int fval = P.v(c.o.var_lb + fv) + __ffs(ldom[fv]) - 1;
                                    ne = ne || P.v(c.o.transvals + voff + j) != fval;
                                }
                                if (!row_any(ne, rbase)) next_set = P.v(c.o.trans + (E.trans_begin + t) * 2 + 1);
                            }
                            if (next_set < 0) {
                                // unknown transition: park the node again and tell the host which translation is needed
                                uint32_t mi = 0;
                                if (lane16 == 0) mi = atomicAdd(&c.ctl[L_.misc0 + MISC_NMISS * CST], 1u);
                                mi = row_get_c<0>(mi);
                                if ((int)mi < c.miss_cap) {
                                    int *rec = c.miss + (size_t)mi * kMissStride;
                                    if (lane16 == 0) {
                                        rec[0] = set;
                                        rec[1] = E.nfirst;
                                    }
                                    for (int j = lane16; j < E.nfirst; j += 16) {
                                        const int fv = P.v(c.o.firstvars + E.first_off + j);
                                        rec[2 + j] = P.v(c.o.var_lb + fv) + __ffs(ldom[fv]) - 1;
                                    }
                                }
                                stat(ST_REQUEUE, 1u);
                                miss = true;
                            }
                        }
                        if (miss) {
                            // (straight to the frontier, not to the stack: this launch cannot do anything more for it)
                            uint32_t pos = 0;
                            if (lane16 == 0) pos = atomicAdd(&c.ctl[L_.out(parity, ro)], 1u);
                            pos = row_get_c<0>(pos);
                            if (pos + 1u > out_cap)
                                raise(ERR_OUT_OVERFLOW);
                            else
                                store_row_node(out_region + (size_t)pos * NS, h0, h1, (uint32_t)set | 0xffff0000u, expire, d);
                        } else {
                            stat(ST_LEAVES, 1u);
                            const uint32_t next_tag = (uint32_t)P.v(c.o.sets + next_set * (int)(sizeof(SetDesc) / 4) + (int)(offsetof(SetDesc, tag) / 4));
                            // (2) signature: key word j = [next set tag, signature variables in queue order, one sticky flag per until]
                            uint32_t new_expire = expire;
                            for (int u = 0; u < c.n_until_cons; u++) {
                                const int y = P.v(c.o.until_y + u);
                                if (!((expire >> u) & 1u) && P.v(c.o.var_lb + y) + __ffs(ldom[y]) - 1 == 1) new_expire |= 1u << u;
                            }
                            uint32_t kw[KT];
                            unsigned long long hterm = 0ull;
#pragma unroll
                            for (int t = 0; t < KT; t++) {
                                const int j = t * 16 + lane16;
                                uint32_t w = 0;
                                if (j == 0)
                                    w = next_tag;
                                else if (j <= c.n_sig) {
                                    const int sv = P.v(c.o.sig_vars + j - 1);
                                    w = (uint32_t)(P.v(c.o.var_lb + sv) + __ffs(ldom[sv]) - 1);
                                } else if (j < KL)
                                    w = (new_expire >> (j - 1 - c.n_sig)) & 1u;
                                kw[t] = w;
                                if (j < KL) hterm ^= key_term(j, w);
                            }
                            const unsigned long long hx = (unsigned long long)row_xor((uint32_t)(hterm >> 32)) << 32 | row_xor((uint32_t)hterm);
                            const unsigned long long h = mix_final(kHashSeed ^ hx);
                            const int owner = key_owner(h, c.world, KL, next_tag);
                            // edge label and the time-advanced block (variableAdvanceOneTimeStep)
                            uint32_t evals[DQ], nblk[DQ];
#pragma unroll
                            for (int q = 0; q < DQ; q++) {
                                const int idx = q * 16 + lane16;
                                evals[q] = idx < N ? (uint32_t)(P.v(c.o.var_lb + idx) + __ffs((int)d[q]) - 1) : 0u;
                                uint32_t nb = 0;
                                if (idx < NK) {
                                    const int pp = idx / N, v = idx - pp * N;
                                    nb = (pp + 1 < K) ? (uint32_t)ldom[idx + N] : (uint32_t)P.v(c.o.var_init + v);
                                }
                                nblk[q] = nb;
                            }
                            if (c.sharded && owner != c.rank) {
                                // the successor state belongs to another shard: candidate record for its owner
                                uint32_t pos = 0;
                                if (lane16 == 0) pos = atomicAdd(&c.ctl[L_.cand0 + (owner * R + ro) * CST], 1u);
                                pos = row_get_c<0>(pos);
                                if (pos + 1u > cand_cap) {
                                    raise(ERR_CAND_OVERFLOW);
                                } else {
                                    uint32_t *rec = c.cand + ((size_t)(owner * R + ro) * cand_cap + pos) * c.CS;
                                    if (lane16 < 6)
                                        rec[lane16] = lane16 == 0 ? h0
                                                    : (lane16 == 1 ? h1
                                                    : (lane16 == 2 ? next_tag : (lane16 == 3 ? new_expire : (lane16 == 4 ? (uint32_t)h : (uint32_t)(h >> 32)))));
#pragma unroll
                                    for (int t = 0; t < KT; t++) {
                                        const int j = t * 16 + lane16;
                                        if (j >= 1 && j < KL) rec[kCandHdr + j - 1] = kw[t];
                                    }
                                    uint32_t *vals = rec + kCandHdr + c.sig_len;
                                    uint32_t *blk = vals + N;
#pragma unroll
                                    for (int q = 0; q < DQ; q++) {
                                        const int idx = q * 16 + lane16;
                                        if (idx < N) vals[idx] = evals[q];
                                        if (idx < NK) blk[idx] = nblk[q];
                                    }
                                }
                            } else {
                                // ---- commit right here: lookup-or-insert the state, append the edge (table_commit of the row)
                                const uint32_t htag = (uint32_t)(h >> 32) | 0x80000000u;
                                uint32_t pos = (uint32_t)h & c.slot_mask;
                                uint32_t e = 0;
                                if (lane16 == 0) e = atomicAdd(&c.ctl[L_.edge0 + ro * CST], 1u);
                                uint32_t idx_state = 0;
                                bool is_new = false, ok = true;
                                // No inner spin on a slot another wavefront -- or ANOTHER ROW OF THIS ONE -- is publishing: a row
                                // that meets a pending slot just goes round this loop again. Rows of one wavefront run an iteration
                                // together (claim, key store and publication happen inside one iteration, before the back edge), so
                                // a pending slot claimed by a sister row is published by the time the waiting row looks again.
                                for (unsigned probes = 0;;) {
                                    unsigned long long sv = 0;
                                    bool claimed = false;
                                    if (lane16 == 0) {
                                        sv = __hip_atomic_load(&c.slots[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        if (sv == 0) {
                                            const unsigned long long want = ((unsigned long long)htag << 32) | kPending;
                                            const unsigned long long old = atomicCAS(&c.slots[pos], 0ull, want);
                                            claimed = old == 0;
                                            sv = old;
                                        }
                                    }
                                    const uint32_t slo = row_get_c<0>((uint32_t)sv), shi = row_get_c<0>((uint32_t)(sv >> 32));
                                    claimed = row_get_c<0>(claimed ? 1u : 0u) != 0u;
                                    if (claimed) {
                                        uint32_t ni = 0;
                                        if (lane16 == 0) ni = atomicAdd(&c.ctl[L_.misc0 + MISC_NSTATES * CST], 1u);
                                        ni = row_get_c<0>(ni);
                                        if (ni >= c.state_cap) {
                                            raise(ERR_STATE_OVERFLOW);
                                            ok = false;
                                            break;
                                        }
#pragma unroll
                                        for (int t = 0; t < KT; t++) {
                                            const int j = t * 16 + lane16;
                                            if (j < KL) __hip_atomic_store(&c.state_keys[(size_t)ni * KL + j], kw[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        }
                                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                        if (lane16 == 0)
                                            __hip_atomic_store(&c.slots[pos], ((unsigned long long)htag << 32) | ni, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        idx_state = ni;
                                        is_new = true;
                                        break;
                                    }
                                    if (shi == htag) {
                                        if (slo == kPending) {  // being published: look again
                                            __builtin_amdgcn_s_sleep(1);
                                            if (++probes > (1u << 24)) {
                                                raise(ERR_TABLE_SPIN);
                                                ok = false;
                                                break;
                                            }
                                            continue;
                                        }
                                        bool diff = false;
#pragma unroll
                                        for (int t = 0; t < KT; t++) {
                                            const int j = t * 16 + lane16;
                                            if (j < KL)
                                                diff = diff || __hip_atomic_load(&c.state_keys[(size_t)slo * KL + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != kw[t];
                                        }
                                        if (!row_any(diff, rbase)) {
                                            idx_state = slo;
                                            break;
                                        }
                                    }
                                    pos = (pos + 1) & c.slot_mask;
                                    if (++probes > c.slot_mask + (1u << 24)) {
                                        raise(ERR_STATE_OVERFLOW);
                                        ok = false;
                                        break;
                                    }
                                }
                                e = row_get_c<0>(e);
                                if (ok && e >= c.edge_cap) {
                                    raise(ERR_EDGE_OVERFLOW);
                                    ok = false;
                                }
                                if (ok) {
                                    uint32_t *er = c.edges + ((size_t)ro * c.edge_cap + e) * c.ES;
                                    if (lane16 < 4) er[lane16] = lane16 == 0 ? h0 : (lane16 == 1 ? h1 : (lane16 == 2 ? idx_state : 0u));
#pragma unroll
                                    for (int q = 0; q < DQ; q++) {
                                        const int kx = q * 16 + lane16;
                                        if (kx < N) er[4 + kx] = evals[q];
                                    }
                                    if (is_new) {
                                        // the leaf opened a new state: its first node is next in this row's chain
                                        stat(ST_NEWSTATES, 1u);
                                        const unsigned long long gid = ((unsigned long long)c.rank << STCSP_GID_SHIFT) | idx_state;
                                        h0 = (uint32_t)gid;
                                        h1 = (uint32_t)(gid >> 32);
                                        set = next_set;
                                        seed = 0;
                                        expire = new_expire;
#pragma unroll
                                        for (int q = 0; q < DQ; q++) d[q] = nblk[q];
                                        cont = true;
                                    }
                                }
                            }
                        }
                    }
                }
            }  // if (have)

            // =========================================================== end of the step (the whole wavefront again)
            const bool last = step >= chain || __builtin_amdgcn_s_memtime() - t_slot > chain_cycles;
            const uint32_t cbits = rows_of(cont), ebits = rows_of(extra);
            const int nC = __popc(cbits), nE = __popc(ebits);
            const uint32_t hdr2 = (uint32_t)set | seed << 16;
            if (last) {
                // everything alive -- the rows' next nodes, the upper children, the stack -- goes to the frontier: one cursor bump
                const int T = nC + nE + sp;
                if (T > 0) {
                    uint32_t pos = 0;
                    if (lane == 0) pos = atomicAdd(&c.ctl[L_.out(parity, ro)], (uint32_t)T);
                    pos = rflu(pos);
                    if (pos + (uint32_t)T > out_cap) {
                        raise(ERR_OUT_OVERFLOW);
                    } else {
                        if (cont) store_row_node(out_region + (size_t)(pos + (uint32_t)__popc(cbits & below)) * NS, h0, h1, hdr2, expire, d);
                        if (extra) store_row_node(out_region + (size_t)(pos + (uint32_t)nC + (uint32_t)__popc(ebits & below)) * NS, h0, h1, hdr2, expire, child);
                        uint32_t *dst = out_region + (size_t)(pos + (uint32_t)(nC + nE)) * NS;
                        for (int w = lane; w < sp * NS; w += 64) dst[w] = (uint32_t)smem[stk_off + w];
                    }
                }
                break;
            }
            // upper children -> stack (a full stack spills to the frontier; cannot happen while chain <= kRowMaxChain)
            if (extra) {
                const int slot = sp + __popc(ebits & below);
                if (slot < kRowStack) {
                    const int sb = stk_off + slot * NS;
                    if (lane16 < 4) smem[sb + lane16] = (int)(lane16 == 0 ? h0 : (lane16 == 1 ? h1 : (lane16 == 2 ? hdr2 : expire)));
#pragma unroll
                    for (int q = 0; q < DQ; q++) {
                        const int idx = q * 16 + lane16;
                        if (idx < NK) smem[sb + 4 + idx] = (int)child[q];
                    }
                } else {
                    uint32_t pos = 0;
                    if (lane16 == 0) pos = atomicAdd(&c.ctl[L_.out(parity, ro)], 1u);
                    pos = row_get_c<0>(pos);
                    if (pos + 1u > out_cap)
                        raise(ERR_OUT_OVERFLOW);
                    else
                        store_row_node(out_region + (size_t)pos * NS, h0, h1, hdr2, expire, child);
                }
            }
            sp = min(sp + nE, kRowStack);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // rows whose chain ended take the youngest records
            const uint32_t ibits = rows_of(!cont);
            const int ntake = min(__popc(ibits), sp);
            const int myrank = __popc(ibits & below);
            have = cont;
            if (!cont && myrank < ntake) {
                const int sb = stk_off + (sp - 1 - myrank) * NS;
                h0 = (uint32_t)smem[sb];
                h1 = (uint32_t)smem[sb + 1];
                const uint32_t sw2 = (uint32_t)smem[sb + 2];
                set = (int)(sw2 & 0xffffu);
                seed = sw2 >> 16;
                expire = (uint32_t)smem[sb + 3];
#pragma unroll
                for (int q = 0; q < DQ; q++) {
                    const int idx = q * 16 + lane16;
                    d[q] = idx < NK ? (uint32_t)smem[sb + 4 + idx] : 0u;
                }
                have = true;
            }
            sp -= ntake;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (!__ballot(have)) break;  // the subtree is exhausted
        }
    }

    // ---- end of the launch: counters and the error code of this wavefront reach global memory
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    {
        const unsigned v = lane < 15 ? (unsigned)wstat[lane] : 0u;
        if (v) atomicAdd(&c.stats[(wave_g % kStatSlots) * kStatWords + lane], (unsigned long long)v);
        const unsigned err = (unsigned)wstat[15];
        if (err && lane == 0) atomicMax(&c.ctl[L_.misc0 + MISC_ERROR * CST], err);
    }
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

}  // namespace dev
}  // namespace stcsp
