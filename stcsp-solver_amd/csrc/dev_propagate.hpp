// dev_propagate.hpp -- wave-level device code of one search node: helpers, the postfix interpreter,
// the wavefront-cooperative revision (revise_point), the lane-per-item sweep and node classification
// (process_node). Included by engine.hip only.
#pragma once
#include "dev_layout.hpp"
namespace stcsp {
namespace dev {
// ------------------------------------------------------------------ device helpers
// View of the program image. L = 1: the whole image is in the workgroup's LDS copy. L = 0: a prefix of `nlds` words is staged,
// the rest is read from global memory -- every read decides at run time, which the compiler turns into FLAT loads through a
// selected pointer (the general kernel had 185 of them; a flat load that lands in LDS goes through the vector-memory path first
// and ties the LDS counter to the vector-memory one). L = 2 (round 4): the engine guarantees that EVERYTHING but the `tables`
// section (tuple bitmaps: vc / uc / v4c) is staged -- v / u / v4 are plain LDS reads, the tables plain global loads; what
// digitinvader's programs (12 KB of descriptors beside 4.9 MB of bitmaps) run under.
// u(): wave-uniform read (scalar load / broadcast LDS read), v(): per-lane read.
template <int L>
struct Img {
    const uint32_t *p;    // the image in global memory
    const uint32_t *lds;  // its staged copy: the whole image when L, else the first `nlds` words
    int nlds;
    __device__ __forceinline__ int v(int off) const;
    __device__ __forceinline__ uint4 v4(int off) const;  // off % 4 == 0 (sections and records are 16-byte aligned)
    __device__ __forceinline__ int u(int off) const;     // wave-uniform offset -> scalar value
    // the same for the tables section (the last and potentially big one), which is never part of a staged
    // prefix of a partly staged image: no boundary test on the paths that read it most
    __device__ __forceinline__ int vc(int off) const;
    __device__ __forceinline__ uint4 v4c(int off) const;
    __device__ __forceinline__ int uc(int off) const;
};
// The compiled program (bytecode, descriptors, tables) is read-only for the lifetime of a launch
// and indexed wave-uniformly: reading it through the constant address space makes hipcc emit
// scalar loads (s_load_dword through the scalar cache) instead of 64-lane vector loads.
typedef const __attribute__((address_space(4))) int *kptr;
__device__ __forceinline__ int kload(const void *base, int idx) {
    return ((kptr)(const __attribute__((address_space(1))) int *)base)[idx];
}
template <>
__device__ __forceinline__ int Img<1>::v(int off) const { return (int)lds[off]; }
template <>
__device__ __forceinline__ uint4 Img<1>::v4(int off) const { return *(const uint4 *)(lds + off); }
template <>
__device__ __forceinline__ int Img<1>::u(int off) const { return __builtin_amdgcn_readfirstlane((int)lds[off]); }
template <>
__device__ __forceinline__ int Img<1>::vc(int off) const { return (int)lds[off]; }
template <>
__device__ __forceinline__ uint4 Img<1>::v4c(int off) const { return *(const uint4 *)(lds + off); }
template <>
__device__ __forceinline__ int Img<1>::uc(int off) const { return __builtin_amdgcn_readfirstlane((int)lds[off]); }
// everything but the tables staged
template <>
__device__ __forceinline__ int Img<2>::v(int off) const { return (int)lds[off]; }
template <>
__device__ __forceinline__ uint4 Img<2>::v4(int off) const { return *(const uint4 *)(lds + off); }
template <>
__device__ __forceinline__ int Img<2>::u(int off) const { return __builtin_amdgcn_readfirstlane((int)lds[off]); }
template <>
__device__ __forceinline__ int Img<2>::vc(int off) const { return (int)p[off]; }
template <>
__device__ __forceinline__ uint4 Img<2>::v4c(int off) const { return *(const uint4 *)(p + off); }
template <>
__device__ __forceinline__ int Img<2>::uc(int off) const { return kload(p, off); }
// image larger than the LDS budget: a prefix of hot sections is staged, the rest is read from HBM/L2
template <>
__device__ __forceinline__ int Img<0>::v(int off) const { return off < nlds ? (int)lds[off] : (int)p[off]; }
template <>
__device__ __forceinline__ uint4 Img<0>::v4(int off) const {
    return off < nlds ? *(const uint4 *)(lds + off) : *(const uint4 *)(p + off);
}
template <>
__device__ __forceinline__ int Img<0>::u(int off) const {
    return off < nlds ? __builtin_amdgcn_readfirstlane((int)lds[off]) : kload(p, off);
}
template <>
__device__ __forceinline__ int Img<0>::vc(int off) const { return (int)p[off]; }
template <>
__device__ __forceinline__ uint4 Img<0>::v4c(int off) const { return *(const uint4 *)(p + off); }
template <>
__device__ __forceinline__ int Img<0>::uc(int off) const { return kload(p, off); }
// After a lane-predicated statement (`if (lane < n) store`) that is followed by a return / break / continue the
// compiler threads the two sides of the predicate straight into the join of the exits: to its uniformity
// analysis the exit then hangs on a divergent branch, and one divergent loop exit turns every value carried
// round the node loops into a VGPR (and every branch on them into exec-mask code). A convergent no-op forces
// the lanes to rejoin first. (tools/uniformity.sh lists the loops with divergent exits.)
#define STCSP_REJOIN() __builtin_amdgcn_wave_barrier()
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
struct WaveStats {  // STCSP_PHASES build only: cycle shares of one node
    unsigned long long cyc_sweep = 0, cyc_wave = 0, cyc_rv_setup = 0, cyc_rv_loop = 0, cyc_rv_wb = 0, cyc_close = 0;
    unsigned rv_blocks = 0, rv_open = 0, rv_lanes = 0;  // blocks, open variables, tuple lanes of the general revisions
    unsigned long long cyc_rv_digits = 0, cyc_rv_eval_bitmap = 0, cyc_rv_eval_code = 0, cyc_rv_support = 0;
    unsigned rv_blocks_code = 0;
    unsigned batches = 0, batch_items = 0, batch_refused = 0, batch_tuples = 0;
    unsigned long long cyc_batch = 0, cyc_batch_ab = 0, cyc_batch_de = 0;
};

// What a wavefront keeps across the nodes it expands in one launch:
//  * the descriptor of the constraint set it last worked under (absolute image offsets, the eager-arc partner
//    entry of each of its block words, the masks derived from them) -- consecutive nodes of a chain or of a
//    cursor region nearly always share the set, so the ~20 LDS reads this takes are paid once, not per node;
//  * its work counters and a sticky error code, which reach global memory once, at the end of the launch
//    (per-node atomics, even LDS ones by lane 0, are a divergent branch and a wait each).
// Everything is wave-uniform (SGPRs) unless marked "per lane".
template <int DR>
struct WaveEnv {
    int set = -1;
    int self_loop = 0, nfirst = 0, first_off = 0, trans_begin = 0, trans_count = 0, nitems = 0, nsmall = 0, iw = 1;
    int rows_abs = 0;   // the set's [N*K][iw] dirty rows
    int sweep_abs = 0;  // its packed sweep records
    int items_abs = 0;  // the ItemDesc record of its item i (wavefront-revised: i >= nsmall) is at items_abs + 12 i
    int next_abs = -1;  // its eager-arc partner entries (-1: the set has none)
    unsigned long long pm[DR] = {};  // block words (bit l = word q*64 + l) that have an eager partner
    uint32_t smallmask = 0;          // per lane: bits of the lane-revised items in dirty word `lane`
    uint32_t bmmask = 0;             // per lane: bits of the wavefront-revised items with a tuple bitmap (general kernels only)
    uint32_t e0[DR] = {};            // per lane: partner entry 0 of block word q*64 + lane
    unsigned n_nodes = 0, n_fails = 0, n_leaves = 0, n_requeue = 0, n_revs = 0, n_wave_revs = 0, n_sweeps = 0, n_skipped = 0, n_new = 0;
    unsigned long long n_evals = 0;
    uint32_t rows_seen = 0;  // per lane: table rows examined by this lane's sweeps
    unsigned err = 0;        // MISC_ERROR code
};

// Evaluate one constraint program on this lane's tuple (the role of solverValidateRe,
// reference src/solveralgorithm.cpp:336-424). varinfo/curval are per-lane registers indexed by
// scope position: varinfo = 1 + slot for lane-enumerated variables (value in lds_vals), 0 for
// wave-uniform ones (value in curval).
template <int L>
__device__ int eval_program(const Ctx &c, const Img<L> &P, int pc0, int code_len, bool uses_valid, int lane, uint32_t varinfo, int curval,
                            const int *lds_vals, int *lds_stk) {
    // Operand stack: the top in `t`, the three entries below it in registers (s1 = most recent), anything deeper in
    // LDS -- expressions rarely nest deeper, so a push / pop is a few register moves instead of an LDS round trip
    // per instruction (a dependent ds_read is ~100 cycles, and an instruction used to cost ~350).
    int t = 0, s1 = 0, s2 = 0, s3 = 0, sp = 0;  // sp = entries below t
    auto push = [&]() {  // push t (the caller then sets the new top)
        if (sp >= 3) lds_stk[(sp - 3) * 64 + lane] = s3;
        s3 = s2;
        s2 = s1;
        s1 = t;
        sp++;
    };
    auto pop = [&]() -> int {  // the entry below t
        const int x = s1;
        s1 = s2;
        s2 = s3;
        if (sp > 3) s3 = lds_stk[(sp - 4) * 64 + lane];
        sp--;
        return x;
    };
    auto peek = [&](int d) -> int {  // d entries below t (d >= 1)
        return d == 1 ? s1 : (d == 2 ? s2 : (d == 3 ? s3 : lds_stk[(sp - d) * 64 + lane]));
    };
    bool valid = true;
    uint32_t dead = 0;
    // The program itself lives in two VGPRs for the duration of the call (word k in lane k): fetching an
    // instruction is a v_readlane instead of a dependent LDS / scalar-memory read per instruction.
    const uint32_t cv0 = lane < code_len ? (uint32_t)P.v(c.o.code + pc0 + lane) : 0u;
    const uint32_t cv1 = 64 + lane < code_len ? (uint32_t)P.v(c.o.code + pc0 + 64 + lane) : 0u;
    auto fetch = [&](int rel) -> int {
        if (rel < 64) return (int)rdlane(cv0, rel);
        if (rel < 128) return (int)rdlane(cv1, rel - 64);
        return P.u(c.o.code + pc0 + rel);
    };
    int pc = 0;
    for (;;) {
        int w = fetch(pc++);
        int op = w & 255, arg = w >> 8;
        switch (op) {
            case OP_END: return t;
            case OP_CONST:
                push();
                t = fetch(pc++);
                break;
            case OP_VAR: {
                push();
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
                int v = arg == 0 ? t : peek(arg);
                bool live = (op == OP_MASK_T) ? (v != 0) : (v == 0);
                dead = (dead << 1) | (live ? 0u : 1u);
                break;
            }
            case OP_MASK_POP: dead >>= 1; break;
            case OP_SEL_IF: {
                const int b = t, a = pop(), cnd = pop();
                t = cnd ? a : b;
                break;
            }
            case OP_SEL_AND: {
                const int a = pop();
                t = a ? t : 0;
                break;
            }
            case OP_SEL_OR: {
                const int a = pop();
                t = a ? 1 : t;
                break;
            }
            case OP_SEL_IMPLY: {
                const int a = pop();
                t = (a == 0) ? 1 : (a <= t);
                break;
            }
            default: {
                const int b = t, a = pop();
                int r = 0;
                switch (op) {
                    case OP_ADD: r = (int)((unsigned)a + (unsigned)b); break;
                    case OP_SUB: r = (int)((unsigned)a - (unsigned)b); break;
                    case OP_MUL: r = (int)((unsigned)a * (unsigned)b); break;
                    case OP_DIV:
                    case OP_MOD: {
                        // one division serves both (hipcc expands a 32-bit signed division into ~30
                        // instructions with a dozen temporaries: two of them set the kernel's VGPR peak)
                        const bool safe = !(b == 0 || (a == INT_MIN && b == -1));
                        const int q = safe ? a / b : 0;
                        r = op == OP_DIV ? q : (safe ? (int)((unsigned)a - (unsigned)q * (unsigned)b) : 0);
                        break;
                    }
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

// Inclusive prefix sum over the 64 lanes with DPP moves (row shifts inside each row of 16, then the two
// row broadcasts): six VALU instructions, no LDS crossbar latency (a __shfl_xor butterfly is six
// dependent ds_bpermute round trips, ~800 cycles). All 64 lanes must be active. Lane 63 ends up with
// the total.
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_add(v), 63); }
__device__ __forceinline__ uint32_t wave_xor32(uint32_t v) {
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_or32(uint32_t v) {  // all 64 lanes active
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_xor64(unsigned long long v) {
    return (unsigned long long)wave_xor32((uint32_t)(v >> 32)) << 32 | wave_xor32((uint32_t)v);
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
// x / d for 0 <= x < 4096, 1 <= d <= 64 via one reciprocal (exact: (x + 0.5) / d is never within
// 1/128 of an integer, far above the float error of 4096 * 2^-22)
__device__ __forceinline__ int small_div(int x, int d) {
    return (int)(((float)x + 0.5f) * __builtin_amdgcn_rcpf((float)d));
}

// Enforce one point constraint at one time point: afterwards every remaining value of every
// scope variable has a supporting tuple (generalised arc consistency on this constraint; the
// reference tightens bounds only, solveralgorithm.cpp:476-523 -- this prunes at least as much).
// Returns false when a domain is wiped out. Rows of changed block words are OR-ed into `dirtyw`.
template <int DR, int L, bool LITE>
__device__ bool revise_point(const Ctx &c, const Img<L> &G, WaveEnv<DR> &S, const ConDesc &C, int item, int p, Dom<DR> &dom,
                             int lane, uint32_t &dirtyw, int *lds_vals, int *lds_stk, int *ldom, WaveStats &ws, bool &pruned) {
    const unsigned long long(&pm)[DR] = S.pm;
    const unsigned long long t_rv0 = PHASE_NOW();
    (void)t_rv0;
    const int s = C.scope_len;
    // per-lane view of scope variable j = lane
    int var = 0;
    if (lane < s) var = G.v(c.o.scope + C.scope_off + lane);
    uint32_t D = dom.gather(p * c.N + var);
    if (lane >= s) D = 0;
    const int n = lane < s ? __popc(D) : 1;
    if (__ballot(lane < s && n == 0)) return false;
    // Constraints with few violating tuples in all (wide disjunctions / implications such as
    // succ >= (seen0 and ... and seen13): one violating tuple out of 2^15): a value loses its support
    // only when the whole product of the OTHER domains is forbidden, so unless
    // prod(all domains) / (largest domain) <= n_forbidden this revision cannot prune anything.
    if (C.n_forbidden >= 0) {
        const unsigned long long open = __ballot(lane < s && n > 1);
        const int nopen = __popcll(open);
        // prod >= 2^nopen and largest <= 32; and any two open variables already give prod >= 2 * largest
        bool skip = (C.n_forbidden <= 1 && nopen >= 2) || C.n_forbidden == 0 || nopen >= 12 ||
                    (1u << nopen) > (unsigned)C.n_forbidden * 32u;
        if (!skip && nopen >= 2) {
            unsigned long long prod = 1;
            int largest = 1;
            for (unsigned long long m = open; m; m &= m - 1) {
                const int nj = (int)rdlane((uint32_t)n, __ffsll((long long)m) - 1);
                prod *= (unsigned long long)nj;
                if (nj > largest) largest = nj;
            }
            skip = prod > (unsigned long long)C.n_forbidden * (unsigned long long)largest;
        }
        if (skip) {
            if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
            return true;
        }
    }
    const bool use_bitmap = C.bitmap_off >= 0;
    const int mystride = (use_bitmap && lane < s) ? G.v(c.o.strides + C.stride_off + lane) : 0;
    // Fast path (tuple-bitmap constraints): at most ONE scope variable is not yet fixed -- the usual
    // situation deep in the tree. No tuple enumeration to set up: lane v looks value v of that variable
    // up (one block of <= 32 tuples), or the single tuple is checked when everything is fixed.
    {
        const unsigned long long open_vars = __ballot(lane < s && n > 1);
        if (use_bitmap && (open_vars & (open_vars - 1)) == 0) {
            const int fixed_part = wave_sum((lane < s && n == 1) ? (__ffs((int)D) - 1) * mystride : 0);
            const int tab = c.o.tables + C.bitmap_off;
            S.n_revs++;
            S.n_wave_revs++;
            if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
            if (!open_vars) {
                S.n_evals += 1;
                return ((rflu((uint32_t)G.vc(tab + (fixed_part >> 5))) >> (fixed_part & 31)) & 1u) != 0;  // rfl: a loaded value is divergent to the compiler
            }
            const int j = __ffsll((long long)open_vars) - 1;
            const uint32_t Dj = rdlane(D, j);
            const int strj = (int)rdlane((uint32_t)mystride, j);
            const bool has = lane < 32 && ((Dj >> lane) & 1u);
            const int bit = fixed_part + lane * strj;
            const bool sat = has && ((((uint32_t)G.vc(tab + (bit >> 5))) >> (bit & 31)) & 1u);
            const uint32_t newD = (uint32_t)__ballot(sat);
            S.n_evals += (unsigned)__popc(Dj);
            if (newD == 0) return false;
            if (newD != Dj) {
                const int w = p * c.N + (int)rdlane((uint32_t)var, j);
#pragma unroll
                for (int q = 0; q < DR; q++)
                    if ((w >> 6) == q && ((pm[q] >> (w & 63)) & 1ull)) pruned = true;  // a next arc hangs on this word
                dom.set(w, newD, lane);
                if (lane == 0) ldom[w] = (int)newD;  // keep the sweep's LDS copy of the block current
                if (lane < S.iw) dirtyw |= (uint32_t)G.v(S.rows_abs + w * S.iw + lane);
                if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));  // a revision is a fixpoint for its own constraint
            }
            return true;
        }
    }
    if constexpr (LITE) {
        // LITE kernels are selected for programs whose wavefront-revised constraints can only take the two
        // exits above (tuple bitmap with at most one violating tuple: engine.hip upload_program); anything
        // else is skipped, which is sound (a leaf's single tuple is checked by the fast path)
        if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
        S.n_skipped++;
        return true;
    }
    const int vlb = lane < s ? G.v(c.o.var_lb + var) : 0;

    // --- general revision: enumerate the product of the OPEN (non-singleton) scope variables, evaluate every
    // tuple (bitmap look-up or postfix program) and collect, per variable, the values that occur in a satisfying
    // tuple. The open variables are split: "low" ones, whose domain sizes multiply to <= 64, are enumerated
    // ACROSS LANES (lane = mixed-radix tuple index; the digits are worked out once per revision), the rest
    // ("high") are stepped wave-uniformly by an odometer, one block of <= 64 tuples per step. Singletons are
    // constants. Most revisions have no high variable: one block.
    const unsigned long long openm = __ballot(lane < s && n > 1);
    // Tuple bitmaps: scope variable 0 (stride 1: normally the X of `X == ...`) is NEVER enumerated. The bits of all its
    // values for one tuple of the OTHER variables are contiguous, so one (possibly straddling) 32-bit window of the
    // bitmap ANDed with its domain says at once whether that tuple has a support and which values of variable 0 it
    // supports: the product shrinks by |D(variable 0)| (11-fold for the `next D == if ...` constraints of digitinvader).
    const unsigned long long enumm = use_bitmap ? (openm & ~1ull) : openm;
    const uint32_t D0 = rdlane(D, 0);
    unsigned long long lowmask = 0, highmask = 0;
    int Plow = 1, nlow = 0;
    unsigned long long nsteps = 1;  // odometer range = product of the high domain sizes
    // Budget. Pruning a value needs the WHOLE product of the other variables refuted; when the odometer range
    // is larger than the budget the revision could never finish, so it is skipped outright. This keeps
    // propagation sound (no value is ever removed without proof) and the search complete: at a leaf every
    // variable is a singleton, the product is 1 and the constraint is checked exactly -- the same argument that
    // makes the reference's weaker, bounds-only propagation (solveralgorithm.cpp:476-523) yield the same automaton.
    const unsigned long long budget = (unsigned long long)(unsigned)(use_bitmap ? c.budget_bitmap : c.budget_code);
    for (unsigned long long m = enumm; m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const int nj = (int)rdlane((uint32_t)n, j);
        if (nlow < kMaxLowVars && Plow * nj <= 64) {
            lowmask |= 1ull << j;
            Plow *= nj;
            nlow++;
        } else {
            highmask |= 1ull << j;
            if (nsteps <= budget) nsteps *= (unsigned long long)nj;
        }
    }
    if (nsteps > budget) {
        if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
        S.n_skipped++;
        return true;
    }
    const unsigned long long t_rv1 = PHASE_NOW();
    (void)t_rv1;
    // low variables: this lane's digit / value bit of each (once), packed 5 bits per variable
    const bool active = lane < Plow;
    int lane_part = 0;     // bitmap index contribution of the low variables
    uint32_t pack = 0;     // value bit of low variable `slot` in bits [5 slot, 5 slot + 5)  (kMaxLowVars <= 6)
    {
        int Pj = 1, slot = 0;
        for (unsigned long long m = lowmask; m; m &= m - 1, slot++) {
            const int j = __ffsll((long long)m) - 1;
            const int nj = (int)rdlane((uint32_t)n, j);
            const uint32_t Dj = rdlane(D, j);
            // Pj, nj, Dj are wave-uniform: powers of two divide by shifting; lane < 64 divides exactly through one reciprocal
            const int q = (Pj & (Pj - 1)) == 0 ? (lane >> (__ffs(Pj) - 1)) : small_div(lane, Pj);
            const int digit = (nj & (nj - 1)) == 0 ? (q & (nj - 1)) : q - nj * small_div(q, nj);
            int bitpos = 0;
            if (nj <= 4) {
                int k = 0;
                for (uint32_t mm = Dj; mm; mm &= mm - 1, k++) bitpos = digit == k ? __ffs((int)mm) - 1 : bitpos;
            } else {
                bitpos = select_kth_fast(Dj, digit);
            }
            pack |= (uint32_t)bitpos << (5 * slot);
            if (use_bitmap)
                lane_part += bitpos * (int)rdlane((uint32_t)mystride, j);
            else
                lds_vals[slot * 64 + lane] = (int)rdlane((uint32_t)vlb, j) + bitpos;
            Pj *= nj;
        }
    }
    // scope lane j: 1 + slot for a low variable (the interpreter reads its value from lds_vals), 0 otherwise
    const uint32_t varinfo = ((lowmask >> lane) & 1ull) ? 1u + (uint32_t)__popcll(lowmask & ((1ull << lane) - 1ull)) : 0u;
    const bool is_high = (highmask >> lane) & 1ull;
    int curbit = lane < s ? (__ffs((int)D) - 1) : 0;  // singletons: their bit; high variables: the odometer's current bit
    int curval = vlb + curbit;
    int digit_h = 0;
    int base_sum = 0;  // bitmap index contribution of the singletons
    if (use_bitmap) base_sum = wave_sum((lane >= 1 && lane < s && n == 1) ? curbit * mystride : 0);  // (variable 0: the window)
    uint32_t acc0 = 0;  // tuple lanes: values of variable 0 this lane's tuples support
    S.n_revs++;
    S.n_wave_revs++;
#ifdef STCSP_PHASES
    ws.rv_open += (unsigned)__popcll(openm);
    ws.rv_lanes += (unsigned)Plow;
#endif
    uint32_t hs = 0;      // scope lanes of high variables: supported value bits so far
    bool satany = false;  // tuple lanes: this low tuple was satisfied in some block
    bool any_sat = false;
    // Tuple bitmaps with an odometer: the steps of the odometer are known ahead (wave-uniform digits), so the bitmap windows of
    // kOdoAhead steps are requested TOGETHER and examined afterwards -- one global-memory round trip (~900 cycles: the bitmaps are
    // megabytes, L2 at best) per group instead of one per step. A fresh state's items over the new time point run the whole
    // budget of 32 steps without pruning anything: digitinvader9 -8 %, the first node of a digitinvader state 115 -> ~90 us. The
    // "everything supported already" test is made once per group: up to kOdoAhead - 1 steps more than before, which can only
    // find supports that are already there.
    constexpr int kOdoAhead = 4;
    if (use_bitmap && highmask) {
        bool wrapped = false;
        // Diagonal first: before the odometer counts through the product, the tuples in which EVERY high variable takes its d-th
        // value (d = 0 .. 3; the last one where a domain is shorter) are looked at. A loose constraint -- digitinvader's
        // `next D7 == if GAMEOVER then -1 else if MISS or A0 or ... or A7 then D8 else D7` with the A's still open -- shows a support
        // for every value of every variable within two of them, where the odometer needs half its range (2^(h-1) + 1 steps for
        // h boolean high variables: 17 of 32) before the last variable has shown its second value; the "everything supported
        // already" exit then ends the revision. Nothing else changes: when the diagonal does not settle it, the odometer
        // enumerates the whole product as before (supports do not depend on the order they are found in).
        if constexpr (kOdoAhead == 4) {
            uint32_t wlo[4], whi[4];
            int cb[4], shv[4];
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int dig = min(d, n - 1);
                cb[d] = is_high ? select_kth_fast(D, dig) : curbit;
                int bit = lane_part + base_sum;
                for (unsigned long long hm = highmask; hm; hm &= hm - 1) {
                    const int j = __ffsll((long long)hm) - 1;
                    bit += (int)rdlane((uint32_t)cb[d], j) * (int)rdlane((uint32_t)mystride, j);
                }
                const int tw = c.o.tables + C.bitmap_off + (bit >> 5), sh = bit & 31;
                shv[d] = sh;
                wlo[d] = active ? (uint32_t)G.vc(tw) : 0u;
                whi[d] = (active && sh + (31 - __clz((int)D0)) >= 32) ? (uint32_t)G.vc(tw + 1) : 0u;
            }
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t sup0 = (uint32_t)((((unsigned long long)whi[d] << 32) | wlo[d]) >> shv[d]) & D0;
                acc0 |= sup0;
                const bool sat = active && sup0 != 0;
                if (__ballot(sat)) {
                    any_sat = true;
                    satany = satany || sat;
                    if (is_high) hs |= 1u << cb[d];
                }
            }
            S.n_evals += (unsigned)(Plow * 4);
#ifdef STCSP_PHASES
            ws.rv_blocks += 4u;
#endif
            wrapped = __ballot((active && !satany) || (is_high && hs != D)) == 0 && wave_or32(acc0) == D0;  // settled: no odometer
        }
        for (unsigned long long step = 0; !wrapped; step += kOdoAhead) {
            const unsigned long long t_b1 = PHASE_NOW();
            (void)t_b1;
            uint32_t wlo[kOdoAhead], whi[kOdoAhead];
            int cbit[kOdoAhead], shv[kOdoAhead];
            int nv = 0;
#pragma unroll
            for (int u = 0; u < kOdoAhead; u++) {
                wlo[u] = whi[u] = 0u;
                cbit[u] = curbit;
                shv[u] = 0;
                if (!wrapped) {
                    int bit = lane_part + base_sum;
                    for (unsigned long long hm = highmask; hm; hm &= hm - 1) {
                        const int j = __ffsll((long long)hm) - 1;
                        bit += (int)rdlane((uint32_t)curbit, j) * (int)rdlane((uint32_t)mystride, j);
                    }
                    const int tw = c.o.tables + C.bitmap_off + (bit >> 5), sh = bit & 31;
                    shv[u] = sh;
                    wlo[u] = active ? (uint32_t)G.vc(tw) : 0u;
                    whi[u] = (active && sh + (31 - __clz((int)D0)) >= 32) ? (uint32_t)G.vc(tw + 1) : 0u;
                    nv = u + 1;
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
                    wrapped = carry;  // wrapped around: product exhausted
                }
            }
#pragma unroll
            for (int u = 0; u < kOdoAhead; u++) {
                if (u < nv) {
                    const uint32_t sup0 = (uint32_t)((((unsigned long long)whi[u] << 32) | wlo[u]) >> shv[u]) & D0;
                    acc0 |= sup0;
                    const bool sat = active && sup0 != 0;
                    const unsigned long long sm = __ballot(sat);
                    if (sm) {
                        any_sat = true;
                        satany = satany || sat;
                        if (is_high) hs |= 1u << cbit[u];
                    }
                }
            }
            S.n_evals += (unsigned)(Plow * nv);
#ifdef STCSP_PHASES
            ws.rv_blocks += (unsigned)nv;
            ws.cyc_rv_eval_bitmap += PHASE_NOW() - t_b1;
#endif
            if (wrapped) break;
            // everything supported already? (every low tuple satisfied at least once, every high value seen, all of variable 0)
            if (__ballot((active && !satany) || (is_high && hs != D)) == 0 && wave_or32(acc0) == D0) break;
            if (step > (1ull << 22)) {
                S.err = max(S.err, (unsigned)ERR_WATCHDOG);
                return false;
            }
        }
    } else
    for (unsigned long long step = 0;; step++) {
        const unsigned long long t_b1 = PHASE_NOW();
        (void)t_b1;
        int res;
        if (use_bitmap) {
            int bit = lane_part + base_sum;
            for (unsigned long long hm = highmask; hm; hm &= hm - 1) {
                const int j = __ffsll((long long)hm) - 1;
                bit += (int)rdlane((uint32_t)curbit, j) * (int)rdlane((uint32_t)mystride, j);
            }
            // window of variable 0's bits at this tuple of the others (second word only when the domain reaches into it)
            const int tw = c.o.tables + C.bitmap_off + (bit >> 5), sh = bit & 31;
            const uint32_t wlo = active ? (uint32_t)G.vc(tw) : 0u;
            const uint32_t whi = (active && sh + (31 - __clz((int)D0)) >= 32) ? (uint32_t)G.vc(tw + 1) : 0u;
            const uint32_t sup0 = (uint32_t)((((unsigned long long)whi << 32) | wlo) >> sh) & D0;
            acc0 |= sup0;
            res = sup0 != 0;
        } else {
            res = eval_program<L>(c, G, C.code_off, C.code_len, C.uses_valid != 0, lane, varinfo, curval, lds_vals, lds_stk);
        }
        S.n_evals += (unsigned)Plow;
        const bool sat = active && res != 0;
        const unsigned long long sm = __ballot(sat);
#ifdef STCSP_PHASES
        ws.rv_blocks++;
        const unsigned long long t_b2 = PHASE_NOW();
        (use_bitmap ? ws.cyc_rv_eval_bitmap : ws.cyc_rv_eval_code) += t_b2 - t_b1;
        if (!use_bitmap) ws.rv_blocks_code++;
#endif
        if (sm) {
            any_sat = true;
            satany = satany || sat;
            if (is_high) hs |= 1u << curbit;
        }
        if (!highmask) break;  // the lanes covered the whole product
        // everything supported already? (every low tuple satisfied at least once, every high value seen, all of variable 0)
        if (__ballot((active && !satany) || (is_high && hs != D)) == 0 && (!use_bitmap || wave_or32(acc0) == D0)) break;
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
        if (step > (1ull << 22)) {
            S.err = max(S.err, (unsigned)ERR_WATCHDOG);
            return false;
        }
    }
    // supported values of the low variables: the value bits of the lanes whose tuple was satisfied
    {
        const unsigned long long t_s = PHASE_NOW();
        (void)t_s;
        int slot = 0;
        for (unsigned long long m = lowmask; m; m &= m - 1, slot++) {
            const int j = __ffsll((long long)m) - 1;
            const uint32_t got = wave_or32(satany ? 1u << ((pack >> (5 * slot)) & 31u) : 0u);
            if (lane == j) hs = got;
        }
        if (use_bitmap) {  // variable 0: the union of the windows
            const uint32_t got0 = wave_or32(acc0);
            if (lane == 0) hs = got0;
        }
#ifdef STCSP_PHASES
        ws.cyc_rv_support += PHASE_NOW() - t_s;
#endif
    }
    const unsigned long long t_rv2 = PHASE_NOW();
    (void)t_rv2;
    // --- write back. No satisfying tuple at all: wipe-out. Otherwise singletons are supported by
    // construction and only the open variables can lose values.
    if (!any_sat) return false;
    for (unsigned long long m = __ballot(((openm >> lane) & 1ull) && hs != D); m; m &= m - 1) {
        const int j = __ffsll((long long)m) - 1;
        const uint32_t newD = rdlane(hs, j);
        if (newD == 0) return false;
        const int w = p * c.N + (int)rdlane((uint32_t)var, j);
#pragma unroll
        for (int q = 0; q < DR; q++)
            if ((w >> 6) == q && ((pm[q] >> (w & 63)) & 1ull)) pruned = true;  // a next arc hangs on this word
        dom.set(w, newD, lane);
        if (lane == 0) ldom[w] = (int)newD;  // keep the sweep's LDS copy of the block current
        if (lane < S.iw) dirtyw |= (uint32_t)G.v(S.rows_abs + w * S.iw + lane);
    }
    // one revision is a fixpoint for this constraint at this point: no need to revisit it for
    // its own changes (supports are whole tuples of surviving values)
    if (lane == (item >> 5)) dirtyw &= ~(1u << (item & 31));
#ifdef STCSP_PHASES
    ws.cyc_rv_setup += t_rv1 - t_rv0;
    ws.cyc_rv_loop += t_rv2 - t_rv1;
    ws.cyc_rv_wb += PHASE_NOW() - t_rv2;
#endif
    return true;
}

// (Re)load the constraint-set part of the wavefront's environment: one per-lane read of the SetDesc words
// (broadcast with readlane), then the per-lane pieces derived from it.
template <int DR, int L>
__device__ __forceinline__ void load_env(const Ctx &c, const Img<L> &P, int set, int lane, WaveEnv<DR> &E) {
    constexpr int W = (int)(sizeof(SetDesc) / 4);
    const uint32_t w = lane < W ? (uint32_t)P.v(c.o.sets + set * W + lane) : 0u;
#define STCSP_SD(f) (int) rdlane(w, (int)(offsetof(SetDesc, f) / 4))
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
    E.items_abs = c.o.items + (STCSP_SD(witem_begin) - E.nsmall) * (int)(sizeof(ItemDesc) / 4);  // wavefront-revised items only
    E.next_abs = next_off >= 0 ? c.o.nextpart + next_off : -1;
#undef STCSP_SD
    {
        const int left = E.nsmall - lane * 32;
        E.smallmask = (lane < E.iw && left > 0) ? (left >= 32 ? 0xffffffffu : ((1u << left) - 1u)) : 0u;
    }
#pragma unroll
    for (int q = 0; q < DR; q++) {
        const int idx = q * 64 + lane;
        E.e0[q] = (E.next_abs >= 0 && idx < c.NK) ? (uint32_t)P.v(E.next_abs + idx * 2) : 0u;
        E.pm[q] = __ballot(E.e0[q] != 0);
    }
}
__device__ __forceinline__ void add_stats(const Ctx &c, int gw, int which, unsigned long long v) {
    if (v) atomicAdd(&c.stats[(gw % kStatSlots) * kStatWords + which], v);
}
// End of a launch: the wavefront's counters and its error code reach global memory (lane k adds counter k).
template <int DR>
__device__ __forceinline__ void flush_env(const Ctx &c, WaveEnv<DR> &E, int slot, int lane) {
    const unsigned rows = (unsigned)wave_sum((int)E.rows_seen);  // every lane is active here
    unsigned long long v = 0;
    switch (lane) {
        case ST_NODES: v = E.n_nodes; break;
        case ST_FAILS: v = E.n_fails; break;
        case ST_LEAVES: v = E.n_leaves; break;
        case ST_REVS: v = E.n_revs; break;
        case ST_EVALS: v = E.n_evals + rows; break;
        case ST_REQUEUE: v = E.n_requeue; break;
        case ST_NEWSTATES: v = E.n_new; break;
        case ST_WAVEREVS: v = E.n_wave_revs; break;
        case ST_SWEEPS: v = E.n_sweeps; break;
        case ST_SKIPPED: v = E.n_skipped; break;
        default: break;
    }
    if (v) atomicAdd(&c.stats[(slot % kStatSlots) * kStatWords + lane], v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // performed before the workgroup's barrier: the finalizing wavefront may sum them (mirror_plan)
    // RETURNING atomic whose result is consumed: the wavefront has then waited for it, so the error word is in place before
    // the workgroup's barrier and the end-of-round ticket behind it (k_expand) -- finalize_round must not plan another round
    // over an overflow. (A non-returning atomic is not waited for by the barrier's workgroup-scope release on gfx9.)
    if (E.err && lane == 0) {
        const uint32_t old = atomicMax(&c.ctl[CtlLayout(c.world).misc0 + MISC_ERROR * CST], (uint32_t)E.err);
        asm volatile("" ::"v"(old));
    }
    STCSP_REJOIN();
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
    STCSP_REJOIN();
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
    unsigned err;  // !ok: the MISC_ERROR code (the caller reports it)
};
template <int DR>
__device__ CommitOut table_commit(const Ctx &c, int lane, int ro, uint32_t kw, unsigned long long h, uint32_t s0, uint32_t s1,
                                  int set, uint32_t tag, const uint32_t (&vals)[DR]);
template <int DR>
__device__ unsigned emit_state_node(const Ctx &c, int lane, int ro, uint32_t *out_base, uint32_t out_cap, int parity,
                                    const CommitOut &co, uint32_t expire, const uint32_t (&blk)[DR], uint32_t seed);

// ------------------------------------------------------------------ one search node
// Propagate the block in `dom` to its fixpoint under the node's constraint set and classify the
// node like solverSolveRe does: failed / branch / leaf (or "miss": a leaf whose constraint-set
// translation the host has not provided yet). Outputs stay in registers; the callers (the
// round-based k_expand, the probe kernel) decide where children and leaves go.
// OR the dirty rows of every block word in `cm` (bit l = word q*64 + l changed) into the lane-striped
// dirty mask, four words per trip (independent reads).
template <int L>
__device__ __forceinline__ void mark_dirty_rows(const Img<L> &P, int rows_abs, int iw, int q, unsigned long long cm, int lane, uint32_t &dirtyw) {
    while (cm) {
        const int l0 = __ffsll((long long)cm) - 1;
        cm &= cm - 1;
        const int l1 = cm ? __ffsll((long long)cm) - 1 : l0;
        cm &= cm - 1;
        const int l2 = cm ? __ffsll((long long)cm) - 1 : l0;
        cm &= cm - 1;
        const int l3 = cm ? __ffsll((long long)cm) - 1 : l0;
        cm &= cm - 1;
        if (lane < iw) {
            const int rows = rows_abs + lane;
            const uint32_t r0 = (uint32_t)P.v(rows + (q * 64 + l0) * iw), r1 = (uint32_t)P.v(rows + (q * 64 + l1) * iw);
            const uint32_t r2 = (uint32_t)P.v(rows + (q * 64 + l2) * iw), r3 = (uint32_t)P.v(rows + (q * 64 + l3) * iw);
            dirtyw |= r0 | r1 | r2 | r3;
        }
    }
}

// Eager consistency of the X == next Y arcs (role of enforceNextConsistency, solveralgorithm.cpp:544-593):
// every block word intersects itself with its partners' domains (SetDesc::next_off), repeated until
// nothing changes (one pass for K = 2). Run after every change of the block, so that these arcs are
// never work items and their prunings do not cost a sweep of their own. Returns false on a wipe-out.
template <int DR, int L>
__device__ bool close_next(const Ctx &c, const Img<L> &P, const WaveEnv<DR> &S, Dom<DR> &dom, int lane, int *ldom, uint32_t &dirtyw) {
    const int base = S.next_abs;
    // what one entry allows: the partner's domain shifted into this word's value numbering
    auto allowed = [](uint32_t e, uint32_t partner) -> uint32_t {
        if (!e) return 0xffffffffu;
        const int sh = (int)((e >> 16) & 0xffu) - 64;
        const bool up = ((e >> 24) & 1u) ? sh >= 0 : sh < 0;  // side 0 (X): Y >> sh; side 1 (Y): X << sh
        const int a = sh >= 0 ? sh : -sh;
        return a >= 32 ? 0u : (up ? partner << a : partner >> a);
    };
    for (int pass = 0; pass < 64; pass++) {
        bool changed = false;
#pragma unroll
        for (int q = 0; q < DR; q++) {
            const int idx = q * 64 + lane;
            const bool in = idx < c.NK;
            const uint32_t e0 = S.e0[q];
            // gathers are executed by every lane (cross-lane reads need the source lanes active)
            uint32_t nd = dom.r[q] & allowed(e0, dom.gather(e0 ? (int)(e0 & 0xffffu) - 1 : 0));
            if (c.K > 2) {  // a word can sit on both sides of arcs only with more than two time points
                const uint32_t e1 = in ? (uint32_t)P.v(base + idx * 2 + 1) : 0u;
                if (__ballot(e1 != 0)) nd &= allowed(e1, dom.gather(e1 ? (int)(e1 & 0xffffu) - 1 : 0));
            }
            if (__ballot(in && nd == 0)) return false;
            const unsigned long long cm = __ballot(nd != dom.r[q]);
            if (cm) {
                if (nd != dom.r[q]) ldom[idx] = (int)nd;  // keep the sweeps' LDS copy current
                dom.r[q] = nd;
                mark_dirty_rows<L>(P, S.rows_abs, S.iw, q, cm, lane, dirtyw);
                changed = true;
            }
        }
        // K = 2: arcs only join a point-0 word with a point-1 word, one partner each -- both ends now
        // hold the same (shifted) set, a second pass cannot change anything
        if (!changed || c.K == 2) break;
    }
    return true;
}

// ---- several dirty wavefront-revised items at once (tuple-parallel batch) ------------------------------------
// A general revision of a tuple-bitmap constraint rarely needs the wavefront: digitinvader's `A3 == (I ne D0 and ...)`
// or `next D2 == if ...` have two or three open variables left and 10-30 tuples to look at, and ~20 of them are dirty
// per node -- one after the other they are ~850 instructions each for a dozen busy lanes. Here up to kBatchItems
// dirty items are revised TOGETHER against the same snapshot of the block (Jacobi among them, like the sweeps of the
// small items): lane k < 16 scans the scope of item k (open variables, product of their domain sizes, bitmap index
// of the singletons), the products are laid side by side on the 64 lanes (a prefix of the items whose products fit),
// every tuple lane decodes its own tuple of its own item and looks its window of variable 0's bits up (see
// revise_point), supports are collected with LDS atomics, and the item lanes AND them into the LDS copy of the block.
// Items with more than 64 tuples / kMaxLowVars open variables, or without a bitmap, are left to revise_point.
// Returns -1 on a wipe-out, 0 when the FIRST dirty item is not batchable (the caller revises it alone), else the
// number of items dealt with. `scr`: kBatchItems records of kBatchRec words; `clr`: 64 words.
constexpr int kBatchItems = 16, kBatchRec = 24, kBatchTuples = 256;  // tuples of one batch: looked at 64 at a time
static_assert(kBatchItems * kBatchRec <= kMaxLowVars * 64, "the batch records live in the lane-value scratch of the general revision");
// record of one batch item (16-byte aligned parts, so that a tuple lane fetches it with four wide LDS reads)
enum { BR_NOPEN = 0, BR_BASE = 1, BR_D0 = 2, BR_BITMAP = 3, BR_WPACK = 4 /* 2 words: block word of open variable q in byte q */, BR_SUP0 = 6,
       BR_META = 7 /* w0 | tuples << 8 | eligible << 20 | no-op << 21 */, BR_STRIDE = 8, BR_SUP = 16 };
static_assert(kMaxLowVars <= 6, "batch record layout");
// per lane: the bits of dirty word `lane` that belong to wavefront-revised items WITH a tuple bitmap and a scope of
// at most kBatchArity variables (the only ones revise_batch can take); worked out once per constraint set
template <int DR, int L>
__device__ __forceinline__ void load_bmmask(const Ctx &c, const Img<L> &P, int lane, WaveEnv<DR> &E) {
    // (row N*K + 1 of the set's dirty rows, laid down by the compiler: cset.cpp compile -- round 3 derived it here from the item
    // records, 32 dependent reads per lane at the start of every slot's first node)
    E.bmmask = lane < E.iw ? (uint32_t)P.v(E.rows_abs + (c.N * c.K + 1) * E.iw + lane) : 0u;
}
// reductions over the rows of 16 lanes (every lane ends up with its row's result): DPP rotations, no LDS
__device__ __forceinline__ int row_ror(int v, int by) {
    switch (by) {
        case 8: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);
    }
}
__device__ __forceinline__ int row_sum(int v) {
    v += row_ror(v, 8);
    v += row_ror(v, 4);
    v += row_ror(v, 2);
    v += row_ror(v, 1);
    return v;
}
__device__ __forceinline__ int row_max(int v) {
    v = max(v, row_ror(v, 8));
    v = max(v, row_ror(v, 4));
    v = max(v, row_ror(v, 2));
    v = max(v, row_ror(v, 1));
    return v;
}
__device__ __forceinline__ int row_prod_capped(int v) {  // factors <= 32; the product stops growing at 2^15
    v = min(v * row_ror(v, 8), 1 << 15);
    v = min(v * row_ror(v, 4), 1 << 15);
    v = min(v * row_ror(v, 2), 1 << 15);
    v = min(v * row_ror(v, 1), 1 << 15);
    return v;
}
template <int DR, int L>
__device__ int revise_batch(const Ctx &c, const Img<L> &G, WaveEnv<DR> &S, Dom<DR> &dom, int lane, uint32_t &dirtyw, int *scr, int *clr,
                            int *ldom, WaveStats &ws, bool &pm_changed, int &first_item) {
    // The item records, scopes and strides sit in one run of image sections (engine.hip upload_program): staged in
    // LDS as a whole or not at all. A partly staged image decides that ONCE here and reads through one pointer
    // (flat loads, issued together) instead of a staged-or-not branch around every single read.
    const uint32_t *ip = (L || c.o.code <= G.nlds) ? G.lds : G.p;
    // ---- A: slot k (lane k for now) takes the k-th dirty item that has a tuple bitmap
    const unsigned long long t_b0 = PHASE_NOW();
    (void)t_b0;
    const uint32_t dwv = lane < S.iw ? (dirtyw & S.bmmask) : 0u;
    int item = 0, ncand = 0;
    for (unsigned long long m = __ballot(dwv != 0); m && ncand < kBatchItems; m &= m - 1) {
        const int w = __ffsll((long long)m) - 1;
        const uint32_t word = rdlane(dwv, w);
        const int cnt = __popc(word), k = lane - ncand;
        if (k >= 0 && k < cnt) item = w * 32 + select_kth_fast(word, k);
        ncand += cnt;
    }
    ncand = min(ncand, kBatchItems);
    first_item = (int)rdlane((uint32_t)item, 0);
    // ---- B: the scopes are scanned FOUR ITEMS AT A TIME, one row of 16 lanes per item and one lane per scope
    // variable: open variables (domain not a singleton; variable 0 is never enumerated, see revise_point), the
    // number of tuples = product of their domain sizes, the bitmap index of the singletons -- row reductions. The
    // row leader (scope position 0 = variable 0) writes the item's record, the open variables their block word and
    // stride at their rank. Scanning stops once the tuples found fill the 64 lanes.
    const int j = lane & 15, row = lane >> 4;
    int scanned = 0, tuples_found = 0;
    for (int p0 = 0; p0 < ncand && tuples_found < kBatchTuples; p0 += 4) {
        const int slot = p0 + row;
        const bool vs = slot < ncand;
        const int it_ = __shfl(item, slot & 63, 64);
        const int it = vs ? it_ : first_item;
        const int ib = S.items_abs + it * (int)(sizeof(ItemDesc) / 4);
        const int point = (int)ip[ib + (int)(offsetof(ItemDesc, point) / 4)];
        const int ar = (int)ip[ib + (int)(offsetof(ItemDesc, arity) / 4)];
        const int scope_off = (int)ip[ib + (int)(offsetof(ItemDesc, idx) / 4)];
        const int bm_off = (int)ip[ib + (int)(offsetof(ItemDesc, idx) / 4) + 1];
        const int stride_off = (int)ip[ib + (int)(offsetof(ItemDesc, idx) / 4) + 2];
        const int nforb = (int)ip[ib + (int)(offsetof(ItemDesc, idx) / 4) + 3];
        const bool in = vs && j < ar;
        const int jj = in ? j : 0;
        const int var = (int)ip[c.o.scope + scope_off + jj];
        const int st = (int)ip[c.o.strides + stride_off + jj];
        const int w = point * c.N + var;
        const uint32_t D = (uint32_t)ldom[w];
        const int n = in ? __popc(D) : 1;
        if (__ballot(in && n == 0)) return -1;
        const bool open = in && n > 1 && j != 0;
        const unsigned rowbits = (unsigned)(__ballot(open) >> (16 * row)) & 0xffffu;
        const int nopen = __popc(rowbits), rank = __popc(rowbits & ((1u << j) - 1u));
        const int prod = row_prod_capped(open ? n : 1);
        const int base = row_sum((in && n == 1 && j != 0) ? (__ffs((int)D) - 1) * st : 0);
        // constraints with few violating tuples in all: the revision cannot prune unless the product of the other
        // domains fits into the forbidden set (see revise_point)
        bool noop = false;
        if (__ballot(vs && nforb >= 0)) {
            const int largest = row_max(n);
            const int n0 = __shfl(n, lane & 48, 64);
            const int nopen_all = nopen + (n0 > 1 ? 1 : 0);
            const long long pall = (long long)prod * (long long)n0;  // (prod capped at 2^15: still far above nforb * largest <= 2048)
            noop = nforb >= 0 && ((nforb <= 1 && nopen_all >= 2) || nforb == 0 || nopen_all >= 12 || (1u << min(nopen_all, 31)) > (unsigned)nforb * 32u ||
                                  (nopen_all >= 2 && pall > (long long)nforb * (long long)largest));
        }
        const bool elig = !noop && nopen <= kMaxLowVars && prod <= kBatchTuples;
        const int rec = (slot & (kBatchItems - 1)) * kBatchRec;
        if (vs && j == 0) {  // the leader holds variable 0: D, w are its domain and block word
            *(uint4 *)&scr[rec + BR_NOPEN] = make_uint4((uint32_t)nopen, (uint32_t)base, D, (uint32_t)bm_off);
            scr[rec + BR_SUP0] = 0;
            scr[rec + BR_META] = w | ((elig ? prod : 0) << 8) | ((elig ? 1 : 0) << 20) | ((noop ? 1 : 0) << 21);
            *(uint4 *)&scr[rec + BR_SUP] = make_uint4(0u, 0u, 0u, 0u);
            *(uint2 *)&scr[rec + BR_SUP + 4] = make_uint2(0u, 0u);
        }
        if (open && rank < kMaxLowVars) {
            ((unsigned char *)&scr[rec + BR_WPACK])[rank] = (unsigned char)w;
            scr[rec + BR_STRIDE + rank] = st;
        }
        STCSP_REJOIN();
        const unsigned long long lead = __ballot(vs && j == 0 && elig);
        for (unsigned long long m = lead; m; m &= m - 1) tuples_found += (int)rdlane((uint32_t)prod, __ffsll((long long)m) - 1);
        scanned = min(p0 + 4, ncand);
        if (p0 == 0 && !(__ballot(elig || noop) & 1ull)) {  // the first of them needs revise_point (the caller revises first_item)
#ifdef STCSP_PHASES
            ws.cyc_batch_ab += PHASE_NOW() - t_b0;
            ws.batch_refused++;
#endif
            return 0;
        }
    }
    clr[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- C: item lanes (lane k = slot k) read their records back and lay the products side by side (a prefix of the eligible items)
    const bool cand = lane < scanned;
    const int rec = (lane & (kBatchItems - 1)) * kBatchRec;
    const uint4 my = *(const uint4 *)&scr[rec + BR_NOPEN];
    const uint2 mywp = *(const uint2 *)&scr[rec + BR_WPACK];
    const int meta = scr[rec + BR_META];
    const int nopen = (int)my.x, w0 = meta & 255;
    const uint32_t D0 = my.z, wp0 = mywp.x, wp1 = mywp.y;
    const bool elig = cand && ((meta >> 20) & 1), noop = cand && ((meta >> 21) & 1);
#ifdef STCSP_PHASES
    const unsigned long long t_b1 = PHASE_NOW();
    ws.cyc_batch_ab += t_b1 - t_b0;
#endif
    const int cnt = elig ? ((meta >> 8) & 4095) : 0;
    const int incl = wave_scan_add(cnt), excl = incl - cnt;
    const bool inb = elig && incl <= kBatchTuples;
    const unsigned long long inbm = __ballot(inb);
    const int nb = __popcll(inbm);
    const int T = inbm ? (int)rdlane((uint32_t)incl, 63 - __clzll((long long)inbm)) : 0;
    // ---- D: tuple lanes, 64 tuples per trip, find their item (the last item lane whose first tuple is <= theirs) and decode their tuple
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const bool tl = t < T;
        int k = 0;
#pragma unroll
        for (int step = kBatchItems / 2; step >= 1; step >>= 1) {
            const int ck = k + step;
            const int e = __shfl(excl, ck & 63, 64);  // (every lane: cross-lane reads need the source lanes active)
            if (ck < kBatchItems && e <= t) k = ck;
        }
        int u = t - __shfl(excl, k, 64);
        const int rk = k * kBatchRec;
        const uint4 hd = *(const uint4 *)&scr[rk + BR_NOPEN];
        const uint2 wp = *(const uint2 *)&scr[rk + BR_WPACK];
        const uint4 s03 = *(const uint4 *)&scr[rk + BR_STRIDE];
        const uint2 s45 = *(const uint2 *)&scr[rk + BR_STRIDE + 4];
        const int nop = tl ? (int)hd.x : 0;
        const int wq[6] = {(int)(wp.x & 255u), (int)((wp.x >> 8) & 255u), (int)((wp.x >> 16) & 255u), (int)(wp.x >> 24), (int)(wp.y & 255u), (int)((wp.y >> 8) & 255u)};
        const int sq[6] = {(int)s03.x, (int)s03.y, (int)s03.z, (int)s03.w, (int)s45.x, (int)s45.y};
        uint32_t Dq[6];
#pragma unroll
        for (int q = 0; q < 6; q++) Dq[q] = q < nop ? (uint32_t)ldom[wq[q]] : 1u;
        int idx = (int)hd.y;
        uint32_t pack = 0;  // value bit of open variable q in bits [5q, 5q + 5)
#pragma unroll
        for (int q = 0; q < 6; q++) {
            if (__ballot(q < nop)) {
                const uint32_t D = Dq[q];
                const int n = __popc(D);
                const int qd = small_div(u, n);  // (u < kBatchTuples: the reciprocal is still exact)
                const int bit = select_kth_fast(D, u - qd * n);
                if (q < nop) {
                    u = qd;
                    idx += bit * sq[q];
                    pack |= (uint32_t)bit << (5 * q);
                }
            }
        }
        // ---- E: the window of variable 0's bits at this tuple; a satisfied tuple supports all of its values
        if (tl) {
            const uint32_t D0k = hd.z;
            const int tw = c.o.tables + (int)hd.w + (idx >> 5), sh = idx & 31;
            const uint32_t wlo = (uint32_t)G.vc(tw);
            const uint32_t whi = (sh + (31 - __clz((int)D0k)) >= 32) ? (uint32_t)G.vc(tw + 1) : 0u;
            const uint32_t sup0 = (uint32_t)((((unsigned long long)whi << 32) | wlo) >> sh) & D0k;
            if (sup0) {
                atomicOr((unsigned *)&scr[rk + BR_SUP0], sup0);
#pragma unroll
                for (int q = 0; q < 6; q++)
                    if (q < nop) atomicOr((unsigned *)&scr[rk + BR_SUP + q], 1u << ((pack >> (5 * q)) & 31u));
            }
        }
        STCSP_REJOIN();
    }
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef STCSP_PHASES
    const unsigned long long t_b2 = PHASE_NOW();
    ws.cyc_batch_de += t_b2 - t_b1;
#endif
    // ---- F: item lanes intersect the block with the supports
    const int wo[6] = {(int)(wp0 & 255u), (int)((wp0 >> 8) & 255u), (int)((wp0 >> 16) & 255u), (int)(wp0 >> 24), (int)(wp1 & 255u), (int)((wp1 >> 8) & 255u)};
    bool nosup = false;
    uint32_t s0 = 0, sup[6] = {0, 0, 0, 0, 0, 0};
    if (inb) {
        const uint4 a = *(const uint4 *)&scr[rec + BR_SUP];
        const uint2 b = *(const uint2 *)&scr[rec + BR_SUP + 4];
        s0 = (uint32_t)scr[rec + BR_SUP0];
        sup[0] = a.x, sup[1] = a.y, sup[2] = a.z, sup[3] = a.w, sup[4] = b.x, sup[5] = b.y;
        nosup = s0 == 0;  // no satisfying tuple at all
        if (s0 != D0) atomicAnd((unsigned *)&ldom[w0], s0);
#pragma unroll
        for (int q = 0; q < 6; q++)
            if (q < nopen) atomicAnd((unsigned *)&ldom[wo[q]], sup[q]);
    }
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    S.n_revs += (unsigned)nb;
    S.n_wave_revs += (unsigned)nb;
    S.n_evals += (unsigned)T;
#ifdef STCSP_PHASES
    ws.batches++;
    ws.batch_items += (unsigned)nb;
    ws.batch_tuples += (unsigned)T;
#endif
    if (__ballot(nosup)) return -1;
    // ---- G: read the intersection back; changed words re-dirty the items that read them
    bool empty = false;
#pragma unroll
    for (int q = 0; q < DR; q++) {
        const int ix = q * 64 + lane;
        const uint32_t nd = ix < c.NK ? (uint32_t)ldom[ix] : dom.r[q];
        if (__ballot(ix < c.NK && nd == 0)) empty = true;
        const unsigned long long cm = __ballot(nd != dom.r[q]);
        dom.r[q] = nd;
        mark_dirty_rows<L>(G, S.rows_abs, S.iw, q, cm, lane, dirtyw);
        pm_changed = pm_changed || (cm & S.pm[q]) != 0;
    }
    if (empty) return -1;
    // an item whose open variables ended up with exactly the supports IT computed is at its fixpoint (its satisfying
    // tuples consist of supported values only); one that lost more through another item of the batch stays dirty
    bool selfok = noop;
    if (inb) {
        bool ok = (uint32_t)ldom[w0] == s0;
#pragma unroll
        for (int q = 0; q < 6; q++) ok = ok && (q >= nopen || (uint32_t)ldom[wo[q]] == sup[q]);
        selfok = ok;
    }
    if (selfok) atomicOr((unsigned *)&clr[item >> 5], 1u << (item & 31));
    STCSP_REJOIN();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dirtyw &= ~(uint32_t)clr[lane];
#ifdef STCSP_PHASES
    ws.cyc_batch += PHASE_NOW() - t_b0;
#endif
    return nb + __popcll(__ballot(noop));
}

template <int DR, int L, bool CS, bool LITE>
__device__ int process_node(const Ctx &c, const Img<L> &P, int lane, int *lds_vals, int *lds_stk, int *ldom, Dom<DR> &dom,
                            const NodeHdr &hd, int gw, WaveEnv<DR> &S, BranchOut &bo, LeafOut<DR> &lo) {
    const int set = hd.set;
    const uint32_t seed = hd.seed, expire = hd.expire;
    if (set != S.set) {
        load_env<DR, L>(c, P, set, lane, S);
        if constexpr (!LITE) load_bmmask<DR, L>(c, P, lane, S);
    }

    // ---- propagate to the GAC fixpoint (role of generalisedArcConsistent, :617-706). Work items
    // are (constraint, time point) pairs; the dirty mask is lane-striped (lane w holds word w).
    // Items [0, nsmall) -- X == next Y arcs, until checks and small extensional point constraints --
    // are revised ONE ITEM PER LANE against a snapshot of the block, their prunings ANDed together
    // through an LDS copy (a Jacobi sweep); the remaining items are revised by the whole
    // wavefront one at a time. Monotone propagators: any fair order reaches the same fixpoint.
    WaveStats ws;
    uint32_t dirtyw = 0;
    if (lane < S.iw) {
        if (seed == 0) {
            int left = S.nitems - lane * 32;
            dirtyw = left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
        } else if (seed != kSeedNone) {
            dirtyw = (uint32_t)P.v(S.rows_abs + (int)(seed - 1) * S.iw + lane);  // word (0, seed var)
        }
    }
    const uint32_t smallmask = S.smallmask;
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
    // the X == next Y arcs that are not items: the parent kept them consistent, the bisection (or the
    // time shift of a fresh state) may have broken them. S.pm[q]: block words (bit l = word q*64 + l) that
    // have an eager partner; a closure is only due when such a word changed (scalar tests from here on)
    const unsigned long long(&pm)[DR] = S.pm;
    bool need_close = false;
    // seed N*K + 1 ("fresh", dev_kernels.hpp STCSP_FRESH_SEED): a new state under the constraint set of the state its leaf belonged
    // to. Its points 0 .. K-2 are the leaf's points 1 .. K-1, where the same items were at their fixpoint on the same domains;
    // only what reads the fresh point K-1 (and `first` / until items, which see point 0 for the first time) can have work: the
    // extra row N*K of the set's dirty rows (cset.cpp compile) -- read by the seed branch above like any other row.
    if (seed == 0 || seed == (uint32_t)(c.N * c.K + 1)) {
        need_close = S.next_abs >= 0;  // fresh state: the time shift may have broken any arc
    } else if (seed != kSeedNone) {
        const int w = (int)seed - 1;  // the parent bisected time-0 word w and was at its fixpoint otherwise
#pragma unroll
        for (int q = 0; q < DR; q++)
            if ((w >> 6) == q) need_close = (pm[q] >> (w & 63)) & 1ull;
    }
    while (consistent) {
        if (need_close) {  // the only call site: after any change of the block, before anything else is revised
            need_close = false;
            const unsigned long long t_cl = PHASE_NOW();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            consistent = close_next<DR, L>(c, P, S, dom, lane, ldom, dirtyw);
            ws.cyc_close += PHASE_NOW() - t_cl;
            if (!consistent) break;
        }
        if (__ballot((dirtyw & smallmask) != 0)) {
            const unsigned long long t_sw = PHASE_NOW();
            // ---- lane-parallel sweep over the dirty small items
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            bool lfail = false;
            S.n_sweeps++;
            // Sets with many small items (the synthetic 64 x 32 family: 1,216) have their dirty items
            // scattered over the item index space -- ~120 dirty per sweep in 19 blocks of 64, six
            // busy lanes per pass. There the dirty bits are compacted first: lane k of pass t takes
            // the (64 t + k)-th dirty item (prefix counts of the per-lane dirty words, a 6-step
            // search over them with ds_bpermute), so a sweep needs ceil(dirty / 64) passes.
            const bool compact = CS && S.nsmall > kCompactSweepItems;  // CS: separate kernel, the search costs ~15 VGPRs
            uint32_t dsm = 0;
            int excl = 0, total_dirty = 0;
            if (compact) {
                dsm = lane < S.iw ? (dirtyw & smallmask) : 0u;
                const int cnt = __popc(dsm);
                const int incl = wave_scan_add(cnt);
                excl = incl - cnt;
                total_dirty = (int)rdlane((uint32_t)incl, 63);
            }
            const int npass = compact ? (total_dirty + 63) >> 6 : (S.nsmall + 63) >> 6;
            for (int t = 0; t < npass; t++) {
                int item;
                bool isd;
                if (compact) {
                    const int k = t * 64 + lane;
                    int w = 0;  // the last lane whose exclusive prefix is <= k owns the k-th dirty bit
#pragma unroll
                    for (int step = 32; step >= 1; step >>= 1) {
                        const int cand = w + step;
                        const int e = __shfl(excl, cand & 63, 64);
                        if (cand < 64 && e <= k) w = cand;
                    }
                    const uint32_t word = (uint32_t)__shfl((int)dsm, w, 64);
                    const int first = __shfl(excl, w, 64);
                    isd = k < total_dirty;
                    item = isd ? w * 32 + select_kth_fast(word, k - first) : 0;
                } else {
                    item = t * 64 + lane;
                    const uint32_t dw0 = rdlane(dirtyw, (2 * t) & 63), dw1 = rdlane(dirtyw, (2 * t + 1) & 63);
                    const uint32_t dw = lane < 32 ? dw0 : dw1;
                    isd = item < S.nsmall && ((dw >> (item & 31)) & 1u);
                }
                unsigned long long dmask = __ballot(isd);
                if (!dmask) continue;
                // the lane's item, unpacked from its 16-byte sweep record (device_types.hpp)
                const uint4 sw = P.v4(S.sweep_abs + (isd ? item : 0) * 4);
                struct {
                    int idx[4], type, arity, r1, r2, aux, toff;
                } it;
                it.idx[0] = (int)(sw.x & 255u);
                it.idx[1] = (int)((sw.x >> 8) & 255u);
                it.idx[2] = (int)((sw.x >> 16) & 255u);
                it.idx[3] = (int)(sw.x >> 24);
                it.type = (int)(sw.y & 3u);
                it.arity = (int)((sw.y >> 2) & 7u);
                it.r1 = (int)((sw.y >> 5) & 63u);
                it.r2 = (int)((sw.y >> 11) & 63u);
                it.aux = (int)sw.y >> 17;
                it.toff = (int)sw.z;
                // the item's domains: a gather from the registers when the block is one register per lane (one ds_bpermute
                // each); wider blocks read the LDS copy instead (it equals `dom` between sweeps; a pass may see the
                // prunings of the passes before it -- domains only shrink, the fixpoint is the same), which is one LDS
                // read where a gather costs DR of them plus the selects
                uint32_t D0, D1, D2, D3;
                if constexpr (DR == 1) {
                    // gathers are executed by every lane (cross-lane reads need the source lanes active)
                    D0 = dom.gather(it.idx[0]), D1 = dom.gather(it.idx[1]);
                    D2 = dom.gather(it.idx[2]), D3 = dom.gather(it.idx[3]);
                } else {
                    D0 = (uint32_t)ldom[it.idx[0]], D1 = (uint32_t)ldom[it.idx[1]];
                    D2 = (uint32_t)ldom[it.idx[2]], D3 = (uint32_t)ldom[it.idx[3]];
                }
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
                    }
#ifndef STCSP_X_NOTAB
                    else {
                        // small extensional constraint: one row of allowed word-variable values per
                        // tuple of the other (<= 3) variables; scan the rows of the current product
                        if (it.arity < 2) D1 = 1u;
                        if (it.arity < 3) D2 = 1u;
                        if (it.arity < 4) D3 = 1u;
                        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                        const int tab = c.o.stables + it.toff;  // (a section of its own: staged in LDS whenever it is small)
                        unsigned nev = 0;  // table rows this lane looks at
                        const int r1p = (it.r1 + 3) & ~3;  // small_row_stride: rows come four per 128-bit read
                        for (uint32_t m3 = D3; m3; m3 &= m3 - 1) {
                            const int b3 = __ffs((int)m3) - 1;
                            for (uint32_t m2 = D2; m2; m2 &= m2 - 1) {
                                const int b2 = __ffs((int)m2) - 1;
                                const int base = tab + r1p * (b2 + it.r2 * b3);
                                uint32_t any = 0;
                                // NQ 128-bit reads (4 rows each) in flight per trip; the wide-block kernels
                                // (DR = 4: big models, tables in HBM) have the registers for four
                                constexpr int NQ = DR >= 4 ? 4 : (DR == 1 ? 1 : 2);  // DR = 1: one read at a time keeps the kernel at 80 VGPRs (6 wavefronts/SIMD)
                                for (int c4 = 0; c4 < r1p; c4 += 4 * NQ) {
                                    const uint32_t nib = (D1 >> c4) & ((1u << (4 * NQ)) - 1u);
                                    if (!nib) continue;
                                    uint4 rr[NQ];
#pragma unroll
                                    for (int g = 0; g < NQ; g++) {  // a non-empty group of D1 bits implies c4 + 4 g < r1p
                                        rr[g] = make_uint4(0u, 0u, 0u, 0u);
#ifdef STCSP_PHASES
                                        // (hipcc 7.2 fails on the instrumented build with a selected-pointer 128-bit read here --
                                        // "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" -- the diagnostic build reads the
                                        // global copy of the section under the partly-staged kernels)
                                        if (g == 0 || ((nib >> (4 * g)) & 15u)) rr[g] = L == 0 ? P.v4c(base + c4 + 4 * g) : P.v4(base + c4 + 4 * g);
#else
                                        if (g == 0 || ((nib >> (4 * g)) & 15u)) rr[g] = P.v4(base + c4 + 4 * g);
#endif
                                    }
                                    uint32_t got = 0;
#pragma unroll
                                    for (int g = 0; g < NQ; g++) {
                                        if (g > 0 && !((nib >> (4 * g)) & 15u)) continue;
                                        const uint32_t rows[4] = {rr[g].x, rr[g].y, rr[g].z, rr[g].w};
#pragma unroll
                                        for (int k = 0; k < 4; k++) {
                                            const uint32_t r = ((nib >> (4 * g + k)) & 1u) ? (rows[k] & D0) : 0u;
                                            s0 |= r;
                                            got |= (r ? 1u : 0u) << (4 * g + k);
                                        }
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
                        S.rows_seen += nev;  // statistics
                        if (s0 == 0) lfail = true;
                        if (s0 != D0) atomicAnd((unsigned *)&ldom[it.idx[0]], s0);
                        if (it.arity > 1 && s1 != D1) atomicAnd((unsigned *)&ldom[it.idx[1]], s1);
                        if (it.arity > 2 && s2 != D2) atomicAnd((unsigned *)&ldom[it.idx[2]], s2);
                        if (it.arity > 3 && s3 != D3) atomicAnd((unsigned *)&ldom[it.idx[3]], s3);
                    }
#endif
                }
                STCSP_REJOIN();
                S.n_revs += (unsigned)__popcll(dmask);  // (after the lane-predicated part: keeps the counter wave-uniform)
            }
            dirtyw &= ~smallmask;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (__ballot(lfail)) {
                consistent = false;
                break;
            }
            // read the intersection back; every changed word re-dirties the items that read it
            bool swept_change = false;
#pragma unroll
            for (int q = 0; q < DR; q++) {
                int idx = q * 64 + lane;
                uint32_t nd = idx < c.NK ? (uint32_t)ldom[idx] : dom.r[q];
                if (__ballot(idx < c.NK && nd == 0)) consistent = false;
                unsigned long long cm = __ballot(nd != dom.r[q]);
                dom.r[q] = nd;
                mark_dirty_rows<L>(P, S.rows_abs, S.iw, q, cm, lane, dirtyw);
                swept_change = swept_change || (cm & pm[q]) != 0;
            }
            need_close = swept_change;  // a word with an eager partner changed
            if (++guard > (1u << 20)) {
                S.err = max(S.err, (unsigned)ERR_WATCHDOG);
                consistent = false;
            }
            ws.cyc_sweep += PHASE_NOW() - t_sw;
            continue;
        }
        const unsigned long long t_wv = PHASE_NOW();
        unsigned long long dm = __ballot(dirtyw != 0);
        if (!dm) break;
        int forced = -1;  // the item revise_batch wants revised alone
        if constexpr (!LITE) {
            if (__ballot(lane < S.iw && (dirtyw & S.bmmask) != 0)) {
                bool pm_changed = false;
                int first = 0;
                const int br = revise_batch<DR, L>(c, P, S, dom, lane, dirtyw, lds_vals, lds_stk, ldom, ws, pm_changed, first);
                if (br != 0) {
                    consistent = br > 0;
                    need_close = pm_changed && S.next_abs >= 0;
                    ws.cyc_wave += PHASE_NOW() - t_wv;
                    if (++guard > (1u << 20)) {
                        S.err = max(S.err, (unsigned)ERR_WATCHDOG);
                        consistent = false;
                    }
                    continue;
                }
                forced = first;
            }
        }
        int wl = __ffsll((long long)dm) - 1;
        uint32_t word = rdlane(dirtyw, wl);
        int b = __ffs((int)word) - 1;
        if (forced >= 0) {
            wl = forced >> 5;
            b = forced & 31;
        }
        int item = wl * 32 + b;
        if (lane == wl) dirtyw &= ~(1u << b);
        // the item record holds the point and the constraint's descriptor: one per-lane read, broadcast by readlane
        const int ibase = S.items_abs + item * (int)(sizeof(ItemDesc) / 4);
        const uint32_t irec = lane < (int)(sizeof(ItemDesc) / 4) ? (uint32_t)P.v(ibase + lane) : 0u;
#define STCSP_ID(f) (int) rdlane(irec, (int)(offsetof(ItemDesc, f) / 4))
        const int ipoint = STCSP_ID(point);
        ConDesc C;
        C.scope_len = STCSP_ID(arity);
        C.scope_off = STCSP_ID(idx[0]);
        C.bitmap_off = STCSP_ID(idx[1]);
        C.stride_off = STCSP_ID(idx[2]);
        C.n_forbidden = STCSP_ID(idx[3]);
        C.code_off = STCSP_ID(toff);
        C.uses_valid = STCSP_ID(r1);
        C.code_len = STCSP_ID(r2);
#undef STCSP_ID
        bool pruned = false;
        consistent = revise_point<DR, L, LITE>(c, P, S, C, item, ipoint, dom, lane, dirtyw, lds_vals, lds_stk, ldom, ws, pruned);
        need_close = pruned && S.next_abs >= 0;
        ws.cyc_wave += PHASE_NOW() - t_wv;
        if (++guard > (1u << 20)) {
            S.err = max(S.err, (unsigned)ERR_WATCHDOG);
            consistent = false;
        }
    }
    S.n_nodes++;
#ifdef STCSP_PHASES
    if (lane == 0) {
        add_stats(c, gw, ST_CYC_SWEEP, ws.cyc_sweep);
        add_stats(c, gw, ST_CYC_WAVE, ws.cyc_wave);
        add_stats(c, gw, ST_CYC_RV_SETUP, ws.cyc_rv_setup);
        add_stats(c, gw, ST_CYC_RV_LOOP, ws.cyc_rv_loop);
        add_stats(c, gw, ST_CYC_RV_WB, ws.cyc_rv_wb);
        add_stats(c, gw, ST_CYC_CLOSE, ws.cyc_close);
        add_stats(c, gw, ST_RV_BLOCKS, ws.rv_blocks);
        add_stats(c, gw, ST_RV_OPEN, ws.rv_open);
        add_stats(c, gw, ST_RV_LANES, ws.rv_lanes);
        add_stats(c, gw, ST_CYC_RV_DIGITS, ws.cyc_rv_digits);
        add_stats(c, gw, ST_CYC_RV_EVAL_BITMAP, ws.cyc_rv_eval_bitmap);
        add_stats(c, gw, ST_CYC_RV_EVAL_CODE, ws.cyc_rv_eval_code);
        add_stats(c, gw, ST_CYC_RV_SUPPORT, ws.cyc_rv_support);
        add_stats(c, gw, ST_RV_BLOCKS_CODE, ws.rv_blocks_code);
        add_stats(c, gw, ST_BATCHES, ws.batches);
        add_stats(c, gw, ST_BATCH_ITEMS, ws.batch_items);
        add_stats(c, gw, ST_BATCH_REFUSED, ws.batch_refused);
        add_stats(c, gw, ST_BATCH_TUPLES, ws.batch_tuples);
        add_stats(c, gw, ST_CYC_BATCH, ws.cyc_batch);
        add_stats(c, gw, ST_CYC_BATCH_AB, ws.cyc_batch_ab);
        add_stats(c, gw, ST_CYC_BATCH_DE, ws.cyc_batch_de);
        if (seed == 0 || seed == (uint32_t)(c.N * c.K + 1)) {  // the first node of a state
            add_stats(c, gw, ST_ROOTS, 1);
            add_stats(c, gw, ST_CYC_ROOT_SWEEP, ws.cyc_sweep);
            add_stats(c, gw, ST_CYC_ROOT_WAVE, ws.cyc_wave);
            add_stats(c, gw, ST_ROOT_BATCHES, ws.batches);
            add_stats(c, gw, ST_ROOT_CYC_BATCH, ws.cyc_batch);
            add_stats(c, gw, ST_ROOT_RV_BLOCKS, ws.rv_blocks);
            add_stats(c, gw, ST_ROOT_REFUSED, ws.batch_refused);
        }
        if (seed == 0 || seed == (uint32_t)(c.N * c.K + 1)) {
            add_stats(c, gw, ST_ROOTS, 1);
            add_stats(c, gw, ST_CYC_ROOT_SWEEP, ws.cyc_sweep);
            add_stats(c, gw, ST_CYC_ROOT_WAVE, ws.cyc_wave);
        }
    }
#endif
    if (!consistent) {
        S.n_fails++;
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

    const unsigned long long t_leaf = PHASE_NOW();
    (void)t_leaf;
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
        if (S.trans_count < 0) {
            // the set's transitions are a table over the captured tuples (translated ahead of need): index = mixed radix of the
            // captured values' bit positions
            const int fs = lane < S.nfirst ? P.v(c.o.fstrides + S.first_off + lane) : 0;
            const int idx = wave_sum(lane < S.nfirst ? (__ffs((int)fd) - 1) * fs : 0);
            next_set = kload(c.tdirect, S.trans_begin + idx);
        }
        for (int t = 0; t < S.trans_count && next_set < 0; t++) {
            const int voff = P.u(c.o.trans + (S.trans_begin + t) * 2);
            // every lane reads (no lane-predicated branch inside this loop: see STCSP_REJOIN)
            const int tv = P.v(c.o.transvals + voff + (lane < S.nfirst ? lane : 0));
            const bool ne = lane < S.nfirst && tv != fval;
            if (!__ballot(ne)) next_set = P.u(c.o.trans + (S.trans_begin + t) * 2 + 1);
        }
        if (next_set < 0) {
            // unknown transition: park the node again and tell the host which translation is needed
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
    // lane j hashes key word j, the terms are XORed across the wavefront (key_hash, device_types.hpp)
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
            nb = (p + 1 < c.K) ? shifted : (uint32_t)P.v(c.o.var_init + next_set * c.o.init_stride + v);
        }
        lo.nblk[q] = nb;
    }
    S.n_leaves++;
#ifdef STCSP_PHASES
    if (lane == 0) add_stats(c, gw, ST_CYC_LEAF, PHASE_NOW() - t_leaf);
#endif
    return OC_LEAF;
}

}  // namespace dev
}  // namespace stcsp
