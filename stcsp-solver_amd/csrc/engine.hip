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
#include <map>
#include <memory>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "cset.hpp"
#include "device_types.hpp"
#include "okfix.hpp"
#include "stcsp_engine.h"

using namespace stcsp;

#include "dev_kernels.hpp"
#include "dev_postproc.hpp"

using namespace stcsp::dev;

namespace {
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
    int chunk_r0 = 0;  // ... as configured (an automatic batch may shrink chunk_r during a solve)
    bool auto_batch = true;
    size_t arena_soft_words = 0;
    // expansions per slot and launch (expand_node): chain_small while a round has <= chain_thresh
    // nodes, else chain_big; a slot stops chaining after chain_heavy cycles in one launch (measured
    // optimum 300-500 k on digitinvader5/7/9, flat on partialorder). chain_small: 16 (round 3: 8) under the general kernels
    // (expensive nodes: fewer, longer rounds -- digitinvader9 27.2 -> 23.4 ms), 4 under the LITE ones
    // (partialorder_12/14/16 lose 3-10 % with 8), 1 when the program has a conditional constraint over more than 2^20 tuples (a case analysis over a dozen variables: the juggling family, round 4: a few hundred to a few thousand expensive nodes in a search 15 node levels deep -- a slot that walks a subtree depth-first holds its round, and every idle wavefront, for the whole walk: level-synchronous rounds are twice as fast, juggling_b6_f6_nosym 1.9 -> 0.9 ms; digitinvader, 260 levels deep with a dear first node per state, is twice as SLOW that way), 2 when a constraint is interpreted (round 3's juggling _nosym: a few thousand
    // uniformly expensive nodes, longer chains only serialise them: juggling_b6_f6_nosym 2.5 -> 1.8 ms); chain_big 4 since the sibling stack (2 before: partialorder_18 81 -> 74.5 ms,
    // synthetic 64 x 32 71 -> 76 M nodes/s; 6 and 8 are slower again). STCSP_CHAIN_SMALL / _BIG / _THRESH / _HEAVY override.
    bool chain_small_auto = true;
    int chain_small = 4, chain_big = 4, chain_thresh = 65536, chain_heavy = 400000;
    int max_blocks = 256 * 4;  // k_expand grid (workgroups): set from the occupancy query

    DevBuf<int> d_arr_data, d_arr_off, d_code, d_miss, d_tdirect;
    std::vector<int32_t> tdirect_dev;  // host copy of what d_tdirect holds (upload_program sends differences only)
    DevBuf<uint32_t> d_state_keys, d_ctl, d_edges, d_arena, d_cand, d_pack, d_img;
    bool img_in_lds = false;
    DevBuf<unsigned long long> d_slots, d_stats;
    uint32_t *h_ctl = nullptr;  // pinned
    int *h_miss = nullptr;      // pinned
    uint32_t cand_cap = 0;
    DevBuf<Plan> d_plan;
    bool dbg_rounds = false;  // STCSP_DEBUG=2: per-launch log (with STCSP_BURST=1 and STCSP_F_PROFILE)
    std::vector<unsigned long long> dbg_nodes;
    std::vector<long long> dbg_open;
    bool compact_sweeps = false;  // some set has more than kCompactSweepItems small items: k_expand<.., .., true, ..>
    bool lite = false;            // no constraint needs the general wavefront revision: k_expand<.., .., .., true>
    bool big = false;             // 1024-thread workgroups around one LDS copy of a LITE program: k_expand<.., true, .., true, true>
    int prefix_need = 0;          // image words that must be staged for the L = 2 kernels (0: not applicable)
    bool prefix_complete = false; // ... and they are: general program, everything but cons / tables in the staged prefix
    bool interpreted = false;     // some wavefront-revised constraint has no tuple bitmap (postfix interpreter: uniformly expensive nodes)
    bool wide_conditional = false;  // some conditional constraint spans more than kWideConditional tuples (the juggling family's `A == if B0 eq 1 then next B0 else if ...`)
    bool host_view_fresh = false;  // h_ctl / h_plan were read after the last device work (expand_local -> commit)
    bool packed = false;      // the outboxes of the last expand_local are packed (pack_ptr / pack_count valid)
    int64_t step_max_rounds = 0, step_min_open = 0;  // expand_local budget (set_expand_budget): 0 = run the frontier dry
    DevBuf<uint32_t> d_xfer;                          // donated open nodes (transfer records)
    DevBuf<uint32_t> d_recv_cand, d_recv_nodes;       // stcsp_engine_solve_sharded: candidate / transfer records received from the peers
    std::vector<DevSegment> h_stack;
    std::vector<uint32_t *> pack_ptr;
    std::vector<int64_t> pack_count;
    bool sharded = false;    // candidate / commit pipeline (world > 1 or STCSP_F_STEPPED)
    DevBuf<Ctx> d_ctx;       // device copy of ctx for k_expand (re-uploaded before a burst)
    Ctx *h_ctx = nullptr;    // pinned staging copy
    Plan *h_plan = nullptr;  // pinned mirror of the plan header (everything before the stack)
    int burst = 8;           // rounds enqueued per host synchronisation (32 for unsharded solves: a launch past the end costs ~3 us, a synchronisation ~25)
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
    uint8_t *h_fail = nullptr;
    size_t h_edge_cap = 0, h_state_cap = 0;
    // device post-processing (dev_postproc.hpp) over the compacted export
    bool exp_on_device = false;  // d_osrc/d_odst/d_oval/d_state_keys describe the last exported automaton
    size_t exp_edges = 0;
    DevBuf<uint8_t> d_pvalid, d_pfinal, d_palive, d_pnodeok;
    DevBuf<uint32_t> d_pcover;
    std::vector<uint8_t> p_valid, p_final, p_alive;

    ~stcsp_engine() {
        // the device writes several of the pinned buffers freed below (progress mirror, streamed result arrays) from
        // launches and copies that may still be in flight after an error return: drain every stream first
        (void)hipSetDevice(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (xstream) (void)hipStreamSynchronize(xstream);
        if (xstream2) (void)hipStreamSynchronize(xstream2);
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
        if (h_fail) (void)hipHostFree(h_fail);
        if (h_miss) (void)hipHostFree(h_miss);
        if (h_stats) (void)hipHostFree(h_stats);
        if (h_begin) (void)hipHostFree(h_begin);
        if (h_progress) (void)hipHostFree(h_progress);
        if (ev_plan) (void)hipEventDestroy(ev_plan);
        if (h_cid) (void)hipHostFree(h_cid);
        if (h_sig) (void)hipHostFree(h_sig);
        if (xstream) (void)hipStreamDestroy(xstream);
        if (xstream2) (void)hipStreamDestroy(xstream2);
        for (int i = 0; i < 2; i++)
            if (ev_x[i]) (void)hipEventDestroy(ev_x[i]);
        for (int i = 0; i < 2; i++)
            if (ev_c[i]) (void)hipEventDestroy(ev_c[i]);
        if (ev_k) (void)hipEventDestroy(ev_k);
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

    // chunk c (values lb + 32 c ...) of variable v's initial domain [lb, ub]
    uint32_t init_chunk(int v, int c) const {
        const long long left = (long long)mgr.ub[v] - (long long)mgr.lb[v] + 1 - 32ll * c;
        return left >= 32 ? 0xffffffffu : (left > 0 ? ((1u << left) - 1u) : 0u);
    }
    int upload_program() {
        int rc = mgr.compile(prog);
        if (rc != STCSP_OK) return fail(rc, "%s", mgr.error.c_str());
        // node header word 2 = set ordinal (kSetBits bits) | dirty seed
        if (prog.sets.size() > (size_t)kSetMask) return fail(STCSP_E_UNSUPPORTED, "%zu constraint sets; node records address at most %u", prog.sets.size(), kSetMask);
        // LITE: every wavefront-revised constraint is a tuple bitmap with at most one violating tuple. Its
        // revision then either cannot prune (two or more open variables: the product of the others exceeds the
        // forbidden set) or is the one-open-variable look-up, so the general enumeration (tuple lanes, odometer,
        // bytecode interpreter, their LDS scratch and ~25 VGPRs) is compiled out: more resident wavefronts.
        lite = true;
        interpreted = false;
        wide_conditional = mgr.n_wide_conditional > 0;
        for (const SetDesc &sd : prog.sets)
            for (int i = sd.nsmall; i < sd.nitems; i++) {
                const ConDesc &cd = prog.cons[prog.items[sd.item_begin + i].con];
                lite = lite && cd.bitmap_off >= 0 && cd.n_forbidden >= 0 && cd.n_forbidden <= 1;
                interpreted = interpreted || cd.bitmap_off < 0;
            }
        if (const char *ev = getenv("STCSP_LITE")) lite = lite && atoi(ev) != 0;  // tuning switch
        if (mgr.W > 1) lite = false;  // (wide domains: every item is revised by dev_wide.hpp's bounds propagation)
        // one contiguous image; every section starts on a 16-byte boundary
        std::vector<uint32_t> img;
        ImgOff o{};
        int tables_end = 0;
        auto put = [&](const void *data, size_t bytes) {
            while (img.size() & 3) img.push_back(0u);
            int off = (int)img.size();
            size_t words = (bytes + 3) / 4;
            img.resize(img.size() + words, 0u);
            if (bytes) memcpy(img.data() + off, data, bytes);
            return off;
        };
        std::vector<uint32_t> init((size_t)ctx.N * mgr.W);  // chunk-major: init[c * N + v]
        for (int c = 0; c < mgr.W; c++)
            for (int v = 0; v < ctx.N; v++) init[(size_t)c * ctx.N + v] = init_chunk(v, c);
        // one row per constraint set when some set starts its fresh time points from its own propagated initial domains
        // (FlatProgram::set_fresh_init; the rows are filled in by fresh_init() once the program runs, known ones are kept)
        own_init_rows = false;
        for (uint8_t f : prog.set_fresh_init) own_init_rows = own_init_rows || f;
        if (own_init_rows) {
            const std::vector<uint32_t> plain = init;
            init.resize((size_t)ctx.N * prog.sets.size());
            for (size_t si = 0; si < prog.sets.size(); si++) {
                auto known = fresh_rows.find(mgr.sets[si]->tag);
                const std::vector<uint32_t> &row = (prog.set_fresh_init[si] && known != fresh_rows.end()) ? known->second : plain;
                std::copy(row.begin(), row.end(), init.begin() + si * ctx.N);
            }
        }
        o.init_stride = own_init_rows ? ctx.N : 0;
        // Sections in the order of how much a node needs them: what every node reads (the lane-per-item sweep
        // reads `sweep` and `itemrows` with per-lane addresses), then what wavefront revisions read before they can
        // start (item records, scopes, strides, the bytecode: small, and every read of them sits in a dependent
        // chain), the ConDesc array (host-side bookkeeping only), the (potentially big) tables last. When the whole
        // image does not fit the LDS budget, the longest affordable prefix of whole sections is staged instead
        // (hot_end[] = candidate cuts) and the rest is read from HBM/L2.
        std::vector<int> hot_end;
        auto cut = [&]() {
            while (img.size() & 3) img.push_back(0u);
            hot_end.push_back((int)img.size());
        };
        o.sets = put(prog.sets.data(), prog.sets.size() * sizeof(SetDesc));
        {
            std::vector<uint32_t> sweep(prog.items.size() * 4);
            for (size_t i = 0; i < prog.items.size(); i++) pack_sweep_item(prog.items[i], &sweep[i * 4]);
            o.sweep = put(sweep.data(), sweep.size() * 4);
        }
        o.nextpart = put(prog.nextpart.data(), prog.nextpart.size() * 4);
        o.var_lb = put(mgr.lb.data(), mgr.lb.size() * 4);
        o.var_init = put(init.data(), init.size() * 4);
        o.sig_vars = put(mgr.sig_vars.data(), mgr.sig_vars.size() * 4);
        o.until_y = put(mgr.until_y.data(), mgr.until_y.size() * 4);
        o.firstvars = put(prog.firstvars.data(), prog.firstvars.size() * 4);
        o.trans = put(prog.trans.data(), prog.trans.size() * sizeof(TransDesc));
        o.transvals = put(prog.transvals.data(), prog.transvals.size() * 4);
        o.fstrides = put(prog.fstrides.data(), prog.fstrides.size() * 4);
        o.arr_off = put(mgr.array_off.data(), mgr.array_off.size() * 4);
        cut();
        o.itemrows = put(prog.itemrows.data(), prog.itemrows.size() * 4);
        cut();
        // the row tables of the lane-revised items: next in line when they are small (every sweep reads them with per-lane
        // addresses: out of LDS that is one ds_read_b128, out of global memory a 64-way divergent vector load per trip of the row
        // loop -- digitinvader9's 268 words used to sit behind 4.9 MB of bitmaps in `tables`); big ones (the synthetic family:
        // 122 KB) go behind the small sections that wavefront revisions need, so that they do not push those out of the prefix
        const bool stables_early = prog.stables.size() <= 2048;
        if (stables_early) {
            o.stables = put(prog.stables.data(), prog.stables.size() * 4);
            cut();
        }
        o.hot_words = (int)img.size();
        {
            // ItemDesc records of the wavefront-revised items only (the lane-revised ones are read through their
            // packed sweep records), set after set: SetDesc::witem_begin
            std::vector<ItemDesc> witems;
            for (const SetDesc &sd : prog.sets)
                witems.insert(witems.end(), prog.items.begin() + sd.item_begin + sd.nsmall, prog.items.begin() + sd.item_begin + sd.nitems);
            o.items = put(witems.data(), witems.size() * sizeof(ItemDesc));
        }
        o.scope = put(prog.scope.data(), prog.scope.size() * 4);
        o.strides = put(prog.strides.data(), prog.strides.size() * 4);
        cut();
        if (lite) {
            // a LITE kernel never reads the bytecode: the tables come first, so that "everything the kernel reads" is a
            // prefix of the image (tables_end) -- what the big-workgroup variant stages
            if (!stables_early) o.stables = put(prog.stables.data(), prog.stables.size() * 4);
            o.tables = put(prog.tables.data(), prog.tables.size() * 4);
            while (img.size() & 3) img.push_back(0u);
            tables_end = (int)img.size();
            o.code = put(prog.code.data(), prog.code.size() * 4);
            o.cons = put(prog.cons.data(), prog.cons.size() * sizeof(ConDesc));
        } else {
            o.code = put(prog.code.data(), prog.code.size() * 4);
            cut();
            if (!stables_early) {
                o.stables = put(prog.stables.data(), prog.stables.size() * 4);
                cut();
            }
            o.cons = put(prog.cons.data(), prog.cons.size() * sizeof(ConDesc));
            o.tables = put(prog.tables.data(), prog.tables.size() * 4);  // last: the part that may be big
            tables_end = 0;
        }
        while (img.size() & 3) img.push_back(0u);
        o.words = (int)img.size();
        // everything the general kernels read through v() / u() ends where the ConDesc array (host bookkeeping) and the tables
        // begin: a staged prefix that reaches this far lets them run the L = 2 kernels (plain LDS reads, dev_propagate.hpp Img)
        prefix_need = lite ? 0 : o.cons;
        HIPCHK(d_img.upload(img));
        HIPCHK(d_code.upload(prog.code));
        // direct transition tables: one look-up per leaf, may be MBs (up to 256 MB): not part of the image, and not uploaded as
        // a whole at every translation stop -- a recompile appends tables and patches a few entries, so only the runs that
        // differ from the device's copy travel
        {
            const std::vector<int32_t> &nt = prog.tdirect;
            if (nt.size() > d_tdirect.n || !d_tdirect.p) {
                HIPCHK(hipStreamSynchronize(stream));
                HIPCHK(d_tdirect.alloc(std::max<size_t>(nt.size() + nt.size() / 2, 1024)));
                tdirect_dev.clear();
            }
            size_t i = 0;
            while (i < nt.size()) {
                if (i < tdirect_dev.size() && tdirect_dev[i] == nt[i]) {
                    i++;
                    continue;
                }
                size_t j = i + 1, same = 0;  // run of differing entries (gaps of < 64 equal ones are not worth a second copy)
                while (j < nt.size() && same < 64) {
                    same = (j < tdirect_dev.size() && tdirect_dev[j] == nt[j]) ? same + 1 : 0;
                    j++;
                }
                j -= same;
                HIPCHK(hipMemcpyAsync(d_tdirect.p + i, nt.data() + i, (j - i) * sizeof(int32_t), hipMemcpyHostToDevice, stream));
                i = j;
            }
            HIPCHK(hipStreamSynchronize(stream));  // (pageable source)
            tdirect_dev = nt;
        }
        ctx.tdirect = d_tdirect.p;
        // bitmaps too big for the host to tabulate: the device fills them in (k_tabulate), once -- the result goes
        // back into the SetManager's table cache, so later compiles place the finished words
        for (const TabulateTodo &td : prog.todo) {
            if (td.scope.size() > 16) return fail(STCSP_E_INTERNAL, "device tabulation of a %zu-ary constraint", td.scope.size());
            TabArgs ta{};
            ta.code = d_code.p + td.code_off;
            ta.product = td.product;
            ta.scope_len = (int)td.scope.size();
            ta.uses_valid = td.uses_valid;
            for (size_t j = 0; j < td.scope.size(); j++) {
                ta.size[j] = mgr.ub[td.scope[j]] - mgr.lb[td.scope[j]] + 1;
                ta.lb[j] = mgr.lb[td.scope[j]];
            }
            if (!d_arr_off.p) HIPCHK(d_arr_off.upload(mgr.array_off));
            ta.arr_off = d_arr_off.p;
            ta.arr_data = d_arr_data.p;
            uint32_t *dst = d_img.p + o.tables + td.tables_off;
            const size_t nwords = (size_t)((td.product + 31) / 32);
            hipLaunchKernelGGL(k_tabulate, dim3((unsigned)((td.product + 255) / 256)), dim3(256), 0, stream, ta, dst);
            HIPCHK(hipGetLastError());
            std::vector<uint32_t> back(nwords);
            HIPCHK(hipMemcpyAsync(back.data(), dst, nwords * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            mgr.store_tabulated(td.key, back.data(), nwords, td.product);
        }
        ctx.img = d_img.p;
        ctx.o = o;
        ctx.code = d_code.p;
        ctx.nsets = (int)prog.sets.size();
        compact_sweeps = false;
        for (const SetDesc &sd : prog.sets) compact_sweeps = compact_sweeps || sd.nsmall > kCompactSweepItems;
        ctx.stack_slots = prog.max_stack + 2;
        ctx.max_iw = 1;
        int max_nfirst = 0;
        for (const SetDesc &sd : prog.sets) {
            ctx.max_iw = std::max(ctx.max_iw, sd.iw);
            max_nfirst = std::max(max_nfirst, sd.nfirst);
        }
        ctx.sib_depth = kSibDepth;
        const size_t scratch = (size_t)4 * wave_scratch_words(ctx.NK, ctx.stack_slots, lite, ctx.sib_depth) * sizeof(int);
        if (scratch > 160 * 1024) return fail(STCSP_E_UNSUPPORTED, "expression stack too deep for LDS");
        // stage the image in LDS when image + scratch leave room for >= 2 workgroups per CU
        img_in_lds = (size_t)o.words * 4 + scratch <= 64 * 1024 && mgr.W == 1;  // (the wide kernels exist in the partly-staged form only)
        if (const char *ev = getenv("STCSP_IMG_LDS")) img_in_lds = img_in_lds && atoi(ev) != 0;  // tuning switch
        ctx.stage_words = img_in_lds ? o.words : 0;
        lds_bytes = scratch + (size_t)ctx.stage_words * 4;
        // Big workgroups: a LITE program too big for four copies per CU but not for one (<= 160 KB with the scratch of 16
        // wavefronts at sibling depth 2) runs as ONE 1024-thread workgroup per CU around one staged copy -- the same 4
        // wavefronts per SIMD as four 256-thread workgroups, every program read out of LDS instead of L2.
        big = false;
        const char *big_env = getenv("STCSP_BIG");  // 0: never; 2: whenever it fits (tests: small programs through the big-workgroup kernel)
        if (lite && tables_end > 0 && (!img_in_lds || (big_env && atoi(big_env) == 2))) {
            int depth = kSibDepth;
            if (const char *ev = getenv("STCSP_BIG_DEPTH")) depth = std::max(1, std::min(kSibDepth, atoi(ev)));
            size_t scratch_big = 0;
            for (;; depth--) {  // the deepest sibling stack that still fits (never below 1: a depth of 0 would overlap the wavefronts' regions)
                scratch_big = (size_t)STCSP_BIG_WAVES * wave_scratch_words(ctx.NK, ctx.stack_slots, true, depth) * sizeof(int);
                if ((size_t)tables_end * 4 + scratch_big <= (size_t)160 * 1024 || depth <= std::min(2, kSibDepth) || depth <= 1) break;
            }
            big = depth >= 1 && (size_t)tables_end * 4 + scratch_big <= (size_t)160 * 1024 &&
                  scratch_big >= (size_t)STCSP_BIG_WAVES * wave_sib_offset(ctx.NK, ctx.stack_slots, true) * sizeof(int);
            if (big_env) big = big && atoi(big_env) != 0;
            if (big) {
                img_in_lds = false;
                ctx.sib_depth = depth;
                ctx.stage_words = tables_end;
                lds_bytes = (size_t)tables_end * 4 + scratch_big;
            }
        }
        const bool try_prefix = !big && !img_in_lds && !(getenv("STCSP_IMG_LDS") && atoi(getenv("STCSP_IMG_LDS")) == 0);
        // Grid = exactly the workgroups that are resident at once: wavefronts take node slots with
        // a static grid stride, so a workgroup that has to wait for a free CU slot would start its
        // share only when another one has finished all of its own (a 2x tail).
        {
            int per_cu = 0;
            hipError_t e;
            const void *fn;
            prefix_complete = false;  // (decided below, from what gets staged)
            switch (DR) {
                case 1: fn = expand_fn<1>(); break;
                case 2: fn = expand_fn<2>(); break;
                default: fn = expand_fn<4>(); break;
            }
            if (big) {
                // more than 64 KB of dynamic LDS has to be asked for, per kernel (the probe kernel stages the same image)
                const void *pf;
                switch (DR) {
                    case 1: pf = probe_fn<1>(); break;
                    case 2: pf = probe_fn<2>(); break;
                    default: pf = probe_fn<4>(); break;
                }
                HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
                HIPCHK(hipFuncSetAttribute(pf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            }
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, big ? STCSP_BIG_WAVES * 64 : 256, lds_bytes);
            if (big && (e != hipSuccess || per_cu < 1)) return fail(STCSP_E_INTERNAL, "big-workgroup kernel does not fit a CU (%zu B of LDS)", lds_bytes);
            if (try_prefix && e == hipSuccess && per_cu > 0) {
                // the longest prefix of whole hot sections that costs no resident workgroup: the kernel's
                // registers allow per_cu workgroups, each may use 160 KB / per_cu of LDS (<= 64 KB)
                const size_t budget = std::min<size_t>((size_t)160 * 1024 / (size_t)per_cu, (size_t)64 * 1024);
                for (int cut : hot_end)
                    if (scratch + (size_t)cut * 4 <= budget) ctx.stage_words = cut;
                if (ctx.stage_words) {
                    int with_prefix = 0;
                    const size_t bytes = scratch + (size_t)ctx.stage_words * 4;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&with_prefix, fn, 256, bytes) == hipSuccess && with_prefix >= per_cu)
                        lds_bytes = bytes;
                    else
                        ctx.stage_words = 0;
                }
            }
            prefix_complete = !lite && !img_in_lds && !big && !compact_sweeps && mgr.W == 1 && prefix_need > 0 && ctx.stage_words >= prefix_need &&
                              !(getenv("STCSP_PREFIX_KERNEL") && atoi(getenv("STCSP_PREFIX_KERNEL")) == 0);
#ifdef STCSP_PHASES
            if (DR == 4) prefix_complete = false;
#endif
            if (prefix_complete) {  // the kernel that will run: its own occupancy
                const void *f2;
                switch (DR) {
                    case 1: f2 = expand_fn<1>(); break;
                    case 2: f2 = expand_fn<2>(); break;
                    default: f2 = expand_fn<4>(); break;
                }
                int pc2 = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc2, f2, 256, lds_bytes) == hipSuccess && pc2 > 0) per_cu = pc2;
                else prefix_complete = false;
            }
            hipDeviceProp_t prop;
            if (e == hipSuccess && per_cu > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess)
                max_blocks = per_cu * prop.multiProcessorCount;
            if (const char *ev = getenv("STCSP_BLOCKS")) if (atoi(ev) > 0) max_blocks = atoi(ev);
            if (getenv("STCSP_DEBUG")) {
                fprintf(stderr, "[engine] staging cuts (words):");
                for (int cut : hot_end) fprintf(stderr, " %d", cut);
                fprintf(stderr, "\n");
            }
            if (getenv("STCSP_DEBUG"))
                fprintf(stderr, "[engine] %s kernel, image %d words (%s: %d words staged), per-wavefront LDS scratch %zu B (stack slots %d), LDS/workgroup %zu B, %d workgroups/CU -> grid %d\n",
                        big ? "LITE big-workgroup" : lite ? "LITE" : prefix_complete ? "general (descriptors in LDS, tables global)" : "general", o.words, img_in_lds || big ? "in LDS" : "global", ctx.stage_words, scratch / 4, ctx.stack_slots, lds_bytes, per_cu, max_blocks);
        }
        return STCSP_OK;
    }

    int create(const stcsp_problem *p, const stcsp_options *o) {
        if (o) opt = *o;
        if (opt.world <= 0) opt.world = 1;
        if (opt.rank < 0 || opt.rank >= opt.world) return fail(STCSP_E_INVALID, "rank %d outside world %d", opt.rank, opt.world);
        sharded = opt.world > 1 || (opt.flags & STCSP_F_STEPPED) != 0;
        if (const char *ev = getenv("STCSP_FAULT")) {
            std::string f(ev);
            const size_t c1 = f.find(','), c2 = c1 == std::string::npos ? c1 : f.find(',', c1 + 1);
            if (c1 != std::string::npos && atoi(f.c_str() + c1 + 1) == opt.rank) {
                fault_call = f.substr(0, c1);
                fault_left = c2 == std::string::npos ? 1 : std::max(1, atoi(f.c_str() + c2 + 1));
            }
        }
        int rc = mgr.init(p, sharded);
        if (rc != STCSP_OK) return fail(rc, "%s", mgr.error.c_str());
        mgr.device_tabulation = !(getenv("STCSP_DEVICE_TABULATE") && atoi(getenv("STCSP_DEVICE_TABULATE")) == 0);
        const int N = mgr.N, K = mgr.K;
        // domains of up to 32 values take one bitset word per (variable, time point), up to 64 two, up to 128 four (W; one W
        // for the whole block: dev_wide.hpp); the block of N*K*W words lives in at most kMaxDomRegs registers per lane
        for (int v = 0; v < N; v++) {
            long long width = (long long)mgr.ub[v] - (long long)mgr.lb[v] + 1;
            if (width > 128)
                return fail(STCSP_E_UNSUPPORTED, "variable %d has %lld values; this engine packs at most 128 (four bitset words) per variable and time point", v, width);
        }
        const int W = mgr.W;
        if ((long long)N * K * W > 64 * kMaxDomRegs)
            return fail(STCSP_E_UNSUPPORTED, "N*K*W = %d*%d*%d exceeds the %d-word register-resident block", N, K, W, 64 * kMaxDomRegs);
        if (mgr.n_until_cons > 32) return fail(STCSP_E_UNSUPPORTED, "more than 32 until constraints");
        if (1 + mgr.n_sig + mgr.n_until_cons > 64) return fail(STCSP_E_UNSUPPORTED, "signature longer than 63 words");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(STCSP_E_DEVICE, "no HIP device available");
        device = opt.device;
        HIPCHK(hipSetDevice(device));
        HIPCHK(hipStreamCreate(&stream));
        {
            // lowest priority: the export must never delay the search, and streams of different priorities get
            // different hardware queues -- with the default 4 queues per process, streams of equal priority share them
            // round-robin with everything else the process created (torch, RCCL), and an export stream that lands in
            // the search stream's queue serialises with the bursts of k_expand (measured: +1.5 ms per solve)
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&xstream, hipStreamNonBlocking, lo));
            HIPCHK(hipStreamCreateWithPriority(&xstream2, hipStreamNonBlocking, lo));
            HIPCHK(hipEventCreateWithFlags(&ev_x[0], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_x[1], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_c[0], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_c[1], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ev_k, hipEventDisableTiming));
        }
        HIPCHK(hipHostMalloc((void **)&h_progress, sizeof(Progress)));
        memset(h_progress, 0, sizeof(Progress));
        HIPCHK(hipEventCreateWithFlags(&ev_plan, hipEventDisableTiming));
        L = CtlLayout(opt.world);
        ctx.N = N;
        ctx.K = K;
        ctx.NK = N * K * W;  // block words (chunk-major for W > 1: word(c, p, v) = c*N*K + p*N + v)
        ctx.W = W;
        ctx.NS = node_stride(N, K * W);
        ctx.sig_len = mgr.n_sig + mgr.n_until_cons;
        ctx.n_sig = mgr.n_sig;
        ctx.n_until_cons = mgr.n_until_cons;
        ctx.KL = 1 + ctx.sig_len;
        ctx.CS = cand_stride(N, K * W, ctx.sig_len);
        ctx.ES = edge_stride(N);
        ctx.world = opt.world;
        ctx.sharded = sharded ? 1 : 0;
        // tests: every leaf becomes a candidate record, also those this shard owns -- so that a group of ONE rank still moves
        // records through its transport (RCCL send / recv to itself) and through k_commit
        if (sharded && getenv("STCSP_FORCE_CANDIDATES") && atoi(getenv("STCSP_FORCE_CANDIDATES")) != 0) ctx.sharded = 2;
        ctx.budget_bitmap = (int)kBudgetBitmapIters;
        ctx.budget_code = (int)kBudgetCodeIters;
        if (const char *ev = getenv("STCSP_BUDGET_BITMAP")) ctx.budget_bitmap = std::max(1, atoi(ev));
        if (const char *ev = getenv("STCSP_BUDGET_CODE")) ctx.budget_code = std::max(1, atoi(ev));
        ctx.rank = opt.rank;
        // constraint sets whose captured `first` variables span few value tuples are translated ahead of need
        // (SURVEY 8(f) row 1): the device then never stops for them. STCSP_PRETRANSLATE=<tuples> (0 = off).
        {
            long long tuples = 65536;  // (~20 us of host time per tuple; the table costs 4 bytes per tuple in HBM)
            int max_sets = 16384;      // (node records address at most 65535 sets)
            if (const char *ev = getenv("STCSP_PRETRANSLATE")) tuples = atoll(ev);
            if (const char *ev = getenv("STCSP_PRETRANSLATE_SETS")) max_sets = std::max(1, atoi(ev));
            // the whole call is bounded as well: 2^20 tuples over all sets, 20 s of host time -- what is left over
            // is translated on demand. STCSP_PRETRANSLATE_TOTAL / _SECONDS override.
            long long total = 1ll << 20;
            double secs = 20.0;
            if (const char *ev = getenv("STCSP_PRETRANSLATE_TOTAL")) total = atoll(ev);
            if (const char *ev = getenv("STCSP_PRETRANSLATE_SECONDS")) secs = atof(ev);
            if (tuples > 0) {
                const int pre = mgr.pretranslate(tuples, max_sets, total, secs);
                if (pre < 0) return fail(pre, "%s", mgr.error.c_str());
            }
        }
        DR = (N * K * W + 63) / 64;
        if (DR == 3) DR = 4;
        HIPCHK(d_arr_data.upload(mgr.array_data));
        ctx.arr_data = d_arr_data.p;
        rc = upload_program();
        if (rc != STCSP_OK) return rc;
        // pools
        // Nodes taken per launch. The default is large -- a round costs ~20 us whatever its size and
        // partialorder_18 runs 46 % faster with 1 M than with 64 k -- and adapts downwards when the
        // frontier arena passes its soft limit (explosive searches: memory ~ depth x batch). Sharded
        // engines keep 64 k (their outbox is sized by the batch for every peer).
        auto_batch = opt.batch_nodes <= 0;
        // (round 3, slots dealt by ticket: 1 M again for blocks of at most two registers per lane -- partialorder_18 56 -> 50.5 ms;
        // the wide blocks of the synthetic family keep 256 k: their frontier is depth x batch x 600 B and 1 M costs them 8 %)
        // Sharded engines: 256 k too -- their outboxes no longer grow with the batch (the planner bounds a round by the room
        // the outboxes have left: dev_kernels.hpp plan_next), so the time-boxed synthetic runs as fast sharded as unsharded.
        int batch = opt.batch_nodes > 0 ? opt.batch_nodes : (sharded ? 262144 : (ctx.NK <= 128 ? 1048576 : 262144));
        if (const char *ev = getenv("STCSP_BATCH")) if (atoi(ev) > 0 && auto_batch) batch = atoi(ev);
        {
            size_t free_b = 0, total_b = 0;
            arena_soft_words = ((size_t)8 << 30) / 4;  // 8 GiB
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b / 16 > ((size_t)1 << 30)) arena_soft_words = std::min(arena_soft_words * 4, free_b / 16 / 4 * 4);
            if (const char *ev = getenv("STCSP_ARENA_SOFT_MB")) if (atoi(ev) > 0) arena_soft_words = (size_t)atoi(ev) * ((1u << 20) / 4);  // tests: force the batch to shrink
        }
        chunk_r = chunk_r0 = std::max(1, (batch + R - 1) / R);
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
        // arena of node segments (grown on demand)
        HIPCHK(d_arena.alloc(small_pools ? (size_t)4 * R * ctx.NS : std::min((size_t)8 * R * chunk_r * ctx.NS, std::max(arena_soft_words, (size_t)64 * R * ctx.NS))));
        HIPCHK(d_plan.alloc(1));
        HIPCHK(d_ctx.alloc(1));
        HIPCHK(hipHostMalloc((void **)&h_ctx, sizeof(Ctx)));
        HIPCHK(hipHostMalloc((void **)&h_plan, sizeof(Plan)));
        if (!sharded && !(opt.time_limit_s > 0) && !(opt.max_search_nodes > 0)) burst = 32;  // (budgets are checked between bursts)
        // (Round 3 enqueued bursts of 4 under HIP runtimes from 7.2 on: there the export streams' device-to-host copies only started
        // when the search stream had drained. What held them back were read_plan's own three small copies, queued on the search
        // stream behind the burst: the copy engine serves its requests in order, whatever stream they come from. With the planner's
        // verdict in the pinned mirror (read_plan_fast) no copy waits behind a burst any more and both runtimes run bursts of 32:
        // partialorder_14 under 7.2, solve + export: 3.99 -> 3.3 ms.)
        if (const char *ev = getenv("STCSP_BURST")) burst = std::max(1, atoi(ev));
        if (const char *ev = getenv("STCSP_CHAIN_SMALL")) {
            chain_small = std::max(1, atoi(ev));
            chain_small_auto = false;
        }
        if (const char *ev = getenv("STCSP_CHAIN_BIG")) chain_big = std::max(1, atoi(ev));
        if (const char *ev = getenv("STCSP_CHAIN_THRESH")) chain_thresh = std::max(0, atoi(ev));
        if (const char *ev = getenv("STCSP_CHAIN_HEAVY")) chain_heavy = std::max(1, atoi(ev));
        if (const char *ev = getenv("STCSP_DEBUG")) dbg_rounds = atoi(ev) >= 2;
        // outbox: [owner][region] x cand_cap records. Unsharded: unused. A slot may meet a leaf in every expansion of its
        // chain, so a region of an owner's outbox receives up to chain x max-take candidates per launch; the planner takes no
        // more nodes per region than the outboxes have room for (plan_next), so their size is independent of the batch: room
        // for 4,096 nodes per region and launch (world 8: 3.4 GB at 100-word records).
        if (chain_small_auto) chain_small = lite ? 4 : (wide_conditional ? 1 : (interpreted ? 2 : 16));  // (what begin() will plan with: the outboxes are sized for it)
        cand_cap = (uint32_t)(sharded ? std::max(std::max(8, std::max(chain_small, chain_big)) * std::min(chunk_r, 4096), 4096) : 64);
        if (const char *ev = getenv("STCSP_CAND_CAP")) if (sharded && atoi(ev) > 0) cand_cap = (uint32_t)std::max(atoi(ev), 2 * std::max(chain_small, chain_big));  // tests: outboxes that fill up
        HIPCHK(d_cand.alloc((size_t)opt.world * R * cand_cap * ctx.CS));
        if (sharded) HIPCHK(d_pack.alloc((size_t)R * cand_cap * ctx.CS));
        sync_ctx();
        return fresh_init();
    }

    int alloc_table(uint32_t slots) {
        // entries of 2^tab_shift words (key + slot word in one or more whole 128-byte lines: dev_layout.hpp Ctx::tab_shift);
        // zeroed here, never again: an entry belongs to the solve whose generation it carries
        ctx.tab_shift = table_entry_shift(ctx.KL);
        const size_t u64s = ((size_t)slots << ctx.tab_shift) / 2;
        HIPCHK(d_slots.alloc(u64s));
        HIPCHK(hipMemsetAsync(d_slots.p, 0, u64s * sizeof(unsigned long long), stream));
        ctx.slots = d_slots.p;
        ctx.slot_mask = slots - 1;
        return STCSP_OK;
    }
    int alloc_states(uint32_t cap) {
        DevBuf<uint32_t> nb;
        HIPCHK(nb.alloc((size_t)cap * ctx.KL));
        { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }  // (k_stream_keys may still read the old pool)
        if (d_state_keys.p && n_states)
            HIPCHK(hipMemcpyAsync(nb.p, d_state_keys.p, (size_t)n_states * ctx.KL * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));
        std::swap(d_state_keys.p, nb.p);
        std::swap(d_state_keys.n, nb.n);
        ctx.state_keys = d_state_keys.p;
        if (!sharded && d_sdeg.n < cap) {  // (the export streams were synchronised above)
            DevBuf<uint32_t> nd;
            HIPCHK(nd.alloc(cap));
            HIPCHK(hipMemsetAsync(nd.p, 0, (size_t)cap * sizeof(uint32_t), stream));
            if (d_sdeg.p && n_states) HIPCHK(hipMemcpyAsync(nd.p, d_sdeg.p, (size_t)std::min<size_t>(d_sdeg.n, n_states) * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            std::swap(d_sdeg.p, nd.p);
            std::swap(d_sdeg.n, nd.n);
        }
        ctx.state_cap = cap;
        return STCSP_OK;
    }
    int alloc_edges(uint32_t cap) {
        DevBuf<uint32_t> nb;
        HIPCHK(nb.alloc((size_t)R * cap * ctx.ES));
        { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }  // (a streaming chunk may still read the old log)
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
    }
    int flush_ctx() {
        sync_ctx();
        h_ctx->tab_gen = ctx.tab_gen;  // (travels with every launch: no reason to refresh the device copy)
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
    bool tail_fresh = false;  // h_ctl / h_stats hold the state after the last device work of the search (read_plan copied them)
    bool ctl_fresh = false;   // h_ctl is what the copying read_plan brought and nothing has been enqueued since (sharded stepping: expand_local
                              // needs the outbox cursors; a second copy + synchronisation of the control block costs ~25 us per superstep)
    // Waiting for a burst on the host: a short pause between two looks at the event (a few hundred nanoseconds), and the core is only
    // given up once the wait has lasted a few milliseconds. (Round 3 yielded after every look: on a box whose cores are shared
    // with other tenants sched_yield() hands the core to whoever is runnable, and the end of a 3-ms burst was then noticed 0.3 ms late
    // -- one bench run in five showed search 3.26 ms around kernels that took 2.90.)
    struct PollWait {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        unsigned looks = 0;
        void operator()() {
#if defined(__x86_64__) || defined(__i386__)
            for (int i = 0; i < 32; i++) __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if ((++looks & 63u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) std::this_thread::yield();
        }
    };
    // End of a burst. Fast path (the common case: the planner says "go on"): wait for the burst's end -- shipping finished parts
    // of the edge log meanwhile -- and take the planner's verdict from the pinned mirror the device wrote it to; no copies.
    // Anything else (done, a pool to grow, a translation, an error, a budget) takes the full read below.
    bool mirror_ok = true;  // STCSP_PLAN_MIRROR=0: always copy (A/B)
    int read_plan_fast(bool &handled) {
        handled = false;
        if (!ctx.progress || !mirror_ok) return STCSP_OK;
        HIPCHK(hipEventRecord(ev_plan, stream));
        PollWait wait;
        for (;;) {
            const hipError_t q = hipEventQuery(ev_plan);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) HIPCHK(q);
            if (streaming) {
                int rc = ship_progress();
                if (rc != STCSP_OK) {
                    (void)hipStreamSynchronize(stream);
                    return rc;
                }
            }
            wait();
        }
        const volatile Progress *pr = h_progress;
#ifndef STCSP_PHASES
        if (pr->status == PS_DONE && !sharded && !dbg_rounds) {
            // the end of an unsharded search: cursors, state count and the summed work counters are in the mirror too
            if (!h_stats) HIPCHK(hipHostMalloc((void **)&h_stats, kStatSlots * kStatWords * sizeof(unsigned long long)));
            memset(h_stats, 0, kStatSlots * kStatWords * sizeof(unsigned long long));
            for (int k = 0; k < kMirrorCounters; k++) h_stats[k] = pr->counters[k];
            for (int i = 0; i < L.words; i++) h_ctl[i] = 0;
            for (int r = 0; r < R; r++) h_ctl[L.edge0 + r * CST] = (uint32_t)pr->edge_seen[r];
            h_ctl[L.misc0 + MISC_NSTATES * CST] = (uint32_t)pr->states_seen;
            h_plan->status = PS_DONE;
            h_plan->rounds = pr->rounds;
            h_plan->open_total = pr->open_total;
            for (int r = 0; r < R; r++) h_plan->edge_seen[r] = (unsigned)pr->edge_seen[r];
            h_plan->states_seen = (unsigned)pr->states_seen;
            levels = h_plan->rounds;
            prog_have = false;
            tail_fresh = true;
            handled = true;
            return STCSP_OK;
        }
#endif
        if (pr->status != PS_RUN) return STCSP_OK;  // (the caller reads everything)
        h_plan->status = PS_RUN;
        h_plan->rounds = pr->rounds;
        h_plan->open_total = pr->open_total;
        for (int r = 0; r < R; r++) h_plan->edge_seen[r] = (unsigned)pr->edge_seen[r];
        h_plan->states_seen = (unsigned)pr->states_seen;
        levels = h_plan->rounds;
        prog_have = false;
        tail_fresh = false;
        handled = true;
        return STCSP_OK;
    }
    bool last_plan_fast = false;  // h_plan holds only what read_plan_fast fills in (status, rounds, open nodes, export cursors)
    int read_plan(bool allow_fast = false) {
        last_plan_fast = false;
        bool waited = false;  // the fast path has seen the burst's end already: what follows are the copies only
        if (allow_fast) {
            bool handled = false;
            int rc = read_plan_fast(handled);
            if (rc != STCSP_OK) return rc;
            if (handled) {
                last_plan_fast = h_plan->status == PS_RUN;
                return STCSP_OK;
            }
            waited = ctx.progress && mirror_ok;
        }
        HIPCHK(hipMemcpyAsync(h_plan, d_plan.p, kPlanHeader, hipMemcpyDeviceToHost, stream));
        // ... and, behind it, what finish() reads when this burst turns out to be the last one: the control block and the
        // statistics (a few KB; a separate copy + synchronisation at the end costs ~40 us of every solve)
        if (!h_stats) HIPCHK(hipHostMalloc((void **)&h_stats, kStatSlots * kStatWords * sizeof(unsigned long long)));
        HIPCHK(hipMemcpyAsync(h_ctl, d_ctl.p, L.words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(h_stats, d_stats.p, kStatSlots * kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        if (streaming && ctx.progress && !waited) {
            // instead of sleeping in the synchronisation: watch the progress mirror and ship what the launches
            // of the running burst have finished
            HIPCHK(hipEventRecord(ev_plan, stream));
            PollWait wait;
            for (;;) {
                const hipError_t q = hipEventQuery(ev_plan);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) HIPCHK(q);
                int rc = ship_progress();
                if (rc != STCSP_OK) {
                    (void)hipStreamSynchronize(stream);  // the burst still writes the progress mirror
                    return rc;
                }
                wait();
            }
            prog_have = false;  // (the caller ships everything up to the end of the burst)
        }
        HIPCHK(hipStreamSynchronize(stream));
        levels = h_plan->rounds;
        ctl_fresh = true;
        tail_fresh = !sharded;  // (the stepping interface enqueues commit / adopt / donate work between its reads)
        return STCSP_OK;
    }
    Progress *h_progress = nullptr;
    hipEvent_t ev_plan = nullptr;
    struct ProgressSnap {
        uint32_t edge_seen[R];
        uint32_t states_seen;
    } prog_snap{};
    bool prog_have = false, prog_sent = false;
    unsigned long long prog_gen = 0;
    int ship_progress() {
        // a snapshot = all words carrying the same tag (= number of the launch that wrote them)
        const volatile unsigned long long *src = (const volatile unsigned long long *)h_progress;
        const unsigned long long s0 = src[R];
        const unsigned long long g = s0 >> 32;
        if (g == prog_gen) {
            // nothing new finalized -- but the NEXT round's launch may have started meanwhile: the round of the snapshot in hand
            // has ended then (same stream, in order), and the snapshot can go a whole round earlier
            if (prog_have && !prog_sent && src[R + 1] > prog_gen) {
                int rc = stream_edges(prog_snap.edge_seen, false);
                if (rc == STCSP_OK) rc = stream_states(prog_snap.states_seen);
                if (rc != STCSP_OK) return rc;
                prog_sent = true;
            }
            return STCSP_OK;
        }
        ProgressSnap s;
        s.states_seen = (uint32_t)s0;
        bool whole = true;
        for (int r = 0; r < R; r++) {
            const unsigned long long e = src[r];
            whole = whole && (e >> 32) == g;
            s.edge_seen[r] = (uint32_t)e;
        }
        if (!whole) return STCSP_OK;  // launch g is still writing (or g + 1 already is): look again
        // launch g has finalized, so the launch of the snapshot in hand has ended: its records are in memory
        if (prog_have && !prog_sent) {
            int rc = stream_edges(prog_snap.edge_seen, false);
            if (rc == STCSP_OK) rc = stream_states(prog_snap.states_seen);
            if (rc != STCSP_OK) return rc;
        }
        prog_snap = s;
        prog_have = true;
        prog_sent = false;
        prog_gen = g;
        return STCSP_OK;
    }
    unsigned launch_seq = 0;  // id of the next k_expand launch (Plan::gate)
    int replan() {
        tail_fresh = false;
        ctl_fresh = false;
        hipLaunchKernelGGL(k_replan, dim3(1), dim3(64), 0, stream, ctx, launch_seq);
        HIPCHK(hipGetLastError());
        return STCSP_OK;
    }

    int begin() {
        HIPCHK(hipSetDevice(device));
        tail_fresh = false;
        ctl_fresh = false;
        std::fill(edge_count.begin(), edge_count.end(), 0u);
        n_states = 0;
        truncated = false;
        levels = 0;
        { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }
        streamed = 0;
        streamed_states = 0;
        for (int r = 0; r < R; r++) streamed_r[r] = 0;
        streaming = !(opt.flags & (STCSP_F_NO_EXPORT | STCSP_F_KEEP_RAW_EDGES)) && !getenv("STCSP_HOST_EXPORT") &&
                    !(getenv("STCSP_STREAM_EXPORT") && atoi(getenv("STCSP_STREAM_EXPORT")) == 0);
        if (const char *sc = getenv("STCSP_STREAM_CHUNK")) stream_chunk_min = (size_t)std::max(1, atoi(sc));
        if (const char *sc = getenv("STCSP_STREAM_CHUNK_IDLE")) stream_chunk_idle = (size_t)std::max(1, atoi(sc));
        ev_x_used[0] = ev_x_used[1] = false;
        ev_k_used = false;
        memset(h_progress, 0, sizeof(Progress));
        prog_have = false;
        prog_gen = 0;
        // the progress mirror: edge-log cursors for the streaming export AND the planner's verdict for the host (read_plan_fast)
        ctx.progress = !(getenv("STCSP_STREAM_POLL") && atoi(getenv("STCSP_STREAM_POLL")) == 0) ? h_progress : nullptr;
        mirror_ok = !(getenv("STCSP_PLAN_MIRROR") && atoi(getenv("STCSP_PLAN_MIRROR")) == 0);
        translation_stops = 0;
        finished = false;
        exp_on_device = false;
        ev_used = 0;
        seconds_expand_kernel = 0;
        expand_launches = 0;
        for (int i = 0; i < L.words; i++) h_ctl[i] = 0;
        if (++ctx.tab_gen == 0u) {  // (2^32 solves on one engine: start over with a clean table)
            HIPCHK(hipMemsetAsync(d_slots.p, 0, d_slots.n * sizeof(unsigned long long), stream));
            ctx.tab_gen = 1u;
        }
        memset(h_plan, 0, sizeof(Plan));
        chunk_r = chunk_r0;
        h_plan->chunk_r = chunk_r;
        // (general kernels: 16 since round 4 -- the time cap chain_heavy is what ends a slot there, the count only has to stay out of
        // its way: digitinvader9 18.8 -> 17.9 ms, digitinvader7 10.1 -> 9.6 with 16 instead of 8; 24 the same; tools/env_sweep.py)
        if (chain_small_auto) chain_small = lite ? 4 : (wide_conditional ? 1 : (interpreted ? 2 : 16));
        h_plan->chain_small = chain_small;
        h_plan->chain_big = chain_big;
        h_plan->chain_thresh = chain_thresh;
        h_plan->chain_heavy = chain_heavy;
        h_plan->world = opt.world;
        // ONE launch (k_begin) zeroes the control block, the statistics and the out-degree mirror and brings the plan header, the
        // root's table entry, key and search node over from the pinned staging buffer [plan words | entry | node]
        const size_t esz = (size_t)1 << ctx.tab_shift;
        const size_t plan_words = (kPlanHeader + sizeof(DevSegment)) / 4;
        static_assert((kPlanHeader + sizeof(DevSegment)) % 4 == 0, "plan header in words");
        const size_t stage_words = plan_words + esz + ctx.NS;
        if (h_begin_words < stage_words) {
            if (h_begin) (void)hipHostFree(h_begin);
            HIPCHK(hipHostMalloc((void **)&h_begin, stage_words * sizeof(uint32_t)));
            h_begin_words = stage_words;
        }
        uint32_t *key = h_begin + plan_words, *slotw = key + esz - 2, *node = key + esz;
        unsigned long long h = 0;
        if (opt.rank == 0) {
            // root state: Signature({}, 0) (solveralgorithm.cpp:951-954) = local state 0 of shard 0.
            // With an empty signature a leaf of set 0 must find it again, so the key is the plain
            // (tag 0); otherwise a reserved tag keeps it apart from a state with an all-zero signature.
            // the root's table entry: [key ... | slot word {generation, state 0}]
            for (size_t i = 0; i < esz; i++) key[i] = 0u;
            key[0] = ctx.sig_len == 0 ? 0u : kRootTag;
            h = key_hash(key, ctx.KL);
            unsigned long long slot = ((unsigned long long)ctx.tab_gen << 32) | 0u;
            memcpy(slotw, &slot, sizeof slot);
            n_states = 1;
            // root search node: initial domains at every point (variable.cpp:24-29), set 0
            for (int i = 0; i < ctx.NS; i++) node[i] = 0u;
            for (int c = 0; c < ctx.W; c++)
                for (int p = 0; p < ctx.K; p++)
                    for (int v = 0; v < ctx.N; v++) node[4 + c * ctx.N * ctx.K + p * ctx.N + v] = init_chunk(v, c);
            h_plan->sp = 1;
            h_plan->stack[0].base = 0;
            h_plan->stack[0].cap = 1;
            h_plan->stack[0].count[0] = 1;
            h_plan->arena_top = (unsigned long long)R * ctx.NS;
            h_plan->open_total = 1;
        }
        h_plan->status = PS_DONE;
        sync_ctx();
        h_plan->arena_words = d_arena.n;  // (what push_caps() sends after a pool has grown)
        h_plan->slot_cap = (unsigned long long)ctx.slot_mask + 1;
        h_plan->edge_cap = ctx.edge_cap;
        h_plan->state_cap = ctx.state_cap;
        h_plan->cand_cap = cand_cap;
        memcpy(h_begin, h_plan, plan_words * 4);
        {
            BeginArgs ba{};
            ba.ctl = d_ctl.p;
            ba.ctl_words = L.words;
            ba.nstates_word = L.misc0 + MISC_NSTATES * CST;
            ba.stats = d_stats.p;
            ba.stats_words = kStatSlots * kStatWords;
            ba.sdeg = d_sdeg.p;
            ba.sdeg_words = d_sdeg.p ? (unsigned long long)d_sdeg.n : 0ull;
            ba.plan_dst = (uint32_t *)d_plan.p;
            ba.plan_words = (int)plan_words;
            ba.stage = h_begin;
            ba.entry_dst = opt.rank == 0 ? (uint32_t *)d_slots.p + ((size_t)((uint32_t)h & ctx.slot_mask) << ctx.tab_shift) : nullptr;
            ba.esz = (int)esz;
            ba.keys_dst = d_state_keys.p;
            ba.KL = ctx.KL;
            ba.node_dst = d_arena.p;
            ba.NS = ctx.NS;
            const unsigned long long work = std::max<unsigned long long>(ba.sdeg_words, (unsigned long long)ba.stats_words);
            hipLaunchKernelGGL(k_begin, dim3((unsigned)std::min<unsigned long long>(512, (work + 255) / 256 + 1)), dim3(256), 0, stream, ba);
            HIPCHK(hipGetLastError());
        }
        // (no synchronisation: everything above is ordered before the first launch on the stream, and the staging buffers
        // are not touched again before a later call has synchronised)
        begun = true;
        t_begin = std::chrono::steady_clock::now();
        return STCSP_OK;
    }

    double elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count(); }

    int read_ctl() {
        HIPCHK(hipMemcpyAsync(h_ctl, d_ctl.p, L.words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        return parse_ctl();
    }
    int parse_ctl() {
        for (int r = 0; r < R; r++) edge_count[r] = h_ctl[L.edge0 + r * CST];
        n_states = h_ctl[L.misc0 + MISC_NSTATES * CST];
        uint32_t e = h_ctl[L.misc0 + MISC_ERROR * CST];
        if (e) {
            static const char *names[] = {"", "watchdog", "table spin", "edge overflow", "state overflow", "unknown set",
                                          "empty domain", "frontier overflow", "candidate overflow",
                                          "frontier overflow in adopt", "frontier overflow in commit", "frontier overflow at a slot's end"};
            return fail(e == ERR_WATCHDOG ? STCSP_E_INTERNAL : STCSP_E_NOMEM, "device reported error %u (%s)", e, e < 12 ? names[e] : "?");
        }
        return STCSP_OK;
    }

    // kernel variant: whole image in LDS or not (L), compacted sweeps (sets with > kCompactSweepItems small
    // items: CS), no general wavefront revision (LITE)
    // (the instrumented build, -DSTCSP_PHASES, cannot be compiled for <4, image in LDS, general>: hipcc 7.2 stops with "Illegal
    // instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base" in those instantiations; there the same programs run the
    // partly-staged kernels instead. The product build has them all.)
#ifdef STCSP_PHASES
#define STCSP_VARIANT(DRT, V) (((DRT) == 4 && ((V) & 5) == 4) ? ((V) & ~4) : (V))
#else
#define STCSP_VARIANT(DRT, V) (V)
#endif
    template <int DRT, typename F>
    void with_variant(F &&f) const {
        const int v = (img_in_lds || big ? 4 : 0) | (compact_sweeps ? 2 : 0) | (lite ? 1 : 0);
        switch (v) {
            case 0: f(std::integral_constant<int, 0>{}); break;
            case 1: f(std::integral_constant<int, 1>{}); break;
            case 2: f(std::integral_constant<int, 2>{}); break;
            case 3: f(std::integral_constant<int, 3>{}); break;
            case 4: f(std::integral_constant<int, 4>{}); break;
            case 5: f(std::integral_constant<int, 5>{}); break;
            case 6: f(std::integral_constant<int, 6>{}); break;
            default: f(std::integral_constant<int, 7>{}); break;
        }
    }
    template <int DRT>
    const void *expand_fn() const {
        const void *fn = nullptr;
        if (mgr.W == 2) return (const void *)k_expand<DRT, false, false, false, false, 2>;
        if (mgr.W > 2) return (const void *)k_expand<DRT, false, false, false, false, 4>;
#ifdef STCSP_PHASES
        if constexpr (DRT != 4)  // (see STCSP_VARIANT)
#endif
        if (prefix_complete) return (const void *)k_expand<DRT, 2, false, false>;
        with_variant<DRT>([&](auto v) {
            constexpr int V = STCSP_VARIANT(DRT, decltype(v)::value);
            fn = (const void *)k_expand<DRT, (V & 4) != 0, (V & 2) != 0, (V & 1) != 0>;
            if constexpr ((V & 5) == 5)
                if (big) fn = (const void *)k_expand<DRT, true, (V & 2) != 0, true, true>;
        });
        return fn;
    }
    template <int DRT>
    const void *probe_fn() const {
        const void *fn = nullptr;
        with_variant<DRT>([&](auto v) {
            constexpr int V = STCSP_VARIANT(DRT, decltype(v)::value);
            fn = (const void *)k_probe<DRT, (V & 4) != 0, (V & 2) != 0, (V & 1) != 0>;
        });
        return fn;
    }
    template <int DRT>
    void launch_expand() {
        const Ctx *cp = (const Ctx *)d_ctx.p;
        if (mgr.W == 2) {
            hipLaunchKernelGGL((k_expand<DRT, false, false, false, false, 2>), dim3(max_blocks), dim3(256), lds_bytes, stream, cp, (const Plan *)d_plan.p, launch_seq++, ctx.tab_gen);
            return;
        }
        if (mgr.W > 2) {
            hipLaunchKernelGGL((k_expand<DRT, false, false, false, false, 4>), dim3(max_blocks), dim3(256), lds_bytes, stream, cp, (const Plan *)d_plan.p, launch_seq++, ctx.tab_gen);
            return;
        }
#ifdef STCSP_PHASES
        if constexpr (DRT != 4)
#endif
        if (prefix_complete) {
            hipLaunchKernelGGL((k_expand<DRT, 2, false, false>), dim3(max_blocks), dim3(256), lds_bytes, stream, cp, (const Plan *)d_plan.p, launch_seq++, ctx.tab_gen);
            return;
        }
        with_variant<DRT>([&](auto v) {
            constexpr int V = STCSP_VARIANT(DRT, decltype(v)::value);
            if constexpr ((V & 5) == 5)
                if (big) {
                    hipLaunchKernelGGL((k_expand<DRT, true, (V & 2) != 0, true, true>), dim3(max_blocks), dim3(STCSP_BIG_WAVES * 64), lds_bytes, stream, cp, (const Plan *)d_plan.p, launch_seq++, ctx.tab_gen);
                    return;
                }
            hipLaunchKernelGGL((k_expand<DRT, (V & 4) != 0, (V & 2) != 0, (V & 1) != 0>), dim3(max_blocks), dim3(256), lds_bytes, stream, cp, (const Plan *)d_plan.p, launch_seq++, ctx.tab_gen);
        });
    }

    // The domains a fresh time point starts from, per constraint set (FlatProgram::set_fresh_init): the set's constraints applied
    // to the plain initial domains at every point (k_probe: one block per set, every until constraint taken as expired), of which
    // the last point's words are kept. Sound for every state under the set: its earlier points hold subsets of the initial
    // domains, so whatever the arcs from them leave of the new point is a subset of what they leave here. Run after every
    // upload of a program with sets that have no row yet (creation, a translation stop, a set import).
    bool own_init_rows = false;
    std::map<int32_t, std::vector<uint32_t>> fresh_rows;  // set tag -> its row
    int fresh_init() {
        if (!own_init_rows || mgr.W != 1) return STCSP_OK;
        std::vector<int> todo;
        for (size_t si = 0; si < prog.sets.size(); si++)
            if (prog.set_fresh_init[si] && !fresh_rows.count(mgr.sets[si]->tag)) todo.push_back((int)si);
        if (todo.empty()) return STCSP_OK;
        const int N = ctx.N, K = ctx.K, ns = (int)prog.sets.size();
        bool all_fixed = true;
        for (int v = 0; v < N; v++) all_fixed = all_fixed && mgr.lb[v] == mgr.ub[v];
        std::vector<uint32_t> blocks((size_t)ns * ctx.NK);
        for (int si = 0; si < ns; si++)
            for (int p = 0; p < K; p++)
                for (int v = 0; v < N; v++) blocks[(size_t)si * ctx.NK + (size_t)p * N + v] = init_chunk(v, 0);
        std::vector<int> outcome((size_t)ns, (int)OC_FAIL);
        if (!all_fixed) {  // (a block of singletons is a leaf: nothing to gain, and its transition look-up is not for here)
            HIPCHK(hipSetDevice(device));
            int rc = flush_ctx();
            if (rc != STCSP_OK) return rc;
            DevBuf<uint32_t> d_blk;
            DevBuf<int> d_out;
            HIPCHK(d_blk.upload(blocks));
            HIPCHK(d_out.alloc((size_t)ns));
            const unsigned grid = (unsigned)std::min<int64_t>((ns + 3) / 4, max_blocks);
            switch (DR) {
                case 1: launch_probe<1>(grid, d_blk.p, ns, -1, 0xffffffffu, d_out.p); break;
                case 2: launch_probe<2>(grid, d_blk.p, ns, -1, 0xffffffffu, d_out.p); break;
                default: launch_probe<4>(grid, d_blk.p, ns, -1, 0xffffffffu, d_out.p); break;
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(blocks.data(), d_blk.p, blocks.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipMemcpyAsync(outcome.data(), d_out.p, (size_t)ns * sizeof(int), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        for (int si : todo) {
            std::vector<uint32_t> row((size_t)N);
            for (int v = 0; v < N; v++) row[v] = outcome[si] == (int)OC_FAIL ? init_chunk(v, 0) : blocks[(size_t)si * ctx.NK + (size_t)(K - 1) * N + v];
            fresh_rows[mgr.sets[si]->tag] = std::move(row);
        }
        // the whole table (at most 4,096 words) in one copy: plain rows for the sets that keep them, as upload_program laid them out
        std::vector<uint32_t> table((size_t)ns * N);
        for (int si = 0; si < ns; si++) {
            auto known = prog.set_fresh_init[si] ? fresh_rows.find(mgr.sets[si]->tag) : fresh_rows.end();
            for (int v = 0; v < N; v++) table[(size_t)si * N + v] = known != fresh_rows.end() ? known->second[v] : init_chunk(v, 0);
        }
        HIPCHK(hipMemcpyAsync(d_img.p + ctx.o.var_init, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIPCHK(hipStreamSynchronize(stream));  // (table is a local)
        return STCSP_OK;
    }

    // stcsp_engine_propagate: process_node on caller-provided blocks (k_probe)
    template <int DRT>
    void launch_probe(unsigned grid, uint32_t *blocks, int n, int set, uint32_t expire, int *outcome) {
        const Ctx *cp = (const Ctx *)d_ctx.p;
        with_variant<DRT>([&](auto v) {
            constexpr int V = STCSP_VARIANT(DRT, decltype(v)::value);
            hipLaunchKernelGGL((k_probe<DRT, (V & 4) != 0, (V & 2) != 0, (V & 1) != 0>), dim3(grid), dim3(256), lds_bytes, stream, cp, blocks, n, set,
                               expire, outcome);
        });
    }
    int propagate(int set, uint32_t expire, uint32_t *blocks, int64_t count, int32_t *outcome, int64_t *skipped) {
        if (sharded) return fail(STCSP_E_STATE, "propagate is for unsharded engines");
        if (mgr.W > 1) return fail(STCSP_E_UNSUPPORTED, "propagate: the node-level seam takes one-word blocks (every domain <= 32 values)");
        if (set < 0 || set >= (int)prog.sets.size() || count < 0 || count > (1 << 24) || !blocks || !outcome)
            return fail(STCSP_E_INVALID, "propagate: bad arguments (set %d of %zu, count %lld)", set, prog.sets.size(), (long long)count);
        if (count == 0) return STCSP_OK;
        // a solve in progress (stepping interface) owns the control block and the statistics; a finished one keeps its
        // cursors (export may still follow): only the error / miss words and the work counters are reset here
        if (begun && !finished) return fail(STCSP_E_STATE, "propagate inside a solve that has not finished");
        tail_fresh = false;
        HIPCHK(hipSetDevice(device));
        int rc = flush_ctx();
        if (rc != STCSP_OK) return rc;
        DevBuf<uint32_t> d_blk;
        DevBuf<int> d_out;
        const size_t words = (size_t)count * ctx.NK;
        HIPCHK(d_blk.alloc(words));
        HIPCHK(d_out.alloc((size_t)count));
        HIPCHK(hipMemcpyAsync(d_blk.p, blocks, words * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        HIPCHK(hipMemsetAsync(d_stats.p, 0, kStatSlots * kStatWords * sizeof(unsigned long long), stream));
        HIPCHK(hipMemsetAsync(d_ctl.p + L.misc0 + MISC_ERROR * CST, 0, sizeof(uint32_t), stream));
        HIPCHK(hipMemsetAsync(d_ctl.p + L.misc0 + MISC_NMISS * CST, 0, sizeof(uint32_t), stream));
        const unsigned grid = (unsigned)std::min<int64_t>((count + 3) / 4, max_blocks);
        switch (DR) {
            case 1: launch_probe<1>(grid, d_blk.p, (int)count, set, expire, d_out.p); break;
            case 2: launch_probe<2>(grid, d_blk.p, (int)count, set, expire, d_out.p); break;
            default: launch_probe<4>(grid, d_blk.p, (int)count, set, expire, d_out.p); break;
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(blocks, d_blk.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(outcome, d_out.p, (size_t)count * sizeof(int), hipMemcpyDeviceToHost, stream));
        // a probed leaf may have asked for a constraint-set translation: that request is not part of a solve
        HIPCHK(hipMemsetAsync(d_ctl.p + L.misc0 + MISC_NMISS * CST, 0, sizeof(uint32_t), stream));
        HIPCHK(hipStreamSynchronize(stream));
        if ((rc = read_ctl())) return rc;  // device errors (watchdog ...)
        if (skipped) {
            stcsp_counters ctr{};
            if ((rc = read_counters(ctr))) return rc;
            *skipped = ctr.skipped_revisions;
        }
        return STCSP_OK;
    }

    // ---- streaming export (unsharded solves that will be exported): the edge log is transposed into the result
    // arrays and copied to the host chunk by chunk on a second stream WHILE the search runs (the copy of
    // partialorder_14's 95 MB is 1.7 ms at PCIe speed: more than a third of its whole solve when done afterwards)
    hipStream_t xstream = nullptr, xstream2 = nullptr;  // chunks alternate between the two: one transposes while the other's copies are on the link
    hipEvent_t ev_x[2] = {nullptr, nullptr};            // ... recorded behind the last transposition kernel of each
    hipEvent_t ev_c[2] = {nullptr, nullptr};            // ... and behind the D2H copies of its last chunk
    hipEvent_t ev_k = nullptr;                          // ... and behind the last key-splitting kernel (xstream)
    bool ev_k_used = false;
    bool ev_x_used[2] = {false, false};
    DevBuf<uint32_t> d_sdeg;                            // out-degree of every state, counted by the transposition kernels as the log streams out
    unsigned chunk_no = 0;
    int sync_xstreams() {
        if (xstream) HIPCHK(hipStreamSynchronize(xstream));
        if (xstream2) HIPCHK(hipStreamSynchronize(xstream2));
        return STCSP_OK;
    }
    bool streaming = false;
    size_t streamed = 0;                 // edge records staged so far
    uint32_t streamed_r[R] = {0};        // ... per region of the edge log
    size_t stream_chunk_min = 32768;     // records per chunk (except the last)
    size_t stream_chunk_idle = 2048;     // ... a smaller chunk goes when both export streams are idle
    int ensure_export_capacity(size_t E) {
        const int N = ctx.N;
        if (d_osrc.n < E) {
            const size_t cap = std::max(E + E / 4 + 256, d_osrc.n * 2);
            DevBuf<long long> ns, nd;
            DevBuf<int32_t> nv;
            HIPCHK(ns.alloc(cap));
            HIPCHK(nd.alloc(cap));
            HIPCHK(nv.alloc(cap * N));
            { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }
            if (streamed) {
                HIPCHK(hipMemcpy(ns.p, d_osrc.p, streamed * sizeof(long long), hipMemcpyDeviceToDevice));
                HIPCHK(hipMemcpy(nd.p, d_odst.p, streamed * sizeof(long long), hipMemcpyDeviceToDevice));
                HIPCHK(hipMemcpy(nv.p, d_oval.p, streamed * N * sizeof(int32_t), hipMemcpyDeviceToDevice));
            }
            std::swap(d_osrc.p, ns.p);
            std::swap(d_osrc.n, ns.n);
            std::swap(d_odst.p, nd.p);
            std::swap(d_odst.n, nd.n);
            std::swap(d_oval.p, nv.p);
            std::swap(d_oval.n, nv.n);
        }
        if (h_edge_cap < E) {
            const size_t cap = std::max(E + E / 4 + 256, h_edge_cap * 2);
            long long *ns = nullptr, *nd = nullptr;
            int32_t *nv = nullptr;
            HIPCHK(hipHostMalloc((void **)&ns, cap * sizeof(long long)));
            HIPCHK(hipHostMalloc((void **)&nd, cap * sizeof(long long)));
            HIPCHK(hipHostMalloc((void **)&nv, cap * N * sizeof(int32_t)));
            { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }
            if (streamed) {
                memcpy(ns, h_osrc, streamed * sizeof(long long));
                memcpy(nd, h_odst, streamed * sizeof(long long));
                memcpy(nv, h_oval, streamed * N * sizeof(int32_t));
            }
            if (h_osrc) (void)hipHostFree(h_osrc);
            if (h_odst) (void)hipHostFree(h_odst);
            if (h_oval) (void)hipHostFree(h_oval);
            h_osrc = ns;
            h_odst = nd;
            h_oval = nv;
            h_edge_cap = cap;
        }
        return STCSP_OK;
    }
    // Stage and ship the edge records logged since the last call. `upto[r]` = cursor of region r as of a launch that has
    // ENDED: either the main stream is synchronised (calls between bursts, final call), or the snapshot of launch g - 1 is
    // handed over only once launch g has been seen to finalize (ship_progress, mid-burst).
    int stream_edges(const uint32_t *upto, bool final) {
        if (!streaming) return STCSP_OK;
        StreamView v{};
        size_t M = 0;
        uint32_t to[R];
        for (int r = 0; r < R; r++) {
            to[r] = std::max(upto[r], streamed_r[r]);  // (an older snapshot than the last one shipped: nothing to do)
            v.from[r] = streamed_r[r];
            v.pref[r] = (uint32_t)M;
            M += to[r] - streamed_r[r];
        }
        v.pref[R] = (uint32_t)M;
        if (M == 0) return STCSP_OK;
        if (!final && M < stream_chunk_min) {
            // a small chunk waits for more -- unless the link has nothing to do (the narrow rounds at the end of a search: what is
            // shipped now is not left for after the search)
            if (M < stream_chunk_idle || hipStreamQuery(xstream) != hipSuccess || hipStreamQuery(xstream2) != hipSuccess) return STCSP_OK;
        }
        if (streamed + M > 0xfffffff0ull) {  // beyond the 32-bit record indices of the export kernels: compacting path decides
            streaming = false;
            return STCSP_OK;
        }
        int rc = ensure_export_capacity(streamed + M);
        if (rc != STCSP_OK) return rc;
        if (!sharded && d_sdeg.n < ctx.state_cap) return fail(STCSP_E_INTERNAL, "out-degree mirror smaller than the state pool");
        v.edges = d_edges.p;
        v.edge_cap = ctx.edge_cap;
        v.ES = ctx.ES;
        v.N = ctx.N;
        const int N = ctx.N;
        const int xi = (int)(chunk_no++ & 1u);
        hipStream_t xs = xi ? xstream2 : xstream;  // (chunks touch disjoint ranges of the arrays)
        // the last chunk of an export, when small: the kernel writes the host arrays itself (see k_stream_edges)
        const bool zero_copy = final && M <= 8192 && !(getenv("STCSP_STREAM_ZC") && atoi(getenv("STCSP_STREAM_ZC")) == 0);
        hipLaunchKernelGGL(k_stream_edges, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, xs, v, (unsigned long long)streamed,
                           sharded ? (unsigned long long)opt.rank << STCSP_GID_SHIFT : 0ull, d_osrc.p, d_odst.p, d_oval.p, sharded ? nullptr : d_sdeg.p,
                           zero_copy ? h_osrc : nullptr, zero_copy ? h_odst : nullptr, zero_copy ? h_oval : nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev_x[xi], xs));
        ev_x_used[xi] = true;
        if (zero_copy) {
            HIPCHK(hipEventRecord(ev_c[xi], xs));
            for (int r = 0; r < R; r++) streamed_r[r] = to[r];
            streamed += M;
            return STCSP_OK;
        }
        // (having the kernel write the pinned host arrays itself instead of these copies: search 3.05 -> 4.4 ms, the kernel then holds
        // its wave slots at the speed of the link -- measured again in round 3 with both HIP runtimes, see create())
        HIPCHK(hipMemcpyAsync(h_osrc + streamed, d_osrc.p + streamed, M * sizeof(long long), hipMemcpyDeviceToHost, xs));
        HIPCHK(hipMemcpyAsync(h_odst + streamed, d_odst.p + streamed, M * sizeof(long long), hipMemcpyDeviceToHost, xs));
        HIPCHK(hipMemcpyAsync(h_oval + streamed * N, d_oval.p + streamed * N, M * N * sizeof(int32_t), hipMemcpyDeviceToHost, xs));
        HIPCHK(hipEventRecord(ev_c[xi], xs));
        for (int r = 0; r < R; r++) streamed_r[r] = to[r];
        streamed += M;
        return STCSP_OK;
    }

    // ... and the state keys (split into the result's cid / sig arrays by the device, written straight into pinned memory)
    int32_t *h_cid = nullptr, *h_sig = nullptr;
    size_t h_key_cap = 0;
    uint32_t streamed_states = 0;
    int stream_states(uint32_t upto) {
        if (upto <= streamed_states) return STCSP_OK;
        const int sl = std::max(ctx.sig_len, 0);
        if (h_key_cap < upto) {
            const size_t cap = std::max<size_t>((size_t)upto + upto / 4 + 256, h_key_cap * 2);
            int32_t *nc = nullptr, *nsg = nullptr;
            HIPCHK(hipHostMalloc((void **)&nc, cap * sizeof(int32_t)));
            HIPCHK(hipHostMalloc((void **)&nsg, std::max<size_t>(cap * sl, 1) * sizeof(int32_t)));
            { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }
            if (streamed_states) {
                memcpy(nc, h_cid, (size_t)streamed_states * sizeof(int32_t));
                memcpy(nsg, h_sig, (size_t)streamed_states * sl * sizeof(int32_t));
            }
            if (h_cid) (void)hipHostFree(h_cid);
            if (h_sig) (void)hipHostFree(h_sig);
            h_cid = nc;
            h_sig = nsg;
            h_key_cap = cap;
        }
        const unsigned long long words = (unsigned long long)(upto - streamed_states) * ctx.KL;
        hipLaunchKernelGGL(k_stream_keys, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, xstream, (const uint32_t *)d_state_keys.p, ctx.KL,
                           ctx.sig_len, streamed_states, upto, h_cid, h_sig);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(ev_k, xstream));
        ev_k_used = true;
        streamed_states = upto;
        return STCSP_OK;
    }

    long long translation_stops = 0;
    int service_misses() {
        uint32_t nm = h_ctl[L.misc0 + MISC_NMISS * CST];
        if (!nm) return STCSP_OK;
        translation_stops++;
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
        const int rcu = upload_program();
        return rcu != STCSP_OK ? rcu : fresh_init();
    }

    // Enqueue bursts of rounds until the device plan stops: done, outbox full (sharded), or
    // truncated by a budget. Pool growth and constraint-set translation are served in between.
    int run_rounds() {
        int rc = replan();
        if (rc != STCSP_OK) return rc;
        const bool prof = opt.flags & STCSP_F_PROFILE;
        const long long rounds_at_entry = levels;
        uint32_t seen[R], seen_states = 0;
        bool have_seen = false;  // edge cursors of the last completed burst, not yet handed to stream_edges
        for (;;) {
            if ((rc = flush_ctx())) return rc;
            // STCSP_F_PROFILE: one pair of events per BURST (the launches of a burst are back to back on the stream;
            // a pair per launch costs two more packets between consecutive kernels and ~6 % of a partialorder_14 solve)
            if (prof) {
                if (ev_used == ev_pool.size()) {
                    hipEvent_t e0, e1;
                    HIPCHK(hipEventCreate(&e0));
                    HIPCHK(hipEventCreate(&e1));
                    ev_pool.emplace_back(e0, e1);
                }
                HIPCHK(hipEventRecord(ev_pool[ev_used].first, stream));
            }
            tail_fresh = false;
            ctl_fresh = false;
            for (int k = 0; k < burst; k++) {
                switch (DR) {
                    case 1: launch_expand<1>(); break;
                    case 2: launch_expand<2>(); break;
                    default: launch_expand<4>(); break;
                }
                HIPCHK(hipGetLastError());
                expand_launches++;
            }
            if (prof) HIPCHK(hipEventRecord(ev_pool[ev_used++].second, stream));
            // the device is busy with the burst just enqueued: now is the time to ship the edges of the previous ones
            if (have_seen && ((rc = stream_edges(seen, false)) || (streaming && (rc = stream_states(seen_states))))) return rc;
            rc = read_plan(true);
            if (rc != STCSP_OK) return rc;
            for (int r = 0; r < R; r++) seen[r] = h_plan->edge_seen[r];
            seen_states = h_plan->states_seen;
            have_seen = true;
            if (dbg_rounds && burst == 1) {
                std::vector<unsigned long long> st(kStatSlots * kStatWords);
                HIPCHK(hipMemcpy(st.data(), d_stats.p, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                unsigned long long nodes = 0;
                for (int sl = 0; sl < kStatSlots; sl++) nodes += st[sl * kStatWords + ST_NODES];
                dbg_nodes.push_back(nodes);
                dbg_open.push_back((long long)h_plan->open_total);
            }
            switch (h_plan->status) {
                case PS_RUN: break;
                case PS_DONE:
                case PS_OUTBOX_FULL: return STCSP_OK;
                case PS_NEED_ARENA: {
                    size_t need = (size_t)h_plan->arena_top + (size_t)R * (std::max(chain_small, chain_big) + 2) * chunk_r * ctx.NS;
                    // past the soft limit an automatic batch shrinks first (deeper, narrower search:
                    // memory ~ depth x batch) and the arena only grows if that is not enough
                    while (auto_batch && need > arena_soft_words && chunk_r > 2048) {
                        chunk_r /= 2;
                        h_plan->chunk_r = chunk_r;
                        HIPCHK(hipMemcpyAsync(&d_plan.p->chunk_r, &h_plan->chunk_r, sizeof(int), hipMemcpyHostToDevice, stream));
                        need = (size_t)h_plan->arena_top + (size_t)R * (std::max(chain_small, chain_big) + 2) * chunk_r * ctx.NS;
                    }
                    if (need > d_arena.n && (rc = grow_arena(need))) return rc;
                    if ((rc = push_caps()) || (rc = replan())) return rc;
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
                return last_plan_fast ? read_plan(false) : STCSP_OK;
            }
            // sharded stepping with a budget: hand control back while there is still work to share
            if (sharded && step_max_rounds > 0 && levels - rounds_at_entry >= step_max_rounds && h_plan->open_total >= step_min_open)
                return last_plan_fast ? read_plan(false) : STCSP_OK;  // (commit / donate / adopt work from the whole plan header)
        }
    }

    bool over_budget() {
        if (opt.time_limit_s > 0 && elapsed() > opt.time_limit_s) return true;
        if (opt.max_search_nodes > 0) {
            // cheap upper bound without a device read: every launch expands at most chain*R*chunk_r nodes
            if (levels * (long long)R * chunk_r0 * std::max(chain_small, chain_big) >= opt.max_search_nodes) {
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

    int solve_unsharded() {
        const bool dbg = getenv("STCSP_DEBUG") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = begin();
        if (rc != STCSP_OK) return rc;
        const auto t1 = std::chrono::steady_clock::now();
        rc = run_rounds();
        if (rc != STCSP_OK) return rc;
        const auto t2 = std::chrono::steady_clock::now();
        rc = finish();
        if (dbg)
            fprintf(stderr, "[solve] begin %.3f ms, rounds %.3f ms, finish %.3f ms\n", std::chrono::duration<double>(t1 - t0).count() * 1e3,
                    std::chrono::duration<double>(t2 - t1).count() * 1e3, std::chrono::duration<double>(std::chrono::steady_clock::now() - t2).count() * 1e3);
        return rc;
    }

    uint32_t *h_begin = nullptr;            // pinned staging of what begin() uploads (root key, its slot, the root node)
    size_t h_begin_words = 0;
    unsigned long long *h_stats = nullptr;  // pinned mirror of the device statistics
    bool stats_fresh = false;               // ... already copied by the caller (finish(): one synchronisation for everything)
    int read_counters(stcsp_counters &ctr) {
        if (!h_stats) HIPCHK(hipHostMalloc((void **)&h_stats, kStatSlots * kStatWords * sizeof(unsigned long long)));
        if (!stats_fresh) {
            HIPCHK(hipMemcpyAsync(h_stats, d_stats.p, kStatSlots * kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        stats_fresh = false;
        const unsigned long long *st = h_stats;
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
        ctr.translation_stops = translation_stops;
#ifdef STCSP_PHASES
        if (tot[ST_NODES])
            fprintf(stderr, "[phases] cycles/node: node load %.0f | process_node %.0f (of which sweeps %.0f, wavefront revisions %.0f) | emit/commit %.0f | total %.0f (nodes %llu)\n",
                    (double)tot[ST_CYC_LOAD] / tot[ST_NODES], (double)tot[ST_CYC_CLASSIFY] / tot[ST_NODES], (double)tot[ST_CYC_SWEEP] / tot[ST_NODES],
                    (double)tot[ST_CYC_WAVE] / tot[ST_NODES], (double)tot[ST_CYC_COMMIT] / tot[ST_NODES],
                    (double)tot[ST_CYC_TOTAL] / tot[ST_NODES], (unsigned long long)tot[ST_NODES]);
        if (tot[ST_ROOTS])
            fprintf(stderr, "[phases] first nodes of states: %llu, sweeps %.0f + wavefront revisions %.0f cycles each (all nodes: %.0f + %.0f)\n", (unsigned long long)tot[ST_ROOTS],
                    (double)tot[ST_CYC_ROOT_SWEEP] / tot[ST_ROOTS], (double)tot[ST_CYC_ROOT_WAVE] / tot[ST_ROOTS], (double)tot[ST_CYC_SWEEP] / tot[ST_NODES], (double)tot[ST_CYC_WAVE] / tot[ST_NODES]),
            fprintf(stderr, "[phases] ... per first node: %.2f batches (%.0f cycles in them), %.2f refused scans, %.1f odometer steps\n",
                    (double)tot[ST_ROOT_BATCHES] / tot[ST_ROOTS], (double)tot[ST_ROOT_CYC_BATCH] / tot[ST_ROOTS], (double)tot[ST_ROOT_REFUSED] / tot[ST_ROOTS], (double)tot[ST_ROOT_RV_BLOCKS] / tot[ST_ROOTS]);
        if (tot[ST_WAVEREVS])
            fprintf(stderr, "[phases] per wavefront revision (completed ones): setup %.0f, enumeration %.0f, write-back %.0f cycles; %.2f revisions per node\n",
                    (double)tot[ST_CYC_RV_SETUP] / tot[ST_WAVEREVS], (double)tot[ST_CYC_RV_LOOP] / tot[ST_WAVEREVS], (double)tot[ST_CYC_RV_WB] / tot[ST_WAVEREVS],
                    (double)tot[ST_WAVEREVS] / tot[ST_NODES]);
        if (tot[ST_RV_BLOCKS])
            fprintf(stderr, "[phases] general wavefront revisions: %.2f tuple blocks, %.2f open variables, %.1f tuple lanes each (%.2f per node incl. fast-path ones)\n",
                    (double)tot[ST_RV_BLOCKS] / std::max(1.0, (double)tot[ST_RV_LANES] > 0 ? (double)tot[ST_CYC_RV_SETUP] > 0 ? (double)tot[ST_WAVEREVS] : 1.0 : 1.0),
                    (double)tot[ST_RV_OPEN] / std::max(1.0, (double)tot[ST_WAVEREVS]), (double)tot[ST_RV_LANES] / std::max(1.0, (double)tot[ST_WAVEREVS]),
                    (double)tot[ST_WAVEREVS] / tot[ST_NODES]);
        if (tot[ST_RV_BLOCKS])
            fprintf(stderr, "[phases] per tuple block: digits %.0f cycles; evaluation: bitmap %.0f (x %llu blocks), bytecode %.0f (x %llu blocks); support sets %.0f\n",
                    (double)tot[ST_CYC_RV_DIGITS] / tot[ST_RV_BLOCKS],
                    (double)tot[ST_CYC_RV_EVAL_BITMAP] / std::max<double>(1.0, (double)(tot[ST_RV_BLOCKS] - tot[ST_RV_BLOCKS_CODE])),
                    (unsigned long long)(tot[ST_RV_BLOCKS] - tot[ST_RV_BLOCKS_CODE]),
                    (double)tot[ST_CYC_RV_EVAL_CODE] / std::max<double>(1.0, (double)tot[ST_RV_BLOCKS_CODE]), (unsigned long long)tot[ST_RV_BLOCKS_CODE],
                    (double)tot[ST_CYC_RV_SUPPORT] / tot[ST_RV_BLOCKS]);
        if (tot[ST_BATCHES] + tot[ST_BATCH_REFUSED])
            fprintf(stderr, "[phases] batched revisions: %.2f batches per node, %.2f items and %.1f tuple lanes each, %.0f cycles each (scan %.0f, tuples %.0f); %.2f refused scans per node (%.0f cycles each)\n",
                    (double)tot[ST_BATCHES] / tot[ST_NODES], (double)tot[ST_BATCH_ITEMS] / std::max<double>(1.0, (double)tot[ST_BATCHES]),
                    (double)tot[ST_BATCH_TUPLES] / std::max<double>(1.0, (double)tot[ST_BATCHES]),
                    (double)tot[ST_CYC_BATCH] / std::max<double>(1.0, (double)tot[ST_BATCHES]),
                    (double)tot[ST_CYC_BATCH_AB] / std::max<double>(1.0, (double)(tot[ST_BATCHES] + tot[ST_BATCH_REFUSED])),
                    (double)tot[ST_CYC_BATCH_DE] / std::max<double>(1.0, (double)tot[ST_BATCHES]),
                    (double)tot[ST_BATCH_REFUSED] / tot[ST_NODES],
                    (double)tot[ST_CYC_BATCH_AB] / std::max<double>(1.0, (double)(tot[ST_BATCHES] + tot[ST_BATCH_REFUSED])));
        if (tot[ST_NODES])
            fprintf(stderr, "[phases] cycles/node: closures of the next arcs %.0f, leaf part of process_node (transition, signature, hash, time shift) %.0f\n",
                    (double)tot[ST_CYC_CLOSE] / tot[ST_NODES], (double)tot[ST_CYC_LEAF] / tot[ST_NODES]);
        if (tot[ST_BLOCKS] && tot[ST_ROUNDS_FINAL])
            fprintf(stderr, "[phases] per working workgroup: image staging %.0f cycles, whole %.0f cycles (%llu workgroup runs); finalize_round %.0f cycles x %llu rounds\n",
                    (double)tot[ST_CYC_STAGE] / tot[ST_BLOCKS], (double)tot[ST_CYC_BLOCK] / tot[ST_BLOCKS], (unsigned long long)tot[ST_BLOCKS],
                    (double)tot[ST_CYC_FINAL] / tot[ST_ROUNDS_FINAL], (unsigned long long)tot[ST_ROUNDS_FINAL]);
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
            if (dbg_rounds) {  // STCSP_DEBUG=2 with STCSP_BURST=1: one line per launch
                const unsigned long long n1 = i < dbg_nodes.size() ? dbg_nodes[i] : 0, n0 = i ? dbg_nodes[i - 1] : 0;
                fprintf(stderr, "[round] %3zu  %8.1f us  %8llu nodes  open after %lld\n", i, ms * 1e3, n1 - n0, i < dbg_open.size() ? dbg_open[i] : -1);
            }
        }
        dbg_nodes.clear();
        dbg_open.clear();
        ev_used = 0;
        // the control block and the statistics: already here when the last burst's read_plan brought them and nothing has run
        // since; else in one go (each synchronous small copy costs ~100 us of host time)
        int rc;
        if (tail_fresh) {
            rc = parse_ctl();
        } else {
            if (!h_stats) HIPCHK(hipHostMalloc((void **)&h_stats, kStatSlots * kStatWords * sizeof(unsigned long long)));
            HIPCHK(hipMemcpyAsync(h_stats, d_stats.p, kStatSlots * kStatWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            rc = read_ctl();
        }
        if (rc != STCSP_OK) return rc;
        stats_fresh = true;
        return read_counters(snap);
    }

    // tests: STCSP_FAULT="<call>,<rank>[,<nth>]" makes the nth invocation (default 1st) of one stepping call fail on one
    // rank -- what the failure protocol of the sharded drivers is tested with (no rank may be left waiting in a collective)
    std::string fault_call;
    int fault_left = 0;
    bool fault_hit(const char *call) {
        if (fault_call != call || fault_left <= 0) return false;
        return --fault_left == 0;
    }
    // ---- sharded stepping
    int expand_local(int64_t *left) {
        if (!begun) return fail(STCSP_E_STATE, "expand_local before begin");
        if (fault_hit("expand_local")) return fail(STCSP_E_INTERNAL, "injected fault in expand_local");
        packed = false;
        const auto t0 = std::chrono::steady_clock::now();
        int rc = run_rounds();
        if (rc != STCSP_OK) return rc;
        if (getenv("STCSP_DEBUG"))
            fprintf(stderr, "[expand_local] %.3f ms, %lld rounds so far, streamed %zu\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3, levels, streamed);
        rc = ctl_fresh ? parse_ctl() : read_ctl();  // outbox cursors for outbox() (the last burst's copying read_plan has them already)
        if (rc != STCSP_OK) return rc;
        host_view_fresh = true;
        if (left) *left = truncated ? 0 : (int64_t)h_plan->open_total;
        return STCSP_OK;
    }
    // The first outbox() call after an expand_local packs the regions of EVERY peer (one kernel per
    // non-empty peer, one synchronisation), back to back in the pack buffer, so that the driver can hand
    // the whole buffer to one all-to-all-v; the other calls only read the cached slices.
    int pack_outboxes(bool sync = true) {
        if (d_pack.n < (size_t)opt.world * R * cand_cap * ctx.CS) HIPCHK(d_pack.alloc((size_t)opt.world * R * cand_cap * ctx.CS));
        pack_ptr.assign(opt.world, nullptr);
        pack_count.assign(opt.world, 0);
        size_t before = 0;
        uint32_t most = 0;
        PackAllArgs pa{};
        for (int peer = 0; peer < opt.world; peer++) {
            uint32_t total = 0;
            for (int r = 0; r < R; r++) total += h_ctl[L.cand0 + (peer * R + r) * CST];
            uint32_t *dst = d_pack.p + before * ctx.CS;
            if (opt.world <= 64) pa.before[peer] = (uint32_t)before;
            else if (total)  // (more owners than PackAllArgs holds: one launch each)
                hipLaunchKernelGGL(k_pack, dim3(std::min<uint32_t>(1024, (total * ctx.CS + 255) / 256)), dim3(256), 0, stream,
                                   d_cand.p + (size_t)peer * R * cand_cap * ctx.CS, cand_cap, ctx.CS, d_ctl.p, L.cand0 + peer * R * CST, dst);
            pack_ptr[peer] = dst;
            pack_count[peer] = total;
            before += total;
            most = std::max(most, total);
        }
        if (opt.world <= 64 && most)  // every owner's outbox in ONE launch (blockIdx.y = owner)
            hipLaunchKernelGGL(k_pack_all, dim3(std::min<uint32_t>(1024, (most * ctx.CS + 255) / 256), (unsigned)opt.world), dim3(256), 0, stream,
                               (const uint32_t *)d_cand.p, cand_cap, ctx.CS, (const uint32_t *)d_ctl.p, L.cand0, pa, d_pack.p);
        HIPCHK(hipGetLastError());
        if (sync) HIPCHK(hipStreamSynchronize(stream));  // (a driver that moves the records on another stream; the native loop's transports are ordered on this one)
        packed = true;
        return STCSP_OK;
    }
    int outbox(int peer, void **ptr, int64_t *count, bool sync = true) {
        if (peer < 0 || peer >= opt.world) return fail(STCSP_E_INVALID, "peer out of range");
        if (!packed) {
            int rc = pack_outboxes(sync);
            if (rc != STCSP_OK) return rc;
        }
        *ptr = pack_ptr[peer];
        *count = pack_count[peer];
        return STCSP_OK;
    }
    int commit(const void *records, int64_t count) {
        ctl_fresh = false;
        if (!begun) return fail(STCSP_E_STATE, "commit before begin");
        if (fault_hit("commit")) return fail(STCSP_E_INTERNAL, "injected fault in commit");
        tail_fresh = false;
        // the outbox has been handed over: empty it
        packed = false;
        HIPCHK(hipMemsetAsync(d_ctl.p + L.cand0, 0, (size_t)(L.edge0 - L.cand0) * sizeof(uint32_t), stream));
        for (int i = L.cand0; i < L.edge0; i++) h_ctl[i] = 0;
        if (count <= 0) {
            HIPCHK(hipStreamSynchronize(stream));
            return STCSP_OK;
        }
        // room for `count` more edges / states and for the segment of new nodes. The host copies of
        // the cursors and the plan are those expand_local() just read (nothing ran in between).
        int rc = STCSP_OK;
        if (!host_view_fresh) {
            if ((rc = read_ctl()) != STCSP_OK) return rc;
            if ((rc = read_plan()) != STCSP_OK) return rc;
        }
        host_view_fresh = false;
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
        // not waited for: the next expand_local() (or finish()) synchronises and reads the cursors,
        // which is also where an overflow reported by k_commit surfaces. `records` must stay valid
        // until then (the driver keeps its receive buffer for the whole superstep).
        return STCSP_OK;
    }

    // ---- frontier redistribution (stcsp_engine.h): the oldest open nodes leave / received ones join
    int donate(int64_t want, void **ptr, int64_t *count) {
        ctl_fresh = false;
        if (!begun || !sharded) return fail(STCSP_E_STATE, "donate is part of the sharded stepping interface (after begin)");
        if (fault_hit("donate")) return fail(STCSP_E_INTERNAL, "injected fault in donate");
        tail_fresh = false;
        *ptr = nullptr;
        *count = 0;
        if (want <= 0) return STCSP_OK;
        int rc = STCSP_OK;
        if (!host_view_fresh && ((rc = read_ctl()) || (rc = read_plan()))) return rc;
        const int sp = h_plan->sp;
        if (sp <= 0) return STCSP_OK;
        h_stack.resize((size_t)sp);
        HIPCHK(hipMemcpy(h_stack.data(), &d_plan.p->stack[0], (size_t)sp * sizeof(DevSegment), hipMemcpyDeviceToHost));
        const int TS = xfer_stride(ctx.N, ctx.K * ctx.W);
        if (d_xfer.n < (size_t)want * TS) HIPCHK(d_xfer.alloc((size_t)want * TS));
        if ((rc = flush_ctx())) return rc;
        int64_t done = 0;
        for (int sg = 0; sg < sp && done < want; sg++) {  // bottom of the stack first: the shallowest nodes
            DonateArgs a{};
            a.seg_base = h_stack[sg].base;
            a.seg_cap = h_stack[sg].cap;
            a.seg = sg;
            int64_t avail = 0;
            for (int r = 0; r < R; r++) avail += (a.count[r] = h_stack[sg].count[r]);
            if (!avail) continue;
            int64_t need = std::min<int64_t>(want - done, avail);
            const int64_t share = need / R;
            for (int r = 0; r < R; r++) need -= (a.take[r] = (int)std::min<int64_t>(a.count[r], share));
            for (int r = 0; r < R && need > 0; r++) {  // what the even shares left over
                const int extra = (int)std::min<int64_t>(a.count[r] - a.take[r], need);
                a.take[r] += extra;
                need -= extra;
            }
            uint32_t acc = 0;
            for (int r = 0; r < R; r++) {
                a.pref[r] = acc;
                acc += (uint32_t)a.take[r];
            }
            a.pref[R] = acc;
            hipLaunchKernelGGL(k_donate, dim3((acc + 3) / 4), dim3(256), 0, stream, ctx, a, d_xfer.p + (size_t)done * TS);
            HIPCHK(hipGetLastError());
            done += acc;
        }
        HIPCHK(hipStreamSynchronize(stream));
        h_plan->open_total -= done;
        *ptr = d_xfer.p;
        *count = done;
        return STCSP_OK;
    }
    int adopt(const void *records, int64_t count) {
        ctl_fresh = false;
        if (!begun || !sharded) return fail(STCSP_E_STATE, "adopt is part of the sharded stepping interface (after begin)");
        if (fault_hit("adopt")) return fail(STCSP_E_INTERNAL, "injected fault in adopt");
        tail_fresh = false;
        if (count <= 0) return STCSP_OK;
        int rc = STCSP_OK;
        if (!host_view_fresh && ((rc = read_ctl()) || (rc = read_plan()))) return rc;  // (waits for a commit in flight)
        host_view_fresh = false;
        const unsigned cap = (unsigned)((count + R - 1) / R + 1);
        if ((size_t)h_plan->arena_top + (size_t)R * cap * ctx.NS > d_arena.n)
            if ((rc = grow_arena((size_t)h_plan->arena_top + (size_t)R * cap * ctx.NS))) return rc;
        if ((rc = push_caps())) return rc;
        hipLaunchKernelGGL(k_open_segment, dim3(1), dim3(64), 0, stream, ctx, cap);
        hipLaunchKernelGGL(k_adopt, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, stream, ctx, (const uint32_t *)records, (long long)count);
        hipLaunchKernelGGL(k_close_segment, dim3(1), dim3(64), 0, stream, ctx);
        HIPCHK(hipGetLastError());
        return STCSP_OK;  // not waited for (like commit): `records` must stay valid until the next expand_local / finish
    }

    // graphTraverse / adversarialTraverse / adversarialTraverse2 on the device (dev_postproc.hpp)
    int postprocess(const stcsp_post_options *po, stcsp_post_result *out) {
        if (sharded) return fail(STCSP_E_STATE, "device post-processing is for unsharded engines (merge shards on the host)");
        if (!exp_on_device) return fail(STCSP_E_STATE, "postprocess needs the device export of a finished solve (export first)");
        const int N = ctx.N;
        const int a1 = po ? po->adversarial_var : -1, op = po ? po->adversarial2_op : -1, ava = po ? po->adversarial2_ava : -1;
        if (a1 >= N || op >= N || ava >= N || a1 < -1 || op < -1 || (op >= 0 && ava < 0))
            return fail(STCSP_E_INVALID, "postprocess: variable index out of range");
        auto t0 = std::chrono::steady_clock::now();
        const size_t E = exp_edges;
        const uint32_t S = n_states;
        auto width = [&](int v) { return mgr.ub[v] - mgr.lb[v] + 1; };
        for (int v : {a1, op, ava})
            if (v >= 0 && width(v) > 32)
                return fail(STCSP_E_UNSUPPORTED, "device adversarial passes keep one 32-bit cover word per state: variable %d has %d values (use the host passes of stcsp_host.h)", v, width(v));
        auto full_mask = [&](int v) { return width(v) >= 32 ? 0xffffffffu : ((1u << width(v)) - 1u); };
        const int wa = op >= 0 ? width(ava) : 0;
        if (d_pvalid.n < S) {
            const size_t cap = (size_t)S + S / 4 + 256;
            HIPCHK(d_pvalid.alloc(cap));
            HIPCHK(d_pfinal.alloc(cap));
            HIPCHK(d_pnodeok.alloc(cap));
        }
        const size_t cover_words = (size_t)S * std::max(1, wa);
        if (d_pcover.n < cover_words) HIPCHK(d_pcover.alloc(cover_words + cover_words / 4 + 256));
        if (d_palive.n < E + 1) HIPCHK(d_palive.alloc(E + E / 4 + 256));
        if (!d_post.p) HIPCHK(d_post.alloc(4));
        const unsigned eb = (unsigned)((E + 255) / 256), sb = (S + 255) / 256;
        const long long *src = d_osrc.p, *dst = d_odst.p;
        const int32_t *val = d_oval.p;
        uint32_t *changed = d_post.p;
        // one round = the kernels `body` enqueues; returns the number of rounds until nothing changed
        auto fixpoint = [&](int &rounds, auto body) -> int {  // HIPCHK returns the error code from the enclosing lambda
            for (rounds = 0;; rounds++) {
                HIPCHK(hipMemsetAsync(changed, 0, sizeof(uint32_t), stream));
                const int rb = body();
                if (rb != STCSP_OK) return rb;
                uint32_t ch = 0;
                HIPCHK(hipMemcpyAsync(&ch, changed, sizeof ch, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                if (!ch) return STCSP_OK;
                if (rounds > (int)S + 8) return fail(STCSP_E_INTERNAL, "post-processing fixpoint did not converge");
            }
        };
        int rounds[3] = {0, 0, 0};
        HIPCHK(hipMemsetAsync(d_palive.p, 1, E + 1, stream));
        // graphTraverse (src/graph.cpp:357-418); the loop bound numSignVar + numUntil is the reference's
        hipLaunchKernelGGL(k_trav_init, dim3(sb), dim3(256), 0, stream, S, (const uint32_t *)d_state_keys.p, ctx.KL, mgr.n_sig,
                           mgr.n_sig + mgr.n_until, (int)(mgr.n_until_cons == 0), d_pvalid.p, d_pfinal.p);
        if (E) {
            int rc = fixpoint(rounds[0], [&] {
                hipLaunchKernelGGL(k_trav_back, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, (const uint8_t *)d_palive.p, d_pvalid.p, changed);
                return (int)STCSP_OK;
            });
            if (rc != STCSP_OK) return rc;
            hipLaunchKernelGGL(k_kill_into_invalid, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, d_palive.p, (const uint8_t *)d_pvalid.p, 1);
        }
        int adver1 = -1, adver2 = -1;
        uint8_t root_valid = 0;
        if (a1 >= 0) {  // adversarialTraverse (src/graph.cpp:304-355)
            const uint32_t full = full_mask(a1);
            int rc = fixpoint(rounds[1], [&] {
                HIPCHK(hipMemsetAsync(d_pcover.p, 0, (size_t)S * sizeof(uint32_t), stream));
                if (E)
                    hipLaunchKernelGGL(k_adv_cover, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, val, N, a1, mgr.lb[a1],
                                       (const uint8_t *)d_palive.p, (const uint8_t *)d_pvalid.p, d_pcover.p);
                hipLaunchKernelGGL(k_adv_check, dim3(sb), dim3(256), 0, stream, S, (const uint32_t *)d_pcover.p, full, d_pvalid.p, changed);
                return (int)STCSP_OK;
            });
            if (rc != STCSP_OK) return rc;
            if (E) hipLaunchKernelGGL(k_kill_into_invalid, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, d_palive.p, (const uint8_t *)d_pvalid.p, 0);
            HIPCHK(hipMemcpyAsync(&root_valid, d_pvalid.p, 1, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            adver1 = root_valid;
        }
        if (op >= 0) {  // adversarialTraverse2 (src/graph.cpp:247-302)
            const uint32_t full = full_mask(op);
            int rc = fixpoint(rounds[2], [&] {
                HIPCHK(hipMemsetAsync(d_pcover.p, 0, cover_words * sizeof(uint32_t), stream));
                if (E)
                    hipLaunchKernelGGL(k_adv2_cover, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, val, N, op, ava, mgr.lb[op], mgr.lb[ava],
                                       wa, (const uint8_t *)d_palive.p, (const uint8_t *)d_pvalid.p, d_pcover.p);
                hipLaunchKernelGGL(k_adv2_check, dim3(sb), dim3(256), 0, stream, S, (const uint32_t *)d_pcover.p, wa, full, d_pvalid.p, d_pnodeok.p, changed);
                if (E)
                    hipLaunchKernelGGL(k_adv2_kill, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, val, N, ava, mgr.lb[ava], wa, full, d_palive.p,
                                       (const uint8_t *)d_pvalid.p, (const uint8_t *)d_pnodeok.p, (const uint32_t *)d_pcover.p);
                return (int)STCSP_OK;
            });
            if (rc != STCSP_OK) return rc;
            HIPCHK(hipMemcpyAsync(&root_valid, d_pvalid.p, 1, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
            adver2 = root_valid;
            // the reference drops the edges into invalid states only when the root survived (graph.cpp:288-301)
            if (root_valid && E)
                hipLaunchKernelGGL(k_kill_into_invalid, dim3(eb), dim3(256), 0, stream, (uint32_t)E, src, dst, d_palive.p, (const uint8_t *)d_pvalid.p, 0);
        }
        HIPCHK(hipGetLastError());
        p_valid.resize(S);
        p_final.resize(S);
        p_alive.resize(E + 1);
        HIPCHK(hipMemcpyAsync(p_valid.data(), d_pvalid.p, S, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipMemcpyAsync(p_final.data(), d_pfinal.p, S, hipMemcpyDeviceToHost, stream));
        if (E) HIPCHK(hipMemcpyAsync(p_alive.data(), d_palive.p, E, hipMemcpyDeviceToHost, stream));
        HIPCHK(hipStreamSynchronize(stream));
        memset(out, 0, sizeof *out);
        out->n_states = S;
        out->n_edges = (int64_t)E;
        out->state_valid = p_valid.data();
        out->state_final = p_final.data();
        out->edge_alive = p_alive.data();
        out->adver1 = adver1;
        out->adver2 = adver2;
        for (int i = 0; i < 3; i++) out->rounds[i] = rounds[i];
        out->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return STCSP_OK;
    }

    // unsharded export: ok-fixpoint + compaction on the device, result arrays land in pinned memory
    int export_device(stcsp_result *res, stcsp_counters &ctr, size_t &E_out) {
        const int N = ctx.N;
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
        if (streaming) {  // the rest of the log (the chunks before it left while the search ran)
            if (getenv("STCSP_DEBUG")) {
                const bool busy = hipStreamQuery(xstream) == hipErrorNotReady || hipStreamQuery(xstream2) == hipErrorNotReady;
                fprintf(stderr, "[export] %zu of %zu edge records handed to the copy stream before the search ended (stream %s)\n", streamed, E, busy ? "still busy" : "idle");
            }
            int rcs = stream_edges(edge_count.data(), true);
            if (rcs != STCSP_OK) return rcs;
        }
        {
            int rcs = ensure_export_capacity(E);
            if (rcs != STCSP_OK) return rcs;
        }
        if (d_fail.n < n_states) {
            HIPCHK(d_fail.alloc((size_t)n_states + n_states / 4 + 256));
            HIPCHK(d_outdeg.alloc((size_t)n_states + n_states / 4 + 256));
        }
        if (!d_post.p) HIPCHK(d_post.alloc(4));
        if (h_state_cap < n_states) {
            if (h_fail) (void)hipHostFree(h_fail);
            h_state_cap = (size_t)n_states + n_states / 4 + 256;
            HIPCHK(hipHostMalloc((void **)&h_fail, h_state_cap));
        }
        {  // the keys of the states the search did not get to ship (all of them when streaming is off)
            int rcs = stream_states(n_states);
            if (rcs != STCSP_OK) return rcs;
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
        const bool quick_try = E && streaming && streamed == E && d_sdeg.n >= n_states && h_begin;
        if (!quick_try) {  // (the quick path's kernel writes every state's flag: nothing to clear in front of it)
            HIPCHK(hipMemsetAsync(d_fail.p, 0, n_states, stream));
            HIPCHK(hipMemsetAsync(d_post.p, 0, 4 * sizeof(uint32_t), stream));
        }
        const unsigned eb = (unsigned)((E + 255) / 256), sb = (n_states + 255) / 256;
        uint32_t live = 0;
        bool quick_final = false;
        if (quick_try) {
            // the whole log went through the transposition kernels, which counted the out-degrees on their way: the first
            // round of the ok-fixpoint is one pass over the states. Nobody without an out-edge (every shipped example):
            // every logged edge is kept and the streamed arrays are the result.
            // ONE synchronisation for everything that is still in flight: the main stream waits for the export streams' last
            // chunks (transposition kernels AND their copies) and the key arrays, marks, copies the verdict and the flags
            for (int i = 0; i < 2; i++)
                if (ev_x_used[i]) HIPCHK(hipStreamWaitEvent(stream, ev_c[i], 0));
            if (ev_k_used) HIPCHK(hipStreamWaitEvent(stream, ev_k, 0));
            volatile uint32_t *changed_p = h_begin;  // (pinned scratch of begin(): idle by now, at least the plan header long)
            *changed_p = 0u;
            hipLaunchKernelGGL(k_post_mark_host, dim3(sb), dim3(256), 0, stream, n_states, (const uint32_t *)d_sdeg.p, d_fail.p, h_fail, (uint32_t *)h_begin);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            const uint32_t changed = *changed_p;
            quick_final = !changed;
            if (!quick_final) {  // somebody failed: the full fixpoint below starts from scratch
                HIPCHK(hipMemsetAsync(d_fail.p, 0, n_states, stream));
                HIPCHK(hipMemsetAsync(d_post.p, 0, 4 * sizeof(uint32_t), stream));
            }
            lap("first fixpoint round");
        }
        if (quick_final) {
            // (no synchronisation of the export streams: the main stream waited for the events behind their last operations --
            // every chunk's copies, the key kernel -- and has just been synchronised; two more calls cost ~30 us here)
            live = (uint32_t)E;
            lap("wait for the streamed chunks");
        } else if (E) {
            HIPCHK(hipMemsetAsync(d_outdeg.p, 0, (size_t)n_states * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_post_outdeg, dim3(eb), dim3(256), 0, stream, v, d_outdeg.p, d_alive.p);
            bool streamed_is_final = false;
            for (int it = 0;; it++) {
                hipLaunchKernelGGL(k_post_mark, dim3(sb), dim3(256), 0, stream, n_states, (const uint32_t *)d_outdeg.p, d_fail.p, d_post.p);
                uint32_t changed = 0;
                HIPCHK(hipMemcpyAsync(&changed, d_post.p, sizeof changed, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                // nobody failed: every logged edge is kept, and the streamed arrays already hold them all
                if (!changed && it == 0 && streaming && streamed == E) streamed_is_final = true;
                if (!changed) break;
                HIPCHK(hipMemsetAsync(d_post.p, 0, sizeof(uint32_t), stream));
                hipLaunchKernelGGL(k_post_kill, dim3(eb), dim3(256), 0, stream, v, d_alive.p, (const uint8_t *)d_fail.p, d_outdeg.p);
                if (it > (int)n_states + 8) return fail(STCSP_E_INTERNAL, "ok-fixpoint did not converge");
            }
            lap("fixpoint");
            { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }  // the last streamed chunk has landed (or nothing was streamed)
            if (streamed_is_final) {
                live = (uint32_t)E;
                lap("wait for the streamed chunks");
            } else {
                hipLaunchKernelGGL(k_post_compact, dim3(eb), dim3(256), 0, stream, v, (const uint8_t *)d_alive.p, d_post.p + 1, d_osrc.p, d_odst.p, d_oval.p);
                HIPCHK(hipGetLastError());
                lap("compact");
                HIPCHK(hipMemcpyAsync(&live, d_post.p + 1, sizeof live, hipMemcpyDeviceToHost, stream));
                HIPCHK(hipStreamSynchronize(stream));
                HIPCHK(hipMemcpyAsync(h_osrc, d_osrc.p, (size_t)live * sizeof(long long), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipMemcpyAsync(h_odst, d_odst.p, (size_t)live * sizeof(long long), hipMemcpyDeviceToHost, stream));
                HIPCHK(hipMemcpyAsync(h_oval, d_oval.p, (size_t)live * N * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
                streaming = false;  // (the arrays no longer hold the streamed layout: a second export() compacts again)
            }
        } else {
            // no edges at all: every non-root state is failed
            HIPCHK(hipMemsetAsync(d_outdeg.p, 0, (size_t)n_states * sizeof(uint32_t), stream));
            hipLaunchKernelGGL(k_post_mark, dim3(sb), dim3(256), 0, stream, n_states, (const uint32_t *)d_outdeg.p, d_fail.p, d_post.p);
        }
        if (!quick_final) {  // (the quick path brought the flags with its one synchronisation)
            HIPCHK(hipMemcpyAsync(h_fail, d_fail.p, n_states, hipMemcpyDeviceToHost, stream));
            HIPCHK(hipStreamSynchronize(stream));
        }
        lap("D2H result arrays");
        { int rcx = sync_xstreams(); if (rcx != STCSP_OK) return rcx; }  // cid / sig arrays complete
        int64_t ok_states = 0;
        for (uint32_t i = 1; i < n_states; i++) ok_states += !h_fail[i];
        ctr.dominance = (int64_t)live - ok_states;  // every ok non-root state is entered by exactly one creating leaf
        lap("state arrays");
        res->state_cid = h_cid;
        res->state_sig = h_sig;
        res->state_fail = h_fail;
        res->edge_src = (const int64_t *)h_osrc;
        res->edge_dst = (const int64_t *)h_odst;
        res->edge_values = h_oval;
        E_out = live;
        exp_edges = live;
        exp_on_device = true;
        return STCSP_OK;
    }

    // Solver::seenConstraints = the initial set + every set a leaf translated to = the distinct set ids of the
    // table's states (the registry may hold more: sets translated ahead of need)
    int32_t used_sets() const {
        const int32_t *cid = exp_on_device ? h_cid : r_cid.data();
        std::vector<uint8_t> seen(mgr.sets.size() + 1, 0);
        std::vector<int32_t> other;  // (sharded runs: ids are content hashes)
        seen[0] = 1;
        int32_t n = 1;
        for (uint32_t i = 0; i < n_states; i++) {
            const int32_t c = cid[i];
            if (c >= 0 && (size_t)c < seen.size()) {
                n += !seen[c];
                seen[c] = 1;
            } else {
                other.push_back(c);
            }
        }
        std::sort(other.begin(), other.end());
        return n + (int32_t)(std::unique(other.begin(), other.end()) - other.begin());
    }
    int export_result(stcsp_result *res) {
        auto t0 = std::chrono::steady_clock::now();
        HIPCHK(hipSetDevice(device));
        const int KL = ctx.KL, sl = ctx.sig_len, N = ctx.N, ES = ctx.ES;
        if (!sharded && !(opt.flags & STCSP_F_KEEP_RAW_EDGES) && !getenv("STCSP_HOST_EXPORT")) {
            stcsp_counters ctr = snap;  // (finish() read them)
            int rcc = finished ? STCSP_OK : read_counters(ctr);
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
            res->n_edges = (int64_t)E;
            res->n_vars = N;
            res->n_constraint_sets = used_sets();
            res->var_is_signature = r_issig.data();
            res->root_final = mgr.n_until_cons == 0;
            res->truncated = truncated;
            seconds_export = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (getenv("STCSP_DEBUG")) fprintf(stderr, "[export] whole export_result    %.3f ms\n", seconds_export * 1e3);
            ctr.seconds_export = seconds_export;
            res->counters = ctr;
            return STCSP_OK;
        }
        if (sharded && streaming) {
            // a shard's export = its states and its RAW leaf-edge log (the ok-fixpoint needs every shard's edges: it runs
            // after the merge): exactly what the streaming export has been shipping during the supersteps
            stcsp_counters ctr{};
            int rcc = read_counters(ctr);
            if (rcc != STCSP_OK) return rcc;
            size_t E = 0;
            for (int r = 0; r < R; r++) E += edge_count[r];
            if (E > 0xfffffff0ull) return fail(STCSP_E_NOMEM, "edge log too large for the device export");
            if ((rcc = stream_edges(edge_count.data(), true)) || (rcc = ensure_export_capacity(E)) || (rcc = stream_states(n_states)) || (rcc = sync_xstreams()))
                return rcc;
            if (streamed != E) return fail(STCSP_E_INTERNAL, "streamed %zu of %zu edge records", streamed, E);
            r_fail.assign(n_states, 0);
            r_issig.assign(mgr.is_sig.begin(), mgr.is_sig.end());
            memset(res, 0, sizeof *res);
            res->n_states = n_states;
            res->sig_len = sl;
            res->n_sig_vars = mgr.n_sig;
            res->n_until = mgr.n_until;
            res->n_until_cons = mgr.n_until_cons;
            res->state_cid = h_cid;
            res->state_sig = h_sig;
            res->state_fail = r_fail.data();
            res->n_edges = (int64_t)E;
            res->edge_src = (const int64_t *)h_osrc;
            res->edge_dst = (const int64_t *)h_odst;
            res->edge_values = h_oval;
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
        if (!sharded) {
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
        res->n_constraint_sets = sharded ? (int32_t)mgr.sets.size() : used_sets();
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
    if (e->sharded) return e->fail(STCSP_E_STATE, "solve() is the unsharded entry point; use the stepping calls when world > 1 or STCSP_F_STEPPED");
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

int stcsp_engine_postprocess(stcsp_engine *e, const stcsp_post_options *options, stcsp_post_result *out) {
    if (!e || !out) return STCSP_E_INVALID;
    return e->postprocess(options, out);
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
int stcsp_engine_set_expand_budget(stcsp_engine *e, int64_t max_rounds, int64_t min_open) {
    if (!e || max_rounds < 0 || min_open < 0) return STCSP_E_INVALID;
    e->step_max_rounds = max_rounds;
    e->step_min_open = min_open;
    return STCSP_OK;
}
int stcsp_engine_node_bytes(const stcsp_engine *e) { return e ? xfer_stride(e->ctx.N, e->ctx.K * e->ctx.W) * 4 : STCSP_E_INVALID; }
int stcsp_engine_donate(stcsp_engine *e, int64_t want, void **ptr, int64_t *count) {
    if (!e || !ptr || !count) return STCSP_E_INVALID;
    return e->donate(want, ptr, count);
}
int stcsp_engine_adopt(stcsp_engine *e, const void *records, int64_t count) { return e ? e->adopt(records, count) : STCSP_E_INVALID; }
int stcsp_engine_counters(stcsp_engine *e, stcsp_counters *out) {
    if (!e || !out) return STCSP_E_INVALID;
    if (!e->begun) return e->fail(STCSP_E_STATE, "counters before a solve");
    return e->read_counters(*out);
}
int stcsp_engine_propagate(stcsp_engine *e, int32_t set, uint32_t expire, uint32_t *blocks, int64_t count, int32_t *outcome, int64_t *skipped) {
    return e ? e->propagate(set, expire, blocks, count, outcome, skipped) : STCSP_E_INVALID;
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
    if (e->mgr.sets.size() != before) {
        const int rcu = e->upload_program();
        return rcu != STCSP_OK ? rcu : e->fresh_init();
    }
    return STCSP_OK;
}

}  // extern "C"

#define HIPCHK_E(e, call)                                                                                         \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return (e)->fail(STCSP_E_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#include "sharded_native.hpp"
