// sharded_native.hpp -- stcsp_engine_solve_sharded(): the superstep loop of a sharded search on the host side of
// ONE rank, over the engine's own stepping calls and a caller-provided transport (include/stcsp_sharded.h); and the
// in-process transport (ranks = host threads, records by hipMemcpyPeerAsync). Included by engine.hip only.
//
// The loop is the one stcsp-solver_amd/sharded.py runs over torch.distributed (kept as the gloo test driver); here
// nothing but C++ and the transport's three collectives sit between two k_expand bursts:
//
//   expand_local            this shard's open nodes, until its frontier is dry or its budget of launch rounds is spent
//   all_gather_i64          per rank: candidates per peer, open nodes, number of constraint sets, status
//   (all_gather_bytes)      constraint-set definitions, only when some rank's count moved
//   donate / all_to_all_v / adopt     open nodes from the shards above the mean to the shards below it (the branch
//                           case of solverSolveRe, reference src/solveralgorithm.cpp:911-939, is the work that moves)
//   all_to_all_v / commit   leaf successor candidates to the owner of the successor state (hash(set, signature) % world)
//
// until no rank has open nodes or candidates. Errors: a rank whose engine failed reports it in the status column of
// the next count table (or in the one-word agreement behind donate / set import / finish); every rank then returns
// an error from the same superstep.
#pragma once
#include <condition_variable>
#include <mutex>

#include "stcsp_sharded.h"

namespace {

// Deterministic redistribution plan from the all-gathered open-node counts (every rank computes the same one): shards
// above the mean give their surplus to the shards below it, largest surplus to largest deficit first. send[i*world+j] =
// nodes rank i ships to rank j; all zero when the poorest shard has at least half the mean.
inline std::vector<int64_t> plan_transfers(const std::vector<int64_t> &left) {
    const int world = (int)left.size();
    std::vector<int64_t> send((size_t)world * world, 0);
    int64_t total = 0, least = left.empty() ? 0 : left[0];
    for (int64_t v : left) {
        total += v;
        least = std::min(least, v);
    }
    if (world == 1 || total == 0) return send;
    const int64_t mean = total / world;
    if (mean == 0 || least * 2 >= mean) return send;
    std::vector<std::pair<int64_t, int>> surplus, deficit;
    for (int r = 0; r < world; r++) {
        if (left[r] > mean) surplus.push_back({left[r] - mean, r});
        if (left[r] < mean) deficit.push_back({mean - left[r], r});
    }
    auto desc = [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a > b; };
    std::sort(surplus.begin(), surplus.end(), desc);
    std::sort(deficit.begin(), deficit.end(), desc);
    size_t si = 0, di = 0;
    while (si < surplus.size() && di < deficit.size()) {
        const int64_t n = std::min(surplus[si].first, deficit[di].first);
        if (n >= 1) send[(size_t)surplus[si].second * world + deficit[di].second] += n;
        surplus[si].first -= n;
        deficit[di].first -= n;
        if (surplus[si].first == 0) si++;
        if (deficit[di].first == 0) di++;
    }
    return send;
}

}  // namespace

// (a member of stcsp_engine in spirit: it only uses the stepping calls the C-ABI exports)
inline int stcsp_solve_sharded_impl(stcsp_engine *e, const stcsp_transport *t, const stcsp_sharded_options *o, stcsp_sharded_stats *st) {
    using clock = std::chrono::steady_clock;
    if (!t || !t->all_gather_i64 || !t->all_to_all_v || !t->all_gather_bytes) return e->fail(STCSP_E_INVALID, "solve_sharded: incomplete transport");
    if (!e->sharded) return e->fail(STCSP_E_STATE, "solve_sharded needs an engine created with world > 1 or STCSP_F_STEPPED");
    const int world = t->world, rank = t->rank;
    if (world != e->opt.world || rank != e->opt.rank)
        return e->fail(STCSP_E_INVALID, "solve_sharded: transport is rank %d of %d, the engine rank %d of %d", rank, world, e->opt.rank, e->opt.world);
    const int64_t budget_rounds = o && o->budget_rounds > 0 ? o->budget_rounds : 8;
    const int64_t share_per_rank = o && o->share_per_rank > 0 ? o->share_per_rank : 64;
    const int64_t max_steps = o && o->max_supersteps > 0 ? o->max_supersteps : 1000000;
    const int64_t csw = e->ctx.CS, nsw = xfer_stride(e->ctx.N, e->ctx.K * e->ctx.W);
    double t_coll = 0;
    stcsp_sharded_stats s{};
    int pending = STCSP_OK;  // this rank's engine error the peers have not heard of yet (e->err holds its text)
    auto guard = [&](int rc) {
        if (pending == STCSP_OK && rc != STCSP_OK) pending = rc;
        return rc;
    };
    auto transport_failed = [&](const char *what) {
        return e->fail(STCSP_E_DEVICE, "solve_sharded: transport failed in %s: %s", what, t->last_error ? t->last_error(t->self) : "?");
    };
    // one-word agreement: every rank leaves with the same verdict
    auto agree = [&](const char *what) -> int {
        const int64_t mine = pending != STCSP_OK ? 1 : 0;
        std::vector<int64_t> all((size_t)world);
        const auto t0 = clock::now();
        if (t->all_gather_i64(t->self, &mine, 1, all.data()) != 0) return transport_failed(what);
        t_coll += std::chrono::duration<double>(clock::now() - t0).count();
        for (int r = 0; r < world; r++)
            if (all[r]) {
                if (pending != STCSP_OK) return pending;  // (e->err already says why)
                return e->fail(STCSP_E_INTERNAL, "solve_sharded: rank %d failed in %s", r, what);
            }
        return STCSP_OK;
    };
    e->step_max_rounds = world > 1 ? budget_rounds : 0;
    e->step_min_open = share_per_rank * world;
    guard(e->begin());
    int64_t last_sets = (int64_t)e->mgr.sets.size();
    bool sets_uniform = true;  // every shard starts from the same registry (the model's own sets + what earlier solves exchanged)
    const int W = world + 4;   // row of the count table: candidates per peer, open nodes, sets, blob words, status
    std::vector<int64_t> row((size_t)W), table((size_t)world * W), send_w((size_t)world), recv_w((size_t)world);
    for (int64_t step = 1;; step++) {
        s.supersteps = step;
        if (step > max_steps && pending == STCSP_OK) pending = e->fail(STCSP_E_INTERNAL, "sharded solve did not terminate in %lld supersteps", (long long)max_steps);
        int64_t left = 0;
        void *out_ptr0 = nullptr;
        std::fill(row.begin(), row.end(), 0);
        auto t_ph = clock::now();
        const bool expanded = pending == STCSP_OK && guard(e->expand_local(&left)) == STCSP_OK;
        s.seconds_expand += std::chrono::duration<double>(clock::now() - t_ph).count();
        t_ph = clock::now();
        if (expanded) {
            for (int p = 0; p < world && pending == STCSP_OK; p++) {
                void *ptr = nullptr;
                int64_t cnt = 0;
                if (guard(e->outbox(p, &ptr, &cnt, /*sync=*/false)) == STCSP_OK) {  // (all_to_all_v is ordered on the engine's stream)
                    row[p] = cnt;
                    if (p == 0) out_ptr0 = ptr;  // (the engine packs the peers back to back: pack_outboxes)
                }
            }
        }
        const int32_t *blob = nullptr;
        int64_t blob_words = 0;
        if (pending == STCSP_OK) guard(stcsp_engine_sets_blob(e, &blob, &blob_words));
        s.seconds_pack += std::chrono::duration<double>(clock::now() - t_ph).count();
        if (pending != STCSP_OK) {
            std::fill(row.begin(), row.end(), 0);
            row[world + 3] = 1;
        } else {
            row[world] = left;
            row[world + 1] = (int64_t)e->mgr.sets.size();
            row[world + 2] = blob_words;
        }
        auto t0 = clock::now();
        if (t->all_gather_i64(t->self, row.data(), W, table.data()) != 0) return transport_failed("the count table");
        t_coll += std::chrono::duration<double>(clock::now() - t0).count();
        for (int r = 0; r < world; r++)
            if (table[(size_t)r * W + world + 3]) {
                if (pending != STCSP_OK) return pending;
                return e->fail(STCSP_E_INTERNAL, "solve_sharded: rank %d failed in superstep %lld", r, (long long)step);
            }
        // ---- constraint sets: somebody met a new one -> everyone learns all of them
        bool sets_moved = false;
        for (int r = 0; r < world; r++) sets_moved = sets_moved || table[(size_t)r * W + world + 1] != last_sets;
        if (sets_moved || !sets_uniform) {
            int64_t width = 1;
            for (int r = 0; r < world; r++) width = std::max(width, table[(size_t)r * W + world + 2]);
            std::vector<int32_t> mine((size_t)width, 0), all((size_t)width * world);
            memcpy(mine.data(), blob, (size_t)blob_words * 4);
            t0 = clock::now();
            if (t->all_gather_bytes(t->self, mine.data(), width * 4, all.data()) != 0) return transport_failed("the constraint-set exchange");
            t_coll += std::chrono::duration<double>(clock::now() - t0).count();
            for (int r = 0; r < world && pending == STCSP_OK; r++)
                if (r != rank) guard(stcsp_engine_sets_import(e, all.data() + (size_t)r * width, table[(size_t)r * W + world + 2]));
            const int rc = agree("the constraint-set exchange");
            if (rc != STCSP_OK) return rc;
            last_sets = (int64_t)e->mgr.sets.size();  // after the import every shard knows the union
            sets_uniform = true;
        }
        // ---- frontier redistribution
        std::vector<int64_t> lefts((size_t)world);
        int64_t total_open = 0, total_cands = 0;
        for (int r = 0; r < world; r++) {
            lefts[r] = table[(size_t)r * W + world];
            total_open += lefts[r];
            for (int p = 0; p < world; p++) total_cands += table[(size_t)r * W + p];
        }
        const std::vector<int64_t> plan = plan_transfers(lefts);
        bool any_transfer = false;
        for (int64_t v : plan) any_transfer = any_transfer || v != 0;
        int64_t n_adopt = 0;
        if (any_transfer) {
            int64_t want = 0;
            for (int p = 0; p < world; p++) {
                send_w[p] = plan[(size_t)rank * world + p] * nsw;
                recv_w[p] = plan[(size_t)p * world + rank] * nsw;
                want += plan[(size_t)rank * world + p];
                n_adopt += plan[(size_t)p * world + rank];
            }
            void *xptr = nullptr;
            int64_t got = 0;
            if (want && guard(e->donate(want, &xptr, &got)) == STCSP_OK && got != want)
                pending = e->fail(STCSP_E_INTERNAL, "planned to donate %lld open nodes, the engine gave %lld", (long long)want, (long long)got);
            const int rc = agree("donate");  // before the exchange: nobody waits for a rank that failed
            if (rc != STCSP_OK) return rc;
            if (e->d_recv_nodes.n < (size_t)std::max<int64_t>(n_adopt * nsw, 1)) HIPCHK_E(e, e->d_recv_nodes.alloc((size_t)(n_adopt * nsw) * 2 + 1024));
            t0 = clock::now();
            if (t->all_to_all_v(t->self, xptr, send_w.data(), e->d_recv_nodes.p, recv_w.data(), (void *)e->stream) != 0) return transport_failed("the node exchange");
            t_coll += std::chrono::duration<double>(clock::now() - t0).count();
            s.nodes_donated += got;
            s.nodes_adopted += n_adopt;
        }
        // ---- leaf successor candidates
        int64_t n_recv = 0, n_send = 0;
        for (int p = 0; p < world; p++) {
            send_w[p] = row[p] * csw;
            recv_w[p] = table[(size_t)p * W + rank] * csw;
            n_send += row[p];
            n_recv += table[(size_t)p * W + rank];
        }
        if (total_cands) {
            if (e->d_recv_cand.n < (size_t)std::max<int64_t>(n_recv * csw, 1)) HIPCHK_E(e, e->d_recv_cand.alloc((size_t)(n_recv * csw) * 2 + 1024));
            t0 = clock::now();
            if (t->all_to_all_v(t->self, out_ptr0, send_w.data(), e->d_recv_cand.p, recv_w.data(), (void *)e->stream) != 0) return transport_failed("the candidate exchange");
            t_coll += std::chrono::duration<double>(clock::now() - t0).count();
        }
        s.candidates_sent += n_send;
        s.candidates_received += n_recv;
        // (an error from here on travels with the next superstep's count table, or with the final agreement)
        t_ph = clock::now();
        guard(e->commit(e->d_recv_cand.p, n_recv));
        if (n_adopt && pending == STCSP_OK) guard(e->adopt(e->d_recv_nodes.p, n_adopt));
        s.seconds_commit += std::chrono::duration<double>(clock::now() - t_ph).count();
        if (total_open == 0 && total_cands == 0) break;
    }
    if (pending == STCSP_OK) guard(e->finish());
    const int rc = agree("the last superstep");
    s.seconds_collectives = t_coll;
    if (st) *st = s;
    return rc;
}

// ---------------------------------------------------------------- in-process transport: ranks = host threads
struct stcsp_local_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    // what the ranks publish for the collective in flight
    std::vector<const void *> ptr;
    std::vector<const int64_t *> counts;
    std::vector<int> device;
    std::vector<stcsp_transport> transports;
    std::vector<std::string> errors;
    struct Member {
        stcsp_local_group *g;
        int rank;
    };
    std::vector<Member> members;
    bool broken = false;  // a rank's copy failed: every later collective of the group fails at once instead of waiting for that rank
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long long gen = generation;
        if (broken) return;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || broken; });
        }
    }
    void give_up() {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

namespace {
int local_all_gather_bytes(void *self, const void *mine, int64_t n, void *all) {
    auto *m = (stcsp_local_group::Member *)self;
    stcsp_local_group *g = m->g;
    g->ptr[m->rank] = mine;
    g->barrier();
    if (g->broken) {
        g->errors[m->rank] = "a peer's transport call failed";
        return -1;
    }
    for (int r = 0; r < g->world; r++) memcpy((char *)all + (size_t)r * n, g->ptr[r], (size_t)n);
    g->barrier();  // (nobody's buffer is reused before everyone has copied)
    return g->broken ? -1 : 0;
}
int local_all_gather_i64(void *self, const int64_t *mine, int32_t n, int64_t *all) { return local_all_gather_bytes(self, mine, (int64_t)n * 8, all); }
int local_all_to_all_v(void *self, const void *send, const int64_t *send_words, void *recv, const int64_t *recv_words, void *stream) {
    auto *m = (stcsp_local_group::Member *)self;
    stcsp_local_group *g = m->g;
    int dev = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess;
    // what this rank sends has been produced by work on its stream: complete before the peers read it
    ok = ok && hipStreamSynchronize((hipStream_t)stream) == hipSuccess;
    g->ptr[m->rank] = send;
    g->counts[m->rank] = send_words;
    g->device[m->rank] = dev;
    g->barrier();
    if (g->broken) {
        g->errors[m->rank] = "a peer's transport call failed";
        return -1;
    }
    size_t roff = 0;
    for (int p = 0; p < g->world && ok; p++) {
        size_t soff = 0;  // where this rank's part starts in peer p's send buffer
        for (int q = 0; q < m->rank; q++) soff += (size_t)g->counts[p][q];
        const size_t words = (size_t)recv_words[p];
        if ((int64_t)words != g->counts[p][m->rank]) {
            g->errors[m->rank] = "all_to_all_v: send and receive counts disagree";
            ok = false;
            break;
        }
        if (words) {
            const uint32_t *src = (const uint32_t *)g->ptr[p] + soff;
            uint32_t *dst = (uint32_t *)recv + roff;
            const hipError_t err = g->device[p] == dev ? hipMemcpyAsync(dst, src, words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream)
                                                       : hipMemcpyPeerAsync(dst, dev, src, g->device[p], words * 4, (hipStream_t)stream);
            if (err != hipSuccess) {
                g->errors[m->rank] = std::string("all_to_all_v: ") + hipGetErrorString(err);
                ok = false;
            }
        }
        roff += words;
    }
    ok = ok && hipStreamSynchronize((hipStream_t)stream) == hipSuccess;
    if (!ok) g->give_up();  // (the peers must not wait for this rank in the collectives to come)
    g->barrier();  // (send buffers may be reused from here on)
    return ok && !g->broken ? 0 : -1;
}
const char *local_last_error(void *self) {
    auto *m = (stcsp_local_group::Member *)self;
    return m->g->errors[m->rank].c_str();
}
}  // namespace

extern "C" {
int stcsp_engine_solve_sharded(stcsp_engine *e, const stcsp_transport *t, const stcsp_sharded_options *o, stcsp_sharded_stats *st) {
    if (!e) return STCSP_E_INVALID;
    (void)hipSetDevice(e->device);  // (one host thread per engine: the thread's current device is the engine's)
    return stcsp_solve_sharded_impl(e, t, o, st);
}
int stcsp_local_group_create(int32_t world, stcsp_local_group **out) {
    if (world < 1 || !out) return STCSP_E_INVALID;
    auto *g = new stcsp_local_group();
    g->world = world;
    g->ptr.assign(world, nullptr);
    g->counts.assign(world, nullptr);
    g->device.assign(world, 0);
    g->errors.assign(world, "");
    g->members.resize(world);
    g->transports.resize(world);
    for (int r = 0; r < world; r++) {
        g->members[r] = {g, r};
        g->transports[r] = stcsp_transport{&g->members[r], r, world, local_all_gather_i64, local_all_gather_bytes, local_all_to_all_v, local_last_error};
    }
    *out = g;
    return STCSP_OK;
}
const stcsp_transport *stcsp_local_group_transport(stcsp_local_group *g, int32_t rank) {
    return (g && rank >= 0 && rank < g->world) ? &g->transports[rank] : nullptr;
}
void stcsp_local_group_destroy(stcsp_local_group *g) { delete g; }
}
