// stcsp_main.cpp -- the `stcsp` command line, functionally as the reference's
// (src/stcsp.y:180-219 main, src/solver.cpp:195-359 solve): same flags, same stdout contract,
// same solutions.dot. The search itself runs on the MI355X engine behind the C-ABI.
//
//   stcsp [-s] [-m<sec>] [-t] [-a] [-z] [-k<K>] [-l<level>] [--binary=<file>] [--shards=<N>] input.csp
//
// --binary=<file> (not in the reference) additionally writes the printed automaton in the compact
// binary form of include/stcsp_host.h.
// --shards=<N> (not in the reference, which is single-threaded) shards the open search frontier and the state table over N
// engines -- one per GPU of the node, round robin when there are fewer GPUs than shards -- driven by N host threads through
// stcsp_engine_solve_sharded() and the in-process transport (include/stcsp_sharded.h: records move between the GPUs with
// hipMemcpyPeerAsync); the shards' automata are merged and post-processed on the host.
//
// Options must be glued to their value (-k3, not -k 3): like the reference, the first argument
// that does not start with '-' is the input file (stcsp.y:199-206).
#include <sys/times.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "stcsp_engine.h"
#include "stcsp_host.h"
#include "stcsp_sharded.h"

static double cpu_time() {  // cpuTime (util.cpp:149-155)
    struct tms b;
    times(&b);
    return (double)(b.tms_utime + b.tms_stime + b.tms_cutime + b.tms_cstime) / (double)sysconf(_SC_CLK_TCK);
}

struct Flags {
    bool print_solution = false, testing = false, adv1 = false, adv2 = false;
    int prefix_k = 2, time_limit = 0, shards = 1;
    const char *file = nullptr;
    const char *binary = nullptr;
};

static int run_once(const Flags &f, bool print_line, double *total) {
    double t_init = cpu_time();
    stcsp_model *model = nullptr;
    int rc = f.file ? stcsp_model_load_file(f.file, f.prefix_k, &model) : STCSP_E_INVALID;
    if (rc != STCSP_OK) {
        // syntax errors go to stdout like yyerror (stcsp.y:221-224); the rest to error.txt in the
        // reference (myLog) -- stderr here
        const char *msg = stcsp_host_last_error();
        if (strncmp(msg, "Line ", 5) == 0)
            printf("%s\n", msg);
        else
            fprintf(stderr, "%s\n", msg);
        return 1;
    }
    const stcsp_problem *p = stcsp_model_problem(model);
    double init_time = cpu_time() - t_init;
    stcsp_options opt;
    memset(&opt, 0, sizeof opt);
    opt.world = 1;
    opt.time_limit_s = f.time_limit;  // -m: the reference exit(0)s silently on SIGALRM (solver.cpp:190-193)
    stcsp_engine *eng = nullptr;
    rc = stcsp_engine_create(p, &opt, &eng);
    if (rc != STCSP_OK) {
        fprintf(stderr, "%s\n", stcsp_engine_last_error(nullptr));
        return 1;
    }
    double t_solve = cpu_time();
    stcsp_result res;
    rc = stcsp_engine_solve(eng, &res);
    if (rc != STCSP_OK) {
        fprintf(stderr, "%s\n", stcsp_engine_last_error(eng));
        return 1;
    }
    if (res.truncated) exit(0);  // time limit: silent exit 0, like the reference
    double solve_time = cpu_time() - t_solve;
    double t_proc = cpu_time();
    stcsp_automaton *a = nullptr;
    stcsp_automaton_build(p, &res, &a);
    // graphTraverse / adversarialTraverse / adversarialTraverse2 (solveralgorithm.cpp:974-983) run on
    // the device over the automaton the export left in HBM; the host only adopts the flags
    stcsp_post_options po = {f.adv1 ? 5 : -1, f.adv2 ? 5 : -1, f.adv2 ? 6 : -1, 0};
    stcsp_post_result post;
    rc = stcsp_engine_postprocess(eng, &po, &post);
    if (rc != STCSP_OK) {
        fprintf(stderr, "%s\n", stcsp_engine_last_error(eng));
        return 1;
    }
    stcsp_automaton_import_flags(a, post.state_valid, post.state_final, post.edge_alive);
    if (f.adv1) printf("adver1: %d; ", post.adver1);
    if (f.adv2) printf("adver2: %d\n", post.adver2);
    if (f.print_solution || f.binary) stcsp_automaton_order_by_label(a);  // reproducible files whatever the GPU's scheduling
    stcsp_automaton_renumber(a);
    double proc_time = cpu_time() - t_proc;
    if (f.print_solution) stcsp_automaton_write_dot(a, "solutions.dot");
    if (f.binary && stcsp_automaton_write_binary(a, f.binary) != STCSP_OK) fprintf(stderr, "cannot write %s\n", f.binary);
    if (print_line) {
        // init_time, var, con, dom, node, fail, solve_time, processTime (solveralgorithm.cpp:1001)
        printf("%.2f\t%d\t%d\t%d\t%d\t%d\t%.2f\t%.5f\n", init_time, p->n_vars, p->n_constraints, (int)res.counters.dominance,
               (int)res.n_states, (int)res.counters.fails, solve_time, proc_time);
        fflush(stdout);
    }
    if (total) *total = solve_time + proc_time;
    stcsp_automaton_free(a);
    stcsp_engine_destroy(eng);
    stcsp_model_free(model);
    return 0;
}

// --shards=N: the same run with the frontier and the state table sharded over N engines (see the header comment)
static int run_sharded(const Flags &f, bool print_line, double *total) {
    double t_init = cpu_time();
    stcsp_model *model = nullptr;
    if (stcsp_model_load_file(f.file, f.prefix_k, &model) != STCSP_OK) {
        const char *msg = stcsp_host_last_error();
        if (strncmp(msg, "Line ", 5) == 0)
            printf("%s\n", msg);
        else
            fprintf(stderr, "%s\n", msg);
        return 1;
    }
    const stcsp_problem *p = stcsp_model_problem(model);
    double init_time = cpu_time() - t_init;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        fprintf(stderr, "no HIP device available\n");
        return 1;
    }
    const int world = f.shards;
    std::vector<stcsp_engine *> eng((size_t)world, nullptr);
    for (int r = 0; r < world; r++) {
        stcsp_options opt;
        memset(&opt, 0, sizeof opt);
        opt.device = r % ndev;
        opt.rank = r;
        opt.world = world;
        opt.time_limit_s = f.time_limit;
        if (stcsp_engine_create(p, &opt, &eng[(size_t)r]) != STCSP_OK) {
            fprintf(stderr, "%s\n", stcsp_engine_last_error(nullptr));
            return 1;
        }
    }
    stcsp_local_group *group = nullptr;
    if (stcsp_local_group_create(world, &group) != STCSP_OK) return 1;
    double t_solve = cpu_time();
    std::vector<int> rcs((size_t)world, 0);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < world; r++)
            th.emplace_back([&, r] { rcs[(size_t)r] = stcsp_engine_solve_sharded(eng[(size_t)r], stcsp_local_group_transport(group, r), nullptr, nullptr); });
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < world; r++)
        if (rcs[(size_t)r] != STCSP_OK) {
            fprintf(stderr, "shard %d: %s\n", r, stcsp_engine_last_error(eng[(size_t)r]));
            return 1;
        }
    std::vector<stcsp_result> res((size_t)world);
    std::vector<const stcsp_result *> resp;
    for (int r = 0; r < world; r++) {
        if (stcsp_engine_export(eng[(size_t)r], &res[(size_t)r]) != STCSP_OK) {
            fprintf(stderr, "shard %d: %s\n", r, stcsp_engine_last_error(eng[(size_t)r]));
            return 1;
        }
        if (res[(size_t)r].truncated) exit(0);  // time limit: silent exit 0, like the reference
        resp.push_back(&res[(size_t)r]);
    }
    stcsp_merged *mg = nullptr;
    if (stcsp_merge_shards(resp.data(), world, &mg) != STCSP_OK) {
        fprintf(stderr, "merging the shards failed\n");
        return 1;
    }
    const stcsp_result *merged = stcsp_merged_result(mg);
    double solve_time = cpu_time() - t_solve;
    double t_proc = cpu_time();
    stcsp_automaton *a = nullptr;
    stcsp_automaton_build(p, merged, &a);
    stcsp_automaton_traverse(a);  // (host passes: the merged automaton lives on the host)
    if (f.adv1) printf("adver1: %d; ", stcsp_automaton_adversarial(a, 5));
    if (f.adv2) printf("adver2: %d\n", stcsp_automaton_adversarial2(a, 5, 6));
    if (f.print_solution || f.binary) stcsp_automaton_order_by_label(a);
    stcsp_automaton_renumber(a);
    double proc_time = cpu_time() - t_proc;
    if (f.print_solution) stcsp_automaton_write_dot(a, "solutions.dot");
    if (f.binary && stcsp_automaton_write_binary(a, f.binary) != STCSP_OK) fprintf(stderr, "cannot write %s\n", f.binary);
    if (print_line) {
        printf("%.2f\t%d\t%d\t%d\t%d\t%d\t%.2f\t%.5f\n", init_time, p->n_vars, p->n_constraints, (int)merged->counters.dominance,
               (int)merged->n_states, (int)merged->counters.fails, solve_time, proc_time);
        fflush(stdout);
    }
    if (total) *total = solve_time + proc_time;
    stcsp_automaton_free(a);
    stcsp_merged_free(mg);
    stcsp_local_group_destroy(group);
    for (stcsp_engine *e : eng) stcsp_engine_destroy(e);
    stcsp_model_free(model);
    return 0;
}

int main(int argc, char **argv) {
    Flags f;
    // Option letters of the reference (getopt string "b:e:cv:l:stk:m:az", src/solver.cpp:211). Short flags
    // may be grouped (-sa); an option value may be attached (-k3, the only form the reference's own
    // main() handles, src/stcsp.y:199-206) or be the next argument (-k 3).
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] != '-' || a[1] == 0) {
            if (!f.file) f.file = a;
            continue;
        }
        if (strncmp(a, "--binary=", 9) == 0) {
            f.binary = a + 9;
            continue;
        }
        if (strncmp(a, "--shards=", 9) == 0) {
            f.shards = atoi(a + 9);
            if (f.shards < 1 || f.shards > 64) {
                fprintf(stderr, "Invalid argument: %s\n", a);
                return 1;
            }
            continue;
        }
        for (const char *q = a + 1; *q; q++) {
            const char o = *q;
            if (strchr("bevlkm", o)) {  // takes a value: the rest of this argument, else the next one
                const char *val = q[1] ? q + 1 : (i + 1 < argc ? argv[++i] : nullptr);
                if (!val) {
                    fprintf(stderr, "Option -%c needs a value\n", o);
                    return 1;
                }
                if (o == 'k' || o == 'm') {
                    char *end = nullptr;
                    const long n = strtol(val, &end, 10);
                    if (end == val || *end || n < 0 || n > 1000000 || (o == 'k' && n <= 0)) {
                        fprintf(stderr, "Invalid argument: -%c %s\n", o, val);
                        return 1;
                    }
                    (o == 'k' ? f.prefix_k : f.time_limit) = (int)n;
                }
                break;  // b, e, v, l: parsed but unused in the reference too
            }
            switch (o) {
                case 's': f.print_solution = true; break;
                case 't': f.testing = true; break;
                case 'a': f.adv1 = true; break;
                case 'z': f.adv2 = true; break;
                default: fprintf(stderr, "Unknown argument: %c\n", o); return 1;
            }
        }
    }
    if (!f.file) {
        printf("No constraints!\n");
        return 0;
    }
    auto run = [&](bool print_line, double *total) { return f.shards > 1 ? run_sharded(f, print_line, total) : run_once(f, print_line, total); };
    int rc = run(!f.testing, nullptr);
    if (rc) return rc;
    if (f.testing) {  // -t: re-solve until the 95% CI half-width < 2.5% of the mean (solver.cpp:295-349)
        std::vector<double> times;
        for (;;) {
            printf("%d ", (int)times.size());
            fflush(stdout);
            double t = 0;
            if ((rc = run(true, &t))) return rc;
            times.push_back(t);
            size_t n = times.size();
            if (n >= 10) {
                double mean = 0, var = 0;
                for (double x : times) mean += x;
                mean /= n;
                for (double x : times) var += (x - mean) * (x - mean);
                var /= (n - 1);
                if (2 * 1.96 * sqrt(var) / sqrt((double)n) < 0.05 * mean) {
                    printf("\nMean execution time is %f pm %f\n", mean, 1.96 * sqrt(var) / sqrt((double)n));
                    break;
                }
            }
        }
    }
    return 0;
}
