// main.cpp -- `gnnpe_main`: drop-in for the reference's `main -m offline` (GNN-PE/src/main.cpp:38-120).
//
// Same flags, same defaults, same input files and the same output files:
//   <f>gnn-pe/membership.txt                       (read;  main.cpp:77-85)
//   <f>gnn-pe/partitions/partition-i/partition_paths.txt   (written; main.cpp:98-108)
//   <f>/gnn-pe/all_paths.txt                       (written; main.cpp:110-119)
// The enumeration, embeddings and text rendering run on MI355X GPUs through the C-ABI in
// include/gnnpe_hip.h; this file is host orchestration and file I/O only.  `-m online` stays the
// reference's own binary for anyone who wants it: it consumes the files written here unchanged.  This tool's own
// `-m online -q <query.graph>` answers the query without any of those files: filter on the GPU, refinement on the
// device too (run_filter); `-m filter` stops after the filter and writes the candidate sets.
//
// Deliberate differences from the reference (all fail-loud instead of silent):
//   - missing membership.txt / partition directories are errors (the reference reads zeros /
//     writes nothing: main.cpp:80,101,110 never check their streams);
//   - `-l 3` writes 4-vertex paths (the rule of custom.h:66-92 with the DFS depth fixed; BASELINE config 5),
//     where the reference enumerates 3-vertex paths whatever -l says and prints a garbage 4th column
//     (SURVEY D4); its online binary cannot read those files.  Other -l values are refused;
//   - path counts beyond 2^32-1 are refused unless --allow-large (the reference's `ui` overflows).
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gnnpe_hip.h"
#include "cli_common.h"
#include "graph_loader.h"
#include "slab_offline.h"

using gnnpe_host::StaticGraph;

namespace {

using namespace cli;

// Grow-only host buffer in pinned memory: device-to-host copies into it run at PCIe speed instead of through the
// driver's pageable staging, and nothing is zero-filled on resize.
struct HostBuf {
    char *p = nullptr;
    size_t cap = 0, len = 0;
    ~HostBuf()
    {
        if (p) gnnpe_pinned_free(p);
    }
    void resize(size_t n)
    {
        if (n > cap) {
            if (p) gnnpe_pinned_free(p);
            p = nullptr;
            cap = std::max(n + n / 8, (size_t)1 << 20);
            void *q = nullptr;
            if (gnnpe_pinned_alloc(cap, &q) != 0) die(std::string("pinned host buffer: ") + gnnpe_last_error());
            p = (char *)q;
        }
        len = n;
    }
    char *data() { return p; }
    void assign(const std::string &s)
    {
        resize(s.size());
        memcpy(p, s.data(), s.size());
    }
};

// Background writer: the GPU renders chunk k+1 while chunk k goes to the file.
class FileSink {
public:
    explicit FileSink(const std::string &path) : path_(path)
    {
        f_ = fopen(path.c_str(), "wb");
        if (!f_) die("cannot open " + path + " for writing");
        setvbuf(f_, nullptr, _IOFBF, 8 << 20);
        th_ = std::thread([this] { run(); });
    }
    HostBuf *acquire()
    {  // a free buffer (two in flight)
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return pending_ < 2; });
        return &bufs_[next_++ & 1];
    }
    void submit(HostBuf *b, size_t n)
    {
        std::unique_lock<std::mutex> lk(mu_);
        queue_.push_back({b, n});
        pending_++;
        cv_.notify_all();
    }
    void write_now(const std::string &s) { submit_copy(s); }
    uint64_t close()
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_ = true;
            cv_.notify_all();
        }
        th_.join();
        if (fclose(f_) != 0 || failed_) die("write error on " + path_);
        return bytes_;
    }

private:
    void submit_copy(const std::string &s)
    {
        HostBuf *b = acquire();
        b->assign(s);
        submit(b, s.size());
    }
    void run()
    {
        for (;;) {
            std::pair<HostBuf *, size_t> item;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !queue_.empty() || done_; });
                if (queue_.empty()) return;
                item = queue_.front();
                queue_.erase(queue_.begin());
            }
            if (item.second && fwrite(item.first->data(), 1, item.second, f_) != item.second) failed_ = true;
            bytes_ += item.second;
            {
                std::unique_lock<std::mutex> lk(mu_);
                pending_--;
                cv_.notify_all();
            }
        }
    }
    std::string path_;
    FILE *f_ = nullptr;
    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::pair<HostBuf *, size_t>> queue_;
    HostBuf bufs_[2];
    int pending_ = 0, next_ = 0;
    bool done_ = false, failed_ = false;
    uint64_t bytes_ = 0;
};

struct Device {
    gnnpe_ctx *ctx = nullptr;
    uint32_t slab_begin = 0, slab_end = 0;
    uint64_t total = 0, base = 0;
};

// -m online / -m filter: the reference's `-m online` (main.cpp:121-185) without its files.  Filter on the GPU -- query plan on the host
// (dfs_query / gen_query_pde), then the leaf test of Partition::query over every enumerated data path.  No
// all_paths.txt, no index.dat: only the data graph and membership.txt (any order gives the same candidate sets).
// Writes <f>gnn-pe/candidates.bin: uint32 n_query_vertices; per query vertex uint32 count + ascending data vertex
// ids -- the reference's candidate_set, ready for its refinement (main.cpp:176-179).  -m online goes on with the
// refinement on the device (gnnpe_refine) and prints the reference's answer line instead of writing the file.
int run_filter(const Options &o)
{
    const auto t0 = Clock::now();
    if (o.path_length != 2) die("-m filter: only -l 2 (the reference's online side only works for it)");
    StaticGraph g;
    std::string err;
    int rc = g.load(o.data_graph, &err, o.strict);
    if (rc == -1) {
        printf("%s\n", err.c_str());
        exit(-1);
    }
    if (rc != 0) die(o.data_graph + ": " + err);
    fputs(g.metadata_text().c_str(), stdout);
    std::vector<uint32_t> sorted_nodes, membership;
    if (gnnpe_host::read_membership(o.dataset_path + "gnn-pe/membership.txt", g.n, o.partition_num, &sorted_nodes,
                                    &membership, &err) != 0)
        die(err);
    uint32_t n_qv = 0, n_qp = 0, *qv = nullptr, *ql = nullptr, *qd = nullptr;
    double *qp = nullptr;
    rc = gnnpe_host_query_plan(o.query_graph.c_str(), o.vde_dim, &n_qv, &n_qp, &qv, &ql, &qd, &qp);
    if (rc == -1) {
        printf("%s\n", gnnpe_last_error());
        exit(-1);
    }
    if (rc != 0) die(o.query_graph + ": " + gnnpe_last_error());
    printf("%u\n", n_qp);  // gen_query_pde prints the plan size (custom.h:629)
    if (gnnpe_device_count() <= 0) die("no HIP device: this tool has no CPU fallback");
    gnnpe_ctx *ctx = gnnpe_create(0);
    if (!ctx) die(std::string("gnnpe_create: ") + gnnpe_last_error());
    const uint32_t n_labels = std::max<uint32_t>(g.labels_count, 1);
    std::vector<double> table((size_t)n_labels * o.vde_dim);
    check(gnnpe_host_label_table(n_labels, o.vde_dim, table.data()), "label table");
    check(load_graph_into(ctx, g), "load_csr");
    check(gnnpe_set_order(ctx, sorted_nodes.data(), membership.data(), o.partition_num), "set_order");
    check(gnnpe_set_label_table(ctx, n_labels, o.vde_dim, table.data()), "set_label_table");
    check(gnnpe_vde(ctx, nullptr, nullptr, nullptr), "vde");
    uint64_t P = 0;
    check(gnnpe_count_paths(ctx, 2, nullptr, &P), "count_paths");
    const uint64_t words = ((uint64_t)g.n + 31) / 32;
    std::vector<uint32_t> bitmap((size_t)n_qv * words);
    double ms = 0.0;
    check(gnnpe_filter_candidates(ctx, n_qp, qv, ql, qd, qp, n_qv, 1e-6 /* custom.h:43 */, bitmap.data(), &ms), "filter");
    if (o.mode == "online") {  // main.cpp:173-181: refinement on the candidate sets, then the answer line
        uint64_t limit = 0xFFFFFFFFull, answers = 0;  // MAX_LIMIT = UINT_MAX (main.cpp:62-69)
        if (o.answers != "MAX") {
            uint32_t lim;
            if (!parse_u32(o.answers, &lim)) die("-n must be MAX or an integer");
            limit = lim;
        }
        double refine_ms = 0.0;
        // the refinement lives in libgnnpe_online.so (include/gnnpe_online.h: out of scope, frozen), beside this binary
        typedef int (*refine_fn)(gnnpe_ctx *, const char *, const uint32_t *, uint64_t, uint64_t *, double *);
        void *online = dlopen((exe_dir() + "libgnnpe_online.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!online) online = dlopen("libgnnpe_online.so", RTLD_NOW | RTLD_GLOBAL);
        if (!online) die(std::string("-m online needs libgnnpe_online.so beside ") + o.tool + ": " + dlerror());
        refine_fn refine = (refine_fn)dlsym(online, "gnnpe_refine");
        if (!refine) die("libgnnpe_online.so does not export gnnpe_refine");
        check(refine(ctx, o.query_graph.c_str(), bitmap.data(), limit, &answers, &refine_ms), "refine");
        gnnpe_destroy(ctx);
        printf("Answer Number: %llu Query Time (ms): %g\n", (unsigned long long)answers, ms + refine_ms);
        if (o.timing)
            fprintf(stderr, "{\"paths\": %llu, \"query_paths\": %u, \"filter_device_ms\": %.3f, \"refine_ms\": %.3f, "
                            "\"end_to_end_s\": %.3f}\n",
                    (unsigned long long)P, n_qp, ms, refine_ms, secs(t0, Clock::now()));
        gnnpe_host_free(qv);
        gnnpe_host_free(ql);
        gnnpe_host_free(qd);
        gnnpe_host_free(qp);
        return 0;
    }
    gnnpe_destroy(ctx);
    const std::string out = o.dataset_path + "gnn-pe/candidates.bin";
    FILE *f = fopen(out.c_str(), "wb");
    if (!f) die("cannot write " + out);
    fwrite(&n_qv, 4, 1, f);
    for (uint32_t u = 0; u < n_qv; u++) {
        std::vector<uint32_t> ids;
        for (uint64_t w = 0; w < words; w++)
            for (uint32_t bits = bitmap[(size_t)u * words + w]; bits; bits &= bits - 1)
                ids.push_back((uint32_t)(w * 32 + __builtin_ctz(bits)));
        const uint32_t c = (uint32_t)ids.size();
        fwrite(&c, 4, 1, f);
        if (c) fwrite(ids.data(), 4, c, f);
    }
    if (fclose(f) != 0) die("write failed on " + out);
    gnnpe_host_free(qv);
    gnnpe_host_free(ql);
    gnnpe_host_free(qd);
    gnnpe_host_free(qp);
    if (o.timing)
        fprintf(stderr, "{\"paths\": %llu, \"query_paths\": %u, \"filter_device_ms\": %.3f, \"end_to_end_s\": %.3f}\n",
                (unsigned long long)P, n_qp, ms, secs(t0, Clock::now()));
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    const double before_main_s = secs_since_process_start();
    Options o = parse_args(argc, argv);
    const auto t_start = Clock::now();

    // main.cpp:58-59: path_length += 1; pde_dim = vde_dim * path_length
    if (o.path_length != 2 && o.path_length != 3)
        die("-l " + std::to_string(o.path_length) + ": only -l 2 and -l 3 are supported (the reference always enumerates "
            "3-vertex paths and mis-prints them for any other -l)");
    if (o.path_length == 3)
        fprintf(stderr, "note: -l 3 writes 4-vertex paths; the reference's online binary only reads -l 2 files (SURVEY D4)\n");
    const uint32_t L = o.path_length + 1;
    if (o.answers != "MAX") {  // main.cpp:62-69 (MAX_LIMIT is an online-only knob)
        uint32_t lim;
        if (!parse_u32(o.answers, &lim)) die("-n must be MAX or an integer");
    }
    if (o.partition_num == 0) die("-p must be >= 1");
    if (o.mode == "online" || o.mode == "filter") return run_filter(o);
    if (o.mode != "offline") return 0;  // the reference does nothing for other modes
    if (o.write_index && gnnpe_index_file_bytes(1, L * o.vde_dim, 0) == 0)  // before the graph is read or a GPU is touched
        die("--index: an entry of " + std::to_string(L * o.vde_dim) + " dimensions (" + std::to_string(16ull * L * o.vde_dim + 4) +
            " bytes) gives a node capacity below 3 in a 4096-byte block (rtnode.cpp:27-28); use a smaller -e");

    // HIP start-up (0.1-0.2 s: runtime initialisation, context, code objects) runs beside the graph load.  An error
    // exit in between (die() -> exit()) must not tear the process down under that thread: it is joined first.
    static int early_ndev = -1;
    static gnnpe_ctx *early_ctx = nullptr;
    static std::thread *warm = nullptr;
    const bool slab_path = o.gpus > 1 || o.transport_explicit;
    if (!slab_path) {
        warm = new std::thread([] {
            early_ndev = gnnpe_device_count();
            if (early_ndev > 0) early_ctx = gnnpe_create(0);
        });
        atexit([] {
            if (warm && warm->joinable()) warm->join();
        });
    }

    StaticGraph g;
    std::string err;
    int rc = g.load(o.data_graph, &err, o.strict);
    if (rc == -1) {  // graph.cpp:166-169
        printf("%s\n", err.c_str());
        exit(-1);
    }
    if (rc != 0) die(o.data_graph + ": " + err);
    fputs(g.metadata_text().c_str(), stdout);  // printGraphMetaData, main.cpp:73
    const auto t_loaded = Clock::now();

    std::vector<uint32_t> sorted_nodes, membership;
    if (gnnpe_host::read_membership(o.dataset_path + "gnn-pe/membership.txt", g.n, o.partition_num, &sorted_nodes,
                                    &membership, &err) != 0)
        die(err);
    const std::string partitions_path = o.dataset_path + "gnn-pe/partitions/";
    for (uint32_t i = 0; i < o.partition_num; i++)
        if (!is_dir(partitions_path + "partition-" + std::to_string(i)))
            die("missing directory " + partitions_path + "partition-" + std::to_string(i) + "/ (the prep step creates it)");

    // ---- device setup ----
    if (warm) {
        warm->join();
        delete warm;
        warm = nullptr;
    }
    const int ndev = early_ndev >= 0 ? early_ndev : gnnpe_device_count();
    if (ndev <= 0) die("no HIP device: this tool has no CPU fallback");
    if (o.gpus < 1 || (o.gpus > ndev && !o.same_device)) die("--gpus " + std::to_string(o.gpus) + " but " + std::to_string(ndev) + " device(s) present");
    std::vector<double> table((size_t)std::max<uint32_t>(g.labels_count, 1) * o.vde_dim);
    check(gnnpe_host_label_table(std::max<uint32_t>(g.labels_count, 1), o.vde_dim, table.data()), "label table");
    if (slab_path) {
        // the north-star split: one thread + one GPU per slab of the order, each with its own rows only, halo by
        // all-to-all-v, concurrent emit and pwrite (host/slab_offline.cpp); `--gpus 1 --transport rccl` runs the same
        // code as a 1-rank communicator
        if (o.sidecars) die("--sidecars is a single-GPU option");
        return slab::run_offline_slabs(o, g, sorted_nodes, membership, table, t_start, t_loaded);
    }
    // --index: the partitions' index.dat files are the work of a SECOND context on the same device, on a thread of its own, beside
    // the text files: the two share nothing but the host arrays, the 22 GB of index blocks cross PCIe while the text is rendered
    // and written, and the files of both kinds go to the file system side by side (one file takes ~10 GB/s on these boxes whatever
    // the number of writers, scripts/fs_write_probe.cpp; sixteen files take 30-40).  The thread builds its context and its count,
    // then waits for the main thread's size check before it writes anything.
    struct IndexJob {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        int go = 0;  // 0: wait, 1: write the files, -1: give up
        double seconds = 0.0;
        std::string error;
    };
    static IndexJob *index_job = nullptr;
    std::vector<std::string> ips, aps;
    for (uint32_t pid = 0; pid < o.partition_num; pid++) {
        ips.push_back(partitions_path + "partition-" + std::to_string(pid) + "/index.dat");
        aps.push_back(partitions_path + "partition-" + std::to_string(pid) + "/aux_index.bin");
    }
    if (o.write_index) {
        index_job = new IndexJob();
        index_job->th = std::thread([&, job = index_job] {
            const auto i0 = Clock::now();
            auto fail = [&](const char *what) { job->error = std::string(what) + ": " + gnnpe_last_error(); };
            // test hook (tests/test_gpu_cli.py): GNNPE_FAULT_RANK=index:create fails this thread as an out-of-memory would
            const char *fault = getenv("GNNPE_FAULT_RANK");
            if (fault && !strcmp(fault, "index:create")) {
                job->error = "gnnpe_create (index): injected fault";
                return;
            }
            gnnpe_ctx *ctx = gnnpe_create(0);
            if (!ctx) return fail("gnnpe_create (index)");
            uint64_t P2 = 0;
            if (load_graph_into(ctx, g) != 0 ||
                gnnpe_set_order(ctx, sorted_nodes.data(), membership.data(), o.partition_num) != 0 ||
                gnnpe_set_label_table(ctx, std::max<uint32_t>(g.labels_count, 1), o.vde_dim, table.data()) != 0 ||
                gnnpe_vde(ctx, nullptr, nullptr, nullptr) != 0 || gnnpe_count_paths(ctx, o.path_length, nullptr, &P2) != 0) {
                fail("index context");
                gnnpe_destroy(ctx);
                return;
            }
            {
                std::unique_lock<std::mutex> lk(job->mu);
                job->cv.wait(lk, [&] { return job->go != 0; });
                if (job->go < 0) {
                    gnnpe_destroy(ctx);
                    return;
                }
            }
            std::vector<const char *> ipp, app;
            for (uint32_t pid = 0; pid < o.partition_num; pid++) {
                ipp.push_back(ips[pid].c_str());
                app.push_back(aps[pid].c_str());
            }
            // every partition's image first, then the files side by side; with --sidecars also the trees' auxiliary index
            // (custom.h:268-364), which the online side otherwise rebuilds on every start
            if (gnnpe_build_index_files(ctx, o.partition_num, ipp.data(), o.sidecars ? app.data() : nullptr) != 0) fail("build_index_files");
            gnnpe_destroy(ctx);
            job->seconds = secs(i0, Clock::now());
        });
        atexit([] {  // an error exit on the main thread: the index thread is told to give up and joined first
            if (index_job && index_job->th.joinable()) {
                {
                    std::unique_lock<std::mutex> lk(index_job->mu);
                    if (index_job->go == 0) index_job->go = -1;
                }
                index_job->cv.notify_all();
                index_job->th.join();
            }
        });
    }
    std::vector<Device> devs(1);
    devs[0].ctx = early_ctx ? early_ctx : gnnpe_create(0);  // (again on this thread if the early one failed: for its message)
    if (!devs[0].ctx) die(std::string("gnnpe_create: ") + gnnpe_last_error());
    check(load_graph_into(devs[0].ctx, g), "load_csr");
    check(gnnpe_set_order(devs[0].ctx, sorted_nodes.data(), membership.data(), o.partition_num), "set_order");
    check(gnnpe_set_label_table(devs[0].ctx, std::max<uint32_t>(g.labels_count, 1), o.vde_dim, table.data()), "set_label_table");
    const auto t_setup = Clock::now();

    // ---- R4 + R2 count on device 0 over the whole order: per-start counts size everything ----
    std::vector<double> hx, hnx, hvde;
    if (o.sidecars) {
        hx.resize((size_t)g.n * o.vde_dim);
        hnx.resize(hx.size());
        hvde.resize(hx.size());
    }
    for (int d = 0; d < o.gpus; d++)
        check(gnnpe_vde(devs[d].ctx, d == 0 && o.sidecars ? hx.data() : nullptr, d == 0 && o.sidecars ? hnx.data() : nullptr,
                        d == 0 && o.sidecars ? hvde.data() : nullptr), "vde");
    std::vector<uint64_t> per_start(g.n);
    uint64_t P = 0;
    check(gnnpe_count_paths(devs[0].ctx, o.path_length, per_start.data(), &P), "count_paths");
    if (P > 0xFFFFFFFFull && !o.allow_large)
        die(std::to_string(P) + " paths exceed the reference's 32-bit path ids (use --allow-large to write anyway)");
    // slabs with equal path counts
    {
        uint64_t acc = 0;
        uint32_t i = 0;
        for (int d = 0; d < o.gpus; d++) {
            devs[d].slab_begin = i;
            devs[d].base = acc;
            const uint64_t target = P * (uint64_t)(d + 1) / (uint64_t)o.gpus;
            while (i < g.n && (acc < target || d == o.gpus - 1)) acc += per_start[i++];
            devs[d].slab_end = i;
            devs[d].total = acc - devs[d].base;
        }
    }
    std::vector<uint64_t> part_count(o.partition_num, 0);  // main.cpp:102: header of partition_paths.txt
    for (uint32_t i = 0; i < g.n; i++) part_count[membership[sorted_nodes[i]]] += per_start[i];
    if (o.write_index) {  // before anything is written
        const std::string big = index_size_problem(part_count, P, (o.path_length + 1) * o.vde_dim,
                                                   // the pair-major (l = 2) and triple-major (l = 3) builds exist at the widths with a specialised
                                                   // enumeration; else the tuple-array build
                                                   (o.vde_dim <= 4 || o.vde_dim == 8) ? 0 : 1);
        if (!big.empty() && !o.allow_large) die(big + " (use --allow-large to write the files anyway)");
        if (!big.empty()) fprintf(stderr, "%s: warning: %s\n", o.tool, big.c_str());
        {
            std::unique_lock<std::mutex> lk(index_job->mu);
            index_job->go = 1;  // the size check has passed: the index thread may write
        }
        index_job->cv.notify_all();
    }
    for (int d = 0; d < o.gpus; d++) {
        if (o.gpus > 1) {
            check(gnnpe_set_slab(devs[d].ctx, devs[d].slab_begin, devs[d].slab_end), "set_slab");
            uint64_t t = 0;
            check(gnnpe_count_paths(devs[d].ctx, o.path_length, nullptr, &t), "count_paths(slab)");
            if (t != devs[d].total) die("slab count mismatch");
        }
    }
    const auto t_counted = Clock::now();

    // ---- emit + render + write, chunk by chunk, slab by slab (rank order = file order) ----
    FileSink all_paths(o.dataset_path + "/gnn-pe/all_paths.txt");  // main.cpp:110 (sic: extra slash)
    all_paths.write_now(std::to_string(P) + "\n");
    std::vector<FileSink *> part_files;
    for (uint32_t i = 0; i < o.partition_num; i++) {
        part_files.push_back(new FileSink(partitions_path + "partition-" + std::to_string(i) + "/partition_paths.txt"));
        part_files[i]->write_now(std::to_string(part_count[i]) + "\n");
    }
    // --sidecars: the rows again as uint32 tuples (paths.bin), so that an online start can skip the text re-parse of
    // gen_pde (custom.h:546-572); read back by gnnpe_host_load_path_sidecar
    FILE *paths_bin = nullptr;
    std::vector<uint32_t> ids_host;
    if (o.sidecars) {
        paths_bin = fopen((o.dataset_path + "gnn-pe/paths.bin").c_str(), "wb");
        if (!paths_bin) die("cannot write paths.bin");
        setvbuf(paths_bin, nullptr, _IOFBF, 8 << 20);
        check(gnnpe_host_write_paths_header(paths_bin, L, P), "paths.bin");
    }
    double t_gpu = 0.0;
    for (int d = 0; d < o.gpus; d++) {
        gnnpe_ctx *ctx = devs[d].ctx;
        const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(o.chunk_paths, devs[d].total));
        void *d_ids = nullptr, *d_part = nullptr, *d_sel = nullptr, *d_text = nullptr;
        const uint64_t text_cap = chunk * (11ull * L + 1) + 64;
        // the emitted rows live in the library's output pool: one allocation (no candidate draw since round 4); when one
        // chunk holds every path the pool times both emit shapes into it and later fills take the faster one
        gnnpe_pool *pool = nullptr;
        check(gnnpe_output_pool_create(ctx, chunk, L, 0, 1u | GNNPE_POOL_NO_CALIBRATION, &pool), "output pool");
        check(gnnpe_output_pool_acquire(pool, &d_ids, nullptr, nullptr), "output pool");
        check(gnnpe_dev_alloc(ctx, chunk * 4, &d_part), "alloc part");
        check(gnnpe_dev_alloc(ctx, chunk * 8, &d_sel), "alloc sel");
        check(gnnpe_dev_alloc(ctx, text_cap, &d_text), "alloc text");
        for (uint64_t b = 0; b < devs[d].total; b += chunk) {
            const uint64_t e = std::min(devs[d].total, b + chunk), cnt = e - b;
            const auto g0 = Clock::now();
            check(gnnpe_fill_paths_device(ctx, b, e, d_ids, nullptr, nullptr), "fill_paths");
            check(gnnpe_path_partitions_device(ctx, b, e, d_part), "path_partitions");
            uint64_t nb = 0;
            check(gnnpe_text_paths(ctx, cnt, L, d_ids, d_text, text_cap, &nb), "text_paths");
            HostBuf *buf = all_paths.acquire();
            buf->resize(nb);
            check(gnnpe_copy_to_host(ctx, buf->data(), d_text, nb), "copy text");
            t_gpu += secs(g0, Clock::now());
            all_paths.submit(buf, nb);
            if (paths_bin) {
                ids_host.resize(cnt * L);
                check(gnnpe_copy_to_host(ctx, ids_host.data(), d_ids, cnt * L * 4), "copy ids");
                if (fwrite(ids_host.data(), 4, cnt * L, paths_bin) != cnt * L) die("write error on paths.bin");
            }
            for (uint32_t pid = 0; pid < o.partition_num; pid++) {
                const auto g1 = Clock::now();
                uint64_t k = 0, pb = 0;
                check(gnnpe_select_partition(ctx, cnt, d_part, pid, devs[d].base + b, d_sel, &k), "select_partition");
                check(gnnpe_text_ids(ctx, k, d_sel, d_text, text_cap, &pb), "text_ids");
                HostBuf *pbuf = part_files[pid]->acquire();
                pbuf->resize(pb);
                check(gnnpe_copy_to_host(ctx, pbuf->data(), d_text, pb), "copy ids text");
                t_gpu += secs(g1, Clock::now());
                part_files[pid]->submit(pbuf, pb);
            }
        }
        gnnpe_output_pool_destroy(pool);
        gnnpe_dev_free(ctx, d_part);
        gnnpe_dev_free(ctx, d_sel);
        gnnpe_dev_free(ctx, d_text);
    }
    uint64_t bytes_all = all_paths.close(), bytes_part = 0;
    for (auto *f : part_files) {
        bytes_part += f->close();
        delete f;
    }
    const auto t_written = Clock::now();

    if (paths_bin && fclose(paths_bin) != 0) die("write error on paths.bin");
    if (o.sidecars) {  // ignored by the reference; carrier for the embedding parity check and the online data load
        FILE *f = fopen((o.dataset_path + "gnn-pe/vde.bin").c_str(), "wb");
        if (!f) die("cannot write vde.bin");
        uint32_t hdr[2] = {g.n, o.vde_dim};
        fwrite(hdr, 4, 2, f);
        fwrite(hx.data(), 8, hx.size(), f);
        fwrite(hnx.data(), 8, hnx.size(), f);
        fwrite(hvde.data(), 8, hvde.size(), f);
        fclose(f);
    }
    double t_index = 0.0, t_index_wait = 0.0;
    if (o.write_index) {  // optional accelerator: the reference online run reuses an existing index.dat (custom.h:222-235)
        const auto i0 = Clock::now();
        index_job->th.join();
        t_index_wait = secs(i0, Clock::now());
        t_index = index_job->seconds;
        if (!index_job->error.empty()) {
            // the second context is an overlap, not a requirement: two copies of the graph and its count state plus
            // the main thread's chunk buffers may not fit where one did.  The text is written and its buffers are freed: build
            // the files on the main context, one after the other, as round 4 did.
            fprintf(stderr, "%s: note: index context failed (%s); building index.dat on the main context instead\n", o.tool,
                    index_job->error.c_str());
            std::vector<const char *> ipp, app;
            for (uint32_t pid = 0; pid < o.partition_num; pid++) {
                ipp.push_back(ips[pid].c_str());
                app.push_back(aps[pid].c_str());
            }
            check(gnnpe_build_index_files(devs[0].ctx, o.partition_num, ipp.data(), o.sidecars ? app.data() : nullptr),
                  "build_index_files (main context)");
            t_index = secs(i0, Clock::now());
        }
        for (auto &ip : ips) warn_if_index_too_large_for_reference(ip);
    }
    for (auto &d : devs) gnnpe_destroy(d.ctx);

    if (o.timing) {
        fprintf(stderr,
                "{\"paths\": %llu, \"gpus\": %d, \"load_s\": %.3f, \"setup_s\": %.3f, \"vde_count_s\": %.3f, "
                "\"emit_render_copy_s\": %.3f, \"write_total_s\": %.3f, \"text_gb_per_s\": %.2f, \"end_to_end_s\": %.3f, "
                "\"all_paths_bytes\": %llu, \"partition_bytes\": %llu, \"index_build_s\": %.3f, \"index_wait_after_text_s\": %.3f, "
                "\"before_main_s\": %.2f}\n",
                (unsigned long long)P, o.gpus, secs(t_start, t_loaded), secs(t_loaded, t_setup), secs(t_setup, t_counted),
                t_gpu, secs(t_counted, t_written), (bytes_all + bytes_part) / 1e9 / std::max(1e-9, secs(t_counted, t_written)),
                secs(t_start, Clock::now()), (unsigned long long)bytes_all,
                (unsigned long long)bytes_part, t_index, t_index_wait, before_main_s);
    }
    // (What a caller's wall-clock holds beyond end_to_end_s: 0.02 s before main() -- `before_main_s` -- and 0.10-0.15 s after it at
    // config 3, the kernel unmapping the process; leaving through _exit() instead of the runtime's unwinding measured the same.)
    return 0;
}
