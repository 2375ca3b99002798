// slab_offline.cpp -- `gnnpe_main -m offline --gpus N`: the north-star split of the offline step in the C++ host.
//
// The reference's offline loop (GNN-PE/src/main.cpp:87-119) walks membership.txt's order on one core.  Here the order is
// cut into N contiguous slabs, one host thread + one GPU per slab:
//   * every GPU loads ONLY its slab's adjacency rows (gnnpe_load_rows) -- per-device CSR bytes ~ 1/N + halo;
//   * 1-hop halo (2 hops for -l 3): the rows of the middle vertices come from their owners by an all-to-all-v of
//     device buffers -- RCCL ncclSend/ncclRecv inside one ncclGroup over xGMI, packed by gnnpe_rows_pack and installed
//     (truncated to the slab's rank range on the last hop) by gnnpe_rows_append;
//   * vde rows of every slab are all-gathered the same way; counts and byte sizes cross between the threads in host
//     memory (one process, so these few words need no collective);
//   * every rank emits, renders and pwrite()s its share of all_paths.txt / partition_paths.txt at its byte offset,
//     concurrently; index.dat of partition i is built by rank i mod N from the tuples gathered with one more
//     all-to-all-v.  The files are byte-identical for any N (tests/test_gpu_cli.py).
// `--same-device` (single-GPU boxes, tests) puts every context on device 0; RCCL refuses duplicate devices in one
// communicator, so the same exchange then runs as device-to-device copies between the contexts' buffers
// (`--transport copy`, also usable across devices with peer access).
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "../../include/gnnpe_hip.h"
#include "cli_common.h"
#include "graph_loader.h"
#include "slab_offline.h"

namespace slab {

using namespace cli;
using gnnpe_host::StaticGraph;

namespace {

// An error inside a rank thread must not exit() the process under its siblings (they may be inside HIP or RCCL calls:
// that ends in SIGABRT, not in exit code 1).  Everything in this file reports by exception; the rank's entry function
// records the first message, releases the barrier for the peers, aborts the communicators, and the main thread exits
// non-zero after joining every rank.  (These two hide cli::die / cli::check inside this namespace on purpose.)
struct RankError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void die(const std::string &msg) { throw RankError(msg); }
void check(int rc, const char *what)
{
    if (rc != 0) die(std::string(what) + ": " + gnnpe_last_error());
}
// Test hook (tests/test_gpu_cli.py): GNNPE_FAULT_RANK=<rank>:<stage> makes that rank fail at the named stage ("init" =
// between the two barriers of the communicator set-up, "halo" = before the halo exchange, "emit" = before the emission,
// "index" = before the partitions' tuples are gathered for --index),
// so that the failure path of an N > 1 run -- the peers' release from barriers and RCCL, the exit code -- is exercised.
void fault_point(int r, const char *stage)
{
    static const char *spec = getenv("GNNPE_FAULT_RANK");
    if (!spec) return;
    const char *colon = strchr(spec, ':');
    if (colon && atoi(spec) == r && !strcmp(colon + 1, stage)) die(std::string("injected fault at stage ") + stage);
}

class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu_);
        if (aborted_) throw RankError("stopped: another rank failed");
        const uint64_t gen = gen_;
        if (++count_ == n_) {
            count_ = 0;
            gen_++;
            cv_.notify_all();
        } else {
            cv_.wait(lk, [&] { return gen_ != gen || aborted_; });
            if (gen_ == gen) throw RankError("stopped: another rank failed");
        }
    }
    void abort()
    {
        std::unique_lock<std::mutex> lk(mu_);
        aborted_ = true;
        cv_.notify_all();
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    uint64_t gen_ = 0;
    bool aborted_ = false;
};

// librccl is loaded only by the slab path with the rccl transport: a plain single-GPU run never pays for it
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void load()
    {
        // GNNPE_RCCL_LIB=<path>: another library with RCCL's entry points -- tests/fake_rccl (a stand-in over device copies between
        // the rank threads) walks the N >= 2 schedule on a single-GPU box, where RCCL itself refuses the same device twice
        if (const char *other = getenv("GNNPE_RCCL_LIB")) {
            lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
            if (!lib) die(std::string("cannot load GNNPE_RCCL_LIB=") + other + ": " + dlerror());
        }
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) die(std::string("cannot load librccl: ") + dlerror());
#define GNNPE_SYM(field, name)                                          \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, name));        \
    if (!field) die(std::string("librccl lacks ") + name)
        GNNPE_SYM(GetUniqueId, "ncclGetUniqueId");
        GNNPE_SYM(CommInitRank, "ncclCommInitRank");
        GNNPE_SYM(CommDestroy, "ncclCommDestroy");
        GNNPE_SYM(CommAbort, "ncclCommAbort");
        GNNPE_SYM(GroupStart, "ncclGroupStart");
        GNNPE_SYM(GroupEnd, "ncclGroupEnd");
        GNNPE_SYM(Send, "ncclSend");
        GNNPE_SYM(Recv, "ncclRecv");
        GNNPE_SYM(GetErrorString, "ncclGetErrorString");
#undef GNNPE_SYM
    }
};

// Collectives between the rank threads.  Device payloads move by RCCL (default) or by device-to-device copies; the
// few host words (counts, sizes) are exchanged through shared memory.  With RCCL a rank's piece for ITSELF takes the
// same route as the pieces for its peers (ncclSend + ncclRecv to its own rank inside the group): one code path for
// every N, so `--gpus 1 --transport rccl` on a single-GPU box executes exactly what runs between 8 GPUs.
class Transport {
public:
    Transport(int n, bool use_rccl) : n_(n), rccl_on_(use_rccl), bar_(n), pub_(n), words_((size_t)n * n, 0), comms_(n, nullptr)
    {
        if (rccl_on_) rccl_.load();
    }
    int size() const { return n_; }
    bool rccl() const { return rccl_on_; }
    void barrier() { bar_.wait(); }

    // called by every rank thread after its context exists (the thread's current device is the context's)
    void init_rank(int r)
    {
        if (!rccl_on_) return;
        if (r == 0 && rccl_.GetUniqueId(&uid_) != ncclSuccess) die("ncclGetUniqueId failed");
        barrier();
        fault_point(r, "init");
        // ncclCommInitRank returns only when all n ranks have called it: a peer that died after the barrier above leaves
        // this call blocked for good -- nothing on this thread can end it, the main thread's deadline does (run_offline_slabs)
        ncclComm_t mine = nullptr;
        ncclResult_t rc = rccl_.CommInitRank(&mine, n_, uid_, r);
        if (rc != ncclSuccess) die(std::string("ncclCommInitRank: ") + rccl_.GetErrorString(rc));
        {
            std::lock_guard<std::mutex> lk(fail_mu_);
            if (!failed_) {
                comms_[r] = mine;
                mine = nullptr;
            }
        }
        if (mine) {  // another rank failed meanwhile: fail() will not see this communicator
            rccl_.CommAbort(mine);
            die("stopped: another rank failed");
        }
        barrier();
    }
    // a rank's communicator for a call; never a null handle into RCCL (fail() on another thread clears the entries)
    ncclComm_t comm_of(int r)
    {
        std::lock_guard<std::mutex> lk(fail_mu_);
        if (!comms_[r]) die("stopped: another rank failed");
        return comms_[r];
    }
    void finish_rank(int r)
    {
        ncclComm_t mine = nullptr;
        {
            std::lock_guard<std::mutex> lk(fail_mu_);  // fail() on another thread may be aborting the communicators
            mine = comms_[r];
            comms_[r] = nullptr;
        }
        if (rccl_on_ && mine) rccl_.CommDestroy(mine);
    }
    // a rank failed: keep the first message, let the peers out of their barriers and out of RCCL
    void fail(int r, const std::string &msg)
    {
        {
            std::lock_guard<std::mutex> lk(fail_mu_);
            if (!failed_) first_error_ = "rank " + std::to_string(r) + ": " + msg;
            failed_ = true;
        }
        bar_.abort();
        std::vector<ncclComm_t> live;
        {
            std::lock_guard<std::mutex> lk(fail_mu_);  // every communicator is aborted or destroyed exactly once
            for (auto &cm : comms_) {
                if (cm) live.push_back(cm);
                cm = nullptr;
            }
        }
        if (rccl_on_)
            for (ncclComm_t cm : live) rccl_.CommAbort(cm);
    }
    bool failed()
    {
        std::lock_guard<std::mutex> lk(fail_mu_);
        return failed_;
    }
    const std::string &first_error() const { return first_error_; }

    // every rank contributes one word per peer; returns what the peers addressed to me: out[p] = word rank p gave for r
    std::vector<uint64_t> exchange_words(int r, const std::vector<uint64_t> &to_peer)
    {
        for (int p = 0; p < n_; p++) words_[(size_t)r * n_ + p] = to_peer[p];
        barrier();
        std::vector<uint64_t> out(n_);
        for (int p = 0; p < n_; p++) out[p] = words_[(size_t)p * n_ + r];
        barrier();
        return out;
    }
    // one word from every rank
    std::vector<uint64_t> gather_word(int r, uint64_t v) { return exchange_words(r, std::vector<uint64_t>(n_, v)); }

    // send[scount[0] | scount[1] | ...] -> recv[rcount[0] | rcount[1] | ...]; counts in elements of esize bytes
    void all_to_all_v(int r, gnnpe_ctx *ctx, const void *send, const std::vector<uint64_t> &scount, void *recv,
                      const std::vector<uint64_t> &rcount, size_t esize)
    {
        std::vector<uint64_t> soff(n_ + 1, 0), roff(n_ + 1, 0);
        for (int p = 0; p < n_; p++) {
            soff[p + 1] = soff[p] + scount[p];
            roff[p + 1] = roff[p] + rcount[p];
        }
        if (rccl_on_) {
            void *stream = nullptr;
            check(gnnpe_get_stream(ctx, &stream), "get_stream");
            if (scount[r] != rcount[r]) die("all_to_all_v: a rank's piece for itself has two sizes");
            // (ncclCommAbort from fail() on another thread may end the calls below with an error: that is its purpose)
            const ncclComm_t comm = comm_of(r);
            nccl_ok(rccl_.GroupStart(), "ncclGroupStart");
            for (int p = 0; p < n_; p++) {  // p == r included: the self piece is an RCCL send/recv pair like the others
                if (scount[p]) nccl_ok(rccl_.Send((const char *)send + soff[p] * esize, scount[p] * esize, ncclUint8, p, comm, (hipStream_t)stream), "ncclSend");
                if (rcount[p]) nccl_ok(rccl_.Recv((char *)recv + roff[p] * esize, rcount[p] * esize, ncclUint8, p, comm, (hipStream_t)stream), "ncclRecv");
            }
            nccl_ok(rccl_.GroupEnd(), "ncclGroupEnd");
            check(gnnpe_sync(ctx), "sync");
            return;
        }
        // copy transport: publish where my pieces are, pull the ones addressed to me
        check(gnnpe_sync(ctx), "sync");  // my send buffer is complete
        pub_[r].ptr = (const char *)send;
        pub_[r].off = soff;
        barrier();
        for (int p = 0; p < n_; p++) {
            const uint64_t cnt = pub_[p].off[r + 1] - pub_[p].off[r];
            if (cnt != rcount[p]) die("all_to_all_v: split sizes disagree between ranks");
            if (cnt) check(gnnpe_copy_device(ctx, (char *)recv + roff[p] * esize, pub_[p].ptr + pub_[p].off[r] * esize, cnt * esize), "peer copy");
        }
        check(gnnpe_sync(ctx), "sync");
        barrier();  // nobody reuses a send buffer before every peer has read it
    }

    // every rank's `count` elements to every rank: recv holds the ranks' pieces back to back (counts[] from gather_word)
    void all_gather_v(int r, gnnpe_ctx *ctx, const void *send, const std::vector<uint64_t> &counts, void *recv, size_t esize)
    {
        std::vector<uint64_t> off(n_ + 1, 0);
        for (int p = 0; p < n_; p++) off[p + 1] = off[p] + counts[p];
        if (rccl_on_) {
            void *stream = nullptr;
            check(gnnpe_get_stream(ctx, &stream), "get_stream");
            const ncclComm_t comm = comm_of(r);
            nccl_ok(rccl_.GroupStart(), "ncclGroupStart");
            for (int p = 0; p < n_; p++) {  // p == r included
                if (counts[r]) nccl_ok(rccl_.Send(send, counts[r] * esize, ncclUint8, p, comm, (hipStream_t)stream), "ncclSend");
                if (counts[p]) nccl_ok(rccl_.Recv((char *)recv + off[p] * esize, counts[p] * esize, ncclUint8, p, comm, (hipStream_t)stream), "ncclRecv");
            }
            nccl_ok(rccl_.GroupEnd(), "ncclGroupEnd");
            check(gnnpe_sync(ctx), "sync");
            return;
        }
        check(gnnpe_sync(ctx), "sync");
        pub_[r].ptr = (const char *)send;
        barrier();
        for (int p = 0; p < n_; p++)
            if (counts[p]) check(gnnpe_copy_device(ctx, (char *)recv + off[p] * esize, pub_[p].ptr, counts[p] * esize), "peer copy");
        check(gnnpe_sync(ctx), "sync");
        barrier();
    }

private:
    void nccl_ok(ncclResult_t rc, const char *what)
    {
        if (rc != ncclSuccess) die(std::string(what) + ": " + rccl_.GetErrorString(rc));
    }
    struct Pub {
        const char *ptr = nullptr;
        std::vector<uint64_t> off;
    };
    int n_;
    bool rccl_on_;
    Barrier bar_;
    std::vector<Pub> pub_;
    std::vector<uint64_t> words_;
    Rccl rccl_;
    ncclUniqueId uid_;
    std::vector<ncclComm_t> comms_;
    std::mutex fail_mu_;
    bool failed_ = false;
    std::string first_error_;
};

struct DevMem {  // device buffer of one context
    gnnpe_ctx *ctx = nullptr;
    void *p = nullptr;
    DevMem(gnnpe_ctx *c, uint64_t bytes) : ctx(c) { check(gnnpe_dev_alloc(c, bytes, &p), "device allocation"); }
    ~DevMem() { gnnpe_dev_free(ctx, p); }
    DevMem(const DevMem &) = delete;
    DevMem &operator=(const DevMem &) = delete;
};

struct PinnedBuf {
    char *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) gnnpe_pinned_free(p); }
    void reserve(size_t n)
    {
        if (n <= cap) return;
        if (p) gnnpe_pinned_free(p);
        void *q = nullptr;
        cap = std::max(n + n / 8, (size_t)1 << 20);
        if (gnnpe_pinned_alloc(cap, &q) != 0) die(std::string("pinned host buffer: ") + gnnpe_last_error());
        p = (char *)q;
    }
};

// Background pwrite()s of one rank: the GPU renders the next chunk while this one goes to its file offset.
class RankWriter {
public:
    explicit RankWriter(int n_bufs) : bufs_(n_bufs), free_(n_bufs)
    {
        for (int i = 0; i < n_bufs; i++) free_[i] = i;
        th_ = std::thread([this] { run(); });
    }
    int acquire(size_t bytes)
    {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !free_.empty(); });
        const int k = free_.back();
        free_.pop_back();
        lk.unlock();
        bufs_[k].reserve(bytes);
        return k;
    }
    char *data(int k) { return bufs_[k].p; }
    void submit(int k, int fd, uint64_t offset, size_t bytes)
    {
        std::unique_lock<std::mutex> lk(mu_);
        queue_.push_back({k, fd, offset, bytes});
        cv_.notify_all();
    }
    void close()
    {
        stop();
        if (failed_) die("write error");
    }
    ~RankWriter() { stop(); }  // an exception on the rank thread unwinds through here: never leave the thread joinable

private:
    void stop()
    {
        {
            std::unique_lock<std::mutex> lk(mu_);
            done_ = true;
            cv_.notify_all();
        }
        if (th_.joinable()) th_.join();
    }
    struct Item {
        int k, fd;
        uint64_t offset;
        size_t bytes;
    };
    void run()
    {
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !queue_.empty() || done_; });
                if (queue_.empty()) return;
                it = queue_.front();
                queue_.pop_front();
            }
            size_t done = 0;
            while (done < it.bytes) {
                const ssize_t w = pwrite(it.fd, bufs_[it.k].p + done, it.bytes - done, (off_t)(it.offset + done));
                if (w <= 0) {
                    failed_ = true;
                    break;
                }
                done += (size_t)w;
            }
            {
                std::unique_lock<std::mutex> lk(mu_);
                free_.push_back(it.k);
                cv_.notify_all();
            }
        }
    }
    std::vector<PinnedBuf> bufs_;
    std::vector<int> free_;
    std::deque<Item> queue_;
    std::thread th_;
    std::mutex mu_;
    std::condition_variable cv_;
    bool done_ = false, failed_ = false;
};

// decimal digits of the ids first .. first+cnt-1, each followed by '\n' (partition_paths.txt lines, main.cpp:102-106)
uint64_t id_lines_bytes(uint64_t first, uint64_t cnt)
{
    uint64_t bytes = 0, lo = first, end = first + cnt, pow10 = 10;
    unsigned digits = 1;
    while (lo < end) {
        while (lo >= pow10) {
            pow10 *= 10;
            digits++;
        }
        const uint64_t hi = std::min(end, pow10);
        bytes += (hi - lo) * (digits + 1);
        lo = hi;
    }
    return bytes;
}

// Slabs of (estimated) equal path counts: paths(s) ~ sum_{b in N(s)} (deg b - 1) x share of edge endpoints ranked after s
// (same estimate as gnn-pe_amd/dist.py:plan_slabs).
std::vector<uint32_t> plan_slabs(const StaticGraph &g, const std::vector<uint32_t> &sorted_nodes, int R)
{
    const uint32_t n = g.n;
    const std::vector<uint32_t> &go = g.enum_offsets(), &gn = g.enum_neighbors();  // the rows the enumeration runs on
    std::vector<double> w(n);
    double dsum = 0.0;
    for (uint32_t v = 0; v < n; v++) dsum += go[v + 1] - go[v];
    double later = dsum;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t s = sorted_nodes[i];
        later -= go[s + 1] - go[s];
        double two_hop = 0.0;
        for (uint32_t q = go[s]; q < go[s + 1]; q++) {
            const uint32_t b = gn[q];
            two_hop += (double)(go[b + 1] - go[b]) - 1.0;
        }
        w[i] = two_hop * (dsum > 0 ? later / dsum : 0.0) + 1e-3;
    }
    std::vector<uint32_t> bounds(R + 1, n);
    bounds[0] = 0;
    const double tot = std::accumulate(w.begin(), w.end(), 0.0);
    double acc = 0.0;
    int next = 1;
    for (uint32_t i = 0; i < n && next < R; i++) {
        acc += w[i];
        while (next < R && acc >= tot * next / R) bounds[next++] = i + 1;
    }
    for (int r = 1; r <= R; r++) bounds[r] = std::max(bounds[r], bounds[r - 1]);
    return bounds;
}

struct Shared {
    const Options *o;
    const StaticGraph *g;
    const std::vector<uint32_t> *sorted_nodes, *membership;
    const std::vector<double> *table;
    std::vector<uint32_t> bounds;
    Transport *tp;
    std::vector<int> part_fd;
    int all_fd = -1;
    uint64_t P = 0;
    std::vector<uint64_t> part_total;
    // reporting
    std::vector<uint64_t> held_entries, owned_entries, halo_rows;
    std::vector<double> t_halo, t_count, t_emit, t_index, t_sizing, t_init;
    uint64_t bytes_all = 0, bytes_part = 0;
};

void rank_main(int r, Shared &S)
{
    const Options &o = *S.o;
    const StaticGraph &g = *S.g;
    Transport &tp = *S.tp;
    const int R = tp.size();
    const uint32_t n = g.n, L = o.path_length + 1, e = o.vde_dim, p = o.partition_num;
    const std::vector<uint32_t> &sn = *S.sorted_nodes, &mem = *S.membership;
    const uint32_t lo = S.bounds[r], hi = S.bounds[r + 1], n_own = hi - lo;
    const std::vector<uint32_t> &go = g.enum_offsets(), &gn = g.enum_neighbors();  // the rows the enumeration runs on (graph_loader.h)
    const uint64_t m2 = go[n];
    const auto t0 = Clock::now();

    gnnpe_ctx *ctx = gnnpe_create(o.same_device ? 0 : r);
    if (!ctx) die(std::string("gnnpe_create: ") + gnnpe_last_error());
    // declared first, destroyed last: every device buffer and the output pool below free themselves through the context
    struct CtxGuard {
        gnnpe_ctx *c;
        ~CtxGuard() { gnnpe_destroy(c); }
    } ctx_guard{ctx};
    tp.init_rank(r);
    S.t_init[r] = secs(t0, Clock::now());  // context + communicator (librccl start-up when the transport is rccl)

    // ---- this rank's rows only ----
    std::vector<uint32_t> rows(sn.begin() + lo, sn.begin() + hi);
    std::vector<uint64_t> roff(n_own + 1, 0);
    for (uint32_t k = 0; k < n_own; k++) roff[k + 1] = roff[k] + (go[rows[k] + 1] - go[rows[k]]);
    std::vector<uint32_t> rnbr(roff[n_own]);
    for (uint32_t k = 0; k < n_own; k++)
        std::copy(gn.begin() + go[rows[k]], gn.begin() + go[rows[k] + 1], rnbr.begin() + roff[k]);
    const uint64_t own_entries = roff[n_own];
    S.owned_entries[r] = own_entries;
    check(gnnpe_load_rows(ctx, n, g.labels.data(), n_own, rows.data(), roff.data(), rnbr.data(), m2), "load_rows");
    if (!g.simple) {  // the same rows as the reference's loader holds them: what gen_vde sums over (include/gnnpe_hip.h)
        std::vector<uint64_t> moff(n_own + 1, 0);
        for (uint32_t k = 0; k < n_own; k++) moff[k + 1] = moff[k] + (g.offsets[rows[k] + 1] - g.offsets[rows[k]]);
        std::vector<uint32_t> mnbr(moff[n_own]);
        for (uint32_t k = 0; k < n_own; k++)
            std::copy(g.neighbors.begin() + g.offsets[rows[k]], g.neighbors.begin() + g.offsets[rows[k] + 1], mnbr.begin() + moff[k]);
        check(gnnpe_set_multigraph_rows(ctx, n_own, moff.data(), mnbr.data()), "set_multigraph_rows");
    }
    check(gnnpe_set_order(ctx, sn.data(), mem.data(), p), "set_order");
    check(gnnpe_set_slab(ctx, lo, hi), "set_slab");
    check(gnnpe_set_label_table(ctx, std::max<uint32_t>(g.labels_count, 1), e, S.table->data()), "set_label_table");

    fault_point(r, "halo");
    // ---- halo: one all-to-all-v of adjacency lists per hop (graph structure: done once) ----
    {
        DevMem d_need(ctx, ((uint64_t)n + 1) * 4), d_degin(ctx, ((uint64_t)n + 1) * 4);
        const uint64_t req_cap = (uint64_t)n_own * (uint64_t)std::max(R - 1, 1) + 1;
        DevMem d_req(ctx, req_cap * 4), d_degout(ctx, req_cap * 4);
        DevMem d_pack(ctx, (own_entries * (uint64_t)std::max(R - 1, 1) + 1) * 4), d_in(ctx, (m2 + 1) * 4);
        std::vector<uint32_t> h_deg;
        const int hops = (int)o.path_length - 1;
        for (int hop = 0; hop < hops; hop++) {
            std::vector<uint64_t> need(R, 0);
            check(gnnpe_halo_need(ctx, (uint32_t)R, S.bounds.data(), d_need.p, n, need.data()), "halo_need");
            const std::vector<uint64_t> req = tp.exchange_words(r, need);  // rows every peer wants from me
            const uint64_t n_need = std::accumulate(need.begin(), need.end(), (uint64_t)0);
            const uint64_t n_req = std::accumulate(req.begin(), req.end(), (uint64_t)0);
            if (n_req >= req_cap) die("halo request buffer too small");
            tp.all_to_all_v(r, ctx, d_need.p, need, d_req.p, req, 4);            // the requested vertex ids
            check(gnnpe_rows_degree(ctx, n_req, d_req.p, d_degout.p), "rows_degree");
            tp.all_to_all_v(r, ctx, d_degout.p, req, d_degin.p, need, 4);        // their degrees, back to the requester
            auto segment_sums = [&](const void *dev, const std::vector<uint64_t> &counts, uint64_t tot) {
                h_deg.resize(tot + 1);
                check(gnnpe_copy_to_host(ctx, h_deg.data(), dev, tot * 4), "copy degrees");
                std::vector<uint64_t> sums(R, 0);
                uint64_t at = 0;
                for (int q = 0; q < R; q++)
                    for (uint64_t k = 0; k < counts[q]; k++) sums[q] += h_deg[at++];
                return sums;
            };
            const std::vector<uint64_t> send_sizes = segment_sums(d_degout.p, req, n_req);
            const std::vector<uint64_t> recv_sizes = segment_sums(d_degin.p, need, n_need);
            const uint64_t n_send = std::accumulate(send_sizes.begin(), send_sizes.end(), (uint64_t)0);
            const uint64_t n_recv = std::accumulate(recv_sizes.begin(), recv_sizes.end(), (uint64_t)0);
            if (n_recv > m2) die("halo receive buffer too small");
            check(gnnpe_rows_pack(ctx, n_req, d_req.p, d_pack.p, n_send + 1), "rows_pack");
            tp.all_to_all_v(r, ctx, d_pack.p, send_sizes, d_in.p, recv_sizes, 4);  // the adjacency lists themselves
            const uint32_t min_rank = hop == hops - 1 ? lo : 0u;  // last hop: entries ranked before the slab are never used
            check(gnnpe_rows_append(ctx, n_need, d_need.p, d_degin.p, d_in.p, n_recv, min_rank), "rows_append");
            S.halo_rows[r] += n_need;
        }
    }
    check(gnnpe_rows_held(ctx, nullptr, &S.held_entries[r], nullptr), "rows_held");  // own rows + truncated halo rows
    const auto t1 = Clock::now();

    // ---- vde of the owned rows, all-gather of the slabs' rows ----
    {
        check(gnnpe_vde(ctx, nullptr, nullptr, nullptr), "vde");
        std::vector<uint64_t> counts(R);
        for (int q = 0; q < R; q++) counts[q] = (uint64_t)(S.bounds[q + 1] - S.bounds[q]) * e;
        DevMem d_send(ctx, (counts[r] + 1) * 8), d_all(ctx, ((uint64_t)n * e + 1) * 8);
        check(gnnpe_vde_pack_slab(ctx, lo, hi, d_send.p), "vde_pack_slab");
        tp.all_gather_v(r, ctx, d_send.p, counts, d_all.p, 8);
        uint64_t off = 0;
        for (int q = 0; q < R; q++) {
            if (q != r && counts[q]) check(gnnpe_vde_unpack_slab(ctx, S.bounds[q], S.bounds[q + 1], (char *)d_all.p + off * 8), "vde_unpack_slab");
            off += counts[q];
        }
        check(gnnpe_sync(ctx), "sync");
    }

    // ---- count; global id base and per-partition totals from the per-start counts ----
    std::vector<uint64_t> per_start(std::max<uint32_t>(n_own, 1));
    uint64_t total = 0;
    check(gnnpe_count_paths(ctx, o.path_length, per_start.data(), &total), "count_paths");
    const std::vector<uint64_t> totals = tp.gather_word(r, total);
    uint64_t base = 0, P = 0;
    for (int q = 0; q < R; q++) {
        if (q < r) base += totals[q];
        P += totals[q];
    }
    if (P > 0xFFFFFFFFull && !o.allow_large)
        die(std::to_string(P) + " paths exceed the reference's 32-bit path ids (use --allow-large to write anyway)");
    std::vector<uint64_t> my_part_cnt(p, 0), my_part_bytes(p, 0);
    {
        uint64_t id = base;
        for (uint32_t i = 0; i < n_own; i++) {
            const uint32_t pid = mem[rows[i]];
            my_part_cnt[pid] += per_start[i];
            my_part_bytes[pid] += id_lines_bytes(id, per_start[i]);
            id += per_start[i];
        }
    }
    const auto t2 = Clock::now();

    // ---- sizing pass: bytes of this rank's rows of all_paths.txt (emit ids, count digits; nothing is rendered) ----
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(o.chunk_paths, 4ull << 20), std::max<uint64_t>(total, 1)));
    const uint64_t text_cap = chunk * (11ull * L + 1) + 64;
    DevMem d_part(ctx, chunk * 4), d_sel(ctx, chunk * 8), d_text(ctx, text_cap);
    struct PoolIds {  // the emitted rows: the library's output pool (one allocation, no candidate draw)
        gnnpe_pool *pool = nullptr;
        void *p = nullptr;
        PoolIds(gnnpe_ctx *c, uint64_t rows, uint32_t L_)
        {
            check(gnnpe_output_pool_create(c, rows, L_, 0, 1u | GNNPE_POOL_NO_CALIBRATION, &pool), "output pool");
            check(gnnpe_output_pool_acquire(pool, &p, nullptr, nullptr), "output pool");
        }
        ~PoolIds() { gnnpe_output_pool_destroy(pool); }
    } d_ids(ctx, chunk, L);
    uint64_t my_all_bytes = 0;
    for (uint64_t b = 0; b < total; b += chunk) {
        const uint64_t en = std::min(total, b + chunk);
        uint64_t nb = 0;
        check(gnnpe_fill_paths_device(ctx, b, en, d_ids.p, nullptr, nullptr), "fill_paths");
        check(gnnpe_text_paths(ctx, en - b, L, d_ids.p, nullptr, 0, &nb), "text size");
        my_all_bytes += nb;
    }
    S.t_sizing[r] = secs(t2, Clock::now());  // what the second enumeration costs: reported, see DESIGN section 4
    // offsets of my bytes inside every file
    const std::vector<uint64_t> all_sizes = tp.gather_word(r, my_all_bytes);
    const std::string hdr_all = std::to_string(P) + "\n";
    uint64_t all_off = hdr_all.size();
    for (int q = 0; q < r; q++) all_off += all_sizes[q];
    std::vector<uint64_t> part_off(p), part_tot(p, 0), part_bytes_tot(p, 0);
    std::vector<std::string> part_hdr(p);
    for (uint32_t pid = 0; pid < p; pid++) {
        const std::vector<uint64_t> c = tp.gather_word(r, my_part_cnt[pid]), by = tp.gather_word(r, my_part_bytes[pid]);
        for (int q = 0; q < R; q++) {
            part_tot[pid] += c[q];
            part_bytes_tot[pid] += by[q];
        }
        part_hdr[pid] = std::to_string(part_tot[pid]) + "\n";
        part_off[pid] = part_hdr[pid].size();
        for (int q = 0; q < r; q++) part_off[pid] += by[q];
    }
    if (o.write_index) {  // every rank sees the same totals: all of them stop here, before any file exists
        const std::string big = index_size_problem(part_tot, P, L * e, 1);
        if (!big.empty() && !o.allow_large) die(big + " (use --allow-large to write the files anyway)");
        if (!big.empty() && r == 0) fprintf(stderr, "%s: warning: %s\n", o.tool, big.c_str());
    }
    const std::string partitions_path = o.dataset_path + "gnn-pe/partitions/";
    if (r == 0) {  // create the files at their final size, headers first (main.cpp:102,113)
        auto create = [&](const std::string &path, const std::string &hdr, uint64_t body) {
            const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) die("cannot open " + path + " for writing");
            if (pwrite(fd, hdr.data(), hdr.size(), 0) != (ssize_t)hdr.size() || ftruncate(fd, (off_t)(hdr.size() + body)) != 0)
                die("cannot size " + path);
            return fd;
        };
        uint64_t body = 0;
        for (int q = 0; q < R; q++) body += all_sizes[q];
        S.all_fd = create(o.dataset_path + "/gnn-pe/all_paths.txt", hdr_all, body);  // main.cpp:110 (sic: extra slash)
        S.bytes_all = hdr_all.size() + body;
        for (uint32_t pid = 0; pid < p; pid++) {
            S.part_fd[pid] = create(partitions_path + "partition-" + std::to_string(pid) + "/partition_paths.txt", part_hdr[pid], part_bytes_tot[pid]);
            S.bytes_part += part_hdr[pid].size() + part_bytes_tot[pid];
        }
        S.P = P;
        S.part_total = part_tot;
    }
    tp.barrier();
    fault_point(r, "emit");

    // ---- emit + render + pwrite at this rank's offsets, all ranks concurrently ----
    std::vector<DevMem *> keep(p, nullptr);  // --index: this rank's tuples of every partition, in path-id order
    std::vector<uint64_t> keep_at(p, 0);
    if (o.write_index)
        for (uint32_t pid = 0; pid < p; pid++) keep[pid] = new DevMem(ctx, (my_part_cnt[pid] + 1) * L * 4);
    {
        RankWriter writer(4);
        for (uint64_t b = 0; b < total; b += chunk) {
            const uint64_t en = std::min(total, b + chunk), cnt = en - b;
            uint64_t nb = 0;
            check(gnnpe_fill_paths_device(ctx, b, en, d_ids.p, nullptr, nullptr), "fill_paths");
            check(gnnpe_path_partitions_device(ctx, b, en, d_part.p), "path_partitions");
            check(gnnpe_text_paths(ctx, cnt, L, d_ids.p, d_text.p, text_cap, &nb), "text_paths");
            int k = writer.acquire(nb);
            check(gnnpe_copy_to_host(ctx, writer.data(k), d_text.p, nb), "copy text");
            writer.submit(k, S.all_fd, all_off, nb);
            all_off += nb;
            for (uint32_t pid = 0; pid < p; pid++) {
                uint64_t kk = 0, pb = 0;
                check(gnnpe_select_partition(ctx, cnt, d_part.p, pid, base + b, d_sel.p, &kk), "select_partition");
                if (!kk) continue;
                check(gnnpe_text_ids(ctx, kk, d_sel.p, d_text.p, text_cap, &pb), "text_ids");
                k = writer.acquire(pb);
                check(gnnpe_copy_to_host(ctx, writer.data(k), d_text.p, pb), "copy ids text");
                writer.submit(k, S.part_fd[pid], part_off[pid], pb);
                part_off[pid] += pb;
                if (o.write_index) {
                    check(gnnpe_gather_rows_device(ctx, kk, L, d_sel.p, base + b, d_ids.p, (char *)keep[pid]->p + keep_at[pid] * L * 4), "gather tuples");
                    keep_at[pid] += kk;
                }
            }
        }
        writer.close();
    }
    const auto t3 = Clock::now();

    // ---- index.dat of partition pid: built by rank pid mod R from every rank's tuples (one all-to-all-v each) ----
    if (o.write_index) {
        fault_point(r, "index");  // (the peers are then inside the first partition's gather or its send / recv group)
        for (uint32_t pid = 0; pid < p; pid++) {
            const int owner = (int)(pid % (uint32_t)R);
            const std::vector<uint64_t> cnts = tp.gather_word(r, my_part_cnt[pid]);
            std::vector<uint64_t> scount(R, 0), rcount(R, 0);
            scount[owner] = my_part_cnt[pid] * L;
            uint64_t tot = 0;
            if (r == owner)
                for (int q = 0; q < R; q++) {
                    rcount[q] = cnts[q] * L;
                    tot += cnts[q];
                }
            DevMem d_all(ctx, (tot + 1) * L * 4);
            tp.all_to_all_v(r, ctx, keep[pid]->p, scount, d_all.p, rcount, 4);
            delete keep[pid];
            keep[pid] = nullptr;
            if (r == owner) {
                void *image = nullptr;
                uint64_t nbytes = 0;
                check(gnnpe_build_index_device(ctx, tot, L, d_all.p, &image, &nbytes, nullptr), "build_index");
                const std::string ip = partitions_path + "partition-" + std::to_string(pid) + "/index.dat";
                check(gnnpe_write_device_file(ctx, image, nbytes, ip.c_str()), "write index.dat");
                warn_if_index_too_large_for_reference(ip);
            }
        }
    }
    tp.barrier();
    const auto t4 = Clock::now();
    S.t_halo[r] = secs(t0, t1);
    S.t_count[r] = secs(t1, t2);
    S.t_emit[r] = secs(t2, t3);
    S.t_index[r] = secs(t3, t4);
    tp.finish_rank(r);
}

// what the main thread waits on: the ranks that have returned, and whether one of them failed
struct Done {
    std::mutex mu;
    std::condition_variable cv;
    int returned = 0;
};

void rank_entry(int r, Shared &S, Done &done)
{
    try {
        rank_main(r, S);
    } catch (const std::exception &ex) {
        S.tp->fail(r, ex.what());
    }
    std::lock_guard<std::mutex> lk(done.mu);
    done.returned++;
    done.cv.notify_all();
}

}  // namespace

int run_offline_slabs(const Options &o, const StaticGraph &g, const std::vector<uint32_t> &sorted_nodes,
                      const std::vector<uint32_t> &membership, const std::vector<double> &table, Clock::time_point t_start,
                      Clock::time_point t_loaded)
{
    const int R = o.gpus;
    // --same-device puts every context on device 0, which RCCL refuses inside one communicator: device copies then -- unless the
    // caller names a library that can (GNNPE_RCCL_LIB, see Rccl::load) and asks for the rccl transport explicitly
    const bool standin = getenv("GNNPE_RCCL_LIB") != nullptr && o.transport_explicit;
    const bool use_rccl = o.transport != "copy" && (!o.same_device || standin);
    try {
    const auto t_tp = Clock::now();
    Transport tp(R, use_rccl);  // loads librccl when the transport is rccl: seconds (its code objects), once per process
    const double rccl_load_s = secs(t_tp, Clock::now());
    Shared S;
    S.o = &o;
    S.g = &g;
    S.sorted_nodes = &sorted_nodes;
    S.membership = &membership;
    S.table = &table;
    S.bounds = plan_slabs(g, sorted_nodes, R);
    S.tp = &tp;
    S.part_fd.assign(o.partition_num, -1);
    S.held_entries.assign(R, 0);
    S.owned_entries.assign(R, 0);
    S.halo_rows.assign(R, 0);
    S.t_halo.assign(R, 0);
    S.t_count.assign(R, 0);
    S.t_emit.assign(R, 0);
    S.t_index.assign(R, 0);
    S.t_sizing.assign(R, 0);
    S.t_init.assign(R, 0);
    std::vector<std::thread> th;
    Done done;
    for (int r = 0; r < R; r++) th.emplace_back(rank_entry, r, std::ref(S), std::ref(done));
    {
        // Once a rank has failed its peers get a deadline: a peer blocked inside RCCL where no abort reaches it (e.g.
        // ncclCommInitRank waiting for the rank that died) would hold a join for ever.  The process then ends from HERE with
        // the first error and exit code 1 -- a plain _exit, no re-exec, no destructors racing the stuck threads.
        std::unique_lock<std::mutex> lk(done.mu);
        bool deadline_set = false;
        Clock::time_point deadline;
        while (done.returned < R) {
            if (!deadline_set && tp.failed()) {
                deadline_set = true;
                deadline = Clock::now() + std::chrono::seconds(5);
            }
            if (deadline_set) {
                if (done.cv.wait_until(lk, deadline) == std::cv_status::timeout && done.returned < R) {
                    fprintf(stderr, "%s: %s (%d of %d ranks did not return within 5 s of the failure: still inside a collective)\n", o.tool,
                            tp.first_error().c_str(), R - done.returned, R);
                    fflush(stderr);
                    _exit(1);
                }
            } else {
                done.cv.wait_for(lk, std::chrono::milliseconds(200));
            }
        }
    }
    for (auto &t : th) t.join();
    if (tp.failed()) {  // every rank thread has returned: a clean non-zero exit from the main thread
        fprintf(stderr, "%s: %s\n", o.tool, tp.first_error().c_str());
        return 1;
    }
    if (S.all_fd >= 0 && close(S.all_fd) != 0) die("write error on all_paths.txt");
    for (int fd : S.part_fd)
        if (fd >= 0 && close(fd) != 0) die("write error on partition_paths.txt");
    if (o.timing) {
        auto mx = [](const std::vector<double> &v) { return *std::max_element(v.begin(), v.end()); };
        std::string per;
        for (int r = 0; r < R; r++)
            per += std::string(r ? ", " : "") + "{\"owned_entries\": " + std::to_string(S.owned_entries[r]) + ", \"held_entries\": " +
                   std::to_string(S.held_entries[r]) + ", \"halo_rows\": " + std::to_string(S.halo_rows[r]) + "}";
        fprintf(stderr,
                "{\"paths\": %llu, \"gpus\": %d, \"transport\": \"%s\", \"load_s\": %.3f, \"librccl_load_s\": %.3f, \"context_comm_init_s\": %.3f, \"setup_halo_s\": %.3f, \"vde_count_s\": %.3f, "
                "\"emit_render_write_s\": %.3f, \"sizing_pass_s\": %.3f, \"index_build_s\": %.3f, \"end_to_end_s\": %.3f, \"all_paths_bytes\": %llu, "
                "\"partition_bytes\": %llu, \"csr_entries\": %llu, \"ranks\": [%s]}\n",
                (unsigned long long)S.P, R, use_rccl ? "rccl" : "copy", secs(t_start, t_loaded), rccl_load_s, mx(S.t_init), mx(S.t_halo), mx(S.t_count), mx(S.t_emit),
                mx(S.t_sizing), mx(S.t_index), secs(t_start, Clock::now()), (unsigned long long)S.bytes_all, (unsigned long long)S.bytes_part,
                (unsigned long long)g.enum_offsets()[g.n], per.c_str());
    }
    } catch (const std::exception &ex) {
        fprintf(stderr, "%s: %s\n", o.tool, ex.what());
        return 1;
    }
    return 0;
}

}  // namespace slab
