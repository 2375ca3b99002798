// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl inside ONE process whose ranks are threads
// (`gnnpe_main --gpus N`, gnn-pe_amd/host/slab_offline.cpp), loaded through GNNPE_RCCL_LIB.
//
// The GPU boxes of this pool have one GPU and RCCL refuses a communicator with the same device twice, so the N >= 2
// branch of slab_offline.cpp -- ncclGetUniqueId / ncclCommInitRank / ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd /
// ncclCommAbort / ncclCommDestroy, with their counts, displacements, group nesting and stream order -- had only ever run with
// a rank as its own peer.  This library implements exactly that part of the API (rccl.h's own declarations: the real header
// is included, so a signature that drifts does not compile) over hipMemcpyAsync between the rank threads' buffers:
//   * a send posted in a group is matched with the peer's recv of the same (source, destination) pair in posting order, as
//     NCCL matches them; the byte counts of the two sides must agree (ncclInvalidArgument otherwise -- a real run would hang
//     or corrupt memory there);
//   * the copy is ordered like the real thing: it waits for the SENDER's stream (an event recorded where ncclSend was
//     enqueued), runs on the RECEIVER's stream, and the sender's stream waits for it before it may reuse the buffer;
//   * ncclGroupEnd returns when every operation of the group has been enqueued (it may wait for the peers' posts, as the
//     real call may); ncclCommInitRank returns when all ranks have called it; ncclCommAbort wakes every waiter of the
//     communicator's world with ncclSystemError.
// FAKE_RCCL_LOG=<file>: at the last ncclCommDestroy / ncclCommAbort of a world one line of JSON -- ranks, groups, sends, recvs,
// bytes moved, largest message, mismatches -- so that a test can see that the schedule really went through here.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct SendPost {
    const void *buf;
    size_t bytes;
    hipEvent_t ready;          // recorded on the sender's stream where ncclSend was enqueued
    hipEvent_t done = nullptr;  // recorded on the receiver's stream behind the copy
    bool taken = false, finished = false;
};

struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int joined = 0, left = 0;
    bool aborted = false;
    // mailbox[src * n + dst]: sends posted and not yet matched, in posting order
    std::vector<std::deque<std::shared_ptr<SendPost>>> box;
    uint64_t groups = 0, sends = 0, recvs = 0, bytes = 0, largest = 0, mismatches = 0, self_pairs = 0;
};

struct Comm {
    std::shared_ptr<World> w;
    int rank;
};

std::mutex g_mu;
std::map<uint64_t, std::shared_ptr<World>> g_worlds;  // by unique id
std::atomic<uint64_t> g_next_id{1};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

void log_world(World &w)
{
    const char *path = getenv("FAKE_RCCL_LOG");
    if (!path) return;
    FILE *f = fopen(path, "a");
    if (!f) return;
    fprintf(f, "{\"ranks\": %d, \"groups\": %llu, \"sends\": %llu, \"recvs\": %llu, \"self_pairs\": %llu, \"bytes\": %llu, \"largest\": %llu, "
               "\"mismatches\": %llu, \"aborted\": %s}\n",
            w.n, (unsigned long long)w.groups, (unsigned long long)w.sends, (unsigned long long)w.recvs, (unsigned long long)w.self_pairs,
            (unsigned long long)w.bytes, (unsigned long long)w.largest, (unsigned long long)w.mismatches, w.aborted ? "true" : "false");
    fclose(f);
}

// the operations of one closed group (or one ungrouped call), in posting order
ncclResult_t run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    ncclResult_t result = ncclSuccess;
    struct Mine {
        std::shared_ptr<SendPost> post;
        hipStream_t stream;
        World *w;
    };
    std::vector<Mine> mine;
    // 1. post every send: the peers' recvs can match from now on
    for (Op &op : ops) {
        if (!op.send) continue;
        World &w = *op.comm->w;
        auto post = std::make_shared<SendPost>();
        post->buf = op.buf;
        post->bytes = op.bytes;
        if (hipEventCreateWithFlags(&post->ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(post->ready, op.stream) != hipSuccess)
            return ncclUnhandledCudaError;
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.box[(size_t)op.comm->rank * w.n + op.peer].push_back(post);
            w.sends++;
            w.self_pairs += op.peer == op.comm->rank;
        }
        w.cv.notify_all();
        mine.push_back(Mine{post, op.stream, &w});
    }
    // 2. every recv: wait for the matching post, copy behind the sender's event on my stream
    for (Op &op : ops) {
        if (op.send) continue;
        World &w = *op.comm->w;
        std::shared_ptr<SendPost> post;
        {
            std::unique_lock<std::mutex> lk(w.mu);
            auto &q = w.box[(size_t)op.peer * w.n + op.comm->rank];
            w.cv.wait(lk, [&] { return w.aborted || !q.empty(); });
            if (w.aborted) return ncclSystemError;
            post = q.front();
            q.pop_front();
            post->taken = true;
            w.recvs++;
            if (post->bytes != op.bytes) {
                w.mismatches++;
                fprintf(stderr, "fake_rccl: rank %d receives %zu bytes from rank %d, which sends %zu\n", op.comm->rank, op.bytes, op.peer, post->bytes);
                result = ncclInvalidArgument;
            } else {
                w.bytes += op.bytes;
                if (op.bytes > w.largest) w.largest = op.bytes;
            }
        }
        hipError_t e = hipStreamWaitEvent(op.stream, post->ready, 0);
        (void)hipEventDestroy(post->ready);  // (released by the runtime once the wait above has passed it)
        if (e == hipSuccess && result == ncclSuccess && op.bytes) e = hipMemcpyAsync(op.buf, post->buf, op.bytes, hipMemcpyDefault, op.stream);
        hipEvent_t done = nullptr;
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done, op.stream);
        {
            std::lock_guard<std::mutex> lk(w.mu);
            post->done = done;
            post->finished = true;
        }
        w.cv.notify_all();
        if (e != hipSuccess) return ncclUnhandledCudaError;
    }
    // 3. my sends: the buffer is mine again once the receiver's copy is behind my stream
    for (Mine &m : mine) {
        std::unique_lock<std::mutex> lk(m.w->mu);
        m.w->cv.wait(lk, [&] { return m.w->aborted || m.post->finished; });
        if (!m.post->finished) return ncclSystemError;
        lk.unlock();
        if (m.post->done) {
            const hipError_t e = hipStreamWaitEvent(m.stream, m.post->done, 0);
            (void)hipEventDestroy(m.post->done);
            if (e != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    return result;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    const uint64_t v = g_next_id++;
    memcpy(id->internal, &v, sizeof(v));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    uint64_t key = 0;
    memcpy(&key, id.internal, sizeof(key));
    if (!key) return ncclInvalidArgument;  // not an id of ncclGetUniqueId
    std::shared_ptr<World> w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto &slot = g_worlds[key];
        if (!slot) {
            slot = std::make_shared<World>();
            slot->n = nranks;
            slot->box.resize((size_t)nranks * nranks);
        }
        w = slot;
    }
    if (w->n != nranks) return ncclInvalidArgument;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->joined++;
        w->cv.notify_all();
        w->cv.wait(lk, [&] { return w->aborted || w->joined >= w->n; });  // like the real call: returns when all ranks are in
        if (w->aborted) return ncclSystemError;
    }
    *comm = reinterpret_cast<ncclComm_t>(new Comm{w, rank});
    return ncclSuccess;
}

static ncclResult_t leave(ncclComm_t comm, bool abort)
{
    if (!comm) return ncclInvalidArgument;
    Comm *c = reinterpret_cast<Comm *>(comm);
    bool last = false;
    {
        std::lock_guard<std::mutex> lk(c->w->mu);
        if (abort) c->w->aborted = true;
        last = ++c->w->left == c->w->n;
        if (!abort)
            for (int p = 0; p < c->w->n; p++)
                if (!c->w->box[(size_t)c->rank * c->w->n + p].empty()) {
                    fprintf(stderr, "fake_rccl: rank %d destroys its communicator with an unmatched send to rank %d\n", c->rank, p);
                    c->w->mismatches++;
                }
    }
    c->w->cv.notify_all();
    if (last || abort) log_world(*c->w);
    delete c;
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { return leave(comm, false); }
ncclResult_t ncclCommAbort(ncclComm_t comm) { return leave(comm, true); }

ncclResult_t ncclGroupStart(void)
{
    t_depth++;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;  // nested groups close with the outermost one
    std::vector<Op> ops;
    ops.swap(t_ops);
    if (!ops.empty()) {
        std::lock_guard<std::mutex> lk(ops[0].comm->w->mu);
        ops[0].comm->w->groups++;
    }
    return run_ops(ops);
}

static ncclResult_t post(bool send, void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm) return ncclInvalidArgument;
    Comm *c = reinterpret_cast<Comm *>(comm);
    const size_t tb = type_bytes(type);
    if (!tb || peer < 0 || peer >= c->w->n || (count && !buf)) return ncclInvalidArgument;
    t_ops.push_back(Op{send, buf, count * tb, peer, c, stream});
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run_ops(ops);
}
ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(true, const_cast<void *>(sendbuff), count, datatype, peer, comm, stream);
}
ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return post(false, recvbuff, count, datatype, peer, comm, stream);
}

const char *ncclGetErrorString(ncclResult_t result)
{
    switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
    case ncclSystemError: return "communicator aborted (fake_rccl)";
    case ncclInvalidArgument: return "invalid argument (fake_rccl)";
    case ncclInvalidUsage: return "invalid usage (fake_rccl)";
    default: return "error (fake_rccl)";
    }
}

}  // extern "C"
