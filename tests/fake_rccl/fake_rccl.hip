// fake_rccl.hip — TEST INFRASTRUCTURE, not part of the product.
//
// RCCL refuses two ranks on one device, and the development box has one GPU, so the
// multi-rank form of the library's native step (mi_part_comm_init + mi_part_spmv_dev: pack and
// exchange on a comm stream, interior rows beside them, boundary rows behind the exchange) could
// not be run there.  This file is a stand-in for librccl with just the entry points that step
// resolves by name (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGroupStart/End,
// ncclSend, ncclRecv, ncclAllGather, ncclGetErrorString), for ranks that are THREADS of one process sharing one
// GPU.  A send copies its payload into a device mailbox on the caller's stream and publishes an
// event; a receive blocks the calling host thread until the matching message is published, makes
// its stream wait for the event and copies the payload out.  Every rank posts all sends of a
// group before it blocks on a receive, so a symmetric exchange cannot deadlock.  What this tests
// is the library's own stream/event ordering, offsets and counts — not RCCL.
//
// Selected with MI355_RCCL_LIBRARY=<path to libfake_rccl.so> (tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

namespace {
struct Id {
    char internal[128];
};
struct Message {
    void* buf = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;
};
struct World {
    int nranks = 0, joined = 0, left = 0;
    std::map<std::tuple<int, int, long long>, Message> box; // (src, dst, sequence) -> message
    std::map<std::pair<int, int>, long long> sent, received;
    std::vector<void*> garbage;
};
struct Comm {
    World* w;
    int rank;
};
struct Op {
    bool send;
    void* buf;
    size_t bytes;
    int peer;
    Comm* comm;
    hipStream_t stream;
};
std::mutex g_mu;
std::condition_variable g_cv;
std::map<std::string, World*> g_worlds;
int g_next_id = 1;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

int run_send(const Op& o)
{
    Message m;
    m.bytes = o.bytes;
    if (hipMalloc(&m.buf, o.bytes ? o.bytes : 8) != hipSuccess) return 1;
    if (hipMemcpyAsync(m.buf, o.buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&m.ready, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(m.ready, o.stream) != hipSuccess) return 1;
    std::lock_guard<std::mutex> lk(g_mu);
    World* w = o.comm->w;
    const long long seq = w->sent[{o.comm->rank, o.peer}]++;
    w->box[std::make_tuple(o.comm->rank, o.peer, seq)] = m;
    g_cv.notify_all();
    return 0;
}

int run_recv(const Op& o)
{
    Message m;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        World* w = o.comm->w;
        const long long seq = w->received[{o.peer, o.comm->rank}]++;
        const auto key = std::make_tuple(o.peer, o.comm->rank, seq);
        g_cv.wait(lk, [&] { return w->box.count(key) != 0; });
        m = w->box[key];
        w->box.erase(key);
        w->garbage.push_back(m.buf);
    }
    if (m.bytes != o.bytes) {
        fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from %d, message holds %zu\n", o.comm->rank, o.bytes, o.peer, m.bytes);
        return 2;
    }
    if (hipStreamWaitEvent(o.stream, m.ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(o.buf, m.buf, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) return 1;
    return 0;
}
} // namespace

extern "C" int ncclGetUniqueId(void* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id, 0, 128);
    snprintf(static_cast<char*>(id), 128, "fake-rccl-world-%d", g_next_id++);
    return 0;
}

extern "C" int ncclCommInitRank(void** comm, int nranks, Id id, int rank)
{
    std::unique_lock<std::mutex> lk(g_mu);
    const std::string key(id.internal, strnlen(id.internal, 128));
    World*& w = g_worlds[key];
    if (!w) {
        w = new World();
        w->nranks = nranks;
    }
    if (w->nranks != nranks || rank < 0 || rank >= nranks) return 4;
    w->joined++;
    g_cv.notify_all();
    World* ww = w;
    g_cv.wait(lk, [&] { return ww->joined >= ww->nranks; }); // collective, like the real one
    *comm = new Comm{ww, rank};
    return 0;
}

extern "C" int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    (void)hipDeviceSynchronize();
    std::lock_guard<std::mutex> lk(g_mu);
    if (++c->w->left == c->w->nranks) {
        for (void* p : c->w->garbage) (void)hipFree(p);
        c->w->garbage.clear();
    }
    delete c;
    return 0;
}

extern "C" int ncclGroupStart()
{
    t_depth++;
    return 0;
}

extern "C" int ncclGroupEnd()
{
    if (--t_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(t_ops);
    for (const Op& o : ops)
        if (o.send)
            if (int rc = run_send(o)) return rc;
    for (const Op& o : ops)
        if (!o.send)
            if (int rc = run_recv(o)) return rc;
    return 0;
}

extern "C" int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s)
{
    if (dtype != 8) return 4; // ncclFloat64 only
    Op o{true, const_cast<void*>(buf), count * 8, peer, static_cast<Comm*>(comm), s};
    if (t_depth > 0) {
        t_ops.push_back(o);
        return 0;
    }
    return run_send(o);
}

extern "C" int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s)
{
    if (dtype != 8) return 4;
    Op o{false, buf, count * 8, peer, static_cast<Comm*>(comm), s};
    if (t_depth > 0) {
        t_ops.push_back(o);
        return 0;
    }
    return run_recv(o);
}

// every rank's `count` doubles to every rank, rank p's slice at recvbuf + p * count (ncclAllGather of rccl.h).  Built from the
// mailbox pieces above: my slice goes out to every peer first, then theirs are awaited — all ranks send before any blocks, so the
// collective cannot deadlock; the local slice is a plain copy on the caller's stream.
extern "C" int ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, int dtype, void* comm, hipStream_t s)
{
    if (dtype != 8) return 4;
    Comm* c = static_cast<Comm*>(comm);
    const int n = c->w->nranks;
    for (int p = 0; p < n; p++)
        if (p != c->rank)
            if (int rc = run_send(Op{true, const_cast<void*>(sendbuf), count * 8, p, c, s})) return rc;
    if (hipMemcpyAsync(static_cast<char*>(recvbuf) + (size_t)c->rank * count * 8, sendbuf, count * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    for (int p = 0; p < n; p++)
        if (p != c->rank)
            if (int rc = run_recv(Op{false, static_cast<char*>(recvbuf) + (size_t)p * count * 8, count * 8, p, c, s})) return rc;
    return 0;
}

extern "C" const char* ncclGetErrorString(int rc)
{
    switch (rc) {
    case 0: return "ok";
    case 1: return "fake_rccl: HIP call failed";
    case 2: return "fake_rccl: message size mismatch";
    default: return "fake_rccl: invalid argument";
    }
}
