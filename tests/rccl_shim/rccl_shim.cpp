// rccl_shim.cpp -- a stand-in for librccl.so with the six entry points genedex_amd/csrc/multi.hip resolves by dlopen
// (ncclCommInitAll, ncclCommDestroy, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv), for boxes with ONE GPU: test
// infrastructure, never shipped.  Every call is appended to the file GDX_RCCL_SHIM_LOG names ("send <rank> -> <peer> <count>
// <dtype> <buffer>" / "recv ..."), and ncclGroupEnd pairs every receive with the matching send of its peer -- same element count
// and type, in posting order, as RCCL does -- and moves the bytes with a device copy on the receiver's stream.  A receive
// without its send, a send nobody receives or a size mismatch is an error (RCCL would hang there), so the offsets and sizes
// gdx_multi_locate_many_gather_dev computes are executed and checked without a second device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
struct Comm {
    int rank, n;
};
struct Op {
    bool send;
    int rank, peer, dtype;
    size_t count;
    const void *buf;
    hipStream_t stream;
    bool done;
};
std::mutex g_mu;
std::vector<Op> g_ops;
int g_depth = 0;
const char *g_error = "shim: unmatched or mismatched transfer";

void log_line(const char *fmt, int a, int b, size_t c, int d, const void *p)
{
    const char *path = getenv("GDX_RCCL_SHIM_LOG");
    if (!path) return;
    if (FILE *f = std::fopen(path, "a")) {
        std::fprintf(f, fmt, a, b, c, d, p);
        std::fclose(f);
    }
}
size_t type_bytes(int dtype) { return (dtype == 0 || dtype == 1) ? 1 : (dtype == 2 || dtype == 3) ? 4 : 8; }

int flush()
{
    for (Op &r : g_ops) {
        if (r.send || r.done) continue;
        Op *s = nullptr;
        for (Op &c : g_ops)
            if (c.send && !c.done && c.rank == r.peer && c.peer == r.rank) {
                s = &c;
                break;
            }
        if (!s || s->count != r.count || s->dtype != r.dtype) return 1;
        if (hipStreamSynchronize(s->stream) != hipSuccess) return 1;  // the sender's data is ready when its stream gets there
        if (hipMemcpyAsync(const_cast<void *>(r.buf), s->buf, r.count * type_bytes(r.dtype), hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
            return 1;
        s->done = r.done = true;
    }
    for (const Op &o : g_ops)
        if (!o.done) return 1;
    g_ops.clear();
    return 0;
}
}  // namespace

extern "C" {
int ncclCommInitAll(void **comms, int n, const int *devs)
{
    log_line("init %d ranks, first device %d (%zu %d %p)\n", n, devs ? devs[0] : -1, static_cast<size_t>(0), 0, nullptr);
    for (int i = 0; i < n; i++) comms[i] = new Comm{i, n};
    return 0;
}
int ncclCommDestroy(void *comm)
{
    delete static_cast<Comm *>(comm);
    return 0;
}
int ncclGroupStart()
{
    std::lock_guard<std::mutex> g(g_mu);
    g_depth++;
    return 0;
}
int ncclGroupEnd()
{
    std::lock_guard<std::mutex> g(g_mu);
    if (--g_depth > 0) return 0;
    return flush();
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> g(g_mu);
    const int rank = static_cast<Comm *>(comm)->rank;
    log_line("send %d -> %d %zu %d %p\n", rank, peer, count, dtype, buf);
    g_ops.push_back({true, rank, peer, dtype, count, buf, stream, false});
    return g_depth == 0 ? flush() : 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t stream)
{
    std::lock_guard<std::mutex> g(g_mu);
    const int rank = static_cast<Comm *>(comm)->rank;
    log_line("recv %d <- %d %zu %d %p\n", rank, peer, count, dtype, buf);
    g_ops.push_back({false, rank, peer, dtype, count, buf, stream, false});
    return g_depth == 0 ? flush() : 0;
}
const char *ncclGetErrorString(int) { return g_error; }
}
