// TEST INFRASTRUCTURE: a stand-in for librccl with the eight entry points rtw_gather_rows uses, moving the bytes between processes through
// named pipes (one per ordered pair of ranks, under $RTW_LOOPBACK_DIR) and through host memory.  It lets two ranks that SHARE one GPU -- which
// RCCL itself refuses -- run the product's gather plan (which rows, which offsets, which peer, in which order) end to end on a one-GPU box.
// It says nothing about RCCL's own behaviour; the 8-GPU run of bench.py does that.  Loaded through RTW_RCCL_LIBRARY (include/rtwin.h).
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
struct LoopComm { int rank, world; std::string dir; std::map<int, int> out, in; };
typedef LoopComm* ncclComm_t;

static std::string pipe_name(const std::string& dir, int from, int to) { return dir + "/p" + std::to_string(from) + "to" + std::to_string(to); }
static int open_pipe(const std::string& path, int flags)
{
    mkfifo(path.c_str(), 0600);      // whichever side comes first makes it
    return open(path.c_str(), flags);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { std::memset(id, 0, sizeof *id); std::strcpy(id->internal, "loopback"); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    const char* d = std::getenv("RTW_LOOPBACK_DIR");
    if (!d || std::strcmp(id.internal, "loopback") != 0) return 4;
    *comm = new LoopComm{ rank, nranks, d, {}, {} };
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return 0;
    for (auto& kv : c->out) close(kv.second);
    for (auto& kv : c->in) close(kv.second);
    delete c; return 0;
}
ncclResult_t ncclGroupStart() { return 0; }
ncclResult_t ncclGroupEnd() { return 0; }
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : "loopback transport error"; }

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t st)
{
    std::vector<char> h(count);
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    if (hipMemcpy(h.data(), buf, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!c->out.count(peer)) c->out[peer] = open_pipe(pipe_name(c->dir, c->rank, peer), O_WRONLY);
    const int fd = c->out[peer];
    if (fd < 0) return 2;
    for (size_t done = 0; done < count;) { const ssize_t n = write(fd, h.data() + done, count - done); if (n <= 0) return 2; done += (size_t)n; }
    return 0;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t st)
{
    std::vector<char> h(count);
    if (!c->in.count(peer)) c->in[peer] = open_pipe(pipe_name(c->dir, peer, c->rank), O_RDONLY);
    const int fd = c->in[peer];
    if (fd < 0) return 2;
    for (size_t done = 0; done < count;) { const ssize_t n = read(fd, h.data() + done, count - done); if (n <= 0) return 2; done += (size_t)n; }
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    if (hipMemcpy(buf, h.data(), count, hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}
}
