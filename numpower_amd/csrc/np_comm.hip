// np_comm_* — the one collective the path has (BASELINE config 5: contiguous batch slabs per GPU, one
// all-gather of the result slabs over xGMI), reachable from the C ABI so that a host which is not Python — the
// PHP extension the north star names — can shard a batched matmul across the GPUs of a node without torch.
//
// One process per GPU (the reference's own model: NDArray::setDevice picks the process's device,
// numpower.c:615-635; it has no multi-device code at all).  RCCL does the transport; it is loaded at
// np_comm_init() with dlopen — librccl.so is 0.5 GB and nothing else in this library needs it, so a process that
// never shards never maps it (and a process that already has torch's copy loaded gets that one: same SONAME).
// Collectives are enqueued on the library stream (np_get_stream), i.e. ordered with the kernels: a GEMM writing a
// slab followed by np_allgather of that slab needs no synchronisation in between.
//
// Rendezvous: rank 0 creates the ncclUniqueId and hands it to the other ranks
//   "tcp://host:port"  rank 0 listens on host:port and serves the 128 bytes to world - 1 connections (peers retry
//                      until it is up); nothing is left behind, a port is fresh by construction
//   any other string   a file path: rank 0 writes <path>.tmp and renames it to <path>, peers poll for <path>;
//                      the caller must pass a path that does not exist yet (rank 0 removes it at destroy)
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdio.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <rccl/rccl.h>

#include <string>

#include "np_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

struct Comm {
    Rccl api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    std::string file_to_remove;
    float *scratch = nullptr;   // world floats on the device (np_comm_max / barrier)
};

Comm g_comm;

int load_rccl(Rccl &r) {
    if (r.handle) return NP_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return np::fail(NP_ERR_DEVICE, "np_comm_init: cannot load librccl.so.1 (%s)", dlerror());
#define NP_SYM(field, name)                                                                        \
    r.field = (decltype(r.field))dlsym(r.handle, name);                                            \
    if (!r.field) return np::fail(NP_ERR_DEVICE, "np_comm_init: librccl has no %s", name)
    NP_SYM(GetUniqueId, "ncclGetUniqueId");
    NP_SYM(CommInitRank, "ncclCommInitRank");
    NP_SYM(CommDestroy, "ncclCommDestroy");
    NP_SYM(AllGather, "ncclAllGather");
    NP_SYM(AllReduce, "ncclAllReduce");
    NP_SYM(GetErrorString, "ncclGetErrorString");
#undef NP_SYM
    return NP_OK;
}

#define NP_RCCL_CHECK(expr)                                                                        \
    do {                                                                                           \
        ncclResult_t rc_ = (expr);                                                                 \
        if (rc_ != ncclSuccess)                                                                    \
            return np::fail(NP_ERR_DEVICE, "%s failed: %s", #expr, g_comm.api.GetErrorString(rc_)); \
    } while (0)

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

bool io_all(int fd, void *buf, size_t n, bool writing) {
    char *p = (char *)buf;
    while (n) {
        const ssize_t k = writing ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            return false;
        }
        p += k;
        n -= (size_t)k;
    }
    return true;
}

bool parse_tcp(const char *endpoint, std::string &host, int &port) {
    if (strncmp(endpoint, "tcp://", 6) != 0) return false;
    const char *hp = endpoint + 6, *colon = strrchr(hp, ':');
    if (!colon) return false;
    host.assign(hp, colon - hp);
    port = atoi(colon + 1);
    return port > 0 && port < 65536 && !host.empty();
}

int resolve(const std::string &host, int port, sockaddr_in &addr) {
    memset(&addr, 0, sizeof(addr));
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host.c_str(), &addr.sin_addr) == 1) return NP_OK;
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(host.c_str(), nullptr, &hints, &res) != 0 || !res)
        return np::fail(NP_ERR_INVALID, "np_comm_init: cannot resolve %s", host.c_str());
    addr.sin_addr = ((sockaddr_in *)res->ai_addr)->sin_addr;
    freeaddrinfo(res);
    return NP_OK;
}

// rank 0: serve `id` to world - 1 peers; peers: fetch it.  timeout_s bounds both sides.
int exchange_tcp(const std::string &host, int port, int rank, int world, ncclUniqueId &id, double timeout_s) {
    sockaddr_in addr;
    if (int rc = resolve(host, port, addr)) return rc;
    const double deadline = now_s() + timeout_s;
    if (rank == 0) {
        const int ls = socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) return np::fail(NP_ERR_DEVICE, "np_comm_init: socket(): %s", strerror(errno));
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (bind(ls, (sockaddr *)&addr, sizeof(addr)) != 0 || listen(ls, world) != 0) {
            const int e = errno;
            close(ls);
            return np::fail(NP_ERR_DEVICE, "np_comm_init: cannot listen on %s:%d: %s", host.c_str(), port, strerror(e));
        }
        timeval tv{1, 0};
        setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));   // accept() wakes up once a second
        for (int served = 0; served < world - 1;) {
            const int c = accept(ls, nullptr, nullptr);
            if (c < 0) {
                if (now_s() > deadline) {
                    close(ls);
                    return np::fail(NP_ERR_DEVICE, "np_comm_init: only %d of %d peers fetched the id within %.0f s", served,
                                    world - 1, timeout_s);
                }
                continue;
            }
            const bool ok = io_all(c, &id, sizeof(id), true);
            close(c);
            if (ok) ++served;
        }
        close(ls);
        return NP_OK;
    }
    for (;;) {
        const int c = socket(AF_INET, SOCK_STREAM, 0);
        if (c < 0) return np::fail(NP_ERR_DEVICE, "np_comm_init: socket(): %s", strerror(errno));
        if (connect(c, (sockaddr *)&addr, sizeof(addr)) == 0) {
            timeval tv{(time_t)timeout_s, 0};
            setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
            const bool ok = io_all(c, &id, sizeof(id), false);
            close(c);
            if (ok) return NP_OK;
        } else {
            close(c);
        }
        if (now_s() > deadline)
            return np::fail(NP_ERR_DEVICE, "np_comm_init: rank %d could not fetch the id from %s:%d within %.0f s", rank,
                            host.c_str(), port, timeout_s);
        usleep(20 * 1000);
    }
}

int exchange_file(const std::string &path, int rank, ncclUniqueId &id, double timeout_s) {
    if (rank == 0) {
        const std::string tmp = path + ".tmp";
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp) return np::fail(NP_ERR_INVALID, "np_comm_init: cannot write %s: %s", tmp.c_str(), strerror(errno));
        const bool ok = fwrite(&id, sizeof(id), 1, fp) == 1;
        fclose(fp);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0)
            return np::fail(NP_ERR_INVALID, "np_comm_init: cannot publish %s: %s", path.c_str(), strerror(errno));
        return NP_OK;
    }
    const double deadline = now_s() + timeout_s;
    for (;;) {
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && (size_t)st.st_size == sizeof(id)) {
            FILE *fp = fopen(path.c_str(), "rb");
            if (fp) {
                const bool ok = fread(&id, sizeof(id), 1, fp) == 1;
                fclose(fp);
                if (ok) return NP_OK;
            }
        }
        if (now_s() > deadline)
            return np::fail(NP_ERR_DEVICE, "np_comm_init: rank %d did not see %s within %.0f s", rank, path.c_str(), timeout_s);
        usleep(20 * 1000);
    }
}

}  // namespace

extern "C" {

int np_comm_init(int rank, int world, const char *endpoint) {
    if (world < 1 || rank < 0 || rank >= world) return np::fail(NP_ERR_INVALID, "np_comm_init: rank %d of %d", rank, world);
    if (!endpoint || !*endpoint) return np::fail(NP_ERR_INVALID, "np_comm_init: no rendezvous endpoint");
    if (g_comm.comm) return np::fail(NP_ERR_INVALID, "np_comm_init: a communicator already exists (np_comm_destroy first)");
    if (int rc = np::ensure_init()) return rc;
    if (int rc = load_rccl(g_comm.api)) return rc;
    ncclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0) NP_RCCL_CHECK(g_comm.api.GetUniqueId(&id));
    if (world > 1) {
        std::string host;
        int port = 0;
        const double timeout_s = 120.0;
        if (parse_tcp(endpoint, host, port)) {
            if (int rc = exchange_tcp(host, port, rank, world, id, timeout_s)) return rc;
        } else {
            if (int rc = exchange_file(endpoint, rank, id, timeout_s)) return rc;
            if (rank == 0) g_comm.file_to_remove = endpoint;
        }
    }
    NP_RCCL_CHECK(g_comm.api.CommInitRank(&g_comm.comm, world, id, rank));
    g_comm.rank = rank;
    g_comm.world = world;
    void *p = nullptr;
    if (int rc = np_malloc(&p, sizeof(float) * (size_t)(world + 1))) {
        (void)g_comm.api.CommDestroy(g_comm.comm);   // no half-made communicator: np_last_error() keeps the allocation message
        g_comm.comm = nullptr;
        return rc;
    }
    g_comm.scratch = (float *)p;
    return NP_OK;
}

// The rendezvous alone (no device, no RCCL): rank 0 hands `bytes128` to the peers, who receive it into theirs.
// Exists so that the multi-process hand-over can be tested on a machine without GPUs (tests/test_comm_rendezvous_cpu.py).
int np_comm_debug_exchange(int rank, int world, const char *endpoint, void *bytes128, double timeout_s) {
    if (world < 1 || rank < 0 || rank >= world || !endpoint || !*endpoint || !bytes128)
        return np::fail(NP_ERR_INVALID, "np_comm_debug_exchange: bad arguments");
    if (world == 1) return NP_OK;
    ncclUniqueId id;
    memcpy(&id, bytes128, sizeof(id));
    std::string host;
    int port = 0;
    int rc;
    if (parse_tcp(endpoint, host, port)) {
        rc = exchange_tcp(host, port, rank, world, id, timeout_s);
    } else {
        rc = exchange_file(endpoint, rank, id, timeout_s);
    }
    if (rc == NP_OK) memcpy(bytes128, &id, sizeof(id));
    return rc;
}

int np_comm_rank(void) { return g_comm.comm ? g_comm.rank : -1; }
int np_comm_world(void) { return g_comm.comm ? g_comm.world : 0; }

int np_allgather(const void *dev_send, void *dev_recv, size_t bytes_per_rank) {
    if (!g_comm.comm) return np::fail(NP_ERR_INVALID, "np_allgather: no communicator (np_comm_init first)");
    if (bytes_per_rank == 0) return NP_OK;
    if (!dev_send || !dev_recv) return np::fail(NP_ERR_INVALID, "np_allgather: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // bytes travel as ncclChar: any slab size, no alignment demand beyond the buffers' own
    NP_RCCL_CHECK(g_comm.api.AllGather(dev_send, dev_recv, bytes_per_rank, ncclChar, g_comm.comm, np::stream()));
    return NP_OK;
}

int np_comm_max(float value, float *host_max) {
    if (!g_comm.comm) return np::fail(NP_ERR_INVALID, "np_comm_max: no communicator (np_comm_init first)");
    if (!host_max) return np::fail(NP_ERR_INVALID, "np_comm_max: null output");
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipMemcpyAsync(g_comm.scratch, &value, sizeof(float), hipMemcpyHostToDevice, np::stream()));
    NP_RCCL_CHECK(g_comm.api.AllReduce(g_comm.scratch, g_comm.scratch + 1, 1, ncclFloat, ncclMax, g_comm.comm, np::stream()));
    NP_HIP_CHECK(hipMemcpyAsync(host_max, g_comm.scratch + 1, sizeof(float), hipMemcpyDeviceToHost, np::stream()));
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    return NP_OK;
}

int np_comm_barrier(void) {
    float ignored = 0.0f;
    return np_comm_max(0.0f, &ignored);   // returns once every rank's stream has reached this point
}

int np_comm_destroy(void) {
    if (!g_comm.comm) return NP_OK;
    (void)hipStreamSynchronize(np::stream());
    const ncclResult_t rc = g_comm.api.CommDestroy(g_comm.comm);
    g_comm.comm = nullptr;
    if (g_comm.scratch) np_free(g_comm.scratch);
    g_comm.scratch = nullptr;
    if (!g_comm.file_to_remove.empty()) {
        unlink(g_comm.file_to_remove.c_str());
        g_comm.file_to_remove.clear();
    }
    if (rc != ncclSuccess) return np::fail(NP_ERR_DEVICE, "ncclCommDestroy failed: %s", g_comm.api.GetErrorString(rc));
    return NP_OK;
}

}  // extern "C"
