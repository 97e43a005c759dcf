// np_comm_* — the one collective the path has (BASELINE config 5: contiguous batch slabs per GPU, one
// all-gather of the result slabs over xGMI), reachable from the C ABI so that a host which is not Python — the
// PHP extension the north star names — can shard a batched matmul across the GPUs of a node without torch.
//
// One process per GPU (the reference's own model: NDArray::setDevice picks the process's device,
// numpower.c:615-635; it has no multi-device code at all).  RCCL does the transport; it is loaded at
// np_comm_init() with dlopen — librccl.so is 0.5 GB and nothing else in this library needs it, so a process that
// never shards never maps it (and a process that already has torch's copy loaded gets that one: same SONAME).
// The handful of RCCL declarations used are restated below, so building libnp_hip.so needs no RCCL header either.
//
// Two streams.  np_allgather() enqueues on the library stream (np_get_stream), i.e. strictly behind the kernels:
// a GEMM writing a slab followed by np_allgather of that slab needs no synchronisation in between — and gets no
// overlap.  The *_async entry points and np_sgemm_strided_batched_allgather() put the collective on the
// communicator's OWN stream behind an event recorded on the library stream: the GEMM of piece c+1 runs while piece c
// travels.  At 8 GPUs config 5 moves 1.75 GiB into every GPU (>= 1.75 ms even at the 7 x 153 GB/s the xGMI mesh
// offers) against ~1 ms of GEMM per rank, so hiding the shorter leg behind the longer one is the one multi-GPU
// optimisation the cost model says matters.  np_comm_wait() orders the library stream behind
// whatever the communication stream has been given (a device-side wait; the host does not block).
//
// How the two streams are ordered.  hipStreamWaitEvent across two streams costs the LAUNCHING THREAD 60-80 us per call
// on this platform (profiles/r03/chunk_overhead.log: 8 pieces 1.05 -> 1.51 ms per slab, GPU idle in between) — more
// than the transfer it is supposed to hide.  So the order is kept on the device instead: a one-lane kernel on the
// producing stream publishes a sequence number (flag_set_kernel), a one-lane kernel on the consuming stream spins on it
// (flag_wait_kernel: s_sleep between device-scope acquire loads).  Which waits may give up: a wait whose producer is THIS
// GPU's own work (the communication stream waiting for the library stream's GEMM) is bounded — it gives up after a
// time-out and raises the process's device-error word (np_internal.h), which np_sync, np_memcpy_d2h, the host-result
// calls and every np_comm_* entry point turn into NP_ERR_DEVICE.  The opposite direction — the library stream waiting
// for transfers to be delivered — depends on OTHER ranks (skew, RCCL's lazy connect, a peer that is minutes late): its
// bound is generous and configurable (np_comm_set_wait_limit, default ten minutes, 0 = never — like the RCCL kernel it
// waits for), and giving up raises the same error word: a result that has not been gathered is never handed to the
// caller as if it had, and a dead peer ends as NP_ERR_DEVICE instead of a hang.  The host can release such a wait early
// (np_comm_destroy raises the abort word every wait polls, and aborts the communicator if RCCL's own kernel is stuck).  The sharded GEMM goes one step further: ONE launch computes the whole
// slab and its workgroups count finished tiles per piece (GemmArgs::progress, np_sgemm.hip); the wait kernel in front
// of piece c's transfer releases it when the count is complete — no launch per piece, no host in the loop.  Both
// need the two streams to sit on different hardware queues (they do: the communication stream is created at high
// priority); np_comm_init proves it with a self-test and falls back to HIP events if the flag does not come through
// (np_comm_set_variant(1) forces the event form, for A/B).
//
// What has and has not run on hardware.  Everything here has run on ONE GPU (world 1, plus RCCL loopback send/recv).  The
// progress-reporting single launch relies on memory-side (sc1) stores of C being visible to the RCCL kernel that the
// wait kernel releases, without a kernel boundary in between; that is proven on one device only, so with world > 1 the
// default is the form with a kernel boundary behind every piece (one GEMM launch per piece + flag_set_kernel, variant
// 2's behaviour).  np_comm_set_variant(3) asks for the single launch at any world size (bench.py times it as its own
// leg, with a parity check behind it, so the first multi-GPU run validates or refutes it).
//
// Rendezvous: rank 0 creates the ncclUniqueId and hands it to the other ranks
//   "tcp://host:port"  rank 0 listens on host:port; a peer connects, says who it is (magic + rank), and gets the
//                      128 bytes back; a connection that does not introduce itself as a rank not yet served is
//                      dropped and does not count (peers retry until rank 0 is up); nothing is left behind
//   any other string   a file path: rank 0 removes a stale <path>, writes <path>.tmp and renames it to <path>, peers
//                      poll for <path>; rank 0 removes it at destroy and on every failure after publishing
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

#include <string>
#include <vector>

#include "np_internal.h"

namespace {

// ---- the slice of RCCL's C API this file uses (rccl.h of ROCm 7.2 = NCCL 2.27 API; values are ABI) ----
typedef struct { char internal[128]; } ncclUniqueId;
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr int ncclChar = 0, ncclFloat = 7;   // ncclDataType_t
constexpr int ncclMax = 2;                   // ncclRedOp_t

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;   // optional
};

struct Comm {
    Rccl api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    int device = -1;            // the communicator, its stream and its flags belong to the device current at np_comm_init
    std::string file_to_remove;
    float *scratch = nullptr;   // world floats on the device (np_comm_max / barrier)
    // the communication stream and its events (created at np_comm_init on the communicator's device)
    hipStream_t stream = nullptr;
    static constexpr int kEvents = 32;
    hipEvent_t produced[kEvents] = {};   // ring: "the library stream has produced this piece"
    unsigned next_event = 0;
    hipEvent_t drained = nullptr;        // "the communication stream has delivered everything given to it so far"
    bool pending = false;                // something was enqueued on the communication stream since the last np_comm_wait
    // device-side ordering (see the header comment)
    static constexpr int kMaxPieces = 64;
    unsigned *flags = nullptr;           // device: [0] produced sequence, [1] drained sequence, [8 .. 8 + kMaxPieces) tile counters (zero between calls)
    unsigned *host_error = nullptr;      // pinned, device-visible, 2 words: the SELF-TEST's error / abort words (real waits use np::device_error_word())
    unsigned produced_seq = 0, drained_seq = 0;
    bool use_flags = false;              // false: HIP events (np_comm_set_variant(1), or the self-test failed)
    hipStream_t flags_ok_for = nullptr;  // the library stream the self-test passed for
    // testing (np_comm_debug_loopback): every gathered piece is ALSO sent from this rank to itself into this scratch
    // buffer — real RCCL p2p traffic on the communication stream of a box that has no peer
    char *loopback = nullptr;
    size_t loopback_bytes = 0;
};

// np_comm_set_variant: 0 = device-side flags where they work; ONE progress-reporting GEMM launch on a one-rank communicator,
// one launch per piece when there are peers (see the header), 1 = HIP events only, 2 = flags, one GEMM launch per piece,
// 3 = flags, one progress-reporting launch whatever the world size
int g_sync_variant = 0;

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
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.handle, "ncclCommAbort");   // optional: only a stuck communicator needs it
    NP_SYM(AllGather, "ncclAllGather");
    NP_SYM(AllReduce, "ncclAllReduce");
    NP_SYM(Send, "ncclSend");
    NP_SYM(Recv, "ncclRecv");
    NP_SYM(GroupStart, "ncclGroupStart");
    NP_SYM(GroupEnd, "ncclGroupEnd");
    NP_SYM(GetErrorString, "ncclGetErrorString");
#undef NP_SYM
    r.GetVersion = (decltype(r.GetVersion))dlsym(r.handle, "ncclGetVersion");
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

// What a peer says before it is handed the id: rank 0 must not count a stray local connection (a port scanner, a
// health check, a second job reusing the port) as a served rank.
struct Hello {
    uint32_t magic;
    int32_t rank, world;
};
constexpr uint32_t kHelloMagic = 0x4e50434dU;   // "NPCM"

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
        std::vector<char> served_rank((size_t)world, 0);
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
            timeval ctv{2, 0};   // a connection that says nothing is dropped after 2 s
            setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &ctv, sizeof(ctv));
            Hello h{};
            const bool known = io_all(c, &h, sizeof(h), false) && h.magic == kHelloMagic && h.world == world &&
                               h.rank > 0 && h.rank < world && !served_rank[(size_t)h.rank];
            const bool ok = known && io_all(c, &id, sizeof(id), true);
            close(c);
            if (ok) {
                served_rank[(size_t)h.rank] = 1;
                ++served;
            }
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
            Hello h{kHelloMagic, rank, world};
            const bool ok = io_all(c, &h, sizeof(h), true) && io_all(c, &id, sizeof(id), false);
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
        // a file left by a run that died after publishing would hand its peers a dead id: rank 0 owns the path
        (void)unlink(path.c_str());
        FILE *fp = fopen(tmp.c_str(), "wb");
        if (!fp) return np::fail(NP_ERR_INVALID, "np_comm_init: cannot write %s: %s", tmp.c_str(), strerror(errno));
        const bool ok = fwrite(&id, sizeof(id), 1, fp) == 1;
        fclose(fp);
        if (!ok || rename(tmp.c_str(), path.c_str()) != 0) {
            const int e = errno;
            (void)unlink(tmp.c_str());
            return np::fail(NP_ERR_INVALID, "np_comm_init: cannot publish %s: %s", path.c_str(), strerror(e));
        }
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

// ---- the communication stream ----

constexpr unsigned long long kWaitTimeoutTicks = 60ull * 100000000ull;   // 60 s of the 100 MHz wall clock: waits for THIS GPU's own work
// Waits for transfers depend on other ranks: generous, configurable (np_comm_set_wait_limit), 0 = never give up — like the RCCL
// kernel they wait for.  A peer that died must end as NP_ERR_DEVICE at the next sync point, not as a hang nobody can leave
// (ADVICE r04): the default is ten minutes.
unsigned long long g_peer_wait_ticks = 600ull * 100000000ull;
#define kWaitForPeers __atomic_load_n(&g_peer_wait_ticks, __ATOMIC_RELAXED)   /* read at launch time, written by np_comm_set_wait_limit from any thread */
constexpr unsigned long long kSelfTestTicks = 20000000ull;               // 200 ms (only ever spent when the test FAILS)

__global__ void flag_set_kernel(unsigned *flag, unsigned value) {
    // everything earlier on this stream is complete and visible (kernel boundary): a memory-side store publishes it
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Spins until *flag has reached `target` (sequence numbers: compared as a signed difference, so wrap-around is fine;
// tile counters: plain >=).  One lane; s_sleep keeps it off the issue ports of the CU it sits on.  timeout_ticks != 0:
// gives up after that long, ORs error_bit into error_word[0] (reported by the host, np::check_device_error) and lets the
// queue drain.  timeout_ticks == 0: never gives up by itself.  Either way it ends when the host raises error_word[1] (abort).
__global__ void flag_wait_kernel(const unsigned *flag, unsigned target, unsigned long long timeout_ticks, unsigned *error_word,
                                 unsigned error_bit) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 0;; ++spins) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // memory-side; the next kernel on this stream starts with an acquire
        if ((int)(v - target) >= 0) return;
        __builtin_amdgcn_s_sleep(16);
        if (timeout_ticks && (spins & 63u) == 63u && wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_fetch_or(error_word, error_bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;   // give up rather than hang the queue: the host reports it (np_sync, np_memcpy_d2h, np_comm_*)
        }
        if ((spins & 4095u) == 4095u &&   // one read over the host link every few milliseconds
            __hip_atomic_load(error_word + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
            __hip_atomic_fetch_or(error_word, error_bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;   // the host asked every wait to end (np_comm_destroy): what it guarded is incomplete, and reported as such
        }
    }
}

// The last act of a pipelined call on the communication stream: the tile counters go back to zero (the next call's GEMM
// counts from there) and the "drained" sequence number is published — one kernel, one hop, instead of a memset + a set.
__global__ void finish_kernel(unsigned *counters, int n, unsigned *flag, unsigned value) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(counters + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int stream_follows(hipStream_t consumer, hipStream_t producer, unsigned *flag, unsigned &seq, hipEvent_t event,
                   unsigned long long timeout_ticks) {
    Comm &c = g_comm;
    if (c.use_flags) {
        ++seq;
        flag_set_kernel<<<1, 1, 0, producer>>>(flag, seq);
        NP_LAUNCH_CHECK("flag_set_kernel");
        flag_wait_kernel<<<1, 1, 0, consumer>>>(flag, seq, timeout_ticks, np::device_error_word(), np::kErrCommWait);
        NP_LAUNCH_CHECK("flag_wait_kernel");
        return NP_OK;
    }
    NP_HIP_CHECK(hipEventRecord(event, producer));
    NP_HIP_CHECK(hipStreamWaitEvent(consumer, event, 0));
    return NP_OK;
}

// Do device-side flags work between the communication stream and `compute`?  The wait goes in FIRST: were the two
// streams multiplexed onto one hardware queue, the set kernel would queue up behind the spinning wait and never run —
// the wait then gives up after 200 ms, raises the error word, and this communicator uses HIP events instead.
int flags_self_test(hipStream_t compute) {
    Comm &c = g_comm;
    c.use_flags = false;
    c.flags_ok_for = compute;
    if (g_sync_variant == 1 || !c.flags || !c.host_error || !np::device_error_word()) return NP_OK;
    // both kernels once with nothing to wait for: their first launch loads code on the host side, and that must not eat
    // into the budget of the wait that is about to spin on the device
    flag_set_kernel<<<1, 1, 0, compute>>>(c.flags + 2, 1u);
    NP_LAUNCH_CHECK("flag_set_kernel");
    flag_wait_kernel<<<1, 1, 0, c.stream>>>(c.flags + 3, 0u, kSelfTestTicks, c.host_error, 1u);
    NP_LAUNCH_CHECK("flag_wait_kernel");
    NP_HIP_CHECK(hipStreamSynchronize(compute));
    NP_HIP_CHECK(hipStreamSynchronize(c.stream));
    *(volatile unsigned *)c.host_error = 0;
    const unsigned seq = ++c.produced_seq;
    flag_wait_kernel<<<1, 1, 0, c.stream>>>(c.flags, seq, kSelfTestTicks, c.host_error, 1u);
    NP_LAUNCH_CHECK("flag_wait_kernel");
    flag_set_kernel<<<1, 1, 0, compute>>>(c.flags, seq);
    NP_LAUNCH_CHECK("flag_set_kernel");
    NP_HIP_CHECK(hipStreamSynchronize(c.stream));
    NP_HIP_CHECK(hipStreamSynchronize(compute));
    c.use_flags = *(volatile unsigned *)c.host_error == 0;
    *(volatile unsigned *)c.host_error = 0;
    return NP_OK;
}

void destroy_stream_objects() {
    for (hipEvent_t &e : g_comm.produced) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (g_comm.drained) (void)hipEventDestroy(g_comm.drained);
    g_comm.drained = nullptr;
    if (g_comm.stream) (void)hipStreamDestroy(g_comm.stream);
    g_comm.stream = nullptr;
    g_comm.pending = false;
    if (g_comm.flags) (void)hipFree(g_comm.flags);
    g_comm.flags = nullptr;
    if (g_comm.host_error) (void)hipHostFree(g_comm.host_error);
    g_comm.host_error = nullptr;
    g_comm.use_flags = false;
    g_comm.flags_ok_for = nullptr;
    g_comm.loopback = nullptr;
    g_comm.loopback_bytes = 0;
}

int create_stream_objects() {
    // highest priority: while a GEMM fills every CU, the collective's few workgroups must still get scheduled promptly
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
        (void)hipGetLastError();
        least = greatest = 0;
    }
    NP_HIP_CHECK(hipStreamCreateWithPriority(&g_comm.stream, hipStreamNonBlocking, greatest));
    for (hipEvent_t &e : g_comm.produced) NP_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    NP_HIP_CHECK(hipEventCreateWithFlags(&g_comm.drained, hipEventDisableTiming));
    const size_t flag_bytes = sizeof(unsigned) * (8 + Comm::kMaxPieces);
    NP_HIP_CHECK(hipMalloc((void **)&g_comm.flags, flag_bytes));
    NP_HIP_CHECK(hipMemset(g_comm.flags, 0, flag_bytes));
    NP_HIP_CHECK(hipHostMalloc((void **)&g_comm.host_error, 2 * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable));
    g_comm.host_error[0] = g_comm.host_error[1] = 0;
    g_comm.produced_seq = g_comm.drained_seq = 0;
    return flags_self_test(np::stream());
}

// The communication stream picks up everything the library stream has been given so far.
int comm_stream_follows_compute() {
    Comm &c = g_comm;
    // the producer is this GPU's own stream: a bounded wait
    return stream_follows(c.stream, np::stream(), c.flags + 0, c.produced_seq, c.produced[c.next_event++ % Comm::kEvents],
                          kWaitTimeoutTicks);
}

// ---- the addressing of the exchange, as pure functions (shared by the real path and by np_comm_debug_plan, which lets a
// test run the arithmetic for any rank / world without a device) ----

// step s of the grouped exchange: rank r sends to r + s while it receives from r - s — every step is a perfect matching
// of the fully connected mesh, every pair of GPUs meets exactly once per direction
inline void p2p_peers(int rank, int world, int step, int *to, int *from) {
    *to = (rank + step) % world;
    *from = (rank - step + world) % world;
}

// piece [lo, lo + count) of this rank's slab inside the replicated result of world * slab items of item_bytes each:
// where it is sent FROM, the base every rank's copy of that piece is received relative to, and how far apart the ranks'
// copies lie (one slab)
inline void piece_addresses(int rank, size_t slab, size_t item_bytes, size_t lo, size_t *send_off, size_t *recv_base_off,
                            size_t *recv_stride) {
    *send_off = ((size_t)rank * slab + lo) * item_bytes;
    *recv_base_off = lo * item_bytes;
    *recv_stride = slab * item_bytes;
}

// recv_base + r * recv_stride <- rank r's `bytes` at send, for every r, on stream s.  Contiguous destinations
// (recv_stride == bytes) are ONE ncclAllGather unless p2p is asked for; anything else is one grouped exchange of
// ncclSend / ncclRecv pairs — every peer's piece travels over that peer's own xGMI link straight into place, no
// staging buffer, no scatter copy.  The own piece is copied only when it is not already in place.
int gather_on(hipStream_t s, const void *send, void *recv_base, size_t bytes, size_t recv_stride, bool p2p) {
    Comm &c = g_comm;
    char *base = (char *)recv_base;
    if (recv_stride == bytes && !p2p) {
        // bytes travel as ncclChar: any slab size, no alignment demand beyond the buffers' own
        NP_RCCL_CHECK(c.api.AllGather(send, recv_base, bytes, ncclChar, c.comm, s));
        return NP_OK;
    }
    char *own = base + (size_t)c.rank * recv_stride;
    if ((const void *)own != send) NP_HIP_CHECK(hipMemcpyAsync(own, send, bytes, hipMemcpyDeviceToDevice, s));
    if (c.loopback && bytes <= c.loopback_bytes) {
        NP_RCCL_CHECK(c.api.GroupStart());
        ncclResult_t rc = c.api.Send(send, bytes, ncclChar, c.rank, c.comm, s);
        if (rc == ncclSuccess) rc = c.api.Recv(c.loopback, bytes, ncclChar, c.rank, c.comm, s);
        const ncclResult_t rc2 = c.api.GroupEnd();
        if (rc != ncclSuccess || rc2 != ncclSuccess)
            return np::fail(NP_ERR_DEVICE, "loopback ncclSend/ncclRecv failed: %s", c.api.GetErrorString(rc != ncclSuccess ? rc : rc2));
    }
    if (c.world == 1) return NP_OK;
    NP_RCCL_CHECK(c.api.GroupStart());
    for (int step = 1; step < c.world; ++step) {
        int to = 0, from = 0;
        p2p_peers(c.rank, c.world, step, &to, &from);
        ncclResult_t rc = c.api.Send(send, bytes, ncclChar, to, c.comm, s);
        if (rc == ncclSuccess) rc = c.api.Recv(base + (size_t)from * recv_stride, bytes, ncclChar, from, c.comm, s);
        if (rc != ncclSuccess) {
            (void)c.api.GroupEnd();
            return np::fail(NP_ERR_DEVICE, "ncclSend/ncclRecv failed: %s", c.api.GetErrorString(rc));
        }
    }
    NP_RCCL_CHECK(c.api.GroupEnd());
    return NP_OK;
}

int need_comm(const char *who) {
    if (!g_comm.comm) return np::fail(NP_ERR_INVALID, "%s: no communicator (np_comm_init first)", who);
    if (int rc = np::ensure_init()) return rc;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != g_comm.device)
        return np::fail(NP_ERR_INVALID, "%s: the communicator belongs to device %d but the library is on device %d "
                                        "(np_set_device back, or np_comm_destroy and np_comm_init again)", who, g_comm.device, cur);
    if (int rc = np::check_device_error(who)) return rc;   // an earlier device-side wait gave up: say so before anything else is built on it
    // the caller may have moved the library onto another stream (np_set_stream): prove the flags again for that one
    if (g_comm.flags_ok_for != np::stream()) return flags_self_test(np::stream());
    return NP_OK;
}

// the caller's patience with transfers (np_comm_set_wait_limit) also bounds the grace periods of destroy / acknowledge: someone
// who declared transfers dead after 2 s does not want np_comm_destroy to hope for 30
double grace_seconds(double most) {
    const double limit = (double)kWaitForPeers / 1e8;      // 0 = never give up: no opinion
    return limit > 0.0 && limit < most ? (limit < 1.0 ? 1.0 : limit) : most;
}

}  // namespace

bool np::comm_transfers_stuck(int device, double grace_s) {
    if (!g_comm.comm || !g_comm.stream || g_comm.device != device) return false;
    const double give_up = now_s() + grace_seconds(grace_s);
    while (hipStreamQuery(g_comm.stream) == hipErrorNotReady) {
        if (now_s() > give_up) {
            (void)hipGetLastError();
            return true;
        }
        usleep(200);
    }
    (void)hipGetLastError();
    return false;
}

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
    std::string published;   // rank 0, file mode: the path peers poll; removed on every failure from here on
    if (world > 1) {
        std::string host;
        int port = 0;
        const double timeout_s = 120.0;
        if (parse_tcp(endpoint, host, port)) {
            if (int rc = exchange_tcp(host, port, rank, world, id, timeout_s)) return rc;
        } else {
            if (int rc = exchange_file(endpoint, rank, id, timeout_s)) return rc;
            if (rank == 0) published = endpoint;
        }
    }
    // from here on a failure must leave nothing behind: no published id (a later init with the same path would hand
    // its peers a dead one and they would hang in ncclCommInitRank), no half-made communicator, no stream
    const auto undo = [&](int rc) {
        if (g_comm.comm) (void)g_comm.api.CommDestroy(g_comm.comm);
        g_comm.comm = nullptr;
        destroy_stream_objects();
        if (!published.empty()) (void)unlink(published.c_str());
        return rc;
    };
    {
        const ncclResult_t rc = g_comm.api.CommInitRank(&g_comm.comm, world, id, rank);
        if (rc != ncclSuccess) {
            g_comm.comm = nullptr;
            return undo(np::fail(NP_ERR_DEVICE, "ncclCommInitRank failed: %s", g_comm.api.GetErrorString(rc)));
        }
    }
    g_comm.rank = rank;
    g_comm.world = world;
    if (hipGetDevice(&g_comm.device) != hipSuccess) g_comm.device = 0;
    if (int rc = create_stream_objects()) return undo(rc);
    void *p = nullptr;
    if (int rc = np_malloc(&p, sizeof(float) * (size_t)(world + 1))) return undo(rc);   // np_last_error() keeps the allocation message
    g_comm.scratch = (float *)p;
    g_comm.file_to_remove = published;
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
    if (int rc = need_comm("np_allgather")) return rc;
    if (bytes_per_rank == 0) return NP_OK;
    if (!dev_send || !dev_recv) return np::fail(NP_ERR_INVALID, "np_allgather: null pointer");
    return gather_on(np::stream(), dev_send, dev_recv, bytes_per_rank, bytes_per_rank, false);
}

int np_allgather_async(const void *dev_send, void *dev_recv_base, size_t bytes, size_t recv_stride_bytes, int mode) {
    if (int rc = need_comm("np_allgather_async")) return rc;
    if (mode != NP_GATHER_AUTO && mode != NP_GATHER_COLLECTIVE && mode != NP_GATHER_P2P)
        return np::fail(NP_ERR_INVALID, "np_allgather_async: unknown mode %d", mode);
    if (bytes == 0) return NP_OK;
    if (!dev_send || !dev_recv_base) return np::fail(NP_ERR_INVALID, "np_allgather_async: null pointer");
    if (recv_stride_bytes < bytes)
        return np::fail(NP_ERR_INVALID, "np_allgather_async: destinations %zu bytes apart would overlap (%zu bytes each)",
                        recv_stride_bytes, bytes);
    if (mode == NP_GATHER_COLLECTIVE && recv_stride_bytes != bytes)
        return np::fail(NP_ERR_INVALID, "np_allgather_async: NP_GATHER_COLLECTIVE needs contiguous destinations (stride == bytes)");
    if (int rc = comm_stream_follows_compute()) return rc;
    g_comm.pending = true;
    return gather_on(g_comm.stream, dev_send, dev_recv_base, bytes, recv_stride_bytes, mode == NP_GATHER_P2P);
}

int np_comm_wait(void) {
    if (int rc = need_comm("np_comm_wait")) return rc;
    if (!g_comm.pending) return NP_OK;
    // the producer is the communication stream, i.e. transfers other ranks take part in: never given up on
    if (int rc = stream_follows(np::stream(), g_comm.stream, g_comm.flags + 1, g_comm.drained_seq, g_comm.drained, kWaitForPeers))
        return rc;
    g_comm.pending = false;
    return NP_OK;
}

int np_comm_set_variant(int variant) {
    if (variant < 0 || variant > 3)
        return np::fail(NP_ERR_INVALID, "np_comm_set_variant: 0 = device-side flags; one progress-reporting GEMM launch without peers, one "
                                        "launch per piece with peers (default), 1 = HIP events, one GEMM launch per piece, 2 = device-side "
                                        "flags, one GEMM launch per piece, 3 = device-side flags, one progress-reporting launch at any world size");
    g_sync_variant = variant;
    if (g_comm.comm) return flags_self_test(np::stream());
    return NP_OK;
}

// testing: from now on every piece gathered through the p2p path is also sent from this rank to itself into
// dev_scratch (pieces larger than `bytes` are not) — the only RCCL transfer a one-GPU box can run next to the GEMM.
// (nullptr, 0) switches it off.
int np_comm_debug_loopback(void *dev_scratch, size_t bytes) {
    if (int rc = need_comm("np_comm_debug_loopback")) return rc;
    g_comm.loopback = (char *)dev_scratch;
    g_comm.loopback_bytes = dev_scratch ? bytes : 0;
    return NP_OK;
}

int np_comm_sync_mode(void) { return g_comm.comm ? (g_comm.use_flags ? 1 : 0) : -1; }

// Piece c of a slab cut into `chunks` pieces: as equal as they come, the first slab % chunks pieces hold one item
// more.  Pure arithmetic (no device, no communicator): every rank cuts its slab the same way, which is what makes
// piece c of rank r land at r * slab + lo of the replicated result.
int np_comm_piece(size_t slab, int chunks, int c, size_t *host_lo, size_t *host_count) {
    if (!host_lo || !host_count) return np::fail(NP_ERR_INVALID, "np_comm_piece: null output");
    if (chunks < 1 || c < 0 || c >= chunks) return np::fail(NP_ERR_INVALID, "np_comm_piece: piece %d of %d", c, chunks);
    const size_t base = slab / (size_t)chunks, extra = slab % (size_t)chunks, i = (size_t)c;
    *host_lo = i * base + (i < extra ? i : extra);
    *host_count = base + (i < extra ? 1 : 0);
    return NP_OK;
}

void *np_comm_stream(void) { return g_comm.comm ? (void *)g_comm.stream : nullptr; }

// The sharded batched matmul of BASELINE config 5 as ONE call: this rank's slab of `slab` products, written in
// place into its window of the replicated result, in `chunks` pieces — piece c's gather is given to the
// communication stream the moment piece c's GEMM has been enqueued, and travels while piece c + 1 is computed.
// ---- how many pieces (chunks = 0): a step model of the pipeline, host arithmetic ----
// Piece c's GEMM ends at (c + 1) T_g / k.  Transfers are serialised on the communication stream; each takes T_x / k (every
// peer's share of the piece arrives over that peer's own xGMI link, all links at once) plus a fixed issue cost, starts when
// its GEMM piece and the previous transfer are done, and — what one GPU could measure (profiles/r04/comm_contention.log) —
// starts LATE by one round of the GEMM's workgroups when the GEMM still holds the CUs.  The step ends with the last
// transfer (or the GEMM).  The smallest k within 2 % of the best modelled step is taken: fewer, larger pieces when it is a
// tie.  Constants: 130 TFLOP/s sustained by the slab GEMM, 153 GB/s per link (MI355X_MICROARCH.md), 256 x 128 tiles, two
// resident workgroups per CU, 10 us per piece.  Unmeasured on more than one GPU; the model is what is pinned
// (tests/test_parallel_cpu.py), not a claim about a node.
constexpr int kModelPieces[5] = {1, 2, 4, 8, 16};
int model_pieces(int world, size_t slab, size_t M, size_t N, size_t K, int cus, double *ms5) {
    const double t_g = 2.0 * (double)slab * (double)M * (double)N * (double)K / 130e12 * 1e3;
    const double t_x = world > 1 ? (double)slab * (double)M * (double)N * 4.0 / 153e9 * 1e3 : 0.0;
    const double tiles = (double)slab * (double)((M + 255) / 256) * (double)((N + 127) / 128);
    const double rounds = tiles / (2.0 * (cus > 0 ? cus : 256));
    const double delay = t_g / (rounds < 1.0 ? 1.0 : rounds);
    const double issue = 0.01;
    double best = 1e300;
    double ms[5];
    for (int i = 0; i < 5; ++i) {
        const int k = kModelPieces[i];
        if ((size_t)k > slab && i > 0) {
            ms[i] = 1e300;
            continue;
        }
        double end_prev = 0.0;
        for (int c = 0; c < k; ++c) {
            double start = (c + 1) * t_g / k;
            if (end_prev > start) start = end_prev;
            if (k > 1 && start < t_g) start += delay;      // the communication kernels wait for CUs while the GEMM runs
            end_prev = start + t_x / k + issue;
        }
        ms[i] = end_prev > t_g ? end_prev : t_g;
        if (ms[i] < best) best = ms[i];
    }
    int pick = 1;
    for (int i = 0; i < 5; ++i) {
        if (ms5) ms5[i] = ms[i];
    }
    for (int i = 0; i < 5; ++i)
        if (ms[i] <= 1.02 * best) {
            pick = kModelPieces[i];
            break;
        }
    return world > 1 ? pick : 1;                            // nothing travels on a one-rank communicator
}

int np_sgemm_strided_batched_allgather(size_t slab, size_t M, size_t N, size_t K, const float *A, size_t stride_a,
                                       const float *B, size_t stride_b, float *C_full, int chunks, int mode) {
    if (int rc = need_comm("np_sgemm_strided_batched_allgather")) return rc;
    if (mode != NP_GATHER_AUTO && mode != NP_GATHER_COLLECTIVE && mode != NP_GATHER_P2P)
        return np::fail(NP_ERR_INVALID, "np_sgemm_strided_batched_allgather: unknown mode %d", mode);
    if (chunks < 0) return np::fail(NP_ERR_INVALID, "np_sgemm_strided_batched_allgather: chunks = %d", chunks);
    if (slab == 0 || M == 0 || N == 0) return NP_OK;
    if (chunks == 0) {   // the library's own choice: the step model above (one piece under NP_GATHER_COLLECTIVE, which moves whole slabs)
        chunks = mode == NP_GATHER_COLLECTIVE ? 1 : model_pieces(g_comm.world, slab, M, N, K, np::num_cus(), nullptr);
    }
    if (!A || !B || !C_full) return np::fail(NP_ERR_INVALID, "np_sgemm_strided_batched_allgather: null pointer");
    if ((size_t)chunks > slab) chunks = (int)slab;
    if (mode == NP_GATHER_COLLECTIVE && chunks > 1)
        return np::fail(NP_ERR_INVALID, "np_sgemm_strided_batched_allgather: NP_GATHER_COLLECTIVE moves whole slabs "
                                        "(chunks = 1); pieces of a slab are not contiguous across ranks");
    const size_t mat = M * N, slab_elems = slab * mat;
    float *mine = C_full + (size_t)g_comm.rank * slab_elems;
    Comm &cm = g_comm;
    const bool single_launch = g_sync_variant == 3 || (g_sync_variant == 0 && cm.world == 1);   // (see the header: with peers only on request)
    if (cm.use_flags && single_launch && chunks <= Comm::kMaxPieces) {
        // ONE launch for the whole slab; its workgroups count finished tiles per piece in cm.flags[8 + c] (zero here:
        // the previous call's last act on the communication stream was to clear them, and np_comm_wait ordered this
        // call's GEMM behind that)
        unsigned tiles = 0;
        unsigned *counters = cm.flags + 8;
        if (int rc = np::sgemm_batched_with_progress(slab, M, N, K, A, stride_a, B, stride_b, mine, mat, counters, chunks, &tiles))
            return rc;
        if (tiles) {
            cm.pending = true;
            int failed = NP_OK;
            for (int c = 0; c < chunks && failed == NP_OK; ++c) {
                size_t lo = 0, count = 0;
                if ((failed = np_comm_piece(slab, chunks, c, &lo, &count)) != NP_OK) break;
                flag_wait_kernel<<<1, 1, 0, cm.stream>>>(counters + c, (unsigned)(count * tiles), kWaitTimeoutTicks,   // producer: this GPU's GEMM
                                                         np::device_error_word(), np::kErrCommWait);
                if (hipGetLastError() != hipSuccess) {
                    failed = np::fail(NP_ERR_DEVICE, "launch of flag_wait_kernel failed");
                    break;
                }
                size_t send_off = 0, recv_off = 0, stride = 0;
                piece_addresses(cm.rank, slab, mat * sizeof(float), lo, &send_off, &recv_off, &stride);
                failed = gather_on(cm.stream, (const char *)C_full + send_off, (char *)C_full + recv_off, count * mat * sizeof(float),
                                   stride, chunks == 1 ? mode == NP_GATHER_P2P : true);
            }
            if (failed != NP_OK) {
                // the GEMM is counting tiles: whatever went wrong above, the counters must be zero again before the next
                // call (its waits would otherwise be released by THIS call's counts) — behind the GEMM, on its own stream
                (void)hipMemsetAsync(counters, 0, sizeof(unsigned) * Comm::kMaxPieces, np::stream());
                return failed;
            }
            // counters back to zero + "drained" published in one kernel; the library stream then waits for that number:
            // whatever the caller enqueues next (or np_sync) sees the gathered result
            ++cm.drained_seq;
            finish_kernel<<<1, 64, 0, cm.stream>>>(counters, chunks, cm.flags + 1, cm.drained_seq);
            NP_LAUNCH_CHECK("finish_kernel");
            flag_wait_kernel<<<1, 1, 0, np::stream()>>>(cm.flags + 1, cm.drained_seq, kWaitForPeers,   // producer: transfers with peers
                                                        np::device_error_word(), np::kErrCommWait);
            NP_LAUNCH_CHECK("flag_wait_kernel");
            cm.pending = false;
            return NP_OK;
        }
        // (a shape whose plan is not a single tiled launch: nothing was launched, take the per-piece form)
    }
    for (int c = 0; c < chunks; ++c) {
        size_t lo = 0, count = 0;
        if (int rc = np_comm_piece(slab, chunks, c, &lo, &count)) return rc;
        if (int rc = np::sgemm_batched_piece(count, slab, M, N, K, A + lo * stride_a, stride_a, B + lo * stride_b, stride_b,
                                             mine + lo * mat, mat))   // planned as the whole slab: same kernels as the one-call form
            return rc;
        // piece c of rank r belongs at C_full + r * slab + lo: destinations one slab apart
        size_t send_off = 0, recv_off = 0, stride = 0;
        piece_addresses(g_comm.rank, slab, mat * sizeof(float), lo, &send_off, &recv_off, &stride);
        if (int rc = np_allgather_async((const char *)C_full + send_off, (char *)C_full + recv_off, count * mat * sizeof(float),
                                        stride, chunks == 1 ? mode : NP_GATHER_P2P))
            return rc;
    }
    return np_comm_wait();   // whatever the caller enqueues next (or np_sync) sees the gathered result
}

// testing: what np_sgemm_strided_batched_allgather(slab items of item_bytes, `chunks` pieces, point to point) makes rank
// `rank` of `world` send and receive — computed by the SAME functions the real path uses (np_comm_piece, piece_addresses,
// p2p_peers), no device, no communicator.  host_out receives records of 6 values {piece, send to, send offset, bytes,
// receive from, receive offset} (offsets in bytes from the start of the replicated result), *host_count how many.
int np_comm_debug_plan(int rank, int world, size_t slab, size_t item_bytes, int chunks, unsigned long long *host_out,
                       size_t max_records, size_t *host_count) {
    if (world < 1 || rank < 0 || rank >= world || chunks < 1 || !host_count || (!host_out && max_records))
        return np::fail(NP_ERR_INVALID, "np_comm_debug_plan: bad arguments");
    if ((size_t)chunks > slab) chunks = slab ? (int)slab : 1;
    size_t n = 0;
    for (int c = 0; c < chunks; ++c) {
        size_t lo = 0, count = 0;
        if (int rc = np_comm_piece(slab, chunks, c, &lo, &count)) return rc;
        size_t send_off = 0, recv_base = 0, stride = 0;
        piece_addresses(rank, slab, item_bytes, lo, &send_off, &recv_base, &stride);
        for (int step = 1; step < world; ++step) {
            int to = 0, from = 0;
            p2p_peers(rank, world, step, &to, &from);
            if (n < max_records) {
                unsigned long long *r = host_out + 6 * n;
                r[0] = (unsigned long long)c;
                r[1] = (unsigned long long)to;
                r[2] = send_off;
                r[3] = count * item_bytes;
                r[4] = (unsigned long long)from;
                r[5] = recv_base + (size_t)from * stride;
            }
            ++n;
        }
    }
    *host_count = n;
    return NP_OK;
}

// the step model behind chunks = 0, for any world / shape, without a device or a communicator: *host_chunks = the piece count
// it picks, host_ms5[i] = the modelled step with 1, 2, 4, 8, 16 pieces (1e300 where the slab has fewer items)
int np_comm_debug_model(int world, size_t slab, size_t M, size_t N, size_t K, int cus, int *host_chunks, double *host_ms5) {
    if (world < 1 || slab == 0 || M == 0 || N == 0 || K == 0 || !host_chunks)
        return np::fail(NP_ERR_INVALID, "np_comm_debug_model: bad arguments");
    *host_chunks = model_pieces(world, slab, M, N, K, cus, host_ms5);
    return NP_OK;
}

// testing: one grouped ncclSend / ncclRecv pair from this rank to itself on the communication stream (the only way
// to run the p2p transport on a box with one GPU); dst <- src, then np_comm_wait().
int np_comm_debug_sendrecv_self(const void *dev_src, void *dev_dst, size_t bytes) {
    if (int rc = need_comm("np_comm_debug_sendrecv_self")) return rc;
    if (!dev_src || !dev_dst || bytes == 0) return np::fail(NP_ERR_INVALID, "np_comm_debug_sendrecv_self: bad arguments");
    if (int rc = comm_stream_follows_compute()) return rc;
    g_comm.pending = true;
    NP_RCCL_CHECK(g_comm.api.GroupStart());
    ncclResult_t rc = g_comm.api.Send(dev_src, bytes, ncclChar, g_comm.rank, g_comm.comm, g_comm.stream);
    if (rc == ncclSuccess) rc = g_comm.api.Recv(dev_dst, bytes, ncclChar, g_comm.rank, g_comm.comm, g_comm.stream);
    const ncclResult_t rc2 = g_comm.api.GroupEnd();
    if (rc != ncclSuccess || rc2 != ncclSuccess)
        return np::fail(NP_ERR_DEVICE, "self ncclSend/ncclRecv failed: %s", g_comm.api.GetErrorString(rc != ncclSuccess ? rc : rc2));
    return np_comm_wait();
}

// testing: `count` self transfers of `bytes` (dst <- src through one grouped ncclSend / ncclRecv pair each) on the
// communication stream, NOT ordered behind the library stream — whatever the library stream is running at the moment (a
// GEMM loop) competes with them for the CUs — each bracketed by its own event pair; returns once the communication stream
// has drained, host_ms[i] = the i-th transfer's duration.  The regression test of "a transfer issued next to a running GEMM
// still gets through" (tests/test_gpu_comm.py), which is what the overlapped pipeline rests on.
int np_comm_debug_loopback_timed(const void *dev_src, void *dev_dst, size_t bytes, int count, float *host_ms) {
    if (int rc = need_comm("np_comm_debug_loopback_timed")) return rc;
    if (!dev_src || !dev_dst || bytes == 0 || count < 1 || count > 64 || !host_ms)
        return np::fail(NP_ERR_INVALID, "np_comm_debug_loopback_timed: bad arguments");
    std::vector<hipEvent_t> ev((size_t)count * 2, nullptr);
    int rc = NP_OK;
    for (hipEvent_t &e : ev)
        if (rc == NP_OK && hipEventCreate(&e) != hipSuccess) rc = np::fail(NP_ERR_DEVICE, "hipEventCreate failed");
    for (int i = 0; i < count && rc == NP_OK; ++i) {
        if (hipEventRecord(ev[2 * i], g_comm.stream) != hipSuccess) rc = np::fail(NP_ERR_DEVICE, "hipEventRecord failed");
        ncclResult_t r0 = g_comm.api.GroupStart();
        if (r0 == ncclSuccess) r0 = g_comm.api.Send(dev_src, bytes, ncclChar, g_comm.rank, g_comm.comm, g_comm.stream);
        if (r0 == ncclSuccess) r0 = g_comm.api.Recv(dev_dst, bytes, ncclChar, g_comm.rank, g_comm.comm, g_comm.stream);
        const ncclResult_t r1 = g_comm.api.GroupEnd();
        if (r0 != ncclSuccess || r1 != ncclSuccess)
            rc = np::fail(NP_ERR_DEVICE, "self ncclSend/ncclRecv failed: %s", g_comm.api.GetErrorString(r0 != ncclSuccess ? r0 : r1));
        if (rc == NP_OK && hipEventRecord(ev[2 * i + 1], g_comm.stream) != hipSuccess) rc = np::fail(NP_ERR_DEVICE, "hipEventRecord failed");
    }
    if (hipStreamSynchronize(g_comm.stream) != hipSuccess && rc == NP_OK) rc = np::fail(NP_ERR_DEVICE, "hipStreamSynchronize failed");
    for (int i = 0; i < count && rc == NP_OK; ++i)
        if (hipEventElapsedTime(host_ms + i, ev[2 * i], ev[2 * i + 1]) != hipSuccess) rc = np::fail(NP_ERR_DEVICE, "hipEventElapsedTime failed");
    for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipGetLastError();
    return rc;
}

int np_comm_max(float value, float *host_max) {
    if (!host_max) return np::fail(NP_ERR_INVALID, "np_comm_max: null output");
    if (int rc = need_comm("np_comm_max")) return rc;
    if (int rc = np_comm_wait()) return rc;   // "every rank's stream has reached this point" includes its gathers
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
    // a wait for peers that is still spinning (a rank that never arrived) is released by the abort word; what it guarded
    // is reported as incomplete by the next np_sync / np_memcpy_d2h
    // — after a grace period: a gather that is merely still travelling is waited for, as before
    unsigned *err = np::device_error_word();
    const double give_up = now_s() + grace_seconds(30.0);
    while (err && hipStreamQuery(np::stream()) == hipErrorNotReady) {
        if (now_s() > give_up) {
            __atomic_store_n(err + 1, 1u, __ATOMIC_RELEASE);
            break;
        }
        usleep(200);
    }
    (void)hipGetLastError();
    // the communication stream may sit in an RCCL kernel whose peer never arrived: the abort word does not reach that kernel,
    // and hipStreamSynchronize on it would never return.  Give it the rest of the grace period, then abort the communicator
    // (ncclCommAbort makes its kernels leave) instead of destroying it.
    bool aborted = false;
    if (g_comm.stream) {
        const double comm_give_up = (now_s() > give_up ? now_s() : give_up) + 5.0;
        while (hipStreamQuery(g_comm.stream) == hipErrorNotReady && now_s() <= comm_give_up) usleep(200);
        (void)hipGetLastError();
        if (hipStreamQuery(g_comm.stream) == hipErrorNotReady && g_comm.api.CommAbort) {
            (void)hipGetLastError();
            if (err) __atomic_store_n(err + 1, 1u, __ATOMIC_RELEASE);
            (void)g_comm.api.CommAbort(g_comm.comm);
            aborted = true;
        }
        (void)hipGetLastError();
        (void)hipStreamSynchronize(g_comm.stream);
    }
    (void)hipStreamSynchronize(np::stream());
    if (err) __atomic_store_n(err + 1, 0u, __ATOMIC_RELEASE);
    const ncclResult_t rc = aborted ? ncclSuccess : g_comm.api.CommDestroy(g_comm.comm);
    g_comm.comm = nullptr;
    destroy_stream_objects();
    if (g_comm.scratch) np_free(g_comm.scratch);
    g_comm.scratch = nullptr;
    if (!g_comm.file_to_remove.empty()) {
        unlink(g_comm.file_to_remove.c_str());
        g_comm.file_to_remove.clear();
    }
    if (rc != ncclSuccess) return np::fail(NP_ERR_DEVICE, "ncclCommDestroy failed: %s", g_comm.api.GetErrorString(rc));
    if (aborted) return np::fail(NP_ERR_DEVICE, "np_comm_destroy: a transfer never completed (a peer is gone); the communicator was aborted");
    return NP_OK;
}

int np_comm_rccl_version(int *host_version) {
    if (!host_version) return np::fail(NP_ERR_INVALID, "np_comm_rccl_version: null output");
    *host_version = 0;
    if (int rc = load_rccl(g_comm.api)) return rc;
    if (!g_comm.api.GetVersion) return np::fail(NP_ERR_DEVICE, "np_comm_rccl_version: librccl has no ncclGetVersion");
    NP_RCCL_CHECK(g_comm.api.GetVersion(host_version));
    return NP_OK;
}

int np_comm_set_wait_limit(double seconds) {
    if (!(seconds >= 0.0) || seconds > 1e7) return np::fail(NP_ERR_INVALID, "np_comm_set_wait_limit: %g s is not a limit (0 = never give up)", seconds);
    unsigned long long ticks = (unsigned long long)(seconds * 1e8);   // wall_clock64 ticks at 100 MHz
    if (seconds > 0.0 && ticks == 0) ticks = 1;
    __atomic_store_n(&g_peer_wait_ticks, ticks, __ATOMIC_RELAXED);
    return NP_OK;
}

}  // extern "C"
