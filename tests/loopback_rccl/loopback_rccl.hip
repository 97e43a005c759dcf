// loopback_rccl — TEST INFRASTRUCTURE, never part of the product: a stand-in for librccl.so.1 that lets several RANKS SHARE ONE
// GPU, so that np_comm's sharded batched matmul (numpower_amd/csrc/np_comm.hip) runs with real peer processes on the one-GPU
// boxes this repository is developed on.  RCCL itself refuses two ranks on one device ("Duplicate GPU detected"), which is why
// tests/test_gpu_comm_multi.py has skipped on every lease.  np_comm loads RCCL with dlopen("librccl.so.1"): a worker started with
// LD_LIBRARY_PATH=tests/loopback_rccl/lib (and no torch in the process) gets THIS library instead — no switch in the product.
//
// What it is: the dozen entry points np_comm.hip uses (ncclGetUniqueId, ncclCommInitRank / Destroy / Abort, ncclAllGather,
// ncclAllReduce of floats, ncclSend / ncclRecv inside ncclGroupStart / End, ncclGetErrorString, ncclGetVersion), with RCCL's
// semantics as np_comm relies on them: every call returns at once, the work is ordered on the stream it was given, a receiving
// rank's stream does not pass the call before the data of every peer has arrived, calls of one communicator execute in the order
// they were made.  Transport: a POSIX shared-memory segment (named after the ncclUniqueId) that every rank maps and registers
// with HIP; data is staged through it with hipMemcpyAsync (device -> segment on the sender's stream, segment -> device on the
// receiver's), and one-lane kernels on the same streams publish / await sequence numbers in the segment with system-scope
// atomics.  What it is NOT: xGMI, peer-to-peer addressing, RCCL's kernels — what a run over this library proves is np_comm's
// own logic with a peer (rendezvous, piece order and addresses, the two streams and their device-side flags, the
// progress-reporting single launch feeding transfers, ragged pieces, abort), not the fabric.  ncclGetVersion answers 99901 so a
// test can tell which library a process got.
//
// Every device-side wait gives up after 30 s (NP_LOOPBACK_GIVE_UP_S) and marks the communicator failed (the next call returns an error): a bug ends as a
// failed test, not as a hung GPU.
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <vector>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
enum { ncclChar = 0, ncclFloat = 7 };
enum { ncclSum = 0, ncclMax = 2 };

constexpr int kMaxRanks = 4;
constexpr size_t kSlot = size_t(1) << 20;             // bytes staged per round and rank (the segment is 24 MiB: a container's /dev/shm may be 64)
unsigned long long g_give_up_ticks = 30ull * 100000000ull;   // 30 s of the 100 MHz wall clock (NP_LOOPBACK_GIVE_UP_S, read at
                                                             // ncclCommInitRank: the dead-peer test wants ncclCommAbort to be what ends a wait)

struct alignas(64) Flag { unsigned v; unsigned pad[15]; };

struct Shared {
    unsigned magic, world;
    unsigned arrived;                                  // ranks that have mapped the segment
    unsigned failed;                                   // a wait gave up, or ncclCommAbort: every wait ends, every call fails
    Flag gather_ready[kMaxRanks], gather_done[kMaxRanks];
    Flag reduce_ready[kMaxRanks], reduce_done[kMaxRanks];
    Flag p2p_ready[kMaxRanks][kMaxRanks], p2p_done[kMaxRanks][kMaxRanks];   // [source][destination]
    alignas(4096) char gather_slot[2][kMaxRanks][kSlot];
    char reduce_slot[2][kMaxRanks][4096];
    char p2p_slot[kMaxRanks][kMaxRanks][kSlot];
};
constexpr unsigned kMagic = 0x4c425243u;   // "LBRC"

struct ncclComm {
    Shared *sh = nullptr;                  // host mapping (== device pointer after hipHostRegister, asked for explicitly below)
    Shared *dev = nullptr;
    int rank = 0, world = 0;
    unsigned gather_seq = 0, reduce_seq = 0;
    unsigned send_seq[kMaxRanks] = {}, recv_seq[kMaxRanks] = {};
    hipStream_t last_stream = nullptr;
    bool any_op = false;
    hipEvent_t order = nullptr;            // calls of one communicator execute in call order, whatever streams they were given
};
typedef ncclComm Comm;

namespace {

struct Pending { bool send; const void *src; void *dst; size_t bytes; int peer; Comm *comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Pending> g_group;

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

__global__ void publish_kernel(unsigned *flag, unsigned value) {
    // the copy in front of this kernel on the stream is complete; make it visible beyond the device before the number is
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits until flag[i * stride_words] has reached `target` for every i < n (sequence numbers, signed difference)
__global__ void await_kernel(const unsigned *flag, unsigned stride_words, int n, unsigned target, unsigned *failed,
                             unsigned long long give_up_ticks) {
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        for (;;) {
            const unsigned v = __hip_atomic_load(flag + (size_t)i * stride_words, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((int)(v - target) >= 0) break;
            if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
            if (wall_clock64() - t0 > give_up_ticks) {
                __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __threadfence_system();
}

// out[i] = max / sum over ranks of the floats every rank staged (read straight from the segment: a few values)
__global__ void reduce_kernel(const char *slots, size_t slot_stride, int world, int count, int op, float *out) {
    for (int i = (int)threadIdx.x; i < count; i += (int)blockDim.x) {
        float acc = 0.0f;
        for (int r = 0; r < world; ++r) {
            const unsigned bits = __hip_atomic_load((const unsigned *)(slots + (size_t)r * slot_stride) + i, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_SYSTEM);
            const float v = __uint_as_float(bits);
            acc = r == 0 ? v : op == ncclMax ? (v > acc ? v : acc) : acc + v;
        }
        out[i] = acc;
    }
}

#define LB_HIP(expr)                                                                       \
    do {                                                                                   \
        if ((expr) != hipSuccess) {                                                        \
            fprintf(stderr, "loopback_rccl: %s failed: %s\n", #expr, hipGetErrorString(hipGetLastError())); \
            return ncclUnhandledCudaError;                                                 \
        }                                                                                  \
    } while (0)

ncclResult_t usable(Comm *c) {
    if (!c || !c->sh) return ncclInvalidArgument;
    if (__atomic_load_n(&c->sh->failed, __ATOMIC_RELAXED)) return ncclInternalError;
    return ncclSuccess;
}

// calls of one communicator run in call order: an op given another stream than its predecessor waits for it
ncclResult_t enter(Comm *c, hipStream_t s) {
    if (c->any_op && c->last_stream != s) LB_HIP(hipStreamWaitEvent(s, c->order, 0));
    return ncclSuccess;
}
ncclResult_t leave(Comm *c, hipStream_t s) {
    LB_HIP(hipEventRecord(c->order, s));
    c->last_stream = s;
    c->any_op = true;
    return ncclSuccess;
}

constexpr unsigned kFlagWords = sizeof(Flag) / sizeof(unsigned);

ncclResult_t all_gather(Comm *c, const char *send, char *recv, size_t bytes, hipStream_t s) {
    for (size_t off = 0; off < bytes || (bytes == 0 && off == 0); off += kSlot) {
        const size_t n = bytes - off < kSlot ? bytes - off : kSlot;
        const unsigned q = ++c->gather_seq, p = q & 1u;
        // the slot of round q was last read in round q - 2, by everyone
        await_kernel<<<1, 1, 0, s>>>(&c->dev->gather_done[0].v, kFlagWords, c->world, q - 2u, &c->dev->failed, g_give_up_ticks);
        if (n) LB_HIP(hipMemcpyAsync(c->sh->gather_slot[p][c->rank], send + off, n, hipMemcpyDeviceToHost, s));
        publish_kernel<<<1, 1, 0, s>>>(&c->dev->gather_ready[c->rank].v, q);
        await_kernel<<<1, 1, 0, s>>>(&c->dev->gather_ready[0].v, kFlagWords, c->world, q, &c->dev->failed, g_give_up_ticks);
        for (int r = 0; r < c->world && n; ++r)
            LB_HIP(hipMemcpyAsync(recv + (size_t)r * bytes + off, c->sh->gather_slot[p][r], n, hipMemcpyHostToDevice, s));
        publish_kernel<<<1, 1, 0, s>>>(&c->dev->gather_done[c->rank].v, q);
        if (bytes == 0) break;
    }
    LB_HIP(hipGetLastError());
    return ncclSuccess;
}

ncclResult_t run_group(std::vector<Pending> &ops) {
    if (ops.empty()) return ncclSuccess;
    Comm *c = ops[0].comm;
    hipStream_t s = ops[0].stream;
    for (const Pending &o : ops)
        if (o.comm != c || o.stream != s) return ncclInvalidUsage;     // (np_comm puts a group on one stream of one communicator)
    if (ncclResult_t rc = usable(c)) return rc;
    if (ncclResult_t rc = enter(c, s)) return rc;
    size_t longest = 0;
    for (const Pending &o : ops) longest = o.bytes > longest ? o.bytes : longest;
    // round k of every send, then round k of every receive: a send of round k + 1 only ever waits for receives of round k
    for (size_t off = 0; off < longest || (longest == 0 && off == 0); off += kSlot) {
        for (int pass = 0; pass < 2; ++pass) {
            for (Pending &o : ops) {
                if (o.send != (pass == 0) || (off >= o.bytes && !(o.bytes == 0 && off == 0))) continue;
                const size_t n = o.bytes - off < kSlot ? o.bytes - off : kSlot;
                if (o.send) {
                    const unsigned q = ++c->send_seq[o.peer];
                    await_kernel<<<1, 1, 0, s>>>(&c->dev->p2p_done[c->rank][o.peer].v, kFlagWords, 1, q - 1u, &c->dev->failed, g_give_up_ticks);
                    if (n) LB_HIP(hipMemcpyAsync(c->sh->p2p_slot[c->rank][o.peer], (const char *)o.src + off, n, hipMemcpyDeviceToHost, s));
                    publish_kernel<<<1, 1, 0, s>>>(&c->dev->p2p_ready[c->rank][o.peer].v, q);
                } else {
                    const unsigned q = ++c->recv_seq[o.peer];
                    await_kernel<<<1, 1, 0, s>>>(&c->dev->p2p_ready[o.peer][c->rank].v, kFlagWords, 1, q, &c->dev->failed, g_give_up_ticks);
                    if (n) LB_HIP(hipMemcpyAsync((char *)o.dst + off, c->sh->p2p_slot[o.peer][c->rank], n, hipMemcpyHostToDevice, s));
                    publish_kernel<<<1, 1, 0, s>>>(&c->dev->p2p_done[o.peer][c->rank].v, q);
                }
            }
        }
        if (longest == 0) break;
    }
    LB_HIP(hipGetLastError());
    return leave(c, s);
}

size_t type_bytes(int datatype) { return datatype == ncclChar ? 1 : datatype == ncclFloat ? 4 : 0; }

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int *version) {
    if (!version) return ncclInvalidArgument;
    *version = 99901;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t rc) {
    switch (rc) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "loopback_rccl: a HIP call failed";
        case ncclSystemError: return "loopback_rccl: a system call failed";
        case ncclInternalError: return "loopback_rccl: the communicator has failed (a wait gave up, or it was aborted)";
        case ncclInvalidArgument: return "loopback_rccl: invalid argument";
        default: return "loopback_rccl: invalid usage";
    }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof(id->internal), "/np_loopback_rccl_%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(Comm **out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || world > kMaxRanks || rank < 0 || rank >= world) return ncclInvalidArgument;
    id.internal[sizeof(id.internal) - 1] = 0;
    const char *name = id.internal;
    if (name[0] != '/') return ncclInvalidArgument;
    int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    const bool creator = fd >= 0;
    const double deadline = now_s() + 60.0;
    if (creator) {
        if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) {
            close(fd);
            shm_unlink(name);
            return ncclSystemError;
        }
    } else {
        for (;;) {
            fd = shm_open(name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size == sizeof(Shared)) break;
            if (fd >= 0) close(fd);
            if (now_s() > deadline) return ncclSystemError;
            usleep(1000);
        }
    }
    void *map = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED) {
        if (creator) shm_unlink(name);
        return ncclSystemError;
    }
    if (const char *e = getenv("NP_LOOPBACK_GIVE_UP_S")) {
        const double sec = atof(e);
        if (sec >= 1.0 && sec <= 600.0) g_give_up_ticks = (unsigned long long)(sec * 1e8);
    }
    Comm *c = new Comm;
    c->sh = (Shared *)map;
    c->rank = rank;
    c->world = world;
    if (creator) {
        c->sh->world = (unsigned)world;
        __atomic_store_n(&c->sh->magic, kMagic, __ATOMIC_RELEASE);
    } else {
        while (__atomic_load_n(&c->sh->magic, __ATOMIC_ACQUIRE) != kMagic) {
            if (now_s() > deadline) return ncclSystemError;
            usleep(1000);
        }
        if (c->sh->world != (unsigned)world) return ncclInvalidArgument;
    }
    if (hipHostRegister(map, sizeof(Shared), hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->dev, map, 0) != hipSuccess ||
        hipEventCreateWithFlags(&c->order, hipEventDisableTiming) != hipSuccess) {
        fprintf(stderr, "loopback_rccl: cannot register the shared segment with HIP: %s\n", hipGetErrorString(hipGetLastError()));
        munmap(map, sizeof(Shared));
        if (creator) shm_unlink(name);
        delete c;
        return ncclUnhandledCudaError;
    }
    __atomic_fetch_add(&c->sh->arrived, 1u, __ATOMIC_ACQ_REL);
    while (__atomic_load_n(&c->sh->arrived, __ATOMIC_ACQUIRE) < (unsigned)world) {
        if (now_s() > deadline) return ncclSystemError;
        usleep(1000);
    }
    if (creator) shm_unlink(name);          // everyone has it mapped: the name can go, the memory lives until the last unmap
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommAbort(Comm *c) {
    if (!c || !c->sh) return ncclInvalidArgument;
    __atomic_store_n(&c->sh->failed, 1u, __ATOMIC_RELEASE);     // every wait of every rank ends
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(Comm *c) {
    if (!c) return ncclInvalidArgument;
    if (c->sh) {
        (void)hipDeviceSynchronize();
        (void)hipHostUnregister(c->sh);
        munmap(c->sh, sizeof(Shared));
    }
    if (c->order) (void)hipEventDestroy(c->order);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, int datatype, Comm *c, hipStream_t s) {
    if (ncclResult_t rc = usable(c)) return rc;
    const size_t tb = type_bytes(datatype);
    if (!tb || (count && (!send || !recv))) return ncclInvalidArgument;
    if (ncclResult_t rc = enter(c, s)) return rc;
    if (ncclResult_t rc = all_gather(c, (const char *)send, (char *)recv, count * tb, s)) return rc;
    return leave(c, s);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, int datatype, int op, Comm *c, hipStream_t s) {
    if (ncclResult_t rc = usable(c)) return rc;
    if (datatype != ncclFloat || (op != ncclMax && op != ncclSum) || count == 0 || count * 4 > 4096 || !send || !recv)
        return ncclInvalidArgument;
    if (ncclResult_t rc = enter(c, s)) return rc;
    const unsigned q = ++c->reduce_seq, p = q & 1u;
    await_kernel<<<1, 1, 0, s>>>(&c->dev->reduce_done[0].v, kFlagWords, c->world, q - 2u, &c->dev->failed, g_give_up_ticks);
    LB_HIP(hipMemcpyAsync(c->sh->reduce_slot[p][c->rank], send, count * 4, hipMemcpyDeviceToHost, s));
    publish_kernel<<<1, 1, 0, s>>>(&c->dev->reduce_ready[c->rank].v, q);
    await_kernel<<<1, 1, 0, s>>>(&c->dev->reduce_ready[0].v, kFlagWords, c->world, q, &c->dev->failed, g_give_up_ticks);
    reduce_kernel<<<1, 64, 0, s>>>(&c->dev->reduce_slot[p][0][0], sizeof(c->dev->reduce_slot[p][0]), c->world, (int)count, op, (float *)recv);
    publish_kernel<<<1, 1, 0, s>>>(&c->dev->reduce_done[c->rank].v, q);
    LB_HIP(hipGetLastError());
    return leave(c, s);
}

ncclResult_t ncclGroupStart(void) {
    ++g_group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void) {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(g_group);
    return run_group(ops);
}

ncclResult_t ncclSend(const void *send, size_t count, int datatype, int peer, Comm *c, hipStream_t s) {
    if (ncclResult_t rc = usable(c)) return rc;
    const size_t tb = type_bytes(datatype);
    if (!tb || peer < 0 || peer >= c->world || (count && !send)) return ncclInvalidArgument;
    g_group.push_back(Pending{true, send, nullptr, count * tb, peer, c, s});
    if (g_group_depth > 0) return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(g_group);
    return run_group(ops);
}

ncclResult_t ncclRecv(void *recv, size_t count, int datatype, int peer, Comm *c, hipStream_t s) {
    if (ncclResult_t rc = usable(c)) return rc;
    const size_t tb = type_bytes(datatype);
    if (!tb || peer < 0 || peer >= c->world || (count && !recv)) return ncclInvalidArgument;
    g_group.push_back(Pending{false, nullptr, recv, count * tb, peer, c, s});
    if (g_group_depth > 0) return ncclSuccess;
    std::vector<Pending> ops;
    ops.swap(g_group);
    return run_group(ops);
}

}  // extern "C"
