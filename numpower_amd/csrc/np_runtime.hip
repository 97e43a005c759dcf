// Runtime + device-buffer layer of libnp_hip.so.
//
// Stands in for the reference's src/gpu_alloc.c (vmalloc / vfree / vmemcpy* / vmemcheck /
// NDArray_VFLOAT, gpu_alloc.c:11-54) and for the implicit CUDA context handling around it.
// Differences that matter on MI355X:
//   * sizes are size_t (the reference's `unsigned int` caps a buffer at 4 GiB; one GPU here
//     has 288 GB of HBM3E);
//   * allocations come from a caching pool, so the one-result-buffer-per-op pattern of the
//     reference's L2 (arithmetics.c:211-231) costs a free-list pop instead of hipMalloc;
//   * everything is ordered on one library stream and asynchronous; the only blocking calls
//     are np_sync, np_memcpy_d2h, np_read_float and the timers.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <chrono>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "np_internal.h"

namespace {

thread_local char g_err[512] = "";

constexpr int kMaxDevices = 16;

// What belongs to ONE device: its stream, its cached blocks, its ticket ring.  NDArray::setDevice
// (numpower.c:615-635 -> cudaSetDevice) only changes which device NEW work and NEW arrays go to; arrays
// of the device the process used before stay alive and usable, so nothing of the old device is torn down
// when the current device changes.
struct DeviceState {
    bool inited = false;
    int cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t cur_stream = nullptr;
    struct FreeBlock { void *ptr; unsigned stagger; };
    std::map<size_t, std::vector<FreeBlock>> free_blocks;   // rounded size -> cached blocks of this device
    unsigned *tickets = nullptr;                         // ring of zeroed device counters (np::next_ticket)
    unsigned ticket_seq = 0;
    unsigned *streamk_flags = nullptr;                   // np_sgemm.hip's stream-K flags (np::streamk_flags), zero between launches
};

struct Runtime {
    std::mutex mu;
    std::mutex result_mu;                                // one host-result call at a time (np::ResultCall)
    bool inited = false;
    int device = 0;                                      // the CURRENT device: new blocks, launches, np_sync
    DeviceState dev[kMaxDevices];
    struct Block { size_t size; int device; unsigned stagger; };
    std::unordered_map<void *, Block> live;              // ptr -> rounded size, owning device, offset from the hipMalloc base
    unsigned large_seq = 0;                              // running count of large blocks obtained from the driver
    float *slots = nullptr;                              // pinned host-result slots (np::ResultCall)
    unsigned *error_words = nullptr;                     // pinned, device-visible: [2 * device] error bits, [2 * device + 1] abort request
    int wait_mode = 2;                                   // np_runtime_set_variant: 0 = hipStreamSynchronize, 1 = spin on a stream-written flag,
                                                         // 2 = spin on the result slots themselves (armed with a sentinel)
    uint32_t wait_seq = 0;
    int armed = 0;                                       // slots [0, armed) hold `sentinel` until the kernel writes them
    uint32_t sentinel = 0;
    size_t reserved = 0;                                 // bytes held (live + cached), all devices
    long live_count = 0;
    DeviceState &cur() { return dev[device]; }
};

Runtime &rt() {
    static Runtime r;
    return r;
}

constexpr size_t kAskDriverFrom = size_t(64) << 20;     // np_malloc: requests from 64 MiB on look at the device's free memory first
constexpr size_t kDriverHeadroom = size_t(256) << 20;   // ... and want this much left next to them

size_t round_size(size_t bytes) {
    if (bytes == 0) bytes = 1;
    if (bytes < (size_t(1) << 20)) {
        size_t s = 256;
        while (s < bytes) s <<= 1;
        return s;
    }
    const size_t g = size_t(2) << 20;   // 2 MiB granules above 1 MiB
    return (bytes + g - 1) / g * g;
}

// Large blocks are handed out at base + k KiB (k = 0..3, cycling): three 400 MB operands whose addresses agree in
// their low bits make a streaming kernel's two reads and one write arrive at the same HBM channel at the same
// time; staggered by 1 KiB per operand the same add runs 1.5-2 % faster (profiles/r01/add_bw_stream_offsets.log,
// profiles/r02/add_offsets.log: 6157 -> 6255-6272 GB/s back to back).  The offset is a property of the block:
// it survives the trip through the free list and is undone when the block goes back to the driver.
constexpr size_t kStaggerFrom = size_t(1) << 20;   // blocks of at least 1 MiB
constexpr size_t kStaggerStep = 1024, kStaggerSlots = 4;

// Gives every cached block (of every device this process has used) back to the driver.
size_t trim_locked(Runtime &r) {
    size_t released = 0;
    for (DeviceState &d : r.dev) {
        for (auto &kv : d.free_blocks) {
            for (const DeviceState::FreeBlock &b : kv.second) {
                (void)hipFree((char *)b.ptr - b.stagger);
                released += kv.first;
            }
            kv.second.clear();
        }
        d.free_blocks.clear();
    }
    r.reserved -= released;
    return released;
}

struct Timer {
    hipEvent_t start, stop;
};

}  // namespace

namespace np {

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

hipStream_t stream() { return rt().cur().cur_stream; }
int g_cus_override = 0;   // np_debug_set_cus: what the planners take for the CU count (a CU-masked library stream reaches fewer)
int num_cus() { return g_cus_override > 0 ? g_cus_override : rt().cur().cus; }

// Entry-point guard.  First use: bind the library to the calling thread's CURRENT HIP device (whatever
// the caller — PHP's setDevice, torch.cuda.set_device — selected; device 0 only if nothing did).  Later:
// the HIP current device is per host thread, so a thread that never selected one would allocate on and
// launch from device 0 while the pool and the stream belong to r.device — re-select r.device whenever
// the thread's current device differs (a thread-local read; no driver call in the steady state).
// SIDE EFFECT (stated in np_hip.h): every np_* entry point leaves the calling thread's current HIP device
// set to the library's device — the same thing the reference's cudaSetDevice does to its process.
int ensure_init() {
    Runtime &r = rt();
    int cur = 0;
    if (!r.inited) {
        if (hipGetDevice(&cur) != hipSuccess) cur = 0;
        return np_init(cur);
    }
    if (hipGetDevice(&cur) == hipSuccess && cur != r.device) NP_HIP_CHECK(hipSetDevice(r.device));
    return NP_OK;
}

float *result_slots(int count);
int result_wait();

// One host-result call at a time.  The pinned slots, the sentinel and the armed count are process-wide,
// so two threads inside np_reduce_all / np_all / np_moments / np_order_stat ... at once would re-arm each
// other's slots and read each other's values (ctypes drops the GIL; a ZTS PHP has real threads).  The guard
// holds result_mu from the moment the slots are armed until the value has been read back; an uncontended
// lock is ~20 ns against the ~8 us such a call costs.
ResultCall::ResultCall(int count) {
    rt().result_mu.lock();
    slot = result_slots(count);
}
ResultCall::~ResultCall() {
    rt().armed = 0;
    rt().result_mu.unlock();
}
int ResultCall::wait() {
    if (int rc = result_wait()) return rc;
    return check_device_error("host result");
}

float *result_slots(int count) {
    Runtime &r = rt();
    if (!r.slots) {
        void *p = nullptr;
        const hipError_t e = hipHostMalloc(&p, 64 * sizeof(float), hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fail(NP_ERR_ALLOC, "pinned result slots: %s", hipGetErrorString(e));
            return nullptr;
        }
        r.slots = (float *)p;
    }
    // Arm the slots the coming kernels will write: a signalling-NaN bit pattern that changes with every call.  The
    // host then only has to watch them change (result_wait) — no flag, no stream operation behind the kernels.
    r.armed = 0;
    if (r.wait_mode == 2 && count > 0 && count <= 32) {
        // payload in [0x7f800001, 0x7fbfffff): always a SIGNALLING NaN, never 0x7fc00000 — the default quiet NaN a
        // reduction may legitimately produce (which would read as "still pending" and cost the 2 ms fallback)
        r.sentinel = 0x7f800001u + (++r.wait_seq % 0x3ffffeu);
        volatile uint32_t *w = (volatile uint32_t *)r.slots;
        for (int i = 0; i < count; ++i) w[i] = r.sentinel;
        r.armed = count;
    }
    return r.slots;
}

constexpr unsigned kTicketRing = 1024;
unsigned *next_tickets(unsigned count);

// np_init allocates the ring (not the first reduction: that one may be inside a stream capture)
static int alloc_tickets_locked(DeviceState &d) {
    if (d.tickets) return NP_OK;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, kTicketRing * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(p, 0, kTicketRing * sizeof(unsigned));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NP_ERR_ALLOC, "ticket ring: %s", hipGetErrorString(e));
    }
    d.tickets = (unsigned *)p;   // one ring per device this process has used
    return NP_OK;
}

// ... and so does the stream-K flag array (np_sgemm.hip): a first stream-K launch inside np_graph_begin .. np_graph_end
// must not meet a hipMalloc + synchronous hipMemset (the reason the ticket ring moved here)
static int alloc_streamk_flags_locked(DeviceState &d) {
    if (d.streamk_flags) return NP_OK;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, kStreamKFlagCount * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(p, 0, kStreamKFlagCount * sizeof(unsigned));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NP_ERR_ALLOC, "stream-K flags: %s", hipGetErrorString(e));
    }
    d.streamk_flags = (unsigned *)p;
    return NP_OK;
}

unsigned *streamk_flags() {
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    DeviceState &d = r.cur();
    if (alloc_streamk_flags_locked(d) != NP_OK) return nullptr;   // (np_init did this already; a device first used some other way)
    return d.streamk_flags;
}

// The device-error words: pinned host memory every kernel can write, one pair per device.  A device-side wait that gives up
// (np_comm.hip's flag_wait_kernel, the stream-K finisher of np_sgemm.hip) ORs its bit into the word of the device it runs on;
// the host reports it at every point where a caller could otherwise consume a wrong result — np_sync, np_memcpy_d2h, a
// host-result call, any np_comm_* entry point (check_device_error) — instead of returning NP_OK with garbage.  STICKY and PER
// DEVICE (round 6; until then one process-wide word, cleared by its first reader — with replicas on several devices or
// threads whoever synchronised first ate the other's error and the owner got NP_OK): the word stays up until
// np_clear_device_error() acknowledges it, and a sync point of device 1 never looks at device 0's word.
static int alloc_error_word_locked(Runtime &r) {
    if (r.error_words) return NP_OK;
    void *p = nullptr;
    const hipError_t e = hipHostMalloc(&p, 2 * kMaxDevices * sizeof(unsigned), hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(NP_ERR_ALLOC, "pinned error words: %s", hipGetErrorString(e));
    }
    r.error_words = (unsigned *)p;
    memset(r.error_words, 0, 2 * kMaxDevices * sizeof(unsigned));
    return NP_OK;
}

unsigned *device_error_word() {
    Runtime &r = rt();
    return r.error_words ? r.error_words + 2 * r.device : nullptr;
}

unsigned long long g_launch_count = 0;

int check_device_error(const char *who) {
    Runtime &r = rt();
    if (!r.error_words) return NP_OK;
    const unsigned bits = __atomic_load_n(r.error_words + 2 * r.device, __ATOMIC_ACQUIRE);
    if (!bits) return NP_OK;
    return fail(NP_ERR_DEVICE, "%s: a device-side wait gave up before this point on device %d (%s%s%s): results produced since the "
                               "last successful np_sync are incomplete and must be discarded; np_clear_device_error() acknowledges",
                who, r.device, bits & kErrCommWait ? "a stream-ordering wait of np_comm timed out" : "",
                (bits & kErrCommWait) && (bits & kErrStreamK) ? "; " : "",
                bits & kErrStreamK ? "a GEMM workgroup folding K-split partial tiles never saw its siblings' (stream-K / in-launch split-K)" : "");
}

size_t g_small_reduce_blocks = 128;   // np_internal.h
size_t g_fold_in_kernel_max = 256;
unsigned *next_ticket() { return next_tickets(1); }

// `count` consecutive tickets (count <= kTicketRing / 4): a run that would cross the end of the ring starts over at slot 0
unsigned *next_tickets(unsigned count) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    DeviceState &d = r.cur();
    if (count == 0 || count > kTicketRing / 4) {
        fail(NP_ERR_INVALID, "internal: %u tickets asked of a ring of %u", count, kTicketRing);
        return nullptr;
    }
    if (alloc_tickets_locked(d) != NP_OK) return nullptr;
    unsigned at = d.ticket_seq % kTicketRing;
    if (at + count > kTicketRing) at = 0;
    d.ticket_seq = at + count;
    return d.tickets + at;
}

// Waiting for a host result.  hipStreamSynchronize costs 5-6 us of driver time per call on top of the kernels, a
// third of what nd::sum() of a small array takes.  The result lives in pinned, device-visible host memory, so the host
// can simply watch it arrive: result_slots(count) filled the slots with a sentinel (a signalling NaN whose payload
// changes from call to call), the last kernel overwrites them, and result_wait spins until none of them holds the
// sentinel any more.  (Mode 1, the round's first version, spins on a flag the command processor writes behind the
// kernels — hipStreamWriteValue32 — which turned out to be a 3.4 us kernel of its own.)  A result that happens to BE
// the sentinel (a NaN with exactly that payload carried through from the input), a wait longer than 2 ms, or a runtime
// that refuses the stream operation all end in hipStreamSynchronize: slower, never wrong.
// NOTE the mode-2 wait returns as soon as the slots have changed, i.e. possibly BEFORE the writing kernel has retired
// (its ticket reset, np_internal.h fold_in_last_workgroup, may still be in flight).  That is safe because everything
// that could observe the difference — the next launch drawing the same ticket, a pool block handed out again — is
// enqueued on the same stream and therefore ordered behind the kernel.
int result_wait() {
    Runtime &r = rt();
    const auto expired = [t0 = std::chrono::steady_clock::now()](unsigned spins) {
        return (spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2);
    };
    if (r.wait_mode == 2 && r.armed > 0 && r.slots) {
        volatile uint32_t *w = (volatile uint32_t *)r.slots;
        const int armed = r.armed;
        r.armed = 0;
        for (unsigned spins = 0;; ++spins) {
            bool pending = false;
            for (int i = 0; i < armed; ++i) pending |= w[i] == r.sentinel;
            if (!pending) return NP_OK;
            __builtin_ia32_pause();
            if (expired(spins)) break;
        }
    } else if (r.wait_mode == 1 && r.slots) {
        volatile uint32_t *flag = (volatile uint32_t *)(r.slots + 63);
        const uint32_t want = ++r.wait_seq;
        const hipError_t e = hipStreamWriteValue32(r.cur().cur_stream, (void *)flag, want, 0);
        if (e == hipSuccess) {
            for (unsigned spins = 0;; ++spins) {
                if (*flag == want) return NP_OK;
                __builtin_ia32_pause();
                if (expired(spins)) break;
            }
        } else {
            (void)hipGetLastError();
            r.wait_mode = 0;   // not supported here: never try again
        }
    }
    NP_HIP_CHECK(hipStreamSynchronize(r.cur().cur_stream));
    return NP_OK;
}

int Scratch::alloc(size_t bytes) { return np_malloc(&ptr, bytes); }
Scratch::~Scratch() {
    if (ptr) np_free(ptr);
}

}  // namespace np

extern "C" {

const char *np_last_error(void) { return g_err; }
const char *np_version(void) { return "numpower_amd 0.1 (gfx950 / CDNA4)"; }

int np_device_count(int *host_count) {
    if (!host_count) return np::fail(NP_ERR_INVALID, "np_device_count: null output");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *host_count = 0;
        return np::fail(NP_ERR_NODEVICE, "No GPU device available (%s)", hipGetErrorString(e));
    }
    *host_count = n;
    return NP_OK;
}

int np_init(int device) {
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return np::fail(NP_ERR_NODEVICE, "No GPU device available or HIP not enabled");
    if (device < 0 || device >= n || device >= kMaxDevices)
        return np::fail(NP_ERR_INVALID, "np_init: device %d out of range (0..%d)", device, (n < kMaxDevices ? n : kMaxDevices) - 1);
    NP_HIP_CHECK(hipSetDevice(device));
    DeviceState &d = r.dev[device];
    if (!d.inited) {
        hipDeviceProp_t prop;
        NP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        d.cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        NP_HIP_CHECK(hipStreamCreateWithFlags(&d.own_stream, hipStreamNonBlocking));
        d.cur_stream = d.own_stream;
        if (int rc = np::alloc_tickets_locked(d)) return rc;
        if (int rc = np::alloc_streamk_flags_locked(d)) return rc;
        if (int rc = np::alloc_error_word_locked(r)) return rc;
        d.inited = true;
    }
    // Switching devices tears nothing down: the old device keeps its stream (work in flight there completes), its
    // cached blocks and its live arrays — cudaSetDevice semantics (numpower.c:633).  Work enqueued from now on goes to
    // `device`; an array of the old device is usable again after switching back (or from the new device wherever the
    // platform gives peers access to each other's memory, as the reference's kernels would assume).
    r.device = device;
    r.inited = true;
    return NP_OK;
}

int np_set_device(int device) { return np_init(device); }

int np_runtime_set_variant(int variant) {
    if (variant < 0 || variant > 2)
        return np::fail(NP_ERR_INVALID, "np_runtime_set_variant: 0 = wait with hipStreamSynchronize, 1 = spin on a stream-written flag, 2 = spin on the result itself (default)");
    rt().wait_mode = variant;
    return NP_OK;
}

int np_sync(void) {
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipStreamSynchronize(rt().cur().cur_stream));
    return np::check_device_error("np_sync");
}

int np_clear_device_error(unsigned *host_bits) {
    if (host_bits) *host_bits = 0;
    if (int rc = np::ensure_init()) return rc;
    Runtime &r = rt();
    if (!r.error_words) return NP_OK;
    // Everything this device was given must have retired before its bookkeeping is touched: a GEMM workgroup that gave up
    // waiting for its siblings leaves its ticket / its stream-K flag where it stood — rings that later launches assume are
    // zero — and kernels on a stream the caller swapped out earlier (np_set_stream) may still hold live tickets (ADVICE r05:
    // a reset that is only ordered behind the CURRENT stream can wipe a healthy fold's count).  So: the whole device —
    // unless the device's communicator has a transfer in flight that waits for a peer (what the error usually IS, after a
    // rank died): the collective library's kernel does not end by itself and a device-wide wait would never return.  That
    // one is np_comm_destroy()'s to end (it aborts the communicator); until then the error stays.
    if (np::comm_transfers_stuck(r.device, 10.0))
        return np::fail(NP_ERR_DEVICE, "np_clear_device_error: device %d's communicator still has transfers in flight that wait for a "
                                       "peer (late, or gone); np_comm_destroy() ends them — acknowledge after it", r.device);
    NP_HIP_CHECK(hipDeviceSynchronize());
    const unsigned bits = __atomic_exchange_n(r.error_words + 2 * r.device, 0u, __ATOMIC_ACQ_REL);
    if (host_bits) *host_bits = bits;
    if (bits & np::kErrStreamK) {
        std::lock_guard<std::mutex> lk(r.mu);
        DeviceState &d = r.cur();
        if (d.tickets) NP_HIP_CHECK(hipMemset(d.tickets, 0, np::kTicketRing * sizeof(unsigned)));
        if (d.streamk_flags) NP_HIP_CHECK(hipMemset(d.streamk_flags, 0, np::kStreamKFlagCount * sizeof(unsigned)));
    }
    return NP_OK;
}

int np_debug_launch_count(unsigned long long *host_count) {
    if (!host_count) return np::fail(NP_ERR_INVALID, "np_debug_launch_count: null output");
    *host_count = __atomic_load_n(&np::g_launch_count, __ATOMIC_RELAXED);
    return NP_OK;
}

__global__ void raise_error_kernel(unsigned *word, unsigned bits) {
    __hip_atomic_fetch_or(word, bits, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int np_debug_raise_device_error(unsigned bits) {
    if (int rc = np::ensure_init()) return rc;
    if (!np::device_error_word()) return np::fail(NP_ERR_DEVICE, "np_debug_raise_device_error: no error word");
    raise_error_kernel<<<1, 1, 0, np::stream()>>>(np::device_error_word(), bits);
    NP_LAUNCH_CHECK("raise_error_kernel");
    return NP_OK;
}

// One wave spins for ~wall_ticks of the 100 MHz wall clock and reports how many shader cycles went by: the clock the
// power manager grants at THIS point of the stream (a VALU-bound kernel runs as fast as that clock, an HBM-bound one does
// not care — tools/pow_clock_probe.py uses it to tell the two apart).
__global__ void clock_probe_kernel(unsigned long long wall_ticks, unsigned long long *out) {
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    unsigned long long w1 = w0;
    while (w1 - w0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64();
    }
    out[0] = __builtin_readcyclecounter() - c0;
    out[1] = w1 - w0;
}

// testing / tools: the shader clock (MHz) as seen by a ~20 us probe kernel enqueued on the library stream now (synchronises)
int np_debug_clock_mhz(float *host_mhz) {
    if (!host_mhz) return np::fail(NP_ERR_INVALID, "np_debug_clock_mhz: null output");
    if (int rc = np::ensure_init()) return rc;
    np::Scratch buf;
    if (int rc = buf.alloc(2 * sizeof(unsigned long long))) return rc;
    clock_probe_kernel<<<1, 64, 0, np::stream()>>>(2000ull, (unsigned long long *)buf.ptr);
    NP_LAUNCH_CHECK("clock_probe_kernel");
    unsigned long long h[2] = {0, 0};
    NP_HIP_CHECK(hipMemcpyAsync(h, buf.ptr, sizeof(h), hipMemcpyDeviceToHost, np::stream()));
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    *host_mhz = h[1] ? (float)((double)h[0] / ((double)h[1] / 100.0)) : 0.0f;
    return NP_OK;
}

// tools: where the workgroups of a launch on the library stream land — HW_REG_HW_ID (wave / simd / cu / sh / se ids) and
// HW_REG_XCC_ID of each workgroup's first wave, after ~10 us of spinning so that a whole grid is resident at once.  With the library
// stream swapped for a CU-masked one (np_set_stream) this is how the mask's bit <-> (XCD, SE, CU) mapping is read off the
// hardware (tools/cu_mask_probe.py).
__global__ void hw_id_kernel(unsigned *out, unsigned long long spin_ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID, all 32 bits
        out[2 * blockIdx.x + 1] = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // XCC_ID
    }
}

int np_debug_set_cus(int cus) {
    if (cus < 0 || cus > 4096) return np::fail(NP_ERR_INVALID, "np_debug_set_cus: %d", cus);
    np::g_cus_override = cus;
    return NP_OK;
}

int np_debug_hw_ids(unsigned *host_out2, size_t workgroups) {
    if (!host_out2 || workgroups == 0 || workgroups > 65536) return np::fail(NP_ERR_INVALID, "np_debug_hw_ids: bad arguments");
    if (int rc = np::ensure_init()) return rc;
    np::Scratch buf;
    if (int rc = buf.alloc(2 * workgroups * sizeof(unsigned))) return rc;
    hw_id_kernel<<<(unsigned)workgroups, 64, 0, np::stream()>>>((unsigned *)buf.ptr, 1000ull);
    NP_LAUNCH_CHECK("hw_id_kernel");
    NP_HIP_CHECK(hipMemcpyAsync(host_out2, buf.ptr, 2 * workgroups * sizeof(unsigned), hipMemcpyDeviceToHost, np::stream()));
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    return NP_OK;
}

int np_set_stream(void *hip_stream) {
    if (int rc = np::ensure_init()) return rc;
    DeviceState &d = rt().cur();
    hipStream_t next = hip_stream ? (hipStream_t)hip_stream : d.own_stream;
    if (next == d.cur_stream) return NP_OK;   // per-chunk callers (parallel.hip_compute) must not block here
    // drain the stream we are leaving so pool reuse stays ordered
    NP_HIP_CHECK(hipStreamSynchronize(d.cur_stream));
    d.cur_stream = next;
    return NP_OK;
}

void *np_get_stream(void) {
    if (np::ensure_init()) return nullptr;
    return (void *)rt().cur().cur_stream;
}

/* ---- timers ---- */

int np_timer_create(void **timer) {
    if (!timer) return np::fail(NP_ERR_INVALID, "np_timer_create: null output");
    if (int rc = np::ensure_init()) return rc;
    Timer *t = new Timer;
    NP_HIP_CHECK(hipEventCreate(&t->start));
    NP_HIP_CHECK(hipEventCreate(&t->stop));
    *timer = t;
    return NP_OK;
}
int np_timer_start(void *timer) {
    if (!timer) return np::fail(NP_ERR_INVALID, "np_timer_start: null timer");
    NP_HIP_CHECK(hipEventRecord(((Timer *)timer)->start, rt().cur().cur_stream));
    return NP_OK;
}
int np_timer_stop(void *timer) {
    if (!timer) return np::fail(NP_ERR_INVALID, "np_timer_stop: null timer");
    NP_HIP_CHECK(hipEventRecord(((Timer *)timer)->stop, rt().cur().cur_stream));
    return NP_OK;
}
int np_timer_elapsed_ms(void *timer, float *host_ms) {
    if (!timer || !host_ms) return np::fail(NP_ERR_INVALID, "np_timer_elapsed_ms: null argument");
    Timer *t = (Timer *)timer;
    NP_HIP_CHECK(hipEventSynchronize(t->stop));
    NP_HIP_CHECK(hipEventElapsedTime(host_ms, t->start, t->stop));
    return NP_OK;
}
int np_timer_destroy(void *timer) {
    if (!timer) return NP_OK;
    Timer *t = (Timer *)timer;
    (void)hipEventDestroy(t->start);
    (void)hipEventDestroy(t->stop);
    delete t;
    return NP_OK;
}

/* ---- launch-bound sequences as HIP graphs ---- */
// A script that applies the same short sequence of ops to small arrays over and over is bound by
// launch latency (~5 us per kernel), not by the kernels.  Everything this library enqueues goes to
// one stream and nothing but the host-result calls synchronises, so a sequence can be captured
// once and replayed as ONE graph launch.  Rules of capture (HIP's): no call that returns a value to
// the host or copies to/from pageable memory inside the captured region, and run the sequence once
// beforehand so that the pool already holds every block it will ask for (hipMalloc cannot be
// captured).
int np_graph_begin(void) {
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipStreamBeginCapture(rt().cur().cur_stream, hipStreamCaptureModeThreadLocal));
    return NP_OK;
}

int np_graph_end(void **graph_exec) {
    if (!graph_exec) return np::fail(NP_ERR_INVALID, "np_graph_end: null output");
    *graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    NP_HIP_CHECK(hipStreamEndCapture(rt().cur().cur_stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return np::fail(NP_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_exec = (void *)exec;
    return NP_OK;
}

int np_graph_launch(void *graph_exec) {
    if (!graph_exec) return np::fail(NP_ERR_INVALID, "np_graph_launch: null graph");
    NP_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, rt().cur().cur_stream));
    return NP_OK;
}

int np_graph_destroy(void *graph_exec) {
    if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return NP_OK;
}

/* ---- device buffers ---- */

int np_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return np::fail(NP_ERR_INVALID, "np_malloc: null output pointer");
    *dev_ptr = nullptr;
    if (int rc = np::ensure_init()) return rc;
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    const size_t sz = round_size(bytes);
    void *p = nullptr;
    unsigned stagger = 0;
    DeviceState &d = r.cur();
    auto it = d.free_blocks.find(sz);
    if (it != d.free_blocks.end() && !it->second.empty()) {
        p = it->second.back().ptr;
        stagger = it->second.back().stagger;
        it->second.pop_back();
    } else {
        const size_t pad = sz >= kStaggerFrom ? kStaggerStep * kStaggerSlots : 0;
        // The driver of this pool does not refuse a request that no longer fits the device: it backs it with host memory (seen in round 6:
        // 308 GB on the 288 GB device succeeds).  A cache that goes back only when hipMalloc FAILS would therefore trade a full device for
        // a slow one — a long-lived process that frees buffers of many sizes would grow into host memory.  So for large requests the driver
        // is asked first, and when the blocks cached here are what stands between the request and the device's free memory, they go back.
        if (sz >= kAskDriverFrom) {
            bool cached = false;
            for (const DeviceState &o : r.dev)
                for (const auto &kv : o.free_blocks) cached = cached || !kv.second.empty();
            size_t free_b = 0, total_b = 0;
            if (cached && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < sz + pad + kDriverHeadroom) {
                for (DeviceState &o : r.dev)
                    if (o.inited) NP_HIP_CHECK(hipStreamSynchronize(o.cur_stream));
                trim_locked(r);
            }
        }
        hipError_t e = hipMalloc(&p, sz + pad);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // give cached blocks back to the driver and retry once
            for (DeviceState &o : r.dev)
                if (o.inited) NP_HIP_CHECK(hipStreamSynchronize(o.cur_stream));
            trim_locked(r);
            e = hipMalloc(&p, sz + pad);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return np::fail(NP_ERR_ALLOC, "device memory allocation failed (%zu bytes: %s)",
                                bytes, hipGetErrorString(e));
            }
        }
        r.reserved += sz;
        if (pad) {
            stagger = (unsigned)(kStaggerStep * (r.large_seq++ % kStaggerSlots));
            p = (char *)p + stagger;
        }
    }
    r.live[p] = Runtime::Block{sz, r.device, stagger};
    r.live_count++;
    *dev_ptr = p;
    return NP_OK;
}

int np_free(void *dev_ptr) {
    if (!dev_ptr) return NP_OK;
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    auto it = r.live.find(dev_ptr);
    if (it == r.live.end())
        return np::fail(NP_ERR_INVALID, "np_free: pointer %p was not allocated by np_malloc", dev_ptr);
    // back into the cache of the device that owns it — which need not be the current one after an
    // NDArray::setDevice: the block is reused the next time that device is current
    r.dev[it->second.device].free_blocks[it->second.size].push_back(DeviceState::FreeBlock{dev_ptr, it->second.stagger});
    r.live.erase(it);
    r.live_count--;
    return NP_OK;
}

long np_live_allocs(void) { return rt().live_count; }

int np_pool_trim(size_t *host_bytes) {
    if (int rc = np::ensure_init()) return rc;
    Runtime &r = rt();
    std::lock_guard<std::mutex> lk(r.mu);
    for (DeviceState &o : r.dev)
        if (o.inited) NP_HIP_CHECK(hipStreamSynchronize(o.cur_stream));
    size_t released = trim_locked(r);
    if (host_bytes) *host_bytes = released;
    return NP_OK;
}

size_t np_pool_reserved_bytes(void) { return rt().reserved; }

int np_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes) {
    if (bytes == 0) return NP_OK;
    if (!dev_dst || !host_src) return np::fail(NP_ERR_INVALID, "np_memcpy_h2d: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // Pageable source: hipMemcpyAsync stages it and returns once the host buffer is reusable,
    // which is the semantic NDArray_ToGPU needs (ndarray.c:1054-1055).
    NP_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, np::stream()));
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    return NP_OK;
}

int np_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return NP_OK;
    if (!host_dst || !dev_src) return np::fail(NP_ERR_INVALID, "np_memcpy_d2h: null pointer");
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, np::stream()));
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    return np::check_device_error("np_memcpy_d2h");   // a read-back of a result a device-side wait gave up on is an error, not data
}

int np_memcpy_d2d(void *dev_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return NP_OK;
    if (!dev_dst || !dev_src) return np::fail(NP_ERR_INVALID, "np_memcpy_d2d: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // large word-aligned copies: the library's own float4 stream beats the runtime's blit by ~20 %
    if (bytes >= (size_t(8) << 20) && bytes % 4 == 0 && ((uintptr_t)dev_dst & 3u) == 0 && ((uintptr_t)dev_src & 3u) == 0)
        return np::device_copy(dev_dst, dev_src, bytes);
    NP_HIP_CHECK(hipMemcpyAsync(dev_dst, dev_src, bytes, hipMemcpyDeviceToDevice, np::stream()));
    return NP_OK;
}

int np_memset0(void *dev_ptr, size_t bytes) {
    if (bytes == 0) return NP_OK;
    if (!dev_ptr) return np::fail(NP_ERR_INVALID, "np_memset0: null pointer");
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipMemsetAsync(dev_ptr, 0, bytes, np::stream()));
    return NP_OK;
}

int np_read_float(const float *dev_ptr, size_t index, float *host_out) {
    if (!dev_ptr || !host_out) return np::fail(NP_ERR_INVALID, "np_read_float: null pointer");
    return np_memcpy_d2h(host_out, dev_ptr + index, sizeof(float));
}

size_t np_avx_body_end(size_t numel_a) {
    if (numel_a < 8) return 0;
    return (numel_a - 7 + 7) / 8 * 8;
}

}  // extern "C"
