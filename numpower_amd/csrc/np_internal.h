// Internal helpers shared by the HIP translation units of libnp_hip.so (not part of the ABI).
#ifndef NUMPOWER_AMD_NP_INTERNAL_H
#define NUMPOWER_AMD_NP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "np_hip.h"

namespace np {

// Records msg as the calling thread's last error and returns code.
int fail(int code, const char *fmt, ...);
// Library stream (never the null stream unless the caller installed it).
hipStream_t stream();
// Makes sure np_init ran (lazily initialises device 0).
int ensure_init();
// Device properties cached at init.
int num_cus();

// Scratch from the caching pool, released back to the pool when the object dies.  Freed blocks
// are only reused by later launches on the same stream, so releasing right after the launch
// that uses them is safe (stream order).
struct Scratch {
    void *ptr = nullptr;
    int alloc(size_t bytes);
    ~Scratch();
};

}  // namespace np

#define NP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return np::fail(NP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define NP_LAUNCH_CHECK(name)                                                           \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess)                                                           \
            return np::fail(NP_ERR_DEVICE, "launch of %s failed: %s", name,             \
                            hipGetErrorString(_e));                                     \
    } while (0)

#endif
