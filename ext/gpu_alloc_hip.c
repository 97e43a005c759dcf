/*
 * gpu_alloc_hip.c — the device-buffer layer of the reference (src/gpu_alloc.h:8-15, implemented
 * for CUDA in src/gpu_alloc.c:11-54) over libnp_hip.so.  Same seven symbols, same signatures, so
 * every call site in ndarray.c / initializers.c / arithmetics.c links unchanged.
 *
 * Differences a maintainer should know:
 *   - blocks come from np_malloc's caching pool: the result allocation every op makes
 *     (arithmetics.c:211-231) does not reach the driver in steady state;
 *   - the reference's `unsigned int size` caps a buffer at 4 GiB — and its call sites pass
 *     `numElements * sizeof(float)`, so a larger array is TRUNCATED, not refused.  Stand-alone
 *     (libnp_hipmath.so) these entry points keep that signature, which is what objects compiled against
 *     the reference's gpu_alloc.h call; in a `--with-hip` tree tools/apply_with_hip.py widens the three
 *     prototypes of src/gpu_alloc.h to size_t and the build defines NP_GPU_ALLOC_WIDE, so the byte
 *     counts arrive whole (one MI355X holds 288 GB);
 *   - nothing here synchronises the device except the two read-backs, which must.
 */
#include <stdio.h>

#include "np_ext_hooks.h"
#include "np_hip.h"

#include <stddef.h>

#ifdef NP_GPU_ALLOC_WIDE
/* a `--with-hip` tree: src/gpu_alloc.h as tools/apply_with_hip.py leaves it (byte counts as size_t) */
typedef size_t np_alloc_bytes;
void vmalloc(void **target, np_alloc_bytes size);
void vmemcpyd2d(char *target, char *dst, np_alloc_bytes size);
void vmemcpyh2d(char *target, char *dst, np_alloc_bytes size);
#else
/* prototypes = src/gpu_alloc.h:8-15 */
typedef unsigned int np_alloc_bytes;
void vmalloc(void **target, unsigned int size);
void vmemcpyd2d(char *target, char *dst, unsigned int size);
void vmemcpyh2d(char *target, char *dst, unsigned int size);
#endif
void vfree(void *target);
void vmemcheck(void);
float NDArray_VFLOAT(char *target);
float NDArray_VFLOATF_I(float *target, int index);

/* gpu_alloc.c:11-17 */
void vmalloc(void **target, np_alloc_bytes size) {
    np_ext_count_device_alloc(+1);
    if (np_malloc(target, (size_t)size) != NP_OK) np_ext_throw("device memory allocation failed");
}

/* gpu_alloc.c:30-33 */
void vfree(void *target) {
    np_ext_count_device_alloc(-1);
    if (np_free(target) != NP_OK) np_ext_throw_last();
}

/* gpu_alloc.c:36-40 (NDARRAY_VCHECK at request shutdown) */
void vmemcheck(void) {
    const int leaked = np_ext_count_device_alloc(0);
    if (leaked != 0) printf("\nVRAM MEMORY LEAK: leaked %d array(s)\n", leaked);
}

/* gpu_alloc.c:20-27.  NOTE the reference's argument order: (source, destination, bytes). */
void vmemcpyd2d(char *target, char *dst, np_alloc_bytes size) {
    if (np_memcpy_d2d(dst, target, (size_t)size) != NP_OK) np_ext_throw_last();
}

void vmemcpyh2d(char *target, char *dst, np_alloc_bytes size) {
    if (np_memcpy_h2d(dst, target, (size_t)size) != NP_OK) np_ext_throw_last();
}

/* gpu_alloc.c:43-54: one float back to the host (blocks). */
float NDArray_VFLOAT(char *target) {
    float value = 0.0f;
    if (np_read_float((const float *)target, 0, &value) != NP_OK) np_ext_throw_last();
    return value;
}

float NDArray_VFLOATF_I(float *target, int index) {
    float value = 0.0f;
    if (np_read_float(target, (size_t)index, &value) != NP_OK) np_ext_throw_last();
    return value;
}
