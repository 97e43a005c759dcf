/*
 * np_ext_hooks.h — the two things the --with-hip glue needs from its host and nothing else.
 *
 * The glue files in this directory (gpu_alloc_hip.c, hip_math.c, hip_math_drivers.c) are plain C
 * over include/np_hip.h.  They never include a Zend header; what the reference's CUDA glue takes
 * from PHP — raising an Error (zend_throw_error, gpu_alloc.c:15) and the per-request device
 * allocation counter (MAIN_MEM_STACK.totalGPUAllocated, gpu_alloc.c:12,31,37; src/buffer.h:13) —
 * comes in through these hooks, defined exactly once per build:
 *
 *   PHP extension build      ext/zend_hooks.c       zend_throw_error / MAIN_MEM_STACK
 *   libnumpower_host.so      numpower_host.cpp      the host library's error handler
 *   libnp_hipmath.so (tests) ext/standalone_hooks.c last message + counter readable from C / ctypes
 */
#ifndef NUMPOWER_AMD_EXT_HOOKS_H
#define NUMPOWER_AMD_EXT_HOOKS_H

#ifdef __cplusplus
extern "C" {
#endif

/* Raise `message` the way the host raises errors; returns to the caller (which then returns). */
void np_ext_throw(const char *message);
/* +1 on vmalloc, -1 on vfree; returns the new count (what vmemcheck reports). */
int np_ext_count_device_alloc(int delta);

/* Raise the C ABI's last error through the host and ACKNOWLEDGE a device error while doing so (np_hip.h: device errors are sticky
 * until np_clear_device_error()).  For a binding the raised exception IS the report; an error word nobody acknowledges would fail
 * every later sync point of the process — a php-fpm worker poisoned by one request.  (np_clear_device_error waits for the device
 * and costs nothing when no error is up.) */
const char *np_last_error(void);
int np_clear_device_error(unsigned *host_bits);
static inline void np_ext_throw_last(void) {
    np_ext_throw(np_last_error());
    (void) np_clear_device_error((unsigned *) 0);
}

#ifdef __cplusplus
}
#endif
#endif
