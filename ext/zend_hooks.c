/*
 * zend_hooks.c — np_ext_hooks.h for the real PHP extension build (`phpize && ./configure
 * --with-hip=<prefix>`, ext/config.m4).  This is the ONLY file of the glue that needs PHP's
 * headers; it is therefore the only one the numpower_amd build does not compile (the build image
 * has no PHP).  Everything it does is two statements, mirroring src/gpu_alloc.c:12,15,31.
 */
#include <Zend/zend.h>

#include "src/buffer.h"        /* MAIN_MEM_STACK (src/buffer.h:9-16) */
#include "np_ext_hooks.h"

void np_ext_throw(const char *message) { zend_throw_error(NULL, "%s", message); }

int np_ext_count_device_alloc(int delta) {
    MAIN_MEM_STACK.totalGPUAllocated += delta;
    return MAIN_MEM_STACK.totalGPUAllocated;
}
