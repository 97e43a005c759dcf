/*
 * standalone_hooks.c — np_ext_hooks.h for libnp_hipmath.so, the glue linked WITHOUT a host above it
 * (tests/test_gpu_ext_abi.py drives it through ctypes): the last raised message and the allocation
 * counter are kept here and can be read back.
 */
#include <stdio.h>
#include <string.h>

#include "np_ext_hooks.h"

static char g_last[512];
static int g_allocs;

void np_ext_throw(const char *message) {
    snprintf(g_last, sizeof(g_last), "%s", message ? message : "");
}

int np_ext_count_device_alloc(int delta) {
    g_allocs += delta;
    return g_allocs;
}

/* test access */
const char *np_ext_last_error(void);
void np_ext_clear_error(void);
int np_ext_device_allocs(void);

const char *np_ext_last_error(void) { return g_last; }
void np_ext_clear_error(void) { g_last[0] = 0; }
int np_ext_device_allocs(void) { return g_allocs; }
