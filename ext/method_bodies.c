/*
 * method_bodies.c — the device branch of the reference's PHP_METHODs, as a plain C program.
 *
 * PHP and its headers are not in this image, so numpower.c itself cannot be compiled here.  What CAN
 * be shown is that the statements its methods execute once `NDArray_DEVICE(nda) == NDARRAY_DEVICE_GPU`
 * compile, link and run against include/numpower_host.h + ext/hip_math.h exactly as they are written
 * in the reference — the same entry-point names, the same `cuda_float_*` function POINTERS handed to the
 * drivers, the same ownership (every result and every input is released with NDArray_FREE):
 *
 *   unary maths          rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_sin);      numpower.c:1636-1660 ...
 *   clip / round         NDArrayMathGPU_ElementWise2F / 1F                           numpower.c (clip, round)
 *   arctan2              NDArrayMathGPU_ElementWise1N(ndx, cuda_float_arctan2, ndy)
 *   add ... pow          if (!NDArray_IsBroadcastable(nda, ndb)) throw; rtn = NDArray_Add_Float(nda, ndb);
 *                                                                                    numpower.c:3364-3389
 *   sum / prod           reduce(nda, &axis_i, NDArray_Add_Float) | NDArray_Sum_Float numpower.c:4620-4645
 *   matmul / dot         NDArray_Matmul(nda, ndb), NDArray_Dot(nda, ndb)
 *   sharded batched matmul  NDArray_CommInit(rank, world, endpoint); NDArray_ShardedBatchedMatmul(a, b, batch, mode)
 *                        (this project's own extension, SURVEY.md section 8e: the reference has no multi-device
 *                        code; run here as a world of one rank — kept, gathered, in 3 overlapped pieces, and with the piece
 *                        count left to the library)
 *   gpu() / cpu()        NDArray_ToGPU / NDArray_ToCPU
 *
 * Zend argument parsing (zval -> NDArray*) and RETURN_NDARRAY are the only parts of a method that are
 * not here; they do not touch the device.  The program writes every result to a file which
 * tests/test_gpu_method_bodies.py checks against the oracle — compiled with `gcc -std=c99 -Wall
 * -Wextra -Werror`, so a signature that drifts from the reference's call sites fails the BUILD.
 *
 * Usage: method_bodies <output file>
 */
#define _POSIX_C_SOURCE 200809L   /* getpid() under -std=c99 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "numpower_host.h"
#include "hip_math.h"

static FILE *g_out;
static int g_failed;

/* record: 32-byte name, int32 ndim, int32 dims[4], then the floats (row-major) */
static void dump(const char *name, NDArray *a) {
    if (a == NULL) {
        fprintf(stderr, "method_bodies: %s returned NULL: %s\n", name, numpower_host_last_error());
        g_failed = 1;
        return;
    }
    char label[32];
    int32_t head[5] = {NDArray_NDIM(a), 1, 1, 1, 1};
    memset(label, 0, sizeof label);
    strncpy(label, name, sizeof label - 1);
    for (int i = 0; i < NDArray_NDIM(a) && i < 4; i++) head[1 + i] = NDArray_SHAPE(a)[i];
    long n = NDArray_NUMELEMENTS(a);
    float *host = (float *) malloc(sizeof(float) * (size_t) (n > 0 ? n : 1));
    /* $rtn->cpu()->toArray(): toArray refuses device arrays (numpower.c:466), as in the reference */
    NDArray *on_host = NDArray_DEVICE(a) == NDARRAY_DEVICE_GPU ? NDArray_ToCPU(a) : NULL;
    if (host == NULL || NDArray_CopyToHostBuffer(on_host ? on_host : a, host) != 0) {
        fprintf(stderr, "method_bodies: reading %s back failed: %s\n", name, numpower_host_last_error());
        g_failed = 1;
    } else {
        fwrite(label, 1, sizeof label, g_out);
        fwrite(head, sizeof(int32_t), 5, g_out);
        fwrite(host, sizeof(float), (size_t) n, g_out);
    }
    if (on_host) NDArray_FREE(on_host);
    free(host);
}

static void dump_scalar(const char *name, double value) {
    char label[32];
    int32_t head[5] = {0, 1, 1, 1, 1};
    float v = (float) value;
    memset(label, 0, sizeof label);
    strncpy(label, name, sizeof label - 1);
    fwrite(label, 1, sizeof label, g_out);
    fwrite(head, sizeof(int32_t), 5, g_out);
    fwrite(&v, sizeof v, 1, g_out);
}

/* deterministic inputs the test can rebuild: x[i] = lo + (hi - lo) * frac(i * 0.6180339887 + seed * 0.37) */
static NDArray *input(int rows, int cols, int seed, float lo, float hi) {   /* rows == 0: a 1-d array of cols */
    int shape[2] = {rows, cols};
    long n = (long) (rows ? rows : 1) * cols;
    float *host = (float *) malloc(sizeof(float) * (size_t) n);
    for (long i = 0; i < n; i++) {
        double t = (double) i * 0.6180339887 + (double) seed * 0.37;
        t -= (double) (long) t;
        host[i] = (float) ((double) lo + ((double) hi - (double) lo) * t);
    }
    NDArray *cpu = rows ? NDArray_FromHostBuffer(host, shape, 2) : NDArray_FromHostBuffer(host, shape + 1, 1);
    free(host);
    return cpu;
}

/* $a->gpu(); a method returns to PHP when this throws, the program stops */
static NDArray *to_gpu(NDArray *host) {
    NDArray *dev = host ? NDArray_ToGPU(host) : NULL;
    if (dev == NULL) {
        fprintf(stderr, "method_bodies: gpu() failed: %s\n", numpower_host_last_error());
        exit(1);
    }
    return dev;
}

/* ---- the method bodies ---------------------------------------------------------------------- */

typedef struct {
    const char *name;
    ElementWiseFloatGPUOperation op;
    float lo, hi;
} UnaryMethod;

/* every unary PHP_METHOD that hands a cuda_float_* pointer to NDArrayMathGPU_ElementWise */
static const UnaryMethod kUnary[] = {
    {"sin", cuda_float_sin, -10, 10},          {"cos", cuda_float_cos, -10, 10},
    {"tan", cuda_float_tan, -1.4f, 1.4f},      {"arcsin", cuda_float_arcsin, -1, 1},
    {"arccos", cuda_float_arccos, -1, 1},      {"arctan", cuda_float_arctan, -10, 10},
    {"sinh", cuda_float_sinh, -8, 8},          {"cosh", cuda_float_cosh, -8, 8},
    {"tanh", cuda_float_tanh, -8, 8},          {"arcsinh", cuda_float_arcsinh, -10, 10},
    {"arccosh", cuda_float_arccosh, 1, 20},    {"arctanh", cuda_float_arctanh, -0.95f, 0.95f},
    {"exp", cuda_float_exp, -10, 10},          {"expm1", cuda_float_expm1, -5, 5},
    {"log", cuda_float_log, 0.01f, 100},       {"log2", cuda_float_log2, 0.01f, 100},
    {"log10", cuda_float_log10, 0.01f, 100},   {"log1p", cuda_float_log1p, -0.9f, 50},
    {"logb", cuda_float_logb, 0.01f, 100},     {"sqrt", cuda_float_sqrt, 0, 100},
    {"reciprocal", cuda_float_reciprocal, 0.1f, 10}, {"negate", cuda_float_negate, -10, 10},
    {"positive", cuda_float_positive, -10, 10}, {"sign", cuda_float_sign, -10, 10},
    {"floor", cuda_float_floor, -10, 10},      {"ceil", cuda_float_ceil, -10, 10},
    {"trunc", cuda_float_trunc, -10, 10},      {"fix", cuda_float_fix, -10, 10},
    {"rint", cuda_float_rint, -10, 10},        {"radians", cuda_float_radians, -360, 360},
    {"degrees", cuda_float_degrees, -7, 7},    {"sinc", cuda_float_sinc, -5, 5},
    /* the two methods tools/apply_with_hip.py re-points (numpower.c:1791 passes cuda_float_arccos for rsqrt,
     * :3153 has no device branch for exp2) */
    {"rsqrt", cuda_float_rsqrt, 0.01f, 100},   {"exp2", cuda_float_exp2, -10, 10},
};

static NDArray *method_unary(NDArray *nda, ElementWiseFloatGPUOperation op) {
    NDArray *rtn = NULL;
    if (NDArray_DEVICE(nda) == NDARRAY_DEVICE_GPU) {
        rtn = NDArrayMathGPU_ElementWise(nda, op);
    }
    return rtn;
}

static NDArray *method_clip(NDArray *nda, double min, double max) {
    return NDArrayMathGPU_ElementWise2F(nda, cuda_float_clip, (float) min, (float) max);
}

static NDArray *method_round(NDArray *nda, long precision) {
    return NDArrayMathGPU_ElementWise1F(nda, cuda_float_round, (float) precision);
}

static NDArray *method_arctan2(NDArray *ndx, NDArray *ndy) {
    return NDArrayMathGPU_ElementWise1N(ndx, cuda_float_arctan2, ndy);
}

typedef NDArray *(*BinaryFloat)(NDArray *, NDArray *);

static NDArray *method_binary(NDArray *nda, NDArray *ndb, BinaryFloat fn) {
    if (!NDArray_IsBroadcastable(nda, ndb)) {
        fprintf(stderr, "method_bodies: not broadcastable\n");
        g_failed = 1;
        return NULL;
    }
    return fn(nda, ndb);
}

int main(int argc, char **argv) {
    if (argc != 2) {
        fprintf(stderr, "usage: %s <output file>\n", argv[0]);
        return 2;
    }
    g_out = fopen(argv[1], "wb");
    if (g_out == NULL) {
        perror(argv[1]);
        return 2;
    }
    const int rows = 257, cols = 255;   /* body + ragged tail for every vector width */

    /* $x->gpu(); NDArray::sin($x) ... ; every result freed, as RETURN_NDARRAY's owner would */
    for (size_t i = 0; i < sizeof kUnary / sizeof kUnary[0]; i++) {
        NDArray *host = input(rows, cols, (int) i, kUnary[i].lo, kUnary[i].hi);
        NDArray *nda = to_gpu(host);
        NDArray *rtn = method_unary(nda, kUnary[i].op);
        dump(kUnary[i].name, rtn);
        if (rtn) NDArray_FREE(rtn);
        NDArray_FREE(nda);
        NDArray_FREE(host);
    }

    NDArray *hx = input(rows, cols, 101, -50, 50), *hy = input(rows, cols, 102, -50, 50);
    NDArray *x = to_gpu(hx), *y = to_gpu(hy);
    NDArray *r;
    r = method_clip(x, -7.25, 11.5);   dump("clip", r);    if (r) NDArray_FREE(r);
    r = method_round(x, 2);            dump("round", r);   if (r) NDArray_FREE(r);
    r = method_arctan2(x, y);          dump("arctan2", r); if (r) NDArray_FREE(r);

    /* binary arithmetic: same shape, a row operand, a 0-d scalar operand */
    static const struct { const char *name; BinaryFloat fn; } kBinary[] = {
        {"add", NDArray_Add_Float}, {"subtract", NDArray_Subtract_Float}, {"multiply", NDArray_Multiply_Float},
        {"divide", NDArray_Divide_Float}, {"mod", NDArray_Mod_Float},
    };
    NDArray *hrow = input(0, cols, 103, 0.5f, 4), *row = to_gpu(hrow);
    NDArray *two = NDArray_CreateFromDoubleScalar(2.5);
    for (size_t i = 0; i < sizeof kBinary / sizeof kBinary[0]; i++) {
        char label[32];
        r = method_binary(x, y, kBinary[i].fn);
        dump(kBinary[i].name, r);
        if (r) NDArray_FREE(r);
        snprintf(label, sizeof label, "%s_row", kBinary[i].name);
        r = method_binary(x, row, kBinary[i].fn);
        dump(label, r);
        if (r) NDArray_FREE(r);
        snprintf(label, sizeof label, "%s_scalar", kBinary[i].name);
        r = method_binary(x, two, kBinary[i].fn);
        dump(label, r);
        if (r) NDArray_FREE(r);
    }
    NDArray *hp = input(rows, cols, 104, 0.25f, 4), *p = to_gpu(hp);
    r = method_binary(p, row, NDArray_Pow_Float);
    dump("pow_row", r);
    if (r) NDArray_FREE(r);

    /* NDArray::sum($x) / sum($x, axis) / prod($p, axis): numpower.c:4620-4645 */
    dump_scalar("sum", (double) NDArray_Sum_Float(x));
    for (int axis_i = 0; axis_i < 2; axis_i++) {
        char label[32];
        snprintf(label, sizeof label, "sum_axis%d", axis_i);
        r = reduce(x, &axis_i, NDArray_Add_Float);
        dump(label, r);
        if (r) NDArray_FREE(r);
    }
    dump_scalar("min", (double) NDArray_Min(x));
    dump_scalar("max", (double) NDArray_Max(x));

    /* NDArray::matmul / dot */
    NDArray *hb = input(cols, 129, 105, -1, 1), *b = to_gpu(hb);
    r = NDArray_Matmul(x, b);
    dump("matmul", r);
    if (r) NDArray_FREE(r);
    r = NDArray_Dot(x, b);
    dump("dot", r);
    if (r) NDArray_FREE(r);

    /* the sharded batched matmul as a PHP method would issue it (world of one: the communicator, the second stream
     * and the chunk pipeline all run; the transfers have no peer to go to) */
    {
        int shape3[3] = {6, 33, 47}, shape3b[3] = {6, 47, 29};
        NDArray *h3a = input(6 * 33, 47, 106, -1, 1), *h3b = input(6 * 47, 29, 107, -1, 1);
        NDArray *ha3 = NDArray_FromHostBuffer(NDArray_FDATA(h3a), shape3, 3), *hb3 = NDArray_FromHostBuffer(NDArray_FDATA(h3b), shape3b, 3);
        NDArray *a3 = to_gpu(ha3), *b3 = to_gpu(hb3);
        char endpoint[64];
        snprintf(endpoint, sizeof endpoint, "/tmp/np_method_bodies_%ld.id", (long) getpid());
        if (NDArray_CommInit(0, 1, endpoint) != 0) {
            fprintf(stderr, "method_bodies: NDArray_CommInit failed: %s\n", numpower_host_last_error());
            g_failed = 1;
        } else {
            static const struct { const char *name; int mode; } kShard[] = {
                {"sharded_keep", NP_SHARD_KEEP}, {"sharded_gather", NP_SHARD_GATHER}, {"sharded_overlap3", 3},
                {"sharded_overlap_auto", NP_SHARD_OVERLAP}};
            for (size_t i = 0; i < sizeof kShard / sizeof kShard[0]; i++) {
                r = NDArray_ShardedBatchedMatmul(a3, b3, 6 * NDArray_CommWorld(), kShard[i].mode);
                dump(kShard[i].name, r);
                if (r) NDArray_FREE(r);
            }
            r = NDArray_ShardedBatchedMatmul(a3, b3, 7, NP_SHARD_GATHER);   /* shares that do not add up: thrown, NULL */
            if (r != NULL || strstr(numpower_host_last_error(), "Batch of 7 is not 1 slab(s) of 6") == NULL) {
                fprintf(stderr, "method_bodies: bad batch was not refused (%s)\n", numpower_host_last_error());
                g_failed = 1;
            }
            if (NDArray_CommDestroy() != 0) g_failed = 1;
        }
        NDArray_FREE(a3);  NDArray_FREE(ha3);  NDArray_FREE(h3a);
        NDArray_FREE(b3);  NDArray_FREE(hb3);  NDArray_FREE(h3b);
    }

    /* $r->cpu(): a device result brought back as a host NDArray */
    r = method_unary(x, cuda_float_exp);
    if (r) {
        NDArray *back = NDArray_ToCPU(r);
        dump("exp_cpu", back);
        if (back) NDArray_FREE(back);
        NDArray_FREE(r);
    }

    NDArray_FREE(b);   NDArray_FREE(hb);
    NDArray_FREE(p);   NDArray_FREE(hp);
    NDArray_FREE(two);
    NDArray_FREE(row); NDArray_FREE(hrow);
    NDArray_FREE(y);   NDArray_FREE(hy);
    NDArray_FREE(x);   NDArray_FREE(hx);
    fclose(g_out);
    if (NDArray_LiveDeviceAllocations() != 0) {
        fprintf(stderr, "method_bodies: %ld device allocations leaked\n", NDArray_LiveDeviceAllocations());
        return 1;
    }
    return g_failed;
}
