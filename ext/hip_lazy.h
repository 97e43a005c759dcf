/*
 * hip_lazy.h — pending elementwise chains behind the NDArray handle.
 *
 * In the reference every PHP-level op — `nd::exp($a)`, `* $b`, `+ 2` — allocates a result and makes a full round trip
 * through memory (ndarray_do_operation_ex, numpower.c:193-229; the unary PHP_METHODs, numpower.c:1608-3357).  With the edits
 * of tools/apply_with_hip.py section 2c the arithmetic operators, the six static arithmetic methods and the unary family
 * APPEND when an operand is a GPU array: the object they return is a real NDArray — shape, device, a buffer from the pool —
 * whose VALUES have not been computed yet; a side table (this file) holds the chain of steps that produces them.  The next
 * appender that meets such an array extends a COPY of its chain; the PHP temporary in between dies unevaluated and costs
 * nothing.  `nd::exp($a) * $b + 2` is ONE np_fused_chain launch (12 B/elem) instead of three launches (28 B/elem),
 * bit-identical to the three-launch form (same op bodies, same AVX-body quirk flags).
 *
 * Who computes the values ("flush"): buffer_get() (src/buffer.c:80) — the ONE function through which a PHP handle becomes
 * an NDArray* (ZVAL_TO_NDARRAY numpower.c:105, ARRAY_OF_NDARRAYS :166, ZVALUUID_TO_NDARRAY :332-335, print_r_ :686,695) —
 * calls NPH_OnBufferGet(), so every consumer that is not an appender (toArray, cpu(), reductions, matmul, comparisons,
 * slicing, printing, iteration ...) sees finished values without knowing about chains: the flush set is closed by
 * construction, not by enumerating methods.  Appenders look their operands up between NPH_LAZY_MARSHAL_BEGIN / _END, which
 * turns that flush off for the lookup.  NDArray_FREE (src/ndarray.c:587) calls NPH_OnFree() so that an array that dies with
 * a pending chain releases the chain's inputs.
 *
 * Mutation: a chain READS its inputs at flush time, so the values must still be the ones the expression saw.  Every
 * non-appender access to an array (any buffer_get, which is what fill / offsetSet / in-place methods go through) first
 * flushes the chains that read that array's buffer — through views too (the root of the `base` links is compared).
 *
 * Threads: the table of pending chains and the appender-scope counter are process-wide, like the reference's MAIN_MEM_STACK
 * (src/buffer.c:13; PHP NTS: one request = one thread) — not for concurrent appenders.  Readers that find nothing pending
 * (NPH_OnFree / NPH_OnBufferGet with an empty table) touch one counter and return.
 *
 * Scope (what numpower_amd/lazy.py defines): linear chains acc = f_k(... f_1(x)) of at most NPH_MAX_OPS unary / binary
 * steps over at most NPH_MAX_INPUTS arrays; a binary step takes another GPU array of the chain's shape, a smaller one that
 * broadcasts onto it (row vector, column, 0-d) or a number.  Anything else — an operand that is itself pending (it is
 * flushed first and joins as an array), a chain that is full, an operand that would have to grow the chain's shape,
 * CPU operands — takes the eager path of section 2b unchanged.  Every step, pow and `** 2` included, runs the arithmetic of
 * its stand-alone kernel: a value does not depend on how it came to be evaluated.
 */
#ifndef NUMPOWER_AMD_EXT_HIP_LAZY_H
#define NUMPOWER_AMD_EXT_HIP_LAZY_H

#include <stddef.h>

#include "np_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

struct NDArray;

#define NPH_MAX_OPS 12
#define NPH_MAX_INPUTS 6
#define NPH_MAX_PENDING 64

/* > 0 while an appender looks its operands up: NPH_OnBufferGet() leaves pending arrays pending */
extern int nph_marshal_lazy;
#define NPH_LAZY_MARSHAL_BEGIN() (nph_marshal_lazy++)
#define NPH_LAZY_MARSHAL_END() (nph_marshal_lazy--)

/* buffer_get(): `a` is about to be handed to a consumer.  Computes its values if they are pending and the values of every
 * pending array whose chain reads a's buffer.  Errors are raised through the host (zend_throw_error); the array is
 * returned to the consumer either way. */
void NPH_OnBufferGet(struct NDArray *a);
/* NDArray_FREE(): one reference to `a` is about to be dropped; when it is the last one a pending chain is discarded. */
void NPH_OnFree(struct NDArray *a);

/* The appenders.  `eager` = the reference function of the operator (NDArray_Add_Float ...), called when the chain cannot
 * take the step — with every pending operand flushed first — and for CPU operands.  NULL + a raised error on failure. */
typedef struct NDArray *(*NPH_EagerBinary)(struct NDArray *, struct NDArray *);
struct NDArray *NPH_LazyBinary(int op, NPH_EagerBinary eager, struct NDArray *a, struct NDArray *b);
/* the unary family with the reference's driver signatures (cuda_math.h:14-15,75-76): `rtn = NDArrayMathGPU_ElementWise(nda,
 * cuda_float_sin)` becomes `rtn = NPH_LazyElementWise(nda, cuda_float_sin)`; a function pointer that is not one of
 * hip_math.c's goes to the driver itself.  (The pointer types are cuda_math.h:10-13's ElementWiseFloatGPUOperation{,1F,2F}
 * under names of their own: this header is included next to the reference's, which typedefs those.) */
typedef void (*NPH_UnaryFn)(int, float *);
typedef void (*NPH_Unary1FFn)(int, float *, float);
typedef void (*NPH_Unary2FFn)(int, float *, float, float);
struct NDArray *NPH_LazyElementWise(struct NDArray *a, NPH_UnaryFn op);
struct NDArray *NPH_LazyElementWise1F(struct NDArray *a, NPH_Unary1FFn op, float val1);
struct NDArray *NPH_LazyElementWise2F(struct NDArray *a, NPH_Unary2FFn op, float val1, float val2);

/* The full reductions as consumers that know about chains: `double value = NDArray_Sum_Float(nda);` of PHP_METHOD(sum)
 * (numpower.c:4638; likewise prod :4744, min :4673, max :4712 and the two `NDArray_Sum_Float(nda) / NDArray_NUMELEMENTS(nda)` of
 * PHP_METHOD(mean) :2660,2675) becomes `NPH_ReduceAll(NP_SUM, NDArray_Sum_Float, nda)`.  A pending operand is reduced INSIDE its
 * chain's kernel (np_fused_chain_reduce: the expression's values never go to memory — nd::sum(nd::exp($a) * $b) reads 8 B/elem in
 * one launch instead of writing 4, reading 4 more and launching twice) and STAYS pending; anything else goes to `eager`, the
 * reference function.  min / max are bit-identical either way; sum / prod are folded in the chain kernel's (deterministic) order,
 * which is not the order np_reduce_all folds stored values in: equal within the bar every reduction is held to (1e-5 of fp64),
 * not bit for bit.  Returns what `eager` returns on failure: a raised error and -1. */
typedef float (*NPH_EagerReduce)(struct NDArray *);
float NPH_ReduceAll(int reduce_op, NPH_EagerReduce eager, struct NDArray *a);

/* reduce() / single_reduce() (src/ndarray.c:570, :509) met a pending operand: `rtn` (allocated by the caller with the reduced
 * shape) = reduction of the chain's values over `axis`, INSIDE the chain's kernel (np_fused_chain_reduce_axis: nd::sum(nd::exp($x), 1)
 * reads 4 B/elem once instead of writing 4 and reading 4 more) — for the last axis of any array and the first axis of a 2-d array,
 * reduce_op NP_SUM / NP_MEAN / NP_MIN / NP_MAX (a product carries the zero-sign quirks of the reference's slice-by-slice multiply: it
 * reduces stored values).  1 = done (the operand stays pending), 0 = not this way (values there, another axis, a product: the caller
 * flushes and reduces the stored values), -1 = an error was raised. */
int NPH_ChainReduceAxisInto(struct NDArray *array, int axis, int reduce_op, struct NDArray *rtn);
/* the rows x cols view np_fused_chain_reduce_axis needs for `axis` of an array shaped like `first`, given the chain's own 2-d view
 * (call->rows x call->cols, fixed by its broadcast operands): 1 + *rows, *cols, *ax (0 = first axis, 1 = last) or 0 = no such view */
struct NPH_ChainCall;
int NPH_ChainAxisView(const struct NDArray *first, int axis, const struct NPH_ChainCall *call, size_t *rows, size_t *cols, int *ax);

/* PHP_RINIT (numpower.c:5250, next to buffer_init): a request starts with no pending chain and with the appender scope closed,
 * whatever the previous request of this process left behind — a fatal error inside ZVAL_TO_NDARRAY (memory limit) bails out with
 * longjmp past NPH_LAZY_MARSHAL_END, and Zend has by then released every array the table still names.  Nothing is dereferenced. */
void NPH_RequestInit(void);

/* 0 = values are there (or were computed now), -1 = an error was raised */
int NPH_Flush(struct NDArray *a);
int NPH_IsPending(const struct NDArray *a);
int NPH_PendingCount(void);
/* on (default) / off: off makes every appender take the eager path (one launch per op, as section 2b alone) */
void NPH_SetLazy(int on);
/* counters for tests and the demo program: chains flushed as one launch, steps those chains held, chains discarded
 * unevaluated, steps that took the eager path although an operand was on the GPU, reductions run inside a chain's kernel */
typedef struct NPH_LazyStats {
    unsigned long flushed_chains, flushed_steps, discarded_chains, eager_steps, fused_reductions;
} NPH_LazyStats;
void NPH_GetLazyStats(NPH_LazyStats *out);

/* One fused launch for a chain given as arrays + steps — what a flush runs, and what NDArray_FusedChain
 * (include/numpower_host.h) is: inputs[i] == NULL stands for a host number, scalars[i]; ops[k].flags / body_end are
 * filled in here (the AVX-body quirks of the stand-alone entry points, so the result is bit-identical to op-by-op).
 * prepare: classification only; -1 + a raised error if the chain is not expressible (the same messages as the eager path). */
typedef struct NPH_ChainCall {
    const float *ptrs[16];
    int kinds[16];
    np_fused_op prog[64];
    size_t rows, cols;
} NPH_ChainCall;
int NPH_PrepareChain(struct NDArray **inputs, const float *scalars, int n_inputs, const np_fused_op *ops, int n_ops,
                     NPH_ChainCall *call);

#ifdef __cplusplus
}
#endif
#endif
