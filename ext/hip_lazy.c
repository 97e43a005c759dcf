/*
 * hip_lazy.c — pending elementwise chains behind the NDArray handle (hip_lazy.h).
 *
 *   NPH_LazyBinary          the `rtn = NDArray_Add_Float(nda, ndb);` ... of ndarray_do_operation_ex (numpower.c:200-218) and of
 *                           PHP_METHOD(add / subtract / multiply / divide / mod / pow) (numpower.c:3384-3550)
 *   NPH_LazyElementWise*    the `rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_sin);` ... of the unary PHP_METHODs
 *                           (numpower.c:1651-3348; clip :2487, round :2959)
 *   NPH_ReduceAll           the `NDArray_Sum_Float(nda)` / Float_Prod / Min / Max of PHP_METHOD(sum, prod, min, max, mean)
 *                           (numpower.c:4638,4744,4673,4712,2660,2675): a pending operand is reduced inside its chain's kernel
 *   NPH_ChainReduceAxisInto reduce() / single_reduce() (src/ndarray.c:570, :509) on a pending operand: sum / mean over the last axis (or
 *                           the first of a 2-d array) inside the chain's kernel
 *   NPH_OnBufferGet         buffer_get (src/buffer.c:80-83): the flush point
 *   NPH_OnFree              NDArray_FREE (src/ndarray.c:587-592): a pending array that dies releases its inputs
 *   NPH_PrepareChain        operand kinds + AVX-body quirk flags of a chain (shared with NDArray_FusedChain of the host
 *                           library, which numpower_amd/lazy.py drives)
 *
 * Plain C over include/np_hip.h, compiled like hip_fast.c: into a `--with-hip` NumPower tree against the reference's own
 * headers (results are the reference's NDArray_EmptyLike arrays and die in its NDArray_FREE), and into libnumpower_host.so
 * against include/numpower_host.h, which is how the GPU test tier runs every line (numpower_amd/lib/lazy_bodies: the text
 * section 2c inserts, verbatim, around a stand-in for the Zend object table).  No CPU arithmetic.
 */
#ifndef NUMPOWER_NDARRAY_HEADER
#define NUMPOWER_NDARRAY_HEADER "numpower_host.h"
#endif
#include NUMPOWER_NDARRAY_HEADER

#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "hip_fast.h"
#include "hip_lazy.h"
#include "hip_math.h"
#include "np_ext_hooks.h"
#include "np_hip.h"

/* the reference's drivers (src/ndmath/cuda/cuda_math.h:14-15,75-76; ext/hip_math_drivers.c here) */
NDArray *NDArrayMathGPU_ElementWise(NDArray *ndarray, ElementWiseFloatGPUOperation op);
NDArray *NDArrayMathGPU_ElementWise1F(NDArray *ndarray, ElementWiseFloatGPUOperation1F op, float val1);
NDArray *NDArrayMathGPU_ElementWise2F(NDArray *ndarray, ElementWiseFloatGPUOperation2F op, float val1, float val2);

typedef struct Chain {
    NDArray *self;                        /* the array whose values this chain produces; NULL = free slot */
    int n_inputs, n_ops;
    NDArray *inputs[NPH_MAX_INPUTS];      /* inputs[0] = where the chain starts; NULL = a host number (scalars[i]) */
    float scalars[NPH_MAX_INPUTS];
    np_fused_op ops[NPH_MAX_OPS];
    size_t rows, cols;                    /* the 2-D view fixed by the first row / column operand */
    int have_2d;
} Chain;

static Chain g_chain[NPH_MAX_PENDING];
static int g_pending;
static int g_lazy_on = 1;
static NPH_LazyStats g_stats;
int nph_marshal_lazy = 0;

static Chain *find_chain(const NDArray *a) {
    if (g_pending == 0 || a == NULL) return NULL;
    for (int i = 0; i < NPH_MAX_PENDING; i++)
        if (g_chain[i].self == a) return &g_chain[i];
    return NULL;
}

static const NDArray *root_of(const NDArray *a) {
    while (a->base != NULL) a = a->base;
    return a;
}

static void raise(const char *fmt, int i) {
    char msg[160];
    snprintf(msg, sizeof msg, fmt, i);
    np_ext_throw(msg);
}

void NPH_RequestInit(void) {
    memset(g_chain, 0, sizeof g_chain);      /* (the arrays these named belonged to a request that is over: not touched) */
    g_pending = 0;
    nph_marshal_lazy = 0;
}

int NPH_IsPending(const NDArray *a) { return find_chain(a) != NULL; }
int NPH_PendingCount(void) { return g_pending; }
void NPH_SetLazy(int on) { g_lazy_on = on != 0; }
void NPH_GetLazyStats(NPH_LazyStats *out) {
    if (out != NULL) *out = g_stats;
}

int NPH_PrepareChain(NDArray **inputs, const float *scalars, int n_inputs, const np_fused_op *ops, int n_ops, NPH_ChainCall *c) {
    if (inputs == NULL || n_inputs < 1 || inputs[0] == NULL) {
        np_ext_throw("fused chain: no input array");
        return -1;
    }
    if (ops == NULL && n_ops > 0) {
        np_ext_throw("fused chain: null op list");
        return -1;
    }
    NDArray *first = inputs[0];
    if (NDArray_NDIM(first) == 0) {
        np_ext_throw("fused chain must start from an array");
        return -1;
    }
    if (NDArray_DEVICE(first) != NDARRAY_DEVICE_GPU) {
        np_ext_throw("fused elementwise chain: operand is on the CPU; numpower_amd only computes on the GPU "
                     "(call ->gpu() first, the CPU path is the reference's own)");
        return -1;
    }
    if (n_inputs > 16 || n_ops > 64) {
        np_ext_throw("fused chain too long");
        return -1;
    }
    const long n = NDArray_NUMELEMENTS(first);
    size_t rows = 1, cols = (size_t) n;
    int have_2d = 0;
    for (int i = 0; i < n_inputs; ++i) {
        NDArray *x = inputs[i];
        if (x == NULL) {
            if (scalars == NULL) {
                raise("fused chain: input %d is null", i);
                return -1;
            }
            c->kinds[i] = NP_HOST_SCALAR;                      /* a PHP number, kept by value */
            c->ptrs[i] = &scalars[i];
            continue;
        }
        if (NDArray_NDIM(x) == 0 && NDArray_DEVICE(x) == NDARRAY_DEVICE_CPU) {
            c->kinds[i] = NP_HOST_SCALAR;
        } else {
            if (NDArray_DEVICE(x) != NDARRAY_DEVICE_GPU) {
                np_ext_throw("Device mismatch, both NDArray MUST be in the same device.");
                return -1;
            }
            if (NDArray_NDIM(x) == 0) {
                c->kinds[i] = NP_SCALAR;
            } else if (NDArray_NUMELEMENTS(x) == n) {
                c->kinds[i] = NP_FULL;                         /* equal element counts: flat elementwise (arithmetics.c:194-197) */
            } else if (NDArray_NUMELEMENTS(x) < n) {
                size_t br = 1, bc = 1;
                const int k = NPH_BroadcastKind(x, first, &br, &bc);
                if (k < 0 || (have_2d && (br != rows || bc != cols))) {
                    np_ext_throw("Can't broadcast arrays.");
                    return -1;
                }
                c->kinds[i] = k;
                rows = br;
                cols = bc;
                have_2d = 1;
            } else {
                np_ext_throw("Can't broadcast arrays.");       /* the accumulator itself would have to grow: not a fused case */
                return -1;
            }
        }
        c->ptrs[i] = NDArray_FDATA(x);
    }
    if (c->kinds[0] != NP_FULL) {
        np_ext_throw("fused chain must start from an array");
        return -1;
    }
    for (int k = 0; k < n_ops; ++k) {
        c->prog[k] = ops[k];
        c->prog[k].flags = 0;
        c->prog[k].body_end = 0;
        if (ops[k].kind == NP_FUSED_BINARY) {
            const int op = ops[k].op;
            if (ops[k].operand < 0 || ops[k].operand >= n_inputs) {
                np_ext_throw("fused chain: operand index out of range");
                return -1;
            }
            if (op == NP_MULTIPLY || op == NP_MOD || op == NP_EQUAL || op == NP_NOT_EQUAL) {
                /* AVX-body bound: element count of the FIRST operand after the scalar expand but before the broadcast
                 * (arithmetics.c:251, logic.c:535); NotEqual loops over the broadcast operand (logic.c:636) */
                const NDArray *other = inputs[ops[k].operand];
                size_t loop_numel_a = (size_t) n;
                if (ops[k].swap && other != NULL && NDArray_NDIM(other) != 0 && op != NP_NOT_EQUAL)
                    loop_numel_a = (size_t) NDArray_NUMELEMENTS(other);
                c->prog[k].flags = NP_QUIRK_AVX_BODY;
                c->prog[k].body_end = np_avx_body_end(loop_numel_a);
            }
        }
    }
    c->rows = rows;
    c->cols = cols;
    return 0;
}

static void release_chain(Chain *c) {
    NDArray *held[NPH_MAX_INPUTS];
    const int n = c->n_inputs;
    for (int i = 0; i < n; i++) held[i] = c->inputs[i];
    c->self = NULL;                                            /* the slot is free BEFORE the inputs go: NDArray_FREE comes back */
    c->n_inputs = c->n_ops = 0;                                /* through NPH_OnFree */
    g_pending--;
    for (int i = 0; i < n; i++)
        if (held[i] != NULL) NDArray_FREE(held[i]);
}

int NPH_Flush(NDArray *a) {
    Chain *c = find_chain(a);
    if (c == NULL) return 0;
    NPH_ChainCall call;
    int rc = NPH_PrepareChain(c->inputs, c->scalars, c->n_inputs, c->ops, c->n_ops, &call);
    if (rc == 0) {
        float *out = NDArray_FDATA(a);
        int st;
        if (c->n_ops == 1 && c->ops[0].kind == NP_FUSED_UNARY) {
            /* a chain of one step IS the stand-alone launch: $c = nd::exp($a) costs what it cost without chains */
            st = np_unary(c->ops[0].op, call.ptrs[0], out, (size_t) NDArray_NUMELEMENTS(a), c->ops[0].p0, c->ops[0].p1);
        } else if (c->n_ops == 1) {
            const int k = c->ops[0].operand, swap = c->ops[0].swap;
            st = np_binary(c->ops[0].op, call.ptrs[swap ? k : 0], call.kinds[swap ? k : 0], call.ptrs[swap ? 0 : k],
                           call.kinds[swap ? 0 : k], out, call.rows, call.cols, call.prog[0].flags, call.prog[0].body_end);
        } else {
            st = np_fused_chain(call.ptrs, call.kinds, c->n_inputs, call.prog, c->n_ops, out, call.rows, call.cols);
        }
        if (st != NP_OK) {
            np_ext_throw_last();
            rc = -1;
        } else {
            g_stats.flushed_chains++;
            g_stats.flushed_steps += (unsigned long) c->n_ops;
        }
    }
    release_chain(c);
    return rc;
}

float NPH_ReduceAll(int reduce_op, NPH_EagerReduce eager, NDArray *a) {
    if (a == NULL || eager == NULL) return -1.0f;
    Chain *c = find_chain(a);
    if (c == NULL) {
        /* values are there.  A GPU array goes straight to np_reduce_all with its size_t count: the reference's own function
         * hands NDArray_NUMELEMENTS to the `int` of cuda_sum_float / cuda_prod_float / cuda_min_float / cuda_max_float
         * (arithmetics.c:41,63,86, ndarray.c:759,946) — same kernel, same value, no 2^31 limit.  CPU arrays: the reference. */
        if (NDArray_NDIM(a) != 0 && NDArray_DEVICE(a) == NDARRAY_DEVICE_GPU && reduce_op >= 0 && reduce_op < NP_REDUCE_OP_COUNT) {
            float r = 0.0f;
            if (np_reduce_all(reduce_op == NP_MEAN ? NP_SUM : reduce_op, NDArray_FDATA(a), (size_t) NDArray_NUMELEMENTS(a), &r) != NP_OK) {
                np_ext_throw_last();
                return -1.0f;
            }
            return reduce_op == NP_MEAN ? r / NDArray_NUMELEMENTS(a) : r;   /* (arithmetics.c:87: float / long) */
        }
        return eager(a);
    }
    if (!g_lazy_on || reduce_op < 0 || reduce_op >= NP_REDUCE_OP_COUNT) {
        if (NPH_Flush(a) != 0) return -1.0f;
        return eager(a);
    }
    NPH_ChainCall call;
    float v = -1.0f;
    if (NPH_PrepareChain(c->inputs, c->scalars, c->n_inputs, c->ops, c->n_ops, &call) != 0) return -1.0f;
    if (np_fused_chain_reduce(call.ptrs, call.kinds, c->n_inputs, call.prog, c->n_ops, reduce_op, call.rows, call.cols, &v) != NP_OK) {
        np_ext_throw_last();
        return -1.0f;
    }
    g_stats.fused_reductions++;
    return v;                                              /* `a` stays pending: nobody has asked for its values yet */
}

int NPH_ChainAxisView(const NDArray *first, int axis, const NPH_ChainCall *call, size_t *rows, size_t *cols, int *ax) {
    const int nd = NDArray_NDIM(first);
    const size_t n = (size_t) NDArray_NUMELEMENTS(first);
    if (nd < 1 || n == 0 || axis < 0 || axis >= nd) return 0;
    if (axis == nd - 1) {
        *cols = (size_t) NDArray_SHAPE(first)[nd - 1];
        *rows = *cols ? n / *cols : 0;
        *ax = 1;
    } else if (axis == 0 && nd == 2) {
        *rows = (size_t) NDArray_SHAPE(first)[0];
        *cols = (size_t) NDArray_SHAPE(first)[1];
        *ax = 0;
    } else {
        return 0;
    }
    const int flat_chain = call->rows == 1 && call->cols == n;             /* no broadcast operand: any rows x cols view will do */
    return flat_chain || (call->rows == *rows && call->cols == *cols);
}

int NPH_ChainReduceAxisInto(NDArray *array, int axis, int reduce_op, NDArray *rtn) {
    if (array == NULL || rtn == NULL) return 0;
    Chain *c = find_chain(array);
    if (c == NULL || !g_lazy_on) return 0;
    if (reduce_op != NP_SUM && reduce_op != NP_MEAN && reduce_op != NP_MIN && reduce_op != NP_MAX) return 0;
    if (NDArray_DEVICE(rtn) != NDARRAY_DEVICE_GPU) return 0;
    NPH_ChainCall call;
    if (NPH_PrepareChain(c->inputs, c->scalars, c->n_inputs, c->ops, c->n_ops, &call) != 0) return -1;
    size_t rows = 0, cols = 0;
    int ax = -1;
    if (!NPH_ChainAxisView(array, axis, &call, &rows, &cols, &ax)) return 0;
    if ((size_t) NDArray_NUMELEMENTS(rtn) != (ax == 1 ? rows : cols)) return 0;
    if (np_fused_chain_reduce_axis(call.ptrs, call.kinds, c->n_inputs, call.prog, c->n_ops, reduce_op, rows, cols, ax,
                                   NDArray_FDATA(rtn)) != NP_OK) {
        np_ext_throw_last();
        return -1;
    }
    g_stats.fused_reductions++;
    return 1;
}

void NPH_OnBufferGet(NDArray *a) {
    if (g_pending == 0 || nph_marshal_lazy > 0 || a == NULL) return;
    (void) NPH_Flush(a);
    /* chains that READ a's buffer (directly or through a view): the consumer may write it */
    const NDArray *root = root_of(a);
    for (int i = 0; i < NPH_MAX_PENDING && g_pending > 0; i++) {
        Chain *c = &g_chain[i];
        if (c->self == NULL) continue;
        for (int k = 0; k < c->n_inputs; k++) {
            if (c->inputs[k] != NULL && root_of(c->inputs[k]) == root) {
                (void) NPH_Flush(c->self);
                break;
            }
        }
    }
}

void NPH_OnFree(NDArray *a) {
    if (g_pending == 0 || a == NULL || a->refcount != 1) return;
    Chain *c = find_chain(a);
    if (c == NULL) return;
    g_stats.discarded_chains++;
    release_chain(c);
}

static int same_shape(const NDArray *a, const NDArray *b) {
    if (NDArray_NDIM(a) != NDArray_NDIM(b)) return 0;
    for (int i = 0; i < NDArray_NDIM(a); i++)
        if (NDArray_SHAPE(a)[i] != NDArray_SHAPE(b)[i]) return 0;
    return 1;
}

static int is_gpu_array(const NDArray *a) {
    return NDArray_NDIM(a) != 0 && NDArray_DEVICE(a) == NDARRAY_DEVICE_GPU;
}

static Chain *free_slot(void) {
    for (int i = 0; i < NPH_MAX_PENDING; i++)
        if (g_chain[i].self == NULL) return &g_chain[i];
    return NULL;
}

/* A new pending array = the chain of `head` (or `head` itself when its values are there) + one step.  `other` is the
 * step's second operand (NULL for a unary step).  -> the new array; NULL with *failed = 0 when the chain cannot take the
 * step (the caller goes eager), NULL with *failed = 1 when an error was raised. */
static NDArray *append_step(NDArray *head, NDArray *other, np_fused_op step, int *failed) {
    *failed = 0;
    if (!is_gpu_array(head)) return NULL;
    const long n = NDArray_NUMELEMENTS(head);
    size_t br = 1, bc = (size_t) n;
    int other_2d = 0;
    if (other != NULL && NDArray_NDIM(other) != 0) {
        if (NDArray_DEVICE(other) != NDARRAY_DEVICE_GPU) return NULL;          /* eager raises the device mismatch */
        if (find_chain(other) != NULL && NPH_Flush(other) != 0) {             /* a pending operand joins as an array */
            *failed = 1;
            return NULL;
        }
        const long m = NDArray_NUMELEMENTS(other);
        if (m == n) {
            if (!same_shape(other, head)) return NULL;                         /* eager: flat op, first operand's shape */
        } else if (m < n) {
            if (NPH_BroadcastKind(other, head, &br, &bc) < 0) return NULL;     /* eager raises "Can't broadcast arrays." */
            other_2d = 1;
        } else {
            return NULL;                                                        /* the chain's shape would have to grow */
        }
    }
    Chain *hc = find_chain(head);
    if (hc != NULL) {
        int need_input = other != NULL;
        if (other != NULL && NDArray_NDIM(other) != 0)
            for (int i = 0; i < hc->n_inputs; i++)
                if (hc->inputs[i] == other) need_input = 0;
        const int conflict = other_2d && hc->have_2d && (hc->rows != br || hc->cols != bc);
        if (hc->n_ops + 1 > NPH_MAX_OPS || hc->n_inputs + need_input > NPH_MAX_INPUTS || conflict) {
            if (NPH_Flush(head) != 0) {                                         /* full: its value becomes the start of a new chain */
                *failed = 1;
                return NULL;
            }
            hc = NULL;
        }
    }
    Chain *slot = free_slot();
    if (slot == NULL) return NULL;
    NDArray *result = NDArray_EmptyLike(head);             /* shape, device and a pool buffer now; values at the flush */
    if (result == NULL) {
        *failed = 1;
        return NULL;
    }
    Chain c;
    memset(&c, 0, sizeof c);
    if (hc != NULL) {
        c = *hc;
    } else {
        c.inputs[0] = head;
        c.n_inputs = 1;
        c.rows = 1;
        c.cols = (size_t) n;
    }
    c.self = result;
    if (other != NULL) {
        int at = -1;
        if (NDArray_NDIM(other) == 0 && NDArray_DEVICE(other) == NDARRAY_DEVICE_CPU) {
            /* a PHP number (ZVAL_TO_NDARRAY made a 0-d CPU array of it, CHECK_INPUT_AND_FREE frees that right after the
             * operator returns, numpower.c:119-135,222-223): kept by value */
            at = c.n_inputs++;
            c.inputs[at] = NULL;
            c.scalars[at] = NDArray_FDATA(other)[0];
        } else {
            for (int i = 0; i < c.n_inputs; i++)
                if (c.inputs[i] == other) at = i;
            if (at < 0) {
                at = c.n_inputs++;
                c.inputs[at] = other;
            }
        }
        step.operand = at;
        if (other_2d) {
            c.rows = br;
            c.cols = bc;
            c.have_2d = 1;
        }
    }
    c.ops[c.n_ops++] = step;
    for (int i = 0; i < c.n_inputs; i++)
        if (c.inputs[i] != NULL) NDArray_ADDREF(c.inputs[i]);  /* the chain reads them at the flush, whatever PHP drops meanwhile */
    *slot = c;
    g_pending++;
    return result;
}

NDArray *NPH_LazyBinary(int op, NPH_EagerBinary eager, NDArray *a, NDArray *b) {
    if (a == NULL || b == NULL || eager == NULL) return NULL;
    if (!NPH_TAKES(a, b)) return eager(a, b);              /* CPU operands: the reference's own code, nothing pending */
    const int pa = find_chain(a) != NULL, pb = find_chain(b) != NULL;
    /* pow is a step like the others: the chain runs np_binary's arithmetic for it, `** 2` with a PHP number included
     * (x * x in both, np_elementwise.hip) — an expression must not change its value with the way it is evaluated */
    if (g_lazy_on) {
        NDArray *head = NULL, *other = NULL;
        int swap = 0;
        if (pa || (!pb && is_gpu_array(a) && (NDArray_NDIM(b) == 0 || NDArray_NUMELEMENTS(b) <= NDArray_NUMELEMENTS(a)))) {
            head = a;
            other = b;
        } else if (is_gpu_array(b)) {
            head = b;
            other = a;
            swap = 1;
        }
        if (head != NULL) {
            np_fused_op step;
            memset(&step, 0, sizeof step);
            step.kind = NP_FUSED_BINARY;
            step.op = op;
            step.swap = swap;
            int failed = 0;
            NDArray *r = append_step(head, other, step, &failed);
            if (r != NULL || failed) return r;
        }
    }
    if (NPH_Flush(a) != 0 || NPH_Flush(b) != 0) return NULL;
    g_stats.eager_steps++;
    return eager(a, b);
}

/* -> a pending array; NULL with *failed = 0: not a chain step (the caller goes eager); NULL with *failed = 1: error raised */
static NDArray *lazy_unary(NDArray *a, int code, float p0, float p1, int *failed) {
    *failed = 0;
    if (a == NULL || code < 0 || !g_lazy_on || !is_gpu_array(a)) return NULL;
    np_fused_op step;
    memset(&step, 0, sizeof step);
    step.kind = NP_FUSED_UNARY;
    step.op = code;
    step.p0 = p0;
    step.p1 = p1;
    return append_step(a, NULL, step, failed);
}

NDArray *NPH_LazyElementWise(NDArray *a, NPH_UnaryFn op) {
    int failed = 0;
    NDArray *r = lazy_unary(a, np_hip_math_unary_code(op), 0.0f, 0.0f, &failed);
    if (r != NULL || failed) return r;
    if (a != NULL && NPH_Flush(a) != 0) return NULL;
    g_stats.eager_steps++;
    return NDArrayMathGPU_ElementWise(a, op);
}

NDArray *NPH_LazyElementWise1F(NDArray *a, NPH_Unary1FFn op, float val1) {
    int failed = 0;
    NDArray *r = lazy_unary(a, np_hip_math_unary1f_code(op), val1, 0.0f, &failed);
    if (r != NULL || failed) return r;
    if (a != NULL && NPH_Flush(a) != 0) return NULL;
    g_stats.eager_steps++;
    return NDArrayMathGPU_ElementWise1F(a, op, val1);
}

NDArray *NPH_LazyElementWise2F(NDArray *a, NPH_Unary2FFn op, float val1, float val2) {
    int failed = 0;
    NDArray *r = lazy_unary(a, np_hip_math_unary2f_code(op), val1, val2, &failed);
    if (r != NULL || failed) return r;
    if (a != NULL && NPH_Flush(a) != 0) return NULL;
    g_stats.eager_steps++;
    return NDArrayMathGPU_ElementWise2F(a, op, val1, val2);
}
