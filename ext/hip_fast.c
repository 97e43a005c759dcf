/*
 * hip_fast.c — the GPU side of the reference's device-dispatching L2 functions, as ONE launch each (hip_fast.h).
 *
 *   NPH_Binary_Float   NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float        src/ndmath/arithmetics.c:160-926
 *                      NDArray_{Greater,Less,LessEqual,GreaterEqual,Equal,NotEqual} src/logic.c:68-670
 *   NPH_ReduceAxisInto the `_reduce(0, 0, axis, array, rtn, operation)` of reduce() src/ndarray.c:572 (:394-429)
 *
 * Plain C over include/np_hip.h.  Compiled twice in this repository's world:
 *   - into a `--with-hip` NumPower tree as src/hip/hip_fast.c (tools/apply_with_hip.py), against the reference's own
 *     headers (-DNUMPOWER_NDARRAY_HEADER='"src/initializers.h"'): results come from the reference's NDArray_EmptyLike
 *     (emalloc + vmalloc) and die in its NDArray_FREE;
 *   - into libnumpower_host.so against include/numpower_host.h, where NDArray_Add_Float & co. ARE these functions
 *     (numpower_host.cpp), which is how the GPU test tier exercises every line below.
 * No CPU arithmetic: an array operand on the host is an error here — in a patched tree the inserted `if (NPH_TAKES(a, b))`
 * never sends one, the reference's AVX2 code below the insertion handles it.
 */
#ifndef NUMPOWER_NDARRAY_HEADER
#define NUMPOWER_NDARRAY_HEADER "numpower_host.h"
#endif
#include NUMPOWER_NDARRAY_HEADER

#include <stddef.h>

#include "hip_fast.h"
#include "np_ext_hooks.h"
#include "np_hip.h"

/* How the smaller operand maps onto the larger one: the patterns NDArray_Broadcast materialises (ndarray.c:1196-1291),
 * with their NumPy meaning (the reference's copy loops leave memory uninitialised for destinations with ndim > 2 and for
 * 1 x C sources with C != R).  -> an np_operand_kind, or -1 for "Can't broadcast arrays." */
int NPH_BroadcastKind(const NDArray *small, const NDArray *large, size_t *rows, size_t *cols) {
    const int ln = NDArray_NDIM(large), sn = NDArray_NDIM(small);
    const long lnum = NDArray_NUMELEMENTS(large);
    if (!NDArray_IsBroadcastable(small, large)) return -1;
    if (sn == 1 && ln > 1) {                                   /* ndarray.c:1202-1223 */
        *cols = (size_t) NDArray_SHAPE(large)[ln - 1];
        *rows = (size_t) (lnum / (long) *cols);
        return NP_ROW;
    }
    if (sn == 2 && ln == 2) {
        const int sr = NDArray_SHAPE(small)[0], sc = NDArray_SHAPE(small)[1];
        const int lr = NDArray_SHAPE(large)[0], lc = NDArray_SHAPE(large)[1];
        *rows = (size_t) lr;
        *cols = (size_t) lc;
        if (sr == 1 && sc == 1) return NP_SCALAR;              /* ndarray.c:1238-1246 */
        if (sr == lr && sc == 1) return NP_COL;                /* ndarray.c:1227-1237 */
        if (sr == 1 && sc == lc) return NP_ROW;                /* ndarray.c:1273-1291 */
    }
    return -1;
}

static int scalar_kind(const NDArray *s) {
    return NDArray_DEVICE(s) == NDARRAY_DEVICE_GPU ? NP_SCALAR : NP_HOST_SCALAR;
}

NDArray *NPH_Binary_Float(int op, NDArray *a, NDArray *b) {
    if (a == NULL || b == NULL) return NULL;
    const int compare = op >= NP_EQUAL && op <= NP_LESS_EQUAL;
    /* arithmetics.c:163-166 / logic.c:70-73 — 0-d operands are exempt from the device check */
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b) && NDArray_NDIM(a) != 0 && NDArray_NDIM(b) != 0) {
        np_ext_throw(compare ? "Devices mismatch in `equal` function"
                             : "Device mismatch, both NDArray MUST be in the same device.");
        return NULL;
    }
    const int a_scalar = NDArray_NDIM(a) == 0, b_scalar = NDArray_NDIM(b) == 0;
    /* where the result lives: with a 0-d operand the other one decides */
    NDArray *place = (a_scalar && !b_scalar) ? b : a;
    if (a_scalar && b_scalar && NDArray_DEVICE(a) != NDARRAY_DEVICE_GPU) {
        if (NDArray_DEVICE(b) != NDARRAY_DEVICE_GPU) {
            np_ext_throw("binary op on two CPU scalars: not a GPU operation");
            return NULL;
        }
        place = b;
    }
    if (NDArray_DEVICE(place) != NDARRAY_DEVICE_GPU) {
        np_ext_throw("binary op: operand is on the CPU; numpower_amd only computes on the GPU "
                     "(call ->gpu() first, the CPU path is the reference's own)");
        return NULL;
    }

    const long na = NDArray_NUMELEMENTS(a), nb = NDArray_NUMELEMENTS(b);
    int ak = NP_FULL, bk = NP_FULL;
    size_t rows = 1, cols = 1;
    NDArray *shape_of = a;
    /* the reference's AVX2 loop bound: NDArray_NUMELEMENTS(a), a = the first operand after the scalar expand but before
     * the broadcast (arithmetics.c:251) */
    size_t loop_numel_a = (size_t) na;

    if (a_scalar || b_scalar) {
        NDArray *full = a_scalar ? b : a;                      /* two 0-d operands: 1 x 1 */
        shape_of = full;
        cols = (size_t) NDArray_NUMELEMENTS(full);
        if (a_scalar && !b_scalar) {
            ak = scalar_kind(a);
            loop_numel_a = (size_t) nb;
        } else if (b_scalar && !a_scalar) {
            bk = scalar_kind(b);
        } else {
            ak = scalar_kind(a);
            bk = scalar_kind(b);
            shape_of = place;
        }
    } else if (na < nb) {                                      /* arithmetics.c:186-189 */
        const int k = NPH_BroadcastKind(a, b, &rows, &cols);
        if (k < 0) {
            np_ext_throw("Can't broadcast arrays.");
            return NULL;
        }
        ak = k;
        shape_of = b;
    } else if (nb < na) {                                      /* arithmetics.c:190-193 */
        const int k = NPH_BroadcastKind(b, a, &rows, &cols);
        if (k < 0) {
            np_ext_throw("Can't broadcast arrays.");
            return NULL;
        }
        bk = k;
    } else {
        cols = (size_t) na;                                    /* equal counts: flat elementwise (arithmetics.c:194-197) */
    }

    NDArray *result = NDArray_EmptyLike(shape_of);             /* shape_of is on the GPU: so is the result */
    if (result == NULL) return NULL;
    /* 0-d x 0-d multiply / divide take the reference's plain short cut (arithmetics.c:302-316,575-580) */
    const int quirk_ops = ((op == NP_MULTIPLY || op == NP_MOD) && !(a_scalar && b_scalar)) ||
                          op == NP_EQUAL || op == NP_NOT_EQUAL;
    const unsigned flags = quirk_ops ? NP_QUIRK_AVX_BODY : 0u;
    /* NotEqual's AVX2 loop runs over the broadcast operand (logic.c:636), every other one over the first operand
     * before the broadcast (arithmetics.c:251, logic.c:535) */
    if (op == NP_NOT_EQUAL) loop_numel_a = rows * cols;
    const size_t body_end = quirk_ops ? np_avx_body_end(loop_numel_a) : 0;
    if (np_binary(op, NDArray_FDATA(a), ak, NDArray_FDATA(b), bk, NDArray_FDATA(result), rows, cols, flags,
                  body_end) != NP_OK) {
        np_ext_throw_last();
        NDArray_FREE(result);
        return NULL;
    }
    return result;
}

int NPH_ReduceAxisInto(NDArray *array, int axis, int reduce_op, unsigned flags, NDArray *rtn) {
    if (array == NULL || rtn == NULL) return -1;
    const int nd = NDArray_NDIM(array);
    if (axis < 0 || axis >= nd) {                              /* reduce() has checked the upper bound (ndarray.c:534-538) */
        np_ext_throw("axis is out of bounds for the array");
        return -1;
    }
    if (NDArray_DEVICE(array) != NDARRAY_DEVICE_GPU || NDArray_DEVICE(rtn) != NDARRAY_DEVICE_GPU) {
        np_ext_throw("axis reduction: operand is on the CPU; numpower_amd only computes on the GPU "
                     "(call ->gpu() first, the CPU path is the reference's own)");
        return -1;
    }
    size_t outer = 1, inner = 1;
    for (int i = 0; i < nd; ++i) {
        if (i < axis) outer *= (size_t) NDArray_SHAPE(array)[i];
        if (i > axis) inner *= (size_t) NDArray_SHAPE(array)[i];
    }
    if ((size_t) NDArray_NUMELEMENTS(rtn) != outer * inner) {
        np_ext_throw("axis reduction: the result array does not have the reduced shape");
        return -1;
    }
    /* NP_QUIRK_AVX_BODY: the reference multiplies slice by slice through NDArray_Multiply_Float, whose AVX2 body turns
     * zero products into -0.0 (arithmetics.c:280-284,397-414).  Its slices have nd - axis - 1 dimensions; 0-d slices take
     * Multiply_Float's plain short cut (arithmetics.c:302-316), so the quirk does not apply to them. */
    if (reduce_op != NP_PROD || nd - axis - 1 < 1) flags &= ~(unsigned) NP_QUIRK_AVX_BODY;
    if (np_reduce_axis(reduce_op, NDArray_FDATA(array), outer, (size_t) NDArray_SHAPE(array)[axis], inner,
                       NDArray_FDATA(rtn), flags) != NP_OK) {
        np_ext_throw_last();
        return -1;
    }
    return 0;
}
