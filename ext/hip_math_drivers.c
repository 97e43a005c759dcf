/*
 * hip_math_drivers.c — NDArrayMathGPU_ElementWise{,1F,2F,1N} with the reference's signatures
 * (src/ndmath/cuda/cuda_math.h:14-15,75-76; CUDA bodies cuda_math.cu:1532-1558), so that
 *
 *     rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_sin);            numpower.c:1651
 *     rtn = NDArrayMathGPU_ElementWise2F(nda, cuda_float_clip, min, max);   numpower.c:2487
 *     rtn = NDArrayMathGPU_ElementWise1F(nda, cuda_float_round, p);     numpower.c:2959
 *     rtn = NDArrayMathGPU_ElementWise1N(ndx, cuda_float_arctan2, ndy); numpower.c:1899
 *
 * and the other ~35 PHP_METHODs compile unchanged.  The reference copies the array and runs the
 * op in place on the copy (16 B/elem: read + write for the copy, read + write for the op).  Here
 * the function pointer is RECOGNISED (hip_math.c: np_hip_math_*_code) and the op runs out of
 * place straight from the input into the fresh result — one 8 B/elem pass, no copy.  A pointer
 * that is not one of hip_math.c's functions (somebody's own in-place kernel wrapper) still gets
 * the reference's behaviour: NDArray_Copy, then op(n, data).
 *
 * Compiles against any header that provides struct NDArray with the reference's layout
 * (src/ndarray.h:52-74), NDArray_EmptyLike / NDArray_Copy / NDArray_FREE and the NDArray_FDATA /
 * NDArray_NUMELEMENTS / NDArray_DEVICE macros: the reference's "src/ndarray.h" +
 * "src/initializers.h" in a PHP build (-DNUMPOWER_NDARRAY_HEADER='"src/initializers.h"'),
 * include/numpower_host.h here.
 */
#ifndef NUMPOWER_NDARRAY_HEADER
#define NUMPOWER_NDARRAY_HEADER "numpower_host.h"
#endif
#include NUMPOWER_NDARRAY_HEADER

#include <stddef.h>

#include "hip_math.h"
#include "np_ext_hooks.h"
#include "np_hip.h"

NDArray *NDArrayMathGPU_ElementWise(NDArray *ndarray, ElementWiseFloatGPUOperation op);
NDArray *NDArrayMathGPU_ElementWise1F(NDArray *ndarray, ElementWiseFloatGPUOperation1F op, float val1);
NDArray *NDArrayMathGPU_ElementWise2F(NDArray *ndarray, ElementWiseFloatGPUOperation2F op, float val1, float val2);
NDArray *NDArrayMathGPU_ElementWise1N(NDArray *ndarray, ElementWiseFloatGPUOperation1N op, NDArray *val1);

static int on_device(NDArray *a) {
    if (a == NULL) {
        np_ext_throw("elementwise op: null array");
        return 0;
    }
    if (NDArray_DEVICE(a) != NDARRAY_DEVICE_GPU) {
        /* the PHP_METHODs only take this branch for GPU arrays (numpower.c:1649); a CPU array here is a
         * caller bug, and running a device kernel on a host pointer would fault */
        np_ext_throw("elementwise op: operand is on the CPU; numpower_amd only computes on the GPU "
                     "(call ->gpu() first, the CPU path is the reference's own)");
        return 0;
    }
    return 1;
}

/* the reference's function-pointer interface counts elements in an `int` (cuda_math.h:14-15): an operation that is not one of
 * this back end's own (those go out of place above, with size_t counts) cannot be handed more than INT_MAX elements */
static int int_count(NDArray *a, int *n) {
    if (NDArray_NUMELEMENTS(a) > 2147483647L) {
        np_ext_throw("array too large for an element-wise operation with an int element count");
        return 0;
    }
    *n = (int)NDArray_NUMELEMENTS(a);
    return 1;
}

/* one pass: out[i] = f(in[i]) */
static NDArray *unary_out_of_place(NDArray *in, int code, float p0, float p1) {
    NDArray *out = NDArray_EmptyLike(in);
    if (out == NULL) return NULL;
    if (np_unary(code, NDArray_FDATA(in), NDArray_FDATA(out), (size_t)NDArray_NUMELEMENTS(in), p0, p1) != NP_OK) {
        np_ext_throw_last();
        NDArray_FREE(out);
        return NULL;
    }
    return out;
}

NDArray *NDArrayMathGPU_ElementWise(NDArray *ndarray, ElementWiseFloatGPUOperation op) {
    if (!on_device(ndarray)) return NULL;
    const int code = np_hip_math_unary_code(op);
    if (code >= 0) return unary_out_of_place(ndarray, code, 0.0f, 0.0f);
    NDArray *rtn = NDArray_Copy(ndarray, NDArray_DEVICE(ndarray));     /* cuda_math.cu:1533-1536 */
    int n = 0;
    if (rtn != NULL && op != NULL) {
        if (!int_count(rtn, &n)) {
            NDArray_FREE(rtn);
            return NULL;
        }
        op(n, NDArray_FDATA(rtn));
    }
    return rtn;
}

NDArray *NDArrayMathGPU_ElementWise1F(NDArray *ndarray, ElementWiseFloatGPUOperation1F op, float val1) {
    if (!on_device(ndarray)) return NULL;
    const int code = np_hip_math_unary1f_code(op);
    if (code >= 0) return unary_out_of_place(ndarray, code, val1, 0.0f);
    NDArray *rtn = NDArray_Copy(ndarray, NDArray_DEVICE(ndarray));     /* cuda_math.cu:1540-1543 */
    int n = 0;
    if (rtn != NULL && op != NULL) {
        if (!int_count(rtn, &n)) {
            NDArray_FREE(rtn);
            return NULL;
        }
        op(n, NDArray_FDATA(rtn), val1);
    }
    return rtn;
}

NDArray *NDArrayMathGPU_ElementWise2F(NDArray *ndarray, ElementWiseFloatGPUOperation2F op, float val1, float val2) {
    if (!on_device(ndarray)) return NULL;
    const int code = np_hip_math_unary2f_code(op);
    if (code >= 0) return unary_out_of_place(ndarray, code, val1, val2);
    NDArray *rtn = NDArray_Copy(ndarray, NDArray_DEVICE(ndarray));     /* cuda_math.cu:1554-1557 */
    int n = 0;
    if (rtn != NULL && op != NULL) {
        if (!int_count(rtn, &n)) {
            NDArray_FREE(rtn);
            return NULL;
        }
        op(n, NDArray_FDATA(rtn), val1, val2);
    }
    return rtn;
}

/* x[i] = f(x[i], y[i]) over numel(x) (cuda_math.cu:1547-1550; the only `op` the reference passes is
 * cuda_float_arctan2).  The reference reads numel(x) elements of y unchecked; a shorter y is refused. */
NDArray *NDArrayMathGPU_ElementWise1N(NDArray *ndarray, ElementWiseFloatGPUOperation1N op, NDArray *val1) {
    if (!on_device(ndarray) || !on_device(val1)) return NULL;
    if (NDArray_NUMELEMENTS(val1) < NDArray_NUMELEMENTS(ndarray)) {
        np_ext_throw("Incompatible shapes");
        return NULL;
    }
    const int code = np_hip_math_binary1n_code(op);
    if (code < 0) {
        NDArray *rtn = NDArray_Copy(ndarray, NDArray_DEVICE(ndarray));
        int n = 0;
        if (rtn != NULL && op != NULL) {
            if (!int_count(rtn, &n)) {
                NDArray_FREE(rtn);
                return NULL;
            }
            op(n, NDArray_FDATA(rtn), NDArray_FDATA(val1));
        }
        return rtn;
    }
    NDArray *out = NDArray_EmptyLike(ndarray);
    if (out == NULL) return NULL;
    if (np_binary(code, NDArray_FDATA(ndarray), NP_FULL, NDArray_FDATA(val1), NP_FULL, NDArray_FDATA(out), 1,
                  (size_t)NDArray_NUMELEMENTS(ndarray), 0, 0) != NP_OK) {
        np_ext_throw_last();
        NDArray_FREE(out);
        return NULL;
    }
    return out;
}
