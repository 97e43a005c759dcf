/*
 * hip_fast.h — GPU early-outs for the reference's L2 functions that pick the device INSIDE the function.
 *
 * `PHP_METHOD(add)` calls NDArray_Add_Float(a, b) for CPU and GPU arrays alike; the switch sits in the function body
 * (src/ndmath/arithmetics.c:160-166,236-262), behind the scalar expand (a full-size NDArray_Zeros + NDArray_Fill
 * temporary, :169-181) and NDArray_Broadcast's materialised copy (:183-197; on the GPU one cudaMemcpy per row or per
 * element, src/ndarray.c:1214-1267).  The six comparisons of src/logic.c:68-670 repeat that skeleton, and reduce()
 * (src/ndarray.c:523-578) issues one NDArray_Add_Float + allocation + copy per slice.  So these symbols cannot be
 * replaced (CPU arrays need the reference's AVX2 body) and the minimal back-end swap of INTEGRATION.md section 2a keeps
 * all of that launch-bound behaviour for GPU arrays.
 *
 * tools/apply_with_hip.py therefore inserts, at the top of each of those functions (after the reference's own
 * device-mismatch check, before the scalar expand):
 *
 *     #ifdef HAVE_NP_HIP
 *         if (NPH_TAKES(a, b)) {
 *             return NPH_Binary_Float(NP_ADD, a, b);
 *         }
 *     #endif
 *
 * GPU operands leave through ONE np_binary launch (operand kinds NP_SCALAR / NP_HOST_SCALAR / NP_ROW / NP_COL instead
 * of temporaries); CPU operands fall through to the reference's code, untouched.  reduce() keeps its own argument
 * checks and result allocation and swaps `_reduce(...)` for NPH_ReduceAxisInto on GPU arrays (one np_reduce_axis); so does
 * single_reduce() for NDArray_Mean_Float, where PHP_METHOD(mean) sends GPU arrays that come with an axis (numpower.c:2677).
 *
 * The file is plain C and compiles against whatever header provides struct NDArray with the reference's layout,
 * NDArray_EmptyLike / NDArray_FREE / NDArray_IsBroadcastable and the NDArray_* accessor macros — the reference's own
 * headers in a PHP build (so results are allocated with emalloc by the reference's NDArray_Empty and released by its
 * NDArray_FREE), include/numpower_host.h in libnumpower_host.so, whose NDArray_*_Float entry points are this same code.
 */
#ifndef NUMPOWER_AMD_EXT_HIP_FAST_H
#define NUMPOWER_AMD_EXT_HIP_FAST_H

#include <stddef.h>

#include "np_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

struct NDArray;

/* true when the fast path takes the call: some operand that is an ARRAY (ndim > 0) lives on the GPU.  0-d operands do
 * not decide (arithmetics.c:163 exempts them from the device check): two 0-d operands stay with the reference. */
#define NPH_TAKES(a, b)                                                                  \
    ((NDArray_NDIM(a) != 0 && NDArray_DEVICE(a) == NDARRAY_DEVICE_GPU) ||               \
     (NDArray_NDIM(b) != 0 && NDArray_DEVICE(b) == NDARRAY_DEVICE_GPU))

/* op: an np_binary_op code — NP_ADD ... NP_POW (arithmetics.c:160-926), NP_EQUAL ... NP_LESS_EQUAL (logic.c:68-670),
 * NP_MAXIMUM / NP_MINIMUM.  Same argument checks, messages and result shape as the reference function of that name;
 * results follow the reference's CPU arithmetic including its AVX2-body quirks (NP_QUIRK_AVX_BODY).  NULL + a raised
 * error on failure. */
struct NDArray *NPH_Binary_Float(int op, struct NDArray *a, struct NDArray *b);

/* How `small` maps onto `large` (the patterns of NDArray_Broadcast, ndarray.c:1196-1291): NP_ROW / NP_COL / NP_SCALAR with
 * the rows x cols view of `large` that np_binary wants, or -1 for "Can't broadcast arrays." */
int NPH_BroadcastKind(const struct NDArray *small, const struct NDArray *large, size_t *rows, size_t *cols);

/* rtn (allocated by the caller with the reduced shape, on the GPU) = reduction of `array` over `axis`;
 * reduce_op = NP_SUM / NP_PROD / NP_MIN / NP_MAX / NP_MEAN.  flags = NP_QUIRK_AVX_BODY when the call stands for
 * reduce(array, &axis, NDArray_Multiply_Float) — the products then carry the zero signs of the reference's slice-by-slice
 * AVX2 multiply — else 0.  0 on success, -1 + a raised error otherwise. */
int NPH_ReduceAxisInto(struct NDArray *array, int axis, int reduce_op, unsigned flags, struct NDArray *rtn);

#ifdef __cplusplus
}
#endif
#endif
