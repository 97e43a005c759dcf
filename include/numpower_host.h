/*
 * numpower_host.h — host-side mirror of the reference's NDArray L2/L3 interface for the hot path.
 *
 * PHP (and its headers) are not available where this was built, so the layer that sits between
 * the Zend glue (numpower.c) and the device back end is provided as a plain C-callable library
 * (libnumpower_host.so, written in C++) with the SAME entry-point names, argument meaning,
 * ownership rules and error messages as the reference's C files, so that numpower.c's
 * PHP_METHODs can call it unchanged:
 *
 *   struct NDArray / NDArrayDescriptor          src/ndarray.h:52-74   (same field order)
 *   NDArray_Zeros/Empty/EmptyLike/Copy/Fill,
 *   NDArray_CreateFrom{Double,Float,Long}Scalar  src/initializers.c:379-448,633-790
 *   NDArray_FREE                                 src/ndarray.c:587-632
 *   NDArray_ToGPU / NDArray_ToCPU                src/ndarray.c:1037-1093
 *   NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float, NDArray_Abs,
 *   NDArray_Sum_Float / Float_Prod / Mean_Float  src/ndmath/arithmetics.c:36-947
 *   NDArray_Min / NDArray_Max                    src/ndarray.c:752-772,939-959
 *   reduce()                                     src/ndarray.c:523-578
 *   NDArrayMathGPU_ElementWise{,1F,2F,1N}        src/ndmath/cuda/cuda_math.cu:1532-1558
 *   NDArray_Matmul / NDArray_FMatmul / NDArray_Dot   src/ndmath/linalg.c:44-82,216-245,354-393
 *
 * Device work goes through the C ABI of include/np_hip.h; this library contains no kernels and
 * no CPU arithmetic.  Arrays on NDARRAY_DEVICE_CPU exist only as staging for ->gpu()/->cpu()
 * and as 0-d scalar operands: an arithmetic call whose array operands live on the CPU fails with
 * an error (the reference's CPU path — AVX2/OpenBLAS — stays the reference's own code; this
 * library never computes on the host).
 *
 * Errors: where the reference calls zend_throw_error(NULL, msg) and returns NULL, these
 * functions call the installed error handler with the same message (default: remember it for
 * numpower_host_last_error()) and return NULL.
 */
#ifndef NUMPOWER_AMD_HOST_H
#define NUMPOWER_AMD_HOST_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDARRAY_DEVICE_CPU 0
#define NDARRAY_DEVICE_GPU 1

typedef struct NDArrayDescriptor {
    const char *type;   /* "float32" */
    int elsize;
    long numElements;
} NDArrayDescriptor;

typedef struct NDArray {
    int uuid;
    int *strides;       /* bytes */
    int *dimensions;
    int ndim;
    char *data;         /* host pointer (device == CPU) or device pointer (device == GPU) */
    struct NDArray *base;
    int flags;
    NDArrayDescriptor *descriptor;
    void *iterator;
    void *php_iterator;
    int refcount;
    int device;
} NDArray;

#define NDArray_FDATA(a) ((float *) ((a)->data))
#define NDArray_NDIM(a) ((int) ((a)->ndim))
#define NDArray_SHAPE(a) ((int *) ((a)->dimensions))
#define NDArray_NUMELEMENTS(a) ((long) ((a)->descriptor->numElements))
#define NDArray_DEVICE(a) ((int) ((a)->device))
#define NDArray_ADDREF(a) ((a)->refcount++)          /* src/ndarray.h:31 */

/* ---- errors (zend_throw_error stand-in) ---- */
typedef void (*numpower_error_handler)(const char *message);
void numpower_host_set_error_handler(numpower_error_handler handler);   /* NULL = default */
const char *numpower_host_last_error(void);   /* "" if none since the last clear */
void numpower_host_clear_error(void);

/* ---- allocation / placement ---- */
/* shape is copied (the reference takes ownership of an emalloc'd shape; copying keeps the C API
 * usable from ctypes — the Zend glue frees its own vector). */
NDArray *NDArray_Zeros(const int *shape, int ndim, const char *type, int device);
NDArray *NDArray_Empty(const int *shape, int ndim, const char *type, int device);
NDArray *NDArray_EmptyLike(NDArray *a);
NDArray *NDArray_Copy(NDArray *a, int device);   /* same-device copy (device must equal a's) */
NDArray *NDArray_Fill(NDArray *a, float fill_value);
NDArray *NDArray_CreateFromDoubleScalar(double scalar);
NDArray *NDArray_CreateFromFloatScalar(float scalar);
NDArray *NDArray_CreateFromLongScalar(long scalar);
/* CPU array from a host buffer (stands in for Create_NDArray_FromZval, initializers.c:30-247) */
NDArray *NDArray_FromHostBuffer(const float *host_data, const int *shape, int ndim);
/* View of slice `index` along the leading axis: what $a[i] / NDArrayIterator_GET return
 * (iterators.c:94-111): shares data, base = a, ADDREFs a. */
NDArray *NDArray_LeadingSlice(NDArray *a, int index);
void NDArray_FREE(NDArray *array);
NDArray *NDArray_ToGPU(NDArray *target);
NDArray *NDArray_ToCPU(NDArray *target);
float NDArray_GetFloatScalar(NDArray *a);        /* ndarray.c:1302-1310 */
/* copy numElements floats of a CPU array into host_out (toArray() plumbing) */
int NDArray_CopyToHostBuffer(NDArray *a, float *host_out);
/* live device allocations (vmemcheck, gpu_alloc.c:36-40) */
long NDArray_LiveDeviceAllocations(void);

/* ---- binary elementwise (arithmetics.c:160-926) ---- */
NDArray *NDArray_Add_Float(NDArray *a, NDArray *b);
NDArray *NDArray_Subtract_Float(NDArray *a, NDArray *b);
NDArray *NDArray_Multiply_Float(NDArray *a, NDArray *b);
NDArray *NDArray_Divide_Float(NDArray *a, NDArray *b);
NDArray *NDArray_Mod_Float(NDArray *a, NDArray *b);
NDArray *NDArray_Pow_Float(NDArray *a, NDArray *b);
int NDArray_IsBroadcastable(const NDArray *array1, const NDArray *array2);   /* ndarray.c:1124-1162 */

/* ---- comparisons (src/logic.c:25-670): 1.0f / 0.0f masks ---- */
NDArray *NDArray_Equal(NDArray *nda, NDArray *ndb);
NDArray *NDArray_NotEqual(NDArray *nda, NDArray *ndb);
NDArray *NDArray_Greater(NDArray *nda, NDArray *ndb);
NDArray *NDArray_GreaterEqual(NDArray *nda, NDArray *ndb);
NDArray *NDArray_Less(NDArray *nda, NDArray *ndb);
NDArray *NDArray_LessEqual(NDArray *nda, NDArray *ndb);
/* elementwise fmaxf / fminf with broadcast (ndarray.c:853-931; GPU arrays are refused there) */
NDArray *NDArray_Maximum(NDArray *a, NDArray *b);
NDArray *NDArray_Minimum(NDArray *a, NDArray *b);
float NDArray_All(NDArray *a);   /* 1 / 0, reproduces the reference's CPU result (logic.c:25-58) */

/* ---- layout (src/manipulation.c:68-130) ---- */
typedef struct NDArray_Dims {   /* src/ndarray.h:40-43 */
    int *ptr;
    int len;
} NDArray_Dims;
/* New contiguous array with the axes permuted (NULL = reverse all axes). */
NDArray *NDArray_Transpose(NDArray *a, NDArray_Dims *permute);

/* ---- initializers (src/initializers.c:458-510,655-660,818-841; the reference's benchmarks/initializers) ----
 * Reference names create CPU arrays as in the reference; the ...On variants take the device, so an
 * array can be born on the GPU without a PCIe copy.  NDArray_ArangeOn reproduces the reference's
 * float recurrence bit for bit (np_arange); on NDARRAY_DEVICE_CPU it raises (host arithmetic stays
 * the reference's own code). */
NDArray *NDArray_Ones(const int *shape, int ndim, const char *type);
NDArray *NDArray_Full(const int *shape, int ndim, double fill_value);
NDArray *NDArray_Identity(int size);
NDArray *NDArray_FullOn(const int *shape, int ndim, double fill_value, int device);
NDArray *NDArray_IdentityOn(int size, int device);
NDArray *NDArray_ArangeOn(double start, double stop, double step, int device);

/* ---- views, layout and equality around the path ----
 * NDArray_ArrayEqual   logic.c:703-716      1 / 0 (0 on shape mismatch)
 * NDArray_AllClose     logic.c:750-772      1 / 0, -1 + error on shape / device mismatch (the
 *                                           reference rejects GPU arrays; here it is one reduction)
 * NDArray_ToContiguous manipulation.c:381-421   strided view -> new contiguous array, one launch
 * NDArray_Diagonal     indexing.c:21-48     (`offset` is ignored, as in the reference)
 * NDArray_Trace        linalg.c:758-767     0-d CPU scalar
 * NDArray_Reshape      manipulation.c:138-162   view sharing data (ADDREFs target)
 * NDArray_Flatten      manipulation.c:169-184   1-D copy
 * NDArray_ExpandDim    manipulation.c:453-513   axis: 0-d or 1-D CPU array
 * NDArray_ConcatenateFlat / NDArray_Append  manipulation.c:293-374 (axis must be -1)
 * NDArray_Slice        manipulation.c:193-283   indexes[i] = CPU array [start(,stop(,step))] */
int NDArray_ArrayEqual(NDArray *a, NDArray *b);
int NDArray_AllClose(NDArray *a, NDArray *b, float rtol, float atol);
NDArray *NDArray_ToContiguous(NDArray *a);
NDArray *NDArray_Diagonal(NDArray *target, int offset);
NDArray *NDArray_Trace(NDArray *a);
NDArray *NDArray_Reshape(NDArray *target, int *new_shape, int ndim);
NDArray *NDArray_Flatten(NDArray *target);
NDArray *NDArray_ExpandDim(NDArray *a, NDArray *axis);
NDArray *NDArray_ConcatenateFlat(NDArray **arrays, int num_arrays);
NDArray *NDArray_Append(NDArray **arrays, int axis, int num_arrays);
NDArray *NDArray_Slice(NDArray *array, NDArray **indexes, int num_indices);

/* ---- fused elementwise chains ----
 * Evaluates acc = inputs[0]; acc = op_k(acc [, inputs[operand_k]]) ... in one kernel; `ops` uses
 * np_fused_op of np_hip.h (flags/body_end are filled in here).  Bit-identical to calling the
 * stand-alone entry points one after the other. */
#include "np_hip.h"
NDArray *NDArray_FusedChain(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops);
/* ... ending in a full reduction (np_reduce_op: sum / prod / min / max / mean) of the chain value,
 * which is never written to memory.  Returns NaN and raises on error. */
float NDArray_FusedChainReduce(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops, int reduce_op);
/* ... ending in a reduction over one axis (same `axis` rules and error as reduce(), ndarray.c:534-538): the
 * result has the chain's shape without that axis.  The last axis of any array and the first axis of a
 * 2-d array run as ONE kernel (np_fused_chain_reduce_axis); any other axis evaluates the chain into a
 * temporary and reduces that. */
NDArray *NDArray_FusedChainReduceAxis(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops, int reduce_op,
                                      int axis);

/* ---- argmax / argmin (src/ndmath/calculation.c:73-194) ----
 * axis = 128 (NDARRAY_MAX_DIMS) reduces the flattened array; indices are returned as floats. */
#ifndef __cplusplus
#include <stdbool.h>
#endif
NDArray *NDArray_ArgMinMaxCommon(NDArray *op, int axis, bool keepdims, bool is_argmax);

/* ---- statistics (src/ndmath/statistics.c:88-154): 0-d CPU scalar results ---- */
NDArray *NDArray_Variance(NDArray *a);
NDArray *NDArray_Std(NDArray *a);
NDArray *NDArray_Average(NDArray *a, NDArray *weights /* may be NULL */);

/* ---- unary elementwise: the reference's drivers with the reference's signatures ----
 * (src/ndmath/cuda/cuda_math.h:10-15,75-76; implemented in ext/hip_math_drivers.c).  `op` is one of the
 * cuda_float_* functions of ext/hip_math.h — also exported by this library — exactly as numpower.c
 * passes them: NDArrayMathGPU_ElementWise(nda, cuda_float_sin) (numpower.c:1651).  Recognised pointers
 * run as ONE out-of-place np_unary pass (no NDArray_Copy first); any other pointer gets the reference's
 * copy + in-place call. */
#ifndef NUMPOWER_AMD_EXT_HIP_MATH_H
typedef void (*ElementWiseFloatGPUOperation)(int, float *);
typedef void (*ElementWiseFloatGPUOperation2F)(int, float *, float, float);
typedef void (*ElementWiseFloatGPUOperation1F)(int, float *, float);
typedef void (*ElementWiseFloatGPUOperation1N)(int, float *, float *);
#endif
NDArray *NDArrayMathGPU_ElementWise(NDArray *ndarray, ElementWiseFloatGPUOperation op);
NDArray *NDArrayMathGPU_ElementWise1F(NDArray *ndarray, ElementWiseFloatGPUOperation1F op, float val1);
NDArray *NDArrayMathGPU_ElementWise2F(NDArray *ndarray, ElementWiseFloatGPUOperation2F op, float val1, float val2);
NDArray *NDArrayMathGPU_ElementWise1N(NDArray *ndarray, ElementWiseFloatGPUOperation1N op, NDArray *val1);
NDArray *NDArray_Abs(NDArray *nda);
/* float_rsqrt (double_math.c:111-126) on the device; numpower.c:1791 passes cuda_float_arccos to the
 * driver for GPU arrays by mistake (no cuda_float_rsqrt exists), so rsqrt gets its own entry point */
NDArray *NDArray_Rsqrt(NDArray *nda);
/* float_exp2 (double_math.c:31-33) on the device: the reference has no GPU exp2 at all (numpower.c:3153 maps
 * the CPU kernel over whatever pointer the array holds; cuda_math.h declares no cuda_float_exp2) */
NDArray *NDArray_Exp2(NDArray *nda);

/* ---- reductions ---- */
float NDArray_Sum_Float(NDArray *a);
float NDArray_Float_Prod(NDArray *a);
float NDArray_Mean_Float(NDArray *a);
float NDArray_Min(NDArray *target);
float NDArray_Max(NDArray *target);
/* NDArray_Median_Float  arithmetics.c:149-158 (calculate_median :111-138): (t[n/2-1] + t[n/2]) / 2.0f for an
 *                       even count, t[n/2] for an odd one, t = the sorted elements.
 * NDArray_Quantile      statistics.c:60-79 (calculate_quantile :14-50): q a 0-d array in [0, 1];
 *                       index = (float)(n-1)*q, linear interpolation between t[(int)index] and the next
 *                       element, in the reference build's arithmetic (one fma); errors "Q must be a
 *                       scalar", "Q must be between 0 and 1".  Returns a 0-d array on the device.
 * The reference refuses device arrays for both ("Median not available for GPU.", "Quantile not
 * available for GPU device.") and sorts a host copy; here the two order statistics come from
 * np_order_stat (radix select on the device). */
float NDArray_Median_Float(NDArray *a);

/* ---- manipulation wrappers (manipulation.c:554-1073, initializers.c:597-625), device arrays ----
 * atleast_* / squeeze are views (reshape) of the contiguous input; swapaxes / rollaxis / moveaxis
 * materialise through NDArray_Transpose (np_permute); concatenate is one pitched copy (np_copy2d) per
 * input, v/h/d/column_stack are concatenate over atleast_2d / 1d / 3d / transposed inputs; diag of a
 * vector is a zero fill + one pitched copy, of a matrix its diagonal.  Error messages are the
 * reference's.  Deviations, where the reference is undefined or wrong: atleast_3d of a 2-d array is
 * numpy's (r, c, 1) (the reference overflows a 2-int buffer there); rollaxis / moveaxis follow
 * numpy's definition (the reference's index shuffling is only right for the cases it agrees with). */
NDArray *NDArray_AtLeast1D(NDArray *a);
NDArray *NDArray_AtLeast2D(NDArray *a);
NDArray *NDArray_AtLeast3D(NDArray *a);
NDArray *NDArray_Squeeze(NDArray *a, NDArray *axis);   /* axis: NULL, a CPU 0-d or 1-d array */
NDArray *NDArray_SwapAxes(NDArray *a, int axis1, int axis2);
NDArray *NDArray_Rollaxis(NDArray *a, int axis, int start);
NDArray *NDArray_Moveaxis(NDArray *a, int *src, int *dest, int n_source, int n_dest);
NDArray *NDArray_Concatenate(NDArray **arrays, int narrays, int axis);
NDArray *NDArray_VSTACK(NDArray **arrays, int narrays);
NDArray *NDArray_HSTACK(NDArray **arrays, int narrays);
NDArray *NDArray_DSTACK(NDArray **arrays, int narrays);
NDArray *NDArray_ColumnStack(NDArray **arrays, int narrays);
NDArray *NDArray_Diag(NDArray *a);
NDArray *NDArray_Quantile(NDArray *target, NDArray *q);
/* operation must be NDArray_Add_Float or NDArray_Multiply_Float (the two the reference passes,
 * numpower.c:4637,4742,2661); anything else is an error. */
NDArray *reduce(NDArray *array, int *axis, NDArray *(*operation)(NDArray *, NDArray *));
/* min/max along an axis: the reference's single_reduce path is unreachable/broken
 * (numpower.c:4654 parses one argument; apply_single_reduce writes element 0 only), so these are
 * additions with NumPy semantics. */
NDArray *NDArray_MinAxis(NDArray *target, int axis);
NDArray *NDArray_MaxAxis(NDArray *target, int axis);

/* ---- matmul ---- */
NDArray *NDArray_Matmul(NDArray *a, NDArray *b);
NDArray *NDArray_FMatmul(NDArray *a, NDArray *b);
NDArray *NDArray_Dot(NDArray *nda, NDArray *ndb);
NDArray *NDArray_Outer(NDArray *a, NDArray *b);   /* linalg.c:724-751 */
NDArray *NDArray_Inner(NDArray *nda, NDArray *ndb);   /* linalg.c:310-345: sum of all products */
/* batch x M x K times batch x K x N -> batch x M x N (BASELINE config 5; no reference entry
 * point: linalg.c:239-242 rejects ndim > 2 with "Stack of matrices not allowed") */
NDArray *NDArray_BatchedMatmul(NDArray *a, NDArray *b);

/* ---- the sharded form of the batched matmul (BASELINE config 5) ----
 * One PHP process per GPU (NDArray::setDevice, numpower.c:615-635, picks the process's device; the reference has
 * no multi-device code).  NDArray_CommInit joins the node's ranks (np_comm_init: endpoint "tcp://127.0.0.1:port"
 * or a file path); 0 on success, -1 + a thrown Error otherwise.
 * NDArray_ShardedBatchedMatmul: a_slab [slab x M x K] and b_slab [slab x K x N] are THIS rank's contiguous share of
 * a batch of `batch` = slab * world products (rank r holds matrices r*slab ... (r+1)*slab - 1).
 *   gather_mode  NP_SHARD_KEEP (0)      -> [slab x M x N], this rank's products only: no collective at all
 *                NP_SHARD_GATHER (1)    -> [batch x M x N] replicated on every rank: the GEMM, then ONE all-gather
 *                k >= 2                 -> the same result, slab computed in k pieces, each piece's transfer
 *                                          overlapped with the next piece's GEMM (np_sgemm_strided_batched_allgather)
 *                NP_SHARD_OVERLAP (-1)  -> the same, the number of pieces chosen by the library's step model (chunks = 0):
 *                                          what a caller without a reason of its own passes
 * Without a communicator the process is a world of one: batch must equal slab.  Errors (thrown like every other
 * method's): device mismatch / shape mismatch with NDArray_Matmul's messages (linalg.c:219-237),
 * "Batch of %d is not %d slab(s) of %d" when the shares do not add up. */
#define NP_SHARD_KEEP 0
#define NP_SHARD_GATHER 1
#define NP_SHARD_OVERLAP (-1)
int NDArray_CommInit(int rank, int world, const char *endpoint);
int NDArray_CommDestroy(void);
int NDArray_CommRank(void);
int NDArray_CommWorld(void);
NDArray *NDArray_ShardedBatchedMatmul(NDArray *a_slab, NDArray *b_slab, int batch, int gather_mode);

#ifdef __cplusplus
}
#endif

#endif /* NUMPOWER_AMD_HOST_H */
