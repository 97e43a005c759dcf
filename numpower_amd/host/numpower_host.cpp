// libnumpower_host.so — host-side mirror of the reference's NDArray L2/L3 entry points for the
// hot path (include/numpower_host.h).  No kernels and no host arithmetic live here: every op
// classifies its operands the way the reference does (scalar expand, broadcast pattern, shape
// and device checks, error messages) and then makes ONE call into the C ABI of np_hip.h.
//
// What the reference does around each op and what happens here instead:
//   scalar operand   Zeros + Fill of a full-size temporary (arithmetics.c:169-181)
//                    -> operand kind NP_SCALAR / NP_HOST_SCALAR, no temporary
//   broadcast        NDArray_Broadcast materialises a full-size copy, on the GPU with one
//                    cudaMemcpy per row or per ELEMENT (ndarray.c:1214-1267)
//                    -> operand kind NP_ROW / NP_COL, index arithmetic in the kernel
//   result buffer    vmalloc + cudaDeviceSynchronize per op (arithmetics.c:216-221)
//                    -> np_malloc from the caching pool, no sync
//   unary ops        NDArray_Copy then in-place kernel (cuda_math.cu:1532-1537)
//                    -> one out-of-place kernel
//   axis reductions  one Add_Float + alloc + D2D copy + free per slice (ndarray.c:394-429)
//                    -> np_reduce_axis (one or two launches)
//   matmul           cublasCreate / cublasSgemm / cublasDestroy per call (linalg.c:55-71)
//                    -> np_sgemm
#include "numpower_host.h"

#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "hip_fast.h"
#include "hip_lazy.h"
#include "np_hip.h"

namespace {

const char *const TYPE_FLOAT32 = "float32";   // src/types.h:5
constexpr int NP_MAX_ND_HOST = 8;   // np_permute / np_strided_copy handle up to 8 axes

thread_local char g_error[512] = "";
numpower_error_handler g_handler = nullptr;

// zend_throw_error(NULL, fmt, ...) stand-in
void throw_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    if (g_handler) g_handler(g_error);
}

// a failed C-ABI call surfaces its own message
bool dev_ok(int rc) {
    if (rc == NP_OK) return true;
    throw_error("%s", np_last_error());
    (void)np_clear_device_error(nullptr);   // the raised error IS the report: a sticky device error is acknowledged with it (np_ext_hooks.h)
    return false;
}

long shape_numel(const int *shape, int ndim) {
    long n = 1;   // ndim == 0 -> 1 (Create_NDArray, initializers.c:265-269)
    for (int i = 0; i < ndim; ++i) n *= shape[i];
    return n;
}

// Generate_Strides (initializers.c:115-135): byte strides, C order.  The struct keeps them as `int` (ndarray.h:52-74): a slab of
// 2 GiB or more under an axis — (4, 10^9) is 16 GB, a small array on this device — does not fit, and the reference's
// `shape[i + 1] * strides[i + 1]` wraps there.  Such a stride is stored as kStrideOverflow: the hot path never looks at strides
// (contiguous-only contract), NDArray_LeadingSlice computes its offset from the extents, and the strided views refuse.
constexpr int kStrideOverflow = INT_MIN;
int *make_strides(const int *shape, int ndim) {
    int *st = (int *)malloc(sizeof(int) * (ndim > 0 ? ndim : 1));
    if (ndim > 0) {
        long long s = (long long)sizeof(float);
        st[ndim - 1] = (int)s;
        for (int i = ndim - 2; i >= 0; --i) {
            s = s <= (long long)INT_MAX ? s * (long long)shape[i + 1] : s;
            st[i] = s <= (long long)INT_MAX ? (int)s : kStrideOverflow;
        }
    }
    return st;
}

bool strides_fit(const NDArray *a) {
    for (int i = 0; i < a->ndim; ++i)
        if (a->strides[i] == kStrideOverflow) return false;
    return true;
}

// Create_NDArray (initializers.c:255-286) without data
NDArray *make_header(const int *shape, int ndim, int device) {
    NDArray *a = (NDArray *)calloc(1, sizeof(NDArray));
    a->dimensions = (int *)malloc(sizeof(int) * (ndim > 0 ? ndim : 1));
    if (ndim > 0) memcpy(a->dimensions, shape, sizeof(int) * ndim);
    a->strides = make_strides(shape, ndim);
    a->ndim = ndim;
    a->descriptor = (NDArrayDescriptor *)malloc(sizeof(NDArrayDescriptor));
    a->descriptor->type = TYPE_FLOAT32;
    a->descriptor->elsize = (int)sizeof(float);
    a->descriptor->numElements = shape_numel(shape, ndim);
    a->refcount = 1;
    a->device = device;
    return a;
}

void free_header(NDArray *a) {
    free(a->strides);
    free(a->dimensions);
    free(a->descriptor);
    free(a);
}

// allocate the data buffer of a fresh header on its device
bool alloc_data(NDArray *a, bool zero) {
    const size_t bytes = (size_t)NDArray_NUMELEMENTS(a) * sizeof(float);
    if (a->device == NDARRAY_DEVICE_GPU) {
        void *p = nullptr;
        if (!dev_ok(np_malloc(&p, bytes ? bytes : sizeof(float)))) return false;
        a->data = (char *)p;
        if (zero && bytes && !dev_ok(np_memset0(p, bytes))) return false;
    } else {
        a->data = (char *)(zero ? calloc(bytes ? bytes : sizeof(float), 1)
                                : malloc(bytes ? bytes : sizeof(float)));
    }
    return true;
}

NDArray *new_array(const int *shape, int ndim, int device, bool zero) {
    NDArray *a = make_header(shape, ndim, device);
    if (!alloc_data(a, zero)) {
        free_header(a);
        return nullptr;
    }
    return a;
}

bool same_shape(const NDArray *a, const NDArray *b) {
    if (a->ndim != b->ndim) return false;
    for (int i = 0; i < a->ndim; ++i)
        if (a->dimensions[i] != b->dimensions[i]) return false;
    return true;
}

bool require_gpu(const NDArray *a, const char *what) {
    if (a->device == NDARRAY_DEVICE_GPU) return true;
    throw_error("%s: operand is on the CPU; numpower_amd only computes on the GPU "
                "(call ->gpu() first, the CPU path is the reference's own)", what);
    return false;
}

// Shared body of NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float (arithmetics.c:160-926) and of the comparison family
// NDArray_{Equal,NotEqual,Greater,GreaterEqual,Less,LessEqual} (logic.c:67-670): ext/hip_fast.c — the same C that a
// `--with-hip` NumPower tree compiles as the GPU early-out of those functions, so the GPU test tier
// that drives this library drives exactly the code a PHP build runs.
NDArray *binary_op(int op, NDArray *a, NDArray *b) { return NPH_Binary_Float(op, a, b); }

NDArray *unary_op(NDArray *x, int op, float p0, float p1) {
    if (!x) return nullptr;
    if (!require_gpu(x, "elementwise op")) return nullptr;
    NDArray *out = new_array(x->dimensions, x->ndim, NDARRAY_DEVICE_GPU, false);
    if (!out) return nullptr;
    if (!dev_ok(np_unary(op, NDArray_FDATA(x), NDArray_FDATA(out), (size_t)NDArray_NUMELEMENTS(x), p0, p1))) {
        NDArray_FREE(out);
        return nullptr;
    }
    return out;
}

// Full reduction to a host float.  The float-returning reference entry points (NDArray_Sum_Float ...) have no
// error channel besides the raised error, so failure returns -1.0f WITH an error raised; callers that go on
// computing with the value pass `ok` and stop when it comes back false.
float reduce_all(NDArray *a, int op, const char *what, bool *ok = nullptr) {
    if (ok) *ok = false;
    if (!a) {
        throw_error("%s: null array", what);
        return -1.0f;
    }
    if (!require_gpu(a, what)) return -1.0f;
    float v = 0.0f;
    if (!dev_ok(np_reduce_all(op, NDArray_FDATA(a), (size_t)NDArray_NUMELEMENTS(a), &v))) return -1.0f;
    if (ok) *ok = true;
    return v;
}

NDArray *reduce_axis(NDArray *array, int axis, int op, bool quirk) {
    if (!array) return nullptr;
    if (axis >= NDArray_NDIM(array) || axis < 0) {   // ndarray.c:534-538
        throw_error("axis %d is out of bounds for array of dimension %d", axis, NDArray_NDIM(array));
        return nullptr;
    }
    if (!require_gpu(array, "axis reduction")) return nullptr;
    const int nd = NDArray_NDIM(array);
    int out_shape[128];
    int j = 0;
    size_t outer = 1, inner = 1;
    for (int i = 0; i < nd; ++i) {
        if (i != axis) out_shape[j++] = array->dimensions[i];
        if (i < axis) outer *= (size_t)array->dimensions[i];
        if (i > axis) inner *= (size_t)array->dimensions[i];
    }
    NDArray *rtn = new_array(out_shape, nd - 1, NDARRAY_DEVICE_GPU, false);
    if (!rtn) return nullptr;
    // ext/hip_fast.c: what a patched reduce() (ndarray.c:572) calls on GPU arrays
    if (NPH_ReduceAxisInto(array, axis, op, quirk ? NP_QUIRK_AVX_BODY : 0u, rtn) != 0) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}

}  // namespace

extern "C" {

/* ---- errors ---- */
void numpower_host_set_error_handler(numpower_error_handler handler) { g_handler = handler; }
const char *numpower_host_last_error(void) { return g_error; }
void numpower_host_clear_error(void) { g_error[0] = '\0'; }

/* ---- allocation / placement ---- */

NDArray *NDArray_Zeros(const int *shape, int ndim, const char *type, int device) {
    (void)type;
    if (ndim > 0 && !shape) return nullptr;
    return new_array(shape, ndim, device, true);
}

NDArray *NDArray_Empty(const int *shape, int ndim, const char *type, int device) {
    (void)type;
    if (ndim > 0 && !shape) return nullptr;
    return new_array(shape, ndim, device, false);
}

NDArray *NDArray_EmptyLike(NDArray *a) {
    return new_array(a->dimensions, a->ndim, a->device, false);
}

NDArray *NDArray_Copy(NDArray *a, int device) {   // initializers.c:742-790
    if (!a) return nullptr;
    if (device != a->device) {
        throw_error("NDArray_Copy: use NDArray_ToGPU / NDArray_ToCPU to change devices");
        return nullptr;
    }
    NDArray *r = new_array(a->dimensions, a->ndim, device, false);
    if (!r) return nullptr;
    const size_t bytes = (size_t)NDArray_NUMELEMENTS(a) * sizeof(float);
    if (device == NDARRAY_DEVICE_GPU) {
        if (!dev_ok(np_memcpy_d2d(r->data, a->data, bytes))) {
            NDArray_FREE(r);
            return nullptr;
        }
    } else {
        memcpy(r->data, a->data, bytes);
    }
    return r;
}

NDArray *NDArray_Fill(NDArray *a, float fill_value) {   // initializers.c:633-648
    if (!a) return nullptr;
    const size_t n = (size_t)NDArray_NUMELEMENTS(a);
    if (a->device == NDARRAY_DEVICE_GPU) {
        if (!dev_ok(np_fill(NDArray_FDATA(a), fill_value, n))) return nullptr;
    } else {
        // plain store loop: placement plumbing, not arithmetic
        for (size_t i = 0; i < n; ++i) NDArray_FDATA(a)[i] = fill_value;
    }
    return a;
}

NDArray *NDArray_CreateFromFloatScalar(float scalar) {   // initializers.c:693-710
    NDArray *r = new_array(nullptr, 0, NDARRAY_DEVICE_CPU, false);
    NDArray_FDATA(r)[0] = scalar;
    return r;
}
NDArray *NDArray_CreateFromDoubleScalar(double scalar) { return NDArray_CreateFromFloatScalar((float)scalar); }
NDArray *NDArray_CreateFromLongScalar(long scalar) { return NDArray_CreateFromFloatScalar((float)scalar); }

NDArray *NDArray_FromHostBuffer(const float *host_data, const int *shape, int ndim) {
    if (!host_data) return nullptr;
    NDArray *r = new_array(shape, ndim, NDARRAY_DEVICE_CPU, false);
    memcpy(r->data, host_data, (size_t)NDArray_NUMELEMENTS(r) * sizeof(float));
    return r;
}

NDArray *NDArray_LeadingSlice(NDArray *a, int index) {   // iterators.c:94-111
    if (!a || a->ndim < 1) {
        throw_error("cannot take a slice of a 0-d array");
        return nullptr;
    }
    if (index < 0 || index >= a->dimensions[0]) {
        throw_error("Index out of bounds");
        return nullptr;
    }
    NDArray *r = make_header(a->dimensions + 1, a->ndim - 1, a->device);
    // (a stride that does not fit the struct's int belongs to a freshly laid out, contiguous array: the slab is its other extents)
    const size_t slab = a->strides[0] != kStrideOverflow ? (size_t)a->strides[0]
                                                         : (size_t)shape_numel(a->dimensions + 1, a->ndim - 1) * sizeof(float);
    r->data = a->data + (size_t)index * slab;
    r->base = a;
    a->refcount++;   // NDArray_ADDREF
    return r;
}

void NDArray_FREE(NDArray *array) {   // ndarray.c:587-632
    if (array == nullptr || array->refcount == -1) return;
    NPH_OnFree(array);   // ext/hip_lazy.c: an array that dies with a pending chain releases the chain's inputs
    if (array->refcount > 0) array->refcount--;
    if (array->refcount == 0) {
        if (array->data != nullptr && array->base == nullptr) {
            if (array->device == NDARRAY_DEVICE_CPU)
                free(array->data);
            else
                (void)np_free(array->data);   // vfree
        }
        if (array->base != nullptr) NDArray_FREE(array->base);
        array->refcount = -1;
        free_header(array);
    }
}

NDArray *NDArray_ToGPU(NDArray *target) {   // ndarray.c:1037-1068
    if (!target) return nullptr;
    int count = 0;
    if (np_device_count(&count) != NP_OK || count <= 0) {
        throw_error("No GPU device available or CUDA not enabled");   // numpower.c:525
        return nullptr;
    }
    if (target->device == NDARRAY_DEVICE_GPU) return NDArray_Copy(target, NDARRAY_DEVICE_GPU);
    NDArray *r = new_array(target->dimensions, target->ndim, NDARRAY_DEVICE_GPU, false);
    if (!r) return nullptr;
    if (!dev_ok(np_memcpy_h2d(r->data, target->data, (size_t)NDArray_NUMELEMENTS(target) * sizeof(float)))) {
        NDArray_FREE(r);
        return nullptr;
    }
    return r;
}

NDArray *NDArray_ToCPU(NDArray *target) {   // ndarray.c:1075-1093
    if (!target) return nullptr;
    if (target->device == NDARRAY_DEVICE_CPU) return NDArray_Copy(target, NDARRAY_DEVICE_CPU);
    NDArray *r = new_array(target->dimensions, target->ndim, NDARRAY_DEVICE_CPU, false);
    if (!r) return nullptr;
    if (!dev_ok(np_memcpy_d2h(r->data, target->data, (size_t)NDArray_NUMELEMENTS(target) * sizeof(float)))) {
        NDArray_FREE(r);
        return nullptr;
    }
    return r;
}

float NDArray_GetFloatScalar(NDArray *a) {   // ndarray.c:1302-1310
    if (a->device == NDARRAY_DEVICE_CPU) return NDArray_FDATA(a)[0];
    float v = 0.0f;
    (void)dev_ok(np_read_float(NDArray_FDATA(a), 0, &v));   // NDArray_VFLOAT
    return v;
}

int NDArray_CopyToHostBuffer(NDArray *a, float *host_out) {
    if (!a || !host_out) return -1;
    if (a->device != NDARRAY_DEVICE_CPU) {   // numpower.c:466
        throw_error("NDArray must be on CPU RAM before it can be converted to a PHP array.");
        return -1;
    }
    memcpy(host_out, a->data, (size_t)NDArray_NUMELEMENTS(a) * sizeof(float));
    return 0;
}

long NDArray_LiveDeviceAllocations(void) { return np_live_allocs(); }

/* ---- binary ---- */
NDArray *NDArray_Add_Float(NDArray *a, NDArray *b) { return binary_op(NP_ADD, a, b); }
NDArray *NDArray_Subtract_Float(NDArray *a, NDArray *b) { return binary_op(NP_SUBTRACT, a, b); }
NDArray *NDArray_Multiply_Float(NDArray *a, NDArray *b) { return binary_op(NP_MULTIPLY, a, b); }
NDArray *NDArray_Divide_Float(NDArray *a, NDArray *b) { return binary_op(NP_DIVIDE, a, b); }
NDArray *NDArray_Mod_Float(NDArray *a, NDArray *b) { return binary_op(NP_MOD, a, b); }
NDArray *NDArray_Pow_Float(NDArray *a, NDArray *b) { return binary_op(NP_POW, a, b); }

/* ---- comparisons (logic.c:67-670) ---- */
NDArray *NDArray_Equal(NDArray *nda, NDArray *ndb) { return binary_op(NP_EQUAL, nda, ndb); }
NDArray *NDArray_NotEqual(NDArray *nda, NDArray *ndb) { return binary_op(NP_NOT_EQUAL, nda, ndb); }
NDArray *NDArray_Greater(NDArray *nda, NDArray *ndb) { return binary_op(NP_GREATER, nda, ndb); }
NDArray *NDArray_GreaterEqual(NDArray *nda, NDArray *ndb) { return binary_op(NP_GREATER_EQUAL, nda, ndb); }
NDArray *NDArray_Less(NDArray *nda, NDArray *ndb) { return binary_op(NP_LESS, nda, ndb); }
NDArray *NDArray_LessEqual(NDArray *nda, NDArray *ndb) { return binary_op(NP_LESS_EQUAL, nda, ndb); }
/* ndarray.c:853-931: the reference throws "NDArray_Maximum not implemented for GPU" */
NDArray *NDArray_Maximum(NDArray *a, NDArray *b) { return binary_op(NP_MAXIMUM, a, b); }
NDArray *NDArray_Minimum(NDArray *a, NDArray *b) { return binary_op(NP_MINIMUM, a, b); }

/* ---- layout (manipulation.c:68-130) ---- */
NDArray *NDArray_Transpose(NDArray *a, NDArray_Dims *permute) {
    if (!a) return nullptr;
    const int n = NDArray_NDIM(a);
    int permutation[128];
    if (n > 128) {
        throw_error("axes don't match array");
        return nullptr;
    }
    if (permute == nullptr) {
        for (int i = 0; i < n; ++i) permutation[i] = n - 1 - i;   // manipulation.c:75-80
    } else {
        if (permute->len != n) {
            throw_error("axes don't match array");                 // manipulation.c:84-87
            return nullptr;
        }
        int reverse[128];
        for (int i = 0; i < n; ++i) reverse[i] = -1;
        for (int i = 0; i < n; ++i) {
            int axis = permute->ptr[i];
            if (axis < 0) axis += n;                               // check_and_adjust_axis
            if (axis < 0 || axis >= n) {
                throw_error("axes don't match array");
                return nullptr;
            }
            if (reverse[axis] != -1) {
                throw_error("repeated axis in transpose");         // manipulation.c:97-100
                return nullptr;
            }
            reverse[axis] = i;
            permutation[i] = axis;
        }
    }
    if (!require_gpu(a, "transpose")) return nullptr;
    int out_shape[128];
    for (int i = 0; i < n; ++i) out_shape[i] = a->dimensions[permutation[i]];
    NDArray *ret = new_array(out_shape, n, NDARRAY_DEVICE_GPU, false);
    if (!ret) return nullptr;
    if (!dev_ok(np_permute(NDArray_FDATA(a), NDArray_FDATA(ret), n, a->dimensions, permutation))) {
        NDArray_FREE(ret);
        return nullptr;
    }
    return ret;
}

/* ---- initializers (src/initializers.c:379-510,633-660,818-841; the reference's phpbench suite) ---- */
// The reference creates every array on the CPU; `...On(…, device)` lets it be born on the GPU
// (np_fill / np_identity / np_arange: no PCIe copy).  With NDARRAY_DEVICE_CPU the fills are plain
// store loops (placement plumbing); arange on the CPU stays the reference's own loop.
NDArray *NDArray_FullOn(const int *shape, int ndim, double fill_value, int device) {   // initializers.c:655-660
    if (ndim > 0 && !shape) return nullptr;
    NDArray *rtn = new_array(shape, ndim, device, false);
    if (!rtn) return nullptr;
    if (!NDArray_Fill(rtn, (float)fill_value)) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}
NDArray *NDArray_Full(const int *shape, int ndim, double fill_value) {
    return NDArray_FullOn(shape, ndim, fill_value, NDARRAY_DEVICE_CPU);
}
NDArray *NDArray_Ones(const int *shape, int ndim, const char *type) {   // initializers.c:458-470
    (void)type;
    return NDArray_FullOn(shape, ndim, 1.0, NDARRAY_DEVICE_CPU);
}

NDArray *NDArray_IdentityOn(int size, int device) {   // initializers.c:479-510
    if (size < 0) {
        throw_error("negative dimensions are not allowed");
        return nullptr;
    }
    if (size == 0) {
        const int shape[1] = {0};
        return new_array(shape, 1, device, false);
    }
    const int shape[2] = {size, size};
    if (device == NDARRAY_DEVICE_GPU) {
        NDArray *rtn = new_array(shape, 2, device, false);
        if (rtn && !dev_ok(np_identity(NDArray_FDATA(rtn), (size_t)size))) {
            NDArray_FREE(rtn);
            return nullptr;
        }
        return rtn;
    }
    NDArray *rtn = new_array(shape, 2, device, true);
    if (rtn)
        for (int i = 0; i < size; ++i) NDArray_FDATA(rtn)[(size_t)i * size + i] = 1.0f;
    return rtn;
}
NDArray *NDArray_Identity(int size) { return NDArray_IdentityOn(size, NDARRAY_DEVICE_CPU); }

NDArray *NDArray_ArangeOn(double start, double stop, double step, int device) {   // initializers.c:818-841
    const double ivalue = ceil((stop - start) / step);   // safe_ceil_to_int, initializers.c:797-807
    if (!(ivalue >= (double)INT_MIN && ivalue <= (double)INT_MAX)) {
        throw_error("arange: overflow while computing length");
        return nullptr;
    }
    const int length = (int)ivalue;
    if (length <= 0) {
        throw_error("arange: zero length");
        return nullptr;
    }
    if (device != NDARRAY_DEVICE_GPU) {
        throw_error("arange: operand is on the CPU; numpower_amd only computes on the GPU "
                    "(the CPU path is the reference's own)");
        return nullptr;
    }
    const int shape[1] = {length};
    NDArray *rtn = new_array(shape, 1, device, false);
    if (rtn && !dev_ok(np_arange(NDArray_FDATA(rtn), start, step, (size_t)length))) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}

/* ---- views, layout and equality around the path ---- */
namespace {

// NDArray_FromNDArrayBase (initializers.c): header over someone else's buffer
NDArray *make_view(NDArray *base, char *data, const int *shape, int ndim) {
    NDArray *r = make_header(shape, ndim, base->device);
    r->data = data;
    r->base = base;
    base->refcount++;   // NDArray_ADDREF
    return r;
}

bool is_c_contiguous(const NDArray *a) {
    if (!strides_fit(a)) return true;   // only make_strides writes the marker: a fresh C-order layout
    int expect = (int)sizeof(float);
    for (int i = a->ndim - 1; i >= 0; --i) {
        if (a->dimensions[i] != 1 && a->strides[i] != expect) return false;
        expect *= a->dimensions[i];
    }
    return true;
}

// gather `shape` elements starting at `data` with byte strides into a fresh contiguous GPU array
NDArray *gather_to_new(const NDArray *like, const char *data, const int *shape, const int *byte_strides, int ndim) {
    NDArray *r = new_array(shape, ndim, NDARRAY_DEVICE_GPU, false);
    if (!r) return nullptr;
    long long st[NP_MAX_ND_HOST];
    for (int i = 0; i < ndim; ++i) st[i] = (long long)byte_strides[i] / (long long)sizeof(float);
    (void)like;
    if (!dev_ok(np_strided_copy((const float *)data, NDArray_FDATA(r), ndim, shape, st))) {
        NDArray_FREE(r);
        return nullptr;
    }
    return r;
}

}  // namespace

int NDArray_ArrayEqual(NDArray *a, NDArray *b) {   // logic.c:703-716
    if (!a || !b) return 0;
    if (!same_shape(a, b)) return 0;
    if (!require_gpu(a, "array_equal") || !require_gpu(b, "array_equal")) return 0;
    int any = 0;
    if (!dev_ok(np_count_mismatch(NP_MISMATCH_EXACT, NDArray_FDATA(a), NDArray_FDATA(b),
                                  (size_t)NDArray_NUMELEMENTS(a), 0.0f, 0.0f, &any)))
        return 0;
    return any ? 0 : 1;
}

int NDArray_AllClose(NDArray *a, NDArray *b, float rtol, float atol) {   // logic.c:750-772
    if (!a || !b) return -1;
    if (!same_shape(a, b)) {
        throw_error("Shape mismatch");
        return -1;
    }
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b)) {
        throw_error("NDArray::allclose() requires both arrays to be on the same device (CPU or GPU).");
        return -1;
    }
    // the reference stops here for GPU arrays ("`allclose` is not compatible with GPU operations.")
    if (!require_gpu(a, "allclose")) return -1;
    int any = 0;
    if (!dev_ok(np_count_mismatch(NP_MISMATCH_ALLCLOSE, NDArray_FDATA(a), NDArray_FDATA(b),
                                  (size_t)NDArray_NUMELEMENTS(a), rtol, atol, &any)))
        return -1;
    return any ? 0 : 1;
}

NDArray *NDArray_ToContiguous(NDArray *a) {   // manipulation.c:381-421
    if (!a) return nullptr;
    if (!require_gpu(a, "ToContiguous")) return nullptr;
    if (a->ndim > NP_MAX_ND_HOST) {
        throw_error("ToContiguous: more than %d dimensions", NP_MAX_ND_HOST);
        return nullptr;
    }
    if (!strides_fit(a)) return NDArray_Copy(a, NDArray_DEVICE(a));   // a fresh C-order layout (make_strides): already contiguous
    return gather_to_new(a, a->data, a->dimensions, a->strides, a->ndim);
}

// The reference ignores `offset` and takes shape[last] elements (indexing.c:29-33), which walks off
// the buffer when rows < cols; min(rows, cols) is the same wherever the reference is defined.
NDArray *NDArray_Diagonal(NDArray *target, int offset) {   // indexing.c:21-48
    (void)offset;
    if (!target) return nullptr;
    if (NDArray_NDIM(target) != 2) {
        throw_error("NDArray_Diagonal: Array must be 2-d.");
        return nullptr;
    }
    if (!require_gpu(target, "diagonal")) return nullptr;
    if (!strides_fit(target)) {
        throw_error("diagonal: rows of 2 GiB or more do not fit the int byte strides of an NDArray");
        return nullptr;
    }
    const int rows = target->dimensions[0], cols = target->dimensions[1];
    const int shape[1] = {rows < cols ? rows : cols};
    const int stride[1] = {target->strides[0] + target->strides[1]};
    return gather_to_new(target, target->data, shape, stride, 1);
}

NDArray *NDArray_Trace(NDArray *a) {   // linalg.c:758-767
    NDArray *diagonal = NDArray_Diagonal(a, 0);
    if (!diagonal) return nullptr;
    bool ok = false;
    const float result = reduce_all(diagonal, NP_SUM, "trace", &ok);
    NDArray_FREE(diagonal);
    if (!ok) return nullptr;
    return NDArray_CreateFromFloatScalar(result);
}

NDArray *NDArray_Reshape(NDArray *target, int *new_shape, int ndim) {   // manipulation.c:138-162
    if (!target) return nullptr;
    if (new_shape == nullptr) {
        throw_error("new shape cannot be null.");
        return nullptr;
    }
    long total = 1;
    for (int i = 0; i < ndim; ++i) total *= new_shape[i];
    if (total != NDArray_NUMELEMENTS(target)) {
        throw_error("incompatible shape in reshape call.");
        return nullptr;
    }
    return make_view(target, target->data, new_shape, ndim);   // shares data, ADDREFs target
}

// The reference's struct holds extents as `int` (ndarray.h:52-74): an axis cannot be longer than INT_MAX whatever the device holds.
static bool fits_extent(long n, const char *what) {
    if (n > 2147483647L) {
        throw_error(what);
        return false;
    }
    return true;
}

NDArray *NDArray_Flatten(NDArray *target) {   // manipulation.c:169-184: always a copy
    if (!target) return nullptr;
    if (!fits_extent(NDArray_NUMELEMENTS(target), "flatten: more than 2^31 - 1 elements do not fit one axis of an NDArray")) return nullptr;
    NDArray *rtn = NDArray_Copy(target, NDArray_DEVICE(target));
    if (!rtn) return nullptr;
    const int n = (int)NDArray_NUMELEMENTS(target);
    rtn->ndim = 1;
    rtn->dimensions[0] = n;
    rtn->strides[0] = (int)sizeof(float);
    return rtn;
}

NDArray *NDArray_ExpandDim(NDArray *a, NDArray *axis) {   // manipulation.c:453-513
    if (!a || !axis) return nullptr;
    if (NDArray_DEVICE(axis) != NDARRAY_DEVICE_CPU) {
        throw_error("expand_dims: axis must be a CPU scalar or vector");
        return nullptr;
    }
    if (NDArray_NDIM(axis) > 1) {
        throw_error("axis must be either a scalar or a vector. Found matrix with %d dimensions.", NDArray_NDIM(axis));
        return nullptr;
    }
    const int n_axis = (int)NDArray_NUMELEMENTS(axis);
    const int out_ndim = n_axis + NDArray_NDIM(a);
    if (out_ndim > NP_MAX_ND_HOST) {
        throw_error("expand_dims: more than %d dimensions", NP_MAX_ND_HOST);
        return nullptr;
    }
    int norm[NP_MAX_ND_HOST];
    for (int i = 0; i < n_axis; ++i) {
        int ax = (int)NDArray_FDATA(axis)[i];
        if (ax < -out_ndim || ax >= out_ndim) {   // check_and_adjust_axis, manipulation.c:41-54
            throw_error("invalid axis or axes provided.");
            return nullptr;
        }
        if (ax < 0) ax += out_ndim;
        norm[i] = ax;
    }
    int out_shape[NP_MAX_ND_HOST];
    int it = 0;
    for (int ax = 0; ax < out_ndim; ++ax) {
        bool found = false;
        for (int i = 0; i < n_axis; ++i) found = found || norm[i] == ax;
        if (found) {
            out_shape[ax] = 1;
        } else {
            // repeated axes leave fewer slots than a has dimensions; the reference then reads past
            // a's shape — report it instead
            if (it >= NDArray_NDIM(a)) {
                throw_error("invalid axis or axes provided.");
                return nullptr;
            }
            out_shape[ax] = a->dimensions[it++];
        }
    }
    return NDArray_Reshape(a, out_shape, out_ndim);
}

NDArray *NDArray_ConcatenateFlat(NDArray **arrays, int num_arrays) {   // manipulation.c:293-361
    if (num_arrays <= 0 || !arrays) {
        throw_error("need at least one array to concatenate");
        return nullptr;
    }
    long total = 0;
    for (int i = 0; i < num_arrays; ++i) {
        if (!arrays[i]) return nullptr;
        total += NDArray_NUMELEMENTS(arrays[i]);
        if (total > 0x7fffffffL) {
            throw_error("total number of elements too large to concatenate");
            return nullptr;
        }
    }
    if (!require_gpu(arrays[0], "append")) return nullptr;
    const int shape[1] = {(int)total};
    NDArray *ret = new_array(shape, 1, NDARRAY_DEVICE_GPU, false);
    if (!ret) return nullptr;
    char *dst = ret->data;
    for (int i = 0; i < num_arrays; ++i) {
        const size_t bytes = (size_t)NDArray_NUMELEMENTS(arrays[i]) * sizeof(float);
        // 0-d operands are host scalars (manipulation.c:344-347); everything else must be on the GPU
        int rc;
        if (NDArray_DEVICE(arrays[i]) == NDARRAY_DEVICE_GPU) {
            rc = np_memcpy_d2d(dst, arrays[i]->data, bytes);
        } else if (NDArray_NDIM(arrays[i]) == 0) {
            rc = np_memcpy_h2d(dst, arrays[i]->data, bytes);
        } else {
            throw_error("Device mismatch, both NDArray MUST be in the same device.");
            NDArray_FREE(ret);
            return nullptr;
        }
        if (!dev_ok(rc)) {
            NDArray_FREE(ret);
            return nullptr;
        }
        dst += bytes;
    }
    return ret;
}

NDArray *NDArray_Append(NDArray **arrays, int axis, int num_arrays) {   // manipulation.c:368-374
    if (axis == -1) return NDArray_ConcatenateFlat(arrays, num_arrays);
    // the reference returns NULL silently here (manipulation.c:368-374: only axis -1 is implemented)
    throw_error("append: only axis -1 (flattened) is implemented");
    return nullptr;
}

/* ---- manipulation wrappers around the layout kernels (manipulation.c:554-1073, initializers.c:597-625) ---- */
namespace {

bool adjust_axis(int *axis, int ndim) {   // check_and_adjust_axis_msg, manipulation.c:40-54
    if (*axis < -ndim || *axis >= ndim) {
        throw_error("Axis is out of bounds for array dimension");
        return false;
    }
    if (*axis < 0) *axis += ndim;
    return true;
}

NDArray *transpose_to(NDArray *a, const int *order, int n) {
    NDArray_Dims dims;
    dims.ptr = const_cast<int *>(order);
    dims.len = n;
    return NDArray_Transpose(a, &dims);
}

}  // namespace

NDArray *NDArray_AtLeast1D(NDArray *a) {   // manipulation.c:555-571
    if (!a) return nullptr;
    if (NDArray_NDIM(a) == 0) {
        int shape[1] = {1};
        return NDArray_Reshape(a, shape, 1);
    }
    return NDArray_Reshape(a, a->dimensions, a->ndim);
}

NDArray *NDArray_AtLeast2D(NDArray *a) {   // manipulation.c:574-589
    if (!a) return nullptr;
    if (NDArray_NDIM(a) < 2) {
        int shape[2] = {1, (int)NDArray_NUMELEMENTS(a)};
        return NDArray_Reshape(a, shape, 2);
    }
    return NDArray_Reshape(a, a->dimensions, a->ndim);
}

// manipulation.c:592-615.  0-d / 1-d -> (1, n, 1) as in the reference.  For a 2-d (r, c) input the
// reference stores a third extent into a two-int allocation (heap overflow, :594-604) on its way to
// (1, r, c); undefined there, numpy's (r, c, 1) here.
NDArray *NDArray_AtLeast3D(NDArray *a) {
    if (!a) return nullptr;
    if (NDArray_NDIM(a) < 2) {
        int shape[3] = {1, (int)NDArray_NUMELEMENTS(a), 1};
        return NDArray_Reshape(a, shape, 3);
    }
    if (NDArray_NDIM(a) == 2) {
        int shape[3] = {a->dimensions[0], a->dimensions[1], 1};
        return NDArray_Reshape(a, shape, 3);
    }
    return NDArray_Reshape(a, a->dimensions, a->ndim);
}

NDArray *NDArray_Squeeze(NDArray *a, NDArray *axis) {   // manipulation.c:732-776, :618-729
    if (!a) return nullptr;
    const int nd = NDArray_NDIM(a);
    if (nd > NP_MAX_ND_HOST) {
        throw_error("squeeze: more than %d dimensions", NP_MAX_ND_HOST);
        return nullptr;
    }
    bool drop[NP_MAX_ND_HOST] = {false};
    if (axis) {   // NDArray_ConvertMultiAxis + NDArray_SqueezeSelected
        if (NDArray_DEVICE(axis) != NDARRAY_DEVICE_CPU) {
            throw_error("squeeze: axis must be a CPU scalar or vector");
            return nullptr;
        }
        const int naxes = (int)NDArray_NUMELEMENTS(axis);
        for (int i = 0; i < naxes; ++i) {
            int ax = (int)NDArray_FDATA(axis)[i];
            if (nd == 0 && NDArray_NDIM(axis) == 0 && (ax == 0 || ax == -1)) continue;
            if (!adjust_axis(&ax, nd)) return nullptr;
            if (drop[ax]) {
                throw_error("duplicate value in 'axis'");
                return nullptr;
            }
            drop[ax] = true;
        }
        for (int i = 0; i < nd; ++i)
            if (drop[i] && a->dimensions[i] != 1) {
                throw_error("cannot select an axis to squeeze out which has size not equal to one");
                return nullptr;
            }
    } else {
        for (int i = 0; i < nd; ++i) drop[i] = a->dimensions[i] == 1;
    }
    int shape[NP_MAX_ND_HOST], out = 0;
    for (int i = 0; i < nd; ++i)
        if (!drop[i]) shape[out++] = a->dimensions[i];
    return NDArray_Reshape(a, shape, out);
}

NDArray *NDArray_SwapAxes(NDArray *a, int axis1, int axis2) {   // manipulation.c:779-803
    if (!a) return nullptr;
    const int n = NDArray_NDIM(a);
    if (!adjust_axis(&axis1, n) || !adjust_axis(&axis2, n)) return nullptr;
    int order[128];
    for (int i = 0; i < n; ++i) order[i] = i;
    order[axis1] = axis2;
    order[axis2] = axis1;
    return transpose_to(a, order, n);
}

// manipulation.c:806-845.  The reference builds the order by shifting the identity right from
// `start` and writing `axis` there, without taking `axis` out of its old place: right whenever the
// axis moves towards the front (the common rollaxis(a, 2) / rollaxis(a, 2, 1)), a "repeated axis"
// error when it moves back.  numpy's definition here (remove, then insert): same results wherever
// the reference has one.
NDArray *NDArray_Rollaxis(NDArray *a, int axis, int start) {
    if (!a) return nullptr;
    const int n = NDArray_NDIM(a);
    if (!adjust_axis(&axis, n)) return nullptr;
    if (start < 0) start += n;
    if (start < 0 || start > n) {
        throw_error("'%s' arg requires %d <= %s < %d, but %d was passed in", "start", -n, "start", n + 1, start);
        return nullptr;
    }
    if (axis < start) start -= 1;
    if (axis == start) return NDArray_Copy(a, NDArray_DEVICE(a));
    int order[128], m = 0;
    for (int i = 0; i < n; ++i)
        if (i != axis) order[m++] = i;
    for (int i = n - 1; i > start; --i) order[i] = order[i - 1];
    order[start] = axis;
    return transpose_to(a, order, n);
}

// manipulation.c:849-891.  The reference lists the untouched axes and then OVERWRITES order[dest[i]]
// with src[i] (instead of inserting), which is only right when every destination lies at the end of
// that list; numpy's definition (insert in order of destination) agrees with it there.
NDArray *NDArray_Moveaxis(NDArray *a, int *src, int *dest, int n_source, int n_dest) {
    if (!a || !src || !dest) return nullptr;
    const int n = NDArray_NDIM(a);
    if (n_source != n_dest) {
        throw_error("`source` and `destination` must have the same number of elements.");
        return nullptr;
    }
    if (n > 128 || n_source > n) {
        throw_error("Axis is out of bounds for array dimension");
        return nullptr;
    }
    int s[128], d[128];
    for (int i = 0; i < n_source; ++i) {
        s[i] = src[i];
        d[i] = dest[i];
        if (!adjust_axis(&s[i], n) || !adjust_axis(&d[i], n)) return nullptr;
        for (int j = 0; j < i; ++j)
            if (s[j] == s[i] || d[j] == d[i]) {
                throw_error("repeated axis in `source` or `destination`");
                return nullptr;
            }
    }
    int order[128];
    for (int i = 0; i < n; ++i) order[i] = -1;
    for (int i = 0; i < n_source; ++i) order[d[i]] = s[i];
    int next = 0;
    for (int i = 0; i < n; ++i) {
        if (order[i] != -1) continue;
        bool moved = true;
        while (moved) {
            moved = false;
            for (int j = 0; j < n_source; ++j)
                if (s[j] == next) { ++next; moved = true; }
        }
        order[i] = next++;
    }
    return transpose_to(a, order, n);
}

NDArray *NDArray_Concatenate(NDArray **arrays, int narrays, int axis) {   // manipulation.c:895-995
    if (narrays <= 0 || !arrays || !arrays[0]) {
        throw_error("need at least one array to concatenate");
        return nullptr;
    }
    const int ndim = NDArray_NDIM(arrays[0]);
    if (ndim == 0) {
        throw_error("zero-dimensional arrays cannot be concatenated");
        return nullptr;
    }
    if (!adjust_axis(&axis, ndim)) return nullptr;
    if (ndim > NP_MAX_ND_HOST) {
        throw_error("concatenate: more than %d dimensions", NP_MAX_ND_HOST);
        return nullptr;
    }
    int shape[NP_MAX_ND_HOST];
    memcpy(shape, arrays[0]->dimensions, sizeof(int) * ndim);
    for (int i = 1; i < narrays; ++i) {
        if (!arrays[i]) return nullptr;
        if (NDArray_NDIM(arrays[i]) != ndim) {
            throw_error("all the input arrays must have same number of dimensions, but the array at index %d has %d "
                        "dimension(s) and the array at index %d has %d dimension(s)", 0, ndim, i, NDArray_NDIM(arrays[i]));
            return nullptr;
        }
        for (int d = 0; d < ndim; ++d) {
            if (d == axis) shape[d] += arrays[i]->dimensions[d];
            else if (shape[d] != arrays[i]->dimensions[d]) {
                throw_error("all the input array dimensions except for the concatenation axis must match exactly, but "
                            "along dimension %d, the array at index %d has size %d and the array at index %d has size %d",
                            d, 0, shape[d], i, arrays[i]->dimensions[d]);
                return nullptr;
            }
        }
    }
    for (int i = 0; i < narrays; ++i) {
        if (NDArray_DEVICE(arrays[i]) != NDArray_DEVICE(arrays[0])) {
            throw_error("Device mismatch, both NDArray MUST be in the same device.");
            return nullptr;
        }
        if (!require_gpu(arrays[i], "concatenate")) return nullptr;
    }
    NDArray *ret = new_array(shape, ndim, NDARRAY_DEVICE_GPU, false);
    if (!ret) return nullptr;
    size_t outer = 1, inner = 1;
    for (int d = 0; d < axis; ++d) outer *= (size_t)shape[d];
    for (int d = axis + 1; d < ndim; ++d) inner *= (size_t)shape[d];
    const size_t dst_pitch = (size_t)shape[axis] * inner;
    size_t offset = 0;
    for (int i = 0; i < narrays; ++i) {   // one pitched copy per input: its rows are slabs of the result's rows
        const size_t width = (size_t)arrays[i]->dimensions[axis] * inner;
        if (!dev_ok(np_copy2d(NDArray_FDATA(ret) + offset, dst_pitch, NDArray_FDATA(arrays[i]), width, width, outer))) {
            NDArray_FREE(ret);
            return nullptr;
        }
        offset += width;
    }
    return ret;
}

namespace {

// concatenate(f(arrays[i])...) with the temporaries released afterwards
NDArray *stack_with(NDArray **arrays, int narrays, NDArray *(*prepare)(NDArray *), int axis_if_1d, int axis) {
    if (narrays <= 0 || !arrays) {
        throw_error("need at least one array to concatenate");
        return nullptr;
    }
    std::vector<NDArray *> parsed((size_t)narrays, nullptr);
    NDArray *result = nullptr;
    bool ok = true;
    for (int i = 0; i < narrays && ok; ++i) {
        parsed[(size_t)i] = arrays[i] ? prepare(arrays[i]) : nullptr;
        ok = parsed[(size_t)i] != nullptr;
    }
    if (ok) result = NDArray_Concatenate(parsed.data(), narrays, NDArray_NDIM(parsed[0]) == 1 ? axis_if_1d : axis);
    for (NDArray *p : parsed)
        if (p) NDArray_FREE(p);
    return result;
}

NDArray *column_of(NDArray *a) {   // NDArray_ColumnStack's per-input step (manipulation.c:1059-1063)
    NDArray *two = NDArray_AtLeast2D(a);
    if (!two) return nullptr;
    NDArray *t = NDArray_Transpose(two, nullptr);
    NDArray_FREE(two);
    return t;
}

}  // namespace

NDArray *NDArray_VSTACK(NDArray **arrays, int narrays) { return stack_with(arrays, narrays, NDArray_AtLeast2D, 0, 0); }   // :999-1012
NDArray *NDArray_HSTACK(NDArray **arrays, int narrays) { return stack_with(arrays, narrays, NDArray_AtLeast1D, 0, 1); }   // :1015-1034
NDArray *NDArray_DSTACK(NDArray **arrays, int narrays) { return stack_with(arrays, narrays, NDArray_AtLeast3D, 2, 2); }   // :1037-1052
// :1055-1073: every input goes through atleast_2d + transpose, so 2-d inputs are stacked TRANSPOSED (numpy leaves
// them as they are); kept as the reference has it
NDArray *NDArray_ColumnStack(NDArray **arrays, int narrays) { return stack_with(arrays, narrays, column_of, 1, 1); }

NDArray *NDArray_Diag(NDArray *a) {   // initializers.c:597-625
    if (!a) return nullptr;
    if (NDArray_NDIM(a) != 1 && NDArray_NDIM(a) != 2) {
        throw_error("Input array must be a vector or 2-dimensional");
        return nullptr;
    }
    if (NDArray_NDIM(a) == 2) return NDArray_Diagonal(a, 0);
    if (!require_gpu(a, "diag")) return nullptr;
    if (!fits_extent(NDArray_NUMELEMENTS(a), "diag: operand too long for one axis of an NDArray")) return nullptr;
    const int n = (int)NDArray_NUMELEMENTS(a);
    const int shape[2] = {n, n};
    NDArray *rtn = new_array(shape, 2, NDARRAY_DEVICE_GPU, true);
    if (!rtn) return nullptr;
    // element i goes to i * (n + 1): a pitched copy of n one-float rows
    if (n > 0 && !dev_ok(np_copy2d(NDArray_FDATA(rtn), (size_t)n + 1, NDArray_FDATA(a), 1, 1, (size_t)n))) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}

// NDArray_Slice (manipulation.c:193-283): indexes[i] is a small CPU array [start], [start, stop] or
// [start, stop, step] for axis i (Slice_GetIndices, indexing.c:59-108).  One-element indexes select
// and drop the axis.  The reference returns a strided VIEW for a single index and a gathered copy
// for several; because every hot-path op reads `numElements` contiguous floats from `data`
// (SURVEY.md §8a), a strided view would be misread by the next op — here the result is a view only
// when it is contiguous, otherwise ONE np_strided_copy launch.
NDArray *NDArray_Slice(NDArray *array, NDArray **indexes, int num_indices) {
    if (!array || !indexes) return nullptr;
    if (num_indices > NDArray_NDIM(array)) {
        throw_error("too many indices for array.");
        return nullptr;
    }
    const int nd = NDArray_NDIM(array);
    if (nd > NP_MAX_ND_HOST) {
        throw_error("slice: more than %d dimensions", NP_MAX_ND_HOST);
        return nullptr;
    }
    if (!strides_fit(array)) {
        throw_error("slice: rows of 2 GiB or more do not fit the int byte strides of an NDArray");
        return nullptr;
    }
    int new_shape[NP_MAX_ND_HOST], new_strides[NP_MAX_ND_HOST];
    int out_nd = 0;
    char *data_ptr = array->data;
    for (int i = 0; i < nd; ++i) {
        const int length = array->dimensions[i];
        if (i >= num_indices) {   // untouched trailing axes
            new_shape[out_nd] = length;
            new_strides[out_nd++] = array->strides[i];
            continue;
        }
        NDArray *ix = indexes[i];
        if (!ix || NDArray_DEVICE(ix) != NDARRAY_DEVICE_CPU) {
            throw_error("Slicing error");
            return nullptr;
        }
        const long cnt = NDArray_NUMELEMENTS(ix);
        const bool has_start = cnt >= 1, has_stop = cnt >= 2, has_step = cnt == 3;
        int step = has_step ? (int)NDArray_FDATA(ix)[2] : 1;
        if (step == 0) {
            throw_error("slice step cannot be zero");
            return nullptr;
        }
        int start, stop, n_steps;
        if (!has_start) {
            start = step < 0 ? length - 1 : 0;
        } else {
            start = (int)NDArray_FDATA(ix)[0];
            if (start < 0) start += length;
            if (start < 0) start = step < 0 ? -1 : 0;
            if (start >= length) start = step < 0 ? length - 1 : length;
        }
        if (!has_stop) {
            stop = step < 0 ? -1 : length;
        } else {
            stop = (int)NDArray_FDATA(ix)[1];
            if (stop < 0) stop += length;
            if (stop < 0) stop = -1;
            if (stop > length) stop = length;
        }
        if ((step < 0 && stop >= start) || (step > 0 && start >= stop))
            n_steps = 0;
        else if (step < 0)
            n_steps = (stop - start + 1) / step + 1;
        else
            n_steps = (stop - start - 1) / step + 1;
        if (n_steps <= 0) {   // manipulation.c:231-235
            n_steps = 0;
            step = 1;
            start = 0;
        }
        data_ptr += (long)array->strides[i] * start;
        if (cnt == 1) continue;   // integer index: axis dropped
        new_shape[out_nd] = n_steps;
        new_strides[out_nd++] = array->strides[i] * step;
    }
    // contiguous result -> view; otherwise gather
    int expect = (int)sizeof(float);
    bool contiguous = true;
    for (int i = out_nd - 1; i >= 0; --i) {
        if (new_shape[i] != 1 && new_strides[i] != expect) contiguous = false;
        expect *= new_shape[i];
    }
    if (contiguous || shape_numel(new_shape, out_nd) == 0) return make_view(array, data_ptr, new_shape, out_nd);
    if (!require_gpu(array, "slice")) return nullptr;
    return gather_to_new(array, data_ptr, new_shape, new_strides, out_nd);
}

/* ---- fused elementwise chains ---- */
// What a Zend glue would flush when a lazily built expression reaches toArray()/cpu()/a reduction:
// inputs[0] is the GPU array the chain starts from (it fixes the result shape); the other inputs
// are GPU arrays of the same element count, smaller GPU arrays that broadcast onto it (row vector,
// column, 0-d: the cases of NDArray_Broadcast, ndarray.c:1196-1291) or 0-d CPU scalars.  Quirk
// flags and AVX-body bounds are set exactly as binary_op() above sets them for the stand-alone
// NDArray_*_Float / comparison entry points, so the fused result is bit-identical.
namespace {
// operand classification + quirk flags: ext/hip_lazy.c (NPH_PrepareChain) — the same C that flushes a pending chain in a
// `--with-hip` NumPower tree, so the chains lazy.py builds and the chains the binding builds are
// prepared by one piece of code
using ChainCall = NPH_ChainCall;
bool prepare_chain(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops, ChainCall &c) {
    return NPH_PrepareChain(inputs, nullptr, n_inputs, ops, n_ops, &c) == 0;
}
}  // namespace

NDArray *NDArray_FusedChain(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops) {
    ChainCall c;
    if (!prepare_chain(inputs, n_inputs, ops, n_ops, c)) return nullptr;
    NDArray *first = inputs[0];
    NDArray *result = new_array(first->dimensions, first->ndim, NDARRAY_DEVICE_GPU, false);
    if (!result) return nullptr;
    if (!dev_ok(np_fused_chain(c.ptrs, c.kinds, n_inputs, c.prog, n_ops, NDArray_FDATA(result), c.rows, c.cols))) {
        NDArray_FREE(result);
        return nullptr;
    }
    return result;
}

// ... and with a full reduction as the chain's last step (sum / prod / min / max / mean of an
// expression without materialising it): returns the value, NaN + error on failure.
float NDArray_FusedChainReduce(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops, int reduce_op) {
    ChainCall c;
    if (!prepare_chain(inputs, n_inputs, ops, n_ops, c)) return NAN;
    float v = NAN;
    if (!dev_ok(np_fused_chain_reduce(c.ptrs, c.kinds, n_inputs, c.prog, n_ops, reduce_op, c.rows, c.cols, &v))) return NAN;
    return v;
}

NDArray *NDArray_FusedChainReduceAxis(NDArray **inputs, int n_inputs, const np_fused_op *ops, int n_ops, int reduce_op,
                                      int axis) {
    ChainCall c;
    if (!prepare_chain(inputs, n_inputs, ops, n_ops, c)) return nullptr;
    NDArray *first = inputs[0];
    const int nd = NDArray_NDIM(first);
    if (axis >= nd || axis < 0) {   // ndarray.c:534-538
        throw_error("axis %d is out of bounds for array of dimension %d", axis, nd);
        return nullptr;
    }
    size_t rows = 0, cols = 0;
    int ax = -1;
    // (which axes reduce inside the chain's kernel: ext/hip_lazy.c, shared with the pending chains of a `--with-hip` tree)
    if (!NPH_ChainAxisView(first, axis, &c, &rows, &cols, &ax)) {
        NDArray *value = NDArray_FusedChain(inputs, n_inputs, ops, n_ops);
        if (!value) return nullptr;
        NDArray *r = reduce_axis(value, axis, reduce_op, false);
        NDArray_FREE(value);
        return r;
    }
    int out_shape[128], j = 0;
    for (int i = 0; i < nd; ++i)
        if (i != axis) out_shape[j++] = first->dimensions[i];
    NDArray *result = new_array(out_shape, nd - 1, NDARRAY_DEVICE_GPU, false);
    if (!result) return nullptr;
    if (!dev_ok(np_fused_chain_reduce_axis(c.ptrs, c.kinds, n_inputs, c.prog, n_ops, reduce_op, rows, cols, ax,
                                           NDArray_FDATA(result)))) {
        NDArray_FREE(result);
        return nullptr;
    }
    return result;
}

/* ---- argmax / argmin (calculation.c:73-194) ---- */
// axis == NDARRAY_MAX_DIMS (128) means "flattened" (numpower.c:2588-2590); the reference moves the
// axis last with a Transpose copy and walks rows — here the (outer, axis, inner) view is reduced
// in place, no copy.
NDArray *NDArray_ArgMinMaxCommon(NDArray *op, int axis, bool keepdims, bool is_argmax) {
    if (!op) return nullptr;
    if (!require_gpu(op, is_argmax ? "argmax" : "argmin")) return nullptr;
    const int nd = NDArray_NDIM(op);
    const bool flat = (axis == 128) || nd == 0;
    if (!flat) {
        if (axis < 0) axis += nd;
        if (axis < 0 || axis >= nd) {
            throw_error("Invalid axis parameter");
            return nullptr;
        }
    }
    size_t outer = 1, inner = 1, len = 1;
    int out_shape[128];
    int out_nd = 0;
    if (flat) {
        len = (size_t)NDArray_NUMELEMENTS(op);
        if (keepdims)
            for (int i = 0; i < nd; ++i) out_shape[out_nd++] = 1;
    } else {
        len = (size_t)op->dimensions[axis];
        for (int i = 0; i < nd; ++i) {
            if (i < axis) outer *= (size_t)op->dimensions[i];
            if (i > axis) inner *= (size_t)op->dimensions[i];
            if (i != axis)
                out_shape[out_nd++] = op->dimensions[i];
            else if (keepdims)
                out_shape[out_nd++] = 1;
        }
    }
    if (len == 0) {
        throw_error("attempt to get %s of an empty sequence", is_argmax ? "argmax" : "argmin");
        return nullptr;
    }
    NDArray *rp = new_array(out_shape, out_nd, NDARRAY_DEVICE_GPU, false);
    if (!rp) return nullptr;
    if (!dev_ok(np_argreduce(is_argmax ? 1 : 0, NDArray_FDATA(op), outer, len, inner, NDArray_FDATA(rp)))) {
        NDArray_FREE(rp);
        return nullptr;
    }
    return rp;
}

/* ---- statistics (statistics.c:88-154): 0-d CPU scalars, like NDArray_CreateFromFloatScalar ---- */
NDArray *NDArray_Variance(NDArray *a) {   // statistics.c:117-130
    if (!a || !require_gpu(a, "variance")) return nullptr;
    float mean = 0.0f, m2 = 0.0f;
    if (!dev_ok(np_moments(NDArray_FDATA(a), (size_t)NDArray_NUMELEMENTS(a), &mean, &m2))) return nullptr;
    return NDArray_CreateFromFloatScalar(m2 / NDArray_NUMELEMENTS(a));
}

NDArray *NDArray_Std(NDArray *a) {   // statistics.c:88-108 (the reference rejects GPU arrays here)
    if (!a || !require_gpu(a, "std")) return nullptr;
    float mean = 0.0f, m2 = 0.0f;
    if (!dev_ok(np_moments(NDArray_FDATA(a), (size_t)NDArray_NUMELEMENTS(a), &mean, &m2))) return nullptr;
    return NDArray_CreateFromFloatScalar(sqrtf(m2 / (float)(NDArray_NUMELEMENTS(a))));
}

NDArray *NDArray_Average(NDArray *a, NDArray *weights) {   // statistics.c:131-154
    if (!a || !require_gpu(a, "average")) return nullptr;
    if (weights == nullptr) {
        bool ok = false;
        float s = reduce_all(a, NP_SUM, "average", &ok);
        if (!ok) return nullptr;   // error already raised; never divide the failure value
        return NDArray_CreateFromFloatScalar(s / NDArray_NUMELEMENTS(a));
    }
    if (NDArray_DEVICE(a) != NDArray_DEVICE(weights)) {
        throw_error("All NDArrays used in a operation must be on the same device.");
        return nullptr;
    }
    if (NDArray_NUMELEMENTS(a) != NDArray_NUMELEMENTS(weights)) {
        throw_error("Can't broadcast arrays.");
        return nullptr;
    }
    float s_aw = 0.0f, s_w = 0.0f;
    if (!dev_ok(np_weighted_sums(NDArray_FDATA(a), NDArray_FDATA(weights), (size_t)NDArray_NUMELEMENTS(a), &s_aw, &s_w)))
        return nullptr;
    return NDArray_CreateFromFloatScalar(s_aw / s_w);
}

float NDArray_All(NDArray *a) {   // logic.c:25-58
    if (!a || !require_gpu(a, "all")) return -1.0f;
    int v = 0;
    if (!dev_ok(np_all(NDArray_FDATA(a), (size_t)NDArray_NUMELEMENTS(a), NP_QUIRK_AVX_BODY, &v))) return -1.0f;
    return (float)v;
}

int NDArray_IsBroadcastable(const NDArray *array1, const NDArray *array2) {   // ndarray.c:1124-1162
    const int n1 = array1->ndim, n2 = array2->ndim;
    if (n1 == 1 && n2 > 1) return array1->dimensions[0] == array2->dimensions[n2 - 1];
    if (n1 > 1 && n2 == 1) return array2->dimensions[0] == array1->dimensions[n1 - 1];
    const int maxd = n1 > n2 ? n1 : n2;
    for (int i = 0; i < maxd; ++i) {
        const int s1 = i < n1 ? array1->dimensions[i] : 1;
        const int s2 = i < n2 ? array2->dimensions[i] : 1;
        if (s1 != s2 && s1 != 1 && s2 != 1) return 0;
    }
    return 1;
}

/* ---- unary ---- */
// NDArrayMathGPU_ElementWise{,1F,2F,1N}: ext/hip_math_drivers.c (reference signatures: the op is a
// cuda_float_* function pointer), linked into this library together with ext/hip_math.c and
// ext/gpu_alloc_hip.c.  Their two hooks into the host (np_ext_hooks.h):
extern "C" void np_ext_throw(const char *message) { throw_error("%s", message ? message : ""); }
extern "C" int np_ext_count_device_alloc(int delta) {
    static int count = 0;   // vmalloc/vfree pairs issued through the reference-named layer (vmemcheck)
    count += delta;
    return count;
}
// rsqrt on the device: the reference's PHP_METHOD hands cuda_float_arccos to the driver for GPU arrays
// (numpower.c:1791, a slip — there is no cuda_float_rsqrt); this is float_rsqrt's definition
// (double_math.c:111-126: 0x5f3759df + one Newton step) as its own entry point.
NDArray *NDArray_Rsqrt(NDArray *nda) { return unary_op(nda, NP_RSQRT, 0.0f, 0.0f); }
// exp2 on the device: PHP_METHOD(NDArray, exp2) calls NDArray_Map(nda, float_exp2) whatever the device
// (numpower.c:3153 — a GPU array's device pointer would be dereferenced on the host) and cuda_math.h has no
// cuda_float_exp2; float_exp2 (double_math.c:31-33) as its own entry point.
NDArray *NDArray_Exp2(NDArray *nda) { return unary_op(nda, NP_EXP2, 0.0f, 0.0f); }
NDArray *NDArray_Abs(NDArray *nda) { return unary_op(nda, NP_ABS, 0.0f, 0.0f); }   // arithmetics.c:934-947

/* ---- reductions ---- */
float NDArray_Sum_Float(NDArray *a) { return reduce_all(a, NP_SUM, "sum"); }
float NDArray_Float_Prod(NDArray *a) { return reduce_all(a, NP_PROD, "prod"); }
float NDArray_Mean_Float(NDArray *a) { return reduce_all(a, NP_MEAN, "mean"); }
float NDArray_Min(NDArray *target) { return reduce_all(target, NP_MIN, "min"); }
float NDArray_Max(NDArray *target) { return reduce_all(target, NP_MAX, "max"); }

float NDArray_Median_Float(NDArray *a) {   // arithmetics.c:149-158, calculate_median :111-138
    if (!a || !require_gpu(a, "median")) return -1.0f;
    const size_t n = (size_t)NDArray_NUMELEMENTS(a);
    if (n == 0) {
        throw_error("median of an empty array");
        return -1.0f;
    }
    float t[2];
    if (n % 2 == 0) {
        if (!dev_ok(np_order_stat(NDArray_FDATA(a), n, n / 2 - 1, t))) return -1.0f;
        return (t[0] + t[1]) / 2.0f;
    }
    if (!dev_ok(np_order_stat(NDArray_FDATA(a), n, n / 2, t))) return -1.0f;
    return t[0];
}

NDArray *NDArray_Quantile(NDArray *target, NDArray *q) {   // statistics.c:60-79
    if (!target || !q) return nullptr;
    if (NDArray_NDIM(q) > 0) {
        throw_error("Q must be a scalar");
        return nullptr;
    }
    float quantile = 0.0f;
    if (NDArray_DEVICE(q) == NDARRAY_DEVICE_GPU) {
        if (!dev_ok(np_read_float(NDArray_FDATA(q), 0, &quantile))) return nullptr;
    } else {
        quantile = NDArray_FDATA(q)[0];
    }
    if (quantile < 0 || quantile > 1) {
        throw_error("Q must be between 0 and 1");
        return nullptr;
    }
    if (!require_gpu(target, "quantile")) return nullptr;
    const size_t n = (size_t)NDArray_NUMELEMENTS(target);
    if (n == 0) {
        throw_error("quantile of an empty array");
        return nullptr;
    }
    // calculate_quantile, statistics.c:31-44
    const float index = (float)(n - 1) * quantile;
    const int lower_index = (int)index;
    const float weight = index - (float)lower_index;
    float t[2];
    if (!dev_ok(np_order_stat(NDArray_FDATA(target), n, (size_t)lower_index, t))) return nullptr;
    // (1 - weight) * lower + weight * upper as gcc -march=native compiles it: the second product
    // rounded, the first fused into the sum (vmulss + vfmadd132ss)
    const float value = fmaf(1.0f - weight, t[0], weight * t[1]);
    NDArray *rtn = new_array(nullptr, 0, NDARRAY_DEVICE_GPU, false);
    if (!rtn) return nullptr;
    if (!dev_ok(np_fill(NDArray_FDATA(rtn), value, 1))) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}

NDArray *reduce(NDArray *array, int *axis, NDArray *(*operation)(NDArray *, NDArray *)) {
    const int ax = axis ? *axis : 0;   // ndarray.c:528-532
    if (operation == NDArray_Add_Float) return reduce_axis(array, ax, NP_SUM, false);
    if (operation == NDArray_Multiply_Float) return reduce_axis(array, ax, NP_PROD, true);
    throw_error("reduce: unsupported operation (only NDArray_Add_Float / NDArray_Multiply_Float)");
    return nullptr;
}
NDArray *NDArray_MinAxis(NDArray *target, int axis) { return reduce_axis(target, axis, NP_MIN, false); }
NDArray *NDArray_MaxAxis(NDArray *target, int axis) { return reduce_axis(target, axis, NP_MAX, false); }

/* ---- matmul ---- */
NDArray *NDArray_FMatmul(NDArray *a, NDArray *b) {   // linalg.c:44-82
    if (!require_gpu(a, "matmul")) return nullptr;
    int shape[2] = {a->dimensions[0], b->dimensions[1]};
    NDArray *result = new_array(shape, 2, NDARRAY_DEVICE_GPU, false);
    if (!result) return nullptr;
    if (!dev_ok(np_sgemm((size_t)a->dimensions[0], (size_t)b->dimensions[1], (size_t)a->dimensions[1],
                         NDArray_FDATA(a), NDArray_FDATA(b), NDArray_FDATA(result)))) {
        NDArray_FREE(result);
        return nullptr;
    }
    return result;
}

NDArray *NDArray_Matmul(NDArray *a, NDArray *b) {   // linalg.c:216-245
    if (!a || !b) return nullptr;
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b)) {
        throw_error("Device mismatch, both NDArray MUST be in the same device.");
        return nullptr;
    }
    if (NDArray_NDIM(a) != NDArray_NDIM(b)) {
        throw_error("Arrays must have the same shape. Broadcasting not implemented.");
        return nullptr;
    }
    if (NDArray_NDIM(a) == 0 && NDArray_NDIM(b) == 0) return NDArray_Multiply_Float(a, b);
    if (NDArray_NDIM(a) == 1 && NDArray_NDIM(b) == 1) return NDArray_Dot(a, b);
    if (a->dimensions[a->ndim - 1] != b->dimensions[b->ndim - 2]) {
        throw_error("Shape mismatch for matmul. cols(a) != rows(b)");
        return nullptr;
    }
    if (NDArray_NDIM(a) > 2 && NDArray_NDIM(b) > 2) {
        throw_error("Stack of matrices not allowed");
        return nullptr;
    }
    return NDArray_FMatmul(a, b);
}

NDArray *NDArray_Dot(NDArray *nda, NDArray *ndb) {   // linalg.c:354-393
    if (!nda || !ndb) return nullptr;
    if (NDArray_DEVICE(nda) != NDArray_DEVICE(ndb)) {
        throw_error("Device mismatch, both NDArray MUST be in the same device.");
        return nullptr;
    }
    if (NDArray_NDIM(nda) == 1 && NDArray_NDIM(ndb) == 1) {
        // NDArray_Inner (linalg.c:310-352): sum(a * b) = a 1 x n times n-vector product
        if (!require_gpu(nda, "dot")) return nullptr;
        if (nda->dimensions[0] != ndb->dimensions[0]) {
            throw_error("Shape mismatch for dot");
            return nullptr;
        }
        NDArray *rtn = new_array(nullptr, 0, NDARRAY_DEVICE_GPU, false);
        if (!rtn) return nullptr;
        if (!dev_ok(np_sgemv(1, (size_t)nda->dimensions[0], NDArray_FDATA(nda), NDArray_FDATA(ndb), NDArray_FDATA(rtn)))) {
            NDArray_FREE(rtn);
            return nullptr;
        }
        return rtn;
    }
    if (NDArray_NDIM(nda) == 2 && NDArray_NDIM(ndb) == 2) return NDArray_Matmul(nda, ndb);
    if (NDArray_NDIM(nda) == 0 || NDArray_NDIM(ndb) == 0) return NDArray_Multiply_Float(nda, ndb);
    if (NDArray_NDIM(nda) > 0 && NDArray_NDIM(ndb) == 1) {   // linalg.c:367-386
        if (!require_gpu(nda, "dot")) return nullptr;
        const int nd = nda->ndim;
        const size_t cols = (size_t)nda->dimensions[nd - 1];
        if ((size_t)ndb->dimensions[0] != cols) {
            throw_error("Shape mismatch for dot");
            return nullptr;
        }
        const size_t rows = (size_t)(NDArray_NUMELEMENTS(nda) / (long)cols);
        NDArray *rtn = new_array(nda->dimensions, nd - 1, NDARRAY_DEVICE_GPU, false);
        if (!rtn) return nullptr;
        if (!dev_ok(np_sgemv(rows, cols, NDArray_FDATA(nda), NDArray_FDATA(ndb), NDArray_FDATA(rtn)))) {
            NDArray_FREE(rtn);
            return nullptr;
        }
        return rtn;
    }
    throw_error("Not implemented");   // linalg.c:387-390
    return nullptr;
}

// NDArray_Inner (linalg.c:310-345): Multiply_Float then Sum_Float of EVERYTHING — for N-D operands it
// is not a per-row inner product but one number, shaped (1, ..., 1).  Here: one fused multiply-sum
// pass (np_fused_chain_reduce), the products never go to memory.
NDArray *NDArray_Inner(NDArray *nda, NDArray *ndb) {
    if (!nda || !ndb) return nullptr;
    if (NDArray_NDIM(nda) == 0 && NDArray_NDIM(ndb) == 0) return NDArray_Multiply_Float(nda, ndb);
    if (NDArray_DEVICE(nda) != NDArray_DEVICE(ndb)) {
        throw_error("Device mismatch, both NDArray must be in the same device.");
        return nullptr;
    }
    if (NDArray_NDIM(nda) == 0 || NDArray_NDIM(ndb) == 0 ||
        nda->dimensions[nda->ndim - 1] != ndb->dimensions[ndb->ndim - 1]) {
        throw_error("Shape is not aligned to perform the inner product.");
        return nullptr;
    }
    // the chain starts from the larger operand (the other one broadcasts onto it)
    NDArray *big = NDArray_NUMELEMENTS(nda) >= NDArray_NUMELEMENTS(ndb) ? nda : ndb;
    NDArray *small = big == nda ? ndb : nda;
    NDArray *inputs[2] = {big, small};
    np_fused_op op{};
    op.kind = NP_FUSED_BINARY;
    op.op = NP_MULTIPLY;
    op.operand = 1;
    op.swap = big == nda ? 0 : 1;
    numpower_host_clear_error();
    const float total = NDArray_FusedChainReduce(inputs, 2, &op, 1, NP_SUM);
    if (g_error[0]) return nullptr;
    const int nd = NDArray_NDIM(nda);
    if (nd <= 1) return NDArray_CreateFromFloatScalar(total);
    int ones[NP_MAX_ND_HOST];
    if (nd > NP_MAX_ND_HOST) return NDArray_CreateFromFloatScalar(total);
    for (int i = 0; i < nd; ++i) ones[i] = 1;
    NDArray *rtn = new_array(ones, nd, NDARRAY_DEVICE_CPU, false);
    if (rtn) NDArray_FDATA(rtn)[0] = total;
    return rtn;
}

NDArray *NDArray_Outer(NDArray *a, NDArray *b) {   // linalg.c:724-751
    if (!a || !b) return nullptr;
    if (NDArray_NDIM(a) != 1 || NDArray_NDIM(b) != 1) {
        throw_error("Invalid operation: NDArray::outer() requires both arrays to be 1-dimensional vectors.");
        return nullptr;
    }
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b)) {
        throw_error("NDArray::outer() requires both arrays to be on the same device (CPU or GPU).");
        return nullptr;
    }
    if (!require_gpu(a, "outer")) return nullptr;
    if (!fits_extent(NDArray_NUMELEMENTS(a), "outer: operand too long for one axis of an NDArray") ||
        !fits_extent(NDArray_NUMELEMENTS(b), "outer: operand too long for one axis of an NDArray")) return nullptr;
    const int shape[2] = {(int)NDArray_NUMELEMENTS(a), (int)NDArray_NUMELEMENTS(b)};
    NDArray *rtn = new_array(shape, 2, NDARRAY_DEVICE_GPU, false);   // every element is written: no Zeros pass
    if (!rtn) return nullptr;
    if (!dev_ok(np_outer(NDArray_FDATA(a), (size_t)shape[0], NDArray_FDATA(b), (size_t)shape[1], NDArray_FDATA(rtn)))) {
        NDArray_FREE(rtn);
        return nullptr;
    }
    return rtn;
}

NDArray *NDArray_BatchedMatmul(NDArray *a, NDArray *b) {
    if (!a || !b) return nullptr;
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b)) {
        throw_error("Device mismatch, both NDArray MUST be in the same device.");
        return nullptr;
    }
    if (NDArray_NDIM(a) != 3 || NDArray_NDIM(b) != 3 || a->dimensions[0] != b->dimensions[0]) {
        throw_error("Arrays must have the same shape. Broadcasting not implemented.");
        return nullptr;
    }
    if (a->dimensions[2] != b->dimensions[1]) {
        throw_error("Shape mismatch for matmul. cols(a) != rows(b)");
        return nullptr;
    }
    if (!require_gpu(a, "matmul")) return nullptr;
    const size_t batch = (size_t)a->dimensions[0], M = (size_t)a->dimensions[1], K = (size_t)a->dimensions[2],
                 N = (size_t)b->dimensions[2];
    int shape[3] = {(int)batch, (int)M, (int)N};
    NDArray *result = new_array(shape, 3, NDARRAY_DEVICE_GPU, false);
    if (!result) return nullptr;
    if (!dev_ok(np_sgemm_strided_batched(batch, M, N, K, NDArray_FDATA(a), M * K, NDArray_FDATA(b), K * N,
                                         NDArray_FDATA(result), M * N))) {
        NDArray_FREE(result);
        return nullptr;
    }
    return result;
}

// ---- the sharded form: one call per PHP method, like every other row of the path ----

int NDArray_CommInit(int rank, int world, const char *endpoint) { return dev_ok(np_comm_init(rank, world, endpoint)) ? 0 : -1; }
int NDArray_CommDestroy(void) { return dev_ok(np_comm_destroy()) ? 0 : -1; }
int NDArray_CommRank(void) { return np_comm_world() > 0 ? np_comm_rank() : 0; }
int NDArray_CommWorld(void) { return np_comm_world() > 0 ? np_comm_world() : 1; }

NDArray *NDArray_ShardedBatchedMatmul(NDArray *a, NDArray *b, int batch, int gather_mode) {
    if (!a || !b) return nullptr;
    if (NDArray_DEVICE(a) != NDArray_DEVICE(b)) {
        throw_error("Device mismatch, both NDArray MUST be in the same device.");
        return nullptr;
    }
    if (NDArray_NDIM(a) != 3 || NDArray_NDIM(b) != 3 || a->dimensions[0] != b->dimensions[0]) {
        throw_error("Arrays must have the same shape. Broadcasting not implemented.");
        return nullptr;
    }
    if (a->dimensions[2] != b->dimensions[1]) {
        throw_error("Shape mismatch for matmul. cols(a) != rows(b)");
        return nullptr;
    }
    if (gather_mode < NP_SHARD_OVERLAP) {
        throw_error("gather mode must be 0 (keep sharded), 1 (gather), -1 (overlapped, piece count chosen by the library) or a number of overlapped pieces");
        return nullptr;
    }
    const int world = NDArray_CommWorld(), slab = a->dimensions[0];
    if (batch < 0 || (long)slab * world != (long)batch) {
        throw_error("Batch of %d is not %d slab(s) of %d", batch, world, slab);
        return nullptr;
    }
    if (gather_mode == NP_SHARD_KEEP || np_comm_world() == 0) return NDArray_BatchedMatmul(a, b);   // no collective on the path
    if (!require_gpu(a, "matmul")) return nullptr;
    const size_t M = (size_t)a->dimensions[1], K = (size_t)a->dimensions[2], N = (size_t)b->dimensions[2];
    int shape[3] = {batch, (int)M, (int)N};
    NDArray *result = new_array(shape, 3, NDARRAY_DEVICE_GPU, false);   // every rank's window is written by its owner
    if (!result) return nullptr;
    if (!dev_ok(np_sgemm_strided_batched_allgather((size_t)slab, M, N, K, NDArray_FDATA(a), M * K, NDArray_FDATA(b), K * N,
                                                   NDArray_FDATA(result),
                                                   gather_mode == NP_SHARD_GATHER ? 1 : (gather_mode == NP_SHARD_OVERLAP ? 0 : gather_mode),
                                                   NP_GATHER_AUTO))) {
        NDArray_FREE(result);
        return nullptr;
    }
    return result;
}

}  // extern "C"
