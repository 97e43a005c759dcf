// The op bodies of the elementwise path — one definition of every binary / unary float_* operation, shared by the
// translation units that apply them (np_elementwise.hip: the stand-alone kernels and the chain interpreter;
// np_fused_static.hip: the compiled chains), so that a fused chain is bit-identical to the op-by-op sequence by
// construction.  Internal to libnp_hip.so; everything lives in an anonymous namespace (one copy per translation unit).
//
// Reference behaviour restated (file:line relative to the reference tree):
//   binary ops      src/ndmath/arithmetics.c:160-926  (+ CUDA kernels cuda_math.cu:593-633)
//   unary ops       src/ndmath/double_math.c:9-265    (+ CUDA kernels cuda_math.cu:207-585)
#ifndef NUMPOWER_AMD_NP_ELEMENTWISE_OPS_H
#define NUMPOWER_AMD_NP_ELEMENTWISE_OPS_H

#include <math.h>

#include "np_internal.h"
#include "np_pow_tables.h"

// Bit-level parity with the reference's CPU results needs every multiply, add and divide rounded
// on its own: no implicit FMA contraction anywhere in a file that includes this header (the two places where the
// reference's build DOES fuse — mod's AVX body and the rsqrt Newton step — say so with an
// explicit __fmaf_rn).  `/` and sqrtf are IEEE-correct under hipcc's defaults; the __f*_rn
// helpers are plain operators in this ROCm, kept only as documentation of intent.
#pragma clang fp contract(off)

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// scalar op bodies
// ------------------------------------------------------------------------------------------

// pow for the common case — x finite and normal, y finite — as 2^(y log2 x) in fp64 (an fp64 FMA issues at the
// fp32 rate on this part: every VALU instruction of a wave64 takes 4 cycles, so one fp64 op does the work of a
// handful of double-float fp32 ops).  glibc's powf, which the reference calls per element (arithmetics.c:912-914),
// is < 0.52 ulp; this is the correctly rounded fp32 power in all but ~2.5e-4 of cases and never more than 1 ulp off.
//
//   log2 x   x = 2^e m, m in [sqrt(1/2), sqrt(2)) (offset split of the bit pattern); the top 5 mantissa bits pick
//            one of 32 intervals with centre c_i: z = m / c_i - 1 (one FMA against the tabulated 1 / c_i, |z| <=
//            0.0153), log2 x = (e + log2 c_i) + z (b1 + z (b2 + ... + z b6)).  The table (np_pow_tables.h, 512
//            bytes, made by tools/gen_pow_tables.py) sits in global memory and is read with a per-lane index: it
//            lives in the L1 / L2 of every CU, and a vector load costs no VALU issue slot — which is what this kernel
//            is short of.  The interval around 1.0 has c = 1 exactly, so powers of numbers next to 1 lose nothing.
//   2^t      t = y log2 x = n + f, |f| <= 1/2: degree-8 polynomial; its tail c4 + ... + c8 f^4 enters scaled by
//            f^4 c4 <= 6e-4 and runs in fp32, two elements per instruction (v_pk_fma_f32); no clamp — rint, the
//            saturating v_cvt_i32_f64 and v_ldexp_f64 turn an out-of-range t into inf / 0 by themselves.
// A negative base is NaN unless y is an integer; zeros, denormals, inf, NaN take the library's powf with its C99
// special cases: both behind wave-uniform branches (a ballot per float4), so positive normal data runs the core
// and nothing else.
//
// How it got here (profiles/r02/pow_r02.log).  Round 1: 2^(y log2 x) with log2 from the atanh series of
// (m - 1) / (m + 1) (fp32 reciprocal + Newton step), all fp64, branch-free sign handling: 67 VALU instructions per
// element, SQ_ACTIVE_INST_VALU = 99 % of the CU-busy cycles, the fp64-saturated chip clocked down to ~1.5 GHz:
// 202-325 us per 1e8 elements.  A quarter of the instructions were v_mov: each polynomial constant was copied to
// a VGPR pair per element so that v_fmac_f64 (dst == addend) could consume it -> the FMAs below are VOP3 v_fma_f64
// with the constant read from an SGPR pair (one constant-bus operand).  With the uniform branches: 48 / element.
// The table instead of the division + series, no clamp: 38 / element, 15 of them fp64 arithmetic (were 29).

// a * b + c with c (resp. a) a uniform constant in an SGPR pair
__device__ __forceinline__ double fma_vvs(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
// (int)x with the instruction's own saturation (C++'s conversion is undefined out of range)
__device__ __forceinline__ int cvt_i32_sat(double x) {
    int n;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(x));
    return n;
}

// One out-of-line copy of the library routine: inlined at each of a thread's elements it made the
// kernel 4800 instructions long and the hot path a walk across the instruction cache.
__device__ __attribute__((noinline)) float pow_slow(float x, float y) { return powf(x, y); }

typedef float v2f __attribute__((ext_vector_type(2)));

// r = c[0] + x (c[1] + x (... c[NC - 1])) in fp32, element pairs packed
template <int N, int NC>
__device__ __forceinline__ void horner32(const float (&x)[N], float (&r)[N], const float (&c)[NC]) {
    if constexpr (N % 2 == 0) {
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            const v2f xv = {x[j], x[j + 1]};
            v2f acc = {c[NC - 1], c[NC - 1]};
#pragma unroll
            for (int i = NC - 2; i >= 0; --i) acc = __builtin_elementwise_fma(acc, xv, (v2f){c[i], c[i]});
            r[j] = acc[0];
            r[j + 1] = acc[1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float acc = c[NC - 1];
#pragma unroll
            for (int i = NC - 2; i >= 0; --i) acc = __builtin_fmaf(acc, x[j], c[i]);
            r[j] = acc;
        }
    }
}

// |x|^y for N elements, |x| normal and finite (xb = its bits), y finite; garbage (never a trap, never an
// out-of-bounds table index) otherwise.
// `tab`: the 32-entry table — in LDS in the streaming kernel (ds_read_b128 per element; read straight from global
// memory the per-lane gathers made the kernel wait on the vector cache: SQ_WAIT_ANY doubled and 213 us became 232,
// profiles/r02/pow_r02.log), kPowLogTab itself elsewhere.
typedef const __attribute__((address_space(3))) PowLogEntry *PowTabLds;
// In LDS: one 512-byte copy per wave.  Entries e and e + 16 share banks, and with a random entry per lane 44 % of the
// lookup's LDS cycles are bank conflicts (SQ_LDS_BANK_CONFLICT 6.1e6 of SQ_LDS_IDX_ACTIVE 1.39e7,
// profiles/r03/pmc_sq_pow_tables.txt) — not on the critical path: the kernel is bound by its fp64 arithmetic on a slow
// box and by HBM on a fast one.  Round 3 first tried a conflict-free LDS layout (16 replicas, entry-major, lane l reads
// replica l & 15: 0 conflicts, profiles/r03/pmc_sq_pow_replicated.txt): staging 8 KB per workgroup made the uncapped
// grid's 49 000 workgroups read 400 MB of table from L2 (218 us against 205), and a grid capped at 8-18 workgroups per CU
// to amortise it ran 220-233 us (profiles/r03/pow_ab.log).  What ships is the register form below: no banks, nothing
// staged, 3 % fewer VALU instructions, and in every back-to-back pair a little faster: 184.7-190.9 us against 184.9-194.3
// on a box where add takes 185-186 (profiles/r03/pow_regtab_ab.log; under the counters 3.18e6 against 3.35e6 GPU cycles for
// five launches, pmc_sq_pow_tables.txt) — pow now runs at add's rate there.  The LDS copy stays as the A/B partner
// (variant 9000).

// The same table held in two VGPRs across the wave and read with ds_bpermute_b32 (the LDS crossbar, no banks involved:
// no conflicts by construction, nothing staged).  `invc`: lanes 0..31 hold the low word of invc[lane], lanes 32..63 the
// high word of invc[lane - 32]; `logc`: the same for logc.  A lookup is four bpermutes (byte address 4 e, + 128 for the high
// words).  bpermute returns 0 for a source lane that is not executing: every lane of the wave must be active at a lookup
// (binary_vec_kernel<..., POWREG = true> keeps them so).
struct PowTabRegs {
    int invc, logc;
    __device__ __forceinline__ void load(unsigned lane) {
        const int *w = (const int *)kPowLogTab;
        invc = w[4 * (lane & 31u) + (lane >> 5)];
        logc = w[4 * (lane & 31u) + 2 + (lane >> 5)];
    }
    __device__ __forceinline__ PowLogEntry operator[](unsigned e) const {
        const int at = (int)(e * 4u);
        const int i_lo = __builtin_amdgcn_ds_bpermute(at, invc), i_hi = __builtin_amdgcn_ds_bpermute(at + 128, invc);
        const int l_lo = __builtin_amdgcn_ds_bpermute(at, logc), l_hi = __builtin_amdgcn_ds_bpermute(at + 128, logc);
        return PowLogEntry{__hiloint2double(i_hi, i_lo), __hiloint2double(l_hi, l_lo)};
    }
};

template <int N, typename Tab>
__device__ __forceinline__ void pow_core_n(const unsigned (&xb)[N], const float *y, float *out, Tab tab) {
    double f[N];
    float f32[N], q32[N];
    int n[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned t0 = xb[k] - 0x3f3504f3u;                     // bits of sqrt(1/2): moves the exponent split there
        const int e = (int)t0 >> 23;
        const float m = __uint_as_float((t0 & 0x007fffffu) + 0x3f3504f3u);
        const PowLogEntry tc = tab[(t0 >> 18) & 31u];
        const double z = fma((double)m, tc[0], -1.0);
        double q = fma_vvs(z, NP_POW_LOG2_B6, NP_POW_LOG2_B5);
        q = fma_vvs(q, z, NP_POW_LOG2_B4);
        q = fma_vvs(q, z, NP_POW_LOG2_B3);
        q = fma_vvs(q, z, NP_POW_LOG2_B2);
        q = fma_vvs(q, z, NP_POW_LOG2_B1);
        const double L = fma(q, z, tc[1] + (double)e);
        const double t = (double)y[k] * L;
        const double nd = rint(t);
        f[k] = t - nd;
        n[k] = cvt_i32_sat(nd);
        f32[k] = (float)f[k];
    }
    const float kExpTail[5] = {9.61812910762847688e-03f, 1.33335581464284411e-03f, 1.54035303933816061e-04f,
                               1.52527338040598377e-05f, 1.32154867901443053e-06f};
    horner32<N, 5>(f32, q32, kExpTail);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double q = fma_vvs((double)q32[k], f[k], 5.55041086648215762e-02);
        q = fma_vvs(q, f[k], 2.40226506959100694e-01);
        q = fma_vvs(q, f[k], 6.93147180559945286e-01);
        q = fma(q, f[k], 1.0);
        out[k] = (float)ldexp(q, n[k]);
    }
}

// the fp64 path applies: base finite, normal and non-zero; exponent finite
__device__ __forceinline__ bool pow_fast_ok(unsigned xb, float y) {
    return (xb - 0x00800000u) < 0x7f000000u && (__float_as_uint(y) & 0x7f800000u) != 0x7f800000u;
}
// negative base: NaN unless y is an integer, whose parity picks the sign (C99 7.12.7.4); OR-ed into the bits
__device__ __forceinline__ unsigned pow_neg_fix(float y) {
    const float ay = fabsf(y);
    const unsigned odd = (ay < 16777216.0f) ? ((unsigned)(int)ay << 31) : 0u;
    return (truncf(ay) != ay) ? 0x7fc00000u : odd;
}

// N powers per call (a float4, or the elements a fused-chain trip holds): the rare cases — a negative base
// anywhere in the wave, an operand outside the fp64 path — are taken by scalar (wave-uniform) branches, so a
// wave of positive normal data runs pow_core and nothing else.
template <int N, typename Tab>
__device__ __forceinline__ void pow_n(const float *x, const float *y, float *out, Tab tab) {
    // rare-case census in 4 integer ops per element: OR of the sign bits; unsigned max of (|x| bits - min
    // normal) — zero / denormal bases wrap to huge values, inf / NaN stay >= 0x7f000000 — and of the |y| bits
    unsigned sign_or = 0u, x_span = 0u, y_top = 0u;
    unsigned xb[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned xs = __float_as_uint(x[k]);
        xb[k] = xs & 0x7fffffffu;
        sign_or |= xs;
        x_span = max(x_span, xb[k] - 0x00800000u);
        y_top = max(y_top, __float_as_uint(y[k]) & 0x7fffffffu);
    }
    pow_core_n<N>(xb, y, out, tab);
    if (__builtin_amdgcn_ballot_w64((int)sign_or < 0) != 0ull) {
#pragma unroll
        for (int k = 0; k < N; ++k)
            if ((int)__float_as_uint(x[k]) < 0) out[k] = __uint_as_float(__float_as_uint(out[k]) | pow_neg_fix(y[k]));
    }
    if (__builtin_amdgcn_ballot_w64(x_span >= 0x7f000000u || y_top >= 0x7f800000u) != 0ull) {
#pragma unroll 1
        for (int k = 0; k < N; ++k)
            if (!pow_fast_ok(__float_as_uint(x[k]) & 0x7fffffffu, y[k])) out[k] = pow_slow(x[k], y[k]);
    }
}

__device__ __forceinline__ float fast_pow(float x, float y) {
    float r;
    pow_n<1>(&x, &y, &r, (const PowLogEntry *)kPowLogTab);
    return r;
}

// `body` = element lies in the range the reference's AVX2 loop covers (only meaningful when the
// caller asked for NP_QUIRK_AVX_BODY); QUIRK=false gives plain IEEE / C semantics.
template <int OP, bool QUIRK>
__device__ __forceinline__ float binary_apply(float a, float b, bool body) {
    if constexpr (OP == NP_ADD) return a + b;
    if constexpr (OP == NP_SUBTRACT) return a - b;
    if constexpr (OP == NP_DIVIDE) return __fdiv_rn(a, b);
    if constexpr (OP == NP_MULTIPLY) {
        float p = a * b;
        if constexpr (QUIRK) {
            // arithmetics.c:403 (body: every zero -> -0.0f) / :410-412 (tail: -0.0f -> +0.0f)
            if (p == 0.0f) p = body ? -0.0f : 0.0f;
        }
        return p;
    }
    if constexpr (OP == NP_MOD) {
        if constexpr (QUIRK) {
            if (body) {
                // arithmetics.c:794: a - floor(a/b)*b; gcc -march=native contracts the
                // sub(mul) into one vfnmadd231ps, i.e. a single rounding.
                float q = floorf(__fdiv_rn(a, b));
                return __fmaf_rn(-q, b, a);
            }
        }
        return fmodf(a, b);   // arithmetics.c:800, cuda_math.cu:628
    }
    if constexpr (OP == NP_POW) return fast_pow(a, b);
    if constexpr (OP == NP_ARCTAN2) return atan2f(a, b);
    // comparisons (src/logic.c:67-670): ordered, 1.0f / 0.0f
    if constexpr (OP == NP_GREATER) return (a > b) ? 1.0f : 0.0f;
    if constexpr (OP == NP_GREATER_EQUAL) return (a >= b) ? 1.0f : 0.0f;
    if constexpr (OP == NP_LESS) return (a < b) ? 1.0f : 0.0f;
    if constexpr (OP == NP_LESS_EQUAL) return (a <= b) ? 1.0f : 0.0f;
    // glibc's fmaxf / fminf (math/s_fmax_template.c): x if x >= y, y if x < y, else the non-NaN one
    if constexpr (OP == NP_MAXIMUM) return (a >= b || b != b) ? a : b;
    if constexpr (OP == NP_MINIMUM) return (a <= b || b != b) ? a : b;
    if constexpr (OP == NP_EQUAL) {
        // AVX2 body: _CMP_EQ_OQ (logic.c:541); tail and CUDA kernel: |a-b| <= 1e-7 (logic.c:552)
        if (QUIRK && body) return (a == b) ? 1.0f : 0.0f;
        return (fabsf(a - b) <= 0.0000001f) ? 1.0f : 0.0f;
    }
    if constexpr (OP == NP_NOT_EQUAL) {
        // AVX2 body: _CMP_NEQ_OQ, ordered: NaN -> 0 (logic.c:642); tail: !(|a-b| <= 1e-7) (logic.c:655)
        if (QUIRK && body) return (a < b || a > b) ? 1.0f : 0.0f;
        return (fabsf(a - b) <= 0.0000001f) ? 0.0f : 1.0f;
    }
    return 0.0f;
}

// exp(x) = 2^n 2^f with n = rint(x log2 e) and f = x log2 e - n from two FMAs (log2 e split hi + lo, so f is good
// to 2^-30 whatever n is), 2^f from the hardware v_exp_f32 (1 ulp), the scaling by v_ldexp_f32 (denormal results
// included): 8 VALU instructions where the device library's expf spends 14 — the same argument reduction, but it
// adds compare / select pairs for the overflow and underflow ends.  Here the argument is clamped into
// [-104, 88.8] (one v_med3: beyond those the result is 0 / inf anyway and n stays a small integer), and the LAST
// fma takes the UNclamped x: a NaN or an infinity comes out of it as NaN / +-inf, v_exp_f32 turns that into
// NaN / inf / 0, v_ldexp_f32 keeps it — no special-case instructions at all.  <= 1.2 ulp against fp64 over the
// whole range (tests/test_gpu_libm_domain.py holds it to the 1e-5 bar against the oracle's glibc; special values
// bit for bit).  The fused chains that end in a reduction are VALU-bound, and exp is their commonest step.
__device__ __forceinline__ float fast_exp(float x) {
    const float log2e_hi = 0x1.715476p+0f, log2e_lo = __uint_as_float(0x32a57060u);   // 1.9259630335e-08
    const float xc = __builtin_amdgcn_fmed3f(x, -104.0f, 88.8f);
    const float n = __builtin_rintf(xc * log2e_hi);
    float f = __builtin_fmaf(xc, log2e_hi, -n);
    f = __builtin_fmaf(x, log2e_lo, f);
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// log1p and the hyperbolic family: the device library's versions are 3-4x too slow to stay under
// the HBM roofline (tools/op_sweep.py: log1pf 2.5, sinhf 2.3, asinhf 1.9 TB/s against 6.4 for expf).
// These are a handful of VALU instructions around expf / logf / sqrtf, accurate to a few ulp over
// the whole fp32 range (tests/test_gpu_parity.py::test_fast_hyperbolic_accuracy holds them to 2e-6
// relative against fp64 and to the 1e-5 bar against the oracle's glibc), with glibc's results at
// +-0, +-inf, NaN and the domain edges.
// BOUNDED: the caller guarantees 0 <= x < 2^30, so (1+x)-1 is 0 or lies in [2^-24, 2^30] and
// v_rcp_f32 (1 ulp, no denormal/overflow corner) can replace the IEEE division.
template <bool BOUNDED = false>
__device__ __forceinline__ float fast_log1p(float x) {
    // log(1+x) * x / ((1+x) - 1): the quotient cancels the rounding error of 1+x (Kahan)
    const float u = 1.0f + x;
    const float d = u - 1.0f;
    const float q = BOUNDED ? x * __builtin_amdgcn_rcpf(d) : x / d;
    float r = (d == 0.0f) ? x : logf(u) * q;   // tiny x (and -0.0) return x itself
    if (u == INFINITY) r = INFINITY;
    return r;   // x = -1 -> -inf, x < -1 -> NaN (logf of a negative), NaN -> NaN
}

__device__ __forceinline__ float fast_sinh(float x) {
    const float a = fabsf(x);
    float r;
    if (a < 0.5f) {          // odd Taylor polynomial: e^a - e^-a would cancel
        const float a2 = a * a;
        r = a + a * a2 * (1.0f / 6 + a2 * (1.0f / 120 + a2 * (1.0f / 5040 + a2 * (1.0f / 362880))));
    } else if (a < 88.0f) {
        const float e = expf(a);
        r = 0.5f * e - 0.5f / e;
    } else {                 // e^a overflows before sinh does (89.416): e^(a/2) * e^(a/2) / 2
        const float h = expf(0.5f * a);
        r = (0.5f * h) * h;
    }
    return copysignf(r, x);
}

__device__ __forceinline__ float fast_cosh(float x) {
    const float a = fabsf(x);
    if (a < 88.0f) {
        const float e = expf(a);
        return 0.5f * e + 0.5f / e;
    }
    const float h = expf(0.5f * a);
    return (0.5f * h) * h;
}

__device__ __forceinline__ float fast_asinh(float x) {
    const float a = fabsf(x);
    float r;
    if (a > 268435456.0f) {  // 2^28: sqrt(a^2 + 1) == a in fp32 (and a^2 overflows later): log(2a)
        r = logf(a) + 0.69314718f;
    } else {                 // log1p(a + a^2 / (1 + sqrt(a^2 + 1))): no cancellation for small a
        const float a2 = a * a;
        // the denominator lies in [2, 2^28]: v_rcp_f32 (1 ulp) instead of a full IEEE division
        r = fast_log1p<true>(a + a2 * __builtin_amdgcn_rcpf(1.0f + sqrtf(a2 + 1.0f)));
    }
    return copysignf(r, x);
}

__device__ __forceinline__ float fast_acosh(float x) {
    if (x < 1.0f) return NAN;
    if (x > 268435456.0f) return logf(x) + 0.69314718f;
    const float t = x - 1.0f;   // exact
    return fast_log1p<true>(t + sqrtf(2.0f * t + t * t));
}

__device__ __forceinline__ float fast_atanh(float x) {
    const float a = fabsf(x);
    // |x| = 1 -> 2/0 = inf -> inf; |x| > 1 -> log1p of something < -1 -> NaN
    return copysignf(0.5f * fast_log1p<false>((2.0f * a) / (1.0f - a)), x);
}

// Not an op of the C ABI: what a chain runs for the step `value ** 2.0f` (np_fused_chain maps it), because np_binary computes
// `$a ** 2` as the correctly rounded x * x and a chain must return the bits of the op-by-op sequence.
constexpr int NP_UNARY_SQUARE = NP_UNARY_OP_COUNT;

template <int OP>
__device__ __forceinline__ float unary_apply(float x, float p0, float p1) {
    if constexpr (OP == NP_ABS) return fabsf(x);
    // sqrtf is correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt;
    // __fsqrt_rn is NOT (it lowers to the native approximation unless OCML_BASIC_ROUNDED_OPERATIONS)
    if constexpr (OP == NP_SQRT) return sqrtf(x);
    if constexpr (OP == NP_EXP) return fast_exp(x);
    if constexpr (OP == NP_EXP2) return exp2f(x);
    if constexpr (OP == NP_EXPM1) return expm1f(x);
    if constexpr (OP == NP_LOG) return logf(x);
    if constexpr (OP == NP_LOG2) return log2f(x);
    if constexpr (OP == NP_LOG10) return log10f(x);
    if constexpr (OP == NP_LOG1P) return fast_log1p<false>(x);
    if constexpr (OP == NP_LOGB) return logbf(x);
    if constexpr (OP == NP_SIN) return sinf(x);
    if constexpr (OP == NP_COS) return cosf(x);
    if constexpr (OP == NP_TAN) return tanf(x);
    if constexpr (OP == NP_ARCSIN) return asinf(x);
    if constexpr (OP == NP_ARCCOS) return acosf(x);
    if constexpr (OP == NP_ARCTAN) return atanf(x);
    // double_math.c:156-162: the constant is built in double from pi ~ 3.1415926535
    if constexpr (OP == NP_DEGREES) return (float)((double)x * (180.0 / 3.1415926535));
    if constexpr (OP == NP_RADIANS) return (float)((double)x * (3.1415926535 / 180.0));
    if constexpr (OP == NP_SINH) return fast_sinh(x);
    if constexpr (OP == NP_COSH) return fast_cosh(x);
    if constexpr (OP == NP_TANH) return tanhf(x);
    if constexpr (OP == NP_ARCSINH) return fast_asinh(x);
    if constexpr (OP == NP_ARCCOSH) return fast_acosh(x);
    if constexpr (OP == NP_ARCTANH) return fast_atanh(x);
    // double_math.c:200-210: the post-adjust can never fire (rounded - floor is 0 or 1), so
    // float_rint is rintf: round half to even.
    if constexpr (OP == NP_RINT) return rintf(x);
    if constexpr (OP == NP_FIX) return truncf(x);
    if constexpr (OP == NP_FLOOR) return floorf(x);
    if constexpr (OP == NP_CEIL) return ceilf(x);
    if constexpr (OP == NP_TRUNC) return truncf(x);
    if constexpr (OP == NP_SINC) {
        // double_math.c:228-235
        const float pi = 3.1415927f;
        if (x == 0.0f) x = 1.0e-20f;
        x = __fmul_rn(pi, x);
        return __fdiv_rn(sinf(x), x);
    }
    if constexpr (OP == NP_NEGATE) return -x;
    if constexpr (OP == NP_SIGN) return (float)((x > 0.0f) - (x < 0.0f));
    if constexpr (OP == NP_CLIP) return fminf(p1, fmaxf(x, p0));
    // double_math.c:254-257 with factor = powf(10, decimals) evaluated on the host (p0)
    if constexpr (OP == NP_ROUND) return __fdiv_rn(roundf(__fmul_rn(x, p0)), p0);
    if constexpr (OP == NP_RSQRT) {
        // double_math.c:111-126; gcc -march=native contracts 1.5 - (x2*y)*y into one fnmadd
        const float x2 = __fmul_rn(x, 0.5f);
        unsigned i = __float_as_uint(x);
        i = 0x5f3759dfu - (i >> 1);
        float y = __uint_as_float(i);
        const float t = __fmul_rn(x2, y);
        const float u = __fmaf_rn(-t, y, 1.5f);
        return __fmul_rn(y, u);
    }
    if constexpr (OP == NP_POSITIVE) return (x < 0.0f) ? -x : x;   // double_math.c:241-244
    if constexpr (OP == NP_RECIPROCAL) return __fdiv_rn(1.0f, x);
    if constexpr (OP == NP_UNARY_SQUARE) return x * x;
    return 0.0f;
}

// ------------------------------------------------------------------------------------------
// memory helpers
// ------------------------------------------------------------------------------------------

// float4 accesses that only assume 4-byte alignment: global_load/store_dwordx4 take dword-aligned
// addresses, so row views and slices that start anywhere ($a[i] of a matrix with an odd row length)
// use the same vector kernels as aligned arrays (they used to fall back to the scalar kernels at
// 65-80 % of the rate; an aligned address costs nothing extra).
typedef v4f v4f_u __attribute__((aligned(4)));

template <bool NT>
__device__ __forceinline__ v4f ld4(const float *p) {
    if constexpr (NT) return __builtin_nontemporal_load((const v4f_u *)p);
    return *(const v4f_u *)p;
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, v4f v) {
    if constexpr (NT)
        __builtin_nontemporal_store(v, (v4f_u *)p);
    else
        *(v4f_u *)p = v;
}

// Division of a 32-bit index by the (run-time, per-launch constant) row length as multiply-high +
// shifts (Granlund-Montgomery round-up method): a plain u32 division is ~40 VALU instructions, which
// is what made exp(X) + row VALU-bound.  q = (t + ((n - t) >> s1)) >> s2 with t = umulhi(m, n).
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned m, unsigned s1, unsigned s2) {
    const unsigned t = __umulhi(m, n);
    return (t + ((n - t) >> s1)) >> s2;
}

// host side: the constants of fast_div for a divisor d >= 1 (d == 1: q = n through m = 0 ... handled by the callers' d >= 2)
inline void fast_div_magic(unsigned long long d, unsigned &m, unsigned &s1, unsigned &s2) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    m = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
    s1 = l ? 1 : 0;
    s2 = l ? l - 1 : 0;
}

}  // namespace

#endif
