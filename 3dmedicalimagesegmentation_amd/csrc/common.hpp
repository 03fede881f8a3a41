// Shared device/host helpers for the gfx950 (CDNA4, wave64) UNETR kernels.
//
// Precision policy: every contraction is written once against a "16-byte chunk" abstraction and
// instantiated twice:
//   PrecF32  : chunk = 4 x f32, contraction on v_mfma_f32_16x16x4_f32  (bit-exact fp32 fma chains;
//              the parity mode that must hold 1e-3 against the CPU oracle)
//   PrecBF16 : chunk = 8 x bf16, contraction on v_mfma_f32_16x16x32_bf16 (fp32 accumulate; perf mode)
//   PrecBF16x3: fp32 storage, operands split into bf16 (hi, lo) pairs, two bf16 MFMAs per chunk pair (below)
// In both modes a "k-block" is 4 chunks (64 bytes) per row: lane group g = lane>>4 owns chunk g, and the
// A and B operands use the same lane->k map, so any k permutation inside a k-block cancels out.
#pragma once
#include "../../include/unetr_hip.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define UNETR_OK 0
#define UNETR_ERR_ARG 1
#define UNETR_ERR_LAUNCH 2
#define UNETR_ERR_UNSUPPORTED 3
#define UNETR_ERR_WORKSPACE 4

#define UNETR_PREC_F32 0
#define UNETR_PREC_BF16 1
#define UNETR_PREC_BF16X3 2

static inline int unetr_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? UNETR_OK : UNETR_ERR_LAUNCH;
}

struct PrecF32 {
    static constexpr int CH = 4;   // elements per 16-byte chunk
    static __device__ __forceinline__ u32x4 pack(const float* v) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        return __builtin_bit_cast(u32x4, t);
    }
    // acc(16x16) += A(16 x 16k) * B(16k x 16): lane group g holds k = 4g+t in element t
    // 16 bytes as they sit in a tensor of this mode's storage type -> operand chunk
    static __device__ __forceinline__ u32x4 from_raw(u32x4 r) { return r; }
    static __device__ __forceinline__ void mma(f32x4& acc, u32x4 a, u32x4 b) {
        f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    }
};

struct PrecBF16 {
    static constexpr int CH = 8;
    static __device__ __forceinline__ u32x4 pack(const float* v) {
        f32x8 t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        bf16x8 b = __builtin_convertvector(t, bf16x8);
        return __builtin_bit_cast(u32x4, b);
    }
    static __device__ __forceinline__ u32x4 from_raw(u32x4 r) { return r; }
    static __device__ __forceinline__ void mma(f32x4& acc, u32x4 a, u32x4 b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                      acc, 0, 0, 0);
    }
};

// PrecBF16x3 : the tolerance-grade mode that is not bound by the fp32 matrix rate.  fp32 STORAGE everywhere (as PrecF32); an
//   operand element a travels as ONE 32-bit word [hi | lo << 16] with hi = bf16(a), lo = bf16(a - hi) (a = hi + lo to 2^-17
//   relative), so every loader, LDS layout and transposing read of the fp32 instantiation moves it unchanged (CH = 4 words per
//   16-byte chunk).  The contraction runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation: the 8 bf16 slots a lane group
//   feeds are its 4 words' (hi, lo) pairs, so MFMA([a_hi, a_hi], [b_hi, b_lo]) + MFMA([a_lo, a_lo], [b_hi, b_lo]) adds
//   a_hi b_hi + a_hi b_lo + a_lo b_hi + a_lo b_lo per element -- the three products of the classic bf16x3 split plus the lo-lo
//   term, which the slot layout gives for free.  Two 16-cycle MFMAs per 16 contraction elements against four 32-cycle fp32 MFMAs:
//   a quarter of the bf16 rate, four times the fp32 rate; products carry ~16 mantissa bits.
struct PrecBF16x3 {
    static constexpr int CH = 4;
    static __device__ __forceinline__ uint32_t split(float a) {
        const __bf16 h = (__bf16)a;
#ifdef UNETR_X3_DROP_LO
        // diagnostic build only (make x3droplo; tools/decompose_bf16_error.py): the lo halves are zero, i.e. bf16 x 1 operands with
        // fp32 storage and fp32 accumulation -- what separates "bf16 operands" from "bf16 storage" in the error of the bf16 mode
        return (uint32_t)__builtin_bit_cast(uint16_t, h);
#endif
        const __bf16 l = (__bf16)(a - (float)h);
        return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
    }
    static __device__ __forceinline__ u32x4 pack(const float* v) { return (u32x4){split(v[0]), split(v[1]), split(v[2]), split(v[3])}; }
    static __device__ __forceinline__ u32x4 from_raw(u32x4 r) {
        const f32x4 f = __builtin_bit_cast(f32x4, r);
        return (u32x4){split(f[0]), split(f[1]), split(f[2]), split(f[3])};
    }
    static __device__ __forceinline__ void mma(f32x4& acc, u32x4 a, u32x4 b) {
        u32x4 ah, al;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ah[e] = __builtin_amdgcn_perm(a[e], a[e], 0x01000100u);      // [hi, hi]
            al[e] = __builtin_amdgcn_perm(a[e], a[e], 0x03020302u);      // [lo, lo]
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
// bf16x3 operand forms straight from raw fp32 bits (the X3 instantiation below is bound by this VALU work, so it is kept minimal):
// hi = the value TRUNCATED to bf16 (its upper 16 bits: no rounding instructions), lo = bf16(a - hi) -- the difference is exact in
// fp32, so hi + lo carries the value to 2^-16 relative, the precision the mode's products have anyway (PrecBF16x3, common.hpp).
typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float x3_rem(uint32_t r) {
#ifdef UNETR_X3_DROP_LO
    return 0.f;
#else
    return __builtin_bit_cast(float, r) - __builtin_bit_cast(float, r & 0xffff0000u);
#endif
}
// the operand that enters both MFMAs as it is: four words [hi | lo << 16]
__device__ __forceinline__ u32x4 x3_words(u32x4 r) {
#ifdef UNETR_X3_DROP_LO
    return PrecBF16x3::from_raw(r);            // (diagnostic build: round-to-nearest hi, zero lo -- bf16 x 1 operands)
#endif
    const bf16x2_ l01 = __builtin_convertvector((f32x2_){x3_rem(r[0]), x3_rem(r[1])}, bf16x2_);
    const bf16x2_ l23 = __builtin_convertvector((f32x2_){x3_rem(r[2]), x3_rem(r[3])}, bf16x2_);
    const uint32_t p01 = __builtin_bit_cast(uint32_t, l01), p23 = __builtin_bit_cast(uint32_t, l23);
    return (u32x4){__builtin_amdgcn_perm(p01, r[0], 0x05040302u), __builtin_amdgcn_perm(p01, r[1], 0x07060302u),
                   __builtin_amdgcn_perm(p23, r[2], 0x05040302u), __builtin_amdgcn_perm(p23, r[3], 0x07060302u)};
}
// the operand that is duplicated: [hi, hi] words and [lo, lo] words (one v_perm / one packed convert of (d, d) per element)
__device__ __forceinline__ void x3_dup(u32x4 r, u32x4& hh, u32x4& ll) {
#ifdef UNETR_X3_DROP_LO
    {
        const u32x4 w = PrecBF16x3::from_raw(r);
#pragma unroll
        for (int e = 0; e < 4; ++e) { hh[e] = __builtin_amdgcn_perm(w[e], w[e], 0x01000100u); ll[e] = 0u; }
        return;
    }
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t re = r[e];
        hh[e] = __builtin_amdgcn_perm(re, re, 0x03020302u);
        const float d = x3_rem(re);
        ll[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_){d, d}, bf16x2_));
    }
}

// one contraction element per lane (the K = 4 shape of v_mfma_f32_16x16x4_f32: lane group g supplies element g): kernels that
// feed the fp32 MFMA element by element collect FOUR steps into one chunk for the policies whose MFMA wants a whole chunk
template <class P> struct ElemMma {
    static constexpr bool PER_ELEMENT = false;
};
template <> struct ElemMma<PrecF32> {
    static constexpr bool PER_ELEMENT = true;
};

// XOR-swizzled LDS tile of 128-byte rows (8 chunks): conflict-free for the 16x16 fragment read
// (lane l reads row l&15, chunk (l>>4)+4*kb) under the ds_read_b128 lane grouping.
__device__ __forceinline__ int lds_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- AdamW element update (torch.optim.AdamW: decoupled weight decay, bias-corrected moments), shared by the streaming optimizer
// kernel (norm_misc.hip) and the fused epilogue of the grouped ViT weight gradient (gemm_bf16.hip) so that both give the same bits
struct AdamWCoef { float decay, b1, b2, eps, step_size, inv_sqrt_bc2; };
__device__ __forceinline__ AdamWCoef adamw_coef(float lr, float b1, float b2, float eps, float wd, float step) {
#pragma clang fp contract(off)
    const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
    return AdamWCoef{1.f - lr * wd, b1, b2, eps, lr / bc1, rsqrtf(bc2)};
}
__device__ __forceinline__ void adamw_elem(float& p, float& m, float& v, float g, const AdamWCoef& c) {
    // every multiply-add is spelled out (contraction off, explicit fmaf): left to the compiler, the same expression was fused
    // differently in the streaming kernel and in the GEMM epilogue and ~0.6 % of the elements differed in the last bit
#pragma clang fp contract(off)
    const float pe = p * c.decay;
    const float me = fmaf(c.b1, m, (1.f - c.b1) * g);
    const float ve = fmaf(c.b2, v, ((1.f - c.b2) * g) * g);
    const float denom = fmaf(sqrtf(ve), c.inv_sqrt_bc2, c.eps);
    p = fmaf(-c.step_size, me / denom, pe);
    m = me;
    v = ve;
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// The same two functions for the bf16-storage epilogues (EpBf: GELU output rounded to bf16, relative spacing 4e-3): libm's erff
// is ~45 instructions with a branch -- 96-128 calls per thread in a 256-wide GEMM tile cost 6-12 us of a 30 us launch -- so erf
// comes from Abramowitz & Stegun 7.1.26 (one v_rcp, one v_exp, five fmas): |error| <= 5.3e-7 on erf in fp32 arithmetic (checked
// over [-8, 8]), i.e. <= 2e-7 absolute on GELU and GELU' -- four orders of magnitude below the bf16 rounding of what is stored.
// exp(-x^2 / 2) is shared between the erf tail and the Gaussian density of GELU'.  The fp32 precision mode keeps erff (EpStd).
__device__ __forceinline__ float erf_poly_tail(float t /* |x| / sqrt(2) */, float e /* exp(-t^2) */) {
    const float k = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * t);
    float p = 1.061405429f;
    p = fmaf(p, k, -1.453152027f); p = fmaf(p, k, 1.421413741f); p = fmaf(p, k, -0.284496736f); p = fmaf(p, k, 0.254829592f);
    return 1.0f - p * k * e;                      // erf(t), t >= 0
}
__device__ __forceinline__ float gelu_fast(float x) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    const float e = __expf(-t * t);
    const float er = copysignf(erf_poly_tail(t, e), x);
    return 0.5f * x * (1.0f + er);
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float t = fabsf(x) * 0.70710678118654752440f;
    const float e = __expf(-t * t);               // = exp(-x^2 / 2)
    const float er = copysignf(erf_poly_tail(t, e), x);
    return 0.5f * (1.0f + er) + x * 0.39894228040143267794f * e;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- activation storage type of the conv side: fp32 (fp32 precision mode) or bf16 (bf16 precision mode: every feature map
// and feature-map gradient between the kernels is stored as bf16 -- half the HBM bytes of passes that are bandwidth-bound;
// statistics, weights, accumulators and reductions stay fp32).  Io<T> = the 4-element / 1-element access of such a tensor.
template <class T> struct Io;
template <> struct Io<float> {
    static constexpr int B16 = 0;
    static constexpr int W = 4;       // elements per 16-byte access
    static __device__ __forceinline__ void ldw(const float* p, float (&v)[4]) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void stw(float* p, const float (&v)[4]) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
    // the 16 bytes as loaded (kept packed while several loads are in flight) and their W elements
    static __device__ __forceinline__ u32x4 ldraw(const float* p) { return *(const u32x4*)p; }
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const uint32_t t = r[e]; v[e] = __builtin_bit_cast(float, t); }     // (bit_cast of the element lvalue itself reads element 0)
    }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Io<uint16_t> {
    static constexpr int B16 = 1;
    static constexpr int W = 8;
    static __device__ __forceinline__ void ldw(const uint16_t* p, float (&v)[8]) {
        const f32x8 t = __builtin_convertvector(*(const bf16x8*)p, f32x8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = t[e];
    }
    static __device__ __forceinline__ void stw(uint16_t* p, const float (&v)[8]) {
        const f32x8 t = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        *(bf16x8*)p = __builtin_convertvector(t, bf16x8);
    }
    static __device__ __forceinline__ u32x4 ldraw(const uint16_t* p) { return *(const u32x4*)p; }
    static __device__ __forceinline__ void unpack(u32x4 r, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, r[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, r[e] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ f32x4 ld4(const uint16_t* p) { return __builtin_convertvector(*(const bf16x4*)p, f32x4); }
    static __device__ __forceinline__ void st4(uint16_t* p, f32x4 v) { *(bf16x4*)p = __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ float ld1(const uint16_t* p) { return __builtin_bit_cast(float, (uint32_t)*p << 16); }
    static __device__ __forceinline__ void st1(uint16_t* p, float v) { __bf16 h = (__bf16)v; *p = __builtin_bit_cast(uint16_t, h); }
};
// bf16 precision mode <=> bf16 activation storage: the element type of the feature maps a kernel instantiated for P reads and writes
template <class P> struct ActOf { typedef float type; };
template <> struct ActOf<PrecBF16> { typedef uint16_t type; };
// P::CH consecutive elements at p (16 bytes either way) as one packed MFMA operand chunk; zero when !ok (p must be a valid
// address even then: callers clamp it)
template <class P>
__device__ __forceinline__ u32x4 act_chunk(const typename ActOf<P>::type* p, bool ok) {
    const u32x4 w = *(const u32x4*)p;
    return ok ? P::from_raw(w) : (u32x4){0u, 0u, 0u, 0u};
}
// LayerNorm backward whose dy arrives as split-K partial slabs (norm_misc.hip; used by unetr_gemm_bf16_ln_bwd in gemm_bf16.hip)
int unetr_layernorm_bwd_partials(const float* dy, int splits, long slab, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, void* dx_bf16, const float* dres, float* dgamma,
                                 float* dbeta, int M, int H, float* ws, size_t ws_bytes, void* stream);

// LayerNorm forward whose rows arrive as split-K partial slabs + (bias, residual) epilogue (norm_misc.hip; unetr_gemm_bf16_ln_fwd)
int unetr_layernorm_fwd_partials(const float* partials, int splits, long slab, const float* bias, const float* res, long ldr, int res_mod,
                                 float* xout, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean, float* rstd,
                                 int M, int H, float eps, void* stream);

// bf16x3 Linear GEMM on fp32-stored operands through the LDS-DMA kernel (gemm_bf16.hip); UNSUPPORTED = take the generic family
int unetr_gemm_x3_dma(const unetr_gemm_bf16_desc* d, const float* A, const float* B, int b_words, float* C, float* ws, size_t ws_bytes, void* stream,
                      int* psp = nullptr);

int unetr_instnorm_stats_finalize2(const float* part, const float* part_b, int nchunk, int B, long V, int C, float eps,
                                   float* stats, float* stats_b, void* stream);

// run CALL with `AT` bound to the activation storage type selected by the run-time flag act16
#define ACT_DISPATCH(act16, ...) do { if (act16) { typedef uint16_t AT; __VA_ARGS__; } else { typedef float AT; __VA_ARGS__; } } while (0)
