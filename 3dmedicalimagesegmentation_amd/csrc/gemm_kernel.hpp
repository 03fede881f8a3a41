#pragma once
// Generic MFMA GEMM family for gfx950: C[M,N] = epilogue(A[M,K] * B[K,N]), batched, split-K.
//
// One kernel template, parameterised by
//   P        precision policy (PrecF32 -> v_mfma_f32_16x16x4_f32, PrecBF16 -> v_mfma_f32_16x16x32_bf16)
//   AL / BL  operand loaders ("give me CH consecutive-k fp32 values of row r"), which is how the same
//            kernel serves torch Linear fwd / dgrad / wgrad, the 1x1x1 conv, and the 2x2x2 transposed
//            conv (a GEMM [voxels, Cin] x [Cin, 8*Cout] with a pixel-shuffle store)
//   EP       epilogue functor (bias, exact GELU, GELU', residual, accumulate, scatter stores)
//   WM,WN    16x16 MFMA tiles per wave; WVM,WVN waves per block.
//
// Data path: fp32 in HBM -> registers (converted to the MFMA operand type) -> XOR-swizzled LDS tile of
// 128-byte rows (2 k-blocks) -> ds_read_b128 fragments -> MFMA.  Global loads of stage t+1 are issued
// before the MFMAs of stage t (register prefetch), the LDS tile is single-buffered.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

// ------------------------------------------------------------------------------------------ loaders
// all-ones when ok, zero otherwise, as a value the optimizer treats as unknown (so `bits & mask` stays a v_and_b32 right
// where it is written instead of becoming a select that hipcc folds into a conditional load)
__device__ __forceinline__ uint32_t opaque_mask(bool ok) {
    uint32_t m = ok ? 0xFFFFFFFFu : 0u;
    asm("" : "+v"(m));
    return m;
}

template <bool VEC>
struct LdRowT {  // element(row,k) = p[b*stride + row*ld + k]        (k contiguous)
    // Branch-free on purpose: out-of-range chunks load from a clamped in-range address and are zeroed by a
    // select, so every load is issued unconditionally and hipcc can keep counted vmcnt waits for the prefetch
    // ring (the first version's nested branches compiled to 537 branches / 89 vmcnt(0) per kernel).
    // VEC requires 16-byte aligned base, ld % 4 == 0 and K % CH == 0 (checked on the host).
    static constexpr bool KCONTIG = true;
    const float* p; long ld, stride; int rows, vec;
    template <int CH>
    __device__ __forceinline__ void load(int b, int row, int k, int kend, float* v) const {
        const bool ok = row < rows && k < kend;
        const float* q = p + (long)b * stride + (long)(ok ? row : 0) * ld + (ok ? k : 0);
        // out-of-range data is masked by an AND on the loaded BITS with an all-ones / zero word the compiler cannot see
        // through (opaque_mask): a select (`ok ? loaded : 0`) is turned back into a conditional load by hipcc (one branch +
        // s_waitcnt per chunk), and a multiply by 0 lets one Inf / NaN at the clamped address poison every padded lane
        const uint32_t mk = opaque_mask(ok);
        if constexpr (VEC) {
#pragma unroll
            for (int c = 0; c < CH / 4; ++c) {
                u32x4 t = *(const u32x4*)(q + 4 * c);
                v[4 * c] = __uint_as_float(t[0] & mk); v[4 * c + 1] = __uint_as_float(t[1] & mk);
                v[4 * c + 2] = __uint_as_float(t[2] & mk); v[4 * c + 3] = __uint_as_float(t[3] & mk);
            }
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const bool okj = ok && (k + j < kend);
                uint32_t t = __float_as_uint(q[okj ? j : 0]);
                v[j] = __uint_as_float(t & opaque_mask(okj));
            }
        }
    }
};
typedef LdRowT<true> LdRow;
typedef LdRowT<false> LdRowS;

struct LdCol {  // element(row,k) = p[b*stride + k*ld + row]        (row contiguous)
    static constexpr bool KCONTIG = false;
    const float* p; long ld, stride; int rows, vec;
    template <int CH>
    __device__ __forceinline__ void load(int b, int row, int k, int kend, float* v) const {
        const bool okr = row < rows;
        const float* q = p + (long)b * stride + (okr ? row : 0);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const bool ok = okr && (k + j < kend);
            uint32_t t = __float_as_uint(q[(long)(ok ? k + j : 0) * ld]);
            v[j] = __uint_as_float(t & opaque_mask(ok));       // bit mask, not a multiply (see LdRowT)
        }
    }
};

// geometry of a 2x2x2 stride-2 transposed conv; m indexes the INPUT grid [B,D,H,W]
struct TcGeom {
    int D, H, W, Cout;
    __device__ __forceinline__ long outvox(int m, int tap) const {
        int x = m % W; int t = m / W; int y = t % H; t /= H; int z = t % D; int b = t / D;
        int a = tap >> 2, bb = (tap >> 1) & 1, c = tap & 1;
        return (((long)b * 2 * D + 2 * z + a) * 2 * H + 2 * y + bb) * 2 * W + 2 * x + c;
    }
};

struct LdTcGatherA {  // dgrad A: element(m, k = tap*Cout+co) = dy[outvox(m,tap)*ld + co]
    static constexpr bool KCONTIG = true;
    const float* p; long ld; int rows; TcGeom g;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
        if (row < rows && k < kend) {
            int tap = k / g.Cout, co = k - tap * g.Cout;
            if (co + CH <= g.Cout && ((ld | co) & 3) == 0) {
                const float* q = p + g.outvox(row, tap) * ld + co;
#pragma unroll
                for (int c = 0; c < CH / 4; ++c) {
                    f32x4 t = *(const f32x4*)(q + 4 * c);
                    v[4 * c] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    int kk = k + j; int tp = kk / g.Cout, cc = kk - tp * g.Cout;
                    v[j] = kk < kend ? p[g.outvox(row, tp) * ld + cc] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = 0.f;
        }
    }
};

struct LdTcGatherB {  // wgrad B: element(n = tap*Cout+co, k = m) = dy[outvox(m,tap)*ld + co]
    static constexpr bool KCONTIG = false;
    const float* p; long ld; int rows; TcGeom g;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
        int tap = row / g.Cout, co = row - tap * g.Cout;
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = (row < rows && k + j < kend) ? p[g.outvox(k + j, tap) * ld + co] : 0.f;
    }
};

struct LdTcWf {  // fwd B: element(n = tap*Cout+co, k = ci) = w[(ci*Cout+co)*8 + tap]
    static constexpr bool KCONTIG = false;
    const float* p; int rows; int Cout;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
        int tap = row / Cout, co = row - tap * Cout;
#pragma unroll
        for (int j = 0; j < CH; ++j) v[j] = (row < rows && k + j < kend) ? p[((long)(k + j) * Cout + co) * 8 + tap] : 0.f;
    }
};

struct LdTcWd {  // dgrad B: element(n = ci, k = tap*Cout+co) = w[(ci*Cout+co)*8 + tap]
    static constexpr bool KCONTIG = true;
    const float* p; int rows; int Cout;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            int kk = k + j; int tap = kk / Cout, co = kk - tap * Cout;
            v[j] = (row < rows && kk < kend) ? p[((long)row * Cout + co) * 8 + tap] : 0.f;
        }
    }
};


// geometry of a KSxKSxKS (KS = 1 or 3), stride-1, "same"-padded conv on a channels-last grid
struct ConvGeom {
    int D, H, W, Cin, KS;
    __device__ __forceinline__ long srcvox(int m, int tap) const {  // -1 when the tap falls in the padding
        if (KS == 1) return m;
        int x = m % W; int t = m / W; int y = t % H; t /= H; int z = t % D; int b = t / D;
        int dz = tap / 9, r = tap - dz * 9, dy = r / 3, dx = r - dy * 3;
        z += dz - 1; y += dy - 1; x += dx - 1;
        if ((unsigned)z >= (unsigned)D || (unsigned)y >= (unsigned)H || (unsigned)x >= (unsigned)W) return -1;
        return (((long)b * D + z) * H + y) * W + x;
    }
};

struct LdIm2colA {  // conv fwd/dgrad A: element(m = voxel, k = tap*Cin+ci) = x[srcvox(m,tap)*ld + ci]
    static constexpr bool KCONTIG = true;
    const float* p; long ld; int rows; ConvGeom g;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
        if (row < rows && k < kend) {
            int tap = k / g.Cin, ci = k - tap * g.Cin;
            if (ci + CH <= g.Cin && ((ld | ci) & 3) == 0 && k + CH <= kend) {
                long sv = g.srcvox(row, tap);
                if (sv >= 0) {
                    const float* q = p + sv * ld + ci;
#pragma unroll
                    for (int c = 0; c < CH / 4; ++c) {
                        f32x4 t = *(const f32x4*)(q + 4 * c);
                        v[4 * c] = t[0]; v[4 * c + 1] = t[1]; v[4 * c + 2] = t[2]; v[4 * c + 3] = t[3];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) v[j] = 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    int kk = k + j; int tp = kk / g.Cin, cc = kk - tp * g.Cin;
                    long sv = kk < kend ? g.srcvox(row, tp) : -1;
                    v[j] = sv >= 0 ? p[sv * ld + cc] : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = 0.f;
        }
    }
};

struct LdIm2colB {  // conv wgrad B: element(n = tap*Cin+ci, k = voxel) = x[srcvox(k,tap)*ld + ci]
    static constexpr bool KCONTIG = false;
    const float* p; long ld; int rows; ConvGeom g;
    template <int CH>
    __device__ __forceinline__ void load(int, int row, int k, int kend, float* v) const {
        int tap = row / g.Cin, ci = row - tap * g.Cin;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            long sv = (row < rows && k + j < kend) ? g.srcvox(k + j, tap) : -1;
            v[j] = sv >= 0 ? p[sv * ld + ci] : 0.f;
        }
    }
};

// ---------------------------------------------------------------------------------------- epilogues
struct EpStd {
    float* C; long ldc, strideC;
    const float* bias;
    const float* res; long ldr, strideR; int res_mod;
    float* pre;
    const float* aux; long ldaux;
    int act, accumulate; float alpha;
    __device__ __forceinline__ void store(int b, int m, int n, float v) const {
        v *= alpha;
        if (bias) v += bias[n];
        long o = (long)b * strideC + (long)m * ldc + n;
        if (pre) pre[o] = v;
        if (act == 1) v = gelu_exact(v);
        else if (act == 2) v *= gelu_grad(aux[(long)b * strideC + (long)m * ldaux + n]);
        if (res) v += res[(long)b * strideR + (long)(m % res_mod) * ldr + n];
        if (accumulate) v += C[o];
        C[o] = v;
    }
};

struct EpTcScatter {  // tconv fwd: (m = input voxel, n = tap*Cout+co) -> y[outvox(m,tap)*ld + co]
    float* y; long ld; TcGeom g;
    __device__ __forceinline__ void store(int, int m, int n, float v) const {
        int tap = n / g.Cout, co = n - tap * g.Cout;
        y[g.outvox(m, tap) * ld + co] = v;
    }
};

struct EpTcWgrad {  // (m = ci, n = tap*Cout+co) -> dw[(ci*Cout+co)*8 + tap]
    float* dw; int Cout;
    __device__ __forceinline__ void store(int, int m, int n, float v) const {
        int tap = n / Cout, co = n - tap * Cout;
        dw[((long)m * Cout + co) * 8 + tap] = v;
    }
};


struct EpConvWgrad {  // (m = co, n = tap*Cin+ci) -> dw[(co*Cin+ci)*KV + tap]   (torch Conv3d layout)
    float* dw; int Cin, KV;
    __device__ __forceinline__ void store(int, int m, int n, float v) const {
        int tap = n / Cin, ci = n - tap * Cin;
        dw[((long)m * Cin + ci) * KV + tap] = v;
    }
};

// ------------------------------------------------------------------------------------------- kernel
// one BM x BN output tile of C = A*B over k in [kbeg, kend): the body shared by the plain and grouped kernels
template <class P, class AL, class BL, class EP, int WM, int WN, int WVM, int WVN>
__device__ __forceinline__ void gemm_tile(int M, int N, int m0, int n0, int kbeg, int kend, int batch, int splits, int bz,
                                          const AL& al, const BL& bl, const EP& ep, float* __restrict__ ws, char* lds) {
    constexpr int NT = 64 * WVM * WVN, BM = 16 * WM * WVM, BN = 16 * WN * WVN, CH = P::CH, SK = 8 * CH;
    constexpr int AIT = (BM * 8 + NT - 1) / NT, BIT = (BN * 8 + NT - 1) / NT;
    constexpr bool AFULL = (BM * 8) % NT == 0, BFULL = (BN * 8) % NT == 0;   // no per-chunk guard (= no branch) when the tile divides evenly
    char* ldsA = lds;
    char* ldsB = lds + BM * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WVN, wn = wave % WVN;
    const int nk = (kend - kbeg + SK - 1) / SK;

    // Prefetch ring of PF stages held as RAW fp32 registers: the loads of stage kt+PF are issued right after
    // stage kt has been written to LDS, and the fp32 -> MFMA-operand conversion happens at LDS-store time, so
    // no instruction between a load and the MFMAs of the PF-1 stages in front of it depends on it.
    constexpr int RAWF = (AIT + BIT) * CH;                       // raw floats per thread per stage
    constexpr int PF = RAWF <= 32 ? 3 : (RAWF <= 48 ? 2 : 1);
    float raw[PF][AIT + BIT][CH];
    auto gload = [&](int kt, int slot) {
        const int k0 = kbeg + kt * SK;
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            int id = tid + i * NT;
            if (AFULL || id < BM * 8) {
                int row = AL::KCONTIG ? (id >> 3) : (id % BM), c = AL::KCONTIG ? (id & 7) : (id / BM);
                al.template load<CH>(batch, m0 + row, k0 + c * CH, kend, raw[slot][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            int id = tid + i * NT;
            if (BFULL || id < BN * 8) {
                int row = BL::KCONTIG ? (id >> 3) : (id % BN), c = BL::KCONTIG ? (id & 7) : (id / BN);
                bl.template load<CH>(batch, n0 + row, k0 + c * CH, kend, raw[slot][AIT + i]);
            }
        }
    };
    auto lstore = [&](int slot) {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            int id = tid + i * NT;
            if (AFULL || id < BM * 8) {
                int row = AL::KCONTIG ? (id >> 3) : (id % BM), c = AL::KCONTIG ? (id & 7) : (id / BM);
                *(u32x4*)(ldsA + lds_tile_off(row, c)) = P::pack(raw[slot][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            int id = tid + i * NT;
            if (BFULL || id < BN * 8) {
                int row = BL::KCONTIG ? (id >> 3) : (id % BN), c = BL::KCONTIG ? (id & 7) : (id / BN);
                *(u32x4*)(ldsB + lds_tile_off(row, c)) = P::pack(raw[slot][AIT + i]);
            }
        }
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Branch-free pipeline body: stages past the end of the K range are loaded as zeros by the loaders (k >= kend)
    // and multiplied harmlessly, so there is no control flow between the loads of stage kt+PF and the MFMAs of
    // stage kt -- with per-stage `if (kt < nk)` guards every pair of loads sat in its own basic block and was
    // followed by s_waitcnt vmcnt(0).  The host picks split-K so that the stage count is a multiple of PF.
#pragma unroll
    for (int s = 0; s < PF; ++s) gload(s, s);
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            lstore(s);
            __syncthreads();
            gload(kt0 + s + PF, s);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                u32x4 a[WM], b[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    a[i] = *(const u32x4*)(ldsA + lds_tile_off((wm * WM + i) * 16 + (lane & 15), kb * 4 + (lane >> 4)));
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    b[j] = *(const u32x4*)(ldsB + lds_tile_off((wn * WN + j) * 16 + (lane & 15), kb * 4 + (lane >> 4)));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) P::mma(acc[i][j], a[i], b[j]);
            }
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int m = m0 + (wm * WM + i) * 16 + 4 * (lane >> 4) + r;
                int n = n0 + (wn * WN + j) * 16 + (lane & 15);
                if (m < M && n < N) {
                    if (splits > 1) ws[((long)bz * M + m) * N + n] = acc[i][j][r];
                    else ep.store(batch, m, n, acc[i][j][r]);
                }
            }
}

template <class P, class AL, class BL, class EP, int WM, int WN, int WVM, int WVN>
__global__ void __launch_bounds__(64 * WVM * WVN)
gemm_kernel(int M, int N, int K, int splits, int kper, AL al, BL bl, EP ep, float* __restrict__ ws) {
    constexpr int BM = 16 * WM * WVM, BN = 16 * WN * WVN;
    __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * 128];
    const int bz = blockIdx.z, batch = bz / splits, split = bz - batch * splits;
    const int kbeg = split * kper, kend = min(K, kbeg + kper);
    gemm_tile<P, AL, BL, EP, WM, WN, WVM, WVN>(M, N, blockIdx.x * BM, blockIdx.y * BN, kbeg, kend, batch, splits, bz, al, bl, ep, ws, lds);
}

// ---- grouped weight-gradient GEMM: many independent dW[N,K] = dY[M,N]^T * X[M,K] problems in ONE launch ----------
// At batch 2 the reduction length is only M = 432 tokens, so a single problem cannot fill 256 CUs without
// split-K slabs; the 48 weight gradients of the 12 transformer blocks can.  Problems travel in the kernel
// argument block (no device-side descriptor memory, hipGraph-capturable).
constexpr int GROUP_MAX = 48;
struct GroupedProblem { const float* dy; const float* x; float* dw; int M, N, K, tile0, mtiles; };
struct GroupedArgs { int n; GroupedProblem p[GROUP_MAX]; };

template <class P, int WM, int WN, int WVM, int WVN>
__global__ void __launch_bounds__(64 * WVM * WVN)
gemm_grouped_wgrad_kernel(GroupedArgs ga) {
    constexpr int BM = 16 * WM * WVM, BN = 16 * WN * WVN;
    __shared__ __attribute__((aligned(16))) char lds[(BM + BN) * 128];
    int pi = 0;
    const int t = blockIdx.x;
    for (int hi_ = ga.n - 1; pi < hi_;) {      // last problem whose first tile <= t (binary search over the kernel-argument table)
        const int mid = (pi + hi_ + 1) >> 1;
        if (t >= ga.p[mid].tile0) pi = mid; else hi_ = mid - 1;
    }
    const GroupedProblem& pr = ga.p[pi];
    const int lt = t - pr.tile0, tm = lt % pr.mtiles, tn = lt / pr.mtiles;
    // output rows = dy columns (N), output cols = x columns (K), reduction over the M tokens
    LdCol al{pr.dy, pr.N, 0, pr.N, 0};
    LdCol bl{pr.x, pr.K, 0, pr.K, 0};
    EpStd ep{pr.dw, pr.K, 0, nullptr, nullptr, 0, 0, pr.N, nullptr, nullptr, 0, 0, 0, 1.0f};
    gemm_tile<P, LdCol, LdCol, EpStd, WM, WN, WVM, WVN>(pr.N, pr.K, tm * BM, tn * BN, 0, pr.M, 0, 1, 0, al, bl, ep, nullptr, lds);
}

// Split-K reduce + epilogue, fixed summation order.  grid (cdiv(N,64), cdiv(M,RPB), batch), 256 threads:
// DEEP = false: the 4 waves own 4 different rows and loop over the (few) slabs;
// DEEP = true : the 4 waves split the (many) slabs of ONE row and combine through LDS (tiny outputs whose
//               K was cut hundreds of ways -- the weight gradients of the conv-side layers).
template <class EP, bool DEEP>
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(int M, int N, int splits, const float* __restrict__ ws, EP ep) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, b = blockIdx.z;
    if (DEEP) {
        const int m = blockIdx.y;
        float s = 0.f;
        if (n < N)
            for (int sp = wv; sp < splits; sp += 4) s += ws[(((long)b * splits + sp) * M + m) * N + n];
        sm[wv][tx] = s;
        __syncthreads();
        if (wv == 0 && n < N) ep.store(b, m, n, (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]));
    } else {
        const int m = blockIdx.y * 4 + wv;
        if (m < M && n < N) {
            float s = 0.f;
            for (int sp = 0; sp < splits; ++sp) s += ws[(((long)b * splits + sp) * M + m) * N + n];
            ep.store(b, m, n, s);
        }
    }
}

// -------------------------------------------------------------------------------------- host launch
template <class P, class AL, class BL, class EP, int WM, int WN, int WVM, int WVN>
int launch_cfg(int M, int N, int K, int batch, AL al, BL bl, EP ep, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int BM = 16 * WM * WVM, BN = 16 * WN * WVN, SK = 8 * P::CH;
    int mt = cdiv(M, BM), nt = cdiv(N, BN);
    int ksteps = cdiv(K, SK);
    long tiles = (long)mt * nt * batch;
    int splits = 1;
    // split-K only when the tile grid leaves most CUs idle (measured at M = 432: N >= 2304 is fastest unsplit, N = 768
    // wants ~6 slabs at K = 3072 and 2-4 at K = 768); each slab gets whole prefetch rings (3 stages)
    if (tiles < 192 && ksteps >= 6) {
        splits = (int)((512 + tiles - 1) / tiles);
        if (splits > ksteps / 3) splits = ksteps / 3;
        if (splits < 1) splits = 1;
    }
    { // tuning hooks (read once): UNETR_GEMM_SPLITS forces the split count
        const int env_splits = getenv("UNETR_GEMM_SPLITS") ? atoi(getenv("UNETR_GEMM_SPLITS")) : 0;
        if (env_splits > 0) splits = std::min(env_splits, std::max(1, ksteps));
    }
    while (splits > 1 && (size_t)splits * batch * M * N * sizeof(float) > ws_bytes) --splits;
    if (splits > 1 && ws == nullptr) splits = 1;
    int steps_per = cdiv(ksteps, splits);
    if (splits > 1 && steps_per > 3) steps_per = (steps_per + 2) / 3 * 3;   // whole prefetch rings (PF = 3)
    int kper = steps_per * SK;
    splits = cdiv(K, kper);
    if (nt > 65535 || (long)batch * splits > 65535) return UNETR_ERR_ARG;
    dim3 grid(mt, nt, batch * splits);
    hipLaunchKernelGGL((gemm_kernel<P, AL, BL, EP, WM, WN, WVM, WVN>), grid, dim3(64 * WVM * WVN), 0, st,
                       M, N, K, splits, kper, al, bl, ep, ws);
    if (splits > 1) {
        if (splits > 24 && M <= 65535)
            hipLaunchKernelGGL((splitk_reduce_kernel<EP, true>), dim3(cdiv(N, 64), M, batch), dim3(256), 0, st, M, N, splits, ws, ep);
        else
            hipLaunchKernelGGL((splitk_reduce_kernel<EP, false>), dim3(cdiv(N, 64), cdiv(M, 4), batch), dim3(256), 0, st, M, N, splits, ws, ep);
    }
    return unetr_check_launch();
}

template <class P, class AL, class BL, class EP>
int launch_gemm(int M, int N, int K, int batch, AL al, BL bl, EP ep, float* ws, size_t ws_bytes, hipStream_t st) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return UNETR_ERR_ARG;
    if (M <= 16 && N <= 16) return launch_cfg<P, AL, BL, EP, 1, 1, 1, 1>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (M <= 16 && N <= 64) return launch_cfg<P, AL, BL, EP, 1, 2, 1, 2>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (M <= 16 && N > 64) return launch_cfg<P, AL, BL, EP, 1, 4, 1, 4>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (M <= 32 && N > 64) return launch_cfg<P, AL, BL, EP, 2, 4, 1, 4>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (N <= 16) return launch_cfg<P, AL, BL, EP, 4, 1, 4, 1>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (N <= 32) return launch_cfg<P, AL, BL, EP, 4, 2, 4, 1>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    const int env_cfg = getenv("UNETR_GEMM_CFG") ? atoi(getenv("UNETR_GEMM_CFG")) : 0;   // tuning hook
    if (env_cfg == 128 || (env_cfg == 0 && M >= 2048 && N >= 128))
        return launch_cfg<P, AL, BL, EP, 4, 4, 2, 2>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    return launch_cfg<P, AL, BL, EP, 2, 2, 2, 2>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
}

template <class AL, class BL, class EP>
int launch_prec(int prec, int M, int N, int K, int batch, AL al, BL bl, EP ep, float* ws, size_t ws_bytes, hipStream_t st) {
    if (prec == UNETR_PREC_BF16) return launch_gemm<PrecBF16>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (prec == UNETR_PREC_F32) return launch_gemm<PrecF32>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    if (prec == UNETR_PREC_BF16X3) return launch_gemm<PrecBF16x3>(M, N, K, batch, al, bl, ep, ws, ws_bytes, st);
    return UNETR_ERR_ARG;
}

template <class P> inline bool kvec_ok(int K) { return K % P::CH == 0; }

inline int vec_ok(const float* p, long ld, long stride) {
    return ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (stride & 3) == 0) ? 1 : 0;
}

}  // namespace

