// Fused kernels of the ViT encoder for small token counts (batch 2 x 216 tokens = 432 rows), bf16 precision mode.
// Reference semantics: MONAI TransformerBlock / SABlock / MLPBlock as built at /root/reference/unetr.py:78-89
// (x + attn(norm1(x)); x + mlp(norm2(x)); nn.LayerNorm eps 1e-5; exact-erf GELU).
//
// MEASURED NEGATIVE RESULT, OFF BY DEFAULT (functional.fused_ln_enabled): on MI355X at 432 rows this kernel costs 13.3 us for
// norm1 -> qkv against 3.1 (LayerNorm) + 6.6 (GEMM) as two launches, 17.5 against 3.1 + 10.5 for norm2 -> linear1.  Every
// column-tile workgroup re-reads its 64 rows as fp32 (twice the bytes of the bf16 rows the plain GEMM stages) through the
// ~70 GB/s one CU gets from L2, which costs more than the launch boundary it removes.  Kept, tested, as an option
// (UNETR_AMD_FUSED_LN=1) and as the record of the experiment.
//
// The idea it tests: at 432 rows every per-layer GEMM is a single wave of workgroups whose time is launch boundary +
// pipeline fill + the L2 -> LDS traffic of its tile, so remove a dependent launch by making LayerNorm the GEMM's prologue:
//
//   unetr_ln_gemm_bf16   y = epilogue(LayerNorm(x) W^T + b)     LayerNorm is the GEMM's prologue
//       one workgroup = 64 rows x BN columns, 8 waves.  The workgroup's weight tile starts streaming into an LDS ring by
//       LDS-DMA (global_load_lds_dwordx4, swizzle on the source address) BEFORE anything else, so the HBM latency of the
//       (cold) weights hides under the prologue: each wave normalises 8 of the 64 rows (24 f32x4 loads in flight per
//       lane, two-pass mean / variance with wave64 shuffles) and writes them as bf16 into an LDS image of the whole K
//       extent (K <= 1024: 64 x K x 2 B <= 128 KB), which the K loop then reads without further global traffic.  The
//       workgroups of column tile 0 also write the normalised rows (bf16), mean and rstd for backward.
//
// Tile -> workgroup map: workgroup L runs on XCD L % 8 (observed, used for speed only); every XCD gets a contiguous run
// of tiles with the row tile fastest, so the weight columns an XCD touches are few (read from HBM once chip-wide) and the
// small activation matrix is what gets shared through L2 / Infinity Cache.
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

struct LnGemmArgs {
    const float* x; long ldx;
    const float* gamma; const float* beta; float eps;
    const uint16_t* W; long ldw;
    const float* bias; int act;
    float* pre; long ldpre;
    uint16_t* Cb; long ldcb;
    float* C; long ldc;
    uint16_t* xn; float* mean; float* rstd;
    int M, N, K, mt, nt;
};

__device__ __forceinline__ bool tile_of(int L, int mt, int nt, int& tm, int& tn) {
    const int T = mt * nt, per = (T + 7) >> 3;
    const int t = (L & 7) * per + (L >> 3);
    if ((L >> 3) >= per || t >= T) return false;
    tm = t % mt; tn = t / mt;
    return true;
}

constexpr int LG_BM = 64;

template <int BN, int NS>
__global__ void __launch_bounds__(512)
ln_gemm_kernel(LnGemmArgs a) {
    constexpr int NT = 512, BM = LG_BM, BK = 64;
    constexpr int STAGE = BN * 128;                       // bytes of one weight stage: BN rows x 64 k bf16
    constexpr int G = STAGE / 16 / NT;                    // LDS-DMA instructions per thread per stage
    static_assert(STAGE % (16 * NT) == 0 && G >= 1, "whole wave DMAs");
    static_assert((NS - 2) * G <= 63, "vmcnt is a 6-bit counter");
    constexpr int NJ = BN / 32;                           // 16-column tiles per wave (waves: 4 along m x 2 along n)
    extern __shared__ __attribute__((aligned(1024))) char lds[];

    int tm, tn;
    if (!tile_of(blockIdx.x, a.mt, a.nt, tm, tn)) return;
    const int m0 = tm * BM, n0 = tn * BN, K = a.K, nk = K / BK;
    char* ring = lds + nk * (BM * 128);                   // A image: nk panels of [64 rows][128 B], then the weight ring
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;

    // ---- weight ring: start the first NS-1 stages now
    const uint16_t* bsrc[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int id = tid + i * NT, r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
        bsrc[i] = a.W + (long)min(n0 + r, a.N - 1) * a.ldw + c * 8;
    }
    auto issue = [&](int kt, int buf) {
        char* lb = ring + buf * STAGE;
#pragma unroll
        for (int i = 0; i < G; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[i] + (long)kt * BK), (lds_void_t*)(lb + (wave * 64 + i * NT) * 16), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);

    // ---- LayerNorm prologue: wave w normalises rows 8w .. 8w+7 of the tile; a lane owns 8-float chunks lane and lane+64
    {
        const int nch = K >> 3;
        const int c0 = lane, c1 = lane + 64;
        const bool v0 = c0 < nch, v1 = c1 < nch;           // K <= 1024: at most two chunks per lane; K < 512: some lanes idle
        const int c0c = v0 ? c0 : 0, c1c = v1 ? c1 : 0;
        f32x4 v[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* xr = a.x + (long)min(m0 + wave * 8 + i, a.M - 1) * a.ldx;
            v[i][0] = *(const f32x4*)(xr + c0c * 8); v[i][1] = *(const f32x4*)(xr + c0c * 8 + 4);
            v[i][2] = *(const f32x4*)(xr + c1c * 8); v[i][3] = *(const f32x4*)(xr + c1c * 8 + 4);
        }
        const f32x4 g0 = *(const f32x4*)(a.gamma + c0c * 8), g1 = *(const f32x4*)(a.gamma + c0c * 8 + 4);
        const f32x4 g2 = *(const f32x4*)(a.gamma + c1c * 8), g3 = *(const f32x4*)(a.gamma + c1c * 8 + 4);
        const f32x4 b0 = *(const f32x4*)(a.beta + c0c * 8), b1 = *(const f32x4*)(a.beta + c0c * 8 + 4);
        const f32x4 b2 = *(const f32x4*)(a.beta + c1c * 8), b3 = *(const f32x4*)(a.beta + c1c * 8 + 4);
        const float invK = 1.0f / (float)K;
        float mu[8], rs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s = v0 ? (v[i][0][0] + v[i][0][1] + v[i][0][2] + v[i][0][3]) + (v[i][1][0] + v[i][1][1] + v[i][1][2] + v[i][1][3]) : 0.f;
            if (v1) s += (v[i][2][0] + v[i][2][1] + v[i][2][2] + v[i][2][3]) + (v[i][3][0] + v[i][3][1] + v[i][3][2] + v[i][3][3]);
            mu[i] = wave_sum(s) * invK;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float q = 0.f;
#pragma unroll
            for (int h = 0; h < 4; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][h][e] - mu[i]; q += (h < 2 ? v0 : v1) ? d * d : 0.f; }
            rs[i] = rsqrtf(wave_sum(q) * invK + a.eps);
        }
        const bool keep = a.xn != nullptr && tn == 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = wave * 8 + i, gm = m0 + r;
            float o[16];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o[e] = (v[i][0][e] - mu[i]) * rs[i] * g0[e] + b0[e];
                o[4 + e] = (v[i][1][e] - mu[i]) * rs[i] * g1[e] + b1[e];
                o[8 + e] = (v[i][2][e] - mu[i]) * rs[i] * g2[e] + b2[e];
                o[12 + e] = (v[i][3][e] - mu[i]) * rs[i] * g3[e] + b3[e];
            }
            const u32x4 p0 = PrecBF16::pack(o), p1 = PrecBF16::pack(o + 8);
            if (v0) *(u32x4*)(lds + (c0 >> 3) * (BM * 128) + lds_tile_off(r, c0 & 7)) = p0;
            if (v1) *(u32x4*)(lds + (c1 >> 3) * (BM * 128) + lds_tile_off(r, c1 & 7)) = p1;
            if (keep && gm < a.M) {
                if (v0) *(u32x4*)(a.xn + (long)gm * K + c0 * 8) = p0;
                if (v1) *(u32x4*)(a.xn + (long)gm * K + c1 * 8) = p1;
                if (lane == 0) { a.mean[gm] = mu[i]; a.rstd[gm] = rs[i]; }
            }
        }
    }

    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- K loop: A fragments from the resident image, weight fragments from the ring (counted vmcnt + raw barrier: the
    // barrier publishes stage kt and the A image, and frees the ring buffer the DMA of stage kt+NS-1 overwrites)
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const char* la = lds + kt * (BM * 128);
        const char* lb = ring + (kt % NS) * STAGE;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const u32x4 af = *(const u32x4*)(la + lds_tile_off(wm * 16 + (lane & 15), kb * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const u32x4 bf = *(const u32x4*)(lb + lds_tile_off((wn * NJ + j) * 16 + (lane & 15), kb * 4 + (lane >> 4)));
                PrecBF16::mma(acc[j], bf, af);          // transposed tile: a lane holds 4 consecutive n of one m
            }
        }
    }

    // ---- epilogue: lane (c, g) holds C[m = tile row c][n = 4g .. 4g+3] of each 16x16 tile
    const int m = m0 + wm * 16 + (lane & 15);
    if (m < a.M) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wn * NJ + j) * 16 + 4 * (lane >> 4);
            if (n >= a.N) continue;
            f32x4 v = acc[j];
            if (a.bias) v += *(const f32x4*)(a.bias + n);
            if (a.pre) *(f32x4*)(a.pre + (long)m * a.ldpre + n) = v;
            if (a.act == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
            }
            if (a.C) *(f32x4*)(a.C + (long)m * a.ldc + n) = v;
            if (a.Cb) *(bf16x4*)(a.Cb + (long)m * a.ldcb + n) = __builtin_convertvector(v, bf16x4);
        }
    }
}

template <int BN, int NS>
int launch_ln_gemm(LnGemmArgs& a, hipStream_t st) {
    a.mt = cdiv(a.M, LG_BM);
    a.nt = cdiv(a.N, BN);
    const size_t lds = (size_t)(a.K / 64) * (LG_BM * 128) + (size_t)NS * BN * 128;
    auto kern = ln_gemm_kernel<BN, NS>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int per = cdiv((long)a.mt * a.nt, 8);
    hipLaunchKernelGGL(kern, dim3(per * 8), dim3(512), lds, st, a);
    return unetr_check_launch();
}

}  // namespace

extern "C" int unetr_ln_gemm_bf16(const unetr_ln_gemm_desc* d, void* stream) {
    if (!d || !d->x || !d->gamma || !d->beta || !d->W || (!d->C && !d->Cb)) return UNETR_ERR_ARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return UNETR_ERR_ARG;
    if (d->K % 64 || d->K > 1024 || d->N % 4 || d->ldx % 4 || d->ldw % 8) return UNETR_ERR_UNSUPPORTED;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!al16(d->x) || !al16(d->gamma) || !al16(d->beta) || !al16(d->W) || (d->bias && !al16(d->bias))) return UNETR_ERR_UNSUPPORTED;
    if ((d->pre && (!al16(d->pre) || d->ldpre % 4)) || (d->C && (!al16(d->C) || d->ldc % 4)) ||
        (d->Cb && (((uintptr_t)d->Cb & 7) || d->ldcb % 4)) || (d->xn && !al16(d->xn)))
        return UNETR_ERR_UNSUPPORTED;
    if (d->xn && (!d->mean || !d->rstd)) return UNETR_ERR_ARG;
    LnGemmArgs a{d->x, d->ldx, d->gamma, d->beta, d->eps, (const uint16_t*)d->W, d->ldw, d->bias, d->act, d->pre, d->ldpre,
                 (uint16_t*)d->Cb, d->ldcb, d->C, d->ldc, (uint16_t*)d->xn, d->mean, d->rstd, d->M, d->N, d->K, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    // one wave of workgroups where possible: 64-column tiles unless that gives more workgroups than CUs
    const long mt = cdiv(d->M, LG_BM);
    int bn = (mt * cdiv(d->N, 64) <= 256 || d->N < 128) ? 64 : 128;
    if (const char* e = getenv("UNETR_LNGEMM_BN")) { const int v = atoi(e); if (v == 64 || v == 128) bn = v; }
    // ring depth: as deep as the 160 KB LDS allows next to the resident A image (K = 768: 96 KB)
    const size_t aimg = (size_t)(d->K / 64) * (LG_BM * 128);
    if (bn == 64) {
        if (aimg + 7 * 64 * 128 <= 160 * 1024 - 1024) return launch_ln_gemm<64, 7>(a, st);
        return launch_ln_gemm<64, 3>(a, st);
    }
    if (aimg + 3 * 128 * 128 <= 160 * 1024 - 1024) return launch_ln_gemm<128, 3>(a, st);
    return launch_ln_gemm<128, 2>(a, st);
}
