// unetr_gemm: Linear fwd / dgrad / wgrad and the 1x1x1 conv through the MFMA GEMM family (gemm_kernel.hpp).
#include "gemm_kernel.hpp"

extern "C" int unetr_abi_version(void) { return UNETR_ABI_VERSION; }

extern "C" int unetr_gemm(const unetr_gemm_desc* d, const float* A, const float* B, float* C,
                          float* ws, size_t ws_bytes, void* stream) {
    if (!d || !A || !B || !C) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    EpStd ep{C, d->ldc, d->strideC, d->bias, d->res, d->ldr, d->strideR, d->res_mod > 0 ? d->res_mod : d->M,
             d->pre, d->aux, d->ldaux, d->act, d->accumulate, d->alpha};
    if (d->act == 2 && !d->aux) return UNETR_ERR_ARG;
    const int M = d->M, N = d->N, K = d->K, bt = d->batch;
    const bool kv = (K % 8) == 0;
    if (d->b_x3words && (d->prec != UNETR_PREC_BF16X3 || bt != 1 || d->a_trans)) return UNETR_ERR_ARG;
    if (d->prec == UNETR_PREC_BF16X3 && bt == 1 && !d->a_trans && M >= 32 &&
        (d->b_x3words || !(getenv("UNETR_X3_GEMM_DMA") && atoi(getenv("UNETR_X3_GEMM_DMA")) == 0))) {
        // bf16x3 Linear forward (B [N,K]) / data gradient (B [K,N]) on the LDS-DMA kernel of gemm_bf16.hip where the shape allows it
        unetr_gemm_bf16_desc q{};
        q.M = M; q.N = N; q.K = K; q.b_kn = d->b_trans ? 1 : 0;
        q.lda = d->lda; q.ldb = d->ldb; q.ldc = d->ldc; q.ldcb = 0;
        q.bias = d->bias; q.res = d->res; q.ldr = d->ldr; q.res_mod = d->res_mod;
        q.pre = d->pre; q.aux = d->aux; q.ldaux = d->ldaux; q.act = d->act; q.accumulate = d->accumulate; q.alpha = d->alpha;
        const int rc = unetr_gemm_x3_dma(&q, A, B, d->b_x3words, C, ws, ws_bytes, stream);
        if (rc != UNETR_ERR_UNSUPPORTED || d->b_x3words) return rc;       // (only this kernel reads a word shadow)
    }
    if (d->b_x3words) return UNETR_ERR_UNSUPPORTED;
    if (!d->a_trans && !d->b_trans) {
        if (kv && vec_ok(A, d->lda, d->strideA) && vec_ok(B, d->ldb, d->strideB)) {
            LdRow al{A, d->lda, d->strideA, M, 1};
            LdRow bl{B, d->ldb, d->strideB, N, 1};
            return launch_prec(d->prec, M, N, K, bt, al, bl, ep, ws, ws_bytes, st);
        }
        LdRowS al{A, d->lda, d->strideA, M, 0};
        LdRowS bl{B, d->ldb, d->strideB, N, 0};
        return launch_prec(d->prec, M, N, K, bt, al, bl, ep, ws, ws_bytes, st);
    } else if (!d->a_trans && d->b_trans) {
        LdCol bl{B, d->ldb, d->strideB, N, 0};
        if (kv && vec_ok(A, d->lda, d->strideA)) {
            LdRow al{A, d->lda, d->strideA, M, 1};
            return launch_prec(d->prec, M, N, K, bt, al, bl, ep, ws, ws_bytes, st);
        }
        LdRowS al{A, d->lda, d->strideA, M, 0};
        return launch_prec(d->prec, M, N, K, bt, al, bl, ep, ws, ws_bytes, st);
    } else if (d->a_trans && d->b_trans) {
        LdCol al{A, d->lda, d->strideA, M, 0};
        LdCol bl{B, d->ldb, d->strideB, N, 0};
        return launch_prec(d->prec, M, N, K, bt, al, bl, ep, ws, ws_bytes, st);
    }
    return UNETR_ERR_UNSUPPORTED;
}

// dw_i[N_i, K_i] = dy_i[M_i, N_i]^T * x_i[M_i, K_i] for i < n, one launch per <= GROUP_MAX problems
extern "C" int unetr_gemm_grouped_wgrad(const unetr_grouped_problem* probs, int n, int prec, void* stream) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    constexpr int BM = 128, BN = 128;
    for (int base = 0; base < n; base += GROUP_MAX) {
        GroupedArgs ga;
        ga.n = std::min(GROUP_MAX, n - base);
        int tiles = 0;
        for (int i = 0; i < ga.n; ++i) {
            const unetr_grouped_problem& q = probs[base + i];
            if (!q.dy || !q.x || !q.dw || q.M <= 0 || q.N <= 0 || q.K <= 0) return UNETR_ERR_ARG;
            GroupedProblem& g = ga.p[i];
            g.dy = q.dy; g.x = q.x; g.dw = q.dw; g.M = q.M; g.N = q.N; g.K = q.K;
            g.tile0 = tiles; g.mtiles = cdiv(q.N, BM);
            tiles += g.mtiles * cdiv(q.K, BN);
        }
        if (prec == UNETR_PREC_BF16)
            hipLaunchKernelGGL((gemm_grouped_wgrad_kernel<PrecBF16, 4, 4, 2, 2>), dim3(tiles), dim3(256), 0, st, ga);
        else if (prec == UNETR_PREC_F32)
            hipLaunchKernelGGL((gemm_grouped_wgrad_kernel<PrecF32, 4, 4, 2, 2>), dim3(tiles), dim3(256), 0, st, ga);
        else if (prec == UNETR_PREC_BF16X3)
            hipLaunchKernelGGL((gemm_grouped_wgrad_kernel<PrecBF16x3, 4, 4, 2, 2>), dim3(tiles), dim3(256), 0, st, ga);
        else
            return UNETR_ERR_ARG;
    }
    return unetr_check_launch();
}
