// unetr_gemm: Linear fwd / dgrad / wgrad and the 1x1x1 conv through the MFMA GEMM family (gemm_kernel.hpp).
#include "gemm_kernel.hpp"

extern "C" int unetr_abi_version(void) { return 1; }

extern "C" int unetr_gemm(const unetr_gemm_desc* d, const float* A, const float* B, float* C,
                          float* ws, size_t ws_bytes, void* stream) {
    if (!d || !A || !B || !C) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    EpStd ep{C, d->ldc, d->strideC, d->bias, d->res, d->ldr, d->strideR, d->res_mod > 0 ? d->res_mod : d->M,
             d->pre, d->aux, d->ldaux, d->act, d->accumulate, d->alpha};
    if (d->act == 2 && !d->aux) return UNETR_ERR_ARG;
    const int M = d->M, N = d->N, K = d->K, bt = d->batch;
    const bool kv = (K % 8) == 0;
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
