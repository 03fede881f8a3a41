// DiceCELoss(to_onehot_y=True, softmax=True) forward + backward (unetr_segmentation_3d.py:404; MONAI 0.6.0
// defaults: include_background, squared_pred=False, jaccard=False, batch=False, reduction="mean",
// smooth_nr = smooth_dr = 1e-5, lambda_dice = lambda_ce = 1).
//
//   p = softmax(logits, 1);  y = one_hot(label)
//   per (b,c): I = sum p*y, G = sum y, Pp = sum p over the volume
//   dice = mean_{b,c} [1 - (2I + nr) / (G + Pp + dr)];  ce = mean_{b,v} -log p[b, label, v];  loss = dice + ce
//
// SIG mode = DiceCELoss(to_onehot_y=False, sigmoid=True) on a multi-label float target [B,C,*spatial]
// (unetr_segmentation_3d.py:477-482, the 4-channel MR task): the Dice term uses p = sigmoid(logits) against the target
// as given; the CE term is MONAI 0.6.0's DiceCELoss.ce with equal channel counts -- softmax cross entropy of the raw
// logits against argmax_c(target) (first maximal channel, torch.argmax's rule).
//
// HBM-bound: one pass over logits+label for the forward (fixed-order partial buffers -> reproducible), one
// pass for the backward that recomputes the softmax and writes dlogits.
#include <algorithm>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

constexpr int MAXC = 16;
constexpr int LVPB = 4096;  // voxels per block

template <int C, bool SIG>
__global__ void __launch_bounds__(256)
dicece_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ label, long V, float* __restrict__ part) {
    // part: [B][nchunk][3*C + 1]  (I[c], Pp[c], G[c], ce_sum)
    __shared__ float red[4][3 * C + 1];
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * LVPB, v1 = std::min<long>(V, v0 + LVPB);
    float accI[C], accP[C], accG[C], ce = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { accI[c] = 0.f; accP[c] = 0.f; accG[c] = 0.f; }
    // softmax / one-hot form on whole 4-voxel groups: one 16-byte load per channel and for the labels, all of an iteration's loads
    // in flight together (one voxel per iteration was 16 dependent memory round trips per thread: 19 us for 35 MB)
    const bool vec = !SIG && (V & 3) == 0 && ((((uintptr_t)logits) | ((uintptr_t)label)) & 15) == 0;
    if (vec) {
        for (long v = v0 + 4 * threadIdx.x; v < v1; v += 1024) {      // (LVPB and V are multiples of 4: whole groups)
            f32x4 zv[C];
#pragma unroll
            for (int c = 0; c < C; ++c) zv[c] = *(const f32x4*)(logits + ((long)b * C + c) * V + v);
            const f32x4 lv = *(const f32x4*)(label + (long)b * V + v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float z[C], mx = -3.0e38f;
#pragma unroll
                for (int c = 0; c < C; ++c) { z[c] = zv[c][e]; mx = fmaxf(mx, z[c]); }
                float se = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) { z[c] = expf(z[c] - mx); se += z[c]; }
                const float inv = 1.f / se;
                const int lab = (int)lv[e];
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float p = z[c] * inv;
                    accP[c] += p;
                    if (c == lab) { accI[c] += p; accG[c] += 1.f; ce -= logf(p); }
                }
            }
        }
    } else
    for (long v = v0 + threadIdx.x; v < v1; v += 256) {
        float z[C], mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < C; ++c) { z[c] = logits[((long)b * C + c) * V + v]; mx = fmaxf(mx, z[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { z[c] = expf(z[c] - mx); se += z[c]; }
        const float inv = 1.f / se;
        if constexpr (SIG) {
            float t[C], tmax = -3.0e38f;
            int lab = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                t[c] = label[((long)b * C + c) * V + v];
                if (t[c] > tmax) { tmax = t[c]; lab = c; }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float x = logits[((long)b * C + c) * V + v];
                const float sg = 1.f / (1.f + expf(-x));
                accP[c] += sg; accI[c] += sg * t[c]; accG[c] += t[c];
                if (c == lab) ce -= logf(z[c] * inv);
            }
        } else {
            const int lab = (int)label[(long)b * V + v];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float p = z[c] * inv;
                accP[c] += p;
                if (c == lab) { accI[c] += p; accG[c] += 1.f; ce -= logf(p); }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float a = wave_sum(accI[c]), p = wave_sum(accP[c]), g = wave_sum(accG[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][C + c] = p; red[wave][2 * C + c] = g; }
    }
    ce = wave_sum(ce);
    if (lane == 0) red[wave][3 * C] = ce;
    __syncthreads();
    if (threadIdx.x < 3 * C + 1)
        part[((long)b * gridDim.x + blockIdx.x) * (3 * C + 1) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// single block: reduce partials in double (fixed order: lanes stride over the chunks, shuffle tree), emit loss terms and the
// Dice gradient coefficients.  One wave per (b, c) -- eight threads walking 216 chunks each serially cost 30 us.
// coef[(b*C+c)*2+0] = dDice/dp coefficient on y:  -2/(B*C*den);  [+1] = constant term: (2I+nr)/(B*C*den^2)
__global__ void __launch_bounds__(256)
dicece_final_kernel(const float* __restrict__ part, int B, int C, int nchunk, long V, float nr, float dr,
                    float* __restrict__ out, float* __restrict__ coef) {
    __shared__ double sdice[4];
    __shared__ double sce[4];
    const int stride = 3 * C + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double dice_acc = 0.0, ce_acc = 0.0;
    for (int i = wave; i < B * C; i += 4) {
        const int b = i / C, c = i - b * C;
        double I = 0.0, Pp = 0.0, G = 0.0;
        for (int k = lane; k < nchunk; k += 64) {
            const float* p = part + ((long)b * nchunk + k) * stride;
            I += (double)p[c]; Pp += (double)p[C + c]; G += (double)p[2 * C + c];
        }
        I = wave_sum_d(I); Pp = wave_sum_d(Pp); G = wave_sum_d(G);
        const double den = G + Pp + (double)dr, num = 2.0 * I + (double)nr;
        dice_acc += 1.0 - num / den;
        if (lane == 0) {
            coef[2 * i] = (float)(-2.0 / ((double)(B * C) * den));
            coef[2 * i + 1] = (float)(num / ((double)(B * C) * den * den));
        }
    }
    for (int i = threadIdx.x; i < B * nchunk; i += 256) ce_acc += (double)part[(long)i * stride + 3 * C];
    ce_acc = wave_sum_d(ce_acc);
    if (lane == 0) { sdice[wave] = dice_acc; sce[wave] = ce_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double dice = (sdice[0] + sdice[1] + sdice[2] + sdice[3]) / (double)(B * C);
        const double ce = (sce[0] + sce[1] + sce[2] + sce[3]) / ((double)B * (double)V);
        out[0] = (float)(dice + ce); out[1] = (float)dice; out[2] = (float)ce;
    }
}

template <int C, bool SIG>
__global__ void __launch_bounds__(256)
dicece_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ label, const float* __restrict__ coef,
                  const float* __restrict__ dloss, float* __restrict__ dlogits, int B, long V) {
    const float up = dloss ? *dloss : 1.f;
    const float ce_scale = 1.f / ((float)B * (float)V);
    const long total = (long)B * V;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / V); const long v = i - (long)b * V;
        float z[C], mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < C; ++c) { z[c] = logits[((long)b * C + c) * V + v]; mx = fmaxf(mx, z[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { z[c] = expf(z[c] - mx); se += z[c]; }
        const float inv = 1.f / se;
        if constexpr (SIG) {
            float t[C], tmax = -3.0e38f;
            int lab = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                t[c] = label[((long)b * C + c) * V + v];
                if (t[c] > tmax) { tmax = t[c]; lab = c; }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float x = logits[((long)b * C + c) * V + v];
                const float sg = 1.f / (1.f + expf(-x));
                const float* cf = coef + ((long)b * C + c) * 2;
                const float dz = (cf[0] * t[c] + cf[1]) * sg * (1.f - sg) + (z[c] * inv - (c == lab ? 1.f : 0.f)) * ce_scale;
                dlogits[((long)b * C + c) * V + v] = up * dz;
            }
            continue;
        }
        const int lab = (int)label[(long)b * V + v];
        float gp[C], dot = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            z[c] *= inv;                                   // p_c
            const float* cf = coef + ((long)b * C + c) * 2;
            gp[c] = (c == lab ? cf[0] : 0.f) + cf[1];      // dDice/dp_c
            dot += z[c] * gp[c];
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float dz = z[c] * (gp[c] - dot) + (z[c] - (c == lab ? 1.f : 0.f)) * ce_scale;
            dlogits[((long)b * C + c) * V + v] = up * dz;
        }
    }
}

}  // namespace

#define DICE_DISPATCH(C_, CALL) \
    switch (C_) {               \
        case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;       \
        case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;       \
        case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
        case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; case 16: CALL(16); break; \
        default: return UNETR_ERR_UNSUPPORTED;                                                                \
    }

extern "C" int unetr_dicece_fwd(const float* logits, const float* label, int B, int C, long V, int sigmoid_multilabel,
                                float smooth_nr, float smooth_dr, float* out, float* coef, float* ws, size_t ws_bytes,
                                void* stream) {
    if (!logits || !label || !out || !coef || B <= 0 || V <= 0 || B > 65535) return UNETR_ERR_ARG;
    if (C < 1 || C > MAXC) return UNETR_ERR_UNSUPPORTED;
    int nchunk = cdiv(V, LVPB);
    if (!ws || (size_t)B * nchunk * (3 * C + 1) * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define CALL_F(CC) hipLaunchKernelGGL((dicece_fwd_kernel<CC, false>), dim3(nchunk, B), dim3(256), 0, st, logits, label, V, ws)
#define CALL_FS(CC) hipLaunchKernelGGL((dicece_fwd_kernel<CC, true>), dim3(nchunk, B), dim3(256), 0, st, logits, label, V, ws)
    if (sigmoid_multilabel) { DICE_DISPATCH(C, CALL_FS) } else { DICE_DISPATCH(C, CALL_F) }
    hipLaunchKernelGGL(dicece_final_kernel, dim3(1), dim3(256), 0, st, ws, B, C, nchunk, V, smooth_nr, smooth_dr, out, coef);
    return unetr_check_launch();
}

extern "C" int unetr_dicece_bwd(const float* logits, const float* label, const float* coef, const float* dloss,
                                float* dlogits, int B, int C, long V, int sigmoid_multilabel, void* stream) {
    if (!logits || !label || !coef || !dlogits || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    if (C < 1 || C > MAXC) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    long total = (long)B * V;
    int blocks = (int)std::max<long>(1, std::min<long>((total + 255) / 256, 8192));
#define CALL_B(CC) hipLaunchKernelGGL((dicece_bwd_kernel<CC, false>), dim3(blocks), dim3(256), 0, st, logits, label, coef, dloss, dlogits, B, V)
#define CALL_BS(CC) hipLaunchKernelGGL((dicece_bwd_kernel<CC, true>), dim3(blocks), dim3(256), 0, st, logits, label, coef, dloss, dlogits, B, V)
    if (sigmoid_multilabel) { DICE_DISPATCH(C, CALL_BS) } else { DICE_DISPATCH(C, CALL_B) }
    return unetr_check_launch();
}
