// Self-supervised ranking pre-training losses of the reference (unetr_ranking_pretraining_3d.py:59-133, 202-236),
// fused.  The reference slices 4 partitions x 4 volumes out of the feature map, builds 576 (reference, similar,
// dissimilar) triplets and evaluates two cosine similarities per triplet per channel in a Python loop
// (~2 300 kernel launches forward).  Every term is a function of the per-channel 16 x 16 cosine Gram matrix of the
// 16 slice vectors, so here:
//
//   rank_gram_kernel   D_c[i][j] = <v_i, v_j> over the slice positions              (one pass over the 16 slices)
//   rank_loss_kernel   G = D / max(|v_i||v_j|, eps)   (torch 1.7.1 CosineSimilarity formula, eps 1e-6)
//                      BT:          sum_{p} sum_{i != j in p} sum_{k not in p} mean_c softplus(-(G_ij - G_ik)/T)
//                      contrastive: sum_{p} sum_{i != j in p} mean_c -log( e^{G_ij/T} / (36 sum_k e^{G_ik/T} + e^{G_ij/T}) )
//                      (the reference's denominator runs over its whole `dissimilar` list for every pair: each of the
//                      16 slices appears in it 36 times)
//                      + dL/dD_c[i][j] kept for backward
//   rank_bwd_kernel    dv_i = sum_j (W_ij + W_ji) v_j written into the 16 slices of the (zeroed) feature gradient
//
// Slice vector i = 4*p + j: partition p (index init_idx + p*part along `slice_dim`), volume j of the 4-volume batch.
#include <algorithm>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

constexpr int NV = 16;            // slice vectors per channel
constexpr int NPAIR = 136;        // upper triangle incl. diagonal
constexpr int RCHUNK = 1024;      // positions per workgroup in the gram pass

struct RankGeom {
    long sb, sc;          // batch / channel strides of feat (elements)
    long sa, sbb;         // strides of the two remaining spatial dims
    long ss;              // stride of the sliced dim
    int na, nb;           // extents of the two remaining spatial dims (R = na*nb)
    int init_idx, part;   // slice index of partition p = init_idx + p*part
    __device__ __forceinline__ long off(int i, int c, int r) const {
        const int p = i >> 2, j = i & 3;
        const int a = r / nb, b = r - a * nb;
        return (long)j * sb + (long)c * sc + (long)(init_idx + p * part) * ss + (long)a * sa + (long)b * sbb;
    }
};

__device__ __forceinline__ int pair_index(int i, int j) {  // i <= j
    return i * NV - (i * (i - 1)) / 2 + (j - i);
}

__global__ void __launch_bounds__(256)
rank_gram_kernel(const float* __restrict__ feat, RankGeom g, int R, float* __restrict__ part) {
    __shared__ float red[4][NPAIR];
    const int c = blockIdx.x, chunk = blockIdx.y;
    const int r0 = chunk * RCHUNK, r1 = min(R, r0 + RCHUNK);
    float acc[NPAIR];
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) acc[k] = 0.f;
    for (int r = r0 + threadIdx.x; r < r1; r += 256) {
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = feat[g.off(i, c, r)];
        int k = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int j = i; j < NV; ++j) acc[k++] += v[i] * v[j];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NPAIR; ++k) {
        float s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < NPAIR)
        part[((long)c * gridDim.y + chunk) * NPAIR + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one 64-lane workgroup per channel
__global__ void __launch_bounds__(64)
rank_loss_kernel(const float* __restrict__ part, int nchunk, int C, float inv_T, float eps, int kind,
                 float* __restrict__ loss_c, float* __restrict__ Wout) {
    __shared__ float D[NV][NV], G[NV][NV], dG[NV][NV], nrm[NV], dn[NV];
    const int c = blockIdx.x, lane = threadIdx.x;
    for (int k = lane; k < NPAIR; k += 64) {
        double s = 0.0;
        for (int q = 0; q < nchunk; ++q) s += (double)part[((long)c * nchunk + q) * NPAIR + k];
        // invert pair_index
        int i = 0, rem = k;
        while (rem >= NV - i) { rem -= NV - i; ++i; }
        int j = i + rem;
        D[i][j] = (float)s; D[j][i] = (float)s;
    }
    __syncthreads();
    if (lane < NV) nrm[lane] = sqrtf(D[lane][lane]);
    __syncthreads();
    for (int e = lane; e < NV * NV; e += 64) {
        int i = e >> 4, j = e & 15;
        G[i][j] = D[i][j] / fmaxf(nrm[i] * nrm[j], eps);
    }
    __syncthreads();
    const float invC = 1.f / (float)C;
    float lsum = 0.f;
    // loss terms + dL/dG, one (i, j) entry per lane-iteration, fixed order (no atomics)
    for (int e = lane; e < NV * NV; e += 64) {
        const int i = e >> 4, j = e & 15, pi = i >> 2, pj = j >> 2;
        float d = 0.f;
        if (kind == 0) {  // Bradley-Terry
            if (pi == pj && i != j) {           // (ref = i, sim = j): sum over dissimilar k
                for (int k = 0; k < NV; ++k)
                    if ((k >> 2) != pi) {
                        float z = -(G[i][j] - G[i][k]) * inv_T;                     // softplus(z)
                        lsum += (z > 0.f ? z : 0.f) + log1pf(expf(-fabsf(z)));
                        float sg = 1.f / (1.f + expf(-z));                          // d softplus / dz
                        d += sg * (-inv_T);
                    }
            } else if (pi != pj) {              // (ref = i, dis = j): sum over similar j' in part(i)
                for (int jj = pi * 4; jj < pi * 4 + 4; ++jj)
                    if (jj != i) {
                        float z = -(G[i][jj] - G[i][j]) * inv_T;
                        float sg = 1.f / (1.f + expf(-z));
                        d += sg * inv_T;
                    }
            }
        } else {          // contrastive
            // S_i = 36 * sum_k exp(G_ik / T); pair (i, j) in one partition: l = -log(num / (S_i + num)), num = exp(G_ij/T)
            float S = 0.f;
            for (int k = 0; k < NV; ++k) S += expf(G[i][k] * inv_T);
            S *= 36.f;
            if (pi == pj && i != j) {
                float num = expf(G[i][j] * inv_T);
                lsum += -logf(num / (S + num));
                d += -inv_T * S / (S + num);                                         // d l_ij / d G_ij (numerator part)
            }
            // every pair (i, j') of row i sees G_ij through S_i: d l_ij' / d G_ij = 36 e_ij / T / (S + num_ij')
            float eij = expf(G[i][j] * inv_T);
            for (int jj = pi * 4; jj < pi * 4 + 4; ++jj)
                if (jj != i) {
                    float num = expf(G[i][jj] * inv_T);
                    d += 36.f * eij * inv_T / (S + num);
                }
        }
        // the reference's ContrastiveLoss iterates zip(reference, similar) over all 576 triplet entries, i.e. every
        // ordered (ref, sim) pair 12 times (unetr_ranking_pretraining_3d.py:225)
        dG[i][j] = d * invC * (kind == 1 ? 12.f : 1.f);
    }
    lsum = wave_sum(lsum) * (kind == 1 ? 12.f : 1.f);
    if (lane == 0) loss_c[c] = lsum * invC;
    __syncthreads();
    // chain rule G = D / max(n_i n_j, eps), n_i = sqrt(D_ii)
    if (lane < NV) {
        const int i = lane;
        float s = 0.f;
        for (int j = 0; j < NV; ++j) {
            if (j == i) continue;                                  // G_ii is the constant 1 (or D_ii/eps with zero gradient in torch as well)
            if (nrm[i] * nrm[j] > eps) s += -(dG[i][j] + dG[j][i]) * D[i][j] / (nrm[i] * nrm[i] * nrm[j]);
        }
        dn[i] = s;
    }
    __syncthreads();
    for (int e = lane; e < NV * NV; e += 64) {
        const int i = e >> 4, j = e & 15;
        float w;
        if (i == j) {
            float denom = fmaxf(nrm[i] * nrm[i], eps);
            float self = (nrm[i] * nrm[i] > eps) ? 0.f : dG[i][i] / denom;         // only the degenerate tiny-norm case has a gradient
            w = self + (nrm[i] > 0.f ? dn[i] / (2.f * nrm[i]) : 0.f);
        } else {
            w = dG[i][j] / fmaxf(nrm[i] * nrm[j], eps);
        }
        Wout[(long)c * NV * NV + e] = w;
    }
}

__global__ void rank_sum_kernel(const float* __restrict__ loss_c, int C, float* __restrict__ out) {
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) s += loss_c[c];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ void __launch_bounds__(256)
rank_bwd_kernel(const float* __restrict__ feat, RankGeom g, int R, const float* __restrict__ W, const float* __restrict__ dloss,
                float* __restrict__ dfeat) {
    __shared__ float Ws[NV][NV];
    const int c = blockIdx.x;
    const float up = dloss ? *dloss : 1.f;
    {
        int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        Ws[i][j] = (W[(long)c * NV * NV + i * NV + j] + W[(long)c * NV * NV + j * NV + i]) * up;
    }
    __syncthreads();
    for (int r = blockIdx.y * RCHUNK + threadIdx.x; r < min(R, (int)(blockIdx.y + 1) * RCHUNK); r += 256) {
        float v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = feat[g.off(i, c, r)];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j) s += Ws[i][j] * v[j];
            dfeat[g.off(i, c, r)] = s;
        }
    }
}

int make_geom(int C, int S1, int S2, int S3, int slice_dim, int init_idx, RankGeom* g, int* R) {
    if (slice_dim < 2 || slice_dim > 4) return UNETR_ERR_ARG;
    const int dims[3] = {S1, S2, S3};
    const long strides[3] = {(long)S2 * S3, (long)S3, 1};
    const int sd = slice_dim - 2;
    const int part = dims[sd] / 4;
    if (part < 1 || init_idx < 0 || init_idx >= part) return UNETR_ERR_ARG;
    int o[2], k = 0;
    for (int d = 0; d < 3; ++d)
        if (d != sd) o[k++] = d;
    g->sb = (long)C * S1 * S2 * S3; g->sc = (long)S1 * S2 * S3;
    g->ss = strides[sd]; g->sa = strides[o[0]]; g->sbb = strides[o[1]];
    g->na = dims[o[0]]; g->nb = dims[o[1]];
    g->init_idx = init_idx; g->part = part;
    *R = g->na * g->nb;
    return UNETR_OK;
}

}  // namespace

extern "C" size_t unetr_ranking_workspace_floats(int C, int S1, int S2, int S3, int slice_dim) {
    const int dims[3] = {S1, S2, S3};
    long R = 1;
    for (int d = 0; d < 3; ++d)
        if (d != slice_dim - 2) R *= dims[d];
    return (size_t)C * cdiv(R, RCHUNK) * NPAIR + C;
}

extern "C" int unetr_ranking_loss_fwd(const float* feat, int C, int S1, int S2, int S3, int slice_dim, int init_idx,
                                      float temperature, int kind, float* loss, float* W, float* ws, size_t ws_floats,
                                      void* stream) {
    if (!feat || !loss || !W || !ws || C <= 0 || temperature <= 0.f || (kind != 0 && kind != 1)) return UNETR_ERR_ARG;
    RankGeom g; int R;
    if (int e = make_geom(C, S1, S2, S3, slice_dim, init_idx, &g, &R)) return e;
    const int nchunk = cdiv(R, RCHUNK);
    if ((size_t)C * nchunk * NPAIR + C > ws_floats) return UNETR_ERR_WORKSPACE;
    if (nchunk > 65535) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* loss_c = ws + (size_t)C * nchunk * NPAIR;
    hipLaunchKernelGGL(rank_gram_kernel, dim3(C, nchunk), dim3(256), 0, st, feat, g, R, ws);
    hipLaunchKernelGGL(rank_loss_kernel, dim3(C), dim3(64), 0, st, ws, nchunk, C, 1.f / temperature, 1e-6f, kind, loss_c, W);
    hipLaunchKernelGGL(rank_sum_kernel, dim3(1), dim3(64), 0, st, loss_c, C, loss);
    return unetr_check_launch();
}

extern "C" int unetr_ranking_loss_bwd(const float* feat, int C, int S1, int S2, int S3, int slice_dim, int init_idx,
                                      const float* W, const float* dloss, float* dfeat, void* stream) {
    if (!feat || !W || !dfeat || C <= 0) return UNETR_ERR_ARG;
    RankGeom g; int R;
    if (int e = make_geom(C, S1, S2, S3, slice_dim, init_idx, &g, &R)) return e;
    hipLaunchKernelGGL(rank_bwd_kernel, dim3(C, cdiv(R, RCHUNK)), dim3(256), 0, (hipStream_t)stream, feat, g, R, W, dloss, dfeat);
    return unetr_check_launch();
}
