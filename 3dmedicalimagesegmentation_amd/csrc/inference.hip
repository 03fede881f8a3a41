// Validation-side kernels around the forward pass (unetr_segmentation_3d.py:103-132): blending of sliding-window
// predictions (monai.inferers.sliding_window_inference, MONAI 0.6.0, called at :110) and the counts behind
// monai.metrics.DiceMetric on argmax / one-hot predictions (:405-406, :485-486, :118-125).  All HBM-bound streaming
// passes; the window forward itself is the training hot path run without autograd.
#include <algorithm>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

// out[b, c, z0+z, y0+y, x0+x] += w(z,y,x) * seg[c, z, y, x];  count[b, z0+z, y0+y, x0+x] += w(z,y,x)
// One launch per window: windows of one volume overlap, and MONAI adds them in window order.
__global__ void __launch_bounds__(256)
sw_accumulate_kernel(const float* __restrict__ seg, const float* __restrict__ imp, float* __restrict__ out,
                     float* __restrict__ count, int C, int rz, int ry, int rx, int D, int H, int W, int z0, int y0, int x0) {
    const long rv = (long)rz * ry * rx, V = (long)D * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rv; i += (long)gridDim.x * 256) {
        const int x = (int)(i % rx), y = (int)((i / rx) % ry), z = (int)(i / ((long)rx * ry));
        const long o = ((long)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
        const float w = imp ? imp[i] : 1.f;
        for (int c = 0; c < C; ++c) out[(long)c * V + o] += w * seg[(long)c * rv + i];
        count[o] += w;
    }
}

__global__ void __launch_bounds__(256)
sw_finalize_kernel(float* __restrict__ out, const float* __restrict__ count, int B, int C, long V) {
    const long total = (long)B * C * V;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long v = i % V, b = i / ((long)C * V);
        out[i] = out[i] / count[b * V + v];
    }
}

// per (b, chunk): [C][3] partial sums of (pred * y, pred, y).  LOGITS: pred = one_hot(argmax_c logits) (first maximal
// channel, torch.argmax's rule) and y = one_hot(label ids); otherwise pred / y are taken as given ([B,C,V] each).
constexpr int DVPB = 4096;
constexpr int DMAXC = 16;
template <bool LOGITS>
__global__ void __launch_bounds__(256)
dice_counts_kernel(const float* __restrict__ pred, const float* __restrict__ y, int C, long V, float* __restrict__ part) {
    __shared__ float red[4][3 * DMAXC];
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * DVPB, v1 = std::min<long>(V, v0 + DVPB);
    float acc[3 * DMAXC];
    for (int k = 0; k < 3 * DMAXC; ++k) acc[k] = 0.f;
    for (long v = v0 + threadIdx.x; v < v1; v += 256) {
        if (LOGITS) {
            float mx = -3.0e38f;
            int am = 0;
            for (int c = 0; c < C; ++c) {
                const float z = pred[((long)b * C + c) * V + v];
                if (z > mx) { mx = z; am = c; }
            }
            const int lab = (int)y[(long)b * V + v];
#pragma unroll
            for (int c = 0; c < DMAXC; ++c) {
                if (c < C) {
                    const float p = c == am ? 1.f : 0.f, t = c == lab ? 1.f : 0.f;
                    acc[3 * c] += p * t; acc[3 * c + 1] += p; acc[3 * c + 2] += t;
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < DMAXC; ++c) {
                if (c < C) {
                    const float p = pred[((long)b * C + c) * V + v], t = y[((long)b * C + c) * V + v];
                    acc[3 * c] += p * t; acc[3 * c + 1] += p; acc[3 * c + 2] += t;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3 * DMAXC; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 3 * C)
        part[((long)b * gridDim.x + blockIdx.x) * (3 * C) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void __launch_bounds__(256)
dice_counts_final_kernel(const float* __restrict__ part, int BC3, int B, int C, int nchunk, double* __restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;     // index into [B][C][3]
    if (i >= BC3) return;
    const int b = i / (3 * C), k = i - b * 3 * C;
    double s = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) s += (double)part[((long)b * nchunk + ch) * (3 * C) + k];
    counts[i] = s;
}

}  // namespace

extern "C" int unetr_sw_accumulate(const float* seg, const float* importance, float* out, float* count, int C,
                                   int rz, int ry, int rx, int D, int H, int W, int z0, int y0, int x0, void* stream) {
    if (!seg || !out || !count || C <= 0 || rz <= 0 || ry <= 0 || rx <= 0) return UNETR_ERR_ARG;
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + rz > D || y0 + ry > H || x0 + rx > W) return UNETR_ERR_ARG;   // window inside the volume
    const long rv = (long)rz * ry * rx;
    const int blocks = (int)std::min<long>((rv + 255) / 256, 8192);
    hipLaunchKernelGGL(sw_accumulate_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, seg, importance, out, count, C, rz, ry, rx,
                       D, H, W, z0, y0, x0);
    return unetr_check_launch();
}

extern "C" int unetr_sw_finalize(float* out, const float* count, int B, int C, long V, void* stream) {
    if (!out || !count || B <= 0 || C <= 0 || V <= 0) return UNETR_ERR_ARG;
    const long total = (long)B * C * V;
    const int blocks = (int)std::min<long>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(sw_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, count, B, C, V);
    return unetr_check_launch();
}

extern "C" int unetr_dice_counts(const float* pred, const float* y, int B, int C, long V, int from_logits, double* counts,
                                 float* ws, size_t ws_bytes, void* stream) {
    if (!pred || !y || !counts || B <= 0 || V <= 0 || B > 65535) return UNETR_ERR_ARG;
    if (C < 1 || C > DMAXC) return UNETR_ERR_UNSUPPORTED;
    const int nchunk = cdiv(V, DVPB);
    if (!ws || (size_t)B * nchunk * 3 * C * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (from_logits) hipLaunchKernelGGL(dice_counts_kernel<true>, dim3(nchunk, B), dim3(256), 0, st, pred, y, C, V, ws);
    else hipLaunchKernelGGL(dice_counts_kernel<false>, dim3(nchunk, B), dim3(256), 0, st, pred, y, C, V, ws);
    const int n = B * C * 3;
    hipLaunchKernelGGL(dice_counts_final_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ws, n, B, C, nchunk, counts);
    return unetr_check_launch();
}
