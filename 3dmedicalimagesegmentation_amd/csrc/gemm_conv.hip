// 3x3x3 / 1x1x1 conv for general shapes: implicit GEMM through the MFMA GEMM family (im2col loaders).
#include "gemm_kernel.hpp"

namespace {
// ---- 3x3x3 / 1x1x1 conv through the GEMM family (im2col loaders; the general-shape path) -------------
__global__ void conv_pack_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int Cin, int Cout, int KV, int mode) {
    long total = (long)Cin * Cout * KV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int tap = (int)(i % KV); long t = i / KV; int ci = (int)(t % Cin); int co = (int)(t / Cin);
        float v = w[i];
        if (mode == 0) o[((long)co * KV + tap) * Cin + ci] = v;                 // [Cout][KV][Cin]
        else o[((long)ci * KV + (KV - 1 - tap)) * Cout + co] = v;               // [Cin][KV flipped][Cout]
    }
}

}  // namespace

extern "C" int unetr_conv_pack_weight(const float* w, float* wpack, int Cin, int Cout, int KS, int mode, void* stream) {
    if (!w || !wpack || (KS != 1 && KS != 3)) return UNETR_ERR_ARG;
    int KV = KS * KS * KS;
    long total = (long)Cin * Cout * KV;
    int blocks = (int)std::min<long>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(conv_pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wpack, Cin, Cout, KV, mode);
    return unetr_check_launch();
}

extern "C" int unetr_conv_gemm_fwd(const float* x, long ldx, const float* wpack, float* y, long ldy, int accumulate,
                                   int B, int D, int H, int W, int Cin, int Cout, int KS, int prec,
                                   float* ws, size_t ws_bytes, void* stream) {
    if (!x || !wpack || !y || (KS != 1 && KS != 3)) return UNETR_ERR_ARG;
    long M = (long)B * D * H * W;
    if (M > 0x7fffffffL) return UNETR_ERR_ARG;
    int KV = KS * KS * KS, K = KV * Cin;
    ConvGeom g{D, H, W, Cin, KS};
    LdIm2colA al{x, ldx, (int)M, g};
    EpStd ep{y, ldy, 0, nullptr, nullptr, 0, 0, (int)M, nullptr, nullptr, 0, 0, accumulate, 1.0f};
    if ((K % 8) == 0 && vec_ok(wpack, K, 0)) {
        LdRow bl{wpack, K, 0, Cout, 1};
        return launch_prec(prec, (int)M, Cout, K, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
    }
    LdRowS bl{wpack, K, 0, Cout, 0};
    return launch_prec(prec, (int)M, Cout, K, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int unetr_conv_gemm_wgrad(const float* x, long ldx, const float* dy, long ldy, float* dw,
                                     int B, int D, int H, int W, int Cin, int Cout, int KS, int prec,
                                     float* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw || (KS != 1 && KS != 3)) return UNETR_ERR_ARG;
    long M = (long)B * D * H * W;
    if (M > 0x7fffffffL) return UNETR_ERR_ARG;
    int KV = KS * KS * KS;
    ConvGeom g{D, H, W, Cin, KS};
    LdCol al{dy, ldy, 0, Cout, 0};
    LdIm2colB bl{x, ldx, KV * Cin, g};
    EpConvWgrad ep{dw, Cin, KV};
    return launch_prec(prec, Cout, KV * Cin, (int)M, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
}
