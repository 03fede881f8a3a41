// 2x2x2 stride-2 transposed conv (fwd / dgrad / wgrad) as GEMMs with gather loaders and a pixel-shuffle epilogue.
#include "gemm_kernel.hpp"

extern "C" int unetr_tconv_fwd(const float* x, long ldx, const float* w, float* y, long ldy,
                               int B, int D, int H, int W, int Cin, int Cout, int prec,
                               float* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y) return UNETR_ERR_ARG;
    long M = (long)B * D * H * W;
    if (M > 0x7fffffffL / 8) return UNETR_ERR_ARG;
    TcGeom g{D, H, W, Cout};
    LdTcWf bl{w, 8 * Cout, Cout};
    EpTcScatter ep{y, ldy, g};
    if ((Cin % 8) == 0 && vec_ok(x, ldx, 0)) {
        LdRow al{x, ldx, 0, (int)M, 1};
        return launch_prec(prec, (int)M, 8 * Cout, Cin, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
    }
    LdRowS al{x, ldx, 0, (int)M, 0};
    return launch_prec(prec, (int)M, 8 * Cout, Cin, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int unetr_tconv_dgrad(const float* dy, long ldy, const float* w, float* dx, long ldx, int accumulate,
                                 int B, int D, int H, int W, int Cin, int Cout, int prec,
                                 float* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w || !dx) return UNETR_ERR_ARG;
    long M = (long)B * D * H * W;
    if (M > 0x7fffffffL / 8) return UNETR_ERR_ARG;
    TcGeom g{D, H, W, Cout};
    LdTcGatherA al{dy, ldy, (int)M, g};
    LdTcWd bl{w, Cin, Cout};
    EpStd ep{dx, ldx, 0, nullptr, nullptr, 0, 0, (int)M, nullptr, nullptr, 0, 0, accumulate, 1.0f};
    return launch_prec(prec, (int)M, Cin, 8 * Cout, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int unetr_tconv_wgrad(const float* x, long ldx, const float* dy, long ldy, float* dw,
                                 int B, int D, int H, int W, int Cin, int Cout, int prec,
                                 float* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw) return UNETR_ERR_ARG;
    long M = (long)B * D * H * W;
    if (M > 0x7fffffffL / 8) return UNETR_ERR_ARG;
    TcGeom g{D, H, W, Cout};
    LdCol al{x, ldx, 0, Cin, 0};
    LdTcGatherB bl{dy, ldy, 8 * Cout, g};
    EpTcWgrad ep{dw, Cout};
    return launch_prec(prec, Cin, 8 * Cout, (int)M, 1, al, bl, ep, ws, ws_bytes, (hipStream_t)stream);
}
