// Dedicated kernels for the 2x2x2 stride-2 transposed conv (nn.ConvTranspose3d(k=2, s=2, bias=False) of MONAI's
// UnetrPrUpBlock / UnetrUpBlock, /root/reference/unetr.py:99-174) at the large-volume, small-channel layers
// (Cin 16..64 -> Cout 8..32 at 24^3 / 48^3 inputs).  There the generic GEMM family (gemm_tconv.hip) spends its time in
// scalar gather loaders with per-element div/mod and in a per-element scatter epilogue (the 32->16 @ 48^3 layer:
// 135 / 230 us forward / weight gradient against a 28 us HBM floor).  k = 2, s = 2 has no overlap between taps:
//     y[b, 2z+dz, 2y+dy, 2x+dx, co] = sum_ci x[b,z,y,x,ci] * w[ci,co,dz,dy,dx]
// so per input voxel m the 8*Cout outputs are 4 contiguous runs of 2*Cout floats (dx = 0,1 adjacent in memory).
//
// Both kernels walk tiles of 64 consecutive input voxels with persistent workgroups.  Per tile, 64 threads compute the
// output-voxel base of their input voxel ONCE (the only div/mod) into an LDS table; every gather / scatter address is then
// table[v] + tap offset + channel.
//
//   tconv2_wgrad : dW[ci,co,tap] = sum_m x[m,ci] * dy[out(m,tap),co].  x and the gathered dy tile are staged voxel-major
//                  in LDS (bf16: MFMA operands with k = voxel come out through ds_read_b64_tr_b16; fp32: ds_read_b32),
//                  each wave owns a set of 16x16 tiles of the [Cin, 8*Cout] result for the whole launch, partials per
//                  workgroup + the fixed-order reduce used by the 3x3x3 weight gradient.
//   tconv2_fwd   : [64 voxels, Cin] x [Cin, 8*Cout]; the repacked weight matrix lives in LDS for the whole launch, each
//                  wave multiplies its 16 voxels against all 8*Cout columns and scatters 64-byte channel runs.
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS3 __attribute__((address_space(3)))

template <class P> struct Elem;
template <> struct Elem<PrecF32> { typedef float type; };
template <> struct Elem<PrecBF16> { typedef uint16_t type; };
template <> struct Elem<PrecBF16x3> { typedef float type; };

constexpr int TV = 64;   // input voxels per tile

// output-voxel index (in units of output voxels) of tap (0,0,0) of input voxel m
__device__ __forceinline__ int out_base(int m, int D, int H, int W) {
    const int x = m % W; int t = m / W; const int y = t % H; t /= H; const int z = t % D; const int b = t / D;
    return ((b * 2 * D + 2 * z) * 2 * H + 2 * y) * 2 * W + 2 * x;
}

// ---------------------------------------------------------------------------------------------- weight gradient
// grid (G, NG): workgroup (g, ng) accumulates result columns [ng*NC, ng*NC + NC) (column = tap*Cout + co) for all Cin rows
// over voxel tiles g, g+G, ...   RT = Cin/16 row tiles, CTW = column tiles per wave (NC = 64*CTW).
template <class P, int RT, int CTW>
__global__ void __launch_bounds__(256)
tconv2_wgrad_kernel(const typename Elem<P>::type* __restrict__ x, long ldx, const typename Elem<P>::type* __restrict__ dy, long lddy,
                    float* __restrict__ part, int M, int D, int H, int W, int Cin, int Cout, int ntiles) {
    typedef typename Elem<P>::type T;
    constexpr int ES = sizeof(T), CH = P::CH, NC = 64 * CTW;
    constexpr int PXI = RT * 16 * ES + 16, PYI = NC * ES + 16;          // image pitches (bytes), padded
    __shared__ __attribute__((aligned(16))) char lds[TV * PXI + TV * PYI + TV * 4];
    char* ximg = lds;
    char* yimg = lds + TV * PXI;
    int* tab = (int*)(lds + TV * PXI + TV * PYI);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int col0 = blockIdx.y * NC;                 // first result column of this workgroup
    const int H2 = 2 * H, W2 = 2 * W;

    f32x4 acc[RT][CTW];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // the x tile and the gathered dy tile of the NEXT voxel tile travel in registers while this tile's MFMAs run (loaded at the
    // top of every tile behind two barriers and an LDS table of output bases, each tile paid two global round trips)
    constexpr int QX = RT * 16 / CH, ITX = (TV * QX + 255) / 256;     // x: pieces of CH elements per voxel
    constexpr int QY = NC / CH, ITY = TV * QY / 256;                  // dy: pieces per voxel row of NC columns
    u32x4 xr[ITX], yr[ITY];
    auto tload = [&](int tile) {
        const int m0 = tile * TV;
#pragma unroll
        for (int j = 0; j < ITX; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / QX, q = id - v * QX;
            const bool ok = id < TV * QX && m0 + v < M;
            xr[j] = act_chunk<P>(x + (long)(ok ? m0 + v : 0) * ldx + (ok ? q * CH : 0), ok);
        }
#pragma unroll
        for (int j = 0; j < ITY; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / QY, q = id - v * QY;
            const int col = col0 + q * CH, tap = col / Cout, co = col - tap * Cout;
            const int m = m0 + v;
            const bool ok = m < M && tap < 8;
            const int ov = (ok ? out_base(m, D, H, W) : 0) + ((tap >> 2) * H2 + ((tap >> 1) & 1)) * W2 + (tap & 1);
            yr[j] = act_chunk<P>(dy + (long)(ok ? ov : 0) * lddy + (ok ? co : 0), ok);
        }
    };
    if ((int)blockIdx.x < ntiles) tload(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                  // previous tile's fragments are read
#pragma unroll
        for (int j = 0; j < ITX; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < TV * QX) {
                const int v = id / QX, q = id - v * QX;
                *(u32x4*)(ximg + v * PXI + q * 16) = xr[j];
            }
        }
#pragma unroll
        for (int j = 0; j < ITY; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / QY, q = id - v * QY;
            *(u32x4*)(yimg + v * PYI + q * 16) = yr[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) tload(tile + gridDim.x);
        // result tile (row tile i = ci block, column tile wv*CTW + j): A = x^T fragments, B = dy fragments, k = voxel
        if constexpr (CH == 8) {
            const int q = c >> 2, p = c & 3;
#pragma unroll
            for (int kb = 0; kb < TV / 32; ++kb) {
                const int v0 = kb * 32 + 8 * g + q, v1 = v0 + 4;
                u32x4 af[RT], bf[CTW];
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS3 s16x4*)(ximg + v0 * PXI + (i * 16 + 4 * p) * 2));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS3 s16x4*)(ximg + v1 * PXI + (i * 16 + 4 * p) * 2));
                    s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    af[i] = __builtin_bit_cast(u32x4, t);
                }
#pragma unroll
                for (int j = 0; j < CTW; ++j) {
                    const int cb = ((wv * CTW + j) * 16 + 4 * p) * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS3 s16x4*)(yimg + v0 * PYI + cb));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS3 s16x4*)(yimg + v1 * PYI + cb));
                    s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    bf[j] = __builtin_bit_cast(u32x4, t);
                }
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CTW; ++j) P::mma(acc[i][j], af[i], bf[j]);
            }
        } else {
#pragma unroll 2
            for (int kb = 0; kb < TV / 16; ++kb) {
                // 4-byte elements (fp32 / split words): four element reads per operand chunk (PrecF32::mma = the four K = 4 MFMAs)
                u32x4 av[RT], bv[CTW];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const int v = kb * 16 + 4 * g + tt;
#pragma unroll
                    for (int i = 0; i < RT; ++i) av[i][tt] = *(const uint32_t*)(ximg + v * PXI + (i * 16 + c) * 4);
#pragma unroll
                    for (int j = 0; j < CTW; ++j) bv[j][tt] = *(const uint32_t*)(yimg + v * PYI + ((wv * CTW + j) * 16 + c) * 4);
                }
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int j = 0; j < CTW; ++j) P::mma(acc[i][j], av[i], bv[j]);
            }
        }
    }
    // partial slab of this workgroup row: part[blockIdx.x][(ci*Cout + co)*8 + tap]; acc rows = ci, cols = (tap, co)
    const long n = (long)Cin * Cout * 8;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CTW; ++j) {
            const int col = col0 + (wv * CTW + j) * 16 + c, tap = col / Cout, co = col - tap * Cout;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int ci = i * 16 + 4 * g + rr;
                if (tap < 8 && ci < Cin) part[(long)blockIdx.x * n + ((long)ci * Cout + co) * 8 + tap] = acc[i][j][rr];
            }
        }
}

__global__ void __launch_bounds__(256)
tconv2_reduce_kernel(const float* __restrict__ part, int G, long n, float* __restrict__ dw) {
    __shared__ float sm[8][33];
    const int o = threadIdx.x & 31, ph = threadIdx.x >> 5;
    for (long i0 = (long)blockIdx.x * 32; i0 < n; i0 += (long)gridDim.x * 32) {
        const long i = i0 + o;
        float s = 0.f;
        if (i < n) {
            int gI = ph;
            for (; gI + 56 < G; gI += 64) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = part[(long)(gI + 8 * u) * n + i];
                s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            for (; gI < G; gI += 8) s += part[(long)gI * n + i];
        }
        sm[ph][o] = s;
        __syncthreads();
        if (ph == 0 && i < n) {
            float t = 0.f;
#pragma unroll
            for (int p = 0; p < 8; ++p) t += sm[p][o];
            dw[i] = t;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------ forward
// weight repack: wr[n = tap*Cout + co][k = ci] (MFMA operand type, k contiguous, rows padded to KP elements)
// KB = k-blocks of 4 chunks (64 bytes) per row; NT = 16-column tiles (= 8*Cout/16).  Each wave: 16 voxels x all columns.
template <class P, int KB, int NT>
__global__ void __launch_bounds__(256)
tconv2_fwd_kernel(const typename Elem<P>::type* __restrict__ x, long ldx, const float* __restrict__ w, typename Elem<P>::type* __restrict__ y, long ldy,
                  int M, int D, int H, int W, int Cin, int Cout, int ntiles) {
    typedef typename Elem<P>::type T;
    constexpr int CH = P::CH, RB = KB * 64;              // bytes per operand row
    constexpr int PA = RB + 16;                          // padded pitches
    __shared__ __attribute__((aligned(16))) char lds[NT * 16 * PA + TV * PA + TV * 4];
    char* wimg = lds;
    char* ximg = lds + NT * 16 * PA;
    int* tab = (int*)(lds + NT * 16 * PA + TV * PA);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int H2 = 2 * H, W2 = 2 * W;
    // weights: resident for the whole launch, packed here from torch's [Cin, Cout, 2,2,2] (row = tap*Cout + co, k = ci; the
    // separate pack launch per call cost more than these few strided L2 reads per workgroup)
    for (int id = threadIdx.x; id < NT * 16 * KB * 4; id += 256) {
        const int row = id / (KB * 4), q = id - row * (KB * 4);
        const int tap = row / Cout, co = row - tap * Cout;
        float v[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int k = q * CH + e;
            v[e] = k < Cin ? w[((long)k * Cout + co) * 8 + tap] : 0.f;
        }
        *(u32x4*)(wimg + row * PA + q * 16) = P::pack(v);
    }
    // the x tile of the NEXT tile is in flight in registers while this tile's MFMAs and stores run (one 64-voxel tile is only
    // one 16-byte piece per thread per k-block: unpipelined, every tile paid a full global-load round trip)
    constexpr int QX = KB * 4;
    constexpr int IT = (TV * QX + 255) / 256;
    u32x4 nx[IT];
    auto xload = [&](int tile) {
        const int m0 = tile * TV;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / QX, q = id - v * QX;
            const bool ok = id < TV * QX && m0 + v < M && q * CH < Cin;
            nx[j] = act_chunk<P>(x + (long)(ok ? m0 + v : 0) * ldx + (ok ? q * CH : 0), ok);
        }
    };
    if ((int)blockIdx.x < ntiles) xload(blockIdx.x);
    // the weight fragments do not depend on the tile: small shapes keep them in registers for the whole launch
    constexpr bool WREG = KB * NT <= 16;
    u32x4 breg[WREG ? KB : 1][WREG ? NT : 1];
    if constexpr (WREG) {
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < NT; ++j) breg[kb][j] = *(const u32x4*)(wimg + (j * 16 + r) * PA + kb * 64 + g * 16);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TV;
        __syncthreads();
        if (threadIdx.x < TV) {
            const int m = m0 + threadIdx.x;
            tab[threadIdx.x] = m < M ? out_base(m, D, H, W) : -1;
        }
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < TV * QX) {
                const int v = id / QX, q = id - v * QX;
                *(u32x4*)(ximg + v * PA + q * 16) = nx[j];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) xload(tile + gridDim.x);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const u32x4 a = *(const u32x4*)(ximg + (wv * 16 + r) * PA + kb * 64 + g * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 b;
                if constexpr (WREG) b = breg[kb][j]; else b = *(const u32x4*)(wimg + (j * 16 + r) * PA + kb * 64 + g * 16);
                P::mma(acc[j], b, a);        // swapped operands: the accumulator tile comes out transposed
            }
        }
        // scatter: acc[j][0..3] = (columns j*16 + 4g + 0..3, voxel wv*16 + r): four consecutive channels of one tap (Cout % 4 == 0)
        // = one 8 / 16-byte store per lane (was four 2-byte stores)
        const int ob = tab[wv * 16 + r];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = j * 16 + 4 * g, tap = col / Cout, co = col - tap * Cout;
            const int toff = ((tap >> 2) * H2 + ((tap >> 1) & 1)) * W2 + (tap & 1);
            if (ob >= 0) Io<T>::st4(y + (long)(ob + toff) * ldy + co, acc[j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ data gradient
// dx[m, ci] = sum_{tap,co} dy[out(m,tap), co] * w[ci,co,tap]: [64 voxels, K = 8*Cout] x [K, Cin].  The gathered dy tile is
// row-major in k (plain b128 fragment reads); wd[ci][k] is resident in LDS.  NT = Cin/16 column tiles per wave.
template <class P, int NT>
__global__ void __launch_bounds__(256)
tconv2_dgrad_kernel(const typename Elem<P>::type* __restrict__ dy, long lddy, const float* __restrict__ w, typename Elem<P>::type* __restrict__ dx, long ldx,
                    int M, int D, int H, int W, int Cin, int Cout, int ntiles) {
    typedef typename Elem<P>::type T;
    constexpr int ES = sizeof(T), CH = P::CH;
    extern __shared__ __attribute__((aligned(16))) char dlds[];
    const int K = 8 * Cout, RB = K * ES, PA = RB + 16, KB = RB / 64, QY = K / CH;
    char* wimg = dlds;
    char* yimg = dlds + NT * 16 * PA;
    int* tab = (int*)(yimg + TV * PA);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int H2 = 2 * H, W2 = 2 * W;
    // weights wd[ci][k = tap*Cout + co] packed here from torch's [Cin, Cout, 2,2,2] (no separate pack launch)
    for (int id = threadIdx.x; id < NT * 16 * (RB / 16); id += 256) {
        const int row = id / (RB / 16), q = id - row * (RB / 16);
        float v[CH];
#pragma unroll
        for (int e = 0; e < CH; ++e) {
            const int k = q * CH + e, tap = k / Cout, co = k - tap * Cout;
            v[e] = row < Cin ? w[((long)row * Cout + co) * 8 + tap] : 0.f;
        }
        *(u32x4*)(wimg + row * PA + q * 16) = P::pack(v);
    }
    // the gathered dy tile of the NEXT voxel tile travels in registers while this tile's MFMAs and stores run (Cout <= 32: at
    // most 8 pieces of 16 bytes per thread with bf16 storage, 16 with fp32)
    constexpr int MAXIT = CH == 8 ? 8 : 16;
    u32x4 yr[MAXIT];
    auto tload = [&](int tile) {
        const int m0 = tile * TV;
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = min(id / QY, TV - 1), q = id - (id / QY) * QY;
            const int col = q * CH, tap = col / Cout, co = col - tap * Cout;
            const int m = m0 + v;
            const bool ok = id < TV * QY && m < M;
            const int ov = (ok ? out_base(m, D, H, W) : 0) + ((tap >> 2) * H2 + ((tap >> 1) & 1)) * W2 + (tap & 1);
            yr[j] = act_chunk<P>(dy + (long)(ok ? ov : 0) * lddy + (ok ? co : 0), ok);
        }
    };
    if ((int)blockIdx.x < ntiles) tload(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TV;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXIT; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < TV * QY) {
                const int v = id / QY, q = id - v * QY;
                *(u32x4*)(yimg + v * PA + q * 16) = yr[j];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) tload(tile + gridDim.x);
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < KB; ++kb) {
            const u32x4 a = *(const u32x4*)(yimg + (wv * 16 + r) * PA + kb * 64 + g * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const u32x4 b = *(const u32x4*)(wimg + (j * 16 + r) * PA + kb * 64 + g * 16);
                P::mma(acc[j], b, a);        // swapped operands: acc[j][0..3] = (channels j*16 + 4g + 0..3, voxel wv*16 + r)
            }
        }
        {
            const int m = m0 + wv * 16 + r;
            if (m < M) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (j * 16 + 4 * g < Cin) Io<T>::st4(dx + (long)m * ldx + j * 16 + 4 * g, acc[j]);      // (Cin % 4 == 0: checked by the host)
            }
        }
    }
}

inline bool tconv2_enabled() {
    const char* e = getenv("UNETR_AMD_TCONV2");      // tuning hook: 0 = generic GEMM family only
    return !e || atoi(e) != 0;
}

template <class P, int RT, int CTW>
void launch_wgrad(int G, int NG, const typename Elem<P>::type* x, long ldx, const typename Elem<P>::type* dy, long lddy, float* ws, int M, int D, int H, int W,
                  int Cin, int Cout, int ntiles, hipStream_t st) {
    hipLaunchKernelGGL((tconv2_wgrad_kernel<P, RT, CTW>), dim3(G, NG), dim3(256), 0, st, x, ldx, dy, lddy, ws, M, D, H, W, Cin, Cout, ntiles);
}

template <class P>
int wgrad2(const typename Elem<P>::type* x, long ldx, const typename Elem<P>::type* dy, long lddy, float* dw, int B, int D, int H, int W, int Cin, int Cout,
           float* ws, size_t ws_bytes, hipStream_t st, bool parts_only = false, long* rows_only = nullptr, long* rows_used = nullptr) {
    const long Ml = (long)B * D * H * W;
    const int M = (int)Ml, ntiles = cdiv(M, TV), N = 8 * Cout, RT = Cin / 16;
    // column tiles per wave: the whole result in one workgroup column when it fits 256 columns, else several column groups
    const int CTW = N >= 256 ? 4 : (N >= 128 ? 2 : 1), NG = cdiv(N, 64 * CTW);
    int G = std::max(1, std::min(ntiles, 768 / NG));
    const long n = (long)Cin * Cout * 8;
    if (rows_only) { *rows_only = G; return UNETR_OK; }
    while (G > 1 && (size_t)G * n * sizeof(float) > ws_bytes) G >>= 1;
    if (!ws || (size_t)G * n * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    if (rows_used) *rows_used = G;
#define TC_WG(RT_, CTW_) launch_wgrad<P, RT_, CTW_>(G, NG, x, ldx, dy, lddy, ws, M, D, H, W, Cin, Cout, ntiles, st)
#define TC_WG_RT(CTW_) \
    switch (RT) { case 1: TC_WG(1, CTW_); break; case 2: TC_WG(2, CTW_); break; case 3: TC_WG(3, CTW_); break; default: TC_WG(4, CTW_); break; }
    if (CTW == 4) { TC_WG_RT(4) } else if (CTW == 2) { TC_WG_RT(2) } else { TC_WG_RT(1) }
    if (!parts_only) hipLaunchKernelGGL(tconv2_reduce_kernel, dim3((int)std::min<long>((n + 31) / 32, 4096)), dim3(256), 0, st, ws, G, n, dw);
    return unetr_check_launch();
}

template <class P, int KB, int NT>
void launch_fwd(int G, const typename Elem<P>::type* x, long ldx, const float* w, typename Elem<P>::type* y, long ldy, int M, int D, int H, int W, int Cin, int Cout,
                int ntiles, hipStream_t st) {
    hipLaunchKernelGGL((tconv2_fwd_kernel<P, KB, NT>), dim3(G), dim3(256), 0, st, x, ldx, w, y, ldy, M, D, H, W, Cin, Cout, ntiles);
}

template <class P>
int fwd2(const typename Elem<P>::type* x, long ldx, const float* w, typename Elem<P>::type* y, long ldy, int B, int D, int H, int W, int Cin, int Cout,
         float* ws, size_t ws_bytes, hipStream_t st) {
    typedef typename Elem<P>::type T;
    const int M = (int)((long)B * D * H * W), ntiles = cdiv(M, TV), N = 8 * Cout, NT = N / 16;
    const int SK = 4 * P::CH, KB = cdiv(Cin, SK), KP = KB * SK;      // k-block = 4 chunks = 64 bytes per operand row
    (void)ws; (void)ws_bytes; (void)KP;
    const int G = std::min(ntiles, 768);
#define TC_FW(KB_, NT_) launch_fwd<P, KB_, NT_>(G, x, ldx, w, y, ldy, M, D, H, W, Cin, Cout, ntiles, st)
#define TC_FW_NT(KB_) \
    switch (NT) { case 4: TC_FW(KB_, 4); break; case 8: TC_FW(KB_, 8); break; case 16: TC_FW(KB_, 16); break; default: return UNETR_ERR_UNSUPPORTED; }
    if (KB == 1) { TC_FW_NT(1) } else if (KB == 2) { TC_FW_NT(2) } else if (KB == 4) { TC_FW_NT(4) } else return UNETR_ERR_UNSUPPORTED;
    return unetr_check_launch();
}

template <class P>
int dgrad2(const typename Elem<P>::type* dy, long lddy, const float* w, typename Elem<P>::type* dx, long ldx, int B, int D, int H, int W, int Cin, int Cout,
           float* ws, size_t ws_bytes, hipStream_t st) {
    typedef typename Elem<P>::type T;
    const int M = (int)((long)B * D * H * W), ntiles = cdiv(M, TV), K = 8 * Cout, NT = cdiv(Cin, 16);
    const int RB = K * (int)sizeof(T), PA = RB + 16;
    const size_t lds = (size_t)NT * 16 * PA + (size_t)TV * PA + TV * 4;
    if (lds > 150 * 1024 || RB % 64) return UNETR_ERR_UNSUPPORTED;
    (void)ws; (void)ws_bytes;
    const int G = std::min(ntiles, lds > 64 * 1024 ? 256 : 768);
#define TC_DG(NT_)                                                                                                      \
    do {                                                                                                                \
        if (lds > 64 * 1024)                                                                                            \
            (void)hipFuncSetAttribute((const void*)tconv2_dgrad_kernel<P, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((tconv2_dgrad_kernel<P, NT_>), dim3(G), dim3(256), lds, st, dy, lddy, w, dx, ldx, M, D, H, W, \
                           Cin, Cout, ntiles);                                                                          \
    } while (0)
    switch (NT) { case 1: TC_DG(1); break; case 2: TC_DG(2); break; case 3: TC_DG(3); break; case 4: TC_DG(4); break; default: return UNETR_ERR_UNSUPPORTED; }
    return unetr_check_launch();
}

}  // namespace

// eligibility: volume large enough for the persistent walk to pay, channels in the range the register / LDS budgets were
// sized for, 16-byte aligned rows.  Anything else stays on the generic GEMM family (same results).
extern "C" int unetr_tconv2_wgrad_supported(long M, int Cin, int Cout, long ldx, long lddy) {
    return tconv2_enabled() && M >= 2048 && M < (1L << 27) && Cin % 16 == 0 && Cin >= 16 && Cin <= 64 && Cout % 8 == 0 && Cout >= 8 &&
           Cout <= 64 && ldx % 4 == 0 && lddy % 4 == 0;
}
extern "C" int unetr_tconv2_fwd_supported(long M, int Cin, int Cout, long ldx, long ldy) {
    return tconv2_enabled() && M >= 2048 && M < (1L << 27) && Cin % 8 == 0 && Cin >= 8 && Cin <= 64 && (Cout == 8 || Cout == 16 || Cout == 32) &&
           ldx % 4 == 0 && ldy % 4 == 0;
}

extern "C" int unetr_tconv2_wgrad(const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int D, int H, int W,
                                  int Cin, int Cout, int prec, float* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw) return UNETR_ERR_ARG;
    if (!unetr_tconv2_wgrad_supported((long)B * D * H * W, Cin, Cout, ldx, lddy) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15))
        return UNETR_ERR_UNSUPPORTED;
    if (prec == UNETR_PREC_BF16) {
        if ((ldx & 7) || (lddy & 7)) return UNETR_ERR_UNSUPPORTED;       // bf16 rows: 16-byte chunks
        return wgrad2<PrecBF16>((const uint16_t*)x, ldx, (const uint16_t*)dy, lddy, dw, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    }
    if (prec == UNETR_PREC_F32) return wgrad2<PrecF32>((const float*)x, ldx, (const float*)dy, lddy, dw, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return wgrad2<PrecBF16x3>((const float*)x, ldx, (const float*)dy, lddy, dw, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

// partial rows of unetr_tconv2_wgrad_parts for this shape (< 0: unsupported)
extern "C" long unetr_tconv2_wgrad_rows(int B, int D, int H, int W, int Cin, int Cout) {
    long rows = -1;
    if (!unetr_tconv2_wgrad_supported((long)B * D * H * W, Cin, Cout, Cin, Cout)) return -1;
    const int rc = wgrad2<PrecF32>(nullptr, Cin, nullptr, Cout, nullptr, B, D, H, W, Cin, Cout, nullptr, 0, nullptr, true, &rows);
    return rc == UNETR_OK ? rows : -1;
}

// unetr_tconv2_wgrad without its reduce launch: the per-workgroup partial sums stay in `part` [rows][Cin Cout 8] for
// unetr_reduce_rows_grouped
extern "C" int unetr_tconv2_wgrad_parts(const void* x, long ldx, const void* dy, long lddy, float* part, size_t part_bytes, long* rows_out,
                                        int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream) {
    if (!x || !dy || !part || !rows_out) return UNETR_ERR_ARG;
    if (!unetr_tconv2_wgrad_supported((long)B * D * H * W, Cin, Cout, ldx, lddy) || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15))
        return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (prec == UNETR_PREC_BF16) {
        if ((ldx & 7) || (lddy & 7)) return UNETR_ERR_UNSUPPORTED;
        return wgrad2<PrecBF16>((const uint16_t*)x, ldx, (const uint16_t*)dy, lddy, part, B, D, H, W, Cin, Cout, part, part_bytes, st, true, nullptr, rows_out);
    }
    if (prec == UNETR_PREC_F32) return wgrad2<PrecF32>((const float*)x, ldx, (const float*)dy, lddy, part, B, D, H, W, Cin, Cout, part, part_bytes, st, true, nullptr, rows_out);
    if (prec == UNETR_PREC_BF16X3) return wgrad2<PrecBF16x3>((const float*)x, ldx, (const float*)dy, lddy, part, B, D, H, W, Cin, Cout, part, part_bytes, st, true, nullptr, rows_out);
    return UNETR_ERR_ARG;
}

extern "C" int unetr_tconv2_fwd(const void* x, long ldx, const float* w, void* y, long ldy, int B, int D, int H, int W,
                                int Cin, int Cout, int prec, float* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y) return UNETR_ERR_ARG;
    if (!unetr_tconv2_fwd_supported((long)B * D * H * W, Cin, Cout, ldx, ldy) || ((uintptr_t)x & 15)) return UNETR_ERR_UNSUPPORTED;
    if ((uintptr_t)y & (prec == UNETR_PREC_BF16 ? 7 : 15)) return UNETR_ERR_UNSUPPORTED;      // four channels per store
    if (prec == UNETR_PREC_BF16) {
        if (ldx & 7) return UNETR_ERR_UNSUPPORTED;
        return fwd2<PrecBF16>((const uint16_t*)x, ldx, w, (uint16_t*)y, ldy, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    }
    if (prec == UNETR_PREC_F32) return fwd2<PrecF32>((const float*)x, ldx, w, (float*)y, ldy, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return fwd2<PrecBF16x3>((const float*)x, ldx, w, (float*)y, ldy, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

extern "C" int unetr_tconv2_dgrad(const void* dy, long lddy, const float* w, void* dx, long ldx, int accumulate,
                                  int B, int D, int H, int W, int Cin, int Cout, int prec, float* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w || !dx) return UNETR_ERR_ARG;
    const long M = (long)B * D * H * W;
    if (accumulate || !tconv2_enabled() || M < 2048 || M >= (1L << 27) || Cin < 8 || Cin > 64 || Cout % 8 || Cout < 8 || Cout > 32 ||
        (lddy & 3) || ((uintptr_t)dy & 15) || (Cin & 3) || (ldx & 3) || ((uintptr_t)dx & (prec == UNETR_PREC_BF16 ? 7 : 15)))
        return UNETR_ERR_UNSUPPORTED;
    if (prec == UNETR_PREC_BF16) {
        if (lddy & 7) return UNETR_ERR_UNSUPPORTED;
        return dgrad2<PrecBF16>((const uint16_t*)dy, lddy, w, (uint16_t*)dx, ldx, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    }
    if (prec == UNETR_PREC_F32) return dgrad2<PrecF32>((const float*)dy, lddy, w, (float*)dx, ldx, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return dgrad2<PrecBF16x3>((const float*)dy, lddy, w, (float*)dx, ldx, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

// ---- the SMALL transposed convs (768 channels at 6^3, 64 / 128 channels at 12^3) as plain bf16-storage GEMMs ---------------
// out[m, co*8 + tap] = sum_ci x[m, ci] * w[ci, co, tap] is x[M, Cin] . W with torch's own weight [Cin, Cout*8] read as the
// [K, N] operand of unetr_gemm_bf16 (b_kn = 1); the data gradient is dyg[M, Cout*8] . W^T with the same matrix as the
// [N, K] operand; the weight gradient x^T . dyg lands in torch's layout as it is (grouped bf16 weight-gradient launch).
// What is left of the "transposed conv" is a pixel shuffle: these two kernels move between the voxel-major output
// y[outvox(m, tap), co] (pitch ldy: possibly the first half of a concatenation buffer) and the GEMM-side matrix.
namespace {
__device__ __forceinline__ long tc_outvox(int m, int tap, int D, int H, int W) {
    int x = m % W; int t = m / W; int y = t % H; t /= H; int z = t % D; int b = t / D;
    return (((long)b * 2 * D + 2 * z + (tap >> 2)) * 2 * H + 2 * y + ((tap >> 1) & 1)) * 2 * W + 2 * x + (tap & 1);
}
// thread (m, co): reads the 8 taps of one output channel (32 contiguous bytes), writes one float into each of the 8 output rows
template <class T>
__global__ void __launch_bounds__(256)
pixel_shuffle2_kernel(const float* __restrict__ t, T* __restrict__ y, long ldy, long M, int D, int H, int W, int Cout) {
    const long total = M * Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / Cout), co = (int)(i - (long)m * Cout);
        const f32x4 a = *(const f32x4*)(t + i * 8), b = *(const f32x4*)(t + i * 8 + 4);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) Io<T>::st1(y + tc_outvox(m, tap, D, H, W) * ldy + co, v[tap]);
    }
}
// thread (m, co): gathers dy at the 8 output voxels of input voxel m, writes 8 bf16 (16 bytes) of dyg[m, co*8 .. co*8+7]
template <class T>
__global__ void __launch_bounds__(256)
pixel_unshuffle2_bf16_kernel(const T* __restrict__ dy, long lddy, uint16_t* __restrict__ g, long M, int D, int H, int W, int Cout) {
    const long total = M * Cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / Cout), co = (int)(i - (long)m * Cout);
        float v[8];
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) v[tap] = Io<T>::ld1(dy + tc_outvox(m, tap, D, H, W) * lddy + co);
        *(u32x4*)(g + i * 8) = PrecBF16::pack(v);
    }
}
}  // namespace

extern "C" int unetr_pixel_shuffle2(const float* t, void* y, long ldy, int B, int D, int H, int W, int Cout, int act16, void* stream) {
    if (!t || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || ldy < Cout) return UNETR_ERR_ARG;
    if ((uintptr_t)t & 15) return UNETR_ERR_UNSUPPORTED;
    const long M = (long)B * D * H * W;
    if (M >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;
    ACT_DISPATCH(act16, hipLaunchKernelGGL(pixel_shuffle2_kernel<AT>, dim3((unsigned)std::min<long>(cdiv(M * Cout, 256), 4096)), dim3(256), 0,
                                           (hipStream_t)stream, t, (AT*)y, ldy, M, D, H, W, Cout));
    return unetr_check_launch();
}

extern "C" int unetr_pixel_unshuffle2_bf16(const void* dy, long lddy, void* g, int B, int D, int H, int W, int Cout, int act16, void* stream) {
    if (!dy || !g || B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0 || lddy < Cout) return UNETR_ERR_ARG;
    if ((uintptr_t)g & 15) return UNETR_ERR_UNSUPPORTED;
    const long M = (long)B * D * H * W;
    if (M >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;
    ACT_DISPATCH(act16, hipLaunchKernelGGL(pixel_unshuffle2_bf16_kernel<AT>, dim3((unsigned)std::min<long>(cdiv(M * Cout, 256), 4096)), dim3(256), 0,
                                           (hipStream_t)stream, (const AT*)dy, lddy, (uint16_t*)g, M, D, H, W, Cout));
    return unetr_check_launch();
}
