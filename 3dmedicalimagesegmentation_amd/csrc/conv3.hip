// 3x3x3 convolution (pad 1, stride 1, no bias) on channels-last feature maps for gfx950, three kernels:
//
//   conv3_fwd_kernel    y[v, co] = sum_{tap, ci} x[v + off(tap), ci] * w[co, ci, tap]
//                       (also the data gradient: same kernel on dy with flipped / transposed packed weights)
//   conv3_wgrad_kernel  dw[co, ci, tap] = sum_v dy[v, co] * x[v + off(tap), ci]
//   pack / reduce helpers
//
// Forward: a workgroup owns a 4x4x16 (z,y,x) output tile.  The 6x6x18 input halo of one channel slab
// (one MFMA k-block: 32 bf16 / 16 f32 channels) is staged ONCE in LDS (fp32 in HBM -> MFMA operand type),
// and all 27 taps read their A fragments straight out of that window with ds_read_b128 at a per-lane
// voxel offset -- no im2col buffer, each input byte crosses the HBM/L2 -> LDS path once per tile.
// B fragments (weights, pre-packed [tap][slab][co][k]) are 16-byte loads from L1/L2, double-buffered
// across taps.  Contraction on 16x16 MFMA tiles (rows = 16 consecutive x voxels, cols = 16 out channels).
//
// Weight gradient: the reduction index is the voxel, so both MFMA operands need "k = voxel" fragments of
// channel-contiguous LDS images.  In bf16 mode that is exactly what ds_read_b64_tr_b16 delivers (each lane
// supplies the address of one row = one voxel, so the 27 shifted windows cost nothing); in fp32 mode one
// ds_read_b32 per element does the same.  Accumulators for (tap, ci-tile) pairs stay in registers while the
// workgroup walks many voxel tiles; per-workgroup partial sums are reduced in a fixed order.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS __attribute__((address_space(3)))

constexpr int TZ = 4, TY = 4, TX = 16;          // output tile
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
constexpr int HZ = 6, HY = 6, HX = 18;          // halo
constexpr int NHALO = HZ * HY * HX;             // 648
constexpr int NVOX = TZ * TY * TX;              // 256

// LDS window layouts (tools/lds_conflicts.py models the bank conflicts of every fragment read; round 1's padded pitches --
// 48 / 80 bytes per voxel -- cost exactly 2x the conflict-free cycles on all of them, which is what the PMC profile showed:
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE):
//   LAY 0  plain rows, no padding.  Pair-mode forward window (32 B per voxel): the 16 lanes of a ds_read_b128 group read
//          16 consecutive 16-byte slots -- conflict-free as it is.
//   LAY 1  64-byte voxels (one k-block of channels, forward slab mode): 16-byte chunk c of the voxel at window column hx is
//          stored in chunk slot c ^ (2 * ((hx >> 2) & 1)).
//   LAY 2  32-byte voxels read by ds_read_b64_tr_b16 (weight gradient, 16-channel slab): the voxel at column hx is stored at
//          column hx ^ ((hx & 8) >> 1), i.e. the two halves of columns 8..15 swap, so the 8 voxels a 32-lane half
//          addresses (x .. x+3 and x+8 .. x+11) cover all 64 banks.
constexpr int FPITCH = 64;  // one k-block of channels per halo voxel (LAY 1)
__device__ __forceinline__ int lay_flip(int h) { return h ^ ((h & 8) >> 1); }
template <int LAY>
__device__ __forceinline__ int lay_off(int hv, int ch, int pitch) {
    if constexpr (LAY == 0) return hv * pitch + ch * 16;
    const int hx = hv % HX;
    if constexpr (LAY == 1) return hv * pitch + ((ch ^ (((hx >> 2) & 1) << 1)) << 4);
    return (hv - hx + lay_flip(hx)) * pitch + ch * 16;
}

template <class P> struct ElemOf;
template <> struct ElemOf<PrecF32> { typedef float type; };
template <> struct ElemOf<PrecBF16> { typedef uint16_t type; };
template <> struct ElemOf<PrecBF16x3> { typedef uint32_t type; };      // packed weights: one split word [hi | lo << 16] per element

__device__ __forceinline__ uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
template <class T> __device__ __forceinline__ T cvt_elem(float f);
template <> __device__ __forceinline__ float cvt_elem<float>(float f) { return f; }
template <> __device__ __forceinline__ uint16_t cvt_elem<uint16_t>(float f) { return f2bf(f); }
template <> __device__ __forceinline__ uint32_t cvt_elem<uint32_t>(float f) { return PrecBF16x3::split(f); }

// ------------------------------------------------------------------------------------- weight packing
// fwd  (mode 0): wp[tap][slab][n = co][k = ci % SL]           source w[co][ci][tap]
// dgrad(mode 1): wp[tap'][slab][n = ci][k = co % SL]          source w[co][ci][26 - tap']   (K = Cout, N = Cin)
template <class T>
__global__ void conv3_pack_kernel(const float* __restrict__ w, T* __restrict__ wp, int Cin, int Cout, int mode, int SL) {
    const int K = mode ? Cout : Cin, N = mode ? Cin : Cout;
    const int nslab = (K + SL - 1) / SL;
    const long total = 27L * nslab * N * SL;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int kk = (int)(i % SL); long t = i / SL; int n = (int)(t % N); t /= N; int slab = (int)(t % nslab); int tap = (int)(t / nslab);
        int k = slab * SL + kk;
        float v = 0.f;
        if (k < K) v = mode ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
        wp[i] = cvt_elem<T>(v);
    }
}

// ------------------------------------------------------------------------------------------- forward
// Stage the 6x6x18 halo of one channel slab (NCH 16-byte chunks per voxel) into LDS.  Loads are issued in
// batches of SB chunks per thread with no dependent instruction in between, so a batch costs ONE memory
// round trip instead of SB (the first version looped load -> convert -> ds_write per chunk and ran the
// 96^3 convs at 1.1 TB/s).  vec: 16-byte loads (aligned, Cin % 4 == 0); otherwise predicated scalar loads.
// XM = how the input tensor is stored: 0 fp32, predicated scalar loads (single-channel image); 1 fp32, 16-byte loads of four
// channels; 2 bf16 (bf16 precision mode: feature maps are stored as bf16), one 16-byte load = the 8 channels of a piece.
template <class P, int NCH, int XM, int LAY = 0>
__device__ __forceinline__ void stage_halo(const void* __restrict__ xv, long ldx, int b, int z0, int y0, int x0, int D, int H, int W,
                                           int c0, int Cin, int pitch, char* halo) {
    constexpr int CH = P::CH, TOTAL = NHALO * NCH, ITERS = (TOTAL + 255) / 256, SB = 6, NQ = CH / 4;
    constexpr bool VEC = XM == 1;
    if constexpr (XM == 2) {
        const uint16_t* __restrict__ xh = (const uint16_t*)xv;
        for (int it0 = 0; it0 < ITERS; it0 += SB) {
            u32x4 buf[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int id = threadIdx.x + (it0 + j) * 256;
                const int hv = id / NCH, ch = id - hv * NCH;
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                const int c = c0 + ch * CH;
                const bool ok = (it0 + j < ITERS) && id < TOTAL && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                                (unsigned)gx < (unsigned)W && c < Cin;
                buf[j] = act_chunk<P>((const typename ActOf<P>::type*)(ok ? xh + ((((long)b * D + gz) * H + gy) * W + gx) * ldx + c : xh), ok);
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int id = threadIdx.x + (it0 + j) * 256;
                if (it0 + j < ITERS && id < TOTAL) {
                    const int hv = id / NCH, ch = id - hv * NCH;
                    *(u32x4*)(halo + lay_off<LAY>(hv, ch, pitch)) = buf[j];
                }
            }
        }
        return;
    }
    const float* __restrict__ x = (const float*)xv;
    for (int it0 = 0; it0 < ITERS; it0 += SB) {
        f32x4 buf[SB][NQ];
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int id = threadIdx.x + (it0 + j) * 256;
            const int hv = id / NCH, ch = id - hv * NCH;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const int c = c0 + ch * CH;
            const bool ok = (it0 + j < ITERS) && id < TOTAL && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                            (unsigned)gx < (unsigned)W && c < Cin;
            // branch-free: always load (from the tensor base when out of range), then select -- a predicated
            // load compiles to a branch + s_waitcnt vmcnt(0) and serialises the whole batch
            const float* q = ok ? x + ((((long)b * D + gz) * H + gy) * W + gx) * ldx + c : x;
#pragma unroll
            for (int c4 = 0; c4 < NQ; ++c4) {
                if constexpr (VEC) {
                    const bool okc = ok && c + 4 * c4 + 4 <= Cin;
                    f32x4 t = *(const f32x4*)(okc ? q + 4 * c4 : x);
                    buf[j][c4] = okc ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool oke = ok && c + 4 * c4 + e < Cin;
                        float t = *(oke ? q + 4 * c4 + e : x);
                        buf[j][c4][e] = oke ? t : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const int id = threadIdx.x + (it0 + j) * 256;
            if (it0 + j < ITERS && id < TOTAL) {
                const int hv = id / NCH, ch = id - hv * NCH;
                float v[CH];
#pragma unroll
                for (int c4 = 0; c4 < NQ; ++c4) { v[4 * c4] = buf[j][c4][0]; v[4 * c4 + 1] = buf[j][c4][1]; v[4 * c4 + 2] = buf[j][c4][2]; v[4 * c4 + 3] = buf[j][c4][3]; }
                *(u32x4*)(halo + lay_off<LAY>(hv, ch, pitch)) = P::pack(v);
            }
        }
    }
}


// Tile id -> tile coordinates.  Workgroups are dealt round-robin to the 8 XCDs (ids b and b+8 share an L2), so
// each XCD gets a CONTIGUOUS run of a locality order (x fastest, then 4 y-tiles, then z, then y-blocks, then
// batch): halos shared by neighbouring tiles are then served by that XCD's L2 instead of being re-fetched
// through the fabric by every XCD (PMC: FETCH_SIZE was 2.8x the algorithmic input with the plain order).
// Pure performance mapping: any placement gives the same result.
__device__ __forceinline__ void tile_coords(int id, int total, int ntx, int nty, int ntz, int& tx, int& ty, int& tz, int& b) {
    const int q = total >> 3, r = total & 7, xcd = id & 7, local = id >> 3;
    int s = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    const int YB = (nty & 3) == 0 ? 4 : 1;
    tx = s % ntx; s /= ntx;
    const int yi = s % YB; s /= YB;
    tz = s % ntz; s /= ntz;
    const int nyb = nty / YB;
    const int yb = s % nyb;
    b = s / nyb;
    ty = yb * YB + yi;
}

// The tiles a persistent workgroup walks (blockIdx.x + k * gridDim.x, k = 0, 1, ...) with their coordinates computed 64 at a
// time, one per lane (tx | ty << 8 | tz << 16 | b << 24), and fetched per tile with one v_readlane: tile_coords is five
// integer divisions on the scalar unit -- ~100 SALU instructions, twice per tile (current + prefetched), on a unit the waves
// of a CU share.  (The host rejects shapes whose tile counts / batch exceed 255.)
struct TileTable {
    unsigned packed;
    __device__ __forceinline__ void fill(int k0, int ntiles, int ntx, int nty, int ntz) {
        const int t = blockIdx.x + (k0 + (int)(threadIdx.x & 63)) * gridDim.x;
        int tx = 0, ty = 0, tz = 0, b = 0;
        if (t < ntiles) tile_coords(t, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        packed = (unsigned)tx | ((unsigned)ty << 8) | ((unsigned)tz << 16) | ((unsigned)b << 24);
    }
    // coordinates of the k-th tile of this workgroup; refills the table when k enters a new group of 64
    __device__ __forceinline__ void get(int k, int ntiles, int ntx, int nty, int ntz, int& tx, int& ty, int& tz, int& b) {
        if ((k & 63) == 0) fill(k, ntiles, ntx, nty, ntz);
        const unsigned pk = __builtin_amdgcn_readlane(packed, k & 63);
        tx = pk & 255u; ty = (pk >> 8) & 255u; tz = (pk >> 16) & 255u; b = pk >> 24;
    }
};

template <class P, int NTB, int XM>
__global__ void __launch_bounds__(256)
conv3_fwd_kernel(const void* __restrict__ x, long ldx, const char* __restrict__ wp, typename ActOf<P>::type* __restrict__ y, long ldy, int accumulate,
                 int D, int H, int W, int Cin, int Cout, int ntx, int nty, int ntz) {
    typedef typename ActOf<P>::type YT;
    constexpr int CH = P::CH, SL = 4 * CH;
    __shared__ __attribute__((aligned(16))) char halo[NHALO * FPITCH];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    int tx, ty, tz, b;
    tile_coords(blockIdx.x, gridDim.x, ntx, nty, ntz, tx, ty, tz, b);
    const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
    const int nt0 = blockIdx.y * NTB;
    const int nslab = (Cin + SL - 1) / SL;
    f32x4 acc[4][NTB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int slab = 0; slab < nslab; ++slab) {
        __syncthreads();
        stage_halo<P, 4, XM, 1>(x, ldx, b, z0, y0, x0, D, H, W, slab * SL, Cin, FPITCH, halo);
        __syncthreads();
        // B fragment of (tap, slab, n-tile j): 16 bytes at wp[((tap*nslab+slab)*Cout + n)*64 + g*16].
        // Fragments are prefetched one GROUP of GT taps ahead (L1/L2 latency ~ a few hundred cycles must hide
        // behind GT*4*NTB MFMAs).
        constexpr int GT = NTB == 1 ? 9 : (NTB == 2 ? 3 : 1), NG = 27 / GT;
        const char* wbase = wp + ((long)slab * Cout + nt0 * 16 + r) * 64 + g * 16;
        const long wtap = (long)nslab * Cout * 64;
        u32x4 bcur[GT][NTB], bnxt[GT][NTB];
#pragma unroll
        for (int t = 0; t < GT; ++t)
#pragma unroll
            for (int j = 0; j < NTB; ++j) bcur[t][j] = *(const u32x4*)(wbase + t * wtap + j * 16 * 64);
        for (int tg = 0; tg < NG; ++tg) {
            if (tg + 1 < NG) {
#pragma unroll
                for (int t = 0; t < GT; ++t)
#pragma unroll
                    for (int j = 0; j < NTB; ++j) bnxt[t][j] = *(const u32x4*)(wbase + ((tg + 1) * GT + t) * wtap + j * 16 * 64);
            }
#pragma unroll
            for (int t = 0; t < GT; ++t) {
                const int tap = tg * GT + t;
                const int dz = tap / 9, rem = tap - dz * 9, dy = rem / 3, dx = rem - dy * 3;
                const char* hbase = halo + (((wv + dz) * HY + dy) * HX) * FPITCH + lay_off<1>(r + dx, g, FPITCH);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32x4 a = *(const u32x4*)(hbase + i * HX * FPITCH);
#pragma unroll
                    for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], a, bcur[t][j]);
                }
            }
#pragma unroll
            for (int t = 0; t < GT; ++t)
#pragma unroll
                for (int j = 0; j < NTB; ++j) bcur[t][j] = bnxt[t][j];
        }
    }
    const int zo = z0 + wv;
    if (zo < D) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yo = y0 + i;
            if (yo >= H) continue;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int xo = x0 + 4 * g + rr;
                if (xo >= W) continue;
                YT* yp = y + ((((long)b * D + zo) * H + yo) * W + xo) * ldy + nt0 * 16 + r;
                if (accumulate) {
                    float old[NTB];
#pragma unroll
                    for (int j = 0; j < NTB; ++j) old[j] = Io<YT>::ld1(yp + j * 16);
#pragma unroll
                    for (int j = 0; j < NTB; ++j) Io<YT>::st1(yp + j * 16, acc[i][j][rr] + old[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < NTB; ++j) Io<YT>::st1(yp + j * 16, acc[i][j][rr]);
                }
            }
        }
    }
}

// ----------------------------------------------------------------- forward, persistent + pipelined
// The first kernel above is latency-bound per tile (PMC: waves wait 64 % of the time; each workgroup does
// load -> wait -> convert -> MFMA -> store for ONE tile).  This version keeps workgroups resident and walks
// tiles: while the 27 taps of tile t run out of LDS, the global loads of tile t+1's halo are already in
// flight into registers (HaloRegs), and are converted + written to the single LDS window right after the
// barrier that ends tile t.
//
// PAIR mode (bf16, <= 16 input channels -- the four largest convolutions of the network): one K=32 MFMA
// contracts TWO taps x 16 channels (lane groups 0,1 -> tap 2p, groups 2,3 -> tap 2p+1) instead of one tap
// zero-padded to 32 channels: 14 MFMAs + 14 ds_read_b128 per 16-voxel row instead of 27, a 31 KB halo
// window (pitch 48 B) instead of 52 KB, and all 14 weight fragments live in registers for the whole launch.
template <class P, int NCH>
struct HaloRegs {
    static constexpr int ITERS = (NHALO * NCH + 255) / 256;
    f32x4 v[ITERS][P::CH / 4];
    unsigned okbits;          // VEC path: bit (j * CH/4 + c4) = this 16-byte piece is inside the volume / channel range
};

// Window loads into registers.  VEC path: the element offset inside the batch item is 32-bit (the host rejects feature
// maps of >= 2^31 elements per item) on top of a wave-uniform 64-bit base; out-of-window pieces read element 0 of the item
// and are zeroed in halo_store from the okbits mask.  Nothing here depends on the loaded values, so the loads stay in
// flight across the MFMA phase that follows (a select / multiply right after the load is either turned back into a
// conditional load behind an exec-mask branch with 64-bit address arithmetic, or waits for the data before the MFMAs).
template <class P, int NCH, int XM>
__device__ __forceinline__ void halo_load(HaloRegs<P, NCH>& R, const void* __restrict__ xv, long ldx, int b, int z0, int y0, int x0,
                                          int D, int H, int W, int c0, int Cin) {
    constexpr int CH = P::CH, TOTAL = NHALO * NCH, NQ = CH / 4;
    constexpr bool VEC = XM == 1;
    static_assert(HaloRegs<P, NCH>::ITERS * NQ <= 32, "okbits is one 32-bit mask");
    if constexpr (XM == 2) {
        // bf16-stored input: the 8 channels of a piece are ONE 16-byte load, already in MFMA operand form; it travels in
        // v[j][0] (bit pattern) and halo_store writes it to the window as it is (both okbits of the piece carry its mask)
        const uint16_t* __restrict__ xh = (const uint16_t*)xv + (long)b * D * H * W * ldx;
        const int ld32 = (int)ldx;
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < HaloRegs<P, NCH>::ITERS; ++j) {
            const int id = threadIdx.x + j * 256;
            const int hv = id / NCH, ch = id - hv * NCH;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const int c = c0 + ch * CH;
            const bool ok = id < TOTAL && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin;
            const int off = ok ? ((gz * H + gy) * W + gx) * ld32 + c : 0;
            R.v[j][0] = __builtin_bit_cast(f32x4, *(const u32x4*)(xh + off));
            bits |= ok ? (((1u << NQ) - 1u) << (j * NQ)) : 0u;
        }
        R.okbits = bits;
        return;
    }
    const float* __restrict__ x = (const float*)xv;
    const float* __restrict__ xb = x + (long)b * D * H * W * ldx;
    const int ld32 = (int)ldx;
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < HaloRegs<P, NCH>::ITERS; ++j) {
        const int id = threadIdx.x + j * 256;
        const int hv = id / NCH, ch = id - hv * NCH;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const int c = c0 + ch * CH;
        const bool ok = id < TOTAL && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin;
        if constexpr (VEC) {
            const int off = ok ? ((gz * H + gy) * W + gx) * ld32 + c : 0;
#pragma unroll
            for (int c4 = 0; c4 < NQ; ++c4) {
                const bool okc = ok && c + 4 * c4 + 4 <= Cin;
                R.v[j][c4] = *(const f32x4*)(xb + (off + (okc ? 4 * c4 : 0)));
                bits |= okc ? (1u << (j * NQ + c4)) : 0u;
            }
        } else {
            const float* q = ok ? xb + (((long)gz * H + gy) * W + gx) * ldx + c : xb;
#pragma unroll
            for (int c4 = 0; c4 < NQ; ++c4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // predicated scalar load (an exec-masked load, no wait): with a single input channel 15 of the 16
                    // element slots of a voxel are padding, and loading a clamped address for each of them cost more
                    // than the real data
                    float t = 0.f;
                    if (ok && c + 4 * c4 + e < Cin) t = q[4 * c4 + e];
                    R.v[j][c4][e] = t;
                }
            }
            bits = 0xffffffffu;
        }
    }
    R.okbits = bits;
}

template <class P, int NCH, int XM = 1, int LAY = 0>
__device__ __forceinline__ void halo_store(const HaloRegs<P, NCH>& R, int pitch, char* halo) {
    constexpr int CH = P::CH, TOTAL = NHALO * NCH, NQ = CH / 4;
#pragma unroll
    for (int j = 0; j < HaloRegs<P, NCH>::ITERS; ++j) {
        const int id = threadIdx.x + j * 256;
        if (id < TOTAL) {
            const int hv = id / NCH, ch = id - hv * NCH;
            u32x4 w;
            if constexpr (XM == 2) {
                w = __builtin_bit_cast(u32x4, R.v[j][0]);
            } else {
                float v[CH];
#pragma unroll
                for (int c4 = 0; c4 < NQ; ++c4) { v[4 * c4] = R.v[j][c4][0]; v[4 * c4 + 1] = R.v[j][c4][1]; v[4 * c4 + 2] = R.v[j][c4][2]; v[4 * c4 + 3] = R.v[j][c4][3]; }
                w = P::pack(v);
            }
            // zero the out-of-window pieces on the PACKED words (dword d holds elements of piece d * NQ / 4)
#pragma unroll
            for (int d = 0; d < 4; ++d) w[d] = ((R.okbits >> (j * NQ + d * NQ / 4)) & 1u) ? w[d] : 0u;
            *(u32x4*)(halo + lay_off<LAY>(hv, ch, pitch)) = w;
        }
    }
}

// ---- the tile-invariant part of the window staging (bf16-stored input, XM == 2), computed ONCE per thread: which window piece
// a thread loads in iteration j, its element offset relative to the window origin, and where it goes in LDS.  Per tile only a
// wave-uniform base offset is added; tiles whose window lies inside the volume (wave-uniform test) skip the per-piece bounds
// checks.  Before, every tile re-derived all of it per piece (div / mod chains, three range checks, 64-bit multiplies): with
// the MFMA loop, the loads and the stores removed the kernel still took 24 of its 52 us at 96^3 x 16 channels.
template <int NCH>
struct HaloPlan {
    static constexpr int ITERS = (NHALO * NCH + 255) / 256;
    int rel[ITERS];           // element offset from the window origin (voxel (z0-1, y0-1, x0-1), channel c0)
    int lo[ITERS];            // byte offset in the LDS window
    unsigned valid;           // okbits of a fully interior window
};
template <class P, int NCH, int LAY>
__device__ __forceinline__ void halo_plan(HaloPlan<NCH>& pl, int H, int W, int ld32, int pitch) {
    constexpr int CH = P::CH, TOTAL = NHALO * NCH, NQ = CH / 4;
    pl.valid = 0;
#pragma unroll
    for (int j = 0; j < HaloPlan<NCH>::ITERS; ++j) {
        const int id = threadIdx.x + j * 256;
        const bool v = id < TOTAL;
        const int hv = v ? id / NCH : 0, ch = v ? id - hv * NCH : 0;
        const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
        pl.rel[j] = ((hz * H + hy) * W + hx) * ld32 + ch * CH;
        pl.lo[j] = lay_off<LAY>(hv, ch, pitch);
        pl.valid |= v ? (((1u << NQ) - 1u) << (j * NQ)) : 0u;
    }
}
template <class P, int NCH>
__device__ __forceinline__ void halo_load_planned(HaloRegs<P, NCH>& R, const HaloPlan<NCH>& pl, const uint16_t* __restrict__ xh,
                                                  int ld32, int z0, int y0, int x0, int D, int H, int W, int c0, int Cin) {
    constexpr int CH = P::CH, TOTAL = NHALO * NCH, NQ = CH / 4;
    const int base = (((z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * ld32 + c0;          // wave-uniform (negative on some borders)
    const bool interior = z0 >= 1 && z0 - 1 + HZ <= D && y0 >= 1 && y0 - 1 + HY <= H && x0 >= 1 && x0 - 1 + HX <= W && c0 + NCH * CH <= Cin;
    if (interior) {
#pragma unroll
        for (int j = 0; j < HaloPlan<NCH>::ITERS; ++j) {
            const bool v = (j + 1) * 256 <= TOTAL || (int)threadIdx.x + j * 256 < TOTAL;
            R.v[j][0] = __builtin_bit_cast(f32x4, *(const u32x4*)(xh + (v ? base + pl.rel[j] : 0)));
        }
        R.okbits = pl.valid;
    } else {
        unsigned bits = 0;
#pragma unroll
        for (int j = 0; j < HaloPlan<NCH>::ITERS; ++j) {
            // (border tiles only: the piece's window coordinates are re-derived -- divisions by compile-time constants)
            const int id = threadIdx.x + j * 256;
            const int hv = id / NCH, ch = id - hv * NCH;
            const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const int c = c0 + ch * CH;
            const bool ok = id < TOTAL && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && c < Cin;
            R.v[j][0] = __builtin_bit_cast(f32x4, *(const u32x4*)(xh + (ok ? base + pl.rel[j] : 0)));
            bits |= ok ? (((1u << NQ) - 1u) << (j * NQ)) : 0u;
        }
        R.okbits = bits;
    }
}
template <class P, int NCH>
__device__ __forceinline__ void halo_store_planned(const HaloRegs<P, NCH>& R, const HaloPlan<NCH>& pl, char* halo) {
    constexpr int TOTAL = NHALO * NCH, NQ = P::CH / 4;
#pragma unroll
    for (int j = 0; j < HaloPlan<NCH>::ITERS; ++j) {
        if ((j + 1) * 256 <= TOTAL || (int)threadIdx.x + j * 256 < TOTAL) {
            u32x4 w = __builtin_bit_cast(u32x4, R.v[j][0]);
            const bool ok = (R.okbits >> (j * NQ)) & 1u;            // (bf16: both bits of a piece carry the same mask)
            w = ok ? w : (u32x4){0u, 0u, 0u, 0u};
            *(u32x4*)(halo + pl.lo[j]) = w;
        }
    }
}

// pair-mode weights: wp[pair][n][k], k = (tap - 2*pair)*16 + ci   (bf16; zero for tap 27 and ci >= K)
__global__ void conv3_pack_pair_kernel(const float* __restrict__ w, uint16_t* __restrict__ wp, int Cin, int Cout, int mode) {
    const int K = mode ? Cout : Cin, N = mode ? Cin : Cout;
    const long total = 14L * N * 32;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int kk = (int)(i & 31); long t = i >> 5; int n = (int)(t % N); int tp = (int)(t / N);
        int tap = 2 * tp + (kk >> 4), k = kk & 15;
        float v = 0.f;
        if (tap < 27 && k < K) v = mode ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
        wp[i] = f2bf(v);
    }
}

// InstanceNorm partial sums inside the persistent conv kernel: every lane keeps running (sum, sum of squares) of its
// valid output voxels per channel tile across the tiles its workgroup walks; a wave's tiles come in non-decreasing batch
// order (tile_coords), so the sums are flushed -- reduced over the four lane groups, written by lane group 0 to row
// (b, workgroup, wave) of part[B][rows][2][Cout] -- when the batch index changes and at the end of the walk.
// (The persistent kernel runs its MFMAs with swapped operands -- weights as A, window as B -- so the accumulator tile is
// the transposed one: lane (r = lane & 15, g = lane >> 4) holds the FOUR consecutive channels 4g..4g+3 of output voxel
// column r.  A row of 16 voxels is then stored as one 512-byte (bf16) / 1-KiB (fp32) contiguous run by ONE store instruction
// of 8 / 16 bytes per lane, instead of sixteen 2-byte-per-lane scatters.)
template <int NTB>
__device__ __forceinline__ void stats_add(const f32x4 (&acc)[4][NTB], const bool (&okv)[4], f32x4 (&s1)[NTB], f32x4 (&s2)[NTB]) {
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = okv[i] ? acc[i][j] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s1[j] += v; s2[j] += v * v;
        }
}
// One partial row per WORKGROUP: the four waves' sums meet in LDS (fixed order: wave 0 + 1 + 2 + 3) -- the consumers of these
// rows re-reduce them in their own prologue (norm_misc.hip: in_fin_*), so the row count is what every consumer block pays for.
// Every wave of a workgroup walks the same tiles, so a flush (batch item changes / end of the walk) is workgroup-uniform and
// may hold barriers.  sred: NTB * 128 floats of LDS.
template <int NTB>
__device__ __forceinline__ void stats_flush(f32x4 (&s1)[NTB], f32x4 (&s2)[NTB], float* __restrict__ prow, int Cout, int n0, int r, int g, int wv,
                                            float* sred) {
    // sum over the 16 voxel columns (lanes with equal g), lane r == 0 of each group holds its four channels
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
        f32x4 a = s1[j], q = s2[j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += __shfl_xor(a[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
        if (r == 0) { *(f32x4*)(sred + (wv * 2 + 0) * NTB * 16 + j * 16 + 4 * g) = a; *(f32x4*)(sred + (wv * 2 + 1) * NTB * 16 + j * 16 + 4 * g) = q; }
        s1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[j] = s1[j];
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * NTB * 16) {
        const int t = threadIdx.x, which = t / (NTB * 16), ch = t - which * (NTB * 16);
        const float v = ((sred[(0 * 2 + which) * NTB * 16 + ch] + sred[(1 * 2 + which) * NTB * 16 + ch]) + sred[(2 * 2 + which) * NTB * 16 + ch]) +
                        sred[(3 * 2 + which) * NTB * 16 + ch];
        prow[which * Cout + n0 + ch] = v;
    }
    __syncthreads();
}
// this workgroup's row of every batch item starts at zero (a workgroup only flushes the batch items its tiles touch)
template <int NTB>
__device__ __forceinline__ void stats_zero_rows(float* __restrict__ part, int nb, long srows, int Cout, int n0) {
    if ((int)threadIdx.x < 2 * NTB * 16) {
        const int t = threadIdx.x, which = t / (NTB * 16), ch = t - which * (NTB * 16);
        for (int bb = 0; bb < nb; ++bb) part[(((long)bb * srows) + blockIdx.x) * 2 * Cout + which * Cout + n0 + ch] = 0.f;
    }
}

// FUSE >= 1: besides y = conv3x3x3(x) the kernel emits part = InstanceNorm partial sums of y (the separate statistics
//   pass over y disappears; the partial buffer must be zero-filled before the launch);
// FUSE == 2: also y3 = conv1x1x1(x) with weights wp3 and its partial sums part3: MONAI's UnetResBlock.conv3 reads the same
//   input as conv1, i.e. the centre tap of the window already staged in LDS (one extra MFMA per 16 voxels);
// FUSE == 3: the same for a single-slab window (Cin <= 32 bf16 / 16 fp32, not PAIR): the 1x1x1 product is formed AFTER the
//   tile's 3x3x3 result has been stored, re-reading the centre tap from the still-resident window, so the two accumulator
//   sets are never live together and the kernel keeps two waves per SIMD (FUSE == 2 on this path needs 364 registers).
// two resident waves per SIMD (256 registers) is requested only where the variant fits without scratch
// FUSE == 4 (data-gradient side of the same block): y += x3 . w3 with a SECOND input x3 ([voxels, K3], e.g. the gradient of
//   the 1x1x1 branch) multiplied by a 1x1x1 weight matrix in the tile epilogue: dx = conv3x3x3^T(dc1) + conv1x1x1^T(dc3) in
//   one pass, instead of a GEMM writing dx followed by an accumulating conv that re-reads it.  x3 fragments are loaded
//   straight from global memory in MFMA operand shape (16 voxels x one 16-byte chunk per lane group).
// FUSE == 5 (data gradient in front of an InstanceNorm's backward): besides y = conv(x) the kernel emits the two per-channel
//   sums the InstanceNorm backward of the tensor this gradient belongs to needs -- sum(g) and sum(g * n) with n = the normalised
//   forward value a * xn + o (xn = y3: the pre-norm tensor, same shape / pitch as y; a, o from the statistics wp3 = [B][Cout][2]
//   (mean, rstd)) and g = y * lrelu'(n) -- as partial rows `part`, so the separate reduction pass over (y, xn) disappears.
// WL (slab mode only) = number of slab-weight images kept in LDS, filled by LDS-DMA (27 x NTB KB each: lane (r, g) of the
// (tap, j) piece holds the 16 bytes that same lane feeds to the MFMA, so the fragment read is a linear ds_read_b128).
//   WL 1: Cin <= one slab -- the weights are loaded once per workgroup and serve every tile it walks;
//   WL 2: two slabs -- both resident; more -- double buffer, the DMA of the next slab's weights runs under this slab's MFMAs.
//   WL 0: weights from global / L2 per tap group with a one-group register prefetch (kept for NTB 4 with several slabs, where
//         two images do not fit).  That path exposes an L2 round trip per group of three taps: at 12^3 x 256 channels, where
//         one workgroup per CU leaves nothing to hide it, a slab took 10 us for 108 MFMAs per wave.
template <int NTB, int WL> constexpr int conv_pipe_lds_bytes(int pitch) { return NHALO * pitch + WL * 27 * NTB * 1024; }
template <class P, int NTB, bool PAIR, int XM, int FUSE, int WL = 0>
__global__ void __launch_bounds__(256, ((PAIR || conv_pipe_lds_bytes<NTB, WL>(64) <= 80 * 1024) &&
                                        (FUSE == 2 ? (PAIR && NTB == 1) : (FUSE == 3 ? NTB == 1 : (FUSE == 1 && (NTB == 1 || (!PAIR && NTB == 2)))))) ? 2 : 1)
conv3_fwd_pipe_kernel(const void* __restrict__ x, long ldx, const char* __restrict__ wp, typename ActOf<P>::type* __restrict__ y, long ldy, int accumulate,
                      int D, int H, int W, int Cin, int Cout, int ntx, int nty, int ntz, int ntiles,
                      float* __restrict__ part, const char* __restrict__ wp3, typename ActOf<P>::type* __restrict__ y3, long ldy3,
                      float* __restrict__ part3, int K3) {
    typedef typename ActOf<P>::type YT;
    constexpr int CH = P::CH, SL = PAIR ? 16 : 4 * CH, NCH = PAIR ? 2 : 4, PITCH = NCH * 16, LAY = PAIR ? 0 : 1;
    // DMAW (pair layout on bf16-stored input: the 96^3 layers): the window goes global -> LDS by LDS-DMA into one of two images
    // (piece id = its 16-byte slot: the pair layout is linear in id), no registers, no ds_write, ONE barrier per tile; pieces
    // outside the volume are fetched from 16 zero bytes of the packed weights (tap 27 of pair 13)
    constexpr bool DMAW = PAIR && XM == 2;
    constexpr int WSTRIDE = HaloPlan<NCH>::ITERS * 256 * 16;      // bytes per window image incl. the slots of the idle lanes
    __shared__ __attribute__((aligned(16))) char halo[DMAW ? 2 * WSTRIDE : NHALO * PITCH];
    constexpr int WLN = PAIR ? 0 : WL, WBYTES = 27 * NTB * 1024;
    __shared__ __attribute__((aligned(16))) char wlds[WLN ? WLN * WBYTES : 16];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
    const int nt0 = blockIdx.y * NTB;
    const int nslab = PAIR ? 1 : (Cin + SL - 1) / SL;
    // this wave's share of the 27 * NTB (tap, j) pieces of one slab's weights: global -> LDS, no registers
    auto wdma = [&](int slab, int buf) {
        const char* src0 = wp + ((long)slab * Cout + nt0 * 16 + r) * 64 + g * 16;
        const long wtap = (long)nslab * Cout * 64;
        for (int p = wv; p < 27 * NTB; p += 4) {
            const int tap = p / NTB, j = p - tap * NTB;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(src0 + tap * wtap + (long)j * 16 * 64), (lds_void_t*)(wlds + buf * WBYTES + p * 1024), 16, 0, 0);
        }
    };
    const bool wrot = WLN > 0 && nslab > WLN;      // more slabs than images: double buffer indexed by a running slab count
    int wit = 0;
    if constexpr (WLN > 0) {
        if (wrot) wdma(0, 0);
        else for (int sl = 0; sl < nslab; ++sl) wdma(sl, sl);
    }

    u32x4 wres[PAIR ? 14 : 1][NTB];
    if constexpr (PAIR) {
#pragma unroll
        for (int tp = 0; tp < 14; ++tp)
#pragma unroll
            for (int j = 0; j < NTB; ++j) wres[tp][j] = *(const u32x4*)(wp + ((long)tp * Cout + (nt0 + j) * 16 + r) * 64 + g * 16);
    }

    constexpr bool has3 = FUSE == 2, late3 = FUSE == 3, any3 = has3 || late3;
    u32x4 w3res[any3 ? NTB : 1];   // 1x1x1 weights: PAIR / single-slab keep them for the whole launch, else reloaded per slab
    if constexpr ((has3 && PAIR) || late3) {
#pragma unroll
        for (int j = 0; j < NTB; ++j) w3res[j] = *(const u32x4*)(wp3 + ((long)(nt0 + j) * 16 + r) * 64 + g * 16);
    }
    constexpr bool BST = FUSE == 5;
    constexpr bool STATS = (FUSE >= 1 && FUSE <= 3) || BST;
    __shared__ float sred[STATS ? NTB * 128 : 1];
    f32x4 rs1[STATS ? NTB : 1], rs2[STATS ? NTB : 1], rt1[any3 ? NTB : 1], rt2[any3 ? NTB : 1];
    f32x4 bsa[BST ? NTB : 1], bso[BST ? NTB : 1];      // FUSE 5: n = xn * bsa + bso for this lane's channels of the current batch item
    if constexpr (STATS) {
#pragma unroll
        for (int j = 0; j < NTB; ++j) { rs1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rs2[j] = rs1[j]; }
    }
    if constexpr (any3) {
#pragma unroll
        for (int j = 0; j < NTB; ++j) { rt1[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; rt2[j] = rt1[j]; }
    }
    int cur_b = -1;
    const long srows = (long)gridDim.x;           // partial rows per batch item: one per workgroup
    if constexpr (STATS) {
        const int nb = ntiles / (ntx * nty * ntz);
        stats_zero_rows<NTB>(part, nb, srows, Cout, nt0 * 16);
        if constexpr (any3) stats_zero_rows<NTB>(part3, nb, srows, Cout, nt0 * 16);
    }

    const int ylane = r * (int)ldy + nt0 * 16 + 4 * g;      // this lane's element offset inside an output row of 16 voxels
    HaloRegs<P, NCH> R;
    HaloPlan<NCH> plan;
    if constexpr (XM == 2) halo_plan<P, NCH, LAY>(plan, H, W, (int)ldx, PITCH);
    const long item = (long)D * H * W * ldx;      // elements per batch item
    auto wload = [&](int b_, int z_, int y_, int x_, int c_) {
        if constexpr (XM == 2) halo_load_planned<P, NCH>(R, plan, (const uint16_t*)x + b_ * item, (int)ldx, z_, y_, x_, D, H, W, c_, Cin);
        else halo_load<P, NCH, XM>(R, x, ldx, b_, z_, y_, x_, D, H, W, c_, Cin);
    };
    // DMAW: window of tile (b_, z_, y_, x_) -> image `win`
    auto window_dma = [&](int win, int b_, int z_, int y_, int x_) {
        if constexpr (DMAW) {
            constexpr int TOTAL = NHALO * NCH;
            const uint16_t* __restrict__ xh = (const uint16_t*)x + b_ * item;
            const char* zsrc = wp + ((long)13 * Cout) * 64 + 32;             // 16 zero bytes
            const int ld32 = (int)ldx;
            const int base = (((z_ - 1) * H + (y_ - 1)) * W + (x_ - 1)) * ld32;
            const bool interior = z_ >= 1 && z_ - 1 + HZ <= D && y_ >= 1 && y_ - 1 + HY <= H && x_ >= 1 && x_ - 1 + HX <= W && NCH * CH <= Cin;
            char* img = halo + win * WSTRIDE + wv * 1024;
#pragma unroll
            for (int j = 0; j < HaloPlan<NCH>::ITERS; ++j) {
                const int id = threadIdx.x + j * 256;
                bool ok = (j + 1) * 256 <= TOTAL || id < TOTAL;
                if (!interior) {
                    const int hv = id / NCH, ch = id - hv * NCH;
                    const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                    ok = ok && (unsigned)(z_ - 1 + hz) < (unsigned)D && (unsigned)(y_ - 1 + hy) < (unsigned)H &&
                         (unsigned)(x_ - 1 + hx) < (unsigned)W && ch * CH < Cin;
                }
                const char* src = ok ? (const char*)(xh + (base + plan.rel[j])) : zsrc;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(img + j * 4096), 16, 0, 0);
            }
        }
    };
    int tile = blockIdx.x;
    TileTable tt;
    int kt = 0;                       // index of the current tile in this workgroup's walk
    int tx = 0, ty = 0, tz = 0, b = 0;
    int n1x = 0, n1y = 0, n1z = 0, n1b = 0;      // DMAW: coordinates of the tile after the current one
    if (tile < ntiles) {
        tt.get(0, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        if constexpr (DMAW) {
            window_dma(0, b, tz * TZ, ty * TY, tx * TX);
            if (tile + (int)gridDim.x < ntiles) {
                tt.get(1, ntiles, ntx, nty, ntz, n1x, n1y, n1z, n1b);
                window_dma(1, n1b, n1z * TZ, n1y * TY, n1x * TX);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            __syncthreads();
        } else {
            wload(b, tz * TZ, ty * TY, tx * TX, 0);
        }
    }
    for (; tile < ntiles; tile += gridDim.x, ++kt) {
        int ntx_ = tx, nty_ = ty, ntz_ = tz, nb_ = b;      // coordinates of the next tile (set when its prefetch is issued)
        if constexpr (DMAW) { ntx_ = n1x; nty_ = n1y; ntz_ = n1z; nb_ = n1b; }
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        f32x4 acc[4][NTB];
        f32x4 acc3[has3 ? 4 : 1][has3 ? NTB : 1];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NTB; ++j) {
                acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (has3) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        if constexpr (STATS) {
            if (b != cur_b) {
                if (cur_b >= 0) {
                    stats_flush<NTB>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, nt0 * 16, r, g, wv, sred);
                    if constexpr (any3) stats_flush<NTB>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, nt0 * 16, r, g, wv, sred);
                }
                cur_b = b;
                if constexpr (BST) {
                    const float* sa = (const float*)wp3 + ((long)b * Cout + nt0 * 16 + 4 * g) * 2;
#pragma unroll
                    for (int j = 0; j < NTB; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float mu = sa[(j * 16 + e) * 2], rs = sa[(j * 16 + e) * 2 + 1];
                            bsa[j][e] = rs; bso[j][e] = -mu * rs;
                        }
                }
            }
        }

        // FUSE 4: the second input's fragments (x3 rows of this tile, 1x1x1 weights) of k-block kb, straight from global memory
        u32x4 w3f0[FUSE == 4 ? NTB : 1], av0[FUSE == 4 ? 4 : 1];
        auto x3_load = [&](int kb, int z0_, int y0_, int x0_, int b_, u32x4 (&wf)[FUSE == 4 ? NTB : 1], u32x4 (&av)[FUSE == 4 ? 4 : 1]) {
            if constexpr (FUSE == 4) {
#pragma unroll
                for (int j = 0; j < NTB; ++j) wf[j] = *(const u32x4*)(wp3 + ((long)kb * Cout + (nt0 + j) * 16 + r) * 64 + g * 16);
                const int c3 = kb * 4 * CH + g * CH, zo_ = z0_ + wv;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yo = y0_ + i, xo = x0_ + r;
                    const bool ok = zo_ < D && yo < H && xo < W && c3 + CH <= K3;
                    av[i] = act_chunk<P>((const YT*)y3 + (ok ? ((((long)b_ * D + zo_) * H + yo) * W + xo) * ldy3 + c3 : 0), ok);
                }
            }
        };
        // FUSE 5: this lane's pre-norm values xn of the tile's output positions (4 channels x 4 rows x NTB), requested before the
        // MFMA phase and kept packed until the epilogue
        typedef typename std::conditional<sizeof(YT) == 2, u32x2, f32x4>::type XnRaw;
        XnRaw xnr[BST ? 4 : 1][BST ? NTB : 1];
        auto xn_load = [&]() {
            if constexpr (BST) {
                const int zo_ = z0 + wv, xo_ = x0 + r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yo_ = y0 + i;
                    const bool ok = zo_ < D && yo_ < H && xo_ < W;
                    const YT* q = (const YT*)y3 + (ok ? ((((long)b * D + zo_) * H + yo_) * W + xo_) * ldy3 + nt0 * 16 + 4 * g : 0);
#pragma unroll
                    for (int j = 0; j < NTB; ++j) xnr[i][j] = *(const XnRaw*)(q + j * 16);
                }
            }
        };
        for (int slab = 0; slab < nslab; ++slab) {
          if constexpr (DMAW) {
            if constexpr (FUSE == 4) x3_load(0, z0, y0, x0, b, w3f0, av0);
            if constexpr (BST) xn_load();
          } else {
            __syncthreads();                       // everyone is done reading the previous window
            if constexpr (XM == 2) halo_store_planned<P, NCH>(R, plan, halo);    // (waits for the prefetched loads)
            else halo_store<P, NCH, XM, LAY>(R, PITCH, halo);
            if constexpr (WLN > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's weight DMA pieces have landed
            __syncthreads();
            // every older VMEM operation (the previous tile's output stores) has retired before the prefetch below is issued: the
            // fragment registers of the MFMA phase are the ones those stores read, and guarding them with the in-order vmcnt
            // counter AFTER the prefetch had been issued made the MFMA phase wait for the prefetch itself (s_waitcnt vmcnt(0)
            // at its top)
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) only
            {   // prefetch the next (tile, slab) window; it lands while the MFMAs below run
                int ntile = tile, nslb = slab + 1;
                if (nslb == nslab) { ntile = tile + gridDim.x; nslb = 0; }
                if constexpr (WLN > 0) {
                    if (wrot && ntile < ntiles) wdma(nslb, (wit + 1) & 1);      // (that image was last read before the barrier above)
                }
                if (ntile < ntiles) {
                    int ax = tx, ay = ty, az = tz, ab = b;
                    if (nslb == 0) {
                        tt.get(kt + 1, ntiles, ntx, nty, ntz, ax, ay, az, ab);
                        ntx_ = ax; nty_ = ay; ntz_ = az; nb_ = ab;
                    }
                    wload(ab, az * TZ, ay * TY, ax * TX, nslb * SL);
                }
                // (requested behind the window prefetch: in flight during the MFMA phase, consumed in the tile epilogue -- loaded
                // there, every tile waited a global round trip for them)
                if constexpr (FUSE == 4) { if (slab == 0) x3_load(0, z0, y0, x0, b, w3f0, av0); }
                if constexpr (BST) { if (slab == nslab - 1) xn_load(); }
            }
          }
            if constexpr (PAIR) {
                // 56 (tap pair, row) steps; the window fragments of the next DPT steps are in flight while a step's MFMAs run
                // (written as load, MFMA, load, MFMA the compiler kept exactly that order with ONE fragment register and a
                // full lgkmcnt(0) wait per MFMA: ~130 cycles of LDS latency exposed 56 times per tile = 80 % of the tile time)
                const char* hb0 = halo + (DMAW ? (kt & 1) * WSTRIDE : 0) + ((wv * HY) * HX + r) * PITCH + (g & 1) * 16;
                const bool upper = (g >> 1) != 0;
                constexpr int NSTEP = 14 * 4, DPT = 12;
                auto frag = [&](int s) {
                    const int tp = s >> 2, i = s & 3;
                    const int tA = 2 * tp, tB = (2 * tp + 1 < 27) ? 2 * tp + 1 : 26;
                    const int offA = (((tA / 9) * HY + (tA % 9) / 3) * HX + tA % 3) * PITCH;
                    const int offB = (((tB / 9) * HY + (tB % 9) / 3) * HX + tB % 3) * PITCH;
                    return *(const u32x4*)(hb0 + (upper ? offB : offA) + i * HX * PITCH);
                };
                u32x4 ring[DPT];
#pragma unroll
                for (int s = 0; s < DPT; ++s) ring[s] = frag(s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    const int tp = s >> 2, i = s & 3;
                    const u32x4 a = ring[s % DPT];
#pragma unroll
                    for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], wres[tp][j], a);
                    if constexpr (has3) {
                        if (tp == 6) {             // w3res is zero on the tap-12 half of the K range
#pragma unroll
                            for (int j = 0; j < NTB; ++j) P::mma(acc3[i][j], w3res[j], a);
                        }
                    }
                    if (s + DPT < NSTEP) ring[s % DPT] = frag(s + DPT);
                    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks every load back in front of its use)
                }
            } else if constexpr (WLN > 0) {
                const int xo3[3] = {lay_off<LAY>(r, g, PITCH), lay_off<LAY>(r + 1, g, PITCH), lay_off<LAY>(r + 2, g, PITCH)};
                const char* wb = wlds + (wrot ? (wit & 1) : slab) * WBYTES + lane * 16;
                if constexpr (has3) {
#pragma unroll
                    for (int j = 0; j < NTB; ++j) w3res[j] = *(const u32x4*)(wp3 + ((long)slab * Cout + (nt0 + j) * 16 + r) * 64 + g * 16);
                }
                // 108 (tap, row) MFMA steps walked as 9 (dz, dx) groups of 12 (dy, output row i) steps: the window fragment of input row
                // dy + i is the same for every (dy, i) with that sum, so a group reads SIX row fragments for its twelve steps (54 window
                // reads per tile instead of 108; with Cout = 16 every fragment feeds one MFMA and the 135 ds_read_b128 of a wave's tile
                // kept the CU's LDS pipe busier than its matrix pipes).  Software-pipelined as before: the six fragments of the next
                // group are requested behind the first six steps of this one, the weight fragments of the next tap one tap ahead.
                const char* hw = halo + (wv * HY * HX) * PITCH;
                auto rfrag = [&](int grp, int row) {
                    const int dz = grp / 3, dx = grp % 3;
                    return *(const u32x4*)(hw + ((dz * HY + row) * HX) * PITCH + xo3[dx]);
                };
                u32x4 fr[2][6], bw[2][NTB];
#pragma unroll
                for (int j = 0; j < NTB; ++j) bw[0][j] = *(const u32x4*)(wb + j * 1024);
#pragma unroll
                for (int q = 0; q < 6; ++q) fr[0][q] = rfrag(0, q);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int grp = 0; grp < 9; ++grp) {
                    const int dz = grp / 3, dx = grp % 3;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int tap = dz * 9 + dy * 3 + dx;
                        const int ord = grp * 3 + dy;                      // position of this tap in the walk (weights double-buffered along it)
                        // the tap that follows in the walk
                        const int ngrp = dy < 2 ? grp : grp + 1, ndy = dy < 2 ? dy + 1 : 0;
                        const int ntap = (ngrp / 3) * 9 + ndy * 3 + ngrp % 3;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (i == 0 && ord + 1 < 27) {
#pragma unroll
                                for (int j = 0; j < NTB; ++j) bw[(ord + 1) & 1][j] = *(const u32x4*)(wb + (ntap * NTB + j) * 1024);
                            }
                            const u32x4 a = fr[grp & 1][dy + i];
#pragma unroll
                            for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], bw[ord & 1][j], a);
                            if constexpr (has3) {
                                if (tap == 13) {
#pragma unroll
                                    for (int j = 0; j < NTB; ++j) P::mma(acc3[i][j], w3res[j], a);
                                }
                            }
                            const int k = dy * 4 + i;
                            if (k < 6 && grp + 1 < 9) fr[(grp + 1) & 1][k] = rfrag(grp + 1, k);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                ++wit;
            } else {
                constexpr int GT = NTB <= 2 ? 3 : 1, NG = 27 / GT;   // small groups: the halo prefetch registers are live here
                // per-lane byte offset of (window column r + dx, chunk g) for the three dx of a tap row (LAY 1 swizzle folded in)
                const int xo3[3] = {lay_off<LAY>(r, g, PITCH), lay_off<LAY>(r + 1, g, PITCH), lay_off<LAY>(r + 2, g, PITCH)};
                const char* wbase = wp + ((long)slab * Cout + nt0 * 16 + r) * 64 + g * 16;
                const long wtap = (long)nslab * Cout * 64;
                u32x4 bcur[GT][NTB], bnxt[GT][NTB];
                if constexpr (has3) {
#pragma unroll
                    for (int j = 0; j < NTB; ++j) w3res[j] = *(const u32x4*)(wp3 + ((long)slab * Cout + (nt0 + j) * 16 + r) * 64 + g * 16);
                }
#pragma unroll
                for (int t = 0; t < GT; ++t)
#pragma unroll
                    for (int j = 0; j < NTB; ++j) bcur[t][j] = *(const u32x4*)(wbase + t * wtap + j * 16 * 64);
                for (int tg = 0; tg < NG; ++tg) {
                    if (tg + 1 < NG) {
#pragma unroll
                        for (int t = 0; t < GT; ++t)
#pragma unroll
                            for (int j = 0; j < NTB; ++j) bnxt[t][j] = *(const u32x4*)(wbase + ((tg + 1) * GT + t) * wtap + j * 16 * 64);
                    }
#pragma unroll
                    for (int t = 0; t < GT; ++t) {
                        const int tap = tg * GT + t;
                        const int dz = tap / 9, rem = tap - dz * 9, dy = rem / 3, dx = rem - dy * 3;
                        // (GT == 3: dx is the unrolled t; otherwise a wave-uniform runtime value -> selects, never an indexed array)
                        const int xo = GT == 3 ? xo3[t % 3] : (dx == 0 ? xo3[0] : (dx == 1 ? xo3[1] : xo3[2]));
                        const char* hbase = halo + (((wv + dz) * HY + dy) * HX) * PITCH + xo;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            u32x4 a = *(const u32x4*)(hbase + i * HX * PITCH);
#pragma unroll
                            for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], bcur[t][j], a);
                            if constexpr (has3) {
                                if (tap == 13) {
#pragma unroll
                                    for (int j = 0; j < NTB; ++j) P::mma(acc3[i][j], w3res[j], a);
                                }
                            }
                        }
                    }
#pragma unroll
                    for (int t = 0; t < GT; ++t)
#pragma unroll
                        for (int j = 0; j < NTB; ++j) bcur[t][j] = bnxt[t][j];
                }
            }
        }
        if constexpr (DMAW) {
            // the image of the next tile has landed (requested a whole tile ago) and so have the previous tile's output stores; after
            // the barrier every wave is done with this tile's image, which the tile after next now streams into.  The output
            // stores of this tile follow, so the wait above never waits for a store it has just issued.
            __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            __syncthreads();
            if (tile + 2 * (int)gridDim.x < ntiles) {
                tt.get(kt + 2, ntiles, ntx, nty, ntz, n1x, n1y, n1z, n1b);
                window_dma(kt & 1, n1b, n1z * TZ, n1y * TY, n1x * TX);
            }
        }
        const int zo = z0 + wv;
        {
            // this lane's voxel column x0 + r of the four output rows; out-of-volume voxels are clamped to voxel 0 of the tensor for
            // the (batched, unconditional) read of the accumulate path and skipped on store
            YT* yrow[4];
            bool okv[4];
            const int xo = x0 + r;
            const long tb = (((long)b * D + zo) * H + y0) * W + x0;          // wave-uniform: voxel (zo, y0, x0)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yo = y0 + i;
                okv[i] = zo < D && yo < H && xo < W;
                yrow[i] = okv[i] ? y + (tb + (long)i * W) * ldy + ylane : y + (nt0 * 16 + 4 * g);
            }
            if constexpr (FUSE == 4) {
                // acc += x3[voxel, :] . w3: A rows = the 16 x-positions of output row i (lane r), chunk g of each 64-byte k-block.
                // The first k-block's operands were requested before the tile's MFMA phase (x3_load), the rest (K3 > 32 bf16
                // channels) are loaded here.
                const int n3 = (K3 + 4 * CH - 1) / (4 * CH);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], w3f0[j], av0[i]);
                }
                for (int kb = 1; kb < n3; ++kb) {
                    u32x4 w3f[NTB], av[4];
                    x3_load(kb, z0, y0, x0, b, w3f, av);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int j = 0; j < NTB; ++j) P::mma(acc[i][j], w3f[j], av[i]);
                    }
                }
            }
            if (accumulate) {
                f32x4 old[4][NTB];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTB; ++j) old[i][j] = Io<YT>::ld4(yrow[i] + j * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTB; ++j)
                        if (okv[i]) Io<YT>::st4(yrow[i] + j * 16, acc[i][j] + old[i][j]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTB; ++j)
                        if (okv[i]) Io<YT>::st4(yrow[i] + j * 16, acc[i][j]);
            }
            if constexpr (STATS && !BST) stats_add<NTB>(acc, okv, rs1, rs2);
            if constexpr (BST) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTB; ++j) {
                        f32x4 xn;
                        if constexpr (sizeof(YT) == 2) xn = __builtin_convertvector(__builtin_bit_cast(bf16x4, xnr[i][j]), f32x4);
                        else xn = __builtin_bit_cast(f32x4, xnr[i][j]);
                        const f32x4 n = xn * bsa[j] + bso[j];
                        f32x4 ge;
#pragma unroll
                        for (int e = 0; e < 4; ++e) ge[e] = n[e] > 0.f ? acc[i][j][e] : 0.01f * acc[i][j][e];
                        if (okv[i]) { rs1[j] += ge; rs2[j] += ge * n; }
                    }
            }
            if constexpr (has3) {
                if (y3) {                                 // (y3 == nullptr: the caller only wants the branch's statistics)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        YT* q = y3 + (yrow[i] - y);          // same pitch as y (checked by the host)
#pragma unroll
                        for (int j = 0; j < NTB; ++j)
                            if (okv[i]) Io<YT>::st4(q + j * 16, acc3[i][j]);
                    }
                }
                stats_add<NTB>(acc3, okv, rt1, rt2);
            }
            if constexpr (late3) {
                // the window of this tile is still in LDS (the next halo_store waits behind the barrier at the loop top)
                const char* hbase = halo + (((wv + 1) * HY + 1) * HX) * PITCH + lay_off<LAY>(r + 1, g, PITCH);      // centre tap
                f32x4 a3[4][NTB];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u32x4 a = *(const u32x4*)(hbase + i * HX * PITCH);
#pragma unroll
                    for (int j = 0; j < NTB; ++j) {
                        a3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        P::mma(a3[i][j], w3res[j], a);
                    }
                }
                if (y3) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        YT* q = y3 + (yrow[i] - y);
#pragma unroll
                        for (int j = 0; j < NTB; ++j)
                            if (okv[i]) Io<YT>::st4(q + j * 16, a3[i][j]);
                    }
                }
                stats_add<NTB>(a3, okv, rt1, rt2);
            }
        }
        tx = ntx_; ty = nty_; tz = ntz_; b = nb_;
    }
    if constexpr (STATS) {
        if (cur_b >= 0) {
            stats_flush<NTB>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, nt0 * 16, r, g, wv, sred);
            if constexpr (any3) stats_flush<NTB>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, nt0 * 16, r, g, wv, sred);
        }
    }
}

// ---- ONE input channel, 16 output channels (the image in front of encoder1: UnetrBasicBlock(in_channels, feature_size),
// unetr.py:90-98), bf16 mode.  The generic pair-mode kernel treats the image as 16 zero-padded channels: 14 K=32 MFMAs per 16
// voxels of which 27 of 448 products are not zero, and a window staged with scalar loads (60 us at 96^3 for 113 MB of output).
// Here the contraction index IS the tap: [16 voxels, 27 taps (K = 32)] x [32, 16 channels] = one MFMA per 16 voxels.  A window of
// 6 x 6 x 18 bf16 image values (1.3 KB) is staged per tile; lane (voxel r, tap group g) gathers its eight taps with ds_read_u16;
// the 1x1x1 branch (UnetResBlock.conv3 on the same input) is x[v] * w3[co] on the VALU -- the exact product of the two bf16
// values, as the MFMA formed it.  Weights are read from the pair-mode packs (element [tap >> 1][co][(tap & 1) * 16] of wp, element
// [co][16] of wp3).  Same tile walk, output layout and InstanceNorm partial rows as conv3_fwd_pipe_kernel<..., PAIR, FUSE 2>.
__global__ void __launch_bounds__(256, 4)
conv3_c1_fwd_kernel(const float* __restrict__ x, const uint16_t* __restrict__ wp, uint16_t* __restrict__ y, long ldy,
                    int D, int H, int W, int ntx, int nty, int ntz, int ntiles, float* __restrict__ part,
                    const uint16_t* __restrict__ wp3, uint16_t* __restrict__ y3, float* __restrict__ part3) {
    constexpr int Cout = 16, NPT = (NHALO + 255) / 256;
    __shared__ uint16_t win[NHALO + 8];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
    const bool has3 = wp3 != nullptr;
    // weight fragment (first MFMA operand: rows = channels): lane (co = r, g) holds taps 8g .. 8g+7 of channel r
    u32x4 wfrag;
    {
        uint16_t wv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * g + j;
            wv8[j] = tap < 27 ? wp[((long)(tap >> 1) * Cout + r) * 32 + (tap & 1) * 16] : (uint16_t)0;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) wfrag[d] = (uint32_t)wv8[2 * d] | ((uint32_t)wv8[2 * d + 1] << 16);
    }
    float w3f[4] = {0.f, 0.f, 0.f, 0.f};
    if (has3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w3f[e] = __builtin_bit_cast(float, (uint32_t)wp3[(long)(4 * g + e) * 32 + 16] << 16);
    }
    // window offsets (elements) of this lane's eight taps relative to voxel (plane wv, row 0, column r); taps 27..31 (zero weights)
    // read the centre tap: any finite value
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tap = 8 * g + j < 27 ? 8 * g + j : 13;
        toff[j] = ((tap / 9) * HY + (tap % 9) / 3) * HX + tap % 3;
    }
    const int base0 = (wv * HY) * HX + r;
    const int cen = (HY + 1) * HX + 1;                    // centre tap
    // InstanceNorm partial sums (see stats_add / stats_flush)
    f32x4 rs1[1] = {{0.f, 0.f, 0.f, 0.f}}, rs2[1] = {{0.f, 0.f, 0.f, 0.f}}, rt1[1] = {{0.f, 0.f, 0.f, 0.f}}, rt2[1] = {{0.f, 0.f, 0.f, 0.f}};
    int cur_b = -1;
    const long srows = (long)gridDim.x;
    __shared__ float sred[128];
    {
        const int nb = ntiles / (ntx * nty * ntz);
        stats_zero_rows<1>(part, nb, srows, Cout, 0);
        if (has3) stats_zero_rows<1>(part3, nb, srows, Cout, 0);
    }
    const long item = (long)D * H * W;
    // this thread's window pieces of tile (b_, z_, y_, x_): image values, 0 outside the volume
    float nxt[NPT];
    auto wload = [&](int b_, int z_, int y_, int x_) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int hz = id / (HY * HX), rem = id - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z_ - 1 + hz, gy = y_ - 1 + hy, gx = x_ - 1 + hx;
            const bool ok = id < NHALO && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            float t = 0.f;
            if (ok) t = x[b_ * item + ((long)gz * H + gy) * W + gx];
            nxt[j] = t;
        }
    };
    TileTable tt;
    int kt = 0, tx = 0, ty = 0, tz = 0, b = 0;
    if ((int)blockIdx.x < ntiles) {
        tt.get(0, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        wload(b, tz * TZ, ty * TY, tx * TX);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++kt) {
        int ax = tx, ay = ty, az = tz, ab = b;
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        if (b != cur_b) {
            if (cur_b >= 0) {
                stats_flush<1>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
                if (has3) stats_flush<1>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
            }
            cur_b = b;
        }
        __syncthreads();                                     // every wave is done with the previous window
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < NHALO) win[id] = f2bf(nxt[j]);
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) {                // the next tile's window is in flight during this tile's work
            tt.get(kt + 1, ntiles, ntx, nty, ntz, ax, ay, az, ab);
            wload(ab, az * TZ, ay * TY, ax * TX);
        }
        const int zo = z0 + wv, xo = x0 + r;
        f32x4 acc[4][1], acc3[4][1];
        bool okv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint16_t* wb = win + base0 + i * HX;
            uint16_t a8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a8[j] = wb[toff[j]];
            u32x4 af;
#pragma unroll
            for (int d = 0; d < 4; ++d) af[d] = (uint32_t)a8[2 * d] | ((uint32_t)a8[2 * d + 1] << 16);
            acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            PrecBF16::mma(acc[i][0], wfrag, af);
            const float xc = __builtin_bit_cast(float, (uint32_t)wb[cen] << 16);
            acc3[i][0] = (f32x4){xc * w3f[0], xc * w3f[1], xc * w3f[2], xc * w3f[3]};
            okv[i] = zo < D && y0 + i < H && xo < W;
        }
        const long tb = (((long)b * D + zo) * H + y0) * W + xo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (okv[i]) {
                Io<uint16_t>::st4(y + (tb + (long)i * W) * ldy + 4 * g, acc[i][0]);
                if (has3 && y3) Io<uint16_t>::st4(y3 + (tb + (long)i * W) * ldy + 4 * g, acc3[i][0]);      // (y3 == nullptr: statistics only)
            }
        }
        stats_add<1>(acc, okv, rs1, rs2);
        if (has3) stats_add<1>(acc3, okv, rt1, rt2);
        tx = ax; ty = ay; tz = az; b = ab;
    }
    if (cur_b >= 0) {
        stats_flush<1>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
        if (has3) stats_flush<1>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
    }
}

// the same kernel for the bf16x3 mode (fp32 feature maps, split weights): the window travels as two bf16 arrays (hi, lo), the 27-tap
// contraction is three MFMAs (w_hi x_hi + w_hi x_lo + w_lo x_hi), the 1x1x1 branch an fp32 product of the recombined halves.  The
// generic slab kernel spent 112 us on this layer (one real channel in a 16-channel slab, 27 MFMA pairs per row).
__global__ void __launch_bounds__(256, 4)
conv3_c1_fwd_x3_kernel(const float* __restrict__ x, const uint32_t* __restrict__ wp, float* __restrict__ y, long ldy,
                    int D, int H, int W, int ntx, int nty, int ntz, int ntiles, float* __restrict__ part,
                    const uint32_t* __restrict__ wp3, float* __restrict__ y3, float* __restrict__ part3) {
    constexpr int Cout = 16, NPT = (NHALO + 255) / 256;
    __shared__ uint16_t win[NHALO + 8], winl[NHALO + 8];          // the window's hi and lo bf16 halves
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 15, g = lane >> 4;
    const bool has3 = wp3 != nullptr;
    // weight fragment (first MFMA operand: rows = channels): lane (co = r, g) holds taps 8g .. 8g+7 of channel r
    // (packed weights of this mode: one word [hi | lo << 16] per element, [tap][co][16 k] with the single input channel at k = 0)
    u32x4 wfrag, wfragl;
    {
        uint32_t wv8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 8 * g + j;
            wv8[j] = tap < 27 ? wp[((long)tap * Cout + r) * 16] : 0u;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            wfrag[d] = (wv8[2 * d] & 0xffffu) | (wv8[2 * d + 1] << 16);
            wfragl[d] = (wv8[2 * d] >> 16) | (wv8[2 * d + 1] & 0xffff0000u);
        }
    }
    float w3f[4] = {0.f, 0.f, 0.f, 0.f};
    if (has3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t t = wp3[(long)(4 * g + e) * 16];
            w3f[e] = __builtin_bit_cast(float, t << 16) + __builtin_bit_cast(float, t & 0xffff0000u);
        }
    }
    // window offsets (elements) of this lane's eight taps relative to voxel (plane wv, row 0, column r); taps 27..31 (zero weights)
    // read the centre tap: any finite value
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int tap = 8 * g + j < 27 ? 8 * g + j : 13;
        toff[j] = ((tap / 9) * HY + (tap % 9) / 3) * HX + tap % 3;
    }
    const int base0 = (wv * HY) * HX + r;
    const int cen = (HY + 1) * HX + 1;                    // centre tap
    // InstanceNorm partial sums (see stats_add / stats_flush)
    f32x4 rs1[1] = {{0.f, 0.f, 0.f, 0.f}}, rs2[1] = {{0.f, 0.f, 0.f, 0.f}}, rt1[1] = {{0.f, 0.f, 0.f, 0.f}}, rt2[1] = {{0.f, 0.f, 0.f, 0.f}};
    int cur_b = -1;
    const long srows = (long)gridDim.x;
    __shared__ float sred[128];
    {
        const int nb = ntiles / (ntx * nty * ntz);
        stats_zero_rows<1>(part, nb, srows, Cout, 0);
        if (has3) stats_zero_rows<1>(part3, nb, srows, Cout, 0);
    }
    const long item = (long)D * H * W;
    // this thread's window pieces of tile (b_, z_, y_, x_): image values, 0 outside the volume
    float nxt[NPT];
    auto wload = [&](int b_, int z_, int y_, int x_) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int hz = id / (HY * HX), rem = id - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z_ - 1 + hz, gy = y_ - 1 + hy, gx = x_ - 1 + hx;
            const bool ok = id < NHALO && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            float t = 0.f;
            if (ok) t = x[b_ * item + ((long)gz * H + gy) * W + gx];
            nxt[j] = t;
        }
    };
    TileTable tt;
    int kt = 0, tx = 0, ty = 0, tz = 0, b = 0;
    if ((int)blockIdx.x < ntiles) {
        tt.get(0, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        wload(b, tz * TZ, ty * TY, tx * TX);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++kt) {
        int ax = tx, ay = ty, az = tz, ab = b;
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        if (b != cur_b) {
            if (cur_b >= 0) {
                stats_flush<1>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
                if (has3) stats_flush<1>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
            }
            cur_b = b;
        }
        __syncthreads();                                     // every wave is done with the previous window
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < NHALO) { const uint32_t t = PrecBF16x3::split(nxt[j]); win[id] = (uint16_t)t; winl[id] = (uint16_t)(t >> 16); }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) {                // the next tile's window is in flight during this tile's work
            tt.get(kt + 1, ntiles, ntx, nty, ntz, ax, ay, az, ab);
            wload(ab, az * TZ, ay * TY, ax * TX);
        }
        const int zo = z0 + wv, xo = x0 + r;
        f32x4 acc[4][1], acc3[4][1];
        bool okv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint16_t* wb = win + base0 + i * HX;
            const uint16_t* wbl = winl + base0 + i * HX;
            uint16_t a8[8], l8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { a8[j] = wb[toff[j]]; l8[j] = wbl[toff[j]]; }
            u32x4 af, afl;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                af[d] = (uint32_t)a8[2 * d] | ((uint32_t)a8[2 * d + 1] << 16);
                afl[d] = (uint32_t)l8[2 * d] | ((uint32_t)l8[2 * d + 1] << 16);
            }
            acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            PrecBF16::mma(acc[i][0], wfrag, af);            // w_hi x_hi + w_hi x_lo + w_lo x_hi
            PrecBF16::mma(acc[i][0], wfrag, afl);
            PrecBF16::mma(acc[i][0], wfragl, af);
            const float xc = __builtin_bit_cast(float, (uint32_t)wb[cen] << 16) + __builtin_bit_cast(float, (uint32_t)wbl[cen] << 16);
            acc3[i][0] = (f32x4){xc * w3f[0], xc * w3f[1], xc * w3f[2], xc * w3f[3]};
            okv[i] = zo < D && y0 + i < H && xo < W;
        }
        const long tb = (((long)b * D + zo) * H + y0) * W + xo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (okv[i]) {
                Io<float>::st4(y + (tb + (long)i * W) * ldy + 4 * g, acc[i][0]);
                if (has3 && y3) Io<float>::st4(y3 + (tb + (long)i * W) * ldy + 4 * g, acc3[i][0]);      // (y3 == nullptr: statistics only)
            }
        }
        stats_add<1>(acc, okv, rs1, rs2);
        if (has3) stats_add<1>(acc3, okv, rt1, rt2);
        tx = ax; ty = ay; tz = az; b = ab;
    }
    if (cur_b >= 0) {
        stats_flush<1>(rs1, rs2, part + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
        if (has3) stats_flush<1>(rt1, rt2, part3 + (((long)cur_b * srows) + blockIdx.x) * 2 * Cout, Cout, 0, r, g, wv, sred);
    }
}

// 1x1x1 weights w3[Cout][Cin] in the B-fragment layout the fused kernel reads at the centre tap:
// pair mode: wp3[n][32], k = 16 + ci (the tap-13 half of pair 6), zero elsewhere; slab mode: wp3[slab][n][SL], k = ci - slab*SL
// transposed = 1 (data gradient): rows n are INPUT channels of the conv and k its output channels: element = w3[k][n]
template <class T>
__global__ void conv3_pack_1x1_kernel(const float* __restrict__ w3, T* __restrict__ wp3, int Cin, int Cout, int pair, int SL, int transposed) {
    const int nslab = pair ? 1 : (Cin + SL - 1) / SL, RW = pair ? 32 : SL;
    const long total = (long)nslab * Cout * RW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int kk = (int)(i % RW); const long t = i / RW; const int n = (int)(t % Cout), slab = (int)(t / Cout);
        const int ci = pair ? kk - 16 : slab * SL + kk;
        const float v = (ci >= 0 && ci < Cin) ? (transposed ? w3[(long)ci * Cout + n] : w3[(long)n * Cin + ci]) : 0.f;
        wp3[i] = cvt_elem<T>(v);
    }
}

// -------------------------------------------------------------------------------------- weight grad
// workgroup = (voxel-tile group, ci slab of CIS 16-channel tiles, 16-channel co tile); 27*CIS (tap, ci-tile) units over
// 4 waves.  CIS = 1 for <= 16 input channels (the 96^3 layers): half the window bytes in LDS, half the transposing reads
// and MFMAs of the 32-channel slab (whose upper ci-tile would be all zeros there).
template <class P, bool HAS3 = false, int CIS = 2> struct WgCfg {
    typedef typename ElemOf<P>::type T;
    static constexpr int ES = sizeof(T);
    // image pitches: 16 B of padding when it is free; with the second dy image (HAS3) in bf16 the padding is dropped
    // so that two workgroups still fit a CU (measured: 76 KB per workgroup = one workgroup per CU, 1.35x slower)
    // FLIP: the bf16 16-channel slab (every weight gradient of the network by default): unpadded 32-byte voxels in the
    // LAY 2 column order -- both transposing reads conflict-free (tools/lds_conflicts.py), 29 KB per workgroup
    static constexpr bool FLIP = ES == 2 && CIS == 1;
    static constexpr int PX = FLIP ? 32 : 16 * CIS * ES + ((HAS3 && ES == 2) ? 0 : 16);   // halo image pitch (16*CIS channels)
    static constexpr int PY = FLIP ? 32 : 16 * ES + ((HAS3 && ES == 2) ? 8 : 16);        // dy image pitch (16 channels)
    static constexpr int LAY = FLIP ? 2 : 0;
    static constexpr int KV = 4 * P::CH;      // voxels per MFMA k-block (32 bf16 / 16 f32)
    static constexpr int NKB = NVOX / KV;
    static constexpr int NU = 27 * CIS;                       // (tap, ci-tile) units
    static constexpr int NUX = NU + (HAS3 ? CIS : 0);         // + centre-tap units fed by dy3 (the 1x1x1 conv)
    static constexpr int UPW = (NUX + 3) / 4;                 // units per wave
};

// XMX: storage of x (see stage_halo); dy / dy3 are feature-map gradients: ActOf<P> (bf16 in bf16 mode -- VECY is then moot:
// a 16-channel dy row is two 16-byte pieces)
// resident weight-gradient workgroups per CU for the 16-channel slab: with the software-pipelined unit loop and the staging plans two
// workgroups of <= 181 registers beat three with spills (5.17 vs 5.26 ms per step); the host sizes the grid to match
#ifndef WG_LB
#define WG_LB 2
#endif
template <int V> struct IntC { static constexpr int value = V; };
// NSL = 16-channel input slabs per workgroup (1 or 2; 2 only on the bf16 / conflict-free-layout / planned-staging path): the dy
// (and dy3) tile is staged ONCE per voxel tile and serves both slabs -- with one slab per workgroup the 32-channel layers staged it
// twice, the 64-channel ones four times (PMC: 1.73x the algorithmic bytes for the family).  Per tile: window of slab 0 + dy ->
// MFMAs of slab 0 (the window of slab 1 in flight) -> window of slab 1 over the same image -> MFMAs of slab 1 (the next tile's
// slab-0 window + dy in flight); two accumulator sets.
template <class P, int XMX, bool VECY, bool HAS3, int CIS, bool PIPE_OK, int NSL = 1>
__global__ void __launch_bounds__(256, CIS == 1 ? ((HAS3 && !WgCfg<P, HAS3, CIS>::FLIP) ? 2 : WG_LB) : 1)   // CIS = 1: WG_LB workgroups per CU
conv3_wgrad_kernel(const void* __restrict__ x, long ldx, const typename ActOf<P>::type* __restrict__ dy, long lddy, float* __restrict__ part,
                   const typename ActOf<P>::type* __restrict__ dy3, long lddy3, float* __restrict__ part3,
                   int D, int H, int W, int Cin, int Cout, int ntx, int nty, int ntz, int ntiles) {
    typedef typename ActOf<P>::type GT;
    using C = WgCfg<P, HAS3, CIS>;
    typedef typename C::T T;
    constexpr int CH = P::CH, WG_UNITS = C::NU, WG_UPW = C::UPW;
    // dy3 != nullptr: also accumulate the 1x1x1 conv's weight gradient dw3[co][ci] = sum_v dy3[v,co] x[v,ci] of the
    // same residual block (MONAI UnetResBlock.conv3 shares its input with conv1): the extra units = centre tap fed by dy3.
    __shared__ __attribute__((aligned(16))) char lds[NHALO * C::PX + (HAS3 ? 2 : 1) * NVOX * C::PY];
    char* ximg = lds;
    char* yimg = lds + NHALO * C::PX;
    char* y3img = yimg + NVOX * C::PY;
    constexpr int nunits = C::NUX;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    static_assert(NSL == 1 || (CIS == 1 && PIPE_OK && XMX == 2 && sizeof(typename ElemOf<P>::type) == 2), "two slabs: fast path only");
    const int ci0 = blockIdx.y * 16 * CIS * NSL, co0 = blockIdx.z * 16;
    // per-unit LDS byte offsets of the shifted window (wave-uniform)
    // FLIP layout: the column part of the address is per lane AND per tap column dx (the LAY 2 permutation is not a shift):
    // the three candidates of this lane's two voxel rows are formed once (they do not depend on the k-block: v & 15 is
    // (8g + q) & 15 for every kb), the per-unit choice is a wave-uniform select
    int uoff[WG_UPW];
#pragma unroll
    for (int ui = 0; ui < WG_UPW; ++ui) {
        int u = wv + 4 * ui;
        int tap = u >= WG_UNITS ? 13 : (u / CIS), cit = u >= WG_UNITS ? u - WG_UNITS : u % CIS;
        int dz = tap / 9, rem = tap - dz * 9, dyy = rem / 3, dx = rem - dyy * 3;
        uoff[ui] = ((dz * HY + dyy) * HX + (C::FLIP ? 0 : dx)) * C::PX + cit * 16 * C::ES;
    }
    // unit ui of wave wv handles tap wv + 4 ui, i.e. tap column dx = (wv + ui) % 3: the three column offsets are rotated
    // ONCE by the wave's phase, so unit ui simply takes entry ui % 3 (a compile-time index), and every per-lane address part
    // of the transposing reads -- lane row (g >> 1), 8-byte piece p, permuted column -- is summed here, outside the loops
    int xs0[3] = {0, 0, 0}, xs1[3] = {0, 0, 0}, xc0 = 0, xc1 = 0, ylane0 = 0, ylane1 = 0;
    if constexpr (C::FLIP) {
        const int q = c >> 2, p = c & 3;
        const int xl = (8 * g + q) & 15;
        const int lane_zy = (g >> 1) * HX * C::PX + 8 * p;             // the k-block's second voxel row for lane groups 2, 3
        int t0[3], t1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { t0[d] = lay_flip(xl + d) * C::PX + lane_zy; t1[d] = lay_flip(xl + 4 + d) * C::PX + lane_zy; }
        const int w3 = wv % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = (w3 + j) % 3;                 // wave-uniform
            xs0[j] = d == 0 ? t0[0] : (d == 1 ? t0[1] : t0[2]);
            xs1[j] = d == 0 ? t1[0] : (d == 1 ? t1[1] : t1[2]);
        }
        xc0 = t0[1]; xc1 = t1[1];                        // centre tap (the 1x1x1 units)
        ylane0 = lay_flip(8 * g + q) * C::PY + 8 * p;
        ylane1 = lay_flip(8 * g + q + 4) * C::PY + 8 * p;
    }
    f32x4 acc[NSL][WG_UPW];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
        for (int ui = 0; ui < WG_UPW; ++ui) acc[sl][ui] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // dy tile(s): 256 voxels x 16 channels
    constexpr int YCH = 16 / CH, YIT = NVOX * YCH / 256, NQ = CH / 4;
    auto load_dy = [&](const GT* __restrict__ src, long ld, int b, int z0, int y0, int x0, f32x4 (&buf)[YIT][NQ]) {
#pragma unroll
        for (int j = 0; j < YIT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / YCH, ch = id - v * YCH;
            const int gz = z0 + (v >> 6), gy = y0 + ((v >> 4) & 3), gx = x0 + (v & 15);
            const int cc = co0 + ch * CH;
            const bool ok = gz < D && gy < H && gx < W && cc < Cout;
            const GT* q = ok ? src + ((((long)b * D + gz) * H + gy) * W + gx) * ld + cc : src;   // branch-free
            if constexpr (CH == 8) {
                // bf16-stored gradient: one 16-byte load = the packed piece (bit pattern kept in buf[j][0])
                buf[j][0] = __builtin_bit_cast(f32x4, act_chunk<P>(q, ok));
            } else {
#pragma unroll
                for (int c4 = 0; c4 < NQ; ++c4) {
                    if constexpr (VECY) {
                        const bool okc = ok && cc + 4 * c4 + 4 <= Cout;
                        f32x4 t = *(const f32x4*)(okc ? q + 4 * c4 : src);
                        buf[j][c4] = okc ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool oke = ok && cc + 4 * c4 + e < Cout;
                            float t = *(oke ? q + 4 * c4 + e : src);
                            buf[j][c4][e] = oke ? t : 0.f;
                        }
                    }
                }
            }
        }
    };
    auto store_dy = [&](const f32x4 (&buf)[YIT][NQ], char* img) {
#pragma unroll
        for (int j = 0; j < YIT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / YCH, ch = id - v * YCH;
            if constexpr (CH == 8) {
                *(u32x4*)(img + (C::FLIP ? lay_flip(v) : v) * C::PY + ch * 16) = __builtin_bit_cast(u32x4, buf[j][0]);
            } else {
                float vals[CH];
#pragma unroll
                for (int c4 = 0; c4 < NQ; ++c4) { vals[4 * c4] = buf[j][c4][0]; vals[4 * c4 + 1] = buf[j][c4][1]; vals[4 * c4 + 2] = buf[j][c4][2]; vals[4 * c4 + 3] = buf[j][c4][3]; }
                *(u32x4*)(img + v * C::PY + ch * 16) = P::pack(vals);
            }
        }
    };
    // PIPE (16-channel slab only: the prefetch registers fit next to two resident workgroups per CU): the window and dy
    // tile of the NEXT tile are loaded into registers while the MFMAs of the current one run.
    constexpr bool PIPE = CIS == 1 && PIPE_OK;
    HaloRegs<P, PIPE ? 16 / CH : 1> R;
    f32x4 ybuf[YIT][NQ], y3buf[HAS3 ? YIT : 1][NQ];
    TileTable tt;
    int kt = 0, tx = 0, ty = 0, tz = 0, b = 0;
    if ((int)blockIdx.x < ntiles) tt.get(0, ntiles, ntx, nty, ntz, tx, ty, tz, b);
    // tile-invariant staging plans (see HaloPlan): window pieces of x, and the two 16-byte pieces of the dy tile per thread
    constexpr bool PLAN = PIPE && XMX == 2;
    HaloPlan<PLAN ? 16 / CH : 1> xplan;
    int dyrel[YIT], dyrel3[HAS3 ? YIT : 1], dylo[YIT];
    if constexpr (PLAN) {
        halo_plan<P, 16 / CH, C::LAY>(xplan, H, W, (int)ldx, C::PX);
#pragma unroll
        for (int j = 0; j < YIT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int v = id / YCH, ch = id - v * YCH;
            const int vz = v >> 6, vy = (v >> 4) & 3, vx = v & 15;
            dyrel[j] = ((vz * H + vy) * W + vx) * (int)lddy + co0 + ch * CH;
            if constexpr (HAS3) dyrel3[j] = ((vz * H + vy) * W + vx) * (int)lddy3 + co0 + ch * CH;
            dylo[j] = (C::FLIP ? lay_flip(v) : v) * C::PY + ch * 16;
        }
    }
    const long xitem = (long)D * H * W * ldx;
    // window + dy tile(s) of tile (b_, z_, y_, x_) into the prefetch registers
    auto tile_load = [&](int b_, int z_, int y_, int x_, int c0_ = -1, bool with_dy = true) {
        if (c0_ < 0) c0_ = ci0;
        if constexpr (PLAN) {
            halo_load_planned<P, 16 / CH>(R, xplan, (const uint16_t*)x + b_ * xitem, (int)ldx, z_, y_, x_, D, H, W, c0_, Cin);
            if (!with_dy) return;
            const bool interior = z_ + TZ <= D && y_ + TY <= H && x_ + TX <= W && co0 + 16 <= Cout;      // wave-uniform
            const long vb = (((long)b_ * D + z_) * H + y_) * W + x_;
            const GT* __restrict__ pdy = dy + vb * lddy;
            const GT* __restrict__ pdy3 = HAS3 ? dy3 + vb * lddy3 : nullptr;
#pragma unroll
            for (int j = 0; j < YIT; ++j) {
                bool ok = true;
                if (!interior) {
                    const int id = threadIdx.x + j * 256, v = id / YCH, ch = id - v * YCH;
                    ok = z_ + (v >> 6) < D && y_ + ((v >> 4) & 3) < H && x_ + (v & 15) < W && co0 + ch * CH < Cout;
                }
                ybuf[j][0] = __builtin_bit_cast(f32x4, act_chunk<P>(pdy + (ok ? dyrel[j] : 0), ok));
                if constexpr (HAS3) y3buf[j][0] = __builtin_bit_cast(f32x4, act_chunk<P>(pdy3 + (ok ? dyrel3[j] : 0), ok));
            }
        } else if constexpr (PIPE) {
            halo_load<P, 16 / CH, XMX>(R, x, ldx, b_, z_, y_, x_, D, H, W, ci0, Cin);
            load_dy(dy, lddy, b_, z_, y_, x_, ybuf);
            if constexpr (HAS3) load_dy(dy3, lddy3, b_, z_, y_, x_, y3buf);
        }
    };
    if constexpr (PIPE) {
        if ((int)blockIdx.x < ntiles) {
            tile_load(b, tz * TZ, ty * TY, tx * TX);
        }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++kt) {
        int ax = tx, ay = ty, az = tz, ab = b;            // the next tile's coordinates
        if (tile + (int)gridDim.x < ntiles) tt.get(kt + 1, ntiles, ntx, nty, ntz, ax, ay, az, ab);
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        __syncthreads();
        if constexpr (PIPE) {
            if constexpr (PLAN) {
                halo_store_planned<P, 16 / CH>(R, xplan, ximg);
#pragma unroll
                for (int j = 0; j < YIT; ++j) {
                    *(u32x4*)(yimg + dylo[j]) = __builtin_bit_cast(u32x4, ybuf[j][0]);
                    if constexpr (HAS3) *(u32x4*)(y3img + dylo[j]) = __builtin_bit_cast(u32x4, y3buf[j][0]);
                }
            } else {
                halo_store<P, 16 / CH, XMX, C::LAY>(R, C::PX, ximg);
                store_dy(ybuf, yimg);
                if constexpr (HAS3) store_dy(y3buf, y3img);
            }
            __syncthreads();
            const int nt = tile + gridDim.x;
            if constexpr (NSL == 2) tile_load(b, z0, y0, x0, ci0 + 16, false);      // this tile's second slab: window only
            else if (nt < ntiles) tile_load(ab, az * TZ, ay * TY, ax * TX);
        } else {
            stage_halo<P, 16 * CIS / CH, XMX, C::LAY>(x, ldx, b, z0, y0, x0, D, H, W, ci0, Cin, C::PX, ximg);
            load_dy(dy, lddy, b, z0, y0, x0, ybuf);
            store_dy(ybuf, yimg);
            if constexpr (HAS3) { load_dy(dy3, lddy3, b, z0, y0, x0, y3buf); store_dy(y3buf, y3img); }
            __syncthreads();
        }

        auto fast_compute = [&](auto slc) __attribute__((always_inline)) {
            constexpr int SLAB = decltype(slc)::value;
            // bf16, 16-channel slab, conflict-free layout: every per-lane address part is k-block independent (formed once before
            // the loop: ylane0/1, xs0/1[3]); per read one add of a wave-uniform (k-block, unit) offset.  The NKB x UPW (k-block,
            // unit) steps run as ONE straight line, software-pipelined: the x fragments of the next DPT steps and the dy fragment
            // of the next k-block are in flight while a step's MFMA runs (as a guarded loop every MFMA waited for its own two
            // reads: lgkmcnt(0) 56 times per tile).  A wave's unit beyond the last one (wave 3 without the 1x1x1 units) reads the
            // centre tap and accumulates into a register that is never written out.
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            constexpr int NST = C::NKB * WG_UPW, DPT = HAS3 ? 3 : 5;      // (fp32-stored x, the image: the guarded loop below -- its prefetch registers leave no room)
            auto aread = [&](int kb, const char* img, u32x4& af) {
                const int ykb = kb * 32 * C::PY;
                s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + ykb + ylane0));
                s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + ykb + ylane1));
                s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                af = __builtin_bit_cast(u32x4, a8);
            };
            auto bread = [&](int st, s16x4& lo, s16x4& hi) {
                const int kb = st / WG_UPW, ui = st % WG_UPW;
                const bool ext1 = HAS3 && ui == WG_UPW - 1 && wv + 4 * ui >= WG_UNITS;     // wave-uniform
                const int uo = ((kb >> 1) * HY + (kb & 1) * 2) * HX * C::PX + uoff[ui];    // window row of the k-block's first voxel row
                lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + uo + (ext1 ? xc0 : xs0[ui % 3])));
                hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + uo + (ext1 ? xc1 : xs1[ui % 3])));
            };
            s16x4 rlo[DPT], rhi[DPT];
            u32x4 af[2], af3 = {0u, 0u, 0u, 0u};          // dy fragment: this k-block's and the next one's; dy3: fetched two steps before its unit
            aread(0, yimg, af[0]);
#pragma unroll
            for (int st = 0; st < DPT; ++st) bread(st, rlo[st], rhi[st]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int kb = st / WG_UPW, ui = st % WG_UPW;
                if (ui == 0 && kb + 1 < C::NKB) aread(kb + 1, yimg, af[(kb + 1) & 1]);
                if constexpr (HAS3) { if (ui == WG_UPW - 3) aread(kb, y3img, af3); }
                const s16x4 blo = rlo[st % DPT], bhi = rhi[st % DPT];
                s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                const bool ext1 = HAS3 && ui == WG_UPW - 1 && wv + 4 * ui >= WG_UNITS;
                P::mma(acc[SLAB][ui], ext1 ? af3 : af[kb & 1], __builtin_bit_cast(u32x4, b8));
                if (st + DPT < NST) bread(st + DPT, rlo[st % DPT], rhi[st % DPT]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (CH == 8 && C::FLIP && XMX == 2) {
            fast_compute(IntC<0>{});
            if constexpr (NSL == 2) {
                // second slab of the same voxel tile: its window replaces the first one's (the dy images stay), and the next tile's
                // first-slab window + dy tiles go in flight under its MFMAs
                __syncthreads();
                halo_store_planned<P, 16 / CH>(R, xplan, ximg);
                __syncthreads();
                if (tile + (int)gridDim.x < ntiles) tile_load(ab, az * TZ, ay * TY, ax * TX);
                fast_compute(IntC<NSL - 1>{});
            }
        } else
        for (int kb = 0; kb < C::NKB; ++kb) {
            if constexpr (CH == 8 && C::FLIP) {
                // bf16, 16-channel slab, conflict-free layout: every per-lane address part is k-block independent (formed once
                // before the loop: ylane0/1, xs0/1[3]); per read one add of a wave-uniform (k-block, unit) offset
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const int ykb = kb * 32 * C::PY;
                s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(yimg + ykb + ylane0));
                s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(yimg + ykb + ylane1));
                s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                const u32x4 afrag = __builtin_bit_cast(u32x4, a8);
                u32x4 afrag3 = afrag;
                if constexpr (HAS3) {
                    s16x4 clo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + ykb + ylane0));
                    s16x4 chi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + ykb + ylane1));
                    s16x8 c8 = {clo[0], clo[1], clo[2], clo[3], chi[0], chi[1], chi[2], chi[3]};
                    afrag3 = __builtin_bit_cast(u32x4, c8);
                }
                const int xkb = ((kb >> 1) * HY + (kb & 1) * 2) * HX * C::PX;       // window row of the k-block's first voxel row
#pragma unroll
                for (int ui = 0; ui < WG_UPW; ++ui) {
                    if (wv + 4 * ui < nunits) {
                        const bool ext1 = HAS3 && ui == WG_UPW - 1 && wv + 4 * ui >= WG_UNITS;     // wave-uniform
                        const int uo = xkb + uoff[ui];
                        s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + uo + (ext1 ? xc0 : xs0[ui % 3])));
                        s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + uo + (ext1 ? xc1 : xs1[ui % 3])));
                        s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                        P::mma(acc[0][ui], ext1 ? afrag3 : afrag, __builtin_bit_cast(u32x4, b8));
                    }
                }
            } else if constexpr (CH == 8) {
                // bf16: k-block = 32 voxels; lane (c = 4q+p, g) addresses voxel rows 8g+q and 8g+4+q
                const int q = c >> 2, p = c & 3;
                const int v0 = kb * 32 + 8 * g + q, v1 = v0 + 4;
                LDS_AS s16x4* ya0 = (LDS_AS s16x4*)(yimg + v0 * C::PY + 8 * p);
                LDS_AS s16x4* ya1 = (LDS_AS s16x4*)(yimg + v1 * C::PY + 8 * p);
                s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ya0);
                s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ya1);
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                const u32x4 afrag = __builtin_bit_cast(u32x4, a8);
                u32x4 afrag3 = afrag;
                if constexpr (HAS3) {
                    s16x4 clo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + v0 * C::PY + 8 * p));
                    s16x4 chi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + v1 * C::PY + 8 * p));
                    s16x8 c8 = {clo[0], clo[1], clo[2], clo[3], chi[0], chi[1], chi[2], chi[3]};
                    afrag3 = __builtin_bit_cast(u32x4, c8);
                }
                const int h0 = (((v0 >> 6) * HY + ((v0 >> 4) & 3)) * HX + (v0 & 15)) * C::PX + 8 * p;
                const int h1 = (((v1 >> 6) * HY + ((v1 >> 4) & 3)) * HX + (v1 & 15)) * C::PX + 8 * p;
#pragma unroll
                for (int ui = 0; ui < WG_UPW; ++ui) {
                    if (wv + 4 * ui < nunits) {
                        s16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + h0 + uoff[ui]));
                        s16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(ximg + h1 + uoff[ui]));
                        s16x8 b8 = {blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
                        P::mma(acc[0][ui], (ui == WG_UPW - 1 && wv + 4 * ui >= WG_UNITS) ? afrag3 : afrag, __builtin_bit_cast(u32x4, b8));
                    }
                }
            } else {
                // 4-byte elements (fp32, or the split words of PrecBF16x3): k-block = 16 voxels; lane (c, g) holds voxels 4g+t for
                // channel c -- four element reads form one operand chunk (PrecF32::mma is exactly the four K = 4 MFMAs)
                u32x4 av, av3 = {0u, 0u, 0u, 0u};
                int hb[4];
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    int v = kb * 16 + 4 * g + tt;
                    av[tt] = *(const uint32_t*)(yimg + v * C::PY + c * 4);
                    if constexpr (HAS3) av3[tt] = *(const uint32_t*)(y3img + v * C::PY + c * 4);
                    hb[tt] = (((v >> 6) * HY + ((v >> 4) & 3)) * HX + (v & 15)) * C::PX + c * 4;
                }
#pragma unroll
                for (int ui = 0; ui < WG_UPW; ++ui) {
                    if (wv + 4 * ui < nunits) {
                        const bool ext = (ui == WG_UPW - 1 && wv + 4 * ui >= WG_UNITS);
                        u32x4 bv;
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) bv[tt] = *(const uint32_t*)(ximg + hb[tt] + uoff[ui]);
                        P::mma(acc[0][ui], ext ? av3 : av, bv);
                    }
                }
            }
        }
        tx = ax; ty = ay; tz = az; b = ab;
    }
    // partial sums: part[blockIdx.x][co][ci][tap]
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
    for (int ui = 0; ui < WG_UPW; ++ui) {
        int u = wv + 4 * ui;
        if (u < nunits) {
            int tap = u / CIS, cit = u >= WG_UNITS ? u - WG_UNITS : u % CIS;
            int ci = ci0 + sl * 16 + cit * 16 + c;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                int co = co0 + 4 * g + rr;
                if (ci < Cin && co < Cout) {
                    if (u < WG_UNITS) part[(((long)blockIdx.x * Cout + co) * Cin + ci) * 27 + tap] = acc[sl][ui][rr];
                    else part3[((long)blockIdx.x * Cout + co) * Cin + ci] = acc[sl][ui][rr];
                }
            }
        }
    }
}

// ---- bf16x3 weight gradient on (hi, lo) bf16 images ---------------------------------------------------------------------------
// The 4-byte fragment path above (fp32 and bf16x3 modes) feeds every MFMA operand with four ds_read_b32 per lane: 8 LDS instructions
// per 2 MFMAs saturate the LDS pipe (366 us for a 96^3 16-channel layer against 105 us for the bf16 kernel).  Voxel-contiguous b128
// fragments are not available -- the 27 taps shift the window along every axis, and a misaligned ds_read_b128 is replayed at 64
// cycles -- but the transposing 16-bit read is: this kernel splits the fp32-stored window and gradient tiles into TWO bf16 images
// each while staging (hi = bf16(v), lo = bf16(v - hi)), both in the conflict-free layout of the bf16 kernel (LAY 2, 32-byte voxels),
// and runs the bf16 kernel's software-pipelined (k-block, unit) line with three MFMAs per step,
//     dw += dy_hi x_hi + dy_hi x_lo + dy_lo x_hi        (v_mfma_f32_16x16x32_bf16, fp32 accumulate; the lo-lo term is 2^-16 of the sum)
// four ds_read_b64_tr_b16 per 3 MFMAs over 32 voxels instead of eight ds_read_b32 per 4 MFMAs.  16-channel input slab, 16-channel
// output tile, one partial row per workgroup -- grid and partial layout of conv3_wgrad_kernel<P, ., ., HAS3, 1, .>.
// XS: fewer than four input channels (the image in front of encoder1): the window is read channel by channel.
template <bool HAS3, bool XS = false>
__global__ void __launch_bounds__(256, 2)
conv3_wgrad_x3_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ dy, long lddy, float* __restrict__ part,
                      const float* __restrict__ dy3, long lddy3, float* __restrict__ part3,
                      int D, int H, int W, int Cin, int Cout, int ntx, int nty, int ntz, int ntiles) {
    constexpr int PX = 32, PY = 32, XB = NHALO * PX, YB = NVOX * PY;
    constexpr int NKB = NVOX / 32, NU = 27, NUX = NU + (HAS3 ? 1 : 0), UPW = (NUX + 3) / 4;
    __shared__ __attribute__((aligned(16))) char lds[2 * XB + (HAS3 ? 4 : 2) * YB];
    char* const xhi = lds;
    char* const xlo = lds + XB;
    char* const yhi = lds + 2 * XB;
    char* const ylo = yhi + YB;
    char* const y3hi = ylo + YB;
    char* const y3lo = y3hi + YB;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const int ci0 = blockIdx.y * 16, co0 = blockIdx.z * 16;
    // per-unit window offsets (wave-uniform) and the per-lane parts of the transposing reads: as in conv3_wgrad_kernel's FLIP path
    int uoff[UPW];
#pragma unroll
    for (int ui = 0; ui < UPW; ++ui) {
        const int u = wv + 4 * ui;
        const int tap = u >= NU ? 13 : u;
        const int dz = tap / 9, rem = tap - dz * 9, dyy = rem / 3;
        uoff[ui] = ((dz * HY + dyy) * HX) * PX;
    }
    int xs0[3], xs1[3], xc0, xc1, ylane0, ylane1;
    {
        const int q = c >> 2, p = c & 3;
        const int xl = (8 * g + q) & 15;
        const int lane_zy = (g >> 1) * HX * PX + 8 * p;
        int t0[3], t1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { t0[d] = lay_flip(xl + d) * PX + lane_zy; t1[d] = lay_flip(xl + 4 + d) * PX + lane_zy; }
        const int w3 = wv % 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = (w3 + j) % 3;
            xs0[j] = d == 0 ? t0[0] : (d == 1 ? t0[1] : t0[2]);
            xs1[j] = d == 0 ? t1[0] : (d == 1 ? t1[1] : t1[2]);
        }
        xc0 = t0[1]; xc1 = t1[1];
        ylane0 = lay_flip(8 * g + q) * PY + 8 * p;
        ylane1 = lay_flip(8 * g + q + 4) * PY + 8 * p;
    }
    f32x4 acc[UPW];
#pragma unroll
    for (int ui = 0; ui < UPW; ++ui) acc[ui] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // four fp32 channels -> 8 bytes of the hi image and 8 bytes of the lo image
    auto put = [&](f32x4 v, char* hi, char* lo, int off) __attribute__((always_inline)) {
        const bf16x4 h = __builtin_convertvector(v, bf16x4);
#ifdef UNETR_X3_DROP_LO
        const bf16x4 l = __builtin_convertvector((f32x4){0.f, 0.f, 0.f, 0.f}, bf16x4);
#else
        const bf16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
#endif
        *(bf16x4*)(hi + off) = h;
        *(bf16x4*)(lo + off) = l;
    };
    constexpr int XIT = (NHALO * 4 + 255) / 256, SB = 6;       // window: 648 voxels x 4 quads of channels; SB loads in flight
    TileTable tt;
    int kt = 0, tx = 0, ty = 0, tz = 0, b = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++kt) {
        tt.get(kt, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        const int x0 = tx * TX, y0 = ty * TY, z0 = tz * TZ;
        __syncthreads();                                       // the previous tile's fragment reads are done
        // dy (and dy3) tile: 256 voxels x 4 quads, one per thread and quarter
        {
            f32x4 yb[4], y3b[HAS3 ? 4 : 1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = threadIdx.x + j * 256, v = id >> 2, qd = id & 3;
                const int gz = z0 + (v >> 6), gy = y0 + ((v >> 4) & 3), gx = x0 + (v & 15), cc = co0 + 4 * qd;
                const bool ok = gz < D && gy < H && gx < W && cc < Cout;
                const long vox = (((long)b * D + gz) * H + gy) * W + gx;
                const f32x4 t = *(const f32x4*)(ok ? dy + vox * lddy + cc : dy);
                yb[j] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (HAS3) {
                    const f32x4 t3 = *(const f32x4*)(ok ? dy3 + vox * lddy3 + cc : dy3);
                    y3b[j] = ok ? t3 : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = threadIdx.x + j * 256, v = id >> 2, qd = id & 3;
                const int off = lay_flip(v) * PY + qd * 8;
                put(yb[j], yhi, ylo, off);
                if constexpr (HAS3) put(y3b[j], y3hi, y3lo, off);
            }
        }
        // window
        for (int it0 = 0; it0 < XIT; it0 += SB) {
            f32x4 xb[SB];
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int id = threadIdx.x + (it0 + j) * 256, hv = id >> 2, qd = id & 3;
                const int hz = hv / (HY * HX), rem = hv - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1, cc = ci0 + 4 * qd;
                const bool ok = (it0 + j < XIT) && hv < NHALO && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                                (unsigned)gx < (unsigned)W && cc < Cin;
                if constexpr (XS) {
                    const float* q = ok ? x + ((((long)b * D + gz) * H + gy) * W + gx) * ldx + cc : x;
                    f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 3; ++e) { const bool oke = ok && cc + e < Cin; const float u = *(oke ? q + e : x); t[e] = oke ? u : 0.f; }
                    xb[j] = t;
                } else {
                    const f32x4 t = *(const f32x4*)(ok ? x + ((((long)b * D + gz) * H + gy) * W + gx) * ldx + cc : x);
                    xb[j] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int id = threadIdx.x + (it0 + j) * 256, hv = id >> 2, qd = id & 3;
                if (it0 + j < XIT && hv < NHALO) {
                    const int hx = hv % HX;
                    put(xb[j], xhi, xlo, (hv - hx + lay_flip(hx)) * PX + qd * 8);
                }
            }
        }
        __syncthreads();
        // the NKB x UPW (k-block, unit) steps as one software-pipelined straight line (see conv3_wgrad_kernel's fast path)
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        constexpr int NST = NKB * UPW, DPT = 3;
        auto tr2 = [&](const char* img, int o0, int o1) __attribute__((always_inline)) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + o0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(img + o1));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(u32x4, v);
        };
        auto bread = [&](int st, u32x4& bh, u32x4& bl) __attribute__((always_inline)) {
            const int kb = st / UPW, ui = st % UPW;
            const bool ext1 = HAS3 && ui == UPW - 1 && wv + 4 * ui >= NU;     // wave-uniform
            const int uo = ((kb >> 1) * HY + (kb & 1) * 2) * HX * PX + uoff[ui];
            const int o0 = uo + (ext1 ? xc0 : xs0[ui % 3]), o1 = uo + (ext1 ? xc1 : xs1[ui % 3]);
            bh = tr2(xhi, o0, o1);
            bl = tr2(xlo, o0, o1);
        };
        u32x4 rbh[DPT], rbl[DPT];
        u32x4 ah[2], al[2], a3h = {0u, 0u, 0u, 0u}, a3l = {0u, 0u, 0u, 0u};
        ah[0] = tr2(yhi, ylane0, ylane1);
        al[0] = tr2(ylo, ylane0, ylane1);
#pragma unroll
        for (int st = 0; st < DPT; ++st) bread(st, rbh[st], rbl[st]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            const int kb = st / UPW, ui = st % UPW;
            if (ui == 0 && kb + 1 < NKB) {
                const int ykb = (kb + 1) * 32 * PY;
                ah[(kb + 1) & 1] = tr2(yhi, ykb + ylane0, ykb + ylane1);
                al[(kb + 1) & 1] = tr2(ylo, ykb + ylane0, ykb + ylane1);
            }
            if constexpr (HAS3) {
                if (ui == UPW - 3) { const int ykb = kb * 32 * PY; a3h = tr2(y3hi, ykb + ylane0, ykb + ylane1); a3l = tr2(y3lo, ykb + ylane0, ykb + ylane1); }
            }
            const u32x4 bh = rbh[st % DPT], bl = rbl[st % DPT];
            const bool ext1 = HAS3 && ui == UPW - 1 && wv + 4 * ui >= NU;
            const u32x4 fh = ext1 ? a3h : ah[kb & 1], fl = ext1 ? a3l : al[kb & 1];
            PrecBF16::mma(acc[ui], fh, bh);
            PrecBF16::mma(acc[ui], fh, bl);
            PrecBF16::mma(acc[ui], fl, bh);
            if (st + DPT < NST) bread(st + DPT, rbh[st % DPT], rbl[st % DPT]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // partial sums: part[blockIdx.x][co][ci][tap]
#pragma unroll
    for (int ui = 0; ui < UPW; ++ui) {
        const int u = wv + 4 * ui;
        if (u < NUX) {
            const int ci = ci0 + c;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int co = co0 + 4 * g + rr;
                if (ci < Cin && co < Cout) {
                    if (u < NU) part[(((long)blockIdx.x * Cout + co) * Cin + ci) * 27 + u] = acc[ui][rr];
                    else part3[((long)blockIdx.x * Cout + co) * Cin + ci] = acc[ui][rr];
                }
            }
        }
    }
}

// Weight gradient of the same single-input-channel conv (+ its 1x1x1 branch): dw[co][tap] = sum_v dy[v, co] x[v + off(tap)],
// dw3[co] = sum_v dy3[v, co] x[v].  The generic kernel contracts over voxels with the window zero-padded to 16 channels (70 us
// at 96^3 for 114 MB of gradient maps); here the tap is the OUTPUT column: per 32-voxel k-block two MFMAs [16 co x 32 voxels] x
// [32 voxels x 16 taps] (taps 0-15 and 16-31; column 27 = the centre voxel again, which the dy3 product reads its result from).
// dy / dy3 tiles (256 voxels x 16 channels, bf16) are staged in the conflict-free transposing-read layout of conv3_wgrad_kernel
// (lay_flip), the image window as bf16 scalars; each wave owns two of the eight k-blocks of a tile and keeps its sums in
// registers across its workgroup's tile walk; one partial row per workgroup: part[workgroup][16][27], part3[workgroup][16].
__global__ void __launch_bounds__(256, 4)
conv3_c1_wgrad_kernel(const float* __restrict__ x, const uint16_t* __restrict__ dy, long lddy, const uint16_t* __restrict__ dy3, long lddy3,
                      float* __restrict__ part, float* __restrict__ part3, int D, int H, int W, int ntx, int nty, int ntz, int ntiles) {
    constexpr int PY = 32, NPT = (NHALO + 255) / 256, WINB = ((NHALO + 8) * 2 + 15) / 16 * 16;
    __shared__ __attribute__((aligned(16))) char lds[WINB + 2 * NVOX * PY];
    uint16_t* win = (uint16_t*)lds;
    char* yimg = lds + WINB;
    char* y3img = yimg + NVOX * PY;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const bool has3 = dy3 != nullptr;
    const int q = c >> 2, p = c & 3;
    const int ylane0 = lay_flip(8 * g + q) * PY + 8 * p, ylane1 = lay_flip(8 * g + q + 4) * PY + 8 * p;
    // window element offset of this lane's column (tap 16 tt + c; taps >= 27 alias the centre) + its voxel row / half row
    int toff[2];
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        const int tap = 16 * t2 + c < 27 ? 16 * t2 + c : 13;
        toff[t2] = ((tap / 9) * HY + (tap % 9) / 3) * HX + tap % 3 + (g >> 1) * HX + 8 * (g & 1);
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acc3 = {0.f, 0.f, 0.f, 0.f};
    const long item = (long)D * H * W;
    float nxt[NPT];
    u32x4 ybuf[2], y3buf[2];
    auto tload = [&](int b_, int z_, int y_, int x_) {
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            const int hz = id / (HY * HX), rem = id - hz * (HY * HX), hy = rem / HX, hx = rem - hy * HX;
            const int gz = z_ - 1 + hz, gy = y_ - 1 + hy, gx = x_ - 1 + hx;
            const bool ok = id < NHALO && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            float t = 0.f;
            if (ok) t = x[b_ * item + ((long)gz * H + gy) * W + gx];
            nxt[j] = t;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = threadIdx.x + j * 256, v = id >> 1, ch = id & 1;
            const int gz = z_ + (v >> 6), gy = y_ + ((v >> 4) & 3), gx = x_ + (v & 15);
            const bool ok = gz < D && gy < H && gx < W;
            const long vox = ok ? ((((long)b_ * D + gz) * H + gy) * W + gx) : 0;
            ybuf[j] = act_chunk<PrecBF16>(dy + vox * lddy + ch * 8, ok);
            if (has3) y3buf[j] = act_chunk<PrecBF16>(dy3 + vox * lddy3 + ch * 8, ok);
        }
    };
    TileTable tt;
    int kt = 0, tx = 0, ty = 0, tz = 0, b = 0;
    if ((int)blockIdx.x < ntiles) {
        tt.get(0, ntiles, ntx, nty, ntz, tx, ty, tz, b);
        tload(b, tz * TZ, ty * TY, tx * TX);
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++kt) {
        int ax = tx, ay = ty, az = tz, ab = b;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const int id = threadIdx.x + j * 256;
            if (id < NHALO) win[id] = f2bf(nxt[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = threadIdx.x + j * 256, v = id >> 1, ch = id & 1;
            *(u32x4*)(yimg + lay_flip(v) * PY + ch * 16) = ybuf[j];
            if (has3) *(u32x4*)(y3img + lay_flip(v) * PY + ch * 16) = y3buf[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) {
            tt.get(kt + 1, ntiles, ntx, nty, ntz, ax, ay, az, ab);
            tload(ab, az * TZ, ay * TY, ax * TX);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kb = wv + 4 * h;                            // this wave's k-block: voxels 32 kb .. 32 kb + 31
            const int ykb = kb * 32 * PY;
            s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(yimg + ykb + ylane0));
            s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(yimg + ykb + ylane1));
            const s16x8 a8 = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            u32x4 af3 = {0u, 0u, 0u, 0u};
            if (has3) {
                s16x4 clo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + ykb + ylane0));
                s16x4 chi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(y3img + ykb + ylane1));
                const s16x8 c8 = {clo[0], clo[1], clo[2], clo[3], chi[0], chi[1], chi[2], chi[3]};
                af3 = __builtin_bit_cast(u32x4, c8);
            }
            const uint16_t* wb = win + ((kb >> 1) * HY + (kb & 1) * 2) * HX;      // window row of the k-block's first voxel row
            u32x4 bf[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                uint16_t b8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = wb[toff[t2] + j];
#pragma unroll
                for (int d = 0; d < 4; ++d) bf[t2][d] = (uint32_t)b8[2 * d] | ((uint32_t)b8[2 * d + 1] << 16);
            }
            PrecBF16::mma(acc[0], __builtin_bit_cast(u32x4, a8), bf[0]);
            PrecBF16::mma(acc[1], __builtin_bit_cast(u32x4, a8), bf[1]);
            if (has3) PrecBF16::mma(acc3, af3, bf[1]);
        }
        tx = ax; ty = ay; tz = az; b = ab;
    }
    // the four waves' sums are added through LDS (fixed order: wave 0 + 1 + 2 + 3): ONE partial row per workgroup
    __syncthreads();
    f32x4* red = (f32x4*)lds;                                // [3 accumulators][4 waves][64 lanes]
    red[(0 * 4 + wv) * 64 + lane] = acc[0];
    red[(1 * 4 + wv) * 64 + lane] = acc[1];
    red[(2 * 4 + wv) * 64 + lane] = acc3;
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            f32x4 t = red[(a * 4 + 0) * 64 + lane];
            t += red[(a * 4 + 1) * 64 + lane]; t += red[(a * 4 + 2) * 64 + lane]; t += red[(a * 4 + 3) * 64 + lane];
            if (a == 0) acc[0] = t; else if (a == 1) acc[1] = t; else acc3 = t;
        }
        // accumulator: lane (column c = tap within its tile, g) holds rows co = 4g .. 4g+3
        const long row = blockIdx.x;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int co = 4 * g + rr;
            part[(row * 16 + co) * 27 + c] = acc[0][rr];
            if (16 + c < 27) part[(row * 16 + co) * 27 + 16 + c] = acc[1][rr];
            if (has3 && c == 11) part3[row * 16 + co] = acc3[rr];
        }
    }
}

// dw[i] = sum_g part[g][i]: 32 outputs x 8 g-phases per workgroup, fixed summation order (reproducible); eight loads of
// a phase are in flight together (the slabs sit in L2: the pass is latency-, not bandwidth-bound)
__global__ void __launch_bounds__(256)
conv3_wgrad_reduce_kernel(const float* __restrict__ part, int G, long n, float* __restrict__ dw, int nb1,
                          const float* __restrict__ part_b, long n_b, float* __restrict__ dw_b) {
    // workgroups >= nb1: the second problem of the same launch (the 1x1x1 weight gradient of a fused residual-block front)
    int bx = blockIdx.x, nbx = nb1;
    if (bx >= nb1) { bx -= nb1; nbx = gridDim.x - nb1; part = part_b; n = n_b; dw = dw_b; }
    __shared__ float sm[8][33];
    const int o = threadIdx.x & 31, ph = threadIdx.x >> 5;
    for (long i0 = (long)bx * 32; i0 < n; i0 += (long)nbx * 32) {
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

// probe of the ds_read_b64_tr_b16 lane map (documented in cdna_hip_programming.md T10); used by a GPU test
__global__ void tr16_probe_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(lds + (grp * 4 + q) * 64 + 4 * p));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

// the pair layout is used for bf16 with <= 16 contraction channels (K = Cin forward, Cout for the data gradient)
template <class P> inline bool use_pair(int K) { return P::CH == 8 && K <= 16; }
inline int conv_pipe_enabled() {   // tuning hook: UNETR_CONV_PIPE=0 selects the one-tile-per-workgroup kernel
    const char* e = getenv("UNETR_CONV_PIPE");
    return e ? atoi(e) : 1;
}

// ---- all weight re-packs of a step in ONE launch (run by the optimizer right after the update) ---------------------------
// kind 0 / 1: 3x3x3 forward / data-gradient layout (pair layout when the contraction has <= 16 channels in bf16),
// kind 2: 1x1x1 forward (fused front), kind 3: 1x1x1 transposed (fused input gradient).
constexpr int PK_MAX = 64;
struct PkProblem { const float* w; void* out; int Cin, Cout, kind, pair, blk0; long total; int staged; };
struct PkArgs { int n; PkProblem p[PK_MAX]; };

template <class T>
__global__ void __launch_bounds__(256) conv3_pack_grouped_kernel(PkArgs a, int SL) {
    int pi = 0, hi_ = a.n - 1;                 // last problem whose first block <= blockIdx.x (binary search: a linear scan of the
    while (pi < hi_) {                         // kernel-argument table cost every block a chain of ~64 dependent scalar loads)
        const int mid = (pi + hi_ + 1) >> 1;
        if ((int)blockIdx.x >= a.p[mid].blk0) pi = mid; else hi_ = mid - 1;
    }
    const PkProblem& pr = a.p[pi];
    const float* __restrict__ w = pr.w;
    T* __restrict__ wp = (T*)pr.out;
    const int Cin = pr.Cin, Cout = pr.Cout;
    if (pr.staged) {
        // slab-mode 3x3x3 packs (the many-channel layers: 16 MB of fp32 weights per step): element by element the gather reads w
        // with a 108-byte stride (one useful value per cache line touched).  Here a block owns (slab of SL reduction channels,
        // 8 output rows n): it reads its 8 x SL x 27 source values as contiguous runs into LDS and writes, per tap, the 8 x SL
        // packed values as ONE contiguous run.
        __shared__ float tile[32 * 217];
        const int mode = pr.kind, K = mode ? Cout : Cin, N = mode ? Cin : Cout, nslab = (K + SL - 1) / SL;
        const int lb = (int)blockIdx.x - pr.blk0, slab = lb % nslab, n0 = (lb / nslab) * 8, k0 = slab * SL;
        const int kn = min(SL, K - k0), nn = min(8, N - n0);
        if (mode == 0) {           // w[n][k][tap]: per n a run of kn * 27 values; tile[n_i * 868 + kk * 27 + tap]
            for (int idx = threadIdx.x; idx < nn * kn * 27; idx += 256) {
                const int ni = idx / (kn * 27), rem = idx - ni * (kn * 27);
                tile[ni * 868 + rem] = w[((long)(n0 + ni) * Cin + k0) * 27 + rem];
            }
        } else {                   // w[k][n][tap]: per k a run of nn * 27 values; tile[kk * 217 + n_i * 27 + tap]
            for (int idx = threadIdx.x; idx < kn * nn * 27; idx += 256) {
                const int kk = idx / (nn * 27), rem = idx - kk * (nn * 27);
                tile[kk * 217 + rem] = w[((long)(k0 + kk) * Cin + n0) * 27 + rem];
            }
        }
        __syncthreads();
        const int kk = threadIdx.x % SL, ni = threadIdx.x / SL;      // (SL 32: 8 rows per pass; SL 16: 16 threads per row, 8 rows of the 16)
        if (ni < nn) {
            for (int tap = 0; tap < 27; ++tap) {
                float v = 0.f;
                if (kk < kn) v = mode ? tile[kk * 217 + ni * 27 + (26 - tap)] : tile[ni * 868 + kk * 27 + tap];
                wp[(((long)tap * nslab + slab) * N + n0 + ni) * SL + kk] = cvt_elem<T>(v);
            }
        }
        return;
    }
    // (32-bit index arithmetic: the host refuses packs of >= 2^31 elements; as 64-bit values every element paid several ~50-
    // instruction divisions)
    const unsigned base = ((unsigned)blockIdx.x - (unsigned)pr.blk0) * 2048u;
    for (int u = 0; u < 8; ++u) {
        const unsigned i = base + threadIdx.x + u * 256;
        if (i >= (unsigned)pr.total) break;
        float v = 0.f;
        if (pr.kind == 4) {
            // 2x2x2 transposed conv, tap-major GEMM operand: out[ci][tap][co] = w[ci][co][tap] (torch ConvTranspose3d layout
            // [in, out, 2, 2, 2]); the problem's Cout field carries the layer's INPUT channels (dim 0 of w), Cin its output channels
            const unsigned tco = Cin, co = i % tco, t = i / tco, tap = t & 7, ci = t >> 3;
            v = w[((long)ci * tco + co) * 8 + tap];
        } else if (pr.kind <= 1) {
            const int mode = pr.kind, K = mode ? Cout : Cin, N = mode ? Cin : Cout;
            if (pr.pair) {
                const int kk = (int)(i & 31); const unsigned t = i >> 5; const int n = (int)(t % (unsigned)N), tp = (int)(t / (unsigned)N);
                const int tap = 2 * tp + (kk >> 4), k = kk & 15;
                if (tap < 27 && k < K) v = mode ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
            } else {
                const int nslab = (K + SL - 1) / SL;
                const int kk = (int)(i % (unsigned)SL); unsigned t = i / (unsigned)SL; const int n = (int)(t % (unsigned)N); t /= (unsigned)N;
                const int slab = (int)(t % (unsigned)nslab), tap = (int)(t / (unsigned)nslab);
                const int k = slab * SL + kk;
                if (k < K) v = mode ? w[((long)k * Cin + n) * 27 + (26 - tap)] : w[((long)n * Cin + k) * 27 + tap];
            }
        } else {
            // 1x1x1: rows n, contraction k.  kind 2: n = co, k = ci, element w3[n][k];  kind 3: n = ci, k = co, element w3[k][n]
            const int K = pr.kind == 2 ? Cin : Cout, N = pr.kind == 2 ? Cout : Cin, RW = pr.pair ? 32 : SL;
            const int kk = (int)(i % (unsigned)RW); const unsigned t = i / (unsigned)RW; const int n = (int)(t % (unsigned)N), slab = (int)(t / (unsigned)N);
            const int k = pr.pair ? kk - 16 : slab * SL + kk;
            if (k >= 0 && k < K) v = pr.kind == 2 ? w[(long)n * K + k] : w[(long)k * N + n];
        }
        wp[i] = cvt_elem<T>(v);
    }
}

template <class P>
int pack_t(const float* w, void* wp, int Cin, int Cout, int mode, hipStream_t st) {
    typedef typename ElemOf<P>::type T;
    const int SL = 4 * P::CH, K = mode ? Cout : Cin, N = mode ? Cin : Cout;
    if (use_pair<P>(K) && conv_pipe_enabled()) {
        long total = 14L * N * 32;
        hipLaunchKernelGGL(conv3_pack_pair_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, w,
                           (uint16_t*)wp, Cin, Cout, mode);
        return unetr_check_launch();
    }
    long total = 27L * ((K + SL - 1) / SL) * N * SL;
    int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL((conv3_pack_kernel<T>), dim3(blocks), dim3(256), 0, st, w, (T*)wp, Cin, Cout, mode, SL);
    return unetr_check_launch();
}

// k3 > 0: FUSE 4 (y3 = second input); bst: FUSE 5 (y3 = the pre-norm tensor xn, wp3 = its statistics, part = backward-statistics rows)
struct FuseArgs { float* part; const void* wp3; void* y3; long ldy3; float* part3; int rows; int k3; int bst = 0; };

// x_f32: the input tensor is fp32 even in bf16 mode (the image in front of encoder1: <= 16 channels, pair layout only)
template <class P>
int fwd_t(const void* x, long ldx, const void* wp, void* yv, long ldy, int accumulate, int B, int D, int H, int W, int Cin, int Cout,
          hipStream_t st, FuseArgs* fz = nullptr, int x_f32 = 0) {
    typedef typename ActOf<P>::type YT;
    YT* y = (YT*)yv;
    constexpr bool B16 = P::CH == 8;
    const int ntx = cdiv(W, TX), nty = cdiv(H, TY), ntz = cdiv(D, TZ);
    const long spatial = (long)B * ntx * nty * ntz;
    const int ntn = Cout / 16;
    int ntb = 1;
    if ((ntn & (ntn - 1)) == 0) {  // power of two
        ntb = std::min(ntn, 8);
        while (ntb > 1 && spatial * (ntn / ntb) < 512) ntb >>= 1;
    }
    int xm = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0 && (Cin & 3) == 0) ? 1 : 0;
    if (B16 && !x_f32) {
        if (((uintptr_t)x & 15) || (ldx & 7) || (Cin & 7)) return UNETR_ERR_UNSUPPORTED;     // bf16 rows: whole 16-byte pieces
        xm = 2;
    }
    if (B16 && fz && fz->k3 > 0 && ((fz->ldy3 & 7) || ((uintptr_t)fz->y3 & 15) || (fz->k3 & 7))) return UNETR_ERR_UNSUPPORTED;
    if ((long)D * H * W * ldx >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;      // 32-bit in-item offsets (halo_load)
    if constexpr (B16) {
        // the single-channel image -> 16 channels with fused statistics (and the 1x1x1 branch): the dedicated one-MFMA kernel
        if (conv_pipe_enabled() && x_f32 && Cin == 1 && Cout == 16 && ldx == 1 && fz && fz->k3 == 0 && !accumulate && use_pair<P>(Cin) &&
            ntx < 256 && nty < 256 && ntz < 256 && B < 256 && (ldy & 3) == 0 && ((uintptr_t)y & 7) == 0 && (long)D * H * W < (1L << 31) &&
            (!fz->y3 || (((uintptr_t)fz->y3 & 7) == 0 && fz->ldy3 == ldy)) && !getenv("UNETR_CONV_C1_OFF")) {
            long cap = 1024;
            if (const char* e = getenv("UNETR_TEST_MAX_WG")) { if (atoi(e) > 0) cap = std::min<long>(cap, atoi(e)); }
            if (cap < spatial) cap = std::max<long>(8, cap / 8 * 8);
            const unsigned gx = (unsigned)std::min<long>(spatial, cap);
            fz->rows = (int)gx;
            hipLaunchKernelGGL(conv3_c1_fwd_kernel, dim3(gx), dim3(256), 0, st, (const float*)x, (const uint16_t*)wp, (uint16_t*)y, ldy, D, H, W,
                               ntx, nty, ntz, (int)spatial, fz->part, (const uint16_t*)fz->wp3, (uint16_t*)fz->y3, fz->part3);
            return unetr_check_launch();
        }
    }
    if constexpr (std::is_same<P, PrecBF16x3>::value) {
        // bf16x3: the same single-channel form on split operands
        if (conv_pipe_enabled() && Cin == 1 && Cout == 16 && ldx == 1 && fz && fz->k3 == 0 && !accumulate && ntx < 256 && nty < 256 && ntz < 256 &&
            B < 256 && (ldy & 3) == 0 && ((uintptr_t)y & 15) == 0 && (long)D * H * W < (1L << 31) &&
            (!fz->y3 || (((uintptr_t)fz->y3 & 15) == 0 && fz->ldy3 == ldy)) && !getenv("UNETR_CONV_C1_OFF")) {
            long cap = 1024;
            if (const char* e = getenv("UNETR_TEST_MAX_WG")) { if (atoi(e) > 0) cap = std::min<long>(cap, atoi(e)); }
            if (cap < spatial) cap = std::max<long>(8, cap / 8 * 8);
            const unsigned gx = (unsigned)std::min<long>(spatial, cap);
            fz->rows = (int)gx;
            hipLaunchKernelGGL(conv3_c1_fwd_x3_kernel, dim3(gx), dim3(256), 0, st, (const float*)x, (const uint32_t*)wp, (float*)y, ldy, D, H, W,
                               ntx, nty, ntz, (int)spatial, fz->part, (const uint32_t*)fz->wp3, (float*)fz->y3, fz->part3);
            return unetr_check_launch();
        }
    }
    if (conv_pipe_enabled() && ntb <= 4 && ntx < 256 && nty < 256 && ntz < 256 && B < 256) {      // (TileTable packs the coordinates in bytes)
        // persistent, software-pipelined kernel: a few resident workgroups per CU walk the tiles
        const bool pair = use_pair<P>(Cin);
        if (B16 && xm != 2 && !pair) return UNETR_ERR_UNSUPPORTED;   // fp32 input in bf16 mode = the image: pair layout only
        // the transposed accumulator tile is stored four channels (8 / 16 bytes) per lane
        if ((ldy & 3) || ((uintptr_t)y & (B16 ? 7 : 15)) || (fz && fz->k3 == 0 && fz->y3 && ((uintptr_t)fz->y3 & (B16 ? 7 : 15)))) return UNETR_ERR_UNSUPPORTED;
        if (pair && ntb > 2) ntb = 2;
        // slab mode: weights resident in LDS (see the kernel); two workgroups per CU only while a workgroup stays under 80 KB
        const int nslab = cdiv(Cin, 4 * P::CH);
        int wl = pair ? 0 : (nslab == 1 ? 1 : 2);
        if (!pair && NHALO * 64 + wl * 27 * ntb * 1024 > 160 * 1024) wl = 0;
        const bool two = pair || NHALO * 64 + wl * 27 * ntb * 1024 <= 80 * 1024;
        long cap = two ? 512 : 256;   // resident workgroups (VGPR / LDS-limited); more would queue behind them
        if (const char* e = getenv("UNETR_CONV_CAP")) { if (pair && atoi(e) > 0) cap = atoi(e); }      // tuning hook (pair layout)
        if (const char* e = getenv("UNETR_CONV_CAP_SLAB")) { if (!pair && two && atoi(e) > 0) cap = atoi(e); }
        if (const char* e = getenv("UNETR_TEST_MAX_WG")) { if (atoi(e) > 0) cap = std::min<long>(cap, atoi(e)); }   // test hook: long tile walks
        // a workgroup that walks several tiles must see them in non-decreasing batch order (the fused statistics are flushed when
        // the batch item changes): tile_coords guarantees that for grids that are multiples of the 8 XCDs
        if (cap < spatial) cap = std::max<long>(8, cap / 8 * 8);
        dim3 pgrid((unsigned)std::min<long>(spatial, cap), ntn / ntb);
#define LAUNCH_PIPE_F(NTB_, PAIR_, XM_, FUSE_, WL_)                                                                               \
    hipLaunchKernelGGL((conv3_fwd_pipe_kernel<P, NTB_, PAIR_, XM_, FUSE_, WL_>), pgrid, dim3(256), 0, st, x, ldx, (const char*)wp, y, ldy, \
                       accumulate, D, H, W, Cin, Cout, ntx, nty, ntz, (int)spatial, fz ? fz->part : nullptr,                       \
                       fz ? (const char*)fz->wp3 : nullptr, fz ? (YT*)fz->y3 : nullptr, fz ? fz->ldy3 : 0, fz ? fz->part3 : nullptr,       \
                       fz ? fz->k3 : 0)
#define LAUNCH_PIPE_V(NTB_, PAIR_, FUSE_, WL_)                                                                                    \
    do {                                                                                                                          \
        if constexpr (B16) {                                                                                                      \
            if (xm == 2) { LAUNCH_PIPE_F(NTB_, PAIR_, 2, FUSE_, WL_); break; }                                                    \
            if constexpr (PAIR_) { if (xm == 1) LAUNCH_PIPE_F(NTB_, true, 1, FUSE_, 0); else LAUNCH_PIPE_F(NTB_, true, 0, FUSE_, 0); }  \
        } else {                                                                                                                  \
            if (xm == 1) LAUNCH_PIPE_F(NTB_, PAIR_, 1, FUSE_, WL_); else LAUNCH_PIPE_F(NTB_, PAIR_, 0, FUSE_, WL_);               \
        }                                                                                                                         \
    } while (0)
#define LAUNCH_PIPE_W(NTB_, PAIR_, WL_)                                                                                           \
    do {                                                                                                                          \
        if (fz && fz->k3 > 0) { LAUNCH_PIPE_V(NTB_, PAIR_, 4, WL_); break; }                                                      \
        if (fz && fz->bst) { LAUNCH_PIPE_V(NTB_, PAIR_, 5, WL_); break; }                                                         \
        if constexpr (!(PAIR_)) {                                                                                                 \
            if (fz && fz->wp3 && late1x1) { LAUNCH_PIPE_V(NTB_, false, 3, WL_); break; }                                          \
        }                                                                                                                         \
        if (fz && fz->wp3) LAUNCH_PIPE_V(NTB_, PAIR_, 2, WL_);                                                                    \
        else if (fz) LAUNCH_PIPE_V(NTB_, PAIR_, 1, WL_);                                                                          \
        else LAUNCH_PIPE_V(NTB_, PAIR_, 0, WL_);                                                                                  \
    } while (0)
// slab mode: NTB 1 / 2 always keep the weights in LDS (one or two images); NTB 4 only fits one
#define LAUNCH_PIPE(NTB_, PAIR_)                                                                                                  \
    do {                                                                                                                          \
        if constexpr (PAIR_) LAUNCH_PIPE_W(NTB_, true, 0);                                                                        \
        else if constexpr (NTB_ == 4) { if (wl == 1) LAUNCH_PIPE_W(4, false, 1); else LAUNCH_PIPE_W(4, false, 0); }                \
        else { if (wl == 1) LAUNCH_PIPE_W(NTB_, false, 1); else LAUNCH_PIPE_W(NTB_, false, 2); }                                   \
    } while (0)
        const bool late1x1 = Cin <= 4 * P::CH;      // single-slab window: the 1x1x1 product is formed after the tile (FUSE 3)
        if (fz && fz->k3 == 0) fz->rows = (int)pgrid.x;          // one partial row per workgroup and batch item, zeroed in-kernel
        if constexpr (P::CH == 8) {
            if (pair) {
                if (ntb == 1) LAUNCH_PIPE(1, true); else LAUNCH_PIPE(2, true);
                return unetr_check_launch();
            }
        }
        switch (ntb) {
            case 1: LAUNCH_PIPE(1, false); break;
            case 2: LAUNCH_PIPE(2, false); break;
            default: LAUNCH_PIPE(4, false); break;
        }
        return unetr_check_launch();
    }
    if (fz) return UNETR_ERR_UNSUPPORTED;       // fused statistics / 1x1 need the persistent kernel
    dim3 grid((unsigned)spatial, ntn / ntb);
    if (B16 && xm != 2) return UNETR_ERR_UNSUPPORTED;
#define LAUNCH_FWD_X(NTB_, XM_) hipLaunchKernelGGL((conv3_fwd_kernel<P, NTB_, XM_>), grid, dim3(256), 0, st, x, ldx, (const char*)wp, y, ldy, \
                                                   accumulate, D, H, W, Cin, Cout, ntx, nty, ntz)
#define LAUNCH_FWD(NTB_)                                                                                                          \
    do {                                                                                                                          \
        if constexpr (B16) LAUNCH_FWD_X(NTB_, 2);                                                                                 \
        else { if (xm == 1) LAUNCH_FWD_X(NTB_, 1); else LAUNCH_FWD_X(NTB_, 0); }                                                  \
    } while (0)
    switch (ntb) {
        case 1: LAUNCH_FWD(1); break;
        case 2: LAUNCH_FWD(2); break;
        case 4: LAUNCH_FWD(4); break;
        case 8: LAUNCH_FWD(8); break;
        default: return UNETR_ERR_UNSUPPORTED;
    }
    return unetr_check_launch();
}

// rows_only != nullptr: just report the number of partial rows (workgroups along the voxel walk) a launch of this shape uses with an
// unlimited partial buffer; parts_only: the partial rows stay in ws ([G][n] then [G][n3]) and NO reduce is launched -- the caller
// reduces them later (unetr_reduce_rows_grouped: every weight-gradient reduction of a backward pass in one launch)
template <class P>
int wgrad_t(const void* x, long ldx, const void* dyv, long lddy, float* dw, const void* dy3v, long lddy3, float* dw3,
            int B, int D, int H, int W, int Cin, int Cout, float* ws, size_t ws_bytes, hipStream_t st, int x_f32 = 0,
            bool parts_only = false, long* rows_only = nullptr, long* rows_used = nullptr) {
    typedef typename ActOf<P>::type GT;
    const GT* dy = (const GT*)dyv;
    const GT* dy3 = (const GT*)dy3v;
    constexpr bool B16 = P::CH == 8;
    const int ntx = cdiv(W, TX), nty = cdiv(H, TY), ntz = cdiv(D, TZ);
    const long ntiles = (long)B * ntx * nty * ntz;
    if ((long)D * H * W * ldx >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;      // 32-bit in-item offsets (halo_load)
    if (ntx > 255 || nty > 255 || ntz > 255 || B > 255) return UNETR_ERR_UNSUPPORTED;     // TileTable packs the coordinates in bytes
    if ((long)D * H * W * lddy >= (1L << 31) || (dy3 && (long)D * H * W * lddy3 >= (1L << 31))) return UNETR_ERR_UNSUPPORTED;   // 32-bit in-tile offsets
    // 16-channel slabs everywhere (measured: 32->16 @ 96^3 200 -> 170 us, 64->32 @ 48^3 133 -> 88 us, step -0.16 ms): the dy tile is
    // re-staged once per slab, but three pipelined workgroups per CU beat two with the 32-channel window.  UNETR_WG_CIS1 = largest
    // Cin that still takes the 16-channel variant (tuning hook).
    if constexpr (B16) {
        // the single-channel image (encoder1's first conv + its 1x1x1 branch): the dedicated tap-column kernel
        if (x_f32 && Cin == 1 && Cout == 16 && ldx == 1 && ((uintptr_t)dy & 15) == 0 && (lddy & 7) == 0 &&
            (!dy3 || (((uintptr_t)dy3 & 15) == 0 && (lddy3 & 7) == 0)) && (long)D * H * W < (1L << 31) && !getenv("UNETR_CONV_C1_OFF")) {
            long G = std::min<long>(1024, ntiles);
            if (const char* e = getenv("UNETR_TEST_MAX_WG")) { if (atoi(e) > 0) G = std::min<long>(G, atoi(e)); }
            if (rows_only) { *rows_only = G; return UNETR_OK; }
            const long n1 = 27L * 16, n31 = dy3 ? 16 : 0, rows = G;
            if (!ws || (size_t)rows * (n1 + n31) * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
            if (rows_used) *rows_used = rows;
            float* wsb = ws + (size_t)rows * n1;
            hipLaunchKernelGGL(conv3_c1_wgrad_kernel, dim3((unsigned)G), dim3(256), 0, st, (const float*)x, (const uint16_t*)dy, lddy, (const uint16_t*)dy3, lddy3,
                               ws, wsb, D, H, W, ntx, nty, ntz, (int)ntiles);
            const int blocks = (int)cdiv(n1, 32), blocks3 = dy3 ? 1 : 0;
            if (!parts_only)
                hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(blocks + blocks3), dim3(256), 0, st, ws, (int)rows, n1, dw, blocks, (const float*)wsb, n31, dw3);
            return unetr_check_launch();
        }
    }
    if constexpr (std::is_same<P, PrecBF16x3>::value) {
        // bf16x3: the (hi, lo) bf16-image kernel (conv3_wgrad_x3_kernel) wherever rows of x / dy / dy3 are whole 16-byte quads of channels
        // (UNETR_X3_WGRAD_TR16=0: the 4-byte fragment path below, kept for A/B runs and for the shapes this form declines)
        const char* e = getenv("UNETR_X3_WGRAD_TR16");
        const bool yq16 = (((uintptr_t)dy | (uintptr_t)dy3) & 15) == 0 && (lddy & 3) == 0 && (Cout & 3) == 0 && (!dy3 || (lddy3 & 3) == 0);
        const bool xq16 = ((uintptr_t)x & 15) == 0 && (ldx & 3) == 0 && (Cin & 3) == 0;
        const bool xs = Cin < 4;                 // the image: scalar window loads
        if (yq16 && (xq16 || xs) && !(e && atoi(e) == 0)) {
            const int nci = cdiv(Cin, 16), nco = cdiv(Cout, 16);
            const long n = 27L * Cin * Cout, n3 = dy3 ? (long)Cin * Cout : 0;
            long G = std::min<long>(ntiles, std::max<long>(1, 512 / ((long)nci * nco)));
            if (const char* t = getenv("UNETR_TEST_MAX_WG")) { if (atoi(t) > 0) G = std::min<long>(G, atoi(t)); }
            if (rows_only) { *rows_only = G; return UNETR_OK; }
            while (G > 1 && (size_t)G * (n + n3) * sizeof(float) > ws_bytes) G >>= 1;
            if (!ws || (size_t)G * (n + n3) * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
            if (nci > 65535 || nco > 65535) return UNETR_ERR_ARG;
            if (rows_used) *rows_used = G;
            float* ws3 = ws + (size_t)G * n;
#define X3W_GO(H3_, XS_) hipLaunchKernelGGL((conv3_wgrad_x3_kernel<H3_, XS_>), dim3((unsigned)G, nci, nco), dim3(256), 0, st, (const float*)x, ldx, (const float*)dy, lddy, ws, \
                                            (const float*)dy3, lddy3, ws3, D, H, W, Cin, Cout, ntx, nty, ntz, (int)ntiles)
            if (dy3) { if (xs) X3W_GO(true, true); else X3W_GO(true, false); }
            else { if (xs) X3W_GO(false, true); else X3W_GO(false, false); }
#undef X3W_GO
            const int blocks = (int)std::min<long>((n + 31) / 32, 16384);
            const int blocks3 = dy3 ? (int)std::min<long>((n3 + 31) / 32, 16384) : 0;
            if (!parts_only)
                hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(blocks + blocks3), dim3(256), 0, st, ws, (int)G, n, dw, blocks, (const float*)ws3, n3, dw3);
            return unetr_check_launch();
        }
    }
    const int cis = Cin <= (getenv("UNETR_WG_CIS1") ? atoi(getenv("UNETR_WG_CIS1")) : (1 << 30)) ? 1 : 2;
    // UNETR_WG_NSL=2: two 16-channel slabs per workgroup (the dy / dy3 tile staged once for both) where the fast path applies and the
    // slabs pair up.  OFF by default: measured SLOWER on MI355X (same box, three interleaved rounds: 5.03 / 5.12 / 5.03 ms per step
    // against 4.93 / 4.92 / 4.91) -- the kernel is bound by latency and instruction issue, not by the 22 % of bytes this saves, and
    // half as many workgroups share a tile's work.  (The same verdict as the 32-channel window of round 2.)
    const int nsl = (B16 && cis == 1 && !x_f32 && Cin % 32 == 0 && getenv("UNETR_WG_NSL") && atoi(getenv("UNETR_WG_NSL")) == 2) ? 2 : 1;
    const int nci = cdiv(Cin, 16 * cis * nsl), nco = cdiv(Cout, 16);
    const long n = 27L * Cin * Cout;
    // persistent workgroups: all of them resident at once (3 per CU with the 16-channel slab, else 2 rounds of 2 per CU)
    long G = std::max<long>(1, (cis == 1 ? ((dy3 && !B16) ? 512 : 256 * WG_LB) : 1024) / ((long)nci * nco));
    G = std::min(G, ntiles);
    if (const char* e = getenv("UNETR_TEST_MAX_WG")) { if (atoi(e) > 0) G = std::min<long>(G, atoi(e)); }   // test hook: long tile walks
    const long n3 = dy3 ? (long)Cin * Cout : 0;
    if (rows_only) { *rows_only = G; return UNETR_OK; }
    while (G > 1 && (size_t)G * (n + n3) * sizeof(float) > ws_bytes) G >>= 1;
    if (!ws || (size_t)G * (n + n3) * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    if (rows_used) *rows_used = G;
    float* ws3 = ws + (size_t)G * n;
    const int vecy3 = (dy3 && ((uintptr_t)dy3 & 15) == 0 && (lddy3 & 3) == 0 && (Cout & 3) == 0) ? 1 : 0;
    if (nci > 65535 || nco > 65535) return UNETR_ERR_ARG;
    int vecx = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0 && (Cin & 3) == 0) ? 1 : 0;      // XMX
    const int vecy = (((uintptr_t)dy & 15) == 0 && (lddy & 3) == 0 && (Cout & 3) == 0) ? 1 : 0;
    const bool vy = vecy && (!dy3 || vecy3);
    if (B16) {
        // bf16-stored gradients (and input, unless it is the fp32 image): whole 16-byte pieces of 8 channels
        if (((uintptr_t)dy & 15) || (lddy & 7) || (Cout & 7) || (dy3 && (((uintptr_t)dy3 & 15) || (lddy3 & 7)))) return UNETR_ERR_UNSUPPORTED;
        if (!x_f32) {
            if (((uintptr_t)x & 15) || (ldx & 7) || (Cin & 7)) return UNETR_ERR_UNSUPPORTED;
            vecx = 2;
        }
    }
#define LAUNCH_WG_C(VX_, VY_, H3_, CIS_)                                                                                          \
    hipLaunchKernelGGL((conv3_wgrad_kernel<P, VX_, VY_, H3_, CIS_, (CIS_ == 1)>), dim3((unsigned)G, nci, nco), dim3(256), 0, st, x, ldx, dy, lddy, \
                       ws, dy3, lddy3, ws3, D, H, W, Cin, Cout, ntx, nty, ntz, (int)ntiles)
#define LAUNCH_WG_2(H3_)                                                                                                          \
    hipLaunchKernelGGL((conv3_wgrad_kernel<P, 2, true, H3_, 1, true, 2>), dim3((unsigned)G, nci, nco), dim3(256), 0, st, x, ldx, dy, lddy, \
                       ws, dy3, lddy3, ws3, D, H, W, Cin, Cout, ntx, nty, ntz, (int)ntiles)
#define LAUNCH_WG(VX_, VY_)                                                                                                        \
    do {                                                                                                                           \
        if (dy3) { if (cis == 1) LAUNCH_WG_C(VX_, VY_, true, 1); else LAUNCH_WG_C(VX_, VY_, true, 2); }                            \
        else { if (cis == 1) LAUNCH_WG_C(VX_, VY_, false, 1); else LAUNCH_WG_C(VX_, VY_, false, 2); }                              \
    } while (0)
    if constexpr (B16) {
        if (vecx == 2 && nsl == 2) { if (dy3) LAUNCH_WG_2(true); else LAUNCH_WG_2(false); }
        else if (vecx == 2) LAUNCH_WG(2, true);
        else if (vecx == 1) LAUNCH_WG(1, true);
        else LAUNCH_WG(0, true);
    } else {
        if (vecx && vy) LAUNCH_WG(1, true);
        else if (vecx) LAUNCH_WG(1, false);
        else if (vy) LAUNCH_WG(0, true);
        else LAUNCH_WG(0, false);
    }
    const int blocks = (int)std::min<long>((n + 31) / 32, 16384);
    const int blocks3 = dy3 ? (int)std::min<long>((n3 + 31) / 32, 16384) : 0;
    if (!parts_only)
        hipLaunchKernelGGL(conv3_wgrad_reduce_kernel, dim3(blocks + blocks3), dim3(256), 0, st, ws, (int)G, n, dw, blocks,
                           (const float*)ws3, n3, dw3);
    return unetr_check_launch();
}

}  // namespace

extern "C" size_t unetr_conv3_packed_bytes(int Cin, int Cout, int mode, int prec) {
    const int SL = prec == UNETR_PREC_BF16 ? 32 : 16, es = prec == UNETR_PREC_BF16 ? 2 : 4;
    const int K = mode ? Cout : Cin, N = mode ? Cin : Cout;
    return (size_t)27 * ((K + SL - 1) / SL) * N * SL * es;
}

extern "C" int unetr_conv3_pack_weight(const float* w, void* wpack, int Cin, int Cout, int mode, int prec, void* stream) {
    if (!w || !wpack || Cin <= 0 || Cout <= 0) return UNETR_ERR_ARG;
    if (prec == UNETR_PREC_BF16) return pack_t<PrecBF16>(w, wpack, Cin, Cout, mode, (hipStream_t)stream);
    if (prec == UNETR_PREC_F32) return pack_t<PrecF32>(w, wpack, Cin, Cout, mode, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return pack_t<PrecBF16x3>(w, wpack, Cin, Cout, mode, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

extern "C" int unetr_conv3_fwd(const void* x, long ldx, const void* wpack, void* y, long ldy, int accumulate,
                               int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream) {
    if (!x || !wpack || !y || B <= 0) return UNETR_ERR_ARG;
    if (Cout % 16) return UNETR_ERR_UNSUPPORTED;
    if (prec == UNETR_PREC_BF16) return fwd_t<PrecBF16>(x, ldx, wpack, y, ldy, accumulate, B, D, H, W, Cin, Cout, (hipStream_t)stream);
    if (prec == UNETR_PREC_F32) return fwd_t<PrecF32>(x, ldx, wpack, y, ldy, accumulate, B, D, H, W, Cin, Cout, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return fwd_t<PrecBF16x3>(x, ldx, wpack, y, ldy, accumulate, B, D, H, W, Cin, Cout, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

// Forward of the first half of MONAI's UnetResBlock in one launch (unetr.py:90-98 and the decoder blocks :135-174):
// y = conv3x3x3(x), its InstanceNorm statistics, and optionally y3 = conv1x1x1(x) with its statistics.
extern "C" int unetr_conv3_fwd_fused(const void* x, long ldx, const void* wpack, void* y, long ldy, float* stats,
                                     const void* w3pack, void* y3, long ldy3, float* stats3, float eps,
                                     int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32,
                                     float* ws, size_t ws_bytes, void* stream) {
    if (!x || !wpack || !y || !stats || B <= 0) return UNETR_ERR_ARG;
    if ((w3pack != nullptr) != (y3 != nullptr) || (w3pack != nullptr) != (stats3 != nullptr)) return UNETR_ERR_ARG;
    if (Cout % 16 || (w3pack && ldy3 != ldy)) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int max_rows = UNETR_CONV3_MAX_ROWS;                               // one partial row per workgroup and batch item
    const size_t per = (size_t)B * max_rows * 2 * Cout;
    if (!ws || per * (w3pack ? 2 : 1) * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    FuseArgs fz{ws, w3pack, y3, ldy3, w3pack ? ws + per : nullptr, 0, 0};
    int rc;
    if (prec == UNETR_PREC_BF16) rc = fwd_t<PrecBF16>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz, x_f32);
    else if (prec == UNETR_PREC_F32) rc = fwd_t<PrecF32>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz);
    else if (prec == UNETR_PREC_BF16X3) rc = fwd_t<PrecBF16x3>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz);
    else return UNETR_ERR_ARG;
    if (rc) return rc;
    const long V = (long)D * H * W;
    if (w3pack) rc = unetr_instnorm_stats_finalize2(ws, ws + per, fz.rows, B, V, Cout, eps, stats, stats3, stream);      // one launch for both sets
    else rc = unetr_instnorm_stats_finalize(ws, fz.rows, B, V, Cout, eps, stats, stream);
    return rc;
}

// The same launch WITHOUT the statistics finalize: the partial rows (part / part3: [B][rows][2][Cout] floats, caller-allocated for
// UNETR_CONV3_MAX_ROWS rows) are handed to the consumer, which reduces them in its own prologue (unetr_instnorm_apply_fin).
extern "C" int unetr_conv3_fwd_parts(const void* x, long ldx, const void* wpack, void* y, long ldy, float* part,
                                     const void* w3pack, void* y3, long ldy3, float* part3, int* rows_out,
                                     int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32, void* stream) {
    if (!x || !wpack || !y || !part || !rows_out || B <= 0) return UNETR_ERR_ARG;
    // y3 may be NULL with w3pack given: only the statistics of the 1x1x1 branch are wanted (the block on the image: its consumers
    // form the branch from the image, unetr_instnorm_apply_fin_img)
    if ((y3 != nullptr && w3pack == nullptr) || (w3pack != nullptr) != (part3 != nullptr)) return UNETR_ERR_ARG;
    if (Cout % 16 || (y3 && ldy3 != ldy)) return UNETR_ERR_UNSUPPORTED;
    if (!y3) ldy3 = ldy;
    hipStream_t st = (hipStream_t)stream;
    FuseArgs fz{part, w3pack, y3, ldy3, part3, 0, 0};
    int rc;
    if (prec == UNETR_PREC_BF16) rc = fwd_t<PrecBF16>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz, x_f32);
    else if (prec == UNETR_PREC_F32) rc = fwd_t<PrecF32>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz);
    else if (prec == UNETR_PREC_BF16X3) rc = fwd_t<PrecBF16x3>(x, ldx, wpack, y, ldy, 0, B, D, H, W, Cin, Cout, st, &fz);
    else return UNETR_ERR_ARG;
    if (rc) return rc;
    if (fz.rows <= 0 || fz.rows > UNETR_CONV3_MAX_ROWS) return UNETR_ERR_WORKSPACE;
    *rows_out = fz.rows;
    return UNETR_OK;
}

// Data gradient dx = conv3x3x3^T(dy; w) of a conv whose INPUT went through InstanceNorm + LeakyReLU (MONAI UnetResBlock: conv2
// reads lrelu(IN(c1))), with the backward statistics of that norm formed in the epilogue: part [B][rows][2][Cin] floats holds
// per-workgroup partial sums of g and g * n (g = dx * lrelu'(n), n = (xn - mean) * rstd from stats [B][Cin][2]); xn has the
// shape / pitch class of dx.  The consumer (unetr_instnorm_bwd_apply_fin) reduces the rows in its prologue.
extern "C" int unetr_conv3_dgrad_stats(const void* dy, long lddy, const void* wpack_dgrad, void* dx, long lddx,
                                       const void* xn, long ldxn, const float* stats, float* part, int* rows_out,
                                       int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream) {
    if (!dy || !wpack_dgrad || !dx || !xn || !stats || !part || !rows_out || B <= 0) return UNETR_ERR_ARG;
    if (Cin % 16) return UNETR_ERR_UNSUPPORTED;
    const int al = prec == UNETR_PREC_BF16 ? 7 : 15;
    if (((uintptr_t)xn & al) || (ldxn & 3)) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    FuseArgs fz{part, stats, const_cast<void*>(xn), ldxn, nullptr, 0, 0, 1};
    int rc;
    // the data gradient is the forward kernel with contraction over the conv's Cout channels and Cin outputs
    if (prec == UNETR_PREC_BF16) rc = fwd_t<PrecBF16>(dy, lddy, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    else if (prec == UNETR_PREC_F32) rc = fwd_t<PrecF32>(dy, lddy, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    else if (prec == UNETR_PREC_BF16X3) rc = fwd_t<PrecBF16x3>(dy, lddy, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    else return UNETR_ERR_ARG;
    if (rc) return rc;
    if (fz.rows <= 0 || fz.rows > UNETR_CONV3_MAX_ROWS) return UNETR_ERR_WORKSPACE;
    *rows_out = fz.rows;
    return UNETR_OK;
}

extern "C" size_t unetr_conv3_packed_1x1_bytes(int Cin, int Cout, int prec) {
    const int SL = prec == UNETR_PREC_BF16 ? 32 : 16;
    return (size_t)((Cin + SL - 1) / SL) * Cout * 64;
}

extern "C" int unetr_conv3_pack_grouped(const unetr_pack_problem* probs, int n, int prec, void* stream) {
    if (!probs || n <= 0 || (prec != UNETR_PREC_BF16 && prec != UNETR_PREC_F32 && prec != UNETR_PREC_BF16X3)) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int SL = prec == UNETR_PREC_BF16 ? 32 : 16;
    for (int base = 0; base < n; base += PK_MAX) {
        PkArgs a;
        a.n = std::min(PK_MAX, n - base);
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const unetr_pack_problem& q = probs[base + i];
            if (!q.w || !q.out || q.Cin <= 0 || q.Cout <= 0 || q.kind < 0 || q.kind > 4) return UNETR_ERR_ARG;
            if (q.kind == 4) {       // transposed-conv tap-major pack: always bf16 (the operand of unetr_gemm_bf16)
                if (prec != UNETR_PREC_BF16) return UNETR_ERR_UNSUPPORTED;
                const long total4 = 8L * q.Cin * q.Cout;
                if (total4 >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;
                a.p[i] = PkProblem{q.w, q.out, q.Cin, q.Cout, 4, 0, blocks, total4, 0};
                blocks += cdiv(total4, 2048);
                continue;
            }
            const int K = (q.kind == 0 || q.kind == 2) ? q.Cin : q.Cout, N = (q.kind == 0 || q.kind == 2) ? q.Cout : q.Cin;
            const int pair = (prec == UNETR_PREC_BF16 && q.kind != 3 && K <= 16 && conv_pipe_enabled()) ? 1 : 0;
            long total;
            if (q.kind <= 1) total = pair ? 14L * N * 32 : 27L * ((K + SL - 1) / SL) * N * SL;
            else total = pair ? (long)N * 32 : (long)((K + SL - 1) / SL) * N * SL;
            if (total >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;
            const int staged = (q.kind <= 1 && !pair) ? 1 : 0;
            a.p[i] = PkProblem{q.w, q.out, q.Cin, q.Cout, q.kind, pair, blocks, total, staged};
            blocks += staged ? ((K + SL - 1) / SL) * cdiv(N, 8) : cdiv(total, 2048);
        }
        if (prec == UNETR_PREC_BF16) hipLaunchKernelGGL((conv3_pack_grouped_kernel<uint16_t>), dim3(blocks), dim3(256), 0, st, a, SL);
        else if (prec == UNETR_PREC_BF16X3) hipLaunchKernelGGL((conv3_pack_grouped_kernel<uint32_t>), dim3(blocks), dim3(256), 0, st, a, SL);
        else hipLaunchKernelGGL((conv3_pack_grouped_kernel<float>), dim3(blocks), dim3(256), 0, st, a, SL);
    }
    return unetr_check_launch();
}

static int pack_1x1(const float* w3, void* w3pack, int K, int N, int prec, int allow_pair, int transposed, hipStream_t st) {
    // rows n < N, contraction index k < K
    if (prec == UNETR_PREC_BF16) {
        const int pair = (allow_pair && use_pair<PrecBF16>(K) && conv_pipe_enabled()) ? 1 : 0;
        hipLaunchKernelGGL((conv3_pack_1x1_kernel<uint16_t>), dim3(cdiv((long)((K + 31) / 32) * N * 32, 256)), dim3(256), 0, st, w3,
                           (uint16_t*)w3pack, K, N, pair, 32, transposed);
    } else if (prec == UNETR_PREC_BF16X3) {
        hipLaunchKernelGGL((conv3_pack_1x1_kernel<uint32_t>), dim3(cdiv((long)((K + 15) / 16) * N * 16, 256)), dim3(256), 0, st, w3,
                           (uint32_t*)w3pack, K, N, 0, 16, transposed);
    } else if (prec == UNETR_PREC_F32) {
        hipLaunchKernelGGL((conv3_pack_1x1_kernel<float>), dim3(cdiv((long)((K + 15) / 16) * N * 16, 256)), dim3(256), 0, st, w3,
                           (float*)w3pack, K, N, 0, 16, transposed);
    } else return UNETR_ERR_ARG;
    return unetr_check_launch();
}

extern "C" int unetr_conv3_pack_1x1(const float* w3, void* w3pack, int Cin, int Cout, int prec, void* stream) {
    if (!w3 || !w3pack || Cin <= 0 || Cout <= 0) return UNETR_ERR_ARG;
    return pack_1x1(w3, w3pack, Cin, Cout, prec, 1, 0, (hipStream_t)stream);
}

// Data gradient of MONAI's UnetResBlock input in one launch: dx = conv3x3x3^T(dc1; w1) + conv1x1x1^T(dc3; w3).
// wpack_dgrad from unetr_conv3_pack_weight(mode 1); w3 is the 1x1x1 weight [Cout, Cin] itself (packed here into ws).
extern "C" int unetr_conv3_dgrad_fused(const void* dc1, long ld1, const void* wpack_dgrad, const void* dc3, long ld3, const float* w3,
                                       const void* w3pack_t, void* dx, long lddx, int B, int D, int H, int W, int Cin, int Cout,
                                       int prec, float* ws, size_t ws_bytes, void* stream) {
    if (!dc1 || !wpack_dgrad || !dc3 || (!w3 && !w3pack_t) || !dx || B <= 0) return UNETR_ERR_ARG;
    if (Cin % 16 || Cout % 4 || (ld3 & 3) || ((uintptr_t)dc3 & 15)) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const void* w3t = w3pack_t;
    if (!w3t) {                                                        // pack the transposed 1x1x1 weights into the workspace
        const size_t need = unetr_conv3_packed_1x1_bytes(Cout, Cin, prec);
        if (!ws || need > ws_bytes) return UNETR_ERR_WORKSPACE;
        int rc = pack_1x1(w3, ws, Cout, Cin, prec, 0, 1, st);          // rows n = block input channel, k = block output channel
        if (rc) return rc;
        w3t = ws;
    }
    FuseArgs fz{nullptr, w3t, const_cast<void*>(dc3), ld3, nullptr, 0, Cout};
    // the data gradient is the same kernel with contraction over the block's Cout channels and Cin outputs
    if (prec == UNETR_PREC_BF16) return fwd_t<PrecBF16>(dc1, ld1, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    if (prec == UNETR_PREC_F32) return fwd_t<PrecF32>(dc1, ld1, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    if (prec == UNETR_PREC_BF16X3) return fwd_t<PrecBF16x3>(dc1, ld1, wpack_dgrad, dx, lddx, 0, B, D, H, W, Cout, Cin, st, &fz);
    return UNETR_ERR_ARG;
}

extern "C" int unetr_conv3_wgrad(const void* x, long ldx, const void* dy, long ldy, float* dw,
                                 const void* dy3, long ldy3, float* dw3,
                                 int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32,
                                 float* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw || B <= 0 || ((dy3 != nullptr) != (dw3 != nullptr))) return UNETR_ERR_ARG;
    if (prec == UNETR_PREC_BF16) return wgrad_t<PrecBF16>(x, ldx, dy, ldy, dw, dy3, ldy3, dw3, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream, x_f32);
    if (prec == UNETR_PREC_F32) return wgrad_t<PrecF32>(x, ldx, dy, ldy, dw, dy3, ldy3, dw3, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    if (prec == UNETR_PREC_BF16X3) return wgrad_t<PrecBF16x3>(x, ldx, dy, ldy, dw, dy3, ldy3, dw3, B, D, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
    return UNETR_ERR_ARG;
}

// number of partial rows unetr_conv3_wgrad_parts writes for this shape (< 0: the shape is not supported)
extern "C" long unetr_conv3_wgrad_rows(int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32, int has3) {
    long rows = -1;
    const void* dummy = (const void*)(uintptr_t)256;           // (aligned, never dereferenced: rows_only returns before any launch)
    int rc = UNETR_ERR_ARG;
    const long ldx = x_f32 && Cin == 1 ? 1 : (Cin + 7) / 8 * 8;
    if (prec == UNETR_PREC_BF16) rc = wgrad_t<PrecBF16>(dummy, ldx, dummy, Cout, nullptr, has3 ? dummy : nullptr, Cout, nullptr, B, D, H, W, Cin, Cout, nullptr, 0, nullptr, x_f32, true, &rows);
    else if (prec == UNETR_PREC_F32) rc = wgrad_t<PrecF32>(dummy, ldx, dummy, Cout, nullptr, has3 ? dummy : nullptr, Cout, nullptr, B, D, H, W, Cin, Cout, nullptr, 0, nullptr, 0, true, &rows);
    else if (prec == UNETR_PREC_BF16X3) rc = wgrad_t<PrecBF16x3>(dummy, ldx, dummy, Cout, nullptr, has3 ? dummy : nullptr, Cout, nullptr, B, D, H, W, Cin, Cout, nullptr, 0, nullptr, 0, true, &rows);
    return rc == UNETR_OK ? rows : -1;
}

// unetr_conv3_wgrad without its reduce launch: the per-workgroup partial sums stay in `part` -- [rows][27 Cin Cout] followed, when dy3
// is given, by [rows][Cin Cout] (rows = unetr_conv3_wgrad_rows) -- for unetr_reduce_rows_grouped
extern "C" int unetr_conv3_wgrad_parts(const void* x, long ldx, const void* dy, long ldy, const void* dy3, long ldy3,
                                       float* part, size_t part_bytes, long* rows_out, int B, int D, int H, int W, int Cin, int Cout, int prec,
                                       int x_f32, void* stream) {
    if (!x || !dy || !part || !rows_out || B <= 0) return UNETR_ERR_ARG;
    float* dummy3 = dy3 ? part : nullptr;                        // (dw / dw3 are not written in this form)
    hipStream_t st = (hipStream_t)stream;
    if (prec == UNETR_PREC_BF16) return wgrad_t<PrecBF16>(x, ldx, dy, ldy, part, dy3, ldy3, dummy3, B, D, H, W, Cin, Cout, part, part_bytes, st, x_f32, true, nullptr, rows_out);
    if (prec == UNETR_PREC_F32) return wgrad_t<PrecF32>(x, ldx, dy, ldy, part, dy3, ldy3, dummy3, B, D, H, W, Cin, Cout, part, part_bytes, st, 0, true, nullptr, rows_out);
    if (prec == UNETR_PREC_BF16X3) return wgrad_t<PrecBF16x3>(x, ldx, dy, ldy, part, dy3, ldy3, dummy3, B, D, H, W, Cin, Cout, part, part_bytes, st, 0, true, nullptr, rows_out);
    return UNETR_ERR_ARG;
}

extern "C" int unetr_debug_tr16(const void* in, void* out, void* stream) {
    if (!in || !out) return UNETR_ERR_ARG;
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)in, (uint16_t*)out);
    return unetr_check_launch();
}
