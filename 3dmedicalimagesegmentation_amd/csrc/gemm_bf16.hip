// unetr_gemm_bf16: the ViT encoder's Linear layers with bf16-STORED operands (reference: monai SABlock / MLPBlock
// nn.Linear calls reached from /root/reference/unetr.py:78-90,190).  The fp32-storage GEMM family (gemm_kernel.hpp)
// converts fp32 -> bf16 on the way into LDS: twice the operand bytes through L2 and a VALU pass per chunk, which caps
// it near 12 % of the MFMA peak on large shapes.  Here the producers (LayerNorm, attention, GELU epilogue, the
// optimizer's weight shadow) already emit bf16, so a tile travels HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4,
// no VGPR round trip, no ds_write) into a double-buffered 128-byte-row image, and the waves only read fragments and
// issue MFMAs.
//
//   C[M,N] (+)= alpha * A[M,K] . B^T + epilogue      A: bf16 [M,K] (k contiguous)
//       b_kn = 0: B is bf16 [N,K] (k contiguous; forward  y = x W^T with W as stored by nn.Linear)
//       b_kn = 1: B is bf16 [K,N] (n contiguous; data gradient dx = dy W with the same W) -- fragments come out of
//                 LDS through ds_read_b64_tr_b16, the transposing read, so no transposed weight copy is kept
//
// LDS image (both operands, b_kn = 0): row r of the tile is 128 bytes = 8 chunks of 8 bf16; chunk c of row r sits
// in slot c ^ ((r >> 1) & 7).  An LDS-DMA writes wave-uniform base + lane * 16, i.e. lane-linear, so the swizzle
// is applied to the per-lane SOURCE address: lane (row r, slot s) fetches global chunk s ^ ((r >> 1) & 7).  The 16
// lanes of one ds_read_b128 group (rows r..r+15, same logical chunk) then hit 16 different 16-byte slots of the
// 256-byte bank row.
//
// Pipeline: one barrier per 64-deep K step.  __syncthreads() drains stage kt's DMA (hipcc emits vmcnt(0) before the
// barrier while an LDS-DMA is in flight) and orders it for every reader; stage kt+1 is issued right after it into
// the other buffer, whose last readers finished before they reached this barrier; then fragments + MFMAs of stage kt.
#include <type_traits>
#include "gemm_kernel.hpp"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS_ __attribute__((address_space(3)))

struct EpBf {
    int vec_ok;          // N, pitches and base pointers allow 16-byte (8-byte bf16) row-segment accesses
    float* C; long ldc;
    uint16_t* Cb; long ldcb;
    const float* bias;
    const float* res; long ldr; int res_mod;
    float* pre;
    const float* aux; long ldaux;
    int act, accumulate; float alpha;
    int tc_d, tc_h, tc_w, tc_cout;      // > 0: transposed-conv scatter of the bf16 output (see unetr_gemm_bf16_desc)
    // element offset in Cb of (input voxel m, column n = tap * tc_cout + co)
    __device__ __forceinline__ long tc_off(int m, int n) const {
        const int tap = n / tc_cout, co = n - tap * tc_cout;
        const int x = m % tc_w; int t = m / tc_w; const int y = t % tc_h; t /= tc_h; const int z = t % tc_d; const int b = t / tc_d;
        const long ov = (((long)b * 2 * tc_d + 2 * z + (tap >> 2)) * 2 * tc_h + 2 * y + ((tap >> 1) & 1)) * 2 * tc_w + 2 * x + (tap & 1);
        return ov * ldcb + co;
    }
    __device__ __forceinline__ void store(int, int m, int n, float v) const {
        v *= alpha;
        if (bias) v += bias[n];
        if (pre) pre[(long)m * ldc + n] = v;
        if (act == 1) v = gelu_fast(v);
        else if (act == 2) v *= gelu_grad_fast(aux[(long)m * ldaux + n]);
        if (res) v += res[(long)(m % res_mod) * ldr + n];
        if (C) {
            if (accumulate) v += C[(long)m * ldc + n];
            C[(long)m * ldc + n] = v;
        }
        if (Cb) {
            __bf16 h = (__bf16)v;
            Cb[tc_cout ? tc_off(m, n) : (long)m * ldcb + n] = __builtin_bit_cast(uint16_t, h);
        }
    }
    // four consecutive columns n..n+3 of row m (n % 4 == 0, all pitches multiples of 4 checked by the host)
    __device__ __forceinline__ void store4(int m, int n, f32x4 v) const {
        v *= alpha;
        if (bias) v += *(const f32x4*)(bias + n);
        if (pre) *(f32x4*)(pre + (long)m * ldc + n) = v;
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
        } else if (act == 2) {
            const f32x4 a = *(const f32x4*)(aux + (long)m * ldaux + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(a[e]);
        }
        if (res) v += *(const f32x4*)(res + (long)(m % res_mod) * ldr + n);
        if (C) {
            if (accumulate) v += *(const f32x4*)(C + (long)m * ldc + n);
            *(f32x4*)(C + (long)m * ldc + n) = v;
        }
        if (Cb) *(bf16x4*)(Cb + (tc_cout ? tc_off(m, n) : (long)m * ldcb + n)) = __builtin_convertvector(v, bf16x4);
    }
    // store4 in two halves: the epilogue's READS (bias, GELU' argument, residual, accumulate target) of a row segment, and the
    // arithmetic + stores.  The kernel requests the reads of ALL its tiles before the first store: the pointers of this struct may
    // alias as far as the compiler knows, so inside store4 the loads of tile t+1 could not move above the stores of tile t and
    // every tile of the epilogue paid its own memory round trip (the GELU' argument comes out of HBM: 5.3 MB per block)
    struct Ops { f32x4 b, a, r, c; };
    __device__ __forceinline__ Ops load4(int m, int n) const {
        Ops o;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        o.b = bias ? *(const f32x4*)(bias + n) : z;
        o.a = act == 2 ? *(const f32x4*)(aux + (long)m * ldaux + n) : z;
        o.r = res ? *(const f32x4*)(res + (long)(m % res_mod) * ldr + n) : z;
        o.c = (C && accumulate) ? *(const f32x4*)(C + (long)m * ldc + n) : z;
        return o;
    }
    __device__ __forceinline__ void finish4(int m, int n, f32x4 v, const Ops& o) const {
        v *= alpha;
        if (bias) v += o.b;
        if (pre) *(f32x4*)(pre + (long)m * ldc + n) = v;
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
        } else if (act == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(o.a[e]);
        }
        if (res) v += o.r;
        if (C) {
            if (accumulate) v += o.c;
            *(f32x4*)(C + (long)m * ldc + n) = v;
        }
        if (Cb) *(bf16x4*)(Cb + (tc_cout ? tc_off(m, n) : (long)m * ldcb + n)) = __builtin_convertvector(v, bf16x4);
    }
};

// b_kn image swizzle: XOR applied to the 16-byte chunk index of reduction row r (chunk PAIRS move, a transposing read
// touches 8 bytes).  The 16 (g, q) rows one ds_read_b64_tr_b16 wave-instruction touches (r = 8g + q) are spread over
// the 8 pair positions of the 256-byte bank row: 16 chunks/row -> bits (q, g&1); 8 chunks/row (two rows per bank
// row, r&1 picks the half) -> bits (q>>1, g&1).
template <int CPR> __device__ __forceinline__ int bkn_x(int r) {
    if constexpr (CPR >= 16) return ((r & 3) | (((r >> 3) & 1) << 2)) << 1;
    else return (((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1;
}

// tile -> (tm, tn): workgroup L runs on XCD L % 8; give every XCD a contiguous run of tiles, m fastest, so the
// column tiles (weights) an XCD touches are few and stay in its own L2 while A is shared through MALL
__device__ __forceinline__ bool tile_of(int L, int mt, int nt, int& tm, int& tn) {
    const int T = mt * nt, per = (T + 7) >> 3;
    const int t = (L & 7) * per + (L >> 3);
    if ((L >> 3) >= per || t >= T) return false;
    tm = t % mt; tn = t / mt;
    return true;
}

// X3 (bf16x3 precision mode, round 4): the SAME kernel on fp32-stored operands.  A 128-byte image row is then 32 fp32 values (a K
// stage is 32 elements), the LDS-DMA moves the raw fp32 bits, and a fragment chunk (4 consecutive k of a row) is split into the four
// [hi | lo << 16] words of PrecBF16x3 in REGISTERS right after its ds_read_b128 (PrecBF16x3::from_raw), two MFMAs per chunk pair.  The
// generic fp32-storage kernel (gemm_kernel.hpp: register staging, one LDS buffer, two barriers per 32-deep step) took 24-27 us per
// launch at 432 rows against 10 us for this kernel in bf16 mode.  [K,N] operand (data gradients): no transposing read exists for
// 4-byte elements -- four ds_read_b32 per chunk from an image of 32 reduction rows x BN columns whose 16-byte chunk c of row r sits
// in slot c ^ (((r >> 2) & 3) << 2), so the four lane groups of a read (rows 4g + t) hit four different 64-byte segments.
// X3 = 2: B arrives as pre-split words (the optimizer-maintained word shadow of the weights: unetr_split_words) -- only the activation
// operand is split in registers, and it becomes the duplicated one.
template <int WM, int WN, int WVM, int WVN, bool BKN, int NS, int X3 = 0>
__global__ void __launch_bounds__(64 * WVM * WVN)
gemm_bf16_kernel(int M, int N, int K, int mt, int nt, int splits, int kper,
                 const typename std::conditional<X3 != 0, float, uint16_t>::type* __restrict__ A, long lda,
                 const typename std::conditional<X3 != 0, float, uint16_t>::type* __restrict__ B, long ldb, EpBf ep,
                 float* __restrict__ ws) {
    typedef typename std::conditional<X3 != 0, float, uint16_t>::type ET;
    constexpr int ESZ = sizeof(ET), CE = 16 / ESZ;           // bytes per element, elements per 16-byte chunk
    constexpr int NT = 64 * WVM * WVN, BM = 16 * WM * WVM, BN = 16 * WN * WVN, BK = 128 / ESZ;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BKN ? BK * BN * ESZ : BN * 128;
    static_assert(!(X3 != 0 && BKN) || BN >= 64, "the [K,N] fp32 image needs >= 16 chunks per row for its swizzle");
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int AIT = A_BYTES / 16 / NT, BIT = B_BYTES / 16 / NT;
    static_assert(A_BYTES % (16 * NT) == 0 && B_BYTES % (16 * NT) == 0, "tile must divide into whole wave DMAs");
    constexpr int G = AIT + BIT;                 // LDS-DMA instructions per thread per stage
    static_assert(NS >= 2 && (NS - 2) * G <= 63, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];

    int tm, tn;
    if (!tile_of(blockIdx.x, mt, nt, tm, tn)) return;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y, kbeg = split * kper, kend = min(K, kbeg + kper);
    const int nk = (kend - kbeg) / BK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WVN, wn = wave % WVN;

    // per-thread source pointers of the DMA pieces at k = kbeg (advanced by BK elements / BK rows per stage)
    const ET* asrc[AIT];
    const ET* bsrc[BIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int id = tid + i * NT, r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
        asrc[i] = A + (long)min(m0 + r, M - 1) * lda + kbeg + c * CE;
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
        const int id = tid + i * NT;
        if constexpr (!BKN) {
            const int r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
            bsrc[i] = B + (long)min(n0 + r, N - 1) * ldb + kbeg + c * CE;
        } else {
            // image: BK rows (reduction index) of BN elements, CPR 16-byte chunks per row, swizzled by bkn_x (bf16: transposing
            // reads) / by the lane group of the 4-byte reads (X3)
            constexpr int CPR = BN / CE;
            const int r = id / CPR, s = id % CPR;
            const int c = X3 != 0 ? (s ^ (((r >> 2) & 3) << 2)) : (s ^ bkn_x<CPR>(r));
            bsrc[i] = B + (long)(kbeg + r) * ldb + min(n0 + c * CE, N - CE);
        }
    }
    auto issue = [&](int kt, int buf) {
        char* la = lds + buf * STAGE;
        char* lb = la + A_BYTES;
#pragma unroll
        for (int i = 0; i < AIT; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(asrc[i] + (long)kt * BK), (lds_void_t*)(la + (wave * 64 + i * NT) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const ET* g = BKN ? bsrc[i] + (long)kt * BK * ldb : bsrc[i] + (long)kt * BK;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(lb + (wave * 64 + i * NT) * 16), 16, 0, 0);
        }
    };

    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NS-stage ring, NS-1 stages in flight.  At the top of step kt: wait until this wave's pieces of stage kt have landed
    // (a counted vmcnt leaves the NS-2 younger stages in flight; the tail drains), and until its own fragment reads of
    // step kt-1 are done; the raw barrier then publishes stage kt to every wave and frees the buffer of step kt-1, which
    // the DMA of stage kt+NS-1 overwrites.  (__syncthreads() would drain vmcnt(0) and serialise the ring.)
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const char* la = lds + (kt % NS) * STAGE;
        const char* lb = la + A_BYTES;
        // all fragment reads of the stage (both 32-deep halves) are issued before its first MFMA, so the second half's LDS latency
        // hides under the first half's MFMAs (read 4, wait, MFMA 4, read 4, wait, MFMA 4 exposed it twice per stage -- with
        // one workgroup per CU nothing else runs on the SIMD meanwhile)
        u32x4 a[2][WM], b[2][WN];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
                a[kb][i] = *(const u32x4*)(la + lds_tile_off((wm * WM + i) * 16 + (lane & 15), kb * 4 + (lane >> 4)));
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if constexpr (!BKN) {
                    b[kb][j] = *(const u32x4*)(lb + lds_tile_off((wn * WN + j) * 16 + (lane & 15), kb * 4 + (lane >> 4)));
                } else if constexpr (X3 != 0) {
                    // lane (n = lane & 15, g = lane >> 4): reduction rows kb * 16 + 4 g + t, t = 0..3, of column n
                    const int col = (wn * WN + j) * 16 + (lane & 15), g = lane >> 4;
                    const char* pb = lb + (kb * 16 + 4 * g) * (BN * 4) + ((((col >> 2) ^ (g << 2))) << 4) + (col & 3) * 4;
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[kb][j][t] = *(const uint32_t*)(pb + t * (BN * 4));
                } else {
                    constexpr int CPR = BN / 8;
                    const int cc = lane & 15, g = lane >> 4, q = cc >> 2, p = cc & 3;
                    const int r0 = kb * 32 + 8 * g + q, r1 = r0 + 4;
                    const int ch = (wn * WN + j) * 2 + (p >> 1);               // logical 16-byte chunk of the 4 columns
                    const int s0 = ch ^ bkn_x<CPR>(r0);
                    const int s1 = ch ^ bkn_x<CPR>(r1);
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(lb + r0 * (BN * 2) + s0 * 16 + (p & 1) * 8));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(lb + r1 * (BN * 2) + s1 * 16 + (p & 1) * 8));
                    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    b[kb][j] = __builtin_bit_cast(u32x4, v);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            if constexpr (X3 == 2) {
                // B holds words already: A is the duplicated operand; acc += B . A_hh + B . A_ll
                u32x4 ah[WM], al[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) x3_dup(a[kb][i], ah[i], al[i]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        PrecBF16::mma(acc[i][j], b[kb][j], ah[i]);
                        PrecBF16::mma(acc[i][j], b[kb][j], al[i]);
                    }
            } else if constexpr (X3 == 1) {
                // raw fp32 bits -> operand forms, once per fragment: A as [hi | lo] words, B duplicated ([hi, hi] / [lo, lo]);
                // acc += B_hh . A + B_ll . A  =  b_hi a_hi + b_hi a_lo + b_lo a_hi + b_lo a_lo
                u32x4 bh[WN], bl[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[kb][i] = x3_words(a[kb][i]);
#pragma unroll
                for (int j = 0; j < WN; ++j) x3_dup(b[kb][j], bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        PrecBF16::mma(acc[i][j], bh[j], a[kb][i]);
                        PrecBF16::mma(acc[i][j], bl[j], a[kb][i]);
                    }
            } else {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) PrecBF16::mma(acc[i][j], b[kb][j], a[kb][i]);    // transposed tile: a lane holds 4 consecutive n of one m
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // The MFMAs above ran with swapped operands, so the accumulator tile is C^T: lane (c, g) holds C[m = tile row c]
    // [n = 4g .. 4g+3] -- one 16-byte store (or 8-byte bf16 store) per tile instead of four 4-byte ones.
    const bool vec4 = ep.vec_ok;
    if (vec4 && splits == 1) {
        // (operand reads first -- clamped addresses for the tiles past the edge --, then arithmetic and stores: all tiles at once
        // where a wave owns <= 4 of them, row of tiles by row of tiles in the larger variants, whose register budget is spoken for)
        constexpr int RB = WM * WN <= 4 ? WM : 1;          // tile rows per batch
#pragma unroll
        for (int i0 = 0; i0 < WM; i0 += RB) {
            EpBf::Ops ops[RB][WN];
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int m = min(m0 + (wm * WM + i0 + i) * 16 + (lane & 15), M - 1);
                    const int n = min(n0 + (wn * WN + j) * 16 + 4 * (lane >> 4), N - 4);
                    ops[i][j] = ep.load4(m, n);
                }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int m = m0 + (wm * WM + i0 + i) * 16 + (lane & 15);
                    const int n = n0 + (wn * WN + j) * 16 + 4 * (lane >> 4);
                    if (m < M && n < N) ep.finish4(m, n, acc[i0 + i][j], ops[i][j]);
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int m = m0 + (wm * WM + i) * 16 + (lane & 15);
            const int n = n0 + (wn * WN + j) * 16 + 4 * (lane >> 4);
            if (m >= M || n >= N) continue;
            if (vec4) {
                if (splits > 1) *(f32x4*)(ws + ((long)split * M + m) * N + n) = acc[i][j];
                else ep.store4(m, n, acc[i][j]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) {
                        if (splits > 1) ws[((long)split * M + m) * N + n + r] = acc[i][j][r];
                        else ep.store(0, m, n + r, acc[i][j][r]);
                    }
            }
        }
}

template <int WM, int WN, int WVM, int WVN, bool BKN, int NS, int X3 = 0>
int launch_bf16(int M, int N, int K, const typename std::conditional<X3 != 0, float, uint16_t>::type* A, long lda,
                const typename std::conditional<X3 != 0, float, uint16_t>::type* B, long ldb, const EpBf& ep,
                float* ws, size_t ws_bytes, hipStream_t st, int* partial_splits = nullptr) {
    constexpr int BM = 16 * WM * WVM, BN = 16 * WN * WVN, BKE = X3 != 0 ? 32 : 64;      // elements per 128-byte K stage
    const int mt = cdiv(M, BM), nt = cdiv(N, BN), ksteps = K / BKE;
    const long tiles = (long)mt * nt;
    int splits = 1;
    // few tiles (batch-2 token counts): the kernel is latency-bound, one workgroup's time is ~ its K steps, so long K
    // ranges are cut into slabs of >= 12 steps (shorter slabs cost more in the reduce launch than they save)
    const int min_steps = (X3 != 0 && getenv("UNETR_X3_SPLIT_K768") == nullptr) ? 48 : 24;       // (X3: 32-element stages; K = 768 stays whole as in bf16 mode)
    if (tiles < 192 && ksteps >= min_steps) splits = std::min(ksteps / (min_steps / 2), (int)((512 + tiles - 1) / tiles));
    if (const char* e = getenv("UNETR_GEMM_SPLITS")) { int v = atoi(e); if (v > 0) splits = std::min(v, ksteps); }
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    if (splits < 1 || ws == nullptr) splits = 1;
    const int kper = cdiv(ksteps, splits) * BKE;
    splits = cdiv(K, kper);
    const int per = cdiv(tiles, 8);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, WVM, WVN, BKN, NS, X3>), dim3(per * 8, splits), dim3(64 * WVM * WVN), 0, st,
                       M, N, K, mt, nt, splits, kper, A, lda, B, ldb, ep, ws);
    // partial_splits: the caller consumes the split partials itself (unetr_gemm_bf16_ln_bwd): no reduce launch
    if (partial_splits) *partial_splits = splits;
    if (splits > 1 && !partial_splits)
        hipLaunchKernelGGL((splitk_reduce_kernel<EpBf, false>), dim3(cdiv(N, 64), cdiv(M, 4), 1), dim3(256), 0, st, M, N, splits, ws, ep);
    return unetr_check_launch();
}

// ---- the large-tile form (M >= 1024 rows: batch >= 8 at 96^3, 160^3 at batch >= 2) -------------------------------------------
// 256 x 256 output tile, 64-deep K tiles, 8 waves as 2 (M) x 4 (N): a wave owns 128 x 64 = 8 x 4 MFMA tiles (128 accumulator
// registers).  LDS = two K-tile buffers of 64 KB, each four half-tiles of 128 rows x 128 B (A rows 0-127 / 128-255, B rows
// likewise) in the swizzled image of the kernel above, filled by LDS-DMA.
//
// Schedule ("ping-pong"): the waves of M-half 0 (group 0: waves 0-3) and of M-half 1 (group 1: waves 4-7) share the four SIMDs
// pairwise and run ONE barrier interval apart: while a group issues the 32 MFMAs of a 64 x 64 half of its tile (512 matrix
// pipe cycles), its SIMD partners read the fragments of their next half out of LDS and issue their share of the DMA -- the
// matrix pipe of every SIMD always has one wave in an MFMA segment.  Per K tile T a wave runs two phases, each
// {mem: ds_reads, two half-tiles of DMA, lgkmcnt(0), s_barrier} {compute: 32 MFMAs, s_barrier}:
//     P0: all 64 B columns (8 reads) + A rows 0-63 of its half (8 reads) -> rows 0-63;    DMA: both A halves of tile T+1
//     P1: A rows 64-127 (8 reads)                                        -> rows 64-127;  DMA: both B halves of tile T+2
// Intervals (one per barrier): group 0 runs mem(p) in interval 2p and compute(p) in 2p+1, group 1 one later (4 intervals per K
// tile; a four-phase variant with 16-MFMA quadrants -- twice the barriers -- measured 2-4 % slower).
// Why the DMA targets are free (write-after-read): B of buffer T%2 is last read in P0 of tile T (group 1: interval 4T+1, retired
// by its lgkmcnt(0) before that interval's barrier) and B of tile T+2 is issued from interval 4T+2 on; A half 0 (read by group 0
// only) is last read in interval 4(T-1)+2 and rewritten from 4T on, A half 1 (group 1) in 4(T-1)+3 / 4T.
// Why the reads see the data (read-after-write): at the end of P1's mem segment of tile T every wave waits until all its DMA
// older than the two B halves of tile T+2 has landed -- vmcnt(4): everything tile T+1 needs -- and the barrier that follows
// (interval 4T+3 ends) precedes the first read of tile T+1 (interval 4T+4).  Nothing in the loop drains to vmcnt(0) except the
// last two K tiles.
// Measured (MI355X, 6912 rows): steady state 1.72 us per K tile of 243 workgroups = 1.25 PFLOP/s (MFMA + DMA alone: 1.28 us --
// the matrix pipe at the clock the chip holds under this load); [6912 x 2304 x 768] bf16 out 30-32 us = 0.8 PFLOP/s, of which
// ~10 us is the 32 MB output burst of 243 workgroups finishing together (HBM write rate, whatever the store shape).
template <int BUF>
struct BufC { static constexpr int value = BUF; };

// Epilogue of the large-tile kernels.  The accumulator tile is C^T (swapped MFMA operands): lane (c, g) holds C[m = tile row c]
// [n = 4g .. 4g+3] of each of the wave's 8 x WN MFMA tiles (128 rows x 16 WN columns at row wr * 128, column wc * 16 WN of the
// workgroup tile).  `lds` = the K-tile buffers, idle by now (>= 8 x 16 KB).
template <int WN>
__device__ __forceinline__ void big_epilogue(f32x4 (&acc)[8][WN], const EpBf& ep, char* lds, int M, int N, int m0, int n0,
                                             int wave, int lane, int wr, int wc) {
    // The epilogue kind is chosen ONCE (a 32-tile loop over the general store4 is too large to unroll: the accumulators would go through scratch)
    const int mb = m0 + wr * 128 + (lane & 15), nb = n0 + wc * (16 * WN) + 4 * (lane >> 4);
    if (!ep.vec_ok) {
        for (int t = 0; t < 8 * WN; ++t) {
            const int i = t / WN, j = t - i * WN, m = mb + i * 16, n = nb + j * 16;
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
            for (int ii = 0; ii < 8; ++ii)
#pragma clang loop unroll(full)
                for (int jj = 0; jj < WN; ++jj) v = (ii == i && jj == j) ? acc[ii][jj] : v;
            if (m < M)
                for (int r = 0; r < 4; ++r)
                    if (n + r < N) ep.store(0, m, n + r, v[r]);
        }
        return;
    }
    // Row-coalesced epilogue through LDS.  In the accumulator layout a store instruction covers 16 rows x 64 B (32 B for bf16):
    // partial lines, 32 instructions per wave -- measured 11.6 us of a 32.7 us launch at [6912 x 2304] bf16 out.  Instead each wave
    // stages 64 rows x 64 columns of alpha * acc + bias (fp32) in its OWN 16 KB of the (now idle) K-tile buffers -- no workgroup
    // barrier: nobody reads the buffers any more (group 1's last fragment reads were retired before the barrier group 0 passed
    // last) and every DMA has landed (vmcnt(0) of the last K tile but one) -- and reads it back as whole rows: lane (l >> 4,
    // l & 15) owns 16 bytes of row 4 * it + (l >> 4), so pre / aux / residual / C are 256-byte and Cb 128-byte row segments.
    // Image: 256-byte rows, 16-byte chunk c of row r at slot c ^ (r & 7) (conflict-free ds_write_b128 of the accumulator layout).
    const int kind = (ep.act == 1 ? 1 : ep.act == 2 ? 2 : 0);
    char* const stg = lds + wave * 16384;
    const int rr = lane >> 4, rc = lane & 15;                 // read-back: row within a group of 4, 16-byte chunk
    const int ncol = n0 + wc * (16 * WN) + rc * 4;
    const bool cok = rc < 4 * WN && ncol < N;                 // this lane's 16-byte chunk exists (WN < 4: chunks 4 WN .. 15 are unused)
    f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (ep.bias && cok) bias4 = *(const f32x4*)(ep.bias + ncol);
    auto half = [&](auto kc, auto hc) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kc)::value, h = decltype(hc)::value;
#pragma clang loop unroll(full)
        for (int i = 0; i < 4; ++i)
#pragma clang loop unroll(full)
            for (int j = 0; j < WN; ++j) {
                const int r = i * 16 + (lane & 15), ch = j * 4 + (lane >> 4);
                *(f32x4*)(stg + r * 256 + ((ch ^ (r & 7)) << 4)) = acc[h * 4 + i][j] * ep.alpha;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma clang loop unroll(full)
        for (int it = 0; it < 16; ++it) {
            const int r = it * 4 + rr, m = m0 + wr * 128 + h * 64 + r;
            f32x4 v = *(const f32x4*)(stg + r * 256 + ((rc ^ (r & 7)) << 4)) + bias4;
            if (m < M && cok) {
                if (ep.pre) *(f32x4*)(ep.pre + (long)m * ep.ldc + ncol) = v;
                if constexpr (KIND == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
                } else if constexpr (KIND == 2) {
                    const f32x4 a = *(const f32x4*)(ep.aux + (long)m * ep.ldaux + ncol);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_fast(a[e]);
                }
                if (ep.res) v += *(const f32x4*)(ep.res + (long)(m % ep.res_mod) * ep.ldr + ncol);
                if (ep.C) {
                    if (ep.accumulate) v += *(const f32x4*)(ep.C + (long)m * ep.ldc + ncol);
                    *(f32x4*)(ep.C + (long)m * ep.ldc + ncol) = v;
                }
                if (ep.Cb) *(bf16x4*)(ep.Cb + (long)m * ep.ldcb + ncol) = __builtin_convertvector(v, bf16x4);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the second half overwrites the staging image)
    };
    if (kind == 1) { half(BufC<1>{}, BufC<0>{}); half(BufC<1>{}, BufC<1>{}); }
    else if (kind == 2) { half(BufC<2>{}, BufC<0>{}); half(BufC<2>{}, BufC<1>{}); }
    else { half(BufC<0>{}, BufC<0>{}); half(BufC<0>{}, BufC<1>{}); }
}

// WN = 16-column MFMA tiles per wave: the tile is 256 rows x 64 WN columns (256 / 192 / 128).  Narrower tiles exist for the
// tile COUNT: [6912 x 3072] is 324 tiles of 256 x 256 (two rounds of 256 CUs, the second 27 % full) but 432 of 256 x 192, and
// N = 768 is 81 / 162 tiles at WN 4 / 2.  The B region of a K tile is 64 WN rows = WN DMA pieces per thread.
template <int WN>
__global__ void __launch_bounds__(512, 2)
gemm_bf16_big_kernel(int M, int N, int K, int mt, int nt, int n_fast,
                     const uint16_t* __restrict__ A, long lda, const uint16_t* __restrict__ B, long ldb, EpBf ep) {
    constexpr int NT = 512, HALF = 128 * 128, BN = 64 * WN, BUFB = 2 * HALF + BN * 128;
    constexpr int LDSB = 2 * BUFB > 8 * 16384 ? 2 * BUFB : 8 * 16384;      // (the epilogue stages 16 KB per wave)
    __shared__ __attribute__((aligned(1024))) char lds[LDSB];

    // tile order: workgroup L runs on XCD L % 8; an XCD gets a contiguous run of tiles, fastest along the operand whose re-use
    // saves more traffic (n_fast: the run shares A rows and walks the weight columns; else the other way round)
    int tm, tn;
    {
        const int T = mt * nt, per = (T + 7) >> 3, L = blockIdx.x;
        const int t = (L & 7) * per + (L >> 3);
        if ((L >> 3) >= per || t >= T) return;
        if (n_fast) { tn = t % nt; tm = t / nt; } else { tm = t % mt; tn = t / mt; }
    }
    const int m0 = tm * 256, n0 = tn * BN;
    const int nk = K / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wave >> 2, wc = wave & 3;

    // DMA sources: piece i of a region covers row (tid + i * 512) >> 3 of the region, chunk (id & 7) ^ ((r >> 1) & 7).  A: two
    // half-tiles of 128 rows (2 pieces each); B: one region of BN rows (WN pieces)
    const uint16_t* asrc[2][2];
    const uint16_t* bsrc[WN];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + i * NT, r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
            asrc[h][i] = A + (long)min(m0 + h * 128 + r, M - 1) * lda + c * 8;
        }
#pragma unroll
    for (int i = 0; i < WN; ++i) {
        const int id = tid + i * NT, r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
        bsrc[i] = B + (long)min(n0 + r, N - 1) * ldb + c * 8;
    }
    auto dma = [&](const uint16_t* const (&src)[2], int T, char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] + (long)T * 64), (lds_void_t*)(dst + (wave * 64 + i * NT) * 16), 16, 0, 0);
    };
    auto dmab = [&](int T, char* dst) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < WN; ++i)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(bsrc[i] + (long)T * 64), (lds_void_t*)(dst + (wave * 64 + i * NT) * 16), 16, 0, 0);
    };
    // fragment read offsets inside a half-tile: row (tile * 16 + lane & 15), 16-byte chunk kh * 4 + (lane >> 4), swizzled; the
    // swizzle term does not depend on the tile index, so a tile is a constant 2048-byte step
    int off[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) off[kh] = lds_tile_off(lane & 15, kh * 4 + (lane >> 4));

    f32x4 acc[8][WN];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 af[4][2], bf[WN][2];
    // prologue: tile 0 whole, B of tile 1; tile 0 must have landed (and be published by the barrier) before the first read
    dma(asrc[0], 0, lds);
    dma(asrc[1], 0, lds + HALF);
    dmab(0, lds + 2 * HALF);
    if (nk > 1) {
        dmab(1, lds + BUFB + 2 * HALF);
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WN) : "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (wr == 1) asm volatile("s_barrier" ::: "memory");        // group 1 runs one interval behind group 0
    __builtin_amdgcn_sched_barrier(0);

#define BIG_MEM_END() do { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BIG_CMP_END(last_) do { __builtin_amdgcn_sched_barrier(0); if (!(last_)) asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define BIG_HALF(qa_) do {                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
                _Pragma("unroll") for (int j = 0; j < WN; ++j)                                       \
                    PrecBF16::mma(acc[(qa_) * 4 + i][j], bf[j][kh], af[i][kh]);                      \
        __builtin_amdgcn_s_setprio(0);                                                               \
    } while (0)

    auto ktile = [&](int T, auto bufc) __attribute__((always_inline)) {
        constexpr int b = decltype(bufc)::value;
        char* const cur = lds + b * BUFB;
        char* const oth = lds + (b ^ 1) * BUFB;
        const char* la = cur + wr * HALF;                                    // this wave's 128 A rows
        const char* lb = cur + 2 * HALF + wc * (16 * WN) * 128;              // this wave's 16 WN B rows
        const bool n1 = T + 1 < nk, n2 = T + 2 < nk;
        {
            // two phases of 32 MFMAs (half the barriers): P0 reads all of B and A rows 0-63 and issues both A halves of tile T+1,
            // P1 reads A rows 64-127 and issues both B halves of tile T+2 (B of buffer T%2 was last read in P0 of this tile)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                for (int j = 0; j < WN; ++j) bf[j][kh] = *(const u32x4*)(lb + off[kh] + j * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][kh] = *(const u32x4*)(la + off[kh] + i * 2048);
            }
            if (n1) { dma(asrc[0], T + 1, oth); dma(asrc[1], T + 1, oth + HALF); }
            BIG_MEM_END();
            BIG_HALF(0);
            BIG_CMP_END(false);
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i][kh] = *(const u32x4*)(la + off[kh] + (4 + i) * 2048);
            }
            if (n2) {
                dmab(T + 2, cur + 2 * HALF);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN) : "memory");
            } else if (n1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            BIG_MEM_END();
            BIG_HALF(1);
            BIG_CMP_END(!n1 && wr == 1);        // group 1 skips its very last barrier: both groups execute the same number
        }
    };
    for (int T = 0; T < nk; T += 2) {
        ktile(T, BufC<0>{});
        if (T + 1 < nk) ktile(T + 1, BufC<1>{});
    }
#undef BIG_MEM_END
#undef BIG_CMP_END
#undef BIG_HALF

    big_epilogue<WN>(acc, ep, lds, M, N, m0, n0, wave, lane, wr, wc);
}

static int launch_bf16_big(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, const EpBf& ep, hipStream_t st, int wn) {
    const int BN = 64 * wn;
    const int mt = cdiv(M, 256), nt = cdiv(N, BN);
    const long tiles = (long)mt * nt;
    // traffic of the two tile orders, in rows of K elements fetched per XCD pass: m fastest re-reads A for every weight column
    // tile an XCD touches (up to 8 XCDs share the grid), n fastest re-reads the weights
    const long cost_m = (long)M * std::min(8, nt) + N, cost_n = M + (long)N * std::min(8, mt);
    const int n_fast = cost_n < cost_m;
    const int per = cdiv(tiles, 8);
#define BIG_GO(W_) hipLaunchKernelGGL(gemm_bf16_big_kernel<W_>, dim3(per * 8), dim3(512), 0, st, M, N, K, mt, nt, n_fast, A, lda, B, ldb, ep)
    if (wn == 2) BIG_GO(2); else if (wn == 3) BIG_GO(3); else BIG_GO(4);
#undef BIG_GO
    return unetr_check_launch();
}

// which large-tile width (columns = 64 wn) serves [M, N]: the one with the least estimated time = rounds of 256 CUs x time of one
// tile (K-loop share ~ wn + 1: MFMAs scale with wn, the A half of the DMA and the fragment reads do not; + a fixed prologue /
// epilogue share).  0 = the 128 x 128 family (too few large tiles to fill the chip).
static int big_tile_width(int M, int N, int K) {
    if (const char* e = getenv("UNETR_GEMM_BIG_WN")) { const int v = atoi(e); if (v >= 2 && v <= 4) return v; }
    int best = 0; double best_t = 1e30;
    for (int wn = 4; wn >= 2; --wn) {
        const long tiles = (long)cdiv(M, 256) * cdiv(N, 64 * wn);
        if (tiles < 128) continue;
        // (measured at 6912 rows: the 128-wide tile beats the 128 x 128 family only on long reductions -- N = 768: K = 4096
        // 60 vs 68 us, K = 3072 62 vs 62, K = 768 34 vs 30)
        if (wn == 2 && K < 2048) continue;
        const double t = (double)cdiv(tiles, 256) * (wn + 1.5);
        if (t < best_t - 1e-9) { best_t = t; best = wn; }
    }
    return best;
}

// ---- grouped weight-gradient GEMM on bf16-stored operands: dW_i[N_i, K_i] = dY_i[M, N_i]^T * X_i[M, K_i] ------------------
// Both operands are reduction-major ([token][feature]), i.e. the b_kn layout on BOTH sides: a stage is 64 tokens x 128
// features of dY and of X, staged by LDS-DMA into two swizzled images (bkn_x<16>), and all MFMA fragments (k = token) come
// out through ds_read_b64_tr_b16.  One launch covers up to 48 problems (descriptors in the kernel arguments); tokens beyond
// M in the last stage read a clamped row and are zeroed in the A fragments (M is a multiple of 8: whole lane groups).
#ifndef GW_WAVES
#define GW_WAVES 4
#endif
constexpr int GW_MAX = 64;     // 12 blocks x 4 Linear layers + patch embedding in ONE launch
struct GwProblem { const uint16_t* dy; const uint16_t* x; float* dw; int M, N, K, tile0, ntn, step_idx; };
// FUSE: the arenas of the fused optimizer epilogue (dw then only NAMES the arena slice: element offset = dw - gb)
struct GwFuse { float* pb; const float* gb; float* mb; float* vb; uint16_t* sb; const float* steps; float lr, b1, b2, eps, wd; uint32_t* sw; };
struct GwArgs { int n; GwFuse f; GwProblem p[GW_MAX]; };

template <int NS, int BKT, int FUSE>                        // BKT = tokens per stage (64 or 32); FUSE: 0 store dW, 1 AdamW, 2 store bf16(dW)
__global__ void __launch_bounds__(256, GW_WAVES)
gemm_bf16_grouped_wgrad_kernel(GwArgs ga) {
    constexpr int BT = 128;                                // output tile 128 x 128
    constexpr int IMG = BKT * BT * 2, STAGE = 2 * IMG, PCS = IMG / 16 / 256, G = 2 * PCS, CPR = 16;
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];
    int pi = 0;
    const int t = blockIdx.x;
    for (int hi_ = ga.n - 1; pi < hi_;) {      // last problem whose first tile <= t (binary search over the kernel-argument table)
        const int mid = (pi + hi_ + 1) >> 1;
        if (t >= ga.p[mid].tile0) pi = mid; else hi_ = mid - 1;
    }
    const GwProblem& pr = ga.p[pi];
    const int lt = t - pr.tile0, tn = lt % pr.ntn, tk = lt / pr.ntn;
    const int n0 = tn * BT, k0 = tk * BT, M = pr.M;
    const int nk = (M + BKT - 1) / BKT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;

    // per-thread source offsets (row inside the stage, clamped column) of the DMA pieces
    int prow[PCS], acol[PCS], bcol[PCS];
#pragma unroll
    for (int i = 0; i < PCS; ++i) {
        const int id = tid + i * 256, r = id / CPR, sl = id % CPR, c = sl ^ bkn_x<CPR>(r);
        prow[i] = r;
        acol[i] = min(n0 + c * 8, pr.N - 8);
        bcol[i] = min(k0 + c * 8, pr.K - 8);
    }
    auto issue = [&](int kt, int buf) {
        char* la = lds + buf * STAGE;
        char* lb = la + IMG;
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const long row = min(kt * BKT + prow[i], M - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(pr.dy + row * pr.N + acol[i]), (lds_void_t*)(la + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            const long row = min(kt * BKT + prow[i], M - 1);
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(pr.x + row * pr.K + bcol[i]), (lds_void_t*)(lb + (wave * 64 + i * 256) * 16), 16, 0, 0);
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) issue(s, s);
    const int cc = lane & 15, g = lane >> 4, q = cc >> 2, p = cc & 3;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + NS - 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NS - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NS - 1 < nk) issue(kt + NS - 1, (kt + NS - 1) % NS);
        const char* la = lds + (kt % NS) * STAGE;
        const char* lb = la + IMG;
#pragma unroll
        for (int kb = 0; kb < BKT / 32; ++kb) {
            const int r0 = kb * 32 + 8 * g + q, r1 = r0 + 4;
            const int x0s = bkn_x<CPR>(r0), x1s = bkn_x<CPR>(r1);
            const bool live = kt * BKT + kb * 32 + 8 * g < M;          // this lane group's 8 tokens exist
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = (wm * 4 + i) * 2 + (p >> 1);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(la + r0 * (BT * 2) + (ch ^ x0s) * 16 + (p & 1) * 8));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(la + r1 * (BT * 2) + (ch ^ x1s) * 16 + (p & 1) * 8));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const u32x4 w = __builtin_bit_cast(u32x4, v);
                a[i] = live ? w : (u32x4){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = (wn * 4 + j) * 2 + (p >> 1);
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(lb + r0 * (BT * 2) + (ch ^ x0s) * 16 + (p & 1) * 8));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(lb + r1 * (BT * 2) + (ch ^ x1s) * 16 + (p & 1) * 8));
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                b[j] = __builtin_bit_cast(u32x4, v);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) PrecBF16::mma(acc[i][j], b[j], a[i]);    // swapped operands: the accumulator tile is dW^T
        }
    }
    // Epilogue.  In the accumulator layout lane (cc, g) holds dW[n = tile row cc][k = 4g .. 4g+3]: a store instruction would touch
    // 64-byte pieces of 16 rows.  Each wave instead turns its 64 x 64 sub-tile through its own 8 KB of the (now free) stage
    // buffers, half at a time, so that 16 lanes cover 256 contiguous bytes of a row (chunk c of staged row r at slot c ^ (r & 15):
    // conflict-free for the row-per-lane writes): the plain gradient store went from 0.213 to 0.180 ms that way, and the fused
    // optimizer epilogue from 0.76 to 0.44 ms.
    static_assert(NS * STAGE >= 4 * 8192, "8 KB of staging per wave");
    __syncthreads();                                       // every wave has read its last fragments out of the stage buffers
    char* stg = lds + wave * 8192;
    const int rr = lane >> 4, rc = lane & 15;              // reader: row within a group of 4, 16-byte chunk of the 256-byte row
    const int kcol = k0 + wn * 64 + rc * 4;
    [[maybe_unused]] AdamWCoef c{};
    [[maybe_unused]] long off0 = 0;
    if constexpr (FUSE == 1) {
        c = adamw_coef(ga.f.lr, ga.f.b1, ga.f.b2, ga.f.eps, ga.f.wd, ga.f.steps[pr.step_idx]);
        off0 = pr.dw - ga.f.gb;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = ii * 16 + cc;
                *(f32x4*)(stg + r * 256 + (((j * 4 + g) ^ (r & 15)) << 4)) = acc[2 * h + ii][j];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (wave-private region: no barrier)
        if constexpr (FUSE == 2) {
            // data-parallel step with bf16 gradient communication: the gradient goes straight into the communication buffer (laid
            // out like the arena) as bf16 -- what the separate cast pass would have made of the fp32 store, which never happens
            const long offg = pr.dw - ga.f.gb;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + rr, n = n0 + wm * 64 + h * 32 + r;
                const f32x4 gv = *(const f32x4*)(stg + r * 256 + ((rc ^ (r & 15)) << 4));
                if (n < pr.N && kcol < pr.K) *(bf16x4*)(ga.f.sb + offg + (long)n * pr.K + kcol) = __builtin_convertvector(gv, bf16x4);
            }
        } else if constexpr (FUSE == 0) {
            // dW is written once and next read by AdamW after 350 MB of other gradients: non-temporal
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 4 + rr, n = n0 + wm * 64 + h * 32 + r;
                const f32x4 gv = *(const f32x4*)(stg + r * 256 + ((rc ^ (r & 15)) << 4));
                if (n < pr.N && kcol < pr.K) __builtin_nontemporal_store(gv, (f32x4*)(pr.dw + (long)n * pr.K + kcol));
            }
        } else {
            // AdamW on the tile instead of the gradient store: the weight's master / moment / shadow slices sit at the same arena
            // offset as dw; p / m / v are streamed once (non-temporal), the bf16 shadow is what the next forward reads
            const GwFuse& f = ga.f;
#pragma unroll 2
            for (int it = 0; it < 8; it += 2) {
                f32x4 gv[2], pv[2], mv[2], vv[2];
                long idx[2];
                bool ok[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = (it + u) * 4 + rr, n = n0 + wm * 64 + h * 32 + r;
                    ok[u] = n < pr.N && kcol < pr.K;
                    idx[u] = off0 + (long)n * pr.K + kcol;
                    gv[u] = *(const f32x4*)(stg + r * 256 + ((rc ^ (r & 15)) << 4));
                    if (ok[u]) {
                        pv[u] = __builtin_nontemporal_load((const f32x4*)(f.pb + idx[u]));
                        mv[u] = __builtin_nontemporal_load((const f32x4*)(f.mb + idx[u]));
                        vv[u] = __builtin_nontemporal_load((const f32x4*)(f.vb + idx[u]));
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!ok[u]) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float pe = pv[u][e], me = mv[u][e], ve = vv[u][e];
                        adamw_elem(pe, me, ve, gv[u][e], c);
                        pv[u][e] = pe; mv[u][e] = me; vv[u][e] = ve;
                    }
                    __builtin_nontemporal_store(pv[u], (f32x4*)(f.pb + idx[u]));
                    __builtin_nontemporal_store(mv[u], (f32x4*)(f.mb + idx[u]));
                    __builtin_nontemporal_store(vv[u], (f32x4*)(f.vb + idx[u]));
                    if (f.sb) *(bf16x4*)(f.sb + idx[u]) = __builtin_convertvector(pv[u], bf16x4);
                    if (f.sw) *(u32x4*)(f.sw + idx[u]) = x3_words(__builtin_bit_cast(u32x4, pv[u]));      // (bf16x3 mode: the word shadow)
                }
            }
        }
        if (h == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of this half are done before it is overwritten
    }
}

__global__ void __launch_bounds__(256) cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n8) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = ((const f32x4*)src)[2 * i], b = ((const f32x4*)src)[2 * i + 1];
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        ((u32x4*)dst)[i] = PrecBF16::pack(v);
    }
}

// out[i] = a[i] + b[i] (fp32) and its bf16 copy: the gradient of a hidden state with two consumers (the next transformer
// block and a skip-path transposed conv, unetr.py:197-201) as ONE launch -- autograd's own sum was an elementwise add kernel
// followed by a cast kernel for the bf16 operand of the block's backward GEMMs
__global__ void __launch_bounds__(256) add_cast_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                            uint16_t* __restrict__ out16, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)a)[i] + ((const f32x4*)b)[i];
        ((f32x4*)out)[i] = v;
        if (out16) ((bf16x4*)out16)[i] = __builtin_convertvector(v, bf16x4);
    }
}

// bf16x3 weight gradients on the bf16 kernels: dW = dY^T X with dY = dYh + dYl, X = Xh + Xl (bf16 halves of the fp32 values) is
// [dYh; dYh; dYl]^T [Xh; Xl; Xh] -- the same contraction over THREE times the rows.  This kernel writes one such stack from an
// fp32 [M, C] matrix: second == 0: rows [hi; hi; lo] (the dY side), second == 1: rows [hi; lo; hi] (the X side).
__global__ void __launch_bounds__(256) split_stack_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n8, long plane, int second) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = ((const f32x4*)src)[2 * i], b = ((const f32x4*)src)[2 * i + 1];
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}, r[8];
        const u32x4 hi = PrecBF16::pack(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[2 * e] = v[2 * e] - __builtin_bit_cast(float, hi[e] << 16);
            r[2 * e + 1] = v[2 * e + 1] - __builtin_bit_cast(float, hi[e] & 0xffff0000u);
        }
        const u32x4 lo = PrecBF16::pack(r);
        ((u32x4*)dst)[i] = hi;
        ((u32x4*)(dst + plane))[i] = second ? lo : hi;
        ((u32x4*)(dst + 2 * plane))[i] = second ? hi : lo;
    }
}

// every stack of a backward pass in one launch (98 launches of ~5 us each before: one per operand)
constexpr int SS_MAX = 64;
struct SsProblem { const float* src; uint16_t* dst; long n8; int second; int blk0; int nblk; };
struct SsArgs { int n; SsProblem p[SS_MAX]; };
__global__ void __launch_bounds__(256) split_stack_grouped_kernel(SsArgs a) {
    int pi = 0, hi_ = a.n - 1;
    while (pi < hi_) {
        const int mid = (pi + hi_ + 1) >> 1;
        if ((int)blockIdx.x >= a.p[mid].blk0) pi = mid; else hi_ = mid - 1;
    }
    const SsProblem& pr = a.p[pi];
    const float* __restrict__ src = pr.src;
    uint16_t* __restrict__ dst = pr.dst;
    const long n8 = pr.n8, plane = n8 * 8;
    const int second = pr.second;
    for (long i = ((long)blockIdx.x - pr.blk0) * 256L + threadIdx.x; i < n8; i += (long)pr.nblk * 256) {
        const f32x4 x = ((const f32x4*)src)[2 * i], y = ((const f32x4*)src)[2 * i + 1];
        float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]}, r[8];
        const u32x4 hi = PrecBF16::pack(v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[2 * e] = v[2 * e] - __builtin_bit_cast(float, hi[e] << 16);
            r[2 * e + 1] = v[2 * e + 1] - __builtin_bit_cast(float, hi[e] & 0xffff0000u);
        }
        const u32x4 lo = PrecBF16::pack(r);
        ((u32x4*)dst)[i] = hi;
        ((u32x4*)(dst + plane))[i] = second ? lo : hi;
        ((u32x4*)(dst + 2 * plane))[i] = second ? hi : lo;
    }
}

__global__ void __launch_bounds__(256) cast_bf16_tail_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long beg, long n) {
    const long i = beg + blockIdx.x * 256L + threadIdx.x;
    if (i < n) { __bf16 h = (__bf16)src[i]; dst[i] = __builtin_bit_cast(uint16_t, h); }
}

}  // namespace

static int gemm_bf16_impl(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C, void* Cb,
                          float* ws, size_t ws_bytes, void* stream, int* psp) {
    if (!d || !A || !B || (!C && !Cb)) return UNETR_ERR_ARG;
    if (d->x3) {
        // bf16x3 mode through the same entry points (the LayerNorm-riding forms below): fp32 A, B fp32 (x3 = 1) or pre-split words (2)
        if (Cb || !C || d->tc_cout > 0 || (d->x3 != 1 && d->x3 != 2)) return UNETR_ERR_ARG;
        return unetr_gemm_x3_dma(d, (const float*)A, (const float*)B, d->x3 == 2, C, ws, ws_bytes, stream, psp);
    }
    const int M = d->M, N = d->N, K = d->K;
    if (M <= 0 || N <= 0 || K <= 0) return UNETR_ERR_ARG;
    // whole 64-deep K stages, 16-byte aligned rows; the transposed-B form also needs whole 8-column chunks
    if (K % 64 || d->lda % 8 || d->ldb % 8 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return UNETR_ERR_UNSUPPORTED;
    if (d->b_kn && (N % 8 || N < 8)) return UNETR_ERR_UNSUPPORTED;
    if (d->act == 2 && !d->aux) return UNETR_ERR_ARG;
    if (d->accumulate && !C) return UNETR_ERR_ARG;       // `pre` (pitch ldc) may be written without C
    hipStream_t st = (hipStream_t)stream;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const int vec_ok = (N % 4 == 0) && (!C || (d->ldc % 4 == 0 && al16(C))) && (!Cb || (d->ldcb % 4 == 0 && ((uintptr_t)Cb & 7) == 0)) &&
                       (!d->bias || al16(d->bias)) && (!d->res || (d->ldr % 4 == 0 && al16(d->res))) && (!d->pre || al16(d->pre)) &&
                       (!d->aux || (d->ldaux % 4 == 0 && al16(d->aux))) && al16(ws);
    // transposed-conv scatter: bf16 output only, whole 4-column groups inside one tap, rows = voxels of the input grid
    if (d->tc_cout > 0 && (C || !Cb || d->pre || d->accumulate || d->tc_cout % 4 || N != 8 * d->tc_cout || d->tc_d <= 0 || d->tc_h <= 0 ||
                           d->tc_w <= 0 || M % (d->tc_d * d->tc_h * d->tc_w))) return UNETR_ERR_UNSUPPORTED;
    EpBf ep{vec_ok, C, d->ldc, (uint16_t*)Cb, d->ldcb, d->bias, d->res, d->ldr, d->res_mod > 0 ? d->res_mod : M,
            d->pre, d->aux, d->ldaux, d->act, d->accumulate, d->alpha, d->tc_d, d->tc_h, d->tc_w, d->tc_cout > 0 ? d->tc_cout : 0};
    const uint16_t* a = (const uint16_t*)A;
    const uint16_t* b = (const uint16_t*)B;
    const int env_cfg = getenv("UNETR_GEMM_CFG") ? atoi(getenv("UNETR_GEMM_CFG")) : 0;   // tuning hooks
    const int env_ns = getenv("UNETR_GEMM_STAGES") ? atoi(getenv("UNETR_GEMM_STAGES")) : 0;
    // (short reductions with few 128 x 128 tiles -- the GEMM-form transposed convs at 12^3 x 2 = 3456 voxels: [3456 x 128 x 512] is 27
    // tiles, [3456 x 512 x 64 / 128] 108 -- take the small-M tiles: 20-24 us per launch on 27-108 workgroups with the 128 x 128 family)
    const bool few128 = K <= 512 && (long)cdiv(M, 128) * cdiv(N, 128) < 128 && !getenv("UNETR_GEMM_FEW128_OFF");
    const bool big = env_cfg == 128 || (env_cfg == 0 && M >= 1024 && N >= 128 && !few128);
    // the 256 x 256 ping-pong kernel: many rows, weights as stored ([N, K]); K tiles of 64 (checked above).  Narrow outputs
    // (N = 768 at 6912 rows: 81 tiles for 256 CUs) keep the 128 x 128 tile, which fills the chip
    if (!d->b_kn && d->tc_cout <= 0 && (env_cfg == 256 || (env_cfg == 0 && M >= 1024))) {
        const int wn = env_cfg == 256 ? (getenv("UNETR_GEMM_BIG_WN") ? big_tile_width(M, N, K) : 4) : big_tile_width(M, N, K);
        if (wn) return launch_bf16_big(M, N, K, a, d->lda, b, d->ldb, ep, st, wn);
    }
#define BF16_GO(WM_, WN_, BKN_, NS_) return launch_bf16<WM_, WN_, 2, 2, BKN_, NS_>(M, N, K, a, d->lda, b, d->ldb, ep, ws, ws_bytes, st, psp)
    // Small token counts (batch 2: M = 432).  Measured per launch on MI355X (tools/probe_encoder.py, us incl. launch boundary):
    //   forward, K = 768:   N = 768: 64x32 5.6 < 32x64 5.7 < 64x64 7.1;  N = 2304: 64x64 6.6 < 64x96 7.5 < 64x128 9.1;
    //                       N = 3072: 64x64 10.5 < 64x96 11.2 < 64x128 13.7
    //   forward, K = 3072, N = 768: 64x64 + 4 split-K slabs 12.5 ~ 64x32 13.0 < 32x64 13.6 < 64x64 unsplit 16.3
    //   [K,N]-operand data gradients: K = 768: 32x64 (N = 768: 6.4, N = 3072: 11.8) < 64x64 (7.3, 13.1) < 64x128;
    //                                 K = 3072, N = 768: 64x64 + split-K 14.4 < 64x128 + split-K 16.6 < 32x64 24.9
    // i.e. short reductions want the most workgroups (a CU pulls only ~70 GB/s from L2), long ones the 64x64 tile cut into
    // K slabs.  cfg codes (UNETR_GEMM_CFG): 6464, 6432, 3264, 64128, 6496.
    int cfg = env_cfg;
    if (!big && (cfg == 0 || cfg == 64)) {
        cfg = 6464;
        if (env_cfg == 0 && K <= 1024 && N <= 1024) cfg = d->b_kn ? 3264 : 6432;
        if (env_cfg == 0 && K <= 1024 && d->b_kn && N > 1024 && N % 64 == 0) cfg = 3264;
        // 512 < M < 1024 (batch 4 at 96^3: 864 rows, the ranking pre-training step; 160^3: 1000 rows): two 64 x 64 tiles per CU move
        // more bytes per CU than one wider tile.  Measured (tools/probe_encoder.py PROBE_M=864 / 1000, us): linear1 64x96 13.1 / 13.7
        // < 64x128 15.4 / 15.8 < 64x64 16.1 / 16.5; qkv at 1000 rows 64x96 10.6 < 64x128 12.5 < 64x64 14.1 (at 864: 64x64 9.8 <
        // 64x96 10.4); linear2 64x32 unsplit 14.7 / 15.6 < 64x64 + split-K 19.7 / 16.2; data gradients: K = 3072 64x128 + split-K
        // 21.0 / 21.2 < 64x64 23.3 / 30.3, N = 3072 64x128 19.6 / 19.9 < 32x64 20.4 / 21.5
        if (env_cfg == 0 && M > 512) {
            if (!d->b_kn) {
                if (N >= 3072 || (N >= 2304 && M >= 960)) cfg = 6496;
                else if (K >= 2048 && N <= 1024) cfg = 6432;
            } else if (N % 128 == 0) {
                if ((K >= 2048 && N <= 1024) || (K <= 1024 && N > 1024)) cfg = 64128;
            }
        }
    }
    if (!d->b_kn) {
        if (big) { if (env_ns == 3) BF16_GO(4, 4, false, 3); if (env_ns == 4) BF16_GO(4, 4, false, 4); BF16_GO(4, 4, false, 2); }
        if (cfg == 6432) { if (env_ns == 6) BF16_GO(2, 1, false, 6); if (env_ns == 8) BF16_GO(2, 1, false, 8); if (env_ns == 12) BF16_GO(2, 1, false, 12); BF16_GO(2, 1, false, 4); }
        if (cfg == 3264) { if (env_ns == 6) BF16_GO(1, 2, false, 6); if (env_ns == 8) BF16_GO(1, 2, false, 8); BF16_GO(1, 2, false, 4); }
        if (cfg == 64128) BF16_GO(2, 4, false, 3);
        if (cfg == 6496) BF16_GO(2, 3, false, 3);
        if (env_ns == 2) BF16_GO(2, 2, false, 2);
        if (env_ns == 6) BF16_GO(2, 2, false, 6);
        if (env_ns == 8) BF16_GO(2, 2, false, 8);
        BF16_GO(2, 2, false, 4);
    }
    if (big) { if (env_ns == 3) BF16_GO(4, 4, true, 3); if (env_ns == 4) BF16_GO(4, 4, true, 4); BF16_GO(4, 4, true, 2); }
    if (cfg == 3264) { if (env_ns == 6) BF16_GO(1, 2, true, 6); if (env_ns == 8) BF16_GO(1, 2, true, 8); if (env_ns == 12) BF16_GO(1, 2, true, 12); BF16_GO(1, 2, true, 4); }
    if (cfg == 64128) BF16_GO(2, 4, true, 3);
    if (env_ns == 2) BF16_GO(2, 2, true, 2);
    if (env_ns == 6) BF16_GO(2, 2, true, 6);
    if (env_ns == 8) BF16_GO(2, 2, true, 8);
    BF16_GO(2, 2, true, 4);
#undef BF16_GO
}

// bf16x3 precision mode: C = A . B^T (b_kn = 0, B [N,K]) or A . B (b_kn = 1, B [K,N]) on fp32-stored operands through the LDS-DMA kernel
// above (X3 instantiations).  Called by unetr_gemm (gemm_std.hip) for the plain Linear shapes; UNSUPPORTED = the generic family.
int unetr_gemm_x3_dma(const unetr_gemm_bf16_desc* d, const float* A, const float* B, int b_words, float* C, float* ws, size_t ws_bytes, void* stream,
                      int* psp) {
    const int M = d->M, N = d->N, K = d->K;
    if (M <= 0 || N <= 0 || K <= 0) return UNETR_ERR_ARG;
    if (K % 32 || d->lda % 4 || d->ldb % 4 || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || N % 4) return UNETR_ERR_UNSUPPORTED;
    if (d->b_kn && N < 64) return UNETR_ERR_UNSUPPORTED;
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    const int vec_ok = (d->ldc % 4 == 0 && al16(C)) && (!d->bias || al16(d->bias)) && (!d->res || (d->ldr % 4 == 0 && al16(d->res))) &&
                       (!d->pre || al16(d->pre)) && (!d->aux || (d->ldaux % 4 == 0 && al16(d->aux))) && al16(ws);
    if (!vec_ok) return UNETR_ERR_UNSUPPORTED;
    EpBf ep{1, C, d->ldc, nullptr, 0, d->bias, d->res, d->ldr, d->res_mod > 0 ? d->res_mod : M, d->pre, d->aux, d->ldaux, d->act,
            d->accumulate, d->alpha, 0, 0, 0, 0};
    hipStream_t st = (hipStream_t)stream;
#define X3_GO(WM_, WN_, BKN_, NS_) do { if (b_words) return launch_bf16<WM_, WN_, 2, 2, BKN_, NS_, 2>(M, N, K, A, d->lda, B, d->ldb, ep, ws, ws_bytes, st, psp); \
                                        return launch_bf16<WM_, WN_, 2, 2, BKN_, NS_, 1>(M, N, K, A, d->lda, B, d->ldb, ep, ws, ws_bytes, st, psp); } while (0)
    // tile rule of the bf16 kernel at small M (a stage is the same 16 KB); many rows: the 128 x 128 tile
    if (M >= 1024 && N >= 128) { if (d->b_kn) X3_GO(4, 4, true, 2); X3_GO(4, 4, false, 2); }
    if (!d->b_kn) {
        if (K <= 1024 && N <= 1024) X3_GO(2, 1, false, 4);
        X3_GO(2, 2, false, 4);
    }
    if (K <= 1024 && N % 64 == 0) X3_GO(1, 2, true, 4);
    X3_GO(2, 2, true, 4);
#undef X3_GO
}

extern "C" int unetr_gemm_bf16(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C, void* Cb,
                               float* ws, size_t ws_bytes, void* stream) {
    return gemm_bf16_impl(d, A, B, C, Cb, ws, ws_bytes, stream, nullptr);
}

// dx = LayerNorm backward of dy = A . B (a plain product: alpha 1, no bias / activation / residual) in two launches: when the
// GEMM is cut into K slabs (the batch-2 data gradients with K = 2304 / 3072) the LayerNorm kernel sums the slabs itself, in the
// order the separate reduce launch would; otherwise the product lands in the scratch matrix C [M, N] first.
extern "C" int unetr_gemm_bf16_ln_bwd(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C,
                                      const float* x, const float* gamma, const float* mean, const float* rstd,
                                      float* dx, void* dx_bf16, const float* dres, float* dgamma, float* dbeta,
                                      float* ln_ws, size_t ln_ws_bytes, float* ws, size_t ws_bytes, void* stream) {
    if (!d || !C) return UNETR_ERR_ARG;
    if (d->bias || d->res || d->pre || d->act || d->accumulate || d->alpha != 1.f || d->ldc != d->N) return UNETR_ERR_UNSUPPORTED;
    int splits = 1;
    if (int e = gemm_bf16_impl(d, A, B, C, nullptr, ws, ws_bytes, stream, &splits)) return e;
    const float* dy = splits > 1 ? ws : C;
    return unetr_layernorm_bwd_partials(dy, splits, (long)d->M * d->N, x, gamma, mean, rstd, dx, dx_bf16, dres, dgamma, dbeta,
                                        d->M, d->N, ln_ws, ln_ws_bytes, stream);
}

// C = A . B + bias + res (the residual-stream output of a block's last Linear) AND the LayerNorm of C that the next layer starts
// with, in two launches: when the GEMM is cut into K slabs the LayerNorm kernel sums the slabs, applies bias and residual, writes
// C and normalises the row it has just formed; otherwise the GEMM writes C through its own epilogue and the plain LayerNorm runs.
extern "C" int unetr_gemm_bf16_ln_fwd(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C,
                                      const float* gamma, const float* beta, float eps, float* y, void* y_bf16, float* mean, float* rstd,
                                      float* ws, size_t ws_bytes, void* stream) {
    if (!d || !C) return UNETR_ERR_ARG;
    if (d->pre || d->act || d->accumulate || d->alpha != 1.f || d->ldc != d->N) return UNETR_ERR_UNSUPPORTED;
    int splits = 1;
    if (int e = gemm_bf16_impl(d, A, B, C, nullptr, ws, ws_bytes, stream, &splits)) return e;
    if (splits > 1)
        return unetr_layernorm_fwd_partials(ws, splits, (long)d->M * d->N, d->bias, d->res, d->ldr, d->res_mod > 0 ? d->res_mod : d->M,
                                            C, gamma, beta, y, y_bf16, mean, rstd, d->M, d->N, eps, stream);
    return unetr_layernorm_fwd(C, gamma, beta, y, y_bf16, mean, rstd, d->M, d->N, eps, stream);
}

// fp32 -> bf16 (round to nearest even), the weight shadow / activation cast
extern "C" int unetr_cast_bf16(const float* src, void* dst, long n, void* stream) {
    if (!src || !dst || n < 0) return UNETR_ERR_ARG;
    if (n == 0) return UNETR_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (((uintptr_t)src & 15) | ((uintptr_t)dst & 15)) == 0;
    const long n8 = al ? n / 8 : 0;
    if (n8) hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)std::min<long>(cdiv(n8, 256), 4096)), dim3(256), 0, st, src, (uint16_t*)dst, n8);
    if (n8 * 8 < n) hipLaunchKernelGGL(cast_bf16_tail_kernel, dim3(cdiv(n - n8 * 8, 256)), dim3(256), 0, st, src, (uint16_t*)dst, n8 * 8, n);
    return unetr_check_launch();
}

// fp32 -> split words [hi | lo << 16] (x3_words): the word shadow of a weight arena
__global__ void __launch_bounds__(256) split_words_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, long n4) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) ((u32x4*)dst)[i] = x3_words(((const u32x4*)src)[i]);
}

extern "C" int unetr_split_words(const float* src, void* dst, long n, void* stream) {
    if (!src || !dst || n <= 0) return UNETR_ERR_ARG;
    if ((n & 3) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return UNETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(split_words_kernel, dim3((unsigned)std::min<long>(cdiv(n / 4, 256), 8192)), dim3(256), 0, (hipStream_t)stream, src, (uint32_t*)dst, n / 4);
    return unetr_check_launch();
}

extern "C" int unetr_split_stack_bf16(const float* src, void* dst, long rows, long cols, int second, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return UNETR_ERR_ARG;
    const long n = rows * cols;
    if ((n & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return UNETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(split_stack_kernel, dim3((unsigned)std::min<long>(cdiv(n / 8, 256), 4096)), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)dst,
                       n / 8, n, second);
    return unetr_check_launch();
}

extern "C" int unetr_split_stack_bf16_grouped(const unetr_split_problem* probs, int n, void* stream) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    for (int base = 0; base < n; base += SS_MAX) {
        SsArgs a;
        a.n = std::min(SS_MAX, n - base);
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const unetr_split_problem& q = probs[base + i];
            const long ne = q.rows * q.cols;
            if (!q.src || !q.dst || q.rows <= 0 || q.cols <= 0) return UNETR_ERR_ARG;
            if ((ne & 7) || (((uintptr_t)q.src | (uintptr_t)q.dst) & 15)) return UNETR_ERR_UNSUPPORTED;
            const int nb = (int)std::min<long>(cdiv(ne / 8, 256), 1024);
            a.p[i] = SsProblem{q.src, (uint16_t*)q.dst, ne / 8, q.second, blocks, nb};
            blocks += nb;
        }
        hipLaunchKernelGGL(split_stack_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return unetr_check_launch();
}

// fp32 -> bf16 over a table of arena ranges (DEVICE memory, 3 longs per range: element offsets lo, hi -- multiples of 8 -- and the
// range's first block of 8192 elements): the gradient ranges the bf16-storing weight-gradient epilogue did not cover
__global__ void __launch_bounds__(256) cast_bf16_ranges_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                               const long* __restrict__ table, int nr) {
    int lo_r = 0, hi_r = nr - 1;
    const long blk = blockIdx.x;
    while (lo_r < hi_r) {
        const int mid = (lo_r + hi_r + 1) >> 1;
        if (table[3 * mid + 2] <= blk) lo_r = mid; else hi_r = mid - 1;
    }
    const long beg = table[3 * lo_r] + (blk - table[3 * lo_r + 2]) * 8192, end = min(table[3 * lo_r + 1], beg + 8192);
    for (long i = beg / 8 + threadIdx.x; i < end / 8; i += 256) {
        const f32x4 a = ((const f32x4*)src)[2 * i], b = ((const f32x4*)src)[2 * i + 1];
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        ((u32x4*)dst)[i] = PrecBF16::pack(v);
    }
}

extern "C" int unetr_cast_bf16_ranges(const float* src_arena, void* dst_arena, const long* table_dev, int n_ranges, long n_blocks, void* stream) {
    if (!src_arena || !dst_arena || !table_dev || n_ranges <= 0 || n_blocks <= 0 || n_blocks > 0x7fffffffL) return UNETR_ERR_ARG;
    if (((uintptr_t)src_arena & 31) || ((uintptr_t)dst_arena & 15)) return UNETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cast_bf16_ranges_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, src_arena, (uint16_t*)dst_arena,
                       table_dev, n_ranges);
    return unetr_check_launch();
}

extern "C" int unetr_add_cast_bf16(const float* a, const float* b, float* out, void* out_bf16, long n, void* stream) {
    if (!a || !b || !out || n <= 0) return UNETR_ERR_ARG;
    if ((n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) || ((uintptr_t)out_bf16 & 7)) return UNETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(add_cast_bf16_kernel, dim3((unsigned)std::min<long>(cdiv(n / 4, 256), 2048)), dim3(256), 0, (hipStream_t)stream, a, b, out,
                       (uint16_t*)out_bf16, n / 4);
    return unetr_check_launch();
}

// dw_i[N_i, K_i] = dy_i[M_i, N_i]^T * x_i[M_i, K_i] on bf16-stored dy / x (dense row-major), one launch per <= 64 problems;
// a: optimizer arenas -> the epilogue applies AdamW (step count a->steps[step_index[i]]) instead of storing dw;
// b16: a with only grad / shadow_bf16 / total set -> the epilogue stores bf16(dw) at the arena offset of dw in shadow_bf16
static int grouped_wgrad_bf16(const unetr_grouped_problem* probs, int n, const unetr_adamw_arena* a, const int* step_index, void* stream,
                              bool b16 = false) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    if (b16 && (!a || !a->grad || !a->shadow_bf16 || a->total <= 0 || ((uintptr_t)a->grad & 15) || ((uintptr_t)a->shadow_bf16 & 7))) return UNETR_ERR_ARG;
    if (a && !b16 && (!a->param || !a->grad || !a->m || !a->v || !a->steps || !step_index || a->total <= 0)) return UNETR_ERR_ARG;
    if (a && !b16 && ((((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->m | (uintptr_t)a->v) & 15) || ((uintptr_t)a->shadow_bf16 & 7))) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += GW_MAX) {
        GwArgs ga;
        ga.n = std::min(GW_MAX, n - base);
        ga.f = GwFuse{};
        if (a) ga.f = GwFuse{a->param, a->grad, a->m, a->v, (uint16_t*)a->shadow_bf16, a->steps, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay,
                             b16 ? nullptr : (uint32_t*)a->shadow_x3};
        int tiles = 0;
        for (int i = 0; i < ga.n; ++i) {
            const unetr_grouped_problem& q = probs[base + i];
            if (!q.dy || !q.x || !q.dw || q.M <= 0 || q.N < 8 || q.K < 8) return UNETR_ERR_ARG;
            if (q.M % 8 || q.N % 8 || q.K % 8 || ((uintptr_t)q.dy & 15) || ((uintptr_t)q.x & 15) || ((uintptr_t)q.dw & 15)) return UNETR_ERR_UNSUPPORTED;
            if (a) {       // dw names a slice of the gradient arena
                const long off = q.dw - a->grad;
                if (q.dw < a->grad || off + (long)q.N * q.K > a->total || (!b16 && step_index[base + i] < 0)) return UNETR_ERR_ARG;
            }
            GwProblem& g = ga.p[i];
            g.dy = (const uint16_t*)q.dy; g.x = (const uint16_t*)q.x; g.dw = q.dw; g.M = q.M; g.N = q.N; g.K = q.K;
            g.tile0 = tiles; g.ntn = cdiv(q.N, 128);
            g.step_idx = a && !b16 ? step_index[base + i] : 0;
            tiles += g.ntn * cdiv(q.K, 128);
        }
        // 32 tokens per stage, two stages: 32 KB of LDS per workgroup -> four workgroups per CU.  The launch is latency-bound per
        // K step (7-14 short steps per tile, operands out of HBM / MALL); measured at 432 rows (tools/probe_gw.py, us):
        // 64 tokens x 2 stages (two workgroups per CU) 201, 32 x 2 178, 32 x 3 190, 32 x 4 203, 64 x 3 268, 64 x 4 245
        if (b16) hipLaunchKernelGGL((gemm_bf16_grouped_wgrad_kernel<2, 32, 2>), dim3(tiles), dim3(256), 0, st, ga);
        else if (a) hipLaunchKernelGGL((gemm_bf16_grouped_wgrad_kernel<2, 32, 1>), dim3(tiles), dim3(256), 0, st, ga);
        else hipLaunchKernelGGL((gemm_bf16_grouped_wgrad_kernel<2, 32, 0>), dim3(tiles), dim3(256), 0, st, ga);
    }
    return unetr_check_launch();
}

extern "C" int unetr_gemm_bf16_grouped_wgrad(const unetr_grouped_problem* probs, int n, void* stream) {
    return grouped_wgrad_bf16(probs, n, nullptr, nullptr, stream);
}

extern "C" int unetr_gemm_bf16_grouped_wgrad_bf16out(const unetr_grouped_problem* probs, int n, const float* grad_arena, void* out_bf16_arena,
                                                     long total, void* stream) {
    unetr_adamw_arena a{};
    a.grad = grad_arena; a.shadow_bf16 = out_bf16_arena; a.total = total; a.shadow_x3 = nullptr;
    return grouped_wgrad_bf16(probs, n, &a, nullptr, stream, true);
}

extern "C" int unetr_gemm_bf16_grouped_wgrad_adamw(const unetr_grouped_problem* probs, int n, const unetr_adamw_arena* a,
                                                   const int* step_index, void* stream) {
    if (!a) return UNETR_ERR_ARG;
    return grouped_wgrad_bf16(probs, n, a, step_index, stream);
}
