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
    __device__ __forceinline__ void store(int, int m, int n, float v) const {
        v *= alpha;
        if (bias) v += bias[n];
        if (pre) pre[(long)m * ldc + n] = v;
        if (act == 1) v = gelu_exact(v);
        else if (act == 2) v *= gelu_grad(aux[(long)m * ldaux + n]);
        if (res) v += res[(long)(m % res_mod) * ldr + n];
        if (C) {
            if (accumulate) v += C[(long)m * ldc + n];
            C[(long)m * ldc + n] = v;
        }
        if (Cb) {
            __bf16 h = (__bf16)v;
            Cb[(long)m * ldcb + n] = __builtin_bit_cast(uint16_t, h);
        }
    }
    // four consecutive columns n..n+3 of row m (n % 4 == 0, all pitches multiples of 4 checked by the host)
    __device__ __forceinline__ void store4(int m, int n, f32x4 v) const {
        v *= alpha;
        if (bias) v += *(const f32x4*)(bias + n);
        if (pre) *(f32x4*)(pre + (long)m * ldc + n) = v;
        if (act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_exact(v[e]);
        } else if (act == 2) {
            const f32x4 a = *(const f32x4*)(aux + (long)m * ldaux + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= gelu_grad(a[e]);
        }
        if (res) v += *(const f32x4*)(res + (long)(m % res_mod) * ldr + n);
        if (C) {
            if (accumulate) v += *(const f32x4*)(C + (long)m * ldc + n);
            *(f32x4*)(C + (long)m * ldc + n) = v;
        }
        if (Cb) *(bf16x4*)(Cb + (long)m * ldcb + n) = __builtin_convertvector(v, bf16x4);
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

template <int WM, int WN, int WVM, int WVN, bool BKN, int NS>
__global__ void __launch_bounds__(64 * WVM * WVN)
gemm_bf16_kernel(int M, int N, int K, int mt, int nt, int splits, int kper,
                 const uint16_t* __restrict__ A, long lda, const uint16_t* __restrict__ B, long ldb, EpBf ep,
                 float* __restrict__ ws) {
    constexpr int NT = 64 * WVM * WVN, BM = 16 * WM * WVM, BN = 16 * WN * WVN, BK = 64;
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BKN ? BK * BN * 2 : BN * 128;
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
    const uint16_t* asrc[AIT];
    const uint16_t* bsrc[BIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int id = tid + i * NT, r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
        asrc[i] = A + (long)min(m0 + r, M - 1) * lda + kbeg + c * 8;
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
        const int id = tid + i * NT;
        if constexpr (!BKN) {
            const int r = id >> 3, c = (id & 7) ^ ((r >> 1) & 7);
            bsrc[i] = B + (long)min(n0 + r, N - 1) * ldb + kbeg + c * 8;
        } else {
            // image: BK rows (reduction index) of BN bf16, CPR 16-byte chunks per row, swizzled by bkn_x
            constexpr int CPR = BN / 8;
            const int r = id / CPR, s = id % CPR;
            const int c = s ^ bkn_x<CPR>(r);
            bsrc[i] = B + (long)(kbeg + r) * ldb + min(n0 + c * 8, N - 8);
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
            const uint16_t* g = BKN ? bsrc[i] + (long)kt * BK * ldb : bsrc[i] + (long)kt * BK;
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
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) PrecBF16::mma(acc[i][j], b[kb][j], a[kb][i]);    // transposed tile: a lane holds 4 consecutive n of one m
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // The MFMAs above ran with swapped operands, so the accumulator tile is C^T: lane (c, g) holds C[m = tile row c]
    // [n = 4g .. 4g+3] -- one 16-byte store (or 8-byte bf16 store) per tile instead of four 4-byte ones.
    const bool vec4 = ep.vec_ok;
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

template <int WM, int WN, int WVM, int WVN, bool BKN, int NS>
int launch_bf16(int M, int N, int K, const uint16_t* A, long lda, const uint16_t* B, long ldb, const EpBf& ep,
                float* ws, size_t ws_bytes, hipStream_t st, int* partial_splits = nullptr) {
    constexpr int BM = 16 * WM * WVM, BN = 16 * WN * WVN;
    const int mt = cdiv(M, BM), nt = cdiv(N, BN), ksteps = K / 64;
    const long tiles = (long)mt * nt;
    int splits = 1;
    // few tiles (batch-2 token counts): the kernel is latency-bound, one workgroup's time is ~ its K steps, so long K
    // ranges are cut into slabs of >= 12 steps (shorter slabs cost more in the reduce launch than they save)
    if (tiles < 192 && ksteps >= 24) splits = std::min(ksteps / 12, (int)((512 + tiles - 1) / tiles));
    if (const char* e = getenv("UNETR_GEMM_SPLITS")) { int v = atoi(e); if (v > 0) splits = std::min(v, ksteps); }
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    if (splits < 1 || ws == nullptr) splits = 1;
    const int kper = cdiv(ksteps, splits) * 64;
    splits = cdiv(K, kper);
    const int per = cdiv(tiles, 8);
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, WVM, WVN, BKN, NS>), dim3(per * 8, splits), dim3(64 * WVM * WVN), 0, st,
                       M, N, K, mt, nt, splits, kper, A, lda, B, ldb, ep, ws);
    // partial_splits: the caller consumes the split partials itself (unetr_gemm_bf16_ln_bwd): no reduce launch
    if (partial_splits) *partial_splits = splits;
    if (splits > 1 && !partial_splits)
        hipLaunchKernelGGL((splitk_reduce_kernel<EpBf, false>), dim3(cdiv(N, 64), cdiv(M, 4), 1), dim3(256), 0, st, M, N, splits, ws, ep);
    return unetr_check_launch();
}

// ---- grouped weight-gradient GEMM on bf16-stored operands: dW_i[N_i, K_i] = dY_i[M, N_i]^T * X_i[M, K_i] ------------------
// Both operands are reduction-major ([token][feature]), i.e. the b_kn layout on BOTH sides: a stage is 64 tokens x 128
// features of dY and of X, staged by LDS-DMA into two swizzled images (bkn_x<16>), and all MFMA fragments (k = token) come
// out through ds_read_b64_tr_b16.  One launch covers up to 48 problems (descriptors in the kernel arguments); tokens beyond
// M in the last stage read a clamped row and are zeroed in the A fragments (M is a multiple of 8: whole lane groups).
constexpr int GW_MAX = 64;     // 12 blocks x 4 Linear layers + patch embedding in ONE launch
struct GwProblem { const uint16_t* dy; const uint16_t* x; float* dw; int M, N, K, tile0, ntn; };
struct GwArgs { int n; GwProblem p[GW_MAX]; };

template <int NS>
__global__ void __launch_bounds__(256)
gemm_bf16_grouped_wgrad_kernel(GwArgs ga) {
    constexpr int BT = 128, BKT = 64;                      // output tile 128 x 128, 64 tokens per stage
    constexpr int IMG = BKT * BT * 2, STAGE = 2 * IMG, PCS = IMG / 16 / 256, G = 2 * PCS, CPR = 16;
    __shared__ __attribute__((aligned(1024))) char lds[NS * STAGE];
    int pi = 0;
    const int t = blockIdx.x;
    for (int i = 1; i < ga.n; ++i) pi = (t >= ga.p[i].tile0) ? i : pi;
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
        for (int kb = 0; kb < 2; ++kb) {
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
    // lane (cc, g) holds dW[n = tile row cc][k = 4g .. 4g+3]: ONE 16-byte store per tile (was four 4-byte stores whose 16 lanes
    // covered 64 bytes each); dW is written once and next read by AdamW after 350 MB of other gradients: non-temporal
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + (wm * 4 + i) * 16 + cc, k = k0 + (wn * 4 + j) * 16 + 4 * g;
            if (n < pr.N && k < pr.K) __builtin_nontemporal_store(acc[i][j], (f32x4*)(pr.dw + (long)n * pr.K + k));
        }
}

__global__ void __launch_bounds__(256) cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n8) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const f32x4 a = ((const f32x4*)src)[2 * i], b = ((const f32x4*)src)[2 * i + 1];
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        ((u32x4*)dst)[i] = PrecBF16::pack(v);
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
    EpBf ep{vec_ok, C, d->ldc, (uint16_t*)Cb, d->ldcb, d->bias, d->res, d->ldr, d->res_mod > 0 ? d->res_mod : M,
            d->pre, d->aux, d->ldaux, d->act, d->accumulate, d->alpha};
    const uint16_t* a = (const uint16_t*)A;
    const uint16_t* b = (const uint16_t*)B;
    const int env_cfg = getenv("UNETR_GEMM_CFG") ? atoi(getenv("UNETR_GEMM_CFG")) : 0;   // tuning hooks
    const int env_ns = getenv("UNETR_GEMM_STAGES") ? atoi(getenv("UNETR_GEMM_STAGES")) : 0;
    const bool big = env_cfg == 128 || (env_cfg == 0 && M >= 1024 && N >= 128);
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

// dw_i[N_i, K_i] = dy_i[M_i, N_i]^T * x_i[M_i, K_i] on bf16-stored dy / x (dense row-major), one launch per <= 48 problems
extern "C" int unetr_gemm_bf16_grouped_wgrad(const unetr_grouped_problem* probs, int n, void* stream) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += GW_MAX) {
        GwArgs ga;
        ga.n = std::min(GW_MAX, n - base);
        int tiles = 0;
        for (int i = 0; i < ga.n; ++i) {
            const unetr_grouped_problem& q = probs[base + i];
            if (!q.dy || !q.x || !q.dw || q.M <= 0 || q.N < 8 || q.K < 8) return UNETR_ERR_ARG;
            if (q.M % 8 || q.N % 8 || q.K % 8 || ((uintptr_t)q.dy & 15) || ((uintptr_t)q.x & 15) || ((uintptr_t)q.dw & 15)) return UNETR_ERR_UNSUPPORTED;
            GwProblem& g = ga.p[i];
            g.dy = (const uint16_t*)q.dy; g.x = (const uint16_t*)q.x; g.dw = q.dw; g.M = q.M; g.N = q.N; g.K = q.K;
            g.tile0 = tiles; g.ntn = cdiv(q.N, 128);
            tiles += g.ntn * cdiv(q.K, 128);
        }
        hipLaunchKernelGGL((gemm_bf16_grouped_wgrad_kernel<2>), dim3(tiles), dim3(256), 0, st, ga);
    }
    return unetr_check_launch();
}
