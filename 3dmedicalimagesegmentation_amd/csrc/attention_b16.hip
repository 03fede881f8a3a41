// Self-attention core on bf16-STORED q/k/v for head dim 64 (MONAI SABlock between qkv and out_proj, built at
// /root/reference/unetr.py:78-89): the bf16-mode fast path.  Same mathematics and the same "query on the lane" / "key on the
// lane" orientations as attention.hip; what changes is the data path, which is what the batch-2 step pays for:
//   * q/k/v (and dO in backward) are read as bf16 straight from the GEMM that produced them: tiles go HBM/L2 -> LDS by
//     LDS-DMA (global_load_lds_dwordx4), no fp32 read, no conversion pass, no ds_write;
//   * one LDS image per operand serves BOTH kinds of fragment read -- rows (ds_read_b128: S = K q, dP = V dO) and
//     columns (ds_read_b64_tr_b16: O^T += V^T P^T, dQ^T += K^T dS^T, ...): 128-byte rows, 16-byte chunk c of row r stored
//     at chunk slot c ^ (((r >> 1) & 3) << 1); both reads are bank-conflict free on it (the swizzle is applied to the
//     per-lane SOURCE address of the DMA, the LDS image itself is lane-linear);
//   * a whole chunk of up to 224 keys is resident, so the scores of one query against the chunk sit in registers (14 tiles)
//     and softmax needs no per-tile rescaling: 28 independent MFMAs, one max / exp / sum pass, 28 MFMAs.  L > 224 (160^3:
//     1000 tokens) loops over chunks with the usual running max / sum;
//   * workgroup = 32 queries (2 waves): 2 x 12 heads x 7 query tiles = 168 workgroups at batch 2, one round on 256 CUs.
//
// Layout: qkv bf16 [B*L, 3*Hd], feature = which*Hd + head*64 + j; out [B*L, Hd]; lse / delta [B, heads, L].
#include <cstdlib>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LDS_AS_ __attribute__((address_space(3)))

constexpr int DH = 64, ROWB = 128;          // head dim, bytes per image row
constexpr int CKEYS = 224, CT = CKEYS / 16; // keys (or queries) resident per chunk, 16-row tiles per chunk
constexpr int IMG = CKEYS * ROWB;           // 28 KB per image
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ int img_off(int row, int chunk) { return row * ROWB + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

// stage `rows` (<= CKEYS, rounded up to 16 by the caller's clamp) rows of 64 bf16 (pitch rs elements) into an image by
// LDS-DMA; rows at or beyond nrows re-read row nrows-1 (finite data, masked by the consumer)
template <int NT>
__device__ __forceinline__ void stage_img(const uint16_t* __restrict__ src, long rs, int row0, int nrows, int rows, char* img) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pieces = rows * 8;
    for (int base = 0; base < pieces; base += NT) {          // (uniform trip count; whole waves: rows % 16 == 0, NT % 64 == 0)
        const int id = base + tid;
        if (base + wave * 64 < pieces) {
            const int r = id >> 3, sl = id & 7, c = sl ^ (((r >> 1) & 3) << 1);
            const uint16_t* g = src + (long)min(row0 + r, nrows - 1) * rs + c * 8;
            __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(img + (base + wave * 64) * 16), 16, 0, 0);
        }
    }
}

// acc (16 image rows of tile rt x 16 lane columns) = sum_d Img[row][d] * vec[d][col]; vec = this lane's 64-vector as 2 frags
__device__ __forceinline__ f32x4 prod_rows(const char* img, int rt, const u32x4 (&vec)[2]) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const u32x4 a = *(const u32x4*)(img + img_off(rt * 16 + c, kb * 4 + g));
        PrecBF16::mma(acc, a, vec[kb]);
    }
    return acc;
}

// out[dt] (16 d x 16 lane columns) += sum over the 32 image rows R0 .. R0+31 of Img^T[d][row] * X[row][col]; X = two
// accumulator tiles x0 (rows 0..15), x1 (rows 16..31) in the MFMA C layout (row = 4*(lane>>4) + reg)
__device__ __forceinline__ void prod_T(f32x4 (&out)[4], f32x4 x0, f32x4 x1, const char* img, int R0) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    const u32x4 b = PrecBF16::pack(xv);
    const int r0 = R0 + 4 * g + q, r1 = r0 + 16;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + (p >> 1);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(img + img_off(r0, ch) + (p & 1) * 8));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)(img + img_off(r1, ch) + (p & 1) * 8));
        s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        PrecBF16::mma(out[dt], __builtin_bit_cast(u32x4, t), b);
    }
}

// The same two products over a FULL chunk (all CT tiles) as one software-pipelined straight line: the image fragments of the
// next DPT tiles are in flight while a tile's MFMAs run, order pinned with sched_barrier.  (As a guarded per-tile loop every
// tile waited for its own two reads: 14 exposed LDS round trips per product -- for 216 tokens the chunk is always full.)
template <class F>
__device__ __forceinline__ void prod_rows_full(const char* img, const u32x4 (&vec)[2], F&& each) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    constexpr int DPT = 5;
    const char* p0 = img + img_off(c, g);                 // (row tiles are 16 rows = 2 KB apart: the swizzle repeats every 8 rows)
    const char* p1 = img + img_off(c, 4 + g);
    u32x4 r0[DPT], r1[DPT];
#pragma unroll
    for (int t = 0; t < DPT; ++t) { r0[t] = *(const u32x4*)(p0 + t * 16 * ROWB); r1[t] = *(const u32x4*)(p1 + t * 16 * ROWB); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        PrecBF16::mma(acc, r0[t % DPT], vec[0]);
        PrecBF16::mma(acc, r1[t % DPT], vec[1]);
        if (t + DPT < CT) { r0[t % DPT] = *(const u32x4*)(p0 + (t + DPT) * 16 * ROWB); r1[t % DPT] = *(const u32x4*)(p1 + (t + DPT) * 16 * ROWB); }
        each(t, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// out[dt] += Img^T . X over all CT/2 row pairs; x(t2, x0, x1) supplies the pair's two accumulator tiles
template <class F>
__device__ __forceinline__ void prod_T_full(f32x4 (&out)[4], const char* img, F&& x) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4, q = c >> 2, p = c & 3;
    constexpr int NS = (CT / 2) * 4, DPT = 6;             // (row pair, d tile) steps
    const char* b0[2];                                    // per-lane address of d tiles 0 / 1 chunk parity (ch = dt*2 + (p>>1))
    const int r0 = 4 * g + q;
    auto addr = [&](int st, int hi) { const int t2 = st >> 2, dt = st & 3; return img + img_off(32 * t2 + r0 + 16 * hi, dt * 2 + (p >> 1)) + (p & 1) * 8; };
    (void)b0;
    s16x4 lo[DPT], hi[DPT];
#pragma unroll
    for (int st = 0; st < DPT; ++st) {
        lo[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)addr(st, 0));
        hi[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)addr(st, 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 b = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        const int t2 = st >> 2, dt = st & 3;
        if (dt == 0) {
            f32x4 x0, x1;
            x(t2, x0, x1);
            float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
            b = PrecBF16::pack(xv);
        }
        s16x8 t = {lo[st % DPT][0], lo[st % DPT][1], lo[st % DPT][2], lo[st % DPT][3], hi[st % DPT][0], hi[st % DPT][1], hi[st % DPT][2], hi[st % DPT][3]};
        PrecBF16::mma(out[dt], __builtin_bit_cast(u32x4, t), b);
        if (st + DPT < NS) {
            lo[st % DPT] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)addr(st + DPT, 0));
            hi[st % DPT] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS_ s16x4*)addr(st + DPT, 1));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// this lane's token row (64 bf16) as two MFMA fragments: chunk kb*4 + (lane>>4)
__device__ __forceinline__ void load_vec(const uint16_t* __restrict__ row, u32x4 (&f)[2]) {
    const int g = (threadIdx.x & 63) >> 4;
    f[0] = *(const u32x4*)(row + g * 8);
    f[1] = *(const u32x4*)(row + (4 + g) * 8);
}

__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// ------------------------------------------------------------------------------------------ forward
template <int NW>
__global__ void __launch_bounds__(64 * NW)
attn16_fwd_kernel(const uint16_t* __restrict__ qkv, float* __restrict__ out, uint16_t* __restrict__ outb, float* __restrict__ lse,
                  int L, int heads, float scale) {
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* kimg = lds;
    char* vimg = lds + IMG;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const uint16_t* qb = qkv + (long)b * L * rs + head * DH;
    const uint16_t* kb_ = qb + Hd;
    const uint16_t* vb = qb + 2 * Hd;
    const int q = blockIdx.x * (16 * NW) + wave * 16 + c, qc = min(q, L - 1);
    u32x4 qf[2];
    load_vec(qb + (long)qc * rs, qf);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, l = 0.f;
    for (int k0 = 0; k0 < L; k0 += CKEYS) {
        const int nkeys = min(CKEYS, L - k0), rows = (nkeys + 31) & ~31, nt = rows >> 4;    // whole 32-key pairs
        if (k0) __syncthreads();                                   // readers of the previous chunk are done
        stage_img<NT>(kb_, rs, k0, L, rows, kimg);
        stage_img<NT>(vb, rs, k0, L, rows, vimg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 s[CT];
        float tmax = NEG_BIG;
        const bool full = nt == CT;                                // wave-uniform
        if (full) {
            prod_rows_full(kimg, qf, [&](int t, f32x4 acc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    acc[r] = key < nkeys ? acc[r] * scale : NEG_BIG;
                    tmax = fmaxf(tmax, acc[r]);
                }
                s[t] = acc;
            });
        } else
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if (t < nt) {
                s[t] = prod_rows(kimg, t, qf);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    s[t][r] = key < nkeys ? s[t][r] * scale : NEG_BIG;
                    tmax = fmaxf(tmax, s[t][r]);
                }
            } else {
                s[t] = (f32x4){NEG_BIG, NEG_BIG, NEG_BIG, NEG_BIG};
            }
        }
        tmax = grp_max(tmax);
        const float mn = fmaxf(m, tmax), alpha = __expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = __expf(s[t][r] - mn); ps += s[t][r]; }
        ps = grp_sum(ps);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
        if (full) {
            prod_T_full(o, vimg, [&](int t2, f32x4& x0, f32x4& x1) { x0 = s[2 * t2]; x1 = s[2 * t2 + 1]; });
        } else {
#pragma unroll
            for (int t2 = 0; t2 < CT / 2; ++t2)
                if (2 * t2 < nt) prod_T(o, s[2 * t2], s[2 * t2 + 1], vimg, 32 * t2);
        }
    }
    if (q < L) {
        const float inv = 1.f / l;
        if (out) {
            float* op = out + ((long)b * L + q) * Hd + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(op + dt * 16) = o[dt] * inv;
        }
        if (outb) {
            uint16_t* ob = outb + ((long)b * L + q) * Hd + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(ob + dt * 16) = __builtin_convertvector(o[dt] * inv, bf16x4);
        }
        if (g == 0) lse[((long)b * heads + head) * L + q] = m + __logf(l);
    }
}

// ------------------------------------------------------------------------------- backward: dQ pass
// query on the lane; also writes delta[b,h,q] = sum_d dO[q,d] O[q,d] (kept as an output of the entry point; the dK/dV role below
// recomputes the values it needs, so the two roles have no dependence and share one launch)
template <int NW>
__device__ __forceinline__ void
attn16_bwd_dq_body(char* lds, int xblk, const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ outb, const uint16_t* __restrict__ doutb,
                   const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, uint16_t* __restrict__ dqkvb,
                   int L, int heads, float scale) {
    constexpr int NT = 64 * NW;
    char* kimg = lds;
    char* vimg = lds + IMG;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const uint16_t* qb = qkv + (long)b * L * rs + head * DH;
    const uint16_t* kb_ = qb + Hd;
    const uint16_t* vb = qb + 2 * Hd;
    const int q = xblk * (16 * NW) + wave * 16 + c, qc = min(q, L - 1);
    u32x4 qf[2], dof[2], of[2];
    load_vec(qb + (long)qc * rs, qf);
    load_vec(doutb + ((long)b * L + qc) * Hd + head * DH, dof);
    load_vec(outb + ((long)b * L + qc) * Hd + head * DH, of);
    const float lq = lse[((long)b * heads + head) * L + qc];
    float dq_ = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, of[kb]), d = __builtin_bit_cast(bf16x8, dof[kb]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dq_ += (float)a[e] * (float)d[e];
    }
    dq_ = grp_sum(dq_);
    if (g == 0 && q < L) delta[((long)b * heads + head) * L + q] = dq_;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += CKEYS) {
        const int nkeys = min(CKEYS, L - k0), rows = (nkeys + 31) & ~31, nt = rows >> 4;
        if (k0) __syncthreads();
        stage_img<NT>(kb_, rs, k0, L, rows, kimg);
        stage_img<NT>(vb, rs, k0, L, rows, vimg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nt == CT) {                                                // full chunk (wave-uniform): pipelined straight lines
            f32x4 dsf[CT];
            prod_rows_full(kimg, qf, [&](int t, f32x4 acc) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    acc[r] = key < nkeys ? __expf(acc[r] * scale - lq) : 0.f;
                }
                dsf[t] = acc;                                          // p
            });
            prod_rows_full(vimg, dof, [&](int t, f32x4 dp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dsf[t][r] *= dp[r] - dq_;
            });
            prod_T_full(dq, kimg, [&](int t2, f32x4& x0, f32x4& x1) { x0 = dsf[2 * t2]; x1 = dsf[2 * t2 + 1]; });
        } else
#pragma unroll
        for (int t2 = 0; t2 < CT / 2; ++t2) {
            if (2 * t2 >= nt) break;
            f32x4 ds[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * t2 + h;
                const f32x4 s = prod_rows(kimg, t, qf);
                const f32x4 dp = prod_rows(vimg, t, dof);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = t * 16 + 4 * g + r;
                    const float p = key < nkeys ? __expf(s[r] * scale - lq) : 0.f;
                    ds[h][r] = p * (dp[r] - dq_);
                }
            }
            prod_T(dq, ds[0], ds[1], kimg, 32 * t2);
        }
    }
    if (q < L) {
        if (dqkv) {
            float* op = dqkv + ((long)b * L + q) * rs + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) *(f32x4*)(op + dt * 16) = dq[dt] * scale;
        }
        uint16_t* ob = dqkvb + ((long)b * L + q) * rs + head * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(ob + dt * 16) = __builtin_convertvector(dq[dt] * scale, bf16x4);
    }
}

// ---------------------------------------------------------------------------- backward: dK/dV pass
// key on the lane: no atomics; loops over chunks of queries (Q and dO images resident, lse / delta of the chunk in LDS; delta is
// computed here from the O and dO rows -- 2 x 128 B per query -- with the same summation order as the dQ role uses)
template <int NW>
__device__ __forceinline__ void
attn16_bwd_dkv_body(char* lds, int xblk, const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ outb, const uint16_t* __restrict__ doutb,
                    const float* __restrict__ lse, float* __restrict__ dqkv, uint16_t* __restrict__ dqkvb, int L, int heads, float scale) {
    constexpr int NT = 64 * NW;
    char* qimg = lds;
    char* doimg = lds + IMG;
    float* lse_t = (float*)(lds + 2 * IMG);
    float* del_t = lse_t + CKEYS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const uint16_t* qb = qkv + (long)b * L * rs + head * DH;
    const uint16_t* kb_ = qb + Hd;
    const uint16_t* vb = qb + 2 * Hd;
    const uint16_t* dob = doutb + (long)b * L * Hd + head * DH;
    const uint16_t* ob_ = outb + (long)b * L * Hd + head * DH;
    const int key = xblk * (16 * NW) + wave * 16 + c, kc = min(key, L - 1);
    u32x4 kf[2], vf[2];
    load_vec(kb_ + (long)kc * rs, kf);
    load_vec(vb + (long)kc * rs, vf);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
    for (int q0 = 0; q0 < L; q0 += CKEYS) {
        const int nq = min(CKEYS, L - q0), rows = (nq + 31) & ~31, nt = rows >> 4;
        if (q0) __syncthreads();
        stage_img<NT>(qb, rs, q0, L, rows, qimg);
        stage_img<NT>(dob, (long)Hd, q0, L, rows, doimg);
        for (int i = threadIdx.x; i < rows; i += NT) {
            const int qq = q0 + i;
            lse_t[i] = qq < L ? lse[((long)b * heads + head) * L + qq] : 1.0e30f;     // p = exp(s - 1e30) = 0 for padding
        }
        // delta of the chunk's queries, one query row per thread (2 x 128 B of O and dO).  The dQ role sums pieces g and 4+g in
        // one chain and then adds the four chains as (0+1)+(2+3); the same order is kept here
        for (int i = threadIdx.x; i < rows; i += NT) {
            const int qq = min(q0 + i, L - 1);
            float part[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                part[gg] = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, *(const u32x4*)(ob_ + (long)qq * Hd + (kb * 4 + gg) * 8));
                    const bf16x8 d = __builtin_bit_cast(bf16x8, *(const u32x4*)(dob + (long)qq * Hd + (kb * 4 + gg) * 8));
#pragma unroll
                    for (int e = 0; e < 8; ++e) part[gg] += (float)a[e] * (float)d[e];
                }
            }
            del_t[i] = q0 + i < L ? (part[0] + part[1]) + (part[2] + part[3]) : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nt == CT) {                                                // full chunk (wave-uniform): pipelined straight lines
            f32x4 pf[CT], dsf[CT];
            prod_rows_full(qimg, kf, [&](int t, f32x4 acc) {
                const f32x4 l4 = *(const f32x4*)(lse_t + t * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = __expf(acc[r] * scale - l4[r]);
                pf[t] = acc;
            });
            prod_rows_full(doimg, vf, [&](int t, f32x4 dp) {
                const f32x4 d4 = *(const f32x4*)(del_t + t * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) dsf[t][r] = pf[t][r] * (dp[r] - d4[r]);
            });
            prod_T_full(dv, doimg, [&](int t2, f32x4& x0, f32x4& x1) { x0 = pf[2 * t2]; x1 = pf[2 * t2 + 1]; });
            prod_T_full(dk, qimg, [&](int t2, f32x4& x0, f32x4& x1) { x0 = dsf[2 * t2]; x1 = dsf[2 * t2 + 1]; });
        } else
#pragma unroll
        for (int t2 = 0; t2 < CT / 2; ++t2) {
            if (2 * t2 >= nt) break;
            f32x4 p[2], ds[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = 2 * t2 + h;
                const f32x4 s = prod_rows(qimg, t, kf);
                const f32x4 dp = prod_rows(doimg, t, vf);
                const f32x4 l4 = *(const f32x4*)(lse_t + t * 16 + 4 * g), d4 = *(const f32x4*)(del_t + t * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(s[r] * scale - l4[r]);
                    p[h][r] = pv;
                    ds[h][r] = pv * (dp[r] - d4[r]);
                }
            }
            prod_T(dv, p[0], p[1], doimg, 32 * t2);
            prod_T(dk, ds[0], ds[1], qimg, 32 * t2);
        }
    }
    if (key < L) {
        if (dqkv) {
            float* kp = dqkv + ((long)b * L + key) * rs + Hd + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *(f32x4*)(kp + dt * 16) = dk[dt] * scale;
                *(f32x4*)(kp + Hd + dt * 16) = dv[dt];
            }
        }
        uint16_t* kb2 = dqkvb + ((long)b * L + key) * rs + Hd + head * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *(bf16x4*)(kb2 + dt * 16) = __builtin_convertvector(dk[dt] * scale, bf16x4);
            *(bf16x4*)(kb2 + Hd + dt * 16) = __builtin_convertvector(dv[dt], bf16x4);
        }
    }
}

// one launch, two roles: workgroups [0, nqb) of x are dQ blocks, [nqb, 2 nqb) are dK/dV blocks.  Neither role fills the chip at
// batch 2 (168 workgroups of 128 threads each), so back to back they cost two launch boundaries and two half-empty rounds
template <int NW>
__global__ void __launch_bounds__(64 * NW)
attn16_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ outb, const uint16_t* __restrict__ doutb,
                  const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, uint16_t* __restrict__ dqkvb,
                  int L, int heads, float scale, int nqb) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int xb = __builtin_amdgcn_readfirstlane(blockIdx.x);
    if (xb < nqb) attn16_bwd_dq_body<NW>(lds, xb, qkv, outb, doutb, lse, delta, dqkv, dqkvb, L, heads, scale);
    else attn16_bwd_dkv_body<NW>(lds, xb - nqb, qkv, outb, doutb, lse, dqkv, dqkvb, L, heads, scale);
}

template <class K, class... A>
void launch_dyn(K kern, dim3 grid, int nt, size_t lds, hipStream_t st, A... args) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(nt), lds, st, args...);
}

// queries per workgroup: 32 (2 waves) while that still gives at most ~2 rounds of workgroups, else 64; the FORWARD kernel takes 128
// (8 waves) once even the 64-query grid is several rounds deep (batch >= 16 at 216 tokens): every workgroup stages all of K and V
// for its (batch item, head), so twice the queries per workgroup halve that traffic (UNETR_ATTN_NW forces 2 / 4 / 8)
inline int pick_nw(int B, int L, int heads) { return ((long)cdiv(L, 32) * heads * B <= 512) ? 2 : 4; }
inline int pick_nw_fwd(int B, int L, int heads) {
    if (const char* e = getenv("UNETR_ATTN_NW")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) return v; }
    const int nw = pick_nw(B, L, heads);
    return (nw == 4 && (long)cdiv(L, 64) * heads * B > 1024) ? 8 : nw;
}

}  // namespace

extern "C" int unetr_attention_bf16_fwd(const void* qkv, float* out, void* out_bf16, float* lse, int B, int L, int heads, int dh,
                                        float scale, void* stream) {
    if (!qkv || (!out && !out_bf16) || !lse || B <= 0 || L <= 0 || heads <= 0 || B > 65535 || heads > 65535) return UNETR_ERR_ARG;
    if (dh != DH || ((uintptr_t)qkv & 15) || (out && ((uintptr_t)out & 15)) || (out_bf16 && ((uintptr_t)out_bf16 & 7))) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = 2 * IMG;
    const int nw = pick_nw_fwd(B, L, heads);
    if (nw == 2)
        launch_dyn(attn16_fwd_kernel<2>, dim3(cdiv(L, 32), heads, B), 128, lds, st, (const uint16_t*)qkv, out, (uint16_t*)out_bf16, lse, L, heads, scale);
    else if (nw == 4)
        launch_dyn(attn16_fwd_kernel<4>, dim3(cdiv(L, 64), heads, B), 256, lds, st, (const uint16_t*)qkv, out, (uint16_t*)out_bf16, lse, L, heads, scale);
    else
        launch_dyn(attn16_fwd_kernel<8>, dim3(cdiv(L, 128), heads, B), 512, lds, st, (const uint16_t*)qkv, out, (uint16_t*)out_bf16, lse, L, heads, scale);
    return unetr_check_launch();
}

extern "C" int unetr_attention_bf16_bwd(const void* qkv, const void* out_bf16, const void* dout_bf16, const float* lse, float* dqkv,
                                        void* dqkv_bf16, float* delta, int B, int L, int heads, int dh, float scale, void* stream) {
    if (!qkv || !out_bf16 || !dout_bf16 || !lse || !dqkv_bf16 || !delta || B <= 0 || L <= 0 || heads <= 0 || B > 65535 || heads > 65535)
        return UNETR_ERR_ARG;
    if (dh != DH || ((uintptr_t)qkv & 15) || ((uintptr_t)out_bf16 & 15) || ((uintptr_t)dout_bf16 & 15) || ((uintptr_t)dqkv_bf16 & 7) ||
        (dqkv && ((uintptr_t)dqkv & 15)))
        return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const uint16_t* q = (const uint16_t*)qkv;
    const uint16_t* o = (const uint16_t*)out_bf16;
    const uint16_t* d = (const uint16_t*)dout_bf16;
    uint16_t* gq = (uint16_t*)dqkv_bf16;
    const size_t lk = 2 * IMG + 2 * CKEYS * sizeof(float);
    if (pick_nw(B, L, heads) == 2) {
        const int nqb = cdiv(L, 32);
        launch_dyn(attn16_bwd_kernel<2>, dim3(2 * nqb, heads, B), 128, lk, st, q, o, d, lse, delta, dqkv, gq, L, heads, scale, nqb);
    } else {
        const int nqb = cdiv(L, 64);
        launch_dyn(attn16_bwd_kernel<4>, dim3(2 * nqb, heads, B), 256, lk, st, q, o, d, lse, delta, dqkv, gq, L, heads, scale, nqb);
    }
    return unetr_check_launch();
}
