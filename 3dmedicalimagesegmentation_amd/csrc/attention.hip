// Multi-head self-attention core for the UNETR ViT encoder (MONAI SABlock between qkv and out_proj),
// flash-style on gfx950: K/V tiles staged in LDS, QK^T and PV on 16x16 MFMA tiles, online softmax with
// wave64 shuffles, nothing of size LxL ever touches HBM.
//
// Orientation ("query on the lane"): the forward computes S^T = K.Q^T so that the accumulator of a
// 16(key) x 16(query) tile has the query on lane&15 and four keys per lane in registers; the softmax
// statistics (running max m, running sum l) are then per-lane scalars, completed across the four lane
// groups with two __shfl_xor.  The accumulator tile is reused as the B operand of the next product
// (O^T += V^T . P^T) without any lane movement, because that product sums over the accumulator's ROW
// index.  The backward uses the same trick twice: a "query on the lane" kernel for dQ and a "key on the
// lane" kernel for dK/dV (which therefore need no atomics); both recompute P from the saved LSE.
//
// Layout: qkv [B*L, 3*Hd], feature = which*Hd + head*DH + j (einops "b h (qkv l d) -> qkv b l h d");
//         out/dout [B*L, Hd] ("b h l d -> b l (h d)"); lse/delta [B, heads, L].
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

constexpr int PT = 80;  // byte pitch of a transposed bf16 image row (32 keys * 2 B + 16 B pad)

template <class P, int DH> struct AttnCfg {
    static constexpr int CH = P::CH;
    static constexpr bool BF = (CH == 8);
    static constexpr int KB = DH / (4 * CH);      // k-blocks over the head dim
    static constexpr int DT = DH / 16;            // 16-wide tiles over the head dim
    static constexpr int PC = DH * (16 / CH) + 16;  // byte pitch of a chunk-image row
    static constexpr int IMG_C = 32 * PC;         // bytes of a 32-row chunk image
    static constexpr int IMG_T = 0;               // no transposed image: bf16 fragments with k = tile row come out through ds_read_b64_tr_b16
};

template <class P, int DH> using AccArr = f32x4[AttnCfg<P, DH>::DT];
template <class P, int DH> using FragArr = u32x4[AttnCfg<P, DH>::KB];

// stage 32 rows x DH fp32 (row i at src + (row0+i)*rs, zero beyond nrows) into a chunk image and, in bf16
// mode, optionally a transposed image [DH][32].
template <class P, int DH, bool WANT_C, bool WANT_T>
__device__ __forceinline__ void stage_tile(const float* __restrict__ src, long rs, int row0, int nrows, char* img_c, char* img_t) {
    using C = AttnCfg<P, DH>;
    constexpr int CH = C::CH;
    const int tid = threadIdx.x;
    if (WANT_C || WANT_T) {
        constexpr int NCH = DH / CH;
        for (int id = tid; id < 32 * NCH; id += 256) {
            int r = id / NCH, ch = id - r * NCH;
            float v[CH];
            if (row0 + r < nrows) {
                const float* q = src + (long)(row0 + r) * rs + ch * CH;
#pragma unroll
                for (int c4 = 0; c4 < CH / 4; ++c4) {
                    f32x4 t = *(const f32x4*)(q + 4 * c4);
                    v[4 * c4] = t[0]; v[4 * c4 + 1] = t[1]; v[4 * c4 + 2] = t[2]; v[4 * c4 + 3] = t[3];
                }
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) v[j] = 0.f;
            }
            *(u32x4*)(img_c + r * C::PC + ch * 16) = P::pack(v);
        }
    }
    (void)img_t;
}

// this lane's DH-vector (one token's head slice) as KB MFMA chunks: chunk index kb*4 + (lane>>4)
template <class P, int DH>
__device__ __forceinline__ void load_vec_frags(const float* __restrict__ rowptr, FragArr<P, DH>& f) {
    constexpr int CH = P::CH;
    const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int kb = 0; kb < AttnCfg<P, DH>::KB; ++kb) {
        float v[CH];
        const float* q = rowptr + (kb * 4 + g) * CH;
#pragma unroll
        for (int c4 = 0; c4 < CH / 4; ++c4) {
            f32x4 t = *(const f32x4*)(q + 4 * c4);
            v[4 * c4] = t[0]; v[4 * c4 + 1] = t[1]; v[4 * c4 + 2] = t[2]; v[4 * c4 + 3] = t[3];
        }
        f[kb] = P::pack(v);
    }
}

// acc[16 image rows (tile rt) x 16 lane columns] = sum_d Img[row][d] * vec[d][col]
template <class P, int DH>
__device__ __forceinline__ f32x4 prod_rows(const char* img_c, int rt, const FragArr<P, DH>& vec) {
    using C = AttnCfg<P, DH>;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < C::KB; ++kb) {
        u32x4 a = *(const u32x4*)(img_c + (rt * 16 + c) * C::PC + (kb * 4 + g) * 16);
        P::mma(acc, a, vec[kb]);
    }
    return acc;
}

// out[dt] (16 d x 16 lane columns) += sum over the 32 tile rows of Img^T[d][row] * X[row][col], where X is
// held as two accumulator tiles x0 (rows 0..15) and x1 (rows 16..31): row = 16*t + 4*(lane>>4) + reg.
template <class P, int DH>
__device__ __forceinline__ void prod_T(AccArr<P, DH>& out, f32x4 x0, f32x4 x1, const char* img_c, const char* img_t) {
    using C = AttnCfg<P, DH>;
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    if constexpr (!C::BF) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            // (4-byte image elements -- fp32 or split words --: four element reads per operand chunk; PrecF32::mma is the four
            // K = 4 MFMAs of before)
            u32x4 a0, a1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a0[t] = *(const uint32_t*)(img_c + (4 * g + t) * C::PC + (dt * 16 + c) * 4);
                a1[t] = *(const uint32_t*)(img_c + (16 + 4 * g + t) * C::PC + (dt * 16 + c) * 4);
            }
            const float xa[4] = {x0[0], x0[1], x0[2], x0[3]}, xb[4] = {x1[0], x1[1], x1[2], x1[3]};
            P::mma(out[dt], a0, P::pack(xa));
            P::mma(out[dt], a1, P::pack(xb));
        }
    } else {
        float xv[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        u32x4 b = PrecBF16::pack(xv);
        // A[d = dt*16 + c][k] with k = tile rows {4g..4g+3, 16+4g..16+4g+3}: transposing read of the row-major chunk image
        // (lane (q = c>>2, p = c&3) addresses row R0 + q, columns 4p..4p+3 of the 16-column block; it receives column c)
        typedef short s16x4_ __attribute__((ext_vector_type(4)));
        typedef short s16x8_ __attribute__((ext_vector_type(8)));
        const int q = c >> 2, p = c & 3;
        (void)img_t;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const char* base = img_c + (dt * 16 + 4 * p) * 2;
            s16x4_ lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(base + (4 * g + q) * C::PC));
            s16x4_ hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_*)(base + (16 + 4 * g + q) * C::PC));
            s16x8_ t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            PrecBF16::mma(out[dt], __builtin_bit_cast(u32x4, t), b);
        }
    }
}

__device__ __forceinline__ float grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

constexpr float NEG_BIG = -1.0e30f;

// ------------------------------------------------------------------------------------------ forward
// RES (all three kernels): the images of EVERY 32-row tile of the head are staged up front into their own LDS block (one
// barrier, all global loads of the workgroup in flight together) and the tile loop then runs without barriers or global
// loads.  At L = 216 the tiled form spent 7 load -> barrier -> MFMA round trips per workgroup (16-20 us per launch, 96
// workgroups); used whenever the blocks fit the 160 KB LDS, else the tiled loop.
template <class P, int DH, bool RES>
__global__ void __launch_bounds__(256)
attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, uint16_t* __restrict__ outb, float* __restrict__ lse,
                int L, int heads, float scale) {
    using C = AttnCfg<P, DH>;
    constexpr int TILE_B = 2 * C::IMG_C;       // K and V chunk images
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const float* qb = qkv + (long)b * L * rs + head * DH;
    const float* kb_ = qb + Hd;
    const float* vb = qb + 2 * Hd;
    const int q = blockIdx.x * 64 + wave * 16 + c, qc = min(q, L - 1);
    u32x4 qf[C::KB];
    load_vec_frags<P, DH>(qb + (long)qc * rs, qf);
    f32x4 o[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = NEG_BIG, l = 0.f;
    if constexpr (RES) {
        for (int k0 = 0; k0 < L; k0 += 32) {
            char* base = lds + (k0 >> 5) * TILE_B;
            stage_tile<P, DH, true, false>(kb_, rs, k0, L, base, nullptr);
            stage_tile<P, DH, false, true>(vb, rs, k0, L, base + C::IMG_C, base + C::IMG_C);
        }
        __syncthreads();
    }
    for (int k0 = 0; k0 < L; k0 += 32) {
        char* kimg = lds + (RES ? (k0 >> 5) * TILE_B : 0);
        char* vimg = kimg + C::IMG_C;
        if constexpr (!RES) {
            __syncthreads();
            stage_tile<P, DH, true, false>(kb_, rs, k0, L, kimg, nullptr);
            stage_tile<P, DH, false, true>(vb, rs, k0, L, vimg, vimg);
            __syncthreads();
        }
        f32x4 s[2];
        float tmax = NEG_BIG;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            s[t] = prod_rows<P, DH>(kimg, t, qf);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int key = k0 + t * 16 + 4 * g + r;
                s[t][r] = key < L ? s[t][r] * scale : NEG_BIG;
                tmax = fmaxf(tmax, s[t][r]);
            }
        }
        tmax = grp_max(tmax);
        const float mn = fmaxf(m, tmax), alpha = expf(m - mn);
        float ps = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[t][r] = expf(s[t][r] - mn); ps += s[t][r]; }
        ps = grp_sum(ps);
        l = l * alpha + ps;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] *= alpha;
        prod_T<P, DH>(o, s[0], s[1], vimg, vimg);
    }
    if (q < L) {
        const float inv = 1.f / l;
        float* op = out + ((long)b * L + q) * Hd + head * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) *(f32x4*)(op + dt * 16) = o[dt] * inv;
        if (outb) {
            uint16_t* ob = outb + ((long)b * L + q) * Hd + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) *(bf16x4*)(ob + dt * 16) = __builtin_convertvector(o[dt] * inv, bf16x4);
        }
        if (g == 0) lse[((long)b * heads + head) * L + q] = m + logf(l);
    }
}

// ------------------------------------------------------------------------------- backward: dQ pass
template <class P, int DH, bool RES>
__global__ void __launch_bounds__(256)
attn_bwd_dq_kernel(const float* __restrict__ qkv, const float* __restrict__ out, const float* __restrict__ dout,
                   const float* __restrict__ lse, float* __restrict__ delta, float* __restrict__ dqkv, uint16_t* __restrict__ dqkvb,
                   int L, int heads, float scale) {
    using C = AttnCfg<P, DH>;
    constexpr int TILE_B = 2 * C::IMG_C + C::IMG_T;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const float* qb = qkv + (long)b * L * rs + head * DH;
    const float* kb_ = qb + Hd;
    const float* vb = qb + 2 * Hd;
    const int q = blockIdx.x * 64 + wave * 16 + c, qc = min(q, L - 1);
    u32x4 qf[C::KB], dof[C::KB];
    load_vec_frags<P, DH>(qb + (long)qc * rs, qf);
    load_vec_frags<P, DH>(dout + ((long)b * L + qc) * Hd + head * DH, dof);
    const float lq = lse[((long)b * heads + head) * L + qc];
    // delta[b,h,q] = sum_d dout[q,d] * out[q,d] for this lane's query: each lane group sums its chunks, two shuffles
    // finish the row; written out for the dK/dV pass that follows on the same stream (no separate delta launch)
    float dq_ = 0.f;
    {
        const float* orow = out + ((long)b * L + qc) * Hd + head * DH;
        const float* drow = dout + ((long)b * L + qc) * Hd + head * DH;
#pragma unroll
        for (int kb = 0; kb < C::KB; ++kb)
#pragma unroll
            for (int c4 = 0; c4 < C::CH / 4; ++c4) {
                const f32x4 a = *(const f32x4*)(orow + (kb * 4 + g) * C::CH + 4 * c4), d = *(const f32x4*)(drow + (kb * 4 + g) * C::CH + 4 * c4);
                dq_ += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
            }
        dq_ = grp_sum(dq_);
        if (g == 0 && q < L) delta[((long)b * heads + head) * L + q] = dq_;
    }
    f32x4 dq[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (RES) {
        for (int k0 = 0; k0 < L; k0 += 32) {
            char* base = lds + (k0 >> 5) * TILE_B;
            stage_tile<P, DH, true, true>(kb_, rs, k0, L, base, base + 2 * C::IMG_C);
            stage_tile<P, DH, true, false>(vb, rs, k0, L, base + C::IMG_C, nullptr);
        }
        __syncthreads();
    }
    for (int k0 = 0; k0 < L; k0 += 32) {
        char* kimg = lds + (RES ? (k0 >> 5) * TILE_B : 0);
        char* vimg = kimg + C::IMG_C;
        char* ktimg = kimg + 2 * C::IMG_C;
        if constexpr (!RES) {
            __syncthreads();
            stage_tile<P, DH, true, true>(kb_, rs, k0, L, kimg, ktimg);
            stage_tile<P, DH, true, false>(vb, rs, k0, L, vimg, nullptr);
            __syncthreads();
        }
        f32x4 ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = prod_rows<P, DH>(kimg, t, qf);
            f32x4 dp = prod_rows<P, DH>(vimg, t, dof);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int key = k0 + t * 16 + 4 * g + r;
                float p = key < L ? expf(s[r] * scale - lq) : 0.f;
                ds[t][r] = p * (dp[r] - dq_);
            }
        }
        prod_T<P, DH>(dq, ds[0], ds[1], kimg, ktimg);
    }
    if (q < L) {
        float* op = dqkv + ((long)b * L + q) * rs + head * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) *(f32x4*)(op + dt * 16) = dq[dt] * scale;
        if (dqkvb) {
            uint16_t* ob = dqkvb + ((long)b * L + q) * rs + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) *(bf16x4*)(ob + dt * 16) = __builtin_convertvector(dq[dt] * scale, bf16x4);
        }
    }
}

// ---------------------------------------------------------------------------- backward: dK/dV pass
template <class P, int DH, bool RES>
__global__ void __launch_bounds__(256)
attn_bwd_dkv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, const float* __restrict__ lse,
                    const float* __restrict__ delta, float* __restrict__ dqkv, uint16_t* __restrict__ dqkvb, int L, int heads, float scale) {
    using C = AttnCfg<P, DH>;
    constexpr int TILE_B = 2 * C::IMG_C + 2 * C::IMG_T + 256;    // q, dout chunk images, their transposed images, [32] lse + [32] delta
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int head = blockIdx.y, b = blockIdx.z, Hd = heads * DH;
    const long rs = 3L * Hd;
    const float* qb = qkv + (long)b * L * rs + head * DH;
    const float* kb_ = qb + Hd;
    const float* vb = qb + 2 * Hd;
    const float* dob = dout + (long)b * L * Hd + head * DH;
    const int key = blockIdx.x * 64 + wave * 16 + c, kc = min(key, L - 1);
    u32x4 kf[C::KB], vf[C::KB];
    load_vec_frags<P, DH>(kb_ + (long)kc * rs, kf);
    load_vec_frags<P, DH>(vb + (long)kc * rs, vf);
    f32x4 dk[C::DT], dv[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = dk[dt]; }
    auto stage_q = [&](int q0, char* base) {
        stage_tile<P, DH, true, true>(qb, rs, q0, L, base, base + 2 * C::IMG_C);
        stage_tile<P, DH, true, true>(dob, (long)Hd, q0, L, base + C::IMG_C, base + 2 * C::IMG_C + C::IMG_T);
        if (threadIdx.x < 32) {
            float* lt = (float*)(base + 2 * C::IMG_C + 2 * C::IMG_T);
            int qq = q0 + threadIdx.x;
            lt[threadIdx.x] = qq < L ? lse[((long)b * heads + head) * L + qq] : 1.0e30f;
            lt[32 + threadIdx.x] = qq < L ? delta[((long)b * heads + head) * L + qq] : 0.f;
        }
    };
    if constexpr (RES) {
        for (int q0 = 0; q0 < L; q0 += 32) stage_q(q0, lds + (q0 >> 5) * TILE_B);
        __syncthreads();
    }
    for (int q0 = 0; q0 < L; q0 += 32) {
        char* qimg = lds + (RES ? (q0 >> 5) * TILE_B : 0);
        char* doimg = qimg + C::IMG_C;
        char* qtimg = qimg + 2 * C::IMG_C;
        char* dotimg = qtimg + C::IMG_T;
        const float* lse_t = (const float*)(dotimg + C::IMG_T);
        const float* del_t = lse_t + 32;
        if constexpr (!RES) {
            __syncthreads();
            stage_q(q0, qimg);
            __syncthreads();
        }
        f32x4 p[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 s = prod_rows<P, DH>(qimg, t, kf);
            f32x4 dp = prod_rows<P, DH>(doimg, t, vf);
            f32x4 l4 = *(const f32x4*)(lse_t + t * 16 + 4 * g), d4 = *(const f32x4*)(del_t + t * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pv = expf(s[r] * scale - l4[r]);
                p[t][r] = pv;
                ds[t][r] = pv * (dp[r] - d4[r]);
            }
        }
        prod_T<P, DH>(dv, p[0], p[1], doimg, dotimg);
        prod_T<P, DH>(dk, ds[0], ds[1], qimg, qtimg);
    }
    if (key < L) {
        float* kp = dqkv + ((long)b * L + key) * rs + Hd + head * DH + 4 * g;
        float* vp = kp + Hd;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            *(f32x4*)(kp + dt * 16) = dk[dt] * scale;
            *(f32x4*)(vp + dt * 16) = dv[dt];
        }
        if (dqkvb) {
            uint16_t* kb = dqkvb + ((long)b * L + key) * rs + Hd + head * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                *(bf16x4*)(kb + dt * 16) = __builtin_convertvector(dk[dt] * scale, bf16x4);
                *(bf16x4*)(kb + Hd + dt * 16) = __builtin_convertvector(dv[dt], bf16x4);
            }
        }
    }
}

constexpr size_t ATTN_LDS_MAX = 150 * 1024;

template <class K, class... A>
static void launch_dyn(K kern, dim3 grid, size_t lds, hipStream_t st, A... args) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, args...);
}

template <class P, int DH>
int launch_fwd(const float* qkv, float* out, uint16_t* outb, float* lse, int B, int L, int heads, float scale, hipStream_t st) {
    using C = AttnCfg<P, DH>;
    const size_t tile = 2 * C::IMG_C, nt = cdiv(L, 32);
    dim3 grid(cdiv(L, 64), heads, B);
    if (tile * nt <= ATTN_LDS_MAX) launch_dyn(attn_fwd_kernel<P, DH, true>, grid, tile * nt, st, qkv, out, outb, lse, L, heads, scale);
    else launch_dyn(attn_fwd_kernel<P, DH, false>, grid, tile, st, qkv, out, outb, lse, L, heads, scale);
    return unetr_check_launch();
}
template <class P, int DH>
int launch_bwd(const float* qkv, const float* out, const float* dout, const float* lse, float* delta, float* dqkv, uint16_t* dqkvb,
               int B, int L, int heads, float scale, hipStream_t st) {
    using C = AttnCfg<P, DH>;
    dim3 grid(cdiv(L, 64), heads, B);
    const size_t nt = cdiv(L, 32), tq = 2 * C::IMG_C + C::IMG_T, tk = 2 * C::IMG_C + 2 * C::IMG_T + 256;
    if (tq * nt <= ATTN_LDS_MAX) launch_dyn(attn_bwd_dq_kernel<P, DH, true>, grid, tq * nt, st, qkv, out, dout, lse, delta, dqkv, dqkvb, L, heads, scale);
    else launch_dyn(attn_bwd_dq_kernel<P, DH, false>, grid, tq, st, qkv, out, dout, lse, delta, dqkv, dqkvb, L, heads, scale);
    if (tk * nt <= ATTN_LDS_MAX) launch_dyn(attn_bwd_dkv_kernel<P, DH, true>, grid, tk * nt, st, qkv, dout, lse, (const float*)delta, dqkv, dqkvb, L, heads, scale);
    else launch_dyn(attn_bwd_dkv_kernel<P, DH, false>, grid, tk, st, qkv, dout, lse, (const float*)delta, dqkv, dqkvb, L, heads, scale);
    return unetr_check_launch();
}

}  // namespace

#define ATTN_DISPATCH(CALL)                                                                   \
    if (prec == UNETR_PREC_F32) {                                                             \
        if (dh == 32) return CALL(PrecF32, 32);                                               \
        if (dh == 64) return CALL(PrecF32, 64);                                               \
        if (dh == 128) return CALL(PrecF32, 128);                                             \
    } else if (prec == UNETR_PREC_BF16) {                                                     \
        if (dh == 32) return CALL(PrecBF16, 32);                                              \
        if (dh == 64) return CALL(PrecBF16, 64);                                              \
        if (dh == 128) return CALL(PrecBF16, 128);                                            \
    } else if (prec == UNETR_PREC_BF16X3) {                                                   \
        if (dh == 32) return CALL(PrecBF16x3, 32);                                            \
        if (dh == 64) return CALL(PrecBF16x3, 64);                                            \
        if (dh == 128) return CALL(PrecBF16x3, 128);                                          \
    }                                                                                         \
    return UNETR_ERR_UNSUPPORTED;

extern "C" int unetr_attention_fwd(const float* qkv, float* out, void* out_bf16, float* lse, int B, int L, int heads, int dh,
                                   float scale, int prec, void* stream) {
    if (!qkv || !out || !lse || B <= 0 || L <= 0 || heads <= 0 || B > 65535 || heads > 65535) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define CALL_FWD(PP, DD) launch_fwd<PP, DD>(qkv, out, (uint16_t*)out_bf16, lse, B, L, heads, scale, st)
    ATTN_DISPATCH(CALL_FWD)
}

extern "C" int unetr_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                                   float* dqkv, void* dqkv_bf16, float* delta, int B, int L, int heads, int dh, float scale,
                                   int prec, void* stream) {
    if (!qkv || !out || !dout || !lse || !dqkv || !delta || B <= 0 || L <= 0 || heads <= 0 || B > 65535 || heads > 65535)
        return UNETR_ERR_ARG;
    if (dh & 3) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define CALL_BWD(PP, DD) launch_bwd<PP, DD>(qkv, out, dout, lse, delta, dqkv, (uint16_t*)dqkv_bf16, B, L, heads, scale, st)
    ATTN_DISPATCH(CALL_BWD)
}
