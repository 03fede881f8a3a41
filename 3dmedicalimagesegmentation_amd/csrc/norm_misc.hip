// HBM-bound kernels of the UNETR hot path: LayerNorm, InstanceNorm(+LeakyReLU+residual), column sums,
// layout moves, patch gather, the 1x1x1 output conv (NCDHW logits) and fused AdamW.
// All reductions are wave64 shuffle reductions + fixed-order partial buffers (bitwise reproducible).
#include <algorithm>
#include <cstdlib>
#include "common.hpp"
#include "../../include/unetr_hip.h"

namespace {

constexpr int LN_MAXV_MAX = 8;  // float4 per lane -> H <= 2048; kernels are instantiated for 3 (H <= 768), 4 and 8

// --------------------------------------------------------------------------------------- LayerNorm
// where layernorm_fwd_kernel takes its rows from: memory (splits <= 1), or the split-K partial slabs of a GEMM + its epilogue
struct LnFwdSrc { int splits; long slab; const float* bias; const float* res; long ldr; int res_mod; float* xout; };

template <int LN_MAXV>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                     float* __restrict__ y, uint16_t* __restrict__ yb, float* __restrict__ mean, float* __restrict__ rstd,
                     int M, int H, float eps, LnFwdSrc src) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = H >> 2;
    const f32x4* xr = (const f32x4*)(x + (long)row * H);
    f32x4 v[LN_MAXV], gam[LN_MAXV], bet[LN_MAXV];
    float s = 0.f;
    // every load of the row -- gamma / beta included -- is requested before the first use (as run-time loops over the slabs the
    // split-K form made one memory round trip per slab and vector, one after the other)
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        const int i = min(lane + 64 * j, nv - 1);
        gam[j] = ((const f32x4*)gamma)[i];
        bet[j] = ((const f32x4*)beta)[i];
    }
    if (src.splits > 1) {
        // the row arrives as split-K partial slabs of the GEMM that produces it: summed from 0.f in slab order, + bias,
        // + residual -- the arithmetic of splitk_reduce_kernel + EpBf::store, whose launch this saves -- and written out
        f32x4 sl[LN_MAXV][4], bi[LN_MAXV], re[LN_MAXV];
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int i = min(lane + 64 * j, nv - 1);
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) sl[j][sp] = ((const f32x4*)(x + (sp < src.splits ? sp : 0) * src.slab + (long)row * H))[i];
            bi[j] = src.bias ? ((const f32x4*)src.bias)[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            re[j] = src.res ? ((const f32x4*)(src.res + (long)(row % src.res_mod) * src.ldr))[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int i = lane + 64 * j;
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (i < nv) {
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    if (sp < src.splits) t += sl[j][sp];
                for (int sp = 4; sp < src.splits; ++sp) t += ((const f32x4*)(x + sp * src.slab + (long)row * H))[i];
                if (src.bias) t += bi[j];
                if (src.res) t += re[j];
                ((f32x4*)(src.xout + (long)row * H))[i] = t;
            }
            v[j] = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < nv ? xr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    const float mu = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int i = lane + 64 * j;
        if (i < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float d = v[j][e] - mu; q += d * d; }
        }
    }
    const float rs = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    f32x4* yr = (f32x4*)(y + (long)row * H);
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int i = lane + 64 * j;
        if (i < nv) {
            const f32x4 g = gam[j], b = bet[j];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * g[e] + b[e];
            if (y) yr[i] = o;
            if (yb) ((bf16x4*)(yb + (long)row * H))[i] = __builtin_convertvector(o, bf16x4);
        }
    }
}

constexpr int LN_RPB = 4;   // rows per block in backward (one per wave): 108 workgroups at M = 432 instead of 27
template <int LN_MAXV>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const float* __restrict__ dy, int splits, long slab, const float* __restrict__ x, const float* __restrict__ gamma,
                     const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                     uint16_t* __restrict__ dxb, const float* __restrict__ dres, float* __restrict__ part, int M, int H) {
    // dy may arrive as `splits` split-K partial slabs of the GEMM that produced it (slab = elements between slabs): they are
    // summed here in slab order from 0.f -- the arithmetic of splitk_reduce_kernel, whose launch this saves
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [4 waves][2][H]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = H >> 2;
    f32x4 dg[LN_MAXV], db[LN_MAXV];
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) { dg[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; db[j] = dg[j]; }
    for (int rr = wave; rr < LN_RPB; rr += 4) {
        const int row = blockIdx.x * LN_RPB + rr;
        if (row >= M) break;
        const float mu = mean[row], rs = rstd[row];
        const f32x4* xr = (const f32x4*)(x + (long)row * H);
        const f32x4* dyr = (const f32x4*)(dy + (long)row * H);
        f32x4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
        // every load of the row is requested before the first use: x, gamma, up to four split-K slabs of dy per vector (a slab
        // past `splits` re-reads slab 0 and is not added -- branch-free) and the residual gradient.  As a run-time loop over
        // the slabs inside the vector loop the kernel made 9-12 memory round trips one after the other, on data the producer
        // GEMM has just written back across the kernel boundary: 8.8 us per launch in the step against 4.3 us warm
        f32x4 xv_[LN_MAXV], gm_[LN_MAXV], sl_[LN_MAXV][4], rs_[LN_MAXV];
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            const int i = min(lane + 64 * j, nv - 1);
            xv_[j] = xr[i];
            gm_[j] = ((const f32x4*)gamma)[i];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) sl_[j][sp] = ((const f32x4*)(dy + (sp < splits ? sp : 0) * slab + (long)row * H))[i];
            rs_[j] = dres ? ((const f32x4*)(dres + (long)row * H))[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            int i = lane + 64 * j;
            if (i < nv) {
                f32x4 xv = xv_[j], dv = {0.f, 0.f, 0.f, 0.f}, gm = gm_[j];
                if (splits <= 1) dv = sl_[j][0];
                else {
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp)
                        if (sp < splits) dv += sl_[j][sp];
                    for (int sp = 4; sp < splits; ++sp) dv += ((const f32x4*)(dy + sp * slab + (long)row * H))[i];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float h = (xv[e] - mu) * rs;
                    xh[j][e] = h;
                    g[j][e] = dv[e] * gm[e];
                    s1 += g[j][e];
                    s2 += g[j][e] * h;
                    dg[j][e] += dv[e] * h;
                    db[j][e] += dv[e];
                }
            } else { xh[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; g[j] = xh[j]; }
        }
        const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
        f32x4* dxr = (f32x4*)(dx + (long)row * H);
#pragma unroll
        for (int j = 0; j < LN_MAXV; ++j) {
            int i = lane + 64 * j;
            if (i < nv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[j][e] - c1 - xh[j][e] * c2);
                if (dres) o += rs_[j];
                dxr[i] = o;
                if (dxb) ((bf16x4*)(dxb + (long)row * H))[i] = __builtin_convertvector(o, bf16x4);
            }
        }
    }
    // cross-wave reduction of the dgamma/dbeta partials
    f32x4* l4 = (f32x4*)lds;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
        int i = lane + 64 * j;
        if (i < nv) { l4[(wave * 2 + 0) * nv + i] = dg[j]; l4[(wave * 2 + 1) * nv + i] = db[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nv; i += 256) {
        int which = i / nv, c = i - which * nv;
        f32x4 s = l4[(0 * 2 + which) * nv + c];
        s += l4[(1 * 2 + which) * nv + c];
        s += l4[(2 * 2 + which) * nv + c];
        s += l4[(3 * 2 + which) * nv + c];
        ((f32x4*)part)[((long)blockIdx.x * 2 + which) * nv + c] = s;
    }
}

// out[which*H + n] = sum_blocks part[(blk*2+which)*H + n]; 64 columns x 4 partial-phases per workgroup
__global__ void __launch_bounds__(256)
ln_finalize_kernel(const float* __restrict__ part, int nblk, int H, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const int which = i / H, n = i - which * H;
    float s = 0.f;
    if (i < 2 * H)
        for (int b = ph; b < nblk; b += 4) s += part[((long)b * 2 + which) * H + n];
    sm[ph][tx] = s;
    __syncthreads();
    if (ph == 0 && i < 2 * H) (which ? dbeta : dgamma)[n] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

// ------------------------------------------------------------------------------------------ colsum
// grid (cdiv(N,64), RB): block sums rows rb, rb+RB, ... of 64 columns
__global__ void __launch_bounds__(256)
colsum_kernel(const float* __restrict__ x, long ld, int M, int N, float* __restrict__ out, int rb_count) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (n < N)
        for (int m = blockIdx.y * 4 + ty; m < M; m += 4 * rb_count) s += x[(long)m * ld + n];
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < N) out[(long)blockIdx.y * N + n] = sm[0][tx] + sm[1][tx] + sm[2][tx] + sm[3][tx];
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int RB, int N, float* __restrict__ out, int accumulate) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int r = 0; r < RB; ++r) s += part[(long)r * N + n];
    out[n] = accumulate ? out[n] + s : s;
}

// grouped column sums: up to CS_MAX independent [M_i, N_i] -> [N_i] problems in one launch (descriptors in kernargs)
constexpr int CS_MAX = 96;       // (x 40 bytes: the kernel-argument block holds 4 KB)
struct ColsumProblem { const float* x; float* out; long ld; int M, N, blk0, vec; };   // vec 2: x is bf16, 16-byte loads of 8 columns
struct ColsumArgs { int n; ColsumProblem p[CS_MAX]; };
__global__ void __launch_bounds__(256)
colsum_grouped_kernel(ColsumArgs a) {
    __shared__ f32x4 sm4[16][16];
    int pi = 0, hi = a.n - 1;                  // last problem whose first block <= blockIdx.x (binary search: ~85 problems, 4000 blocks)
    while (pi < hi) {
        const int mid = (pi + hi + 1) >> 1;
        if ((int)blockIdx.x >= a.p[mid].blk0) pi = mid; else hi = mid - 1;
    }
    const ColsumProblem& pr = a.p[pi];
    const int n0 = ((int)blockIdx.x - pr.blk0) * 64;
    if (pr.vec == 2) {
        // bf16 rows: thread = (8 columns cg of the block's 64, row phase ph of 32); same fixed order as below
        const int cg = threadIdx.x & 7, ph = threadIdx.x >> 3, n = n0 + 8 * cg;
        const uint16_t* xb = (const uint16_t*)pr.x;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (n < pr.N) {
            // eight rows in flight per thread (one load per iteration cost a memory round trip per row: 14 of them for 432 rows)
            int m = ph;
            for (; m + 7 * 32 < pr.M; m += 8 * 32) {
                u32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *(const u32x4*)(xb + (long)(m + 32 * u) * pr.ld + n);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bf16x8 tv = __builtin_bit_cast(bf16x8, t[u]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] += (float)tv[e];
                }
            }
            for (; m < pr.M; m += 32) {
                const bf16x8 t = __builtin_bit_cast(bf16x8, *(const u32x4*)(xb + (long)m * pr.ld + n));
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += (float)t[e];
            }
        }
        __shared__ float smb[32][64];
#pragma unroll
        for (int e = 0; e < 8; ++e) smb[ph][8 * cg + e] = s[e];
        __syncthreads();
        if (threadIdx.x < 64 && n0 + (int)threadIdx.x < pr.N) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = (smb[4 * q][threadIdx.x] + smb[4 * q + 1][threadIdx.x]) + (smb[4 * q + 2][threadIdx.x] + smb[4 * q + 3][threadIdx.x]);
            pr.out[n0 + threadIdx.x] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        return;
    }
    if (pr.vec) {
        // 16-byte loads: thread = (4 columns cg, row phase ph of 16); a block still owns 64 columns.  (With one 4-byte load per
        // lane the launch was issue-bound: 95 MB of fp32 gradients at 2 TB/s.)  Fixed order: rows ph, ph+16, ... in groups of
        // four independent loads, then the 16 phases pairwise.
        const int cg = threadIdx.x & 15, ph = threadIdx.x >> 4, n = n0 + 4 * cg;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (n < pr.N) {
            const float* px = pr.x + n;
            int m = ph;
            for (; m + 7 * 16 < pr.M; m += 8 * 16) {          // eight rows in flight per thread
                f32x4 t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(px + (long)(m + 16 * u) * pr.ld);
                s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
            }
            for (; m + 48 < pr.M; m += 64) {
                const f32x4 t0 = *(const f32x4*)(px + (long)m * pr.ld), t1 = *(const f32x4*)(px + (long)(m + 16) * pr.ld);
                const f32x4 t2 = *(const f32x4*)(px + (long)(m + 32) * pr.ld), t3 = *(const f32x4*)(px + (long)(m + 48) * pr.ld);
                s += (t0 + t1) + (t2 + t3);
            }
            for (; m < pr.M; m += 16) s += *(const f32x4*)(px + (long)m * pr.ld);
        }
        sm4[ph][cg] = s;
        __syncthreads();
        if (ph == 0 && n < pr.N) {
            f32x4 t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = sm4[2 * q][cg] + sm4[2 * q + 1][cg];
            *(f32x4*)(pr.out + n) = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        return;
    }
    float (*sm)[64] = (float (*)[64])&sm4[0][0];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = n0 + tx;
    float s = 0.f;
    if (n < pr.N) {
        int m = ty;
        for (; m + 28 < pr.M; m += 32) {           // eight independent loads in flight per thread (fixed summation tree)
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pr.x[(long)(m + 4 * u) * pr.ld + n];
            s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; m < pr.M; m += 4) s += pr.x[(long)m * pr.ld + n];
    }
    sm[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && n < pr.N) pr.out[n] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

// ---- every weight-gradient reduction of a backward pass in ONE launch: dst_p[i] = sum_g part_p[g][i] for each problem p (the
// per-workgroup partial sums of the 3x3x3 / 2x2x2 conv weight-gradient kernels).  Same body and summation order as
// conv3_wgrad_reduce_kernel (32 outputs x 8 row phases per workgroup, eight loads of a phase in flight, phases added in order);
// a workgroup finds its problem by binary search over the kernel-argument table.  Fourteen ~6 us launches per step before.
constexpr int RR_MAX = 48;
struct RrProblem { const float* part; float* dst; long n; int G; int blk0; int nblk; int vec; };
struct RrArgs { int n; RrProblem p[RR_MAX]; };
__global__ void __launch_bounds__(256) reduce_rows_grouped_kernel(RrArgs a) {
    int pi = 0, hi_ = a.n - 1;
    while (pi < hi_) {
        const int mid = (pi + hi_ + 1) >> 1;
        if ((int)blockIdx.x >= a.p[mid].blk0) pi = mid; else hi_ = mid - 1;
    }
    const RrProblem& pr = a.p[pi];
    const float* __restrict__ part = pr.part;
    float* __restrict__ dw = pr.dst;
    const long n = pr.n;
    const int G = pr.G, bx = (int)blockIdx.x - pr.blk0, nbx = pr.nblk;
    const int o = threadIdx.x & 31, ph = threadIdx.x >> 5;
    if (pr.vec) {
        // 16-byte loads: a lane owns FOUR consecutive elements (a block iteration covers 128), same eight row phases and the same
        // summation tree per element as the scalar form below, so both give the same bits (with one 4-byte load per lane the launch
        // read its 130 MB of partial rows at 2 TB/s: issue-bound, as the column sums were before they went to 16-byte loads)
        __shared__ f32x4 sm4[8][33];
        for (long i0 = (long)bx * 128; i0 < n; i0 += (long)nbx * 128) {
            const long i = i0 + 4 * o;
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (i < n) {
                int gI = ph;
                for (; gI + 56 < G; gI += 64) {
                    f32x4 t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(part + (long)(gI + 8 * u) * n + i);
                    s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
                }
                for (; gI < G; gI += 8) s += *(const f32x4*)(part + (long)gI * n + i);
            }
            sm4[ph][o] = s;
            __syncthreads();
            if (ph == 0 && i < n) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < 8; ++p) t += sm4[p][o];
                *(f32x4*)(dw + i) = t;
            }
            __syncthreads();
        }
        return;
    }
    __shared__ float sm[8][33];
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

// ------------------------------------------------------------------------------------ InstanceNorm
// x: [B, V, C] pitch ld.  thread -> (channel vec cv, voxel phase); block covers VPB voxels of one batch item.
constexpr int IN_VPB = 1024;

// Thread -> (channel group cv of W channels, voxel phase); W = Io<T>::W = one 16-byte access (4 fp32 / 8 bf16 channels): with
// bf16 storage a 4-channel granule would be an 8-byte access and the passes turn issue-bound instead of bandwidth-bound.
template <int NS, int W>  // NS sums per channel
__device__ __forceinline__ void in_block_reduce(float (&acc)[NS][W], int cvn, int nphase, float* lds, float* part_out, int C) {
    // lds: [NS][nphase][cvn][W] floats
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    if (ph < nphase)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < W; ++e) lds[((s * nphase + ph) * cvn + cv) * W + e] = acc[s][e];
    __syncthreads();
    for (int i = threadIdx.x; i < NS * C; i += blockDim.x) {
        const int s = i / C, c = i - s * C;
        float t = 0.f;
        for (int p = 0; p < nphase; ++p) t += lds[(s * nphase + p) * C + c];
        part_out[(long)s * C + c] = t;       // part_out: [NS][C]
    }
}

template <class T>
__global__ void __launch_bounds__(256)
in_stats_kernel(const T* __restrict__ x, long ld, long V, int C, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = Io<T>::W;
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * IN_VPB, v1 = std::min<long>(V, v0 + IN_VPB);
    float acc[2][W];
#pragma unroll
    for (int e = 0; e < W; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    if (ph < nphase)
        for (long v = v0 + ph; v < v1; v += nphase) {
            float t[W];
            Io<T>::ldw(x + ((long)b * V + v) * ld + W * cv, t);
#pragma unroll
            for (int e = 0; e < W; ++e) { acc[0][e] += t[e]; acc[1][e] += t[e] * t[e]; }
        }
    in_block_reduce<2, W>(acc, cvn, nphase, lds, part + ((long)b * gridDim.x + blockIdx.x) * 2 * C, C);
}

__global__ void in_stats_final_kernel(const float* __restrict__ part, int nchunk, long V, int C, float eps,
                                      float* __restrict__ stats, int B) {
    // one wave per (b, c): lanes stride over the chunk partials, fixed-order shuffle tree in double
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < nchunk; k += 64) {
        const float* p = part + ((long)b * nchunk + k) * 2 * C;
        s += (double)p[c];
        q += (double)p[C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) {
        double mu = s / (double)V, var = q / (double)V - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mu;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__device__ __forceinline__ float lrelu_f(float v) { return v > 0.f ? v : 0.01f * v; }

template <class T>
__global__ void __launch_bounds__(256)
in_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ sa, const T* __restrict__ x2, long ldx2,
                const float* __restrict__ sb, T* __restrict__ y, long ldy, int B, long V, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    const int cvn = C / W;
    const long total = (long)B * V * cvn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cv = (int)(i % cvn); long vox = i / cvn; int b = (int)(vox / V);
        float t[W], t2[W], o[W];
        Io<T>::ldw(x + vox * ldx + W * cv, t);
        if (x2) Io<T>::ldw(x2 + vox * ldx2 + W * cv, t2);
        const float* s = sa + ((long)b * C + W * cv) * 2;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = (t[e] - s[2 * e]) * s[2 * e + 1];
        if (x2) {
            const float* s2 = sb + ((long)b * C + W * cv) * 2;
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] += (t2[e] - s2[2 * e]) * s2[2 * e + 1];
        }
        if (lrelu) {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = lrelu_f(o[e]);
        }
        Io<T>::stw(y + vox * ldy + W * cv, o);
    }
}

// The same pass for channel counts whose 16-byte pieces divide the block (C / W a power of two <= 256: every layer of the
// network): block = (voxel chunk, batch item), thread = (piece cv, phase), so the statistics of the thread's channels are
// folded into multiply-add coefficients ONCE instead of being re-read from global memory for every element (the generic
// kernel above issues 32 scalar loads of statistics per 16 bytes of data).
template <class T, bool DUAL>
__global__ void __launch_bounds__(256)
in_apply_hoist_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ sa, const T* __restrict__ x2, long ldx2,
                      const float* __restrict__ sb, T* __restrict__ y, long ldy, long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W, U = 2;
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const float* s1 = sa + ((long)b * C + W * cv) * 2;
    const float* s2 = DUAL ? sb + ((long)b * C + W * cv) * 2 : nullptr;
    float a1[W], o1[W], a2[DUAL ? W : 1];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
        if (DUAL) { a2[e] = s2[2 * e + 1]; o1[e] -= s2[2 * e] * a2[e]; }      // both offsets in one constant
    }
    const float slope = lrelu ? 0.01f : 1.f;
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = DUAL ? x2 + ((long)b * V) * ldx2 + W * cv : nullptr;
    T* py = y + ((long)b * V) * ldy + W * cv;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rt[U], rt2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rt[u] = *(const u32x4*)(px + vc * ldx);
            rt2[u] = DUAL ? *(const u32x4*)(px2 + vc * ldx2) : rt[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t[W], t2[W], o[W];
            Io<T>::unpack(rt[u], t);
            if (DUAL) Io<T>::unpack(rt2[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float n = fmaf(t[e], a1[e], o1[e]);
                if (DUAL) n = fmaf(t2[e], a2[e], n);
                o[e] = n > 0.f ? n : slope * n;
            }
            if (live[u]) Io<T>::stw(py + (v + (long)u * nphase) * ldy, o);
        }
    }
}

// backward stage 1: per (b,c) sums of g, g*n1, g*n2 with g = dy * lrelu'(n1+n2).  DUAL = the block-end form (two normalised
// inputs summed before the activation).  One block = `vpb` consecutive voxels of one batch element (the host sizes vpb so
// that the grid is about three blocks per CU: with 1024-voxel blocks the prologue / block reduction were most of a block's
// life and the pass ran at 2.2 TB/s).  Loaded data stays packed (16 bytes = 4 registers) until it is used, addresses are a
// uniform base + a 32-bit per-lane offset, and the statistics are folded into one multiply-add per element.
template <class T, bool DUAL, int U>
__global__ void __launch_bounds__(256)
in_bwd_reduce_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                     const T* __restrict__ x2, long ldx2, const float* __restrict__ sb, long V, long vpb, int C, int lrelu,
                     float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = Io<T>::W;
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb;
    const int n = (int)(std::min<long>(V, v0 + vpb) - v0);
    float acc[3][W];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < W; ++e) acc[s][e] = 0.f;
    if (ph < nphase) {
        const float* s1 = sa + ((long)b * C + W * cv) * 2;
        const float* s2 = DUAL ? sb + ((long)b * C + W * cv) * 2 : nullptr;
        // n = t * a + o  with a = rstd, o = -mean * rstd
        float a1[W], o1[W], a2[DUAL ? W : 1], o2[DUAL ? W : 1];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
            if (DUAL) { a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e]; }
        }
        const float slope = lrelu ? 0.01f : 1.f;
        auto add = [&](u32x4 rg, u32x4 rt, u32x4 rt2) {
            float g[W], t[W], t2[W];
            Io<T>::unpack(rg, g);
            Io<T>::unpack(rt, t);
            if (DUAL) Io<T>::unpack(rt2, t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float n1 = fmaf(t[e], a1[e], o1[e]);
                const float n2 = DUAL ? fmaf(t2[e], a2[e], o2[e]) : 0.f;
                const float ge = (n1 + n2) > 0.f ? g[e] : slope * g[e];
                acc[0][e] += ge;
                acc[1][e] = fmaf(ge, n1, acc[1][e]);
                if (DUAL) acc[2][e] = fmaf(ge, n2, acc[2][e]);
            }
        };
        // uniform bases (scalar registers) + 32-bit byte offsets per lane
        const char* bd = (const char*)(dy + ((long)b * V + v0) * lddy);
        const char* bx = (const char*)(x + ((long)b * V + v0) * ldx);
        const char* bx2 = DUAL ? (const char*)(x2 + ((long)b * V + v0) * ldx2) : nullptr;
        const unsigned esz = sizeof(T);
        unsigned od = (unsigned)(ph * lddy + W * cv) * esz, ox = (unsigned)(ph * ldx + W * cv) * esz;
        unsigned ox2 = DUAL ? (unsigned)(ph * ldx2 + W * cv) * esz : 0u;
        const unsigned sd = (unsigned)(nphase * lddy) * esz, sx = (unsigned)(nphase * ldx) * esz, sx2 = DUAL ? (unsigned)(nphase * ldx2) * esz : 0u;
        // several voxels per iteration, all their loads issued before the first use
        int v = ph;
#pragma unroll 1
        for (; v + (U - 1) * nphase < n; v += U * nphase) {
            u32x4 rg[U], rt[U], rt2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                rt2[u] = (u32x4){0u, 0u, 0u, 0u};
                rg[u] = *(const u32x4*)(bd + (size_t)(od + u * sd));
                rt[u] = *(const u32x4*)(bx + (size_t)(ox + u * sx));
                if (DUAL) rt2[u] = *(const u32x4*)(bx2 + (size_t)(ox2 + u * sx2));
            }
            od += U * sd; ox += U * sx; ox2 += U * sx2;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // one voxel's unpacked values live at a time: the empty asm redefines this voxel's packed registers together
                // with the accumulators the previous voxel wrote, so the compiler cannot interleave the four voxels'
                // arithmetic (which cost 244 VGPRs = two waves per SIMD)
                static_assert(W == 4 || W == 8, "");
                if constexpr (DUAL) asm volatile("" : "+v"(rt2[u]));
                if (W == 8) asm volatile("" : "+v"(rg[u]), "+v"(rt[u]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                                         "+v"(acc[1][3]), "+v"(acc[1][4 % W]), "+v"(acc[1][5 % W]), "+v"(acc[1][6 % W]), "+v"(acc[1][7 % W]));
                else asm volatile("" : "+v"(rg[u]), "+v"(rt[u]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]));
                add(rg[u], rt[u], rt2[u]);
            }
        }
#pragma unroll 1
        for (; v < n; v += nphase) {
            const u32x4 rg = *(const u32x4*)(bd + (size_t)od), rt = *(const u32x4*)(bx + (size_t)ox);
            const u32x4 rt2 = DUAL ? *(const u32x4*)(bx2 + (size_t)ox2) : rt;
            od += sd; ox += sx; ox2 += sx2;
            add(rg, rt, rt2);
        }
    }
    in_block_reduce<3, W>(acc, cvn, nphase, lds, part + ((long)b * gridDim.x + blockIdx.x) * 3 * C, C);
}

__global__ void in_bwd_final_kernel(const float* __restrict__ part, int nchunk, long V, int C, int B, float* __restrict__ sums) {
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= B * C) return;
    const int b = i / C, c = i - b * C;
    double s[3] = {0.0, 0.0, 0.0};
    int k = lane;
    for (; k + 3 * 64 < nchunk; k += 4 * 64) {             // four chunk rows in flight per lane (same summation order)
        float t[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = part + ((long)b * nchunk + k + 64 * u) * 3 * C;
            t[u][0] = p[c]; t[u][1] = p[C + c]; t[u][2] = p[2 * C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s[0] += (double)t[u][0]; s[1] += (double)t[u][1]; s[2] += (double)t[u][2]; }
    }
    for (; k < nchunk; k += 64) {
        const float* p = part + ((long)b * nchunk + k) * 3 * C;
        s[0] += (double)p[c]; s[1] += (double)p[C + c]; s[2] += (double)p[2 * C + c];
    }
    for (int j = 0; j < 3; ++j) {
        double t = wave_sum_d(s[j]);
        if (lane == 0) sums[3 * i + j] = (float)(t / (double)V);
    }
}

template <class T>
__global__ void __launch_bounds__(256)
in_bwd_apply_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                    const T* __restrict__ x2, long ldx2, const float* __restrict__ sb, const float* __restrict__ sums,
                    T* __restrict__ dx, long lddx, T* __restrict__ dx2, long lddx2, int B, long V, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    const int cvn = C / W;
    const long total = (long)B * V * cvn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cv = (int)(i % cvn); long vox = i / cvn; int b = (int)(vox / V);
        const float* s1 = sa + ((long)b * C + W * cv) * 2;
        const float* sm = sums + ((long)b * C + W * cv) * 3;
        float g[W], t[W], t2[W], n1[W], n2[W], o[W];
        Io<T>::ldw(dy + vox * lddy + W * cv, g);
        Io<T>::ldw(x + vox * ldx + W * cv, t);
        if (x2) Io<T>::ldw(x2 + vox * ldx2 + W * cv, t2);
#pragma unroll
        for (int e = 0; e < W; ++e) { n1[e] = (t[e] - s1[2 * e]) * s1[2 * e + 1]; n2[e] = 0.f; }
        const float* s2 = nullptr;
        if (x2) {
            s2 = sb + ((long)b * C + W * cv) * 2;
#pragma unroll
            for (int e = 0; e < W; ++e) n2[e] = (t2[e] - s2[2 * e]) * s2[2 * e + 1];
        }
        if (lrelu) {
#pragma unroll
            for (int e = 0; e < W; ++e) g[e] = (n1[e] + n2[e]) > 0.f ? g[e] : 0.01f * g[e];
        }
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = s1[2 * e + 1] * (g[e] - sm[3 * e] - n1[e] * sm[3 * e + 1]);
        Io<T>::stw(dx + vox * lddx + W * cv, o);
        if (x2) {
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = s2[2 * e + 1] * (g[e] - sm[3 * e] - n2[e] * sm[3 * e + 2]);
            Io<T>::stw(dx2 + vox * lddx2 + W * cv, o);
        }
    }
}

// hoisted-coefficient form of the pass above (see in_apply_hoist_kernel): dx = a1*g - a1*m0 - a1*m1*n1 with g the
// lrelu-masked gradient, n1 = t*a1 + o1; the second branch alike.  56 scalar loads of statistics per 16 bytes before.
template <class T, bool DUAL>
__global__ void __launch_bounds__(256)
in_bwd_apply_hoist_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                          const T* __restrict__ x2, long ldx2, const float* __restrict__ sb, const float* __restrict__ sums,
                          T* __restrict__ dx, long lddx, T* __restrict__ dx2, long lddx2, long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W, U = 2;
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const float* s1 = sa + ((long)b * C + W * cv) * 2;
    const float* s2 = DUAL ? sb + ((long)b * C + W * cv) * 2 : nullptr;
    const float* sm = sums + ((long)b * C + W * cv) * 3;
    float a1[W], o1[W], k0[W], k1[W], a2[DUAL ? W : 1], o2[DUAL ? W : 1], q0[DUAL ? W : 1], q2[DUAL ? W : 1];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
        k0[e] = a1[e] * sm[3 * e]; k1[e] = a1[e] * sm[3 * e + 1];
        if (DUAL) {
            a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e];
            q0[e] = a2[e] * sm[3 * e]; q2[e] = a2[e] * sm[3 * e + 2];
        }
    }
    const float slope = lrelu ? 0.01f : 1.f;
    const T* pg = dy + ((long)b * V) * lddy + W * cv;
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = DUAL ? x2 + ((long)b * V) * ldx2 + W * cv : nullptr;
    T* pd = dx + ((long)b * V) * lddx + W * cv;
    T* pd2 = DUAL ? dx2 + ((long)b * V) * lddx2 + W * cv : nullptr;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rg[U], rt[U], rt2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rg[u] = *(const u32x4*)(pg + vc * lddy);
            rt[u] = *(const u32x4*)(px + vc * ldx);
            rt2[u] = DUAL ? *(const u32x4*)(px2 + vc * ldx2) : rt[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[W], t[W], t2[W], o[W], p[W];
            Io<T>::unpack(rg[u], g);
            Io<T>::unpack(rt[u], t);
            if (DUAL) Io<T>::unpack(rt2[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float n1 = fmaf(t[e], a1[e], o1[e]);
                const float n2 = DUAL ? fmaf(t2[e], a2[e], o2[e]) : 0.f;
                const float ge = (n1 + n2) > 0.f ? g[e] : slope * g[e];
                o[e] = fmaf(a1[e], ge, -k0[e]) - k1[e] * n1;
                if (DUAL) p[e] = fmaf(a2[e], ge, -q0[e]) - q2[e] * n2;
            }
            if (live[u]) {
                Io<T>::stw(pd + (v + (long)u * nphase) * lddx, o);
                if (DUAL) Io<T>::stw(pd2 + (v + (long)u * nphase) * lddx2, p);
            }
        }
    }
}

// ---- finalisation folded into the consumer ------------------------------------------------------------------------------
// The tiny finalize launches (partial rows -> per-(batch item, channel) statistics) cost ~5 us each inside the captured step
// and there were twenty of them.  Here every consumer block reduces the partial rows of ITS batch item in its prologue:
// part[nrows][NSP][C] floats -> sums of columns (s, c), accumulated in double in a fixed order (thread = (column, row phase),
// 8 rows in flight per thread, phases added in order through LDS), so every block of a launch forms the same bits.  The rows sit
// in L2 (written by the launch before); a block reads nrows * NSP * C * 4 bytes (<= 64 KB for the network's layers).
// red: 256 doubles of LDS; out: NSP * C doubles of LDS (sums, not yet divided).  C * NSP <= IN_FIN_MAXCOL.
constexpr int IN_FIN_MAXCOL = 3 * 128;
constexpr int IN_FIN_NT = 1024;         // threads of a FIN block: ONE block per CU, so that a CU pulls its batch item's rows once
// columns [0, ncol) of part (and, when part2 != nullptr, columns [ncol, 2 ncol) of the combined result from part2, which has
// nrows2 rows of the same width): out[col] = sum over rows, double, fixed order
__device__ __forceinline__ void in_fin_sums(const float* __restrict__ part, int nrows, const float* __restrict__ part2, int nrows2, int ncol,
                                            double* red, double* out) {
    const int tcol = part2 ? 2 * ncol : ncol;
    const int cpb = tcol < IN_FIN_NT ? tcol : IN_FIN_NT, nph = IN_FIN_NT / cpb;
    const int lc = threadIdx.x % cpb, ph = threadIdx.x / cpb;
    for (int c0 = 0; c0 < tcol; c0 += cpb) {
        const int col = c0 + lc;
        double s = 0.0;
        if (ph < nph && col < tcol) {
            const bool second = col >= ncol;
            const float* __restrict__ p = second ? part2 + (col - ncol) : part + col;
            const int nr = second ? nrows2 : nrows;
            int k = ph;
            for (; k + 15 * nph < nr; k += 16 * nph) {
                float t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = p[(long)(k + u * nph) * ncol];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += (double)t[u];
            }
            for (; k + 3 * nph < nr; k += 4 * nph) {
                float t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = p[(long)(k + u * nph) * ncol];
#pragma unroll
                for (int u = 0; u < 4; ++u) s += (double)t[u];
            }
            for (; k < nr; k += nph) s += (double)p[(long)k * ncol];
        }
        if (ph < nph) red[ph * cpb + lc] = s;
        __syncthreads();
        if (ph == 0 && col < tcol) {
            double t = 0.0;
            for (int q = 0; q < nph; ++q) t += red[q * cpb + lc];
            out[col] = t;
        }
        __syncthreads();
    }
}
// (mean, rstd) from the sums of one statistics set: sums[0..C) = sum, sums[C..2C) = sum of squares -> st[C][2] (LDS); block
// x == 0 also writes them to stats_out_b[C][2] for the consumers that come later (the backward kernels)
__device__ __forceinline__ void in_fin_stats(const double* sums, int C, long V, float eps, float* st, float* __restrict__ stats_out_b) {
    for (int c = threadIdx.x; c < C; c += IN_FIN_NT) {
        const double mu = sums[c] / (double)V;
        double var = sums[C + c] / (double)V - mu * mu;
        if (var < 0.0) var = 0.0;
        const float m = (float)mu, r = (float)(1.0 / sqrt(var + (double)eps));
        st[2 * c] = m; st[2 * c + 1] = r;
        if (blockIdx.x == 0 && stats_out_b) { stats_out_b[2 * c] = m; stats_out_b[2 * c + 1] = r; }
    }
}

// in_apply_hoist_kernel with the statistics finalize in its prologue (forward): pa / pb = partial rows of x / x2
template <class T, bool DUAL, int U>
__global__ void __launch_bounds__(IN_FIN_NT)
in_apply_fin_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ pa, int rows_a, const T* __restrict__ x2, long ldx2,
                    const float* __restrict__ pb, int rows_b, float* __restrict__ stats_a, float* __restrict__ stats_b, float eps,
                    T* __restrict__ y, long ldy, long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    __shared__ double red[IN_FIN_NT];
    __shared__ double sums[4 * 128];
    __shared__ float sta[2 * 128], stb[DUAL ? 2 * 128 : 2];
    const int cvn = C / W, nphase = IN_FIN_NT / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = DUAL ? x2 + ((long)b * V) * ldx2 + W * cv : nullptr;
    T* py = y + ((long)b * V) * ldy + W * cv;
    in_fin_sums(pa + (long)b * rows_a * 2 * C, rows_a, DUAL ? pb + (long)b * rows_b * 2 * C : nullptr, rows_b, 2 * C, red, sums);
    in_fin_stats(sums, C, V, eps, sta, stats_a + (long)b * C * 2);
    if (DUAL) in_fin_stats(sums + 2 * C, C, V, eps, stb, stats_b + (long)b * C * 2);
    __syncthreads();
    float a1[W], o1[W], a2[DUAL ? W : 1];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = sta[2 * (W * cv + e) + 1]; o1[e] = -sta[2 * (W * cv + e)] * a1[e];
        if (DUAL) { a2[e] = stb[2 * (W * cv + e) + 1]; o1[e] -= stb[2 * (W * cv + e)] * a2[e]; }
    }
    const float slope = lrelu ? 0.01f : 1.f;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rt[U], rt2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rt[u] = *(const u32x4*)(px + vc * ldx);
            rt2[u] = DUAL ? *(const u32x4*)(px2 + vc * ldx2) : rt[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t[W], t2[W], o[W];
            Io<T>::unpack(rt[u], t);
            if (DUAL) Io<T>::unpack(rt2[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float n = fmaf(t[e], a1[e], o1[e]);
                if (DUAL) n = fmaf(t2[e], a2[e], n);
                o[e] = n > 0.f ? n : slope * n;
            }
            if (live[u]) Io<T>::stw(py + (v + (long)u * nphase) * ldy, o);
        }
    }
}

// in_bwd_apply_hoist_kernel with in_bwd_final_kernel folded into its prologue: part = [B][nrows][NSP][C] partial sums of
// (g, g n1[, g n2]) from in_bwd_reduce_kernel (NSP 3) or from the epilogue of the conv that produced dy (NSP 2, single form)
template <class T, bool DUAL, int U>
__global__ void __launch_bounds__(IN_FIN_NT)
in_bwd_apply_fin_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                        const T* __restrict__ x2, long ldx2, const float* __restrict__ sb, const float* __restrict__ part, int nrows, int NSP,
                        T* __restrict__ dx, long lddx, T* __restrict__ dx2, long lddx2, long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    __shared__ double red[IN_FIN_NT];
    __shared__ double sums[IN_FIN_MAXCOL];
    const int cvn = C / W, nphase = IN_FIN_NT / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    in_fin_sums(part + (long)b * nrows * NSP * C, nrows, nullptr, 0, NSP * C, red, sums);
    const float* s1 = sa + ((long)b * C + W * cv) * 2;
    const float* s2 = DUAL ? sb + ((long)b * C + W * cv) * 2 : nullptr;
    float a1[W], o1[W], k0[W], k1[W], a2[DUAL ? W : 1], o2[DUAL ? W : 1], q0[DUAL ? W : 1], q2[DUAL ? W : 1];
    const double iv = 1.0 / (double)V;
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const int c = W * cv + e;
        const float m0 = (float)(sums[c] * iv), m1 = (float)(sums[C + c] * iv);
        a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
        k0[e] = a1[e] * m0; k1[e] = a1[e] * m1;
        if (DUAL) {
            const float m2 = (float)(sums[2 * C + c] * iv);
            a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e];
            q0[e] = a2[e] * m0; q2[e] = a2[e] * m2;
        }
    }
    const float slope = lrelu ? 0.01f : 1.f;
    const T* pg = dy + ((long)b * V) * lddy + W * cv;
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = DUAL ? x2 + ((long)b * V) * ldx2 + W * cv : nullptr;
    T* pd = dx + ((long)b * V) * lddx + W * cv;
    T* pd2 = DUAL ? dx2 + ((long)b * V) * lddx2 + W * cv : nullptr;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rg[U], rt[U], rt2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rg[u] = *(const u32x4*)(pg + vc * lddy);
            rt[u] = *(const u32x4*)(px + vc * ldx);
            rt2[u] = DUAL ? *(const u32x4*)(px2 + vc * ldx2) : rt[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[W], t[W], t2[W], o[W], p[W];
            Io<T>::unpack(rg[u], g);
            Io<T>::unpack(rt[u], t);
            if (DUAL) Io<T>::unpack(rt2[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float n1 = fmaf(t[e], a1[e], o1[e]);
                const float n2 = DUAL ? fmaf(t2[e], a2[e], o2[e]) : 0.f;
                const float ge = (n1 + n2) > 0.f ? g[e] : slope * g[e];
                o[e] = fmaf(a1[e], ge, -k0[e]) - k1[e] * n1;
                if (DUAL) p[e] = fmaf(a2[e], ge, -q0[e]) - q2[e] * n2;
            }
            if (live[u]) {
                Io<T>::stw(pd + (v + (long)u * nphase) * lddx, o);
                if (DUAL) Io<T>::stw(pd2 + (v + (long)u * nphase) * lddx2, p);
            }
        }
    }
}

// ---- the residual block on the IMAGE (encoder1: UnetrBasicBlock(in_channels, feature_size), unetr.py:90-98): its 1x1x1 branch
// c3[v][c] = sum_ci w3[c][ci] * img[v][ci] has <= 4 input channels, so every pass that reads c3 (32 B per voxel in bf16 storage, 64
// in fp32) can form it from the image (4 - 16 B per voxel) instead, and c3 / its gradient are never stored: per step that is six
// full-resolution tensors (c3 written by the conv, read by the block-end apply, the backward reduction and the backward apply;
// dc3 written by the backward apply and read by the weight gradient) that do not move.  The value is formed exactly as the conv
// kernel forms it before its store: operands rounded to the mode's operand type, fp32 products, result rounded to the storage type
// (bit-identical to the stored tensor for one input channel).  dw3 = sum_v dc3[v][c] * img[v][ci] comes out of the backward apply
// as per-block partial rows.
template <class T, int W, int CIN>
struct ImgBranch {
    float w[W][CIN];
    const float* img;
    int cin;
    static __device__ __forceinline__ float rnd(float v) {
        if constexpr (Io<T>::B16) { const __bf16 h = (__bf16)v; return (float)h; }
        return v;
    }
    // (CIN = 1 or 4 at compile time; a 2- or 3-channel image runs the 4-channel instance with zero weights for the missing ones)
    __device__ __forceinline__ void init(const float* __restrict__ w3, int cin_, int c0, const float* __restrict__ img_b) {
        img = img_b; cin = cin_;
#pragma unroll
        for (int e = 0; e < W; ++e)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) w[e][ci] = ci < cin_ ? rnd(w3[(long)(c0 + e) * cin_ + ci]) : 0.f;
    }
    // the voxel's image values, already rounded to the operand type
    __device__ __forceinline__ void load(long v, float (&xi)[CIN]) const {
        if constexpr (CIN == 1) xi[0] = img[v];
        else {
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) xi[ci] = ci < cin ? img[v * cin + ci] : 0.f;
        }
    }
    __device__ __forceinline__ void value(float (&xi)[CIN], float (&t2)[W]) const {
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) xi[ci] = rnd(xi[ci]);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float a = w[e][0] * xi[0];
#pragma unroll
            for (int ci = 1; ci < CIN; ++ci) a = fmaf(w[e][ci], xi[ci], a);
            t2[e] = rnd(a);
        }
    }
};

template <class T, int U, int CIN>
__global__ void __launch_bounds__(IN_FIN_NT)
in_apply_fin_img_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ pa, int rows_a, const float* __restrict__ img, int cin,
                        const float* __restrict__ w3, const float* __restrict__ pb, int rows_b, float* __restrict__ stats_a,
                        float* __restrict__ stats_b, float eps, T* __restrict__ y, long ldy, long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    __shared__ double red[IN_FIN_NT];
    __shared__ double sums[4 * 128];
    __shared__ float sta[2 * 128], stb[2 * 128];
    const int cvn = C / W, nphase = IN_FIN_NT / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const T* px = x + ((long)b * V) * ldx + W * cv;
    T* py = y + ((long)b * V) * ldy + W * cv;
    ImgBranch<T, W, CIN> br;
    br.init(w3, cin, W * cv, img + (long)b * V * cin);
    in_fin_sums(pa + (long)b * rows_a * 2 * C, rows_a, pb + (long)b * rows_b * 2 * C, rows_b, 2 * C, red, sums);
    in_fin_stats(sums, C, V, eps, sta, stats_a + (long)b * C * 2);
    in_fin_stats(sums + 2 * C, C, V, eps, stb, stats_b + (long)b * C * 2);
    __syncthreads();
    float a1[W], o1[W], a2[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = sta[2 * (W * cv + e) + 1]; o1[e] = -sta[2 * (W * cv + e)] * a1[e];
        a2[e] = stb[2 * (W * cv + e) + 1]; o1[e] -= stb[2 * (W * cv + e)] * a2[e];
    }
    const float slope = lrelu ? 0.01f : 1.f;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rt[U];
        float xi[U][CIN];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rt[u] = *(const u32x4*)(px + vc * ldx);
            br.load(vc, xi[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t[W], t2[W], o[W];
            Io<T>::unpack(rt[u], t);
            br.value(xi[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float n = fmaf(t[e], a1[e], o1[e]);
                n = fmaf(t2[e], a2[e], n);
                o[e] = n > 0.f ? n : slope * n;
            }
            if (live[u]) Io<T>::stw(py + (v + (long)u * nphase) * ldy, o);
        }
    }
}

// backward stage 1 of the same block end (in_bwd_reduce_kernel<T, true, U> with the second branch formed from the image)
template <class T, int U, int CIN>
__global__ void __launch_bounds__(256)
in_bwd_reduce_img_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                         const float* __restrict__ img, int cin, const float* __restrict__ w3, const float* __restrict__ sb, long V, long vpb,
                         int C, int lrelu, float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = Io<T>::W;
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    float acc[3][W];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int e = 0; e < W; ++e) acc[s][e] = 0.f;
    if (ph < nphase) {
        const float* s1 = sa + ((long)b * C + W * cv) * 2;
        const float* s2 = sb + ((long)b * C + W * cv) * 2;
        float a1[W], o1[W], a2[W], o2[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
            a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
            a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e];
        }
        ImgBranch<T, W, CIN> br;
        br.init(w3, cin, W * cv, img + (long)b * V * cin);
        const float slope = lrelu ? 0.01f : 1.f;
        const T* pg = dy + ((long)b * V) * lddy + W * cv;
        const T* px = x + ((long)b * V) * ldx + W * cv;
        for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
            u32x4 rg[U], rt[U];
            float xi[U][CIN];
            bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long vv = v + (long)u * nphase;
                live[u] = vv < v1;
                const long vc = live[u] ? vv : v;
                rg[u] = *(const u32x4*)(pg + vc * lddy);
                rt[u] = *(const u32x4*)(px + vc * ldx);
                br.load(vc, xi[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float g[W], t[W], t2[W];
                Io<T>::unpack(rg[u], g);
                Io<T>::unpack(rt[u], t);
                br.value(xi[u], t2);
                if (live[u]) {
#pragma unroll
                    for (int e = 0; e < W; ++e) {
                        const float n1 = fmaf(t[e], a1[e], o1[e]);
                        const float n2 = fmaf(t2[e], a2[e], o2[e]);
                        const float ge = (n1 + n2) > 0.f ? g[e] : slope * g[e];
                        acc[0][e] += ge;
                        acc[1][e] = fmaf(ge, n1, acc[1][e]);
                        acc[2][e] = fmaf(ge, n2, acc[2][e]);
                    }
                }
            }
        }
    }
    in_block_reduce<3, W>(acc, cvn, nphase, lds, part + ((long)b * gridDim.x + blockIdx.x) * 3 * C, C);
}

// backward stage 2: dx of the 3x3x3 branch; the 1x1x1 branch's gradient dc3 is formed per voxel, multiplied with the image and
// summed: dw3_part[(b * gridDim.x + blockIdx.x)][C][cin] (rows for unetr_reduce_rows_grouped); dc3 itself is never stored
template <class T, int U, int CIN>
__global__ void __launch_bounds__(IN_FIN_NT)
in_bwd_apply_fin_img_kernel(const T* __restrict__ dy, long lddy, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                            const float* __restrict__ img, int cin, const float* __restrict__ w3, const float* __restrict__ sb,
                            const float* __restrict__ part, int nrows, T* __restrict__ dx, long lddx, float* __restrict__ dw3_part,
                            long V, long vpb, int C, int lrelu) {
    constexpr int W = Io<T>::W;
    __shared__ double red[IN_FIN_NT];
    __shared__ double sums[IN_FIN_MAXCOL];
    __shared__ float wsum[(IN_FIN_NT / 64) * 128 * CIN];        // [wave][C][CIN]
    const int cvn = C / W, nphase = IN_FIN_NT / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    in_fin_sums(part + (long)b * nrows * 3 * C, nrows, nullptr, 0, 3 * C, red, sums);
    const float* s1 = sa + ((long)b * C + W * cv) * 2;
    const float* s2 = sb + ((long)b * C + W * cv) * 2;
    float a1[W], o1[W], k0[W], k1[W], a2[W], o2[W], q0[W], q2[W];
    const double iv = 1.0 / (double)V;
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const int c = W * cv + e;
        const float m0 = (float)(sums[c] * iv), m1 = (float)(sums[C + c] * iv), m2 = (float)(sums[2 * C + c] * iv);
        a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
        k0[e] = a1[e] * m0; k1[e] = a1[e] * m1;
        a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e];
        q0[e] = a2[e] * m0; q2[e] = a2[e] * m2;
    }
    ImgBranch<T, W, CIN> br;
    br.init(w3, cin, W * cv, img + (long)b * V * cin);
    float dwa[W][CIN];
#pragma unroll
    for (int e = 0; e < W; ++e)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) dwa[e][ci] = 0.f;
    const float slope = lrelu ? 0.01f : 1.f;
    const T* pg = dy + ((long)b * V) * lddy + W * cv;
    const T* px = x + ((long)b * V) * ldx + W * cv;
    T* pd = dx + ((long)b * V) * lddx + W * cv;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rg[U], rt[U];
        float xi[U][CIN];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rg[u] = *(const u32x4*)(pg + vc * lddy);
            rt[u] = *(const u32x4*)(px + vc * ldx);
            br.load(vc, xi[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float g[W], t[W], t2[W], o[W];
            Io<T>::unpack(rg[u], g);
            Io<T>::unpack(rt[u], t);
            br.value(xi[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float n1 = fmaf(t[e], a1[e], o1[e]);
                const float n2 = fmaf(t2[e], a2[e], o2[e]);
                const float ge = (n1 + n2) > 0.f ? g[e] : slope * g[e];
                o[e] = fmaf(a1[e], ge, -k0[e]) - k1[e] * n1;
                // the gradient of the 1x1x1 branch as the weight-gradient kernel would have read it back: rounded to the storage type
                float p = ImgBranch<T, W, CIN>::rnd(fmaf(a2[e], ge, -q0[e]) - q2[e] * n2);
                p = live[u] ? p : 0.f;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) dwa[e][ci] = fmaf(p, xi[u][ci], dwa[e][ci]);      // (xi was rounded by value())
            }
            if (live[u]) Io<T>::stw(pd + (v + (long)u * nphase) * lddx, o);
        }
    }
    // block sum of dwa over the voxel phases: lanes of a wave that share cv (cvn <= 64 divides 64: lane % cvn), then the waves in
    // order through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int e = 0; e < W; ++e)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            float t = dwa[e][ci];
            for (int o_ = cvn; o_ < 64; o_ <<= 1) t += __shfl_xor(t, o_, 64);
            dwa[e][ci] = t;
        }
    if (lane < cvn) {
#pragma unroll
        for (int e = 0; e < W; ++e)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) wsum[(wave * 128 + W * lane + e) * CIN + ci] = dwa[e][ci];
    }
    __syncthreads();
    float* prow = dw3_part + ((long)b * gridDim.x + blockIdx.x) * C * cin;
    for (int i = threadIdx.x; i < C * cin; i += IN_FIN_NT) {
        const int c = i / cin, ci = i - c * cin;
        float t = 0.f;
        for (int wv_ = 0; wv_ < IN_FIN_NT / 64; ++wv_) t += wsum[(wv_ * 128 + c) * CIN + ci];
        prow[i] = t;
    }
}

// ------------------------------------------------------------------------------------ layout moves
// per batch: src [R, Ccols] (pitch lds_) -> dst [Ccols, R] (pitch ldd)
template <class TS, class TD>
__global__ void __launch_bounds__(256)
transpose_kernel(const TS* __restrict__ src, long ld_s, long bs_s, TD* __restrict__ dst, long ld_d, long bs_d,
                 long R, long Cc, int accumulate) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long r0 = (long)blockIdx.x * 32, c0 = (long)blockIdx.y * 32;
    const TS* s = src + (long)blockIdx.z * bs_s;
    TD* d = dst + (long)blockIdx.z * bs_d;
    for (int j = ty; j < 32; j += 8) {
        long r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < Cc) ? Io<TS>::ld1(s + r * ld_s + c) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        long c = c0 + j, r = r0 + tx;
        if (r < R && c < Cc) {
            float v = tile[tx][j];
            if (accumulate) v += Io<TD>::ld1(d + c * ld_d + r);
            Io<TD>::st1(d + c * ld_d + r, v);
        }
    }
}

// out (fp32) and / or out16 (bf16: the GEMM operand of the bf16-storage path -- no fp32 patches + cast pass in between)
// VEC: one channel, P and W multiples of 4, 16-byte aligned tensors, < 2^31 elements: four consecutive values of a patch row per
// thread (one 16-byte load, 16 / 8-byte stores) with 32-bit index arithmetic -- element by element the kernel spent its time in
// ~10 64-bit divisions per value
template <bool VEC>
__global__ void patch_gather_kernel(const float* __restrict__ x, float* __restrict__ out, uint16_t* __restrict__ out16, int B, int C, int D, int H, int W, int P) {
    const int gd = D / P, gh = H / P, gw = W / P;
    const long pd = (long)P * P * P * C;
    const long total = (long)B * gd * gh * gw * pd;
    if constexpr (VEC) {
        const unsigned total4 = (unsigned)(total >> 2), upd = (unsigned)pd, P4 = (unsigned)P >> 2;
        for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
            const unsigned i = i4 << 2, f = i % upd, tok = i / upd;
            unsigned t = f >> 2;                                   // (C == 1: f = (p1 * P + p2) * P + p3, p3 a multiple of 4)
            const unsigned p3 = (t % P4) << 2; t /= P4; const unsigned p2 = t % (unsigned)P, p1 = t / (unsigned)P;
            unsigned u = tok;
            const unsigned w3 = u % (unsigned)gw; u /= (unsigned)gw; const unsigned w2 = u % (unsigned)gh; u /= (unsigned)gh;
            const unsigned w1 = u % (unsigned)gd, b = u / (unsigned)gd;
            const f32x4 v = *(const f32x4*)(x + (((long)b * D + w1 * P + p1) * H + w2 * P + p2) * W + w3 * P + p3);
            if (out) *(f32x4*)(out + i) = v;
            if (out16) *(bf16x4*)(out16 + i) = __builtin_convertvector(v, bf16x4);
        }
        return;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long f = i % pd; long tok = i / pd;
        int c = (int)(f % C); long t = f / C; int p3 = (int)(t % P); t /= P; int p2 = (int)(t % P); int p1 = (int)(t / P);
        int w3 = (int)(tok % gw); t = tok / gw; int w2 = (int)(t % gh); t /= gh; int w1 = (int)(t % gd); int b = (int)(t / gd);
        const float v = x[((((long)b * C + c) * D + w1 * P + p1) * H + w2 * P + p2) * W + w3 * P + p3];
        if (out) out[i] = v;
        if (out16) { __bf16 h = (__bf16)v; out16[i] = __builtin_bit_cast(uint16_t, h); }
    }
}

// y[i] += inc[i] (i < n): the per-parameter AdamW step counters advanced by the 0/1 "has a gradient" mask (one tiny launch
// that is part of the captured step; a torch elementwise add did this before)
__global__ void counter_add_kernel(float* __restrict__ y, const float* __restrict__ inc, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += inc[i];
}

template <class T>
__global__ void add_rows_kernel(T* __restrict__ y, long ldy, const T* __restrict__ a, long lda, long rows, int cols, int accumulate) {
    constexpr int W = Io<T>::W;
    const int cvn = cols / W;
    const long total = rows * cvn;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cv = (int)(i % cvn); long r = i / cvn;
        T* yp = y + r * ldy + W * cv;
        float v[W];
        Io<T>::ldw(a + r * lda + W * cv, v);
        if (accumulate) {
            float o[W];
            Io<T>::ldw(yp, o);
#pragma unroll
            for (int e = 0; e < W; ++e) v[e] += o[e];
        }
        Io<T>::stw(yp, v);
    }
}

// ------------------------------------------------------------------------- out conv (1x1x1 + bias)
constexpr int OC_MAXCO = 16, OC_MAXCI = 64;
template <class T>
__global__ void __launch_bounds__(256)
outconv_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ logits, int B, long V, int Cin, int Cout) {
    __shared__ float sw[OC_MAXCO * OC_MAXCI + OC_MAXCO];
    for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[OC_MAXCO * OC_MAXCI + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const long total = (long)B * V;
    for (long vox = (long)blockIdx.x * blockDim.x + threadIdx.x; vox < total; vox += (long)gridDim.x * blockDim.x) {
        int b = (int)(vox / V); long v = vox - (long)b * V;
        float acc[OC_MAXCO];
#pragma unroll
        for (int co = 0; co < OC_MAXCO; ++co) acc[co] = sw[OC_MAXCO * OC_MAXCI + co];
        for (int c4 = 0; c4 < Cin; c4 += 4) {
            f32x4 t = Io<T>::ld4(x + vox * ldx + c4);
#pragma unroll
            for (int co = 0; co < OC_MAXCO; ++co)
                if (co < Cout) {
                    const float* ww = sw + co * Cin + c4;
                    acc[co] += t[0] * ww[0] + t[1] * ww[1] + t[2] * ww[2] + t[3] * ww[3];
                }
        }
#pragma unroll
        for (int co = 0; co < OC_MAXCO; ++co)
            if (co < Cout) logits[((long)b * Cout + co) * V + v] = acc[co];
    }
}

// dx[vox, ci] = sum_co dl[b,co,v] * w[co,ci];  per-block partials part[blk][Cout (+ Cout*Cin)]: the bias gradient and, when
// WG (Cout * Cin <= 64: 4 classes x 16 features), the weight gradient dw[co,ci] = sum_vox dl[co,vox] * x[vox,ci] from the
// SAME pass over dl (the x row is one extra 64-byte read per voxel).  Before, dw took two exact-fp32 GEMMs of shape
// [4 x 884736] x [884736 x 16] on 16x16 tiles with a deep split-K reduce: 224 us per step for 64 numbers.
template <bool WG, class T>
__global__ void __launch_bounds__(256)
outconv_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ w, const T* __restrict__ x, long ldx,
                   T* __restrict__ dx, long lddx, float* __restrict__ part, int B, long V, int Cin, int Cout) {
    __shared__ float sw[OC_MAXCO * OC_MAXCI];
    __shared__ float red[4][OC_MAXCO + 64];
    for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) sw[i] = w[i];
    __syncthreads();
    float bsum[OC_MAXCO];
#pragma unroll
    for (int co = 0; co < OC_MAXCO; ++co) bsum[co] = 0.f;
    float wsum[WG ? 64 : 1];
#pragma unroll
    for (int i = 0; i < (WG ? 64 : 1); ++i) wsum[i] = 0.f;
    const long total = (long)B * V;
    for (long vox = (long)blockIdx.x * blockDim.x + threadIdx.x; vox < total; vox += (long)gridDim.x * blockDim.x) {
        int b = (int)(vox / V); long v = vox - (long)b * V;
        float g[OC_MAXCO];
#pragma unroll
        for (int co = 0; co < OC_MAXCO; ++co) {
            g[co] = co < Cout ? dl[((long)b * Cout + co) * V + v] : 0.f;
            bsum[co] += g[co];
        }
        for (int c4 = 0; c4 < Cin; c4 += 4) {
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int co = 0; co < OC_MAXCO; ++co)
                if (co < Cout) {
                    const float* ww = sw + co * Cin + c4;
                    o[0] += g[co] * ww[0]; o[1] += g[co] * ww[1]; o[2] += g[co] * ww[2]; o[3] += g[co] * ww[3];
                }
            Io<T>::st4(dx + vox * lddx + c4, o);
        }
        if constexpr (WG) {
            // (WG: Cout <= 4, Cin <= 16, Cin % 4 == 0 -- checked by the launcher; slot co*16 + ci)
#pragma unroll
            for (int c4 = 0; c4 < 16; c4 += 4) {
                if (c4 < Cin) {
                    const f32x4 xv = Io<T>::ld4(x + vox * ldx + c4);
#pragma unroll
                    for (int co = 0; co < 4; ++co)
#pragma unroll
                        for (int e = 0; e < 4; ++e) wsum[co * 16 + c4 + e] += g[co] * xv[e];
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < OC_MAXCO; ++co) {
        float s = wave_sum(bsum[co]);
        if (lane == 0) red[wave][co] = s;
    }
    if constexpr (WG) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            float s = wave_sum(wsum[i]);
            if (lane == 0) red[wave][OC_MAXCO + i] = s;
        }
    }
    __syncthreads();
    const int np = Cout + (WG ? Cout * Cin : 0);
    if ((int)threadIdx.x < np) {
        int src = threadIdx.x;
        if (WG && (int)threadIdx.x >= Cout) { const int k = threadIdx.x - Cout, co = k / Cin, ci = k - co * Cin; src = OC_MAXCO + co * 16 + ci; }
        part[(long)blockIdx.x * np + threadIdx.x] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// The small-head form (Cout <= 4, Cin in {8, 16} for bf16 maps / {4, 8, 16} for fp32: the launcher checks): one thread = one
// 16-byte piece of a voxel's channel row, so a wave reads x and writes dx as contiguous 1 KB runs (the one-thread-per-voxel
// form above issued 8-byte pieces 32 bytes apart and ran at 1.8 TB/s).  Same partial layout as above.
template <class T>
__global__ void __launch_bounds__(256)
outconv_bwd_small_kernel(const float* __restrict__ dl, const float* __restrict__ w, const T* __restrict__ x, long ldx,
                         T* __restrict__ dx, long lddx, float* __restrict__ part, int B, long V, int Cin, int Cout) {
    constexpr int W = Io<T>::W;
    __shared__ float red[4][4 + 64];
    const int cvn = Cin / W;                       // 1, 2 or 4 pieces per voxel
    const int cv = threadIdx.x % cvn;
    float wr[4][W];                                // this thread's columns of the weight
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int e = 0; e < W; ++e) wr[co][e] = co < Cout ? w[co * Cin + W * cv + e] : 0.f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    float wsum[4][W];
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int e = 0; e < W; ++e) wsum[co][e] = 0.f;
    const long total = (long)B * V * cvn;
    // (one voxel piece per iteration: batching two or four pieces' loads was tried -- 122 / 144 VGPRs instead of ~100, and the
    // resident waves it costs are worth more to this streaming loop than the loads it puts in flight)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long vox = i / cvn;
        const int b = (int)(vox / V); const long v = vox - (long)b * V;
        float xv[W];
        Io<T>::ldw(x + vox * ldx + W * cv, xv);
        float g[4];
#pragma unroll
        for (int co = 0; co < 4; ++co) g[co] = co < Cout ? dl[((long)b * Cout + co) * V + v] : 0.f;
        float o[W];
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = g[0] * wr[0][e] + g[1] * wr[1][e] + g[2] * wr[2][e] + g[3] * wr[3][e];
        Io<T>::stw(dx + vox * lddx + W * cv, o);
#pragma unroll
        for (int co = 0; co < 4; ++co) {
            bsum[co] += cv == 0 ? g[co] : 0.f;
#pragma unroll
            for (int e = 0; e < W; ++e) wsum[co][e] = fmaf(g[co], xv[e], wsum[co][e]);
        }
    }
    // lanes with the same cv (lane % cvn: cvn divides 64 and the block size) hold the same channels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        float s = wave_sum(bsum[co]);
        if (lane == 0) red[wave][co] = s;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = wsum[co][e];
            for (int o = 32; o >= cvn; o >>= 1) t += __shfl_xor(t, o, 64);
            if (lane < cvn) red[wave][4 + co * 16 + W * lane + e] = t;
        }
    }
    __syncthreads();
    const int np = Cout + Cout * Cin;
    if ((int)threadIdx.x < np) {
        int src = threadIdx.x;
        if ((int)threadIdx.x >= Cout) { const int k = threadIdx.x - Cout, co = k / Cin, ci = k - co * Cin; src = 4 + co * 16 + ci; }
        part[(long)blockIdx.x * np + threadIdx.x] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// Larger heads (Cout > 4 or Cin > 16; the reference's default is 14 BTCV classes, unetr_segmentation_3d.py:303): weight-gradient
// partials part[blk][Cout*Cin], dw[co,ci] = sum_vox dl[b,co,v] * x[vox,ci], on fp32- or bf16-stored feature maps.  A workgroup
// stages 128 voxels of dl (all classes) and of x (all channels) in LDS as fp32; thread t owns the (co, ci) pairs t, t+256, ...
// (<= 4 of them: Cout <= 16, Cin <= 64), so the dl value is a broadcast read and the x row a conflict-free one.
constexpr int OCW_VT = 128;
template <class T>
__global__ void __launch_bounds__(256)
outconv_wgrad_generic_kernel(const float* __restrict__ dl, const T* __restrict__ x, long ldx, float* __restrict__ part,
                             int B, long V, int Cin, int Cout) {
    __shared__ float sdl[OC_MAXCO][OCW_VT];
    __shared__ float sx[OCW_VT][OC_MAXCI + 1];
    const int np = Cout * Cin;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int pco[4], pci[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = threadIdx.x + k * 256;
        pco[k] = p < np ? p / Cin : 0;
        pci[k] = p < np ? p - pco[k] * Cin : 0;
    }
    const long total = (long)B * V;
    const long ntile = (total + OCW_VT - 1) / OCW_VT;
    for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
        const long v0 = t * OCW_VT;
        for (int i = threadIdx.x; i < Cout * OCW_VT; i += 256) {
            const int co = i / OCW_VT, j = i - co * OCW_VT;
            const long vox = v0 + j;
            float g = 0.f;
            if (vox < total) { const int b = (int)(vox / V); const long v = vox - (long)b * V; g = dl[((long)b * Cout + co) * V + v]; }
            sdl[co][j] = g;
        }
        const int c4n = Cin >> 2;
        for (int i = threadIdx.x; i < c4n * OCW_VT; i += 256) {
            const int j = i / c4n, c4 = (i - j * c4n) * 4;
            const long vox = v0 + j;
            f32x4 xv = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (vox < total) xv = Io<T>::ld4(x + vox * ldx + c4);
            sx[j][c4] = xv[0]; sx[j][c4 + 1] = xv[1]; sx[j][c4 + 2] = xv[2]; sx[j][c4 + 3] = xv[3];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((int)threadIdx.x + k * 256 < np) {
                float s = acc[k];
                for (int j = 0; j < OCW_VT; ++j) s = fmaf(sdl[pco[k]][j], sx[j][pci[k]], s);
                acc[k] = s;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if ((int)threadIdx.x + k * 256 < np) part[(long)blockIdx.x * np + threadIdx.x + k * 256] = acc[k];
}

// ---- the LAST residual block's end folded into the out conv (decoder2 -> UnetOutBlock, unetr.py:165-175,206-207) ------------------
// out = lrelu(norm(c2) + norm(c3)) is consumed by the 1x1x1 out conv only, so it is never stored: the forward kernel forms it per
// voxel from c2 / c3 (statistics finalized in its prologue from the convs' partial rows, as in_apply_fin_kernel) and writes the
// logits; the backward kernel forms it again (for the out conv's weight gradient and the lrelu mask), computes dout = W^T dl,
// stores it, and accumulates the InstanceNorm backward sums (g, g n2, g n3 with g = dout lrelu'(.)) on the way -- the separate
// reduction pass over (dout, c2, c3) disappears.  `out` and `dout` are rounded to the storage type T where the unfused sequence
// stored them, so both sequences see the same values.  Small heads only (Cout <= 4, C = 1 / 2 / 4 pieces of W channels).
template <class T> __device__ __forceinline__ float store_round(float v);
template <> __device__ __forceinline__ float store_round<float>(float v) { return v; }
template <> __device__ __forceinline__ float store_round<uint16_t>(float v) { const __bf16 h = (__bf16)v; return (float)h; }

template <class T, int U>
__global__ void __launch_bounds__(IN_FIN_NT)
outconv_in_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ pa, int rows_a, const T* __restrict__ x2, long ldx2,
                      const float* __restrict__ pb, int rows_b, float* __restrict__ stats_a, float* __restrict__ stats_b, float eps,
                      const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ logits,
                      long V, long vpb, int C, int Cout) {
    constexpr int W = Io<T>::W;
    __shared__ double red[IN_FIN_NT];
    __shared__ double sums[4 * 128];
    __shared__ float sta[2 * 128], stb[2 * 128];
    const int cvn = C / W, nphase = IN_FIN_NT / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = x2 + ((long)b * V) * ldx2 + W * cv;
    in_fin_sums(pa + (long)b * rows_a * 2 * C, rows_a, pb + (long)b * rows_b * 2 * C, rows_b, 2 * C, red, sums);
    in_fin_stats(sums, C, V, eps, sta, stats_a + (long)b * C * 2);
    in_fin_stats(sums + 2 * C, C, V, eps, stb, stats_b + (long)b * C * 2);
    __syncthreads();
    float a1[W], o1[W], a2[W], wr[4][W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = sta[2 * (W * cv + e) + 1]; o1[e] = -sta[2 * (W * cv + e)] * a1[e];
        a2[e] = stb[2 * (W * cv + e) + 1]; o1[e] -= stb[2 * (W * cv + e)] * a2[e];
#pragma unroll
        for (int co = 0; co < 4; ++co) wr[co][e] = co < Cout ? w[co * C + W * cv + e] : 0.f;
    }
    float bz[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) bz[co] = (bias && co < Cout) ? bias[co] : 0.f;
    float* pl = logits + (long)b * Cout * V;
    for (long v = v0 + ph; v < v1; v += (long)U * nphase) {
        u32x4 rt[U], rt2[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long vv = v + (long)u * nphase;
            live[u] = vv < v1;
            const long vc = live[u] ? vv : v;
            rt[u] = *(const u32x4*)(px + vc * ldx);
            rt2[u] = *(const u32x4*)(px2 + vc * ldx2);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t[W], t2[W], d[4] = {0.f, 0.f, 0.f, 0.f};
            Io<T>::unpack(rt[u], t);
            Io<T>::unpack(rt2[u], t2);
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float n = fmaf(t[e], a1[e], o1[e]);
                n = fmaf(t2[e], a2[e], n);
                const float o = store_round<T>(n > 0.f ? n : 0.01f * n);
#pragma unroll
                for (int co = 0; co < 4; ++co) d[co] = fmaf(o, wr[co][e], d[co]);
            }
            // the cvn pieces of a voxel sit in adjacent lanes
#pragma unroll
            for (int co = 0; co < 4; ++co)
                for (int o = 1; o < cvn; o <<= 1) d[co] += __shfl_xor(d[co], o, 64);
            if (live[u] && cv == 0) {
                const long vv = v + (long)u * nphase;
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < Cout) pl[(long)co * V + vv] = d[co] + bz[co];
            }
        }
    }
}

// part_in: this block's row [3][C] of the InstanceNorm backward sums (rows [B][gridDim.x]); part_oc: its row of the out conv's
// bias / weight gradient partials [Cout + Cout * C] (rows [B * gridDim.x])
template <class T>
__global__ void __launch_bounds__(256)
outconv_in_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ w, const T* __restrict__ x, long ldx, const float* __restrict__ sa,
                      const T* __restrict__ x2, long ldx2, const float* __restrict__ sb, T* __restrict__ dx, long lddx,
                      float* __restrict__ part_in, float* __restrict__ part_oc, long V, long vpb, int C, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W = Io<T>::W;
    __shared__ float red[4][4 + 64];
    const int cvn = C / W, nphase = 256 / cvn;
    const int cv = threadIdx.x % cvn, ph = threadIdx.x / cvn;
    const int b = blockIdx.y;
    const long v0 = (long)blockIdx.x * vpb, v1 = std::min<long>(V, v0 + vpb);
    const float* s1 = sa + ((long)b * C + W * cv) * 2;
    const float* s2 = sb + ((long)b * C + W * cv) * 2;
    float a1[W], o1[W], a2[W], o2[W], wr[4][W], wsum[4][W], acc[3][W];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < W; ++e) {
        a1[e] = s1[2 * e + 1]; o1[e] = -s1[2 * e] * a1[e];
        a2[e] = s2[2 * e + 1]; o2[e] = -s2[2 * e] * a2[e];
        acc[0][e] = acc[1][e] = acc[2][e] = 0.f;
#pragma unroll
        for (int co = 0; co < 4; ++co) { wr[co][e] = co < Cout ? w[co * C + W * cv + e] : 0.f; wsum[co][e] = 0.f; }
    }
    const T* px = x + ((long)b * V) * ldx + W * cv;
    const T* px2 = x2 + ((long)b * V) * ldx2 + W * cv;
    T* pd = dx + ((long)b * V) * lddx + W * cv;
    const float* pg = dl + (long)b * Cout * V;
#pragma unroll 1
    for (long v = v0 + ph; v < v1; v += nphase) {
        const u32x4 rt = *(const u32x4*)(px + v * ldx), rt2 = *(const u32x4*)(px2 + v * ldx2);
        float g[4];
#pragma unroll
        for (int co = 0; co < 4; ++co) g[co] = co < Cout ? pg[(long)co * V + v] : 0.f;
        float t[W], t2[W], dq[W];
        Io<T>::unpack(rt, t);
        Io<T>::unpack(rt2, t2);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float n1 = fmaf(t[e], a1[e], o1[e]), n2 = fmaf(t2[e], a2[e], o2[e]);
            const float sn = n1 + n2;
            const float o = store_round<T>(sn > 0.f ? sn : 0.01f * sn);
            dq[e] = store_round<T>(g[0] * wr[0][e] + g[1] * wr[1][e] + g[2] * wr[2][e] + g[3] * wr[3][e]);
            const float ge = sn > 0.f ? dq[e] : 0.01f * dq[e];
            acc[0][e] += ge;
            acc[1][e] = fmaf(ge, n1, acc[1][e]);
            acc[2][e] = fmaf(ge, n2, acc[2][e]);
#pragma unroll
            for (int co = 0; co < 4; ++co) wsum[co][e] = fmaf(g[co], o, wsum[co][e]);
        }
        Io<T>::stw(pd + v * lddx, dq);
#pragma unroll
        for (int co = 0; co < 4; ++co) bsum[co] += cv == 0 ? g[co] : 0.f;
    }
    in_block_reduce<3, W>(acc, cvn, nphase, lds, part_in + ((long)b * gridDim.x + blockIdx.x) * 3 * C, C);
    // out conv partials: lanes with the same cv (lane % cvn: cvn divides 64 and the block size) hold the same channels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        float sbz = wave_sum(bsum[co]);
        if (lane == 0) red[wave][co] = sbz;
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float tt = wsum[co][e];
            for (int o = 32; o >= cvn; o >>= 1) tt += __shfl_xor(tt, o, 64);
            if (lane < cvn) red[wave][4 + co * 16 + W * lane + e] = tt;
        }
    }
    __syncthreads();
    const int np = Cout + Cout * C;
    if ((int)threadIdx.x < np) {
        int src = threadIdx.x;
        if ((int)threadIdx.x >= Cout) { const int k = threadIdx.x - Cout, co = k / C, ci = k - co * C; src = 4 + co * 16 + ci; }
        part_oc[((long)b * gridDim.x + blockIdx.x) * np + threadIdx.x] = red[0][src] + red[1][src] + red[2][src] + red[3][src];
    }
}

// out0[n] = sum_r part[r][n] for n < n0, out1[n - n0] for the rest (bias gradient, then weight gradient)
__global__ void outconv_final_kernel(const float* __restrict__ part, int R, int N, int n0, float* __restrict__ out0, float* __restrict__ out1) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += part[(long)r * N + n];
    if (n < n0) out0[n] = s; else out1[n - n0] = s;
}

// -------------------------------------------------------------------------------------------- AdamW
// GB16: gradients come as bf16 (the data-parallel communication buffer after the all-reduce); gscale = 1 / world size
template <bool GB16>
__global__ void __launch_bounds__(256)
adamw_kernel(float* __restrict__ p, const void* __restrict__ gsrc, float gscale, float* __restrict__ m, float* __restrict__ v,
             long n4, long n, float lr, float b1, float b2, float eps, float wd, const float* __restrict__ step_dev,
             uint16_t* __restrict__ shadow, uint32_t* __restrict__ words) {
    const AdamWCoef c = adamw_coef(lr, b1, b2, eps, wd, *step_dev);
    auto grad1 = [&](long i) -> float {
        if (GB16) { uint32_t u = (uint32_t)((const uint16_t*)gsrc)[i] << 16; return __builtin_bit_cast(float, u) * gscale; }
        return ((const float*)gsrc)[i] * gscale;
    };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        // streamed once per step: non-temporal accesses keep the 2.8 GB of optimizer traffic from evicting what the next forward reads
        f32x4 pv = __builtin_nontemporal_load((f32x4*)p + i), mv = __builtin_nontemporal_load((f32x4*)m + i), vv = __builtin_nontemporal_load((f32x4*)v + i), gv;
        if (GB16) gv = __builtin_convertvector(__builtin_nontemporal_load((const bf16x4*)gsrc + i), f32x4) * gscale;
        else gv = __builtin_nontemporal_load((const f32x4*)gsrc + i) * gscale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adamw_elem(pe, me, ve, gv[e], c);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        __builtin_nontemporal_store(pv, (f32x4*)p + i); __builtin_nontemporal_store(mv, (f32x4*)m + i); __builtin_nontemporal_store(vv, (f32x4*)v + i);
        if (shadow) ((bf16x4*)shadow)[i] = __builtin_convertvector(pv, bf16x4);
        if (words) ((u32x4*)words)[i] = x3_words(__builtin_bit_cast(u32x4, pv));        // (bf16x3 mode: the word shadow of the arena)
    }
    // tail (n not a multiple of 4)
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
        long i = n4 * 4 + threadIdx.x;
        float pe = p[i], me = m[i], ve = v[i];
        adamw_elem(pe, me, ve, grad1(i), c);
        p[i] = pe; m[i] = me; v[i] = ve;
        if (shadow) { __bf16 h = (__bf16)p[i]; shadow[i] = __builtin_bit_cast(uint16_t, h); }
        if (words) { const u32x4 w4 = x3_words((u32x4){__builtin_bit_cast(uint32_t, p[i]), 0u, 0u, 0u}); words[i] = w4[0]; }
    }
}

// AdamW over a table of arena ranges: block -> range by binary search over the running block counts; 4096 elements per block
constexpr int AR_BLOCK = 4096;
__global__ void __launch_bounds__(256)
adamw_ranges_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                    uint16_t* __restrict__ shadow, const float* __restrict__ steps, const long* __restrict__ table, int nr,
                    float lr, float b1, float b2, float eps, float wd, uint32_t* __restrict__ words) {
    int lo_r = 0, hi_r = nr - 1;
    const long blk = blockIdx.x;
    while (lo_r < hi_r) {                                  // last range whose first block <= blk (uniform: scalar loads)
        const int mid = (lo_r + hi_r + 1) >> 1;
        if (table[4 * mid + 3] <= blk) lo_r = mid; else hi_r = mid - 1;
    }
    const long lo = table[4 * lo_r], hi = table[4 * lo_r + 1];
    const AdamWCoef c = adamw_coef(lr, b1, b2, eps, wd, steps[table[4 * lo_r + 2]]);
    const long beg = lo + (blk - table[4 * lo_r + 3]) * AR_BLOCK, end = min(hi, beg + AR_BLOCK);
    for (long i = beg / 4 + threadIdx.x; i < end / 4; i += 256) {
        f32x4 pv = __builtin_nontemporal_load((f32x4*)p + i), mv = __builtin_nontemporal_load((f32x4*)m + i), vv = __builtin_nontemporal_load((f32x4*)v + i);
        const f32x4 gv = __builtin_nontemporal_load((const f32x4*)g + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = pv[e], me = mv[e], ve = vv[e];
            adamw_elem(pe, me, ve, gv[e], c);
            pv[e] = pe; mv[e] = me; vv[e] = ve;
        }
        __builtin_nontemporal_store(pv, (f32x4*)p + i); __builtin_nontemporal_store(mv, (f32x4*)m + i); __builtin_nontemporal_store(vv, (f32x4*)v + i);
        if (shadow) ((bf16x4*)shadow)[i] = __builtin_convertvector(pv, bf16x4);
        if (words) ((u32x4*)words)[i] = x3_words(__builtin_bit_cast(u32x4, pv));
    }
}

// voxel chunks for the (chunk, batch item) kernels: about `blocks` workgroups in all, each a whole number of `step`-voxel
// iterations of its threads (at least one)
inline void in_chunks(long V, int B, long step, long blocks, long& vpb, int& nchunk) {
    vpb = std::max<long>(step, ((V * B + blocks - 1) / blocks + step - 1) / step * step);
    nchunk = (int)((V + vpb - 1) / vpb);
}

inline int grid_for(long total, int per_block = 256, int cap = 8192) {
    long b = (total + per_block - 1) / per_block;
    return (int)std::max<long>(1, std::min<long>(b, cap));
}

}  // namespace

// ============================================================================================ C ABI
extern "C" int unetr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, void* y_bf16,
                                   float* mean, float* rstd, int M, int H, float eps, void* stream) {
    if (!x || !gamma || !beta || (!y && !y_bf16) || !mean || !rstd || M <= 0) return UNETR_ERR_ARG;
    if ((H & 3) || H > LN_MAXV_MAX * 256) return UNETR_ERR_UNSUPPORTED;
    const LnFwdSrc src{1, 0, nullptr, nullptr, 0, 1, nullptr};
#define LN_FWD(V_) hipLaunchKernelGGL(layernorm_fwd_kernel<V_>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, \
                                      (uint16_t*)y_bf16, mean, rstd, M, H, eps, src)
    if (H <= 768) LN_FWD(3); else if (H <= 1024) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
    return unetr_check_launch();
}

// rows = split-K partial slabs + epilogue (see LnFwdSrc); declared in common.hpp for unetr_gemm_bf16_ln_fwd
int unetr_layernorm_fwd_partials(const float* partials, int splits, long slab, const float* bias, const float* res, long ldr, int res_mod,
                                 float* xout, const float* gamma, const float* beta, float* y, void* y_bf16, float* mean, float* rstd,
                                 int M, int H, float eps, void* stream) {
    if (!partials || splits < 2 || !xout || !gamma || !beta || (!y && !y_bf16) || !mean || !rstd || M <= 0) return UNETR_ERR_ARG;
    if ((H & 3) || H > LN_MAXV_MAX * 256 || (res && (ldr & 3))) return UNETR_ERR_UNSUPPORTED;
    const LnFwdSrc src{splits, slab, bias, res, ldr, res_mod > 0 ? res_mod : M, xout};
    const float* x = partials;
#define LN_FWD(V_) hipLaunchKernelGGL(layernorm_fwd_kernel<V_>, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, \
                                      (uint16_t*)y_bf16, mean, rstd, M, H, eps, src)
    if (H <= 768) LN_FWD(3); else if (H <= 1024) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
    return unetr_check_launch();
}

// dy as `splits` partial slabs (see the kernel); declared in common.hpp for unetr_gemm_bf16_ln_bwd
int unetr_layernorm_bwd_partials(const float* dy, int splits, long slab, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, void* dx_bf16, const float* dres, float* dgamma,
                                 float* dbeta, int M, int H, float* ws, size_t ws_bytes, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || ((dgamma == nullptr) != (dbeta == nullptr)) || M <= 0) return UNETR_ERR_ARG;
    if ((H & 3) || H > LN_MAXV_MAX * 256) return UNETR_ERR_UNSUPPORTED;
    int nblk = cdiv(M, LN_RPB);
    if ((size_t)nblk * 2 * H * sizeof(float) > ws_bytes || !ws) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define LN_BWD(V_) hipLaunchKernelGGL(layernorm_bwd_kernel<V_>, dim3(nblk), dim3(256), 4 * 2 * H * sizeof(float), st, dy, splits, slab, x, gamma, mean, \
                                      rstd, dx, (uint16_t*)dx_bf16, dres, ws, M, H)
    if (H <= 768) LN_BWD(3); else if (H <= 1024) LN_BWD(4); else LN_BWD(8);
    // dgamma == dbeta == NULL: the caller reduces the [nblk][2][H] partials left in ws itself (grouped, off the critical path)
    if (dgamma) hipLaunchKernelGGL(ln_finalize_kernel, dim3(cdiv(2 * H, 64)), dim3(256), 0, st, ws, nblk, H, dgamma, dbeta);
    return unetr_check_launch();
}

extern "C" int unetr_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                   const float* rstd, float* dx, void* dx_bf16, const float* dres, float* dgamma,
                                   float* dbeta, int M, int H, float* ws, size_t ws_bytes, void* stream) {
    return unetr_layernorm_bwd_partials(dy, 1, 0, x, gamma, mean, rstd, dx, dx_bf16, dres, dgamma, dbeta, M, H, ws, ws_bytes, stream);
}

extern "C" int unetr_colsum(const float* x, long ld, int M, int N, float* out, int accumulate,
                            float* ws, size_t ws_bytes, void* stream) {
    if (!x || !out || M <= 0 || N <= 0) return UNETR_ERR_ARG;
    int RB = std::max(1, std::min(cdiv(M, 64), 256));
    if (!ws || (size_t)RB * N * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64), RB), dim3(256), 0, st, x, ld, M, N, ws, RB);
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 256)), dim3(256), 0, st, ws, RB, N, out, accumulate);
    return unetr_check_launch();
}

extern "C" int unetr_colsum_grouped(const unetr_colsum_problem* probs, int n, void* stream) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    for (int base = 0; base < n; base += CS_MAX) {
        ColsumArgs a;
        a.n = std::min(CS_MAX, n - base);
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const unetr_colsum_problem& q = probs[base + i];
            if (!q.x || !q.out || q.M <= 0 || q.N <= 0) return UNETR_ERR_ARG;
            int vec = q.N % 4 == 0 && q.ld % 4 == 0 && (((uintptr_t)q.x | (uintptr_t)q.out) & 15) == 0;
            if (q.x_bf16) {
                if (q.N % 8 || q.ld % 8 || ((uintptr_t)q.x & 15)) return UNETR_ERR_UNSUPPORTED;
                vec = 2;
            }
            a.p[i] = ColsumProblem{(const float*)q.x, q.out, q.ld, q.M, q.N, blocks, vec};
            blocks += cdiv(q.N, 64);
        }
        hipLaunchKernelGGL(colsum_grouped_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    return unetr_check_launch();
}

static int in_check(int C, long ld, int act16 = 0) {
    const int W = act16 ? 8 : 4;            // one 16-byte access per thread
    if ((C % W) || C > 1024 || (ld % W) || (256 % (C / W)) != 0) return UNETR_ERR_UNSUPPORTED;
    return UNETR_OK;
}

extern "C" int unetr_reduce_rows_grouped(const unetr_reduce_problem* probs, int n, void* stream) {
    if (!probs || n <= 0) return UNETR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += RR_MAX) {
        RrArgs a;
        a.n = std::min(RR_MAX, n - base);
        int blocks = 0;
        for (int i = 0; i < a.n; ++i) {
            const unetr_reduce_problem& q = probs[base + i];
            if (!q.part || !q.dst || q.rows <= 0 || q.n <= 0) return UNETR_ERR_ARG;
            const int vec = (q.n % 4 == 0 && (((uintptr_t)q.part | (uintptr_t)q.dst) & 15) == 0) ? 1 : 0;      // whole 16-byte quads per row
            const int nb = (int)std::min<long>(vec ? (q.n + 127) / 128 : (q.n + 31) / 32, 8192);
            a.p[i] = RrProblem{q.part, q.dst, q.n, q.rows, blocks, nb, vec};
            blocks += nb;
        }
        hipLaunchKernelGGL(reduce_rows_grouped_kernel, dim3(blocks), dim3(256), 0, st, a);
    }
    return unetr_check_launch();
}

extern "C" int unetr_instnorm_stats(const void* x, long ld, int B, long V, int C, float eps, float* stats,
                                    float* ws, size_t ws_bytes, int act16, void* stream) {
    if (!x || !stats || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    if (int e = in_check(C, ld, act16)) return e;
    int nchunk = cdiv(V, IN_VPB);
    if (!ws || (size_t)B * nchunk * 2 * C * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int cvn = C / (act16 ? 8 : 4), nphase = 256 / cvn;
    ACT_DISPATCH(act16, hipLaunchKernelGGL(in_stats_kernel<AT>, dim3(nchunk, B), dim3(256), (size_t)2 * nphase * C * 4, st, (const AT*)x, ld, V, C, ws));
    hipLaunchKernelGGL(in_stats_final_kernel, dim3(B * C), dim3(64), 0, st, ws, nchunk, V, C, eps, stats, B);
    return unetr_check_launch();
}

// statistics from partial sums produced elsewhere (the fused conv forward): part [B][nchunk][2][C]
__global__ void __launch_bounds__(256)
in_stats_final_block_kernel(const float* __restrict__ part, int nchunk, long V, int C, float eps, float* __restrict__ stats,
                            const float* __restrict__ part_b, float* __restrict__ stats_b) {
    // blockIdx.y == 1: the second statistics set of the same launch (the 1x1x1 branch of a fused residual-block front)
    if (blockIdx.y) { part = part_b; stats = stats_b; }
    __shared__ double sm[2][4];
    const int i = blockIdx.x, b = i / C, c = i - b * C, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0, q = 0.0;
    int k = threadIdx.x;
    for (; k + 3 * 256 < nchunk; k += 4 * 256) {          // four rows in flight per thread (up to 4096 partial rows per batch item)
        float ps[4], pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* p = part + ((long)b * nchunk + k + 256 * u) * 2 * C;
            ps[u] = p[c]; pq[u] = p[C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)ps[u]; q += (double)pq[u]; }
    }
    for (; k < nchunk; k += 256) {
        const float* p = part + ((long)b * nchunk + k) * 2 * C;
        s += (double)p[c];
        q += (double)p[C + c];
    }
    s = wave_sum_d(s);
    q = wave_sum_d(q);
    if (lane == 0) { sm[0][wave] = s; sm[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]);
        q = (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3]);
        double mu = s / (double)V, var = q / (double)V - mu * mu;
        if (var < 0.0) var = 0.0;
        stats[2 * i] = (float)mu;
        stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

extern "C" int unetr_instnorm_stats_finalize(const float* part, int nchunk, int B, long V, int C, float eps, float* stats,
                                             void* stream) {
    if (!part || !stats || nchunk <= 0 || B <= 0 || C <= 0 || V <= 0) return UNETR_ERR_ARG;
    hipLaunchKernelGGL(in_stats_final_block_kernel, dim3(B * C), dim3(256), 0, (hipStream_t)stream, part, nchunk, V, C, eps, stats,
                       (const float*)nullptr, (float*)nullptr);
    return unetr_check_launch();
}

// two statistics sets with the same geometry in ONE launch (declared in common.hpp; conv3.hip: conv + 1x1x1 branch)
int unetr_instnorm_stats_finalize2(const float* part, const float* part_b, int nchunk, int B, long V, int C, float eps,
                                   float* stats, float* stats_b, void* stream) {
    if (!part || !stats || !part_b || !stats_b || nchunk <= 0 || B <= 0 || C <= 0 || V <= 0) return UNETR_ERR_ARG;
    hipLaunchKernelGGL(in_stats_final_block_kernel, dim3(B * C, 2), dim3(256), 0, (hipStream_t)stream, part, nchunk, V, C, eps, stats,
                       part_b, stats_b);
    return unetr_check_launch();
}

extern "C" int unetr_instnorm_apply(const void* x, long ldx, const float* sa, const void* x2, long ldx2, const float* sb,
                                    void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream) {
    if (!x || !sa || !y || (x2 && !sb)) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    if ((C % W) || (ldx % W) || (ldy % W) || (x2 && (ldx2 % W))) return UNETR_ERR_UNSUPPORTED;
    long total = (long)B * V * (C / W);
    const int cvn = C / W;
    const bool al = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)x2) & 15) == 0;
    if (al && cvn <= 256 && (cvn & (cvn - 1)) == 0 && B <= 65535) {
        long vpb; int nchunk;
        in_chunks(V, B, 2 * (256 / cvn), 2048, vpb, nchunk);
        if (x2) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_apply_hoist_kernel<AT, true>), dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, (const AT*)x, ldx, sa,
                                                       (const AT*)x2, ldx2, sb, (AT*)y, ldy, V, vpb, C, lrelu));
        else ACT_DISPATCH(act16, hipLaunchKernelGGL((in_apply_hoist_kernel<AT, false>), dim3(nchunk, B), dim3(256), 0, (hipStream_t)stream, (const AT*)x, ldx, sa,
                                                    (const AT*)x2, ldx2, sb, (AT*)y, ldy, V, vpb, C, lrelu));
        return unetr_check_launch();
    }
    ACT_DISPATCH(act16, hipLaunchKernelGGL(in_apply_kernel<AT>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const AT*)x, ldx, sa,
                                           (const AT*)x2, ldx2, sb, (AT*)y, ldy, B, V, C, lrelu));
    return unetr_check_launch();
}

// blocks of the FIN kernels: every block re-reduces its batch item's partial rows (up to 128 KB), and a CU pulls ~70 GB/s from
// L2: with the plain passes' 8 blocks of 256 threads per CU that prologue cost more than the 5 us finalize launch it replaces
// (measured, round 4) -- so ONE block of 1024 threads per CU, more voxels in flight per thread
static const long IN_FIN_BLOCKS = 256;        // one 1024-thread block per CU
static inline bool in_fin_ok(int C, int W, int nsp) {
    const int cvn = C / W;
    return C % W == 0 && cvn >= 1 && cvn <= IN_FIN_NT && (cvn & (cvn - 1)) == 0 && C <= 128 && nsp * C <= IN_FIN_MAXCOL;
}

/* y = lrelu?(norm(x) [+ norm(x2)]) with the statistics formed in the kernel's prologue from the partial rows of the conv(s) that
 * produced x / x2 (unetr_conv3_fwd_parts); stats_a / stats_b [B][C][2] are written for the later consumers */
extern "C" int unetr_instnorm_apply_fin(const void* x, long ldx, const float* part_a, int rows_a, const void* x2, long ldx2,
                                        const float* part_b, int rows_b, float* stats_a, float* stats_b, float eps,
                                        void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream) {
    if (!x || !part_a || !stats_a || !y || rows_a <= 0 || B <= 0 || V <= 0 || (x2 && (!part_b || !stats_b || rows_b <= 0))) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    if (!in_fin_ok(C, W, 2) || (ldx % W) || (ldy % W) || (x2 && (ldx2 % W)) || B > 65535) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)x2) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W;
    long vpb; int nchunk;
    hipStream_t st = (hipStream_t)stream;
    if (x2) {
        int u = 2;                           // voxels in flight per thread (2 / 3 / 4 measured inside the step: 63.7 / 66.4 / 63.4 us per step)
        if (const char* e = getenv("UNETR_IN_U_FWD2")) { const int v = atoi(e); if (v >= 2 && v <= 4) u = v; }
        in_chunks(V, B, u * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
#define FWD2(U_) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_apply_fin_kernel<AT, true, U_>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, st, (const AT*)x, ldx, part_a, rows_a, \
                                                       (const AT*)x2, ldx2, part_b, rows_b, stats_a, stats_b, eps, (AT*)y, ldy, V, vpb, C, lrelu))
        if (u == 2) FWD2(2); else if (u == 3) FWD2(3); else FWD2(4);
#undef FWD2
    } else {
        in_chunks(V, B, 4 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
        ACT_DISPATCH(act16, hipLaunchKernelGGL((in_apply_fin_kernel<AT, false, 4>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, st, (const AT*)x, ldx, part_a, rows_a,
                                               (const AT*)x2, ldx2, part_b, rows_b, stats_a, stats_b, eps, (AT*)y, ldy, V, vpb, C, lrelu));
    }
    return unetr_check_launch();
}

/* second half of unetr_instnorm_bwd with the finalize of the partial sums in the kernel's prologue: part [B][nrows][nsp][C]
 * (nsp 3: rows of in_bwd_reduce_kernel; nsp 2: single form, rows written by unetr_conv3_dgrad_stats) */
extern "C" int unetr_instnorm_bwd_apply_fin(const void* dy, long lddy, const void* x, long ldx, const float* sa,
                                            const void* x2, long ldx2, const float* sb, const float* part, int nrows, int nsp,
                                            void* dx, long lddx, void* dx2, long lddx2, int B, long V, int C, int lrelu, int act16, void* stream) {
    if (!dy || !x || !sa || !dx || !part || nrows <= 0 || (x2 && (!sb || !dx2)) || (nsp != 2 && nsp != 3) || (x2 && nsp != 3)) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    if (!in_fin_ok(C, W, nsp) || B > 65535) return UNETR_ERR_UNSUPPORTED;
    if ((lddy % W) || (ldx % W) || (lddx % W) || (x2 && ((ldx2 % W) || (lddx2 % W)))) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)x2 | (uintptr_t)dx | (uintptr_t)dx2) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W;
    long vpb; int nchunk;
    hipStream_t st = (hipStream_t)stream;
    if (x2) {
        in_chunks(V, B, 1 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
        ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_apply_fin_kernel<AT, true, 1>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, st, (const AT*)dy, lddy, (const AT*)x, ldx, sa,
                                               (const AT*)x2, ldx2, sb, part, nrows, nsp, (AT*)dx, lddx, (AT*)dx2, lddx2, V, vpb, C, lrelu));
    } else {
        in_chunks(V, B, 2 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
        ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_apply_fin_kernel<AT, false, 2>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, st, (const AT*)dy, lddy, (const AT*)x, ldx, sa,
                                               (const AT*)x2, ldx2, sb, part, nrows, nsp, (AT*)dx, lddx, (AT*)dx2, lddx2, V, vpb, C, lrelu));
    }
    return unetr_check_launch();
}

/* The block end of the residual block on the image (<= 4 input channels): y = lrelu(norm(x) + norm(c3)) with c3 = conv1x1x1(img; w3)
 * formed per voxel from the image instead of read from memory (see ImgBranch).  part_b = the partial rows of c3's statistics. */
extern "C" int unetr_instnorm_apply_fin_img(const void* x, long ldx, const float* part_a, int rows_a, const float* img, int Cin, const float* w3,
                                            const float* part_b, int rows_b, float* stats_a, float* stats_b, float eps,
                                            void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream) {
    if (!x || !part_a || !stats_a || !y || !img || !w3 || !part_b || !stats_b || rows_a <= 0 || rows_b <= 0 || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    if (Cin < 1 || Cin > 4 || !in_fin_ok(C, W, 2) || (ldx % W) || (ldy % W) || B > 65535 || 64 % (C / W)) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W;
    long vpb; int nchunk;
    in_chunks(V, B, 2 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
#define IMG_APPLY(CIN_) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_apply_fin_img_kernel<AT, 2, CIN_>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, (hipStream_t)stream, \
                                                              (const AT*)x, ldx, part_a, rows_a, img, Cin, w3, part_b, rows_b, stats_a, stats_b, eps, (AT*)y, ldy, V, vpb, C, lrelu))
    if (Cin == 1) IMG_APPLY(1); else IMG_APPLY(4);
#undef IMG_APPLY
    return unetr_check_launch();
}

/* backward of the same block end: dx of the 3x3x3 branch, and the weight gradient of the 1x1x1 branch as partial rows
 * dw3_part [*rows_out][C][Cin] (caller-allocated for UNETR_IN_IMG_MAX_ROWS rows) for unetr_reduce_rows_grouped; the branch's
 * feature-map gradient is never stored.  ws: >= B * 1024 * 3 * C floats. */
extern "C" int unetr_instnorm_bwd_img(const void* dy, long lddy, const void* x, long ldx, const float* sa, const float* img, int Cin,
                                      const float* w3, const float* sb, void* dx, long lddx, float* dw3_part, int* rows_out,
                                      int B, long V, int C, int lrelu, float* ws, size_t ws_bytes, int act16, void* stream) {
    if (!dy || !x || !sa || !img || !w3 || !sb || !dx || !dw3_part || !rows_out || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    if (int e = in_check(C, ldx, act16)) return e;
    const int W = act16 ? 8 : 4;
    if (Cin < 1 || Cin > 4 || !in_fin_ok(C, W, 3) || (lddy % W) || (lddx % W) || B > 65535 || 64 % (C / W)) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W, nphase = 256 / cvn;
    const long step = 2L * nphase;
    long vpb = std::max<long>(2 * step, cdiv(cdiv((long)V * B, 768L), step) * step);
    const int nchunk = (int)cdiv(V, vpb);
    if (!ws || (size_t)B * nchunk * 3 * C * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
#define IMG_RED(CIN_) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_reduce_img_kernel<AT, 2, CIN_>), dim3(nchunk, B), dim3(256), (size_t)3 * nphase * C * 4, st, \
                                                            (const AT*)dy, lddy, (const AT*)x, ldx, sa, img, Cin, w3, sb, V, vpb, C, lrelu, ws))
    if (Cin == 1) IMG_RED(1); else IMG_RED(4);
#undef IMG_RED
    long vpa; int nca;
    in_chunks(V, B, 1 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpa, nca);
    if ((long)nca * B > UNETR_IN_IMG_MAX_ROWS) return UNETR_ERR_UNSUPPORTED;
#define IMG_BAPPLY(CIN_) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_apply_fin_img_kernel<AT, 1, CIN_>), dim3(nca, B), dim3(IN_FIN_NT), 0, st, (const AT*)dy, lddy, \
                                                               (const AT*)x, ldx, sa, img, Cin, w3, sb, ws, nchunk, (AT*)dx, lddx, dw3_part, V, vpa, C, lrelu))
    if (Cin == 1) IMG_BAPPLY(1); else IMG_BAPPLY(4);
#undef IMG_BAPPLY
    *rows_out = nca * B;
    return unetr_check_launch();
}

extern "C" int unetr_instnorm_bwd(const void* dy, long lddy, const void* x, long ldx, const float* sa,
                                  const void* x2, long ldx2, const float* sb, void* dx, long lddx, void* dx2, long lddx2,
                                  int B, long V, int C, int lrelu, float* ws, size_t ws_bytes, int act16, void* stream) {
    if (!dy || !x || !sa || !dx || (x2 && (!sb || !dx2))) return UNETR_ERR_ARG;
    if (int e = in_check(C, ldx, act16)) return e;
    const int W = act16 ? 8 : 4;
    if ((lddy % W) || (lddx % W) || (x2 && ((ldx2 % W) || (lddx2 % W)))) return UNETR_ERR_UNSUPPORTED;
    // about three blocks per CU, each at least two full iterations of its threads (a fixed 1024-voxel block left the 12^3 x
    // 128-channel layer on TWO blocks: 51 us for 0.9 MB); offsets inside a block are 32-bit
    const int cvn = C / W, nphase = 256 / cvn;
    int UD = 2;                                  // voxels in flight per thread: dual / single form
    const int US = 4;
    if (const char* e = getenv("UNETR_IN_U_RED2")) { const int v = atoi(e); if (v >= 2 && v <= 4) UD = v; }
    const long step = (long)(x2 ? UD : US) * nphase;
    long vpb = std::max<long>(2 * step, cdiv(cdiv((long)V * B, 768L), step) * step);
    const long ldmax = std::max(std::max(lddy, ldx), x2 ? ldx2 : 0L);
    while (vpb > 2 * step && (vpb + 256) * ldmax * 4 >= (1L << 31)) vpb -= step;
    if ((vpb + 256) * ldmax * 4 >= (1L << 31)) return UNETR_ERR_UNSUPPORTED;
    int nchunk = (int)cdiv(V, vpb);
    size_t need = ((size_t)B * nchunk * 3 * C + (size_t)B * C * 3) * sizeof(float);
    if (!ws || need > ws_bytes) return UNETR_ERR_WORKSPACE;
    float* sums = ws + (size_t)B * nchunk * 3 * C;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds_bytes = (size_t)3 * nphase * C * 4;
#define IN_RED(DUAL_, U_) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_reduce_kernel<AT, DUAL_, U_>), dim3(nchunk, B), dim3(256), lds_bytes, st, (const AT*)dy, lddy, \
                                                                 (const AT*)x, ldx, sa, (const AT*)x2, ldx2, sb, V, vpb, C, lrelu, ws))
    if (x2) { if (UD == 2) IN_RED(true, 2); else if (UD == 3) IN_RED(true, 3); else IN_RED(true, 4); } else IN_RED(false, 4);
#undef IN_RED
    // the finalize of the partial sums rides in the prologue of the apply kernel where that form exists (UNETR_IN_FIN=0: the
    // separate finalize launch, kept for A/B measurements and as the route for shapes the folded form declines)
    const char* fe = getenv("UNETR_IN_FIN");
    const bool fin_on = !(fe && atoi(fe) == 0);
    if (fin_on) {
        const int rc = unetr_instnorm_bwd_apply_fin(dy, lddy, x, ldx, sa, x2, ldx2, sb, ws, nchunk, 3, dx, lddx, dx2, lddx2, B, V, C, lrelu, act16, stream);
        if (rc != UNETR_ERR_UNSUPPORTED) return rc;
    }
    hipLaunchKernelGGL(in_bwd_final_kernel, dim3(B * C), dim3(64), 0, st, ws, nchunk, V, C, B, sums);
    long total = (long)B * V * (C / W);
    const bool al = (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)x2 | (uintptr_t)dx | (uintptr_t)dx2) & 15) == 0;
    if (al && B <= 65535) {          // (in_check: C / W divides the block)
        long vpa; int nca;
        in_chunks(V, B, 2 * nphase, 2048, vpa, nca);
        if (x2) ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_apply_hoist_kernel<AT, true>), dim3(nca, B), dim3(256), 0, st, (const AT*)dy, lddy, (const AT*)x, ldx, sa,
                                                       (const AT*)x2, ldx2, sb, sums, (AT*)dx, lddx, (AT*)dx2, lddx2, V, vpa, C, lrelu));
        else ACT_DISPATCH(act16, hipLaunchKernelGGL((in_bwd_apply_hoist_kernel<AT, false>), dim3(nca, B), dim3(256), 0, st, (const AT*)dy, lddy, (const AT*)x, ldx, sa,
                                                    (const AT*)x2, ldx2, sb, sums, (AT*)dx, lddx, (AT*)dx2, lddx2, V, vpa, C, lrelu));
        return unetr_check_launch();
    }
    ACT_DISPATCH(act16, hipLaunchKernelGGL(in_bwd_apply_kernel<AT>, dim3(grid_for(total)), dim3(256), 0, st, (const AT*)dy, lddy, (const AT*)x, ldx, sa,
                                           (const AT*)x2, ldx2, sb, sums, (AT*)dx, lddx, (AT*)dx2, lddx2, B, V, C, lrelu));
    return unetr_check_launch();
}

extern "C" int unetr_nchw_to_nhwc(const float* x, void* y, long ldy, int B, int C, long V, int act16, void* stream) {
    if (!x || !y) return UNETR_ERR_ARG;
    // per batch: src [C rows, V cols] -> dst [V, C]
    if (cdiv(C, 32) > 65535 || B > 65535) return UNETR_ERR_ARG;
    ACT_DISPATCH(act16, hipLaunchKernelGGL((transpose_kernel<float, AT>), dim3(cdiv(C, 32), cdiv(V, 32), B), dim3(256), 0, (hipStream_t)stream,
                                           x, V, (long)C * V, (AT*)y, ldy, V * ldy, (long)C, V, 0));
    return unetr_check_launch();
}
extern "C" int unetr_nhwc_to_nchw(const void* x, long ldx, float* y, int B, int C, long V, int accumulate, int act16, void* stream) {
    if (!x || !y) return UNETR_ERR_ARG;
    if (cdiv(C, 32) > 65535 || B > 65535) return UNETR_ERR_ARG;
    ACT_DISPATCH(act16, hipLaunchKernelGGL((transpose_kernel<AT, float>), dim3(cdiv(V, 32), cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream,
                                           (const AT*)x, ldx, V * ldx, y, V, (long)C * V, V, (long)C, accumulate));
    return unetr_check_launch();
}

extern "C" int unetr_patch_gather(const float* x, float* patches, void* patches_bf16, int B, int C, int D, int H, int W, int P, void* stream) {
    if (!x || (!patches && !patches_bf16) || P <= 0 || D % P || H % P || W % P) return UNETR_ERR_ARG;
    long total = (long)B * C * D * H * W;
    const bool vec = C == 1 && P % 4 == 0 && W % 4 == 0 && total < (1L << 31) && (((uintptr_t)x | (uintptr_t)patches) & 15) == 0 &&
                     ((uintptr_t)patches_bf16 & 7) == 0;
    if (vec) hipLaunchKernelGGL(patch_gather_kernel<true>, dim3(grid_for(total / 4)), dim3(256), 0, (hipStream_t)stream, x, patches, (uint16_t*)patches_bf16, B, C, D, H, W, P);
    else hipLaunchKernelGGL(patch_gather_kernel<false>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, patches, (uint16_t*)patches_bf16, B, C, D, H, W, P);
    return unetr_check_launch();
}

extern "C" int unetr_counter_add(float* y, const float* inc, int n, void* stream) {
    if (!y || !inc || n <= 0) return UNETR_ERR_ARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, inc, n);
    return unetr_check_launch();
}

extern "C" int unetr_copy_rows(void* y, long ldy, const void* a, long lda, long rows, int cols, int accumulate, int act16, void* stream) {
    const int W = act16 ? 8 : 4;
    if (!y || !a || (cols % W) || (ldy % W) || (lda % W)) return UNETR_ERR_ARG;
    ACT_DISPATCH(act16, hipLaunchKernelGGL(add_rows_kernel<AT>, dim3(grid_for(rows * (cols / W))), dim3(256), 0, (hipStream_t)stream, (AT*)y, ldy,
                                           (const AT*)a, lda, rows, cols, accumulate));
    return unetr_check_launch();
}

extern "C" int unetr_outconv_fwd(const void* x, long ldx, const float* w, const float* bias, float* logits,
                                 int B, long V, int Cin, int Cout, int act16, void* stream) {
    if (!x || !w || !logits) return UNETR_ERR_ARG;
    if (Cout > OC_MAXCO || Cin > OC_MAXCI || (Cin & 3) || (ldx & 3)) return UNETR_ERR_UNSUPPORTED;
    ACT_DISPATCH(act16, hipLaunchKernelGGL(outconv_fwd_kernel<AT>, dim3(grid_for((long)B * V)), dim3(256), 0, (hipStream_t)stream, (const AT*)x, ldx, w, bias,
                                           logits, B, V, Cin, Cout));
    return unetr_check_launch();
}

extern "C" int unetr_outconv_bwd(const float* dlogits, const void* x, long ldx, const float* w, void* dx, long lddx,
                                 float* dw, float* dbias, int B, long V, int Cin, int Cout,
                                 float* ws, size_t ws_bytes, int act16, void* stream) {
    if (!dlogits || !x || !w || !dx || !dw || !dbias) return UNETR_ERR_ARG;
    if (Cout > OC_MAXCO || Cin > OC_MAXCI || (Cin & 3) || (ldx & 3) || (lddx & 3)) return UNETR_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int nblk = grid_for((long)B * V, 256, 1024);
    const bool wg = Cout <= 4 && Cin <= 16 && ((uintptr_t)x & 15) == 0;      // weight gradient from the same pass
    const int Wd = act16 ? 8 : 4;
    const bool small = wg && (Cin == Wd || Cin == 2 * Wd || Cin == 4 * Wd) && (ldx % Wd) == 0 && (lddx % Wd) == 0 && ((uintptr_t)dx & 15) == 0;
    const int np = Cout + (wg ? Cout * Cin : 0);
    size_t part_bytes = (size_t)nblk * np * sizeof(float);
    size_t part_al = (part_bytes + 255) & ~(size_t)255;
    if (!ws || part_al + 4096 > ws_bytes) return UNETR_ERR_WORKSPACE;
    if (small) ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_bwd_small_kernel<AT>), dim3(nblk), dim3(256), 0, st, dlogits, w, (const AT*)x, ldx, (AT*)dx, lddx, ws, B, V, Cin, Cout));
    else if (wg) ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_bwd_kernel<true, AT>), dim3(nblk), dim3(256), 0, st, dlogits, w, (const AT*)x, ldx, (AT*)dx, lddx, ws, B, V, Cin, Cout));
    else ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_bwd_kernel<false, AT>), dim3(nblk), dim3(256), 0, st, dlogits, w, (const AT*)x, ldx, (AT*)dx, lddx, ws, B, V, Cin, Cout));
    if (int e = unetr_check_launch()) return e;
    float* ws2 = (float*)((char*)ws + part_al);
    size_t ws2_bytes = ws_bytes - part_al;
    // dbias (and dw) = column sums of the [nblk, np] partials (ws2 holds the colsum's own partials)
    {
        int RB = std::max(1, std::min(cdiv(nblk, 64), 256));
        if ((size_t)RB * np * sizeof(float) > ws2_bytes) return UNETR_ERR_WORKSPACE;
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(np, 64), RB), dim3(256), 0, st, ws, (long)np, nblk, np, ws2, RB);
        hipLaunchKernelGGL(outconv_final_kernel, dim3(cdiv(np, 256)), dim3(256), 0, st, ws2, RB, np, Cout, dbias, dw);
    }
    if (wg) return unetr_check_launch();
    // larger heads (the reference's default is 14 classes): per-workgroup partials of dw from fp32- or bf16-stored x, then the
    // same column-sum pair.  `ws` is free again: the launches that read the bias partials are ahead of these in the stream.
    {
        const int npw = Cout * Cin;
        const long ntile = ((long)B * V + OCW_VT - 1) / OCW_VT;
        const int nb2 = (int)std::min<long>(ntile, 1024);
        const size_t p2 = ((size_t)nb2 * npw * sizeof(float) + 255) & ~(size_t)255;
        const int RB = std::max(1, std::min(cdiv(nb2, 64), 256));
        if (p2 + (size_t)RB * npw * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
        float* wsb = (float*)((char*)ws + p2);
        ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_wgrad_generic_kernel<AT>), dim3(nb2), dim3(256), 0, st, dlogits, (const AT*)x, ldx, ws, B, V, Cin, Cout));
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(npw, 64), RB), dim3(256), 0, st, ws, (long)npw, nb2, npw, wsb, RB);
        hipLaunchKernelGGL(outconv_final_kernel, dim3(cdiv(npw, 256)), dim3(256), 0, st, wsb, RB, npw, 0, dbias, dw);
    }
    return unetr_check_launch();
}

/* The last residual block's end + the out conv in one pass each way (see outconv_in_fwd_kernel).  Forward: logits [B][Cout][V] from
 * c2 / c3 (the block's second 3x3x3 conv and its 1x1x1 branch, channels-last, C channels) and their InstanceNorm partial rows
 * part_a / part_b [B][rows][2][C]; stats_a / stats_b [B][C][2] are written for backward.  UNSUPPORTED = take the unfused sequence. */
extern "C" int unetr_outconv_in_fwd(const void* c2, long ld2, const float* part_a, int rows_a, const void* c3, long ld3, const float* part_b,
                                    int rows_b, float* stats_a, float* stats_b, float eps, const float* w, const float* bias, float* logits,
                                    int B, long V, int C, int Cout, int act16, void* stream) {
    if (!c2 || !c3 || !part_a || !part_b || !stats_a || !stats_b || !w || !logits || rows_a <= 0 || rows_b <= 0 || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    if (Cout < 1 || Cout > 4 || !in_fin_ok(C, W, 2) || C > 16 || 64 % (C / W) || (ld2 % W) || (ld3 % W) || B > 65535) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)c2 | (uintptr_t)c3) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W;
    long vpb; int nchunk;
    in_chunks(V, B, 2 * (IN_FIN_NT / cvn), IN_FIN_BLOCKS, vpb, nchunk);
    ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_in_fwd_kernel<AT, 2>), dim3(nchunk, B), dim3(IN_FIN_NT), 0, (hipStream_t)stream, (const AT*)c2, ld2, part_a, rows_a,
                                           (const AT*)c3, ld3, part_b, rows_b, stats_a, stats_b, eps, w, bias, logits, V, vpb, C, Cout));
    return unetr_check_launch();
}

/* rows of in_part (per batch item) that unetr_outconv_in_bwd writes for this shape; 0 = unsupported */
extern "C" long unetr_outconv_in_bwd_rows(int B, long V, int C, int act16) {
    const int W = act16 ? 8 : 4;
    if (B <= 0 || V <= 0 || C % W || C > 16 || 64 % (C / W)) return 0;
    const int nphase = 256 / (C / W);
    const long step = 2L * nphase;
    const long vpb = std::max<long>(2 * step, cdiv(cdiv((long)V * B, 768L), step) * step);
    return cdiv(V, vpb);
}

/* Backward of the pair: dout [B][V][C] (pitch lddo, storage type of c2) = W^T dlogits; in_part [B][rows][3][C] = partial sums of the
 * block end's InstanceNorm backward (rows = unetr_outconv_in_bwd_rows; consumed by unetr_instnorm_bwd_apply_fin with nsp = 3 and
 * dy = dout); dw [Cout][C], dbias [Cout].  ws: the out conv's partial rows + their column sums. */
extern "C" int unetr_outconv_in_bwd(const float* dlogits, const void* c2, long ld2, const float* sa, const void* c3, long ld3, const float* sb,
                                    const float* w, void* dout, long lddo, float* in_part, float* dw, float* dbias,
                                    int B, long V, int C, int Cout, float* ws, size_t ws_bytes, int act16, void* stream) {
    if (!dlogits || !c2 || !c3 || !sa || !sb || !w || !dout || !in_part || !dw || !dbias || B <= 0 || V <= 0) return UNETR_ERR_ARG;
    const int W = act16 ? 8 : 4;
    const long rows = unetr_outconv_in_bwd_rows(B, V, C, act16);
    if (Cout < 1 || Cout > 4 || rows <= 0 || rows > 65535 || (ld2 % W) || (ld3 % W) || (lddo % W) || B > 65535) return UNETR_ERR_UNSUPPORTED;
    if ((((uintptr_t)c2 | (uintptr_t)c3 | (uintptr_t)dout) & 15) != 0) return UNETR_ERR_UNSUPPORTED;
    const int cvn = C / W, nphase = 256 / cvn;
    const long step = 2L * nphase;
    const long vpb = std::max<long>(2 * step, cdiv(cdiv((long)V * B, 768L), step) * step);
    const int nblk = (int)rows * B, np = Cout + Cout * C;
    const size_t part_al = ((size_t)nblk * np * sizeof(float) + 255) & ~(size_t)255;
    const int RB = std::max(1, std::min(cdiv(nblk, 64), 256));
    if (!ws || part_al + (size_t)RB * np * sizeof(float) > ws_bytes) return UNETR_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    ACT_DISPATCH(act16, hipLaunchKernelGGL((outconv_in_bwd_kernel<AT>), dim3((unsigned)rows, B), dim3(256), (size_t)3 * nphase * C * 4, st, dlogits, w, (const AT*)c2, ld2, sa,
                                           (const AT*)c3, ld3, sb, (AT*)dout, lddo, in_part, ws, V, vpb, C, Cout));
    float* ws2 = (float*)((char*)ws + part_al);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(np, 64), RB), dim3(256), 0, st, ws, (long)np, nblk, np, ws2, RB);
    hipLaunchKernelGGL(outconv_final_kernel, dim3(cdiv(np, 256)), dim3(256), 0, st, ws2, RB, np, Cout, dbias, dw);
    return unetr_check_launch();
}

static int adamw_launch(float* p, const void* g, int g_bf16, float gscale, float* m, float* v, long n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, const float* step_dev, void* shadow_bf16, void* shadow_x3, void* stream) {
    if (!p || !g || !m || !v || !step_dev || n <= 0 || (reinterpret_cast<uintptr_t>(shadow_bf16) & 7) || (reinterpret_cast<uintptr_t>(shadow_x3) & 15)) return UNETR_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) return UNETR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(g) & (g_bf16 ? 7 : 15)) return UNETR_ERR_ARG;
    long n4 = n >> 2;
    // grid cap: 4096 grid-stride workgroups fill every CU to its thread limit -- the right shape when AdamW runs alone, the wrong
    // one when it runs on a side stream underneath a chain of short latency-bound kernels (they then wait for CU slots until the
    // optimizer kernel has drained): UNETR_ADAMW_GRID limits the resident workgroups in that launch form
    int cap = 4096;
    if (const char* e = getenv("UNETR_ADAMW_GRID")) { const int v = atoi(e); if (v >= 64) cap = v; }
    dim3 grid(grid_for(std::max<long>(n4, 1), 256, cap));
    if (g_bf16)
        hipLaunchKernelGGL(adamw_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, p, g, gscale, m, v, n4, n, lr, beta1, beta2, eps,
                           weight_decay, step_dev, (uint16_t*)shadow_bf16, (uint32_t*)shadow_x3);
    else
        hipLaunchKernelGGL(adamw_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, p, g, gscale, m, v, n4, n, lr, beta1, beta2, eps,
                           weight_decay, step_dev, (uint16_t*)shadow_bf16, (uint32_t*)shadow_x3);
    return unetr_check_launch();
}

extern "C" int unetr_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, const float* step_dev, void* shadow_bf16, void* stream) {
    return adamw_launch(p, g, 0, 1.0f, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, shadow_bf16, nullptr, stream);
}

extern "C" int unetr_adamw_ranges(const unetr_adamw_arena* a, const long* table_dev, int n_ranges, long n_blocks, void* stream) {
    if (!a || !a->param || !a->grad || !a->m || !a->v || !a->steps || !table_dev || n_ranges <= 0 || n_blocks <= 0) return UNETR_ERR_ARG;
    if (((uintptr_t)a->param | (uintptr_t)a->grad | (uintptr_t)a->m | (uintptr_t)a->v | (uintptr_t)a->shadow_x3) & 15 || ((uintptr_t)a->shadow_bf16 & 7)) return UNETR_ERR_ARG;
    if (n_blocks > 0x7fffffffL) return UNETR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(adamw_ranges_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, a->param, a->grad, a->m, a->v,
                       (uint16_t*)a->shadow_bf16, a->steps, table_dev, n_ranges, a->lr, a->beta1, a->beta2, a->eps, a->weight_decay,
                       (uint32_t*)a->shadow_x3);
    return unetr_check_launch();
}

// the data-parallel form: gradients are read from the all-reduced communication buffer (fp32 or bf16) and averaged
// (gscale = 1 / world size) inside the update, so no copy back into the gradient arena is needed
extern "C" int unetr_adamw_reduced(float* p, const void* g, int g_is_bf16, float gscale, float* m, float* v, long n, float lr,
                                   float beta1, float beta2, float eps, float weight_decay, const float* step_dev,
                                   void* shadow_bf16, void* shadow_x3, void* stream) {
    return adamw_launch(p, g, g_is_bf16, gscale, m, v, n, lr, beta1, beta2, eps, weight_decay, step_dev, shadow_bf16, shadow_x3, stream);
}
