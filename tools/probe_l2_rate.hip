// Per-CU L2 -> LDS rate of the three ways a GEMM tile can be staged on gfx950 (what bounds the 432-row ViT GEMMs):
//   mode 0: LDS-DMA (global_load_lds_dwordx4), NS pieces in flight per thread
//   mode 1: global_load_dwordx4 -> registers -> ds_write_b128, NS pieces in flight per thread
//   mode 2: half the pieces by each path
// One 256-thread workgroup per CU; every workgroup streams the SAME `span` bytes (an operand set that sits in L2, as the weights /
// activations of one GEMM launch do) `reps` times.  Prints GB/s per CU and chip-wide.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_l2_rate.hip -o /tmp/probe_l2_rate && /tmp/probe_l2_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NS>
__global__ void __launch_bounds__(256) probe(const char* __restrict__ src, size_t span, int reps, unsigned* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) char lds[NS * 4096 * 2];
    const int tid = threadIdx.x, wave = tid >> 6;
    const size_t npiece = span / 4096;                       // 4 KB per workgroup-wide piece (16 B per thread)
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        // offset the start per workgroup so that CUs do not walk in lockstep
        size_t p0 = (blockIdx.x * 37u) % npiece;
        for (size_t p = 0; p < npiece; p += NS) {
            u32x4 reg[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const size_t q = (p0 + p + s) % npiece;
                const char* g = src + q * 4096 + tid * 16;
                const bool dma = MODE == 0 || (MODE == 2 && (s & 1) == 0);
                if (dma) __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(lds + s * 4096 + wave * 1024), 16, 0, 0);
                else reg[s] = *(const u32x4*)g;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const bool dma = MODE == 0 || (MODE == 2 && (s & 1) == 0);
                if (!dma) *(u32x4*)(lds + NS * 4096 + s * 4096 + tid * 16) = reg[s];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            acc += *(const unsigned*)(lds + ((tid * 4 + p) & (NS * 4096 * 2 - 4)));
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int NS>
static void run(const char* buf, size_t span, unsigned* sink, const char* name) {
    const int reps = 40, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NS><<<cus, 256>>>(buf, span, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE, NS><<<cus, 256>>>(buf, span, reps, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)span * reps;
    printf("%-34s NS %2d span %5.1f MB: %7.1f GB/s per CU   %6.2f TB/s chip\n", name, NS, span / 1e6, bytes / ms / 1e6, bytes * cus / ms / 1e9);
}

int main() {
    const size_t cap = 64u << 20;
    char* buf; unsigned* sink;
    hipMalloc(&buf, cap); hipMalloc(&sink, 64);
    hipMemset(buf, 1, cap);
    for (size_t span : {(size_t)4 << 20, (size_t)16 << 20}) {
        run<0, 4>(buf, span, sink, "LDS-DMA");
        run<0, 8>(buf, span, sink, "LDS-DMA");
        run<0, 12>(buf, span, sink, "LDS-DMA");
        run<1, 4>(buf, span, sink, "registers + ds_write");
        run<1, 8>(buf, span, sink, "registers + ds_write");
        run<1, 12>(buf, span, sink, "registers + ds_write");
        run<2, 8>(buf, span, sink, "half DMA, half registers");
        run<2, 12>(buf, span, sink, "half DMA, half registers");
    }
    return 0;
}
