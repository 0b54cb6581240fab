// Does gfx950 drop LDS atomics whose address lies beyond the workgroup's LDS allocation, and what do they cost?
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds_oob ubench_lds_oob.hip && ./ubench_lds_oob
// Every workgroup (several resident per CU, so a stray write would land in a neighbour's allocation) fills its own
// dynamic LDS with a pattern, fires atomics at addresses beyond its allocation, and checks that its pattern is intact;
// a second kernel times ds_add_u32 with all lanes in range / half the lanes out of range / half the lanes masked off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void oob_kernel(int lds_words, int reach_words, int *bad, int *self_hits)
{
    extern __shared__ unsigned lds[];
    for (int k = threadIdx.x; k < lds_words; k += blockDim.x) lds[k] = 0xA5000000u + k;
    __syncthreads();
    // atomics at word offsets lds_words + 64 .. lds_words + reach_words (beyond the allocation and its granule padding)
    for (int k = lds_words + 64 + threadIdx.x; k < lds_words + reach_words; k += blockDim.x) {
        unsigned addr = (unsigned)k * 4u;
        asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    int wrong = 0;
    for (int k = threadIdx.x; k < lds_words; k += blockDim.x) wrong += lds[k] != 0xA5000000u + k;
    if (wrong) atomicAdd(bad, wrong);
    // and a read from out of range: expected 0
    unsigned addr = (unsigned)(lds_words + 4096 + threadIdx.x) * 4u, v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    if (v != 0u) atomicAdd(self_hits, 1);
}

// mode 0: every lane adds to a random in-range word; 1: odd lanes are sent out of range; 2: odd lanes masked off
__global__ void time_kernel(int mode, int iters, int lds_words, unsigned *sink)
{
    extern __shared__ unsigned lds[];
    for (int k = threadIdx.x; k < lds_words; k += blockDim.x) lds[k] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const bool odd = threadIdx.x & 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            s = s * 1664525u + 1013904223u;
            unsigned w = (s >> 10) % (unsigned)lds_words;
            unsigned addr = w * 4u;
            if (mode == 1 && odd) addr += 0x40000u;            // + 256 KiB: beyond any allocation
            if (mode == 2) {
                if (!odd) asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
            } else {
                asm volatile("ds_add_u32 %0, %1" ::"v"(addr), "v"(1u) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned tot = 0;
    for (int k = threadIdx.x; k < lds_words; k += blockDim.x) tot += lds[k];
    if (tot == 0xdeadbeefu) sink[0] = tot;
    if (threadIdx.x == 0) atomicAdd(&sink[1 + mode], tot == 0 ? 0u : 1u), atomicAdd(&sink[8 + mode], tot);
}

int main()
{
    int *d_bad, *d_hits;
    unsigned *d_sink;
    CHECK(hipMalloc(&d_bad, 4)); CHECK(hipMalloc(&d_hits, 4)); CHECK(hipMalloc(&d_sink, 64));
    CHECK(hipMemset(d_bad, 0, 4)); CHECK(hipMemset(d_hits, 0, 4)); CHECK(hipMemset(d_sink, 0, 64));
    const int lds_words = 2048;     // 8 KiB per workgroup: many workgroups share a CU's 160 KiB
    hipLaunchKernelGGL(oob_kernel, dim3(256 * 12), dim3(256), lds_words * 4, 0, lds_words, 38000, d_bad, d_hits);
    CHECK(hipDeviceSynchronize());
    int bad = -1, hits = -1;
    CHECK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hits, d_hits, 4, hipMemcpyDeviceToHost));
    printf("out-of-range ds_add_u32: corrupted words seen by any workgroup: %d (0 = dropped); non-zero out-of-range reads: %d\n", bad, hits);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int mode = 0; mode < 3; mode++) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(time_kernel, dim3(256 * 4), dim3(256), 2310 * 4, 0, mode, iters, 2310, d_sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        // 4 workgroups x 4 waves per CU share the CU's LDS pipe: wave-instructions per CU = 16 * iters * 8
        printf("mode %d (%s): %.3f ms, %.1f ns per wave-level ds_add_u32 per CU\n", mode,
               mode == 0 ? "64 lanes in range" : mode == 1 ? "32 in range, 32 out of range" : "32 in range, 32 masked off",
               best, best * 1e6 / (16.0 * iters * 8));
    }
    unsigned sink[16];
    CHECK(hipMemcpy(sink, d_sink, 64, hipMemcpyDeviceToHost));
    printf("adds that landed per launch set (x3 reps): mode0 %u mode1 %u mode2 %u\n", sink[8], sink[9], sink[10]);
    return 0;
}
