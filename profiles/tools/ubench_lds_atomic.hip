// Throughput of ds_add_u32 on gfx950 as a function of active lanes and address pattern (one LDS pipe per CU).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds_atomic ubench_lds_atomic.hip && ./ubench_lds_atomic
// Addresses are prepared before the timed loop (8 per lane, rotated by a v_add per use), so the loop is
// 1 VALU + 1 ds_add_u32 per wave-instruction.  16 waves per CU (4 workgroups of 256 threads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pattern 0: random words of a 2304-word histogram; 1: lane l -> word l (conflict free); 2: all lanes one word;
// 3: random among 32 consecutive words; 4: ds_read_b128 broadcast instead of an atomic; 5: no LDS op (VALU only)
__global__ void k(int pattern, int active, int iters, unsigned *sink)
{
    __shared__ unsigned lds[2304 + 64];
    for (int i = threadIdx.x; i < 2304 + 64; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        s = s * 1664525u + 1013904223u;
        unsigned w = pattern == 0 ? (s >> 9) % 2304u : pattern == 1 ? (unsigned)lane : pattern == 2 ? 7u : (s >> 9) % 32u;
        a[u] = w * 4u;
    }
    const bool on = (lane * 2654435761u >> 26) % 64 < (unsigned)active;      // a scattered subset of `active` lanes
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (pattern == 0) { a[u] += 52u; if (a[u] >= 9216u) a[u] -= 9216u; }       // keep moving over the bins
            if (pattern == 4) {
                uint4 v;
                unsigned addr = (a[u] & 0xff0u);
                addr = __builtin_amdgcn_readfirstlane(addr);
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                acc += v.x;
            } else if (pattern == 5) {
                acc += a[u];
            } else if (on) {
                asm volatile("ds_add_u32 %0, %1" ::"v"(a[u]), "v"(1u) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned tot = acc;
    for (int i = threadIdx.x; i < 2304; i += blockDim.x) tot += lds[i];
    if (tot == 0xdeadbeefu) sink[0] = tot;
}

int main()
{
    unsigned *d_sink;
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    struct { int pattern, active; const char *name; } cases[] = {
        {5, 64, "VALU only (loop overhead)"},
        {0, 64, "random bins, 64 lanes"}, {0, 48, "random bins, 48 lanes"}, {0, 32, "random bins, 32 lanes"},
        {0, 29, "random bins, 29 lanes"}, {0, 16, "random bins, 16 lanes"}, {0, 8, "random bins, 8 lanes"},
        {0, 1, "random bins, 1 lane"}, {1, 64, "conflict free, 64 lanes"}, {1, 32, "conflict free, 32 lanes"},
        {2, 64, "one word, 64 lanes"}, {3, 64, "32 consecutive words, 64 lanes"}, {4, 64, "ds_read_b128 broadcast"},
    };
    for (auto &c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(256 * 4), dim3(256), 0, 0, c.pattern, c.active, iters, d_sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-34s %.3f ms  %.2f ns per wave-instruction per CU\n", c.name, best, best * 1e6 / (16.0 * iters * 8));
    }
    return 0;
}
