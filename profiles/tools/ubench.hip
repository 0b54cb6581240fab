// VALU issue-rate microbenchmark for gfx950 (diagnostic, not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define ITER 4096

#define KERNEL(name, body, decl)                                                   \
    __global__ __launch_bounds__(256) void name(float *out, int n)                \
    {                                                                              \
        decl;                                                                      \
        for (int i = 0; i < n; i++) { REP8(body) }                                 \
        if (threadIdx.x == 1023) out[0] = 0;                                       \
    }

__global__ __launch_bounds__(256) void k_fma32(float *out, int n)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 1.0001f, c = 0.5f;
    for (int i = 0; i < n; i++) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
#define UNARY(name, ins, T, cons)                                                                                       \
    __global__ __launch_bounds__(256) void name(float *out, int n)                                                     \
    {                                                                                                                  \
        T a0 = threadIdx.x + 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;                                 \
        for (int i = 0; i < n; i++) {                                                                                  \
            asm volatile(ins " %0, %0\n " ins " %1, %1\n " ins " %2, %2\n " ins " %3, %3\n " ins " %4, %4\n " ins      \
                             " %5, %5\n " ins " %6, %6\n " ins " %7, %7\n"                                             \
                         : "+" cons(a0), "+" cons(a1), "+" cons(a2), "+" cons(a3), "+" cons(a4), "+" cons(a5),         \
                           "+" cons(a6), "+" cons(a7));                                                                \
        }                                                                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }
#define BINARY(name, ins, T, cons)                                                                                      \
    __global__ __launch_bounds__(256) void name(float *out, int n)                                                     \
    {                                                                                                                  \
        T a0 = threadIdx.x + 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8, b = 3;                          \
        for (int i = 0; i < n; i++) {                                                                                  \
            asm volatile(ins " %0, %0, %8\n " ins " %1, %1, %8\n " ins " %2, %2, %8\n " ins " %3, %3, %8\n " ins       \
                             " %4, %4, %8\n " ins " %5, %5, %8\n " ins " %6, %6, %8\n " ins " %7, %7, %8\n"            \
                         : "+" cons(a0), "+" cons(a1), "+" cons(a2), "+" cons(a3), "+" cons(a4), "+" cons(a5),         \
                           "+" cons(a6), "+" cons(a7)                                                                  \
                         : cons(b));                                                                                   \
        }                                                                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }
UNARY(k_sqrt32, "v_sqrt_f32", float, "v")
UNARY(k_fract32, "v_fract_f32", float, "v")
UNARY(k_cvt_f32_i32, "v_cvt_f32_i32", float, "v")
UNARY(k_cvt_i32_f32, "v_cvt_i32_f32", float, "v")
UNARY(k_rsq32, "v_rsq_f32", float, "v")
UNARY(k_floor32, "v_floor_f32", float, "v")
BINARY(k_mul32, "v_mul_f32", float, "v")
BINARY(k_add32, "v_add_f32", float, "v")
BINARY(k_subu32, "v_sub_u32", unsigned, "v")
BINARY(k_andb32, "v_and_b32", unsigned, "v")
BINARY(k_mul64, "v_mul_f64", double, "v")
BINARY(k_add64, "v_add_f64", double, "v")
BINARY(k_pkmul32, "v_pk_mul_f32", double, "v")
BINARY(k_pkadd32, "v_pk_add_f32", double, "v")
BINARY(k_mullo, "v_mul_lo_u32", unsigned, "v")
UNARY(k_rsq64, "v_rsq_f64", double, "v")
UNARY(k_cvt_f64_i32_fake, "v_rndne_f64", double, "v")

__global__ __launch_bounds__(256) void k_cmp32(float *out, int n)
{
    float a0 = threadIdx.x, b = 3;
    unsigned long long m = 0;
    for (int i = 0; i < n; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n"
                     "v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 %0, %1, %2\n"
                     : "=s"(m) : "v"(a0), "v"(b) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)m;
}
__global__ __launch_bounds__(256) void k_dsadd(float *out, int n)
{
    __shared__ unsigned h[4096];
    for (int k = threadIdx.x; k < 4096; k += 256) h[k] = 0;
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u) >> 20;  // pseudo-random bin 0..4095
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            atomicAdd(&h[idx], 1u);
            idx = (idx * 1664525u + 1013904223u) & 4095u;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)h[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_dsadd_third(float *out, int n)
{
    __shared__ unsigned h[4096];
    for (int k = threadIdx.x; k < 4096; k += 256) h[k] = 0;
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u) >> 20;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if ((idx & 3u) == 0u) atomicAdd(&h[idx], 1u);   // ~1/4 of the lanes active
            idx = (idx * 1664525u + 1013904223u) & 4095u;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = (float)h[threadIdx.x];
}

typedef void (*kern_t)(float *, int);
static void run(const char *name, kern_t k, float *out, double extra_per_iter)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, ITER);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD = 8 waves * ITER * 8
    double winst = 8.0 * ITER * 8.0;
    double ns_per = ms * 1e6 / winst;
    printf("%-18s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", name, ms, ns_per, ns_per * 2.4);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
#define R(k) run(#k, k, out, 0)
    R(k_fma32); R(k_mul32); R(k_add32); R(k_subu32); R(k_andb32); R(k_mullo); R(k_cvt_f32_i32); R(k_cvt_i32_f32); R(k_fract32);
    R(k_floor32); R(k_sqrt32); R(k_rsq32); R(k_cmp32); R(k_pkmul32); R(k_pkadd32); R(k_mul64); R(k_add64); R(k_rsq64);
    R(k_cvt_f64_i32_fake); R(k_dsadd); R(k_dsadd_third);
    return 0;
}
