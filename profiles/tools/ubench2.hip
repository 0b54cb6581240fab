// VALU issue-cost calibration for gfx950, second edition (diagnostic, not part of the product).
//
// Round 2's table (ubench.hip) timed 0.3 ms kernels with HIP events and priced cycles at a nominal 2.4 GHz: a fixed
// launch + clock-ramp offset of ~0.085 ms sat in every row (2.77 "cycles" for v_mul_f32 instead of the 2 the SIMD-32
// needs for a wave64), which made the issue-slot model of the tile kernel come out above 1.  Here
//   * every kernel runs >= 20 ms (ITER is scaled per instruction class),
//   * cycles are counted INSIDE the kernel with s_memtime (shader-clock cycles) around the loop of every wave, the
//     effective clock comes from s_memrealtime (100 MHz) over the same interval -- nothing depends on a nominal clock,
//   * workgroups are sized (1024 threads, 64 KiB of LDS) so that exactly two fit a CU: 8 waves on every SIMD for the
//     whole kernel -- a first version with 2048 x 256-thread workgroups left the SIMDs with 4..8 waves each at any
//     time; every wave records HW_ID / XCC_ID, and the cost of one wave-instruction to its SIMD is
//     (last end - first start of the waves of that SIMD) / wave-instructions issued there,
//   * occupancy rows (1, 2, 4 waves per SIMD) show what a lone wave sustains (the guide: 4 cycles for v_fma_f32),
//   * "mix" rows replay the instruction mix of the RDF tile kernel's pair chain and compare the measured cycles with
//     the sum of the per-class costs (is pricing a mixed stream with per-class costs legitimate?).
// Build + run:  hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench2 profiles/tools/ubench2.hip && /tmp/ubench2
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

struct Stamp { unsigned long long t0, t1, real; unsigned hw, xcc; };

#define STAMP_BEGIN                                                   \
    const unsigned long long T0_ = __builtin_amdgcn_s_memtime();      \
    const unsigned long long R0_ = __builtin_amdgcn_s_memrealtime();
// HW_ID (hwreg 4): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID (hwreg 20): xcc_id[3:0]
#define STAMP_END                                                                              \
    const unsigned long long T1_ = __builtin_amdgcn_s_memtime();                               \
    const unsigned long long R1_ = __builtin_amdgcn_s_memrealtime();                           \
    if ((threadIdx.x & 63) == 0) {                                                             \
        Stamp &S_ = st[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];                   \
        S_.t0 = T0_; S_.t1 = T1_; S_.real = R1_ - R0_;                                         \
        S_.hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));                        \
        S_.xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));                      \
    }

#define UNARY(name, ins, T, cons)                                                                                       \
    __global__ __launch_bounds__(1024) void name(float *out, Stamp *st, int n)                                          \
    {                                                                                                                  \
        T a0 = threadIdx.x + 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8;                                 \
        STAMP_BEGIN                                                                                                    \
        for (int i = 0; i < n; i++) {                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < 8; u_++) asm volatile(ins " %0, %0\n " ins " %1, %1\n " ins " %2, %2\n " ins " %3, %3\n " ins " %4, %4\n " ins      \
                             " %5, %5\n " ins " %6, %6\n " ins " %7, %7\n"                                             \
                         : "+" cons(a0), "+" cons(a1), "+" cons(a2), "+" cons(a3), "+" cons(a4), "+" cons(a5),         \
                           "+" cons(a6), "+" cons(a7));                                                                \
        }                                                                                                              \
        STAMP_END                                                                                                      \
        out[blockIdx.x * 1024 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }
#define BINARY(name, ins, T, cons)                                                                                      \
    __global__ __launch_bounds__(1024) void name(float *out, Stamp *st, int n)                                          \
    {                                                                                                                  \
        T a0 = threadIdx.x + 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8, b = 3;                          \
        STAMP_BEGIN                                                                                                    \
        for (int i = 0; i < n; i++) {                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < 8; u_++) asm volatile(ins " %0, %0, %8\n " ins " %1, %1, %8\n " ins " %2, %2, %8\n " ins " %3, %3, %8\n " ins       \
                             " %4, %4, %8\n " ins " %5, %5, %8\n " ins " %6, %6, %8\n " ins " %7, %7, %8\n"            \
                         : "+" cons(a0), "+" cons(a1), "+" cons(a2), "+" cons(a3), "+" cons(a4), "+" cons(a5),         \
                           "+" cons(a6), "+" cons(a7)                                                                  \
                         : cons(b));                                                                                   \
        }                                                                                                              \
        STAMP_END                                                                                                      \
        out[blockIdx.x * 1024 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }
#define TERNARY(name, ins, T, cons)                                                                                     \
    __global__ __launch_bounds__(1024) void name(float *out, Stamp *st, int n)                                          \
    {                                                                                                                  \
        T a0 = threadIdx.x + 1, a1 = 2, a2 = 3, a3 = 4, a4 = 5, a5 = 6, a6 = 7, a7 = 8, b = 3, c = 1;                   \
        STAMP_BEGIN                                                                                                    \
        for (int i = 0; i < n; i++) {                                                                                  \
            _Pragma("unroll") for (int u_ = 0; u_ < 8; u_++) asm volatile(ins " %0, %0, %8, %9\n " ins " %1, %1, %8, %9\n " ins " %2, %2, %8, %9\n " ins                \
                             " %3, %3, %8, %9\n " ins " %4, %4, %8, %9\n " ins " %5, %5, %8, %9\n " ins                \
                             " %6, %6, %8, %9\n " ins " %7, %7, %8, %9\n"                                              \
                         : "+" cons(a0), "+" cons(a1), "+" cons(a2), "+" cons(a3), "+" cons(a4), "+" cons(a5),         \
                           "+" cons(a6), "+" cons(a7)                                                                  \
                         : cons(b), cons(c));                                                                          \
        }                                                                                                              \
        STAMP_END                                                                                                      \
        out[blockIdx.x * 1024 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                          \
    }

TERNARY(k_fma32, "v_fma_f32", float, "v")
BINARY(k_mul32, "v_mul_f32", float, "v")
BINARY(k_add32, "v_add_f32", float, "v")
BINARY(k_min32, "v_min_f32", float, "v")
BINARY(k_subu32, "v_sub_u32", unsigned, "v")
BINARY(k_andb32, "v_and_b32", unsigned, "v")
BINARY(k_lshl, "v_lshlrev_b32", unsigned, "v")
TERNARY(k_lshladd, "v_lshl_add_u32", unsigned, "v")
BINARY(k_mullo, "v_mul_lo_u32", unsigned, "v")
UNARY(k_mov, "v_mov_b32", unsigned, "v")
UNARY(k_cvt_f32_i32, "v_cvt_f32_i32", float, "v")
UNARY(k_cvt_i32_f32, "v_cvt_i32_f32", float, "v")
UNARY(k_cvt_f32_u32, "v_cvt_f32_u32", float, "v")
UNARY(k_fract32, "v_fract_f32", float, "v")
UNARY(k_floor32, "v_floor_f32", float, "v")
UNARY(k_rndne32, "v_rndne_f32", float, "v")
UNARY(k_sqrt32, "v_sqrt_f32", float, "v")
UNARY(k_rsq32, "v_rsq_f32", float, "v")
UNARY(k_rcp32, "v_rcp_f32", float, "v")
BINARY(k_pkmul32, "v_pk_mul_f32", double, "v")
TERNARY(k_pkfma32, "v_pk_fma_f32", double, "v")
BINARY(k_mul64, "v_mul_f64", double, "v")
BINARY(k_add64, "v_add_f64", double, "v")
TERNARY(k_fma64, "v_fma_f64", double, "v")
UNARY(k_rsq64, "v_rsq_f64", double, "v")
UNARY(k_sqrt64, "v_sqrt_f64", double, "v")
UNARY(k_rcp64, "v_rcp_f64", double, "v")
UNARY(k_rndne64, "v_rndne_f64", double, "v")
// candidates for cheaper forms of the half-rate steps of the pair chain (min, shift-add, compare)
BINARY(k_max32, "v_max_f32", float, "v")
TERNARY(k_med3, "v_med3_f32", float, "v")
BINARY(k_minu32, "v_min_u32", unsigned, "v")
BINARY(k_mini32, "v_min_i32", unsigned, "v")
BINARY(k_lshr, "v_lshrrev_b32", unsigned, "v")
BINARY(k_ashr, "v_ashrrev_i32", unsigned, "v")
TERNARY(k_bfe, "v_bfe_u32", unsigned, "v")
BINARY(k_or, "v_or_b32", unsigned, "v")
BINARY(k_xor, "v_xor_b32", unsigned, "v")
BINARY(k_addu32, "v_add_u32", unsigned, "v")
TERNARY(k_add3, "v_add3_u32", unsigned, "v")
TERNARY(k_madu24, "v_mad_u32_u24", unsigned, "v")
BINARY(k_mulu24, "v_mul_u32_u24", unsigned, "v")
TERNARY(k_and_or, "v_and_or_b32", unsigned, "v")
TERNARY(k_perm, "v_perm_b32", unsigned, "v")
BINARY(k_ldexp, "v_ldexp_f32", float, "v")
UNARY(k_cvt_u32_f32, "v_cvt_u32_f32", float, "v")
UNARY(k_trunc32, "v_trunc_f32", float, "v")
BINARY(k_mul_legacy, "v_mul_legacy_f32", float, "v")

__global__ __launch_bounds__(1024) void k_cmp32(float *out, Stamp *st, int n)
{
    float a0 = threadIdx.x, b = 3;
    unsigned long long m = 0;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n"
                     "v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 %0, %1, %2\n"
                     : "=s"(m) : "v"(a0), "v"(b) : "vcc");
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = (float)m;
}
// VOP3 compare into an SGPR pair with the |x| modifier, as the tile kernel's `unsafe` test is encoded
__global__ __launch_bounds__(1024) void k_cmp32_sgpr(float *out, Stamp *st, int n)
{
    float a0 = threadIdx.x, b = 3;
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++) asm volatile("v_cmp_nlt_f32 %0, |%4|, %5\n v_cmp_nlt_f32 %1, |%4|, %5\n v_cmp_nlt_f32 %2, |%4|, %5\n v_cmp_nlt_f32 %3, |%4|, %5\n"
                     "v_cmp_nlt_f32 %0, |%4|, %5\n v_cmp_nlt_f32 %1, |%4|, %5\n v_cmp_nlt_f32 %2, |%4|, %5\n v_cmp_nlt_f32 %3, |%4|, %5\n"
                     : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "v"(b));
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = (float)(m0 + m1 + m2 + m3);
}
// the fixed-point bin form's compare: 16-bit, as SDWA (what the compiler emits) and as v_cmp_lt_u16
__global__ __launch_bounds__(1024) void k_cmp_sdwa(float *out, Stamp *st, int n)
{
    unsigned a0 = threadIdx.x * 977u;
    unsigned b = 77u;
    asm volatile("s_mov_b32 %0, %0" : "+s"(b));
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++)
            asm volatile("v_cmp_lt_u32_sdwa %0, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n v_cmp_lt_u32_sdwa %1, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n"
                         "v_cmp_lt_u32_sdwa %2, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n v_cmp_lt_u32_sdwa %3, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n"
                         "v_cmp_lt_u32_sdwa %0, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n v_cmp_lt_u32_sdwa %1, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n"
                         "v_cmp_lt_u32_sdwa %2, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n v_cmp_lt_u32_sdwa %3, %4, %5 src0_sel:WORD_0 src1_sel:DWORD\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "s"(b));
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = (float)(m0 + m1 + m2 + m3);
}
__global__ __launch_bounds__(1024) void k_cmp_u16(float *out, Stamp *st, int n)
{
    unsigned a0 = threadIdx.x * 977u;
    unsigned b = 77u;
    asm volatile("s_mov_b32 %0, %0" : "+s"(b));
    unsigned long long m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++)
            asm volatile("v_cmp_lt_u16_e64 %0, %4, %5\n v_cmp_lt_u16_e64 %1, %4, %5\n v_cmp_lt_u16_e64 %2, %4, %5\n v_cmp_lt_u16_e64 %3, %4, %5\n"
                         "v_cmp_lt_u16_e64 %0, %4, %5\n v_cmp_lt_u16_e64 %1, %4, %5\n v_cmp_lt_u16_e64 %2, %4, %5\n v_cmp_lt_u16_e64 %3, %4, %5\n"
                         : "=s"(m0), "=s"(m1), "=s"(m2), "=s"(m3) : "v"(a0), "s"(b));
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = (float)(m0 + m1 + m2 + m3);
}
__global__ __launch_bounds__(1024) void k_cndmask(float *out, Stamp *st, int n)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, b = 3;
    unsigned long long msk = 0x5555555555555555ull;
    asm volatile("s_mov_b64 %0, %0" : "+s"(msk));
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++)
            asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n"
                         "v_cndmask_b32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n"
                         "v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(msk));
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// the same min / shift with IEEE mode off (amdgpu-ieee=false: no sNaN quieting pass) -- is their half rate a mode effect?
// SGPR spill traffic: v_writelane_b32 / v_readlane_b32 pairs (what an SGPR spill costs the vector pipe)
__global__ __launch_bounds__(1024) void k_rwlane(float *out, Stamp *st, int n)
{
    unsigned v0 = threadIdx.x, v1 = 1;
    unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u_ = 0; u_ < 8; u_++) asm volatile("v_writelane_b32 %4, %0, 1\n v_writelane_b32 %5, %1, 2\n v_writelane_b32 %4, %2, 3\n v_writelane_b32 %5, %3, 4\n"
                     "v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %5, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %5, 4\n"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1));
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = (float)(v0 + v1 + s0 + s1 + s2 + s3);
}
// ---- the RDF tile kernel's pair chain (ZF form, always-add), 8 independent pairs per iteration -----------------------
// per pair: 2 v_sub_u32, 2 v_cvt_f32_i32, v_sub_f32, 3 v_mul_f32, 2 v_fma_f32, v_sqrt_f32, v_min_f32, v_fract_f32,
//           v_add_f32 (-1/2), v_cmp (|.| modifier) + s_or_b64, v_cvt_i32_f32, v_lshl_add_u32   = 17 VALU + 1 SALU
// LDS: mode 0 no atomic, 1 one ds_add_u32 per pair into a 2310-bin histogram (addresses from the chain)
#define PAIR(qx, qy, qz)                                                                                               \
    asm volatile("v_sub_u32 %0, %6, %9\n v_sub_u32 %1, %7, %10\n v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n"        \
                 "v_sub_f32 %2, %8, %11\n v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %0, %0, %12\n"       \
                 "v_fma_f32 %0, %1, %13, %0\n v_fma_f32 %0, %2, %2, %0\n v_sqrt_f32 %0, %0\n v_min_f32 %0, %0, %14\n"  \
                 "v_fract_f32 %1, %0\n v_add_f32 %1, -0.5, %1\n v_cmp_nlt_f32 %4, |%1|, %15\n s_or_b64 %5, %5, %4\n"   \
                 "v_cvt_i32_f32 %0, %0\n v_lshl_add_u32 %3, %0, 2, %16\n"                                              \
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(addr), "=&s"(m), "+s"(macc)                                  \
                 : "v"(qx), "v"(qy), "v"(qz), "v"(cx), "v"(cy), "v"(cz), "v"(s3), "v"(s4), "v"(clampv), "v"(hmg),      \
                   "v"(hbase));

template <int LDSMODE>
__global__ __launch_bounds__(1024) void k_mix(float *out, Stamp *st, int n)
{
    __shared__ unsigned hist[2310 + 32];
    for (int k = threadIdx.x; k < 2342; k += 1024) hist[k] = 0;
    __syncthreads();
    unsigned cx = threadIdx.x * 2654435761u, cy = cx * 1664525u + 1013904223u;
    float cz = (float)(threadIdx.x & 63) * 3.0f;
    unsigned qx = (blockIdx.x * 1024u + threadIdx.x) * 40503u + 17u, qy = qx * 22695477u + 1u;
    float qz = 100.0f;
    const float s3 = 2.5e-13f, s4 = 2.6e-13f, clampv = 2310.5f + (float)(threadIdx.x & 31), hmg = 0.4992f;
    const unsigned hbase = (unsigned)(size_t)hist;
    float t0, t1, t2;
    unsigned addr;
    unsigned long long m, macc = 0;
    float sink = 0.f;
    STAMP_BEGIN
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            PAIR(qx, qy, qz)
            if (LDSMODE == 1) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
            qx += 0x9e3779b9u; qy += 0x7f4a7c15u; qz += 1.0f;      // (3 more VALU per pair: counted in the model below)
        }
    }
    STAMP_END
    out[blockIdx.x * 1024 + threadIdx.x] = sink + (float)macc + t0 + t1 + t2 + (float)addr;
}

typedef void (*kern_t)(float *, Stamp *, int);

struct Row { const char *name; double cyc, ghz, ms; };

// wgs workgroups of `threads` threads with `lds` bytes of dynamic LDS each (64 KiB x 1024 threads: exactly two per CU,
// 8 waves on every SIMD for the whole kernel).  Per SIMD (XCC, SE, SH, CU, SIMD from HW_ID): cycles from the first
// wave's start to the last wave's end / wave-instructions issued there; the row reports the median over the SIMDs.
static Row run(const char *name, kern_t k, float *out, Stamp *d_st, int wgs, int threads, size_t lds, double inst_per_iter,
               int iters)
{
    const int wpw = threads / 64;
    std::vector<Stamp> h((size_t)wgs * wpw);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), lds, 0, out, d_st, 64);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), lds, 0, out, d_st, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h.data(), d_st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<const Stamp *>> simd;
    std::vector<double> ghz;
    for (auto &s : h) {
        const unsigned key = ((s.xcc & 15u) << 16) | (s.hw & 0xff30u);      // xcc | se, sh, cu | simd
        simd[key].push_back(&s);
        ghz.push_back(s.real ? (double)(s.t1 - s.t0) / (double)s.real * 0.1 : 0.0);
    }
    std::vector<double> per;
    int wmin = 1 << 30, wmax = 0;
    for (auto &kv : simd) {
        unsigned long long lo = ~0ull, hi = 0;
        for (auto *s : kv.second) { lo = std::min(lo, s->t0); hi = std::max(hi, s->t1); }
        per.push_back((double)(hi - lo) / ((double)iters * inst_per_iter * kv.second.size()));
        wmin = std::min(wmin, (int)kv.second.size()); wmax = std::max(wmax, (int)kv.second.size());
    }
    std::sort(per.begin(), per.end());
    std::sort(ghz.begin(), ghz.end());
    const double med = per[per.size() / 2], g = ghz[ghz.size() / 2];
    printf("%-16s %4d SIMDs, %d..%d waves each  %7.2f ms  cycles/wave-instr/SIMD %7.3f  (SIMD p5..p95 %6.3f .. %6.3f)  "
           "by wall clock %6.3f  clock %.3f GHz\n", name, (int)simd.size(), wmin, wmax, ms, med, per[per.size() / 20],
           per[per.size() * 19 / 20], ms * 1e-3 * g * 1e9 / ((double)iters * inst_per_iter * (double)h.size() / 1024.0), g);
    fflush(stdout);
    Row r = {name, med, g, ms};
    return r;
}

int main(int argc, char **argv)
{
    const double target_ms = argc > 1 ? atof(argv[1]) : 25.0;
    float *out; Stamp *d_st;
    hipMalloc(&out, 512 * 1024 * 4 * 2);
    hipMalloc(&d_st, 512 * 16 * sizeof(Stamp));
    const size_t LDS2 = 64 * 1024;       // two workgroups per CU
    // iterations for ~target_ms at `c` cycles per instruction, `inst` instructions per iteration, wps waves per SIMD, 2.1 GHz
    auto iters_for = [&](double c, double inst, int wps) { return (int)(target_ms * 1e-3 * 2.1e9 / (c * inst * wps)); };
    printf("# cycles (s_memtime) per wave-instruction per SIMD; 512 workgroups x 1024 threads x 64 KiB LDS = two per CU = 8 waves on\n"
           "# every SIMD for the whole kernel; >= %.0f ms per kernel; 64 independent instructions (8 accumulators) per loop trip\n", target_ms);
#define R(k, c) run(#k, k, out, d_st, 512, 1024, LDS2, 64.0, iters_for(c, 64.0, 8))
    Row fma = R(k_fma32, 2); Row mul = R(k_mul32, 2); Row add = R(k_add32, 2); Row mn = R(k_min32, 2);
    Row subu = R(k_subu32, 2); R(k_andb32, 2); R(k_lshl, 2); Row lsa = R(k_lshladd, 2); R(k_mullo, 4); R(k_mov, 2);
    Row cvtfi = R(k_cvt_f32_i32, 4); Row cvtif = R(k_cvt_i32_f32, 4); R(k_cvt_f32_u32, 4);
    Row fract = R(k_fract32, 4); R(k_floor32, 4); R(k_rndne32, 4);
    Row sq = R(k_sqrt32, 8); R(k_rsq32, 8); R(k_rcp32, 8);
    Row cmp = R(k_cmp32, 4); Row cmps = R(k_cmp32_sgpr, 4); R(k_cndmask, 2); R(k_rwlane, 4);
    R(k_pkmul32, 4); R(k_pkfma32, 4); R(k_mul64, 4); R(k_add64, 4); R(k_fma64, 4); R(k_rndne64, 4);
    R(k_rsq64, 16); R(k_sqrt64, 16); R(k_rcp64, 16);
    printf("# candidates for cheaper forms of the half-rate steps\n");
    R(k_max32, 4); R(k_med3, 4); R(k_minu32, 2); R(k_mini32, 2); R(k_lshr, 4); R(k_ashr, 4); R(k_bfe, 4); R(k_or, 2); R(k_xor, 2);
    R(k_addu32, 2); R(k_add3, 2); R(k_madu24, 4); R(k_mulu24, 4); R(k_and_or, 2); R(k_perm, 4); R(k_ldexp, 4); R(k_cvt_u32_f32, 4);
    R(k_trunc32, 4); R(k_mul_legacy, 2); R(k_cmp_sdwa, 4); R(k_cmp_u16, 4);
    printf("# occupancy: 256 workgroups of 64 x wps x 4 threads with 100 KiB LDS = one per CU, wps waves per SIMD\n");
    for (int wps : {1, 2, 4}) {
        run("k_mul32", k_mul32, out, d_st, 256, 256 * wps, 100 * 1024, 64.0, iters_for(wps >= 4 ? 2.0 : 8.0 / wps, 64.0, wps));
        run("k_sqrt32", k_sqrt32, out, d_st, 256, 256 * wps, 100 * 1024, 64.0, iters_for(wps >= 4 ? 8.0 : 16.0 / wps, 64.0, wps));
    }
    printf("# the tile kernel's pair chain: 8 pairs per loop trip; VALU per pair = 17 chain + 3 bookkeeping (2 v_add_u32, 1 v_add_f32)\n");
    // model: sum of the per-class costs measured above
    const double model = 2 * subu.cyc + 2 * cvtfi.cyc + add.cyc + 3 * mul.cyc + 2 * fma.cyc + sq.cyc + mn.cyc + fract.cyc + add.cyc +
                         cmps.cyc + cvtif.cyc + lsa.cyc          /* the chain */
                         + 2 * subu.cyc + add.cyc;                /* bookkeeping: 2 int adds, f32 add */
    (void)cmp;
    Row mix0 = run("k_mix<no LDS>", k_mix<0>, out, d_st, 512, 1024, LDS2 - 16 * 1024, 8.0, iters_for(50, 8.0, 8));
    Row mix1 = run("k_mix<ds_add>", k_mix<1>, out, d_st, 512, 1024, LDS2 - 16 * 1024, 8.0, iters_for(60, 8.0, 8));
    printf("pair chain: measured %.2f cycles per wave-level pair without LDS atomics, %.2f with one ds_add_u32 per pair; "
           "sum of the per-class VALU costs %.2f\n",
           mix0.cyc, mix1.cyc, model);
    printf("model/measured (no LDS) = %.3f\n", model / mix0.cyc);
    return 0;
}
