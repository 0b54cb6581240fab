// MFMA go / no-go microbenchmark for the all-pairs RDF tile kernel (round-3 review, item 3a; diagnostic, not product).
//
// Question: does moving the distance part of the pair chain onto the matrix pipe (v_mfma_f32_16x16x4_f32: d^2 of a 16 x 16
// tile of compact atom groups in one instruction, C preloaded with |a|^2, A = (-2 a, 1), B = (b, |b|^2)) buy >= 1.25 x on
// the pair chain when 40 % of the group pairs need BOTH candidate images (a second MFMA and a second binning tail)?
//
// Three kernels with the launch shape of the product kernel (256-thread workgroups, 30 KB of LDS each: five per CU), all
// looping over partners staged in LDS, all ending in the product's binning tail (sqrt, min, fract, compare, convert,
// shift-add, LDS atomic into a 2310-bin histogram + 32 trash words, scalar OR of the "unsafe" flags):
//   valu   the product's ZF chain: two u32 differences, two conversions, an f32 difference, 3 mul + 2 fma (17 VALU / pair)
//   mfma1  one MFMA per 16 x 16 tile + 4 tails per lane (single image)
//   mfma14 the same with a second MFMA + tails for 2 of every 5 tiles (the 40 % both-images share of DESIGN 4.1)
// Reported: nanoseconds and shader cycles (s_memtime) per wave-level pair and SIMD, and the ratios.  Kernels run >= 20 ms.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_mfma profiles/tools/ubench_mfma.hip && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

constexpr int NB = 2310, TRASH = 32, NPART = 512, THREADS = 256;
typedef float float4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool tail(unsigned *hist, float t, float clampv, float hmg)
{
    float q = __builtin_amdgcn_sqrtf(t);
    q = __builtin_fminf(q, clampv);
    const bool unsafe = !(fabsf(__builtin_amdgcn_fractf(q) - 0.5f) < hmg);
    atomicAdd(&hist[(int)q], 1u);
    return unsafe;
}

// partners: uint4 (ux, uy, uz, zf bits) in LDS; centre per lane: two atoms (as the product)
__global__ __launch_bounds__(THREADS, 5) void k_valu(const uint4 *__restrict__ src, unsigned *out, unsigned long long *cyc, int iters)
{
    extern __shared__ __align__(16) unsigned char lds[];
    uint4 *tq = reinterpret_cast<uint4 *>(lds);
    unsigned *hist = reinterpret_cast<unsigned *>(tq + NPART);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < NPART; k += THREADS) tq[k] = src[k];
    for (int k = tid; k < NB + TRASH; k += THREADS) hist[k] = 0u;
    __syncthreads();
    const uint4 ca = src[NPART + 2 * lane], cb = src[NPART + 2 * lane + 1];
    const float zaf = __uint_as_float(ca.w), zbf = __uint_as_float(cb.w);
    const float s3 = 2.9e-13f, s4 = 2.9e-13f, hmg = 0.4992f;
    const float clampv = (float)NB + 0.5f + (float)(lane & 31);
    unsigned any = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        for (int j0 = 4 * wave; j0 < NPART; j0 += 16) {
            uint4 qj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) qj[u] = tq[j0 + u];
            bool f = false;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                {
                    const float fx = (float)(int)(qj[u].x - ca.x), fy = (float)(int)(qj[u].y - ca.y), dz = __uint_as_float(qj[u].w) - zaf;
                    f |= tail(hist, fmaf(dz, dz, fmaf(fy * fy, s4, fx * fx * s3)), clampv, hmg);
                }
                {
                    const float fx = (float)(int)(qj[u].x - cb.x), fy = (float)(int)(qj[u].y - cb.y), dz = __uint_as_float(qj[u].w) - zbf;
                    f |= tail(hist, fmaf(dz, dz, fmaf(fy * fy, s4, fx * fx * s3)), clampv, hmg);
                }
            }
            any += __builtin_amdgcn_readfirstlane(__ballot(f) != 0 ? 1u : 0u);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    __syncthreads();
    unsigned s = any;
    for (int k = tid; k < NB; k += THREADS) s += hist[k];
    out[blockIdx.x * THREADS + tid] = s;
}

// groups of 16 atoms: B operand per lane = component (lane / 16) of partner (lane % 16) of the group: one float per lane
// per group from LDS; A operand and C (|a|^2 of the lane's four rows) fixed per wave (16 centres)
template <int BOTH_OF_5>
__global__ __launch_bounds__(THREADS, 5) void k_mfma(const float *__restrict__ srcf, unsigned *out, unsigned long long *cyc, int iters)
{
    extern __shared__ __align__(16) unsigned char lds[];
    float *tb = reinterpret_cast<float *>(lds);                       // [NPART / 16][64] B operands, group-major
    unsigned *hist = reinterpret_cast<unsigned *>(tb + NPART * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < NPART * 4; k += THREADS) tb[k] = srcf[k];
    for (int k = tid; k < NB + TRASH; k += THREADS) hist[k] = 0u;
    __syncthreads();
    const float a = srcf[NPART * 4 + lane];                           // A[row = lane % 16][k = lane / 16]
    float4v c0, c1;
#pragma unroll
    for (int i = 0; i < 4; i++) { c0[i] = srcf[NPART * 4 + 64 + 4 * (lane / 16) + i]; c1[i] = c0[i] + 3.0f; }
    const float hmg = 0.4992f;
    const float clampv = (float)NB + 0.5f + (float)(lane & 31);
    unsigned any = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // a wave meets the groups wave, wave + 4, ...: 32 groups = 512 partners, 16 x 16 pairs each (the valu kernel: 128 x 512 per
        // workgroup = 64 x 512 x 2 per wave ... here 16 centres per wave: 4 x fewer pairs per trip, accounted for below)
        for (int g = wave; g < NPART / 16; g += 4) {
            const float b = tb[g * 64 + lane];
            bool f = false;
            float4v d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++) f |= tail(hist, d[i], clampv, hmg);
            if (BOTH_OF_5 > 0 && (g % 5) < BOTH_OF_5) {               // (wave-uniform: the second candidate image of the tile)
                float4v e = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; i++) f |= tail(hist, e[i], clampv, hmg);
            }
            any += __builtin_amdgcn_readfirstlane(__ballot(f) != 0 ? 1u : 0u);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) atomicAdd(cyc, t1 - t0);
    __syncthreads();
    unsigned s = any;
    for (int k = tid; k < NB; k += THREADS) s += hist[k];
    out[blockIdx.x * THREADS + tid] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

int main()
{
    const int blocks = 256 * 5 * 4;         // four rounds of five workgroups per CU
    std::vector<uint4> hq(NPART + 128);
    srand(7);
    for (auto &q : hq) {
        q.x = (unsigned)rand() * 2u; q.y = (unsigned)rand() * 2u; q.z = (unsigned)rand() * 2u;
        const float z = (float)(rand() % 4000) * 0.5f;
        q.w = *reinterpret_cast<const unsigned *>(&z);
    }
    std::vector<float> hf(NPART * 4 + 64 + 64);
    for (auto &v : hf) v = (float)(rand() % 2000) * 0.01f;
    for (int g = 0; g < NPART / 16; g++)
        for (int j = 0; j < 16; j++) hf[g * 64 + 48 + j] = 1.0e6f + (float)(rand() % 4000000);      // |b|^2: d^2 up to ~ 2310^2
    uint4 *dq; float *df; unsigned *dout; unsigned long long *dcyc;
    CK(hipMalloc(&dq, hq.size() * sizeof(uint4))); CK(hipMalloc(&df, hf.size() * sizeof(float)));
    CK(hipMalloc(&dout, (size_t)blocks * THREADS * sizeof(unsigned))); CK(hipMalloc(&dcyc, 8));
    CK(hipMemcpy(dq, hq.data(), hq.size() * sizeof(uint4), hipMemcpyHostToDevice));
    CK(hipMemcpy(df, hf.data(), hf.size() * sizeof(float), hipMemcpyHostToDevice));
    const size_t lds = 30 * 1024;            // as the product: five workgroups per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch, double wave_pairs_per_block_iter, int iters) {
        double best_ms = 1e30; unsigned long long cyc = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(dcyc, 0, 8));
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_ms) { best_ms = ms; CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost)); }
        }
        const double wp = wave_pairs_per_block_iter * iters * blocks;       // wave-level pairs of the launch
        const double ns_simd = best_ms * 1e6 * 1024.0 / wp;                  // per SIMD (1024 of them)
        // s_memtime runs at 100 MHz on this part when it is the constant clock; report wave-resident cycles too
        const double res = (double)cyc / ((double)blocks * 4.0) / (wave_pairs_per_block_iter / 4.0 * iters);
        printf("%-8s %8.2f ms   %7.3f ns per wave-level pair and SIMD   (%.1f s_memtime ticks per wave-level pair of a resident wave)\n",
               name, best_ms, ns_simd, res);
        return ns_simd;
    };
    // valu: per workgroup and iteration 128 centres x 512 partners = 65536 pairs = 1024 wave-level pairs
    const double v = run("valu", [&](int it) { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(THREADS), lds, 0, dq, dout, dcyc, it); }, 1024.0, 180);
    // mfma: per workgroup and iteration 4 waves x 8 groups x 256 pairs = 8192 pairs = 128 wave-level pairs (+ 40 % second images)
    const double m1 = run("mfma1", [&](int it) { hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(THREADS), lds, 0, df, dout, dcyc, it); }, 128.0, 1440);
    const double m14 = run("mfma14", [&](int it) { hipLaunchKernelGGL(k_mfma<2>, dim3(blocks), dim3(THREADS), lds, 0, df, dout, dcyc, it); }, 128.0, 1440);
    printf("per PAIR (second images are work, not pairs): valu / mfma1 = %.3f x, valu / mfma14 = %.3f x\n", v / m1, v / m14);
    printf("go / no-go (round-3 review: >= 1.25 x at the 40 %% both-images share): %s\n", v / m14 >= 1.25 ? "GO" : "NO-GO");
    return 0;
}
