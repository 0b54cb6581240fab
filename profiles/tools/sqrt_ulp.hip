// Exhaustive accuracy check of v_sqrt_f32 (__builtin_amdgcn_sqrtf) on gfx950: every float in
// [1, 4) (the relative error pattern of sqrt repeats with period 4), against the correctly
// rounded result computed in double.  Backs the guard constant of the RDF fast path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

__global__ void k(unsigned long long *worst, double *maxrel)
{
    unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;          // 2^24 mantissas x 2 exponents
    unsigned bits = 0x3f800000u + idx;                             // [1, 4)
    if (bits >= 0x40800000u) return;
    float x = __uint_as_float(bits);
    float got = __builtin_amdgcn_sqrtf(x);
    double ex = sqrt((double)x);
    float cr = (float)ex;                                          // correctly rounded (double sqrt is exact enough)
    int du = abs((int)(__float_as_uint(got) - __float_as_uint(cr)));
    double rel = fabs((double)got - ex) / ex;
    atomicMax(worst, (unsigned long long)du);
    // max relative error via integer compare on the bits of a positive double
    atomicMax((unsigned long long *)maxrel, (unsigned long long)__double_as_longlong(rel));
}

int main()
{
    unsigned long long *w; double *m;
    hipMalloc(&w, 8); hipMalloc(&m, 8);
    hipMemset(w, 0, 8); hipMemset(m, 0, 8);
    hipLaunchKernelGGL(k, dim3((1u << 25) / 256), dim3(256), 0, 0, w, m);
    unsigned long long hw; double hm;
    hipMemcpy(&hw, w, 8, hipMemcpyDeviceToHost); hipMemcpy(&hm, m, 8, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 over all floats in [1,4): max |result - correctly rounded| = %llu ulp, max relative error = %.4g (= %.3f x 2^-24)\n",
           hw, hm, hm * 16777216.0);
    return 0;
}
