#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
__device__ __forceinline__ float sqrt_rn(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    if (!(x > 0.f) || !(x < 3.0e38f)) return s;
    const float sd = __uint_as_float(__float_as_uint(s) - 1u);
    const float su = __uint_as_float(__float_as_uint(s) + 1u);
    const float vp = fmaf(-sd, s, x);
    const float vs = fmaf(-su, s, x);
    if (vp <= 0.f) s = sd;
    if (vs > 0.f) s = su;
    return s;
}
__global__ void k(const float* x, float* a, float* b, float* c, float* q, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = sqrt_rn(x[i]); b[i] = sqrtf(x[i]); c[i] = (float)sqrt((double)x[i]); q[i] = x[i] / (x[(i + 1) % n] + 1.0f); }
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) { float e = (rand() % 40) - 30; h[i] = ((rand() & 0xFFFFFF) / 16777216.0f + 0.5f) * powf(2.f, e); }
    float *dx, *da, *db, *dc, *dq;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dq, n * 4);
    hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, dc, dq, n);
    float *ra = (float*)malloc(n * 4), *rb = (float*)malloc(n * 4), *rc = (float*)malloc(n * 4), *rq = (float*)malloc(n * 4);
    hipMemcpy(ra, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rb, db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rc, dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(rq, dq, n * 4, hipMemcpyDeviceToHost);
    int ea = 0, eb = 0, ec = 0, eq = 0;
    for (int i = 0; i < n; ++i) {
        volatile float r = sqrtf(h[i]);
        volatile float qq = h[i] / (h[(i + 1) % n] + 1.0f);
        ea += memcmp((void*)&r, &ra[i], 4) != 0; eb += memcmp((void*)&r, &rb[i], 4) != 0; ec += memcmp((void*)&r, &rc[i], 4) != 0;
        eq += memcmp((void*)&qq, &rq[i], 4) != 0;
    }
    printf("mismatches of %d: sqrt_rn %d, sqrtf %d, (float)sqrt(double) %d, div %d\n", n, ea, eb, ec, eq);
    return 0;
}
