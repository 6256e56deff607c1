#include <hip/hip_runtime.h>
__global__ void k(double* p) {
    double v = p[threadIdx.x], w = v * 2.0, acc = 1.0;
    double b;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(v));
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(w));
    p[threadIdx.x] = b + acc;
}
int main() {
    double h[64], *d; for (int i = 0; i < 64; ++i) h[i] = i + 0.5;
    hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    // expect: b = v[row_base+1]; acc = 1 + v[row_base+2] * (2*v[lane])
    int bad = 0;
    for (int i = 0; i < 64; ++i) { int rb = i & ~15; double v = i + 0.5; double e = (rb + 1 + 0.5) + 1.0 + (rb + 2 + 0.5) * (2.0 * v); if (h[i] != e) { ++bad; if (bad < 5) printf("lane %d got %f expect %f\n", i, h[i], e); } }
    printf("bad=%d\n", bad);
    return 0;
}
