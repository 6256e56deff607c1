// Which DP-ALU instructions take a DPP row_newbcast operand on gfx950, and do they compute what
// the ISA text says?  (design aid)  Result: only v_mov_b64_dpp and v_fmac_f64_dpp assemble
// (v_mul_f64 / v_add_f64 / v_max_f64 are VOP3-only here: "dpp variant of this instruction is not
// supported"); v_fmac_f64_dpp D, S0, S1 computes D += S0[lane k of the row] * S1, accepts a neg
// modifier on S1 and D == S0.  hipcc never forms it from update_dpp + fma (GCNDPPCombine leaves
// the 64-bit move alone) and pads consecutive dependent inline-asm statements with s_nop 0.
//   hipcc -O3 --offload-arch=gfx950 dpp64_fused_test.hip -o dpp64_fused_test && ./dpp64_fused_test
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(double *p)
{
    double v = p[threadIdx.x], w = p[64 + threadIdx.x], acc = p[128 + threadIdx.x], b;
    asm volatile(
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b64_dpp %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_fmac_f64_dpp %0, %0, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"    // D == S0
        "v_fmac_f64_dpp %0, %2, -%3 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"   // neg modifier
        : "+v"(acc), "=&v"(b) : "v"(v), "v"(w));
    p[threadIdx.x] = acc;
    p[64 + threadIdx.x] = b;
}

int main()
{
    double h[192], *d, v[64], w[64], acc[64], a1[64];
    for (int i = 0; i < 192; ++i) h[i] = (i % 64) * 0.25 + 0.5 + i / 64;
    for (int i = 0; i < 64; ++i) { v[i] = h[i]; w[i] = h[64 + i]; acc[i] = h[128 + i]; }
    for (int i = 0; i < 64; ++i) a1[i] = __builtin_fma(v[(i & ~15) + 2], w[i], acc[i]);
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 2;
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 128 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int rb = i & ~15;
        const double a2 = __builtin_fma(a1[rb + 1], w[i], a1[i]);
        const double a3 = __builtin_fma(v[rb + 4], -w[i], a2);
        if (h[i] != a3 || h[64 + i] != v[rb + 7]) {
            if (++bad < 5) printf("lane %d got %.17g / %.17g expect %.17g / %.17g\n", i, h[i], h[64 + i], a3, v[rb + 7]);
        }
    }
    printf("bad=%d\n", bad);
    return bad != 0;
}
