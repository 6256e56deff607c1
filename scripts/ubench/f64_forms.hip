// Lone-wave issue interval of the f64 instruction forms the rollout kernels use (design aid):
// does a 3-VGPR-operand FMA issue slower than a multiply, an add, or an FMA with a scalar operand?
//   hipcc -O3 --offload-arch=gfx950 f64_forms.hip -o f64_forms && ./f64_forms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define FMA_VVV(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define FMA_VVS(k) "v_fma_f64 %" #k ", %" #k ", %8, %10\n"
#define FMAC(k)    "v_fmac_f64 %" #k ", %8, %9\n"
#define MUL(k)     "v_mul_f64 %" #k ", %" #k ", %8\n"
#define ADD(k)     "v_add_f64 %" #k ", %" #k ", %9\n"
#define MOV64(k)   "v_mov_b64 %" #k ", %8\n"

template <int MODE>
__global__ void __launch_bounds__(64) k(int trips, double *out)
{
    double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9, sc = 1.5e-9;
#define BODY(I) asm volatile(REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) REP8(I) \
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(sc));
    for (int t = 0; t < trips; ++t) {
        if (MODE == 0) { BODY(FMA_VVV) }
        if (MODE == 1) { BODY(FMA_VVS) }
        if (MODE == 2) { BODY(FMAC) }
        if (MODE == 3) { BODY(MUL) }
        if (MODE == 4) { BODY(ADD) }
        if (MODE == 5) { BODY(MOV64) }
    }
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *what)
{
    double *out;
    CK(hipMalloc(&out, 8 * 64));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int trips = 8192;
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, 16, out);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, trips, out);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, 2 * trips, out);
        CK(hipEventRecord(e2));
        CK(hipDeviceSynchronize());
        float a, b;
        CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2));
        const double ns = (b - a) * 1e6 / ((double)trips * 64);
        if (ns < best) best = ns;
    }
    printf("%-44s %.3f ns per instruction\n", what, best);
}

int main()
{
    run<0>("v_fma_f64  d, d, v, v   (3 VGPR operands)");
    run<1>("v_fma_f64  d, d, v, s   (2 VGPR + 1 SGPR)");
    run<2>("v_fmac_f64 d, v, v      (2 VGPR + tied dst)");
    run<3>("v_mul_f64  d, d, v");
    run<4>("v_add_f64  d, d, v");
    run<5>("v_mov_b64  d, v");
    return 0;
}
