// Lone-wave cost of trajectory stores and of v_rcp_f64 on gfx950, and the raw accuracy of
// v_rcp_f64 with 0 / 1 / 2 Newton steps (design aid, not product code).
//   hipcc -O3 --offload-arch=gfx950 store_rcp.hip -o store_rcp && ./store_rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 2000

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

#define FMA8 \
    "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n" \
    "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n" \
    "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"

// The first store of an iteration also pays for the loop's pointer update waiting until the store
// has read its address registers (~40 ticks here, absent in the rollout kernels where offsets are
// loop-invariant); the cost of a store as such is the increment from 1 to 3 (x2) or 1 to 2 (x4).
// TEST 0: 8 independent FMAs per iteration (baseline)
//      1: + 1 global_store_dwordx2       2: + 3 global_store_dwordx2     3: + 1 global_store_dwordx4
//      4: + 2 global_store_dwordx4       5: + 1 v_rcp_f64 (independent)  6: + 1 global_store_dword
template <int TEST>
__global__ void k(double *buf, unsigned long long *cyc, double seed, double *out)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5,
           a6 = seed + 6, a7 = seed + 7;
    double m = 1.0000001, c = 1e-9, rr = seed + 9;
    double *p = buf + threadIdx.x * 2;
    typedef double d2 __attribute__((ext_vector_type(2)));
    unsigned long long t0 = now();
    for (int it = 0; it < ITER; ++it) {
        asm volatile(FMA8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c));
        if (TEST == 1 || TEST == 2) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(a0) : "memory");
        if (TEST == 2) {
            asm volatile("global_store_dwordx2 %0, %1, off offset:1024" ::"v"(p), "v"(a1) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off offset:2048" ::"v"(p), "v"(a2) : "memory");
        }
        if (TEST == 3 || TEST == 4) {
            d2 v = {a0, a1};
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
        }
        if (TEST == 4) {
            d2 v = {a2, a3};
            asm volatile("global_store_dwordx4 %0, %1, off offset:2048" ::"v"(p), "v"(v) : "memory");
        }
        if (TEST == 5) asm volatile("v_rcp_f64 %0, %1" : "=v"(rr) : "v"(a0));
        if (TEST == 6) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"((float)a0) : "memory");
        p += 64 * 2 * 4;   // next 4 KB: streaming, like the trajectory
        if ((it & 255) == 255) p = buf + threadIdx.x * 2;
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + rr;
}

__global__ void rcp_accuracy(const double *x, double *r0, double *r1, double *r2, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double y;
    asm volatile("v_rcp_f64 %0, %1" : "=v"(y) : "v"(d));
    r0[i] = y;
    const double e = __builtin_fma(-d, y, 1.0);
    const double y1 = __builtin_fma(y, e, y);
    r1[i] = y1;
    r2[i] = __builtin_fma(y1, e * e, y1);
}

template <int TEST>
static double run(double *buf, unsigned long long *cyc, double *out)
{
    k<TEST><<<1, 64>>>(buf, cyc, 1.0, out);
    CK(hipDeviceSynchronize());
    k<TEST><<<1, 64>>>(buf, cyc, 1.0, out);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    return (double)c / ITER;   // s_memtime ticks per iteration
}

int main()
{
    double *buf, *out;
    unsigned long long *cyc;
    CK(hipMalloc(&buf, 64 << 20)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8));
    // s_memtime ticks: the same unit as scripts/ubench/issue_cost (an independent f64 FMA = 4.8 ticks
    // = 2.1 ns, i.e. close to shader cycles)
    const double ticks_to_cycles = 1.0;
    const double base = run<0>(buf, cyc, out);
    printf("baseline 8 independent f64 FMA + loop: %.1f ticks per iteration\n", base);
    const char *names[] = {"", "+1 global_store_dwordx2", "+3 global_store_dwordx2", "+1 global_store_dwordx4",
                           "+2 global_store_dwordx4", "+1 v_rcp_f64", "+1 global_store_dword"};
    const double t[] = {0, run<1>(buf, cyc, out), run<2>(buf, cyc, out), run<3>(buf, cyc, out),
                        run<4>(buf, cyc, out), run<5>(buf, cyc, out), run<6>(buf, cyc, out)};
    for (int i = 1; i <= 6; ++i)
        printf("%-26s: +%.1f ticks per iteration\n", names[i], (t[i] - base) * ticks_to_cycles);

    const int n = 1 << 20;
    std::vector<double> x(n), r0(n), r1(n), r2(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 40) - 20);
    double *dx, *d0, *d1, *d2;
    CK(hipMalloc(&dx, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8)); CK(hipMalloc(&d2, n * 8));
    CK(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice));
    rcp_accuracy<<<n / 256, 256>>>(dx, d0, d1, d2, n);
    CK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost));
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        const long double ex = 1.0L / (long double)x[i];
        e0 = fmax(e0, (double)fabsl(((long double)r0[i] - ex) / ex));
        e1 = fmax(e1, (double)fabsl(((long double)r1[i] - ex) / ex));
        e2 = fmax(e2, (double)fabsl(((long double)r2[i] - ex) / ex));
    }
    printf("v_rcp_f64 max relative error: raw %.3e, 1 Newton step %.3e, + e^2 step %.3e (2^-53 = 1.11e-16)\n",
           e0, e1, e2);
    return 0;
}
