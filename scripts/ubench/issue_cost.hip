// Lone-wave instruction cost model on gfx950 (design aid, not product code).
// Each test runs ITER iterations of a 16-instruction block in one wave and reports
// cycles per instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 2000
#define REP16(x) x x x x x x x x x x x x x x x x

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

template <int TEST>
__global__ void k(double *out, unsigned long long *cyc, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5,
           a6 = seed + 6, a7 = seed + 7;
    double m = 1.0000001, c = 1e-9;
    int i0 = (int)seed, i1 = 3;
    unsigned long long t0 = now();
    for (int it = 0; it < ITER; ++it) {
        if (TEST == 0) {  // dependent f64 fma chain
            REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));)
        } else if (TEST == 1) {  // 8 independent f64 fma chains
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n"
                "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n"
                "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n"
                "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(m), "v"(c));
        } else if (TEST == 2) {  // 2 independent chains
            asm volatile(
                "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n"
                : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
        } else if (TEST == 3) {  // f64 fma + s_mov interleaved (8 + 8): does SALU take a slot?
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n s_mov_b32 s20, 1\n v_fma_f64 %1, %1, %8, %9\n s_mov_b32 s21, 2\n"
                "v_fma_f64 %2, %2, %8, %9\n s_mov_b32 s20, 3\n v_fma_f64 %3, %3, %8, %9\n s_mov_b32 s21, 4\n"
                "v_fma_f64 %4, %4, %8, %9\n s_mov_b32 s20, 5\n v_fma_f64 %5, %5, %8, %9\n s_mov_b32 s21, 6\n"
                "v_fma_f64 %6, %6, %8, %9\n s_mov_b32 s20, 7\n v_fma_f64 %7, %7, %8, %9\n s_mov_b32 s21, 8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(m), "v"(c) : "s20", "s21");
        } else if (TEST == 4) {  // independent v_mov_b32 x16
            asm volatile(
                "v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n"
                "v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n"
                "v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n"
                "v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n v_mov_b32 %0, %2\n v_mov_b32 %1, %2\n"
                : "+v"(i0), "+v"(i1) : "v"(it));
        } else if (TEST == 5) {  // dependent DPP quad_perm movs
            REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf" : "+v"(i0));)
        } else if (TEST == 6) {  // f64 add dependent chain
            REP16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a0) : "v"(c));)
        } else if (TEST == 7) {  // v_mov_b64 independent
            asm volatile(
                "v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n"
                "v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n"
                "v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n"
                "v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n v_mov_b64 %0, %2\n v_mov_b64 %1, %2\n"
                : "+v"(a0), "+v"(a1) : "v"(m));
        } else if (TEST == 8) {  // ds_bpermute dependent (LDS crossbar latency)
            REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(i0) : "v"(i1));)
        } else if (TEST == 9) {  // f64 fma independent mixed with f32 fma independent (8+8)
            float f0 = 1.f, f1 = 2.f;
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f64 %1, %1, %8, %9\n v_fma_f32 %11, %11, %11, %11\n"
                "v_fma_f64 %2, %2, %8, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f64 %3, %3, %8, %9\n v_fma_f32 %11, %11, %11, %11\n"
                "v_fma_f64 %4, %4, %8, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f64 %5, %5, %8, %9\n v_fma_f32 %11, %11, %11, %11\n"
                "v_fma_f64 %6, %6, %8, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f64 %7, %7, %8, %9\n v_fma_f32 %11, %11, %11, %11\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(m), "v"(c), "v"(f0), "v"(f1));
        } else if (TEST == 10) {  // v_cndmask_b32 independent
            asm volatile(
                "v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
                "v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
                "v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
                "v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
                : "+v"(i0), "+v"(i1) : "v"(it), "v"(i1) : "vcc");
        } else if (TEST == 11) {  // v_rcp_f64 dependent
            REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a0));)
        } else if (TEST == 12) {  // f64 mul independent x8
            asm volatile(
                "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(m));
        }
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1;
}

template <int T>
void run(const char *name, int blocks, int threads)
{
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double n_instr = 16.0 * ITER;
    printf("%-44s blocks=%4d thr=%4d  memtime ticks/instr=%7.3f  wall ns/instr=%7.3f\n", name, blocks,
           threads, (double)h[0] / n_instr, ms * 1e6 / n_instr);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    // s_memtime counts at a fixed 100 MHz on gfx9 (REFCLK); wall ns/instr is the reliable column
    run<0>("f64 fma dependent chain", 1, 64);
    run<1>("f64 fma 8 independent chains", 1, 64);
    run<2>("f64 fma 2 independent chains", 1, 64);
    run<12>("f64 mul 8 independent", 1, 64);
    run<6>("f64 add dependent chain", 1, 64);
    run<3>("f64 fma indep + s_mov interleaved (16 instr)", 1, 64);
    run<9>("f64 fma indep + f32 fma interleaved", 1, 64);
    run<4>("v_mov_b32 independent", 1, 64);
    run<7>("v_mov_b64 independent", 1, 64);
    run<10>("v_cndmask_b32 independent", 1, 64);
    run<5>("v_mov_b32_dpp quad_perm dependent (+s_nop)", 1, 64);
    run<8>("ds_bpermute_b32 dependent", 1, 64);
    run<11>("v_rcp_f64 dependent", 1, 64);
    // co-residency: 2, 4, 8 waves in one workgroup (same CU, 1/2 per SIMD)
    run<1>("f64 fma 8 indep, 4 waves/WG (1 per SIMD)", 1, 256);
    run<1>("f64 fma 8 indep, 8 waves/WG (2 per SIMD)", 1, 512);
    run<1>("f64 fma 8 indep, 16 waves/WG (4 per SIMD)", 1, 1024);
    run<0>("f64 fma dependent, 8 waves/WG (2 per SIMD)", 1, 512);
    run<0>("f64 fma dependent, 16 waves/WG (4 per SIMD)", 1, 1024);
    run<1>("f64 fma 8 indep, 16 lanes active", 1, 16);
    run<1>("f64 fma 8 indep, 1024 WG x 64", 1024, 64);
    return 0;
}
