// v_mfma_f64_4x4x4_4b_f64 as a cross-lane contraction primitive for a LONE wave (design aid).
//
// Questions: (1) which lane supplies A(i,k), B(k,j) and which lane receives D(i,j) in each of the
// four 16-lane blocks; (2) what one of them costs a lone wave's instruction stream (independent
// issue interval; dependent through C, through A, through B; result consumed by a VALU
// instruction; VALU result consumed by the MFMA; independent VALU work issued beside it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define ITER 2000

__device__ __forceinline__ double mfma(double a, double b, double c)
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// ---- layout probe: for every source lane p of block 0, A = onehot(p), B = 1 + lane ----------
__global__ void layout_probe(double *dA, double *dB)
{
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        const double a = (lane == p) ? 1.0 : 0.0;
        const double b = 1.0 + lane;
        dA[p * 64 + lane] = mfma(a, b, 0.0);    // D lanes fed by A-lane p, valued by their B lane
        dB[p * 64 + lane] = mfma(b, a, 0.0);    // D lanes fed by B-lane p, valued by their A lane
    }
}

template <int TEST>
__global__ void timing(double *out, double seed)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5,
           a6 = seed + 6, a7 = seed + 7;
    double x = 1.0000001 * seed, y = 0.999999 * seed, c = 1e-9;
    asm volatile("" : "+v"(x), "+v"(y), "+v"(c));
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (TEST == 0) {          // 8 independent accumulators (C chains are 8 apart)
                a0 = mfma(x, y, a0); a1 = mfma(x, y, a1); a2 = mfma(x, y, a2); a3 = mfma(x, y, a3);
                a4 = mfma(x, y, a4); a5 = mfma(x, y, a5); a6 = mfma(x, y, a6); a7 = mfma(x, y, a7);
            } else if (TEST == 1) {   // dependent through C
                a0 = mfma(x, y, a0); a0 = mfma(x, y, a0); a0 = mfma(x, y, a0); a0 = mfma(x, y, a0);
                a0 = mfma(x, y, a0); a0 = mfma(x, y, a0); a0 = mfma(x, y, a0); a0 = mfma(x, y, a0);
            } else if (TEST == 2) {   // dependent through A
                a0 = mfma(a0, y, c); a0 = mfma(a0, y, c); a0 = mfma(a0, y, c); a0 = mfma(a0, y, c);
                a0 = mfma(a0, y, c); a0 = mfma(a0, y, c); a0 = mfma(a0, y, c); a0 = mfma(a0, y, c);
            } else if (TEST == 3) {   // dependent through B
                a0 = mfma(x, a0, c); a0 = mfma(x, a0, c); a0 = mfma(x, a0, c); a0 = mfma(x, a0, c);
                a0 = mfma(x, a0, c); a0 = mfma(x, a0, c); a0 = mfma(x, a0, c); a0 = mfma(x, a0, c);
            } else if (TEST == 4) {   // MFMA -> dependent FMA -> MFMA (4 + 4)
                a0 = mfma(x, y, a0); a0 = __builtin_fma(a0, y, c); a0 = mfma(x, y, a0); a0 = __builtin_fma(a0, y, c);
                a0 = mfma(x, y, a0); a0 = __builtin_fma(a0, y, c); a0 = mfma(x, y, a0); a0 = __builtin_fma(a0, y, c);
            } else if (TEST == 5) {   // 4 independent MFMA chains + 4 independent FMA chains, interleaved
                a0 = mfma(x, y, a0); a4 = __builtin_fma(a4, y, c); a1 = mfma(x, y, a1); a5 = __builtin_fma(a5, y, c);
                a2 = mfma(x, y, a2); a6 = __builtin_fma(a6, y, c); a3 = mfma(x, y, a3); a7 = __builtin_fma(a7, y, c);
            } else if (TEST == 6) {   // 1 dependent MFMA chain (through B) + 3 independent FMAs per MFMA
                a0 = mfma(x, a0, c); a4 = __builtin_fma(a4, y, c); a5 = __builtin_fma(a5, y, c); a6 = __builtin_fma(a6, y, c);
                a0 = mfma(x, a0, c); a4 = __builtin_fma(a4, y, c); a5 = __builtin_fma(a5, y, c); a6 = __builtin_fma(a6, y, c);
            } else if (TEST == 7) {   // 8 independent FMA chains (the baseline slot)
                a0 = __builtin_fma(a0, y, c); a1 = __builtin_fma(a1, y, c); a2 = __builtin_fma(a2, y, c); a3 = __builtin_fma(a3, y, c);
                a4 = __builtin_fma(a4, y, c); a5 = __builtin_fma(a5, y, c); a6 = __builtin_fma(a6, y, c); a7 = __builtin_fma(a7, y, c);
            } else if (TEST == 8) {   // 2 independent MFMA chains (through B)
                a0 = mfma(x, a0, c); a1 = mfma(x, a1, c); a0 = mfma(x, a0, c); a1 = mfma(x, a1, c);
                a0 = mfma(x, a0, c); a1 = mfma(x, a1, c); a0 = mfma(x, a0, c); a1 = mfma(x, a1, c);
            } else if (TEST == 9) {   // dependent MFMA (through B) + 1 independent FMA per MFMA
                a0 = mfma(x, a0, c); a4 = __builtin_fma(a4, y, c); a0 = mfma(x, a0, c); a5 = __builtin_fma(a5, y, c);
                a0 = mfma(x, a0, c); a4 = __builtin_fma(a4, y, c); a0 = mfma(x, a0, c); a5 = __builtin_fma(a5, y, c);
            } else if (TEST == 10) {  // VALU mul -> MFMA (as A) -> VALU mul ...
                a0 = a0 * y; a0 = mfma(a0, y, c); a0 = a0 * y; a0 = mfma(a0, y, c);
                a0 = a0 * y; a0 = mfma(a0, y, c); a0 = a0 * y; a0 = mfma(a0, y, c);
            }
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int T>
void run(const char *name, int per_iter_instr)
{
    double *out;
    hipMalloc(&out, sizeof(double) * 64);
    hipLaunchKernelGGL(timing<T>, dim3(1), dim3(64), 0, 0, out, 1.0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(timing<T>, dim3(1), dim3(64), 0, 0, out, 1.0);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double n = (double)per_iter_instr * 2 * ITER;
    printf("%-64s ns per instruction %7.3f   ns per group of %d: %8.3f\n", name, best * 1e6 / n,
           per_iter_instr, best * 1e6 / (2.0 * ITER));
    hipFree(out);
}

int main()
{
    double *dA, *dB;
    hipMalloc(&dA, sizeof(double) * 64 * 64);
    hipMalloc(&dB, sizeof(double) * 64 * 64);
    hipLaunchKernelGGL(layout_probe, dim3(1), dim3(64), 0, 0, dA, dB);
    hipDeviceSynchronize();
    std::vector<double> hA(64 * 64), hB(64 * 64);
    hipMemcpy(hA.data(), dA, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost);
    hipMemcpy(hB.data(), dB, sizeof(double) * 64 * 64, hipMemcpyDeviceToHost);
    printf("layout (lanes 0..15 = block 0; value = 1 + lane of the partner operand):\n");
    for (int p = 0; p < 16; ++p) {
        printf(" A lane %2d feeds D lanes:", p);
        for (int l = 0; l < 64; ++l)
            if (hA[p * 64 + l] != 0.0) printf(" %d(B lane %d)", l, (int)hA[p * 64 + l] - 1);
        printf("\n");
    }
    for (int p = 0; p < 16; ++p) {
        printf(" B lane %2d feeds D lanes:", p);
        for (int l = 0; l < 64; ++l)
            if (hB[p * 64 + l] != 0.0) printf(" %d(A lane %d)", l, (int)hB[p * 64 + l] - 1);
        printf("\n");
    }
    int cross = 0;
    for (int p = 0; p < 64; ++p)
        for (int l = 0; l < 64; ++l)
            if ((p / 16) != (l / 16) && (hA[p * 64 + l] != 0.0 || hB[p * 64 + l] != 0.0)) ++cross;
    printf(" entries that cross 16-lane blocks: %d\n", cross);

    run<7>("v_fma_f64, 8 independent chains", 8);
    run<0>("mfma 4x4x4 f64, 8 independent accumulators", 8);
    run<8>("mfma, 2 independent chains through B", 8);
    run<1>("mfma, dependent through C", 8);
    run<2>("mfma, dependent through A", 8);
    run<3>("mfma, dependent through B", 8);
    run<4>("mfma -> dependent v_fma_f64 -> mfma (4 + 4)", 8);
    run<10>("v_mul_f64 -> mfma (as A) -> v_mul_f64 (4 + 4)", 8);
    run<5>("4 independent mfma + 4 independent v_fma_f64", 8);
    run<9>("dependent mfma (B) + 1 independent v_fma_f64 each (4 + 4)", 8);
    run<6>("dependent mfma (B) + 3 independent v_fma_f64 each (2 + 6)", 8);
    return 0;
}
