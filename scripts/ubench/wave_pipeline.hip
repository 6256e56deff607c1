// Cost of handing doubles between the waves of one workgroup (one wave per SIMD) through LDS with
// s_barrier, per step of a lock-step loop -- the feasibility number for a wave-specialised rollout
// pipeline (geometry wave / dynamics wave / policy wave).  Design aid, not product code.
//   hipcc -O3 --offload-arch=gfx950 wave_pipeline.hip -o wave_pipeline && ./wave_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 4000

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// WAVES waves; every iteration: WORK independent-ish FMAs, write NX doubles per lane to LDS, barrier,
// read NX doubles written by the next wave, fold them in.  BARRIERS per iteration: 1 or 2.
template <int WAVES, int WORK, int NX, int BARRIERS>
__global__ void __launch_bounds__(WAVES * 64) k(unsigned long long *cyc, double seed, double *out)
{
    __shared__ double sh[2][WAVES][8][64];
    const int w = threadIdx.x / 64, l = threadIdx.x % 64;
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + w;
    const double m = 1.0000001, c = 1e-9;
    unsigned long long t0 = now();
    for (int it = 0; it < ITER; ++it) {
        const int buf = it & 1;
#pragma unroll
        for (int j = 0; j < WORK; ++j) a[j & 7] = __builtin_fma(a[j & 7], m, c);
#pragma unroll
        for (int x = 0; x < NX; ++x) sh[buf][w][x][l] = a[x];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        if (BARRIERS == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j & 7] = __builtin_fma(a[j & 7], m, c);
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) a[x] += sh[buf][(w + 1) % WAVES][x][l];
    }
    unsigned long long t1 = now();
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
}

template <int WAVES, int WORK, int NX, int BARRIERS>
void run(const char *what)
{
    unsigned long long *cyc;
    double *out;
    CK(hipMalloc(&cyc, 8));
    CK(hipMalloc(&out, 8 * WAVES * 64));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<WAVES, WORK, NX, BARRIERS>), dim3(1), dim3(WAVES * 64), 0, 0, cyc, 1.0, out);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-58s %7.1f ticks per iteration\n", what, (double)h / ITER);
    CK(hipFree(cyc));
    CK(hipFree(out));
}

int main()
{
    run<1, 64, 0, 1>("1 wave, 64 FMAs, no exchange (barrier of one wave)");
    run<4, 64, 0, 1>("4 waves, 64 FMAs, barrier only");
    run<4, 64, 2, 1>("4 waves, 64 FMAs, 2 doubles through LDS + 1 barrier");
    run<4, 64, 6, 1>("4 waves, 64 FMAs, 6 doubles through LDS + 1 barrier");
    run<4, 64, 6, 2>("4 waves, 64 FMAs, 6 doubles through LDS + 2 barriers");
    run<3, 64, 6, 1>("3 waves, 64 FMAs, 6 doubles through LDS + 1 barrier");
    run<4, 16, 6, 1>("4 waves, 16 FMAs, 6 doubles through LDS + 1 barrier");
    run<4, 0, 2, 1>("4 waves, 0 FMAs, 2 doubles through LDS + 1 barrier");
    return 0;
}
