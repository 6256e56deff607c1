// Does a buffer store get cheaper when lanes are masked off (EXEC) or dropped by the range check?
// Lone wave, loop of 8 independent f64 FMAs + 3 buffer_store_dwordx2 per iteration (the rollout
// kernels' pattern: SGPR base, per-iteration SGPR offset, per-lane VGPR offset).  Design aid.
//   hipcc -O3 --offload-arch=gfx950 store_mask.hip -o store_mask && ./store_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define ITER 2000

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// MODE 0: no stores; 1: 3 stores, all lanes; 2: 3 stores, EXEC = 0x7777.. for the whole loop;
// 3: EXEC = 0x3333..; 4: all lanes active, lanes with (lane & 3) >= 2 out of range (dropped);
// 5: EXEC = 0x0000ffff0000ffff (two whole 16-lane rows off)
template <int MODE>
__global__ void k(double *buf, size_t bytes, unsigned long long *cyc, double seed, double *out)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i;
    const double m = 1.0000001, c = 1e-9;
    const int lane = threadIdx.x;
    unsigned voff = lane * 8;
    if (MODE == 4 && (lane & 3) >= 2) voff = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
    if (MODE == 2) asm volatile("s_mov_b32 exec_lo, 0x77777777\n\ts_mov_b32 exec_hi, 0x77777777");
    if (MODE == 3) asm volatile("s_mov_b32 exec_lo, 0x33333333\n\ts_mov_b32 exec_hi, 0x33333333");
    if (MODE == 5) asm volatile("s_mov_b32 exec_lo, 0x0000ffff\n\ts_mov_b32 exec_hi, 0x0000ffff");
    unsigned soff = 0;
    unsigned long long t0 = now();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], m, c);
        if (MODE != 0) {
            typedef int v2i __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                union { double d; v2i i; } u;
                u.d = a[s];
                __builtin_amdgcn_raw_buffer_store_b64(u.i, rs, (int)(voff + (MODE == 4 && (lane & 3) >= 2 ? 0 : s * 512)),
                                                      (int)soff, 16);
            }
            soff += 1536;
            if ((it & 255) == 255) soff = 0;
        }
    }
    unsigned long long t1 = now();
    asm volatile("s_mov_b64 exec, -1");
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[threadIdx.x] = s;
}

template <int MODE> void run(const char *what, double base)
{
    unsigned long long *cyc;
    double *out, *buf;
    const size_t bytes = 1536 * 256 + 4096;
    CK(hipMalloc(&cyc, 8));
    CK(hipMalloc(&out, 8 * 64));
    CK(hipMalloc(&buf, bytes));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, buf, bytes, cyc, 1.0, out);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h;
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-62s %6.1f ticks per iteration (+%.1f)\n", what, (double)h / ITER, (double)h / ITER - base);
}

int main()
{
    run<0>("8 FMAs, no stores", 0);
    const double base = 0;
    run<1>("+ 3 stores, all 64 lanes", base);
    run<2>("+ 3 stores, EXEC = 0x7777.. (lane 3 of every quad off)", base);
    run<3>("+ 3 stores, EXEC = 0x3333.. (lanes 2, 3 of every quad off)", base);
    run<4>("+ 3 stores, all lanes active, lanes 2, 3 of every quad out of range", base);
    run<5>("+ 3 stores, EXEC = 0x0000ffff0000ffff (two 16-lane rows off)", base);
    return 0;
}
