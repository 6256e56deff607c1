// stream_copy.hip -- measured HBM stream-copy peak of the box (the denominator SURVEY 8d asks
// to report beside the nominal 8 TB/s).  Copies a buffer far larger than the 256 MB Infinity
// Cache with 16-byte accesses, plain and nontemporal, and reports read+write bytes / time.
//   hipcc -O3 --offload-arch=gfx950 stream_copy.hip -o stream_copy && ./stream_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double double2_t __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const double2_t *__restrict__ src,
                                                   double2_t *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else dst[i] = src[i];
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void read_kernel(const double2_t *__restrict__ src,
                                                   double *__restrict__ sink, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double2_t v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e300) *sink = acc;   // never true: keeps the loads alive
}

template <typename F>
static double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;            // 4 GiB per buffer
    const size_t n = bytes / sizeof(double2_t);
    double2_t *a, *b;
    double *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    for (int blocks_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * blocks_per_cu;
        double t;
        t = time_ms([&] { copy_kernel<false><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy   plain  grid %5d: %7.1f GB/s (read+write)\n", grid, 2.0 * bytes / t / 1e6);
        t = time_ms([&] { copy_kernel<true><<<grid, 256>>>(a, b, n); }, 10);
        printf("copy   nontmp grid %5d: %7.1f GB/s (read+write)\n", grid, 2.0 * bytes / t / 1e6);
        t = time_ms([&] { read_kernel<false><<<grid, 256>>>(a, sink, n); }, 10);
        printf("read   plain  grid %5d: %7.1f GB/s\n", grid, 1.0 * bytes / t / 1e6);
        t = time_ms([&] { read_kernel<true><<<grid, 256>>>(a, sink, n); }, 10);
        printf("read   nontmp grid %5d: %7.1f GB/s\n", grid, 1.0 * bytes / t / 1e6);
    }
    double t = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
    printf("hipMemcpyDtoD          : %7.1f GB/s (read+write)\n", 2.0 * bytes / t / 1e6);
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(sink));
    return 0;
}
