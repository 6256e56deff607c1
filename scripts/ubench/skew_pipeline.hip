// Can a SECOND instruction stream be put behind one rollout?  Explicit Euler (remy_swimmer_env.py:88-91)
// makes theta_{t+1} -- hence every sin / cos / cos(th_i - th_k) and the factorisation of the mass matrix of
// step t + 1 -- a function of thetadot_t alone, which is known when the dynamics of step t START.  A
// "geometry" wave G can therefore run one step ahead of the "dynamics" wave D:
//
//      D(t-1) --thetadot_t--> G(t): geometry of theta_{t+1} --> D(t+1)          D(t) runs meanwhile
//
// no barrier, the hand-over through double-buffered LDS slots with a per-lane sequence number.  The loop
// closes over TWO steps:  2 P >= D + G + 2 X  and  P >= max(D, G)   (P = period per step, D / G = the
// two streams' issue times, X = publish -> visible latency).  This file measures X and the period of a
// model pipeline (independent-ish FMAs as the streams' work, real data dependences through the slots,
// results checked against the same recurrence run by ONE wave).  Design aid, not product code.
//   hipcc -O3 --offload-arch=gfx950 skew_pipeline.hip -o skew_pipeline && ./skew_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 4000;
constexpr uint32_t SPIN_CAP = 200000;   // every poll loop gives up (and the launch reports it)

__device__ __forceinline__ unsigned long long now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__device__ __forceinline__ uint32_t hw_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

// ---- LDS primitives (per-lane slots: lane l owns bytes [8 l, 8 l + 8) of every 512-byte item) ----
typedef double v2d __attribute__((ext_vector_type(2)));

template <int O0, int O1>
__device__ __forceinline__ void lds_w2(uint32_t a, double x, double y)
{
    asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(a), "v"(x), "v"(y), "n"(O0), "n"(O1)
                 : "memory");
}
__device__ __forceinline__ void lds_w1(uint32_t a, double x)
{
    asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_wseq(uint32_t a, uint32_t s)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(s) : "memory");
}

// One attempt: the sequence number FIRST, then NX doubles (LDS serves a wave's requests in order, the
// writer stores the data before the number: a fresh number means fresh data), one wait for all of it.
template <int NX>
struct Slot {
    double v[NX];
};

template <int NX>
__device__ __forceinline__ uint32_t lds_try(uint32_t aseq, uint32_t adata, Slot<NX> &S);

template <>
__device__ __forceinline__ uint32_t lds_try<1>(uint32_t aseq, uint32_t adata, Slot<1> &S)
{
    uint32_t s;
    asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(s), "=&v"(S.v[0]) : "v"(aseq), "v"(adata) : "memory");
    return s;
}
template <>
__device__ __forceinline__ uint32_t lds_try<2>(uint32_t aseq, uint32_t adata, Slot<2> &S)
{
    uint32_t s;
    v2d p0;
    asm volatile("ds_read_b32 %0, %2\n\tds_read2st64_b64 %1, %3 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(s), "=&v"(p0) : "v"(aseq), "v"(adata) : "memory");
    S.v[0] = p0.x; S.v[1] = p0.y;
    return s;
}
template <>
__device__ __forceinline__ uint32_t lds_try<8>(uint32_t aseq, uint32_t adata, Slot<8> &S)
{
    uint32_t s;
    v2d p0, p1, p2, p3;
    asm volatile("ds_read_b32 %0, %5\n\t"
                 "ds_read2st64_b64 %1, %6 offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %6 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %3, %6 offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %4, %6 offset0:6 offset1:7\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(s), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(aseq), "v"(adata) : "memory");
    S.v[0] = p0.x; S.v[1] = p0.y; S.v[2] = p1.x; S.v[3] = p1.y;
    S.v[4] = p2.x; S.v[5] = p2.y; S.v[6] = p3.x; S.v[7] = p3.y;
    return s;
}
template <>
__device__ __forceinline__ uint32_t lds_try<14>(uint32_t aseq, uint32_t adata, Slot<14> &S)
{
    uint32_t s;
    v2d p0, p1, p2, p3, p4, p5, p6;
    asm volatile("ds_read_b32 %0, %8\n\t"
                 "ds_read2st64_b64 %1, %9 offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %2, %9 offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %3, %9 offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %4, %9 offset0:6 offset1:7\n\t"
                 "ds_read2st64_b64 %5, %9 offset0:8 offset1:9\n\t"
                 "ds_read2st64_b64 %6, %9 offset0:10 offset1:11\n\t"
                 "ds_read2st64_b64 %7, %9 offset0:12 offset1:13\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(s), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6)
                 : "v"(aseq), "v"(adata) : "memory");
    S.v[0] = p0.x; S.v[1] = p0.y; S.v[2] = p1.x; S.v[3] = p1.y; S.v[4] = p2.x; S.v[5] = p2.y;
    S.v[6] = p3.x; S.v[7] = p3.y; S.v[8] = p4.x; S.v[9] = p4.y; S.v[10] = p5.x; S.v[11] = p5.y;
    S.v[12] = p6.x; S.v[13] = p6.y;
    return s;
}

template <int NX>
__device__ __forceinline__ void lds_publish(uint32_t aseq, uint32_t adata, const Slot<NX> &S, uint32_t seq)
{
    if constexpr (NX == 1) {
        lds_w1(adata, S.v[0]);
    } else {
        if constexpr (NX >= 2) lds_w2<0, 1>(adata, S.v[0], S.v[1]);
        if constexpr (NX >= 4) lds_w2<2, 3>(adata, S.v[2], S.v[3]);
        if constexpr (NX >= 6) lds_w2<4, 5>(adata, S.v[4], S.v[5]);
        if constexpr (NX >= 8) lds_w2<6, 7>(adata, S.v[6], S.v[7]);
        if constexpr (NX >= 10) lds_w2<8, 9>(adata, S.v[8], S.v[9]);
        if constexpr (NX >= 12) lds_w2<10, 11>(adata, S.v[10], S.v[11]);
        if constexpr (NX >= 14) lds_w2<12, 13>(adata, S.v[12], S.v[13]);
    }
    lds_wseq(aseq, seq);
}

// spin until the slot carries `expect`; false after SPIN_CAP attempts (the launch then reports a failure)
template <int NX>
__device__ __forceinline__ bool lds_wait(uint32_t aseq, uint32_t adata, Slot<NX> &S, uint32_t expect,
                                         unsigned long long &spins)
{
#pragma nounroll
    for (uint32_t i = 0; i < SPIN_CAP; ++i) {
        const uint32_t s = lds_try<NX>(aseq, adata, S);
        if (__all(s == expect)) return true;   // every lane's copy (a 64-lane LDS write is several passes)
        ++spins;
    }
    return false;
}

// WORK independent-ish FMAs over eight accumulators (dependence distance 8: issue-bound like the rollout step)
template <int WORK>
__device__ __forceinline__ void work(double (&a)[8], double m, double c)
{
#pragma unroll
    for (int j = 0; j < WORK; ++j)
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j & 7]) : "v"(m), "v"(c));
}

struct Result {
    unsigned long long cycles, spins_g, spins_d;
    uint32_t hw_g, hw_d, fail;
};

// ---- (1) raw LDS latencies of a lone wave ----------------------------------------------------------
__global__ void __launch_bounds__(64) lone_latency(unsigned long long *out)
{
    __shared__ double sh[16 * 64];
    const uint32_t a = (uint32_t)(size_t)(&sh[threadIdx.x]) ;
    double v = 1.0;
    sh[threadIdx.x] = 0.0;
    __syncthreads();
    unsigned long long t0 = now();
    for (int i = 0; i < ITER; ++i)
        asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    unsigned long long t1 = now();
    for (int i = 0; i < ITER; ++i)
        asm volatile("ds_write_b64 %1, %0\n\ts_waitcnt lgkmcnt(0)" ::"v"(v), "v"(a) : "memory");
    unsigned long long t2 = now();
    for (int i = 0; i < ITER; ++i)
        asm volatile("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(a) : "memory");
    unsigned long long t3 = now();
    Slot<14> S;
    uint32_t s = 0;
    for (int i = 0; i < ITER; ++i) s += lds_try<14>(a, a, S);
    unsigned long long t4 = now();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = t2 - t1;
        out[2] = t3 - t2;
        out[3] = t4 - t3;
        out[4] = s + (unsigned long long)S.v[3];
    }
}

// ---- (2) ping-pong: one-way publish -> visible latency between two waves of a workgroup ------------
// NX doubles + the sequence number each way.
template <int NX>
__global__ void __launch_bounds__(128) pingpong(Result *res, double seed)
{
    __shared__ double data[2][NX][64];
    __shared__ uint32_t seqs[2][64];
    const int w = threadIdx.x / 64, l = threadIdx.x % 64;
    const uint32_t a_my = (uint32_t)(size_t)&data[w][0][l], a_other = (uint32_t)(size_t)&data[1 - w][0][l];
    const uint32_t s_my = (uint32_t)(size_t)&seqs[w][l], s_other = (uint32_t)(size_t)&seqs[1 - w][l];
    seqs[w][l] = 0;
    __syncthreads();
    Slot<NX> S;
    for (int x = 0; x < NX; ++x) S.v[x] = seed + x;
    unsigned long long spins = 0;
    bool ok = true;
    unsigned long long t0 = now();
    for (uint32_t i = 1; i <= (uint32_t)ITER && ok; ++i) {
        if (w == 0) {
            lds_publish<NX>(s_my, a_my, S, i);
            ok = lds_wait<NX>(s_other, a_other, S, i, spins);
        } else {
            ok = lds_wait<NX>(s_other, a_other, S, i, spins);
            S.v[0] += 1.0;
            lds_publish<NX>(s_my, a_my, S, i);
        }
    }
    unsigned long long t1 = now();
    Result &r = res[blockIdx.x];
    if (l == 0) {
        if (w == 0) {
            r.cycles = t1 - t0;
            r.spins_g = spins;
            r.hw_g = hw_id();
            r.fail = ok ? 0 : 1;
        } else {
            r.spins_d = spins;
            r.hw_d = hw_id();
            if (!ok) atomicOr(&r.fail, 2u);
        }
    }
    if (S.v[0] == -1.0) res[0].cycles = 0;   // keep S alive
}

// ---- (3) the skewed pipeline model ---------------------------------------------------------------
// The recurrence (both forms compute exactly this, so their results must agree bit for bit):
//   thd_0 = seed;  geo_0[x] = seed + x
//   G(t):  b[0] += thd_t;  WG FMAs on b[];  geo_{t+1}[x] = b[x & 7]                       (needs thd_t)
//   D(t):  a[x & 7] += geo_t[x] for x < NG;  WD FMAs on a[];  thd_{t+1} = a[0]            (needs geo_t)
// Two waves: wave 0 = G, wave 1 = D; the slots are double-buffered by step parity.
template <int WG, int WD, int NG>
__global__ void __launch_bounds__(128) skew2(Result *res, double *out, double seed, int iters)
{
    __shared__ double geo[2][NG][64];
    __shared__ double thds[2][1][64];
    __shared__ uint32_t gseq[2][64], tseq[2][64];
    const int w = threadIdx.x / 64, l = threadIdx.x % 64;
    const double m = 0.999999, c = 1e-7;
    // step 0's inputs are there before the loop starts
    if (w == 0) {
        for (int x = 0; x < NG; ++x) geo[0][x][l] = seed + x;
        gseq[0][l] = 1;       // geometry of step t carries sequence number t + 1
        gseq[1][l] = 0;
        thds[0][0][l] = seed;
        tseq[0][l] = 1;       // thetadot of step t carries sequence number t + 1
        tseq[1][l] = 0;
    }
    __syncthreads();
    unsigned long long spins = 0;
    bool ok = true;
    double fin = 0.0;
    unsigned long long t0 = now();
    if (w == 0) {   // ---- G
        double b[8];
        for (int i = 0; i < 8; ++i) b[i] = seed + 0.5 * i;
        for (int t = 0; t < iters && ok; ++t) {
            Slot<1> T;
            ok = lds_wait<1>((uint32_t)(size_t)&tseq[t & 1][l], (uint32_t)(size_t)&thds[t & 1][0][l], T,
                             (uint32_t)t + 1u, spins);
            b[0] += T.v[0];
            work<WG>(b, m, c);
            Slot<NG> O;
#pragma unroll
            for (int x = 0; x < NG; ++x) O.v[x] = b[x & 7];
            lds_publish<NG>((uint32_t)(size_t)&gseq[(t + 1) & 1][l], (uint32_t)(size_t)&geo[(t + 1) & 1][0][l], O,
                            (uint32_t)t + 2u);
        }
        fin = b[0];
    } else {        // ---- D
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed - 0.25 * i;
        for (int t = 0; t < iters && ok; ++t) {
            Slot<NG> I;
            ok = lds_wait<NG>((uint32_t)(size_t)&gseq[t & 1][l], (uint32_t)(size_t)&geo[t & 1][0][l], I,
                              (uint32_t)t + 1u, spins);
#pragma unroll
            for (int x = 0; x < NG; ++x) a[x & 7] += I.v[x];
            work<WD>(a, m, c);
            Slot<1> O;
            O.v[0] = a[0];
            lds_publish<1>((uint32_t)(size_t)&tseq[(t + 1) & 1][l], (uint32_t)(size_t)&thds[(t + 1) & 1][0][l], O,
                           (uint32_t)t + 2u);
        }
        fin = a[0];
    }
    unsigned long long t1 = now();
    Result &r = res[blockIdx.x];
    if (l == 0) {
        if (w == 0) {
            r.spins_g = spins;
            r.hw_g = hw_id();
            if (!ok) atomicOr(&r.fail, 1u);
        } else {
            r.cycles = t1 - t0;
            r.spins_d = spins;
            r.hw_d = hw_id();
            if (!ok) atomicOr(&r.fail, 2u);
        }
    }
    out[(size_t)blockIdx.x * 128 + threadIdx.x] = fin;
}

// the same recurrence in ONE wave (no LDS): today's single-stream shape
template <int WG, int WD, int NG>
__global__ void __launch_bounds__(64) skew1(Result *res, double *out, double seed, int iters)
{
    const int l = threadIdx.x;
    const double m = 0.999999, c = 1e-7;
    double a[8], b[8], geo[NG], thd = seed;
    for (int i = 0; i < 8; ++i) {
        b[i] = seed + 0.5 * i;
        a[i] = seed - 0.25 * i;
    }
    for (int x = 0; x < NG; ++x) geo[x] = seed + x;
    unsigned long long t0 = now();
    for (int t = 0; t < iters; ++t) {
        // G(t) on thd_t
        b[0] += thd;
        work<WG>(b, m, c);
        double gn[NG];
#pragma unroll
        for (int x = 0; x < NG; ++x) gn[x] = b[x & 7];
        // D(t) on geo_t
#pragma unroll
        for (int x = 0; x < NG; ++x) a[x & 7] += geo[x];
        work<WD>(a, m, c);
        thd = a[0];
#pragma unroll
        for (int x = 0; x < NG; ++x) geo[x] = gn[x];
    }
    unsigned long long t1 = now();
    if (l == 0) {
        res[blockIdx.x].cycles = t1 - t0;
        res[blockIdx.x].hw_g = hw_id();
    }
    out[(size_t)blockIdx.x * 128 + l] = b[0];
    out[(size_t)blockIdx.x * 128 + 64 + l] = a[0];
}


// ---- (4) the same pipeline with a STREAMED hand-over (what a hand-written kernel would do) ------------
//  * the roles are wave-uniform (scalar branch), the poll loops are hand-written;
//  * D issues the sequence-number read and all data reads at once, checks the number as soon as IT is back
//    (partial s_waitcnt: LDS returns a wave's reads in order) and consumes the pairs as they arrive;
//  * G stores each pair as soon as it is computed (the stores sit between the FMAs), the number last.
template <int NH>
__device__ __forceinline__ bool stream_reads(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[NH],
                                             unsigned long long &spins);

template <>
__device__ __forceinline__ bool stream_reads<4>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[4],
                                                  unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "s_waitcnt lgkmcnt(4)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk_ok_%=\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk_retry_%=\n"
                 ".Lsk_ok_%=:"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

template <>
__device__ __forceinline__ bool stream_reads<7>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[7],
                                                  unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "ds_read2st64_b64 %6, %[ad] offset0:8 offset1:9\n\t"
                 "ds_read2st64_b64 %7, %[ad] offset0:10 offset1:11\n\t"
                 "ds_read2st64_b64 %8, %[ad] offset0:12 offset1:13\n\t"
                 "s_waitcnt lgkmcnt(7)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk_ok_%=\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk_retry_%=\n"
                 ".Lsk_ok_%=:"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

template <>
__device__ __forceinline__ bool stream_reads<14>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[14],
                                                  unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "ds_read2st64_b64 %6, %[ad] offset0:8 offset1:9\n\t"
                 "ds_read2st64_b64 %7, %[ad] offset0:10 offset1:11\n\t"
                 "ds_read2st64_b64 %8, %[ad] offset0:12 offset1:13\n\t"
                 "ds_read2st64_b64 %9, %[ad] offset0:14 offset1:15\n\t"
                 "ds_read2st64_b64 %10, %[ad] offset0:16 offset1:17\n\t"
                 "ds_read2st64_b64 %11, %[ad] offset0:18 offset1:19\n\t"
                 "ds_read2st64_b64 %12, %[ad] offset0:20 offset1:21\n\t"
                 "ds_read2st64_b64 %13, %[ad] offset0:22 offset1:23\n\t"
                 "ds_read2st64_b64 %14, %[ad] offset0:24 offset1:25\n\t"
                 "ds_read2st64_b64 %15, %[ad] offset0:26 offset1:27\n\t"
                 "s_waitcnt lgkmcnt(14)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk_ok_%=\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk_retry_%=\n"
                 ".Lsk_ok_%=:"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}


// the cautious reader: poll the sequence number ALONE (a miss costs one 4-byte LDS read, not the drain of all data
// reads), then issue the data reads -- they cannot be stale -- and consume them as they arrive
template <int NH>
__device__ __forceinline__ bool two_phase_reads(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[NH],
                                                unsigned long long &spins);

template <>
__device__ __forceinline__ bool two_phase_reads<4>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[4],
                                                     unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk2_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk2_ok_%=\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk2_retry_%=\n"
                 ".Lsk2_ok_%=:\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "s_nop 0"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

template <>
__device__ __forceinline__ bool two_phase_reads<7>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[7],
                                                     unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk2_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk2_ok_%=\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk2_retry_%=\n"
                 ".Lsk2_ok_%=:\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "ds_read2st64_b64 %6, %[ad] offset0:8 offset1:9\n\t"
                 "ds_read2st64_b64 %7, %[ad] offset0:10 offset1:11\n\t"
                 "ds_read2st64_b64 %8, %[ad] offset0:12 offset1:13\n\t"
                 "s_nop 0"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

template <>
__device__ __forceinline__ bool two_phase_reads<14>(uint32_t aseq, uint32_t adata, uint32_t expect, v2d (&p)[14],
                                                     unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk2_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk2_ok_%=\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk2_retry_%=\n"
                 ".Lsk2_ok_%=:\n\t"
                 "ds_read2st64_b64 %2, %[ad] offset0:0 offset1:1\n\t"
                 "ds_read2st64_b64 %3, %[ad] offset0:2 offset1:3\n\t"
                 "ds_read2st64_b64 %4, %[ad] offset0:4 offset1:5\n\t"
                 "ds_read2st64_b64 %5, %[ad] offset0:6 offset1:7\n\t"
                 "ds_read2st64_b64 %6, %[ad] offset0:8 offset1:9\n\t"
                 "ds_read2st64_b64 %7, %[ad] offset0:10 offset1:11\n\t"
                 "ds_read2st64_b64 %8, %[ad] offset0:12 offset1:13\n\t"
                 "ds_read2st64_b64 %9, %[ad] offset0:14 offset1:15\n\t"
                 "ds_read2st64_b64 %10, %[ad] offset0:16 offset1:17\n\t"
                 "ds_read2st64_b64 %11, %[ad] offset0:18 offset1:19\n\t"
                 "ds_read2st64_b64 %12, %[ad] offset0:20 offset1:21\n\t"
                 "ds_read2st64_b64 %13, %[ad] offset0:22 offset1:23\n\t"
                 "ds_read2st64_b64 %14, %[ad] offset0:24 offset1:25\n\t"
                 "ds_read2st64_b64 %15, %[ad] offset0:26 offset1:27\n\t"
                 "s_nop 0"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11]), "=&v"(p[12]), "=&v"(p[13])
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

// spin on (sequence number, one double)
__device__ __forceinline__ bool poll1(uint32_t aseq, uint32_t adata, uint32_t expect, double &v,
                                      unsigned long long &spins)
{
    uint32_t s, cnt;
    asm volatile("s_mov_b32 %[cnt], 0\n"
                 ".Lsk_p_retry_%=:\n\t"
                 "ds_read_b32 %[s], %[as]\n\t"
                 "ds_read_b64 %[v], %[ad]\n\t"
                 "s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_ne_u32_e32 vcc, %[ex], %[s]\n\t"
                 "s_cbranch_vccz .Lsk_p_ok_%=\n\t"
                 "s_add_u32 %[cnt], %[cnt], 1\n\t"
                 "s_cmp_lt_u32 %[cnt], %[cap]\n\t"
                 "s_cbranch_scc1 .Lsk_p_retry_%=\n"
                 ".Lsk_p_ok_%=:"
                 : [s] "=&v"(s), [cnt] "=&s"(cnt), [v] "=&v"(v)
                 : [as] "v"(aseq), [ad] "v"(adata), [ex] "s"(expect), [cap] "s"(SPIN_CAP)
                 : "vcc", "scc", "memory");
    spins += cnt;
    return cnt < SPIN_CAP;
}

template <int I, int NH, int WD>
__device__ __forceinline__ void consume(v2d (&p)[NH], double (&a)[8], double m, double c)
{
    if constexpr (I < NH) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(p[I]) : "n"(NH - 1 - I));
        a[(2 * I) & 7] += p[I].x;
        a[(2 * I + 1) & 7] += p[I].y;
        work<WD / NH>(a, m, c);
        consume<I + 1, NH, WD>(p, a, m, c);
    }
}

template <int I, int NH, int WG>
__device__ __forceinline__ void produce(uint32_t adata, double (&b)[8], double m, double c)
{
    if constexpr (I < NH) {
        work<WG / NH>(b, m, c);
        asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(adata), "v"(b[(2 * I) & 7]),
                     "v"(b[(2 * I + 1) & 7]), "n"(2 * I), "n"(2 * I + 1) : "memory");
        produce<I + 1, NH, WG>(adata, b, m, c);
    }
}

// recurrence (one-wave form: skew1s):  G(t): b[0] += thd_t; for i < NH: WG/NH FMAs, geo_{t+1}[2i, 2i+1] = b[2i&7], b[(2i+1)&7]
//                                      D(t): for i < NH: a[..] += geo_t[2i, 2i+1], WD/NH FMAs;  thd_{t+1} = a[0]
template <int WG, int WD, int NH, bool TWO_PHASE = false>
__global__ void __launch_bounds__(128) skew2s(Result *res, double *out, double seed, int iters)
{
    __shared__ double geo[2][2 * NH][64];
    __shared__ double thds[2][64];
    __shared__ uint32_t gseq[2][64], tseq[2][64];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), l = threadIdx.x % 64;
    const double m = 0.999999, c = 1e-7;
    if (w == 0) {
        for (int x = 0; x < 2 * NH; ++x) geo[0][x][l] = seed + x;
        gseq[0][l] = 1;
        gseq[1][l] = 0;
        thds[0][l] = seed;
        tseq[0][l] = 1;
        tseq[1][l] = 0;
    }
    __syncthreads();
    unsigned long long spins = 0;
    bool ok = true;
    double fin = 0.0;
    const uint32_t a_geo[2] = {(uint32_t)(size_t)&geo[0][0][l], (uint32_t)(size_t)&geo[1][0][l]};
    const uint32_t a_gs[2] = {(uint32_t)(size_t)&gseq[0][l], (uint32_t)(size_t)&gseq[1][l]};
    const uint32_t a_th[2] = {(uint32_t)(size_t)&thds[0][l], (uint32_t)(size_t)&thds[1][l]};
    const uint32_t a_ts[2] = {(uint32_t)(size_t)&tseq[0][l], (uint32_t)(size_t)&tseq[1][l]};
    unsigned long long t0 = now();
    if (w == 0) {   // ---- G
        double b[8];
        for (int i = 0; i < 8; ++i) b[i] = seed + 0.5 * i;
#pragma nounroll
        for (int t = 0; t < iters && ok; ++t) {
            double thd;
            ok = poll1(a_ts[t & 1], a_th[t & 1], (uint32_t)t + 1u, thd, spins);
            b[0] += thd;
            produce<0, NH, WG>(a_geo[(t + 1) & 1], b, m, c);
            lds_wseq(a_gs[(t + 1) & 1], (uint32_t)t + 2u);
        }
        fin = b[0];
    } else {        // ---- D
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = seed - 0.25 * i;
#pragma nounroll
        for (int t = 0; t < iters && ok; ++t) {
            v2d p[NH];
            ok = TWO_PHASE ? two_phase_reads<NH>(a_gs[t & 1], a_geo[t & 1], (uint32_t)t + 1u, p, spins)
                           : stream_reads<NH>(a_gs[t & 1], a_geo[t & 1], (uint32_t)t + 1u, p, spins);
            consume<0, NH, WD>(p, a, m, c);
            lds_w1(a_th[(t + 1) & 1], a[0]);
            lds_wseq(a_ts[(t + 1) & 1], (uint32_t)t + 2u);
        }
        fin = a[0];
    }
    unsigned long long t1 = now();
    Result &r = res[blockIdx.x];
    if (l == 0) {
        if (w == 0) {
            r.spins_g = spins;
            r.hw_g = hw_id();
            if (!ok) atomicOr(&r.fail, 1u);
        } else {
            r.cycles = t1 - t0;
            r.spins_d = spins;
            r.hw_d = hw_id();
            if (!ok) atomicOr(&r.fail, 2u);
        }
    }
    out[(size_t)blockIdx.x * 128 + threadIdx.x] = fin;
}

template <int WG, int WD, int NH>
__global__ void __launch_bounds__(64) skew1s(Result *res, double *out, double seed, int iters)
{
    const int l = threadIdx.x;
    const double m = 0.999999, c = 1e-7;
    double a[8], b[8], geo[2 * NH], thd = seed;
    for (int i = 0; i < 8; ++i) {
        b[i] = seed + 0.5 * i;
        a[i] = seed - 0.25 * i;
    }
    for (int x = 0; x < 2 * NH; ++x) geo[x] = seed + x;
    unsigned long long t0 = now();
#pragma nounroll
    for (int t = 0; t < iters; ++t) {
        b[0] += thd;
        double gn[2 * NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            work<WG / NH>(b, m, c);
            gn[2 * i] = b[(2 * i) & 7];
            gn[2 * i + 1] = b[(2 * i + 1) & 7];
        }
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            a[(2 * i) & 7] += geo[2 * i];
            a[(2 * i + 1) & 7] += geo[2 * i + 1];
            work<WD / NH>(a, m, c);
        }
        thd = a[0];
#pragma unroll
        for (int x = 0; x < 2 * NH; ++x) geo[x] = gn[x];
    }
    unsigned long long t1 = now();
    if (l == 0) {
        res[blockIdx.x].cycles = t1 - t0;
        res[blockIdx.x].hw_g = hw_id();
    }
    out[(size_t)blockIdx.x * 128 + l] = b[0];
    out[(size_t)blockIdx.x * 128 + 64 + l] = a[0];
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <typename K1, typename K2>
void run_pair(const char *tag, int WG, int WD, int NG, K1 k1, K2 k2, int blocks)
{
    Result *res;
    double *o1, *o2;
    CK(hipMalloc(&res, sizeof(Result) * blocks));
    CK(hipMalloc(&o1, 8 * 128 * blocks));
    CK(hipMalloc(&o2, 8 * 128 * blocks));
    std::vector<Result> h(blocks);
    double one = 0, two = 0;
    unsigned long long sg = 0, sd = 0;
    uint32_t fail = 0, same_simd = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(res, 0, sizeof(Result) * blocks));
        hipLaunchKernelGGL(k1, dim3(blocks), dim3(64), 0, 0, res, o1, 1.25, ITER);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), res, sizeof(Result) * blocks, hipMemcpyDeviceToHost));
        std::vector<double> c1;
        for (auto &r : h) c1.push_back((double)r.cycles / ITER);
        one = median(c1);
        CK(hipMemset(res, 0, sizeof(Result) * blocks));
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(128), 0, 0, res, o2, 1.25, ITER);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), res, sizeof(Result) * blocks, hipMemcpyDeviceToHost));
        std::vector<double> c2;
        sg = sd = 0;
        fail = same_simd = 0;
        for (auto &r : h) {
            c2.push_back((double)r.cycles / ITER);
            sg += r.spins_g;
            sd += r.spins_d;
            fail |= r.fail;
            // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13]
            if (((r.hw_g >> 4) & 3) == ((r.hw_d >> 4) & 3)) ++same_simd;
        }
        two = median(c2);
    }
    // same recurrence -> same bits
    std::vector<double> a(128 * blocks), b(128 * blocks);
    CK(hipMemcpy(a.data(), o1, 8 * 128 * blocks, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), o2, 8 * 128 * blocks, hipMemcpyDeviceToHost));
    const bool same = memcmp(a.data(), b.data(), 8 * 128 * blocks) == 0;
    printf("%s G %3d D %3d FMAs, %2d doubles G->D, %4d workgroups: one wave %7.1f | two waves %7.1f ticks/step "
           "(x %.3f)  spins/step G %.2f D %.2f  same-SIMD pairs %u  %s%s\n",
           tag, WG, WD, NG, blocks, one, two, two / one, (double)sg / blocks / ITER, (double)sd / blocks / ITER,
           same_simd, same ? "bits equal" : "BITS DIFFER", fail ? "  POLL GAVE UP" : "");
    fflush(stdout);
    CK(hipFree(res));
    CK(hipFree(o1));
    CK(hipFree(o2));
}

template <int WG, int WD, int NG>
void run_skew(int blocks)
{
    run_pair("simple  ", WG, WD, NG, skew1<WG, WD, NG>, skew2<WG, WD, NG>, blocks);
}
template <int WG, int WD, int NH>
void run_skews(int blocks)
{
    run_pair("streamed", WG, WD, 2 * NH, skew1s<WG, WD, NH>, skew2s<WG, WD, NH>, blocks);
}

template <int WG, int WD, int NH>
void run_skewt(int blocks)
{
    run_pair("2-phase ", WG, WD, 2 * NH, skew1s<WG, WD, NH>, skew2s<WG, WD, NH, true>, blocks);
}

template <int NX>
void run_pingpong()
{
    Result *res;
    CK(hipMalloc(&res, sizeof(Result)));
    Result h;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(res, 0, sizeof(Result)));
        hipLaunchKernelGGL((pingpong<NX>), dim3(1), dim3(128), 0, 0, res, 1.0);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(&h, res, sizeof(Result), hipMemcpyDeviceToHost));
    printf("ping-pong, %2d doubles + sequence number each way: round trip %7.1f ticks -> one way X = %6.1f  "
           "(polls that missed per round trip: %.2f + %.2f; SIMD %u / %u)%s\n",
           NX, (double)h.cycles / ITER, (double)h.cycles / ITER / 2, (double)h.spins_g / ITER,
           (double)h.spins_d / ITER, (h.hw_g >> 4) & 3, (h.hw_d >> 4) & 3, h.fail ? "  POLL GAVE UP" : "");
    fflush(stdout);
    CK(hipFree(res));
}

int main()
{
    {
        unsigned long long *d, h[5];
        CK(hipMalloc(&d, 40));
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(lone_latency, dim3(1), dim3(64), 0, 0, d);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h, d, 40, hipMemcpyDeviceToHost));
        printf("lone wave: ds_read_b64 + wait %.1f | ds_write_b64 + wait %.1f | write then read + wait %.1f | "
               "sequence number + 14 doubles (8 LDS instructions) + wait %.1f ticks\n",
               (double)h[0] / ITER, (double)h[1] / ITER, (double)h[2] / ITER, (double)h[3] / ITER);
        CK(hipFree(d));
    }
    run_pingpong<1>();
    run_pingpong<2>();
    run_pingpong<8>();
    run_pingpong<14>();
    // n = 3-like splits (today: 113 instructions per step in one stream)
    run_skew<40, 72, 8>(1);
    run_skew<56, 56, 14>(1);
    run_skew<64, 48, 14>(1);
    run_skew<56, 56, 14>(128);
    run_skew<56, 56, 8>(128);
    // n = 6-like splits (today: 239 instructions per step in one stream)
    run_skew<64, 176, 8>(1);
    run_skew<120, 120, 14>(1);
    run_skew<140, 100, 14>(1);
    run_skew<120, 120, 14>(128);
    run_skew<120, 120, 14>(1024);
    // the limit: no work at all (2 P = 2 X + the LDS instructions themselves)
    run_skew<0, 0, 2>(1);
    run_skew<8, 8, 14>(1);
    // ---- streamed hand-over
    run_skews<0, 0, 4>(1);
    run_skews<0, 0, 7>(1);
    run_skews<0, 0, 14>(1);
    run_skews<56, 56, 4>(1);       // n = 3-like
    run_skews<56, 56, 7>(1);
    run_skews<42, 70, 7>(1);
    run_skews<56, 56, 7>(128);
    run_skews<112, 112, 7>(1);     // n = 6-like
    run_skews<126, 126, 14>(1);
    run_skews<98, 154, 14>(1);
    run_skews<154, 98, 14>(1);
    run_skews<126, 126, 14>(128);
    run_skews<126, 126, 14>(512);
    run_skews<126, 126, 14>(1024);
    run_skews<196, 196, 14>(1);    // n = 8-like
    // ---- two-phase reader (sequence number polled alone)
    run_skewt<0, 0, 4>(1);
    run_skewt<0, 0, 7>(1);
    run_skewt<0, 0, 14>(1);
    run_skewt<56, 56, 4>(1);
    run_skewt<56, 56, 7>(1);
    run_skewt<42, 70, 7>(1);
    run_skewt<112, 112, 7>(1);
    run_skewt<126, 126, 14>(1);
    run_skewt<98, 154, 14>(1);
    run_skewt<126, 126, 14>(128);
    run_skewt<196, 196, 14>(1);
    return 0;
}
