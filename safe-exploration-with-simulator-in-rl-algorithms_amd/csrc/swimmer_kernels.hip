// swimmer_kernels.hip -- gfx950 kernels + the C ABI of include/swimmer_hip.h.
//
// All arithmetic is fp64 on the vector ALUs; there is no MFMA (the largest contraction on this
// path is 8x8) and no LDS in the hot loops (neighbour data moves by DPP).
//
//   step_kernel<N,TWIN,NT>     one physics step, SoA in / SoA out, one env per lane; HBM-bound
//                              at large n_env (algorithmic traffic 16 (2n+2) + 8 (n-1) + 8 bytes
//                              per env-step); NT = nontemporal accesses for streaming batches.
//                              A two-envs-per-lane variant with 16-byte accesses measured SLOWER
//                              (4.87 vs 5.62 TB/s: 90 VGPRs, fewer loads in flight) and was dropped.
//   accel_kernel<N,TWIN>       accelerations only
//   rollout_oct3_kernel        n = 3, H steps in one launch, one segment per lane WITH LANE ROLES: two
//                              mirror quads per rollout (sine / cosine, Gdot_x / Gdot_y;
//                              swimmer_oct3.h): the latency form, instruction-issue bound; up to
//                              8192 rollouts (one wave per SIMD)
//   rollout_quad3_kernel       n = 3, one DPP quad per rollout (swimmer_quad3.h): 8193 .. 16384 rollouts
//   rollout_row_kernel<N>      n = 4..8, one segment per lane, one rollout per 16-lane DPP row
//                              (swimmer_row.h)
//   rollout_kernel<N,ARS,TWIN> any n, ONE ROLLOUT PER LANE: the throughput form for batches that
//                              fill the chip, and the only form of the twin model
//   ars_update_kernel          sigma_R, policy step, V2 statistics merge; pure latency between
//                              two rollout launches: one round of loads, then LDS only
//   traj_moments_kernel<D>     full first/second moments of a trajectory buffer; HBM-bound
//   + the native ARS iteration pipeline (sw_ars_pipeline_*: copy stream, progress flag, 4-slot
//     buffer ring; the covariance pass rides along in the next rollout launch, SideJob)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>

#include <chrono>
#include "../../include/swimmer_hip.h"
#include "swimmer_device.h"
#include "swimmer_quad3.h"
#include "swimmer_oct3.h"
#include "swimmer_row.h"
#include "swimmer_twin.h"

// Where a rollout kernel's hot loop starts inside a 64-byte line of code.  A lone wave's issue rate depends on
// it: the same instructions, byte for byte, ran 0.2267 and 0.2346 ms per launch (n = 3) after an unrelated
// change elsewhere in the file had moved the loop by 16 bytes, and the one-step loops of the row kernel lose
// 4-11 % when their head is not 8-byte aligned (profiles/r03_p_ab_n3.log, r03_p_loop_pad_sweep_*.log).
// SW_PIN_LOOP aligns the code that follows to a line and puts PAD s_nops (4 bytes each) behind the boundary;
// the pads below are the best of a sweep over 0..7 on the GPU (scripts/ab_probe.sh over builds with
// -DSW_OCT_LOOP_PAD=k -DSW_QUAD_LOOP_PAD=k -DSW_ROW_LOOP_PAD=k).  Sweep again after changing what lies
// between a pin and its loop.
#define SW_PIN_LOOP(PAD) asm volatile(".p2align 6\n\t.fill %0, 4, 0xbf800000" ::"n"(PAD))
#ifndef SW_OCT_UNROLL
#define SW_OCT_UNROLL 8   // steps per trip of the mirror-quad kernel's main loop (4 or 8)
#endif
// mirror-quad kernel, by what the loop carries (trajectory capture, V2 moment sums): every instantiation is
// its own code, with its own best offset (profiles/r03_x_inst_sweep.log); -DSW_OCT_LOOP_PAD=k overrides all four
#ifdef SW_OCT_LOOP_PAD
constexpr int oct_loop_pad(bool, bool) { return SW_OCT_LOOP_PAD; }
#else
constexpr int oct_loop_pad(bool traj, bool mom) { return traj ? (mom ? 5 : 6) : (mom ? 6 : 2); }
#endif
#ifndef SW_QUAD_LOOP_PAD
#define SW_QUAD_LOOP_PAD 0
#endif
// row kernel, n = 4..8; swept with capture + moments for every n, and for the other three forms at n = 6;
// -DSW_ROW_LOOP_PAD=k overrides all of them for a sweep
#ifdef SW_ROW_LOOP_PAD
constexpr int row_loop_pad(int, bool, bool) { return SW_ROW_LOOP_PAD; }
#else
constexpr int row_loop_pad(int n, bool traj, bool mom)
{
    if (n == 6 && !(traj && mom)) return traj ? 6 : 4;
    return n == 4 ? 5 : n == 5 ? 0 : n == 6 ? 5 : n == 7 ? 6 : 1;
}
#endif

// steps per trip of the quad kernel's loop (measurement knob; 2 measured +5 ns per step)
#ifndef SW_QUAD_UNROLL
#define SW_QUAD_UNROLL 4
#endif
// Cache policy of the trajectory stores of the segment-per-lane kernels (gfx940+ encoding:
// 1 = sc0, 2 = nt, 16 = sc1).  sc1 = device-scope write-through: the 65 MB a launch stores do not
// pile up as dirty lines in the XCDs' L2s, so the write-back at the end of the kernel -- which sits
// on the critical path rollout -> update -- is short.  Same-box A/B (profiles/r02_d_ab_store_policy.log):
// plain 0.2564 ms per launch / 0.2675 ms per iteration, nt 0.2550 / 0.2658, sc0 0.2566 / 0.2673,
// sc1 0.2520 / 0.2628 (sc0 + sc1 and sc1 + nt the same as sc1).  (The lane kernel's stores in the
// saturated regime gain nothing from nontemporal stores: 4.58 vs 4.55 TB/s, scripts/saturated_probe.py.)
#ifndef SW_TRAJ_STORE_AUX
#define SW_TRAJ_STORE_AUX 16
#endif
// n = 3 rollouts: the mirror-quad kernel (swimmer_oct3.h) by default, or the quad kernel
#ifndef SW_N3_DEFAULT_OCT
#define SW_N3_DEFAULT_OCT true
#endif

namespace {

constexpr int kWave = 64;
constexpr int kStepBlock = 256;
constexpr int kRollBlock = 64;   // one wave per workgroup: every wave gets a SIMD to itself
constexpr int kOctBlock = 128;   // mirror-quad kernel: 8 rollouts per wave, 16 (= one V2 moment row) per workgroup
constexpr int kMomGroup = 16;    // rollouts per V2 moment row (same partition in every kernel)
constexpr int64_t kQuadMaxRollouts = 16384;  // above this every SIMD already has a wave
constexpr int64_t kRowMaxRollouts = 8192;    // row kernel (n >= 4): 4 rollouts per wave
constexpr int kRowBlock = 256;               // 16 rollouts = one V2 moment row per workgroup
constexpr int kUpdBlock = 256;        // update kernel: threads per workgroup up to kUpdWideFrom directions ...
constexpr int kUpdBlockWide = 1024;   // ... and beyond (a thread's share of the directions stays short)
constexpr int32_t kUpdWideFrom = 1025;
constexpr int64_t kStepStreamBytes = (int64_t)256 << 20;   // beyond the Infinity Cache: nontemporal accesses
constexpr double kHalfPi = 1.57079632679489661923;  // math.pi / 2 (remy_swimmer_env.py:65)
constexpr double kTwinStart = 0.001;                // SwimmerEnvironment.cpp:41

sw::Consts make_consts(const sw_params *p)
{
    sw::Consts c;
    c.l = p->l_i;
    c.h = p->h;
    c.dirx = p->dir_x;
    c.diry = p->dir_y;
    c.kl_nm = p->k * p->l_i / ((double)p->n * p->m_i);
    c.h_kl_nm = c.h * c.kl_nm;
    c.six_k_m = 6.0 * p->k / p->m_i;
    c.kl_m = p->k * p->l_i / p->m_i;
    c.c12 = 12.0 / (p->m_i * p->l_i * p->l_i);
    return c;
}

sw::TwinConsts make_twin_consts(const sw_params *p)
{
    return sw::TwinConsts{p->l_i, p->m_i, p->k, p->h, p->dir_x, p->dir_y};
}

inline bool is_twin(const sw_params *p) { return (p->flags & SW_FLAG_MODEL_TWIN) != 0; }

// Argument check shared by the entry points and the internal launchers (touches no HIP state).
int validate_params(const sw_params *p)
{
    if (!p) return SW_ERR_NULL;
    if (p->n < 2 || p->n > SW_MAX_SEGMENTS) return SW_ERR_SEGMENTS;
    if (p->flags & ~(SW_FLAG_ROLLOUT_LANE | SW_FLAG_ROLLOUT_QUAD | SW_FLAG_MODEL_TWIN)) return SW_ERR_PARAM;
    if (!(p->l_i > 0.0) || !(p->m_i > 0.0) || !isfinite(p->l_i) || !isfinite(p->m_i) ||
        !isfinite(p->k) || !isfinite(p->h) || !isfinite(p->dir_x) || !isfinite(p->dir_y))
        return SW_ERR_PARAM;
    return SW_OK;
}

// First call of every PUBLIC entry point: drops, once, whatever error an earlier HIP call of this
// thread left behind (a failed call of the caller's, hipErrorNotReady from an event query, ...), so
// that launch_status() reports OUR launches and does not blame a stale error on them.  Internal
// launchers use validate_params(): clearing again in the middle of an entry point would discard
// the error of a launch the entry point itself made a moment earlier.
int check_params(const sw_params *p)
{
    (void)hipGetLastError();
    return validate_params(p);
}

// ------------------------------------------------------------------------------------
// TWIN selects the reference's native model (swimmer_twin.h) instead of the Gym model.
// NT = nontemporal loads and stores: for batches that stream through HBM (larger than the
// 256 MiB Infinity Cache) they measured +6..8 % (16.8 M envs: 5.91 -> 6.37 TB/s); smaller
// batches keep plain accesses so that a step loop stays cache resident.
template <bool NT> __device__ __forceinline__ double ld_f64(const double *p)
{
    return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ __forceinline__ void st_f64(double v, double *p)
{
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#define SW_LD(p) ld_f64<NT>(p)
#define SW_ST(v, p) st_f64<NT>(v, p)

template <int N, bool TWIN, bool NT>
__global__ void __launch_bounds__(kStepBlock)
step_kernel(sw::Consts C, sw::TwinConsts T, int64_t n_env, const double *__restrict__ sin_,
            const double *__restrict__ act, double *__restrict__ sout,
            double *__restrict__ reward, int32_t *__restrict__ status)
{
    constexpr int M = N - 1;
    const int64_t e = (int64_t)blockIdx.x * kStepBlock + threadIdx.x;
    if (e >= n_env) return;
    double gdx = SW_LD(&sin_[e]), gdy = SW_LD(&sin_[n_env + e]);
    double th[N], thd[N], u[M];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        th[i] = SW_LD(&sin_[(int64_t)(2 + 2 * i) * n_env + e]);
        thd[i] = SW_LD(&sin_[(int64_t)(3 + 2 * i) * n_env + e]);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) u[i] = SW_LD(&act[(int64_t)i * n_env + e]);
    double r;
    const bool in_range = sw::track_angle_range<N>(0.0, th) < sw::kAngleLimit;
    const bool ok = TWIN ? sw::twin_step<N>(T, gdx, gdy, th, thd, u, r)
                         : sw::euler_step<N>(C, gdx, gdy, th, thd, u, r);
    if (!in_range) {   // outside sincos_fast's range: NaN out, SW_STATUS_RANGE
        gdx = gdy = r = __builtin_nan("");
#pragma unroll
        for (int i = 0; i < N; ++i) th[i] = thd[i] = __builtin_nan("");
    }
    SW_ST(gdx, &sout[e]);
    SW_ST(gdy, &sout[n_env + e]);
    bool fin = isfinite(gdx) && isfinite(gdy);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        SW_ST(th[i], &sout[(int64_t)(2 + 2 * i) * n_env + e]);
        SW_ST(thd[i], &sout[(int64_t)(3 + 2 * i) * n_env + e]);
        fin = fin && isfinite(th[i]) && isfinite(thd[i]);
    }
    if (reward) SW_ST(r, &reward[e]);
    if (status)
        status[e] = (ok ? 0 : SW_STATUS_SINGULAR) | (fin ? 0 : SW_STATUS_NONFINITE) |
                    (in_range ? 0 : SW_STATUS_RANGE);
}

// One physics step per stored transition, COMPARED with the stored next state instead of written out: the
// estimator's objective I(x) (ars/estimator.py:36-62) is the sum over every stored transition of
// || sim_step(s_t, a_t) - s_{t+1} ||_2.  Reads state + action + stored next state (16 d + 8 m bytes per transition),
// writes ONE double per workgroup: the fixed-order sum of its transitions' distances (lanes by shuffle tree, the four
// waves in order) -- deterministic, and the d doubles per transition the step kernel would store, the difference
// kernel would read back and the norm kernel would reduce never exist.  Out-of-range angles give NaN (as in step_kernel).
template <int N, bool NT>
__global__ void __launch_bounds__(kStepBlock)
step_residual_kernel(sw::Consts C, int64_t n_env, const double *__restrict__ sin_, const double *__restrict__ act,
                     const double *__restrict__ next_ref, double *__restrict__ partial)
{
    constexpr int M = N - 1;
    __shared__ double wsum[kStepBlock / kWave];
    const int64_t e = (int64_t)blockIdx.x * kStepBlock + threadIdx.x;
    double dist = 0.0;
    if (e < n_env) {
        double gdx = SW_LD(&sin_[e]), gdy = SW_LD(&sin_[n_env + e]);
        double th[N], thd[N], u[M];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            th[i] = SW_LD(&sin_[(int64_t)(2 + 2 * i) * n_env + e]);
            thd[i] = SW_LD(&sin_[(int64_t)(3 + 2 * i) * n_env + e]);
        }
#pragma unroll
        for (int i = 0; i < M; ++i) u[i] = SW_LD(&act[(int64_t)i * n_env + e]);
        // the stored next state: its loads are in flight while the step is computed
        double rx = SW_LD(&next_ref[e]), ry = SW_LD(&next_ref[n_env + e]), rth[N], rthd[N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            rth[i] = SW_LD(&next_ref[(int64_t)(2 + 2 * i) * n_env + e]);
            rthd[i] = SW_LD(&next_ref[(int64_t)(3 + 2 * i) * n_env + e]);
        }
        const bool in_range = sw::track_angle_range<N>(0.0, th) < sw::kAngleLimit;
        double r;
        (void)sw::euler_step<N>(C, gdx, gdy, th, thd, u, r);
        double q = (gdx - rx) * (gdx - rx);
        q = __builtin_fma(gdy - ry, gdy - ry, q);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            q = __builtin_fma(th[i] - rth[i], th[i] - rth[i], q);
            q = __builtin_fma(thd[i] - rthd[i], thd[i] - rthd[i], q);
        }
        dist = in_range ? sqrt(q) : __builtin_nan("");
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) dist += __shfl_down(dist, off, kWave);
    if (threadIdx.x % kWave == 0) wsum[threadIdx.x / kWave] = dist;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = wsum[0];
#pragma unroll
        for (int w = 1; w < kStepBlock / kWave; ++w) t += wsum[w];
        partial[blockIdx.x] = t;
    }
}

template <int N, bool TWIN>
__global__ void __launch_bounds__(kStepBlock)
accel_kernel(sw::Consts C, sw::TwinConsts T, int64_t n_env, const double *__restrict__ sin_,
             const double *__restrict__ act, double *__restrict__ gdd, double *__restrict__ tdd)
{
    constexpr int M = N - 1;
    const int64_t e = (int64_t)blockIdx.x * kStepBlock + threadIdx.x;
    if (e >= n_env) return;
    double gdx = sin_[e], gdy = sin_[n_env + e];
    double th[N], thd[N], u[M], a[N], ax, ay;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        th[i] = sin_[(int64_t)(2 + 2 * i) * n_env + e];
        thd[i] = sin_[(int64_t)(3 + 2 * i) * n_env + e];
    }
#pragma unroll
    for (int i = 0; i < M; ++i) u[i] = act[(int64_t)i * n_env + e];
    if (TWIN) sw::accelerations_twin<N>(T, gdx, gdy, th, thd, u, ax, ay, a);
    else sw::accelerations<N>(C, gdx, gdy, th, thd, u, ax, ay, a);
    if (!(sw::track_angle_range<N>(0.0, th) < sw::kAngleLimit)) {
        ax = ay = __builtin_nan("");
#pragma unroll
        for (int i = 0; i < N; ++i) a[i] = __builtin_nan("");
    }
    gdd[e] = ax;
    gdd[n_env + e] = ay;
#pragma unroll
    for (int i = 0; i < N; ++i) tdd[(int64_t)i * n_env + e] = a[i];
}

// Gym reset: Gdot = 0, theta = pi/2, thetadot = 0 (remy_swimmer_env.py:64-66); the native
// twin's env_start sets every observation entry to 0.001 (SwimmerEnvironment.cpp:39-42).
__global__ void reset_kernel(int n, int twin, int64_t n_env, double *__restrict__ state)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    state[e] = twin ? kTwinStart : 0.0;
    state[n_env + e] = twin ? kTwinStart : 0.0;
    for (int i = 0; i < n; ++i) {
        state[(int64_t)(2 + 2 * i) * n_env + e] = twin ? kTwinStart : kHalfPi;
        state[(int64_t)(3 + 2 * i) * n_env + e] = twin ? kTwinStart : 0.0;
    }
}

// One swimmer handed over in HOST memory (sw_env1, the batch-1 drop-in surfaces).  io = the
// handle's pinned, device-mapped block (SW_ENV1_* offsets).  One wave: lane l pulls double l of
// [state | action] -- ONE read burst over the bus instead of 25 round trips -- every lane then
// runs the same step on broadcast copies, lane 0 posts the results and, behind a system-scope
// fence, the sequence number the host spins on.
template <int N, bool TWIN, bool ACCEL>
__global__ void __launch_bounds__(kWave)
env1_kernel(sw::Consts C, sw::TwinConsts T, double *__restrict__ io, int32_t *__restrict__ status,
            uint32_t *__restrict__ seq_flag, uint32_t seq)
{
    constexpr int D = 2 * N + 2, M = N - 1;
    const int lane = threadIdx.x;
    const double mine = (lane < SW_ENV1_ACTION + M) ? io[lane] : 0.0;   // state 0..17, action 18..24
    double gdx = __shfl(mine, 0, kWave), gdy = __shfl(mine, 1, kWave);
    double th[N], thd[N], u[M > 0 ? M : 1];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        th[i] = __shfl(mine, 2 + 2 * i, kWave);
        thd[i] = __shfl(mine, 3 + 2 * i, kWave);
    }
#pragma unroll
    for (int i = 0; i < M; ++i) u[i] = __shfl(mine, SW_ENV1_ACTION + i, kWave);
    const bool in_range = sw::track_angle_range<N>(0.0, th) < sw::kAngleLimit;
    if (ACCEL) {
        double ax, ay, a[N];
        if (TWIN) sw::accelerations_twin<N>(T, gdx, gdy, th, thd, u, ax, ay, a);
        else sw::accelerations<N>(C, gdx, gdy, th, thd, u, ax, ay, a);
        if (lane == 0) {
            io[SW_ENV1_GDD] = in_range ? ax : __builtin_nan("");
            io[SW_ENV1_GDD + 1] = in_range ? ay : __builtin_nan("");
#pragma unroll
            for (int i = 0; i < N; ++i) io[SW_ENV1_TDD + i] = in_range ? a[i] : __builtin_nan("");
        }
    } else {
        double r;
        const bool ok = TWIN ? sw::twin_step<N>(T, gdx, gdy, th, thd, u, r)
                             : sw::euler_step<N>(C, gdx, gdy, th, thd, u, r);
        if (!in_range) {
            gdx = gdy = r = __builtin_nan("");
#pragma unroll
            for (int i = 0; i < N; ++i) th[i] = thd[i] = __builtin_nan("");
        }
        bool fin = isfinite(gdx) && isfinite(gdy);
#pragma unroll
        for (int i = 0; i < N; ++i) fin = fin && isfinite(th[i]) && isfinite(thd[i]);
        if (lane == 0) {
            io[SW_ENV1_NEXT] = gdx;
            io[SW_ENV1_NEXT + 1] = gdy;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                io[SW_ENV1_NEXT + 2 + 2 * i] = th[i];
                io[SW_ENV1_NEXT + 3 + 2 * i] = thd[i];
            }
            io[SW_ENV1_REWARD] = r;
            *status = (ok ? 0 : SW_STATUS_SINGULAR) | (fin ? 0 : SW_STATUS_NONFINITE) |
                      (in_range ? 0 : SW_STATUS_RANGE);
        }
    }
    static_assert(D <= SW_ENV1_ACTION && SW_ENV1_ACTION + M <= SW_ENV1_NEXT && SW_ENV1_NEXT + D <= SW_ENV1_REWARD,
                  "I/O block layout");
    if (lane == 0) {
        __threadfence_system();   // the results are visible to the host before the sequence number is
        __hip_atomic_store(seq_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------
// Rollouts.  ARS = false: policies[r][m][d] given per rollout.  ARS = true: rollout r is
// direction dir_begin + (r >> 1) with sign + (r even) / - (r odd); its policy
// P +- nu * delta is built here (ars_agent.py:141-142), so the perturbed policies never
// exist in HBM.  The V2 whitening P diag(inv_std) (ars/environment.py:32-33) is folded into
// the register copy of the policy once per rollout instead of once per step.
template <int N, bool ARS, bool TWIN>
__global__ void __launch_bounds__(kRollBlock)
rollout_kernel(sw::Consts C, sw::TwinConsts T, int64_t n_roll, int32_t H, const double *__restrict__ policies,
               const double *__restrict__ deltas, int64_t dir_begin, double nu,
               const double *__restrict__ mean, const double *__restrict__ inv_std,
               const double *__restrict__ state0, double *__restrict__ returns,
               double *__restrict__ traj, double *__restrict__ final_state,
               double *__restrict__ moments, int32_t *__restrict__ status)
{
    constexpr int D = 2 * N + 2, M = N - 1;
    const int64_t r = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    const bool active = r < n_roll;
    const bool v2 = (mean != nullptr);

    double m1[D], m2[D];  // V2 moment sums of (s - c), c = reset state
#pragma unroll
    for (int j = 0; j < D; ++j) m1[j] = m2[j] = 0.0;

    if (active) {
        // ---- policy into registers ----
        double W[M][D];
        if (ARS) {
            const int64_t dir = dir_begin + (r >> 1);
            const double sgn = (r & 1) ? -1.0 : 1.0;
            const double *dl = deltas + dir * (M * D);
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const double t = __dmul_rn(nu, dl[i * D + j]);
                    W[i][j] = __dadd_rn(policies[i * D + j], sgn * t);
                }
        } else {
            const double *pl = policies + r * (M * D);
#pragma unroll
            for (int i = 0; i < M; ++i)
#pragma unroll
                for (int j = 0; j < D; ++j) W[i][j] = pl[i * D + j];
        }
        if (v2) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const double sc = inv_std[j];
#pragma unroll
                for (int i = 0; i < M; ++i) W[i][j] = __dmul_rn(W[i][j], sc);
            }
        }
        // action = W (s - mu) = W s - W mu: the constant part once per rollout
        double nbias[M];
#pragma unroll
        for (int i = 0; i < M; ++i) {
            nbias[i] = 0.0;
            if (v2) {
#pragma unroll
                for (int j = 0; j < D; ++j) nbias[i] = __builtin_fma(-W[i][j], mean[j], nbias[i]);
            }
        }

        // ---- start state ----
        double gdx, gdy, th[N], thd[N];
        if (state0) {
            gdx = state0[r];
            gdy = state0[n_roll + r];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                th[i] = state0[(int64_t)(2 + 2 * i) * n_roll + r];
                thd[i] = state0[(int64_t)(3 + 2 * i) * n_roll + r];
            }
        } else {
            gdx = gdy = TWIN ? kTwinStart : 0.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                th[i] = TWIN ? kTwinStart : kHalfPi;
                thd[i] = TWIN ? kTwinStart : 0.0;
            }
        }

        double total = 0.0;
        bool ok = true;
        double thmax = 0.0;  // largest |theta| fed to sincos_fast
        for (int32_t t = 0; t < H; ++t) {
            thmax = sw::track_angle_range<N>(thmax, th);
            // action = W (s - mu)   (ars/environment.py:29 / :34); two partial sums
            double sm[D];
            sm[0] = gdx;
            sm[1] = gdy;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                sm[2 + 2 * i] = th[i];
                sm[3 + 2 * i] = thd[i];
            }
            double u[M];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                double a0 = __builtin_fma(W[i][0], sm[0], nbias[i]), a1 = W[i][1] * sm[1];
#pragma unroll
                for (int j = 2; j < D; j += 2) {
                    a0 = __builtin_fma(W[i][j], sm[j], a0);
                    a1 = __builtin_fma(W[i][j + 1], sm[j + 1], a1);
                }
                u[i] = a0 + a1;
            }
            double rew;
            ok = (TWIN ? sw::twin_step<N>(T, gdx, gdy, th, thd, u, rew)
                       : sw::euler_step<N>(C, gdx, gdy, th, thd, u, rew)) && ok;
            total += rew;
            if (traj) {
                double *tp = traj + (int64_t)t * D * n_roll + r;
                tp[0] = gdx;
                tp[n_roll] = gdy;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    tp[(int64_t)(2 + 2 * i) * n_roll] = th[i];
                    tp[(int64_t)(3 + 2 * i) * n_roll] = thd[i];
                }
            }
            if (moments) {
                m1[0] += gdx;
                m2[0] = __builtin_fma(gdx, gdx, m2[0]);
                m1[1] += gdy;
                m2[1] = __builtin_fma(gdy, gdy, m2[1]);
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double a = th[i] - kHalfPi;
                    m1[2 + 2 * i] += a;
                    m2[2 + 2 * i] = __builtin_fma(a, a, m2[2 + 2 * i]);
                    m1[3 + 2 * i] += thd[i];
                    m2[3 + 2 * i] = __builtin_fma(thd[i], thd[i], m2[3 + 2 * i]);
                }
            }
        }
        bool fin = isfinite(gdx) && isfinite(gdy);
#pragma unroll
        for (int i = 0; i < N; ++i) fin = fin && isfinite(th[i]) && isfinite(thd[i]);
        const bool in_range = thmax < sw::kAngleLimit;
        // an angle outside sincos_fast's range makes every later number meaningless: fail
        // loudly (NaN return + status bit) instead of returning finite garbage
        returns[r] = in_range ? total : __builtin_nan("");
        if (status)
            status[r] = (ok ? 0 : SW_STATUS_SINGULAR) | (fin ? 0 : SW_STATUS_NONFINITE) |
                        (in_range ? 0 : SW_STATUS_RANGE);
        if (final_state) {
            final_state[r] = gdx;
            final_state[n_roll + r] = gdy;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                final_state[(int64_t)(2 + 2 * i) * n_roll + r] = th[i];
                final_state[(int64_t)(3 + 2 * i) * n_roll + r] = thd[i];
            }
        }
    }

    if (moments) {
        // fixed-order butterfly over each group of 16 lanes (deterministic); one row of
        // partial sums per 16 rollouts, the same partition the quad kernel produces
        const int64_t n_rows = (n_roll + kMomGroup - 1) / kMomGroup;
        const int64_t row = (int64_t)blockIdx.x * (kRollBlock / kMomGroup) + threadIdx.x / kMomGroup;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            double a = m1[j], b = m2[j];
#pragma unroll
            for (int off = kMomGroup / 2; off > 0; off >>= 1) {
                a += __shfl_down(a, off, kMomGroup);
                b += __shfl_down(b, off, kMomGroup);
            }
            if (threadIdx.x % kMomGroup == 0 && row < n_rows) {
                moments[row * (2 * D) + j] = a;
                moments[row * (2 * D) + D + j] = b;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Safe exploration (safe_ars/ars.py:101-153): every real step of a rollout is gated by a ONE-STEP look-ahead in a
// simulator -- `isSafe` = sim_env.set_state(obs) + sim_env.step(action) + cost(sim obs) <= sim_thresh (:111-122,
// called at :141).  One rollout per lane, the whole H-step loop in one launch: per step the action (policy @ obs,
// :139), one Euler step with the SIMULATOR's constants on a copy of the state, the cost of where that lands, and --
// if the gate is open -- the real step.  A refused step leaves the real env where it is (:150-151), so the same
// action is proposed and refused for the rest of the horizon: the lane stops stepping and only repeats its state
// into the trajectory.  Costs (include/swimmer_hip.h SW_COST_*): |obs[j]|, or max_i |thetadot_i| (the reference's
// own experiment, safe_ars/experiment.py:45).
template <int N>
__device__ __forceinline__ double safe_cost(int32_t kind, int32_t index, double gdx, double gdy,
                                            const double (&th)[N], const double (&thd)[N])
{
    if (kind == SW_COST_MAX_ABS_THETADOT) {
        double c = fabs(thd[0]);
#pragma unroll
        for (int i = 1; i < N; ++i) c = fmax(c, fabs(thd[i]));   // np.max: NaN handled by the caller's <= test
        bool nan = false;
#pragma unroll
        for (int i = 0; i < N; ++i) nan = nan || (thd[i] != thd[i]);
        return nan ? __builtin_nan("") : c;                      // np.max propagates NaN, fmax would drop it
    }
    double v = (index == 0) ? gdx : gdy;                         // |obs[index]|, obs = [Gdx, Gdy, th_1, thd_1, ...]
#pragma unroll
    for (int i = 0; i < N; ++i) {
        v = (index == 2 + 2 * i) ? th[i] : v;
        v = (index == 3 + 2 * i) ? thd[i] : v;
    }
    return fabs(v);
}

template <int N>
__global__ void __launch_bounds__(kRollBlock)
safe_rollout_kernel(sw::Consts Creal, sw::Consts Csim, int64_t n_roll, int32_t H,
                    const double *__restrict__ policies, int32_t cost_kind, int32_t cost_index,
                    double sim_thresh, double real_thresh, double *__restrict__ returns,
                    double *__restrict__ traj, int32_t *__restrict__ first_refused,
                    int32_t *__restrict__ violations, int32_t *__restrict__ status)
{
    constexpr int D = 2 * N + 2, M = N - 1;
    const int64_t r = (int64_t)blockIdx.x * kRollBlock + threadIdx.x;
    if (r >= n_roll) return;
    double W[M][D];
    const double *pl = policies + r * (M * D);
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) W[i][j] = pl[i * D + j];
    double gdx = 0.0, gdy = 0.0, th[N], thd[N];      // real_env.reset() (:133)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        th[i] = kHalfPi;
        thd[i] = 0.0;
    }
    auto record = [&](int32_t t) {
        double *tp = traj + (int64_t)t * D * n_roll + r;
        tp[0] = gdx;
        tp[n_roll] = gdy;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            tp[(int64_t)(2 + 2 * i) * n_roll] = th[i];
            tp[(int64_t)(3 + 2 * i) * n_roll] = thd[i];
        }
    };
    double total = 0.0, thmax = 0.0;
    bool ok = true;
    int32_t refused_at = H, over = 0;
    for (int32_t t = 0; t < H; ++t) {
        thmax = sw::track_angle_range<N>(thmax, th);
        double sm[D];
        sm[0] = gdx;
        sm[1] = gdy;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sm[2 + 2 * i] = th[i];
            sm[3 + 2 * i] = thd[i];
        }
        double u[M];                                  // ac = policy @ obs (:139)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            double a0 = W[i][0] * sm[0], a1 = W[i][1] * sm[1];
#pragma unroll
            for (int j = 2; j < D; j += 2) {
                a0 = __builtin_fma(W[i][j], sm[j], a0);
                a1 = __builtin_fma(W[i][j + 1], sm[j + 1], a1);
            }
            u[i] = a0 + a1;
        }
        // the simulator's look-ahead from the real state (:120-121)
        double sgx = gdx, sgy = gdy, sth[N], sthd[N], srew;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            sth[i] = th[i];
            sthd[i] = thd[i];
        }
        (void)sw::euler_step<N>(Csim, sgx, sgy, sth, sthd, u, srew);
        if (!(safe_cost<N>(cost_kind, cost_index, sgx, sgy, sth, sthd) <= sim_thresh)) {   // :122, NaN refuses
            refused_at = t;
            break;
        }
        double rew;
        ok = sw::euler_step<N>(Creal, gdx, gdy, th, thd, u, rew) && ok;                     // :142
        total += rew;
        over += (safe_cost<N>(cost_kind, cost_index, gdx, gdy, th, thd) > real_thresh) ? 1 : 0;   // :143-144
        if (traj) record(t);
    }
    if (traj)
        for (int32_t t = refused_at; t < H; ++t) record(t);      // :151: the unchanged state, step after step
    bool fin = isfinite(gdx) && isfinite(gdy);
#pragma unroll
    for (int i = 0; i < N; ++i) fin = fin && isfinite(th[i]) && isfinite(thd[i]);
    const bool in_range = thmax < sw::kAngleLimit;
    returns[r] = in_range ? total : __builtin_nan("");
    if (first_refused) first_refused[r] = refused_at;
    if (violations) violations[r] = over;
    if (status)
        status[r] = (ok ? 0 : SW_STATUS_SINGULAR) | (fin ? 0 : SW_STATUS_NONFINITE) | (in_range ? 0 : SW_STATUS_RANGE);
}

constexpr int kMomBlock = 256;
constexpr int kMomTChunk = 32;  // steps per 256-thread tile (8 measured slower: less work per workgroup)
constexpr uint32_t kCovMaxTiles = 4096;   // tiles per pass: bounds the fixed-order merge

// Tiling of a covariance pass over traj [H][D][n_roll] for workgroups of `block` threads:
// nbx tiles of `block` rollouts x ny tiles of tchunk steps.  A function of (n_roll, H, block)
// only, so a pass always sums in the same order (bit-reproducible results).
struct CovTiling {
    uint32_t nbx, ny;
    int32_t tchunk;
};

// Long chains (D >= 12, n >= 5) in 256-thread workgroups: the D (D + 1) / 2 + D sums do not fit one
// lane's registers, so the four waves of the workgroup work on the SAME 64 rollouts and split the
// sums between them (moments_split): a tile is 64 rollouts wide and four times as long.
__host__ __device__ constexpr bool cov_split(int D, int block) { return D >= 12 && block >= 256; }

// riding: the pass rides along in a rollout launch (or is the flush of a pass owed to one: same tiling,
// same order of summation).  Otherwise it is the standalone pass of sw_traj_moments_f64, alone on the chip.
CovTiling cov_tiling(int64_t n_roll, int32_t H, int block, int D, bool riding)
{
    CovTiling t;
    const int cols = cov_split(D, block) ? kWave : block;   // rollouts per tile
    t.nbx = (uint32_t)((n_roll + cols - 1) / cols);
    // steps per tile: long tiles for the one-wave workgroups of the quad kernel's launches -- a tile
    // ends with a cross-lane reduction of all its accumulators, and once a batch puts a wave on
    // every SIMD those epilogues are the rollouts' time (2048 directions on one GPU, n = 3: launch
    // 0.2798 ms with 64 steps per tile, 0.2663 with 128, 0.384 with 32; no difference at 512
    // directions; profiles/r02_f_cov_tile_sweep.log)
    int32_t base = (block >= kMomBlock) ? (cov_split(D, block) ? 4 * kMomTChunk : kMomTChunk) : 128;
    if (!riding) {
        // alone on the chip the pass is fastest with ~one workgroup per CU for the split tiles, one per two
        // CUs for the wide ones -- fewer leave CUs idle, more lengthen the merge (0.06 us per tile):
        // profiles/r03_p_cov_tchunk_sweep.log
        const int64_t target = cov_split(D, block) ? 256 : 128;
        const int64_t want = ((int64_t)H * t.nbx + target - 1) / target;
        base = (int32_t)(want < 8 ? 8 : (want > H ? H : want));
    }
    static const char *env = getenv("SWIMMER_COV_TCHUNK");   // measurement knob
    if (env && atoi(env) > 0) base = atoi(env);
    const uint32_t ny_max = (uint32_t)((H + base - 1) / base);
    const uint32_t cap = kCovMaxTiles / t.nbx > 0 ? kCovMaxTiles / t.nbx : 1u;
    const uint32_t ny = ny_max < cap ? ny_max : cap;
    t.tchunk = (int32_t)((H + (int32_t)ny - 1) / (int32_t)ny);
    t.ny = (uint32_t)((H + t.tchunk - 1) / t.tchunk);
    return t;
}

// acc buffer of a covariance pass: [count | sum x (D) | sum x x^T (D x D)] followed by the
// pass's scratch: one ticket counter (a double slot whose first 4 bytes are the counter; all-zero
// bits = 0) and one row of D + D*D partial sums per tile.
__host__ __device__ constexpr int cov_sums(int D) { return 1 + D + D * D; }

// One tile of the full first / second moment sums of a trajectory buffer [H][D][n_roll]:
// BLOCK rollouts x the steps [t0, t1) (x = state - reset pivot), written to the tile's scratch
// row.  NO atomics on the sums: the tile that finishes last (ticket counter) adds all rows to acc
// in tile order, so the result does not depend on the order the tiles ran in.  Workgroups of any
// size that is a multiple of 64 can run it: the standalone traj_moments_kernel and the covariance
// workgroups that ride along in a rollout launch (SideJob).  For long chains the upper triangle is
// accumulated JB rows at a time (re-reading the tile from cache) so that the accumulators stay in
// registers.
// A tile's row of partial sums inside the pass's scratch.  The scratch is stored TRANSPOSED -- entry j of
// tile i at [j * n_tiles + i] -- so that the merge reads one entry of consecutive tiles with consecutive
// lanes (moments_tile); a tile's own stores are strided (fire and forget).
struct TileRow {
    double *p;
    int64_t stride;
    // agent-scope store (written through the XCD's L2): visible to the merging workgroup on another XCD once
    // the store has completed, without a write-back of everything else that is dirty in this L2
    __device__ __forceinline__ void put(int j, double v) const
    {
        __hip_atomic_store(p + (int64_t)j * stride, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

template <int D, int BLOCK, int J0, int JB>
__device__ __forceinline__ void moments_pass(int64_t n_roll, const double *__restrict__ traj,
                                             const TileRow tile_row, int64_t bx, int32_t t0, int32_t t1,
                                             double *sh /* [BLOCK / 64][D + JB * D] */)
{
    constexpr int NW = BLOCK / kWave, W = D + JB * D;
    constexpr int J1 = (J0 + JB < D) ? J0 + JB : D;   // rows [J0, J1) of the upper triangle
    const int64_t r = bx * BLOCK + threadIdx.x;
    const int w = threadIdx.x / kWave, l = threadIdx.x % kWave;
    double s1[D], s2[JB][D];
#pragma unroll
    for (int j = 0; j < D; ++j) s1[j] = 0.0;
#pragma unroll
    for (int a = 0; a < JB; ++a)
#pragma unroll
        for (int g = 0; g < D; ++g) s2[a][g] = 0.0;
    if (r < n_roll) {
        for (int32_t t = t0; t < t1; ++t) {
            const double *tp = traj + (int64_t)t * D * n_roll + r;
            double x[D];
#pragma unroll
            for (int j = J0; j < D; ++j) {   // later passes need columns >= J0 only
                const double c = (j >= 2 && (j & 1) == 0) ? kHalfPi : 0.0;
                x[j] = tp[(int64_t)j * n_roll] - c;
            }
            if (J0 == 0) {
#pragma unroll
                for (int j = 0; j < D; ++j) s1[j] += x[j];
            }
#pragma unroll
            for (int f = J0; f < J1; ++f)
#pragma unroll
                for (int g = f; g < D; ++g) s2[f - J0][g] = __builtin_fma(x[f], x[g], s2[f - J0][g]);
        }
    }
    __syncthreads();   // the previous pass has drained sh
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const bool live = (j < D) ? (J0 == 0) : (J0 + (j - D) / D < J1 && (j - D) % D >= J0 + (j - D) / D);
        if (!live) continue;   // compile-time after unrolling
        double v = (j < D) ? s1[j < D ? j : 0] : s2[(j >= D ? j - D : 0) / D][(j >= D ? j - D : 0) % D];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if (l == 0) sh[w * W + j] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < W; j += BLOCK) {
        const int f = J0 + (j - D) / D, g = (j - D) % D;
        const bool live = (j < D) ? (J0 == 0) : (f < J1 && g >= f);
        if (!live) continue;
        double v = 0.0;
        for (int i = 0; i < NW; ++i) v += sh[i * W + j];   // fixed order over the waves
        if (j < D) tile_row.put(j, v);
        else tile_row.put(D + f * D + g, v);                // upper triangle only
    }
}

template <int D, int BLOCK, int JB, int J0 = 0>
struct MomentsPasses {
    static __device__ __forceinline__ void run(int64_t n_roll, const double *__restrict__ traj,
                                               const TileRow tile_row, int64_t bx, int32_t t0,
                                               int32_t t1, double *sh)
    {
        if constexpr (J0 < D) {
            moments_pass<D, BLOCK, J0, JB>(n_roll, traj, tile_row, bx, t0, t1, sh);
            MomentsPasses<D, BLOCK, JB, J0 + JB>::run(n_roll, traj, tile_row, bx, t0, t1, sh);
        }
    }
};

// ---- long chains: the sums of a tile split over the four waves of the workgroup -----------------
// items 0 .. D-1 are the first moments, item D + p is pair p = (f, g), f <= g, of the upper triangle
// in row-major order.  Wave w owns items [w * PER, (w + 1) * PER): ~30 accumulators for D = 14
// instead of 119, so ONE pass over the tile suffices (the multi-pass form re-reads it 2-4 times),
// two steps' loads are in flight at a time, and the waves' loads of the same 64 rollouts hit in the
// vector L1 / L2 after the first one.  Each wave reduces its own sums over the lanes and writes them
// to the tile's row: no LDS, no barrier.
__host__ __device__ constexpr int pair_row(int D, int p)
{
    int f = 0;
    while (p >= D - f) {
        p -= D - f;
        ++f;
    }
    return f;
}
__host__ __device__ constexpr int pair_col(int D, int p)
{
    int f = 0;
    while (p >= D - f) {
        p -= D - f;
        ++f;
    }
    return f + p;
}

// item Q of the tile's sums, with every index a compile-time constant (a loop variable, even
// fully unrolled, left the pair lookup to the optimiser, which put x[] and acc[] in scratch)
template <int D, int Q>
__device__ __forceinline__ void moments_item_add(double &a, const double (&x)[D])
{
    if constexpr (Q < D) {
        a += x[Q];
    } else {
        constexpr int f = pair_row(D, Q - D), g = pair_col(D, Q - D);
        a = __builtin_fma(x[f], x[g], a);
    }
}

template <int D, int Q>
__device__ __forceinline__ void moments_item_store(double v, const TileRow tile_row)
{
    if constexpr (Q < D) {
        tile_row.put(Q, v);
    } else {
        constexpr int f = pair_row(D, Q - D), g = pair_col(D, Q - D);
        tile_row.put(D + f * D + g, v);   // upper triangle only
    }
}

template <int D, int WV, int... I>
__device__ __forceinline__ void moments_split_wave(int64_t n_roll, const double *__restrict__ traj,
                                                   const TileRow tile_row, int64_t r0, int32_t t0,
                                                   int32_t t1, int32_t nap, std::integer_sequence<int, I...>)
{
    constexpr int ITEMS = D + D * (D + 1) / 2, PER = (ITEMS + 3) / 4, Q0 = WV * PER;
    constexpr int CNT = (int)sizeof...(I);   // = min(PER, ITEMS - Q0)
    // the lowest state column this wave multiplies: columns below it are never loaded
    constexpr int JMIN = (Q0 < D) ? 0 : pair_row(D, Q0 - D);
    const int l = threadIdx.x % kWave;
    double acc[CNT];
#pragma unroll
    for (int q = 0; q < CNT; ++q) acc[q] = 0.0;
    // buffer loads: the step's slab base (wave-uniform, scalar arithmetic) is the resource's base, the
    // column is the scalar offset, the lane the 32-bit vector offset -- no per-lane 64-bit address
    // arithmetic in the loop.  One slab (D n_roll doubles) is < 4 GiB for every supported n_roll.
    const uint32_t lane_bytes = (uint32_t)l * 8u;
    const uint32_t col_bytes = (uint32_t)(n_roll * 8);
    auto load = [&](double (&x)[D], int32_t t) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(traj + ((int64_t)t * D * n_roll + r0)), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
        for (int j = JMIN; j < D; ++j) {
            const double c = (j >= 2 && (j & 1) == 0) ? kHalfPi : 0.0;
            typedef unsigned int v2u __attribute__((ext_vector_type(2)));
            union { v2u i; double d; } u;
            u.i = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_bytes, (int)((uint32_t)j * col_bytes), 0);
            x[j] = u.d - c;
        }
    };
    if (r0 + l < n_roll && t0 < t1) {
        // THREE steps' loads in flight (a riding wave is alone with its memory latency: the tile's time is
        // steps x latency / depth), no conditionals in the steady state (they cost register copies).  Every
        // accumulator still adds its steps in order: the sums do not depend on the depth.
        double xa[D], xb[D], xc[D];
        auto add = [&](const double (&x)[D]) { (moments_item_add<D, Q0 + I>(acc[I], x), ...); };
        int32_t t = t0;
        load(xa, t);
        if (t + 1 < t1) load(xb, t + 1);
        for (; t + 4 < t1; t += 3) {     // xa, xb hold steps t, t + 1
            load(xc, t + 2);
            add(xa);
            load(xa, t + 3);
            add(xb);
            load(xb, t + 4);
            add(xc);
            for (int32_t z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(16);   // measurement knob (SWIMMER_COV_NAP)
        }
        const int32_t left = t1 - t;     // 1..4 steps, xa (and xb if left >= 2) loaded
        if (left >= 3) load(xc, t + 2);
        add(xa);
        if (left >= 4) load(xa, t + 3);
        if (left >= 2) add(xb);
        if (left >= 3) add(xc);
        if (left >= 4) add(xa);
    }
#pragma unroll
    for (int q = 0; q < CNT; ++q) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, kWave);
    }
    if (l == 0) (moments_item_store<D, Q0 + I>(acc[I], tile_row), ...);
}

template <int D, int WV>
__device__ __forceinline__ void moments_split_part(int64_t n_roll, const double *__restrict__ traj,
                                                   const TileRow tile_row, int64_t r0, int32_t t0, int32_t t1,
                                                   int32_t nap)
{
    constexpr int ITEMS = D + D * (D + 1) / 2, PER = (ITEMS + 3) / 4, Q0 = WV * PER;
    constexpr int CNT = (Q0 + PER <= ITEMS) ? PER : ITEMS - Q0;
    moments_split_wave<D, WV>(n_roll, traj, tile_row, r0, t0, t1, nap, std::make_integer_sequence<int, CNT>{});
}

template <int D>
__device__ __forceinline__ void moments_split(int64_t n_roll, const double *__restrict__ traj, const TileRow tile_row,
                                              int64_t bx, int32_t t0, int32_t t1, int32_t nap)
{
    const int64_t r0 = bx * kWave;
    switch (__builtin_amdgcn_readfirstlane(threadIdx.x / kWave)) {   // scalar: the waves' addresses stay uniform
    case 0: moments_split_part<D, 0>(n_roll, traj, tile_row, r0, t0, t1, nap); break;
    case 1: moments_split_part<D, 1>(n_roll, traj, tile_row, r0, t0, t1, nap); break;
    case 2: moments_split_part<D, 2>(n_roll, traj, tile_row, r0, t0, t1, nap); break;
    default: moments_split_part<D, 3>(n_roll, traj, tile_row, r0, t0, t1, nap); break;
    }
}

// Tile `tile` of `n_tiles` (tile = by * nbx + bx).  acc = [sums | counter | n_tiles rows].
template <int D, int BLOCK>
__device__ __forceinline__ void moments_tile(int64_t n_roll, int32_t H, const double *__restrict__ traj,
                                             double *__restrict__ acc, int64_t bx, int32_t t0, int32_t t1,
                                             uint32_t tile, uint32_t n_tiles, int32_t nap = 0)
{
    // rows of the upper triangle per pass: as many as keep the accumulators (D + the rows' entries)
    // plus one state inside 256 VGPRs -- one pass up to D = 12, 2 / 3 / 4 passes for D = 14 / 16 / 18
    constexpr int JB = (D <= 12) ? D : (D == 14 ? 7 : (D == 16 ? 6 : 5));
    constexpr int W = D + D * D;
    __shared__ uint32_t ticket;
    double *rows = acc + cov_sums(D) + 1;              // [W][n_tiles]
    const TileRow row{rows + tile, (int64_t)n_tiles};
    if constexpr (cov_split(D, BLOCK)) {
        static_assert(BLOCK == 4 * kWave, "moments_split: four waves per workgroup");
        moments_split<D>(n_roll, traj, row, bx, t0, t1, nap);
    } else {
        __shared__ double sh[(BLOCK / kWave) * (D + JB * D)];
        MomentsPasses<D, BLOCK, JB>::run(n_roll, traj, row, bx, t0, t1, sh);
    }
    // The row's stores are agent-scope write-through stores (TileRow::put): once they have COMPLETED the row is
    // visible device-wide.  Every wave therefore waits for its own stores (s_waitcnt vmcnt(0): a workgroup-scope
    // release fence does NOT emit that wait on gfx950) before the barrier behind which thread 0 takes the ticket,
    // so the ticket can never become visible before the row has reached memory.  No device-scope release
    // here: it would write back the whole L2 -- in a rollout launch that is the trajectories -- per tile.
    // (tests/test_isa_contracts.py checks the wait in the built code.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0)
        ticket = atomicAdd(reinterpret_cast<uint32_t *>(acc + cov_sums(D)), 1u);
    __syncthreads();
    if (ticket != n_tiles - 1u) return;
    // Last tile to finish: add every tile's row to acc.  The merge is a chain of memory latencies (the
    // rows were written by other XCDs: every load misses), so it is laid out for loads in flight, not
    // for arithmetic: a wave takes kMergeEntries entries at a time, lane l of it the tiles l, l + 64, ...
    // of each (consecutive lanes = consecutive addresses), four interleaved partial sums per lane, then a
    // shuffle tree over the lanes; the totals meet in LDS and are added to acc by one thread per entry
    // (one more latency, not one per group).  The order depends on n_tiles only, never on which tile
    // ran last.
    __threadfence();
    constexpr int kMergeEntries = 8, NWV = BLOCK / kWave, ITEMS = D + D * (D + 1) / 2;
    __shared__ double merged[ITEMS];
    const int w = threadIdx.x / kWave, l = threadIdx.x % kWave;
    for (int q0 = w * kMergeEntries; q0 < ITEMS; q0 += NWV * kMergeEntries) {
        const double *col[kMergeEntries];
#pragma unroll
        for (int e = 0; e < kMergeEntries; ++e) {
            const int q = (q0 + e < ITEMS) ? q0 + e : ITEMS - 1;       // the last group repeats an entry
            const int j = (q < D) ? q : D + pair_row(D, q - D) * D + pair_col(D, q - D);
            col[e] = rows + (int64_t)j * n_tiles;
        }
        double a[kMergeEntries][4];
#pragma unroll
        for (int e = 0; e < kMergeEntries; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[e][i] = 0.0;
        uint32_t t = (uint32_t)l;
        for (; t + 3u * kWave < n_tiles; t += 4u * kWave) {
#pragma unroll
            for (int e = 0; e < kMergeEntries; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) a[e][i] += col[e][t + (uint32_t)(i * kWave)];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (t + (uint32_t)(i * kWave) < n_tiles) {
#pragma unroll
                for (int e = 0; e < kMergeEntries; ++e) a[e][i] += col[e][t + (uint32_t)(i * kWave)];
            }
        }
#pragma unroll
        for (int e = 0; e < kMergeEntries; ++e) {
            double v = (a[e][0] + a[e][1]) + (a[e][2] + a[e][3]);
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
            if (l == 0 && q0 + e < ITEMS) merged[q0 + e] = v;
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < ITEMS; q += BLOCK) {
        const double v = merged[q];
        if (q < D) {
            acc[1 + q] += v;
        } else {   // mirror into both halves
            const int f = pair_row(D, q - D), g = pair_col(D, q - D);
            acc[1 + D + f * D + g] += v;
            if (g != f) acc[1 + D + g * D + f] += v;
        }
    }
    if (threadIdx.x == 0) {
        acc[0] += (double)n_roll * (double)H;
        *reinterpret_cast<uint32_t *>(acc + cov_sums(D)) = 0u;   // ready for the next pass
    }
}

// standalone covariance pass: 1-D grid of nbx * ny tiles of BLOCK rollouts x tchunk steps
template <int D, int BLOCK>
__global__ void __launch_bounds__(BLOCK)
traj_moments_kernel(int64_t n_roll, int32_t H, const double *__restrict__ traj,
                    double *__restrict__ acc, uint32_t nbx, int32_t tchunk)
{
    const uint32_t bx = blockIdx.x % nbx, by = blockIdx.x / nbx;
    const int32_t t0 = (int32_t)by * tchunk;
    moments_tile<D, BLOCK>(n_roll, H, traj, acc, bx, t0, min(H, t0 + tchunk), blockIdx.x, gridDim.x);
}

// What a rollout launch of the ARS pipeline carries besides its rollouts (both optional):
//  * a progress flag: workgroup 0 stores flag_value to host-visible memory when it starts, i.e.
//    "everything enqueued on this stream before this launch has completed".  The host paces
//    itself on it, so the critical stream carries no event-record packets (measured: one costs
//    ~4 us between two kernels);
//  * the covariance pass over the PREVIOUS iteration's trajectories, run by extra workgroups
//    behind the rollout workgroups of the same grid: no second queue, no cross-queue events
//    (measured: a concurrent kernel on another queue costs the rollout launch ~5 us whatever
//    its size).  Those workgroups finish long before the rollouts do.
struct SideJob {
    uint32_t *flag;
    uint32_t flag_value;
    uint32_t first_cov_block;   // = number of rollout workgroups; UINT32_MAX: no covariance pass
    uint32_t cov_nbx;           // covariance tiles along the rollout axis
    uint32_t cov_tiles;         // covariance tiles in all
    int32_t cov_tchunk;         // steps per covariance tile
    int32_t cov_nap;            // s_sleep rounds per two steps of a split tile (load pacing)
    int32_t cov_H;
    int64_t cov_rolls;
    const double *cov_traj;
    double *cov_acc;
};

template <int D, int BLOCK>
__device__ __forceinline__ void side_cov_tile(const SideJob &sj)
{
    const uint32_t b = blockIdx.x - sj.first_cov_block;
    const uint32_t bx = b % sj.cov_nbx, by = b / sj.cov_nbx;
    const int32_t t0 = (int32_t)by * sj.cov_tchunk;
    moments_tile<D, BLOCK>(sj.cov_rolls, sj.cov_H, sj.cov_traj, sj.cov_acc, bx, t0,
                           min(sj.cov_H, t0 + sj.cov_tchunk), b, sj.cov_tiles, sj.cov_nap);
}

__device__ __forceinline__ void side_flag(const SideJob &sj)
{
    if (sj.flag && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(sj.flag, sj.flag_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// This lane's pre-combined policy row V_i = c12 (W_{i-1} - W_i), W = (P +- nu delta) diag(inv_std)
// (ars_agent.py:141-142, environment.py:32-34; u_{-1} = u_{n-1} = 0: free ends), and
// nbias = -V_i . mean.  cols[j]: the observation column of entry j (the quad kernel keeps its
// row in rotated order).  Branch-free on purpose: every lane loads both neighbouring rows with a
// clamped row index and SELECTS afterwards, so all 4 D + 2 D loads are in flight together and the
// launch pays one memory latency instead of ~40 serial ones (measured: the prologue was most of
// the ~6 us fixed cost of a rollout launch).
template <int D, int M, bool ARS>
__device__ __forceinline__ void load_policy_row(const double *__restrict__ pl,
                                                const double *__restrict__ dl, double sgn, double nu,
                                                const double *__restrict__ mean,
                                                const double *__restrict__ inv_std, double c12,
                                                int seg, const int (&cols)[D], double (&V)[D],
                                                double &nbias)
{
    const int a_up = (seg >= 1) ? seg - 1 : 0, a_dn = (seg <= M - 1) ? seg : M - 1;
    double pu[D], pd[D], du[D], dd[D], is[D], mn[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        pu[j] = pl[a_up * D + cols[j]];
        pd[j] = pl[a_dn * D + cols[j]];
    }
    if (ARS) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            du[j] = dl[a_up * D + cols[j]];
            dd[j] = dl[a_dn * D + cols[j]];
        }
    }
    if (inv_std) {   // uniform
#pragma unroll
        for (int j = 0; j < D; ++j) is[j] = inv_std[cols[j]];
    }
    if (mean) {      // uniform
#pragma unroll
        for (int j = 0; j < D; ++j) mn[j] = mean[cols[j]];
    }
    nbias = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        double wu = pu[j], wd = pd[j];
        if (ARS) {   // ars_agent.py:141-142
            wu = __dadd_rn(wu, sgn * __dmul_rn(nu, du[j]));
            wd = __dadd_rn(wd, sgn * __dmul_rn(nu, dd[j]));
        }
        if (inv_std) {   // environment.py:32-33
            wu = __dmul_rn(wu, is[j]);
            wd = __dmul_rn(wd, is[j]);
        }
        const double up = (seg >= 1) ? wu : 0.0;
        const double dn = (seg <= M - 1) ? wd : 0.0;
        V[j] = c12 * (up - dn);
        if (mean) nbias = __builtin_fma(-V[j], mn[j], nbias);
    }
}

// ------------------------------------------------------------------------------------
// n = 3, one segment per lane (swimmer_quad3.h): 16 rollouts per 64-thread workgroup.
// TRAJ / MOM are compile-time so the hot loop carries no per-step uniform branches.
template <bool ARS, bool TRAJ, bool MOM>
__global__ void __launch_bounds__(kRollBlock)
rollout_quad3_kernel(sw::Consts C, int64_t n_roll, int32_t H, const double *__restrict__ policies,
                     const double *__restrict__ deltas, int64_t dir_begin, double nu,
                     const double *__restrict__ mean, const double *__restrict__ inv_std,
                     const double *__restrict__ state0, double *__restrict__ returns,
                     double *__restrict__ traj, double *__restrict__ final_state,
                     double *__restrict__ moments, int32_t *__restrict__ status, SideJob side)
{
    side_flag(side);
    if (blockIdx.x >= side.first_cov_block) {   // a covariance workgroup riding along (uniform)
        side_cov_tile<8, kRollBlock>(side);
        return;
    }
    // This wave's speed IS the iteration time: first in line at the instruction arbiter when a
    // covariance workgroup of the same launch (or, multi-GPU, a collective's wave) lands on its
    // SIMD.  (Reserving the SIMD outright -- allocating all 512 registers by touching v255 / a255
    // -- measured neutral on one GPU and would serialise the covariance workgroups behind the
    // rollouts once a batch fills the chip, so it is not done.)
    __builtin_amdgcn_s_setprio(3);
    constexpr int D = 8, M = 2;
    const int lane = threadIdx.x;
    const int q = lane & 3;
    const int seg = (q == 3) ? 0 : q;              // lane 3 mirrors lane 0
    const int64_t r_raw = (int64_t)blockIdx.x * kMomGroup + (lane >> 2);
    const bool valid = r_raw < n_roll;
    const int64_t r = valid ? r_raw : n_roll - 1;  // surplus quads recompute the last rollout
    const sw::Quad3Lane L = sw::quad3_lane(seg);
    const int cth = 2 + 2 * seg, cthd = 3 + 2 * seg;

    // ---- this lane's policy row: V_i = c12 (W_{i-1} - W_i), W = (P +- nu delta) diag(inv_std)
    // (ars_agent.py:141-142, environment.py:32-34), u_{-1} = u_2 = 0 (free ends); columns in
    // this lane's rotated order [Gdx, Gdy, th_i, thd_i, th_i1, thd_i1, th_i2, thd_i2]
    const int seg1 = (seg + 1) % 3, seg2 = (seg + 2) % 3;
    const int cols[D] = {0, 1, cth, cthd, 2 + 2 * seg1, 3 + 2 * seg1, 2 + 2 * seg2, 3 + 2 * seg2};
    double V[D], nbias;   // nbias = -V . mean: tq = V . (obs - mean) without per-step subtractions
    load_policy_row<D, M, ARS>(ARS ? policies : policies + r * (M * D),
                               ARS ? deltas + (dir_begin + (r >> 1)) * (M * D) : nullptr,
                               (r & 1) ? -1.0 : 1.0, nu, mean, inv_std, C.c12, seg, cols, V, nbias);

    // ---- start state ----
    double gdx = 0.0, gdy = 0.0, th = kHalfPi, thd = 0.0;
    if (state0) {
        gdx = state0[r];
        gdy = state0[n_roll + r];
        th = state0[(int64_t)cth * n_roll + r];
        thd = state0[(int64_t)cthd * n_roll + r];
    }
    // Trajectory stores go through a buffer resource (SGPR base + per-step SGPR offset +
    // per-lane VGPR offset): one store instruction per value and one scalar add per step,
    // no per-store 64-bit address arithmetic.  The host picks this kernel only when the
    // whole trajectory buffer is < 4 GiB (32-bit offsets; out-of-range stores are dropped
    // by the hardware range check, never written elsewhere).
    const uint32_t off_th = (uint32_t)(((int64_t)cth * n_roll + r) * 8);
    const uint32_t off_thd = (uint32_t)(((int64_t)cthd * n_roll + r) * 8);
    // Gdot is replicated on every lane (each lane integrates its own copy, equal up to rounding):
    // lanes of segment 0 record (store and sum) x, the others y.  ONE store of a per-lane selected
    // value: a store costs ~16 issue cycles (measured), the select 2 x 4.4.  The rollout's
    // Gdot_y is lane 1's copy: lane 2's store is dropped by the buffer range check.
    const uint32_t kDrop = 0xfffffff0u;
    const uint32_t off_g = (q == 2) ? kDrop : (uint32_t)(((int64_t)(seg == 0 ? 0 : 1) * n_roll + r) * 8);
    const double selx = (seg == 0) ? 1.0 : 0.0, sely = 1.0 - selx;
    const uint32_t slab = (uint32_t)(D * n_roll * 8);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        traj, 0, TRAJ ? (int)(uint32_t)((int64_t)H * slab) : 0, 0x00020000);
    uint32_t soff = 0;
    auto store_cell = [&](double v, uint32_t voff) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        union { double d; v2i i; } u;
        u.d = v;
        __builtin_amdgcn_raw_buffer_store_b64(u.i, trs, (int)voff, (int)soff, SW_TRAJ_STORE_AUX);
    };

    // The angle is carried in reduced form theta = r + K pi/2 (swimmer_device.h, Angle): no
    // per-step range reduction or quadrant logic in sin / cos.
    sw::Angle A = sw::angle_make(th);
    double thmax = 0.0, det = 1.0;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(thmax) : "v"(thmax), "v"(th));
    double m1th = 0.0, m2th = 0.0, m1thd = 0.0, m2thd = 0.0;
    double m1g = 0.0, m2g = 0.0;   // sums of this lane's Gdot component and its square
    // neighbours' angular velocities for the next step: exchanged at the END of a step (behind
    // the stores and moment updates), so the DPP reads never wait on the Euler update that just
    // wrote them.  The neighbours' ANGLES are never exchanged: the policy is linear in them and
    // theta_j(t+1) = theta_j(t) + h thetadot_j(t), so the angle part of this lane's torque balance,
    //     Th(t) = -V . mean + sum_j V[theta_j] theta_j(t),
    // is carried along as Th(t+1) = Th(t) + sum_j (h V[theta_j]) thetadot_j(t) -- three FMAs on
    // velocities that are exchanged anyway, instead of three FMAs on angles plus four DPP moves.
    // (It also spares the per-step cancellation of V . theta against V . mean, ~1e4 against ~1
    // once the whitening is on.)
    double w1 = sw::dpp_f64<sw::kDppNext1>(thd), w2 = sw::dpp_f64<sw::kDppNext2>(thd);
    double Th = __builtin_fma(V[2], th, nbias);
    Th = __builtin_fma(V[4], sw::dpp_f64<sw::kDppNext1>(th), Th);
    Th = __builtin_fma(V[6], sw::dpp_f64<sw::kDppNext2>(th), Th);
    const double hV2 = C.h * V[2], hV4 = C.h * V[4], hV6 = C.h * V[6];
    const sw::TrigK K = sw::trig_consts();
    double magic = 6755399441055744.0;   // 1.5 * 2^52, pinned in a VGPR pair for angle_keep_reduced
    asm volatile("" : "+v"(magic));
    sw::Quad3Geo G = sw::quad3_geometry(A, K), Gn;
    // one step: consumes the geometry Gc of theta_t, produces Gx for theta_{t+1}
    auto one_step = [&](const sw::Quad3Geo &Gc, sw::Quad3Geo &Gx) {
        // this segment's torque balance c12 (u_{i-1} - u_i) = V_i . (obs - mean): the carried
        // angle part + the velocity part (the neighbours' angular velocities arrive by DPP and are
        // reused by the physics step).  One accumulator: the kernel is issue-bound, not chain-bound.
        // theta_{t+1} needs thetadot_t only: advance the angle first and start its range test; the
        // policy's eight FMAs sit between the vector compare and the scalar branch that waits for it.
        // Its sin / cos and the neighbour exchange run beside this step's solve (software pipelining
        // across steps, swimmer_quad3.h)
        A.r = __builtin_fma(C.h, thd, A.r);
        const unsigned long long outside = sw::angle_range_test(A.r);
        double tq = __builtin_fma(V[0], gdx, Th);
        tq = __builtin_fma(V[1], gdy, tq);
        tq = __builtin_fma(V[3], thd, tq);
        tq = __builtin_fma(V[5], w1, tq);
        tq = __builtin_fma(V[7], w2, tq);
        Th = __builtin_fma(hV2, thd, Th);
        Th = __builtin_fma(hV4, w1, Th);
        Th = __builtin_fma(hV6, w2, Th);
        sw::angle_keep_reduced(A, thmax, magic, outside);   // untaken branch; rare re-normalisation
        const double th_next = sw::angle_theta(A);
        Gx = sw::quad3_geometry(A, K);
        det = sw::quad3_dynamics(C, L, Gc, gdx, gdy, thd, w1, w2, tq);
        th = th_next;
        // the return comes out of the per-component sums in the epilogue (linearity), no
        // per-step reward arithmetic
        const double gsel = __builtin_fma(selx, gdx, sely * gdy);
        m1g += gsel;
        if (TRAJ) {
            store_cell(th, off_th);
            store_cell(thd, off_thd);
            store_cell(gsel, off_g);
            soff += slab;
        }
        if (MOM) {
            const double a = th - kHalfPi;
            m1th += a;
            m2th = __builtin_fma(a, a, m2th);
            m1thd += thd;
            m2thd = __builtin_fma(thd, thd, m2thd);
            m2g = __builtin_fma(gsel, gsel, m2g);
        }
        w1 = sw::dpp_f64<sw::kDppNext1>(thd);
        w2 = sw::dpp_f64<sw::kDppNext2>(thd);
    };
    // four steps per trip, the geometry ping-pongs between G and Gn (no register copies)
    int32_t t = 0;
#if SW_QUAD_UNROLL == 4
    SW_PIN_LOOP(SW_QUAD_LOOP_PAD);
    for (; t + 4 <= H; t += 4) {
        one_step(G, Gn);
        one_step(Gn, G);
        one_step(G, Gn);
        one_step(Gn, G);
    }
#endif
    for (; t + 2 <= H; t += 2) {
        one_step(G, Gn);
        one_step(Gn, G);
    }
    if (t < H) one_step(G, Gn);
    asm("v_max_f64 %0, %1, |%2|" : "=v"(thmax) : "v"(thmax), "v"(th));
    // the joint-acceleration system is the chain's (scaled) mass matrix: positive definite for
    // every finite configuration, so its determinant can only fail to be positive once the state
    // is no longer finite -- the last step's says so
    const double detmin = det;

    // ---- per-rollout outputs (quad lanes 0..2 hold the state; lane 0 the return) ----
    int code = ((detmin > 0.0) ? 0 : SW_STATUS_SINGULAR) |
               ((isfinite(th) && isfinite(thd) && isfinite(gdx) && isfinite(gdy)) ? 0 : SW_STATUS_NONFINITE) |
               ((thmax < sw::kAngleLimit) ? 0 : SW_STATUS_RANGE);
    code |= __builtin_amdgcn_mov_dpp(code, sw::kDppNext1, 0xf, 0xf, true) |
            __builtin_amdgcn_mov_dpp(code, sw::kDppNext2, 0xf, 0xf, true);
    // sum of the rewards Gdot_t . direction (remy_swimmer_env.py:238-243), by linearity:
    // lane 0 holds sum Gdot_x, lane 1 sum Gdot_y
    const double sgy = sw::dpp_f64<sw::kDppNext1>(m1g);
    if (valid && q == 0) {
        const double total = __builtin_fma(C.dirx, m1g, C.diry * sgy);
        returns[r] = (code & SW_STATUS_RANGE) ? __builtin_nan("") : total;
        if (status) status[r] = code;
    }
    if (final_state && valid && q < 3) {
        final_state[(int64_t)cth * n_roll + r] = th;
        final_state[(int64_t)cthd * n_roll + r] = thd;
        if (q < 2) final_state[(int64_t)q * n_roll + r] = (q == 0) ? gdx : gdy;
    }
    if (MOM) {
        if (!valid) m1th = m2th = m1thd = m2thd = m1g = m2g = 0.0;
        // sum over the 16 rollouts of the wave, per segment lane: xor-butterfly over lane>>2
#pragma unroll
        for (int off = 4; off < kWave; off <<= 1) {
            m1th += __shfl_xor(m1th, off, kWave);
            m2th += __shfl_xor(m2th, off, kWave);
            m1thd += __shfl_xor(m1thd, off, kWave);
            m2thd += __shfl_xor(m2thd, off, kWave);
            m1g += __shfl_xor(m1g, off, kWave);
            m2g += __shfl_xor(m2g, off, kWave);
        }
        if (lane < 3) {
            double *row = moments + (int64_t)blockIdx.x * (2 * D);
            row[cth] = m1th;
            row[cthd] = m1thd;
            row[D + cth] = m2th;
            row[D + cthd] = m2thd;
            if (lane < 2) {
                row[lane] = m1g;
                row[D + lane] = m2g;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// n = 3 with lane roles (swimmer_oct3.h): two mirror quads per rollout, 8 rollouts per wave, two
// waves = 16 rollouts = one V2 moment row per 128-thread workgroup.
template <bool ARS, bool TRAJ, bool MOM>
__global__ void __launch_bounds__(kOctBlock)
rollout_oct3_kernel(sw::Consts C, int64_t n_roll, int32_t H, const double *__restrict__ policies,
                    const double *__restrict__ deltas, int64_t dir_begin, double nu,
                    const double *__restrict__ mean, const double *__restrict__ inv_std,
                    const double *__restrict__ state0, double *__restrict__ returns,
                    double *__restrict__ traj, double *__restrict__ final_state,
                    double *__restrict__ moments, int32_t *__restrict__ status, SideJob side)
{
    side_flag(side);
    if (blockIdx.x >= side.first_cov_block) {   // a covariance workgroup riding along (uniform)
        side_cov_tile<8, kOctBlock>(side);
        return;
    }
    __builtin_amdgcn_s_setprio(3);   // as in the quad kernel
    constexpr int D = 8, M = 2;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int q = lane & 3;
    const int seg = (q == 3) ? 0 : q;              // lane 3 of a quad mirrors lane 0
    const bool cosine = (lane & 8) != 0;           // quad B of the rollout: cosine / Gdot_y roles
    const int64_t r_raw = (int64_t)blockIdx.x * kMomGroup + wave * 8 + (lane >> 4) * 2 + ((lane >> 2) & 1);
    const bool valid = r_raw < n_roll;
    const int64_t r = valid ? r_raw : n_roll - 1;  // surplus rollouts recompute the last one
    const sw::OctLane O = sw::oct3_lane(C, seg, cosine);
    const int cth = 2 + 2 * seg, cthd = 3 + 2 * seg;

    // this lane's policy row in its rotated order [Gdx, Gdy, th_i, thd_i, th_i1, thd_i1, th_i2, thd_i2]
    const int seg1 = (seg + 1) % 3, seg2 = (seg + 2) % 3;
    const int cols[D] = {0, 1, cth, cthd, 2 + 2 * seg1, 3 + 2 * seg1, 2 + 2 * seg2, 3 + 2 * seg2};
    double V[D], nbias;
    load_policy_row<D, M, ARS>(ARS ? policies : policies + r * (M * D),
                               ARS ? deltas + (dir_begin + (r >> 1)) * (M * D) : nullptr,
                               (r & 1) ? -1.0 : 1.0, nu, mean, inv_std, C.c12, seg, cols, V, nbias);
    // Gdot in the roles: Pu = the component this quad integrates, Pv = its partner's
    const double VPu = cosine ? V[1] : V[0], VPv = cosine ? V[0] : V[1];

    double gdx = 0.0, gdy = 0.0, th = kHalfPi, thd = 0.0;
    if (state0) {
        gdx = state0[r];
        gdy = state0[n_roll + r];
        th = state0[(int64_t)cth * n_roll + r];
        thd = state0[(int64_t)cthd * n_roll + r];
    }
    double Pu = cosine ? gdy : gdx, Pv = cosine ? gdx : gdy;

    // trajectory cells through a buffer resource (as in the quad kernel); quad A records theta,
    // thetadot and Gdot_x, quad B Gdot_y; every other lane's store is dropped by the range check
    const uint32_t kDrop = 0xfffffff0u;
    const bool rec = !cosine && q < 3;
    const uint32_t off_th = rec ? (uint32_t)(((int64_t)cth * n_roll + r) * 8) : kDrop;
    const uint32_t off_thd = rec ? (uint32_t)(((int64_t)cthd * n_roll + r) * 8) : kDrop;
    const uint32_t off_g = (q == 0) ? (uint32_t)(((int64_t)(cosine ? 1 : 0) * n_roll + r) * 8) : kDrop;
    const uint32_t slab = (uint32_t)(D * n_roll * 8);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        traj, 0, TRAJ ? (int)(uint32_t)((int64_t)H * slab) : 0, 0x00020000);
    uint32_t soff = 0;
    auto store_cell = [&](double v, uint32_t voff) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        union { double d; v2i i; } u;
        u.d = v;
        __builtin_amdgcn_raw_buffer_store_b64(u.i, trs, (int)voff, (int)soff, SW_TRAJ_STORE_AUX);
    };

    // the angle in reduced form + the polynomial this lane currently evaluates (swimmer_oct3.h)
    const int designation = cosine ? 1 : 0;
    double thmax = 0.0, det = 1.0;
    sw::OctTrig A;
    A.r = th;
    A.kd = 0.0;
    sw::oct3_renorm(A, designation, thmax);
    double m1th = 0.0, m2th = 0.0, m1thd = 0.0, m2thd = 0.0, m1g = 0.0, m2g = 0.0;
    double w1 = sw::dpp_f64<sw::kDppNext1>(thd), w2 = sw::dpp_f64<sw::kDppNext2>(thd);
    double Th = __builtin_fma(V[2], th, nbias);
    Th = __builtin_fma(V[4], sw::dpp_f64<sw::kDppNext1>(th), Th);
    Th = __builtin_fma(V[6], sw::dpp_f64<sw::kDppNext2>(th), Th);
    const double hV2 = C.h * V[2], hV4 = C.h * V[4], hV6 = C.h * V[6];
    sw::OctGeo G = sw::oct3_geometry(A), Gn;
    double magic = 6755399441055744.0;   // 1.5 * 2^52, pinned in a VGPR pair for oct3_keep_reduced
    asm volatile("" : "+v"(magic));
    auto one_step = [&](const sw::OctGeo &Gc, sw::OctGeo &Gx) {
        // theta_{t+1} needs thetadot_t only: advance the angle first and start its range test, the
        // policy's eight FMAs sit between the vector compare and the scalar branch that waits for it
        A.r = __builtin_fma(C.h, thd, A.r);
        const unsigned long long outside = sw::oct3_range_test(A.r);
        double tq = __builtin_fma(VPu, Pu, Th);
        tq = __builtin_fma(VPv, Pv, tq);
        tq = __builtin_fma(V[3], thd, tq);
        tq = __builtin_fma(V[5], w1, tq);
        tq = __builtin_fma(V[7], w2, tq);
        Th = __builtin_fma(hV2, thd, Th);
        Th = __builtin_fma(hV4, w1, Th);
        Th = __builtin_fma(hV6, w2, Th);
        sw::oct3_keep_reduced(A, thmax, magic, designation, outside);   // untaken branch; rare re-normalisation
        const double th_next = __builtin_fma(A.kd, sw::kPio2Hi, A.r);
        Gx = sw::oct3_geometry(A);
        det = sw::oct3_dynamics(C, O, Gc, Pu, Pv, thd, w1, w2, tq);
        th = th_next;
        m1g += Pu;
        if (TRAJ) {
            store_cell(th, off_th);
            store_cell(thd, off_thd);
            store_cell(Pu, off_g);
            soff += slab;
        }
        if (MOM) {
            const double a = th - kHalfPi;
            m1th += a;
            m2th = __builtin_fma(a, a, m2th);
            m1thd += thd;
            m2thd = __builtin_fma(thd, thd, m2thd);
            m2g = __builtin_fma(Pu, Pu, m2g);
        }
        w1 = sw::dpp_f64<sw::kDppNext1>(thd);
        w2 = sw::dpp_f64<sw::kDppNext2>(thd);
        Pv = sw::dpp_row_f64<sw::kDppRowRor8>(Pu);
    };
    // the geometry ping-pongs between G and Gn (no register copies): an even number of steps per trip
    int32_t t = 0;
    SW_PIN_LOOP(oct_loop_pad(TRAJ, MOM));
#if SW_OCT_UNROLL == 8
    // eight steps per trip: the loop's back edge costs a lone wave ~8-13 ns (2 / 4 / 8 steps per trip:
    // 0.2292 / 0.2272 / 0.2250 ms per launch, each at its best loop offset; profiles/r03_t, r03_w)
    for (; t + 8 <= H; t += 8) {
        one_step(G, Gn);
        one_step(Gn, G);
        one_step(G, Gn);
        one_step(Gn, G);
        one_step(G, Gn);
        one_step(Gn, G);
        one_step(G, Gn);
        one_step(Gn, G);
    }
#endif
    for (; t + 4 <= H; t += 4) {
        one_step(G, Gn);
        one_step(Gn, G);
        one_step(G, Gn);
        one_step(Gn, G);
    }
    for (; t + 2 <= H; t += 2) {
        one_step(G, Gn);
        one_step(Gn, G);
    }
    if (t < H) one_step(G, Gn);
    thmax = fmax(thmax, fabs(th));

    // ---- per-rollout outputs: quad A lanes 0..2 hold (theta, thetadot), A lane 0 Gdot_x, B lane 0 Gdot_y
    int code = ((det > 0.0) ? 0 : SW_STATUS_SINGULAR) |
               ((isfinite(th) && isfinite(thd) && isfinite(Pu) && isfinite(Pv)) ? 0 : SW_STATUS_NONFINITE) |
               ((thmax < sw::kAngleLimit) ? 0 : SW_STATUS_RANGE);
    code |= __builtin_amdgcn_mov_dpp(code, sw::kDppNext1, 0xf, 0xf, true) |
            __builtin_amdgcn_mov_dpp(code, sw::kDppNext2, 0xf, 0xf, true);
    code |= __builtin_amdgcn_mov_dpp(code, sw::kDppRowRor8, 0xf, 0xf, true);
    const double sg_other = sw::dpp_row_f64<sw::kDppRowRor8>(m1g);   // on A: sum Gdot_y
    if (valid && !cosine && q == 0) {
        const double total = __builtin_fma(C.dirx, m1g, C.diry * sg_other);
        returns[r] = (code & SW_STATUS_RANGE) ? __builtin_nan("") : total;
        if (status) status[r] = code;
    }
    if (final_state && valid) {
        if (rec) {
            final_state[(int64_t)cth * n_roll + r] = th;
            final_state[(int64_t)cthd * n_roll + r] = thd;
        }
        if (q == 0) final_state[(int64_t)(cosine ? 1 : 0) * n_roll + r] = Pu;
    }
    if (MOM) {
        __shared__ double shm[kOctBlock / kWave][16][6];
        if (!valid) m1th = m2th = m1thd = m2thd = m1g = m2g = 0.0;
        // sum over the 8 rollouts of the wave, per (quad half, segment) lane: lane bits 2, 4, 5
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int off = (k == 0) ? 4 : (k == 1 ? 16 : 32);
            m1th += __shfl_xor(m1th, off, kWave);
            m2th += __shfl_xor(m2th, off, kWave);
            m1thd += __shfl_xor(m1thd, off, kWave);
            m2thd += __shfl_xor(m2thd, off, kWave);
            m1g += __shfl_xor(m1g, off, kWave);
            m2g += __shfl_xor(m2g, off, kWave);
        }
        if (lane < 16) {
            shm[wave][lane][0] = m1th;
            shm[wave][lane][1] = m2th;
            shm[wave][lane][2] = m1thd;
            shm[wave][lane][3] = m2thd;
            shm[wave][lane][4] = m1g;
            shm[wave][lane][5] = m2g;
        }
        __syncthreads();
        // row lanes 0..2: segments (quad A); row lane 0: Gdot_x sums; row lane 8: Gdot_y sums (quad B)
        if (tid < 3) {
            double *row = moments + (int64_t)blockIdx.x * (2 * D);
            row[2 + 2 * tid] = shm[0][tid][0] + shm[1][tid][0];
            row[D + 2 + 2 * tid] = shm[0][tid][1] + shm[1][tid][1];
            row[3 + 2 * tid] = shm[0][tid][2] + shm[1][tid][2];
            row[D + 3 + 2 * tid] = shm[0][tid][3] + shm[1][tid][3];
            if (tid < 2) {
                const int src = tid * 8;
                row[tid] = shm[0][src][4] + shm[1][src][4];
                row[D + tid] = shm[0][src][5] + shm[1][src][5];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// The safe-exploration gate (safe_rollout_kernel above has the semantics) for n = 3 in the MIRROR-QUAD form of
// rollout_oct3_kernel: two quads of eight lanes per rollout, lane roles, reduced angles.  The geometry of a step
// (sin / cos, cos(th_i - th_k), ...) depends on the angles only and is therefore SHARED by the simulator's look-ahead
// and the real step: per step one geometry, two `oct3_dynamics` (simulator constants on copies of Gdot / thetadot,
// real constants), the cost of the simulated next state on the lanes that own the observed quantity, one AND over
// the rollout's eight lanes (three DPP-ANDs: the two mirror quads differ by rounding, the decision must not), and
// the real step committed through selects.  170 instructions per env-step (186 with the violation count) instead of
// ~640 in the lane form: 0.354 ms per 1024 gated rollouts x 1000 steps against 1.13 ms (profiles/r04_z).
// A refused rollout keeps recomputing the same refused step (its state no longer changes), as in the reference.
__device__ __forceinline__ double oct_sel(bool take, double a, double b) { return take ? a : b; }

template <bool TRAJ, bool VIOL>
__global__ void __launch_bounds__(kOctBlock)
safe_rollout_oct3_kernel(sw::Consts Cr, sw::Consts Cs, double tq_ratio, int64_t n_roll, int32_t H,
                         const double *__restrict__ policies, int32_t cost_kind, int32_t cost_index,
                         double sim_thresh, double real_thresh, double *__restrict__ returns,
                         double *__restrict__ traj, int32_t *__restrict__ first_refused,
                         int32_t *__restrict__ violations, int32_t *__restrict__ status)
{
    __builtin_amdgcn_s_setprio(3);
    constexpr int D = 8, M = 2;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int q = lane & 3;
    const int seg = (q == 3) ? 0 : q;              // lane 3 of a quad mirrors lane 0
    const bool cosine = (lane & 8) != 0;           // quad B: cosine / Gdot_y roles
    const int64_t r_raw = (int64_t)blockIdx.x * kMomGroup + wave * 8 + (lane >> 4) * 2 + ((lane >> 2) & 1);
    const bool valid = r_raw < n_roll;
    const int64_t r = valid ? r_raw : n_roll - 1;  // surplus rollouts recompute the last one
    const sw::OctLane Or = sw::oct3_lane(Cr, seg, cosine), Os = sw::oct3_lane(Cs, seg, cosine);
    const int cth = 2 + 2 * seg, cthd = 3 + 2 * seg;
    const int seg1 = (seg + 1) % 3, seg2 = (seg + 2) % 3;
    const int cols[D] = {0, 1, cth, cthd, 2 + 2 * seg1, 3 + 2 * seg1, 2 + 2 * seg2, 3 + 2 * seg2};
    double V[D], nbias;
    load_policy_row<D, M, false>(policies + r * (M * D), nullptr, 1.0, 0.0, nullptr, nullptr, Cr.c12, seg, cols, V,
                                 nbias);
    const double VPu = cosine ? V[1] : V[0], VPv = cosine ? V[0] : V[1];

    // which quantity of the (simulated, resp. real) next state this lane contributes to the cost: obs =
    // [Gdx (quad A's Pu), Gdy (quad B's Pu), th_1, thd_1, ...] -- quad A's segment lanes own (theta_i, thetadot_i)
    const bool segA = !cosine && q < 3;
    bool own_pu = false, own_th = false, own_thd = false;
    if (cost_kind == SW_COST_MAX_ABS_THETADOT) {
        own_thd = segA;
    } else if (cost_index == 0) {
        own_pu = !cosine && q == 0;
    } else if (cost_index == 1) {
        own_pu = cosine && q == 0;
    } else {
        const bool mine = segA && seg == ((cost_index - 2) >> 1);
        own_thd = mine && ((cost_index - 2) & 1);
        own_th = mine && !((cost_index - 2) & 1);
    }
    const bool owner = own_pu || own_th || own_thd;

    double th = kHalfPi, thd = 0.0, Pu = 0.0, Pv = 0.0;       // real_env.reset() (:133)
    const uint32_t kDrop = 0xfffffff0u;
    const bool rec = !cosine && q < 3;
    const uint32_t off_th = rec ? (uint32_t)(((int64_t)cth * n_roll + r) * 8) : kDrop;
    const uint32_t off_thd = rec ? (uint32_t)(((int64_t)cthd * n_roll + r) * 8) : kDrop;
    const uint32_t off_g = (q == 0) ? (uint32_t)(((int64_t)(cosine ? 1 : 0) * n_roll + r) * 8) : kDrop;
    const uint32_t slab = (uint32_t)(D * n_roll * 8);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        traj, 0, TRAJ ? (int)(uint32_t)((int64_t)H * slab) : 0, 0x00020000);
    uint32_t soff = 0;
    auto store_cell = [&](double v, uint32_t voff) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        union { double d; v2i i; } u;
        u.d = v;
        __builtin_amdgcn_raw_buffer_store_b64(u.i, trs, (int)voff, (int)soff, SW_TRAJ_STORE_AUX);
    };
    const int designation = cosine ? 1 : 0;
    double thmax = 0.0, det = 1.0;
    sw::OctTrig A;
    A.r = th;
    A.kd = 0.0;
    sw::oct3_renorm(A, designation, thmax);
    double ret = 0.0;
    int32_t taken = 0, over = 0;
    bool alive = true;                                    // once refused, always refused
    double w1 = 0.0, w2 = 0.0;
    double Th = __builtin_fma(V[2], th, nbias);
    Th = __builtin_fma(V[4], sw::dpp_f64<sw::kDppNext1>(th), Th);
    Th = __builtin_fma(V[6], sw::dpp_f64<sw::kDppNext2>(th), Th);
    const double hV2 = Cr.h * V[2], hV4 = Cr.h * V[4], hV6 = Cr.h * V[6];
    sw::OctGeo G = sw::oct3_geometry(A), Gn;
    double magic = 6755399441055744.0;
    asm volatile("" : "+v"(magic));
    // AND of a per-lane flag over the eight lanes of this lane's rollout (two rotations inside the quad, then the
    // mirror quad eight lanes away)
    auto all_of_rollout = [&](bool f) -> bool {
        int v = f ? 1 : 0;
        v &= __builtin_amdgcn_mov_dpp(v, 1 | (2 << 2) | (3 << 4) | (0 << 6), 0xf, 0xf, true);   // [1,2,3,0]
        v &= __builtin_amdgcn_mov_dpp(v, 2 | (3 << 2) | (0 << 4) | (1 << 6), 0xf, 0xf, true);   // [2,3,0,1]
        v &= __builtin_amdgcn_mov_dpp(v, sw::kDppRowRor8, 0xf, 0xf, true);
        return v != 0;
    };
    auto one_step = [&](const sw::OctGeo &Gc, sw::OctGeo &Gx) {
        double tq = __builtin_fma(VPu, Pu, Th);
        tq = __builtin_fma(VPv, Pv, tq);
        tq = __builtin_fma(V[3], thd, tq);
        tq = __builtin_fma(V[5], w1, tq);
        tq = __builtin_fma(V[7], w2, tq);
        // the simulator's look-ahead from the real state (:120-121): same geometry, its own constants, copies
        double Pus = Pu, thds = thd;
        (void)sw::oct3_dynamics(Cs, Os, Gc, Pus, Pv, thds, w1, w2, tq * tq_ratio);
        const double ths = __builtin_fma(Cs.h, thd, th);
        const double vs = own_thd ? thds : (own_th ? ths : Pus);
        const bool safe = all_of_rollout(!owner || (fabs(vs) <= sim_thresh)) && alive;     // :122, NaN refuses
        alive = safe;
        // the real step on copies, committed where the gate is open (:142)
        double Pur = Pu, thdr = thd;
        const double det_new = sw::oct3_dynamics(Cr, Or, Gc, Pur, Pv, thdr, w1, w2, tq);
        const double r_new = __builtin_fma(Cr.h, thd, A.r);
        double Th_new = __builtin_fma(hV2, thd, Th);
        Th_new = __builtin_fma(hV4, w1, Th_new);
        Th_new = __builtin_fma(hV6, w2, Th_new);
        A.r = oct_sel(safe, r_new, A.r);
        const unsigned long long outside = sw::oct3_range_test(A.r);
        Th = oct_sel(safe, Th_new, Th);
        thd = oct_sel(safe, thdr, thd);
        Pu = oct_sel(safe, Pur, Pu);
        det = oct_sel(safe, det_new, det);
        sw::oct3_keep_reduced(A, thmax, magic, designation, outside);
        th = __builtin_fma(A.kd, sw::kPio2Hi, A.r);
        Gx = sw::oct3_geometry(A);
        ret += oct_sel(safe, Pu, 0.0);
        taken += safe ? 1 : 0;
        if (VIOL) {
            const double vr = own_thd ? thd : (own_th ? th : Pu);
            // cost > real_thresh (:143): any owner lane over the threshold (max |thetadot_i|), on a step that was taken
            const bool fine = all_of_rollout(!owner || !(fabs(vr) > real_thresh));
            over += (safe && !fine) ? 1 : 0;
        }
        if (TRAJ) {
            store_cell(th, off_th);
            store_cell(thd, off_thd);
            store_cell(Pu, off_g);
            soff += slab;
        }
        w1 = sw::dpp_f64<sw::kDppNext1>(thd);
        w2 = sw::dpp_f64<sw::kDppNext2>(thd);
        Pv = sw::dpp_row_f64<sw::kDppRowRor8>(Pu);
    };
    int32_t t = 0;
    for (; t + 2 <= H; t += 2) {
        one_step(G, Gn);
        one_step(Gn, G);
    }
    if (t < H) one_step(G, Gn);
    thmax = fmax(thmax, fabs(th));

    int code = ((det > 0.0) ? 0 : SW_STATUS_SINGULAR) |
               ((isfinite(th) && isfinite(thd) && isfinite(Pu) && isfinite(Pv)) ? 0 : SW_STATUS_NONFINITE) |
               ((thmax < sw::kAngleLimit) ? 0 : SW_STATUS_RANGE);
    code |= __builtin_amdgcn_mov_dpp(code, sw::kDppNext1, 0xf, 0xf, true) |
            __builtin_amdgcn_mov_dpp(code, sw::kDppNext2, 0xf, 0xf, true);
    code |= __builtin_amdgcn_mov_dpp(code, sw::kDppRowRor8, 0xf, 0xf, true);
    const double ret_other = sw::dpp_row_f64<sw::kDppRowRor8>(ret);   // on A: sum of the taken steps' Gdot_y
    if (valid && !cosine && q == 0) {
        const double total = __builtin_fma(Cr.dirx, ret, Cr.diry * ret_other);
        returns[r] = (code & SW_STATUS_RANGE) ? __builtin_nan("") : total;
        if (first_refused) first_refused[r] = taken;      // the gate never re-opens: steps taken = first refused step
        if (violations) violations[r] = over;
        if (status) status[r] = code;
    }
}

// ------------------------------------------------------------------------------------
// n = 4..8, one segment per lane, one rollout per 16-lane DPP row (swimmer_row.h).
// 256-thread workgroups: 4 waves x 4 rows = 16 rollouts = one V2 moment row.
// Two waves per SIMD must fit for n <= 6 (256 registers each): the covariance workgroups that ride along
// are waves of THIS kernel, and once a batch puts a rollout wave on every SIMD (2048 directions on one
// GPU) a wave that needs more than half the register file cannot join it -- the pass would run after the
// rollouts (measured at 264 registers: launch 0.69 -> 0.89 ms, profiles/r03_h_ab_row_registers.log).
template <int N, bool ARS, bool TRAJ, bool MOM>
__global__ void __launch_bounds__(kRowBlock, (N <= 6 ? 2 : 1))
rollout_row_kernel(sw::Consts C, int64_t n_roll, int32_t H, const double *__restrict__ policies,
                   const double *__restrict__ deltas, int64_t dir_begin, double nu,
                   const double *__restrict__ mean, const double *__restrict__ inv_std,
                   const double *__restrict__ state0, double *__restrict__ returns,
                   double *__restrict__ traj, double *__restrict__ final_state,
                   double *__restrict__ moments, int32_t *__restrict__ status, SideJob side)
{
    side_flag(side);
    if (blockIdx.x >= side.first_cov_block) {   // a covariance workgroup riding along (uniform)
        side_cov_tile<2 * N + 2, kRowBlock>(side);
        return;
    }
    __builtin_amdgcn_s_setprio(3);   // as in the quad kernel
    constexpr int D = 2 * N + 2, M = N - 1;
    const int tid = threadIdx.x;
    const int q = tid & 15;                        // lane inside the row
    const bool owner = q < N;                      // lanes 0..N-1 own the segments' cells
    const bool cosine = q >= 8;                    // lane i + 8 mirrors lane i and evaluates the cosine
    const int seg = ((q & 7) < N) ? (q & 7) : 0;   // lanes N..7 / N+8..15 mirror lane 0 / 8
    const int64_t r_raw = (int64_t)blockIdx.x * kMomGroup + (tid >> 4);
    const bool valid = r_raw < n_roll;
    const int64_t r = valid ? r_raw : n_roll - 1;  // surplus rows recompute the last rollout
    const sw::RowLane<N> L = sw::row_lane<N>(C, seg);
    const int cth = 2 + 2 * seg, cthd = 3 + 2 * seg;

    // ---- this lane's policy row: V_i = c12 (W_{i-1} - W_i), W = (P +- nu delta) diag(inv_std)
    // (ars_agent.py:141-142, environment.py:32-34); u_{-1} = u_{n-1} = 0 (free ends)
    double V[D], nbias;   // nbias = -V . mean: tq = V . (obs - mean) without per-step subtractions
    {
        int cols[D];
#pragma unroll
        for (int j = 0; j < D; ++j) cols[j] = j;   // canonical order
        load_policy_row<D, M, ARS>(ARS ? policies : policies + r * (M * D),
                                   ARS ? deltas + (dir_begin + (r >> 1)) * (M * D) : nullptr,
                                   (r & 1) ? -1.0 : 1.0, nu, mean, inv_std, C.c12, seg, cols, V, nbias);
    }

    // ---- start state ----
    double gdx = 0.0, gdy = 0.0, th = kHalfPi, thd = 0.0;
    if (state0) {
        gdx = state0[r];
        gdy = state0[n_roll + r];
        th = state0[(int64_t)cth * n_roll + r];
        thd = state0[(int64_t)cthd * n_roll + r];
    }
    // trajectory cells through a buffer resource; lanes that own no cell get an offset
    // beyond the buffer, which the hardware range check drops
    const uint32_t slab = (uint32_t)(D * n_roll * 8);
    const uint32_t kDrop = 0xfffffff0u;
    const uint32_t off_th = (owner && valid) ? (uint32_t)(((int64_t)cth * n_roll + r) * 8) : kDrop;
    const uint32_t off_thd = (owner && valid) ? (uint32_t)(((int64_t)cthd * n_roll + r) * 8) : kDrop;
    // Gdot is replicated (bit-identical on all lanes): lane 0 stores x, lane 1 stores y, with two
    // store instructions.  (The quad kernel's per-lane select + single store has the same
    // instruction count here but measured 5 % slower: the select lands on the serial chain.)
    const uint32_t off_gx = (q == 0 && valid) ? (uint32_t)(r * 8) : kDrop;
    const uint32_t off_gy = (q == 1 && valid) ? (uint32_t)((n_roll + r) * 8) : kDrop;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        traj, 0, TRAJ ? (int)(uint32_t)((int64_t)H * slab) : 0, 0x00020000);
    uint32_t soff = 0;
    auto store_cell = [&](double v, uint32_t voff) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        union { double d; v2i i; } u;
        u.d = v;
        __builtin_amdgcn_raw_buffer_store_b64(u.i, trs, (int)voff, (int)soff, SW_TRAJ_STORE_AUX);
    };

    double thmax = 0.0, rq_last = 1.0;
    double m1th = 0.0, m2th = 0.0, m1thd = 0.0, m2thd = 0.0;
    double sgx = 0.0, sgy = 0.0, qgx = 0.0, qgy = 0.0;   // sums of Gdot and Gdot^2 over the steps
    // theta = r + K pi/2 and the polynomial this lane evaluates of r (swimmer_oct3.h, OctTrig)
    const int designation = cosine ? 1 : 0;
    sw::OctTrig A;
    A.r = th;
    A.kd = 0.0;
    sw::oct3_renorm(A, designation, thmax);
    // one step: policy + physics (swimmer_row.h; the other segments' angles and angular velocities are
    // read straight out of their lanes by fused broadcast-FMAs), then the step's records
    auto one_step = [&](auto slow) {
        const double rq = sw::row_step<N, decltype(slow)::value>(C, L, V, nbias, cosine, designation, gdx, gdy,
                                                                 A, th, thd, thmax);
        rq_last = rq;   // the system is the chain's mass matrix: a pivot can only fail to be positive once the
                        // state is no longer finite, and then the last step's says so
        // the return comes out of the per-component sums in the epilogue (linearity)
        sgx += gdx;
        sgy += gdy;
        if (TRAJ) {
            store_cell(th, off_th);
            store_cell(thd, off_thd);
            store_cell(gdx, off_gx);
            store_cell(gdy, off_gy);
            soff += slab;
        }
        if (MOM) {
            const double a = th - kHalfPi;
            m1th += a;
            m2th = __builtin_fma(a, a, m2th);
            m1thd += thd;
            m2thd = __builtin_fma(thd, thd, m2thd);
            qgx = __builtin_fma(gdx, gdx, qgx);
            qgy = __builtin_fma(gdy, gdy, qgy);
        }
    };
    // Range check once per trip of four steps (the mirror-quad kernel's per-step asm check would cost
    // registers this kernel does not have at n >= 6): a trip whose angles move at most kTripSlack runs
    // unchecked after one re-normalisation at its start if needed -- r ends at most that far past pi/4,
    // where the polynomials are still accurate to 2.5e-16 (swimmer_oct3.h); a faster trip runs in the
    // second loop, which checks inside every step (exact for any angular velocity).  What thetadot GAINS inside
    // an unchecked trip is not in that bound: an angle travels up to 6 h^2 |thetadotdot| further (0.06 rad at
    // 10 000 rad/s^2) before the next trip start sees the speed.  The polynomials degrade smoothly out there --
    // 2.0e-16 at pi/4 + 0.04, 1.7e-15 at + 0.10, 9e-14 at + 0.25 (tests/test_trig_range.py) -- and rollouts with
    // first-step accelerations of 3 000 ... 20 000 rad/s^2 stay within 1e-6 (relative) of the oracle
    // (tests/test_hip_parity.py::test_violent_accelerations_inside_an_unchecked_trip).  One step per loop
    // body either way: unrolled, n >= 6 would leave the 256 architectural registers.
    auto too_fast = [&]() -> bool {
        return __any((4.0 * C.h) * fabs(thd) > sw::kTripSlack);
    };
    int32_t t = 0;
    SW_PIN_LOOP(row_loop_pad(N, TRAJ, MOM));
    while (t < H) {   // two loops, not one loop with two bodies: merged, the compiler reconciles the bodies'
                      // register assignments with copies on the common path (profiles/r03_g_ab_range_check_variants.log)
        while (t < H) {                              // unchecked trips of (up to) four steps
            if (__builtin_expect(too_fast(), 0)) break;
            const double reach = __builtin_fma(4.0 * C.h, fabs(thd), fabs(A.r));
            if (__builtin_expect(__any(reach > sw::kPio4), 0)) sw::oct3_renorm(A, designation, thmax);
            const int32_t t_end = min(H, t + 4);
#pragma unroll 1
            for (; t < t_end; ++t) one_step(std::false_type{});
        }
#pragma unroll 1
        for (; t < H && too_fast(); ++t) one_step(std::true_type{});   // checks inside every step
    }

    thmax = fmax(thmax, fabs(th));
    // ---- per-rollout outputs ----
    {
        double bad[N], big[N], piv[N];
        const bool fin = isfinite(th) && isfinite(thd) && isfinite(gdx) && isfinite(gdy);
        sw::RowGather<N>::run(fin ? 0.0 : 1.0, bad);
        sw::RowGather<N>::run(thmax, big);
        sw::RowGather<N>::run(rq_last, piv);      // every segment lane's last 1 / pivot
        double nbad = 0.0, tmax = 0.0, pmin = 1.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            nbad += bad[k];
            tmax = fmax(tmax, big[k]);
            pmin = fmin(pmin, piv[k]);
        }
        const int code = ((pmin > 0.0) ? 0 : SW_STATUS_SINGULAR) |
                         ((nbad == 0.0) ? 0 : SW_STATUS_NONFINITE) |
                         ((tmax < sw::kAngleLimit) ? 0 : SW_STATUS_RANGE);
        if (valid && q == 0) {
            // sum of the rewards Gdot_t . direction (remy_swimmer_env.py:238-243), by linearity
            const double total = __builtin_fma(C.dirx, sgx, C.diry * sgy);
            returns[r] = (code & SW_STATUS_RANGE) ? __builtin_nan("") : total;
            if (status) status[r] = code;
        }
    }
    if (final_state && valid && owner) {
        final_state[(int64_t)cth * n_roll + r] = th;
        final_state[(int64_t)cthd * n_roll + r] = thd;
        if (q < 2) final_state[(int64_t)q * n_roll + r] = (q == 0) ? gdx : gdy;
    }
    if (MOM) {
        __shared__ double shm[kRowBlock / kWave][16][6];
        double m1g = (q == 0) ? sgx : sgy, m2g = (q == 0) ? qgx : qgy;   // lane 0: x, lane 1: y
        if (!valid || !owner) m1th = m2th = m1thd = m2thd = m1g = m2g = 0.0;
        // sum over the 4 rows of the wave (lane bits 4, 5), then over the 4 waves through LDS
#pragma unroll
        for (int off = 16; off < kWave; off <<= 1) {
            m1th += __shfl_xor(m1th, off, kWave);
            m2th += __shfl_xor(m2th, off, kWave);
            m1thd += __shfl_xor(m1thd, off, kWave);
            m2thd += __shfl_xor(m2thd, off, kWave);
            m1g += __shfl_xor(m1g, off, kWave);
            m2g += __shfl_xor(m2g, off, kWave);
        }
        const int wv = tid / kWave, ln = tid % kWave;
        if (ln < 16) {
            shm[wv][ln][0] = m1th;
            shm[wv][ln][1] = m2th;
            shm[wv][ln][2] = m1thd;
            shm[wv][ln][3] = m2thd;
            shm[wv][ln][4] = m1g;
            shm[wv][ln][5] = m2g;
        }
        __syncthreads();
        if (tid < N) {
            double acc[6];
#pragma unroll
            for (int v = 0; v < 6; ++v)
                acc[v] = (shm[0][tid][v] + shm[1][tid][v]) + (shm[2][tid][v] + shm[3][tid][v]);
            double *row = moments + (int64_t)blockIdx.x * (2 * D);
            row[2 + 2 * tid] = acc[0];
            row[D + 2 + 2 * tid] = acc[1];
            row[3 + 2 * tid] = acc[2];
            row[D + 3 + 2 * tid] = acc[3];
            if (tid < 2) {
                row[tid] = acc[4];
                row[D + tid] = acc[5];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// The safe-exploration gate for n = 4..8 in the ROW form of rollout_row_kernel (one segment per lane, one rollout per
// 16-lane DPP row): per env-step two `row_step`s -- the simulator's look-ahead on copies of the state with the
// simulator's constants and policy scaling, then the real step on copies, committed through selects where the gate
// is open.  Gdot is replicated bit-identically on the row's lanes and lane i + 8 mirrors lane i bit for bit, so every
// lane of a row computes the SAME cost from broadcasts and the decision needs no vote.  ~2 x 239 instructions per
// env-step at n = 6 instead of ~2 x 1054 in the lane form.  Range checks per trip as in rollout_row_kernel (a refused
// rollout's angles do not move at all, so the trip's travel bound holds a fortiori).
template <int N>
__global__ void __launch_bounds__(kRowBlock)
safe_rollout_row_kernel(sw::Consts Cr, sw::Consts Cs, int64_t n_roll, int32_t H,
                        const double *__restrict__ policies, int32_t cost_kind, int32_t cost_index,
                        double sim_thresh, double real_thresh, int32_t want_violations,
                        double *__restrict__ returns, double *__restrict__ traj, int32_t has_traj,
                        int32_t *__restrict__ first_refused, int32_t *__restrict__ violations,
                        int32_t *__restrict__ status)
{
    __builtin_amdgcn_s_setprio(3);
    constexpr int D = 2 * N + 2, M = N - 1;
    const int tid = threadIdx.x;
    const int q = tid & 15;
    const bool owner = q < N;
    const bool cosine = q >= 8;
    const int seg = ((q & 7) < N) ? (q & 7) : 0;
    const int64_t r_raw = (int64_t)blockIdx.x * kMomGroup + (tid >> 4);
    const bool valid = r_raw < n_roll;
    const int64_t r = valid ? r_raw : n_roll - 1;
    const sw::RowLane<N> Lr = sw::row_lane<N>(Cr, seg), Ls = sw::row_lane<N>(Cs, seg);
    const int cth = 2 + 2 * seg, cthd = 3 + 2 * seg;
    double V[D], Vs[D], nbias, nbias_s;       // the policy rows scaled with each model's 12 / (m l^2)
    {
        int cols[D];
#pragma unroll
        for (int j = 0; j < D; ++j) cols[j] = j;
        load_policy_row<D, M, false>(policies + r * (M * D), nullptr, 1.0, 0.0, nullptr, nullptr, Cr.c12, seg, cols,
                                     V, nbias);
        load_policy_row<D, M, false>(policies + r * (M * D), nullptr, 1.0, 0.0, nullptr, nullptr, Cs.c12, seg, cols,
                                     Vs, nbias_s);
    }
    double gdx = 0.0, gdy = 0.0, th = kHalfPi, thd = 0.0;      // real_env.reset() (:133)
    const uint32_t slab = (uint32_t)(D * n_roll * 8);
    const uint32_t kDrop = 0xfffffff0u;
    const uint32_t off_th = (owner && valid) ? (uint32_t)(((int64_t)cth * n_roll + r) * 8) : kDrop;
    const uint32_t off_thd = (owner && valid) ? (uint32_t)(((int64_t)cthd * n_roll + r) * 8) : kDrop;
    const uint32_t off_gx = (q == 0 && valid) ? (uint32_t)(r * 8) : kDrop;
    const uint32_t off_gy = (q == 1 && valid) ? (uint32_t)((n_roll + r) * 8) : kDrop;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(
        traj, 0, has_traj ? (int)(uint32_t)((int64_t)H * slab) : 0, 0x00020000);
    uint32_t soff = 0;
    auto store_cell = [&](double v, uint32_t voff) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        union { double d; v2i i; } u;
        u.d = v;
        __builtin_amdgcn_raw_buffer_store_b64(u.i, trs, (int)voff, (int)soff, SW_TRAJ_STORE_AUX);
    };
    double thmax = 0.0, rq_last = 1.0, ret_x = 0.0, ret_y = 0.0;
    int32_t taken = 0, over = 0;
    bool alive = true;
    const int designation = cosine ? 1 : 0;
    sw::OctTrig A;
    A.r = th;
    A.kd = 0.0;
    sw::oct3_renorm(A, designation, thmax);
    // cost(obs) of a state held in row form: every lane of the row gets the same value (broadcasts of bit-identical
    // copies); index decoded once: which segment lane owns it, and whether it is theta or thetadot
    const int cseg = (cost_index >= 2) ? ((cost_index - 2) >> 1) : 0;
    const bool c_thd = cost_index >= 2 && ((cost_index - 2) & 1);
    auto cost_of = [&](double gx, double gy, double th_, double thd_) -> double {
        if (cost_kind == SW_COST_MAX_ABS_THETADOT) {
            double all[N];
            sw::RowGather<N>::run(thd_, all);
            double c = fabs(all[0]);
            bool nan = all[0] != all[0];
#pragma unroll
            for (int k = 1; k < N; ++k) {
                c = fmax(c, fabs(all[k]));
                nan = nan || (all[k] != all[k]);
            }
            return nan ? __builtin_nan("") : c;
        }
        if (cost_index == 0) return fabs(gx);
        if (cost_index == 1) return fabs(gy);
        double all[N];
        sw::RowGather<N>::run(c_thd ? thd_ : th_, all);
        double v = all[0];
#pragma unroll
        for (int k = 1; k < N; ++k) v = (cseg == k) ? all[k] : v;
        return fabs(v);
    };
    auto one_step = [&](auto slow) {
        // the simulator's look-ahead from the real state (:120-121), on copies
        double sgx = gdx, sgy = gdy, sth = th, sthd = thd, smax = 0.0;
        sw::OctTrig As = A;
        (void)sw::row_step<N, decltype(slow)::value>(Cs, Ls, Vs, nbias_s, cosine, designation, sgx, sgy, As, sth, sthd,
                                                     smax);
        const bool safe = (cost_of(sgx, sgy, sth, sthd) <= sim_thresh) && alive;      // :122, NaN refuses
        alive = safe;
        // the real step on copies, committed where the gate is open (:142)
        double rgx = gdx, rgy = gdy, rth = th, rthd = thd, rmax = thmax;
        sw::OctTrig Ar = A;
        const double rq = sw::row_step<N, decltype(slow)::value>(Cr, Lr, V, nbias, cosine, designation, rgx, rgy, Ar, rth,
                                                                 rthd, rmax);
        gdx = safe ? rgx : gdx;
        gdy = safe ? rgy : gdy;
        th = safe ? rth : th;
        thd = safe ? rthd : thd;
        A.r = safe ? Ar.r : A.r;
        if (decltype(slow)::value) {      // the checked loop re-normalises inside the step: take all of it
            A.kd = safe ? Ar.kd : A.kd;
            A.selS = safe ? Ar.selS : A.selS;
            A.selC = safe ? Ar.selC : A.selC;
#pragma unroll
            for (int k = 0; k < 7; ++k) A.k[k] = safe ? Ar.k[k] : A.k[k];
            thmax = safe ? rmax : thmax;
        }
        rq_last = safe ? rq : rq_last;
        ret_x += safe ? gdx : 0.0;
        ret_y += safe ? gdy : 0.0;
        taken += safe ? 1 : 0;
        if (want_violations) over += (safe && (cost_of(gdx, gdy, th, thd) > real_thresh)) ? 1 : 0;   // :143
        store_cell(th, off_th);
        store_cell(thd, off_thd);
        store_cell(gdx, off_gx);
        store_cell(gdy, off_gy);
        soff += slab;
    };
    auto too_fast = [&]() -> bool { return __any((4.0 * Cr.h) * fabs(thd) > sw::kTripSlack); };
    int32_t t = 0;
    while (t < H) {
        while (t < H) {                              // unchecked trips of (up to) four steps
            if (__builtin_expect(too_fast(), 0)) break;
            const double reach = __builtin_fma(4.0 * Cr.h, fabs(thd), fabs(A.r));
            if (__builtin_expect(__any(reach > sw::kPio4), 0)) sw::oct3_renorm(A, designation, thmax);
            const int32_t t_end = min(H, t + 4);
#pragma unroll 1
            for (; t < t_end; ++t) one_step(std::false_type{});
        }
#pragma unroll 1
        for (; t < H && too_fast(); ++t) one_step(std::true_type{});
    }
    thmax = fmax(thmax, fabs(th));
    {
        double bad[N], big[N], piv[N];
        const bool fin = isfinite(th) && isfinite(thd) && isfinite(gdx) && isfinite(gdy);
        sw::RowGather<N>::run(fin ? 0.0 : 1.0, bad);
        sw::RowGather<N>::run(thmax, big);
        sw::RowGather<N>::run(rq_last, piv);
        double nbad = 0.0, tmax = 0.0, pmin = 1.0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            nbad += bad[k];
            tmax = fmax(tmax, big[k]);
            pmin = fmin(pmin, piv[k]);
        }
        const int code = ((pmin > 0.0) ? 0 : SW_STATUS_SINGULAR) | ((nbad == 0.0) ? 0 : SW_STATUS_NONFINITE) |
                         ((tmax < sw::kAngleLimit) ? 0 : SW_STATUS_RANGE);
        if (valid && q == 0) {
            const double total = __builtin_fma(Cr.dirx, ret_x, Cr.diry * ret_y);
            returns[r] = (code & SW_STATUS_RANGE) ? __builtin_nan("") : total;
            if (first_refused) first_refused[r] = taken;
            if (violations) violations[r] = over;
            if (status) status[r] = code;
        }
    }
}

// ------------------------------------------------------------------------------------
// two sums with one pair of barriers (the update kernel is pure latency: every barrier counts)
template <int BLOCK>
__device__ __forceinline__ void block_sum2(double &a, double &b, double (*sh2)[2])
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        a += __shfl_down(a, off, kWave);
        b += __shfl_down(b, off, kWave);
    }
    const int w = threadIdx.x / kWave, l = threadIdx.x % kWave;
    __syncthreads();
    if (l == 0) {
        sh2[w][0] = a;
        sh2[w][1] = b;
    }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < BLOCK / kWave; ++i) {
        ta += sh2[i][0];
        tb += sh2[i][1];
    }
    a = ta;
    b = tb;
}

// Where the update finds an iteration's results.  After the all-gather every rank's segment
// [2*chunk returns | rows_chunk moment rows] sits at rank*seg_len in one buffer; the kernel
// indexes that layout directly so no repacking kernels run between the collective and the
// update.  Separate returns / moments arrays are the world = 1 special case.
struct GatherView {
    const double *ret_base;
    const double *mom_base;
    int64_t seg_len;      // doubles between consecutive ranks' segments
    int32_t chunk;        // direction slots per rank
    int32_t rows_chunk;   // moment rows per rank
    int32_t world;
};

__device__ __forceinline__ double ret_at(const GatherView &g, int32_t dir, int sign_idx)
{
    const int32_t rank = dir / g.chunk, local = dir - rank * g.chunk;
    return g.ret_base[rank * g.seg_len + 2 * local + sign_idx];
}

// used(i): all directions (top_b == 0) or the top_b by max(r+, r-), ties to the higher index
// (argsort ascending, reversed: ars_agent.py:105-108).  With top_b active every workgroup first
// stages the N keys in LDS and ranks them there (N^2 / 256 comparisons per thread), leaving a
// byte mask; directions beyond the LDS capacity fall back to ranking from global memory.
constexpr int kTopBMaxDirs = 6144;   // 48 KB of keys + 6 KB of flags

__device__ __forceinline__ bool rank_used_global(const GatherView &g, int32_t n_dir, int64_t top_b, int32_t i)
{
    const double ki = fmax(ret_at(g, i, 0), ret_at(g, i, 1));
    int64_t rank = 0;
    for (int32_t j = 0; j < n_dir; ++j) {
        const double kj = fmax(ret_at(g, j, 0), ret_at(g, j, 1));
        rank += (kj > ki) || (kj == ki && j > i);
    }
    return rank < top_b;
}

// grid = m*d + 1 workgroups.  Workgroup e < m*d updates policy entry e; the last one merges
// the V2 statistics.  The kernel sits on the critical path between two rollout launches and
// is pure latency, so every workgroup first pulls what it needs with ONE round of loads (the
// 2 n_dir returns and its delta column into LDS / registers; all moment rows in parallel) and
// then only touches LDS: ~5 us instead of ~17 us for the load-then-use-per-pass version.
constexpr int kUpdMaxDirs = kTopBMaxDirs;
constexpr int kTopBSortDirs = 2048;   // top-b by a bitonic sort in LDS up to here, by ranking beyond

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
ars_update_kernel(int d, int md, int32_t n_dir, GatherView gv,
                  const double *__restrict__ deltas, double *__restrict__ policy, double alpha,
                  double b, int64_t top_b, double *__restrict__ running, double n_new,
                  double *__restrict__ mean, double *__restrict__ inv_std,
                  double *__restrict__ sigma_out)
{
    __shared__ double sh2[BLOCK / kWave][2];
    __shared__ double rp_s[kUpdMaxDirs], rm_s[kUpdMaxDirs];   // r+ and r- of every direction
    __shared__ unsigned char flag[kUpdMaxDirs];
    const int e = blockIdx.x;
    if (e < md) {
        const bool select = top_b > 0 && top_b < n_dir;
        const bool in_lds = n_dir <= kUpdMaxDirs;
        // one round of global loads: returns -> LDS, this workgroup's delta column -> registers
        constexpr int kMaxPer = (kUpdMaxDirs + BLOCK - 1) / BLOCK;
        double dcol[kMaxPer];
        if (in_lds) {
#pragma unroll
            for (int q = 0; q < kMaxPer; ++q) {
                const int32_t i = threadIdx.x + q * BLOCK;
                dcol[q] = (i < n_dir) ? deltas[(int64_t)i * md + e] : 0.0;
            }
            for (int32_t i = threadIdx.x; i < n_dir; i += BLOCK) {
                rp_s[i] = ret_at(gv, i, 0);
                rm_s[i] = ret_at(gv, i, 1);
            }
            __syncthreads();
            if (select && n_dir <= kTopBSortDirs) {
                // Up to 2048 directions: a bitonic sort of (key, index) in LDS, best first -- key = max(r+, r-)
                // descending, ties to the higher index, NaN keys first (np.argsort puts NaN last and the reference
                // reverses it, ars_agent.py:105-108).  log2(P) (log2(P) + 1) / 2 compare-exchange stages of P / 2
                // pairs each (45 stages at 512 directions: ~4 us) instead of N^2 / BLOCK comparisons per thread with
                // the key list re-read for every direction (~25 us on the critical path between two rollout launches).
                __shared__ double skey[kTopBSortDirs];
                __shared__ uint16_t sidx[kTopBSortDirs];
                uint32_t P2 = 2;
                while (P2 < (uint32_t)n_dir) P2 <<= 1;
                for (uint32_t i = threadIdx.x; i < P2; i += BLOCK) {
                    double k = -HUGE_VAL;
                    if (i < (uint32_t)n_dir) {
                        // NOT fmax: Python's max(a, b) = (b > a) ? b : a (safe_ars / ars_agent sort_directions)
                        const double a = rp_s[i], b = rm_s[i];
                        k = (b > a) ? b : a;
                        k = (k != k) ? HUGE_VAL : k;
                    }
                    skey[i] = k;
                    sidx[i] = (uint16_t)i;
                }
                __syncthreads();
                for (uint32_t k = 2; k <= P2; k <<= 1) {
                    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                        for (uint32_t t = threadIdx.x; t < P2 / 2; t += BLOCK) {
                            const uint32_t lo = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), hi = lo | j;
                            const double ka = skey[lo], kb = skey[hi];
                            const uint32_t ia = sidx[lo], ib = sidx[hi];
                            // a goes before b?  padding (index >= n_dir) always goes last
                            const bool a_first = (ia < (uint32_t)n_dir) &&
                                                 ((ib >= (uint32_t)n_dir) || ka > kb || (ka == kb && ia > ib));
                            const bool best_first = (lo & k) == 0;     // direction of this bitonic block
                            if (a_first != best_first) {
                                skey[lo] = kb;
                                skey[hi] = ka;
                                sidx[lo] = (uint16_t)ib;
                                sidx[hi] = (uint16_t)ia;
                            }
                        }
                        __syncthreads();
                    }
                }
                for (uint32_t pos = threadIdx.x; pos < P2; pos += BLOCK) {
                    const uint32_t i = sidx[pos];
                    if (i < (uint32_t)n_dir) flag[i] = (int64_t)pos < top_b;
                }
                __syncthreads();
            } else if (select) {
                for (int32_t i = threadIdx.x; i < n_dir; i += BLOCK) {
                    const double ki = fmax(rp_s[i], rm_s[i]);
                    int32_t rank = 0;
                    for (int32_t j = 0; j < n_dir; ++j) {
                        const double kj = fmax(rp_s[j], rm_s[j]);
                        rank += (kj > ki) || (kj == ki && j > i);
                    }
                    flag[i] = rank < top_b;
                }
                __syncthreads();
            }
        }
        auto rplus = [&](int32_t i) { return in_lds ? rp_s[i] : ret_at(gv, i, 0); };
        auto rminus = [&](int32_t i) { return in_lds ? rm_s[i] : ret_at(gv, i, 1); };
        auto dir_used = [&](int32_t i) -> bool {
            if (!select) return true;
            return in_lds ? (flag[i] != 0) : rank_used_global(gv, n_dir, top_b, i);
        };
        // np.std(used_rewards): two-pass, ddof = 0 (ars_agent.py:123)
        double s = 0.0, cnt = 0.0;
        for (int32_t i = threadIdx.x; i < n_dir; i += BLOCK)
            if (dir_used(i)) {
                s += rplus(i) + rminus(i);
                cnt += 2.0;
            }
        block_sum2<BLOCK>(s, cnt, sh2);
        const double mu = s / cnt;
        double v = 0.0, g = 0.0;
        if (in_lds) {
#pragma unroll   // static index into dcol[] (a runtime index would send it to scratch)
            for (int q = 0; q < kMaxPer; ++q) {
                const int32_t i = threadIdx.x + q * BLOCK;
                if (i < n_dir && dir_used(i)) {
                    const double rp = rp_s[i], rm = rm_s[i];
                    const double a = rp - mu, c = rm - mu;
                    v += a * a + c * c;
                    g = __builtin_fma(rp - rm, dcol[q], g);
                }
            }
        } else {
            for (int32_t i = threadIdx.x; i < n_dir; i += BLOCK)
                if (dir_used(i)) {
                    const double rp = rplus(i), rm = rminus(i);
                    const double a = rp - mu, c = rm - mu;
                    v += a * a + c * c;
                    g = __builtin_fma(rp - rm, deltas[(int64_t)i * md + e], g);
                }
        }
        block_sum2<BLOCK>(v, g, sh2);
        if (threadIdx.x == 0) {
            const double sigma = sqrt(v / cnt);
            // divisor: b as given (ars_agent.py:128: all directions used, b only divides), or with
            // a true top-b truncation the number of directions used, len(order) (safe_ars/ars.py:64)
            const double div = (top_b > 0) ? 0.5 * cnt : b;
            const double grad = g / (div * sigma);
            policy[e] = policy[e] + alpha * grad;           // ars_agent.py:130
            if (e == 0 && sigma_out) *sigma_out = sigma;
        }
    } else if (running != nullptr) {
        // V2 statistics over every state seen since training began (np.mean / np.cov with
        // ddof = 1, ars_agent.py:179-182).  The reference recomputes them two-pass over the
        // whole (ever-growing) list; here `running` = {n, mean - c, M2 = sum (x - mean)^2} and
        // each iteration's batch is MERGED into it (Chan et al.): the batch's own mean and M2
        // come from its sums about the pivot c (reset state; every rollout starts there, so
        // |mean_b - c| is never large against the batch's spread), and the merge itself adds
        // non-negative terms only -- no cancellation that grows with the length of training.
        // The workgroup's 256 threads form G = 256 / 2d row groups x 2d columns: thread (rg, j) sums
        // column j over the rows whose GLOBAL index (rank-major) is congruent to rg mod G, in
        // ascending order, eight loads in flight per round (one round up to 8 G rows: 128 rows for
        // n = 3) -- the loop used to run over 4 row groups only and paid one memory latency per 16
        // rows, 4 rounds at 512 directions.  The G partial sums are added in ascending group
        // order.  Global row indices make the grouping -- and every bit of the result --
        // independent of the world size for row-aligned shards, and identical on every rank.
        constexpr int kMaxCols = 2 * (2 * SW_MAX_SEGMENTS + 2);     // 2d <= 36
        constexpr int kMaxGroups = BLOCK / 12;                  // 2d >= 12 (n = 2): G <= 21 (16 for n = 3)
        __shared__ double part[kMaxGroups][kMaxCols];
        __shared__ double bsum[kMaxCols];
        const int cols = 2 * d, G = BLOCK / cols;
        const int rg = threadIdx.x / cols, j = threadIdx.x - rg * cols;
        const int32_t total = gv.world * gv.rows_chunk;
        if (rg < G) {
            double acc = 0.0;
            for (int32_t g0 = rg; g0 < total; g0 += 8 * G) {
                double v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int32_t g = g0 + q * G;
                    v[q] = 0.0;
                    if (g < total) {
                        const int32_t r = g / gv.rows_chunk, row = g - r * gv.rows_chunk;
                        v[q] = gv.mom_base[r * gv.seg_len + (int64_t)row * cols + j];
                    }
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) acc += v[q];
            }
            part[rg][j] = acc;
        }
        __syncthreads();
        if (threadIdx.x < cols) {
            double t = part[0][threadIdx.x];
            for (int g = 1; g < G; ++g) t += part[g][threadIdx.x];
            bsum[threadIdx.x] = t;
        }
        __syncthreads();
        const double n0 = running[0], n1 = n0 + n_new;
        if (threadIdx.x < d && n_new > 0.0) {
            const int c_ = threadIdx.x;
            const double s1 = bsum[c_], s2 = bsum[d + c_];
            const double mb = s1 / n_new;                       // batch mean - c
            const double m2b = __builtin_fma(-s1, mb, s2);      // batch sum (x - mean_b)^2
            const double mr = running[1 + c_], m2 = running[1 + d + c_];
            const double delta = mb - mr;
            const double mr1 = __builtin_fma(delta, n_new / n1, mr);
            const double m21 = (m2 + m2b) + delta * delta * (n0 * (n_new / n1));
            running[1 + c_] = mr1;
            running[1 + d + c_] = m21;
            const double c = (c_ >= 2 && (c_ & 1) == 0) ? kHalfPi : 0.0;
            mean[c_] = c + mr1;
            inv_std[c_] = 1.0 / sqrt(m21 / (n1 - 1.0));          // diag(cov) ** -0.5
        }
        __syncthreads();
        if (threadIdx.x == 0) running[0] = n1;
    }
}

// ------------------------------------------------------------------------------------
// Calibration of the latency-bound rollouts' ceiling on THIS device: one wave issuing `trips` x 64
// independent instructions of one class (mode 0: v_fma_f64 on 8 accumulators; mode 1: v_mov_b32).
// bench.py times it with HIP events and prices the rollout kernel's per-step instruction mix
// with the two intervals (roofline.issue_bound).
// A grid of such waves (sw_issue_probe_grid: 256 workgroups x 4 waves = one wave on every SIMD) gives the
// same intervals with the WHOLE chip issuing -- f64 on every SIMD lowers the clock the chip sustains.
__global__ void __launch_bounds__(256) issue_probe_kernel(int32_t trips, int32_t mode, double *out)
{
    double a0 = 1.0 + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,
           a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    int b0 = threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6,
        b7 = b0 + 7;
#define SW_FMA8 "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n" \
                "v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n" \
                "v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
#define SW_MOV8 "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n" \
                "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
    if (mode == 0) {
        for (int32_t t = 0; t < trips; ++t)
            asm volatile(SW_FMA8 SW_FMA8 SW_FMA8 SW_FMA8 SW_FMA8 SW_FMA8 SW_FMA8 SW_FMA8
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(c));
    } else {
        for (int32_t t = 0; t < trips; ++t)
            asm volatile(SW_MOV8 SW_MOV8 SW_MOV8 SW_MOV8 SW_MOV8 SW_MOV8 SW_MOV8 SW_MOV8
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
    }
#undef SW_FMA8
#undef SW_MOV8
    // (every wave of a grid writes the same 64 doubles' worth of don't-care values)
    out[threadIdx.x % kWave] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (double)(b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7);
}

// ---- dispatch on the segment count -------------------------------------------------
#define SW_DISPATCH_N(n, CALL)                 \
    switch (n) {                               \
    case 2: { constexpr int NN = 2; CALL; } break; \
    case 3: { constexpr int NN = 3; CALL; } break; \
    case 4: { constexpr int NN = 4; CALL; } break; \
    case 5: { constexpr int NN = 5; CALL; } break; \
    case 6: { constexpr int NN = 6; CALL; } break; \
    case 7: { constexpr int NN = 7; CALL; } break; \
    case 8: { constexpr int NN = 8; CALL; } break; \
    default: return SW_ERR_SEGMENTS;           \
    }

#define SW_DISPATCH_QUAD(ARS, HAS_TRAJ, HAS_MOM, STREAM, ...)                                   \
    do {                                                                                         \
        if (HAS_TRAJ) {                                                                          \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_quad3_kernel<ARS, true, true>), dim3(grid),          \
                                   dim3(kRollBlock), 0, STREAM, __VA_ARGS__);                    \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_quad3_kernel<ARS, true, false>), dim3(grid),         \
                                   dim3(kRollBlock), 0, STREAM, __VA_ARGS__);                    \
        } else {                                                                                 \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_quad3_kernel<ARS, false, true>), dim3(grid),         \
                                   dim3(kRollBlock), 0, STREAM, __VA_ARGS__);                    \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_quad3_kernel<ARS, false, false>), dim3(grid),        \
                                   dim3(kRollBlock), 0, STREAM, __VA_ARGS__);                    \
        }                                                                                        \
    } while (0)

#define SW_DISPATCH_OCT(ARS, HAS_TRAJ, HAS_MOM, STREAM, ...)                                    \
    do {                                                                                         \
        if (HAS_TRAJ) {                                                                          \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_oct3_kernel<ARS, true, true>), dim3(grid),           \
                                   dim3(kOctBlock), 0, STREAM, __VA_ARGS__);                     \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_oct3_kernel<ARS, true, false>), dim3(grid),          \
                                   dim3(kOctBlock), 0, STREAM, __VA_ARGS__);                     \
        } else {                                                                                 \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_oct3_kernel<ARS, false, true>), dim3(grid),          \
                                   dim3(kOctBlock), 0, STREAM, __VA_ARGS__);                     \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_oct3_kernel<ARS, false, false>), dim3(grid),         \
                                   dim3(kOctBlock), 0, STREAM, __VA_ARGS__);                     \
        }                                                                                        \
    } while (0)

// Kernel choice for rollouts: the quad (segment-per-lane) kernel while it still finds idle
// SIMDs, the lane-per-rollout kernel beyond; sw_params.flags can force either.
bool use_quad3(const sw_params *p, int64_t n_roll, int32_t H, bool with_traj)
{
    if (p->n != 3 || is_twin(p)) return false;   // segment-per-lane kernels: Gym model only
    if (p->flags & SW_FLAG_ROLLOUT_LANE) return false;
    // the quad kernel addresses the trajectory buffer with 32-bit byte offsets
    if (with_traj && (int64_t)H * 8 * n_roll * 8 >= ((int64_t)1 << 32)) return false;
    if (n_roll >= ((int64_t)1 << 25)) return false;
    if (p->flags & SW_FLAG_ROLLOUT_QUAD) return true;
    return n_roll <= kQuadMaxRollouts;
}

// n = 3 with lane roles (two mirror quads per rollout, swimmer_oct3.h): 8 rollouts per wave, so it
// keeps one wave per SIMD up to 8192 rollouts; beyond that the quad kernel (16 per wave) takes over.
// SWIMMER_N3_KERNEL=quad|oct overrides the default (measurement knob).
constexpr int64_t kOctMaxRollouts = 8192;
bool use_oct3(const sw_params *p, int64_t n_roll, int32_t H, bool with_traj)
{
    static const char *env = getenv("SWIMMER_N3_KERNEL");
    const bool want = env ? (env[0] == 'o') : SW_N3_DEFAULT_OCT;
    return want && n_roll <= kOctMaxRollouts && use_quad3(p, n_roll, H, with_traj);
}

// n = 4..8: the row (segment-per-lane) kernel while it still finds idle SIMDs.
bool use_row(const sw_params *p, int64_t n_roll, int32_t H, bool with_traj)
{
    if (p->n < 4 || is_twin(p)) return false;
    if (p->flags & SW_FLAG_ROLLOUT_LANE) return false;
    if (with_traj && (int64_t)H * (2 * p->n + 2) * n_roll * 8 >= ((int64_t)1 << 32) - 256) return false;
    if (n_roll >= ((int64_t)1 << 24)) return false;
    if (p->flags & SW_FLAG_ROLLOUT_QUAD) return true;
    return n_roll <= kRowMaxRollouts;
}

#define SW_LAUNCH_ROW(NN, ARS, HAS_TRAJ, HAS_MOM, STREAM, ...)                                  \
    do {                                                                                         \
        if (HAS_TRAJ) {                                                                          \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_row_kernel<NN, ARS, true, true>), dim3(grid),        \
                                   dim3(kRowBlock), 0, STREAM, __VA_ARGS__);                     \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_row_kernel<NN, ARS, true, false>), dim3(grid),       \
                                   dim3(kRowBlock), 0, STREAM, __VA_ARGS__);                     \
        } else {                                                                                 \
            if (HAS_MOM)                                                                         \
                hipLaunchKernelGGL((rollout_row_kernel<NN, ARS, false, true>), dim3(grid),       \
                                   dim3(kRowBlock), 0, STREAM, __VA_ARGS__);                     \
            else                                                                                 \
                hipLaunchKernelGGL((rollout_row_kernel<NN, ARS, false, false>), dim3(grid),      \
                                   dim3(kRowBlock), 0, STREAM, __VA_ARGS__);                     \
        }                                                                                        \
    } while (0)

#define SW_DISPATCH_ROW(n, ARS, HAS_TRAJ, HAS_MOM, STREAM, ...)                                 \
    switch (n) {                                                                                 \
    case 4: SW_LAUNCH_ROW(4, ARS, HAS_TRAJ, HAS_MOM, STREAM, __VA_ARGS__); break;                \
    case 5: SW_LAUNCH_ROW(5, ARS, HAS_TRAJ, HAS_MOM, STREAM, __VA_ARGS__); break;                \
    case 6: SW_LAUNCH_ROW(6, ARS, HAS_TRAJ, HAS_MOM, STREAM, __VA_ARGS__); break;                \
    case 7: SW_LAUNCH_ROW(7, ARS, HAS_TRAJ, HAS_MOM, STREAM, __VA_ARGS__); break;                \
    default: SW_LAUNCH_ROW(8, ARS, HAS_TRAJ, HAS_MOM, STREAM, __VA_ARGS__); break;               \
    }

int launch_status()
{
    return hipGetLastError() == hipSuccess ? SW_OK : SW_ERR_LAUNCH;
}

const SideJob kNoSide{nullptr, 0u, UINT32_MAX, 1u, 0u, 0, 0, 0, 0, nullptr, nullptr};

// Attach a covariance pass over (cov_traj, cov_rolls, cov_H) to a launch of `roll_blocks` rollout
// workgroups of `block` threads; returns the number of extra workgroups.
unsigned side_attach_cov(SideJob &sj, unsigned roll_blocks, int block, int sj_D)
{
    sj.first_cov_block = roll_blocks;
    if (!sj.cov_traj || sj.cov_rolls <= 0 || sj.cov_H <= 0) {
        sj.first_cov_block = UINT32_MAX;
        return 0;
    }
    const CovTiling t = cov_tiling(sj.cov_rolls, sj.cov_H, block, sj_D, true);
    sj.cov_nbx = t.nbx;
    sj.cov_tchunk = t.tchunk;
    sj.cov_tiles = t.nbx * t.ny;
    static const char *nap_env = getenv("SWIMMER_COV_NAP");   // measurement knob
    sj.cov_nap = nap_env ? atoi(nap_env) : 0;
    return sj.cov_tiles;
}

}  // namespace

// =====================================================================================
extern "C" {

int sw_abi_version(void) { return SW_ABI_VERSION; }
int sw_max_segments(void) { return SW_MAX_SEGMENTS; }

const char *sw_strerror(int code)
{
    switch (code) {
    case SW_OK: return "ok";
    case SW_ERR_NULL: return "a required pointer is NULL";
    case SW_ERR_SEGMENTS: return "number of segments outside 2..8";
    case SW_ERR_SIZE: return "bad size argument";
    case SW_ERR_PARAM: return "non-finite or non-positive physical parameter";
    case SW_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error code";
    }
}

int sw_issue_probe(int32_t mode, int32_t trips, double *scratch64, void *stream)
{
    (void)hipGetLastError();
    if (!scratch64) return SW_ERR_NULL;
    if (trips < 0 || (mode != 0 && mode != 1)) return SW_ERR_SIZE;
    hipLaunchKernelGGL(issue_probe_kernel, dim3(1), dim3(kWave), 0, (hipStream_t)stream, trips, mode,
                       scratch64);
    return launch_status();
}

int sw_issue_probe_grid(int32_t mode, int32_t trips, int32_t workgroups, int32_t waves_per_workgroup,
                        double *scratch64, void *stream)
{
    (void)hipGetLastError();
    if (!scratch64) return SW_ERR_NULL;
    if (trips < 0 || (mode != 0 && mode != 1) || workgroups < 1 || workgroups > 65536 ||
        waves_per_workgroup < 1 || waves_per_workgroup > 4)
        return SW_ERR_SIZE;
    hipLaunchKernelGGL(issue_probe_kernel, dim3((unsigned)workgroups), dim3(kWave * waves_per_workgroup), 0,
                       (hipStream_t)stream, trips, mode, scratch64);
    return launch_status();
}

int64_t sw_moments_blocks(int64_t n_roll)
{
    return n_roll <= 0 ? 0 : (n_roll + kMomGroup - 1) / kMomGroup;
}

int sw_reset_f64(const sw_params *p, int64_t n_env, double *state, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!state) return SW_ERR_NULL;
    if (n_env < 0) return SW_ERR_SIZE;
    if (n_env == 0) return SW_OK;
    const unsigned grid = (unsigned)((n_env + 255) / 256);
    hipLaunchKernelGGL(reset_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p->n,
                       is_twin(p) ? 1 : 0, n_env, state);
    return launch_status();
}

int sw_step_f64(const sw_params *p, int64_t n_env, const double *state_in, const double *action,
                double *state_out, double *reward, int32_t *status, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_env < 0) return SW_ERR_SIZE;
    if (n_env == 0) return SW_OK;
    if (!state_in || !action || !state_out) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    const unsigned grid = (unsigned)((n_env + kStepBlock - 1) / kStepBlock);
    const sw::TwinConsts T = make_twin_consts(p);
    const int d = 2 * p->n + 2;
    bool nt = n_env * (int64_t)(8 * (2 * d + p->n)) > kStepStreamBytes;
    static const char *nt_env = getenv("SWIMMER_STEP_NT");   // measurement knob: "0" / "1" force it
    if (nt_env && (nt_env[0] == '0' || nt_env[0] == '1')) nt = nt_env[0] == '1';
    if (is_twin(p)) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((step_kernel<NN, true, false>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, T, n_env, state_in, action,
                                               state_out, reward, status));
    } else if (nt) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((step_kernel<NN, false, true>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, T, n_env, state_in, action,
                                               state_out, reward, status));
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((step_kernel<NN, false, false>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, T, n_env, state_in, action,
                                               state_out, reward, status));
    }
    return launch_status();
}

int sw_step_residual_f64(const sw_params *p, int64_t n_env, const double *state, const double *action,
                         const double *next_ref, double *partial, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (is_twin(p)) return SW_ERR_PARAM;          // the estimator's objective is defined on the Gym model
    if (n_env < 0) return SW_ERR_SIZE;
    if (n_env == 0) return SW_OK;
    if (!state || !action || !next_ref || !partial) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    const unsigned grid = (unsigned)((n_env + kStepBlock - 1) / kStepBlock);
    const int d = 2 * p->n + 2;
    const bool nt = n_env * (int64_t)(8 * (2 * d + p->n)) > kStepStreamBytes;
    if (nt) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((step_residual_kernel<NN, true>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, n_env, state, action, next_ref, partial));
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((step_residual_kernel<NN, false>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, n_env, state, action, next_ref, partial));
    }
    return launch_status();
}

int64_t sw_step_residual_blocks(int64_t n_env) { return n_env <= 0 ? 0 : (n_env + kStepBlock - 1) / kStepBlock; }

int sw_accel_f64(const sw_params *p, int64_t n_env, const double *state, const double *action,
                 double *gdd, double *tdd, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_env < 0) return SW_ERR_SIZE;
    if (n_env == 0) return SW_OK;
    if (!state || !action || !gdd || !tdd) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    const unsigned grid = (unsigned)((n_env + kStepBlock - 1) / kStepBlock);
    const sw::TwinConsts T = make_twin_consts(p);
    if (is_twin(p)) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((accel_kernel<NN, true>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, T, n_env, state, action, gdd, tdd));
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((accel_kernel<NN, false>), dim3(grid), dim3(kStepBlock), 0,
                                               (hipStream_t)stream, C, T, n_env, state, action, gdd, tdd));
    }
    return launch_status();
}

int sw_rollout_f64(const sw_params *p, int64_t n_roll, int32_t H, const double *policies,
                   const double *mean, const double *inv_std, const double *state0,
                   double *returns, double *traj, double *final_state, double *moments,
                   int32_t *status, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_roll < 0 || H < 0) return SW_ERR_SIZE;
    if (n_roll == 0) return SW_OK;
    if (!policies || !returns) return SW_ERR_NULL;
    if ((mean == nullptr) != (inv_std == nullptr)) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    if (use_oct3(p, n_roll, H, traj != nullptr)) {
        const unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SW_DISPATCH_OCT(false, traj != nullptr, moments != nullptr,
                        (hipStream_t)stream, C, n_roll, H, policies, (const double *)nullptr,
                        (int64_t)0, 0.0, mean, inv_std, state0, returns, traj, final_state,
                        moments, status, kNoSide);
        return launch_status();
    }
    if (use_quad3(p, n_roll, H, traj != nullptr)) {
        const unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SW_DISPATCH_QUAD(false, traj != nullptr, moments != nullptr,
                         (hipStream_t)stream, C, n_roll, H, policies, (const double *)nullptr,
                         (int64_t)0, 0.0, mean, inv_std, state0, returns, traj, final_state,
                         moments, status, kNoSide);
        return launch_status();
    }
    if (use_row(p, n_roll, H, traj != nullptr)) {
        const unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SW_DISPATCH_ROW(p->n, false, traj != nullptr, moments != nullptr, (hipStream_t)stream, C,
                        n_roll, H, policies, (const double *)nullptr, (int64_t)0, 0.0, mean,
                        inv_std, state0, returns, traj, final_state, moments, status, kNoSide);
        return launch_status();
    }
    const unsigned grid = (unsigned)((n_roll + kRollBlock - 1) / kRollBlock);
    const sw::TwinConsts T = make_twin_consts(p);
    if (is_twin(p)) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((rollout_kernel<NN, false, true>), dim3(grid),
                                               dim3(kRollBlock), 0, (hipStream_t)stream, C, T, n_roll, H,
                                               policies, (const double *)nullptr, (int64_t)0, 0.0, mean,
                                               inv_std, state0, returns, traj, final_state, moments,
                                               status));
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((rollout_kernel<NN, false, false>), dim3(grid),
                                               dim3(kRollBlock), 0, (hipStream_t)stream, C, T, n_roll, H,
                                               policies, (const double *)nullptr, (int64_t)0, 0.0, mean,
                                               inv_std, state0, returns, traj, final_state, moments,
                                               status));
    }
    return launch_status();
}

int sw_safe_rollouts_f64(const sw_params *real, const sw_params *sim, int64_t n_roll, int32_t H,
                         const double *policies, int32_t cost_kind, int32_t cost_index, double sim_thresh,
                         double real_thresh, double *returns, double *traj, int32_t *first_refused,
                         int32_t *violations, int32_t *status, void *stream)
{
    int rc = check_params(real);
    if (rc) return rc;
    rc = validate_params(sim);
    if (rc) return rc;
    if (real->n != sim->n) return SW_ERR_SEGMENTS;
    if (n_roll < 0 || H < 0) return SW_ERR_SIZE;
    if (cost_kind != SW_COST_ABS_OBS && cost_kind != SW_COST_MAX_ABS_THETADOT) return SW_ERR_SIZE;
    if (cost_kind == SW_COST_ABS_OBS && (cost_index < 0 || cost_index >= 2 * real->n + 2)) return SW_ERR_SIZE;
    if (!(sim_thresh == sim_thresh) || !(real_thresh == real_thresh)) return SW_ERR_PARAM;
    if (n_roll == 0) return SW_OK;
    if (!policies || !returns) return SW_ERR_NULL;
    const sw::Consts Cr = make_consts(real), Cs = make_consts(sim);
    if (use_oct3(real, n_roll, H, traj != nullptr)) {
        // n = 3, up to 8192 rollouts: the mirror-quad form (one geometry, two dynamics per env-step)
        const unsigned ogrid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        const double tq_ratio = Cs.c12 / Cr.c12;
#define SW_LAUNCH_SAFE_OCT(TRAJ, VIOL)                                                                          \
    hipLaunchKernelGGL((safe_rollout_oct3_kernel<TRAJ, VIOL>), dim3(ogrid), dim3(kOctBlock), 0,                 \
                       (hipStream_t)stream, Cr, Cs, tq_ratio, n_roll, H, policies, cost_kind, cost_index,       \
                       sim_thresh, real_thresh, returns, traj, first_refused, violations, status)
        if (traj) {
            if (violations) SW_LAUNCH_SAFE_OCT(true, true); else SW_LAUNCH_SAFE_OCT(true, false);
        } else {
            if (violations) SW_LAUNCH_SAFE_OCT(false, true); else SW_LAUNCH_SAFE_OCT(false, false);
        }
#undef SW_LAUNCH_SAFE_OCT
        return launch_status();
    }
    if (use_row(real, n_roll, H, traj != nullptr)) {
        // n = 4..8 while SIMDs are idle: the row form (two row_steps per env-step)
        const unsigned rgrid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
#define SW_LAUNCH_SAFE_ROW(NN_)                                                                                 \
    hipLaunchKernelGGL((safe_rollout_row_kernel<NN_>), dim3(rgrid), dim3(kRowBlock), 0, (hipStream_t)stream, Cr, \
                       Cs, n_roll, H, policies, cost_kind, cost_index, sim_thresh, real_thresh,                 \
                       violations ? 1 : 0, returns, traj, traj ? 1 : 0, first_refused, violations, status)
        switch (real->n) {
        case 4: SW_LAUNCH_SAFE_ROW(4); break;
        case 5: SW_LAUNCH_SAFE_ROW(5); break;
        case 6: SW_LAUNCH_SAFE_ROW(6); break;
        case 7: SW_LAUNCH_SAFE_ROW(7); break;
        default: SW_LAUNCH_SAFE_ROW(8); break;
        }
#undef SW_LAUNCH_SAFE_ROW
        return launch_status();
    }
    const unsigned grid = (unsigned)((n_roll + kRollBlock - 1) / kRollBlock);
    SW_DISPATCH_N(real->n, hipLaunchKernelGGL((safe_rollout_kernel<NN>), dim3(grid), dim3(kRollBlock), 0,
                                              (hipStream_t)stream, Cr, Cs, n_roll, H, policies, cost_kind,
                                              cost_index, sim_thresh, real_thresh, returns, traj, first_refused,
                                              violations, status));
    return launch_status();
}

// side: what the launch carries besides the rollouts (pipeline only); *side_taken tells whether
// the chosen kernel could take it (the segment-per-lane kernels can, the lane kernel cannot).
static int launch_ars_rollouts(const sw_params *p, int64_t dir_begin, int64_t n_dir, int32_t H,
                               const double *policy, const double *deltas, double nu,
                               const double *mean, const double *inv_std, double *returns,
                               double *traj, double *moments, int32_t *status, void *stream,
                               const SideJob *side, bool *side_taken)
{
    int rc = validate_params(p);   // the public entry points have cleared stale errors already
    if (rc) return rc;
    if (n_dir < 0 || H < 0 || dir_begin < 0) return SW_ERR_SIZE;
    if (side_taken) *side_taken = false;
    if (n_dir == 0) return SW_OK;
    if (!policy || !deltas || !returns) return SW_ERR_NULL;
    if ((mean == nullptr) != (inv_std == nullptr)) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    const int64_t n_roll = 2 * n_dir;
    if (use_oct3(p, n_roll, H, traj != nullptr)) {
        unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SideJob sj = side ? *side : kNoSide;
        grid += side_attach_cov(sj, grid, kOctBlock, 2 * p->n + 2);
        if (side_taken) *side_taken = side != nullptr;
        SW_DISPATCH_OCT(true, traj != nullptr, moments != nullptr,
                        (hipStream_t)stream, C, n_roll, H, policy, deltas, dir_begin, nu, mean,
                        inv_std, (const double *)nullptr, returns, traj, (double *)nullptr,
                        moments, status, sj);
        return launch_status();
    }
    if (use_quad3(p, n_roll, H, traj != nullptr)) {
        unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SideJob sj = side ? *side : kNoSide;
        grid += side_attach_cov(sj, grid, kRollBlock, 2 * p->n + 2);
        if (side_taken) *side_taken = side != nullptr;
        SW_DISPATCH_QUAD(true, traj != nullptr, moments != nullptr,
                         (hipStream_t)stream, C, n_roll, H, policy, deltas, dir_begin, nu, mean,
                         inv_std, (const double *)nullptr, returns, traj, (double *)nullptr,
                         moments, status, sj);
        return launch_status();
    }
    if (use_row(p, n_roll, H, traj != nullptr)) {
        unsigned grid = (unsigned)((n_roll + kMomGroup - 1) / kMomGroup);
        SideJob sj = side ? *side : kNoSide;
        grid += side_attach_cov(sj, grid, kRowBlock, 2 * p->n + 2);
        if (side_taken) *side_taken = side != nullptr;
        SW_DISPATCH_ROW(p->n, true, traj != nullptr, moments != nullptr, (hipStream_t)stream, C,
                        n_roll, H, policy, deltas, dir_begin, nu, mean, inv_std,
                        (const double *)nullptr, returns, traj, (double *)nullptr, moments, status,
                        sj);
        return launch_status();
    }
    const unsigned grid = (unsigned)((n_roll + kRollBlock - 1) / kRollBlock);
    const sw::TwinConsts T = make_twin_consts(p);
    if (is_twin(p)) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((rollout_kernel<NN, true, true>), dim3(grid),
                                               dim3(kRollBlock), 0, (hipStream_t)stream, C, T, n_roll, H,
                                               policy, deltas, dir_begin, nu, mean, inv_std,
                                               (const double *)nullptr, returns, traj, (double *)nullptr,
                                               moments, status));
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((rollout_kernel<NN, true, false>), dim3(grid),
                                               dim3(kRollBlock), 0, (hipStream_t)stream, C, T, n_roll, H,
                                               policy, deltas, dir_begin, nu, mean, inv_std,
                                               (const double *)nullptr, returns, traj, (double *)nullptr,
                                               moments, status));
    }
    return launch_status();
}

int sw_ars_rollouts_f64(const sw_params *p, int64_t dir_begin, int64_t n_dir, int32_t H,
                        const double *policy, const double *deltas, double nu, const double *mean,
                        const double *inv_std, double *returns, double *traj, double *moments,
                        int32_t *status, void *stream)
{
    (void)hipGetLastError();   // public entry point: drop a stale error once (see check_params)
    return launch_ars_rollouts(p, dir_begin, n_dir, H, policy, deltas, nu, mean, inv_std, returns,
                               traj, moments, status, stream, nullptr, nullptr);
}

static int launch_update(const sw_params *p, int64_t n_dir, const GatherView &gv,
                         const double *deltas, double *policy, double alpha, double b,
                         int64_t top_b, double *running, int64_t n_new_states, double *mean,
                         double *inv_std, double *sigma_out, void *stream)
{
    const int d = 2 * p->n + 2, md = (p->n - 1) * d;
    // The kernel is pure latency between two rollout launches; a thread's share of the directions (and of
    // the moment rows) sets it.  256 threads per workgroup up to 1024 directions, 1024 beyond: 2048
    // directions 11.6 -> ~6 us (rocprofv3).  The summation order is a function of n_dir only, so every
    // rank of a sharded run and the single-process run of the same problem still get the same bits.
    if (n_dir >= kUpdWideFrom)
        hipLaunchKernelGGL(ars_update_kernel<kUpdBlockWide>, dim3(md + 1), dim3(kUpdBlockWide), 0,
                           (hipStream_t)stream, d, md, (int32_t)n_dir, gv, deltas, policy, alpha, b, top_b,
                           running, (double)n_new_states, mean, inv_std, sigma_out);
    else
        hipLaunchKernelGGL(ars_update_kernel<kUpdBlock>, dim3(md + 1), dim3(kUpdBlock), 0, (hipStream_t)stream,
                           d, md, (int32_t)n_dir, gv, deltas, policy, alpha, b, top_b, running,
                           (double)n_new_states, mean, inv_std, sigma_out);
    return launch_status();
}

int sw_ars_update_f64(const sw_params *p, int64_t n_dir, const double *returns,
                      const double *deltas, double *policy, double alpha, double b, int64_t top_b,
                      const double *moments, int64_t n_moment_rows, double *running,
                      int64_t n_new_states, double *mean, double *inv_std, double *sigma_out,
                      void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_dir <= 0 || n_dir > INT32_MAX / 4 || n_moment_rows < 0 || n_moment_rows > INT32_MAX ||
        n_new_states < 0)
        return SW_ERR_SIZE;
    if (!returns || !deltas || !policy) return SW_ERR_NULL;
    if (running && (!moments || !mean || !inv_std)) return SW_ERR_NULL;
    const GatherView gv{returns, moments, 0, (int32_t)n_dir, (int32_t)n_moment_rows, 1};
    return launch_update(p, n_dir, gv, deltas, policy, alpha, b, top_b, running, n_new_states,
                         mean, inv_std, sigma_out, stream);
}

int sw_ars_update_gathered_f64(const sw_params *p, int64_t n_dir, const double *gathered,
                               int32_t world, int64_t chunk, int64_t rows_chunk,
                               const double *deltas, double *policy, double alpha, double b,
                               int64_t top_b, double *running, int64_t n_new_states, double *mean,
                               double *inv_std, double *sigma_out, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n_dir <= 0 || n_dir > INT32_MAX / 4 || world < 1 || chunk < 1 || rows_chunk < 0 ||
        chunk > INT32_MAX / 4 || rows_chunk > INT32_MAX || (int64_t)world * chunk < n_dir ||
        n_new_states < 0)
        return SW_ERR_SIZE;
    if (!gathered || !deltas || !policy) return SW_ERR_NULL;
    if (running && (!mean || !inv_std)) return SW_ERR_NULL;
    const int d = 2 * p->n + 2;
    const int64_t seg_len = 2 * chunk + rows_chunk * 2 * d;
    const GatherView gv{gathered, gathered + 2 * chunk, seg_len, (int32_t)chunk,
                        (int32_t)rows_chunk, world};
    return launch_update(p, n_dir, gv, deltas, policy, alpha, b, top_b, running, n_new_states,
                         mean, inv_std, sigma_out, stream);
}

// One covariance pass with workgroups of `block` (64 or 256) threads.  The pipeline runs the pass
// it still owes with the block size of the rollout kernel that would have carried it, so a flushed
// pass sums in exactly the order the ride-along pass would have (bit-identical resume).
static int launch_traj_moments(const sw_params *p, int64_t n_roll, int32_t H, const double *traj,
                               double *acc, int block, bool riding, void *stream)
{
    if (n_roll < 0 || H < 0) return SW_ERR_SIZE;
    if (n_roll == 0 || H == 0) return SW_OK;
    if (!traj || !acc) return SW_ERR_NULL;
    const CovTiling t = cov_tiling(n_roll, H, block, 2 * p->n + 2, riding);
    const dim3 grid(t.nbx * t.ny);
    if (block == kRollBlock) {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((traj_moments_kernel<2 * NN + 2, kRollBlock>), grid,
                                               dim3(kRollBlock), 0, (hipStream_t)stream, n_roll, H,
                                               traj, acc, t.nbx, t.tchunk));
    } else if (block == kOctBlock) {   // owed by a mirror-quad launch (n = 3 only)
        if (p->n != 3) return SW_ERR_SIZE;
        hipLaunchKernelGGL((traj_moments_kernel<8, kOctBlock>), grid, dim3(kOctBlock), 0,
                           (hipStream_t)stream, n_roll, H, traj, acc, t.nbx, t.tchunk);
    } else {
        SW_DISPATCH_N(p->n, hipLaunchKernelGGL((traj_moments_kernel<2 * NN + 2, kMomBlock>), grid,
                                               dim3(kMomBlock), 0, (hipStream_t)stream, n_roll, H,
                                               traj, acc, t.nbx, t.tchunk));
    }
    return launch_status();
}

int64_t sw_cov_acc_doubles(const sw_params *p, int64_t n_roll, int32_t H)
{
    if (validate_params(p) != SW_OK || n_roll < 0 || H < 0) return -1;   // pure host arithmetic
    const int d = 2 * p->n + 2;
    int64_t tiles = 0;
    if (n_roll > 0 && H > 0) {
        for (int riding = 0; riding < 2; ++riding)
            for (int block : {kRollBlock, kOctBlock, kMomBlock}) {
                const CovTiling t = cov_tiling(n_roll, H, block, d, riding != 0);
                const int64_t n = (int64_t)t.nbx * t.ny;
                tiles = n > tiles ? n : tiles;
            }
    }
    return cov_sums(d) + 1 + tiles * (d + d * d);
}

int sw_traj_moments_f64(const sw_params *p, int64_t n_roll, int32_t H, const double *traj,
                        double *acc, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    return launch_traj_moments(p, n_roll, H, traj, acc, kMomBlock, false, stream);
}

// ---- one swimmer per call (include/swimmer_hip.h, sw_env1) -----------------------------
struct sw_env1 {
    double *io_host = nullptr, *io_dev = nullptr;   // SW_ENV1_DOUBLES doubles + {status, seq}
    hipStream_t stream = nullptr;
    uint32_t seq = 0;
    int device = 0;   // the device that was current at sw_env1_create: every launch goes there
};

int sw_env1_create(sw_env1 **out)
{
    if (!out) return SW_ERR_NULL;
    sw_env1 *e = new (std::nothrow) sw_env1();
    if (!e) return SW_ERR_LAUNCH;
    const size_t bytes = sizeof(double) * SW_ENV1_DOUBLES + 64;
    bool ok = hipGetDevice(&e->device) == hipSuccess;
    ok = ok && hipHostMalloc((void **)&e->io_host, bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
    if (ok) {
        memset(e->io_host, 0, bytes);
        ok = hipHostGetDevicePointer((void **)&e->io_dev, e->io_host, 0) == hipSuccess &&
             hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess;
        // (normal priority: a batch-1 step has nothing to overtake)
    }
    if (!ok) {
        sw_env1_destroy(e);
        return SW_ERR_LAUNCH;
    }
    *out = e;
    return SW_OK;
}

void sw_env1_destroy(sw_env1 *e)
{
    if (!e) return;
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamDestroy(e->stream);
    }
    if (e->io_host) (void)hipHostFree(e->io_host);
    delete e;
}

double *sw_env1_io(sw_env1 *e) { return e ? e->io_host : nullptr; }

static int env1_run(sw_env1 *e, const sw_params *p, bool accel, int32_t *status)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!e) return SW_ERR_NULL;
    const sw::Consts C = make_consts(p);
    const sw::TwinConsts T = make_twin_consts(p);
    int32_t *st_host = reinterpret_cast<int32_t *>(e->io_host + SW_ENV1_DOUBLES);
    int32_t *st_dev = reinterpret_cast<int32_t *>(e->io_dev + SW_ENV1_DOUBLES);
    const volatile uint32_t *flag_host = reinterpret_cast<const volatile uint32_t *>(st_host + 1);
    uint32_t *flag_dev = reinterpret_cast<uint32_t *>(st_dev + 1);
    const uint32_t seq = ++e->seq;
    // the handle's stream and mapped block belong to e->device: launch there whatever the caller's current
    // device is, and leave the caller's current device as it was
    int caller_device = e->device;
    if (hipGetDevice(&caller_device) != hipSuccess) return SW_ERR_LAUNCH;
    if (caller_device != e->device && hipSetDevice(e->device) != hipSuccess) return SW_ERR_LAUNCH;
#define SW_ENV1_LAUNCH(TW, AC)                                                                     \
    SW_DISPATCH_N(p->n, hipLaunchKernelGGL((env1_kernel<NN, TW, AC>), dim3(1), dim3(kWave), 0,     \
                                           e->stream, C, T, e->io_dev, st_dev, flag_dev, seq))
    if (is_twin(p)) {
        if (accel) { SW_ENV1_LAUNCH(true, true); } else { SW_ENV1_LAUNCH(true, false); }
    } else {
        if (accel) { SW_ENV1_LAUNCH(false, true); } else { SW_ENV1_LAUNCH(false, false); }
    }
#undef SW_ENV1_LAUNCH
    rc = launch_status();
    if (caller_device != e->device) (void)hipSetDevice(caller_device);
    if (rc) return rc;
    for (int64_t spins = 0; *flag_host != seq; ++spins) {
        if ((spins & 0xffff) == 0xffff) {
            // not hot any more: a stream that has drained without the flag moving has failed
            const hipError_t q = hipStreamQuery(e->stream);
            if (q == hipSuccess) {
                if (*flag_host == seq) break;
                return SW_ERR_LAUNCH;
            }
            if (q != hipErrorNotReady) return SW_ERR_LAUNCH;
        }
        __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (status) *status = accel ? 0 : *st_host;
    return SW_OK;
}

int sw_env1_step(sw_env1 *e, const sw_params *p, int32_t *status) { return env1_run(e, p, false, status); }
int sw_env1_accel(sw_env1 *e, const sw_params *p) { return env1_run(e, p, true, nullptr); }

// ---- ARS iteration pipeline ---------------------------------------------------------
// Host-side enqueue logic of one ARS iteration in native code: the caller's stream (the
// critical path: rollouts -> [all-gather] -> update), a copy stream for the H2D of the deltas,
// and a ring of SW_PIPELINE_SLOTS buffer slots per process.
//
// Measured on MI355X (profiles/), in the order the design reacted to it:
//  * a device-side cross-stream wait in front of the rollout kernel (hipStreamWaitEvent on the
//    H2D copy or on a covariance pass) delays that kernel by 13-18 us even when the awaited
//    work finished long ago -> the critical stream carries NO device-side waits; the ring is
//    deep enough that every dependency completed iterations earlier and the host only confirms;
//  * an event RECORD between two kernels of the critical stream costs ~4 us -> none either:
//    every rollout launch stores its index to a host-visible flag when it starts (SideJob),
//    which tells the host that everything enqueued before it -- the previous update included --
//    has completed;
//  * a covariance pass launched as its own kernel on a side stream costs the concurrent rollout
//    launch ~5 us whatever its size -> the pass over iteration i's trajectories rides along in
//    the rollout launch of iteration i + 1 as extra workgroups (SideJob); the last one owed is
//    flushed by sw_ars_pipeline_sync_cov.
struct sw_ars_pipeline {
    hipStream_t copy = nullptr;
    hipEvent_t h2d_done[SW_PIPELINE_SLOTS] = {};
    bool h2d_valid[SW_PIPELINE_SLOTS] = {};
    uint32_t *flag_host = nullptr, *flag_dev = nullptr;   // progress flag (pinned, mapped)
    uint32_t launches = 0;                                 // rollout launches issued so far
    hipStream_t last_main = nullptr;                       // the stream of the launches so far ...
    bool main_seen = false;                                // ... (may be the null stream)
    int timing = 0;                                        // 0 off, k: time every k-th launch
    int64_t timing_launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed;  // around the sampled rollout launches
    // covariance pass owed: the trajectories of the latest rollout launch
    const double *cov_traj = nullptr;
    double *cov_acc = nullptr;
    int64_t cov_rolls = 0;
    int32_t cov_H = 0;
    int cov_block = 0;                                     // workgroup size the pass is run with
    sw_params cov_params = {};
};

namespace {

__global__ void flag_kernel(uint32_t *flag, uint32_t value)
{
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Spin until the GPU has STARTED rollout launch number `need` (0-based) of this pipeline.
int wait_flag(const sw_ars_pipeline *pl, uint32_t need)
{
    const volatile uint32_t *f = pl->flag_host;
    // Health check of the stream (instead of spinning forever behind a faulted launch) only after 20 ms
    // without progress, then every 20 ms: hipStreamQuery is NOT free for the device -- to learn whether
    // the last kernel is done the runtime enqueues a marker (a barrier packet with a system-scope release)
    // behind it, and the next rollout launch then starts 5.9 us late.  Queried every ~50 us of spinning,
    // as until round 3, that was one marker per iteration: 233.1 -> 227 us per iteration at the headline
    // config (profiles/r03_u_gap_probe.log).
    auto last = std::chrono::steady_clock::now();
    for (int64_t spins = 0;; ++spins) {
        if ((int32_t)(*f - need) >= 0) return SW_OK;
        if ((spins & 0xfff) == 0xfff) {
            const auto now = std::chrono::steady_clock::now();
            if (now - last >= std::chrono::milliseconds(20)) {
                last = now;
                const hipError_t q = hipStreamQuery(pl->last_main);
                if (q == hipSuccess) return ((int32_t)(*f - need) >= 0) ? SW_OK : SW_ERR_LAUNCH;
                if (q != hipErrorNotReady) return SW_ERR_LAUNCH;
            }
        }
        __builtin_ia32_pause();
    }
}

int flush_owed_cov(sw_ars_pipeline *pl, hipStream_t stream)
{
    if (!pl->cov_traj) return SW_OK;
    const int rc = launch_traj_moments(&pl->cov_params, pl->cov_rolls, pl->cov_H, pl->cov_traj,
                                       pl->cov_acc, pl->cov_block, true, stream);
    pl->cov_traj = nullptr;
    return rc;
}

}  // namespace

int sw_ars_pipeline_create(sw_ars_pipeline **out)
{
    if (!out) return SW_ERR_NULL;
    sw_ars_pipeline *pl = new (std::nothrow) sw_ars_pipeline();
    if (!pl) return SW_ERR_LAUNCH;
    const unsigned evf = hipEventDisableTiming;
    // The copy stream is created with the HIGHEST priority the device offers: streams of one
    // priority share a small pool of hardware queues, and a copy stream that lands on the hardware
    // queue of the caller's (normal-priority) stream has its 64 KB H2D queued BEHIND the rollout
    // kernel it is meant to run ahead of -- the host then waits a whole kernel for every copy
    // (measured: the 4th pipeline of a process ran its iterations in 2.15 ms instead of 0.80,
    // profiles/r03_a_outlier_probe.log).  Another priority class is another queue pool.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    bool ok = hipStreamCreateWithPriority(&pl->copy, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
              hipHostMalloc((void **)&pl->flag_host, 64, hipHostMallocMapped | hipHostMallocCoherent) ==
                  hipSuccess;
    if (ok) {
        *pl->flag_host = 0u;
        ok = hipHostGetDevicePointer((void **)&pl->flag_dev, pl->flag_host, 0) == hipSuccess;
    }
    for (int i = 0; i < SW_PIPELINE_SLOTS && ok; ++i)
        ok = hipEventCreateWithFlags(&pl->h2d_done[i], evf) == hipSuccess;
    if (!ok) {
        sw_ars_pipeline_destroy(pl);
        return SW_ERR_LAUNCH;
    }
    *out = pl;
    return SW_OK;
}

void sw_ars_pipeline_destroy(sw_ars_pipeline *pl)
{
    if (!pl) return;
    if (pl->copy) (void)hipStreamSynchronize(pl->copy);
    if (pl->main_seen) (void)hipStreamSynchronize(pl->last_main);   // kernels still write the flag
    for (auto &e : pl->timed) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (int i = 0; i < SW_PIPELINE_SLOTS; ++i)
        if (pl->h2d_done[i]) (void)hipEventDestroy(pl->h2d_done[i]);
    if (pl->copy) (void)hipStreamDestroy(pl->copy);
    if (pl->flag_host) (void)hipHostFree(pl->flag_host);
    delete pl;
}

int sw_ars_pipeline_slots(void) { return SW_PIPELINE_SLOTS; }

// The host may refill deltas_host[slot] once the H2D copy that last read it has completed.
int sw_ars_pipeline_host_slot_wait(sw_ars_pipeline *pl, int slot)
{
    if (!pl) return SW_ERR_NULL;
    if (slot < 0 || slot >= SW_PIPELINE_SLOTS) return SW_ERR_SIZE;
    if (pl->h2d_valid[slot] && hipEventSynchronize(pl->h2d_done[slot]) != hipSuccess)
        return SW_ERR_LAUNCH;
    return SW_OK;
}

// Everything the covariance accumulators are owed is in them when this returns.
int sw_ars_pipeline_sync_cov(sw_ars_pipeline *pl)
{
    if (!pl) return SW_ERR_NULL;
    if (!pl->main_seen) return SW_OK;
    const int rc = flush_owed_cov(pl, pl->last_main);
    if (rc) return rc;
    return hipStreamSynchronize(pl->last_main) == hipSuccess ? SW_OK : SW_ERR_LAUNCH;
}

int sw_ars_pipeline_timing(sw_ars_pipeline *pl, int enable)
{
    if (!pl) return SW_ERR_NULL;
    pl->timing = enable > 0 ? enable : 0;
    pl->timing_launches = 0;
    if (enable) {
        for (auto &e : pl->timed) {
            (void)hipEventDestroy(e.first);
            (void)hipEventDestroy(e.second);
        }
        pl->timed.clear();
    }
    return SW_OK;
}

int sw_ars_pipeline_rollout_ms(sw_ars_pipeline *pl, double *mean_ms, int64_t *launches)
{
    if (!pl || !mean_ms || !launches) return SW_ERR_NULL;
    double tot = 0.0;
    for (auto &e : pl->timed) {
        if (hipEventSynchronize(e.second) != hipSuccess) return SW_ERR_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.first, e.second) != hipSuccess) return SW_ERR_LAUNCH;
        tot += ms;
    }
    *launches = (int64_t)pl->timed.size();
    *mean_ms = pl->timed.empty() ? 0.0 : tot / (double)pl->timed.size();
    return SW_OK;
}

int sw_ars_pipeline_next_slot(sw_ars_pipeline *pl)
{
    return pl ? (int)(pl->launches % (uint32_t)SW_PIPELINE_SLOTS) : -1;
}

int sw_ars_iteration_rollouts_f64(sw_ars_pipeline *pl, int slot, const sw_params *p,
                                  int64_t n_dir_total, int64_t dir_begin, int64_t n_dir, int32_t H,
                                  const double *deltas_host, double *deltas_dev,
                                  const double *policy, double nu, const double *mean,
                                  const double *inv_std, double *returns, double *traj,
                                  double *moments, double *cov_acc, int32_t *status, void *stream)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (!pl || !deltas_host || !deltas_dev) return SW_ERR_NULL;
    if (n_dir < 0 || dir_begin < 0 || H < 0 || n_dir_total < dir_begin + n_dir) return SW_ERR_SIZE;
    // The slot is a function of the pipeline's own call count, never of the caller's bookkeeping:
    // call k uses slot k mod SLOTS, and every call counts -- also a rank's call with an empty
    // shard (world > N), which launches nothing but the progress flag.
    const uint32_t k = pl->launches;
    if (slot != (int)(k % (uint32_t)SW_PIPELINE_SLOTS)) return SW_ERR_SIZE;
    if (cov_acc && !traj && n_dir > 0) return SW_ERR_NULL;
    hipStream_t main = (hipStream_t)stream;
    if (pl->main_seen && pl->last_main != main) {
        // the progress flag orders work on ONE stream; a new stream starts from a clean slate
        if (hipStreamSynchronize(pl->last_main) != hipSuccess) return SW_ERR_LAUNCH;
    }
    pl->last_main = main;
    pl->main_seen = true;
    const size_t bytes = (size_t)n_dir_total * (size_t)((p->n - 1) * (2 * p->n + 2)) * sizeof(double);
    // Call k reuses the buffers of call k - SLOTS: its device deltas were last read by update
    // k - SLOTS (done once launch k - SLOTS + 1 has started) and its trajectories by the
    // covariance workgroups of launch k - SLOTS + 1 (done once launch k - SLOTS + 2 has started).
    if (k >= (uint32_t)SW_PIPELINE_SLOTS) {
        rc = wait_flag(pl, k + 2u - (uint32_t)SW_PIPELINE_SLOTS);
        if (rc) return rc;
    }
    if (hipMemcpyAsync(deltas_dev, deltas_host, bytes, hipMemcpyHostToDevice, pl->copy) != hipSuccess)
        return SW_ERR_LAUNCH;
    if (hipEventRecord(pl->h2d_done[slot], pl->copy) != hipSuccess) return SW_ERR_LAUNCH;
    pl->h2d_valid[slot] = true;
    // host-confirmed: the deltas have landed -> the rollout launch needs no device-side wait
    if (hipEventSynchronize(pl->h2d_done[slot]) != hipSuccess) return SW_ERR_LAUNCH;
    if (n_dir == 0) {
        // empty shard: keep the flag sequence going (the host paces the ring on it)
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, main, pl->flag_dev, k);
        rc = launch_status();
        if (rc) return rc;
        pl->launches = k + 1u;
        return SW_OK;
    }
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    const bool timed_launch = pl->timing > 0 && (pl->timing_launches++ % pl->timing) == 0;
    if (timed_launch) {
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess ||
            hipEventRecord(ev.first, main) != hipSuccess)
            return SW_ERR_LAUNCH;
    }
    const bool oct = use_oct3(p, 2 * n_dir, H, traj != nullptr);
    const bool quad = use_quad3(p, 2 * n_dir, H, traj != nullptr);   // (true for oct launches too)
    const bool row = !quad && use_row(p, 2 * n_dir, H, traj != nullptr);
    const int block = oct ? kOctBlock : (quad ? kRollBlock : kMomBlock);   // kRowBlock == kMomBlock
    SideJob sj = kNoSide;
    sj.flag = pl->flag_dev;
    sj.flag_value = k;
    // the owed pass rides along when this launch's kernel can carry it in the tiling it is owed in
    const bool ride = pl->cov_traj && (quad || row) && pl->cov_params.n == p->n &&
                      pl->cov_block == block;
    if (pl->cov_traj && !ride) {
        rc = flush_owed_cov(pl, main);
        if (rc) return rc;
    }
    if (ride) {
        sj.cov_traj = pl->cov_traj;
        sj.cov_acc = pl->cov_acc;
        sj.cov_rolls = pl->cov_rolls;
        sj.cov_H = pl->cov_H;
    }
    bool taken = false;
    if (!quad && !row) {
        // the lane kernel takes no side job: the flag as a launch of its own in front of it
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, main, pl->flag_dev, k);
        rc = launch_status();   // a failed flag launch must surface here, not iterations later in wait_flag
        if (rc) return rc;
        sj = kNoSide;
    }
    rc = launch_ars_rollouts(p, dir_begin, n_dir, H, policy, deltas_dev, nu, mean, inv_std,
                             returns, traj, moments, status, stream, &sj, &taken);
    if (timed_launch) {
        (void)hipEventRecord(ev.second, main);
        pl->timed.push_back(ev);
    }
    if (rc) return rc;
    if (ride) pl->cov_traj = nullptr;   // rode along in this launch
    pl->launches = k + 1u;
    // this launch's trajectories are owed a covariance pass: the next launch carries it
    if (cov_acc) {
        pl->cov_traj = traj;
        pl->cov_acc = cov_acc;
        pl->cov_rolls = 2 * n_dir;
        pl->cov_H = H;
        pl->cov_block = block;
        pl->cov_params = *p;
    }
    return SW_OK;
}

int sw_ars_iteration_update_f64(sw_ars_pipeline *pl, int slot, const sw_params *p, int64_t n_dir,
                                const double *gathered, int32_t world, int64_t chunk,
                                int64_t rows_chunk, const double *deltas_dev, double *policy,
                                double alpha, double b, int64_t top_b, double *running,
                                int64_t n_new_states, double *mean, double *inv_std,
                                double *sigma_out, void *stream)
{
    if (!pl) return SW_ERR_NULL;
    if (slot < 0 || slot >= SW_PIPELINE_SLOTS) return SW_ERR_SIZE;
    return sw_ars_update_gathered_f64(p, n_dir, gathered, world, chunk, rows_chunk, deltas_dev,
                                      policy, alpha, b, top_b, running, n_new_states, mean,
                                      inv_std, sigma_out, stream);
}

}  // extern "C"
