/*
 * swimmer_hip.h -- C ABI of libswimmer_hip.so, the MI355X (gfx950) implementation of the
 * reference's swimmer-physics + ARS-rollout hot path.
 *
 * Every entry point is `extern "C"`, takes plain device pointers and sizes, allocates
 * nothing, keeps no global state, never throws, and is asynchronous on the HIP stream it
 * is given (`stream` is a hipStream_t passed as void*; NULL = the default stream).  All
 * arithmetic is IEEE fp64.  All pointers are DEVICE pointers unless marked "host".
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   sw_step_f64          SwimmerEnv.step -> next_observation -> compute_accelerations ->
 *                        solve, + get_reward     envs/gym_swimmer/swimmer/remy_swimmer_env.py:41-56,
 *                                                :69-93, :95-214, :238-243
 *   sw_accel_f64         SwimmerEnv.compute_accelerations           remy_swimmer_env.py:95-114
 *   sw_reset_f64         SwimmerEnv.reset                           remy_swimmer_env.py:58-67
 *   sw_rollout_f64       Environment.select_action + .rollout       ars/environment.py:19-57
 *   sw_ars_rollouts_f64  ARSAgent.runOneIteration's perturb + 2N rollouts loop
 *                                                                   ars/ars_agent.py:137-172
 *   sw_ars_update_f64    ARSAgent.sort_directions / update_policy and the V2 statistics
 *                                                                   ars/ars_agent.py:97-130, :176-182
 *   sw_traj_moments_f64  np.mean / np.cov over the saved states     ars/ars_agent.py:180-182
 *   sw_env1_step         the same step for ONE swimmer handed over in host memory (the Gym
 *                        surface and the RL-Glue env_step, SwimmerEnvironment.cpp:53-68)
 *   sw_step_residual_f64 Estimator.I: every stored transition re-simulated and compared with its stored next
 *                        state                                       ars/estimator.py:36-62
 *   sw_safe_rollouts_f64 Safe_ARS.isSafe + Safe_ARS.rollout (the one-step simulator look-ahead that gates
 *                        every real step)                            safe_ars/ars.py:111-153
 *
 * Layouts (d = 2n+2 observation size, m = n-1 action size):
 *   state, SoA    [d][n_env]   field-major: row f holds field f of every env; fields are
 *                              the reference's observation order [Gdx, Gdy, th1, thd1, ...,
 *                              thn, thdn] (remy_swimmer_env.py:216-224).  Coalesced: lane e
 *                              of a wave reads element e of each row.
 *   action, SoA   [m][n_env]
 *   policies, AoS [n_roll][m][d]   exactly numpy's np.array(list_of_(m,d)_matrices)
 *   deltas,  AoS  [n_dir][m][d]    ars_agent.py:137-138
 *   traj          [H][d][n_roll]   step-major, then field, then rollout (coalesced stores);
 *                                  the reference's trajectories[r][t][f] is traj[t][f][r]
 *   returns       [n_roll];  for the ARS entry point rollout 2i is P+nu*delta_i and 2i+1 is
 *                            P-nu*delta_i, the reference's `rewards` order (ars_agent.py:161-169)
 */
#ifndef SWIMMER_HIP_H
#define SWIMMER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SW_ABI_VERSION 3
#define SW_MAX_SEGMENTS 8 /* kernels are instantiated for n = 2..8 */

/* status codes (return values) */
#define SW_OK 0
#define SW_ERR_NULL 1        /* a required pointer is NULL */
#define SW_ERR_SEGMENTS 2    /* n outside 2..SW_MAX_SEGMENTS */
#define SW_ERR_SIZE 3        /* negative / zero size where not allowed */
#define SW_ERR_PARAM 4       /* non-finite or non-positive l_i / m_i, non-finite k / h */
#define SW_ERR_LAUNCH 5      /* hipLaunch failed; see hipGetLastError on the caller side */

/* per-env / per-rollout status bits written to the optional `status` arrays */
#define SW_STATUS_OK 0
#define SW_STATUS_SINGULAR 1 /* a pivot of the joint-acceleration system was <= 0 or not
                                finite: numpy.linalg.solve would raise LinAlgError
                                (remy_swimmer_env.py:212) */
#define SW_STATUS_NONFINITE 2 /* the new state contains inf / nan */
#define SW_STATUS_RANGE 4     /* an angle reached |theta| >= 3e9 rad, outside the range of the
                                 in-kernel sin/cos: outputs are NaN (a simulation that far gone
                                 has ulp(theta) > 4e-7 rad and no meaning left) */

/* sw_params.flags: force one of the two rollout kernels (default: chosen from n and n_roll).
 * Both compute the same rollouts; they differ in summation order only (a few ulp per step). */
#define SW_FLAG_ROLLOUT_LANE 1 /* one rollout per lane (throughput form, any n) */
#define SW_FLAG_ROLLOUT_QUAD 2 /* one segment per lane (latency form): a DPP quad per rollout for
                                  n = 3, a 16-lane DPP row per rollout for n = 4..8; n = 2 ignores it */

/* sw_params.flags: which of the reference's two swimmer models the kernels integrate.
 * Default (bit clear): the Gym env (envs/gym_swimmer/swimmer/remy_swimmer_env.py, explicit
 * Euler, reset = (0, 0, pi/2, 0, ...)).  Bit set: the native RL-Glue environment
 * (rlglue/environment/SwimmerEnvironment.cpp:102-277: its (5n+2)-unknown formulation with
 * its quirks, semi-implicit Euler, start state = all 0.001) -- a different numerical model
 * (they give different accelerations for the same state).  Rollouts of the twin model run on
 * the lane-per-rollout kernel. */
#define SW_FLAG_MODEL_TWIN 4

/* Physical parameters of one swimmer model: SwimmerEnv.__init__ (remy_swimmer_env.py:16-39).
 * max_u is not here: the reference never enforces it (actions are not clipped). */
typedef struct sw_params {
    int32_t n;        /* segments */
    int32_t flags;    /* 0, or SW_FLAG_* bits (rollout kernel choice, model choice) */
    double l_i;       /* segment length */
    double m_i;       /* segment mass */
    double k;         /* viscous friction coefficient */
    double h;         /* explicit-Euler time step */
    double dir_x;     /* reward = Gdot_new . direction */
    double dir_y;
} sw_params;

int sw_abi_version(void);
const char *sw_strerror(int code);
int sw_max_segments(void);

/* state[f][e] <- reset state (Gdot = 0, theta = pi/2, thetadot = 0). */
int sw_reset_f64(const sw_params *p, int64_t n_env, double *state, void *stream);

/* One physics step for n_env independent swimmers.  state_out may alias state_in.
 * reward and status may be NULL. */
int sw_step_f64(const sw_params *p, int64_t n_env, const double *state_in,
                const double *action, double *state_out, double *reward,
                int32_t *status, void *stream);

/* Accelerations only: gdd [2][n_env], tdd [n][n_env]. */
/* The estimator's objective in one pass (ars/estimator.py:36-62): for every stored transition (state, action,
 * stored next state; SoA like sw_step_f64) the Euclidean distance || step(state, action) - next_ref ||_2, summed in a
 * fixed order per workgroup: partial[b] = sum over transitions [b * B, (b + 1) * B), B = n_env / sw_step_residual_blocks
 * rounded up to the workgroup size (256).  I(x) is the sum of `partial` (sw_step_residual_blocks(n_env) doubles).  Nothing
 * else is written: the simulated next states never reach memory.  Gym model only (SW_FLAG_MODEL_TWIN: SW_ERR_PARAM). */
int sw_step_residual_f64(const sw_params *p, int64_t n_env, const double *state, const double *action,
                         const double *next_ref, double *partial, void *stream);
int64_t sw_step_residual_blocks(int64_t n_env);

int sw_accel_f64(const sw_params *p, int64_t n_env, const double *state,
                 const double *action, double *gdd, double *tdd, void *stream);

/* n_roll independent H-step rollouts of a linear policy, one policy per rollout.
 *   mean, inv_std : [d] each, both NULL -> ARS V1 action a = P s; both given -> V2 action
 *                   a = (P diag(inv_std)) (s - mean), inv_std = diag(cov) ** -0.5
 *   state0        : [d][n_roll] start states, NULL -> reset state
 *   returns       : [n_roll] sum of the H rewards
 *   traj          : NULL or [H][d][n_roll], every post-step state
 *   final_state   : NULL or [d][n_roll]
 *   moments       : NULL or [sw_moments_blocks(n_roll)][2d] partial sums, one row per 16
 *                   consecutive rollouts, of sum(s - c) and sum((s - c)^2) over all their
 *                   post-step states, c = reset state
 *   status        : NULL or [n_roll] */
int sw_rollout_f64(const sw_params *p, int64_t n_roll, int32_t H, const double *policies,
                   const double *mean, const double *inv_std, const double *state0,
                   double *returns, double *traj, double *final_state, double *moments,
                   int32_t *status, void *stream);

/* Safe exploration, safe_ars/ars.py Safe_ARS.rollout (:124-153): n_roll rollouts of H steps from the reset state,
 * every real step gated by a one-step look-ahead in a simulator -- isSafe (:111-122) = one step of the swimmer with
 * the parameters `sim` from the real state under the proposed action policy @ obs (:139), and
 * cost(simulated observation) <= sim_thresh.  A refused step leaves the state where it is (:150-151; the rollout
 * then stays refused: it proposes the same action again).  `real` and `sim` must have the same number of segments;
 * the model flag of `real` / `sim` is ignored (Gym model).  The whole loop is ONE launch: n = 3 up to 8192 rollouts in the
 * mirror-quad form (the look-ahead and the real step share the step's geometry), n = 4..8 in the row form while SIMDs are
 * idle, one rollout per lane otherwise (or with SW_FLAG_ROLLOUT_LANE in real->flags); same results to rounding.
 *   policies      : [n_roll][m][d]
 *   cost_kind     : SW_COST_ABS_OBS            cost = |obs[cost_index]|, obs = [Gdx, Gdy, th_1, thd_1, ...]
 *                   SW_COST_MAX_ABS_THETADOT   cost = max_i |thetadot_i|  (safe_ars/experiment.py:45; cost_index unused)
 *   returns       : [n_roll] sum of the rewards of the steps taken
 *   traj          : NULL or [H][d][n_roll]: the state after step t, the unchanged state where step t was refused
 *   first_refused : NULL or [n_roll]: the first refused step (H: none)
 *   violations    : NULL or [n_roll]: real steps whose cost exceeded real_thresh (the reference prints each, :143-144)
 *   status        : NULL or [n_roll] */
#define SW_COST_ABS_OBS 0
#define SW_COST_MAX_ABS_THETADOT 1
int sw_safe_rollouts_f64(const sw_params *real, const sw_params *sim, int64_t n_roll, int32_t H,
                         const double *policies, int32_t cost_kind, int32_t cost_index, double sim_thresh,
                         double real_thresh, double *returns, double *traj, int32_t *first_refused,
                         int32_t *violations, int32_t *status, void *stream);

/* Number of partial-moment rows sw_rollout_f64 / sw_ars_rollouts_f64 write for n_roll
 * rollouts (= ceil(n_roll / 16), whichever kernel runs). */
int64_t sw_moments_blocks(int64_t n_roll);

/* The ARS exploration batch: for directions i in [dir_begin, dir_begin + n_dir) run the two
 * rollouts P + nu*delta_i and P - nu*delta_i (perturbation fused into the kernel prologue).
 *   policy  : [m][d]          deltas : [>= dir_begin + n_dir][m][d]
 *   returns : [2 * n_dir] local slice, entry 2j / 2j+1 = +/- rollout of direction dir_begin+j
 *   traj    : NULL or [H][d][2 * n_dir];  moments as in sw_rollout_f64 with n_roll = 2 n_dir */
int sw_ars_rollouts_f64(const sw_params *p, int64_t dir_begin, int64_t n_dir, int32_t H,
                        const double *policy, const double *deltas, double nu,
                        const double *mean, const double *inv_std, double *returns,
                        double *traj, double *moments, int32_t *status, void *stream);

/* ARS policy update + V2 statistics.
 *   returns       : [2 * n_dir] all returns of the iteration (after the all-gather)
 *   policy        : [m][d], updated in place:
 *                   P += alpha / (div * sigma_R) * sum_{i in used}(r_i+ - r_i-) delta_i
 *                   sigma_R = population std (ddof = 0) of the used returns.
 *   top_b         : 0 -> ars/ars_agent.py's behaviour: every direction is used, b is only a
 *                   divisor (ars_agent.py:176-177, :126-128);  > 0 -> safe_ars/ars.py's
 *                   Basic_ARS: only the top_b directions by max(r+, r-) are used (:95-96),
 *                   sigma_R is taken over their returns (:60) and the divisor is the NUMBER OF
 *                   DIRECTIONS USED, len(order) = min(top_b, n_dir) (:64) -- `b` is ignored
 *   moments       : NULL (V1) or [n_moment_rows][2d] partial sums of this iteration
 *   running       : NULL (V1) or [1 + 2d] statistics of every state since training began
 *                   (ars_agent.py:171, :180: never cleared): {count n, mean - c (d),
 *                   M2 = sum (s - mean)^2 (d)}, c = reset state; zero before the first call.
 *                   Each iteration's batch is merged in (Chan et al. pairwise update): no
 *                   cancellation that grows with the length of training
 *   n_new_states  : states added this iteration (2 * n_dir * H over all ranks)
 *   mean, inv_std : NULL (V1) or [d] each, overwritten with the new mean and
 *                   var ** -0.5 (var with ddof = 1, np.cov's default)
 *   sigma_out     : NULL or [1] */
int sw_ars_update_f64(const sw_params *p, int64_t n_dir, const double *returns,
                      const double *deltas, double *policy, double alpha, double b,
                      int64_t top_b, const double *moments, int64_t n_moment_rows,
                      double *running, int64_t n_new_states, double *mean, double *inv_std,
                      double *sigma_out, void *stream);

/* The same update reading an all-gathered buffer in place (no repacking between the
 * collective and the update):  gathered = `world` segments of
 * seg_len = 2*chunk + rows_chunk*2d doubles, segment r =
 * [2*chunk returns of directions r*chunk .. | rows_chunk moment rows]  (zero padded).
 * world = 1 is a single rank's own segment. */
int sw_ars_update_gathered_f64(const sw_params *p, int64_t n_dir, const double *gathered,
                               int32_t world, int64_t chunk, int64_t rows_chunk,
                               const double *deltas, double *policy, double alpha, double b,
                               int64_t top_b, double *running, int64_t n_new_states,
                               double *mean, double *inv_std, double *sigma_out, void *stream);

/* Full first and second moments of recorded trajectories (for the `covariance` attribute):
 * acc[0] += count, acc[1..d] += sum(s - c), acc[1+d + f*d + g] += sum((s-c)_f (s-c)_g),
 * over traj [H][d][n_roll].  HBM-bound: reads the trajectory buffer exactly once.
 * acc holds sw_cov_acc_doubles(p, n_roll, H) doubles: the 1 + d + d*d sums, then the pass's
 * scratch (a ticket counter and the tiles' partial sums, laid out [entry][tile]); the caller zeroes
 * ALL of it before the first call and leaves the scratch part alone afterwards.  No floating-point
 * atomics: the tile that finishes last merges the partial sums in an order that depends on the
 * number of tiles only, so the same call on the same data gives the same bits.  Passes over one acc
 * must be stream-ordered (one at a time). */
int64_t sw_cov_acc_doubles(const sw_params *p, int64_t n_roll, int32_t H);
int sw_traj_moments_f64(const sw_params *p, int64_t n_roll, int32_t H, const double *traj,
                        double *acc, void *stream);


/* ---- one swimmer, one step per call: the batch-1 drop-in surfaces -------------------------
 * SwimmerEnv.step / next_observation (remy_swimmer_env.py:41-56, :69-93) and the native twin's
 * env_step (rlglue/environment/SwimmerEnvironment.cpp:53-68) hand over ONE state and ONE action
 * and need the next state before they return.  A handle owns a small pinned, device-mapped
 * I/O block and a stream of its own: the caller writes state and action into the block,
 * sw_env1_step launches ONE kernel that reads them over the bus, writes next state, reward and
 * status back into the block and then a sequence number the host spins on -- one launch and
 * one host wait per step; no allocation, no memcpy call, no stream synchronisation.
 * Offsets into the block, in doubles (sized for SW_MAX_SEGMENTS): */
#define SW_ENV1_STATE 0    /* in : [d]  observation order [Gdx, Gdy, th1, thd1, ...] */
#define SW_ENV1_ACTION 18  /* in : [m] */
#define SW_ENV1_NEXT 32    /* out: [d]  (sw_env1_step) */
#define SW_ENV1_REWARD 50  /* out: [1]  (sw_env1_step) */
#define SW_ENV1_GDD 52     /* out: [2]  (sw_env1_accel) */
#define SW_ENV1_TDD 54     /* out: [n]  (sw_env1_accel) */
#define SW_ENV1_DOUBLES 64
typedef struct sw_env1 sw_env1;
int sw_env1_create(sw_env1 **out);
void sw_env1_destroy(sw_env1 *e);
/* A handle is used by one thread at a time (it owns one I/O block and one sequence counter) and
 * belongs to the device that was current when it was created: sw_env1_step / sw_env1_accel launch on that
 * device whatever the caller's current device is, and leave the caller's current device unchanged.
 * HOST pointer to the handle's I/O block (SW_ENV1_DOUBLES doubles), valid until destroy. */
double *sw_env1_io(sw_env1 *e);
/* One physics step of the swimmer in the block (model chosen by p->flags); BLOCKING: the
 * outputs are in the block when it returns.  *status (host, may be NULL) receives SW_STATUS_*. */
int sw_env1_step(sw_env1 *e, const sw_params *p, int32_t *status);
/* compute_accelerations of the state / action in the block (remy_swimmer_env.py:95-114). */
int sw_env1_accel(sw_env1 *e, const sw_params *p);

/* ---- the exchange step of the sharded ARS iteration, straight into RCCL -----------------
 * One all-gather of every rank's packed result segment per iteration replaces the serial loop
 * over directions (ars/ars_agent.py:160 "TODO ... PARALLEL"); see sw_ars_update_gathered_f64 for
 * the layout.  These entry points issue it from native code ON THE CALLER'S STREAM (the critical
 * stream rollouts -> all-gather -> update), without torch.distributed in between.  RCCL is
 * resolved at run time (dlopen by soname: the copy a torch process has loaded already), so the
 * library has no link-time dependency on it; sw_comm_available() says whether it was found.
 * Rank 0 draws the id, the caller distributes its SW_COMM_ID_BYTES bytes by whatever means it has
 * (torch.distributed broadcast, MPI, a file), every rank creates its communicator (collective). */
#define SW_COMM_ID_BYTES 128
/* sw_comm_create is COLLECTIVE over the ranks and binds the communicator to the CURRENT device
 * (hipSetDevice first); one communicator per rank, one GPU per rank. */
typedef struct sw_comm sw_comm;
int sw_comm_available(void);
int sw_comm_unique_id(uint8_t *id /* host, SW_COMM_ID_BYTES */);
int sw_comm_create(sw_comm **out, const uint8_t *id, int32_t world, int32_t rank);
void sw_comm_destroy(sw_comm *c);
/* recv[r * count .. (r+1) * count) <- rank r's send[0 .. count)  (device pointers). */
int sw_comm_all_gather_f64(sw_comm *c, const double *send, double *recv, int64_t count, void *stream);
const char *sw_comm_last_error(sw_comm *c);

/* ---- measurement aid ---------------------------------------------------------------------
 * One wave issuing trips x 64 independent instructions of one class (mode 0: v_fma_f64,
 * mode 1: v_mov_b32); scratch64: 64 doubles.  bench.py times it with HIP events to calibrate the
 * ceiling of the latency-bound rollout kernels (one instruction per ~2 ns for a lone wave) on the
 * device it runs on.  Replaces nothing in the reference. */
int sw_issue_probe(int32_t mode, int32_t trips, double *scratch64, void *stream);
/* The same with `workgroups` x `waves_per_workgroup` (1..4) such waves at once: 256 x 4 puts one wave on
 * every SIMD of an MI355X, i.e. the intervals with the whole chip issuing (a chip full of f64 work sustains
 * a lower clock than a lone wave sees: bench.py prices the full-chip launches with these). */
int sw_issue_probe_grid(int32_t mode, int32_t trips, int32_t workgroups, int32_t waves_per_workgroup,
                        double *scratch64, void *stream);

/* ---- host helper: the reference's random stream ----------------------------------------
 * out[i] = 2*u_i - 1 with u_i the next doubles of NumPy's legacy MT19937 generator
 * (np.random.rand), continuing from the state (key[624], *pos) in the form
 * np.random.get_state() / set_state() use; the state is advanced in place.  Replaces the
 * N calls of 2*np.random.rand(m, d)-1 in ars/ars_agent.py:137-138 (same values, an order of
 * magnitude faster, so the host keeps ahead of the GPU at any rank count).  HOST pointers.
 * The vector width (baseline x86-64 / AVX2 / AVX-512) is chosen at run time from the CPU's features;
 * every width produces the same bits. */
int sw_mt19937_uniform_pm1(uint32_t *key, int32_t *pos, int64_t n, double *out);
/* Test hook: run the stream at a given width (0 baseline, 1 AVX2, 2 AVX-512; -1 = widest available, the
 * default); a width the CPU lacks falls back to the next one down.  Returns the width that will run. */
int sw_mt19937_force_isa(int level);

/* ---- ARS iteration pipeline (host-side enqueue logic in native code) ------------------
 * Replaces the serial body of ARSAgent.runOneIteration (ars/ars_agent.py:137-182) with a
 * schedule over a ring of SW_PIPELINE_SLOTS buffer sets.  A pipeline owns one extra HIP stream
 * (H2D copies of the deltas) and 64 bytes of pinned host memory (a progress flag); it owns no
 * device memory: every buffer is passed per call, one set per `slot`.  The slot of a call is
 * sw_ars_pipeline_next_slot() = (number of sw_ars_iteration_rollouts_f64 calls so far) mod
 * SW_PIPELINE_SLOTS -- the pipeline's own count, so that a caller's iteration counter (reset by
 * a checkpoint load, say) can never shift the ring; a call with another slot is refused with
 * SW_ERR_SIZE.  Every rank calls once per iteration, also with an empty shard (n_dir = 0: nothing
 * but the progress flag is launched).  Use the pipeline with ONE stream.  The caller's stream carries kernels
 * only -- no cross-stream waits, no event records: every rollout launch stores its index to the
 * progress flag when it starts, and the host paces buffer reuse on that.
 *
 *   sw_ars_iteration_rollouts_f64   copy stream: deltas_host (pinned) -> deltas_dev
 *                                   caller's stream: ONE launch = the 2*n_dir rollouts of this
 *                                   rank's shard + (extra workgroups) the covariance pass
 *                                   sw_traj_moments_f64 over the PREVIOUS call's traj -> its
 *                                   cov_acc; this call's traj is owed a pass (if cov_acc given;
 *                                   cov_acc: sw_cov_acc_doubles(p, 2*n_dir, H) doubles, zeroed)
 *   ... caller all-gathers its segment [returns | moment rows] between ranks ...
 *   sw_ars_iteration_update_f64     caller's stream: sw_ars_update_gathered_f64 on the gathered
 *                                   buffer
 *
 * Before refilling deltas_host of a slot the host calls sw_ars_pipeline_host_slot_wait;
 * before reading cov_acc it calls sw_ars_pipeline_sync_cov (runs the pass still owed and
 * synchronises the stream). */
#define SW_PIPELINE_SLOTS 4
typedef struct sw_ars_pipeline sw_ars_pipeline;

int sw_ars_pipeline_create(sw_ars_pipeline **out);
void sw_ars_pipeline_destroy(sw_ars_pipeline *pl);
int sw_ars_pipeline_slots(void);
int sw_ars_pipeline_next_slot(sw_ars_pipeline *pl);
int sw_ars_pipeline_host_slot_wait(sw_ars_pipeline *pl, int slot);
int sw_ars_pipeline_sync_cov(sw_ars_pipeline *pl);
/* enable = k > 0: record HIP events around every k-th rollout launch on its stream (resets the
 * log; each timed launch costs ~10 us of pipeline bubbles, so sample sparsely); 0: off */
int sw_ars_pipeline_timing(sw_ars_pipeline *pl, int enable);
int sw_ars_pipeline_rollout_ms(sw_ars_pipeline *pl, double *mean_ms, int64_t *launches);

int sw_ars_iteration_rollouts_f64(sw_ars_pipeline *pl, int slot, const sw_params *p,
                                  int64_t n_dir_total, int64_t dir_begin, int64_t n_dir,
                                  int32_t H, const double *deltas_host /* pinned host */,
                                  double *deltas_dev, const double *policy, double nu,
                                  const double *mean, const double *inv_std, double *returns,
                                  double *traj, double *moments, double *cov_acc,
                                  int32_t *status, void *stream);

int sw_ars_iteration_update_f64(sw_ars_pipeline *pl, int slot, const sw_params *p,
                                int64_t n_dir, const double *gathered, int32_t world,
                                int64_t chunk, int64_t rows_chunk, const double *deltas_dev,
                                double *policy, double alpha, double b, int64_t top_b,
                                double *running, int64_t n_new_states, double *mean,
                                double *inv_std, double *sigma_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SWIMMER_HIP_H */
