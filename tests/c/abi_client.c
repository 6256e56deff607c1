/* A plain-C client of the C ABI (include/swimmer_hip.h): no Python, no torch -- device
 * memory from the HIP runtime, the library called exactly as a foreign binding would.
 * Prints the next state of one physics step and the return of one 100-step rollout for a
 * fixed input; tests/test_c_client.py compares the numbers with the oracle. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "swimmer_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SW(x) do { int rc_ = (x); if (rc_ != SW_OK) { fprintf(stderr, "%s -> %s\n", #x, sw_strerror(rc_)); return 3; } } while (0)

int main(void)
{
    const sw_params p = {3, 0, 0.8, 1.2, 10.2, 1e-3, 1.0, 0.0};
    const int d = 8, m = 2;
    /* one swimmer: SoA [d][1] is just the observation vector */
    const double state[8] = {0.1, -0.2, 1.0, 0.5, 2.0, -0.3, -1.0, 0.25};
    const double action[2] = {1.5, -2.5};
    double policy[16];
    for (int i = 0; i < 16; ++i) policy[i] = 0.01 * (i - 7);
    double *d_state, *d_action, *d_next, *d_reward, *d_policy, *d_ret;
    int32_t *d_status;
    CHECK_HIP(hipMalloc((void **)&d_state, sizeof state));
    CHECK_HIP(hipMalloc((void **)&d_next, sizeof state));
    CHECK_HIP(hipMalloc((void **)&d_action, sizeof action));
    CHECK_HIP(hipMalloc((void **)&d_reward, sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_policy, sizeof policy));
    CHECK_HIP(hipMalloc((void **)&d_ret, sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_status, sizeof(int32_t)));
    CHECK_HIP(hipMemcpy(d_state, state, sizeof state, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_action, action, sizeof action, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_policy, policy, sizeof policy, hipMemcpyHostToDevice));

    printf("abi %d max_segments %d\n", sw_abi_version(), sw_max_segments());
    CHECK_SW(sw_step_f64(&p, 1, d_state, d_action, d_next, d_reward, d_status, NULL));
    double next[8], reward, ret;
    int32_t status;
    CHECK_HIP(hipMemcpy(next, d_next, sizeof next, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&reward, d_reward, sizeof reward, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&status, d_status, sizeof status, hipMemcpyDeviceToHost));
    printf("step");
    for (int i = 0; i < d; ++i) printf(" %.17g", next[i]);
    printf(" reward %.17g status %d\n", reward, (int)status);
    /* one 100-step rollout of a linear policy from the state above (V1 action) */
    CHECK_SW(sw_rollout_f64(&p, 1, 100, d_policy, NULL, NULL, d_state, d_ret, NULL, NULL, NULL, d_status, NULL));
    CHECK_HIP(hipMemcpy(&ret, d_ret, sizeof ret, hipMemcpyDeviceToHost));
    printf("rollout %.17g\n", ret);
    /* argument errors come back as codes, never as crashes */
    const sw_params bad = {9, 0, 1.0, 1.0, 10.0, 1e-3, 1.0, 0.0};
    printf("bad_n %d null_ptr %d\n", sw_step_f64(&bad, 1, d_state, d_action, d_next, NULL, NULL, NULL),
           sw_step_f64(&p, 1, NULL, d_action, d_next, NULL, NULL, NULL));
    (void)m;
    return 0;
}
