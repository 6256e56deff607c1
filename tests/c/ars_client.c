/* The ARS iteration pipeline of the C ABI driven from plain C (no Python, no torch): the
 * schedule a foreign binding would run -- refill the pinned deltas of a slot, enqueue the
 * rollouts, (all-gather: one rank here), enqueue the update -- for more iterations than the
 * pipeline has buffer slots.  usage: ars_client <deltas.bin> <iters> <N> <H> <out.bin>
 * deltas.bin: iters x N x m x d doubles.  out.bin: policy (m d) | mean (d) | inv_std (d) |
 * cov_acc (1 + d + d d) | returns of the last iteration (2 N).
 * tests/test_c_client.py compares with the CPU oracle of the reference's ARS loop. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "swimmer_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_SW(x) do { int rc_ = (x); if (rc_ != SW_OK) { fprintf(stderr, "%s -> %s\n", #x, sw_strerror(rc_)); return 3; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 6) return 1;
    const int iters = atoi(argv[2]), N = atoi(argv[3]), H = atoi(argv[4]);
    const sw_params p = {3, 0, 0.8, 1.2, 10.2, 1e-3, 1.0, 0.0};   /* the "real world" of ars/plot_graph.py */
    const int d = 8, m = 2, md = m * d;
    const double alpha = 0.0075, nu = 0.01;
    const size_t per_iter = (size_t)N * md;
    double *all = malloc(sizeof(double) * per_iter * iters);
    FILE *f = fopen(argv[1], "rb");
    if (!all || !f || fread(all, sizeof(double), per_iter * iters, f) != per_iter * iters) return 1;
    fclose(f);

    const int64_t rows = sw_moments_blocks(2 * N);
    const size_t seg = 2 * (size_t)N + (size_t)rows * 2 * d, ncov = 1 + d + d * d;
    double *dh[SW_PIPELINE_SLOTS], *dd[SW_PIPELINE_SLOTS], *traj[SW_PIPELINE_SLOTS];
    for (int s = 0; s < SW_PIPELINE_SLOTS; ++s) {
        CHECK_HIP(hipHostMalloc((void **)&dh[s], sizeof(double) * per_iter, 0));
        CHECK_HIP(hipMalloc((void **)&dd[s], sizeof(double) * per_iter));
        CHECK_HIP(hipMalloc((void **)&traj[s], sizeof(double) * (size_t)H * d * 2 * N));
    }
    double *policy, *mean, *inv_std, *running, *cov_acc, *sigma, *send;
    int32_t *status;
    CHECK_HIP(hipMalloc((void **)&policy, sizeof(double) * md));
    CHECK_HIP(hipMalloc((void **)&mean, sizeof(double) * d));
    CHECK_HIP(hipMalloc((void **)&inv_std, sizeof(double) * d));
    CHECK_HIP(hipMalloc((void **)&running, sizeof(double) * (1 + 2 * d)));
    /* the covariance accumulator carries the pass's scratch behind the 1 + d + d d sums */
    const int64_t cov_len = sw_cov_acc_doubles(&p, 2 * N, H);
    if (cov_len < (int64_t)ncov) return 1;
    CHECK_HIP(hipMalloc((void **)&cov_acc, sizeof(double) * (size_t)cov_len));
    CHECK_HIP(hipMalloc((void **)&sigma, sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&send, sizeof(double) * seg));
    CHECK_HIP(hipMalloc((void **)&status, sizeof(int32_t) * 2 * N));
    CHECK_HIP(hipMemset(policy, 0, sizeof(double) * md));        /* initial_w = 'Zero' */
    CHECK_HIP(hipMemset(mean, 0, sizeof(double) * d));           /* first V2 iteration: mean 0, cov I */
    CHECK_HIP(hipMemset(running, 0, sizeof(double) * (1 + 2 * d)));
    CHECK_HIP(hipMemset(cov_acc, 0, sizeof(double) * (size_t)cov_len));
    CHECK_HIP(hipMemset(send, 0, sizeof(double) * seg));
    CHECK_HIP(hipMemset(status, 0, sizeof(int32_t) * 2 * N));
    double ones[8] = {1, 1, 1, 1, 1, 1, 1, 1};
    CHECK_HIP(hipMemcpy(inv_std, ones, sizeof ones, hipMemcpyHostToDevice));

    sw_ars_pipeline *pl;
    CHECK_SW(sw_ars_pipeline_create(&pl));
    for (int it = 0; it < iters; ++it) {
        const int s = sw_ars_pipeline_next_slot(pl);   /* the pipeline's own call count mod slots */
        CHECK_SW(sw_ars_pipeline_host_slot_wait(pl, s));
        memcpy(dh[s], all + per_iter * it, sizeof(double) * per_iter);
        CHECK_SW(sw_ars_iteration_rollouts_f64(pl, s, &p, N, 0, N, H, dh[s], dd[s], policy, nu, mean, inv_std,
                                               send, traj[s], send + 2 * N, cov_acc, status, NULL));
        /* one rank: the "gathered" buffer is this rank's own segment */
        CHECK_SW(sw_ars_iteration_update_f64(pl, s, &p, N, send, 1, N, rows, dd[s], policy, alpha, (double)N, 0,
                                             running, (int64_t)2 * N * H, mean, inv_std, sigma, NULL));
    }
    CHECK_SW(sw_ars_pipeline_sync_cov(pl));
    CHECK_HIP(hipDeviceSynchronize());

    double *out = malloc(sizeof(double) * (md + 2 * d + ncov + 2 * N));
    CHECK_HIP(hipMemcpy(out, policy, sizeof(double) * md, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + md, mean, sizeof(double) * d, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + md + d, inv_std, sizeof(double) * d, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + md + 2 * d, cov_acc, sizeof(double) * ncov, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + md + 2 * d + ncov, send, sizeof(double) * 2 * N, hipMemcpyDeviceToHost));
    int32_t bad = 0, *hs = malloc(sizeof(int32_t) * 2 * N);
    CHECK_HIP(hipMemcpy(hs, status, sizeof(int32_t) * 2 * N, hipMemcpyDeviceToHost));
    for (int i = 0; i < 2 * N; ++i) bad |= hs[i];
    f = fopen(argv[5], "wb");
    if (!f || fwrite(out, sizeof(double), md + 2 * d + ncov + 2 * N, f) != (size_t)(md + 2 * d + ncov + 2 * N)) return 1;
    fclose(f);
    sw_ars_pipeline_destroy(pl);
    printf("iterations %d status %d\n", iters, (int)bad);
    return 0;
}
