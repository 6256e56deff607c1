/* swimmer_oracle.h -- C interface of the CPU oracle (test infrastructure, see swimmer_oracle.c). */
#ifndef SWIMMER_ORACLE_H
#define SWIMMER_ORACLE_H

#define SWO_NMAX 16

#ifdef __cplusplus
extern "C" {
#endif

typedef struct swo_params {
    int n;          /* number of segments */
    double l_i;     /* segment length     (remy_swimmer_env.py:26) */
    double m_i;     /* segment mass       (:28) */
    double k;       /* viscous friction   (:27) */
    double h;       /* Euler time step    (:29) */
    double dir_x;   /* reward direction   (:23) */
    double dir_y;
} swo_params;

int swo_accelerations(const swo_params *p, const double *state, const double *u,
                      double *gdd, double *tdd);
int swo_step(const swo_params *p, const double *state, const double *u,
             double *next, double *reward);
void swo_reset(const swo_params *p, double *state);
int swo_rollout(const swo_params *p, int H, const double *policy, const double *mean,
                const double *cov_diag, const double *state0, double *ret, double *traj);
int swo_num_threads(void);
void swo_set_num_threads(int t);
int swo_step_batch(const swo_params *p, long n_env, const double *states,
                   const double *actions, double *next, double *rewards);
int swo_rollout_batch(const swo_params *p, long n_roll, int H, const double *policies,
                      const double *mean, const double *cov_diag, double *returns,
                      double *traj);

/* native twin (rlglue/environment/SwimmerEnvironment.cpp), twin_oracle.c */
int swt_accelerations(const swo_params *p, const double *state, const double *u,
                      double *gdd, double *tdd);
int swt_system(const swo_params *p, const double *state, const double *u, double *A_out,
               double *B_out, double *X_out);
int swt_step(const swo_params *p, const double *state, const double *u,
             double *next, double *reward);
int swt_step_batch(const swo_params *p, long n_env, const double *states,
                   const double *actions, double *next, double *rewards);

#ifdef __cplusplus
}
#endif
#endif
