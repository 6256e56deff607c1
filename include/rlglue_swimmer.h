/*
 * rlglue_swimmer.h -- the RL-Glue 3.x C environment plug-in interface, as exported by
 * librlglue_swimmer_hip.so: the five entry points the reference's native environment
 * exports (rlglue/environment/SwimmerEnvironment.h:38-42) with RL-Glue's public struct
 * layouts (RL-Glue 3.04 `rlglue/RL_common.h`; restated here because RL-Glue is not
 * installed -- the layouts are its published C ABI).
 *
 * An RL-Glue environment loader (`librlenvironment` / `RL_glue`) binds exactly these
 * symbols; replacing the reference's SwimmerEnvironment object file with this library runs
 * the same environment (same task spec, same start state, same messages, same physics
 * model incl. its quirks) with the physics step executed on the MI355X through
 * sw_step_f64(SW_FLAG_MODEL_TWIN).
 *
 * Differences from the reference plug-in, all outside the arithmetic:
 *   - env_step does not print the state to stdout every step (SwimmerEnvironment.cpp:61);
 *   - parameters start at the values of rlglue/parameters.txt instead of zero-initialised
 *     globals; the "set parameters" message re-reads ../parameters.txt (or the file named by
 *     the SWIMMER_PARAMETERS environment variable), same `key value` format (:297-326);
 *   - an out-of-range action aborts with a message, as the reference's assert does (:56-59).
 */
#ifndef RLGLUE_SWIMMER_H
#define RLGLUE_SWIMMER_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    unsigned int numInts;
    unsigned int numDoubles;
    unsigned int numChars;
    int *intArray;
    double *doubleArray;
    char *charArray;
} rl_abstract_type_t;

typedef rl_abstract_type_t observation_t;
typedef rl_abstract_type_t action_t;

typedef struct {
    double reward;
    const observation_t *observation;
    int terminal;
} reward_observation_terminal_t;

const char *env_init(void);
const observation_t *env_start(void);
const reward_observation_terminal_t *env_step(const action_t *this_action);
void env_cleanup(void);
const char *env_message(const char *message);

#ifdef __cplusplus
}
#endif
#endif /* RLGLUE_SWIMMER_H */
