// rlglue_env.cpp -- RL-Glue C environment plug-in (include/rlglue_swimmer.h) whose physics
// step runs on the GPU: the native twin of the reference's
// rlglue/environment/SwimmerEnvironment.cpp (env_* callbacks :14-98, save/load :328-358,
// parameter file :297-326).  One swimmer, one step per call: this is the drop-in boundary
// of the reference's native component, not a throughput path (the batched path is the C ABI
// in swimmer_hip.h).

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/rlglue_swimmer.h"
#include "../../include/swimmer_hip.h"

namespace {

// parameters (file-scope like the reference's globals, SwimmerEnvironment.cpp:3-9), initialised
// to rlglue/parameters.txt's values
struct Params {
    double dir_x = 1.0, dir_y = 0.0;
    unsigned n_seg = 3;
    double max_u = 5.0, l_i = 1.0, k = 10.0, m_i = 1.0, h_global = 0.01;
} g_par;

observation_t g_obs{}, g_saved{};
reward_observation_terminal_t g_ro{};
std::vector<double> g_obs_buf, g_saved_buf;
sw_env1 *g_env1 = nullptr;   // pinned, device-mapped I/O block + stream (swimmer_hip.h)
std::string g_task_spec, g_param_msg;

void fail(const char *what)
{
    std::fprintf(stderr, "swimmer RL-Glue environment: %s\n", what);
    std::abort();
}

sw_params make_params()
{
    sw_params p;
    p.n = (int32_t)g_par.n_seg;
    p.flags = SW_FLAG_MODEL_TWIN;
    p.l_i = g_par.l_i;
    p.m_i = g_par.m_i;
    p.k = g_par.k;
    p.h = g_par.h_global;
    p.dir_x = g_par.dir_x;
    p.dir_y = g_par.dir_y;
    return p;
}

void free_device()
{
    if (g_env1) sw_env1_destroy(g_env1);
    g_env1 = nullptr;
}

// `key value` lines (SwimmerEnvironment.cpp:297-326)
bool set_parameters(const std::string &file)
{
    std::ifstream in(file);
    if (!in.is_open()) return false;
    std::string line;
    while (std::getline(in, line)) {
        std::stringstream ss(line);
        std::string name;
        ss >> name;
        if (name == "n_seg") ss >> g_par.n_seg;
        else if (name == "max_u") ss >> g_par.max_u;
        else if (name == "l_i") ss >> g_par.l_i;
        else if (name == "k") ss >> g_par.k;
        else if (name == "m_i") ss >> g_par.m_i;
        else if (name == "h_global") ss >> g_par.h_global;
        else if (name == "direction") ss >> g_par.dir_x >> g_par.dir_y;
    }
    return true;
}

}  // namespace

extern "C" {

const char *env_init(void)
{
    const unsigned n_obs = 2 + 2 * g_par.n_seg, n_action = g_par.n_seg - 1;
    if (g_par.n_seg < 2 || g_par.n_seg > SW_MAX_SEGMENTS) fail("n_seg outside 2..8");
    g_obs_buf.assign(n_obs, 0.0);
    g_saved_buf.assign(n_obs, 0.0);
    g_obs = observation_t{0, n_obs, 0, nullptr, g_obs_buf.data(), nullptr};
    g_saved = observation_t{0, n_obs, 0, nullptr, g_saved_buf.data(), nullptr};
    g_ro.observation = &g_obs;
    g_ro.reward = 0;
    g_ro.terminal = 0;
    free_device();
    if (sw_env1_create(&g_env1) != SW_OK) fail("no GPU / sw_env1_create failed (there is no CPU fallback)");
    // the reference's task specification string, character for character (:30)
    g_task_spec = "VERSION RL-Glue-3.0 PROBLEMTYPE continuing DISCOUNTFACTOR 0.9 OBSERVATIONS DOUBLES (" +
                  std::to_string(n_obs) + " UNSPEC UNSPEC) ACTIONS DOUBLES (" + std::to_string(n_action) +
                  " " + std::to_string(-g_par.max_u) + " " + std::to_string(g_par.max_u) +
                  ") REWARDS (UNSPEC UNSPEC) EXTRA SwimmerEnvironment(C++) by Leon Zheng";
    return g_task_spec.c_str();
}

const observation_t *env_start(void)
{
    for (unsigned i = 0; i < g_obs.numDoubles; ++i) g_obs.doubleArray[i] = 0.001;   // :39-42
    g_saved_buf = g_obs_buf;                                                        // save_state()
    return &g_obs;
}

const reward_observation_terminal_t *env_step(const action_t *a)
{
    if (!a || a->numDoubles != g_par.n_seg - 1) fail("action has the wrong number of doubles");
    for (unsigned i = 0; i + 1 < g_par.n_seg; ++i)
        if (!(std::fabs(a->doubleArray[i]) <= g_par.max_u)) fail("action outside [-max_u, max_u]");
    const unsigned n_obs = g_obs.numDoubles, n_act = g_par.n_seg - 1;
    const sw_params p = make_params();
    if (!g_env1) fail("env_step before env_init");
    // ONE launch and one host wait per step: observation and action go over in the handle's pinned,
    // device-mapped block, which the kernel reads and writes directly (no hipMemcpy calls)
    double *io = sw_env1_io(g_env1);
    std::memcpy(io + SW_ENV1_STATE, g_obs.doubleArray, sizeof(double) * n_obs);
    std::memcpy(io + SW_ENV1_ACTION, a->doubleArray, sizeof(double) * n_act);
    const int rc = sw_env1_step(g_env1, &p, nullptr);
    if (rc != SW_OK) fail(sw_strerror(rc));
    std::memcpy(g_obs.doubleArray, io + SW_ENV1_NEXT, sizeof(double) * n_obs);
    const double r = io[SW_ENV1_REWARD];
    g_ro.observation = &g_obs;
    g_ro.reward = r;          // calculate_reward: Gdot_new . direction (:273-277)
    g_ro.terminal = 0;        // check_terminal (:279-282)
    return &g_ro;
}

void env_cleanup(void)
{
    free_device();
    g_obs_buf.clear();
    g_saved_buf.clear();
    g_obs = observation_t{};
    g_saved = observation_t{};
}

const char *env_message(const char *message)
{
    if (std::strcmp(message, "what is your name?") == 0)
        return "My name is swimmer_environment, C++ edition!";
    if (std::strcmp(message, "save state") == 0) {
        g_saved_buf = g_obs_buf;
        return "saved_observation has the value of this_observation";
    }
    if (std::strcmp(message, "load state") == 0) {
        if (g_saved_buf.size() == g_obs_buf.size())
            std::memcpy(g_obs_buf.data(), g_saved_buf.data(), sizeof(double) * g_obs_buf.size());
        return "this_observation has the value of saved_observation";
    }
    if (std::strcmp(message, "set parameters") == 0) {
        const char *file = std::getenv("SWIMMER_PARAMETERS");
        if (!set_parameters(file ? file : "../parameters.txt"))
            std::fprintf(stderr, "Unable to open file for setting environment parameters\n");
        g_param_msg = "Environment parameters are: n_seg=" + std::to_string(g_par.n_seg) +
                      "; max_u=" + std::to_string(g_par.max_u) + "; l_i=" + std::to_string(g_par.l_i) +
                      "; k=" + std::to_string(g_par.k) + "; m_i=" + std::to_string(g_par.m_i) +
                      "; h_global=" + std::to_string(g_par.h_global);
        return g_param_msg.c_str();
    }
    return "SwimmerEnvironment(C++) does not respond to that message.";
}

}  // extern "C"
