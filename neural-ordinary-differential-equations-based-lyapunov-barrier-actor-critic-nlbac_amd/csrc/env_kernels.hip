// Batched device simulators: N independent environments advanced by ONE launch — the reference's
// simulators (one numpy env per process, float64) restated per lane, in float64, with the reference's step
// contracts: U/envs/unicycle_env.py:57-152 (+ the barrier signal of NU/envs/unicycle_env.py:116-144),
// C/envs/simulated_cars_env.py:66-146, P/envs/pvtol_env.py:85-216 (+ NP/envs/pvtol_env.py:144-220).
// One lane per environment: these are a few dozen flops per step — the point is that observations, rewards and the
// Lyapunov inputs are produced where the replay and the policy live (no host round trip per env step), not speed of
// the arithmetic.  Checked against traces recorded from the reference's own envs (tests/test_device_envs_gpu.py).
#include "common.h"

// obs (7) = [x, y, cos, sin, compass(2), exp(-dist)]  (unicycle_env.py:257-273)
__device__ __forceinline__ void unicycle_obs(const double* st, double gx, double gy, double* o) {
    const double rx = gx - st[0], ry = gy - st[1];
    const double dist = sqrt(rx * rx + ry * ry);
    const double c = cos(st[2]), s = sin(st[2]);
    const double v0 = rx * c + ry * s, v1 = -rx * s + ry * c;
    const double n = sqrt(v0 * v0 + v1 * v1) + 0.001;
    o[0] = st[0]; o[1] = st[1]; o[2] = c; o[3] = s; o[4] = v0 / n; o[5] = v1 / n; o[6] = exp(-dist);
}

struct UniEnv { double dt, gx, gy, goal_size, reward_goal, hz_r, l_p, little_b, capital_b; int max_steps, n_hz; };

__global__ __launch_bounds__(256) void unicycle_env_step_kernel(int n, const UniEnv E, const double* hazards,
                                                                const double* action, double* state, int* ep_step,
                                                                double* last_dist, double* obs, double* reward,
                                                                double* constraint, double* signal, double* center,
                                                                double* next_center, int* done, double* info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* st = state + 3 * i;
    const double a0 = action[2 * i], a1 = action[2 * i + 1];
    center[2 * i] = st[0] + E.l_p * cos(st[2]);
    center[2 * i + 1] = st[1] + E.l_p * sin(st[2]);
    double th = st[2];
    st[0] = st[0] + E.dt * (cos(th) * a0);
    st[1] = st[1] + E.dt * (sin(th) * a0);
    st[2] = st[2] + E.dt * a1;
    th = st[2];                                         // small drag along the new heading
    st[0] = st[0] - E.dt * 0.1 * cos(th) * cos(th);
    st[1] = st[1] - E.dt * 0.1 * sin(th) * cos(th);
    const double ncx = st[0] + E.l_p * cos(th), ncy = st[1] + E.l_p * sin(th);
    next_center[2 * i] = ncx; next_center[2 * i + 1] = ncy;
    ep_step[i] += 1;
    const double dx = E.gx - ncx, dy = E.gy - ncy;
    const double dist = sqrt(dx * dx + dy * dy);
    double r = -((a0 - 2.5) * (a0 - 2.5)) * 0.1 + (last_dist[i] - dist) * 30.0;
    last_dist[i] = dist;
    int d;
    double goal = 0.0;
    if (dist <= E.goal_size) { goal = 1.0; r += E.reward_goal; d = 1; }
    else d = ep_step[i] >= E.max_steps;
    double sig = E.little_b, nviol = 0.0, cost = 0.0;
    for (int k = 0; k < E.n_hz; ++k) {
        const double hx = ncx - hazards[2 * k], hy = ncy - hazards[2 * k + 1];
        const double d2 = hx * hx + hy * hy;
        if (d2 < E.hz_r * E.hz_r) {
            sig = (sig == E.little_b) ? E.capital_b : sig + E.capital_b;
            nviol += 1.0;
            cost += (E.hz_r - sqrt(d2)) / E.hz_r;
        }
    }
    unicycle_obs(st, E.gx, E.gy, obs + 7 * i);
    reward[i] = r; constraint[i] = dist; signal[i] = sig; done[i] = d;
    info[3 * i] = goal; info[3 * i + 1] = nviol; info[3 * i + 2] = cost;
}

// obs (11) = [x, y, cos, sin, vx, vy, thrust, operator x, compass(2), exp(-dist)]  (pvtol_env.py:361-406)
__device__ __forceinline__ void pvtol_obs(const double* st, double gx, double gy, double* o) {
    const double rx = gx - st[0], ry = gy - st[1];
    const double dist = sqrt(rx * rx + ry * ry);
    const double c = cos(st[2]), s = sin(st[2]);
    const double v0 = rx * c + ry * s, v1 = -rx * s + ry * c;
    const double n = sqrt(v0 * v0 + v1 * v1) + 0.001;
    o[0] = st[0]; o[1] = st[1]; o[2] = c; o[3] = s; o[4] = st[3]; o[5] = st[4]; o[6] = st[5]; o[7] = st[6];
    o[8] = v0 / n; o[9] = v1 / n; o[10] = exp(-dist);
}

struct PvEnv { double dt, gx, gy, goal_size, reward_goal, hz_r, follow, little_b, capital_b; int max_steps, n_hz; };

__global__ __launch_bounds__(256) void pvtol_env_step_kernel(int n, const PvEnv E, const double* hazards,
                                                             const double* action, double* state, int* ep_step,
                                                             double* obs, double* reward, double* constraint,
                                                             double* signal, double* lya_pre, int* done, double* info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* st = state + 7 * i;
    pvtol_obs(st, E.gx, E.gy, lya_pre + 11 * i);           // the Lyapunov input before the step = the current observation
    const double a0 = action[2 * i], a1 = action[2 * i + 1];
    double x[6] = {st[0], st[1], st[2], st[3], st[4], st[5]};
    const double f[6] = {x[3], x[4], 0.0, -sin(x[2]) * x[5], cos(x[2]) * x[5] - 1.0, 0.0};
    const double gu[6] = {0.0, 0.0, a1, 0.0, 0.0, a0};
    for (int k = 0; k < 6; ++k) x[k] = x[k] + E.dt * (f[k] + gu[k]);
    const double op = st[6] + E.follow * (x[0] - st[6]);
    for (int k = 0; k < 6; ++k) st[k] = x[k];
    st[6] = op;
    ep_step[i] += 1;
    const double dx = E.gx - st[0], dy = E.gy - st[1];
    const double dist = sqrt(dx * dx + dy * dy);
    double r = -1e-3 * dist;
    int d;
    double goal = 0.0;
    if (dist <= E.goal_size) { goal = 1.0; r += E.reward_goal; d = 1; }
    else d = ep_step[i] >= E.max_steps;
    double sig = E.little_b, nviol = 0.0, cost = 0.0;
    for (int k = 0; k < E.n_hz; ++k) {
        const double hx = st[0] - hazards[2 * k], hy = st[1] - hazards[2 * k + 1];
        const double d2 = hx * hx + hy * hy;
        if (d2 < E.hz_r * E.hz_r) {
            sig = (sig == E.little_b) ? E.capital_b : sig + E.capital_b;
            nviol += 1.0;
            cost += (E.hz_r - sqrt(d2)) / E.hz_r;
        }
    }
    pvtol_obs(st, E.gx, E.gy, obs + 11 * i);
    reward[i] = r; constraint[i] = dist; signal[i] = sig; done[i] = d;
    info[3 * i] = goal; info[3 * i + 1] = nviol; info[3 * i + 2] = cost;
}

struct CarsEnv { double dt, kp, k_brake, should_keep, keep_thre, reward_goal; int max_steps; };

__global__ __launch_bounds__(256) void cars_env_step_kernel(int n, const CarsEnv E, const double* action, double* state,
                                                            double* t, int* ep_step, double* obs, double* reward,
                                                            double* constraint, double* lya_pre, double* lya_next,
                                                            int* done, double* info) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double* st = state + 10 * i;
    const double a = action[i];
    double pos[5], vel[5], acc[5];
    for (int k = 0; k < 5; ++k) { pos[k] = st[2 * k]; vel[k] = st[2 * k + 1]; }
    for (int k = 0; k < 5; ++k) {
        double vd = 3.0;
        if (k == 0) vd -= 4.0 * sin(t[i]);
        acc[k] = E.kp * (vd - vel[k]);
    }
    acc[1] += -E.k_brake * (pos[0] - pos[1]) * ((pos[0] - pos[1]) < 6.5 ? 1.0 : 0.0);
    acc[2] += -E.k_brake * (pos[1] - pos[2]) * ((pos[1] - pos[2]) < 6.5 ? 1.0 : 0.0);
    acc[3] = 0.0;
    acc[4] += -E.k_brake * (pos[2] - pos[4]) * ((pos[2] - pos[4]) < 13.0 ? 1.0 : 0.0);
    for (int k = 0; k < 5; ++k) acc[k] *= 1.1;
    for (int k = 0; k < 4; ++k) lya_pre[4 * i + k] = st[4 + k];
    for (int k = 0; k < 5; ++k) {
        const double f_p = vel[k], f_v = (k == 3) ? 0.0 : acc[k], g_v = (k == 3) ? 1.0 : 0.0;
        st[2 * k] = st[2 * k] + E.dt * (f_p + 0.0 * a);
        st[2 * k + 1] = st[2 * k + 1] + E.dt * (f_v + g_v * a);
    }
    t[i] += E.dt;
    ep_step[i] += 1;
    const double d34 = st[4] - st[6], d45 = st[6] - st[8];
    double r = -0.5 * fabs(a * a) / (double)E.max_steps;
    const double reached = fabs(d34 - E.should_keep) < E.keep_thre ? 1.0 : 0.0;
    r += E.reward_goal * reached;
    for (int k = 0; k < 5; ++k) { obs[10 * i + 2 * k] = st[2 * k] / 100.0; obs[10 * i + 2 * k + 1] = st[2 * k + 1] / 30.0; }
    for (int k = 0; k < 4; ++k) lya_next[4 * i + k] = st[4 + k];
    reward[i] = r; constraint[i] = fabs(d34 - E.should_keep);
    done[i] = ep_step[i] >= E.max_steps;
    info[3 * i] = reached;
    info[3 * i + 1] = (d34 < 2.5 ? 1.0 : 0.0) + (d45 < 2.5 ? 1.0 : 0.0);
    info[3 * i + 2] = fabs(d34 - 2.5) * (d34 < 2.5 ? 1.0 : 0.0) + fabs(d45 - 2.5) * (d45 < 2.5 ? 1.0 : 0.0);
}

#define ENV_GRID(n) dim3(nlbac_ceil_div((n), 256)), dim3(256), 0, (hipStream_t)s

extern "C" int nlbac_unicycle_env_step(int n, const double* params /* dt, goal x, goal y, goal size, goal reward,
                                       hazard radius, l_p, barrier signal off / on */, int max_steps,
                                       const double* hazards, int n_hz, const double* action, double* state,
                                       int* ep_step, double* last_dist, double* obs, double* reward, double* constraint,
                                       double* signal, double* center, double* next_center, int* done, double* info,
                                       nlbac_stream_t s) {
    NLBAC_REQUIRE(n >= 1 && params && hazards && action && state && ep_step && last_dist && obs && reward && constraint &&
                      signal && center && next_center && done && info, "nlbac_unicycle_env_step: null pointer");
    UniEnv E = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8], max_steps, n_hz};
    hipLaunchKernelGGL(unicycle_env_step_kernel, ENV_GRID(n), n, E, hazards, action, state, ep_step, last_dist, obs,
                       reward, constraint, signal, center, next_center, done, info);
    NLBAC_CHECK_LAUNCH("nlbac_unicycle_env_step");
    return 0;
}

extern "C" int nlbac_pvtol_env_step(int n, const double* params /* dt, goal x, goal y, goal size, goal reward, hazard
                                    radius, operator follow, barrier signal off / on */, int max_steps,
                                    const double* hazards, int n_hz, const double* action, double* state, int* ep_step,
                                    double* obs, double* reward, double* constraint, double* signal, double* lya_pre,
                                    int* done, double* info, nlbac_stream_t s) {
    NLBAC_REQUIRE(n >= 1 && params && hazards && action && state && ep_step && obs && reward && constraint && signal &&
                      lya_pre && done && info, "nlbac_pvtol_env_step: null pointer");
    PvEnv E = {params[0], params[1], params[2], params[3], params[4], params[5], params[6], params[7], params[8], max_steps, n_hz};
    hipLaunchKernelGGL(pvtol_env_step_kernel, ENV_GRID(n), n, E, hazards, action, state, ep_step, obs, reward,
                       constraint, signal, lya_pre, done, info);
    NLBAC_CHECK_LAUNCH("nlbac_pvtol_env_step");
    return 0;
}

extern "C" int nlbac_cars_env_step(int n, const double* params /* dt, kp, k_brake, should_keep, keep threshold, goal
                                   reward */, int max_steps, const double* action, double* state, double* t,
                                   int* ep_step, double* obs, double* reward, double* constraint, double* lya_pre,
                                   double* lya_next, int* done, double* info, nlbac_stream_t s) {
    NLBAC_REQUIRE(n >= 1 && params && action && state && t && ep_step && obs && reward && constraint && lya_pre &&
                      lya_next && done && info, "nlbac_cars_env_step: null pointer");
    CarsEnv E = {params[0], params[1], params[2], params[3], params[4], params[5], max_steps};
    hipLaunchKernelGGL(cars_env_step_kernel, ENV_GRID(n), n, E, action, state, t, ep_step, obs, reward, constraint,
                       lya_pre, lya_next, done, info);
    NLBAC_CHECK_LAUNCH("nlbac_cars_env_step");
    return 0;
}
