"""Deterministic synthetic replay transitions and network weights.

Everything here is generated from ``numpy.random.RandomState`` (a frozen
stream), so the golden-fixture generator (``oracle/gen_golden.py``), the
tests and ``bench.py`` reproduce identical inputs from a seed alone — the
fixtures then only need to hold expected outputs.

Transition layout follows SURVEY.md §8(d) "Synthetic inputs": the obs vector
is built like ``U/envs/unicycle_env.py:257-273``, the next state is one true
env Euler step (``:102-103``), centre / next-centre are the look-ahead points
(``:95-107``) and the minibatch field order is that of
``U/sac_cbf_clf/replay_memory.py:24-25``.
"""
import numpy as np

L_P = 0.03  # look-ahead distance, U/sac_cbf_clf/sac_cbf_clf.py:13

FIELDS = ("obs", "action", "reward", "constraint", "center", "next_center",
          "next_obs", "mask", "t", "next_t")


def _unicycle_obs(state, goal):
    x, y, th = state[:, 0], state[:, 1], state[:, 2]
    rel = goal[None, :] - state[:, :2]
    dist = np.linalg.norm(rel, axis=1)
    c, s = np.cos(th), np.sin(th)
    # vec @ R with R = [[c,-s],[s,c]]
    v0 = rel[:, 0] * c + rel[:, 1] * s
    v1 = -rel[:, 0] * s + rel[:, 1] * c
    n = np.sqrt(v0 * v0 + v1 * v1) + 0.001
    return np.stack([x, y, c, s, v0 / n, v1 / n, np.exp(-dist)], axis=1)


def unicycle_transitions(n, seed=1, env=None):
    """Return a dict of float64 numpy arrays (the reference replay holds
    float64 numpy rows too), ``n`` transitions."""
    from .envspec import UnicycleSpec
    env = env or UnicycleSpec()
    rs = np.random.RandomState(seed)
    dt, goal = env.dt, env.goal_pos
    state = np.stack([rs.uniform(-3, 3, n), rs.uniform(-3, 3, n),
                      rs.uniform(-np.pi, np.pi, n)], axis=1)
    lo, hi = env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64)
    action = rs.uniform(lo, hi, size=(n, 2))
    obs = _unicycle_obs(state, goal)
    center = state[:, :2] + L_P * np.stack([np.cos(state[:, 2]), np.sin(state[:, 2])], 1)
    # one env Euler step: x += dt * g(x) u ; then the small drag term
    nxt = state.copy()
    nxt[:, 0] += dt * np.cos(state[:, 2]) * action[:, 0]
    nxt[:, 1] += dt * np.sin(state[:, 2]) * action[:, 0]
    nxt[:, 2] += dt * action[:, 1]
    drag = np.cos(nxt[:, 2])
    nxt[:, 0] -= dt * 0.1 * np.cos(nxt[:, 2]) * drag
    nxt[:, 1] -= dt * 0.1 * np.sin(nxt[:, 2]) * drag
    next_obs = _unicycle_obs(nxt, goal)
    next_center = nxt[:, :2] + L_P * np.stack([np.cos(nxt[:, 2]), np.sin(nxt[:, 2])], 1)
    d_prev = np.linalg.norm(goal[None] - center, axis=1)
    d_next = np.linalg.norm(goal[None] - next_center, axis=1)
    reward = -np.square(action[:, 0] - 2.5) * 0.1 + (d_prev - d_next) * 30
    t = rs.randint(1, 1200, n).astype(np.float64) * dt
    return dict(obs=obs, action=action, reward=reward, constraint=d_next,
                center=center, next_center=next_center, next_obs=next_obs,
                mask=np.ones(n), t=t, next_t=t + dt)


FIELDS_BARRIER = ("obs", "action", "reward", "constraint", "barrier_signal", "center", "next_center",
                  "next_obs", "mask", "t", "next_t")     # NU/sac_cbf_clf/replay_memory.py:24-25


def unicycle_barrier_transitions(n, seed=1, env=None):
    """Unicycle transitions plus the barrier signal of the learned-certificate variant
    (NU/envs/unicycle_env.py:116-144): 0, and -20 for every hazard whose disc holds the next look-ahead point."""
    from .envspec import UnicycleSpec
    env = env or UnicycleSpec()
    tr = unicycle_transitions(n, seed, env)
    d2 = ((tr["next_center"][:, None, :] - np.asarray(env.hazards_locations)[None]) ** 2).sum(2)
    tr["barrier_signal"] = -20.0 * (d2 < env.hazards_radius ** 2).sum(1).astype(np.float64)
    return tr


def pvtol_barrier_transitions(n, seed=1, env=None):
    """Pvtol transitions plus the barrier signal of the learned-certificate copy (NP/envs/pvtol_env.py:144-220):
    0, and -0.1 per hazard disc holding the next position."""
    from .envspec import PvtolSpec
    env = env or PvtolSpec()
    tr = pvtol_transitions(n, seed, env)
    d2 = ((tr["next_obs"][:, None, :2] - np.asarray(env.hazard_locations)[None]) ** 2).sum(2)
    tr["barrier_signal"] = -0.1 * (d2 < env.hazards_radius ** 2).sum(1).astype(np.float64)
    return tr


def cars_transitions(n, seed=1, env=None):
    """Synthetic SimulatedCars transitions: states near the env's reset line-up
    (C/envs/simulated_cars_env.py:158-176) with random spreads, one true env step (``:66-99``),
    reward / constraint / Lyapunov inputs as ``step`` returns them (``:99-146``)."""
    from .envspec import SimulatedCarsSpec
    env = env or SimulatedCarsSpec()
    rs = np.random.RandomState(seed)
    dt = env.dt
    pos = np.array([42.0, 34.0, 26.0, 18.0, 10.0])[None] + rs.uniform(-3, 3, (n, 5)) + rs.uniform(0, 20, (n, 1))
    vel = 3.0 + rs.normal(0, 0.8, (n, 5))
    t = rs.randint(0, 300, n).astype(np.float64) * dt
    state = np.zeros((n, 10))
    state[:, ::2], state[:, 1::2] = pos, vel
    action = rs.uniform(-3.0, 3.0, (n, 1))
    vels_des = 3.0 * np.ones((n, 5))
    vels_des[:, 0] -= 4 * np.sin(t)
    acc = env.kp * (vels_des - vel)
    acc[:, 1] += -env.k_brake * (pos[:, 0] - pos[:, 1]) * ((pos[:, 0] - pos[:, 1]) < 6.5)
    acc[:, 2] += -env.k_brake * (pos[:, 1] - pos[:, 2]) * ((pos[:, 1] - pos[:, 2]) < 6.5)
    acc[:, 3] = 0.0
    acc[:, 4] += -env.k_brake * (pos[:, 2] - pos[:, 4]) * ((pos[:, 2] - pos[:, 4]) < 13.0)
    acc *= 1.1
    f = np.zeros((n, 10))
    f[:, ::2], f[:, 1::2] = vel, acc
    f[:, 7] = 0.0
    g = np.zeros((n, 10))
    g[:, 7] = 1.0
    nxt = state + dt * (f + g * action)

    def to_obs(x):
        o = x.copy()
        o[:, ::2] /= 100.0
        o[:, 1::2] /= 30.0
        return o
    d34 = nxt[:, 4] - nxt[:, 6]
    reward = -0.5 * np.abs(action[:, 0] ** 2) / env.max_episode_steps + 2.0 * (np.abs(d34 - env.should_keep) < 0.5)
    return dict(obs=to_obs(state), action=action, reward=reward, constraint=np.abs(d34 - env.should_keep),
                center=state[:, 4:8].copy(), next_center=nxt[:, 4:8].copy(), next_obs=to_obs(nxt),
                mask=np.ones(n), t=t, next_t=t + dt)


def _pvtol_obs(state, goal):
    """P/envs/pvtol_env.py:361-406 (compass = (goal - xy) @ R(theta), normalised with +0.001)."""
    th = state[:, 2]
    rel = goal[None, :] - state[:, :2]
    dist = np.linalg.norm(rel, axis=1)
    c, s = np.cos(th), np.sin(th)
    v0 = rel[:, 0] * c + rel[:, 1] * s
    v1 = -rel[:, 0] * s + rel[:, 1] * c
    n = np.sqrt(v0 * v0 + v1 * v1) + 0.001
    return np.stack([state[:, 0], state[:, 1], c, s, state[:, 3], state[:, 4], state[:, 5], state[:, 6],
                     v0 / n, v1 / n, np.exp(-dist)], axis=1)


def pvtol_transitions(n, seed=1, env=None):
    """Synthetic Pvtol transitions: random states around the flight corridor, one true env Euler step
    (P/envs/pvtol_env.py:85-160); the Lyapunov inputs are the observations before / after the step, as
    ``step`` returns them (``:82``)."""
    from .envspec import PvtolSpec
    env = env or PvtolSpec()
    rs = np.random.RandomState(seed)
    dt, goal = env.dt, env.goal_pos
    st = np.zeros((n, 7))
    st[:, 0], st[:, 1] = rs.uniform(-5.5, 5.5, n), rs.uniform(-5.5, 5.5, n)
    # a quarter of the rows sit close to a hazard so that the obstacle barriers are active on part of a minibatch
    kind = rs.uniform(0, 1, n)
    hz = np.asarray(env.hazard_locations)[rs.randint(0, len(env.hazard_locations), n)]
    near = kind < 0.25
    st[near, :2] = (hz + rs.uniform(-0.45, 0.45, (n, 2)))[near]
    st[:, 2] = rs.uniform(-0.8, 0.8, n)
    st[:, 3], st[:, 4] = rs.normal(0, 1.0, n), rs.normal(0, 1.0, n)
    st[:, 5] = rs.uniform(0.3, 1.8, n)
    st[:, 6] = st[:, 0] + rs.uniform(-2.5, 2.5, n)
    lo, hi = env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64)
    action = rs.uniform(lo, hi, size=(n, 2))
    obs = _pvtol_obs(st, goal)
    f = np.zeros((n, 6))
    f[:, 0], f[:, 1] = st[:, 3], st[:, 4]
    f[:, 3] = -np.sin(st[:, 2]) * st[:, 5]
    f[:, 4] = np.cos(st[:, 2]) * st[:, 5] - 1.0
    nxt = st.copy()
    nxt[:, :6] += dt * f
    nxt[:, 2] += dt * action[:, 1]
    nxt[:, 5] += dt * action[:, 0]
    nxt[:, 6] = st[:, 6] + env.safety_operator_follow * (nxt[:, 0] - st[:, 6])
    next_obs = _pvtol_obs(nxt, goal)
    dist = np.linalg.norm(goal[None] - nxt[:, :2], axis=1)
    t = rs.randint(0, 2000, n).astype(np.float64) * dt
    return dict(obs=obs, action=action, reward=-1e-3 * dist, constraint=dist, center=obs.copy(),
                next_center=next_obs.copy(), next_obs=next_obs, mask=np.ones(n), t=t, next_t=t + dt)


def quadrotor_transitions(n, seed=1, env=None):
    """Synthetic transitions of the Quadrotor-like task (``envspec.QuadrotorLikeSpec``; no reference code exists):
    random planar-quadrotor states, one Euler step of the rigid-body model, reward / cost = (minus) the distance to
    the goal, barrier signal D1 outside the allowed x / z range and D2 inside the obstacle (README.md:190); the
    observation is the state and the Lyapunov inputs are the observations (the NP pattern)."""
    from .envspec import QuadrotorLikeSpec
    env = env or QuadrotorLikeSpec()
    rs = np.random.RandomState(seed)
    dt = env.dt
    st = np.stack([rs.uniform(-2.2, 2.2, n), rs.normal(0, 0.8, n), rs.uniform(-0.1, 2.2, n), rs.normal(0, 0.8, n),
                   rs.uniform(-0.4, 0.4, n), rs.normal(0, 1.5, n)], axis=1)
    near = rs.uniform(0, 1, n) < 0.2          # a fifth of the rows sit around the obstacle
    st[near, 0] = env.obstacle[0] + rs.uniform(-0.4, 0.4, n)[near]
    st[near, 2] = env.obstacle[1] + rs.uniform(-0.4, 0.4, n)[near]
    lo, hi = env.action_space.low.astype(np.float64), env.action_space.high.astype(np.float64)
    action = rs.uniform(lo, hi, size=(n, 2))
    T = action.sum(1)
    f = np.stack([st[:, 1], np.sin(st[:, 4]) * T / env.MASS, st[:, 3], np.cos(st[:, 4]) * T / env.MASS - env.G,
                  st[:, 5], (action[:, 1] - action[:, 0]) * env.ARM / np.sqrt(2.0) / env.IYY], axis=1)
    nxt = st + dt * f
    dist = np.sqrt((nxt[:, 0] - env.goal_pos[0]) ** 2 + (nxt[:, 2] - env.goal_pos[1]) ** 2)
    out = (nxt[:, 0] < env.x_range[0]) | (nxt[:, 0] > env.x_range[1]) | (nxt[:, 2] < env.z_range[0]) | (nxt[:, 2] > env.z_range[1])
    hit = (nxt[:, 0] - env.obstacle[0]) ** 2 + (nxt[:, 2] - env.obstacle[1]) ** 2 < env.obstacle_radius ** 2
    sig = env.D1 * out.astype(np.float64) + env.D2 * hit.astype(np.float64)
    t = rs.randint(0, 500, n).astype(np.float64) * dt
    return dict(obs=st, action=action, reward=-dist + 250.0 * (dist < 0.05), constraint=dist, barrier_signal=sig,
                center=st.copy(), next_center=nxt.copy(), next_obs=nxt, mask=np.ones(n), t=t, next_t=t + dt)


# ---------------------------------------------------------------------------
# network shapes (reference key names; U/sac_cbf_clf/model.py:37-133,177-206)
# ---------------------------------------------------------------------------

def qnet_shapes(num_inputs, num_actions, hidden):
    i = num_inputs + num_actions
    return [("linear1", hidden, i), ("linear2", hidden, hidden), ("linear3", 1, hidden),
            ("linear4", hidden, i), ("linear5", hidden, hidden), ("linear6", 1, hidden)]


def lya_shapes(num_inputs, hidden):
    return [("linear1", hidden, num_inputs), ("linear2", hidden, hidden), ("linear3", 1, hidden)]


def policy_shapes(num_inputs, num_actions, hidden):
    return [("linear1", hidden, num_inputs), ("linear2", hidden, hidden),
            ("mean_linear", num_actions, hidden), ("log_std_linear", num_actions, hidden)]


def node_affine_shapes(n_s, n_u, hidden=100, f_hidden_layers=4, g_hidden_layers=3):
    f = [("f_net.0", hidden, n_s)]
    f += [("f_net.%d" % (2 * i), hidden, hidden) for i in range(1, f_hidden_layers)]
    f += [("f_net.%d" % (2 * f_hidden_layers), n_s, hidden)]
    g = [("g_net.0", hidden, n_s)]
    g += [("g_net.%d" % (2 * i), hidden, hidden) for i in range(1, g_hidden_layers)]
    g += [("g_net.%d" % (2 * g_hidden_layers), n_s * n_u, hidden)]
    return f + g


def node_single_shapes(in_dim, out_dim, hidden=64, depth=4):
    """C/sac_cbf_clf/model.py:186-194: ``net`` = depth Linear layers (ReLU between)."""
    dims = [in_dim] + [hidden] * (depth - 1) + [out_dim]
    return [("net.%d" % (2 * i), dims[i + 1], dims[i]) for i in range(depth)]


def synth_state_dict(shapes, seed, kind="xavier"):
    """Deterministic float32 weights for a list of (name, out, in) layers.

    ``xavier``: W ~ U(±sqrt(6/(in+out))) (reference rule ``model.py:14-17``),
    biases small non-zero so bias gradients/updates are exercised.
    ``default``: W, b ~ U(±1/sqrt(in)) (PyTorch ``nn.Linear`` default, which
    the reference NODE keeps, ``model.py:186-206``).
    """
    rs = np.random.RandomState(seed)
    sd = {}
    for name, n_out, n_in in shapes:
        if kind == "xavier":
            bw = np.sqrt(6.0 / (n_in + n_out))
            bb = 0.05
        else:
            bw = bb = 1.0 / np.sqrt(n_in)
        sd[name + ".weight"] = rs.uniform(-bw, bw, size=(n_out, n_in)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-bb, bb, size=(n_out,)).astype(np.float32)
    return sd


def unicycle_agent_weights(hidden, seed=0):
    """All seven Unicycle nets (targets start as copies, like hard_update)."""
    critic = synth_state_dict(qnet_shapes(7, 2, hidden), seed * 10 + 1)
    lya = synth_state_dict(lya_shapes(2, hidden), seed * 10 + 2)
    policy = synth_state_dict(policy_shapes(7, 2, hidden), seed * 10 + 3)
    backup = synth_state_dict(policy_shapes(7, 2, hidden), seed * 10 + 4)
    node = synth_state_dict(node_affine_shapes(3, 2), seed * 10 + 5, kind="default")
    return dict(critic=critic, lyapunov=lya, policy=policy, backup_policy=backup, node=node)


def cars_agent_weights(hidden, seed=0):
    critic = synth_state_dict(qnet_shapes(10, 1, hidden), seed * 10 + 1)
    lya = synth_state_dict(lya_shapes(4, hidden), seed * 10 + 2)
    policy = synth_state_dict(policy_shapes(10, 1, hidden), seed * 10 + 3)
    backup = synth_state_dict(policy_shapes(10, 1, hidden), seed * 10 + 4)
    node = synth_state_dict(node_single_shapes(12, 10), seed * 10 + 5, kind="default")
    return dict(critic=critic, lyapunov=lya, policy=policy, backup_policy=backup, node=node)


def unicycle_barrier_agent_weights(hidden, seed=0):
    """NU: no backup policy; a BarrierNetwork on (obs, action) (NU/sac_cbf_clf/model.py:67-84)."""
    W = unicycle_agent_weights(hidden, seed)
    del W["backup_policy"]
    W["barrier"] = synth_state_dict(lya_shapes(7 + 2, hidden), seed * 10 + 6)
    return W


def pvtol_agent_weights(hidden, seed=0):
    critic = synth_state_dict(qnet_shapes(11, 2, hidden), seed * 10 + 1)
    lya = synth_state_dict(lya_shapes(11, hidden), seed * 10 + 2)
    policy = synth_state_dict(policy_shapes(11, 2, hidden), seed * 10 + 3)
    backup = synth_state_dict(policy_shapes(11, 2, hidden), seed * 10 + 4)
    node = synth_state_dict(node_affine_shapes(6, 2), seed * 10 + 5, kind="default")
    return dict(critic=critic, lyapunov=lya, policy=policy, backup_policy=backup, node=node)


def pvtol_barrier_agent_weights(hidden, seed=0):
    W = pvtol_agent_weights(hidden, seed)
    del W["backup_policy"]
    W["barrier"] = synth_state_dict(lya_shapes(11 + 2, hidden), seed * 10 + 6)
    return W


QUADROTOR_NODE_HIDDEN = 128


def quadrotor_agent_weights(hidden, seed=0):
    """One controller, a BarrierNetwork on (obs, action), a single-net NODE 8 -> 128 -> 128 -> 128 -> 6."""
    critic = synth_state_dict(qnet_shapes(6, 2, hidden), seed * 10 + 1)
    lya = synth_state_dict(lya_shapes(6, hidden), seed * 10 + 2)
    policy = synth_state_dict(policy_shapes(6, 2, hidden), seed * 10 + 3)
    node = synth_state_dict(node_single_shapes(8, 6, hidden=QUADROTOR_NODE_HIDDEN), seed * 10 + 5, kind="default")
    barrier = synth_state_dict(lya_shapes(6 + 2, hidden), seed * 10 + 6)
    return dict(critic=critic, lyapunov=lya, policy=policy, node=node, barrier=barrier)


def agent_weights(env_name, hidden, seed=0):
    return {"QuadrotorLike": quadrotor_agent_weights, "PvtolBarrier": pvtol_barrier_agent_weights, "Unicycle": unicycle_agent_weights, "SimulatedCars": cars_agent_weights, "Pvtol": pvtol_agent_weights,
            "UnicycleBarrier": unicycle_barrier_agent_weights}[env_name](hidden, seed)


def transitions(env_name, n, seed=1, env=None):
    return {"QuadrotorLike": quadrotor_transitions, "PvtolBarrier": pvtol_barrier_transitions, "Unicycle": unicycle_transitions, "SimulatedCars": cars_transitions, "Pvtol": pvtol_transitions,
            "UnicycleBarrier": unicycle_barrier_transitions}[env_name](n, seed, env)


# Fixture / test environment constants.  With the reference's Pvtol constants (|y| < 100 corridor; operator following
# at 0.7 per step, which makes its relative-degree-3 barrier 0.001 (x - op) + 0.41) the y_max / y_min / operator
# barriers only become active for states (|y| > 90, operator 500 m away) that drive the randomly initialised policy far
# into saturation, where the reference's own autograd gradient of Normal.log_prob is rounding noise ((x - mean)
# quantised against a std of 1e-9), and the operator barrier is a 1000:1 cancellation.  The fixtures therefore use a
# tighter corridor, a slower operator (barrier 0.216 (x - op) + 0.21) and a shorter leash - plain ``env`` attributes the
# agent reads - so that every barrier family is active, and well conditioned in fp32, on ordinary states.
FIXTURE_ENV = {"Pvtol": dict(y_min=-15.0, y_max=15.0, operator_dist=0.5, safety_operator_follow=0.2)}


def fixture_env(env_name, seed=0):
    from .envspec import make_env
    return make_env(env_name, seed, **FIXTURE_ENV.get(env_name, {}))


def fields(env_name):
    return FIELDS_BARRIER if (env_name.endswith("Barrier") or env_name == "QuadrotorLike") else FIELDS


def normal_eps(n_draws, batch, n_u, seed):
    """Pre-drawn reparameterisation noise ε ~ N(0,1), float32, one (B, n_u)
    array per policy sample in the reference draw order."""
    rs = np.random.RandomState(seed)
    return [rs.standard_normal((batch, n_u)).astype(np.float32) for _ in range(n_draws)]
