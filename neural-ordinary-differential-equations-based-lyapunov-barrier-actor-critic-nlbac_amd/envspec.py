"""Plain-object stand-ins for the attributes ``SAC_CBF_CLF`` reads from ``env``.

The reference builds gym environments (``envs/unicycle_env.py``); ``gym`` is
not part of this build and the simulators are out of scope (SURVEY.md §2 row
8).  The update path only reads constants from ``env``:
``dynamics_mode``, ``dt``, ``hazards_locations``, ``hazards_radius``,
``safe_action_space.low/high``, ``action_space.{shape,high,low,seed,sample}``
and ``seed()``.  A real gym env exposing the same attributes works unchanged.

Constants follow ``U/envs/unicycle_env.py:14-40`` (bounds ±3.5/±12, seven
hazards on a 1.5-spaced grid, radius 0.5, dt 0.02, goal (2.5, 2.5)).
"""
import numpy as np


class Box:
    """Minimal ``gym.spaces.Box`` look-alike (shape/low/high/seed/sample)."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is not None and low.shape != tuple(shape):
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high = low, high
        self.shape = low.shape
        self.dtype = dtype
        self._rng = np.random.RandomState()

    def seed(self, s=None):
        self._rng = np.random.RandomState(s)
        return [s]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class UnicycleSpec:
    """Constants of ``UnicycleEnv`` (U/envs/unicycle_env.py:14-40)."""

    dynamics_mode = "Unicycle"
    n_s, n_u, obs_dim, lya_in = 3, 2, 7, 2

    def __init__(self, seed=0):
        lo = np.array([-3.5, -12.0])
        hi = np.array([3.5, 12.0])
        self.safe_action_space = Box(lo, hi)
        self.action_space = Box(lo, hi)
        self.observation_space = Box(-1e10, 1e10, shape=(7,))
        self.hazards_radius = 0.5
        self.hazards_locations = np.array(
            [[0., 0.], [0., 1.], [0., -1.], [-1., 1.], [-1., -1.], [1., -1.], [1., 1.]]) * 1.5
        self.dt = 0.02
        self.goal_pos = np.array([2.5, 2.5])
        self.max_episode_steps = 1200
        self.seed(seed)

    def seed(self, s=None):
        self.action_space.seed(s)
        return [s]


class SimulatedCarsSpec:
    """Constants of ``SimulatedCarsEnv`` (C/envs/simulated_cars_env.py:16-40): five cars on a line, the
    4th is controlled (acceleration in [-3, 3]); obs = state with positions /100 and velocities /30."""

    dynamics_mode = "SimulatedCars"
    n_s, n_u, obs_dim, lya_in = 10, 1, 10, 4

    def __init__(self, seed=0):
        self.action_space = Box(-3.0, 3.0, shape=(1,))
        self.safe_action_space = Box(-3.0, 3.0, shape=(1,))
        self.observation_space = Box(-1e10, 1e10, shape=(10,))
        self.dt = 0.02
        self.max_episode_steps = 300
        self.kp, self.k_brake = 4.0, 20.0
        self.should_keep = 9.5
        self.seed(seed)

    def seed(self, s=None):
        self.action_space.seed(s)
        return [s]


class PvtolSpec:
    """Constants of ``PvtolEnv`` (P/envs/pvtol_env.py:16-62): planar VTOL, state [x, y, theta, vx, vy, thrust] plus
    the x-position of a safety operator that follows the vehicle; obs (11) = [x, y, cos, sin, vx, vy, thrust,
    operator x, compass(2), exp(-dist to goal)]."""

    dynamics_mode = "Pvtol"
    n_s, n_u, obs_dim, lya_in = 6, 2, 11, 11

    def __init__(self, seed=0, y_min=-100.0, y_max=100.0, operator_dist=1.0, safety_operator_follow=0.7):
        lo, hi = np.array([-3.5, -15.0]), np.array([3.5, 15.0])
        self.action_space = Box(lo, hi)
        self.safe_action_space = Box(lo, hi)
        self.observation_space = Box(-1e10, 1e10, shape=(11,))
        self.dt = 0.02
        self.max_episode_steps = 2000
        self.goal_pos = np.array([4.5, 4.5])
        self.safety_operator_follow = safety_operator_follow
        self.operator_dist = operator_dist
        self.y_min, self.y_max = y_min, y_max
        self.hazard_locations = np.array([[-2.5, -2.5], [-2.5, 2.5], [0.0, -3.5], [0.0, 3.5], [-4.5, 0.0]])
        self.hazards_radius = 0.25
        self.seed(seed)

    def seed(self, s=None):
        self.action_space.seed(s)
        return [s]


class QuadrotorLikeSpec:
    """BASELINE configs[4] "Quadrotor (safe-control-gym) + neural barrier certificate".  The reference's Quadrotor code
    is an EMPTY submodule (neural_barrier_certificate/safe-control-gym/); only prose exists
    (/root/reference/README.md:66-72, 190-192): a 2D quadrotor, no pre-defined CBFs, barrier signals D1 = -1.0 outside
    the allowed range and D2 = -10.0 on collision, a NODE on normalised [state (6) | action (2)] -> 6 with
    de-normalised outputs.  This is a SYNTHETIC stand-in of that shape — NO REFERENCE PARITY EXISTS for it: planar
    quadrotor state [x, x', z, z', theta, theta'] (the 2D quadrotor of safe-control-gym's paper), observation = state,
    two rotor thrusts as the action, dt 0.02; every constant below is this build's choice."""

    dynamics_mode = "Quadrotor"
    n_s, n_u, obs_dim, lya_in = 6, 2, 6, 6
    MASS, IYY, ARM, G = 0.027, 1.4e-5, 0.0397, 9.8

    def __init__(self, seed=0):
        hover = self.MASS * self.G / 2.0
        lo, hi = np.array([0.5 * hover] * 2), np.array([1.5 * hover] * 2)
        self.action_space = Box(lo, hi)
        self.safe_action_space = Box(lo, hi)
        self.observation_space = Box(-1e10, 1e10, shape=(6,))
        self.dt = 0.02
        self.max_episode_steps = 500
        self.goal_pos = np.array([0.0, 1.0])                 # (x, z)
        self.x_range, self.z_range = (-2.0, 2.0), (0.0, 2.0)
        self.obstacle, self.obstacle_radius = np.array([0.6, 0.6]), 0.25
        self.D1, self.D2 = -1.0, -10.0                       # README.md:190
        # normalisation of the NODE's inputs [state | action] and de-normalisation of its outputs (d state / dt)
        self.node_in_mean = np.array([0.0, 0.0, 1.0, 0.0, 0.0, 0.0, hover, hover])
        self.node_in_std = np.array([1.0, 1.0, 0.5, 1.0, 0.3, 2.0, 0.3 * hover, 0.3 * hover])
        self.node_out_mean = np.zeros(6)
        self.node_out_std = np.array([1.0, 4.0, 1.0, 4.0, 2.0, 60.0])
        self.seed(seed)

    @property
    def node_normalizer(self):
        return (self.node_in_mean, self.node_in_std, self.node_out_mean, self.node_out_std)

    def seed(self, s=None):
        self.action_space.seed(s)
        return [s]


def make_env(name, seed=0, **overrides):
    """``UnicycleBarrier`` is the learned-barrier-certificate copy (``neural_barrier_certificate/``): the same
    Unicycle constants (its ``dynamics_mode`` is still ``'Unicycle'``); the agent class differs, not the env."""
    if name in ("Unicycle", "UnicycleBarrier"):
        return UnicycleSpec(seed)
    if name == "SimulatedCars":
        return SimulatedCarsSpec(seed)
    if name in ("Pvtol", "PvtolBarrier"):
        return PvtolSpec(seed, **overrides)
    if name == "QuadrotorLike":
        return QuadrotorLikeSpec(seed)
    raise Exception("Dynamics mode not supported.")
