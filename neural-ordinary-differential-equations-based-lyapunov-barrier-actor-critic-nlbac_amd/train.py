"""Reference-shaped training driver: the loop of ``U/main.py:20-165`` — warm-up with random
actions, one ``update_parameters`` per ``updates_per_step`` once the replay holds a batch, transitions pushed to the
controller replay and the NODE replay, and each copy's rules for handing control to the backup controller
(``U/main.py:108-142``, ``C/main.py:100-112``, ``P/main.py:128-200``; the learned-barrier copies have none) — on the
gym-free simulators of ``nlbac_amd.envs`` and the MI355X agent.  Logging is one
line per episode (the reference's wandb / spinup loggers are host tooling and out of scope).

    python -m nlbac_amd.train --env Unicycle --gamma_b 50 --max_episodes 200 --cuda --updates_per_step 2 \\
        --batch_size 128 --seed 0 --start_steps 1000          (the reference README's command line)

``--device_replay`` keeps both replays in HBM (same index stream, gather on the device).
"""
import argparse
import os
import time

import numpy as np


def get_args(argv=None):
    p = argparse.ArgumentParser(description="NLBAC training on MI355X (reference argument names)")
    p.add_argument('--env', default="Unicycle",
                   choices=["Unicycle", "SimulatedCars", "Pvtol", "UnicycleBarrier", "PvtolBarrier"])
    p.add_argument('--policy', default="Gaussian")
    p.add_argument('--gamma', type=float, default=0.99)
    p.add_argument('--tau', type=float, default=0.005)
    p.add_argument('--lr', type=float, default=0.0003)
    p.add_argument('--alpha', type=float, default=0.2)
    p.add_argument('--automatic_entropy_tuning', type=bool, default=True)
    p.add_argument('--seed', type=int, default=12345)
    p.add_argument('--batch_size', type=int, default=256)
    p.add_argument('--max_episodes', type=int, default=200)
    p.add_argument('--hidden_size', type=int, default=256)
    p.add_argument('--updates_per_step', type=int, default=1)
    p.add_argument('--start_steps', type=int, default=5000)
    p.add_argument('--target_update_interval', type=int, default=1)
    p.add_argument('--Lagrangian_multiplier_update_interval', type=int, default=8)
    p.add_argument('--NODE_model_update_interval', type=int, default=10)
    p.add_argument('--backup_update_interval', type=int, default=20)
    p.add_argument('--replay_size', type=int, default=10000000)
    p.add_argument('--cuda', action="store_true")
    p.add_argument('--gamma_b', type=float, default=20.0)
    p.add_argument('--solver', default="euler", choices=["euler", "rk4", "dopri5"])
    p.add_argument('--output', default=None, help="directory for save_model at the reference's cadence")
    p.add_argument('--device_replay', action="store_true")
    p.add_argument('--device_rng', action="store_true",
                   help="with --device_replay: draw minibatch indices (and the policy noise) on the device instead of "
                        "replaying the reference's host random.sample stream")
    p.add_argument('--hipgraphs', action="store_true",
                   help="replay each update as hipGraphs (pays at the reference's small batch sizes, where the host's "
                        "launch rate bounds an update)")
    p.add_argument('--max_steps', type=int, default=0, help="stop after this many env steps (0: run all episodes)")
    p.add_argument('--vector_envs', type=int, default=0,
                   help="N > 0: N environments stepping on the device (train_vectorized: no transition touches the host, "
                        "primary controller only); runs --max_steps env steps (default 100000)")
    return p.parse_args(argv)


class _Handover:
    """Which controller acts and which transitions reach the controller replay — the hand-over logic of the reference
    drivers, one subclass per ``main.py`` copy.  ``use_backup``: the backup controller acts this step (and the
    transition is kept out of ``memory``)."""
    first_backup_episode = None        # hand-over is allowed from this episode on (None: this copy has no backup)
    memory_t_shift = 0                 # SimulatedCars stamps the controller replay one step earlier (C/main.py:97-99)

    def __init__(self, env):
        self.env = env
        self.allowed = False

    def begin_episode(self, i_episode):
        if self.first_backup_episode is not None and i_episode >= self.first_backup_episode:
            self.allowed = True

    @property
    def use_backup(self):
        return False

    def on_backup_action(self):
        pass

    def after_step(self, episode_steps, lya_in, next_lya_in, next_obs, info):
        pass


class _UnicycleHandover(_Handover):
    """U/main.py:39-142: stuck for 8 checks (the look-ahead point moved less than 0.1 in 40 steps) -> the backup
    controller for at most 30 steps or until it has moved sqrt(0.6) away; allowed after episode 3."""
    first_backup_episode = 4
    STUCK2, CHECKS, MAX_BACKUP, AWAY2 = 0.01, 8, 30, 0.6

    def begin_episode(self, i_episode):
        super().begin_episode(i_episode)
        self.backup = False
        self.positions, self.backup_time, self.violation_time = [], 0, 0
        self.x0 = self.y0 = 0.0

    @property
    def use_backup(self):
        return self.backup and self.allowed

    def on_backup_action(self):
        self.backup_time += 1

    def after_step(self, episode_steps, lya_in, next_lya_in, next_obs, info):
        self.positions.append(np.asarray(next_lya_in))
        if episode_steps < 50:
            return
        diff = self.positions[-1] - self.positions[-40]
        moved = diff[0] * diff[0] + diff[1] * diff[1]
        if self.allowed and not self.backup:
            if moved <= self.STUCK2:
                self.violation_time += 1
                if self.violation_time >= self.CHECKS:
                    self.backup, self.violation_time = True, 0
                    self.x0, self.y0 = next_lya_in[0], next_lya_in[1]
            if moved > self.STUCK2 and self.violation_time > 0:
                self.violation_time = 0
        if self.backup and self.allowed:
            if self.backup_time >= self.MAX_BACKUP:
                self.backup, self.backup_time = False, 0
            dx, dy = next_lya_in[0] - self.x0, next_lya_in[1] - self.y0
            if dx * dx + dy * dy >= self.AWAY2:
                self.backup, self.backup_time = False, 0


class _CarsHandover(_Handover):
    """C/main.py:41-112: from episode 0 on; the 4th car closer than 2.5 to the 5th while the following distance is met
    -> the backup controller, for at most 15 steps, or from 5 steps on once both gaps exceed 2.5."""
    first_backup_episode = 0
    memory_t_shift = -1

    def begin_episode(self, i_episode):
        super().begin_episode(i_episode)
        self.backup, self.backup_time = False, 0

    @property
    def use_backup(self):
        return self.backup and self.allowed

    def on_backup_action(self):
        self.backup_time += 1

    def after_step(self, episode_steps, lya_in, next_lya_in, next_obs, info):
        d34 = next_obs[4] * 100.0 - next_obs[6] * 100.0
        d45 = next_obs[6] * 100.0 - next_obs[8] * 100.0
        if self.allowed and not self.backup:
            if d45 < 2.5 and info.get('reached', 0) != 0:
                self.backup = True
        if self.backup and self.allowed:
            if self.backup_time >= 15:
                self.backup, self.backup_time = False, 0
            if self.backup_time >= 5 and d34 > 2.5 and d45 > 2.5:
                self.backup, self.backup_time = False, 0


class _PvtolHandover(_Handover):
    """P/main.py:40-200: two reasons to hand over, each with its own timers — trapped (moved^2 <= 0.015 in 40 steps,
    8 checks; back after 30 steps or 1.0 away) and running away from the safety operator towards the goal (back after
    15 steps or once within 0.9 operator_dist); allowed from episode 3 on."""
    first_backup_episode = 3

    def begin_episode(self, i_episode):
        super().begin_episode(i_episode)
        self.obs_b = self.y_b = False
        self.positions = []
        self.obs_time = self.y_time = self.viol_obs = self.viol_y = 0
        self.x0 = self.y0 = 0.0

    @property
    def use_backup(self):
        return (self.obs_b and self.allowed) or (self.y_b and self.allowed)

    def on_backup_action(self):
        if self.obs_b and self.y_b:
            self.obs_time += 1
            self.y_time += 1
        elif self.obs_b and not self.y_b:
            self.obs_time += 1
        else:
            self.y_time += 1

    def after_step(self, episode_steps, lya_in, next_lya_in, next_obs, info):
        self.positions.append(np.asarray(next_lya_in))
        if episode_steps < 50:
            return
        env, nx, pv = self.env, next_lya_in, lya_in
        diff = self.positions[-1] - self.positions[-40]
        moved = diff[0] * diff[0] + diff[1] * diff[1]
        if self.allowed and not self.obs_b:
            if moved <= 0.015:
                self.viol_obs += 1
                if self.viol_obs >= 8:
                    self.obs_b, self.viol_obs = True, 0
                    self.x0, self.y0 = nx[0], nx[1]
            if moved > 0.015 and self.viol_obs > 0:
                self.viol_obs = 0
        if self.obs_b and self.allowed:
            if self.obs_time >= 30:
                self.obs_b, self.obs_time = False, 0
            dx, dy = nx[0] - self.x0, nx[1] - self.y0
            if dx * dx + dy * dy >= 1.0:
                self.obs_b, self.obs_time = False, 0
        running = ((nx[0] <= 4.5 and nx[0] - pv[0] > 0 and nx[0] - nx[7] > env.operator_dist) or
                   (nx[0] > 4.5 and nx[0] - pv[0] < 0 and nx[7] - nx[0] > env.operator_dist))
        if self.allowed and not self.y_b:
            if running:
                self.viol_y += 1
                if self.viol_y >= 1:
                    self.y_b, self.viol_y = True, 0
            if (not running) and self.viol_y > 0:
                self.viol_y = 0
        if self.y_b and self.allowed:
            if self.y_time >= 15:
                self.y_b, self.y_time = False, 0
            if ((nx[0] <= 4.5 and nx[0] - nx[7] <= 0.9 * env.operator_dist) or
                    (nx[0] > 4.5 and nx[7] - nx[0] <= 0.9 * env.operator_dist)):
                self.y_b, self.y_time = False, 0


def make_handover(env_name, env, has_backup):
    if not has_backup:
        return _Handover(env)          # NU / NP: one controller, every transition is kept (NU/main.py:36-90)
    return {"Unicycle": _UnicycleHandover, "SimulatedCars": _CarsHandover, "Pvtol": _PvtolHandover}[env_name](env)


def train(agent, env, dynamics_model, args, memory, node_memory, log=print, trace=None):
    """Returns a list of per-episode dicts (reward, length, safety violations, updates).  ``trace``: a list that
    receives one (use_backup, pushed_to_memory) pair per env step (driver tests)."""
    barrier = args.env.endswith("Barrier")
    pvtol = args.env.startswith("Pvtol")
    has_backup = getattr(agent, "backup_policy", None) is not None
    ho = make_handover(args.env, env, has_backup and not barrier)
    total_numsteps = updates = 0
    history = []
    for i_episode in range(args.max_episodes):
        ho.begin_episode(i_episode)
        episode_reward = episode_cost = episode_steps = 0
        done = False
        obs = env.reset()
        t0 = time.perf_counter()
        while not done:
            if len(memory) > args.batch_size:
                for _ in range(args.updates_per_step):
                    extra = (i_episode,) if pvtol else ()
                    agent.update_parameters(memory, args.batch_size, updates, dynamics_model, node_memory,
                                            args.NODE_model_update_interval, *extra)
                    updates += 1
            warm = args.start_steps > total_numsteps
            acting_backup = ho.use_backup
            if acting_backup:
                action = agent.select_action_backup(obs, warmup=warm)
                ho.on_backup_action()
            else:
                action = agent.select_action(obs, warmup=warm)
            out = env.step(action)
            next_obs, reward, constraint = out[:3]
            sig = (out[3],) if barrier else ()
            lya_in, next_lya_in, done, info = out[-4:]
            episode_steps += 1
            total_numsteps += 1
            episode_reward += reward
            episode_cost += (info.get('num_safety_violation', 0) + info.get('num_safety_violation_obstacles', 0) +
                             info.get('num_safety_violation_safety_operator', 0) +
                             info.get('num_safety_violation_y_min', 0) + info.get('num_safety_violation_y_max', 0))
            mask = 1 if episode_steps == env.max_episode_steps else float(not done)
            row = (obs, action, reward, constraint) + sig + (lya_in, next_lya_in, next_obs, mask)
            pushed = not acting_backup
            if pushed:
                k = episode_steps + ho.memory_t_shift
                memory.push(*row, t=k * env.dt, next_t=(k + 1) * env.dt)
            node_memory.push(*row, t=episode_steps * env.dt, next_t=(episode_steps + 1) * env.dt)
            ho.after_step(episode_steps, lya_in, next_lya_in, next_obs, info)
            if trace is not None:
                trace.append((bool(acting_backup), bool(pushed)))
            obs = next_obs
            if os.environ.get("NLBAC_TRAIN_CHECKSUM") == "2" and total_numsteps % 100 == 0:
                import torch
                torch.cuda.synchronize()
                log("  step %d updates %d r %.9g a %.9g checksum %s" % (total_numsteps, updates, episode_reward, float(np.sum(action)),
                    " ".join("%.17g" % float(a.theta.double().sum()) for a in agent.arenas)))
            if args.max_steps and total_numsteps >= args.max_steps:
                done = True
        if args.output and ((i_episode % max(1, int(args.max_episodes / 2)) == 0) or i_episode == args.max_episodes - 1):
            agent.save_model(args.output)
        rec = dict(episode=i_episode, reward=float(episode_reward), length=episode_steps, violations=int(episode_cost),
                   total_steps=total_numsteps, updates=updates, seconds=time.perf_counter() - t0)
        history.append(rec)
        log("episode %(episode)d  reward %(reward).2f  length %(length)d  safety violations %(violations)d  "
            "steps %(total_steps)d  updates %(updates)d  %(seconds).1f s" % rec)
        if os.environ.get("NLBAC_TRAIN_CHECKSUM"):      # run-to-run determinism check: parameter sums to the last bit
            import torch
            torch.cuda.synchronize()
            log("  checksum " + " ".join("%.17g" % float(a.theta.double().sum()) for a in agent.arenas))
        if args.max_steps and total_numsteps >= args.max_steps:
            break
    return history


def train_vectorized(agent, env, args, n_steps, log=print, memory=None, check=None):
    """Vectorised rollouts: ``env`` is one of ``nlbac_amd.envs.device`` — N environments stepping in
    lock step on the device — and nothing of a transition touches the host: the policy acts on the (N, obs) device
    tensor (one forward for all lanes), the simulator launch writes the next observations / rewards / constraints /
    Lyapunov inputs, the transition rows are assembled in the agent's minibatch layout on the device and appended to a
    ``DeviceReplayMemory`` (which serves both the controller updates and the NODE fit), ``updates_per_step`` updates
    follow every vector step.  The reference's driver is one host environment at a time (``*/main.py::train``,
    restated by ``train`` above and pinned against it); this is the same data flow widened to N lanes, WITHOUT the
    backup-controller hand-over heuristics (which are per-episode host control flow): the primary controller acts in
    every lane.  Time-limit terminations keep mask 1 (U/main.py:150), finished lanes are reset in place.
    ``check``: a callable(step_index, rows) the tests use to look at the rows that were appended.
    Returns dict(steps, updates, episodes, mean_return)."""
    import torch
    from .sac_cbf_clf.replay_memory import DeviceReplayMemory
    dev, lay, N = agent.device, agent.lay, env.n
    barrier = lay.sig is not None
    if memory is None:
        memory = DeviceReplayMemory(args.replay_size, args.seed, agent, device_rng=True)
    lo = torch.as_tensor(np.asarray(env.action_space.low), dtype=torch.float32, device=dev)
    hi = torch.as_tensor(np.asarray(env.action_space.high), dtype=torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(args.seed))
    rows = torch.zeros(N, lay.LD, dtype=torch.float32, device=dev)
    obs = env.reset().to(torch.float32).clone()
    ep_ret = torch.zeros(N, dtype=torch.float64, device=dev)
    ret_sum = torch.zeros((), dtype=torch.float64, device=dev)
    n_done = torch.zeros((), dtype=torch.int64, device=dev)
    # (does the task's NODE-fit schedule look at the episode count?  The Pvtol copy's does: tasks.PvtolTask.fit_due)
    from .sac_cbf_clf.tasks import _Task
    needs_episode = type(agent.task).fit_due is not _Task.fit_due
    steps = updates = it = 0
    if getattr(agent, "backup_policy", None) is not None:
        log("vectorised: the backup controller is trained by every update but never acts (the hand-over heuristics are "
            "per-episode host control flow): its replay distribution is the primary controller's, unlike the reference's")
    while steps < n_steps:
        if steps < args.start_steps:
            action = lo + (hi - lo) * torch.rand(N, lay.act_dim, generator=gen, device=dev)
        else:
            action = agent.policy.sample(obs)[0]
        out = env.step(action)
        nobs, reward, constraint = out[:3]
        lya_in, next_lya_in, done = out[-4], out[-3], out[-2]
        t = env.ep_step.to(torch.float32)                     # (already advanced by this step)
        timeout = env.ep_step >= int(env.max_episode_steps)
        rows.zero_()
        rows[:, lay.obs:lay.obs + lay.obs_dim] = obs
        rows[:, lay.act:lay.act + lay.act_dim] = action
        rows[:, lay.rew], rows[:, lay.con] = reward.float(), constraint.float()
        if barrier:
            rows[:, lay.sig] = out[3].float()
        rows[:, lay.lya:lay.lya + lay.lya_dim] = lya_in.float()
        rows[:, lay.nlya:lay.nlya + lay.lya_dim] = next_lya_in.float()
        rows[:, lay.nobs:lay.nobs + lay.obs_dim] = nobs.float()
        rows[:, lay.mask] = torch.where(timeout, torch.ones_like(reward), 1.0 - done.to(reward.dtype)).float()
        rows[:, lay.t], rows[:, lay.nt] = t * float(env.dt), (t + 1.0) * float(env.dt)
        memory.push_rows(rows)
        if check is not None:
            check(it, rows)
        ep_ret += reward
        fin = done > 0
        ret_sum += torch.where(fin, ep_ret, torch.zeros_like(ep_ret)).sum()
        n_done += fin.sum()
        ep_ret = torch.where(fin, torch.zeros_like(ep_ret), ep_ret)
        steps += N
        it += 1
        if len(memory) > args.batch_size:
            # the reference driver's trailing argument (P/main.py: the Pvtol copy stops fitting its NODE after episode
            # 100): lanes finish episodes on their own, so the counter is finished episodes per lane, 1-based like the
            # reference's.  Only a task whose fit schedule depends on it pays the scalar read-back (a host sync), and only
            # on the iterations whose update can fit the NODE at all.
            i_episode = None
            if needs_episode and any((updates + k) % args.NODE_model_update_interval == 0 for k in range(args.updates_per_step)):
                i_episode = 1 + int(n_done) // N
            for _ in range(args.updates_per_step):
                agent.update_parameters(memory, args.batch_size, updates, None, memory, args.NODE_model_update_interval,
                                        i_episode)
                updates += 1
        env.reset(fin)                                         # (finished lanes start over; the others keep their state)
        obs = env.obs.to(torch.float32).clone()
    torch.cuda.synchronize()
    eps = int(n_done)
    res = dict(steps=steps, updates=updates, episodes=eps, mean_return=float(ret_sum) / max(eps, 1))
    log("vectorised: %(steps)d env steps in %(episodes)d finished episodes, %(updates)d updates, mean return %(mean_return).2f" % res)
    return res


def main(argv=None):
    args = get_args(argv)
    import nlbac_amd  # noqa: F401
    from . import envs
    barrier = args.env.endswith("Barrier")
    if barrier:
        from .neural_barrier_certificate.sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
        from .neural_barrier_certificate.sac_cbf_clf.replay_memory import DeviceReplayMemory, ReplayMemory
    else:
        from .sac_cbf_clf.sac_cbf_clf import SAC_CBF_CLF
        from .sac_cbf_clf.replay_memory import DeviceReplayMemory, ReplayMemory
    from .sac_cbf_clf.dynamics import DynamicsModel
    env = envs.make(args.env, args.seed)
    if args.seed >= 0:       # U/main.py:253-258: every generator is seeded before the agent (and its networks) exist
        import random
        import torch
        env.seed(args.seed)
        random.seed(args.seed)
        env.action_space.seed(args.seed)
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    agent = SAC_CBF_CLF(env.observation_space.shape[0], env.action_space, env, args)
    agent.use_graphs = bool(args.hipgraphs)
    agent.solver = args.solver
    if args.vector_envs > 0:
        assert args.env != "SimulatedCars", "the vectorised driver covers the envs whose NODE takes no time input"
        from .envs import device as device_envs
        args.replay_size = min(args.replay_size, 1 << 20)
        return train_vectorized(agent, device_envs.make(args.env, args.vector_envs, seed=max(args.seed, 0)), args,
                                args.max_steps or 100000)
    dynamics_model = DynamicsModel(env, args)
    if args.device_replay:
        cap = min(args.replay_size, 1 << 20)
        memory = DeviceReplayMemory(cap, args.seed, agent, device_rng=args.device_rng)
        node_memory = DeviceReplayMemory(cap, args.seed + 1 if args.device_rng else args.seed, agent,
                                         device_rng=args.device_rng)
    else:
        memory, node_memory = ReplayMemory(args.replay_size, args.seed), ReplayMemory(args.replay_size, args.seed)
    return train(agent, env, dynamics_model, args, memory, node_memory)


if __name__ == "__main__":
    main()
