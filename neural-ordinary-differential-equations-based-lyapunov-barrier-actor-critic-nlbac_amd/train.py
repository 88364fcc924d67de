"""Reference-shaped training driver: the loop of ``U/main.py:20-165`` — warm-up with random
actions, one ``update_parameters`` per ``updates_per_step`` once the replay holds a batch, transitions pushed to the
controller replay and the NODE replay, the stuck-detection heuristic that hands control to the backup controller
(``U/main.py:108-142``) — on the gym-free simulators of ``nlbac_amd.envs`` and the MI355X agent.  Logging is one
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
    return p.parse_args(argv)


def train(agent, env, dynamics_model, args, memory, node_memory, log=print):
    """Returns a list of per-episode dicts (reward, length, safety violations, updates)."""
    barrier = args.env.endswith("Barrier")
    pvtol = args.env.startswith("Pvtol")
    unicycle = args.env.startswith("Unicycle")
    has_backup = getattr(agent, "backup_policy", None) is not None
    total_numsteps = updates = 0
    start_using_backup = False
    history = []
    for i_episode in range(args.max_episodes):
        use_backup = False
        if i_episode > 3:
            start_using_backup = has_backup and unicycle      # the stuck heuristic below is the Unicycle driver's
        positions_record = []
        backup_time = violation_time = 0
        episode_reward = episode_cost = episode_steps = 0
        x_init_diff = y_init_diff = 0.0
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
            if use_backup and start_using_backup:
                action = agent.select_action_backup(obs, warmup=warm)
                backup_time += 1
            else:
                action = agent.select_action(obs, warmup=warm)
            out = env.step(action)
            next_obs, reward, constraint = out[:3]
            sig = (out[3],) if barrier else ()
            lya_in, next_lya_in, done, info = out[-4:]
            episode_steps += 1
            total_numsteps += 1
            episode_reward += reward
            episode_cost += info.get('num_safety_violation', 0) + info.get('num_safety_violation_obstacles', 0)
            mask = 1 if episode_steps == env.max_episode_steps else float(not done)
            row = (obs, action, reward, constraint) + sig + (lya_in, next_lya_in, next_obs, mask)
            tt = dict(t=episode_steps * env.dt, next_t=(episode_steps + 1) * env.dt)
            if not (start_using_backup and use_backup):
                memory.push(*row, **tt)
            node_memory.push(*row, **tt)
            if unicycle:          # U/main.py:108-142: stuck for 8 checks -> backup controller for up to 30 steps
                positions_record.append(np.asarray(next_lya_in))
                if episode_steps >= 50:
                    diff = positions_record[-1] - positions_record[-40]
                    moved = float(diff[0] * diff[0] + diff[1] * diff[1])
                    if start_using_backup and not use_backup:
                        if moved <= 0.01:
                            violation_time += 1
                            if violation_time >= 8:
                                use_backup, violation_time = True, 0
                                x_init_diff, y_init_diff = next_lya_in[0], next_lya_in[1]
                        elif violation_time > 0:
                            violation_time = 0
                    if use_backup and start_using_backup:
                        if backup_time >= 30:
                            use_backup, backup_time = False, 0
                        dx, dy = next_lya_in[0] - x_init_diff, next_lya_in[1] - y_init_diff
                        if dx * dx + dy * dy >= 0.6:
                            use_backup, backup_time = False, 0
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
