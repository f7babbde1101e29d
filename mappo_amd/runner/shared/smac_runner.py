"""SMACRunner — API of `onpolicy/runner/shared/smac_runner.py:11-214` (training loop, warmup, collect, insert,
log_train, eval).  SMAC-style vec-env contract (env_wrappers.py:363-379): `reset() -> (obs, share_obs, avail)`,
`step(actions [N, M, 1]) -> (obs, share_obs, rewards, dones [N, M], infos, avail)` with
`infos[i][m]['bad_transition']`.  The rollout step is fused like MPERunner's: the policy kernels write actions /
log-probs / values into buffer slot `step`, `insert` stores what the environment returned and derives masks,
active_masks and bad_masks exactly as smac_runner.py:129-151."""
import time
from functools import reduce

import numpy as np
import torch

from .base_runner import Runner, _t2n, env_takes_device_actions


class SMACRunner(Runner):
    def __init__(self, config):
        super(SMACRunner, self).__init__(config)
        self._rollout_graph = None          # None -> "warm" -> CUDAGraph
        # a device-resident env without data-dependent host control flow (mappo_amd.envs.synthetic) lets the whole episode
        # (T x collect / env.step / insert + compute) replay as one hipGraph, as in MPERunner.rollout
        # (data-parallel runs, too: there is no collective inside the rollout; the capture uses thread-local error mode so that
        # RCCL's watchdog thread stays legal, and falls back to eager launches if it fails — as MPERunner.rollout)
        self._use_graph = bool(getattr(self.all_args, "use_hip_graph", True)) and bool(getattr(self.envs, "graph_safe", False))
        self._dist_present = config.get("dist_group") is not None
        self._fuse_insert = bool(getattr(self.all_args, "fuse_rollout_step", True))

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads
        last_battles_game = np.zeros(self.n_rollout_threads, dtype=np.float32)
        last_battles_won = np.zeros(self.n_rollout_threads, dtype=np.float32)
        for episode in range(episodes):
            train_infos, infos = self.run_episode(episode, episodes)
            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0:
                end = time.time()
                print("\n Map {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n".format(
                    getattr(self.all_args, "map_name", "synthetic"), self.algorithm_name, self.experiment_name, episode, episodes,
                    total_num_steps, self.num_env_steps, int(total_num_steps / (end - start))))
                if self.env_name == "StarCraft2" and infos is not None:
                    battles_won, battles_game, incre_won, incre_game = [], [], [], []
                    for i, info in enumerate(infos):
                        if "battles_won" in info[0].keys():
                            battles_won.append(info[0]["battles_won"]); incre_won.append(info[0]["battles_won"] - last_battles_won[i])
                        if "battles_game" in info[0].keys():
                            battles_game.append(info[0]["battles_game"]); incre_game.append(info[0]["battles_game"] - last_battles_game[i])
                    incre_win_rate = np.sum(incre_won) / np.sum(incre_game) if np.sum(incre_game) > 0 else 0.0
                    print("incre win rate is {}.".format(incre_win_rate))
                    self.log_env({"incre_win_rate": [incre_win_rate]}, total_num_steps)
                    if battles_game:
                        last_battles_game, last_battles_won = battles_game, battles_won
                shape = list(self.buffer.active_masks.shape)
                train_infos["dead_ratio"] = 1 - float(self.buffer.active_masks.sum().item()) / reduce(lambda x, y: x * y, shape)
                self.log_train(train_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def run_episode(self, episode=0, episodes=1):
        if self.use_linear_lr_decay:
            self.trainer.policy.lr_decay(episode, episodes)
        infos = self.rollout()
        return self.train(), infos

    def _rollout_body(self):
        infos = None
        pol, b = self.trainer.policy, self.buffer
        pol.actor._counter_dev.add_(self.episode_length)   # fresh sampling stream per (replayed) episode
        # device envs + recurrent policies: the insert of step k - 1's env output rides in step k's launch (mappo_recurrent_rollout_step)
        fuse = (self._fuse_insert and env_takes_device_actions(self.envs)
                and pol.can_fuse_recurrent_step(b.n_rollout_threads * b.num_agents))
        pending = None
        for step in range(self.episode_length):
            out = None
            if pending is not None:
                self.trainer.prep_rollout()
                out = pol.collect_step_fused_recurrent(b, step, pending)
                if out is None:                                 # not the layout the fused launch reads: plain insert, then collect
                    self._insert_pending(pending)
            if out is not None:
                actions, rnn_states, rnn_states_critic = out
                values, action_log_probs = b.value_preds[step], b.action_log_probs[step]
            else:
                values, actions, action_log_probs, rnn_states, rnn_states_critic = self.collect(step)
            obs, share_obs, rewards, dones, infos, available_actions = self.envs.step(actions)
            if fuse and (infos is None or isinstance(infos, torch.Tensor)):
                pending = (obs, share_obs if self.use_centralized_V else obs, rewards, dones, infos, available_actions, rnn_states,
                           rnn_states_critic)
            else:
                pending = None
                self.insert((obs, share_obs, rewards, dones, infos, available_actions, values, actions, action_log_probs,
                             rnn_states, rnn_states_critic))
        if pending is not None:
            self._insert_pending(pending)                       # the last env output: slot T (read by compute())
        self.compute()
        return infos

    def _insert_pending(self, pending):
        obs, share_obs, rewards, dones, infos, available_actions, rnn_states, rnn_states_critic = pending
        self.insert((obs, share_obs, rewards, dones, infos, available_actions, None, None, None, rnn_states, rnn_states_critic))

    def rollout(self):
        if not self._use_graph:
            return self._rollout_body()
        if self._rollout_graph is None:
            infos = self._rollout_body()
            self._rollout_graph = "warm"
            return infos
        if self._rollout_graph == "warm":
            torch.cuda.synchronize()
            mode = dict(capture_error_mode="thread_local") if self._dist_present else {}
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, **mode):
                    self._rollout_body()
                self._rollout_graph = g
            except Exception as e:
                if not self._dist_present:
                    raise
                import warnings                              # never let a failed capture take a multi-GPU run down
                warnings.warn(f"hipGraph capture of the rollout failed ({e}); launching eagerly")
                torch.cuda.synchronize()
                self._use_graph = False
                return self._rollout_body()
        self._rollout_graph.replay()
        return None

    # smac_runner.py:98-108
    def warmup(self):
        obs, share_obs, available_actions = self.envs.reset()
        if not self.use_centralized_V:
            share_obs = obs
        b = self.buffer
        b._put(b.share_obs[0], share_obs)
        b._put(b.obs[0], obs)
        b._put(b.available_actions[0], available_actions)

    # smac_runner.py:110-127
    @torch.no_grad()
    def collect(self, step):
        self.trainer.prep_rollout()
        b = self.buffer
        actions, rnn_states, rnn_states_critic = self.trainer.policy.collect_into(b, step, use_available_actions=True)
        if not env_takes_device_actions(self.envs):
            actions = _t2n(actions)
        return b.value_preds[step], actions, b.action_log_probs[step], rnn_states, rnn_states_critic

    # smac_runner.py:129-151
    def insert(self, data):
        obs, share_obs, rewards, dones, infos, available_actions, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        b = self.buffer
        dev, N, Ma = b.device, b.n_rollout_threads, b.num_agents
        recurrent = self.trainer._use_recurrent_policy or self.trainer._use_naive_recurrent
        if (infos is None or isinstance(infos, torch.Tensor)) and b.insert_smac_fused(
                share_obs if self.use_centralized_V else obs, obs, rewards, dones, infos, available_actions,
                *((rnn_states, rnn_states_critic) if recurrent else ())):
            return                                                                           # device env: one kernel did it all
        dones_t = torch.as_tensor(dones).to(dev).view(N, Ma)
        dones_env = dones_t.all(dim=1)                                                       # :132
        keep_env = (~dones_env).to(torch.float32)
        masks = keep_env.view(N, 1, 1).expand(N, Ma, 1)                                      # :137-138
        active_masks = torch.where(dones_env.view(N, 1), torch.ones((), device=dev),
                                   (~dones_t).to(torch.float32)).view(N, Ma, 1)              # :140-142
        if isinstance(infos, torch.Tensor):                                                  # device env: bad_transition [N, M] bool
            bad_masks = (~infos.to(dev)).to(torch.float32).view(N, Ma, 1)
        else:                                                                                # :144
            bad_masks = torch.tensor([[[0.0] if info[agent_id]["bad_transition"] else [1.0] for agent_id in range(Ma)]
                                      for info in infos], dtype=torch.float32, device=dev)
        rnn_a = rnn_c = None
        if self.trainer._use_recurrent_policy or self.trainer._use_naive_recurrent:
            k = keep_env.view(N, 1, 1, 1)
            rnn_a = rnn_states.view(N, Ma, b.recurrent_N, -1) * k                           # :134-135
            rnn_c = rnn_states_critic.view(N, Ma, b.recurrent_N, -1) * k
        if not self.use_centralized_V:
            share_obs = obs
        b.insert_env(share_obs, obs, rewards, masks, rnn_a, rnn_c, bad_masks, active_masks, available_actions)

    # smac_runner.py:160-214
    @torch.no_grad()
    def eval(self, total_num_steps):
        """Deterministic `act` on the eval envs until `eval_episodes` episodes have ended: per-episode reward sums
        (`eval_average_episode_rewards`) and the win rate from `infos[i][0]['won']`.  rnn states and masks of an env are reset
        when all of its agents report done, as in the training rollout.  Returns the win rate."""
        envs = self.eval_envs
        if envs is None:
            return None
        dev = self.device
        N, Ma = self.n_eval_rollout_threads, self.num_agents
        R = N * Ma
        f32 = lambda x: x.to(dev, torch.float32) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x), dtype=torch.float32).to(dev)
        obs, share_obs, avail = envs.reset()
        rnn_states = torch.zeros(R, self.recurrent_N, self.hidden_size, device=dev)
        masks = torch.ones(R, 1, device=dev)
        battles_won, episodes = 0, 0
        # per-env running sums (the reference keeps ONE list for all eval threads and clears it whenever any of them ends,
        # smac_runner.py:186,199-200: with more than one eval thread that mixes episodes and, on current numpy, breaks when two
        # end in the same step; the per-env sum is what `eval_average_episode_rewards` is meant to average)
        episode_rewards, running = [], np.zeros((N, Ma, 1), dtype=np.float32)
        n_eval = int(getattr(self.all_args, "eval_episodes", 32))
        win_rate = 0.0
        while True:
            self.trainer.prep_rollout()
            actions, rnn_states = self.trainer.policy.act(f32(obs).reshape(R, -1), rnn_states, masks, f32(avail).reshape(R, -1),
                                                          deterministic=True)
            actions = actions.view(N, Ma, -1)
            if not env_takes_device_actions(envs):
                actions = _t2n(actions)
            obs, share_obs, rewards, dones, infos, avail = envs.step(actions)
            running += _t2n(f32(rewards)).reshape(N, Ma, 1)
            dones_env = torch.as_tensor(dones).to(dev).view(N, Ma).all(dim=1)                # :190
            keep = (~dones_env).to(torch.float32).view(N, 1).expand(N, Ma).reshape(R)
            rnn_states = rnn_states.view(R, self.recurrent_N, -1) * keep.view(R, 1, 1)       # :192
            masks = keep.view(R, 1).clone()                                                  # :194-195
            for i in np.nonzero(_t2n(dones_env))[0]:                                          # :197-203
                episodes += 1
                episode_rewards.append(running[i].copy())
                running[i] = 0.0
                if not torch.is_tensor(infos) and infos is not None and infos[i][0].get("won", False):
                    battles_won += 1
            if episodes >= n_eval:                                                           # :205-214
                self.log_env({"eval_average_episode_rewards": np.array(episode_rewards)}, total_num_steps)
                win_rate = battles_won / episodes
                print("eval win rate is {}.".format(win_rate))
                if self.use_wandb:
                    import wandb
                    wandb.log({"eval_win_rate": win_rate}, step=total_num_steps)
                elif self.writter is not None:
                    self.writter.add_scalars("eval_win_rate", {"eval_win_rate": win_rate}, total_num_steps)
                break
        return win_rate

    def log_train(self, train_infos, total_num_steps):
        train_infos["average_step_rewards"] = float(self.buffer.rewards.mean().item())       # :154
        super().log_train(train_infos, total_num_steps)
