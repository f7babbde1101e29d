"""MPERunner — API of `onpolicy/runner/shared/mpe_runner.py:11-183` (training loop, warmup, collect, insert, eval).

The rollout step is fused: `collect` runs the actor (sampling + log-prob) and critic kernels on buffer slot
`step` and their outputs land directly in `buffer.{actions, action_log_probs, value_preds}[step]`
(R_MAPPOPolicy.collect_into); `insert` only stores what the environment produced.  Nothing crosses PCIe when
the vec-env yields device tensors (mappo_amd.envs.synthetic); NumPy-returning envs (the reference's
Dummy/SubprocVecEnv) work too — their arrays are uploaded by the buffer."""
import time

import numpy as np
import torch

from .base_runner import Runner, _t2n, env_takes_device_actions


class MPERunner(Runner):
    def __init__(self, config):
        super(MPERunner, self).__init__(config)
        self._onehot = None
        self._rollout_graph = None          # None -> "warm" -> CUDAGraph
        self._fuse_step = bool(getattr(self.all_args, "fuse_rollout_step", True))
        self._next_values = None
        # the rollout has no collective in it, so data-parallel ranks capture it too
        self._use_graph = bool(getattr(self.all_args, "use_hip_graph", True)) and bool(getattr(self.envs, "graph_safe", False))
        self._dist_present = config.get("dist_group") is not None
        # CPU vec-envs: pinned double-buffered staging, so that their NumPy output feeds the same one-launch step
        self._staging, self._eye_np = None, None
        if not env_takes_device_actions(self.envs) and bool(getattr(self.all_args, "host_staging", True)):
            from mappo_amd.utils.host_staging import HostStaging
            self._staging = HostStaging(self.device)

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads
        for episode in range(episodes):
            train_infos, infos = self.run_episode(episode, episodes)
            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads
            if episode % self.save_interval == 0 or episode == episodes - 1:
                self.save()
            if episode % self.log_interval == 0:
                end = time.time()
                print("\n Scenario {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\n".format(
                    getattr(self.all_args, "scenario_name", "synthetic"), self.algorithm_name, self.experiment_name,
                    episode, episodes, total_num_steps, self.num_env_steps, int(total_num_steps / (end - start))))
                env_infos = {}
                if self.env_name == "MPE" and infos is not None:
                    for agent_id in range(self.num_agents):
                        idv_rews = [info[agent_id]["individual_reward"] for info in infos
                                    if "individual_reward" in info[agent_id].keys()]
                        env_infos["agent%i/individual_rewards" % agent_id] = idv_rews
                train_infos["average_episode_rewards"] = float(self.buffer.rewards.mean().item()) * self.episode_length
                print("average episode rewards is {}".format(train_infos["average_episode_rewards"]))
                self.log_train(train_infos, total_num_steps)
                self.log_env(env_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def run_episode(self, episode=0, episodes=1):
        """One iteration of the hot loop (mpe_runner.py:22-40): T x (collect, env.step, insert), compute, train."""
        if self.use_linear_lr_decay:
            self.trainer.policy.lr_decay(episode, episodes)
        infos = self.rollout()
        return self.train(), infos

    def _rollout_body(self):
        infos = None
        self.trainer.policy.actor._counter_dev.add_(self.episode_length)   # fresh sampling stream per (replayed) episode
        fuse = self._fuse_step and self.trainer.policy.can_fuse_step()
        pending = None                     # env output of the previous step, not yet in the buffer (fused path)
        for step in range(self.episode_length):
            if fuse:
                # one launch: insert(step - 1) + collect(step) (mappo_rollout_step); same buffer contents as the plain loop
                actions = self.trainer.policy.collect_step_fused(self.buffer, step, pending, self.use_centralized_V)
                if actions is None:        # env output without the expected device layout: plain insert, then collect
                    self.insert(pending + (None,) * 6)
                    actions = self.trainer.policy.collect_step_fused(self.buffer, step, None, self.use_centralized_V)
                elif pending is not None:
                    self.buffer.step = step % self.episode_length
                if self._staging is not None:
                    self._staging.consumed()                # the launch above read the previous upload
                actions_env = self._actions_env(actions)
                obs, rewards, dones, infos = self.envs.step(actions_env)
                pending = self._stage(obs, rewards, dones)
                continue
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = self.collect(step)
            obs, rewards, dones, infos = self.envs.step(actions_env)
            obs, rewards, dones = self._stage(obs, rewards, dones)
            data = obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic
            self.insert(data)
            if self._staging is not None:
                self._staging.consumed()
        if fuse:
            # the last env output: its insert + the bootstrap value of compute() in one launch (critic workgroups + insert)
            b = self.buffer
            if self._next_values is None:
                self._next_values = torch.empty(b.n_rollout_threads * b.num_agents, device=b.device)
            nv = self.trainer.policy.collect_step_fused(b, self.episode_length, pending, self.use_centralized_V,
                                                        values_only=self._next_values)
            if nv is None:
                self.insert(pending + (None,) * 6)
                self.compute()
            else:
                b.step = 0
                self.trainer.prep_rollout()
                b.compute_returns(nv, self.trainer.value_normalizer)
            if self._staging is not None:
                self._staging.consumed()                      # AFTER the fallback insert / compute: they read `pending`, too
            return infos
        self.compute()
        return infos

    def rollout(self):
        """T x (collect, env.step, insert) + compute().  With a vec-env that declares `graph_safe` (device-resident,
        no host control flow that depends on data: mappo_amd.envs.synthetic) the episode is captured into one
        hipGraph after a first eager pass and replayed; any other env runs eagerly, step by step."""
        if not self._use_graph:
            return self._rollout_body()
        if self._rollout_graph is None:
            infos = self._rollout_body()
            self._rollout_graph = "warm"
            return infos
        if self._rollout_graph == "warm":
            torch.cuda.synchronize()
            mode = dict(capture_error_mode="thread_local") if self._dist_present else {}     # RCCL's watchdog thread stays legal
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

    # mpe_runner.py:81-93
    def warmup(self):
        obs = self.envs.reset()
        b = self.buffer
        obs_t = torch.as_tensor(obs, dtype=torch.float32).to(b.device)
        b.obs[0].copy_(obs_t)
        b.share_obs[0].copy_(self._share_obs(obs_t))

    def _share_obs(self, obs_t):
        if self.use_centralized_V:                      # mpe_runner.py:133-135: all agents' obs, repeated per agent
            N = obs_t.shape[0]
            return obs_t.reshape(N, 1, -1).expand(N, self.num_agents, -1)
        return obs_t

    # mpe_runner.py:95-123
    @torch.no_grad()
    def collect(self, step):
        self.trainer.prep_rollout()
        b = self.buffer
        actions, rnn_states, rnn_states_critic = self.trainer.policy.collect_into(b, step)
        actions_env = None
        if getattr(self.envs, "consumes_actions", True) and self._staging is not None:
            actions_env = self._host_onehot(actions)
        elif getattr(self.envs, "consumes_actions", True) and getattr(self.envs, "accepts_index_actions", False):
            actions_env = actions
        elif getattr(self.envs, "consumes_actions", True):      # synthetic envs ignore the actions: skip the one-hot
            if self._onehot is None:
                self._onehot = torch.eye(self.envs.action_space[0].n, device=b.device)
            actions_env = self._onehot[actions.view(b.n_rollout_threads, b.num_agents).long()]   # np.eye(n)[actions]
            if not env_takes_device_actions(self.envs):
                actions_env = self._host_actions(actions_env)
        return b.value_preds[step], actions, b.action_log_probs[step], rnn_states, rnn_states_critic, actions_env

    def _stage(self, obs, rewards, dones):
        """Env output -> what the fused step / insert kernels read.  Host arrays go through the pinned staging blocks."""
        if self._staging is None:
            return obs, rewards, dones
        d = self._staging.upload(obs=obs, rewards=rewards, dones=np.asarray(dones, dtype=np.bool_) if not torch.is_tensor(dones) else dones)
        return d["obs"], d["rewards"], d["dones"]

    def _host_actions(self, actions_env):
        if self._staging is not None:
            return self._staging.download("actions_env", actions_env)
        return _t2n(actions_env)

    def _host_onehot(self, actions):
        """One-hot env actions for a CPU env (mpe_runner.py:119, `np.eye(n)[actions]`): the integer actions come down through
        pinned memory (N x M floats instead of N x M x n) and the one-hot is built on the host, as the reference builds it."""
        b = self.buffer
        a = self._staging.download("actions", actions.view(b.n_rollout_threads, b.num_agents))
        if self._eye_np is None:
            self._eye_np = np.eye(self.envs.action_space[0].n, dtype=np.float32)
        return self._eye_np[a.astype(np.int64)]

    # mpe_runner.py:125-139
    def _actions_env(self, actions):
        b = self.buffer
        if not getattr(self.envs, "consumes_actions", True):     # synthetic envs ignore the actions: skip the one-hot
            return None
        if self._staging is not None:
            return self._host_onehot(actions)
        if getattr(self.envs, "accepts_index_actions", False):   # device env that decodes the buffer's action indices itself
            return actions
        if self._onehot is None:
            self._onehot = torch.eye(self.envs.action_space[0].n, device=b.device)
        actions_env = self._onehot[actions.view(b.n_rollout_threads, b.num_agents).long()]       # np.eye(n)[actions]
        if not env_takes_device_actions(self.envs):
            actions_env = self._host_actions(actions_env)
        return actions_env

    def insert(self, data):
        obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        b = self.buffer
        dev = b.device
        recurrent = self.trainer._use_recurrent_policy or self.trainer._use_naive_recurrent
        if b.insert_mpe_fused(obs, rewards, dones, self.use_centralized_V, *((rnn_states, rnn_states_critic) if recurrent else ())):
            return                                                                         # one kernel did it all
        obs_t = obs if (torch.is_tensor(obs) and obs.device == dev) else torch.as_tensor(obs, dtype=torch.float32).to(dev)
        dones_t = dones if (torch.is_tensor(dones) and dones.device == dev) else torch.as_tensor(dones).to(dev)
        masks = torch.logical_not(dones_t).view(b.n_rollout_threads, b.num_agents, 1)     # bool; cast by the slot copy
        rnn_a = rnn_c = None
        if self.trainer._use_recurrent_policy or self.trainer._use_naive_recurrent:
            keep = masks.view(b.n_rollout_threads, b.num_agents, 1, 1).to(torch.float32)
            rnn_a = rnn_states.view(b.n_rollout_threads, b.num_agents, b.recurrent_N, -1) * keep
            rnn_c = rnn_states_critic.view(b.n_rollout_threads, b.num_agents, b.recurrent_N, -1) * keep
        b.insert_env(self._share_obs(obs_t), obs_t, rewards, masks, rnn_a, rnn_c)

    @torch.no_grad()
    def eval(self, total_num_steps):
        """mpe_runner.py:141-183: deterministic `act` on the eval envs, average episode reward."""
        envs = self.eval_envs
        if envs is None:
            return
        eval_episode_rewards = []
        obs = torch.as_tensor(envs.reset(), dtype=torch.float32).to(self.device)
        N = obs.shape[0]
        R = N * self.num_agents
        rnn_states = torch.zeros(R, self.recurrent_N, self.hidden_size, device=self.device)
        masks = torch.ones(R, 1, device=self.device)
        eye = torch.eye(envs.action_space[0].n, device=self.device)
        for _ in range(self.episode_length):
            self.trainer.prep_rollout()
            action, rnn_states = self.trainer.policy.act(obs.reshape(R, -1), rnn_states, masks, deterministic=True)
            actions_env = eye[action.view(N, self.num_agents)]
            if not env_takes_device_actions(envs):
                actions_env = _t2n(actions_env)
            obs, rewards, dones, _ = envs.step(actions_env)
            obs = torch.as_tensor(obs, dtype=torch.float32).to(self.device)
            dones_t = torch.as_tensor(dones).to(self.device).view(R)
            eval_episode_rewards.append(torch.as_tensor(rewards, dtype=torch.float32).to(self.device))
            rnn_states = rnn_states * (~dones_t).view(R, 1, 1)
            masks = (~dones_t).to(torch.float32).view(R, 1)
        rew = torch.stack(eval_episode_rewards)
        info = {"eval_average_episode_rewards": [float(rew.sum(0).mean().item())]}
        print("eval average episode rewards of agent: " + str(info["eval_average_episode_rewards"][0]))
        self.log_env(info, total_num_steps)
