"""MPERunner for share_policy=False — API of `onpolicy/runner/separated/mpe_runner.py:13-268`: every agent has its own policy,
trainer and SeparatedReplayBuffer; an iteration is T x (collect, env.step, insert), compute, train, all per agent.

Data stay in HBM: `collect` runs each agent's actor + critic kernels on its buffer slot (outputs land in the slot,
R_MAPPOPolicy.collect_into), `insert` stores the agent's column of the env output.  The share_obs of an agent under
use_centralized_V is the concatenation of all agents' observations of the thread (separated/mpe_runner.py:84-96,160-166)."""
import time

import numpy as np
import torch

from .base_runner import Runner, _t2n, env_takes_device_actions


class MPERunner(Runner):
    def __init__(self, config):
        super().__init__(config)
        self._eye = None

    def run(self):
        self.warmup()
        start = time.time()
        episodes = int(self.num_env_steps) // self.episode_length // self.n_rollout_threads
        for episode in range(episodes):
            train_infos, _ = self.run_episode(episode, episodes)
            total_num_steps = (episode + 1) * self.episode_length * self.n_rollout_threads
            if episode % self.save_interval == 0 or episode == episodes - 1:
                if self.save_dir is not None:
                    self.save()
            if episode % self.log_interval == 0:
                end = time.time()
                print("\\n Scenario {} Algo {} Exp {} updates {}/{} episodes, total num timesteps {}/{}, FPS {}.\\n".format(
                    getattr(self.all_args, "scenario_name", "synthetic"), self.algorithm_name, self.experiment_name, episode, episodes,
                    total_num_steps, self.num_env_steps, int(total_num_steps / (end - start))))
                for agent_id in range(self.num_agents):
                    train_infos[agent_id]["average_episode_rewards"] = float(self.buffer[agent_id].rewards.mean().item()) * self.episode_length
                self.log_train(train_infos, total_num_steps)
            if episode % self.eval_interval == 0 and self.use_eval:
                self.eval(total_num_steps)

    def run_episode(self, episode=0, episodes=1):
        if self.use_linear_lr_decay:
            for agent_id in range(self.num_agents):
                self.trainer[agent_id].policy.lr_decay(episode, episodes)
        infos = None
        for step in range(self.episode_length):
            values, actions, action_log_probs, rnn_states, rnn_states_critic, actions_env = self.collect(step)
            obs, rewards, dones, infos = self.envs.step(actions_env)
            self.insert((obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic))
        self.compute()
        return self.train(), infos

    def _dev(self, x, dtype=torch.float32):
        if torch.is_tensor(x):
            return x.to(self.device) if x.dtype == torch.bool and dtype == torch.bool else x.to(self.device, dtype)
        return torch.as_tensor(np.asarray(x), dtype=dtype).to(self.device)

    def _share(self, obs_t):
        N = obs_t.shape[0]
        return obs_t.reshape(N, -1)                                 # all agents' obs of a thread, side by side

    # separated/mpe_runner.py:81-96
    def warmup(self):
        obs = self._dev(self.envs.reset())
        share = self._share(obs)
        for agent_id in range(self.num_agents):
            b = self.buffer[agent_id]
            b.share_obs[0].copy_(share if self.use_centralized_V else obs[:, agent_id])
            b.obs[0].copy_(obs[:, agent_id])

    # separated/mpe_runner.py:98-151
    @torch.no_grad()
    def collect(self, step):
        N = self.n_rollout_threads
        values, actions, logps, rnn_a, rnn_c = [], [], [], [], []
        for agent_id in range(self.num_agents):
            tr, b = self.trainer[agent_id], self.buffer[agent_id]
            tr.prep_rollout()
            act, ra, rc = tr.policy.collect_into(b, step)           # outputs land in b.{actions, action_log_probs, value_preds}[step]
            values.append(b.value_preds[step]); actions.append(act.view(N, 1)); logps.append(b.action_log_probs[step])
            rnn_a.append(ra); rnn_c.append(rc)
        actions = torch.stack(actions, dim=1)                       # [N, M, 1]
        n_act = self.envs.action_space[0].n
        if self._eye is None:
            self._eye = torch.eye(n_act, device=self.device)
        actions_env = self._eye[actions.view(N, self.num_agents).long()]          # np.eye(n)[action] per agent, [N, M, n]
        if not env_takes_device_actions(self.envs):
            actions_env = _t2n(actions_env)
        stack = lambda xs: torch.stack([x.view(N, *x.shape[1:]) if x is not None else torch.zeros(N, self.recurrent_N, self.hidden_size,
                                                                                                 device=self.device) for x in xs], dim=1)
        return torch.stack(values, dim=1), actions, torch.stack(logps, dim=1), stack(rnn_a), stack(rnn_c), actions_env

    # separated/mpe_runner.py:153-178
    def insert(self, data):
        obs, rewards, dones, infos, values, actions, action_log_probs, rnn_states, rnn_states_critic = data
        N = self.n_rollout_threads
        obs, rewards = self._dev(obs), self._dev(rewards)
        dones = self._dev(dones, torch.bool).view(N, self.num_agents)
        masks = (~dones).to(torch.float32).view(N, self.num_agents, 1)
        share = self._share(obs)
        recurrent = self.trainer[0]._use_recurrent_policy or self.trainer[0]._use_naive_recurrent
        for agent_id in range(self.num_agents):
            b = self.buffer[agent_id]
            ra = rc = None
            if recurrent:
                keep = masks[:, agent_id].view(N, 1, 1)
                ra = rnn_states[:, agent_id].reshape(N, self.recurrent_N, -1) * keep
                rc = rnn_states_critic[:, agent_id].reshape(N, self.recurrent_N, -1) * keep
            b.insert_env(share if self.use_centralized_V else obs[:, agent_id], obs[:, agent_id], rewards[:, agent_id].reshape(N, 1),
                         masks[:, agent_id], ra, rc)

    # separated/mpe_runner.py:180-236
    @torch.no_grad()
    def eval(self, total_num_steps):
        envs = self.eval_envs
        if envs is None:
            return
        obs = self._dev(envs.reset())
        N = obs.shape[0]
        rnn = [torch.zeros(N, self.recurrent_N, self.hidden_size, device=self.device) for _ in range(self.num_agents)]
        masks = torch.ones(N, self.num_agents, 1, device=self.device)
        eye = torch.eye(envs.action_space[0].n, device=self.device)
        rews = []
        for _ in range(self.episode_length):
            acts = []
            for agent_id in range(self.num_agents):
                self.trainer[agent_id].prep_rollout()
                a, rnn[agent_id] = self.trainer[agent_id].policy.act(obs[:, agent_id], rnn[agent_id], masks[:, agent_id], deterministic=True)
                acts.append(a.view(N))
            actions_env = eye[torch.stack(acts, dim=1)]
            if not env_takes_device_actions(envs):
                actions_env = _t2n(actions_env)
            obs, rewards, dones, _ = envs.step(actions_env)
            obs = self._dev(obs)
            dones = self._dev(dones, torch.bool).view(N, self.num_agents)
            rews.append(self._dev(rewards).view(N, self.num_agents))
            masks = (~dones).to(torch.float32).view(N, self.num_agents, 1)
            for agent_id in range(self.num_agents):
                rnn[agent_id] = rnn[agent_id].view(N, self.recurrent_N, -1) * masks[:, agent_id].view(N, 1, 1)
        rew = torch.stack(rews)                                       # [T, N, M]
        for agent_id in range(self.num_agents):
            avg = float(rew[:, :, agent_id].sum(0).mean().item())
            print("eval average episode rewards of agent%i: " % agent_id + str(avg))
            self.log_env({f"agent{agent_id}/eval_average_episode_rewards": [avg]}, total_num_steps)
