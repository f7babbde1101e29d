"""Runner for share_policy=False — API of `onpolicy/runner/separated/base_runner.py:14-160`: one R_MAPPOPolicy, R_MAPPO trainer
and SeparatedReplayBuffer PER AGENT (lists indexed by agent id), all on the HIP kernels of the shared path.  Every agent's
networks live in their own flat parameter buffer; `compute` / `train` run the agents one after the other (the same
launches as the shared runner with M = 1 — batching the agents into one launch is the next step)."""
import os

import torch

from mappo_amd.runner.shared.base_runner import Runner as _SharedRunner, _t2n, env_takes_device_actions  # noqa: F401
from mappo_amd.utils.separated_buffer import SeparatedReplayBuffer


class Runner(_SharedRunner):
    def __init__(self, config):
        self._separated_config = config
        super().__init__(config)

    def _build(self, config):
        """Called by the shared Runner's constructor in place of the single policy / trainer / buffer."""
        from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO as TrainAlgo
        from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy as Policy
        self.policy, self.trainer, self.buffer = [], [], []
        for agent_id in range(self.num_agents):
            share_space = self.envs.share_observation_space[agent_id] if self.use_centralized_V else self.envs.observation_space[agent_id]
            self.policy.append(Policy(self.all_args, self.envs.observation_space[agent_id], share_space, self.envs.action_space[agent_id],
                                      device=self.device))
            # every agent samples from its own Philox stream (same args.seed, same step counter and row index otherwise)
            self.policy[-1].actor._seed = (self.policy[-1].actor._seed + agent_id * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        if self.model_dir is not None:
            self.restore()
        for agent_id in range(self.num_agents):
            share_space = self.envs.share_observation_space[agent_id] if self.use_centralized_V else self.envs.observation_space[agent_id]
            self.trainer.append(TrainAlgo(self.all_args, self.policy[agent_id], device=self.device, dist_group=config.get("dist_group")))
            self.buffer.append(SeparatedReplayBuffer(self.all_args, self.envs.observation_space[agent_id], share_space,
                                                     self.envs.action_space[agent_id], device=self.device))

    # separated/base_runner.py:110-118
    @torch.no_grad()
    def compute(self):
        for agent_id in range(self.num_agents):
            tr, b = self.trainer[agent_id], self.buffer[agent_id]
            tr.prep_rollout()
            next_value = tr.policy.get_values(b.share_obs[-1], b.rnn_states_critic[-1], b.masks[-1])
            b.compute_returns(next_value, tr.value_normalizer)

    # separated/base_runner.py:120-129
    def train(self):
        train_infos = []
        for agent_id in range(self.num_agents):
            self.trainer[agent_id].prep_training()
            train_infos.append(self.trainer[agent_id].train(self.buffer[agent_id]))
            self.buffer[agent_id].after_update()
        return train_infos

    # separated/base_runner.py:131-146: actor_agent{i}.pt / critic_agent{i}.pt (+ vnorm_agent{i}.pt as the shared runner adds)
    def save(self):
        for agent_id in range(self.num_agents):
            pol, tr = self.policy[agent_id], self.trainer[agent_id]
            torch.save(pol.actor.state_dict(), os.path.join(self.save_dir, f"actor_agent{agent_id}.pt"))
            torch.save(pol.critic.state_dict(), os.path.join(self.save_dir, f"critic_agent{agent_id}.pt"))
            if tr.value_normalizer is not None:
                torch.save(tr.value_normalizer.state_dict(), os.path.join(self.save_dir, f"vnorm_agent{agent_id}.pt"))

    def restore(self):
        for agent_id in range(self.num_agents):
            pol = self.policy[agent_id]
            pol.actor.load_state_dict(torch.load(os.path.join(str(self.model_dir), f"actor_agent{agent_id}.pt"), weights_only=True))
            pol.critic.load_state_dict(torch.load(os.path.join(str(self.model_dir), f"critic_agent{agent_id}.pt"), weights_only=True))

    def restore_value_normalizers(self):
        for agent_id in range(self.num_agents):
            path = os.path.join(str(self.model_dir), f"vnorm_agent{agent_id}.pt")
            if self.trainer[agent_id].value_normalizer is not None and os.path.exists(path):
                self.trainer[agent_id].value_normalizer.load_state_dict(torch.load(path, weights_only=True))

    def log_train(self, train_infos, total_num_steps):
        for agent_id, info in enumerate(train_infos):
            super().log_train({f"agent{agent_id}/{k}": v for k, v in info.items()}, total_num_steps)
