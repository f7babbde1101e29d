"""Hyper-parameter surface of the hot path — same flag names, defaults and (inverted) store_false semantics
as the reference's `onpolicy/config.py:156-287`, so launch scripts and `all_args` namespaces carry over.
Flags this build adds are grouped at the end and all default to the reference's behaviour."""
import argparse


def get_config():
    p = argparse.ArgumentParser(description="mappo_amd", formatter_class=argparse.RawDescriptionHelpFormatter)
    off = dict(action="store_false", default=True)     # passing the flag turns the feature OFF (config.py quirk)
    on = dict(action="store_true", default=False)
    # prepare
    p.add_argument("--algorithm_name", type=str, default="mappo", choices=["rmappo", "mappo"])
    p.add_argument("--experiment_name", type=str, default="check")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--cuda", **off)
    p.add_argument("--cuda_deterministic", **off)
    p.add_argument("--n_training_threads", type=int, default=1)
    p.add_argument("--n_rollout_threads", type=int, default=32)
    p.add_argument("--n_eval_rollout_threads", type=int, default=1)
    p.add_argument("--n_render_rollout_threads", type=int, default=1)
    p.add_argument("--num_env_steps", type=int, default=10e6)
    p.add_argument("--user_name", type=str, default="marl")
    p.add_argument("--use_wandb", **off)
    # env
    p.add_argument("--env_name", type=str, default="StarCraft2")
    p.add_argument("--use_obs_instead_of_state", **on)
    # replay buffer
    p.add_argument("--episode_length", type=int, default=200)
    # network
    p.add_argument("--share_policy", **off)
    p.add_argument("--use_centralized_V", **off)
    p.add_argument("--stacked_frames", type=int, default=1)
    p.add_argument("--use_stacked_frames", **on)
    p.add_argument("--hidden_size", type=int, default=64)
    p.add_argument("--layer_N", type=int, default=1)
    p.add_argument("--use_ReLU", **off)
    p.add_argument("--use_popart", **on)
    p.add_argument("--use_valuenorm", **off)
    p.add_argument("--use_feature_normalization", **off)
    p.add_argument("--use_orthogonal", **off)
    p.add_argument("--gain", type=float, default=0.01)
    # recurrent
    p.add_argument("--use_naive_recurrent_policy", **on)
    p.add_argument("--use_recurrent_policy", **off)
    p.add_argument("--recurrent_N", type=int, default=1)
    p.add_argument("--data_chunk_length", type=int, default=10)
    # optimizer
    p.add_argument("--lr", type=float, default=5e-4)
    p.add_argument("--critic_lr", type=float, default=5e-4)
    p.add_argument("--opti_eps", type=float, default=1e-5)
    p.add_argument("--weight_decay", type=float, default=0)
    # ppo
    p.add_argument("--ppo_epoch", type=int, default=15)
    p.add_argument("--use_clipped_value_loss", **off)
    p.add_argument("--clip_param", type=float, default=0.2)
    p.add_argument("--num_mini_batch", type=int, default=1)
    p.add_argument("--entropy_coef", type=float, default=0.01)
    p.add_argument("--value_loss_coef", type=float, default=1)
    p.add_argument("--use_max_grad_norm", **off)
    p.add_argument("--max_grad_norm", type=float, default=10.0)
    p.add_argument("--use_gae", **off)
    p.add_argument("--gamma", type=float, default=0.99)
    p.add_argument("--gae_lambda", type=float, default=0.95)
    p.add_argument("--use_proper_time_limits", **on)
    p.add_argument("--use_huber_loss", **off)
    p.add_argument("--use_value_active_masks", **off)
    p.add_argument("--use_policy_active_masks", **off)
    p.add_argument("--huber_delta", type=float, default=10.0)
    # run / save / log / eval / render / pretrained
    p.add_argument("--use_linear_lr_decay", **on)
    p.add_argument("--save_interval", type=int, default=1)
    p.add_argument("--log_interval", type=int, default=5)
    p.add_argument("--use_eval", **on)
    p.add_argument("--eval_interval", type=int, default=25)
    p.add_argument("--eval_episodes", type=int, default=32)
    p.add_argument("--save_gifs", **on)
    p.add_argument("--use_render", **on)
    p.add_argument("--render_episodes", type=int, default=5)
    p.add_argument("--ifi", type=float, default=0.1)
    p.add_argument("--model_dir", type=str, default=None)
    # ---- additions of this build (defaults keep the reference's semantics) ----
    p.add_argument("--exact_minibatch_order", **on,
                   help="with num_mini_batch == 1 the minibatch is the whole buffer and its permutation only reorders "
                        "the terms of sums; by default the kernels then stream the buffer in place. Set to gather by "
                        "the random permutation anyway (what the reference's generator does).")
    p.add_argument("--unfused_update", **on,
                   help="run each PPO update as separate forward / fused-loss / backward launches (the standalone C-ABI ops) "
                        "instead of the one-launch-per-network update kernels; same results")
    p.add_argument("--use_hip_graph", **off,
                   help="by default the launch-bound inner loops (PPO updates; rollout with a graph-safe env) are captured "
                        "into hipGraphs after one eager pass; pass the flag to always launch eagerly")
    p.add_argument("--host_staging", **off,
                   help="by default the NumPy output of a CPU vec-env goes through pinned double-buffered staging blocks "
                        "(mappo_amd/utils/host_staging.py) and feeds the same one-launch rollout step as a device env; pass the "
                        "flag to upload with plain torch copies")
    p.add_argument("--fuse_rollout_step", **off,
                   help="by default an MLP policy's rollout step (insert of the previous env output + get_actions + get_values) "
                        "is ONE kernel launch (mappo_rollout_step); pass the flag to use the separate insert / actor / critic launches")
    p.add_argument("--dual_update", **off,
                   help="by default the actor's and the critic's fused update run in ONE launch, half the CUs each "
                        "(mappo_actor_critic_update); pass the flag to launch them one after the other")
    p.add_argument("--concurrent_update", **on,
                   help="launch the actor and critic update kernels side by side on a split grid (two streams); measured "
                        "slower than back-to-back full-size launches on MI355X, kept for experiments")
    p.add_argument("--perm_device", type=str, default="cuda", choices=["cuda", "cpu"],
                   help="where torch.randperm runs; 'cpu' reproduces the reference's permutation stream bit-exactly")
    return p


def mpe_defaults(parser):
    """train_mpe.py:52-61."""
    parser.add_argument("--scenario_name", type=str, default="simple_spread")
    parser.add_argument("--num_landmarks", type=int, default=3)
    parser.add_argument("--num_agents", type=int, default=2)
    return parser


def apply_algorithm_name(all_args):
    """train_mpe.py:68-80 / train_smac.py:77-86: rmappo = GRU policy, mappo = MLP policy."""
    if all_args.algorithm_name == "rmappo":
        all_args.use_recurrent_policy, all_args.use_naive_recurrent_policy = True, False
    elif all_args.algorithm_name == "mappo":
        all_args.use_recurrent_policy, all_args.use_naive_recurrent_policy = False, False
    else:
        raise NotImplementedError(all_args.algorithm_name)
    return all_args
