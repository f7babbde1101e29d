"""CPU test of the drop-in import paths: after mappo_amd.install_as_onpolicy() every `onpolicy.*` module of the
hot path resolves to this package's mirror and exposes the reference's class / function names.  (No GPU call is made: the
HIP library is loaded lazily by the first op.)"""
import importlib
import sys


def test_onpolicy_aliases_resolve():
    import mappo_amd
    for k in [k for k in sys.modules if k == "onpolicy" or k.startswith("onpolicy.")]:
        del sys.modules[k]
    mappo_amd.install_as_onpolicy()
    want = {
        "onpolicy.config": ["get_config"],
        "onpolicy.utils.util": ["check", "get_gard_norm", "update_linear_schedule", "huber_loss", "mse_loss", "get_shape_from_obs_space",
                                "get_shape_from_act_space"],
        "onpolicy.utils.valuenorm": ["ValueNorm"],
        "onpolicy.utils.shared_buffer": ["SharedReplayBuffer"],
        "onpolicy.algorithms.r_mappo.r_mappo": ["R_MAPPO"],
        "onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy": ["R_MAPPOPolicy"],
        "onpolicy.algorithms.r_mappo.algorithm.r_actor_critic": ["R_Actor", "R_Critic"],
        "onpolicy.runner.shared.base_runner": ["Runner"],
        "onpolicy.runner.shared.mpe_runner": ["MPERunner"],
        "onpolicy.runner.shared.smac_runner": ["SMACRunner"],
    }
    for alias, names in want.items():
        mod = importlib.import_module(alias)
        assert mod.__name__.startswith("mappo_amd."), (alias, mod.__name__)
        for n in names:
            assert hasattr(mod, n), f"{alias}.{n}"
    from onpolicy.runner.shared.mpe_runner import MPERunner as A
    from mappo_amd.runner.shared.mpe_runner import MPERunner as B
    assert A is B
    # constructor signatures of the reference: positional names as the reference's call sites use them
    import inspect
    from onpolicy.utils.shared_buffer import SharedReplayBuffer
    from onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from onpolicy.algorithms.r_mappo.r_mappo import R_MAPPO
    assert list(inspect.signature(SharedReplayBuffer.__init__).parameters)[1:6] == ["args", "num_agents", "obs_space", "cent_obs_space", "act_space"]
    assert list(inspect.signature(R_MAPPOPolicy.__init__).parameters)[1:6] == ["args", "obs_space", "cent_obs_space", "act_space", "device"]
    assert list(inspect.signature(R_MAPPO.__init__).parameters)[1:4] == ["args", "policy", "device"]


def test_sampling_seed_is_rank_keyed():
    """every rank gets the same args.seed (identical parameter init) but its own sampling stream."""
    from mappo_amd.distributed import sampling_seed
    seeds = [sampling_seed(1, r) for r in range(8)]
    assert len(set(seeds)) == 8 and seeds[0] == 1
    assert all(0 <= s < 2 ** 64 for s in seeds)
    assert sampling_seed(1, 3) == sampling_seed(1, 3)
