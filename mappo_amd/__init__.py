"""mappo_amd — MI355X-native MAPPO training hot path (rollout forward -> GAE -> PPO update) behind the
`onpolicy` Runner / R_MAPPOPolicy / R_MAPPO / SharedReplayBuffer API of Chen001117/mappo.

    from mappo_amd.runner.shared.mpe_runner import MPERunner
    from mappo_amd.algorithms.r_mappo.r_mappo import R_MAPPO
    from mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy import R_MAPPOPolicy
    from mappo_amd.utils.shared_buffer import SharedReplayBuffer

`mappo_amd.install_as_onpolicy()` registers these modules under the reference's import paths
(`onpolicy.utils.shared_buffer`, `onpolicy.algorithms.r_mappo.r_mappo`, ...), see INTEGRATION.md."""
import importlib
import sys
import types

_ALIASES = {
    "onpolicy.config": "mappo_amd.config",
    "onpolicy.utils.util": "mappo_amd.utils.util",
    "onpolicy.utils.valuenorm": "mappo_amd.utils.valuenorm",
    "onpolicy.utils.shared_buffer": "mappo_amd.utils.shared_buffer",
    "onpolicy.algorithms.r_mappo.r_mappo": "mappo_amd.algorithms.r_mappo.r_mappo",
    "onpolicy.algorithms.r_mappo.algorithm.rMAPPOPolicy": "mappo_amd.algorithms.r_mappo.algorithm.rMAPPOPolicy",
    "onpolicy.algorithms.r_mappo.algorithm.r_actor_critic": "mappo_amd.algorithms.r_mappo.algorithm.r_actor_critic",
    "onpolicy.runner.shared.base_runner": "mappo_amd.runner.shared.base_runner",
    "onpolicy.runner.shared.mpe_runner": "mappo_amd.runner.shared.mpe_runner",
    "onpolicy.runner.shared.smac_runner": "mappo_amd.runner.shared.smac_runner",
    "onpolicy.utils.separated_buffer": "mappo_amd.utils.separated_buffer",
    "onpolicy.runner.separated.base_runner": "mappo_amd.runner.separated.base_runner",
    "onpolicy.runner.separated.mpe_runner": "mappo_amd.runner.separated.mpe_runner",
}


def install_as_onpolicy(force=False):
    """Make `import onpolicy.<hot-path module>` resolve to this package (drop-in for the reference's runners
    and launch scripts).  Parent packages that are not importable are created as empty namespaces."""
    for alias, target in _ALIASES.items():
        if alias in sys.modules and not force:
            continue
        parts = alias.split(".")
        for i in range(1, len(parts)):
            pkg = ".".join(parts[:i])
            if pkg not in sys.modules:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
        mod = importlib.import_module(target)
        sys.modules[alias] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
