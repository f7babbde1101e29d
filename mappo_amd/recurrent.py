"""GRU (recurrent policy) path — `onpolicy/algorithms/utils/rnn.py:7-80` + the chunked generators.

Built on the GRU kernels of mappo_amd/csrc/gru.hip (single-step cell for rollouts, L-step masked sequence
with BPTT for training).  Until those entry points exist in libmappo_hip.so these functions refuse loudly."""


def _missing(*_a, **_k):
    raise NotImplementedError("recurrent (GRU) policies: the GRU kernels are not part of this build yet — "
                              "use algorithm_name=mappo (MLP policy); there is no torch fallback")


actor_step = actor_sequence_logits = critic_forward = train_recurrent = ppo_update_recurrent = _missing
