"""Flat parameter layout shared by the HIP kernels and the nn.Module views (include/mappo_hip.h,
`mappo_net_desc`): state_dict order of the reference minus the never-used `fc_h` template layer
(mlp.py:20-22), each network padded to a multiple of 256 floats inside the joint actor|critic buffer."""
SEG_ALIGN = 256


def net_layout(desc, head_prefix):
    """[(state_dict key, offset, shape)] for one network.  head_prefix: 'act.action_out.linear' | 'v_out'."""
    D, H, A = desc.in_dim, desc.hidden, desc.out_dim
    out, p = [], 0

    def add(key, *shape):
        nonlocal p
        n = 1
        for s in shape:
            n *= s
        out.append((key, p, tuple(shape)))
        p += n

    if desc.use_feature_norm:
        add("base.feature_norm.weight", D); add("base.feature_norm.bias", D)
    add("base.mlp.fc1.0.weight", H, D); add("base.mlp.fc1.0.bias", H)
    add("base.mlp.fc1.2.weight", H); add("base.mlp.fc1.2.bias", H)
    for l in range(desc.layer_N):
        add(f"base.mlp.fc2.{l}.0.weight", H, H); add(f"base.mlp.fc2.{l}.0.bias", H)
        add(f"base.mlp.fc2.{l}.2.weight", H); add(f"base.mlp.fc2.{l}.2.bias", H)
    if desc.recurrent:
        add("rnn.rnn.weight_ih_l0", 3 * H, H); add("rnn.rnn.weight_hh_l0", 3 * H, H)
        add("rnn.rnn.bias_ih_l0", 3 * H); add("rnn.rnn.bias_hh_l0", 3 * H)
        add("rnn.norm.weight", H); add("rnn.norm.bias", H)
    add(head_prefix + ".weight", A, H); add(head_prefix + ".bias", A)
    return out, p


def padded(n):
    return (n + SEG_ALIGN - 1) // SEG_ALIGN * SEG_ALIGN


def pack_state_dict(flat, layout, state_dict, base=0):
    """Copy the tensors of a (reference-keyed) state_dict into `flat` (1-D torch tensor)."""
    import torch
    for key, off, shape in layout:
        t = torch.as_tensor(state_dict[key], dtype=flat.dtype).reshape(-1)
        flat[base + off: base + off + t.numel()].copy_(t)


def unpack_to_dict(flat, layout, base=0):
    return {key: flat[base + off: base + off + _numel(shape)].reshape(shape) for key, off, shape in layout}


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n
