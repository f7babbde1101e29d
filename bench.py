#!/usr/bin/env python3
"""bench.py — agent-steps/sec of one full MAPPO iteration (collect + GAE + PPO) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`,
WORLD_SIZE set) this process is one of the ranks; started bare (`python bench.py --gpus N`) it spawns that launcher itself as a
child process and relays rank 0's JSON line (launch_ranks).

A "step" is one training iteration of BASELINE.json configs[1]: episode_length=25 rollout steps over
n_rollout_threads=1024 (--scaling strong, the default: 1024 threads IN TOTAL, sharded over the ranks as north_star and
BASELINE configs[3]/[4] state it; --scaling weak: 1024 threads PER GPU) x 3 agents of MPE
simple_spread-shaped synthetic observations (obs 18, share_obs 54, Discrete(5)) with the fused actor/critic
kernels writing into the HBM replay buffer, the bootstrap value + GAE scan, ppo_epoch=10 x num_mini_batch=1
PPO updates (forward, fused loss, backward, [RCCL all-reduce], clip + Adam) and after_update.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     : the dominant kernel of the step — mlp_update16_dual_kernel: forward + PPO / value loss + backward of the
                 actor AND critic MLPs in one launch, bound by the fp32 MFMA rate; algorithmic flops 6*MACs/sample
                 (forward, dW, dX) / launch duration from HIP events attached to the dispatch.  `traffic` = HBM bytes per
                 launch from the committed rocprofv3 PMC passes over this very kernel (profiles/r03/dual_update_hbm_pmc.json,
                 keyed by a hash of the kernel's sources: a stale measurement is reported as null).
  gae_roofline : the GAE scan kernel at BASELINE configs[4] per-GPU size (T=400, R=16 384: 105 MB), HBM-bound.
  ppo_loss_roofline : the standalone fused PPO loss kernel (mappo_ppo_loss_fwd_bwd, north_star's HBM-roofline kernel)
                 on this step's buffer and at BASELINE configs[4] size: 4*(3A+8) B/sample / launch duration.
  cpu_baseline : the CPU oracle (a port of the reference's NumPy/torch-CPU path, oracle/mappo_oracle.py) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N == 1 only).
  kernels      : mean device time of the other hot kernels from the same HIP-event hook (informative).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, dense fp32 matrix peak
HBM_PMC_FILE = os.path.join(ROOT, "profiles", "r03", "dual_update_hbm_pmc.json")   # scripts/pmc_hbm.sh: FETCH_SIZE / WRITE_SIZE passes


def kernel_source_sha256():
    """Hash of the sources the dominant kernel is compiled from (same recipe as scripts/pmc_hbm.sh): the committed PMC traffic is
    only reported while it describes the kernel that is actually running."""
    import hashlib
    h = hashlib.sha256()
    for f in ("mlp_upd16.h", "mlp_core.h", "common.h"):
        with open(os.path.join(ROOT, "mappo_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def measured_traffic(kernel, samples):
    """HBM bytes per launch of `kernel` from the committed PMC summary (separate rocprofv3 --pmc passes, as the microarch
    guide prescribes: FETCH_SIZE doubled for 16-B-per-lane streaming reads on gfx950, WRITE_SIZE as reported), or None
    when there is no measurement of this kernel at this size."""
    try:
        with open(HBM_PMC_FILE) as f:
            m = json.load(f)
    except (OSError, ValueError):
        return None, "no PMC summary committed"
    if m.get("kernel") != kernel or int(m.get("samples", -1)) != int(samples):
        return None, f"PMC summary is for {m.get('kernel')} at {m.get('samples')} samples"
    try:
        if m.get("source_sha256") != kernel_source_sha256():
            return None, "PMC summary is stale: the kernel's sources changed since scripts/pmc_hbm.sh measured it"
    except OSError:
        return None, "kernel sources not found: cannot tell whether the PMC summary is current"
    return int(m["traffic_bytes"]), m.get("note", "")


def ppo_loss_roofline(runner, timer, A):
    """The standalone fused PPO loss kernel (mappo_ppo_loss_fwd_bwd) with HIP events attached to its dispatch:
    (a) on this run's replay buffer (one launch over the whole buffer, as --unfused_update issues it),
    (b) on synthetic inputs of BASELINE configs[4] per-GPU size (T=400 x N=256 x M=64 = 6 553 600 samples)."""
    from mappo_amd import ops
    tr, b = runner.trainer, runner.buffer
    cfg = tr._cfg
    out = {}
    dev = b.device

    def run(tag, B, logits, values, avail, actions, oldlp, adv, active, vold, ret, vn):
        mom = torch.zeros(4, dtype=torch.float64, device=dev)
        ops.minibatch_moments(ret, active, None, B, mom)
        dl, dv = torch.empty(B, A, device=dev), torch.empty(B, device=dev)
        st = torch.zeros(6, dtype=torch.float64, device=dev)
        timer.reset(); timer.active = {"ppo_loss"}
        if len(timer.pool) < 12:
            timer.reserve(12)
        for _ in range(10):
            ops.ppo_loss_fwd_bwd(logits, values, None, avail, actions, oldlp, adv, active, vold, ret, vn, mom, dl, dv, st, cfg)
        torch.cuda.synchronize()
        us = timer.mean_us("ppo_loss")
        timer.active = set()
        nbytes = B * 4 * (3 * A + 8)
        ach = nbytes / (us * 1e-6) / 1e9
        out[tag] = dict(bound="hbm", kernel="ppo_loss_kernel", samples=B, bytes_per_launch=nbytes, launch_us=us, achieved=ach,
                        peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)

    T = b.episode_length
    S = T * b.n_rollout_threads * b.num_agents
    adv = tr.compute_advantages(b)
    run("this_buffer", S, torch.randn(S, A, device=dev), b.value_preds[:T].reshape(S).clone(), b.available_actions[:T].reshape(S, A),
        b.actions.view(S), b.action_log_probs.view(S), adv, b.active_masks[:T].view(S), b.value_preds[:T].view(S),
        b.returns[:T].view(S), tr.value_normalizer.state)
    Bc5 = 400 * 256 * 64
    g = lambda *s: torch.randn(*s, device=dev)
    run("configs4_size", Bc5, g(Bc5, A), g(Bc5), torch.ones(Bc5, A, device=dev), torch.randint(0, A, (Bc5,), device=dev).float(),
        -torch.rand(Bc5, device=dev) - 1.0, g(Bc5), torch.ones(Bc5, device=dev), g(Bc5), g(Bc5), tr.value_normalizer.state)
    return out


def gae_roofline(runner, timer):
    """The GAE scan kernel at BASELINE configs[4] per-GPU size (T=400, R=256 x 64 = 16 384 series): 16 B per agent-step
    (read r, v, mask; write ret) + 8 R for the slot-T edge = 105 MB per launch, HIP events on the dispatch."""
    from mappo_amd import ops
    dev = runner.buffer.device
    T, R = 400, 256 * 64
    g = lambda *s: torch.randn(*s, device=dev)
    rew, val, ret = g(T, R), g(T + 1, R), torch.empty(T + 1, R, device=dev)
    masks = (torch.rand(T + 1, R, device=dev) > 0.02).float()
    vn = runner.trainer.value_normalizer.state if runner.trainer.value_normalizer is not None else None
    timer.reset(); timer.active = {"gae_scan"}
    if len(timer.pool) < 12:
        timer.reserve(12)
    for _ in range(10):
        ops.gae_scan(rew, val, val[T].clone(), masks, None, ret, vn, 0.99, 0.95)
    torch.cuda.synchronize()
    us = timer.mean_us("gae_scan")
    timer.active = set()
    nbytes = 16 * T * R + 8 * R
    ach = nbytes / (us * 1e-6) / 1e9
    return dict(bound="hbm", kernel="gae_scan_kernel", agent_steps=T * R, bytes_per_launch=nbytes, launch_us=us, achieved=ach,
                peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c2", "c4"), default="c2",
                    help="c2 = BASELINE configs[1], the configuration the metric is quoted on (default); c4 = configs[3], the SMAC MMM2 shape "
                         "(10 agents, obs 176 / state 322 / 18 actions, GRU policy, T=400, 512 rollout threads in total, ppo_epoch 5 x 2 "
                         "minibatches) that SURVEY 8(e) names for the strong-scaling target — no roofline / cpu_baseline objects there")
    ap.add_argument("--n_rollout_threads", type=int, default=None, help="global under --scaling strong, per GPU under weak (default 1024; c4: 512)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): n_rollout_threads is the GLOBAL count, sharded over the ranks (north_star's >= 6x at "
                         "8 GPUs is a strong-scaling target); weak: n_rollout_threads per GPU")
    ap.add_argument("--env", choices=("synthetic", "mpe"), default="synthetic",
                    help="synthetic (default, BASELINE.json: MPE-shaped N(0,1) observations) | mpe: the GPU-vectorised simple_spread "
                         "environment (mappo_amd/envs/mpe_spread.py: real dynamics, rewards and resets, still no process boundary)")
    ap.add_argument("--episode_length", type=int, default=None)
    ap.add_argument("--ppo_epoch", type=int, default=None)
    ap.add_argument("--num_mini_batch", type=int, default=None)
    ap.add_argument("--exact_minibatch_order", action="store_true")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_companion", action="store_true", help="N > 1: skip the weak-scaling companion measurement")
    ap.add_argument("--cpu_threads", type=int, default=0, help="0 = all cores of this box (max 16)")
    ap.add_argument("--dry_launch", action="store_true",
                    help="launcher rehearsal without a GPU: every rank prints its RANK / WORLD_SIZE / LOCAL_RANK and exits (CPU test of --gpus N)")
    ns = ap.parse_args()
    dflt = dict(c2=(1024, 25, 10, 1), c4=(512, 400, 5, 2))[ns.config]
    for k, v in zip(("n_rollout_threads", "episode_length", "ppo_epoch", "num_mini_batch"), dflt):
        if getattr(ns, k) is None:
            setattr(ns, k, v)
    if ns.config == "c4" and ns.steps == 20 and ns.warmup == 3:
        ns.steps, ns.warmup = 3, 2                            # an iteration is ~50 ms x 400 steps of rollout: keep the default run short
    return ns


def make_args(ns):
    from mappo_amd.config import get_config
    a = get_config().parse_known_args([])[0]
    a.algorithm_name = "mappo"
    a.use_recurrent_policy = False
    a.use_naive_recurrent_policy = False
    a.env_name = "MPE"
    a.episode_length = ns.episode_length
    a.n_rollout_threads = ns.local_threads
    a.ppo_epoch = ns.ppo_epoch
    a.num_mini_batch = ns.num_mini_batch
    a.lr = a.critic_lr = 7e-4                         # train_mpe_spread.sh:15-17
    a.exact_minibatch_order = ns.exact_minibatch_order
    a.seed = 1
    return a


class KernelTimer:
    """Arms the C-side HIP-event hook (mappo_profile_arm) before calls of profiled ops.  Events are created
    up front so that arming inside the timed region costs one ctypes call."""

    def __init__(self):
        self.active = set()
        self.pool, self.used = [], {}

    def reserve(self, n):
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); e.record()                      # forces creation of the hipEvent handles
            self.pool.append((s, e))

    def arm(self, ops, kernel, label):
        if label not in self.active or not self.pool:
            return
        s, e = self.pool.pop()
        ops.profile_arm(kernel, s, e)
        self.used.setdefault(label, []).append((s, e))

    def mean_us(self, kernel):
        ps = self.used.get(kernel, [])
        return 1e3 * sum(s.elapsed_time(e) for s, e in ps) / len(ps) if ps else None

    def reset(self):
        self.used = {}


def install_timer(timer):
    """Wrap the ops the trainer calls so that each call arms the hook for its dominant kernel."""
    from mappo_amd import ops
    for name, kernel, label in (("ppo_loss_fwd_bwd", "ppo_loss", "ppo_loss"), ("mlp_backward", "mlp_bwd", "mlp_backward"),
                                ("actor_update", "mlp_bwd", "actor_update"), ("critic_update", "mlp_bwd", "critic_update"),
                                ("mlp_forward", "mlp_fwd", "mlp_forward"), ("gae_scan", "gae", "gae_scan"),
                                ("actor_critic_update", "mlp_bwd", "actor_critic_update"), ("reduce_clip_adam", "slab_reduce", "slab_reduce"),
                                ("slab_reduce", "slab_reduce", "slab_reduce"), ("actor_act", "act", "actor_act"),
                                ("rollout_step", "act", "rollout_step")):
        orig = getattr(ops, name)

        def wrapped(*a, _orig=orig, _k=kernel, _l=label, **kw):
            timer.arm(ops, _k, _l)
            return _orig(*a, **kw)
        setattr(ops, name, wrapped)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(ns, n_threads):
    """Oracle iterations on the host cores: BASELINE configs[0] shape (N=8, the reference's own CPU case) and
    this run's configs[1] shape (N=1024 global), each with torch.set_num_threads(1) (the reference default, config.py:168-169)
    and with all cores; same shapes / hyper-parameters / synthetic generator as the GPU run; bounded: a few iterations."""
    from oracle import mappo_oracle as O
    T, M, D, A = ns.episode_length, 3, 18, 5

    def one(N, threads, iters):
        torch.set_num_threads(threads)
        oa = O.default_args(episode_length=T, n_rollout_threads=N, ppo_epoch=ns.ppo_epoch, num_mini_batch=ns.num_mini_batch,
                            lr=7e-4, critic_lr=7e-4)
        r = O.RunnerRef(oa, O.SyntheticMPEEnvRef(N, M, D, T, seed=1), M, D, D * M, A, seed=1)
        r.warmup()
        t0 = time.perf_counter()
        for _ in range(iters):
            _, timing = r.run_iteration()
        dt = (time.perf_counter() - t0) / iters
        return dict(agent_steps_per_s=T * N * M / dt, seconds_per_iteration=dt, threads=threads, n_rollout_threads=N,
                    split_s={k: round(float(v), 4) for k, v in timing.items()})

    # warm torch's CPU kernels on a tiny twin first
    one(8, n_threads, 1)
    N2 = ns.n_rollout_threads
    runs = dict(config1_N8_1thread=one(8, 1, 3), config1_N8_allcores=one(8, n_threads, 3),
                config2_1thread=one(N2, 1, 1), config2_allcores=one(N2, n_threads, 2))
    head = runs["config2_allcores"]
    steps = T * N2 * M
    total = sum(r["seconds_per_iteration"] * (3 if "config1" in k else (1 if "1thread" in k else 2)) for k, r in runs.items())
    return dict(value=head["agent_steps_per_s"], unit="agent-steps/s", cores=n_threads, kind="port",
                sample=f"{steps} agent-steps per iteration (T={T} x N={N2} x M={M}, ppo_epoch={ns.ppo_epoch}) of the same workload, "
                       f"torch-CPU/NumPy oracle (oracle/mappo_oracle.py), 2 iterations on {n_threads} threads = `value`; also 1 thread "
                       f"and BASELINE configs[0] (N=8) in `runs`; {total:.1f} s of CPU work in all",
                cpu_model=_cpu_model(), os_cpu_count=os.cpu_count(), runs=runs, seconds=head["seconds_per_iteration"])


def main_c4(ns, world, rank, device, result_out, force_dp):
    """BASELINE configs[3] (SMAC MMM2 shape, recurrent policy): same timing contract, strong or weak scaling over rollout threads."""
    from mappo_amd.envs.synthetic import SyntheticSMACEnv
    from mappo_amd.runner.shared.smac_runner import SMACRunner
    from mappo_amd.distributed import DataParallel
    from mappo_amd.config import get_config
    M, D, S, A = 10, 176, 322, 18
    a = get_config().parse_known_args([])[0]
    a.algorithm_name = "rmappo"
    a.use_recurrent_policy, a.use_naive_recurrent_policy = True, False
    a.env_name = "StarCraft2"
    a.episode_length, a.n_rollout_threads, a.ppo_epoch, a.num_mini_batch = ns.episode_length, ns.local_threads, ns.ppo_epoch, ns.num_mini_batch
    a.lr = a.critic_lr = 5e-4
    a.gain = 1.0                                               # train_smac.sh
    a.seed = 1
    torch.manual_seed(a.seed)
    env = SyntheticSMACEnv(a.n_rollout_threads, M, D, S, A, seed=1 + rank, device=device, pool_steps=a.episode_length)
    dp = DataParallel() if (world > 1 or force_dp) else None
    runner = SMACRunner(dict(all_args=a, envs=env, eval_envs=None, num_agents=M, device=device, run_dir=None, dist_group=dp))
    runner.warmup()
    for i in range(max(ns.warmup, 2)):
        runner.run_episode(i, ns.warmup + ns.steps)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(ns.steps):
        info, _ = runner.run_episode(ns.warmup + i, ns.warmup + ns.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    global_steps = a.episode_length * ns.global_threads * M
    out = dict(metric="agent-steps/sec (collect+GAE+PPO), SMAC MMM2-shaped synthetic env, recurrent policy", value=global_steps * ns.steps / dt,
               unit="agent-steps/s", n_gpus=world, steps=ns.steps, warmup=ns.warmup, ms_per_step=1e3 * dt / ns.steps, higher_is_better=True,
               scaling=ns.scaling, vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload="BASELINE configs[3]: SMAC MMM2 shape, 10 agents, obs 176 / state 322 / Discrete(18), "
                                    f"n_rollout_threads={ns.global_threads} in total ({ns.scaling} scaling: {a.n_rollout_threads} on rank 0 of {world}), "
                                    f"episode_length={a.episode_length}, GRU policy (rmappo, data_chunk_length {a.data_chunk_length}), "
                                    f"ppo_epoch={a.ppo_epoch}, num_mini_batch={a.num_mini_batch}, lr 5e-4",
                           n_rollout_threads_global=ns.global_threads, n_rollout_threads_rank0=a.n_rollout_threads, episode_length=a.episode_length,
                           num_agents=M, ppo_epoch=a.ppo_epoch, num_mini_batch=a.num_mini_batch, agent_steps_per_step=global_steps,
                           parallelism=f"dp{world}", rccl_ranks=ns.rccl_ranks),
               roofline=None, cpu_baseline=None,
               note="secondary configuration (SURVEY 8e strong-scaling shape); the roofline / cpu_baseline objects belong to the default --config c2 line",
               last_train_info={k: float(v) for k, v in info.items()})
    if rank == 0:
        print(json.dumps(out), file=result_out, flush=True)
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(ns):
    """`python bench.py --gpus N` without a launcher around it: this process — which has made NO GPU call (importing torch
    and counting devices does not initialise HIP) — starts `python -m torch.distributed.run` with N ranks of this same script as
    a CHILD process (never an exec: a process that touched the GPU must not be replaced), relays rank 0's single JSON line
    and exits with the child's code."""
    import socket
    import subprocess
    if not ns.dry_launch:
        n_dev = torch.cuda.device_count()
        if n_dev < ns.gpus:
            print(f"bench.py: --gpus {ns.gpus} but only {n_dev} GPU(s) visible on this node", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ns.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    if ns.dry_launch:
        for ln in sorted(lines):
            print(ln, flush=True)
        return proc.returncode
    result = [ln for ln in lines if ln.lstrip().startswith("{") and '"metric"' in ln]
    for ln in lines:
        if ln not in result:
            print(ln, file=sys.stderr)                          # anything else a rank wrote to stdout is not the result line
    if proc.returncode != 0 or len(result) != 1:
        print(f"bench.py: {ns.gpus}-rank run failed (exit code {proc.returncode}, {len(result)} result lines)", file=sys.stderr)
        return proc.returncode or 1
    print(result[0], flush=True)
    return 0


def main():
    ns = parse()
    if ns.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(ns))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if ns.gpus > 1 and world != ns.gpus:
        sys.exit(f"bench.py: --gpus {ns.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if ns.dry_launch:
        print(json.dumps(dict(rank=rank, world=world, local_rank=int(os.environ.get("LOCAL_RANK", "0")), gpus=ns.gpus)), flush=True)
        return
    from mappo_amd.distributed import shard_threads
    if ns.scaling == "strong":                              # global thread count fixed: this rank's contiguous share
        lo, hi = shard_threads(ns.n_rollout_threads, rank, world)
        ns.local_threads, ns.global_threads = hi - lo, ns.n_rollout_threads
    else:
        ns.local_threads, ns.global_threads = ns.n_rollout_threads, ns.n_rollout_threads * world
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dp = os.environ.get("MAPPO_BENCH_FORCE_DP") == "1"     # rehearsal of the data-parallel code path on ONE rank
    result_out = sys.stdout
    if world > 1 or force_dp:
        # RCCL prints its version banner on stdout when a communicator is created; the contract is ONE JSON line on stdout, so
        # the process's fd 1 is pointed at stderr and the JSON line goes to a duplicate of the original stdout
        sys.stdout.flush()
        result_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world and (ns.gpus <= 1 or dist.get_world_size() == ns.gpus), \
            f"process group has {dist.get_world_size()} ranks, expected {max(ns.gpus, world)}"
        ns.rccl_ranks = dist.get_world_size()
    else:
        torch.cuda.set_device(0)
        ns.rccl_ranks = 0
    device = torch.device("cuda", torch.cuda.current_device())

    from mappo_amd.envs.synthetic import SyntheticMPEEnv
    from mappo_amd.runner.shared.mpe_runner import MPERunner
    from mappo_amd.distributed import DataParallel

    if ns.config == "c4":
        return main_c4(ns, world, rank, device, result_out, force_dp)
    args = make_args(ns)
    M, D, A = 3, 18, 5
    torch.manual_seed(args.seed)                           # identical initial replicas on every rank
    if ns.env == "mpe":
        from mappo_amd.envs.mpe_spread import SimpleSpreadVecEnv
        env = SimpleSpreadVecEnv(args.n_rollout_threads, M, 3, args.episode_length, seed=1 + rank, device=device)
    else:
        env = SyntheticMPEEnv(args.n_rollout_threads, M, D, A, args.episode_length, seed=1 + rank, device=device)
    dp = DataParallel() if (world > 1 or force_dp) else None
    runner = MPERunner(dict(all_args=args, envs=env, eval_envs=None, num_agents=M, device=device, run_dir=None, dist_group=dp))
    timer = KernelTimer()
    install_timer(timer)
    timer.reserve(4 * (args.episode_length + 4 * args.ppo_epoch * args.num_mini_batch + 8) + 64)

    runner.warmup()
    for i in range(max(ns.warmup, 2)):                     # >= 2: eager pass, then hipGraph capture + first replay
        runner.run_episode(i, ns.warmup + ns.steps)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(ns.steps):
        info, _ = runner.run_episode(ns.warmup + i, ns.warmup + ns.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- N > 1: the same K steps with the per-GPU work FIXED (weak scaling: configs[1]'s 1024 threads on EVERY rank) as a
    # companion figure — `value` above stays the strong-scaling number north_star asks for; a 1.3 ms iteration of ~50 dependent
    # launches cannot strong-scale, and the pair shows how much of the gap is launch latency and how much is the all-reduce ----
    companion = None
    if (world > 1 or force_dp) and ns.scaling == "strong" and not ns.no_companion:
        import copy
        args_w = copy.copy(args)
        args_w.n_rollout_threads = ns.n_rollout_threads
        if ns.env == "mpe":
            env_w = SimpleSpreadVecEnv(args_w.n_rollout_threads, M, 3, args_w.episode_length, seed=101 + rank, device=device)
        else:
            env_w = SyntheticMPEEnv(args_w.n_rollout_threads, M, D, A, args_w.episode_length, seed=101 + rank, device=device)
        torch.manual_seed(args.seed)
        runner_w = MPERunner(dict(all_args=args_w, envs=env_w, eval_envs=None, num_agents=M, device=device, run_dir=None, dist_group=dp))
        runner_w.warmup()
        for i in range(max(ns.warmup, 2)):
            runner_w.run_episode(i, ns.warmup + ns.steps)
        barrier()
        t0 = time.perf_counter()
        for i in range(ns.steps):
            runner_w.run_episode(ns.warmup + i, ns.warmup + ns.steps)
        barrier()
        dt_w = time.perf_counter() - t0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt_w], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_w = float(t.item())
        steps_w = args_w.episode_length * args_w.n_rollout_threads * world * M
        companion = dict(scaling="weak", value=steps_w * ns.steps / dt_w, unit="agent-steps/s", ms_per_step=1e3 * dt_w / ns.steps,
                         n_rollout_threads_global=args_w.n_rollout_threads * world, n_rollout_threads_per_rank=args_w.n_rollout_threads,
                         steps=ns.steps, note="same run, same ranks, same timing contract; per-GPU work fixed at configs[1]'s size")
        del runner_w, env_w

    # ---- device time of the two phases of a step (same hipGraph path, three more iterations, HIP events) ----
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ph = [0.0, 0.0]
    for i in range(3):
        ev[0].record(); runner.rollout(); ev[1].record(); runner.train(); ev[2].record()
        torch.cuda.synchronize()
        ph[0] += ev[0].elapsed_time(ev[1]) / 3; ph[1] += ev[1].elapsed_time(ev[2]) / 3
    phase_ms = dict(rollout_T_steps_plus_gae=ph[0], train_ppo_epochs=ph[1])

    # ---- per-kernel device time: two more iterations of the SAME step, launched eagerly so that every dispatch of
    # the hot kernels carries HIP events (hipGraph replays cannot be instrumented from the host) ----
    graph_flags = (runner._use_graph, runner.trainer._use_graph)
    runner._use_graph = runner.trainer._use_graph = False
    labels = ["actor_critic_update", "actor_update", "critic_update", "rollout_step", "mlp_forward", "gae_scan", "slab_reduce", "actor_act", "ppo_loss",
              "mlp_backward"]
    timer.active = set(labels)
    for i in range(2):
        runner.run_episode(0, 1)
    torch.cuda.synchronize()
    kern = {k: timer.mean_us(k) for k in labels if timer.mean_us(k) is not None}
    timer.active = set()
    runner._use_graph, runner.trainer._use_graph = graph_flags

    per_gpu_steps = args.episode_length * args.n_rollout_threads * M      # this rank's share
    global_steps = args.episode_length * ns.global_threads * M
    value = global_steps * ns.steps / dt
    S = per_gpu_steps // args.num_mini_batch                 # samples per update-kernel launch
    H = 64
    macs_critic = D * M * H + H * H + H                     # forward MACs per sample (share_obs 54 -> 64 -> 64 -> 1)
    macs_actor = D * H + H * H + H * A                      # obs 18 -> 64 -> 64 -> 5
    roofline = None
    note = ("algorithmic flops = 6 x forward MACs per sample (forward, dW, dX; SURVEY.md 8d); the kernel issues fewer: the input "
            "layer needs no dX (feature-norm gradients come from the dW products)")
    if kern.get("actor_critic_update"):
        us = kern["actor_critic_update"]
        flops = S * 6 * (macs_critic + macs_actor)
        achieved = flops / (us * 1e-6) / 1e12
        traffic, tnote = measured_traffic("mlp_update16_dual_kernel", S)
        roofline = dict(bound="mfma", kernel="mlp_update16_dual_kernel<relu, layer_N=1> (actor + critic update in one launch, one wave per 16-sample tile)",
                        achieved=achieved, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / MFMA_F32_PEAK_TFLOPS,
                        traffic=traffic, traffic_note=tnote, algorithmic_bytes=S * 4 * ((D * M + D) + (3 + 4 + A)) + 4 * 22678,
                        flops_per_launch=flops, launch_us=us, note=note)
    elif kern.get("critic_update"):
        us = kern["critic_update"]
        flops = S * 6 * macs_critic                          # forward + dW + dX, 2 flop per MAC
        achieved = flops / (us * 1e-6) / 1e12
        roofline = dict(bound="mfma", kernel="mlp_update2_kernel<relu, layer_N=1, HEAD=critic, wide>", achieved=achieved,
                        peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / MFMA_F32_PEAK_TFLOPS,
                        traffic=None, flops_per_launch=flops, launch_us=us, note=note)
    loss_roof = ppo_loss_roofline(runner, timer, A)
    gae_roof = gae_roofline(runner, timer) if rank == 0 else None
    out = dict(metric="agent-steps/sec (collect+GAE+PPO), MPE simple_spread 3-agent", value=value, unit="agent-steps/s",
               n_gpus=world, steps=ns.steps, warmup=ns.warmup, ms_per_step=1e3 * dt / ns.steps, higher_is_better=True,
               scaling=ns.scaling, vs_baseline=None, dtype="f32",
               data="synthetic" if ns.env == "synthetic" else "MPE simple_spread dynamics on the GPU (mappo_amd/envs/mpe_spread.py), random-init weights",
               config=dict(workload="BASELINE configs[1]: MPE simple_spread shape, 3 agents, obs 18 / share_obs 54 / Discrete(5), "
                                    f"n_rollout_threads={ns.global_threads} in total ({ns.scaling} scaling: {args.n_rollout_threads} on rank 0 of "
                                    f"{world}), episode_length={args.episode_length}, "
                                    f"MLP policy (mappo), ppo_epoch={args.ppo_epoch}, num_mini_batch={args.num_mini_batch}, lr 7e-4, env={ns.env}",
                           n_rollout_threads_global=ns.global_threads, n_rollout_threads_rank0=args.n_rollout_threads,
                           episode_length=args.episode_length,
                           num_agents=M, ppo_epoch=args.ppo_epoch, num_mini_batch=args.num_mini_batch,
                           agent_steps_per_step=global_steps, parallelism=f"dp{world}", rccl_ranks=ns.rccl_ranks,
                           exact_minibatch_order=bool(args.exact_minibatch_order), hip_graph=bool(graph_flags[1])),
               roofline=roofline, ppo_loss_roofline=loss_roof, gae_roofline=gae_roof, kernels_us=kern, phase_ms=phase_ms,
               weak_scaling_companion=companion,
               last_train_info={k: float(v) for k, v in info.items()})
    if rank == 0:
        if world == 1 and not ns.no_cpu_baseline:
            n_threads = ns.cpu_threads or min(os.cpu_count() or 1, 16)
            try:
                out["cpu_baseline"] = cpu_baseline(ns, n_threads)
            except Exception as e:                             # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = dict(value=None, unit="agent-steps/s", cores=n_threads, kind="port", sample=f"failed: {e}")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=result_out, flush=True)
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
