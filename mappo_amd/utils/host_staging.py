"""Pinned, double-buffered host <-> device staging for CPU vec-envs (the reference's Dummy/SubprocVecEnv
and ShareSubprocVecEnv hand NumPy arrays across the process boundary, env_wrappers.py:257-272,363-379).

Per rollout step a host env produces a handful of small `[N, M, *]` arrays (obs, rewards, dones, SMAC: share_obs /
available_actions / bad_transition) and consumes the actions.  Going through `torch.as_tensor(x).to(device)` costs, per array,
a pageable-memory staging copy inside the driver and a blocking transfer; downloading the actions with `.cpu()` synchronises
the whole device.  Here:

  * every array has TWO pinned host blocks and TWO device twins; `upload()` memcpys the env's arrays into the free pinned
    block and issues the H2D copies on a dedicated copy stream — the compute stream waits on an event, the host never does;
    the block of step t is reused at step t + 2, by which time its copy has long finished (the event is checked anyway);
  * the device twins have exactly the layout the fused rollout kernels read (`mappo_rollout_step`, `mappo_insert_smac`),
    so a host env takes the same one-launch step as a device env;
  * `download()` copies the actions into pinned memory on the copy stream and waits for THAT event only.

What cannot overlap is the data dependency of RL itself: the env needs the actions of step t before it can produce the
observations of step t + 1."""
import numpy as np
import torch


class HostStaging:
    SLOTS = 2

    def __init__(self, device):
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._up = {}          # name -> dict(host=[pinned tensors], np=[numpy views], dev=[device twins])
        self._down = {}
        self._k = 0            # upload slot of the current step
        self._up_done = [torch.cuda.Event() for _ in range(self.SLOTS)]
        self._read_done = [None] * self.SLOTS      # compute-stream event: kernels that read slot k have been enqueued before it
        self._pending_reads = []                   # per-upload events consumed() has not recorded yet
        self._dk = 0
        self._down_done = [torch.cuda.Event() for _ in range(self.SLOTS)]

    @staticmethod
    def _dtype_of(arr):
        a = np.asarray(arr) if not torch.is_tensor(arr) else arr
        return torch.bool if str(a.dtype) in ("bool", "torch.bool") else torch.float32

    def _blocks(self, table, name, shape, dtype):
        e = table.get(name)
        if e is None or tuple(e["host"][0].shape) != tuple(shape) or e["host"][0].dtype != dtype:
            host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(self.SLOTS)]
            e = dict(host=host, np=[h.numpy() for h in host], dev=[torch.empty(shape, dtype=dtype, device=self.device) for _ in range(self.SLOTS)])
            table[name] = e
        return e

    def _packed(self, specs):
        """One pinned byte block + one device byte block per slot holding all arrays of an upload back to back (16-byte
        aligned): ONE H2D copy per step instead of one per array.  specs = ((name, shape, torch dtype), ...)."""
        key = tuple((n, tuple(sh), dt) for n, sh, dt in specs)
        e = self._up.get("__packed__")
        if e is not None and e["key"] == key:
            return e
        offs, total = [], 0
        for n, sh, dt in specs:
            nbytes = int(np.prod(sh)) * (1 if dt == torch.bool else 4)
            offs.append((total, nbytes))
            total += (nbytes + 15) & ~15
        host = [torch.empty(total, dtype=torch.uint8).pin_memory() for _ in range(self.SLOTS)]
        dev = [torch.empty(total, dtype=torch.uint8, device=self.device) for _ in range(self.SLOTS)]
        views_np, views_dev = [], []
        for k in range(self.SLOTS):
            vn, vd = {}, {}
            for (n, sh, dt), (o, nb) in zip(specs, offs):
                vn[n] = host[k][o:o + nb].view(dt).view(sh).numpy()
                vd[n] = dev[k][o:o + nb].view(dt).view(sh)
            views_np.append(vn); views_dev.append(vd)
        e = dict(key=key, host=host, dev=dev, np=views_np, views=views_dev)
        self._up["__packed__"] = e
        return e

    def upload(self, **arrays):
        """NumPy (or host torch) arrays -> device tensors of the same shape (bool stays bool, everything else float32), valid
        on the current stream.  None values and device tensors are passed through untouched."""
        k = self._k
        self._k = (k + 1) % self.SLOTS
        self._up_done[k].synchronize()                       # the H2D copy that last read this pinned block is done
        out, host_arrays = {}, []
        for name, arr in arrays.items():
            if arr is None or (torch.is_tensor(arr) and arr.device == self.device):
                out[name] = arr
                continue
            a = arr.numpy() if torch.is_tensor(arr) else np.asarray(arr)
            host_arrays.append((name, a, torch.bool if a.dtype == np.bool_ else torch.float32))
        if host_arrays:
            e = self._packed(tuple((n, a.shape, dt) for n, a, dt in host_arrays))
            for n, a, _ in host_arrays:
                np.copyto(e["np"][k][n], a, casting="unsafe")    # one memcpy (+ dtype conversion) into pinned memory
                out[n] = e["views"][k][n]
            cur = torch.cuda.current_stream(self.device)
            if self._read_done[k] is not None:
                if self._read_done[k] in self._pending_reads:      # never consumed(): everything enqueued so far may read it
                    self._read_done[k].record(cur)
                    self._pending_reads.remove(self._read_done[k])
                self.copy_stream.wait_event(self._read_done[k])    # kernels that still read the device twin of this slot
            with torch.cuda.stream(self.copy_stream):
                e["dev"][k].copy_(e["host"][k], non_blocking=True)
                self._up_done[k].record(self.copy_stream)
            cur.wait_event(self._up_done[k])
            ev = torch.cuda.Event()
            self._read_done[k] = ev                          # recorded by consumed()
            self._pending_reads.append(ev)                   # (two uploads without a consumed() in between: both stay protected)
        return out

    def consumed(self):
        """Call after enqueueing the kernels that read the tensors of the last upload(): their slot may be refilled once
        everything enqueued so far has run."""
        cur = torch.cuda.current_stream(self.device)
        for ev in self._pending_reads:
            ev.record(cur)
        self._pending_reads = []

    def download(self, name, tensor):
        """Device tensor -> NumPy array in pinned memory (valid until the call after next), waiting for this copy only."""
        k = self._dk
        self._dk = (k + 1) % self.SLOTS
        e = self._blocks(self._down, name, tensor.shape, tensor.dtype)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.copy_stream.wait_event(ev)
        with torch.cuda.stream(self.copy_stream):
            e["host"][k].copy_(tensor, non_blocking=True)
            self._down_done[k].record(self.copy_stream)
        self._down_done[k].synchronize()
        return e["np"][k]
