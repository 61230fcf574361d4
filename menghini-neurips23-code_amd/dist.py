"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm)
on the xGMI mesh; "gloo" on CPU for the world_size-2 tests.

The pseudolabel pool shards contiguously over ranks (rank r owns images [r*ceil(N/g), (r+1)*ceil(N/g)))
so that the gathered embeddings are in dataset order; the only data-path collective is one
all-gather of the [N/g, E] f32 embeddings.  Prompt gradients (<= 2.1 MB) are
all-reduced once per step.
"""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world():
    return (dist.get_rank(), dist.get_world_size()) if is_dist() else (0, 1)


def init_from_env(backend=None):
    """Initialise from torchrun's RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* when present."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    # a 1-rank group is only formed when the C ABI's own communicator is asked for (GRIP_NATIVE_COMM=1 under a launcher): that is
    # how its RCCL calls run inside the real data path on a one-GPU box
    solo_native = ws == 1 and os.environ.get("GRIP_NATIVE_COMM") == "1" and "RANK" in os.environ
    if (ws <= 1 and not solo_native) or is_dist():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("GRIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device_index())
    dist.init_process_group(backend=backend)
    return world()


def local_device_index():
    """GPU of this rank: LOCAL_RANK, or 0 for every rank when GRIP_SINGLE_DEVICE=1 (several ranks sharing one GPU
    over gloo: how the N > 1 path is exercised on a one-GPU box)."""
    return 0 if os.environ.get("GRIP_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))


def pin_to_gpu_numa_node(device_index):
    """Restrict this process to the CPUs of the NUMA node its GPU hangs off (sysfs: the PCI device's numa_node -> the node's
    cpulist), so the rank's host side -- the sequential leaderboard scan, kernel launches, the staging copies of the input
    pipeline -- runs next to its GPU instead of across the socket interconnect.  Returns the node number, or None when the
    topology cannot be read (containers often hide it) or GRIP_NUMA_PIN=0; never raises."""
    if os.environ.get("GRIP_NUMA_PIN", "1") == "0":
        return None
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def _via_host():
    return dist.get_backend() == "gloo"


def trace(line):
    """$GRIP_COMM_TRACE=<file>: one line per collective this process enters ("rank<r> <op> tag=<what> ..."), whatever the transport (torch's RCCL
    group, gloo, the C ABI's communicator).  Developer / test switch: tests/test_gpu_dist.py counts the exchange steps of a pass with it."""
    path = os.environ.get("GRIP_COMM_TRACE")
    if path:
        with open(path, "a") as f:
            f.write(f"rank{world()[0]} {line}\n")


class NativeComm:
    """The C ABI's own RCCL communicator (include/grip_amd.h: grip_comm_*, grip_allgather_embeddings, grip_allreduce_mean):
    what a host that is not PyTorch binds for the two exchange steps.  The ncclUniqueId travels through torch.distributed's
    object broadcast here (any side channel works).  Opt-in for the Python host (GRIP_NATIVE_COMM=1); the default path
    below uses torch.distributed's RCCL process group, which is the same library."""

    def __init__(self):
        import ctypes
        from . import native
        self.lib = native.lib()
        self.rank, self.ws = world()
        os.environ.setdefault("GRIP_RCCL_LIBRARY", os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        ident = (ctypes.c_uint8 * 128)()
        if self.rank == 0:
            native.check(self.lib.grip_comm_unique_id(ident))
        box = [bytes(ident)]
        if self.ws > 1:
            dist.broadcast_object_list(box, src=0)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(box[0])
        h = ctypes.c_void_p()
        native.check(self.lib.grip_comm_init_rank(buf, self.ws, self.rank, ctypes.byref(h)))
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.grip_comm_destroy(self.handle)
            self.handle = None

    def _trace(self, what):
        trace("native " + what)

    def allgather(self, local, per):
        from . import native
        import ctypes
        self._trace(f"allgather rows={per} ranks={self.ws}")
        local = local.contiguous()
        out = torch.empty(self.ws * per, local.shape[1], dtype=torch.float32, device=local.device)
        native.check(self.lib.grip_allgather_embeddings(self.handle, ctypes.c_void_p(local.data_ptr()), ctypes.c_void_p(out.data_ptr()), per,
                                                        local.shape[1], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return out

    def allreduce_mean_(self, flat):
        from . import native
        import ctypes
        self._trace(f"allreduce n={flat.numel()} ranks={self.ws}")
        native.check(self.lib.grip_allreduce_mean(self.handle, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return flat


_NATIVE = None


def native_comm():
    """The process-wide NativeComm when GRIP_NATIVE_COMM=1 and the ranks talk RCCL, else None."""
    global _NATIVE
    if os.environ.get("GRIP_NATIVE_COMM") != "1" or not is_dist() or _via_host():
        return None
    if _NATIVE is None:
        import atexit
        _NATIVE = NativeComm()
        atexit.register(_NATIVE.close)      # ncclCommDestroy before the process group / HIP runtime go away
    return _NATIVE


def shard_range(n, rank=None, world_size=None):
    """Contiguous shard [lo, hi) of an ordered pool of n units, equal padded length `per`."""
    if rank is None:
        rank, world_size = world()
    per = (n + world_size - 1) // world_size
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi, per


def allgather_rows(local, n_total, per, tag="rows"):
    """local [<= per, E] rows of this rank's shard -> [n_total, E] in global order on every rank.
    Pads the last shard to `per` rows and drops the padding after the gather.  `tag` names the exchange step in $GRIP_COMM_TRACE."""
    rank, ws = world()
    if ws == 1 and native_comm() is None:
        return local[:n_total]
    e = local.shape[1]
    trace(f"allgather tag={tag} rows_per_rank={per} width={e} ranks={ws}")
    if local.shape[0] != per:
        pad = torch.zeros(per, e, dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
        local = pad
    if _via_host() and local.is_cuda:          # gloo moves host memory only
        host = torch.empty(ws * per, e, dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous())
        return host[:n_total].to(local.device)
    nc = native_comm()
    if nc is not None and local.is_cuda and local.dtype == torch.float32:
        return nc.allgather(local, per)[:n_total]
    out = torch.empty(ws * per, e, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out[:n_total]


def allgather_selected(local_rows, idx, n_total, tag="selected_rows"):
    """Rows of a scattered selection of the pool on every rank.  `idx`: ascending global row numbers, the same array on every
    rank (the rows a screen-and-refine scan asked for); `local_rows` [m_r, E]: this rank's rows of `idx` that fall into its
    contiguous shard (shard_range), in order.  Returns [len(idx), E] in the order of `idx`.  One all-gather, padded to the
    largest per-rank count."""
    import numpy as np
    rank, ws = world()
    if ws == 1:
        return local_rows
    per = (n_total + ws - 1) // ws
    idx = np.asarray(idx, dtype=np.int64)
    bounds = np.searchsorted(idx, np.arange(ws + 1, dtype=np.int64) * per)
    counts = np.diff(bounds)
    assert local_rows.shape[0] == counts[rank], (local_rows.shape, counts, rank)
    m = int(max(counts.max(), 1))
    out = allgather_rows(local_rows, ws * m, m, tag=tag)
    pick = np.concatenate([r * m + np.arange(counts[r], dtype=np.int64) for r in range(ws)])
    return out[torch.as_tensor(pick, device=out.device)]


# ------------------------------------------------------------------------------------------ data-parallel batches (the trainer's loops)
def rank_batches(order, batch_size, rank=None, world_size=None):
    """The index batches rank `rank` of `world_size` processes walks in one epoch over the samples `order` -- what HF accelerate's prepared
    DataLoader gives the reference's loops (`accelerator.prepare(train_loader)`, e.g. textual_prompt.py:239; accelerate's BatchSamplerShard with
    split_batches=False, even_batches=True): EVERY rank runs full batches of `batch_size` (so the global batch is world_size x batch_size),
    batch j of the epoch goes to rank j % world_size, a ragged last batch is completed with samples from the START of the epoch's order and
    further such batches are appended until the count divides by world_size (the duplicates are what test_predictions drops again,
    textual_prompt.py:291-294).  One process: the plain batches, ragged tail kept.  Held equal to accelerate's sampler in tests/test_dist_gloo.py."""
    if rank is None:
        rank, world_size = world()
    order = [int(i) for i in order]
    batches = [order[i: i + batch_size] for i in range(0, len(order), batch_size)]
    if world_size == 1 or not batches:
        return batches
    fill = [i for b in batches[:world_size] for i in b]       # the padding cycles through the epoch's first world_size batches
    while len(fill) < world_size * batch_size:
        fill = fill + fill
    at = batch_size - len(batches[-1])
    batches[-1] = batches[-1] + fill[:at]
    while len(batches) % world_size:
        batches.append(fill[at: at + batch_size])
        at += batch_size
    return batches[rank::world_size]


class RankBatchSampler:
    """`batch_sampler` of the trainer's DataLoaders: a fresh permutation per epoch from a generator that is seeded alike on every rank
    (shuffle) or the dataset order, dealt to the ranks by rank_batches."""

    def __init__(self, n, batch_size, shuffle, seed=0):
        self.n, self.batch_size = int(n), int(batch_size)
        self.generator = torch.Generator().manual_seed(seed) if shuffle else None

    def __iter__(self):
        order = torch.randperm(self.n, generator=self.generator).tolist() if self.generator is not None else range(self.n)
        return iter(rank_batches(order, self.batch_size))

    def __len__(self):
        return len(rank_batches(range(self.n), self.batch_size))


def gather_in_dataset_order(local_rows, n, batch_size):
    """Per-sample rows every rank computed for ITS batches of the un-shuffled dataset (rank_batches(range(n), batch_size), concatenated in
    order) -> [n, ...] rows in dataset order on every rank: `accelerator.gather` + the de-duplication of the padded tail
    (textual_prompt.py:285-294; a duplicated sample's rows are identical, the first occurrence is kept)."""
    rank, ws = world()
    if ws == 1 or n == 0:
        return local_rows[:n]
    per_rank = [[i for b in rank_batches(range(n), batch_size, r, ws) for i in b] for r in range(ws)]
    m = len(per_rank[0])
    assert all(len(p) == m for p in per_rank) and local_rows.shape[0] == m, (local_rows.shape, [len(p) for p in per_rank])
    flat = allgather_rows(local_rows.reshape(m, -1), ws * m, m, tag="per_sample_rows")
    import numpy as np
    uniq, first = np.unique(np.array([i for p in per_rank for i in p], dtype=np.int64), return_index=True)      # position of each sample's FIRST occurrence
    assert len(uniq) == n and uniq[0] == 0 and uniq[-1] == n - 1
    return flat[torch.from_numpy(first).to(flat.device)].reshape((n,) + tuple(local_rows.shape[1:]))


def broadcast_array_(a, src=0):
    """A host numpy array (same shape / dtype on every rank) overwritten with rank `src`'s content."""
    rank, ws = world()
    if ws == 1:
        return a
    trace(f"broadcast bytes={a.nbytes} src={src}")
    t = torch.from_numpy(a)
    if _via_host():
        dist.broadcast(t, src=src)
    else:                                   # RCCL moves device memory
        d = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(d, src=src)
        t.copy_(d.cpu())
    return a


def allreduce_sum_(t):
    """In-place sum of a small tensor over the ranks (epoch statistics)."""
    if not is_dist() or world()[1] == 1:
        return t
    if _via_host() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
    else:
        dist.all_reduce(t)
    return t


def allreduce_mean_(tensors):
    """In-place mean all-reduce of the (tiny) prompt gradients, flattened into one message."""
    rank, ws = world()
    if (ws == 1 and native_comm() is None) or not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    trace(f"allreduce_mean n={flat.numel()} ranks={ws}")
    if _via_host() and flat.is_cuda:
        h = flat.cpu()
        dist.all_reduce(h)
        flat = h.to(flat.device)
        flat /= ws
    elif native_comm() is not None and flat.is_cuda and flat.dtype == torch.float32:
        native_comm().allreduce_mean_(flat)          # the C ABI returns the mean
    else:
        dist.all_reduce(flat)
        flat /= ws
    off = 0
    for t in tensors:
        t.copy_(flat[off: off + t.numel()].view_as(t))
        off += t.numel()


def barrier():
    if is_dist():
        dist.barrier()
