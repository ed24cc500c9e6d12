"""Utterance sharding over the GPUs of one node: one process per GPU (torchrun), utterances are
independent units (every op on the path is per sample), so ranks never exchange data inside the
sampler — torch.distributed (RCCL on ROCm, gloo in CPU tests) is used for the start/stop barriers
and the optional gather of results only."""
import os

import torch


def init(backend=None, single_rank_group=False):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world, local_rank).  single_rank_group: form the
    group even when the launcher started ONE rank (the N-rank code path - RCCL init, barrier - exercised on a one-GPU box)."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = dict(device_id=torch.device("cuda", local)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def parse_cpulist(text):
    """'0-3,8,10-11' (the format of /sys/devices/system/node/node*/cpulist) -> [0, 1, 2, 3, 8, 10, 11]"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(local_rank, sysfs="/sys"):
    """Pin this rank's launch thread(s) to the CPUs of its GPU's NUMA node: on an 8-GPU node the sampler's host loop (one C-ABI
    call per score evaluation + the fused SDE updates) otherwise migrates across sockets and its dispatch latency with it.
    Best effort - returns the CPU list it pinned to, or None when the topology is not exposed (containers, single-node boxes with
    numa_node = -1) or the platform has no sched_setaffinity."""
    try:
        prop = torch.cuda.get_device_properties(local_rank)
        bdf = f"{getattr(prop, 'pci_domain_id', 0):04x}:{prop.pci_bus_id:02x}:{prop.pci_device_id:02x}.0"
        with open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")) as f:
            cpus = parse_cpulist(f.read())
        allowed = os.sched_getaffinity(0)
        cpus = [c for c in cpus if c in allowed]
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, AttributeError, ValueError, RuntimeError, AssertionError):
        return None


def self_launch(n_procs, script, argv):
    """Re-execute `script` as `n_procs` ranks of ONE node under torch.distributed.run (one process per GPU; rendezvous on
    127.0.0.1 with a free port).  Used when a tool is started as plain `python tool.py --gpus N` without a launcher:
    replaces the current process, never returns."""
    import socket
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)
    os.execv(sys.executable, cmd)


def timed_steps(step, steps, warmup, sync=None, group_ready=True, single_rank_group=False):
    """The bench contract's timed region: `warmup` untimed steps, then EXACTLY `steps` steps bracketed by
    (device sync, barrier, device sync) on both sides; returns (seconds = MAX over ranks, last step's result).
    `sync` = torch.cuda.synchronize on a GPU rank, None on CPU.  single_rank_group: run the barrier / gather lines through an
    initialised one-rank group too (bench.py --dist-world1: the N-rank code path exercised on one GPU)."""
    import time
    import torch.distributed as dist
    multi = group_ready and dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_group)

    def fence():
        if sync is not None:
            sync()
        if multi:
            dist.barrier()
        if sync is not None:
            sync()

    out = None
    for i in range(warmup):
        out = step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if multi:
        per_rank = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank, elapsed)
        elapsed = max(per_rank)
    return elapsed, per_rank, out


def shard_indices(n_items, rank, world, lengths=None):
    """Indices of the utterances rank `rank` processes.  With `lengths`, items are dealt longest-first in
    a serpentine order so every rank gets a similar amount of audio (tail effect of ragged batches)."""
    idx = list(range(n_items))
    if lengths is not None:
        order = sorted(idx, key=lambda i: -lengths[i])
        mine = []
        for k, i in enumerate(order):
            rnd, pos = divmod(k, world)
            owner = pos if rnd % 2 == 0 else world - 1 - pos
            if owner == rank:
                mine.append(i)
        return sorted(mine)
    return idx[rank::world]


def group_by_length(lengths, max_batch):
    """Batches of equal-length utterances (each utterance keeps its own padded frame count, so a batched
    call equals per-utterance calls): list of index lists, at most max_batch long."""
    by_len = {}
    for i, n in enumerate(lengths):
        by_len.setdefault(int(n), []).append(i)
    batches = []
    for n in sorted(by_len):
        ids = by_len[n]
        for k in range(0, len(ids), max_batch):
            batches.append(ids[k:k + max_batch])
    return batches


def bucket_by_frames(lengths, max_batch, hop=128, multiple=64):
    """Micro-batches for utterances of DIFFERENT lengths: the reference pads every utterance's spectrogram to its own
    multiple of 64 frames (util/other.py:102-109), and every op of the path is per utterance, so utterances whose padded
    frame count roundup(1 + L // hop, multiple) is equal can share a batch and still equal their single-utterance runs
    (2-10 s at 16 kHz: 17 buckets).  Returns index lists (at most max_batch long), longest bucket first."""
    by_t = {}
    for i, n in enumerate(lengths):
        by_t.setdefault(-(-(1 + int(n) // hop) // multiple) * multiple, []).append(i)
    batches = []
    for t in sorted(by_t, reverse=True):
        ids = sorted(by_t[t], key=lambda i: -int(lengths[i]))
        for k in range(0, len(ids), max_batch):
            batches.append(ids[k:k + max_batch])
    return batches


def gather_objects(obj, rank, world):
    """All ranks' python objects on rank 0 (None elsewhere)."""
    if world == 1:
        return [obj]
    import torch.distributed as dist
    out = [None] * world if rank == 0 else None
    dist.gather_object(obj, out, dst=0)
    return out


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def finish():
    """Leave the process group together (last line of a sharded tool)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
