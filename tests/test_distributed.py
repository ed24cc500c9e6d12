"""Utterance sharding (world_size 2, gloo, CPU): every utterance is processed exactly once, sharded ==
unsharded, and the host-side launch/barrier/gather plumbing works.  The per-utterance work is a
deterministic stand-in here (the kernels are covered by the -m gpu suite): what is tested is the
partitioning contract the multi-GPU bench relies on."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, {root!r})
from storm_amd import distributed as D
rank, world, local = D.init(backend="gloo")
lengths = [32000, 64000, 48000, 64000, 32000, 80000, 64000, 16000, 48000]
mine = D.shard_indices(len(lengths), rank, world, lengths)
batches = D.group_by_length([lengths[i] for i in mine], max_batch=2)
done = []
for b in batches:
    ids = [mine[k] for k in b]
    assert len(set(lengths[i] for i in ids)) == 1          # equal-length batches only
    for i in ids:
        g = torch.Generator().manual_seed(i)               # per-utterance work: seeded by the utterance id
        done.append((i, float(torch.randn(4, generator=g).sum())))
D.barrier()
res = D.gather_objects(done, rank, world)
if rank == 0:
    print("RESULT " + json.dumps(res))
D.barrier()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env={**os.environ, "OMP_NUM_THREADS": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    per_rank = json.loads(line[len("RESULT "):])
    assert len(per_rank) == 2
    allv = sorted(tuple(v) for r in per_rank for v in r)
    assert [i for i, _ in allv] == list(range(9))                          # each utterance exactly once
    for i, v in allv:                                                      # sharded == unsharded value
        g = torch.Generator().manual_seed(i)
        assert abs(v - float(torch.randn(4, generator=g).sum())) < 1e-6
    # balanced by audio length (serpentine deal)
    lengths = [32000, 64000, 48000, 64000, 32000, 80000, 64000, 16000, 48000]
    loads = [sum(lengths[i] for i, _ in r) for r in per_rank]
    assert abs(loads[0] - loads[1]) <= 32000


def test_shard_helpers():
    from storm_amd import distributed as D
    assert D.shard_indices(10, 1, 4) == [1, 5, 9]
    parts = [D.shard_indices(7, r, 3, [5, 9, 1, 7, 3, 8, 2]) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(7))
    assert D.group_by_length([4, 8, 4, 4, 8], 2) == [[0, 2], [3], [1, 4]]


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` with no launcher must become two ranks itself (the driver starts it exactly like that),
    shard the batch, time the SLOWEST rank between barriers and print one JSON line with n_gpus == 2.  Run here on CPU
    ranks (gloo) with a stand-in step: the launch / rendezvous / timing skeleton is bench.py's own code."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--selftest-cpu"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1                                   # rank 0 only
    r = json.loads(line[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["utterances_per_rank"] == 16
    assert len(r["per_rank_s"]) == 2
    # rank 1's stand-in step is twice as slow: the reported time is at least its 3 x 40 ms
    assert r["ms_per_step"] >= 39.0
    assert abs(r["value"] - 32 * 3 / (r["ms_per_step"] * 3e-3)) < 1e-6 * r["value"]


def test_bench_gpus_flag_must_match_the_launcher():
    """A launcher that started a different number of ranks than --gpus says is an error, not a silently mislabelled line."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-cpu"], capture_output=True,
                         text=True, timeout=120, env=env, cwd=ROOT)
    assert out.returncode != 0 and "--gpus 2" in out.stderr


LANGEVIN_WORKER = r'''
import os, sys, json, torch
sys.path.insert(0, {root!r})
import torch.distributed as dist
from storm_amd import distributed as D
from storm_amd import ops
rank, world, local = D.init(backend="gloo")
g = torch.Generator().manual_seed(11)
sn_all, zn_all = torch.rand(5, generator=g) + 0.5, torch.rand(5, generator=g) + 0.5   # per-row norms of the unsharded batch
rows = [0, 1, 2] if rank == 0 else [3, 4]                                              # uneven shards
sn, zn = ops.langevin_group_norms(sn_all[rows], zn_all[rows], dist.group.WORLD)        # a REAL ProcessGroup
if rank == 0:
    print("RESULT " + json.dumps([float(sn), float(zn), float(sn_all.mean()), float(zn_all.mean())]))
D.barrier()
'''


def test_langevin_group_norms_two_ranks(tmp_path):
    """The sharded `langevin` corrector (langevin_group=): the 3-float all-reduce through torch.distributed over a real
    ProcessGroup (gloo here, RCCL on the GPUs) reproduces the unsharded batch means (correctors.py:53-55) on every rank."""
    import json
    script = tmp_path / "worker.py"
    script.write_text(LANGEVIN_WORKER.format(root=ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env={**os.environ, "OMP_NUM_THREADS": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    sn, zn, sn_ref, zn_ref = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0][7:])
    assert abs(sn - sn_ref) < 1e-6 and abs(zn - zn_ref) < 1e-6


def test_numa_pinning_helpers(tmp_path):
    """bench.py / enhancement.py pin a rank's launch thread to its GPU's NUMA node (8-GPU nodes): the cpulist parser, and the
    best-effort contract - no GPU / no topology in sysfs -> None, nothing changed."""
    from storm_amd import distributed as D
    assert D.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert D.parse_cpulist("5") == [5] and D.parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    assert D.pin_to_gpu_numa(0, sysfs=str(tmp_path)) is None
    assert os.sched_getaffinity(0) == before


@pytest.mark.gpu
def test_bench_under_the_driver_launch_line_with_an_rccl_group_of_one():
    """The driver starts the multi-GPU bench as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`.  On the one GPU a test box has, run exactly that line with N = 1 and --dist-world1, so that
    bench.py's own init_process_group("nccl", device_id=...), the barrier / all_gather_object fences of distributed.timed_steps and the
    leave-together barrier execute on ROCm / RCCL before the driver runs them at N = 8; and the --include-h2d variant of the step."""
    import json
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--N", "2", "--batch", "2",
           "--seconds", "1", "--no-cpu-baseline", "--no-roofline", "--dist-world1", "--include-h2d"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    r = json.loads(line[0])
    assert r["n_gpus"] == 1 and r["process_group"] == {"backend": "nccl", "world_size": 1}
    assert r["config"]["nfe_per_utterance"] == 4 and len(r["per_rank_s"]) == 1
    assert r["from_host"]["value_from_host_wavs"] > 0 and r["from_host"]["bytes_per_step"] == 2 * 2 * 16000 * 4
