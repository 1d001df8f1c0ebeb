"""Clip-parallel sharding and result gather (one process per GPU, RCCL over xGMI; gloo on CPU in tests).

The path shards by independent clips; there is NO data-path collective.  Mirrors what the reference's
evaluation harness does: round-robin `islice(docs, rank, None, world_size)`
(src/lmms-eval/lmms_eval/utils.py:675-681) and `gather_object` of the results
(src/lmms-eval/lmms_eval/evaluator.py:519-546) - here one fixed-shape int32 all_gather per batch
(<= clips_per_rank x max_new_tokens ids + lengths: latency-bound, a few KiB).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def shard_clips(n_clips: int, rank: int, world: int) -> List[int]:
    """Clip indices owned by `rank`: i with i % world == rank (round-robin, like lmms-eval)."""
    return list(range(rank, n_clips, world))


def gather_results(local_ids: Sequence[Sequence[int]], max_new_tokens: int, clips_per_rank: int, device, group=None):
    """All-gather variable-length id lists.  Returns {rank: [ids...]} on every rank.

    Wire format: int32 [clips_per_rank, 1 + max_new_tokens] per rank (col 0 = length, -1 = no clip)."""
    import torch.distributed as dist
    live = dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    buf = torch.full((clips_per_rank, 1 + max_new_tokens), -1, dtype=torch.int32)
    for i, ids in enumerate(local_ids):
        n = min(len(ids), max_new_tokens)
        buf[i, 0] = n
        if n:
            buf[i, 1:1 + n] = torch.tensor(list(ids)[:n], dtype=torch.int32)
    buf = buf.to(device)
    if not live:
        bufs = [buf]
    else:
        bufs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf, group=group)
    out = {}
    for rk, b in enumerate(bufs):
        b = b.cpu()
        out[rk] = [b[i, 1:1 + int(b[i, 0])].tolist() for i in range(clips_per_rank) if int(b[i, 0]) >= 0]
    return out


def merge_round_robin(per_rank: dict, n_clips: int) -> List[Optional[List[int]]]:
    """Undo shard_clips: results in original clip order."""
    world = len(per_rank)
    out: List[Optional[List[int]]] = [None] * n_clips
    for rk, res in per_rank.items():
        for j, ids in enumerate(res):
            idx = rk + j * world
            if idx < n_clips:
                out[idx] = ids
    return out


class DistShim:
    """The two calls the lmms-eval harness makes on `lm.accelerator` when `lm.world_size > 1`
    (src/lmms-eval/lmms_eval/evaluator.py:426-428 `gather` of the per-rank instance count, :457 and :610-611
    `wait_for_everyone`), over torch.distributed - RCCL on the GPU ranks, gloo on CPU in tests.  The reference adaptor gets
    them from accelerate.Accelerator (lmms_eval/models/auroracap.py:70-134); this path has no model to `prepare`, so the
    process group is all it needs.  Rank / world / device come from the launcher's environment (torchrun / accelerate
    launch both export RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT)."""

    def __init__(self, backend: Optional[str] = None):
        import os
        import torch.distributed as dist
        self._dist = dist
        self.num_processes = int(os.environ.get("WORLD_SIZE", 1))
        self.process_index = int(os.environ.get("RANK", 0))
        self.local_process_index = int(os.environ.get("LOCAL_RANK", 0))
        backend = backend or os.environ.get("AURORA_DIST_BACKEND", "nccl")
        self.backend = backend
        if self.num_processes > 1 and not dist.is_initialized():
            if backend == "nccl":
                torch.cuda.set_device(self.local_process_index)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_process_index))
            else:
                dist.init_process_group(backend)
        self.device = torch.device("cuda", self.local_process_index) if backend == "nccl" else torch.device("cpu")

    is_main_process = property(lambda self: self.process_index == 0)
    is_local_main_process = property(lambda self: self.local_process_index == 0)

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        """accelerate's gather: tensors of every rank concatenated along dim 0 (a 0-dim tensor counts as [1])."""
        src = t.reshape(1) if t.dim() == 0 else t
        if self.num_processes == 1:
            return src.clone()
        home = src.device
        buf = src.to(self.device).contiguous()
        out = [torch.empty_like(buf) for _ in range(self.num_processes)]
        self._dist.all_gather(out, buf)
        return torch.cat(out, 0).to(home)

    def wait_for_everyone(self) -> None:
        if self.num_processes > 1:
            self._dist.barrier()


# ---------------------------------------------------------------------------------------------------
# Host placement.  On N ranks the host is the shared resource: every rank's Python thread enqueues ~1.6e4 kernel launches and ~2.6e2
# graph replays per cycle of 128 captions (bench.py reports `host.enqueue_s_per_cycle`).  Each rank is therefore bound to cores of the
# NUMA node its GPU hangs off, a disjoint slice per local rank, so that eight enqueue loops neither migrate nor share cores.
# (The reference harness leaves placement to `accelerate launch`; lmms_eval/evaluator.py:406-411 is one Python loop per rank too.)
# ---------------------------------------------------------------------------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]  (the format of sysfs `local_cpulist`)"""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def rank_cpu_slice(local_cpus: Sequence[int], sharers: Sequence[int], me: int, per_rank: int = 0) -> List[int]:
    """Cores of `local_cpus` for local rank `me` when the local ranks in `sharers` (sorted; `me` among them) hang off the same NUMA
    node: equal contiguous slices in rank order (at most `per_rank` cores each when > 0), never empty (falls back to all of them)."""
    cpus, sharers = sorted(local_cpus), sorted(sharers)
    if not cpus or me not in sharers:
        return list(cpus)
    n = len(cpus) // len(sharers)
    if n < 1:
        return list(cpus)
    k = sharers.index(me)
    sl = cpus[k * n:(k + 1) * n]
    return sl[:per_rank] if per_rank > 0 else sl


def gpu_local_cpus(device_index: int) -> List[int]:
    """CPUs of the NUMA node GPU `device_index` is attached to (sysfs `local_cpulist` of its PCI function); [] when unknown."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as fh:
            return parse_cpulist(fh.read())
    except Exception:                                          # noqa: BLE001 - no sysfs / no GPU: leave the affinity alone
        return []


def pin_rank_to_gpu_numa(local_rank: int, local_world: int, per_rank: int = 0) -> Optional[List[int]]:
    """Bind this process to its slice of the cores local to its GPU.  Returns the cores, or None if nothing was changed (unknown
    topology, or a launcher / cgroup already restricted the process to cores outside that node)."""
    import os
    if not hasattr(os, "sched_setaffinity"):
        return None
    mine = gpu_local_cpus(local_rank)
    if not mine:
        return None
    sharers = [r for r in range(local_world) if r == local_rank or gpu_local_cpus(r) == mine]
    want = set(rank_cpu_slice(mine, sharers, local_rank, per_rank)) & set(os.sched_getaffinity(0))
    if not want:
        return None
    os.sched_setaffinity(0, want)
    return sorted(want)
