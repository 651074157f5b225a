"""Multi-GPU exchange for the sharded read phases: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

The reference has no counterpart (threads over shared arrays, E:1424-1507).  Reads are sharded
by blocks of the global pair ordinal, so sampling decisions do not depend on the shard.  Two
objects need reducing:

* the 2-bit count table after phase A: every rank holds min(3, c_r) per slot and needs
  min(3, sum_r c_r) = min(3, sum_r min(3, c_r)).  Done as a reduce-scatter by hand on the PACKED
  table -- all_to_all of 1/world slices, local saturating merge (HIP kernel), all_gather -- which
  moves 2*(world-1)/world of the packed bytes per GPU instead of 4x that for a u8 all-reduce;
* the per-peak vote counters after phase C: plain SUM all-reduce of u32 (clamped to 254 on export).
Phase B runs replicated on every rank (identical inputs -> identical peak ids), so there is no
peak exchange."""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch
import torch.distributed as dist


class _DevView:
    """zero-copy torch view of a raw device allocation"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, nbytes: int, device: int) -> torch.Tensor:
    return torch.as_tensor(_DevView(ptr, nbytes), device=f"cuda:{device}")


class GpuAdapter:
    """how the exchange touches an Engine's device buffers"""

    def counts_tensor(self, eng) -> torch.Tensor:
        p, n = eng.counts_buffer()
        return device_tensor(p, n, eng.device)

    def merge(self, eng, other: torch.Tensor, byte_offset: int):
        eng.counts_merge(other.data_ptr(), byte_offset, other.numel())

    def filter_tensor(self, eng) -> Optional[torch.Tensor]:
        p, n = eng.filter_buffer()
        return device_tensor(p, n, eng.device).view(torch.int32) if n else None

    def sync(self, eng):
        eng.synchronize()
        torch.cuda.synchronize(eng.device)


class Exchange:
    def __init__(self, rank: int, world: int, local_rank: int, backend: str, adapter=None, own_group: bool = True):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.adapter = adapter or GpuAdapter()
        self.own_group = own_group
        if own_group and not dist.is_initialized():
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        self.backend = backend

    @classmethod
    def from_env(cls, backend: Optional[str] = None, adapter=None):
        rank = int(os.environ["RANK"])
        world = int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        return cls(rank, world, local, backend or ("nccl" if torch.cuda.is_available() else "gloo"), adapter)

    def close(self):
        if self.own_group and dist.is_initialized():
            dist.destroy_process_group()

    def barrier(self):
        dist.barrier()

    def _dev(self):
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def broadcast_flag(self, flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self._dev())
        dist.broadcast(t, src=0)
        return bool(t.item())

    # ---- phase A: packed saturating reduce-scatter + all-gather
    def merge_counts(self, eng):
        table = self.adapter.counts_tensor(eng)          # uint8 view of the packed 2-bit table
        n = table.numel()
        w = self.world
        if n % (4 * w):
            raise ValueError(f"table of {n} bytes does not split into {w} word-aligned slices")
        sl = n // w
        self.adapter.sync(eng)
        recv = torch.empty_like(table)
        dist.all_to_all_single(recv, table)              # recv[j*sl:(j+1)*sl] = rank j's copy of MY slice
        self.adapter.sync(eng)                           # RCCL runs on torch's stream, the merge kernel on the engine's
        mine = self.rank * sl
        for j in range(w):
            if j != self.rank:
                self.adapter.merge(eng, recv[j * sl:(j + 1) * sl], mine)
        self.adapter.sync(eng)
        del recv
        dist.all_gather_into_tensor(table, table[mine:mine + sl].clone())
        self.adapter.sync(eng)

    # ---- phase C
    def sum_votes(self, eng):
        t = self.adapter.filter_tensor(eng)
        self.adapter.sync(eng)
        if t is not None and t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        self.adapter.sync(eng)
