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
Phase B has two forms: replicated on every rank (identical inputs -> identical peak ids, no exchange;
the default while the index fits one GPU), or reference-sharded (`sharded_scan`): each rank scans a contiguous
contig range, new-peak counts are all-gathered to give every rank its id base (contig order = rank order, so ids
equal the sequential ones), and peak loci + (hash, id) registrations are all-gathered and replayed everywhere."""
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

    # reference-sharded phase B
    def scan_local(self, eng, hit_ratio, match_ratio):
        return eng.ref_scan_local(hit_ratio, match_ratio)

    def scan_emit(self, eng, id_base, n_new):
        pl, pr, n_regs = eng.ref_scan_emit(id_base)
        loci = device_tensor(pl, 8 * n_new, eng.device).view(torch.int32) if n_new else torch.empty(0, dtype=torch.int32, device=f"cuda:{eng.device}")
        regs = device_tensor(pr, 8 * n_regs, eng.device).view(torch.int32) if n_regs else torch.empty(0, dtype=torch.int32, device=f"cuda:{eng.device}")
        return loci, regs

    def peaks_install(self, eng, n_total, n_sel_total, max_peak, loci_all, regs_all):
        eng.peaks_install(n_total, n_sel_total, max_peak, loci_all.data_ptr() if loci_all.numel() else 0,
                          regs_all.data_ptr() if regs_all.numel() else 0, regs_all.numel() // 2)

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

    # ---- phase B, reference-sharded
    def all_gather_var(self, t: torch.Tensor) -> torch.Tensor:
        """concatenation over ranks (in rank order) of 1-D tensors of different lengths"""
        n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(self.world)]
        dist.all_gather(sizes, n)
        sizes = [int(x.item()) for x in sizes]
        m = max(sizes)
        if m == 0:
            return t
        padded = torch.zeros(m, dtype=t.dtype, device=t.device)
        padded[:t.numel()] = t
        out = torch.empty(m * self.world, dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, padded)
        return torch.cat([out[r * m:r * m + sizes[r]] for r in range(self.world)])

    def sharded_scan(self, eng, hit_ratio: float, match_ratio: float, max_peak: int) -> int:
        """phase B when every rank holds only its contig shard of the index; returns the global raw peak count"""
        n_new, n_sel = self.adapter.scan_local(eng, hit_ratio, match_ratio)
        mine = torch.tensor([n_new, n_sel], dtype=torch.int64, device=self._dev())
        allc = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allc, mine)
        news = [int(c[0].item()) for c in allc]
        id_base = sum(news[:self.rank])
        n_total, n_sel_total = sum(news), sum(int(c[1].item()) for c in allc)
        loci, regs = self.adapter.scan_emit(eng, id_base, n_new)
        self.adapter.sync(eng)
        loci_all = self.all_gather_var(loci)
        regs_all = self.all_gather_var(regs)
        self.adapter.sync(eng)
        self.adapter.peaks_install(eng, n_total, n_sel_total, max_peak, loci_all, regs_all)
        self.adapter.sync(eng)
        return n_total

    # ---- phase C
    def sum_votes(self, eng):
        t = self.adapter.filter_tensor(eng)
        self.adapter.sync(eng)
        if t is not None and t.numel():
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        self.adapter.sync(eng)
