"""Multi-GPU exchange for the sharded read phases: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests and when ranks share a GPU).

The reference has no counterpart (threads over shared arrays, E:1424-1507).  Every rank parses a contiguous run of the FASTQs
(`fastq_plan`: each rank counts the lines of its share of the bytes, the pieces are all-gathered, so the global line numbers --
and with them the sampling decisions, E:1037-1044 -- do not depend on the split).  Two objects need reducing:

* the 2-bit count table after phase A: every rank holds min(3, c_r) per slot and needs
  min(3, sum_r c_r) = min(3, sum_r min(3, c_r)).  Done as a reduce-scatter by hand on the PACKED
  table -- all_to_all of 1/world slices, local saturating merge (HIP kernel), all_gather -- which
  moves 2*(world-1)/world of the packed bytes per GPU instead of 4x that for a u8 all-reduce;
* the per-peak vote counters after phase C: plain SUM all-reduce of u32 (clamped to 254 on export).
Phase B has two forms: replicated on every rank (identical inputs -> identical peak ids, no exchange;
the default while the index fits one GPU), or reference-sharded (`sharded_scan`): each rank scans a contiguous
contig range, new-peak counts are all-gathered to give every rank its id base (contig order = rank order, so ids
equal the sequential ones), and peak loci + (hash, id) registrations are all-gathered and replayed everywhere.

Backends.  "nccl": the collectives run on the engine's device buffers (zero-copy torch views).  "gloo" with a GPU engine
(more ranks than GPUs, e.g. two ranks on the one GPU of a test box, where RCCL refuses duplicate devices): the same Exchange
methods stage every collective through host memory -- device buffer -> host -> gloo -> device -- so the whole N > 1 path of
`extract_ref` runs and is tested wherever one GPU exists."""
from __future__ import annotations

import os
import sys
from typing import Optional

import numpy as np
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

    def group_counts(self, eng):
        return eng.ref_scan_group_counts()

    def set_group_totals(self, eng, totals, max_peak):
        return eng.set_group_totals(totals, max_peak)

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
    def __init__(self, rank: int, world: int, local_rank: int, backend: str, adapter=None, own_group: bool = True,
                 device: Optional[int] = None):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.device = local_rank if device is None else device     # the GPU this rank's engine lives on
        self.adapter = adapter or GpuAdapter()
        self.own_group = own_group
        if own_group and not dist.is_initialized():       # (from_env has made the group already when it had to ask the ranks for the backend)
            if backend == "nccl":
                torch.cuda.set_device(self.device)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        self.backend = backend
        self.staged_bytes = 0       # bytes that went through host memory (gloo with device buffers)
        # bytes THIS rank sent + received in each exchange since the last reset (bench.py: GB/s per exchange, so that the first run on
        # a real 8-GPU node explains itself)
        self.moved = {"merge_counts": 0, "sharded_scan": 0, "sum_votes": 0}

    @classmethod
    def from_env(cls, backend: Optional[str] = None, adapter=None):
        rank = int(os.environ["RANK"])
        world = int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        device = local
        if backend is None:
            backend = os.environ.get("LHGT_DIST_BACKEND")
        n_dev = torch.cuda.device_count()
        if backend is None:
            # RCCL wants one GPU per rank; ranks that share a GPU exchange through host memory over gloo.  What ONE rank sees does
            # not settle it: without LOCAL_WORLD_SIZE (srun, mpirun, hand-set RANK / WORLD_SIZE) rank 0 of two ranks on a one-GPU
            # node sees nothing wrong (LOCAL_RANK 0 < 1 GPU) while rank 1 does, and two ranks that pick different backends hang in
            # their first collective.  So the group is created with both backends -- "cpu:gloo,cuda:nccl": CPU tensors
            # travel over gloo, device tensors over RCCL, whose communicator is only made by the first device collective -- every
            # rank says what it sees, and the maximum decides for all.
            mine = n_dev == 0 or local >= n_dev
            for var in ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE"):
                v = os.environ.get(var, "")
                if v.isdigit() and int(v) > n_dev:
                    mine = True
            if not dist.is_initialized():
                if n_dev and not mine:
                    torch.cuda.set_device(local % n_dev)
                init_backend = os.environ.get("LHGT_DIST_INIT_BACKEND") or ("cpu:gloo,cuda:nccl" if n_dev else "gloo")   # (the variable: CPU tests that fake a GPU count)
                dist.init_process_group(backend=init_backend, rank=rank, world_size=world)
            flag = torch.tensor([1 if mine else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            shared = bool(flag.item())
            backend = "gloo" if shared else "nccl"
            if shared and n_dev and rank == 0:
                print(f"localhgt_amd.dist: more ranks than GPUs on a node ({n_dev} here): collectives staged through host memory over gloo "
                      f"(results identical, exchanges much slower); LHGT_DIST_BACKEND=nccl to insist on RCCL", file=sys.stderr, flush=True)
        if n_dev:
            device = local % n_dev
        return cls(rank, world, local, backend, adapter, device=device)

    def close(self):
        if self.own_group and dist.is_initialized():
            dist.destroy_process_group()

    def barrier(self):
        """never through NCCL when the ranks may share a GPU: with the gloo backend (and its 'cpu:gloo,cuda:nccl' group) the
        barrier is an all-reduce of one host integer -- dist.barrier() picks its device by the torch version's own rule"""
        if self.backend == "gloo":
            dist.all_reduce(torch.zeros(1, dtype=torch.int32))
        else:
            dist.barrier(device_ids=[self.device])

    def _dev(self):
        return torch.device("cuda", self.device) if self.backend == "nccl" else torch.device("cpu")

    # ---- collectives; with gloo a device tensor travels through host memory
    def _staged(self, t: torch.Tensor) -> bool:
        return self.backend != "nccl" and t.is_cuda

    def _all_to_all(self, out: torch.Tensor, inp: torch.Tensor, out_splits=None, in_splits=None):
        if self._staged(inp):
            h_in = inp.cpu()
            h_out = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_to_all_single(h_out, h_in, out_splits, in_splits)
            out.copy_(h_out)
            self.staged_bytes += (h_in.numel() + h_out.numel()) * h_in.element_size()
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits)

    def _all_gather_into(self, out: torch.Tensor, inp: torch.Tensor):
        if self._staged(inp):
            h_in = inp.cpu()
            h_out = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_gather_into_tensor(h_out, h_in)
            out.copy_(h_out)
            self.staged_bytes += (h_in.numel() + h_out.numel()) * h_in.element_size()
        else:
            dist.all_gather_into_tensor(out, inp)

    def _all_reduce_sum(self, t: torch.Tensor):
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            self.staged_bytes += 2 * h.numel() * h.element_size()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def _gather_small(self, values) -> list:
        """every rank's list of ints, in rank order"""
        mine = torch.tensor(list(values), dtype=torch.int64, device=self._dev())
        allv = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(allv, mine)
        return [[int(x) for x in v.tolist()] for v in allv]

    def broadcast_flag(self, flag: bool) -> bool:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=self._dev())
        dist.broadcast(t, src=0)
        return bool(t.item())

    def agree(self, value: int) -> int:
        """the maximum of an int over the ranks (used to take a decision -- fall back, retry -- on every rank or on none)"""
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    # ---- ingest: the line structure of a FASTQ, each rank counting 1/world of it
    def fastq_plan(self, eng, path: str, want_len_sums: bool = False, chunk=None):
        """(start u64[n], n_lines i64[n][, len_sums i64[n, 4]]) for ALL chunks of the file (chunk: bytes per chunk, default the loader's)"""
        st, cn, sums = eng.fastq_plan_part(path, self.rank, self.world, want_len_sums, chunk=chunk)
        cols = [st.view(np.int64), cn] + ([sums[:, r].copy() for r in range(4)] if want_len_sums else [])
        flat = np.stack(cols, axis=1).reshape(-1) if len(st) else np.zeros(0, dtype=np.int64)
        allv = self.all_gather_var(torch.from_numpy(flat).to(self._dev())).cpu().numpy().reshape(-1, len(cols))
        start, counts = np.ascontiguousarray(allv[:, 0]).view(np.uint64), np.ascontiguousarray(allv[:, 1])
        return (start, counts, np.ascontiguousarray(allv[:, 2:6])) if want_len_sums else (start, counts)

    # ---- phase A: packed saturating reduce-scatter + all-gather
    def merge_counts(self, eng):
        table = self.adapter.counts_tensor(eng)          # uint8 view of the packed 2-bit table
        n = table.numel()
        w = self.world
        if n % 4:
            raise ValueError(f"table of {n} bytes is not made of 32-bit words")
        sl = 4 * -(-(n // 4) // w)                        # word-aligned slices; the last ones are shorter (or empty) when w does not divide
        sizes = [max(0, min(sl, n - j * sl)) for j in range(w)]
        mine, mine_n = self.rank * sl, sizes[self.rank]
        self.adapter.sync(eng)
        recv = torch.empty(w * mine_n, dtype=table.dtype, device=table.device)
        self._all_to_all(recv, table, [mine_n] * w, sizes)   # recv[j*mine_n:(j+1)*mine_n] = rank j's copy of MY slice
        self.moved["merge_counts"] += (n - mine_n) + (w - 1) * mine_n + (w - 1) * sl + (n - mine_n)   # all-to-all out + in, all-gather out + in
        self.adapter.sync(eng)                           # RCCL runs on torch's stream, the merge kernel on the engine's
        for j in range(w):
            if j != self.rank and mine_n:
                self.adapter.merge(eng, recv[j * mine_n:(j + 1) * mine_n], mine)
        self.adapter.sync(eng)
        del recv
        if len(set(sizes)) == 1:
            self._all_gather_into(table, table[mine:mine + sl].clone())
        else:                                            # unequal slices: gather them padded, keep the table's n bytes
            padded = torch.zeros(sl, dtype=table.dtype, device=table.device)
            padded[:mine_n] = table[mine:mine + mine_n]
            out = torch.empty(w * sl, dtype=table.dtype, device=table.device)
            self._all_gather_into(out, padded)
            table.copy_(out[:n])
        self.adapter.sync(eng)

    # ---- phase B, reference-sharded
    def all_gather_var(self, t: torch.Tensor) -> torch.Tensor:
        """concatenation over ranks (in rank order) of 1-D tensors of different lengths"""
        sizes = [v[0] for v in self._gather_small([t.numel()])]
        m = max(sizes)
        if m == 0:
            return t
        padded = torch.zeros(m, dtype=t.dtype, device=t.device)
        padded[:t.numel()] = t
        out = torch.empty(m * self.world, dtype=t.dtype, device=t.device)
        self._all_gather_into(out, padded)
        return torch.cat([out[r * m:r * m + sizes[r]] for r in range(self.world)])

    def sharded_scan(self, eng, hit_ratio: float, match_ratio: float, max_peak: int, emulated_threads: int = 1) -> int:
        """phase B when every rank holds only its contig shard of the index; returns the global raw peak count.
        emulated_threads > 1 (the reference's -t N, lhgt_set_thread_emulation): the contig groups of split_ref cut across the
        ranks' shards, so the per-group peak counts are summed over the ranks first -- they fix each thread's id range, the
        sentinel lines of the interval file and whether a peak holds the invisible id 0 (first_id)."""
        from ._lib import LocalHGTError
        err, mine = None, 0
        try:
            n_new, n_sel = self.adapter.scan_local(eng, hit_ratio, match_ratio)
        except LocalHGTError as e:                         # 1: only the -t N emulation refuses (the caller falls back to -t 1 -- on EVERY rank), 2: anything else
            err, mine = e, (1 if e.code == 9 else 2)   # LHGT_E_EMULATION
        except Exception as e:                             # noqa: BLE001 -- a rank that failed alone must not leave the others in the gather below
            err, mine = e, 2
        worst = self.agree(mine)
        if worst:                                          # the same class and code on every rank, so that all fall back or all stop
            if mine == worst:
                raise err
            raise LocalHGTError(9 if worst == 1 else 5, "-t N emulation: refused on another rank" if worst == 1 else
                                f"reference-sharded scan failed on another rank (rank {self.rank} stops with it)")
        allc = self._gather_small([n_new, n_sel])
        news = [c[0] for c in allc]
        first_id = 0
        if emulated_threads > 1:
            groups = self._gather_small(self.adapter.group_counts(eng))
            first_id = self.adapter.set_group_totals(eng, [sum(g[j] for g in groups) for j in range(emulated_threads)], max_peak)
        id_base = first_id + sum(news[:self.rank])
        n_total, n_sel_total = sum(news), sum(c[1] for c in allc)
        loci, regs = self.adapter.scan_emit(eng, id_base, n_new)
        self.adapter.sync(eng)
        loci_all = self.all_gather_var(loci)
        regs_all = self.all_gather_var(regs)
        mine_words, all_words = loci.numel() + regs.numel(), loci_all.numel() + regs_all.numel()
        self.moved["sharded_scan"] += 4 * ((self.world - 1) * mine_words + (all_words - mine_words))      # all-gather: mine to the others, theirs to me
        if first_id:                                     # no peak holds id 0: the loci table starts with an empty row
            loci_all = torch.cat([torch.zeros(2 * first_id, dtype=loci_all.dtype, device=loci_all.device), loci_all])
        self.adapter.sync(eng)
        self.adapter.peaks_install(eng, n_total + first_id, n_sel_total, max_peak + first_id, loci_all, regs_all)
        self.adapter.sync(eng)
        return n_total

    # ---- phase C
    def sum_votes(self, eng):
        t = self.adapter.filter_tensor(eng)
        self.adapter.sync(eng)
        if t is not None and t.numel():
            self._all_reduce_sum(t)
            self.moved["sum_votes"] += 2 * (2 * (self.world - 1) * t.numel() * t.element_size()) // self.world   # ring all-reduce: 2 (w - 1) / w of the buffer out, the same in
        self.adapter.sync(eng)
