"""Synthetic FASTA / FASTQ files (bases from the device-side generator k_synth.hip, so files and resident workloads agree)."""
import os
import shutil


def write_fasta(path, ref, n_contigs, contig_len):
    with open(path, "wb") as f:
        for c in range(n_contigs):
            f.write(b">g%d\n" % (c + 1))
            f.write(ref[c * contig_len:(c + 1) * contig_len].tobytes())
            f.write(b"\n")


def write_fastq(path, mate, n, L, suffix):
    """4-line records `@r<9 digits>/<suffix>`, vectorised (4 M records in a second or two)"""
    import numpy as np
    a = mate.reshape(n, L)
    ids = np.char.zfill(np.arange(n).astype("U9"), 9)
    head = np.char.add(np.char.add("@r", ids), "/" + suffix)
    hb = np.frombuffer("".join(head.tolist()).encode(), dtype=np.uint8).reshape(n, -1)
    hl = hb.shape[1]
    rec = np.empty((n, hl + 1 + L + 1 + 2 + L + 1), dtype=np.uint8)
    rec[:, :hl] = hb
    rec[:, hl] = 10
    rec[:, hl + 1: hl + 1 + L] = a
    rec[:, hl + 1 + L] = 10
    rec[:, hl + 2 + L] = ord("+")
    rec[:, hl + 3 + L] = 10
    rec[:, hl + 4 + L: hl + 4 + 2 * L] = ord("I")
    rec[:, hl + 4 + 2 * L] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def synth_files(tmp, k, e, n_contigs, contig_len, n_pairs, device, seed_ref=1, seed_reads=2, snp_permille=0, sample_contigs=0):
    from localhgt_amd.engine import Engine
    with Engine(k, e, device=device) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        ref = eng.synth_reference(seed_ref, n_contigs, contig_len, want_host=True)
        if snp_permille or sample_contigs:
            eng.synth_options(snp_permille, 20, sample_contigs)
        m1, m2 = eng.synth_pairs(seed_ref, seed_reads, n_contigs, contig_len, 0, n_pairs, 150, want_host=True)
    fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
    write_fasta(fa, ref, n_contigs, contig_len)
    write_fastq(f1, m1, n_pairs, 150, "1")
    write_fastq(f2, m2, n_pairs, 150, "2")
    return fa, f1, f2


class near_gpu:
    """while the input files are written: this process on the CPUs of the GPU's NUMA node, so that their pages land in that node's page
    cache -- the container has a CPU quota, not a CPU set, and where a file's pages sit decides a good part of the from-files rate
    (Which socket).  Does nothing where the node cannot be found."""

    def __init__(self, device=0, away=False):
        self.device, self.old, self.away = device, None, away

    def __enter__(self):
        try:
            if self.device < 0:
                return self
            import torch
            pr = torch.cuda.get_device_properties(self.device)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cpus = set()
            for tok in open(f"/sys/bus/pci/devices/{bdf}/local_cpulist").read().strip().split(","):
                a, _, b = tok.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            cur = os.sched_getaffinity(0)
            pick = cur - cpus if self.away else cur & cpus
            if len(pick) >= 4 and pick != cur:
                os.sched_setaffinity(0, pick)
                self.old = cur
        except Exception:
            pass
        return self

    def __exit__(self, *exc):
        if self.old:
            os.sched_setaffinity(0, self.old)
        return False


def synth_files_sliced(tmp, k, e, n_contigs, contig_len, n_pairs, device, slice_pairs=4_000_000):
    """like synth_files for inputs of tens of GB: the pairs generated and written slice by slice"""
    from localhgt_amd.engine import Engine
    fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
    with Engine(k, e, device=device) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        write_fasta(fa, eng.synth_reference(1, n_contigs, contig_len, want_host=True), n_contigs, contig_len)
        for path in (f1, f2):
            open(path, "wb").close()
        for p0 in range(0, n_pairs, slice_pairs):
            n = min(slice_pairs, n_pairs - p0)
            eng.pairs_clear()
            m1, m2 = eng.synth_pairs(1, 2, n_contigs, contig_len, p0, n, 150, want_host=True)
            for path, m, suf in ((f1, m1, "1"), (f2, m2, "2")):
                part = path + ".part"
                write_fastq(part, m, n, 150, suf)      # read ids restart per slice: the path looks at the first one only
                with open(path, "ab") as dst, open(part, "rb") as src:
                    shutil.copyfileobj(src, dst, 1 << 24)
                os.remove(part)
    return fa, f1, f2
