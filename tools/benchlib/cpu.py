"""The CPU legs of bench.py, on a bounded sample of the same synthetic generator (files on disk, shared by all three):
  reference  the COMPILED reference (oracle/_ref/extract_ref_raw: /root/reference/src/extract_ref_normal_peak.cpp built -O2 by
             oracle/build_ref.sh, unmodified), `-t 10` as `localhgt bkp` passes it, started in the background at the very beginning
             of the run (host cores only) and joined at the end
  port       the C restatement (oracle/lhgt_oracle.c) on all host cores -- the checker: the GPU runs the very same files and the
             interval files must be identical (`identical_to_gpu`, the run fails otherwise)
The oracle and the reference binary are test infrastructure: timed and compared here, never on the product path."""
import os
import re
import subprocess
import sys
import threading
import time

from . import ROOT

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_raw")
CPU_K, CPU_E, CPU_CONTIGS, CPU_CONTIG_LEN = 32, 3, 20, 1_000_000


def usable_cpus():
    """CPUs this process may use at once: the hardware threads, cut to the affinity mask and to the cgroup's quota where one is set
    (the GPU boxes: cpu.max = 16 CPUs on a 256-thread host -- rounds 1-4 quoted "256 cores" for a leg that 16 CPUs ran)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = max(1, min(n, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def make_files(tmp, n_pairs, device=0):
    """(child process of bench.py: it uses the GPU) the FASTA / FASTQ files of the CPU legs and -- built by the product, byte-identical
    to the reference's (tests/test_gpu_parity.py) -- the index file, so that the reference's clock covers its phases, not read_ref"""
    from localhgt_amd.engine import Engine
    from .files import synth_files
    fa, f1, f2 = synth_files(tmp, CPU_K, CPU_E, CPU_CONTIGS, CPU_CONTIG_LEN, n_pairs, device)
    with Engine(CPU_K, CPU_E, device=device) as eng:
        eng.rng_seed(1)
        eng.coder_generate()
        eng.index_build(fa, f"{fa}.k{CPU_K}.h{CPU_E}.index.dat", fa + ".genome.len.txt")
    return fa, f1, f2


def make_files_in_child(tmp, n_pairs, timeout_s=300):
    """this process must not touch the GPU yet (the PMC children need it whole): the files come from a child"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--make-cpu-files", tmp, "--cpu-pairs", str(n_pairs)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout_s)
    paths = [os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq")]
    if res.returncode != 0 or not all(os.path.exists(p) for p in paths):
        raise RuntimeError(f"cpu files: rc {res.returncode}: {res.stdout.decode(errors='replace')[-300:]}")
    return paths


class ReferenceRun:
    """oracle/_ref/extract_ref_raw in the background; join() -> its record"""

    def __init__(self, fa, f1, f2, tmp, n_pairs, threads=10):
        self.n_pairs, self.threads = n_pairs, threads
        self.interval = os.path.join(tmp, "interval.reference.txt")
        self.log = os.path.join(tmp, "reference.stdout")
        argv = [REF_BIN, f1, f2, fa, self.interval, "0.1", "0.08", str(threads), str(CPU_K), "3000000", str(CPU_E), "1", "1"]
        self.t0 = time.time()
        self.t1 = None
        self.proc = subprocess.Popen(argv, stdout=open(self.log, "wb"), stderr=subprocess.STDOUT, cwd=tmp)
        self._waiter = threading.Thread(target=self._wait, daemon=True)
        self._waiter.start()

    def _wait(self):
        self.proc.wait()
        self.t1 = time.time()

    def join(self, timeout_s=600):
        self._waiter.join(timeout_s)
        if self.t1 is None:
            self.proc.kill()
            return {"error": f"the reference binary had not finished after {timeout_s} s"}
        out = open(self.log, errors="replace").read()
        count_s = re.findall(r"K-mer counting is finished. It costs (\d+) seconds", out)
        total_s = re.findall(r"Finish with time:\s*(\d+)", out)
        raw = re.findall(r"No\. of raw BKPs:\s*(\d+)", out)
        wall = self.t1 - self.t0
        if self.proc.returncode != 0 or not total_s:
            return {"error": f"rc {self.proc.returncode}", "tail": out[-300:]}
        return {"value": round(self.n_pairs / wall / 1e6, 6), "unit": "M paired-reads/s", "threads": self.threads, "wall_s": round(wall, 2),
                "own_clock_s": {"count": int(count_s[0]) if count_s else None, "total": int(total_s[0])},
                "raw_peaks_last_thread_line": int(raw[-1]) if raw else None,
                "what": f"oracle/_ref/extract_ref_raw (the reference's src/extract_ref_normal_peak.cpp, g++ -O2 as its Makefile:2, unmodified), -t {self.threads} "
                        f"(the CLI's default), whole process with a cached index, {self.n_pairs} pairs vs {CPU_CONTIGS} x {CPU_CONTIG_LEN} bp, "
                        "run in the background on host cores while the GPU legs ran; its own clock prints whole seconds "
                        "(counting includes the 4 GiB memset and the 5e7 rand() draws, E:1416-1422)",
                "interval": self.interval}


def port_and_gpu(fa, f1, f2, tmp, n_pairs, device, reference=None):
    """the restatement on all host cores, then the product on the SAME files: the baseline is only worth quoting if both sides
    computed the same thing -- the interval files must be the same, byte for byte"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api
    from conftest import build_oracle
    from localhgt_amd import extract_ref
    orc = oracle_api.Oracle(build_oracle())
    orc.set_pretouch(True)     # table page faults before the phase timers, like the reference's memsets do (E:1416, 1458)
    cores = usable_cpus()
    k, e = CPU_K, CPU_E
    cpu_iv, gpu_iv, gpu_iv10 = (os.path.join(tmp, x) for x in ("interval.port.txt", "interval.gpu.txt", "interval.gpu_t10.txt"))
    rc, rep = orc.run(f1, f2, fa, cpu_iv, 0.1, 0.08, cores, k, 3000000, e, 1, 1.0)
    if rc != 0:
        return {"error": f"oracle rc {rc}"}
    rep_g = extract_ref.run(extract_ref.Args(f1, f2, fa, gpu_iv, 0.1, 0.08, 1, k, 3000000, e, 1, 1.0), device=device, log=lambda *x: None)
    identical = open(cpu_iv, "rb").read() == open(gpu_iv, "rb").read() and int(rep.n_peaks) == rep_g["n_peaks"] and \
        int(rep.n_filtered) == rep_g["n_filtered"]
    t = rep.t_count + rep.t_scan + rep.t_vote
    out = {"value": round(n_pairs / t / 1e6, 6), "unit": "M paired-reads/s", "cores": cores, "kind": "port",
           "sample": f"{n_pairs} pairs x 150 bp vs {CPU_CONTIGS} x {CPU_CONTIG_LEN} bp synthetic contigs, k={k} e={e}, "
                     f"phases A+B+C of oracle/lhgt_oracle.c (index build excluded): "
                     f"A {rep.t_count:.2f}s B {rep.t_scan:.2f}s C {rep.t_vote:.2f}s",
           "raw_peaks": int(rep.n_peaks), "filtered_peaks": int(rep.n_filtered), "interval_lines": sum(1 for _ in open(cpu_iv)),
           "identical_to_gpu": bool(identical),
           "gpu_same_files": {"total_s": round(rep_g["total_s"], 3), "raw_peaks": rep_g["n_peaks"], "filtered_peaks": rep_g["n_filtered"],
                              "kernels_ms": round(rep_g["count_kernel_ms"] + rep_g["scan_kernel_ms"] + rep_g["vote_kernel_ms"], 1)}}
    if reference is not None:
        ref = dict(reference)
        iv = ref.pop("interval", None)
        if "error" not in ref and iv and os.path.exists(iv):
            # the product as the drop-in runs it for `-t 10`: the reference's thread partition, id ranges and sentinel lines emulated
            rep_10 = extract_ref.run(extract_ref.Args(f1, f2, fa, gpu_iv10, 0.1, 0.08, reference["threads"], k, 3000000, e, 1, 1.0), device=device, log=lambda *x: None)
            ref["identical_to_gpu"] = open(iv, "rb").read() == open(gpu_iv10, "rb").read()
            ref["gpu_same_files_total_s"] = round(rep_10["total_s"], 3)
        out["reference"] = ref
    return out
