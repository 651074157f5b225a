#!/usr/bin/env python3
"""What N ranks sharing ONE host can ingest: the host side of the FASTQ loader alone -- no GPU in the timed part --
run as 1 / 2 / 4 / 8 processes on one pair of files, the way the ranks of a multi-GPU `extract_ref` run it (localhgt_amd/dist.py:
every rank counts the lines of its 1/N of both files, the pieces are exchanged, every rank parses its contiguous 1/N of fq1's
chunks against the whole plan: lhgt_fastq_plan_part + the planned parse), and for one process also the single pass that
`extract_ref` takes on one GPU (host_fastq_stream.cpp).  Aggregate M pairs/s = pairs of the files / (slowest rank's line count +
slowest rank's parse).  The consumer only hands slabs back (lhgt_fastq_parse_rate), so this is the ceiling the host sets for the
GPUs, not an end-to-end figure.

usage: ingest_scaling.py [n_pairs=32000000] [out.txt]      (writes the files with the device generator first: needs the GPU for that)
       ingest_scaling.py --files fq1 fq2 [packed.lhgp] [out.txt]
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                  # noqa: E402
from localhgt_amd import _lib                        # noqa: E402

CHUNK = 2 << 20


def plan_part(h, path, part, parts, chunk):
    n, total = C.c_long(0), C.c_long(0)
    assert h.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, None, None, 0, C.byref(n), C.byref(total), None) == 0
    st, cnt = np.zeros(n.value, np.uint64), np.zeros(n.value, np.int64)
    rc = h.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, st.ctypes.data_as(C.POINTER(C.c_uint64)), cnt.ctypes.data_as(C.POINTER(C.c_long)),
                                n.value, C.byref(n), C.byref(total), None)
    assert rc == 0
    return st, cnt


def worker(argv):
    """--worker step fq1 fq2 part parts threads emulate workdir"""
    step, fq1, fq2, part, parts, threads, emulate, work = argv[0], argv[1], argv[2], int(argv[3]), int(argv[4]), int(argv[5]), int(argv[6]), argv[7]
    os.environ["LHGT_INGEST_THREADS"] = str(threads)
    h = _lib.load(require_gpu=False)
    t0 = time.time()
    if step == "plan":
        c1, c2 = C.c_long(0), C.c_long(0)
        assert h.lhgt_fastq_pair_chunk_bytes(fq1.encode(), fq2.encode(), C.byref(c1), C.byref(c2)) == 0      # fq2 in as many chunks as fq1
        a = plan_part(h, fq1, part, parts, c1.value)
        b = plan_part(h, fq2, part, parts, c2.value)
        dt = time.time() - t0
        np.savez(os.path.join(work, f"plan_{parts}_{part}.npz"), s1=a[0], c1=a[1], s2=b[0], c2=b[1])
        print(json.dumps({"step": "plan", "part": part, "s": dt}), flush=True)
        return
    seen, kept, bases, secs = C.c_long(0), C.c_long(0), C.c_long(0), C.c_double(0)
    if step == "packed":          # fq1 = the packed sample: the loader's host side is pread of its records, nothing else
        from localhgt_amd import pack
        hdr = pack.read_header(fq1)
        rc = h.lhgt_packed_read_rate(fq1.encode(), hdr.data_offset, hdr.stride, hdr.n_pairs, part, parts, threads, C.byref(secs))
        print(json.dumps({"step": "packed", "part": part, "rc": rc, "s": secs.value, "pairs": hdr.n_pairs}), flush=True)
        return
    if step == "single":
        os.environ["LHGT_INGEST_STREAM"] = "1"
        rc = h.lhgt_fastq_parse_rate(fq1.encode(), fq2.encode(), 100.0, None, threads, CHUNK, emulate, None, None, 0, None, None, 0, 0, 1,
                                     C.byref(seen), C.byref(kept), C.byref(bases), C.byref(secs), None)
        why = C.create_string_buffer(200)
        path = h.lhgt_ingest_last_path(why, 200)
        print(json.dumps({"step": "single", "rc": rc, "s": secs.value, "kept": kept.value, "bases": bases.value, "single_pass": path, "why": why.value.decode()}), flush=True)
        return
    z = [np.load(os.path.join(work, f"plan_{parts}_{p}.npz")) for p in range(parts)]
    s1, c1 = np.concatenate([x["s1"] for x in z]), np.concatenate([x["c1"] for x in z])
    s2, c2 = np.concatenate([x["s2"] for x in z]), np.concatenate([x["c2"] for x in z])
    u64, lp = C.POINTER(C.c_uint64), C.POINTER(C.c_long)
    rc = h.lhgt_fastq_parse_rate(fq1.encode(), fq2.encode(), 100.0, None, threads, CHUNK, emulate, s1.ctypes.data_as(u64), c1.ctypes.data_as(lp), len(s1),
                                 s2.ctypes.data_as(u64), c2.ctypes.data_as(lp), len(s2), part, parts, C.byref(seen), C.byref(kept), C.byref(bases), C.byref(secs), None)
    print(json.dumps({"step": "parse", "part": part, "rc": rc, "s": secs.value, "kept": kept.value, "bases": bases.value, "path": h.lhgt_ingest_last_path(None, 0)}), flush=True)


def run_step(step, fq1, fq2, parts, threads, emulate, work):
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", step, fq1, fq2, str(p), str(parts), str(threads), str(emulate), work],
                              stdout=subprocess.PIPE, text=True) for p in range(parts)]
    out = []
    for p in procs:
        o, _ = p.communicate(timeout=600)
        assert p.returncode == 0, o
        out.append(json.loads(o.strip().splitlines()[-1]))
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "--worker":
        return worker(args[1:])
    work = tempfile.mkdtemp(prefix="lhgt_ingest_", dir="/tmp")
    packed = None
    if args and args[0] == "--files":
        fq1, fq2 = args[1], args[2]
        out_path = args[3] if len(args) > 3 else None
        if out_path and out_path.endswith(".lhgp"):
            packed, out_path = out_path, (args[4] if len(args) > 4 else None)
    else:
        import bench
        n_pairs = int(args[0]) if args else 32_000_000
        out_path = args[1] if len(args) > 1 else None
        t0 = time.time()
        _, fq1, fq2 = bench.synth_files_sliced(work, 32, 3, 100, 1_000_000, n_pairs, 0)
        print(f"files written in {time.time() - t0:.0f} s", flush=True)
        from localhgt_amd import pack
        packed = os.path.join(work, "s.lhgp")
        pack.pack(fq1, fq2, packed, max_threads=10)
    size = os.path.getsize(fq1) + os.path.getsize(fq2)
    n_pairs = sum(1 for _ in open(fq1, "rb")) // 4 if size < 1 << 28 else os.path.getsize(fq1) // 318     # bench's records are 318 bytes
    hw = os.cpu_count() or 8
    lines = [f"host-side FASTQ ingest alone (no GPU in the timed part), {n_pairs} pairs = {size / 1e9:.1f} GB of text in the page cache, {hw} hardware threads; "
             f"aggregate = pairs / (slowest rank's line count + slowest rank's parse)",
             "processes x threads each | line count s (max over ranks) | parse s (max) | aggregate M pairs/s | GB/s of text"]
    quota = 0
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = 0 if q == "max" else int(q) // int(per)
    except (OSError, ValueError):
        pass
    lines[0] += f"; CPU quota of this container: {quota or 'none'} CPUs"
    base = quota or min(hw, 48)
    for emulate in (1, 10):
        for rep in range(2):
            for t in sorted({max(2, base // 2), base, base * 3 // 2}):
                r = run_step("single", fq1, fq2, 1, t, emulate, work)[0]
                lines.append(f"1 x {t:3d}, ONE PASS (-t {emulate:2d}{' emulated' if emulate > 1 else ''}) | - | {r['s']:.3f} | {n_pairs / r['s'] / 1e6:.1f} | {size / r['s'] / 1e9:.1f}"
                             f"   [single pass taken: {r['single_pass']} {r['why']}; kept {r['kept']}]")
                print(lines[-1], flush=True)
    for parts in (1, 2, 4, 8):
        for t in sorted({max(2, base // parts), max(2, base * 3 // 2 // parts)}):
            for rep in range(2):
                a = run_step("plan", fq1, fq2, parts, t, 1, work)
                b = run_step("parse", fq1, fq2, parts, t, 1, work)
                assert all(x["rc"] == 0 for x in b) and sum(x["kept"] for x in b) == n_pairs, b
                tp, tq = max(x["s"] for x in a), max(x["s"] for x in b)
                how = {2: "columns on the plans' line numbers", 0: "chunk loop"}.get(b[0].get("path"), "?")
                lines.append(f"{parts} x {t:3d}, planned (count, exchange, parse 1/{parts}: {how}) | {tp:.3f} | {tq:.3f} | {n_pairs / (tp + tq) / 1e6:.1f} | {size / (tp + tq) / 1e9:.1f}")
                print(lines[-1], flush=True)
    if packed:
        psize = os.path.getsize(packed)
        lines.append(f"the same pairs as a packed sample (localhgt_pack: {psize / 1e9:.2f} GB, {size / psize:.1f} x less than the text): the host side of its load is pread of "
                     f"1/N of the records into two buffers (lhgt_packed_read_rate; the loader's are pinned)")
        for parts in (1, 2, 4, 8):
            for t in sorted({max(2, base // parts), max(2, base * 3 // 2 // parts)}):
                for rep in range(2):
                    b = run_step("packed", packed, "-", parts, t, 1, work)
                    assert all(x["rc"] == 0 for x in b), b
                    tq = max(x["s"] for x in b)
                    lines.append(f"{parts} x {t:3d}, packed records (pread 1/{parts}) | - | {tq:.3f} | {n_pairs / tq / 1e6:.1f} | {psize / tq / 1e9:.1f} (of records)")
                    print(lines[-1], flush=True)
    if out_path:
        open(out_path, "w").write("\n".join(lines) + "\n")
    import shutil
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
