#!/usr/bin/env python3
"""the packer on the host's CPUs against the packer that goes through the GPU's resident store (round 6, late): N pairs as FASTQ in the page cache,
seconds of either and whether the two files are the same.  usage: pack_host_time.py [pairs]"""
import os, sys, tempfile, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib.files import near_gpu, synth_files_sliced
from localhgt_amd import pack
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32_000_000
os.system("nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null")
tmp = tempfile.mkdtemp(prefix="lhgt_pkh_", dir="/tmp")
with near_gpu(0):
    fa, f1, f2 = synth_files_sliced(tmp, 32, 3, 100, 1_000_000, pairs, 0)
print(f"FASTQ: 2 x {os.path.getsize(f1) / 1e9:.2f} GB", flush=True)
res = {}
for name, host in (("host", True), ("gpu", False), ("host again", True)):
    out = os.path.join(tmp, name.replace(" ", "_") + ".lhgp")
    t0 = time.time()
    hdr = pack.pack(f1, f2, out, max_threads=10, host=host, log=lambda *a: None)
    dt = time.time() - t0
    h = hashlib.sha256()
    with open(out, "rb") as f:
        f.seek(hdr["data_offset"])
        while True:
            b = f.read(1 << 26)
            if not b: break
            h.update(b)
    res[name] = h.hexdigest()
    print(f"{name:10s}: packed in {dt:.1f} s = {pairs / dt / 1e6:.1f} M pairs/s, {os.path.getsize(out) / 1e9:.2f} GB, records sha256 {res[name][:16]}", flush=True)
    os.unlink(out)
print("same records:", len(set(res.values())) == 1)
