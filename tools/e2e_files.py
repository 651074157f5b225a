"""End-to-end timing of the file-based path (FASTA + FASTQ on disk -> interval file), i.e. what `extract_ref` does
inside pipeline.sh, including index build, FASTQ parsing, H2D and packing.  Usage: e2e_files.py [n_pairs] [n_contigs]"""
import os, sys, time, tempfile, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from localhgt_amd.engine import Engine
from localhgt_amd import extract_ref, synth

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n_contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 100
CL = 1_000_000
tmp = tempfile.mkdtemp(prefix="lhgt_e2e_", dir="/tmp")
t0 = time.time()
with Engine(32, 3) as eng:
    eng.rng_seed(1); eng.coder_generate()
    ref = eng.synth_reference(1, n_contigs, CL, want_host=True)
    m1, m2 = eng.synth_pairs(1, 2, n_contigs, CL, 0, n_pairs, 150, want_host=True)
fa, f1, f2 = (os.path.join(tmp, x) for x in ("ref.fa", "s.1.fq", "s.2.fq"))
with open(fa, "wb") as f:
    for c in range(n_contigs):
        f.write(b">g%d\n" % (c + 1)); f.write(ref[c * CL:(c + 1) * CL].tobytes()); f.write(b"\n")
lut = np.zeros(256, dtype=np.uint8); lut[ord("A")] = 0; lut[ord("C")] = 1; lut[ord("G")] = 2; lut[ord("T")] = 3; lut[ord("N")] = 4
synth.write_fastq(lut[m1].reshape(n_pairs, 150), f1, "1")
synth.write_fastq(lut[m2].reshape(n_pairs, 150), f2, "2")
print(f"inputs written in {time.time() - t0:.1f}s: fasta {os.path.getsize(fa) / 1e6:.0f} MB, fastq 2 x {os.path.getsize(f1) / 1e6:.0f} MB", flush=True)
out = {}
for run in ("index built in-run", "index cached"):
    t0 = time.time()
    rep = extract_ref.run(extract_ref.parse_argv([f1, f2, fa, os.path.join(tmp, "interval.txt"), "0.1", "0.08", "8", "32", "300000000", "3", "1", "1"]),
                          log=lambda *a: None)
    rep["wall_s"] = time.time() - t0
    out[run] = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rep.items()}
    print(run, json.dumps(out[run]), flush=True)
import shutil; shutil.rmtree(tmp)
