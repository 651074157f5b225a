#!/usr/bin/env python3
"""phase B of configs[2] (index form unless PACKED=1) per LHGT_LITE_STRIDE (one process per stride: the knob is read once): ms, tiles treated exactly, digests"""
import sys, os, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from localhgt_amd.engine import Engine
    with Engine(32, 3) as g:
        g.rng_seed(1); g.coder_generate()
        if os.environ.get("PACKED"): g.set_reference_form(True)
        g.synth_reference(1, 13000, 1_000_000)
        g.synth_pairs(1, 2, 13000, 1_000_000, 0, 100_000_000, 150)
        g.counts_clear(); g.count_kmers()
        ms = []
        for _ in range(3):
            n = g.ref_scan(0.1, 0.08, 300_000_000); ms.append((g.phase_ms(1), g.phase_ms(3)))
        print(f"stride {os.environ.get('LHGT_LITE_STRIDE', '8'):>3}: B {min(m[0] for m in ms):7.1f} ms (probe kernel {min(m[1] for m in ms):7.1f})  {g.scan_info()}  peaks {n} loci {g.digest(g.DIGEST_LOCI)} peak_kmer {g.digest(g.DIGEST_PEAK_KMER)}", flush=True)
else:
    for st in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8", "10", "12", "16"]):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, LHGT_LITE_STRIDE=st))
