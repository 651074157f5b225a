import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from localhgt_amd.engine import Engine
g = Engine(32, 3)
g.rng_seed(1); g.coder_generate(); g.set_reference_form(True)
g.synth_reference(1, 13000, 1_000_000)
g.synth_options(0, 20, 300)
g.synth_pairs(1, 2, 13000, 1_000_000, 0, 1_000_000, 150)
g.count_kmers()
for dbg in (4096 | (1 << 24), 1 << 24, 4096 | (1 << 24)):
    g.set_debug(dbg)
    t = time.time(); n = g.ref_scan(0.1, 0.08, 300_000_000); print("scan", dbg, round(time.time() - t, 2), "s", g.scan_info()["form"], g.slot_list(), flush=True)
