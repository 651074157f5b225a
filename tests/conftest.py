import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: multi-GiB CPU tables (k=32 oracle runs)")


def build_oracle():
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liblhgt_oracle.so")
    src = os.path.join(odir, "lhgt_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", odir], check=True, capture_output=True)
    return so


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU restatement (oracle/lhgt_oracle.c). Checker only."""
    import oracle_api
    return oracle_api.Oracle(build_oracle())


@pytest.fixture(scope="session")
def case_inputs(tmp_path_factory):
    """Materialise a seeded case once per session and verify the input digests against the golden meta."""
    import json
    import cases
    cache = {}

    def get(name):
        if name in cache:
            return cache[name]
        d = str(tmp_path_factory.mktemp(name))
        fa, f1, f2 = cases.materialise(cases.CASES[name], d)
        meta = json.load(open(os.path.join(cases.GOLDEN_DIR, name, "meta.json")))
        for key, p in (("ref.fa", fa), ("s.1.fq", f1), ("s.2.fq", f2)):
            assert cases.sha256_file(p) == meta["sha256"][key], f"generator drift: {name}/{key}"
        cache[name] = (fa, f1, f2, meta)
        return cache[name]

    return get


# ---------------------------------------------------------------------------------------------- the reference binary at k = 32 (GPU sessions)
# tests/test_gpu_refbinary.py compares the product with oracle/_ref/extract_ref_z on 20 x 1 Mbp / 400 000 pairs at k = 32.  The
# reference needs about two minutes per run (4 + 16 GiB of tables, 5*10^7 rand() calls, a single-threaded index build), so its two
# runs (-t 1; -t 10 with its threads in creation order) are started at the beginning of a `-m gpu` session and joined by the test.
REFBIN_K, REFBIN_E, REFBIN_NC, REFBIN_PAIRS = 32, 3, 20, 400_000
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "extract_ref_z")
REF_SHIM = os.path.join(ROOT, "oracle", "_ref", "libseqthreads.so")


def refbin_copy_inputs(src, dst):
    import shutil
    os.makedirs(dst)
    shutil.copy(os.path.join(src, "ref.fa"), os.path.join(dst, "ref.fa"))
    for f in ("s.1.fq", "s.2.fq"):
        os.symlink(os.path.join(src, f), os.path.join(dst, f))
    return dst


@pytest.fixture(scope="session")
def refbin_big(tmp_path_factory):
    import shutil
    import time
    import bench
    base = str(tmp_path_factory.mktemp("refbin"))
    src = os.path.join(base, "src")
    os.makedirs(src)
    bench.synth_files(src, REFBIN_K, REFBIN_E, REFBIN_NC, 1_000_000, REFBIN_PAIRS, 0)
    procs = {}
    if os.path.exists(REF_BIN):
        for t in (1, 10):
            d = refbin_copy_inputs(src, os.path.join(base, f"ref_t{t}"))
            env = dict(os.environ, LD_PRELOAD=REF_SHIM) if t > 1 else dict(os.environ)
            argv = [REF_BIN, "s.1.fq", "s.2.fq", "ref.fa", "i.txt", "0.1", "0.08", str(t), str(REFBIN_K), "3000000", str(REFBIN_E), "1", "1"]
            procs[t] = (d, subprocess.Popen(argv, cwd=d, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), time.time())
    yield {"base": base, "src": src, "procs": procs}
    for d, p, _ in procs.values():
        if p.poll() is None:
            p.kill()
    shutil.rmtree(base, ignore_errors=True)


@pytest.fixture(scope="session", autouse=True)
def _start_reference_runs_early(request):
    """a GPU session that will reach tests/test_gpu_refbinary.py starts the reference's runs with its first test"""
    wanted = any("test_gpu_refbinary" in item.nodeid and "reference_binary" in item.nodeid for item in request.session.items)
    if wanted and os.path.exists(REF_BIN):
        request.getfixturevalue("refbin_big")
