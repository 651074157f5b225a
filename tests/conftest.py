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
