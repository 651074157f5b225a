"""`extract_ref --batch MANIFEST` (round 6): one process, one context, the reference of consecutive samples resident on the GPU.
The contract is per sample the one of scripts/pipeline.sh:35 -- every sample's interval file (and the index / genome.len.txt the
first one builds) is byte-equal to what a call of its own writes and to the reference binary's golden."""
import os
import shutil
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# cases that share one reference (ref_seed 11): the first builds the index (sample = 1: quirk Q3 has nothing to show), the others find
# it -- so the sampled one is the golden made with the index in place
SHARED = ["k24_base", "k24_sample_half_cached", "k24_fq2_longer", "k24_t4"]


def _stage(case_inputs, tmp_path, names, one_ref=True):
    """the cases' files side by side; with one_ref every sample names the same ref.fa"""
    samples, fa_shared = [], None
    for i, name in enumerate(names):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        d = tmp_path / f"s{i}_{name}"
        d.mkdir()
        if one_ref:
            if fa_shared is None:
                fa_shared = str(tmp_path / "ref.fa")
                shutil.copy(fa, fa_shared)
            else:
                assert cases.sha256_file(fa) == cases.sha256_file(fa_shared)
            fa2 = fa_shared
        else:
            fa2 = str(d / "ref.fa")
            shutil.copy(fa, fa2)
        interval = str(d / "interval.txt")
        samples.append((name, case, cases.extract_ref_argv(case, f1, f2, fa2, interval), interval, fa2, meta))
    return samples


def _check_golden(samples):
    for name, case, argv, interval, fa2, meta in samples:
        gold = os.path.join(cases.GOLDEN_DIR, name)
        assert open(interval).read() == open(os.path.join(gold, "interval.txt")).read(), name
        assert open(fa2 + ".genome.len.txt").read() == open(os.path.join(gold, "genome.len.txt")).read(), name


@pytest.mark.parametrize("ref_form,slot_list", [("index", None), ("packed", None), ("packed", "2")])
def test_batch_samples_equal_single_runs_and_goldens(case_inputs, tmp_path, monkeypatch, ref_form, slot_list):
    """four samples against one reference in one session: each file equals the golden of the reference binary and the file of a
    run of its own; the reference is loaded once.  slot_list "2": the packed reference with its slot list from the first scan on
    (what a catalogue-sized reference gets by itself from its second sample on) -- kept across the samples, same files."""
    from localhgt_amd import extract_ref
    if slot_list:
        monkeypatch.setenv("LHGT_SLOT_LIST", slot_list)
    samples = _stage(case_inputs, tmp_path, SHARED)
    if ref_form == "packed":     # the goldens of the cached cases were made on an index in place: its coder header is what packed runs read
        a0 = extract_ref.parse_argv(samples[0][2])
        extract_ref.run(a0, log=lambda *a: None, ref_form="index", emulate_threads=False)
        os.remove(samples[0][3])
    lines = []
    reps = extract_ref.run_batch([extract_ref.parse_argv(s[2]) for s in samples], log=lines.append, ref_form=ref_form)
    assert all(r is not None for r in reps)
    assert [r["ref_reused"] for r in reps] == [False, True, True, True]
    assert [r["emulated_threads"] for r in reps] == [1, 1, 1, 4]
    if slot_list:
        assert all(r["slot_list_bytes"] > 0 and r["scan_form"] in ("slot-first", "slot-single") for r in reps), [(r["slot_list_bytes"], r["scan_form"]) for r in reps]
    _check_golden(samples)
    index = f"{samples[0][4]}.k24.h3.index.dat"
    assert cases.sha256_file(index) == samples[0][5]["sha256"]["index.dat"]
    # ... and equal to single runs (a context of its own per sample), which find the same index in place
    for name, case, argv, interval, fa2, meta in samples[1:]:
        batch_bytes = open(interval, "rb").read()
        os.remove(interval)
        extract_ref.run(extract_ref.parse_argv(argv), log=lambda *a: None, ref_form=ref_form)
        assert open(interval, "rb").read() == batch_bytes, name


def test_batch_through_the_executable_with_changing_references(case_inputs, tmp_path):
    """bin/extract_ref --batch MANIFEST as a user starts it (no environment variable of ours): samples of three different
    references and two different (k, e) in one manifest -- the context is replaced where k or e change, the reference where the
    file changes -- plus a sample whose FASTQ does not exist: reported, exit status 1, every other sample's files as the goldens"""
    names = ["k24_seed7", "k24_base", "k24_t4", "k20_e2", "k24_seed7"]
    samples = _stage(case_inputs, tmp_path, names, one_ref=False)
    # samples 1 and 2 share a reference file (sample = 1 in both: a fresh and a cached index give the same files)
    samples[2] = samples[2][:2] + (samples[2][2][:2] + [samples[1][4]] + samples[2][2][3:], samples[2][3], samples[1][4], samples[2][5])
    manifest = tmp_path / "batch.txt"
    bad = list(samples[0][2])
    bad[0] = str(tmp_path / "missing.1.fq")
    bad[3] = str(tmp_path / "missing.interval.txt")
    with open(manifest, "w") as f:
        f.write("# one extract_ref call per line\n\n")
        for i, s in enumerate(samples):
            f.write(" ".join(s[2]) + "\n")
            if i == 1:
                f.write(" ".join(bad) + "   # a sample that cannot be read\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("LHGT_")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "extract_ref"), "--batch", str(manifest)], env=env, capture_output=True, text=True)
    assert res.returncode == 1, res.stdout[-2000:] + res.stderr[-2000:]
    assert "error: sample 3" in res.stdout and "missing.1.fq" in res.stdout
    assert res.stdout.count("reference: resident from the previous sample") == 0    # the failed sample dropped the context: sample 4 reloads
    assert not os.path.exists(bad[3])
    _check_golden(samples)


def test_batch_keeps_the_reference_across_a_manifest(case_inputs, tmp_path):
    """the executable on a manifest of three samples of one reference: loaded once, and a changed coder (another seed with no index in
    place, packed form) is a new coder for the slot list too (lhgt_coder_set drops the list)"""
    from localhgt_amd import extract_ref
    samples = _stage(case_inputs, tmp_path, ["k24_base", "k24_fq2_longer", "k24_base"])
    argvs = [list(s[2]) for s in samples]
    argvs[1][10] = "5"                       # another seed: with no index file the packed form draws another coder
    argvs[1][3] = str(tmp_path / "seed5.interval.txt")
    os.environ["LHGT_SLOT_LIST"] = "2"
    try:
        reps = extract_ref.run_batch([extract_ref.parse_argv(a) for a in argvs], log=lambda *a: None, ref_form="packed")
        assert [r["ref_reused"] for r in reps] == [False, True, True]
        assert all(r["slot_list_bytes"] > 0 for r in reps)
        got = [open(a[3], "rb").read() for a in argvs]
        os.environ["LHGT_SLOT_LIST"] = "0"
        for a, want in zip(argvs, got):
            os.remove(a[3])
            extract_ref.run(extract_ref.parse_argv(a), log=lambda *x: None, ref_form="packed")
            assert open(a[3], "rb").read() == want
    finally:
        del os.environ["LHGT_SLOT_LIST"]
    assert got[0] == got[2] == open(os.path.join(cases.GOLDEN_DIR, "k24_base", "interval.txt"), "rb").read()
    assert got[1] != got[0]
