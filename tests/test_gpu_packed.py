"""A sample packed by `localhgt_pack` (round 6, localhgt_amd/pack.py + csrc/k_packed.hip) through `extract_ref`: the files of the run on
the FASTQ text, i.e. the reference's goldens -- every read kept, sampling by global ordinal (index built in the run and cached: quirk
Q3), --sample > 1 (the base count of cal_sam_ratio from the header), a second file laid out wider than the first (quirk Q4), N runs
and lower case, and the reference's -t N through the header's thread chunks.  Files the loader pairs by anything but their line
numbers are refused by the packer and stay FASTQ."""
import os
import shutil
import subprocess
import sys

import pytest

import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CLEAN = ["k24_base", "k24_sample_half_fresh", "k24_sample_half_cached", "k24_sample_bases", "k24_fq2_longer", "k24_nrun_lower", "k21_e3", "k20_e2",
         "k24_t4", "k24_t8_sample_half", "k24_t3_fq2_longer", "k24_t10_sample_bases", "k24_seed7"]


@pytest.fixture(scope="module")
def packed(case_inputs, tmp_path_factory):
    from localhgt_amd import pack
    cache = {}

    def get(name):
        if name not in cache:
            fa, f1, f2, meta = case_inputs(name)
            out = str(tmp_path_factory.mktemp("packed") / f"{name}.lhgp")
            hdr = pack.pack(f1, f2, out, max_threads=12, log=lambda *a: None)
            cache[name] = (out, hdr)
        return cache[name]

    return get


@pytest.mark.parametrize("ref_form", ["index", "packed"])
@pytest.mark.parametrize("name", CLEAN)
def test_packed_sample_gives_the_reference_files(case_inputs, packed, name, ref_form, tmp_path):
    from localhgt_amd import extract_ref
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    sample, hdr = packed(name)
    assert hdr["n_pairs"] == hdr["lines"] // 4 and os.path.getsize(sample) == hdr["data_offset"] + hdr["n_pairs"] * hdr["stride"]
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(tmp_path / "interval.txt")
    argv = cases.extract_ref_argv(case, sample, "-", fa2, interval)
    forms = (["index"] if case.preexisting_index else []) + [ref_form]      # a golden made with the index already in place (quirk Q3)
    for form in forms:
        rep = extract_ref.run(extract_ref.parse_argv(argv), log=lambda *a: None, emulate_threads=case.threads > 1, ref_form=form)
    gold = os.path.join(cases.GOLDEN_DIR, name)
    assert rep["n_peaks"] == meta["raw_peaks"] and rep["emulated_threads"] == case.threads
    assert open(interval).read() == open(os.path.join(gold, "interval.txt")).read()
    # ... and the same pairs kept as from the text (on a reference copy of its own: whether the index is built in the run decides the
    # sampling stream, quirk Q3)
    d = tmp_path / "fq"
    d.mkdir()
    fa3 = str(d / "ref.fa")
    shutil.copy(fa, fa3)
    for form in forms:
        rep_fq = extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa3, str(d / "fq.txt"))), log=lambda *a: None,
                                 emulate_threads=case.threads > 1, ref_form=form)
    assert rep["pairs_kept"] == rep_fq["pairs_kept"] and rep["pairs_seen"] == rep_fq["pairs_seen"], (rep["pairs_kept"], rep_fq["pairs_kept"])
    assert open(str(d / "fq.txt")).read() == open(interval).read()


def test_quirk_q4_is_in_the_header(packed):
    _, hdr = packed("k24_fq2_longer")
    assert 0 < hdr["q4_first_pair"] < hdr["n_pairs"]          # fq2 is wider than fq1: its last records lie behind size(fq1) (E:1419-1445)
    _, hdr = packed("k24_base")
    assert hdr["q4_first_pair"] == hdr["n_pairs"]


@pytest.mark.parametrize("name", ["k24_fq2_surplus", "k24_fq2_stray2", "k24_fq2_short", "k24_fq2_short_nonl", "k24_long_line_fq1"])
def test_files_the_loader_pairs_otherwise_are_refused(case_inputs, name, tmp_path):
    """unequal files (surplus, foreign or missing records: paired by read ID or against a stale line, E:356-402) and a line beyond the
    reference's buffers are no packed samples: the packer says so and writes nothing usable"""
    from localhgt_amd import _lib, pack
    fa, f1, f2, meta = case_inputs(name)
    out = str(tmp_path / "s.lhgp")
    with pytest.raises((SystemExit, _lib.LocalHGTError)):
        pack.pack(f1, f2, out, max_threads=4, log=lambda *a: None)
    assert not pack.is_packed(out)


def test_a_thread_count_the_header_does_not_hold_falls_back_with_a_warning(case_inputs, packed, tmp_path):
    from localhgt_amd import extract_ref
    case = cases.CASES["k24_base"]
    fa, f1, f2, meta = case_inputs("k24_base")
    sample, hdr = packed("k24_base")
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    interval = str(tmp_path / "i.txt")
    argv = cases.extract_ref_argv(case, sample, "-", fa2, interval)
    argv[6] = "40"                                        # -t 40: packed with --max-threads 12
    lines = []
    rep = extract_ref.run(extract_ref.parse_argv(argv), log=lines.append)
    assert rep["emulated_threads"] == 1 and any("warning" in x and "packed with the thread chunks" in x for x in lines), lines
    assert open(interval).read() == open(os.path.join(cases.GOLDEN_DIR, "k24_base", "interval.txt")).read()


def test_two_ranks_read_a_packed_sample(case_inputs, packed, tmp_path):
    """world 2 on one GPU (gloo): each rank reads half of the packed pairs; sampling by the global read ordinal; the -t 10 thread chunks
    cut across the ranks"""
    from test_gpu_world2 import _launch
    for name in ("k24_sample_half_cached", "k24_t10_sample_bases"):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        sample, hdr = packed(name)
        d = tmp_path / name
        d.mkdir()
        fa2 = str(d / "ref.fa")
        shutil.copy(fa, fa2)
        interval = str(d / "S.interval.txt")
        if case.preexisting_index:
            from localhgt_amd import extract_ref
            extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, f1, f2, fa2, str(d / "first.txt"))), log=lambda *a: None)
        _launch(2, cases.extract_ref_argv(case, sample, "-", fa2, interval))
        assert open(interval).read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read(), name


def test_the_packer_as_an_executable(case_inputs, tmp_path):
    fa, f1, f2, meta = case_inputs("k24_seed7")
    out = str(tmp_path / "s.lhgp")
    env = {k: v for k, v in os.environ.items() if not k.startswith("LHGT_")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "localhgt_pack"), f1, f2, out, "--max-threads", "10"], env=env, capture_output=True, text=True)
    assert res.returncode == 0 and "pairs" in res.stdout, res.stdout + res.stderr[-2000:]
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    case = cases.CASES["k24_seed7"]
    argv = cases.extract_ref_argv(case, out, "-", fa2, str(tmp_path / "i.txt"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "extract_ref")] + argv, env=env, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert open(tmp_path / "i.txt").read() == open(os.path.join(cases.GOLDEN_DIR, "k24_seed7", "interval.txt")).read()


def test_a_damaged_record_fails_the_load(case_inputs, packed, tmp_path):
    """a packed sample is input like any other: a record whose read lengths the packer cannot have written -- beyond the reference's 500
    bases (E:1004), or beyond what the stride holds, which the gather would copy from the NEXT records -- is not loaded; the run stops
    with a format error (round 6, late)"""
    import struct
    from localhgt_amd import _lib, extract_ref
    case = cases.CASES["k24_base"]
    fa, f1, f2, meta = case_inputs("k24_base")
    sample, hdr = packed("k24_base")
    for la in (60000, 501):
        bad = str(tmp_path / f"bad{la}.lhgp")
        shutil.copy(sample, bad)
        with open(bad, "r+b") as f:
            f.seek(hdr["data_offset"] + 17 * hdr["stride"])
            f.write(struct.pack("<H", la))
        fa2 = str(tmp_path / f"ref{la}.fa")
        shutil.copy(fa, fa2)
        with pytest.raises(_lib.LocalHGTError) as ei:
            extract_ref.run(extract_ref.parse_argv(cases.extract_ref_argv(case, bad, "-", fa2, str(tmp_path / "i.txt"))), log=lambda *a: None)
        assert ei.value.code == 4 and "hold read lengths no packed sample holds" in str(ei.value), str(ei.value)


@pytest.mark.parametrize("name", ["k24_base", "k24_fq2_longer", "k24_nrun_lower", "k21_e3"])
def test_host_packer_writes_the_bytes_the_gpu_packer_writes(case_inputs, packed, name, tmp_path):
    """`localhgt_pack --host` (no GPU touched: lhgt_fastq_pack_host) against the packer that goes through the resident store: the same
    records, byte for byte, and the same header but for the sources' timestamps -- so everything the tests above say of a packed sample
    holds for one packed on a machine without a GPU"""
    from localhgt_amd import pack
    fa, f1, f2, meta = case_inputs(name)
    gpu_file, gpu_hdr = packed(name)
    out = str(tmp_path / "host.lhgp")
    hdr = pack.pack(f1, f2, out, max_threads=12, host=True, log=lambda *a: None)
    assert {k: v for k, v in hdr.items() if k != "sources"} == {k: v for k, v in gpu_hdr.items() if k != "sources"}
    with open(out, "rb") as a, open(gpu_file, "rb") as b:
        a.seek(hdr["data_offset"])
        b.seek(hdr["data_offset"])
        assert a.read() == b.read()
