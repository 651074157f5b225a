"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the
host rows (RNG/coder, sampling ratio) match libc / the oracle, BED and CLI match the
reference's goldens."""
import ctypes
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from localhgt_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "localhgt_hip.h")).read()
    declared = set(re.findall(r"\b(lhgt_[a-z0-9_]+)\s*\(", header))
    declared.discard("lhgt_ctx")
    handle = lib.load(require_gpu=False)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/localhgt_hip.h but not exported"
    assert declared == set(lib.SIGNATURES), "python binding and header disagree"
    assert handle.lhgt_abi_version() == 1


def test_no_cpu_fallback(lib):
    """device work on a host-only context fails loudly; creating a GPU context without a GPU fails too"""
    from localhgt_amd.engine import Engine
    eng = Engine(24, 3, device=-1)
    with pytest.raises(lib.LocalHGTError) as ei:
        eng.count_kmers()
    assert ei.value.code == 8
    with pytest.raises(lib.LocalHGTError):
        eng.hash_sequence(b"ACGT" * 20)
    import torch
    if not torch.cuda.is_available():
        with pytest.raises((lib.LocalHGTError, RuntimeError)):
            Engine(24, 3, device=0)


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "localhgt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_api" not in src and "lhgt_oracle" not in src and "oracle/" not in src, f


@pytest.mark.parametrize("k,e,seed", [(24, 3, 1), (32, 3, 1), (32, 3, 7), (22, 5, 3), (20, 2, 0), (8, 9, 12345), (31, 4, 2**31 + 5)])
def test_coder_matches_libc_rand(oracle, k, e, seed):
    from localhgt_amd.engine import Engine
    with Engine(k, e, device=-1) as eng:
        eng.rng_seed(seed)
        eng.coder_generate()
        oracle.srand(seed)
        assert (eng.coder_get() == oracle.random_coder(k, e)).all()


def test_sampling_array_matches_libc_rand_after_coder(oracle):
    """quirk Q3: the stream position of get_random depends on whether random_coder ran first"""
    from localhgt_amd.engine import Engine
    for with_coder in (True, False):
        with Engine(24, 3, device=-1) as eng:
            eng.rng_seed(5)
            oracle.srand(5)
            if with_coder:
                eng.coder_generate()
                oracle.random_coder(24, 3)
            eng.sampling_init(50.0)
            want = oracle.sampling_array(200000)
            got = eng.sampling_get(200000)
            assert got.dtype == np.float32 and (got == want).all()
            assert got.max() < 100.0


def test_sampling_array_begun_early_is_the_same_array():
    """lhgt_sampling_begin fills on a host thread of its own (next to the line count and the reference load, extract_ref.run);
    joined by sampling_init it is the synchronous array -- whole, cut short by sampling_reserve, or dropped when ratio >= 100"""
    from localhgt_amd.engine import Engine
    with Engine(24, 3, device=-1) as ref:
        ref.rng_seed(9)
        ref.coder_generate()
        ref.sampling_init(37.5)
        want = ref.sampling_get(50_000_000)
    with Engine(24, 3, device=-1) as eng:
        eng.rng_seed(9)
        eng.coder_generate()
        eng.sampling_begin()
        eng.sampling_init(37.5)
        assert (eng.sampling_get(50_000_000) == want).all()
        # cut short: at least the reserved entries are the stream's, the rest are the stream's or still 0
        eng.rng_seed(9)
        eng.coder_generate()
        eng.sampling_begin()
        eng.sampling_reserve(1_000_000)
        eng.sampling_init(37.5)
        got = eng.sampling_get(50_000_000)
        assert (got[:1_000_000] == want[:1_000_000]).all()
        assert ((got == want) | (got == 0)).all()
        # ratio >= 100: nobody looks at the draws, the array is dropped
        eng.rng_seed(9)
        eng.sampling_begin()
        eng.sampling_init(100.0)
        with pytest.raises(Exception):
            eng.sampling_get(10)
        # a new seed while a fill is in flight joins it first
        eng.sampling_begin()
        eng.rng_seed(9)
        eng.coder_generate()
        eng.sampling_init(37.5)
        assert (eng.sampling_get(200_000) == want[:200_000]).all()


def test_sam_ratio_matches_oracle(oracle, case_inputs):
    from localhgt_amd.engine import Engine
    fa, f1, f2, _ = case_inputs("k24_seed7")
    with Engine(24, 3, device=-1) as eng:
        for sample in (1.0, 0.5, 0.001, 700000.0, 2e9):
            assert eng.sam_ratio(f1, sample) == oracle.sam_ratio(f1, sample)


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c.bed_defined])
def test_bed_matches_reference_script(name, tmp_path):
    from localhgt_amd import get_bed_file
    gold = os.path.join(cases.GOLDEN_DIR, name)
    ref = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(gold, "genome.len.txt"), ref + ".genome.len.txt")
    interval = str(tmp_path / "interval.txt")
    shutil.copy(os.path.join(gold, "interval.txt"), interval)
    n = get_bed_file.write_bed(ref, interval)
    assert open(interval + ".bed").read() == open(os.path.join(gold, "interval.txt.bed")).read()
    meta = json.load(open(os.path.join(gold, "meta.json")))
    assert meta["bed_stdout"] == f"extracted ref length is: {n}\n"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "get_bed_file.py"), ref, interval],
                         capture_output=True, text=True, check=True)
    assert out.stdout == meta["bed_stdout"]


def test_bed_reports_inconsistent_ids(tmp_path):
    from localhgt_amd import get_bed_file
    gold = os.path.join(cases.GOLDEN_DIR, "k24_short_mid")
    ref = str(tmp_path / "ref.fa")
    shutil.copy(os.path.join(gold, "genome.len.txt"), ref + ".genome.len.txt")
    interval = str(tmp_path / "interval.txt")
    shutil.copy(os.path.join(gold, "interval.txt"), interval)
    with pytest.raises(get_bed_file.InconsistentReferenceIds):
        get_bed_file.write_bed(ref, interval)


def test_cli_command_line_matches_reference_driver():
    from localhgt_amd import cli
    gold = json.load(open(os.path.join(cases.GOLDEN_DIR, "cmdline.json")))
    parser = cli.build_parser()
    for name in ("defaults", "sample_half", "all_flags"):
        o = parser.parse_args(gold[name]["args"])
        ours = cli.run_order(o, o.fq1, o.fq2, "/X/pipeline.sh")
        theirs = re.sub(r"bash \S+/pipeline\.sh", "bash /X/pipeline.sh", gold[name]["os_system"][0])
        assert ours == theirs


def test_cli_help_lists_the_reference_flags():
    from localhgt_amd import cli
    gold = json.load(open(os.path.join(cases.GOLDEN_DIR, "cmdline.json")))["help"]["stdout"]
    ours = cli.build_parser().format_help()
    gold, ours = gold[gold.index("required arguments:"):], ours[ours.index("required arguments:"):]
    ref_flags = set(re.findall(r"^\s+(--?[a-z_0-9]+)", gold, flags=re.M))
    our_flags = set(re.findall(r"^\s+(--?[a-z_0-9]+)", ours, flags=re.M))
    assert ref_flags <= our_flags
    for flag in ref_flags - {"-h"}:  # same default text per flag
        d_ref = re.search(re.escape(flag) + r" \x08.*?\(default:\s*([^)]*)\)", gold, flags=re.S)
        d_our = re.search(re.escape(flag) + r" \x08.*?\(default:\s*([^)]*)\)", ours, flags=re.S)
        assert d_ref and d_our and d_ref.group(1).split() == d_our.group(1).split(), flag


def test_cli_dry_run_prints_pipeline_command(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "localhgt"), "bkp", "-r", "ref.fa", "--fq1", "a.1.fq",
                          "--fq2", "a.2.fq", "--dry-run", "--pipeline", "/opt/LocalHGT/scripts/pipeline.sh"],
                         capture_output=True, text=True, check=True)
    assert "bash /opt/LocalHGT/scripts/pipeline.sh ref.fa a.1.fq a.2.fq sample ./ 0.1 0.08 10 32 300000000 3 1 2000000000 1 1 20" in out.stdout


def test_extract_ref_argv_parsing():
    from localhgt_amd import extract_ref
    a = extract_ref.parse_argv(["a.fq", "b.fq", "r.fa", "out.txt", "0.1", "0.08", "10", "32", "300000000", "3", "1", "2000000000"])
    assert (a.k, a.e, a.threads, a.max_peak, a.seed, a.sample) == (32, 3, 10, 300000000, 1, 2e9)
    assert a.hit_ratio == float(np.float32(0.1)) and a.match_ratio == float(np.float32(0.08))
    assert extract_ref.index_name("r.fa", 32, 3) == "r.fa.k32.h3.index.dat"
    a = extract_ref.parse_argv(["a", "b", "r", "o", "0.2", "0.05", "4", "24.0", "1e3", "4", "9", "0.5"])   # stod accepts these
    assert (a.k, a.max_peak, a.sample) == (24, 1000, 0.5)


def _digest(lib, f1, f2, ratio=100.0, rnd=None, rank=0, world=1, block=64, threads=1, chunk=1 << 40):
    import ctypes as C
    h = lib.load(require_gpu=False)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    rp = rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None
    rc = h.lhgt_fastq_parse_digest(f1.encode(), f2.encode(), float(ratio), rp, rank, world, block, threads, chunk,
                                   C.byref(seen), C.byref(kept), C.byref(dig))
    return rc, seen.value, kept.value, dig.value


def test_batch_manifest_holds_one_extract_ref_call_per_line(tmp_path):
    """`extract_ref --batch MANIFEST` (round 6): the 12 arguments of scripts/pipeline.sh:35 per line, parsed like argv (stod then
    truncation, E:1352-1371); comments, blank lines and shell quoting; a line with another number of arguments is refused with its number"""
    from localhgt_amd import extract_ref
    m = tmp_path / "batch.txt"
    m.write_text("# two samples of one reference\n\n"
                 "a.1.fq a.2.fq ref.fa out/a.interval.txt 0.1 0.08 10 32 300000000 3 1 2000000000\n"
                 "'b 1.fq' b.2.fq ref.fa out/b.interval.txt 0.1 0.08 10.9 32 3e8 3 7 0.5   # a path with a blank, threads 10.9 -> 10\n")
    got = extract_ref.read_manifest(str(m))
    assert [a.fq1 for a in got] == ["a.1.fq", "b 1.fq"] and [a.threads for a in got] == [10, 10]
    assert got[0] == extract_ref.parse_argv("a.1.fq a.2.fq ref.fa out/a.interval.txt 0.1 0.08 10 32 300000000 3 1 2000000000".split())
    assert (got[1].max_peak, got[1].seed, got[1].sample) == (300000000, 7, 0.5)
    m.write_text("a.1.fq a.2.fq ref.fa out 0.1 0.08 10 32 300000000 3 1\n")
    with pytest.raises(SystemExit) as ei:
        extract_ref.read_manifest(str(m))
    assert "batch.txt:1" in str(ei.value) and "11 arguments" in str(ei.value)
    with pytest.raises(SystemExit):
        extract_ref.main(["--batch"])


def test_parallel_fastq_parser_is_split_invariant(lib, oracle, case_inputs, tmp_path):
    """any chunk size / thread count gives the same kept pairs, in the same order, with the same mate-2 flags"""
    fa, f1, f2, _ = case_inputs("k24_fq2_longer")          # fq2 longer than fq1: exercises the Q4 flag
    base = _digest(lib, f1, f2)
    assert base[0] == 0 and base[1] == base[2] > 1000
    for threads, chunk in ((4, 1000), (8, 333), (3, 65536), (16, 150)):
        assert _digest(lib, f1, f2, threads=threads, chunk=chunk) == base, (threads, chunk)
    # sampling + sharding: the two shards partition the sampled set
    oracle.srand(3)
    rnd = np.resize(oracle.sampling_array(1_000_000), 50_000_000)   # any 50 M floats do for a split-invariance check
    full = _digest(lib, f1, f2, ratio=40.0, rnd=rnd)
    assert 0 < full[2] < base[2]
    for threads, chunk in ((4, 777), (7, 4096)):
        assert _digest(lib, f1, f2, ratio=40.0, rnd=rnd, threads=threads, chunk=chunk) == full
        a = _digest(lib, f1, f2, ratio=40.0, rnd=rnd, rank=0, world=2, block=16, threads=threads, chunk=chunk)
        b = _digest(lib, f1, f2, ratio=40.0, rnd=rnd, rank=1, world=2, block=16, threads=threads, chunk=chunk)
        assert a[2] + b[2] == full[2] and a[2] > 0 and b[2] > 0
    # no trailing newline, CRLF, truncated last record
    raw1, raw2 = open(f1, "rb").read(), open(f2, "rb").read()
    variants = {
        "nonl": (raw1[:-1], raw2[:-1]),
        "crlf": (raw1[:20000].replace(b"\n", b"\r\n"), raw2[:20800].replace(b"\n", b"\r\n")),
    }
    n_lines = raw1[:30000].count(b"\n")
    cut1 = b"\n".join(raw1.split(b"\n")[: n_lines - n_lines % 4 + 2]) + b"\n"      # ends after a sequence line
    cut2 = b"\n".join(raw2.split(b"\n")[: n_lines - n_lines % 4 + 2]) + b"\n"
    variants["partial"] = (cut1, cut2)
    for name, (a1, a2) in variants.items():
        p1, p2 = str(tmp_path / f"{name}.1.fq"), str(tmp_path / f"{name}.2.fq")
        open(p1, "wb").write(a1)
        open(p2, "wb").write(a2)
        if name == "crlf" and a1.count(b"\n") != a2.count(b"\n"):
            n = min(a1.count(b"\n"), a2.count(b"\n"))
            open(p1, "wb").write(b"\n".join(a1.split(b"\n")[:n]) + b"\n")
            open(p2, "wb").write(b"\n".join(a2.split(b"\n")[:n]) + b"\n")
        one = _digest(lib, p1, p2)
        assert one[0] == 0, name
        for threads, chunk in ((4, 500), (5, 97)):
            assert _digest(lib, p1, p2, threads=threads, chunk=chunk) == one, (name, threads, chunk)
    # a second file with fewer records is read like the reference reads it (E:356-367): every pair of fq1 is kept, the ones
    # behind fq2's end with an empty mate 2 (golden k24_fq2_short), whatever the split ...
    p2s = str(tmp_path / "short.2.fq")
    open(p2s, "wb").write(b"\n".join(raw2.split(b"\n")[:400]) + b"\n")
    short = _digest(lib, f1, p2s)
    assert short[:3] == (0, base[1], base[1]) and short[3] != base[3]
    assert _digest(lib, f1, p2s, threads=4, chunk=1000) == short
    # ... one with more records is read like the reference reads it: phase C stops with fq1, phase A counts the surplus
    # records of fq2 that start inside size(fq1) (none here: the files share their first 400 lines)
    rc, seen, kept, _ = _digest(lib, p2s, f2, threads=4, chunk=1000)
    assert (rc, seen, kept) == (0, 100, 100)


def _plan(lib, path, chunk, parts):
    """the pieces lhgt_fastq_plan_part makes for `parts` ranks, concatenated the way localhgt_amd.dist does"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    starts, counts = [], []
    for part in range(parts):
        n, tot = C.c_long(0), C.c_long(0)
        assert h.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, None, None, 0, C.byref(n), C.byref(tot), None) == 0
        st, cn = (C.c_uint64 * max(n.value, 1))(), (C.c_long * max(n.value, 1))()
        assert h.lhgt_fastq_plan_part(path.encode(), chunk, part, parts, st, cn, n.value, C.byref(n), C.byref(tot), None) == 0
        starts += list(st)[:n.value]
        counts += list(cn)[:n.value]
    assert len(starts) == tot.value
    return np.array(starts, dtype=np.uint64), np.array(counts, dtype=np.int64)


def _digest_planned(lib, f1, f2, plan1, plan2, part, parts, state=None, ratio=100.0, rnd=None, threads=3, emulate=1, chunk=1 << 40):
    import ctypes as C
    h = lib.load(require_gpu=False)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(state or 0)
    cnt = (C.c_long * 3)()
    rp = rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None
    u64, lp = C.POINTER(C.c_uint64), C.POINTER(C.c_long)
    rc = h.lhgt_fastq_parse_digest_planned(f1.encode(), f2.encode(), float(ratio), rp, 0, 1, 1, threads, chunk, emulate,
                                           plan1[0].ctypes.data_as(u64), plan1[1].ctypes.data_as(lp), len(plan1[0]),
                                           plan2[0].ctypes.data_as(u64), plan2[1].ctypes.data_as(lp), len(plan2[0]),
                                           part, parts, 0 if state is None else 1, C.byref(seen), C.byref(kept), C.byref(dig), cnt)
    return rc, seen.value, kept.value, dig.value, list(cnt)


@pytest.mark.parametrize("name", ["k24_fq2_longer", "k24_fq2_surplus", "k24_fq2_stray2", "k24_fq2_short_nonl", "k24_t3_fq2_longer"])
def test_rank_ranges_of_the_fastqs_add_up_to_the_single_rank_parse(lib, oracle, case_inputs, name):
    """multi-GPU ingest: every rank counts the lines of its share of both files, the pieces are exchanged, every rank
    parses only its run of fq1's chunks -- the parts, in rank order, give exactly the pairs (order, flags, bases) of the
    single-rank parse, with sampling by the global read ordinal, quirk Q4, surplus / foreign fq2 records and -t N emulation"""
    case = cases.CASES[name]
    fa, f1, f2, _ = case_inputs(name)
    rnd, ratio = None, 100.0
    if float(case.sample) < 1:
        oracle.srand(case.seed)
        rnd = np.resize(oracle.sampling_array(1_000_000), 50_000_000)
        ratio = 100.0 * float(case.sample)
    import ctypes as C
    h = lib.load(require_gpu=False)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    cnt = (C.c_long * 3)()
    rp = rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None
    assert h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), ratio, rp, 0, 1, 1, 2, 1 << 40, case.threads, C.byref(seen), C.byref(kept),
                                             C.byref(dig), cnt) == 0
    whole = (seen.value, kept.value, dig.value, list(cnt))
    assert kept.value > 500
    for chunk, parts in ((30000, 2), (7777, 3), (100000, 8), (1 << 22, 2)):
        plan1, plan2 = _plan(lib, f1, chunk, parts), _plan(lib, f2, chunk, parts)
        assert int(plan1[1].sum()) == sum(1 for _ in open(f1, "rb"))
        state, kept_sum, cnt_sum, kept_parts = None, 0, [0, 0, 0], []
        for part in range(parts):
            rc, sn, kp, dg, c3 = _digest_planned(lib, f1, f2, plan1, plan2, part, parts, state=state if part else None, ratio=ratio, rnd=rnd,
                                                 emulate=case.threads)
            assert rc == 0 and sn == whole[0]
            state, kept_sum, cnt_sum = dg, kept_sum + kp, [a + b for a, b in zip(cnt_sum, c3)]
            kept_parts.append(kp)
        assert (kept_sum, state, cnt_sum) == whole[1:], (chunk, parts)
        if chunk < 100000 and parts <= 3:
            assert min(kept_parts) > 0.5 * kept_sum / parts            # contiguous, about equal shares
    # a plan that is not made of whole lines is refused before anything is parsed
    bad = (plan1[0].copy(), plan1[1])
    if len(bad[0]) > 1:
        bad[0][1] += 1
        assert _digest_planned(lib, f1, f2, bad, plan2, 0, 2)[0] == 1


@pytest.mark.parametrize("name", ["k24_sample_bases", "k24_fq2_surplus", "k24_seed7"])
def test_sampling_ratio_from_the_line_plan_equals_cal_sam_ratio(lib, oracle, case_inputs, name, monkeypatch):
    """--sample > 1 (the CLI default): the base count of cal_sam_ratio (E:1244-1270) falls out of the line count the loader makes
    anyway (per-chunk sums of line lengths by line index mod 4) -- the same double as the extra pass gives, whatever the chunking"""
    from localhgt_amd.engine import Engine
    fa, f1, f2, _ = case_inputs(name)
    eng = Engine(24, 3, device=-1)
    want = oracle.sam_ratio(f1, 700000.0)
    assert eng.sam_ratio(f1, 700000.0) == want
    for chunk in ("256", "1000", "77777", None):
        if chunk:
            monkeypatch.setenv("LHGT_INGEST_CHUNK_BYTES", chunk)
        else:
            monkeypatch.delenv("LHGT_INGEST_CHUNK_BYTES")
        p1, p2 = eng.fastq_plan(f1, True, other=f2)
        assert eng.sam_ratio_from_plan(p1, 700000.0) == want, chunk
        assert int(p1[1].sum()) == sum(1 for _ in open(f1, "rb")) and int(p2[1].sum()) == sum(1 for _ in open(f2, "rb"))
    assert eng.sam_ratio_from_plan(p1, 0.5) == 50.0
    eng.close()


def test_fq2_with_foreign_records_in_front_is_resynchronised_like_the_reference(lib, oracle, case_inputs, tmp_path):
    """E:368-402: the first read IDs differ, so phase C re-reads fq2 from byte 1 until a line carries fq1's first ID and pairs
    fq1's line g with fq2's line g + 8; phase A (E:1426-1448) reads each file on its own -- mate 2 of pair n is fq2's read n + 2 and is
    sampled as such, the two foreign reads are counted and never voted.  Counts against the oracle's whole run (golden-pinned)."""
    import ctypes as C
    name = "k24_fq2_stray2"
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    rc, rep = oracle.run(f1, f2, fa2, str(tmp_path / "i.txt"), case.hit_ratio, case.match_ratio, 1, case.k, case.max_peak, case.e, case.seed,
                         float(case.sample))
    assert rc == 0 and open(tmp_path / "i.txt").read() == open(os.path.join(cases.GOLDEN_DIR, name, "interval.txt")).read()
    oracle.srand(case.seed)
    cc = oracle.random_coder(case.k, case.e)          # index built in-run (quirk Q3)
    rnd = oracle.sampling_array(50_000_000)
    table = np.zeros(1 << case.k, dtype=np.uint8)
    c2 = oracle.count(f2, os.path.getsize(f1), case.k, case.e, cc, 50.0, rnd, table)
    h = lib.load(require_gpu=False)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    cnt = (C.c_long * 3)()
    assert h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), 50.0, rnd.ctypes.data_as(C.POINTER(C.c_float)), 0, 1, 1, 4, 50000, 1,
                                             C.byref(seen), C.byref(kept), C.byref(dig), cnt) == 0
    assert (cnt[0], cnt[1], cnt[2]) == (rep.pairs_counted, c2, rep.pairs_voted)
    assert kept.value > max(cnt[0], cnt[1])            # entries with only one mate counted exist: the ordinals are shifted by two
    # refused: no line with fq1's first ID anywhere in fq2; a shift inside a record
    lines2 = open(f2, "rb").read().split(b"\n")
    none2 = str(tmp_path / "none.2.fq")
    open(none2, "wb").write(b"\n".join(lines2[:8] + lines2[12:]) + b"\n" if False else b"\n".join(lines2[:8]) + b"\n")
    assert h.lhgt_fastq_parse_digest_threads(f1.encode(), none2.encode(), 100.0, None, 0, 1, 1, 2, 50000, 1, C.byref(seen), C.byref(kept),
                                             C.byref(dig), None) == 4
    odd2 = str(tmp_path / "odd.2.fq")
    open(odd2, "wb").write(b"\n".join(lines2[:3] + lines2[8:]))
    assert h.lhgt_fastq_parse_digest_threads(f1.encode(), odd2.encode(), 100.0, None, 0, 1, 1, 2, 50000, 1, C.byref(seen), C.byref(kept),
                                             C.byref(dig), None) == 4


def test_surplus_records_of_fq2_are_counted_not_voted(lib, oracle, case_inputs):
    """fq2 with 300 records more than fq1 and shorter headers, fq1 with a trailing blank line: entries = fq1's pairs + the
    surplus mate-2 reads whose sequence line starts at <= size(fq1) -- what the reference's phase A visits (E:1438-1445)"""
    fa, f1, f2, _ = case_inputs("k24_fq2_surplus")
    n1 = sum(1 for _ in open(f1)) // 4
    table = np.zeros(1 << 12, dtype=np.uint8)
    cc = oracle.random_coder(12, 3)
    counted2 = oracle.count(f2, os.path.getsize(f1), 12, 3, cc, 100.0, None, table)
    assert counted2 > n1
    base = _digest(lib, f1, f2)
    assert base[:3] == (0, n1, counted2)
    for threads, chunk in ((4, 1000), (7, 333)):
        assert _digest(lib, f1, f2, threads=threads, chunk=chunk) == base


def test_parser_counts_match_the_oracle_reader(lib, oracle, case_inputs):
    """pairs seen == sequence lines the reference's getline loop would visit (oracle count with every read kept)"""
    fa, f1, f2, _ = case_inputs("k24_seed7")
    rc, seen, kept, _ = _digest(lib, f1, f2, threads=4, chunk=10000)
    table = np.zeros(1 << 12, dtype=np.uint8)
    cc = oracle.random_coder(12, 3)
    assert rc == 0 and seen == kept == oracle.count(f1, 1 << 40, 12, 3, cc, 100.0, None, table)


# ------------------------------------------------------------------ -t N read partition
def _thread_chunks(lib, path, size, threads):
    import ctypes as C
    h = lib.load(require_gpu=False)
    arr = [(C.c_long * threads)() for _ in range(3)]
    rc = h.lhgt_fastq_thread_chunks(path.encode(), size, threads, *arr)
    return rc, [list(a) for a in arr]


@pytest.mark.parametrize("name", ["k24_t4", "k24_t3_fq2_longer", "k24_t8_sample_half", "k24_t10_sample_bases"])
def test_thread_partition_matches_the_reference_log(lib, oracle, case_inputs, name):
    """entry byte, last record and read count of every thread chunk against the `>>> Thread: final read` lines the reference
    printed when the golden was made (E:1105), and get_fq_start against the oracle's literal restatement"""
    import re
    case = cases.CASES[name]
    fa, f1, f2, meta = case_inputs(name)
    t = case.threads
    size1 = os.path.getsize(f1)
    log = [re.match(r">>> Thread: final read (\d+) : (\S+).* read num: (\d+)", ln).groups() for ln in meta["thread_log"]]
    assert len(log) == 2 * t
    sampling = float(case.sample) != 1.0
    for path, rows in ((f1, log[:t]), (f2, log[t:])):
        rc, (entry, first, count) = _thread_chunks(lib, path, size1, t)
        assert rc == 0
        data = open(path, "rb").read()
        lines = data.split(b"\n")
        for i in range(t):
            start = i * (size1 // t)
            assert int(rows[i][0]) == start                              # the log prints the chunk's start byte
            assert entry[i] == oracle.get_fq_start(data, start)
            assert first[i] % 4 == 0 and data[entry[i]:entry[i] + 1] == b"@"
            seqs = (count[i] + 2) // 4                                    # lines with local index % 4 == 1 among the consumed ones
            if not sampling:
                assert seqs == int(rows[i][2])
            # the last header the thread saw (read_first_line, E:1105) is the header of its last consumed record
            last_header = first[i] + 4 * ((count[i] - 1) // 4)
            assert lines[last_header].split(b" ")[0].decode() == rows[i][1]


def test_thread_entry_equals_the_literal_scan_on_random_text(lib, oracle):
    """the restated automaton of get_fq_start (host_fastx.cpp: thread_entry) against the oracle's literal double loop with its
    persistent newline counter (orc_get_fq_start, E:44-89): byte soups over the five characters the scan tells apart, at newline
    densities from FASTQ-like to hostile, every start offset of short texts and sampled ones of long texts"""
    h = lib.load(require_gpu=False)
    rng = np.random.default_rng(20261004)
    checked = 0
    for case in range(260):
        n = int(rng.integers(1, 300)) if case < 200 else int(rng.integers(1500, 6000))
        p_nl = (0.02, 0.1, 0.3, 0.6)[case % 4]
        rest = (1 - p_nl) / 4
        text = bytes(rng.choice(np.frombuffer(b"\n+@AI", dtype=np.uint8), size=n, p=[p_nl, rest, rest, rest, rest]).tolist())
        if case % 5 == 0:                                  # real records in the soup, so that answers are found, not only refusals
            rec = b"@r1/1\nACGT\n+\nIIII\n" * int(rng.integers(1, 40))
            cut = int(rng.integers(0, len(text) + 1))
            text = text[:cut] + rec + text[cut:]
        n = len(text)
        starts = range(0, n + 3) if n < 400 else [int(x) for x in rng.integers(0, n + 3, size=60)]
        for start in starts:
            assert h.lhgt_fastq_thread_entry(text, n, start) == oracle.get_fq_start(text, start), (case, start)
            checked += 1
    assert checked > 30000


def test_thread_partition_flags_and_refusals(lib, oracle, case_inputs, tmp_path):
    """pairs_counted / mate-2 counts / voted pairs of the partitioned parse equal the oracle's -t N run; one thread = the plain
    parse; files too small for the thread count are refused like every input on which the reference reads stale bytes"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    for name in ("k24_t4", "k24_t3_fq2_longer", "k24_t8_sample_half"):
        case = cases.CASES[name]
        fa, f1, f2, meta = case_inputs(name)
        work = tmp_path / name
        work.mkdir()
        fa2 = str(work / "ref.fa")
        shutil.copy(fa, fa2)
        rc, rep = oracle.run_threads(f1, f2, fa2, str(work / "i.txt"), case.hit_ratio, case.match_ratio, case.threads, case.k, case.max_peak,
                                     case.e, case.seed, float(case.sample))
        assert rc == 0
        ratio = 100.0 * float(case.sample) if case.sample <= 1 else None
        rnd = None
        if ratio is not None and ratio < 100:
            oracle.srand(case.seed)
            oracle.random_coder(case.k, case.e)          # index built in-run: the coder draws come first (quirk Q3)
            rnd = oracle.sampling_array(50_000_000)
        seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
        cnt = (C.c_long * 3)()
        rc = h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), ratio if ratio is not None else 100.0,
                                               rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None, 0, 1, 4096, 4, 20000,
                                               case.threads, C.byref(seen), C.byref(kept), C.byref(dig), cnt)
        assert rc == 0
        assert (cnt[0], cnt[1], cnt[2]) == (rep.pairs_counted, int(rep.t_count), rep.pairs_voted), name
    # emulate_threads = 1 is the plain parse
    fa, f1, f2, _ = case_inputs("k24_t4")
    a = _digest(lib, f1, f2, threads=3, chunk=5000)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    assert h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), 100.0, None, 0, 1, 4096, 3, 5000, 1, C.byref(seen), C.byref(kept),
                                             C.byref(dig), None) == 0
    assert (0, seen.value, kept.value, dig.value) == a
    # 40 threads on a file of 6 records: chunks near EOF / overlapping
    tiny1, tiny2 = str(tmp_path / "t.1.fq"), str(tmp_path / "t.2.fq")
    open(tiny1, "wb").write(b"\n".join(open(f1, "rb").read().split(b"\n")[:24]) + b"\n")
    open(tiny2, "wb").write(b"\n".join(open(f2, "rb").read().split(b"\n")[:24]) + b"\n")
    assert h.lhgt_fastq_parse_digest_threads(tiny1.encode(), tiny2.encode(), 100.0, None, 0, 1, 4096, 2, 5000, 40, C.byref(seen),
                                             C.byref(kept), C.byref(dig), None) == 9      # LHGT_E_EMULATION: only the -t N emulation refuses (extract_ref falls back to -t 1)


def test_parallel_sam_ratio_equals_the_getline_pass(lib, oracle, case_inputs, tmp_path):
    """cal_sam_ratio (E:1244-1270) computed chunk-parallel == the oracle's line-by-line pass, whatever the line structure"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    fa, f1, f2, _ = case_inputs("k24_fq2_longer")
    raw = open(f2, "rb").read()
    variants = {"plain": raw, "nonl": raw[:-1], "crlf": raw[:50000].replace(b"\n", b"\r\n"), "blank_tail": raw + b"\n\n",
                "ragged": b"".join(ln + b"\n" for i, ln in enumerate(raw.split(b"\n")[:4001]) if i % 7), "empty": b"", "one": b"@x\nACGT"}
    for name, data in variants.items():
        p = str(tmp_path / f"{name}.fq")
        open(p, "wb").write(data)
        r, n = C.c_double(0), C.c_long(0)
        assert h.lhgt_fastq_sam_ratio(p.encode(), C.c_double(700000.0), C.byref(r), C.byref(n)) == 0
        want = oracle.sam_ratio(p, 700000.0)
        assert (r.value == want) or (np.isinf(r.value) and np.isinf(want)), (name, r.value, want)


@pytest.mark.parametrize("idx", range(10))
def test_thread_partition_fuzz_against_the_oracle(lib, oracle, tmp_path, idx):
    """random small inputs (ragged reads, padded fq2 headers, sampling) at -t 2..8: mate-1 / mate-2 / voted counts of the product's
    partitioned parse equal the oracle's -t N run wherever the product accepts the input (it refuses what the reference reads as garbage)"""
    import ctypes as C
    from test_gpu_fuzz import _make_case
    h = lib.load(require_gpu=False)
    d = tmp_path / "c"
    d.mkdir()
    k, e, seed, sample, hit, match, max_peak = _make_case(300 + idx, str(d), k_max=16)
    threads = 2 + idx % 7
    f1, f2, fa = str(d / "s.1.fq"), str(d / "s.2.fq"), str(d / "ref.fa")
    rc, rep = oracle.run_threads(f1, f2, fa, str(d / "i.txt"), 0.1, 0.08, threads, k, 100000 * threads, e, seed, sample)
    ratio = 100.0 * sample if sample <= 1 else oracle.sam_ratio(f1, sample)
    rnd = None
    if ratio < 100:
        oracle.srand(seed)
        oracle.random_coder(k, e)                 # the index was built in the oracle's run: the coder draws come first (quirk Q3)
        rnd = oracle.sampling_array(50_000_000)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    cnt = (C.c_long * 3)()
    rc2 = h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), ratio, rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None,
                                            0, 1, 4096, 3, 3000, threads, C.byref(seen), C.byref(kept), C.byref(dig), cnt)
    if rc in (-4, -6):
        assert rc2 == 4                            # chunk start near EOF / no re-synchronisation: refused by both
        return
    assert rc in (0, -5, -7)                       # -5 / -7: too many peaks for the id ranges -- the read partition is still defined
    if rc2 == 4:
        return                                     # overlapping chunks or a chunk entered off a record boundary: the product refuses
    assert rc2 == 0
    if rc == 0:
        assert (cnt[0], cnt[1], cnt[2]) == (rep.pairs_counted, int(rep.t_count), rep.pairs_voted), (threads, k, e, sample)


# ------------------------------------------------------------------ the single-pass loader == the planned loader (round 5)
def _digest_full(lib, f1, f2, ratio=100.0, rnd=None, threads=3, chunk=20000, emulate=1, stream=True):
    """(rc, seen, kept, digest, [mate 1 counted, mate 2 counted, voted], path, why) -- path 1 = the single pass did it"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    seen, kept, dig = C.c_long(0), C.c_long(0), C.c_uint64(0)
    cnt = (C.c_long * 3)()
    rp = rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None
    os.environ["LHGT_INGEST_STREAM"] = "1" if stream else "0"
    try:
        rc = h.lhgt_fastq_parse_digest_threads(f1.encode(), f2.encode(), float(ratio), rp, 0, 1, 4096, threads, chunk, emulate, C.byref(seen),
                                               C.byref(kept), C.byref(dig), cnt)
    finally:
        os.environ.pop("LHGT_INGEST_STREAM", None)
    why = C.create_string_buffer(256)
    path = h.lhgt_ingest_last_path(why, 256)
    return rc, seen.value, kept.value, dig.value, list(cnt), path, why.value.decode()


def test_single_pass_loader_equals_the_planned_loader(lib, oracle, case_inputs, tmp_path):
    """host_fastq_stream.cpp against host_fastx.cpp's two passes: the same pairs in the same order with the same flags (one digest)
    for every golden pair of files and for files made to hurt -- no final newline, CRLF, blank tail, a record cut short, fq2
    shorter / longer / with other header lengths (the chunks of the two files drift against each other), a line longer than a
    chunk's margin, foreign records in front -- at chunk sizes from one record to the whole file, with sampling, with the
    reference's -t N partition; and the single pass must really have been the one that ran wherever nothing forbids it"""
    oracle.srand(11)
    rnd = np.resize(oracle.sampling_array(1_000_000), 50_000_000)
    files = {}
    for name in ("k24_seed7", "k24_fq2_longer", "k24_fq2_surplus", "k24_fq2_stray2", "k24_fq2_short", "k24_fq2_short_nonl", "k24_t4", "k24_t10_sample_bases",
                 "k24_t3_fq2_longer"):
        fa, f1, f2, _ = case_inputs(name)
        files[name] = (f1, f2)
    raw1, raw2 = open(files["k24_seed7"][0], "rb").read(), open(files["k24_seed7"][1], "rb").read()
    l1, l2 = raw1.split(b"\n"), raw2.split(b"\n")

    def put(name, a, b):
        p1, p2 = str(tmp_path / f"{name}.1.fq"), str(tmp_path / f"{name}.2.fq")
        open(p1, "wb").write(a)
        open(p2, "wb").write(b)
        files[name] = (p1, p2)

    put("nonl", raw1[:-1], raw2[:-1])
    put("nonl2", raw1, raw2[:-1])
    put("crlf", b"\r\n".join(l1[:4000]) + b"\r\n", b"\r\n".join(l2[:4000]) + b"\r\n")
    put("blank_tail", raw1 + b"\n\n", raw2 + b"\n")
    put("cut_record", b"\n".join(l1[:4002]) + b"\n", b"\n".join(l2[:4002]) + b"\n")
    put("fq2_short", raw1, b"\n".join(l2[:2000]) + b"\n")
    put("fq2_short_nonl", raw1, b"\n".join(l2[:2002]))
    # longer headers in fq2 only: equal line counts, fq2 5 % larger -- the columns' chunks drift by lines
    put("drift", raw1, b"\n".join((ln + b" extra:comment" if i % 4 == 0 and ln else ln) for i, ln in enumerate(l2)))
    # ragged read lengths
    rng = np.random.default_rng(5)
    rag1, rag2 = [], []
    for i in range(0, min(len(l1), len(l2)) - 4, 4):
        a, b = int(rng.integers(30, 150)), int(rng.integers(30, 150))
        rag1 += [l1[i], l1[i + 1][:a], b"+", l1[i + 3][:a]]
        rag2 += [l2[i], l2[i + 1][:b], b"+", l2[i + 3][:b]]
    put("ragged", b"\n".join(rag1) + b"\n", b"\n".join(rag2) + b"\n")
    # a quality line of 70 000 characters: longer than the margin a chunk reads past its end
    long1 = l1[:400] + [b"@long/1", b"ACGT" * 25, b"+", b"I" * 70000] + l1[400:]
    long2 = l2[:400] + [b"@long/2", b"ACGT" * 25, b"+", b"I" * 70000] + l2[400:]
    put("long_line", b"\n".join(long1), b"\n".join(long2))
    put("foreign_front", raw1, b"@stray/2\nACGT\n+\nIIII\n" + raw2)
    put("other_first_id", raw1, b"@other/2\n" + b"\n".join(l2[1:]))
    taken, left = set(), {}
    for name, (f1, f2) in files.items():
        t = cases.CASES[name].threads if name in cases.CASES else 1
        for emulate in sorted({1, t, 4}):
            for ratio in (100.0, 35.0):
                want = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=2, chunk=1 << 40, emulate=emulate, stream=False)
                assert want[5] == 0
                for threads, chunk in ((1, 1 << 22), (3, 50000), (5, 7777), (4, 640)):
                    got = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=threads, chunk=chunk, emulate=emulate, stream=True)
                    assert got[:5] == want[:5], (name, emulate, ratio, threads, chunk, got[5:], want[:3])
                    if got[5] == 1:
                        taken.add((name, emulate))
                    else:
                        left[(name, emulate)] = got[6]
    # the plain files went through the single pass, also under the thread emulation where the reference's threads land on records
    for name in ("k24_seed7", "k24_fq2_longer", "k24_fq2_surplus", "k24_fq2_short", "k24_fq2_short_nonl", "nonl", "nonl2", "crlf", "blank_tail", "cut_record",
                 "fq2_short", "fq2_short_nonl", "drift", "ragged"):
        assert (name, 1) in taken, (name, left.get((name, 1)))
    assert ("k24_t4", 4) in taken and ("k24_seed7", 4) in taken, left
    # ... and these did not, for the reason given
    assert "first read IDs differ" in left[("k24_fq2_stray2", 1)] and "first read IDs differ" in left[("foreign_front", 1)]
    assert "first read IDs differ" in left[("other_first_id", 1)]
    assert "longer than a chunk's margin" in left[("long_line", 1)]


def test_planned_parse_takes_columns_when_the_plans_lie_on_the_grid(lib, oracle, case_inputs, tmp_path, monkeypatch):
    """plans made at lhgt_fastq_pair_chunk_bytes' sizes (fq2 in as many chunks as fq1: what extract_ref plans under --sample > 1 and
    in multi-rank runs) let every part be parsed column-wise with pread + newline lists (path 2): the parts' digests chain to the
    digest of the line-by-line loader; plans at other sizes go through the chunk loop (path 0) with the same result"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    oracle.srand(9)
    rnd = np.resize(oracle.sampling_array(1_000_000), 50_000_000)
    seeded = 0
    for name in ("k24_seed7", "k24_fq2_longer", "k24_fq2_short_nonl", "k24_t4", "k24_t10_sample_bases", "k24_fq2_stray2"):
        fa, f1, f2, _ = case_inputs(name)
        t = cases.CASES[name].threads
        for emulate in sorted({1, t}):
            for ratio in (100.0, 35.0):
                want = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=2, chunk=1 << 40, emulate=emulate, stream=False)
                for chunk, parts in ((50000, 1), (7777, 3), (300000, 2)):
                    monkeypatch.setenv("LHGT_INGEST_CHUNK_BYTES", str(chunk))
                    c1, c2 = C.c_long(0), C.c_long(0)
                    assert h.lhgt_fastq_pair_chunk_bytes(f1.encode(), f2.encode(), C.byref(c1), C.byref(c2)) == 0 and c1.value == chunk
                    for on_grid in (True, False):
                        plan1, plan2 = _plan(lib, f1, c1.value, parts), _plan(lib, f2, c2.value if on_grid else c1.value + 13, parts)
                        state, kept, cnt, paths = None, 0, [0, 0, 0], set()
                        rc = 0
                        for part in range(parts):
                            rc, sn, kp, dg, c3 = _digest_planned(lib, f1, f2, plan1, plan2, part, parts, state=state if part else None, ratio=ratio, rnd=rnd,
                                                                 emulate=emulate, chunk=c1.value)
                            if rc:
                                break
                            state, kept, cnt = dg, kept + kp, [a + b for a, b in zip(cnt, c3)]
                            paths.add(h.lhgt_ingest_last_path(None, 0))
                        assert rc == want[0], (name, emulate, ratio, chunk, parts, on_grid)
                        if rc == 0:
                            assert (kept, state, cnt) == (want[2], want[3], want[4]), (name, emulate, ratio, chunk, parts, on_grid, paths)
                            if on_grid and name != "k24_fq2_stray2":
                                assert paths == {2}, (name, emulate, chunk, parts, paths)
                                seeded += 1
                            if not on_grid:
                                assert 2 not in paths
    assert seeded > 40


def _digest_slabs(lib, f1, f2, ratio=100.0, rnd=None, threads=3, chunk=20000, emulate=1, stream=True):
    """the same digest, but through the slab pool the GPU loader parses into (lhgt_fastq_parse_rate): bases and records read back from
    the blocks that would be copied to the device"""
    import ctypes as C
    h = lib.load(require_gpu=False)
    seen, kept, bases, dig, secs = C.c_long(0), C.c_long(0), C.c_long(0), C.c_uint64(0), C.c_double(0)
    rp = rnd.ctypes.data_as(C.POINTER(C.c_float)) if rnd is not None else None
    os.environ["LHGT_INGEST_STREAM"] = "1" if stream else "0"
    try:
        rc = h.lhgt_fastq_parse_rate(f1.encode(), f2.encode(), float(ratio), rp, threads, chunk, emulate, None, None, 0, None, None, 0, 0, 1,
                                     C.byref(seen), C.byref(kept), C.byref(bases), C.byref(secs), C.byref(dig))
    finally:
        os.environ.pop("LHGT_INGEST_STREAM", None)
    return rc, seen.value, kept.value, dig.value


def test_slab_blocks_hold_what_the_vector_parse_holds(lib, oracle, case_inputs, tmp_path):
    """the loader's slab form (one run of bases per chunk, the mates of a pair back to back, records behind them) read back == the
    digest of the vector form, for both loaders, sampling and the thread emulation included; chunks small enough to spill"""
    oracle.srand(4)
    rnd = np.resize(oracle.sampling_array(1_000_000), 50_000_000)
    for name in ("k24_seed7", "k24_fq2_longer", "k24_fq2_short_nonl", "k24_t4", "k24_fq2_stray2"):
        fa, f1, f2, _ = case_inputs(name)
        for emulate in (1, 4):
            for ratio in (100.0, 35.0):
                want = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=2, chunk=1 << 40, emulate=emulate, stream=False)
                for stream in (True, False):
                    for threads, chunk in ((3, 50000), (4, 2000), (2, 1 << 22)):
                        got = _digest_slabs(lib, f1, f2, ratio=ratio, rnd=rnd, threads=threads, chunk=chunk, emulate=emulate, stream=stream)
                        assert got[0] == want[0] and (got[0] != 0 or got == want[:4]), (name, emulate, ratio, stream, threads, chunk)   # a refusal is a refusal either way


@pytest.mark.parametrize("idx", range(24))
def test_single_pass_loader_fuzz(lib, oracle, tmp_path, idx):
    """random small inputs of the whole-run fuzz (ragged reads, N runs, padded fq2 headers, CRLF, sampling): single pass == planned
    loader, plain and under -t 2..8, whichever of the two ends up doing the work"""
    from test_gpu_fuzz import _make_case
    d = tmp_path / "c"
    d.mkdir()
    k, e, seed, sample, hit, match, max_peak = _make_case(700 + idx, str(d), k_max=16)
    f1, f2 = str(d / "s.1.fq"), str(d / "s.2.fq")
    oracle.srand(seed)
    rnd = np.resize(oracle.sampling_array(200_000), 50_000_000)
    ratio = 100.0 * sample if sample <= 1 else 40.0
    for emulate in (1, 2 + idx % 7):
        want = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=2, chunk=1 << 40, emulate=emulate, stream=False)
        for threads, chunk in ((3, 3000), (4, 500), (2, 1 << 22)):
            got = _digest_full(lib, f1, f2, ratio=ratio, rnd=rnd, threads=threads, chunk=chunk, emulate=emulate, stream=True)
            assert got[:5] == want[:5], (idx, emulate, threads, chunk, got[5:])


def test_fasta_line_structure_on_the_host(lib, oracle, tmp_path):
    """the host half of the FASTA loaders ('>' lines by memchr, newline counts per 4 KiB block, lengths from the counts; the bases
    go to the GPU as text): genome.len.txt and the contig count against the restatement's read_ref on files with unusual line
    structure (tests/cases.py: odd_fastas; the restatement is pinned on them against the reference binary)"""
    import ctypes as C
    import cases
    h = lib.load(require_gpu=False)
    for k in (16, 40):
        for name, text in cases.odd_fastas():
            fa = str(tmp_path / f"{name}.fa")
            open(fa, "wb").write(text)
            ns, nc, nb = C.c_long(0), C.c_long(0), C.c_long(0)
            lib.check(h.lhgt_fasta_scan(fa.encode(), k, (fa + ".len").encode(), C.byref(ns), C.byref(nc), C.byref(nb)))
            cc = np.zeros(300, dtype=np.int16)
            assert oracle.index_build(fa, fa + ".oidx", fa + ".olen", min(k, 32), 3, cc) == nc.value or k > 32, name
            if k <= 32:
                assert open(fa + ".len").read() == open(fa + ".olen").read(), name
            headers = sum(1 for ln in text.split(b"\n") if ln.startswith(b">"))
            assert ns.value == headers + 1, name
            want = [len(b"".join(part.split(b"\n")[1:])) for part in (b"\n" + text).split(b"\n>")]
            want[0] = len(text.split(b"\n>")[0].replace(b"\n", b"")) if not text.startswith(b">") else 0
            got = [int(ln.split("\t")[2]) for ln in open(fa + ".len")]
            assert got == [w for w in want if w > k], (name, k)


def test_packed_sample_header_on_the_host(lib, tmp_path):
    """localhgt_amd/pack.py without a GPU: what the loader reads back from a packed sample's header -- cal_sam_ratio from the stored base
    count (E:1244-1270, 1392-1398), the thread chunks of the reference's -t N or the refusal stored in their place, a thread count the
    file was not packed for, and the host side of a load (lhgt_packed_read_rate reads the records of part i of n, nothing else)"""
    import json
    import struct
    from localhgt_amd import _lib, pack
    n_pairs, stride = 1000, 4 + 24 * 6
    hdr = {"version": 1, "n_pairs": n_pairs, "stride": stride, "max_len": 150, "q4_first_pair": 990, "fq1_bases": 150_000, "data_offset": pack.DATA_OFFSET,
           "lines": 4000, "max_threads": 3,
           "threads": {"2": {"first1": [0, 2001], "count1": [2000, 1999], "first2": [0, 2001], "count2": [2000, 1999]},
                       "3": {"refused": [9, "thread 2 of 3 starts inside a record"]}}}
    blob = json.dumps(hdr).encode()
    path = str(tmp_path / "s.lhgp")
    with open(path, "wb") as f:
        f.write(pack.MAGIC + struct.pack("<Q", len(blob)) + blob)
        f.truncate(pack.DATA_OFFSET + n_pairs * stride)
    assert pack.is_packed(path) and not pack.is_packed(__file__) and not pack.is_packed(str(tmp_path / "absent"))
    h = pack.read_header(path)
    assert (h.n_pairs, h.stride, h.q4_first_pair, h.data_offset) == (n_pairs, stride, 990, pack.DATA_OFFSET)
    assert h.ratio(0.5) == 50.0 and h.ratio(1) == 100.0
    assert h.ratio(60_000) == 100.0 * 60_000 / (2.0 * 150_000)                 # --sample > 1: bases wanted over twice the bases of fq1
    f1, c1, f2, c2 = h.thread_chunks(2)
    assert list(f1) == [0, 2001] and list(c2) == [2000, 1999] and c1.dtype == np.int64
    for t in (3, 7):                                                           # the stored refusal; a count the sample was not packed for
        with pytest.raises(_lib.LocalHGTError) as ex:
            h.thread_chunks(t)
        assert ex.value.code == 9
    with pytest.raises(SystemExit):
        pack.read_header(__file__)
    h_lib = lib.load(require_gpu=False)
    secs = ctypes.c_double(-1)
    for part in range(3):
        assert h_lib.lhgt_packed_read_rate(path.encode(), pack.DATA_OFFSET, stride, n_pairs, part, 3, 2, ctypes.byref(secs)) == 0 and secs.value >= 0
    with open(path, "r+b") as f:
        f.truncate(pack.DATA_OFFSET + (n_pairs - 10) * stride)                 # the records end early: an I/O error, not a short read taken for data
    assert h_lib.lhgt_packed_read_rate(path.encode(), pack.DATA_OFFSET, stride, n_pairs, 2, 3, 2, ctypes.byref(secs)) != 0


def test_host_packer_writes_the_store_records(lib, tmp_path):
    """`localhgt_pack --host` (lhgt_fastq_pack_host; round 6, late): a record-aligned pair of files packed on the host's CPUs, no GPU touched --
    every record against a literal restatement of the store's layout ([u16 len1][u16 len2][mate 1: hi, lo, not-a-base planes of
    len / 32 + 1 words, bit 31 - b of word w = base 32 w + b][mate 2 likewise], zero to the stride), ragged lengths 0 .. 259, lower case,
    N's and other letters; the header's pair count, stride, base count of fq1 (cal_sam_ratio, E:1264-1265) and quirk Q4's first pair
    (mate 2's record starting behind size(fq1), E:1419-1445); files that are no clean pair are refused"""
    import struct
    from localhgt_amd import _lib, pack
    rng = np.random.default_rng(5)
    letters = np.frombuffer(b"ACGTNacgtnRy", dtype=np.uint8)
    n = 3000

    def reads():
        return [letters[rng.choice(12, size=int(rng.integers(0, 260)), p=[.21, .21, .21, .21, .01, .03, .03, .03, .03, .01, .01, .01])].tobytes() for _ in range(n)]

    r1, r2 = reads(), reads()
    paths = []
    for name, rs, tag in (("a.1.fq", r1, b"/1"), ("a.2.fq", r2, b"/2")):
        paths.append(str(tmp_path / name))
        with open(paths[-1], "wb") as f:
            for i, r in enumerate(rs):
                f.write(b"@read%d%s\n" % (i, tag) + r + b"\n+\n" + b"I" * len(r) + b"\n")
    out = str(tmp_path / "a.lhgp")
    hdr = pack.pack(paths[0], paths[1], out, max_threads=4, host=True, log=lambda *a: None)
    raw = open(out, "rb").read()
    assert hdr["n_pairs"] == n and hdr["max_len"] == max(map(len, r1 + r2)) and hdr["stride"] == 4 + 24 * ((hdr["max_len"] + 31) // 32 + 1)
    assert len(raw) == hdr["data_offset"] + n * hdr["stride"] and hdr["fq1_bases"] == sum(map(len, r1))
    # quirk Q4: the first pair whose mate 2 record starts at or behind size(fq1) in fq2
    size1, at, q4 = os.path.getsize(paths[0]), 0, n
    for i, r in enumerate(r2):
        if at >= size1:
            q4 = i
            break
        at += len(b"@read%d/2\n" % i) + 2 * len(r) + 4
    assert hdr["q4_first_pair"] == q4
    code = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}

    def planes(s):
        wpr = (len(s) + 31) // 32 + 1
        w = np.zeros(3 * wpr, dtype=np.uint32)
        for j, c in enumerate(s):
            q, b = divmod(j, 32)
            v = code.get(c, 4)
            if v == 4:
                w[2 * wpr + q] |= 0x80000000 >> b
            else:
                w[q] |= (0x80000000 >> b) if v & 2 else 0
                w[wpr + q] |= (0x80000000 >> b) if v & 1 else 0
        return w

    for i in range(n):
        rec = raw[hdr["data_offset"] + i * hdr["stride"]: hdr["data_offset"] + (i + 1) * hdr["stride"]]
        assert struct.unpack("<HH", rec[:4]) == (len(r1[i]), len(r2[i]))
        want = np.concatenate([planes(r1[i]), planes(r2[i])])
        assert (np.frombuffer(rec[4:4 + 4 * len(want)], dtype=np.uint32) == want).all() and not any(rec[4 + 4 * len(want):]), i
    assert pack.read_header(out).thread_chunks(2)[0].shape == (2,)
    # no clean pair: one record fewer in fq2
    short = str(tmp_path / "b.2.fq")
    with open(short, "wb") as f:
        for i, r in enumerate(r2[:-1]):
            f.write(b"@read%d/2\n" % i + r + b"\n+\n" + b"I" * len(r) + b"\n")
    with pytest.raises((SystemExit, _lib.LocalHGTError)):
        pack.pack(paths[0], short, str(tmp_path / "b.lhgp"), max_threads=2, host=True, log=lambda *a: None)
