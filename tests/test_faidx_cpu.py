"""BED -> extracted FASTA (`samtools faidx -r`, pipeline.sh:37) and the .fai index.
samtools is absent, so the check is the C-ABI (host code, no GPU needed) against oracle/faidx_port.py plus hand-written
known answers for the documented samtools behaviour (parity unpinned, see the oracle's header)."""
import os
import subprocess
import sys

import pytest

import cases

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import faidx_port  # noqa: E402

from localhgt_amd import faidx  # noqa: E402
from localhgt_amd._lib import LocalHGTError  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FA = (">chrA first contig\n"
      "ACGTACGTAC\n"
      "GTACGTACGT\n"
      "acgtn\n"
      ">chrB/1\tdescription\n"
      "TTTTTGGGGGCCCCCAAAAA\n"
      ">chrC:1-5 a name with a colon\n"
      "AAAACCCCGGGGTTTT\n"
      "AC\n")


def _write(tmp_path, name, text):
    p = str(tmp_path / name)
    with open(p, "w") as f:
        f.write(text)
    return p


def test_fai_known_answer(tmp_path):
    fa = _write(tmp_path, "ref.fa", FA)
    assert faidx.build_fai(fa) == 3
    want = "chrA\t25\t19\t10\t11\nchrB/1\t20\t67\t20\t21\nchrC:1-5\t18\t118\t16\t17\n"
    assert open(fa + ".fai").read() == want == faidx_port.fai_text(fa)


def test_extract_known_answer(tmp_path):
    fa = _write(tmp_path, "ref.fa", FA)
    bed = _write(tmp_path, "r.bed", "chrA:3-14\nchrB/1:18-500\nchrA:21-25\nchrC:1-5\nchrC:1-5:2-3\nchrA:26-30\nchrB/1\n\nchrA:1,0-1,2\n")
    out = str(tmp_path / "out.fa")
    n_reg, n_bases = faidx.extract_regions(fa, bed, out, 8)
    want = (">chrA:3-14\nGTACGTAC\nGTAC\n"          # crosses a line break of the file; wrapped at 8
            ">chrB/1:18-500\nAAA\n"                   # end past the contig: truncated
            ">chrA:21-25\nacgtn\n"                     # case and N kept; last, shorter line of the sequence
            ">chrC:1-5\nAAAACCCC\nGGGGTTTT\nAC\n"     # the whole string is a sequence name: taken whole
            ">chrC:1-5:2-3\nAA\n"
            ">chrA:26-30\n"                            # begins past the end: header only
            ">chrB/1\nTTTTTGGG\nGGCCCCCA\nAAAA\n"
            ">chrA:1,0-1,2\nCGT\n")                    # thousands separators
    assert open(out).read() == want == faidx_port.extract_text(fa, bed, 8)
    assert (n_reg, n_bases) == (8, 12 + 3 + 5 + 18 + 2 + 0 + 20 + 3)


def test_unknown_region_and_ragged_lines_fail(tmp_path):
    fa = _write(tmp_path, "ref.fa", FA)
    bed = _write(tmp_path, "r.bed", "chrA:1-4\nnope:1-4\n")
    with pytest.raises(LocalHGTError, match="Failed to fetch sequence in nope:1-4"):
        faidx.extract_regions(fa, bed, str(tmp_path / "o.fa"))
    bad = _write(tmp_path, "bad.fa", ">x\nACGT\nAC\nACGT\n")
    with pytest.raises(LocalHGTError, match="Different line length in sequence 'x'"):
        faidx.build_fai(bad, None)
    with pytest.raises(ValueError):
        faidx_port.fai_table(bad)


def test_crlf_and_no_trailing_newline(tmp_path):
    fa = str(tmp_path / "crlf.fa")
    with open(fa, "wb") as f:
        f.write(b">s1\r\nACGTAC\r\nGTACGT\r\nAC\r\n>s2\r\nGGGG")
    bed = _write(tmp_path, "r.bed", "s1:5-9\ns2:2-3")
    out = str(tmp_path / "o.fa")
    faidx.extract_regions(fa, bed, out)
    assert open(out).read() == ">s1:5-9\nACGTA\n>s2:2-3\nGG\n" == faidx_port.extract_text(fa, bed)
    faidx.build_fai(fa)
    assert open(fa + ".fai").read() == "s1\t14\t5\t6\t8\ns2\t4\t30\t4\t4\n" == faidx_port.fai_text(fa)


@pytest.mark.parametrize("name", ["k24_base", "k24_nrun_lower", "k24_seed7"])
def test_golden_case_bed_to_fasta(case_inputs, tmp_path, name):
    """the .bed the reference wrote for a golden case (tests/golden/<case>/interval.txt.bed) through the extractor:
    equals the restatement, every record is its region of the reference, ends past a contig are cut"""
    fa, _, _, _ = case_inputs(name)
    bed = os.path.join(cases.GOLDEN_DIR, name, "interval.txt.bed")
    out = str(tmp_path / "specific.ref.fasta")
    n_reg, n_bases = faidx.extract_regions(fa, bed, out)
    text = open(out).read()
    assert text == faidx_port.extract_text(fa, bed)
    regions = [l.strip() for l in open(bed) if l.strip()]
    assert n_reg == len(regions) > 0 and text.count(">") == n_reg
    lens = {r[0]: r[1] for r in faidx_port.fai_table(fa)}
    total = 0
    for reg in regions:
        nm, span = reg.rsplit(":", 1)
        b, e = (int(x) for x in span.split("-"))
        total += max(0, min(e, lens[nm]) - b + 1)
    assert n_bases == total
    assert all(len(l) <= 60 for l in text.splitlines() if not l.startswith(">"))


def test_cli_matches_pipeline_usage(case_inputs, tmp_path):
    """`localhgt_faidx faidx -r x.bed ref.fa > out` (pipeline.sh:37) and `localhgt_faidx faidx ref.fa` (B:156)"""
    fa, _, _, _ = case_inputs("k24_seed7")
    bed = os.path.join(cases.GOLDEN_DIR, "k24_seed7", "interval.txt.bed")
    exe = os.path.join(ROOT, "bin", "localhgt_faidx")
    res = subprocess.run([sys.executable, exe, "faidx", "-r", bed, fa], capture_output=True, text=True, check=True)
    assert res.stdout == faidx_port.extract_text(fa, bed)
    import shutil
    fa2 = str(tmp_path / "ref.fa")
    shutil.copy(fa, fa2)
    subprocess.run([sys.executable, exe, "faidx", fa2], check=True)
    assert open(fa2 + ".fai").read() == faidx_port.fai_text(fa2)
    bad = subprocess.run([sys.executable, exe, "faidx", "-r", _write(tmp_path, "b.bed", "zz:1-2\n"), fa], capture_output=True, text=True)
    assert bad.returncode == 1 and "Failed to fetch sequence in zz:1-2" in bad.stderr
