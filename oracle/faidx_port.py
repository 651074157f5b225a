"""TEST INFRASTRUCTURE ONLY -- CPU restatement of `samtools faidx` for the two forms on LocalHGT's path
(scripts/pipeline.sh:37 `samtools faidx -r bed ref > out`, scripts/infer_HGT_breakpoint.py:156 `samtools faidx ref`).

PARITY UNPINNED: samtools/htslib are a third-party dependency of the reference (no version pinned by it; its README asks for
"samtools"), absent from /root/reference and from this image.  The restatement follows the published behaviour of samtools 1.x
(htslib faidx.c: fai_build / fai_parse_region / fai_fetch, samtools faidx.c: write_output): plain Python, line by line, and
deliberately shares no code with localhgt_amd/csrc/host_faidx.cpp."""


def fai_table(fasta_path):
    """[(name, length, offset, linebases, linewidth)] -- the rows of <fasta>.fai"""
    rows, cur, pos = [], None, 0
    with open(fasta_path, "rb") as f:
        for raw in f:
            if raw.startswith(b">"):
                if cur:
                    rows.append(tuple(cur))
                name = raw[1:].split()[0].decode() if raw[1:].split() else ""
                cur = [name, 0, pos + len(raw), 0, 0]
                state = {"first": True, "short": False}
            elif cur is not None:
                bases = len(raw.rstrip())
                if state["short"] and bases:
                    raise ValueError(f"Different line length in sequence '{cur[0]}'")
                if state["first"]:
                    if bases or raw.strip(b"\n"):
                        cur[3], cur[4] = bases, len(raw)
                        state["first"] = False
                elif bases != cur[3] or len(raw) != cur[4]:
                    if bases > cur[3]:
                        raise ValueError(f"Different line length in sequence '{cur[0]}'")
                    state["short"] = True
                cur[1] += bases
            pos += len(raw)
    if cur:
        rows.append(tuple(cur))
    return rows


def fai_text(fasta_path):
    return "".join("%s\t%d\t%d\t%d\t%d\n" % r for r in fai_table(fasta_path))


def _sequence(fasta_path, name):
    out, on = [], False
    with open(fasta_path, "rb") as f:
        for raw in f:
            if raw.startswith(b">"):
                if on:
                    break
                on = raw[1:].split()[:1] == [name.encode()]
            elif on:
                out.append(raw.strip())
    return b"".join(out)


def extract_text(fasta_path, regions_path, width=60):
    names = [r[0] for r in fai_table(fasta_path)]
    out = []
    for line in open(regions_path):
        reg = line.strip()
        if not reg:
            continue
        if reg in names:
            name, beg, end = reg, 1, None
        else:
            name, _, span = reg.rpartition(":")
            if name not in names:
                raise KeyError(f"Failed to fetch sequence in {reg}")
            span = span.replace(",", "")
            b, dash, e = span.partition("-")
            beg, end = int(b), (int(e) if e else None)
        seq = _sequence(fasta_path, name)
        lo = max(beg, 1) - 1
        hi = len(seq) if end is None else min(end, len(seq))
        sub = seq[lo:hi] if lo < hi else b""
        out.append(">" + reg + "\n")
        out.extend(sub[i:i + width].decode() + "\n" for i in range(0, len(sub), width))
    return "".join(out)
