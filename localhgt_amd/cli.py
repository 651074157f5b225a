"""`localhgt bkp` -- the reference's command line for the breakpoint stage, kept flag for flag
(reference: scripts/localhgt.py:37-64 and the duplicate parser scripts/infer_HGT_breakpoint.py:189-212).

`bkp` checks its inputs the way check_input does (B:123-166), then assembles the same
`bash <dir>/pipeline.sh <16 tokens>` command (B:28-31) and runs it (B:33-34).  pipeline.sh is the
reference's own, unchanged: it calls `extract_ref` (this package's GPU engine, bin/extract_ref) and
`get_bed_file.py`, then samtools/bwa.  `--pipeline` points at a pipeline.sh (default: next to this
program, as in the reference); `--dry-run` prints the command without running it.
The `event` sub-command and the `--use_kmer 0` whole-reference alignment are outside this
package's scope and are refused with a message."""
from __future__ import annotations

import argparse
import os
import sys
from shutil import which


def build_parser(prog: str = "localhgt bkp") -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog=prog, description="Detect HGT breakpoints from metagenomics sequencing data.",
                                add_help=False, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    required = p.add_argument_group("required arguments")
    optional = p.add_argument_group("optional arguments")
    required.add_argument("-r", type=str, help="<str> Uncompressed reference file, which contains all the representative references of concerned bacteria.", metavar="\b")
    required.add_argument("--fq1", type=str, help="<str> Uncompressed fastq 1 file.", metavar="\b")
    required.add_argument("--fq2", type=str, help="<str> Uncompressed fastq 2 file.", metavar="\b")
    required.add_argument("-s", type=str, default="sample", help="<str> Sample name.", metavar="\b")
    required.add_argument("-o", type=str, default="./", help="<str> Output folder.", metavar="\b")
    optional.add_argument("-k", type=int, default=32, help="<int> kmer length.", metavar="\b")
    optional.add_argument("-t", type=int, default=10, help="<int> number of threads.", metavar="\b")
    optional.add_argument("-e", type=int, default=3, help="<int> number of hash functions (1-9).", metavar="\b")
    optional.add_argument("-a", type=int, default=1, help="<0/1> 1 indicates retain reads with XA tag.", metavar="\b")
    optional.add_argument("-q", type=int, default=20, help="<int> minimum read mapping quality in BAM.", metavar="\b")
    optional.add_argument("--seed", type=int, default=1, help="<int> seed to initialize a pseudorandom number generator.", metavar="\b")
    optional.add_argument("--use_kmer", type=int, default=1, help="<1/0> 1 means using kmer to extract HGT-related segment, 0 means using original reference.", metavar="\b")
    optional.add_argument("--hit_ratio", type=float, default=0.1, help="<float> minimum fuzzy kmer match ratio to extract a reference fragment.", metavar="\b")
    optional.add_argument("--match_ratio", type=float, default=0.08, help="<float> minimum exact kmer match ratio to extract a reference fragment.", metavar="\b")
    optional.add_argument("--max_peak", type=int, default=300000000, help="<int> maximum candidate BKP count.", metavar="\b")
    optional.add_argument("--sample", type=float, default=2000000000, help="<float> down-sample in kmer counting: (0-1) means sampling proportion, (>1) means sampling base count (bp).", metavar="\b")
    optional.add_argument("--refine_fq", type=int, default=0, help="<0/1> 1 indicates refine the input fastq file using fastp (recommended).", metavar="\b")
    optional.add_argument("--read_info", type=int, default=1, help="<0/1> 1 indicates including reads info, 0 indicates not (just for evaluation).", metavar="\b")
    optional.add_argument("-h", "--help", action="help")
    ours = p.add_argument_group("localhgt-mi355x additions")
    ours.add_argument("--pipeline", type=str, default=None, help="path of the reference's pipeline.sh (default: next to this program)")
    ours.add_argument("--dry-run", action="store_true", help="print the command and stop")
    return p


def run_order(o, fastq_1: str, fastq_2: str, shell_script: str) -> str:
    """Accept_Parameters.get_order (B:28-31): 16 tokens after the script path."""
    return (f"bash {shell_script} {o.r} {fastq_1} {fastq_2} {o.s} {o.o} {o.hit_ratio} {o.match_ratio} {o.t} {o.k} "
            f"{o.max_peak} {o.e} {o.seed} {o.sample} {o.read_info} {o.a} {o.q}\n")


def is_file_zipped(path: str) -> bool:
    suffixes = ['.gz', '.bz2', '.xz', '.zip', '.tar', '.tgz']
    return path[-3:].lower() in suffixes or path[-4:].lower() in suffixes


def check_input(o, need_tools=("samtools", "bwa")) -> None:
    """check_input (B:123-166): same messages, same exit status 1."""
    def die(msg):
        print(msg)
        sys.exit(1)
    if not os.path.isfile(o.r):
        die("Error: reference file is not detected.")
    if not os.path.isfile(o.fq1) or not os.path.isfile(o.fq2):
        die("Error: fastq file is not detected.")
    if is_file_zipped(o.r):
        die("Error: reference file should be uncompressed.")
    for tool in need_tools:
        if which(tool) is None:
            die(f"Error: {tool} is not installed.")
    if o.refine_fq == 1 and which("fastp") is None:
        die("Error: fastp is not installed.")
    if not os.path.isdir(o.o):
        print(f"Output folder {o.o} is constructed.")
        os.makedirs(o.o, exist_ok=True)
    if not os.path.isfile(o.r + ".fai") and which("samtools"):
        print("construct samtools index for the refernece...")
        os.system(f"samtools faidx {o.r}")
    if is_file_zipped(o.fq1) or is_file_zipped(o.fq2):
        if (o.refine_fq == 0 and o.use_kmer == 1) or o.fq1[-3:] != ".gz":
            die("Error: input fastq file should be uncompressed.")


def refine_fastq(o):
    if o.refine_fq == 1:  # B:99-109
        f1, f2 = f"{o.o}/{o.s}_refined_1.fq", f"{o.o}/{o.s}_refined_2.fq"
        print("refine input fastq files...")
        os.system(f"fastp -i {o.fq1} -I {o.fq2} -o {f1} -O {f2}")
        return f1, f2
    return o.fq1, o.fq2


def detect_breakpoint(o, prog_dir: str) -> int:
    shell_script = o.pipeline or (prog_dir + "/pipeline.sh")
    if o.dry_run:
        print("Running command:")
        print(run_order(o, o.fq1, o.fq2, shell_script))
        return 0
    check_input(o)
    fastq_1, fastq_2 = refine_fastq(o)
    if o.use_kmer != 1:
        print("## --use_kmer 0 (whole-reference bwa alignment) bypasses the k-mer stage and is not part of localhgt-mi355x.")
        return 1
    if which("extract_ref") is None:
        print("Error: extract_ref is not installed, please check the installation.")
        return 1
    cmd = run_order(o, fastq_1, fastq_2, shell_script)
    print("Running command:")
    print(cmd)
    os.system(cmd)
    return 0


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    prog_dir = os.path.dirname(sys.argv[0])
    if argv and argv[0] == "event":
        print("`localhgt event` (event inference from breakpoints) is outside localhgt-mi355x; use the reference's.")
        return 1
    if argv and argv[0] == "bkp":
        argv = argv[1:]
    p = build_parser()
    if not argv:
        p.print_help()
        return 0
    o = p.parse_args(argv)
    if o.r is None:
        p.print_help()
        return 0
    return detect_breakpoint(o, prog_dir)


if __name__ == "__main__":
    sys.exit(main())
