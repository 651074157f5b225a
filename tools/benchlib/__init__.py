"""Helpers of the repo-root bench.py (measurement only; nothing here is on the product path).

  files.py     synthetic FASTA / FASTQ files for the from-files and CPU legs
  recall.py    where the synthetic sample's transfers are, and whether an interval file holds them
  pmc.py       rocprofv3 --pmc child runs of bench.py and their per-kernel sums
  roofline.py  the byte models (SURVEY 8d's and the bytes the algorithm as built must move) and the roofline entries
  legs.py      the timed loop and the secondary workloads
  cpu.py       the CPU legs: the restatement (oracle/) and the compiled reference (oracle/_ref/) on the same files
  compact.py   the < 4 KB line the driver parses, assembled from the detail record
  launch.py    --gpus N without a launcher, --dry-run
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
METRIC = "M paired-reads/s k-mer sketch->peak, UHGG-scale ref; %HBM roofline @1/2/4/8 GPU"
