#!/bin/bash
# round-3 evidence, part C (GPU box): extract_ref from 20 GB of FASTQ; random-probe rates inside / beyond the Infinity Cache;
# extract_ref from files at catalogue scale under the CLI's default sampling -- a 13 Gbase FASTA
# (packed reference, no index file) and 50 M pairs (32 GB of FASTQ), --sample 2000000000 -t 10
out=gpurun_out/$1; mkdir -p $out/profiles
timeout -k 10 300 python3 tools/e2e_big.py 32000000 100 > $out/profiles/e2e_from_files_32m_pairs.txt 2>&1 || { tail -5 $out/profiles/e2e_from_files_32m_pairs.txt; exit 1; }
timeout -k 10 100 ./tools/probe_shapes mall > $out/profiles/probe_mall_share.txt 2>&1 || exit 1
LHGT_REF_FORM=packed LHGT_TRACE=1 E2E_ONLY_PACKED=1 timeout -k 10 1000 python3 tools/e2e_big.py 50000000 13000 2000000000 > $out/profiles/e2e_from_files_13gbase_50m_pairs_default_sample.txt 2>&1 || { tail -5 $out/profiles/e2e_from_files_13gbase_50m_pairs_default_sample.txt; exit 1; }
tail -12 $out/profiles/e2e_from_files_13gbase_50m_pairs_default_sample.txt
