#!/bin/bash
# Build the REAL reference binaries from the sources where they lie under
# /root/reference (never copied into this repo).  Outputs go to oracle/_ref/ only
# (git-ignored, but shipped to the GPU box by gpurun like any other built artefact).
#   extract_ref_z   : extract_ref_normal_peak.cpp + zero_new.h  (deterministic oracle)
#   extract_ref_raw : the same source with no shim               (what `make` would build, at -O2)
#   count_diff_kmer : src/count_diff_kmer.cpp
#   libseqthreads.so: oracle/seq_threads.c, LD_PRELOADed for -t N runs
#   libfixedtime.so : oracle/fixed_time.c, LD_PRELOADed for count_diff_kmer
# Test infrastructure only: nothing in the product path may execute these.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
ref="${LHGT_REFERENCE_DIR:-/root/reference}"
if [ ! -f "$ref/src/extract_ref_normal_peak.cpp" ]; then
    echo "build_ref.sh: $ref not present; keeping whatever is already in oracle/_ref" >&2
    exit 0
fi
mkdir -p "$here/_ref"
g++ -O2 -std=c++11 -pthread -w -include "$here/zero_new.h" -o "$here/_ref/extract_ref_z" "$ref/src/extract_ref_normal_peak.cpp"
g++ -O2 -std=c++11 -pthread -w -o "$here/_ref/extract_ref_raw" "$ref/src/extract_ref_normal_peak.cpp"
g++ -O2 -std=c++11 -pthread -w -o "$here/_ref/count_diff_kmer" "$ref/src/count_diff_kmer.cpp"
# determinism shims for multi-threaded runs of those binaries (our own files, preloaded; the sources stay untouched):
#   libseqthreads.so : threads run one after the other in creation order (the -t N contract)
gcc -O2 -shared -fPIC -o "$here/_ref/libseqthreads.so" "$here/seq_threads.c"
#   libfixedtime.so  : time() returns LHGT_FIXED_TIME (count_diff_kmer seeds its coder and sampling from it)
gcc -O2 -shared -fPIC -o "$here/_ref/libfixedtime.so" "$here/fixed_time.c"
echo "built: $(ls "$here/_ref")"
