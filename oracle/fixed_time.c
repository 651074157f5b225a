/* fixed_time.c -- determinism shim for the reference's count_diff_kmer (test infrastructure only).
 * The tool seeds its coder and its sampling from time(0) (count_diff_kmer.cpp:87-89, 223-225), so two runs differ.
 * Preloaded (LD_PRELOAD) next to seq_threads.c, time() returns LHGT_FIXED_TIME (default 1): the run becomes a function of
 * its inputs.  Sources untouched. */
#include <stdlib.h>
#include <time.h>
time_t time(time_t* t) {
    const char* e = getenv("LHGT_FIXED_TIME");
    time_t v = e ? (time_t)atol(e) : (time_t)1;
    if (t) *t = v;
    return v;
}
