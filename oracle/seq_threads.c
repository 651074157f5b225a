/* seq_threads.c -- determinism shim for running the REFERENCE binary with -t N (test infrastructure only).
 *
 * The reference's threads race on its tables (non-atomic counters, cross-thread peak_kmer writes), so a
 * multi-threaded run is not reproducible.  Preloaded into oracle/_ref/extract_ref_z (LD_PRELOAD), this file makes
 * pthread_create run the thread's function to completion before it returns: the N "threads" of every fork-join phase
 * (E:1426-1507) execute one after the other in creation order.  That is one legal schedule of the reference -- the one
 * without any race -- and it is the contract of the `-t N` emulation: per-thread byte chunks of the
 * FASTQs, per-chunk sampling ordinals, per-thread peak id ranges, later thread wins in peak_kmer.  Sources untouched.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>

static uintptr_t next_id = 1;

int pthread_create(pthread_t* thread, const pthread_attr_t* attr, void* (*start)(void*), void* arg) {
    (void)attr;
    start(arg);
    *thread = (pthread_t)(next_id++ << 12);   /* non-zero, never dereferenced: join below ignores it */
    return 0;
}
int pthread_join(pthread_t thread, void** ret) {
    (void)thread;
    if (ret) *ret = 0;
    return 0;
}
int pthread_detach(pthread_t thread) { (void)thread; return 0; }
