/* localhgt_hip.h -- C-ABI of the MI355X k-mer sketch->peak engine (liblocalhgt_hip.so).
 *
 * The reference has no in-process FFI for this path: its boundary is the `extract_ref`
 * process (scripts/pipeline.sh:35; argv at src/extract_ref_normal_peak.cpp:1352-1364).
 * This header is the boundary a replacement binds instead; each entry point names the
 * reference code it replaces ("E" = /root/reference/src/extract_ref_normal_peak.cpp).
 * Plain C types only, opaque context, every call returns an int status (0 = ok) and never
 * aborts; lhgt_last_error() gives the text of the last failure on the calling thread.
 * One context per process and GPU; calls on one context must not overlap.
 */
#ifndef LOCALHGT_HIP_H
#define LOCALHGT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LHGT_ABI_VERSION 1
#define LHGT_CODER_SLOTS 300      /* E:21,1186: choose_coder[300] */
#define LHGT_MAX_READ_LEN 500     /* E:1004-1005: int reads_int[500] */
#define LHGT_MAX_RANDOM 50000000L /* E:40 */

enum {
    LHGT_OK = 0,
    LHGT_E_ARG = 1,        /* bad argument */
    LHGT_E_IO = 2,         /* file open/read/write */
    LHGT_E_HIP = 3,        /* HIP runtime error (message has the HIP string) */
    LHGT_E_FORMAT = 4,     /* malformed FASTA/FASTQ/index */
    LHGT_E_STATE = 5,      /* call order violated (e.g. ref_scan before index_load) */
    LHGT_E_TOO_MANY_PEAKS = 6, /* E:272-274 "Too many peaks!" (the reference overruns its arrays) */
    LHGT_E_NOMEM = 7,
    LHGT_E_NO_DEVICE = 8,
    LHGT_E_EMULATION = 9   /* only the -t N emulation refuses this input (the reference's -t N run is undefined on it); the
                              -t 1 result is defined: `extract_ref` falls back to it with a warning */
};

typedef struct lhgt_ctx lhgt_ctx;

int lhgt_abi_version(void);
const char* lhgt_last_error(void);
int lhgt_device_count(int* n);

/* ---- context: k-mer length k (<= 32), e hash functions (1..9).  Allocates the 2-bit count
 * table of 2^k slots in HBM (replaces `new char[2^k]` + memset, E:1375-1376,1416). */
int lhgt_ctx_create(int device, int k, int e, lhgt_ctx** out);
int lhgt_ctx_destroy(lhgt_ctx* ctx);
/* lhgt_ctx_destroy keeps one peak_kmer table of 4 GiB or more (16 GiB at k = 32) per device and size for the next context of the
 * process -- a process that handles sample after sample would free and allocate it every time, and that allocation was measured
 * to take anything from 0 to 2 s.  lhgt_pool_trim frees what is kept. */
int lhgt_pool_trim(void);

/* ---- R: glibc rand() stream and the position coder (host side, private random_r state) */
int lhgt_rng_seed(lhgt_ctx* ctx, unsigned seed);                       /* srand(seed), E:1386 */
int lhgt_coder_generate(lhgt_ctx* ctx);                                /* random_coder, E:1182-1222 */
int lhgt_coder_set(lhgt_ctx* ctx, const int16_t* cc /*[300]*/);        /* saved_random_coder, E:1224-1242 */
int lhgt_coder_get(lhgt_ctx* ctx, int16_t* cc /*[300]*/);
int lhgt_sampling_init(lhgt_ctx* ctx, double ratio_percent);           /* get_random, E:1332-1340; no draw when ratio >= 100 */
int lhgt_sampling_get(lhgt_ctx* ctx, float* out, long n);              /* first n values of random_array (tests) */
/* read n looks at random_array[n % 5*10^7] (E:1037-1044) and nothing draws from the stream after get_random: a run that knows it
 * will see at most n_reads reads per file (from the line count) needs only that many entries; the loaders refuse a read beyond
 * them.  0 (default) = all 5*10^7.  Call before lhgt_sampling_init. */
int lhgt_sampling_reserve(lhgt_ctx* ctx, long n_reads);
/* get_random started early, on a host thread of its own: the draws depend on nothing but the seed and the coder's draws before
 * them (E:1386-1422), so they can run next to the line count of the FASTQ files and the reference load.  Assumes ratio < 100;
 * lhgt_sampling_init(ratio) joins the fill (lhgt_sampling_reserve before it cuts it short) and drops the array when ratio >= 100,
 * where nothing observes the draws.  Same array as the synchronous lhgt_sampling_init. */
int lhgt_sampling_begin(lhgt_ctx* ctx);

/* ---- H: hash of every k-mer of one sequence (parity probe for E:1052-1081 / 786-811).
 * out_hash[(j*e)+i], out_valid[j]; runs the same device code as every phase. */
int lhgt_hash_sequence(lhgt_ctx* ctx, const uint8_t* ascii, long len, uint32_t* out_hash, uint8_t* out_valid);

/* ---- I: index build / load (read_ref E:727-886, read side E:888-945).  File formats are the
 * reference's: <ref>.k<k>.h<e>.index.dat and <ref>.genome.len.txt. */
int lhgt_index_build(lhgt_ctx* ctx, const char* fasta_path, const char* index_path, const char* genome_len_path,
                     long* n_contigs, long* n_bases);
int lhgt_index_load(lhgt_ctx* ctx, const char* index_path, long* n_contigs, long* n_bases);
/* only shard `rank` of `world` becomes resident: contiguous contig groups of about equal index bytes (the role of
 * split_ref, E:1280-1330); contig numbers stay global.  For the reference-sharded phase B below. */
int lhgt_index_load_shard(lhgt_ctx* ctx, const char* index_path, int shard_rank, int shard_world, long* n_contigs, long* n_bases);
/* resident index straight from sequences already in host memory (bench / tests): contig c is
 * ascii[off[c] .. off[c+1]) */
int lhgt_index_from_memory(lhgt_ctx* ctx, const uint8_t* ascii, const uint64_t* off, long n_contigs);
/* ---- The resident form of the reference.  form 0 (default): the index file's layout, e stored hashes
 * per position (4e bytes per base: 156 GB for a 13 Gbase catalogue), read by phase B as read_index does (E:933-945).  form 1:
 * the BASES, as three bit-planes over the indexed contigs (3/8 byte per base: 4.9 GB), and phase B recomputes the hashes read_ref
 * would have stored (E:786-813; an invalid k-mer is hash 0, E:808-810) -- same flags, peaks, ids and votes, no index file read.
 * The packed form is filled from bases only: lhgt_reference_load_fasta, lhgt_index_from_memory, lhgt_synth_reference*;
 * lhgt_index_load refuses it.  Changing the form drops a resident reference of the other form. */
int lhgt_set_reference_form(lhgt_ctx* ctx, int form);
int lhgt_reference_info(lhgt_ctx* ctx, int* form, unsigned long long* resident_bytes);
/* the coder of an existing index file (saved_random_coder E:1224-1242) without its hashes */
int lhgt_index_read_coder(lhgt_ctx* ctx, const char* index_path);
/* read_ref (E:727-886) + read_index (E:888-979) without the file in between: the FASTA's contigs longer than k become resident
 * in the context's form, numbered as the index numbers them; genome_len_path (nullable) also writes genome.len.txt (E:773, 878) */
int lhgt_reference_load_fasta(lhgt_ctx* ctx, const char* fasta_path, const char* genome_len_path, long* n_contigs, long* n_bases);
/* host-only: the FASTA's line structure as the two loaders above see it (no GPU, no bases touched): '>' lines found with memchr,
 * newlines counted per 4 KiB block on all cores, sequence lengths from those counts -- std::getline semantics of read_ref
 * (E:761-880).  Writes genome.len.txt (nullable) as read_ref does (E:773, 878) for the sequences longer than k and returns their
 * number and bases; n_sequences = every sequence of the file, the one before the first '>' line included. */
int lhgt_fasta_scan(const char* fasta_path, int k, const char* genome_len_path, long* n_sequences, long* n_contigs, long* n_bases);

/* ---- reads: sampling ratio (cal_sam_ratio E:1244-1270) and the resident pair store */
int lhgt_fastq_sam_ratio(const char* fq1, double sample, double* ratio_percent, long* n_records);
/* Parse both FASTQs in lock-step (E:350-359), keep pair n iff random_array[n % 5e7] < ratio
 * (E:413-419, 1037-1044), mark mate 2 as not-counted once its byte cursor passed size(fq1)
 * (E:1419-1445), keep pairs of block (n / shard_block) % shard_world == shard_rank, upload
 * and 2-bit pack them.  The store stays resident for phases A and C.  Parsing is multi-threaded (LHGT_INGEST_THREADS,
 * default min(32, cores)): the reference's pairing is purely line-indexed, so any line start is a split point.
 * When the first read IDs of the two files differ, phase C pairs fq1's line g with fq2's line g + s, s = the first line of fq2
 * (read from byte 1) that carries fq1's first ID (E:368-402); fq2's records in front of it are counted by phase A only, each file
 * sampled by its own read ordinal.  When fq2 runs out first, the remaining reads of fq1 are voted against what std::getline leaves
 * behind (E:356-367): an empty mate 2, or fq2's last line when that has no newline.  Refused: no line with fq1's first ID (the
 * reference spins through 10^9 failed reads), s inside a record, reads longer than LHGT_MAX_READ_LEN. */
int lhgt_pairs_load_fastq(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent,
                          int shard_rank, int shard_world, long shard_block, long* n_pairs_seen, long* n_pairs_kept);
/* Multi-GPU ingest (the reference's threads split the files by byte ranges, E:1426-1434, and lose the records at
 * the boundaries -- here global line numbers make every split exact).  A file is cut into chunks of chunk_bytes
 * (lhgt_fastq_plan_chunk_bytes(): what the loader uses), chunk c starting at the first line start at or after c * chunk_bytes.
 * lhgt_fastq_plan_part counts the lines of part `part` of `n_parts` of the chunks (start[i], n_lines[i] for its n_out chunks;
 * start == NULL: only n_out and n_chunks_total; len_sums: also the summed line lengths of every chunk by line index inside the
 * chunk mod 4, from which cal_sam_ratio's base count (E:1244-1270) follows without its extra pass).  The host layer all-gathers the pieces of both files, in part order, and
 * lhgt_pairs_load_fastq_planned parses fq1's chunks [part * n / n_parts, (part + 1) * n / n_parts) against the same lines of fq2:
 * a contiguous run of pairs per rank, kept by the GLOBAL read ordinal (E:1037-1044), each rank touching 1/n_parts of the text.
 * Quirk Q4, surplus fq2 records (last part), fq2 re-synchronisation (first part) and -t N emulation as in lhgt_pairs_load_fastq. */
long lhgt_fastq_plan_chunk_bytes(void);
int lhgt_fastq_plan_part(const char* fq, long chunk_bytes, int part, int n_parts, uint64_t* start, long* n_lines, long cap, long* n_out,
                         long* n_chunks_total, long* len_sums_or_null /*[4 * n_out]*/);
int lhgt_pairs_load_fastq_planned(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent, const uint64_t* start1,
                                  const long* n_lines1, long n1, const uint64_t* start2, const long* n_lines2, long n2, int part,
                                  int n_parts, long* n_pairs_seen, long* n_pairs_kept);
/* host-only probe of the planned parse (tests): chain != 0 continues the digest from *digest, so the parts of a split run in
 * order give the digest of the unsplit parse; start1 == NULL: plans made inside, as lhgt_fastq_parse_digest_threads */
int lhgt_fastq_parse_digest_planned(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int shard_rank,
                                    int shard_world, long shard_block, int threads, long chunk_bytes, int emulate_threads,
                                    const uint64_t* start1, const long* n_lines1, long n1, const uint64_t* start2, const long* n_lines2, long n2,
                                    int part, int n_parts, int chain, long* n_pairs_seen, long* n_pairs_kept, uint64_t* digest,
                                    long* counts_or_null);
/* The reference's -t N as it runs WITHOUT its races (contract = the reference binary with its threads run
 * one after the other in creation order).  threads > 1 makes the calls below restate: the per-thread byte chunks of the
 * FASTQs (get_fq_start E:44-89; lines consumed while the cursor before the line is <= the chunk end, E:1019-1026; a record cut
 * by a boundary is lost as there), sampling ordinals counted per chunk (E:1037) -- lhgt_pairs_load_fastq; the contig groups of
 * split_ref with peak ids from j * (max_peak / threads) (E:1280-1330, 229-237) -- lhgt_ref_scan; one sentinel line per thread
 * (E:515-548) -- lhgt_write_intervals.  Default 1: every read, one id range (the -t 1 result whatever -t says).
 * Peak ids stay dense (the tile scan's sequential ones, + 1 when thread 0 finds no peak, because only its first peak can hold
 * the invisible id 0): nothing observable depends on the bases j * (max_peak / N), only on the order of the ids, on which peaks
 * share a thread's range and on the range's capacity.  With a reference shard resident (lhgt_index_load_shard) the groups cut
 * across the ranks: lhgt_ref_scan_local, lhgt_ref_scan_group_counts, [sum over ranks], lhgt_set_group_totals, lhgt_ref_scan_emit,
 * lhgt_peaks_install.  Inputs on which the reference's behaviour is undefined (a chunk entered within 1000 bytes of EOF,
 * overlapping chunks, a thread's peaks overflowing its id range, threads that re-synchronise fq2 at different line offsets) are
 * refused with LHGT_E_EMULATION (round 6: a code of its own; the text starts with "-t N emulation" / "Too many peaks! thread"):
 * `extract_ref` then falls back to the -t 1 result with a warning. */
int lhgt_set_thread_emulation(lhgt_ctx* ctx, int threads);
/* where thread i of `threads` enters a FASTQ (byte), the global index of its first line and the lines it consumes;
 * size_for_chunks = size of fq1 (also for fq2, E:1419), < 0 = this file's */
int lhgt_fastq_thread_chunks(const char* fq, long size_for_chunks, int threads, long* entry_byte, long* first_line, long* n_lines);
/* the chunk sizes at which the two files of a pair should be planned (lhgt_fastq_plan_part) so that lhgt_pairs_load_fastq_planned can
 * read its part like the single-pass loader does (host_fastq_stream.cpp: fq2 cut into as many chunks as fq1); plans made at any
 * other size give the same pairs through the line-by-line chunk loop */
int lhgt_fastq_pair_chunk_bytes(const char* fq1, const char* fq2, long* chunk1, long* chunk2);
/* host-only rate probe of the FASTQ loader (tools/ingest_scaling.py): the parse as lhgt_pairs_load_fastq runs it -- worker threads
 * filling slabs with kept bases and per-pair records -- into host memory, nothing uploaded.  start1 null: the whole files (single
 * pass first, as in the loader); otherwise part `part` of `n_parts` of the planned parse (plans from lhgt_fastq_plan_part). */
int lhgt_fastq_parse_rate(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int threads, long chunk_bytes,
                          int emulate_threads, const uint64_t* start1, const long* n_lines1, long n1, const uint64_t* start2,
                          const long* n_lines2, long n2, int part, int n_parts, long* n_pairs_seen, long* n_pairs_kept, long* n_bases,
                          double* seconds, uint64_t* digest_or_null /* lhgt_fastq_parse_digest's digest, read back from the slabs */);
/* which way the calling thread's last FASTQ parse went: 1 = in one pass over the text (host_fastq_stream.cpp), 0 = line count, then
 * parse (the planned loader: always with plans made elsewhere, or when the single pass met something only the planned loader decides
 * -- `why`, nullable, says what), -1 = none yet.  The pairs are the same either way. */
int lhgt_ingest_last_path(char* why, long cap);
/* get_fq_start (E:44-89) on text in memory: the byte at which a thread whose chunk starts at `start` enters the file, -1 where
 * the reference's stream would hit EOF while looking (host only; the property test's handle on the restated scan) */
long lhgt_fastq_thread_entry(const uint8_t* text, long n, long start);
int lhgt_fastq_parse_digest_threads(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null,
                                    int shard_rank, int shard_world, long shard_block, int threads, long chunk_bytes,
                                    int emulate_threads, long* n_pairs_seen, long* n_pairs_kept, uint64_t* digest,
                                    long* counts_or_null /*[3]: mate 1 counted, mate 2 counted, voted*/);

/* host-only probe of the same parser (tests): FNV-1a digest of every kept pair in order; threads/chunk_bytes explicit */
int lhgt_fastq_parse_digest(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null,
                            int shard_rank, int shard_world, long shard_block, int threads, long chunk_bytes,
                            long* n_pairs_seen, long* n_pairs_kept, uint64_t* digest);
/* Append pairs from host memory: mate m of pair p is seq_m[off_m[p] .. off_m[p+1]).
 * count_mate2 (optional, one byte per pair) = 0 excludes mate 2 from phase A only. */
int lhgt_pairs_append(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2,
                      const uint64_t* off2, long n_pairs, const uint8_t* count_mate2);
/* The same with one flag byte per pair: bit 0 / bit 1 = mate 1 / mate 2 is counted in phase A, bit 2 = the pair is re-scanned
 * and voted in phase C (NULL = 7 for every pair).  What a reference thread chunk boundary loses (E:1022-1026) or a longer
 * fq2 adds (E:1438-1445) is expressed this way by lhgt_pairs_load_fastq. */
int lhgt_pairs_append_flags(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2,
                            const uint64_t* off2, long n_pairs, const uint8_t* pair_flags);
/* ---- a sample kept PACKED on disk (round 6; no reference counterpart: the reference reads FASTQ text, E:1020-1044, 350-419).
 * `localhgt_pack fq1 fq2 out.lhgp` loads a record-aligned pair of files whole (every read kept, no thread emulation) and writes the
 * resident store's own records at a fixed stride -- [u16 len1][u16 len2][mate 1: hi, lo, not-a-base planes of len1 / 32 + 1 words]
 * [mate 2 likewise], 148 bytes for 150-base pairs against ~640 of text -- behind a header that holds what the loader decides from
 * the TEXT: the lines each thread of the reference's -t N consumes for every N (lhgt_fastq_thread_chunks), the first pair whose mate 2
 * lies behind size(fq1) (quirk Q4), the bases of fq1 (cal_sam_ratio).  lhgt_pairs_load_packed reads part `part` of `n_parts` of the
 * records into pinned memory (pread on all host threads, nothing parsed), and the GPU decides which pairs the run keeps -- the sampling
 * array by global ordinal and Q4 at threads = 1, the thread chunks otherwise: the rules of lhgt_pairs_load_fastq -- and lays them out
 * as batches.  Same resident pairs (in another order: phases A and C do not depend on it), same tables, same files.
 * lhgt_pairs_batches / _batch_info: the resident store as it stands; lhgt_pairs_store_write: its records to `path` from data_offset on
 * (refused unless the store is that of a clean pair of files read whole). */
int lhgt_pairs_batches(lhgt_ctx* ctx, long* n_batches);
/* the same file -- the same bytes -- without a GPU in the machine (round 6, late): the loader's own host parse of the two files (every read
 * kept, no thread emulation: the loops of read_fastq E:1020-1044 / slide_reads E:350-419 as lhgt_pairs_load_fastq restates them), run twice --
 * for the longest read, which decides the stride (returned), and for the records, 2-bit packed by host threads.  Refuses what
 * lhgt_pairs_store_write refuses.  threads < 1: the CPUs the process may use. */
int lhgt_fastq_pack_host(const char* fq1, const char* fq2, const char* out_path, unsigned long long data_offset, int threads, long* stride,
                         long* n_pairs, long* q4_first_pair, unsigned long long* bases1, int* max_len);
int lhgt_pairs_batch_info(lhgt_ctx* ctx, long batch, long* n_pairs, unsigned long long* n_words, int* max_len);
int lhgt_pairs_store_write(lhgt_ctx* ctx, const char* path, unsigned long long data_offset, long stride, long* n_pairs, long* q4_first_pair,
                           unsigned long long* bases1);
int lhgt_pairs_load_packed(lhgt_ctx* ctx, const char* path, unsigned long long data_offset, long stride, long n_pairs_total, long q4_first_pair,
                           double ratio_percent, int threads, const long* first1, const long* count1, const long* first2, const long* count2,
                           int part, int n_parts, long* n_pairs_seen, long* n_pairs_kept);
/* measurement handle (tools/ingest_scaling.py), no GPU: the host side of lhgt_pairs_load_packed alone -- the records of part `part`
 * of `n_parts` read in the loader's chunks into two host buffers by `threads` threads (0: the loader's own choice) */
int lhgt_packed_read_rate(const char* path, unsigned long long data_offset, long stride, long n_pairs_total, int part, int n_parts, int threads,
                          double* seconds);
/* with count-on-load lhgt_pairs_load_fastq closes a batch every Mi pairs and runs phase A on it at once, behind the parsing of the
 * next batch (the coder must be set: load or build the index first); lhgt_count_kmers then only counts what is not counted yet and
 * reports the whole kernel time.  lhgt_counts_clear makes every batch uncounted again. */
int lhgt_set_count_on_load(lhgt_ctx* ctx, int on);
int lhgt_pairs_clear(lhgt_ctx* ctx);
int lhgt_pairs_count(lhgt_ctx* ctx, long* n_pairs);

/* ---- A: saturating k-mer count of every resident read into the 2-bit table (read_fastq E:981-1107).
 * Result per slot = min(3, occurrences): order independent, equal to the -t 1 reference. */
int lhgt_count_kmers(lhgt_ctx* ctx);
/* two implementations of the same result: 1 = radix partition + LDS apply (k_count_part.hip), 0 = one device
 * compare-and-swap per hash behind a pre-check load (k_count.hip); -1 (default) picks by k: partition from k >= 26. */
int lhgt_set_count_mode(lhgt_ctx* ctx, int mode);
int lhgt_counts_clear(lhgt_ctx* ctx);

/* ---- count_diff_kmer.cpp ("C"), the stand-alone phase-A tool, as it behaves with time() fixed and its 10 threads in creation
 * order (--compat of bin/count_diff_kmer).  Its coder is a `bool` array: a non-ACGT base codes 1 in every projection on both
 * strands and voids nothing (C:155-160, 124) -- lhgt_set_count_compat; one rand() % 6 per k-mer offset (C:226-232) --
 * lhgt_coder_generate_count_diff; reads come in 10 byte chunks entered at the previous '@', tokenised by `>>`, with a byte
 * budget that overruns into the next chunk (C:53-153) and per-chunk sampling from srand(seed) -- lhgt_reads_load_count_diff
 * (kept reads are appended as mate-1-only entries; call once per file, size_for_chunks = size of fq1 for both). */
int lhgt_set_count_compat(lhgt_ctx* ctx, int on);
int lhgt_coder_generate_count_diff(lhgt_ctx* ctx);
int lhgt_reads_load_count_diff(lhgt_ctx* ctx, const char* fq, long size_for_chunks, int ratio_percent, unsigned seed, long* n_reads_kept);

/* ---- multi-GPU plumbing (no reference counterpart).  Device pointers are handed
 * to the host layer (torch.distributed/RCCL); merge = per-slot saturating add of 2-bit fields. */
int lhgt_counts_buffer(lhgt_ctx* ctx, void** dev_ptr, size_t* bytes);
int lhgt_counts_merge(lhgt_ctx* ctx, const void* dev_other, size_t byte_offset, size_t bytes);
int lhgt_filter_buffer(lhgt_ctx* ctx, void** dev_ptr, size_t* bytes); /* u32 votes per peak */

/* ---- B: reference scan + peak registry (read_index E:888-979, slide_window E:550-725,
 * add_peak/merge_peak E:239-301).  hit_ratio/match_ratio are the float32 values of E:1368-1369. */
int lhgt_ref_scan(lhgt_ctx* ctx, float hit_ratio, float match_ratio, long max_peak, long* n_peaks);

/* ---- B, reference-sharded (an index larger than one GPU): each rank scans its contig shard,
 * ranks exchange new-peak counts (id_base = peaks of all lower ranks, contig order = rank order), then all-gather
 * the peak loci and the (hash, id) registrations that lhgt_peaks_install replays into every rank's peak_kmer. */
int lhgt_ref_scan_local(lhgt_ctx* ctx, float hit_ratio, float match_ratio, long* n_new_local, long* n_selected_local);
/* under -t N emulation: new peaks of this rank's contigs per split_ref group (counts[threads]); the totals over all ranks fix the
 * id ranges and *first_id (0, or 1 when thread 0 found no peak), which every rank adds to its id base */
int lhgt_ref_scan_group_counts(lhgt_ctx* ctx, long* counts, int threads);
int lhgt_set_group_totals(lhgt_ctx* ctx, const long* totals, int threads, long max_peak, long* first_id);
int lhgt_ref_scan_emit(lhgt_ctx* ctx, long id_base, void** dev_loci /* int32[2*n_new_local] */,
                       void** dev_regs /* uint32[2*n_regs]: hash, id */, long* n_regs);
int lhgt_peaks_install(lhgt_ctx* ctx, long n_peaks_total, long n_selected_total, long max_peak, const void* dev_loci_all,
                       const void* dev_regs_all, long n_regs_all);

/* ---- C: read re-scan + split-read vote (slide_reads E:313-506, Split_reads E:91-202) */
int lhgt_vote(lhgt_ctx* ctx);

/* ---- D: interval file (count_filtered_peak E:515-548) */
int lhgt_write_intervals(lhgt_ctx* ctx, const char* path, long* n_filtered);

/* ---- next row, SURVEY.md 8(f) rank 3: BED -> extracted FASTA without samtools (host only, no context needed).
 *      lhgt_faidx_extract replaces `samtools faidx -r ${interval_file}.bed $original_ref > $extracted_ref` (scripts/pipeline.sh:37):
 *      one ">NAME:BEG-END" record per region line, bases as stored, line_width (<= 0: 60) per line, END truncated at the contig's
 *      length; out_path "-" = stdout.  lhgt_faidx_build replaces `samtools faidx $ref` (scripts/infer_HGT_breakpoint.py:156) and
 *      writes the five-column .fai (fai_path may be NULL to validate only).  samtools is absent here: parity unpinned. */
int lhgt_faidx_build(const char* fasta_path, const char* fai_path, long* n_sequences);
int lhgt_faidx_extract(const char* fasta_path, const char* regions_path, const char* out_path, int line_width, long* n_regions,
                       long* n_bases);

/* ---- introspection for parity tests (device -> host copies) */
int lhgt_counts_export_u8(lhgt_ctx* ctx, uint64_t first_slot, uint64_t n_slots, uint8_t* out);
int lhgt_counts_histogram(lhgt_ctx* ctx, uint64_t out[4]);            /* cal_tab_empty_rate, count_diff_kmer.cpp:26-50 */
int lhgt_flags_export(lhgt_ctx* ctx, uint64_t first_pos, uint64_t n_pos, uint8_t* out);
int lhgt_peaks_export(lhgt_ctx* ctx, int32_t* loci /*[2*n]*/, uint8_t* filter /*[n]*/, long n);
int lhgt_peak_kmer_export(lhgt_ctx* ctx, uint64_t first_slot, uint64_t n_slots, uint32_t* out);
/* test handle: the 1024 + 1 group bounds of the dense vote (k_vote.hip: first peak id of runs of whole contigs with about equal shares of
 * the peaks, [0] = 0, [1024] = all ones), as the last lhgt_ref_scan left them; *valid = 0 if it left none (a vote bitmap is kept, or the
 * registry was installed from records) */
int lhgt_vote_groups_export(lhgt_ctx* ctx, uint32_t* out, int n, int* valid);

/* position-sensitive checksum of a whole device table (parity at sizes whose tables cannot travel through the host):
 * out[0] = sum_i mix(i, v[i] & mask) mod 2^64, out[1] = number of i with v[i] & mask != 0.
 * what: 0 = count table (packed words), 1 = flags per reference position, 2 = peak_kmer, 3 = peak_loci, 4 = votes (u32) */
int lhgt_digest(lhgt_ctx* ctx, int what, uint64_t mask, uint64_t out[2]);

/* ---- synthetic workload generated on the device (bench.py / tests; no reference counterpart).
 * Bases are a pure function of (seed, contig, position); see localhgt_amd/csrc/k_synth.hip. */
int lhgt_synth_reference(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, uint8_t* host_ascii_or_null);
int lhgt_synth_reference_shard(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, int shard_rank, int shard_world,
                               uint8_t* host_ascii_or_null);
/* the same base stream cut into contigs at cuts[0] = 0 < ... < cuts[n_cuts-1] = n_contigs*contig_len (pieces of <= k bases are not
 * indexed, E:772): a reference with a ragged length distribution under the same synthetic reads */
int lhgt_synth_reference_cuts(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, const uint64_t* cuts, long n_cuts,
                              uint8_t* host_ascii_or_null);
int lhgt_synth_pairs(lhgt_ctx* ctx, uint64_t ref_seed, uint64_t reads_seed, long n_contigs, long contig_len,
                     long first_pair, long n_pairs, int read_len, uint8_t* host_seq1_or_null, uint8_t* host_seq2_or_null);

/* knobs of the synthetic sample: positions per thousand at which a sample genome differs from the reference (default 0; 10 =
 * the "snp0.01" of the reference's test data), reads per thousand carrying one N (default 20), number of contigs the sample is
 * drawn from (0 = half of the reference; a metagenome holds far fewer of a catalogue's genomes). */
int lhgt_synth_options(lhgt_ctx* ctx, int snp_permille, int n_permille, long sample_contigs);
/* a share of the synthetic pairs (per thousand) gets reads of long_len bases instead of the read length lhgt_synth_pairs is
 * called with: mixed batches, as a sample with reads of several lengths makes them (0 = none) */
int lhgt_synth_read_mix(lhgt_ctx* ctx, int long_permille, int long_len);

/* switches for profiling / A-B runs.  bit0: lhgt_vote skips judge_base (outputs wrong);
 * bit2: never use the vote prefilter; bit4: without its LDS-resident first level (the fold); bit5: generic vote kernel even on the
 * sparse path; bit6: ref_flags never uses the saturated-line summary; bit7: chunked tile scan at any size; bit8: no tile is
 * settled by window_good alone (bits 2-8: outputs unchanged); bit9 / bit10: the sparse vote kernel stops after its first /
 * second filter level (stage timing, outputs wrong); bit11: the queued sparse vote kernel votes every pair with more than 8 bitmap
 * survivors directly (exercises that branch; outputs unchanged); bit12 / bit13 / bit14: lhgt_ref_scan takes the single-first (lite) / the exact /
 * the trio-first form of its first two steps whatever the table looks like (default for e <= 3: trio-first below 45 % of the slots at 3, lite from 90 %,
 * in between exact unless a trial on a few runs of tiles favours lite; outputs unchanged); bit16: phase A's partition takes round 3's
 * sorted-tile scatters (histogram pass, shared regions, global run cursors) instead of round 4's direct form (k = 32, e = 3; outputs
 * unchanged); bit17 / bit18: the queued sparse vote kernel reads peak_kmer / the read records with plain instead of non-temporal
 * loads (outputs unchanged); bit19: the generic vote kernel walks every pair with six hit offsets, without the bound that proves most
 * pairs with long event lists unable to vote (k <= 23; outputs unchanged); bit20: the vote bitmap takes its three-quarter (3 MiB) form whatever
 * the number of registered k-mers (k > 25; outputs unchanged); bit21: phase A's direct form in its Small geometry (two 64 KiB
 * workgroups per CU; outputs unchanged); bit22: stage ablation of phase A's direct form, the stages named by LHGT_PART_ABLATE (timing
 * only, the table comes out WRONG); bit23: register_peaks looks every "count > 0" up in the table instead of taking what the
 * trio-first probe kernels recorded (outputs unchanged); bit24: lhgt_ref_scan takes the trio-first form answered from the slot
 * list, which it builds at once if there is none (lhgt_slot_list; outputs unchanged); bit25: a slot list that exists is not used;
 * bit26: stage ablation of the slot-first kernel, the stage named by LHGT_SLOTS_ABLATE (1: no listed position is followed, 2: none
 * probes the table; timing only, the flags come out WRONG); bit27: a dense peak set (no bitmap) is voted in the shared-line-fill form
 * (k_vote_shared.hip) whatever the store's size and grouping (e <= 3; outputs unchanged); bit28: never in that form; bit29: the peaks' k-mers are registered by partition (lhgt_registry_info) whatever their number (k >= 20), bit30: never.
 * The environment variable LHGT_DEBUG presets the flags of every new context. */
int lhgt_set_debug(lhgt_ctx* ctx, int flags);
/* the context's kernels run only on the CUs whose bits are set in mask[0 .. n_words) (n_words = 0: all CUs again): two contexts
 * with complementary masks share a GPU without sharing a CU (bench.py: pipelined_samples).  Nothing may be in flight. */
int lhgt_set_cu_mask(lhgt_ctx* ctx, const uint32_t* mask, int n_words);

/* ---- timing of the last call of each phase kernel group, HIP events on the ctx stream (ms) */
int lhgt_phase_ms(lhgt_ctx* ctx, int phase /*0=A 1=B 2=C (all kernels of the phase), 3 = the ref_flags kernel alone*/, float* ms);
/* ---- which form the last lhgt_ref_scan took (k_scan.hip): *lite = 0 exact (all e probes everywhere); 1 single-first, for a nearly
 *      full table (one probe per position until a hash reads 3, complete probes at every 8th position and in the tiles that cannot
 *      be settled from that); 2 trio-first, for a sparse table (probes until a hash does not read 3; complete probes only near
 *      windows that reach the trio threshold).  n_tiles_exact = tiles that got the exact treatment. */
int lhgt_scan_info(lhgt_ctx* ctx, int* lite, double* frac_slots_at_3, long* n_tiles, long* n_tiles_exact);
/* ---- how the last lhgt_ref_scan registered its peaks' k-mers in peak_kmer (add_peak, E:247-267; k_scan.hip): *chunks = 0 with one atomicMax
 *      per (slot, id) from the walk over the reference (register_peaks), n > 0 (round 6) routed by slot through two scatter passes and
 *      applied per 2^(k-17)-slot slice of the table in LDS, in n chunks of the reference (a dense peak set without a vote bitmap: half a G of
 *      records or more; LHGT_REGISTER_PART=0 never, =1 whenever k >= 20; debug bit 29 / 30 likewise).  records_bound = selected
 *      positions x e, records_direct = records that found their region full and went to the table at once.  Measurement only. */
int lhgt_registry_info(lhgt_ctx* ctx, int* chunks, unsigned long long* records_bound, unsigned long long* records_direct);
/* The slot list of the resident reference: every position with a k-mer, grouped by the top bits of the slot its largest hash
 * addresses (6 bytes per position; 10 when the list under the largest hash also carries every position's second-largest hash, which
 * it does when that leaves room on the device; + 7.5 % where the buckets' regions are sized from a sampled histogram -- every 8th tile,
 * LHGT_SLOT_LIST_SAMPLE, round 6: one pass of atomics over the reference instead of two, done over exactly if a bucket runs over).
 * A context that scans sample after sample against one resident reference answers the first
 * question of the sparse-table scan ("does this hash of the position read 3?", E:580) for all positions from a stream over that
 * list and each bucket's counters in LDS instead of one random probe per position; same flags where they are read, same peaks
 * and votes.  mode 0: never build one and drop the one there is; 1 (default; environment LHGT_SLOT_LIST): build it before the
 * SECOND sparse-form scan of the same resident reference when that pays -- 2^32 positions or more, and the last sparse scan sent
 * at most half of the tiles to the exact fill --, and use it under the same condition; 2: build it before the first such scan and
 * use it always; -1: leave the mode.  entries / bytes (nullable): the list as it stands (0 = none: not built yet, no memory for
 * it, e > 3, or positions beyond 2^34).  No reference counterpart (the reference walks its index file once per run, E:888-979). */
int lhgt_slot_list(lhgt_ctx* ctx, int mode, unsigned long long* entries, unsigned long long* bytes);
/* what the last build of a slot list cost: the time of its kernels (histogram, offsets, placing pass; HIP events, every attempt of the
 * build), without its allocations -- an 80-140 GB hipMalloc takes 0 or 3 seconds by what the process freed before (LHGT_TRACE prints the
 * wall time next to it); 0 if this context has built none.  Measurement only (bench.py: slot_list_build_ms, break_even_samples). */
int lhgt_slot_list_build_ms(lhgt_ctx* ctx, double* ms);
/* ---- which kernel the last lhgt_vote took (k_vote.hip): *form = 0 the generic kernel probing peak_kmer itself (dense peak sets), 1 the
 *      generic kernel behind the L2-resident bitmap, 2 the queued sparse kernel behind the bitmap, 3 the 128 KiB LDS fold in front of
 *      bitmap and peak_kmer, 4 (round 6) the shared-line-fill form of a dense peak set under a deep sample: reads grouped by their
 *      smallest hash, a workgroup fetches every DISTINCT slot its 32 reads probe once (k_vote_shared.hip; environment
 *      LHGT_SHARED_VOTE=0 never, =1 whenever e <= 3, default: when the grouping finds LHGT_SHARED_MIN = 3 reads per occupied bucket); *bitmap_bits = log2 of the bits the bitmap's mask spans (0: no bitmap), *three_quarter = 1 when only
 *      three quarters of them are used (3 MiB instead of 4).  Measurement only. */
int lhgt_vote_info(lhgt_ctx* ctx, int* form, int* bitmap_bits, int* three_quarter);
/* ---- work counters for the roofline's "bytes the implemented algorithm must move" (bench.py).  enable = 1: count from
 *      zero from now on; 0: stop; -1: leave as it is.  out (nullable, 8 values): [0] keys phase A's partition brought to its final
 *      buckets = valid k-mers x e of the counted mates, minus the keys that were applied to the table on the way because a tile row, a
 *      piece or a region was full (hot k-mers: none on uniform synthetic reads; so on hot inputs the roofline's needed bytes are
 *      slightly understated); the direct kernel of k < 26 reports the upper bound k-mer positions x e; [1] count-table probes of phase B's
 *      probe kernel in the last lhgt_ref_scan while counting was on (e per position with a k-mer in the exact form; the hashes
 *      ref_flags_lite / ref_flags_trio marked as probed, summed before the fill of the unsettled tiles; the slot-first form: the
 *      probes of the positions it followed beyond the slot list, their number in [2]); [3] probes that
 *      went on from the LDS fold to the L2 bitmap; [4] probes that went on from the bitmap to peak_kmer; [5] pairs voted in the
 *      lane-per-offset form behind the filters; [6] distinct peak_kmer slots the shared-line-fill vote fetched (its line fills) and [7] the probes it answered outside its LDS sets (one line fill each;
 *      [5] then counts the pairs whose events were walked).  No reference counterpart: measurement only. */
int lhgt_work_stats(lhgt_ctx* ctx, int enable, unsigned long long out[8]);
int lhgt_stream(lhgt_ctx* ctx, void** hip_stream);
int lhgt_synchronize(lhgt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
