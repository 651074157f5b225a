// lhgt_common.hpp -- context, error plumbing and device-side data layout shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <string>
#include <thread>
#include <vector>
#include "../../include/localhgt_hip.h"

namespace lhgt {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
const char* last_error();
void entry_context(::lhgt_ctx* ctx);   // cabi.hip: the context whose call runs on this thread (whose optional structures an allocation may drop)
#define LHGT_HIP(expr)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            lhgt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return LHGT_E_HIP;                                                              \
        }                                                                                   \
    } while (0)
#define LHGT_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != LHGT_OK) return rc_; \
    } while (0)
#define LHGT_FAIL(code, ...)      \
    do {                          \
        lhgt::set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)
// top of every entry point that touches the GPU: a host-only context fails loudly (there is no CPU fallback), and the calling
// thread is bound to the context's device, so lazily allocated buffers and launches land on the right GPU whatever the caller
// (or another engine in the same process) made current in between
#define LHGT_DEVICE_ENTRY(ctx)                                                                                        \
    do {                                                                                                              \
        if ((ctx) && (ctx)->device < 0)                                                                               \
            LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");                 \
        if (ctx) LHGT_HIP(hipSetDevice((ctx)->device));                                                               \
        lhgt::entry_context(ctx);                                                                                     \
    } while (0)

// ---------------------------------------------------------------- hash parameters (by value to kernels)
// mask[i][m] has bit (k-1-z) set iff choose_coder[z*e+i] == m  
struct HashParams {
    uint32_t mask[9][3];
    int k, e;
};

// ---------------------------------------------------------------- resident read store
// One batch = n_pairs pairs.  Read (m, p) owns 3*wpr words at words[off[m][p]]: the hi-bit
// plane, the lo-bit plane and the not-a-base plane of its 2-bit codes (A=0 C=1 G=2 T=3),
// 32 bases per word, first base at the MSB, wpr = ceil(len/32)+1 (one zero pad word so a
// k-bit window can always be cut from two consecutive words).
struct ReadBatchDev {
    const uint32_t* words;
    const uint32_t* off[2];
    const uint16_t* len[2];
    // nullable (= every bit set).  Per pair: bit 0 / bit 1 = mate 1 / mate 2 is counted in phase A, bit 2 = the pair is re-scanned
    // and voted in phase C.  Quirk Q4 (mate 2 past size(fq1) is not counted) clears bit 1; surplus records of a longer fq2 are
    // mate-2-only entries (bit 1 alone); the -t N emulation (host_threads.cpp) clears whatever a thread chunk boundary loses.
    const uint8_t* flags;
    long n_pairs;
};
constexpr uint8_t PAIR_COUNT1 = 1, PAIR_COUNT2 = 2, PAIR_VOTE = 4, PAIR_ALL = 7;
struct ReadBatch {
    ReadBatchDev d{};
    bool counted = false;   // the loader ran phase A on this batch (count-on-load): the next lhgt_count_kmers skips it once; reset by lhgt_counts_clear
    void* alloc[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t n_words = 0;
    int max_len = 0;
    uint64_t n_kmers = 0;  // valid-or-not k-mer positions of both mates (for rates)
    // reads with more than FAST_NK k-mer offsets (> 159 bases at k = 32).  The fast forms of phases A and C take reads up to that
    // length as they are; phase A's direct scatters take the longer ones cut into segments (k_count_part.hip), and the sparse votes,
    // when a batch has only a FEW of them, list their pairs for the generic kernel (k_vote.hip) -- round 4 sent the whole batch of
    // millions of pairs down the generic paths for one 250-base read.  -1 = not counted.
    long n_long = -1;
};
constexpr int FAST_NK = 128;

// ---------------------------------------------------------------- resident index
// d_index = the index file minus its 1200-byte header: per contig [u32 len][(len-k+1)*e u32].
struct ContigDev {
    uint64_t hash_word;  // word offset of the contig's first hash in d_index
    uint64_t flat_base;  // position of base 0 in the flat per-position arrays
    uint32_t len;
    uint32_t ref_index;  // 1,2,3.. in index order (E:905,963; quirk Q7)
};
constexpr int PF_BITS = 25;  // vote prefilter: 2^25 bits = 4 MiB, one XCD's L2
constexpr int FASTA_BLK = 4096;  // bytes of FASTA text per newline count on the host and per workgroup of strip_fasta_block
constexpr int VG_N = 1024;  // contig groups of the dense vote's bound (k_vote.hip): one 16-bit counter each
constexpr int TILE = 2000;  // positions per scan tile; multiple of 50 so peak buckets never straddle tiles
struct TileDev {
    uint32_t contig;
    uint32_t j0;
};
// One launch dimension holds at most 2^32 - 1 work-items (the dispatch packet's grid size is 32 bits wide): 16.7 M workgroups of
// 256, which one workgroup per 2000-position tile uses up at 33.5 Gbase.  Kernels launched per tile (or per entry of a tile
// list) therefore take a 2-D grid and the number of workgroups wanted; block2d() is the workgroup's linear number.
inline dim3 blocks2d(long n) {
    const long gx = 1L << 20;
    return n <= gx ? dim3((unsigned)(n > 0 ? n : 1)) : dim3((unsigned)gx, (unsigned)((n + gx - 1) / gx));
}
__device__ __forceinline__ long block2d() { return (long)blockIdx.y * gridDim.x + blockIdx.x; }

}  // namespace lhgt

struct lhgt_ctx {
    int device = 0, k = 0, e = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // non-blocking: the index upload, which may run next to the FASTQ loader on another host thread
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    float phase_ms[4] = {0, 0, 0, 0};   // A, B, C (all their kernels), and the ref_flags kernel alone
    // R
    char rng_state[128];
    void* rng = nullptr;  // struct random_data*
    bool seeded = false;
    int16_t cc[LHGT_CODER_SLOTS];
    bool have_coder = false;
    lhgt::HashParams hp{};
    std::vector<float> random_array;
    long sampling_reads = 0;     // lhgt_sampling_reserve: reads the run can look at (0 = unknown: all 5*10^7 entries are filled)
    long sampling_filled = 0;    // entries of random_array the last lhgt_sampling_init filled
    // lhgt_sampling_begin: the fill on a host thread of its own, next to the line count and the reference load; it stops at
    // fill_limit (lowered by lhgt_sampling_reserve / lhgt_sampling_init once the number of reads, or ratio >= 100, is known)
    std::thread* fill_thread = nullptr;
    std::atomic<long> fill_limit{0};
    long fill_done = 0;          // written by the fill thread, read after the join
    double ratio = 100.0;
    // A
    uint32_t* d_counts = nullptr;  // 2-bit saturating counters, 16 per word
    size_t counts_words = 0;
    bool counts_touched = false;   // something may have been counted since the last lhgt_counts_clear (a loader that has to take back a
                                   // count-on-load can only do so by clearing the table: allowed while this was false at its start)
    // I
    uint32_t* d_index = nullptr;
    size_t index_words = 0;
    // the packed form of the resident reference (lhgt_set_reference_form): three bit-planes over the flat positions instead of
    // the e stored hashes per position; phase B recomputes the hashes (lhgt_hash.hpp: RefSource)
    bool ref_packed = false;
    uint32_t* d_ref_planes = nullptr;
    size_t ref_plane_words = 0;
    std::vector<lhgt::ContigDev> contigs;
    lhgt::ContigDev* d_contigs = nullptr;
    lhgt::TileDev* d_tiles = nullptr;
    long n_tiles = 0;
    uint64_t n_pos = 0;
    bool index_resident = false;  // an index (possibly with zero contigs longer than k) has been installed
    uint8_t* d_flags = nullptr;
    uint32_t* d_satline = nullptr;   // one bit per 64-byte line of the count table: all 256 slots hold 3
    uint8_t* d_tile_good = nullptr;  // per tile: it contains a good window
    uint32_t* d_active_tiles = nullptr;  // tiles with a good window within reach (compacted per scan)
    uint8_t* d_nzmask = nullptr;  // per position: bit i = hash i has a non-zero count (E:250's `record_ref_hit > 0`)
    double scan_frac3 = 0.0;      // fraction of the table's slots holding 3 at the last scan
    long scan_n_need = 0;         // tiles the lite form had to treat exactly
    int scan_form = 0;            // 0 exact, 1 single-first (lite), 2 trio-first
    bool scan_slots = false;      // ... the trio-first form answered from the slot list (k_scan.hip: ref_flags_slots)
    // the slot list of the resident reference: every position with a k-mer, grouped by the top bits of its hash 0
    uint32_t* d_sl_lo = nullptr;             // low 32 bits of the flat position
    uint16_t* d_sl_hi = nullptr;             // low 14 bits of the slot | position bits 32-33 << 14
    uint32_t* d_sl_mid = nullptr;            // (list under the largest hash, when there is room) every entry's second-largest hash
    unsigned long long* d_sl_end = nullptr;  // [sl_buckets], or null: where a bucket's entries end when its region was sized from a sampled histogram (round 6)
    unsigned long long sl_capacity = 0;      // entries the list's arrays hold (= sl_entries for an exact histogram)
    unsigned long long* d_sl_off = nullptr;  // [sl_buckets + 1]
    unsigned long long sl_entries = 0;
    long sl_buckets = 0;
    double sl_build_ms = 0.0;                // kernel time of the last slot_list_build that built a list (events around its kernels, without the allocations)
    bool sl_in_use = false;                  // a scan is running on the list: an allocation out of memory must not drop it (cabi.hip: drop_optional)
    int sl_state = 0;                        // 0 not tried for this reference, 1 built, -1 tried and left (no memory, e > 3, positions beyond 2^34)
    int sl_mode = 1;                         // lhgt_slot_list / LHGT_SLOT_LIST: 0 never, 1 before the second sparse scan of a reference, 2 before the first
    int sl_sparse_scans = 0;                 // sparse-form scans of the resident reference so far
    bool sl_smallest = false;                // the list is kept under every position's smallest hash (for a saturated table), not its largest (a sparse one)
    uint32_t sl_unlisted = 0;                // k-mers all of whose hashes are 0 (packed form): no entry speaks for them
    double sl_need_share = 0.0;              // share of the tiles the last sparse-form scan sent to the fill
    bool scan_lite = false;       // the last scan took the lite form of B1/B2: d_nzmask then holds the per-hash probe state, not nz bits
    // reads
    std::vector<lhgt::ReadBatch> batches;
    long n_pairs = 0;
    unsigned long long store_gen = 0;   // counts every change of the resident read store (what k_vote_shared.hip keeps per store is stale after one)
    void* vshared = nullptr;            // lhgt_vshared* (k_vote_shared.hip): keys and order of the store's reads, arena
    // B/C
    uint32_t* d_peak_kmer = nullptr;
    int32_t* d_loci = nullptr;
    uint32_t* d_filter = nullptr;
    uint32_t* d_tile_count = nullptr;
    uint32_t* d_tile_sel = nullptr;     // selected positions per tile (interval_select): what the registry by partition sizes its chunks by
    uint8_t* d_rg_buf = nullptr;        // the registry by partition's record buffers + cursors (k_scan.hip: register_partitioned), kept between scans
    size_t rg_buf_bytes = 0;
    int rg_chunks = 0;                  // chunks of the last registry by partition (0: the direct kernel ran)
    unsigned long long rg_bound = 0, rg_direct = 0;   // its record bound, and the records that found their region full
    long n_peaks = -1, max_peak = 0;
    long id_end = 0;   // one past the largest peak id in use (= n_peaks, + 1 under -t N emulation when thread 0 found no peak: then no peak holds id 0)
    uint32_t* d_prefilter = nullptr;  // 2^PF_BITS-bit folded bitmap of slots holding a peak id (L2-resident), or unused
    uint32_t* d_prefilter_fold = nullptr;  // 64 or 128 KiB fold of it, copied into LDS by the fold vote kernels
    uint32_t* d_vote_groups = nullptr;     // [VG_N + 1] first peak id of every contig group of the dense vote's bound (k_vote.hip), made by lhgt_ref_scan (k_scan.hip)
    bool vote_groups_ok = false;           // ... and whether it describes the registered peaks (a plain lhgt_ref_scan: ids ascend with the tiles)
    uint32_t* d_revote = nullptr;          // vote_kernel_fold's deferred pairs: [0] = how many, then the pairs of the batch being voted
    size_t revote_cap = 0;                 // words
    bool prefilter_on = false;
    uint32_t pf_mask = 0;             // low address bits indexing the prefilter
    int vote_form = 0;                // which kernel the last lhgt_vote took: 0 generic on peak_kmer, 1 generic behind the bitmap, 2 queued behind the bitmap, 3 LDS fold + bitmap, 4 shared line fills (k_vote_shared.hip)
    bool pf_q3 = false;               // the bitmap is three quarters of what pf_mask spans (lhgt_hash.hpp: PF_Q3)
    int pf2 = 0;                      // folded prefilter: shift of the address bits picking a key's second bit (0 = one bit per key)
    unsigned long long n_selected = 0;  // peak positions inside good intervals (new + merged) of the last scan
    // reference-sharded scan (k_scan.hip): this rank's new peaks / registrations as records for the exchange
    int32_t* d_emit_loci = nullptr;
    uint32_t* d_emit_regs = nullptr;
    long emit_loci_cap = 0, emit_regs_cap = 0;
    long local_new = -1;       // new peaks found by the last lhgt_ref_scan_local
    long peaks_cap = 0;        // entries allocated in d_loci / d_filter (grow-only)
    void* d_voted = nullptr;   // phase D: compacted (id, contig, pos) of voted peaks + counter
    long voted_cap = 0;
    bool voted = false;
    // partitioned count (k_count_part.hip): two key buffers and the bucket histogram/offset block
    uint32_t* d_part_keys[2] = {nullptr, nullptr};
    size_t part_keys_cap = 0;  // keys per buffer
    long part_reserve_pairs = 0;          // a loader that counts batch by batch: the largest batch it will close (0 = no wish)
    uint32_t* d_part_meta = nullptr;
    int synth_snp_permille = 0, synth_n_permille = 20;   // k_synth.hip: lhgt_synth_options
    long synth_sample_contigs = 0;
    int synth_long_permille = 0, synth_long_len = 0;     // lhgt_synth_read_mix
    // the reference's -t N, race-free (lhgt_set_thread_emulation): read partition in host_fastx.cpp, contig groups with their own
    // id ranges in k_scan.hip, one sentinel line per thread in lhgt_write_intervals
    bool count_on_load = false;              // lhgt_set_count_on_load: the FASTQ loader counts every batch as soon as it is resident
    float count_on_load_ms = 0.f;            // kernel time of those counts (added to phase_ms[0] by lhgt_count_kmers)
    int emu_threads = 1;
    long emu_each_peaks = 0;                 // max_peak / N of the last scan
    std::vector<long> emu_range_end;         // per thread: one past its last peak id (dense ids; k_scan.hip: thread_id_ranges)
    bool emu_ranges_pending = false;         // lhgt_set_group_totals fixed the ranges for the next lhgt_peaks_install
    std::vector<uint32_t> all_lens;          // a reference SHARD is resident: the lengths of ALL indexed contigs (empty = the resident ones are all)
    std::vector<long> contig_first_tile;     // tile index of every resident contig's first tile
    bool count_compat = false;               // count_diff_kmer.cpp's bool coder (lhgt_set_count_compat)
    int count_mode = -1;       // -1 = by k (partition from k >= 26), 0 = direct CAS kernel, 1 = radix partition
    unsigned long long* d_digest = nullptr;   // 16 bytes: lhgt_digest's accumulator
    // lhgt_work_stats: while enabled, phase A adds up the keys it routes and the sparse vote kernels the probes that go on to the next
    // filter level (8 x u64 on the device; null = off, the kernels then issue no extra instruction but a scalar add per pair)
    unsigned long long* d_stats = nullptr;
    bool stats_on = false;
    bool stats_scan = false;   // the last lhgt_ref_scan ran while counting was on (lhgt_work_stats [1] of the exact form is computed on the host)
    unsigned long long stats_host[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // contributions known on the host (direct count kernel: upper bound of its keys)
    int debug = 0;             // ablation switches for profiling (bit0: vote skips judge_base); results are wrong when set
    // FASTQ loader (host_fastx.cpp): pinned slabs the parse threads write into (bases + per-pair records), pinned chunk descriptors of the open batch,
    // events that tell when a slab's copy has left the host
    void* ingest_pool = nullptr;             // lhgt::SlabPool*
    uint8_t* h_ingest_slabs = nullptr;       // hipHostMalloc: n_slabs x slab_bytes
    uint32_t* h_ingest_meta = nullptr;       // hipHostMalloc: the open batch's chunk descriptors (lhgt::ChunkDesc x ingest_meta_cap)
    long ingest_meta_cap = 0;
    std::vector<hipEvent_t> ingest_events;
    uint8_t* h_packed_stage = nullptr;       // hipHostMalloc: two chunks of packed records on their way to the device (k_packed.hip)
    size_t packed_stage_bytes = 0;
    uint32_t* d_ingest_start = nullptr;      // device copy of the start offsets of the batch being packed
    long ingest_start_cap = 0;
    // grow-only device workspaces (ASCII staging and packed planes of one contig / one upload)
    uint8_t* d_ws_ascii = nullptr;
    size_t ws_ascii_cap = 0;
    uint32_t* d_ws_words = nullptr;
    size_t ws_words_cap = 0;
};

int lhgt_count_batch_partitioned(lhgt_ctx* ctx, const lhgt::ReadBatch& b);
int lhgt_count_one_batch_async(lhgt_ctx* ctx, lhgt::ReadBatch& b, hipEvent_t t0, hipEvent_t t1);
void lhgt_ingest_pool_free(lhgt_ctx* ctx);   // host_fastx.cpp: the SlabPool object behind ctx->ingest_pool

namespace lhgt {
// host helpers implemented across the .cpp/.hip files
int build_hash_params(const int16_t* cc, int k, int e, HashParams* hp);
int rng_next(lhgt_ctx* ctx);  // one rand() draw from the private glibc stream
void sampling_join(lhgt_ctx* ctx);   // waits for a fill started by lhgt_sampling_begin (host_rng.cpp)
// Device memory through a process-wide cache of large blocks (cabi.hip).  The driver wipes freed VRAM before it hands it out
// again, so a hipMalloc shortly after a large hipFree waits for that -- measured at seconds: 16 GiB "in 0 s or 2.2 s"
// (profiles/r03/e2e_peak_kmer_alloc.txt), and in round 5 whichever allocation came first after a closed context's 10+ GB of
// batches, staging and key buffers (phase A on load 1.4-2.7 s instead of 85 ms, the scan 2 s instead of 8 ms, at random).  A
// process that runs sample after sample therefore keeps blocks of >= DEV_CACHE_MIN bytes when they are freed -- sizes rounded up
// to eighth-steps of a power of two (256 MiB at most) so that a slightly different request still fits -- and hands them to the next request of
// that class; lhgt_pool_trim frees them, and so does any allocation that would otherwise fail (a parked block must never be the
// reason a 156 GB index does not fit).
constexpr size_t DEV_CACHE_MIN = (size_t)64 << 20;
hipError_t dev_alloc_raw(void** p, size_t bytes);
size_t dev_cached_bytes();        // what the process keeps parked (reusable by the next dev_malloc of the same size class, released by an allocation that would fail)
hipError_t dev_free(void* p);          // blocks that came from dev_malloc go back to the cache; anything else to hipFree
bool big_release_all();               // frees every cached block; true if there was one
template <class T>
inline hipError_t dev_malloc(T** p, size_t bytes) { return dev_alloc_raw((void**)p, bytes); }
int upload_pairs(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2, const uint64_t* off2,
                 long n_pairs, const uint8_t* pair_flags);
int index_layout(lhgt_ctx* ctx, const std::vector<uint32_t>& lens, uint32_t first_ref_index = 1);  // allocates d_index, tiles, flags
int index_install(lhgt_ctx* ctx, const uint32_t* host_words, size_t n_words, bool words_on_device);
int index_install_shard(lhgt_ctx* ctx, const uint32_t* host_words, size_t n_words, int rank, int world);
int ws_reserve(lhgt_ctx* ctx, size_t ascii_bytes, size_t plane_words);
void slot_list_drop(lhgt_ctx* ctx);   // k_scan.hip: the resident reference changes
void vshared_free(lhgt_ctx* ctx);     // k_vote_shared.hip
int hash_contig_to_device(lhgt_ctx* ctx, const uint8_t* ascii, long len, uint32_t* d_out, uint8_t* d_valid);
int hash_contig_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long len, uint32_t* d_out, uint8_t* d_valid);
int hash_span_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long span_len, const uint64_t* coff, const uint64_t* out_word,
                        long n_c, uint32_t* d_out);
int write_index_lens(lhgt_ctx* ctx);
// one span of back-to-back sequences in device memory becomes resident in the context's reference form: hashed into d_index
// (hash_span_dev_ascii) or packed into the flat planes.  contig_of[c] = resident contig of span sequence c, or -1 (not indexed).
int install_span_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long span_len, const uint64_t* coff, const long* contig_of, long n_c);
int install_pairs_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, const uint64_t* start, const uint16_t* lens, long n,
                            const uint8_t* pair_flags);
// the loader's form: the batch as a list of chunks.  Per chunk the bases of its mate-1 reads and of its mate-2 reads lie back to
// back at b1 / b2 of d_ascii, and n + 1 records at d_meta + mo tell where every pair starts inside them (and inside the chunk's
// packed words); the device expands (descriptor + record) into the batch's start / length / word-offset / flag arrays
// (round 5: b2 = CHUNK_INTERLEAVED -- ONE run of bases at b1, the mates of a pair back to back: record i holds where mate 1 and mate 2
// of pair i start in it, record i + 1's rel1 is where pair i ends)
struct ChunkPairMeta { uint32_t rel1, rel2, relw, flags; };
struct ChunkDesc { uint32_t pair0, n, b1, b2, wbase, mo; };
constexpr uint32_t CHUNK_INTERLEAVED = 0xFFFFFFFFu;
int install_pairs_chunked(lhgt_ctx* ctx, const uint8_t* d_ascii, const ChunkPairMeta* d_meta, const ChunkDesc* desc, long n_desc, long n,
                          uint64_t n_words, int max_len, uint64_t n_kmers, long n_long);
int strip_fasta_text(lhgt_ctx* ctx, const uint8_t* d_text, uint64_t text_len, const uint64_t* kept_before, long n_blocks,
                     const uint64_t* seg, long n_seg, uint8_t* d_out);
void ingest_free(lhgt_ctx* ctx);
void pairs_truncate(lhgt_ctx* ctx, size_t n_batches);   // frees the resident batches from number n_batches on
int stage_ascii(lhgt_ctx* ctx, size_t dev_off, const uint8_t* src, size_t bytes);
int upload_locked_ahead(lhgt_ctx* ctx, hipStream_t st, void* d_dst, const void* src, size_t bytes);   // file mapping -> device, page-locking one piece ahead
}  // namespace lhgt
