// host_fastx.hpp -- what the host-side FASTA / FASTQ ingest shares between its translation units (host_fastx.cpp: the planned
// two-pass loader, the FASTA loaders, the C-ABI entry points; host_fastq_stream.cpp: the single-pass FASTQ loader).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <time.h>
#include <emmintrin.h>   // SSE2: part of the x86-64 baseline
#include "lhgt_common.hpp"

namespace lhgt {

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) {
            if (n > ((size_t)64 << 20)) {     // tearing down the page tables of a multi-GB mapping takes ~10 ms per GB: not on the caller's clock
                const uint8_t* q = p;
                const size_t m = n;
                std::thread([q, m] { munmap((void*)q, m); }).detach();
            } else munmap((void*)p, n);
        }
        if (fd >= 0) close(fd);
    }
    int open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) LHGT_FAIL(LHGT_E_IO, "cannot open %s", path);
        struct stat sb;
        if (fstat(fd, &sb)) LHGT_FAIL(LHGT_E_IO, "cannot stat %s", path);
        n = (size_t)sb.st_size;
        if (n) {
            void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) LHGT_FAIL(LHGT_E_IO, "cannot mmap %s", path);
            p = (const uint8_t*)m;
            const char* adv = getenv("LHGT_MMAP_ADVICE");     // experiment knob: seq (default) | none | willneed | hugepage
            if (!adv || !strcmp(adv, "seq")) madvise((void*)p, n, MADV_SEQUENTIAL);
            else if (!strcmp(adv, "willneed")) madvise((void*)p, n, MADV_WILLNEED);
            else if (!strcmp(adv, "hugepage")) madvise((void*)p, n, MADV_HUGEPAGE);
        }
        return LHGT_OK;
    }
};

// std::getline semantics: a trailing '\n' does not start another (empty) line
struct LineCursor {
    const uint8_t* p;
    size_t n, cur = 0;
    LineCursor(const Mapped& m) : p(m.p), n(m.n) {}
    bool next(const uint8_t** s, size_t* len, size_t* start) {
        if (cur >= n) return false;
        const uint8_t* st = p + cur;
        const uint8_t* nl = (const uint8_t*)memchr(st, '\n', n - cur);
        *s = st;
        *start = cur;
        if (nl) { *len = (size_t)(nl - st); cur += *len + 1; }
        else { *len = n - cur; cur = n; }
        return true;
    }
};

// get_read_ID (E:303-311): cut at the first '/', then at the first ' ', then at the first '\t'
inline size_t read_id_len(const uint8_t* s, size_t len) {
    size_t n = len;
    for (size_t i = 0; i < n; i++) if (s[i] == '/') { n = i; break; }
    for (size_t i = 0; i < n; i++) if (s[i] == ' ') { n = i; break; }
    for (size_t i = 0; i < n; i++) if (s[i] == '\t') { n = i; break; }
    return n;
}

inline double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
inline bool ingest_trace() { static int t = getenv("LHGT_INGEST_TRACE") ? 1 : 0; return t != 0; }

// A pool of equal slabs of pinned host memory (allocated by the GPU loader, absent for host-only callers).  A parse task
// writes the bases it keeps straight into a slab -- mate 1 into its first half, mate 2 into the second -- so the upload is one
// asynchronous copy per mate from page-locked memory (57 GB/s instead of 5.6 GB/s from pageable vectors, tools/h2d_rates.hip)
// with no staging copy and no per-chunk registration.
struct SlabPool {
    uint8_t* base = nullptr;
    size_t slab_bytes = 0, half_bytes = 0;   // a slab: [bases: 2 x half][ChunkPairMeta x (CHUNK_META_CAP + 1)]
    int k = 0;
    std::vector<int> free_ids;
    std::mutex mu;
    std::condition_variable cv;
    bool closed = false;
    void close() { { std::lock_guard<std::mutex> lk(mu); closed = true; } cv.notify_all(); }
    void reopen() { std::lock_guard<std::mutex> lk(mu); closed = false; }
    int acquire() {   // blocks until a slab is free; -1 once the pool is closed
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return closed || !free_ids.empty(); });
        if (closed) return -1;
        const int id = free_ids.back();
        free_ids.pop_back();
        return id;
    }
    void release(int id) {
        { std::lock_guard<std::mutex> lk(mu); free_ids.push_back(id); }
        cv.notify_one();
    }
};

// per-pair record a parse thread leaves next to the bases in its slab: where the pair's mates start inside the chunk's two base
// runs and inside its packed words; entry n (one past the last pair) holds the totals.  The GPU turns (chunk bases + these) into
// the batch's start / length / word-offset arrays (k_ingest.hip: expand_chunk_meta), so the calling thread touches no pair.
constexpr int CHUNK_META_CAP = 16384;      // pairs per chunk with in-slab metadata (2 MiB chunks hold ~6600 150-bp pairs); more: vectors

struct ParsedChunk {
    std::vector<uint8_t> s1, s2, flags;   // flags: PAIR_COUNT1 | PAIR_COUNT2 | PAIR_VOTE per kept pair
    std::vector<uint64_t> o1, o2;
    // with a slab: the kept bases lie in slab[0 .. n1), pair after pair, mate 1 then mate 2 (every base is written once, where the
    // copy to the device takes it from; round 4 kept the mates in two halves and moved the second one up at the end), the per-pair
    // records in meta[0 .. n_meta] instead of o1 / o2 / flags
    uint8_t* slab = nullptr;
    ChunkPairMeta* meta = nullptr;
    size_t cap = 0, n1 = 0;               // room for bases in the slab, bases written
    long n_meta = 0;
    uint32_t words = 0;
    uint64_t nkm = 0;
    long n_long = 0;                      // reads with more than FAST_NK k-mer offsets (lhgt_common.hpp: ReadBatch::n_long)
    int max_len = 0, k = 0;
    int slab_id = -1;
    bool src_slack = false;               // 16 readable bytes follow every source line (a worker's own text buffer): copies in 16-byte steps
    int rc = LHGT_OK;
    std::string err;
    void use_slab(uint8_t* base, size_t half_bytes, int id, int k_) {
        slab = base;
        cap = 2 * half_bytes;
        meta = (ChunkPairMeta*)(base + cap);
        slab_id = id;
        k = k_;
    }
    size_t bases_bytes() const { return slab ? n1 : s1.size() + s2.size(); }
    long n_pairs() const { return slab ? n_meta : (long)o1.size() - 1; }
    void spill() {   // unusual line structure (or more pairs than the slab's record area holds): continue in vectors
        s1.clear();
        s2.clear();
        o1.assign(1, 0);
        o2.assign(1, 0);
        flags.clear();
        for (long i = 0; i < n_meta; i++) {
            const size_t a = meta[i].rel1, m = meta[i].rel2, e = i + 1 < n_meta ? meta[i + 1].rel1 : n1;
            s1.insert(s1.end(), slab + a, slab + m);
            s2.insert(s2.end(), slab + m, slab + e);
            o1.push_back(s1.size());
            o2.push_back(s2.size());
            flags.push_back((uint8_t)meta[i].flags);
        }
        slab = nullptr;
        meta = nullptr;
    }
    static void copy16(uint8_t* dst, const uint8_t* src, size_t n) {   // writes and reads up to 15 bytes past n: the callers own that slack
        for (size_t i = 0; i < n; i += 16) _mm_storeu_si128((__m128i*)(dst + i), _mm_loadu_si128((const __m128i*)(src + i)));
    }
    void push(const uint8_t* a, size_t la, const uint8_t* b, size_t lb, uint8_t fl) {
        if (slab && (n1 + la + lb + 32 > cap || n_meta >= CHUNK_META_CAP)) spill();
        if (slab) {
            meta[n_meta++] = ChunkPairMeta{(uint32_t)n1, (uint32_t)(n1 + la), words, fl};
            if (src_slack) { copy16(slab + n1, a, la); copy16(slab + n1 + la, b, lb); }
            else { memcpy(slab + n1, a, la); memcpy(slab + n1 + la, b, lb); }
            n1 += la + lb;
            words += 3u * (uint32_t)((la + 31) / 32 + 1) + 3u * (uint32_t)((lb + 31) / 32 + 1);
            if ((int)la > max_len) max_len = (int)la;
            if ((int)lb > max_len) max_len = (int)lb;
            if ((int)la >= k) nkm += la - k + 1;
            if ((int)lb >= k) nkm += lb - k + 1;
            n_long += ((int)la - k + 1 > FAST_NK) + ((int)lb - k + 1 > FAST_NK);
        } else {
            s1.insert(s1.end(), a, a + la);
            s2.insert(s2.end(), b, b + lb);
            o1.push_back(s1.size());
            o2.push_back(s2.size());
            flags.push_back(fl);
        }
    }
    // slab chunks end as ONE contiguous block -- [bases][pad to 16 bytes][n + 1 records] -- so the calling thread issues one copy per chunk
    size_t block_bytes() const { return meta_off() + (size_t)(n_meta + 1) * sizeof(ChunkPairMeta); }
    size_t meta_off() const { return (n1 + 15) & ~(size_t)15; }
    void finish() {
        if (!slab) return;
        meta[n_meta] = ChunkPairMeta{(uint32_t)n1, (uint32_t)n1, words, 0u};
        ChunkPairMeta* dst = (ChunkPairMeta*)(slab + meta_off());
        memmove(dst, meta, (size_t)(n_meta + 1) * sizeof(ChunkPairMeta));   // the record area lies behind the bases: dst <= meta
        meta = dst;
    }
};

struct ChunkPlan {
    std::vector<size_t> start;   // byte offset of the first line of each chunk, plus file size at the end
    std::vector<long> line0;     // global index of that line, plus total line count at the end
};

inline size_t line_start_at_or_after(const uint8_t* p, size_t n, size_t from) {
    if (from == 0) return 0;
    if (from >= n) return n;
    const uint8_t* nl = (const uint8_t*)memchr(p + from - 1, '\n', n - (from - 1));
    return nl ? (size_t)(nl - p) + 1 : n;
}

// newlines in [q, end): 16 bytes at a time (compare, mask, popcount): ~4x a memchr per 60-150-byte line
inline long count_nl(const uint8_t* q, const uint8_t* end) {
    long c = 0;
    const __m128i nl16 = _mm_set1_epi8('\n');
    for (; q + 64 <= end; q += 64) {
        const unsigned m0 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)q), nl16));
        const unsigned m1 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(q + 16)), nl16));
        const unsigned m2 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(q + 32)), nl16));
        const unsigned m3 = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(q + 48)), nl16));
        c += __builtin_popcountll((unsigned long long)m0 | ((unsigned long long)m1 << 16) | ((unsigned long long)m2 << 32) | ((unsigned long long)m3 << 48));
    }
    for (; q < end; q++) c += *q == '\n';
    return c;
}

inline long count_lines(const uint8_t* p, size_t b0, size_t b1, size_t n) {
    long c = count_nl(p + b0, p + b1);
    if (b1 == n && n > b0 && p[n - 1] != '\n') c++;   // last line without a newline is still a line (std::getline)
    return c;
}

template <class Fn>
inline void parallel_for(long n, int threads, Fn fn) {
    if (threads <= 1 || n <= 1) { for (long i = 0; i < n; i++) fn(i); return; }
    std::vector<std::thread> th;
    std::atomic<long> next{0};
    int t = (int)(n < threads ? n : threads);
    for (int w = 0; w < t; w++) th.emplace_back([&]() { for (long i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& x : th) x.join();
}

inline size_t n_plan_chunks(size_t file_bytes, size_t chunk_bytes) { return file_bytes ? (file_bytes + chunk_bytes - 1) / chunk_bytes : 1; }


// ---------------------------------------------------------------- the reference's -t N read partition, by line numbers (host_fastx.cpp)
struct ThreadPart {
    std::vector<long> first, count;   // per thread chunk: global index of its first line, lines it consumes
    // sequence line g: in which chunk, and is it kept there?  (-1 = no chunk consumes it)
    int keep(long g, double ratio, const float* random_array) const {
        size_t i = (size_t)(std::upper_bound(first.begin(), first.end(), g) - first.begin());
        while (i > 0) {   // chunks are in file order; a later chunk never starts before an earlier one
            i--;
            if (g < first[i]) continue;
            if (g >= first[i] + count[i]) return -1;
            const long local = g - first[i];
            if (local % 4 != 1) return -1;
            return (ratio >= 100.0 || (double)random_array[(local / 4) % LHGT_MAX_RANDOM] < ratio) ? 1 : 0;
        }
        return -1;
    }
};

struct ThreadEmu {
    ThreadPart f1, f2;
    std::vector<long> pos1;   // byte at which each thread enters fq1 (and seeks fq2 to, E:350-352)
};

// How phase C pairs the lines of the two files (E:350-402).  It reads both in lock-step, line g of fq1 with line g + shift of fq2:
// shift is 0 when the first read IDs agree; otherwise the reference rewinds fq2 (to byte 1 at -t 1) and reads on until a line
// carries fq1's first ID (E:376-397), and the lock-step continues from there.  Once fq2 has run out std::getline leaves the string
// as it was: empty when fq2's last line ended with a newline, that last line otherwise (E:356-367) -- the `stale` partner.
// Phase A reads each file on its own (E:1426-1448), so fq2's records in front of the shift and behind fq1's end are counted, not voted.
struct PairLayout {
    long shift = 0;
    long lines1 = 0, lines2 = 0;
    const uint8_t* stale = nullptr;
    size_t stale_len = 0;
};

// Plans made elsewhere and the share of fq1's chunks this caller parses: the ranks of a multi-GPU run each count the lines of
// 1/world of both files, exchange the pieces, and parse chunks [part * nc / n_parts, (part + 1) * nc / n_parts) (lhgt_pairs_load_fastq_planned)
struct ParseShare {
    const ChunkPlan* p1 = nullptr;
    const ChunkPlan* p2 = nullptr;
    int part = 0, n_parts = 1;
    // the plans as they came in (chunk by chunk, empty chunks included): lets the parse check whether they lie on the column grid of
    // the single-pass loader (host_fastq_stream.cpp: seeded columns)
    const uint64_t *raw_start1 = nullptr, *raw_start2 = nullptr;
    const long *raw_count1 = nullptr, *raw_count2 = nullptr;
    long raw_n1 = 0, raw_n2 = 0;
};

// host_fastx.cpp
long thread_entry(const uint8_t* p, long n, long start);   // get_fq_start (E:44-89)
int ingest_default_threads();                              // the CPUs the process may use (affinity mask, cgroup quota), at most 48; LHGT_INGEST_THREADS
// host_fastq_stream.cpp: the single-pass loader.  LHGT_OK: every pair of the files went through consume(), plan1 / plan2 hold the
// line plans made on the way.  STREAM_RETRY (not an error code of the C-ABI): this pass does not decide the input -- *why says what
// it met -- and whatever consume() has seen must be dropped and the files given to the planned loader.
extern const int STREAM_RETRY;
bool stream_chunking(size_t n1, size_t n2, size_t chunk_bytes, size_t* ch1, size_t* ch2);
bool plan_columns(const Mapped& m, size_t ch, size_t c_lo, size_t c_hi, int threads, uint64_t* start, long* count, long* len_sums);
// line numbers made elsewhere (the planned loader's plans, when they lie on the column grid): columns [col_lo, col_hi) are parsed
// with no chain to wait for; emu / stale come from the caller's parse_setup
struct StreamSeeds {
    const long *P1 = nullptr, *P2 = nullptr;              // ncols + 1 entries each: lines in front of every column
    const uint64_t *start1 = nullptr, *start2 = nullptr;  // first line start of every column (checked against what the columns see)
    long ncols = 0, col_lo = 0, col_hi = 0;
    const ThreadEmu* emu = nullptr;
    const uint8_t* stale = nullptr;
    size_t stale_len = 0;
};
int parse_pairs_stream(const Mapped& m1, const Mapped& m2, const char* fq1, const char* fq2, double ratio, const float* random_array,
                       int shard_rank, int shard_world, long shard_block, int threads, size_t chunk_bytes, int emulate_threads,
                       const std::function<int(SlabPool**)>& prepare, const std::function<int(ParsedChunk&)>& consume,
                       const std::function<void(bool)>& idle, ChunkPlan* plan1, ChunkPlan* plan2, std::string* why, const StreamSeeds* seeds = nullptr);

}  // namespace lhgt
