// host_fastq_stream.cpp -- the FASTQ pair loader in ONE pass over the text (round 5).
//
// The planned loader (host_fastx.cpp) reads every byte twice: a line count of both files first -- the reference pairs line g of fq1
// with line g of fq2 (E:356-367), so a chunk can only be paired once the number of lines in front of it is known -- then the parse.
// At 32 M pairs the count was 0.10-0.16 s in front of a 0.25 s parse, both bound by per-line memchr calls and by the page faults of
// 48 threads on two mappings.  Here a worker takes COLUMN c -- chunk c of fq1 and the chunk of fq2 at the same relative position --
// and does everything for it while the text is in its cache:
//   1. the text of both chunks is read with pread() into the worker's own buffers (no page faults, no shared mapping), and one
//      SIMD sweep lists the offsets of its newlines (32 bytes per compare with AVX2);
//   2. the line counts are handed down a chain: column c waits for the global line numbers of its first lines (published by the
//      worker of column c - 1, which started earlier), adds its counts and publishes those of column c + 1 at once;
//   3. with the line numbers known, the pairs of fq1's chunk are cut straight from the two offset lists.  The few partner lines
//      that fall outside the fq2 chunk (the files drift against each other by a handful of lines) are walked to in the mapping.
// The result is the planned loader's, chunk by chunk: same pairs, same flags, same order (lhgt_fastq_parse_digest gives one digest
// for both; tests/test_host_cpu.py runs every odd file through both).  What this pass cannot decide locally it does not guess:
// differing first read IDs (E:368-402), a -t N thread that enters a file off a record boundary, files whose lines drift apart by
// more than DRIFT_MAX, a line longer than its chunk's margin, any error -- it returns STREAM_RETRY, the caller drops what was
// delivered and gives the input to the planned loader, whose refusals and messages are the contract.
#include "host_fastx.hpp"
#include <immintrin.h>
#include <linux/futex.h>
#include <sys/syscall.h>

namespace lhgt {

// ---------------------------------------------------------------- newline offsets of a block of text
// out[i] = offset of the i-th '\n' in t[0, n); returns how many.  out must hold n entries (a block of nothing but newlines).
static size_t scan_nl_sse2(const uint8_t* t, size_t n, uint32_t* out) {
    size_t k = 0, i = 0;
    const __m128i nl = _mm_set1_epi8('\n');
    for (; i + 16 <= n; i += 16) {
        unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128((const __m128i*)(t + i)), nl));
        while (m) { out[k++] = (uint32_t)(i + (size_t)__builtin_ctz(m)); m &= m - 1; }
    }
    for (; i < n; i++) if (t[i] == '\n') out[k++] = (uint32_t)i;
    return k;
}
__attribute__((target("avx2"))) static size_t scan_nl_avx2(const uint8_t* t, size_t n, uint32_t* out) {
    size_t k = 0, i = 0;
    const __m256i nl = _mm256_set1_epi8('\n');
    for (; i + 64 <= n; i += 64) {
        const uint64_t lo = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(t + i)), nl));
        const uint64_t hi = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(t + i + 32)), nl));
        uint64_t m = lo | (hi << 32);
        while (m) { out[k++] = (uint32_t)(i + (size_t)__builtin_ctzll(m)); m &= m - 1; }
    }
    for (; i < n; i++) if (t[i] == '\n') out[k++] = (uint32_t)i;
    return k;
}
static size_t scan_nl(const uint8_t* t, size_t n, uint32_t* out) {
    static const bool avx2 = __builtin_cpu_supports("avx2") && !getenv("LHGT_NO_AVX2");
    return avx2 ? scan_nl_avx2(t, n, out) : scan_nl_sse2(t, n, out);
}

namespace {

// A word that is 0 until its owner publishes; waiters sleep in the kernel instead of spinning (the GPU boxes run under a CPU quota:
// forty workers polling with sched_yield were measured to eat the quota the working ones needed -- the wait grew with the thread
// count until it was two thirds of a worker's time)
struct Gate {
    std::atomic<uint32_t> open{0};
    void wait(const std::atomic<bool>& stop) {
        for (int i = 0; i < 64 && !open.load(std::memory_order_acquire); i++) _mm_pause();
        while (!open.load(std::memory_order_acquire) && !stop.load()) {
            struct timespec ts = {0, 2000000};      // 2 ms: a stop request wakes nobody
            syscall(SYS_futex, (uint32_t*)&open, FUTEX_WAIT_PRIVATE, 0u, &ts, nullptr, 0);
        }
    }
    void release() {
        open.store(1, std::memory_order_release);
        syscall(SYS_futex, (uint32_t*)&open, FUTEX_WAKE_PRIVATE, INT32_MAX, nullptr, nullptr, 0);
    }
};

constexpr size_t MARGIN = (size_t)64 << 10;     // bytes read past a chunk's end to find the newline that closes its last line
constexpr long DRIFT_MAX = 1L << 16;            // partner lines outside the fq2 chunk a column may walk to (per side)

// one chunk of one file as a worker sees it: the text of [rb, rb + len) and the newline offsets in it
struct ChunkView {
    const uint8_t* t = nullptr;      // t[i] = file byte rb + i
    size_t rb = 0, len = 0;
    const uint32_t* nl = nullptr;    // newline offsets relative to rb
    size_t n_nl = 0;
    size_t S = 0, E = 0;             // the chunk's lines start in [S, E) (file bytes)
    size_t i_first = 0;              // index in nl of the newline that ends the chunk's first line
    long lines = 0;
    bool tail_open = false;          // the chunk's last line runs to the end of the file without a newline
    // line j of the chunk, j in [0, lines)
    void line(long j, const uint8_t** s, size_t* ln, size_t* start, size_t file_n) const {
        const size_t a = j == 0 ? S : rb + nl[i_first + (size_t)j - 1] + 1;
        const size_t b = (tail_open && j == lines - 1) ? file_n : rb + nl[i_first + (size_t)j];
        *s = t + (a - rb);
        *ln = b - a;
        *start = a;
    }
    long newlines_before(size_t byte) const {   // newlines in [S, byte), S <= byte <= E
        const uint32_t rel = (uint32_t)(byte - rb);
        return (long)(std::lower_bound(nl + i_first, nl + n_nl, rel) - (nl + i_first));
    }
};

struct Scratch {
    std::unique_ptr<uint8_t[]> text[2];
    std::unique_ptr<uint32_t[]> nl[2];
    size_t cap[2] = {0, 0};
    void reserve(int f, size_t bytes) {
        if (bytes <= cap[f]) return;
        cap[f] = bytes + bytes / 8;
        text[f].reset(new uint8_t[cap[f] + 64]);
        nl[f].reset(new uint32_t[cap[f] + 64]);
    }
};

// the thread chunks of the -t N emulation by BYTES (host_fastx.cpp: ThreadPart has them by line numbers, which this pass only
// learns on the way): thread i consumes the lines that start in [pos[i], stop[i]); first[i] = global number of the line at pos[i],
// filled in by the column that holds it (or by whoever needs it first)
struct ByteParts {
    const uint8_t* p = nullptr;
    size_t n = 0;
    std::vector<long> pos, stop;
    std::unique_ptr<std::atomic<long>[]> first;
    std::atomic<bool>* odd = nullptr;      // set when a thread turns out to enter off a record boundary (the planned loader refuses that)
    // Do all lines that start in [a, b) belong to ONE thread's chunk, whose first line number is known?  Then the keep test of a
    // sequence line g there is sampled((g - first) / 4) -- no search, no atomic: -> first line number, or -1
    long whole(size_t a, size_t b) const {
        if (a >= b) return -1;
        const size_t i1 = (size_t)(std::upper_bound(pos.begin(), pos.end(), (long)a) - pos.begin());
        if (i1 == 0) return -1;
        const size_t i = i1 - 1;
        if ((long)b > stop[i] || (i + 1 < pos.size() && pos[i + 1] < (long)b)) return -1;
        const long f = first[i].load(std::memory_order_acquire);
        return f >= 0 && f % 4 == 0 ? f : -1;
    }
    // the reference's keep test for the sequence line g that starts at byte sa (host_fastx.cpp: ThreadPart::keep).  *hint: the
    // part the caller's previous line fell into -- a column's lines come in file order, so the search is a step or two
    int keep(long g, size_t sa, double ratio, const float* random_array, size_t* hint) const {
        size_t i = *hint < pos.size() ? *hint : 0;
        if ((long)sa < pos[i]) i = (size_t)(std::upper_bound(pos.begin(), pos.end(), (long)sa) - pos.begin());   // (not in file order: a walk back in fq2)
        else { while (i + 1 < pos.size() && pos[i + 1] <= (long)sa) i++; i++; }
        if (i == 0) return -1;
        i--;
        *hint = i;
        if ((long)sa >= stop[i]) return -1;
        long f = first[i].load(std::memory_order_acquire);
        if (f < 0) {                     // not published yet (a partner line beyond this column's chunk): count back to the entry
            f = g - count_nl(p + pos[i], p + sa);
            if (f % 4) odd->store(true);
            first[i].store(f, std::memory_order_release);
        }
        const long local = g - f;
        if (local % 4 != 1) return -1;
        return (ratio >= 100.0 || (double)random_array[(local / 4) % LHGT_MAX_RANDOM] < ratio) ? 1 : 0;
    }
};

static int pread_all(int fd, uint8_t* dst, size_t len, size_t off) {
    while (len) {
        const ssize_t r = pread(fd, dst, len, (off_t)off);
        if (r <= 0) return -1;
        dst += r; off += (size_t)r; len -= (size_t)r;
    }
    return 0;
}

// chunk c of a file of n bytes cut at multiples of ch: text + newline list + line range.  false: a line longer than the margin
// (or a read error) -- the planned loader's business.
static bool view_chunk(const Mapped& m, size_t ch, long c, bool use_pread, Scratch* sc, int f, ChunkView* v) {
    const size_t n = m.n, lo = (size_t)c * ch, hi = lo + ch;
    *v = ChunkView();
    if (lo >= n) { v->S = v->E = n; return true; }
    const size_t rb = lo ? lo - 1 : 0, re = hi - 1 + MARGIN < n ? hi - 1 + MARGIN : n;
    v->rb = rb;
    v->len = re - rb;
    sc->reserve(f, v->len);
    if (use_pread) {
        if (pread_all(m.fd, sc->text[f].get(), v->len, rb)) return false;
        v->t = sc->text[f].get();
    } else v->t = m.p + rb;
    v->n_nl = scan_nl(v->t, v->len, sc->nl[f].get());
    v->nl = sc->nl[f].get();
    // S: byte 0, or one past the first newline at or after lo - 1 (= the first newline of the block)
    if (lo == 0) { v->S = 0; v->i_first = 0; }
    else if (v->n_nl == 0) {
        if (re < n) return false;                       // no line start within chunk + margin
        v->S = v->E = n;
        return true;
    } else { v->S = rb + v->nl[0] + 1; v->i_first = 1; }
    // E: one past the first newline at or after hi - 1, or the end of the file
    if (hi - 1 >= n) v->E = n;
    else {
        const uint32_t* q = std::lower_bound(v->nl, v->nl + v->n_nl, (uint32_t)(hi - 1 - rb));
        if (q == v->nl + v->n_nl) {
            if (re < n) return false;                   // the last line does not end within the margin
            v->E = n;
        } else v->E = rb + *q + 1;
    }
    if (v->E <= v->S) { v->E = v->S; v->lines = 0; return true; }
    // newlines in [S, E) + an unterminated last line
    const uint32_t* e_it = std::lower_bound(v->nl + v->i_first, v->nl + v->n_nl, (uint32_t)(v->E - rb));
    v->lines = (long)(e_it - (v->nl + v->i_first));
    if (v->E == n && m.p[n - 1] != '\n') { v->lines++; v->tail_open = true; }
    return true;
}

// the line in front of the one that starts at byte q (q = n stands for "behind the last line")
static bool line_before(const uint8_t* p, size_t n, size_t q, size_t* s, size_t* len) {
    if (q == 0) return false;
    size_t e = q - 1;                                   // the newline that ends it ...
    if (q == n && p[n - 1] != '\n') e = n;              // ... unless it is the file's unterminated last line
    const uint8_t* r = e ? (const uint8_t*)memrchr(p, '\n', e) : nullptr;
    *s = r ? (size_t)(r - p) + 1 : 0;
    *len = e - *s;
    return true;
}

}  // namespace

const int STREAM_RETRY = -1000;   // not an error code of the C-ABI: "give this input to the planned loader"

// The column grid of two files: fq1 in chunks of ch1 bytes, fq2 cut into the same NUMBER of chunks, so that column c of either
// file holds about the same lines.  false: no grid (an empty file, sizes a factor of two apart).
bool stream_chunking(size_t n1, size_t n2, size_t chunk_bytes, size_t* ch1, size_t* ch2) {
    if (n1 == 0 || n2 == 0 || n2 > 2 * n1 || n1 > 2 * n2 || chunk_bytes < 64) return false;
    *ch1 = std::min<size_t>(chunk_bytes, (size_t)1 << 30);      // newline offsets inside a chunk are 32-bit
    const size_t ncols = n_plan_chunks(n1, *ch1);
    *ch2 = std::max<size_t>((n2 + ncols - 1) / ncols, 64);
    return true;
}

// The line plan of chunks [c_lo, c_hi) of one file cut at multiples of ch -- first line start, number of lines, and (len_sums
// non-null) the summed line lengths by line index inside the chunk mod 4 (cal_sam_ratio's base count, E:1244-1270, folded into
// the count) -- read with pread and one SIMD sweep per chunk.  false: a chunk the sweep does not handle (a line longer than its
// margin); the caller falls back to its line-by-line pass.
bool plan_columns(const Mapped& m, size_t ch, size_t c_lo, size_t c_hi, int threads, uint64_t* start, long* count, long* len_sums) {
    std::atomic<bool> ok{true};
    const bool use_pread = !(getenv("LHGT_INGEST_IO") && !strcmp(getenv("LHGT_INGEST_IO"), "mmap"));
    std::atomic<long> next{(long)c_lo};
    auto work = [&]() {
        Scratch sc;
        for (long c; ok.load() && (c = next.fetch_add(1)) < (long)c_hi;) {
            ChunkView v;
            if (!view_chunk(m, ch, c, use_pread, &sc, 0, &v)) { ok.store(false); return; }
            const size_t i = (size_t)c - c_lo;
            start[i] = v.S;
            count[i] = v.lines;
            if (len_sums) {
                long sums[4] = {0, 0, 0, 0};
                size_t a = v.S;
                for (long j = 0; j < v.lines; j++) {
                    const size_t b = (v.tail_open && j == v.lines - 1) ? m.n : v.rb + v.nl[v.i_first + (size_t)j];
                    sums[j & 3] += (long)(b - a);
                    a = b + 1;
                }
                for (int r = 0; r < 4; r++) len_sums[4 * i + (size_t)r] = sums[r];
            }
        }
    };
    std::vector<std::thread> th;
    const long n = (long)(c_hi - c_lo);
    for (int w = 1; w < threads && w < n; w++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
    return ok.load();
}

int parse_pairs_stream(const Mapped& m1, const Mapped& m2, const char* fq1, const char* fq2, double ratio, const float* random_array,
                       int shard_rank, int shard_world, long shard_block, int threads, size_t chunk_bytes, int emulate_threads,
                       const std::function<int(SlabPool**)>& prepare, const std::function<int(ParsedChunk&)>& consume,
                       const std::function<void(bool)>& idle, ChunkPlan* plan1, ChunkPlan* plan2, std::string* why, const StreamSeeds* seeds) {
    auto retry = [&](const char* reason) { *why = reason; return STREAM_RETRY; };
    const size_t n1 = m1.n, n2 = m2.n;
    size_t ch1 = 0, ch2 = 0;
    if (!stream_chunking(n1, n2, chunk_bytes, &ch1, &ch2)) return retry("an empty file, or files of very different size");
    const long ncols = (long)n_plan_chunks(n1, ch1);
    const long col_lo = seeds ? seeds->col_lo : 0, col_hi = seeds ? seeds->col_hi : ncols, n_my = col_hi - col_lo;
    if (seeds && (seeds->ncols != ncols || col_lo < 0 || col_hi > ncols || col_lo > col_hi)) return retry("plans off the column grid");
    const char* io = getenv("LHGT_INGEST_IO");                      // pread (default) | mmap
    const bool use_pread = !(io && !strcmp(io, "mmap"));
    if (use_pread && getenv("LHGT_INGEST_NOREUSE")) {                // experiment knob: reads that do not touch the page cache's LRU state
        (void)posix_fadvise(m1.fd, 0, 0, POSIX_FADV_NOREUSE);
        (void)posix_fadvise(m2.fd, 0, 0, POSIX_FADV_NOREUSE);
    }
    const double t0 = now_s();
    // ---- what can be decided from bytes alone, before the first column (with seeds the caller has decided all of it from its plans)
    if (!seeds) {
        LineCursor a(m1), b(m2);
        const uint8_t *s1, *s2;
        size_t l1, l2, st;
        if (!a.next(&s1, &l1, &st) || !b.next(&s2, &l2, &st)) return retry("no first line");
        const size_t i1 = read_id_len(s1, l1), i2 = read_id_len(s2, l2);
        if (i1 != i2 || memcmp(s1, s2, i1)) return retry("first read IDs differ (E:368-402)");
    }
    const uint8_t* stale = m2.p;
    size_t stale_len = 0;
    if (seeds) { stale = seeds->stale; stale_len = seeds->stale_len; }
    else if (m2.p[n2 - 1] != '\n') { size_t s, l; line_before(m2.p, n2, n2, &s, &l); stale = m2.p + s; stale_len = l; }   // E:356-367
    std::atomic<bool> odd_entry{false};
    ByteParts bp1, bp2;
    std::unique_ptr<std::atomic<long>[]> g2_at;                            // line number of fq2 at the byte thread i enters fq1 at
    const bool emu = emulate_threads > 1;
    const ThreadEmu* emu_lines = seeds ? seeds->emu : nullptr;               // with seeds: the thread chunks by line numbers, from the plans
    if (emu && !seeds) {
        const long each = (long)n1 / emulate_threads;
        for (int f = 0; f < 2; f++) {
            const Mapped& m = f ? m2 : m1;
            ByteParts& bp = f ? bp2 : bp1;
            bp.p = m.p; bp.n = m.n; bp.odd = &odd_entry;
            bp.first.reset(new std::atomic<long>[(size_t)emulate_threads]);
            for (int i = 0; i < emulate_threads; i++) {
                const long start = (long)i * each, end = i == emulate_threads - 1 ? (long)n1 : (long)(i + 1) * each;
                const long pos = thread_entry(m.p, (long)m.n, start);
                if (pos < 0) return retry("a thread enters within 1000 bytes of the end");
                const long stop = (size_t)end + 1 >= m.n ? (long)m.n : (long)line_start_at_or_after(m.p, m.n, (size_t)end + 1);
                if (i > 0 && pos < std::max(bp.pos.back(), bp.stop.back())) return retry("overlapping thread chunks");
                bp.pos.push_back(pos);
                bp.stop.push_back(stop);
                bp.first[(size_t)i].store(pos == 0 ? 0 : -1);
            }
        }
        g2_at.reset(new std::atomic<long>[(size_t)emulate_threads]);
        for (int i = 0; i < emulate_threads; i++) {
            g2_at[(size_t)i].store(-1);
            if (bp1.stop[(size_t)i] <= bp1.pos[(size_t)i]) continue;        // a thread without a line compares nothing (E:350-402)
            const size_t pos = (size_t)bp1.pos[(size_t)i];
            if (pos >= n2) return retry("a thread seeks fq2 behind its end");
            const uint8_t* e1 = (const uint8_t*)memchr(m1.p + pos, '\n', n1 - pos);
            const uint8_t* e2 = (const uint8_t*)memchr(m2.p + pos, '\n', n2 - pos);
            const size_t l1 = e1 ? (size_t)(e1 - (m1.p + pos)) : n1 - pos, l2 = e2 ? (size_t)(e2 - (m2.p + pos)) : n2 - pos;
            const size_t i1 = read_id_len(m1.p + pos, l1), i2 = read_id_len(m2.p + pos, l2);
            if (i1 != i2 || memcmp(m1.p + pos, m2.p + pos, i1)) return retry("a thread finds another read ID at its byte of fq2 (E:368-402)");
            if (pos == 0) g2_at[(size_t)i].store(0);
        }
    }
    // ---- the columns
    std::unique_ptr<std::atomic<long>[]> P1(new std::atomic<long>[(size_t)ncols + 1]), P2(new std::atomic<long>[(size_t)ncols + 1]);
    std::unique_ptr<Gate[]> gate(new Gate[(size_t)ncols + 1]);       // gate[c] opens when P1[c] and P2[c] are known
    for (long c = 0; c <= ncols; c++) { P1[(size_t)c].store(seeds ? seeds->P1[c] : -1); P2[(size_t)c].store(seeds ? seeds->P2[c] : -1); }
    if (!seeds) { P1[0].store(0); P2[0].store(0); gate[0].release(); }
    std::vector<size_t> S1((size_t)ncols + 1, n1), S2((size_t)ncols + 1, n2);   // chunk starts, for the plans handed back
    std::vector<ParsedChunk> out((size_t)n_my);
    std::unique_ptr<std::atomic<int>[]> ready(new std::atomic<int>[(size_t)n_my + 1]);
    for (long c = 0; c < n_my; c++) ready[(size_t)c].store(0);
    std::atomic<long> next{col_lo};
    std::atomic<bool> stop{false};
    std::atomic<int> fail{0};                       // 1 = retry with the planned loader, 2 = error in err_rc / err_msg
    std::mutex mu;
    std::condition_variable cv_ready;
    std::string fail_why, err_msg;
    int err_rc = LHGT_OK;
    auto give_up = [&](int kind, const std::string& msg, int rc = LHGT_OK) {
        std::lock_guard<std::mutex> lk(mu);
        if (!fail.load()) { fail.store(kind); if (kind == 1) fail_why = msg; else { err_msg = msg; err_rc = rc; } }
        stop.store(true);
    };
    std::atomic<SlabPool*> pool_ptr{nullptr};
    std::atomic<bool> pool_known{false};
    std::mutex stage_mu;
    double stage_s[5] = {0, 0, 0, 0, 0};            // summed over the workers: slab wait, read + newline scan, chain wait, pairs, (unused)
    auto sampled = [&](long n) { return ratio >= 100.0 || (double)random_array[n % LHGT_MAX_RANDOM] < ratio; };

    auto worker = [&]() {
        Scratch sc;
        std::vector<std::pair<size_t, size_t>> before;     // partner lines in front of the fq2 chunk: (start, length)
        double my_s[4] = {0, 0, 0, 0};
        struct Sum { double* a; double* b; std::mutex* m; ~Sum() { std::lock_guard<std::mutex> lk(*m); for (int i = 0; i < 4; i++) a[i] += b[i]; } } sum{stage_s, my_s, &stage_mu};
        while (!pool_known.load(std::memory_order_acquire)) { if (stop.load()) return; usleep(50); }
        SlabPool* pool = pool_ptr.load();
        for (;;) {
            int slab_id = -1;
            double ta = now_s();
            if (pool) {                                  // before the column number: every earlier column already holds its slab
                slab_id = pool->acquire();
                if (slab_id < 0) return;
            }
            const long c = next.fetch_add(1);
            if (c >= col_hi || stop.load()) { if (pool && slab_id >= 0) pool->release(slab_id); return; }
            double tb = now_s();
            my_s[0] += tb - ta;
            ParsedChunk& ch = out[(size_t)(c - col_lo)];
            if (pool) ch.use_slab(pool->base + (size_t)slab_id * pool->slab_bytes, pool->half_bytes, slab_id, pool->k);
            ch.src_slack = use_pread;                    // the text buffers end in 64 spare bytes
            ch.o1.assign(1, 0);
            ch.o2.assign(1, 0);
            // 1. text and newline lists of both chunks
            ChunkView v1, v2;
            const bool seen = view_chunk(m1, ch1, c, use_pread, &sc, 0, &v1) && view_chunk(m2, ch2, c, use_pread, &sc, 1, &v2);
            if (!seen) give_up(1, "a line longer than a chunk's margin");
            ta = now_s();
            my_s[1] += ta - tb;
            // 2. the chain of line numbers: wait for this column's, publish the next one's
            if (!seeds) gate[(size_t)c].wait(stop);
            long g0 = P1[(size_t)c].load(std::memory_order_acquire), h0 = P2[(size_t)c].load(std::memory_order_acquire);
            if (seeds && seen && (v1.lines != seeds->P1[c + 1] - g0 || v2.lines != seeds->P2[c + 1] - h0 || (v1.lines && v1.S != seeds->start1[c]) ||
                                  (v2.lines && v2.S != seeds->start2[c])))
                give_up(1, "the plans do not describe these columns");
            if (g0 < 0 || h0 < 0) { g0 = h0 = 0; }       // stopping: numbers no longer matter, the chain must still move
            tb = now_s();
            my_s[2] += tb - ta;
            S1[(size_t)c] = v1.S;
            S2[(size_t)c] = v2.S;
            if (emu && seen && !seeds) {                 // threads that enter inside these chunks learn their first line's number
                for (int f = 0; f < 2; f++) {
                    const ChunkView& v = f ? v2 : v1;
                    ByteParts& bp = f ? bp2 : bp1;
                    const long base = f ? h0 : g0;
                    for (size_t i = 0; i < bp.pos.size(); i++) {
                        const size_t pos = (size_t)bp.pos[i];
                        if (pos < v.S || pos >= v.E) continue;
                        const long fl = base + v.newlines_before(pos);
                        if (fl % 4) odd_entry.store(true);
                        bp.first[i].store(fl, std::memory_order_release);
                    }
                }
                for (int i = 0; i < emulate_threads; i++) {        // ... and fq2's line number at the byte a thread seeks it to
                    const size_t pos = (size_t)bp1.pos[(size_t)i];
                    if (bp1.stop[(size_t)i] > bp1.pos[(size_t)i] && pos >= v2.S && pos < v2.E)
                        g2_at[(size_t)i].store(h0 + v2.newlines_before(pos), std::memory_order_release);
                }
            }
            if (!seeds) {
                P1[(size_t)c + 1].store(g0 + v1.lines, std::memory_order_release);
                P2[(size_t)c + 1].store(h0 + v2.lines, std::memory_order_release);
                gate[(size_t)c + 1].release();
            }
            // 3. the pairs of fq1's chunk
            if (seen && !stop.load() && v1.lines > 0) {
                const long h1 = h0 + v2.lines;           // fq2's chunk holds lines [h0, h1)
                bool ok = true;
                before.clear();
                if (g0 < h0) {                           // partner lines in front of fq2's chunk: walk back from its first line
                    const long need = std::min(h0 - g0, v1.lines);
                    if (h0 - g0 > DRIFT_MAX) { give_up(1, "the files' lines drift apart"); ok = false; }
                    else {
                        size_t q = v2.S, s, l;
                        for (long i = 0; i < h0 - g0 && ok; i++) {
                            if (!line_before(m2.p, n2, q, &s, &l)) { give_up(1, "line numbers of fq2 do not add up"); ok = false; break; }
                            if (i >= h0 - g0 - need) before.emplace_back(s, l);
                            q = s;
                        }
                        std::reverse(before.begin(), before.end());   // before[j] = line g0 + j
                    }
                }
                size_t hint1 = 0, hint2 = 0;
                // a chunk that lies inside one thread's chunk (485 of 4853 columns at -t 10 hold a boundary): its lines' keep test is
                // arithmetic on the line number
                const long whole1 = emu && !emu_lines ? bp1.whole(v1.S, v1.E) : -1, whole2 = emu && !emu_lines ? bp2.whole(v2.S, v2.E) : -1;
                LineCursor fwd(m2);                      // partner lines behind fq2's chunk
                fwd.cur = v2.E;
                long fwd_idx = h1, walked = 0;
                for (long j = 0; j < v1.lines && ok; j++) {
                    const long g = g0 + j;
                    if (g % 4 != 1) continue;
                    const uint8_t *a, *b;
                    size_t la, lb, sa, sb;
                    v1.line(j, &a, &la, &sa, n1);
                    bool have2 = true;
                    const long h = g;                    // first read IDs agree: line g pairs with line g (E:356-367)
                    if (h < h0) {
                        if ((size_t)j < before.size()) { sb = before[(size_t)j].first; lb = before[(size_t)j].second; b = m2.p + sb; }
                        else { give_up(1, "line numbers of fq2 do not add up"); ok = false; break; }
                    } else if (h < h1) v2.line(h - h0, &b, &lb, &sb, n2);
                    else {
                        while (have2 && fwd_idx <= h) {
                            have2 = fwd.next(&b, &lb, &sb);
                            fwd_idx++;
                            if (++walked > DRIFT_MAX) { give_up(1, "the files' lines drift apart"); ok = false; break; }
                        }
                        if (!ok) break;
                        if (!have2) { fwd_idx = h + 1; fwd.cur = n2; }
                    }
                    if (!have2) { b = stale; lb = stale_len; sb = n2; }       // fq2 has run out (E:356-367)
                    const long n = g / 4;
                    uint8_t fl;
                    if (emu_lines)
                        fl = (uint8_t)((emu_lines->f1.keep(g, ratio, random_array) == 1 ? PAIR_COUNT1 | PAIR_VOTE : 0) |
                                       (have2 && emu_lines->f2.keep(h, ratio, random_array) == 1 ? PAIR_COUNT2 : 0));
                    else if (emu) {
                        const bool in2 = h >= h0 && h < h1;      // the partner line lies in fq2's chunk (not walked to)
                        const int k1 = whole1 >= 0 ? (int)sampled((g - whole1) / 4) : bp1.keep(g, sa, ratio, random_array, &hint1);
                        const int k2 = !have2 ? 0 : whole2 >= 0 && in2 ? ((h - whole2) % 4 == 1 ? (int)sampled((h - whole2) / 4) : -1)
                                                                       : bp2.keep(h, sb, ratio, random_array, &hint2);
                        fl = (uint8_t)((k1 == 1 ? PAIR_COUNT1 | PAIR_VOTE : 0) | (k2 == 1 ? PAIR_COUNT2 : 0));
                    }
                    else
                        fl = (uint8_t)((sampled(n) ? PAIR_COUNT1 | PAIR_VOTE : 0) | (have2 && sb <= n1 && sampled(h / 4) ? PAIR_COUNT2 : 0));   // quirk Q4
                    if (!fl || (shard_world > 1 && (n / shard_block) % shard_world != shard_rank)) continue;
                    if (!(fl & (PAIR_COUNT1 | PAIR_VOTE))) la = 0;
                    if (!(fl & (PAIR_COUNT2 | PAIR_VOTE))) lb = 0;
                    if (la > LHGT_MAX_READ_LEN || lb > LHGT_MAX_READ_LEN) { give_up(1, "a read longer than the reference's buffers"); ok = false; break; }
                    ch.push(a, la, b, lb, fl);
                }
                if (odd_entry.load()) give_up(1, "a thread enters a file off a record boundary");
            }
            ch.finish();
            my_s[3] += now_s() - tb;
            { std::lock_guard<std::mutex> lk(mu); ready[(size_t)(c - col_lo)].store(1); }
            cv_ready.notify_all();
        }
    };
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    const int nt = (int)(n_my < threads ? n_my : threads);
    for (int w = 0; w < nt; w++) th.emplace_back(worker);
    // the calling thread allocates (pinned slabs, device staging) while the workers read and scan their first columns
    SlabPool* pool = nullptr;
    int rc = prepare ? prepare(&pool) : LHGT_OK;
    const std::string prepare_err = rc != LHGT_OK ? std::string(last_error()) : std::string();
    if (rc != LHGT_OK) stop.store(true);
    pool_ptr.store(pool);
    pool_known.store(true, std::memory_order_release);
    long n_consumed = 0;
    double t_wait = 0, t_consume = 0;
    for (long c = 0; c < n_my && rc == LHGT_OK && !fail.load(); c++) {
        const double t1 = now_s();
        {
            std::unique_lock<std::mutex> lk(mu);
            while (!ready[(size_t)c].load()) {
                lk.unlock();
                idle(false);
                lk.lock();
                if (ready[(size_t)c].load()) break;
                if (cv_ready.wait_for(lk, std::chrono::microseconds(200)) == std::cv_status::timeout && pool) {
                    lk.unlock();
                    idle(true);
                    lk.lock();
                }
            }
        }
        const double t2 = now_s();
        if (fail.load()) break;
        ParsedChunk& ch = out[(size_t)c];
        if (ch.slab_id >= 0 && !ch.slab) { pool->release(ch.slab_id); ch.slab_id = -1; }
        rc = consume(ch);
        ch.slab_id = -1;
        n_consumed = c + 1;
        ParsedChunk().s1.swap(ch.s1);
        ParsedChunk().s2.swap(ch.s2);
        ParsedChunk().o1.swap(ch.o1);
        ParsedChunk().o2.swap(ch.o2);
        t_wait += t2 - t1;
        t_consume += now_s() - t2;
    }
    const std::string consume_err = !prepare_err.empty() ? prepare_err : rc != LHGT_OK ? std::string(last_error()) : std::string();
    stop.store(true);
    next.store(ncols);
    if (pool) pool->close();
    for (auto& t : th) t.join();
    if (pool) {
        for (long c = n_consumed; c < n_my; c++)
            if (out[(size_t)c].slab_id >= 0) pool->release(out[(size_t)c].slab_id);
        pool->reopen();
    }
    if (rc != LHGT_OK) LHGT_FAIL(rc, "%s", consume_err.c_str());
    if (fail.load() == 2) LHGT_FAIL(err_rc, "%s", err_msg.c_str());
    if (fail.load() == 1) return retry(fail_why.c_str());
    if (seeds) {
        if (ingest_trace())
            fprintf(stderr, "[lhgt ingest] columns [%ld, %ld) of %ld on given line numbers (%s): %d threads, waited for columns %.3fs, consume(+upload) %.3fs, whole pass %.3fs; "
                    "per worker: slab wait %.3fs, read + newline scan %.3fs, pairs %.3fs\n", col_lo, col_hi, ncols, use_pread ? "pread" : "mmap", nt, t_wait, t_consume,
                    now_s() - t0, stage_s[0] / std::max(nt, 1), stage_s[1] / std::max(nt, 1), stage_s[3] / std::max(nt, 1));
        return LHGT_OK;
    }
    // ---- what only the whole pass can tell
    const long lines1 = P1[(size_t)ncols].load(), lines2 = P2[(size_t)ncols].load();
    if (emu) {
        if (odd_entry.load()) return retry("a thread enters a file off a record boundary");
        for (int i = 0; i < emulate_threads; i++) {
            if (bp1.stop[(size_t)i] <= bp1.pos[(size_t)i]) continue;
            const long f1 = bp1.first[(size_t)i].load(), g2 = g2_at[(size_t)i].load();
            if (f1 < 0 || g2 < 0 || f1 != g2) return retry("the threads pair the files at different line offsets");
        }
        if (lines2 < lines1)
            for (long g = lines2; g < lines1; g++)
                if (g % 4 == 1) return retry("fq2 has fewer records than fq1");
    }
    // the plans this pass has made on the way (for the records of fq2 behind fq1's end, and for the caller's bookkeeping)
    for (int f = 0; f < 2; f++) {
        ChunkPlan* pl = f ? plan2 : plan1;
        const std::vector<size_t>& S = f ? S2 : S1;
        const std::atomic<long>* P = f ? P2.get() : P1.get();
        const size_t n = f ? n2 : n1;
        pl->start.clear();
        pl->line0.assign(1, 0);
        for (long c = 0; c < ncols; c++) {
            const size_t st = S[(size_t)c], en = c + 1 < ncols ? S[(size_t)c + 1] : n;
            if (en <= st) continue;
            pl->start.push_back(st);
            pl->line0.push_back(P[(size_t)c + 1].load());
        }
        if (pl->start.empty()) { pl->start.push_back(0); pl->line0.push_back(0); }
        pl->start.push_back(n);
    }
    if (ingest_trace())
        fprintf(stderr, "[lhgt ingest] one pass (%s): %d threads, %ld columns of %zu + %zu bytes, %.1f MB of %s: waited for columns %.3fs, consume(+upload) %.3fs, whole pass %.3fs; "
                "per worker: slab wait %.3fs, read + newline scan %.3fs, chain wait %.3fs, pairs %.3fs\n",
                use_pread ? "pread" : "mmap", nt, ncols, ch1, ch2, 1e-6 * (double)n1, fq1, t_wait, t_consume, now_s() - t0,
                stage_s[0] / nt, stage_s[1] / nt, stage_s[2] / nt, stage_s[3] / nt);
    return LHGT_OK;
}

}  // namespace lhgt
