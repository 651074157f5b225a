// host_fastx.cpp -- FASTA / FASTQ ingest on the host (mmap + memchr), reference line semantics.
#include "host_fastx.hpp"
#include <sched.h>
#include <cctype>
#include <cstring>

namespace lhgt {



// ---------------------------------------------------------------- FASTA streaming (read_ref E:761-880)
// Calls fn(name, ref_index, seq, len, cumulative_len) for every contig with len > k, in file
// order.  ref_index counts every '>' line, skipped contigs included (E:825, quirk Q7).
template <class Fn>
static int for_each_contig(const char* fasta_path, int k, Fn fn) {
    Mapped fa;
    LHGT_TRY(fa.open(fasta_path));
    LineCursor lc(fa);
    std::vector<uint8_t> seq;
    std::string pending = "start", name;
    long ref_index = 0, cum = 0;
    const uint8_t* s;
    size_t len, start;
    while (lc.next(&s, &len, &start)) {
        if (len > 0 && s[0] == '>') {
            name = pending;
            size_t idl = read_id_len(s, len);
            pending.assign((const char*)s + (idl ? 1 : 0), idl ? idl - 1 : 0);
            cum += (long)seq.size();
            if ((long)seq.size() > k) LHGT_TRY(fn(name, ref_index, seq.data(), (long)seq.size(), cum));
            ref_index++;
            seq.clear();
        } else {
            seq.insert(seq.end(), s, s + len);
        }
    }
    cum += (long)seq.size();
    if ((long)seq.size() > k) LHGT_TRY(fn(pending, ref_index, seq.data(), (long)seq.size(), cum));
    return LHGT_OK;
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

// ---------------------------------------------------------------- parallel lock-step FASTQ parser
// The reference pairs line i of fq1 with line i of fq2 and treats lines with i % 4 == 1 as sequences
// (E:356-367, 403-419): purely line-indexed.  So ANY line start is a valid split point: pass 1 counts the
// lines of byte chunks of both files, pass 2 parses the fq1 chunks in parallel, each thread locating its
// first global line in fq2 through fq2's per-chunk line counts.  Results are delivered in file order.
}  // extern "C"

namespace lhgt {




// Chunk c of a file starts at the first line start at or after byte c * chunk_bytes.  plan_range counts the lines of chunks
// [c_lo, c_hi): start[] gets c_hi - c_lo + 1 boundaries, count[] the lines between them.  A whole plan is the range [0, n_chunks);
// the ranks of a multi-GPU run each count a share of the chunks and exchange the pieces (lhgt_fastq_plan_part).
static void plan_range(const Mapped& m, size_t chunk_bytes, size_t c_lo, size_t c_hi, int threads, std::vector<size_t>* start, std::vector<long>* count) {
    const size_t nchunks = n_plan_chunks(m.n, chunk_bytes);
    start->assign(c_hi - c_lo + 1, 0);
    for (size_t c = c_lo; c <= c_hi; c++) (*start)[c - c_lo] = c >= nchunks ? m.n : line_start_at_or_after(m.p, m.n, c * chunk_bytes);
    count->assign(c_hi - c_lo, 0);
    static const int populate = getenv("LHGT_MMAP_POPULATE") ? atoi(getenv("LHGT_MMAP_POPULATE")) : 0;   // experiment knob: 1 = MADV_POPULATE_READ per chunk
    parallel_for((long)(c_hi - c_lo), threads, [&](long i) {
        const size_t a = (*start)[(size_t)i], b = (*start)[(size_t)i + 1];
        if (populate && b > a) {
            const size_t pa = a & ~(size_t)4095;
            (void)madvise((void*)(m.p + pa), b - pa, 22 /* MADV_POPULATE_READ */);
        }
        (*count)[(size_t)i] = count_lines(m.p, a, b, m.n);
    });
}

// chunks without a byte of their own (a line longer than a chunk) are dropped
static int plan_from_arrays(const Mapped& m, const uint64_t* start, const long* count, long n, ChunkPlan* pl) {
    pl->start.clear();
    pl->line0.assign(1, 0);
    uint64_t prev = 0;
    for (long c = 0; c < n; c++) {
        const uint64_t st = start[c], en = c + 1 < n ? start[c + 1] : m.n;
        if (st < prev || en < st || en > m.n || (st > 0 && st < m.n && m.p[st - 1] != '\n') || count[c] < 0 || (c == 0 && st != 0))
            LHGT_FAIL(LHGT_E_ARG, "FASTQ plan: chunk %ld [%llu, %llu) is not a run of whole lines of the %zu-byte file", c, (unsigned long long)st, (unsigned long long)en, m.n);
        prev = st;
        if (en == st) {
            if (count[c]) LHGT_FAIL(LHGT_E_ARG, "FASTQ plan: empty chunk %ld with %ld lines", c, count[c]);
            continue;
        }
        pl->start.push_back((size_t)st);
        pl->line0.push_back(pl->line0.back() + count[c]);
    }
    if (pl->start.empty()) { pl->start.push_back(0); pl->line0.push_back(0); }
    pl->start.push_back(m.n);
    return LHGT_OK;
}

static ChunkPlan plan_chunks(const Mapped& m, size_t chunk_bytes, int threads) {
    std::vector<size_t> st;
    std::vector<long> cnt;
    plan_range(m, chunk_bytes, 0, n_plan_chunks(m.n, chunk_bytes), threads, &st, &cnt);
    std::vector<uint64_t> st64(st.begin(), st.end() - 1);
    ChunkPlan pl;
    (void)plan_from_arrays(m, st64.data(), cnt.data(), (long)cnt.size(), &pl);   // boundaries made here are line starts by construction
    return pl;
}

// ---------------------------------------------------------------- the reference's -t N read partition
// Thread i of N reads the byte chunk [i*(size/N), (i+1)*(size/N)] (the last one to `size`; E:1426-1434) -- `size` being fq1's
// for BOTH files (E:1419, quirk Q4).  It enters the file at get_fq_start(start) and consumes lines while the byte offset of the
// line's start is <= end (E:1022-1026); sampling ordinals count sequence lines from the chunk's first line (E:1037).  A record
// cut by a boundary is lost or half-read exactly as there.  One ThreadPart per file: the global line ranges the chunks consume.

// Where thread chunk `start` enters the file (the reference's get_fq_start, E:44-89), restated as the automaton it is.
// The reference looks at byte pairs (p[j], p[j+1]) through windows of 1000 positions: window w covers j = w .. w + 999, the first
// window is w = start, and every window that ends without a result is followed by the one that starts ONE BYTE EARLIER (w - 1),
// down to w = 1.  Two things survive from window to window: whether a "\n+" line start has been seen (`armed`) and how many
// newlines were counted since (`nl`; since the beginning of the scan while not armed).  Only newlines change that state:
//   a newline followed by '+' arms the scan and restarts the count; every newline then counts;
//   the third newline after a "\n+", when followed by '@', is the answer: the record after the '+' line's quality line;
//   any newline that brings the count to three without that ends the window, and a window entered with the count already at
//   three ends at its first byte unless that byte is a newline (the count then moves past three and the window scans on).
// Reaching the last byte of the file (j + 1 >= n) is where the reference's ifstream hits EOF and reads stale bytes: -1.
// No window finding anything leaves the thread at byte 0.  Checked against the oracle's literal loop (orc_get_fq_start) on
// random byte soups (tests/test_host_cpu.py).
long thread_entry(const uint8_t* p, long n, long start) {
    bool armed = false;
    int nl = 0;
    for (long w = start; w > 0; w--) {
        long j = w;
        const long w_end = w + 1000;                          // one past the window's last position
        if (nl == 3) {                                        // the count was left at three: only a newline right here keeps the window alive
            if (j + 1 >= n) return -1;
            if (p[j] != '\n') continue;
        }
        for (;;) {
            // next newline this window can look at: positions j .. min(w_end, n - 1) - 1
            const long lim = w_end < n - 1 ? w_end : n - 1;
            const uint8_t* q = j < lim ? (const uint8_t*)memchr(p + j, '\n', (size_t)(lim - j)) : nullptr;
            if (!q) {
                if (w_end > n - 1) return -1;                 // the window walks into the file's last byte
                break;                                        // 1000 positions without a decision
            }
            const long at = (long)(q - p);
            const uint8_t after = p[at + 1];
            if (after == '+') { armed = true; nl = 0; }
            nl++;
            if (armed && after == '@' && nl == 3) return at + 1;
            if (nl == 3) break;
            j = at + 1;
        }
    }
    return 0;
}


static long line_index_of(const Mapped& m, const ChunkPlan& pl, size_t byte_pos) {
    if (byte_pos >= m.n) return pl.line0.back();
    size_t c = (size_t)(std::upper_bound(pl.start.begin(), pl.start.end(), byte_pos) - pl.start.begin()) - 1;
    return pl.line0[c] + count_lines(m.p, pl.start[c], byte_pos, m.n + 1);   // newlines in [chunk start, byte_pos)
}

static int thread_part(const Mapped& m, const ChunkPlan& pl, long size_for_chunks, int threads, const char* path, ThreadPart* out,
                       std::vector<long>* pos_out = nullptr) {
    const long each = size_for_chunks / threads;
    for (int i = 0; i < threads; i++) {
        const long start = (long)i * each, end = i == threads - 1 ? size_for_chunks : (long)(i + 1) * each;
        const long pos = thread_entry(m.p, (long)m.n, start);
        if (pos < 0) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: thread %d would enter %s within 1000 bytes of its end (the reference reads stale bytes there)", threads, i, path);
        const long first = line_index_of(m, pl, (size_t)pos);
        if (first % 4) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: thread %d enters %s at line %ld, not at a record (the reference would take quality lines for reads)", threads, i, path, first);
        const size_t next = line_start_at_or_after(m.p, m.n, (size_t)end + 1);    // first line the chunk does NOT consume
        const long stop = (size_t)end + 1 >= m.n ? pl.line0.back() : line_index_of(m, pl, next);
        out->first.push_back(first);
        out->count.push_back(stop > first ? stop - first : 0);
        if (pos_out) pos_out->push_back(pos);
    }
    for (int i = 0; i + 1 < threads; i++)   // the reference would count such reads twice; only files of a few records per thread get here
        if (out->first[i + 1] < out->first[i] + out->count[i])
            LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: the chunks of threads %d and %d of %s overlap (file too small for that many threads)", threads, i, i + 1, path);
    return LHGT_OK;
}



// line g of a planned file: its bytes (without the newline) and its start offset
static bool line_at(const Mapped& m, const ChunkPlan& pl, long g, const uint8_t** s, size_t* len, size_t* start) {
    if (g < 0 || g >= pl.line0.back()) return false;
    const long c = (long)(std::upper_bound(pl.line0.begin(), pl.line0.end(), g) - pl.line0.begin()) - 1;
    LineCursor k(m);
    k.cur = pl.start[(size_t)c];
    for (long skip = g - pl.line0[(size_t)c]; skip >= 0; skip--)
        if (!k.next(s, len, start)) return false;
    return true;
}

static bool same_id(const uint8_t* a, size_t la, const uint8_t* b, size_t lb) {
    const size_t ia = read_id_len(a, la), ib = read_id_len(b, lb);
    return ia == ib && memcmp(a, b, ia) == 0;
}

// The reference's search (E:376-397): fq2 read line by line from byte `from` (a first partial line counts) until get_read_ID of a
// line equals `id`.  Returns the index of that line, -1 when no line has it (the reference then spins through 10^9 failing reads).
static long find_id_line(const Mapped& m2, const ChunkPlan& p2, size_t from, const uint8_t* id, size_t idl) {
    const uint8_t* p = m2.p;
    const size_t n = m2.n;
    if (from >= n) return -1;
    auto is_hit = [&](size_t q) {
        const uint8_t* nl = (const uint8_t*)memchr(p + q, '\n', n - q);
        const size_t len = nl ? (size_t)(nl - (p + q)) : n - q;
        return read_id_len(p + q, len) == idl && memcmp(p + q, id, idl) == 0;
    };
    if (idl == 0) {   // an empty ID (fq1 opens with an empty line or a delimiter): every line has to be looked at
        for (size_t q = from; q < n;) {
            if (is_hit(q)) return line_index_of(m2, p2, q);
            const uint8_t* nl = (const uint8_t*)memchr(p + q, '\n', n - q);
            if (!nl) break;
            q = (size_t)(nl - p) + 1;
        }
        return -1;
    }
    for (size_t q = from; q + idl <= n;) {
        const uint8_t* hit = (const uint8_t*)memmem(p + q, n - q, id, idl);
        if (!hit) break;
        const size_t at = (size_t)(hit - p);
        if ((at == from || p[at - 1] == '\n') && is_hit(at)) return line_index_of(m2, p2, at);
        q = at + 1;
    }
    return -1;
}

// Parse fq1 chunk c (all lines starting in it) against the lines of fq2 that phase C pairs them with.
static void parse_chunk(const Mapped& m1, const Mapped& m2, const ChunkPlan& p1, const ChunkPlan& p2, long c, double ratio,
                        const float* random_array, int shard_rank, int shard_world, long shard_block, const ThreadEmu* emu,
                        const PairLayout& lay, ParsedChunk* out) {
    out->o1.assign(1, 0);
    out->o2.assign(1, 0);
    const long g0 = p1.line0[c], g1 = p1.line0[c + 1];
    if (g0 == g1) { out->finish(); return; }
    // fq2 cursor at global line g0 + shift
    LineCursor k1(m1), k2(m2);
    k1.cur = p1.start[c];
    const uint8_t *a, *b;
    size_t la, lb, sa, sb;
    const long h0 = g0 + lay.shift;
    if (h0 < lay.lines2) {
        long c2 = (long)(std::upper_bound(p2.line0.begin(), p2.line0.end(), h0) - p2.line0.begin()) - 1;
        k2.cur = p2.start[c2];
        for (long skip = h0 - p2.line0[c2]; skip > 0; skip--) k2.next(&b, &lb, &sb);
    } else k2.cur = m2.n;
    const size_t size1 = m1.n;
    auto sampled = [&](long n) { return ratio >= 100.0 || (double)random_array[n % LHGT_MAX_RANDOM] < ratio; };
    for (long g = g0; g < g1; g++) {
        k1.next(&a, &la, &sa);
        const bool have2 = k2.next(&b, &lb, &sb);
        if (!have2) { b = lay.stale; lb = lay.stale_len; sb = m2.n; }   // fq2 has run out (E:356-367)
        if (g % 4 != 1) continue;
        const long n = g / 4;
        uint8_t fl;
        if (emu) {   // each mate by its own file's thread chunks; phase C follows fq1's (E:350-359)
            fl = (uint8_t)((emu->f1.keep(g, ratio, random_array) == 1 ? PAIR_COUNT1 | PAIR_VOTE : 0) |
                           (have2 && emu->f2.keep(g + lay.shift, ratio, random_array) == 1 ? PAIR_COUNT2 : 0));
        } else {
            // phase A samples every file by its own read ordinal (E:1037-1044): with a shift mate 2 is fq2's read n + shift / 4.
            // Quirk Q4: mate 2 is counted only while its line starts at <= size(fq1) (E:1419-1445)
            fl = (uint8_t)((sampled(n) ? PAIR_COUNT1 | PAIR_VOTE : 0) |
                           (have2 && sb <= size1 && sampled((g + lay.shift) / 4) ? PAIR_COUNT2 : 0));
        }
        if (!fl || (n / shard_block) % shard_world != shard_rank) continue;
        // a mate no phase reads is kept empty: the reference does not touch such a line either -- its buffers are filled under
        // `r < down_sam_ratio` only (E:1044, 404) -- so it may be longer than they are
        if (!(fl & (PAIR_COUNT1 | PAIR_VOTE))) la = 0;
        if (!(fl & (PAIR_COUNT2 | PAIR_VOTE))) lb = 0;
        if (la > LHGT_MAX_READ_LEN || lb > LHGT_MAX_READ_LEN) {
            out->rc = LHGT_E_FORMAT;
            out->err = "read " + std::to_string(n) + " longer than " + std::to_string(LHGT_MAX_READ_LEN) + " bases (the reference's buffers, E:1004)";
            return;
        }
        out->push(a, la, b, lb, fl);
    }
    out->finish();
}

// records of fq2 that phase C pairs with no line of fq1 -- those in front of the shift and those behind fq1's last line -- are
// still read by phase A (E:1426-1448): mate-2-only entries, lines [gA, gB) of fq2
static int parse_fq2_only(const Mapped& m1, const Mapped& m2, const ChunkPlan& p2, long gA, long gB, double ratio, const float* random_array,
                          int shard_rank, int shard_world, long shard_block, const ThreadEmu* emu, ParsedChunk* tail) {
    tail->o1.assign(1, 0);
    tail->o2.assign(1, 0);
    if (gA >= gB) return LHGT_OK;
    long c2 = (long)(std::upper_bound(p2.line0.begin(), p2.line0.end(), gA) - p2.line0.begin()) - 1;
    LineCursor k2(m2);
    k2.cur = p2.start[c2];
    const uint8_t* b;
    size_t lb, sb;
    for (long skip = gA - p2.line0[c2]; skip > 0; skip--) k2.next(&b, &lb, &sb);
    for (long g = gA; g < gB && k2.next(&b, &lb, &sb); g++) {
        if (sb > m1.n) break;
        if (g % 4 != 1) continue;
        const long n = g / 4;
        const bool keep = emu ? emu->f2.keep(g, ratio, random_array) == 1 : (ratio >= 100.0 || (double)random_array[n % LHGT_MAX_RANDOM] < ratio);
        if (!keep || (n / shard_block) % shard_world != shard_rank) continue;
        if (lb > LHGT_MAX_READ_LEN) LHGT_FAIL(LHGT_E_FORMAT, "read %ld longer than %d bases (the reference's buffers, E:1004)", n, LHGT_MAX_READ_LEN);
        tail->s2.insert(tail->s2.end(), b, b + lb);
        tail->o1.push_back(0);
        tail->o2.push_back(tail->s2.size());
        tail->flags.push_back(PAIR_COUNT2);
    }
    return LHGT_OK;
}


// everything the parse needs to know before the first chunk: plans, thread chunks, pairing layout.  Returns LHGT_E_FORMAT with
// a message that starts with "-t N emulation" where only the emulation refuses the input (the caller may fall back to -t 1).
static int parse_setup(const Mapped& m1, const Mapped& m2, const ChunkPlan& p1, const ChunkPlan& p2, const char* fq1, const char* fq2,
                       int emulate_threads, ThreadEmu* emu_store, PairLayout* lay) {
    lay->lines1 = p1.line0.back();
    lay->lines2 = p2.line0.back();
    lay->shift = 0;
    lay->stale = m2.p;
    lay->stale_len = 0;
    if (m2.n && m2.p[m2.n - 1] != '\n') {   // std::getline hits EOF inside the last line: the next call fails before it clears the string
        const uint8_t* s;
        size_t len, st;
        if (line_at(m2, p2, lay->lines2 - 1, &s, &len, &st)) { lay->stale = s; lay->stale_len = len; }
    }
    const uint8_t *a, *b;
    size_t la, lb, sa, sb;
    if (emulate_threads > 1) {
        LHGT_TRY(thread_part(m1, p1, (long)m1.n, emulate_threads, fq1, &emu_store->f1, &emu_store->pos1));
        LHGT_TRY(thread_part(m2, p2, (long)m1.n, emulate_threads, fq2, &emu_store->f2));
        // phase C: every thread seeks fq2 to the BYTE it entered fq1 at and compares the IDs of the first lines; when they differ
        // it rewinds fq2 by 10^9 bytes (to byte 1 at most) and reads on until a line carries fq1's ID (E:350-397).  One shift for
        // all threads is what this loader can express -- anything else is refused
        bool have = false;
        for (int i = 0; i < emulate_threads; i++) {
            const long g = emu_store->f1.first[i];
            if (g >= lay->lines1 || emu_store->f1.count[i] == 0) continue;
            if (!line_at(m1, p1, g, &a, &la, &sa)) continue;
            const size_t pos = (size_t)emu_store->pos1[i];
            long g2 = -1;
            if (pos < m2.n) {
                const uint8_t* nl = (const uint8_t*)memchr(m2.p + pos, '\n', m2.n - pos);
                if (same_id(a, la, m2.p + pos, nl ? (size_t)(nl - (m2.p + pos)) : m2.n - pos)) g2 = line_index_of(m2, p2, pos);
            }
            if (g2 < 0) g2 = find_id_line(m2, p2, pos > 1000000001UL ? pos - 1000000000UL : 1, a, read_id_len(a, la));
            if (g2 < 0) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: no line of %s carries the read ID of thread %d's first record (the reference spins through 10^9 failed reads)", emulate_threads, fq2, i);
            if (have && g2 - g != lay->shift) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: the threads find their first records at different line offsets in %s (%ld and %ld)", emulate_threads, fq2, lay->shift, g2 - g);
            lay->shift = g2 - g;
            have = true;
        }
        if (lay->lines2 < lay->lines1 + lay->shift)
            for (long g = lay->lines2 - lay->shift; g < lay->lines1; g++)
                if (g % 4 == 1) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: %s has fewer records than %s", emulate_threads, fq2, fq1);
    } else if (lay->lines1 > 0) {
        line_at(m1, p1, 0, &a, &la, &sa);
        const bool have2 = line_at(m2, p2, 0, &b, &lb, &sb);
        if (!have2 || !same_id(a, la, b, lb)) {
            const long g2 = find_id_line(m2, p2, 1, a, read_id_len(a, la));
            if (g2 < 0) LHGT_FAIL(LHGT_E_FORMAT, "paired-end reads not consistent: no line of %s carries the first read ID of %s (the reference spins through 10^9 failed reads)", fq2, fq1);
            lay->shift = g2;
        }
    }
    if (lay->shift % 4) LHGT_FAIL(emulate_threads > 1 ? LHGT_E_EMULATION : LHGT_E_FORMAT, "%s%s pairs with %s at line offset %ld, inside a record (the reference would take quality lines for reads)",
                                  emulate_threads > 1 ? "-t N emulation: " : "", fq2, fq1, lay->shift);
    if (lay->shift < 0) LHGT_FAIL(LHGT_E_EMULATION, "-t %d emulation: %s is %ld lines behind %s", emulate_threads, fq2, -lay->shift, fq1);
    return LHGT_OK;
}

// consume(chunk) is called on the calling thread, in file order, while the worker threads parse ahead (at most `pool` slabs, or
// 2 x threads chunks, in flight).  A chunk that used a slab keeps it until the consumer hands it back (pool->release), which it
// does once its copy to the device has completed; idle(true/false) is called while the consumer waits for the next chunk so it
// can poll for finished copies (and must free at least one slab when asked to block and every slab is out).
// which way the last FASTQ parse of this thread went (lhgt_ingest_last_path): 1 = the single pass, 0 = the two planned passes, and why
static thread_local int g_last_path = -1;
static thread_local std::string g_last_path_why;

// `reset` (nullable): drops everything consume() has been given so far.  With it, and without plans made elsewhere, the files first
// go through the single-pass loader (host_fastq_stream.cpp); where that pass does not decide the input, the pairs it delivered
// are dropped and the two passes below run as if it had not been tried.  LHGT_INGEST_STREAM=0 turns the first attempt off.
template <class Consume, class Idle>
static int parse_pairs(const char* fq1, const char* fq2, double ratio, const float* random_array, int shard_rank, int shard_world,
                       long shard_block, int threads, size_t chunk_bytes, int emulate_threads, long* n_pairs_seen, SlabPool* pool_in,
                       Consume consume, Idle idle, const std::function<int(SlabPool**)>& prepare = nullptr,
                       const std::function<int(const Mapped&, const Mapped&, ParseShare*, ChunkPlan*, ChunkPlan*)>& share_fn = nullptr,
                       long sampling_filled = LHGT_MAX_RANDOM, const std::function<int()>& reset = nullptr) {
    Mapped m1, m2;
    LHGT_TRY(m1.open(fq1));
    LHGT_TRY(m2.open(fq2));
    if (threads < 1) threads = 1;
    auto sampling_covers = [&](long lines1, long lines2) -> int {   // a sampling array filled for fewer reads than the files hold (lhgt_sampling_reserve) must fail loudly
        if (ratio < 100.0 && sampling_filled < LHGT_MAX_RANDOM && (lines1 + 2) / 4 > sampling_filled)
            LHGT_FAIL(LHGT_E_STATE, "the sampling array was filled for %ld reads (lhgt_sampling_reserve), %s holds %ld", sampling_filled, fq1, (lines1 + 2) / 4);
        if (ratio < 100.0 && sampling_filled < LHGT_MAX_RANDOM && (lines2 + 2) / 4 > sampling_filled)
            LHGT_FAIL(LHGT_E_STATE, "the sampling array was filled for %ld reads (lhgt_sampling_reserve), %s holds %ld", sampling_filled, fq2, (lines2 + 2) / 4);
        return LHGT_OK;
    };
    const bool stream_on = !(getenv("LHGT_INGEST_STREAM") && atoi(getenv("LHGT_INGEST_STREAM")) == 0);
    g_last_path = 0;
    g_last_path_why = share_fn ? "plans made elsewhere" : !reset ? "the caller cannot take pairs back" : !stream_on ? "LHGT_INGEST_STREAM=0" : "";
    if (reset && !share_fn && stream_on) {
        ChunkPlan q1, q2;
        std::string why;
        const int src = parse_pairs_stream(m1, m2, fq1, fq2, ratio, random_array, shard_rank, shard_world, shard_block, threads, chunk_bytes,
                                           emulate_threads, prepare, std::function<int(ParsedChunk&)>(consume), std::function<void(bool)>(idle), &q1, &q2, &why);
        if (src == LHGT_OK) {
            const long lines1 = q1.line0.back(), lines2 = q2.line0.back();
            // fq2 longer than fq1: its surplus records are counted by phase A (quirk Q4), as below
            if (lines2 > lines1) {
                ParsedChunk tail;
                const ThreadEmu* no_emu = nullptr;
                ThreadEmu emu_store;
                if (emulate_threads > 1) {               // the surplus is kept by fq2's thread chunks: by line numbers, now that the plans exist
                    LHGT_TRY(thread_part(m1, q1, (long)m1.n, emulate_threads, fq1, &emu_store.f1, &emu_store.pos1));
                    LHGT_TRY(thread_part(m2, q2, (long)m1.n, emulate_threads, fq2, &emu_store.f2));
                    no_emu = &emu_store;
                }
                LHGT_TRY(parse_fq2_only(m1, m2, q2, lines1, lines2, ratio, random_array, shard_rank, shard_world, shard_block, no_emu, &tail));
                if (tail.o1.size() > 1) LHGT_TRY(consume(tail));
            }
            if (n_pairs_seen) *n_pairs_seen = (lines1 + 2) / 4;
            g_last_path = 1;
            return sampling_covers(lines1, lines2);
        }
        if (src != STREAM_RETRY) return src;
        g_last_path_why = why;
        if (ingest_trace()) fprintf(stderr, "[lhgt ingest] the single pass leaves %s to the planned loader: %s\n", fq1, why.c_str());
        LHGT_TRY(reset());
    }
    double t0 = now_s();
    // the line count runs on helper threads; meanwhile the calling thread may allocate (prepare: pinned slabs, device staging)
    ChunkPlan p1s, p2s;
    ParseShare share;
    SlabPool* pool = pool_in;
    {
        int src = LHGT_OK;
        std::string serr;
        std::thread planner([&] {
            if (share_fn) { src = share_fn(m1, m2, &share, &p1s, &p2s); if (src != LHGT_OK) serr = last_error(); }
            else {   // both files at once; counting newlines is bound by memory bandwidth, which more threads than the parse uses still raise
                // (measured on 2 x 10 GB in the page cache, 256 cores: 16-24 threads per file count in 0.08-0.14 s, 48 in 0.17-0.3, 96 in
                // 0.3-0.45 -- first-touch faults of the mappings contend on the address space's locks)
                const int pt = threads > 24 ? 24 : threads;
                std::thread t2([&] { p2s = plan_chunks(m2, chunk_bytes, pt); });
                p1s = plan_chunks(m1, chunk_bytes, pt);
                t2.join();
            }
        });
        const int prc = prepare ? prepare(&pool) : LHGT_OK;
        planner.join();
        LHGT_TRY(prc);
        if (src != LHGT_OK) LHGT_FAIL(src, "%s", serr.c_str());
    }
    const ChunkPlan& p1 = share.p1 ? *share.p1 : p1s;
    const ChunkPlan& p2 = share.p2 ? *share.p2 : p2s;
    double t_plan = now_s() - t0, t_parse = 0, t_consume = 0;
    ThreadEmu emu_store;
    PairLayout lay;
    LHGT_TRY(parse_setup(m1, m2, p1, p2, fq1, fq2, emulate_threads, &emu_store, &lay));
    const ThreadEmu* emu = emulate_threads > 1 ? &emu_store : nullptr;
    const long nc_all = (long)p1.start.size() - 1;
    const long c_lo = nc_all * share.part / share.n_parts, c_hi = nc_all * (share.part + 1) / share.n_parts, nc = c_hi - c_lo;
    // fq2's records in front of the shift: counted by phase A, paired with nothing (the rank that owns fq1's first chunk)
    if (lay.shift > 0 && share.part == 0) {
        ParsedChunk head;
        LHGT_TRY(parse_fq2_only(m1, m2, p2, 0, lay.shift < lay.lines2 ? lay.shift : lay.lines2, ratio, random_array, shard_rank, shard_world, shard_block, emu, &head));
        if (head.o1.size() > 1) LHGT_TRY(consume(head));
    }
    // Plans that lie on the column grid of the single-pass loader (fq1 cut at chunk_bytes, fq2 into as many chunks: what
    // lhgt_fastq_pair_chunk_bytes tells the planner) let this part's columns be read with pread and cut from SIMD newline lists
    // like there, all at once -- their line numbers are known.  Same pairs, same part of the chunks; where a column does not see
    // what its plan says (or the files drift apart) the pairs are taken back and the loop below parses the part.
    bool parsed = false;
    if (share.raw_start1 && reset && stream_on && lay.shift == 0 && nc_all > 0) {
        size_t ch1 = 0, ch2 = 0;
        if (stream_chunking(m1.n, m2.n, chunk_bytes, &ch1, &ch2) && ch1 == chunk_bytes && share.raw_n1 == (long)n_plan_chunks(m1.n, ch1) &&
            share.raw_n2 == (long)n_plan_chunks(m2.n, ch2)) {
            const long ncols = share.raw_n1;
            std::vector<long> P1((size_t)ncols + 1, 0), P2((size_t)ncols + 1, 0), ne;     // ne: columns whose fq1 chunk holds a byte (the chunks of p1)
            std::vector<uint64_t> S1((size_t)ncols, m1.n), S2((size_t)ncols, m2.n);
            for (long c = 0; c < ncols; c++) {
                P1[(size_t)c + 1] = P1[(size_t)c] + share.raw_count1[c];
                P2[(size_t)c + 1] = P2[(size_t)c] + (c < share.raw_n2 ? share.raw_count2[c] : 0);
                S1[(size_t)c] = share.raw_start1[c];
                if (c < share.raw_n2) S2[(size_t)c] = share.raw_start2[c];
                if ((c + 1 < ncols ? share.raw_start1[c + 1] : (uint64_t)m1.n) > share.raw_start1[c]) ne.push_back(c);
            }
            if ((long)ne.size() == nc_all && P1[(size_t)ncols] == lay.lines1 && P2[(size_t)ncols] == lay.lines2) {
                StreamSeeds seeds;
                seeds.P1 = P1.data(); seeds.P2 = P2.data(); seeds.start1 = S1.data(); seeds.start2 = S2.data();
                seeds.ncols = ncols;
                seeds.col_lo = c_lo < nc_all ? ne[(size_t)c_lo] : ncols;
                seeds.col_hi = c_hi < nc_all ? ne[(size_t)c_hi] : ncols;
                seeds.emu = emu;
                seeds.stale = lay.stale; seeds.stale_len = lay.stale_len;
                ChunkPlan q1, q2;
                std::string why;
                const int src = parse_pairs_stream(m1, m2, fq1, fq2, ratio, random_array, shard_rank, shard_world, shard_block, threads, chunk_bytes,
                                                   emulate_threads, [&](SlabPool** out) -> int { *out = pool; return LHGT_OK; },
                                                   std::function<int(ParsedChunk&)>(consume), std::function<void(bool)>(idle), &q1, &q2, &why, &seeds);
                if (src == LHGT_OK) { parsed = true; g_last_path = 2; }
                else if (src != STREAM_RETRY) return src;
                else {
                    g_last_path_why = why;
                    if (ingest_trace()) fprintf(stderr, "[lhgt ingest] part %d/%d: the columns leave the part to the chunk loop: %s\n", share.part, share.n_parts, why.c_str());
                    LHGT_TRY(reset());
                }
            }
        }
    }
    if (!parsed) {
        std::vector<ParsedChunk> out((size_t)nc);
        std::vector<std::atomic<int>> ready((size_t)nc);
        for (auto& r : ready) r.store(0);
        std::atomic<long> next{0}, in_flight{0};
        std::atomic<bool> stop{false};
        std::mutex mu;
        std::condition_variable cv_ready, cv_room;
        const long max_in_flight = pool ? (long)1 << 30 : 2L * threads;   // with a pool the slabs bound the look-ahead
        auto worker = [&]() {
            for (;;) {
                int slab_id = -1;
                if (pool) {                                    // before the index: chunks are handed out in file order, so
                    slab_id = pool->acquire();                 // every earlier chunk already holds its slab -> no deadlock
                    if (slab_id < 0) return;
                } else {
                    std::unique_lock<std::mutex> lk(mu);
                    cv_room.wait(lk, [&] { return in_flight.load() < max_in_flight || stop.load(); });
                }
                const long c = next.fetch_add(1);
                if (c >= nc || stop.load()) { if (pool && slab_id >= 0) pool->release(slab_id); return; }
                in_flight.fetch_add(1);
                ParsedChunk& ch = out[(size_t)c];
                if (pool) ch.use_slab(pool->base + (size_t)slab_id * pool->slab_bytes, pool->half_bytes, slab_id, pool->k);
                parse_chunk(m1, m2, p1, p2, c_lo + c, ratio, random_array, shard_rank, shard_world, shard_block, emu, lay, &ch);
                { std::lock_guard<std::mutex> lk(mu); ready[(size_t)c].store(1); }
                cv_ready.notify_all();
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)(nc < threads ? nc : threads);
        for (int w = 0; w < nt; w++) th.emplace_back(worker);
        int rc = LHGT_OK;
        std::string err;
        long n_consumed = 0;
        for (long c = 0; c < nc && rc == LHGT_OK; c++) {
            double t1 = now_s();
            {
                std::unique_lock<std::mutex> lk(mu);
                while (!ready[(size_t)c].load()) {
                    lk.unlock();
                    idle(false);
                    lk.lock();
                    if (ready[(size_t)c].load()) break;
                    if (cv_ready.wait_for(lk, std::chrono::microseconds(200)) == std::cv_status::timeout && pool) {
                        lk.unlock();
                        idle(true);    // every slab may be out with its copy pending: wait for the oldest copy
                        lk.lock();
                    }
                }
            }
            double t2 = now_s();
            ParsedChunk& ch = out[(size_t)c];
            if (ch.rc != LHGT_OK) { rc = ch.rc; err = ch.err; }
            else {
                if (ch.slab_id >= 0 && !ch.slab) { pool->release(ch.slab_id); ch.slab_id = -1; }   // spilled to vectors: the slab is free
                rc = consume(ch);         // owns ch.slab_id from here (hands it back when its copy is done)
                if (rc != LHGT_OK) err = last_error();
                ch.slab_id = -1;
            }
            n_consumed = c + 1;
            ParsedChunk().s1.swap(ch.s1);   // free the chunk's vectors now
            ParsedChunk().s2.swap(ch.s2);
            ParsedChunk().o1.swap(ch.o1);
            ParsedChunk().o2.swap(ch.o2);
            in_flight.fetch_sub(1);
            cv_room.notify_one();
            t_parse += t2 - t1;
            t_consume += now_s() - t2;
        }
        stop.store(true);
        next.store(nc);
        cv_room.notify_all();
        if (pool) pool->close();          // workers waiting for a slab give up
        for (auto& t : th) t.join();
        if (pool) {                       // slabs of chunks that were parsed but never consumed (only after an error)
            for (long c = n_consumed; c < nc; c++)
                if (out[(size_t)c].slab_id >= 0) pool->release(out[(size_t)c].slab_id);
            pool->reopen();
        }
        if (rc != LHGT_OK) LHGT_FAIL(rc, "%s", err.c_str());
    }
    // fq2 longer than fq1: phase C stops with fq1 (E:356), but phase A counts every record of fq2 whose sequence line starts
    // at a byte offset <= size(fq1) (E:1419-1445, quirk Q4) -- surplus records become mate-2-only entries (the rank that owns fq1's last chunk)
    if (lay.lines2 > lay.lines1 + lay.shift && share.part == share.n_parts - 1) {
        ParsedChunk tail;
        LHGT_TRY(parse_fq2_only(m1, m2, p2, lay.lines1 + lay.shift, lay.lines2, ratio, random_array, shard_rank, shard_world, shard_block, emu, &tail));
        if (tail.o1.size() > 1) LHGT_TRY(consume(tail));
    }
    if (ingest_trace())
        fprintf(stderr, "[lhgt ingest] part %d/%d: %d threads, chunks [%ld, %ld) of %ld = %.1f MB of %.1f MB of %s: line count %.3fs, parse %.3fs, consume(+upload) %.3fs\n",
                share.part, share.n_parts, threads, c_lo, c_hi, nc_all, 1e-6 * (double)(p1.start[(size_t)c_hi] - p1.start[(size_t)c_lo]), 1e-6 * (double)m1.n, fq1,
                t_plan, t_parse, t_consume);
    if (n_pairs_seen) *n_pairs_seen = (p1.line0.back() + 2) / 4;   // lines with index % 4 == 1
    return sampling_covers(p1.line0.back(), p2.line0.back());
}

// CPUs this process may use at once: the hardware threads, cut to the cgroup's quota where one is set (a container that is given
// 16 CPUs of a 256-thread host sees 256 in /proc; threads beyond the quota only get the whole group throttled)
static int usable_cpus() {
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc == 0 ? 4 : (int)hc;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    for (const char* path : {"/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"}) {
        FILE* f = fopen(path, "r");
        if (!f) continue;
        char a[64] = {0}, b[64] = {0};
        const int got = fscanf(f, "%63s %63s", a, b);
        fclose(f);
        long quota = got >= 1 && strcmp(a, "max") ? atol(a) : -1, period = got >= 2 ? atol(b) : 100000;
        if (got == 1) {                                  // cgroup v1: the period lives in its own file
            FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (g) { if (fscanf(g, "%ld", &period) != 1) period = 100000; fclose(g); }
        }
        if (quota > 0 && period > 0) { const int q = (int)((quota + period - 1) / period); if (q >= 1 && q < n) n = q; }
        break;
    }
    return n;
}

static int default_threads();
int ingest_default_threads() { return default_threads(); }     // (k_packed.hip)
static int default_threads() {
    const char* e = getenv("LHGT_INGEST_THREADS");
    if (e && atoi(e) > 0) return atoi(e);
    static const int n = usable_cpus();
    return n > 48 ? 48 : n;   // the parse is bound by host memory bandwidth well before that
}

// ---------------------------------------------------------------- FASTA by line structure, bases never touched on the host
// read_ref (E:761-880) reads the file with std::getline: a line that starts with '>' opens the next sequence, every other line
// is appended to the open one (so its bases are the bytes of its lines without the '\n'; a '\r' stays in).  The same sequences
// fall out of the positions of the '>' lines and the newline counts between them -- found by memchr and SSE2 compares on all
// cores, block by block -- and the bases themselves go to the GPU as file text (strip_fasta_block, k_ingest.hip).
struct FastaSeq {
    uint64_t text_begin = 0, text_end = 0;   // its lines in the file, newlines included
    uint64_t len = 0;                        // its bases
    uint64_t name_at = 0;                    // header text after '>' up to get_read_ID's cut (E:303-311); sequence 0 is called "start"
    uint32_t name_len = 0;
};
struct FastaIndex {
    std::vector<FastaSeq> seqs;              // seqs[i] follows the i-th '>' line; seqs[0] = the lines before the first one
    std::vector<uint64_t> nl_before;         // newlines before each FASTA_BLK-byte block of the file, and the total at the end
    uint64_t nl_in(const uint8_t* p, uint64_t a, uint64_t b) const {
        if (a >= b) return 0;
        const uint64_t ba = a / FASTA_BLK, bb = b / FASTA_BLK;
        if (ba == bb) return (uint64_t)count_nl(p + a, p + b);
        return (uint64_t)count_nl(p + a, p + (ba + 1) * FASTA_BLK) + (nl_before[bb] - nl_before[ba + 1]) + (uint64_t)count_nl(p + bb * FASTA_BLK, p + b);
    }
    std::string name(const uint8_t* p, size_t i) const { return i == 0 ? std::string("start") : std::string((const char*)p + seqs[i].name_at, seqs[i].name_len); }
};

static void fasta_scan(const Mapped& fa, int threads, FastaIndex* fx) {
    const uint8_t* p = fa.p;
    const uint64_t n = fa.n;
    const uint64_t n_blk = (n + FASTA_BLK - 1) / FASTA_BLK;
    const uint64_t CH = (uint64_t)1024 * FASTA_BLK;   // 4 MiB per task
    const long n_ch = (long)((n + CH - 1) / CH);
    std::vector<uint32_t> nl(n_blk);
    std::vector<std::vector<uint64_t>> hdr((size_t)n_ch);
    parallel_for(n_ch, threads, [&](long c) {
        const uint64_t b0 = (uint64_t)c * CH, b1 = b0 + CH < n ? b0 + CH : n;
        for (uint64_t x = b0; x < b1; x += FASTA_BLK) nl[x / FASTA_BLK] = (uint32_t)count_nl(p + x, p + (x + FASTA_BLK < b1 ? x + FASTA_BLK : b1));
        for (const uint8_t* q = p + b0; q < p + b1; q++) {   // '>' is rare outside header lines: memchr runs at memory speed
            q = (const uint8_t*)memchr(q, '>', (size_t)(p + b1 - q));
            if (!q) break;
            const uint64_t at = (uint64_t)(q - p);
            if (at == 0 || p[at - 1] == '\n') hdr[(size_t)c].push_back(at);
        }
    });
    fx->nl_before.assign(n_blk + 1, 0);
    for (uint64_t b = 0; b < n_blk; b++) fx->nl_before[b + 1] = fx->nl_before[b] + nl[b];
    fx->seqs.assign(1, FastaSeq());
    for (const auto& v : hdr)
        for (uint64_t h : v) {
            const uint8_t* e = (const uint8_t*)memchr(p + h, '\n', (size_t)(n - h));
            const uint64_t he = e ? (uint64_t)(e - p) : n;
            fx->seqs.back().text_end = h;
            FastaSeq q;
            q.text_begin = he < n ? he + 1 : n;
            const size_t idl = read_id_len(p + h, (size_t)(he - h));
            q.name_at = h + (idl ? 1 : 0);
            q.name_len = (uint32_t)(idl ? idl - 1 : 0);
            fx->seqs.push_back(q);
        }
    fx->seqs.back().text_end = n;
    parallel_for((long)fx->seqs.size(), threads, [&](long i) {
        FastaSeq& q = fx->seqs[(size_t)i];
        q.len = (q.text_end - q.text_begin) - fx->nl_in(p, q.text_begin, q.text_end);
    });
}

// Spans of consecutive sequences, about span_bases bases each (one long sequence is a span of its own): the span's text is
// copied to the GPU as it is in the file, stripped of header lines and newlines there, and fn sees its bases back to back in
// device memory: fn(d_bases, n_bases, coff[0..n_c], first sequence, n_c) with sequence first + c at [coff[c], coff[c+1]).
template <class Fn>
static int fasta_spans(lhgt_ctx* ctx, const Mapped& fa, const FastaIndex& fx, uint64_t span_bases, Fn fn) {
    const uint8_t* p = fa.p;
    const size_t ns = fx.seqs.size();
    struct Span { size_t s0, s1; uint64_t bases, A, B; };
    std::vector<Span> spans;
    for (size_t s0 = 0; s0 < ns;) {
        size_t s1 = s0;
        uint64_t bases = 0;
        while (s1 < ns && (s1 == s0 || bases + fx.seqs[s1].len <= span_bases)) bases += fx.seqs[s1++].len;
        if (bases) spans.push_back({s0, s1, bases, fx.seqs[s0].text_begin & ~(uint64_t)(FASTA_BLK - 1), fx.seqs[s1 - 1].text_end});
        s0 = s1;
    }
    // Page-locking a span of the mapping costs about as much as copying it (~25 ms per GiB each, tools/h2d_rates.hip), so the
    // NEXT span is locked on a helper thread while this one is copied, stripped and consumed, and the one before the previous one is
    // released there as well.  Locked ranges must not overlap: a span is locked from the first page boundary at or after the end of its
    // predecessor's range; the few bytes before it travel as a small pageable copy.
    const uint64_t PAGE = 4096;
    std::vector<void*> locked(spans.size(), nullptr);
    std::vector<uint64_t> lock_from(spans.size(), 0);
    uint64_t prev_end = 0;
    for (size_t i = 0; i < spans.size(); i++) {
        lock_from[i] = spans[i].A > prev_end ? spans[i].A : prev_end;
        prev_end = (spans[i].B + PAGE - 1) & ~(PAGE - 1);
    }
    auto lock = [&](size_t i) {
        const uint64_t to = (spans[i].B + PAGE - 1) & ~(PAGE - 1);   // inside the mapping: it ends at a page boundary
        if (lock_from[i] >= to || to - lock_from[i] < ((uint64_t)1 << 20)) return;
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return; }
        void* q = (void*)(p + lock_from[i]);
        if (hipHostRegister(q, (size_t)(to - lock_from[i]), hipHostRegisterDefault) == hipSuccess) locked[i] = q;
        else (void)hipGetLastError();      // the runtime refuses some mappings: the copy is a pageable one then
    };
    auto unlock = [&](size_t i) { if (locked[i]) { (void)hipHostUnregister(locked[i]); locked[i] = nullptr; } };
    std::vector<uint64_t> kept, seg, coff;
    std::thread helper;
    int rc = LHGT_OK;
    if (!spans.empty()) lock(0);
    for (size_t i = 0; i < spans.size() && rc == LHGT_OK; i++) {
        if (helper.joinable()) helper.join();
        // span i - 1 stays locked while span i is copied: span i's text starts inside the last page of that range, and the runtime must
        // not be asked to copy from a registered range while another thread unregisters it (seen once as an abort inside the
        // runtime: "pure virtual method called")
        helper = std::thread([&, i] { if (i > 1) unlock(i - 2); if (i + 1 < spans.size()) lock(i + 1); });
        const Span& sp = spans[i];
        const size_t s0 = sp.s0, s1 = sp.s1;
        const uint64_t A = sp.A, B = sp.B, bases = sp.bases;
        const uint64_t text_len = B - A, n_blocks = (text_len + FASTA_BLK - 1) / FASTA_BLK;
        kept.assign(n_blocks, 0);
        seg.assign(2 * (s1 - s0), 0);
        coff.assign(s1 - s0 + 1, 0);
        for (size_t s = s0; s < s1; s++) {
            seg[2 * (s - s0)] = fx.seqs[s].text_begin - A;
            seg[2 * (s - s0) + 1] = fx.seqs[s].text_end - A;
            coff[s - s0 + 1] = coff[s - s0] + fx.seqs[s].len;
        }
        uint64_t acc = 0;
        size_t sc = s0;
        for (uint64_t b = 0; b < n_blocks; b++) {
            kept[b] = acc;
            const uint64_t x0 = A + b * FASTA_BLK, x1 = x0 + FASTA_BLK < B ? x0 + FASTA_BLK : B;
            while (sc < s1 && fx.seqs[sc].text_end <= x0) sc++;
            for (size_t s = sc; s < s1 && fx.seqs[s].text_begin < x1; s++) {
                const uint64_t a = fx.seqs[s].text_begin > x0 ? fx.seqs[s].text_begin : x0, e = fx.seqs[s].text_end < x1 ? fx.seqs[s].text_end : x1;
                if (a >= e) continue;
                const bool whole = a == x0 && e == x0 + FASTA_BLK;
                acc += (e - a) - (whole ? fx.nl_before[x0 / FASTA_BLK + 1] - fx.nl_before[x0 / FASTA_BLK] : (uint64_t)count_nl(p + a, p + e));
            }
        }
        auto body = [&]() -> int {
            if (acc != bases) LHGT_FAIL(LHGT_E_STATE, "FASTA span: %llu bases by blocks, %llu by sequences", (unsigned long long)acc, (unsigned long long)bases);
            const size_t toff = ((size_t)bases + 32 + 255) & ~(size_t)255;   // bases at the front of the workspace, the text behind them
            LHGT_TRY(ws_reserve(ctx, toff + (size_t)text_len + 32, 0));
            // the part of the text in front of the locked range (less than a page, or all of it when nothing is locked), then the rest
            const uint64_t cut = locked[i] ? (lock_from[i] < B ? lock_from[i] : B) : B;
            if (cut > A) LHGT_HIP(hipMemcpyAsync(ctx->d_ws_ascii + toff, p + A, (size_t)(cut - A), hipMemcpyHostToDevice, ctx->stream));
            if (B > cut) LHGT_HIP(hipMemcpyAsync(ctx->d_ws_ascii + toff + (cut - A), p + cut, (size_t)(B - cut), hipMemcpyHostToDevice, ctx->stream));
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            LHGT_TRY(strip_fasta_text(ctx, ctx->d_ws_ascii + toff, text_len, kept.data(), (long)n_blocks, seg.data(), (long)(s1 - s0), ctx->d_ws_ascii));
            return fn((const uint8_t*)ctx->d_ws_ascii, (long)bases, coff.data(), s0, (long)(s1 - s0));
        };
        rc = body();
    }
    if (helper.joinable()) helper.join();
    for (size_t i = 0; i < spans.size(); i++) unlock(i);
    return rc;
}

}  // namespace lhgt

extern "C" {

// cal_sam_ratio (E:1244-1270) / E:1392-1398
int lhgt_fastq_sam_ratio(const char* fq1, double sample, double* ratio_percent, long* n_records) {
    if (!fq1 || !ratio_percent) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (sample <= 1) {
        *ratio_percent = 100 * sample;
        if (n_records) *n_records = -1;
        return LHGT_OK;
    }
    Mapped m;
    LHGT_TRY(m.open(fq1));
    // the reference's extra getline pass over fq1 (E:1244-1270), on all parse threads: every chunk sums its line lengths by
    // (local line index mod 4); which residue holds the sequence lines follows from the lines before the chunk
    long i = 0, bases = 0;
    {
        const size_t CH = (size_t)8 << 20;
        const size_t nchunks = m.n ? (m.n + CH - 1) / CH : 0;
        std::vector<size_t> st;
        for (size_t c = 0; c < nchunks; c++) {
            const size_t b = lhgt::line_start_at_or_after(m.p, m.n, c * CH);
            if (st.empty() || b > st.back()) st.push_back(b);
        }
        if (st.empty() || st.back() != m.n) st.push_back(m.n);
        const long nc = (long)st.size() - 1;
        std::vector<long> cnt((size_t)(nc > 0 ? nc : 0)), sums((size_t)(nc > 0 ? nc : 0) * 4, 0);
        lhgt::parallel_for(nc, lhgt::default_threads(), [&](long c) {
            lhgt::LineCursor lc(m);
            lc.cur = st[(size_t)c];
            const uint8_t* s;
            size_t len, start;
            long k = 0;
            while (lc.cur < st[(size_t)c + 1] && lc.next(&s, &len, &start)) { sums[(size_t)c * 4 + (k & 3)] += (long)len; k++; }
            cnt[(size_t)c] = k;
        });
        for (long c = 0; c < nc; c++) {
            bases += sums[(size_t)c * 4 + (size_t)(((1 - i) % 4 + 4) % 4)];
            i += cnt[(size_t)c];
        }
    }
    bases *= 2;
    *ratio_percent = 100 * sample / (double)bases;
    if (n_records) *n_records = i / 4;
    return LHGT_OK;
}


}  // extern "C"

void lhgt_ingest_pool_free(lhgt_ctx* ctx) {
    delete (lhgt::SlabPool*)ctx->ingest_pool;
    ctx->ingest_pool = nullptr;
}

extern "C" {

// The loader as a pipeline (the reference reads line by line on one thread per chunk, E:1020-1026, 356-367):
//   worker threads   parse 2 MiB chunks of fq1 against the same lines of fq2, writing kept bases into pinned slabs
//   calling thread   takes the chunks in file order: one asynchronous copy per mate from the slab into the device staging
//                    area, per-pair metadata (start / word offset / length / flags) into pinned arrays; a slab goes back to
//                    the workers when its copy has completed (event)
//   GPU              at >= 4 Mi pairs or 1 GiB of bases the batch is closed: metadata copied, pack_bases32 -> resident batch
// so parsing, PCIe and packing overlap, and nothing is copied twice on the host.
}  // extern "C"

using ShareFn = std::function<int(const lhgt::Mapped&, const lhgt::Mapped&, lhgt::ParseShare*, lhgt::ChunkPlan*, lhgt::ChunkPlan*)>;

// The GPU boxes are two-socket hosts that give a container a CPU QUOTA, not a CPU set (profiles/r05/host_cpu_quota.txt: 16 CPUs' worth
// of 256, affinity 0-255): the loader's threads float over both sockets, and with them the pinned slabs they first touch and the
// side of the link their copies start from.  While a load runs, the calling thread -- and the workers it creates, which inherit its
// mask -- keep to the CPUs of the NUMA node the GPU hangs on (sysfs: .../numa_node, local_cpulist of its PCI function).
// LHGT_INGEST_NUMA=off: leave the mask alone; =other: the other node(s) (A/B).
struct NodeAffinity {
    cpu_set_t old;
    bool on = false;
    explicit NodeAffinity(int device) {
        const std::string mode = getenv("LHGT_INGEST_NUMA") ? getenv("LHGT_INGEST_NUMA") : "gpu";
        if (mode == "off" || mode == "0") return;
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) { (void)hipGetLastError(); return; }
        for (char* c = bdf; *c; c++) *c = (char)tolower(*c);
        char path[256];
        snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
        FILE* f = fopen(path, "r");
        if (!f) return;
        char list[4096] = {0};
        const bool got = fgets(list, sizeof list, f) != nullptr;
        fclose(f);
        if (!got) return;
        cpu_set_t want, cur;
        CPU_ZERO(&want);
        for (char* tok = strtok(list, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
            int a = 0, b = 0;
            const int n = sscanf(tok, "%d-%d", &a, &b);
            if (n == 1) b = a;
            if (n >= 1) for (int c = a; c <= b && c < CPU_SETSIZE; c++) CPU_SET(c, &want);
        }
        if (sched_getaffinity(0, sizeof cur, &cur) != 0) return;
        cpu_set_t pick;
        CPU_ZERO(&pick);
        for (int c = 0; c < CPU_SETSIZE; c++)
            if (CPU_ISSET(c, &cur) && (CPU_ISSET(c, &want) != 0) == (mode != "other")) CPU_SET(c, &pick);
        if (CPU_COUNT(&pick) < 4 || CPU_COUNT(&pick) == CPU_COUNT(&cur)) return;       // nothing to choose from, or nothing to choose
        old = cur;
        on = sched_setaffinity(0, sizeof pick, &pick) == 0;
        if (on && ingest_trace()) fprintf(stderr, "[lhgt ingest] threads kept to %d CPUs %s the GPU's NUMA node (%s)\n", CPU_COUNT(&pick), mode == "other" ? "away from" : "of", bdf);
    }
    ~NodeAffinity() { if (on) sched_setaffinity(0, sizeof old, &old); }
};

static int load_fastq_impl(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent, int shard_rank,
                           int shard_world, long shard_block, long* n_pairs_seen, long* n_pairs_kept, const ShareFn& share_fn) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !fq1 || !fq2) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world || shard_block < 1)
        LHGT_FAIL(LHGT_E_ARG, "bad shard spec %d/%d block %ld", shard_rank, shard_world, shard_block);
    lhgt::sampling_join(ctx);
    if (ratio_percent < 100.0 && (long)ctx->random_array.size() != LHGT_MAX_RANDOM)
        LHGT_FAIL(LHGT_E_STATE, "lhgt_sampling_init(ratio) must precede lhgt_pairs_load_fastq when ratio < 100");
    NodeAffinity near_gpu(ctx->device);
    // a batch is closed at 4 Mi pairs (1 Mi with count-on-load, so that phase A of one batch hides behind the parsing of the next;
    // every batch costs phase A one sweep of the count table, which is why they are not smaller)
    // (round 5: with count-on-load the limits GROW -- 1, 2, 4 Mi pairs -- so that a small input still overlaps its count with its
    // parse while a large one pays 9 table sweeps per 32 M pairs instead of 32: at 1 Mi pairs per batch phase A on load took 0.2 s
    // per 32 M pairs, which the single-pass parse no longer hides)
    const long BATCH_PAIRS_MAX = 4L << 20, META_CAP = BATCH_PAIRS_MAX + (1L << 18);
    const size_t BATCH_BYTES_MAX = (size_t)1280 << 20, CHUNK = (size_t)lhgt_fastq_plan_chunk_bytes();
    long BATCH_PAIRS = ctx->count_on_load ? 1L << 20 : BATCH_PAIRS_MAX;
    const long batch_hook = getenv("LHGT_INGEST_BATCH_PAIRS") ? atol(getenv("LHGT_INGEST_BATCH_PAIRS")) : 0;   // test hook: small files close batches too
    if (batch_hook >= 64 && batch_hook < BATCH_PAIRS) BATCH_PAIRS = batch_hook;
    if (ctx->count_on_load) ctx->part_reserve_pairs = BATCH_PAIRS_MAX;     // phase A's key buffers: made once, for the largest batch
    size_t BATCH_BYTES = ctx->count_on_load ? (size_t)320 << 20 : BATCH_BYTES_MAX;
    const size_t HALF = CHUNK + 1024, SLAB = 2 * HALF + sizeof(ChunkPairMeta) * (CHUNK_META_CAP + 1);
    const long DESC_CAP = 1L << 16;                                       // chunks per batch
    // device staging of one batch: the chunks' blocks (bases + records) one after the other.  TWO of them: the copies of batch
    // b + 1 (on the copy stream) run while batch b is expanded, packed and counted (on the context's stream)
    const size_t STAGE = (BATCH_BYTES_MAX + 2 * SLAB + (size_t)META_CAP * sizeof(ChunkPairMeta) + 255) & ~(size_t)255;
    const int threads = default_threads();
    // slabs: one per parse thread plus what rides out the calling thread's pauses (a batch close allocates and launches; the first
    // one allocates phase A's key buffers): with 25 slabs for 16 threads the workers stood still for half of the load
    const int n_slabs = std::max(threads + threads / 3 + 4, (int)std::min<size_t>(4096, ((size_t)280 << 20) / SLAB));
    // staging, pinned slabs and the pinned chunk descriptors are allocated by `prepare` below, on this thread, while helper threads count lines
    SlabPool* pool = nullptr;
    ChunkDesc* desc_base = nullptr;
    double t_alloc = 0;
    std::vector<ChunkDesc> desc_pageable;     // only when the host refuses page-locked memory
    hipEvent_t buf_free[2] = {nullptr, nullptr}, copied = nullptr, copied2 = nullptr;
    hipStream_t cs2 = nullptr;         // a second copy stream: the chunks' copies alternate between two queues, so one copy's set-up hides behind the other's transfer
    auto prepare = [&](SlabPool** pool_out) -> int {
        const double t_a0 = now_s();
        LHGT_TRY(ws_reserve(ctx, 2 * STAGE, 0));
        // (prepare runs again when the single pass hands the files to the planned loader)
        for (auto& e : buf_free) if (!e) LHGT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));   // waits sleep: the host's CPUs are the parse threads' 
        if (!copied) LHGT_HIP(hipEventCreateWithFlags(&copied, hipEventDisableTiming));
        if (!copied2) LHGT_HIP(hipEventCreateWithFlags(&copied2, hipEventDisableTiming));
        if (!cs2 && !getenv("LHGT_ONE_COPY_STREAM")) LHGT_HIP(hipStreamCreateWithFlags(&cs2, hipStreamNonBlocking));
        pool = (SlabPool*)ctx->ingest_pool;
        if (!pool || pool->slab_bytes != SLAB || (int)ctx->ingest_events.size() != n_slabs || ctx->ingest_meta_cap != 2 * DESC_CAP) {
            lhgt_ingest_pool_free(ctx);
            ingest_free(ctx);
            pool = nullptr;
            if (getenv("LHGT_NO_PINNED") ||     // test hook for the fallback below
                hipHostMalloc(&ctx->h_ingest_slabs, (size_t)n_slabs * SLAB, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc(&ctx->h_ingest_meta, (size_t)2 * DESC_CAP * sizeof(ChunkDesc), hipHostMallocDefault) != hipSuccess) {
                // no page-locked memory to be had (a locked-memory limit): the same pipeline on pageable buffers -- chunks in
                // vectors, copied synchronously; slower, same result
                (void)hipGetLastError();
                ingest_free(ctx);
                desc_pageable.resize((size_t)2 * DESC_CAP);
                desc_base = desc_pageable.data();
                *pool_out = nullptr;
                t_alloc = now_s() - t_a0;
                return LHGT_OK;
            }
            ctx->ingest_meta_cap = 2 * DESC_CAP;
            pool = new SlabPool();
            pool->base = ctx->h_ingest_slabs;
            pool->slab_bytes = SLAB;
            pool->half_bytes = HALF;
            for (int i = 0; i < n_slabs; i++) pool->free_ids.push_back(i);
            ctx->ingest_pool = pool;
            ctx->ingest_events.resize((size_t)n_slabs);
            for (auto& e : ctx->ingest_events) LHGT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
        }
        pool->k = ctx->k;
        desc_base = (ChunkDesc*)ctx->h_ingest_meta;
        *pool_out = pool;
        t_alloc = now_s() - t_a0;
        return LHGT_OK;
    };
    const int k = ctx->k;
    hipStream_t cs = ctx->copy_stream;
    size_t fill = 0;
    long n_open = 0, kept = 0, n_desc = 0, n_batches = 0, n_long = 0;
    uint64_t words = 0, nkm = 0;
    int max_len = 0;
    bool used[2] = {false, false};
    std::vector<int> out_slabs;          // FIFO of slabs whose copies are in flight (event = ingest_events[slab id])
    size_t out_head = 0;
    auto reap = [&](bool block) {
        while (out_head < out_slabs.size()) {
            const int id = out_slabs[out_head];
            if (block) { (void)hipEventSynchronize(ctx->ingest_events[(size_t)id]); block = false; }
            else if (hipEventQuery(ctx->ingest_events[(size_t)id]) != hipSuccess) { (void)hipGetLastError(); break; }
            pool->release(id);
            out_head++;
        }
    };
    auto stage_base = [&]() { return ctx->d_ws_ascii + (size_t)(n_batches & 1) * STAGE; };
    auto desc = [&]() { return desc_base + (size_t)(n_batches & 1) * DESC_CAP; };
    // the open batch's staging buffer and descriptors were last used two batches ago: its pack kernel must have read them
    auto open_batch = [&]() -> int {
        const int bi = (int)(n_batches & 1);
        if (used[bi]) {
            LHGT_HIP(hipEventSynchronize(buf_free[bi]));           // host: the descriptors are rewritten
            LHGT_HIP(hipStreamWaitEvent(cs, buf_free[bi], 0));      // device: so is the staging buffer
            if (cs2) LHGT_HIP(hipStreamWaitEvent(cs2, buf_free[bi], 0));
        }
        return LHGT_OK;
    };
    auto stage_sync = [&](size_t off, const uint8_t* src, size_t bytes) -> int {   // pageable source (the rare chunks in vectors): copied before it goes away
        if (!bytes) return LHGT_OK;
        LHGT_HIP(hipMemcpyAsync(stage_base() + off, src, bytes, hipMemcpyHostToDevice, cs));
        LHGT_HIP(hipStreamSynchronize(cs));
        return LHGT_OK;
    };
    std::vector<hipEvent_t> count_ev;    // count-on-load: an event pair around every batch's phase A
    auto flush = [&]() -> int {
        if (n_open == 0) return LHGT_OK;
        const int bi = (int)(n_batches & 1);
        LHGT_HIP(hipEventRecord(copied, cs));
        LHGT_HIP(hipStreamWaitEvent(ctx->stream, copied, 0));
        if (cs2) {
            LHGT_HIP(hipEventRecord(copied2, cs2));
            LHGT_HIP(hipStreamWaitEvent(ctx->stream, copied2, 0));
        }
        int rc = install_pairs_chunked(ctx, stage_base(), (const ChunkPairMeta*)stage_base(), desc(), n_desc, n_open, words, max_len, nkm, n_long);
        LHGT_HIP(hipEventRecord(buf_free[bi], ctx->stream));
        used[bi] = true;
        if (rc == LHGT_OK && ctx->count_on_load) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            const bool have_e0 = hipEventCreate(&e0) == hipSuccess;
            if (have_e0 && hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); e0 = nullptr; }
            if (e0 && e1) {
                count_ev.push_back(e0);
                count_ev.push_back(e1);
                rc = lhgt_count_one_batch_async(ctx, ctx->batches.back(), e0, e1);   // runs on while the next batch is parsed and copied
            }
        }
        reap(false);
        fill = 0; n_open = 0; words = 0; nkm = 0; max_len = 0; n_desc = 0; n_long = 0;
        n_batches++;
        if (BATCH_PAIRS < BATCH_PAIRS_MAX) { BATCH_PAIRS *= 2; BATCH_BYTES = std::min(BATCH_BYTES * 2, BATCH_BYTES_MAX); }
        return rc;
    };
    std::vector<ChunkPairMeta> conv;     // records of a chunk that came in vectors (spilled, head / tail records of fq2, no pinned memory)
    // taking back what the single-pass attempt delivered (parse_pairs): batches installed by this call, and -- count-on-load --
    // what they added to the count table, which only a clear can undo: offered only while nothing else was counted before
    const size_t batches_at_entry = ctx->batches.size();
    const bool can_reset = !(ctx->count_on_load && ctx->counts_touched);
    auto reset = [&]() -> int {
        (void)hipStreamSynchronize(cs);
        if (cs2) (void)hipStreamSynchronize(cs2);
        (void)hipStreamSynchronize(ctx->stream);
        if (pool) while (out_head < out_slabs.size()) pool->release(out_slabs[out_head++]);
        bool counted = false;
        for (size_t i = batches_at_entry; i < ctx->batches.size(); i++) counted = counted || ctx->batches[i].counted;
        lhgt::pairs_truncate(ctx, batches_at_entry);
        if (counted) LHGT_TRY(lhgt_counts_clear(ctx));
        for (hipEvent_t e : count_ev) hipEventDestroy(e);
        count_ev.clear();
        fill = 0; n_open = 0; kept = 0; n_desc = 0; n_batches = 0; words = 0; nkm = 0; max_len = 0; n_long = 0;
        used[0] = used[1] = false;
        if (ctx->count_on_load) { BATCH_PAIRS = batch_hook >= 64 && batch_hook < (1L << 20) ? batch_hook : 1L << 20; BATCH_BYTES = (size_t)320 << 20; }
        return LHGT_OK;
    };
    int rc = parse_pairs(fq1, fq2, ratio_percent, ctx->random_array.data(), shard_rank, shard_world, shard_block, threads, CHUNK,
                         ctx->emu_threads, n_pairs_seen, (SlabPool*)nullptr,
                         [&](ParsedChunk& ch) -> int {
                             const long n = ch.n_pairs();
                             if (n <= 0) { if (ch.slab_id >= 0) pool->release(ch.slab_id); return LHGT_OK; }
                             const size_t b1n = ch.slab ? ch.n1 : ch.s1.size(), b2n = ch.slab ? 0 : ch.s2.size();
                             const size_t moff = (b1n + b2n + 15) & ~(size_t)15, block = moff + (size_t)(n + 1) * sizeof(ChunkPairMeta);
                             if (fill + block > STAGE || n_open + n > META_CAP || n_desc >= DESC_CAP) LHGT_TRY(flush());
                             if (block > STAGE || n > META_CAP) LHGT_FAIL(LHGT_E_FORMAT, "ingest: one chunk holds %ld pairs / %zu bases", n, b1n + b2n);
                             if (n_open == 0) LHGT_TRY(open_batch());
                             uint32_t ch_words = 0;
                             if (ch.slab) {
                                 hipStream_t q = cs2 && (n_desc & 1) ? cs2 : cs;
                                 LHGT_HIP(hipMemcpyAsync(stage_base() + fill, ch.slab, block, hipMemcpyHostToDevice, q));
                                 LHGT_HIP(hipEventRecord(ctx->ingest_events[(size_t)ch.slab_id], q));
                                 out_slabs.push_back(ch.slab_id);
                                 ch_words = ch.words;
                                 if (ch.max_len > max_len) max_len = ch.max_len;
                                 nkm += ch.nkm;
                                 n_long += ch.n_long;
                             } else {                      // a chunk in pageable vectors: its records are made here, everything copied before it goes away
                                 conv.resize((size_t)n + 1);
                                 for (long i = 0; i < n; i++) {
                                     const uint32_t l1 = (uint32_t)(ch.o1[i + 1] - ch.o1[i]), l2 = (uint32_t)(ch.o2[i + 1] - ch.o2[i]);
                                     conv[(size_t)i] = ChunkPairMeta{(uint32_t)ch.o1[i], (uint32_t)ch.o2[i], ch_words, ch.flags[(size_t)i]};
                                     ch_words += 3 * ((l1 + 31) / 32 + 1) + 3 * ((l2 + 31) / 32 + 1);
                                     if ((int)l1 > max_len) max_len = (int)l1;
                                     if ((int)l2 > max_len) max_len = (int)l2;
                                     if ((int)l1 >= k) nkm += l1 - k + 1;
                                     if ((int)l2 >= k) nkm += l2 - k + 1;
                                     n_long += ((int)l1 - k + 1 > FAST_NK) + ((int)l2 - k + 1 > FAST_NK);
                                 }
                                 conv[(size_t)n] = ChunkPairMeta{(uint32_t)ch.o1[n], (uint32_t)ch.o2[n], ch_words, 0u};
                                 LHGT_TRY(stage_sync(fill, ch.s1.data(), b1n));
                                 LHGT_TRY(stage_sync(fill + b1n, ch.s2.data(), b2n));
                                 LHGT_TRY(stage_sync(fill + moff, (const uint8_t*)conv.data(), (size_t)(n + 1) * sizeof(ChunkPairMeta)));
                             }
                             // slab chunks: one run of bases, mate 2 of a pair behind its mate 1 (b2 = CHUNK_INTERLEAVED); chunks from vectors: two runs
                             desc()[n_desc++] = ChunkDesc{(uint32_t)n_open, (uint32_t)n, (uint32_t)fill, ch.slab ? lhgt::CHUNK_INTERLEAVED : (uint32_t)(fill + b1n), (uint32_t)words,
                                                          (uint32_t)((fill + moff) / sizeof(ChunkPairMeta))};
                             fill = (fill + block + 15) & ~(size_t)15;
                             words += ch_words;
                             n_open += n;
                             kept += n;
                             reap(false);
                             if (n_open >= BATCH_PAIRS || fill >= BATCH_BYTES) return flush();
                             return LHGT_OK;
                         },
                         [&](bool block) { reap(block); }, prepare, share_fn, ctx->random_array.empty() ? LHGT_MAX_RANDOM : ctx->sampling_filled,
                         can_reset ? std::function<int()>(reset) : std::function<int()>(nullptr));
    const double t_f0 = now_s();
    if (rc == LHGT_OK) rc = flush();
    (void)hipStreamSynchronize(cs);               // whatever happened: no copy may still read a slab,
    if (cs2) { (void)hipStreamSynchronize(cs2); (void)hipStreamDestroy(cs2); }
    (void)hipStreamSynchronize(ctx->stream);      // no kernel the staging buffers
    if (ingest_trace()) fprintf(stderr, "[lhgt ingest] staging + pinned pool %.3fs (behind the line count), %ld batches, last batch + drain %.3fs\n", t_alloc, n_batches, now_s() - t_f0);
    for (auto& e : buf_free) if (e) (void)hipEventDestroy(e);
    if (copied) (void)hipEventDestroy(copied);
    if (copied2) (void)hipEventDestroy(copied2);
    if (pool) {                                    // ... so every slab is free again, also one a failed chunk still held
        std::lock_guard<std::mutex> lk(pool->mu);
        pool->free_ids.clear();
        for (int i = 0; i < n_slabs; i++) pool->free_ids.push_back(i);
    }
    for (size_t i = 0; i + 1 < count_ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, count_ev[i], count_ev[i + 1]) == hipSuccess) ctx->count_on_load_ms += ms;
    }
    for (hipEvent_t e : count_ev) hipEventDestroy(e);
    if (rc != LHGT_OK) return rc;
    if (n_pairs_kept) *n_pairs_kept = kept;
    return LHGT_OK;
}

extern "C" {

int lhgt_pairs_load_fastq(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent, int shard_rank,
                          int shard_world, long shard_block, long* n_pairs_seen, long* n_pairs_kept) {
    return load_fastq_impl(ctx, fq1, fq2, ratio_percent, shard_rank, shard_world, shard_block, n_pairs_seen, n_pairs_kept, nullptr);
}

// ---- multi-GPU ingest: every rank counts the lines of ITS share of both files, the host layer all-gathers the
// pieces, and every rank parses only its contiguous run of fq1's chunks against the whole plan.  The reference's threads do the
// same by byte ranges (E:1426-1434) but lose records at the boundaries; here the global line numbers make every split exact, and
// sampling (n % 5*10^7 of the GLOBAL read ordinal, E:1037-1044) does not depend on the split.
long lhgt_fastq_plan_chunk_bytes(void) {   // LHGT_INGEST_CHUNK_BYTES: test hook (small files cut into many chunks)
    const char* e = getenv("LHGT_INGEST_CHUNK_BYTES");
    const long v = e ? atol(e) : 0;
    return v >= 256 ? v : (long)((size_t)2 << 20);
}

// the chunk sizes at which the two files of a pair should be planned so that the planned parse can take the single-pass loader's
// columns (fq2 is cut into as many chunks as fq1); both = lhgt_fastq_plan_chunk_bytes() where no such grid exists
int lhgt_fastq_pair_chunk_bytes(const char* fq1, const char* fq2, long* chunk1, long* chunk2) {
    if (!fq1 || !fq2 || !chunk1 || !chunk2) LHGT_FAIL(LHGT_E_ARG, "null argument");
    struct stat a, b;
    if (stat(fq1, &a)) LHGT_FAIL(LHGT_E_IO, "cannot stat %s", fq1);
    if (stat(fq2, &b)) LHGT_FAIL(LHGT_E_IO, "cannot stat %s", fq2);
    const long ch = lhgt_fastq_plan_chunk_bytes();
    size_t c1 = 0, c2 = 0;
    *chunk1 = *chunk2 = ch;
    if (stream_chunking((size_t)a.st_size, (size_t)b.st_size, (size_t)ch, &c1, &c2) && (long)c1 == ch) *chunk2 = (long)c2;
    return LHGT_OK;
}

int lhgt_fastq_plan_part(const char* fq, long chunk_bytes, int part, int n_parts, uint64_t* start, long* n_lines, long cap, long* n_out,
                         long* n_chunks_total, long* len_sums) {
    if (!fq || chunk_bytes < 1 || n_parts < 1 || part < 0 || part >= n_parts || !n_out) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    Mapped m;
    LHGT_TRY(m.open(fq));
    const size_t nch = n_plan_chunks(m.n, (size_t)chunk_bytes);
    const size_t c_lo = nch * (size_t)part / (size_t)n_parts, c_hi = nch * (size_t)(part + 1) / (size_t)n_parts;
    *n_out = (long)(c_hi - c_lo);
    if (n_chunks_total) *n_chunks_total = (long)nch;
    if (!start || !n_lines) return LHGT_OK;       // size query
    if (cap < (long)(c_hi - c_lo)) LHGT_FAIL(LHGT_E_ARG, "room for %ld chunks, part %d/%d of %s has %zu", cap, part, n_parts, fq, c_hi - c_lo);
    std::vector<size_t> st;
    std::vector<long> cnt;
    const double t0 = now_s();
    // pread + one SIMD sweep per chunk (host_fastq_stream.cpp), the line lengths for cal_sam_ratio from the same newline lists; a
    // file with a line longer than a chunk's margin goes through the mapping, line by line, as before
    if (!(getenv("LHGT_INGEST_STREAM") && atoi(getenv("LHGT_INGEST_STREAM")) == 0) && chunk_bytes >= 64 && chunk_bytes <= (1L << 30) &&
        plan_columns(m, (size_t)chunk_bytes, c_lo, c_hi, default_threads(), start, n_lines, len_sums)) {
        if (ingest_trace())
            fprintf(stderr, "[lhgt ingest] part %d/%d: lines of chunks [%zu, %zu) of %zu of %s counted in %.3fs (pread + newline lists%s)\n", part, n_parts, c_lo, c_hi,
                    nch, fq, now_s() - t0, len_sums ? ", with line lengths" : "");
        return LHGT_OK;
    }
    plan_range(m, (size_t)chunk_bytes, c_lo, c_hi, default_threads(), &st, &cnt);
    for (size_t i = 0; i < c_hi - c_lo; i++) { start[i] = st[i]; n_lines[i] = cnt[i]; }
    if (len_sums)      // cal_sam_ratio's pass (E:1244-1270) folded into the line count: line lengths by (line index inside the chunk) mod 4
        parallel_for((long)(c_hi - c_lo), default_threads(), [&](long i) {
            LineCursor lc(m);
            lc.cur = st[(size_t)i];
            const uint8_t* s;
            size_t len, at;
            long q = 0, sums[4] = {0, 0, 0, 0};
            while (lc.cur < st[(size_t)i + 1] && lc.next(&s, &len, &at)) { sums[q & 3] += (long)len; q++; }
            for (int r = 0; r < 4; r++) len_sums[4 * i + r] = sums[r];
        });
    if (ingest_trace())
        fprintf(stderr, "[lhgt ingest] part %d/%d: lines of bytes [%zu, %zu) of %zu of %s counted in %.3fs\n", part, n_parts, st.front(), st.back(), m.n, fq, now_s() - t0);
    return LHGT_OK;
}

int lhgt_pairs_load_fastq_planned(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent, const uint64_t* start1,
                                  const long* n_lines1, long n1, const uint64_t* start2, const long* n_lines2, long n2, int part,
                                  int n_parts, long* n_pairs_seen, long* n_pairs_kept) {
    if (!start1 || !n_lines1 || !start2 || !n_lines2 || n1 < 1 || n2 < 1 || n_parts < 1 || part < 0 || part >= n_parts)
        LHGT_FAIL(LHGT_E_ARG, "bad FASTQ plan");
    return load_fastq_impl(ctx, fq1, fq2, ratio_percent, 0, 1, 1, n_pairs_seen, n_pairs_kept,
                           [&](const Mapped& m1, const Mapped& m2, ParseShare* sh, ChunkPlan* p1, ChunkPlan* p2) -> int {
                               LHGT_TRY(plan_from_arrays(m1, start1, n_lines1, n1, p1));
                               LHGT_TRY(plan_from_arrays(m2, start2, n_lines2, n2, p2));
                               sh->p1 = p1; sh->p2 = p2; sh->part = part; sh->n_parts = n_parts;
                               sh->raw_start1 = start1; sh->raw_count1 = n_lines1; sh->raw_n1 = n1;
                               sh->raw_start2 = start2; sh->raw_count2 = n_lines2; sh->raw_n2 = n2;
                               return LHGT_OK;
                           });
}

// count_diff_kmer.cpp's reader (C:53-153): the file in 10 thread chunks of `size_for_chunks` bytes (fq1's size for both files,
// C:328-355).  A chunk is entered at the nearest '@' at or before its start (C:61-69; thread 0 from byte 0), read token by token
// (`>>`, C:91) while the summed token lengths, counted from `start`, stay below `end` (C:70, 93-96) -- chunks overrun into their
// successors and those reads are counted twice, as there; sampling restarts from srand(seed) in every chunk, one rand() % 100
// per sequence token (C:87-89, 107-109).  Kept reads become mate-1-only entries of the pair store.  Reads of a chunk must
// have one length <= 150 (the tool cuts every read to its chunk's first read's length and overruns its buffers otherwise).
int lhgt_reads_load_count_diff(lhgt_ctx* ctx, const char* fq, long size_for_chunks, int ratio_percent, unsigned seed, long* n_reads_kept) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !fq) LHGT_FAIL(LHGT_E_ARG, "null argument");
    Mapped m;
    LHGT_TRY(m.open(fq));
    const long n = (long)m.n, size = size_for_chunks < 0 ? n : size_for_chunks, each = size / 10;
    std::vector<uint8_t> seq;
    std::vector<uint64_t> off(1, 0);
    auto ws = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); };
    for (int t = 0; t < 10; t++) {
        const long start = t * each, end = t == 9 ? size : (t + 1) * each;
        if (start > 0 && start >= n) LHGT_FAIL(LHGT_E_FORMAT, "count_diff_kmer: thread %d would start behind the end of %s", t, fq);
        long pos = 0;
        for (long i = start; i > 0; i--) if (m.p[i] == '@') { pos = i; break; }
        long cur = pos, add_size = start, tok = 0, read_len = 0;
        LHGT_TRY(lhgt_rng_seed(ctx, seed));
        for (;;) {
            while (cur < n && ws(m.p[cur])) cur++;
            if (cur >= n) break;
            const long t0 = cur;
            while (cur < n && !ws(m.p[cur])) cur++;
            const long len = cur - t0;
            if (add_size >= end) break;
            add_size += len;
            if (tok % 4 == 1) {
                if (tok == 1) read_len = len;
                if (len != read_len || len > 150) LHGT_FAIL(LHGT_E_FORMAT, "count_diff_kmer: reads of unequal length or longer than 150 in %s (the reference overruns its buffers, C:75-76, 103-105)", fq);
                if (rng_next(ctx) % 100 < ratio_percent) {
                    seq.insert(seq.end(), m.p + t0, m.p + t0 + len);
                    off.push_back(seq.size());
                }
            }
            tok++;
        }
    }
    const long kept = (long)off.size() - 1;
    if (kept > 0) {
        std::vector<uint64_t> off2((size_t)kept + 1, 0);
        std::vector<uint8_t> fl((size_t)kept, PAIR_COUNT1), none(1, 0);
        LHGT_TRY(lhgt_pairs_append_flags(ctx, seq.data(), off.data(), none.data(), off2.data(), kept, fl.data()));
    }
    if (n_reads_kept) *n_reads_kept = kept;
    return LHGT_OK;
}

// Host-only probe of the parser (tests): FNV-1a digest over every kept pair (lengths, bases, mate-2 flag) in order.
int lhgt_fastq_parse_digest(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int shard_rank,
                            int shard_world, long shard_block, int threads, long chunk_bytes, long* n_pairs_seen, long* n_pairs_kept,
                            uint64_t* digest) {
    return lhgt_fastq_parse_digest_threads(fq1, fq2, ratio_percent, random_array_or_null, shard_rank, shard_world, shard_block, threads,
                                           chunk_bytes, 1, n_pairs_seen, n_pairs_kept, digest, nullptr);
}

// the same with the reference's -t N read partition (emulate_threads > 1); counts[3] (optional) = entries with mate 1 counted
// in phase A, with mate 2 counted, voted in phase C
int lhgt_fastq_parse_digest_threads(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int shard_rank,
                                    int shard_world, long shard_block, int threads, long chunk_bytes, int emulate_threads,
                                    long* n_pairs_seen, long* n_pairs_kept, uint64_t* digest, long* counts) {
    return lhgt_fastq_parse_digest_planned(fq1, fq2, ratio_percent, random_array_or_null, shard_rank, shard_world, shard_block, threads, chunk_bytes,
                                           emulate_threads, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1, 0, n_pairs_seen, n_pairs_kept, digest, counts);
}

// ... and with plans made elsewhere (start1 non-null: lhgt_fastq_plan_part pieces, concatenated) and only part `part` of `n_parts`
// parsed.  chain != 0: *digest is the state to continue from, so that the parts of a split, run in order, give the digest of the whole.
int lhgt_fastq_parse_digest_planned(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int shard_rank,
                                    int shard_world, long shard_block, int threads, long chunk_bytes, int emulate_threads,
                                    const uint64_t* start1, const long* n_lines1, long n1, const uint64_t* start2, const long* n_lines2, long n2,
                                    int part, int n_parts, int chain, long* n_pairs_seen, long* n_pairs_kept, uint64_t* digest, long* counts) {
    if (!fq1 || !fq2 || !digest || chunk_bytes < 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (ratio_percent < 100.0 && !random_array_or_null) LHGT_FAIL(LHGT_E_ARG, "sampling needs the random array");
    uint64_t h = chain ? *digest : 1469598103934665603ull;
    const uint64_t h_start = h;
    ShareFn share_fn = nullptr;
    if (start1)
        share_fn = [&](const Mapped& m1, const Mapped& m2, ParseShare* sh, ChunkPlan* p1, ChunkPlan* p2) -> int {
            if (!n_lines1 || !start2 || !n_lines2 || n1 < 1 || n2 < 1 || n_parts < 1 || part < 0 || part >= n_parts) LHGT_FAIL(LHGT_E_ARG, "bad FASTQ plan");
            LHGT_TRY(plan_from_arrays(m1, start1, n_lines1, n1, p1));
            LHGT_TRY(plan_from_arrays(m2, start2, n_lines2, n2, p2));
            sh->p1 = p1; sh->p2 = p2; sh->part = part; sh->n_parts = n_parts;
            sh->raw_start1 = start1; sh->raw_count1 = n_lines1; sh->raw_n1 = n1;
            sh->raw_start2 = start2; sh->raw_count2 = n_lines2; sh->raw_n2 = n2;
            return LHGT_OK;
        };
    long kept = 0;
    auto mix = [&](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } };
    long cnt[3] = {0, 0, 0};
    int rc = parse_pairs(fq1, fq2, ratio_percent, random_array_or_null, shard_rank, shard_world, shard_block, threads, (size_t)chunk_bytes,
                         emulate_threads, n_pairs_seen, (SlabPool*)nullptr, [&](ParsedChunk& ch) -> int {
                             long n = (long)ch.o1.size() - 1;
                             for (long i = 0; i < n; i++) {
                                 for (int q = 0; q < 3; q++) cnt[q] += (ch.flags[i] >> q) & 1;
                                 uint64_t l1 = ch.o1[i + 1] - ch.o1[i], l2 = ch.o2[i + 1] - ch.o2[i];
                                 mix((const uint8_t*)&l1, 8);
                                 mix(ch.s1.data() + ch.o1[i], l1);
                                 mix((const uint8_t*)&l2, 8);
                                 mix(ch.s2.data() + ch.o2[i], l2);
                                 { const uint8_t c2 = (ch.flags[i] & PAIR_COUNT2) ? 1 : 0; mix(&c2, 1); if (!(ch.flags[i] & PAIR_VOTE)) { const uint8_t f = ch.flags[i]; mix(&f, 1); } }
                             }
                             kept += n;
                             return LHGT_OK;
                         },
                         [](bool) {}, nullptr, share_fn, LHGT_MAX_RANDOM,
                         [&]() -> int { h = h_start; kept = 0; cnt[0] = cnt[1] = cnt[2] = 0; return LHGT_OK; });
    if (rc != LHGT_OK) return rc;
    *digest = h;
    if (n_pairs_kept) *n_pairs_kept = kept;
    if (counts) for (int q = 0; q < 3; q++) counts[q] = cnt[q];
    return LHGT_OK;
}

// Host-only rate probe of the loader (tools/ingest_scaling.py): the parse exactly as lhgt_pairs_load_fastq runs it -- worker threads
// writing kept bases and per-pair records into a pool of slabs -- with a consumer that only adds up what it is handed and gives the
// slab back (no GPU, no copy).  Without plans (start1 null) the single pass is tried first, as in the loader; with plans the
// caller's part of the planned parse runs.  seconds = wall time of the call (mapping the files included).
int lhgt_fastq_parse_rate(const char* fq1, const char* fq2, double ratio_percent, const float* random_array_or_null, int threads, long chunk_bytes,
                          int emulate_threads, const uint64_t* start1, const long* n_lines1, long n1, const uint64_t* start2, const long* n_lines2,
                          long n2, int part, int n_parts, long* n_pairs_seen, long* n_pairs_kept, long* n_bases, double* seconds, uint64_t* digest_or_null) {
    if (!fq1 || !fq2 || chunk_bytes < 256 || threads < 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (ratio_percent < 100.0 && !random_array_or_null) LHGT_FAIL(LHGT_E_ARG, "sampling needs the random array");
    const double t0 = now_s();
    ShareFn share_fn = nullptr;
    if (start1)
        share_fn = [&](const Mapped& m1, const Mapped& m2, ParseShare* sh, ChunkPlan* p1, ChunkPlan* p2) -> int {
            if (!n_lines1 || !start2 || !n_lines2 || n1 < 1 || n2 < 1 || n_parts < 1 || part < 0 || part >= n_parts) LHGT_FAIL(LHGT_E_ARG, "bad FASTQ plan");
            LHGT_TRY(plan_from_arrays(m1, start1, n_lines1, n1, p1));
            LHGT_TRY(plan_from_arrays(m2, start2, n_lines2, n2, p2));
            sh->p1 = p1; sh->p2 = p2; sh->part = part; sh->n_parts = n_parts;
            sh->raw_start1 = start1; sh->raw_count1 = n_lines1; sh->raw_n1 = n1;
            sh->raw_start2 = start2; sh->raw_count2 = n_lines2; sh->raw_n2 = n2;
            return LHGT_OK;
        };
    const size_t HALF = (size_t)chunk_bytes + 1024, SLAB = 2 * HALF + sizeof(ChunkPairMeta) * (CHUNK_META_CAP + 1);
    const int n_slabs = std::max(threads + threads / 3 + 4, (int)std::min<size_t>(4096, ((size_t)280 << 20) / SLAB));
    std::unique_ptr<uint8_t[]> mem(new uint8_t[(size_t)n_slabs * SLAB]);
    SlabPool pool;
    pool.base = mem.get();
    pool.slab_bytes = SLAB;
    pool.half_bytes = HALF;
    pool.k = 32;
    for (int i = 0; i < n_slabs; i++) pool.free_ids.push_back(i);
    long kept = 0, bases = 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const uint8_t* p, size_t n) { for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; } };
    int rc = parse_pairs(fq1, fq2, ratio_percent, random_array_or_null, 0, 1, 1, threads, (size_t)chunk_bytes, emulate_threads, n_pairs_seen, (SlabPool*)nullptr,
                         [&](ParsedChunk& ch) -> int {
                             kept += ch.n_pairs();
                             bases += (long)ch.bases_bytes();
                             if (digest_or_null) {          // lhgt_fastq_parse_digest's digest, read back from the slab's block as the device would
                                                                  const long n = ch.n_pairs();
                                 for (long i = 0; i < n; i++) {
                                     const uint8_t *a, *b;
                                     uint64_t l1, l2;
                                     uint8_t fl;
                                     if (ch.slab) {
                                         const ChunkPairMeta r = ch.meta[i], q = ch.meta[i + 1];
                                         a = ch.slab + r.rel1; l1 = r.rel2 - r.rel1; b = ch.slab + r.rel2; l2 = q.rel1 - r.rel2; fl = (uint8_t)r.flags;
                                     } else {
                                         a = ch.s1.data() + ch.o1[i]; l1 = ch.o1[i + 1] - ch.o1[i]; b = ch.s2.data() + ch.o2[i]; l2 = ch.o2[i + 1] - ch.o2[i]; fl = ch.flags[i];
                                     }
                                     mix((const uint8_t*)&l1, 8); mix(a, l1); mix((const uint8_t*)&l2, 8); mix(b, l2);
                                     { const uint8_t c2 = (fl & PAIR_COUNT2) ? 1 : 0; mix(&c2, 1); if (!(fl & PAIR_VOTE)) mix(&fl, 1); }
                                 }
                             }
                             if (ch.slab_id >= 0) pool.release(ch.slab_id);
                             return LHGT_OK;
                         },
                         [](bool) {}, [&](SlabPool** out) -> int { *out = &pool; return LHGT_OK; }, share_fn, LHGT_MAX_RANDOM,
                         [&]() -> int { kept = 0; bases = 0; h = 1469598103934665603ull; return LHGT_OK; });
    if (digest_or_null) *digest_or_null = h;
    if (n_pairs_kept) *n_pairs_kept = kept;
    if (n_bases) *n_bases = bases;
    if (seconds) *seconds = now_s() - t0;
    return rc;
}

// Host-only packer (round 6, late): the file `lhgt_pairs_load_fastq` + `lhgt_pairs_store_write` write for a record-aligned pair of files --
// the same bytes -- without a GPU in the machine: the loader's own parse (every read kept, no thread emulation), twice -- once for the
// longest read (the stride) and the checks, once to write -- and pack_bases (k_ingest.hip) restated for a host thread: bit 31 - b of word
// w of a plane = base 32 w + b.  The store's order is the files' order here as there (chunks are handed over in file order).
namespace {
struct BaseCodeTable {                                         // lhgt_hash.hpp: base_code (E:1112-1151: upper and lower case ACGT only; 4 = not a base)
    uint8_t t[256];
    BaseCodeTable() { memset(t, 4, sizeof t); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; }
};
const BaseCodeTable g_base_code;
inline uint32_t base_code_host(uint8_t c) { return g_base_code.t[c]; }
void pack_mate_host(uint32_t* dst, const uint8_t* s, size_t len) {
    const size_t wpr = (len + 31) / 32 + 1;                   // the last word of every plane stays zero
    for (size_t w = 0; w < wpr; w++) {
        uint32_t hi = 0, lo = 0, nb = 0;
        const size_t base = 32 * w, n = len > base ? std::min<size_t>(32, len - base) : 0;
        for (size_t b = 0; b < n; b++) {
            const uint32_t c = base_code_host(s[base + b]), bit = 0x80000000u >> b;
            if (c == 4) nb |= bit;
            else { if (c & 2) hi |= bit; if (c & 1) lo |= bit; }
        }
        dst[w] = hi; dst[wpr + w] = lo; dst[2 * wpr + w] = nb;
    }
}
}  // namespace

int lhgt_fastq_pack_host(const char* fq1, const char* fq2, const char* out_path, unsigned long long data_offset, int threads, long* stride_out,
                         long* n_pairs_out, long* q4_first_pair, unsigned long long* bases1, int* max_len_out) {
    if (!fq1 || !fq2 || !out_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (threads < 1) threads = ingest_default_threads();
    const long chunk_bytes = lhgt_fastq_plan_chunk_bytes();
    // pass 1: the longest read, the pairs, the flags a clean pair of files gives (7 everywhere, then 5 -- mate 2 behind size(fq1), quirk Q4 -- to the end)
    long total = 0, q4 = -1, seen = 0;
    unsigned long long bases = 0;
    int max_len = 0, bad_rc = LHGT_OK;
    int rc = parse_pairs(fq1, fq2, 100.0, nullptr, 0, 1, 1, threads, (size_t)chunk_bytes, 1, &seen, (SlabPool*)nullptr, [&](ParsedChunk& ch) -> int {
                             const long n = (long)ch.o1.size() - 1;
                             for (long i = 0; i < n; i++) {
                                 const uint64_t la = ch.o1[i + 1] - ch.o1[i], lb = ch.o2[i + 1] - ch.o2[i];
                                 const uint8_t f = ch.flags[i];
                                 if (f == PAIR_ALL && q4 < 0) { /* counted */ }
                                 else if (f == (PAIR_COUNT1 | PAIR_VOTE)) { if (q4 < 0) q4 = total + i; }
                                 else { set_error("pair %ld carries flags %d: not a record-aligned pair of files read whole (the FASTQ loader is the way for those)", total + i, (int)f); bad_rc = LHGT_E_FORMAT; return bad_rc; }
                                 if ((int)la > max_len) max_len = (int)la;
                                 if ((int)lb > max_len) max_len = (int)lb;
                                 bases += la;
                             }
                             total += n;
                             return LHGT_OK;
                         },
                         [](bool) {}, nullptr, nullptr, LHGT_MAX_RANDOM,
                         [&]() -> int { total = 0; q4 = -1; bases = 0; max_len = 0; return LHGT_OK; });
    if (rc != LHGT_OK) return rc;
    if (seen != total) LHGT_FAIL(LHGT_E_FORMAT, "the loader kept %ld of %ld pairs: not a clean pair of files; keep them as FASTQ", total, seen);
    const long stride = 4 + 24 * ((long)(max_len + 31) / 32 + 1);
    // pass 2: the records
    const int fd = ::open(out_path, O_WRONLY);
    if (fd < 0) LHGT_FAIL(LHGT_E_IO, "cannot open %s for writing", out_path);
    long at = 0;
    std::vector<uint8_t> buf;
    rc = parse_pairs(fq1, fq2, 100.0, nullptr, 0, 1, 1, threads, (size_t)chunk_bytes, 1, &seen, (SlabPool*)nullptr, [&](ParsedChunk& ch) -> int {
                         const long n = (long)ch.o1.size() - 1;
                         if (n <= 0) return LHGT_OK;
                         if (buf.size() < (size_t)n * (size_t)stride) buf.resize((size_t)n * (size_t)stride);
                         std::atomic<int> io_bad{0};
                         parallel_for((n + 4095) / 4096, threads, [&](long c) {        // a slice of the chunk: packed and written by one thread
                             const long i0 = c * 4096, i1 = std::min(n, (c + 1) * 4096);
                             memset(buf.data() + (size_t)i0 * (size_t)stride, 0, (size_t)(i1 - i0) * (size_t)stride);
                             for (long i = i0; i < i1; i++) {
                                 const size_t la = ch.o1[i + 1] - ch.o1[i], lb = ch.o2[i + 1] - ch.o2[i];
                                 uint8_t* r = buf.data() + (size_t)i * (size_t)stride;
                                 const uint16_t hd[2] = {(uint16_t)la, (uint16_t)lb};
                                 memcpy(r, hd, 4);
                                 uint32_t* w = reinterpret_cast<uint32_t*>(r + 4);
                                 pack_mate_host(w, ch.s1.data() + ch.o1[i], la);
                                 pack_mate_host(w + 3 * ((la + 31) / 32 + 1), ch.s2.data() + ch.o2[i], lb);
                             }
                             const size_t want = (size_t)(i1 - i0) * (size_t)stride;
                             size_t done = 0;
                             while (done < want) {
                                 const ssize_t wr = pwrite(fd, buf.data() + (size_t)i0 * (size_t)stride + done, want - done,
                                                           (off_t)(data_offset + (unsigned long long)(at + i0) * (unsigned long long)stride + done));
                                 if (wr <= 0) { io_bad = 1; return; }
                                 done += (size_t)wr;
                             }
                         });
                         if (io_bad) { set_error("write to %s failed", out_path); return LHGT_E_IO; }
                         at += n;
                         return LHGT_OK;
                     },
                     [](bool) {}, nullptr, nullptr, LHGT_MAX_RANDOM, [&]() -> int { at = 0; return LHGT_OK; });
    close(fd);
    if (rc != LHGT_OK) return rc;
    if (at != total) LHGT_FAIL(LHGT_E_IO, "%ld records written for %ld pairs", at, total);
    if (stride_out) *stride_out = stride;
    if (n_pairs_out) *n_pairs_out = total;
    if (q4_first_pair) *q4_first_pair = q4 < 0 ? total : q4;
    if (bases1) *bases1 = bases;
    if (max_len_out) *max_len_out = max_len;
    return LHGT_OK;
}

int lhgt_ingest_last_path(char* why, long cap) {
    if (why && cap > 0) { strncpy(why, lhgt::g_last_path_why.c_str(), (size_t)cap - 1); why[cap - 1] = 0; }
    return lhgt::g_last_path;
}

long lhgt_fastq_thread_entry(const uint8_t* text, long n, long start) {
    return text && n >= 0 && start >= 0 ? lhgt::thread_entry(text, n, start) : -2;
}

// where the reference's thread i of `threads` enters a FASTQ and which lines it consumes (tests; E:44-89, 1019-1026)
int lhgt_fastq_thread_chunks(const char* fq, long size_for_chunks, int threads, long* entry_byte, long* first_line, long* n_lines) {
    if (!fq || threads < 1 || !entry_byte || !first_line || !n_lines) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    Mapped m;
    LHGT_TRY(m.open(fq));
    ChunkPlan pl = plan_chunks(m, (size_t)1 << 20, default_threads());
    ThreadPart tp;
    std::vector<long> pos;
    LHGT_TRY(thread_part(m, pl, size_for_chunks < 0 ? (long)m.n : size_for_chunks, threads, fq, &tp, &pos));
    for (int i = 0; i < threads; i++) { entry_byte[i] = pos[i]; first_line[i] = tp.first[i]; n_lines[i] = tp.count[i]; }
    return LHGT_OK;
}

// read_ref (E:727-886): header of 300 words (word j = cc[j] | cc[j+1] << 16, the reference
// writes 4 bytes from a short array), then per contig [len][hashes]; genome.len.txt beside it.
int lhgt_index_build(lhgt_ctx* ctx, const char* fasta_path, const char* index_path, const char* genome_len_path,
                     long* n_contigs, long* n_bases) {
    LHGT_DEVICE_ENTRY(ctx);
    NodeAffinity near_gpu(ctx ? ctx->device : 0);   // the upload's host threads and staging next to the GPU (see NodeAffinity)
    if (!ctx || !fasta_path || !index_path || !genome_len_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder: call lhgt_coder_generate or lhgt_coder_set first");
    FILE* idx = fopen(index_path, "wb");
    if (!idx) LHGT_FAIL(LHGT_E_IO, "cannot write %s", index_path);
    FILE* lenf = fopen(genome_len_path, "w");
    if (!lenf) { fclose(idx); LHGT_FAIL(LHGT_E_IO, "cannot write %s", genome_len_path); }
    for (int j = 0; j < LHGT_CODER_SLOTS; j++) {
        uint32_t w = (uint16_t)ctx->cc[j] | ((uint32_t)(uint16_t)(j + 1 < LHGT_CODER_SLOTS ? ctx->cc[j + 1] : 0) << 16);
        fwrite(&w, 4, 1, idx);
    }
    long contigs = 0, bases = 0;
    // The file's sequences are taken in spans of ~64 Mbase; a span's text goes to the GPU as it is, is stripped, packed and hashed
    // there by one launch each, straight into the file's layout ([u32 len][(len-k+1)*e u32] per contig), and comes back as one
    // copy and one write (a catalogue like UHGG has hundreds of thousands of contigs: two launches, a copy and two writes per
    // contig is what the first version did; the second still built every span byte by byte on one host thread).
    const int k = ctx->k, e = ctx->e;
    Mapped fa;
    int rc = fa.open(fasta_path);
    FastaIndex fx;
    if (rc == LHGT_OK) {
        fasta_scan(fa, default_threads(), &fx);
        long cum = 0;
        for (size_t i = 0; i < fx.seqs.size() && rc == LHGT_OK; i++) {
            const long len = (long)fx.seqs[i].len;
            cum += len;
            if (len <= k) continue;                                        // E:772, 836; it still used up a ref_index (quirk Q7)
            if (len >= (1L << 32) - 4096) { set_error("contig %s has %ld bases", fx.name(fa.p, i).c_str(), len); rc = LHGT_E_FORMAT; break; }
            fprintf(lenf, "%s\t%ld\t%ld\t%ld\n", fx.name(fa.p, i).c_str(), (long)i, len, cum);
            contigs++;
            bases += len;
        }
    }
    std::vector<uint64_t> ow;
    std::vector<uint32_t> host;
    uint32_t* d_out = nullptr;
    size_t d_cap = 0;
    if (rc == LHGT_OK)
        rc = fasta_spans(ctx, fa, fx, (uint64_t)64 << 20, [&](const uint8_t* d_bases, long n_bases_span, const uint64_t* coff, size_t first, long n_c) -> int {
            uint64_t out_words = 0;
            ow.assign((size_t)n_c, ~0ull);
            for (long c = 0; c < n_c; c++) {
                const long len = (long)fx.seqs[first + (size_t)c].len;
                if (len <= k) continue;
                ow[(size_t)c] = out_words + 1;
                out_words += 1 + (uint64_t)(len - k + 1) * e;
            }
            if (out_words == 0) return LHGT_OK;
            if (out_words > d_cap) {
                if (d_out) lhgt::dev_free(d_out);
                d_out = nullptr;
                d_cap = out_words + out_words / 4;
                LHGT_HIP(lhgt::dev_malloc(&d_out, d_cap * 4));
            }
            LHGT_TRY(hash_span_dev_ascii(ctx, d_bases, n_bases_span, coff, ow.data(), n_c, d_out));
            host.resize(out_words);
            LHGT_HIP(hipMemcpyAsync(host.data(), d_out, out_words * 4, hipMemcpyDeviceToHost, ctx->stream));
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            for (long c = 0; c < n_c; c++)
                if (ow[(size_t)c] != ~0ull) host[ow[(size_t)c] - 1] = (uint32_t)(coff[c + 1] - coff[c]);
            if (fwrite(host.data(), 4, out_words, idx) != out_words) LHGT_FAIL(LHGT_E_IO, "short write to %s", index_path);
            return LHGT_OK;
        });
    if (d_out) lhgt::dev_free(d_out);
    fclose(idx);
    fclose(lenf);
    if (n_contigs) *n_contigs = contigs;
    if (n_bases) *n_bases = bases;
    return rc;
}

int lhgt_index_load(lhgt_ctx* ctx, const char* index_path, long* n_contigs, long* n_bases) {
    return lhgt_index_load_shard(ctx, index_path, 0, 1, n_contigs, n_bases);
}

// saved_random_coder (E:1224-1242) alone: the coder of an existing index file, none of its hashes
int lhgt_index_read_coder(lhgt_ctx* ctx, const char* index_path) {
    if (!ctx || !index_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    FILE* f = fopen(index_path, "rb");
    if (!f) LHGT_FAIL(LHGT_E_IO, "cannot open %s", index_path);
    uint32_t w[LHGT_CODER_SLOTS];
    const size_t got = fread(w, 4, LHGT_CODER_SLOTS, f);
    fclose(f);
    if (got != (size_t)LHGT_CODER_SLOTS) LHGT_FAIL(LHGT_E_FORMAT, "%s: not an index file (no coder header)", index_path);
    int16_t cc[LHGT_CODER_SLOTS];
    for (int i = 0; i < LHGT_CODER_SLOTS; i++) cc[i] = (int16_t)w[i];
    return lhgt_coder_set(ctx, cc);
}

// The reference made resident straight from the FASTA, in the context's form (lhgt_set_reference_form): what read_ref
// (E:727-886) would put into the index file and read_index (E:888-979) would read back -- the same contigs (length > k), the same
// sequential numbering, the same hashes -- without the file in between.  In the packed form only the bases become resident
// (3/8 byte per base instead of 4e) and phase B recomputes the hashes.  genome_len_path (nullable): also write genome.len.txt.
int lhgt_reference_load_fasta(lhgt_ctx* ctx, const char* fasta_path, const char* genome_len_path, long* n_contigs, long* n_bases) {
    LHGT_DEVICE_ENTRY(ctx);
    NodeAffinity near_gpu(ctx ? ctx->device : 0);   // the upload's host threads and staging next to the GPU (see NodeAffinity)
    if (!ctx || !fasta_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder: call lhgt_coder_generate, lhgt_coder_set or lhgt_index_read_coder first");
    const int k = ctx->k;
    Mapped fa;
    LHGT_TRY(fa.open(fasta_path));
    FastaIndex fx;
    fasta_scan(fa, default_threads(), &fx);
    FILE* lenf = nullptr;
    if (genome_len_path && !(lenf = fopen(genome_len_path, "w"))) LHGT_FAIL(LHGT_E_IO, "cannot write %s", genome_len_path);
    std::vector<uint32_t> lens;
    long cum = 0;
    for (size_t i = 0; i < fx.seqs.size(); i++) {
        const long len = (long)fx.seqs[i].len;
        cum += len;
        if (len <= k) continue;                                            // E:772, 836
        if (len >= (1L << 32) - 4096) {
            if (lenf) fclose(lenf);
            LHGT_FAIL(LHGT_E_FORMAT, "contig %s has %ld bases", fx.name(fa.p, i).c_str(), len);
        }
        if (lenf) fprintf(lenf, "%s\t%ld\t%ld\t%ld\n", fx.name(fa.p, i).c_str(), (long)i, len, cum);
        lens.push_back((uint32_t)len);
    }
    if (lenf) fclose(lenf);
    LHGT_TRY(index_layout(ctx, lens));
    LHGT_TRY(write_index_lens(ctx));
    long ci = 0;
    std::vector<long> contig_of;
    int rc = fasta_spans(ctx, fa, fx, (uint64_t)256 << 20, [&](const uint8_t* d_bases, long n_bases_span, const uint64_t* coff, size_t first, long n_c) -> int {
        contig_of.assign((size_t)n_c, -1L);
        for (long c = 0; c < n_c; c++)
            if ((long)fx.seqs[first + (size_t)c].len > k) contig_of[(size_t)c] = ci++;
        return install_span_dev_ascii(ctx, d_bases, n_bases_span, coff, contig_of.data(), n_c);
    });
    if (rc != LHGT_OK) ctx->index_resident = false;      // a half-filled reference must not pass for a resident one
    if (n_contigs) *n_contigs = (long)ctx->contigs.size();
    if (n_bases) *n_bases = (long)ctx->n_pos;
    return rc;
}

// host-only view of what the two FASTA loaders make of a file (tests): every sequence, the indexed ones, genome.len.txt
int lhgt_fasta_scan(const char* fasta_path, int k, const char* genome_len_path, long* n_sequences, long* n_contigs, long* n_bases) {
    if (!fasta_path || k < 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    Mapped fa;
    LHGT_TRY(fa.open(fasta_path));
    FastaIndex fx;
    fasta_scan(fa, default_threads(), &fx);
    FILE* lenf = nullptr;
    if (genome_len_path && !(lenf = fopen(genome_len_path, "w"))) LHGT_FAIL(LHGT_E_IO, "cannot write %s", genome_len_path);
    long cum = 0, contigs = 0, bases = 0;
    for (size_t i = 0; i < fx.seqs.size(); i++) {
        const long len = (long)fx.seqs[i].len;
        cum += len;
        if (len <= k) continue;
        if (lenf) fprintf(lenf, "%s\t%ld\t%ld\t%ld\n", fx.name(fa.p, i).c_str(), (long)i, len, cum);
        contigs++;
        bases += len;
    }
    if (lenf) fclose(lenf);
    if (n_sequences) *n_sequences = (long)fx.seqs.size();
    if (n_contigs) *n_contigs = contigs;
    if (n_bases) *n_bases = bases;
    return LHGT_OK;
}

int lhgt_index_load_shard(lhgt_ctx* ctx, const char* index_path, int shard_rank, int shard_world, long* n_contigs, long* n_bases) {
    LHGT_DEVICE_ENTRY(ctx);
    NodeAffinity near_gpu(ctx ? ctx->device : 0);   // the upload's host threads and staging next to the GPU (see NodeAffinity)
    if (!ctx || !index_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    Mapped m;
    LHGT_TRY(m.open(index_path));
    if (m.n < 4 * LHGT_CODER_SLOTS || m.n % 4) LHGT_FAIL(LHGT_E_FORMAT, "%s: not an index file (size %zu)", index_path, m.n);
    const uint32_t* w = (const uint32_t*)m.p;
    int16_t cc[LHGT_CODER_SLOTS];
    for (int i = 0; i < LHGT_CODER_SLOTS; i++) cc[i] = (int16_t)w[i];  // saved_random_coder: low half of each word
    LHGT_TRY(lhgt_coder_set(ctx, cc));
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) LHGT_FAIL(LHGT_E_ARG, "bad shard spec %d/%d", shard_rank, shard_world);
    const int irc = index_install_shard(ctx, w + LHGT_CODER_SLOTS, m.n / 4 - LHGT_CODER_SLOTS, shard_rank, shard_world);
    if (irc != LHGT_OK) { ctx->index_resident = false; return irc; }
    if (n_contigs) *n_contigs = (long)ctx->contigs.size();
    if (n_bases) *n_bases = (long)ctx->n_pos;
    return LHGT_OK;
}

}  // extern "C"
