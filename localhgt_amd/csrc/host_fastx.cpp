// host_fastx.cpp -- FASTA / FASTQ ingest on the host (mmap + memchr), reference line semantics.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include "lhgt_common.hpp"

namespace lhgt {

struct Mapped {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) munmap((void*)p, n);
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
            madvise((void*)p, n, MADV_SEQUENTIAL);
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
static size_t read_id_len(const uint8_t* s, size_t len) {
    size_t n = len;
    for (size_t i = 0; i < n; i++) if (s[i] == '/') { n = i; break; }
    for (size_t i = 0; i < n; i++) if (s[i] == ' ') { n = i; break; }
    for (size_t i = 0; i < n; i++) if (s[i] == '\t') { n = i; break; }
    return n;
}

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
    LineCursor lc(m);
    const uint8_t* s;
    size_t len, start;
    long i = 0, bases = 0;
    while (lc.next(&s, &len, &start)) {
        if (i % 4 == 1) bases += (long)len;
        i++;
    }
    bases *= 2;
    *ratio_percent = 100 * sample / (double)bases;
    if (n_records) *n_records = i / 4;
    return LHGT_OK;
}

int lhgt_pairs_load_fastq(lhgt_ctx* ctx, const char* fq1, const char* fq2, double ratio_percent, int shard_rank,
                          int shard_world, long shard_block, long* n_pairs_seen, long* n_pairs_kept) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx || !fq1 || !fq2) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world || shard_block < 1)
        LHGT_FAIL(LHGT_E_ARG, "bad shard spec %d/%d block %ld", shard_rank, shard_world, shard_block);
    if (ratio_percent < 100.0 && (long)ctx->random_array.size() != LHGT_MAX_RANDOM)
        LHGT_FAIL(LHGT_E_STATE, "lhgt_sampling_init(ratio) must precede lhgt_pairs_load_fastq when ratio < 100");
    Mapped m1, m2;
    LHGT_TRY(m1.open(fq1));
    LHGT_TRY(m2.open(fq2));
    const size_t size1 = m1.n;  // E:1419: both mates are cut at size(fq1) in phase A
    LineCursor c1(m1), c2(m2);
    const long CHUNK = 1 << 20;
    std::vector<uint8_t> s1, s2, cnt2;
    std::vector<uint64_t> o1{0}, o2{0};
    long lines = 0, kept = 0;
    const uint8_t *a, *b;
    size_t la, lb, sa, sb;
    auto flush = [&]() -> int {
        long n = (long)o1.size() - 1;
        if (n == 0) return LHGT_OK;
        int rc = upload_pairs(ctx, s1.data(), o1.data(), s2.data(), o2.data(), n, cnt2.data());
        s1.clear(); s2.clear(); cnt2.clear();
        o1.assign(1, 0); o2.assign(1, 0);
        return rc;
    };
    while (c1.next(&a, &la, &sa)) {
        bool have2 = c2.next(&b, &lb, &sb);
        if (!have2) LHGT_FAIL(LHGT_E_FORMAT, "%s has fewer lines than %s", fq2, fq1);
        if (lines == 0) {  // E:368-402: the two first read IDs must agree
            size_t ia = read_id_len(a, la), ib = read_id_len(b, lb);
            if (ia != ib || memcmp(a, b, ia))
                LHGT_FAIL(LHGT_E_FORMAT, "paired-end reads not consistent: first records of %s and %s differ", fq1, fq2);
        }
        if (lines % 4 == 1) {
            long n = lines / 4;
            bool keep = ratio_percent >= 100.0 || (double)ctx->random_array[n % LHGT_MAX_RANDOM] < ratio_percent;
            if (keep && (n / shard_block) % shard_world == shard_rank) {
                if (la > LHGT_MAX_READ_LEN || lb > LHGT_MAX_READ_LEN)
                    LHGT_FAIL(LHGT_E_FORMAT, "read %ld longer than %d bases (the reference's buffers, E:1004)", n, LHGT_MAX_READ_LEN);
                s1.insert(s1.end(), a, a + la);
                s2.insert(s2.end(), b, b + lb);
                o1.push_back(s1.size());
                o2.push_back(s2.size());
                cnt2.push_back(sb <= size1 ? 1 : 0);  // quirk Q4
                kept++;
                if ((long)o1.size() - 1 >= CHUNK) LHGT_TRY(flush());
            }
        }
        lines++;
    }
    {   // the reference reads fq2 on its own in phase A; a longer fq2 would be counted but never voted
        size_t extra_len, extra_start;
        const uint8_t* extra;
        if (c2.next(&extra, &extra_len, &extra_start))
            LHGT_FAIL(LHGT_E_FORMAT, "%s has more lines than %s", fq2, fq1);
    }
    LHGT_TRY(flush());
    if (n_pairs_seen) *n_pairs_seen = lines / 4;
    if (n_pairs_kept) *n_pairs_kept = kept;
    return LHGT_OK;
}

// read_ref (E:727-886): header of 300 words (word j = cc[j] | cc[j+1] << 16, the reference
// writes 4 bytes from a short array), then per contig [len][hashes]; genome.len.txt beside it.
int lhgt_index_build(lhgt_ctx* ctx, const char* fasta_path, const char* index_path, const char* genome_len_path,
                     long* n_contigs, long* n_bases) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
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
    std::vector<uint32_t> host;
    uint32_t* d_out = nullptr;
    size_t d_cap = 0;
    const int k = ctx->k, e = ctx->e;
    int rc = for_each_contig(fasta_path, k, [&](const std::string& name, long ref_index, const uint8_t* seq, long len, long cum) -> int {
        fprintf(lenf, "%s\t%ld\t%ld\t%ld\n", name.c_str(), ref_index, len, cum);
        size_t need = (size_t)(len - k + 1) * e;
        if (need > d_cap) {
            if (d_out) hipFree(d_out);
            d_cap = need + need / 4;
            LHGT_HIP(hipMalloc(&d_out, d_cap * 4));
        }
        LHGT_TRY(hash_contig_to_device(ctx, seq, len, d_out, nullptr));
        host.resize(need);
        LHGT_HIP(hipMemcpyAsync(host.data(), d_out, need * 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        uint32_t u = (uint32_t)len;
        if (fwrite(&u, 4, 1, idx) != 1 || fwrite(host.data(), 4, need, idx) != need)
            LHGT_FAIL(LHGT_E_IO, "short write to %s", index_path);
        contigs++;
        bases += len;
        return LHGT_OK;
    });
    if (d_out) hipFree(d_out);
    fclose(idx);
    fclose(lenf);
    if (n_contigs) *n_contigs = contigs;
    if (n_bases) *n_bases = bases;
    return rc;
}

int lhgt_index_load(lhgt_ctx* ctx, const char* index_path, long* n_contigs, long* n_bases) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx || !index_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    Mapped m;
    LHGT_TRY(m.open(index_path));
    if (m.n < 4 * LHGT_CODER_SLOTS || m.n % 4) LHGT_FAIL(LHGT_E_FORMAT, "%s: not an index file (size %zu)", index_path, m.n);
    const uint32_t* w = (const uint32_t*)m.p;
    int16_t cc[LHGT_CODER_SLOTS];
    for (int i = 0; i < LHGT_CODER_SLOTS; i++) cc[i] = (int16_t)w[i];  // saved_random_coder: low half of each word
    LHGT_TRY(lhgt_coder_set(ctx, cc));
    LHGT_TRY(index_install(ctx, w + LHGT_CODER_SLOTS, m.n / 4 - LHGT_CODER_SLOTS, false));
    if (n_contigs) *n_contigs = (long)ctx->contigs.size();
    if (n_bases) *n_bases = (long)ctx->n_pos;
    return LHGT_OK;
}

}  // extern "C"
