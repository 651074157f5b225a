// host_faidx.cpp -- SURVEY.md 8(f) rank 3: BED -> extracted FASTA, the `samtools faidx -r $interval.bed $ref > $ID.specific.ref.fasta`
// step of the reference's scripts/pipeline.sh:37, and the `.fai` index `samtools faidx $ref` writes for it
// (scripts/infer_HGT_breakpoint.py:156).  Host only (mmap + memcpy): the reference is read where it lies, no second pass.
//
// samtools / htslib are not part of /root/reference and not installed in the image, so this row's parity is UNPINNED: the
// format follows the published behaviour of samtools 1.x faidx (htslib faidx.c):
//   .fai line   = NAME \t LENGTH \t OFFSET \t LINEBASES \t LINEWIDTH     NAME = header up to the first white space,
//                 OFFSET = byte of the first base, LINEBASES/LINEWIDTH from the first sequence line; every line of a
//                 sequence but the last must have that length ("Different line length in sequence" otherwise)
//   region      = NAME:BEG-END, 1-based inclusive; END past the sequence is truncated; the whole string is tried as a name first
//   output      = ">REGION\n" then the bases as stored (case kept), 60 per line
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include "lhgt_common.hpp"

namespace lhgt {

struct FaiEntry {
    std::string name;
    long length = 0;
    size_t offset = 0;
    long linebases = 0, linewidth = 0;
};

struct FaMap {
    const uint8_t* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~FaMap() {
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
        }
        return LHGT_OK;
    }
};

static inline bool is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// one pass over the FASTA: the table `samtools faidx` would write
static int fai_scan(const FaMap& m, std::vector<FaiEntry>* out) {
    out->clear();
    FaiEntry cur;
    bool have = false, first_line = true, short_seen = false;
    size_t pos = 0;
    while (pos < m.n) {
        const uint8_t* st = m.p + pos;
        const uint8_t* nl = (const uint8_t*)memchr(st, '\n', m.n - pos);
        const size_t raw = nl ? (size_t)(nl - st) : m.n - pos;          // bytes before the '\n'
        const size_t width = nl ? raw + 1 : raw;                        // bytes the line occupies
        if (raw > 0 && st[0] == '>') {
            if (have) out->push_back(cur);
            cur = FaiEntry();
            size_t e = 1;
            while (e < raw && !is_space(st[e])) e++;
            cur.name.assign((const char*)st + 1, e - 1);
            cur.offset = pos + width;
            have = true;
            first_line = true;
            short_seen = false;
        } else if (have) {
            size_t bases = raw;
            while (bases > 0 && is_space(st[bases - 1])) bases--;       // "\r\n" files: the CR belongs to the line width
            if (bases > 0 || raw > 0) {
                if (short_seen && bases > 0) LHGT_FAIL(LHGT_E_FORMAT, "Different line length in sequence '%s'", cur.name.c_str());
                if (first_line) {
                    cur.linebases = (long)bases;
                    cur.linewidth = (long)width;
                    first_line = false;
                } else if ((long)bases != cur.linebases || (long)width != cur.linewidth) {
                    if ((long)bases > cur.linebases) LHGT_FAIL(LHGT_E_FORMAT, "Different line length in sequence '%s'", cur.name.c_str());
                    short_seen = true;                                    // allowed once: the last line of the sequence
                }
                cur.length += (long)bases;
            } else if (!first_line) short_seen = true;                    // a blank line may only end a sequence
        }
        pos += width;
    }
    if (have) out->push_back(cur);
    return LHGT_OK;
}

// copy bases [beg0, end0) of an indexed sequence, skipping line terminators
static void fai_fetch(const FaMap& m, const FaiEntry& s, long beg0, long end0, std::string* seq) {
    seq->clear();
    if (beg0 >= end0 || s.linebases <= 0) return;
    seq->reserve((size_t)(end0 - beg0));
    long pos = beg0;
    while (pos < end0) {
        const long line = pos / s.linebases, col = pos % s.linebases;
        long take = s.linebases - col;
        if (take > end0 - pos) take = end0 - pos;
        seq->append((const char*)m.p + s.offset + (size_t)line * (size_t)s.linewidth + (size_t)col, (size_t)take);
        pos += take;
    }
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

int lhgt_faidx_build(const char* fasta_path, const char* fai_path, long* n_sequences) {
    if (!fasta_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    FaMap m;
    LHGT_TRY(m.open(fasta_path));
    std::vector<FaiEntry> tab;
    LHGT_TRY(fai_scan(m, &tab));
    if (fai_path) {
        FILE* f = fopen(fai_path, "w");
        if (!f) LHGT_FAIL(LHGT_E_IO, "cannot write %s", fai_path);
        for (const FaiEntry& s : tab) fprintf(f, "%s\t%ld\t%zu\t%ld\t%ld\n", s.name.c_str(), s.length, s.offset, s.linebases, s.linewidth);
        fclose(f);
    }
    if (n_sequences) *n_sequences = (long)tab.size();
    return LHGT_OK;
}

int lhgt_faidx_extract(const char* fasta_path, const char* regions_path, const char* out_path, int line_width, long* n_regions,
                       long* n_bases) {
    if (!fasta_path || !regions_path || !out_path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (line_width <= 0) line_width = 60;
    FaMap m, r;
    LHGT_TRY(m.open(fasta_path));
    LHGT_TRY(r.open(regions_path));
    std::vector<FaiEntry> tab;
    LHGT_TRY(fai_scan(m, &tab));
    std::unordered_map<std::string, size_t> by_name;
    for (size_t i = 0; i < tab.size(); i++) by_name.emplace(tab[i].name, i);   // first of equal names wins, as in htslib
    const bool to_stdout = strcmp(out_path, "-") == 0;
    FILE* f = to_stdout ? stdout : fopen(out_path, "w");
    if (!f) LHGT_FAIL(LHGT_E_IO, "cannot write %s", out_path);
    long regions = 0, bases = 0;
    int rc = LHGT_OK;
    std::string seq, reg;
    size_t pos = 0;
    while (pos < r.n && rc == LHGT_OK) {
        const uint8_t* st = r.p + pos;
        const uint8_t* nl = (const uint8_t*)memchr(st, '\n', r.n - pos);
        size_t len = nl ? (size_t)(nl - st) : r.n - pos;
        pos += nl ? len + 1 : len;
        while (len > 0 && is_space(st[len - 1])) len--;
        if (len == 0) continue;
        reg.assign((const char*)st, len);
        long beg = 1, end = -1;                       // 1-based inclusive; end < 0 = to the end of the sequence
        auto it = by_name.find(reg);
        if (it == by_name.end()) {
            const size_t colon = reg.rfind(':');
            bool ok = colon != std::string::npos;
            if (ok) {
                it = by_name.find(reg.substr(0, colon));
                ok = it != by_name.end();
            }
            if (ok) {
                std::string span;
                for (size_t i = colon + 1; i < reg.size(); i++) if (reg[i] != ',') span.push_back(reg[i]);
                char* e1 = nullptr;
                const long b = strtol(span.c_str(), &e1, 10);
                if (e1 == span.c_str()) ok = false;
                else {
                    beg = b;
                    if (*e1 == '-') {
                        if (e1[1]) {
                            char* e2 = nullptr;
                            end = strtol(e1 + 1, &e2, 10);
                            if (*e2 || e2 == e1 + 1) ok = false;
                        }
                    } else if (*e1) ok = false;
                    else end = -1;
                }
            }
            if (!ok) {
                set_error("Failed to fetch sequence in %s", reg.c_str());
                rc = LHGT_E_ARG;
                break;
            }
        }
        const FaiEntry& s = tab[it->second];
        long beg0 = beg < 1 ? 0 : beg - 1;
        long end0 = (end < 0 || end > s.length) ? s.length : end;       // truncated at the end of the sequence
        if (beg0 > end0) beg0 = end0;
        fai_fetch(m, s, beg0, end0, &seq);
        fprintf(f, ">%s\n", reg.c_str());
        for (size_t i = 0; i < seq.size(); i += (size_t)line_width) {
            const size_t w = seq.size() - i < (size_t)line_width ? seq.size() - i : (size_t)line_width;
            fwrite(seq.data() + i, 1, w, f);
            fputc('\n', f);
        }
        regions++;
        bases += (long)seq.size();
    }
    if (to_stdout) fflush(f);
    else fclose(f);
    if (n_regions) *n_regions = regions;
    if (n_bases) *n_bases = bases;
    return rc;
}

}  // extern "C"
