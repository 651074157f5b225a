/* lhgt_oracle.c -- CPU restatement of LocalHGT's k-mer sketch->peak path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under localhgt_amd/ may import, link or execute
 * this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / the timed CPU baseline, never as the product.
 *
 * PARITY STATUS: pinned.  The reference's own tests hold no vectors for this path
 * (SURVEY.md 4), so this restatement is pinned against outputs of the reference itself,
 * compiled from /root/reference by oracle/build_ref.sh (oracle/_ref/extract_ref_z) and run
 * in the build container by tests/golden/make_golden.py; the committed fixtures under
 * tests/golden/ carry those outputs.
 *
 * All file:line citations are to /root/reference/src/extract_ref_normal_peak.cpp ("E").
 * Contract: `-t 1` semantics (Q2/Q5), zero tail of the per-contig hit
 * arrays (Q1), fq2 counting cut at size(fq1) (Q4), hash 0 == "invalid" in the index (Q6),
 * sequential contig numbering in phase B (Q7), peak id 0 invisible (Q5).
 * Threads here only shard reads; saturating counters use CAS so any thread count gives
 * the t=1 result.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

#define ORC_CODER_SLOTS 300       /* E:21, E:1186 */
#define ORC_MAX_RANDOM 50000000L  /* E:40 */
#define ORC_MAX_READ 500          /* E:1004-1005 stack buffers */

/* ---------------------------------------------------------------- base maps (E:1109-1180) */
/* three 1-bit projections of a base; 5 = not a base.  map0: A,T->1 C,G->0; map1: A,C->1;
 * map2: A,G->1 (upper and lower case). */
static int orc_proj(int map, int ch) {
    int base;
    switch (ch) {
        case 'A': case 'a': base = 0; break;
        case 'C': case 'c': base = 1; break;
        case 'G': case 'g': base = 2; break;
        case 'T': case 't': base = 3; break;
        default: return 5;
    }
    static const int bit[3][4] = {{1, 0, 0, 1}, {1, 1, 0, 0}, {1, 0, 1, 0}};
    return bit[map][base];
}
/* complement (E:1165-1180): always upper case, 0 for anything else */
static int orc_comp(int ch) {
    switch (ch) {
        case 'A': case 'a': return 'T';
        case 'T': case 't': return 'A';
        case 'C': case 'c': return 'G';
        case 'G': case 'g': return 'C';
        default: return 0;
    }
}

/* ---------------------------------------------------------------- hash H (E:1052-1081, 786-811) */
/* e hashes of the k-mer starting at s.  Returns 1 when every base is valid.  The O(k)
 * inner loop and the u32 wrap-around of base[z] = 2^(k-1-z) are kept literally. */
int orc_hash_kmer(const unsigned char* s, int k, int e, const short* cc, uint32_t* out) {
    int all_valid = 1;
    for (int i = 0; i < e; i++) {
        uint32_t fwd = 0, rc = 0;
        int valid = 1;
        for (int z = 0; z < k; z++) {
            int m = orc_proj(cc[z * e + i], s[z]);
            if (m == 5) { valid = 0; break; }
            int n = orc_proj(cc[(k - 1 - z) * e + i], orc_comp(s[z]));
            fwd += (uint32_t)m << (k - 1 - z);
            rc += (uint32_t)n << z;
        }
        out[i] = fwd > rc ? rc : fwd;
        if (!valid) all_valid = 0;
    }
    return all_valid;
}

/* ---------------------------------------------------------------- RNG R (E:1182-1222, 1332-1340) */
void orc_srand(unsigned seed) { srand(seed); }

/* random_coder: t = e/3+1 draws of rand()%6 per position, rows of permu concatenated */
void orc_random_coder(int k, int e, short* cc) {
    static const short permu[18] = {0, 1, 2, 0, 2, 1, 1, 2, 0, 1, 0, 2, 2, 0, 1, 2, 1, 0};
    for (int i = 0; i < ORC_CODER_SLOTS; i++) cc[i] = 100;
    int t = e / 3 + 1;
    short row[64];
    for (int j = 0; j < k; j++) {
        for (int z = 0; z < t; z++) {
            int r = rand() % 6;
            for (int w = 0; w < 3; w++) row[3 * z + w] = permu[r * 3 + w];
        }
        for (int i = 0; i < e; i++) cc[j * e + i] = row[i];
    }
}

/* get_random: float32((rand()%100000)/1000.0) */
float* orc_sampling_array(long n) {
    float* a = (float*)malloc(sizeof(float) * (size_t)n);
    for (long i = 0; i < n; i++) a[i] = (float)((rand() % 100000) / 1000.0);
    return a;
}
void orc_free(void* p) { free(p); }

/* ---------------------------------------------------------------- small file helpers */
typedef struct { unsigned char* p; long n; } orc_buf;
static orc_buf orc_slurp(const char* path) {
    orc_buf b = {NULL, -1};
    FILE* f = fopen(path, "rb");
    if (!f) return b;
    fseek(f, 0, SEEK_END);
    b.n = ftell(f);
    fseek(f, 0, SEEK_SET);
    b.p = (unsigned char*)malloc((size_t)b.n + 1);
    if (b.n && fread(b.p, 1, (size_t)b.n, f) != (size_t)b.n) { free(b.p); b.p = NULL; b.n = -1; }
    fclose(f);
    return b;
}
/* getline-style iteration: returns 0 at EOF; a trailing '\n' does not create an empty line */
static int orc_next_line(const orc_buf* b, long* cur, const unsigned char** s, long* len) {
    if (*cur >= b->n) return 0;
    const unsigned char* st = b->p + *cur;
    const unsigned char* nl = (const unsigned char*)memchr(st, '\n', (size_t)(b->n - *cur));
    *s = st;
    if (nl) { *len = nl - st; *cur += *len + 1; }
    else { *len = b->n - *cur; *cur = b->n; }
    return 1;
}
long orc_file_size(const char* path) {
    struct stat sb;
    if (stat(path, &sb)) return -1;
    return (long)sb.st_size;
}
/* get_read_ID (E:303-311): text before the first '/', then before ' ', then before '\t' */
static long orc_read_id_len(const unsigned char* s, long len) {
    long n = len;
    for (long i = 0; i < n; i++) if (s[i] == '/') { n = i; break; }
    for (long i = 0; i < n; i++) if (s[i] == ' ') { n = i; break; }
    for (long i = 0; i < n; i++) if (s[i] == '\t') { n = i; break; }
    return n;
}

/* ---------------------------------------------------------------- index writer I (E:727-886) */
static void orc_emit_contig(FILE* idx, FILE* lenf, const char* name, int ref_index,
                            const unsigned char* seq, long len, long cum, int k, int e, const short* cc) {
    if (len <= k) return;  /* E:772, 836 */
    fprintf(lenf, "%s\t%d\t%ld\t%ld\n", name, ref_index, len, cum);
    uint32_t u = (uint32_t)len;
    fwrite(&u, 4, 1, idx);
    long nk = len - k + 1;
    uint32_t* row = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nk * e);
    uint32_t h[16];
    for (long j = 0; j < nk; j++) {
        /* a bad base invalidates every hash of the k-mer; invalid is stored as 0 (E:808-810) */
        int ok = orc_hash_kmer(seq + j, k, e, cc, h);
        for (int i = 0; i < e; i++) row[j * e + i] = ok ? h[i] : 0;
    }
    fwrite(row, 4, (size_t)nk * e, idx);
    free(row);
}

/* Returns number of contigs written, <0 on error.  Header: 300 u32 words, word j =
 * cc[j] | cc[j+1]<<16 (the reference writes 4 bytes starting at a short, E:755-757);
 * the word after the last short is whatever follows the array: 0 here. */
int orc_index_build(const char* fasta, const char* index_path, const char* len_path,
                    int k, int e, const short* cc) {
    orc_buf fa = orc_slurp(fasta);
    if (fa.n < 0) return -1;
    FILE* idx = fopen(index_path, "wb");
    FILE* lenf = fopen(len_path, "w");
    if (!idx || !lenf) return -2;
    for (int j = 0; j < ORC_CODER_SLOTS; j++) {
        uint16_t lo = (uint16_t)cc[j];
        uint16_t hi = j + 1 < ORC_CODER_SLOTS ? (uint16_t)cc[j + 1] : 0;
        uint32_t w = (uint32_t)lo | ((uint32_t)hi << 16);
        fwrite(&w, 4, 1, idx);
    }
    unsigned char* seq = (unsigned char*)malloc((size_t)fa.n + 1);
    long slen = 0, cum = 0, cur = 0, len;
    const unsigned char* s;
    char name[4096] = "start", pending[4096] = "start";
    int ref_index = 0, written = 0;
    while (orc_next_line(&fa, &cur, &s, &len)) {
        if (len > 0 && s[0] == '>') {
            /* the sequence accumulated so far belongs to `pending` (E:763-764) */
            strcpy(name, pending);
            long idl = orc_read_id_len(s, len);
            long nl = idl > 1 ? idl - 1 : 0;
            if (nl > 4095) nl = 4095;
            memcpy(pending, s + 1, (size_t)nl);
            pending[nl] = 0;
            cum += slen;
            if (slen > k) { orc_emit_contig(idx, lenf, name, ref_index, seq, slen, cum, k, e, cc); written++; }
            ref_index += 1;  /* counts skipped contigs too (E:825, quirk Q7) */
            slen = 0;
        } else {
            memcpy(seq + slen, s, (size_t)len);
            slen += len;
        }
    }
    cum += slen;
    if (slen > k) { orc_emit_contig(idx, lenf, pending, ref_index, seq, slen, cum, k, e, cc); written++; }
    fclose(idx);
    fclose(lenf);
    free(seq);
    free(fa.p);
    return written;
}

/* saved_random_coder (E:1224-1242): low 16 bits of each header word */
int orc_index_header(const char* index_path, short* cc) {
    FILE* f = fopen(index_path, "rb");
    if (!f) return -1;
    uint32_t w[ORC_CODER_SLOTS];
    size_t got = fread(w, 4, ORC_CODER_SLOTS, f);
    fclose(f);
    if (got != ORC_CODER_SLOTS) return -2;
    for (int i = 0; i < ORC_CODER_SLOTS; i++) cc[i] = (short)w[i];
    return 0;
}

/* ---------------------------------------------------------------- sampling ratio (E:1244-1270, 1392-1398) */
double orc_sam_ratio(const char* fq1, double sample) {
    if (sample <= 1) return 100 * sample;
    orc_buf b = orc_slurp(fq1);
    if (b.n < 0) return -1;
    long cur = 0, len, i = 0, bases = 0;
    const unsigned char* s;
    while (orc_next_line(&b, &cur, &s, &len)) {
        if (i % 4 == 1) bases += len;
        i++;
    }
    free(b.p);
    bases *= 2;
    return 100 * sample / (double)bases;
}

/* ---------------------------------------------------------------- kept-read list (E:1020-1044, 356-419) */
typedef struct { const unsigned char* s; int len; } orc_seq;
typedef struct { orc_seq* v; long n; } orc_seqs;

/* Sequence lines of a FASTQ in record order; `byte_limit` reproduces the `add_size > end`
 * stop (E:1022-1026): a line is consumed only while the byte offset of its start is <= limit.
 * keep[] (optional) receives the sampling decision per record. */
static orc_seqs orc_scan_fastq(const orc_buf* b, long byte_limit) {
    orc_seqs r = {NULL, 0};
    long cap = 1 << 16, cur = 0, len, lines = 0;
    r.v = (orc_seq*)malloc(sizeof(orc_seq) * (size_t)cap);
    const unsigned char* s;
    for (;;) {
        long start = cur;
        if (!orc_next_line(b, &cur, &s, &len)) break;
        if (start > byte_limit) break;
        if (lines % 4 == 1) {
            if (r.n == cap) { cap *= 2; r.v = (orc_seq*)realloc(r.v, sizeof(orc_seq) * (size_t)cap); }
            r.v[r.n].s = s;
            r.v[r.n].len = (int)len;
            r.n++;
        }
        lines++;
    }
    return r;
}
static int orc_keep(long ordinal, const float* rnd, double ratio) {
    if (!rnd) return 1;  /* caller promises ratio >= 100: every r in [0, 99.999] passes */
    float r = rnd[ordinal % ORC_MAX_RANDOM];
    return (double)r < ratio;
}

/* ---------------------------------------------------------------- phase A (E:981-1107) */
typedef struct {
    const orc_seqs* seqs; long lo, hi;
    int k, e; const short* cc; const float* rnd; double ratio;
    uint8_t* table; long kept;
    int too_long;   /* a SAMPLED read longer than the reference's buffers (E:1004-1005, filled only under `r < down_sam_ratio`, E:1044) */
} orc_count_job;

static void* orc_count_worker(void* arg) {
    orc_count_job* j = (orc_count_job*)arg;
    uint32_t h[16];
    for (long n = j->lo; n < j->hi; n++) {
        if (!orc_keep(n, j->rnd, j->ratio)) continue;
        j->kept++;
        const orc_seq* q = &j->seqs->v[n];
        if (q->len > ORC_MAX_READ) { j->too_long = 1; return NULL; }
        for (int p = 0; p + j->k <= q->len; p++) {
            if (!orc_hash_kmer(q->s + p, j->k, j->e, j->cc, h)) continue;
            for (int i = 0; i < j->e; i++) {
                /* if (T[h] < 3) T[h]++  (E:1082-1084), made race-free */
                uint8_t* slot = &j->table[h[i]];
                uint8_t old = __atomic_load_n(slot, __ATOMIC_RELAXED);
                while (old < 3 && !__atomic_compare_exchange_n(slot, &old, (uint8_t)(old + 1), 1,
                                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            }
        }
    }
    return NULL;
}

/* Count one FASTQ into table[2^k] (u8, values 0..3).  byte_limit = size(fq1) for both mates
 * (E:1419, 1438-1445, quirk Q4).  rnd may be NULL when ratio >= 100.  Returns reads kept. */
long orc_count_fastq(const char* fq, long byte_limit, int k, int e, const short* cc,
                     double ratio, const float* rnd, uint8_t* table, int threads) {
    orc_buf b = orc_slurp(fq);
    if (b.n < 0) return -1;
    orc_seqs seqs = orc_scan_fastq(&b, byte_limit);
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    orc_count_job* jobs = (orc_count_job*)calloc((size_t)threads, sizeof(orc_count_job));
    for (int t = 0; t < threads; t++) {
        orc_count_job jb = {&seqs, seqs.n * t / threads, seqs.n * (t + 1) / threads, k, e, cc, rnd, ratio, table, 0, 0};
        jobs[t] = jb;
        pthread_create(&th[t], NULL, orc_count_worker, &jobs[t]);
    }
    long kept = 0;
    int too_long = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); kept += jobs[t].kept; too_long |= jobs[t].too_long; }
    free(th); free(jobs); free(seqs.v); free(b.p);
    return too_long ? -3 : kept;   /* the reference overruns its stack buffers there: undefined, refused */
}

/* ---------------------------------------------------------------- phase B (E:888-979, 550-725, 239-301) */
typedef struct {
    int32_t* loci;      /* [2*max_peak]: ref_index, pos   (E:215) */
    uint32_t* peak_kmer; /* [2^k]                         (E:217) */
    long n_peaks;       /* running id counter (t=1)       (E:232-235) */
    long max_peak;
    long slided, extracted;
    int too_many;
} orc_peaks;

/* add_peak + merge_peak: new peak unless same contig and same 50-bp bucket as the last
 * added one; k-mers with hit>0 at the peak position point at the (last) id. */
static void orc_add_peak(orc_peaks* P, int ref_index, int pos, const uint32_t* hidx,
                         const uint8_t* hit, int ref_len, int k, int e) {
    long my = P->n_peaks;
    int merged = 0;
    if (my > 0 && ref_index == P->loci[2 * my - 2] && pos / 50 == P->loci[2 * my - 1] / 50) merged = 1;
    long id = merged ? my - 1 : my;
    if (!merged) {
        if (my >= P->max_peak) { P->too_many = 1; return; }  /* E:272-274 (the reference overruns) */
        P->loci[2 * my] = ref_index;
        P->loci[2 * my + 1] = pos;
    }
    if (pos >= 0 && pos <= ref_len - k + 1)  /* E:247, 262 */
        for (int p = 0; p < e; p++)
            if (hit[(long)e * pos + p] > 0) P->peak_kmer[hidx[(long)e * pos + p]] = (uint32_t)id;
    if (!merged) P->n_peaks = my + 1;
}

static void orc_slide_window(orc_peaks* P, const uint8_t* hit, const uint32_t* hidx, int ref_len,
                             int ref_index, int k, int e, float hit_ratio, float match_ratio,
                             uint8_t* dbg_flags) {
    const int window = 500, w = 5, DIFF = 2;            /* E:556, 32, 31 */
    const int skip_n = 2 * k, skip_s = k;               /* E:1377-1378 */
    int one_min = (int)(window * hit_ratio);            /* float32 product, truncated (E:559-560) */
    int three_min = (int)(window * match_ratio);
    short* single = (short*)calloc((size_t)ref_len, sizeof(short));
    short* trio = (short*)calloc((size_t)ref_len, sizeof(short));
    uint8_t* peak = (uint8_t*)calloc((size_t)ref_len, 1);
    int* iv = (int*)calloc((size_t)(2 * (ref_len / window + 2)), sizeof(int));
    int n_iv = 0, one = 0, three = 0, conti = 0, good = 0, start = 0, end = 0;
    for (int j = 0; j < ref_len; j++) {
        int hc = 0;
        for (int p = 0; p < e; p++) if (hit[(long)e * j + p] == 3) hc++;  /* least_depth, E:580 */
        trio[j] = hc == e;
        single[j] = hc > 0;
        if (j < window) { one += single[j]; three += trio[j]; }
        else { one += single[j] - single[j - window]; three += trio[j] - trio[j - window]; }
        good = one >= one_min && three >= three_min;
        if (!conti && good) { start = j - 2 * window; if (start < 1) start = 1; conti = 1; }
        if (conti && !good) {
            end = j + 2 * window;
            if (end > ref_len) end = ref_len;
            if (n_iv > 0 && start - iv[2 * n_iv - 1] < window) iv[2 * n_iv - 1] = end;
            else { iv[2 * n_iv] = start; iv[2 * n_iv + 1] = end; n_iv++; }
            conti = 0;
        }
        /* contrast test (E:644-671); `left` is updated literally, it is not a true slide */
        if (j > skip_n + 2 * w) {
            int left = 0, right = 0;
            for (int n = 0; n < w; n++) right += single[j - n];
            for (int m = skip_s; m < skip_n; m++) {
                if (m == skip_s) for (int n = 0; n < w; n++) left += single[j - w - n];
                else left = left - single[j - m - w + 1] + single[j - w - w + 1 - m];
                int diff = left - right;
                if (diff >= DIFF) peak[j - m - w] = 1;
                if (diff <= -DIFF) peak[j] = 1;
            }
        }
    }
    if (conti && good) {
        end = ref_len;
        if (n_iv > 0 && start - iv[2 * n_iv - 1] < window) iv[2 * n_iv - 1] = end;
        else { iv[2 * n_iv] = start; iv[2 * n_iv + 1] = end; n_iv++; }
    }
    for (int i = 0; i < n_iv; i++) {
        for (int j = iv[2 * i]; j < iv[2 * i + 1]; j++) {
            if (dbg_flags) dbg_flags[j] |= 4;
            if (peak[j]) orc_add_peak(P, ref_index, j, hidx, hit, ref_len, k, e);
        }
        P->extracted += iv[2 * i + 1] - iv[2 * i];
    }
    if (dbg_flags) for (int j = 0; j < ref_len; j++) dbg_flags[j] |= (uint8_t)(single[j] | (trio[j] << 1) | (peak[j] << 3));
    free(single); free(trio); free(peak); free(iv);
}

/* Scan the whole index sequentially (t=1): contigs numbered 1,2,... in file order (E:905, 963).
 * dbg_flags (optional, one byte per reference position in index order): bit0 single,
 * bit1 trio, bit2 inside a good interval, bit3 peak flag.  Returns raw peak count or <0. */
long orc_ref_scan(const char* index_path, const uint8_t* table, int k, int e, float hit_ratio,
                  float match_ratio, long max_peak, int32_t* loci, uint32_t* peak_kmer,
                  uint8_t* dbg_flags, long* extracted_out) {
    FILE* f = fopen(index_path, "rb");
    if (!f) return -1;
    fseek(f, 4L * ORC_CODER_SLOTS, SEEK_SET);
    orc_peaks P = {loci, peak_kmer, 0, max_peak, 0, 0, 0};
    int ref_index = 1;
    long flat = 0;
    uint32_t ref_len_u;
    while (fread(&ref_len_u, 4, 1, f) == 1) {
        int ref_len = (int)ref_len_u;
        long nk = (long)ref_len - k + 1;
        uint32_t* hidx = (uint32_t*)calloc((size_t)ref_len * e, 4);
        uint8_t* hit = (uint8_t*)calloc((size_t)ref_len * e, 1);   /* zero tail: quirk Q1 */
        if (fread(hidx, 4, (size_t)nk * e, f) != (size_t)nk * e) { free(hidx); free(hit); fclose(f); return -2; }
        for (long q = 0; q < nk * e; q++) hit[q] = hidx[q] ? table[hidx[q]] : 0;  /* E:933-945, Q6 */
        orc_slide_window(&P, hit, hidx, ref_len, ref_index, k, e, hit_ratio, match_ratio,
                         dbg_flags ? dbg_flags + flat : NULL);
        P.slided += ref_len;
        flat += ref_len;
        free(hidx); free(hit);
        ref_index++;
    }
    fclose(f);
    if (extracted_out) *extracted_out = P.extracted;
    if (P.too_many) return -3;
    return P.n_peaks;
}

/* ---------------------------------------------------------------- phase C (E:313-506, 91-202) */
typedef struct { int chr, count, first_id; } orc_chr;
typedef struct { orc_chr v[2 * ORC_MAX_READ]; int n; int base_hits; } orc_vote_state;

static orc_chr* orc_find(orc_vote_state* S, int chr) {
    for (int i = 0; i < S->n; i++) if (S->v[i].chr == chr) return &S->v[i];
    return NULL;
}
/* judge_base (E:118-159) for one k-mer offset: ids[i] = peak id seen by hash i (0 = none) */
static void orc_judge(orc_vote_state* S, const uint32_t* ids, int e, const int32_t* loci) {
    int sel_chr = 0, sel_id = 0, sel_num = 0, any = 0;
    for (int i = 0; i < e; i++) {
        if (!ids[i]) continue;
        any = 1;
        int chr = loci[2 * (long)ids[i]];
        orc_chr* c = orc_find(S, chr);
        if (c) {
            if (c->count >= sel_num) { sel_id = (int)ids[i]; sel_chr = chr; sel_num = c->count; }
        } else if (sel_id == 0) { sel_id = (int)ids[i]; sel_chr = chr; sel_num = 0; }
    }
    if (!any) return;
    orc_chr* c = orc_find(S, sel_chr);
    if (c) c->count++;
    else { S->v[S->n].chr = sel_chr; S->v[S->n].count = 1; S->v[S->n].first_id = sel_id; S->n++; }
    S->base_hits++;
}
static int orc_chr_cmp(const void* a, const void* b) { return ((const orc_chr*)a)->chr - ((const orc_chr*)b)->chr; }
/* check_split (E:161-202): contigs in ascending order (std::map), top-2 counts among those >= 6 */
static void orc_check_split(orc_vote_state* S, uint8_t* peak_filter) {
    qsort(S->v, (size_t)S->n, sizeof(orc_chr), orc_chr_cmp);
    int largest = 0, second = 0, n_f = 0;
    for (int i = 0; i < S->n; i++) {
        int c = S->v[i].count;
        if (c < 6) continue;  /* MIN_BASE_NUM, E:29 */
        n_f++;
        if (c >= largest) { second = largest; largest = c; }
        else if (c >= second) second = c;
    }
    if (n_f < 2) return;
    for (int i = 0; i < S->n; i++) {
        int c = S->v[i].count;
        if (c < 6 || (c != largest && c != second)) continue;
        uint8_t* slot = &peak_filter[S->v[i].first_id];
        uint8_t old = __atomic_load_n(slot, __ATOMIC_RELAXED);  /* if (<254) ++, race-free (E:194-196) */
        while (old < 254 && !__atomic_compare_exchange_n(slot, &old, (uint8_t)(old + 1), 1,
                                                         __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    }
}

typedef struct {
    const orc_seqs *s1, *s2; long lo, hi;
    int k, e; const short* cc; const float* rnd; double ratio;
    const uint32_t* peak_kmer; const int32_t* loci; uint8_t* peak_filter; long kept;
    int too_long;   /* a sampled pair with a mate longer than the reference's buffers */
} orc_vote_job;

static void orc_vote_mate(orc_vote_state* S, const orc_seq* q, const orc_vote_job* j) {
    uint32_t h[16], ids[16];
    for (int p = 0; p + j->k <= q->len; p++) {
        int valid = orc_hash_kmer(q->s + p, j->k, j->e, j->cc, h);
        for (int i = 0; i < j->e; i++) ids[i] = valid ? j->peak_kmer[h[i]] : 0;  /* E:454-457 */
        orc_judge(S, ids, j->e, j->loci);
    }
}
static void* orc_vote_worker(void* arg) {
    orc_vote_job* j = (orc_vote_job*)arg;
    orc_vote_state* S = (orc_vote_state*)malloc(sizeof(orc_vote_state));
    for (long n = j->lo; n < j->hi; n++) {
        if (!orc_keep(n, j->rnd, j->ratio)) continue;
        j->kept++;
        if (j->s1->v[n].len > ORC_MAX_READ || j->s2->v[n].len > ORC_MAX_READ) { j->too_long = 1; break; }
        S->n = 0; S->base_hits = 0;
        orc_vote_mate(S, &j->s1->v[n], j);
        orc_vote_mate(S, &j->s2->v[n], j);
        if (S->base_hits >= 6) orc_check_split(S, j->peak_filter);  /* E:496 */
    }
    free(S);
    return NULL;
}

static long orc_vote_chunk(const orc_buf* b1, const orc_buf* b2, long start, long end, const orc_vote_job* j);
/* Lock-step pass over both FASTQs (E:350-359).  Returns pairs kept. */
long orc_vote(const char* fq1, const char* fq2, int k, int e, const short* cc, double ratio,
              const float* rnd, const uint32_t* peak_kmer, const int32_t* loci, uint8_t* peak_filter,
              int threads) {
    orc_buf b1 = orc_slurp(fq1), b2 = orc_slurp(fq2);
    if (b1.n < 0 || b2.n < 0) return -1;
    orc_seqs s1 = orc_scan_fastq(&b1, b1.n), s2 = orc_scan_fastq(&b2, b2.n);
    {   /* first read IDs differ (fq2 is re-scanned for fq1's, E:368-402) or fq2 runs out first (E:356-367): the literal pass */
        long c1 = 0, c2 = 0, l1 = 0, l2 = 0;
        const unsigned char *a = (const unsigned char*)"", *b = (const unsigned char*)"";
        orc_next_line(&b1, &c1, &a, &l1);
        orc_next_line(&b2, &c2, &b, &l2);
        long i1 = orc_read_id_len(a, l1), i2 = orc_read_id_len(b, l2);
        if (s2.n < s1.n || i1 != i2 || memcmp(a, b, (size_t)i1)) {
            orc_vote_job jb = {&s1, &s2, 0, 0, k, e, cc, rnd, ratio, peak_kmer, loci, peak_filter, 0};
            long kept = orc_vote_chunk(&b1, &b2, 0, b1.n, &jb);
            free(s1.v); free(s2.v); free(b1.p); free(b2.p);
            return kept;
        }
    }
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    orc_vote_job* jobs = (orc_vote_job*)calloc((size_t)threads, sizeof(orc_vote_job));
    for (int t = 0; t < threads; t++) {
        orc_vote_job jb = {&s1, &s2, s1.n * t / threads, s1.n * (t + 1) / threads, k, e, cc, rnd, ratio,
                           peak_kmer, loci, peak_filter, 0};
        jobs[t] = jb;
        pthread_create(&th[t], NULL, orc_vote_worker, &jobs[t]);
    }
    long kept = 0;
    int too_long = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); kept += jobs[t].kept; too_long |= jobs[t].too_long; }
    free(th); free(jobs); free(s1.v); free(s2.v); free(b1.p); free(b2.p);
    return too_long ? -3 : kept;
}

/* ---------------------------------------------------------------- phase D (E:515-548) */
/* One thread range (t=1): leading sentinel "1\t1\t1", merge while same contig and gap < 500. */
long orc_write_intervals(const char* path, const int32_t* loci, const uint8_t* peak_filter, long n_peaks) {
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    int start = 1, end = 1, chr = 1;
    long total = 0;
    for (long i = 0; i < n_peaks; i++) {
        if (peak_filter[i] < 1) continue;  /* MIN_READS, E:37 */
        int c = loci[2 * i], pos = loci[2 * i + 1];
        if (chr == c && pos - 500 - end < 500) end = pos + 500;
        else {
            fprintf(f, "%d\t%d\t%d\n", chr, start, end);
            total += end - start;
            chr = c; start = pos - 500; end = pos + 500;
        }
    }
    fprintf(f, "%d\t%d\t%d\n", chr, start, end);
    total += end - start;
    fclose(f);
    return total;
}

/* ---------------------------------------------------------------- whole run (E:1342-1519) */
/* bench.py's cpu_baseline sets this: fault the tables in before the phase timers start, as the reference's
 * memsets do (E:1416, 1458), so the timed phases exclude the fixed page-fault cost.  Tests leave it off. */
static int orc_pretouch = 0;
void orc_set_pretouch(int on) { orc_pretouch = on; }

typedef struct {
    double t_index, t_count, t_scan, t_vote, t_total;
    long pairs_counted, pairs_voted, n_peaks, n_filtered;
} orc_report;

static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

/* The 12-argument contract of extract_ref.  `threads` shards reads only (results are the
 * t=1 results).  Tables are calloc'd, so untouched pages stay unmapped. */
int orc_run(const char* fq1, const char* fq2, const char* fasta, const char* interval_path,
            double hit_ratio_d, double match_ratio_d, int threads, int k, long max_peak, int e,
            unsigned seed, double sample, orc_report* rep) {
    double t0 = orc_now();
    size_t slots = (size_t)1 << k;
    uint8_t* table = (uint8_t*)calloc(slots, 1);
    short cc[ORC_CODER_SLOTS], cc_file[ORC_CODER_SLOTS];
    if (!table) return -1;
    srand(seed);                                         /* E:1386 */
    double ratio = orc_sam_ratio(fq1, sample);           /* E:1392-1398 */
    char index_path[4096], len_path[4096];
    snprintf(index_path, sizeof index_path, "%s.k%d.h%d.index.dat", fasta, k, e);  /* E:1401 */
    snprintf(len_path, sizeof len_path, "%s.genome.len.txt", fasta);
    FILE* probe = fopen(index_path, "rb");
    if (probe) fclose(probe);
    else {                                               /* E:1404-1410: consumes rand() (quirk Q3) */
        orc_random_coder(k, e, cc);
        if (orc_index_build(fasta, index_path, len_path, k, e, cc) < 0) return -2;
    }
    if (orc_index_header(index_path, cc_file)) return -3; /* E:1417 */
    if (orc_pretouch) memset(table, 0, slots);           /* E:1416: pages resident before counting starts */
    double t1 = orc_now();
    float* rnd = ratio >= 100.0 ? NULL : orc_sampling_array(ORC_MAX_RANDOM);  /* E:1422 */
    long size1 = orc_file_size(fq1);
    long c1 = orc_count_fastq(fq1, size1, k, e, cc_file, ratio, rnd, table, threads);
    long c2 = orc_count_fastq(fq2, size1, k, e, cc_file, ratio, rnd, table, threads);
    if (c1 < 0 || c2 < 0) return -4;
    double t2 = orc_now();
    int32_t* loci = (int32_t*)calloc((size_t)max_peak * 2, 4);
    uint8_t* peak_filter = (uint8_t*)calloc((size_t)max_peak, 1);
    uint32_t* peak_kmer = (uint32_t*)calloc(slots, 4);
    if (!loci || !peak_filter || !peak_kmer) return -1;
    if (orc_pretouch) memset(peak_kmer, 0, slots * 4);   /* E:1458 (fixed cost, outside the phase timers) */
    double t2b = orc_now();
    long n_peaks = orc_ref_scan(index_path, table, k, e, (float)hit_ratio_d, (float)match_ratio_d,
                                max_peak, loci, peak_kmer, NULL, NULL);
    if (n_peaks < 0) return -5;
    double t3 = orc_now();
    long voted = orc_vote(fq1, fq2, k, e, cc_file, ratio, rnd, peak_kmer, loci, peak_filter, threads);
    if (voted < 0) return -6;
    double t4 = orc_now();
    orc_write_intervals(interval_path, loci, peak_filter, n_peaks);
    if (rep) {
        rep->t_index = t1 - t0; rep->t_count = t2 - t1; rep->t_scan = t3 - t2b; rep->t_vote = t4 - t3;
        rep->t_total = orc_now() - t0;
        rep->pairs_counted = c1; rep->pairs_voted = voted; rep->n_peaks = n_peaks;
        long nf = 0;
        for (long i = 0; i < n_peaks; i++) nf += peak_filter[i] >= 1;
        rep->n_filtered = nf;
    }
    free(table); free(loci); free(peak_filter); free(peak_kmer); free(rnd);
    return 0;
}

/* ================================================================ `-t N` as the reference executes it WITHOUT its races
 * (SURVEY.md 8f rank 4).  Contract = the reference binary with its N threads run one after the other in creation order
 * (oracle/seq_threads.c preloaded into oracle/_ref/extract_ref_z): per-thread byte chunks of the FASTQs found by
 * get_fq_start's forward scan (E:44-89), lines consumed while the byte cursor before the line is <= the chunk end
 * (E:1019-1026), sampling ordinals counted per chunk (E:1037), contig groups of split_ref (E:1280-1330) with peak ids
 * starting at j * (max_peak / N) (E:229-237), one sentinel line per thread in the interval file (E:515-548). */

/* get_fq_start (E:44-89), literally: the newline counter x and the flag persist across restarts of the outer loop.
 * Returns -1 where the reference's ifstream would run into EOF (its behaviour there is garbage-in): chunk starts must lie
 * more than 1000 bytes before the end of the file. */
long orc_get_fq_start(const unsigned char* p, long n, long start) {
    long pos = 0;
    int flag = 0, done = 0, x = 0;
    for (long i = start; i > 0; i--) {
        for (long j = i; j < i + 1000; j++) {
            if (j + 1 >= n) return -1;
            const char chr1 = (char)p[j], chr2 = (char)p[j + 1];
            if (chr1 == '\n' && chr2 == '+') { flag = 1; x = 0; }             /* third field */
            if (flag) {
                if (chr1 == '\n') x += 1;                                     /* forth field */
                if (chr1 == '\n' && chr2 == '@' && x == 3) { pos = j + 1; done = 1; break; }
            } else if (chr1 == '\n') x += 1;
            if (x == 3) break;
        }
        if (done) break;
    }
    return pos;
}

/* phase A over one thread chunk [start, end] of one file (read_fastq, E:981-1107) */
static long orc_count_chunk(const orc_buf* b, long start, long end, int k, int e, const short* cc, double ratio,
                            const float* rnd, uint8_t* table) {
    long pos = orc_get_fq_start(b->p, b->n, start);
    if (pos < 0) return -1;
    long cur = pos, add_size = pos, lines = 0, len, kept = 0;
    const unsigned char* s;
    uint32_t h[16];
    while (orc_next_line(b, &cur, &s, &len)) {
        if (add_size > end) break;                                            /* E:1022-1026 */
        add_size += len + 1;
        if (lines % 4 == 1 && orc_keep(lines / 4, rnd, ratio)) {               /* ordinal inside the chunk, E:1037 */
            if (len > ORC_MAX_READ) return -3;
            kept++;
            for (int q = 0; q + k <= len; q++) {
                if (!orc_hash_kmer(s + q, k, e, cc, h)) continue;
                for (int i = 0; i < e; i++) if (table[h[i]] < 3) table[h[i]]++;
            }
        }
        lines++;
    }
    return kept;
}

/* phase C over one thread chunk (slide_reads, E:313-506): fq2 is entered at the same byte offset and re-synchronised on the
 * read ID of fq1's first line (E:368-402) */
static long orc_vote_chunk(const orc_buf* b1, const orc_buf* b2, long start, long end, const orc_vote_job* j) {
    long pos = orc_get_fq_start(b1->p, b1->n, start);
    if (pos < 0 || pos > b2->n) return -1;
    long cur1 = pos, cur2 = pos, add_size = pos, lines = 0, len1, len2 = 0, kept = 0;
    const unsigned char *s1, *s2 = (const unsigned char*)"";
    orc_vote_state* S = (orc_vote_state*)malloc(sizeof(orc_vote_state));
    while (orc_next_line(b1, &cur1, &s1, &len1)) {
        /* std::getline on a finished stream: the first failing call clears the string when the file ended with a newline (the
         * sentry still succeeds), but leaves the last line in place when EOF was hit inside it (eofbit set: the sentry fails) */
        if (!orc_next_line(b2, &cur2, &s2, &len2) && b2->n > 0 && b2->p[b2->n - 1] == '\n') { s2 = (const unsigned char*)""; len2 = 0; }
        if (add_size > end) break;
        add_size += len1 + 1;
        if (lines == 0) {
            long i1 = orc_read_id_len(s1, len1), i2 = orc_read_id_len(s2, len2);
            if (i1 != i2 || memcmp(s1, s2, (size_t)i1)) {
                cur2 = pos - 1000000000L;
                if (cur2 < 1) cur2 = 1;
                for (;;) {
                    if (!orc_next_line(b2, &cur2, &s2, &len2)) { free(S); return -2; }   /* the reference spins 1e9 times, then reads garbage */
                    i2 = orc_read_id_len(s2, len2);
                    if (i1 == i2 && !memcmp(s1, s2, (size_t)i1)) break;
                }
            }
        }
        if (lines % 4 == 1 && orc_keep(lines / 4, j->rnd, j->ratio)) {
            if (len1 > ORC_MAX_READ || len2 > ORC_MAX_READ) { free(S); return -3; }
            kept++;
            S->n = 0; S->base_hits = 0;
            orc_seq q1 = {s1, (int)len1}, q2 = {s2, (int)len2};
            orc_vote_mate(S, &q1, j);
            orc_vote_mate(S, &q2, j);
            if (S->base_hits >= 6) orc_check_split(S, j->peak_filter);
        }
        lines++;
    }
    free(S);
    return kept;
}

/* split_ref (E:1280-1330): contig groups from genome.len.txt; cut[3*i] = start byte, [3*i+1] = end byte, [3*i+2] = first ref_index */
static int orc_split_ref(const char* index_path, const char* len_path, int k, int e, int threads, long* cut /*[300]*/) {
    memset(cut, 0, sizeof(long) * 300);
    FILE* f = fopen(len_path, "r");
    if (!f) return -1;
    long index_size = orc_file_size(index_path), each = index_size / threads + 1;
    long pos = 300 * 4, start_byte = pos, end_byte, start_ref_index = 1, count_ref_index = 0, slide;
    int cut_index = 0, ref_len, ref_index;
    char name[4096];
    while (fscanf(f, "%4095s %d %d %ld", name, &ref_index, &ref_len, &slide) == 4) {
        count_ref_index += 1;
        long add = 4L * ((long)(ref_len - k + 1) * e + 1);
        if (pos - start_byte > each) {
            end_byte = pos + add;
            cut[3 * cut_index] = start_byte; cut[3 * cut_index + 1] = end_byte; cut[3 * cut_index + 2] = start_ref_index;
            cut_index += 1;
            start_byte = end_byte;
            start_ref_index = count_ref_index + 1;
        }
        pos += add;
        if (pos >= index_size) break;
    }
    if (start_byte != index_size) {
        cut[3 * cut_index] = start_byte; cut[3 * cut_index + 1] = index_size; cut[3 * cut_index + 2] = start_ref_index;
        cut[3 * cut_index + 3] = 0;
    }
    fclose(f);
    return 0;
}

/* The 12-argument contract with `threads` = the reference's -t (> 1), executed without its races.  Unsupported inputs (a chunk
 * start within 1000 bytes of EOF, fq2 without the read ID of a chunk's first record, a thread's peaks overflowing its id
 * range) return an error instead of the reference's undefined behaviour. */
int orc_run_threads(const char* fq1, const char* fq2, const char* fasta, const char* interval_path,
                    double hit_ratio_d, double match_ratio_d, int threads, int k, long max_peak, int e,
                    unsigned seed, double sample, orc_report* rep) {
    if (threads < 1 || threads > 99) return -9;
    size_t slots = (size_t)1 << k;
    uint8_t* table = (uint8_t*)calloc(slots, 1);
    short cc[ORC_CODER_SLOTS], cc_file[ORC_CODER_SLOTS];
    if (!table) return -1;
    srand(seed);
    double ratio = orc_sam_ratio(fq1, sample);
    char index_path[4096], len_path[4096];
    snprintf(index_path, sizeof index_path, "%s.k%d.h%d.index.dat", fasta, k, e);
    snprintf(len_path, sizeof len_path, "%s.genome.len.txt", fasta);
    FILE* probe = fopen(index_path, "rb");
    if (probe) fclose(probe);
    else {
        orc_random_coder(k, e, cc);
        if (orc_index_build(fasta, index_path, len_path, k, e, cc) < 0) return -2;
    }
    if (orc_index_header(index_path, cc_file)) return -3;
    float* rnd = ratio >= 100.0 ? NULL : orc_sampling_array(ORC_MAX_RANDOM);
    orc_buf b1 = orc_slurp(fq1), b2 = orc_slurp(fq2);
    if (b1.n < 0 || b2.n < 0) return -4;
    const long size = b1.n, each_size = size / threads;                                  /* E:1419-1420 */
    long c1 = 0, c2 = 0;
    for (int pass = 0; pass < 2; pass++)                                                  /* fq1 by all threads, then fq2 (E:1426-1448) */
        for (int i = 0; i < threads; i++) {
            long start = i * each_size, end = i == threads - 1 ? size : (i + 1) * each_size;
            long got = orc_count_chunk(pass ? &b2 : &b1, start, end, k, e, cc_file, ratio, rnd, table);
            if (got < 0) return -4;
            if (pass) c2 += got; else c1 += got;
        }
    int32_t* loci = (int32_t*)calloc((size_t)max_peak * 2 + 2, 4);
    uint8_t* peak_filter = (uint8_t*)calloc((size_t)max_peak + 1, 1);
    uint32_t* peak_kmer = (uint32_t*)calloc(slots, 4);
    if (!loci || !peak_filter || !peak_kmer) return -1;
    /* phase B: one thread per contig group (E:1468-1489), ids from j * (max_peak / threads) (E:229-237) */
    long cut[300], peak_index[100], each_peaks = max_peak / threads, total_peaks = 0;
    for (int j = 0; j < threads; j++) peak_index[j] = each_peaks * j;
    if (orc_split_ref(index_path, len_path, k, e, threads, cut)) return -5;
    FILE* f = fopen(index_path, "rb");
    if (!f) return -5;
    for (int i = 0; i < threads; i++) {
        long start = cut[3 * i], end = cut[3 * i + 1];
        if (start == 0) break;
        int ref_index = (int)cut[3 * i + 2];
        orc_peaks P = {loci, peak_kmer, peak_index[i], max_peak, 0, 0, 0};
        fseek(f, start, SEEK_SET);
        long start_point = start;
        uint32_t ref_len_u;
        while (fread(&ref_len_u, 4, 1, f) == 1) {                                         /* read_index, E:921-972 */
            int ref_len = (int)ref_len_u;
            long nk = (long)ref_len - k + 1;
            uint32_t* hidx = (uint32_t*)calloc((size_t)ref_len * e, 4);
            uint8_t* hit = (uint8_t*)calloc((size_t)ref_len * e, 1);
            if (fread(hidx, 4, (size_t)nk * e, f) != (size_t)nk * e) { fclose(f); return -5; }
            for (long q = 0; q < nk * e; q++) hit[q] = hidx[q] ? table[hidx[q]] : 0;
            orc_slide_window(&P, hit, hidx, ref_len, ref_index, k, e, (float)hit_ratio_d, (float)match_ratio_d, NULL);
            free(hidx); free(hit);
            start_point += 4 + nk * e * 4;
            ref_index += 1;
            if (start_point >= end) break;
        }
        if (P.too_many || P.n_peaks - each_peaks * i > each_peaks) { fclose(f); return -7; }   /* the reference runs into the next thread's ids */
        total_peaks += P.n_peaks - peak_index[i];
        peak_index[i] = P.n_peaks;
    }
    fclose(f);
    /* phase C */
    orc_vote_job jb = {NULL, NULL, 0, 0, k, e, cc_file, rnd, ratio, peak_kmer, loci, peak_filter, 0};
    long voted = 0;
    for (int i = 0; i < threads; i++) {
        long start = i * each_size, end = i == threads - 1 ? size : (i + 1) * each_size;
        long got = orc_vote_chunk(&b1, &b2, start, end, &jb);
        if (got < 0) return -6;
        voted += got;
    }
    /* phase D (E:515-548): every thread's id range with its own sentinel state */
    FILE* out = fopen(interval_path, "w");
    if (!out) return -8;
    long nf = 0;
    for (int j = 0; j < threads; j++) {
        int start = 1, end = 1, chr = 1;
        for (long i = each_peaks * j; i < peak_index[j]; i++) {
            if (peak_filter[i] < 1) continue;
            nf++;
            int c = loci[2 * i], pos = loci[2 * i + 1];
            if (chr == c && pos - 500 - end < 500) end = pos + 500;
            else { fprintf(out, "%d\t%d\t%d\n", chr, start, end); chr = c; start = pos - 500; end = pos + 500; }
        }
        fprintf(out, "%d\t%d\t%d\n", chr, start, end);
    }
    fclose(out);
    if (rep) {
        memset(rep, 0, sizeof *rep);
        rep->pairs_counted = c1; rep->pairs_voted = voted; rep->n_peaks = total_peaks; rep->n_filtered = nf;
        rep->t_count = (double)c2;   /* mate-2 reads counted, for the partition tests */
    }
    free(table); free(loci); free(peak_filter); free(peak_kmer); free(rnd); free(b1.p); free(b2.p);
    return 0;
}

/* ================================================================ count_diff_kmer.cpp ("C"), the stand-alone phase-A tool
 * Citations C:n = /root/reference/src/count_diff_kmer.cpp:n.  The tool seeds its coder and its sampling from time(0)
 * (C:87-89, 223-225) and runs 10 racing threads, so its contract here is the binary run with time() fixed
 * (oracle/fixed_time.c) and its threads in creation order (oracle/seq_threads.c).  What differs from extract_ref's phase A:
 *   - the coder is a `bool` array, so the "invalid" code 5 collapses to 1 (C:155-160): a non-ACGT base is never rejected, it
 *     codes 1 in every projection, on the forward strand and -- its complement being the NUL byte -- on the reverse strand too;
 *   - one rand() per k-mer position for the coder (C:226-232; extract_ref draws e/3+1);
 *   - 10 thread chunks of fq1's size (C:308, 331-343); a chunk is entered at the nearest '@' at or BEFORE its start (C:61-69;
 *     thread 0's `pos` is uninitialised there and behaves as 0 in the -O2 binary), tokens are read with `>>` (C:91), the byte
 *     budget counts token lengths only, starts at `start` and stops at `add_size >= end` (C:70, 93-96): chunks overrun into
 *     their successors and those reads are counted twice;
 *   - every read of a chunk is cut to the length of the chunk's first read (C:103-105);
 *   - sampling: srand(time) at the head of every chunk, then r = rand() % 100 < ratio per sequence token (C:87-89, 107-109).
 * hist[v] = slots holding v.  Returns <0 on inputs where the binary reads out of bounds (reads of unequal length or longer
 * than 150, a chunk starting at or behind the end of a file). */
static int orc_cdk_bit(int map, int ch) { int m = orc_proj(map, ch); return m == 5 ? 1 : m; }

int orc_count_diff_kmer(const char* fq1, const char* fq2, int k, int ratio, unsigned time_seed, uint64_t hist[4]) {
    static const short permu[18] = {0, 1, 2, 0, 2, 1, 1, 2, 0, 1, 0, 2, 2, 0, 1, 2, 1, 0};
    short cc[100];
    if (k < 1 || k > 32) return -9;
    srand(time_seed);                                           /* C:223-225 */
    for (int j = 0; j < k; j++) {
        int r = rand() % 6;
        for (int i = 0; i < 3; i++) cc[j * 3 + i] = permu[r * 3 + i];
    }
    size_t slots = (size_t)1 << k;
    uint8_t* table = (uint8_t*)calloc(slots, 1);
    orc_buf b[2] = {orc_slurp(fq1), orc_slurp(fq2)};
    if (!table || b[0].n < 0 || b[1].n < 0) return -1;
    const long size = b[0].n, each = size / 10;
    int rc = 0;
    for (int f = 0; f < 2 && !rc; f++)
        for (int t = 0; t < 10 && !rc; t++) {
            const long start = t * each, end = t == 9 ? size : (t + 1) * each;
            if (start >= b[f].n && start > 0) { rc = -4; break; }
            long pos = 0;                                       /* C:60-69 */
            for (long i = start; i > 0; i--) if (b[f].p[i] == '@') { pos = i; break; }
            long cur = pos, add_size = start, tok = 0;
            int read_len = 0;
            srand(time_seed);                                   /* C:87-89 */
            for (;;) {
                while (cur < b[f].n && (b[f].p[cur] == ' ' || (b[f].p[cur] >= 9 && b[f].p[cur] <= 13))) cur++;   /* operator>> */
                if (cur >= b[f].n) break;
                long t0 = cur;
                while (cur < b[f].n && !(b[f].p[cur] == ' ' || (b[f].p[cur] >= 9 && b[f].p[cur] <= 13))) cur++;
                const unsigned char* s = b[f].p + t0;
                const long len = cur - t0;
                if (add_size >= end) break;                     /* C:93-95 */
                add_size += len;
                if (tok % 4 == 1) {
                    if (tok == 1) read_len = (int)len;          /* C:103-105 */
                    if (len != read_len || len > 150) { rc = -3; break; }
                    int r = rand() % 100;
                    if (r < ratio)
                        for (int j = 0; j + k <= read_len; j++)
                            for (int i = 0; i < 3; i++) {
                                uint32_t fwd = 0, rcw = 0;
                                for (int z = 0; z < k; z++) {
                                    fwd += (uint32_t)orc_cdk_bit(cc[z * 3 + i], s[j + z]) << (k - 1 - z);
                                    rcw += (uint32_t)orc_cdk_bit(cc[(k - 1 - z) * 3 + i], orc_comp(s[j + z])) << z;
                                }
                                uint32_t h = fwd > rcw ? rcw : fwd;
                                if (table[h] < 3) table[h]++;
                            }
                }
                tok++;
            }
        }
    if (!rc) {
        hist[0] = hist[1] = hist[2] = hist[3] = 0;
        for (size_t q = 0; q < slots; q++) hist[table[q]]++;
    }
    free(table); free(b[0].p); free(b[1].p);
    return rc;
}

#ifdef ORC_MAIN
int main(int argc, char** argv) {
    if (argc < 13) {
        fprintf(stderr, "usage: %s fq1 fq2 ref.fa interval_out hit_ratio match_ratio threads k max_peak e seed sample\n", argv[0]);
        return 2;
    }
    orc_report rep;
    /* ORC_EMULATE_THREADS=1: argv[7] is the reference's -t, executed without its races (orc_run_threads); otherwise it only
     * shards the reads over worker threads and the result is the -t 1 one */
    int rc = (getenv("ORC_EMULATE_THREADS") ? orc_run_threads : orc_run)(argv[1], argv[2], argv[3], argv[4], atof(argv[5]), atof(argv[6]),
                     (int)atof(argv[7]), (int)atof(argv[8]), (long)atof(argv[9]), (int)atof(argv[10]), (unsigned)atof(argv[11]),
                     atof(argv[12]), &rep);
    if (rc) { fprintf(stderr, "oracle failed: %d\n", rc); return 1; }
    printf("{\"index_s\": %.3f, \"count_s\": %.3f, \"scan_s\": %.3f, \"vote_s\": %.3f, \"total_s\": %.3f, "
           "\"pairs\": %ld, \"raw_peaks\": %ld, \"filtered_peaks\": %ld}\n",
           rep.t_index, rep.t_count, rep.t_scan, rep.t_vote, rep.t_total, rep.pairs_counted, rep.n_peaks, rep.n_filtered);
    return 0;
}
#endif
