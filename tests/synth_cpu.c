/* synth_cpu.c -- TEST INFRASTRUCTURE: a host twin of the device-side workload generator (localhgt_amd/csrc/k_synth.hip)
 * that writes the very files tools/benchlib/files.py: synth_files() writes on a GPU box -- ref.fa (one line per contig),
 * s.1.fq / s.2.fq (`@r<9 digits>/<mate>`, 150 bases, `+`, 150 x `I`) -- so that the REAL reference binary can be run on
 * BASELINE configs[1] at full size in a container without a GPU (tests/golden/configs1_full/make_golden.sh) and the product can
 * be compared with its output on the GPU box, where the same bytes come from the device generator (sha256 of the three files
 * is part of the golden).  Every base is a pure function of (seed, contig, position): splitmix64 finaliser, as there.
 *   gcc -O2 -fopenmp -o synth_cpu synth_cpu.c
 *   synth_cpu <dir> <n_contigs> <contig_len> <n_pairs> [ref_seed=1] [reads_seed=2] [n_permille=20] [snp_permille=0] [sample_contigs=0]
 * Nothing in the product path uses this file. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint32_t ref_code(uint64_t seed, uint32_t contig, uint64_t pos) {
    return (uint32_t)(mix64(seed ^ mix64(((uint64_t)contig << 40) ^ pos)) >> 62);
}

typedef struct {
    uint64_t ref_seed, reads_seed, contig_len;
    uint32_t n_contigs, transfer_len, read_len, frag_min, frag_max, n_permille, snp_permille, n_sample;
} Spec;

static inline void transfer_sites(const Spec* s, uint32_t i, uint64_t* r0, uint64_t* d0) {
    uint64_t h = mix64(s->ref_seed * 0x51ED2701ull + i);
    uint64_t span = s->contig_len - 3ull * s->transfer_len;
    *r0 = s->transfer_len + (h % span);
    *d0 = s->transfer_len + (mix64(h) % span);
}
static inline uint32_t sample_code_nosnp(const Spec* s, uint32_t g, uint64_t x) {
    uint64_t r0, d0;
    transfer_sites(s, g >> 1, &r0, &d0);
    uint32_t rec = g & ~1u, don = g | 1u;
    if (!(g & 1)) {
        if (x < r0) return ref_code(s->ref_seed, rec, x);
        if (x < r0 + s->transfer_len) return ref_code(s->ref_seed, don, d0 + (x - r0));
        return ref_code(s->ref_seed, rec, x - s->transfer_len);
    }
    return ref_code(s->ref_seed, don, x < d0 ? x : x + s->transfer_len);
}
static inline uint32_t sample_code(const Spec* s, uint32_t g, uint64_t x) {
    uint32_t c = sample_code_nosnp(s, g, x);
    if (s->snp_permille) {
        const uint64_t h = mix64(s->ref_seed * 0x2545F491ull ^ mix64(((uint64_t)g << 40) ^ x));
        if (h % 1000 < s->snp_permille) c = (c + 1u + (uint32_t)((h >> 32) % 3)) & 3u;
    }
    return c;
}

static void pair_bases(const Spec* s, uint64_t p, uint8_t* o1, uint8_t* o2) {
    const uint64_t L = s->read_len;
    uint64_t h = mix64(s->reads_seed ^ mix64(p));
    uint32_t g = (uint32_t)(h % s->n_sample);
    uint64_t h2 = mix64(h);
    uint64_t glen = (g & 1) ? s->contig_len - s->transfer_len : s->contig_len + s->transfer_len;
    uint64_t flen = s->frag_min + h2 % (s->frag_max - s->frag_min + 1);
    uint64_t h3 = mix64(h2);
    uint64_t start = h3 % (glen - flen + 1);
    int flip = (mix64(h3) >> 63) != 0;
    uint64_t h4 = mix64(h3 ^ 0xA5A5A5A5ull);
    for (uint32_t b = 0; b < L; b++) {
        uint32_t left = sample_code(s, g, start + b);
        uint32_t right = 3u - sample_code(s, g, start + flen - 1 - b);
        uint8_t c1 = "ACGT"[flip ? right : left], c2 = "ACGT"[flip ? left : right];
        if (h4 % 1000 < s->n_permille) {
            uint32_t col = (uint32_t)((h4 >> 20) % L);
            if (col == b) { if ((h4 >> 40) & 1) c1 = 'N'; else c2 = 'N'; }
        }
        o1[b] = c1;
        o2[b] = c2;
    }
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: synth_cpu dir n_contigs contig_len n_pairs [ref_seed reads_seed n_permille snp_permille sample_contigs]\n"); return 2; }
    const char* dir = argv[1];
    Spec s;
    s.n_contigs = (uint32_t)atol(argv[2]);
    s.contig_len = (uint64_t)atol(argv[3]);
    const uint64_t n_pairs = (uint64_t)atol(argv[4]);
    s.ref_seed = argc > 5 ? (uint64_t)atol(argv[5]) : 1;
    s.reads_seed = argc > 6 ? (uint64_t)atol(argv[6]) : 2;
    s.n_permille = argc > 7 ? (uint32_t)atol(argv[7]) : 20;
    s.snp_permille = argc > 8 ? (uint32_t)atol(argv[8]) : 0;
    const long sc = argc > 9 ? atol(argv[9]) : 0;
    s.n_sample = sc > 0 && sc <= (long)s.n_contigs ? (uint32_t)sc & ~1u : (s.n_contigs / 2) & ~1u;
    s.transfer_len = 3000; s.read_len = 150; s.frag_min = 300; s.frag_max = 500;
    if (s.n_contigs < 4 || s.contig_len < 16000) { fprintf(stderr, "need >= 4 contigs of >= 16 kb\n"); return 2; }
    char path[4096];
    /* ref.fa: >g<c+1>, the contig on one line */
    snprintf(path, sizeof path, "%s/ref.fa", dir);
    FILE* f = fopen(path, "wb");
    if (!f) { perror(path); return 1; }
    uint8_t* buf = (uint8_t*)malloc(s.contig_len + 1);
    for (uint32_t c = 0; c < s.n_contigs; c++) {
        fprintf(f, ">g%u\n", c + 1);
#pragma omp parallel for schedule(static)
        for (long x = 0; x < (long)s.contig_len; x++) buf[x] = "ACGT"[ref_code(s.ref_seed, c, (uint64_t)x)];
        buf[s.contig_len] = '\n';
        if (fwrite(buf, 1, s.contig_len + 1, f) != s.contig_len + 1) { perror("write"); return 1; }
    }
    fclose(f);
    free(buf);
    /* FASTQs: records of 13 + 1 + L + 1 + 2 + L + 1 bytes (header `@r<9 digits>/<mate>`) */
    const uint64_t L = s.read_len, REC = 13 + 1 + L + 1 + 2 + L + 1, SL = 1 << 18;
    uint8_t* r1 = (uint8_t*)malloc(SL * REC);
    uint8_t* r2 = (uint8_t*)malloc(SL * REC);
    snprintf(path, sizeof path, "%s/s.1.fq", dir);
    FILE* f1 = fopen(path, "wb");
    snprintf(path, sizeof path, "%s/s.2.fq", dir);
    FILE* f2 = fopen(path, "wb");
    if (!f1 || !f2) { perror("fastq"); return 1; }
    for (uint64_t p0 = 0; p0 < n_pairs; p0 += SL) {
        const uint64_t n = n_pairs - p0 < SL ? n_pairs - p0 : SL;
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)n; i++) {
            uint8_t *a = r1 + (uint64_t)i * REC, *b = r2 + (uint64_t)i * REC;
            char id[16];
            snprintf(id, sizeof id, "@r%09llu/", (unsigned long long)(p0 + (uint64_t)i));   /* numpy: arange(n).astype("U9") zero-filled to 9 */
            memcpy(a, id, 12); a[12] = '1'; a[13] = '\n';
            memcpy(b, id, 12); b[12] = '2'; b[13] = '\n';
            pair_bases(&s, p0 + (uint64_t)i, a + 14, b + 14);
            a[14 + L] = '\n'; a[15 + L] = '+'; a[16 + L] = '\n'; memset(a + 17 + L, 'I', L); a[17 + 2 * L] = '\n';
            b[14 + L] = '\n'; b[15 + L] = '+'; b[16 + L] = '\n'; memset(b + 17 + L, 'I', L); b[17 + 2 * L] = '\n';
        }
        if (fwrite(r1, REC, n, f1) != n || fwrite(r2, REC, n, f2) != n) { perror("write"); return 1; }
    }
    fclose(f1);
    fclose(f2);
    return 0;
}
