// k_synth.hip -- synthetic workload generated on the device from seeds: iid
// uniform ACGT contigs, and 150 bp pairs drawn from a sample made of the first half of the
// contigs taken in (recipient, donor) pairs with one 3 kb cut-and-paste transfer each.
// Every base is a pure function of (seed, contig, position), so reads need no stored reference
// and any rank can generate any shard.  Bench/test support only: not part of the reference.
#include "lhgt_hash.hpp"

namespace lhgt {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint32_t ref_code(uint64_t seed, uint32_t contig, uint64_t pos) {
    return (uint32_t)(mix64(seed ^ mix64(((uint64_t)contig << 40) ^ pos)) >> 62);
}

struct SynthSpec {
    uint64_t ref_seed, reads_seed;
    uint32_t n_contigs;
    uint64_t contig_len;
    uint32_t transfer_len;  // 3000
    uint32_t read_len;      // 150
    uint32_t frag_min, frag_max;  // 300..500
    uint32_t n_permille;    // reads carrying one N, per thousand (20)
    uint32_t snp_permille;  // positions of a sample genome that differ from the reference, per thousand (0; 10 = the "snp0.01" of the
                            // reference's test data, test/run_BKP_detection.sh)
    uint32_t n_sample;      // contigs the sample is made of (even; default: half of the reference)
    uint32_t long_permille, long_len;   // pairs per thousand whose two reads have long_len bases instead of read_len (lhgt_synth_read_mix)
};
// the read length of pair p (both mates)
__host__ __device__ __forceinline__ uint32_t pair_read_len(const SynthSpec& s, uint64_t p) {
    return s.long_permille && mix64(s.reads_seed * 0x6A09E667ull ^ mix64(p ^ 0x5bd1e995ull)) % 1000 < s.long_permille ? s.long_len : s.read_len;
}

// donor cut position / recipient insert position of sample pair i
__host__ __device__ __forceinline__ void transfer_sites(const SynthSpec& s, uint32_t i, uint64_t* r0, uint64_t* d0) {
    uint64_t h = mix64(s.ref_seed * 0x51ED2701ull + i);
    uint64_t span = s.contig_len - 3ull * s.transfer_len;
    *r0 = s.transfer_len + (h % span);
    *d0 = s.transfer_len + (mix64(h) % span);
}

// base x of sample genome g (even g = recipient with the insert, odd g = donor with the deletion)
__device__ __forceinline__ uint32_t sample_code_nosnp(const SynthSpec& s, uint32_t g, uint64_t x);
// a SNP belongs to the sample genome, not to a read: every read over the position shows it
__device__ __forceinline__ uint32_t sample_code(const SynthSpec& s, uint32_t g, uint64_t x) {
    uint32_t c = sample_code_nosnp(s, g, x);
    if (s.snp_permille) {
        const uint64_t h = mix64(s.ref_seed * 0x2545F491ull ^ mix64(((uint64_t)g << 40) ^ x));
        if (h % 1000 < s.snp_permille) c = (c + 1u + (uint32_t)((h >> 32) % 3)) & 3u;
    }
    return c;
}
__device__ __forceinline__ uint32_t sample_code_nosnp(const SynthSpec& s, uint32_t g, uint64_t x) {
    uint64_t r0, d0;
    transfer_sites(s, g >> 1, &r0, &d0);
    uint32_t rec = g & ~1u, don = g | 1u;
    if (!(g & 1)) {
        if (x < r0) return ref_code(s.ref_seed, rec, x);
        if (x < r0 + s.transfer_len) return ref_code(s.ref_seed, don, d0 + (x - r0));
        return ref_code(s.ref_seed, rec, x - s.transfer_len);
    }
    return ref_code(s.ref_seed, don, x < d0 ? x : x + s.transfer_len);
}

// bases [flat0, flat0 + n) of the reference laid out as one stream (contig c = stream positions [c*contig_len, (c+1)*contig_len))
__global__ void __launch_bounds__(256) synth_flat_ascii(SynthSpec s, uint64_t flat0, uint64_t n, uint8_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t x = flat0 + i;
    out[i] = "ACGT"[ref_code(s.ref_seed, (uint32_t)(x / s.contig_len), x % s.contig_len)];
}

// thread = one base of one mate; out1/out2 are [n][stride] ASCII, stride = the longer of the two read lengths in use
__global__ void __launch_bounds__(256) synth_pairs_ascii(SynthSpec s, uint64_t first_pair, uint64_t n_pairs,
                                                         uint8_t* __restrict__ out1, uint8_t* __restrict__ out2) {
    uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = s.long_permille && s.long_len > s.read_len ? s.long_len : s.read_len;
    if (t >= n_pairs * stride) return;
    uint64_t pi = t / stride;
    uint32_t b = (uint32_t)(t % stride);
    uint64_t p = first_pair + pi;
    const uint64_t L = pair_read_len(s, p);
    if (b >= L) { out1[pi * stride + b] = 'A'; out2[pi * stride + b] = 'A'; return; }   // behind a shorter read's end: never read
    uint64_t h = mix64(s.reads_seed ^ mix64(p));
    uint32_t g = (uint32_t)(h % s.n_sample);
    uint64_t h2 = mix64(h);
    uint64_t glen = (g & 1) ? s.contig_len - s.transfer_len : s.contig_len + s.transfer_len;
    uint64_t flen = s.frag_min + h2 % (s.frag_max - s.frag_min + 1);
    uint64_t h3 = mix64(h2);
    uint64_t start = h3 % (glen - flen + 1);
    bool flip = (mix64(h3) >> 63) != 0;
    // left mate reads the fragment forward, right mate is the reverse complement of its tail
    uint32_t left = sample_code(s, g, start + b);
    uint32_t right = 3u - sample_code(s, g, start + flen - 1 - b);
    uint8_t c1 = "ACGT"[flip ? right : left], c2 = "ACGT"[flip ? left : right];
    uint64_t h4 = mix64(h3 ^ 0xA5A5A5A5ull);
    if (h4 % 1000 < s.n_permille) {
        uint32_t col = (uint32_t)((h4 >> 20) % L);
        if (col == b) { if ((h4 >> 40) & 1) c1 = 'N'; else c2 = 'N'; }
    }
    out1[pi * stride + b] = c1;
    out2[pi * stride + b] = c2;
}

static SynthSpec make_spec(const lhgt_ctx* ctx, uint64_t ref_seed, uint64_t reads_seed, long n_contigs, long contig_len, int read_len) {
    SynthSpec s;
    s.snp_permille = (uint32_t)ctx->synth_snp_permille;
    s.n_sample = ctx->synth_sample_contigs > 0 && ctx->synth_sample_contigs <= n_contigs ? (uint32_t)ctx->synth_sample_contigs & ~1u : ((uint32_t)n_contigs / 2) & ~1u;
    s.ref_seed = ref_seed; s.reads_seed = reads_seed;
    s.n_contigs = (uint32_t)n_contigs; s.contig_len = (uint64_t)contig_len;
    s.transfer_len = 3000; s.read_len = (uint32_t)read_len; s.frag_min = 300; s.frag_max = 500; s.n_permille = (uint32_t)ctx->synth_n_permille;
    s.long_permille = (uint32_t)ctx->synth_long_permille; s.long_len = (uint32_t)ctx->synth_long_len;
    return s;
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

// Build the resident index of a synthetic reference of n_contigs x contig_len bases.
// host_ascii (optional, n_contigs*contig_len bytes) receives the bases for writing a FASTA.
int lhgt_synth_reference(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, uint8_t* host_ascii) {
    return lhgt_synth_reference_shard(ctx, ref_seed, n_contigs, contig_len, 0, 1, host_ascii);
}

// The same base stream cut into contigs at cuts[0] = 0 < cuts[1] < ... < cuts[n_cuts-1] = n_contigs*contig_len: a reference
// with a ragged length distribution (a real catalogue has hundreds of thousands of contigs from a few hundred bases to
// megabases) over which the SAME synthetic reads can be scanned -- reads that straddle a cut simply match on either side of it.
// Pieces of length <= k are not indexed (E:772).  Only pieces [piece0, piece1) become resident; contig numbers are global.
static int synth_reference_pieces(lhgt_ctx* ctx, const SynthSpec& s, const uint64_t* cuts, long n_cuts, long piece0, long piece1,
                                  uint8_t* host_ascii) {
    const int k = ctx->k, e = ctx->e;
    std::vector<uint32_t> lens, lens_all;
    uint32_t first_ref_index = 1;
    for (long p = 0; p + 1 < n_cuts; p++) {
        const uint64_t len = cuts[p + 1] - cuts[p];
        if (len >= (1ull << 32)) LHGT_FAIL(LHGT_E_ARG, "contig of %llu bases", (unsigned long long)len);
        if ((long)len <= k) continue;
        lens_all.push_back((uint32_t)len);
        if (p < piece0) first_ref_index++;
        else if (p < piece1) lens.push_back((uint32_t)len);
    }
    LHGT_TRY(index_layout(ctx, lens, first_ref_index));
    if (lens.size() != lens_all.size()) ctx->all_lens = lens_all;
    LHGT_TRY(write_index_lens(ctx));
    // spans of whole pieces, at most ~256 Mbase each (one huge piece is its own span)
    const uint64_t SPAN = 256ull << 20;
    long ci = 0;   // next resident contig
    for (long p = piece0; p < piece1;) {
        long q = p;
        while (q < piece1 && (q == p || cuts[q + 1] - cuts[p] <= SPAN)) q++;
        const uint64_t span0 = cuts[p], span_len = cuts[q] - cuts[p];
        std::vector<uint64_t> coff((size_t)(q - p) + 1);
        std::vector<long> contig_of((size_t)(q - p));
        for (long r = p; r <= q; r++) coff[r - p] = cuts[r] - span0;
        for (long r = p; r < q; r++) {
            const uint64_t len = cuts[r + 1] - cuts[r];
            contig_of[r - p] = (long)len <= k ? -1L : ci++;
        }
        LHGT_TRY(ws_reserve(ctx, (size_t)span_len + 32, 0));
        hipLaunchKernelGGL(synth_flat_ascii, dim3((unsigned)((span_len + 255) / 256)), dim3(256), 0, ctx->stream, s, span0, span_len, ctx->d_ws_ascii);
        LHGT_HIP(hipGetLastError());
        if (host_ascii) LHGT_HIP(hipMemcpyAsync(host_ascii + (span0 - cuts[piece0]), ctx->d_ws_ascii, (size_t)span_len, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_TRY(install_span_dev_ascii(ctx, ctx->d_ws_ascii, (long)span_len, coff.data(), contig_of.data(), q - p));
        p = q;
    }
    (void)e;
    return LHGT_OK;
}

// Same reference, but only contigs [rank*n/world, (rank+1)*n/world) become resident (reference-sharded phase B).
int lhgt_synth_reference_shard(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, int shard_rank, int shard_world,
                               uint8_t* host_ascii) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (shard_world < 1 || shard_rank < 0 || shard_rank >= shard_world) LHGT_FAIL(LHGT_E_ARG, "bad shard spec %d/%d", shard_rank, shard_world);
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder set");
    if (n_contigs < 4 || contig_len < 16000 || contig_len >= (1L << 32) - 4096) LHGT_FAIL(LHGT_E_ARG, "need >= 4 contigs of 16 kb .. 4 Gb");
    SynthSpec s = make_spec(ctx, ref_seed, 0, n_contigs, contig_len, 150);
    std::vector<uint64_t> cuts((size_t)n_contigs + 1);
    for (long c = 0; c <= n_contigs; c++) cuts[c] = (uint64_t)c * (uint64_t)contig_len;
    const long c0 = n_contigs * shard_rank / shard_world, c1 = n_contigs * (shard_rank + 1) / shard_world;
    return synth_reference_pieces(ctx, s, cuts.data(), n_contigs + 1, c0, c1, host_ascii);
}

int lhgt_synth_reference_cuts(lhgt_ctx* ctx, uint64_t ref_seed, long n_contigs, long contig_len, const uint64_t* cuts, long n_cuts,
                              uint8_t* host_ascii) {
    if (!ctx || !cuts) LHGT_FAIL(LHGT_E_ARG, "null argument");
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder set");
    if (n_contigs < 4 || contig_len < 16000 || contig_len >= (1L << 32) - 4096) LHGT_FAIL(LHGT_E_ARG, "need >= 4 contigs of 16 kb .. 4 Gb");
    if (n_cuts < 2 || cuts[0] != 0 || cuts[n_cuts - 1] != (uint64_t)n_contigs * (uint64_t)contig_len) LHGT_FAIL(LHGT_E_ARG, "cuts must run from 0 to n_contigs*contig_len");
    for (long i = 1; i < n_cuts; i++) if (cuts[i] <= cuts[i - 1]) LHGT_FAIL(LHGT_E_ARG, "cuts must ascend strictly");
    SynthSpec s = make_spec(ctx, ref_seed, 0, n_contigs, contig_len, 150);
    return synth_reference_pieces(ctx, s, cuts, n_cuts, 0, n_cuts - 1, host_ascii);
}

// Knobs of the synthetic sample (defaults: no SNPs, 20 reads per thousand carry one N, the sample is half of the contigs).
int lhgt_synth_options(lhgt_ctx* ctx, int snp_permille, int n_permille, long sample_contigs) {
    if (!ctx || snp_permille < 0 || snp_permille > 1000 || n_permille < 0 || n_permille > 1000 || sample_contigs < 0 || sample_contigs == 1)
        LHGT_FAIL(LHGT_E_ARG, "bad synthetic options");
    ctx->synth_snp_permille = snp_permille;
    ctx->synth_n_permille = n_permille;
    ctx->synth_sample_contigs = sample_contigs;
    return LHGT_OK;
}

// A share of the synthetic pairs with longer reads (both mates long_len bases; 0 = all read_len): the batches a real run sees when
// a sample holds reads of more than one length.
int lhgt_synth_read_mix(lhgt_ctx* ctx, int long_permille, int long_len) {
    if (!ctx || long_permille < 0 || long_permille > 1000 || long_len < 0 || long_len > 300) LHGT_FAIL(LHGT_E_ARG, "bad read mix");
    ctx->synth_long_permille = long_permille;
    ctx->synth_long_len = long_len;
    return LHGT_OK;
}

// Append pairs [first_pair, first_pair + n_pairs) of the synthetic sample to the resident store.
// host_seq1/2 (optional, n_pairs*read_len bytes each) receive the bases for writing FASTQs.
int lhgt_synth_pairs(lhgt_ctx* ctx, uint64_t ref_seed, uint64_t reads_seed, long n_contigs, long contig_len,
                     long first_pair, long n_pairs, int read_len, uint8_t* host_seq1, uint8_t* host_seq2) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    LHGT_DEVICE_ENTRY(ctx);
    if (n_contigs < 4 || contig_len < 16000 || read_len < 1 || read_len > 300 || n_pairs < 0) LHGT_FAIL(LHGT_E_ARG, "bad synthetic spec");
    SynthSpec s = make_spec(ctx, ref_seed, reads_seed, n_contigs, contig_len, read_len);
    const int stride = s.long_permille && (int)s.long_len > read_len ? (int)s.long_len : read_len;
    if ((host_seq1 || host_seq2) && stride != read_len) LHGT_FAIL(LHGT_E_ARG, "host copies of mixed-length synthetic reads are not laid out");
    long CH = 16L << 20;  // pairs per resident batch (one scan launch each)
    while (CH * stride >= (1L << 32)) CH >>= 1;   // synth_pairs_ascii: one work-item per base, a launch holds 2^32 - 1
    for (long o = 0; o < n_pairs; o += CH) {
        long n = n_pairs - o < CH ? n_pairs - o : CH;
        size_t bytes = (size_t)n * stride;
        LHGT_TRY(ws_reserve(ctx, 2 * bytes + 32, 0));
        uint8_t *d1 = ctx->d_ws_ascii, *d2 = ctx->d_ws_ascii + bytes;
        hipLaunchKernelGGL(synth_pairs_ascii, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ctx->stream, s,
                           (uint64_t)(first_pair + o), (uint64_t)n, d1, d2);
        LHGT_HIP(hipGetLastError());
        if (host_seq1) LHGT_HIP(hipMemcpyAsync(host_seq1 + (size_t)o * read_len, d1, bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (host_seq2) LHGT_HIP(hipMemcpyAsync(host_seq2 + (size_t)o * read_len, d2, bytes, hipMemcpyDeviceToHost, ctx->stream));
        std::vector<uint64_t> start((size_t)2 * n);
        std::vector<uint16_t> lens((size_t)2 * n);
        for (long r = 0; r < 2 * n; r++) start[r] = (uint64_t)r * stride;
        for (long r = 0; r < n; r++) lens[r] = lens[n + r] = (uint16_t)pair_read_len(s, (uint64_t)(first_pair + o + r));
        LHGT_TRY(install_pairs_dev_ascii(ctx, ctx->d_ws_ascii, start.data(), lens.data(), n, nullptr));
    }
    return LHGT_OK;
}

}  // extern "C"
