// k_packed.hip -- a sample kept PACKED on disk (round 6): the resident read store's own records,
// written once by `localhgt_pack`, read back with no parse.
//
// From FASTQ text one GPU's share of a host (16 CPUs) delivers 86-100 M pairs/s and N ranks on that host share the same rate: the host
// side is at copy speed (4 GB/s of text per CPU, 637 bytes per pair).  Only fewer bytes per pair scale it.  A packed sample holds, per
// pair and at a fixed stride, what the loader would have made of the text -- the two read lengths and each mate's three bit-planes (hi
// bit, lo bit, not-a-base: lhgt_common.hpp: ReadBatchDev), 148 bytes for 150-base pairs -- and, in its header, everything the loader
// decides from the TEXT and the run's parameters do not change: the lines every thread of the reference's `-t N` would consume, for
// every N (get_fq_start E:44-89 and the chunk loops E:1019-1026 depend on the bytes around N - 1 file positions; they are computed
// from the text at pack time, lhgt_fastq_thread_chunks), the first pair whose mate 2 lies behind size(fq1) (quirk Q4, E:1419-1445),
// the bases of fq1 (cal_sam_ratio, E:1244-1270).  What depends on the run -- which reads the sampling array keeps (E:1037-1044), by
// global ordinal or by the ordinal inside a thread's chunk -- is decided here, on the GPU, from those tables: the host only reads the
// file into pinned memory (pread) and hands it over.  Only record-aligned pairs of files are packed (same number of records, same
// first read ID, no line beyond the reference's buffers): everything else stays with the FASTQ loader, which is the contract.
#include <algorithm>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>
#include "host_fastx.hpp"

namespace lhgt {

constexpr int PK_MAX_THREADS = 99;         // split_ref holds 100 groups (E:1284); lhgt_set_thread_emulation takes 1..99
struct PackedRule {                         // how a pair's flags follow from its ordinal (host_fastx.cpp: parse_chunk, ThreadPart::keep)
    int threads;                            // 1: global ordinals, quirk Q4 by q4_first_pair; > 1: the thread chunks below
    long q4_first_pair;
    double ratio;
    long first1[PK_MAX_THREADS], count1[PK_MAX_THREADS], first2[PK_MAX_THREADS], count2[PK_MAX_THREADS];
};

// ThreadPart::keep for sequence line g: 1 kept, 0 in a chunk but not sampled, -1 in no chunk
__device__ __forceinline__ int pk_keep(const long* first, const long* count, int threads, long g, double ratio, const float* __restrict__ random_array) {
    for (int i = threads - 1; i >= 0; i--) {      // chunks are in file order and do not overlap (thread_part refuses overlaps)
        if (g < first[i]) continue;
        if (g >= first[i] + count[i]) return -1;
        const long local = g - first[i];
        if (local % 4 != 1) return -1;
        return (ratio >= 100.0 || (double)random_array[(local / 4) % LHGT_MAX_RANDOM] < ratio) ? 1 : 0;
    }
    return -1;
}

// one thread per pair of the chunk: its flags, and for a kept pair its place in the batch (unordered: phases A and C do not depend on
// the order of the pairs) and its words' place
__global__ void __launch_bounds__(256) pk_flags(const uint8_t* __restrict__ raw, long stride, long pair0, long n, PackedRule rule,
                                                const float* __restrict__ random_array, uint8_t* __restrict__ fl_out,
                                                unsigned long long* __restrict__ totals /* [0] kept pairs, [1] their words, [2] k-mer positions (k below), [3] long reads, [4] longest kept read,
                                                                                          [5] records that are none: a length beyond LHGT_MAX_READ_LEN or beyond what the stride holds */,
                                                int k) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    uint32_t words = 0, nkm = 0, nlong = 0, longest = 0;
    uint8_t fl = 0;
    bool corrupt = false;
    if (i < n) {
        const long p = pair0 + i, g = 4 * p + 1;
        const uint16_t* hd = reinterpret_cast<const uint16_t*>(raw + i * stride);
        uint32_t la = hd[0], lb = hd[1];
        // the file is input like any other: a record whose lengths the packer cannot have written (the text loader refuses reads beyond the
        // reference's buffers, E:1004; pk_gather copies 3 (len / 32 + 1) words per mate from the record) is kept out and fails the load
        corrupt = la > (uint32_t)LHGT_MAX_READ_LEN || lb > (uint32_t)LHGT_MAX_READ_LEN ||
                  4 + 4 * (long)(3u * ((la + 31u) / 32u + 1u) + 3u * ((lb + 31u) / 32u + 1u)) > stride;
        if (corrupt) { /* fl stays 0 */ }
        else if (rule.threads > 1) {
            fl = (uint8_t)((pk_keep(rule.first1, rule.count1, rule.threads, g, rule.ratio, random_array) == 1 ? PAIR_COUNT1 | PAIR_VOTE : 0) |
                           (pk_keep(rule.first2, rule.count2, rule.threads, g, rule.ratio, random_array) == 1 ? PAIR_COUNT2 : 0));
        } else {
            const bool s = rule.ratio >= 100.0 || (double)random_array[p % LHGT_MAX_RANDOM] < rule.ratio;
            fl = (uint8_t)(s ? (PAIR_COUNT1 | PAIR_VOTE | (p < rule.q4_first_pair ? PAIR_COUNT2 : 0)) : 0);
        }
        if (fl) {
            // a mate no phase reads is kept empty (host_fastx.cpp: parse_chunk)
            if (!(fl & (PAIR_COUNT1 | PAIR_VOTE))) la = 0;
            if (!(fl & (PAIR_COUNT2 | PAIR_VOTE))) lb = 0;
            words = 3u * ((la + 31u) / 32u + 1u) + 3u * ((lb + 31u) / 32u + 1u);
            nkm = (la >= (uint32_t)k ? la - k + 1 : 0u) + (lb >= (uint32_t)k ? lb - k + 1 : 0u);
            nlong = ((int)la - k + 1 > FAST_NK) + ((int)lb - k + 1 > FAST_NK);
            longest = la > lb ? la : lb;
        }
        fl_out[i] = fl;
    }
    // wave sums, one atomic each
    unsigned long long kept = fl != 0, w = words, km = nkm, lg = nlong;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        kept += __shfl_xor(kept, d, 64); w += __shfl_xor(w, d, 64); km += __shfl_xor(km, d, 64); lg += __shfl_xor(lg, d, 64);
        const uint32_t o = __shfl_xor(longest, d, 64);
        longest = o > longest ? o : longest;
    }
    const unsigned long long n_corrupt = (unsigned long long)__popcll(__ballot(corrupt));
    if ((threadIdx.x & 63) == 0 && n_corrupt) atomicAdd(totals + 5, n_corrupt);
    if ((threadIdx.x & 63) == 0 && kept) {
        atomicAdd(totals, kept); atomicAdd(totals + 1, w); atomicAdd(totals + 2, km); atomicAdd(totals + 3, lg);
        atomicMax(totals + 4, (unsigned long long)longest);
    }
}

// the kept pairs of the chunk into the batch: a wave takes its pairs' places together (one atomic per wave for the pair slots, one
// for the words), a lane copies its own record
__global__ void __launch_bounds__(256) pk_gather(const uint8_t* __restrict__ raw, long stride, long n, const uint8_t* __restrict__ fl_in,
                                                 long n_batch, unsigned long long* __restrict__ cursors /* [0] pairs placed, [1] words placed */,
                                                 uint32_t* __restrict__ words, uint32_t* __restrict__ off, uint16_t* __restrict__ len, uint8_t* __restrict__ flags) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const uint8_t fl = i < n ? fl_in[i] : 0;
    uint32_t la = 0, lb = 0;
    if (fl) {
        const uint16_t* hd = reinterpret_cast<const uint16_t*>(raw + i * stride);
        la = hd[0];
        lb = hd[1];
    }
    const uint32_t wa_src = 3u * ((la + 31u) / 32u + 1u);          // where mate 2's planes start in the record: behind mate 1's as stored
    if (!(fl & (PAIR_COUNT1 | PAIR_VOTE))) la = 0;
    if (!(fl & (PAIR_COUNT2 | PAIR_VOTE))) lb = 0;
    const uint32_t wa = fl ? 3u * ((la + 31u) / 32u + 1u) : 0u, wb = fl ? 3u * ((lb + 31u) / 32u + 1u) : 0u;
    const unsigned long long bal = __ballot(fl != 0);
    if (!bal) return;
    // exclusive prefix of the words over the wave's kept lanes
    uint32_t incl = wa + wb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
    const uint32_t wave_words = __shfl(incl, 63, 64);
    unsigned long long slot0 = 0, word0 = 0;
    if (lane == 0) { slot0 = atomicAdd(cursors, (unsigned long long)__popcll(bal)); word0 = atomicAdd(cursors + 1, (unsigned long long)wave_words); }
    slot0 = __shfl(slot0, 0, 64);
    word0 = __shfl(word0, 0, 64);
    if (!fl) return;
    const long slot = (long)slot0 + __popcll(bal & ((1ull << lane) - 1ull));
    const uint32_t w0 = (uint32_t)word0 + incl - (wa + wb);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(raw + i * stride + 4);
    // a mate that is kept empty has one zero word per plane; a kept mate's planes are copied as stored
    if (la || (fl & (PAIR_COUNT1 | PAIR_VOTE))) { for (uint32_t q = 0; q < wa; q++) words[w0 + q] = src[q]; }
    else { for (uint32_t q = 0; q < wa; q++) words[w0 + q] = 0u; }
    if (lb || (fl & (PAIR_COUNT2 | PAIR_VOTE))) { for (uint32_t q = 0; q < wb; q++) words[w0 + wa + q] = src[wa_src + q]; }
    else { for (uint32_t q = 0; q < wb; q++) words[w0 + wa + q] = 0u; }
    off[slot] = w0;
    off[n_batch + slot] = w0 + wa;
    len[slot] = (uint16_t)la;
    len[n_batch + slot] = (uint16_t)lb;
    flags[slot] = fl;
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

// ---- the resident store as it stands (the packer's source): how many batches, a batch's sizes, its arrays
int lhgt_pairs_batches(lhgt_ctx* ctx, long* n_batches) {
    if (!ctx || !n_batches) LHGT_FAIL(LHGT_E_ARG, "null argument");
    *n_batches = (long)ctx->batches.size();
    return LHGT_OK;
}

int lhgt_pairs_batch_info(lhgt_ctx* ctx, long b, long* n_pairs, unsigned long long* n_words, int* max_len) {
    if (!ctx || b < 0 || b >= (long)ctx->batches.size()) LHGT_FAIL(LHGT_E_ARG, "no such batch");
    const ReadBatch& rb = ctx->batches[(size_t)b];
    if (n_pairs) *n_pairs = rb.d.n_pairs;
    if (n_words) *n_words = rb.n_words;
    if (max_len) *max_len = rb.max_len;
    return LHGT_OK;
}

// Writes every resident pair, batch after batch, at `stride` bytes per pair from byte `data_offset` of `path` on: [u16 len1][u16 len2]
// [mate 1: 3 planes of len1 / 32 + 1 words][mate 2 likewise].  The store must be what lhgt_pairs_load_fastq makes of a record-aligned
// pair of files with every read kept and no thread emulation: every pair voted and counted, mate 2 uncounted from some pair on at
// most (quirk Q4).  Returns the pairs written, the first pair whose mate 2 is not counted (the number of pairs if none) and the
// bases of all mates 1 (cal_sam_ratio's count, E:1264-1265).
int lhgt_pairs_store_write(lhgt_ctx* ctx, const char* path, unsigned long long data_offset, long stride, long* n_pairs, long* q4_first_pair,
                           unsigned long long* bases1) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !path || stride < 8 || (stride & 3)) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    const int fd = ::open(path, O_WRONLY);
    if (fd < 0) LHGT_FAIL(LHGT_E_IO, "cannot open %s for writing", path);
    long total = 0, q4 = -1;
    unsigned long long bases = 0;
    int rc = LHGT_OK;
    for (const ReadBatch& rb : ctx->batches) {
        const long n = rb.d.n_pairs;
        if (!n) continue;
        std::vector<uint32_t> words((size_t)rb.n_words), off((size_t)2 * n);
        std::vector<uint16_t> len((size_t)2 * n);
        std::vector<uint8_t> fl((size_t)n, PAIR_ALL);
        hipError_t e = hipMemcpyAsync(words.data(), rb.d.words, (size_t)rb.n_words * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(off.data(), rb.d.off[0], (size_t)2 * n * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(len.data(), rb.d.len[0], (size_t)2 * n * 2, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rb.d.flags) e = hipMemcpyAsync(fl.data(), rb.d.flags, (size_t)n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { set_error("store export failed: %s", hipGetErrorString(e)); rc = LHGT_E_HIP; break; }
        // written in slices of 64 Ki pairs, assembled by all threads
        const long SL = 65536;
        std::vector<uint8_t> buf((size_t)std::min(n, SL) * (size_t)stride);
        for (long p0 = 0; p0 < n && rc == LHGT_OK; p0 += SL) {
            const long m = std::min(SL, n - p0);
            memset(buf.data(), 0, (size_t)m * (size_t)stride);
            std::atomic<int> bad{0};
            parallel_for((m + 4095) / 4096, ingest_default_threads(), [&](long c) {
                for (long i = c * 4096; i < std::min(m, (c + 1) * 4096); i++) {
                    const long p = p0 + i;
                    const uint32_t la = len[(size_t)p], lb = len[(size_t)(n + p)];
                    const uint32_t wa = 3u * ((la + 31u) / 32u + 1u), wb = 3u * ((lb + 31u) / 32u + 1u);
                    if (4 + 4 * (size_t)(wa + wb) > (size_t)stride) { bad = 1; continue; }
                    uint8_t* r = buf.data() + (size_t)i * (size_t)stride;
                    const uint16_t hd[2] = {(uint16_t)la, (uint16_t)lb};
                    memcpy(r, hd, 4);
                    memcpy(r + 4, words.data() + off[(size_t)p], 4 * (size_t)wa);
                    memcpy(r + 4 + 4 * (size_t)wa, words.data() + off[(size_t)(n + p)], 4 * (size_t)wb);
                }
            });
            if (bad) { set_error("a pair's record does not fit the stride of %ld bytes", stride); rc = LHGT_E_ARG; break; }
            for (long i = 0; i < m; i++) {
                const long p = p0 + i;
                const uint8_t f = fl[(size_t)p];
                // the store of a clean pair of files under -t 1 with every read kept: 7 everywhere, then 5 (mate 2 behind size(fq1)) to the end
                if (f == PAIR_ALL && q4 < 0) { /* counted */ }
                else if (f == (PAIR_COUNT1 | PAIR_VOTE)) { if (q4 < 0) q4 = total + p; }
                else { set_error("pair %ld carries flags %d: not a record-aligned pair of files read whole (the FASTQ loader is the way for those)", total + p, (int)f); rc = LHGT_E_FORMAT; break; }
                bases += len[(size_t)p];
            }
            if (rc != LHGT_OK) break;
            const size_t want = (size_t)m * (size_t)stride;
            size_t done = 0;
            while (done < want) {
                const ssize_t w = pwrite(fd, buf.data() + done, want - done, (off_t)(data_offset + (unsigned long long)(total + p0) * (unsigned long long)stride + done));
                if (w <= 0) { set_error("write to %s failed", path); rc = LHGT_E_IO; break; }
                done += (size_t)w;
            }
        }
        if (rc != LHGT_OK) break;
        total += n;
    }
    close(fd);
    if (rc != LHGT_OK) return rc;
    if (n_pairs) *n_pairs = total;
    if (q4_first_pair) *q4_first_pair = q4 < 0 ? total : q4;
    if (bases1) *bases1 = bases;
    return LHGT_OK;
}

}  // extern "C"

// records [p0, p0 + m) of the file into dst, by all host threads
static bool packed_read(int fd, unsigned long long data_offset, long stride, long p0, long m, uint8_t* dst, int nthreads) {
    std::atomic<int> bad{0};
    const long pieces = std::min<long>(4L * nthreads, std::max<long>(1, m / 4096));
    parallel_for(pieces, nthreads, [&](long q) {
        const long a = m * q / pieces, b = m * (q + 1) / pieces;
        size_t want = (size_t)(b - a) * (size_t)stride, done = 0;
        while (done < want) {
            const ssize_t r = pread(fd, dst + (size_t)a * (size_t)stride + done, want - done, (off_t)(data_offset + (unsigned long long)(p0 + a) * (unsigned long long)stride + done));
            if (r <= 0) { bad = 1; return; }
            done += (size_t)r;
        }
    });
    return !bad;
}

extern "C" {

// Measurement handle (tools/ingest_scaling.py), no GPU: the host side of lhgt_pairs_load_packed alone -- part `part` of `n_parts` of the
// records read in the loader's chunks into two buffers by `threads` host threads (0: the loader's own choice).
int lhgt_packed_read_rate(const char* path, unsigned long long data_offset, long stride, long n_pairs_total, int part, int n_parts, int threads, double* seconds) {
    if (!path || stride < 8 || n_pairs_total < 0 || n_parts < 1 || part < 0 || part >= n_parts) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) LHGT_FAIL(LHGT_E_IO, "cannot open %s", path);
    const long p_lo = (long)((__int128)n_pairs_total * part / n_parts), p_hi = (long)((__int128)n_pairs_total * (part + 1) / n_parts);
    const long CH = std::min<long>(2L << 20, std::max<long>(1, (long)(((size_t)1 << 29) / (size_t)stride)));
    const size_t raw_bytes = (size_t)CH * (size_t)stride;
    std::vector<uint8_t> buf(2 * raw_bytes, 1);          // touched: a loader's pinned buffers are
    const int nthreads = threads > 0 ? threads : ingest_default_threads();
    const double t0 = now_s();
    int c = 0;
    bool ok = true;
    for (long p = p_lo; p < p_hi && ok; p += CH, c++) ok = packed_read(fd, data_offset, stride, p, std::min(CH, p_hi - p), buf.data() + (size_t)(c & 1) * raw_bytes, nthreads);
    if (seconds) *seconds = now_s() - t0;
    close(fd);
    if (!ok) LHGT_FAIL(LHGT_E_IO, "%s: the packed records end early", path);
    return LHGT_OK;
}

// Part `part` of `n_parts` of a packed sample (a contiguous run of its pairs: read ordinals are global) becomes resident: the records
// are read into pinned memory by all host threads, go to the GPU as they are, and the GPU decides which pairs the run keeps (the
// sampling array by global ordinal and quirk Q4 at threads = 1; the thread chunks first1 / count1 / first2 / count2 of the reference's
// -t threads otherwise -- host_fastx.cpp: parse_chunk's rules), and lays the kept ones out as batches.
int lhgt_pairs_load_packed(lhgt_ctx* ctx, const char* path, unsigned long long data_offset, long stride, long n_pairs_total, long q4_first_pair,
                           double ratio_percent, int threads, const long* first1, const long* count1, const long* first2, const long* count2,
                           int part, int n_parts, long* n_seen, long* n_kept) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !path || stride < 8 || (stride & 3) || n_pairs_total < 0 || n_parts < 1 || part < 0 || part >= n_parts || threads < 1 || threads > PK_MAX_THREADS)
        LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (threads > 1 && (!first1 || !count1 || !first2 || !count2)) LHGT_FAIL(LHGT_E_ARG, "thread chunks missing");
    sampling_join(ctx);
    const bool need_array = ratio_percent < 100.0;
    if (need_array && ctx->random_array.empty()) LHGT_FAIL(LHGT_E_STATE, "lhgt_sampling_init must precede a sampled load");
    if (need_array && ctx->sampling_filled < std::min<long>(n_pairs_total, LHGT_MAX_RANDOM))
        LHGT_FAIL(LHGT_E_STATE, "the sampling array holds %ld entries, the sample has %ld reads per file (lhgt_sampling_reserve)", ctx->sampling_filled, n_pairs_total);
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) LHGT_FAIL(LHGT_E_IO, "cannot open %s", path);
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    const long p_lo = (long)((__int128)n_pairs_total * part / n_parts), p_hi = (long)((__int128)n_pairs_total * (part + 1) / n_parts);
    PackedRule rule{};
    rule.threads = threads;
    rule.q4_first_pair = q4_first_pair;
    rule.ratio = ratio_percent;
    for (int i = 0; i < threads && threads > 1; i++) { rule.first1[i] = first1[i]; rule.count1[i] = count1[i]; rule.first2[i] = first2[i]; rule.count2[i] = count2[i]; }
    // the sampling array on the device: the entries a read of this sample can look at
    float* d_random = nullptr;
    if (need_array) {
        const size_t n_ent = (size_t)std::min<long>(std::max<long>(n_pairs_total, 1), LHGT_MAX_RANDOM);
        LHGT_HIP(dev_malloc(&d_random, n_ent * 4));
        LHGT_HIP(hipMemcpyAsync(d_random, ctx->random_array.data(), n_ent * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    // chunks of <= 2 Mi pairs through two pinned host buffers and two device buffers: while chunk c is copied and its flags are
    // made, the host reads chunk c + 1; the one wait per chunk is for its totals (kept pairs and words size the batch), and behind
    // it the next chunk's copy and the gather of this one are queued together.  The pinned buffers stay with the context (a
    // session loads sample after sample; pinning 600 MB costs ~0.1 s).
    const long CH = std::min<long>(2L << 20, std::max<long>(1, (long)(((size_t)1 << 29) / (size_t)stride)));
    const size_t raw_bytes = (size_t)CH * (size_t)stride;
    uint8_t* d_raw[2] = {nullptr, nullptr};
    uint8_t* d_fl[2] = {nullptr, nullptr};
    unsigned long long* d_tot = nullptr;        // [c & 1][8]: totals of chunk c, gather cursors behind them
    int rc = LHGT_OK;
    if (ctx->packed_stage_bytes < 2 * raw_bytes) {
        if (ctx->h_packed_stage) { (void)hipHostFree(ctx->h_packed_stage); ctx->h_packed_stage = nullptr; ctx->packed_stage_bytes = 0; }
        if (hipHostMalloc((void**)&ctx->h_packed_stage, 2 * raw_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("no pinned memory for the packed records"); rc = LHGT_E_NOMEM; }
        else ctx->packed_stage_bytes = 2 * raw_bytes;
    }
    uint8_t* h_raw[2] = {ctx->h_packed_stage, ctx->h_packed_stage ? ctx->h_packed_stage + raw_bytes : nullptr};
    for (int i = 0; i < 2 && rc == LHGT_OK; i++)
        if (dev_malloc(&d_raw[i], raw_bytes) != hipSuccess || dev_malloc(&d_fl[i], (size_t)CH) != hipSuccess) { set_error("no device memory for the packed records"); rc = LHGT_E_NOMEM; }
    if (rc == LHGT_OK && dev_malloc(&d_tot, 256) != hipSuccess) { set_error("no device memory"); rc = LHGT_E_NOMEM; }
    unsigned long long* h_tot = nullptr;        // pinned: the totals come back without a staging copy
    if (rc == LHGT_OK && hipHostMalloc((void**)&h_tot, 256, hipHostMallocDefault) != hipSuccess) { set_error("no pinned memory"); rc = LHGT_E_NOMEM; }
    const int nthreads = ingest_default_threads();
    auto read_chunk = [&](long p0, long m, uint8_t* dst) -> int {
        if (!packed_read(fd, data_offset, stride, p0, m, dst, nthreads)) { set_error("%s: the packed records end before pair %ld", path, p0 + m); return LHGT_E_IO; }
        return LHGT_OK;
    };
    // chunk c: copy + flags + totals, all queued
    auto enqueue = [&](int c, long p0, long m) -> int {
        const int s = c & 1;
        if (hipMemcpyAsync(d_raw[s], h_raw[s], (size_t)m * (size_t)stride, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(d_tot + 16 * s, 0, 128, ctx->stream) != hipSuccess) { set_error("upload of the packed records failed"); return LHGT_E_HIP; }
        hipLaunchKernelGGL(pk_flags, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_raw[s], stride, p0, m, rule, d_random, d_fl[s], d_tot + 16 * s, ctx->k);
        if (hipMemcpyAsync(h_tot + 16 * s, d_tot + 16 * s, 48, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipEventRecord(s ? ctx->ev3 : ctx->ev2, ctx->stream) != hipSuccess) { set_error("copy"); return LHGT_E_HIP; }
        return LHGT_OK;
    };
    long kept_total = 0;
    double t_read = 0, t_dev = 0;
    const double t_all = now_s();
    if (rc == LHGT_OK && p_hi > p_lo) {
        long p_at = p_lo;
        int c = 0;
        long m_cur = std::min(CH, p_hi - p_at);
        double t0 = now_s();
        rc = read_chunk(p_at, m_cur, h_raw[0]);
        t_read += now_s() - t0;
        if (rc == LHGT_OK) rc = enqueue(0, p_at, m_cur);
        while (rc == LHGT_OK && m_cur > 0) {
            const long m = m_cur;
            const int s = c & 1;
            p_at += m;
            const long m_next = std::min(CH, p_hi - p_at);
            if (m_next > 0) {         // the next chunk's records meanwhile (its host buffer's last copy was waited for two chunks ago)
                t0 = now_s();
                rc = read_chunk(p_at, m_next, h_raw[s ^ 1]);
                t_read += now_s() - t0;
                if (rc != LHGT_OK) break;
            }
            t0 = now_s();
            if (hipEventSynchronize(s ? ctx->ev3 : ctx->ev2) != hipSuccess) { set_error("the packed records' flags kernel failed: %s", hipGetErrorString(hipGetLastError())); rc = LHGT_E_HIP; break; }
            unsigned long long tot[6];
            memcpy(tot, h_tot + 16 * s, 48);
            if (tot[5]) {
                set_error("%s: %llu of the records of pairs %ld .. %ld hold read lengths no packed sample holds (beyond %d bases or beyond the stride of %ld bytes): not a packed sample, or damaged",
                          path, tot[5], p_at - m, p_at - 1, LHGT_MAX_READ_LEN, stride);
                rc = LHGT_E_FORMAT;
                break;
            }
            if (m_next > 0) { rc = enqueue(c + 1, p_at, m_next); if (rc != LHGT_OK) break; }      // behind this chunk's flags, in front of its gather
            const long nb = (long)tot[0];
            if (nb > 0) {
                if (tot[1] >= (1ull << 32)) { set_error("batch too large: %llu plane words", tot[1]); rc = LHGT_E_ARG; break; }
                ReadBatch b;
                b.n_words = tot[1];
                b.n_kmers = tot[2];
                b.n_long = (long)tot[3];
                b.max_len = (int)tot[4];
                const size_t words_b = (tot[1] * 4 + 16 + 255) & ~(size_t)255, off_b = ((size_t)2 * nb * 4 + 255) & ~(size_t)255, len_b = ((size_t)2 * nb * 2 + 255) & ~(size_t)255;
                uint8_t* blk = nullptr;
                if (dev_malloc(&blk, words_b + off_b + len_b + (size_t)nb) != hipSuccess) { set_error("no device memory for a batch of %ld pairs", nb); rc = LHGT_E_NOMEM; break; }
                b.alloc[0] = blk;
                uint32_t* d_words = (uint32_t*)blk;
                uint32_t* d_off = (uint32_t*)(blk + words_b);
                uint16_t* d_len = (uint16_t*)(blk + words_b + off_b);
                uint8_t* d_flags = blk + words_b + off_b + len_b;
                hipLaunchKernelGGL(pk_gather, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_raw[s], stride, m, d_fl[s], nb, d_tot + 16 * s + 8, d_words, d_off, d_len, d_flags);
                if (hipGetLastError() != hipSuccess) { dev_free(blk); set_error("the packed records' gather failed"); rc = LHGT_E_HIP; break; }
                b.d.words = d_words;
                b.d.off[0] = d_off;
                b.d.off[1] = d_off + nb;
                b.d.len[0] = d_len;
                b.d.len[1] = d_len + nb;
                b.d.flags = d_flags;
                b.d.n_pairs = nb;
                ctx->batches.push_back(b);
                ctx->store_gen++;
                ctx->n_pairs += nb;
                kept_total += nb;
            }
            t_dev += now_s() - t0;
            c++;
            m_cur = m_next;
        }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == LHGT_OK) { set_error("the packed records' gather failed: %s", hipGetErrorString(hipGetLastError())); rc = LHGT_E_HIP; }
    if (h_tot) (void)hipHostFree(h_tot);
    for (void* p : {(void*)d_raw[0], (void*)d_raw[1], (void*)d_fl[0], (void*)d_fl[1], (void*)d_tot, (void*)d_random}) if (p) dev_free(p);
    if (rc != LHGT_OK) return rc;
    if (ingest_trace())
        fprintf(stderr, "[lhgt ingest] packed part %d/%d: pairs [%ld, %ld) of %ld, %.1f MB of records: read %.3fs (%d threads), device %.3fs, %ld kept, %.3fs in all\n",
                part, n_parts, p_lo, p_hi, n_pairs_total, 1e-6 * (double)(p_hi - p_lo) * (double)stride, t_read, nthreads, t_dev, kept_total, now_s() - t_all);
    if (n_seen) *n_seen = p_hi - p_lo;
    if (n_kept) *n_kept = kept_total;
    return LHGT_OK;
}

}  // extern "C"
