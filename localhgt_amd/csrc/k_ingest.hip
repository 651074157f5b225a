// k_ingest.hip -- 2-bit packing of bases, the resident pair store, the index builder/loader.
#include <cstring>
#include <time.h>
#include <thread>
#include "lhgt_hash.hpp"

namespace lhgt {

// ---------------------------------------------------------------- pack kernel
// Thread (r, w) turns bases [32w, 32w+32) of sequence r into one word of each plane.
// Sequence r is ascii[start[r] .. start[r]+len); len = len16[r], or single_len when len16 is null (one contig).
// Its record starts at words[word_off[r]] and is [hi | lo | not-a-base], wpr = ceil(len/32)+1 words each
// (last word = zero pad).
__global__ void __launch_bounds__(256) pack_bases(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ start,
                                                  const uint16_t* __restrict__ len16, long single_len,
                                                  const uint64_t* __restrict__ word_off, long n_seq, int max_wpr,
                                                  uint32_t* __restrict__ words) {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long r = t / max_wpr;
    int w = (int)(t % max_wpr);
    if (r >= n_seq) return;
    uint64_t b0 = start[r];
    long len = len16 ? (long)len16[r] : single_len;
    int wpr = (int)((len + 31) / 32) + 1;
    if (w >= wpr) return;
    uint32_t hi = 0, lo = 0, nb = 0;
    long base = 32L * w;
    const uint8_t* s = ascii + b0 + base;
    int n = (int)(len - base < 32 ? (len - base < 0 ? 0 : len - base) : 32);
    for (int b = 0; b < n; b++) {
        uint32_t c = base_code(s[b]);
        uint32_t bit = 0x80000000u >> b;
        if (c == 4) nb |= bit;
        else {
            if (c & 2) hi |= bit;
            if (c & 1) lo |= bit;
        }
    }
    uint32_t* rec = words + word_off[r];
    rec[w] = hi;
    rec[wpr + w] = lo;
    rec[2 * wpr + w] = nb;
}

// the loader's form: start offsets and word offsets as u32, mate 2 of pair p at index n + p (the batch's own layout)
__global__ void __launch_bounds__(256) pack_bases32(const uint8_t* __restrict__ ascii, const uint32_t* __restrict__ start,
                                                    const uint16_t* __restrict__ len16, const uint32_t* __restrict__ word_off,
                                                    long n_seq, int max_wpr, uint32_t* __restrict__ words) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = t / max_wpr;
    const int w = (int)(t % max_wpr);
    if (r >= n_seq) return;
    const long len = (long)len16[r];
    const int wpr = (int)((len + 31) / 32) + 1;
    if (w >= wpr) return;
    uint32_t hi = 0, lo = 0, nb = 0;
    const long base = 32L * w;
    const uint8_t* s = ascii + start[r] + base;
    const int n = (int)(len - base < 32 ? (len - base < 0 ? 0 : len - base) : 32);
    for (int b = 0; b < n; b++) {
        const uint32_t c = base_code(s[b]);
        const uint32_t bit = 0x80000000u >> b;
        if (c == 4) nb |= bit;
        else {
            if (c & 2) hi |= bit;
            if (c & 1) lo |= bit;
        }
    }
    uint32_t* rec = words + word_off[r];
    rec[w] = hi;
    rec[wpr + w] = lo;
    rec[2 * wpr + w] = nb;
}

// ---------------------------------------------------------------- hash every position of one packed sequence
__global__ void __launch_bounds__(256) hash_positions(const uint32_t* __restrict__ rec, long len, HashParams hp,
                                                      uint32_t* __restrict__ out, uint8_t* __restrict__ out_valid) {
    long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long nk = len - hp.k + 1;
    if (j >= nk) return;
    int wpr = (int)((len + 31) / 32) + 1;
    // windows are cut relative to a word-aligned base so the 32-bit in-word offset stays small
    const uint32_t* hi = rec + (j >> 5);
    int jj = (int)(j & 31);
    uint32_t whi = plane_window(hi, jj, hp.k);
    uint32_t wlo = plane_window(hi + wpr, jj, hp.k);
    uint32_t wnb = plane_window(hi + 2 * wpr, jj, hp.k);
    uint32_t rhi = brev_k(whi, hp.k), rlo = brev_k(wlo, hp.k);
    bool valid = wnb == 0;
    for (int i = 0; i < hp.e; i++) {
        uint32_t h = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
        out[j * hp.e + i] = valid ? h : 0u;  // invalid k-mers are stored as hash 0 (E:808-810, quirk Q6)
    }
    if (out_valid) out_valid[j] = valid;
}

// ---------------------------------------------------------------- hash every position of MANY sequences in one launch
// The sequences lie back to back in one packed span (planes of span_len bases, wps words each).  Thread = span position x;
// its sequence c is found by binary search in coff[0..n_c] (offsets inside the span), j = x - coff[c].  Sequences whose
// out_word is ~0 (length <= k: not indexed, E:772) are skipped; a window never crosses into the next sequence because
// j <= len - k.  One launch per span instead of two per contig: a catalogue like UHGG has hundreds of thousands of contigs.
__global__ void __launch_bounds__(256) hash_span_positions(const uint32_t* __restrict__ planes, long span_len, int wps,
                                                           const uint64_t* __restrict__ coff, int n_c,
                                                           const uint64_t* __restrict__ out_word, HashParams hp,
                                                           uint32_t* __restrict__ out) {
    const long x = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= span_len) return;
    int lo = 0, hi = n_c;                 // last c with coff[c] <= x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long)coff[mid] <= x) lo = mid; else hi = mid; }
    const long j = x - (long)coff[lo], len = (long)(coff[lo + 1] - coff[lo]);
    const uint64_t ow = out_word[lo];
    if (ow == ~0ull || j + hp.k > len) return;
    const uint32_t* hiw = planes + (x >> 5);
    const int r = (int)(x & 31);
    const uint32_t whi = plane_window(hiw, r, hp.k), wlo = plane_window(hiw + wps, r, hp.k), wnb = plane_window(hiw + 2 * wps, r, hp.k);
    const uint32_t rhi = brev_k(whi, hp.k), rlo = brev_k(wlo, hp.k);
    uint32_t* o = out + ow + (uint64_t)j * hp.e;
    for (int i = 0; i < hp.e; i++) o[i] = wnb == 0 ? hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]) : 0u;   // quirk Q6
}

// ---------------------------------------------------------------- the packed form of the resident reference
// A span of back-to-back sequences (ASCII, device) -> the flat bit-planes.  The span's indexed contigs occupy the flat positions
// [fb[0], fb[n]); contig c starts at flat position fb[c] and at span offset src[c] (sequences that are not indexed lie between
// them in the span and are skipped).  Thread = one word of the planes; words at the ends of the range are shared with the
// neighbouring spans, hence atomicOr into planes that were cleared when the layout was made.
__global__ void __launch_bounds__(256) pack_span_flat(const uint8_t* __restrict__ ascii, const uint64_t* __restrict__ fb,
                                                      const uint64_t* __restrict__ src, int n, uint32_t* __restrict__ planes,
                                                      uint64_t plane_words) {
    const uint64_t F0 = fb[0], F1 = fb[n];
    const uint64_t w = (F0 >> 5) + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (F1 == F0 || w > ((F1 - 1) >> 5)) return;
    const uint64_t x0 = w * 32 > F0 ? w * 32 : F0, x1 = w * 32 + 32 < F1 ? w * 32 + 32 : F1;
    int lo = 0, hi = n;                   // last c with fb[c] <= x0
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fb[mid] <= x0) lo = mid; else hi = mid; }
    int c = lo;
    uint64_t c_end = fb[c + 1];
    uint64_t c_src = src[c], c_fb = fb[c];       // flat position x of contig c is ascii[c_src + (x - c_fb)]
    uint32_t whi = 0, wlo = 0, wnb = 0;
    for (uint64_t x = x0; x < x1; x++) {
        while (x >= c_end) { c++; c_end = fb[c + 1]; c_src = src[c]; c_fb = fb[c]; }
        const uint32_t code = base_code(ascii[c_src + (x - c_fb)]);
        const uint32_t bit = 0x80000000u >> (x & 31);
        if (code == 4) wnb |= bit;
        else {
            if (code & 2) whi |= bit;
            if (code & 1) wlo |= bit;
        }
    }
    // the hi and lo planes interleaved word by word, the not-a-base plane behind them (round 5): a k-mer's 2 x 2 words of bases are 16
    // consecutive bytes -- ONE load and mostly one memory line for a position looked up on its own (k_scan.hip: ref_flags_slots; its
    // positions are known to hold a k-mer) where three separate planes were six loads in three lines
    if (whi) atomicOr(planes + 2 * w, whi);
    if (wlo) atomicOr(planes + 2 * w + 1, wlo);
    if (wnb) atomicOr(planes + 2 * plane_words + w, wnb);
}

// ---------------------------------------------------------------- FASTA text -> its sequences back to back
// The text of a stretch of a FASTA file lies in device memory as it is on disk (header lines, newlines and all); the host has
// found the '>' lines and counted the newlines (host_fastx.cpp: FastaIndex).  seg[2s], seg[2s+1] = the bytes [begin, end) of
// sequence s's lines inside the text; kept_before[b] = how many bytes of sequences that are not '\n' precede text block b
// (FASTA_BLK bytes each).  A kept byte goes to out[kept_before + its rank inside the block]: the same stream of bases std::getline
// and string concatenation build in read_ref (E:761-880; a '\r' stays in, as there), without a host pass over the bases.
// Thread = 16 consecutive bytes, workgroup = one block.
__global__ void __launch_bounds__(FASTA_BLK / 16) strip_fasta_block(const uint8_t* __restrict__ text, uint64_t text_len,
                                                                    const uint64_t* __restrict__ kept_before,
                                                                    const uint64_t* __restrict__ seg, int n_seg,
                                                                    uint8_t* __restrict__ out) {
    __shared__ int wsum[FASTA_BLK / 16 / 64];
    const uint64_t x0 = (uint64_t)blockIdx.x * FASTA_BLK + (uint64_t)threadIdx.x * 16;
    uint8_t c[16];
    uint32_t keep = 0;
    if (x0 < text_len) {
        if (x0 + 16 <= text_len) {
            const uint4 v = *(const uint4*)(text + x0);
            memcpy(c, &v, 16);
        } else {
            for (int i = 0; i < 16; i++) c[i] = x0 + i < text_len ? text[x0 + i] : (uint8_t)'\n';
        }
        int lo = -1, hi = n_seg;              // last sequence whose lines begin at or before x0 (-1: none)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg[2 * mid] <= x0) lo = mid; else hi = mid; }
        int s = lo;
        uint64_t s_beg = s >= 0 ? seg[2 * s] : 0, s_end = s >= 0 ? seg[2 * s + 1] : 0;
        uint64_t nxt = s + 1 < n_seg ? seg[2 * (s + 1)] : ~0ull;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint64_t x = x0 + i;
            while (x >= nxt) { s++; s_beg = nxt; s_end = seg[2 * s + 1]; nxt = s + 1 < n_seg ? seg[2 * (s + 1)] : ~0ull; }
            if (s >= 0 && x >= s_beg && x < s_end && c[i] != '\n') keep |= 1u << i;
        }
    }
    const int cnt = __popc(keep), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int off = 0;
    for (int q = 0; q < wv; q++) off += wsum[q];
    uint8_t* o = out + kept_before[blockIdx.x] + (uint64_t)(off + incl - cnt);
    for (int i = 0; i < 16; i++)
        if ((keep >> i) & 1u) *o++ = c[i];
}

// the length word in front of every contig's hashes in the resident index ([u32 len][(len-k+1)*e u32], E:785, 847)
__global__ void __launch_bounds__(256) write_contig_lens(const ContigDev* __restrict__ contigs, long n, uint32_t* __restrict__ index) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) index[contigs[c].hash_word - 1] = contigs[c].len;
}

int ws_reserve(lhgt_ctx* ctx, size_t ascii_bytes, size_t plane_words) {
    if (ascii_bytes > ctx->ws_ascii_cap) {
        if (ctx->d_ws_ascii) lhgt::dev_free(ctx->d_ws_ascii);
        ctx->ws_ascii_cap = ascii_bytes + ascii_bytes / 4 + 4096;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_ws_ascii, ctx->ws_ascii_cap));
    }
    if (plane_words > ctx->ws_words_cap) {
        if (ctx->d_ws_words) lhgt::dev_free(ctx->d_ws_words);
        ctx->ws_words_cap = plane_words + plane_words / 4 + 1024;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_ws_words, ctx->ws_words_cap * 4));
    }
    return LHGT_OK;
}

// pack one sequence that already sits in device memory and hash all its positions into d_out
int hash_contig_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long len, uint32_t* d_out, uint8_t* d_valid) {
    if (len < ctx->k) return LHGT_OK;
    int wpr = (int)((len + 31) / 32) + 1;
    LHGT_TRY(ws_reserve(ctx, 0, (size_t)3 * wpr + 8));
    uint64_t offs[2] = {0, 0};  // start[0], word_off[0]
    uint64_t* d_meta = (uint64_t*)(ctx->d_ws_words + (size_t)3 * wpr + 2 - ((size_t)3 * wpr) % 2);
    LHGT_HIP(hipMemcpyAsync(d_meta, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    long threads = wpr;
    hipLaunchKernelGGL(pack_bases, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_ascii,
                       d_meta, (const uint16_t*)nullptr, len, d_meta + 1, 1L, wpr, ctx->d_ws_words);
    long nk = len - ctx->k + 1;
    hipLaunchKernelGGL(hash_positions, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ws_words, len,
                       ctx->hp, d_out, d_valid);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));  // offs is on the stack; the workspace is reused by the next call
    return LHGT_OK;
}

// Pack a span of back-to-back sequences that sits in device memory (d_ascii[0 .. span_len)) and hash all their positions:
// sequence c of the span covers [coff[c], coff[c+1]) and its hashes go to d_out[out_word[c] ...] (out_word ~0 = skip).
int hash_span_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long span_len, const uint64_t* coff, const uint64_t* out_word,
                        long n_c, uint32_t* d_out) {
    if (span_len <= 0 || n_c <= 0) return LHGT_OK;
    const int wps = (int)((span_len + 31) / 32) + 1;
    const size_t meta_words = 2 * ((size_t)2 * n_c + 4);
    LHGT_TRY(ws_reserve(ctx, 0, (size_t)3 * wps + 8 + meta_words));
    uint64_t* d_meta = (uint64_t*)(ctx->d_ws_words + (((size_t)3 * wps + 3) & ~(size_t)1));
    uint64_t* d_coff = d_meta + 2;
    uint64_t* d_ow = d_coff + n_c + 1;
    const uint64_t offs[2] = {0, 0};
    LHGT_HIP(hipMemcpyAsync(d_meta, offs, sizeof offs, hipMemcpyHostToDevice, ctx->stream));
    LHGT_HIP(hipMemcpyAsync(d_coff, coff, (size_t)(n_c + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    LHGT_HIP(hipMemcpyAsync(d_ow, out_word, (size_t)n_c * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(pack_bases, dim3((unsigned)((wps + 255) / 256)), dim3(256), 0, ctx->stream, d_ascii, d_meta,
                       (const uint16_t*)nullptr, span_len, d_meta + 1, 1L, wps, ctx->d_ws_words);
    hipLaunchKernelGGL(hash_span_positions, dim3((unsigned)((span_len + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ws_words,
                       span_len, wps, d_coff, (int)n_c, d_ow, ctx->hp, d_out);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));   // the host arrays and the workspace are reused by the next span
    return LHGT_OK;
}

int install_span_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, long span_len, const uint64_t* coff, const long* contig_of, long n_c) {
    if (span_len <= 0 || n_c <= 0) return LHGT_OK;
    if (!ctx->ref_packed) {
        std::vector<uint64_t> ow((size_t)n_c);
        for (long c = 0; c < n_c; c++) ow[c] = contig_of[c] < 0 ? ~0ull : ctx->contigs[(size_t)contig_of[c]].hash_word;
        return hash_span_dev_ascii(ctx, d_ascii, span_len, coff, ow.data(), n_c, ctx->d_index);
    }
    std::vector<uint64_t> meta;           // fb[0..n] then src[0..n)
    std::vector<uint64_t> src;
    for (long c = 0; c < n_c; c++)
        if (contig_of[c] >= 0) {
            const ContigDev& cd = ctx->contigs[(size_t)contig_of[c]];
            if (!meta.empty() && meta.back() != cd.flat_base) LHGT_FAIL(LHGT_E_STATE, "span contigs are not consecutive in the flat layout");
            if (meta.empty()) meta.push_back(cd.flat_base);
            meta.push_back(cd.flat_base + cd.len);
            src.push_back(coff[c]);
        }
    const long n = (long)src.size();
    if (n == 0) return LHGT_OK;
    const uint64_t F0 = meta.front(), F1 = meta.back();
    meta.insert(meta.end(), src.begin(), src.end());
    LHGT_TRY(ws_reserve(ctx, 0, meta.size() * 2 + 8));
    uint64_t* d_meta = (uint64_t*)ctx->d_ws_words;
    LHGT_HIP(hipMemcpyAsync(d_meta, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    const uint64_t n_w = ((F1 - 1) >> 5) - (F0 >> 5) + 1;
    hipLaunchKernelGGL(pack_span_flat, dim3((unsigned)((n_w + 255) / 256)), dim3(256), 0, ctx->stream, d_ascii, d_meta, d_meta + n + 1, (int)n,
                       ctx->d_ref_planes, (uint64_t)ctx->ref_plane_words);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));   // the host arrays and the workspace are reused by the next span
    return LHGT_OK;
}

// d_text[0 .. text_len) -> d_out: see strip_fasta_block.  kept_before (n_blocks entries) and seg (2 * n_seg entries) are host
// arrays; they travel through the word workspace.
int strip_fasta_text(lhgt_ctx* ctx, const uint8_t* d_text, uint64_t text_len, const uint64_t* kept_before, long n_blocks,
                     const uint64_t* seg, long n_seg, uint8_t* d_out) {
    if (text_len == 0 || n_blocks <= 0) return LHGT_OK;
    if (n_blocks != (long)((text_len + FASTA_BLK - 1) / FASTA_BLK)) LHGT_FAIL(LHGT_E_ARG, "strip_fasta_text: %ld blocks for %llu bytes", n_blocks, (unsigned long long)text_len);
    const size_t n64 = (size_t)n_blocks + 2 * (size_t)n_seg;
    LHGT_TRY(ws_reserve(ctx, 0, 2 * n64 + 8));
    uint64_t* d_meta = (uint64_t*)ctx->d_ws_words;
    LHGT_HIP(hipMemcpyAsync(d_meta, kept_before, (size_t)n_blocks * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_seg) LHGT_HIP(hipMemcpyAsync(d_meta + n_blocks, seg, (size_t)n_seg * 16, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(strip_fasta_block, dim3((unsigned)n_blocks), dim3(FASTA_BLK / 16), 0, ctx->stream, d_text, text_len, d_meta,
                       d_meta + n_blocks, (int)n_seg, d_out);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));   // the host arrays and the workspace are reused
    return LHGT_OK;
}

int write_index_lens(lhgt_ctx* ctx) {
    const long n = (long)ctx->contigs.size();
    if (n == 0 || ctx->ref_packed) return LHGT_OK;
    hipLaunchKernelGGL(write_contig_lens, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_contigs, n, ctx->d_index);
    LHGT_HIP(hipGetLastError());
    return LHGT_OK;
}

int hash_contig_to_device(lhgt_ctx* ctx, const uint8_t* ascii, long len, uint32_t* d_out, uint8_t* d_valid) {
    if (len < ctx->k) return LHGT_OK;
    LHGT_TRY(ws_reserve(ctx, (size_t)len + 32, 0));
    LHGT_HIP(hipMemcpyAsync(ctx->d_ws_ascii, ascii, (size_t)len, hipMemcpyHostToDevice, ctx->stream));
    return hash_contig_dev_ascii(ctx, ctx->d_ws_ascii, len, d_out, d_valid);
}

// RAII hipHostRegister of a host range (no-op if the runtime refuses, e.g. for some file mappings)
struct HostPin {
    void* p = nullptr;
    HostPin(const void* ptr, size_t bytes) {
        if (bytes >= (1u << 20) && hipHostRegister((void*)ptr, bytes, hipHostRegisterDefault) == hipSuccess) p = (void*)ptr;
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) hipHostUnregister(p); }
};

// Host range (typically a file mapping) -> device, in pieces of 128 MiB whose page-locking runs one piece AHEAD of the copies on a
// helper thread: locking costs about what the copy costs (~25 ms per GiB each; an unlocked range copies at a tenth of the rate,
// tools/h2d_rates.hip), so doing one after the other -- as the first version did, a GiB at a time -- doubled the upload.  Locked
// ranges are whole pages inside [src, src + bytes) and never overlap; the few bytes before the first and after the last page
// boundary travel as pageable copies.  A piece the runtime refuses to lock is copied pageable.  Returns after the last copy.
int upload_locked_ahead(lhgt_ctx* ctx, hipStream_t st, void* d_dst, const void* src, size_t bytes) {
    if (!bytes) return LHGT_OK;
    const uintptr_t PAGE = 4096, PIECE = (uintptr_t)128 << 20;
    const uintptr_t a = (uintptr_t)src, e = a + bytes, a_up = (a + PAGE - 1) & ~(PAGE - 1), e_dn = e & ~(PAGE - 1);
    std::vector<uintptr_t> cut(1, a);
    for (uintptr_t x = a_up + PIECE; x < e; x += PIECE) cut.push_back(x);
    cut.push_back(e);
    const size_t np = cut.size() - 1;
    std::vector<void*> locked(np, nullptr);
    auto range = [&](size_t i, uintptr_t* lo, uintptr_t* hi) {
        *lo = cut[i] > a_up ? cut[i] : a_up;
        *hi = cut[i + 1] < e_dn ? cut[i + 1] : e_dn;
    };
    auto lock = [&](size_t i) {
        uintptr_t lo, hi;
        range(i, &lo, &hi);
        if (hi <= lo || hi - lo < ((uintptr_t)1 << 20)) return;
        if (hipSetDevice(ctx->device) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipHostRegister((void*)lo, (size_t)(hi - lo), hipHostRegisterDefault) == hipSuccess) locked[i] = (void*)lo;
        else (void)hipGetLastError();
    };
    auto unlock = [&](size_t i) { if (locked[i]) { (void)hipHostUnregister(locked[i]); locked[i] = nullptr; } };
    auto copy = [&](uintptr_t from, uintptr_t to) -> hipError_t {
        return to > from ? hipMemcpyAsync((uint8_t*)d_dst + (from - a), (const void*)from, (size_t)(to - from), hipMemcpyHostToDevice, st) : hipSuccess;
    };
    std::thread helper;
    hipError_t err = hipSuccess;
    lock(0);
    for (size_t i = 0; i < np && err == hipSuccess; i++) {
        if (helper.joinable()) helper.join();
        helper = std::thread([&, i] { if (i > 0) unlock(i - 1); if (i + 1 < np) lock(i + 1); });
        uintptr_t lo, hi;
        range(i, &lo, &hi);
        if (locked[i]) {
            err = copy(cut[i], lo);
            if (err == hipSuccess) err = copy(lo, hi);
            if (err == hipSuccess) err = copy(hi, cut[i + 1]);
        } else err = copy(cut[i], cut[i + 1]);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
    }
    if (helper.joinable()) helper.join();
    for (size_t i = 0; i < np; i++) unlock(i);
    if (err != hipSuccess) LHGT_FAIL(LHGT_E_HIP, "upload: %s", hipGetErrorString(err));
    return LHGT_OK;
}

// ---------------------------------------------------------------- resident pairs
static void free_batch(ReadBatch& b) {
    for (void*& p : b.alloc) if (p) { lhgt::dev_free(p); p = nullptr; }
}

// Install n pairs whose ASCII bases already sit in device memory: sequence r (r < n: mate 1 of pair r, else
// mate 2 of pair r-n) is d_ascii[start[r] .. start[r]+len[r]).
int install_pairs_dev_ascii(lhgt_ctx* ctx, const uint8_t* d_ascii, const uint64_t* start, const uint16_t* lens, long n,
                            const uint8_t* pair_flags) {
    const int k = ctx->k;
    std::vector<uint64_t> word_off(2 * n);
    uint64_t words = 0, nkm = 0;
    int max_len = 0;
    long n_long = 0;
    for (long r = 0; r < 2 * n; r++) {
        uint64_t len = lens[r];
        if (len > LHGT_MAX_READ_LEN) LHGT_FAIL(LHGT_E_FORMAT, "read longer than %d bases", LHGT_MAX_READ_LEN);
        word_off[r] = words;
        words += 3 * ((len + 31) / 32 + 1);
        if ((int)len > max_len) max_len = (int)len;
        if ((long)len >= k) nkm += len - k + 1;
        if ((long)len - k + 1 > FAST_NK) n_long++;
    }
    if (words >= (1ull << 32)) LHGT_FAIL(LHGT_E_ARG, "batch too large: %llu plane words (split the append)", (unsigned long long)words);
    ReadBatch b;
    b.n_words = words;
    b.max_len = max_len;
    b.n_kmers = nkm;
    b.n_long = n_long;
    uint32_t *d_words, *d_off32;
    uint16_t* d_len;
    uint8_t* d_cnt = nullptr;
    uint64_t *d_start, *d_word_off;
    LHGT_HIP(lhgt::dev_malloc(&d_words, words * 4 + 16));
    b.alloc[0] = d_words;
    LHGT_HIP(lhgt::dev_malloc(&d_off32, (size_t)2 * n * 4));
    b.alloc[1] = d_off32;
    LHGT_HIP(lhgt::dev_malloc(&d_len, (size_t)2 * n * 2));
    b.alloc[2] = d_len;
    if (pair_flags) {
        LHGT_HIP(lhgt::dev_malloc(&d_cnt, (size_t)n));
        b.alloc[3] = d_cnt;
        LHGT_HIP(hipMemcpyAsync(d_cnt, pair_flags, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    LHGT_HIP(lhgt::dev_malloc(&d_start, (size_t)2 * n * 8));
    b.alloc[4] = d_start;
    LHGT_HIP(lhgt::dev_malloc(&d_word_off, (size_t)2 * n * 8));
    b.alloc[5] = d_word_off;
    LHGT_HIP(hipMemcpyAsync(d_start, start, (size_t)2 * n * 8, hipMemcpyHostToDevice, ctx->stream));
    LHGT_HIP(hipMemcpyAsync(d_word_off, word_off.data(), word_off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint32_t> off32(word_off.begin(), word_off.end());
    LHGT_HIP(hipMemcpyAsync(d_off32, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    LHGT_HIP(hipMemcpyAsync(d_len, lens, (size_t)2 * n * 2, hipMemcpyHostToDevice, ctx->stream));
    int max_wpr = (max_len + 31) / 32 + 1;
    long threads = 2 * n * max_wpr;
    hipLaunchKernelGGL(pack_bases, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, d_ascii,
                       d_start, d_len, 0L, d_word_off, 2 * n, max_wpr, d_words);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    lhgt::dev_free(d_start);
    lhgt::dev_free(d_word_off);
    b.alloc[4] = b.alloc[5] = nullptr;
    b.d.words = d_words;
    b.d.off[0] = d_off32;
    b.d.off[1] = d_off32 + n;
    b.d.len[0] = d_len;
    b.d.len[1] = d_len + n;
    b.d.flags = d_cnt;
    b.d.n_pairs = n;
    ctx->batches.push_back(b);
    ctx->store_gen++;
    ctx->n_pairs += n;
    return LHGT_OK;
}

// one thread per pair: its chunk by binary search over the descriptors, then chunk bases + the pair's record
__global__ void __launch_bounds__(256) expand_chunk_meta(const ChunkDesc* __restrict__ desc, int n_desc, const ChunkPairMeta* __restrict__ meta, long n,
                                                         uint32_t* __restrict__ start, uint32_t* __restrict__ woff, uint16_t* __restrict__ len,
                                                         uint8_t* __restrict__ flags) {
    const long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    int lo = 0, hi = n_desc;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((long)desc[mid].pair0 <= m) lo = mid; else hi = mid; }
    const ChunkDesc d = desc[lo];
    const ChunkPairMeta* r = meta + d.mo + (m - d.pair0);
    const ChunkPairMeta a = r[0], b = r[1];
    const bool il = d.b2 == CHUNK_INTERLEAVED;
    const uint32_t l1 = il ? a.rel2 - a.rel1 : b.rel1 - a.rel1, l2 = il ? b.rel1 - a.rel2 : b.rel2 - a.rel2;
    start[m] = d.b1 + a.rel1;
    start[n + m] = (il ? d.b1 : d.b2) + a.rel2;
    len[m] = (uint16_t)l1;
    len[n + m] = (uint16_t)l2;
    const uint32_t w1 = d.wbase + a.relw;
    woff[m] = w1;
    woff[n + m] = w1 + 3u * ((l1 + 31u) / 32u + 1u);
    flags[m] = (uint8_t)a.flags;
}

int install_pairs_chunked(lhgt_ctx* ctx, const uint8_t* d_ascii, const ChunkPairMeta* d_meta, const ChunkDesc* desc, long n_desc, long n,
                          uint64_t n_words, int max_len, uint64_t n_kmers, long n_long) {
    if (n <= 0) return LHGT_OK;
    if (n_words >= (1ull << 32)) LHGT_FAIL(LHGT_E_ARG, "batch too large: %llu plane words", (unsigned long long)n_words);
    ReadBatch b;
    b.n_words = n_words;
    b.max_len = max_len;
    b.n_kmers = n_kmers;
    b.n_long = n_long;
    // one allocation per batch: words | offsets | lengths | flags (the allocator call is not free, and a file has dozens of batches)
    const size_t words_b = (n_words * 4 + 16 + 255) & ~(size_t)255, off_b = ((size_t)2 * n * 4 + 255) & ~(size_t)255, len_b = ((size_t)2 * n * 2 + 255) & ~(size_t)255;
    uint8_t* blk = nullptr;
    LHGT_HIP(lhgt::dev_malloc(&blk, words_b + off_b + len_b + (size_t)n));
    b.alloc[0] = blk;
    uint32_t* d_words = (uint32_t*)blk;
    uint32_t* d_off32 = (uint32_t*)(blk + words_b);
    uint16_t* d_len = (uint16_t*)(blk + words_b + off_b);
    uint8_t* d_fl = blk + words_b + off_b + len_b;
    const long need = 2 * n + (n_desc * (long)sizeof(ChunkDesc) + 3) / 4;
    if (need > ctx->ingest_start_cap) {
        if (ctx->d_ingest_start) lhgt::dev_free(ctx->d_ingest_start);
        ctx->d_ingest_start = nullptr;
        ctx->ingest_start_cap = need + need / 4;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_ingest_start, (size_t)ctx->ingest_start_cap * 4));
    }
    hipStream_t st = ctx->stream;
    ChunkDesc* d_desc = (ChunkDesc*)(ctx->d_ingest_start + 2 * n);
    LHGT_HIP(hipMemcpyAsync(d_desc, desc, (size_t)n_desc * sizeof(ChunkDesc), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(expand_chunk_meta, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_desc, (int)n_desc, d_meta, n, ctx->d_ingest_start,
                       d_off32, d_len, d_fl);
    const int max_wpr = (max_len + 31) / 32 + 1;
    const long threads = 2 * n * max_wpr;
    hipLaunchKernelGGL(pack_bases32, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, d_ascii, ctx->d_ingest_start, d_len, d_off32,
                       2 * n, max_wpr, d_words);
    LHGT_HIP(hipGetLastError());
    b.d.words = d_words;
    b.d.off[0] = d_off32;
    b.d.off[1] = d_off32 + n;
    b.d.len[0] = d_len;
    b.d.len[1] = d_len + n;
    b.d.flags = d_fl;
    b.d.n_pairs = n;
    ctx->batches.push_back(b);
    ctx->store_gen++;
    ctx->n_pairs += n;
    return LHGT_OK;
}

void pairs_truncate(lhgt_ctx* ctx, size_t n_batches) {
    while (ctx->batches.size() > n_batches) {
        ctx->n_pairs -= ctx->batches.back().d.n_pairs;
        free_batch(ctx->batches.back());
        ctx->batches.pop_back();
        ctx->store_gen++;
    }
}

void ingest_free(lhgt_ctx* ctx) {
    for (hipEvent_t e : ctx->ingest_events) hipEventDestroy(e);
    ctx->ingest_events.clear();
    if (ctx->h_ingest_slabs) { hipHostFree(ctx->h_ingest_slabs); ctx->h_ingest_slabs = nullptr; }
    if (ctx->h_ingest_meta) { hipHostFree(ctx->h_ingest_meta); ctx->h_ingest_meta = nullptr; }
    if (ctx->d_ingest_start) { lhgt::dev_free(ctx->d_ingest_start); ctx->d_ingest_start = nullptr; }
    ctx->ingest_meta_cap = ctx->ingest_start_cap = 0;
}

// Copy a host range into the ASCII staging area at dev_off (pinned for the copy; the stream is drained before
// returning so the pin and the caller's buffer may go away).
int stage_ascii(lhgt_ctx* ctx, size_t dev_off, const uint8_t* src, size_t bytes) {
    if (!bytes) return LHGT_OK;
    if (dev_off + bytes > ctx->ws_ascii_cap) LHGT_FAIL(LHGT_E_STATE, "ASCII staging overflow (%zu + %zu > %zu)", dev_off, bytes, ctx->ws_ascii_cap);
    HostPin pin(src, bytes);
    LHGT_HIP(hipMemcpyAsync(ctx->d_ws_ascii + dev_off, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

int upload_pairs(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2, const uint64_t* off2,
                 long n, const uint8_t* pair_flags) {
    if (n <= 0) return LHGT_OK;
    size_t bytes1 = off1[n] - off1[0], bytes2 = off2[n] - off2[0];
    std::vector<uint64_t> start(2 * n);
    std::vector<uint16_t> lens(2 * n);
    for (long p = 0; p < n; p++) {
        uint64_t l1 = off1[p + 1] - off1[p], l2 = off2[p + 1] - off2[p];
        if (l1 > LHGT_MAX_READ_LEN || l2 > LHGT_MAX_READ_LEN) LHGT_FAIL(LHGT_E_FORMAT, "read longer than %d bases", LHGT_MAX_READ_LEN);
        start[p] = off1[p] - off1[0];
        start[n + p] = bytes1 + (off2[p] - off2[0]);
        lens[p] = (uint16_t)l1;
        lens[n + p] = (uint16_t)l2;
    }
    LHGT_TRY(ws_reserve(ctx, bytes1 + bytes2 + 32, 0));
    // first-touch pageable H2D runs at ~5.6 GB/s on this platform, a registered range at ~57 GB/s, and
    // registering costs ~25 ms/GiB (tools/h2d_rates.hip): stage_ascii pins each source range for its copy
    LHGT_TRY(stage_ascii(ctx, 0, seq1 + off1[0], bytes1));
    LHGT_TRY(stage_ascii(ctx, bytes1, seq2 + off2[0], bytes2));
    return install_pairs_dev_ascii(ctx, ctx->d_ws_ascii, start.data(), lens.data(), n, pair_flags);
}

// ---------------------------------------------------------------- index layout / install
int index_layout(lhgt_ctx* ctx, const std::vector<uint32_t>& lens, uint32_t first_ref_index) {
    const int k = ctx->k, e = ctx->e;
    slot_list_drop(ctx);
    for (void* p : {(void*)ctx->d_index, (void*)ctx->d_ref_planes, (void*)ctx->d_contigs, (void*)ctx->d_tiles, (void*)ctx->d_flags, (void*)ctx->d_nzmask, (void*)ctx->d_tile_good, (void*)ctx->d_active_tiles, (void*)ctx->d_tile_count, (void*)ctx->d_tile_sel, (void*)ctx->d_rg_buf})
        if (p) lhgt::dev_free(p);
    ctx->d_ref_planes = nullptr; ctx->ref_plane_words = 0;
    ctx->d_index = nullptr; ctx->d_contigs = nullptr; ctx->d_tiles = nullptr; ctx->d_flags = nullptr; ctx->d_nzmask = nullptr; ctx->d_tile_good = nullptr; ctx->d_active_tiles = nullptr; ctx->d_tile_count = nullptr; ctx->d_tile_sel = nullptr; ctx->d_rg_buf = nullptr; ctx->rg_buf_bytes = 0;
    ctx->contigs.clear();
    ctx->contig_first_tile.clear();
    ctx->all_lens.clear();
    std::vector<TileDev> tiles;
    uint64_t word = 0, flat = 0;
    uint32_t ref_index = first_ref_index;   // contig numbers stay the global ones when only a shard is resident
    for (uint32_t len : lens) {
        if ((long)len <= k) LHGT_FAIL(LHGT_E_FORMAT, "contig of length %u <= k in the index", len);
        ContigDev c;
        c.hash_word = word + 1;
        c.flat_base = flat;
        c.len = len;
        c.ref_index = ref_index++;
        ctx->contigs.push_back(c);
        ctx->contig_first_tile.push_back((long)tiles.size());
        for (uint32_t j0 = 0; j0 < len; j0 += TILE) tiles.push_back(TileDev{(uint32_t)(ctx->contigs.size() - 1), j0});
        word += 1 + (uint64_t)(len - k + 1) * e;
        flat += len;
    }
    ctx->index_words = word;
    ctx->n_pos = flat;
    ctx->n_tiles = (long)tiles.size();
    ctx->index_resident = true;
    if (ctx->contigs.empty()) return LHGT_OK;
    if (ctx->ref_packed) {
        ctx->ref_plane_words = (size_t)((flat + 31) / 32) + 2;    // a window is cut from two consecutive words
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_ref_planes, 3 * ctx->ref_plane_words * 4));
        LHGT_HIP(hipMemsetAsync(ctx->d_ref_planes, 0, 3 * ctx->ref_plane_words * 4, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_index, word * 4));
    }
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_contigs, ctx->contigs.size() * sizeof(ContigDev)));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_tiles, tiles.size() * sizeof(TileDev)));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_flags, flat));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_nzmask, flat));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_tile_good, tiles.size() + 8));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_active_tiles, (tiles.size() + 1) * 4));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_tile_sel, (tiles.size() + 16) * 4));
    LHGT_HIP(lhgt::dev_malloc(&ctx->d_tile_count, (tiles.size() + 16) * 4));  // counts, total, then small counters (selected positions, active tiles, saturated lines)
    LHGT_HIP(hipMemcpyAsync(ctx->d_contigs, ctx->contigs.data(), ctx->contigs.size() * sizeof(ContigDev), hipMemcpyHostToDevice, ctx->copy_stream));
    LHGT_HIP(hipMemcpyAsync(ctx->d_tiles, tiles.data(), tiles.size() * sizeof(TileDev), hipMemcpyHostToDevice, ctx->copy_stream));
    LHGT_HIP(hipStreamSynchronize(ctx->copy_stream));
    return LHGT_OK;
}

int index_install(lhgt_ctx* ctx, const uint32_t* w, size_t n_words, bool /*words_on_device*/) {
    return index_install_shard(ctx, w, n_words, 0, 1);
}

// Keep only shard `rank` of `world`: contiguous contig groups of about equal index bytes (the role of
// split_ref, E:1280-1330, but balanced exactly).  Contig numbers (ref_index) remain global.
int index_install_shard(lhgt_ctx* ctx, const uint32_t* w_all, size_t n_words_all, int rank, int world) {
    const int k = ctx->k, e = ctx->e;
    if (ctx->ref_packed)
        LHGT_FAIL(LHGT_E_STATE, "the packed reference form is filled from the bases (lhgt_reference_load_fasta), not from the hashes of an index file");
    std::vector<uint32_t> lens_all;
    std::vector<size_t> starts;
    size_t pos = 0;
    while (pos < n_words_all) {
        uint32_t len = w_all[pos];
        if ((long)len <= k) LHGT_FAIL(LHGT_E_FORMAT, "index: contig length %u <= k at word %zu", len, pos);
        size_t step = 1 + (size_t)(len - k + 1) * e;
        if (pos + step > n_words_all) LHGT_FAIL(LHGT_E_FORMAT, "index: truncated contig record at word %zu", pos);
        lens_all.push_back(len);
        starts.push_back(pos);
        pos += step;
    }
    starts.push_back(pos);
    const size_t nc = lens_all.size();
    size_t c0 = 0, c1 = nc;
    if (world > 1) {   // contig c belongs to the shard its first word falls in
        auto shard_of = [&](size_t c) { return (size_t)(((__uint128_t)starts[c] * world) / (n_words_all ? n_words_all : 1)); };
        c0 = 0;
        while (c0 < nc && shard_of(c0) < (size_t)rank) c0++;
        c1 = c0;
        while (c1 < nc && shard_of(c1) == (size_t)rank) c1++;
    }
    std::vector<uint32_t> lens(lens_all.begin() + c0, lens_all.begin() + c1);
    const uint32_t* w = w_all + starts[c0];
    const size_t n_words = starts[c1] - starts[c0];
    LHGT_TRY(index_layout(ctx, lens, (uint32_t)c0 + 1));
    if (lens.size() != lens_all.size()) ctx->all_lens = lens_all;
    // on its own non-blocking stream: a plain hipMemcpy would serialise with the FASTQ loader's stream (legacy default-stream rule)
    return upload_locked_ahead(ctx, ctx->copy_stream, ctx->d_index, w, n_words * 4);
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

int lhgt_pairs_append_flags(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2,
                            const uint64_t* off2, long n_pairs, const uint8_t* pair_flags) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !seq1 || !off1 || !seq2 || !off2 || n_pairs < 0) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    const long CH = 4L << 20;  // pairs per device batch (keeps 32-bit word offsets inside a batch)
    for (long p = 0; p < n_pairs; p += CH) {
        long n = n_pairs - p < CH ? n_pairs - p : CH;
        LHGT_TRY(upload_pairs(ctx, seq1, off1 + p, seq2, off2 + p, n, pair_flags ? pair_flags + p : nullptr));
    }
    return LHGT_OK;
}

int lhgt_pairs_append(lhgt_ctx* ctx, const uint8_t* seq1, const uint64_t* off1, const uint8_t* seq2,
                      const uint64_t* off2, long n_pairs, const uint8_t* count_mate2) {
    if (!count_mate2 || n_pairs <= 0) return lhgt_pairs_append_flags(ctx, seq1, off1, seq2, off2, n_pairs, nullptr);
    std::vector<uint8_t> fl((size_t)n_pairs);
    for (long p = 0; p < n_pairs; p++) fl[p] = (uint8_t)(PAIR_COUNT1 | PAIR_VOTE | (count_mate2[p] ? PAIR_COUNT2 : 0));
    return lhgt_pairs_append_flags(ctx, seq1, off1, seq2, off2, n_pairs, fl.data());
}

int lhgt_pairs_clear(lhgt_ctx* ctx) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    for (auto& b : ctx->batches) free_batch(b);
    ctx->batches.clear();
    ctx->n_pairs = 0;
    ctx->store_gen++;
    lhgt::vshared_free(ctx);      // what the shared-line-fill vote kept about this store's reads (15 GB per 100 M pairs; the blocks stay with the process)
    return LHGT_OK;
}

int lhgt_pairs_count(lhgt_ctx* ctx, long* n_pairs) {
    if (!ctx || !n_pairs) LHGT_FAIL(LHGT_E_ARG, "null argument");
    *n_pairs = ctx->n_pairs;
    return LHGT_OK;
}

int lhgt_hash_sequence(lhgt_ctx* ctx, const uint8_t* ascii, long len, uint32_t* out_hash, uint8_t* out_valid) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !ascii || !out_hash) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder set");
    long nk = len - ctx->k + 1;
    if (nk <= 0) return LHGT_OK;
    uint32_t* d_out;
    uint8_t* d_valid;
    LHGT_HIP(lhgt::dev_malloc(&d_out, (size_t)nk * ctx->e * 4));
    LHGT_HIP(lhgt::dev_malloc(&d_valid, (size_t)nk));
    int rc = hash_contig_to_device(ctx, ascii, len, d_out, d_valid);
    if (rc == LHGT_OK) {
        hipMemcpyAsync(out_hash, d_out, (size_t)nk * ctx->e * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (out_valid) hipMemcpyAsync(out_valid, d_valid, (size_t)nk, hipMemcpyDeviceToHost, ctx->stream);
        hipStreamSynchronize(ctx->stream);
    }
    lhgt::dev_free(d_out);
    lhgt::dev_free(d_valid);
    return rc;
}

int lhgt_index_from_memory(lhgt_ctx* ctx, const uint8_t* ascii, const uint64_t* off, long n_contigs) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !ascii || !off || n_contigs < 0) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (!ctx->have_coder) LHGT_FAIL(LHGT_E_STATE, "no coder set");
    std::vector<uint32_t> lens;
    std::vector<long> src;
    for (long c = 0; c < n_contigs; c++) {
        uint64_t len = off[c + 1] - off[c];
        if ((long)len > ctx->k) { lens.push_back((uint32_t)len); src.push_back(c); }  // E:772: shorter contigs are not indexed
    }
    LHGT_TRY(index_layout(ctx, lens));
    LHGT_TRY(write_index_lens(ctx));
    // spans of consecutive source contigs (short ones included and skipped), ~256 MB of bases per upload and launch
    const uint64_t SPAN = 256ull << 20;
    size_t ci = 0;
    for (long p = 0; p < n_contigs;) {
        long q = p;
        while (q < n_contigs && (q == p || off[q + 1] - off[p] <= SPAN)) q++;
        const uint64_t span_len = off[q] - off[p];
        std::vector<uint64_t> coff((size_t)(q - p) + 1);
        std::vector<long> contig_of((size_t)(q - p));
        for (long r = p; r <= q; r++) coff[r - p] = off[r] - off[p];
        for (long r = p; r < q; r++) contig_of[r - p] = (long)(off[r + 1] - off[r]) > ctx->k ? (long)ci++ : -1L;
        if (span_len) {
            LHGT_TRY(ws_reserve(ctx, (size_t)span_len + 32, 0));
            LHGT_TRY(stage_ascii(ctx, 0, ascii + off[p], (size_t)span_len));
            LHGT_TRY(install_span_dev_ascii(ctx, ctx->d_ws_ascii, (long)span_len, coff.data(), contig_of.data(), q - p));
        }
        p = q;
    }
    return LHGT_OK;
}

}  // extern "C"
