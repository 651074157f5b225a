// k_vote.hip -- phase C: re-scan of the resident pairs against peak_kmer and the split-read
// vote (Peaks::slide_reads E:313-506, Split_reads::judge_base E:118-159, check_split E:161-202).
//
// One wave per pair.  The e probes of every k-mer offset (mate 1 then mate 2, E:430-495) are
// done by all lanes; offsets where some probe hits a peak are compacted, in offset order, into
// an LDS event list.  Only pairs with >= 6 such offsets (MIN_BASE_NUM, E:29,496) run the
// order-dependent judge_base logic, sequentially on lane 0 over the few events.
#include <algorithm>
#include "lhgt_hash.hpp"

namespace lhgt {

// Per-wave LDS: events[max_ev][e] of (peak id, contig); the contig of a hit is fetched by the lane
// that found it.  judge_base then runs out of registers: lane l holds events l, l+64, .. of the
// current 64-event chunk and entries l, l+64, .. of the contig table (TR registers deep); an event
// is broadcast with readlane, the table searched with one compare + ballot per register row.
// EC = compile-time number of hashes (3), or 0 for the generic runtime-e form: the judge is bound by
// instruction issue (mostly scalar control flow), so dead iterations and row walks are compiled away.
template <int TR, int EC>
__device__ __forceinline__ void judge_pair(const uint32_t* ev, int n_ev, int e_rt, int lane, uint32_t* __restrict__ filter) {
    constexpr int EM = EC ? EC : 9;
    const int e = EC ? EC : e_rt;
    int tchr[TR], tcnt[TR], tfirst[TR];
#pragma unroll
    for (int r = 0; r < TR; r++) { tchr[r] = 0; tcnt[r] = 0; tfirst[r] = 0; }
    int n_tab = 0;
    for (int q0 = 0; q0 < n_ev; q0 += 64) {
        uint32_t eid[EM], echr[EM];
        const int myq = q0 + lane;
#pragma unroll
        for (int i = 0; i < EM; i++)
            if (i < e) {
                eid[i] = myq < n_ev ? ev[((size_t)myq * e + i) * 2] : 0u;
                echr[i] = myq < n_ev ? ev[((size_t)myq * e + i) * 2 + 1] : 0u;
            }
        const int nq = n_ev - q0 < 64 ? n_ev - q0 : 64;
        for (int qq = 0; qq < nq; qq++) {
            const int q = __builtin_amdgcn_readfirstlane(qq);
            int sel_chr = 0, sel_id = 0, sel_num = 0, sel_slot = -1;
            int last_chr = -1, s = -1, cnt = 0;   // lookup of the previous hash of this event (counts do not move inside an event)
#pragma unroll
            for (int i = 0; i < EM; i++) {
                if (!EC && i >= e) continue;
                const int id = __builtin_amdgcn_readlane((int)eid[i], q);
                if (!id) continue;
                const int chr = __builtin_amdgcn_readlane((int)echr[i], q);
                if (chr != last_chr) {
                    last_chr = chr;
                    s = -1;
                    cnt = 0;
                    {   // row 0: the whole table while it has <= 64 entries (the usual case)
                        const unsigned long long bal = __ballot(lane < n_tab && tchr[0] == chr);
                        if (bal) {
                            s = __ffsll((long long)bal) - 1;
                            cnt = __builtin_amdgcn_readlane(tcnt[0], s);
                        }
                    }
                    if (TR > 1 && s < 0 && n_tab > 64) {
#pragma unroll
                        for (int r = 1; r < TR; r++) {
                            if (s >= 0 || r * 64 >= n_tab) continue;
                            const unsigned long long bal = __ballot(r * 64 + lane < n_tab && tchr[r] == chr);
                            if (bal) {
                                const int l = __ffsll((long long)bal) - 1;
                                s = r * 64 + l;
                                cnt = __builtin_amdgcn_readlane(tcnt[r], l);
                            }
                        }
                    }
                }
                // among the hashes that hit, prefer the contig with the largest running count (ties: later
                // hash, `>=` at E:131); an unseen contig is taken only if nothing is selected yet (E:140-144)
                if (s >= 0) {
                    if (cnt >= sel_num) { sel_id = id; sel_chr = chr; sel_num = cnt; sel_slot = s; }
                } else if (sel_id == 0) { sel_id = id; sel_chr = chr; sel_num = 0; sel_slot = -1; }
            }
            const int slot = sel_slot >= 0 ? sel_slot : n_tab;
            const bool mine = lane == (slot & 63);
#pragma unroll
            for (int r = 0; r < TR; r++)
                if ((slot >> 6) == r) {                 // wave-uniform: only the owning row is touched
                    if (sel_slot >= 0) tcnt[r] = mine ? sel_num + 1 : tcnt[r];
                    else {                              // first peak of the contig (E:150-152)
                        tchr[r] = mine ? sel_chr : tchr[r];
                        tcnt[r] = mine ? 1 : tcnt[r];
                        tfirst[r] = mine ? sel_id : tfirst[r];
                    }
                }
            if (sel_slot < 0) n_tab++;
        }
    }
    // check_split: contigs with >= 6 offsets; the two largest counts (with multiplicity) vote (E:161-202)
    int largest = 0, n_f = 0;
#pragma unroll
    for (int r = 0; r < TR; r++) {
        const int c = (r * 64 + lane < n_tab && tcnt[r] >= 6) ? tcnt[r] : 0;
        n_f += __popcll(__ballot(c > 0));
        largest = c > largest ? c : largest;
    }
    for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(largest, d); largest = o > largest ? o : largest; }
    if (n_f > 1) {
        int n_at = 0, second = 0;
#pragma unroll
        for (int r = 0; r < TR; r++) {
            const int c = (r * 64 + lane < n_tab && tcnt[r] >= 6) ? tcnt[r] : 0;
            n_at += __popcll(__ballot(c == largest));
            second = (c < largest && c > second) ? c : second;
        }
        for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(second, d); second = o > second ? o : second; }
        if (n_at > 1) second = largest;
#pragma unroll
        for (int r = 0; r < TR; r++) {
            const int c = (r * 64 + lane < n_tab) ? tcnt[r] : 0;
            if (c >= 6 && (c == largest || c == second)) atomicAdd(&filter[tfirst[r]], 1u);  // clamped to 254 at export (E:194)
        }
    }
}

// Any read length up to 500, probes of one 64-offset slice at a time.  PF: consult the L2-resident folded bitmap
// first (exact negatives: a clear bit means no slot folding onto it holds a peak), so sparse peak sets never
// touch the 16 GiB peak_kmer array except for true hits and the few false positives.
template <int TR, bool PF, bool NT>
__global__ void __launch_bounds__(256) vote_kernel(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                   const uint32_t* __restrict__ prefilter,
                                                   const int32_t* __restrict__ loci, uint32_t* __restrict__ filter,
                                                   int max_ev, int waves_per_block, int debug, uint32_t pf_mask) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    if (wib >= waves_per_block) return;
    const int e = hp.e, k = hp.k;
    uint32_t* ev = lds + (size_t)wib * max_ev * e * 2;
    const long wave = (long)blockIdx.x * waves_per_block + wib;
    const long n_waves = (long)gridDim.x * waves_per_block;
    for (long p = wave; p < b.n_pairs; p += n_waves) {
        int n_ev = 0;
        for (int m = 0; m < 2; m++) {
            const int len = b.len[m][p];
            const int nk = len - k + 1;
            if (nk <= 0) continue;
            const int wpr = ((len + 31) >> 5) + 1;
            const uint32_t* rec = b.words + b.off[m][p];
            for (int j0 = 0; j0 < nk; j0 += 64) {
                const int j = j0 + lane;
                uint32_t ids[9], chrs[9];
                bool hit = false;
                if (j < nk && plane_window(rec + 2 * wpr, j, k) == 0) {
                    uint32_t whi = plane_window(rec, j, k), wlo = plane_window(rec + wpr, j, k);
                    uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
#pragma unroll
                    for (int i = 0; i < 9; i++)
                        if (i < e) {
                            const uint32_t h = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
                            if (PF) {
                                const uint32_t fb = h & pf_mask;
                                ids[i] = ((prefilter[fb >> 5] >> (fb & 31u)) & 1u) ? peak_kmer[h] : 0u;
                            } else if (NT) {   // tables of 1 GiB and more (k >= 28): nothing to keep in the caches
                                // `nt`: +11 % probe rate on a table far beyond the caches (profiles/r01_probe_policy_microbench.txt)
                                ids[i] = __builtin_nontemporal_load(peak_kmer + h);
                            } else ids[i] = peak_kmer[h];  // 0 = no peak (E:454)
                            hit |= ids[i] != 0;
                        }
#pragma unroll
                    for (int i = 0; i < 9; i++)
                        if (i < e) chrs[i] = ids[i] ? (uint32_t)loci[2 * (long)ids[i]] : 0u;  // count_peak_kmer's peak_chr (E:455)
                }
                unsigned long long bal = __ballot(hit);
                if (bal) {
                    if (hit) {
                        int slot = n_ev + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
                        for (int i = 0; i < 9; i++)
                            if (i < e) {
                                ev[((size_t)slot * e + i) * 2] = ids[i];
                                ev[((size_t)slot * e + i) * 2 + 1] = chrs[i];
                            }
                    }
                    n_ev += __popcll(bal);
                }
            }
        }
        if (n_ev < 6 || (debug & 1)) continue;  // base_hits = offsets with any hit (E:149-157, 496)
        __builtin_amdgcn_wave_barrier();
        if (e == 3) judge_pair<TR, 3>(ev, n_ev, e, lane, filter);
        else judge_pair<TR, 0>(ev, n_ev, e, lane, filter);
        __builtin_amdgcn_wave_barrier();
    }
}

// phase D helper: peaks with at least MIN_READS (1, E:37) votes, as (id, contig, pos) in any order;
// the host sorts the few survivors by id, which is the order count_filtered_peak walks them (E:525).
__global__ void __launch_bounds__(256) compact_voted(const uint32_t* __restrict__ filter, const int32_t* __restrict__ loci, long n,
                                                     unsigned long long* __restrict__ counter, int32_t* __restrict__ out, long cap) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || filter[i] < 1) return;
    unsigned long long slot = atomicAdd(counter, 1ull);
    if ((long)slot < cap) {
        out[3 * slot] = (int32_t)i;
        out[3 * slot + 1] = loci[2 * i];
        out[3 * slot + 2] = loci[2 * i + 1];
    }
}

}  // namespace lhgt

using namespace lhgt;

extern "C" {

int lhgt_vote(lhgt_ctx* ctx) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "lhgt_ref_scan must precede lhgt_vote");
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    for (const ReadBatch& b : ctx->batches) {
        int nk = b.max_len - ctx->k + 1;
        if (nk <= 0) continue;
        int max_ev = 2 * nk;
        size_t per_wave = (size_t)max_ev * ctx->e * 2 * 4;
        int wpb = (int)(65536 / per_wave);
        if (wpb > 4) wpb = 4;
        if (wpb < 1) wpb = 1;
        long blocks = (b.d.n_pairs + wpb - 1) / wpb;
        if (blocks > 256L * 16) blocks = 256L * 16;
#define LHGT_VOTE(TR_, PF_, NT_)                                                                                          \
    hipLaunchKernelGGL((vote_kernel<TR_, PF_, NT_>), dim3((unsigned)blocks), dim3(64 * wpb), per_wave * wpb, ctx->stream, b.d, \
                       ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter, ctx->d_loci, ctx->d_filter, max_ev, wpb, ctx->debug, ctx->pf_mask)
        const bool nt = ctx->k >= 28;
        if (max_ev <= 256) {
            if (ctx->prefilter_on) LHGT_VOTE(4, true, false);
            else if (nt) LHGT_VOTE(4, false, true);
            else LHGT_VOTE(4, false, false);
        } else {
            if (ctx->prefilter_on) LHGT_VOTE(16, true, false);
            else if (nt) LHGT_VOTE(16, false, true);
            else LHGT_VOTE(16, false, false);
        }
#undef LHGT_VOTE
    }
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    LHGT_HIP(hipEventSynchronize(ctx->ev1));
    LHGT_HIP(hipEventElapsedTime(&ctx->phase_ms[2], ctx->ev0, ctx->ev1));
    ctx->voted = true;
    return LHGT_OK;
}

int lhgt_filter_buffer(lhgt_ctx* ctx, void** dev_ptr, size_t* bytes) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx || !dev_ptr || !bytes) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    *dev_ptr = ctx->d_filter;
    *bytes = (size_t)ctx->n_peaks * 4;
    return LHGT_OK;
}

int lhgt_peaks_export(lhgt_ctx* ctx, int32_t* loci, uint8_t* filter, long n) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    if (n > ctx->n_peaks) LHGT_FAIL(LHGT_E_ARG, "asked for %ld peaks, have %ld", n, ctx->n_peaks);
    if (n == 0) return LHGT_OK;
    if (loci) LHGT_HIP(hipMemcpy(loci, ctx->d_loci, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (filter) {
        std::vector<uint32_t> v((size_t)n);
        LHGT_HIP(hipMemcpy(v.data(), ctx->d_filter, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (long i = 0; i < n; i++) filter[i] = (uint8_t)(v[i] > 254 ? 254 : v[i]);  // `if (< 254) ++` saturates at 254
    }
    return LHGT_OK;
}

// count_filtered_peak (E:515-548), single thread range: leading sentinel "1 1 1", merge while the
// contig is the same and the gap to the running end is < 500.  Only the voted peaks leave the GPU.
int lhgt_write_intervals(lhgt_ctx* ctx, const char* path, long* n_filtered) {
    if (ctx && ctx->device < 0) LHGT_FAIL(LHGT_E_NO_DEVICE, "host-only context: no GPU work possible (no CPU fallback)");
    if (!ctx || !path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    const long n = ctx->n_peaks;
    std::vector<int32_t> rec;
    long nf = 0;
    if (n > 0) {
        long cap = ctx->voted_cap;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (!ctx->d_voted) {
                if (cap < 4096) cap = 4096;
                LHGT_HIP(hipMalloc(&ctx->d_voted, (size_t)cap * 12 + 8));
                ctx->voted_cap = cap;
            }
            unsigned long long* d_cnt = (unsigned long long*)((char*)ctx->d_voted + (size_t)ctx->voted_cap * 12);
            LHGT_HIP(hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
            hipLaunchKernelGGL(compact_voted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_filter, ctx->d_loci, n,
                               d_cnt, (int32_t*)ctx->d_voted, ctx->voted_cap);
            unsigned long long cnt = 0;
            LHGT_HIP(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            nf = (long)cnt;
            if (nf <= ctx->voted_cap) break;
            hipFree(ctx->d_voted);      // more voted peaks than room: grow once and redo
            ctx->d_voted = nullptr;
            cap = nf + nf / 8;
        }
        rec.resize((size_t)nf * 3);
        if (nf) LHGT_HIP(hipMemcpy(rec.data(), ctx->d_voted, (size_t)nf * 12, hipMemcpyDeviceToHost));
    }
    std::vector<long> order((size_t)nf);
    for (long i = 0; i < nf; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](long a, long b) { return rec[3 * a] < rec[3 * b]; });
    FILE* f = fopen(path, "w");
    if (!f) LHGT_FAIL(LHGT_E_IO, "cannot write %s", path);
    int start = 1, end = 1, chr = 1;
    for (long q = 0; q < nf; q++) {
        const int c = rec[3 * order[q] + 1], pos = rec[3 * order[q] + 2];
        if (chr == c && pos - 500 - end < 500) end = pos + 500;
        else {
            fprintf(f, "%d\t%d\t%d\n", chr, start, end);
            chr = c; start = pos - 500; end = pos + 500;
        }
    }
    fprintf(f, "%d\t%d\t%d\n", chr, start, end);
    fclose(f);
    if (n_filtered) *n_filtered = nf;
    return LHGT_OK;
}

}  // extern "C"
