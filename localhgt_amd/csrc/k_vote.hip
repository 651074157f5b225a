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

constexpr int LF_BITS = 19;                 // LDS-resident fold of the vote prefilter: 2^19 bits = 64 KiB
constexpr int LF_WORDS = (1 << LF_BITS) / 32;

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
template <int TR, int PF, bool NT>
__global__ void __launch_bounds__(256) vote_kernel(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                   const uint32_t* __restrict__ prefilter, const uint32_t* __restrict__ lds_fold,
                                                   const int32_t* __restrict__ loci, uint32_t* __restrict__ filter,
                                                   int max_ev, int waves_per_block, int debug, uint32_t pf_mask) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int e = hp.e, k = hp.k;
    if (wib >= waves_per_block) return;
    uint32_t* ev = lds + (size_t)wib * (max_ev * e * 2 + 64);
    uint32_t* stage = ev + (size_t)max_ev * e * 2;   // 64 words: the current read's record, staged once per read
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
            __builtin_amdgcn_wave_barrier();
            if (lane < 3 * wpr) stage[lane] = rec[lane];     // <= 51 words for 500 bases, one coalesced load
            __builtin_amdgcn_wave_barrier();
            for (int j0 = 0; j0 < nk; j0 += 64) {
                const int j = j0 + lane;
                uint32_t ids[9], chrs[9];
                bool hit = false;
                if (j < nk && plane_window(stage + 2 * wpr, j, k) == 0) {
                    uint32_t whi = plane_window(stage, j, k), wlo = plane_window(stage + wpr, j, k);
                    uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
#pragma unroll
                    for (int i = 0; i < 9; i++)
                        if (i < e) {
                            const uint32_t h = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
                            if (PF == 1) {
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

// Sparse-path form (prefilter on, <= 128 k-mer offsets per mate, e <= 3).  The generic kernel walks a pair slice by slice
// and every slice costs three dependent round trips; when the probes are cache hits that chain, not bandwidth, is the cost.
// Here the pair's four slices are handled together: all window words in flight at once, then all 12 filter probes at once,
// then the rare peak_kmer / contig loads.
template <int PF>
__global__ void __launch_bounds__(PF == 2 ? 1024 : 256) vote_kernel_sparse(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                                           const uint32_t* __restrict__ prefilter, const uint32_t* __restrict__ lds_fold,
                                                                           const int32_t* __restrict__ loci, uint32_t* __restrict__ filter,
                                                                           int max_ev, int waves_per_block, int debug, uint32_t pf_mask) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int e = hp.e, k = hp.k;
    const uint32_t* lfilter = lds;
    if (PF == 2) {
        for (int i = threadIdx.x; i < LF_WORDS; i += blockDim.x) lds[i] = lds_fold[i];
        __syncthreads();
    }
    if (wib >= waves_per_block) return;
    uint32_t* ev = lds + (PF == 2 ? LF_WORDS : 0) + (size_t)wib * (max_ev * e * 2 + 64);   // same per-wave stride as the generic kernel
    const long wave = (long)blockIdx.x * waves_per_block + wib;
    const long n_waves = (long)gridDim.x * waves_per_block;
    for (long p = wave; p < b.n_pairs; p += n_waves) {
        int nk[2], wpr[2];
        const uint32_t* rec[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int len = b.len[m][p];
            nk[m] = len - k + 1;
            wpr[m] = ((len + 31) >> 5) + 1;
            rec[m] = b.words + b.off[m][p];
        }
        // LDS-staged windows: the wave fetches each read's record (3 planes x wpr words, <= 18 words for 159 bases) with one
        // coalesced load and every lane cuts its windows out of LDS -- 2 vector-memory instructions per pair instead of 24.
        // (Prefetching the next pair's metadata and words one iteration ahead was tried and was slower.)
        uint32_t* stage = ev;   // the event area is free until the compaction below
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int nw = 3 * wpr[m];
            if (lane < nw) stage[m * 32 + lane] = rec[m][lane];
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t w[4][6];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int m = s >> 1, j = (s & 1) * 64 + lane;
            const uint32_t* q = stage + m * 32 + (j < nk[m] ? (j >> 5) : 0);
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                w[s][2 * pl] = q[pl * wpr[m]];
                w[s][2 * pl + 1] = q[pl * wpr[m] + 1];
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t hs[4][3], ids[4][3];
        bool ok[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int m = s >> 1, j = (s & 1) * 64 + lane, r = j & 31;
            auto win = [&](uint32_t a, uint32_t c) { return (uint32_t)(((((uint64_t)a << 32) | c) << r) >> 32) >> (32 - k); };
            const uint32_t whi = win(w[s][0], w[s][1]), wlo = win(w[s][2], w[s][3]), wnb = win(w[s][4], w[s][5]);
            const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
            ok[s] = j < nk[m] && wnb == 0;
#pragma unroll
            for (int i = 0; i < 3; i++) hs[s][i] = i < e ? hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]) : 0u;
        }
        // first filter level for all 12 hashes, then the second, then the table itself: each level only for survivors
        uint32_t f1[4][3];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const uint32_t h = hs[s][i];
                if (PF == 2) { const uint32_t lb = h & ((1u << LF_BITS) - 1u); f1[s][i] = (lfilter[lb >> 5] >> (lb & 31u)) & 1u; }
                else { const uint32_t fb = h & pf_mask; f1[s][i] = (prefilter[fb >> 5] >> (fb & 31u)) & 1u; }
                if (!(ok[s] && i < e)) f1[s][i] = 0u;
            }
        if (PF == 2) {
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (f1[s][i]) { const uint32_t fb = hs[s][i] & pf_mask; f1[s][i] = (prefilter[fb >> 5] >> (fb & 31u)) & 1u; }
        }
        bool any = false;
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int i = 0; i < 3; i++) {
                ids[s][i] = f1[s][i] ? peak_kmer[hs[s][i]] : 0u;   // 0 = no peak (E:454)
                any |= ids[s][i] != 0u;
            }
        if (!__ballot(any)) continue;   // the usual case on the sparse path: no lane of the pair hit anything
        int n_ev = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {   // slices in offset order: mate 1 (0..63, 64..127), mate 2 (E:430-495)
            bool hit = false;
#pragma unroll
            for (int i = 0; i < 3; i++) hit |= ids[s][i] != 0u;
            const unsigned long long bal = __ballot(hit);
            if (bal) {
                if (hit) {
                    const int slot = n_ev + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        if (i < e) {
                            ev[((size_t)slot * e + i) * 2] = ids[s][i];
                            ev[((size_t)slot * e + i) * 2 + 1] = ids[s][i] ? (uint32_t)loci[2 * (long)ids[s][i]] : 0u;
                        }
                }
                n_ev += __popcll(bal);
            }
        }
        if (n_ev < 6 || (debug & 1)) continue;
        __builtin_amdgcn_wave_barrier();
        if (e == 3) judge_pair<4, 3>(ev, n_ev, e, lane, filter);
        else judge_pair<4, 0>(ev, n_ev, e, lane, filter);
        __builtin_amdgcn_wave_barrier();
    }
}

// 64 KiB fold of the 2^PF_BITS-bit bitmap: word w = OR of the bitmap words w, w + LF_WORDS, ... (same low address bits)
__global__ void __launch_bounds__(256) fold_prefilter(const uint32_t* __restrict__ prefilter, int words, uint32_t* __restrict__ fold) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= LF_WORDS) return;
    uint32_t acc = 0;
    for (int j = w; j < words; j += LF_WORDS) acc |= prefilter[j];
    fold[w] = acc;
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
        size_t per_wave = ((size_t)max_ev * ctx->e * 2 + 64) * 4;   // events + 64 staging words (the sparse kernel stages inside the event area)
        int wpb = (int)(65536 / per_wave);
        if (wpb > 4) wpb = 4;
        if (wpb < 1) wpb = 1;
        long blocks = (b.d.n_pairs + wpb - 1) / wpb;
        if (blocks > 256L * 16) blocks = 256L * 16;
#define LHGT_VOTE(TR_, PF_, NT_, THREADS_, LDS_)                                                                           \
    hipLaunchKernelGGL((vote_kernel<TR_, PF_, NT_>), dim3((unsigned)blocks), dim3(THREADS_), LDS_, ctx->stream, b.d, ctx->hp,  \
                       ctx->d_peak_kmer, ctx->d_prefilter, ctx->d_prefilter_fold, ctx->d_loci, ctx->d_filter, max_ev, wpb,    \
                       ctx->debug, ctx->pf_mask)
        const bool nt = ctx->k >= 28;
        // sparse peak sets on a folded (k > PF_BITS) bitmap: 16-wave workgroups that keep a 64 KiB fold of it in LDS
        const size_t lds2 = (size_t)LF_WORDS * 4 + 16 * per_wave;
        const bool sparse_ok = ctx->prefilter_on && nk <= 128 && ctx->e <= 3 && !(ctx->debug & 32);
#define LHGT_VOTE_SPARSE(PF_, THREADS_, LDS_)                                                                                  \
    hipLaunchKernelGGL((vote_kernel_sparse<PF_>), dim3((unsigned)blocks), dim3(THREADS_), LDS_, ctx->stream, b.d, ctx->hp,     \
                       ctx->d_peak_kmer, ctx->d_prefilter, ctx->d_prefilter_fold, ctx->d_loci, ctx->d_filter, max_ev, wpb,    \
                       ctx->debug, ctx->pf_mask)
        if (sparse_ok && ctx->k > PF_BITS && lds2 <= 160 * 1024 && !(ctx->debug & 16)) {   // LDS first level (495 vs 530 ms on configs[2])
            wpb = 16;
            blocks = (b.d.n_pairs + wpb - 1) / wpb;
            if (blocks > 256) blocks = 256;     // one resident workgroup per CU
            static bool attr_set = false;
            if (!attr_set) {
                LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel_sparse<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set = true;
            }
            hipLaunchKernelGGL(fold_prefilter, dim3(LF_WORDS / 256), dim3(256), 0, ctx->stream, ctx->d_prefilter, (int)((ctx->pf_mask + 1ull) / 32), ctx->d_prefilter_fold);
            LHGT_VOTE_SPARSE(2, 1024, lds2);
        } else if (sparse_ok) {
            LHGT_VOTE_SPARSE(1, 64 * wpb, per_wave * wpb);
        } else if (max_ev <= 256) {
            if (ctx->prefilter_on) LHGT_VOTE(4, 1, false, 64 * wpb, per_wave * wpb);
            else if (nt) LHGT_VOTE(4, 0, true, 64 * wpb, per_wave * wpb);
            else LHGT_VOTE(4, 0, false, 64 * wpb, per_wave * wpb);
        } else {
            if (ctx->prefilter_on) LHGT_VOTE(16, 1, false, 64 * wpb, per_wave * wpb);
            else if (nt) LHGT_VOTE(16, 0, true, 64 * wpb, per_wave * wpb);
            else LHGT_VOTE(16, 0, false, 64 * wpb, per_wave * wpb);
        }
#undef LHGT_VOTE
#undef LHGT_VOTE_SPARSE
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
