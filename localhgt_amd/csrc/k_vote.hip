// k_vote.hip -- phase C: re-scan of the resident pairs against peak_kmer and the split-read
// vote (Peaks::slide_reads E:313-506, Split_reads::judge_base E:118-159, check_split E:161-202).
//
// One wave per pair.  The e probes of every k-mer offset (mate 1 then mate 2, E:430-495) are
// done by all lanes; offsets where some probe hits a peak are compacted, in offset order, into
// an LDS event list.  Only pairs with >= 6 such offsets (MIN_BASE_NUM, E:29,496) run the
// order-dependent judge_base logic, sequentially on lane 0 over the few events.
#include <algorithm>
#include "lhgt_hash.hpp"
#include "k_vote_judge.hpp"

namespace lhgt {

// Any read length up to 500, probes of one 64-offset slice at a time.  PF: consult the L2-resident folded bitmap
// first (exact negatives: a clear bit means no slot folding onto it holds a peak), so sparse peak sets never
// touch the 16 GiB peak_kmer array except for true hits and the few false positives.
template <int TR, int PF, bool NT>
__global__ void __launch_bounds__(256) vote_kernel(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                   const uint32_t* __restrict__ prefilter, const uint32_t* __restrict__ lds_fold,
                                                   const int32_t* __restrict__ loci, uint32_t* __restrict__ filter,
                                                   int max_ev, int waves_per_block, int debug, uint32_t pf_mask, int pf2,
                                                   const uint32_t* __restrict__ pair_list, long list_base,
                                                   const uint32_t* __restrict__ groups /* PF == 0, nullable: first peak id of VG_N runs of whole contigs (k_scan.hip: vote_group_bounds) */) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int e = hp.e, k = hp.k;
    if (wib >= waves_per_block) return;
    // 1024 16-bit counters per wave for the bound below: long event lists (k <= 23, long reads), and (round 6) every dense form -- without a
    // bitmap in front nearly every offset of a read is an event
    constexpr int BOUND_WORDS = (TR >= 8 || PF == 0) ? 512 : 0;
    uint32_t* ev = lds + (size_t)wib * (max_ev * e * 2 + 64 + BOUND_WORDS);
    uint32_t* stage = ev + (size_t)max_ev * e * 2;   // 64 words: the current read's record, staged once per read
    // the group table behind the waves' regions (the workgroup has exactly waves_per_block waves: every one of them gets here)
    const bool by_groups = PF == 0 && groups != nullptr;
    uint32_t* gtab = lds + (size_t)waves_per_block * (max_ev * e * 2 + 64 + BOUND_WORDS);
    if (by_groups) {
        for (int i = threadIdx.x; i < VG_N; i += 64 * waves_per_block) gtab[i] = groups[i];
        __syncthreads();
    }
    // group of a peak id: the last j with gtab[j] <= id (gtab[0] = 0)
    auto group_of = [&](uint32_t id) {
        int g = 0;
#pragma unroll
        for (int step = VG_N / 2; step >= 1; step >>= 1)
            if (gtab[g + step] <= id) g += step;
        return g;
    };
    const long wave = (long)blockIdx.x * waves_per_block + wib;
    const long n_waves = (long)gridDim.x * waves_per_block;
    // pair_list (vote_kernel_fold's deferred pairs): [0] = how many, then the pairs; null = every pair of the batch
    const long n_items = pair_list ? (long)pair_list[0] : b.n_pairs;
    for (long it = wave; it < n_items; it += n_waves) {
        // (list_base: a list of GLOBAL pair numbers over all resident batches -- k_vote_shared.hip -- holds other batches' pairs too)
        const long p = pair_list ? (long)pair_list[1 + it] - list_base : it;
        if (p < 0 || p >= b.n_pairs) continue;
        if (b.flags && !(b.flags[p] & PAIR_VOTE)) continue;   // counted only (surplus fq2 record, thread-chunk emulation)
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
                                ids[i] = pf_pass(prefilter[pf_word(h, pf_mask)], h, pf2) ? peak_kmer[h] : 0u;
                            } else if (NT) {   // tables of 1 GiB and more (k >= 28): nothing to keep in the caches
                                // `nt`: +11 % probe rate on a table far beyond the caches (profiles/r01_probe_policy_microbench.txt)
                                ids[i] = __builtin_nontemporal_load(peak_kmer + h);
                            } else ids[i] = peak_kmer[h];  // 0 = no peak (E:454)
                            hit |= ids[i] != 0;
                        }
                    // (by groups: the contigs are fetched below, and only where the bound over the groups asks for them)
                    if (!by_groups) {
#pragma unroll
                        for (int i = 0; i < 9; i++)
                            if (i < e) chrs[i] = ids[i] ? (uint32_t)loci[2 * (long)ids[i]] : 0u;  // count_peak_kmer's peak_chr (E:455)
                    } else {
#pragma unroll
                        for (int i = 0; i < 9; i++) chrs[i] = 0u;
                    }
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
        if (by_groups) {
            // Round 6, late: the bound FIRST, over groups of whole contigs instead of hashed contigs -- a peak's group follows from its id
            // (ten steps through a table in LDS), its contig costs a random access (loci: 63 of this kernel's 152 ms in the CLI's default
            // regime).  A contig's final count is at most the entries that name it, hence at most its group's: no group at six -> no
            // vote.  ONE group at six (the read's own contig and its neighbours): the contigs of THAT group's entries are fetched, the
            // first one's counted exactly and taken out -- what is left of the group below six -> no vote.  Everything else fetches the
            // contigs of all entries and goes on as before (the hashed bound, then the walk).
            bool cleared = false;
            if (n_ev >= 32 && !(debug & (1 << 19))) {
                uint32_t* hist = stage + 64;
#pragma unroll
                for (int i = 0; i < BOUND_WORDS / 64; i++) hist[lane + 64 * i] = 0u;
                __builtin_amdgcn_wave_barrier();
                for (int q = lane; q < n_ev * e; q += 64) {
                    const uint32_t id = ev[(size_t)q * 2];
                    if (id) {
                        const uint32_t g = (uint32_t)group_of(id);
                        ev[(size_t)q * 2 + 1] = g;                                        // (kept for the second look; a contig takes its place later)
                        atomicAdd(&hist[g >> 1], 1u << ((g & 1u) * 16u));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                int n6 = 0;
                uint32_t big = 0;
#pragma unroll
                for (int i = 0; i < BOUND_WORDS / 64; i++) {
                    const uint32_t w = hist[lane + 64 * i], lo = w & 0xffffu, hi = w >> 16;
                    n6 += (lo >= 6u) + (hi >= 6u);
                    const uint32_t klo = (lo << 16) | (uint32_t)(2 * (lane + 64 * i)), khi = (hi << 16) | (uint32_t)(2 * (lane + 64 * i) + 1);
                    big = klo > big ? klo : big;
                    big = khi > big ? khi : big;
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) {
                    n6 += __shfl_xor(n6, d, 64);
                    const uint32_t o = (uint32_t)__shfl_xor((int)big, d, 64);
                    big = o > big ? o : big;
                }
                if (n6 == 0) cleared = true;
                else if (n6 == 1) {
                    const uint32_t gsel = big & 0xffffu, total = big >> 16;
                    uint32_t star = 0, n_star = 0;
                    bool have_star = false;
                    for (int q0 = 0; q0 < n_ev * e; q0 += 64) {
                        const int q = q0 + lane;
                        const uint32_t id = q < n_ev * e ? ev[(size_t)q * 2] : 0u;
                        const bool in_g = id != 0u && ev[(size_t)q * 2 + 1] == gsel;
                        const uint32_t chr = in_g ? (uint32_t)loci[2 * (long)id] : 0u;
                        const unsigned long long bal = __ballot(in_g);
                        if (!have_star && bal) {
                            star = (uint32_t)__builtin_amdgcn_readlane((int)chr, __ffsll((long long)bal) - 1);
                            have_star = true;
                        }
                        if (have_star) n_star += (uint32_t)__popcll(__ballot(in_g && chr == star));
                    }
                    if (have_star && total - n_star < 6u) cleared = true;
                }
            }
            if (cleared) continue;
            for (int q = lane; q < n_ev * e; q += 64) {                                    // the contigs of all entries (count_peak_kmer's peak_chr, E:455)
                const uint32_t id = ev[(size_t)q * 2];
                ev[(size_t)q * 2 + 1] = id ? (uint32_t)loci[2 * (long)id] : 0u;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (BOUND_WORDS && n_ev >= 32 && !(debug & (1 << 19))) {
            // Round 4: bound the outcome before walking the events.  judge_base adds every hit offset to exactly ONE of the contigs its
            // hashes point at (E:118-159), so a contig's final count is at most the number of (event, hash) entries that name it, and
            // check_split votes only if TWO contigs reach six (E:161-202).  The entries' contigs are counted into 1024 hashed 16-bit
            // counters: with at most one counter at six or more, and that one below twelve, at most one contig can reach six -- the
            // pair cannot vote and its walk (one dependent chain over hundreds of events: 2.0 of the 2.3 s of a k = 21 step on a
            // 50 Gbase reference, all of it for pairs that vote nothing) is skipped.  A pair the bound cannot clear is walked as before.
            // Round 6: ONE counter at twelve or more is the rule where the peak set is dense -- the read's own contig, named by an entry
            // of nearly every offset, while the other entries scatter over the catalogue.  The contig of the counter's first entry is
            // then counted exactly and taken out: if what is left of the counter stays below six, still only one contig can reach six
            // (the dense vote of the CLI's default regime: 262 -> 152 ms; the walk was 14 k instructions per pair, for 1253 votes in 6.67 M pairs).
            uint32_t* hist = stage + 64;
#pragma unroll
            for (int i = 0; i < BOUND_WORDS / 64; i++) hist[lane + 64 * i] = 0u;
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < n_ev * e; q += 64) {
                const uint32_t id = ev[(size_t)q * 2], chr = ev[(size_t)q * 2 + 1];
                if (id) {
                    const uint32_t h = (chr * 2654435761u) >> 22;                       // 10 bits
                    atomicAdd(&hist[h >> 1], 1u << ((h & 1u) * 16u));                   // <= 9 * max_ev entries: a half never carries
                }
            }
            __builtin_amdgcn_wave_barrier();
            int n6 = 0, n12 = 0;
            uint32_t big = 0;                   // count << 16 | counter number of the lane's largest counter
#pragma unroll
            for (int i = 0; i < BOUND_WORDS / 64; i++) {
                const uint32_t w = hist[lane + 64 * i], lo = w & 0xffffu, hi = w >> 16;
                n6 += (lo >= 6u) + (hi >= 6u);
                n12 += (lo >= 12u) + (hi >= 12u);
                const uint32_t klo = (lo << 16) | (uint32_t)(2 * (lane + 64 * i)), khi = (hi << 16) | (uint32_t)(2 * (lane + 64 * i) + 1);
                big = klo > big ? klo : big;
                big = khi > big ? khi : big;
            }
            int both = n6 | (n12 << 16);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                both += __shfl_xor(both, d, 64);
                const uint32_t o = (uint32_t)__shfl_xor((int)big, d, 64);
                big = o > big ? o : big;
            }
            if ((both & 0xffff) <= 1 && (both >> 16) == 0) continue;
            if ((both & 0xffff) == 1) {         // one counter at six or more, and it is a large one
                const uint32_t hb = big & 0xffffu, total = big >> 16;
                uint32_t star = 0, n_star = 0;
                bool have_star = false;
                for (int q0 = 0; q0 < n_ev * e; q0 += 64) {
                    const int q = q0 + lane;
                    const uint32_t id = q < n_ev * e ? ev[(size_t)q * 2] : 0u, chr = q < n_ev * e ? ev[(size_t)q * 2 + 1] : 0u;
                    if (!have_star) {
                        const unsigned long long in_b = __ballot(id != 0u && ((chr * 2654435761u) >> 22) == hb);
                        if (in_b) {
                            star = (uint32_t)__builtin_amdgcn_readlane((int)chr, __ffsll((long long)in_b) - 1);
                            have_star = true;
                            // (entries before this chunk are not in the counter: none of them is the star's)
                        }
                    }
                    if (have_star) n_star += (uint32_t)__popcll(__ballot(id != 0u && chr == star));
                }
                if (have_star && total - n_star < 6u) continue;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (e == 3) judge_pair<TR, 3>(ev, n_ev, e, lane, filter);
        else judge_pair<TR, 0>(ev, n_ev, e, lane, filter);
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Round 3: the LDS fold at 128 KiB and the judge DEFERRED.  An LDS probe costs a fraction of the L1 miss every bitmap probe is
// (DESIGN.md 4), so what the first level screens out is nearly free -- and a fold screens by its bits per key.  (Round 2's form --
// 64 KiB of fold beside 16 event areas of 5.7 KiB for an in-kernel judge -- was removed in round 4.)  Here a wave
// keeps queues only (480 words) and a pair with a peak k-mer among its survivors, or with more survivors than a queue holds, is
// appended to a list that the generic kernel votes from scratch afterwards (vote_kernel<.., pair_list>): same hits in the same
// offset order, the votes are sums.  That leaves 128 KiB for the fold: twice the bits per key -- 100 M pairs from 300 genomes
// of a ragged 13 Gbase catalogue (597 195 k-mers): 1.14 insertions per fold bit instead of 2.3, 46 % of the probes go on to the
// L2 bitmap instead of all of them.
constexpr int LF2_BITS = 20;
constexpr int LF2_WORDS = (1 << LF2_BITS) / 32;
constexpr int VF_Q = 368, VF_Q2 = 48, VF_WAVE_WORDS = VF_Q + VF_Q2 + 64;   // per wave: first queue (its head stages the records), second queue, scratch
constexpr int VF_WAVES = 16;
// SPM (round 5): slices of 64 k-mer offsets per mate -- 2 takes reads of up to FAST_NK = 128 offsets (159 bases at k = 32), 4 reads of
// up to 256 offsets (287 bases: 250-base reads; the staging words, 32 per mate, hold their three planes of <= 10 words)
template <int EC, int SPM>   // EC = 3: e known at compile time (no uniform branch per hash), 0: e <= 3 at run time
__global__ void __launch_bounds__(64 * VF_WAVES) vote_kernel_fold(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                                  const uint32_t* __restrict__ prefilter, const uint32_t* __restrict__ lds_fold,
                                                                  int fold_words, uint32_t* __restrict__ revote, int debug, uint32_t pf_mask, int pf2,
                                                                  unsigned long long* __restrict__ stats /* nullable: lhgt_work_stats */) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int e = EC ? EC : hp.e, k = hp.k;
    if (b.n_pairs <= 0) return;
    for (int i = threadIdx.x; i < fold_words; i += blockDim.x) lds[i] = lds_fold[i];
    __syncthreads();
    const uint32_t lf_mask = (uint32_t)fold_words * 32u - 1u;
    uint32_t* Q = lds + LF2_WORDS + (size_t)wib * VF_WAVE_WORDS;
    uint32_t* Q2 = Q + VF_Q;
    uint32_t* dump = Q2 + VF_Q2;
    uint32_t* stage = Q;
    const long wave = (long)blockIdx.x * VF_WAVES + wib;
    const long n_waves = (long)gridDim.x * VF_WAVES;
    // A wave's pairs form a chain of dependent loads (offsets and lengths, then the record) in front of a few microseconds of work:
    // the record of the next pair and the offsets of the one after it are fetched while the current pair is worked on
    struct Meta { int len[2]; uint32_t off[2]; bool vote; };
    auto load_meta = [&](long p) {
        Meta m;
        const long pc = p < b.n_pairs ? p : b.n_pairs - 1;     // clamped: every load unconditional, a pair past the end is never voted
        m.len[0] = b.len[0][pc]; m.len[1] = b.len[1][pc];
        m.off[0] = b.off[0][pc]; m.off[1] = b.off[1][pc];
        m.vote = p < b.n_pairs && (!b.flags || (b.flags[pc] & PAIR_VOTE));
        return m;
    };
    auto load_rec = [&](const Meta& m, int mate) {
        const int w3 = 3 * (((m.len[mate] + 31) >> 5) + 1);
        return lane < w3 ? (b.words + m.off[mate])[lane] : 0u;
    };
    Meta m_next = load_meta(wave), m_next2 = load_meta(wave + n_waves);
    uint32_t rw[2] = {load_rec(m_next, 0), load_rec(m_next, 1)};
    // pending third-level probes: lane i < n_pend holds one survivor of the second level and its pair; a pair's survivors are
    // appended together, so they sit in consecutive lanes of ONE flush: its hits are counted there and the pair is deferred once.
    // (A pair deferred here AND by an overflow would be voted twice: an overflowing pair never enters this queue.)
    uint32_t pend_h = 0u, pend_p = 0u;
    int n_pend = 0;
    unsigned long long st_l2 = 0, st_hbm = 0;   // wave-uniform: probes sent on to the L2 bitmap / to peak_kmer
    auto flush_pending = [&]() {
        const bool hit = lane < n_pend && peak_kmer[pend_h] != 0u;
        const unsigned long long hits = __ballot(hit);
        if (hits) {
            const uint32_t prev_p = __shfl_up(pend_p, 1, 64);
            const bool head = lane < n_pend && (lane == 0 || prev_p != pend_p);
            const unsigned long long heads = __ballot(head);
            if (head) {
                const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;      // heads of the later pairs
                const int end = above ? lane + 1 + __ffsll((long long)above) - 1 : n_pend;     // this pair's lanes: [lane, end)
                const unsigned long long seg = (end >= 64 ? ~0ull : (1ull << end) - 1ull) & ~((1ull << lane) - 1ull);
                // base_hits >= 6 (E:496) needs six offsets with a hit, i.e. at least six probes that found a peak id: one or two are what
                // foreign k-mers bring by collision (714 probes against 2*10^5 registered k-mers in 2^32 slots: every 30th pair)
                if (__popcll(hits & seg) >= 6) revote[1u + atomicAdd(revote, 1u)] = pend_p;
            }
        }
        n_pend = 0;
    };
    for (long p = wave; p < b.n_pairs; p += n_waves) {
        const Meta cur = m_next;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < 2; m++)
            if (lane < 32) stage[m * 32 + lane] = rw[m];
        __builtin_amdgcn_wave_barrier();
        m_next = m_next2;
        rw[0] = load_rec(m_next, 0);
        rw[1] = load_rec(m_next, 1);
        m_next2 = load_meta(p + 2 * n_waves);
        if (!cur.vote) continue;
        int nk[2], wpr[2];
#pragma unroll
        for (int m = 0; m < 2; m++) {
            nk[m] = cur.len[m] - k + 1;
            wpr[m] = ((cur.len[m] + 31) >> 5) + 1;
        }
        if (nk[0] > 64 * SPM || nk[1] > 64 * SPM) {     // a read longer than this instance's windows and staging words take: the generic kernel votes the pair
            if (lane == 0) revote[1u + atomicAdd(revote, 1u)] = (uint32_t)p;
            continue;
        }
        // In three sweeps -- all window words, all hashes, all fold probes -- and every load unconditional (a load under a lane mask
        // is an exec region with its own wait: a dozen LDS round trips in a row instead of one); a hash the run does not have
        // (i >= e) reads mask 0 and is switched off with its `ok` bit.
        constexpr int NS = 2 * SPM;
        uint32_t hs[NS][3], f1[NS][3], wd[NS][6];
        bool ok[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int m = s / SPM, j = (s % SPM) * 64 + lane;
            const uint32_t* q = stage + m * 32 + (j < nk[m] ? (j >> 5) : 0);
            const int wp = wpr[m];
            wd[s][0] = q[0]; wd[s][1] = q[1]; wd[s][2] = q[wp]; wd[s][3] = q[wp + 1]; wd[s][4] = q[2 * wp]; wd[s][5] = q[2 * wp + 1];
        }
#pragma unroll
        for (int s = 0; s < NS; s++) {
            const int m = s / SPM, j = (s % SPM) * 64 + lane, r = j & 31;
            auto win = [&](uint32_t a, uint32_t c) { return window32(a, c, r) >> (32 - k); };
            const uint32_t whi = win(wd[s][0], wd[s][1]), wlo = win(wd[s][2], wd[s][3]), wnb = win(wd[s][4], wd[s][5]);
            const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
            ok[s] = j < nk[m] && wnb == 0;
#pragma unroll
            for (int i = 0; i < 3; i++) hs[s][i] = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
        }
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int i = 0; i < 3; i++) f1[s][i] = lds[(hs[s][i] & lf_mask) >> 5];
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int i = 0; i < 3; i++) f1[s][i] = (uint32_t)pf_pass(f1[s][i], hs[s][i], pf2) & (uint32_t)(ok[s] & (i < e));
        __builtin_amdgcn_wave_barrier();
        int c = 0;
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int i = 0; i < 3; i++) c += (int)f1[s][i];
        const int incl = wave_incl_scan(c, lane);
        const int T = (debug & 512) ? 0 : __shfl(incl, 63, 64);   // bit9 / bit10: stage timing (tools/ablate_vote.py), outputs wrong
        if (T == 0) continue;
        bool defer = T > VF_Q;
        if (!defer) {
            st_l2 += (unsigned long long)T;
            int slot = incl - c;
#pragma unroll
            for (int s = 0; s < NS; s++)
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    uint32_t* dst = f1[s][i] ? Q + slot : dump + lane;   // no branch: dead candidates land in a scratch word
                    *dst = hs[s][i];
                    slot += (int)f1[s][i];
                }
            __builtin_amdgcn_wave_barrier();
            int T2 = 0;
            for (int q0 = 0; q0 < T; q0 += 128) {   // second level, the L2 bitmap: two gathers in flight per round
                const int qa = q0 + lane, qb = qa + 64;
                const uint32_t ha = qa < T ? Q[qa] : 0u, hb = qb < T ? Q[qb] : 0u;
                const uint32_t wa = qa < T ? prefilter[pf_word(ha, pf_mask)] : 0u;
                const uint32_t wb = qb < T ? prefilter[pf_word(hb, pf_mask)] : 0u;
                const bool pa = qa < T && pf_pass(wa, ha, pf2), pb = qb < T && pf_pass(wb, hb, pf2);
                const unsigned long long ba = __ballot(pa), bb = __ballot(pb);
                const int sa = T2 + __popcll(ba & ((1ull << lane) - 1ull));
                const int sb = T2 + __popcll(ba) + __popcll(bb & ((1ull << lane) - 1ull));
                if (pa && sa < VF_Q2) Q2[sa] = ha;
                if (pb && sb < VF_Q2) Q2[sb] = hb;
                T2 += __popcll(ba) + __popcll(bb);
            }
            __builtin_amdgcn_wave_barrier();
            if (debug & 1024) T2 = 0;
            if (T2 > VF_Q2) defer = true;
            else if (T2 > 0) {
                st_hbm += (unsigned long long)T2;
                // third level, peak_kmer itself: not now -- that is one HBM round trip per pair in this wave's chain -- but from a
                // wave-wide register queue of (hash, pair) that is probed when the next pair's survivors no longer fit
                if (n_pend + T2 > 64) flush_pending();
                if (lane >= n_pend && lane < n_pend + T2) {
                    pend_h = Q2[lane - n_pend];
                    pend_p = (uint32_t)p;
                }
                n_pend += T2;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (defer && lane == 0) revote[1u + atomicAdd(revote, 1u)] = (uint32_t)p;
    }
    flush_pending();
    if (stats && lane == 0) { atomicAdd(stats + 3, st_l2); atomicAdd(stats + 4, st_hbm); }
}

// Sparse path without the LDS fold (the usual one: bitmap exact for k <= 25, or too many k-mers for a 64 KiB fold).  One pair
// per wave iteration as above, first level = the L2-resident bitmap.  Its few survivors (12 of a pair's 714 probes on configs[2])
// are not probed at once -- that is one HBM round trip per pair for a dozen lanes -- but QUEUED across pairs, tagged with the
// pair's iteration number, and probed 64 at a time by a full-width gather.  A pair one of whose probes finds a peak id is then
// voted from scratch in the lane-per-offset form (hits in offset order for the judge); the others are done.  A pair's entries
// are appended together and never split over two flushes, so no pair is voted twice.  (Straight-line code, one site per step:
// as lambdas called from several places the steps became real calls with their captures in scratch memory.)
// Queue of 1024 entries, flushed at 512: a pair may bring up to 512 survivors.  (Round 1: 384 / 192, sized for the dozen survivors
// per pair of a 2.3 M-k-mer peak set; a reference of 118 k ragged contigs registers 14 M k-mers, a third of the bitmap probes
// pass, and 235 survivors per pair sent every pair down the direct path.)
constexpr int VQ_CAP = 1024, VQ_FLUSH = 512;
template <bool Q3>   // the bitmap is three quarters of the 4 MiB its mask spans (lhgt_hash.hpp: PF_Q3)
__global__ void __launch_bounds__(256) vote_kernel_queued(ReadBatchDev b, HashParams hp, const uint32_t* __restrict__ peak_kmer,
                                                          const uint32_t* __restrict__ prefilter, const int32_t* __restrict__ loci,
                                                          uint32_t* __restrict__ filter, int max_ev, int waves_per_block, int debug,
                                                          uint32_t pf_mask, int pf2, unsigned long long* __restrict__ stats /* nullable: lhgt_work_stats */,
                                                          uint32_t* __restrict__ long_pairs /* nullable: [0] = how many, then the pairs with a read of more than FAST_NK offsets */) {
    extern __shared__ __align__(16) uint32_t lds[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int e = hp.e, k = hp.k;
    if (wib >= waves_per_block) return;
    unsigned long long st_hbm = 0, st_revote = 0;   // wave-uniform: probes sent on to peak_kmer, pairs voted in the lane-per-offset form
    // per wave: [queue hashes | queue tags | 64 dump words | 64 staging words of the scan] and, over the same words, the vote's
    // [events | 64 staging words]: by the time a pair is voted the queue is empty and the tags to vote sit in registers
    const int ev_words = max_ev * e * 2;
    const int wave_words = ev_words + 64 > 2 * VQ_CAP + 128 ? ev_words + 64 : 2 * VQ_CAP + 128;
    uint32_t* qh = lds + (size_t)wib * wave_words;
    uint32_t* qi = qh + VQ_CAP;    // tags; during a flush its front collects the pairs to vote
    uint32_t* stage = qh + 2 * VQ_CAP + 64;
    uint32_t* ev = qh;
    uint32_t* vstage = ev + ev_words;
    const long wave = (long)blockIdx.x * waves_per_block + wib;
    const long n_waves = (long)gridDim.x * waves_per_block;
    const uint32_t m512 = (debug & 512) ? 0u : 1u;   // stage ablation: stop after the bitmap / (1024) before the peak_kmer gathers
    const bool skip_gather = (debug & 1024) != 0;
    const bool nt_probe = !(debug & (1 << 17)), nt_rec = !(debug & (1 << 18));   // A/B switches of the two non-temporal hints (bits set: plain loads)
    int qn = 0;
    uint32_t it = 0;
    for (long p = wave;; p += n_waves, it++) {
        const bool live = p < b.n_pairs;   // one more round after the last pair drains the queue
        int T = 0;
        bool mine = live && !(b.flags && !(b.flags[p] & PAIR_VOTE));
        if (mine && long_pairs && (b.len[0][p] - k + 1 > FAST_NK || b.len[1][p] - k + 1 > FAST_NK)) {
            if (lane == 0) long_pairs[1u + atomicAdd(long_pairs, 1u)] = (uint32_t)p;   // voted by the generic kernel behind this one
            mine = false;
        }
        if (mine) {
            // two round trips: the four descriptors together, then both records (lane index clamped instead of a lane-masked
            // load, which the compiler would wait for on its own)
            const int len0 = b.len[0][p], len1 = b.len[1][p];
            const uint32_t* rec0 = b.words + b.off[0][p];
            const uint32_t* rec1 = b.words + b.off[1][p];
            const int nk[2] = {len0 - k + 1, len1 - k + 1};
            const int wpr[2] = {((len0 + 31) >> 5) + 1, ((len1 + 31) >> 5) + 1};
            // (bit 18: the records as non-temporal loads -- they stream through once and should not push bitmap lines out of the L2)
            const uint32_t* ra0 = rec0 + (lane < 3 * wpr[0] ? lane : 3 * wpr[0] - 1);
            const uint32_t* ra1 = rec1 + (lane < 3 * wpr[1] ? lane : 3 * wpr[1] - 1);
            const uint32_t rw[2] = {nt_rec ? __builtin_nontemporal_load(ra0) : *ra0, nt_rec ? __builtin_nontemporal_load(ra1) : *ra1};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int m = 0; m < 2; m++)
                if (lane < 32) stage[m * 32 + lane] = rw[m];
            __builtin_amdgcn_wave_barrier();
            uint32_t hs[4][3], f1[4][3];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int m = s >> 1, j = (s & 1) * 64 + lane, r = j & 31;
                const uint32_t* q = stage + m * 32 + (j < nk[m] ? (j >> 5) : 0);
                const int wp = wpr[m];
                auto win = [&](uint32_t a, uint32_t c) { return window32(a, c, r) >> (32 - k); };
                const uint32_t whi = win(q[0], q[1]), wlo = win(q[wp], q[wp + 1]), wnb = win(q[2 * wp], q[2 * wp + 1]);
                const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
                const bool ok = j < nk[m] && wnb == 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const uint32_t h = i < e ? hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]) : 0u;
                    hs[s][i] = h;
                    // unconditional load (a dead lane probes word 0), masked afterwards: under `ok &&` every load would sit in its own
                    // lane-masked branch next to its use and be waited for singly
                    const uint32_t pass = pf_pass(prefilter[pf_word_t<Q3>(h, pf_mask)], h, pf2) ? 1u : 0u;
                    f1[s][i] = (ok && i < e) ? (pass & m512) : 0u;
                }
            }
            int c = 0;
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int i = 0; i < 3; i++) c += (int)f1[s][i];
            const int incl = wave_incl_scan(c, lane);
            T = __shfl(incl, 63, 64);
            if (T > 0 && T <= ((debug & 2048) ? 8 : VQ_CAP - VQ_FLUSH)) {   // qn < VQ_FLUSH here, so the pair's entries always fit
                int slot = qn + incl - c;
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const int at = f1[s][i] ? slot : 2 * VQ_CAP + lane;   // dead candidates land in the dump words behind the tags
                        qh[at] = hs[s][i];
                        if (f1[s][i]) qi[slot] = it;
                        slot += (int)f1[s][i];
                    }
                qn += T;
                st_hbm += (unsigned long long)T;
            }
        }
        const bool direct = T > ((debug & 2048) ? 8 : VQ_CAP - VQ_FLUSH);   // not a sparse pair: vote it as it is (bit 11: test hook)
        if (qn >= VQ_FLUSH || direct || !live) {
            // flush: full-width gathers over the queue; the tags of pairs with a hit are collected at the front of qi
            __builtin_amdgcn_wave_barrier();
            int nv = 0, cur_cnt = 0;
            uint32_t cur = 0xffffffffu;   // a pair's entries are contiguous (possibly over two rounds): count its hit entries as they come
            uint32_t idv[VQ_CAP / 64], tagv[VQ_CAP / 64];
#pragma unroll
            for (int u = 0; u < VQ_CAP / 64; u++) {   // all gathers of the flush in flight together
                const int q = u * 64 + lane;
                const bool in = q < qn && !skip_gather;
                tagv[u] = in ? qi[q] : 0xffffffffu;
                idv[u] = 0u;
                if (in) idv[u] = nt_probe ? __builtin_nontemporal_load(peak_kmer + qh[q]) : peak_kmer[qh[q]];
            }
#pragma unroll
            for (int u = 0; u < VQ_CAP / 64; u++) {
                const uint32_t tag = tagv[u];
                const bool hit = idv[u] != 0u;
                unsigned long long bal = __ballot(hit);
                while (bal) {
                    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)tag, __ffsll((long long)bal) - 1);
                    const unsigned long long same = __ballot(hit && tag == t);
                    bal &= ~same;
                    if (t != cur) {
                        // base_hits >= 6 (E:496) needs six offsets with a hit, so at least six hit entries: fewer cannot vote
                        if (cur_cnt >= 6) { if (lane == 0) qi[nv] = cur; nv++; }
                        cur = t;
                        cur_cnt = 0;
                    }
                    cur_cnt += __popcll(same);
                }
            }
            if (cur_cnt >= 6) { if (lane == 0) qi[nv] = cur; nv++; }
            qn = 0;
            if (direct) {
                if (lane == 0) qi[nv] = it;
                nv++;
            }
            st_revote += (unsigned long long)nv;
            __builtin_amdgcn_wave_barrier();
            uint32_t vt[VQ_CAP / 64];   // the tags leave LDS: the events below are written over the queue
#pragma unroll
            for (int u = 0; u < VQ_CAP / 64; u++) vt[u] = u * 64 + lane < nv ? qi[u * 64 + lane] : 0u;
            __builtin_amdgcn_wave_barrier();
            // the generic kernel's treatment of those pairs: every offset, masked probes, hits compacted in offset order, judge
            for (int v = 0; v < nv; v++) {
                uint32_t tv = 0;
#pragma unroll
                for (int u = 0; u < VQ_CAP / 64; u++)
                    if ((v >> 6) == u) tv = (uint32_t)__builtin_amdgcn_readlane((int)vt[u], v & 63);
                const long pv = wave + (long)tv * n_waves;
                int n_ev = 0;
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int len = b.len[m][pv];
                    const int nkm = len - k + 1;
                    if (nkm <= 0) continue;
                    const int wprm = ((len + 31) >> 5) + 1;
                    const uint32_t* rec = b.words + b.off[m][pv];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 3 * wprm) vstage[lane] = rec[lane];
                    __builtin_amdgcn_wave_barrier();
                    for (int j0 = 0; j0 < nkm; j0 += 64) {
                        const int j = j0 + lane;
                        uint32_t ids[3] = {0u, 0u, 0u}, chrs[3] = {0u, 0u, 0u};
                        bool hit = false;
                        if (j < nkm && plane_window(vstage + 2 * wprm, j, k) == 0) {
                            const uint32_t whi = plane_window(vstage, j, k), wlo = plane_window(vstage + wprm, j, k);
                            const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
#pragma unroll
                            for (int i = 0; i < 3; i++)
                                if (i < e) {
                                    const uint32_t h = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
                                    ids[i] = pf_pass(prefilter[pf_word_t<Q3>(h, pf_mask)], h, pf2) ? peak_kmer[h] : 0u;   // 0 = no peak (E:454)
                                    hit |= ids[i] != 0u;
                                }
#pragma unroll
                            for (int i = 0; i < 3; i++)
                                if (i < e) chrs[i] = ids[i] ? (uint32_t)loci[2 * (long)ids[i]] : 0u;
                        }
                        const unsigned long long bal = __ballot(hit);
                        if (bal) {
                            if (hit) {
                                const int slot = n_ev + __popcll(bal & ((1ull << lane) - 1ull));
#pragma unroll
                                for (int i = 0; i < 3; i++)
                                    if (i < e) {
                                        ev[((size_t)slot * e + i) * 2] = ids[i];
                                        ev[((size_t)slot * e + i) * 2 + 1] = chrs[i];
                                    }
                            }
                            n_ev += __popcll(bal);
                        }
                    }
                }
                if (n_ev >= 6 && !(debug & 1)) {   // base_hits = offsets with any hit (E:149-157, 496)
                    __builtin_amdgcn_wave_barrier();
                    if (e == 3) judge_pair<4, 3>(ev, n_ev, e, lane, filter);
                    else judge_pair<4, 0>(ev, n_ev, e, lane, filter);
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!live) break;
    }
    if (stats && lane == 0) { atomicAdd(stats + 4, st_hbm); atomicAdd(stats + 5, st_revote); }
}

// fold of the 2^pf_bits-bit bitmap onto fold_words words (64 or 128 KiB): word w = OR of the bitmap words w, w + fold_words, ... (same low address bits)
__global__ void __launch_bounds__(256) fold_prefilter(const uint32_t* __restrict__ prefilter, int words, uint32_t* __restrict__ fold, int fold_words) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= fold_words) return;
    uint32_t acc = 0;
    for (int j = w; j < words; j += fold_words) acc |= prefilter[j];
    fold[w] = acc;
}

__global__ void stats_add_u32(const uint32_t* __restrict__ v, unsigned long long* __restrict__ out) { atomicAdd(out, (unsigned long long)*v); }

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

int lhgt_vote_shared(lhgt_ctx* ctx, bool* done, const uint32_t** d_list, unsigned long long* d_stats);   // k_vote_shared.hip

extern "C" {

int lhgt_vote(lhgt_ctx* ctx) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "lhgt_ref_scan must precede lhgt_vote");
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    unsigned long long* d_stats = ctx->stats_on ? ctx->d_stats : nullptr;
    // a dense peak set under a deep sample: the probes of overlapping reads share their line fills (k_vote_shared.hip); what that
    // form leaves over -- pairs with a long read, pairs whose events found no room -- comes back as a list of global pair numbers
    bool shared_done = false;
    const uint32_t* shared_list = nullptr;
    if (!ctx->prefilter_on) LHGT_TRY(lhgt_vote_shared(ctx, &shared_done, &shared_list, d_stats));
    long pair_base = 0;
    for (const ReadBatch& b : ctx->batches) {
        const long base_here = pair_base;
        pair_base += b.d.n_pairs;
        int nk = b.max_len - ctx->k + 1;
        if (nk <= 0) continue;
        if (shared_done) {
            ctx->vote_form = 4;
            const int ev_all = 2 * nk;
            const size_t pw = ((size_t)ev_all * ctx->e * 2 + 64) * 4 + 512 * 4;      // (a dense instance: the counters of the vote bound at every TR)
            int w = (int)(65536 / pw);
            w = w > 4 ? 4 : w < 1 ? 1 : w;
            long nb = (b.d.n_pairs + w - 1) / w;
            if (nb > 256L * 16) nb = 256L * 16;
#define LHGT_VOTE_REST(TR_)                                                                                                          \
    do {                                                                                                                             \
        if (pw * w > 65536) LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel<TR_, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((vote_kernel<TR_, 0, true>), dim3((unsigned)nb), dim3(64 * w), pw * w, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter, \
                           ctx->d_prefilter_fold, ctx->d_loci, ctx->d_filter, ev_all, w, ctx->debug, 0u, 0, shared_list, base_here, (const uint32_t*)nullptr);  \
    } while (0)
            if (ev_all <= 256) LHGT_VOTE_REST(4);
            else if (ev_all <= 512) LHGT_VOTE_REST(8);
            else LHGT_VOTE_REST(16);
#undef LHGT_VOTE_REST
            continue;
        }
        // a batch of short reads with a few long ones (lhgt_common.hpp: ReadBatch::n_long): the sparse forms take the pairs of short
        // reads and list the others, which the generic kernel votes behind them (votes are sums: the order is free)
        const bool mixed = nk > FAST_NK && b.n_long >= 0 && b.n_long * 8 <= 2 * b.d.n_pairs && ctx->e <= 3 && ctx->prefilter_on && !(ctx->debug & 32);
        const int nk_all = nk;
        // ... and a batch of reads of 160-287 bases (256 offsets) takes the fold kernel's wide instance where the fold applies (e = 3);
        // its few still longer pairs, if any, are listed like a mixed batch's
        static const double fold_max = getenv("LHGT_FOLD_MAX") ? atof(getenv("LHGT_FOLD_MAX")) : 0.0;   // bit insertions per fold bit (0 = the defaults)
        const double fold_ins = (double)(ctx->n_selected * (unsigned long long)ctx->e * (ctx->pf2 ? 2 : 1));
        const bool fold_ok = fold_ins <= (fold_max > 0 ? fold_max : 1.15) * (double)(1ull << LF2_BITS);
        const bool wide = !mixed && nk > FAST_NK && nk <= 2 * FAST_NK && ctx->e == 3 && ctx->prefilter_on && !(ctx->debug & (32 | 16)) && ctx->k > PF_BITS &&
                          !ctx->pf_q3 && fold_ok;
        if (mixed || wide) nk = FAST_NK;         // (the sparse forms' own sizes below are those of short reads; vote_list sizes for nk_all)
        int max_ev = 2 * nk;
        size_t per_wave = ((size_t)max_ev * ctx->e * 2 + 64) * 4;   // events + 64 staging words
        int wpb = (int)(65536 / per_wave);
        if (wpb > 4) wpb = 4;
        if (wpb < 1) wpb = 1;
        long blocks = (b.d.n_pairs + wpb - 1) / wpb;
        if (blocks > 256L * 16) blocks = 256L * 16;
        // long reads with many hashes need more than the default 64 KiB of dynamic LDS for one wave's event list (500 bases, e = 9:
        // 67.5 KiB); gfx950 has 160 KiB per workgroup
#define LHGT_VOTE(TR_, PF_, NT_, THREADS_, LDS_)                                                                           \
    do {                                                                                                                   \
        if ((size_t)(LDS_) > 65536)                                                                                        \
            LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel<TR_, PF_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((vote_kernel<TR_, PF_, NT_>), dim3((unsigned)blocks), dim3(THREADS_), LDS_, ctx->stream, b.d, ctx->hp,  \
                           ctx->d_peak_kmer, ctx->d_prefilter, ctx->d_prefilter_fold, ctx->d_loci, ctx->d_filter, max_ev, wpb,    \
                           ctx->debug, ctx->pf_mask | (ctx->pf_q3 ? PF_Q3 : 0u), ctx->pf2, (const uint32_t*)nullptr, 0L, (PF_) == 0 ? vgroups : (const uint32_t*)nullptr); \
    } while (0)
        const bool nt = ctx->k >= 28;
        // the dense forms' bound over groups of whole contigs (vote_kernel): the table lhgt_ref_scan left, 4 KiB of LDS per workgroup
        const bool groups_off = getenv("LHGT_VOTE_GROUPS") && !atoi(getenv("LHGT_VOTE_GROUPS"));       // (read per vote: the tests switch it)
        // (where most probes hit: with a quarter of the slots registered or more.  On a sparser registry the events are few and their
        // contigs are better fetched while the probes are still in flight -- 58 -> 65 ms on the batch leg's 1.5 M peaks)
        const bool groups_pay = (double)ctx->n_selected * (double)ctx->e >= 0.25 * (double)((size_t)1 << ctx->k) || (getenv("LHGT_VOTE_GROUPS") && atoi(getenv("LHGT_VOTE_GROUPS")) == 2);
        const uint32_t* vgroups = ctx->vote_groups_ok && !groups_off && groups_pay && !ctx->prefilter_on ? ctx->d_vote_groups : nullptr;
        const size_t vg_lds = vgroups ? (size_t)VG_N * 4 : 0;
        // the pairs on a list (the fold form's deferred pairs; a mixed batch's pairs with a long read) in the generic form, sized for
        // the batch's longest read
        auto vote_list = [&](const uint32_t* list) -> int {
            const int ev_all = 2 * nk_all;
            size_t pw = ((size_t)ev_all * ctx->e * 2 + 64) * 4 + (ev_all > 256 ? 512 * 4 : 0);
            int w = (int)(65536 / pw);
            w = w > 4 ? 4 : w < 1 ? 1 : w;
            long nb = (b.d.n_pairs + w - 1) / w;
            if (nb > 256L * 16) nb = 256L * 16;
#define LHGT_VOTE_LIST(TR_)                                                                                                          \
    do {                                                                                                                             \
        if (pw * w > 65536) LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel<TR_, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((vote_kernel<TR_, 1, false>), dim3((unsigned)nb), dim3(64 * w), pw * w, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter, \
                           ctx->d_prefilter_fold, ctx->d_loci, ctx->d_filter, ev_all, w, ctx->debug, ctx->pf_mask | (ctx->pf_q3 ? PF_Q3 : 0u), ctx->pf2, list, 0L, (const uint32_t*)nullptr);   \
    } while (0)
            if (ev_all <= 256) LHGT_VOTE_LIST(4);
            else if (ev_all <= 512) LHGT_VOTE_LIST(8);
            else LHGT_VOTE_LIST(16);
#undef LHGT_VOTE_LIST
            return LHGT_OK;
        };
        auto list_room = [&]() -> int {
            const size_t need = (size_t)b.d.n_pairs + 1;
            if (ctx->revote_cap < need) {
                if (ctx->d_revote) LHGT_HIP(lhgt::dev_free(ctx->d_revote));
                ctx->d_revote = nullptr;
                LHGT_HIP(lhgt::dev_malloc(&ctx->d_revote, need * 4));
                ctx->revote_cap = need;
            }
            LHGT_HIP(hipMemsetAsync(ctx->d_revote, 0, 4, ctx->stream));
            return LHGT_OK;
        };
        const bool sparse_ok = ctx->prefilter_on && nk <= FAST_NK && ctx->e <= 3 && !(ctx->debug & 32);
        // LDS first level while the fold still screens.  The 128 KiB fold with the judge deferred (vote_kernel_fold) up to 1.15 bit
        // insertions per fold bit: 47 % of foreign probes pass on to the L2 bitmap, a pair's survivors (333 +- 13 of 714) still fit
        // the wave's queue -- beyond that the overflowing pairs would flood the deferred list.  Round 2 kept a 64 KiB fold up to a
        // quarter insertion per bit; measured in round 3 on 100 M pairs from 300 genomes of the 13 Gbase reference (205 410 registered
        // k-mers): 296 ms without a fold, 146 ms with 64 KiB (0.78 insertions per bit), 128 KiB below.  2.3 M k-mers (configs[2])
        // fill any fold that fits.
        ctx->vote_form = ctx->prefilter_on ? 1 : 0;
        if (sparse_ok && ctx->k > PF_BITS && fold_ok && !ctx->pf_q3 && !(ctx->debug & 16)) {
            ctx->vote_form = 3;
            const int fold_words = (int)std::min<unsigned long long>(LF2_WORDS, (ctx->pf_mask + 1ull) / 32);
            const size_t lds3 = (size_t)(LF2_WORDS + VF_WAVES * VF_WAVE_WORDS) * 4;
            LHGT_TRY(list_room());
            LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel_fold<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel_fold<0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            LHGT_HIP(hipFuncSetAttribute((const void*)vote_kernel_fold<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
            hipLaunchKernelGGL(fold_prefilter, dim3((fold_words + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_prefilter, (int)((ctx->pf_mask + 1ull) / 32),
                               ctx->d_prefilter_fold, fold_words);
            long fb = (b.d.n_pairs + VF_WAVES - 1) / VF_WAVES;
            if (fb > 256) fb = 256;             // one resident workgroup per CU
            if (wide)               // reads of 160-287 bases throughout: four slices of 64 offsets per mate
                hipLaunchKernelGGL((vote_kernel_fold<3, 4>), dim3((unsigned)fb), dim3(64 * VF_WAVES), lds3, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter,
                                   ctx->d_prefilter_fold, fold_words, ctx->d_revote, ctx->debug, ctx->pf_mask, ctx->pf2, d_stats);
            else if (ctx->e == 3)
                hipLaunchKernelGGL((vote_kernel_fold<3, 2>), dim3((unsigned)fb), dim3(64 * VF_WAVES), lds3, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter,
                                   ctx->d_prefilter_fold, fold_words, ctx->d_revote, ctx->debug, ctx->pf_mask, ctx->pf2, d_stats);
            else
                hipLaunchKernelGGL((vote_kernel_fold<0, 2>), dim3((unsigned)fb), dim3(64 * VF_WAVES), lds3, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer, ctx->d_prefilter,
                                   ctx->d_prefilter_fold, fold_words, ctx->d_revote, ctx->debug, ctx->pf_mask, ctx->pf2, d_stats);
            // the deferred pairs, from scratch in the lane-per-offset form (hits in offset order for the judge); the list's length
            // is read on the device
            LHGT_TRY(vote_list((const uint32_t*)ctx->d_revote));
            if (d_stats) hipLaunchKernelGGL(stats_add_u32, dim3(1), dim3(1), 0, ctx->stream, (const uint32_t*)ctx->d_revote, d_stats + 5);
            if (getenv("LHGT_TRACE")) {
                uint32_t n_def = 0;
                LHGT_HIP(hipMemcpyAsync(&n_def, ctx->d_revote, 4, hipMemcpyDeviceToHost, ctx->stream));
                LHGT_HIP(hipStreamSynchronize(ctx->stream));
                fprintf(stderr, "[lhgt] vote: 128 KiB fold (%.2f insertions per bit), %u of %ld pairs deferred to the lane-per-offset form\n",
                        fold_ins / (double)(1ull << LF2_BITS), n_def, b.d.n_pairs);
            }
        } else if (sparse_ok) {
            ctx->vote_form = 2;
            const size_t per_wave_q = (size_t)std::max(max_ev * ctx->e * 2 + 64, 2 * VQ_CAP + 128) * 4;
            wpb = (int)(65536 / per_wave_q);
            if (wpb > 4) wpb = 4;
            if (wpb < 1) wpb = 1;
            blocks = (b.d.n_pairs + wpb - 1) / wpb;
            if (blocks > 256L * 16) blocks = 256L * 16;
            uint32_t* long_list = nullptr;
            if (mixed) { LHGT_TRY(list_room()); long_list = ctx->d_revote; }
            if (ctx->pf_q3)
                hipLaunchKernelGGL(vote_kernel_queued<true>, dim3((unsigned)blocks), dim3(64 * wpb), per_wave_q * wpb, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer,
                                   ctx->d_prefilter, ctx->d_loci, ctx->d_filter, max_ev, wpb, ctx->debug, ctx->pf_mask, ctx->pf2, d_stats, long_list);
            else
                hipLaunchKernelGGL(vote_kernel_queued<false>, dim3((unsigned)blocks), dim3(64 * wpb), per_wave_q * wpb, ctx->stream, b.d, ctx->hp, ctx->d_peak_kmer,
                                   ctx->d_prefilter, ctx->d_loci, ctx->d_filter, max_ev, wpb, ctx->debug, ctx->pf_mask, ctx->pf2, d_stats, long_list);
            if (mixed) LHGT_TRY(vote_list((const uint32_t*)ctx->d_revote));
        } else if (max_ev <= 256) {
            if (ctx->prefilter_on) LHGT_VOTE(4, 1, false, 64 * wpb, per_wave * wpb);
            else {
                // a dense instance: every wave also holds the 2 KiB of counters of the vote bound
                const size_t per_wave_b = per_wave + 512 * 4;
                wpb = (int)((65536 - vg_lds) / per_wave_b);
                wpb = wpb > 4 ? 4 : wpb < 1 ? 1 : wpb;
                blocks = (b.d.n_pairs + wpb - 1) / wpb;
                if (blocks > 256L * 16) blocks = 256L * 16;
                if (nt) LHGT_VOTE(4, 0, true, 64 * wpb, per_wave_b * wpb + vg_lds);
                else LHGT_VOTE(4, 0, false, 64 * wpb, per_wave_b * wpb + vg_lds);
            }
        } else {
            // long event lists: every wave also holds the 2 KiB of counters of the vote bound (vote_kernel, TR >= 8)
            const size_t per_wave_b = per_wave + 512 * 4;
            wpb = (int)((65536 - vg_lds) / per_wave_b);
            if (wpb > 4) wpb = 4;
            if (wpb < 1) wpb = 1;
            blocks = (b.d.n_pairs + wpb - 1) / wpb;
            if (blocks > 256L * 16) blocks = 256L * 16;
            if (max_ev <= 512) {   // a table row per 64 events: 8 rows instead of 16 (k = 21 with 150-base reads has 260 events)
                if (ctx->prefilter_on) LHGT_VOTE(8, 1, false, 64 * wpb, per_wave_b * wpb);
                else if (nt) LHGT_VOTE(8, 0, true, 64 * wpb, per_wave_b * wpb + vg_lds);
                else LHGT_VOTE(8, 0, false, 64 * wpb, per_wave_b * wpb + vg_lds);
            } else {
                if (ctx->prefilter_on) LHGT_VOTE(16, 1, false, 64 * wpb, per_wave_b * wpb);
                else if (nt) LHGT_VOTE(16, 0, true, 64 * wpb, per_wave_b * wpb + vg_lds);
                else LHGT_VOTE(16, 0, false, 64 * wpb, per_wave_b * wpb + vg_lds);
            }
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
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !dev_ptr || !bytes) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    *dev_ptr = ctx->d_filter;
    *bytes = (size_t)ctx->id_end * 4;
    return LHGT_OK;
}

// test handle: the contig groups of the dense vote's bound as lhgt_ref_scan left them (first peak id of each; VG_N + 1 values), *valid = 0
// when the registered peaks have none (an installed registry, a vote bitmap in front)
int lhgt_vote_groups_export(lhgt_ctx* ctx, uint32_t* out, int n, int* valid) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out || n != VG_N + 1) LHGT_FAIL(LHGT_E_ARG, "room for %d values, the table has %d", n, VG_N + 1);
    if (valid) *valid = ctx->vote_groups_ok ? 1 : 0;
    if (!ctx->vote_groups_ok) return LHGT_OK;
    LHGT_HIP(hipMemcpyAsync(out, ctx->d_vote_groups, (size_t)(VG_N + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

int lhgt_peaks_export(lhgt_ctx* ctx, int32_t* loci, uint8_t* filter, long n) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    if (n > ctx->id_end) LHGT_FAIL(LHGT_E_ARG, "asked for %ld peak ids, have %ld", n, ctx->id_end);
    if (n == 0) return LHGT_OK;
    if (loci) LHGT_HIP(hipMemcpyAsync(loci, ctx->d_loci, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream)); LHGT_HIP(hipStreamSynchronize(ctx->stream));
    if (filter) {
        std::vector<uint32_t> v((size_t)n);
        LHGT_HIP(hipMemcpyAsync(v.data(), ctx->d_filter, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream)); LHGT_HIP(hipStreamSynchronize(ctx->stream));
        for (long i = 0; i < n; i++) filter[i] = (uint8_t)(v[i] > 254 ? 254 : v[i]);  // `if (< 254) ++` saturates at 254
    }
    return LHGT_OK;
}

// count_filtered_peak (E:515-548), single thread range: leading sentinel "1 1 1", merge while the
// contig is the same and the gap to the running end is < 500.  Only the voted peaks leave the GPU.
int lhgt_write_intervals(lhgt_ctx* ctx, const char* path, long* n_filtered) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !path) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->n_peaks < 0) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    const long n = ctx->id_end;
    std::vector<int32_t> rec;
    long nf = 0;
    if (n > 0) {
        long cap = ctx->voted_cap;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (!ctx->d_voted) {
                if (cap < 4096) cap = 4096;
                LHGT_HIP(lhgt::dev_malloc(&ctx->d_voted, (size_t)cap * 12 + 8));
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
            lhgt::dev_free(ctx->d_voted);      // more voted peaks than room: grow once and redo
            ctx->d_voted = nullptr;
            cap = (nf + nf / 8 + 1) & ~1L;   // even: the 64-bit counter sits behind cap 12-byte records and must be 8-byte aligned (round 4: an odd cap faulted on the first sample with more than 4096 voted peaks)
        }
        rec.resize((size_t)nf * 3);
        if (nf) LHGT_HIP(hipMemcpyAsync(rec.data(), ctx->d_voted, (size_t)nf * 12, hipMemcpyDeviceToHost, ctx->stream)); LHGT_HIP(hipStreamSynchronize(ctx->stream));
    }
    std::vector<long> order((size_t)nf);
    for (long i = 0; i < nf; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](long a, long b) { return rec[3 * a] < rec[3 * b]; });
    FILE* f = fopen(path, "w");
    if (!f) LHGT_FAIL(LHGT_E_IO, "cannot write %s", path);
    // one pass per thread id range (E:520-543): a single one normally; under -t N emulation thread j owns
    // [j * (max_peak / N), its last id), starts from the sentinel state again and writes its own last line
    const int n_ranges = ctx->emu_threads > 1 && !ctx->emu_range_end.empty() ? ctx->emu_threads : 1;
    long q = 0;
    for (int j = 0; j < n_ranges; j++) {
        const long id_hi = n_ranges > 1 ? ctx->emu_range_end[j] : n;
        int start = 1, end = 1, chr = 1;
        for (; q < nf && rec[3 * order[q]] < id_hi; q++) {
            const int c = rec[3 * order[q] + 1], pos = rec[3 * order[q] + 2];
            if (chr == c && pos - 500 - end < 500) end = pos + 500;
            else {
                fprintf(f, "%d\t%d\t%d\n", chr, start, end);
                chr = c; start = pos - 500; end = pos + 500;
            }
        }
        fprintf(f, "%d\t%d\t%d\n", chr, start, end);
    }
    fclose(f);
    if (n_filtered) *n_filtered = nf;
    return LHGT_OK;
}

}  // extern "C"
