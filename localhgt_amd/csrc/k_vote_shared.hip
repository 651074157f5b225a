// k_vote_shared.hip -- phase C for a DEEP sample on a dense peak set: the probes of reads that overlap share their line fills.
//
// Peaks::slide_reads (E:313-506) looks every (k-mer, hash) of every read up in peak_kmer; where the peak set is too dense for
// the on-chip filters (k_vote.hip: vote_kernel, the "dense" form) that is one 128-byte line fill of the fabric per probe -- 714 per
// pair, 71 G per 100 M pairs, 1.5 s at the fabric's rate -- although at 100x coverage every k-mer of the sample is probed by
// ~80 reads.  Neither cache sees that reuse: reads arrive in file order.  Here the sharing is explicit (round 6):
//
//   keys     every READ (both mates alike) gets a key: the smallest hash 0 over its k-mers -- the read's champion k-mer.  Two reads
//            share a champion with the probability of their Jaccard overlap, so the reads under one key all contain one locus.
//   group    the read ids are grouped by key (counting sort on a product of the key; the read store itself stays where it is).
//   probe    a workgroup takes the <= 64 reads of one bucket, enters every slot they probe into a hash set in LDS, fetches each
//            DISTINCT slot of peak_kmer once (and the contig of the id it holds, peak_loci), and answers the reads' probes from the set.
//            A pair votes only if TWO contigs collect six hit offsets each (check_split E:161-202), and judge_base credits every hit
//            offset to one of the contigs its hashes name (E:118-159): what a read keeps is therefore what an upper bound of "which
//            contigs can reach six" needs -- its hits and hit offsets, the contig most of its hits name (exactly counted), and the hits
//            on all other contigs in 128 hashed 4-bit counters (64 bytes).
//   filter   per PAIR (one lane): the two reads' records added up; a pair in which at most one contig can reach six is done.
//   vote     the pairs that are left -- a few per cent where the sample's reads sit on their own contigs -- are voted from scratch by
//            the generic kernel (vote_kernel on a pair list): the same hits in the same order through the same judge_pair, so the
//            shared form never computes a vote itself and cannot differ.
//
// (Two earlier forms of this round, measured and dropped -- DESIGN.md 4: per-read hit lists in an arena, judged per pair -- at the
// SNP leg's 125 true hits per read the arena wants 200 GB --; and the set walked hash by hash instead of six probes together.)
// Pairs the form does not take (a read of more than FAST_NK offsets) go to the generic kernel as well.  The engine picks the form when
// the dense form would run, e <= 3, the grouping finds at least LHGT_SHARED_MIN (3) reads per occupied bucket and the peak set is
// not so dense that foreign hits alone fill the counters; keys and order are kept while the read store does not change.
#include <algorithm>
#include <cstring>
#include "lhgt_hash.hpp"
#include "k_vote_judge.hpp"

namespace lhgt {

constexpr int VS_G = 64;                  // reads per work item at most (one bucket of the grouping, or a piece of a large one)
constexpr int VS_WAVES = 8;
constexpr int VS_SET = 2048;              // entries of the LDS set: a k-mer (8 bytes) and the contigs its e <= 3 slots name (12 bytes), 40 KiB -- the
constexpr int VS_SET_BITS = 11;           // <= 64 reads of a bucket hold ~240 distinct k-mers per champion locus, a few hundred more where keys share a bucket
constexpr int VS_PROBES = 8;
constexpr int VS_REC = 20;                // LDS words per staged record (three planes of <= 6 words: reads of <= FAST_NK offsets)
constexpr unsigned long long VS_EMPTY64 = ~0ull;      // no canonical k-mer: all ones is poly-T, whose reverse complement (poly-A, all zeros) is the smaller
constexpr uint32_t VS_NOKEY = 0xffffffffu;
// what a read leaves for the pair filter: 80 bytes
constexpr int VS_CB = 128;      // contig buckets of the filter (64: 5 % of the SNP leg's pairs passed, most of them on foreign hits that met in a bucket)
struct VsReadRec {
    unsigned long long sk[VS_CB / 64][4];   // bit b of sk[w][j] = bit j of bucket 64 w + b's count (saturating at 15) of the hits NOT on `star`
    uint32_t star;              // the contig most hits name (the contig of an offset whose e hashes agree; 0: no hit)
    uint32_t n_star;            // hits on it, exactly
    uint32_t n_hits;            // probes that found an id
    uint32_t n_ev;              // offsets with a hit
};
__device__ __forceinline__ uint32_t vs_contig_bucket(uint32_t chr) { return (chr * 0x9E3779B1u) >> 25; }     // 7 bits: VS_CB buckets

struct VsBatches {                         // the resident batches as one numbering of pairs
    const ReadBatchDev* d;
    const uint32_t* base;                  // [n + 1]: first global pair of every batch
    int n;
};
__device__ __forceinline__ int vs_batch_of(const VsBatches& t, uint32_t P) {
    int lo = 0, hi = t.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.base[mid] <= P) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// A read's key is the MINIMUM of its ~119 hashes: keys crowd towards zero (the smallest of n uniform values has mean 2^k / n), so
// the bucket of a key is taken from a product that spreads them again -- equal keys still meet, which is all the grouping needs.
__device__ __forceinline__ uint32_t vs_bucket(uint32_t key, int shift) { return (key * 0x9E3779B1u) >> shift; }

// the e hashes of offset s * 64 + lane of a read whose record (three planes of wpr words) sits at `rec`
struct VsRead { int nk, wpr; };
__device__ __forceinline__ VsRead vs_shape(int len, int k) { return VsRead{len - k + 1, ((len + 31) >> 5) + 1}; }
__device__ __forceinline__ bool vs_hashes(const uint32_t* rec, const VsRead& rd, int s, int lane, int k, int e, const HashParams& hp, uint32_t h[3]) {
    const int j = s * 64 + lane, r = j & 31;
    const uint32_t* q = rec + (j < rd.nk ? (j >> 5) : 0);
    const int wp = rd.wpr;
    auto win = [&](uint32_t a, uint32_t c) { return window32(a, c, r) >> (32 - k); };
    const uint32_t whi = win(q[0], q[1]), wlo = win(q[wp], q[wp + 1]), wnb = win(q[2 * wp], q[2 * wp + 1]);
    const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
#pragma unroll
    for (int i = 0; i < 3; i++) h[i] = i < e ? hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]) : 0u;
    return j < rd.nk && wnb == 0;
}

// the k-mer at offset s * 64 + lane as ONE key: the smaller of its 2-bit code (hi plane << 32 | lo plane) and its reverse
// complement's.  A k-mer and its reverse complement have the same e hashes (each hash is min(forward, reverse), E:447-452), so the
// set can hold k-mers instead of slots: one entry, one compare-and-swap and one lookup per k-mer instead of e, and the hashes are
// only computed once per DISTINCT k-mer of a work item (from the key: vs_key_hash).
__device__ __forceinline__ bool vs_kmer(const uint32_t* rec, const VsRead& rd, int s, int lane, int k, unsigned long long* key) {
    const int j = s * 64 + lane, r = j & 31;
    const uint32_t* q = rec + (j < rd.nk ? (j >> 5) : 0);
    const int wp = rd.wpr;
    auto win = [&](uint32_t a, uint32_t c) { return window32(a, c, r) >> (32 - k); };
    const uint32_t whi = win(q[0], q[1]), wlo = win(q[wp], q[wp + 1]), wnb = win(q[2 * wp], q[2 * wp + 1]);
    const uint32_t kmask = k >= 32 ? 0xffffffffu : (1u << k) - 1u;
    const uint32_t chi = ~brev_k(whi, k) & kmask, clo = ~brev_k(wlo, k) & kmask;
    const unsigned long long f = ((unsigned long long)whi << 32) | wlo, c = ((unsigned long long)chi << 32) | clo;
    *key = f < c ? f : c;
    return j < rd.nk && wnb == 0;
}
__device__ __forceinline__ uint32_t vs_key_hash(unsigned long long key, int k, const uint32_t* __restrict__ m) {
    const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
    return hash_from_windows(hi, lo, brev_k(hi, k), brev_k(lo, k), m);
}

// ---------------------------------------------------------------- keys
// One wave per pair: the champion of either mate (the smallest hash 0 over its valid k-mers), a count per key bucket, and a
// descriptor per read -- where its record lies and how long it is -- so that the probe kernel reaches a record in one step.  Pairs
// that are not voted (flags), reads without a valid k-mer: no key -- they have no events.  A pair with a read beyond FAST_NK
// offsets: no key, listed for the generic kernel.
__global__ void __launch_bounds__(256) vs_read_keys(VsBatches t, HashParams hp, uint32_t n_pairs, int shift, uint32_t* __restrict__ keys,
                                                    unsigned long long* __restrict__ desc, uint32_t* __restrict__ hist, uint32_t* __restrict__ fallback) {
    __shared__ uint32_t stage_all[4 * 64];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    uint32_t* stage = stage_all + wib * 64;
    const int k = hp.k;
    const long n_waves = (long)gridDim.x * 4;
    for (long P = (long)blockIdx.x * 4 + wib; P < (long)n_pairs; P += n_waves) {
        const int bi = __builtin_amdgcn_readfirstlane(vs_batch_of(t, (uint32_t)P));
        const ReadBatchDev& b = t.d[bi];
        const long p = P - (long)t.base[bi];
        uint32_t key[2] = {VS_NOKEY, VS_NOKEY};
        const bool voted = !b.flags || (b.flags[p] & PAIR_VOTE);
        const int len[2] = {b.len[0][p], b.len[1][p]};
        const uint32_t* rec[2] = {b.words + b.off[0][p], b.words + b.off[1][p]};
        if (voted && (len[0] - k + 1 > FAST_NK || len[1] - k + 1 > FAST_NK)) {
            if (lane == 0) fallback[1u + atomicAdd(fallback, 1u)] = (uint32_t)P;
        } else if (voted) {
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const VsRead rd = vs_shape(len[m], k);
                const uint32_t w = rec[m][lane < 3 * rd.wpr ? lane : 3 * rd.wpr - 1];
                __builtin_amdgcn_wave_barrier();
                stage[lane] = w;
                __builtin_amdgcn_wave_barrier();
                uint32_t best = VS_NOKEY;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    uint32_t h[3];
                    const bool ok = vs_hashes(stage, rd, s, lane, k, 1, hp, h);
                    if (ok && h[0] < best) best = h[0];
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) { const uint32_t o = __shfl_xor(best, d, 64); best = o < best ? o : best; }
                key[m] = best;
            }
        }
        if (lane < 2) {
            const uint32_t kk = lane ? key[1] : key[0];
            keys[2 * P + lane] = kk;
            desc[2 * P + lane] = (unsigned long long)(uintptr_t)(lane ? rec[1] : rec[0]) | ((unsigned long long)(uint32_t)(lane ? len[1] : len[0]) << 48);
            if (kk != VS_NOKEY) atomicAdd(&hist[vs_bucket(kk, shift)], 1u);
        }
    }
}

// exclusive scan of the bucket counts in three steps (8192 counters per workgroup), with the number of occupied buckets on the side
constexpr int VS_SCAN = 8192;
__global__ void __launch_bounds__(1024) vs_scan_sums(const uint32_t* __restrict__ v, long n, uint32_t* __restrict__ sums, unsigned long long* __restrict__ occupied) {
    __shared__ uint32_t red[16], red2[16];
    const long base = (long)blockIdx.x * VS_SCAN;
    uint32_t acc = 0, occ = 0;
    for (int i = threadIdx.x; i < VS_SCAN; i += 1024)
        if (base + i < n) { const uint32_t x = v[base + i]; acc += x; occ += x != 0u; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); occ += __shfl_xor(occ, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; red2[threadIdx.x >> 6] = occ; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, o = 0;
        for (int w = 0; w < 16; w++) { a += red[w]; o += red2[w]; }
        sums[blockIdx.x] = a;
        atomicAdd(occupied, (unsigned long long)o);
    }
}
__global__ void __launch_bounds__(1024) vs_scan_bases(uint32_t* __restrict__ sums, long n_blocks, uint32_t* __restrict__ total) {
    __shared__ uint32_t part[1024];
    // n_blocks <= 2^24 / 8192 = 2048: two entries per thread
    const long per = (n_blocks + 1023) / 1024;
    uint32_t acc = 0;
    for (long i = threadIdx.x * per; i < (threadIdx.x + 1) * per && i < n_blocks; i++) acc += sums[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; i++) { const uint32_t x = part[i]; part[i] = run; run += x; }
        *total = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (long i = threadIdx.x * per; i < (threadIdx.x + 1) * per && i < n_blocks; i++) { const uint32_t x = sums[i]; sums[i] = run; run += x; }
}
__global__ void __launch_bounds__(1024) vs_scan_apply(uint32_t* __restrict__ v, long n, const uint32_t* __restrict__ sums, uint32_t* __restrict__ copy) {
    __shared__ uint32_t part[1024];
    const long base = (long)blockIdx.x * VS_SCAN;
    uint32_t x[VS_SCAN / 1024];
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < VS_SCAN / 1024; i++) {
        const long at = base + (long)threadIdx.x * (VS_SCAN / 1024) + i;
        x[i] = at < n ? v[at] : 0u;
        acc += x[i];
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 64) {     // one wave scans the 1024 partial sums, 16 per lane
        uint32_t loc[16], s = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { loc[i] = part[threadIdx.x * 16 + i]; s += loc[i]; }
        uint32_t incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)threadIdx.x >= d) incl += o; }
        uint32_t run = incl - s;
#pragma unroll
        for (int i = 0; i < 16; i++) { part[threadIdx.x * 16 + i] = run; run += loc[i]; }
    }
    __syncthreads();
    uint32_t run = sums[blockIdx.x] + part[threadIdx.x];
#pragma unroll
    for (int i = 0; i < VS_SCAN / 1024; i++) {
        const long at = base + (long)threadIdx.x * (VS_SCAN / 1024) + i;
        if (at < n) { v[at] = run; copy[at] = run; }     // v keeps the buckets' starts, copy becomes the scatter's cursors
        run += x[i];
    }
}
__global__ void __launch_bounds__(256) vs_scatter(const uint32_t* __restrict__ keys, uint32_t n_reads, int shift, uint32_t* __restrict__ cursor,
                                                  uint32_t* __restrict__ order) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= (long)n_reads) return;
    const uint32_t kk = keys[r];
    if (kk == VS_NOKEY) return;
    order[atomicAdd(&cursor[vs_bucket(kk, shift)], 1u)] = (uint32_t)r;
}
// the work items of the probe kernel: every occupied bucket, in pieces of at most VS_G reads
__global__ void __launch_bounds__(256) vs_make_items(const uint32_t* __restrict__ start, long n_buckets, const uint32_t* __restrict__ total,
                                                     uint2* __restrict__ items, uint32_t* __restrict__ n_items) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_buckets) return;
    const uint32_t s = start[b], e = b + 1 < n_buckets ? start[b + 1] : *total;
    if (e == s) return;
    const uint32_t pieces = (e - s + VS_G - 1) / VS_G;
    const uint32_t at = atomicAdd(n_items, pieces);
    for (uint32_t q = 0; q < pieces; q++) items[at + q] = make_uint2(s + q * VS_G, e - s - q * VS_G < (uint32_t)VS_G ? e - s - q * VS_G : (uint32_t)VS_G);
}

// ---------------------------------------------------------------- probe
__device__ __forceinline__ uint32_t vs_slot(unsigned long long key) {
    return (((uint32_t)key ^ ((uint32_t)(key >> 32) * 0x85EBCA6Bu)) * 0x9E3779B1u) >> (32 - VS_SET_BITS);
}
// The k-mers of a lane (two slices of 64 offsets) walk the set TOGETHER, in straight-line selects: one LDS round trip serves both
// (rounds 6's earlier forms: the probes one after the other were the kernel's time -- 127 us per work item, 6 x the estimate --, and
// written with `if`s every probe became an exec region with its own copies of the slot registers, 250 instructions per round:
// profiles/r06).  A probe is one 64-bit compare-and-swap: it finds the entry empty (and takes it), or holding its own k-mer, or moves
// on; a probe that is done plays its compare-and-swap on the lane's dummy word, which never matches.  where[i] = the entry a k-mer
// ended at, VS_OUT if the set had no room for it within VS_PROBES entries (it then asks peak_kmer itself), VS_DEAD for an offset
// beyond the read or a k-mer with an N.
constexpr uint32_t VS_OUT = 0xfffeu, VS_DEAD = 0xffffu;
template <int N>
__device__ __forceinline__ void vs_insert_n(unsigned long long* keys, unsigned long long* dummy /* one per lane, never VS_EMPTY64 */,
                                            const unsigned long long (&key)[N], uint32_t live, uint32_t (&where)[N]) {
    uint32_t s[N], todo[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        s[i] = vs_slot(key[i]);
        todo[i] = (live >> i) & 1u;
        where[i] = todo[i] ? VS_OUT : VS_DEAD;
    }
#pragma unroll 1
    for (int t = 0; t < VS_PROBES; t++) {
        unsigned long long old[N];
        uint32_t left = 0u;
#pragma unroll
        for (int i = 0; i < N; i++) old[i] = atomicCAS(todo[i] ? &keys[s[i]] : dummy, VS_EMPTY64, key[i]);
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t fin = todo[i] & (uint32_t)((old[i] == VS_EMPTY64) | (old[i] == key[i]));
            where[i] = fin ? s[i] : where[i];
            todo[i] &= ~fin;
            s[i] = todo[i] ? ((s[i] + 1) & (VS_SET - 1)) : s[i];
            left |= todo[i];
        }
        if (!__any(left != 0u)) break;
    }
}

// One work item per workgroup: <= 64 reads of one bucket.  Their descriptors, then their records, come into LDS once; every slot
// they probe enters the set; every distinct slot is fetched once -- the id, and for an id the contig of its peak --; the reads'
// probes are answered from the set and summed up into the read's record for the pair filter.
__global__ void __launch_bounds__(64 * VS_WAVES, 6) vs_probe(HashParams hp, const uint2* __restrict__ items, uint32_t n_items, const uint32_t* __restrict__ order,
                                                          const unsigned long long* __restrict__ desc, const uint32_t* __restrict__ peak_kmer,
                                                          const int32_t* __restrict__ loci, VsReadRec* __restrict__ read_rec,
                                                          unsigned long long* __restrict__ stats /* nullable: [0] slots fetched for the sets, [1] probes answered outside them */,
                                                          int ablate /* stage timing (LHGT_VS_ABLATE; outputs wrong): 1 no inserts, 2 no fetches, 4 no lookups, 8 no records, 16 nothing after the reads' records */) {
    __shared__ __align__(16) unsigned long long skey[VS_SET];   // the set's k-mers
    __shared__ uint32_t sval[3 * VS_SET];            // per entry: the contig each of its e slots names (0: the slot holds no id)
    __shared__ uint32_t recs[VS_G * VS_REC];
    __shared__ unsigned long long rdesc[VS_G];
    __shared__ uint32_t rid[VS_G + 2];              // [VS_G], [VS_G + 1]: the item's sums for lhgt_work_stats
    __shared__ __align__(8) unsigned long long dummy[64 * VS_WAVES];   // one word per lane for the compare-and-swaps and counts that do not count (never all ones)
    __shared__ uint32_t counters[VS_WAVES * 64];     // per wave: the VS_CB = 128 contig buckets of the read at hand, 16 bits each (bucket b in half b >> 6 of word b & 63; a read has <= 384 hits)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int k = hp.k, e = hp.e;
    const long item = block2d();
    if (item >= (long)n_items) return;            // (the whole workgroup: a 2-D grid is rounded up)
    const uint2 it = items[item];
    const int n_here = (int)it.y;
    if ((int)threadIdx.x < n_here) {
        const uint32_t r = order[it.x + threadIdx.x];
        rid[threadIdx.x] = r;
        rdesc[threadIdx.x] = desc[r];
    }
    for (int i = threadIdx.x; i < VS_SET; i += 64 * VS_WAVES) skey[i] = VS_EMPTY64;
    dummy[threadIdx.x] = 0ull;
    if (threadIdx.x < 2) rid[VS_G + threadIdx.x] = 0u;
    __syncthreads();
    for (int idx = threadIdx.x; idx < n_here * VS_REC; idx += 64 * VS_WAVES) {
        const int q = idx / VS_REC, w = idx - q * VS_REC;
        const unsigned long long d = rdesc[q];
        const int w3 = 3 * ((((int)(d >> 48) + 31) >> 5) + 1);
        recs[idx] = w < w3 ? reinterpret_cast<const uint32_t*>((uintptr_t)(d & 0xffffffffffffull))[w] : 0u;
    }
    __syncthreads();
    if (ablate & 16) return;
    // 1. every k-mer of the item's reads enters the set; a lane remembers WHERE each of its two k-mers ended (16 bits each, one
    //    register per read, a wave has at most VS_G / VS_WAVES = 8 reads) so that step 3 neither builds nor searches them again
    constexpr int RPW = VS_G / VS_WAVES;
    uint32_t wh[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        wh[rr] = VS_DEAD | (VS_DEAD << 16);
        const int q = wib + rr * VS_WAVES;
        if (q < n_here && !(ablate & 1)) {
            const VsRead rd = vs_shape((int)(rdesc[q] >> 48), k);
            unsigned long long key[2];
            uint32_t where[2], live = 0u;
#pragma unroll
            for (int s = 0; s < 2; s++)
                if (vs_kmer(recs + q * VS_REC, rd, s, lane, k, &key[s])) live |= 1u << s;
            vs_insert_n<2>(skey, dummy + threadIdx.x, key, live, where);
            wh[rr] = where[0] | (where[1] << 16);
        }
    }
    __syncthreads();
    // 2. per distinct k-mer: its e hashes (from the key), one fetch per slot, all of a thread's in flight together; the set's values
    //    become the CONTIG of the id each slot holds (count_peak_kmer's peak_chr, E:455; contigs number from 1), 0 where it holds none
    {
        constexpr int U = VS_SET / (64 * VS_WAVES);
        int n_mine = 0;
        uint32_t val[U][3];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long key = skey[u * 64 * VS_WAVES + threadIdx.x];
#pragma unroll
            for (int i = 0; i < 3; i++) {
                val[u][i] = 0u;
                if (key != VS_EMPTY64 && i < e && !(ablate & 2)) { val[u][i] = __builtin_nontemporal_load(peak_kmer + vs_key_hash(key, k, hp.mask[i])); n_mine++; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (val[u][i]) val[u][i] = (uint32_t)loci[2 * (long)val[u][i]];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 3; i++) sval[3 * (u * 64 * VS_WAVES + threadIdx.x) + i] = val[u][i];
        if (stats) {       // (one global atomic per workgroup: 48 M waves adding to one address took 300 ms)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) n_mine += __shfl_xor(n_mine, d, 64);
            if (lane == 0) atomicAdd(&rid[VS_G], (uint32_t)n_mine);
        }
    }
    __syncthreads();
    if (stats && threadIdx.x == 0) atomicAdd(stats, (unsigned long long)rid[VS_G]);
    // 3. the reads again: the contigs of every k-mer's slots from where step 1 left it, summed up per read
    uint32_t* cnt = counters + wib * 64;
    uint32_t* my_dummy = reinterpret_cast<uint32_t*>(dummy + threadIdx.x);       // its low half (the high half stays zero)
    unsigned long long st_out = 0;
#pragma unroll
    for (int rr = 0; rr < RPW; rr++) {
        const int q = wib + rr * VS_WAVES;
        if (q >= n_here || (ablate & 4)) continue;
        const uint32_t r = rid[q];
        uint32_t chr[6], outside = 0u;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint32_t w = (wh[rr] >> (s * 16)) & 0xffffu;
            const uint32_t at = 3u * (w < (uint32_t)VS_SET ? w : 0u);          // unconditional loads, masked afterwards
#pragma unroll
            for (int i = 0; i < 3; i++) chr[3 * s + i] = (w < (uint32_t)VS_SET && i < e) ? sval[at + i] : 0u;
            if (w == VS_OUT) outside |= 1u << s;
        }
        if (__any(outside != 0u)) {       // the set had no room for this k-mer: its key again, peak_kmer itself, then the contig of the id
            const VsRead rd = vs_shape((int)(rdesc[q] >> 48), k);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                unsigned long long key;
                vs_kmer(recs + q * VS_REC, rd, s, lane, k, &key);
                if ((outside >> s) & 1u) {
#pragma unroll
                    for (int i = 0; i < 3; i++)
                        if (i < e) {
                            const uint32_t id = __builtin_nontemporal_load(peak_kmer + vs_key_hash(key, k, hp.mask[i]));
                            chr[3 * s + i] = id ? (uint32_t)loci[2 * (long)id] : 0u;
                        }
                }
            }
            if (stats) st_out += (unsigned long long)(__popcll(__ballot(outside & 1u)) + __popcll(__ballot(outside & 2u))) * (unsigned long long)e;
        }
        if (ablate & 8) continue;
        const unsigned long long any0 = __ballot((chr[0] | chr[1] | chr[2]) != 0u), any1 = __ballot((chr[3] | chr[4] | chr[5]) != 0u);
        const int n_ev = __popcll(any0) + __popcll(any1);
        if (n_ev == 0) continue;                    // the read's record stays zero
        // the star: where the read sits, its hits on its own contig are the many and foreign ones the few -- and at an offset of its
        // own contig ALL hashes hit and name that contig, which foreign hits (independent collisions) practically never do.  So: the
        // contig of an offset (the one nearest the read's middle) whose e hashes agree; without one, the contig of any hit.
        const bool ag0 = chr[0] != 0u && (e < 2 || chr[1] == chr[0]) && (e < 3 || chr[2] == chr[0]);
        const bool ag1 = chr[3] != 0u && (e < 2 || chr[4] == chr[3]) && (e < 3 || chr[5] == chr[3]);
        const unsigned long long agree1 = __ballot(ag1), agree0 = __ballot(ag0);
        uint32_t star;
        if (agree1) star = (uint32_t)__builtin_amdgcn_readlane((int)chr[3], __ffsll((long long)agree1) - 1);
        else if (agree0) star = (uint32_t)__builtin_amdgcn_readlane((int)chr[0], 63 - __clzll((long long)agree0));
        else {
            const unsigned long long pick = any1 ? any1 : any0;
            const uint32_t c0 = any1 ? (chr[3] ? chr[3] : chr[4] ? chr[4] : chr[5]) : (chr[0] ? chr[0] : chr[1] ? chr[1] : chr[2]);
            star = (uint32_t)__builtin_amdgcn_readlane((int)c0, __ffsll((long long)pick) - 1);
        }
        cnt[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        int mine_hits = 0, mine_star = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) {          // (no branch per probe: what does not count adds to the lane's dummy word)
            const bool hit = chr[i] != 0u, st = chr[i] == star;
            mine_hits += hit;
            mine_star += hit & st;
            const uint32_t bk = vs_contig_bucket(chr[i]);
            atomicAdd(hit && !st ? &cnt[bk & 63u] : my_dummy, 1u << ((bk >> 6) * 16u));
        }
        int both = mine_hits | (mine_star << 16);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) both += __shfl_xor(both, d, 64);
        __builtin_amdgcn_wave_barrier();
        unsigned long long sl[VS_CB / 64][4];
#pragma unroll
        for (int w = 0; w < VS_CB / 64; w++) {
            const uint32_t cw = (cnt[lane] >> (16 * w)) & 0xffffu, c = cw < 15u ? cw : 15u;
            sl[w][0] = __ballot(c & 1u); sl[w][1] = __ballot(c & 2u); sl[w][2] = __ballot(c & 4u); sl[w][3] = __ballot(c & 8u);
        }
        if (lane == 0) {
            VsReadRec o;
#pragma unroll
            for (int w = 0; w < VS_CB / 64; w++)
#pragma unroll
                for (int j = 0; j < 4; j++) o.sk[w][j] = sl[w][j];
            o.star = star;
            o.n_star = (uint32_t)(both >> 16);
            o.n_hits = (uint32_t)(both & 0xffff);
            o.n_ev = (uint32_t)n_ev;
            read_rec[r] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (stats) {
        if (lane == 0 && st_out) atomicAdd(&rid[VS_G + 1], (uint32_t)st_out);
        __syncthreads();
        if (threadIdx.x == 0 && rid[VS_G + 1]) atomicAdd(stats + 1, (unsigned long long)rid[VS_G + 1]);
    }
}

// ---------------------------------------------------------------- filter
// One lane per pair: can two different contigs collect six hit offsets each?  Upper bounds only: a contig's count is at most the
// hits that name it (judge_base credits an offset to ONE contig, E:118-159).  Known per read: the hits on its star contig, exactly;
// the hits on all other contigs by hashed bucket (VS_CB = 128 of them; saturated at 15, which reads as "enough").  Per bucket b: O_b = the two reads'
// other-hits added, and the stars that hash there with their exact counts; the most contigs of bucket b that can reach six:
//   two  if O_b >= 12 (two others), or a star s with max(0, 6 - s) + 6 <= O_b (it and an other), or two different stars s1, s2 with
//        max(0, 6 - s1) + max(0, 6 - s2) <= O_b
//   one  if O_b >= 6 or some star s with s + O_b >= 6
// (a read's hits on the OTHER read's star sit in its O_b of that star's bucket: adding O_b to the star covers them).  The pair can
// vote only if the buckets add up to two or more; it then goes on the list for the generic kernel.
__device__ __forceinline__ uint32_t vs_sliced_at(const unsigned long long (&w)[5], uint32_t b) {
    return (uint32_t)((w[0] >> b) & 1ull) | (uint32_t)((w[1] >> b) & 1ull) << 1 | (uint32_t)((w[2] >> b) & 1ull) << 2 | (uint32_t)((w[3] >> b) & 1ull) << 3 |
           (uint32_t)((w[4] >> b) & 1ull) << 4;
}
__global__ void vs_count_list(const uint32_t* __restrict__ list, unsigned long long* __restrict__ out) { atomicAdd(out, (unsigned long long)list[0]); }
__global__ void __launch_bounds__(256) vs_filter(uint32_t n_pairs, const VsReadRec* __restrict__ read_rec, uint32_t* __restrict__ list, int debug) {
    const long P = (long)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (P < (long)n_pairs) {
        const VsReadRec a = read_rec[2 * P], b = read_rec[2 * P + 1];
        if (a.n_hits + b.n_hits >= 12u && a.n_ev + b.n_ev >= 6u) {
            const uint32_t ba = vs_contig_bucket(a.star), bb = vs_contig_bucket(b.star);
            const bool has_a = a.n_star > 0u, has_b = b.n_star > 0u && !(has_a && a.star == b.star);
            const uint32_t sa = a.n_star + (has_a && b.n_star > 0u && a.star == b.star ? b.n_star : 0u), sb = b.n_star;
            int total = 0;
            uint32_t Oa = 0u, Ob = 0u;           // the other-hits of the stars' buckets (64: saturated)
#pragma unroll
            for (int w = 0; w < VS_CB / 64; w++) {
                // bit-sliced sum of 64 pairs of 4-bit counters: five planes
                unsigned long long o[5], carry = 0ull;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const unsigned long long x = a.sk[w][j], y = b.sk[w][j];
                    o[j] = x ^ y ^ carry;
                    carry = (x & y) | (carry & (x ^ y));
                }
                o[4] = carry;
                // a saturated counter (15) stands for "15 or more": read it as large
                const unsigned long long sat = (a.sk[w][0] & a.sk[w][1] & a.sk[w][2] & a.sk[w][3]) | (b.sk[w][0] & b.sk[w][1] & b.sk[w][2] & b.sk[w][3]);
                // >= 6: bit 4, or bit 3, or bits 2 and 1;  >= 12: bit 4, or bits 3 and 2
                const unsigned long long ge6 = o[4] | o[3] | (o[2] & o[1]) | sat, ge12 = o[4] | (o[3] & o[2]) | sat;
                unsigned long long star_mask = 0ull;
                if (has_a && (int)(ba >> 6) == w) { star_mask |= 1ull << (ba & 63u); Oa = ((sat >> (ba & 63u)) & 1ull) ? 64u : vs_sliced_at(o, ba & 63u); }
                if (has_b && (int)(bb >> 6) == w) { star_mask |= 1ull << (bb & 63u); Ob = ((sat >> (bb & 63u)) & 1ull) ? 64u : vs_sliced_at(o, bb & 63u); }
                total += __popcll(ge6 & ~star_mask) + __popcll(ge12 & ~star_mask);          // buckets without a star: one contig, or two
            }
            auto need = [](uint32_t s) { return s >= 6u ? 0u : 6u - s; };
            auto bucket_with = [&](uint32_t O, uint32_t s1, bool two, uint32_t s2) {
                if (O >= 12u || need(s1) + 6u <= O || (two && (need(s2) + 6u <= O || need(s1) + need(s2) <= O))) return 2;
                if (O >= 6u || s1 + O >= 6u || (two && s2 + O >= 6u)) return 1;
                return 0;
            };
            if (has_a && has_b && ba == bb) total += bucket_with(Oa, sa, true, sb);
            else {
                if (has_a) total += bucket_with(Oa, sa, false, 0u);
                if (has_b) total += bucket_with(Ob, sb, false, 0u);
            }
            keep = total >= 2 || (debug & (1 << 19));
        }
    }
    // the wave's kept pairs in one append
    const unsigned long long bal = __ballot(keep);
    if (bal) {
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(list, (uint32_t)__popcll(bal));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (keep) list[1u + base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint32_t)P;
    }
}

}  // namespace lhgt

using namespace lhgt;

// what the shared form keeps between votes of the same read store
struct lhgt_vshared {
    unsigned long long store_gen = ~0ull;
    int k = 0;
    uint32_t mask0[3] = {0, 0, 0};            // the keys are hashes under this coder
    long n_pairs = 0;
    uint32_t n_listed = 0, n_items = 0;
    double reads_per_bucket = 0.0;
    ReadBatchDev* d_batches = nullptr;
    uint32_t* d_base = nullptr;
    int n_batches_cap = 0;
    uint32_t* d_keys = nullptr;                // [2 * n_pairs]
    unsigned long long* d_desc = nullptr;      // [2 * n_pairs]: record address | length << 48
    uint32_t* d_order = nullptr;               // [2 * n_pairs]
    uint2* d_items = nullptr;                  // [2 * n_pairs]: (first entry of order, reads) of every work item
    uint32_t* d_hist = nullptr;                // [n_buckets + sums]: the buckets' starts
    uint32_t* d_cursor = nullptr;              // [n_buckets]
    VsReadRec* d_read_rec = nullptr;           // [2 * n_pairs]
    uint32_t* d_list = nullptr;                // [1 + n_pairs]: the pairs for the generic kernel ([0] = how many)
    uint32_t* d_list_keys = nullptr;           // the pairs the key pass listed (long reads): kept with the order
    uint32_t n_list_keys = 0;
    uint32_t* d_small = nullptr;               // [1] listed reads, [2] work items, [4..5] occupied buckets (u64)
    long cap_pairs = 0;
    float keys_ms = 0.f;
};

namespace lhgt {
void vshared_free(lhgt_ctx* ctx) {
    lhgt_vshared* v = (lhgt_vshared*)ctx->vshared;
    if (!v) return;
    for (void* p : {(void*)v->d_batches, (void*)v->d_base, (void*)v->d_keys, (void*)v->d_desc, (void*)v->d_order, (void*)v->d_items, (void*)v->d_hist, (void*)v->d_cursor,
                    (void*)v->d_read_rec, (void*)v->d_list, (void*)v->d_list_keys, (void*)v->d_small})
        if (p) (void)dev_free(p);
    delete v;
    ctx->vshared = nullptr;
}
}  // namespace lhgt

static const int VS_BUCKET_BITS_MAX = 24;

// keys, grouping and the decision; *use = the shared form pays on this store
static int vshared_prepare(lhgt_ctx* ctx, bool* use) {
    *use = false;
    static const int mode = getenv("LHGT_SHARED_VOTE") ? atoi(getenv("LHGT_SHARED_VOTE")) : -1;     // 0 never, 1 whenever the form applies, default by the grouping
    const bool forced = mode == 1 || (ctx->debug & (1 << 27));
    if (mode == 0 || (ctx->debug & (1 << 28))) return LHGT_OK;
    if (ctx->e > 3 || ctx->n_pairs < 1 || ctx->n_pairs >= (1L << 31) || ctx->batches.empty()) return LHGT_OK;
    if (!forced && ctx->n_pairs < (1L << 20)) return LHGT_OK;          // small stores: nothing to win
    // a peak set so dense that foreign hits alone fill the filter's counters (a pair's ~714 probes x the share of slots with an id, over
    // 64 buckets: six per bucket at 54 %) leaves the filter nothing to decide: beyond a quarter of the slots the dense form runs as it is
    const double density = std::min(1.0, (double)ctx->n_selected * ctx->e / (double)(1ull << ctx->k));
    if (!forced && density > 0.25) return LHGT_OK;
    lhgt_vshared* v = (lhgt_vshared*)ctx->vshared;
    if (!v) { v = new lhgt_vshared(); ctx->vshared = v; }
    const long np = ctx->n_pairs;
    const bool fresh = v->store_gen == ctx->store_gen && v->k == ctx->k && memcmp(v->mask0, ctx->hp.mask[0], sizeof v->mask0) == 0 && v->n_pairs == np;
    const int nb_bits = std::min(ctx->k, VS_BUCKET_BITS_MAX);
    const long n_buckets = 1L << nb_bits;
    const int shift = 32 - nb_bits;                      // vs_bucket: the top nb_bits of a 32-bit product
    if (!fresh) {
        hipEvent_t t0 = ctx->ev2, t1 = ctx->ev3;
        LHGT_HIP(hipEventRecord(t0, ctx->stream));
        const int nb = (int)ctx->batches.size();
        if (nb > v->n_batches_cap) {
            if (v->d_batches) dev_free(v->d_batches);
            if (v->d_base) dev_free(v->d_base);
            v->d_batches = nullptr; v->d_base = nullptr;
            v->n_batches_cap = nb + nb / 2 + 8;
            LHGT_HIP(dev_malloc(&v->d_batches, (size_t)v->n_batches_cap * sizeof(ReadBatchDev)));
            LHGT_HIP(dev_malloc(&v->d_base, (size_t)(v->n_batches_cap + 1) * 4));
        }
        std::vector<ReadBatchDev> hb((size_t)nb);
        std::vector<uint32_t> base((size_t)nb + 1, 0u);
        for (int i = 0; i < nb; i++) { hb[(size_t)i] = ctx->batches[(size_t)i].d; base[(size_t)i + 1] = base[(size_t)i] + (uint32_t)ctx->batches[(size_t)i].d.n_pairs; }
        LHGT_HIP(hipMemcpyAsync(v->d_batches, hb.data(), hb.size() * sizeof(ReadBatchDev), hipMemcpyHostToDevice, ctx->stream));
        LHGT_HIP(hipMemcpyAsync(v->d_base, base.data(), base.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));      // the host vectors go out of scope
        if (np > v->cap_pairs) {
            for (void* p : {(void*)v->d_keys, (void*)v->d_desc, (void*)v->d_order, (void*)v->d_items, (void*)v->d_read_rec, (void*)v->d_list, (void*)v->d_list_keys}) if (p) dev_free(p);
            v->d_keys = v->d_order = nullptr; v->d_desc = nullptr; v->d_items = nullptr; v->d_read_rec = nullptr; v->d_list = v->d_list_keys = nullptr;
            v->cap_pairs = np + np / 16;
            LHGT_HIP(dev_malloc(&v->d_keys, (size_t)v->cap_pairs * 8));
            LHGT_HIP(dev_malloc(&v->d_desc, (size_t)v->cap_pairs * 16));
            LHGT_HIP(dev_malloc(&v->d_order, (size_t)v->cap_pairs * 8));
            LHGT_HIP(dev_malloc(&v->d_items, (size_t)v->cap_pairs * 16));
            LHGT_HIP(dev_malloc(&v->d_read_rec, (size_t)v->cap_pairs * 2 * sizeof(VsReadRec)));
            LHGT_HIP(dev_malloc(&v->d_list, (size_t)(v->cap_pairs + 1) * 4));
            LHGT_HIP(dev_malloc(&v->d_list_keys, (size_t)(v->cap_pairs + 1) * 4));
        }
        const long n_sum_blocks = (n_buckets + VS_SCAN - 1) / VS_SCAN;
        if (!v->d_hist) LHGT_HIP(dev_malloc(&v->d_hist, (size_t)((1L << VS_BUCKET_BITS_MAX) + 4096) * 4));
        if (!v->d_cursor) LHGT_HIP(dev_malloc(&v->d_cursor, (size_t)(1L << VS_BUCKET_BITS_MAX) * 4));
        if (!v->d_small) LHGT_HIP(dev_malloc(&v->d_small, 64));
        uint32_t* d_sums = v->d_hist + n_buckets;
        LHGT_HIP(hipMemsetAsync(v->d_hist, 0, (size_t)n_buckets * 4, ctx->stream));
        LHGT_HIP(hipMemsetAsync(v->d_small, 0, 64, ctx->stream));
        LHGT_HIP(hipMemsetAsync(v->d_list_keys, 0, 4, ctx->stream));
        VsBatches t{v->d_batches, v->d_base, nb};
        long blocks = (np + 3) / 4;
        if (blocks > 256L * 32) blocks = 256L * 32;
        hipLaunchKernelGGL(vs_read_keys, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, ctx->hp, (uint32_t)np, shift, v->d_keys, v->d_desc, v->d_hist, v->d_list_keys);
        unsigned long long* d_occ = (unsigned long long*)(v->d_small + 4);
        hipLaunchKernelGGL(vs_scan_sums, dim3((unsigned)n_sum_blocks), dim3(1024), 0, ctx->stream, v->d_hist, n_buckets, d_sums, d_occ);
        hipLaunchKernelGGL(vs_scan_bases, dim3(1), dim3(1024), 0, ctx->stream, d_sums, n_sum_blocks, v->d_small + 1);
        hipLaunchKernelGGL(vs_scan_apply, dim3((unsigned)n_sum_blocks), dim3(1024), 0, ctx->stream, v->d_hist, n_buckets, d_sums, v->d_cursor);
        hipLaunchKernelGGL(vs_scatter, dim3((unsigned)((2 * np + 255) / 256)), dim3(256), 0, ctx->stream, v->d_keys, (uint32_t)(2 * np), shift, v->d_cursor, v->d_order);
        hipLaunchKernelGGL(vs_make_items, dim3((unsigned)((n_buckets + 255) / 256)), dim3(256), 0, ctx->stream, v->d_hist, n_buckets, v->d_small + 1, v->d_items, v->d_small + 2);
        LHGT_HIP(hipGetLastError());
        uint32_t small[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        LHGT_HIP(hipMemcpyAsync(small, v->d_small, 32, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipMemcpyAsync(&v->n_list_keys, v->d_list_keys, 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipEventRecord(t1, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        LHGT_HIP(hipEventElapsedTime(&v->keys_ms, t0, t1));
        unsigned long long occ = 0;
        memcpy(&occ, small + 4, 8);
        v->n_listed = small[1];
        v->n_items = small[2];
        v->reads_per_bucket = occ ? (double)v->n_listed / (double)occ : 0.0;
        v->store_gen = ctx->store_gen;
        v->k = ctx->k;
        memcpy(v->mask0, ctx->hp.mask[0], sizeof v->mask0);
        v->n_pairs = np;
        if (getenv("LHGT_TRACE"))
            fprintf(stderr, "[lhgt] shared vote: %u of %ld reads keyed, %llu of %ld buckets occupied (%.1f reads per bucket, %u work items), %u pairs with a long read, %.1f ms\n",
                    v->n_listed, 2 * np, occ, n_buckets, v->reads_per_bucket, v->n_items, v->n_list_keys, v->keys_ms);
    }
    // from 3 reads per occupied bucket on (measured: 10 M pairs from 300 genomes, 5.6 reads per bucket, 203 line fills per pair instead of 714:
    // 59.8 against the dense kernel's 136.5 ms, plus 20 ms of keys and grouping once per store; per pair the form costs ~1.6 ns + 0.02 ns per
    // line fill against 14.3 ns, so it still pays where two reads in three share nothing -- below that the grouping is not worth its pass)
    static const double min_share = getenv("LHGT_SHARED_MIN") ? atof(getenv("LHGT_SHARED_MIN")) : 3.0;
    *use = forced || (v->reads_per_bucket >= min_share && (double)v->n_list_keys * 8.0 <= (double)np);
    return LHGT_OK;
}

// the shared form's part of the vote of the resident store: the pairs that can vote at all come back as a list of global pair numbers
// (d_list: [0] = how many) for the generic kernel; every other pair is done
int lhgt_vote_shared(lhgt_ctx* ctx, bool* done, const uint32_t** d_list, unsigned long long* d_stats) {
    *done = false;
    *d_list = nullptr;
    bool use = false;
    LHGT_TRY(vshared_prepare(ctx, &use));
    if (!use) return LHGT_OK;
    lhgt_vshared* v = (lhgt_vshared*)ctx->vshared;
    const long np = ctx->n_pairs;
    LHGT_HIP(hipMemsetAsync(v->d_read_rec, 0, (size_t)np * 2 * sizeof(VsReadRec), ctx->stream));
    LHGT_HIP(hipMemcpyAsync(v->d_list, v->d_list_keys, (size_t)(v->n_list_keys + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
    unsigned long long* st = d_stats ? d_stats + 6 : nullptr;
    if (v->n_items)
        hipLaunchKernelGGL(vs_probe, blocks2d((long)v->n_items), dim3(64 * VS_WAVES), 0, ctx->stream, ctx->hp, v->d_items, v->n_items, v->d_order, v->d_desc,
                           ctx->d_peak_kmer, ctx->d_loci, v->d_read_rec, st, getenv("LHGT_VS_ABLATE") ? atoi(getenv("LHGT_VS_ABLATE")) : 0);
    if (!(ctx->debug & 1))
        hipLaunchKernelGGL(vs_filter, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t)np, v->d_read_rec, v->d_list, ctx->debug);
    LHGT_HIP(hipGetLastError());
    if (d_stats) hipLaunchKernelGGL(vs_count_list, dim3(1), dim3(1), 0, ctx->stream, (const uint32_t*)v->d_list, d_stats + 5);
    if (getenv("LHGT_TRACE")) {
        uint32_t n_l = 0;
        LHGT_HIP(hipMemcpyAsync(&n_l, v->d_list, 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        fprintf(stderr, "[lhgt] shared vote: %u work items, %u of %ld pairs can vote and go to the generic kernel\n", v->n_items, n_l, np);
    }
    *d_list = v->d_list;
    *done = true;
    return LHGT_OK;
}
