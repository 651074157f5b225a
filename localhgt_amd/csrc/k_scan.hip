// k_scan.hip -- phase B: reference scan, window/peak stencil and the peak registry
// (read_index E:888-979, slide_window E:550-725, add_peak/merge_peak E:239-301).
//
// The reference walks every contig sequentially; here each step is restated as an
// order-independent per-position rule over tiles of TILE positions (TILE % 50 == 0):
//   B1 ref_flags      single/trio from the e gathered counts                     (E:573-595, 933-945)
//   B2 window_good    500-wide window sums -> good-window bit, per-tile summary    (E:597-615)
//   B3 interval_select "inside a merged good interval" from the nearest good      (E:617-638, 675-686)
//                     window on each side; 5-wide contrast peaks inside it        (E:644-671)
//                     and first-peak-of-its-50bp-bucket flags                      (E:288-301)
//   B4 tile_scan      exclusive scan of new-peak counts = sequential peak ids     (E:232-235, 275)
//   B5 register       peak_loci + peak_kmer[h] = max id (ids grow in write order) (E:247-267)
// flags byte per reference position: bit0 single, bit1 trio, bit2 good window, bit3 peak (computed inside intervals only),
// bit4 inside a good interval, bit5 selected (peak & interval), bit6 new peak.
#include <type_traits>
#include <chrono>
#include "lhgt_hash.hpp"

namespace lhgt {

constexpr int BT = 256;        // threads per scan block
constexpr int WINDOW = 500;    // E:556
constexpr int HL2 = 512;                    // halo of B2: the window sums look 499 positions back
constexpr int HL4 = 96, HR4 = 80;                // halo of the contrast test: 2k+14 back, 2k+9 forward
constexpr int HALO3 = 2500;    // B3: 2*window on each side plus the 500 merge gap (E:618, 625, 629)

__device__ __forceinline__ uint32_t count_of(const uint32_t* __restrict__ T, uint32_t h) {
    return (T[h >> 4] >> ((h & 15u) * 2u)) & 3u;
}

// ---- B0: one bit per 64-byte line of the count table (256 slots): every slot of the line holds 3.  When most lines are like
// that (a deep sample saturates the table: 100 M pairs put 71 G increments on 4.3 G slots), ref_flags asks this 2 MiB,
// L2-resident bitmap first and touches HBM only for the mixed lines.
// n_sat[0] = saturated lines, n_sat[1] = slots holding 3 in every 16th line (how full the table is decides between the two forms
// of B1 below).
__global__ void __launch_bounds__(256) table_line_summary(const uint32_t* __restrict__ counts, size_t n_lines, uint32_t* __restrict__ satline,
                                                          unsigned long long* __restrict__ n_sat) {
    // grid-stride over runs of 64 lines per wave: the two totals leave with ONE atomic pair per wave at the end (one pair per 64
    // lines kept 262 k waves queueing on two addresses: 4.3 ms for a pass that streams 1 GiB)
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6), wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned long long my_sat = 0;
    int slots3 = 0;
    for (size_t l0 = wave * 64; l0 < n_lines; l0 += n_waves * 64) {
        const size_t line = l0 + lane;
        bool sat = false;
        if (line < n_lines) {
            const uint4* p = (const uint4*)(counts + line * 16);
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            sat = (a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w & c.x & c.y & c.z & c.w & d.x & d.y & d.z & d.w) == 0xffffffffu;
            if ((line & 15) == 0) {   // every 16th line is plenty for a fraction (1 M lines of the k = 32 table)
                const uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
                for (int q = 0; q < 16; q++) slots3 += __popc(w[q] & (w[q] >> 1) & 0x55555555u);
            }
        }
        const unsigned long long bal = __ballot(sat);
        if (lane == 0) {
            satline[l0 >> 5] = (uint32_t)bal;            // 64 consecutive lines -> two words (n_lines is a multiple of 64)
            satline[(l0 >> 5) + 1] = (uint32_t)(bal >> 32);
            my_sat += (unsigned long long)__popcll(bal);
        }
    }
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) slots3 += __shfl_xor(slots3, dd, 64);
    if (lane == 0) {
        if (my_sat) atomicAdd(n_sat, my_sat);
        if (slots3) atomicAdd(n_sat + 1, (unsigned long long)slots3);
    }
}

// lhgt_work_stats: probes of the single-first / trio-first probe kernels, read off the per-position probe-state bytes they leave
// (pstate = hashes that read 3 in bits 0-2, hashes PROBED in bits 4-6): sum of popcount(bits 4-6)
__global__ void __launch_bounds__(256) pstate_probe_sum(const uint8_t* __restrict__ ps, uint64_t n, unsigned long long* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) acc += (unsigned long long)__popc((uint32_t)(ps[i] >> 4) & 7u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// Which hash of a position to ask first (round 5).  A hash is min(fwd, rc) of two uniform words: slots thin out towards the top
// of the table (density 2(1 - x)), and the reads' k-mers with them, so the SMALLER a hash the likelier its slot reads 3.  The
// single-first form wants a 3 soonest -- smallest hash first --, a position it probes completely then wants a "not 3" -- the largest
// next --; the trio-first form wants a "not 3" soonest: largest first.  The record bits (flags, pstate) stay by hash number.
// t-th hash to probe of e <= 3: lo = number of the smallest hash, hi = of the largest (of the first e), mid the third.
struct ProbeOrder { int lo, mid, hi; };
__device__ __forceinline__ ProbeOrder probe_order(const uint32_t (&h)[3], int e) {
    ProbeOrder o{0, 1, 2};
    if (e == 2) { o.lo = h[1] < h[0] ? 1 : 0; o.hi = 1 - o.lo; o.mid = o.hi; }
    else if (e >= 3) {
        o.lo = h[1] < h[0] ? 1 : 0;
        o.lo = h[2] < (o.lo ? h[1] : h[0]) ? 2 : o.lo;
        o.hi = h[1] >= h[0] ? 1 : 0;
        o.hi = h[2] >= (o.hi ? h[1] : h[0]) ? 2 : o.hi;
        if (o.hi == o.lo) o.hi = (o.lo + 1) % 3;        // three equal hashes
        o.mid = 3 - o.lo - o.hi;
    } else o.mid = o.hi = 0;
    return o;
}
__device__ __forceinline__ uint32_t pick3(const uint32_t (&h)[3], int i) { return i == 0 ? h[0] : i == 1 ? h[1] : h[2]; }

// ---- B1
template <bool SAT>
__global__ void __launch_bounds__(BT) ref_flags(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                const RefSource rs, const uint32_t* __restrict__ counts,
                                                int k, int e, uint8_t* __restrict__ flags, uint8_t* __restrict__ nzmask,
                                                const uint32_t* __restrict__ satline, long n_blk) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        if (j >= c.len) break;
        uint8_t f = 0, nz = 0;
        if (j < nk) {  // the last k-1 positions have no k-mer: zero (quirk Q1 contract)
            const RefKmer km = ref_kmer(rs, c, j, k, e);
            int hc = 0;
            for (int i = 0; i < e; i++) {
                uint32_t h = ref_hash(rs, km, i);
                uint32_t cnt = 0u;                                 // hash 0 = invalid (E:936-941)
                if (h != 0) {
                    if (SAT && ((satline[h >> 13] >> ((h >> 8) & 31u)) & 1u)) cnt = 3u;   // the whole 256-slot line is saturated
                    else cnt = count_of(counts, h);
                }
                if (cnt == 3u) hc++;                               // least_depth 3 (E:580)
                if (cnt > 0u && i < 8) nz |= (uint8_t)(1u << i);    // record_ref_hit > 0 (E:250, 265), reused by register_peaks
            }
            f = (uint8_t)((hc > 0) | ((hc == e) << 1));
        }
        flags[c.flat_base + j] = f;
        nzmask[c.flat_base + j] = nz;
    }
}

// ---- B1 for a nearly saturated table ("lite").  What the later steps need from a tile is (1) the exact `single` bit of every
// position and (2) whether every position is a good window; `trio` only enters through the window sums.  With most slots at 3,
//   * `single` is settled by the first hash that reads 3: 1.03 probes per position instead of e;
//   * every 8th position is probed completely, and the exact trio count of those alone is a LOWER bound of a window's
//     `three` -- 58 of 62 sampled positions on configs[2] against the 40 needed -- so "every position of this tile is a good
//     window" is proven without the other probes (window_lite);
//   * only tiles that cannot be proven that way get their remaining hashes probed (ref_flags_fill, with the 512 positions their
//     window sums look back on) and go through the exact window_good.
// flags bit 7 = "all e hashes probed: the trio bit is exact"; pstate = per hash: bits 0-2 reads 3, bits 4-6 probed.
// Same peaks, ids and votes as the exact form (every consumer sees exact inputs); the trio bit of positions inside proven
// tiles stays a lower bound, nothing reads it.  e <= 3.
template <bool SAT>
__global__ void __launch_bounds__(BT) ref_flags_lite(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                     const RefSource rs, const uint32_t* __restrict__ counts,
                                                     int k, int e, uint8_t* __restrict__ flags, uint8_t* __restrict__ pstate,
                                                     const uint32_t* __restrict__ satline, const uint32_t* __restrict__ list /* nullable */, long n_blk,
                                                     int stride /* every stride-th position is probed completely */) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[list ? list[blk] : blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    const uint32_t full = (1u << e) - 1u;
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        if (j >= c.len) break;
        uint8_t f = 0x80, ps = 0x70;   // the last k-1 positions have no k-mer: exact zeros (quirk Q1 contract)
        if (j < nk) {
            const RefKmer km = ref_kmer(rs, c, j, k, e);
            uint32_t h[3];
#pragma unroll
            for (int i = 0; i < 3; i++) h[i] = i < e ? ref_hash(rs, km, i) : 0u;
            const bool sample = (j % stride) == 0;
            uint32_t known = 0, is3 = 0;
            // smallest hash first (probe_order); a sampled position then asks its largest
            const ProbeOrder po = probe_order(h, e);
            const int second = sample ? po.hi : po.mid, third = sample ? po.mid : po.hi;
#pragma unroll
            for (int t = 0; t < 3; t++)
                // a sampled position goes on until `single` AND `trio` are decided: a hash that reads 3 and one that does not settle
                // both (round 4: before, all e hashes -- 3 probes where 2.5 do); any other position stops at the first 3
                if (t < e && (is3 == 0u || (sample && known == is3))) {
                    const int i = t == 0 ? po.lo : t == 1 ? second : third;
                    const uint32_t hv = pick3(h, i);
                    uint32_t cnt = 0u;                                 // hash 0 = invalid (E:936-941)
                    if (hv != 0) {
                        if (SAT && ((satline[hv >> 13] >> ((hv >> 8) & 31u)) & 1u)) cnt = 3u;
                        else cnt = count_of(counts, hv);
                    }
                    known |= 1u << i;
                    if (cnt == 3u) is3 |= 1u << i;                     // least_depth 3 (E:580)
                }
            // both flags are exact once every hash was probed, or one read 3 (single) and one did not (no trio)
            const bool exact = known == full || (is3 != 0u && known != is3);
            f = (uint8_t)((is3 != 0u) | ((known == full && is3 == full) << 1) | (exact ? 0x80 : 0));
            ps = (uint8_t)(is3 | (known << 4));
        }
        flags[c.flat_base + j] = f;
        pstate[c.flat_base + j] = ps;
    }
}

// ---- B1 for a SPARSE table ("trio-first").  A window is good only if `three` -- positions whose e hashes ALL read 3 -- reaches
// its threshold (E:610-615), and on a table that is mostly empty (the default down-sampled run: 15 % of the slots at 3; a sample
// of a few hundred genomes: 20 %) almost no position outside the sampled genomes is such a position.  So the hashes of a position
// are probed one after the other until one does NOT read 3: 1 + f + f^2 probes instead of 3 (1.24 at f = 0.2), and `trio` is
// exact everywhere.  Tiles with no window reaching the `three` threshold, nor within three tiles of one, are done: no good
// window, no interval, nothing of them is ever read again.  The others get their remaining probes (ref_flags_fill) and the exact
// window_good, like the unsettled tiles of the single-first form.  flags bit 7 = "single and trio are exact" (some probed hash
// read 3, or all were probed); pstate as above.  Same peaks, ids and votes as the exact form.
__global__ void __launch_bounds__(BT) ref_flags_trio(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                     const RefSource rs, const uint32_t* __restrict__ counts,
                                                     int k, int e, uint8_t* __restrict__ flags, uint8_t* __restrict__ pstate, long n_blk) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    const uint32_t full = (1u << e) - 1u;
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        if (j >= c.len) break;
        uint8_t f = 0x80, ps = 0x70;   // the last k-1 positions have no k-mer: exact zeros (quirk Q1 contract)
        if (j < nk) {
            const RefKmer km = ref_kmer(rs, c, j, k, e);
            uint32_t h[3];
#pragma unroll
            for (int i = 0; i < 3; i++) h[i] = i < e ? ref_hash(rs, km, i) : 0u;
            uint32_t known = 0, is3 = 0, stop_nz = 0;
            bool all3 = true;
            const ProbeOrder po = probe_order(h, e);                              // largest hash first: the likeliest "not 3"
#pragma unroll
            for (int t = 0; t < 3; t++)
                if (t < e && all3) {
                    const int i = t == 0 ? po.hi : t == 1 ? (e == 2 ? po.lo : po.mid) : po.lo;
                    const uint32_t hv = pick3(h, i);
                    const uint32_t cnt = hv != 0 ? count_of(counts, hv) : 0u;       // hash 0 = invalid (E:936-941)
                    known |= 1u << i;
                    if (cnt == 3u) is3 |= 1u << i;                                    // least_depth 3 (E:580)
                    else { all3 = false; stop_nz = cnt > 0u; }
                }
            const bool exact = known == full || is3 != 0u;
            f = (uint8_t)((is3 != 0u) | ((is3 == full) << 1) | (exact ? 0x80 : 0));
            // bit 3 (round 5): the count of the one probed hash that did not read 3 is > 0 -- what register_peaks asks of every hash
            // of a selected position (E:250, 265); with it, and the fill's record below, it probes the table for unprobed hashes only
            ps = (uint8_t)(is3 | (stop_nz << 3) | (known << 4));
        }
        flags[c.flat_base + j] = f;
        pstate[c.flat_base + j] = ps;
    }
}

// slot-first form: the probe-state bytes of the listed tiles (and of their look-back) back to "nothing known" -- the sweep writes none,
// what is there is the last scan's
__global__ void __launch_bounds__(BT) clear_pstate_tiles(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                         const uint32_t* __restrict__ list, uint8_t* __restrict__ pstate, long n_blk) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[list[blk]];
    const ContigDev c = contigs[t.contig];
    for (int i0 = threadIdx.x; i0 < TILE + HL2; i0 += BT) {
        const long j = (long)t.j0 - HL2 + i0;
        if (j >= 0 && j < c.len) pstate[c.flat_base + j] = 0;
    }
}

// the remaining probes of the listed tiles and of the HL2 positions their window sums look back on
__global__ void __launch_bounds__(BT) ref_flags_fill(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                     const uint32_t* __restrict__ list, const RefSource rs,
                                                     const uint32_t* __restrict__ counts, int k, int e, uint8_t* __restrict__ flags,
                                                     uint8_t* __restrict__ pstate, long n_blk, int record_nz /* the trio-first form: see below */,
                                                     const uint32_t* __restrict__ cand /* mark_need_tiles' input, or null */, long n_tiles) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const long q0 = list[blk];
    const TileDev t = tiles[q0];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    const uint32_t full = (1u << e) - 1u;
    // the look-back positions are the body of the tile in front: if that tile is listed too (mark_need_tiles' rule, asked again here) its
    // workgroup fills them -- once; without this both did whenever they ran side by side, which neighbours in the list do
    bool prev_listed = false;
    if (cand && q0 > 0 && tiles[q0 - 1].contig == t.contig)
        for (long q = q0 - 4; q <= q0 + 2; q++)
            if (q >= 0 && q < n_tiles && tiles[q].contig == t.contig && cand[q]) prev_listed = true;
    for (int i0 = (prev_listed ? HL2 : 0) + threadIdx.x; i0 < TILE + HL2; i0 += BT) {
        const long j = (long)t.j0 - HL2 + i0;
        if (j < 0 || j >= c.len) continue;
        const uint8_t f = flags[c.flat_base + j];
        if (f & 0x80) continue;                     // already exact (a neighbouring workgroup may write the same values meanwhile)
        const uint8_t ps = pstate[c.flat_base + j];
        if (ps & 0x80) continue;                    // ... and has written this byte already
        uint32_t known = (ps >> 4) & 7u, is3 = ps & 7u;
        uint32_t nz = is3 | (((ps >> 3) & 1u) ? (known & ~is3) : 0u);   // trio-first: the one probed hash that did not read 3 has its "> 0" in bit 3
        if (j < nk) {
            const RefKmer km = ref_kmer(rs, c, j, k, e);
#pragma unroll
            for (int i = 0; i < 3; i++)
                if (i < e && !((known >> i) & 1u)) {
                    const uint32_t h = ref_hash(rs, km, i);
                    const uint32_t cnt = h != 0 ? count_of(counts, h) : 0u;
                    if (cnt == 3u) is3 |= 1u << i;
                    if (cnt > 0u) nz |= 1u << i;
                }
        }
        known = full;
        flags[c.flat_base + j] = (uint8_t)((f & 0x7c) | (is3 != 0u) | ((is3 == full) << 1) | 0x80);
        // record_nz: every hash's count is known here, so bits 4-6 take "count > 0" per hash and bit 7 says so (register_peaks then
        // probes nothing for this position).  The single-first form probes several hashes that do not read 3 without keeping their
        // counts, so there the byte stays "probed" and register_peaks looks the counts up, as before.
        pstate[c.flat_base + j] = record_nz ? (uint8_t)(is3 | (nz << 4) | 0x80) : (uint8_t)(is3 | (known << 4));
    }
}

// (Round 6 tried this kernel rearranged around its memory round trips -- plane words in LDS, a thread's flag and state bytes fetched
// up front, the look-ups of a position leaving together and the next position's on their way before this one's bytes are stored:
// 296 -> 294 ms on the default-sample leg.  The fill is bound by the line fills of its ~11 G look-ups, not by their latency.  Leaving a
// listed neighbour's body to that neighbour -- below -- took 3 % off: 294 -> 285 ms.)
// exclusive prefix over the block's per-thread values: shuffle scan inside each wave, then the few wave totals through LDS
// (the first version had every thread add up its predecessors: a third of interval_select's instructions)
__device__ __forceinline__ int block_excl_sum(int v, int* sh /*[BT]*/) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    int off = 0;
    for (int q = 0; q < wv; q++) off += sh[q];
    __syncthreads();
    return off + incl - v;
}

// ---- the window kernels' common ground: one WAVE per tile, no block barrier.  These passes run over every tile of the reference
// and are bound by a tile's chain of memory round trips (descriptors, flags, flags again), not by bytes: a workgroup per tile
// kept 7 tiles in flight per CU (27 ms per 13 Gbase, ten times a streaming pass over the flags); waves keep 32.  A lane loads the
// flag bytes of positions lane, lane + 64, ... (the tile and the HL2 positions its windows look back on); the two bits per
// position become ballot words in LDS; rank(i) = bits set at positions <= i = set bits before the word + popcount inside it; a
// lane then tests 32 consecutive positions, walking the plane bits that enter and leave the window.
constexpr int WL_WORDS = (TILE + HL2 + 63) / 64;   // 40 ballot words per plane
constexpr int WL_PER = 32;                         // consecutive positions per lane in the window test
static_assert(63 * WL_PER >= TILE - WL_PER && 64 * WL_PER >= TILE && HL2 % WL_PER == 0 && HL2 % 64 == 0 && WL_WORDS <= 64, "lanes cover the tile; aligned 32-bit groups");
struct WavePlanes {
    unsigned long long B1[WL_WORDS + 1], B3[WL_WORDS + 1];   // one zero pad word: a 32-bit group may straddle two words
    int R1[WL_WORDS + 1], R3[WL_WORDS + 1];
    uint32_t G[64];                                          // window_good: lane l's good bits of positions 32 l .. 32 l + 31
    __device__ __forceinline__ static int rank(const unsigned long long* B, const int* R, int i) {
        return R[i >> 6] + __popcll(B[i >> 6] & (~0ull >> (63 - (i & 63))));
    }
    __device__ __forceinline__ static uint32_t bits32(const unsigned long long* B, int x) {
        const int w = x >> 6, o = x & 63;
        unsigned long long v = B[w] >> o;
        if (o > 32) v |= B[w + 1] << (64 - o);
        return (uint32_t)v;
    }
};
// fs[r] = flag byte of position lo + lane + 64 r (0 outside the contig); planes and ranks of this wave's tile.  m3: the trio
// predicate is (f & m3) == m3 -- 0x02, or 0x82 where only completely probed positions count (lite form).
__device__ __forceinline__ void wave_planes(const uint8_t* __restrict__ F, long lo, long len, int m3, int lane, uint8_t (&fs)[WL_WORDS], WavePlanes& P) {
    constexpr int NW = TILE + HL2;
    if (lo >= 0 && lo + NW <= len) {
        // a tile inside its contig with its whole look-back (all but the first and the last of a contig): ONE address and immediate
        // offsets, no bounds test per byte (round 4: the tests and their 64-bit address arithmetic were a third of the kernel's vector
        // instructions, and 134 registers held three waves per SIMD where this kernel lives on the tiles it has in flight)
        const uint8_t* __restrict__ base = F + lo + lane;
#pragma unroll
        for (int r = 0; r < WL_WORDS; r++) fs[r] = (64 * r + 63 < NW || lane + 64 * r < NW) ? base[64 * r] : (uint8_t)0;
    } else {
#pragma unroll
        for (int r = 0; r < WL_WORDS; r++) {       // all loads in flight together
            const int i = lane + 64 * r;
            const long pos = lo + i;
            fs[r] = (i < NW && pos >= 0 && pos < len) ? F[pos] : (uint8_t)0;
        }
    }
#pragma unroll
    for (int r = 0; r < WL_WORDS; r++) {
        const unsigned long long b1 = __ballot(fs[r] & 1), b3 = __ballot((fs[r] & m3) == m3);
        if (lane == 0) { P.B1[r] = b1; P.B3[r] = b3; }
    }
    if (lane == 0) { P.B1[WL_WORDS] = 0ull; P.B3[WL_WORDS] = 0ull; }
    __builtin_amdgcn_wave_barrier();
    const int c1 = lane < WL_WORDS ? __popcll(P.B1[lane]) : 0, c3 = lane < WL_WORDS ? __popcll(P.B3[lane]) : 0;
    int i1 = c1, i3 = c3;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t1 = __shfl_up(i1, d, 64), t3 = __shfl_up(i3, d, 64);
        if (lane >= d) { i1 += t1; i3 += t3; }
    }
    if (lane < WL_WORDS) { P.R1[lane] = i1 - c1; P.R3[lane] = i3 - c3; }
    if (lane == WL_WORDS - 1) { P.R1[WL_WORDS] = i1; P.R3[WL_WORDS] = i3; }
    __builtin_amdgcn_wave_barrier();
}
// this lane's 32 positions (32 lane ..): bit u set = position 32 lane + u passes both thresholds
__device__ __forceinline__ uint32_t wave_window_test(const WavePlanes& P, int lane, int n_here, int one_min, int three_min) {
    const int jj0 = lane * WL_PER, i0 = jj0 + HL2;
    uint32_t good = 0;
    if (jj0 < n_here) {
        int a1 = WavePlanes::rank(P.B1, P.R1, i0 - 1), b1 = WavePlanes::rank(P.B1, P.R1, i0 - WINDOW - 1);
        int a3 = WavePlanes::rank(P.B3, P.R3, i0 - 1), b3 = WavePlanes::rank(P.B3, P.R3, i0 - WINDOW - 1);
        const uint32_t in1 = WavePlanes::bits32(P.B1, i0), out1 = WavePlanes::bits32(P.B1, i0 - WINDOW);
        const uint32_t in3 = WavePlanes::bits32(P.B3, i0), out3 = WavePlanes::bits32(P.B3, i0 - WINDOW);
#pragma unroll
        for (int u = 0; u < WL_PER; u++) {
            a1 += (in1 >> u) & 1u; b1 += (out1 >> u) & 1u;
            a3 += (in3 >> u) & 1u; b3 += (out3 >> u) & 1u;
            if (jj0 + u < n_here && a1 - b1 >= one_min && a3 - b3 >= three_min) good |= 1u << u;
        }
    }
    return good;
}

// does any window of the tile reach the `three` threshold?  (exact trio sums; `one` is not looked at: it is a lower bound here)
__global__ void __launch_bounds__(BT) window_trio(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, int three_min,
                                                  const uint8_t* __restrict__ flags, uint32_t* __restrict__ cand, long n_todo) {
    __shared__ WavePlanes Ps[BT / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long idx = block2d() * (BT / 64) + wv;
    if (idx >= n_todo) return;                 // wave-uniform; nothing below synchronises across waves
    const TileDev t = tiles[idx];
    const ContigDev c = contigs[t.contig];
    const long len = c.len;
    WavePlanes& P = Ps[wv];
    uint8_t fs[WL_WORDS];
    wave_planes(flags + c.flat_base, (long)t.j0 - HL2, len, 0x02, lane, fs, P);
    uint32_t any = 0;
    if (P.R3[WL_WORDS] >= three_min) {
        const long rest = len - (long)t.j0;
        any = wave_window_test(P, lane, rest < TILE ? (int)rest : TILE, 0, three_min);   // `one` is not looked at: it is a lower bound here
    }
    const unsigned long long bal = __ballot(any != 0u);
    if (lane == 0) cand[idx] = bal ? 1u : 0u;
}

// tiles that need exact flags: within three tiles (the 2560-position reach of the interval rules plus the contrast halo) of a
// tile with a candidate window, inside the same contig; every other tile holds no good window (tile_good = 0) and is done
__global__ void __launch_bounds__(256) mark_need_tiles(const TileDev* __restrict__ tiles, const uint32_t* __restrict__ cand, long n_tiles,
                                                       uint8_t* __restrict__ tile_good, uint32_t* __restrict__ need, unsigned int* __restrict__ n_need) {
    const long q0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q0 >= n_tiles) return;
    const uint32_t contig = tiles[q0].contig;
    bool reach = false;
    for (long q = q0 - 3; q <= q0 + 3; q++)
        if (q >= 0 && q < n_tiles && tiles[q].contig == contig && cand[q]) reach = true;
    if (reach) need[atomicAdd(n_need, 1u)] = (uint32_t)q0;
    else tile_good[q0] = 0;
}

// ---- B2: good-window bit per position (E:597-615).  A good position lies inside a merged interval whatever its neighbours do
// (distance 0 to a good window, E:618), so its "inside" bit is set here as well.  One byte per tile: bit 0 some position is
// good, bit 1 all are, bit 2 all have a hit (`single`) -- mark_active_tiles uses them to keep whole runs of covered reference
// away from interval_select.
__global__ void __launch_bounds__(BT) window_good(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                  const uint32_t* __restrict__ list /* nullable: the tiles to do */, int keep7,
                                                  int one_min, int three_min, uint8_t* __restrict__ flags, uint8_t* __restrict__ tile_good, long n_todo) {
    __shared__ WavePlanes Ps[BT / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long idx = block2d() * (BT / 64) + wv;
    if (idx >= n_todo) return;                 // wave-uniform; nothing below synchronises across waves
    const uint32_t tile = list ? list[idx] : (uint32_t)idx;
    const TileDev t = tiles[tile];
    const ContigDev c = contigs[t.contig];
    const long len = c.len, lo = (long)t.j0 - HL2;   // the window sums look back only
    uint8_t* F = flags + c.flat_base;
    constexpr int NW = TILE + HL2;
    WavePlanes& P = Ps[wv];
    uint8_t fs[WL_WORDS];
    wave_planes(F, lo, len, 0x02, lane, fs, P);
    const long rest = len - (long)t.j0;
    const int n_here = rest < TILE ? (int)rest : TILE;
    uint32_t good = 0;
    if (P.R1[WL_WORDS] >= one_min && P.R3[WL_WORDS] >= three_min)   // otherwise no window of this tile can reach the thresholds
        good = wave_window_test(P, lane, n_here, one_min, three_min);
    int n_good = __popc(good);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_good += __shfl_xor(n_good, d, 64);
    if (n_good) {
        // the good bits go through LDS so that the flags are written as they were read: lane = position mod 64, from registers
        P.G[lane] = good;
        __builtin_amdgcn_wave_barrier();
        if (lo >= 0 && lo + NW <= len) {          // (see wave_planes)
            uint8_t* __restrict__ base = F + lo + lane;
#pragma unroll
            for (int r = HL2 / 64; r < WL_WORDS; r++) {
                const int jj = lane + 64 * r - HL2;
                if ((64 * r + 63 < NW || lane + 64 * r < NW) && ((P.G[jj >> 5] >> (jj & 31)) & 1u))
                    base[64 * r] = (uint8_t)((fs[r] & 3) | 4 | 16 | keep7);
            }
        } else {
#pragma unroll
            for (int r = HL2 / 64; r < WL_WORDS; r++) {
                const int i = lane + 64 * r, jj = i - HL2;
                const long j = lo + i;
                if (i < NW && j < len && ((P.G[jj >> 5] >> (jj & 31)) & 1u))
                    F[j] = (uint8_t)((fs[r] & 3) | 4 | (j >= 1 ? 16 : 0) | keep7);   // its own two flags, good window, inside (E:618)
            }
        }
    }
    if (lane == 0) {
        const int all_good = n_good == n_here;
        const int all_single = WavePlanes::rank(P.B1, P.R1, NW - 1) - WavePlanes::rank(P.B1, P.R1, HL2 - 1) == n_here;
        // bits 3 / 4: the first HR4 / last HL4 positions all have a hit -- what the neighbouring tiles' contrast tests reach into
        const int head = n_here >= HR4 && WavePlanes::rank(P.B1, P.R1, HL2 + HR4 - 1) - WavePlanes::rank(P.B1, P.R1, HL2 - 1) == HR4;
        const int tail = n_here == TILE && WavePlanes::rank(P.B1, P.R1, NW - 1) - WavePlanes::rank(P.B1, P.R1, NW - 1 - HL4) == HL4;
        tile_good[tile] = (uint8_t)((n_good ? 1 : 0) | (all_good << 1) | (all_single << 2) | (head << 3) | (tail << 4));
    }
}

// ---- B2 on the lite flags: the single sums are exact, the trio sums count only completely probed positions (a lower bound).
// A tile all of whose positions pass both thresholds with those sums is settled as window_good would settle it: all good, all
// inside.  Any other tile is listed for the exact treatment (ref_flags_fill + window_good) and left untouched.
__global__ void __launch_bounds__(BT) window_lite(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                  int one_min, int three_min, uint8_t* __restrict__ flags, uint8_t* __restrict__ tile_good,
                                                  uint32_t* __restrict__ need, unsigned int* __restrict__ n_need,
                                                  const uint32_t* __restrict__ pilot /* nullable: only count, over these tiles */, long n_todo) {
    __shared__ WavePlanes Ps[BT / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long idx = block2d() * (BT / 64) + wv;
    if (idx >= n_todo) return;                 // wave-uniform; nothing below synchronises across waves
    const uint32_t tile = pilot ? pilot[idx] : (uint32_t)idx;
    const TileDev t = tiles[tile];
    const ContigDev c = contigs[t.contig];
    const long len = c.len, lo = (long)t.j0 - HL2;
    uint8_t* F = flags + c.flat_base;
    constexpr int NW = TILE + HL2;
    WavePlanes& P = Ps[wv];
    uint8_t fs[WL_WORDS];
    wave_planes(F, lo, len, 0x82, lane, fs, P);
    const long rest = len - (long)t.j0;
    const int n_here = rest < TILE ? (int)rest : TILE;
    int n_good = __popc(wave_window_test(P, lane, n_here, one_min, three_min));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n_good += __shfl_xor(n_good, d, 64);
    if (pilot) {                     // trial run over a few runs of tiles: how many would the lower bound settle?
        if (lane == 0 && n_good != n_here) atomicAdd(n_need, 1u);
        return;
    }
    if (n_good != n_here) {          // not provable from the lower bound: exact treatment
        if (lane == 0) {
            tile_good[tile] = 0;
            need[atomicAdd(n_need, 1u)] = tile;
        }
        return;
    }
    if (lo >= 0 && lo + NW <= len) {              // (see wave_planes: every position exists, none is position 0)
        uint8_t* __restrict__ base = F + lo + lane;
#pragma unroll
        for (int r = HL2 / 64; r < WL_WORDS; r++)
            if (64 * r + 63 < NW || lane + 64 * r < NW) base[64 * r] = (uint8_t)(fs[r] | 4 | 16);
    } else {
#pragma unroll
        for (int r = HL2 / 64; r < WL_WORDS; r++) {   // the tile's own positions: good window, inside (E:618), written from the registers
            const int i = lane + 64 * r;
            const long j = lo + i;
            if (i < NW && j < len) F[j] = (uint8_t)(fs[r] | 4 | (j >= 1 ? 16 : 0));
        }
    }
    if (lane == 0) {
        const int all_single = WavePlanes::rank(P.B1, P.R1, NW - 1) - WavePlanes::rank(P.B1, P.R1, HL2 - 1) == n_here;
        const int head = n_here >= HR4 && WavePlanes::rank(P.B1, P.R1, HL2 + HR4 - 1) - WavePlanes::rank(P.B1, P.R1, HL2 - 1) == HR4;
        const int tail = n_here == TILE && WavePlanes::rank(P.B1, P.R1, NW - 1) - WavePlanes::rank(P.B1, P.R1, NW - 1 - HL4) == HL4;
        tile_good[tile] = (uint8_t)(1 | 2 | (all_single << 2) | (head << 3) | (tail << 4));
    }
}

// ---- B3: interval mask, contrast peaks inside it, new-peak flags.
// (1) Good-window bits of [j0-2560, j0+TILE+2560) as one ballot word per 64 positions; the nearest good window on either side
//     of a position is a clz/ctz inside its word or the carry of a neighbouring word; "inside a merged interval" follows
//     (E:617-638, 675-686).  Tiles with no good window in themselves or their two neighbours on either side stop after reading
//     five bytes: nothing of them can be inside.
// (2) The contrast test of E:644-671 in closed form, only for tiles that have positions inside an interval (peaks elsewhere are
//     never looked at, E:688-692).  With W5[t] = singles over t-4..t, the literal update of `left` (E:658) telescopes to
//     left_m = W5[j-5] + W5[j-m-5] - W5[j-k-5], so
//       the test run AT j marks j       iff  W5[j-5] - W5[j-k-5] - W5[j] + min_{m in [k,2k)} W5[j-m-5] <= -2
//       a test run at j' = j+m+5 marks j iff  W5[j] + max_{t in [j+k, j+2k)} (W5[t] - W5[t-k] - W5[t+5]) >= 2
//     (t = j+m; tests exist only for 2k+10 < j' < len): sliding-window extrema of width k; a thread owns 8 consecutive
//     positions and shares the part of the window they have in common.
constexpr int H3 = 2560;                         // >= HALO3, multiple of 64
constexpr int NW3 = (TILE + 2 * H3 + 63) / 64;   // ballot words per tile
constexpr int N4 = TILE + HL4 + HR4;
__global__ void __launch_bounds__(256) mark_active_tiles(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                         const uint8_t* __restrict__ tile_good, long n_tiles,
                                                         uint32_t* __restrict__ active, unsigned int* __restrict__ n_active, int no_settle) {
    long q0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q0 >= n_tiles) return;
    const uint32_t contig = tiles[q0].contig;
    bool reach = false;   // the 2560-position halo spans at most two tiles of the same contig on either side
    for (long q = q0 - 2; q <= q0 + 2; q++)
        if (q >= 0 && q < n_tiles && tiles[q].contig == contig && (tile_good[q] & 1)) reach = true;
    if (!reach) return;
    // settled by window_good alone: every position of the tile is good (so inside, bit already set) and every position the
    // contrast test can look at -- HL4 before to HR4 after the tile, all inside the contig -- has a hit, so no sum differs from its
    // neighbours and no position is a peak: nothing selected, no new peak (tile_count stays 0)
    const long j0 = tiles[q0].j0, len = contigs[contig].len;
    const bool settled = (tile_good[q0] & 6) == 6 && j0 >= HL4 && j0 + TILE + HR4 <= len && q0 > 0 && q0 + 1 < n_tiles &&
                         tiles[q0 - 1].contig == contig && tiles[q0 + 1].contig == contig && (tile_good[q0 - 1] & 16) && (tile_good[q0 + 1] & 8);
    if (!settled || no_settle) active[atomicAdd(n_active, 1u)] = (uint32_t)q0;
}

// one workgroup per ACTIVE tile (a tile with a good window in itself or within two tiles: launching millions of workgroups
// that return at once costs ~25 ns each in the dispatcher); tile_count of the others was zeroed by the caller
__global__ void __launch_bounds__(BT) interval_select(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, int k,
                                                      const uint32_t* __restrict__ active,
                                                      uint8_t* __restrict__ flags, uint32_t* __restrict__ tile_count, uint32_t* __restrict__ tile_sel,
                                                      unsigned long long* __restrict__ n_selected, long n_blk) {
    __shared__ unsigned long long gw[NW3];
    __shared__ int prevw[NW3], nextw[NW3];       // last good index in words <= w / first good index in words >= w
    __shared__ int P1[N4], part[BT];
    __shared__ int8_t W[N4], G[N4];
    __shared__ uint8_t sel[TILE], ins[TILE];
    __shared__ int n_new, n_sel, n_ins;
    static_assert(H3 >= HALO3 && H3 % 64 == 0 && H3 <= 2 * TILE, "halo");
    const long blk = block2d();
    if (blk >= n_blk) return;
    const uint32_t tile = active[blk];
    const TileDev t = tiles[tile];
    const ContigDev c = contigs[t.contig];
    const long len = c.len, lo = (long)t.j0 - H3;
    uint8_t* F = flags + c.flat_base;
    constexpr int NONE_LO = -(1 << 28), NONE_HI = 1 << 28;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { n_new = 0; n_sel = 0; n_ins = 0; }
    {
        // all of a wave's flag bytes first (unconditional loads on an index clamped into the contig), then the ballots: a load and its
        // wait per word made 28 dependent round trips of this loop -- most of a tile's 50 us (round 4; the ISA had
        // global_load_ubyte / s_waitcnt vmcnt(0) back to back)
        constexpr int WPW = (NW3 + BT / 64 - 1) / (BT / 64);
        uint8_t fb[WPW];
#pragma unroll
        for (int q = 0; q < WPW; q++) {
            const long pos = lo + 64L * (wv + q * (BT / 64)) + lane;
            fb[q] = F[pos < 0 ? 0 : pos < len ? pos : len - 1];
        }
#pragma unroll
        for (int q = 0; q < WPW; q++) {
            const int w = wv + q * (BT / 64);
            const long pos = lo + 64L * w + lane;
            const unsigned long long bal = __ballot(pos >= 0 && pos < len && ((fb[q] >> 2) & 1));
            if (lane == 0 && w < NW3) gw[w] = bal;
        }
    }
    __syncthreads();
    {
        // prefix maximum of "last good index in word w" and suffix minimum of "first good index in word w" over the NW3 <= 128 words: two
        // waves, shuffles (round 4: two single threads walked the words one LDS round trip at a time)
        static_assert(NW3 <= 128 && BT >= 128, "two waves cover the ballot words");
        const int w = threadIdx.x;
        int pv = NONE_LO, nv = NONE_HI;
        if (w < 128) {
            const unsigned long long g = w < NW3 ? gw[w] : 0ull;
            if (g) { pv = 64 * w + 63 - __clzll((long long)g); nv = 64 * w + __ffsll((long long)g) - 1; }
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int tp = __shfl_up(pv, d, 64), tn = __shfl_down(nv, d, 64);
                if (lane >= d) pv = tp > pv ? tp : pv;
                if (lane + d < 64) nv = tn < nv ? tn : nv;
            }
            if (w == 63) part[0] = pv;          // all of wave 0, for wave 1's prefix
            if (w == 64) part[1] = nv;          // all of wave 1, for wave 0's suffix
        }
        __syncthreads();
        if (w < 128) {
            if (wv == 1) pv = part[0] > pv ? part[0] : pv;
            else nv = part[1] < nv ? part[1] : nv;
            if (w < NW3) { prevw[w] = pv; nextw[w] = nv; }
        }
    }
    __syncthreads();
    // inside a merged interval: within 2*window of a good window (E:618, 625), or in a gap the
    // merge rule closes: start_next - end_prev < window  <=>  next - prev <= 4*window + window (E:629)
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        uint8_t inside = 0;
        if (j < len) {
            const int i = jj + H3, w = i >> 6, bpos = i & 63;
            const unsigned long long word = gw[w];
            const unsigned long long below = word & ((2ull << bpos) - 1ull), above = (word >> bpos) << bpos;
            const int p = below ? 64 * w + 63 - __clzll((long long)below) : (w > 0 ? prevw[w - 1] : NONE_LO);
            const int n = above ? 64 * w + __ffsll((long long)above) - 1 : (w + 1 < NW3 ? nextw[w + 1] : NONE_HI);
            const int dp = i - p, dn = n - i;   // huge when absent
            inside = j >= 1 && (dp <= 2 * WINDOW || dn <= 2 * WINDOW || dp + dn <= 5 * WINDOW);
        }
        ins[jj] = inside;
        sel[jj] = 0;
        if (inside) n_ins = 1;
    }
    __syncthreads();
    if (!n_ins) {
        if (threadIdx.x == 0) tile_count[tile] = 0u;
        return;
    }
    // contrast peaks
    {
        const long lo4 = (long)t.j0 - HL4;
        constexpr int CH = (N4 + BT - 1) / BT;
        const int b = threadIdx.x * CH, en = b + CH < N4 ? b + CH : N4;
        int s1 = 0;
        uint8_t f1[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) {                     // the bytes first, one wait (see the ballots above)
            const long pos = lo4 + b + q;
            f1[q] = F[pos < 0 ? 0 : pos < len ? pos : len - 1];
        }
#pragma unroll
        for (int q = 0; q < CH; q++) {
            const int i = b + q;
            const long pos = lo4 + i;
            if (i < en) {
                s1 += (pos >= 0 && pos < len) ? (f1[q] & 1) : 0;
                P1[i] = s1;
            }
        }
        int o1 = block_excl_sum(s1, part);
        for (int i = b; i < en; i++) P1[i] += o1;
        __syncthreads();
        // every position of the tile and its halo has a hit and none of it hangs over a contig end: all 5-wide sums are 5, both
        // contrasts are 0 - 5 + 5 and no position can be a peak (a present genome is one long run of such tiles)
        const bool flat = lo4 >= 0 && lo4 + N4 <= len && P1[N4 - 1] == N4;
        if (!flat) {
        for (int i = threadIdx.x; i < N4; i += BT) W[i] = (int8_t)(i >= 5 ? P1[i] - P1[i - 5] : 0);
        __syncthreads();
        for (int i = threadIdx.x; i < N4; i += BT) {
            long jp = lo4 + i + 5;   // position of the test that uses t = i
            bool ok = i >= k + 5 && i + 5 < N4 && jp > 2 * k + 10 && jp < len;
            G[i] = (int8_t)(ok ? W[i] - W[i - k] - W[i + 5] : -100);
        }
        __syncthreads();
        constexpr int PER = 8;
        for (int blk = threadIdx.x; blk < TILE / PER; blk += BT) {
            const int jj0 = blk * PER, i0 = jj0 + HL4;
            bool need = false;
#pragma unroll
            for (int q = 0; q < PER; q++) need |= ins[jj0 + q] != 0;
            if (!need) continue;
            // self test: window of W over [i-2k-4, i-k-5]; gather test: window of G over [i+k, i+2k-1]
            const int sa = i0 - 2 * k - 4, sg = i0 + k;
            int cmin = 127, cmax = -128;
            for (int q = PER - 1; q < k; q++) {          // part shared by the 8 windows
                int w = W[sa + q], g = G[sg + q];
                cmin = w < cmin ? w : cmin;
                cmax = g > cmax ? g : cmax;
            }
            int lmin[PER], lmax[PER], rmin[PER], rmax[PER];
            lmin[PER - 1] = 127; lmax[PER - 1] = -128;   // suffix extrema of the first PER-1 values
#pragma unroll
            for (int q = PER - 2; q >= 0; q--) {
                int w = W[sa + q], g = G[sg + q];
                lmin[q] = w < lmin[q + 1] ? w : lmin[q + 1];
                lmax[q] = g > lmax[q + 1] ? g : lmax[q + 1];
            }
            rmin[0] = 127; rmax[0] = -128;               // prefix extrema of the PER-1 values after the shared part
#pragma unroll
            for (int q = 1; q < PER; q++) {
                int w = W[sa + k + q - 1], g = G[sg + k + q - 1];
                rmin[q] = w < rmin[q - 1] ? w : rmin[q - 1];
                rmax[q] = g > rmax[q - 1] ? g : rmax[q - 1];
            }
#pragma unroll
            for (int q = 0; q < PER; q++) {
                const long j = (long)t.j0 + jj0 + q;
                if (j >= len || !ins[jj0 + q]) continue;
                const int i = i0 + q;
                int mn = lmin[q] < cmin ? lmin[q] : cmin;
                mn = rmin[q] < mn ? rmin[q] : mn;
                int mx = lmax[q] > cmax ? lmax[q] : cmax;
                mx = rmax[q] > mx ? rmax[q] : mx;
                int peak = (j > 2 * k + 10 && W[i - 5] - W[i - k - 5] - W[i] + mn <= -2) || (W[i] + mx >= 2);
                sel[jj0 + q] = (uint8_t)peak;
            }
        }
        }
        __syncthreads();
    }
    {
        constexpr int PT4 = (TILE + BT - 1) / BT;
        uint8_t fo[PT4];
#pragma unroll
        for (int q = 0; q < PT4; q++) {                    // the tile's own bytes first, then the stores (a load per store and its wait: 8 round trips)
            const long j = (long)t.j0 + threadIdx.x + q * BT;
            fo[q] = F[j < len ? j : len - 1];
        }
#pragma unroll
        for (int q = 0; q < PT4; q++) {
            const int jj = threadIdx.x + q * BT;
            const long j = (long)t.j0 + jj;
            if (jj < TILE && j < len && ins[jj]) {
                const uint8_t s = sel[jj];
                F[j] = (uint8_t)((fo[q] & 7) | (s << 3) | (1 << 4) | (s << 5));   // peak (inside intervals only), inside, selected
                if (s) atomicAdd(&n_sel, 1);
            }
        }
    }
    __syncthreads();
    // a selected position opens a new peak iff it is the first selected one of its 50-bp bucket (E:296); nothing selected (most tiles
    // that get here): nothing to look for
    if (n_sel) {
        for (int bk = threadIdx.x; bk < TILE / 50; bk += BT) {
            for (int q = 0; q < 50; q++) {
                int jj = bk * 50 + q;
                if (sel[jj]) {
                    F[(long)t.j0 + jj] |= 1 << 6;
                    atomicAdd(&n_new, 1);
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        tile_count[tile] = (uint32_t)n_new;
        tile_sel[tile] = (uint32_t)n_sel;
        if (n_sel) atomicAdd(n_selected, (unsigned long long)n_sel);
    }
}

// ---- B4: in-place exclusive scan of tile_count[0..n); total lands in tile_count[n].  One 1024-thread block walks the
// array in coalesced pieces of 1024 with a running carry (wave shuffles + 16 wave totals in LDS per piece).
__global__ void __launch_bounds__(1024) tile_scan(uint32_t* __restrict__ v, long n) {
    __shared__ unsigned long long wtot[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (long base = 0; base < n; base += 1024) {
        const long i = base + threadIdx.x;
        const unsigned long long x = i < n ? v[i] : 0ull;
        unsigned long long incl = x;
        for (int d = 1; d < 64; d <<= 1) {
            unsigned long long o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        unsigned long long woff = 0;
        for (int q = 0; q < wv; q++) woff += wtot[q];
        const unsigned long long carry = carry_s;
        const unsigned long long excl = carry + woff + incl - x;
        if (i < n) v[i] = (uint32_t)(excl > 0xffffffffull ? 0xffffffffull : excl);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) v[n] = (uint32_t)(carry_s > 0xffffffffull ? 0xffffffffull : carry_s);
}

// the same scan for long arrays (configs[2] has 6.5 M tiles: 10 ms in one workgroup): chunk sums, their scan, chunk-local scans
constexpr int SCAN_PER = 16, SCAN_CHUNK = 1024 * SCAN_PER;
__device__ __forceinline__ unsigned long long block_excl_sum_u64(unsigned long long v, unsigned long long* sh /*[16]*/, unsigned long long* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    unsigned long long off = 0, all = 0;
    for (int q = 0; q < (int)(blockDim.x >> 6); q++) {
        if (q < wv) off += sh[q];
        all += sh[q];
    }
    __syncthreads();
    if (total) *total = all;
    return off + incl - v;
}
__global__ void __launch_bounds__(1024) tile_chunk_sums(const uint32_t* __restrict__ v, long n, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long sh[16];
    const long b = (long)blockIdx.x * SCAN_CHUNK + (long)threadIdx.x * SCAN_PER;
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) s += b + q < n ? v[b + q] : 0u;
    unsigned long long total;
    block_excl_sum_u64(s, sh, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) chunk_bases(unsigned long long* __restrict__ sums, long n_chunks) {   // in place; total -> sums[n_chunks]
    __shared__ unsigned long long sh[16];
    unsigned long long carry = 0;
    for (long base = 0; base < n_chunks; base += 1024) {
        const long i = base + threadIdx.x;
        const unsigned long long x = i < n_chunks ? sums[i] : 0ull;
        unsigned long long total;
        const unsigned long long ex = block_excl_sum_u64(x, sh, &total);
        if (i < n_chunks) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[n_chunks] = carry;
}
__global__ void __launch_bounds__(1024) tile_chunk_scan(uint32_t* __restrict__ v, long n, const unsigned long long* __restrict__ sums, long n_chunks) {
    __shared__ unsigned long long sh[16];
    const long b = (long)blockIdx.x * SCAN_CHUNK + (long)threadIdx.x * SCAN_PER;
    uint32_t x[SCAN_PER];
    unsigned long long s = 0;
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) { x[q] = b + q < n ? v[b + q] : 0u; s += x[q]; }
    unsigned long long run = sums[blockIdx.x] + block_excl_sum_u64(s, sh, nullptr);
#pragma unroll
    for (int q = 0; q < SCAN_PER; q++) {
        if (b + q < n) v[b + q] = (uint32_t)(run > 0xffffffffull ? 0xffffffffull : run);
        run += x[q];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long total = sums[n_chunks];
        v[n] = (uint32_t)(total > 0xffffffffull ? 0xffffffffull : total);
    }
}

// ---- B5
__global__ void __launch_bounds__(BT) register_peaks(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                     const RefSource rs, const uint32_t* __restrict__ counts,
                                                     const uint8_t* __restrict__ flags, const uint8_t* __restrict__ nzmask,
                                                     const uint8_t* __restrict__ pstate /* trio-first form: the probe state with its "> 0" records */,
                                                     const uint32_t* __restrict__ tile_base,
                                                     int k, int e, int32_t* __restrict__ loci, uint32_t* __restrict__ peak_kmer,
                                                     uint32_t* __restrict__ prefilter /* nullable */, uint32_t pf_mask, int pf2,
                                                     uint32_t first_id /* 1 under -t N emulation when thread 0 finds no peak, else 0 */, long n_blk) {
    __shared__ int incl[TILE], part[BT];
    if (block2d() >= n_blk) return;
    // the tiles from the LAST one down (round 6): ids ascend with the tiles and the larger id wins a slot, so the tiles that run first
    // now hold the ids that stay -- a later, smaller id finds `peak_kmer[h] >= id` with a plain load and leaves the slot alone, which
    // is a line read instead of the read-modify-write of an atomic (7.7 G registrations into 4.3 G slots in the CLI's default regime:
    // more than half of them lose).  A stale load only ever shows a SMALLER id: the atomic then decides, as before.
    const long blk = n_blk - 1 - block2d();
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long len = c.len, nk = len - k + 1;
    const uint8_t* F = flags + c.flat_base;
    uint32_t base = tile_base[blk];
    if (tile_base[blk + 1] == base) return;  // no peak in this tile (uniform exit)
    base += first_id;
    constexpr int CH = (TILE + BT - 1) / BT;
    const int b = threadIdx.x * CH, en = b + CH < TILE ? b + CH : TILE;
    int s = 0;
    for (int jj = b; jj < en; jj++) {
        long j = (long)t.j0 + jj;
        s += (j < len) ? (F[j] >> 6) & 1 : 0;
        incl[jj] = s;
    }
    int off = block_excl_sum(s, part);
    for (int jj = b; jj < en; jj++) incl[jj] += off;
    __syncthreads();
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        if (j >= len) break;
        uint8_t f = F[j];
        if (!((f >> 5) & 1)) continue;
        uint32_t id = base + (uint32_t)incl[jj] - 1u;  // merged peaks take the id of their bucket's first peak
        if ((f >> 6) & 1) {
            loci[2 * (long)id] = (int32_t)c.ref_index;
            loci[2 * (long)id + 1] = (int32_t)j;
        }
        if (j < nk) {  // E:247,262; beyond nk the hit array is zero anyway
            const RefKmer km = ref_kmer(rs, c, j, k, e);
            uint32_t nz = nzmask ? nzmask[c.flat_base + j] : 0u, have = nzmask ? 0xffu : 0u;
            if (pstate) {                 // trio-first form (e <= 3): what the probe kernels already know about "count > 0" per hash
                const uint32_t ps = pstate[c.flat_base + j];
                if ((f & 0x82u) == 0x82u) { nz = 7u; have = 7u; }                          // an exact trio bit: every hash reads 3 (the slot-first form leaves no other record of it)
                else if (ps & 0x80u) { nz = (ps >> 4) & 7u; have = 7u; }                    // filled: every hash
                else {
                    const uint32_t known = (ps >> 4) & 7u, is3 = ps & 7u;
                    nz = is3 | (((ps >> 3) & 1u) ? (known & ~is3) : 0u);
                    have = known;                                                           // the others are looked up below
                }
            }
            for (int i = 0; i < e; i++) {
                uint32_t h = ref_hash(rs, km, i);
                // hit > 0 for this hash: the bit the probe kernels recorded, where they did (hashes 8.., and those the single-first /
                // trio-first forms left unprobed or without a record, are probed again)
                if (i < 8 && ((have >> i) & 1u) ? ((nz >> i) & 1u) != 0u : (h != 0 && count_of(counts, h) > 0)) {
                    if (__builtin_nontemporal_load(peak_kmer + h) < id) atomicMax(&peak_kmer[h], id);  // later (larger) id wins
                    if (prefilter) {
                        atomicOr(&prefilter[pf_word(h, pf_mask)], pf_word_bits(h, pf2));
                    }
                }
            }
        }
    }
}

// ---- B5 for a reference shard: the same walk as register_peaks, but the results leave as records for the
// exchange instead of touching peak_kmer: loci of this shard's new peaks (their ids are id_base + local id), and
// one (hash, id) registration per peak position and hash with count > 0.  Order of registrations is irrelevant
// (they are replayed with atomicMax).
__global__ void __launch_bounds__(BT) emit_peaks(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs,
                                                 const RefSource rs, const uint32_t* __restrict__ counts,
                                                 const uint8_t* __restrict__ flags, const uint8_t* __restrict__ nzmask,
                                                 const uint32_t* __restrict__ tile_base,
                                                 int k, int e, uint32_t id_base, int32_t* __restrict__ loci_out,
                                                 uint32_t* __restrict__ regs_out, unsigned long long* __restrict__ n_regs, long n_blk) {
    __shared__ int incl[TILE], part[BT];
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long len = c.len, nk = len - k + 1;
    const uint8_t* F = flags + c.flat_base;
    const uint32_t base = tile_base[blk];
    if (tile_base[blk + 1] == base) return;
    constexpr int CH = (TILE + BT - 1) / BT;
    const int b = threadIdx.x * CH, en = b + CH < TILE ? b + CH : TILE;
    int s = 0;
    for (int jj = b; jj < en; jj++) {
        long j = (long)t.j0 + jj;
        s += (j < len) ? (F[j] >> 6) & 1 : 0;
        incl[jj] = s;
    }
    int off = block_excl_sum(s, part);
    for (int jj = b; jj < en; jj++) incl[jj] += off;
    __syncthreads();
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        long j = (long)t.j0 + jj;
        if (j >= len) break;
        uint8_t f = F[j];
        if (!((f >> 5) & 1)) continue;
        uint32_t lid = base + (uint32_t)incl[jj] - 1u;
        if ((f >> 6) & 1) {
            loci_out[2 * (long)lid] = (int32_t)c.ref_index;
            loci_out[2 * (long)lid + 1] = (int32_t)j;
        }
        if (j < nk) {
            const RefKmer km = ref_kmer(rs, c, j, k, e);
            const uint32_t nz = nzmask ? nzmask[c.flat_base + j] : 0u;
            for (int i = 0; i < e; i++) {
                uint32_t h = ref_hash(rs, km, i);
                if (nzmask && i < 8 ? ((nz >> i) & 1u) != 0u : (h != 0 && count_of(counts, h) > 0)) {
                    unsigned long long slot = atomicAdd(n_regs, 1ull);
                    regs_out[2 * slot] = h;
                    regs_out[2 * slot + 1] = id_base + lid;
                }
            }
        }
    }
}

// replay of gathered registrations on every rank
__global__ void __launch_bounds__(256) replay_regs(const uint32_t* __restrict__ regs, long n, uint32_t* __restrict__ peak_kmer,
                                                   uint32_t* __restrict__ prefilter /* nullable */, uint32_t pf_mask, int pf2) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t h = regs[2 * i], id = regs[2 * i + 1];
        atomicMax(&peak_kmer[h], id);
        if (prefilter) {
            atomicOr(&prefilter[pf_word(h, pf_mask)], pf_word_bits(h, pf2));
        }
    }
}

// ---- B5 by partition (round 6).  In the regime `localhgt bkp` runs by default (--sample 2000000000 on a half-of-the-
// catalogue sample) the peak set is dense: 55 M peaks register 7.7 G (hash, id) pairs into a 16 GiB table, and register_peaks above is
// what the fabric allows for that many random line fills and write-backs (12 G line transactions, 261 ms).  "The larger id wins a
// slot" does not depend on the order of the registrations (E:247-267: every thread of the reference overwrites in its own order, the
// emulation's rule is the last writer = the largest id), so they are routed by slot like phase A's keys:
//   rg_emit   a workgroup takes RG_TILES tiles, a wave each, and goes over their selected positions twice -- first counting its records
//             per top-8-bit bucket of the slot in LDS, then, with one global cursor bump per bucket, writing each record
//             (slot << 32 | id) to its place: the 8-byte stores of one run land next to each other within microseconds and leave L2
//             as whole lines (the pass is bound by their number: 220 G stores/s);
//   rg_split  a slab of 16 Ki records of one bucket, held in registers, goes to the 512 sub-buckets of the next nine bits the same way;
//   rg_apply  a final bucket = 2^(k-17) slots of peak_kmer (128 KiB at k = 32) lives in LDS while its records are applied with
//             ds_max_u32, and goes back as it came: one sequential sweep of the table per chunk instead of a line fill per record.
// A slot is min(forward, reverse complement), so its density falls linearly (2(1-x)): bucket q of nb expects the share
// (2(nb-q)-1)/nb^2 of the records, regions are laid out by that expectation + 1/16 + a pad (rg_base), and a record that finds its
// region full (repeats) is applied to the table at once -- rg_apply loads its slice behind the scatters, so the table is exact whatever
// overflows.  The records of the whole reference need 17 bytes each in flight; what the device can spare decides the number of CHUNKS
// (runs of tile groups with equal shares of the selected positions, from interval_select's per-tile counts), each with its own emit /
// split / apply and table sweep.  Default regime, 7.68 G records in 3 chunks of 43.6 GB: 82 + 37 + 29 = 148 ms instead of 261
// (profiles/r06/kernel_stats_default_sample_registry_by_partition.txt).
constexpr int RG_TILES = 8, RG_BT = 512;
constexpr int RG_B1 = 256, RG_B2 = 512, RG_L2 = 17;            // 8 + 9 bits of fan-out
constexpr int RG_SBT = 1024, RG_PER = 16, RG_SLAB = RG_SBT * RG_PER;
constexpr unsigned long long RG_PAD1 = 8192, RG_PAD2 = 64;
// first record of bucket q of 2^nb_log: floor(ucap * q (2 nb - q) / nb^2) + q * pad, even (16-byte aligned) -- integer arithmetic, the same
// value in every kernel and on the host
__host__ __device__ inline unsigned long long rg_base(unsigned long long ucap, unsigned long long q, int nb_log, unsigned long long pad) {
    const unsigned long long nb = 1ull << nb_log, A = q * (2 * nb - q);
    const int S = 2 * nb_log;
    unsigned long long v;
    if (S < 20) v = (A * ucap) >> S;
    else v = (A * (ucap >> 20) + ((A * (ucap & 0xfffffull)) >> 20)) >> (S - 20);
    return (v + q * pad) & ~1ull;
}

// selected positions in front of every group of RG_TILES tiles (three small kernels: sums of 1024 groups, their prefix, the groups)
__device__ __forceinline__ unsigned long long rg_group_sel(const uint32_t* __restrict__ tile_sel, long n_tiles, long g) {
    unsigned long long x = 0;
    for (int q = 0; q < RG_TILES; q++) { const long t = g * RG_TILES + q; if (t < n_tiles) x += tile_sel[t]; }
    return x;
}
__global__ void __launch_bounds__(1024) rg_group_sums(const uint32_t* __restrict__ tile_sel, long n_tiles, long n_groups, unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long sh[16];
    const long g = (long)blockIdx.x * 1024 + threadIdx.x;
    unsigned long long total;
    block_excl_sum_u64(g < n_groups ? rg_group_sel(tile_sel, n_tiles, g) : 0ull, sh, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) rg_group_prefix(const uint32_t* __restrict__ tile_sel, long n_tiles, long n_groups, const unsigned long long* __restrict__ sums,
                                                        unsigned long long* __restrict__ pre) {
    __shared__ unsigned long long sh[16];
    const long g = (long)blockIdx.x * 1024 + threadIdx.x;
    const unsigned long long ex = block_excl_sum_u64(g < n_groups ? rg_group_sel(tile_sel, n_tiles, g) : 0ull, sh, nullptr);
    if (g < n_groups) pre[g] = sums[blockIdx.x] + ex;
    if (g == 0) pre[n_groups] = sums[gridDim.x];
}
// first group of every chunk: the first g with pre[g] >= c * sel_per_chunk (a group belongs to the chunk pre[g] / sel_per_chunk)
__global__ void rg_chunk_bounds(const unsigned long long* __restrict__ pre, long n_groups, unsigned long long sel_per_chunk, int n_chunks, long* __restrict__ bounds) {
    const int c = threadIdx.x;
    if (c > n_chunks) return;
    if (c == n_chunks) { bounds[c] = n_groups; return; }
    const unsigned long long want = (unsigned long long)c * sel_per_chunk;
    long lo = 0, hi = n_groups;            // first g in [0, n_groups] with pre[g] >= want
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (pre[mid] >= want) hi = mid; else lo = mid + 1; }
    bounds[c] = lo;
}

// A WAVE per tile.  Everything a tile's walk reads from memory is fetched in the wave's prologue, in batches of independent loads: the
// flag bytes (their ballots give the list of SELECTED positions -- bit 5; 44 % of an active tile's in the default regime --, the ballot
// words of "opens a peak", bit 6, with their running counts: the ids, and of "exact trio", bits 7 and 1), the probe-state (or nz) byte of
// every selected position, and the tile's words of the packed planes.  Both passes then run dense over the list out of LDS -- pass 0
// counts the group's records per bucket, one cursor bump per bucket reserves the runs, pass 1 hashes again and writes -- with no load
// to wait for in pass 1, so its scattered stores are never waited for either.
constexpr int RG_WORDS = (TILE + 63) / 64;     // 32 ballot words per tile
constexpr int RG_PW = (TILE + 31) / 32 + 3;    // plane words a tile's windows can touch
static_assert(RG_TILES == RG_BT / 64, "a tile per wave");
// STAGED: the packed reference (the index form reads its stored hashes inside the loops); E: the number of hashes when it is 1 .. 3, so
// that the loop over them unrolls and the masks stay in scalar registers (0: any e) -- with the general loop the compiler waits for
// every outstanding memory operation, the previous records' stores included, before each hash
template <bool STAGED, int E>
__global__ void __launch_bounds__(RG_BT, 4) rg_emit(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, const RefSource rs,
                                                 const uint32_t* __restrict__ counts, const uint8_t* __restrict__ flags, const uint8_t* __restrict__ nzmask,
                                                 const uint8_t* __restrict__ pstate, const uint32_t* __restrict__ tile_base, int k, int e,
                                                 int32_t* __restrict__ loci, uint32_t* __restrict__ peak_kmer, uint32_t first_id, long n_tiles, long g_first, long g_end,
                                                 const unsigned long long* __restrict__ group_pre, unsigned long long sel_per_chunk, unsigned long long chunk,
                                                 unsigned long long ucap, unsigned long long pad1, unsigned long long* __restrict__ cur1, unsigned long long* __restrict__ buf1,
                                                 unsigned long long* __restrict__ n_direct,
                                                 uint32_t* __restrict__ prefilter /* nullable: the vote's bitmap, where a forced run has one */, uint32_t pf_mask, int pf2,
                                                 int ablate /* timing only (LHGT_RG_ABLATE): 1 no record is stored, 2 no pass 1, 3 the prologue alone, 4 no cursor bumps */) {
    __shared__ unsigned short list[RG_TILES][RG_WORDS * 64];
    __shared__ uint8_t laux[RG_TILES][RG_WORDS * 64];
    __shared__ uint32_t lpl[RG_TILES][3 * RG_PW];             // hi / lo interleaved, then the not-a-base words
    __shared__ unsigned long long m6[RG_TILES][RG_WORDS], m82[RG_TILES][RG_WORDS];
    __shared__ unsigned short c6[RG_TILES][RG_WORDS];
    __shared__ int n_list[RG_TILES];
    __shared__ unsigned int hist[RG_B1];
    __shared__ unsigned long long gpos[RG_B1], glim[RG_B1];
    const long g = g_first + block2d();
    if (g >= g_end) return;
    const unsigned long long s0 = group_pre[g];
    if (group_pre[g + 1] == s0 || s0 / sel_per_chunk != chunk) return;     // nothing selected here (or not this chunk's group)
    for (int b = threadIdx.x; b < RG_B1; b += RG_BT) hist[b] = 0;
    const int lane = threadIdx.x & 63, ti = threadIdx.x >> 6;
    const int shift1 = k - 8;
    const long blk = g * RG_TILES + ti;
    uint32_t base = 0;
    bool live = blk < n_tiles;
    if (live) { base = tile_base[blk]; live = tile_base[blk + 1] != base; }   // no peak in this tile (wave-uniform)
    base += first_id;
    TileDev t{};
    ContigDev c{};
    if (live) { t = tiles[blk]; c = contigs[t.contig]; }
    const long len = c.len, nk = len - k + 1;
    const uint8_t* aux = pstate ? pstate : nzmask;       // one byte per position, or none (single-first form: every count is looked up)
    const uint64_t x0 = c.flat_base + (uint64_t)t.j0;    // flat position of the tile's first
    if (live) {
        const uint8_t* F = flags + x0;
        const uint8_t* A = aux ? aux + x0 : nullptr;
        const long left = len - (long)t.j0;              // positions of the contig from the tile's first on
        const int last = (int)((left < TILE ? left : (long)TILE) - 1);
        unsigned int r5 = 0, r6 = 0;
#pragma unroll
        for (int half = 0; half < 2; half++) {
            uint8_t fb[RG_WORDS / 2], ab[RG_WORDS / 2];
#pragma unroll
            for (int q = 0; q < RG_WORDS / 2; q++) {                 // unconditional loads on an index clamped into the tile
                const int jj = (half * (RG_WORDS / 2) + q) * 64 + lane;
                fb[q] = F[jj <= last ? jj : last];
                if (jj > last) fb[q] = 0;
            }
#pragma unroll
            for (int q = 0; q < RG_WORDS / 2; q++) {
                const int jj = (half * (RG_WORDS / 2) + q) * 64 + lane;
                ab[q] = A ? A[jj <= last ? jj : last] : (uint8_t)0;
            }
#pragma unroll
            for (int q = 0; q < RG_WORDS / 2; q++) {
                const int it = half * (RG_WORDS / 2) + q;
                const unsigned long long b5 = __ballot((fb[q] >> 5) & 1), b6 = __ballot((fb[q] >> 5) & (fb[q] >> 6) & 1), b82 = __ballot((fb[q] & 0x82u) == 0x82u);
                if ((fb[q] >> 5) & 1) {
                    const unsigned int slot = r5 + (unsigned int)__popcll(b5 & ((1ull << lane) - 1ull));
                    list[ti][slot] = (unsigned short)(it * 64 + lane);
                    laux[ti][slot] = ab[q];
                }
                if (lane == 0) { m6[ti][it] = b6; m82[ti][it] = b82; c6[ti][it] = (unsigned short)r6; }
                r5 += (unsigned int)__popcll(b5);
                r6 += (unsigned int)__popcll(b6);
            }
        }
        if (lane == 0) n_list[ti] = (int)r5;
        if (STAGED) {
            // plane words (x0 >> 5) .. of the tile, as far as the windows of its positions reach (ref_kmer reads word q and q + 1 of each plane)
            const uint64_t w0 = x0 >> 5, w_last = ((x0 + (uint64_t)last) >> 5) + 1;
            for (int wi = lane; wi < RG_PW; wi += 64) {
                const uint64_t w = w0 + (uint64_t)wi <= w_last ? w0 + (uint64_t)wi : w_last;
                const uint2 hl = *(const uint2*)(rs.planes + 2 * w);
                lpl[ti][2 * wi] = hl.x;
                lpl[ti][2 * wi + 1] = hl.y;
                lpl[ti][2 * RG_PW + wi] = rs.planes[2 * rs.plane_words + w];
            }
        }
    }
    __syncthreads();
    unsigned int direct = 0;
    const int n_sel = live && ablate != 3 ? n_list[ti] : 0;
    const int nk_rel = (int)(nk - (long)t.j0 < (long)TILE ? nk - (long)t.j0 : (long)TILE);      // tile-relative positions below this have a k-mer (E:247,262; beyond nk the hit array is zero anyway)
    const int xr = (int)(x0 & 31);
    // k-mer windows of tile-relative position jj out of the staged plane words
    auto kmer_at = [&](int jj) {
        RefKmer km{};
        const int q = (xr + jj) >> 5, r = (xr + jj) & 31;
        const uint32_t* P = &lpl[ti][2 * q];
        const uint32_t* NB = &lpl[ti][2 * RG_PW + q];
        km.whi = window32(P[0], P[2], r) >> (32 - k);
        km.wlo = window32(P[1], P[3], r) >> (32 - k);
        km.valid = (window32(NB[0], NB[1], r) >> (32 - k)) == 0u;
        km.rhi = brev_k(km.whi, k);
        km.rlo = brev_k(km.wlo, k);
        return km;
    };
    // "count > 0" per hash from the state byte (as in register_peaks): nz where the probe kernels left a record, have = which hashes they did
    auto decode = [&](uint32_t ax, bool trio, uint32_t* have) -> uint32_t {
        if (!pstate) { *have = nzmask ? 0xffu : 0u; return nzmask ? ax : 0u; }
        const uint32_t known = (ax >> 4) & 7u, is3 = ax & 7u;
        const bool filled = (ax & 0x80u) != 0u;
        const uint32_t part = is3 | (((ax >> 3) & 1u) ? (known & ~is3) : 0u);
        *have = trio || filled ? 7u : known;
        return trio ? 7u : filled ? known : part;
    };
    // RESOLVED (packed reference, e <= 3): pass 0 also settles the look-ups -- where the probe kernels left no record of a hash its count
    // is read from the table, a random access -- and leaves "hash i registers" in the state byte's place, so that pass 1 loads nothing:
    // with the look-ups inside pass 1 every one of them also waited for the stores of the positions before it (gfx950 counts loads and
    // stores in one counter; 126 ms instead of 85 for 7.7 G records).  The loads of one step of pass 0 are independent and leave together.
    constexpr bool RESOLVED = STAGED && E >= 1 && E <= 3;
    constexpr int EN = E ? E : 1;
    // one dense pass over the selected positions: PASS 0 counts per bucket, PASS 1 writes
    auto walk = [&](auto pass_tag) {
        constexpr int PASS = decltype(pass_tag)::value;
        for (int n0 = 0; n0 < n_sel; n0 += 64) {
            const int n = n0 + lane;
            const bool in = n < n_sel;
            const int jj = in ? list[ti][n] : 0, w = jj >> 6, bit = jj & 63;
            uint32_t id = 0;
            if (PASS == 1 && in) {
                const unsigned long long w6 = m6[ti][w];
                id = base + (uint32_t)c6[ti][w] + (uint32_t)__popcll(w6 & ((2ull << bit) - 1ull)) - 1u;  // merged peaks take the id of their bucket's first peak
                if ((w6 >> bit) & 1ull) {
                    loci[2 * (long)id] = (int32_t)c.ref_index;
                    loci[2 * (long)id + 1] = (int32_t)(t.j0 + jj);
                }
            }
            const bool has_kmer = in && jj < nk_rel;
            const uint32_t ax = in ? laux[ti][n] : 0u;
            if (RESOLVED && PASS == 1 && ax == 0u) continue;
            if (!RESOLVED && !has_kmer) continue;
            const RefKmer km = STAGED ? kmer_at(jj) : ref_kmer(rs, c, (long)t.j0 + jj, k, e);
            uint32_t have = 7u, nz = ax;
            if (!(RESOLVED && PASS == 1)) nz = decode(ax, (m82[ti][w] >> bit) & 1ull, &have);
            if constexpr (RESOLVED) {
                uint32_t h[EN];
#pragma unroll
                for (int i = 0; i < EN; i++) h[i] = km.valid ? hash_from_windows(km.whi, km.wlo, km.rhi, km.rlo, rs.hp.mask[i]) : 0u;
                if (PASS == 0) {
                    uint32_t need = has_kmer ? ~have & ((1u << EN) - 1u) : 0u;
#pragma unroll
                    for (int i = 0; i < EN; i++) if (h[i] == 0u) need &= ~(1u << i);
                    if (__any(need != 0u)) {
                        uint32_t word[EN];
#pragma unroll
                        for (int i = 0; i < EN; i++) word[i] = counts[((need >> i) & 1u) ? h[i] >> 4 : 0u];          // unconditional: the loads of a step leave together
#pragma unroll
                        for (int i = 0; i < EN; i++)
                            if (((need >> i) & 1u) && ((word[i] >> ((h[i] & 15u) * 2u)) & 3u) != 0u) nz |= 1u << i;
                    }
                    nz = has_kmer ? nz & ((1u << EN) - 1u) : 0u;
                    if (in) laux[ti][n] = (uint8_t)nz;
                }
#pragma unroll
                for (int i = 0; i < EN; i++) {
                    if (!((nz >> i) & 1u)) continue;
                    const uint32_t b = h[i] >> shift1;
                    if (PASS == 0) atomicAdd(&hist[b], 1u);
                    else {
                        const unsigned long long pos = gpos[b] + atomicAdd(&hist[b], 1u);
                        if (ablate == 1) { direct += (unsigned int)(pos & 1); continue; }
                        if (pos < glim[b]) buf1[pos] = ((unsigned long long)h[i] << 32) | id;
                        else { atomicMax(&peak_kmer[h[i]], id); direct++; }
                        if (prefilter) atomicOr(&prefilter[pf_word(h[i], pf_mask)], pf_word_bits(h[i], pf2));
                    }
                }
            } else {
                for (int i = 0; i < e; i++) {
                    const uint32_t h = ref_hash(rs, km, i);
                    if (i < 8 && ((have >> i) & 1u) ? ((nz >> i) & 1u) != 0u : (h != 0 && count_of(counts, h) > 0)) {
                        const uint32_t b = h >> shift1;
                        if (PASS == 0) atomicAdd(&hist[b], 1u);
                        else {
                            const unsigned long long pos = gpos[b] + atomicAdd(&hist[b], 1u);
                            if (pos < glim[b]) buf1[pos] = ((unsigned long long)h << 32) | id;
                            else { atomicMax(&peak_kmer[h], id); direct++; }
                            if (prefilter) atomicOr(&prefilter[pf_word(h, pf_mask)], pf_word_bits(h, pf2));
                        }
                    }
                }
            }
        }
    };
    walk(std::integral_constant<int, 0>{});
    __syncthreads();
    for (int b = threadIdx.x; b < RG_B1; b += RG_BT) {
        const unsigned int n = hist[b];
        hist[b] = 0;
        gpos[b] = n ? rg_base(ucap, b, 8, pad1) + (ablate == 4 ? (unsigned long long)(blockIdx.x & 1023) * n : atomicAdd(&cur1[b], (unsigned long long)n)) : 0ull;
        glim[b] = rg_base(ucap, b + 1, 8, pad1);
    }
    __syncthreads();
    if (ablate != 2) walk(std::integral_constant<int, 1>{});
    if (direct) atomicAdd(n_direct, (unsigned long long)direct);
}

__global__ void __launch_bounds__(RG_SBT) rg_split(const unsigned long long* __restrict__ buf1, const unsigned long long* __restrict__ cur1, unsigned long long ucap,
                                                   unsigned long long pad1, unsigned long long pad2, int k,
                                                   unsigned long long* __restrict__ cur2, unsigned long long* __restrict__ buf2, uint32_t* __restrict__ peak_kmer,
                                                   unsigned long long* __restrict__ n_direct) {
    __shared__ unsigned long long cnt[RG_B1], before[RG_B1 + 1];
    __shared__ unsigned int hist[RG_B2];
    __shared__ unsigned long long gpos[RG_B2], glim[RG_B2];
    for (int b = threadIdx.x; b < RG_B1; b += RG_SBT) {
        const unsigned long long lo = rg_base(ucap, b, 8, pad1), cap = rg_base(ucap, b + 1, 8, pad1) - lo;
        const unsigned long long n = cur1[b];
        cnt[b] = n < cap ? n : cap;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int b = 0; b < RG_B1; b++) { before[b] = run; run += (cnt[b] + RG_SLAB - 1) / RG_SLAB; }
        before[RG_B1] = run;
    }
    __syncthreads();
    const unsigned long long items = before[RG_B1];
    const int shift2 = k - RG_L2;
    unsigned int direct = 0;
    for (unsigned long long item = blockIdx.x; item < items; item += gridDim.x) {
        int lo = 0, hi = RG_B1;                 // the bucket of this slab: the last q with before[q] <= item
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (before[mid] <= item) lo = mid; else hi = mid; }
        const int q = lo;
        const unsigned long long slab = item - before[q], first = slab * RG_SLAB;
        const unsigned long long n = cnt[q] - first < (unsigned long long)RG_SLAB ? cnt[q] - first : (unsigned long long)RG_SLAB;
        const unsigned long long* src = buf1 + rg_base(ucap, q, 8, pad1) + first;
        for (int s = threadIdx.x; s < RG_B2; s += RG_SBT) hist[s] = 0;
        __syncthreads();
        unsigned long long rec[RG_PER];
        unsigned int rank[RG_PER];
#pragma unroll
        for (int r = 0; r < RG_PER; r++) {
            const unsigned long long i = (unsigned long long)r * RG_SBT + threadIdx.x;
            rec[r] = i < n ? __builtin_nontemporal_load(src + i) : ~0ull;
        }
#pragma unroll
        for (int r = 0; r < RG_PER; r++)
            if (rec[r] != ~0ull) rank[r] = atomicAdd(&hist[((uint32_t)(rec[r] >> 32) >> shift2) & (RG_B2 - 1)], 1u);
        __syncthreads();
        for (int s = threadIdx.x; s < RG_B2; s += RG_SBT) {
            const unsigned int m = hist[s];
            const unsigned long long fb = (unsigned long long)q * RG_B2 + s;
            gpos[s] = m ? rg_base(ucap, fb, RG_L2, pad2) + atomicAdd(&cur2[fb], (unsigned long long)m) : 0ull;
            glim[s] = rg_base(ucap, fb + 1, RG_L2, pad2);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RG_PER; r++)
            if (rec[r] != ~0ull) {
                const uint32_t h = (uint32_t)(rec[r] >> 32);
                const int s = (h >> shift2) & (RG_B2 - 1);
                const unsigned long long pos = gpos[s] + rank[r];
                if (pos < glim[s]) buf2[pos] = rec[r];       // (id and slot bits as two arrays, 6 bytes: the 2-byte scattered stores made this pass 2.8 x slower)
                else { atomicMax(&peak_kmer[h], (uint32_t)rec[r]); direct++; }
            }
        __syncthreads();
    }
    if (direct) atomicAdd(n_direct, (unsigned long long)direct);
}

__global__ void __launch_bounds__(1024) rg_apply(const unsigned long long* __restrict__ buf2, const unsigned long long* __restrict__ cur2,
                                                 unsigned long long ucap, unsigned long long pad2, int k, uint32_t* __restrict__ peak_kmer, long n_final) {
    extern __shared__ uint32_t rg_tab[];
    const long fb = block2d();
    if (fb >= n_final) return;
    const unsigned long long lo = rg_base(ucap, (unsigned long long)fb, RG_L2, pad2), cap = rg_base(ucap, (unsigned long long)fb + 1, RG_L2, pad2) - lo;
    unsigned long long n = cur2[fb];
    if (n > cap) n = cap;
    if (n == 0) return;
    const int sb = k - RG_L2;
    const uint32_t slots = 1u << sb, mask = slots - 1u;
    uint32_t* T = peak_kmer + ((size_t)fb << sb);
    const unsigned long long* R = buf2 + lo;
    if (n * 64 < slots) {                       // a handful of records: not worth the slice
        for (unsigned long long i = threadIdx.x; i < n; i += 1024) { const unsigned long long rec = R[i]; atomicMax(&T[(uint32_t)(rec >> 32) & mask], (uint32_t)rec); }
        return;
    }
    for (uint32_t i = threadIdx.x * 4; i < slots; i += 4096) *(uint4*)&rg_tab[i] = *(const uint4*)&T[i];
    __syncthreads();
    unsigned long long i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        const unsigned long long r0 = __builtin_nontemporal_load(R + i), r1 = __builtin_nontemporal_load(R + i + 1024), r2 = __builtin_nontemporal_load(R + i + 2048),
                                 r3 = __builtin_nontemporal_load(R + i + 3072);
        atomicMax(&rg_tab[(uint32_t)(r0 >> 32) & mask], (uint32_t)r0);
        atomicMax(&rg_tab[(uint32_t)(r1 >> 32) & mask], (uint32_t)r1);
        atomicMax(&rg_tab[(uint32_t)(r2 >> 32) & mask], (uint32_t)r2);
        atomicMax(&rg_tab[(uint32_t)(r3 >> 32) & mask], (uint32_t)r3);
    }
    for (; i < n; i += 1024) { const unsigned long long rec = R[i]; atomicMax(&rg_tab[(uint32_t)(rec >> 32) & mask], (uint32_t)rec); }
    __syncthreads();
    for (uint32_t j = threadIdx.x * 4; j < slots; j += 4096) *(uint4*)&T[j] = *(const uint4*)&rg_tab[j];
}

// ---- B1 for a sparse table with the reference resident across samples ("slot-first", round 5).  The trio-first kernel above asks
// the table ONE question per reference position before anything else -- does the slot of its first hash read 3? -- and on a sparse
// table that first probe is all most positions ever cost: 13 G random 128-byte line fills for 13 Gbase, 232 ms at the fabric's line
// rate whatever the sample.  The question does not depend on the sample, only its answer does.  So a context that scans a second
// sample against the same resident reference turns the question round, once: the SLOT LIST holds every position with a k-mer,
// grouped by the top bits of ONE of its hashes (slot_list_key below) -- bucket b = the 2^14 slots [b << 14, (b + 1) << 14), an
// entry = (low 14 bits of the slot, flat position): 6 bytes, two arrays (u32 low position bits; u16 slot bits | position bits
// 32-33 << 14).  A scan then streams the list (78 GB for 13 Gbase, 14 ms) next to each bucket's 4 KiB of counters in LDS, and only
// the positions whose slot reads 3 -- a fifth -- go on to their other hashes: one line for the k-mer's hashes or bases, then 1 + f
// probes.  Only positions whose e hashes ALL read 3 are written (flags 0x83 = single, trio, exact): the flags are cleared
// beforehand, the probe-state bytes of the tiles that go on to the fill too (clear_pstate_tiles), and "nothing known" is what
// ref_flags_fill and register_peaks take a zero byte for -- the tiles near a candidate window get every probe from the fill, an
// exact trio bit tells register_peaks that all counts are > 0.  The trio bit is exact everywhere, which is all window_trio reads.
// Same peaks, ids and votes as every other form (the form tests run it next to them).  Deep focused sample on 13 Gbase (packed):
// the probe kernel 325 -> 150 ms, phase B 340 -> 181 ms (tools/slot_list_leg.py); 7.0 G line fills where trio-first makes 17.4 G.
constexpr int SL_BITS = 14;
constexpr uint32_t SL_SLOTS = 1u << SL_BITS;
// A position is listed under the LARGEST of its e hashes.  A hash is min(fwd, rc) of two uniform words, so slots thin out towards the
// top of the table (density 2(1 - x)) and so do the reads' k-mers: the largest of three hashes addresses the slot least likely to
// read 3 -- 19 % of the positions go on where 27 % would under hash 0 (deep focused sample, 18 % of all slots at 3).  Any hash of an
// invalid k-mer is 0 (quirk Q6), then all are.
__device__ __forceinline__ uint32_t slot_list_key(const RefSource& rs, const RefKmer& km, int e, bool smallest) {
    uint32_t h = ref_hash(rs, km, 0);
    for (int i = 1; i < e && i < 3; i++) {
        const uint32_t g = ref_hash(rs, km, i);
        if (smallest) h = g != 0u && (h == 0u || g < h) ? g : h;     // the smallest hash that is one (0 = no hash, E:936-941)
        else h = g > h ? g : h;
    }
    return h;
}
// The list for a SATURATED table (round 5, "slot-single") turns the same idea round: there the single-first form wants to know of
// every position whether ANY of its hashes reads 3, almost all do, and the hash likeliest to is the smallest.  Listed under their
// smallest hash, the positions whose slot reads 3 are settled -- `single`, which is what the flags are preset to -- without being
// touched, and only the few per cent whose slot does not are followed to their other hashes (ref_single_slots).

__global__ void __launch_bounds__(BT) slot_list_hist(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, const RefSource rs,
                                                     int k, int e, int smallest, uint32_t* __restrict__ hist /* [nb], then [nb]: k-mers whose hashes are all 0 */, long nb, long n_blk,
                                                     int stride /* > 1: every stride-th tile only -- a SAMPLE of the histogram, slot_list_build */) {
    const long blk = block2d() * stride;
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        const long j = (long)t.j0 + jj;
        if (j >= nk) break;
        const RefKmer km = ref_kmer(rs, c, j, k, e);
        const uint32_t h = slot_list_key(rs, km, e, smallest != 0);
        if (h != 0) atomicAdd(&hist[h >> SL_BITS], 1u);     // hash 0 = invalid (E:936-941): never at 3, not listed
        else if (!rs.index && km.valid) atomicAdd(&hist[nb], 1u);   // a k-mer all of whose hashes are 0 (2^-93 each): no entry speaks for it
    }
}
// one workgroup: hist -> exclusive offsets (u64), hist cleared to serve as the fill's cursors
// scale > 1: hist is a sample of every scale-th tile; a bucket's region is its scaled count + six standard deviations of that estimate
// (the sampled count is Poisson: scale * sqrt(h)) + a pad -- 7.5 % more than the entries at 13 Gbase and scale 8; a bucket that still
// runs over (repeats that the sample missed) makes the build fall back to the exact histogram
__device__ __forceinline__ unsigned long long slot_list_cap(uint32_t h, int scale) {
    return scale > 1 ? (unsigned long long)h * (unsigned)scale + (unsigned long long)(6.0f * (float)scale * sqrtf((float)h + 1.0f)) + 64ull : (unsigned long long)h;
}
__global__ void __launch_bounds__(1024) slot_list_offsets(uint32_t* __restrict__ hist, long nb, unsigned long long* __restrict__ off, int scale) {
    __shared__ unsigned long long part[1024];
    const long per = (nb + 1023) / 1024, b0 = (long)threadIdx.x * per, b1 = b0 + per < nb ? b0 + per : nb;
    unsigned long long s = 0;
    for (long b = b0; b < b1; b++) s += slot_list_cap(hist[b], scale);
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; i++) { const unsigned long long v = part[i]; part[i] = run; run += v; }
        off[nb] = run;
    }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (long b = b0; b < b1; b++) { off[b] = run; run += slot_list_cap(hist[b], scale); hist[b] = 0u; }
}
// regions sized from a sample: where each bucket's entries end (its cursor), and how many there are in all
__global__ void __launch_bounds__(256) slot_list_ends(const unsigned long long* __restrict__ off, const uint32_t* __restrict__ cur, long nb,
                                                      unsigned long long* __restrict__ end, unsigned long long* __restrict__ total) {
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long n = 0;
    if (b < nb) { n = cur[b]; end[b] = off[b] + n; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d, 64);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(total, n);
}
__global__ void __launch_bounds__(BT) slot_list_fill(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, const RefSource rs,
                                                     int k, int e, int smallest, const unsigned long long* __restrict__ off, uint32_t* __restrict__ cur,
                                                     uint32_t* __restrict__ lo, uint16_t* __restrict__ hi, uint32_t* __restrict__ mid /* nullable */, long n_blk,
                                                     uint32_t* __restrict__ est /* nullable; regions sized from a sample: [0] set when a bucket runs over its region,
                                                                                  [1] the k-mers no entry speaks for (the sampled histogram did not see them all) */) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    for (int jj = threadIdx.x; jj < TILE; jj += BT) {
        const long j = (long)t.j0 + jj;
        if (j >= nk) break;
        const RefKmer km = ref_kmer(rs, c, j, k, e);
        const uint32_t h = slot_list_key(rs, km, e, smallest != 0);
        if (h == 0) {
            if (est && !rs.index && km.valid) atomicAdd(est + 1, 1u);
            continue;
        }
        const uint32_t b = h >> SL_BITS;
        const uint32_t rank = atomicAdd(&cur[b], 1u);
        if (est && (unsigned long long)rank >= off[b + 1] - off[b]) { est[0] = 1u; continue; }   // (cur[b] keeps counting: the build is done over)
        const unsigned long long at = off[b] + rank;
        const uint64_t x = c.flat_base + (uint64_t)j;
        lo[at] = (uint32_t)x;
        hi[at] = (uint16_t)((h & (SL_SLOTS - 1u)) | ((uint32_t)(x >> 32) << SL_BITS));
        if (mid) {                                       // the second-largest of the e hashes (the listed one is the largest)
            uint32_t a0 = ref_hash(rs, km, 0), a1 = e > 1 ? ref_hash(rs, km, 1) : 0u, a2 = e > 2 ? ref_hash(rs, km, 2) : 0u;
            if (a1 > a0) { const uint32_t t = a0; a0 = a1; a1 = t; }
            if (a2 > a0) { const uint32_t t = a0; a0 = a2; a2 = t; }
            mid[at] = a2 > a1 ? a2 : a1;
        }
    }
}

// One workgroup of ST threads per bucket.  It streams the bucket's entries 4 ST at a time (four loads per thread in flight), tests
// them against the counters in LDS and appends those whose slot reads 3 to a ring in LDS; whenever the ring holds 4 ST of them every
// thread follows FOUR at once -- their four lines of hashes / bases are requested together, then their four probes -- so the
// dependent chain of a survivor runs with all lanes busy and four requests per lane in flight (a wave that followed 64 survivors one
// per lane, between two loads of its stream, reached 33 G lines/s of the fabric's 56; waves with rings of their own, no barrier: 38;
// U = 8 at a time and 512 or 1024 threads: the same time to a tenth of a millisecond -- what is left is not latency).
template <bool PACKED, int ST, int U, bool MID>
__global__ void __launch_bounds__(ST) ref_flags_slots(const unsigned long long* __restrict__ off, const unsigned long long* __restrict__ end_of /* nullable: off[b + 1] */,
                                                      const uint32_t* __restrict__ lo,
                                                      const uint16_t* __restrict__ hi, const uint32_t* __restrict__ mid /* MID: every entry's second-largest hash */,
                                                      const RefSource rs, const ContigDev* __restrict__ contigs,
                                                      int n_contigs, const uint32_t* __restrict__ counts, int slice_words, int k, int e,
                                                      uint8_t* __restrict__ flags,
                                                      unsigned long long* __restrict__ stats /* nullable: [0] probes of the table, [1] positions followed */, long n_buckets,
                                                      int ablate /* timing only (debug bit 26, LHGT_SLOTS_ABLATE): 1 no survivor is followed, 2 none probes the table */) {
    __shared__ uint32_t slice[SL_SLOTS / 16];
    constexpr int SL_CHUNK = U * ST, SL_RING = 2 * SL_CHUNK;
    __shared__ uint32_t ring_lo[SL_RING];
    __shared__ uint8_t ring_hi[SL_RING];
    __shared__ uint32_t ring_mid[MID ? SL_RING : 1];
    __shared__ uint32_t s_tail;
    const long b = block2d();
    if (b >= n_buckets) return;
    const unsigned long long begin = off[b], end = end_of ? end_of[b] : off[b + 1];
    if (begin == end) return;                                   // uniform
    uint32_t any3 = 0;
    for (int i = threadIdx.x; i < slice_words; i += ST) {
        const uint32_t w = counts[(size_t)b * (SL_SLOTS / 16) + i];
        slice[i] = w;
        any3 |= w & (w >> 1) & 0x55555555u;
    }
    if (threadIdx.x == 0) s_tail = 0u;
    if (!__syncthreads_or(any3 != 0u)) return;                  // no slot of the bucket reads 3: none of its positions goes on
    const int lane = threadIdx.x & 63;
    unsigned long long probes = 0, followed = 0;
    // follow n <= SL_CHUNK survivors from ring position `from`: thread t takes t, t + ST, ...
    auto follow = [&](uint32_t from, uint32_t n) {
        if (ablate == 1) return;
        const uint32_t first = from & (SL_RING - 1);
        const uint64_t x_first = (uint64_t)ring_lo[first] | ((uint64_t)ring_hi[first] << 32);   // n >= 1: a listed position, safe to read for idle lanes
        uint64_t x[U];
        bool on[U], all3[U];
        uint32_t hm[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t idx = threadIdx.x + (uint32_t)u * ST, at = (from + idx) & (SL_RING - 1);
            on[u] = idx < n;
            x[u] = on[u] ? (uint64_t)ring_lo[at] | ((uint64_t)ring_hi[at] << 32) : x_first;
            hm[u] = MID && on[u] ? ring_mid[at] : 0u;
            all3[u] = on[u];
            if (on[u]) followed++;
        }
        if (MID) {
            // the entry brought the position's second-largest hash along: its probe first -- four at once, no look at the reference --
            // and only the third of the positions that pass it (or all, at e = 2: they are trio) go on to their line of bases / hashes
            uint32_t cm[U];
#pragma unroll
            for (int u = 0; u < U; u++) cm[u] = counts[on[u] && hm[u] != 0u && e > 1 && ablate != 2 ? hm[u] >> 4 : 0u];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (on[u] && e > 1) {
                    probes++;
                    all3[u] = hm[u] != 0u && ((cm[u] >> ((hm[u] & 15u) * 2u)) & 3u) == 3u;
                }
            }
        }
        // the four lines, all requested before any is looked at: a k-mer's 2 x 2 interleaved words of bases (packed form) or its e
        // stored hashes (index form); a lane with nothing to ask reads the first survivor's line (one line for all of them: a cache hit)
        const bool need_ref = MID ? e > 2 : e > 1;
        uint32_t h1[U], h2[U];
        if (need_ref) {
            uint32_t w[U][4];
            if (PACKED) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint32_t* p = rs.planes + 2 * ((all3[u] ? x[u] : x_first) >> 5);   // [hi, lo] of word x >> 5, [hi, lo] of the next: 16 bytes, one load (a listed position holds a k-mer: no look at the not-a-base plane)
#pragma unroll
                    for (int q = 0; q < 4; q++) w[u][q] = p[q];
                }
            } else {
                const uint32_t* p[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t xx = all3[u] ? x[u] : x_first;
                    int a = 0, z = n_contigs;                   // last contig with flat_base <= x
                    while (z - a > 1) { const int md = (a + z) >> 1; if (contigs[md].flat_base <= xx) a = md; else z = md; }
                    const ContigDev c = contigs[a];
                    p[u] = rs.index + c.hash_word + (xx - c.flat_base) * (uint64_t)e;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int q = 0; q < 3; q++) w[u][q] = p[u][q < e ? q : e - 1];
            }
            // the hashes, the listed one (the largest) aside -- it reads 3 --, the others sorted, the larger one first
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t a0, a1, a2;
                if (PACKED) {
                    const int r = (int)((all3[u] ? x[u] : x_first) & 31);
                    const uint32_t whi = window32(w[u][0], w[u][2], r) >> (32 - k), wlo = window32(w[u][1], w[u][3], r) >> (32 - k);
                    const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
                    a0 = hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[0]);
                    a1 = e > 1 ? hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[1]) : 0u;
                    a2 = e > 2 ? hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[2]) : 0u;
                } else {
                    a0 = w[u][0];
                    a1 = e > 1 ? w[u][1] : 0u;
                    a2 = e > 2 ? w[u][2] : 0u;
                }
                if (a1 > a0) { const uint32_t t = a0; a0 = a1; a1 = t; }
                if (a2 > a0) { const uint32_t t = a0; a0 = a2; a2 = t; }
                if (a2 > a1) { const uint32_t t = a1; a1 = a2; a2 = t; }
                h1[u] = a1; h2[u] = a2;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) h1[u] = h2[u] = 0u;
        }
        // the probes that are left, four at once each time: a lane with nothing to ask reads word 0 (a line every wave shares -- a
        // cache hit, not a line fill; the branches the compiler made of "only if" probes waited one by one)
        if (!MID) {
            uint32_t c1[U];
#pragma unroll
            for (int u = 0; u < U; u++) c1[u] = counts[e > 1 && on[u] && ablate != 2 ? h1[u] >> 4 : 0u];
#pragma unroll
            for (int u = 0; u < U; u++) {
                all3[u] = on[u] && (e < 2 || (h1[u] != 0u && ((c1[u] >> ((h1[u] & 15u) * 2u)) & 3u) == 3u));
                if (on[u] && e > 1) probes++;
            }
        }
        uint32_t c2[U];
#pragma unroll
        for (int u = 0; u < U; u++) c2[u] = counts[e > 2 && all3[u] && ablate != 2 ? h2[u] >> 4 : 0u];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (e > 2 && all3[u]) {
                probes++;
                all3[u] = h2[u] != 0u && ((c2[u] >> ((h2[u] & 15u) * 2u)) & 3u) == 3u;
            }
            if (all3[u]) flags[x[u]] = 0x83;                    // single, trio, exact -- the one random store of a listed trio position
        }
    };
    uint32_t head = 0;
    for (unsigned long long c0 = begin; c0 < end; c0 += SL_CHUNK) {
        uint32_t lo4[U], hi4[U], mid4[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned long long i = c0 + threadIdx.x + (unsigned long long)u * ST;
            const bool ok = i < end;
            lo4[u] = ok ? lo[i] : 0u;
            hi4[u] = ok ? (uint32_t)hi[i] : 0x10000u;           // bit 16: no entry
            mid4[u] = MID && ok ? mid[i] : 0u;
        }
        unsigned long long m[U];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t sl = hi4[u] & (SL_SLOTS - 1u);
            const bool pass = !(hi4[u] & 0x10000u) && ((slice[sl >> 4] >> ((sl & 15u) * 2u)) & 3u) == 3u;
            m[u] = __ballot(pass);
            mine |= pass ? 1u << u : 0u;
        }
        uint32_t total = 0;
#pragma unroll
        for (int u = 0; u < U; u++) total += (uint32_t)__popcll(m[u]);
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(&s_tail, total);
        base = (uint32_t)__shfl((int)base, 0, 64);
#pragma unroll
        for (int u = 0; u < U; u++) {
            if ((mine >> u) & 1u) {
                const uint32_t at = (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u))) & (SL_RING - 1);
                ring_lo[at] = lo4[u];
                ring_hi[at] = (uint8_t)(hi4[u] >> SL_BITS);
                if (MID) ring_mid[at] = mid4[u];
            }
            base += (uint32_t)__popcll(m[u]);
        }
        __syncthreads();
        const uint32_t tail = s_tail;
        while (tail - head >= (uint32_t)SL_CHUNK) {              // uniform
            follow(head, SL_CHUNK);
            head += SL_CHUNK;
        }
        __syncthreads();                                         // the next appends reuse the ring slots just read
    }
    const uint32_t tail = s_tail;
    if (tail != head) follow(head, tail - head);
    if (stats) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { probes += __shfl_xor(probes, d, 64); followed += __shfl_xor(followed, d, 64); }
        if (lane == 0 && probes) atomicAdd(stats, probes);
        if (lane == 0 && followed) atomicAdd(stats + 1, followed);
    }
}

// ---- B1 for a nearly saturated table with the packed reference and its slot list resident ("slot-single", round 5).  What the later
// steps need from the single-first form (ref_flags_lite) is the exact `single` bit of every position and a lower bound of the trio
// sums.  Here the flags are preset to "single" (0x01), and three small kernels take back what is not:
//   no_kmer_flags      positions without a k-mer -- a not-a-base inside the window, the last k - 1 of a contig -- get 0x80 (exact zeros);
//   ref_single_slots   the list, kept under every position's SMALLEST hash: an entry whose slot reads 3 is done without being touched;
//                      the others (a few per mille) are followed to their other hashes and written exactly: 0x80 when none reads 3,
//                      0x81 (single, not trio) otherwise;
//   ref_trio_runs      the lower bound: 32 consecutive positions of every 250 (64 of any 500-position window, as many as every 8th
//                      position gave; consecutive ones share their lines of bases).  A position not yet marked exact has a smallest
//                      hash that reads 3 -- the sweep just said so -- so only its other hashes are probed, largest first, until one
//                      does not read 3; those whose e hashes all do are written 0x83.
// window_lite, the fill of the tiles it cannot settle and window_good follow as in the single-first form (the probe-state bytes of
// those tiles cleared first: nothing is known of them).  Same peaks, ids and votes; e <= 3, packed form only (which positions hold
// no k-mer is read off the not-a-base plane; the index form would have to stream its 12 bytes per base for it).
__global__ void __launch_bounds__(256) no_kmer_flags(const uint32_t* __restrict__ nb, uint64_t n_words, int k, uint8_t* __restrict__ flags, uint64_t n_pos) {
    const uint64_t w = (uint64_t)block2d() * 256 + threadIdx.x;
    if (w >= n_words) return;
    const uint32_t a = nb[w], b = nb[w + 1];
    if ((a | b) == 0u) return;                               // all 32 windows that start in this word are clean
    for (int r = 0; r < 32; r++) {
        const uint64_t x = w * 32 + (uint64_t)r;
        if (x < n_pos && (window32(a, b, r) >> (32 - k)) != 0u) flags[x] = 0x80;
    }
}
__global__ void __launch_bounds__(256) contig_tail_flags(const ContigDev* __restrict__ contigs, long n_contigs, int k, uint8_t* __restrict__ flags) {
    const long c = (long)blockIdx.x * (256 / 32) + (threadIdx.x >> 5);
    if (c >= n_contigs) return;
    const ContigDev cd = contigs[c];
    const long nk = (long)cd.len - k + 1, from = nk > 0 ? nk : 0;
    for (long j = from + (threadIdx.x & 31); j < (long)cd.len; j += 32) flags[cd.flat_base + j] = 0x80;
}
template <int ST>
__global__ void __launch_bounds__(ST) ref_single_slots(const unsigned long long* __restrict__ off, const unsigned long long* __restrict__ end_of /* nullable: off[b + 1] */,
                                                       const uint32_t* __restrict__ lo,
                                                       const uint16_t* __restrict__ hi, const RefSource rs, const uint32_t* __restrict__ counts,
                                                       int slice_words, int k, int e, uint8_t* __restrict__ flags,
                                                       unsigned long long* __restrict__ stats /* nullable: [0] probes, [1] positions followed */, long n_buckets) {
    constexpr int SL_CHUNK = 4 * ST, SL_RING = 2 * SL_CHUNK;
    __shared__ uint32_t slice[SL_SLOTS / 16];
    __shared__ uint32_t ring_lo[SL_RING];
    __shared__ uint8_t ring_hi[SL_RING];
    __shared__ uint32_t s_tail;
    const long b = block2d();
    if (b >= n_buckets) return;
    const unsigned long long begin = off[b], end = end_of ? end_of[b] : off[b + 1];
    if (begin == end) return;                                   // uniform
    uint32_t not3 = 0;
    for (int i = threadIdx.x; i < slice_words; i += ST) {
        const uint32_t w = counts[(size_t)b * (SL_SLOTS / 16) + i];
        slice[i] = w;
        not3 |= ~(w & (w >> 1)) & 0x55555555u;
    }
    if (threadIdx.x == 0) s_tail = 0u;
    if (!__syncthreads_or(not3 != 0u)) return;                  // every slot of the bucket reads 3: all its positions are `single`
    const int lane = threadIdx.x & 63;
    unsigned long long probes = 0, followed = 0;
    auto follow = [&](uint32_t from, uint32_t n) {
        const uint32_t first = from & (SL_RING - 1);
        const uint64_t x_first = (uint64_t)ring_lo[first] | ((uint64_t)ring_hi[first] << 32);
        uint64_t x[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = threadIdx.x + (uint32_t)u * ST, at = (from + idx) & (SL_RING - 1);
            on[u] = idx < n;
            x[u] = on[u] ? (uint64_t)ring_lo[at] | ((uint64_t)ring_hi[at] << 32) : x_first;
        }
        uint32_t w[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t* p = rs.planes + 2 * (x[u] >> 5);
#pragma unroll
            for (int q = 0; q < 4; q++) w[u][q] = p[q];
        }
        // the listed hash (the smallest that is one) does not read 3; the others, the smaller one first (the likelier 3)
        uint32_t h1[4], h2[4], c1[4], c2[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = (int)(x[u] & 31);
            const uint32_t whi = window32(w[u][0], w[u][2], r) >> (32 - k), wlo = window32(w[u][1], w[u][3], r) >> (32 - k);
            const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
            uint32_t a0 = hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[0]);
            uint32_t a1 = e > 1 ? hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[1]) : 0u;
            uint32_t a2 = e > 2 ? hash_from_windows(whi, wlo, rhi, rlo, rs.hp.mask[2]) : 0u;
            // ascending with the zeros (no hash: never a 3) last; a0 then is the listed one
            a0 = a0 ? a0 : 0xffffffffu; a1 = a1 ? a1 : 0xffffffffu; a2 = a2 ? a2 : 0xffffffffu;
            if (a1 < a0) { const uint32_t t = a0; a0 = a1; a1 = t; }
            if (a2 < a0) { const uint32_t t = a0; a0 = a2; a2 = t; }
            if (a2 < a1) { const uint32_t t = a1; a1 = a2; a2 = t; }
            h1[u] = a1 == 0xffffffffu ? 0u : a1;
            h2[u] = a2 == 0xffffffffu ? 0u : a2;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) c1[u] = counts[on[u] && h1[u] != 0u ? h1[u] >> 4 : 0u];
        bool none[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            none[u] = on[u] && !(h1[u] != 0u && ((c1[u] >> ((h1[u] & 15u) * 2u)) & 3u) == 3u);
            if (on[u]) followed++;
            if (on[u] && h1[u] != 0u) probes++;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) c2[u] = counts[none[u] && h2[u] != 0u ? h2[u] >> 4 : 0u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (none[u] && h2[u] != 0u) {
                probes++;
                none[u] = ((c2[u] >> ((h2[u] & 15u) * 2u)) & 3u) != 3u;
            }
            // exact either way (bit 7): no hash reads 3 -> zeros; another one does -> single, and NOT trio -- which also tells ref_trio_runs
            // that this position's listed hash does not read 3 (it takes every position without the bit for one whose listed hash does)
            if (on[u]) flags[x[u]] = none[u] ? 0x80 : 0x81;
        }
    };
    uint32_t head = 0;
    for (unsigned long long c0 = begin; c0 < end; c0 += SL_CHUNK) {
        uint32_t lo4[4], hi4[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned long long i = c0 + threadIdx.x + (unsigned long long)u * ST;
            const bool ok = i < end;
            lo4[u] = ok ? lo[i] : 0u;
            hi4[u] = ok ? (uint32_t)hi[i] : 0x10000u;           // bit 16: no entry
        }
        unsigned long long m[4];
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t sl = hi4[u] & (SL_SLOTS - 1u);
            const bool pass = !(hi4[u] & 0x10000u) && ((slice[sl >> 4] >> ((sl & 15u) * 2u)) & 3u) != 3u;
            m[u] = __ballot(pass);
            mine |= pass ? 1u << u : 0u;
        }
        const uint32_t total = (uint32_t)(__popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]));
        uint32_t base = 0;
        if (lane == 0 && total) base = atomicAdd(&s_tail, total);
        base = (uint32_t)__shfl((int)base, 0, 64);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if ((mine >> u) & 1u) {
                const uint32_t at = (base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[u], 0u))) & (SL_RING - 1);
                ring_lo[at] = lo4[u];
                ring_hi[at] = (uint8_t)(hi4[u] >> SL_BITS);
            }
            base += (uint32_t)__popcll(m[u]);
        }
        __syncthreads();
        const uint32_t tail = s_tail;
        while (tail - head >= (uint32_t)SL_CHUNK) {              // uniform
            follow(head, SL_CHUNK);
            head += SL_CHUNK;
        }
        __syncthreads();
    }
    const uint32_t tail = s_tail;
    if (tail != head) follow(head, tail - head);
    if (stats) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { probes += __shfl_xor(probes, d, 64); followed += __shfl_xor(followed, d, 64); }
        if (lane == 0 && probes) atomicAdd(stats, probes);
        if (lane == 0 && followed) atomicAdd(stats + 1, followed);
    }
}
constexpr int RUN_PERIOD = 250, RUN_LEN = 32;      // TILE = 8 periods; BT = 8 runs x 32 positions: one position per thread
static_assert(TILE % RUN_PERIOD == 0 && (TILE / RUN_PERIOD) * RUN_LEN == BT && 2 * RUN_PERIOD <= WINDOW, "runs tile the tile; every window holds two of them");
__global__ void __launch_bounds__(BT) ref_trio_runs(const TileDev* __restrict__ tiles, const ContigDev* __restrict__ contigs, const RefSource rs,
                                                    const uint32_t* __restrict__ counts, int k, int e, uint8_t* __restrict__ flags,
                                                    unsigned long long* __restrict__ stats /* nullable: [0] probes */, long n_blk) {
    const long blk = block2d();
    if (blk >= n_blk) return;
    const TileDev t = tiles[blk];
    const ContigDev c = contigs[t.contig];
    const long nk = (long)c.len - k + 1;
    const long j = (long)t.j0 + (threadIdx.x >> 5) * RUN_PERIOD + (threadIdx.x & 31);
    unsigned long long probes = 0;
    // ref_single_slots has run: a position whose listed hash -- its smallest -- does not read 3 carries the exact bit by now (0x80 /
    // 0x81: not trio), and so does one without a k-mer; every other position's smallest hash READS 3 and need not be asked again:
    // largest hash first (the likeliest "not 3"), then the middle one -- 1.9 probes per position of a run where all three took 2.9
    if (j < nk && !(flags[c.flat_base + j] & 0x80)) {
        const RefKmer km = ref_kmer(rs, c, j, k, e);
        uint32_t h[3];
#pragma unroll
        for (int i = 0; i < 3; i++) h[i] = i < e ? ref_hash(rs, km, i) : 0u;
        const ProbeOrder po = probe_order(h, e);
        bool all3 = h[0] != 0u && (e < 2 || h[1] != 0u) && (e < 3 || h[2] != 0u);   // a hash that is none never reads 3 (and the listed one is the smallest that is one)
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (q < e - 1 && all3) {
                const uint32_t hv = pick3(h, q == 0 ? po.hi : po.mid);
                probes++;
                all3 = count_of(counts, hv) == 3u;
            }
        if (all3) flags[c.flat_base + j] = 0x83;                  // single, trio, exact
    }
    if (stats) {   // one atomic per workgroup: 26 M waves on one address took 0.3 s of the (untimed) counting step
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) probes += __shfl_xor(probes, d, 64);
        __shared__ unsigned long long part[BT / 64];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = probes;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long v = 0;
            for (int q = 0; q < BT / 64; q++) v += part[q];
            if (v) atomicAdd(stats, v);
        }
    }
}

}  // namespace lhgt

using namespace lhgt;

static RefSource ref_source(const lhgt_ctx* ctx) {
    RefSource rs{};
    rs.index = ctx->ref_packed ? nullptr : ctx->d_index;
    rs.planes = ctx->d_ref_planes;
    rs.plane_words = ctx->ref_plane_words;
    rs.hp = ctx->hp;
    return rs;
}

// The slot list of the resident reference (see ref_flags_slots).  Built once per reference: a histogram pass over the positions'
// hash 0, the bucket offsets, a placing pass -- 2 x 13 G atomics on 2^18 counters for 13 Gbase, a second or two; kept until the
// reference is replaced.  Not built (the trio-first kernel scans) when a position needs more than 34 bits, when e > 3, or when
// it would leave less than LHGT_SLOT_LIST_HEADROOM_GB (default 40) of the device's memory free.
namespace lhgt {
void slot_list_drop(lhgt_ctx* ctx) {
    for (void* p : {(void*)ctx->d_sl_lo, (void*)ctx->d_sl_hi, (void*)ctx->d_sl_off, (void*)ctx->d_sl_mid, (void*)ctx->d_sl_end}) if (p) (void)lhgt::dev_free(p);
    ctx->d_sl_lo = nullptr; ctx->d_sl_hi = nullptr; ctx->d_sl_off = nullptr; ctx->d_sl_mid = nullptr; ctx->d_sl_end = nullptr;
    ctx->sl_capacity = 0;
    ctx->sl_entries = 0; ctx->sl_buckets = 0; ctx->sl_state = 0; ctx->sl_sparse_scans = 0; ctx->sl_need_share = 0.0;
}
}  // namespace lhgt
static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static int slot_list_build(lhgt_ctx* ctx, bool smallest) {
    const bool trace = getenv("LHGT_TRACE") != nullptr;
    ctx->sl_state = -1;                                          // whatever happens below: one attempt per reference
    if (ctx->e > 3 || ctx->n_tiles == 0 || ctx->n_pos >= (1ull << 34)) return LHGT_OK;
    const long nb = ctx->k > SL_BITS ? 1L << (ctx->k - SL_BITS) : 1;
    static const double headroom_gb = getenv("LHGT_SLOT_LIST_HEADROOM_GB") ? atof(getenv("LHGT_SLOT_LIST_HEADROOM_GB")) : 40.0;
    size_t free_b = 0, total_b = 0;
    LHGT_HIP(hipMemGetInfo(&free_b, &total_b));
    const double need = 6.0 * (double)ctx->n_pos + 12.0 * (double)nb;
    if ((double)free_b - need < headroom_gb * 1e9 && ctx->d_rg_buf) {              // the registry by partition's record buffers (idle between scans): the list is worth more
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        lhgt::dev_free(ctx->d_rg_buf);
        ctx->d_rg_buf = nullptr;
        ctx->rg_buf_bytes = 0;
    }
    if ((double)free_b - need < headroom_gb * 1e9 && lhgt::big_release_all())     // blocks parked by earlier contexts of the process count as free
        LHGT_HIP(hipMemGetInfo(&free_b, &total_b));
    if ((double)free_b - need < headroom_gb * 1e9) {   // (round 6: the absolute headroom alone -- "or a quarter of what is free" could leave a few GB for the next, larger sample)
        if (trace) fprintf(stderr, "[lhgt] slot list: %.1f GB wanted, %.1f GB free -- not built\n", need / 1e9, (double)free_b / 1e9);
        return LHGT_OK;
    }
    const double t0 = wall_s();
    // Round 6 (late): the regions from a SAMPLED histogram -- every 8th tile, a sixteenth of the exact pass's 0.5 s of atomics -- with six
    // standard deviations of slack (slot_list_cap), then ONE pass over the reference that places the entries and notices a bucket that
    // runs over its region (repeats the sample missed); only then is the build done over with the exact histogram, as until now.
    // LHGT_SLOT_LIST_SAMPLE=<n>: every n-th tile (default 8; 0 or 1: the exact histogram at once).
    static const int sample_env = getenv("LHGT_SLOT_LIST_SAMPLE") ? atoi(getenv("LHGT_SLOT_LIST_SAMPLE")) : 8;
    const int sample = sample_env >= 2 && sample_env <= 64 && ctx->n_tiles >= 64L * sample_env ? sample_env : 1;
    const dim3 grid = blocks2d(ctx->n_tiles), blk(BT);
    uint32_t unlisted = 0;
    unsigned long long n_entries = 0, n_cap = 0;
    bool estimated = false;
    // what the build costs is reported as the time of its KERNELS (events around the histogram and around the placing pass, every
    // attempt): the allocations in between take 0 or 3 seconds by the state the process' earlier frees left the driver in (an 84 GB
    // hipMalloc behind a dropped list of the other kind: tools/r06/slot_list_build_time.py) -- the wall time goes to the trace
    float kernels_ms = 0.f;
    auto add_elapsed = [&]() { float ms = 0.f; if (hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3) == hipSuccess) kernels_ms += ms; else (void)hipGetLastError(); };
    for (int attempt = sample > 1 ? 0 : 1; attempt < 2; attempt++) {
        estimated = attempt == 0;
        const int stride = estimated ? sample : 1;
        uint32_t* d_hist = nullptr;          // [nb] counts, then cursors; [nb] the unlisted k-mers; [nb + 1], [nb + 2]: the estimated fill's notes
        unsigned long long* d_total = nullptr;
        auto give_up = [&]() {
            if (d_hist) lhgt::dev_free(d_hist);
            if (d_total) lhgt::dev_free(d_total);
            slot_list_drop(ctx); ctx->sl_state = -1;
            (void)hipGetLastError();
        };
        if (lhgt::dev_malloc(&d_hist, (size_t)(nb + 4) * 4) != hipSuccess || lhgt::dev_malloc(&ctx->d_sl_off, (size_t)(nb + 1) * 8) != hipSuccess ||
            (estimated && (lhgt::dev_malloc(&ctx->d_sl_end, (size_t)nb * 8) != hipSuccess || lhgt::dev_malloc(&d_total, 8) != hipSuccess))) {
            give_up();
            return LHGT_OK;
        }
        LHGT_HIP(hipMemsetAsync(d_hist, 0, (size_t)(nb + 4) * 4, ctx->stream));
        LHGT_HIP(hipEventRecord(ctx->ev2, ctx->stream));
        hipLaunchKernelGGL(slot_list_hist, blocks2d((ctx->n_tiles + stride - 1) / stride), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->k, ctx->e,
                           smallest ? 1 : 0, d_hist, nb, ctx->n_tiles, stride);
        LHGT_HIP(hipMemcpyAsync(&unlisted, d_hist + nb, 4, hipMemcpyDeviceToHost, ctx->stream));
        hipLaunchKernelGGL(slot_list_offsets, dim3(1), dim3(1024), 0, ctx->stream, d_hist, nb, ctx->d_sl_off, stride);
        LHGT_HIP(hipMemcpyAsync(&n_cap, ctx->d_sl_off + nb, 8, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        add_elapsed();
        if (estimated) {                     // the sample's slack has to fit like the list itself
            size_t f1 = 0, t1 = 0;
            LHGT_HIP(hipMemGetInfo(&f1, &t1));
            if ((double)f1 - 6.0 * (double)n_cap < headroom_gb * 1e9) {
                lhgt::dev_free(d_hist); lhgt::dev_free(d_total);
                slot_list_drop(ctx); ctx->sl_state = -1;
                continue;                    // the exact histogram's regions are smaller
            }
        }
        if (lhgt::dev_malloc(&ctx->d_sl_lo, (size_t)(n_cap + 64) * 4) != hipSuccess || lhgt::dev_malloc(&ctx->d_sl_hi, (size_t)(n_cap + 64) * 2) != hipSuccess) {
            give_up();
            if (trace) fprintf(stderr, "[lhgt] slot list: no memory for %llu entries -- not built\n", n_cap);
            return LHGT_OK;
        }
        // the sparse-table list (largest hash) takes every entry's second-largest hash along when there is room for it (4 more bytes per
        // position: 130 GB in all for 13 Gbase; LHGT_SLOT_LIST_MID=0: never): the followed positions then ask the table before the reference
        static const bool mid_ok = !(getenv("LHGT_SLOT_LIST_MID") && atoi(getenv("LHGT_SLOT_LIST_MID")) == 0);
        if (!smallest && ctx->e >= 2 && mid_ok) {
            size_t f2 = 0, t2 = 0;
            LHGT_HIP(hipMemGetInfo(&f2, &t2));
            if ((double)f2 - 4.0 * (double)n_cap < headroom_gb * 1e9 && lhgt::big_release_all())   // blocks parked by the process count as free here too
                LHGT_HIP(hipMemGetInfo(&f2, &t2));
            if ((double)f2 - 4.0 * (double)n_cap >= headroom_gb * 1e9) {
                if (lhgt::dev_malloc(&ctx->d_sl_mid, (size_t)(n_cap + 64) * 4) != hipSuccess) { ctx->d_sl_mid = nullptr; (void)hipGetLastError(); }
            }
        }
        uint32_t* d_est = estimated ? d_hist + nb + 1 : nullptr;
        LHGT_HIP(hipEventRecord(ctx->ev2, ctx->stream));
        hipLaunchKernelGGL(slot_list_fill, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->k, ctx->e, smallest ? 1 : 0, ctx->d_sl_off, d_hist,
                           ctx->d_sl_lo, ctx->d_sl_hi, ctx->d_sl_mid, ctx->n_tiles, d_est);
        LHGT_HIP(hipGetLastError());
        n_entries = n_cap;
        if (estimated) {
            uint32_t note[2] = {0, 0};
            LHGT_HIP(hipMemsetAsync(d_total, 0, 8, ctx->stream));
            hipLaunchKernelGGL(slot_list_ends, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_sl_off, d_hist, nb, ctx->d_sl_end, d_total);
            LHGT_HIP(hipMemcpyAsync(note, d_est, 8, hipMemcpyDeviceToHost, ctx->stream));
            LHGT_HIP(hipMemcpyAsync(&n_entries, d_total, 8, hipMemcpyDeviceToHost, ctx->stream));
            LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            add_elapsed();
            unlisted = note[1];
            if (note[0]) {                   // a bucket ran over its estimated region: once more, exactly
                if (trace) fprintf(stderr, "[lhgt] slot list: a bucket ran over the region the sampled histogram gave it -- built again from the exact one\n");
                lhgt::dev_free(d_hist); lhgt::dev_free(d_total);
                slot_list_drop(ctx); ctx->sl_state = -1;
                continue;
            }
        }
        if (!estimated) {
            LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            add_elapsed();
        }
        lhgt::dev_free(d_hist);
        if (d_total) lhgt::dev_free(d_total);
        break;
    }
    ctx->sl_capacity = n_cap;
    ctx->sl_entries = n_entries;
    ctx->sl_buckets = nb;
    ctx->sl_state = 1;
    ctx->sl_smallest = smallest;
    ctx->sl_unlisted = unlisted;
    ctx->sl_build_ms = (double)kernels_ms;
    if (trace) fprintf(stderr, "[lhgt] slot list (by the %s hash%s; regions from %s): %llu positions in %ld buckets, %.1f GB, built in %.2f s (its kernels %.2f s)\n", smallest ? "smallest" : "largest", ctx->d_sl_mid ? ", with the second-largest" : "", estimated ? "a sampled histogram" : "the exact histogram", n_entries, nb, (ctx->d_sl_mid ? 10.0 : 6.0) * (double)n_cap / 1e9, wall_s() - t0, 1e-3 * (double)kernels_ms);
    return LHGT_OK;
}

// B1-B4 on the resident contigs: flags, and tile_count turned into exclusive local ids.
static int scan_local(lhgt_ctx* ctx, float hit_ratio, float match_ratio, uint32_t* total_new, unsigned long long* n_selected) {
    const int k = ctx->k, e = ctx->e;
    *total_new = 0;
    *n_selected = 0;
    if (ctx->n_tiles == 0) return LHGT_OK;
    int one_min = (int)(WINDOW * hit_ratio);      // float32 product truncated, E:559-560
    int three_min = (int)(WINDOW * match_ratio);
    const dim3 grid = blocks2d(ctx->n_tiles), blk(BT);
    const long nt = ctx->n_tiles;
    static const int lite_stride_env = getenv("LHGT_LITE_STRIDE") ? atoi(getenv("LHGT_LITE_STRIDE")) : 0;
    const int lite_stride = lite_stride_env >= 2 && lite_stride_env <= 64 ? lite_stride_env : 8;
    // summary of the count table (one streaming pass, ~0.3 ms per GiB): saturated 64-byte lines (their bitmap is consulted first
    // when at least half the lines are) and slots holding 3 (a nearly full table takes the lite form of B1/B2)
    bool use_sat = false;
    double frac3 = 0.0;
    const size_t n_lines = ctx->counts_words / 16;
    unsigned long long* d_nsat = (unsigned long long*)(ctx->d_tile_count + ((ctx->n_tiles + 2) & ~1L)) + 2;   // [2], then the need count
    if (n_lines >= 64 && !(ctx->debug & 64)) {
        if (!ctx->d_satline) LHGT_HIP(lhgt::dev_malloc(&ctx->d_satline, n_lines / 8 + 16));
        LHGT_HIP(hipMemsetAsync(d_nsat, 0, 16, ctx->stream));
        const size_t want = (n_lines + 255) / 256;
        hipLaunchKernelGGL(table_line_summary, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, ctx->stream, ctx->d_counts, n_lines,
                           ctx->d_satline, d_nsat);
        unsigned long long n_sat[2] = {0, 0};
        LHGT_HIP(hipMemcpyAsync(n_sat, d_nsat, 16, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        use_sat = 2 * n_sat[0] >= n_lines;
        frac3 = (double)n_sat[1] / ((double)((n_lines + 15) / 16) * 256.0);
    }
    // bit 12 forces the lite form, bit 13 the exact one
    ctx->scan_frac3 = frac3;
    ctx->scan_n_need = 0;
    // In between the clear cases a trial decides: the lite kernels on 64 runs of 65 consecutive tiles spread over the reference
    // (the first tile of a run only supplies the look-back of the second); lite if they settle at least 40 % of the others.
    // Three forms of B1/B2 (e <= 3): exact (every hash of every position); single-first ("lite": a nearly full table -- probe until a
    // hash reads 3); trio-first (a sparse table -- probe until a hash does NOT read 3).  Below 45 % of the slots at 3 the sparse
    // form asks 1 + f + f^2 <= 1.65 probes per position instead of 3 and random positions almost never make a candidate window;
    // above, the old rule: lite from 90 %, exact in between unless a trial of the lite kernels settles 40 % of its tiles.
    // bit 12 / 13 / 14 force single-first / exact / trio-first.
    const bool force_any = (ctx->debug & (4096 | 8192 | 16384 | (1 << 24))) != 0;
    const bool sparse_form = e <= 3 && ((ctx->debug & 16384) || ((ctx->debug & (1 << 24)) && !(ctx->debug & (4096 | 8192))) || (!force_any && frac3 < 0.45 && n_lines >= 64 && !(ctx->debug & 64)));
    double pilot_settled = -1.0;
    if (e <= 3 && !force_any && !sparse_form && frac3 >= 0.2 && frac3 < 0.9 && ctx->n_tiles >= 65 * 64 * 4) {
        std::vector<uint32_t> pl, pw;
        for (int r = 0; r < 64; r++) {
            const long t0 = (ctx->n_tiles - 65) * r / 63;
            for (int q = 0; q < 65; q++) {
                pl.push_back((uint32_t)(t0 + q));
                if (q) pw.push_back((uint32_t)(t0 + q));
            }
        }
        unsigned int* d_cnt = (unsigned int*)(d_nsat + 2);
        uint32_t* d_list = ctx->d_active_tiles;
        LHGT_HIP(hipMemsetAsync(d_cnt, 0, 4, ctx->stream));
        LHGT_HIP(hipMemcpyAsync(d_list, pl.data(), pl.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        LHGT_HIP(hipMemcpyAsync(d_list + pl.size(), pw.data(), pw.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(ref_flags_lite<false>, dim3((unsigned)pl.size()), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts,
                           k, e, ctx->d_flags, ctx->d_nzmask, ctx->d_satline, d_list, (long)pl.size(), lite_stride);
        hipLaunchKernelGGL(window_lite, dim3((unsigned)((pw.size() + 3) / 4)), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, one_min, three_min, ctx->d_flags,
                           ctx->d_tile_good, (uint32_t*)nullptr, d_cnt, d_list + pl.size(), (long)pw.size());
        unsigned int n_not = 0;
        LHGT_HIP(hipMemcpyAsync(&n_not, d_cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        pilot_settled = 1.0 - (double)n_not / (double)pw.size();
    }
    // measured (13 Gbase; slots at 3 / tiles the trial settles: phase B exact -> lite, ms): 81.8 % / 99.9 % (100 M pairs) 791 -> 368;
    // 74.2 % / 92.7 % (50 M) 878 -> 494; 67.4 % / 32.3 % (35 M) 946 -> 890; 58.6 % / 0.5 % (25 M) 1067 -> 1169; 24.5 % / 25.1 % (configs[1]) 97 -> 105
    ctx->scan_lite = sparse_form || (e <= 3 && !(ctx->debug & (8192 | 16384)) && ((ctx->debug & 4096) || (pilot_settled >= 0.0 ? pilot_settled >= 0.4 : frac3 >= 0.65)));
    ctx->scan_form = sparse_form ? 2 : ctx->scan_lite ? 1 : 0;
    ctx->stats_scan = ctx->stats_on;
    // slot-first instead of the trio-first kernel: the list is there, or this is the moment to build it -- a second sparse scan of the
    // same resident reference (LHGT_SLOT_LIST: 0 never, 1 that rule, 2 at the first sparse scan; lhgt_slot_list), debug bit 24 now;
    // bit 14 means the trio-first KERNEL, bit 25 leaves a list that exists unused (A/B)
    // Where it does not pay, left alone unless asked for (bit 24, mode 2): a small reference -- a bucket of 2^14 slots wants some 10^4
    // positions for a workgroup's set-up, i.e. a few Gbase (1 Gbase, index form: the kernel 52 ms where trio-first's takes 45) --; and a
    // sample that makes most tiles candidates (configs[1]: every genome sampled), whose fill then asks all e probes of every position
    // where trio-first's asks the missing ones: when the last scan of a list form sent more than half of the tiles to the fill, this
    // one takes the position-ordered kernel (and reports its own share for the next).
    // The single-first form has a list kernel too (slot-single: packed form, the list under the SMALLEST hash); a context keeps the
    // list of the form that built it -- one of 78 GB is what fits -- and only a scan that asks (bit 24) swaps it.
    ctx->scan_slots = false;
    const bool list_form = sparse_form || (ctx->scan_form == 1 && ctx->ref_packed && e <= 3);
    const bool want_smallest = !sparse_form;
    if (list_form && !(ctx->debug & (1 << 25)) && (!(ctx->debug & (16384 | 4096)) || (ctx->debug & (1 << 24)))) {
        const bool asked = (ctx->debug & (1 << 24)) || ctx->sl_mode == 2;
        const bool pays = ctx->n_pos >= (1ull << 32) && ctx->sl_need_share <= 0.5;
        if ((ctx->debug & (1 << 24)) && ctx->sl_state == 1 && ctx->sl_smallest != want_smallest) {
            const int scans = ctx->sl_sparse_scans;
            LHGT_HIP(hipStreamSynchronize(ctx->stream));
            slot_list_drop(ctx);
            ctx->sl_sparse_scans = scans;
        }
        if (ctx->sl_state == 0 && (asked || (ctx->sl_mode == 1 && ctx->sl_sparse_scans >= 1 && pays))) LHGT_TRY(slot_list_build(ctx, want_smallest));
        ctx->scan_slots = ctx->sl_state == 1 && ctx->sl_smallest == want_smallest && (asked || pays) && (sparse_form || ctx->sl_unlisted == 0);
    }
    struct InUse { lhgt_ctx* c; ~InUse() { c->sl_in_use = false; } } in_use{ctx};   // every return below ends with the stream drained
    ctx->sl_in_use = ctx->scan_slots;
    if (list_form) ctx->sl_sparse_scans++;
    if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] table: %.1f %% of the slots at 3, trial settles %.1f %% -> %s B1\n", 100.0 * frac3, 100.0 * pilot_settled, sparse_form ? "trio-first" : ctx->scan_lite ? "single-first (lite)" : "exact");
    LHGT_HIP(hipEventRecord(ctx->ev2, ctx->stream));
    if (sparse_form && ctx->scan_slots) {
        // slot-first: the list answers "does hash 0 read 3" for every position; flags / pstate start from "nothing known"
        LHGT_HIP(hipMemsetAsync(ctx->d_flags, 0, ctx->n_pos, ctx->stream));
        unsigned long long* st = ctx->stats_on && ctx->d_stats ? ctx->d_stats + 1 : nullptr;
        if (st) LHGT_HIP(hipMemsetAsync(st, 0, 16, ctx->stream));
        const int slice_words = (int)(ctx->counts_words < SL_SLOTS / 16 ? ctx->counts_words : SL_SLOTS / 16);
        const int ablate = (ctx->debug & (1 << 26)) && getenv("LHGT_SLOTS_ABLATE") ? atoi(getenv("LHGT_SLOTS_ABLATE")) : 0;
        auto launch = [&](auto kern) {
            hipLaunchKernelGGL(kern, blocks2d(ctx->sl_buckets), blk, 0, ctx->stream, ctx->d_sl_off, ctx->d_sl_end, ctx->d_sl_lo, ctx->d_sl_hi, ctx->d_sl_mid, ref_source(ctx),
                               ctx->d_contigs, (int)ctx->contigs.size(), ctx->d_counts, slice_words, k, e, ctx->d_flags, st, ctx->sl_buckets, ablate);
        };
        if (ctx->ref_packed) { if (ctx->d_sl_mid) launch(ref_flags_slots<true, BT, 4, true>); else launch(ref_flags_slots<true, BT, 4, false>); }
        else if (ctx->d_sl_mid) launch(ref_flags_slots<false, BT, 4, true>);
        else launch(ref_flags_slots<false, BT, 4, false>);
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
    } else if (sparse_form) {
        hipLaunchKernelGGL(ref_flags_trio, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags,
                           ctx->d_nzmask, nt);
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
        if (ctx->stats_on && ctx->d_stats) {
            LHGT_HIP(hipMemsetAsync(ctx->d_stats + 1, 0, 8, ctx->stream));
            hipLaunchKernelGGL(pstate_probe_sum, dim3(8192), dim3(256), 0, ctx->stream, ctx->d_nzmask, ctx->n_pos, ctx->d_stats + 1);
        }
    }
    if (sparse_form) {
        unsigned int* d_nneed = (unsigned int*)(d_nsat + 2);
        LHGT_HIP(hipMemsetAsync(d_nneed, 0, 4, ctx->stream));
        hipLaunchKernelGGL(window_trio, blocks2d((nt + 3) / 4), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, three_min, ctx->d_flags, ctx->d_tile_count, nt);   // tile_count: free until the id scan
        hipLaunchKernelGGL(mark_need_tiles, dim3((unsigned)((ctx->n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tiles, ctx->d_tile_count,
                           ctx->n_tiles, ctx->d_tile_good, ctx->d_active_tiles, d_nneed);
        unsigned int n_need = 0;
        LHGT_HIP(hipMemcpyAsync(&n_need, d_nneed, 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] tiles %ld, near a window that reaches the trio threshold %u\n", ctx->n_tiles, n_need);
        ctx->scan_n_need = n_need;
        ctx->sl_need_share = (double)n_need / (double)ctx->n_tiles;
        if (n_need) {   // the list sits in d_active_tiles, which mark_active_tiles overwrites only after these two have run
            if (ctx->scan_slots)   // (two launches: a tile's look-back is another tile's body, and the fill tells "written in this launch" by the byte)
                hipLaunchKernelGGL(clear_pstate_tiles, blocks2d(n_need), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, ctx->d_nzmask, (long)n_need);
            hipLaunchKernelGGL(ref_flags_fill, blocks2d(n_need), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, ref_source(ctx),
                               ctx->d_counts, k, e, ctx->d_flags, ctx->d_nzmask, (long)n_need, 1, ctx->d_tile_count, ctx->n_tiles);
            hipLaunchKernelGGL(window_good, blocks2d(((long)n_need + 3) / 4), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, 0x80, one_min, three_min,
                               ctx->d_flags, ctx->d_tile_good, (long)n_need);
        }
    } else if (ctx->scan_lite && ctx->scan_slots) {
        // slot-single: flags preset to `single`, taken back where there is no k-mer and where the list's followed positions find no 3;
        // the trio lower bound from runs of 32 positions
        LHGT_HIP(hipMemsetAsync(ctx->d_flags, 0x01, ctx->n_pos, ctx->stream));
        const uint64_t n_words = (ctx->n_pos + 31) / 32;
        hipLaunchKernelGGL(no_kmer_flags, blocks2d((long)((n_words + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ref_planes + 2 * ctx->ref_plane_words, n_words, k,
                           ctx->d_flags, ctx->n_pos);
        hipLaunchKernelGGL(contig_tail_flags, dim3((unsigned)((ctx->contigs.size() + 7) / 8)), dim3(256), 0, ctx->stream, ctx->d_contigs, (long)ctx->contigs.size(), k, ctx->d_flags);
        unsigned long long* st = ctx->stats_on && ctx->d_stats ? ctx->d_stats + 1 : nullptr;
        if (st) LHGT_HIP(hipMemsetAsync(st, 0, 16, ctx->stream));
        const int slice_words = (int)(ctx->counts_words < SL_SLOTS / 16 ? ctx->counts_words : SL_SLOTS / 16);
        hipLaunchKernelGGL(ref_single_slots<BT>, blocks2d(ctx->sl_buckets), blk, 0, ctx->stream, ctx->d_sl_off, ctx->d_sl_end, ctx->d_sl_lo, ctx->d_sl_hi, ref_source(ctx), ctx->d_counts,
                           slice_words, k, e, ctx->d_flags, st, ctx->sl_buckets);
        hipLaunchKernelGGL(ref_trio_runs, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags, st, nt);
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
    } else if (ctx->scan_lite) {
        if (use_sat)
            hipLaunchKernelGGL(ref_flags_lite<true>, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags,
                               ctx->d_nzmask, ctx->d_satline, (const uint32_t*)nullptr, nt, lite_stride);
        else
            hipLaunchKernelGGL(ref_flags_lite<false>, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags,
                               ctx->d_nzmask, ctx->d_satline, (const uint32_t*)nullptr, nt, lite_stride);
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
        if (ctx->stats_on && ctx->d_stats) {
            LHGT_HIP(hipMemsetAsync(ctx->d_stats + 1, 0, 8, ctx->stream));
            hipLaunchKernelGGL(pstate_probe_sum, dim3(8192), dim3(256), 0, ctx->stream, ctx->d_nzmask, ctx->n_pos, ctx->d_stats + 1);
        }
    }
    if (ctx->scan_lite && !sparse_form) {
        unsigned int* d_nneed = (unsigned int*)(d_nsat + 2);
        LHGT_HIP(hipMemsetAsync(d_nneed, 0, 4, ctx->stream));
        hipLaunchKernelGGL(window_lite, blocks2d((nt + 3) / 4), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, one_min, three_min, ctx->d_flags, ctx->d_tile_good,
                           ctx->d_active_tiles, d_nneed, (const uint32_t*)nullptr, nt);
        unsigned int n_need = 0;
        LHGT_HIP(hipMemcpyAsync(&n_need, d_nneed, 4, hipMemcpyDeviceToHost, ctx->stream));
        LHGT_HIP(hipStreamSynchronize(ctx->stream));
        if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] tiles %ld, not settled by the lower bound %u\n", ctx->n_tiles, n_need);
        ctx->scan_n_need = n_need;
        ctx->sl_need_share = (double)n_need / (double)ctx->n_tiles;
        if (n_need) {   // the list sits in d_active_tiles, which mark_active_tiles overwrites only after these two have run
            if (ctx->scan_slots)
                hipLaunchKernelGGL(clear_pstate_tiles, blocks2d(n_need), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, ctx->d_nzmask, (long)n_need);
            hipLaunchKernelGGL(ref_flags_fill, blocks2d(n_need), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, ref_source(ctx),
                               ctx->d_counts, k, e, ctx->d_flags, ctx->d_nzmask, (long)n_need, 0, (const uint32_t*)nullptr, ctx->n_tiles);   // (window_lite's list follows no rule the fill could ask again)
            hipLaunchKernelGGL(window_good, blocks2d(((long)n_need + 3) / 4), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ctx->d_active_tiles, 0x80, one_min, three_min,
                               ctx->d_flags, ctx->d_tile_good, (long)n_need);
        }
    } else if (!ctx->scan_lite) {
        if (use_sat)
            hipLaunchKernelGGL(ref_flags<true>, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags,
                               ctx->d_nzmask, ctx->d_satline, nt);
        else
            hipLaunchKernelGGL(ref_flags<false>, grid, blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx), ctx->d_counts, k, e, ctx->d_flags,
                               ctx->d_nzmask, ctx->d_satline, nt);
        LHGT_HIP(hipEventRecord(ctx->ev3, ctx->stream));
        hipLaunchKernelGGL(window_good, blocks2d((nt + 3) / 4), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, (const uint32_t*)nullptr, 0, one_min, three_min,
                           ctx->d_flags, ctx->d_tile_good, nt);
    }
    unsigned long long* d_nsel = (unsigned long long*)(ctx->d_tile_count + ((ctx->n_tiles + 2) & ~1L));
    LHGT_HIP(hipMemsetAsync(d_nsel, 0, 8, ctx->stream));
    unsigned int* d_nact = (unsigned int*)(d_nsel + 1);
    LHGT_HIP(hipMemsetAsync(d_nact, 0, 4, ctx->stream));
    LHGT_HIP(hipMemsetAsync(ctx->d_tile_count, 0, (size_t)(ctx->n_tiles + 1) * 4, ctx->stream));
    LHGT_HIP(hipMemsetAsync(ctx->d_tile_sel, 0, (size_t)(ctx->n_tiles + 1) * 4, ctx->stream));
    hipLaunchKernelGGL(mark_active_tiles, dim3((unsigned)((ctx->n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_tiles, ctx->d_contigs,
                       ctx->d_tile_good, ctx->n_tiles, ctx->d_active_tiles, d_nact, ctx->debug & 256);
    unsigned int n_active = 0;
    LHGT_HIP(hipMemcpyAsync(&n_active, d_nact, 4, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] tiles %ld, for interval_select %u\n", ctx->n_tiles, n_active);
    if (n_active)
        hipLaunchKernelGGL(interval_select, blocks2d(n_active), blk, 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, k, ctx->d_active_tiles,
                           ctx->d_flags, ctx->d_tile_count, ctx->d_tile_sel, d_nsel, (long)n_active);
    if (ctx->n_tiles <= 4L * SCAN_CHUNK && !((ctx->debug & 128) && ctx->n_tiles >= 3))
        hipLaunchKernelGGL(tile_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_tile_count, ctx->n_tiles);
    else {   // the active-tile list is free again: its first words hold the chunk sums (u64, far fewer than n_tiles / 2)
        const long n_chunks = (ctx->n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        unsigned long long* sums = (unsigned long long*)ctx->d_active_tiles;
        hipLaunchKernelGGL(tile_chunk_sums, dim3((unsigned)n_chunks), dim3(1024), 0, ctx->stream, ctx->d_tile_count, ctx->n_tiles, sums);
        hipLaunchKernelGGL(chunk_bases, dim3(1), dim3(1024), 0, ctx->stream, sums, n_chunks);
        hipLaunchKernelGGL(tile_chunk_scan, dim3((unsigned)n_chunks), dim3(1024), 0, ctx->stream, ctx->d_tile_count, ctx->n_tiles, sums, n_chunks);
    }
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipMemcpyAsync(total_new, ctx->d_tile_count + ctx->n_tiles, 4, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipMemcpyAsync(n_selected, d_nsel, 8, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    LHGT_HIP(hipEventElapsedTime(&ctx->phase_ms[3], ctx->ev2, ctx->ev3));
    return LHGT_OK;
}

// -t N emulation: the contig groups of split_ref (E:1280-1330) and their id ranges.  Thread j scans group j
// and numbers its peaks from j * (max_peak / N) (E:229-237).  Nothing that leaves the process depends on those bases: what counts
// is (1) the ORDER of the ids -- a later thread's id wins in peak_kmer (atomicMax; the contract runs the threads in creation order)
// and the interval file walks the ranges in thread order --, (2) which peaks share a range (one sentinel line per thread,
// E:520-543), (3) that a range holds at most max_peak / N peaks, and (4) that id 0 means "no peak" (E:454), which only thread 0's
// first peak can have.  So the ids stay the DENSE sequential ones of the tile scan -- tables, digests and the vote all-reduce are
// sized by the number of peaks, not by (N-1) * max_peak / N -- shifted by one when thread 0 finds nothing (first_id = 1: no peak is
// invisible then), and the per-thread ranges are kept as running ends for lhgt_write_intervals.
static std::vector<long> thread_groups(const lhgt_ctx* ctx, long* first_local) {   // first contig (global, 0-based) of every group
    const int k = ctx->k, e = ctx->e, N = ctx->emu_threads;
    std::vector<uint32_t> lens;
    if (!ctx->all_lens.empty()) lens = ctx->all_lens;
    else for (const ContigDev& c : ctx->contigs) lens.push_back(c.len);
    *first_local = ctx->all_lens.empty() || ctx->contigs.empty() ? 0 : (long)ctx->contigs[0].ref_index - 1;
    long words = 0;
    for (uint32_t len : lens) words += 1 + (long)(len - k + 1) * e;
    // split_ref over the whole index: a group closes with the contig at which the bytes before it exceed index_size / N + 1
    const long nc = (long)lens.size(), index_size = 1200 + 4 * words, each = index_size / N + 1;
    std::vector<long> group_first(1, 0);
    long pos = 1200, start_byte = 1200;
    for (long c = 0; c < nc; c++) {
        const long add = 4 * ((long)(lens[c] - k + 1) * e + 1);
        if (pos - start_byte > each) {
            start_byte = pos + add;
            if (c + 1 < nc) group_first.push_back(c + 1);
        }
        pos += add;
    }
    return group_first;
}

// new peaks of the resident contigs per group (the whole reference, or this rank's shard of it)
static int group_counts_local(lhgt_ctx* ctx, uint32_t total, std::vector<long>* counts) {
    const int N = ctx->emu_threads;
    long first_local = 0;
    const std::vector<long> gf = thread_groups(ctx, &first_local);
    if ((long)gf.size() > N) LHGT_FAIL(LHGT_E_STATE, "split_ref made %zu groups for %d threads", gf.size(), N);
    const long nl = (long)ctx->contigs.size();
    auto clampl = [&](long g) { const long x = g - first_local; return x < 0 ? 0 : (x > nl ? nl : x); };
    std::vector<uint32_t> at(gf.size() + 1, 0);   // peaks before the first resident contig of each group
    for (size_t g = 0; g <= gf.size(); g++) {
        const long lc = g < gf.size() ? clampl(gf[g]) : nl;
        if (lc >= nl) at[g] = total;
        else LHGT_HIP(hipMemcpyAsync(&at[g], ctx->d_tile_count + ctx->contig_first_tile[(size_t)lc], 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    counts->assign((size_t)N, 0);
    for (size_t g = 0; g < gf.size(); g++) (*counts)[g] = (long)at[g + 1] - (long)at[g];
    return LHGT_OK;
}

// the id ranges that follow from the per-group totals: first_id, running range ends
static int thread_id_ranges(lhgt_ctx* ctx, long max_peak, const std::vector<long>& totals, long* first_id) {
    const int N = ctx->emu_threads;
    const long each_peaks = max_peak / N;
    ctx->emu_each_peaks = each_peaks;
    long total = 0;
    for (int j = 0; j < N; j++) {
        if (totals[j] > each_peaks)
            LHGT_FAIL(LHGT_E_EMULATION, "Too many peaks! thread %d of %d found %ld, its id range holds %ld (the reference runs into the next thread's ids): appoint a larger max_peak_num (see --max_peak).", j, N, totals[j], each_peaks);
        total += totals[j];
    }
    *first_id = total > 0 && totals[0] == 0 ? 1 : 0;
    ctx->emu_range_end.assign((size_t)N, 0);
    long run = *first_id;
    for (int j = 0; j < N; j++) { run += totals[j]; ctx->emu_range_end[j] = run; }
    return LHGT_OK;
}

// peak tables sized for `total` peaks; peak_kmer cleared; prefilter decided from the global selected count
static int peaks_prepare(lhgt_ctx* ctx, uint32_t total, unsigned long long n_selected, long max_peak) {
    const size_t slots = (size_t)1 << ctx->k;
    ctx->n_peaks = -1;
    ctx->vote_groups_ok = false;          // (lhgt_ref_scan makes the dense vote's contig groups behind this; an installed registry has none)
    ctx->n_selected = n_selected;
    if ((long)total > max_peak)
        LHGT_FAIL(LHGT_E_TOO_MANY_PEAKS, "Too many peaks! %u > max_peak %ld: reduce the sampling size, or appoint a larger max_peak_num (see --max_peak).", total, max_peak);
    if (!ctx->d_peak_kmer) {
        const auto t0 = std::chrono::steady_clock::now();
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_peak_kmer, slots * 4));          // a closed context's table, if the process has kept one
        if (getenv("LHGT_TRACE"))
            fprintf(stderr, "[lhgt] peak_kmer: %.1f GiB taken or allocated in %.3f s\n", (double)(slots * 4) / (1ull << 30),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    LHGT_HIP(hipMemsetAsync(ctx->d_peak_kmer, 0, slots * 4, ctx->stream));  // E:1458
    // vote prefilter (k_vote.hip): a bitmap over the low pf_bits address bits (exact when pf_bits = k, folded otherwise).
    // Sized at >= 16 bits per registered k-mer (false positives <= 6 %) but no larger: a 4 MiB bitmap does not stay
    // resident in a 4 MiB L2 next to the read stream (21 % of its probes missed), a 256 KiB one does.
    static const int pf_cap = getenv("LHGT_PF_BITS") ? atoi(getenv("LHGT_PF_BITS")) : PF_BITS;   // A/B: a smaller bitmap (24 = 2 MiB: half an XCD's L2)
    const int pf_lim = pf_cap >= 19 && pf_cap < PF_BITS ? pf_cap : PF_BITS;
    const int pf_max = ctx->k < pf_lim ? ctx->k : pf_lim;
    const unsigned long long n_keys = n_selected * (unsigned long long)ctx->e;
    // Round 4 (profiles/r04/vote_l2_variants.txt): a 4 MiB bitmap is the WHOLE L2 of an XCD -- 3.3 % of its probes miss, and every
    // peak_kmer probe that follows costs its own line plus 2.2 more bitmap misses (75 fabric reads per pair where 16 are needed);
    // at 2 MiB the bitmap is resident (0.3 % misses, the level 296 instead of 331 ms per 100 M pairs) but lets 41 instead of 16
    // probes per pair through: 371 instead of 385 ms; THREE quarters of the 4 MiB (lhgt_hash.hpp: PF_Q3) are still resident (298 ms)
    // and let 20 through: 360 ms.  So where 16 bits per key ask for the last doubling and 6 bits per key fit 2^24, the bitmap is
    // 3 MiB (LHGT_PF_Q3=0: 2 MiB; LHGT_PF_BITS=25: the whole 4 MiB).
    int pf_bits = 19;
    while (pf_bits < pf_max && (1ull << pf_bits) < 16 * n_keys) pf_bits++;
    static const bool q3_off = getenv("LHGT_PF_Q3") && !atoi(getenv("LHGT_PF_Q3"));
    ctx->pf_q3 = false;
    if (pf_bits == PF_BITS && PF_BITS == 25 && ctx->k > PF_BITS && (1ull << 24) >= 6 * n_keys && !getenv("LHGT_PF_BITS")) {
        if (q3_off) pf_bits = 24;
        else ctx->pf_q3 = true;
    }
    if ((ctx->debug & (1 << 20)) && ctx->k > PF_BITS) { ctx->pf_q3 = true; pf_bits = PF_BITS; }   // test hook: the three-quarter bitmap whatever the key count
    if (pf_bits > pf_max) pf_bits = pf_max;
    ctx->pf_mask = (uint32_t)((1ull << pf_bits) - 1ull);
    ctx->pf2 = ctx->k - pf_bits >= 5 ? pf_bits : 0;   // five address bits above the fold: a second, independent bit per key
    static const bool mixed_off = getenv("LHGT_PF_MIXED") && !atoi(getenv("LHGT_PF_MIXED"));   // A/B: two bits from the key's own address bits, on the new words
    if (ctx->pf_q3 && !mixed_off) ctx->pf2 = PF_MIXED;   // the three-quarter form: three bits from the mixed key (lhgt_hash.hpp)
    // On while the bitmap still screens enough: with two bits per key, n_keys = 0.5 x 2^pf_bits leaves 63 % of the bits set and lets
    // 40 % of foreign probes through -- 285 of a pair's 714, which the queued kernel still holds in its queue -- and bitmap +
    // survivors (330 + 0.4 x 1430 ms per 100 M pairs) still beat 714 HBM probes per pair (1430 ms).  Round 1 stopped at 1/8 with a
    // queue for a dozen survivors: a ragged 13 Gbase reference (118 k contigs, 217 k peaks at their ends, 14 M registered k-mers)
    // fell off that cliff, 389 -> 1381 ms of phase C.
    ctx->prefilter_on = !(ctx->debug & 4) && 2 * n_keys <= (1ull << pf_bits);
    if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] peaks %u, registered positions %llu (%llu k-mers), prefilter 2^%d bits %s\n", total, n_selected, n_keys, pf_bits, ctx->prefilter_on ? "on" : "off");
    if (ctx->prefilter_on) {
        if (!ctx->d_prefilter) {
            LHGT_HIP(lhgt::dev_malloc(&ctx->d_prefilter, (size_t)(1u << PF_BITS) / 8));
            LHGT_HIP(lhgt::dev_malloc(&ctx->d_prefilter_fold, (size_t)128 * 1024));
        }
        LHGT_HIP(hipMemsetAsync(ctx->d_prefilter, 0, ((size_t)1 << pf_bits) / 8, ctx->stream));
    }
    if ((long)total + 1 > ctx->peaks_cap) {     // grow-only: no allocator traffic in steady state
        if (ctx->d_loci) { lhgt::dev_free(ctx->d_loci); ctx->d_loci = nullptr; }
        if (ctx->d_filter) { lhgt::dev_free(ctx->d_filter); ctx->d_filter = nullptr; }
        ctx->peaks_cap = (long)total + 1 + total / 8;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_loci, (size_t)ctx->peaks_cap * 8));
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_filter, (size_t)ctx->peaks_cap * 4));
    }
    LHGT_HIP(hipMemsetAsync(ctx->d_filter, 0, ((size_t)total + 1) * 4, ctx->stream));  // E:1457
    LHGT_HIP(hipMemsetAsync(ctx->d_loci, 0, ((size_t)total + 1) * 8, ctx->stream));
    return LHGT_OK;
}

// The contig groups of the dense vote's bound (k_vote.hip): VG_N runs of whole contigs with about equal shares of the peak ids.  Ids
// ascend with the tiles (register_peaks / rg_emit: tile_base + first_id + running count), so a contig's ids are the run
// [tile_base[its first tile], tile_base[the next contig's first tile]) + first_id; bounds[j] = the first id of the first contig that starts
// at or behind j / VG_N of the ids (bounds[0] = 0: ids below first_id belong to nobody), bounds[VG_N] = all ones.
__global__ void __launch_bounds__(1024) vote_group_bounds(const TileDev* __restrict__ tiles, const uint32_t* __restrict__ tile_base, long n_tiles,
                                                          uint32_t first_id, uint32_t* __restrict__ bounds) {
    const int j = threadIdx.x;
    if (j == 0) { bounds[0] = 0u; bounds[VG_N] = 0xffffffffu; }
    if (j == 0 || j >= VG_N) return;
    const uint32_t total = tile_base[n_tiles];
    const uint32_t want = (uint32_t)(((unsigned long long)total * (unsigned long long)j + VG_N - 1) / VG_N);
    long lo = 0, hi = n_tiles;                 // first tile whose base is >= want
    while (lo < hi) { const long mid = (lo + hi) >> 1; if (tile_base[mid] >= want) hi = mid; else lo = mid + 1; }
    long q = lo;
    if (q < n_tiles && q > 0 && tiles[q - 1].contig == tiles[q].contig) {   // inside a contig: on to the next contig's first tile
        const uint32_t c = tiles[q].contig;
        long a = q, b = n_tiles;               // first tile of a later contig (the tiles' contigs ascend)
        while (a < b) { const long mid = (a + b) >> 1; if (tiles[mid].contig > c) b = mid; else a = mid + 1; }
        q = a;
    }
    bounds[j] = q < n_tiles ? tile_base[q] + first_id : 0xffffffffu;
}

// The registry by partition (rg_emit / rg_split / rg_apply above) in front of the vote of a DENSE peak set: *done = false when the
// direct kernel is the better (or the only possible) way -- few records (a bitmap in front of the vote says so, too), no memory for a
// tenth of the records in flight.  LHGT_REGISTER_PART=0 never, =1 whenever k >= 20; debug bit 29 / bit 30 likewise;
// LHGT_REGISTER_CHUNKS / LHGT_REGISTER_TIGHT (tests): that many chunks; regions sized for a 1/TIGHT of the records, so that most find them full.
static int register_partitioned(lhgt_ctx* ctx, uint32_t first_id, unsigned long long n_sel, bool* done) {
    *done = false;
    ctx->rg_chunks = 0;
    ctx->rg_bound = 0;
    ctx->rg_direct = 0;
    static const int mode = getenv("LHGT_REGISTER_PART") ? atoi(getenv("LHGT_REGISTER_PART")) : -1;
    const bool forced = mode == 1 || (ctx->debug & (1 << 29));
    if (mode == 0 || (ctx->debug & (1 << 30)) || (ctx->prefilter_on && !forced) || ctx->k < 20 || ctx->k > 32 || ctx->n_tiles == 0 || n_sel == 0) return LHGT_OK;
    const unsigned long long e = (unsigned long long)ctx->e;
    if (!forced && n_sel * e < (1ull << 29)) return LHGT_OK;          // half a G of records: 20 ms of the direct kernel
    if (n_sel * e >= (1ull << 35)) return LHGT_OK;                    // (rg_base's arithmetic)
    const int want_chunks = getenv("LHGT_REGISTER_CHUNKS") ? atoi(getenv("LHGT_REGISTER_CHUNKS")) : 0;      // (read per scan: the tests vary them)
    const int tight = getenv("LHGT_REGISTER_TIGHT") ? atoi(getenv("LHGT_REGISTER_TIGHT")) : 0;
    const long n_groups = (ctx->n_tiles + RG_TILES - 1) / RG_TILES;
    const long n_final = (long)RG_B1 * RG_B2;
    const unsigned long long pad1 = tight > 1 ? 2 : RG_PAD1, pad2 = tight > 1 ? 2 : RG_PAD2;
    const long n_gblk = (n_groups + 1023) / 1024;
    const size_t pre_bytes = ((size_t)(n_groups + 1 + n_gblk + 1 + 16) * 8 + 255) & ~(size_t)255;                   // group prefix, the sums of 1024 groups, chunk bounds
    const size_t fixed = pre_bytes + (size_t)(RG_B1 + n_final + 8) * 8;                                               // + cursors, counter
    // bytes of the two record buffers for `nc` chunks
    auto plan = [&](int nc, unsigned long long* sel_per, unsigned long long* ucap, size_t* b1, size_t* b2) {
        *sel_per = (n_sel + (unsigned long long)nc - 1) / (unsigned long long)nc;
        unsigned long long u = (*sel_per + (unsigned long long)RG_TILES * TILE) * e;      // a group belongs to the chunk its first position falls into
        u += u / 16;
        if (tight > 1) u = u / (unsigned long long)tight + 1;
        *ucap = u;
        *b1 = (size_t)(rg_base(u, RG_B1, 8, pad1) + 16) * 8;
        *b2 = (size_t)(rg_base(u, (unsigned long long)n_final, RG_L2, pad2) + 16) * 8;
    };
    unsigned long long sel_per = 0, ucap = 0;
    size_t b1 = 0, b2 = 0;
    int nc = want_chunks > 0 ? std::min(want_chunks, 32) : 1;
    plan(nc, &sel_per, &ucap, &b1, &b2);
    if (want_chunks <= 0 && ctx->rg_buf_bytes < fixed + b1 + b2) {
        // what the device can spare: free and parked blocks less 8 GB, and no more than 34 GB (LHGT_REGISTER_GB) -- inside the 40 GB the
        // slot list leaves free (it goes first when an allocation fails: cabi.hip), which is worth more than fewer chunks (a chunk more = a sweep of the table more, ~10 ms); up to 8 chunks
        static const double cap_gb = getenv("LHGT_REGISTER_GB") ? atof(getenv("LHGT_REGISTER_GB")) : 34.0;
        size_t free_b = 0, total_b = 0;
        LHGT_HIP(hipMemGetInfo(&free_b, &total_b));
        const size_t avail = free_b + lhgt::dev_cached_bytes() + ctx->rg_buf_bytes;
        const size_t spare = avail > ((size_t)8 << 30) ? avail - ((size_t)8 << 30) : 0;
        // (all of a device without a slot list -- one chunk of 130 GB -- made the first scan 4.6 s slower: that much fresh memory is not free
        // to touch.)  Up to 48 GB where that still leaves 40 GB free: three chunks instead of four in the default regime, 5-10 ms
        const size_t lo_cap = (size_t)(cap_gb * 1e9), hi_cap = std::max(lo_cap, (size_t)48e9);
        const size_t beyond = spare > (size_t)40e9 ? spare - (size_t)40e9 : 0;
        const size_t room = std::min(spare, std::max(lo_cap, std::min(hi_cap, beyond)));
        while (nc < 8 && fixed + b1 + b2 > std::max(room, ctx->rg_buf_bytes)) { nc++; plan(nc, &sel_per, &ucap, &b1, &b2); }
        if (fixed + b1 + b2 > std::max(room, ctx->rg_buf_bytes)) {
            if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] registry by partition: %.1f GB wanted for 8 chunks, %.1f GB to spare -- the direct kernel\n", (double)(fixed + b1 + b2) / 1e9, (double)room / 1e9);
            return LHGT_OK;
        }
    }
    if (ctx->rg_buf_bytes < fixed + b1 + b2) {
        if (ctx->d_rg_buf) { lhgt::dev_free(ctx->d_rg_buf); ctx->d_rg_buf = nullptr; ctx->rg_buf_bytes = 0; }
        const double t_alloc = wall_s();
        const hipError_t ae = lhgt::dev_malloc(&ctx->d_rg_buf, fixed + b1 + b2);
        if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] registry by partition: %.1f GB taken or allocated in %.3f s\n", (double)(fixed + b1 + b2) / 1e9, wall_s() - t_alloc);
        if (ae != hipSuccess) {
            (void)hipGetLastError();
            if (getenv("LHGT_TRACE")) fprintf(stderr, "[lhgt] registry by partition: no %.1f GB -- the direct kernel\n", (double)(fixed + b1 + b2) / 1e9);
            return LHGT_OK;
        }
        ctx->rg_buf_bytes = fixed + b1 + b2;
    }
    unsigned long long* d_pre = (unsigned long long*)ctx->d_rg_buf;
    unsigned long long* d_sums = d_pre + n_groups + 1;
    long* d_bounds = (long*)(d_sums + n_gblk + 1);
    unsigned long long* d_cur1 = (unsigned long long*)(ctx->d_rg_buf + pre_bytes);
    unsigned long long* d_cur2 = d_cur1 + RG_B1;
    unsigned long long* d_direct = d_cur2 + n_final;
    unsigned long long* d_buf1 = (unsigned long long*)(ctx->d_rg_buf + fixed);
    unsigned long long* d_buf2 = (unsigned long long*)(ctx->d_rg_buf + fixed + b1);
    hipLaunchKernelGGL(rg_group_sums, dim3((unsigned)n_gblk), dim3(1024), 0, ctx->stream, ctx->d_tile_sel, ctx->n_tiles, n_groups, d_sums);
    hipLaunchKernelGGL(chunk_bases, dim3(1), dim3(1024), 0, ctx->stream, d_sums, n_gblk);
    hipLaunchKernelGGL(rg_group_prefix, dim3((unsigned)n_gblk), dim3(1024), 0, ctx->stream, ctx->d_tile_sel, ctx->n_tiles, n_groups, d_sums, d_pre);
    hipLaunchKernelGGL(rg_chunk_bounds, dim3(1), dim3(64), 0, ctx->stream, d_pre, n_groups, sel_per, nc, d_bounds);
    long bounds[65];
    LHGT_HIP(hipMemcpyAsync(bounds, d_bounds, (size_t)(nc + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    LHGT_HIP(hipMemsetAsync(d_direct, 0, 8, ctx->stream));
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    const size_t lds = (size_t)4 << (ctx->k - RG_L2);
    if (lds > 65536) LHGT_HIP(hipFuncSetAttribute((const void*)rg_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int n_cu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess) n_cu = prop.multiProcessorCount; else (void)hipGetLastError(); }
    const uint8_t* pstate = ctx->scan_form == 2 && !(ctx->debug & (1 << 23)) ? ctx->d_nzmask : nullptr;
    for (int c = 0; c < nc; c++) {
        LHGT_HIP(hipMemsetAsync(d_cur1, 0, (size_t)(RG_B1 + n_final) * 8, ctx->stream));
        if (bounds[c + 1] <= bounds[c]) continue;
        const int ablate = getenv("LHGT_RG_ABLATE") ? atoi(getenv("LHGT_RG_ABLATE")) : 0;
#define RG_EMIT(ST_, E_)                                                                                                                                               \
        hipLaunchKernelGGL((rg_emit<ST_, E_>), blocks2d(bounds[c + 1] - bounds[c]), dim3(RG_BT), 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, rsrc, ctx->d_counts,    \
                           ctx->d_flags, ctx->scan_lite ? nullptr : ctx->d_nzmask, pstate, ctx->d_tile_count, ctx->k, ctx->e, ctx->d_loci, ctx->d_peak_kmer, first_id,  \
                           ctx->n_tiles, bounds[c], bounds[c + 1], d_pre, sel_per, (unsigned long long)c, ucap, pad1, d_cur1, d_buf1, d_direct,                         \
                           ctx->prefilter_on ? ctx->d_prefilter : nullptr, ctx->pf_mask | (ctx->pf_q3 ? PF_Q3 : 0u), ctx->pf2, ablate)
        const RefSource rsrc = ref_source(ctx);
        if (rsrc.index) { RG_EMIT(false, 0); }
        else if (ctx->e == 3) { RG_EMIT(true, 3); }
        else if (ctx->e == 2) { RG_EMIT(true, 2); }
        else if (ctx->e == 1) { RG_EMIT(true, 1); }
        else { RG_EMIT(true, 0); }
#undef RG_EMIT
        if (ablate) continue;          // (timing runs of rg_emit alone: its records are not all there)
        hipLaunchKernelGGL(rg_split, dim3((unsigned)(2 * n_cu)), dim3(RG_SBT), 0, ctx->stream, d_buf1, d_cur1, ucap, pad1, pad2, ctx->k, d_cur2, d_buf2, ctx->d_peak_kmer, d_direct);
        hipLaunchKernelGGL(rg_apply, blocks2d(n_final), dim3(1024), lds, ctx->stream, d_buf2, d_cur2, ucap, pad2, ctx->k, ctx->d_peak_kmer, n_final);
    }
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipMemcpyAsync(&ctx->rg_direct, d_direct, 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->rg_chunks = nc;
    ctx->rg_bound = n_sel * e;
    *done = true;
    if (getenv("LHGT_TRACE"))
        fprintf(stderr, "[lhgt] registry by partition: <= %llu records in %d chunk(s), %.1f GB of record buffers\n", n_sel * e, nc, (double)(b1 + b2) / 1e9);
    return LHGT_OK;
}

extern "C" {

int lhgt_ref_scan(lhgt_ctx* ctx, float hit_ratio, float match_ratio, long max_peak, long* n_peaks) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (!ctx->index_resident) LHGT_FAIL(LHGT_E_STATE, "no index resident: call lhgt_index_load first");
    if (max_peak < 1) LHGT_FAIL(LHGT_E_ARG, "max_peak must be positive");
    const int k = ctx->k, e = ctx->e;
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    uint32_t total = 0;
    unsigned long long n_sel = 0;
    LHGT_TRY(scan_local(ctx, hit_ratio, match_ratio, &total, &n_sel));   // no contig longer than k: zero tiles, zero peaks
    const bool emu = ctx->emu_threads > 1;
    long first_id = 0;
    ctx->emu_range_end.clear();
    ctx->emu_ranges_pending = false;
    if (emu) {
        if (!ctx->all_lens.empty()) LHGT_FAIL(LHGT_E_STATE, "-t N emulation on a reference shard goes through lhgt_ref_scan_local / lhgt_ref_scan_group_counts / lhgt_set_group_totals");
        std::vector<long> counts;
        LHGT_TRY(group_counts_local(ctx, total, &counts));
        LHGT_TRY(thread_id_ranges(ctx, max_peak, counts, &first_id));
    }
    if ((long)total > max_peak)
        LHGT_FAIL(LHGT_E_TOO_MANY_PEAKS, "Too many peaks! %u > max_peak %ld: reduce the sampling size, or appoint a larger max_peak_num (see --max_peak).", total, max_peak);
    const long id_end = (long)total + first_id;
    LHGT_TRY(peaks_prepare(ctx, (uint32_t)id_end, n_sel, max_peak + first_id));
    ctx->vote_groups_ok = false;
    if (ctx->n_tiles > 0 && !ctx->prefilter_on) {       // a dense vote to come: the contig groups of its bound
        if (!ctx->d_vote_groups) LHGT_HIP(lhgt::dev_malloc(&ctx->d_vote_groups, (size_t)(VG_N + 1) * 4));
        hipLaunchKernelGGL(vote_group_bounds, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_tiles, ctx->d_tile_count, ctx->n_tiles, (uint32_t)first_id, ctx->d_vote_groups);
        ctx->vote_groups_ok = true;
    }
    bool by_partition = false;
    LHGT_TRY(register_partitioned(ctx, (uint32_t)first_id, n_sel, &by_partition));
    if (ctx->n_tiles > 0 && !by_partition)
        hipLaunchKernelGGL(register_peaks, blocks2d(ctx->n_tiles), dim3(BT), 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx),
                       ctx->d_counts, ctx->d_flags, ctx->scan_lite ? nullptr : ctx->d_nzmask,
                       ctx->scan_form == 2 && !(ctx->debug & (1 << 23)) ? ctx->d_nzmask : nullptr /* debug bit 23: look every count up (A/B) */,
                       ctx->d_tile_count, k, e, ctx->d_loci, ctx->d_peak_kmer,
                       ctx->prefilter_on ? ctx->d_prefilter : nullptr, ctx->pf_mask | (ctx->pf_q3 ? PF_Q3 : 0u), ctx->pf2, (uint32_t)first_id, ctx->n_tiles);
    LHGT_HIP(hipGetLastError());
    LHGT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    LHGT_HIP(hipEventSynchronize(ctx->ev1));
    LHGT_HIP(hipEventElapsedTime(&ctx->phase_ms[1], ctx->ev0, ctx->ev1));
    ctx->n_peaks = total;
    ctx->id_end = id_end;
    ctx->max_peak = max_peak;
    ctx->voted = false;
    if (n_peaks) *n_peaks = total;
    return LHGT_OK;
}

// which way the last lhgt_ref_scan registered its peaks: *chunks = 0 the direct kernel, n = by partition in n chunks
int lhgt_registry_info(lhgt_ctx* ctx, int* chunks, unsigned long long* records_bound, unsigned long long* records_direct) {
    if (!ctx) LHGT_FAIL(LHGT_E_ARG, "null context");
    if (chunks) *chunks = ctx->rg_chunks;
    if (records_bound) *records_bound = ctx->rg_bound;
    if (records_direct) *records_direct = ctx->rg_direct;
    return LHGT_OK;
}

// ---- reference-sharded phase B (BASELINE config #5: index larger than one GPU).
// Every rank holds a contiguous contig range (lhgt_index_load_shard) and the complete count table.
//   1. lhgt_ref_scan_local: B1-B4 on the local contigs -> #new peaks, #selected positions
//   2. ranks exchange the counts; id_base = new peaks of all lower ranks (contig order = rank order)
//   3. lhgt_ref_scan_emit: loci of the local new peaks and (hash, id) registrations as device records
//   4. ranks all-gather both record sets; lhgt_peaks_install replays them into the local peak_kmer
// after which lhgt_vote runs unchanged on this rank's read shard.
int lhgt_ref_scan_local(lhgt_ctx* ctx, float hit_ratio, float match_ratio, long* n_new_local, long* n_selected_local) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !n_new_local || !n_selected_local) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->index_resident) LHGT_FAIL(LHGT_E_STATE, "no index resident: call lhgt_index_load_shard first");
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    uint32_t total = 0;
    unsigned long long n_sel = 0;
    ctx->emu_range_end.clear();
    ctx->emu_ranges_pending = false;
    LHGT_TRY(scan_local(ctx, hit_ratio, match_ratio, &total, &n_sel));
    LHGT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    LHGT_HIP(hipEventSynchronize(ctx->ev1));
    LHGT_HIP(hipEventElapsedTime(&ctx->phase_ms[1], ctx->ev0, ctx->ev1));
    ctx->local_new = total;
    ctx->n_selected = n_sel;
    ctx->n_peaks = -1;
    *n_new_local = total;
    *n_selected_local = (long)n_sel;
    return LHGT_OK;
}

// -t N emulation on a reference shard: the split_ref groups cut across the ranks' shards.  Every rank reports the new peaks of
// its contigs per group; the host layer sums them over the ranks and hands the totals to every rank, which fixes the id ranges
// (thread_id_ranges above) before lhgt_ref_scan_emit / lhgt_peaks_install; *first_id is added to every rank's id base.
int lhgt_ref_scan_group_counts(lhgt_ctx* ctx, long* counts, int n) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !counts) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->emu_threads <= 1 || n != ctx->emu_threads) LHGT_FAIL(LHGT_E_ARG, "room for %d groups, the context emulates %d threads", n, ctx->emu_threads);
    if (ctx->local_new < 0) LHGT_FAIL(LHGT_E_STATE, "lhgt_ref_scan_local must precede lhgt_ref_scan_group_counts");
    std::vector<long> c;
    LHGT_TRY(group_counts_local(ctx, (uint32_t)ctx->local_new, &c));
    for (int j = 0; j < n; j++) counts[j] = c[(size_t)j];
    return LHGT_OK;
}

int lhgt_set_group_totals(lhgt_ctx* ctx, const long* totals, int n, long max_peak, long* first_id) {
    if (!ctx || !totals || !first_id || max_peak < 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if (ctx->emu_threads <= 1 || n != ctx->emu_threads) LHGT_FAIL(LHGT_E_ARG, "%d totals, the context emulates %d threads", n, ctx->emu_threads);
    std::vector<long> t(totals, totals + n);
    for (long x : t) if (x < 0) LHGT_FAIL(LHGT_E_ARG, "negative group total");
    LHGT_TRY(thread_id_ranges(ctx, max_peak, t, first_id));
    ctx->emu_ranges_pending = true;
    return LHGT_OK;
}

int lhgt_ref_scan_emit(lhgt_ctx* ctx, long id_base, void** d_loci, void** d_regs, long* n_regs) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !d_loci || !d_regs || !n_regs) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (ctx->local_new < 0) LHGT_FAIL(LHGT_E_STATE, "lhgt_ref_scan_local must precede lhgt_ref_scan_emit");
    if (id_base < 0 || id_base + ctx->local_new > 0xffffffffL) LHGT_FAIL(LHGT_E_ARG, "peak ids overflow 32 bits");
    const long need_loci = ctx->local_new + 1, need_regs = (long)ctx->n_selected * ctx->e + 1;
    if (need_loci > ctx->emit_loci_cap) {
        if (ctx->d_emit_loci) lhgt::dev_free(ctx->d_emit_loci);
        ctx->d_emit_loci = nullptr;
        ctx->emit_loci_cap = need_loci + need_loci / 8;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_emit_loci, (size_t)ctx->emit_loci_cap * 8));
    }
    if (need_regs > ctx->emit_regs_cap) {
        if (ctx->d_emit_regs) lhgt::dev_free(ctx->d_emit_regs);
        ctx->d_emit_regs = nullptr;
        ctx->emit_regs_cap = need_regs + need_regs / 8;
        LHGT_HIP(lhgt::dev_malloc(&ctx->d_emit_regs, (size_t)ctx->emit_regs_cap * 8 + 8));
    }
    unsigned long long* d_cnt = (unsigned long long*)(ctx->d_emit_regs + (size_t)ctx->emit_regs_cap * 2);
    LHGT_HIP(hipMemsetAsync(d_cnt, 0, 8, ctx->stream));
    unsigned long long cnt = 0;
    if (ctx->n_tiles > 0 && ctx->local_new > 0) {
        hipLaunchKernelGGL(emit_peaks, blocks2d(ctx->n_tiles), dim3(BT), 0, ctx->stream, ctx->d_tiles, ctx->d_contigs, ref_source(ctx),
                           ctx->d_counts, ctx->d_flags, ctx->scan_lite ? nullptr : ctx->d_nzmask, ctx->d_tile_count, ctx->k, ctx->e, (uint32_t)id_base, ctx->d_emit_loci,
                           ctx->d_emit_regs, d_cnt, ctx->n_tiles);
        LHGT_HIP(hipGetLastError());
        LHGT_HIP(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    LHGT_HIP(hipStreamSynchronize(ctx->stream));
    *d_loci = ctx->d_emit_loci;
    *d_regs = ctx->d_emit_regs;
    *n_regs = (long)cnt;
    return LHGT_OK;
}

int lhgt_peaks_install(lhgt_ctx* ctx, long n_peaks_total, long n_selected_total, long max_peak, const void* d_loci_all,
                       const void* d_regs_all, long n_regs_all) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || n_peaks_total < 0 || n_regs_all < 0 || max_peak < 1) LHGT_FAIL(LHGT_E_ARG, "bad argument");
    if ((n_peaks_total && !d_loci_all) || (n_regs_all && !d_regs_all)) LHGT_FAIL(LHGT_E_ARG, "null record buffer");
    if (n_peaks_total > 0xffffffffL) LHGT_FAIL(LHGT_E_TOO_MANY_PEAKS, "more than 2^32 peaks");
    // under -t N emulation the id ranges must have been fixed from the global per-group totals; a silent single range would be
    // neither the reference's -t 1 nor its -t N file
    if (ctx->emu_threads > 1 && !ctx->emu_ranges_pending)
        LHGT_FAIL(LHGT_E_STATE, "-t %d emulation: lhgt_set_group_totals must precede lhgt_peaks_install", ctx->emu_threads);
    if (ctx->emu_threads <= 1) ctx->emu_range_end.clear();
    ctx->emu_ranges_pending = false;
    LHGT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    LHGT_TRY(peaks_prepare(ctx, (uint32_t)n_peaks_total, (unsigned long long)n_selected_total, max_peak));
    if (n_peaks_total)
        LHGT_HIP(hipMemcpyAsync(ctx->d_loci, d_loci_all, (size_t)n_peaks_total * 8, hipMemcpyDeviceToDevice, ctx->stream));
    if (n_regs_all) {
        long blocks = (n_regs_all + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(replay_regs, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const uint32_t*)d_regs_all, n_regs_all,
                           ctx->d_peak_kmer, ctx->prefilter_on ? ctx->d_prefilter : nullptr, ctx->pf_mask | (ctx->pf_q3 ? PF_Q3 : 0u), ctx->pf2);
        LHGT_HIP(hipGetLastError());
    }
    LHGT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    LHGT_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    LHGT_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->phase_ms[1] += ms;
    ctx->n_peaks = n_peaks_total;
    ctx->id_end = n_peaks_total;
    ctx->max_peak = max_peak;
    ctx->voted = false;
    return LHGT_OK;
}

int lhgt_flags_export(lhgt_ctx* ctx, uint64_t first_pos, uint64_t n_pos, uint8_t* out) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (first_pos + n_pos > ctx->n_pos) LHGT_FAIL(LHGT_E_ARG, "position range outside the reference");
    LHGT_HIP(hipMemcpyAsync(out, ctx->d_flags + first_pos, n_pos, hipMemcpyDeviceToHost, ctx->stream)); LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

int lhgt_peak_kmer_export(lhgt_ctx* ctx, uint64_t first_slot, uint64_t n_slots, uint32_t* out) {
    LHGT_DEVICE_ENTRY(ctx);
    if (!ctx || !out) LHGT_FAIL(LHGT_E_ARG, "null argument");
    if (!ctx->d_peak_kmer) LHGT_FAIL(LHGT_E_STATE, "no scan done");
    if (first_slot + n_slots > (1ull << ctx->k)) LHGT_FAIL(LHGT_E_ARG, "slot range outside the table");
    LHGT_HIP(hipMemcpyAsync(out, ctx->d_peak_kmer + first_slot, n_slots * 4, hipMemcpyDeviceToHost, ctx->stream)); LHGT_HIP(hipStreamSynchronize(ctx->stream));
    return LHGT_OK;
}

}  // extern "C"
