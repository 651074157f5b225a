// k_count_part.hip -- phase A as a radix partition: the MI355X-first form of the k-mer count.
//
// Measured on MI355X (profiles/r01_probe_rates_microbench.txt, profiles/r02/): device atomics top out at 18-27 G/s whatever
// the locality, a random load costs a 128-byte line fill (~50 G lines/s), streaming runs at ~5 TB/s.  714 random
// read-modify-writes per read pair therefore cap the direct kernel (k_count.hip) at ~33 M pairs/s.  Here the hashes are first
// routed by their top bits so that every final bucket covers 2^16 table slots = one 16 KiB slice of the 2-bit table, which is
// then updated inside LDS:
//   P1 part_scatter_reads(_reg)  hash every k-mer, route 32-bit keys by the top B1 <= 8 bits      (tile sort in LDS, 128 KiB)
//   P2 part_scatter_keys16       route each level-1 segment by the next B2 <= 8 bits, write the
//                                low 16 bits -- all that still matters inside a bucket             (tile sort in LDS, 128 KiB)
//   P3 part_apply                one workgroup per final bucket: slice -> LDS, saturating
//                                2-bit increments by LDS compare-and-swap, slice -> HBM
// No histogram pass: every bucket owns a fixed region of the key buffer sized by its EXPECTED load plus 1/16 and 512 keys, and
// runs are placed with one global atomicAdd per tile and bucket.  A hash is min(forward word, reverse-complement word), the
// minimum of two roughly uniform values, so its density falls linearly (2(1-x)): bucket q of nb expects the share
// (2(nb-q)-1)/nb^2 of the keys -- twice the mean for the first, next to nothing for the last (part_region below).  A key that
// finds its bucket full -- heavily repeated k-mers: poly-A, adapters -- is applied to the table at once with the direct
// kernel's CAS loop; part_apply of the same chunk starts after the scatters and loads its slice from the table, so nothing is
// lost or counted twice and the result does not depend on how much overflowed.  (The first version counted the keys of every
// final bucket in a separate pass that hashed all reads once more: 69 ms of 534 on configs[2].)
// HBM traffic is 4 + 4 + 2 + 2 = 12 B per key streamed plus one table sweep per chunk, instead of one random 128-byte line
// (and its write-back) per key.  The result is the same table: min(3, count).
#include <algorithm>
#include "lhgt_hash.hpp"

namespace lhgt {

// if (T[h] < 3) T[h]++ on the packed table in HBM (E:1082-1084), race-free; k_count.hip holds the same loop for the direct kernel
__device__ __forceinline__ void part_sat_inc(uint32_t* __restrict__ T, uint32_t h) {
    uint32_t* w = T + (h >> 4);
    const uint32_t sh = (h & 15u) * 2u;
    uint32_t old = *w;
    while (((old >> sh) & 3u) != 3u) {
        const uint32_t seen = atomicCAS(w, old, old + (1u << sh));
        if (seen == old) break;
        old = seen;
    }
}

constexpr int SLICE_BITS = 16;            // slots per final bucket: its keys fit 16 bits, its slice of the 2-bit table is 16 KiB of LDS
constexpr int MAX_B1 = 8;                 // 256-way fan-out per scatter pass (B2 = k - 16 - B1 <= 8 for k <= 32)
constexpr int NBK = 1 << MAX_B1;          // buckets of one tile sort
constexpr int TILE_KEYS = 16384;          // keys sorted per workgroup tile of the generic read scatter (64 KiB of LDS)
constexpr int PT = 1024;                  // threads per workgroup in the read-side passes (16 waves hide the per-read load chain)
constexpr int TILE_KEYS2 = 32768;         // keys per tile of the key scatter (128 KiB of LDS: one workgroup per CU)
constexpr int PK = 1024;                  // threads per workgroup in the key scatter: 32 keys per thread stay in registers (one workgroup per CU)
constexpr int KPT = TILE_KEYS2 / PK;
constexpr int PA = 256;                   // threads per workgroup in apply

struct PartGeom {
    int k, slot_bits, b1, b2;             // b1 + b2 = k - slot_bits
    int nb1, nb2, nb;                     // buckets: level 1, per level-1 segment, final
};

__host__ inline PartGeom part_geom(int k) {
    PartGeom g;
    g.k = k;
    g.slot_bits = k < SLICE_BITS ? k : SLICE_BITS;
    int B = k - g.slot_bits;
    g.b1 = B < MAX_B1 ? B : MAX_B1;
    g.b2 = B - g.b1;
    g.nb1 = 1 << g.b1; g.nb2 = 1 << g.b2; g.nb = 1 << B;
    return g;
}

// Region layout of a key buffer for a chunk of at most n keys: final bucket q owns [part_region(q), part_region(q + 1)); a
// level-1 segment is the union of its nb2 final buckets, so both buffers use the same coordinates.
struct PartCap {
    unsigned long long n;
    uint32_t nb;
};
__host__ __device__ inline uint32_t part_region(const PartCap& c, uint32_t q) {
    const unsigned long long tri = (unsigned long long)q * (2ull * c.nb - q);                 // nb^2 (1 - (1 - q/nb)^2)
    const unsigned long long t1 = tri / c.nb, r1 = tri % c.nb;                                // n * tri / nb^2 without leaving 64 bits
    const unsigned long long share = (c.n * t1 + c.n * r1 / c.nb) / c.nb;
    return (uint32_t)((share + share / 16 + (unsigned long long)q * 512ull + 7ull) & ~7ull);   // multiples of 8 keys: apply reads 16-byte groups
}

// keys of read (m, p) at offsets lane, lane+64, ...: calls f(key) for each of the e hashes of valid k-mers.
// LDS-staged windows: the wave fetches the read's record (3 planes x wpr words, <= 51 words for 500 bases) with one coalesced
// load into its 64-word LDS area `stage`, and every lane cuts its windows out of LDS (instead of 6 global loads per offset).
template <class F>
__device__ __forceinline__ void for_each_key(const ReadBatchDev& b, const HashParams& hp, long p, int m, int lane, uint32_t* stage, F f) {
    if (b.flags && !((b.flags[p] >> m) & 1)) return;  // quirk Q4, thread-chunk emulation
    const int len = b.len[m][p];
    const int nk = len - hp.k + 1;
    if (nk <= 0) return;
    const int wpr = ((len + 31) >> 5) + 1;
    const uint32_t* rec = b.words + b.off[m][p];
    __builtin_amdgcn_wave_barrier();                 // the previous read's windows have been cut
    if (lane < 3 * wpr) stage[lane] = rec[lane];
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < nk; j += 64) {
        if (plane_window(stage + 2 * wpr, j, hp.k) != 0) continue;
        uint32_t whi = plane_window(stage, j, hp.k), wlo = plane_window(stage + wpr, j, hp.k);
        uint32_t rhi = brev_k(whi, hp.k), rlo = brev_k(wlo, hp.k);
        for (int i = 0; i < hp.e; i++) f(hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]));
    }
}

// exclusive scan of hist[0 .. nbk), nbk <= 256, by the first 256 threads of the workgroup (wave shuffles + 4 wave totals);
// EVERY thread of the workgroup calls it (it holds a barrier).  Returns this thread's exclusive prefix (threads >= nbk: unused).
__device__ __forceinline__ uint32_t bucket_excl_scan(const uint32_t* hist, int nbk, uint32_t* wsum /*[4]*/) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const uint32_t v = t < nbk ? hist[t] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (t < 256 && lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (int q = 0; q < wv && q < 4; q++) off += wsum[q];
    return off + incl - v;
}

// Sort the tile's keys (already counted into hist[nbk]) by bucket inside LDS and copy the runs out.
// Called by the whole workgroup; `place(emit)` must call emit(key) for every key of the tile again.
// Tile bucket q is the union of the final buckets first + q*step .. first + (q+1)*step - 1 and owns their regions of `out`;
// cursors[q] counts the keys sent to it so far (it may run past the region: the keys beyond go straight to the table).
// dl[bk] = (delta, limit): sorted position i of bucket bk goes to out[i + delta] while i < limit, beyond that its region is full.
// `prefetch()` runs between the placement and the copy-out: loads issued there (the next tile's input) fly during the stores.
template <int NT, class Place, class Prefetch>
__device__ __forceinline__ void tile_sort_flush(uint32_t* sorted, uint32_t* hist, uint32_t* lofs, uint32_t* lcur, uint2* dl,
                                                uint32_t* wsum, int nbk, int shift, uint32_t bmask,
                                                uint32_t* __restrict__ cursors, PartCap pc, uint32_t first, uint32_t step,
                                                uint32_t* __restrict__ out, uint32_t* __restrict__ counts, Place place, Prefetch prefetch) {
    // exclusive scan of hist (nbk <= 256) and reservation of the global runs: the atomicAdd's answer is first needed by the
    // copy-out, so its round trip to the L2 hides behind the placement
    const uint32_t o = bucket_excl_scan(hist, nbk, wsum);
    const bool mine = (int)threadIdx.x < nbk;
    uint32_t gb = 0, r0 = 0, cap = 0;
    if (mine) {
        lofs[threadIdx.x] = o;
        lcur[threadIdx.x] = o;
        const uint32_t c = hist[threadIdx.x];
        if (c) gb = atomicAdd(&cursors[threadIdx.x], c);
        r0 = part_region(pc, first + threadIdx.x * step);
        cap = part_region(pc, first + (threadIdx.x + 1) * step) - r0;
    }
    __syncthreads();
    place([&](uint32_t key) {
        uint32_t bk = (key >> shift) & bmask;
        sorted[atomicAdd(&lcur[bk], 1u)] = key;
    });
    if (mine) dl[threadIdx.x] = make_uint2(r0 + gb - o, o + (cap > gb ? cap - gb : 0u));
    prefetch();
    __syncthreads();
    const uint32_t total = lofs[nbk - 1] + hist[nbk - 1];
    for (uint32_t i = threadIdx.x; i < total; i += NT) {
        const uint32_t key = sorted[i];
        const uint2 d = dl[(key >> shift) & bmask];
        if (i < d.y) out[i + d.x] = key;                    // consecutive i of one bucket -> consecutive addresses
        else part_sat_inc(counts, key);                     // region full: count it now (see the header)
    }
    __syncthreads();
}

// ---- P1: reads -> level-1 segments.  A tile = reads_per_tile reads (<= TILE_KEYS keys).
__global__ void __launch_bounds__(PT) part_scatter_reads(ReadBatchDev b, long pair0, long npairs, HashParams hp, PartGeom g,
                                                         int reads_per_tile, PartCap pc, uint32_t* __restrict__ cur1,
                                                         uint32_t* __restrict__ out, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sorted[TILE_KEYS];
    __shared__ uint32_t hist[NBK], lofs[NBK], lcur[NBK], wsum[4];
    __shared__ uint2 dl[NBK];
    __shared__ uint32_t stage_all[(PT / 64) * 64];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    uint32_t* stage = stage_all + wib * 64;
    const int shift = g.k - g.b1;
    const uint32_t bmask = (uint32_t)g.nb1 - 1u;
    const long n_reads = 2 * npairs;
    const long n_tiles = (n_reads + reads_per_tile - 1) / reads_per_tile;
    for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const long r0 = t * reads_per_tile, r1 = r0 + reads_per_tile < n_reads ? r0 + reads_per_tile : n_reads;
        if (threadIdx.x < NBK) hist[threadIdx.x] = 0;
        __syncthreads();
        for (long r = r0 + wib; r < r1; r += PT / 64)
            for_each_key(b, hp, pair0 + (r >> 1), (int)(r & 1), lane, stage, [&](uint32_t key) { atomicAdd(&hist[g.b1 ? (key >> shift) & bmask : 0], 1u); });
        __syncthreads();
        tile_sort_flush<PT>(sorted, hist, lofs, lcur, dl, wsum, g.nb1, g.b1 ? shift : 0, g.b1 ? bmask : 0u, cur1, pc, 0u, (uint32_t)g.nb2, out, counts, [&](auto emit) {
            for (long r = r0 + wib; r < r1; r += PT / 64) for_each_key(b, hp, pair0 + (r >> 1), (int)(r & 1), lane, stage, emit);
        }, [] {});
    }
}

// ---- P1, common case (<= 128 k-mer offsets per read, e <= 3): every lane keeps the keys of its offsets in
// registers between the histogram and the placement, so reads are loaded and hashed once per tile.
// Measured split of one launch (3.3 M pairs, configs[1]): hashing 2.6 ms, histogram atomics 0.1, placement 1.9, the 9.5 GB of
// stores 2.2 (4.2 TB/s) -- they add up to the 6.9 ms of the launch.
// One 1024-thread workgroup per CU with a 128 KiB tile: 256-key (1 KiB) runs.  Phase A on configs[2]: 64 KiB tiles in two
// 512-thread workgroups per CU 460 ms, 48 KiB x 3 581 ms, 128 KiB x 1 450 ms (earlier, with the histogram pass: 64 KiB x 1 540,
// 64 KiB x 2 520, 32 KiB x 4 568): run length matters more than overlapping the phases of several workgroups.
constexpr int PT1 = 1024;
constexpr int TILE_KEYS1 = 32768;   // 128 KiB tile: one workgroup per CU, runs of 256 keys
constexpr int RW = 6;   // reads per wave per tile -> at most (PT1/64)*RW = 96 reads per tile (91 at 150 bp, e = 3)
// KC / EC: compile-time k and e of the common case (32, 3; 0 = run-time values): shifts by 32 - k vanish, the level-1 bucket is
// the key's top byte, the loops over the hashes lose their tests -- about a tenth of the kernel's vector instructions
template <int KC, int EC>
__global__ void __launch_bounds__(PT1) part_scatter_reads_reg(ReadBatchDev b, long pair0, long npairs, HashParams hp, PartGeom g,
                                                             int reads_per_tile, PartCap pc, uint32_t* __restrict__ cur1,
                                                             uint32_t* __restrict__ out, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sorted[TILE_KEYS1];
    __shared__ uint32_t hist2[2][NBK], lofs[NBK], lcur[NBK], wsum[4], s_total;
    __shared__ uint32_t dump[128];          // per-lane dummy counter and dummy word of the branch-free placement
    __shared__ uint2 dl[NBK];
    constexpr int STAGE_W = 20;             // <= 18 record words per read on this path (<= 159 bases) + the word a window may look past
    __shared__ uint32_t stage_all[(PT1 / 64) * RW * STAGE_W];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    uint32_t* stage = stage_all + wib * RW * STAGE_W;
    const bool r_zero = (lane & 31) == 0;   // the window starts at a word boundary: v_alignbit by 32 would give the NEXT word
    const uint32_t sh_win = 32u - (uint32_t)(lane & 31);
    const int shift = KC == 32 ? 24 : g.b1 ? g.k - g.b1 : 0;
    const uint32_t bmask = KC == 32 ? 0xffu : g.b1 ? (uint32_t)g.nb1 - 1u : 0u;
    const int k = KC ? KC : hp.k, e = EC ? EC : hp.e;
    const long n_reads = 2 * npairs;
    const long n_tiles = (n_reads + reads_per_tile - 1) / reads_per_tile;
    if (n_reads <= 0) return;
    // The wave's RW reads arrive in two round trips to memory: all descriptors (unconditional loads on clamped indices -- a load
    // under a lane- or wave-dependent branch is waited for on its own), then all records.  With one workgroup per CU nothing else
    // hides them, so they are software-pipelined across tiles, and so is the work itself (round 3): a tile's stretches -- hashing
    // (VALU), placement (LDS atomics), copy-out (LDS reads + stores) -- sit on different units but were serialised by the barriers
    // between them; now the NEXT tile is hashed (into registers and the other histogram) in the same barrier interval as this
    // tile's copy-out, so the stores drain while the waves hash.
    int lens[RW];
    uint32_t offs[RW], recw[RW];
    auto load_descriptors = [&](long t) {        // t >= n_tiles: every length 0
        const long r0 = t * reads_per_tile, r1 = r0 + reads_per_tile < n_reads ? r0 + reads_per_tile : n_reads;
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const long r = r0 + wib + rr * (PT1 / 64);
            const long rc = r < r1 ? r : n_reads - 1;
            const long p = pair0 + (rc >> 1);
            const int m = (int)(rc & 1);
            const int len = b.len[m][p];
            offs[rr] = b.off[m][p];
            const bool counted = !b.flags || ((b.flags[p] >> m) & 1);   // quirk Q4, thread-chunk emulation
            lens[rr] = r < r1 && counted ? len : 0;
        }
    };
    auto load_records = [&] {
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int wpr = ((lens[rr] + 31) >> 5) + 1;
            recw[rr] = b.words[offs[rr] + (lane < 3 * wpr ? lane : 0)];
        }
    };
    uint32_t key[RW][2][3];
    unsigned long long live = 0;   // bit (rr*6 + it*3 + i), RW*6 <= 64
    // hash the tile whose records sit in recw[] into key[] / live, counting its buckets in hist
    auto hash_tile = [&](uint32_t* hist) {
        live = 0;
#pragma unroll
        for (int rr = 0; rr < RW; rr++) stage[rr * STAGE_W + (lane < STAGE_W ? lane : 0)] = recw[rr];   // lanes >= 3 wpr hold a repeat of word 0
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            const int len = lens[rr];
            const int nk = len - k + 1;
            if (nk <= 0) continue;
            const int wpr = ((len + 31) >> 5) + 1;
            // The read's record sits in the wave's staging words, word w of plane p at p * wpr + w.  Offset j = 64 it + lane needs the
            // two consecutive words 2 it + (lane >> 5) and the next of each plane: ONE ds_read2_b32 per plane (a broadcast: the wave
            // reads three distinct words), one v_alignbit, one select for the lanes whose window starts at a word boundary.
            // (Round 2 cut the windows from the record held in registers with readlane -- three readlanes, four moves of their scalar
            // results, two selects and an exec region per plane: 15 vector instructions per plane, more than the hashing proper.)
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int j = it * 64 + lane;
                auto window = [&](int plane) {
                    const uint32_t* w = stage + rr * STAGE_W + plane * wpr + 2 * it + (lane >> 5);
                    const uint32_t w0 = w[0], w1 = w[1];
                    const uint32_t a = __builtin_amdgcn_alignbit(w0, w1, sh_win);
                    return (r_zero ? w0 : a) >> (32 - k);
                };
                if (j >= nk || window(2) != 0) continue;
                const uint32_t whi = window(0), wlo = window(1);
                const uint32_t rhi = brev_k(whi, k), rlo = brev_k(wlo, k);
#pragma unroll
                for (int i = 0; i < 3; i++)
                    if (i < e) {
                        const uint32_t h = hash_from_windows(whi, wlo, rhi, rlo, hp.mask[i]);
                        key[rr][it][i] = h;
                        live |= 1ull << (rr * 6 + it * 3 + i);
                        atomicAdd(&hist[KC == 32 ? h >> 24 : (h >> shift) & bmask], 1u);
                    }
            }
        }
    };
    load_descriptors(blockIdx.x);
    load_records();
    if (threadIdx.x < NBK) hist2[0][threadIdx.x] = 0;
    __syncthreads();
    hash_tile(hist2[0]);
    load_descriptors((long)blockIdx.x + gridDim.x);
    int cur = 0;
    const int nbk = KC == 32 ? 256 : g.nb1;
    for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        uint32_t* hist = hist2[cur];
        __syncthreads();                    // this tile's histogram is complete; the previous copy-out has read sorted / dl
        // exclusive scan of hist (nbk <= 256) and reservation of the global runs: the atomicAdd's answer is first needed by the
        // copy-out, so its round trip to the L2 hides behind the placement
        const uint32_t o = bucket_excl_scan(hist, nbk, wsum);
        const bool mine = (int)threadIdx.x < nbk;
        uint32_t gb = 0, r0 = 0, cap = 0;
        if (mine) {
            lofs[threadIdx.x] = o;
            lcur[threadIdx.x] = o;
            const uint32_t c = hist[threadIdx.x];
            if (c) gb = atomicAdd(&cur1[threadIdx.x], c);
            r0 = part_region(pc, threadIdx.x * (uint32_t)g.nb2);
            cap = part_region(pc, (threadIdx.x + 1) * (uint32_t)g.nb2) - r0;
            if ((int)threadIdx.x == nbk - 1) s_total = o + c;
        }
        if (threadIdx.x < NBK) hist2[cur ^ 1][threadIdx.x] = 0;     // the next tile's histogram
        __syncthreads();
        // Placement, six keys (one read) at a time with NO branch per key: a dead slot (offset beyond the read, k-mer with an N, padding
        // read) takes its ticket from a per-lane dummy counter and writes into a per-lane dummy word, so the six LDS atomics are
        // issued back to back and waited for once.  (As `if (live) sorted[atomicAdd(..)] = key` every key was its own exec region with
        // its own s_waitcnt: 36 exposed LDS round trips per tile and wave.)
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            uint32_t pos[6];
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const bool lv = (live >> (rr * 6 + u)) & 1ull;
                const uint32_t kk = key[rr][u / 3][u % 3];
                uint32_t* ctr = lv ? &lcur[KC == 32 ? kk >> 24 : (kk >> shift) & bmask] : &dump[lane];
                pos[u] = atomicAdd(ctr, 1u);
            }
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const bool lv = (live >> (rr * 6 + u)) & 1ull;
                uint32_t* dst = lv ? &sorted[pos[u]] : &dump[64 + lane];
                *dst = key[rr][u / 3][u % 3];
            }
        }
        if (mine) dl[threadIdx.x] = make_uint2(r0 + gb - o, o + (cap > gb ? cap - gb : 0u));
        load_records();                     // the next tile's records (its descriptors came in during the last hashing)
        __syncthreads();
        const uint32_t total = s_total;
        // copy-out, four keys per thread and round (the LDS reads of a round are issued together), then the tail one by one
        uint32_t i0 = threadIdx.x;
        for (; i0 + 3 * PT1 < total; i0 += 4 * PT1) {
            uint32_t kk[4];
            uint2 d[4];
            bool over = false;
#pragma unroll
            for (int u = 0; u < 4; u++) kk[u] = sorted[i0 + u * PT1];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = dl[KC == 32 ? kk[u] >> 24 : (kk[u] >> shift) & bmask];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * PT1;
                if (i < d[u].y) out[i + d[u].x] = kk[u];       // consecutive i of one bucket -> consecutive addresses
                else over = true;
            }
            if (over) {                                         // region full (heavily repeated k-mers): count those keys now (see the header)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * PT1 >= d[u].y) part_sat_inc(counts, kk[u]);
            }
        }
        for (; i0 < total; i0 += PT1) {
            const uint32_t kk = sorted[i0];
            const uint2 d = dl[KC == 32 ? kk >> 24 : (kk >> shift) & bmask];
            if (i0 < d.y) out[i0 + d.x] = kk;
            else part_sat_inc(counts, kk);
        }
        hash_tile(hist2[cur ^ 1]);          // the next tile, while this one's stores drain
        load_descriptors(t + 2L * gridDim.x);
        cur ^= 1;
    }
}

// ---- P2: level-1 segment -> its nb2 final buckets, as 16-BIT keys.  Inside a final bucket only the low slot_bits = 16 bits of
// a key still carry information (the rest is the bucket's number), so the tile is sorted and written as uint16: this pass
// writes, and apply reads, 2 bytes per key instead of 4 -- 12 instead of 16 bytes of HBM traffic per key over the three passes.
// Tiles of TILE_KEYS2 keys, never straddling segments; a thread keeps its KPT keys in registers between the histogram and the
// placement, all loads in flight at once.  Runs are copied out bucket by bucket (a wave per bucket: the bucket number, which the
// 16-bit key no longer holds, is the loop variable); a key that finds its region full is rebuilt from (bucket, low bits) and
// counted at once.  (A second level exists only for k >= 25, where slot_bits is 16.)
template <int KC>     // KC = 32: b1 = b2 = 8, sixteen slot bits -- the final bucket inside a segment is bits 16..23 of the key
__global__ void __launch_bounds__(PK) part_scatter_keys16(const uint32_t* __restrict__ in, const uint32_t* __restrict__ cur1, PartGeom g,
                                                          PartCap pc, uint32_t* __restrict__ cur2, uint16_t* __restrict__ out,
                                                          uint32_t* __restrict__ counts) {
    __shared__ uint32_t sorted[TILE_KEYS2];   // (bucket << 16) | low 16 bits: the copy-out below needs no search for the bucket
    __shared__ uint32_t hist[NBK], lofs[NBK], lcur[NBK], wsum[4];
    __shared__ uint2 dl[NBK];                 // (delta, limit) per bucket: sorted position i -> out[i + delta] while i < limit
    __shared__ uint32_t dump[128];            // per-lane dummy counter and dummy word of the branch-free placement
    __shared__ uint32_t tile_pref[NBK + 1];   // tiles before segment s
    __shared__ uint32_t seg_at[NBK], seg_len[NBK];
    if ((int)threadIdx.x < g.nb1) {
        const uint32_t s = threadIdx.x, r0 = part_region(pc, s << g.b2), cap = part_region(pc, (s + 1) << g.b2) - r0;
        seg_at[s] = r0;
        seg_len[s] = cur1[s] < cap ? cur1[s] : cap;
        hist[s] = (seg_len[s] + TILE_KEYS2 - 1) / TILE_KEYS2;      // tiles of segment s
    } else if (threadIdx.x < NBK) hist[threadIdx.x] = 0;
    __syncthreads();
    {
        const uint32_t o = bucket_excl_scan(hist, g.nb1, wsum);
        if ((int)threadIdx.x < g.nb1) {
            tile_pref[threadIdx.x] = o;
            if ((int)threadIdx.x == g.nb1 - 1) tile_pref[g.nb1] = o + hist[threadIdx.x];
        }
    }
    __syncthreads();
    const uint32_t n_tiles = tile_pref[g.nb1];
    const int shift = KC == 32 ? 16 : g.slot_bits;
    const uint32_t bmask = KC == 32 ? 0xffu : (uint32_t)g.nb2 - 1u;
    // tile t: its segment s (last s with tile_pref[s] <= t) and key range [k0, k1); t >= n_tiles: empty
    auto tile_bounds = [&](uint32_t t, int& s, uint32_t& k0, uint32_t& k1) {
        if (t >= n_tiles) { s = 0; k0 = k1 = 0; return; }
        int lo = 0, hi = g.nb1;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (tile_pref[mid] <= t) lo = mid; else hi = mid; }
        s = lo;
        const uint32_t seg0 = seg_at[s], seg1 = seg0 + seg_len[s];
        k0 = seg0 + (t - tile_pref[s]) * TILE_KEYS2;
        k1 = k0 + TILE_KEYS2 < seg1 ? k0 + TILE_KEYS2 : seg1;
    };
    // a thread's KPT keys of a tile, all loads in flight at once.  The next tile's are issued before the copy-out of this one
    // (the registers are free by then): with one workgroup per CU nothing else would hide their latency.
    uint32_t key[KPT];
    auto load_keys = [&](uint32_t k0, uint32_t k1) {
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            const uint32_t i = k0 + u * PK + threadIdx.x;
            key[u] = i < k1 ? in[i] : 0u;
        }
    };
    int s; uint32_t k0, k1;
    tile_bounds(blockIdx.x, s, k0, k1);
    load_keys(k0, k1);
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        __syncthreads();                    // the previous tile's copy-out has read hist / lofs / sorted
        if (threadIdx.x < NBK) hist[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (k0 + u * PK + threadIdx.x < k1) atomicAdd(&hist[(key[u] >> shift) & bmask], 1u);
        __syncthreads();
        const uint32_t first = (uint32_t)s << g.b2;
        const uint32_t o = bucket_excl_scan(hist, g.nb2, wsum);
        const bool mine = (int)threadIdx.x < g.nb2;
        uint32_t gb = 0, r0 = 0, cap = 0;
        if (mine) {
            lofs[threadIdx.x] = o;
            lcur[threadIdx.x] = o;
            const uint32_t c = hist[threadIdx.x];
            if (c) gb = atomicAdd(&cur2[first + threadIdx.x], c);   // answer needed by the copy-out only: in flight during the placement
            r0 = part_region(pc, first + threadIdx.x);
            cap = part_region(pc, first + threadIdx.x + 1) - r0;
        }
        __syncthreads();
        // placement, eight keys at a time without a branch per key (see part_scatter_reads_reg): the LDS atomics of a group are
        // issued back to back; a slot beyond the tile's end takes its ticket from a per-lane dummy counter
#pragma unroll
        for (int u0 = 0; u0 < KPT; u0 += 8) {
            uint32_t pos[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool lv = k0 + (u0 + u) * PK + threadIdx.x < k1;
                uint32_t* ctr = lv ? &lcur[(key[u0 + u] >> shift) & bmask] : &dump[threadIdx.x & 63];
                pos[u] = atomicAdd(ctr, 1u);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const bool lv = k0 + (u0 + u) * PK + threadIdx.x < k1;
                const uint32_t bk = (key[u0 + u] >> shift) & bmask;
                uint32_t* dst = lv ? &sorted[pos[u]] : &dump[64 + (threadIdx.x & 63)];
                *dst = (bk << 16) | (key[u0 + u] & 0xffffu);
            }
        }
        if (mine) dl[threadIdx.x] = make_uint2(r0 + gb - o, o + (cap > gb ? cap - gb : 0u));
        tile_bounds(t + gridDim.x, s, k0, k1);
        load_keys(k0, k1);
        __syncthreads();
        const uint32_t total = lofs[g.nb2 - 1] + hist[g.nb2 - 1];
        uint32_t i0 = threadIdx.x;
        for (; i0 + 3 * PK < total; i0 += 4 * PK) {   // four keys per thread and round: their LDS reads go out together
            uint32_t v[4];
            uint2 d[4];
            bool over = false;
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = sorted[i0 + u * PK];
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = dl[v[u] >> 16];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = i0 + u * PK;
                if (i < d[u].y) out[i + d[u].x] = (uint16_t)v[u];           // consecutive i of one bucket -> consecutive addresses
                else over = true;
            }
            if (over) {                                                       // region full: count those keys now (see the header)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (i0 + u * PK >= d[u].y) part_sat_inc(counts, ((first + (v[u] >> 16)) << shift) | (v[u] & 0xffffu));
            }
        }
        for (; i0 < total; i0 += PK) {
            const uint32_t v = sorted[i0], bk = v >> 16;
            const uint2 d = dl[bk];
            if (i0 < d.y) out[i0 + d.x] = (uint16_t)v;
            else part_sat_inc(counts, ((first + bk) << shift) | (v & 0xffffu));
        }
    }
}

// The eight 16-bit keys of a 16-byte group against a slice of the table in LDS: if (T[h] < 3) T[h]++ (E:1082-1084), race-free.  The eight
// words are read first (eight ds_read_b32 in flight), then only the keys whose slot does not read 3 enter the compare-and-swap loop.
// A word read early may be stale by then -- slots only ever grow, and stay at 3: a stale 3 is a 3, and a stale smaller value makes
// the compare-and-swap fail and hand back the word as it is.  (Until round 6 every key read its word through a `volatile` generic
// pointer -- a flat_load with s_waitcnt vmcnt(0) each, one after the other, which also waited for the next keys' global loads:
// part_apply2 35.7 -> 34.1 ms per 100 M pairs; the kernel moves its 169 GB at 5 TB/s either way.)
__device__ __forceinline__ void apply_group8(uint32_t* slice, const uint4& v, uint32_t n_valid /* keys of the group that count (>= 8: all) */) {
    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
    uint32_t key[8], o[8];
#pragma unroll
    for (int q = 0; q < 8; q++) key[q] = (q & 1) ? w4[q >> 1] >> 16 : w4[q >> 1] & 0xffffu;
#pragma unroll
    for (int q = 0; q < 8; q++) o[q] = slice[key[q] >> 4];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const uint32_t sh = (key[q] & 15u) * 2u;
        uint32_t cur = o[q];
        if ((uint32_t)q < n_valid && ((cur >> sh) & 3u) != 3u) {
            uint32_t* w = slice + (key[q] >> 4);
            do {
                const uint32_t seen = atomicCAS(w, cur, cur + (1u << sh));
                if (seen == cur) break;
                cur = seen;
            } while (((cur >> sh) & 3u) != 3u);
        }
    }
}

// ---- P3: apply one final bucket inside LDS.  K16: the bucket's keys are the 16-bit ones of part_scatter_keys16, read eight at
// a time (regions start at multiples of 8 keys); otherwise (k <= 24: no second level) they are the level-1 keys, masked.
template <bool K16>
__global__ void __launch_bounds__(PA) part_apply(const void* __restrict__ keys_v, const uint32_t* __restrict__ n_keys /*[nb]*/, PartGeom g,
                                                 PartCap pc, uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t slice[];   // 2^slot_bits / 16 words
    const uint32_t fb = blockIdx.x;
    const uint32_t k0 = part_region(pc, fb), cap = part_region(pc, fb + 1) - k0;
    const uint32_t k1 = k0 + (n_keys[fb] < cap ? n_keys[fb] : cap);
    if (k0 == k1) return;                 // untouched slice: nothing to read or write
    const int words = ((1 << g.slot_bits) + 15) >> 4;
    uint32_t* T = counts + (size_t)fb * words;
    for (int i = threadIdx.x; i < words; i += PA) slice[i] = T[i];
    __syncthreads();
    const uint32_t smask = (1u << g.slot_bits) - 1u;
    auto sat_inc_lds = [&](uint32_t s) {          // if (T[h] < 3) T[h]++  (E:1082-1084), race-free
        uint32_t* w = slice + (s >> 4);
        const uint32_t sh = (s & 15u) * 2u;
        uint32_t o = *w;
        while (((o >> sh) & 3u) != 3u) {
            const uint32_t seen = atomicCAS(w, o, o + (1u << sh));
            if (seen == o) break;
            o = seen;
        }
    };
    if (K16) {
        const uint16_t* keys = (const uint16_t*)keys_v;
        constexpr int U = 2;              // 16-byte groups in flight per thread
        for (uint32_t base = k0; base < k1; base += U * PA * 8) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = base + (u * PA + threadIdx.x) * 8;
                v[u] = i < k1 ? *(const uint4*)(keys + i) : make_uint4(0, 0, 0, 0);   // the group is inside the region even when k1 cuts it
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = base + (u * PA + threadIdx.x) * 8;
                apply_group8(slice, v[u], i < k1 ? k1 - i : 0u);
            }
        }
    } else {
        const uint32_t* keys = (const uint32_t*)keys_v;
        constexpr int U = 8;              // keys in flight per thread
        for (uint32_t base = k0; base < k1; base += U * PA) {
            uint32_t sl[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = base + u * PA + threadIdx.x;
                sl[u] = i < k1 ? keys[i] & smask : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (sl[u] != 0xffffffffu) sat_inc_lds(sl[u]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += PA) T[i] = slice[i];
}

// =====================================================================================================================
// Round 4: the two scatters without a histogram pass, without a scan and without a global atomic ("direct" form; k = 32, e = 3).
//
// What round 3's counters said (profiles/r03/sq_phase_a_after.txt): both scatters spend a quarter of their wave cycles issuing and
// half of them waiting -- a tile goes through hashing, histogram atomics, a scan, placement atomics and a copy-out that looks every
// key's bucket up again, with workgroup barriers between stretches that are each bound by a different unit (vector issue, LDS
// atomics, the store path), one 1024-thread workgroup per CU: the stretches ADD UP (stage ablation, profiles/r04/).  Here
//   * a tile is 256 buckets x a FIXED number of slots (64 four-byte keys in the read scatter, 128 two-byte keys in the key
//     scatter): a key takes its slot with ONE LDS atomic right after it is hashed or loaded -- no histogram, no scan, and the
//     keys never wait in registers for a second pass.  A bucket that runs over its slots (+ 5 sigma; hot k-mers) sends the key
//     straight to the table, like a full region always did: the result stays exact;
//   * tiles are 64 KiB and workgroups 512 threads, so TWO workgroups share a CU and drift apart: one hashes (vector issue) or takes
//     tickets (LDS atomics) while the other copies out (LDS reads, stores);
//   * the level-1 digit is the key's MIDDLE byte (bits 16..23), not its top byte: a hash is min(forward, reverse complement),
//     whose density falls linearly over the TOP bits, so top-byte buckets hold from twice the mean to nothing and no fixed slot
//     count fits them; the middle byte is uniform.  The second level sorts by the top byte (128 slots hold twice the mean of a
//     10 Ki-key tile + 5 sigma);
//   * every (level-1 bucket, workgroup) pair owns a PIECE of the key buffer, and the 512 pieces of a level-1 bucket (a segment)
//     are scattered by TWO workgroups (256 pieces each) into their own halves of the segment's 256 final-bucket regions, which
//     nobody else writes: run cursors live in LDS, no global atomicAdd anywhere (round 3: one per bucket and tile, issued early to
//     hide its round trip).  Consecutive runs of a piece come from the same CU, so their lines meet in ONE L2;
//   * the copy-out moves 16 bytes per lane (ds_read_b128 -> global_store_dwordx4, four buckets per wave instruction, every store
//     16-byte aligned): a bucket only ever writes a multiple of 4 (8) keys, the remainder is CARRIED to the head of its row for
//     the next tile and flushed key by key once, at the end.
// part_apply2 is part_apply over the two half regions of a final bucket q = key >> 16.
// Two geometries of the same kernels (debug bit 21 picks Small at run time; the table is the same):
//   Small  512-thread workgroups with 64 KiB tiles, two per CU; two workgroups per level-1 segment in the key scatter
//   Big    1024-thread workgroups with 128 KiB tiles, one per CU
struct GeomSmall {
    static constexpr int T = 512;      // threads per workgroup of both scatters
    static constexpr int S1 = 64;      // slots per bucket of the read scatter's tile: 256 x 64 x 4 B = 64 KiB
    static constexpr int RW = 3;       // reads per wave and tile: 24 reads, <= 8.6 K keys, 33.5 per bucket: 64 is + 5 sigma
    static constexpr int S2 = 128;     // 16-bit slots per bucket of the key scatter's tile: 256 x 128 x 2 B = 64 KiB
    static constexpr int KPT = 20;     // keys per thread and tile of the key scatter: 10 Ki keys, 40 per bucket, 80 for the first (+ 5 sigma = 125)
    static constexpr int GRID = 512;   // workgroups of the read scatter = pieces per level-1 bucket
    static constexpr int HALVES = 2;   // workgroups per level-1 segment in the key scatter, each with its own part of every final region
    static constexpr int CG = 8;       // keys a final bucket writes at a time (16 bytes): rows of 128 slots have no room for a longer carry
};
struct GeomBig {
    static constexpr int T = 1024;
    static constexpr int S1 = 128;     // 256 x 128 x 4 B = 128 KiB
    static constexpr int RW = 4;       // 64 reads, <= 22.8 K keys, 89 per bucket: 128 is + 4 sigma
    static constexpr int S2 = 256;     // 256 x 256 x 2 B = 128 KiB
    static constexpr int KPT = 20;     // 20 Ki keys, 80 per bucket, 160 for the first (+ a carry of <= 63 + 3 sigma: a row that runs over sends keys to the table);
                                       // 16 until round 6: 214 -> 206 ms per 100 M pairs (fewer barriers and row sweeps per key, a quarter more loads in flight);
                                       // 24 spills (128 registers), 32-key groups (half lines) lose 2-5 ms to whole lines at either size
    static constexpr int GRID = 256;
    static constexpr int HALVES = 1;
    static constexpr int CG = 64;      // whole 128-byte lines (round 6, see part_keys16_direct)
};
// keys a piece holds: its expected share of the chunk's keys + 1/16 + 512, a multiple of 32 keys (128 B)
__host__ __device__ inline uint32_t piece_keys(unsigned long long n_keys, int grid) {
    const unsigned long long mean = n_keys / (unsigned long long)(NBK * grid);
    return (uint32_t)((mean + mean / 16 + 512ull + 31ull) & ~31ull);
}
// The 16-bit key buffer of the direct form, SEGMENT-major: the 256 final buckets (top byte t) of segment m (middle byte) lie side by
// side -- the 256 streams a workgroup of the key scatter writes stay within 23 MB -- each sized by the expected load of its top byte
// (part_region over nb = 256 buckets and 1/256 of the chunk's keys) and split into two half regions, one per workgroup of the segment.
__host__ __device__ inline PartCap seg_cap(unsigned long long n_keys) { return PartCap{n_keys / NBK + NBK, (uint32_t)NBK}; }
// Every (half) region starts on a 128-byte line (64 keys) and holds a whole number of lines, so that a bucket which only ever writes
// whole lines (GeomBig::CG) writes every line of its region exactly once.
__host__ __device__ inline uint32_t seg_region(const PartCap& sc, uint32_t t) { return (part_region(sc, t) + 63u) & ~63u; }
__host__ __device__ inline uint32_t seg_size(const PartCap& sc) { return seg_region(sc, NBK); }
__host__ __device__ inline uint32_t half_region(const PartCap& sc, uint32_t t, int halves) { return ((seg_region(sc, t + 1) - seg_region(sc, t)) / (uint32_t)halves) & ~63u; }
__host__ __device__ inline size_t final_region(const PartCap& sc, uint32_t m, uint32_t t, uint32_t half, int halves) {
    return (size_t)m * seg_size(sc) + seg_region(sc, t) + (size_t)half * half_region(sc, t, halves);
}

// the segments of the chunk's long reads (more than FAST_NK k-mer offsets), in any order: entry = pair (inside the chunk) | mate << 24 |
// segment << 25; list[0] counts them
constexpr int LRS_ROUNDS = 16;            // reads per thread of long_read_segments
__global__ void __launch_bounds__(256) long_read_segments(ReadBatchDev b, long pair0, long npairs, int k, uint32_t* __restrict__ list, uint32_t cap) {
    // ONE atomic per workgroup of 4096 reads: one per read, or per wave, is a chain of round trips to one address of the L2 -- 1.2 ms per
    // chunk of 4 Mi pairs of 250-base reads, 36 ms per 100 M pairs, whoever adds (round 6)
    __shared__ uint32_t wsum[4], wbase;
    const long r0 = (long)blockIdx.x * (256 * LRS_ROUNDS);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t n_seg[LRS_ROUNDS], mine = 0;
#pragma unroll
    for (int i = 0; i < LRS_ROUNDS; i++) {
        const long r = r0 + (long)i * 256 + threadIdx.x;
        const int nk = r < 2 * npairs ? (int)((r & 1) ? b.len[1][pair0 + (r >> 1)] : b.len[0][pair0 + (r >> 1)]) - k + 1 : 0;
        n_seg[i] = nk > FAST_NK ? (uint32_t)((nk + FAST_NK - 1) / FAST_NK) : 0u;
        mine += n_seg[i];
    }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        wbase = total ? atomicAdd(list, total) : 0u;
    }
    __syncthreads();
    uint32_t at = wbase + incl - mine;
    for (int q = 0; q < wv; q++) at += wsum[q];
#pragma unroll
    for (int i = 0; i < LRS_ROUNDS; i++) {
        const long r = r0 + (long)i * 256 + threadIdx.x;
        for (uint32_t sg = 0; sg < n_seg[i]; sg++, at++)
            if (at < cap) list[1 + at] = (uint32_t)(r >> 1) | ((uint32_t)(r & 1) << 24) | (sg << 25);
    }
}

// LIST (round 5): the reads are SEGMENTS of long reads, listed by long_read_segments -- a read of more than FAST_NK k-mer offsets is
// cut into runs of FAST_NK offsets (offset 128 s is word 4 s of each plane, so a segment's windows are cut from its own <= 6 words
// per plane like a short read's), which go through this scatter like reads of <= 159 bases and land in the same pieces, behind the
// keys the first launch (the short reads; it passes the long ones over) left there.  Same keys as the read would give whole.
template <class G, bool LIST>
__global__ void __launch_bounds__(G::T) part_reads_direct(ReadBatchDev b, long pair0, long npairs, HashParams hp, uint32_t piece,
                                                         uint32_t* __restrict__ cnt1 /*[bucket][workgroup]*/, uint32_t* __restrict__ out,
                                                         uint32_t* __restrict__ counts, int ablate /* stage timing (LHGT_PART_ABLATE), results wrong: 1 no stores, 2 no placement */,
                                                         const uint32_t* __restrict__ list /* LIST: [0] = how many, then pair | mate << 24 | segment << 25 */) {
    // Rows of S1 slots, RS1 = S1 + 4 words apart: every row starts 16 bytes further round the banks than the one before (all buckets
    // fill at the same pace: with rows a multiple of the bank count apart a wave's stores would crowd into the few banks of the
    // current position), and a slot's byte address is ONE multiply-add of bucket and ticket.  (Until the instruction counters of the
    // 24-bit version were read -- profiles/r04/sq_phase_a_direct.txt: 42 lane-instructions per key, more than round 3's 38 -- rows
    // were S1 apart and rotated by 4 (bucket & 7) slots: eight instructions per key for the address alone.)
    constexpr int RS1 = G::S1 + 4;
    constexpr uint32_t ROW_BYTES = RS1 * 4u, LIMIT = G::S1 * 4u;   // tickets count BYTES: a ticket is the slot's offset in its row
    __shared__ __align__(16) uint32_t tile[NBK * RS1];
    __shared__ uint32_t cnt[NBK], cur[NBK];  // cnt: bytes taken in the row (4 per key); cur: keys already in the piece
    __shared__ uint32_t dump[G::T + 64];     // per-thread dummy counter (zeroed every tile: a dead offset's ticket never looks like an overflow), per-lane dummy word
    constexpr int STAGE_W = 20;              // <= 18 record words per read on this path (<= 159 bases) + the word a window may look past
    constexpr int NW = G::T / 64;
    __shared__ uint32_t stage_all[1 + NW * G::RW * STAGE_W];   // one word in front: the first window of a read looks at the word before it
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    uint32_t* stage = stage_all + 1 + wib * G::RW * STAGE_W;
    // window of offset j = bits j .. j + 31 of the plane = alignbit(W[q], W[q + 1], 32 - r) with q = j >> 5, r = j & 31 -- but a shift of
    // 32 does not exist.  With q' = (j - 1) >> 5 and shift (32 - r) & 31 the same instruction gives W[q' + 1] = W[q] for r = 0 and is
    // unchanged for r > 0: no select per window (the word in front of a read's first is read and shifted out)
    const int wsel = ((lane + 31) >> 5) - 1;
    const uint32_t sh_win = (32u - (uint32_t)(lane & 31)) & 31u;
    constexpr int k = 32;
    const long n_reads = LIST ? (long)list[0] : 2 * npairs;
    constexpr int RPT = NW * G::RW;           // reads per tile (even)
    const long n_tiles = (n_reads + RPT - 1) / RPT;
    if (threadIdx.x < NBK) { cnt[threadIdx.x] = 0; cur[threadIdx.x] = LIST ? cnt1[threadIdx.x * G::GRID + blockIdx.x] : 0; }
    // A wave's reads of a tile are r0 + wave + 8 rr with r0 even: their mate (r & 1) is the parity of the wave's number, the same
    // for every read the wave will ever see -- the descriptor arrays of that mate are picked ONCE (indexing b.len[m] with a
    // run-time m made the compiler fetch the pointer from the kernel-argument block with a vector load and wait for it, read by
    // read: a chain of dependent round trips at the top of every tile).
    const int mate = wib & 1;
    const uint16_t* __restrict__ len_m = mate ? b.len[1] : b.len[0];
    const uint32_t* __restrict__ off_m = mate ? b.off[1] : b.off[0];
    // (every load unconditional -- without per-pair flags the flag byte is read from the length array and ignored: a load under
    // `if (b.flags)` is followed by its own s_waitcnt vmcnt(0), which also waits for the record loads issued just before)
    const bool have_flags = b.flags != nullptr;
    const uint8_t* __restrict__ flag_m = have_flags ? b.flags : (const uint8_t*)len_m;
    // Three tiles are in flight per wave: the one being hashed (its lengths in dA, its records in LDS), the next one (descriptors
    // arrived in dB, records on their way into recB) and the one after (descriptors on their way into dC) -- two dependent round
    // trips to memory, each given a whole tile's time.
    struct Desc { int len[G::RW]; uint32_t off[G::RW]; uint32_t stride[LIST ? G::RW : 1]; };
    // LIST: a tile's list entries are a round trip of their own in front of its descriptors -- requested one tile earlier still (round 6:
    // taken inside load_descriptors, every tile waited for them: 253 ms per 100 M pairs of 250-base reads where the keys' share is 168)
    struct Ent { uint32_t e[LIST ? G::RW : 1]; };
    auto load_entries = [&](long t, Ent& en) {
        if constexpr (LIST) {
            const long r0 = t * RPT, r1 = r0 + RPT < n_reads ? r0 + RPT : n_reads;
#pragma unroll
            for (int rr = 0; rr < G::RW; rr++) {
                const long r = r0 + wib + rr * NW;
                en.e[rr] = list[1 + (r < r1 ? r : (n_reads > 0 ? n_reads - 1 : 0))];
            }
        }
    };
    auto load_descriptors = [&](long t, Desc& d, const Ent& en) {        // t >= n_tiles: every length 0
        const long r0 = t * RPT, r1 = r0 + RPT < n_reads ? r0 + RPT : n_reads;
        int len[G::RW];
        uint32_t fl[G::RW];
        if constexpr (LIST) {
            const uint32_t (&ent)[G::RW] = en.e;
#pragma unroll
            for (int rr = 0; rr < G::RW; rr++) {
                const long p = pair0 + (long)(ent[rr] & 0xffffffu);
                const int m = (int)((ent[rr] >> 24) & 1u);
                len[rr] = m ? b.len[1][p] : b.len[0][p];
                d.off[rr] = m ? b.off[1][p] : b.off[0][p];
                fl[rr] = have_flags ? b.flags[p] : 0xffu;
            }
#pragma unroll
            for (int rr = 0; rr < G::RW; rr++) {
                const long r = r0 + wib + rr * NW;
                const int m = (int)((ent[rr] >> 24) & 1u), seg = (int)(ent[rr] >> 25);
                const bool counted = (fl[rr] >> m) & 1u;
                const int nk_left = len[rr] - k + 1 - FAST_NK * seg;                    // offsets from this segment's first one on
                const int nkv = nk_left < FAST_NK ? nk_left : FAST_NK;
                d.stride[rr] = (uint32_t)(((len[rr] + 31) >> 5) + 1);
                d.off[rr] += 4u * (uint32_t)seg;                                        // FAST_NK offsets = 4 words of a plane
                d.len[rr] = r < r1 && counted && nkv > 0 ? nkv + k - 1 : 0;
            }
            return;
        }
#pragma unroll
        for (int rr = 0; rr < G::RW; rr++) {
            const long r = r0 + wib + rr * NW;
            const long rc = r < r1 ? r : (n_reads > 0 ? n_reads - 2 + mate : 0);   // clamped to the last read of this wave's mate
            const long p = pair0 + (rc >> 1);
            len[rr] = len_m[p];
            d.off[rr] = off_m[p];
            fl[rr] = flag_m[p];
        }
#pragma unroll
        for (int rr = 0; rr < G::RW; rr++) {
            const long r = r0 + wib + rr * NW;
            const bool counted = !have_flags || ((fl[rr] >> mate) & 1u);   // quirk Q4, thread-chunk emulation
            // a read of more than 159 bases (more than FAST_NK offsets: it would overrun its staging words and the tile's rows) is
            // passed over here: its segments come with the LIST launch behind this one (lhgt_count_batch_partitioned)
            d.len[rr] = r < r1 && counted && len[rr] - k + 1 <= FAST_NK ? len[rr] : 0;
        }
    };
    auto load_records = [&](const Desc& d, uint32_t (&rec)[G::RW]) {
#pragma unroll
        for (int rr = 0; rr < G::RW; rr++) {
            const int wpr = ((d.len[rr] + 31) >> 5) + 1;
            if constexpr (LIST) {       // the segment's wpr words of each plane of the long read's record (planes d.stride words apart)
                const int pl = (lane >= wpr) + (lane >= 2 * wpr), w = lane - pl * wpr;
                rec[rr] = b.words[d.off[rr] + (lane < 3 * wpr ? (uint32_t)pl * d.stride[rr] + (uint32_t)w : 0u)];
            } else rec[rr] = b.words[d.off[rr] + (lane < 3 * wpr ? lane : 0)];
        }
    };
    Desc dA, dB, dC;
    Ent entC, entN;
    uint32_t recA[G::RW], recB[G::RW];
    if (n_reads > 0) {
        load_entries(blockIdx.x, entC);
        load_entries((long)blockIdx.x + gridDim.x, entN);
        load_descriptors(blockIdx.x, dA, entC);
        load_descriptors((long)blockIdx.x + gridDim.x, dB, entN);
        load_entries((long)blockIdx.x + 2L * gridDim.x, entC);
        load_records(dA, recA);
    }
    __syncthreads();
    // piece (bucket, workgroup) = piece number workgroup * 256 + bucket: the 256 streams a workgroup writes lie side by side in 25 MB of
    // the buffer (a dozen 2 MiB pages) -- bucket-major, they were 50 MB apart and every store instruction missed the CU's TLB
    // Level-1 keys leave as 24 bits: inside piece (m, w) the middle byte IS m, so a key is (top byte << 16 | low 16 bits); four keys
    // make three words (12 bytes per lane, global_store_dwordx3) -- 9 instead of 12 bytes per key through the level-1 buffer.
    const size_t piece_words = (size_t)piece / 4 * 3;
    uint32_t* const my_out = out + (size_t)blockIdx.x * NBK * piece_words;
    const size_t bucket_stride = piece_words;
    for (long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        // this tile's records go to the wave's staging words; the next tile's records and the descriptors of the one after are
        // requested now and not looked at before the next iteration
        int cl[G::RW];
#pragma unroll
        for (int rr = 0; rr < G::RW; rr++) {
            cl[rr] = dA.len[rr];
            stage[rr * STAGE_W + (lane < STAGE_W ? lane : 0)] = recA[rr];   // lanes >= 3 wpr hold a repeat of word 0
        }
        __builtin_amdgcn_wave_barrier();
        load_records(dB, recB);
        load_entries(t + 3L * gridDim.x, entN);
        load_descriptors(t + 2L * gridDim.x, dC, entC);
        // Software pipeline over the wave's reads: the tickets of read r are drawn (six LDS atomics issued back to back), then read
        // r + 1 is HASHED while they are in flight, then the slots of read r are computed and stored.  (Hash, tickets, wait, stores
        // read by read left the vector unit idle during every wait and the LDS idle during every hash: with all sixteen waves of the
        // workgroup in the same stretch at the same time, the two times added up.)
        uint32_t key[2][6], pos[6];
        bool live[2][2];
        dump[threadIdx.x] = 0u;
        auto hash_read = [&](int rr, uint32_t (&kk)[6], bool (&lv)[2]) {
            const int len = cl[rr];
            const int nk = len - k + 1;                             // <= 0: a padding read, every offset dead
            const int wpr = ((len + 31) >> 5) + 1;
            // all six window word pairs of the read first (unconditional: the staging words behind a short read are there, just not
            // meant), ONE wait, then the arithmetic
            uint32_t ww[2][3][2];
#pragma unroll
            for (int it = 0; it < 2; it++)
#pragma unroll
                for (int plane = 0; plane < 3; plane++) {
                    const uint32_t* w = stage + rr * STAGE_W + plane * wpr + 2 * it + wsel;
                    ww[it][plane][0] = w[0];
                    ww[it][plane][1] = w[1];
                }
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int j = it * 64 + lane;
                auto window = [&](int plane) { return __builtin_amdgcn_alignbit(ww[it][plane][0], ww[it][plane][1], sh_win); };
                lv[it] = (j < nk) & (window(2) == 0);
                const uint32_t whi = window(0), wlo = window(1);
                const uint32_t rhi = __brev(whi), rlo = __brev(wlo);
                const uint32_t fx = whi ^ wlo, rx = ~(rhi ^ rlo);
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    // hash_from_windows at k = 32, where the three masks cover every bit: the final `^ kmask` is a NOT, folded into
                    // the outer select's truth table (0x35 = ~0xCA)
                    const uint32_t* mk = hp.mask[i];
                    const uint32_t fwd = __builtin_amdgcn_bitop3_b32(mk[0], fx, bit_select(mk[1], whi, wlo), 0x35);
                    const uint32_t rc = bit_select(mk[0], rx, bit_select(mk[1], rhi, rlo));
                    kk[it * 3 + i] = fwd < rc ? fwd : rc;
                }
            }
        };
        // six keys, six tickets back to back; a dead offset (beyond the read, a k-mer with an N) draws from a per-lane dummy counter
        // and writes a per-lane dummy word: no branch per key
        auto draw = [&](const uint32_t (&kk)[6], const bool (&lv)[2]) {
#pragma unroll
            for (int u = 0; u < 6; u++) {
                uint32_t* ctr = lv[u / 3] ? &cnt[(kk[u] >> 16) & 0xffu] : &dump[threadIdx.x];
                pos[u] = atomicAdd(ctr, 4u);
            }
        };
        // -> did a bucket of this read run over its slots?  (one maximum over the six tickets: a dead offset's are small)
        auto place = [&](const uint32_t (&kk)[6], const bool (&lv)[2]) -> bool {
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const bool ok = lv[u / 3] && pos[u] < LIMIT;
                const uint32_t bk = (kk[u] >> 16) & 0xffu;
                uint32_t* dst = ok ? (uint32_t*)((char*)tile + (bk * ROW_BYTES + pos[u])) : &dump[G::T + lane];
                *dst = kk[u];
            }
            const uint32_t m0 = max(max(pos[0], pos[1]), pos[2]), m1 = max(max(pos[3], pos[4]), pos[5]);
            return max(m0, m1) >= LIMIT;
        };
        if (ablate & 2) {                                           // stage timing: the hashes only (kept alive through the dummy words)
#pragma unroll
            for (int rr = 0; rr < G::RW; rr++) {
                hash_read(rr, key[0], live[0]);
                dump[G::T + lane] = key[0][0] ^ key[0][1] ^ key[0][2] ^ key[0][3] ^ key[0][4] ^ key[0][5];
            }
        } else {
            hash_read(0, key[0], live[0]);
#pragma unroll
            for (int rr = 0; rr < G::RW; rr++) {
                draw(key[rr & 1], live[rr & 1]);
                if (rr + 1 < G::RW) hash_read(rr + 1, key[(rr + 1) & 1], live[(rr + 1) & 1]);
                const bool over = place(key[rr & 1], live[rr & 1]);
                // a bucket ran over its slots (hot k-mers): those keys go to the table now (the read's keys are still in registers)
                if (__ballot(over)) {
#pragma unroll
                    for (int u = 0; u < 6; u++)
                        if (live[rr & 1][u / 3] && pos[u] >= LIMIT) part_sat_inc(counts, key[rr & 1][u]);
                }
            }
        }
        __syncthreads();
        // copy-out: a row is LPR lanes x 4 keys, a wave instruction covers 64 / LPR buckets; a bucket writes a multiple of four keys and
        // carries the rest to the head of its row
        constexpr int LPR = G::S1 / 4, BPI = 64 / LPR;
#pragma unroll
        for (int g = 0; g < NBK / (NW * BPI); g++) {
            const int bq = wib * (NBK / NW) + g * BPI + lane / LPR;
            const uint32_t j = (uint32_t)(lane % LPR) * 4u;
            const uint32_t n_all = cnt[bq] >> 2, c = cur[bq];
            const uint32_t have = n_all < (uint32_t)G::S1 ? n_all : (uint32_t)G::S1;       // keys that found a slot
            const uint32_t full = have & ~3u;
            const uint32_t room = piece > c ? piece - c : 0u;                            // piece and c are multiples of 4
            const uint32_t put = full < room ? full : room;
            const uint4 v = *(const uint4*)&tile[bq * RS1 + j];
            if (!(ablate & 1)) {
                if (j < put) {
                    // four keys (bytes: low 16 bits, middle byte, top byte) -> three words of 24-bit keys (low 16 bits, top byte): three
                    // byte permutes (the shifts and masks they replace were twelve instructions)
                    uint32_t* dst = my_out + (size_t)bq * bucket_stride + (size_t)(c + j) / 4 * 3;
                    const uint32_t w0 = __builtin_amdgcn_perm(v.y, v.x, 0x04030100u), w1 = __builtin_amdgcn_perm(v.z, v.y, 0x05040301u),
                                   w2 = __builtin_amdgcn_perm(v.w, v.z, 0x07050403u);
                    *(uint3*)dst = make_uint3(w0, w1, w2);          // global_store_dwordx3
                } else if (j < full) {                              // piece full: the rest goes straight to the table (exact either way)
                    part_sat_inc(counts, v.x); part_sat_inc(counts, v.y); part_sat_inc(counts, v.z); part_sat_inc(counts, v.w);
                }
            } else if (v.x + v.y == 0x12345u && v.z == v.w) my_out[c + j] = v.x;
            if (j == full && have > full) *(uint4*)&tile[bq * RS1] = v;                  // the carry (its first have - full words) to slot 0
            if (lane % LPR == 0) { cur[bq] = c + put; cnt[bq] = (have - full) << 2; }
        }
        __syncthreads();
        dA = dB;
        dB = dC;
        entC = entN;
#pragma unroll
        for (int rr = 0; rr < G::RW; rr++) recA[rr] = recB[rr];
    }
    // the carried keys (fewer than four per bucket and workgroup: 10^5 of a launch's 3 * 10^9): straight to the table -- the pieces only
    // hold whole groups of four
    if (threadIdx.x < NBK) {
        const int bq = threadIdx.x;
        const uint32_t rem = cnt[bq] >> 2;
        for (uint32_t i = 0; i < rem; i++) part_sat_inc(counts, tile[bq * RS1 + i]);
        cnt1[bq * G::GRID + blockIdx.x] = cur[bq];
    }
}

// level-1 segment m (= the 512 pieces of bucket m, one per workgroup of the read scatter) -> its 256 final buckets (top byte t,
// middle byte m), as 16-bit keys.  Two workgroups per segment (256 pieces each), each writing its own half of every final region.
// CG = keys a bucket writes at a time.  8 (round 4): 16 bytes, a bucket's runs of a tile begin and end anywhere inside a line, and the
// L2 has long written a half-filled line back when the next tile completes it: WRITE_SIZE 199 GB per 100 M pairs for 143 GB of keys
// (profiles/r06/pmc_live_uhgg.json).  64 (round 6, GeomBig): a bucket keeps up to 63 keys back and only ever writes whole, aligned
// 128-byte lines, each exactly once.
template <class G, int CG>
__global__ void __launch_bounds__(G::T) part_keys16_direct(const uint32_t* __restrict__ in, const uint32_t* __restrict__ cnt1, uint32_t piece,
                                                          PartCap pc, uint32_t* __restrict__ cur2 /*[half][final bucket]*/, uint16_t* __restrict__ out,
                                                          uint32_t* __restrict__ counts, int ablate /* 1 no stores, 2 no placement */) {
    // rows of S2 two-byte slots, RS2 = S2 + 8 slots (16 bytes) apart; tickets count BYTES (2 per key): as in the read scatter
    constexpr int RS2 = G::S2 + 8;
    constexpr uint32_t ROW_BYTES = RS2 * 2u, LIMIT = G::S2 * 2u;
    __shared__ __align__(16) uint16_t tile[NBK * RS2];
    __shared__ uint32_t cnt[NBK], cur[NBK], rcap[NBK], hist[NBK], pref[NBK + 1], pcnt[NBK], wsum[4];
    __shared__ size_t rbase[NBK];             // a chunk of 8 Mi pairs holds more than 2^32 final keys
    __shared__ uint32_t dump[G::T + 64];      // per-thread dummy counter (zeroed every tile), per-lane dummy word
    const uint32_t m = blockIdx.x / (uint32_t)G::HALVES, half = blockIdx.x % (uint32_t)G::HALVES;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    constexpr int NW = G::T / 64;
    constexpr int PPW = G::GRID / G::HALVES;                        // pieces per workgroup (= 256)
    static_assert(PPW == NBK, "a workgroup of the key scatter walks 256 pieces");
    const uint32_t w0 = half * (uint32_t)PPW;                      // first piece of this part
    constexpr uint32_t TK = (uint32_t)G::T * G::KPT;
    if (threadIdx.x < NBK) {
        const uint32_t c0 = cnt1[m * G::GRID + w0 + threadIdx.x];
        const uint32_t c = c0 < piece ? c0 : piece;
        pcnt[threadIdx.x] = c;
        hist[threadIdx.x] = (c + TK - 1u) / TK;                  // tiles of this piece
        cnt[threadIdx.x] = 0;
        cur[threadIdx.x] = 0;
        rbase[threadIdx.x] = final_region(pc, m, threadIdx.x, half, G::HALVES);   // final bucket (top byte = threadIdx.x, middle byte = m); pc = seg_cap
        rcap[threadIdx.x] = half_region(pc, threadIdx.x, G::HALVES);
    }
    __syncthreads();
    {
        const uint32_t o = bucket_excl_scan(hist, NBK, wsum);
        if (threadIdx.x < NBK) {
            pref[threadIdx.x] = o;
            if (threadIdx.x == NBK - 1) pref[NBK] = o + hist[threadIdx.x];
        }
    }
    __syncthreads();
    const uint32_t n_tiles = pref[NBK];                           // pref[w] = tiles before piece w
    // A tile is TK consecutive slots of ONE piece (its last tile is short): every load of the kernel is unconditional, from one
    // base pointer per tile, on a clamped index -- so that the compiler can count them (s_waitcnt vmcnt(N)).  Loads under a
    // lane-dependent branch, or issued on only one side of a uniform one, are waited for with vmcnt(0), which also waits for the
    // NEXT tile's loads issued just before: no prefetch at all (the first version, which walked the pieces as one stream: load,
    // placement and copy-out times simply added up).
    uint32_t pw = 0;                                              // piece of the tile being loaded (uniform, only moves forward)
    uint32_t key[G::KPT], knext[G::KPT];
    static_assert(G::KPT % 4 == 0, "a thread loads whole groups of four 24-bit keys");
    const uint32_t hi_m = m << 16;                                // the middle byte every key of this segment has
    auto key32 = [&](uint32_t k24) { return ((k24 >> 16) << 24) | hi_m | (k24 & 0xffffu); };
    auto load_keys = [&](uint32_t tl, uint32_t (&kk)[G::KPT]) -> uint32_t {   // returns the tile's number of keys (0 past the end)
        const uint32_t tc = tl < n_tiles ? tl : n_tiles - 1u;    // past the end: the last tile once more, ignored
        while (tc >= pref[pw + 1]) pw++;                          // uniform; tc < n_tiles = pref[256]
        const uint32_t o = (tc - pref[pw]) * TK, left = pcnt[pw] - o;   // pcnt, o: multiples of 4
        const uint32_t nv = left < TK ? left : TK;
        // piece (bucket m, workgroup w) = piece number w * 256 + m, piece / 4 * 3 words each; group g of four keys = words 3 g .. 3 g + 2
        const uint32_t* const base = in + ((size_t)(w0 + pw) * NBK + m) * ((size_t)piece / 4 * 3) + (size_t)o / 4 * 3;
        const uint32_t ng = nv / 4;
#pragma unroll
        for (int u = 0; u < G::KPT / 4; u++) {
            const uint32_t gi = (uint32_t)u * (uint32_t)G::T + threadIdx.x;
            const uint32_t* q = base + (size_t)(gi < ng ? gi : ng - 1u) * 3;
            const uint3 w3 = *(const uint3*)q;                    // global_load_dwordx3
            // three words -> four 24-bit keys (top byte << 16 | low 16 bits), kept that way: the ticket needs the top byte, the store the
            // low half; only the overflow path wants the 32-bit key back (key32 below)
            kk[4 * u + 0] = w3.x & 0xffffffu;
            kk[4 * u + 1] = __builtin_amdgcn_perm(w3.y, w3.x, 0x0c050403u);
            kk[4 * u + 2] = __builtin_amdgcn_perm(w3.z, w3.y, 0x0c040302u);
            kk[4 * u + 3] = w3.z >> 8;
        }
        return tl < n_tiles ? nv : 0u;
    };
    if (n_tiles == 0) {
        if (threadIdx.x < NBK) cur2[half * (uint32_t)(NBK * NBK) + threadIdx.x * (uint32_t)NBK + m] = 0u;
        return;
    }
    uint32_t nv_cur = load_keys(0, key);
    for (uint32_t tl = 0; tl < n_tiles; tl++) {
        const uint32_t nv_next = load_keys(tl + 1, knext);       // the next tile's keys: requested before this tile is placed, first looked at a whole tile later
        uint32_t ovmask = 0;
        dump[threadIdx.x] = 0u;
        if (ablate & 2) {
            uint32_t x = 0;
#pragma unroll
            for (int u = 0; u < G::KPT; u++) x ^= key[u];
            dump[G::T + lane] = x;
        } else {
            // all tickets of the tile back to back, then the stores.  A tile holds whole groups of four keys: one liveness test per group
            const uint32_t ng_cur = nv_cur >> 2;
            uint32_t pos[G::KPT];
            bool lvg[G::KPT / 4];
#pragma unroll
            for (int g4 = 0; g4 < G::KPT / 4; g4++) lvg[g4] = (uint32_t)g4 * (uint32_t)G::T + threadIdx.x < ng_cur;
#pragma unroll
            for (int u = 0; u < G::KPT; u++) {
                uint32_t* ctr = lvg[u / 4] ? &cnt[key[u] >> 16] : &dump[threadIdx.x];
                pos[u] = atomicAdd(ctr, 2u);
            }
            uint32_t mx = 0;
#pragma unroll
            for (int u = 0; u < G::KPT; u++) {
                const bool ok = lvg[u / 4] && pos[u] < LIMIT;
                uint16_t* dst = ok ? (uint16_t*)((char*)tile + ((key[u] >> 16) * ROW_BYTES + pos[u])) : (uint16_t*)&dump[G::T + lane];
                *dst = (uint16_t)key[u];
                mx = max(mx, pos[u]);
            }
            // a bucket over its slots (a dead key's tickets are small: its counter starts every tile at 0): which keys, for the table below
            if (__ballot(mx >= LIMIT)) {
#pragma unroll
                for (int u = 0; u < G::KPT; u++)
                    if (lvg[u / 4] && pos[u] >= LIMIT) ovmask |= 1u << u;
            }
        }
        __syncthreads();
        // copy-out: a row is LPR lanes x 8 keys (16 bytes); multiples of CG keys leave, the rest is carried
        constexpr int LPR = G::S2 / 8, BPI = 64 / LPR;
        static_assert(CG % 8 == 0 && CG <= 64 && G::S2 % CG == 0, "a bucket writes whole 16-byte groups, at most a line at a time");
#pragma unroll
        for (int g = 0; g < NBK / (NW * BPI); g++) {
            const int bq = wib * (NBK / NW) + g * BPI + lane / LPR;
            const uint32_t j = (uint32_t)(lane % LPR) * 8u;
            const uint32_t n_all = cnt[bq] >> 1, c = cur[bq];
            const uint32_t have = n_all < (uint32_t)G::S2 ? n_all : (uint32_t)G::S2;
            const uint32_t full = have & ~(uint32_t)(CG - 1);
            const uint32_t room = rcap[bq] > c ? rcap[bq] - c : 0u;                      // both multiples of 64 (half_region)
            const uint32_t put = full < room ? full : room;
            const uint4 v = *(const uint4*)&tile[bq * RS2 + j];
            if (!(ablate & 1)) {
                if (j < put) *(uint4*)(out + rbase[bq] + c + j) = v;
                else if (j < full) {                                // region full: count those keys now (see the header of this file)
                    const uint32_t hi16 = ((uint32_t)bq << 24) | (m << 16);
                    const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 8; q++) part_sat_inc(counts, hi16 | ((w4[q >> 1] >> ((q & 1) * 16)) & 0xffffu));
                }
            } else if (v.x + v.y == 0x12345u && v.z == v.w) out[c + j] = (uint16_t)v.x;
            // the carry (have - full keys, fewer than CG) to the head of the row: every lane of the row has read its group above, so a
            // group may land on one another lane held
            if (j >= full && j < have) *(uint4*)&tile[bq * RS2 + (j - full)] = v;
            if (lane % LPR == 0) { cur[bq] = c + put; cnt[bq] = (have - full) << 1; }
        }
        __syncthreads();
        // keys whose bucket was over its slots: straight to the table, HERE -- a global read-modify-write loop in the middle of the
        // placement makes every later s_waitcnt a vmcnt(0), and the second group of tickets then waits for the NEXT tile's loads
        if (__ballot(ovmask != 0u)) {
#pragma unroll
            for (int u = 0; u < G::KPT; u++)
                if ((ovmask >> u) & 1u) part_sat_inc(counts, key32(key[u]));
        }
#pragma unroll
        for (int u = 0; u < G::KPT; u++) key[u] = knext[u];
        nv_cur = nv_next;
    }
    if (threadIdx.x < NBK) {
        const int bq = threadIdx.x;
        uint32_t c = cur[bq];
        const uint32_t rem = cnt[bq] >> 1;
        for (uint32_t i = 0; i < rem; i++) {
            const uint16_t kk = tile[bq * RS2 + i];
            if (c < rcap[bq]) out[rbase[bq] + c++] = kk;
            else part_sat_inc(counts, ((uint32_t)bq << 24) | (m << 16) | kk);
        }
        cur2[half * (uint32_t)(NBK * NBK) + (uint32_t)bq * (uint32_t)NBK + m] = c;   // [part][final bucket = top byte * 256 + middle byte]
    }
}

// part_apply<true> over the two half regions of a final bucket
template <int HALVES>
__global__ void __launch_bounds__(PA) part_apply2(const uint16_t* __restrict__ keys, const uint32_t* __restrict__ n_keys /*[2][nb]*/, PartGeom g,
                                                  PartCap pc, uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t slice[];   // 2^16 / 16 words
    const uint32_t fb = blockIdx.x;                                 // slice of the table = key >> 16 = (top byte, middle byte)
    const uint32_t hr = half_region(pc, fb >> 8, HALVES);
    const size_t r0 = final_region(pc, fb & 0xffu, fb >> 8, 0, HALVES);   // pc = seg_cap
    uint32_t nh[HALVES], n_all = 0;
#pragma unroll
    for (int h = 0; h < HALVES; h++) {
        const uint32_t n = n_keys[(uint32_t)h * (uint32_t)g.nb + fb];
        nh[h] = n < hr ? n : hr;
        n_all += nh[h];
    }
    if (n_all == 0) return;               // untouched slice: nothing to read or write
    const int words = ((1 << g.slot_bits) + 15) >> 4;
    uint32_t* T = counts + (size_t)fb * words;
    for (int i = threadIdx.x; i < words; i += PA) slice[i] = T[i];
    __syncthreads();
    constexpr int U = 2;              // 16-byte groups in flight per thread
#pragma unroll
    for (int h = 0; h < HALVES; h++) {
        const uint16_t* const kh = keys + r0 + (size_t)h * hr;        // this part's keys: positions 0 .. nh[h]
        const uint32_t k1 = nh[h];
        for (uint32_t base = 0; base < k1; base += U * PA * 8) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = base + (u * PA + threadIdx.x) * 8;
                // (read once: non-temporal loads leave the L2 to the table's slices, 35.8 -> 33.7 ms per 100 M pairs)
                typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
                v4u_t t4 = {0u, 0u, 0u, 0u};
                if (i < k1) t4 = __builtin_nontemporal_load((const v4u_t*)(kh + i));   // the group is inside the half region even when k1 cuts it
                v[u] = make_uint4(t4.x, t4.y, t4.z, t4.w);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = base + (u * PA + threadIdx.x) * 8;
                apply_group8(slice, v[u], i < k1 ? k1 - i : 0u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += PA) T[i] = slice[i];
}

// lhgt_work_stats [0]: the sum of a chunk's cursors -- the sorted tiles' level-1 cursors (every key the read scatter routed), the
// direct form's final-bucket cursors (keys that reached a region; the few applied to the table on the way are not in them)
__global__ void __launch_bounds__(256) part_sum_cursors(const uint32_t* __restrict__ cur1, int n, unsigned long long* __restrict__ out) {
    unsigned long long v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += cur1[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}

}  // namespace lhgt

using namespace lhgt;

// Partitioned count of one resident batch; called by lhgt_count_kmers.  Key buffers live in ctx.
int lhgt_count_batch_partitioned(lhgt_ctx* ctx, const lhgt::ReadBatch& b) {
    const PartGeom g = part_geom(ctx->k);
    const int batch_nk = b.max_len - ctx->k + 1;
    if (batch_nk <= 0) return LHGT_OK;
    // round 4's direct form of the two scatters (k = 32, e = 3); LHGT_DEBUG bit 16: round 3's sorted tiles.  It takes reads of up to
    // FAST_NK k-mer offsets (159 bases) as they are; longer reads (round 5) go through it cut into segments of FAST_NK offsets
    // (part_reads_direct<G, true> on the list long_read_segments makes) -- round 4 sent a batch with ONE long read through the generic
    // scatter, and a batch of 250-base reads altogether.
    const bool direct_form = ctx->k == 32 && ctx->e == 3 && !(ctx->debug & 65536) && b.n_long >= 0;
    const bool have_long = direct_form && batch_nk > FAST_NK;
    const int max_nk = have_long ? FAST_NK : batch_nk;
    // keys a pair brings: at most 2 max_nk e; with long reads in the batch its own average (+ 1/8: the chunks of a batch differ) --
    // a region or piece that still runs over sends its keys straight to the table, exact either way
    const long keys_per_pair = have_long ? (long)((double)b.n_kmers * ctx->e / (double)std::max(1L, b.d.n_pairs) * 1.125) + 8 : 2L * max_nk * ctx->e;
    int reads_per_tile = (int)(TILE_KEYS / ((long)max_nk * ctx->e));
    if (reads_per_tile < 1) LHGT_FAIL(LHGT_E_ARG, "read of %d bases with e=%d exceeds the partition tile", b.max_len, ctx->e);
    // chunk of pairs whose bucket regions fit the two key buffers (u32 offsets: < 2^32 keys per buffer)
    auto cap_of = [&](long np) { return PartCap{(unsigned long long)np * keys_per_pair, (uint32_t)g.nb}; };
    auto need_of = [&](long np) {
        const PartCap c = cap_of(np);
        return c.n + c.n / 16 + (unsigned long long)g.nb * 512ull + 64;
    };
    static const long chunk_env = getenv("LHGT_PART_CHUNK") ? atol(getenv("LHGT_PART_CHUNK")) : 0;   // pairs per chunk of the direct form (A/B)
    long chunk_max = direct_form ? (chunk_env > 0 ? chunk_env : (8L << 20)) : (4L << 20);   // the direct form addresses its buffers with 64 bits
    if (have_long) {                      // as many keys per chunk as 8 Mi pairs of 150-base reads bring, whatever the reads' lengths
        const long same_keys = (long)((8L << 20) * 714.0 / (double)keys_per_pair);
        chunk_max = std::max(1L << 16, std::min(chunk_max, same_keys));
    }
    long want = b.d.n_pairs < chunk_max ? b.d.n_pairs : chunk_max;
    while (!direct_form && want > 1 && (need_of(want) >= (1ull << 32) || cap_of(want).n >= (1ull << 32))) want /= 2;
    // keys the two buffers must hold for chunks of np pairs: buffer 0 takes need x 4 bytes, buffer 1 need x 2
    auto keys_for = [&](long np) -> size_t {
        if (!direct_form) return (size_t)need_of(np);
        // buffer 0 holds 65536 pieces of 24-bit keys, buffer 1 the final regions
        const size_t need_pieces = std::max((size_t)piece_keys(cap_of(np).n, GeomBig::GRID) * (size_t)(NBK * GeomBig::GRID),
                                            (size_t)piece_keys(cap_of(np).n, GeomSmall::GRID) * (size_t)(NBK * GeomSmall::GRID)) + 64;   // either geometry
        const size_t need_final = (size_t)seg_size(seg_cap(cap_of(np).n)) * NBK + 64;
        return std::max((need_pieces * 3 + 3) / 4, need_final);
    };
    size_t need = keys_for(want);
    if (ctx->part_keys_cap < need) {
        // A loader that counts batch by batch says how large its batches will get (part_reserve_pairs): the buffers are made for
        // that at once -- growing them batch after batch is a free and an allocation of tens of GB each time, and hipMalloc right
        // after such a free was measured to take seconds (lhgt_common.hpp: dev_free).  Where the
        // device has no room for the wish the chunk is halved until its buffers fit (the table and a 156 GB index come first).
        long want_alloc = std::max(want, std::min(chunk_max, ctx->part_reserve_pairs));
        while (!direct_form && want_alloc > want && (need_of(want_alloc) >= (1ull << 32) || cap_of(want_alloc).n >= (1ull << 32))) want_alloc /= 2;
        for (int i = 0; i < 2; i++) {
            if (ctx->d_part_keys[i]) lhgt::dev_free(ctx->d_part_keys[i]);
            ctx->d_part_keys[i] = nullptr;
        }
        ctx->part_keys_cap = 0;
        for (;;) {
            const size_t n_alloc = keys_for(want_alloc);
            hipError_t err = hipSuccess;
            for (int i = 0; i < 2 && err == hipSuccess; i++) {
                const size_t bytes = n_alloc * (i == 0 ? 4 : 2) + 64;   // level-1 keys are 32-bit (24-bit in the direct form), final keys 16-bit
                err = lhgt::dev_malloc(&ctx->d_part_keys[i], bytes);
            }
            if (err == hipSuccess) { ctx->part_keys_cap = n_alloc; break; }
            (void)hipGetLastError();
            for (int i = 0; i < 2; i++) {
                if (ctx->d_part_keys[i]) lhgt::dev_free(ctx->d_part_keys[i]);
                ctx->d_part_keys[i] = nullptr;
            }
            if (err != hipErrorOutOfMemory || want_alloc <= (64L << 10))
                LHGT_FAIL(LHGT_E_HIP, "phase A: no room for the key buffers of %ld pairs (%zu keys): %s", want_alloc, n_alloc, hipGetErrorString(err));
            want_alloc = want_alloc > want ? want : want_alloc / 2;      // first drop the wish, then halve the chunk
            if (want_alloc < want) want = want_alloc;
        }
        need = keys_for(want);
    }
    const long chunk_pairs = want;
    uint32_t* d_list = nullptr;
    uint32_t list_cap = 0;
    if (have_long) {                      // room for every segment of the batch's long reads (<= 500 bases: four segments)
        const size_t n = (size_t)std::min<long>(b.n_long, 2 * b.d.n_pairs) * 4 + 64;
        if (n >= ((size_t)1 << 32)) LHGT_FAIL(LHGT_E_ARG, "batch of %ld long reads", b.n_long);
        list_cap = (uint32_t)n - 1;
        LHGT_HIP(lhgt::dev_malloc(&d_list, n * 4));
    }
    if (!ctx->d_part_meta) LHGT_HIP(lhgt::dev_malloc(&ctx->d_part_meta, (size_t)(2 * 65536 + NBK * 512) * 4));
    uint32_t* cur2 = ctx->d_part_meta;     // keys sent to each final bucket (direct form: to each half of its region)
    uint32_t* cur1 = cur2 + 2 * 65536;     // keys sent to each level-1 segment (direct form: to each (bucket, workgroup) piece)
    const int grid = 256 * 2;   // persistent-style grids: LDS admits two of these workgroups per CU
    for (long p0 = 0; p0 < b.d.n_pairs; p0 += chunk_pairs) {
        long np = b.d.n_pairs - p0 < chunk_pairs ? b.d.n_pairs - p0 : chunk_pairs;
        const PartCap pc = cap_of(np);
        if (direct_form) {
            // both behind the context's debug flags (lhgt_set_debug; ADVICE r4: the geometry was an environment variable read once into
            // a static -- untestable per context -- and the ablation switch, which makes tables WRONG, sat beside it):
            //   bit 21: the Small geometry (two 64 KiB workgroups per CU; the table is the same: test_direct_form_...)
            //   bit 22: stage ablation for timing, stages from LHGT_PART_ABLATE (bits 0-1: the read scatter, bits 4-5: the key scatter)
            const int ablate = (ctx->debug & (1 << 22)) && getenv("LHGT_PART_ABLATE") ? atoi(getenv("LHGT_PART_ABLATE")) : 0;
            const int geom = (ctx->debug & (1 << 21)) ? 0 : 1;
            const PartCap sc = seg_cap(pc.n);
            const size_t slice_bytes = (size_t)(((1 << g.slot_bits) + 15) >> 4) * 4;
            auto run = [&](auto G_) {
                using G = decltype(G_);
                const uint32_t piece = piece_keys(pc.n, G::GRID);
                // (a batch of long reads only -- 250-base reads throughout -- has nothing for the first launch but a walk over its lengths:
                // 2.5 ms per chunk, 72 ms per 100 M pairs; the pieces' cursors, which it leaves for the list launch, are cleared instead)
                if (have_long && b.n_long == 2 * b.d.n_pairs)
                    hipMemsetAsync(cur1, 0, (size_t)NBK * G::GRID * 4, ctx->stream);
                else
                    hipLaunchKernelGGL((part_reads_direct<G, false>), dim3(G::GRID), dim3(G::T), 0, ctx->stream, b.d, p0, np, ctx->hp, piece, cur1, ctx->d_part_keys[0],
                                       ctx->d_counts, ablate & 3, (const uint32_t*)nullptr);
                if (have_long) {          // the long reads of the chunk, segment by segment, behind the short ones in the same pieces
                    hipMemsetAsync(d_list, 0, 4, ctx->stream);
                    hipLaunchKernelGGL(long_read_segments, dim3((unsigned)((2 * np + 256 * LRS_ROUNDS - 1) / (256 * LRS_ROUNDS))), dim3(256), 0, ctx->stream, b.d, p0, np, ctx->k, d_list, list_cap);
                    hipLaunchKernelGGL((part_reads_direct<G, true>), dim3(G::GRID), dim3(G::T), 0, ctx->stream, b.d, p0, np, ctx->hp, piece, cur1, ctx->d_part_keys[0],
                                       ctx->d_counts, ablate & 3, (const uint32_t*)d_list);
                }
                // LHGT_PART_CG=8: round 4's 16-byte copy-out of the key scatter (A/B; the table is the same)
                const char* cg_env = getenv("LHGT_PART_CG");
                if (cg_env && atoi(cg_env) == 8)
                    hipLaunchKernelGGL((part_keys16_direct<G, 8>), dim3(G::HALVES * NBK), dim3(G::T), 0, ctx->stream, ctx->d_part_keys[0], cur1, piece, sc, cur2,
                                       (uint16_t*)ctx->d_part_keys[1], ctx->d_counts, (ablate >> 4) & 3);
                else
                    hipLaunchKernelGGL((part_keys16_direct<G, G::CG>), dim3(G::HALVES * NBK), dim3(G::T), 0, ctx->stream, ctx->d_part_keys[0], cur1, piece, sc, cur2,
                                       (uint16_t*)ctx->d_part_keys[1], ctx->d_counts, (ablate >> 4) & 3);
                if (ctx->stats_on && ctx->d_stats)    // keys that reached a final bucket's region (the few sent straight to the table are not in it)
                    hipLaunchKernelGGL(part_sum_cursors, dim3(1), dim3(256), 0, ctx->stream, cur2, G::HALVES * g.nb, ctx->d_stats);
                hipLaunchKernelGGL(part_apply2<G::HALVES>, dim3(g.nb), dim3(PA), slice_bytes, ctx->stream, (const uint16_t*)ctx->d_part_keys[1], cur2, g, sc,
                                   ctx->d_counts);
            };
            if (geom == 0) run(GeomSmall{});
            else run(GeomBig{});
            LHGT_HIP(hipGetLastError());
            continue;
        }
        LHGT_HIP(hipMemsetAsync(ctx->d_part_meta, 0, (size_t)(2 * 65536 + 256) * 4, ctx->stream));   // cur2, and cur1 behind its 2 x 65536 entries
        if (max_nk <= 128 && ctx->e <= 3) {
            int rpt = (int)(TILE_KEYS1 / ((long)max_nk * ctx->e));
            if (rpt > (PT1 / 64) * RW) rpt = (PT1 / 64) * RW;
            if (ctx->k == 32 && ctx->e == 3)
                hipLaunchKernelGGL((part_scatter_reads_reg<32, 3>), dim3(256), dim3(PT1), 0, ctx->stream, b.d, p0, np, ctx->hp, g, rpt, pc, cur1,
                                   ctx->d_part_keys[0], ctx->d_counts);
            else
                hipLaunchKernelGGL((part_scatter_reads_reg<0, 0>), dim3(256), dim3(PT1), 0, ctx->stream, b.d, p0, np, ctx->hp, g, rpt, pc, cur1,
                                   ctx->d_part_keys[0], ctx->d_counts);
        } else
            hipLaunchKernelGGL(part_scatter_reads, dim3(grid), dim3(PT), 0, ctx->stream, b.d, p0, np, ctx->hp, g, reads_per_tile, pc, cur1,
                               ctx->d_part_keys[0], ctx->d_counts);
        if (ctx->stats_on && ctx->d_stats)
            hipLaunchKernelGGL(part_sum_cursors, dim3(1), dim3(256), 0, ctx->stream, cur1, g.nb1, ctx->d_stats);
        const size_t slice_bytes = (size_t)(((1 << g.slot_bits) + 15) >> 4) * 4;
        if (g.b2 > 0) {
            if (ctx->k == 32)
                hipLaunchKernelGGL(part_scatter_keys16<32>, dim3(256), dim3(PK), 0, ctx->stream, ctx->d_part_keys[0], cur1, g, pc, cur2,
                                   (uint16_t*)ctx->d_part_keys[1], ctx->d_counts);
            else
                hipLaunchKernelGGL(part_scatter_keys16<0>, dim3(256), dim3(PK), 0, ctx->stream, ctx->d_part_keys[0], cur1, g, pc, cur2,
                                   (uint16_t*)ctx->d_part_keys[1], ctx->d_counts);
            hipLaunchKernelGGL((part_apply<true>), dim3(g.nb), dim3(PA), slice_bytes, ctx->stream, (const void*)ctx->d_part_keys[1], cur2, g, pc,
                               ctx->d_counts);
        } else   // without a second level the level-1 segments are the final buckets
            hipLaunchKernelGGL((part_apply<false>), dim3(g.nb), dim3(PA), slice_bytes, ctx->stream, (const void*)ctx->d_part_keys[0], cur1, g, pc,
                               ctx->d_counts);
        LHGT_HIP(hipGetLastError());
    }
    if (d_list) { LHGT_HIP(hipStreamSynchronize(ctx->stream)); lhgt::dev_free(d_list); }
    return LHGT_OK;
}
