// lhgt_hash.hpp -- the e position-dependent 1-bit-per-base hashes, O(1) per k-mer.
//
// Reference: for hash i the k-mer s hashes to min(fwd, rc) with
//   fwd = sum_z proj[cc[z][i]](s[z]) << (k-1-z),   rc = sum_z proj[cc[k-1-z][i]](comp(s[z])) << z
// (/root/reference/src/extract_ref_normal_peak.cpp:1058-1081, maps :1109-1180).  The three
// projections are bit-planes of the 2-bit code (A=0 C=1 G=2 T=3, hi/lo bits):
//   map0 (A,T->1) = ~(hi^lo), map1 (A,C->1) = ~hi, map2 (A,G->1) = ~lo,
// and the complemented base has both bits flipped, so the rc word is the same formula on the
// bit-reversed window with hi/lo un-negated.  mask[i][m] selects the positions where hash i
// uses map m (verified there against index bytes).
#pragma once
#include "lhgt_common.hpp"

namespace lhgt {

// k-bit window starting at base j of a plane (32 bases per word, first base at the MSB).
// v_alignbit_b32 takes the 32 bits that start r bits into the word pair in one full-rate instruction (a 64-bit shift runs at a quarter)
// The r = 0 case is picked with a bit-select on a lane mask, not with `r ? ... : hi`: behind a condition the compiler sinks the load
// of `lo` into an exec region of its own -- one region and one wait per window, twelve in a row per read pair in the vote kernels.
__device__ __forceinline__ uint32_t window32(uint32_t hi, uint32_t lo, int r) {
    return __builtin_amdgcn_bitop3_b32(r ? ~0u : 0u, __builtin_amdgcn_alignbit(hi, lo, 32 - r), hi, 0xCA);   // m ? a : b
}
__device__ __forceinline__ uint32_t plane_window(const uint32_t* __restrict__ w, int j, int k) {
    int q = j >> 5, r = j & 31;
    return window32(w[q], w[q + 1], r) >> (32 - k);
}

__device__ __forceinline__ uint32_t brev_k(uint32_t x, int k) { return __brev(x) >> (32 - k); }

// The three masks of a hash PARTITION the k low bits -- every position uses exactly one projection (build_hash_params refuses any
// other coder entry) -- so "(p0 & m0) | (p1 & m1) | (p2 & m2)" is two bit-selects (m ? a : b) instead of three ANDs and
// two ORs, and the forward word's three complements become one XOR with the k-bit mask at the end (the windows have no bit above k):
// 6 vector instructions per hash instead of 14 in the round-2 form.  The masks are kernel arguments, i.e. scalar registers.
// (written as (a & m) | (b & ~m) the compiler makes an AND and an AND-OR of it, with ~m kept in a scalar register: the gfx950
// three-input bit operation does it in one; truth table 0xCA = m ? a : b)
__device__ __forceinline__ uint32_t bit_select(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
__device__ __forceinline__ uint32_t hash_from_windows(uint32_t whi, uint32_t wlo, uint32_t rhi, uint32_t rlo,
                                                      const uint32_t* __restrict__ m) {
    const uint32_t kmask = m[0] | m[1] | m[2];
    const uint32_t fwd = bit_select(m[0], whi ^ wlo, bit_select(m[1], whi, wlo)) ^ kmask;        // = (~(whi^wlo) & m0) | (~whi & m1) | (~wlo & m2)
    const uint32_t rc = bit_select(m[0], ~(rhi ^ rlo), bit_select(m[1], rhi, rlo));              // = (~(rhi^rlo) & m0) | (rhi & m1) | (rlo & m2)
    return fwd < rc ? fwd : rc;
}

// Where phase B takes the e hashes of reference position (contig c, offset j) from.  Two resident forms of the reference:
//   index  : the index file's own layout, [u32 len][(len-k+1)*e u32] per contig (E:785-813) -- 4e bytes per base, read as stored;
//   packed : three bit-planes (hi bit, lo bit, not-a-base) over the flat positions of the indexed contigs, the first two
//            interleaved word by word -- 3/8 byte per base,
//            the hashes recomputed where they are needed (156 GB -> 4.9 GB for a 13 Gbase catalogue).
//            An invalid k-mer hashes to 0 exactly as the index stores it (E:808-810, quirk Q6).
struct RefSource {
    const uint32_t* index;    // non-null: index form
    const uint32_t* planes;   // packed form: word x >> 5 of the hi / lo plane is planes[2 * (x >> 5) + 0 / 1], of the not-a-base plane planes[2 * plane_words + (x >> 5)]; position x at bit 31 - (x & 31)
    uint64_t plane_words;
    HashParams hp;
};
struct RefKmer {
    const uint32_t* stored;   // index form: the e hashes
    uint32_t whi, wlo, rhi, rlo;
    bool valid;
};
__device__ __forceinline__ RefKmer ref_kmer(const RefSource& rs, const ContigDev& c, long j, int k, int e) {
    RefKmer km{};
    if (rs.index) {
        km.stored = rs.index + c.hash_word + j * e;
    } else {
        const uint64_t x = c.flat_base + (uint64_t)j;
        const uint32_t* w = rs.planes + 2 * (x >> 5);
        const uint32_t* nb = rs.planes + 2 * rs.plane_words + (x >> 5);
        const int r = (int)(x & 31);
        km.whi = window32(w[0], w[2], r) >> (32 - k);
        km.wlo = window32(w[1], w[3], r) >> (32 - k);
        km.valid = (window32(nb[0], nb[1], r) >> (32 - k)) == 0u;
        km.rhi = brev_k(km.whi, k);
        km.rlo = brev_k(km.wlo, k);
    }
    return km;
}
__device__ __forceinline__ uint32_t ref_hash(const RefSource& rs, const RefKmer& km, int i) {
    if (rs.index) return km.stored[i];
    return km.valid ? hash_from_windows(km.whi, km.wlo, km.rhi, km.rlo, rs.hp.mask[i]) : 0u;
}

// ASCII -> 2-bit code, 4 = not a base (E:1112-1151: upper and lower case ACGT only)
__host__ __device__ __forceinline__ uint32_t base_code(uint8_t c) {
    switch (c | 0x20) {
        case 'a': return 0;
        case 'c': return 1;
        case 'g': return 2;
        case 't': return 3;
        default: return 4;
    }
}

// Vote prefilter (k_scan.hip registers, k_vote.hip tests): word (h & pf_mask) >> 5 of the bitmap, bit h & 31 in it.  When the
// bitmap is a fold of the table (pf2 != 0: fewer address bits than k) a key also sets a second bit of the same word, chosen by
// the address bits above the fold -- a one-load blocked Bloom filter: 2.3 M keys in 2^25 bits pass 1.6 % of foreign probes
// instead of 6.9 %.  A clear bit is still an exact negative.
// Word of the bitmap that key h sets / tests.  Bit 31 of the mask argument flags the three-quarter bitmap (3 MiB of the 4 MiB a 25-bit
// mask spans: what an XCD's 4 MiB L2 can keep next to the streams that pass through it): the 2^20 word numbers fold onto 3 * 2^18.
constexpr uint32_t PF_Q3 = 0x80000000u;
// Round 4, late: in the three-quarter form word and bits come from pf_mix(h) = h * an odd constant.  (1) The fold of 2^20 word numbers
// onto 3 * 2^18 by (w * 3) >> 2 sent two of every four source words to one target: half of all probes met words with twice the keys,
// and those words made most of the false positives; the product's HIGH word over the mixed key spreads the keys evenly over the
// 786 432 words.  (2) Three bits per key instead of two (11 bits per key on configs[2]: 1.3 % instead of 2.8 % of foreign probes pass),
// taken from the low twelve bits of the mixed key, which the word does not depend on.
constexpr uint32_t PF_Q3_WORDS = 3u << 18;
constexpr int PF_MIXED = 0x100;           // flag in pf2: the bits of a key in the three-quarter form (below)
__device__ __forceinline__ uint32_t pf_mix(uint32_t h) { return h * 0x9E3779B1u; }
template <bool Q3>
__device__ __forceinline__ uint32_t pf_word_t(uint32_t h, uint32_t pf_mask) {
    if (Q3) return __umulhi(pf_mix(h), PF_Q3_WORDS);
    return (h & pf_mask & ~PF_Q3) >> 5;
}
__device__ __forceinline__ uint32_t pf_word(uint32_t h, uint32_t pf_mask) {       // flag looked at at run time (the kernels that are not hot)
    return (pf_mask & PF_Q3) ? __umulhi(pf_mix(h), PF_Q3_WORDS) : (h & pf_mask & ~PF_Q3) >> 5;
}
// the word takes the mixed key's bits from ~12 up: a key's three positions come from the twelve below -- bits 0..4, 5..9, and a product
// hash of all twelve (it need not be independent of the key's own other two positions, only look random next to the bits OTHER keys
// set; (m >> 10) & 31 was tried: bits 12..14 are the same for all keys of a word, the third position then adds fill and screens nothing)
__device__ __forceinline__ uint32_t pf_third_of(uint32_t m) { return (__umul24(m & 0xfffu, 0x9E37u) >> 11) & 31u; }
__device__ __forceinline__ uint32_t pf_word_bits(uint32_t h, int pf2) {
    if (pf2 & PF_MIXED) {
        const uint32_t m = pf_mix(h);
        return (1u << (m & 31u)) | (1u << ((m >> 5) & 31u)) | (1u << pf_third_of(m));
    }
    return (1u << (h & 31u)) | (pf2 ? 1u << ((h >> pf2) & 31u) : 0u);
}
// (both bits shifted down and ANDed: a variable shift takes its count modulo 32 by itself, so this is four vector instructions
// where building the two-bit mask and comparing took eight; with pf2 = 0 the second shift picks the first bit again)
__device__ __forceinline__ bool pf_pass(uint32_t word, uint32_t h, int pf2) {
    if (pf2 & PF_MIXED) {
        const uint32_t m = pf_mix(h);
        return ((word >> (m & 31u)) & (word >> ((m >> 5) & 31u)) & (word >> pf_third_of(m)) & 1u) != 0u;
    }
    return ((word >> (h & 31u)) & (word >> ((h >> pf2) & 31u)) & 1u) != 0u;
}

}  // namespace lhgt
