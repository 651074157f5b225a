// k_vote_judge.hpp -- judge_base / check_split on a pair's event list (shared by k_vote.hip and k_vote_shared.hip)
#pragma once
#include "lhgt_common.hpp"

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

}  // namespace lhgt
