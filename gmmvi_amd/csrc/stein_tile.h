// Tiling of the Stein moment contraction (stein.hip: f32 matrix-core kernel; stein_split.hip: bf16 split-operand kernel)
#pragma once
#include "common.h"

// A wave-private LDS image: ordering its writes against the same wave's later reads only needs the wave's own LDS queue
// drained (and the compiler kept from moving the accesses).  A workgroup-scope fence would also wait for every outstanding
// GLOBAL access -- the prefetched rows of the next chunk -- ~1 us each time.
#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)


constexpr int SM_RS = 72;          // LDS row stride (floats) of the transposed images: 64 samples + 8
constexpr int SM_NBMAX = 5;        // most components stacked in one tile row

// Tiling of the padded dimension DP (covers D in (previous DP, DP]): MT row tiles for the D + 1 rows of [g; 1], NB components
// stacked along NT column tiles -- the (NT, NB) with the fewest padded columns among NT <= NTMAX (accumulators: 4 MT NT
// registers), ties to the smaller tile.
template <int DP>
struct SteinTile {
    static constexpr int D1 = DP + 1 < 64 ? DP + 1 : 64;           // D <= 63
    static constexpr int MT = (D1 + 15) / 16;
    static constexpr int NTMAX = MT <= 2 ? 6 : (MT == 3 ? 6 : 7);
    static constexpr int pick_nt() {
        int best = 1;
        long best_num = 0, best_den = 1;                        // efficiency best_num / best_den
        for (int nt = 1; nt <= NTMAX; ++nt) {
            int nb = (16 * nt) / D1;
            if (nb > SM_NBMAX) nb = SM_NBMAX;
            if (nb < 1) continue;
            const long num = (long)nb * D1, den = 16L * nt;
            if (num * best_den > best_num * den) { best = nt; best_num = num; best_den = den; }
        }
        return best;
    }
    static constexpr int NT = pick_nt();
    static constexpr int NB = (16 * NT) / D1 < SM_NBMAX ? (16 * NT) / D1 : SM_NBMAX;
    static constexpr int PREV = DP == 2 ? 0 : DP == 4 ? 2 : DP == 8 ? 4 : DP == 10 ? 8 : DP == 12 ? 10 : DP == 16 ? 12 : DP == 20 ? 16
                                : DP == 24 ? 20 : DP == 32 ? 24 : DP == 40 ? 32 : DP == 50 ? 40 : 50;     // D > PREV
};


