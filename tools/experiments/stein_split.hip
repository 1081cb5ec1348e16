// NOT part of the library (not in csrc/Makefile): the split-operand (bf16 matrix-core) form of the Stein moment contraction as it
// was measured in round 4 -- profiles/r04_notes.md "Stein moments on the bf16 matrix cores (not kept)".  It builds when placed in
// gmmvi_amd/csrc next to stein_tile.h with the launch hook described there.
// Stein moment contraction on the bf16 matrix cores (stein.hip holds the f32 kernel, the finalisation and the entry points).
#include "common.h"
#include "bf16_split.h"
#include "wave_reduce.h"
#include "stein_tile.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wave_max(float v) { return gmmvi_wave_max(v); }

// =====================================================================================================================
// The same contraction on the bf16 matrix cores: v_mfma_f32_16x16x32_bf16 through three-way split operands (bf16_split.h: six
// partial products per tile and 32 samples, 6 x 16 cycles against 8 x 32 for the f32 instruction -- 2.7 x the matrix-core rate at
// the accuracy of an f32 contraction).  Same grid, same wave ranges, same partial slab as stein_moment_kernel; what differs:
//   * a chunk is 32 samples (one k-step); lane = (sample s = lane & 31, feature half h = lane >> 5): the lane loads its half of
//     the x / grad rows, forms g = grad log p~ - grad log q and, for each of the NB stacked components, (x - mu) e and e, and
//     splits every value ONCE into three bf16 -- (D + 1)(NB + 1) values per sample instead of a split behind every fragment read;
//   * the wave-private images are SAMPLE-major, [plane][sample][row]: the lane's values of one sample are contiguous, a split
//     pair of features is one ds_write_b32 per plane; the operand fragments (8 consecutive samples of one row) come out of two
//     ds_read_b64_tr_b16 (each hands lane i of a 16-lane group column i of a 4-sample x 16-row block).  Rows 0 .. D of an image
//     row are [g; 1], rows RA .. hold the stacked components: (x - mu_c) e_c at RA + c D + j, the e_c themselves (the "1"
//     column of [x - mu; 1]) behind them at RA + NB D + c, so that feature pairs stay 4-byte aligned for every c.  Row stride
//     != 0 mod 128 bytes: the four sample rows of a transposed read fall into disjoint banks;
//   * the MFMAs of chunk t run from fragments held in registers while the vector unit stages chunk t + 1 (same basic block; an
//     MFMA occupies the issue port for 8 of its 16 cycles); the rows of chunk t + 2 are requested as soon as chunk t + 1 has
//     left its registers, in FRONT of chunk t's MFMAs, which then cover the trip to the L2.  This file is compiled without
//     SLP vectorisation: packed f32 instructions beside MFMAs cost more than the two plain ones they replace;
//   * which sample sits in which k-slot is the same for both operands, which is all the contraction needs.
// Fast instances only (D == DP, rows 8-byte aligned, weights from ld - bg); everything else runs the f32 kernel above.
// =====================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int sm_i32x4 __attribute__((ext_vector_type(4)));
typedef short sm_s16x4 __attribute__((ext_vector_type(4)));

template <int DP>
struct SteinSplit {
    using ST = SteinTile<DP>;
    static constexpr int F0 = ((DP / 2 + 1) / 2) * 2;                // features of the lower lane half (even); the upper half: DP - F0 <= F0
    static constexpr int F1 = DP - F0;
    static constexpr int RA = 16 * ST::MT, RB = 16 * ST::NT;
    // row stride (bf16 elements) of a sample's row: 8 bytes x an odd number -- the 32 lanes of a lane half write their pairs
    // (4 bytes at that stride) into 32 different banks -- such that the four sample rows of a transposed read (32 bytes each)
    // fall into disjoint banks
    static constexpr bool stride_ok(int rs) {
        if (rs % 8 != 4) return false;
        int o[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) o[i] = (i * 2 * rs) % 256;
        for (int i = 0; i < 4; ++i)
            for (int j = i + 1; j < 4; ++j) {
                int d = o[i] > o[j] ? o[i] - o[j] : o[j] - o[i];
                if (d > 128) d = 256 - d;
                if (d < 32) return false;
            }
        return true;
    }
    static constexpr int pick_rs() {
        int rs = RA + RB;
        while (!stride_ok(rs)) ++rs;
        return rs;
    }
    static constexpr int RS = pick_rs();
    static constexpr int PLANE = 32 * RS;                                  // elements of one plane
    static constexpr int WAVE_ELEMS = 3 * PLANE;
    static constexpr size_t lds_bytes() {
        size_t b = (size_t)4 * WAVE_ELEMS * 2;
        const size_t c = (size_t)16 * ST::MT * (16 * ST::NT + 1) * 4;         // the merge tile
        return b > c ? b : c;
    }
};

// the three planes of a split pair -> rows (r, r + 1) of one sample (p: plane 0, 4-byte aligned)
__device__ __forceinline__ void sm_put_pair(unsigned short* p, int plane, float a, float b) {
    uint32_t p1, p2, p3;
    split_pair(a, b, p1, p2, p3);
    *reinterpret_cast<uint32_t*>(p) = p1;
    *reinterpret_cast<uint32_t*>(p + plane) = p2;
    *reinterpret_cast<uint32_t*>(p + 2 * plane) = p3;
}
// 8 consecutive samples of one row of a sample-major image -> an MFMA operand (p: this lane's address for samples k .. k + 3)
__device__ __forceinline__ bf16x8 sm_tr_frag(const unsigned short* p, int rs) {
    typedef sm_s16x4 __attribute__((address_space(3))) * lds_ptr;
    const sm_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const sm_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * rs));
    sm_i32x4 r;
    r[0] = __builtin_bit_cast(int2, lo).x; r[1] = __builtin_bit_cast(int2, lo).y;
    r[2] = __builtin_bit_cast(int2, hi).x; r[3] = __builtin_bit_cast(int2, hi).y;
    return __builtin_bit_cast(bf16x8, r);
}

template <int DP>
__global__ __launch_bounds__(256, (SteinTile<DP>::MT <= 1 ? 2 : 1))
void stein_moment_bf16_kernel(int K, int N, int wave_range, int stacks, int R, const float* __restrict__ packed,
                              const float* __restrict__ X, const float* __restrict__ TG, const float* __restrict__ QG,
                              const float* __restrict__ ld, const float* __restrict__ bg, float* __restrict__ part,
                              float* __restrict__ part_m, int mode) {
    using ST = SteinTile<DP>;
    using SS = SteinSplit<DP>;
    constexpr int MT = ST::MT, NT = ST::NT, NB = ST::NB, D = DP, D1 = DP + 1;
    constexpr int F0 = SS::F0, F1 = SS::F1, RA = SS::RA, RS = SS::RS, PL = SS::PLANE;
    constexpr bool EVEN = F0 == F1;                       // both lane halves carry the same number of features
    extern __shared__ float sm[];
    __shared__ float sm_m[4][NB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per_xcd = (stacks * R + 7) / 8;             // item placement: see stein_moment_kernel
    const int item = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (item >= stacks * R) return;
    const int range_id = item / stacks;
    const int k0 = (item - range_id * stacks) * NB;
    const int nb = min(NB, K - k0);
    unsigned short* T = reinterpret_cast<unsigned short*>(sm) + (size_t)wave * SS::WAVE_ELEMS;       // [3][32 samples][RS]
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(T);
        for (int e = lane; e < SS::WAVE_ELEMS / 2; e += 64) z[e] = 0u;
        WAVE_LDS_SYNC();
        if (lane < 32) T[lane * RS + D] = 0x3F80;           // [g; 1]: bf16 1.0 (its two lower planes stay zero)
    }
    const int s = lane & 31, h = lane >> 5;
    const int fbase = h ? F0 : 0, fcnt = h ? F1 : F0;
    const int q = lane >> 4, c16 = lane & 15;
    int cidx[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + c16;
        const int comp = col < NB * D ? col / D : col - NB * D;
        cidx[nt] = comp < nb ? comp : NB;
    }
    // transposed fragment reads: lane 4 g + p of a 16-lane group addresses sample 8 q + g, rows 4 p .. 4 p + 3
    const unsigned short* fr = T + (8 * q + ((lane >> 2) & 3)) * RS + 4 * (lane & 3);
    float mu[NB][F0];
#pragma unroll
    for (int c = 0; c < NB; ++c)
#pragma unroll
        for (int i = 0; i < F0; ++i)
            mu[c][i] = c < nb ? packed[(size_t)(k0 + c) * Pack<DP>::STRIDE + fbase + ((EVEN || i < fcnt) ? i : (i & 1))] : 0.f;
    const int w_begin = min(N, (range_id * 4 + wave) * wave_range);
    const int w_end = min(N, w_begin + wave_range);
    float c_lim[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) c_lim[c] = c < nb ? 3.0e38f : -3.0e38f;
    auto log_weights = [&](int n_lane, float (&a)[NB]) {           // (clamped loads, weight "-inf" beyond the range: see above)
        const int n = min(n_lane, N - 1);
        const float lim = (n_lane < w_end) ? 3.0e38f : -3.0e38f;
        const float bgv = bg[n];
#pragma unroll
        for (int c = 0; c < NB; ++c) a[c] = fminf(ld[(size_t)min(k0 + c, K - 1) * N + n] - bgv, fminf(lim, c_lim[c]));
    };
    float M[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) M[c] = -3.0e38f;
    for (int n0 = w_begin; n0 < w_end; n0 += 64) {
        float a[NB];
        log_weights(n0 + lane, a);
#pragma unroll
        for (int c = 0; c < NB; ++c) M[c] = fmaxf(M[c], a[c]);
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) M[c] = wave_max(M[c]);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
    struct Rows { float a[NB], x[F0], t[F0], q[F0]; };
    auto fetch = [&](Rows& r, int n0) {
        const size_t row = (size_t)min(n0 + s, N - 1) * D + fbase;
        log_weights(n0 + s, r.a);
#pragma unroll
        for (int i = 0; i < F0; i += 2) {
            const int o = (EVEN || i < fcnt) ? i : 0;              // the shorter half re-reads its first pair (never stored)
            const float2 vx = *reinterpret_cast<const float2*>(X + row + o);
            const float2 vt = *reinterpret_cast<const float2*>(TG + row + o);
            const float2 vq = *reinterpret_cast<const float2*>(QG + row + o);
            r.x[i] = vx.x; r.x[i + 1] = vx.y; r.t[i] = vt.x; r.t[i + 1] = vt.y; r.q[i] = vq.x; r.q[i + 1] = vq.y;
        }
    };
    unsigned short* wa = T + s * RS + fbase;                 // this lane's sample row: its first [g; 1] row ...
    unsigned short* wb = wa + RA;                            // ... and its first row of component 0
    unsigned short* we = T + s * RS + RA + NB * D;           // the e rows of its sample
    auto stage = [&](const Rows& r) {
        float e[NB + 1];
#pragma unroll
        for (int c = 0; c < NB; ++c) e[c] = (r.a[c] > -1.0e38f) ? __expf(r.a[c] - M[c]) : 0.f;
        e[NB] = 0.f;
        // no branches in here (the MFMAs of the previous chunk are scheduled into this code): the shorter lane half stages its
        // first pair again instead of skipping (same values to the same address), both halves write the e rows
#pragma unroll
        for (int i = 0; i < F0; i += 2) {
            const int o = (EVEN || i < fcnt) ? i : 0;
            sm_put_pair(wa + o, PL, r.t[i] - r.q[i], r.t[i + 1] - r.q[i + 1]);                  // g = grad log p~ - grad log q (:248)
#pragma unroll
            for (int c = 0; c < NB; ++c)
                sm_put_pair(wb + c * D + o, PL, (r.x[i] - mu[c][i]) * e[c], (r.x[i + 1] - mu[c][i + 1]) * e[c]);
        }
#pragma unroll
        for (int c = 0; c < NB; c += 2) sm_put_pair(we + c, PL, e[c], e[c + 1]);
    };
    bf16x8 af[MT][3], bf[NT][3];
    auto read_frags = [&]() {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[mt][p] = sm_tr_frag(fr + p * PL + 16 * mt, RS);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt][p] = sm_tr_frag(fr + p * PL + RA + 16 * nt, RS);
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {                      // smallest partial products first
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][2], bf[nt][0], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][0], bf[nt][2], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][1], bf[nt][1], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][1], bf[nt][0], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][0], bf[nt][1], acc[mt][nt], 0, 0, 0);
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt][0], bf[nt][0], acc[mt][nt], 0, 0, 0);
            }
    };
    // chunk t's MFMAs beside chunk t + 1's staging.  The chunk staged last lies beyond the range: clamped rows, weights 0, never
    // multiplied.
    if (w_begin < w_end) {
        Rows r0;
        fetch(r0, w_begin);
        stage(r0);
        fetch(r0, w_begin + 32);
        WAVE_LDS_SYNC();
        read_frags();
        WAVE_LDS_SYNC();
        for (int n0 = w_begin; n0 < w_end; n0 += 32) {
            if (!(mode & 1)) stage(r0);
            if (!(mode & 4)) fetch(r0, n0 + 64);
            if (!(mode & 2)) multiply();
            WAVE_LDS_SYNC();
            if (!(mode & 8)) read_frags();
            WAVE_LDS_SYNC();
        }
    }

    // ---- merge the four waves: as in stein_moment_kernel --------------------------------------------------------------
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NB; ++c) sm_m[wave][c] = M[c];
    }
    __syncthreads();
    float Mall[NB], fsc[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        Mall[c] = fmaxf(fmaxf(sm_m[0][c], sm_m[1][c]), fmaxf(sm_m[2][c], sm_m[3][c]));
        fsc[c] = __expf(M[c] - Mall[c]);
    }
    constexpr int CW = 16 * NT + 1;
    float* C = sm;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float fs = 1.f;
#pragma unroll
                for (int c = 0; c < NB; ++c) fs = (cidx[nt] == c) ? fsc[c] : fs;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* dst = C + (16 * mt + 4 * q + r) * CW + 16 * nt + c16;
                        const float v = acc[mt][nt][r] * fs;
                        *dst = (w == 0) ? v : *dst + v;
                    }
            }
        }
        __syncthreads();
    }
    constexpr int DD = D1 * D1;
    for (int el = tid; el < nb * DD; el += 256) {
        const int comp = el / DD, rem = el - comp * DD;
        const int i = rem / D1, j = rem - i * D1;
        part[((size_t)(k0 + comp) * R + range_id) * DD + rem] = C[i * CW + (j < D ? comp * D + j : NB * D + comp)];
    }
    if (tid < nb) part_m[(size_t)(k0 + tid) * R + range_id] = Mall[tid];
}


template <int DP>
static int launch_split(gmmvi_ctx* ctx, int K, int N, int wave_range, int stacks, int R, const float* packed, const float* X,
                        const float* tgrad, const float* qgrad, const float* ld, const float* bg, float* part, float* part_m) {
    const size_t shmem = SteinSplit<DP>::lds_bytes();
    static bool attr_done = false;
    if (!attr_done && shmem > 64 * 1024) {
        GMMVI_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)stein_moment_bf16_kernel<DP>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        attr_done = true;
    }
    const int per_xcd = (stacks * R + 7) / 8;
    hipLaunchKernelGGL((stein_moment_bf16_kernel<DP>), dim3(8 * per_xcd), dim3(256), shmem, ctx->stream, K, N, wave_range, stacks, R,
                       packed, X, tgrad, qgrad, ld, bg, part, part_m, getenv("GMMVI_STEIN_SPLIT_MODE") ? atoi(getenv("GMMVI_STEIN_SPLIT_MODE")) : 0);
    return GMMVI_OK;
}

bool gmmvi_stein_split_built(int dp) { return dp == 10 || dp == 20 || dp == 32 || dp == 40 || dp == 50; }

int gmmvi_stein_split_launch(gmmvi_ctx* ctx, int dp, int K, int N, int wave_range, int stacks, int R, const float* packed,
                             const float* X, const float* tgrad, const float* qgrad, const float* ld, const float* bg,
                             float* part, float* part_m) {
    switch (dp) {
#define GMMVI_SS(DPV) case DPV: return launch_split<DPV>(ctx, K, N, wave_range, stacks, R, packed, X, tgrad, qgrad, ld, bg, part, part_m)
        GMMVI_SS(10); GMMVI_SS(20); GMMVI_SS(32); GMMVI_SS(40); GMMVI_SS(50);
#undef GMMVI_SS
        default: return gmmvi_fail(ctx, GMMVI_ERR_ARG, "stein split kernel: dimension not built");
    }
}
