// march_kernels.hip — kernel 2 of the partitioned (long-HRIR) path: the marched CMAC of tile_march.hpp.
// Its own translation unit because it is built with -fno-slp-vectorize (airwave_amd/build.py): hipcc's SLP pass packs
// the complex multiply-accumulates into v_pk_fma_f32 — no faster per FMA on gfx950 — and in doing so lifts the kernel from
// 132 to 240 VGPRs (3 -> 2 waves per SIMD); tools/ubench/one_march.hip shows both allocations in seconds.
#include "kernels.hpp"

namespace awk {

// Kernel 2 of the partitioned path (tile_march.hpp): grid = (slots * LG / 256, streams).  LG lanes (channel pairs)
// per bin-pair slot, summed across lanes (DPP) before the store; q0 > 0 passes accumulate into W.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LG>
__device__ __forceinline__ float lane_group_sum(float v) {
    if constexpr (LG >= 2) v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
    if constexpr (LG >= 4) v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
    if constexpr (LG >= 8) v = dpp_add<0x141>(v);      // row_half_mirror: lane i <-> 7 - i of each 8
    return v;
}
__device__ __forceinline__ void march_st(cf *p, cf v) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f x = {v.x, v.y};
    __builtin_nontemporal_store(x, reinterpret_cast<v2f *>(p));
}

#ifndef AW_MARCH_STREAMS
#define AW_MARCH_STREAMS 4      // streams a workgroup walks with one table load (each stream re-reads 4 MB of tables otherwise)
#endif
#ifndef AW_MARCH_MIN_WAVES
#define AW_MARCH_MIN_WAVES 1    // waves per SIMD the register allocation must allow (4 = 128 VGPRs, with 24 spilled dwords)
#endif
template <int PQ, int LG, bool ACC>
__global__ void __launch_bounds__(kMarchThreads, AW_MARCH_MIN_WAVES) aw_part_march_kernel(TileParams p, int q0, int n_streams) {
    const int g = (int)(blockIdx.x * kMarchThreads + threadIdx.x);
    const int j_raw = g / LG, pl = g % LG;
    const bool real = j_raw < kMarchSlots;               // the last workgroup's spare lanes repeat the last slot: whole lane groups that compute and (first pass) store the same values to the same addresses as the real one
    const int j = real ? j_raw : kMarchSlots - 1;
    const long long s0 = (long long)blockIdx.y * AW_MARCH_STREAMS;
    const long long s1 = s0 + AW_MARCH_STREAMS < n_streams ? s0 + AW_MARCH_STREAMS : n_streams;
    // After the lane-group sum every lane of a group holds the same two sums.  First pass (no accumulate): EVERY lane
    // stores — even lanes W[i], odd lanes W[pi] (duplicates write equal values to equal addresses), so the steady state
    // has one unpredicated store and no branch.  Accumulating passes keep one loader/storer per bin.
    const bool odd = (pl & 1) != 0;
    const int woff = (LG > 1 && odd) ? march_bins(j).pi : march_bins(j).i;
    march_thread<PQ, (LG < 8)>(p, s0, s1, j, pl, q0, [&](long long stream, int b, const MarchBins &mb, cf ai, cf ap) {
        ai.x = lane_group_sum<LG>(ai.x); ai.y = lane_group_sum<LG>(ai.y);
        ap.x = lane_group_sum<LG>(ap.x); ap.y = lane_group_sum<LG>(ap.y);
        cf *w = p.wspec + (stream * p.n_blocks + b) * (long long)kN;
#ifdef AW_ABL_MARCH_NOSTORE     // timing ablation only (wrong results)
        if (ai.x != 1.2345e-30f) return;
#endif
        if constexpr (!ACC) {
            // streaming stores: W is read by the inverse kernel only
            if constexpr (LG == 1) { march_st(w + mb.i, ai); march_st(w + mb.pi, ap); }
            else march_st(w + woff, odd ? ap : ai);
        } else {
            if (real && pl == 0) w[mb.i] = ai + w[mb.i];
            if (real && pl == (LG > 1 ? 1 : 0) && mb.pi != mb.i) w[mb.pi] = ap + w[mb.pi];
        }
    });
}

// Marched CMAC: passes of at most 8 partitions; lane groups of 1, 2, 4 or 8 channel pairs.
template <int PQ, bool ACC>
static void launch_march_pq(const TileParams &p, int n_streams, int q0, hipStream_t stream) {
    const int lg = p.n_pairs <= 1 ? 1 : p.n_pairs <= 2 ? 2 : p.n_pairs <= 4 ? 4 : 8;
    const dim3 grid((unsigned)(((long long)kMarchSlots * lg + kMarchThreads - 1) / kMarchThreads),
                    (unsigned)((n_streams + AW_MARCH_STREAMS - 1) / AW_MARCH_STREAMS)), block(kMarchThreads);
    switch (lg) {
        case 1: hipLaunchKernelGGL((aw_part_march_kernel<PQ, 1, ACC>), grid, block, 0, stream, p, q0, n_streams); break;
        case 2: hipLaunchKernelGGL((aw_part_march_kernel<PQ, 2, ACC>), grid, block, 0, stream, p, q0, n_streams); break;
        case 4: hipLaunchKernelGGL((aw_part_march_kernel<PQ, 4, ACC>), grid, block, 0, stream, p, q0, n_streams); break;
        default: hipLaunchKernelGGL((aw_part_march_kernel<PQ, 8, ACC>), grid, block, 0, stream, p, q0, n_streams); break;
    }
}

hipError_t launch_part_march(const TileParams &p, int n_streams, hipStream_t stream, StageTimer *tm) {
    if (n_streams <= 0 || p.n_blocks <= 0) return hipSuccess;
    if (p.n_pairs > 8) return hipErrorInvalidValue;
    if (tm) tm->begin();
    for (int q0 = 0; q0 < p.partitions; q0 += 8) {
        const bool small = p.partitions - q0 <= 4;
        if (q0 == 0) { if (small) launch_march_pq<4, false>(p, n_streams, q0, stream); else launch_march_pq<8, false>(p, n_streams, q0, stream); }
        else { if (small) launch_march_pq<4, true>(p, n_streams, q0, stream); else launch_march_pq<8, true>(p, n_streams, q0, stream); }
    }
    if (tm) tm->end("aw_part_march_kernel");
    return hipGetLastError();
}

}  // namespace awk
