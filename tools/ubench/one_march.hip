// Probe build: only the marched CMAC kernel of the partitioned path, to read its register allocation in seconds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -Iairwave_amd/csrc -Iinclude -S --cuda-device-only -o /tmp/march.s tools/ubench/one_march.hip
#include "device/tile_march.hpp"
#ifndef ONE_PQ
#define ONE_PQ 8
#endif
#ifndef ONE_LG
#define ONE_LG 4
#endif
namespace awk {
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LG>
__device__ __forceinline__ float lane_group_sum(float v) {
    if constexpr (LG >= 2) v = dpp_add<0xB1>(v);
    if constexpr (LG >= 4) v = dpp_add<0x4E>(v);
    if constexpr (LG >= 8) v = dpp_add<0x141>(v);
    return v;
}
__global__ void __launch_bounds__(kMarchThreads) one_march(TileParams p, int q0) {
    constexpr int LG = ONE_LG;
    const int g = (int)(blockIdx.x * kMarchThreads + threadIdx.x);
    const int j_raw = g / LG, pl = g % LG;
    const int j = j_raw < kMarchSlots ? j_raw : kMarchSlots - 1;
    const long long stream = blockIdx.y * 4;
    const bool odd = (pl & 1) != 0;
    const int woff = (LG > 1 && odd) ? march_bins(j).pi : march_bins(j).i;
    march_thread<ONE_PQ, (ONE_LG < 8)>(p, stream, stream + 4, j, pl, q0, [&](long long st, int b, const MarchBins &mb, cf ai, cf ap) {
        ai.x = lane_group_sum<LG>(ai.x); ai.y = lane_group_sum<LG>(ai.y);
        ap.x = lane_group_sum<LG>(ap.x); ap.y = lane_group_sum<LG>(ap.y);
        cf *w = p.wspec + (st * p.n_blocks + b) * (long long)kN;
        if constexpr (LG == 1) { w[mb.i] = ai; w[mb.pi] = ap; }
        else w[woff] = odd ? ap : ai;
    });
}
}
