// fft_core.hip — cost of ONE 8192-point pair transform of the tile code with no global memory traffic: pass 1 (radix-16 +
// twiddles + scatter to LDS rows), barrier, per-wave 512-point sub-FFTs of two rows.  One 512-thread workgroup per CU,
// REPS transforms each (two pairs per barrier interval, like the kernels).
// Both row-transform forms: the 8 x 8 x 8 one of rounds 1-3 (k_core) and the half-wave one of round 4 (k_core16).  Build with the kernels'
// -fno-slp-vectorize (without it hipcc packs the butterflies into v_pk_* and everything runs 25 % slower):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Iairwave_amd/csrc -Iinclude tools/ubench/fft_core.hip -o tools/ubench/fft_core
#include "device/tile_ols.hpp"
#include "device/gpu_ctx.hpp"
#include "device/tile_lw16.hpp"
#include <cstdio>
#include <vector>
namespace awk {
template <int VARIANT>
__global__ void __launch_bounds__(kThreads) k_core(const cf *tw1, const cf *twa_g, const cf *twb_g, cf *sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int t = ctx.tid(), lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds(), *buf1 = buf0 + kBufElems, *twa = buf1 + kBufElems, *twb = twa + kTwaElems;
    const cf w1 = tw1[t];
    twa[t] = twa_g[t];
    if (t < kTwbElems) twb[t] = twb_g[t];
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = mk(0.001f * (t + j), 0.002f * (t - j));
    cf acc = mk(0.f, 0.f);
    for (int r = 0; r < reps; ++r) {
        if (r > 0) ctx.barrier();
        {
            cf pw[16];
            tw_powers(ctx.opaque(w1), pw);
            cf y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = x[j];
            if constexpr (!(VARIANT & 1)) pair_pass1(y, pw, buf0, t);
            else { fft16<false>(y); for (int k1 = 1; k1 < 16; ++k1) y[k1] = cmul(y[k1], pw[k1]); acc = acc + y[3]; }   // no scatter
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = mk(x[j].y, x[j].x);
            if constexpr (!(VARIANT & 1)) pair_pass1(y, pw, buf1, t);
            else { fft16<false>(y); for (int k1 = 1; k1 < 16; ++k1) y[k1] = cmul(y[k1], pw[k1]); acc = acc + y[5]; }
        }
        if constexpr (!(VARIANT & 8)) ctx.barrier();
        ctx.stagger(wave, 0);
        if constexpr (VARIANT & 2) { acc = acc + buf0[t]; continue; }      // no sub-FFTs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            cf *buf = h == 0 ? buf0 : buf1;
            cf *row0 = buf + wave_row(wave, 0) * kRowStride;
            cf *row1 = buf + wave_row(wave, 1) * kRowStride;
            cf z[2][8];
            ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
            ctx.wave_sync();
            sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) { acc = acc + z[0][kc]; acc = acc + z[1][kc]; }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j].x += 1e-6f * acc.x;     // dependency between repetitions
    }
    sink[blockIdx.x * kThreads + t] = acc;
}
}
namespace awk {
__global__ void __launch_bounds__(kThreads, 4) k_core1(const cf *tw1, const cf *twa_g, const cf *twb_g, cf *sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int t = ctx.tid(), lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds(), *twa = buf0 + kBufElems, *twb = twa + kTwaElems;
    const cf w1 = tw1[t];
    twa[t] = twa_g[t];
    if (t < kTwbElems) twb[t] = twb_g[t];
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = mk(0.001f * (t + j), 0.002f * (t - j));
    cf acc = mk(0.f, 0.f);
    for (int r = 0; r < reps; ++r) {
        if (r > 0) ctx.barrier();
        {
            cf pw[16];
            tw_powers(ctx.opaque(w1), pw);
            cf y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = x[j];
            pair_pass1(y, pw, buf0, t);
        }
        ctx.barrier();
        cf *row0 = buf0 + wave_row(wave, 0) * kRowStride;
        cf *row1 = buf0 + wave_row(wave, 1) * kRowStride;
        cf z[2][8];
        ctx.template ld8x2<64>(z[0], row0 + lane, z[1], row1 + lane);
        ctx.wave_sync();
        sub_fft512x2<false>(ctx, z, row0, row1, twa, twb, lane);
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) { acc = acc + z[0][kc]; acc = acc + z[1][kc]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j].x += 1e-6f * acc.x;
    }
    sink[blockIdx.x * kThreads + t] = acc;
}
}

namespace awk {
// The same 512-point row transforms on the 16-point core (tile_ols.hpp, sub_fft512h_fwd): a HALF-wave per row, 16 values per lane.
template <int VARIANT>
__global__ void __launch_bounds__(kThreads) k_core16(const cf *tw1, const cf *twa_g, const cf *, cf *sink, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    GpuCtx ctx{reinterpret_cast<cf *>(smem), nullptr};
    const int t = ctx.tid(), lane = ctx.lane(), wave = ctx.wave();
    cf *buf0 = ctx.lds(), *buf1 = buf0 + kBufElems, *twh = buf1 + kBufElems;
    const cf w1 = tw1[t];
    twh[t] = hl_twiddle(twa_g, t);
    cf x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = mk(0.001f * (t + j), 0.002f * (t - j));
    cf acc = mk(0.f, 0.f);
    for (int r = 0; r < reps; ++r) {
        if (r > 0) ctx.barrier();
        {
            cf pw[16];
            tw_powers(ctx.opaque(w1), pw);
            cf y[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = x[j];
            pair_pass1(y, pw, buf0, t);
#pragma unroll
            for (int j = 0; j < 16; ++j) y[j] = mk(x[j].y, x[j].x);
            pair_pass1(y, pw, buf1, t);
        }
        ctx.barrier();
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            cf *buf = hh == 0 ? buf0 : buf1;
            const HLane L = hl_make(ctx, buf, twh, lane, wave);
            cf z[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) z[j] = ctx.ld(L.row + L.h + 32 * j);
            sub_fft512h_fwd(ctx, z, L);
#pragma unroll
            for (int kc = 0; kc < 16; ++kc) acc = acc + z[kc];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j].x += 1e-6f * acc.x;
    }
    sink[blockIdx.x * kThreads + t] = acc;
}
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    using namespace awk;
    std::vector<cf> tw1(512), twa(512), twb(64);
    for (int t = 0; t < 512; ++t) { tw1[t] = mk(cosf(-2 * 3.14159265f * t / 8192), sinf(-2 * 3.14159265f * t / 8192)); }
    for (int ka = 0; ka < 8; ++ka) for (int l = 0; l < 64; ++l) twa[ka * 64 + l] = mk(cosf(-2 * 3.14159265f * l * ka / 512), sinf(-2 * 3.14159265f * l * ka / 512));
    for (int kb = 0; kb < 8; ++kb) for (int l = 0; l < 8; ++l) twb[kb * 8 + l] = mk(cosf(-2 * 3.14159265f * l * kb / 64), sinf(-2 * 3.14159265f * l * kb / 64));
    cf *d1, *da, *db, *sink;
    CK(hipMalloc((void **)&d1, 512 * 8)); CK(hipMalloc((void **)&da, 512 * 8)); CK(hipMalloc((void **)&db, 64 * 8)); CK(hipMalloc((void **)&sink, 1024 * 512 * 8));
    CK(hipMemcpy(d1, tw1.data(), 512 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, twa.data(), 512 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, twb.data(), 64 * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    auto run = [&](auto kern, const char *name) {
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(256), dim3(kThreads), kLdsBytes, 0, d1, da, db, sink, reps);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-44s %.3f us per pair transform per CU\n", name, best * 1e3 / (2.0 * reps));
    };
    run(k_core<0>, "full (pass 1 + barrier + sub-FFTs)");
    {   // one pair per barrier interval, 78 KB of LDS, two workgroups per CU (512 workgroups, REPS transforms each)
        (void)hipFuncSetAttribute((const void *)k_core1, hipFuncAttributeMaxDynamicSharedMemorySize, kInvLdsBytes);
        for (int wgs : {256, 512}) {
            float best = 1e9f;
            for (int it = 0; it < 3; ++it) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_core1, dim3(wgs), dim3(kThreads), kInvLdsBytes, 0, d1, da, db, sink, reps);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("one pair per interval, %d workgroups (%d per CU): %.3f us per pair transform per CU\n", wgs, wgs / 256, best * 1e3 / (reps * (wgs / 256.0)));
        }
    }
    {
        cf *d512 = da, *d32 = db;
        auto k = k_core16<0>;
        (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
        float best = 1e9f;
        for (int it = 0; it < 3; ++it) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(kThreads), kLdsBytes, 0, d1, d512, d32, sink, reps);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%-44s %.3f us per pair transform per CU\n", "16-point core: half-wave per row", best * 1e3 / (2.0 * reps));
    }
    run(k_core<1>, "pass 1 without its LDS scatter");
    run(k_core<2>, "no sub-FFTs (pass 1 + barriers only)");
    run(k_core<3>, "butterflies of pass 1 only");
    run(k_core<8>, "full without the second barrier (racy)");
    return 0;
}
