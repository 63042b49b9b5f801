// ola_bench.hip — the overlap-add tile kernel (tile_ola.hpp) alone, on cfg-2-shaped synthetic data (128 streams x 10 s, 4320 taps), built in
// seconds; the counterpart of tile_bench.hip (timing only: tables and input are random).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -Iairwave_amd/csrc -Iinclude tools/ubench/ola_bench.hip -o tools/ubench/ola_bench
//   -DTB_CS=14 -DTB_H=7 ; run: ola_bench [seconds of warm-up launches] [workgroups]
#include "device/ola_kernel.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#ifndef TB_CS
#define TB_CS 8
#endif
#ifndef TB_H
#define TB_H 7
#endif
#define TB_NP ((TB_CS + 1) / 2)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
    using namespace awk;
    const double sustain = argc > 1 ? atof(argv[1]) : 0.3;
    const int wgs = argc > 2 ? atoi(argv[2]) : 256;
    const int S = 128, C = TB_CS, taps = 4320; const long long F = 480000;
    const int hop = 512 * TB_H, hist = taps - 1;
    std::vector<float> h_in((size_t)4 << 20);
    for (size_t i = 0; i < h_in.size(); ++i) h_in[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    float *d_in, *d_out, *d_hist; cf2 *d_tab; cf *d1, *da;
    const size_t n_in = (size_t)S * F * C, n_out = (size_t)S * F * 2, n_hist = (size_t)S * hist * C;
    CK(hipMalloc((void **)&d_in, (n_in + 64) * 4)); CK(hipMalloc((void **)&d_out, n_out * 4)); CK(hipMalloc((void **)&d_hist, n_hist * 4));
    CK(hipMemset(d_hist, 0, n_hist * 4));
    for (size_t off = 0; off < n_in; off += h_in.size()) CK(hipMemcpy(d_in + off, h_in.data(), std::min(h_in.size(), n_in - off) * 4, hipMemcpyHostToDevice));
    const size_t n_tab = (size_t)(TB_NP + 1) * kN;
    std::vector<cf2> tab(n_tab);
    for (size_t i = 0; i < n_tab; ++i) { tab[i].a = mk(1e-4f * (i % 97), -1e-4f * (i % 89)); tab[i].b = mk(2e-5f * (i % 83), 1e-5f * (i % 79)); }
    CK(hipMalloc((void **)&d_tab, n_tab * sizeof(cf2))); CK(hipMemcpy(d_tab, tab.data(), n_tab * sizeof(cf2), hipMemcpyHostToDevice));
    std::vector<cf> tw1(512), twa(512);
    for (int t = 0; t < 512; ++t) tw1[t] = mk(cosf(-2 * 3.14159265f * t / 8192), sinf(-2 * 3.14159265f * t / 8192));
    for (int ka = 0; ka < 8; ++ka) for (int l = 0; l < 64; ++l) twa[ka * 64 + l] = mk(cosf(-2 * 3.14159265f * l * ka / 512), sinf(-2 * 3.14159265f * l * ka / 512));
    CK(hipMalloc((void **)&d1, 512 * 8)); CK(hipMalloc((void **)&da, 512 * 8));
    CK(hipMemcpy(d1, tw1.data(), 512 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(da, twa.data(), 512 * 8, hipMemcpyHostToDevice));
    TileParams p{};
    p.in = d_in; p.out = d_out; p.hist = d_hist; p.tab = d_tab; p.tw1 = d1; p.twa = da; p.twb = da; p.zeros = d_in;
    p.frames = F; p.n_channels = C; p.n_pairs = TB_NP; p.hop = hop; p.hist_len = hist;
    p.tiles_per_stream = (int)((F + hop - 1) / hop);
    const long long n_tiles = (long long)S * p.tiles_per_stream;
    auto kern = aw_fused_ola_kernel<TB_CS, TB_NP, TB_H>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (double spent = 0; spent < sustain;) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), kLdsBytes, 0, p, n_tiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        spent += ms * 1e-3;
    }
    for (int it = 0; it < 8; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(kThreads), kLdsBytes, 0, p, n_tiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("ola kernel <%d,%d,H=%d> on %d workgroups: %.3f ms for %lld blocks of %d frames (whole call, no boundary launch) -> %.2f us per block per CU, %.2f G frames/s\n",
           TB_CS, TB_NP, TB_H, wgs, best, n_tiles, hop, best * 1e3 / (n_tiles / (double)wgs), (double)S * F / best / 1e6);
    return 0;
}
