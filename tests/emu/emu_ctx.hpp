#pragma once
// emu_ctx.hpp — TEST-ONLY thread emulation of the HIP tile kernels: the execution context shared by emu_harness.cpp and emu_lw.cpp.
//
// Compiles airwave_amd/csrc/device/tile_ols.hpp (the exact code the GPU runs) with g++ and
// executes one workgroup as 512 std::threads: workgroup barriers are std::barrier(512),
// wave-level syncs are std::barrier(64).  It exists so index math, twiddles and LDS hazards can
// be checked on the CPU-only build container before spending GPU minutes.  It is NOT part of the
// product library and nothing under airwave_amd/ links it.
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../airwave_amd/csrc/device/tile_ols.hpp"
#include "../../airwave_amd/csrc/device/tile_ola.hpp"
#include "../../airwave_amd/csrc/device/tile_march.hpp"
#include "../../airwave_amd/csrc/device/tile_lw.hpp"
#include "../../airwave_amd/csrc/device/tile_lw16.hpp"
#include "../../airwave_amd/csrc/device/eq_cascade.hpp"
#include "../../airwave_amd/csrc/host/eq.hpp"
#include "../../airwave_amd/csrc/host/tables.hpp"

namespace {

alignas(16) float g_zeros[1024] = {0};

struct EmuShared {
    std::barrier<> wg;
    std::vector<std::unique_ptr<std::barrier<>>> wave;
    std::vector<awk::cf> lds;
    std::vector<awk::cf> xs;      // cross-lane swap mailbox: [thread][2]
    explicit EmuShared(int threads = awk::kThreads, size_t lds_elems = (size_t)awk::kLdsElems) : wg(threads), lds(lds_elems), xs((size_t)threads * 2) {
        for (int w = 0; w < threads / 64; ++w) wave.emplace_back(new std::barrier<>(64));
    }
};

struct EmuCtx {
    int tid_;
    EmuShared *sh;
    int tid() const { return tid_; }
    int lane() const { return tid_ & 63; }
    int wave() const { return tid_ >> 6; }
    awk::cf *lds() const { return sh->lds.data(); }
    awk::cf opaque(awk::cf v) const { return v; }
    int opaque_i(int v) const { return v; }
    void stamp(int) const {}
    void sched_fence() const {}
    void sched_fence_hard() const {}
    template <int P> void prio() const {}
    void flush_stamps() const {}
    awk::cf ld(const awk::cf *p) const { return *p; }
    void stagger(int, int) const {}
    void barrier() const { sh->wg.arrive_and_wait(); }
    // lanes with bit b = 0: hi' = partner.lo ; lanes with bit b = 1: lo' = partner.hi ; partner = lane ^ (1 << b)
    void xswap(awk::cf &lo, awk::cf &hi, int bit) const {
        sh->xs[(size_t)tid_ * 2] = lo; sh->xs[(size_t)tid_ * 2 + 1] = hi;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const int p = tid_ ^ (1 << bit);
        if ((tid_ >> bit) & 1) lo = sh->xs[(size_t)p * 2 + 1]; else hi = sh->xs[(size_t)p * 2];
        sh->wave[tid_ >> 6]->arrive_and_wait();
    }
    void wave_sync() const { sh->wave[tid_ >> 6]->arrive_and_wait(); }
    template <int STRIDE> void ld8x2(awk::cf (&a)[8], const awk::cf *p0, awk::cf (&b)[8], const awk::cf *p1) const {
        for (int i = 0; i < 8; ++i) { a[i] = p0[i * STRIDE]; b[i] = p1[i * STRIDE]; }
    }
    void st_stream(awk::cf *q, awk::cf v) const { *q = v; }
    awk::cf ld_stream(const awk::cf *q) const { return *q; }
    void st_stream4(float *q, float a, float b, float c, float d) const { q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
    // buffer loads: every dword at or past the descriptor's size reads as zero (the hardware's range check)
    struct Buf { const unsigned char *base; unsigned bytes; };
    Buf buf(const void *base, unsigned bytes) const { return Buf{static_cast<const unsigned char *>(base), bytes}; }
    template <int N> void buf_ld(const Buf &b, unsigned off, int imm, float *dst) const {
        for (int i = 0; i < N; ++i) {
            const unsigned long long o = (unsigned long long)off + (unsigned)imm + 4u * i;
            if (o + 4 <= b.bytes) std::memcpy(&dst[i], b.base + o, 4); else dst[i] = 0.0f;
        }
    }
    awk::cf xchg1(awk::cf v) const {          // value of lane ^ 1
        sh->xs[(size_t)tid_ * 2] = v;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const awk::cf r = sh->xs[(size_t)(tid_ ^ 1) * 2];
        sh->wave[tid_ >> 6]->arrive_and_wait();
        return r;
    }
    // in-register scan moves of eq_cascade.hpp (DPP on the GPU): value of another lane of the wave, or zero / fill
    double lane_value(double v, int src_lane, double otherwise) const {     // src_lane < 0: no source
        double *box = reinterpret_cast<double *>(&sh->xs[(size_t)tid_ * 2]);
        *box = v;
        sh->wave[tid_ >> 6]->arrive_and_wait();
        const double r = src_lane < 0 ? otherwise : *reinterpret_cast<const double *>(&sh->xs[(size_t)((tid_ & ~63) + src_lane) * 2]);
        sh->wave[tid_ >> 6]->arrive_and_wait();
        return r;
    }
    void fma_in_place(double &x, double a, double c, double, double) const { x = __builtin_fma(a, x, c); }
    template <int D> double row_shr(double v) const { const int l = tid_ & 63; return lane_value(v, (l & 15) >= D ? l - D : -1, 0.0); }
    double row_bcast15(double v) const { const int l = tid_ & 63; return lane_value(v, ((l >> 4) & 1) ? (l & ~15) - 1 : -1, 0.0); }
    double row_bcast31(double v) const { const int l = tid_ & 63; return lane_value(v, l >= 32 ? 31 : -1, 0.0); }
    double wave_shr1(double v, double fill) const { const int l = tid_ & 63; return lane_value(v, l > 0 ? l - 1 : -1, fill); }
};

}  // namespace

