// Checks the semantics of the cross-lane swap primitives used for the register<->lane-field transposes
// (v_permlane32_swap, v_permlane16_swap, bank-masked DPP row_ror:8 and row_shl/shr:4) against the definition
//   lanes with bit b = 0: hi' = partner.lo ;  lanes with bit b = 1: lo' = partner.hi ;  partner = lane ^ (1 << b)
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void xswap(unsigned &lo, unsigned &hi, int bit) {
    if (bit == 5) { auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
    else if (bit == 4) { auto r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false); lo = r[0]; hi = r[1]; }
    else if (bit == 3) {
        const unsigned t = hi;
        hi = __builtin_amdgcn_update_dpp(hi, lo, 0x128, 0xf, 0x3, false);   // lanes 0-7 of every row: hi <- lo of lane + 8
        lo = __builtin_amdgcn_update_dpp(lo, t, 0x128, 0xf, 0xc, false);    // lanes 8-15: lo <- old hi of lane - 8
    } else {
        const unsigned t = hi;
        hi = __builtin_amdgcn_update_dpp(hi, lo, 0x104, 0xf, 0x5, false);   // row_shl:4, banks 0 and 2: hi <- lo of lane + 4
        lo = __builtin_amdgcn_update_dpp(lo, t, 0x114, 0xf, 0xa, false);    // row_shr:4, banks 1 and 3: lo <- old hi of lane - 4
    }
}

__global__ void k(unsigned *o) {
    for (int bit = 2; bit <= 5; ++bit) {
        unsigned lo = threadIdx.x, hi = 1000 + threadIdx.x;
        xswap(lo, hi, bit);
        o[(bit - 2) * 128 + threadIdx.x] = lo;
        o[(bit - 2) * 128 + 64 + threadIdx.x] = hi;
    }
}

int main() {
    unsigned *d, h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int bit = 2; bit <= 5; ++bit)
        for (int l = 0; l < 64; ++l) {
            const int p = l ^ (1 << bit), b = (l >> bit) & 1;
            const unsigned elo = b ? 1000u + p : (unsigned)l, ehi = b ? 1000u + l : (unsigned)p;
            if (h[(bit - 2) * 128 + l] != elo || h[(bit - 2) * 128 + 64 + l] != ehi) {
                if (bad < 8) printf("bit %d lane %d: got (%u, %u) expected (%u, %u)\n", bit, l, h[(bit - 2) * 128 + l], h[(bit - 2) * 128 + 64 + l], elo, ehi);
                ++bad;
            }
        }
    printf(bad ? "MISMATCHES: %d\n" : "xlane swaps OK (%d)\n", bad);
    return bad != 0;
}
