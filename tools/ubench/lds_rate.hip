// LDS instruction throughput on gfx950 with inline asm (exact opcodes, many ops in flight).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X X X X X X X X

template <int OP>
__global__ void __launch_bounds__(512) k_lds(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned lane = threadIdx.x;
    // conflict-free, lane-contiguous addresses
    unsigned a8 = lane * 8, a16 = lane * 16, a4 = lane * 4;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {        // 16 x ds_read_b64
            asm volatile(REP8("ds_read_b64 v[20:21], %0 offset:0\n ds_read_b64 v[22:23], %0 offset:4096\n")
                         "s_waitcnt lgkmcnt(0)\n" ::"v"(a8) : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 1) { // 8 x ds_read2_b64 (same bytes as 16 read_b64)
            asm volatile(REP8("ds_read2_b64 v[20:23], %0 offset0:0 offset1:64\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a8)
                         : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 2) { // 8 x ds_read_b128
            asm volatile(REP8("ds_read_b128 v[20:23], %0 offset:0\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a16)
                         : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 3) { // 16 x ds_write_b64
            asm volatile(REP8("ds_write_b64 %0, v[20:21] offset:0\n ds_write_b64 %0, v[22:23] offset:4096\n")
                         "s_waitcnt lgkmcnt(0)\n" ::"v"(a8) : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 4) { // 8 x ds_write2st64_b64
            asm volatile(REP8("ds_write2st64_b64 %0, v[20:21], v[22:23] offset0:0 offset1:8\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a8)
                         : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 5) { // 8 x ds_write_b128
            asm volatile(REP8("ds_write_b128 %0, v[20:23] offset:0\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a16)
                         : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 6) { // 16 x ds_read_b32
            asm volatile(REP8("ds_read_b32 v20, %0 offset:0\n ds_read_b32 v21, %0 offset:2048\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a4)
                         : "v20", "v21", "memory");
        } else if (OP == 7) { // 16 x ds_write_b32
            asm volatile(REP8("ds_write_b32 %0, v20 offset:0\n ds_write_b32 %0, v21 offset:2048\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a4)
                         : "v20", "v21", "memory");
        } else if (OP == 8) { // 16 x ds_bpermute_b32
            asm volatile(REP8("ds_bpermute_b32 v20, %0, v22\n ds_bpermute_b32 v21, %0, v23\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a4)
                         : "v20", "v21", "v22", "v23", "memory");
        } else if (OP == 10) { // 16 x ds_read_b64 with EXEC = 0
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n"
                         REP8("ds_read_b64 v[20:21], %0 offset:0\n ds_read_b64 v[22:23], %0 offset:4096\n")
                         "s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[20:21]\n" ::"v"(a8) : "v20", "v21", "v22", "v23", "s20", "s21", "memory");
        } else if (OP == 11) { // 16 x ds_write_b64 with EXEC = 0
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n"
                         REP8("ds_write_b64 %0, v[20:21] offset:0\n ds_write_b64 %0, v[22:23] offset:4096\n")
                         "s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, s[20:21]\n" ::"v"(a8) : "v20", "v21", "v22", "v23", "s20", "s21", "memory");
        } else if (OP == 9) { // 8 x ds_read2st64_b64
            asm volatile(REP8("ds_read2st64_b64 v[20:23], %0 offset0:0 offset1:8\n") "s_waitcnt lgkmcnt(0)\n" ::"v"(a8)
                         : "v20", "v21", "v22", "v23", "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int OP> int run(const char *name, int bytes_per_iter_per_lane, int instr_per_iter, float *out) {
    CK(hipFuncSetAttribute((const void *)k_lds<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    const int iters = 4000;
    for (int threads : {64, 256, 512}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_lds<OP>, dim3(256), dim3(threads), 64 * 1024, 0, out, iters); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lds<OP>, dim3(256), dim3(threads), 64 * 1024, 0, out, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double cyc = ms * 1e-3 * 2.4e9;
        const double waves = threads / 64.0;
        printf("%-22s waves/CU %2.0f: %.3f ms  CU-cycles per wave-instr %.2f   B/clk/CU %.1f (nominal 2.4 GHz)\n", name, waves, ms,
               cyc / (iters * instr_per_iter * waves), (double)iters * bytes_per_iter_per_lane * threads / cyc);
    }
    return 0;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
    run<0>("16x ds_read_b64", 128, 16, out);
    run<1>("8x ds_read2_b64", 128, 8, out);
    run<9>("8x ds_read2st64_b64", 128, 8, out);
    run<2>("8x ds_read_b128", 128, 8, out);
    run<3>("16x ds_write_b64", 128, 16, out);
    run<4>("8x ds_write2st64_b64", 128, 8, out);
    run<5>("8x ds_write_b128", 128, 8, out);
    run<6>("16x ds_read_b32", 64, 16, out);
    run<7>("16x ds_write_b32", 64, 16, out);
    run<8>("16x ds_bpermute_b32", 64, 16, out);
    run<10>("16x ds_read_b64 EXEC=0", 128, 16, out);
    run<11>("16x ds_write_b64 EXEC=0", 128, 16, out);
    return 0;
}
