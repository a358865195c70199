// tools/ubench/issue_rate2.hip -- what the instruction classes of the expander and the pack kernel cost when a SIMD is
// full (WAVES waves per SIMD, all running the same loop of 32 instructions over 16 independent registers per lane):
// cycles per wave-instruction per SIMD.  Mixed loops tell whether scalar instructions, s_nop and waits ride along for
// free beside vector instructions of other waves or cost issue time of their own.  (Measurement tool, not product.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP16(STMT) _Pragma("unroll") for (int i = 0; i < 16; i++) { STMT; }

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t u, int iters) {
    uint32_t z[16];
    __shared__ uint32_t lds[1024];
    for (int i = 0; i < 16; i++) z[i] = threadIdx.x * 2654435761u + i;
    lds[threadIdx.x] = u; lds[threadIdx.x + 256] = u; lds[threadIdx.x + 512] = u; lds[threadIdx.x + 768] = u;
    __syncthreads();
    uint32_t s0 = u, s1 = u + 1, s2 = u + 2, s3 = u + 3;
    for (int it = 0; it < iters; it++) {
        // every mode: 32 instructions of the measured kind (or 16 + 16 of a mix) per iteration
        if (MODE == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[i]) : "v"(u))) REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 1) { REP16(asm volatile("v_add_u32 %0, %1, %0" : "+v"(z[i]) : "s"(s0))) REP16(asm volatile("v_add_u32 %0, %1, %0" : "+v"(z[i]) : "s"(s0))) }
        if (MODE == 2) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(z[i]) : "v"(u) : "vcc")) REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(z[i]) : "v"(u) : "vcc")) }
        if (MODE == 3) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(z[i]) : "v"(u), "s"((uint64_t)s0))) REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(z[i]) : "v"(u), "s"((uint64_t)s0))) }
        if (MODE == 4) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(z[i]), "v"(u) : "vcc")) REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(z[i]), "v"(u) : "vcc")) }
        if (MODE == 5) { REP16(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(z[i]) : "v"(u) : "vcc")) }
        if (MODE == 6) { REP16(asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s1) : "v"(z[i]))) REP16(asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s2) : "v"(z[i]))) }
        if (MODE == 7) { REP16(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s1) : "v"(z[i]))) REP16(asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s2) : "v"(z[i]))) }
        if (MODE == 8) { REP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s1) : "s"(s0))) REP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s2) : "s"(s0))) }
        if (MODE == 9) { REP16(asm volatile("v_add_u32 %0, %0, %2\n s_add_u32 %1, %1, %3" : "+v"(z[i]), "+s"(s1) : "v"(u), "s"(s0))) }
        if (MODE == 10) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 11) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 3" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 12) { REP16(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(z[i]))) REP16(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(z[i]))) }
        if (MODE == 13) { REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(z[i]) : "s"(s0))) REP16(asm volatile("v_mbcnt_hi_u32_b32 %0, %1, %0" : "+v"(z[i]) : "s"(s0))) }
        if (MODE == 14) { REP16(asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(z[i]) : "v"(u))) REP16(asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 15) { REP16(asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(z[i]))) REP16(asm volatile("v_ffbl_b32 %0, %0" : "+v"(z[i]))) }
        if (MODE == 16) { REP16(asm volatile("ds_read_b32 %0, %1" : "=v"(z[i]) : "v"((threadIdx.x & 255) * 4))) REP16(asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(z[i]) : "v"((threadIdx.x & 255) * 4))) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 17) { REP16(asm volatile("ds_write_b32 %1, %0" : : "v"(z[i]), "v"((threadIdx.x & 255) * 4))) REP16(asm volatile("ds_write_b32 %1, %0 offset:1024" : : "v"(z[i]), "v"((threadIdx.x & 255) * 4))) asm volatile("s_waitcnt lgkmcnt(0)"); }
        if (MODE == 18) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 19) { REP16(asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(z[i]) : "v"(u))) REP16(asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(z[i]) : "v"(u))) }
        if (MODE == 20) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_cmp_lt_u32 %2, %3" : "+v"(z[i]) : "v"(u), "s"(s0), "s"(s1) : "scc")) }
        if (MODE == 21) { REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(z[i]) : "s"(s0))) REP16(asm volatile("v_bitop3_b32 %0, %0, %1, %1 bitop3:0x96" : "+v"(z[i]) : "s"(s0))) }
        if (MODE == 22) { REP16(asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(z[i]))) REP16(asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(z[i]))) }
        if (MODE == 23) { REP16(asm volatile("v_sub_u32 %0, %0, %1" : "+v"(z[i]) : "v"(u))) REP16(asm volatile("v_or_b32 %0, 0x80808080, %0" : "+v"(z[i]))) }
    }
    uint32_t s = s1 + s2 + s3;
    for (int i = 0; i < 16; i++) s += z[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static int g_waves = 4;
template <int MODE>
void run(const char *name, uint32_t *d, int per_iter = 32) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * g_waves), dim3(256), 0, 0, d, 0x01020304u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-44s %7.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz (%d per iteration)\n", name, best, best * 1e-3 * 2.4e9 / ((double)g_waves * iters * per_iter), per_iter);
}

int main(int argc, char **argv) {
    if (argc > 1) g_waves = atoi(argv[1]);
    printf("waves per SIMD: %d\n", g_waves);
    uint32_t *d; hipMalloc(&d, 256 * 256 * 8 * 4);
    run<0>("v_add_u32 v,v", d); run<1>("v_add_u32 s,v", d); run<21>("v_bitop3 v,s,s", d); run<22>("v_lshrrev/lshlrev imm", d); run<23>("v_sub / v_or literal", d);
    run<2>("v_cndmask vcc (VOP2)", d); run<3>("v_cndmask_e64 sgpr pair", d);
    run<4>("v_cmp -> vcc", d); run<5>("v_cmp -> vcc ; v_cndmask vcc (pairs)", d, 32);
    run<6>("v_readlane -> sgpr", d); run<7>("v_readfirstlane", d);
    run<8>("s_add_u32 only", d); run<9>("v_add ; s_add interleaved (16+16)", d, 32); run<20>("v_add ; s_cmp interleaved (16+16)", d, 32);
    run<10>("v_add ; s_nop 0 (16+16)", d, 32); run<11>("v_add ; s_nop 3 (16+16)", d, 32); run<18>("v_add ; s_waitcnt lgkmcnt(0) (16+16)", d, 32);
    run<12>("v_add_u32 dpp row_shr", d); run<13>("v_mbcnt_lo/hi", d); run<14>("v_lshl_or / v_and_or", d); run<15>("v_bfe / v_ffbl", d); run<19>("v_add3 / v_or3", d);
    run<16>("ds_read_b32 (32 + wait)", d); run<17>("ds_write_b32 (32 + wait)", d);
    return 0;
}
