// tools/ubench/valu_rate.hip -- issue rate of the VALU instructions the kernels of this repository lean on.
// 4 waves per SIMD, 16 independent accumulators per lane, 32 instructions per loop iteration; prints the
// average cycles per wave-instruction per SIMD at the 2.4 GHz peak clock.  (Measurement tool, not product.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BODY32(STMT) _Pragma("unroll") for (int r = 0; r < 2; r++) _Pragma("unroll") for (int i = 0; i < 16; i++) { STMT; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, uint32_t u, int iters) {
    f32x2 x[16]; float y[16]; uint32_t z[16];
    for (int i = 0; i < 16; i++) { x[i] = f32x2{(float)threadIdx.x + i, (float)i}; y[i] = (float)threadIdx.x * i; z[i] = threadIdx.x * 2654435761u + i; }
    const f32x2 c = f32x2{a, a * 0.5f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) BODY32(asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a)))
        if (MODE == 1) BODY32(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a)))
        if (MODE == 2) BODY32(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(y[i]) : "v"(a)))
        if (MODE == 3) BODY32(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c)))
        if (MODE == 4) BODY32(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c)))
        if (MODE == 5) BODY32(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(c)))
        if (MODE == 6) BODY32(asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[i]) : "v"(u)))
        if (MODE == 7) BODY32(asm volatile("v_and_b32 %0, %0, %1" : "+v"(z[i]) : "v"(u)))
        if (MODE == 8) BODY32(asm volatile("v_mov_b32 %0, %1" : "+v"(z[i]) : "v"(z[(i + 1) & 15])))
        if (MODE == 9) BODY32(asm volatile("v_cvt_f32_ubyte1 %0, %1" : "+v"(y[i]) : "v"(z[i])))
        if (MODE == 10) BODY32(asm volatile("v_cvt_u32_f32 %0, %1" : "+v"(z[i]) : "v"(y[i])))
        if (MODE == 11) BODY32(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(z[i]) : "v"(u), "v"(z[(i + 1) & 15])))
        if (MODE == 12) BODY32(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(z[i]) : "v"(u), "v"(z[(i + 1) & 15])))
        if (MODE == 13) BODY32(asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(z[i]) : "v"(u), "v"(z[(i + 1) & 15])))
        if (MODE == 14) BODY32(asm volatile("v_dot4_u32_u8 %0, %0, %1, %2" : "+v"(z[i]) : "v"(u), "v"(z[(i + 1) & 15])))
        if (MODE == 15) BODY32(asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(z[i])))
        if (MODE == 16) BODY32(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(z[i]) : "v"(u) : "vcc"))
        if (MODE == 17) BODY32(asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(z[i]) : "v"(u)))
        if (MODE == 18) BODY32(asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(z[i]) : "v"(u)))
        if (MODE == 19) BODY32(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&x[i]) : "v"(*(const double *)&c)))
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += x[i].x + x[i].y + y[i] + (float)z[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10000;
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, 1.0001f, 0x01020304u, iters);   // 4 waves per SIMD
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-16s %.3f ms  %.2f cycles per wave-instruction\n", name, best, best * 1e-3 * 2.4e9 / (4.0 * iters * 32));
}

int main() {
    float *d; hipMalloc(&d, 256 * 1024 * 4);
    run<0>("v_add_f32", d); run<1>("v_mul_f32", d); run<2>("v_fma_f32", d);
    run<3>("v_pk_add_f32", d); run<4>("v_pk_mul_f32", d); run<5>("v_pk_fma_f32", d);
    run<6>("v_add_u32", d); run<7>("v_and_b32", d); run<8>("v_mov_b32", d);
    run<9>("v_cvt_f32_ubyte1", d); run<10>("v_cvt_u32_f32", d); run<11>("v_perm_b32", d);
    run<12>("v_bitop3_b32", d); run<13>("v_sad_u8", d); run<14>("v_dot4_u32_u8", d);
    run<15>("v_add_u32 dpp", d); run<16>("v_cndmask_b32", d); run<17>("v_lshl_add_u32", d);
    run<18>("v_bcnt_u32_b32", d); run<19>("v_mul_f64", d);
    return 0;
}
