// Issue rate of the packed 16-bit minimum / maximum instructions the median kernel is made of (gfx950): a wave runs
// a long unrolled stream of one instruction kind on 16 independent registers; cycles per instruction from s_memtime
// (100 MHz constant clock -> scaled by the shader clock measured alongside with s_memrealtime is not needed: we
// report instructions per microsecond and per SIMD, which is what the kernel's time is made of).
//   hipcc -O3 --offload-arch=gfx950 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(64) void k_rate(uint32_t *out, int iters) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = 0x04000400u + threadIdx.x * 0x00010001u + i * 0x00030005u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int a = (i + 5) & 15, b = (i + 11) & 15;
                if (KIND == 0) asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 1) asm volatile("v_pk_max_u16 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 2) asm volatile("v_pk_minimum3_f16 %0, %1, %2, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 3) asm volatile("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]), "v"(r[(i + 3) & 15]));
                if (KIND == 4) asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]), "v"(r[(i + 3) & 15]));
                if (KIND == 5) asm volatile("v_pk_min_f16 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 6) asm volatile("v_min_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 7) asm volatile("v_min3_u32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]), "v"(r[(i + 3) & 15]));
                if (KIND == 8) asm volatile("v_pk_min_i16 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
                if (KIND == 9) asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(r[i]) : "v"(r[a]), "v"(r[b]));
            }
    }
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) x ^= r[i];
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int KIND>
static void run(const char *name, uint32_t *d, int waves) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(waves), dim3(64), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(waves), dim3(64), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)waves * iters * 64.0;   // wave instructions
    printf("%-34s waves %6d  %.3f ms  %.1f wave-instructions per us per SIMD (1024 SIMDs)\n", name, waves, ms, insts / (ms * 1e3) / 1024.0);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 64 * 16384 * 4);
    for (int waves : {1024, 2048, 8192}) {
        run<0>("v_pk_min_u16", d, waves);
        run<1>("v_pk_max_u16", d, waves);
        run<8>("v_pk_min_i16", d, waves);
        run<9>("v_pk_add_u16", d, waves);
        run<5>("v_pk_min_f16", d, waves);
        run<2>("v_pk_minimum3_f16 a, b, b", d, waves);
        run<3>("v_pk_minimum3_f16 a, b, c", d, waves);
        run<4>("v_pk_maximum3_f16 a, b, c", d, waves);
        run<6>("v_min_u32", d, waves);
        run<7>("v_min3_u32", d, waves);
    }
    return 0;
}
