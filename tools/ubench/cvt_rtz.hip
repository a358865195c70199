// Does v_cvt_pk_u8_f32 follow MODE.fp_round?  (If round-toward-zero made it truncate, the 3x3 filter's v_trunc_f32 in
// front of every output byte -- 16 of ~200 instructions per row -- could go.)  Prints the conversion of a few values in
// the default mode (round to nearest even) and with the single-precision round mode set to toward-zero.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *in, uint32_t *out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const float f = in[i];
    out[i] = __builtin_amdgcn_cvt_pk_u8_f32(f, 0u, 0u);
    // MODE[1:0] = single-precision round mode: 0 nearest even, 1 +inf, 2 -inf, 3 toward zero
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3");
    uint32_t r;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(f));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0");
    out[n + i] = r;
}
// every one of the 2^32 float bit patterns: v_cvt_pk_u8_f32 under round-toward-zero against truncate-and-saturate
// (negatives, -0 and NaN to 0; >= 255 and +inf to 255) -- the definition both convolution kernels and the oracle share
__global__ void k_all(unsigned long long *bad, uint32_t *first_bad) {
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) << 8;
    uint32_t mism = 0;
    for (uint32_t i = 0; i < 256; i++) {
        const uint32_t bits = base + i;
        const float f = __uint_as_float(bits);
        uint32_t r;
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\ts_nop 1\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\t"
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1" : "=v"(r) : "v"(f));
        const uint32_t want = !(f > 0.0f) ? 0u : f >= 255.0f ? 255u : (uint32_t)f;
        if (r != want) { mism++; atomicMin(first_bad, bits); }
    }
    if (mism) atomicAdd(bad, (unsigned long long)mism);
}

int main() {
    const float v[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 3.999f, 254.5f, 254.999f, 255.0f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.6f, -3.f, __builtin_nanf(""), 127.99999f};
    const int n = sizeof v / sizeof *v;
    float *d; uint32_t *o; uint32_t h[2 * 32];
    hipMalloc(&d, sizeof v); hipMalloc(&o, sizeof h);
    hipMemcpy(d, v, sizeof v, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(h, o, sizeof(uint32_t) * 2 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%14.6f  rne %3u  rtz %3u  want(trunc+sat) %3d\n", v[i], h[i], h[n + i], v[i] != v[i] ? 0 : v[i] < 0 ? 0 : v[i] > 255 ? 255 : (int)v[i]);
    unsigned long long *bad; uint32_t *fb; unsigned long long hb = 0; uint32_t hfb = 0xffffffffu;
    hipMalloc(&bad, 8); hipMalloc(&fb, 4);
    hipMemcpy(bad, &hb, 8, hipMemcpyHostToDevice); hipMemcpy(fb, &hfb, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_all, dim3(1u << 16), dim3(256), 0, 0, bad, fb);   // 2^16 x 256 threads x 256 values = 2^32
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hfb, fb, 4, hipMemcpyDeviceToHost);
    printf("all 2^32 float bit patterns under round-toward-zero: %llu mismatches against truncate-and-saturate", hb);
    if (hb) printf(" (first: bits 0x%08x)", hfb);
    printf("\n");
    return hb != 0;
}
