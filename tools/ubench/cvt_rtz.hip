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
int main() {
    const float v[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 3.999f, 254.5f, 254.999f, 255.0f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.6f, -3.f, __builtin_nanf(""), 127.99999f};
    const int n = sizeof v / sizeof *v;
    float *d; uint32_t *o; uint32_t h[2 * 32];
    hipMalloc(&d, sizeof v); hipMalloc(&o, sizeof h);
    hipMemcpy(d, v, sizeof v, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(h, o, sizeof(uint32_t) * 2 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) printf("%14.6f  rne %3u  rtz %3u  want(trunc+sat) %3d\n", v[i], h[i], h[n + i], v[i] != v[i] ? 0 : v[i] < 0 ? 0 : v[i] > 255 ? 255 : (int)v[i]);
    return 0;
}
