// How fast does v_mfma_f32_32x32x2_f32 issue per SIMD with W waves per SIMD and C independent accumulator chains per wave?
// (K2 / K4 of the select pass run 2-3 waves per SIMD with 3 / 1 chains.)  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int C>
__global__ void k(float* out, int iters, float a, float b) {
    f32x16 acc[C];
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    if (s == 123.456f) out[0] = s;
}
template <int C>
void run(int waves_per_simd) {
    float* out;
    (void)hipMalloc(&out, 4);
    const int iters = 2000 / C;
    const int block = 256 * waves_per_simd;   // 4 SIMDs x waves_per_simd waves, one workgroup per CU
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<C><<<256, block>>>(out, 10, 1.0f, 2.0f);
    (void)hipEventRecord(e0);
    k<C><<<256, block>>>(out, iters, 1.0f, 2.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * C * waves_per_simd;
    const double flops = mfma_per_simd * 1024 * 4096.0;
    printf("waves/SIMD %d chains %d: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per SIMD (64 cycles @2.4 GHz = 26.7 ns)\n",
           waves_per_simd, C, ms, flops / ms / 1e9, ms * 1e6 / mfma_per_simd);
}
int main() {
    for (int w = 1; w <= 3; ++w) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); }
    return 0;
}
