// v_cvt_pk_u8_f32 on gfx950: rounding / saturation of the float -> u8 conversion, and its issue cost beside v_exp_f32.
// build: hipcc --offload-arch=gfx950 -O2 -o cvt_pk_u8_probe cvt_pk_u8_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>

__global__ void conv(const float* x, unsigned* y, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    unsigned w = 0xAABBCCDDu;
    w = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0u, w);
    unsigned w2 = 0xAABBCCDDu;
    w2 = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 2u, w2);
    y[2 * i] = w;
    y[2 * i + 1] = w2;
}

template <int WHICH>
__global__ void rate(float* out, long long* cyc) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = 0.001f * (threadIdx.x + i);
    unsigned w[4] = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if constexpr (WHICH == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            else if constexpr (WHICH == 1) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(w[i & 3]) : "v"(v[i]));
            else asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i]));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[threadIdx.x] = s + (float)(w[0] + w[1] + w[2] + w[3]);
    if (threadIdx.x == 0) cyc[WHICH] = t1 - t0;
}

int main() {
    const float xs[] = {0.0f, 0.49f, 0.5f, 0.51f, 0.99f, 1.0f, 1.49f, 1.5f, 1.51f, 2.5f, 3.5f, 126.5f, 254.4f, 254.5f, 255.0f, 255.6f,
                        300.0f, 1e9f, -0.4f, -0.6f, -1.0f, -1e9f, INFINITY, -INFINITY, NAN};
    const int n = sizeof(xs) / sizeof(xs[0]);
    float* dx; unsigned* dy;
    hipMalloc(&dx, sizeof(xs)); hipMalloc(&dy, 2 * n * sizeof(unsigned));
    hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
    conv<<<1, 64>>>(dx, dy, n);
    unsigned y[2 * 64];
    hipMemcpy(y, dy, 2 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("x = %12g -> u8 %3u   (byte 0 word %08x, byte 2 word %08x)\n", xs[i], y[2 * i] & 0xff, y[2 * i], y[2 * i + 1]);
    float* dout; long long* dc;
    hipMalloc(&dout, 64 * sizeof(float)); hipMalloc(&dc, 4 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) {
        rate<0><<<1, 64>>>(dout, dc); rate<1><<<1, 64>>>(dout, dc); rate<2><<<1, 64>>>(dout, dc);
    }
    long long c[4];
    hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
    printf("issue cost over 4096 instructions (s_memtime ticks at 100 MHz -> relative only): v_exp_f32 %lld, v_cvt_pk_u8_f32 %lld, v_add_f32 %lld\n",
           c[0], c[1], c[2]);
    return 0;
}
