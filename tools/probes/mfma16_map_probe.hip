// Probe the lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 with A = fp4 (e2m1), B = fp8 (e4m3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
// mode 0: A ones everywhere, B ones only in lane L.  mode 1: B ones everywhere, A ones only in lane L.
__global__ void probe(int mode, int L, float* out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    const bool a1 = mode == 0 || l == L, b1 = mode == 1 || l == L;
    for (int j = 0; j < 8; ++j) { a[j] = a1 ? 0x22222222 : 0; b[j] = b1 ? 0x38383838 : 0; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = c[i];
}
int main() {
    float* dO; (void)hipMalloc(&dO, 1024);
    float hO[256];
    for (int mode = 0; mode < 2; ++mode)
        for (int L = 0; L < 64; L += (L < 4 ? 1 : 15)) {
            probe<<<1, 64>>>(mode, L, dO);
            (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
            printf("mode %d L %2d: nonzero (lane:reg=value):", mode, L);
            int n = 0;
            for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i)
                if (hO[l * 4 + i] != 0.f) { if (n < 10) printf(" %d:%d=%g", l, i, hO[l * 4 + i]); ++n; }
            printf("  [%d nonzero]\n", n);
        }
    return 0;
}
