// Overlap of k-sets: A (fp4 or fp8) lane La ones x B (fp8) lane Lb ones -> C[row La&15][col Lb&15] = #common k
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int FMT_A>
__global__ void probe(int La, int Lb, int jlo, int jhi, float* out) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (l == La) ? (FMT_A == 4 ? 0x22222222 : 0x38383838) : 0;
        b[j] = (l == Lb && j >= jlo && j < jhi) ? 0x38383838 : 0;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT_A, 0, 0, 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i];
    out[l] = s;
}
int main() {
    float* dO; (void)hipMalloc(&dO, 256);
    float hO[64];
    for (int fmt = 0; fmt <= 4; fmt += 4) {
        printf("A format %s: overlap table rows La = 0,16,32,48 x cols Lb = 0,16,32,48 (full B lane)\n", fmt ? "fp4" : "fp8");
        for (int ia = 0; ia < 4; ++ia) {
            for (int ib = 0; ib < 4; ++ib) {
                if (fmt) probe<4><<<1, 64>>>(16 * ia, 16 * ib, 0, 8, dO); else probe<0><<<1, 64>>>(16 * ia, 16 * ib, 0, 8, dO);
                (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
                float t = 0; for (int l = 0; l < 64; ++l) t += hO[l];
                printf(" %3g", t);
            }
            printf("\n");
        }
        printf("  B lane 0 dword ranges vs A lane 0/16/32/48:\n");
        for (int d = 0; d < 8; d += 2) {
            printf("   B dwords [%d,%d):", d, d + 2);
            for (int ia = 0; ia < 4; ++ia) {
                if (fmt) probe<4><<<1, 64>>>(16 * ia, 0, d, d + 2, dO); else probe<0><<<1, 64>>>(16 * ia, 0, d, d + 2, dO);
                (void)hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost);
                float t = 0; for (int l = 0; l < 64; ++l) t += hO[l];
                printf(" %3g", t);
            }
            printf("\n");
        }
    }
    return 0;
}
