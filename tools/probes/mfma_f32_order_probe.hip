// Does v_mfma_f32_32x32x2_f32 accumulate its two k-steps as sequential fused multiply-adds (k = 0, then k = 1)?
// If so an fp32 dot-product chain in k order can run on the matrix pipe bit-exactly (contract C4 of the statistics kernels).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A /*[32][2]*/, const float* B /*[2][32]*/, const float* C /*[32][32]*/, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], B[h * 32 + r], c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}
int main() {
    float hA[64], hB[64], hC[1024], hD[1024];
    srand(7);
    auto rnd = []() { return (float)((rand() % 20001) - 10000) / 3000.0f * (1.0f + (rand() % 1000) * 1e-3f); };
    int cnt[4] = {0, 0, 0, 0}, total = 0;
    float *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, 256); (void)hipMalloc(&dB, 256); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    for (int rep = 0; rep < 50; ++rep) {
        for (float& x : hA) x = rnd();
        for (float& x : hB) x = rnd();
        for (float& x : hC) x = rnd() * 3.0f;
        (void)hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, hC, 4096, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            const float a0 = hA[i * 2], a1 = hA[i * 2 + 1], b0 = hB[j], b1 = hB[32 + j], c = hC[i * 32 + j], d = hD[i * 32 + j];
            cnt[0] += d == fmaf(a1, b1, fmaf(a0, b0, c));                       // sequential fused, k ascending
            cnt[1] += d == fmaf(a0, b0, fmaf(a1, b1, c));                       // sequential fused, k descending
            cnt[2] += d == (float)((double)a0 * b0 + (double)a1 * b1 + (double)c);  // one rounding at the end
            cnt[3] += d == ((a0 * b0 + c) + a1 * b1);                           // unfused
            ++total;
        }
    }
    printf("of %d outputs: == fma(a1,b1,fma(a0,b0,c)) %d | == fma(a0,b0,fma(a1,b1,c)) %d | == single rounding %d | == unfused %d\n",
           total, cnt[0], cnt[1], cnt[2], cnt[3]);
    return 0;
}
