// Second look at what a staging piece costs a wave that is alone on its SIMD among MFMAs (round 4).  The first probe
// (dma_issue_probe.hip) charged the piece with everything its modes added to the loop - a taken branch, a counted wait and the
// scalar address updates; this one separates them.  Loop trip = 16 x v_mfma_f32_32x32x16_bf16 (512 matrix cycles), no branch
// but the loop's own; the source window is wrapped with s_and, not with a branch.
//   0: MFMAs only                               1: + the scalar address updates of 4 pieces (no memory instruction)
//   2: + 4 x global_load_lds_dwordx4 (K5's density: one 1-KiB piece per 4 MFMAs), vmcnt(16) once per trip
//   3: as 2 without the vmcnt wait in the loop (throttled by the hardware counter only)
//   4: + 16 x global_load_lds_dword (the same bytes as 256-B rows, one behind every MFMA), vmcnt(48) once per trip
//   5: + 8 x global_load_lds_dwordx4 (twice K5's density)
//   6: as 2 with the four pieces back to back behind MFMA 0
//   7: as 2 with s_nop 0 between m0 write and the load replaced by placing the m0 write one MFMA earlier
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue_probe2 dma_issue_probe2.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

#define MF(acc) ".if %c[mode] != 8\n\tv_mfma_f32_32x32x16_bf16 v[" acc "], v[32:35], v[32:35], v[" acc "]\n\t.endif\n\t"
#define ADDR "s_add_u32 s84, s84, 0x6000\n\ts_and_b32 s84, s84, s85\n\ts_add_u32 s80, s86, s84\n\ts_addc_u32 s81, s87, 0\n\t"
#define X4(off) "s_add_u32 m0, s82, " off "\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], s[80:81]\n\t"
#define X1(off, v) "s_add_u32 m0, s82, " off "\n\ts_nop 0\n\tglobal_load_lds_dword " v ", s[80:81]\n\t"

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const unsigned char* src, unsigned long long* out,
                                                                                         int iters, unsigned stride, unsigned mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned vo = (lane >> 4) * stride + (lane & 15) * 16;
    const unsigned v1a = lane * 4, v1b = stride + lane * 4, v1c = 2 * stride + lane * 4, v1d = 3 * stride + lane * 4;
    const unsigned char* base = src + ((size_t)(blockIdx.x * 4 + wv) * 1048576);
    unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base), ghi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32));
    unsigned ld = __builtin_amdgcn_readfirstlane(lds_base + wv * 16384);
    unsigned long long t0, t1;
    asm volatile(
        "v_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\t"
        "s_mov_b32 s86, %[glo]\n\ts_mov_b32 s87, %[ghi]\n\ts_mov_b32 s80, %[glo]\n\ts_mov_b32 s81, %[ghi]\n\ts_mov_b32 s82, %[ld]\n\ts_mov_b32 s83, %[it]\n\ts_mov_b32 s84, 0\n\ts_mov_b32 s85, %[mask]\n\t"
        "s_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
        ".Lp_%=:\n\t"
        // ---- group 0
        MF("0:15")
        ".if %c[mode] == 1\n\t" ADDR ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 3 || %c[mode] == 5 || %c[mode] == 7 || %c[mode] == 8\n\t" ADDR X4("0") ".endif\n\t"
        ".if %c[mode] == 6\n\t" ADDR X4("0") ADDR X4("1024") ADDR X4("2048") ADDR X4("3072") ".endif\n\t"
        ".if %c[mode] == 4\n\t" ADDR X1("0", "%[v1a]") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("256", "%[v1b]") ".endif\n\t"
        MF("0:15")
        ".if %c[mode] == 4\n\t" X1("512", "%[v1c]") ".endif\n\t"
        ".if %c[mode] == 5 || %c[mode] == 8\n\t" ADDR X4("4096") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("768", "%[v1d]") ".endif\n\t"
        // ---- group 1
        MF("0:15")
        ".if %c[mode] == 1\n\t" ADDR ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 3 || %c[mode] == 5 || %c[mode] == 7 || %c[mode] == 8\n\t" ADDR X4("1024") ".endif\n\t"
        ".if %c[mode] == 4\n\t" ADDR X1("1024", "%[v1a]") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("1280", "%[v1b]") ".endif\n\t"
        MF("0:15")
        ".if %c[mode] == 4\n\t" X1("1536", "%[v1c]") ".endif\n\t"
        ".if %c[mode] == 5 || %c[mode] == 8\n\t" ADDR X4("5120") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("1792", "%[v1d]") ".endif\n\t"
        // ---- group 2
        MF("0:15")
        ".if %c[mode] == 1\n\t" ADDR ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 3 || %c[mode] == 5 || %c[mode] == 7 || %c[mode] == 8\n\t" ADDR X4("2048") ".endif\n\t"
        ".if %c[mode] == 4\n\t" ADDR X1("2048", "%[v1a]") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("2304", "%[v1b]") ".endif\n\t"
        MF("0:15")
        ".if %c[mode] == 4\n\t" X1("2560", "%[v1c]") ".endif\n\t"
        ".if %c[mode] == 5 || %c[mode] == 8\n\t" ADDR X4("6144") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("2816", "%[v1d]") ".endif\n\t"
        // ---- group 3
        MF("0:15")
        ".if %c[mode] == 1\n\t" ADDR ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 3 || %c[mode] == 5 || %c[mode] == 7 || %c[mode] == 8\n\t" ADDR X4("3072") ".endif\n\t"
        ".if %c[mode] == 4\n\t" ADDR X1("3072", "%[v1a]") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("3328", "%[v1b]") ".endif\n\t"
        MF("0:15")
        ".if %c[mode] == 4\n\t" X1("3584", "%[v1c]") ".endif\n\t"
        ".if %c[mode] == 5 || %c[mode] == 8\n\t" ADDR X4("7168") ".endif\n\t"
        MF("16:31")
        ".if %c[mode] == 4\n\t" X1("3840", "%[v1d]") ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 6 || %c[mode] == 7\n\ts_waitcnt vmcnt(16)\n\t.endif\n\t"
        ".if %c[mode] == 5 || %c[mode] == 8\n\ts_waitcnt vmcnt(32)\n\t.endif\n\t"
        ".if %c[mode] == 4\n\ts_waitcnt vmcnt(48)\n\t.endif\n\t"
        "s_sub_u32 s83, s83, 1\n\ts_cmp_lg_u32 s83, 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
        : [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [glo] "s"(glo), [ghi] "s"(ghi), [ld] "s"(ld), [it] "s"(iters), [mask] "s"(mask), [vo] "v"(vo), [v1a] "v"(v1a), [v1b] "v"(v1b), [v1c] "v"(v1c), [v1d] "v"(v1d), [mode] "i"(MODE)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "a255",
          "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "memory", "scc");
    if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int MODE>
static void run(const unsigned char* src, unsigned long long* dout, int iters, const char* what, unsigned mask, int kib) {
    std::vector<unsigned long long> h(1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 65536, 0, src, dout, iters, 6144u, mask);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h.data(), dout, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("mode %d window %4u KiB: cycles per trip: median %7.1f  p10 %7.1f  p90 %7.1f | %6.3f ms, %5.2f TB/s, ~%4.0f MHz | %s [%s]\n", MODE,
           (mask + 1) >> 10, (double)h[512] / iters, (double)h[102] / iters, (double)h[921] / iters, ms,
           1024.0 * iters * kib * 1024.0 / (ms * 1e-3) / 1e12, (double)h[512] / (ms * 1e3), what, hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned char* src; unsigned long long* dout;
    (void)hipMalloc(&src, (size_t)1100 << 20);
    (void)hipMemset(src, 1, (size_t)1100 << 20);
    (void)hipMalloc(&dout, 1024 * 8);
    const int iters = 2000;
    for (unsigned mask : {0u, 0x1ffffu, 0x7ffffu}) {   // 16 KiB touched per wave (2 MiB per XCD: L2-resident) / 64 KiB per wave (64 MiB in all: L2 misses, Infinity-Cache resident)
        run<0>(src, dout, iters, "MFMAs only", mask, 0);
        run<1>(src, dout, iters, "+ scalar address updates of 4 pieces", mask, 0);
        run<2>(src, dout, iters, "+ 4 x global_load_lds_dwordx4, vmcnt(16) per trip", mask, 4);
        run<3>(src, dout, iters, "+ 4 x global_load_lds_dwordx4, no wait", mask, 4);
        run<6>(src, dout, iters, "+ 4 x global_load_lds_dwordx4 back to back", mask, 4);
        run<5>(src, dout, iters, "+ 8 x global_load_lds_dwordx4", mask, 8);
        run<4>(src, dout, iters, "+ 16 x global_load_lds_dword (256-B rows), vmcnt(48) per trip", mask, 4);
        run<8>(src, dout, iters, "NO MFMAs, 8 x global_load_lds_dwordx4 per trip (bandwidth of the pattern)", mask, 8);
    }
    return 0;
}
