// What one 1-KiB staging piece costs a wave that is alone on its SIMD and busy with MFMAs (round 4, K5 64-row form):
// per iteration 4 x v_mfma_f32_32x32x16_bf16 (128 matrix cycles) + ONE piece in one of several forms; cycles per iteration by
// s_memtime, median over waves.  256 workgroups x 4 waves (one per SIMD: 512-register kernel), source = a 256 MiB buffer walked
// with a 6 KiB row stride (Infinity-Cache / HBM traffic like K5's).
//   0: MFMAs only                         1: global_load_lds_dwordx4 (saddr + 32-bit lane offset), vmcnt(16) throttle
//   2: global_load_dwordx4 -> VGPR ring   3: as 2 + ds_write_b128 of the piece loaded 16 iterations ago
//   4: as 1 with contiguous lane offsets  5: as 2 into the ACCUMULATOR file, ds_write from there
//   6: the piece as FOUR global_load_lds_dword (256 B = one key row each), one behind each MFMA
//   7: ONE global_load_lds_dword per iteration (256 B)      8: TWO per iteration, behind MFMA 0 and MFMA 2
//   9: four global_load_lds_dword back to back behind MFMA 1
// build: hipcc --offload-arch=gfx950 -O3 -o dma_issue_probe dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const unsigned char* src, unsigned long long* out,
                                                                                         int iters, unsigned stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // per-lane source offset: 4 rows x 16 chunks, row stride `stride` (a swizzled gather like K5's) or contiguous
    const unsigned vo = MODE == 4 ? lane * 16 : (lane >> 4) * stride + (lane & 15) * 16;
    const unsigned v1a = lane * 4, v1b = stride + lane * 4, v1c = 2 * stride + lane * 4, v1d = 3 * stride + lane * 4;   // one 256-B row per instruction
    const unsigned char* base = src + ((size_t)(blockIdx.x * 4 + wv) * 65536);
    unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)(size_t)base), ghi = __builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32));
    unsigned ld = __builtin_amdgcn_readfirstlane(lds_base + wv * 16384);
    unsigned ldv = lds_base + wv * 16384 + lane * 16;
    unsigned long long t0, t1;
    asm volatile(
        "v_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 1.0\n\tv_mov_b32 v34, 1.0\n\tv_mov_b32 v35, 1.0\n\t"
        "s_mov_b32 s80, %[glo]\n\ts_mov_b32 s81, %[ghi]\n\ts_mov_b32 s82, %[ld]\n\ts_mov_b32 s83, %[it]\n\ts_mov_b32 s84, 0\n\t"
        "s_memtime %[t0]\n\ts_waitcnt lgkmcnt(0)\n\t"
        ".Lp_%=:\n\t"
        "v_mfma_f32_32x32x16_bf16 v[0:15], v[32:35], v[32:35], v[0:15]\n\t"
        ".if %c[mode] == 6 || %c[mode] == 7 || %c[mode] == 8\n\t"
        "s_mov_b32 m0, s82\n\tglobal_load_lds_dword %[v1a], s[80:81]\n\t"
        ".endif\n\t"
        "v_mfma_f32_32x32x16_bf16 v[16:31], v[32:35], v[32:35], v[16:31]\n\t"
        ".if %c[mode] == 6\n\t"
        "s_add_u32 m0, s82, 256\n\tglobal_load_lds_dword %[v1b], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 9\n\t"
        "s_mov_b32 m0, s82\n\tglobal_load_lds_dword %[v1a], s[80:81]\n\t"
        "s_add_u32 m0, s82, 256\n\tglobal_load_lds_dword %[v1b], s[80:81]\n\t"
        "s_add_u32 m0, s82, 512\n\tglobal_load_lds_dword %[v1c], s[80:81]\n\t"
        "s_add_u32 m0, s82, 768\n\tglobal_load_lds_dword %[v1d], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 1 || %c[mode] == 4\n\t"
        "s_mov_b32 m0, s82\n\ts_add_u32 s84, s84, 1\n\tglobal_load_lds_dwordx4 %[vo], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 2 || %c[mode] == 3\n\t"
        "global_load_dwordx4 v[40:43], %[vo], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 5\n\t"
        "global_load_dwordx4 a[40:43], %[vo], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] != 0\n\t"
        "s_add_u32 s80, s80, 0x6000\n\ts_addc_u32 s81, s81, 0\n\t"
        ".endif\n\t"
        "v_mfma_f32_32x32x16_bf16 v[0:15], v[32:35], v[32:35], v[0:15]\n\t"
        ".if %c[mode] == 6 || %c[mode] == 8\n\t"
        "s_add_u32 m0, s82, 512\n\tglobal_load_lds_dword %[v1c], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 3\n\t"
        "ds_write_b128 %[ldv], v[44:47]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 5\n\t"
        "ds_write_b128 %[ldv], a[44:47]\n\t"
        ".endif\n\t"
        "v_mfma_f32_32x32x16_bf16 v[16:31], v[32:35], v[32:35], v[16:31]\n\t"
        ".if %c[mode] == 6\n\t"
        "s_add_u32 m0, s82, 768\n\tglobal_load_lds_dword %[v1d], s[80:81]\n\t"
        ".endif\n\t"
        ".if %c[mode] == 6 || %c[mode] == 9\n\t"
        "s_waitcnt vmcnt(48)\n\t"
        ".endif\n\t"
        ".if %c[mode] != 0 && %c[mode] != 6 && %c[mode] != 9\n\t"
        "s_waitcnt vmcnt(16)\n\t"
        ".endif\n\t"
        ".if %c[mode] != 0\n\t"
        "s_and_b32 s85, s83, 15\n\ts_cmp_eq_u32 s85, 0\n\ts_cbranch_scc0 .Lq_%=\n\t"
        "s_mov_b32 s80, %[glo]\n\ts_mov_b32 s81, %[ghi]\n\t"     // rewind every 16 pieces (a 384 KiB window per wave, 64 MiB in all: Infinity-Cache resident like one head of K, V)
        ".Lq_%=:\n\t"
        ".endif\n\t"
        "s_sub_u32 s83, s83, 1\n\ts_cmp_lg_u32 s83, 0\n\ts_cbranch_scc1 .Lp_%=\n\t"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
        "s_memtime %[t1]\n\ts_waitcnt lgkmcnt(0)\n\t"
        : [t0] "=&s"(t0), [t1] "=&s"(t1)
        : [glo] "s"(glo), [ghi] "s"(ghi), [ld] "s"(ld), [it] "s"(iters), [vo] "v"(vo), [ldv] "v"(ldv), [v1a] "v"(v1a), [v1b] "v"(v1b), [v1c] "v"(v1c), [v1d] "v"(v1d), [mode] "i"(MODE)
        : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v40", "v41", "v42",
          "v43", "v44", "v45", "v46", "v47", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a255", "s80", "s81", "s82", "s83", "s84", "s85", "memory");
    if (lane == 0) out[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int MODE>
static void run(const unsigned char* src, unsigned long long* dout, int iters) {
    std::vector<unsigned long long> h(1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 65536, 0, src, dout, iters, 6144u);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), dout, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("mode %d: cycles per iteration (4 MFMAs = 128 matrix cycles + one piece): median %.1f  p10 %.1f  p90 %.1f   [%s]\n", MODE,
           (double)h[512] / iters, (double)h[102] / iters, (double)h[921] / iters, hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned char* src; unsigned long long* dout;
    hipMalloc(&src, (size_t)1100 << 20);
    hipMemset(src, 1, (size_t)1100 << 20);
    hipMalloc(&dout, 1024 * 8);
    const int iters = 4000;
    run<0>(src, dout, iters); run<1>(src, dout, iters); run<4>(src, dout, iters); run<2>(src, dout, iters); run<3>(src, dout, iters); run<5>(src, dout, iters);
    run<7>(src, dout, iters); run<8>(src, dout, iters); run<6>(src, dout, iters); run<9>(src, dout, iters);
    return 0;
}
