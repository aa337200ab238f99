/* A plain-C host of librsa_hip.so: no Python, no torch -- only the HIP runtime C API and include/rsa.h.
 *
 *   rsa_host_demo <in.bin> <out.bin> B H S D top_k p_remain first_frame_blocks [fp8]
 *
 * in.bin  : q, k, v as raw bf16 [B,H,S,D] each, back to back;  out.bin : O as raw bf16 [B,S,H,D].
 * Layout: the Wan variant (visual tokens only, rectified_wan21_attn.py:297-313), no neighbour matrix.
 * This is what a non-Python integrator does: size the workspace, allocate, one call, one stream.
 * tests/test_gpu_c_host.py runs it and compares out.bin byte for byte with the Python binding's result. */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rsa.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_RSA(x) do { int s_ = (x); if (s_ != RSA_OK) { fprintf(stderr, "%s: %s (%s)\n", #x, rsa_status_string(s_), rsa_last_hip_error()); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 10) {
        fprintf(stderr, "usage: %s in.bin out.bin B H S D top_k p_remain first_frame_blocks [fp8]\n", argv[0]);
        return 1;
    }
    const int B = atoi(argv[3]), H = atoi(argv[4]), S = atoi(argv[5]), D = atoi(argv[6]), top_k = atoi(argv[7]);
    const float p = (float)atof(argv[8]);
    const int ffb = atoi(argv[9]);
    const int fp8 = argc > 10 && strcmp(argv[10], "fp8") == 0;
    const size_t n = (size_t)B * H * S * D, bytes = n * 2;

    rsa_layout lay;
    /* struct sizes and major.minor of the header this host was compiled against must be the library's (rsa_buffers grew in 0.5.0) */
    if (rsa_abi_check(RSA_HEADER_VERSION, sizeof(rsa_buffers), sizeof(rsa_layout)) != RSA_OK) {
        fprintf(stderr, "librsa_hip %d does not match the header this demo was built with (%d)\n", rsa_version(), RSA_HEADER_VERSION);
        return 1;
    }
    memset(&lay, 0, sizeof lay);
    lay.B = B; lay.H = H; lay.D = D; lay.S = S;
    lay.NB_total = (S + RSA_BLOCK - 1) / RSA_BLOCK;
    lay.NBv = lay.NB_total; lay.n_txt = 0; lay.kv_valid = S; lay.pool_valid = S;
    lay.text_end_block = lay.NB_total; lay.first_frame_blocks = ffb;
    lay.q_text_valid = 0; lay.kv_text_valid = S; lay.dtype = RSA_BF16;

    unsigned short* h_in = (unsigned short*)malloc(3 * bytes);
    unsigned short* h_out = (unsigned short*)malloc(bytes);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(h_in, 1, 3 * bytes, f) != 3 * bytes) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    fclose(f);

    size_t sizes[RSA_NUM_BUFFERS], total = 0, sizes8[4], total8 = 0;
    CHECK_RSA(rsa_buffer_bytes(&lay, sizes, &total));
    if (fp8) CHECK_RSA(rsa_fp8_operand_bytes(&lay, sizes8, &total8));

    void *d_qkv = NULL, *d_out = NULL, *d_ws = NULL, *d_ws8 = NULL;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_HIP(hipMalloc(&d_qkv, 3 * bytes));
    CHECK_HIP(hipMalloc(&d_out, bytes));
    CHECK_HIP(hipMalloc(&d_ws, total));            /* hipMalloc is 256-byte aligned */
    if (fp8) CHECK_HIP(hipMalloc(&d_ws8, total8));
    CHECK_HIP(hipMemcpyAsync(d_qkv, h_in, 3 * bytes, hipMemcpyHostToDevice, stream));

    rsa_tensor4 t[3];
    for (int i = 0; i < 3; ++i) {
        t[i].ptr = (const char*)d_qkv + i * bytes;
        t[i].stride_b = (int64_t)H * S * D; t[i].stride_h = (int64_t)S * D; t[i].stride_s = D;
    }
    rsa_out4 o;                                    /* [B, S, H, D]: strides of the b, h, s axes in elements */
    o.ptr = d_out; o.stride_b = (int64_t)S * H * D; o.stride_h = D; o.stride_s = (int64_t)H * D;

    if (fp8)
        CHECK_RSA(rsa_rectified_attention_fp8(&lay, t[0], t[1], t[2], NULL, top_k, p, d_ws, total, d_ws8, total8, o,
                                              stream));
    else
        CHECK_RSA(rsa_rectified_attention(&lay, t[0], t[1], t[2], NULL, top_k, p, d_ws, total, o, stream));
    CHECK_HIP(hipMemcpyAsync(h_out, d_out, bytes, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    f = fopen(argv[2], "wb");
    if (!f || fwrite(h_out, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    fclose(f);
    printf("rsa_host_demo: librsa_hip %d, %s K5, B=%d H=%d S=%d D=%d top_k=%d p=%g ffb=%d, workspace %zu B%s\n",
           rsa_version(), fp8 ? "fp8" : "bf16", B, H, S, D, top_k, p, ffb, total, fp8 ? " + fp8 images" : "");
    hipFree(d_qkv); hipFree(d_out); hipFree(d_ws); if (d_ws8) hipFree(d_ws8);
    free(h_in); free(h_out);
    return 0;
}
