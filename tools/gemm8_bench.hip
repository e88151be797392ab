// tools/gemm8_bench.hip — stand-alone A/B of the GEMM kernels on the rollout's shapes (no torch): the 4-wave global_load_lds ring of
// gemm_bf16.h vs the 8-wave large-tile kernel of gemm8_bf16.h, interleaved rounds in one process, cycling 12 weight copies (cold L2
// like consecutive layers), every configuration checked against a naive fp32-accumulate kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off tools/gemm8_bench.hip -o tools/_bin/gemm8_bench
#define LMRL_G8_PROBE 1
#include "../lmrl-gym_amd/csrc/gemm8_bf16.h"
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#include <algorithm>

namespace lmrl {   // the few symbols of capi.hip / gpt2.hip the headers reference
unsigned g_prof_mask = 0;
int g_gemm_variant = 0;
void set_error(const char *, ...) {}
void prof_begin(int, hipStream_t, double) {}
void prof_end(int, hipStream_t) {}
}  // namespace lmrl
using namespace lmrl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void ref_gemm(const uint16_t *A, const uint16_t *W, const float *bias, float *C, int M, int N, int K) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), m = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; k++) s = fmaf(bf16_to_f32(A[(size_t)m * K + k]), bf16_to_f32(W[(size_t)n * K + k]), s);
    C[(size_t)m * N + n] = s + bias[n];
}
__global__ void fill(uint16_t *p, size_t n, uint32_t seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = f32_to_bf16_rn(((float)(x & 0xffff) / 32768.f - 1.f) * scale);     // uniform [-scale, scale): full-range random operands
    }
}
__global__ void fillf(float *p, size_t n, uint32_t seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
    }
}
__global__ void cmp_f32(const float *C, const float *R, size_t n, float *maxerr) {
    float e = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        e = fmaxf(e, fabsf(C[i] - R[i]) / (1.f + fabsf(R[i])));
    atomicMax(reinterpret_cast<int *>(maxerr), __float_as_int(e));
}
__global__ void cmp_bf16(const uint16_t *C, const float *R, size_t n, float *maxerr) {
    float e = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        e = fmaxf(e, fabsf(bf16_to_f32(C[i]) - R[i]) / (1.f + fabsf(R[i])));
    atomicMax(reinterpret_cast<int *>(maxerr), __float_as_int(e));
}

struct Cfg { std::string name; std::function<hipError_t(const GemmArgs &, hipStream_t)> run; int bm, bn; bool f32out = false; bool ln = false; };

int main(int argc, char **argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "quick";
    struct Shape { int M, N, K; const char *what; };
    std::vector<Shape> shapes = {{7168, 2304, 768, "prefill qkv"}, {7168, 768, 768, "prefill proj"}, {7168, 3072, 768, "prefill fc"},
                                 {7168, 768, 3072, "prefill fc2"}, {1024, 50432, 768, "lm head"}, {1024, 2304, 768, "decode qkv"},
                                 {1024, 3072, 768, "decode fc"}, {1024, 768, 3072, "decode fc2"}, {4096, 4096, 4096, "4096^3"}};
    if (quick) { shapes = {shapes[0], shapes[2], shapes[4]}; }
    if (argc > 1 && std::string(argv[1]) == "decode") { shapes = {{1024, 2304, 768, "decode qkv"}, {1024, 768, 768, "decode proj"}, {1024, 3072, 768, "decode fc"}, {1024, 768, 3072, "decode fc2"}}; }
    if (argc > 1 && std::string(argv[1]) == "resid") { shapes = {{1024, 768, 768, "decode proj"}, {1024, 768, 3072, "decode fc2"}, {7168, 768, 768, "prefill proj"}, {7168, 768, 3072, "prefill fc2"}}; }
    const bool train = argc > 1 && std::string(argv[1]) == "train";
    if (train) { shapes = {{16384, 3072, 768, "train fc"}, {16384, 768, 3072, "train fc2"}, {16384, 2304, 768, "train qkv"}, {16384, 768, 768, "train proj"}, {8192, 50432, 768, "train head"}, {16384, 768, 2304, "train dx qkv"}, {8192, 768, 50304, "train dx head"}}; }
    std::vector<Cfg> cfgs_train = {
        {"g8 256x256 2x4 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 256, 2, 4, 2, EPI_F32>(g, s); }, 256, 256, true},
        {"g8 256x256 2x2 s2 f32out (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 256, 2, 2, 2, EPI_F32>(g, s); }, 256, 256, true},
        {"g8 256x256 4x2 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 256, 4, 2, 2, EPI_F32>(g, s); }, 256, 256, true},
        {"g8 256x192 2x4 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 192, 2, 4, 2, EPI_F32>(g, s); }, 256, 192, true},
        {"g8 256x192 4x2 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 192, 4, 2, 2, EPI_F32>(g, s); }, 256, 192, true},
        {"g8 128x192 2x4 s3 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 192, 2, 4, 3, EPI_F32>(g, s); }, 128, 192, true},
        {"g8 256x128 2x2 s3 f32out (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 128, 2, 2, 3, EPI_F32>(g, s); }, 256, 128, true},
        {"g8 128x256 2x2 s3 f32out (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 256, 2, 2, 3, EPI_F32>(g, s); }, 128, 256, true},
        {"g8 128x128 2x4 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_F32>(g, s); }, 128, 128, true},
    };
    std::vector<Cfg> cfgs = {
        {"ring 128x64 s2 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 64, 2, EPI_BF16>(g, s); }, 128, 64},
        {"ring 64x64 s3 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 3, EPI_BF16>(g, s); }, 64, 64},
        {"ring 128x128 s2 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 128, 2, EPI_BF16>(g, s); }, 128, 128},
        {"g8 256x256 2x4 s2", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 256, 2, 4, 2, EPI_BF16>(g, s); }, 256, 256},
        {"g8 256x128 2x4 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 128, 2, 4, 3, EPI_BF16>(g, s); }, 256, 128},
        {"g8 256x128 4x2 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 128, 4, 2, 3, EPI_BF16>(g, s); }, 256, 128},
        {"g8 128x256 2x4 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 256, 2, 4, 3, EPI_BF16>(g, s); }, 128, 256},
        {"g8 128x128 2x4 s2", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 2x4 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 3, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 2x4 PAIR", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 4, EPI_BF16, 0, true>(g, s); }, 128, 128},
        {"g8 128x128 2x4 PAIR BF16_LN", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 4, EPI_BF16_LN, 3, true>(g, s); }, 128, 128, false, true},
        {"g8 128x128 2x4 PAIR GELU_LN", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 4, EPI_GELU_BF16_LN, 3, true>(g, s); }, 128, 128, false, true},
        {"g8 64x128 2x4 PAIR", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 128, 2, 4, 4, EPI_BF16, 0, true>(g, s); }, 64, 128},
        {"g8 128x128 4x4 s2 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 2, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 4x4 s3 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 3, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 4x4 s4 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 4, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 4x4 s3 BF16_LN", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 3, EPI_BF16_LN, 3>(g, s); }, 128, 128, false, true},
        {"g8 64x128 2x4 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 128, 2, 4, 4, EPI_BF16>(g, s); }, 64, 128},
        {"g8 64x64 4x2 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 4, EPI_BF16>(g, s); }, 64, 64},
        {"WD 64x64 4x2 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 4, EPI_BF16, 0, false, true>(g, s); }, 64, 64},
        {"WD 64x64 4x2 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 3, EPI_BF16, 0, false, true>(g, s); }, 64, 64},
        {"WD 64x64 2x2 s4 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 2, 2, 4, EPI_BF16, 0, false, true>(g, s); }, 64, 64},
        {"WD 64x64 1x4 s4 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 1, 4, 4, EPI_BF16, 0, false, true>(g, s); }, 64, 64},
        {"WD 64x64 2x4 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 2, 4, 4, EPI_BF16, 0, false, true>(g, s); }, 64, 64},
        {"WD 128x128 4x4 s3 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 3, EPI_BF16, 0, false, true>(g, s); }, 128, 128},
        {"WD 128x128 4x4 s4 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 4, 4, 4, EPI_BF16, 0, false, true>(g, s); }, 128, 128},
        {"WD 128x128 2x4 s3", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 3, EPI_BF16, 0, false, true>(g, s); }, 128, 128},
        {"WD 128x128 2x8 s4 (16w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 8, 4, EPI_BF16, 0, false, true>(g, s); }, 128, 128},
        {"WD 128x64 4x2 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 64, 4, 2, 4, EPI_BF16, 0, false, true>(g, s); }, 128, 64},
        {"WD 64x128 2x4 s4", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 128, 2, 4, 4, EPI_BF16, 0, false, true>(g, s); }, 64, 128},
        {"g8 128x128 2x2 s3 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 2, 3, EPI_BF16>(g, s); }, 128, 128},
        {"g8 128x128 2x4 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_F32>(g, s); }, 128, 128, true},
        {"g8 256x256 2x4 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<256, 256, 2, 4, 2, EPI_F32>(g, s); }, 256, 256, true},
        {"ring 128x64 s2 f32out", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 64, 2, EPI_F32>(g, s); }, 128, 64, true},
        {"ring 128x64 s2 GELU_LN", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 64, 2, EPI_GELU_BF16_LN, 3>(g, s); }, 128, 64, false, true},
        {"g8 128x128 2x4 s2 GELU_LN", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_GELU_BF16_LN, 3>(g, s); }, 128, 128, false, true},
        {"ring 128x64 s2 BF16_LN", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 64, 2, EPI_BF16_LN, 3>(g, s); }, 128, 64, false, true},
        {"g8 128x128 2x4 s2 BF16_LN", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_BF16_LN, 3>(g, s); }, 128, 128, false, true},
        {"ring 64x64 s4 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 4, EPI_BF16>(g, s); }, 64, 64},
        {"ring 64x64 s6 (4w)", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 6, EPI_BF16>(g, s); }, 64, 64},
        {"ring 64x64 s4 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 4, EPI_RESID_F32_STATS>(g, s); }, 64, 64, false, true},
        {"ring 64x64 s3 GELU_LN", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 3, EPI_GELU_BF16_LN, 3>(g, s); }, 64, 64, false, true},
        {"g8 64x64 4x2 s3 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 3, EPI_RESID_F32_STATS>(g, s); }, 64, 64, false, true},
        {"g8 64x64 4x2 s4 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 4, EPI_RESID_F32_STATS>(g, s); }, 64, 64, false, true},
        {"g8 64x64 4x2 s6 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 64, 4, 2, 6, EPI_RESID_F32_STATS>(g, s); }, 64, 64, false, true},
        {"g8 64x128 2x4 s4 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<64, 128, 2, 4, 4, EPI_RESID_F32_STATS>(g, s); }, 64, 128, false, true},
        {"g8 128x64 4x2 s4 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 64, 4, 2, 4, EPI_RESID_F32_STATS>(g, s); }, 128, 64, false, true},
        {"ring 64x64 s6 RESID_STATS", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<64, 64, 6, EPI_RESID_F32_STATS>(g, s); }, 64, 64, false, true},
        {"ring 128x64 s2 GELU", [](const GemmArgs &g, hipStream_t s) { return gemm_launch_glds<128, 64, 2, EPI_GELU_BF16>(g, s); }, 128, 64, false, true},
        {"g8 128x128 2x4 s2 GELU", [](const GemmArgs &g, hipStream_t s) { return gemm8_launch<128, 128, 2, 4, 2, EPI_GELU_BF16>(g, s); }, 128, 128, false, true},
    };
    if (train) cfgs = cfgs_train;
    if (const char *f = getenv("LMRL_CFG")) {            // LMRL_CFG="WD,g8 64x64": only configurations whose name contains one of the comma-separated pieces
        std::vector<std::string> keys; std::string cur;
        for (const char *p = f;; p++) { if (*p == ',' || !*p) { if (!cur.empty()) keys.push_back(cur); cur.clear(); if (!*p) break; } else cur += *p; }
        std::vector<Cfg> kept;
        for (auto &c : cfgs) for (auto &k : keys) if (c.name.find(k) != std::string::npos) { kept.push_back(c); break; }
        cfgs = kept;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    for (const Shape &sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, NWC = getenv("LMRL_NWC") ? atoi(getenv("LMRL_NWC")) : (N > 10000 ? 2 : 12);   // LMRL_NWC=1: hot L2
        uint16_t *A, *W, *C; float *bias, *R, *err;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)NWC * N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&R, (size_t)M * N * 4)); CK(hipMalloc(&err, 4));
        fill<<<1024, 256, 0, st>>>(A, (size_t)M * K, 1u, 1.0f);
        fill<<<1024, 256, 0, st>>>(W, (size_t)NWC * N * K, 2u, 0.05f);
        fillf<<<64, 256, 0, st>>>(bias, (size_t)N, 3u);
        float *stats, *colsum;
        CK(hipMalloc(&stats, (size_t)M * 24 * 8)); CK(hipMalloc(&colsum, (size_t)N * 4));
        fillf<<<1024, 256, 0, st>>>(stats, (size_t)M * 48, 5u); fillf<<<64, 256, 0, st>>>(colsum, (size_t)N, 6u);
        ref_gemm<<<dim3((N + 63) / 64, (M + 3) / 4), 256, 0, st>>>(A, W, bias, R, M, N, K);
        CK(hipStreamSynchronize(st));
        printf("%-13s M=%5d N=%5d K=%5d\n", sh.what, M, N, K);
        std::vector<std::vector<float>> us(cfgs.size());
        std::vector<float> errs(cfgs.size(), -1.f);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rnd = 0; rnd < 3; rnd++) {
            for (size_t c = 0; c < cfgs.size(); c++) {
                if (N % cfgs[c].bn) continue;
                GemmArgs g{A, W, bias, C, M, N, K, K, N, N};
                if (cfgs[c].ln) {
                    if (K != 768 && cfgs[c].name.find("RESID") == std::string::npos) continue;
                    if (cfgs[c].name.find("RESID") != std::string::npos) { if (N != 768) continue; g.xb = reinterpret_cast<uint16_t *>(R); }
                    g.stats = reinterpret_cast<float2 *>(stats); g.colsum = colsum; g.nslots = 24; g.inv_d = 1.f / 768.f; g.eps = 1e-5f;
                }
                if (rnd == 0 && cfgs[c].ln) errs[c] = 0.f;
                if (rnd == 0 && !cfgs[c].ln) {
                    CK(hipMemsetAsync(C, 0, (size_t)M * N * 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
                    CK(cfgs[c].run(g, st));
                    if (cfgs[c].f32out) cmp_f32<<<512, 256, 0, st>>>(reinterpret_cast<float *>(C), R, (size_t)M * N, err);
                    else cmp_bf16<<<512, 256, 0, st>>>(C, R, (size_t)M * N, err);
                    CK(hipMemcpyAsync(&errs[c], err, 4, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
                }
                for (int w = 0; w < 2; w++) CK(cfgs[c].run(g, st));
                const int n = 20;
                CK(hipEventRecord(e0, st));
                for (int it = 0; it < n; it++) { g.W = W + (size_t)(it % NWC) * N * K; CK(cfgs[c].run(g, st)); }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us[c].push_back(ms * 1e3f / n);
            }
        }
        // per-workgroup timeline of the g8 configurations (one extra launch each): cycles from the first workgroup's entry
        for (size_t c = 0; c < cfgs.size(); c++) {
            if (us[c].empty()) continue;
            const size_t nwg = 8 * 4096;
            unsigned long long *pb; CK(hipMalloc(&pb, nwg * 4 * 8)); CK(hipMemsetAsync(pb, 0, nwg * 4 * 8, st));
            CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g8_probe), &pb, sizeof(pb), 0, hipMemcpyHostToDevice, st));
            GemmArgs g{A, W + (size_t)(NWC - 1) * N * K, bias, C, M, N, K, K, N, N};
            if (cfgs[c].ln) { g.stats = reinterpret_cast<float2 *>(stats); g.colsum = colsum; g.nslots = 24; g.inv_d = 1.f / 768.f; g.eps = 1e-5f; g.xb = reinterpret_cast<uint16_t *>(R); }
            CK(cfgs[c].run(g, st));
            std::vector<unsigned long long> h(nwg * 4);
            CK(hipMemcpyAsync(h.data(), pb, nwg * 4 * 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            unsigned long long *nul = nullptr;
            CK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g8_probe), &nul, sizeof(nul), 0, hipMemcpyHostToDevice, st)); CK(hipStreamSynchronize(st));
            unsigned long long t00 = ~0ull, tend = 0; double a[3] = {0, 0, 0}, start_avg = 0, start_max = 0; int cnt = 0;
            for (size_t w = 0; w < nwg; w++) if (h[w * 4]) t00 = std::min(t00, h[w * 4]);
            for (size_t w = 0; w < nwg; w++) {
                if (!h[w * 4] || !h[w * 4 + 3]) continue;
                cnt++; tend = std::max(tend, h[w * 4 + 3]);
                start_avg += (double)(h[w * 4] - t00); start_max = std::max(start_max, (double)(h[w * 4] - t00));
                for (int k = 0; k < 3; k++) a[k] += (double)(h[w * 4 + k + 1] - h[w * 4 + k]);
            }
            if (cnt) printf("   [timeline %-22s] %4d WGs: start avg %7.0f max %7.0f | prologue %7.0f | K loop %7.0f | epilogue %7.0f | kernel %7.0f cycles\n",
                            cfgs[c].name.c_str(), cnt, start_avg / cnt, start_max, a[0] / cnt, a[1] / cnt, a[2] / cnt, (double)(tend - t00));
            CK(hipFree(pb));
        }
        for (size_t c = 0; c < cfgs.size(); c++) {
            if (us[c].empty()) continue;
            std::sort(us[c].begin(), us[c].end());
            const int tiles = ((M + cfgs[c].bm - 1) / cfgs[c].bm) * (N / cfgs[c].bn);
            printf("   %-24s %8.1f us (med %8.1f)  %7.0f TF   tiles %5d   relerr %.4f%s\n", cfgs[c].name.c_str(), us[c][0], us[c][us[c].size() / 2],
                   2.0 * M * N * K / us[c][0] / 1e6, tiles, errs[c], errs[c] > 0.02f ? "   <-- MISMATCH" : "");
        }
        fflush(stdout);
        CK(hipFree(stats)); CK(hipFree(colsum)); CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C)); CK(hipFree(bias)); CK(hipFree(R)); CK(hipFree(err));
    }
    return 0;
}
