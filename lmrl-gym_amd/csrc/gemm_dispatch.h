// gemm_dispatch.h — shape policy: which bf16 GEMM kernel (gemm_bf16.h 4-wave rings, gemm8_bf16.h 8-wave large tiles) runs a given
// (M, N, K, epilogue).  Every choice below is a measurement (profiles/r01_gemm_config_sweep.txt, profiles/r02_gemm8_bench.txt).
#pragma once
#include "gemm8_bf16.h"

namespace lmrl {

// LayerNorm-fused epilogues: the slot layout is 2 per 64-column tile, so BN is fixed to 64 and the forced sweep
// configurations / v1 kernels do not apply; otherwise the same shape policy as gemm_launch below.
extern int g_gemm_variant;
template <int EPI, int NQ>
inline hipError_t gemm_launch_ln_nq(const GemmArgs &g, hipStream_t s) {
    // same measured shape policy as gemm_launch below; variant 101 = the former all-2-stage decode policy (A/B hook)
    // M >= 2048 (prefill): the LN consumers (qkv, fc: wide N) run on the 8-wave 128x128 tile of gemm8_bf16.h, two workgroups per CU
    // (measured +15..19 % over the 4-wave 128x64 ring, profiles/r02_gemm8_bench.txt); the residual producers (N = d_model) stay on the ring
    if constexpr (EPI != EPI_RESID_F32_STATS && NQ > 0) {
        if (g.M >= 2048 && g.N % 128 == 0 && g_gemm_variant == 206) return gemm8_launch<128, 128, 4, 4, 2, EPI, NQ>(g, s);      // A/B: 16 waves x 2 WGs per CU
        if (g.M >= 2048 && g.N % 128 == 0 && g_gemm_variant != 105) return gemm8_launch<128, 128, 2, 4, 2, EPI, NQ>(g, s);
        // decode (M ~ 1024) LN consumers with a wide N (qkv 144 tiles, fc 192 tiles of 128x128): one 8-wave workgroup per CU moves half the
        // bytes per flop of the 64x64 ring and measured 12.5 vs 15.0 us (profiles/r02_gemm8_bench.txt); 3 slots: nothing else shares the LDS
        if (g.M >= 512 && g.N % 128 == 0 && g.K < 2048 && (long)((g.M + 127) / 128) * (g.N / 128) >= 128 && g_gemm_variant != 105) {
            // (paired K-steps — one barrier per two K-steps on a 4-slot ring, g8_mainloop_pair — are 5 % faster in isolation, 12.6 -> 11.9 us,
            //  and measured SLOWER inside the episode, 49.72 vs 49.22 ms on the same box: variant 107 keeps the A/B)
            if (g.K % 128 == 0 && g_gemm_variant == 107) return gemm8_launch<128, 128, 2, 4, 4, EPI, NQ, true>(g, s);
            // round 4: 16 waves (4 x 4, 32 x 32 per wave) = four waves per SIMD on the one workgroup a CU holds — more fragment reads per MFMA, but the
            // per-step latency chain of a single resident workgroup overlaps better: 48.43 -> 48.14 ms per episode in situ (variant 110 = the 8-wave form)
            if (g_gemm_variant == 110) return gemm8_launch<128, 128, 2, 4, 3, EPI, NQ>(g, s);
            if (g_gemm_variant == 117) return gemm8_launch<128, 128, 4, 4, 4, EPI, NQ>(g, s);      // A/B: 16 waves, 4 slots
            if (g_gemm_variant == 118) return gemm8_launch<128, 128, 4, 4, 2, EPI, NQ>(g, s);      // A/B: 16 waves, 2 slots
            return gemm8_launch<128, 128, 4, 4, 3, EPI, NQ>(g, s);
        }
    }
    if constexpr (EPI == EPI_RESID_F32_STATS) {      // tools/ab_rollout_variants.py (A/B hooks; the product path never sets a variant)
        if (g.M >= 2048 && g_gemm_variant == 201) return gemm8_launch<128, 128, 2, 4, 2, EPI, NQ>(g, s);
        if (g.M >= 2048 && g_gemm_variant == 203) return gemm_launch_glds<128, 64, 3, EPI, NQ>(g, s);
        if (g.M >= 2048 && g_gemm_variant == 204) return gemm8_launch<128, 64, 4, 2, 2, EPI, NQ>(g, s);      // 8 waves (32 x 32 per wave), 48 KB ring: 3 WGs per CU
        if (g.M >= 2048 && g_gemm_variant == 205) return gemm8_launch<128, 64, 4, 2, 3, EPI, NQ>(g, s);      // ... 72 KB ring: 2 WGs per CU
    }
    if (g.M >= 2048) return gemm_launch_glds<128, 64, 2, EPI, NQ>(g, s);
    // (decode residual producers, N = d_model, 192 tiles: the 8-wave 64x64 form of gemm8_bf16.h is ~10 % faster in isolation but measured
    //  SLOWER inside the episode — 14.6 vs 13.4 us — so they stay on the 4-wave ring; profiles/r02_gemm8_bench.txt)
    // round 4 (with the MFMA results in VGPRs, build.py): the LONG-K residual producer (fc2: 192 tiles x 48 K-steps) on 8 waves (4 x 2, 16 x 32 per wave)
    // and a 4-slot ring: 48.2 -> 47.85 ms per episode in situ; the short-K one (proj, 12 K-steps) gains nothing from it (48.15), 3 slots lose (49.06).
    // Variant 116 = the 4-wave ring for both (A/B hook)
    if (g.K >= 2048 && g.K % 128 == 0 && g_gemm_variant == 124) return gemm8_launch<64, 64, 4, 2, 4, EPI, NQ, true>(g, s);      // A/B (round 6): fc2 on paired K-steps (one barrier per two)
    if (g.K < 2048 && g.K % 128 == 0 && g_gemm_variant == 125) return gemm8_launch<64, 64, 4, 2, 4, EPI, NQ, true>(g, s);       // A/B: proj likewise
    if (g.K >= 2048 && g_gemm_variant == 120) return gemm8_launch<64, 64, 4, 2, 6, EPI, NQ>(g, s);      // A/B (round 6): 6-slot ring (96 KB), 5 stages in flight
    if (g.K >= 2048 && g_gemm_variant == 121) return gemm8_launch<64, 64, 4, 2, 5, EPI, NQ>(g, s);
    if (g.K >= 2048 && g_gemm_variant != 116) return gemm8_launch<64, 64, 4, 2, 4, EPI, NQ>(g, s);
    if (g.K >= 2048) return gemm_launch_glds<64, 64, 4, EPI, NQ>(g, s);
    if (g_gemm_variant == 101) return gemm_launch_glds<64, 64, 2, EPI, NQ>(g, s);
    const long tiles = (long)((g.M + 63) / 64) * (g.N / 64);
    if (tiles <= 256 && g_gemm_variant == 122) return gemm_launch_glds<64, 64, 6, EPI, NQ>(g, s);          // A/B (round 6): proj on a 6-slot ring
    if (tiles <= 256 && g_gemm_variant == 123) return gemm8_launch<64, 64, 4, 2, 6, EPI, NQ>(g, s);       // A/B: proj on the 8-wave kernel, 6 slots
    if (tiles <= 512) return gemm_launch_glds<64, 64, 4, EPI, NQ>(g, s);
    if (tiles <= 768) return gemm_launch_glds<64, 64, 3, EPI, NQ>(g, s);
    return gemm_launch_glds<64, 64, 2, EPI, NQ>(g, s);
}
// Slots per row are padded to 8*NQ (zero filled): NQ = 1 (d_model <= 256), 3 (768: GPT-2-small), 4 (1024: medium), 5 (1280: large).
inline int ln_fusion_nq(int d_model) {
    const int need = d_model / 64 * 2;
    return need <= 8 ? 1 : (need == 24 ? 3 : (need == 32 ? 4 : (need == 40 ? 5 : 0)));   // 0: no folded configuration -> stand-alone LayerNorm
}
template <int EPI>
inline hipError_t gemm_launch_ln(const GemmArgs &g, hipStream_t s) {
    static_assert(EPI == EPI_RESID_F32_STATS || EPI == EPI_BF16_LN || EPI == EPI_GELU_BF16_LN || EPI == EPI_BF16_LN_KV, "LN-fused epilogues only");
    if (EPI == EPI_RESID_F32_STATS) return gemm_launch_ln_nq<EPI, 0>(g, s);
    switch (g.nslots / 8) {
        case 1: return gemm_launch_ln_nq<EPI, 1>(g, s);
        case 3: return gemm_launch_ln_nq<EPI, 3>(g, s);
        case 4: return gemm_launch_ln_nq<EPI, 4>(g, s);
        case 5: return gemm_launch_ln_nq<EPI, 5>(g, s);
        default: return hipErrorInvalidValue;
    }
}

// Large-tile choice for the train step's products (M = B*T = 16 k rows, or the vocabulary; one 8-wave workgroup per CU for the 256-row tiles,
// two for 128x128).  Cost = whole rounds of the 256 CUs x tile area (a partly filled round costs a full tile time), from
// tools/gemm8_bench.hip `train` on one box (profiles/r03_train_gemm_tiles.txt): N = 768 (proj, fc2, every dX): 256x192 = 256 tiles = ONE round,
// 71.5 vs 90.4 us at K = 3072 (256x256: 192 tiles, a quarter of the CUs idle); N = 2304 (qkv): 768 tiles of 256x192 = 3 full rounds, 75.5 vs
// 92.4 (256x256, 2.25 rounds) / 82.2 us (128x128); N = 3072: a tie, 256x256 kept.  hipBLASLt on the same bf16 -> fp32 shapes:
// 71.8 / 77.4 / 97.0 us (profiles/r03_vs_hipblaslt_f32out.txt).
enum TrainTile { TT_NONE = 0, TT_256x256, TT_256x192, TT_128x128 };
inline TrainTile pick_train_tile(int M, int N, int K) {
    if (M < 2048 || N % 64 != 0) return TT_NONE;
    const long mt256 = (M + 255) / 256;
    double best = 1e300;
    TrainTile t = TT_NONE;
    if (M >= 4096 && N % 256 == 0) {
        const long tiles = mt256 * (N / 256);
        if (tiles >= 160) { best = (double)((tiles + 255) / 256) * 256 * 256; t = TT_256x256; }
    }
    if (M >= 4096 && N % 192 == 0) {
        const long tiles = mt256 * (N / 192);
        const double c = (double)((tiles + 255) / 256) * 256 * 192;
        if (tiles >= 160 && c < best) { best = c; t = TT_256x192; }
    }
    if (N % 128 == 0 && K < 2048) {
        const long tiles = (long)((M + 127) / 128) * (N / 128);
        const double c = (double)((tiles + 511) / 512) * 2 * 128 * 128 * 1.1;      // two co-resident workgroups per CU, ~10 % less efficient each
        if (c < best) { best = c; t = TT_128x128; }
    }
    return t;
}

// The train step's fused epilogues (gemm8_bf16.h only): EPI_F32_GELU_BF16, EPI_BF16_HEADS, EPI_GELU_BWD_BF16.  bf16 outputs store fragment
// PAIRS, so the 256x192 tile runs as 4 x 2 waves (64 x 96 per wave) there.
template <int EPI>
inline hipError_t gemm_launch_train(const GemmArgs &g, hipStream_t s) {
    static_assert(EPI == EPI_F32_GELU_BF16 || EPI == EPI_BF16_HEADS || EPI == EPI_GELU_BWD_BF16 || EPI == EPI_BF16_CE, "train-step epilogues only");
    switch (pick_train_tile(g.M, g.N, g.K)) {
        case TT_256x256: return gemm8_launch<256, 256, 2, 4, 2, EPI>(g, s);
        case TT_256x192: return gemm8_launch<256, 192, 4, 2, 2, EPI>(g, s);
        default: break;
    }
    if (g.N % 128 == 0) return gemm8_launch<128, 128, 2, 4, 2, EPI>(g, s);
    return hipErrorInvalidValue;
}

// Tile choice: keep >= ~1 workgroup per CU (256 CUs) when the problem allows it.
template <int EPI>
inline hipError_t gemm_launch(const GemmArgs &g, hipStream_t s) {
    const long t128 = (long)((g.M + 127) / 128) * (g.N / 128);
    if (g_gemm_variant == 1) {
        if (g.N % 128 == 0 && t128 >= 192) return gemm_launch_cfg<128, 128, EPI>(g, s);
        if (g.N % 128 == 0 && (long)((g.M + 63) / 64) * (g.N / 128) >= 192) return gemm_launch_cfg<64, 128, EPI>(g, s);
        return gemm_launch_cfg<64, 64, EPI>(g, s);
    }
    switch (g_gemm_variant) {   // forced configurations for tools/bench_gemm.py
        case 10: return gemm_launch_glds<128, 128, 2, EPI>(g, s);
        case 11: return gemm_launch_glds<128, 128, 3, EPI>(g, s);
        case 12: return gemm_launch_glds<128, 128, 4, EPI>(g, s);
        case 20: return gemm_launch_glds<128, 64, 2, EPI>(g, s);
        case 21: return gemm_launch_glds<128, 64, 3, EPI>(g, s);
        case 22: return gemm_launch_glds<128, 64, 4, EPI>(g, s);
        case 30: return gemm_launch_glds<64, 64, 2, EPI>(g, s);
        case 31: return gemm_launch_glds<64, 64, 3, EPI>(g, s);
        case 32: return gemm_launch_glds<64, 64, 4, EPI>(g, s);
        case 100: if (g.M < 2048) return gemm_launch_glds<128, 64, 2, EPI>(g, s); break;
        case 101: if (g.M < 2048) return gemm_launch_glds<64, 64, 2, EPI>(g, s); break;
        case 102: if (g.M < 2048) return g.K >= 2048 ? gemm_launch_glds<128, 64, 3, EPI>(g, s) : gemm_launch_glds<64, 64, 2, EPI>(g, s); break;
        case 103: if (g.M < 2048) return g.K >= 2048 ? gemm_launch_glds<64, 64, 3, EPI>(g, s) : gemm_launch_glds<64, 64, 2, EPI>(g, s); break;
        case 104: if (g.M >= 2048) return gemm_launch_glds<128, 128, 2, EPI>(g, s); break;
        case 40: if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) return gemm8_launch<128, 128, 2, 4, 2, EPI>(g, s); break;
        case 41: if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) return gemm8_launch<256, 128, 4, 2, 3, EPI>(g, s); break;
        case 42: if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) return gemm8_launch<128, 256, 2, 4, 3, EPI>(g, s); break;
        case 43: if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) return gemm8_launch<256, 128, 4, 2, 2, EPI>(g, s); break;
        case 44: if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) return gemm8_launch<256, 256, 2, 4, 2, EPI>(g, s); break;
        default: break;
    }
    // measured on MI355X (profiles/r01_gemm_config_sweep.txt + in-situ sweeps): the prefill GEMMs (M >= 2048, thousands of
    // tiles) want occupancy: 2-stage rings, 3 workgroups per CU.  The decode GEMMs (M = 1024) have only 192-768 tiles, i.e.
    // 1-3 per CU, and are bound by the latency chain of their K loop: as many stages as still leave every tile resident
    // at once (768 tiles: 3 stages = 48 KiB -> 3 WG/CU; 192 tiles: 4 stages = 64 KiB -> 2 WG/CU).
    if constexpr (EPI == EPI_F32 || EPI == EPI_GELU_SPLIT3) {
        // decode-sized products with a LONG K (the bf16x3 rollout mode: K' = 3 K = 2304 / 9216 at M = one row per env): the 8-wave 128 x 128 tile,
        // one workgroup per CU with a 3-slot ring (as the bf16 engine's decode qkv / fc), instead of the 4-wave 2-slot ring the K >= 2048 rule below picks
        if (g.M >= 512 && g.M < 2048 && g.N % 128 == 0 && g.K >= 2048 && t128 >= 128 && g_gemm_variant != 109)
            return g_gemm_variant == 110 ? gemm8_launch<128, 128, 2, 4, 3, EPI>(g, s) : gemm8_launch<128, 128, 4, 4, 3, EPI>(g, s);   // 16 waves: as the bf16 engine's decode qkv / fc
        if (g_gemm_variant != 108) {             // 108: the round-2 policy below (A/B hook)
            switch (pick_train_tile(g.M, g.N, g.K)) {
                case TT_256x256: return gemm8_launch<256, 256, 2, 4, 2, EPI>(g, s);
                case TT_256x192: return gemm8_launch<256, 192, 2, 4, 2, EPI>(g, s);
                case TT_128x128: return gemm8_launch<128, 128, 2, 4, 2, EPI>(g, s);
                default: break;
            }
        }
    }
    if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) {
        // fp32-output products of the train step's bf16-matmul mode (M = B*T = 16 k rows, or the vocabulary): 256x256 tiles halve the LDS
        // and L2 traffic per flop; taken when the tile count fills whole rounds of the 256 CUs (tools/bench_train_gemm.py, profiles/r02_train_gemm_sweep.txt:
        // fc / fc2 / dX / head products 95-1450 us vs 117-3690 us; the 576-tile qkv product stays on 128x128)
        const long t256 = (long)((g.M + 255) / 256) * (g.N / 256);
        if (g.M >= 4096 && g.N % 256 == 0 && t256 >= 160 && (g.K >= 2048 || t256 <= 256 || t256 % 256 == 0 || t256 >= 1024))
            return gemm8_launch<256, 256, 2, 4, 2, EPI>(g, s);
    }
    if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || EPI == EPI_F32 || EPI == EPI_GELU_SPLIT3) {
        if (g.M >= 2048 && g.N % 128 == 0 && g.K < 2048) return gemm8_launch<128, 128, 2, 4, 2, EPI>(g, s);   // wide prefill GEMMs: 8-wave 128x128 tiles
    }
    if constexpr (EPI == EPI_RESID_F32) {      // the train forward's projection + residual add (separate residual operand): same tile as its EPI_F32 form
        if (g.resid && g.M >= 2048 && g.N % 128 == 0 && g.K < 2048) return gemm8_launch<128, 128, 2, 4, 2, EPI>(g, s);
    }
    if (g.M >= 2048) return gemm_launch_glds<128, 64, 2, EPI>(g, s);
    if (g.K >= 2048) {
        // dW products (K = B*T): more 64x64 tiles than can be co-resident (512) -> 128x128 tiles in one round (dw fc: 165 vs 228 us)
        if (g.N % 128 == 0 && (long)((g.M + 63) / 64) * (g.N / 64) > 512) return gemm_launch_glds<128, 128, 2, EPI>(g, s);
        return gemm_launch_glds<64, 64, 4, EPI>(g, s);
    }
    const long tiles = (long)((g.M + 63) / 64) * (g.N / 64);
    if (tiles <= 512) return gemm_launch_glds<64, 64, 4, EPI>(g, s);      // 64 KiB ring -> 2 WG/CU -> 512 resident tiles
    if (tiles <= 768) return gemm_launch_glds<64, 64, 3, EPI>(g, s);      // 48 KiB ring -> 3 WG/CU -> 768 resident tiles
    return gemm_launch_glds<64, 64, 2, EPI>(g, s);
}

}  // namespace lmrl
