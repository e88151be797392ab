// mt19937.hip — kernels + C ABI for batched CPython-compatible MT19937 streams (see mt19937.h).
#include "../../include/lmrl_amd.h"
#include "common.h"
#include "mt19937.h"

namespace lmrl {

constexpr size_t kMtSeedLdsBytes = (size_t)kMtN * kMtSeedLanes * sizeof(uint32_t);   // 78 KiB

// init_genrand(19650218) table, uploaded once per process.
static uint32_t *g_table_d = nullptr;

int mt_table(const uint32_t **out) {
    if (!g_table_d) {
        uint32_t h[kMtN];
        mt_init_table(h);
        LMRL_CHECK_HIP(hipMalloc(&g_table_d, sizeof(h)));
        LMRL_CHECK_HIP(hipMemcpy(g_table_d, h, sizeof(h), hipMemcpyHostToDevice));
    }
    *out = g_table_d;
    return LMRL_OK;
}

static hipError_t mt_seed_attr();

// One thread per stream, 32 streams per workgroup with the MT state in LDS during seeding (mt19937.h).
__global__ __launch_bounds__(kMtSeedLanes) void mt_seed_kernel(void *mt, const uint64_t *seeds, const uint8_t *mask, const uint32_t *table, int n) {
    extern __shared__ __attribute__((aligned(16))) uint32_t mt_lds[];
    const int e = blockIdx.x * kMtSeedLanes + threadIdx.x;
    const bool on = e < n && (!mask || mask[e]);
    mt_seed_and_twist_lds(mt_lds, mt, n, e, on, on ? seeds[e] : 0ull, table);
}

static hipError_t mt_seed_attr() {
    static bool done = false;
    if (done) return hipSuccess;
    done = true;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&mt_seed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMtSeedLdsBytes);
}

__global__ void mt_stream_kernel(void *mt, uint32_t *out, int n_out, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    MtRef r = mt_ref(mt, n, e);
    for (int k = 0; k < n_out; k++) out[(size_t)k * n + e] = mt_next(r);
}

__global__ void mt_randbelow_kernel(void *mt, const uint32_t *bounds, uint32_t *out, int n_draws, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    MtRef r = mt_ref(mt, n, e);
    for (int k = 0; k < n_draws; k++) out[(size_t)k * n + e] = mt_randbelow(r, bounds[k]);
}

}  // namespace lmrl

using namespace lmrl;

extern "C" {

size_t lmrl_mt_bytes(int n) { return (size_t)(kMtN + 1) * (size_t)n * sizeof(uint32_t); }

int lmrl_mt_seed(void *mt_d, const uint64_t *seeds_d, const uint8_t *mask_d, int n, void *stream) {
    LMRL_REQUIRE(mt_d && seeds_d && n >= 0, "lmrl_mt_seed: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    const uint32_t *table;
    int rc = mt_table(&table);
    if (rc) return rc;
    LMRL_CHECK_HIP(mt_seed_attr());
    hipLaunchKernelGGL(mt_seed_kernel, dim3(ceil_div(n, kMtSeedLanes)), dim3(kMtSeedLanes), kMtSeedLdsBytes, as_stream(stream), mt_d, seeds_d, mask_d, table, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_mt_stream(void *mt_d, uint32_t *out_d, int n_out, int n, void *stream) {
    LMRL_REQUIRE(mt_d && out_d && n >= 0 && n_out >= 0, "lmrl_mt_stream: bad argument");
    if (n == 0 || n_out == 0) return LMRL_OK;
    hipLaunchKernelGGL(mt_stream_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), mt_d, out_d, n_out, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_mt_randbelow(void *mt_d, const uint32_t *bounds_d, uint32_t *out_d, int n_draws, int n, void *stream) {
    LMRL_REQUIRE(mt_d && bounds_d && out_d && n >= 0 && n_draws >= 0, "lmrl_mt_randbelow: bad argument");
    if (n == 0 || n_draws == 0) return LMRL_OK;
    hipLaunchKernelGGL(mt_randbelow_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), mt_d, bounds_d, out_d, n_draws, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
