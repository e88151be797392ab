// common.h — shared host-side helpers for liblmrl_amd.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace lmrl {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define LMRL_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::lmrl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LMRL_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define LMRL_CHECK_LAUNCH()  LMRL_CHECK_HIP(hipGetLastError())

#define LMRL_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            ::lmrl::set_error("%s:%d: %s", __FILE__, __LINE__, msg); \
            return LMRL_ERR_ARG;                                  \
        }                                                         \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

// ---- optional per-kernel-class HIP-event timing (bench.py's live roofline numbers). Off by default: a scope costs
// two hipEventRecord calls on the launch stream only when its tag is enabled through lmrl_prof_enable().
enum ProfTag {
    PROF_GEMM_128x128 = 0, PROF_GEMM_64x128, PROF_GEMM_64x64, PROF_ATTN_DECODE, PROF_ATTN_CHUNK, PROF_LM_HEAD_SAMPLE,
    PROF_LAYERNORM, PROF_EMBED, PROF_WORDLE_STEP, PROF_WORDLE_RESET, PROF_TOKENS, PROF_SAMPLE_REDUCE, PROF_N_TAGS
};
extern unsigned g_prof_mask;
void prof_begin(int tag, hipStream_t s, double work);
void prof_end(int tag, hipStream_t s);
// a start/stop event pair for hipExtLaunchKernelGGL (timestamps the kernel itself, like rocprofv3's kernel trace, instead of
// bracketing the launch on the stream); false when the tag is disabled
bool prof_kernel_events(int tag, double work, hipEvent_t *start, hipEvent_t *stop);
// device counter (bytes) for kernels whose algorithmic traffic is data dependent; null unless the tag is enabled
unsigned long long *prof_byte_counter(int tag);
struct ProfScope {
    int tag; hipStream_t s; bool on;
    ProfScope(int t, hipStream_t st, double work) : tag(t), s(st), on((g_prof_mask >> t) & 1u) { if (on) prof_begin(tag, s, work); }
    ~ProfScope() { if (on) prof_end(tag, s); }
};

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace lmrl
