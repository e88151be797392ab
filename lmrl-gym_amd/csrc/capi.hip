// capi.hip — library-wide C ABI pieces (error text, version, device probe).
#include <stdarg.h>
#include <vector>

#include "../../include/lmrl_amd.h"
#include "common.h"

namespace lmrl {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

unsigned g_prof_mask = 0;
namespace {
struct ProfRec { hipEvent_t a, b; int tag; double work; };
std::vector<ProfRec> g_recs;
std::vector<size_t> g_open[PROF_N_TAGS];
const char *kTagNames[PROF_N_TAGS] = {"gemm_bf16_128x128", "gemm_bf16_64x128", "gemm_bf16_64x64", "attention_decode",
                                      "attention_chunk", "lm_head_sample", "layernorm", "embed", "wordle_step", "wordle_reset",
                                      "wordle_tokens", "sample_reduce"};
}  // namespace
static unsigned long long *g_counters_d = nullptr;   // [PROF_N_TAGS]
unsigned long long *prof_byte_counter(int tag) {
    if (!((g_prof_mask >> tag) & 1u)) return nullptr;
    if (!g_counters_d) {
        if (hipMalloc(&g_counters_d, sizeof(unsigned long long) * PROF_N_TAGS) != hipSuccess) return nullptr;
        (void)hipMemset(g_counters_d, 0, sizeof(unsigned long long) * PROF_N_TAGS);
    }
    return g_counters_d + tag;
}
void prof_begin(int tag, hipStream_t s, double work) {
    ProfRec r;
    r.tag = tag; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    g_open[tag].push_back(g_recs.size() - 1);
}
bool prof_kernel_events(int tag, double work, hipEvent_t *start, hipEvent_t *stop) {
    if (!((g_prof_mask >> tag) & 1u)) return false;
    ProfRec r;
    r.tag = tag; r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return false;
    g_recs.push_back(r);
    *start = r.a; *stop = r.b;
    return true;
}
void prof_end(int tag, hipStream_t s) {
    if (g_open[tag].empty()) return;
    const size_t i = g_open[tag].back();
    g_open[tag].pop_back();
    (void)hipEventRecord(g_recs[i].b, s);
}
}  // namespace lmrl

extern "C" {

void lmrl_prof_enable(unsigned mask) { lmrl::g_prof_mask = mask; }

void lmrl_prof_reset(void) {
    for (auto &r : lmrl::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    lmrl::g_recs.clear();
    for (auto &o : lmrl::g_open) o.clear();
    if (lmrl::g_counters_d) (void)hipMemset(lmrl::g_counters_d, 0, sizeof(unsigned long long) * lmrl::PROF_N_TAGS);
}

int lmrl_prof_n_tags(void) { return lmrl::PROF_N_TAGS; }

const char *lmrl_prof_tag_name(int tag) { return (tag >= 0 && tag < lmrl::PROF_N_TAGS) ? lmrl::kTagNames[tag] : ""; }

/* Synchronises the device, then sums the recorded scopes of `tag`. */
int lmrl_prof_read(int tag, double *total_ms, double *total_work, long long *launches) {
    if (tag < 0 || tag >= lmrl::PROF_N_TAGS || !total_ms || !total_work || !launches) return LMRL_ERR_ARG;
    if (hipDeviceSynchronize() != hipSuccess) return LMRL_ERR_HIP;
    double ms = 0, w = 0; long long n = 0;
    for (auto &r : lmrl::g_recs) {
        if (r.tag != tag) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        ms += t; w += r.work; n++;
    }
    if (lmrl::g_counters_d) {   // data-dependent byte counts accumulated on the device
        unsigned long long c = 0;
        if (hipMemcpy(&c, lmrl::g_counters_d + tag, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess && c > 0) w = (double)c;
    }
    *total_ms = ms; *total_work = w; *launches = n;
    return LMRL_OK;
}

const char *lmrl_last_error(void) { return lmrl::g_err; }

int lmrl_version(void) { return 100; }

const char *lmrl_device_arch(void) {
    static char arch[256] = "";
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return "";
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return "";
    snprintf(arch, sizeof(arch), "%s", p.gcnArchName);
    return arch;
}
}
