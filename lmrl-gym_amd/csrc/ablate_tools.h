// TOOLS-ONLY.  Included by gpt2.hip only under -DLMRL_TOOLS (`python lmrl-gym_amd/build.py --tools` -> liblmrl_amd_tools.so); the product library
// liblmrl_amd.so neither defines nor honours any of this and rejects the flag bits.  tools/bench_ablate_decode.py is the only user.
#pragma once
#include <hip/hip_runtime.h>
/* flag bits 16-25: TIMING-ONLY ablation of the single-token decode layers (tools/bench_ablate_decode.py): the named launch is left out, results are
 * garbage — measures, inside the real dependent chain, what removing or hiding that launch could buy at most.  Never set by the package. */
#define LMRL_FWD_ABLATE_SHIFT 16
#define LMRL_ABLATE_QKV 1u
#define LMRL_ABLATE_ATTN 2u
#define LMRL_ABLATE_PROJ 4u
#define LMRL_ABLATE_FC 8u
#define LMRL_ABLATE_FC2 16u
#define LMRL_ABLATE_FC2_SPLITK2 64u  /* fc2 as two concurrent half-K launches (two streams, racy): split-K = 2 without its seam */
#define LMRL_ABLATE_FC2_HALFK 128u   /* fc2 over half of K only: the cost of a K loop of half the length */
#define LMRL_ABLATE_PROJ_AUX_SERIAL 256u /* calibration of the two above: proj on the aux stream but AFTER the attention (same chain): the cost of a fork / join */
#define LMRL_ABLATE_ATTN_HALF_BYTES 512u /* the decode attention reads only the first half of the cached positions (half the cache lines): an optimistic bound on what a 1-byte cache element could buy (reading half of every 128-B row instead changes nothing: the line is fetched whole) */
#define LMRL_ABLATE_FC2_SEAM3 1024u /* fc2 as ONE split-K = 3 launch (64 x 64 tiles, 8 waves) + a plain reduce launch: what the split costs WITH its seam (the reduce lacks the bf16 / stats stores: a lower bound) */
#define LMRL_ABLATE_FC2_SEAM2 2048u /* the same with split-K = 2 */
#define LMRL_ABLATE_FC2_SEAM6 4096u /* the same on 128 x 128 tiles, split-K = 6 */
#define LMRL_ABLATE_PROJ_CONCURRENT 32u /* proj GEMM on an auxiliary stream behind the qkv GEMM only: concurrent with the attention launch (stale data) */

namespace lmrl {
struct AblateAux { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
}
