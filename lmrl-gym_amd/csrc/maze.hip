// maze.hip — batched Maze state-transition kernels for gfx950.
//
// Replaces MazeEnv.reset / MazeEnv.step / update_position / the three reward functions of
// llm_rl_scripts/maze/env/env.py:104-131,161-214.  The MDP state is 5 ints per env, so the
// mapping is one LANE per env over struct-of-arrays state (coalesced 4-byte accesses); the
// wall grid (<= a few hundred bytes) is read through L1/L2.  Observation TEXT is rendered on
// the host from (position, goal, wall bits) — see lmrl-gym_amd/envs/maze.py.
#include "../../include/lmrl_amd.h"
#include "common.h"
#include "mt19937.h"

namespace lmrl {

struct MazeCtx {
    uint8_t *grid_d = nullptr;   // [R][C], 1 = wall
    int32_t *free_d = nullptr;   // [n_free][2] row-major np.argwhere(maze == 0)  (env.py:195)
    int32_t *goals_d = nullptr;  // [n_goals][2]
    int R = 0, C = 0, n_free = 0, n_goals = 0, max_steps = -1;
    float rew[3] = {0.f, -4.f, -1.f};
};

enum { M_POSR = 0, M_POSC = 1, M_GOALR = 2, M_GOALC = 3, M_STEPS = 4 };

__device__ __forceinline__ bool is_wall(const uint8_t *grid, int R, int C, int r, int c) {
    if (r < 0 || r >= R || c < 0 || c >= C) return true;
    return grid[r * C + c] == 1;
}

__global__ void maze_reset_kernel(const int32_t *free_cells, int n_free, const int32_t *goals, int n_goals, int32_t *st,
                                  void *mt, const uint64_t *seeds, const int32_t *opt_goal, const int32_t *opt_init,
                                  const uint8_t *mask, const uint32_t *table, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (mask && !mask[e]) return;
    const bool has_goal = opt_goal && opt_goal[2 * e] >= 0;
    const bool has_init = opt_init && opt_init[2 * e] >= 0;
    MtRef r = mt_ref(mt, n, e);
    if (!(has_goal && has_init)) {   // random_state.reset(seed): random.seed(seed)   (env.py:187, randomness.py:9-14)
        mt_seed(r, seeds[e], table);
        mt_twist(r);
        r.idx[e] = 0;
    }
    int gr, gc;
    if (has_goal) {
        gr = opt_goal[2 * e];
        gc = opt_goal[2 * e + 1];
    } else {  // random.choice(self.valid_goals)   (env.py:193)
        uint32_t k = mt_randbelow(r, (uint32_t)n_goals);
        gr = goals[2 * k];
        gc = goals[2 * k + 1];
    }
    int pr, pc;
    if (has_init) {
        pr = opt_init[2 * e];
        pc = opt_init[2 * e + 1];
    } else {  // positions = argwhere(maze == 0); positions.remove(goal); random.choice(positions)   (env.py:195-202)
        int gidx = -1;
        for (int k = 0; k < n_free; k++)
            if (free_cells[2 * k] == gr && free_cells[2 * k + 1] == gc) { gidx = k; break; }
        const int m = gidx >= 0 ? n_free - 1 : n_free;
        uint32_t k = mt_randbelow(r, (uint32_t)m);
        if (gidx >= 0 && (int)k >= gidx) k++;
        pr = free_cells[2 * k];
        pc = free_cells[2 * k + 1];
    }
    st[(size_t)M_POSR * n + e] = pr;
    st[(size_t)M_POSC * n + e] = pc;
    st[(size_t)M_GOALR * n + e] = gr;
    st[(size_t)M_GOALC * n + e] = gc;
    st[(size_t)M_STEPS * n + e] = 0;
}

__device__ __forceinline__ uint8_t wall_bits(const uint8_t *grid, int R, int C, int r, int c) {
    // order of delta_descriptions: right, left, above, below   (env.py:27, 59, 74)
    return (uint8_t)((is_wall(grid, R, C, r, c + 1) ? 1 : 0) | (is_wall(grid, R, C, r, c - 1) ? 2 : 0) |
                     (is_wall(grid, R, C, r - 1, c) ? 4 : 0) | (is_wall(grid, R, C, r + 1, c) ? 8 : 0));
}

__global__ void maze_step_kernel(const uint8_t *grid, int R, int C, int max_steps, float r_goal, float r_illegal,
                                 float r_else, int32_t *st, const uint8_t *action, const uint8_t *active, float *reward,
                                 uint8_t *done, uint8_t *kind, uint8_t *walls, int n) {
    int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    if (active && !active[e]) return;
    int pr = st[(size_t)M_POSR * n + e], pc = st[(size_t)M_POSC * n + e];
    const int gr = st[(size_t)M_GOALR * n + e], gc = st[(size_t)M_GOALC * n + e];
    int steps = st[(size_t)M_STEPS * n + e];
    if (max_steps >= 0 && steps >= max_steps) {   // env.py:164-165
        reward[e] = -1.0f;
        done[e] = 1;
        kind[e] = LMRL_MAZE_KIND_FAILURE;
        walls[e] = wall_bits(grid, R, C, pr, pc);
        return;
    }
    const uint8_t a = action[e];
    if (a < 4) {   // update_position (env.py:104-107), deltas env.py:94-99
        const int dr = (a == LMRL_MAZE_UP) ? -1 : (a == LMRL_MAZE_DOWN) ? 1 : 0;
        const int dc = (a == LMRL_MAZE_LEFT) ? -1 : (a == LMRL_MAZE_RIGHT) ? 1 : 0;
        if (!is_wall(grid, R, C, pr + dr, pc + dc)) {
            pr += dr;
            pc += dc;
        }
    }
    const bool at_goal = (pr == gr && pc == gc);
    const float rew = at_goal ? r_goal : (a >= 4 ? r_illegal : r_else);   // env.py:109-131
    st[(size_t)M_POSR * n + e] = pr;
    st[(size_t)M_POSC * n + e] = pc;
    reward[e] = rew;
    walls[e] = wall_bits(grid, R, C, pr, pc);
    if (at_goal) {   // env.py:173-174
        done[e] = 1;
        kind[e] = LMRL_MAZE_KIND_SUCCESS;
        return;
    }
    st[(size_t)M_STEPS * n + e] = steps + 1;   // env.py:177
    done[e] = 0;
    kind[e] = (a >= 4) ? LMRL_MAZE_KIND_OBS_ONLY : LMRL_MAZE_KIND_OBS;   // env.py:179-184
}

int mt_table(const uint32_t **out);  // mt19937.hip

}  // namespace lmrl

using namespace lmrl;

struct lmrl_maze_ctx : public MazeCtx {};

extern "C" {

lmrl_maze_ctx *lmrl_maze_create(const uint8_t *grid, int rows, int cols, const int32_t *valid_goals, int n_goals,
                                int max_steps, const float rewards[3]) {
    if (!grid || rows <= 0 || cols <= 0 || !valid_goals || n_goals <= 0 || !rewards) {
        set_error("lmrl_maze_create: bad argument");
        return nullptr;
    }
    for (int k = 0; k < n_goals; k++) {   // assert all maze[goal] == 0   (env.py:144)
        int r = valid_goals[2 * k], c = valid_goals[2 * k + 1];
        if (r < 0 || r >= rows || c < 0 || c >= cols || grid[r * cols + c] != 0) {
            set_error("lmrl_maze_create: goal %d is not a free cell", k);
            return nullptr;
        }
    }
    int n_free = 0;
    int32_t *free_h = new int32_t[2 * (size_t)rows * cols];
    for (int r = 0; r < rows; r++)
        for (int c = 0; c < cols; c++)
            if (grid[r * cols + c] == 0) {
                free_h[2 * n_free] = r;
                free_h[2 * n_free + 1] = c;
                n_free++;
            }
    lmrl_maze_ctx *ctx = new lmrl_maze_ctx();
    ctx->R = rows; ctx->C = cols; ctx->n_free = n_free; ctx->n_goals = n_goals; ctx->max_steps = max_steps;
    for (int k = 0; k < 3; k++) ctx->rew[k] = rewards[k];
    bool ok = hipMalloc(&ctx->grid_d, (size_t)rows * cols) == hipSuccess &&
              hipMalloc(&ctx->free_d, sizeof(int32_t) * 2 * (n_free > 0 ? n_free : 1)) == hipSuccess &&
              hipMalloc(&ctx->goals_d, sizeof(int32_t) * 2 * n_goals) == hipSuccess &&
              hipMemcpy(ctx->grid_d, grid, (size_t)rows * cols, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->free_d, free_h, sizeof(int32_t) * 2 * n_free, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(ctx->goals_d, valid_goals, sizeof(int32_t) * 2 * n_goals, hipMemcpyHostToDevice) == hipSuccess;
    delete[] free_h;
    if (!ok) {
        set_error("lmrl_maze_create: device allocation/copy failed (is a GPU visible?)");
        lmrl_maze_destroy(ctx);
        return nullptr;
    }
    return ctx;
}

void lmrl_maze_destroy(lmrl_maze_ctx *ctx) {
    if (!ctx) return;
    if (ctx->grid_d) (void)hipFree(ctx->grid_d);
    if (ctx->free_d) (void)hipFree(ctx->free_d);
    if (ctx->goals_d) (void)hipFree(ctx->goals_d);
    delete ctx;
}

size_t lmrl_maze_state_bytes(int n) { return (size_t)5 * (size_t)n * sizeof(int32_t); }

int lmrl_maze_reset(lmrl_maze_ctx *ctx, void *state_d, void *mt_d, const uint64_t *seeds_d, const int32_t *opt_goal_d,
                    const int32_t *opt_init_d, const uint8_t *mask_d, int n, void *stream) {
    LMRL_REQUIRE(ctx && state_d && mt_d && seeds_d && n >= 0, "lmrl_maze_reset: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    const uint32_t *table;
    int rc = mt_table(&table);
    if (rc) return rc;
    hipLaunchKernelGGL(maze_reset_kernel, dim3(ceil_div(n, 64)), dim3(64), 0, as_stream(stream), ctx->free_d, ctx->n_free,
                       ctx->goals_d, ctx->n_goals, (int32_t *)state_d, mt_d, seeds_d, opt_goal_d, opt_init_d, mask_d,
                       table, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}

int lmrl_maze_step(lmrl_maze_ctx *ctx, void *state_d, const uint8_t *action_d, const uint8_t *active_d, float *reward_d,
                   uint8_t *done_d, uint8_t *kind_d, uint8_t *walls_d, int n, void *stream) {
    LMRL_REQUIRE(ctx && state_d && action_d && reward_d && done_d && kind_d && walls_d && n >= 0,
                 "lmrl_maze_step: null pointer or negative n");
    if (n == 0) return LMRL_OK;
    hipLaunchKernelGGL(maze_step_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, as_stream(stream), ctx->grid_d, ctx->R,
                       ctx->C, ctx->max_steps, ctx->rew[0], ctx->rew[1], ctx->rew[2], (int32_t *)state_d, action_d,
                       active_d, reward_d, done_d, kind_d, walls_d, n);
    LMRL_CHECK_LAUNCH();
    return LMRL_OK;
}
}
