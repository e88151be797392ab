"""Lock-step Maze rollouts entirely on the device: one-item histories (last_k = 1, below) and item windows (last_k > 1: `MazeRolloutEngine`'s
docstring; both with the online PPO scripts' chains built from the device record, `ppo_records`).

The MI355X counterpart of `interact_environment(maze_env, GPT2PPOPolicy(...), bsize=B)` (LLM_RL/environment.py:154-207 with
ppo/gpt2/interface.py:507-546) on the reference's Maze harness (maze/bc/fully_observed_bc.py:230-283: `last_k=1`, `max_steps=100`).
With last_k = 1 the policy's prompt is the observation of the current cell — a pure function of (goal, cell)
(maze/env/env.py:8-81) — so the host renders and tokenises every distinct observation ONCE per (maze, tokenizer), and prefills each of
them ONCE per set of weights into a prompt-prefix cache (one KV row per observation).  By default the envs then only POINT at their row
(`lmrl_gpt2_kv_attach`): the decode attention reads the prompt positions from the cache row itself (`lmrl_gpt2_forward_prefixed`), envs
standing on the same cell share those bytes through L2, and only the generated tokens' rows are written per env;
`prefix_indexed=False` copies the rows per env per turn instead (`lmrl_gpt2_kv_gather`, bit-identical).  A turn is then

    obs row <- env state | K/V + last hidden <- prefix cache | generate <= max_new ids | ids -> action code | lmrl_maze_step | record

with no host round trip; one turn is one hipGraph replay.  The reference re-encodes the text and re-runs the whole prompt through the
model for every env on every turn.
"""
from __future__ import annotations

import ctypes
from typing import Callable, List, Optional

import numpy as np

from . import _lib
from .envs import maze as M
from .gpt2 import FWD_RAGGED_ALWAYS, FWD_SKINNY, GPT2Engine, SampleParams

_TOK_BYTES = 16


class _CTraj(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("pos", "gen", "gen_len", "action", "reward", "kind", "n_turns", "live", "ep_reward", "obs_idx",
                                               "out_tok", "out_len", "gen_active", "act", "stepping")]


def maze_out_str_process(x: str) -> str:
    """The Maze scripts' `out_str_process` (e.g. maze/bc/fully_observed_bc.py:265): exactly one trailing newline."""
    return x.removesuffix("\n") + "\n"


def _decode_one(tokenizer, i: int) -> str:
    try:
        return tokenizer.decode([i], skip_special_tokens=True)
    except TypeError:
        special = getattr(tokenizer, "all_special_ids", None)
        if special is None:
            special = [t for t in (getattr(tokenizer, "pad_token_id", None),) if t is not None]
        return "" if i in special else tokenizer.decode([i])


def token_byte_table(tokenizer, vocab: int):
    """(bytes uint8 [vocab][16], blen uint8 [vocab]) for `lmrl_maze_tok_create`: the decoded string of every id.  blen 0: the id adds
    nothing to the text (special tokens under skip_special_tokens=True; ids the tokenizer decodes to ''), 255: non-ASCII or longer than
    16 bytes — such a token cannot occur in an action string, and for byte-level BPE (decode = concatenation of token bytes, invalid
    UTF-8 -> U+FFFD) any text containing it is not an action.  Ids beyond len(tokenizer) (padded model vocabularies) are 255."""
    tb = np.zeros((vocab, _TOK_BYTES), dtype=np.uint8)
    bl = np.full(vocab, 255, dtype=np.uint8)
    for i in range(min(vocab, len(tokenizer))):
        s = _decode_one(tokenizer, i)
        if s == "":
            bl[i] = 0
        elif s.isascii() and len(s) <= _TOK_BYTES:
            b = s.encode("ascii")
            tb[i, : len(b)] = np.frombuffer(b, dtype=np.uint8)
            bl[i] = len(b)
    return tb, bl


class _CHist(ctypes.Structure):            # lmrl_maze_hist (include/lmrl_amd.h)
    _fields_ = [(n, ctypes.c_void_p) for n in ("hist", "item_off", "n_items", "feed_start", "feed_len", "cache_len", "base", "prompt_len", "win_floor", "flags")] + \
               [("hcap", ctypes.c_int32), ("max_items", ctypes.c_int32)]


class MazeRolloutEngine:
    """B lock-step Maze episodes driven by a GPT-2 policy on one GPU.

    env.last_k == 1 (the fully observed scripts): the prompt is a function of (goal, cell) — prompt table + prefix cache, below.
    env.last_k > 1 (round 6; partially_observed_bc.py:241 runs last_k = 40): the prompt is the text of the item window
    `(history + [action] + [observation])[-last_k:]` (maze/env/env.py:182-184), left-truncated to `max_input_length` tokens.  The engine keeps a
    persistent per-env KV cache: while the window only grows ("append" turns) just the action's unforwarded tail and the new observation go through
    the model; once it slides or is truncated every position shifts (GPT-2's absolute position embeddings) and the window is forwarded again from
    position 0 ("re-prefill" turns).  Needs a tokenizer whose encoding of a concatenation is the concatenation of the encodings (byte level).

    `value_engine` + `q1_head` (+ `q2_head`) + `beta` make it the ILQL value policy (value_rl_base/gpt2/generation.py:97-119), as in
    `WordleRolloutEngine`.  `prefix_cache=False` prefills the observation tokens per env per turn instead (16-token chunks): the
    cross-check of the cache, and the fallback for tables too large to prefill; `prefix_indexed=False` keeps the cache but copies its rows
    into every env's own cache each turn (the cross-check of the indexed attention)."""

    def __init__(self, engine: GPT2Engine, tokenizer, env: M.MazeEnv, batch: int, max_new_tokens: int = 8, eos_token_id: Optional[int] = None,
                 max_input_length: int = 256, in_str_process: Optional[Callable[[str], str]] = None, prefix_cache: bool = True,
                 prefix_indexed: bool = True,
                 max_turns: Optional[int] = None, value_engine: Optional[GPT2Engine] = None, q1_head: Optional[dict] = None,
                 q2_head: Optional[dict] = None, beta: float = 0.0, session_flags: int = FWD_RAGGED_ALWAYS):
        import torch
        t = torch
        # text_env_eval(concurrent=n): a twin engine needs its own vectorised env (device state) — possible when the caller handed over a MazeEnv
        self._twin_args = dict(tokenizer=tokenizer, env=env if isinstance(env, M.MazeEnv) else None, batch=batch, max_new_tokens=max_new_tokens,
                               eos_token_id=eos_token_id, max_input_length=max_input_length, in_str_process=in_str_process, prefix_cache=prefix_cache,
                               prefix_indexed=prefix_indexed, max_turns=max_turns, value_engine=value_engine, q1_head=q1_head, q2_head=q2_head, beta=beta,
                               session_flags=session_flags)
        self._lanes = None
        venv = env.as_batched() if isinstance(env, M.MazeEnv) else env
        self.last_k = int(venv.last_k)
        if self.last_k != 1:
            prefix_cache = False                       # a prompt is no longer a function of (goal, cell)
        if max_turns is None:
            if venv.max_steps is None:
                raise ValueError("MazeRolloutEngine: env without max_steps needs max_turns")
            max_turns = venv.max_steps + 1            # the step after max_steps steps returns Failure (env.py:164-165)
        self.eng, self.tok, self.env, self.B = engine, tokenizer, venv, batch
        self.max_new, self.T = max_new_tokens, int(max_turns)
        self.eos = eos_token_id if eos_token_id is not None else getattr(tokenizer, "eos_token_id", None)
        self.pad = getattr(tokenizer, "pad_token_id", 0) or 0
        self.dev, self._L = engine.device, _lib.lib()
        self.in_str_process = in_str_process or (lambda x: x)
        self._max_input_length = int(max_input_length)
        self.prefix_cache = prefix_cache
        self.prefix_indexed = prefix_indexed        # read the prompt rows from the prefix cache (no per-turn copy) vs copy them per env
        # ---- observation table: one row per (goal slot, cell)
        maze, goals = venv.maze, venv.valid_goals
        R, C = maze.shape
        goal_slot = np.full(R * C, -1, dtype=np.int32)
        rows: List[List[int]] = []
        self._obs_text = {}
        for gi, g in enumerate(goals.tolist()):
            goal_slot[g[0] * C + g[1]] = gi
            for r in range(R):
                for c in range(C):
                    ids: List[int] = []
                    if maze[r, c] == 0:
                        text = venv.describe_function(maze, [r, c], [int(g[0]), int(g[1])], None, [])
                        self._obs_text[(gi, r, c)] = text
                        ids = list(tokenizer.encode(self.in_str_process(text)))
                        if len(ids) > max_input_length:               # Truncation.LEFT (interface.py:519-524)
                            ids = ids[len(ids) - max_input_length:]
                        ids = ids or [self.pad]
                    rows.append(ids)
        self.obs_cap = -(-max(len(x) for x in rows) // 16) * 16
        self.max_obs_len = max(len(x) for x in rows)
        obs_tok = np.zeros((len(rows), self.obs_cap), dtype=np.int32)
        obs_len = np.array([len(x) for x in rows], dtype=np.int32)
        for i, x in enumerate(rows):
            obs_tok[i, : len(x)] = x
        self.obs_tok_h, self.obs_len_h, self.goal_slot = obs_tok, obs_len, goal_slot
        tb, bl = token_byte_table(tokenizer, engine.cfg.vocab)
        self._tok = self._L.lmrl_maze_tok_create(obs_tok.ctypes.data, obs_len.ctypes.data, len(rows), self.obs_cap, goal_slot.ctypes.data, R, C,
                                                 tb.ctypes.data, bl.ctypes.data, engine.cfg.vocab, max_new_tokens, self.T)
        if not self._tok:
            raise _lib.LmrlError(self._L.lmrl_last_error().decode())
        # ---- sessions
        tmax = self.obs_cap + -(-max_new_tokens // 16) * 16        # whole 16-token prompt chunks (per-turn prefill mode) + the generated ids
        if self.last_k != 1:
            tmax = -(-int(max_input_length) // 16) * 16 + -(-max_new_tokens // 16) * 16
        self.veng, self.q1, self.q2, self.beta = value_engine, q1_head, q2_head, float(beta)
        assert (value_engine is None) == (q1_head is None), "value_engine and q1_head come together"
        self.engines = [engine] + ([value_engine] if value_engine is not None else [])
        if batch <= 16:
            session_flags |= FWD_SKINNY             # a handful of envs (configs[0]: 8): decode products on the skinny-M kernels (csrc/skinny_gemm.h)
        self.sessions = [e.session(batch, tmax, flags=session_flags) for e in self.engines]
        self.ses = self.sessions[0]
        self.vses = self.sessions[1] if value_engine is not None else None
        self.qh = [t.zeros(batch, value_engine.cfg.d_model, dtype=t.bfloat16, device=self.dev) for _ in range(2)] if value_engine is not None else None
        self.caches = None
        B, G, T = batch, max_new_tokens, self.T
        z = lambda *shape, dt: t.zeros(*shape, dtype=dt, device=self.dev)
        self.traj = dict(pos=z(B, T, dt=t.int32), gen=z(B, T, G, dt=t.int32), gen_len=z(B, T, dt=t.int32), action=z(B, T, dt=t.uint8),
                         reward=z(B, T, dt=t.float32), kind=z(B, T, dt=t.uint8), n_turns=z(B, dt=t.int32), live=z(B, dt=t.uint8),
                         ep_reward=z(B, dt=t.float32), obs_idx=z(B, dt=t.int32), out_tok=z(B, G, dt=t.int32), out_len=z(B, dt=t.int32),
                         gen_active=z(B, dt=t.uint8), act=z(B, dt=t.uint8), stepping=z(B, dt=t.uint8))
        self._ctraj = _CTraj(*[self.traj[n].data_ptr() for n, _ in _CTraj._fields_])
        self.chunk_tok, self.chunk_cnt = z(B * 16, dt=t.int32), z(B, dt=t.int32)
        self.next_tok, self.next_cnt = z(B, dt=t.int32), z(B, dt=t.int32)
        self.epoch = z(1, dt=t.int32)                  # 4th Philox counter word: episode base + turn (device side: graph replays advance it)
        self.env._alloc(batch)
        self.turn_graph = None
        self.episodes = 0                              # episode batches run by text_env_eval so far: successive calls draw fresh sampler noise
        if prefix_cache:
            self.refresh_prefix_cache()
        # the tokenizer's encoding of the four action strings: what a legal action is in later prompts (last_k > 1: the next prompt is
        # tokenizer.encode(window text)) and in the PPO chains (`ppo_records`: TokenTrajectory.from_text_trajectory re-encodes the action text), whatever
        # ids the policy generated to spell it
        acts = [list(tokenizer.encode(self.in_str_process(a))) for a in ("move left\n", "move right\n", "move up\n", "move down\n")]
        act_len = max(len(a) for a in acts)
        act_cap = act_len + 1
        at = np.zeros((4, act_cap), dtype=np.int32)
        for i, a in enumerate(acts):
            at[i, :len(a)] = a
            at[i, act_cap - 1] = len(a)
        _lib.check(self._L.lmrl_maze_tok_set_actions(self._tok, at.ctypes.data, act_cap), "lmrl_maze_tok_set_actions")
        self._act_len = act_len
        if self.last_k != 1:
            # ---- item-window state: token history of the whole episode + item offsets (2 items per turn after the first observation);
            # requires an encoding that is concatenative across items (byte level; BPE: items end in a newline)
            item_max = max(act_len, max_new_tokens)
            max_items = 2 * T + 2
            hcap = (T + 1) * (self.max_obs_len + item_max)
            self.hist = dict(hist=z(B, hcap, dt=t.int32), item_off=z(B, max_items + 1, dt=t.int32), n_items=z(B, dt=t.int32), feed_start=z(B, dt=t.int32),
                             feed_len=z(B, dt=t.int32), cache_len=z(B, dt=t.int32), base=z(B, dt=t.int32), prompt_len=z(B, dt=t.int32), win_floor=z(B, dt=t.int32),
                             flags=z(1, dt=t.int32))
            self._chist = _CHist(*[self.hist[n].data_ptr() for n, _ in _CHist._fields_[:10]], hcap, max_items)
            self._turn_i = 0
            self._turn_graphs = {}
            # static schedule: turn t (0-based) has 2 t + 1 items and at most (t + 1) observations + t actions of tokens
            pb = lambda i: (i + 1) * self.max_obs_len + i * act_len
            self._prompt_bound = [min(pb(i), self._max_input_length) for i in range(T + 1)]
            self._append_turn = [2 * i + 1 <= self.last_k and pb(i) <= self._max_input_length for i in range(T + 1)]
            self._append_chunks = -(-(self.max_obs_len + act_len) // 16)

    def close(self):
        if getattr(self, "_tok", None):
            self._L.lmrl_maze_tok_destroy(self._tok)
            self._tok = None
        self.env.close()
        self._drop_lanes()

    _script = None      # (steer ids int32 [T][max_new][B], turn counter int32 [1], this turn's rows [max_new][B], strength)

    def set_scripted_actions(self, codes, strength: float = 30.0) -> None:
        """Synthetic-workload hook (bench.py / tools, as `WordleRolloutEngine`'s scripted guesses): `codes` int [T][B] of LMRL_MAZE_LEFT..DOWN — in
        turn t the sampler of env b is steered (+`strength` on one logit per generated position; every logit is still computed and sampled) towards
        spelling that move's text with the tokenizer's own ids, so that a random-init policy walks and its history window fills as a trained
        policy's does.  None switches it off.  Captured turn graphs are dropped (the steer operand becomes part of a turn)."""
        import torch
        t = torch
        self.turn_graph = None
        if codes is None:
            self._script = None
            return
        codes = np.asarray(codes, dtype=np.int64)
        assert codes.shape == (self.T, self.B) and codes.min() >= 0 and codes.max() < 4
        acts = [list(self.tok.encode(self.in_str_process(a))) for a in ("move left\n", "move right\n", "move up\n", "move down\n")]
        table = np.full((4, self.max_new), -1, dtype=np.int32)
        for i, a in enumerate(acts):
            table[i, :min(len(a), self.max_new)] = a[:self.max_new]
        steer = np.full((self.T + 4 * self.T + 8, self.max_new, self.B), -1, dtype=np.int32)   # [T (+ slack: capture warm-ups advance the counter too)][max_new][B]
        steer[:self.T] = table[codes].transpose(0, 2, 1)
        self._script = (t.from_numpy(steer).to(self.dev), t.zeros(1, dtype=t.int32, device=self.dev),
                        t.full((self.max_new, self.B), -1, dtype=t.int32, device=self.dev), float(strength))

    def _drop_lanes(self):
        for twin, _ in (self._lanes or [])[1:]:
            twin.close()
        self._lanes = None

    # ---- prompt-prefix cache ------------------------------------------------------------------------------------------------------
    def refresh_prefix_cache(self):
        """Prefill every distinct observation once (per engine); call again after the weights change (`set_params`)."""
        import torch
        t = torch
        n, cap = self.obs_tok_h.shape
        self.caches = []
        self.turn_graph = None      # a captured turn replays attach_prefix_from against the OLD cache sessions: never keep it across a refresh
        lens = t.from_numpy(self.obs_len_h).to(self.dev)
        for e in self.engines:
            ses = e.session(n, cap)
            ses.reset()
            for c0 in range(0, self.max_obs_len, 16):
                toks = t.from_numpy(np.ascontiguousarray(self.obs_tok_h[:, c0:c0 + 16]).reshape(-1)).to(self.dev)
                cnt = t.clamp(lens - c0, 0, 16).to(t.int32)
                ses.forward(toks, cnt, 16)
            self.caches.append(ses)

    def set_params(self, engine: GPT2Engine) -> None:
        """Swap in freshly trained policy weights (PPOPolicy.set_params, ppo/base_interface.py:821-823)."""
        self.eng = self.engines[0] = engine
        self.sessions[0] = self.ses = engine.session(self.B, self.ses.tmax, flags=self.ses.flags)
        self.turn_graph = None
        self._drop_lanes()          # twins hold the old weights' sessions / prefix caches: rebuilt on the next concurrent call
        if self.prefix_cache:
            self.refresh_prefix_cache()

    def load_params(self, params) -> None:
        """The trainer's fp32 parameters into the policy engine IN PLACE (`GPT2Engine.load_params`: captured turns stay valid, they read the same
        weight buffers), then the observation prefix cache again — its K/V rows were computed by the old weights."""
        self.eng.load_params(params)
        self._drop_lanes()
        if self.prefix_cache:
            self.refresh_prefix_cache()

    # ---- one lock-step turn ----------------------------------------------------------------------------------------------------------
    def _turn(self, temperature: float, top_k: int, sample_seed: int, logits_out=None, turn_index: int = 0):
        L, sp, tr, B = self._L, _lib.stream_ptr(), ctypes.byref(self._ctraj), self.B
        ck = _lib.check
        ck(L.lmrl_maze_tok_turn(self._tok, tr, _lib.ptr(self.env.state), B, sp), "maze_tok_turn")
        if self.last_k != 1:
            # item window on the persistent cache: this turn's feed = the action's unforwarded tail + the new observation (append), or the whole
            # window from position 0 (re-prefill)
            ti = min(turn_index, self.T)
            append = self._append_turn[ti]
            hs = ctypes.byref(self._chist)
            len1 = _lib.ptr(self.vses.len) if self.vses is not None else None
            n_chunks = self._append_chunks if append else -(-self._prompt_bound[ti] // 16)
            ck(L.lmrl_maze_hist_observe(self._tok, tr, hs, self.last_k, self._max_input_length, 0 if append else 1, n_chunks * 16, _lib.ptr(self.ses.len), len1,
                                        B, sp), "maze_hist_observe")
            for ses in self.sessions:
                ses._len_bound = 0
                ses.shared_prefix = 0
                ses._prefix = None
            for j in range(n_chunks):
                ck(L.lmrl_maze_hist_chunk(hs, j, 16, _lib.ptr(self.chunk_tok), _lib.ptr(self.chunk_cnt), B, sp), "maze_hist_chunk")
                for ses in self.sessions:
                    ses.forward(self.chunk_tok, self.chunk_cnt, 16, len_bound_after=self._prompt_bound[ti])
        elif self.prefix_cache:
            for ses, cache in zip(self.sessions, self.caches):
                if self.prefix_indexed:
                    ses.attach_prefix_from(cache, self.traj["obs_idx"], self.max_obs_len)
                else:
                    ses.gather_prefix_from(cache, self.traj["obs_idx"], self.max_obs_len)
        else:
            for ses in self.sessions:
                ses.reset()
            for j in range(-(-self.max_obs_len // 16)):
                ck(L.lmrl_maze_tok_prompt(self._tok, tr, j, 16, _lib.ptr(self.chunk_tok), _lib.ptr(self.chunk_cnt), B, sp), "maze_tok_prompt")
                for ses in self.sessions:
                    ses.forward(self.chunk_tok, self.chunk_cnt, 16)
        script = self._script
        if script is not None:      # this turn's steer rows: row `turn counter` of the scripted table (a device-side index: the same launch in every replay)
            ck(L.lmrl_gather_rows_bytes(script[0].data_ptr(), script[1].data_ptr(), script[2].data_ptr(), 1, self.max_new * B * 4, sp), "lmrl_gather_rows_bytes")
        for k in range(self.max_new):
            p = SampleParams(temperature, top_k, sample_seed, k, script[3] if script is not None else 0.0, self.beta, self.pad, _lib.ptr(self.epoch))
            qops = [None, None]
            if self.vses is not None:
                dv = self.veng.cfg.d_model
                for i, head in enumerate((self.q1, self.q2)):
                    if head is not None:
                        ck(L.lmrl_gemm_bf16(_lib.ptr(self.vses.last_hidden), _lib.ptr(head["w1"]), _lib.ptr(head["b1"]), _lib.ptr(self.qh[i]),
                                            B, dv, dv, dv, dv, dv, 4, sp), "q head dense1 + relu")
                        qops[i] = (self.qh[i], head["w2"], head["b2"])
            self.ses.sample(p, steer_tok=script[2][k] if script is not None else None, active=self.traj["gen_active"], logits_out=logits_out, q1=qops[0],
                            q2=qops[1], want_logprob=False)
            ck(L.lmrl_gen_accept(_lib.ptr(self.ses.token), _lib.ptr(self.traj["gen_active"]), _lib.ptr(self.traj["out_tok"]),
                                 _lib.ptr(self.traj["out_len"]), _lib.ptr(self.next_tok), _lib.ptr(self.next_cnt),
                                 -1 if self.eos is None else int(self.eos), self.max_new, B, sp), "lmrl_gen_accept")
            if k < self.max_new - 1:
                for ses in self.sessions:
                    ses.forward(self.next_tok, self.next_cnt, 1)
        ck(L.lmrl_maze_tok_action(self._tok, tr, B, sp), "maze_tok_action")
        if self.last_k != 1:
            ck(L.lmrl_maze_hist_action(self._tok, tr, ctypes.byref(self._chist), _lib.ptr(self.ses.len),
                                       _lib.ptr(self.vses.len) if self.vses is not None else None, B, sp), "maze_hist_action")
        e = self.env
        ck(L.lmrl_maze_step(e._ctx, _lib.ptr(e.state), _lib.ptr(self.traj["act"]), _lib.ptr(self.traj["stepping"]), _lib.ptr(e.reward),
                            _lib.ptr(e.done), _lib.ptr(e.kind), _lib.ptr(e.walls), B, sp), "lmrl_maze_step")
        ck(L.lmrl_maze_tok_result(self._tok, tr, _lib.ptr(e.reward), _lib.ptr(e.done), _lib.ptr(e.kind), B, sp), "maze_tok_result")
        self.epoch.add_(1)
        if script is not None:
            script[1].add_(1)

    def _reset_script_counter(self):
        if self._script is not None:
            self._script[1].zero_()

    def _capture_history_turns(self, temperature: float, top_k: int, sample_seed: int):
        """last_k > 1: one hipGraph per KIND of turn — the append turn (a fixed number of 16-token chunks) and one re-prefill turn per distinct
        chunk count of the schedule (the window's token bound grows until it reaches max_input_length)."""
        import torch
        t = torch
        self._graph_args = (temperature, top_k, sample_seed)
        self._logits = t.empty(self.B, self.eng.cfg.vocab_padded, dtype=t.float32, device=self.dev) if top_k > 0 else None
        self._turn_graphs = {}
        kinds = {}
        for ti in range(self.T):
            key = ("a",) if self._append_turn[ti] else ("r", -(-self._prompt_bound[ti] // 16))
            kinds.setdefault(key, ti)
        for key, ti in kinds.items():
            self.env.reset_device([0] * self.B)
            self._reset_script_counter()
            self._hist_begin()
            _lib.check(self._L.lmrl_maze_tok_begin(self._tok, ctypes.byref(self._ctraj), self.B, _lib.stream_ptr()), "maze_tok_begin")
            self._turn(temperature, top_k, sample_seed, self._logits, turn_index=ti)           # eager warm-up
            t.cuda.synchronize()
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g, capture_error_mode="thread_local"):
                self._turn(temperature, top_k, sample_seed, self._logits, turn_index=ti)
            self._turn_graphs[key] = g
        self.turn_graph = True

    def _hist_begin(self):
        _lib.check(self._L.lmrl_maze_hist_begin(ctypes.byref(self._chist), _lib.ptr(self.ses.len), _lib.ptr(self.vses.len) if self.vses is not None else None,
                                                self.B, _lib.stream_ptr()), "maze_hist_begin")
        for ses in self.sessions:
            ses._len_bound = 0
            ses.shared_prefix = 0
            ses._prefix = None
        self._turn_i = 0

    def history_flags(self) -> int:
        """last_k > 1: 0 when every turn of the last episodes ran as scheduled (bit 0: an append turn would have needed a re-prefill — a prompt outgrew
        the static bound; bit 1: history buffer overflow).  One readback."""
        return int(self.hist["flags"].cpu().numpy()[0]) if self.last_k != 1 else 0

    def capture_turn(self, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0):
        """One turn (~ max_new x n_layer x 7 launches) as a hipGraph; `run_episode(..., use_graph=True)` replays it per turn."""
        import torch
        t = torch
        if self.last_k != 1:
            return self._capture_history_turns(temperature, top_k, sample_seed)
        self._graph_args = (temperature, top_k, sample_seed)
        self._logits = t.empty(self.B, self.eng.cfg.vocab_padded, dtype=t.float32, device=self.dev) if top_k > 0 else None
        self.env.reset_device([0] * self.B)
        _lib.check(self._L.lmrl_maze_tok_begin(self._tok, ctypes.byref(self._ctraj), self.B, _lib.stream_ptr()), "maze_tok_begin")
        self._turn(temperature, top_k, sample_seed, self._logits)           # eager warm-up
        t.cuda.synchronize()
        g = t.cuda.CUDAGraph()
        with t.cuda.graph(g, capture_error_mode="thread_local"):
            self._turn(temperature, top_k, sample_seed, self._logits)
        self.turn_graph = g
        return g

    def run_episode(self, seeds, options=None, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0, episode: int = 0,
                    use_graph: bool = False, sync_every: int = 8, max_turns: Optional[int] = None):
        """One full episode for all B envs; returns the device record dict (read after a sync).  The host peeks at the live flags every
        `sync_every` turns to stop early (0: never — fixed T turns, fully asynchronous)."""
        turn = self._episode_begin(seeds, options, temperature, top_k, sample_seed, episode, use_graph)
        n = self.T if max_turns is None else min(self.T, max_turns)
        for i in range(n):
            if sync_every and i and i % sync_every == 0 and not bool(self.traj["live"].any().item()):
                break
            turn()
        return self.traj

    def _episode_begin(self, seeds, options, temperature, top_k, sample_seed, episode, use_graph):
        """Reset + first bookkeeping of an episode on the CURRENT stream; returns the callable that enqueues one lock-step turn."""
        import torch
        t = torch
        if use_graph:
            if self.turn_graph is None or self._graph_args != (temperature, top_k, sample_seed):
                self.capture_turn(temperature, top_k, sample_seed)
            logits = self._logits
        else:
            logits = t.empty(self.B, self.eng.cfg.vocab_padded, dtype=t.float32, device=self.dev) if top_k > 0 else None
        assert len(seeds) == self.B
        self.env.reset_device(list(seeds), options)
        _lib.check(self._L.lmrl_maze_tok_begin(self._tok, ctypes.byref(self._ctraj), self.B, _lib.stream_ptr()), "maze_tok_begin")
        self.epoch.fill_(int(episode) << 12)
        if self._script is not None:
            self._script[1].zero_()
        if self.last_k != 1:
            self._hist_begin()

            def turn():
                ti = min(self._turn_i, self.T - 1)
                self._turn_i += 1
                if use_graph:
                    key = ("a",) if self._append_turn[ti] else ("r", -(-self._prompt_bound[ti] // 16))
                    for ses in self.sessions:          # (host-side bounds only: the replayed launches carry their own)
                        ses._len_bound = 0
                    self._turn_graphs[key].replay()
                else:
                    self._turn(temperature, top_k, sample_seed, logits, turn_index=ti)
            return turn
        return self.turn_graph.replay if use_graph else (lambda: self._turn(temperature, top_k, sample_seed, logits))

    # ---- the online-RL hand-over: the finished episodes as PPO data, on the device --------------------------------------------------
    def ppo_records(self, n: Optional[int] = None):
        """The finished episodes (of the first `n` envs; default all) as `algorithms.ppo_device.PPORecords`: one token trajectory per transition (observation ids ++ action ids,
        reward on the action's last token), chained per episode — the chains the Maze / chess online scripts build from `raw_results`
        (llm_rl_scripts/maze/ppo/train_ppo_online.py:444-465) after `TokenTrajectory.from_text_trajectory`.  Two small readbacks (the number of
        transitions, the longest chain) size the arrays.  The action ids: a legal action as the tokenizer's encoding of its dict key (the
        reference's re-tokenisation, any tokenizer); any other string as its generated ids (special tokens dropped, a newline id appended when
        the text has no trailing newline) — the re-tokenisation whenever encode(decode(ids)) == ids — or, for a tokenizer whose ids are UTF-8
        bytes, as the bytes of its decoded tokens (exact also for multi-byte tokens); the observation ids are the
        prompt table's rows (`in_str_process` must be the identity and no prompt may have been left-truncated: checked)."""
        import torch
        from .algorithms.ppo_device import PPORecords
        t, L, sp = torch, self._L, _lib.stream_ptr()
        if self.last_k != 1:
            return self._ppo_records_history(n)
        if self.in_str_process("\x00probe") != "\x00probe" or self.obs_len_h.max() >= self._max_input_length:
            raise ValueError("ppo_records: the PPO chains tokenise the raw observation text — in_str_process must be the identity and prompts untruncated")
        B = self.B if n is None else int(n)
        off = t.empty(B + 1, dtype=t.int32, device=self.dev)
        _lib.check(L.lmrl_exclusive_scan_i32(_lib.ptr(self.traj["n_turns"]), _lib.ptr(off), B, sp), "lmrl_exclusive_scan_i32")
        N = int(off[B:].cpu().numpy()[0])
        if N == 0:
            raise ValueError("ppo_records: no transition recorded (run an episode first)")
        nl = self.tok.encode("\n")
        assert len(nl) == 1, "the newline must be one token"
        byte_ids = self._ids_are_bytes()
        cap = int(self.max_obs_len) + max(self.max_new * (_TOK_BYTES if byte_ids else 1), self._act_len) + 1
        z = lambda *s_, dt: t.zeros(*s_, dtype=dt, device=self.dev)
        tokens, ia, rw = z(N, cap, dt=t.int32), z(N, cap, dt=t.uint8), z(N, cap, dt=t.float32)
        n_tok, chain, pos, last = z(N, dt=t.int32), z(N, dt=t.int32), z(N, dt=t.int32), z(N, dt=t.uint8)
        done, total = z(B, dt=t.uint8), z(B, dt=t.int32)
        _lib.check(L.lmrl_maze_tok_ppo_records(self._tok, ctypes.byref(self._ctraj), _lib.ptr(self.env.state), B, self.B, _lib.ptr(off), int(nl[0]), 1 if byte_ids else 0, cap, _lib.ptr(tokens),
                                               _lib.ptr(ia), _lib.ptr(rw), _lib.ptr(n_tok), _lib.ptr(chain), _lib.ptr(pos), _lib.ptr(last), _lib.ptr(done),
                                               _lib.ptr(total), sp), "lmrl_maze_tok_ppo_records")
        return PPORecords(tokens, ia, rw, n_tok, done, chain, pos, last, n_chains=B, chain_len_bound=max(int(total.cpu().numpy().max()), 1))

    def _ensure_spaced_tables(self):
        """Host tables of `lmrl_maze_tok_set_spaced`: every observation / action text encoded behind the joining space of the partially observed
        script's state text (" ".join(item texts), partially_observed_ppo_online.py:378-379).  Concatenating them is `tokenizer.encode(joined text)` only
        for tokenizers that split between an item's end and the space: probed here on real item pairs (byte-level: always; GPT-2 BPE: items end in a
        newline, the space goes with the next word)."""
        if getattr(self, "_spaced_ready", False):
            return
        enc = lambda s_: list(self.tok.encode(s_))
        C = self.env.maze.shape[1]
        R = self.env.maze.shape[0]
        n_rows = len(self.obs_len_h)
        rows = [[] for _ in range(n_rows)]
        for (gi, r, c), text in self._obs_text.items():
            rows[(gi * R + r) * C + c] = enc(" " + text)
        acts = ("move left\n", "move right\n", "move up\n", "move down\n")
        sp_acts = [enc(" " + a) for a in acts]
        some_obs = [self._obs_text[k] for k in list(self._obs_text)[:3]]
        for a in some_obs + list(acts):
            for b in some_obs + list(acts):
                if enc(a + " " + b) != enc(a) + enc(" " + b):
                    raise ValueError("ppo_records (last_k > 1): this tokenizer does not encode ' '.join(items) item by item — use the host text path")
        cap = max(1, max(len(x) for x in rows))
        ot, ol = np.zeros((n_rows, cap), dtype=np.int32), np.array([len(x) for x in rows], dtype=np.int32)
        for i, x in enumerate(rows):
            ot[i, :len(x)] = x
        acap = max(len(a) for a in sp_acts) + 1
        at = np.zeros((4, acap), dtype=np.int32)
        for i, a in enumerate(sp_acts):
            at[i, :len(a)] = a
            at[i, acap - 1] = len(a)
        _lib.check(self._L.lmrl_maze_tok_set_spaced(self._tok, ot.ctypes.data, ol.ctypes.data, cap, at.ctypes.data, acap), "lmrl_maze_tok_set_spaced")
        self._spaced_ready = True

    def _ids_are_bytes(self) -> bool:
        """Are this tokenizer's ids the text's UTF-8 bytes?  Then `ppo_records` exports an action string outside the dict as encode(its decoded text)
        exactly, also when the policy spelled it with multi-byte tokens (valid UTF-8); otherwise as its generated ids (equal whenever
        encode(decode(ids)) == ids)."""
        if getattr(self, "_byte_ids", None) is None:
            probe = [self._obs_text[k] for k in list(self._obs_text)[:3]] + ["move left\n", "move down\n"]
            probe += [_decode_one(self.tok, i) for i in range(min(self.eng.cfg.vocab, 1024))]
            self._byte_ids = all(list(self.tok.encode(x)) == list(x.encode("utf-8")) for x in probe if x)
        return self._byte_ids

    def _ppo_records_history(self, n: Optional[int] = None):
        """`ppo_records` for item windows (last_k > 1): the chains of the partially observed online script (llm_rl_scripts/maze/ppo/
        partially_observed_ppo_online.py:372-398: `last_k = 40`) — per transition the window's item texts joined by single spaces as ONE non-action
        text, then the action, reward [0, r] — built on the device from the episode record (`lmrl_maze_tok_ppo_records_hist`: the window of every turn
        is rebuilt from the recorded cells, action codes and step kinds).  Two passes: the row lengths first (the longest row sizes the arrays)."""
        import torch
        from .algorithms.ppo_device import PPORecords
        t, L, sp = torch, self._L, _lib.stream_ptr()
        if self.in_str_process("\x00probe") != "\x00probe" or self.obs_len_h.max() >= self._max_input_length:
            raise ValueError("ppo_records: the PPO chains tokenise the raw observation text — in_str_process must be the identity and observations untruncated")
        if self.last_k > 64:
            raise ValueError("ppo_records: item windows of at most 64 items on the device (last_k = %d)" % self.last_k)
        if self.history_flags() != 0:
            raise ValueError("ppo_records: the last episode violated its turn schedule (history_flags = %d)" % self.history_flags())
        self._ensure_spaced_tables()
        B = self.B if n is None else int(n)
        off = t.empty(B + 1, dtype=t.int32, device=self.dev)
        _lib.check(L.lmrl_exclusive_scan_i32(_lib.ptr(self.traj["n_turns"]), _lib.ptr(off), B, sp), "lmrl_exclusive_scan_i32")
        N = int(off[B:].cpu().numpy()[0])
        if N == 0:
            raise ValueError("ppo_records: no transition recorded (run an episode first)")
        nl = self.tok.encode("\n")
        assert len(nl) == 1, "the newline must be one token"
        z = lambda *s_, dt: t.zeros(*s_, dtype=dt, device=self.dev)
        n_tok, chain, pos, last = z(N, dt=t.int32), z(N, dt=t.int32), z(N, dt=t.int32), z(N, dt=t.uint8)
        done, total = z(B, dt=t.uint8), z(B, dt=t.int32)
        call = lambda cap, tokens, ia, rw: _lib.check(L.lmrl_maze_tok_ppo_records_hist(
            self._tok, ctypes.byref(self._ctraj), _lib.ptr(self.env.state), B, self.B, _lib.ptr(off), self.last_k, int(nl[0]), 1 if self._ids_are_bytes() else 0, cap,
            _lib.ptr(tokens) if tokens is not None else None, _lib.ptr(ia) if ia is not None else None, _lib.ptr(rw) if rw is not None else None,
            _lib.ptr(n_tok), _lib.ptr(chain), _lib.ptr(pos), _lib.ptr(last), _lib.ptr(done), _lib.ptr(total), sp), "lmrl_maze_tok_ppo_records_hist")
        call(0, None, None, None)
        cap = max(int(n_tok.max().item()), 2)
        tokens, ia, rw = z(N, cap, dt=t.int32), z(N, cap, dt=t.uint8), z(N, cap, dt=t.float32)
        call(cap, tokens, ia, rw)
        return PPORecords(tokens, ia, rw, n_tok, done, chain, pos, last, n_chains=B, chain_len_bound=max(int(total.cpu().numpy().max()), 1))

    def ppo_data(self, inference, *, gamma: float, lam: float, kl_weight: float, max_length: Optional[int] = None, n: Optional[int] = None, **kw):
        """`ppo_dataset_loader` of the Maze online script on the episodes this engine just ran -> (DevicePPODataset, all_kls): see
        `algorithms.ppo_device.ppo_data_from_records`."""
        from .algorithms.ppo_device import ppo_data_from_records
        return ppo_data_from_records(inference, self.ppo_records(n), gamma=gamma, lam=lam, kl_weight=kl_weight, max_length=max_length, **kw)

    def ppo_rollouts(self, inference, n_rollouts: int, seed_generator=None, env_options=None, *, gamma: float, lam: float, kl_weight: float,
                     max_length: Optional[int] = None, use_advantage_whitening: bool = True, temperature: float = 1.0, top_k: int = 0, sample_seed: int = 0,
                     use_graph: bool = True, **kw):
        """One data-collection round of the Maze online PPO loop on the device: `text_env_eval(n_rollouts, bsize=B)` + `ppo_dataset_loader`
        (maze/ppo/train_ppo_online.py:431-483) -> (DevicePPODataset over all transitions of all rollouts, all_kls, summary).  Per episode batch: the
        lock-step episode, then its PPO data while the record is in the engine's buffers; advantages whitened once over the action tokens of the whole
        round; `summary` has `text_env_eval`'s shape (from the per-env counters; no text is built)."""
        import torch
        from .algorithms.ppo_device import DevicePPODataset
        from . import dist as D
        parts, kls, stats = [], [], []
        launched = 0
        while launched < n_rollouts:
            actual = min(n_rollouts - launched, self.B)
            seeds = [0] * self.B
            seeds[:actual] = [next(seed_generator) for _ in range(actual)] if seed_generator is not None else np.random.randint(0, 2 ** 31 - 1, size=actual).tolist()
            self.run_episode(seeds, env_options, temperature=temperature, top_k=top_k, sample_seed=sample_seed, episode=self.episodes, use_graph=use_graph)
            self.episodes += 1
            launched += actual
            ds, kl = self.ppo_data(inference, gamma=gamma, lam=lam, kl_weight=kl_weight, max_length=max_length, n=actual, use_advantage_whitening=False, **kw)
            parts.append(ds); kls.append(kl)
            kind = self.traj["kind"][:actual].cpu().numpy(); nt = self.traj["n_turns"][:actual].cpu().numpy()
            last_kind = kind[np.arange(actual), np.maximum(nt - 1, 0)]
            stats.append(np.stack([self.traj["ep_reward"][:actual].cpu().numpy().astype(np.float64), ((last_kind == 1) | (last_kind == 2)).astype(np.float64),
                                   nt.astype(np.float64)]))
        ds = DevicePPODataset.concat(parts)
        if use_advantage_whitening:
            adv = ds.old_advantages
            ds.old_advantages = D.whiten_distributed(adv.view(-1), ds.should_take_action.view(-1), shift_mean=True).view(adv.shape)
        st = np.concatenate(stats, axis=1)
        summ = lambda x: dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))
        return ds, (kls[0] if len(kls) == 1 else torch.cat(kls)), dict(reward=summ(st[0].astype(np.float32)), done=summ(st[1].astype(np.float32)),
                                                                       length=summ(st[2].astype(np.int64)))

    # ---- host views ----------------------------------------------------------------------------------------------------------------
    def records(self):
        """Host copies per env: dict(pos [n][2], gen list of id lists, action uint8 [n], reward f32 [n], kind uint8 [n], goal (r, c))."""
        h = {k: v.cpu().numpy() for k, v in self.traj.items()}
        st = self.env.positions()
        out = []
        for b in range(self.B):
            n = int(h["n_turns"][b])
            pos = np.stack([h["pos"][b, :n] >> 16, h["pos"][b, :n] & 0xFFFF], axis=1) if n else np.zeros((0, 2), dtype=np.int32)
            gen = [h["gen"][b, i, : h["gen_len"][b, i]].tolist() for i in range(n)]
            out.append(dict(pos=pos, gen=gen, action=h["action"][b, :n].copy(), reward=h["reward"][b, :n].copy(), kind=h["kind"][b, :n].copy(),
                            goal=(int(st[b, 2]), int(st[b, 3])), final_pos=(int(st[b, 0]), int(st[b, 1])), live=bool(h["live"][b])))
        return out

    def _decode(self, ids: List[int]) -> str:
        try:
            return self.tok.decode(ids, skip_special_tokens=True)
        except TypeError:
            special = getattr(self.tok, "all_special_ids", None)
            if special is None:
                special = [t for t in (getattr(self.tok, "pad_token_id", None),) if t is not None]
            special = set(special)
            return self.tok.decode([i for i in ids if i not in special])

    def interactions(self):
        """The finished episodes as `List[List[InteractionTransition]]` — what `interact_environment` returns for the same rollouts
        (LLM_RL/environment.py:154-207; histories per maze/env/env.py:161-184 with last_k = 1)."""
        h = {k: v.cpu().numpy() for k, v in self.traj.items() if k in ("pos", "gen", "gen_len", "reward", "kind", "n_turns")}
        return self._build_interactions(h, self.env.positions())

    def snapshot_records(self):
        """Enqueue device -> pinned-host copies of the episode record and the env state behind the episode's kernels (current stream) and return a
        handle for `_build_interactions_from(handle)`: the buffers may be reused by the next episode as soon as this returns (stream order)."""
        import torch
        names = ("pos", "gen", "gen_len", "reward", "kind", "n_turns")
        if getattr(self, "_pinned", None) is None:
            mk = lambda x: torch.empty(x.shape, dtype=x.dtype, pin_memory=True)
            self._pinned = [dict({n: mk(self.traj[n]) for n in names}, state=mk(self.env.state)) for _ in range(2)]
            self._pin_i = 0
        buf = self._pinned[self._pin_i]
        self._pin_i ^= 1
        for n in names:
            buf[n].copy_(self.traj[n], non_blocking=True)
        buf["state"].copy_(self.env.state, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return buf, ev

    def _build_interactions_from(self, handle):
        import gc
        buf, ev = handle
        ev.synchronize()
        was = gc.isenabled()
        gc.disable()                    # tens of thousands of live tuples are created: no cyclic garbage among them (as WordleRolloutEngine)
        try:
            return self._build_interactions({k: v.numpy() for k, v in buf.items() if k != "state"}, buf["state"].numpy().T)
        finally:
            if was:
                gc.enable()

    def _build_interactions(self, h, st):
        """Host lists from host copies of the record (`h`) and the env state rows (`st`: row, col, goal row, goal col per env).  The observation
        texts are the ones rendered once per (goal, cell) for the token table (`_obs_text`: the same `describe_function` call the reference makes
        per step); Text objects are immutable and shared; generated ids are decoded once per distinct sequence."""
        from .environment import InteractionTransition, Text
        C = self.env.maze.shape[1]
        obs_cache, act_cache = self.__dict__.setdefault("_obs_Text", {}), {}
        fail, succ = (Text("Failure\n", False),), (Text("Success\n", False),)

        def desc(gi, r, c):
            key = (gi, r, c)
            t = obs_cache.get(key)
            if t is None:
                t = obs_cache[key] = (Text(self._obs_text[key], False),)
            return t

        def action(ids):
            t = act_cache.get(ids)
            if t is None:
                t = act_cache[ids] = Text(maze_out_str_process(self._decode(list(ids))), True)
            return t
        pos, gen, gen_len, reward, kinds, n_turns = h["pos"], h["gen"], h["gen_len"], h["reward"], h["kind"], h["n_turns"]
        out = []
        for b in range(self.B):
            n = int(n_turns[b])
            gi = int(self.goal_slot[int(st[b, 2]) * C + int(st[b, 3])])
            pr, pc = (pos[b, :n] >> 16).tolist(), (pos[b, :n] & 0xFFFF).tolist()
            gl, kd, rw = gen_len[b, :n].tolist(), kinds[b, :n].tolist(), reward[b, :n].tolist()
            gb = gen[b]
            trans = []
            items = ()                  # last_k > 1: every item of the episode so far; a history is its last `last_k` items (maze/env/env.py:182-184)
            for i in range(n):
                obs_i = desc(gi, pr[i], pc[i])
                if self.last_k == 1:
                    pre = obs_i
                else:
                    items = items + obs_i if i == 0 else items
                    pre = items[-self.last_k:]
                act = action(tuple(gb[i, :gl[i]].tolist()))
                post_action = pre + (act,)
                kind = kd[i]
                nxt = None
                if kind == M.KIND_FAILURE:
                    post, done = fail, True
                elif kind == M.KIND_SUCCESS:
                    post, done = succ, True
                else:
                    nxt = desc(gi, pr[i + 1], pc[i + 1]) if i + 1 < n else desc(gi, int(st[b, 0]), int(st[b, 1]))
                    post, done = nxt, False
                if self.last_k != 1 and nxt is not None:
                    # an action string outside the action dict: the env returns (observation,) alone and the window restarts there (env.py:179-180)
                    items = nxt if kind == M.KIND_OBS_ONLY else items + (act,) + nxt
                    post = items[-self.last_k:]
                trans.append(InteractionTransition(pre, post_action, post, float(rw[i]), done))
            out.append(trans)
        return out

    def _eval_lanes(self, n: int, main):
        """[(engine, stream)]: this engine on the caller's stream + n - 1 twins (same weights, tokenizer and maze; own env state, sessions, prefix
        cache, records and turn graph) on their own streams.  Built once, dropped by `set_params`."""
        import torch
        if self._twin_args["env"] is None:
            return [(self, main)]                   # built from an already vectorised env: no second env state to be had
        if self._lanes is None:
            self._lanes = [(self, None)]
        while len(self._lanes) < n:
            st = torch.cuda.Stream(device=self.dev)
            with torch.cuda.stream(st):
                twin = MazeRolloutEngine(self.eng, **self._twin_args)
            self._lanes.append((twin, st))
        return [(self, main)] + self._lanes[1:n]

    def text_env_eval(self, n_rollouts: int, seed_generator=None, env_options=None, temperature: float = 1.0, top_k: int = 0,
                      sample_seed: int = 0, interaction_callback=None, use_graph: bool = True, concurrent: int = 1, sync_every: int = 8):
        """`text_env_eval(env, policy, n_rollouts, bsize=B, env_options=...)` (LLM_RL/environment.py:211-267) with the whole lock-step
        loop on the device: ceil(n / B) episode batches, the same (interactions, summary) return value.  The sampler's episode word keeps
        counting across calls (`self.episodes`, as GPT2PPOPolicy splits its PRNG key on every act(), ppo/gpt2/interface.py:524-526): two PPO
        rounds with the same `sample_seed` do not replay the same noise."""
        import torch
        inter, rewards, dones, lengths = [], [], [], []
        main = torch.cuda.current_stream(self.dev)
        n_batches = -(-n_rollouts // self.B)
        lanes = self._eval_lanes(min(int(concurrent), n_batches), main) if concurrent > 1 and n_batches >= 2 else [(self, main)]
        for _, st in lanes[1:]:
            st.wait_stream(main)
        options = [env_options] * self.B if env_options is not None else None

        def absorb(handles):
            for eng, handle, actual in handles or ():
                for ep in eng._build_interactions_from(handle)[:actual]:
                    inter.append(ep)
                    rewards.append(sum(t.reward for t in ep)); dones.append(ep[-1].done); lengths.append(len(ep))
                    if interaction_callback is not None:
                        interaction_callback(ep)
        launched, pending = 0, None
        while launched < n_rollouts:
            # one GROUP of up to len(lanes) episode batches, their turns enqueued alternately: every lane is a dependent chain of small launches,
            # two chains fill each other's idle CUs (as WordleRolloutEngine.text_env_eval(concurrent=n)); batches are independent in the reference too
            group = []
            for eng, st in lanes:
                if launched >= n_rollouts:
                    break
                actual = min(n_rollouts - launched, self.B)
                seeds = [0] * self.B
                seeds[:actual] = [next(seed_generator) for _ in range(actual)] if seed_generator is not None else \
                    np.random.randint(0, 2 ** 31 - 1, size=actual).tolist()
                with torch.cuda.stream(st):
                    turn = eng._episode_begin(seeds, options, temperature, top_k, sample_seed, self.episodes, use_graph)
                self.episodes += 1
                launched += actual
                group.append([eng, st, turn, actual, True])
            for i in range(self.T):
                if sync_every and i and i % sync_every == 0:
                    for g in group:
                        if g[4]:
                            with torch.cuda.stream(g[1]):
                                g[4] = bool(g[0].traj["live"].any().item())       # waits for THIS lane's stream only; the other lanes' queued turns keep running
                    if not any(g[4] for g in group):
                        break
                for g in group:
                    if g[4]:
                        with torch.cuda.stream(g[1]):
                            g[2]()
            handles = []
            for eng, st, _, actual, _ in group:
                with torch.cuda.stream(st):
                    handles.append((eng, eng.snapshot_records(), actual))
            # the previous group's host lists are built now, while this group's turns run on the device (sync_every = 0: everything above was
            # enqueued without waiting; with early-exit peeks the host has already waited for most of it)
            absorb(pending)
            pending = handles
        absorb(pending)
        for _, st in lanes[1:]:
            main.wait_stream(st)
        summ = lambda x: dict(mean=np.mean(x), std=np.std(x), min=np.min(x), max=np.max(x))
        return inter, dict(reward=summ(np.asarray(rewards, dtype=np.float32)), done=summ(np.asarray(dones, dtype=np.float32)), length=summ(lengths))
