"""Task harness: the configurations of the reference's task scripts, composed from this package.

    python scripts/harness.py ilql     --train-data d.jsonl [--eval-data e.jsonl]    # llm_rl_scripts/wordle/ilql/train_ilql_gpt2.py
    python scripts/harness.py ppo      [--bc-data d.jsonl]                           # llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py
    python scripts/harness.py ppo      --env maze --device-rollouts 1                # llm_rl_scripts/maze/ppo/train_ppo_online.py (device-resident loop)
    python scripts/harness.py bc-eval                                                # llm_rl_scripts/wordle/bc/eval_bc_gpt2.py
    python scripts/harness.py maze-eval                                              # llm_rl_scripts/maze/bc/fully_observed_bc.py (eval part)
    python scripts/harness.py gen-data --n-data 1000 --out d.jsonl                   # llm_rl_scripts/wordle/misc/data_gen.py

Hyper-parameter names and defaults are the reference scripts' own.  Not carried over: tyro, wandb, GCS, mesh-shape flags (one
process per GPU + `torch.distributed.run` instead).  `--model` is a checkpoint directory in the reference layout
(`lmrl_gym_amd.checkpoints`) or `random:<tiny|small|medium>`; the GPT-2 tokenizer is `transformers.AutoTokenizer('gpt2')` when its
files are available, else the Wordle-alphabet adapter (`datasets.WordleTokenizer`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ILQL_DEFAULTS = dict(epochs=10, max_steps=None, lr=3e-5, weight_decay=0.0, train_bsize=32, grad_accum_steps=1, max_length=512, log_every=256,
                     policy_max_input_length=256, policy_max_output_length=256, policy_do_sample=True, policy_temperature=None,
                     policy_top_p=None, policy_top_k=None, policy_bsize=32, policy_n_rollouts=32, beta=32.0, polyak_alpha=0.005,
                     hard_update_every=None, gamma=0.99, tau=0.7, cql_weight=0.01, bad_word_reward=-10.0,
                     bf16_activations=False, gradient_checkpointing=False)                                   # train_ilql_gpt2.py:41-117, :369
PPO_DEFAULTS = dict(n_rounds=1, epochs=1, max_steps=None, lr=1e-5, weight_decay=0.0, train_bsize=32, train_bc_bsize=None, grad_accum_steps=None,
                    rollout_bsize=32, n_rollouts=128, ppo_data_bsize=32, max_input_length=512, max_output_length=512, policy_do_sample=True,
                    policy_temperature=None, policy_top_p=None, policy_top_k=None, gamma=1.0, lam=0.95, use_advantage_whitening=True,
                    init_kl_coef=0.001, kl_target=None, kl_horizon=None, cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0,
                    bc_loss_weight=1.0, bad_word_reward=-10.0, bf16_activations=False, gradient_checkpointing=False)   # train_ppo_gpt2.py:40-112, :211
FILTERED_BC_DEFAULTS = dict(n_rounds=1, epochs=1, max_steps=None, lr=1e-5, weight_decay=0.0, train_bsize=32, grad_accum_steps=None, rollout_bsize=32,
                            n_rollouts=128, filter_percengage=0.3, bf16_activations=False, gradient_checkpointing=False, max_input_length=512,
                            max_output_length=512, policy_do_sample=True, policy_temperature=None, policy_top_p=None, policy_top_k=None,
                            bad_word_reward=-10.0)                                    # wordle/online_filtered_bc/train_online_filtered_bc_gpt2.py:40-87
BC_EVAL_DEFAULTS = dict(policy_n_rollouts=32, policy_bsize=1, policy_max_input_length=256, policy_max_output_length=256, policy_do_sample=True,
                        policy_temperature=None, policy_top_p=None, policy_top_k=None)                         # eval_bc_gpt2.py
MAZE_EVAL_DEFAULTS = dict(maze_name="double_t_maze", describe_function="describe_observation_only_walls", reward_function="standard_reward",
                          last_k=1, max_steps=100, generation_bsize=4, max_input_length=128, max_output_length=8)   # fully_observed_bc.py:230-283


def _add(ap: argparse.ArgumentParser, defaults: dict):
    for k, v in defaults.items():
        flag = "--" + k.replace("_", "-")
        if isinstance(v, bool):
            ap.add_argument(flag, type=lambda s: s.lower() in ("1", "true", "yes"), default=v)
        elif v is None:
            ap.add_argument(flag, type=float, default=None)
        else:
            ap.add_argument(flag, type=type(v), default=v)


def _tokenizer(wordle: bool = True):
    from lmrl_gym_amd.datasets import ByteTokenizer, WordleTokenizer
    try:
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained("gpt2", local_files_only=True)
        tok.add_special_tokens({"pad_token": "<|pad|>"})
        return tok
    except Exception:
        return WordleTokenizer() if wordle else ByteTokenizer()


def _model(spec: str, vocab: int):
    from lmrl_gym_amd import checkpoints as C
    from lmrl_gym_amd.gpt2 import GPT2Config, init_hf_style_state_dict
    if spec.startswith("random:"):
        cfg = dict(tiny=GPT2Config(2, 2, 128, 256, vocab, 128), small=GPT2Config.gpt2_small(vocab), medium=GPT2Config.gpt2_medium(vocab))[spec.split(":")[1]]
        return cfg, init_hf_style_state_dict(cfg, seed=0)
    import torch
    loader = C.load_hf_pytorch_gpt2 if os.path.exists(os.path.join(spec, "model.safetensors")) or os.path.exists(os.path.join(spec, "pytorch_model.bin")) \
        else C.load_gpt2_checkpoint
    cfg, sd = loader(spec)
    return cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def _engine(cfg, params):
    from lmrl_gym_amd import _lib
    from lmrl_gym_amd.gpt2 import GPT2Engine
    return GPT2Engine(cfg, {k: (v.detach().cpu() if hasattr(v, "detach") else v) for k, v in params.items()}, _lib.require_gpu())


def _wordle_env(a):
    from lmrl_gym_amd.envs import wordle as W
    vocab = W.Vocabulary.from_file(a.vocab_file) if os.path.exists(a.vocab_file) else W.Vocabulary.builtin(a.vocab_file)
    return vocab, W.ReformatWordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=getattr(a, "bad_word_reward", -1.0))


def _device_rollouts(engine, vocab, tok, bsize, bad_word_reward, max_new_tokens):
    """Wordle rollouts with env + policy + loop on the device (`WordleRolloutEngine`); needs single-token letters/symbols, i.e. the
    GPT-2 tokenizer or the Wordle adapter."""
    from lmrl_gym_amd.rollout import WordleRolloutEngine, WordleTokenTable
    table = getattr(tok, "table", None) or WordleTokenTable.from_tokenizer(tok)
    return WordleRolloutEngine(engine, vocab, bsize, tokens=table, max_new_tokens=max_new_tokens, bad_word_reward=bad_word_reward)


def _log(tag, obj):
    print(json.dumps({tag: obj}, default=lambda o: float(o) if isinstance(o, (np.floating, np.integer)) else str(o)), flush=True)


# --------------------------------------------------------------------------------------------------------------- commands
def cmd_gen_data(a):
    from lmrl_gym_amd import datasets as DS
    vocab, _ = _wordle_env(a)
    n = DS.write_jsonl(a.out, DS.generate_wordle_dataset(vocab, a.n_data, a.prob_smart, seed=a.seed))
    _log("gen_data", dict(items=n, path=a.out))


def cmd_ilql(a):
    import torch
    from lmrl_gym_amd import _lib, datasets as DS, environment as E
    from lmrl_gym_amd.algorithms import ilql
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.policies import GPT2ValuePolicy, heads_to_engine_layout
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    dev = _lib.require_gpu()
    tok = _tokenizer()
    cfg, sd = _model(a.model, max(len(tok), 50257))
    mm = dict(matmul="bf16" if a.bf16_activations else "f32")                 # the scripts' bf16_activations / gradient_checkpointing flags
    base = GPT2F32(sd, cfg.n_head, device=dev, gradient_checkpointing=a.gradient_checkpointing, **mm)
    target_base, pi_beta = GPT2F32(sd, cfg.n_head, device=dev, **mm), _engine(cfg, sd)
    d, V = cfg.d_model, cfg.vocab
    g = torch.Generator().manual_seed(0)
    # MLPHead init of train_ilql_gpt2.py:224-231: layer-2 kernel 0, layer-2 bias -4.4 (Q heads) ; V head likewise
    mk = lambda out: MLPHeadF32({"dense1.kernel": torch.randn(d, d, generator=g) * 0.02, "dense1.bias": torch.zeros(d),
                                 "dense2.kernel": torch.zeros(d, out), "dense2.bias": torch.full((out,), -4.4)}, dev, **mm)
    tr = ilql.GPT2ILQLTrain(base, mk(V), mk(V), mk(1), tok.pad_token_id, dict(gamma=a.gamma, tau=a.tau, cql_weight=a.cql_weight),
                            target_base=target_base, lr=a.lr, weight_decay=a.weight_decay, grad_accum_steps=a.grad_accum_steps,
                            polyak_alpha=a.polyak_alpha, hard_update_every=None if a.hard_update_every is None else int(a.hard_update_every))
    bs = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, a.max_length)
    train = DS.ilql_dataset_from_jsonl(a.train_data, tok, bs)
    vocab, env = _wordle_env(a)

    def evaluate():
        hl = lambda h: heads_to_engine_layout({k: v.detach().cpu() for k, v in h.p.items()}, cfg.vocab_padded, dev)
        if a.device_rollouts:       # ILQL value policy, env and loop on the GPU
            from lmrl_gym_amd.rollout import WordleRolloutEngine, WordleTokenTable
            table = getattr(tok, "table", None) or WordleTokenTable.from_tokenizer(tok)
            ro = WordleRolloutEngine(pi_beta, vocab, a.policy_bsize, tokens=table, max_new_tokens=min(a.policy_max_output_length, 12),
                                     bad_word_reward=a.bad_word_reward, value_engine=_engine(cfg, base.p), q1_head=hl(tr.q1), q2_head=hl(tr.q2),
                                     beta=a.beta)
            _, summary = ro.text_env_eval(a.policy_n_rollouts, seed_generator=iter(range(10 ** 9)), temperature=a.policy_temperature or 1.0,
                                          top_k=int(a.policy_top_k or 0))
            ro.close()
            return summary
        pol = GPT2ValuePolicy(pi_beta, _engine(cfg, base.p), hl(tr.q1), hl(tr.q2), a.beta, tok, max_input_length=a.policy_max_input_length,
                              max_new_tokens=a.policy_max_output_length, do_sample=a.policy_do_sample, temperature=a.policy_temperature,
                              top_k=None if a.policy_top_k is None else int(a.policy_top_k), top_p=a.policy_top_p, eos_token_id=tok.encode("\n")[0],
                              out_str_process=lambda x: x.removesuffix("\n") + "\n")
        _, summary = E.text_env_eval(env, pol, n_rollouts=a.policy_n_rollouts, bsize=a.policy_bsize, seed_generator=iter(range(10 ** 9)), verbose=False)
        return summary

    step = 0
    for epoch in range(a.epochs):
        for batch in DS.dataloader(np.random.default_rng(epoch), train, a.train_bsize, truncate=True):
            _, loss, logs = tr.step(batch["input_ids"], batch["should_take_action"], batch["rewards"], batch["dones"],
                                    next_token_ids=batch["next_token_ids"], next_dones=batch["next_dones"])
            step += 1
            if step % a.log_every == 0 or step == 1:
                _log("train", dict(step=step, epoch=epoch, loss=loss, losses=logs["losses"]))
            if a.max_steps is not None and step >= int(a.max_steps):
                break
        if a.max_steps is not None and step >= int(a.max_steps):
            break
    _log("eval", evaluate())
    if a.out:
        from lmrl_gym_amd import checkpoints as C
        C.save_gpt2_checkpoint(os.path.join(a.out, "base"), cfg, base.p)
        for name, h in (("q1_head", tr.q1), ("q2_head", tr.q2), ("v_head", tr.v)):
            C.save_head_checkpoint(os.path.join(a.out, name), h.p)


def cmd_ppo(a):
    import torch
    from lmrl_gym_amd import _lib, datasets as DS, environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.algorithms.ppo_inference import (GPT2PPOInference, text_trajectory_chains_from_interactions, text_trajectory_chains_from_transitions,
                                                       text_trajectory_chains_partially_observed)
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    tok = _tokenizer(wordle=a.env == "wordle")       # FEN / SAN text needs every byte: the byte tokenizer when no GPT-2 files are around
    cfg, sd = _model(a.model, max(len(tok), 50257))
    mm = dict(matmul="bf16" if a.bf16_activations else "f32")
    pol_f32 = GPT2F32(sd, cfg.n_head, device=dev, gradient_checkpointing=a.gradient_checkpointing, **mm)
    init_f32 = GPT2F32(sd, cfg.n_head, device=dev, **mm)
    head = LinearHeadF32(dict(kernel=torch.zeros(cfg.d_model, 1), bias=torch.tensor([-4.1])), dev)       # train_ppo_gpt2.py:254-260
    kw = dict(cliprange_value=a.cliprange_value, cliprange=a.cliprange, value_loss_coef=a.value_loss_coef)
    max_len = a.max_input_length + a.max_output_length
    policy = GPT2PPOPolicy(_engine(cfg, sd), tok, max_input_length=a.max_input_length, max_new_tokens=a.max_output_length, do_sample=a.policy_do_sample,
                           temperature=a.policy_temperature, top_k=None if a.policy_top_k is None else int(a.policy_top_k), top_p=a.policy_top_p,
                           eos_token_id=tok.encode("\n")[0], out_str_process=lambda x: x.removesuffix("\n") + "\n")
    inf = GPT2PPOInference(pol_f32, head, tok.pad_token_id, initial_policy=init_f32, tokenizer=tok, loss_kwargs=kw, bc_loss_weight=a.bc_loss_weight)
    tr = ppo.GPT2PPOTrain(pol_f32, head, tok.pad_token_id, kw, lr=a.lr, weight_decay=a.weight_decay, grad_accum_steps=int(a.grad_accum_steps or 1),
                          bc_loss_weight=a.bc_loss_weight)
    ctl = ppo.AdaptiveKLController(a.init_kl_coef, a.kl_target, int(a.kl_horizon)) if a.kl_target is not None and a.kl_horizon is not None \
        else ppo.FixedKLController(a.init_kl_coef)
    bs = BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_len)
    bc = DS.MaskDataset.from_jsonl(a.bc_data, tok, bs) if a.bc_data else None
    if a.env == "chess":        # configs[3]: online PPO against the chess env (chess/ppo/train_ppo_gpt2_online.py:201-222)
        from lmrl_gym_amd.envs import chess as C
        if a.device_rollouts:
            raise SystemExit("--device-rollouts is the Wordle token loop; the chess env steps its boards on the device through the text protocol")
        start = C.large_piece_random_endgame(a.chess_pieces) if a.chess_pieces else None
        env = C.FenChessHistoryEnv(max_moves=a.chess_max_moves, from_position=start, random_opponent=bool(a.chess_random_opponent),
                                   **({} if a.chess_random_opponent else dict(engine_path=a.chess_engine, engine_options={"Use NNUE": a.chess_use_nnue},
                                                                              movetime_ms=a.chess_movetime_ms)))
        vocab = None
    elif a.env == "maze":       # configs[0]'s online script: llm_rl_scripts/maze/ppo/train_ppo_online.py:260-300, 431-483
        from lmrl_gym_amd.envs import maze as M
        env = M.setup_maze_env(a.maze_name, a.maze_describe_function, a.maze_reward_function, last_k=a.maze_last_k, max_steps=a.maze_max_steps)
        vocab = None
    else:
        vocab, env = _wordle_env(a)
    step = 0
    limit = None if a.max_steps is None else int(a.max_steps)
    # --device-rollouts: the whole iteration stays in HBM — rollout records -> PPO data -> device batches -> train steps -> the trainer's
    # parameters copied into the engine in place (WordleRolloutEngine.ppo_rollouts, algorithms/ppo_device.py); --resident 0: the host-array path
    resident = bool(a.device_rollouts) and bool(a.resident)
    if resident and a.env == "maze":
        from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
        ro_res = MazeRolloutEngine(policy.engine, tok, env, a.rollout_bsize, max_new_tokens=min(a.max_output_length, 12), eos_token_id=tok.encode("\n")[0],
                                   max_input_length=a.max_input_length)
    else:
        ro_res = _device_rollouts(policy.engine, vocab, tok, a.rollout_bsize, a.bad_word_reward, min(a.max_output_length, 12)) if resident else None
    for rnd in range(a.n_rounds):
        if limit is not None and step >= limit:
            break
        if resident:
            warp = dict(top_k=int(a.policy_top_k or 0)) if a.env == "maze" else dict(top_k=int(a.policy_top_k or 0), top_p=float(a.policy_top_p or 0.0))
            ds, kls, summary = ro_res.ppo_rollouts(inf, a.n_rollouts, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)), gamma=a.gamma, lam=a.lam,
                                                   kl_weight=ctl.value, max_length=max_len, use_advantage_whitening=a.use_advantage_whitening,
                                                   temperature=a.policy_temperature or 1.0, sample_seed=rnd, bsize=max(a.ppo_data_bsize, 64), **warp)
            mean_kl = float(kls.cpu().numpy().mean()) if kls.numel() else 0.0
            ctl.update(mean_kl, a.train_bsize)
            _log("data_collection", dict(round=rnd, env_interaction=summary, mean_kl=mean_kl, kl_ctrl_value=ctl.value, n_chains=len(ds), device_resident=True))
            bc_iter = None
            for epoch in range(a.epochs):
                if limit is not None and step >= limit:
                    break
                for batch in ds.batches(np.random.default_rng(rnd * 1000 + epoch), min(a.train_bsize, len(ds)), truncate=True,
                                        width=ds.trimmed_width() if a.trim_batches else None):
                    extra = {}
                    if bc is not None:
                        if bc_iter is None:
                            bc_iter = DS.dataloader(np.random.default_rng(7 + step), bc, int(a.train_bc_bsize or a.train_bsize), truncate=True)
                        try:
                            bb = next(bc_iter)
                        except StopIteration:
                            bc_iter = DS.dataloader(np.random.default_rng(7 + step), bc, int(a.train_bc_bsize or a.train_bsize), truncate=True)
                            bb = next(bc_iter)
                        extra = dict(bc_data_input_ids=bb["input_ids"], bc_data_input_training_mask=bb["input_training_mask"])
                    _, loss, logs = tr.step(**batch, **extra)
                    step += 1
                    _log("train", dict(step=step, round=rnd, loss=loss))
                    if limit is not None and step >= limit:
                        break
            ro_res.load_params(pol_f32.p)
            continue
        if a.device_rollouts and a.env == "maze":
            from lmrl_gym_amd.maze_rollout import MazeRolloutEngine
            ro = MazeRolloutEngine(policy.engine, tok, env, a.rollout_bsize, max_new_tokens=min(a.max_output_length, 12), eos_token_id=tok.encode("\n")[0],
                                   max_input_length=a.max_input_length)
            raw, summary = ro.text_env_eval(a.n_rollouts, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)), temperature=a.policy_temperature or 1.0,
                                            top_k=int(a.policy_top_k or 0), sample_seed=rnd)
            ro.close()
        elif a.device_rollouts:     # env + policy + lock-step loop on the GPU; same (interactions, summary) as text_env_eval
            ro = _device_rollouts(policy.engine, vocab, tok, a.rollout_bsize, a.bad_word_reward, min(a.max_output_length, 12))
            raw, summary = ro.text_env_eval(a.n_rollouts, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)), temperature=a.policy_temperature or 1.0,
                                            top_k=int(a.policy_top_k or 0), top_p=float(a.policy_top_p or 0.0), sample_seed=rnd, concurrent=a.rollout_lanes)
            ro.close()
        else:
            raw, summary = E.text_env_eval(env, policy, n_rollouts=a.n_rollouts, bsize=a.rollout_bsize, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)),
                                           verbose=False)
        if a.env == "maze" and a.maze_last_k != 1:      # item windows: the partially observed script's joined-state chains (partially_observed_ppo_online.py:372-398)
            chains = text_trajectory_chains_partially_observed(raw)
        else:
            chains = text_trajectory_chains_from_transitions(raw) if a.env in ("chess", "maze") else text_trajectory_chains_from_interactions(raw, tok, max_len, a.gamma)
        datas, kls = inf.get_ppo_data_from_text_trajectory_chain(chains, bsize=a.ppo_data_bsize, max_length=max_len, gamma=a.gamma, lam=a.lam,
                                                                 kl_weight=ctl.value, use_advantage_whitening=a.use_advantage_whitening)
        mean_kl = float(kls.mean()) if len(kls) else 0.0
        ctl.update(mean_kl, a.train_bsize)
        _log("data_collection", dict(round=rnd, env_interaction=summary, mean_kl=mean_kl, kl_ctrl_value=ctl.value, n_chains=len(chains)))
        ds = ppo.PPODataset.from_ppo_data_list(datas, tok, bs)
        bc_iter = None
        for epoch in range(a.epochs):
            if limit is not None and step >= limit:
                break
            for batch in DS.dataloader(np.random.default_rng(rnd * 1000 + epoch), ds, min(a.train_bsize, len(ds)), truncate=True):
                extra = {}
                if bc is not None:
                    if bc_iter is None:
                        bc_iter = DS.dataloader(np.random.default_rng(7 + step), bc, int(a.train_bc_bsize or a.train_bsize), truncate=True)
                    try:
                        bb = next(bc_iter)
                    except StopIteration:
                        bc_iter = DS.dataloader(np.random.default_rng(7 + step), bc, int(a.train_bc_bsize or a.train_bsize), truncate=True)
                        bb = next(bc_iter)
                    extra = dict(bc_data_input_ids=bb["input_ids"], bc_data_input_training_mask=bb["input_training_mask"])
                _, loss, logs = tr.step(**batch, **extra)
                step += 1
                _log("train", dict(step=step, round=rnd, loss=loss))
                if a.max_steps is not None and step >= int(a.max_steps):
                    break
        policy.set_params(_engine(cfg, pol_f32.p))
    _, summary = E.text_env_eval(env, policy, n_rollouts=a.n_rollouts, bsize=a.rollout_bsize, seed_generator=iter(range(10 ** 8, 10 ** 9)), verbose=False)
    _log("eval", summary)


def cmd_filtered_bc(a):
    """Online filtered BC (LLM_RL/algorithms/online_filtered_bc/train.py + wordle/online_filtered_bc/train_online_filtered_bc_gpt2.py:198-278):
    every round: rollouts with the current policy, keep the top `filter_percengage` episodes by total reward, BC on their action tokens."""
    import torch
    from lmrl_gym_amd import _lib, datasets as DS, environment as E
    from lmrl_gym_amd.algorithms import bc
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32
    dev = _lib.require_gpu()
    tok = _tokenizer()
    cfg, sd = _model(a.model, max(len(tok), 50257))
    model = GPT2F32(sd, cfg.n_head, device=dev, matmul="bf16" if a.bf16_activations else "f32", gradient_checkpointing=a.gradient_checkpointing)
    tr = bc.GPT2BCTrain(model, tok.pad_token_id, lr=a.lr, weight_decay=a.weight_decay, grad_accum_steps=int(a.grad_accum_steps or 1))
    policy = GPT2PPOPolicy(_engine(cfg, sd), tok, max_input_length=a.max_input_length, max_new_tokens=a.max_output_length, do_sample=a.policy_do_sample,
                           temperature=a.policy_temperature, top_k=None if a.policy_top_k is None else int(a.policy_top_k), top_p=a.policy_top_p,
                           eos_token_id=tok.encode("\n")[0], out_str_process=lambda x: x.removesuffix("\n") + "\n")
    vocab, env = _wordle_env(a)
    bs = BlockingStrategy(Padding.RIGHT, Truncation.LEFT, a.max_input_length + a.max_output_length)
    step = 0
    limit = None if a.max_steps is None else int(a.max_steps)
    for rnd in range(a.n_rounds):
        if limit is not None and step >= limit:      # the step budget ends the run: no further rollouts, epochs or rounds
            break
        if a.device_rollouts:
            ro = _device_rollouts(policy.engine, vocab, tok, a.rollout_bsize, a.bad_word_reward, min(a.max_output_length, 12))
            raw, summary = ro.text_env_eval(a.n_rollouts, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)), temperature=a.policy_temperature or 1.0,
                                            top_k=int(a.policy_top_k or 0), top_p=float(a.policy_top_p or 0.0), sample_seed=rnd, concurrent=a.rollout_lanes)
            ro.close()
        else:
            raw, summary = E.text_env_eval(env, policy, n_rollouts=a.n_rollouts, bsize=a.rollout_bsize, seed_generator=iter(range(rnd * 10 ** 6, 10 ** 9)),
                                           verbose=False)
        segs = [[(t.text, float(t.is_action)) for t in ep[-1].post_transition_history] for ep in raw]
        rewards = [sum(t.reward for t in ep) for ep in raw]
        top = sorted(range(len(segs)), key=lambda i: rewards[i], reverse=True)[: int(len(segs) * a.filter_percengage)]
        _log("data_collection", dict(round=rnd, env_interaction=summary, kept=len(top), reward_cutoff=min((rewards[i] for i in top), default=None)))
        if not top:
            continue
        ds = DS.MaskDataset.blocked_from_str_segments([segs[i] for i in top], tok, bs)
        for epoch in range(a.epochs):
            if limit is not None and step >= limit:
                break
            for batch in DS.dataloader(np.random.default_rng(rnd * 1000 + epoch), ds, min(a.train_bsize, len(ds)), truncate=True):
                _, loss, _ = tr.step(batch["input_ids"], batch["input_training_mask"] > 0)
                step += 1
                _log("train", dict(step=step, round=rnd, loss=loss))
                if limit is not None and step >= limit:
                    break
        policy.set_params(_engine(cfg, model.p))
    _, summary = E.text_env_eval(env, policy, n_rollouts=a.n_rollouts, bsize=a.rollout_bsize, seed_generator=iter(range(10 ** 8, 10 ** 9)), verbose=False)
    _log("eval", summary)


def cmd_bc_eval(a):
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    tok = _tokenizer()
    cfg, sd = _model(a.model, max(len(tok), 50257))
    policy = GPT2PPOPolicy(_engine(cfg, sd), tok, max_input_length=a.policy_max_input_length, max_new_tokens=a.policy_max_output_length,
                           do_sample=a.policy_do_sample, temperature=a.policy_temperature, top_k=None if a.policy_top_k is None else int(a.policy_top_k),
                           top_p=a.policy_top_p, eos_token_id=tok.encode("\n")[0], out_str_process=lambda x: x.removesuffix("\n") + "\n")
    vocab, env = _wordle_env(a)
    if a.device_rollouts:       # same call, env + policy + loop on the GPU
        ro = _device_rollouts(policy.engine, vocab, tok, a.policy_bsize, -1.0, min(a.policy_max_output_length, 12))
        _, summary = ro.text_env_eval(a.policy_n_rollouts, seed_generator=iter(range(10 ** 9)), temperature=a.policy_temperature or 1.0,
                                      top_k=int(a.policy_top_k or 0), concurrent=a.rollout_lanes)
        ro.close()
    else:
        _, summary = E.text_env_eval(env, policy, n_rollouts=a.policy_n_rollouts, bsize=a.policy_bsize, seed_generator=iter(range(10 ** 9)), verbose=False)
    _log("eval", summary)


def cmd_maze_eval(a):
    from lmrl_gym_amd import environment as E
    from lmrl_gym_amd.envs import maze as M
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    tok = _tokenizer(wordle=False)
    cfg, sd = _model(a.model, max(len(tok), 50257))
    policy = GPT2PPOPolicy(_engine(cfg, sd), tok, max_input_length=a.max_input_length, max_new_tokens=a.max_output_length, do_sample=False,
                           eos_token_id=tok.encode("\n")[0], out_str_process=lambda x: x.removesuffix("\n") + "\n")
    env = M.setup_maze_env(a.maze_name, a.describe_function, a.reward_function, last_k=a.last_k, max_steps=a.max_steps)
    starts = [tuple(p) for p in np.argwhere(M.double_t_maze() == 0).tolist()]        # one rollout per free cell (fully_observed_bc.py:262-283)
    results = []
    for i in range(0, len(starts), a.generation_bsize):
        opts = [dict(init_position=p) for p in starts[i:i + a.generation_bsize]]
        inter = E.interact_environment(env, policy, env_seed=[None] * len(opts), env_options=opts, bsize=len(opts))
        results += [dict(start=starts[i + k], reward=sum(t.reward for t in ep), steps=len(ep), done=ep[-1].done) for k, ep in enumerate(inter)]
    _log("maze_eval", dict(avg_reward=float(np.mean([r["reward"] for r in results])), avg_steps=float(np.mean([r["steps"] for r in results])),
                           n=len(results), move_accuracy=M.compute_move_accuracy(policy)))


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name, fn, defaults in (("ilql", cmd_ilql, ILQL_DEFAULTS), ("ppo", cmd_ppo, PPO_DEFAULTS), ("filtered-bc", cmd_filtered_bc, FILTERED_BC_DEFAULTS),
                               ("bc-eval", cmd_bc_eval, BC_EVAL_DEFAULTS),
                               ("maze-eval", cmd_maze_eval, MAZE_EVAL_DEFAULTS)):
        p = sub.add_parser(name)
        p.set_defaults(fn=fn)
        p.add_argument("--model", default="random:tiny", help="checkpoint directory (reference layout or HF PyTorch) or random:<tiny|small>")
        p.add_argument("--vocab-file", default="wordle_official_400.txt")
        p.add_argument("--out", default=None)
        p.add_argument("--device-rollouts", type=int, default=0, help="1 = run the Wordle rollouts (bc-eval, ilql evaluation, ppo data collection) on the device-resident engine")
        p.add_argument("--rollout-lanes", type=int, default=1, help="device rollouts: independent episode batches in flight at once "
                                                                     "(WordleRolloutEngine.text_env_eval(concurrent=n); >= 4 batches use the graph path)")
        _add(p, defaults)
    sub.choices["ilql"].add_argument("--train-data", required=True)
    sub.choices["ilql"].add_argument("--eval-data", default=None)
    sub.choices["ppo"].add_argument("--bc-data", default=None)
    pp = sub.choices["ppo"]
    pp.add_argument("--env", default="wordle", choices=["wordle", "chess", "maze"])
    pp.add_argument("--maze-name", default="double_t_maze"); pp.add_argument("--maze-describe-function", default="describe_observation_give_position")
    pp.add_argument("--maze-reward-function", default="standard_reward"); pp.add_argument("--maze-last-k", type=int, default=1)
    pp.add_argument("--maze-max-steps", type=int, default=100)
    pp.add_argument("--resident", type=int, default=1, help="with --device-rollouts 1: 1 (default) = the device-resident iteration, 0 = device rollouts feeding the host-array PPO data path")
    pp.add_argument("--trim-batches", type=int, default=1, help="device-resident loop: train on batches cut to the round's longest episode (multiple of 64) instead of "
                                                                  "max_input_length + max_output_length columns — same loss and gradients, fewer padded rows")
    pp.add_argument("--chess-engine", default=os.environ.get("CHESS_ENGINE_PATH"), help="UCI engine binary (the reference: stockfish/stockfish-ubuntu-20.04-x86-64-avx2)")
    pp.add_argument("--chess-use-nnue", default="true", help="'false' for a binary built without the net file")
    pp.add_argument("--chess-random-opponent", type=int, default=0)
    pp.add_argument("--chess-pieces", default=None, help="e.g. kQK: start every round from a random endgame of this material (large_piece_random_endgame)")
    pp.add_argument("--chess-max-moves", type=int, default=400)
    pp.add_argument("--chess-movetime-ms", type=int, default=100)
    g = sub.add_parser("gen-data")
    g.set_defaults(fn=cmd_gen_data)
    g.add_argument("--n-data", type=int, default=1000); g.add_argument("--prob-smart", type=float, default=0.5)
    g.add_argument("--seed", type=int, default=0); g.add_argument("--out", required=True); g.add_argument("--vocab-file", default="wordle_official_400.txt")
    return ap


def main(argv=None):
    a = build_parser().parse_args(argv)
    a.fn(a)


if __name__ == "__main__":
    main()
