"""GPU tier: the pieces compose the way the reference's task scripts use them at toy scale:
device data generation -> jsonl -> ILQL / MC / BC datasets -> train steps -> value policy on the HIP engine -> text_env_eval."""
import numpy as np
import pytest

import lmrl_gym_amd  # noqa: F401

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_generated_dataset_replays_on_the_oracle_env(tmp_path):
    from lmrl_gym_amd import datasets as DS
    from lmrl_gym_amd.envs import wordle as W
    from oracle.wordle import OracleWordleEnv
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    items, seeds = DS.generate_wordle_dataset(vocab, 150, prob_smart=0.7, seed=4, bsize=64, return_seeds=True)
    assert len(items) == 150 and len(seeds) == 150
    wins = 0
    for it, sd in zip(items, seeds):
        o = OracleWordleEnv(vocab.all_vocab, True, -1.0)
        hist = o.reset(sd)
        seq, rew = it["sequence"], it["reward"]
        assert seq[0] == ("Wordle:\n", 0.0) and len(rew) == len(seq) - 1 and it["done"]
        done, k = False, 1
        while not done:
            hist, r, done = o.step(hist + ((seq[k][0], True),))
            assert seq[k][1] == 1.0 and seq[k + 1] == (hist[-1][0], 0.0)
            assert rew[k - 1] == float(r) and rew[k] == 0.0
            k += 2
        assert k == len(seq)
        wins += rew[-2] == 0.0
    assert wins > 0          # the "smart" branch does solve some games
    p = tmp_path / "wordle.jsonl"
    DS.write_jsonl(str(p), items)
    tok = DS.WordleTokenizer()
    mc = DS.mc_data_from_jsonl(str(p), tok, gamma=1.0)
    for d, it in zip(mc[:20], items[:20]):
        total = sum(it["reward"])
        first_action = int(np.argmax(d.should_take_action))
        assert abs(float(d.returns[first_action]) - total) < 1e-5      # undiscounted return-to-go of the first action token


def test_toy_ilql_pipeline_end_to_end(tmp_path):
    from lmrl_gym_amd import datasets as DS, environment as E
    from lmrl_gym_amd.algorithms import ilql
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.policies import GPT2ValuePolicy, heads_to_engine_layout
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, MLPHeadF32
    from lmrl_gym_amd import _lib
    dev = _lib.require_gpu()
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    tok = DS.WordleTokenizer()
    p = tmp_path / "train.jsonl"
    DS.write_jsonl(str(p), DS.generate_wordle_dataset(vocab, 64, prob_smart=0.5, seed=1))
    ds = DS.ilql_dataset_from_jsonl(str(p), tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, 80))
    cfg = GPT2Config(2, 2, 128, 256, 50257, 128)
    sd = init_hf_style_state_dict(cfg, seed=0)
    base = GPT2F32(sd, cfg.n_head, device=dev)
    d, V = cfg.d_model, cfg.vocab
    g = torch.Generator().manual_seed(0)
    mk = lambda out, b2: MLPHeadF32({"dense1.kernel": torch.randn(d, d, generator=g) * 0.05, "dense1.bias": torch.zeros(d),
                                     "dense2.kernel": torch.zeros(d, out), "dense2.bias": torch.full((out,), b2)}, dev)
    tr = ilql.GPT2ILQLTrain(base, mk(V, -4.4), mk(V, -4.4), mk(1, -4.4), tok.pad_token_id, dict(gamma=0.99, tau=0.7, cql_weight=0.01), lr=1e-3)
    losses = []
    for epoch in range(2):
        for batch in DS.dataloader(np.random.default_rng(epoch), ds, 16):
            _, loss, logs = tr.step(batch["input_ids"], batch["should_take_action"], batch["rewards"], batch["dones"])
            assert np.isfinite(loss)
            losses.append(loss)
    assert len(losses) == 8 and losses[-1] < losses[0]
    # trained weights -> bf16 rollout engine -> ILQL value policy -> evaluation on the (reformatted) Wordle env
    eng = GPT2Engine(cfg, {k: v.detach().cpu() for k, v in base.p.items()}, dev)
    hl = lambda h: heads_to_engine_layout({k: v.detach().cpu() for k, v in h.p.items()}, cfg.vocab_padded, dev)
    pol = GPT2ValuePolicy(eng, eng, hl(tr.q1), hl(tr.q2), 4.0, tok, max_input_length=96, max_new_tokens=8, do_sample=True, seed=3,
                          eos_token_id=tok.eos_token_id, out_str_process=lambda x: x.removesuffix("\n") + "\n")
    env = W.ReformatWordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=-10.0)
    inter, summary = E.text_env_eval(env, pol, n_rollouts=6, bsize=3, seed_generator=iter(range(100)), verbose=False)
    assert len(inter) == 6 and all(ep[-1].done for ep in inter)
    assert set(summary["reward"]) >= {"mean", "std", "min", "max"} and np.isfinite(summary["reward"]["mean"])


def test_checkpoint_roundtrip_drives_the_engine(tmp_path):
    """weights -> reference checkpoint layout (flax msgpack + config.json) -> loaded back -> rollout engine: identical
    hidden states to the engine built from the original state dict."""
    from lmrl_gym_amd import _lib, checkpoints as C
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    dev = _lib.require_gpu()
    cfg = GPT2Config(2, 2, 128, 256, 500, 64)
    sd = init_hf_style_state_dict(cfg, seed=9)
    C.save_gpt2_checkpoint(str(tmp_path / "policy"), cfg, sd)
    cfg2, sd2 = C.load_gpt2_checkpoint(str(tmp_path / "policy"))
    e1, e2 = GPT2Engine(cfg, sd, dev), GPT2Engine(cfg2, {k: torch.from_numpy(v) for k, v in sd2.items()}, dev)
    toks = torch.randint(0, 500, (4 * 8,), generator=torch.Generator().manual_seed(0)).to(torch.int32).to(dev)
    cnt = torch.tensor([8, 3, 8, 5], dtype=torch.int32, device=dev)
    h = [e.session(4, 16).forward(toks, cnt, 8).clone() for e in (e1, e2)]
    assert torch.equal(h[0], h[1])


def test_toy_online_ppo_round(tmp_path):
    """One round of the online PPO loop of llm_rl_scripts/wordle/ppo/train_ppo_gpt2.py at toy scale: rollouts with the policy
    on the HIP engine -> text chains -> PPOData (both policies' log-probs, values, KL-penalised GAE, whitening) -> PPODataset
    -> train steps with the BC auxiliary batch -> new weights pushed back into the rollout policy."""
    from lmrl_gym_amd import _lib, datasets as DS, environment as E
    from lmrl_gym_amd.algorithms import ppo
    from lmrl_gym_amd.algorithms.common import BlockingStrategy, Padding, Truncation
    from lmrl_gym_amd.algorithms.ppo_inference import GPT2PPOInference, text_trajectory_chains_from_interactions
    from lmrl_gym_amd.envs import wordle as W
    from lmrl_gym_amd.gpt2 import GPT2Config, GPT2Engine, init_hf_style_state_dict
    from lmrl_gym_amd.policies import GPT2PPOPolicy
    from lmrl_gym_amd.train.gpt2_f32 import GPT2F32, LinearHeadF32
    dev = _lib.require_gpu()
    vocab = W.Vocabulary.builtin("wordle_official_400.txt")
    tok = DS.WordleTokenizer()
    cfg = GPT2Config(2, 2, 128, 256, 50257, 128)
    sd = init_hf_style_state_dict(cfg, seed=2)
    # bias the LM head towards the Wordle alphabet so that sampled actions decode to letters (a BC-initialised policy's role)
    letters = tok.table.letter_sp + tok.table.letter_first + [tok.table.newline]
    sd["wte.weight"][letters] += 1.0
    sd["ln_f.bias"] = sd["wte.weight"][letters].mean(0) * 0.5
    pol_f32, init_f32 = GPT2F32(sd, cfg.n_head, device=dev), GPT2F32(sd, cfg.n_head, device=dev)
    head = LinearHeadF32(dict(kernel=torch.zeros(cfg.d_model, 1), bias=torch.tensor([-4.1])), dev)
    max_len = 96
    policy = GPT2PPOPolicy(GPT2Engine(cfg, sd, dev), tok, max_input_length=72, max_new_tokens=12, do_sample=True, seed=1,
                           eos_token_id=tok.eos_token_id, out_str_process=lambda x: x.removesuffix("\n") + "\n")
    env = W.ReformatWordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=-10.0)
    raw, summary = E.text_env_eval(env, policy, n_rollouts=8, bsize=4, seed_generator=iter(range(1000)), verbose=False)
    assert len(raw) == 8 and np.isfinite(summary["reward"]["mean"])

    class AnyTok:                 # sampled text is arbitrary GPT-2 text only in a real run; here: Wordle alphabet or unknown -> pad id
        pad_token_id, eos_token_id = tok.pad_token_id, tok.eos_token_id

        def encode(self, s):
            try:
                return tok.encode(s)
            except AssertionError:
                return [tok.table.newline]
    chains = text_trajectory_chains_from_interactions(raw, AnyTok(), max_len, gamma=1.0)
    assert 1 <= len(chains) <= 8
    kw = dict(cliprange_value=0.2, cliprange=0.2, value_loss_coef=1.0)
    inf = GPT2PPOInference(pol_f32, head, tok.pad_token_id, initial_policy=init_f32, tokenizer=AnyTok(), loss_kwargs=kw, bc_loss_weight=1.0)
    ctl = ppo.AdaptiveKLController(init_kl_coef=0.001, target=0.1, horizon=10000)
    datas, kls = inf.get_ppo_data_from_text_trajectory_chain(chains, bsize=4, max_length=max_len, gamma=1.0, lam=0.95, kl_weight=ctl.value)
    assert np.allclose(kls, 0.0, atol=1e-5)               # policy == initial policy in round 0
    ctl.update(float(kls.mean()), 4)
    ds = ppo.PPODataset.from_ppo_data_list(datas, tok, BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, max_len))
    bc = DS.MaskDataset.blocked_from_str_segments([[("Wordle:\n", 0.0), ("s t a r e\n", 1.0), ("b b y b b\n", 0.0)]] * 4, tok,
                                                  BlockingStrategy(Padding.RIGHT, Truncation.RIGHT, 32))
    tr = ppo.GPT2PPOTrain(pol_f32, head, tok.pad_token_id, kw, lr=1e-4, bc_loss_weight=1.0)
    before = pol_f32.p["h.0.mlp.c_fc.weight"].clone()
    for batch in DS.dataloader(np.random.default_rng(0), ds, min(4, len(ds)), truncate=True):
        ev_loss, _ = inf.eval_loss(**batch, bc_data_input_ids=bc.input_ids, bc_data_input_training_mask=bc.input_training_mask)
        _, loss, logs = tr.step(**batch, bc_data_input_ids=bc.input_ids, bc_data_input_training_mask=bc.input_training_mask)
        assert np.isfinite(loss) and abs(ev_loss - loss) < 1e-5 * max(1.0, abs(loss)) and set(logs) == {"ppo", "bc", "total_loss"}
    assert not torch.equal(before, pol_f32.p["h.0.mlp.c_fc.weight"])
    policy.set_params(GPT2Engine(cfg, {k: v.detach().cpu() for k, v in pol_f32.p.items()}, dev))
    raw2, _ = E.text_env_eval(env, policy, n_rollouts=4, bsize=4, seed_generator=iter(range(4)), verbose=False)
    assert len(raw2) == 4


def test_harness_script_subcommands_at_toy_scale(tmp_path, capsys):
    """scripts/harness.py: every subcommand runs end to end with toy-sized overrides of the reference
    defaults (random-init tiny model, offline tokenizers)."""
    import importlib.util, json, os
    spec = importlib.util.spec_from_file_location("harness", os.path.join(os.path.dirname(__file__), "..", "scripts", "harness.py"))
    H = importlib.util.module_from_spec(spec); spec.loader.exec_module(H)
    data = str(tmp_path / "d.jsonl")
    H.main(["gen-data", "--n-data", "48", "--out", data, "--prob-smart", "0.6"])
    H.main(["ilql", "--train-data", data, "--epochs", "1", "--max-steps", "2", "--train-bsize", "8", "--max-length", "80", "--log-every", "1",
            "--policy-n-rollouts", "4", "--policy-bsize", "4", "--policy-max-input-length", "88", "--policy-max-output-length", "8", "--beta", "4",
            "--out", str(tmp_path / "ckpt")])
    H.main(["ilql", "--train-data", data, "--epochs", "1", "--max-steps", "1", "--train-bsize", "8", "--max-length", "80", "--policy-n-rollouts", "5",
            "--policy-bsize", "4", "--beta", "4", "--device-rollouts", "1"])
    assert os.path.exists(tmp_path / "ckpt" / "base" / "params.msgpack") and os.path.exists(tmp_path / "ckpt" / "q1_head" / "params.msgpack")
    H.main(["ppo", "--bc-data", data, "--n-rollouts", "4", "--rollout-bsize", "4", "--ppo-data-bsize", "4", "--train-bsize", "2", "--max-steps", "1",
            "--max-input-length", "72", "--max-output-length", "12"])
    H.main(["ppo", "--n-rollouts", "6", "--rollout-bsize", "4", "--ppo-data-bsize", "4", "--train-bsize", "2", "--max-steps", "1",
            "--max-input-length", "96", "--max-output-length", "6", "--device-rollouts", "1", "--resident", "0", "--policy-top-k", "50"])
    # the device-resident iteration (records -> PPO data -> device batches -> steps -> weights back in place), two rounds, BC batch, trimmed batches
    H.main(["ppo", "--bc-data", data, "--n-rollouts", "10", "--rollout-bsize", "4", "--train-bsize", "4", "--n-rounds", "2", "--max-steps", "3",
            "--max-input-length", "96", "--max-output-length", "6", "--device-rollouts", "1", "--trim-batches", "1", "--bf16-activations", "1",
            "--policy-top-k", "40", "--policy-top-p", "0.95"])
    # online filtered BC (wordle/online_filtered_bc): rollouts -> top 50 % by reward -> BC on the action tokens, text path and device loop
    H.main(["filtered-bc", "--n-rollouts", "6", "--rollout-bsize", "3", "--filter-percengage", "0.5", "--train-bsize", "2", "--max-steps", "1",
            "--max-input-length", "96", "--max-output-length", "8"])
    H.main(["filtered-bc", "--n-rollouts", "6", "--rollout-bsize", "4", "--filter-percengage", "0.5", "--train-bsize", "2", "--max-steps", "1",
            "--max-input-length", "96", "--max-output-length", "6", "--device-rollouts", "1", "--bf16-activations", "1"])
    # configs[3] at toy scale: online PPO against the chess env (random opponent; and the reference-built engine when present)
    H.main(["ppo", "--env", "chess", "--chess-random-opponent", "1", "--chess-pieces", "kQK", "--chess-max-moves", "3", "--n-rollouts", "4", "--rollout-bsize", "4",
            "--ppo-data-bsize", "4", "--train-bsize", "2", "--max-steps", "1", "--max-input-length", "160", "--max-output-length", "8"])
    # configs[0]'s online script at toy scale (maze/ppo/train_ppo_online.py): text path, device rollouts -> host data, and the device-resident loop
    mz = ["ppo", "--env", "maze", "--maze-max-steps", "3", "--n-rollouts", "6", "--rollout-bsize", "4", "--ppo-data-bsize", "4", "--train-bsize", "4",
          "--max-steps", "2", "--max-input-length", "160", "--max-output-length", "6"]
    H.main(mz)
    H.main(mz + ["--device-rollouts", "1", "--resident", "0"])
    H.main(mz + ["--device-rollouts", "1", "--n-rounds", "2", "--trim-batches", "1", "--bf16-activations", "1", "--policy-top-k", "40"])      # (fused top-k epilogue)
    # the partially observed twin (maze/ppo/partially_observed_ppo_online.py: item windows, joined-state chains): host path and the device-resident loop
    po = [x if x != "160" else "320" for x in mz] + ["--maze-last-k", "5", "--maze-describe-function", "describe_observation_only_walls"]
    H.main(po)
    H.main(po + ["--device-rollouts", "1", "--trim-batches", "1"])
    sf = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "stockfish")
    if os.path.exists(sf):
        H.main(["ppo", "--env", "chess", "--chess-engine", sf, "--chess-use-nnue", "false", "--chess-movetime-ms", "10", "--chess-max-moves", "2", "--n-rollouts", "2",
                "--rollout-bsize", "2", "--ppo-data-bsize", "2", "--train-bsize", "2", "--max-steps", "1", "--max-input-length", "160", "--max-output-length", "8"])
    H.main(["bc-eval", "--model", str(tmp_path / "ckpt" / "base"), "--policy-n-rollouts", "4", "--policy-bsize", "2", "--policy-max-input-length", "88",
            "--policy-max-output-length", "8"])
    H.main(["bc-eval", "--device-rollouts", "1", "--policy-n-rollouts", "6", "--policy-bsize", "4"])
    H.main(["bc-eval", "--device-rollouts", "1", "--policy-n-rollouts", "40", "--policy-bsize", "8", "--rollout-lanes", "2"])       # 5 batches: graph path, two lanes
    H.main(["maze-eval", "--max-steps", "3", "--generation-bsize", "8", "--max-input-length", "160", "--max-output-length", "6"])
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    tags = [next(iter(l)) for l in lines]
    assert tags.count("eval") >= 13 and tags.count("data_collection") >= 10 and "gen_data" in tags and "data_collection" in tags and "maze_eval" in tags and "train" in tags
    me = next(l["maze_eval"] for l in lines if "maze_eval" in l)
    assert me["n"] == 26 and 0.0 <= me["move_accuracy"] <= 100.0


def test_reference_style_code_through_compat_imports():
    """Code written against the reference's import paths (compat/) runs on the HIP env and reproduces the oracle env."""
    import os, sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "compat"))
    sys.path.insert(0, root)
    try:
        from LLM_RL.environment import Text, TextPolicy, text_env_eval
        from llm_rl_scripts.wordle.env.env import ReformatWordleEnvironment, WordleEnvironment
        from llm_rl_scripts.wordle.env.game import Vocabulary
        from lmrl_gym_amd.envs.wordle import Vocabulary as V2
        from oracle.wordle import OracleWordleEnv
        vocab = V2.builtin("wordle_official_400.txt")
        assert Vocabulary is V2
        env = ReformatWordleEnvironment(WordleEnvironment(vocab, require_words_in_vocab=True, bad_word_reward=-10.0))

        class Scripted(TextPolicy):
            def __init__(self):
                self.rng = np.random.RandomState(0)

            def act(self, text_history):
                return text_history + (Text(" ".join(vocab.all_vocab[self.rng.randint(len(vocab.all_vocab))]) + "\n", True),)

        inter, summary = text_env_eval(env, Scripted(), n_rollouts=5, seed_generator=iter([11, 12, 13, 14, 15]), bsize=1, verbose=False)
        assert len(inter) == 5
        for ep, seed in zip(inter, [11, 12, 13, 14, 15]):
            o = OracleWordleEnv(vocab.all_vocab, True, -10.0)
            hist = o.reset(seed)
            for tr in ep:
                hist, r, d = o.step(hist + ((tr.post_action_history[-1].text, True),))
                assert tr.post_transition_history[-1].text == hist[-1][0] and float(tr.reward) == float(r) and tr.done == d
    finally:
        sys.path.remove(root)
        for k in [k for k in sys.modules if k == "LLM_RL" or k.startswith("LLM_RL.") or k == "llm_rl_scripts" or k.startswith("llm_rl_scripts.")]:
            del sys.modules[k]
