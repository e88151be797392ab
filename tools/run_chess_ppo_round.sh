#!/bin/bash
# configs[3] on ONE GPU: one online-PPO round against the chess env at the config's sizes — 4096 lock-step boards (device half-step
# kernels), GPT-2-medium policy (random init, bf16_activations), the reference-built Stockfish pool as the opponent (UCI_LimitStrength +
# Elo floor, 5 ms per move), PPO data + train steps at train_bsize 32.  usage: tools/run_chess_ppo_round.sh [outfile]
OUT=${1:-gpurun_out/chess_ppo_round.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
{
echo "## ppo --env chess, 4096 boards, GPT-2-medium, engine pool (oracle/_ref/stockfish), max 2 agent moves per game, 2 train steps"
time python scripts/harness.py ppo --env chess --model random:medium --bf16-activations true --chess-engine oracle/_ref/stockfish --chess-use-nnue false \
  --chess-movetime-ms 5 --chess-max-moves 2 --n-rollouts 4096 --rollout-bsize 4096 --max-input-length 256 --max-output-length 8 --ppo-data-bsize 64 \
  --max-steps 2 2>&1 | tail -6
} > $OUT 2>&1
tail -12 $OUT
