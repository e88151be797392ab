#!/bin/bash
# Row H at GPT-2-SMALL size (12 layers, d = 768, V = 50257+pad) on one GPU: the task scripts' loops with the scripts' own defaults for
# sequence lengths / batch sizes (train_bsize 32, max_length 512, gamma / tau / cql / beta, bad_word_reward -10), bounded by --max-steps:
#   gen-data -> jsonl -> ILQL (train steps + device-resident evaluation rollouts) -> PPO round (device rollouts -> PPO data -> train steps)
#   -> BC eval (device rollouts).  Random-init weights (no checkpoints offline).  usage: tools/run_harness_small.sh [outfile]
OUT=${1:-gpurun_out/harness_gpt2_small.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
D=/tmp/harness_small; rm -rf $D; mkdir -p $D
{
echo "## gen-data"; time python scripts/harness.py gen-data --n-data 600 --out $D/train.jsonl 2>&1 | tail -2
echo "## ilql (GPT-2-small, train_bsize 32, max_length 512, 3 steps, 64 device rollouts)"
time python scripts/harness.py ilql --model random:small --train-data $D/train.jsonl --max-steps 3 --epochs 1 --log-every 1 --policy-n-rollouts 64 --policy-bsize 64 --device-rollouts 1 2>&1 | tail -8
echo "## ppo (GPT-2-small, 128 rollouts x bsize 64 on the device loop, 2 train steps)"
time python scripts/harness.py ppo --model random:small --n-rollouts 128 --rollout-bsize 64 --max-steps 2 --device-rollouts 1 2>&1 | tail -8
echo "## bc-eval (GPT-2-small, 64 rollouts on the device loop)"
time python scripts/harness.py bc-eval --model random:small --policy-n-rollouts 64 --policy-bsize 64 --device-rollouts 1 2>&1 | tail -4
} > $OUT 2>&1
tail -40 $OUT
