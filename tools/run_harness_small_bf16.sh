#!/bin/bash
# tools/run_harness_small.sh with the scripts' bf16_activations flag on (the bf16-matmul train mode): ILQL and PPO loops at GPT-2-small size
OUT=${1:-gpurun_out/harness_gpt2_small_bf16.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
D=/tmp/harness_small_bf16; rm -rf $D; mkdir -p $D
{
python scripts/harness.py gen-data --n-data 600 --out $D/train.jsonl 2>&1 | tail -1
echo "## ilql --bf16-activations (GPT-2-small, train_bsize 32, max_length 512, 3 steps, 64 device rollouts)"
time python scripts/harness.py ilql --model random:small --train-data $D/train.jsonl --max-steps 3 --epochs 1 --log-every 1 --policy-n-rollouts 64 --policy-bsize 64 --device-rollouts 1 --bf16-activations 1 2>&1 | tail -6
echo "## ppo --bf16-activations (GPT-2-small, 128 rollouts x bsize 64 on the device loop, 2 train steps)"
time python scripts/harness.py ppo --model random:small --n-rollouts 128 --rollout-bsize 64 --max-steps 2 --device-rollouts 1 --bf16-activations 1 2>&1 | tail -6
} > $OUT 2>&1
tail -30 $OUT
