import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
print(json.dumps(d.get("sampling_warpers")))
print(json.dumps(d.get("maze_rollout", {}).get("by_batch")))
pi = d.get("ppo_iteration", {}).get("bf16", {})
print({k: (v.get("value") if isinstance(v, dict) else v) for k, v in pi.items() if k in ("value", "host_path", "trimmed_batches")})
print(d.get("larger_batches"))
