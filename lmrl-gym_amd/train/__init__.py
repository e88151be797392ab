"""fp32 train path: GPT-2 forward/backward, value heads, AdamW — every op is a HIP kernel behind the C ABI."""
