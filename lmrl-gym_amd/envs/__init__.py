"""Device-backed text environments (Wordle, Maze) behind the LLM_RL.environment protocol."""
