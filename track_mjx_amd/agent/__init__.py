"""track_mjx.agent.mlp_ppo mirror: PPO learner for the tracking task (PyTorch-ROCm host + HIP kernels)."""
