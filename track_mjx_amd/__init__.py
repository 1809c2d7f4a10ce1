"""track_mjx_amd — MI355X-native hot path of talmolab/track-mjx (rodent tracking rollout + PPO)."""
__version__ = "0.1.0"
