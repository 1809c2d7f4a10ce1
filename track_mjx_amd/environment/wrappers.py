"""wrappers.wrap mirror (reference: track_mjx/environment/wrappers.py:18-56).

In the reference `wrap` stacks brax's EpisodeWrapper and VmapWrapper and then the
(LSTM)AutoResetWrapperTracking.  Here batching is native and the episode / auto-reset logic runs
inside the K3 kernel (csrc/env_core.h: tm_step_prologue / tm_step_post), so `wrap` only switches
those semantics on for the env's handle and returns the env itself.  `action_repeat` is brax EpisodeWrapper's: the env's own step
runs that many times per `step` with the same action, rewards summed, the step counter advanced by `action_repeat`
(include/tmjx.h: tmjx_set_action_repeat).
"""
from __future__ import annotations

from .task import MultiClipTracking


def wrap(env: MultiClipTracking, episode_length: int = 1000, action_repeat: int = 1, randomization_fn=None,
         use_lstm: bool = True, hidden_state_dim: int = 128, hidden_layer_num: int = 2) -> MultiClipTracking:
    """Episode (steps/truncation) + auto-reset semantics (wrappers.py:104-144, brax EpisodeWrapper).

    `use_lstm`, `hidden_state_dim`, `hidden_layer_num` are accepted for signature parity: the LSTM
    auto-reset wrapper differs from the plain one only by an unused `info["hidden_state"]`
    (wrappers.py:59-144 vs 278-310), which is not materialised."""
    if randomization_fn is not None:
        # brax's DomainRandomizationVmapWrapper steps every env with its own copy of the mjx.Model; the physics kernel reads ONE model
        # from constant memory (csrc/dmodel.h), and no reference config or call site passes a randomization_fn (ppo.py:466-474)
        raise NotImplementedError("domain randomisation (a per-env model) is not supported: the device model is one constant per handle")
    env.configure_wrappers(int(episode_length), auto_reset=True, action_repeat=int(action_repeat))
    return env
