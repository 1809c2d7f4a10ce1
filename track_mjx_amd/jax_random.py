"""Same-seed random numbers: the threefry2x32 counter-based generator and the `jax.random` functions the reference's
reset / learner call, restated in numpy so that `reset(key)` draws the clip, start frame and reset noise the reference
draws from the same key.

Third-party semantics (jax is pinned to 0.6.2 by the reference's pyproject.toml and is not installed here): restated
from the published algorithm — Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (Threefry-2x32, 20 rounds)
and jax/_src/prng.py, jax/_src/random.py.  Pinned (tests/test_jax_random.py) by the Random123 known-answer vectors that
jax's own test-suite checks (tests/random_test.py: testThreefry2x32) and by the values jax's documentation prints:
legacy layout — bits(key 1701), split(key 0), normal(key 0, (3,)); partitionable layout — split(key 0).  randint, the
partitionable `bits` / `uniform` and `permutation` are integer arithmetic on those words plus one exact bits->float
step, composed as in jax/_src/random.py from memory: no published value pins them here ("parity unpinned" for those
compositions until they can be run against jax).

`partitionable` selects jax's `jax_threefry_partitionable` layout of counters (default True since jax 0.5.0, hence
for the pinned 0.6.2); False is the legacy layout.

Reference call sites: task/multi_clip_tracking.py:85-89 (split 3, randint), task/single_clip_tracking.py:134,153-161
(split 3, uniform with the SAME key for qpos and qvel noise), agent/mlp_ppo/ppo.py:443-451,727-737 (PRNGKey, split,
fold_in), ppo.py:304-307 (permutation).
"""
from __future__ import annotations

import math

import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))
PARTITIONABLE_DEFAULT = True


def _rotl(x, d):
    return (x << _U32(d)) | (x >> _U32(32 - d))


def threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds. key: (..., 2) uint32 (one key, or one per element); x0, x1: uint32 arrays. Returns (y0, y1)."""
    key = np.asarray(key, dtype=_U32)
    k0, k1 = key[..., 0], key[..., 1]            # scalars, or arrays broadcastable against x0 / x1 (one key per element)
    ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
    shape = np.broadcast_shapes(np.shape(x0), np.shape(x1), np.shape(k0))
    x0 = np.array(np.broadcast_to(np.asarray(x0, dtype=_U32), shape), dtype=_U32, copy=True)
    x1 = np.array(np.broadcast_to(np.asarray(x1, dtype=_U32), shape), dtype=_U32, copy=True)
    with np.errstate(over="ignore"):
        x0 += ks[0]
        x1 += ks[1]
        for r in range(5):
            for d in _ROT[r & 1]:
                x0 += x1
                x1 = _rotl(x1, d)
                x1 ^= x0
            x0 += ks[(r + 1) % 3]
            x1 += ks[(r + 2) % 3] + _U32(r + 1)
    return x0, x1


def PRNGKey(seed: int) -> np.ndarray:
    """threefry_seed: [high 32 bits, low 32 bits] of the (64-bit) seed."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([seed >> 32, seed & 0xFFFFFFFF], dtype=_U32)


def _legacy_2x32(key, counts):
    """threefry_2x32 on a flat count array (legacy layout): the array is cut in two halves that form the two words."""
    n = counts.shape[0]
    odd = n % 2
    if odd:
        counts = np.concatenate([counts, np.zeros(1, _U32)])
    h = counts.shape[0] // 2
    y0, y1 = threefry2x32(key, counts[:h], counts[h:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def split(key, num: int = 2, partitionable: bool | None = None) -> np.ndarray:
    """jax.random.split -> (num, 2) uint32."""
    p = PARTITIONABLE_DEFAULT if partitionable is None else partitionable
    key = np.asarray(key, dtype=_U32)
    if p:
        y0, y1 = threefry2x32(key, np.zeros(num, _U32), np.arange(num, dtype=_U32))
        return np.stack([y0, y1], axis=-1)
    return _legacy_2x32(key, np.arange(2 * num, dtype=_U32)).reshape(num, 2)


def fold_in(key, data: int) -> np.ndarray:
    d = PRNGKey(data)
    y0, y1 = threefry2x32(np.asarray(key, dtype=_U32), d[:1], d[1:])
    return np.array([y0[0], y1[0]], dtype=_U32)


def bits(key, shape=(), partitionable: bool | None = None) -> np.ndarray:
    """jax.random.bits(key, shape, uint32)."""
    p = PARTITIONABLE_DEFAULT if partitionable is None else partitionable
    key = np.asarray(key, dtype=_U32)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    if p:   # one counter per element: the 64-bit row-major index as (hi, lo); the two output words are xor-ed
        idx = np.arange(n, dtype=np.uint64)
        y0, y1 = threefry2x32(key, (idx >> np.uint64(32)).astype(_U32), (idx & np.uint64(0xFFFFFFFF)).astype(_U32))
        return (y0 ^ y1).reshape(shape)
    return _legacy_2x32(key, np.arange(n, dtype=_U32)).reshape(shape)


def uniform(key, shape=(), minval: float = 0.0, maxval: float = 1.0, partitionable: bool | None = None) -> np.ndarray:
    """jax.random.uniform, float32: mantissa bits | 1.0, minus 1, scaled; max(minval, .)."""
    b = bits(key, shape, partitionable)
    f = ((b >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)


def randint(key, shape, minval: int, maxval: int, partitionable: bool | None = None) -> np.ndarray:
    """jax.random.randint, int32: two draws combined so that the modulo bias is 2^-64-small."""
    k1, k2 = split(key, 2, partitionable)
    hi_bits, lo_bits = bits(k1, shape, partitionable), bits(k2, shape, partitionable)
    span = _U32((int(maxval) - int(minval)) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        mult = _U32(65536) % span
        mult = (mult * mult) % span
        off = ((hi_bits % span) * mult + (lo_bits % span)) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


_ERFINV_LO = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164,
              0.246640727, 1.50140941)
_ERFINV_HI = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047,
              1.00167406, 2.83297682)


def _erfinv_f32(x: np.ndarray) -> np.ndarray:
    """Giles' single-precision erfinv (the polynomial XLA's ErfInv uses for f32)."""
    x = x.astype(np.float32)
    w = -np.log((np.float32(1) - x) * (np.float32(1) + x)).astype(np.float32)
    small = w < np.float32(5)
    wl = np.where(small, w - np.float32(2.5), np.sqrt(np.maximum(w, 0)).astype(np.float32) - np.float32(3)).astype(np.float32)
    p = np.where(small, np.float32(_ERFINV_LO[0]), np.float32(_ERFINV_HI[0])).astype(np.float32)
    for a, b in zip(_ERFINV_LO[1:], _ERFINV_HI[1:]):
        p = (np.where(small, np.float32(a), np.float32(b)) + p * wl).astype(np.float32)
    return (p * x).astype(np.float32)


def normal(key, shape=(), partitionable: bool | None = None) -> np.ndarray:
    """jax.random.normal, float32: sqrt(2) erfinv(u), u uniform on (-1, 1).  Transcendentals: equal to the XLA result to
    rounding, not bit for bit."""
    lo = np.nextafter(np.float32(-1), np.float32(0))
    u = uniform(key, shape, lo, 1.0, partitionable)
    return (np.float32(math.sqrt(2.0)) * _erfinv_f32(u)).astype(np.float32)


def permutation(key, n: int, partitionable: bool | None = None) -> np.ndarray:
    """jax.random.permutation(key, n): rounds of stable sorts by fresh 32-bit keys."""
    x = np.arange(n, dtype=np.int32)
    rounds = int(math.ceil(3 * math.log(max(1, n)) / math.log(2 ** 32 - 1)))
    key = np.asarray(key, dtype=_U32)
    for _ in range(rounds):
        key, sub = split(key, 2, partitionable)
        x = x[np.argsort(bits(sub, (n,), partitionable), kind="stable")]
    return x


def reset_draws(key, n_clips: int, nq: int, nv: int, noise_scale: float, partitionable: bool | None = None):
    """What MultiClipTracking.reset(rng) draws from one env's key (multi_clip_tracking.py:85-89, single_clip_tracking.py:134-161):
    returns (clip_idx, start_frame, qpos_noise[nq], qvel_noise[nv]).  Both noises use rng1 — the reference's own re-use."""
    _, start_rng, clip_rng = split(key, 3, partitionable)
    start_frame = int(randint(start_rng, (), 0, 44, partitionable))
    clip_idx = int(randint(clip_rng, (), 0, n_clips, partitionable))
    _, rng1, _ = split(key, 3, partitionable)
    qn = uniform(rng1, (nq,), -noise_scale, noise_scale, partitionable)
    vn = uniform(rng1, (nv,), -noise_scale, noise_scale, partitionable)
    return clip_idx, start_frame, qn, vn


def reset_draws_batch(keys, n_clips: int, nq: int, nv: int, noise_scale: float):
    """reset_draws for one key per env (keys: (n, 2) uint32, e.g. split(key_env, num_envs) as brax's vmapped reset receives them),
    vectorised over the envs; partitionable layout only.  Returns clip_idx[n], start_frame[n] (int32), qpos_noise[nq, n],
    qvel_noise[nv, n] (float32, env index contiguous as the C-ABI takes them)."""
    keys = np.asarray(keys, dtype=_U32).reshape(-1, 2)
    n = keys.shape[0]

    def split3(k):      # (n, 2) -> three (n, 2)
        y0, y1 = threefry2x32(k[:, None, :], np.zeros((n, 3), _U32), np.broadcast_to(np.arange(3, dtype=_U32), (n, 3)))
        return [np.stack([y0[:, j], y1[:, j]], axis=-1) for j in range(3)]

    def bits_n(k, m):   # (n, 2) -> (n, m) words
        y0, y1 = threefry2x32(k[:, None, :], np.zeros((n, m), _U32), np.broadcast_to(np.arange(m, dtype=_U32), (n, m)))
        return y0 ^ y1

    def randint_n(k, span):
        y0, y1 = threefry2x32(k[:, None, :], np.zeros((n, 2), _U32), np.broadcast_to(np.arange(2, dtype=_U32), (n, 2)))
        k1, k2 = np.stack([y0[:, 0], y1[:, 0]], -1), np.stack([y0[:, 1], y1[:, 1]], -1)
        hb, lb = bits_n(k1, 1)[:, 0], bits_n(k2, 1)[:, 0]
        sp = _U32(span)
        mult = _U32(65536) % sp
        mult = (mult * mult) % sp
        with np.errstate(over="ignore"):
            return (((hb % sp) * mult + (lb % sp)) % sp).astype(np.int32)

    _, start_rng, clip_rng = split3(keys)
    start_frame, clip_idx = randint_n(start_rng, 44), randint_n(clip_rng, n_clips)
    rng1 = split3(keys)[1]
    b = bits_n(rng1, max(nq, nv))
    f = ((b >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(-noise_scale), np.float32(noise_scale)
    u = np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)
    return clip_idx, start_frame, np.ascontiguousarray(u[:, :nq].T), np.ascontiguousarray(u[:, :nv].T)


class SgdKeys:
    """The learner's key plumbing for ONE device, from the seed down to the minibatch shuffles (track_mjx/agent/mlp_ppo/ppo.py):
         train          :443-451  key = PRNGKey(seed); global_key, local_key = split(key); local_key = fold_in(local_key, process_id);
                                   local_key, key_env, eval_key = split(local_key, 3)
         epoch loop     :729-730  epoch_key, local_key = split(local_key); epoch_keys = split(epoch_key, local_devices_to_use)
         training_step  :324      key_sgd, key_generate_unroll, new_key = split(key, 3)            (carry: new_key)
         sgd_step       :303-307  key, key_perm, key_grad = split(key, 3); x = permutation(key_perm, x) for every leaf
       One process per GPU here: rank r plays local device r of the reference's single process (process_id 0, local_devices = world size),
       so a run with the same seed shuffles every device's roll-out rows exactly as the reference's pmap replica r does."""

    def __init__(self, seed: int, process_id: int = 0, device_index: int = 0, local_devices: int = 1, partitionable: bool | None = None):
        self.part = partitionable
        _global_key, local_key = split(PRNGKey(seed), 2, self.part)
        local_key = fold_in(local_key, process_id)
        self.local_key, self.key_env, self.eval_key = split(local_key, 3, self.part)
        self.device_index, self.local_devices = int(device_index), int(local_devices)
        self.key = None          # the training_step carry of the current epoch
        self.key_sgd = None
        self.key_generate_unroll = None

    def start_epoch(self) -> None:
        epoch_key, self.local_key = split(self.local_key, 2, self.part)
        self.key = split(epoch_key, self.local_devices, self.part)[self.device_index]

    def start_training_step(self) -> None:
        if self.key is None:
            self.start_epoch()
        self.key_sgd, self.key_generate_unroll, self.key = split(self.key, 3, self.part)
        self._sgd_carry = self.key_sgd

    # ---- acting noise of the roll-outs (parity mode: two normal() draws per control step on the host)
    def start_unrolls(self) -> None:
        """training_step's scan over the unrolls starts from key_generate_unroll (ppo.py:343-348)."""
        self._unroll_carry = self.key_generate_unroll

    def start_unroll(self) -> None:
        """ppo.py:333-340: current_key, next_key = split(current_key); generate_unroll(env, state, policy, current_key, unroll_length)."""
        self._step_carry, self._unroll_carry = split(self._unroll_carry, 2, self.part)

    def act_noise(self, n_env: int, latents: int, action_size: int):
        """One actor_step of brax's generate_unroll (third-party, restated: `current_key, next_key = split(current_key)` per step; the policy
        gets current_key) through make_inference_fn.policy (ppo_networks.py:46-96: key_sample, key_network = split(key_sample)) and
        IntentionNetwork.__call__ (intention_network.py:104,78-81: _, encoder_rng = split(key); eps = normal(encoder_rng, logvar.shape))
        and NormalTanhDistribution.sample_no_postprocessing (normal(key_sample, loc.shape)).  Returns (eps [n, latents], noise [n, A])
        for ALL envs of this device, float32."""
        current, self._step_carry = split(self._step_carry, 2, self.part)
        key_sample, key_network = split(current, 2, self.part)
        _, encoder_rng = split(key_network, 2, self.part)
        return normal(encoder_rng, (n_env, latents), self.part), normal(key_sample, (n_env, action_size), self.part)

    def permutation(self, n: int) -> np.ndarray:
        """The next sgd_step's row permutation (call once per update, in order)."""
        self._sgd_carry, key_perm, _key_grad = split(self._sgd_carry, 3, self.part)
        return permutation(key_perm, n, self.part)
