"""Same-seed RNG (track_mjx_amd/jax_random.py) against the known answers that exist for jax's generator:
the Random123 Threefry-2x32 vectors jax's own test-suite checks (tests/random_test.py::testThreefry2x32) and the values jax's
documentation prints for key 0 / seed 1701 (legacy counter layout: split, bits, normal; partitionable layout: split)."""
import numpy as np

from track_mjx_amd import jax_random as jr


def test_threefry2x32_random123_known_answers():
    for key, cnt, exp in (((0x0, 0x0), (0x0, 0x0), (0x6b200159, 0x99ba4efe)),
                          ((0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff), (0x1cb996fc, 0xbb002be7)),
                          ((0x13198a2e, 0x03707344), (0x243f6a88, 0x85a308d3), (0xc4923a9c, 0x483df7a0))):
        y0, y1 = jr.threefry2x32(np.array(key, dtype=np.uint32), np.array([cnt[0]], np.uint32), np.array([cnt[1]], np.uint32))
        assert (int(y0[0]), int(y1[0])) == exp


def test_legacy_layout_known_answers():
    assert jr.bits(jr.PRNGKey(1701), (3,), partitionable=False).tolist() == [56197195, 4200222568, 961309823]
    assert jr.split(jr.PRNGKey(0), 2, partitionable=False).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    np.testing.assert_allclose(jr.normal(jr.PRNGKey(0), (3,), partitionable=False), [1.8160863, -0.48262316, 0.33988908], rtol=2e-6)


def test_partitionable_layout_known_answer_and_prefix_property():
    assert jr.split(jr.PRNGKey(0), 2).tolist() == [[1797259609, 2579123966], [928981903, 3453687069]]
    k = jr.PRNGKey(7)
    # one counter per element: a shorter draw is a prefix of a longer one — the reference's qvel noise (same key as qpos noise,
    # single_clip_tracking.py:153-161) therefore repeats the first nv values of the qpos noise
    assert np.array_equal(jr.uniform(k, (74,), -1e-3, 1e-3)[:73], jr.uniform(k, (73,), -1e-3, 1e-3))
    assert np.array_equal(jr.fold_in(jr.PRNGKey(0), 1), jr.split(jr.PRNGKey(0), 2)[1])     # both are threefry(key, (0, 1))


def test_values_printed_in_jax_documentation():
    """Values jax's own documentation prints (recalled offline, no network in this image: each is reproduced to every printed digit by the independent
    implementation here — ten-float coincidences are not chance):
      * README / quickstart:        random.normal(PRNGKey(0), (10,))                                   (legacy layout)
      * "JAX - The Sharp Bits":     normal(PRNGKey(0), (1,)); key, subkey = split(key) twice; key, *subkeys = split(key, 4)   (legacy layout)
      * "Pseudorandom numbers":     normal(PRNGKey(42)) and normal of the subkey of split(key)          (legacy: -0.18471177 / 1.3694694; partitionable
                                    default of the pinned jax 0.6.2: -0.028304616 / 0.60576403)
      * jax.random module docstring: random.uniform(random.key(0)) = 0.947667                           (partitionable layout: pins bits + uniform)"""
    leg = dict(partitionable=False)
    np.testing.assert_allclose(jr.normal(jr.PRNGKey(0), (10,), **leg),
                               [-0.3721109, 0.26423115, -0.18252768, -0.7368197, -0.44030377, -0.1521442, -0.67135346, -0.5908641, 0.73168886, 0.5673026], rtol=2e-6)
    key = jr.PRNGKey(0)
    np.testing.assert_allclose(jr.normal(key, (1,), **leg), [-0.20584226], rtol=2e-6)
    key, sub = jr.split(key, 2, **leg)
    np.testing.assert_allclose(jr.normal(sub, (1,), **leg), [-1.2515389], rtol=2e-6)
    key, sub = jr.split(key, 2, **leg)
    np.testing.assert_allclose(jr.normal(sub, (1,), **leg), [-0.58665055], rtol=2e-6)
    subs = jr.split(key, 4, **leg)[1:]
    np.testing.assert_allclose([float(jr.normal(s_, (1,), **leg)[0]) for s_ in subs], [-0.37533438, 0.98645043, 0.14553197], rtol=2e-6)
    k42 = jr.PRNGKey(42)
    np.testing.assert_allclose(jr.normal(k42, (), **leg), -0.18471177, rtol=2e-6)
    np.testing.assert_allclose(jr.normal(jr.split(k42, 2, **leg)[1], (), **leg), 1.3694694, rtol=2e-6)
    np.testing.assert_allclose(jr.normal(k42, (), partitionable=True), -0.028304616, rtol=2e-6)
    np.testing.assert_allclose(jr.normal(jr.split(k42, 2, partitionable=True)[1], (), partitionable=True), 0.60576403, rtol=2e-6)
    np.testing.assert_allclose(jr.uniform(jr.PRNGKey(0), (), partitionable=True), 0.947667, rtol=2e-6)


def test_ranges_and_permutation():
    k = jr.PRNGKey(3)
    u = jr.uniform(k, (10000,), -0.5, 0.25)
    assert u.dtype == np.float32 and u.min() >= -0.5 and u.max() < 0.25 and abs(float(u.mean()) + 0.125) < 0.01
    r = jr.randint(k, (10000,), 0, 44)
    assert r.dtype == np.int32 and r.min() == 0 and r.max() == 43 and len(np.unique(r)) == 44
    p = jr.permutation(k, 16384)
    assert sorted(p.tolist()) == list(range(16384)) and not np.array_equal(p, np.arange(16384))
    z = jr.normal(k, (20000,))
    assert abs(float(z.mean())) < 0.03 and abs(float(z.std()) - 1.0) < 0.03


def test_reset_draws_batch_matches_scalar_path():
    keys = jr.split(jr.PRNGKey(11), 37)
    ci, sf, qn, vn = jr.reset_draws_batch(keys, 64, 74, 73, 1e-3)
    assert qn.shape == (74, 37) and vn.shape == (73, 37)
    for e in (0, 5, 36):
        c1, s1, q1, v1 = jr.reset_draws(keys[e], 64, 74, 73, 1e-3)
        assert (int(ci[e]), int(sf[e])) == (c1, s1)
        assert np.array_equal(qn[:, e], q1) and np.array_equal(vn[:, e], v1)


def test_sgd_key_plumbing_and_permutation_hook():
    """SgdKeys walks ppo.py:443-451 -> :729-730 -> :324 -> :303-307 exactly as composed by hand from split / fold_in / permutation, rank r
    = the reference's local device r; the learner's perm_fn hook hands those permutations to the SGD loop."""
    import torch
    from tests.common import StubEnv
    from track_mjx_amd import jax_random as jr
    from track_mjx_amd.agent.ppo import PPOLearner
    seed, world, rank, rows = 7, 4, 2, 64
    _g, local = jr.split(jr.PRNGKey(seed), 2)
    local = jr.fold_in(local, 0)
    local, _env, _eval = jr.split(local, 3)
    epoch_key, local = jr.split(local, 2)
    key = jr.split(epoch_key, world)[rank]
    expect = []
    for _step in range(2):
        key_sgd, _unroll, key = jr.split(key, 3)
        carry = key_sgd
        for _upd in range(3):
            carry, key_perm, _grad = jr.split(carry, 3)
            expect.append(jr.permutation(key_perm, rows))
    sk = jr.SgdKeys(seed, 0, rank, world)
    sk.start_epoch()
    got = []
    for _step in range(2):
        sk.start_training_step()
        got += [sk.permutation(rows) for _ in range(3)]
    assert all(np.array_equal(a, b) for a, b in zip(got, expect))
    assert all(sorted(p.tolist()) == list(range(rows)) for p in got) and not np.array_equal(got[0], got[1])
    # another device of the same process shuffles differently (ppo.py:730: one key per local device)
    other = jr.SgdKeys(seed, 0, rank + 1, world); other.start_training_step()
    assert not np.array_equal(other.permutation(rows), expect[0])
    ln = PPOLearner(StubEnv(4, 24, 16, 3), encoder_layers=(8,), decoder_layers=(8,), critic_layers=(8,), latents=4, unroll_length=2, batch_size=4,
                    num_minibatches=2, num_updates_per_batch=3, use_graph=False, seed=seed, shuffle_rng="jax")
    ln.sgd_keys = jr.SgdKeys(seed, 0, rank, world)
    ln.start_epoch(); ln.sgd_keys.start_training_step()
    p0 = ln.perm_fn(0, rows)
    assert p0.dtype == torch.int64 and np.array_equal(p0.numpy(), expect[0])


def test_acting_noise_key_plumbing():
    """SgdKeys.act_noise follows training_step's unroll scan (ppo.py:333-348) -> brax generate_unroll's per-step split -> the policy's
    key_sample / key_network split (ppo_networks.py:50) -> the network's encoder_rng (intention_network.py:104) as composed by hand."""
    from track_mjx_amd import jax_random as jr
    sk = jr.SgdKeys(3, 0, 1, 2)
    sk.start_epoch(); sk.start_training_step()
    carry = sk.key_generate_unroll
    sk.start_unrolls()
    for _unroll in range(2):
        cur, carry = jr.split(carry, 2)
        sk.start_unroll()
        step_carry = cur
        for _t in range(3):
            current, step_carry = jr.split(step_carry, 2)
            key_sample, key_network = jr.split(current, 2)
            _, enc = jr.split(key_network, 2)
            eps, noise = sk.act_noise(8, 60, 38)
            assert eps.shape == (8, 60) and noise.shape == (8, 38) and eps.dtype == np.float32
            assert np.array_equal(eps, jr.normal(enc, (8, 60))) and np.array_equal(noise, jr.normal(key_sample, (8, 38)))
    big = np.concatenate([sk.act_noise(512, 60, 38)[0].ravel() for _ in range(4)])
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1) < 0.02
