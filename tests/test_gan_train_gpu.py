"""GPU parity tests for the GAN's discriminator-side training (misc_py/gan-infilling-100.py:1048-1088 tower,
:1390-1440 train op): emdenoise.gan_trainer.DiscriminatorTrainer against the oracle's PyTorch-CPU float64 autograd
(oracle/gan_graph.py discriminator_tower, adam_step, clip_by_global_norm).  As for graph D' (tests/test_train_gpu.py)
the gradient of a leaky-relu network is discontinuous in the forward values, so full-network gradients carry a
mask-flip tolerance; the optimizer arithmetic is checked tightly on the trainer's own gradient."""
import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def dev():
    return torch.device("cuda", 0)


def flat(d, names):
    return np.concatenate([np.asarray(d[n], np.float64).ravel() for n in names])


S = 512
OFFSETS = ((S // 3, S // 5), (S // 7, S // 2), (S // 4 + 3, S // 9))


def images(T, seed):
    return (2.0 * synthetic_lq(T, S, S, seed=seed) - 1.0).astype(np.float32)


@pytest.mark.parametrize("label,adapt", [(0.95, 1.0), (0.05, 1.7)])
def test_discriminator_tower_gradients(label, adapt):
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT
    from oracle import gan_graph as GG

    w = GN.discriminator_synthetic_weights()
    img = images(1, 11)
    ref = GG.discriminator_tower(img, label, w, OFFSETS, adapt=adapt)
    tr = GT.DiscriminatorTrainer(w, dev())
    tr.zero_grad()
    res = tr.tower(torch.from_numpy(img).to(dev()), label, OFFSETS, adapt=adapt).cpu().numpy()
    g = tr.gradients()
    names = list(g)
    l2_term = 5e-5 * sum(0.5 * float((np.asarray(w[n], np.float64) ** 2).sum()) for n in names)
    # the tower adds the data term only; the oracle's gradient also holds adapt * 5e-5 * v
    refg = {n: ref["grads"][n] - adapt * 5e-5 * np.asarray(w[n], np.float64) for n in names}
    live = [n for n in names if np.abs(refg[n]).max() > 1e-10]
    a, b = flat(g, live), flat(refg, live)
    print(f"D tower label {label}: out {res[0]:.6f} vs {ref['output'][0]:.6f}; loss {res[1]:.6f} vs {ref['loss'] - l2_term:.6f}; "
          f"grads rel-l2 {rel_l2(a, b):.2e} cos {cosine(a, b):.5f}")
    assert abs(res[0] - ref["output"][0]) < 1e-4 and abs(res[1] - (ref["loss"] - l2_term)) < 1e-3
    assert rel_l2(a, b) < 6e-2 and cosine(a, b) > 0.998
    st = tr.state_dict()
    for n, v in ref["moving"].items():
        assert np.allclose(st[n], v, rtol=1e-4, atol=1e-6), n


def test_discriminator_step_matches_oracle_optimizer():
    """Two towers (a 'real' and a 'generated' image with their labels), averaged, + l2, clipped to norm 15, Adam."""
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT
    from oracle import gan_graph as GG

    w = GN.discriminator_synthetic_weights()
    imgs = images(2, 21)
    labels, adapts = [0.9, 0.1], [1.0, 1.4]
    tr = GT.DiscriminatorTrainer(w, dev(), learning_rate=1e-4)
    names = list(tr.trainable)
    params = {n: np.asarray(w[n], np.float64) for n in names}
    m = {n: np.zeros_like(v) for n, v in params.items()}
    v = {n: np.zeros_like(p) for n, p in params.items()}
    for step in (1, 2):
        cur = tr.state_dict()
        res = tr.step(torch.from_numpy(imgs).to(dev()), labels, [OFFSETS, OFFSETS], adapts=adapts).cpu().numpy()
        towers = [GG.discriminator_tower(imgs[k:k + 1], labels[k], cur, OFFSETS, adapt=adapts[k]) for k in range(2)]
        gref = {n: 0.5 * (towers[0]["grads"][n] + towers[1]["grads"][n]) for n in names}
        gclip, gn = GG.clip_by_global_norm(gref, 15.0)
        g = tr.gradients()                 # the trainer's summed gradient (data + l2), before averaging
        a, b = flat(g, names) * 0.5, flat(gref, names)
        print(f"step {step}: outs {res[:, 0]} vs {[float(t['output'][0]) for t in towers]}; grads rel-l2 {rel_l2(a, b):.2e} "
              f"cos {cosine(a, b):.5f}; |g| {gn:.3f}")
        assert rel_l2(a, b) < 6e-2 and cosine(a, b) > 0.998
        # optimizer arithmetic on the trainer's OWN averaged gradient
        own = {n: 0.5 * np.asarray(g[n], np.float64) for n in names}
        own, _ = GG.clip_by_global_norm(own, 15.0)
        newp, m, v = GG.adam_step({n: np.asarray(cur[n], np.float64) for n in names}, own, m, v, step, 1e-4)
        st = tr.state_dict()
        upd = rel_l2(flat(st, names) - flat(cur, names), flat(newp, names) - flat(cur, names))
        worst = sorted(((float(np.abs((st[n] - cur[n]) - (newp[n] - cur[n])).max()), n) for n in names), reverse=True)[:4]
        assert upd < 2e-3, (step, upd, worst)


def test_generator_tower_gradients():
    """_generator_tower_fn (:982-1046): adversarial + feature-matching loss through the discriminator, the crops and
    the generator, against the oracle's autograd (float64).  Mask-flip tolerance as above (leaky_relu kinks)."""
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT
    from oracle import gan_graph as GG

    wg, wd = GN.synthetic_weights(), GN.discriminator_synthetic_weights()
    hq = images(1, 31)
    lq = GN.gen_lq(hq[..., 0])[..., None]
    ref = GG.generator_tower(lq, hq, wg, wd, OFFSETS)
    D = GT.DiscriminatorTrainer(wd, dev())
    tr = GT.GeneratorTrainer(wg, D, dev())
    tr.zero_grad()
    out, res, stat = tr.tower(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), OFFSETS)
    res, stat = res.cpu().numpy(), float(stat.cpu().numpy()[0])
    g = tr.gradients()
    names = [n for n in g if np.abs(ref["grads"][n]).max() > 1e-9]
    a, b = flat(g, names), flat(ref["grads"], names)
    worst = sorted(((rel_l2(g[n], ref["grads"][n]), n) for n in names), reverse=True)[:3]
    print(f"G tower: out {rel_l2(out.cpu().numpy(), ref['output']):.2e}; D(fake) {res[0]:.6f} vs {ref['d_fake'][0]:.6f}; "
          f"loss {res[1] + stat:.4f} vs {ref['loss']:.4f}; stat {stat / 12:.5f} vs {ref['stat_loss']:.5f}; "
          f"grads rel-l2 {rel_l2(a, b):.2e} cos {cosine(a, b):.5f}; worst {worst}")
    assert rel_l2(out.cpu().numpy(), ref["output"]) < 1e-3    # north-star bar for images; measured 3e-4 (unfused training forward)
    assert abs(res[0] - ref["d_fake"][0]) < 1e-4 and abs(res[1] + stat - ref["loss"]) < 2e-3 * ref["loss"]
    assert rel_l2(a, b) < 8e-2 and cosine(a, b) > 0.997


def test_generator_step_optimizer_arithmetic():
    """One generator step: the tower's gradient, clipped to global norm 50, through Adam(beta1 0.5), checked on the
    trainer's own gradient; the discriminator's parameters must not move."""
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT
    from oracle import gan_graph as GG

    wg, wd = GN.synthetic_weights(), GN.discriminator_synthetic_weights()
    hq = images(1, 41)
    lq = GN.gen_lq(hq[..., 0])[..., None]
    D = GT.DiscriminatorTrainer(wd, dev())
    tr = GT.GeneratorTrainer(wg, D, dev(), learning_rate=2e-4)
    names = list(tr.trainable)
    cur, dcur = tr.state_dict(), D.state_dict()
    res = tr.step(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), [OFFSETS]).cpu().numpy()
    g = {n: np.asarray(v, np.float64) for n, v in tr.gradients().items()}
    own, gn = GG.clip_by_global_norm(g, 50.0)
    zeros = {n: np.zeros_like(v) for n, v in g.items()}
    newp, _, _ = GG.adam_step({n: np.asarray(cur[n], np.float64) for n in names}, own, zeros, dict(zeros), 1, 2e-4)
    st = tr.state_dict()
    upd = rel_l2(flat(st, names) - flat(cur, names), flat(newp, names) - flat(cur, names))
    print(f"G step: D(fake) {res[0, 0]:.5f}, -log D {res[0, 1]:.4f}, 12*stat {res[0, 2]:.3f}; |g| {gn:.1f} (clip 50); update rel-l2 {upd:.2e}")
    assert gn > 50.0, "this configuration is meant to exercise the clipping"
    assert upd < 2e-3
    dst = D.state_dict()
    assert all(np.array_equal(dst[n], dcur[n]) for n in dcur)
