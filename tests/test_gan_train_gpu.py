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


def test_generator_moving_statistics_phase():
    """The first 250 000 iterations of a reference run (train_batch_norm_on, gan-infilling-100.py:1644): the generator's train op
    also runs tower 0's batch-norm update ops with batch_norm_on_ph = True (:866-871, :1384, :1708-1712) -- one forward pass on BATCH
    statistics that assigns the moving averages (decay 0.9997, Bessel-corrected variance) -- while the tower GRADIENTS are evaluated
    with batch_norm_on_ph False in both phases (:1668).  At S = 512: every updated moving statistic against the oracle's float64
    restatement (oracle/gan_graph.py generator_moving_update); then the tower of the NEXT iteration (moving statistics, now the
    refreshed ones) against the oracle's tower on the oracle's refreshed statistics -- the gradient parity of phase one."""
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT
    from oracle import gan_graph as GG

    wg, wd = dict(GN.synthetic_weights()), GN.discriminator_synthetic_weights()
    # the shipped moving statistics were calibrated ON batch statistics (moving ~ batch: an update of 3e-4 x nothing); move them
    # away first so that the update is a vector float32 can resolve
    for n in list(wg):
        if n.endswith("/moving_mean"):
            wg[n] = (wg[n] + 0.25).astype(np.float32)
        elif n.endswith("/moving_variance"):
            wg[n] = (wg[n] * 1.5).astype(np.float32)
    hq = images(1, 41)
    lq = GN.gen_lq(hq[..., 0])[..., None]
    out_ref, upd = GG.generator_moving_update(lq, wg, S)
    D = GT.DiscriminatorTrainer(wd, dev())
    tr = GT.GeneratorTrainer(wg, D, dev())
    before = {n: v.copy() for n, v in tr.state_dict().items() if "moving_" in n}
    tr.update_moving_statistics(torch.from_numpy(lq).to(dev()))
    torch.cuda.synchronize()
    st = tr.state_dict()
    assert set(upd) == set(before), "every moving statistic of the generator is refreshed, nothing else"
    worst = 0.0
    for n, want in upd.items():
        step_ref, step_got = want - np.asarray(wg[n], np.float64), st[n].astype(np.float64) - before[n]
        worst = max(worst, rel_l2(step_got, step_ref))      # the UPDATE (3e-4 of the distance to the batch statistic), not the value
        assert rel_l2(st[n], want) < 1e-6, n
    print(f"generator moving-statistics update at {S} px: worst relative error of an update vector {worst:.2e}")
    assert worst < 5e-3
    # phase one's gradients: moving statistics (the refreshed ones) -- the oracle's tower on ITS refreshed statistics
    wg2 = dict(wg)
    wg2.update({n: v.astype(np.float32) for n, v in upd.items()})
    ref = GG.generator_tower(lq, hq, wg2, wd, OFFSETS)
    tr.zero_grad()
    out, res, stat = tr.tower(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), OFFSETS)
    g = tr.gradients()
    names = [n for n in g if np.abs(ref["grads"][n]).max() > 1e-9]
    a, b = flat(g, names), flat(ref["grads"], names)
    print(f"G tower after the update: out {rel_l2(out.cpu().numpy(), ref['output']):.2e}, grads rel-l2 {rel_l2(a, b):.2e} cos {cosine(a, b):.5f}")
    assert rel_l2(out.cpu().numpy(), ref["output"]) < 1e-3
    assert rel_l2(a, b) < 8e-2 and cosine(a, b) > 0.997
    # and through gan_iteration: the flag refreshes the statistics, its absence leaves them alone
    G2 = GT.GeneratorTrainer(wg, GT.DiscriminatorTrainer(wd, dev()), dev())
    x, t = torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())
    m0 = G2.moving.clone()
    GT.gan_iteration(G2, G2.D, x, t, [OFFSETS], train="gen", batch_norm_on=False)
    assert torch.equal(G2.moving, m0)
    GT.gan_iteration(G2, G2.D, x, t, [OFFSETS], train="gen", batch_norm_on=True)
    torch.cuda.synchronize()
    assert not torch.equal(G2.moving, m0)


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


# ------------------------------------------------------------------------------------------------ per-kernel checks
def _t64(a):
    return torch.from_numpy(np.asarray(a, np.float64))


@pytest.mark.parametrize("B,H,W,Cc,stride", [(2, 12, 16, 64, 1), (1, 9, 7, 128, 1), (2, 12, 16, 64, 2), (1, 7, 10, 768, 2), (1, 2, 2, 64, 1)])
def test_dw3x3_reflect_backward(B, H, W, Cc, stride):
    from emdenoise import train_ops as TO
    from oracle import gan_graph as GG
    from tests.test_ops_gpu import out_act, rnd, to_act

    x = _t64(rnd((B, H, W, Cc), 1)).requires_grad_(True)
    w = _t64(rnd((3, 3, Cc, 1), 2, 0.4)).requires_grad_(True)
    y = GG.depthwise_valid_t(GG.reflect_pad_t(x, 1), w, stride)
    dy = rnd(tuple(y.shape), 3)
    gx, gw = torch.autograd.grad(y, (x, w), _t64(dy))
    xa, dya = to_act(x.detach().numpy().astype(np.float32), ld=Cc + 8, c0=4), to_act(dy)
    dw = torch.zeros((9, Cc), dtype=torch.float32, device=dev())
    TO.dw3x3_reflect_wgrad(xa, dya, dw, stride=stride)
    wdev = torch.from_numpy(w.detach().numpy().reshape(9, Cc).astype(np.float32)).to(dev())
    dx = TO.dw3x3_reflect_bwd_data(dya, wdev, out_act(B, H, W, Cc), stride=stride)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy().reshape(3, 3, Cc, 1), gw.numpy()) < 2e-5
    assert rel_l2(dx.torch().cpu().numpy(), gx.numpy()) < 2e-6


def test_first_and_last_layer_backward():
    from emdenoise import train_ops as TO
    from oracle import gan_graph as GG
    from tests.test_ops_gpu import out_act, rnd, to_act

    B, H, W = 1, 20, 24
    d32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    # 7x7 reflect depthwise on the 1-channel image: forward into channel 0 of a 4-channel tensor, and dW
    x = rnd((B, H, W, 1), 4)
    w = _t64(rnd((7, 7, 1, 1), 5, 0.2)).requires_grad_(True)
    y = GG.depthwise_valid_t(GG.reflect_pad_t(_t64(x), 3), w, 1)
    dd = rnd((B, H, W, 1), 6)
    (gw,) = torch.autograd.grad(y, w, _t64(dd))
    d4 = TO.dw7_c1_reflect(d32(x), d32(w.detach().numpy().reshape(49)), out_act(B, H, W, 4))
    dd4 = np.zeros((B, H, W, 4), np.float32)
    dd4[..., 0:1] = dd
    dw49 = torch.zeros(49, dtype=torch.float32, device=dev())
    TO.dw7_c1_reflect_wgrad(d32(x), to_act(dd4), dw49)
    torch.cuda.synchronize()
    got = d4.torch().cpu().numpy()
    assert rel_l2(got[..., 0:1], y.detach().numpy()) < 2e-6 and np.all(got[..., 1:] == 0)
    assert rel_l2(dw49.cpu().numpy().reshape(7, 7, 1, 1), gw.numpy()) < 2e-5
    # last conv (reflect pad + 3x3 -> 1 channel): dW and dx
    xin = _t64(rnd((B, H, W, 32), 7)).requires_grad_(True)
    wl = _t64(rnd((3, 3, 32, 1), 8, 0.2)).requires_grad_(True)
    yl = GG.depthwise_valid_t(GG.reflect_pad_t(xin, 1), wl, 1).sum(-1, keepdim=True)
    dyl = rnd((B, H, W, 1), 9)
    gx, gwl = torch.autograd.grad(yl, (xin, wl), _t64(dyl))
    dwl = torch.zeros((9, 32), dtype=torch.float32, device=dev())
    TO.conv3x3_cout1_reflect_wgrad(to_act(xin.detach().numpy().astype(np.float32)), d32(dyl), dwl)
    dx = TO.conv3x3_cout1_reflect_bwd_data(d32(dyl), d32(wl.detach().numpy().reshape(9, 32)), out_act(B, H, W, 32))
    torch.cuda.synchronize()
    assert rel_l2(dwl.cpu().numpy().reshape(3, 3, 32, 1), gwl.numpy()) < 2e-5
    assert rel_l2(dx.torch().cpu().numpy(), gx.numpy()) < 2e-6


def test_feature_loss_crop_scatter_and_tanh():
    from emdenoise import gan as GN
    from emdenoise import train_ops as TO
    from oracle import gan_graph as GG
    from tests.test_ops_gpu import rnd

    d32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    a, b = rnd((1, 8, 8, 64), 10), rnd((1, 8, 8, 64), 11)
    at = _t64(a).requires_grad_(True)
    loss = 12.0 * (at - _t64(b)).abs().mean()
    (ga,) = torch.autograd.grad(loss, at)
    base = rnd((1, 8, 8, 64), 12)
    dy, acc = d32(base), torch.zeros(1, dtype=torch.float32, device=dev())
    TO.l1_feature(d32(a), d32(b), 12.0, dy, True, acc)
    torch.cuda.synchronize()
    assert abs(float(acc.cpu()[0]) - float(loss)) < 1e-5 * float(loss)
    assert rel_l2(dy.cpu().numpy() - base, ga.numpy()) < 1e-5
    # crops: gradient of multiscale_crops w.r.t. the image = scatter of the crop gradients through the mirrored indices
    S = 32
    img = _t64(rnd((1, S, S, 1), 13)).requires_grad_(True)
    offs = ((3, 40), (0, 9), (30, 2))
    crops = GG.multiscale_crops(img, offs)
    # the large crop comes back resized; take its gradient at the 3S/4 crop itself for this check
    pad = (3 * S) // 4
    ridx = torch.from_numpy(GG.reflect_indices(S, pad))
    xp = img[:, ridx][:, :, ridx]
    large = xp[:, 30:30 + pad, 2:2 + pad]
    gs, gm, gl = rnd(tuple(crops[0].shape), 14), rnd(tuple(crops[1].shape), 15), rnd(tuple(large.shape), 16)
    (gi,) = torch.autograd.grad((crops[0] * _t64(gs)).sum() + (crops[1] * _t64(gm)).sum() + (large * _t64(gl)).sum(), img)
    dimg = torch.zeros((1, S, S, 1), dtype=torch.float32, device=dev())
    for gcrop, (y0, x0) in zip((gs, gm, gl), offs):
        n = gcrop.shape[1]
        g4 = np.zeros((1, n, n, 4), np.float32)
        g4[..., 0:1] = gcrop
        TO.crop_scatter(d32(g4), 4, dimg, y0, x0, n, S)
    torch.cuda.synchronize()
    assert rel_l2(dimg.cpu().numpy(), gi.numpy()) < 2e-6
    small_dev, _, _ = GN.multiscale_crops(d32(img.detach().numpy()), offs)
    assert np.array_equal(small_dev.cpu().numpy(), crops[0].detach().numpy().astype(np.float32))
    # tanh
    y, dyt = np.tanh(rnd((1000,), 17)), rnd((1000,), 18)
    assert rel_l2(TO.tanh_bwd(d32(dyt), d32(y)).cpu().numpy(), dyt * (1 - y.astype(np.float64) ** 2)) < 1e-6


def test_bn_inference_parameter_gradients():
    """Two moving-statistics batch norms + leaky_relu after a pointwise conv: the fold and d(gamma, beta) of both."""
    from emdenoise import ops
    from emdenoise import train_ops as TO
    from tests.test_ops_gpu import out_act, rnd, to_act

    Cc, eps = 64, 0.01
    d32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    r = rnd((2, 8, 8, Cc), 20, 1.5)
    P = {k: _t64(v).requires_grad_(k[0] in "gb") for k, v in {
        "g1": rnd((Cc,), 21, 0.3) + 1, "b1": rnd((Cc,), 22, 0.3), "m1": rnd((Cc,), 23, 0.3), "v1": np.abs(rnd((Cc,), 24)) + 0.5,
        "g2": rnd((Cc,), 25, 0.3) + 1, "b2": rnd((Cc,), 26, 0.3), "m2": rnd((Cc,), 27, 0.3), "v2": np.abs(rnd((Cc,), 28)) + 0.5}.items()}
    rt = _t64(r).requires_grad_(True)
    z1 = (rt - P["m1"]) * (P["g1"] / torch.sqrt(P["v1"] + eps)) + P["b1"]
    z = (z1 - P["m2"]) * (P["g2"] / torch.sqrt(P["v2"] + eps)) + P["b2"]
    y = torch.nn.functional.leaky_relu(z, 0.2)
    dy = rnd((2, 8, 8, Cc), 29)
    gr, gg1, gb1, gg2, gb2 = torch.autograd.grad(y, (rt, P["g1"], P["b1"], P["g2"], P["b2"]), _t64(dy))
    dv = {k: d32(v.detach().numpy()) for k, v in P.items()}
    f = TO.bn_infer_fold2(dv["g1"], dv["b1"], dv["m1"], dv["v1"], dv["g2"], dv["b2"], dv["m2"], dv["v2"], eps)
    ra, dya = to_act(r), to_act(dy)
    ya = ops.affine_act(ra, f["scale"], f["shift"], out_act(2, 8, 8, Cc), act=ops.ACT_LEAKY)
    s1, t1, t2 = (torch.empty(Cc, dtype=torch.float32, device=dev()) for _ in range(3))
    TO.chan_reduce(dya, s1, ra, f["mprime"], f["rprime"], t2, f["scale"], f["shift"], TO.MASK_LEAKY)
    TO.chan_reduce(dya, s1, ra, f["mean1"], f["rstd1"], t1, f["scale"], f["shift"], TO.MASK_LEAKY)
    grads = [torch.zeros(Cc, dtype=torch.float32, device=dev()) for _ in range(4)]
    TO.bn_infer_grads(s1, t1, t2, f["a2"], *grads)
    torch.cuda.synchronize()
    assert rel_l2(ya.torch().cpu().numpy(), y.detach().numpy()) < 2e-6
    for got, ref in zip(grads, (gg1, gb1, gg2, gb2)):
        assert rel_l2(got.cpu().numpy(), ref.numpy()) < 2e-5


def test_gan_loop_graph_replay():
    """gan_iteration captured into a hipGraph (towers on 2 streams; crop offsets and Adam rates refreshed on the device).
    (a) The first replay equals the eager single-stream iteration to summation order.  (b) Every later replay is checked
    against an eager tower evaluated on the loop's OWN current state just before the replay: D(fake), the adversarial
    and the feature-matching loss must agree -- stale weights, folds, crops or rates inside the graph would show; the
    free-running trajectories themselves separate quickly (Adam's first steps are +-lr per weight, so summation-order
    noise in near-zero gradients flips signs)."""
    from emdenoise import gan as GN
    from emdenoise import gan_trainer as GT

    wg, wd = GN.synthetic_weights(), GN.discriminator_synthetic_weights()
    rng = np.random.default_rng(3)
    pad = (3 * S) // 4
    offs = [[tuple((int(rng.integers(0, S + 2 * pad - n + 1)), int(rng.integers(0, S + 2 * pad - n + 1))) for n in (S // 4, S // 2, pad))
             for _ in range(2)] for _ in range(3)]

    def batch(it):
        hq = images(2, 50 + it)
        lq = GN.gen_lq(hq[..., 0])[..., None]
        return torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())

    # (a) first iteration, eager vs graph
    De, Ge = GT.DiscriminatorTrainer(wd, dev()), None
    Ge = GT.GeneratorTrainer(wg, De, dev())
    x, t = batch(0)
    eg, ed = GT.gan_iteration(Ge, De, x, t, offs[0])
    D = GT.DiscriminatorTrainer(wd, dev())
    G = GT.GeneratorTrainer(wg, D, dev())
    loop = GT.GanLoop(G, D, streams=2)
    gnames, dnames = list(G.trainable), list(D.trainable)
    g0, d0 = flat(wg, gnames), flat(wd, dnames)
    rg, rd = loop.iteration(x, t, offs[0])
    torch.cuda.synchronize()
    ug = rel_l2(flat(G.state_dict(), gnames) - g0, flat(Ge.state_dict(), gnames) - g0)
    ud = rel_l2(flat(D.state_dict(), dnames) - d0, flat(De.state_dict(), dnames) - d0)
    print(f"iteration 0: update diff G {ug:.2e} D {ud:.2e}")
    assert np.allclose(rg.cpu().numpy(), eg.cpu().numpy(), rtol=1e-4) and np.allclose(rd.cpu().numpy(), ed.cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert ug < 1e-3 and ud < 1e-3 and G.t == 1 and D.t == 1
    # (b) later replays against eager towers on the loop's own state
    for it in (1, 2):
        x, t = batch(it)
        before = flat(G.state_dict(), gnames)
        want = []
        for k in range(2):
            G.zero_grad()
            _, r, st = G.tower(x[k:k + 1].contiguous(), t[k:k + 1].contiguous(), offs[it][k])
            want.append(np.concatenate([r.cpu().numpy(), st.cpu().numpy()]))
        rg, rd = loop.iteration(x, t, offs[it])
        torch.cuda.synchronize()
        got = rg.cpu().numpy()
        print(f"iteration {it}: replay {got[:, 0]} vs eager-on-same-state {[w[0] for w in want]}")
        assert np.allclose(got, np.stack(want), rtol=2e-5, atol=1e-6), (it, got, want)
        moved = np.abs(flat(G.state_dict(), gnames) - before)
        assert 0.1 * 2e-4 < np.median(moved[moved > 0]) < 2.5 * 2e-4      # the weights did move: Adam's early steps are ~lr per weight
    assert G.t == 3 and D.t == 3


def test_policy_driven_iterations():
    """The reference trains ONE of the two networks per iteration, chosen by the host policy (gan-infilling-100.py:1700-1704,
    :1903-1939), with randomly flipped labels and adapt weights (:1733-1737, :1772-1776): emdenoise.gan_policy.GanPolicy drives
    gan_iteration(train=, labels=, adapts=).  Checked: only the chosen network's weights move; the per-image labels and adapt
    weights reach the discriminator towers (its losses change accordingly); the policy state advances."""
    from emdenoise import gan as GN
    from emdenoise import gan_policy as GP
    from emdenoise import gan_trainer as GT

    wg, wd = GN.synthetic_weights(), GN.discriminator_synthetic_weights()
    D = GT.DiscriminatorTrainer(wd, dev())
    G = GT.GeneratorTrainer(wg, D, dev())
    rng = np.random.default_rng(11)
    policy = GP.GanPolicy(rng, effective_batch_size=2)
    pad = (3 * S) // 4
    T = 2
    hq = images(T, 90)
    x, t = torch.from_numpy(GN.gen_lq(hq[..., 0])[..., None]).to(dev()), torch.from_numpy(hq).to(dev())
    seen = []
    for counter in range(1, 4):
        offs = [tuple((int(rng.integers(0, S + 2 * pad - n + 1)), int(rng.integers(0, S + 2 * pad - n + 1))) for n in (S // 4, S // 2, pad))
                for _ in range(T)]
        g_before, d_before = G.params.clone(), D.params.clone()
        lr_g, _ = GP.learning_rates(counter)
        train = "gen" if policy.train_gen else "discr"
        labels, adapts = policy.labels(T, T)
        rg, rd = GT.gan_iteration(G, D, x, t, offs, lr_gen=lr_g, labels=labels, adapts=adapts, train=train,
                                  batch_norm_on=GP.batch_norm_on(counter))
        torch.cuda.synchronize()
        g_moved, d_moved = not torch.equal(G.params, g_before), not torch.equal(D.params, d_before)
        assert (g_moved, d_moved) == ((True, False) if train == "gen" else (False, True)), (counter, train, g_moved, d_moved)
        assert (rd is None) == (train == "gen")
        preds_fake = rg[:, 0].cpu().numpy() if rd is None else rd[:T, 0].cpu().numpy()
        preds_real = [] if rd is None else rd[T:, 0].cpu().numpy()
        seen.append(train)
        policy.observe(counter, preds_fake, preds_real)
    assert seen == ["discr", "gen", "discr"]          # the file's constants alternate the trainee every iteration, discriminator first
    assert policy.pred_avg != 0.5
    # labels and adapt weights reach the towers: the same images with flipped labels / another adapt give other losses
    offs = [tuple((pad, pad) for _ in range(3)) for _ in range(T)]
    a = GT.gan_iteration(G, D, x, t, offs, labels=[1e-8, 1e-8, 0.95, 0.95], adapts=[1.0] * 4, train="discr")[1].cpu().numpy().copy()
    D2 = GT.DiscriminatorTrainer(D.state_dict(), dev())
    G2 = GT.GeneratorTrainer(G.state_dict(), D2, dev())
    b = GT.gan_iteration(G2, D2, x, t, offs, labels=[0.95, 1e-8, 0.95, 0.95], adapts=[1.0, 3.0, 1.0, 1.0], train="discr")[1].cpu().numpy()
    assert not np.isclose(a[0, 1], b[0, 1])
