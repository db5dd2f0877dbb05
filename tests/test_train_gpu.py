"""GPU parity tests for graph D' TRAINING (misc_py/denoiser-multi-gpu.py:752-782 tower, :1011-1077 train op):
emdenoise.trainer.DenoiserTrainer (hand-written backward through the C ABI) against the oracle's PyTorch-CPU float64
autograd of the same graph (oracle/denoiser_graph.py tower_gradients), same seeded weights and LQ/HQ pairs.

What bounds gradient parity.  The network is piecewise linear in its relu6 / clip units, so its gradient is a
DISCONTINUOUS function of the forward values: a forward difference of relative size d flips the mask of ~d of the
units, and the gradient then differs by ~sqrt(d) -- for ANY two implementations, e.g. the oracle run in float32 vs
float64 (d ~ 1e-5) disagrees by 3e-3 on the full gradient.  The split-bf16 forward here has d ~ 1.4e-4.  Hence:
  * test_gradients_smooth_regime: batch-norm gamma/beta chosen so that no unit comes near a kink => the whole
    reverse pass (batch-norm backward, weight/data gradients, resampling, concat fan-in, residual accumulation)
    is checked at a tight tolerance;
  * test_gradients_reference_regime: shipped synthetic weights; tolerance reflects the mask flips (measured 2.6e-2
    at 128 px; bound 6e-2) together with cosine similarity; loss / mse / output / moving statistics stay tight;
  * per-op kernels (masks included) are checked at 2e-5 / 2e-6 in tests/test_train_ops_gpu.py.
The oracle is run in float64 (its float32 autograd works too -- see DESIGN.md 4 on the round-1 crash record -- and is used where
the float32-vs-float64 spread itself is the quantity of interest).
"""
import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_pair

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))


def dev():
    return torch.device("cuda", 0)


def weights(smooth=False):
    from emdenoise import denoiser as D

    w = D.synthetic_weights(variant="Dprime")
    if smooth:
        outer = sorted((n for n in w if n.startswith("nn/BatchNorm") and n.endswith("/gamma")),
                       key=lambda n: int(n.split("/")[1].split("_")[1]) if "_" in n.split("/")[1] else 0)
        for n in outer:   # relu6 units: z = 0.15*xhat + 3 stays inside (0, 6) for |xhat| < 20
            w[n] = np.full_like(w[n], 0.15)
            w[n[:-5] + "beta"] = np.full_like(w[n], 3.0)
        last = outer[-1]  # the output unit is also clipped to [0,1]: z = 0.02*xhat + 0.5
        w[last] = np.full_like(w[last], 0.02)
        w[last[:-5] + "beta"] = np.full_like(w[last], 0.5)
    return w


def flat(d, names):
    return np.concatenate([np.asarray(d[n], np.float64).ravel() for n in names])


def run_tower(w, lq, hq):
    from emdenoise import trainer as TR

    tr = TR.DenoiserTrainer(w, dev())
    tr.zero_grad()
    out, res = tr.tower(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()))
    torch.cuda.synchronize()
    return tr, out.cpu().numpy(), res.cpu().numpy()


def test_gradients_smooth_regime():
    from oracle import denoiser_graph as G

    S, B = 64, 2
    w = weights(smooth=True)
    lq, hq = synthetic_pair(B, S, S, seed=5)
    acts = []
    ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64, trace=acts)
    tr, out, res = run_tower(w, lq, hq)
    o = ref["out"].numpy()
    lo, hi = min(float(a.min()) for a in acts), max(float(a.max()) for a in acts)
    assert 0.01 < lo and hi < 5.99 and 0.001 < o.min() and o.max() < 0.999, "the smooth regime must stay off every kink"
    g = tr.gradients()
    names = [n for n in g if np.abs(ref["grads"][n]).max() > 1e-9]
    e_all = rel_l2(flat(g, names), flat(ref["grads"], names))
    worst = max((rel_l2(g[n], ref["grads"][n]), n) for n in names)
    print(f"smooth regime: out {rel_l2(out, o):.2e}  all-grads {e_all:.2e}  worst tensor {worst[0]:.2e} {worst[1]}")
    assert rel_l2(out, o) < 1e-4
    assert abs(res[0] - ref["mse"]) < 2e-5 * ref["mse"] and abs(res[1] - ref["loss"]) < 2e-5 * ref["loss"]
    assert e_all < 5e-4
    assert worst[0] < 5e-3
    # gradients that are zero analytically (bias / beta in front of a batch norm): zero here, rounding noise in TF
    gmax = max(float(np.abs(ref["grads"][n]).max()) for n in names)
    for n in g:
        if n not in names:
            assert float(np.abs(g[n]).max()) < 1e-6 * gmax, n


def test_gradients_reference_regime():
    from oracle import denoiser_graph as G

    S, B = 128, 1
    w = weights()
    lq, hq = synthetic_pair(B, S, S, seed=3)
    ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64)
    tr, out, res = run_tower(w, lq, hq)
    g = tr.gradients()
    names = [n for n in g if np.abs(ref["grads"][n]).max() > 1e-9]
    a, b = flat(g, names), flat(ref["grads"], names)
    print(f"reference regime: out {rel_l2(out, ref['out'].numpy()):.2e} all-grads {rel_l2(a, b):.2e} cos {cosine(a, b):.6f}")
    assert rel_l2(out, ref["out"].numpy()) < 1e-3                     # north_star bar for images
    assert abs(res[0] - ref["mse"]) < 1e-4 * ref["mse"] and abs(res[1] - ref["loss"]) < 1e-4 * ref["loss"]
    assert rel_l2(a, b) < 6e-2 and cosine(a, b) > 0.998                  # mask flips, see the module docstring
    st = tr.state_dict()
    for n, v in ref["moving"].items():                                  # decay-0.999 moving statistics after one tower
        assert rel_l2(st[n], v) < 1e-6, n


def test_gradients_with_the_oracles_forward_values():
    """VERDICT r1 item 1e -- is the 2-3e-2 of the reference regime really mask flips?  The trainer's forward pass is teacher
    forced at every conv output (DenoiserTrainer.teacher: the depthwise, pointwise / conv / transposed-conv results are replaced
    by the oracle's float64 values right after the kernels have produced them), so batch statistics, activations and the relu6 /
    clip masks of the backward pass are the oracle's to float32 rounding; the BACKWARD pass is untouched (same kernels, same
    split-bf16 GEMMs).  Measured: 1.74e-2 free running -> 1.74e-5 forced (bar 5e-5, the per-kernel level): the discrepancy of the
    free-running run IS its forward error acting through the masks, not the backward kernels."""
    from emdenoise import trainer as TR
    from oracle import denoiser_graph as G

    S, B = 64, 1
    w = weights()
    lq, hq = synthetic_pair(B, S, S, seed=3)
    saved = {}
    ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64, saved=saved)
    tr = TR.DenoiserTrainer(w, dev())
    teacher = {}
    for key, L in tr.layers.items():
        if L.scope not in saved:
            continue
        t = {k: v.numpy() for k, v in saved[L.scope].items()}
        if L.kind in ("conv", "deconv") and L.bn:      # the trainer's GEMM leaves the bias to the batch mean (it cancels there)
            t["r"] = t["r"] - np.asarray(w[L.scope + "/" + L.bname], np.float64)
        if L.kind == "conv" and L.cout == 1:            # the final 3x3 -> 1 conv likewise runs without its bias
            pass
        teacher[L.scope] = t
    x, t_ = torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())
    names = [n for n in ref["grads"] if np.abs(ref["grads"][n]).max() > 1e-9]
    b = flat(ref["grads"], names)
    tr.zero_grad()
    tr.tower(x, t_)
    free = rel_l2(flat(tr.gradients(), names), b)
    tr.teacher = teacher
    tr.zero_grad()
    out, res = tr.tower(x, t_)
    torch.cuda.synchronize()
    forced = rel_l2(flat(tr.gradients(), names), b)
    print(f"gradient vs float64 oracle at {S} px: free running {free:.2e}; with the oracle's conv outputs forced into the forward pass "
          f"{forced:.2e} (output {rel_l2(out.cpu().numpy(), ref['out'].numpy()):.1e}, loss {abs(res[1].item() - ref['loss']) / ref['loss']:.1e})")
    assert rel_l2(out.cpu().numpy(), ref["out"].numpy()) < 2e-6
    assert forced < 5e-5 and forced < free / 100


# Bars of the 512^2 tower test = about 2x what the run on record measures (see the docstring); before round 3 they were scaled by the
# oracle's own float32-vs-float64 noise and let rel L2 0.44 / cosine 0.935 pass.
BAR_512 = {"norm_ratio": 5e-3, "rel": 3.5e-2, "cos": 0.9995, "worst_norm": 0.15}


def test_tower_at_512_against_the_committed_golden():
    """BASELINE configs[3]'s size: ONE tower of graph D' on a 512x512 LQ/HQ pair against tests/golden/dprime_tower_512.json
    (float64 oracle autograd, tests/golden/make_train_golden.py): loss, mse, 64 output probes, the gradient norm of every
    trainable variable, and the relative L2 error / cosine of the whole 38.5 M-element gradient estimated from 48 seeded
    Rademacher projections.  The gradient of a relu6 / clip network is a discontinuous function of the forward values (the
    fixture records that the oracle's own float32 run is 0.11 from its float64 run at this size), so the bars are empirical:
    MEASURED on MI355X (round 3, gpurun_out/r3o/train512.log): |g| 73.7935 vs 73.8028 (1.3e-4), estimated relative L2 1.52e-2,
    cosine 0.99988, worst per-variable norm ratio off by 7.0e-2; BAR_512 is about twice that.  The backward KERNELS are held to
    2e-5 / 1e-5 on their own at these layer shapes in tests/test_train_ops_gpu.py (free of mask flips)."""
    import hashlib
    import json
    import os

    meta = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dprime_tower_512.json")))
    S = meta["S"]
    lq, hq = synthetic_pair(1, S, S, seed=meta["seed"])
    assert hashlib.sha256(lq.tobytes()).hexdigest() == meta["lq_sha256"], "synthetic input generator changed"
    tr, out, res = run_tower(weights(), lq, hq)
    assert abs(res[0] - meta["mse"]) < 1e-4 * meta["mse"] and abs(res[1] - meta["loss"]) < 1e-4 * meta["loss"]
    pr = np.array(meta["probes"])
    assert rel_l2(out[0, pr[:, 0], pr[:, 1], 0], meta["out_values"]) < 1e-3 and abs(out.mean() - meta["out_mean"]) < 1e-4
    g = tr.gradients()
    names = meta["names"]
    norms = np.array([np.linalg.norm(np.asarray(g[n], np.float64)) for n in names])
    ref_norms = np.array(meta["grad_l2"])
    big = ref_norms > 1e-3 * ref_norms.max()
    worst_norm = float(np.max(np.abs(norms[big] / ref_norms[big] - 1.0)))
    a = flat(g, names)
    K, ps = meta["K"], meta["projection_seed"]
    proj = []
    for k in range(K):
        sgn = np.random.default_rng(ps + k).integers(0, 2, a.size, dtype=np.int8)
        proj.append(float(a[sgn == 1].sum() - a[sgn == 0].sum()))
    e = np.array(proj) - np.array(meta["projections"])
    rel = float(np.sqrt(np.mean(e * e)) / meta["flat_l2"])                      # E[(v.e)^2] = |e|^2
    na = float(np.linalg.norm(a))
    cos = (na * na + meta["flat_l2"] ** 2 - float(np.mean(e * e))) / (2 * na * meta["flat_l2"])
    print(f"D' tower at 512^2 vs golden: |g| {na:.4f} vs {meta['flat_l2']:.4f}, estimated rel L2 {rel:.2e} cos {cos:.5f}, worst "
          f"per-variable norm ratio off by {worst_norm:.2e}; oracle float32 vs float64: {meta['oracle_f32_vs_f64']}")
    # At this size the oracle's OWN float32 run is 0.11 (cosine 0.9935) from its float64 run (fixture: oracle_f32_vs_f64): two
    # thirds of a million relu6 / clip units sit within rounding distance of a kink.  The bar for the split-bf16 forward (about
    # 10x float32's forward error => about 3x its mask flips) is therefore set relative to that figure, not to a constant.
    o = meta["oracle_f32_vs_f64"]
    assert abs(na / meta["flat_l2"] - 1.0) < BAR_512["norm_ratio"]
    assert rel < BAR_512["rel"] and cos > BAR_512["cos"]
    assert worst_norm < BAR_512["worst_norm"]


def test_tower_batch_and_accumulation():
    """Two towers of one image each ADD their gradient sets; one tower of two images normalises over both."""
    from oracle import denoiser_graph as G

    S = 64
    w = weights(smooth=True)
    lq, hq = synthetic_pair(2, S, S, seed=7)
    from emdenoise import trainer as TR

    tr = TR.DenoiserTrainer(w, dev())
    tr.zero_grad()
    for k in range(2):
        tr.tower(torch.from_numpy(lq[k:k + 1]).to(dev()), torch.from_numpy(hq[k:k + 1]).to(dev()), update_moving=(k == 0))
    g = tr.gradients()
    r0 = G.tower_gradients(lq[0:1], hq[0:1], w, S, dtype=torch.float64)
    r1 = G.tower_gradients(lq[1:2], hq[1:2], w, S, dtype=torch.float64)
    names = [n for n in g if np.abs(r0["grads"][n]).max() > 1e-9]
    ref = flat(r0["grads"], names) + flat(r1["grads"], names)
    assert rel_l2(flat(g, names), ref) < 1e-3
    st = tr.state_dict()
    for n, v in r0["moving"].items():   # only the first tower moves the statistics (:701-707)
        assert rel_l2(st[n], v) < 1e-6, n


def test_gradient_accumulation_run_to_run_spread():
    """The parameter gradients are accumulated with float atomics (csrc/wgrad.hip: M-slices of a weight-gradient GEMM, towers on
    concurrent streams), whose order is not fixed: a training step is NOT bit-reproducible.  What is guaranteed and checked
    here: the forward pass (output, loss, moving statistics) is bit-identical run to run, and the accumulated gradient differs
    by float32 reassociation only -- relative L2 below 1e-6 between repeats (measured ~1e-7), three orders of magnitude under
    the gradient's own parity bar."""
    from emdenoise import trainer as TR

    S = 128
    w = weights()
    lq, hq = synthetic_pair(4, S, S, seed=21)
    x, t = torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())
    runs = []
    for rep in range(3):
        tr = TR.DenoiserTrainer(w, dev())
        res = tr.local_gradients(x, t, 1, 4)              # 4 towers of one image on 4 streams: concurrent atomics
        torch.cuda.synchronize()
        runs.append((res.cpu().numpy().copy(), tr.grads.detach().cpu().numpy().astype(np.float64), tr.moving.detach().cpu().numpy().copy()))
    spread = max(rel_l2(runs[k][1], runs[0][1]) for k in (1, 2))
    print(f"gradient accumulation, 3 repeats of 4 concurrent towers at {S} px: run-to-run relative L2 {spread:.2e}, "
          f"bit-identical: {all(np.array_equal(runs[k][1], runs[0][1]) for k in (1, 2))}")
    for k in (1, 2):
        np.testing.assert_array_equal(runs[k][0], runs[0][0])      # mse / loss per tower: deterministic
        np.testing.assert_array_equal(runs[k][2], runs[0][2])      # moving statistics: deterministic
    assert spread < 1e-6


def test_batched_per_image_towers_equal_towers_of_one():
    """the one-image towers of a rank (misc_py/denoiser-multi-gpu.py:763) as ONE batched pass with per-image
    batch-norm statistics (DenoiserTrainer.tower(per_image=True); emd_bn_stats_images_f32, emd_bn_*_images_f32) against the
    same images as separate towers: outputs, per-image mse / loss and the moving statistics (image 0) bit for bit -- every
    per-image reduction runs exactly as it would alone, every other kernel is batch-independent -- and the accumulated
    gradient to the run-to-run spread of its float atomics (1e-6)."""
    from emdenoise import trainer as TR

    S, B = 64, 4
    w = weights()
    lq, hq = synthetic_pair(B, S, S, seed=17)
    x, t = torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())
    a = TR.DenoiserTrainer(w, dev())
    outs, ress = [], []
    a.zero_grad()
    for k in range(B):
        o, r = a.tower(x[k:k + 1].contiguous(), t[k:k + 1].contiguous(), update_moving=(k == 0))
        outs.append(o.clone())
        ress.append(r.clone())
    torch.cuda.synchronize()
    b = TR.DenoiserTrainer(w, dev())
    b.zero_grad()
    ob, rb = b.tower(x, t, update_moving=True, per_image=True)
    torch.cuda.synchronize()
    assert torch.equal(ob, torch.cat(outs)) and torch.equal(rb, torch.stack(ress))
    assert torch.equal(a.moving, b.moving)
    ga, gb = a.grads.detach().cpu().numpy().astype(np.float64), b.grads.detach().cpu().numpy().astype(np.float64)
    spread = rel_l2(gb, ga)
    print(f"batched per-image towers vs {B} towers of one at {S} px: outputs, losses, moving statistics identical; gradient rel L2 {spread:.2e}")
    assert spread < 2e-6
    # the same pass with its weight-gradient launches on a side stream (opt-in, EMD_T_WGRAD_STREAM=1)
    c = TR.DenoiserTrainer(w, dev())
    c.zero_grad()
    oc, rc = c.tower(x, t, update_moving=True, per_image=True, wgrad_stream=True)
    torch.cuda.synchronize()
    assert torch.equal(oc, ob) and torch.equal(rc, rb) and torch.equal(c.moving, b.moving)
    assert rel_l2(c.grads.detach().cpu().numpy().astype(np.float64), gb) < 2e-6
    # and through train_step: the same update from either form
    pa, pb = TR.DenoiserTrainer(w, dev()), TR.DenoiserTrainer(w, dev())
    pa.train_step(x, t, tower_batch=1, streams=2)
    pb.train_step(x, t, tower_batch=1, batched=True)
    torch.cuda.synchronize()
    assert rel_l2(pb.params.cpu().numpy(), pa.params.cpu().numpy()) < 1e-7


def test_affine_in_the_consumers_loads_changes_nothing():
    """Round 4: the affine + relu6 of a separable conv that feeds only the next one is applied in that one's loads (ops.PreAct,
    emd_dw3x3_pre_act_f32 / emd_dw3x3_wgrad_pre_f32; DenoiserTrainer.lazy_affine, the default) -- against the written-out form
    (lazy_affine = False): outputs, losses and moving statistics bit for bit (the loads rebuild affine_relu6_kernel's bits), the
    accumulated gradient to the run-to-run spread of its float atomics.  A tower of several images (batch statistics) and a batched
    pass of one-image towers (per-image statistics)."""
    from emdenoise import train_ops as TO, trainer as TR

    S, B = 64, 3
    w = weights()
    lq, hq = synthetic_pair(B, S, S, seed=23)
    x, t = torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev())
    for per_image in (False, True):
        res = {}
        # (affine in the loads, the never-written gradient: TO.DwGrad / emd_dw3x3_bn_bwd_*_f32, the consumer's depthwise weight gradient inside
        # its reduction pass): the default, each step back, all written out
        for mode in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
            tr = TR.DenoiserTrainer(w, dev())
            assert tr.lazy_affine and tr.fuse_dw_bn_bwd and tr.fuse_dw_wgrad and tr.fuse_dw_both
            tr.lazy_affine, tr.fuse_dw_bn_bwd, tr.fuse_dw_wgrad = mode
            tr.fuse_dw_both = mode[0]     # (the two depthwise gradients of a written input in one pass: on with the default only)
            # the per-channel steps of the BN chain inside the kernels that finish the reductions in front of them (emd_*_stats_fold_f32,
            # emd_bn_bwd_reduce_prep_f32): off in the all-written-out form
            assert tr.fuse_fold and TO.FUSE_PREP and not tr.fuse_cout1_grad
            tr.fuse_fold = TO.FUSE_PREP = mode[0]
            tr.fuse_cout1_grad = mode == (True, True, False)     # (opt-in: the final conv's data gradient never written, TO.Cout1Grad)
            try:
                tr.zero_grad()
                o, r = tr.tower(x, t, update_moving=True, per_image=per_image)
                torch.cuda.synchronize()
            finally:
                TO.FUSE_PREP = True
            res[mode] = (o.clone(), r.clone(), tr.moving.clone(), tr.grads.detach().cpu().numpy().astype(np.float64))
        ref = res[(False, False, False)]
        for mode in ((True, True, True), (True, True, False), (True, False, False)):
            assert torch.equal(res[mode][0], ref[0]) and torch.equal(res[mode][1], ref[1]) and torch.equal(res[mode][2], ref[2])
            spread = rel_l2(res[mode][3], ref[3])
            print(f"affine in the consumer's loads {mode[0]}, gradient never written {mode[1]}, consumer's depthwise weight gradient in the reduction {mode[2]} vs all written out, per_image={per_image}: "
                  f"outputs / losses / moving statistics identical; gradient rel L2 {spread:.2e}")
            assert spread < 2e-6


def test_train_steps_follow_the_oracle():
    """Free-running: three optimizer steps (2 towers of 1 image, averaged; Nesterov momentum 0.9, lr 1e-3; moving
    statistics from tower 0) against the oracle's float64 loop, in the regime where no unit sits on a kink."""
    from emdenoise import trainer as TR
    from oracle import denoiser_graph as G

    S = 64
    w = weights(smooth=True)
    tr = TR.DenoiserTrainer(w, dev())
    params = {n: np.asarray(w[n], np.float64) for n in tr.trainable}
    moving = {n: np.asarray(w[n], np.float64) for n in tr.moving_names}
    accum = {n: np.zeros_like(v) for n, v in params.items()}
    names = list(params)
    for step in range(3):
        lq, hq = synthetic_pair(2, S, S, seed=100 + step)
        before = flat(tr.state_dict(), names)
        res = tr.train_step(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), tower_batch=1).cpu().numpy()
        cur = dict(params)
        cur.update(moving)
        towers = [G.tower_gradients(lq[k:k + 1], hq[k:k + 1], cur, S, dtype=torch.float64) for k in range(2)]
        grads = {n: 0.5 * (towers[0]["grads"][n] + towers[1]["grads"][n]) for n in params}
        old = flat(params, names)
        params, accum = G.nesterov_step(params, grads, accum, lr=0.001, momentum=0.9)
        moving.update(towers[0]["moving"])
        st = tr.state_dict()
        delta = rel_l2(flat(st, names) - before, flat(params, names) - old)
        print(f"step {step}: losses {res[:, 1]} vs {[round(t['loss'], 5) for t in towers]}; update rel-l2 {delta:.2e}")
        for k in range(2):
            assert abs(res[k, 1] - towers[k]["loss"]) < 1e-4 * towers[k]["loss"], (step, k, res[k], towers[k]["loss"])
        assert delta < 1e-3
        for n, v in moving.items():
            assert rel_l2(st[n], v) < 1e-5, n


def test_train_steps_reference_regime_teacher_forced():
    """Shipped synthetic weights, three optimizer steps.  In this early transient the trajectory is chaotic (a 2 %
    difference in the first update changes the next gradient by O(1)), so every step is checked at the TRAINER's own
    parameters: loss, gradient (mask-flip tolerance, module docstring), and the Nesterov update applied to the
    trainer's own gradient."""
    from emdenoise import trainer as TR
    from oracle import denoiser_graph as G

    S = 64
    tr = TR.DenoiserTrainer(weights(), dev())
    names = list(tr.trainable)
    accum = {n: np.zeros(s, np.float64) for n, s in tr.trainable.items()}
    for step in range(3):
        lq, hq = synthetic_pair(2, S, S, seed=100 + step)
        cur = tr.state_dict()
        res = tr.train_step(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), tower_batch=1).cpu().numpy()
        g = tr.gradients()   # the sum over the two towers
        towers = [G.tower_gradients(lq[k:k + 1], hq[k:k + 1], cur, S, dtype=torch.float64) for k in range(2)]
        ref = {n: towers[0]["grads"][n] + towers[1]["grads"][n] for n in names}
        live = [n for n in names if np.abs(ref[n]).max() > 1e-9]
        a, b = flat(g, live), flat(ref, live)
        print(f"step {step}: losses {res[:, 1]} vs {[round(t['loss'], 5) for t in towers]}; grads rel-l2 {rel_l2(a, b):.2e} cos {cosine(a, b):.5f}")
        for k in range(2):
            assert abs(res[k, 1] - towers[k]["loss"]) < 1e-4 * towers[k]["loss"], (step, k, res[k], towers[k]["loss"])
        assert rel_l2(a, b) < 6e-2 and cosine(a, b) > 0.998, (step, rel_l2(a, b), cosine(a, b))
        newp, accum = G.nesterov_step({n: np.asarray(cur[n], np.float64) for n in names},
                                      {n: 0.5 * np.asarray(g[n], np.float64) for n in names}, accum, lr=0.001, momentum=0.9)
        st = tr.state_dict()
        upd = rel_l2(flat(st, names) - flat(cur, names), flat(newp, names) - flat(cur, names))
        assert upd < 3e-4, (step, upd)      # fp32 rounding of the parameter update itself
        for n, v in towers[0]["moving"].items():   # (absolute floor: a moving mean can sit near zero)
            assert np.allclose(st[n], v, rtol=1e-4, atol=1e-6), (step, n, rel_l2(st[n], v))


def test_streams_and_graph_match_eager():
    """Towers issued on 4 HIP streams, and the same captured into a hipGraph and replayed, against the plain
    one-stream eager step: same losses, same update (parameter gradients meet only through float atomics, so the
    steps agree to summation order; smooth-regime weights keep the second step free of relu6 mask flips)."""
    from emdenoise import trainer as TR

    S = 64
    w = weights(smooth=True)
    names = [n for n in w if not n.endswith(("/moving_mean", "/moving_variance"))]
    runs = {}
    for mode, kw in (("eager", {}), ("streams", {"streams": 4}), ("graph", {"streams": 4, "graph": True})):
        tr = TR.DenoiserTrainer(w, dev())
        out = []
        for step in range(3):   # a larger learning rate than the reference's: weights stale by one step would show
            lq, hq = synthetic_pair(4, S, S, seed=200 + step)
            res = tr.train_step(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()), tower_batch=1,
                                learning_rate=0.003, **kw)
            torch.cuda.synchronize()
            out.append((res.cpu().numpy(), flat(tr.state_dict(), names), flat(tr.state_dict(), [n for n in w if n not in names])))
        runs[mode] = out
    p0 = flat(w, names)
    for mode in ("streams", "graph"):
        for step, tol in ((0, 1e-5), (1, 1e-4), (2, 1e-4)):
            r, p, m = runs[mode][step]
            re, pe, me = runs["eager"][step]
            prev = p0 if step == 0 else runs["eager"][step - 1][1]
            assert np.allclose(r[:, :2], re[:, :2], rtol=1e-5 if step == 0 else 1e-4), (mode, step)
            assert rel_l2(p - prev, pe - prev) < tol, (mode, step, rel_l2(p - prev, pe - prev))
            assert rel_l2(m, me) < 1e-5, (mode, step)


def test_trainer_checkpoint_round_trip(tmp_path):
    """The trainer writes a TensorFlow checkpoint bundle that the inference engine's loader reads back."""
    from emdenoise import denoiser as D
    from emdenoise import tf_checkpoint as ck
    from emdenoise import trainer as TR

    tr = TR.DenoiserTrainer(weights(), dev())
    lq, hq = synthetic_pair(1, 64, 64, seed=9)
    tr.train_step(torch.from_numpy(lq).to(dev()), torch.from_numpy(hq).to(dev()))
    prefix = tr.save_checkpoint(str(tmp_path), global_step=1)
    st = tr.state_dict()
    loaded = D.load_weights(str(tmp_path), variant="Dprime")
    assert list(loaded) == list(st) and all(np.array_equal(loaded[k], st[k]) for k in st)
    everything = ck.read_checkpoint(prefix)
    assert int(everything["global_step"]) == 1 and float(np.abs(everything["nn/1x1/kernel/Momentum"]).max()) > 0
