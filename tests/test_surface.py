"""The reference-named graph surface (SURVEY.md 8b "Graph surface"):
  architecture(inputs, ground_truth, phase, params)                      machine_learning/denoiser.py:58-61
  get_model_fn(num_gpus, variable_strategy, num_workers) -> _model_fn    misc_py/denoiser-multi-gpu.py:634-717
  generator_architecture(inputs, phase, params, train_batch_norm)        misc_py/gan-infilling-100.py:133
CPU part: names, argument order and error behaviour (nothing here touches a GPU).  GPU part: the values, against the oracle."""
import inspect

import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq, synthetic_pair


def test_signatures_mirror_the_reference():
    import emdenoise
    from emdenoise import gan, trainer

    p = list(inspect.signature(emdenoise.architecture).parameters)
    assert p[:4] == ["inputs", "ground_truth", "phase", "params"]                      # denoiser.py:58-61
    assert inspect.signature(emdenoise.architecture).parameters["phase"].default is False
    p = list(inspect.signature(trainer.get_model_fn).parameters)
    assert p[:3] == ["num_gpus", "variable_strategy", "num_workers"]                   # denoiser-multi-gpu.py:634
    fn = trainer.get_model_fn(2, "GPU", 1)
    p = list(inspect.signature(fn).parameters)
    assert p == ["features", "labels", "mode", "params"]                               # :637
    p = list(inspect.signature(gan.generator_architecture).parameters)
    assert p[:4] == ["inputs", "phase", "params", "train_batch_norm"]                  # gan-infilling-100.py:133
    assert emdenoise.get_model_fn is trainer.get_model_fn


def test_missing_engine_or_trainer_is_an_argument_error_not_a_stub():
    import emdenoise
    from emdenoise import gan, trainer

    x = torch.zeros(1, 32, 32, 1)
    with pytest.raises(ValueError, match="engine"):
        emdenoise.architecture(x)
    with pytest.raises(ValueError, match="trainer"):
        emdenoise.architecture(x, None, True)
    with pytest.raises(ValueError, match="engine"):
        gan.generator_architecture(x)
    with pytest.raises(ValueError, match="trainer"):
        gan.generator_architecture(x, train_batch_norm=True)
    fn = trainer.get_model_fn(1, "CPU", 1)
    with pytest.raises(ValueError, match="trainer"):
        fn([x], [x], True, None)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.gpu
def test_architecture_dispatches_on_phase():
    """phase=False -> the inference engine; phase=True -> the tower's training-mode forward (batch statistics), checked against the
    oracle's architecture(phase=True) output of the same tower; no moving statistic moves (the builder has no update op)."""
    import emdenoise
    from emdenoise import denoiser as D, trainer as TR
    from oracle import denoiser_graph as G

    dev = torch.device("cuda", 0)
    S = 64
    w = D.synthetic_weights(variant="Dprime")
    lq, hq = synthetic_pair(2, S, S, seed=8)
    x = torch.from_numpy(lq).to(dev)
    eng = emdenoise.DenoiserEngine(w, dev, "bf16x3", variant="Dprime")
    y0 = emdenoise.architecture(x, None, False, {"engine": eng})
    assert torch.equal(y0, eng.forward(x))
    tr = TR.DenoiserTrainer(w, dev)
    before = tr.moving.clone()
    y1 = emdenoise.architecture(x, torch.from_numpy(hq).to(dev), True, {"trainer": tr})
    ref = G.tower_gradients(lq, hq, w, S, dtype=torch.float64)
    assert y1.shape == (2, S, S, 1) and rel_l2(y1.cpu().numpy(), ref["out"].numpy()) < 1e-3
    assert torch.equal(tr.moving, before)
    assert not torch.equal(y1, y0)        # batch statistics != moving statistics


@pytest.mark.gpu
def test_model_fn_returns_the_reference_list():
    """[losses, preds, mses, update_ops] + tower_grads (denoiser-multi-gpu.py:709-715) for two towers: every tower uses image 0 of
    its shard (:763), gradients are separate sets in trainable-variable order, update_ops are tower 0's moving statistics, the
    stacked prediction is the last tower's (:711)."""
    from emdenoise import denoiser as D, trainer as TR
    from oracle import denoiser_graph as G

    from tests.test_train_gpu import weights as train_weights

    dev = torch.device("cuda", 0)
    S = 64
    w = train_weights(smooth=True)      # no relu6 / clip unit near a kink: the whole reverse pass compares tightly (tests/test_train_gpu.py)
    lq, hq = synthetic_pair(4, S, S, seed=12)
    tr = TR.DenoiserTrainer(w, dev)
    fn = TR.get_model_fn(2, "GPU", 1, trainer=tr)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    features, labels = [up(lq[0:2]), up(lq[2:4])], [up(hq[0:2]), up(hq[2:4])]
    out = fn(features, labels, True, None)
    losses, preds, mses, update_ops = out[:4]
    grads = out[4:]
    assert len(losses) == len(mses) == len(grads) == 2 and preds.shape == (1, 1, S, S, 1)
    names = list(tr.trainable)
    assert all(len(g) == len(names) for g in grads)
    for i, first in enumerate((0, 2)):
        ref = G.tower_gradients(lq[first:first + 1], hq[first:first + 1], w, S, dtype=torch.float64)
        assert abs(float(mses[i]) - ref["mse"]) < 1e-4 * ref["mse"] and abs(float(losses[i]) - ref["loss"]) < 1e-4 * ref["loss"]
        nz = [k for k, n in enumerate(names) if np.abs(ref["grads"][n]).max() > 1e-9]
        a = np.concatenate([grads[i][k].cpu().numpy().astype(np.float64).ravel() for k in nz])
        b = np.concatenate([np.asarray(ref["grads"][names[k]], np.float64).ravel() for k in nz])
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        assert cos > 0.99999 and rel_l2(a, b) < 2e-3, (i, cos, rel_l2(a, b))
        if i == 1:
            assert rel_l2(preds[0].cpu().numpy(), ref["out"].numpy()) < 1e-3      # the LAST tower's prediction
        if i == 0:
            assert update_ops["__changed__"]
            for n, v in ref["moving"].items():
                assert rel_l2(update_ops[n].cpu().numpy(), v) < 1e-6, n
    assert not torch.equal(grads[0][5], grads[1][5])


@pytest.mark.gpu
def test_generator_architecture_train_batch_norm():
    """train_batch_norm=True == the batch-statistics phase: GeneratorTrainer.update_moving_statistics (its parity with the oracle is
    tests/test_gan_train_gpu.py::test_generator_moving_statistics_phase); falsy == GeneratorEngine.forward."""
    from emdenoise import gan, gan_trainer as GT

    dev = torch.device("cuda", 0)
    S = 64
    w = gan.synthetic_weights()
    hq = synthetic_lq(1, S, S, seed=2) * 2.0 - 1.0
    lq = gan.gen_lq(hq[0, :, :, 0], frac=1.0 / 16).reshape(1, S, S, 1).astype(np.float32)
    x = torch.from_numpy(lq).to(dev)
    eng = gan.GeneratorEngine(w, dev)
    assert torch.equal(gan.generator_architecture(x, False, {"engine": eng}, None), eng.forward(x))
    dtr = GT.DiscriminatorTrainer(gan.discriminator_synthetic_weights(), dev)
    gtr = GT.GeneratorTrainer(w, dtr, dev)
    twin = GT.GeneratorTrainer(w, GT.DiscriminatorTrainer(gan.discriminator_synthetic_weights(), dev), dev)
    before = gtr.state_dict()
    y = gan.generator_architecture(x, True, {"trainer": gtr}, True)
    want = twin.update_moving_statistics(x)
    assert torch.equal(y, want) and y.shape == (1, S, S, 1)
    after, ta = gtr.state_dict(), twin.state_dict()
    moved = [n for n in after if n.endswith(("moving_mean", "moving_variance")) and not np.array_equal(after[n], before[n])]
    assert moved and all(np.array_equal(after[n], ta[n]) for n in after)
    with pytest.raises(ValueError, match="ONE image"):
        gan.generator_architecture(torch.cat([x, x]), True, {"trainer": gtr}, True)
