"""CPU tests of emdenoise.gan_policy: the host-side policy of the in-filling GAN's training loop
(misc_py/gan-infilling-100.py:1607-1647, :1724-1776, :1903-1939) -- schedules, label flips, adapt weights, trainee switching."""
import math

import numpy as np

from emdenoise import gan_policy as P


class Seq:
    """A 'generator' that returns a scripted sequence from .random()."""

    def __init__(self, values):
        self.values = list(values)

    def random(self):
        return self.values.pop(0)


def test_learning_rate_schedule():
    assert P.learning_rates(0) == (0.0002, 0.0001) and P.learning_rates(349999) == (0.0002, 0.0001)
    g, d = P.learning_rates(350000)                       # step 1 of 8
    assert math.isclose(g, 0.0002 * (1 - 1 / 8)) and math.isclose(d, g / 2)
    g, _ = P.learning_rates(649999)                       # step (299999 // 50000) + 1 = 6
    assert math.isclose(g, 0.0002 * (1 - 6 / 8))
    assert P.learning_rates(700000)[0] == 0.0             # step 8: rate 0, still inside the loop
    assert P.learning_rates(700001) is None               # :1639-1641: save and quit
    assert P.batch_norm_on(249999) and not P.batch_norm_on(250000)


def test_label_flips_and_adapt():
    p = P.GanPolicy(Seq([0.5, 1e-9, 0.25, 0.5, 1e-9]))
    assert p.pred_avg == 0.5 and p.pred_avg_real == 0.5
    prob = 0.01 * 0.5 ** 7
    # generated image, no flip (0.5 > prob): label 1e-8, adapt = 10 e^-p (1 - e^-p^2)
    label, adapt = p.fake_label()
    assert label == 1e-8 and math.isclose(adapt, 10 * math.exp(-0.5) * (1 - math.exp(-0.25)))
    # generated image, flipped (1e-9 <= prob): label in [0.9, 1), adapt 1
    label, adapt = p.fake_label()
    assert math.isclose(label, 0.9 + 0.1 * 0.25 - 1e-8) and adapt == 1.0 and 1e-9 <= prob
    # natural image, no flip: label in [0.9, 1) drawn AFTER the flip draw; adapt always 1
    p.rng = Seq([0.5, 0.75])
    label, adapt = p.real_label()
    assert math.isclose(label, 0.9 + 0.075 - 1e-8) and adapt == 1.0
    p.rng = Seq([1e-9])
    assert p.real_label() == (1e-8, 1.0)
    labels, adapts = P.GanPolicy(np.random.default_rng(0)).labels(3, 2)
    assert len(labels) == len(adapts) == 5 and all(a == 1.0 for a in adapts[3:])


def test_trainee_switching_with_the_files_constants():
    """trainee_switch_skip_n = 1, max_num_since_training_change = 0 (:124-126): num_since_change >= 0 always holds, so the trainee
    alternates every iteration, starting with the discriminator (train_gen = False, :1611)."""
    p = P.GanPolicy(np.random.default_rng(1))
    assert p.train_gen is False
    seq = [p.observe(c, [0.4], [0.6]) for c in range(1, 6)]
    assert seq == [True, False, True, False, True]
    # slow averages: pred_avg <- 0.99 pred_avg + 0.01 mean(pred on fakes); pred_avg_real from 1 - mean(pred on reals)
    q = P.GanPolicy(np.random.default_rng(1))
    q.observe(1, [0.2], [0.9])
    assert math.isclose(q.pred_avg, 0.99 * 0.5 + 0.01 * 0.2) and math.isclose(q.pred_avg_real, 0.99 * 0.5 + 0.01 * (1 - 0.9))
    assert q.avg_pred == 0.0 and q.avg_pred_real == 0.0


def test_trainee_switching_thresholds():
    """With a positive max_num_since_training_change the 0.3 / 0.7 thresholds decide (:1925-1939)."""
    p = P.GanPolicy(np.random.default_rng(2), max_num_since_training_change=3)
    assert p.observe(1, [0.1]) is True and p.num_since_change == 0      # discriminator too good: train the generator
    assert p.observe(2, [0.1]) is True and p.num_since_change == 1
    assert p.observe(3, [0.9]) is False and p.num_since_change == 0     # generator fooling it: train the discriminator
    assert p.observe(4, [0.5]) is True and p.num_since_change == 0      # in between: alternate
    p.num_since_change = 3
    assert p.observe(5, [0.1]) is False and p.num_since_change == 1     # the cap forces a switch
    # sums accumulate between switch points when trainee_switch_skip_n > 1
    r = P.GanPolicy(np.random.default_rng(3), effective_batch_size=2, trainee_switch_skip_n=2, max_num_since_training_change=9)
    assert r.observe(1, [0.1, 0.2]) is False and math.isclose(r.avg_pred, 0.3)
    r.observe(2, [0.1, 0.2])
    assert math.isclose(r.pred_avg, 0.99 * 0.5 + 0.01 * (0.6 / 4))
