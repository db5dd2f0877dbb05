"""Host-side training POLICY of the in-filling GAN's loop (misc_py/gan-infilling-100.py:1607-1647, :1724-1767, :1826-1939): which
network trains next, the learning-rate schedule, the random label flips and the `adapt` loss weights.  None of it is arithmetic on
images -- it is a few scalars per iteration computed from running averages of the discriminator's predictions -- so it stays on
the host, as in the reference; `gan_trainer.gan_iteration` takes what it decides (labels, adapts, train=) as inputs.

The reference draws from numpy's global generator; here the generator is passed in (``numpy.random.Generator`` or anything with
``.random()``), so a run is reproducible.  Every formula cites the line it restates."""
from __future__ import annotations

import math

# :123-126
TRAINEE_SWITCH_SKIP_N = 1
MAX_NUM_SINCE_TRAINING_CHANGE = 0


def learning_rates(counter: int):
    """(generator lr, discriminator lr) at iteration `counter`, or None once training is over (:1636-1647): 2e-4 up to 350 000
    iterations, then 2e-4 * (1 - step/8) with step = (counter - 350000) // 50000 + 1; stop beyond 700 000; discriminator = half."""
    if counter < 350000:
        rate = 0.0002
    else:
        if counter > 700000:
            return None
        step = (counter - 350000) // 50000 + 1
        rate = 0.0002 * (1 - step / 8)
    return rate, rate / 2


def batch_norm_on(counter: int) -> bool:
    """train_batch_norm_on (:1644): the generator's batch norms take batch statistics during its train op for the first 250 000 iterations."""
    return counter < 250000


class GanPolicy:
    """The loop's mutable scalars (:1610-1621) and the three decisions made from them."""

    def __init__(self, rng, effective_batch_size: int = 1, trainee_switch_skip_n: int = TRAINEE_SWITCH_SKIP_N,
                 max_num_since_training_change: int = MAX_NUM_SINCE_TRAINING_CHANGE):
        self.rng = rng
        self.ebs = effective_batch_size
        self.skip_n = trainee_switch_skip_n
        self.max_since = max_num_since_training_change
        self.train_gen = False          # :1611
        self.avg_pred = 0.0             # :1612 running SUM of the predictions on generated images since the last switch point
        self.avg_pred_real = 0.0        # :1613 ... on natural images
        self.num_since_change = 0.0     # :1614
        self.pred_avg = 0.5             # :1619 slow average (0.99 / 0.01) of the predictions on generated images
        self.pred_avg_real = 0.5        # :1620 ... of 1 - prediction on natural images

    # ---- :1733-1737
    def fake_label(self):
        """(label, adapt) for ONE generated image shown to the discriminator."""
        prob = 0.01 * (1.0 - self.pred_avg) ** 7
        no_flip = self.rng.random() > prob
        label = 1.0e-8 if no_flip else 0.9 + 0.1 * self.rng.random() - 1.0e-8
        adapt = 10 * math.exp(-self.pred_avg) * (1 - math.exp(-self.pred_avg ** 2)) if no_flip else 1.0
        return label, adapt

    # ---- :1772-1776
    def real_label(self):
        """(label, adapt) for ONE natural image."""
        prob = 0.01 * (1.0 - self.pred_avg_real) ** 7
        no_flip = self.rng.random() > prob
        label = 0.9 + 0.1 * self.rng.random() - 1.0e-8 if no_flip else 1.0e-8
        return label, 1.0

    def labels(self, n_fake: int, n_real: int):
        """Labels and adapt rates of one discriminator step, generated images first (the order of :1724-1795)."""
        fakes = [self.fake_label() for _ in range(n_fake)]
        reals = [self.real_label() for _ in range(n_real)]
        both = fakes + reals
        return [b[0] for b in both], [b[1] for b in both]

    # ---- :1826-1827, :1903-1939
    def observe(self, counter: int, preds_fake, preds_real=()):
        """Account for an iteration's discriminator predictions (on generated / natural images) and, at a switch point
        (`counter % trainee_switch_skip_n == 0`), update the slow averages and decide which network trains next.  Returns
        ``train_gen`` for the NEXT iteration."""
        self.avg_pred += float(sum(preds_fake))
        self.avg_pred_real += float(sum(preds_real))
        if counter % self.skip_n:
            return self.train_gen
        self.avg_pred /= self.skip_n * self.ebs
        self.pred_avg = 0.99 * self.pred_avg + 0.01 * self.avg_pred
        if self.avg_pred_real:
            self.avg_pred_real /= self.skip_n * self.ebs
            self.avg_pred_real = 1.0 - self.avg_pred_real
            self.pred_avg_real = 0.99 * self.pred_avg_real + 0.01 * self.avg_pred_real
        if self.num_since_change >= self.max_since:
            self.num_since_change = 1
            self.train_gen = not self.train_gen
        elif self.avg_pred < 0.3:
            self.num_since_change = self.num_since_change + 1 if self.train_gen else 0
            self.train_gen = True
        elif self.avg_pred > 0.7:
            self.num_since_change = self.num_since_change + 1 if not self.train_gen else 0
            self.train_gen = False
        else:
            self.num_since_change = 0
            self.train_gen = not self.train_gen
        self.avg_pred = 0.0
        self.avg_pred_real = 0.0
        return self.train_gen
