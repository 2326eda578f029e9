"""Warm-up + cosine learning-rate schedule stepped once per iteration (reference utils/cap_scheduler.py:4-81).

lr(step): for the first epoch a linear ramp from warmup_init_lr + 0.1*(init-warmup_init) to init_lr, afterwards
max(min_lr, min_lr + (init-min)*(1+cos(pi*step/total))/2).  `step()` increments first, then returns the new LR and
writes it into every param group of the optimizer it was given -- same observable sequence as the reference,
including Q8 (train_xe calls step() once before the loop of every epoch as well).
"""
import math


class CosineLRScheduler(object):
    _FIELDS = ('init_lr', 'warmup_init_lr', 'min_lr', 'num_epochs', 'num_its_per_epoch', 'warmup_factor',
               'warmup_epochs', 'global_steps')

    def __init__(self, optimizer, num_epochs, num_its_per_epoch, init_lr=5e-4, min_lr=1e-4, warmup_init_lr=1e-5,
                 warmup_factor=0.1, warmup_epochs=1, **kwargs):
        self.optimizer = optimizer
        self.init_lr, self.min_lr, self.warmup_init_lr = init_lr, min_lr, warmup_init_lr
        self.num_epochs, self.num_its_per_epoch = num_epochs, num_its_per_epoch
        self.warmup_factor, self.warmup_epochs = warmup_factor, warmup_epochs
        self.global_steps = 0

    def lr_at(self, step):
        if step // self.num_its_per_epoch < 1:
            alpha = (float(step) / self.num_its_per_epoch) / self.warmup_epochs
            span = self.init_lr - self.warmup_init_lr
            return span * (self.warmup_factor * (1.0 - alpha) + alpha) + self.warmup_init_lr
        total = self.num_epochs * self.num_its_per_epoch
        cos = (self.init_lr - self.min_lr) * (1 + math.cos(math.pi * step / total)) / 2 + self.min_lr
        return max(self.min_lr, cos)

    def step(self):
        self.global_steps += 1
        lr = self.lr_at(self.global_steps)
        self.update(lr)
        return lr

    def update(self, lr):
        for group in self.optimizer.param_groups:
            group['lr'] = lr

    def state_dict(self):
        return {k: getattr(self, k) for k in self._FIELDS}

    def load_state_dict(self, state_dict):
        for k in self._FIELDS:
            setattr(self, k, state_dict[k])
