"""CPU restatement of the reference's epoch driver and loss bookkeeping (TEST INFRASTRUCTURE).

trainer / train_epoch / validate <- train.py:72-92, :95-122, :125-156;  RunningLoss.log <- models/modules/loss.py:270-293.
Quirk kept: ONE interval list collects both training and validation losses and is only cleared by log(), so training
steps left over at the end of an epoch (iter % report != 0) are averaged into the next *validation* entry."""
from . import step as ostep


class RunLog:
    def __init__(self):
        self.train, self.valid, self.intv, self.lr = [], [], [], []
        self.avg_dice, self.best_dice, self.is_best = 1.0, 1.0, False

    def log(self, iteration, training):
        if self.intv:
            avg = tuple(sum(v) / len(self.intv) for v in zip(*self.intv))
            self.intv = []
            if training:
                self.train.append((iteration,) + avg)
            else:
                self.valid.append((iteration,) + avg)
                self.avg_dice = avg[1]
                self.is_best = self.avg_dice < self.best_dice
                if self.is_best:
                    self.best_dice = self.avg_dice


def run_training(sd, cfg, train_batches, valid_batches, n_epochs, report=20, gamma=0.9):
    """Returns the RunLog after n_epochs of the reference cadence (epoch 0 starts with a validation pass)."""
    opt = ostep.make_optimizer(sd, cfg)
    log = RunLog()
    it = 0
    best_events = []

    def validate():
        for x, y in valid_batches:
            _, ce, dice, fl = ostep.eval_step(sd, cfg, x, y)
            log.intv.append((ce, dice, fl))
        log.log(it, False)
        best_events.append(log.is_best)              # Model.save(): checkpoint always, best-model copy iff is_best

    for epoch in range(n_epochs):
        log.lr.append((it, opt.param_groups[0]['lr']))
        if epoch == 0:
            validate()
        for x, y in train_batches:
            o = ostep.train_step(sd, opt, cfg, x, y)
            log.intv.append(tuple(o[:3]))
            if it % report == 0:
                log.log(it, True)
            log.lr.append((it, opt.param_groups[0]['lr']))
            it += 1
        validate()
        for g in opt.param_groups:                   # StepLR(step_size=1, gamma) once per epoch (train.py:91)
            g['lr'] *= gamma
    log.best_events = best_events
    return log
