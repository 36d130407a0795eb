"""Patience-based early stopping (interface of src/utils/early_stopping.py:9-88)."""
from __future__ import annotations


class EarlyStopping:
    def __init__(self, patience=20, min_delta=0.0, mode='min', verbose=True):
        self.patience = patience
        self.min_delta = min_delta
        self.mode = mode
        self.verbose = verbose
        self.reset()
        sign = -1.0 if mode == 'min' else 1.0
        # improvement must exceed min_delta in the good direction
        self.is_better = lambda new, best: sign * (new - best) > min_delta

    def reset(self):
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.best_epoch = 0

    def __call__(self, score, epoch) -> bool:
        """Record ``score`` of ``epoch``; returns True when it is the best so far."""
        if self.best_score is None or self.is_better(score, self.best_score):
            if self.verbose:
                if self.best_score is None:
                    print(f"early stopping: first score {score:.6f}")
                else:
                    print(f"early stopping: improved {self.best_score:.6f} -> {score:.6f}")
            first = self.best_score is None
            self.best_score, self.best_epoch = score, epoch
            if not first:
                self.counter = 0
            return True
        self.counter += 1
        if self.verbose:
            print(f"early stopping: no improvement ({score:.6f} vs best {self.best_score:.6f}), "
                  f"{self.counter}/{self.patience}")
        if self.counter >= self.patience:
            self.early_stop = True
            if self.verbose:
                print(f"early stopping: stop; best epoch {self.best_epoch}, best score {self.best_score:.6f}")
        return False
