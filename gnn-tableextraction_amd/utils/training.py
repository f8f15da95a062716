"""EarlyStopping with the reference's semantics (src/utils/training.py:14-49): keep the best
validation loss; save the weights whenever the loss does not get worse (ties included); count
epochs that got worse; once the counter reaches ``patience`` save once more and stop."""
import datetime
import os

import torch


class EarlyStopping:
    def __init__(self, weights, name, patience=50):
        self.patience = patience
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.weights_path = weights
        self.dt = datetime.datetime.now().strftime("%d%m_%H%M")
        self.name = name

    def step(self, loss, model):
        if self.best_score is None:
            self.save_checkpoint(model)
            self.best_score = loss
        elif loss > self.best_score:
            self.counter += 1
            if self.counter >= self.patience:
                self.save_checkpoint(model)
                self.early_stop = True
        else:
            self.save_checkpoint(model)
            self.best_score = loss
            self.counter = 0
        return self.early_stop, self.counter

    def save_checkpoint(self, model):
        """WEIGHTS/{name}.pt = model.state_dict() (what model_predict.py:120 loads)."""
        os.makedirs(str(self.weights_path), exist_ok=True)
        state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        torch.save(state, f'{self.weights_path}/{self.name}.pt')
