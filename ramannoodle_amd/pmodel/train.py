"""One training epoch of PotGNN (``ramannoodle/pmodel/torch/_train.py:24-95``)."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader


def train_single_epoch(model, training_set, validation_set, batch_size: int, optimizer,
                       loss_function):
    """Shuffled mini-batches through forward / loss / backward / ``optimizer.step()``, then
    the validation set in batches of at most 100 in evaluation mode.

    Returns ``(mean training loss, mean validation loss, mean per-component variance of
    the validation predictions [6])``.  Forward and backward run on the device; the
    optimiser is whatever ``torch.optim`` object the caller built on ``model.parameters()``.
    """
    train_loader = DataLoader(training_set, batch_size=batch_size, shuffle=True,
                              generator=torch.Generator())
    validation_loader = DataLoader(validation_set, batch_size=min(100, len(validation_set)),
                                   shuffle=False)
    model.train()
    train_losses = []
    for lattice, atomic_numbers, position, polarizability in train_loader:
        out = model.forward(lattice, atomic_numbers, position)
        loss = loss_function(out, polarizability)
        train_losses.append(float(loss))
        loss.backward()
        optimizer.step()
        optimizer.zero_grad()

    model.eval()
    validation_losses, validation_vars = [], []
    for lattice, atomic_numbers, position, polarizability in validation_loader:
        out = model.forward(lattice, atomic_numbers, position)
        validation_losses.append(float(loss_function(out, polarizability)))
        validation_vars.append(torch.var(out, dim=0).detach().cpu().numpy().copy())
    return (float(np.mean(train_losses)), float(np.mean(validation_losses)),
            np.mean(validation_vars, axis=0))
