"""One training epoch of PotGNN (``ramannoodle/pmodel/torch/_train.py:24-95``)."""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import DataLoader

from ramannoodle_amd import parallel


def train_single_epoch(model, training_set, validation_set, batch_size: int, optimizer,
                       loss_function):
    """Shuffled mini-batches through forward / loss / backward / ``optimizer.step()``, then
    the validation set in batches of at most 100 in evaluation mode.

    Returns ``(mean training loss, mean validation loss, mean per-component variance of
    the validation predictions [6])``.  Forward and backward run on the device; the
    optimiser is whatever ``torch.optim`` object the caller built on ``model.parameters()``.
    """
    # the shuffling generator lives on torch's default device, as the reference's does (_train.py:51-58): under
    # torch.set_default_device("cuda") -- the reference's GPU recipe -- the sampler draws its permutation there and a
    # CPU generator is refused.  Every rank of a data-parallel run builds it the same way (same device type, same
    # default seed), so the ranks' mini-batches stay in step.
    default_device = torch.get_default_device()
    train_loader = DataLoader(training_set, batch_size=batch_size, shuffle=True,
                              generator=torch.Generator(device=default_device))
    validation_loader = DataLoader(validation_set, batch_size=min(100, len(validation_set)),
                                   shuffle=False)
    # data-parallel (``model.enable_data_parallel()``): every rank draws the same shuffled
    # mini-batches (same generator state), evaluates its contiguous block of each with the
    # BatchNorm statistics of the whole batch, and the gradients are averaged over the ranks,
    # so a step equals the single-process step on the full mini-batch (blocks weighted by size).
    group = getattr(model, "data_parallel_group", None)
    # The host side of a step (loss, optimiser on <= 1.2 MB of parameters) is tiny: torch's
    # intra-op thread pool only adds wake-up stalls there (measured on a 1-GPU MI355X box:
    # random 50-80 ms pauses per step with the default pool, none with one thread).
    host_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _run_epoch(model, train_loader, validation_loader, optimizer, loss_function, group, default_device)
    finally:
        torch.set_num_threads(host_threads)


def _run_epoch(model, train_loader, validation_loader, optimizer, loss_function, group, device):
    model.train()
    train_losses = []
    for batch in train_loader:
        weight = 1.0
        if group is not None:
            global_items = batch[0].shape[0]
            if global_items < torch.distributed.get_world_size(group):
                continue  # a block would be empty: every rank skips the same batch
            batch = parallel.batch_shard(tuple(batch), group)  # balanced blocks, none empty
            # blocks of a ragged batch differ by one structure: weight the rank's loss by its
            # share so that the averaged step is the full-batch step
            weight = parallel.rank_loss_weight(batch[0].shape[0], global_items, group)
        lattice, atomic_numbers, position, polarizability = batch
        # the batch goes to the default device (_train.py:65-69); the result lives where the batch lives
        out = model.forward(lattice.to(device), atomic_numbers.to(device), position.to(device))
        loss = loss_function(out, polarizability.to(out.device)) * weight
        loss.backward()
        if group is not None:
            parallel.average_gradients(model, group)
            train_losses.append(parallel.mean_over_ranks(float(loss.detach()), group))
        else:
            train_losses.append(float(loss.detach()))
        optimizer.step()
        optimizer.zero_grad()

    model.eval()
    validation_losses, validation_vars = [], []
    for lattice, atomic_numbers, position, polarizability in validation_loader:
        out = model.forward(lattice.to(device), atomic_numbers.to(device), position.to(device))
        validation_losses.append(float(loss_function(out, polarizability.to(out.device)).detach()))
        validation_vars.append(torch.var(out, dim=0).detach().cpu().numpy().copy())
    return (float(np.mean(train_losses)), float(np.mean(validation_losses)),
            np.mean(validation_vars, axis=0))
