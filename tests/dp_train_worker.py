"""One rank of the data-parallel training test (``test_gpu_parity.py``): evaluates its block of
the fixture batch with cross-rank BatchNorm statistics, averages the gradients over the ranks
and (rank 0) writes them out.  usage: dp_train_worker.py RANK WORLD PORT OUT.npz"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.conftest import load_golden
    from tests.helpers import product_model_from_golden
    from ramannoodle_amd import parallel

    g = load_golden("triclinic20_train")
    model = product_model_from_golden(g)
    model.enable_data_parallel()
    model.train()
    s = g["train/target"].shape[0]
    batch = (torch.tensor(g["lattice"], dtype=torch.float32).expand(s, 3, 3),
             torch.tensor(g["atomic_numbers"]).expand(s, -1),
             torch.tensor(g["pos_batch"][:s], dtype=torch.float32),
             torch.tensor(g["train/target"]))
    lat, zs, pos, target = parallel.batch_shard(batch)
    if len(sys.argv) > 5 and sys.argv[5] == "device":
        device_mode(model, lat, zs, pos, target, rank, world, out_path)
        return
    out = model.forward(lat, zs, pos)
    loss = torch.nn.MSELoss()(out, target)
    loss.backward()
    parallel.average_gradients(model)
    total = torch.tensor([float(loss.detach())], dtype=torch.float64)
    dist.all_reduce(total)
    gathered = [torch.zeros_like(out.detach()) for _ in range(world)]
    dist.all_gather(gathered, out.detach().contiguous())
    if rank == 0:
        sd = model.state_dict()
        arrays = {"grad/" + n: p.grad.numpy() for n, p in model.named_parameters()}
        arrays["loss"] = np.array(total.item() / world)
        arrays["out"] = torch.cat(gathered).numpy()
        arrays["running_mean"] = sd["_to_polarizability_embedding.1.running_mean"].numpy()
        arrays["running_var"] = sd["_to_polarizability_embedding.1.running_var"].numpy()
        np.savez(out_path, **arrays)
    dist.barrier()
    dist.destroy_process_group()


def device_mode(model, lat, zs, pos, target, rank, world, out_path):
    """Two steps of device-resident Adam; rank 0 writes both ranks' resulting state dicts."""
    from ramannoodle_amd import parallel
    from ramannoodle_amd.pmodel import DeviceAdam

    opt = DeviceAdam(model, lr=1e-3)
    for _ in range(2):
        out = model.forward(lat, zs, pos)
        torch.nn.MSELoss()(out, target).backward()
        parallel.average_gradients(model)
        opt.step()
    sd = model.state_dict()
    flat = torch.cat([v.reshape(-1).float() for v in sd.values() if v.is_floating_point()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        arrays, offset = {}, 0
        for key, value in sd.items():
            if value.is_floating_point():
                n = value.numel()
                arrays["sd/" + key] = gathered[0][offset:offset + n].reshape(value.shape).numpy()
                arrays["sd1/" + key] = gathered[1][offset:offset + n].reshape(value.shape).numpy()
                offset += n
        np.savez(out_path, **arrays)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
