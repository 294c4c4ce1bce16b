"""Worker of tests/test_ddp_rccl_gpu.py: one process per GPU over RCCL (backend "nccl"), launched by torch.distributed.run.
  mode `step`   : the reference's DDP wiring (trainer.py:196-210: G and D wrapped separately, find_unused_parameters, buffer
                  broadcast) around one full G+D step on the HIP kernels; rank 0 writes whether all replicas are bit-identical
  mode `syncbn` : sync_bn=True generator forward/backward on this rank's shard of a fixed batch; rank 0 writes the gathered
                  images, its running statistics and the DDP-averaged gradients
usage: python -m torch.distributed.run --nproc-per-node N tests/ddp_rccl_worker.py <mode> <out.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP

from oracle import synth

CFG = dict(dim_z=24, shared_dim=16, shared_dim_feat=32, G_shared=True, G_shared_feat=True, hier=True, n_classes=10,
           SN_eps=1e-6, BN_eps=1e-5, adam_eps=1e-6, G_ch=32, D_ch=32, G_attn="16", D_attn="16", resolution=32,
           class_cond=True, instance_cond=True, toggle_grads=True, num_D_steps=1, num_D_accumulations=1,
           num_G_accumulations=1, split_D=False, DiffAugment="", DA=False, D_ortho=0.0, G_ortho=0.0, ema=True,
           ema_decay=0.9, ema_start=0)
B_TOTAL = 8


def models(cfg, dev):
    import ic_gan_amd.BigGAN as M
    G = M.Generator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    D = M.Discriminator(**{**cfg, "skip_init": True, "embedded_optimizers": False})
    G.load_state_dict(synth.synth_state(synth.spec_of(G.state_dict()), 11))
    D.load_state_dict(synth.synth_state(synth.spec_of(D.state_dict()), 22))
    return M, G.to(dev), D.to(dev)


def syncbn_inputs(cfg, dim_z):
    z, lab, fg = synth.CondSampler(cfg, dim_z, B_TOTAL, 9)()
    wts = torch.from_numpy(np.random.RandomState(3).standard_normal((B_TOTAL, 3, 32, 32))).float()
    return z, lab, fg, wts


def main():
    mode, out_path = sys.argv[1], sys.argv[2]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("ICG_TEST_BACKEND", "nccl")
    if os.environ.get("ICG_TEST_SHARED_GPU") == "1":        # both ranks on GPU 0 (1-GPU box): RCCL refuses that, gloo moves the
        local = 0                                            # CUDA tensors through the host -- same kernels, same DDP / SyncBN code
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    dist.init_process_group(backend)
    if mode == "step":
        from ic_gan_amd import train_fns, utils
        from ic_gan_amd.optim import FusedAdam
        M, G, D = models(CFG, dev)
        if rank == 1:                   # the per-forward buffer broadcast must overwrite this
            with torch.no_grad():
                G.linear.u0.add_(1.0)
        G_ema = M.Generator(**{**CFG, "skip_init": True, "no_optim": True}).to(dev)
        ema = utils.ema(G, G_ema, 0.9, 0)
        opt_d = FusedAdam(D.parameters(), lr=2e-3, betas=(0.0, 0.999), eps=1e-6)
        opt_g = FusedAdam(G.parameters(), lr=1e-3, betas=(0.0, 0.999), eps=1e-6)
        Gd = DDP(G, device_ids=[local], output_device=local, find_unused_parameters=True)
        Dd = DDP(D, device_ids=[local], output_device=local, find_unused_parameters=True)
        GD = M.G_D(Gd, Dd, optimizer_G=opt_g, optimizer_D=opt_d)
        gb = 4
        train = train_fns.GAN_training_function(Gd, Dd, GD, ema, {"itr": 1}, CFG, synth.CondSampler(CFG, G.dim_z, gb, 50 + rank),
                                                embedded_optimizers=False, device=dev, batch_size=gb)
        x, y, f = synth.synth_batch(CFG, gb, seed=70 + rank)
        Gd.train(); Dd.train()
        losses = [train(x.to(dev), y.to(dev), f.to(dev)) for _ in range(2)]
        flat = torch.cat([p.detach().reshape(-1) for p in list(G.parameters()) + list(D.parameters())])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        if rank == 0:
            torch.save({"identical": all(bool(torch.equal(gathered[0], g)) for g in gathered[1:]),
                        "finite": bool(torch.isfinite(flat).all()), "losses": losses, "world": world}, out_path)
    elif mode == "syncbn":
        cfg = dict(CFG, sync_bn=True)
        _, G, _ = models(cfg, dev)
        Gd = DDP(G, device_ids=[local], output_device=local)
        z, lab, fg, wts = syncbn_inputs(cfg, G.dim_z)
        per = B_TOTAL // world
        sl = slice(rank * per, (rank + 1) * per)
        Gd.train()
        img = Gd(z[sl].to(dev), lab[sl].to(dev), fg[sl].to(dev))
        (img * wts[sl].to(dev)).sum().backward()
        grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()])
        src = img.detach().contiguous()                  # the generator returns channels-last strides; collectives want dense NCHW
        imgs = [torch.empty(tuple(src.shape), device=dev, dtype=src.dtype) for _ in range(world)]
        dist.all_gather(imgs, src)
        if rank == 0:
            torch.save({"img": torch.cat(imgs, 0).cpu(), "grads": grads.cpu(), "world": world,
                        "rm": G.blocks[0][0].bn1.stored_mean.cpu(), "rv": G.output_layer[0].stored_var.cpu()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
