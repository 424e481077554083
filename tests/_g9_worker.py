"""One rank of the world-size-2 check against G9 (tests/golden/g9_gather_w2.npz: the reference's Shuffle-BN collectives and
global enqueue captured on two gloo ranks): runs the build's `--shuffle_bn gather` path -- ContrastTrainer._shuffle_bn /
_shuffle_bn_attn + MoCo.forward -- on this rank's inputs and returns what the reference recorded for the same rank.

device 'cpu': host logic only (the C-ABI wrappers are replaced by torch-CPU stand-ins); device 'cuda': the HIP kernels
underneath, both ranks on ONE GPU, gloo carrying the collectives."""
import argparse
import os

import numpy as np
import torch
import torch.distributed as dist


def _sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def run_rank(rank, world, port, golden_path, device, out_path, prec="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if device == "cpu":
        from tests.test_dp_gloo import _install_cpu_standins
        _install_cpu_standins()
    from moma_amd.backbones.resnet_cifar import resnet8
    from moma_amd.MoMA.mem_moco import MoCo
    from moma_amd.MoMA.criterion_moco_att import CMO
    from moma_amd.learning.contrast_trainer import ContrastTrainer
    dev = torch.device(device)
    g = np.load(golden_path)
    B, K, d, steps = int(g["B"]), int(g["K"]), int(g["d"]), int(g["steps"])
    res = {}
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        r = f"{p}r{rank}_"
        attn = str(g[p + "attn"])
        opt = argparse.Namespace(head="mlp", s_dim=64, t_dim=64, feat_dim=d, attn=attn, mem="MoCo", nce_k=K, nce_t=0.15,
                                 local_rank=rank, node_rank=0, ngpus_per_node=world, rank=rank, world_size=world,
                                 shuffle_bn="gather", moma_prec=prec)
        mt = resnet8(num_classes=10)
        mt.load_state_dict(_sd(g, p + "t."))
        mt.train()
        kd = CMO(opt)
        kd.load_state_dict(_sd(g, p + "kd."))
        contrast = MoCo(d, K, 0.15, precision=prec)
        contrast.memory.copy_(torch.from_numpy(g[p + "memory0"]))
        mt, kd, contrast = mt.to(dev), kd.to(dev), contrast.to(dev)
        trainer = ContrastTrainer(opt)
        trainer.local_group = dist.new_group(list(range(world)))
        xs, qs = torch.from_numpy(g[r + "x"]).to(dev), torch.from_numpy(g[r + "q"]).to(dev)
        torch.manual_seed(int(g[p + "perm_seed_rank0"]) + rank)      # host randperm stream (rank 0's is broadcast)
        ks, aks, qouts, logits, idx, mems = [], [], [], [], [], []
        for t in range(steps):
            with torch.no_grad():
                if attn == "self":
                    k, all_k = trainer._shuffle_bn(xs[t], mt, kd.embed_t)
                    q = qs[t]
                else:
                    q, k, all_k = trainer._shuffle_bn_attn(xs[t], mt, kd.embed_t, kd, qs[t])
                lg, _ = contrast(q=q.float(), k=k.float(), all_k=all_k.float())
            ks.append(k.float().cpu().numpy().copy()); aks.append(all_k.float().cpu().numpy().copy()); qouts.append(q.float().cpu().numpy().copy())
            logits.append(lg.float().cpu().numpy().copy()); idx.append(int(contrast.index)); mems.append(contrast.memory.float().cpu().numpy().copy())
        res[r + "k"] = np.stack(ks); res[r + "all_k"] = np.stack(aks); res[r + "q_out"] = np.stack(qouts)
        res[r + "logits"] = np.stack(logits); res[r + "index"] = np.array(idx, dtype=np.int64); res[r + "memory"] = np.stack(mems)
        res[r + "t_after.bn1.running_mean"] = mt.state_dict()["bn1.running_mean"].float().cpu().numpy()
    np.savez(f"{out_path}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


def compare(golden_path, out_path, world, atol_k, atol_logits):
    g = np.load(golden_path)
    for rank in range(world):
        res = np.load(f"{out_path}.rank{rank}.npz")
        for key in res.files:
            ref, got = g[key], res[key]
            assert ref.shape == got.shape, key
            if key.endswith("index"):
                assert np.array_equal(ref, got), key                      # pointer: exact
            else:
                tol = atol_logits if key.endswith("logits") else atol_k
                np.testing.assert_allclose(got, ref, rtol=0, atol=tol * max(1.0, np.abs(ref).max()), err_msg=key)
    # the enqueue ORDER is the shuffled global order on every rank: the same rows in the same slots on both ranks
    r0, r1 = np.load(f"{out_path}.rank0.npz"), np.load(f"{out_path}.rank1.npz")
    for key in r0.files:
        if key.endswith("memory") or key.endswith("all_k"):
            assert np.array_equal(r0[key], r1[key.replace("r0_", "r1_")]), key
