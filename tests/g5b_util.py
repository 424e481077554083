"""Rebuild the big tensors of the G5b step trace (tests/golden/g5b_step_trace_big.npz) from their seeds -- the fixture
stores seeds and checksums instead of a 65536 x d queue and 3 attention modules (tests/golden/make_golden.py:g5b)."""
import numpy as np
import torch

K_BIG = 65536


def sd(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def big_queue(g, p, d):
    gq = torch.Generator().manual_seed(int(g[p + "seeds"][0]))
    mem = torch.nn.functional.normalize(torch.randn(K_BIG, d, generator=gq))
    assert abs(mem.double().sum().item() - float(g[p + "memory0_sum"])) < 1e-6, "torch RNG stream changed"
    np.testing.assert_array_equal(mem[7].numpy(), g[p + "memory0_row7"])
    return mem


def fill_attention_(cmo, g, p):
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from the fixture's generator, in the order make_golden.py drew them."""
    gw = torch.Generator().manual_seed(int(g[p + "seeds"][1]))
    with torch.no_grad():
        for name in ("atts_q", "atts_k", "atts_queue"):
            att = getattr(cmo, name)
            for lin in (att.qkv, att.proj):
                b = 1.0 / np.sqrt(lin.in_features)
                lin.weight.copy_((torch.rand(lin.weight.shape, generator=gw) * 2 - 1) * b)
                lin.bias.copy_((torch.rand(lin.bias.shape, generator=gw) * 2 - 1) * b)
    assert abs(cmo.atts_q.qkv.weight.double().sum().item() - float(g[p + "attsq_qkv_w_sum"])) < 1e-6
    for k, v in sd(g, p + "kd.").items():                      # the (small) mlp heads are stored
        mod, rest = k.split(".", 1)
        getattr(cmo, mod).load_state_dict({rest: v}, strict=False)


def batches(g, p, steps=10, B=8):
    gen = torch.Generator().manual_seed(int(g[p + "data_seed"]))
    images = torch.randn(steps, B, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 100, (steps, B), generator=gen)
    if p + "images_sum" in g.files:
        assert abs(images.double().sum().item() - float(g[p + "images_sum"])) < 1e-5, "torch RNG stream changed"
    return images, labels
