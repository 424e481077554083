"""Host-side logic that needs no GPU: the epoch learning-rate schedule against the reference's values, the projection
heads of CMO (all four kinds, reference state-dict keys), macro-F1, the bounded EMA-table cache."""
import argparse
import math

import numpy as np
import pytest
import torch
import torch.nn as nn


def _opt(**kw):
    base = dict(learning_rate=0.05, lr_decay_epochs=[30, 40, 60], lr_decay_rate=0.1, cosine=False, epochs=60)
    base.update(kw)
    return argparse.Namespace(**base)


def _ref_lr(epoch, opt):
    """reference helper/util.py:37-50 restated (values, not code)"""
    lr = opt.learning_rate
    if opt.cosine:
        eta_min = lr * opt.lr_decay_rate ** 3
        return eta_min + (lr - eta_min) * (1 + math.cos(math.pi * epoch / opt.epochs)) / 2
    steps = sum(epoch > e for e in opt.lr_decay_epochs)
    return lr * opt.lr_decay_rate ** steps if steps else lr


@pytest.mark.parametrize("cosine", [False, True])
def test_adjust_learning_rate_matches_reference_schedule(cosine):
    from moma_amd.helper.util import adjust_learning_rate
    opt = _opt(cosine=cosine)
    m = nn.Linear(2, 2)
    optim = torch.optim.SGD([{"params": [m.weight]}, {"params": [m.bias], "lr": 0.7}], lr=opt.learning_rate)
    for epoch in [1, 2, 15, 30, 31, 40, 41, 59, 60]:
        lr = adjust_learning_rate(epoch, opt, optim)
        assert lr == pytest.approx(_ref_lr(epoch, opt), rel=1e-12)
        assert all(g["lr"] == lr for g in optim.param_groups)          # EVERY group is rewritten each epoch
    if cosine:                                                          # known answers: start, middle, end of the cosine
        assert adjust_learning_rate(0, opt, optim) == pytest.approx(0.05)
        assert adjust_learning_rate(30, opt, optim) == pytest.approx((0.05 + 0.05e-3) / 2)
        assert adjust_learning_rate(60, opt, optim) == pytest.approx(0.05e-3)


@pytest.mark.parametrize("head,keys", [
    ("None", []),
    ("linear", ["1.weight", "1.bias"]),
    ("mlp", ["1.weight", "1.bias", "3.weight", "3.bias"]),
    ("mlp_byol", ["1.weight", "1.bias", "2.weight", "2.bias", "2.running_mean", "2.running_var", "2.num_batches_tracked",
                  "4.weight", "4.bias"]),
])
def test_cmo_heads_all_kinds(head, keys):
    """reference MoMA/criterion_moco_att.py:254-305: module indices (state-dict keys), output shape, unit L2 norm."""
    from moma_amd.MoMA.criterion_moco_att import CMO
    opt = argparse.Namespace(head=head, s_dim=48, t_dim=40, feat_dim=48 if head == "None" else 16, attn="self")
    cmo = CMO(opt)
    assert sorted(k for k in cmo.embed_s.state_dict()) == sorted(keys)
    x = torch.randn(6, 48, 1, 1)
    y = cmo.embed_s(x)
    assert y.shape == (6, opt.feat_dim)
    torch.testing.assert_close(y.norm(dim=1), torch.ones(6), rtol=1e-5, atol=1e-5)
    assert cmo.embed_t(torch.randn(6, 40, 1, 1)).shape == (6, 40 if head == "None" else 16)
    assert {n.split(".")[0] for n, _ in cmo.named_parameters() if n.startswith("atts")} == {"atts_q", "atts_k", "atts_queue"}


def test_macro_f1_known_answer():
    from moma_amd.helper.loops_moma import macro_f1
    cm = np.array([[5, 1, 0], [2, 3, 0], [0, 0, 0]])                    # class 2 never predicted right -> counts 0
    f0 = 2 * (5 / 7) * (5 / 6) / ((5 / 7) + (5 / 6))
    f1 = 2 * (3 / 4) * (3 / 5) / ((3 / 4) + (3 / 5))
    assert macro_f1(cm) == pytest.approx((f0 + f1) / 3)


def test_graphed_inference_contract_on_cpu_is_eager():
    """On CPU tensors (or with grad enabled) GraphedInference is an eager call under the SAME return contract as a replay:
    ([pooled feature], logits); full_feats=True hands out the whole feature list."""
    from moma_amd.helper.graphs import GraphedInference
    from moma_amd.backbones.resnet_cifar import resnet8
    m = resnet8(num_classes=10).eval()
    g = GraphedInference(m)
    x = torch.randn(2, 3, 32, 32)
    with torch.no_grad():
        feats, logits = g(x, is_feat=True)
        ref_feats, ref_logits = m(x, is_feat=True)
        all_feats, _ = g(x, is_feat=True, full_feats=True)
    assert len(feats) == 1 and torch.equal(feats[0], ref_feats[-1]) and torch.equal(logits, ref_logits)
    assert len(all_feats) == len(ref_feats)


def test_hip_env_switch_is_set_at_import_and_respects_the_caller():
    """moma_amd/hip_env.py: importing the package sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (before the process's first HIP call); a
    value the caller exported is left alone and configure() reports that the switch is not in place."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import os, moma_amd; from moma_amd import hip_env; print(os.environ[hip_env.SWITCH], hip_env.configure())"
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "True"]
    env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = "1"
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "False"]


def test_loop_refuses_the_dual_queue_memories_up_front():
    """--mem MoCoST / MoCoSSTT: the reference's loop passes (q, k, all_k) to every memory (helper/loops_moma.py:331) and dies in its
    first step on these two (no k_t is ever computed).  Here the step object says so when it is built, before any compute."""
    import argparse
    import pytest
    import torch
    from moma_amd.helper.loops_moma import MomaStep

    class Dual:
        memory_s = None
    with pytest.raises(NotImplementedError, match="k_t"):
        MomaStep([None, None], [None, None, None], None, Dual(), None, argparse.Namespace(distill="moma"), torch.device("cpu"))
    MomaStep([None, None], [None, None, None], None, Dual(), None, argparse.Namespace(distill="kd"), torch.device("cpu"))   # (kd: no memory used)
