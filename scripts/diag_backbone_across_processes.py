"""Which stock component gives different bits in different PROCESSES under torch.backends.cudnn.deterministic?  One process: seeded
resnet8x4 forward + backward (convolutions, BatchNorm, pooling, the classifier), and separately a stack of Linear layers, each
hashed; the driver compares N processes.     usage: python scripts/diag_backbone_across_processes.py [N]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False
    from moma_amd.backbones import model_dict
    def digest(ts):
        h = hashlib.sha256()
        for t in ts:
            h.update(t.detach().float().cpu().numpy().tobytes())
        return h.hexdigest()[:16]
    torch.manual_seed(5)
    net = model_dict["resnet8x4"](num_classes=2).cuda().train()
    x = torch.randn(32, 3, 32, 32, device="cuda")
    feats, logit = net(x, is_feat=True)
    out_fwd = digest([logit] + list(feats))
    logit.square().sum().backward()
    conv_w = [p.grad for n, p in net.named_parameters() if p.dim() == 4]
    bn_w = [p.grad for n, p in net.named_parameters() if p.dim() == 1]
    fc_w = [p.grad for n, p in net.named_parameters() if p.dim() == 2]
    torch.manual_seed(6)
    mlp = torch.nn.Sequential(torch.nn.Linear(256, 512), torch.nn.ReLU(), torch.nn.Linear(512, 128)).cuda()
    z = torch.randn(32, 256, device="cuda", requires_grad=True)
    o = mlp(z)
    o.square().sum().backward()
    print("RESULT", out_fwd, digest(conv_w), digest(bn_w), digest(fc_w), digest([o, z.grad] + [p.grad for p in mlp.parameters()]))
    sys.exit(0)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = []
for r in range(N):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    if p.returncode != 0 or not line:
        print("FAILED", p.stderr[-1500:])
        sys.exit(1)
    rows.append(line[0].split()[1:])
for name, col in zip(("backbone forward", "conv weight gradients", "BatchNorm / bias gradients", "classifier gradient", "Linear stack fwd + bwd"), zip(*rows)):
    print(f"{name}: {len(set(col))} distinct over {N} processes   {' '.join(col)}")
