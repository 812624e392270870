"""Training trajectories of NAFNet-64 [1,1,1,28] in fp32 and in bf16 storage on the same synthetic denoising task (same data, same
initial weights, AdamW): the loss curves should coincide up to bf16 noise.  python tools/traj_bf16.py [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from basicsr.archs import build_network
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def batch(i, B=8, S=128):
    g = torch.Generator(device=dev).manual_seed(1000 + i)
    base = torch.rand((B, 3, S // 8, S // 8), generator=g, device=dev)
    gt = torch.nn.functional.interpolate(base, size=(S, S), mode="bilinear", align_corners=False)
    lq = (gt + 25.0 / 255.0 * torch.randn((B, 3, S, S), generator=g, device=dev)).clamp(0, 1)
    return lq, gt


curves = {}
for dt in ("fp32", "bf16"):
    torch.manual_seed(0)
    net = build_network(dict(type="NAFNetBaseline", act_dtype=dt, **CFG)).to(dev)   # the reference's default init (beta = gamma = 0)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, betas=(0.9, 0.9), weight_decay=0.0, fused=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, iters, eta_min=1e-6)
    losses = []
    for i in range(iters):
        lq, gt = batch(i)
        opt.zero_grad(set_to_none=True)
        loss = (net(lq) - gt).abs().mean()
        loss.backward()
        opt.step(); sched.step()
        losses.append(float(loss))
    with torch.no_grad():
        lq, gt = batch(10 ** 6)
        out = net(lq).clamp(0, 1)
        psnr = float(-10 * torch.log10(((out - gt) ** 2).mean()))
        psnr_in = float(-10 * torch.log10(((lq - gt) ** 2).mean()))
    curves[dt] = (losses, psnr)
    print(f"{dt}: held-out PSNR {psnr:.2f} dB (noisy input {psnr_in:.2f} dB)")
print("iter   L1 fp32    L1 bf16    ratio")
for i in list(range(0, iters, max(1, iters // 15))) + [iters - 1]:
    a = sum(curves["fp32"][0][max(0, i - 4):i + 1]) / len(curves["fp32"][0][max(0, i - 4):i + 1])
    b = sum(curves["bf16"][0][max(0, i - 4):i + 1]) / len(curves["bf16"][0][max(0, i - 4):i + 1])
    print(f"{i:4d}   {a:.5f}    {b:.5f}    {b / a:.3f}")
