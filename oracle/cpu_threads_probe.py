import sys, time, torch
sys.path.insert(0, '.')
from dcpt_amd.keyed_init import keyed_input, keyed_state_dict
from oracle import nafnet_oracle as O
CFG = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
P = {k: v.requires_grad_(True) for k, v in keyed_state_dict(O.nafnet_param_shapes(**CFG), seed=0).items()}
for B in (1, 4):
    x = keyed_input("x", (B, 3, 256, 256)); gt = keyed_input("gt", (B, 3, 256, 256))
    for nt in (16, 32, 64):
        torch.set_num_threads(nt)
        def step():
            for p in P.values(): p.grad = None
            y, _ = O.nafnet_forward(x, P); O.l1_loss(y, gt).backward()
        step(); t0 = time.perf_counter(); step(); dt = time.perf_counter() - t0
        print(f"B={B} threads={nt}: {dt:.2f} s/step -> {B*0.065536/dt:.4f} MP/s", flush=True)
