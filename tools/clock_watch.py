"""Shader clock and socket power while a workload runs back to back (rocm-smi polled from a thread):
    python tools/clock_watch.py     -> fp32 / bf16x3 / bf16 GEMMs at the level-3 shape, the three training steps, a streaming copy"""
import os, sys, json, subprocess, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import functional as DF
dev = torch.device("cuda:0")

def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            d = json.loads(r.stdout)["card0"]
            sclk = [v for k, v in d.items() if "sclk" in k.lower()]
            pw = [v for k, v in d.items() if "power" in k.lower() and "W" in k]
            out.append((sclk[0] if sclk else "?", pw[0] if pw else "?"))
        except Exception as e:  # noqa: BLE001
            out.append(("err", str(e)[:60]))
        time.sleep(0.05)

def watch(name, fn, seconds=2.5):
    fn(); torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    dt = time.time() - t0
    stop.set(); th.join()
    tail = out[len(out) // 2:]   # second half: settled
    print(f"{name:34s} {dt / n * 1e6:9.1f} us/iter   sclk {[s for s, _ in tail][:6]}   power {[p for _, p in tail][:6]}", flush=True)

torch.manual_seed(0)
B, H, W, Ci, Co = 32, 32, 32, 1024, 512
x = torch.randn(B, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(Co, Ci, 1, 1, device=dev) / Ci ** 0.5
xb, wb = x.bfloat16(), w.bfloat16()
big = torch.empty(1 << 28, device=dev); big2 = torch.empty_like(big)
with torch.no_grad():
    watch("idle-ish: 1 GiB copy", lambda: big2.copy_(big))
    DF.set_gemm_precision("fp32")
    watch("fp32 MFMA NT 32768x512x1024", lambda: DF.conv_nobias(x, w))
    DF.set_gemm_precision("bf16x3")
    watch("bf16x3 NT 32768x512x1024", lambda: DF.conv_nobias(x, w))
    DF.set_gemm_precision("fp32")
# one level-3 NAFBlock forward + backward (B = 32, 512 channels, 32 x 32): fp32, bf16x3, bf16 storage
from basicsr.archs.nafnet_arch import NAFBlock
from dcpt_amd.keyed_init import fill_module_
blk = fill_module_(NAFBlock(512)).to(dev)
xx = torch.randn(32, 512, 32, 32, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
go = torch.randn(32, 512, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
def step32():
    blk(xx).backward(go)
xb = xx.detach().bfloat16().requires_grad_(True); gb = go.bfloat16()
def step16():
    DF.nafblock_bf16(xb, blk.fused_params()).backward(gb)
watch("level-3 NAFBlock fp32", step32)
DF.set_gemm_precision("bf16x3")
watch("level-3 NAFBlock bf16x3", step32)
DF.set_gemm_precision("fp32")
watch("level-3 NAFBlock bf16 storage", step16)
