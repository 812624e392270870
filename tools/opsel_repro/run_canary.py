import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from basicsr.archs import build_network
from dcpt_amd.keyed_init import fill_module_
from dcpt_amd import functional as DF
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcanary.so"))
lib.canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
FULL = dict(img_channel=3, width=64, middle_blk_num=1, enc_blk_nums=[1, 1, 1, 28], dec_blk_nums=[1, 1, 1, 1])
net = fill_module_(build_network(dict(type="NAFNetBaseline", act_dtype="bf16", **FULL))).cuda().eval()
d_in = (torch.rand((4, 1024, 34, 34), device="cuda") - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
u_in = [(torch.rand((4, 128, 272, 272), device="cuda") - 0.5).bfloat16().contiguous(memory_format=torch.channels_last),
        (torch.rand((4, 64, 544, 544), device="cuda") - 0.5).bfloat16().contiguous(memory_format=torch.channels_last)]
a32 = torch.rand(4096, 4096, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def disturbers():
    return {"none": lambda: None,
            "bf16 middle block": lambda: net.middle_blks[0](d_in),
            "bf16 up3 (128-tile kernel)": lambda: DF.up_ps(u_in[0], net.ups[3][0].weight, u_in[1]),
            "torch fp32 matmul": lambda: a32 @ a32}
for name, fn in disturbers().items():
    out = torch.zeros(1 + 4 * 64, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.no_grad():
        with torch.cuda.stream(sB):
            for _ in range(40): fn()
        with torch.cuda.stream(sA):
            for _ in range(30):
                assert lib.canary_launch(out.data_ptr(), 2048, 200, sA.cuda_stream) == 0
        with torch.cuda.stream(sB):
            for _ in range(40): fn()
    torch.cuda.synchronize()
    o = out.cpu().tolist()
    print(f"{name:30s} corrupted (lane, register) pairs: {o[0]}", [(o[1 + 4 * k], o[2 + 4 * k], o[3 + 4 * k], hex(o[4 + 4 * k] & 0xffffffff)) for k in range(min(o[0], 6))], flush=True)
