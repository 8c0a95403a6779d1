"""Attention forward alone at slab size: python scripts/attn_bench.py [R] [dtype bf16|fp16] [chunk rows, default 512]"""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stdiff_amd import ops
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
tdt = torch.float16 if (sys.argv[2] if len(sys.argv) > 2 else "fp16") == "fp16" else torch.bfloat16
H, heads = 1024, 16
CH = int(sys.argv[3]) if len(sys.argv) > 3 else 512
lens = [min(CH, R - i) for i in range(0, R, CH)]
cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device='cuda')
qkv = torch.randn(R, 3 * H, device='cuda').to(tdt)
out = torch.empty(R, H, device='cuda', dtype=tdt)
ops.attention_fwd(qkv, cu, CH, heads, out=out)
q, k, v = [qkv[:CH, i * H:(i + 1) * H].float().view(CH, heads, 64).permute(1, 0, 2) for i in range(3)]
ref = torch.softmax(q @ k.transpose(1, 2) * 64 ** -0.5, -1) @ v
err = float((out[:CH].float().view(CH, heads, 64).permute(1, 0, 2) - ref).abs().max() / ref.abs().max())
t0 = time.time()
while time.time() - t0 < 1.5:
    ops.attention_fwd(qkv, cu, CH, heads, out=out)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention_fwd(qkv, cu, CH, heads, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"attn fwd {tdt} R={R}: {us:.1f} us/call  {4.0 * CH * H * R / us / 1e6:.0f} TFLOP/s  chunk {CH}  relerr {err:.2e}  lib {os.environ.get('STDIFF_LIB_F16', 'default')}")
