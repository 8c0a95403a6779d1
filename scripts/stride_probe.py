"""Does the row stride of the A operand set the K-loop pace?  The qkv-shaped GEMM (K = 1024, N = 3072, plain bias
epilogue) with A = a contiguous [R, 1024] buffer against A = the first 1024 columns of the [R, 4096] MLP buffer (8 KB row
stride, what mlp.fc2 reads): python scripts/stride_probe.py [R] [dtype]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import ops
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
model = B.build_model(dtype, dev)
eng = model.engine()
ws = eng.workspace([512] * (R // 512))
pw = eng.pw
for buf in (ws.xn, ws.u):
    buf.normal_(0, 0.5)
W, b = pw.w["blks.0.attn.qkv.weight"], pw.b("blks.0.attn.qkv.bias")
cases = {"lda 1024": ws.xn, "lda 4096 (cols 0:1024)": ws.u[:, :1024], "lda 4096 (cols 2048:3072)": ws.u[:, 2048:3072]}
t0 = time.time()
while time.time() - t0 < 2.0:
    for A in cases.values():
        ops.gemm(A, W, b, out=ws.qkv)
    torch.cuda.synchronize()
for name, A in cases.items():
    ts = []
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); ops.gemm(A, W, b, out=ws.qkv); e.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(e) * 1e3)
    us = sorted(ts)[2]
    print(f"{name:28s} {us:8.1f} us  {2.0 * R * 3072 * 1024 / us / 1e6:7.0f} TFLOP/s")
