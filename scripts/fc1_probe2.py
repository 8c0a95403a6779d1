"""First GPU process on a fresh box: the fc1 GEMM through the engine's own buffers, then with each operand replaced."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import ops, _lib as L
dev = torch.device('cuda:0')
R = 65536
model = B.build_model("bf16", dev)
eng = model.engine()
lens = [512] * (R // 512)
ws = eng.workspace(lens)
pw = eng.pw
for buf in (ws.xp, ws.s, ws.xn, ws.o, ws.u, ws.h, ws.mod):
    buf.normal_(0, 0.5)

def t(A, W, b, C, tag, epi=L.EPI_GELU):
    for _ in range(2):
        ops.gemm(A, W, b, out=C, epi=epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(A, W, b, out=C, epi=epi)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"{tag:40s} A {A.data_ptr():#x} W {W.data_ptr():#x} C {C.data_ptr():#x}: {us:7.1f} us = {2.0 * A.shape[0] * W.shape[0] * W.shape[1] / us / 1e6:6.0f} TF", flush=True)

W, b = pw.w["blks.0.mlp.fc1.weight"], pw.b("blks.0.mlp.fc1.bias")
t(ws.xn, W, b, ws.u, "engine buffers, block 0")
W5, b5 = pw.w["blks.5.mlp.fc1.weight"], pw.b("blks.5.mlp.fc1.bias")
t(ws.xn, W5, b5, ws.u, "engine buffers, block 5 weights")
t(ws.xn, W, b, ws.u, "bias epilogue instead of GELU", epi=L.EPI_BIAS)
t(ws.xn, W.clone(), b, ws.u, "W cloned")
t(ws.xn, W, b.clone(), ws.u, "bias cloned")
t(ws.xn.clone(), W, b, ws.u, "A cloned")
t(ws.xn, W, b, torch.empty_like(ws.u), "C fresh")
t(ws.s, W, b, ws.u, "A = ws.s")
t(ws.xn, W, b, ws.mod[:, :4096], "C = mod[:, :4096] (ldc 6144)")
t(ws.xn, W, b, ws.u, "engine buffers again")
import subprocess
print(subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout[-800:])
