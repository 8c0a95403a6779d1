"""Raw-file export of everything a C/C++ host needs to run the reverse loop through libstdiff_hip*.so without Python:
the packed MFMA operands and biases of a DiT_stDiff, the scheduler's step coefficients and the inputs of one
sample_stDiff call.  examples/c_host/sample_host.cpp reads it; tests/test_gpu_host_c.py compares the two results."""
import os

import numpy as np
import torch

from .engine import step_coefficients


def export_for_c_host(model, sched, x_cond, gt, mask, path, seed, chunk=512):
    """Writes <path>/manifest.txt ("name bytes" per line, scalars as "key = value") and <path>/<name>.bin."""
    os.makedirs(path, exist_ok=True)
    eng = model.engine()
    pw = eng.pw
    pw.ensure_fused()
    tensors = {}
    for n, t in pw.w.items():
        tensors[n] = t
    for n in pw.w:
        tensors[n[:-len(".weight")] + ".bias"] = pw.b(n[:-len(".weight")] + ".bias")
    if pw.fused is not None:
        for n, t in pw.fused.items():
            tensors["fused." + n] = t
    tensors["freqs"] = pw.freqs
    tensors["coef"] = step_coefficients(sched)
    tensors["x_cond"] = torch.as_tensor(np.asarray(x_cond), dtype=torch.float32)
    tensors["gt"] = torch.as_tensor(np.asarray(gt), dtype=torch.float32)
    tensors["mask"] = torch.as_tensor(np.asarray(mask), dtype=torch.float32)
    N = tensors["gt"].shape[0]
    lines = [f"G = {pw.G}", f"Gp = {pw.Gp}", f"H = {pw.H}", f"heads = {pw.heads}", f"depth = {pw.depth}",
             f"T = {len(sched)}", f"N = {N}", f"chunk = {chunk}", f"seed = {int(seed)}",
             f"dtype = {0 if pw.tdt == torch.float32 else 1}", f"h16 = {1 if pw.tdt == torch.float16 else 0}",
             f"fused = {0 if pw.fused is None else 1}"]
    for n, t in tensors.items():
        raw = t.detach().contiguous().cpu()
        if raw.dtype == torch.bfloat16:
            raw = raw.view(torch.int16)
        raw.numpy().tofile(os.path.join(path, n + ".bin"))
        lines.append(f"{n} {raw.numel() * raw.element_size()}")
    with open(os.path.join(path, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def export_train_for_c_host(engine, batches, path):
    """Training counterpart: the flat fp32 parameter buffer of a TrainEngine (fp32 compute) with the offset of every
    tensor, the noise-schedule tables, the hyper-parameters and `batches` = [(x, x_cond, noise, timesteps), ...];
    examples/c_host/train_host.cpp runs the iterations and writes the updated flat buffer."""
    os.makedirs(path, exist_ok=True)
    assert engine.tdt == torch.float32, "the C++ training host is written for the fp32 path"
    lines = [f"G = {engine.G}", f"Gp = {engine.Gp}", f"H = {engine.H}", f"heads = {engine.heads}",
             f"depth = {engine.depth}", f"T = {engine.T}", f"total = {engine.total}", f"steps = {len(batches)}",
             f"C = {batches[0][0].shape[0]}", f"lr_e12 = {int(round(engine.lr * 1e12))}"]
    for n in engine.names:
        lines.append(f"off.{n} = {engine.offs[n]}")
    tensors = {"flat_p": engine.flat_p, "mask": engine.mask, "sa": engine.sa, "sb": engine.sb, "freqs": engine.pw.freqs}
    for i, (x, xc, noise, t) in enumerate(batches):
        tensors[f"x{i}"], tensors[f"xc{i}"], tensors[f"noise{i}"], tensors[f"t{i}"] = x, xc, noise, t
    for n, t in tensors.items():
        raw = t.detach().contiguous().cpu()
        raw.numpy().tofile(os.path.join(path, n + ".bin"))
        lines.append(f"{n} {raw.numel() * raw.element_size()}")
    with open(os.path.join(path, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
