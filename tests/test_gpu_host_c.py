"""GPU: the C ABI as a C caller sees it.  examples/c_host/sample_host_* is a C++ program (hipMalloc'ed buffers, the
entry points of include/stdiff_hip.h, no Python, no torch) that runs the whole sample_stDiff reverse loop from an
export of the model's packed operands; its result must equal the Python path's BIT FOR BIT (same library, same launch
sequence, same Philox stream).  The Python path itself is pinned by the oracle and the goldens in the other test files.

Starts child GPU processes; collected AFTER tests/test_gpu_folds.py (which used to leave the parent unable to start
any: DESIGN section 6a', fixed in round 3 by the per-device stream cache)."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype,H,heads", [("fp16", 256, 4), ("bf16", 128, 4), ("fp32", 128, 2), ("fp16", 512, 8)])
def test_c_host_reverse_loop_equals_python(dtype, H, heads, tmp_path):
    from stdiff_amd.export import export_for_c_host
    from stdiff_amd.model.sample import sample_stDiff
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.process.data import get_data_loader
    exe = os.path.join(ROOT, "examples", "c_host", "sample_host_f16" if dtype == "fp16" else "sample_host_bf16")
    if not os.path.isfile(exe):          # normally built by __graft_entry__.build(); the same image has hipcc
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True, capture_output=True, timeout=300)
    assert os.path.isfile(exe), "examples/c_host was not built (python -c 'import __graft_entry__ as g; g.build()')"
    N, G, depth, B, T, seed = 1200, 72, 2, 512, 12, 20261111
    p = O.make_params(G, H, depth, 33)
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype=dtype)
    m.load_state_dict(p)
    m = m.to(DEV).eval()
    mask = O.kfold_mask(G, 2)
    gt = ((O.synth_expression(N, G, 9) * mask) * 2 - 1).astype(np.float32)
    sched = NoiseScheduler(T, beta_schedule="cosine")
    export_for_c_host(m, sched, gt, gt, mask, str(tmp_path), seed, chunk=B)
    dl = get_data_loader(gt, gt, batch_size=B, is_shuffle=False)
    want = sample_stDiff(m, dl, sched, mask=mask, gt=gt, device=torch.device(DEV), num_step=T, sample_shape=(N, G),
                         is_condi=True, sample_intermediate=T, model_pred_type='noise', seed=seed, is_tqdm=False)
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    got = np.fromfile(os.path.join(str(tmp_path), "result.bin"), dtype=np.float32).reshape(N, G)
    known = mask.astype(bool)
    assert np.array_equal(got[:, known], gt[:, known])
    assert np.array_equal(got, want), float(np.abs(got - want).max())


@pytest.mark.parametrize("G", [72, 50])
def test_c_host_training_iterations_equal_python(G, tmp_path):
    """examples/c_host/train_host: three reference training iterations (add-noise + splice, stdiff_dit_train_forward,
    masked MSE, stdiff_dit_backward, gradient norm, clip + AdamW) driven from C++ on an export of the flat parameter
    buffer, against TrainEngine.step on the same batches, noise and timesteps (fp32 compute; G = 50 takes the
    transposed-copy path of odd gene counts).  Gradients are bit-identical by construction (same launches); loss and
    gradient norm are summed with float atomics, so the comparison allows their last bits where clipping applies."""
    from stdiff_amd.export import export_train_for_c_host
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.train_engine import TrainEngine
    exe = os.path.join(ROOT, "examples", "c_host", "train_host")
    if not os.path.isfile(exe):
        subprocess.run(["make", "-C", os.path.dirname(exe)], check=True, capture_output=True, timeout=300)
    H, depth, heads, T, C, steps = 128, 2, 4, 1000, 96, 3
    p = O.make_params(G, H, depth, 51)
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype="fp32")
    m.load_state_dict(p)
    m = m.to(DEV)
    mask = O.kfold_mask(G, 1)
    te = TrainEngine(m, lr=1e-4, diffusion_step=T, mask=mask)
    st = (O.synth_expression(steps * C, G, 8) * 2 - 1).astype(np.float32)
    gen = torch.Generator().manual_seed(G)
    batches = []
    for i in range(steps):
        x = torch.from_numpy(st[i * C:(i + 1) * C])
        batches.append((x, (x * torch.from_numpy(mask)).float(), torch.randn(C, G, generator=gen),
                        torch.randint(1, T, (C,), generator=gen)))
    export_train_for_c_host(te, batches, str(tmp_path))
    losses, norms = [], []
    for x, xc, noise, tt in batches:
        loss, gn = te.step(x.to(DEV), xc.to(DEV), noise=noise.to(DEV), timesteps=tt.to(DEV))
        losses.append(float(loss))
        norms.append(float(gn))
    want = te.flat_p.cpu().numpy()
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-1500:])
    got = np.fromfile(os.path.join(str(tmp_path), "result.bin"), dtype=np.float32)
    got_loss = np.fromfile(os.path.join(str(tmp_path), "loss.bin"), dtype=np.float32)
    assert got.shape == want.shape
    assert np.allclose(got_loss, losses, rtol=1e-5, atol=0)
    if max(norms) < 1.0:                        # no clipping: the update does not see the norm's last bits
        assert np.array_equal(got, want), float(np.abs(got - want).max())
    else:
        assert float(np.abs(got - want).max()) <= 1e-4 * float(np.abs(want).max())
    assert float(np.abs(want - np.fromfile(os.path.join(str(tmp_path), "flat_p.bin"), dtype=np.float32)).max()) > 1e-5
