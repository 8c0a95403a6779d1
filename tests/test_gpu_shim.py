"""GPU: the drop-in boundary.  The reference driver's own import lines (test-stDiff.py:11-14,20) resolve to
this build through the top-level `model/` and `process/` shim packages, and a tiny train -> sample runs through the
names they bind, with the reference's call signatures; only the device string differs from the driver."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

# verbatim from the reference driver (test-stDiff.py:11-14 and :20)
DRIVER_IMPORTS = """
from model.stDiff_model import DiT_stDiff
from model.stDiff_scheduler import NoiseScheduler
from model.stDiff_train import normal_train_stDiff
from model.sample import sample_stDiff
from process.data import *
"""


def test_reference_import_lines_and_tiny_flow():
    ns = {}
    exec(DRIVER_IMPORTS, ns)
    import stdiff_amd.model.sample
    import stdiff_amd.model.stDiff_model
    assert ns["DiT_stDiff"] is stdiff_amd.model.stDiff_model.DiT_stDiff
    assert ns["sample_stDiff"] is stdiff_amd.model.sample.sample_stDiff
    DiT_stDiff, NoiseScheduler = ns["DiT_stDiff"], ns["NoiseScheduler"]
    normal_train_stDiff, sample_stDiff, get_data_loader = ns["normal_train_stDiff"], ns["sample_stDiff"], ns["get_data_loader"]

    # the driver's per-fold flow (test-stDiff.py:92-179) on a tiny synthetic problem
    device = torch.device('cuda:0')
    gene_num, N, B, T = 40, 200, 64, 50
    mask = O.kfold_mask(gene_num, 0)
    seq = O.synth_expression(N, gene_num, 3)
    st = O.synth_expression(150, gene_num, 4)
    data_seq, data_seq_masked = seq * 2 - 1, (seq * mask) * 2 - 1
    data_spatial_masked = (st * mask) * 2 - 1
    dataloader = get_data_loader(data_seq, data_seq_masked, batch_size=B, is_shuffle=True)
    torch.manual_seed(1202)
    model = DiT_stDiff(input_size=gene_num, hidden_size=128, depth=2, num_heads=4, classes=6, mlp_ratio=4.0, dit_type='dit')
    model.to(device)
    assert normal_train_stDiff(model, dataloader=dataloader, lr=1e-3, num_epoch=5, device=device, pred_type='noise',
                               mask=mask, diffusion_step=T, is_tqdm=False) is None
    noise_scheduler = NoiseScheduler(num_timesteps=T, beta_schedule='cosine')
    dl = get_data_loader(data_spatial_masked, data_spatial_masked, batch_size=B, is_shuffle=False)
    imputation = sample_stDiff(model, device=device, dataloader=dl, noise_scheduler=noise_scheduler, mask=mask,
                               gt=data_spatial_masked, num_step=T, sample_shape=(150, gene_num), is_condi=True,
                               sample_intermediate=T, model_pred_type='noise', is_classifier_guidance=False, omega=0.2,
                               is_tqdm=False)
    assert isinstance(imputation, np.ndarray) and imputation.dtype == np.float32 and imputation.shape == (150, gene_num)
    assert np.isfinite(imputation).all()
    known = mask.astype(bool)
    assert np.array_equal(imputation[:, known], data_spatial_masked[:, known].astype(np.float32))


def test_classifier_guidance_is_identity():
    """Quirk Q1 (model/sample.py:78-86): the guided branch combines a conditional and an "unconditional" prediction
    of a model that ignores condi_flag, so (1 + w) e - w e = e; the build warns and returns the unguided result."""
    from stdiff_amd.model.sample import sample_stDiff
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.process.data import get_data_loader
    Nr, Gn, Hh, depth, heads, B, T = 90, 48, 128, 2, 4, 40, 12
    m = DiT_stDiff(Gn, Hh, depth, 'dit', heads, 6)
    m.load_state_dict(O.make_params(Gn, Hh, depth, 8))
    m = m.to("cuda:0")
    mask = O.kfold_mask(Gn, 1)
    gt = ((O.synth_expression(Nr, Gn, 2) * mask) * 2 - 1).astype(np.float32)
    kw = dict(mask=mask, gt=gt, device=torch.device("cuda:0"), num_step=T, sample_shape=(Nr, Gn), is_condi=True,
              sample_intermediate=T, model_pred_type='noise', seed=77, is_tqdm=False)
    dl = get_data_loader(gt, gt, batch_size=B, is_shuffle=False)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # the unguided run must not warn
        plain = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), is_classifier_guidance=False, **kw)
    with pytest.warns(UserWarning, match="is_classifier_guidance"):
        guided = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), is_classifier_guidance=True, omega=0.3, **kw)
    assert np.array_equal(plain, guided)
    # and two calls without a seed draw different noise (the reference's torch.randn advances the global generator)
    kw.pop("seed")
    a = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), **kw)
    b = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), **kw)
    assert not np.array_equal(a, b)
    torch.manual_seed(5)
    c = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), **kw)
    torch.manual_seed(5)
    d = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule='cosine'), **kw)
    assert np.array_equal(c, d)
