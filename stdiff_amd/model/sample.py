"""sample_stDiff / model_sample_stDiff with the reference signatures (model/sample.py:6-101), running the
masked-DDPM reverse loop as a replayed hipGraph of fused gfx950 kernels (stdiff_amd/engine.py::ReverseSampler)."""
import warnings

import numpy as np
import torch
from tqdm import tqdm

from .. import engine as E


def _collect_condition(dataloader, device):
    """One pass over the DataLoader: the condition rows and the chunk (mini-batch) lengths, in loader order.
    The reference re-reads and re-uploads the same batches every step (sample.py:9-10); they never change."""
    from ..process.data import DeviceLoader
    if isinstance(dataloader, DeviceLoader) and not dataloader.is_shuffle and dataloader.world == 1 \
            and dataloader.x_cond.device == torch.empty(0, device=device).device:
        return dataloader.x_cond, list(dataloader.lens)            # already resident, already in chunk order
    conds, lens = [], []
    for _, x_cond in dataloader:
        conds.append(x_cond.float().to(device))
        lens.append(int(x_cond.shape[0]))
    return torch.cat(conds, dim=0).contiguous(), lens


def model_sample_stDiff(model, device, dataloader, total_sample, time, is_condi, condi_flag):
    """Chunked denoiser call (sample.py:6-20): eps for every DataLoader batch of `total_sample`, concatenated."""
    if not is_condi:
        raise ValueError("is_condi=False passes y=None to cond_layer, which the reference cannot run either")
    x_cond, lens = _collect_condition(dataloader, device)
    eng = model.engine()
    t = torch.full((sum(lens),), int(time), dtype=torch.int64, device=eng.pw.device)
    return eng.forward(total_sample.float().to(eng.pw.device).contiguous(), t, x_cond, chunk_lens=lens)


def sample_stDiff(model,
                  dataloader,
                  noise_scheduler,
                  mask=None,
                  gt=None,
                  device=torch.device('cuda:0'),
                  num_step=1000,
                  sample_shape=(7060, 2000),
                  is_condi=False,
                  sample_intermediate=200,
                  model_pred_type: str = 'noise',
                  is_classifier_guidance=False,
                  omega=0.1,
                  *,
                  x_T=None,
                  step_noise=None,
                  seed=None,
                  use_graph=True,
                  is_tqdm=True,
                  lanes=None):
    """Reverse diffusion with the known genes re-spliced after every step (sample.py:22-101).

    Returns a host float32 ndarray [N, G].  Keyword-only extras (not in the reference): ``x_T`` / ``step_noise``
    inject the initial draw [N,G] and the per-step draws [n_steps-ish, N, G] (row i is used at timestep
    timesteps[0]-i) for bit-reproducible parity runs; otherwise noise comes from an in-kernel Philox stream keyed
    by ``seed`` (default: one draw from torch's global CPU generator per call, which advances that generator),
    the global row index and the timestep.  ``lanes``: concurrent chunk
    groups on separate HIP streams (default: engine.default_lanes; results do not depend on it)."""
    if not is_condi:
        raise ValueError("is_condi=False passes y=None to cond_layer (sample.py:13-14): unsupported by the reference "
                         "model as well")
    if model_pred_type not in ('noise', 'x_start'):
        raise NotImplementedError()
    if is_classifier_guidance:
        # sample.py:78-86 references an undefined name and the model ignores condi_flag, so cond == uncond and
        # (1+w)e - w e == e: guidance is the identity.
        warnings.warn("is_classifier_guidance has no effect: DiT_stDiff ignores condi_flag (identity guidance)")
    model.eval()
    eng = model.engine()
    dev = eng.pw.device
    N, G = int(sample_shape[0]), int(sample_shape[1])
    mask_t = torch.as_tensor(np.asarray(mask), dtype=torch.float32).to(dev).contiguous()
    gt_t = torch.as_tensor(np.asarray(gt), dtype=torch.float32).to(dev).contiguous()
    x_cond, lens = _collect_condition(dataloader, dev)
    if sum(lens) != N or gt_t.shape != (N, G) or mask_t.shape != (G,):
        raise ValueError(f"sample_shape {sample_shape} does not match dataloader rows {sum(lens)} / gt {tuple(gt_t.shape)}")
    timesteps = list(range(num_step))[::-1]
    if sample_intermediate:
        timesteps = timesteps[:sample_intermediate]
    if seed is None:
        # one draw from the global generator per call: successive calls differ (as the reference's torch.randn draws
        # do) and torch.manual_seed still fixes the sequence
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # NOT a collective: under a process group each rank that calls this samples ITS OWN matrix with its own
        # draw (fold-parallel drivers call it a different number of times per rank).  Spot-sharded sampling of ONE
        # matrix goes through parallel.sample_stDiff_sharded, whose seed is explicit and equal on every rank.
    with torch.no_grad():
        smp = E.LanedSampler(eng, noise_scheduler, lens, x_cond, gt_t, mask_t, pred_type=model_pred_type, seed=seed,
                             use_graph=use_graph, lanes=1 if step_noise is not None else lanes)
        smp.init(None if x_T is None else torch.as_tensor(x_T, dtype=torch.float32).to(dev).contiguous())
        z = None if step_noise is None else torch.as_tensor(step_noise, dtype=torch.float32).to(dev).contiguous()
        if is_tqdm:
            # progress in slices so the host never blocks the replay queue for long
            bar = tqdm(total=len(timesteps))
            sl = max(1, len(timesteps) // 20)
            done = 0
            while done < len(timesteps):
                part = timesteps[done:done + sl]
                if done == 0:
                    smp.run(part, z)
                else:
                    smp.run_more(len(part))
                done += len(part)
                torch.cuda.synchronize()
                bar.set_description_str(desc=f'time: {part[-1]}')
                bar.update(len(part))
            bar.close()
        else:
            smp.run(timesteps, z)
        recon_x = smp.x.detach().cpu().numpy()
    return recon_x
