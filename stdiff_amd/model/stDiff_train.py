"""normal_train_stDiff with the reference signature (model/stDiff_train.py:15-87) over the HIP training engine
(stdiff_amd/train_engine.py): per batch -- noise, timesteps in [1, T), add_noise + known-gene splice, denoiser
forward, masked MSE, hand-written backward, clip_grad_norm_(1.0), AdamW(weight_decay=0)."""
import os

import torch
from tqdm import tqdm

from ..train_engine import TrainEngine


def normal_train_stDiff(model,
                        dataloader,
                        lr: float = 1e-4,
                        num_epoch: int = 1400,
                        pred_type: str = 'noise',
                        diffusion_step: int = 1000,
                        device=torch.device('cuda:0'),
                        is_tqdm: bool = True,
                        is_tune: bool = False,
                        mask=None):
    """Trains `model` in place and returns None, like the reference.  `pred_type` only labels the progress bar
    (stDiff_train.py:85).  The epoch-mean loss is read back once per epoch (the reference syncs on loss.item()
    every batch, :81).  Under an initialised torch.distributed process group every rank trains on ITS batches and
    gradients are averaged over ranks before the clip."""
    model.to(device)
    model.train()
    # lazy_update: the clip + AdamW update of iteration i runs under the forward of iteration i + 1 (same results)
    eng = TrainEngine(model, lr=lr, diffusion_step=diffusion_step, mask=mask, lazy_update=True)
    # Each iteration is replayed as one hipGraph per batch size (STDIFF_TRAIN_GRAPH=0: eager enqueue).  The graph has
    # the weight-gradient GEMMs on parallel branches (train_engine.py), which the eager enqueue cannot afford:
    # 3.42 against 3.55 ms per 512-cell step at H=1024.  With a process group the step is always eager (the bucketed
    # all-reduce is issued from the host between backward segments).
    use_graph = os.environ.get("STDIFF_TRAIN_GRAPH", "1") == "1"
    dev = eng.device
    t_epoch = tqdm(range(num_epoch), ncols=100) if is_tqdm else range(num_epoch)
    for epoch in t_epoch:
        epoch_loss = torch.zeros(1, device=dev)
        n = 0
        for i, (x, x_cond) in enumerate(dataloader):
            x = x.float().to(dev, non_blocking=True).contiguous()
            x_cond = x_cond.float().to(dev, non_blocking=True).contiguous()
            # join=False: this stream does not wait for the iteration, so the next batch is uploaded and staged while it runs
            loss, _ = eng.step(x, x_cond, use_graph=use_graph, join=False)
            eng.accumulate(epoch_loss, loss)
            n = i + 1
        if epoch + 1 == num_epoch:
            eng.finish()                        # the last iteration's update
        eng.sync()
        epoch_loss = float(epoch_loss) / max(n, 1)
        if is_tqdm:
            t_epoch.set_postfix_str(f'{pred_type} loss:{epoch_loss:.5f}')
        if is_tune:
            try:
                from ray.air import session
            except ImportError as e:      # ray is optional here; the reference imports it unconditionally (:9)
                raise ImportError("is_tune=True needs ray (ray.air.session.report)") from e
            session.report({'loss': epoch_loss})
