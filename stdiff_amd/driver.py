"""5-fold gene cross-validation driver of the reference (`diffusion_impute`, test-stDiff.py:62-187) over the MI355X
hot path: arrays in, arrays out (the reference reads .h5ad through scanpy; preprocessing stays outside).

Per fold (test-stDiff.py:73-185): KFold(n_splits, shuffle=True, random_state=rand) over the genes -> mask (test genes
0); `[0,1] -> [-1,1]` mapping of the scRNA matrix, its masked copy, and the masked spatial matrix; a fresh
DiT_stDiff under torch.manual_seed(1202); normal_train_stDiff on shuffled (seq, seq*mask) batches unless a
checkpoint `stDiff%d.pt` exists (then load_state_dict, :151); sample_stDiff over unshuffled spatial batches with
gt = masked spatial; the fold's TEST genes are scattered into the result (:181), which is mapped back to [0,1]
(:184).  Folds are independent: `folds=` selects a subset so that ranks / GPUs can split them."""
import os

import numpy as np
import torch

from .model.sample import sample_stDiff
from .model.stDiff_model import DiT_stDiff
from .model.stDiff_scheduler import NoiseScheduler
from .model.stDiff_train import normal_train_stDiff
from .process.data import get_data_loader


def kfold_gene_splits(gene_num, n_splits=5, rand=0):
    """[(train_ind, test_ind)] of sklearn's KFold(n_splits, shuffle=True, random_state=rand).split(range(gene_num))
    (test-stDiff.py:73-79): RandomState(rand) shuffles arange(n) once; fold k's test set is the k-th consecutive slice
    (the first n % n_splits folds one longer), SORTED as sklearn returns it; train = the rest, ascending."""
    idx = np.arange(gene_num)
    np.random.RandomState(rand).shuffle(idx)
    sizes = np.full(n_splits, gene_num // n_splits, dtype=int)
    sizes[: gene_num % n_splits] += 1
    out, start = [], 0
    for k in range(n_splits):
        test_mask = np.zeros(gene_num, dtype=bool)
        test_mask[idx[start:start + sizes[k]]] = True
        out.append((np.flatnonzero(~test_mask), np.flatnonzero(test_mask)))
        start += sizes[k]
    return out


def fold_mask(gene_num, test_ind):
    mask = np.ones((gene_num,), dtype='float32')                     # test-stDiff.py:92-96
    mask[test_ind] = 0
    return mask


def diffusion_impute(data_seq_array, data_spatial_array, *, lr=0.00016046744893538737, depth=6, num_epoch=900,
                     diffusion_step=1500, batch_size=512, hidden_size=1024, head=16, n_splits=5, rand=0,
                     device=torch.device('cuda:0'), save_dir=None, folds=None, compute_dtype=None, is_tqdm=False,
                     device_loader=True, sample_seed=None, sample_noise=None, _raw=False):
    """Returns (impu [cells, genes] in [0,1], with every gene imputed by the fold that held it out; columns of folds not
    in `folds` stay 0), like the reference's `impu`.  Defaults are the reference's (test-stDiff.py:27-34,64-70).

    save_dir: directory of `stDiff%d.pt` checkpoints (1-based fold index, :88); existing files are loaded instead of
    training, new ones are written after training (the reference has the save commented out; loading is what makes
    its checkpoints usable here: same state-dict keys).
    sample_noise (parity runs): callable, 1-based fold index -> (x_T [cells, genes], step_noise [steps-1, cells, genes])
    injected into sample_stDiff in place of the in-kernel Philox draws."""
    seq0 = np.ascontiguousarray(data_seq_array, dtype=np.float32)
    st0 = np.ascontiguousarray(data_spatial_array, dtype=np.float32)
    cell_num, gene_num = st0.shape
    if seq0.shape[1] != gene_num:
        raise ValueError("scRNA and spatial matrices must share their gene axis")
    splits = kfold_gene_splits(gene_num, n_splits, rand)
    torch.manual_seed(10)                                            # :75
    all_pred_res = np.zeros_like(st0)
    loader_kw = {"device": device} if device_loader else {}
    for k, (train_ind, test_ind) in enumerate(splits):
        idx = k + 1
        if folds is not None and k not in folds:
            continue
        mask = fold_mask(gene_num, test_ind)
        seq = seq0 * 2 - 1                                           # :103-110
        data_seq_masked = (seq0 * mask) * 2 - 1
        data_spatial_masked = (st0 * mask) * 2 - 1
        dataloader = get_data_loader(seq, data_seq_masked, batch_size=batch_size, is_shuffle=True, **loader_kw)
        seed = 1202                                                  # :118-121
        torch.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
        kw = {} if compute_dtype is None else {"compute_dtype": compute_dtype}
        model = DiT_stDiff(input_size=gene_num, hidden_size=hidden_size, depth=depth, num_heads=head, classes=6,
                           mlp_ratio=4.0, dit_type='dit', **kw)
        model.to(device)
        model.train()
        ckpt = None if save_dir is None else os.path.join(save_dir, 'stDiff%d.pt' % idx)
        if ckpt is not None and os.path.isfile(ckpt):
            model.load_state_dict(torch.load(ckpt, map_location=device, weights_only=True))      # :151
        else:
            normal_train_stDiff(model, dataloader=dataloader, lr=lr, num_epoch=num_epoch,
                                diffusion_step=diffusion_step, device=device, pred_type='noise', mask=mask,
                                is_tqdm=is_tqdm)
            if ckpt is not None:
                os.makedirs(save_dir, exist_ok=True)
                torch.save(model.state_dict(), ckpt)
        noise_scheduler = NoiseScheduler(num_timesteps=diffusion_step, beta_schedule='cosine')
        dataloader = get_data_loader(data_spatial_masked, data_spatial_masked, batch_size=batch_size, is_shuffle=False,
                                     **loader_kw)
        model.eval()
        inject = {}
        if sample_noise is not None:
            inject["x_T"], inject["step_noise"] = sample_noise(idx)
        imputation = sample_stDiff(model, device=device, dataloader=dataloader, noise_scheduler=noise_scheduler,
                                   mask=mask, gt=data_spatial_masked, num_step=diffusion_step,
                                   sample_shape=(cell_num, gene_num), is_condi=True,
                                   sample_intermediate=diffusion_step, model_pred_type='noise',
                                   is_classifier_guidance=False, omega=0.2, is_tqdm=is_tqdm, seed=sample_seed,
                                   **inject)
        all_pred_res[:, test_ind] = imputation[:, test_ind]          # :181
    if _raw:                      # [-1, 1] values of the folds that ran, zeros elsewhere (diffusion_impute_fold_parallel)
        return all_pred_res
    return (all_pred_res + 1) / 2                                    # :184


def diffusion_impute_fold_parallel(data_seq_array, data_spatial_array, *, group=None, n_splits=5, **kw):
    """diffusion_impute with the gene folds spread over the ranks of `group` (the folds are independent --
    each trains its own model on its own mask and imputes its own held-out genes, test-stDiff.py:79-182 -- so they are
    a second, trivially parallel axis next to spot sharding).  Rank r runs folds r, r + W, ...; every rank passes the same
    arrays and keyword arguments (`device` per rank).  The per-fold columns are merged with ONE all-reduce (each gene is
    written by exactly one fold, every other rank holds zeros there), and every rank returns the full matrix in [0, 1].
    Without a process group this is diffusion_impute.  Pass `sample_seed` for results that do not depend on the rank
    count (the default seed is drawn per call from each rank's own generator)."""
    from .parallel import _dist
    dist, rank, world = _dist(group)
    mine = [k for k in range(n_splits) if k % world == rank]
    if kw.get("folds") is not None:
        mine = [k for k in mine if k in kw["folds"]]
    kw = dict(kw, folds=mine, n_splits=n_splits, _raw=True)
    raw = diffusion_impute(data_seq_array, data_spatial_array, **kw)
    if dist is not None and world > 1:
        nccl = dist.get_backend(group) == "nccl"
        t = torch.from_numpy(np.ascontiguousarray(raw))
        if nccl:
            t = t.to(kw.get("device", torch.device("cuda:0")))
        dist.all_reduce(t, group=group)                 # x + 0 + ... + 0: exact, whatever the order
        raw = t.cpu().numpy()
    return (raw + 1) / 2
