"""CPU (no GPU needed): the C-ABI library loads and exports every symbol include/stdiff_hip.h declares, the host
logic (tables, module surface, loaders, chunking, flat-buffer bucketing) behaves like the reference's, and the
product path refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from stdiff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "stdiff_hip.h")).read()
    declared = set(re.findall(r"\b(stdiff_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib._SIGS) == declared            # the ctypes table binds exactly the header's entry points
    lib16 = ctypes.CDLL(_lib.LIB_PATH_F16)           # the fp16 build: same entry points
    assert not [s for s in sorted(declared) if not hasattr(lib16, s)]
    L = _lib.lib()
    assert L.stdiff_abi_version() == 14 and L.stdiff_h16_format() == 0
    assert _lib.lib(True).stdiff_abi_version() == 14 and _lib.lib(True).stdiff_h16_format() == 1
    assert b"aligned" in L.stdiff_error_string(-3)


def test_struct_layouts_match_header():
    """ctypes mirrors of the header structs: field order and count (sizes are checked by the GPU tests running)."""
    from stdiff_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "stdiff_hip.h")).read()

    def fields(name):
        body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\} " + name + ";", hdr, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
        return out
    for cname, cls in (("stdiff_gemm_args", _lib.GemmArgs), ("stdiff_step_args", _lib.StepArgs),
                       ("stdiff_block_weights", _lib.BlockWeights), ("stdiff_dit_weights", _lib.DitWeights),
                       ("stdiff_dit_workspace", _lib.DitWorkspace), ("stdiff_adamw_args", _lib.AdamWArgs),
                       ("stdiff_train_linear", _lib.TrainLinear), ("stdiff_train_block", _lib.TrainBlock),
                       ("stdiff_train_weights", _lib.TrainWeights), ("stdiff_train_block_acts", _lib.TrainBlockActs),
                       ("stdiff_train_acts", _lib.TrainActs)):
        assert fields(cname) == [f[0].rstrip("_") if f[0] == "in_" else f[0] for f in cls._fields_], cname


def test_struct_sizes_and_offsets_match_a_c_compiler(tmp_path):
    """The header is plain C: gcc's sizeof / offsetof of every struct field against the ctypes mirrors (a mirror whose
    field types drift from the header would still have the right names)."""
    import subprocess
    from stdiff_amd import _lib
    pairs = (("stdiff_gemm_args", _lib.GemmArgs), ("stdiff_step_args", _lib.StepArgs),
             ("stdiff_block_weights", _lib.BlockWeights), ("stdiff_dit_weights", _lib.DitWeights),
             ("stdiff_dit_workspace", _lib.DitWorkspace), ("stdiff_adamw_args", _lib.AdamWArgs),
             ("stdiff_train_linear", _lib.TrainLinear), ("stdiff_train_block", _lib.TrainBlock),
             ("stdiff_train_weights", _lib.TrainWeights), ("stdiff_train_block_acts", _lib.TrainBlockActs),
             ("stdiff_train_acts", _lib.TrainActs))
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "stdiff_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f in cls._fields_:
            cf = "in" if f[0] == "in_" else f[0]
            lines.append(f'  printf("{cname}.{f[0]} %zu\\n", offsetof({cname}, {cf}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs:
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got[f"{cname}.{f[0]}"]) == getattr(cls, f[0]).offset, (cname, f[0])


def test_no_cpu_fallback():
    from stdiff_amd import ops
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    with pytest.raises(RuntimeError, match="MI355X"):
        ops.gemm(torch.zeros(4, 64), torch.zeros(4, 64))
    m = DiT_stDiff(40, 128, 1, 'dit', 4, 6)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(4, 40), torch.zeros(4, dtype=torch.long), torch.zeros(4, 40))
    ns = NoiseScheduler(10)
    with pytest.raises(RuntimeError, match="MI355X"):
        ns.add_noise(torch.zeros(4, 8), torch.zeros(4, 8), torch.zeros(4, dtype=torch.long))
    with pytest.raises(RuntimeError, match="MI355X"):
        ns.step(torch.zeros(4, 8), torch.tensor(3), torch.zeros(4, 8))


def test_module_surface_matches_reference_state_dict():
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    for G, H, depth, heads in ((40, 128, 2, 4), (1000, 1024, 6, 16)):
        if H == 1024:
            with torch.device("meta"):
                m = DiT_stDiff(G, H, depth, 'dit', heads, 6)
        else:
            m = DiT_stDiff(G, H, depth, 'dit', heads, 6)
        sd = m.state_dict()
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == O.param_shapes(G, H, depth)
        n = sum(v.numel() for v in sd.values())
        assert n == 3 * G * H + 18 * depth * H * H + 3 * H * H + (depth * 15 + 268) * H + G   # = 3GH+111H^2+358H+G at depth 6
    m = DiT_stDiff(40, 128, 2, 'dit', 4, 6)
    sd = m.state_dict()
    # reference init (stDiff_model.py:199-229): zero adaLN / output layers, zero biases, small time MLP
    for k, v in sd.items():
        if "adaLN" in k or k.startswith("out_layer.linear") or k.endswith(".bias"):
            assert float(v.abs().max()) == 0.0, k
    assert 0.01 < float(sd["time_emb.mlp.0.weight"].std()) < 0.03
    bound = (6 / (128 + 3 * 128)) ** 0.5
    assert float(sd["blks.0.attn.qkv.weight"].abs().max()) <= bound + 1e-6     # xavier_uniform
    m.load_state_dict(O.make_params(40, 128, 2, 3), strict=True)
    with pytest.raises(KeyError):
        DiT_stDiff(40, 128, 2, 'unet', 4, 6)
    with pytest.raises(AssertionError):
        DiT_stDiff(40, 130, 2, 'dit', 4, 6)
    assert m.input_size == 40 and m.hidden_size == 128 and m.depth == 2 and m.num_heads == 4 and m.classes == 6


@pytest.mark.parametrize("sched,T", [("cosine", 1000), ("cosine", 1500), ("linear", 1000), ("quadratic", 1000), ("cosine", 20)])
def test_scheduler_tables_golden(golden_dir, sched, T):
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.engine import step_coefficients
    g = np.load(os.path.join(golden_dir, "g3_tables.npz"))
    ns = NoiseScheduler(T, beta_schedule=sched)
    assert len(ns) == T
    for n in O.TABLE_NAMES:
        t = getattr(ns, n)
        assert t.dtype == torch.float32 and t.device.type == "cpu" and t.shape == (T,)
        assert np.array_equal(t.numpy(), g[f"{sched}_{T}_{n}"]), n     # same host as the recorded reference
    var = np.array([float(ns.get_variance(torch.tensor(t))) for t in range(T)])
    assert np.array_equal(var, g[f"{sched}_{T}_variance"])
    c = step_coefficients(ns)
    assert c.shape == (T, 5) and float(c[0, 4]) == 0.0
    assert np.allclose(c[1:, 4].numpy() ** 2, var[1:], rtol=1e-6)
    with pytest.raises(ValueError):
        NoiseScheduler(10, beta_schedule="sigmoid")


def test_data_loader_chunking():
    from stdiff_amd.process.data import get_data_loader
    x = np.arange(70 * 3, dtype=np.float64).reshape(70, 3)
    dl = get_data_loader(x, x + 1, batch_size=32, is_shuffle=False)
    batches = list(dl)
    assert [b[0].shape[0] for b in batches] == [32, 32, 6]           # drop_last=False (process/data.py:75)
    assert batches[0][0].dtype == torch.float32 and torch.equal(batches[2][1], torch.from_numpy(x[64:] + 1).float())
    torch.manual_seed(5)
    a = torch.cat([b[0] for b in get_data_loader(x, x, batch_size=32, is_shuffle=True)])
    assert not torch.equal(a, torch.from_numpy(x).float()) and torch.equal(a.sort(0).values, torch.from_numpy(x).float())


def test_device_loader_needs_gpu():
    from stdiff_amd.process.data import get_data_loader
    x = np.zeros((8, 4), dtype=np.float32)
    with pytest.raises(RuntimeError, match="MI355X"):
        get_data_loader(x, x, batch_size=4, device="cpu")
    with pytest.raises(ValueError):
        get_data_loader(x, x, batch_size=4, world=2)


def test_cv_gene_splits_match_sklearn():
    """the driver's fold construction is sklearn's KFold(shuffle=True, random_state) (test-stDiff.py:73)."""
    from sklearn.model_selection import KFold
    from stdiff_amd.driver import kfold_gene_splits, fold_mask
    from stdiff_amd.synth import kfold_mask
    for n, k, rs in ((1000, 5, 0), (33, 5, 0), (158, 5, 3), (40, 4, 7)):
        ref = list(KFold(n_splits=k, shuffle=True, random_state=rs).split(np.arange(n)))
        got = kfold_gene_splits(n, k, rs)
        assert len(ref) == len(got)
        for (tr, te), (gtr, gte) in zip(ref, got):
            assert np.array_equal(tr, gtr) and np.array_equal(te, gte)
        assert np.array_equal(fold_mask(n, got[0][1]), kfold_mask(n, 0, k, rs))


def test_sharding_plan():
    from stdiff_amd.parallel import shard_chunks
    lens = [512] * 39 + [32]                       # BASELINE configs[1]: 20000 rows in chunks of 512
    for world in (1, 2, 3, 8):
        plan = [shard_chunks(lens, r, world) for r in range(world)]
        assert plan[0][0] == 0 and plan[-1][1] == 20000
        assert all(plan[i][1] == plan[i + 1][0] for i in range(world - 1))       # contiguous, chunk aligned
        assert sum(len(p[2]) for p in plan) == len(lens)
        assert max(len(p[2]) for p in plan) - min(len(p[2]) for p in plan) <= 1
    assert shard_chunks([5], 1, 2) == (5, 5, [])    # more ranks than chunks: empty shard
