"""CPU-only checks of the boundary and the host logic (no GPU compute calls).

  * the C-ABI library loads and exports every symbol the header declares, and the ctypes
    signature table mirrors the header one to one;
  * the host mirror of the reference interface: state_dict keys, shapes and seeded
    initialisation are identical to what the unmodified reference produced (golden state0);
  * presets carry the reference's YAML schema; config errors surface as KeyError like the
    reference's `config[...]` lookups; the product path refuses to run without a HIP device;
  * image grid / checkpoint callback behaviour (reference utils.py).
"""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden, sub

HEADER = os.path.join(ROOT, "include", "vision_amd.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(vamd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vision_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in include/vision_amd.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), (set(names) ^ set(_lib.SIGNATURES))


def test_header_argument_counts_match_ctypes_table():
    from vision_amd import _lib
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    for m in re.finditer(r"\b(vamd_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(_lib.SIGNATURES[name][1]), (name, n, len(_lib.SIGNATURES[name][1]))


def test_version_and_error_string_without_gpu():
    from vision_amd import _lib
    assert _lib.lib.vamd_version() >= 1
    # argument validation happens before any device work: a bad descriptor is reported, not run
    d = _lib.ConvDesc(B=1, H=4, W=4, Cin=8, Cout=8, ksize=5, stride=1, pad=2, upsample=0, dtype=0, algo=0)
    import ctypes as C
    rc = _lib.lib.vamd_conv2d_fwd(C.byref(d), None, None, None, None, None, None, None, None)
    assert rc != 0
    assert _lib.lib.vamd_last_error()


@pytest.mark.parametrize("name", ["t0", "t1", "t0_drop", "t0_pool"])
def test_seeded_init_and_state_dict_match_reference(name):
    """torch.manual_seed(seed) + construction consumes the RNG exactly like the reference's
    DDPMModel.__init__ (models/ddpm.py:236-262): same keys, same order, same bits."""
    import vision_amd
    t, cfg, _ = load_golden(name)
    torch.manual_seed(cfg["manual_seed"])
    model = vision_amd.get_model("ddpm", cfg)
    sd = model.state_dict()
    ref = sub(t, "state0")
    assert set(sd.keys()) == set(ref.keys())
    for k, v in sd.items():
        assert v.shape == ref[k].shape and v.dtype == ref[k].dtype, k
        assert torch.equal(v, ref[k]), k


def test_plan_matches_oracle_plan():
    """Every preset: the parameter containers generated from the layer plan carry exactly the
    state_dict keys and shapes of the oracle's template (which is pinned to the reference)."""
    import vision_amd
    from oracle import restatement as R
    from vision_amd.plan import make_plan
    from vision_amd.presets import get_preset
    for preset in ("flowers", "butterfly", "ffhq"):
        cfg = get_preset(preset)["model_config"]
        tmpl = R.state_template(cfg)
        sd = vision_amd.get_model("ddpm", cfg).state_dict()
        assert list(sd.keys()) == list(tmpl.keys()), preset
        assert all(tuple(sd[k].shape) == tuple(tmpl[k][0]) and sd[k].dtype == tmpl[k][1] for k in sd), preset
        plan = make_plan(cfg)
        assert plan.layers[0].res_in == cfg["patch_size"] and plan.layers[-1].res_out == cfg["patch_size"]
        attn = [l.res_in for l in plan.layers if l.kind == "attn"]
        assert all(r in cfg["attn_res"] or l_is_mid for r, l_is_mid in
                   zip(attn, [l.section == "Middle" for l in plan.layers if l.kind == "attn"]))


def test_presets_follow_reference_yaml_schema(tmp_path):
    import yaml
    from vision_amd.presets import get_preset, write_yaml
    for name in ("flowers", "butterfly", "ffhq"):
        p = get_preset(name)
        assert set(p) >= {"model_config", "data_config", "trainer_config", "log_config"}
        mc = p["model_config"]
        for key in ("name", "h_dims", "down_index", "attn_res", "patch_size", "learning_rate",
                    "weight_decay", "manual_seed", "diffusion_steps", "sampling_period", "grid", "with_conv"):
            assert key in mc, (name, key)
        path = tmp_path / f"{name}_ddpm.yaml"
        write_yaml(name, str(path))
        assert yaml.safe_load(open(path)) == p
    b = get_preset("butterfly")
    assert b["data_config"]["batch_size"] == 64 and b["model_config"]["patch_size"] == 224


def test_missing_config_key_raises_keyerror():
    import vision_amd
    _, cfg, _ = load_golden("t0")
    bad = {k: v for k, v in cfg.items() if k != "h_dims"}
    with pytest.raises(KeyError):
        vision_amd.get_model("ddpm", bad)
    with pytest.raises(KeyError):
        vision_amd.get_model("no_such_model", cfg)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU refusal")
def test_product_path_refuses_to_run_without_gpu():
    import vision_amd
    _, cfg, B = load_golden("t0")
    model = vision_amd.get_model("ddpm", cfg)
    x = torch.rand(B, 3, cfg["patch_size"], cfg["patch_size"])
    with pytest.raises(RuntimeError):
        model.p_loss(x)
    with pytest.raises(RuntimeError):
        model.generate(2, 2)


def test_make_grid_layout_matches_torchvision_convention():
    from vision_amd.utils import make_grid
    imgs = torch.arange(6 * 3 * 4 * 5, dtype=torch.float32).reshape(6, 3, 4, 5)
    g = make_grid(imgs, nrow=4, padding=2)
    # torchvision.utils.make_grid: xmaps = min(nrow, N), ymaps = ceil(N / xmaps), cell = size + padding
    assert g.shape == (3, 2 * (4 + 2) + 2, 4 * (5 + 2) + 2)
    assert torch.equal(g[:, 2:6, 2:7], imgs[0])
    assert torch.equal(g[:, 2:6, 9:14], imgs[1])
    assert torch.equal(g[:, 8:12, 2:7], imgs[4])
    assert float(g[:, 0, :].abs().sum()) == 0.0
    gn = make_grid(imgs, nrow=4, normalize=True)
    assert float(gn.min()) == 0.0 and abs(float(gn.max()) - 1.0) < 1e-6


def test_checkpoint_callback_keeps_best_and_last(tmp_path):
    """Reference utils.py:21-63: save on improvement of the monitored metric, keep top-k = 1,
    copy the config next to the checkpoints on train start."""
    import vision_amd
    from vision_amd.utils import SafetensorsCheckpoint
    _, cfg, _ = load_golden("t0")
    model = vision_amd.get_model("ddpm", cfg)
    cfg_path = tmp_path / "cfg.yaml"
    cfg_path.write_text("model_config: {}\n")

    class _T:
        current_epoch, global_step, is_global_zero = 0, 0, True
        callback_metrics = {}
        lightning_module = model

    cb = SafetensorsCheckpoint(str(cfg_path), str(tmp_path / "ck"), monitor="val_loss", mode="min")
    tr = _T()
    cb.on_train_start(tr, model)
    assert os.path.exists(tmp_path / "config.yaml")      # reference utils.py:93-95 copies it one level up
    for epoch, vl in enumerate([0.9, 0.5, 0.7]):
        tr.current_epoch, tr.global_step = epoch, 10 * (epoch + 1)
        tr.callback_metrics = {"val_loss": torch.tensor(vl), "VL": torch.tensor(vl)}
        cb.on_validation_end(tr, model)
    files = sorted(f for f in os.listdir(tmp_path / "ck") if f.endswith(".safetensors"))
    assert len(files) == 1 and "0.50" in files[0], files
    from safetensors.torch import load_file
    sd = load_file(str(tmp_path / "ck" / files[0]))
    ref = model.state_dict()
    assert set(sd) == set(ref)
    assert all(torch.equal(sd[k], ref[k]) for k in sd)


def test_flops_model_counts_every_conv():
    from vision_amd.flops import train_flops_per_image
    from vision_amd.presets import get_preset
    f = train_flops_per_image(get_preset("flowers")["model_config"])
    b = train_flops_per_image(get_preset("butterfly")["model_config"])
    h = train_flops_per_image(get_preset("ffhq")["model_config"])
    assert 1e11 < f < 2e11 and b == f and h > 5 * f


@pytest.mark.parametrize("n,world", [(10, 4), (7, 2), (3, 4), (8, 2), (0, 2), (5, 1)])
def test_imagefolder_shards_have_equal_length_on_every_rank(n, world):
    """Per-rank index lists like DistributedSampler(drop_last=False): padded with the head of the list to a
    multiple of the world size, so every rank runs the same number of batches (and collectives)."""
    from vision_amd.trainer import ImageFolderDataModule
    order = list(range(100, 100 + n))
    parts = [ImageFolderDataModule.shard(order, r, world) for r in range(world)]
    assert len({len(p) for p in parts}) == 1
    assert len(parts[0]) == (-(-n // world) if world > 1 else n)
    flat = [i for p in parts for i in p]
    assert set(flat) == set(order)                          # nothing dropped
    if n:
        assert max(flat.count(i) for i in order) - min(flat.count(i) for i in order) <= (1 if n >= world else world)


def test_imagefolder_datamodule_on_disk(tmp_path):
    """Class sub-directories of PNGs -> Resize + ToTensor batches in [0, 1], 80/10/10 split, and the same
    shuffled order on every rank of an epoch (the permutation is drawn from a per-epoch seed)."""
    from PIL import Image
    import numpy as np
    from vision_amd.trainer import ImageFolderDataModule
    rng = np.random.default_rng(0)
    for c in ("a", "b"):
        os.makedirs(tmp_path / c)
        for i in range(10):
            Image.fromarray(rng.integers(0, 256, (12, 9, 3), dtype=np.uint8)).save(tmp_path / c / f"{i}.png")
    dm = ImageFolderDataModule(str(tmp_path), patch_size=8, batch_size=3, use_manual_split=True)
    dm.setup()
    assert dm.num_classes == 2 and {k: len(v) for k, v in dm.splits.items()} == {"train": 16, "valid": 2, "test": 2}
    r0 = list(dm.train_dataloader(0, 2))
    r1 = list(dm.train_dataloader(1, 2))
    assert len(r0) == len(r1) == 3 and [b[0].shape[0] for b in r0] == [3, 3, 2]
    x, y = r0[0]
    assert x.shape == (3, 3, 8, 8) and x.dtype == torch.float32 and 0.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert y.dtype == torch.int64
    # the order depends on the global seed and the epoch the trainer sets, not on how often a loader was asked for
    ys = lambda it: torch.cat([b[1] for b in it]).tolist()
    torch.manual_seed(7)
    a0, a0_again = ys(dm.train_dataloader()), ys(dm.train_dataloader())
    dm.epoch = 1
    a1 = ys(dm.train_dataloader())
    dm.epoch = 0
    torch.manual_seed(8)
    b0 = ys(dm.train_dataloader())
    assert a0 == a0_again and sorted(a0) == sorted(a1) == sorted(b0)
    paths = lambda: [p for p, _ in dm.splits["train"]]
    assert len(paths()) == 16


def test_shuffle_order_follows_seed_and_epoch(tmp_path):
    from vision_amd.trainer import ImageFolderDataModule
    dm = ImageFolderDataModule.__new__(ImageFolderDataModule)
    dm.splits = {"train": [(f"img{i}", i) for i in range(40)]}
    dm.batch_size, dm.device, dm.gpu_transform, dm.epoch, dm._pipeline = 40, "cpu", False, 0, None
    dm._load = lambda path: torch.zeros(3, 2, 2)
    order = lambda: next(iter(dm.train_dataloader()))[1].tolist()
    torch.manual_seed(42)
    e0 = order()
    assert e0 == order()                         # same seed, same epoch: same order (every rank deals from it)
    dm.epoch = 3
    e3 = order()
    torch.manual_seed(43)
    dm.epoch = 0
    other = order()
    assert sorted(e0) == sorted(e3) == sorted(other) == list(range(40))
    assert e0 != e3 and e0 != other


def test_cpu_baseline_reports_its_sample():
    """bench.py's CPU leg (the oracle on the flowers model, batch 8): the object names how many steps it timed."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(seconds_budget=0.5)           # one step here; the default budget covers the 20 the config names
    assert out["kind"] == "port" and out["unit"] == "images/s" and out["steps"] >= 1 and out["batch"] == 8
    assert out["value"] > 0 and out["cores"] >= 1 and "training steps" in out["sample"]


def test_loader_lookahead_announces_each_next_batch_once():
    """vision_amd.trainer._with_lookahead: every batch is yielded once and in order; before batch i is yielded, batch i + 1
    (if there is one) has been announced exactly once; without an announcer the loader is passed through."""
    from vision_amd.trainer import _with_lookahead
    seen = []
    out = []
    for b in _with_lookahead(iter(range(5)), seen.append):
        out.append((b, list(seen)))
    assert [b for b, _ in out] == [0, 1, 2, 3, 4]
    assert [s for _, s in out] == [[1], [1, 2], [1, 2, 3], [1, 2, 3, 4], [1, 2, 3, 4]]
    assert list(_with_lookahead(iter([]), seen.append)) == [] and seen == [1, 2, 3, 4]
    assert list(_with_lookahead([7, 8], None)) == [7, 8]
